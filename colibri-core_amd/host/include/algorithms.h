// algorithms.h — gap-mask helpers of the C++ face.
//
// Same names, arguments and results as the reference's include/algorithms.h:10-16 (bodies: src/algorithms.cpp), written from scratch. A gap mask has
// bit i set when token i of a pattern is a gap (bits 0..30). The skipgram passes of the device path enumerate the same masks
// (colibri_hip.hip gap_masks); these host functions are what callers of the reference use around computeskipgrams / PatternPointer masks.
#ifndef COLIBRI_AMD_ALGORITHMS_H
#define COLIBRI_AMD_ALGORITHMS_H
#include <cstdint>
#include <unordered_map>
#include <utility>
#include <vector>

#include "common.h"

/** (begin, length) gaps -> mask (src/algorithms.cpp:21-30) */
uint32_t vector2mask(const std::vector<std::pair<int, int>>& skips);
/** mask of an n-token pattern -> its runs of gaps as (begin, length), left to right (src/algorithms.cpp:32-53) */
std::vector<std::pair<int, int>> mask2vector(const uint32_t mask, const int n);
/** every (begin, length) with begin >= leftmargin that ends rightmargin tokens before the end, longest first per begin (src/algorithms.cpp:6-19; obsolete there) */
std::vector<std::pair<int, int>> get_consecutive_gaps(const int n, const int leftmargin = 1, const int rightmargin = 1);
/** complement of the mask, cut by shifting it n bits up and down again in 32 bits, as the reference does (src/algorithms.cpp:55-60) */
uint32_t reversemask(uint32_t mask, const unsigned int n);
/** gaps at the head / at the tail of an n-token mask (src/algorithms.cpp:62-76) */
int maskheadskip(uint32_t mask, const unsigned int n);
int masktailskip(uint32_t mask, const unsigned int n);
/** all gap masks of an n-token pattern: never at either end; at most maxskips separate gaps once n - 2 >= maxskips (src/algorithms.cpp:78-94) */
std::vector<uint32_t> compute_skip_configurations(const int n, const int maxskips);
#endif
