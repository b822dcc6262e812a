// patternlist.hpp — one pattern per line (PatternModelOptions::DOPATTERNPERLINE, colibri-patternmodeller -L; reference
// include/patternmodel.h:1008-1009, :1055-1058): every non-empty line of at most MAXLENGTH tokens is ONE pattern, the whole line, no
// sub-n-grams, threshold 1. On the device that is a group-by over the lines with the line's bytes as key — the grouping kernels of
// flexgrams.hpp (64-bit hash of the bytes, bytes verified against the group's representative, reseeded on a collision) over
// (offset, length) views into the corpus, one pass per line length so that every result segment has one length. gfx950 only.
#pragma once
#include "flexgrams.hpp"

namespace colibri {

// line s = positions [start, delimpos[s]): first position, token count, byte offset of its first token
__global__ __launch_bounds__(kBlock) void ppl_lines_kernel(const uint32_t* __restrict__ delimpos, uint32_t nlines, const uint32_t* __restrict__ tokstart, uint32_t* __restrict__ line_pos,
                                                           uint32_t* __restrict__ line_ntok, unsigned long long* __restrict__ line_off, unsigned long long* __restrict__ unit) {
    for (uint32_t s = blockIdx.x * kBlock + threadIdx.x; s <= nlines; s += gridDim.x * kBlock) {
        unit[s] = s;  // "one reference per line": the weight the grouping kernel adds per member
        if (s == nlines) break;
        const uint32_t start = s ? delimpos[s - 1] + 1 : 0u;
        line_pos[s]          = start;
        line_ntok[s]         = delimpos[s] - start;
        line_off[s]          = tokstart[start];
    }
}
// byte length of the lines with exactly n tokens, 0 for all others (they take no part in this pass)
__global__ __launch_bounds__(kBlock) void ppl_select_kernel(const uint32_t* __restrict__ line_pos, const uint32_t* __restrict__ line_ntok, uint32_t nlines,
                                                            const uint32_t* __restrict__ tokstart, uint32_t n, uint32_t* __restrict__ flen) {
    for (uint32_t s = blockIdx.x * kBlock + threadIdx.x; s < nlines; s += gridDim.x * kBlock)
        flen[s] = line_ntok[s] == n ? tokstart[line_pos[s] + n] - tokstart[line_pos[s]] : 0u;
}
// one result per group: representative = its first line, count = its number of lines
__global__ __launch_bounds__(kBlock) void ppl_results_kernel(const uint32_t* __restrict__ isrep, const unsigned long long* __restrict__ rank, const FSlot* __restrict__ table,
                                                             const uint32_t* __restrict__ slot_of, const uint32_t* __restrict__ line_pos, uint32_t nlines, uint32_t base, uint32_t res_cap,
                                                             uint32_t* __restrict__ res_rep, uint32_t* __restrict__ res_cnt) {
    for (uint32_t s = blockIdx.x * kBlock + threadIdx.x; s < nlines; s += gridDim.x * kBlock) {
        if (!isrep[s]) continue;
        const uint64_t r = (uint64_t)base + rank[s];
        if (r >= res_cap) continue;
        res_rep[r] = line_pos[s];
        res_cnt[r] = table[slot_of[s]].cnt;
    }
}

}  // namespace colibri
