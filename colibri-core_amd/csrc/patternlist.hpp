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


// ---- MAXBACKOFFLENGTH < n - 1 (reference include/patternmodel.h:1139-1152): at such an order a window is counted iff every sub-pattern of
// b = MAXBACKOFFLENGTH tokens survived order b — the (n-1)-grams are not consulted, so their survivor ids cannot name the window. The candidate
// windows of the order are grouped by their bytes with the kernels above (items = positions instead of lines).
// runlen[i] = number of consecutive positions from i on whose b-gram survived (0 if the one at i did not): a window of n tokens at i is a
// candidate iff runlen[i] >= n - b + 1. One thread per run start walks its run twice; runs never leave a sentence.
__global__ __launch_bounds__(kBlock) void backoff_runs_kernel(const uint32_t* __restrict__ ids_b, uint32_t npos, uint32_t* __restrict__ runlen) {
    for (uint32_t i = blockIdx.x * kBlock + threadIdx.x; i < npos; i += gridDim.x * kBlock) {
        const bool valid = ids_b[i] != kInvalid;
        if (!valid) {
            runlen[i] = 0;
            continue;
        }
        if (i > 0 && ids_b[i - 1] != kInvalid) continue;  // inside a run: its start writes this entry
        uint32_t len = 1;
        while (i + len < npos && ids_b[i + len] != kInvalid) ++len;
        for (uint32_t k = 0; k < len; ++k) runlen[i + k] = len - k;
    }
}
// byte view of the candidate windows of order n (flen = 0: no candidate at this position), their number, and one unit of weight each
__global__ __launch_bounds__(kBlock) void backoff_select_kernel(const uint32_t* __restrict__ runlen, const uint32_t* __restrict__ tokstart, uint32_t npos, uint32_t n, uint32_t need,
                                                                uint32_t* __restrict__ flen, unsigned long long* __restrict__ off, unsigned long long* __restrict__ unit,
                                                                DevState* __restrict__ st) {
    uint32_t c = 0;
    for (uint32_t i = blockIdx.x * kBlock + threadIdx.x; i <= npos; i += gridDim.x * kBlock) {
        unit[i] = i;
        if (i == npos) break;
        const bool cand = runlen[i] >= need;
        flen[i]         = cand ? tokstart[i + n] - tokstart[i] : 0u;
        off[i]          = tokstart[i];
        c += cand;
    }
    for (int o = 32; o > 0; o >>= 1) c += __shfl_down(c, o, kWave);
    if ((threadIdx.x & (kWave - 1)) == 0 && c) atomicAdd(&st->admitted, c);
}
// keep[i] = 1 for the representative of a group that reaches the threshold (the order's results), found += groups
__global__ __launch_bounds__(kBlock) void backoff_keep_kernel(const uint32_t* __restrict__ isrep, const FSlot* __restrict__ table, const uint32_t* __restrict__ slot_of, uint32_t npos,
                                                              uint32_t threshold, uint32_t* __restrict__ keep, DevState* __restrict__ st) {
    uint32_t f = 0;
    for (uint32_t i = blockIdx.x * kBlock + threadIdx.x; i < npos; i += gridDim.x * kBlock) {
        const bool rep = isrep[i] != 0;
        keep[i]        = rep && table[slot_of[i]].cnt >= threshold;
        f += rep;
    }
    for (int o = 32; o > 0; o >>= 1) f += __shfl_down(f, o, kWave);
    if ((threadIdx.x & (kWave - 1)) == 0 && f) atomicAdd(&st->found, f);
}
// results of the kept groups, and their result index parked in the group's table slot (rep field) for the id pass
__global__ __launch_bounds__(kBlock) void backoff_results_kernel(const uint32_t* __restrict__ keep, const unsigned long long* __restrict__ rank, FSlot* __restrict__ table,
                                                                 const uint32_t* __restrict__ slot_of, uint32_t npos, uint32_t base, uint32_t res_cap, uint32_t* __restrict__ res_rep,
                                                                 uint32_t* __restrict__ res_cnt, DevState* __restrict__ st) {
    for (uint32_t i = blockIdx.x * kBlock + threadIdx.x; i < npos; i += gridDim.x * kBlock) {
        if (!keep[i]) continue;
        const uint64_t r = (uint64_t)base + rank[i];
        if (r >= res_cap) {
            st->overflow = 1;
            continue;
        }
        FSlot& sl  = table[slot_of[i]];
        res_rep[r] = i;
        res_cnt[r] = sl.cnt;
        sl.rep     = (uint32_t)r | 0x80000000u;  // marks "kept"; the low bits are the result index
    }
}
// ids[i] = result index of the window at i if its group was kept (occurrences for the forward index), kInvalid otherwise
__global__ __launch_bounds__(kBlock) void backoff_ids_kernel(const uint32_t* __restrict__ flen, const FSlot* __restrict__ table, const uint32_t* __restrict__ slot_of, uint32_t npos,
                                                             uint32_t* __restrict__ ids, DevState* __restrict__ st) {
    uint32_t v = 0;
    for (uint32_t i = blockIdx.x * kBlock + threadIdx.x; i < npos; i += gridDim.x * kBlock) {
        uint32_t id = kInvalid;
        if (flen[i]) {
            const uint32_t r = table[slot_of[i]].rep;
            if (r & 0x80000000u) id = r & 0x7FFFFFFFu;
        }
        ids[i] = id;
        v += id != kInvalid;
    }
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, kWave);
    if ((threadIdx.x & (kWave - 1)) == 0 && v) atomicAdd(&st->valid, v);
}

}  // namespace colibri
