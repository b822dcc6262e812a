// kshard2.hpp — key-sharded counting, second form (round 4): what crosses the links is 4 bytes per window out and one bit (+ 4 bytes per SURVIVING window) back.
//
// The multi-GPU form of PatternModel::train's order loop (reference include/patternmodel.h:1078-1245; the one cross-shard dependency is the global count of a candidate
// before the prune of its order, :1195-1245). Round 3's protocol (kshard.hpp) sent 8-byte (key, position) records at order 2 and 16-byte ones above, the owner ran
// level B itself, and what survived came back as 4-byte positions / 8-byte (item, id) pairs through route kernels: 2.49 GB per rank and step at 8 x 125 M tokens.
// Here every order runs on the order-2 engine (bigram2.hpp / chain.hpp: the exact key of an n-gram is (number of its leading (n-1)-gram, class of its last token)), and
//   * the SOURCE partitions its records completely — level A in the emit kernel, level B per slot as on one device, then one more pass (ks2_hist / ks2_move) that cuts
//     every (A, B) bin by the w mix bits that complete the owner's own A' bin and writes the records in (owner, A', B) order as TWO arrays: the in-bin key (<= 31 bits, 4
//     bytes: the bin fixes the rest) and the window's position. Only the keys travel, with a table of the runs' lengths; the positions stay;
//   * the OWNER counts what it receives as it lies: a final bin is eight runs, one per source, found through the scanned tables (bi2_count_kernel<.., KEY4>: no level B,
//     no partition at all on this side). Every record's entry of code_at says whether its key survived and which survivor of the bin it is;
//   * FEEDBACK per source, in the order the source sent: one bit per record, then the survivors' numbers (dense per owner) — the source walks its own send order, counts
//     bits, and has (position, number) pairs: exactly what the count kernel's position lists are on one device, so the next order's emit (chain_emit_kernel) is unchanged;
//   * EXPORTS: a kept pattern goes to the lowest rank that holds an occurrence, as (index in that rank's stream, global count); the rank looks the position up.
// Mix bit layout, from the top: [A: 8, the top w of them the owner][B: 9 - bshift][C: w][in-bin key]. The owner's final bin = (A's low 8 - w bits, B, C), read as
// (A': its top 8 bits, B': the rest) — the eight pieces of a source's (A, B) bin are neighbours in its send order.
// Global numbering of the survivors of an order: owner d's dense numbers shifted by the kept counts of the owners before it (the caller gathers them), order 2's dense
// head behind all owners — identical on every rank, so (number, class) is the same key everywhere.
#pragma once
#include "chain.hpp"
#include "kshard.hpp"

namespace colibri {

constexpr uint32_t kKs2Bins = (uint32_t)kBins * kBi2BBins;  // entries of a (source -> owner) table: (A' << 9) | B
constexpr uint32_t kKs2Raw  = 0x80000000u;                  // an export's representative that is a corpus position already (head bigrams), not an index in a stream
constexpr uint32_t kKs2Tile = 4096;                         // records per tile of the feedback kernels (128 bitmap words)

struct Ks2State {  // source side, one order (zeroed before)
    uint32_t rowsum[kKsWorld * kBins];   // records per (owner, A')
    uint32_t rowbase[kKsWorld * kBins];  // their places in the send order
    uint32_t dbase[kKsWorld + 1];        // first record of each owner's share
    uint32_t overflow;
};
struct Ks2Segs {  // the segments of a concatenation (one per peer): first record and first tile of each; [world]: the totals
    uint32_t base[kKsWorld + 1], tbase[kKsWorld + 1];
};
__device__ __forceinline__ uint32_t ks2_seg_of_tile(const Ks2Segs& sg, uint32_t tile) {
    uint32_t s = 0;
#pragma unroll
    for (int k = 1; k < kKsWorld; ++k) s += tile >= sg.tbase[k] ? 1u : 0u;  // (tbase is non-decreasing; empty segments share a value and are skipped)
    return s;
}

// the fields an order's Bi2State starts with in a key-sharded run (after the clear, before the emit kernel)
__global__ void ks2_set_kernel(Bi2State* __restrict__ bs, uint32_t cskip, uint32_t bshift) {
    bs->cskip      = cskip;
    bs->bshift_fix = bshift + 1u;
}

// ---- source: (A, B) bins -> (owner, A', B) runs ------------------------------------------------------------------------------------------------------------------------
// block a = A bin a: the (B, C) histograms level B left per slot (bi2_levelB_kernel's cbhist), summed over the A bin's sub-regions. tab: [world][kKs2Bins] records per
// (owner, A', B'). (The first version swept the records a second time for this: 0.28 ms per 85 M records.)
__global__ __launch_bounds__(kKsThreads) void ks2_hist_kernel(const uint32_t* __restrict__ cbhist, uint32_t region, const Bi2State* __restrict__ bs, uint32_t w, uint32_t nsub,
                                                               uint32_t* __restrict__ tab, Ks2State* __restrict__ ks) {
    __shared__ uint32_t histL[kKsWorld * kBi2BBins];  // [B][C] = the A bin's share of the owner's bins, in their order
    const uint32_t      a = blockIdx.x, W = 1u << w, bsh = bs->bshift, lb = 9 - bsh;  // lb: bits of B'
    const uint32_t      nE = ((uint32_t)kBi2BBins >> bsh) << w;                       // final bins of this A bin
    for (uint32_t e = threadIdx.x; e < nE; e += kKsThreads) {
        uint32_t v = 0;
        for (uint32_t s = 0; s < nsub; ++s) v += cbhist[(size_t)(s * kBins + a) * (8 * kBi2BBins) + e];
        histL[e] = v;
    }
    if (threadIdx.x < nsub && bs->curA[bi2_cur(threadIdx.x * kBins + a)] > region) ks->overflow = 1;
    __syncthreads();
    // the owner's bin (17 - bsh bits) = (alow, e): A' = its top 8 bits, B' = its low 9 - bsh; this block's bins are W whole rows
    const uint32_t d = a >> (8 - w), alow = a & ((1u << (8 - w)) - 1u);
    for (uint32_t e = threadIdx.x; e < W * kBi2BBins; e += kKsThreads) {
        const uint32_t row = e >> 9, bp = e & 511u;  // row: which of the W rows of this A bin
        tab[((size_t)d * kBins + ((alow << w) | row)) * kBi2BBins + bp] = bp < (1u << lb) ? histL[(row << lb) | bp] : 0u;
    }
    const uint32_t wave = threadIdx.x / kWave, lane = threadIdx.x & (kWave - 1);
    if (wave < W) {  // wave r: the records of A' = (alow, r)
        uint32_t v = 0;
        for (uint32_t b = lane; b < (1u << lb); b += kWave) v += histL[(wave << lb) | b];
        for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, kWave);
        if (lane == 0) ks->rowsum[d * kBins + ((alow << w) | wave)] = v;
    }
}
// one block: the rows' places, the owners' shares
__global__ __launch_bounds__(kKsThreads) void ks2_rows_kernel(Ks2State* __restrict__ ks, uint32_t world, const DevState* __restrict__ st, uint32_t* __restrict__ admitted_out) {
    __shared__ uint32_t wsumL[kKsThreads / kWave];
    static_assert(kKsWorld * kBins == 2 * kKsThreads, "two rows per lane");
    const uint32_t e0 = threadIdx.x * 2, v0 = e0 < world * kBins ? ks->rowsum[e0] : 0u, v1 = e0 + 1 < world * kBins ? ks->rowsum[e0 + 1] : 0u;
    uint32_t       total;
    const uint32_t ex = bi2_block_scan<kKsThreads>(v0 + v1, &total, wsumL);
    ks->rowbase[e0]     = ex;
    ks->rowbase[e0 + 1] = ex + v0;
    if ((e0 & (kBins - 1)) == 0) ks->dbase[e0 / kBins] = ex;  // (kBins is even: a row pair never straddles two owners)
    if (threadIdx.x == 0) {
        for (uint32_t d = world; d <= (uint32_t)kKsWorld; ++d) ks->dbase[d] = total;
        *admitted_out = st->admitted;
    }
}
// pdrop (chain_emit_kernel's records when key and position do not fit 64 bits): the position lacks the three bits above `pshift` — its bucket mod 8, which is the
// sub-region the record lies in (an XCD's blocks emit the records of the buckets dealt to it into the sub-region of its number)
__global__ __launch_bounds__(kKsThreads) void ks2_move_kernel(const unsigned long long* __restrict__ recsB, uint32_t region, const Bi2State* __restrict__ bs, uint32_t w, uint32_t nsub,
                                                               const uint32_t* __restrict__ tab, const Ks2State* __restrict__ ks, const uint32_t* __restrict__ boff, uint32_t pshift,
                                                               uint32_t pdrop, uint32_t* __restrict__ key4, uint32_t* __restrict__ posbuf) {
    __shared__ uint32_t curL[kKsWorld * kBi2BBins], inL[kBi2BBins], outL[kBi2BBins], wsumL[8];
    const uint32_t      a = blockIdx.x, W = 1u << w, bsh = bs->bshift, pb = bs->posbits, lb = 9 - bsh;
    const uint32_t      bbit = pb + bs->kbits - 17;  // B = record bits [bbit + 8 : bbit + bsh], C = the w bits below, the in-bin key what is left above the position
    const uint32_t      cbit = bbit + bsh - w;
    const uint32_t      kmask = (1u << (cbit - pb)) - 1u;  // (<= 31 bits: the caller checked)
    const uint32_t      d = a >> (8 - w), alow = a & ((1u << (8 - w)) - 1u);
    const unsigned long long pmask = (1ull << pb) - 1ull;
    for (uint32_t r = 0; r < W; ++r) {  // row r of this A bin's share = A' (alow, r): its B' bins' places
        const uint32_t row = d * kBins + ((alow << w) | r);
        if (threadIdx.x < (uint32_t)kBi2BBins) inL[threadIdx.x] = tab[(size_t)row * kBi2BBins + threadIdx.x];
        __syncthreads();
        bi2_scan512(inL, outL, wsumL);
        if (threadIdx.x < (1u << lb)) curL[(r << lb) | threadIdx.x] = ks->rowbase[row] + outL[threadIdx.x];
        __syncthreads();
    }
    // (tried: one wave per (A, B) bin with ballot ranks instead of the LDS atomics — the runs level B leaves per slot are ~50 records: 0.64 ms against 0.47 at order 2)
    (void)boff;
    for (uint32_t s = 0; s < nsub; ++s) {
        const uint32_t slot = s * kBins + a, n = min(bs->curA[bi2_cur(slot)], region);
        const size_t   base = (size_t)slot * region;
        for (uint32_t j0 = 0; j0 < n; j0 += 4 * kKsThreads) {
            unsigned long long r[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const uint32_t j = j0 + k * kKsThreads + threadIdx.x;
                r[k]             = j < n ? recsB[base + j] : 0ull;
            }
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                if (j0 + k * kKsThreads + threadIdx.x < n) {
                    const uint32_t e  = (uint32_t)(r[k] >> cbit) & ((1u << (lb + w)) - 1u);  // (B, C): the slot is sorted by B, so a wave's records fall into a few neighbouring bins
                    const uint32_t at = atomicAdd(&curL[e], 1u);
                    uint32_t       p  = (uint32_t)(r[k] & pmask);
                    if (pdrop) p = ((p >> pshift) << (pshift + 3)) | (s << pshift) | (p & ((1u << pshift) - 1u));
                    key4[at]   = (uint32_t)(r[k] >> pb) & kmask;
                    posbuf[at] = p;
                }
            }
        }
    }
}

// Round 5: one block per SLOT (sub-region, A bin) instead of one per A bin: 2048 blocks instead of 256 — a block per CU streaming 332 000 records through LDS cursors
// had nothing to hide its round trips behind (0.46 ms at order 2 for 0.68 GB in, 0.68 GB out). A slot's records start, per (B, C), where the slots before it in the
// same A bin end: the A bin's places as before (tab rows scanned, rowbase) plus the (B, C) counts level B left for the sub-regions before this one (cbhist).
__global__ __launch_bounds__(kKsThreads) void ks2_move_slot_kernel(const unsigned long long* __restrict__ recsB, uint32_t region, const Bi2State* __restrict__ bs, uint32_t w, uint32_t nsub,
                                                                    const uint32_t* __restrict__ tab, const Ks2State* __restrict__ ks, const uint32_t* __restrict__ cbhist, uint32_t pshift,
                                                                    uint32_t pdrop, uint32_t* __restrict__ key4, uint32_t* __restrict__ posbuf) {
    __shared__ uint32_t curL[kKsWorld * kBi2BBins], inL[kBi2BBins], outL[kBi2BBins], wsumL[8];
    const uint32_t      slot = blockIdx.x, s = slot / kBins, a = slot % kBins;
    if (s >= nsub) return;
    const uint32_t n = min(bs->curA[bi2_cur(slot)], region);
    if (n == 0) return;
    const uint32_t W = 1u << w, bsh = bs->bshift, pb = bs->posbits, lb = 9 - bsh;
    const uint32_t bbit = pb + bs->kbits - 17, cbit = bbit + bsh - w, kmask = (1u << (cbit - pb)) - 1u;
    const uint32_t d = a >> (8 - w), alow = a & ((1u << (8 - w)) - 1u);
    const unsigned long long pmask = (1ull << pb) - 1ull;
    for (uint32_t r = 0; r < W; ++r) {
        const uint32_t row = d * kBins + ((alow << w) | r);
        if (threadIdx.x < (uint32_t)kBi2BBins) inL[threadIdx.x] = tab[(size_t)row * kBi2BBins + threadIdx.x];
        __syncthreads();
        bi2_scan512(inL, outL, wsumL);
        if (threadIdx.x < (1u << lb)) curL[(r << lb) | threadIdx.x] = ks->rowbase[row] + outL[threadIdx.x];
        __syncthreads();
    }
    for (uint32_t e = threadIdx.x; e < (1u << (lb + w)); e += kKsThreads) {
        uint32_t before = 0;
        for (uint32_t t = 0; t < s; ++t) before += cbhist[(size_t)(t * kBins + a) * (8 * kBi2BBins) + e];
        curL[e] += before;
    }
    __syncthreads();
    const size_t base = (size_t)slot * region;
    for (uint32_t j0 = 0; j0 < n; j0 += 4 * kKsThreads) {
        unsigned long long r[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const uint32_t j = j0 + k * kKsThreads + threadIdx.x;
            r[k]             = j < n ? recsB[base + j] : 0ull;
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            if (j0 + k * kKsThreads + threadIdx.x < n) {
                const uint32_t e  = (uint32_t)(r[k] >> cbit) & ((1u << (lb + w)) - 1u);
                const uint32_t at = atomicAdd(&curL[e], 1u);
                uint32_t       p  = (uint32_t)(r[k] & pmask);
                if (pdrop) p = ((p >> pshift) << (pshift + 3)) | (s << pshift) | (p & ((1u << pshift) - 1u));
                key4[at]   = (uint32_t)(r[k] >> pb) & kmask;
                posbuf[at] = p;
            }
        }
    }
}

// ---- owner: the received tables -> the run bounds the count kernel reads --------------------------------------------------------------------------------------------------
// block a' (512 threads): per source the exclusive scan of row (source, a') -> oboff[(source * 256 + a') * 513 + b]; rowtot[source * 256 + a']
__global__ __launch_bounds__(kBi2BBins) void ks2_owner_rows_kernel(const uint32_t* __restrict__ tab_recv, uint32_t world, uint32_t* __restrict__ oboff, uint32_t* __restrict__ rowtot) {
    __shared__ uint32_t inL[kBi2BBins], outL[kBi2BBins], wsumL[8];
    const uint32_t      ap = blockIdx.x;
    for (uint32_t s = 0; s < (uint32_t)kKsWorld; ++s) {
        inL[threadIdx.x] = s < world ? tab_recv[((size_t)s * kBins + ap) * kBi2BBins + threadIdx.x] : 0u;
        __syncthreads();
        const uint32_t tot = bi2_scan512(inL, outL, wsumL);
        uint32_t* const bo = oboff + (size_t)(s * kBins + ap) * (kBi2BBins + 1);
        bo[threadIdx.x]    = outL[threadIdx.x];
        if (threadIdx.x == 0) {
            bo[kBi2BBins]            = tot;
            rowtot[s * kBins + ap] = tot;
        }
        __syncthreads();
    }
}
// ks_owner_init2_kernel for this form: Bi2State of the owner's pass (zeroed before): records per slot = source * 256 + A', their places, the agreed B-bin shift
__global__ __launch_bounds__(kKsThreads) void ks2_owner_init_kernel(Bi2State* __restrict__ obs, uint32_t* __restrict__ slotbase, const uint32_t* __restrict__ rowtot, KsBases kb,
                                                                     uint32_t bshift) {
    __shared__ uint32_t cntL[kKsSlots], offL[kKsSlots], wsumL[4];
    for (uint32_t s = threadIdx.x; s < (uint32_t)kKsSlots; s += kKsThreads) cntL[s] = rowtot[s];
    __syncthreads();
    for (int g = 0; g < kKsWorld; ++g) bi2_scan256(cntL + g * kBins, offL + g * kBins, wsumL);
    for (uint32_t s = threadIdx.x; s < (uint32_t)kKsSlots; s += kKsThreads) {
        obs->curA[bi2_cur(s)] = cntL[s];
        slotbase[s]  = kb.rbase[s >> 8] + offL[s];
    }
    if (threadIdx.x == 0) {
        obs->kbits      = 48;  // (not read: the keys are in-bin keys already)
        obs->posbits    = 31;  // a record's position = its place in the receive buffer
        obs->bshift_fix = bshift + 1u;
    }
}
// ks_compact2_kernel for this form: a representative is a place in the receive buffer -> (source << 28 | index in the source's stream); head survivors this rank
// exports carry its own lowest position, flagged
__global__ __launch_bounds__(kBlock) void ks2_compact_kernel(const uint32_t* __restrict__ sp_rep, const uint32_t* __restrict__ sp_cnt, const Bi2State* __restrict__ obs,
                                                              const Bi2State* __restrict__ sbs, const uint32_t* __restrict__ headg, KsBases kb, uint32_t world, uint32_t* __restrict__ res_rep,
                                                              uint32_t* __restrict__ res_cnt, uint32_t res_cap) {
    const uint32_t res_base = obs->res_base, lane = threadIdx.x & (kWave - 1);
    if (blockIdx.x + 1 < gridDim.x) {
        const uint32_t nwaves = (gridDim.x - 1) * (kBlock / kWave);
        for (uint32_t g = blockIdx.x * (kBlock / kWave) + threadIdx.x / kWave; g < (uint32_t)kBi2Final; g += nwaves) {
            const uint32_t f   = ((g & (uint32_t)(kBins - 1)) * kBi2BBins) + (g >> 8);
            const uint32_t off = obs->binkept[f];
            const uint32_t n   = ((f + 1 < (uint32_t)kBi2Final) ? obs->binkept[f + 1] : obs->kept_bins) - off;
            if (n == 0) continue;
            const uint32_t src = obs->binoff[f];
            for (uint32_t j = lane; j < n; j += kWave) {
                const uint32_t r = res_base + off + j;
                if (r < res_cap) {
                    const uint32_t idx = sp_rep[src + j];
                    uint32_t       s   = 0;
#pragma unroll
                    for (int k = 1; k < kKsWorld; ++k) s += ((uint32_t)k < world && idx >= kb.rbase[k]) ? 1u : 0u;
                    uint32_t rb = kb.rbase[0];
#pragma unroll
                    for (int k = 1; k < kKsWorld; ++k) rb = s == (uint32_t)k ? kb.rbase[k] : rb;
                    res_rep[r] = (s << 28) | (idx - rb);
                    res_cnt[r] = sp_cnt[src + j];
                }
            }
        }
    } else {
        uint32_t       r    = res_base + obs->kept_bins + obs->headbase[threadIdx.x];
        const uint32_t bits = reinterpret_cast<const uint16_t*>(obs->headsurv)[threadIdx.x];
#pragma unroll
        for (int q = 0; q < 16; ++q)
            if (bits & (1u << q)) {
                if (r < res_cap) {
                    res_rep[r] = kKs2Raw | ~sbs->headposinv[threadIdx.x * 16 + q];
                    res_cnt[r] = headg[threadIdx.x * 16 + q];
                }
                ++r;
            }
    }
}
// the owner's survivors -> (representative | count << 32), to the representative's rank (flagged ones: this rank itself)
struct Ks2RouteExports {
    const uint32_t* rep;
    const uint32_t* cnt;
    const uint32_t* n_dev;
    uint32_t        len, self;
    typedef unsigned long long Out;
    __device__ __forceinline__ uint32_t count(uint32_t l) const {
        const uint32_t n = *n_dev, b = l * len;
        return b < n ? min(len, n - b) : 0u;
    }
    __device__ __forceinline__ bool get(uint32_t l, uint32_t j, uint32_t& dst, Out& out) const {
        const uint32_t at = l * len + j, p = rep[at];
        dst               = (p & kKs2Raw) ? self : p >> 28;
        out               = (unsigned long long)((p & kKs2Raw) ? p : (p & 0x0FFFFFFFu)) | ((unsigned long long)cnt[at] << 32);
        return true;
    }
};

// ---- owner: feedback = per source, in stream order, one bit per record, then the survivors' dense numbers -------------------------------------------------------------
// tile = kKs2Tile records of one source's stream. tcnt[tile] = records of the tile whose key survived. One WAVE per tile, no barrier (a block-wide scan per tile for
// one number cost 1 ms per 120 M records)
__global__ __launch_bounds__(kKsThreads) void ks2_fb_count_kernel(const uint32_t* __restrict__ code_at, Ks2Segs sg, uint32_t ntiles, uint32_t* __restrict__ tcnt) {
    const uint32_t lane = threadIdx.x & (kWave - 1), nwaves = gridDim.x * (kKsThreads / kWave);
    for (uint32_t tile = blockIdx.x * (kKsThreads / kWave) + threadIdx.x / kWave; tile < ntiles; tile += nwaves) {
        const uint32_t s = ks2_seg_of_tile(sg, tile), t0 = (tile - sg.tbase[s]) * kKs2Tile, n = sg.base[s + 1] - sg.base[s];
        const uint32_t* const src = code_at + sg.base[s];
        uint32_t       c = 0;
#pragma unroll 4
        for (uint32_t k = 0; k < kKs2Tile / (4 * kWave); ++k) {
            const uint32_t j0 = t0 + (k * kWave + lane) * 4;
            if (j0 + 3 < n) {
                const uint4 v = *reinterpret_cast<const uint4*>(src + j0);
                c += (v.x != kInvalid) + (v.y != kInvalid) + (v.z != kInvalid) + (v.w != kInvalid);
            } else {
                for (uint32_t q = 0; q < 4; ++q) c += (j0 + q < n && src[j0 + q] != kInvalid) ? 1u : 0u;
            }
        }
        for (int off = 32; off > 0; off >>= 1) c += __shfl_down(c, off, kWave);
        if (lane == 0) tcnt[tile] = c;
    }
}
struct Ks2FbInfo {
    uint32_t ncodes[kKsWorld];  // survivors per source
    uint32_t off[kKsWorld + 1];  // first word of each source's section of the feedback buffer: [bits: ceil(n / 32) words][numbers]
    uint32_t kept_bins;          // this owner's survivors of the order (without the head): the caller shifts the owners' numbers by them
    uint32_t head_windows;       // order 2: the windows of the surviving head pairs, over all ranks (colibri_kshard_head_windows)
};
// after the scan of tcnt (scan[ntiles] = total): the sections' places
__global__ void ks2_fb_info_kernel(const uint32_t* __restrict__ tscan, Ks2Segs sg, uint32_t world, uint32_t ntiles, const Bi2State* __restrict__ obs, Ks2FbInfo* __restrict__ fi) {
    uint32_t off = 0;
    for (uint32_t s = 0; s < (uint32_t)kKsWorld; ++s) {
        const uint32_t c0 = tscan[min(sg.tbase[s], ntiles)], c1 = tscan[min(sg.tbase[s + 1], ntiles)];
        fi->ncodes[s] = s < world ? c1 - c0 : 0u;
        fi->off[s]    = off;
        if (s < world) off += (sg.base[s + 1] - sg.base[s] + 31) / 32 + (c1 - c0);
    }
    fi->off[kKsWorld] = off;
    fi->kept_bins     = obs->kept_bins;
    fi->head_windows  = obs->head_windows;
}
__global__ __launch_bounds__(kKsThreads) void ks2_fb_write_kernel(const uint32_t* __restrict__ code_at, Ks2Segs sg, uint32_t ntiles, const uint32_t* __restrict__ tscan,
                                                                   const Ks2FbInfo* __restrict__ fi, const Bi2State* __restrict__ obs, uint32_t* __restrict__ fb) {
    const uint32_t lane = threadIdx.x & (kWave - 1), nwaves = gridDim.x * (kKsThreads / kWave);  // one wave per tile, no barrier: 16 rows of 64 lanes x 4 records
    for (uint32_t tile = blockIdx.x * (kKsThreads / kWave) + threadIdx.x / kWave; tile < ntiles; tile += nwaves) {
        const uint32_t s = ks2_seg_of_tile(sg, tile), lt = tile - sg.tbase[s], n = sg.base[s + 1] - sg.base[s], nwords = (n + 31) / 32;
        const uint32_t* const src = code_at + sg.base[s];
        uint32_t* const       sec = fb + fi->off[s];
        uint32_t              at  = nwords + (tscan[tile] - tscan[sg.tbase[s]]);  // the tile's first number in the section
        for (uint32_t k = 0; k < kKs2Tile / (4 * kWave); ++k) {
            const uint32_t j0 = lt * kKs2Tile + (k * kWave + lane) * 4;
            uint32_t       v[4];
            if (j0 + 3 < n) {
                const uint4 e = *reinterpret_cast<const uint4*>(src + j0);
                v[0] = e.x, v[1] = e.y, v[2] = e.z, v[3] = e.w;
            } else {
                for (uint32_t q = 0; q < 4; ++q) v[q] = j0 + q < n ? src[j0 + q] : kInvalid;
            }
            uint32_t nib = 0;
#pragma unroll
            for (int q = 0; q < 4; ++q) nib |= (v[q] != kInvalid ? 1u : 0u) << q;
            uint32_t word = nib << (4 * (lane & 7u));  // eight lanes make a bitmap word
            word |= __shfl_xor(word, 1, kWave);
            word |= __shfl_xor(word, 2, kWave);
            word |= __shfl_xor(word, 4, kWave);
            const uint32_t wi = lt * (kKs2Tile / 32) + k * (kWave / 8) + lane / 8;
            if ((lane & 7u) == 0 && wi < nwords) sec[wi] = word;
            uint32_t       total;
            uint32_t       o = at + bi2_wave_excl_scan((uint32_t)__popc(nib), &total);
#pragma unroll
            for (int q = 0; q < 4; ++q)
                if (v[q] != kInvalid) sec[o++] = obs->binkept[v[q] >> 10] + (v[q] & 1023u);
            at += total;
        }
    }
}

// ---- source: feedback -> (position, global number) pairs of the surviving windows, one flat array (bi2_pospart_kernel walks it) -----------------------------------------
// tile = 1024 bitmap words (32 768 records) of one owner's section. sg: the send order's shares (base) and their tiles; sec[d]: first word of owner d's section.
struct Ks2Secs {
    uint32_t off[kKsWorld + 1];
    uint32_t gbase[kKsWorld];  // what owner d's dense numbers are shifted by
};
// tile = kKs2Tile records of one owner's share (a lane: four consecutive records = one nibble of a bitmap word): every load of a tile is independent of the others (the
// first version walked a word's bits lane by lane — a chain of ~27 dependent look-ups per lane: 1.4 ms for 85 M records)
__global__ __launch_bounds__(kKsThreads) void ks2_dec_count_kernel(const uint32_t* __restrict__ fbr, Ks2Segs sg, Ks2Secs sc, uint32_t ntiles, uint32_t* __restrict__ tcnt) {
    const uint32_t lane = threadIdx.x & (kWave - 1), nwaves = gridDim.x * (kKsThreads / kWave);  // one wave per tile (128 bitmap words), no barrier
    for (uint32_t tile = blockIdx.x * (kKsThreads / kWave) + threadIdx.x / kWave; tile < ntiles; tile += nwaves) {
        const uint32_t d = ks2_seg_of_tile(sg, tile), w0 = (tile - sg.tbase[d]) * (kKs2Tile / 32), nwords = (sg.base[d + 1] - sg.base[d] + 31) / 32;
        uint32_t       c = 0;
#pragma unroll
        for (uint32_t k = 0; k < kKs2Tile / 32 / kWave; ++k) {
            const uint32_t wi = w0 + k * kWave + lane;
            c += wi < nwords ? (uint32_t)__popc(fbr[sc.off[d] + wi]) : 0u;
        }
        for (int off = 32; off > 0; off >>= 1) c += __shfl_down(c, off, kWave);
        if (lane == 0) tcnt[tile] = c;
    }
}
// ... and straight into the position buckets: a tile's surviving windows are one tile of bi2_pospart_kernel's partition (the flat (position, number) arrays between
// the two cost a write and a read of 8 bytes per surviving window)
static_assert(kKs2Tile == kBi2Tile && kKsThreads == kBi2Threads && kBi2Per == 4, "a feedback tile is a partition tile: four consecutive records per lane");
__global__ __launch_bounds__(kBi2Threads, kBi2Threads / 128) void ks2_dec_pospart_kernel(const uint32_t* __restrict__ fbr, Ks2Segs sg, Ks2Secs sc, uint32_t ntiles,
                                                                                          const uint32_t* __restrict__ tscan, const uint32_t* __restrict__ posbuf, Bi2State* __restrict__ bs,
                                                                                          DevState* __restrict__ st, uint32_t* __restrict__ plist, Bi2Lists pl, uint32_t* __restrict__ pcode,
                                                                                          bool part = true /* false: only ids_out (an indexed model's last order) */,
                                                                                          uint32_t* __restrict__ ids_out = nullptr /* indexed models: also gid_off + number at every surviving
                                                                                              window's position (the forward index is emitted from it in position order) */,
                                                                                          uint32_t gid_off = 0) {
    __shared__ Bi2PospartLds L;
    const uint32_t           shard = blockIdx.x & (uint32_t)(kBi2Shards - 1);
    for (uint32_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const uint32_t d = ks2_seg_of_tile(sg, tile), lt = tile - sg.tbase[d], n = sg.base[d + 1] - sg.base[d], nwords = (n + 31) / 32;
        const uint32_t wi = lt * (kKs2Tile / 32) + threadIdx.x / 8, j0 = lt * kKs2Tile + threadIdx.x * 4;
        const uint32_t nib = wi < nwords ? (fbr[sc.off[d] + wi] >> (4 * (threadIdx.x & 7u))) & 15u : 0u;  // (bits beyond the share's last record are clear)
        uint32_t       p[4] = {0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu}, g[4] = {0u, 0u, 0u, 0u};
#pragma unroll
        for (int k = 0; k < 4; ++k)
            if ((nib >> k) & 1u) p[k] = posbuf[sg.base[d] + j0 + k];
        uint32_t       total;
        const uint32_t ex = bi2_block_scan<kBi2Threads>((uint32_t)__popc(nib), &total, L.wsumL);
        const uint32_t cb = sc.off[d] + nwords + (tscan[tile] - tscan[sg.tbase[d]]) + ex;  // the lane's first number in the owner's section
        uint32_t       q  = 0;
#pragma unroll
        for (int k = 0; k < 4; ++k)
            if ((nib >> k) & 1u) g[k] = sc.gbase[d] + fbr[cb + q++];
        if (ids_out != nullptr) {
#pragma unroll
            for (int k = 0; k < 4; ++k)
                if ((nib >> k) & 1u) ids_out[p[k]] = gid_off + g[k];
        }
        if (part) bi2_pospart_tile(L, p, g, shard, bs, st, plist, pl, pcode);
    }
}
// the exports this rank receives (owner by owner): (index in the stream it sent to that owner | count << 32), or a flagged position -> the result arrays
__global__ __launch_bounds__(kBlock) void ks2_append_exports_kernel(const unsigned long long* __restrict__ ex, Ks2Segs exs /* base: first export of each owner */, Ks2Segs sg /* send shares */,
                                                                     const uint32_t* __restrict__ posbuf, uint32_t* __restrict__ rep, uint32_t* __restrict__ cnt) {
    const uint32_t n = exs.base[kKsWorld];
    for (uint32_t j = blockIdx.x * kBlock + threadIdx.x; j < n; j += gridDim.x * kBlock) {
        const unsigned long long e = ex[j];
        const uint32_t           p = (uint32_t)e;
        uint32_t                 d = 0;
#pragma unroll
        for (int k = 1; k < kKsWorld; ++k) d += j >= exs.base[k] ? 1u : 0u;
        uint32_t sb = sg.base[0];
#pragma unroll
        for (int k = 1; k < kKsWorld; ++k) sb = d == (uint32_t)k ? sg.base[k] : sb;
        rep[j] = (p & kKs2Raw) ? (p & ~kKs2Raw) : posbuf[sb + p];
        cnt[j] = (uint32_t)(e >> 32);
    }
}
// ---- indexed models: the references stay on the rank that holds them, keyed by the patterns' global numbers ---------------------------------------------------------------
// order 1: ids[i] = class id of the token at i if its (global) count reaches the threshold
__global__ __launch_bounds__(kBlock) void ks2_uni_ids_kernel(const uint32_t* __restrict__ cls, const uint32_t* __restrict__ cnt1, uint32_t thr, uint32_t npos, uint32_t* __restrict__ ids) {
    for (uint32_t i = blockIdx.x * kBlock + threadIdx.x; i < npos; i += gridDim.x * kBlock) {
        const uint32_t v = cls[i];
        ids[i]           = (v && cnt1[v] >= thr) ? v : kInvalid;
    }
}
// order 2's head windows (list (shard 8, bucket): position, kBi2HeadCode | pair): ids[position] = gid_off + the pair's number, where the pair survived
__global__ __launch_bounds__(kBlock) void ks2_head_ids_kernel(const Bi2State* __restrict__ bs, const uint32_t* __restrict__ plist, const uint32_t* __restrict__ pcode, Bi2Lists pl,
                                                               uint32_t nbuckets, const uint32_t* __restrict__ headid, uint32_t gid_off, uint32_t* __restrict__ ids) {
    for (uint32_t b = blockIdx.x; b < nbuckets; b += gridDim.x) {
        uint32_t first, cap;
        bi2_list_of(pl, kBi2Shards, b, first, cap);
        const uint32_t n = min(bs->pcur[bi2_pc(kBi2Shards * kBi2Buckets + b)], cap);
        for (uint32_t j = threadIdx.x; j < n; j += kBlock) {
            const uint32_t r = headid[pcode[first + j] & 0xFFFu];
            if (r != kInvalid) ids[plist[first + j]] = gid_off + r;
        }
    }
}
// the global number of every pattern this rank just took over for export: what its representative window carries (the exporter holds an occurrence: the window is
// among the ones the feedback named). from_class (order 1, rank 0): the representative IS the class id
__global__ __launch_bounds__(kBlock) void ks2_export_gids_kernel(const uint32_t* __restrict__ rep, uint32_t n, const uint32_t* __restrict__ ids, bool from_class, uint32_t* __restrict__ gid) {
    for (uint32_t j = blockIdx.x * kBlock + threadIdx.x; j < n; j += gridDim.x * kBlock) gid[j] = from_class ? rep[j] : ids[rep[j]];
}

// order 2's head pairs: the same survivors on every rank (all-reduced counts) -> the same numbers, behind all owners'. headid[h] = number, or kInvalid
__global__ __launch_bounds__(kBlock) void ks2_headid_kernel(const uint32_t* __restrict__ headsurv /* bit h: pair h survived */, uint32_t first, uint32_t* __restrict__ headid) {
    uint32_t       hk   = 0;
    const uint32_t bits = reinterpret_cast<const uint16_t*>(headsurv)[threadIdx.x];
    static_assert(kBi2HeadN == kBlock * 16, "16 head pairs per lane");
    hk = (uint32_t)__popc(bits);
    uint32_t tot;
    uint32_t r = first + block_exclusive_scan(hk, &tot);
#pragma unroll
    for (int q = 0; q < 16; ++q) headid[threadIdx.x * 16 + q] = (bits & (1u << q)) ? r++ : kInvalid;
}

}  // namespace colibri
