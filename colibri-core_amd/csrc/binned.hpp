// binned.hpp — radix-partition + LDS count: the order-n pass WITHOUT per-window global atomics.
//
// Device-scope atomics run memory-side on MI355X at ~18 G/s (one address: ~83 M/s), and every 16-byte slot update moves a
// 64-byte sector each way, so the open-addressed global table (kernels.hpp §2) tops out near 4 % of the HBM roofline. This
// path produces the same result with streaming traffic only:
//   emit     per 2048-window tile the block-local reduce (same LDS election as count_kernel) turns windows into records
//            {exact 64-bit key, representative position, tile count, 16 SpookyHash bits}; rep_of[i] remembers each window's
//            representative. Level A of the partition is fused in: the tile's records are counting-sorted by the top 8 hash
//            bits inside LDS and leave as one contiguous run per bin into that bin's fixed-capacity region
//   scatter  level B: the same tile-local counting sort on the next (up to) 8 hash bits inside each A region -> up to 65 536
//            final bins (fewer, fuller bins when an order has few records)
//   count    one block per final bin builds that bin's table entirely in LDS (64-bit CAS + add on LDS), applies the threshold,
//            reserves a result range with ONE global atomic, and writes the survivor id at each representative position
//   resolve  ids[i] = ids_at[rep_of[i]]   (a gather that stays inside the tile)
// Exactness is unchanged: keys are the exact (id_left, id_right) pairs; hash bits only choose bins. If a final bin ever
// holds more distinct keys than its LDS table (adversarial hash skew), `st->overflow_bin` is raised and the host re-runs
// the training with the global-table kernels.
#pragma once
#include <cstddef>
#include "kernels.hpp"

namespace colibri {

struct __attribute__((aligned(16))) Rec {
    uint64_t key;
    uint32_t pos;   // representative position (token position index)
    uint32_t meta;  // [31:16] 16 SpookyHash bits (A bin = [31:24], B bin = [23:16]); [15:0] occurrences inside the tile
};

constexpr int      kBins       = 256;                // per level
constexpr int      kFinalBins  = kBins * kBins;      // 65 536
constexpr int      kSub        = 8;                  // level A: every A bin is fed through 8 sub-regions, one per group of emit blocks (see bin_emit_kernel)
constexpr int      kASlots     = kBins * kSub;       // 2048 (A bin, sub-region) slots
#ifndef COLIBRI_SCAT_TILE
#define COLIBRI_SCAT_TILE 2048
#endif
constexpr int      kScatTile   = COLIBRI_SCAT_TILE;  // records per scatter tile (16 B each of LDS staging)
constexpr int      kScatPer    = kScatTile / kBlock; // 8 per lane (2048-record tiles: 32 KB of staging, 4 blocks per CU — 4096 ran 0.4 ms slower per order-2 pass)
constexpr int      kBinSlots   = 2048;               // LDS table of one final bin
constexpr uint32_t kBinMaxLoad = 1900;               // distinct keys a final bin may hold
constexpr uint32_t kBinBigBin  = 4096;               // records from which a final bin counts as big (twice what the count kernel holds in registers)
constexpr int      kBinBigCap  = 4096;
constexpr int      kBinQueues  = 8;

// small device-side bookkeeping of one binned pass
struct BinState {
    uint32_t nrec;               // records emitted
    uint32_t overflow_bin;       // a final bin exceeded its LDS table -> host falls back to the global-table path
    uint32_t bshift;             // level B uses 256 >> bshift sub-bins: small orders get fewer, fuller final bins
    uint32_t kept_total;         // survivors of this order (written by bin_kept_scan_kernel)
    uint32_t tprefA[kASlots + 1];  // level-B tiles per slot, exclusive scan (first array: 16-byte aligned for locate_tile's vector loads)
    uint32_t histA[kASlots];       // records per A slot (slot = sub-region * kBins + A bin: the kSub cursors of a bin lie 1 KB apart, not in one cache line)
    uint32_t offA[kASlots + 1];    // first record of the slot's region
    uint32_t curA[kASlots];        // emit cursors
    uint32_t histAt[kBins];        // records per A bin (all its sub-regions)
    uint32_t hist2[kFinalBins];  // records per final bin, then (after the scan) their offsets
    uint32_t total2;             // sum (written by the scan)
    uint32_t cur2[kFinalBins];   // level-B scatter cursors; afterwards bin_count leaves each bin's survivor count here, and the scan turns them into dense offsets
    uint32_t found_part[kBins];  // distinct keys, accumulated per A bin (a single counter would serialise 65 536 atomics)
    uint32_t kept_part[kBins];   // survivors, accumulated per A bin
    uint32_t res_base;           // first result index of this pass's survivors (written by bin_kept_scan_kernel)
    uint32_t nbig;               // final bins with more than kBinBigBin records (hot keys): listed by bin_scan2_kernel, counted first by bin_count_body
    uint32_t big[kBinBigCap];
    uint32_t nextbin[kBinQueues * 16];  // the count kernel's bin queues (one counter per 64 bytes)
    uint32_t walk_mode;          // (experiments) 1 / 2: the count kernel walks the bins in a fixed order per block / takes them from the queues, whatever their sizes
    uint32_t maxbin;             // records of the largest final bin (bin_scan2_kernel)
};

// -------------------------------------------------------------------------------------------------------------------
// emit: scan + SpookyHash + block-local reduce -> records
// -------------------------------------------------------------------------------------------------------------------
// LIST = false: one lane per corpus position (orders 1 and 2). LIST = true: one lane per entry of the active list = the positions
// that still carry a survivor id of order n-1 (orders >= 3, where only a few percent of the positions can start a window).
// Everything downstream of the key (Rec::pos, rep_of, ids_at, sp_rep) then lives in LIST-ENTRY space: the id scatter of bin_count
// and the gather of bin_resolve touch an array as small as the list (cache-resident at orders >= 4) and the resolve reads it densely;
// compact_bins / the shard kernels translate representatives back to corpus positions through the list.
// The level-A partition is fused in: the tile's records are counting-sorted by A bin inside LDS (the election arrays are dead by
// then and are reused as the staging buffer) and leave as one contiguous run per (tile, bin) into that bin's fixed-capacity
// sub-region [slot * region, (slot+1) * region) of `recs`, slot = (block group) * kBins + A bin. A sub-region that would overflow raises
// st->radix_overflow (global-table rerun).
// ids_at (optional) is reset at every record position, which spares a fill of the whole array per order.
template <class KeyFn, bool LIST>
__global__ __launch_bounds__(kBlock) void bin_emit_kernel(KeyFn keyfn, Rec* __restrict__ recs, uint32_t region, uint32_t* __restrict__ rep_of, DevState* __restrict__ st,
                                                           BinState* __restrict__ bs, uint32_t npos, const uint32_t* __restrict__ list, const uint32_t* __restrict__ nlist,
                                                           uint32_t* __restrict__ ids_at, uint8_t* __restrict__ flags_at = nullptr, uint32_t sbits = 0, uint32_t slice = 0) {
    // sbits / slice: an order of a corpus beyond ~128 M tokens is counted in 2^sbits passes, each over the windows whose hash bits [47:40] (the ones
    // below the bin bits) select `slice`: a final bin then stays within its LDS table. Every pass walks all items; rep_of is written where the window
    // belongs to the pass (and, by the first pass, where it is not admissible at all).
    if (st->done) return;
    const uint32_t nitems = LIST ? *nlist : npos;
    const uint32_t smask  = (1u << sbits) - 1u;
    // phase E (election): keyL u64[2048] | winL u32[4096]   -- 32 KB, later reused as recL Rec[2048]
    __shared__ __attribute__((aligned(16))) unsigned char rawL[kCountTile * sizeof(Rec)];
    static_assert(kCountTile * sizeof(Rec) >= kCountTile * sizeof(uint64_t) + kCountLSlot * sizeof(uint32_t), "staging buffer must cover the election arrays");
    uint64_t* const keyL = reinterpret_cast<uint64_t*>(rawL);
    uint32_t* const winL = reinterpret_cast<uint32_t*>(rawL + kCountTile * sizeof(uint64_t));
    Rec* const      recL = reinterpret_cast<Rec*>(rawL);
    __shared__ uint32_t cntL[kCountTile];
    __shared__ uint32_t histL[kBins], offL[kBins], gbaseL[kBins];
    __shared__ uint32_t redL[kBlock / kWave];
    const uint32_t ntiles = (nitems + kCountTile - 1) / kCountTile;
    uint32_t       nadm   = 0;
    for (uint32_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const uint32_t base = tile * kCountTile;
        uint64_t       key[kCountPer], hash[kCountPer];
        uint32_t       posn[kCountPer];
        bool           adm[kCountPer], other[kCountPer];  // other: admissible, but a window of another pass
        histL[threadIdx.x] = 0;
        // three separate sweeps — list entries, then the key functor's loads, then the LDS writes — so that every global load of
        // the tile is in flight before the first LDS access (the compiler does not move a load across an LDS operation)
#pragma unroll
        for (int k = 0; k < kCountPer; ++k) {
            const uint32_t j = base + k * kBlock + threadIdx.x;
            posn[k]          = LIST ? (j < nitems ? list[j] : 0u) : j;
        }
#pragma unroll
        for (int k = 0; k < kCountPer; ++k) {
            const uint32_t j = base + k * kBlock + threadIdx.x;
            key[k]           = 0;
            hash[k]          = 0;
            adm[k]           = (j < nitems) && keyfn(posn[k], npos, key[k], hash[k]);
            other[k]         = adm[k] && ((uint32_t)(hash[k] >> 40) & smask) != slice;
            if (adm[k] && slice == 0) ++nadm;  // every pass sees every window: the first one counts them
            adm[k] = adm[k] && !other[k];
        }
#pragma unroll
        for (int k = 0; k < kCountPer; ++k) {
            const uint32_t e = k * kBlock + threadIdx.x;
            cntL[e]          = 0;
            if (adm[k]) {
                keyL[e]                                     = key[k];
                winL[(uint32_t)hash[k] & (kCountLSlot - 1)] = e;
            }
        }
        __syncthreads();
        // A window whose election slot is held by a DIFFERENT key gets a second chance below: two frequent keys sharing a slot would otherwise make
        // every occurrence of the loser its own record, tile after tile, and flood one A-bin sub-region. The losers elect among themselves in another
        // slot of the same array (stale first-round winners there carry other keys and are ignored by the key check).
        uint32_t rep[kCountPer];
        bool     lost[kCountPer];
        bool     anylost = false;
#pragma unroll
        for (int k = 0; k < kCountPer; ++k) {
            const uint32_t e = k * kBlock + threadIdx.x;
            rep[k]           = e;
            lost[k]          = false;
            if (adm[k]) {
                const uint32_t w = winL[(uint32_t)hash[k] & (kCountLSlot - 1)];
                if (w != e) {
                    if (keyL[w] == key[k]) {
                        rep[k] = w;
                        atomicAdd(&cntL[w], 1u);
                    } else {
                        lost[k] = anylost = true;
                    }
                }
            }
        }
        if (__syncthreads_or(anylost)) {  // the barrier that ends the election — unless some window lost its slot to another key (rare): then a second round
#pragma unroll
            for (int k = 0; k < kCountPer; ++k)
                if (lost[k]) winL[(uint32_t)(hash[k] >> 20) & (kCountLSlot - 1)] = k * kBlock + threadIdx.x;
            __syncthreads();
#pragma unroll
            for (int k = 0; k < kCountPer; ++k) {
                if (!lost[k]) continue;
                const uint32_t e = k * kBlock + threadIdx.x;
                const uint32_t w = winL[(uint32_t)(hash[k] >> 20) & (kCountLSlot - 1)];
                if (w != e && keyL[w] == key[k]) {
                    rep[k] = w;
                    atomicAdd(&cntL[w], 1u);
                }
            }
            __syncthreads();
        }
        // election done: keyL / winL are dead from here, cntL is complete
        uint32_t rank[kCountPer];
#pragma unroll
        for (int k = 0; k < kCountPer; ++k) {
            const uint32_t e = k * kBlock + threadIdx.x, j = base + e;
            if (j < nitems && (adm[k] || (!other[k] && slice == 0))) rep_of[j] = adm[k] ? base + rep[k] : kInvalid;  // an ITEM index: the corpus position, or the list entry (LIST)
            rank[k] = kInvalid;
            if (adm[k] && rep[k] == e) rank[k] = atomicAdd(&histL[(uint32_t)(hash[k] >> 56)], 1u);
        }
        __syncthreads();
        {
            uint32_t       tot;
            const uint32_t h  = histL[threadIdx.x];
            offL[threadIdx.x] = block_exclusive_scan(h, &tot);
            uint32_t g        = 0;
            if (h) {
                // one reservation per (tile, bin). An atomic on ONE address completes every ~12 ns on MI355X whatever the load, and
                // every tile of the corpus reserves in all 256 bins: with a single cursor per bin that is 0.6 ms of serialised
                // atomics per 100 M windows. Blocks are therefore split into kSub groups, each with its own cursor and sub-region.
                const uint32_t slot = (blockIdx.x & (uint32_t)(kSub - 1)) * kBins + threadIdx.x;
                const uint32_t at   = atomicAdd(&bs->curA[slot], h);
                if (at + h > region) st->radix_overflow = 1;  // sub-region full: the host re-runs on the global table (1 = A region, 2 = final bin, 3 = id range)
                g = slot * region + min(at, region - min(region, h));
            }
            gbaseL[threadIdx.x] = g;
        }
        __syncthreads();
        uint32_t nrec_tile = 0;
#pragma unroll
        for (int k = 0; k < kCountPer; ++k) {
            if (rank[k] != kInvalid) {
                const uint32_t e  = k * kBlock + threadIdx.x;
                const uint32_t hb = (uint32_t)(hash[k] >> 48);
                Rec            r;
                r.key  = key[k];
                r.pos  = base + e;  // item index (corpus position, or list entry when LIST): what ids_at / rep_of / sp_rep are indexed by
                r.meta = (hb << 16) | (1u + cntL[e]);
                recL[offL[hb >> 8] + rank[k]] = r;
                if (ids_at != nullptr) ids_at[base + e] = kInvalid;  // "no survivor here" until bin_count says otherwise: only record positions are ever read back
                if (flags_at != nullptr) flags_at[base + e] = 0;     // flag mode (see bin_count_kernel): one byte per item instead of an id
            }
        }
        __syncthreads();
        nrec_tile = offL[kBins - 1] + histL[kBins - 1];
        for (uint32_t j = threadIdx.x; j < nrec_tile; j += kBlock) {
            const Rec      x = recL[j];
            const uint32_t a = x.meta >> 24;
            recs[gbaseL[a] + (j - offL[a])] = x;
        }
        __syncthreads();  // recL (= keyL / winL) is rewritten by the next tile
    }
    for (int off = 32; off > 0; off >>= 1) nadm += __shfl_down(nadm, off, kWave);
    if ((threadIdx.x & (kWave - 1)) == 0) redL[threadIdx.x / kWave] = nadm;
    __syncthreads();
    if (threadIdx.x == 0) {
        const uint32_t a = redL[0] + redL[1] + redL[2] + redL[3];
        if (a) atomicAdd(&st->admitted, a);
    }
}

// one block: slot record counts (left in curA by the fused emit), region bases, per-A-bin totals and the tile prefix of the level-B kernels
__global__ __launch_bounds__(kBlock) void bin_offsets_kernel(BinState* __restrict__ bs, uint32_t region) {
    // lane a owns the kSub slots of A bin a; slots are numbered sub-region-major, so the tile prefix is built one sub-region at a time
    uint32_t hsum = 0, tbase = 0;
    for (int g = 0; g < kSub; ++g) {
        const uint32_t s = g * kBins + threadIdx.x;
        const uint32_t h = min(bs->curA[s], region);
        const uint32_t t = (h + kScatTile - 1) / kScatTile;
        uint32_t       tt;
        const uint32_t tp = block_exclusive_scan(t, &tt);
        bs->histA[s]      = h;
        bs->offA[s]       = s * region;
        bs->tprefA[s]     = tbase + tp;
        tbase += tt;
        hsum += h;
    }
    uint32_t tot;
    block_exclusive_scan(hsum, &tot);
    bs->histAt[threadIdx.x] = hsum;
    const uint32_t tt       = tbase;
    if (threadIdx.x == 0) {
        bs->nrec            = tot;
        bs->offA[kASlots]   = kASlots * region;
        bs->tprefA[kASlots] = tt;
        // aim at <= ~1024 records per final bin: nB = smallest power of two >= nrec / (256 * 1024), at most 256
        uint32_t nb = 1;
        while (nb < (uint32_t)kBins && (uint64_t)nb * kBins * 1024u < tot) nb <<= 1;
        uint32_t sh = 0;
        while ((uint32_t)kBins >> sh > nb) ++sh;
        bs->bshift = sh;
    }
}

// which A bin / which tile of which of its sub-regions does block `t` own? Block-cooperative: lane l looks at slots 8l .. 8l+7
// (one round trip to the prefix table; a binary search from global memory is 11 dependent loads, ~10 us per block, which at
// 23 000 short-lived blocks per order-2 pass was most of the level-B kernels' time). Must be called by the whole block.
__device__ __forceinline__ bool locate_tile(const BinState* bs, uint32_t t, uint32_t& a, uint32_t& begin, uint32_t& end) {
    static_assert(kASlots == kBlock * 8, "one lane per 8 slots");
    static_assert(offsetof(BinState, tprefA) % 16 == 0, "vector loads");
    if (t >= bs->tprefA[kASlots]) return false;  // (block-uniform) the grid is sized for the corpus, a small order has few tiles: leave before the search
    __shared__ uint32_t slotL;
    if (threadIdx.x == 0) slotL = kInvalid;
    __syncthreads();
    const uint4    lo4 = reinterpret_cast<const uint4*>(bs->tprefA)[threadIdx.x * 2], hi4 = reinterpret_cast<const uint4*>(bs->tprefA)[threadIdx.x * 2 + 1];
    const uint32_t nxt = bs->tprefA[threadIdx.x * 8 + 8];
    const uint32_t tp[9] = {lo4.x, lo4.y, lo4.z, lo4.w, hi4.x, hi4.y, hi4.z, hi4.w, nxt};
#pragma unroll
    for (int g = 0; g < 8; ++g)
        if (tp[g] <= t && t < tp[g + 1]) slotL = threadIdx.x * 8 + g;  // at most one (lane, g): empty slots have tp[g] == tp[g+1]
    __syncthreads();
    const uint32_t slot = slotL;
    if (slot == kInvalid) return false;
    a     = slot & (uint32_t)(kBins - 1);
    begin = bs->offA[slot] + (t - bs->tprefA[slot]) * kScatTile;
    end   = min(bs->offA[slot] + bs->histA[slot], begin + (uint32_t)kScatTile);
    return true;
}

// level-B histogram: records per final bin
__global__ __launch_bounds__(kBlock) void bin_hist2_kernel(const Rec* __restrict__ recs, const DevState* __restrict__ st, BinState* __restrict__ bs) {
    if (st->done) return;
    __shared__ uint32_t histL[kBins];
    uint32_t            a, begin, end;
    if (!locate_tile(bs, blockIdx.x, a, begin, end)) return;
    histL[threadIdx.x] = 0;
    __syncthreads();
    const uint32_t bsh = bs->bshift;
    // a tile is at most kScatTile records: all of a lane's loads are issued before its first LDS atomic (the compiler keeps the
    // program order of a global load and a following LDS atomic, which would serialise them one by one)
    uint32_t m[kScatPer];
#pragma unroll
    for (int q = 0; q < kScatPer; ++q) {
        const uint32_t j = begin + q * kBlock + threadIdx.x;
        m[q]             = (j < end) ? recs[j].meta : 0u;
    }
#pragma unroll
    for (int q = 0; q < kScatPer; ++q) {
        const uint32_t j = begin + q * kBlock + threadIdx.x;
        if (j < end) atomicAdd(&histL[((m[q] >> 16) & 255u) >> bsh], 1u);
    }
    __syncthreads();
    if (histL[threadIdx.x]) atomicAdd(&bs->hist2[a * kBins + threadIdx.x], histL[threadIdx.x]);
}

// final-bin offsets: hist2 -> exclusive offsets in (A, B) order, one block per A bin (the A totals are already known)
__global__ __launch_bounds__(kBlock) void bin_scan2_kernel(BinState* __restrict__ bs) {
    const uint32_t a = blockIdx.x;
    uint32_t       before, tot;
    block_exclusive_scan(threadIdx.x < a ? bs->histAt[threadIdx.x] : 0u, &before);  // records in the A bins before this one
    const uint32_t h = bs->hist2[a * kBins + threadIdx.x];
    const uint32_t o = block_exclusive_scan(h, &tot);
    bs->hist2[a * kBins + threadIdx.x] = before + o;
    if (a == kBins - 1 && threadIdx.x == 0) bs->total2 = before + tot;
    if (h > kBinBigBin) {
        const uint32_t k = atomicAdd(&bs->nbig, 1u);
        if (k < (uint32_t)kBinBigCap) bs->big[k] = a * kBins + threadIdx.x;
        atomicMax(&bs->maxbin, h);
    }
}

// level B: tile-local counting sort + one reserved output run per (tile, sub-bin), inside each A region
__global__ __launch_bounds__(kBlock) void bin_scatter_kernel(const Rec* __restrict__ in, Rec* __restrict__ out, const DevState* __restrict__ st, BinState* __restrict__ bs) {
    if (st->done) return;
    __shared__ Rec      recL[kScatTile];
    __shared__ uint32_t histL[kBins], offL[kBins], gbaseL[kBins];
    uint32_t            a = 0, begin, end;
    if (!locate_tile(bs, blockIdx.x, a, begin, end)) return;
    const uint32_t bsh = bs->bshift;
    histL[threadIdx.x] = 0;
    __syncthreads();
    // the records travel as plain 16-byte vectors (x, y = key; z = pos; w = meta): a Rec array here ended up in scratch memory
    static_assert(sizeof(Rec) == sizeof(uint4), "a record is one 16-byte vector");
    const uint4* const in4   = reinterpret_cast<const uint4*>(in);
    uint4* const       recL4 = reinterpret_cast<uint4*>(recL);
    uint4    r[kScatPer];
    uint32_t rank[kScatPer];
#pragma unroll
    for (int q = 0; q < kScatPer; ++q) {  // loads first, LDS atomics after (see bin_hist2_kernel)
        const uint32_t j = begin + q * kBlock + threadIdx.x;
        r[q]             = (j < end) ? in4[j] : make_uint4(0u, 0u, 0u, 0u);
    }
#pragma unroll
    for (int q = 0; q < kScatPer; ++q) {
        const uint32_t j = begin + q * kBlock + threadIdx.x;
        rank[q]          = 0;
        if (j < end) {
            const uint32_t b = ((r[q].w >> 16) & 255u) >> bsh;
            rank[q]          = atomicAdd(&histL[b], 1u);
        }
    }
    __syncthreads();
    {
        uint32_t       tot;
        const uint32_t h   = histL[threadIdx.x];
        offL[threadIdx.x]  = block_exclusive_scan(h, &tot);
        uint32_t g         = 0;
        if (h) g = bs->hist2[a * kBins + threadIdx.x] + atomicAdd(&bs->cur2[a * kBins + threadIdx.x], h);  // hist2 holds offsets after the scan
        gbaseL[threadIdx.x] = g;
    }
    __syncthreads();
#pragma unroll
    for (int q = 0; q < kScatPer; ++q) {
        const uint32_t j = begin + q * kBlock + threadIdx.x;
        if (j < end) {
            const uint32_t b         = ((r[q].w >> 16) & 255u) >> bsh;
            recL4[offL[b] + rank[q]] = r[q];
        }
    }
    __syncthreads();
    const uint32_t n    = end - begin;
    uint4* const   out4 = reinterpret_cast<uint4*>(out);
    for (uint32_t j = threadIdx.x; j < n; j += kBlock) {
        const uint4    x = recL4[j];
        const uint32_t b = ((x.w >> 16) & 255u) >> bsh;
        out4[gbaseL[b] + (j - offL[b])] = x;
    }
}

// -------------------------------------------------------------------------------------------------------------------
// count: one block per final bin, table in LDS
// -------------------------------------------------------------------------------------------------------------------
// Survivors are written WITHOUT any global atomic: bin f owns the index range [hist2[f], hist2[f+1]) of the per-order sparse
// arrays (it has at least as many records as survivors); the survivor id handed to the next order is id_base + that sparse
// index — ids only have to be unique. bin_kept_scan_kernel + compact_bins_kernel turn the sparse arrays into the dense result list.
// insert one record per lane into the bin's LDS table; returns the slot (kInvalid for lanes that are idle or were folded into
// their wave's leader). Must be called by all lanes of the block together (wave ballots inside).
// wide (owner-side merge only): a record count of 0xFFFF stands for "65 535 or more — read wide[pos]".
__device__ __forceinline__ uint32_t bin_insert(const bool valid, const Rec& x, unsigned long long* keyT, uint32_t* cntT, uint32_t* repT, const uint32_t smask,
                                               const uint32_t nslots, uint32_t& nnew, uint32_t* failL, const uint32_t* __restrict__ wide = nullptr) {
    // Heavy hitters arrive as many records with the same key (one per 2048-window tile). Lanes that hold the same key as the
    // wave's first active lane fold into that lane before touching LDS: a hot bin then costs one LDS update per wave
    // instead of 64 serialised ones on the same address.
    uint32_t       cnt   = x.meta & 0xFFFFu;
    uint32_t       pos   = x.pos;
    if (wide != nullptr && valid && cnt == 0xFFFFu) cnt = wide[pos];
    bool           alive = valid;
    const uint64_t act   = __ballot(valid);
    if (act) {
        const int                leader = __builtin_ctzll(act);
        const unsigned long long k0     = __shfl((unsigned long long)x.key, leader, kWave);
        const bool               same   = valid && x.key == k0;
        const uint64_t           sm     = __ballot(same);
        if (__popcll(sm) > 1) {
            uint32_t c = same ? cnt : 0u, pmin = same ? pos : 0xFFFFFFFFu;
            for (int off = 32; off > 0; off >>= 1) {
                c += __shfl_xor(c, off, kWave);
                pmin = min(pmin, __shfl_xor(pmin, off, kWave));
            }
            if (same) {
                alive = (int)(threadIdx.x & (kWave - 1)) == leader;
                cnt   = c;
                pos   = pmin;
            }
        }
    }
    if (!alive) return kInvalid;
    uint32_t s  = (uint32_t)mix64(x.key) & smask;
    bool     ok = false;
    for (uint32_t probe = 0; probe < nslots; ++probe) {
        const unsigned long long old = atomicCAS(&keyT[s], (unsigned long long)kEmptyKey, (unsigned long long)x.key);
        if (old == kEmptyKey) {
            ++nnew;
            ok = true;
            break;
        }
        if (old == x.key) {
            ok = true;
            break;
        }
        s = (s + 1) & smask;
    }
    if (!ok) {
        *failL = 1;
        return kInvalid;
    }
    atomicAdd(&cntT[s], cnt);
    atomicMin(&repT[s], pos);  // smallest representative position: deterministic
    return s;
}

constexpr int kBinRegPer = 8;  // records per lane held in registers: bins of up to 2048 records are read from HBM exactly once
constexpr int kBinStream = 8;  // rows of 256 records in flight while a bigger bin streams the rest (one row per memory round trip: the hottest trigram of a 10^9-token
                               // corpus — ~200 000 records, one per emit tile — held its block for ~2.6 ms of a 2.8 ms kernel)

// MERGE = the owner-side merge of a sharded pass: the "positions" are indices into the received candidate list, counts may be
// wide, and instead of sparse result arrays the bin leaves, per candidate of a surviving key, (f << 11 | rank of the key among
// the bin's survivors) — which shard_reply_radix_kernel turns into a dense global id once the per-bin survivor counts are
// scanned — with kExportBitR on the lowest-numbered candidate of the key (the lowest contributing rank), whose cnt_at entry
// also receives the summed count.
constexpr uint32_t kExportBitR = 0x80000000u;
// REPLY = the owner side of a key-sharded pass (kshard.hpp): `ids_at` is an array of 64-bit replies indexed by the record's place j in `recs` (pre-filled with
// all-ones = "the key did not survive"): (the record's tagged item | survivor id << 32), the id carrying the owner's rank tag `idtag` above the sparse index —
// ks_route_* sends (item, id) back to the record's source, whose ids_at then looks as after a local count
template <bool MERGE, bool REPLY = false>
__device__ __forceinline__ void bin_count_one(const uint32_t f, const uint32_t begin, const uint32_t end, const Rec* __restrict__ recs, DevState* __restrict__ st,
                                              BinState* __restrict__ bs, uint32_t threshold, uint32_t* __restrict__ sp_rep, uint32_t* __restrict__ sp_cnt,
                                              unsigned long long* __restrict__ sp_key, uint32_t* __restrict__ ids_at, unsigned long long* keyT, uint32_t* cntT,
                                              uint32_t* repT, uint32_t* idT, uint32_t* redL, uint32_t* failL, const uint32_t* __restrict__ wide, uint32_t* __restrict__ cnt_at,
                                              uint8_t* __restrict__ flags_at, bool dense_code, uint32_t idtag = 0) {
    // all loads of the (first 2048) records are issued before anything else: a bin is latency-bound, not bandwidth-bound
    Rec xr[kBinRegPer];
#pragma unroll
    for (int q = 0; q < kBinRegPer; ++q) {
        const uint32_t j = begin + q * kBlock + threadIdx.x;
        if (j < end) xr[q] = recs[j];
    }
    // table size follows the bin: >= 2x its records (so never more than half full), a power of two in [256, 2048] — small
    // orders touch mostly tiny bins and must not pay 40 KB of LDS initialisation each
    uint32_t nslots = kBlock;
    while (nslots < (uint32_t)kBinSlots && nslots < 2u * (end - begin)) nslots <<= 1;
    const uint32_t smask = nslots - 1;
    for (uint32_t s = threadIdx.x; s < nslots; s += kBlock) {
        keyT[s] = kEmptyKey;
        cntT[s] = 0;
        repT[s] = 0xFFFFFFFFu;
    }
    if (threadIdx.x == 0) *failL = 0;
    __syncthreads();
    uint32_t nnew = 0;
    uint32_t sl[kBinRegPer];
#pragma unroll
    for (int q = 0; q < kBinRegPer; ++q) {
        const uint32_t j = begin + q * kBlock + threadIdx.x;
        sl[q]            = kInvalid;
        if (begin + q * kBlock < end) sl[q] = bin_insert(j < end, xr[q], keyT, cntT, repT, smask, nslots, nnew, failL, MERGE ? wide : nullptr);  // block-uniform guard
    }
    const uint32_t rest = begin + kBinRegPer * kBlock;  // bins larger than the register window (hot bins) stream the remainder, kBinStream rows in flight
    for (uint32_t j0 = rest; j0 < end; j0 += kBinStream * kBlock) {
        Rec x[kBinStream];
#pragma unroll
        for (int q = 0; q < kBinStream; ++q) {
            const uint32_t j = j0 + q * kBlock + threadIdx.x;
            x[q]             = Rec{};
            if (j < end) x[q] = recs[j];
        }
#pragma unroll
        for (int q = 0; q < kBinStream; ++q)
            if (j0 + q * kBlock < end) bin_insert(j0 + q * kBlock + threadIdx.x < end, x[q], keyT, cntT, repT, smask, nslots, nnew, failL, MERGE ? wide : nullptr);  // block-uniform guard
    }
    // distinct keys of this bin
    for (int off = 32; off > 0; off >>= 1) nnew += __shfl_down(nnew, off, kWave);
    if ((threadIdx.x & (kWave - 1)) == 0) redL[threadIdx.x / kWave] = nnew;
    __syncthreads();
    const uint32_t distinct = redL[0] + redL[1] + redL[2] + redL[3];
    if (*failL || distinct > kBinMaxLoad) {
        if (threadIdx.x == 0) {
            bs->overflow_bin = 1;
            st->radix_overflow       = 2;  // sticky across orders: BinState is zeroed per order
        }
        return;
    }
    // survivors -> this bin's own range of the sparse result arrays (no global atomic)
    const uint32_t per = nslots / kBlock;  // 1..8 consecutive slots per lane
    uint32_t       keep = 0;
    for (uint32_t q = 0; q < per; ++q) {
        const uint32_t s = threadIdx.x * per + q;
        keep += (keyT[s] != kEmptyKey && cntT[s] >= threshold);
    }
    uint32_t       total;
    const uint32_t excl = block_exclusive_scan(keep, &total);
    if (threadIdx.x == 0) {
        atomicAdd(&bs->found_part[f >> 8], distinct);
        bs->cur2[f] = total;  // this bin's survivors (cur2 is free after the level-B scatter; empty bins hold 0 there already)
        if (total) atomicAdd(&bs->kept_part[f >> 8], total);
    }
    const uint32_t id_base = st->id_base;  // survivor ids of this order start here
    uint32_t       r       = begin + excl;
    for (uint32_t q = 0; q < per; ++q) {
        const uint32_t s  = threadIdx.x * per + q;
        uint32_t       id = kInvalid;
        if (keyT[s] != kEmptyKey && cntT[s] >= threshold) {
            if (MERGE) {
                cnt_at[repT[s]] = cntT[s];
                id              = (f << 11) | (r - begin);
            } else {
                sp_rep[r] = repT[s];
                sp_cnt[r] = cntT[s];
                if (sp_key != nullptr) sp_key[r] = keyT[s];  // sharded runs: the sparse arrays ARE the local candidate list
                // dense_code: (bin, rank among the bin's survivors) — bin_resolve_kernel turns it into the survivor's RESULT index once the per-bin
                // survivor counts are scanned (the per-pass modes use result indices as ids: they are the pattern numbers of the forward index)
                id = dense_code ? ((f << 11) | (r - begin)) : (REPLY ? (idtag | (id_base + r)) : id_base + r);
            }
            ++r;
        }
        idT[s] = id;
    }
    __syncthreads();
    if (total == 0 || (ids_at == nullptr && flags_at == nullptr)) return;  // nothing in this bin survives (ids_at keeps the kInvalid the emit kernel wrote), or nobody needs the ids
    // survivor id at every representative position of a surviving key (ids_at was pre-filled with kInvalid)
#pragma unroll
    for (int q = 0; q < kBinRegPer; ++q) {
        const uint32_t j = begin + q * kBlock + threadIdx.x;
        if (j < end) {
            uint32_t s = sl[q];
            if (s == kInvalid) {  // folded into a wave leader: look the key up
                s = (uint32_t)mix64(xr[q].key) & smask;
                while (keyT[s] != xr[q].key) s = (s + 1) & smask;
            }
            const uint32_t id = idT[s];
            if (id != kInvalid) {
                if (REPLY)
                    reinterpret_cast<unsigned long long*>(ids_at)[j] = (unsigned long long)xr[q].pos | ((unsigned long long)id << 32);
                else if (!MERGE && flags_at != nullptr)
                    flags_at[xr[q].pos] = 1;  // flag mode: the next order builds its keys without this order's ids (KeyTrigramCls)
                else
                    ids_at[xr[q].pos] = (MERGE && xr[q].pos == repT[s]) ? (id | kExportBitR) : id;
            }
        }
    }
    for (uint32_t j0 = rest; j0 < end; j0 += kBinStream * kBlock) {
        Rec x[kBinStream];
#pragma unroll
        for (int q = 0; q < kBinStream; ++q) {
            const uint32_t j = j0 + q * kBlock + threadIdx.x;
            x[q]             = Rec{};
            if (j < end) x[q] = recs[j];
        }
#pragma unroll
        for (int q = 0; q < kBinStream; ++q) {
            const uint32_t j = j0 + q * kBlock + threadIdx.x;
            if (j < end) {
                uint32_t s = (uint32_t)mix64(x[q].key) & smask;
                while (keyT[s] != x[q].key) s = (s + 1) & smask;
                const uint32_t id = idT[s];
                if (id != kInvalid) {
                    if (REPLY)
                        reinterpret_cast<unsigned long long*>(ids_at)[j] = (unsigned long long)x[q].pos | ((unsigned long long)id << 32);
                    else if (!MERGE && flags_at != nullptr)
                        flags_at[x[q].pos] = 1;
                    else
                        ids_at[x[q].pos] = (MERGE && x[q].pos == repT[s]) ? (id | kExportBitR) : id;
                }
            }
        }
    }
}

template <bool MERGE, bool REPLY = false>
__device__ __forceinline__ void bin_count_body(const Rec* __restrict__ recs, DevState* __restrict__ st, BinState* __restrict__ bs, uint32_t threshold,
                                               uint32_t* __restrict__ sp_rep, uint32_t* __restrict__ sp_cnt, unsigned long long* __restrict__ sp_key,
                                               uint32_t* __restrict__ ids_at, const uint32_t* __restrict__ wide, uint32_t* __restrict__ cnt_at, uint8_t* __restrict__ flags_at,
                                               bool dense_code, uint32_t idtag = 0) {
    if (st->done) return;
    __shared__ unsigned long long keyT[kBinSlots];
    __shared__ uint32_t           cntT[kBinSlots], repT[kBinSlots], idT[kBinSlots];
    __shared__ uint32_t           redL[kBlock / kWave], failL;
    __shared__ uint32_t           nextL;
    auto one = [&](const uint32_t f, const bool big_pass, const bool skip_big) {
        const uint32_t begin = bs->hist2[f];
        const uint32_t end   = (f + 1 < (uint32_t)kFinalBins) ? bs->hist2[f + 1] : bs->total2;
        if (begin >= end || (!big_pass && skip_big && end - begin > kBinBigBin)) return;  // (block-uniform)
        bin_count_one<MERGE, REPLY>(f, begin, end, recs, st, bs, threshold, sp_rep, sp_cnt, sp_key, ids_at, keyT, cntT, repT, idT, redL, &failL, wide, cnt_at, flags_at, dense_code, idtag);
        __syncthreads();
    };
    // the big bins first (a hot key: one record per emit tile, ~200 000 records for the hottest trigram of 10^9 tokens), so that none of them starts when the
    // other blocks are about to finish; then the rest, handed out dynamically — the block that drew a big bin simply takes fewer of the others
    const uint32_t nbig     = bs->nbig;
    const bool     skip_big = nbig <= (uint32_t)kBinBigCap;  // (more big bins than the list holds: the walk below takes them all)
    if (skip_big)
        for (uint32_t k = blockIdx.x; k < nbig; k += gridDim.x) one(bs->big[k], true, true);
    // persistent blocks walk the final bins, eight at a time from kBinQueues queues (queue q = the walk indices g with g mod 8 == q). Small orders use only the
    // first (256 >> bshift) sub-bins of every A bin, i.e. the non-empty bins are f = a * 256 + b with b small: the walk index is the transposed one
    // (a fastest), so that consecutive non-empty bins go to different blocks.
    // A fixed share per block (walk index g = block, block + grid, ...) is balanced to one bin and costs nothing; handing the bins out from queues costs ~40 us per
    // launch at 100 M tokens (8192 tickets on 8 counters: measured 5.38 -> 5.50 ms per step) and pays only when one bin outweighs a block's whole share:
    // the queues are used when the largest bin holds more than twice the records of an average share.
    const uint32_t mode  = bs->walk_mode;
    const uint32_t nwalk = (uint32_t)kBins * ((uint32_t)kBins >> bs->bshift);  // bins (a, b) with b >= 256 >> bshift are empty by construction: the transposed walk ends before them
    if (mode == 1 || (mode == 0 && (unsigned long long)bs->maxbin * gridDim.x <= 2ull * bs->total2)) {
        for (uint32_t g = blockIdx.x; g < nwalk; g += gridDim.x) one(((g & (uint32_t)(kBins - 1)) << 8) | (g >> 8), false, skip_big);
        return;
    }
    const uint32_t q = blockIdx.x & (uint32_t)(kBinQueues - 1);
    if (threadIdx.x == 0) nextL = atomicAdd(&bs->nextbin[q * 16], 8u);
    __syncthreads();
    uint32_t t = nextL;
    while (t * kBinQueues + q < nwalk) {
        __syncthreads();  // everybody has read nextL
        if (threadIdx.x == 0) nextL = atomicAdd(&bs->nextbin[q * 16], 8u);  // the ticket after this one: on its way while these eight bins are counted
#pragma unroll 1
        for (uint32_t k = 0; k < 8; ++k) {
            const uint32_t g = (t + k) * kBinQueues + q;
            if (g < nwalk) one(((g & (uint32_t)(kBins - 1)) << 8) | (g >> 8), false, skip_big);
        }
        __syncthreads();
        t = nextL;
    }
}

__global__ __launch_bounds__(kBlock) void bin_count_kernel(const Rec* __restrict__ recs, DevState* __restrict__ st, BinState* __restrict__ bs, uint32_t threshold,
                                                            uint32_t* __restrict__ sp_rep, uint32_t* __restrict__ sp_cnt, unsigned long long* __restrict__ sp_key,
                                                            uint32_t* __restrict__ ids_at, uint8_t* __restrict__ flags_at = nullptr, bool dense_code = false) {
    bin_count_body<false>(recs, st, bs, threshold, sp_rep, sp_cnt, sp_key, ids_at, nullptr, nullptr, flags_at, dense_code);
}
// owner side of a key-sharded pass (kshard.hpp): reply[j] = (tagged item | global survivor id << 32) of record j if its key survived (pre-filled with all-ones)
__global__ __launch_bounds__(kBlock) void bin_count_reply_kernel(const Rec* __restrict__ recs, DevState* __restrict__ st, BinState* __restrict__ bs, uint32_t threshold,
                                                                  uint32_t* __restrict__ sp_rep, uint32_t* __restrict__ sp_cnt, unsigned long long* __restrict__ reply, uint32_t idtag) {
    bin_count_body<false, true>(recs, st, bs, threshold, sp_rep, sp_cnt, nullptr, reinterpret_cast<uint32_t*>(reply), nullptr, nullptr, nullptr, false, idtag);
}
// owner-side merge of a sharded n-gram pass (see bin_count_one<MERGE>)
__global__ __launch_bounds__(kBlock) void bin_merge_count_kernel(const Rec* __restrict__ recs, DevState* __restrict__ st, BinState* __restrict__ bs, uint32_t threshold,
                                                                  uint32_t* __restrict__ ids_at, const uint32_t* __restrict__ wide, uint32_t* __restrict__ cnt_at) {
    bin_count_body<true>(recs, st, bs, threshold, nullptr, nullptr, nullptr, ids_at, wide, cnt_at, nullptr, false);
}

// ---- sharded radix passes: the sparse per-bin arrays ARE the local candidate list ---------------------------------------------
// After bin_kept_scan_kernel (threshold 1: every distinct key "survives") cur2 holds dense offsets in bin order, and bins are
// A-bin-major = owner-major (owner_of): one ordered compaction yields the send buffers already grouped by owner, and the owner
// boundaries are cur2 at the first bin of each owner's A-bin block. handle = the candidate's sparse index.
__global__ __launch_bounds__(kBlock) void shard_compact_bins_kernel(const unsigned long long* __restrict__ sp_key, const uint32_t* __restrict__ sp_cnt, const BinState* __restrict__ bs,
                                                                     unsigned long long* __restrict__ keys, uint32_t* __restrict__ counts, uint32_t* __restrict__ handles) {
    const uint32_t lane = threadIdx.x & (kWave - 1), nwaves = gridDim.x * (kBlock / kWave);
    for (uint32_t g = blockIdx.x * (kBlock / kWave) + threadIdx.x / kWave; g < (uint32_t)kFinalBins; g += nwaves) {
        const uint32_t f   = ((g & (uint32_t)(kBins - 1)) << 8) | (g >> 8);
        const uint32_t off = bs->cur2[f];
        const uint32_t n   = ((f + 1 < (uint32_t)kFinalBins) ? bs->cur2[f + 1] : bs->kept_total) - off;
        if (n == 0) continue;
        const uint32_t src = bs->hist2[f];
        for (uint32_t j = lane; j < n; j += kWave) {
            keys[off + j]    = sp_key[src + j];
            counts[off + j]  = sp_cnt[src + j];
            handles[off + j] = src + j;
        }
    }
}
// bounds[r] = first candidate of owner r (r = 0 .. world), from the dense per-bin offsets
__global__ void shard_owner_bounds_kernel(const BinState* __restrict__ bs, uint32_t world, uint32_t* __restrict__ bounds) {
    const uint32_t r = threadIdx.x;
    if (r > world) return;
    const uint32_t a = (r * (uint32_t)kBins + world - 1) / world;  // smallest A bin with (a * world) >> 8 == r
    bounds[r]        = a < (uint32_t)kBins ? bs->cur2[a * kBins] : bs->kept_total;
}

// ---- owner-side merge, front and back end -----------------------------------------------------------------------------------
// merge_emit: the received candidate list (keys, counts; sources concatenated in rank order) -> records, partitioned by A bin like
// bin_emit_kernel does, but binned by a SALTED mix of the key: all keys an owner receives share their owner under the plain mix, a
// different mix spreads them over all 65 536 final bins again. No election: a source sends every key once.
constexpr uint64_t kMergeSalt = 0x9E3779B97F4A7C15ull;
__global__ __launch_bounds__(kBlock) void merge_emit_kernel(const unsigned long long* __restrict__ keys, const uint32_t* __restrict__ counts, uint32_t n, Rec* __restrict__ recs,
                                                             uint32_t region, DevState* __restrict__ st, BinState* __restrict__ bs, uint32_t* __restrict__ ids_at,
                                                             uint32_t* __restrict__ cnt_at) {
    __shared__ Rec      recL[kCountTile];
    __shared__ uint32_t histL[kBins], offL[kBins], gbaseL[kBins];
    const uint32_t      ntiles = (n + kCountTile - 1) / kCountTile;
    for (uint32_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const uint32_t base = tile * kCountTile;
        uint64_t       key[kCountPer], hash[kCountPer];
        uint32_t       cnt[kCountPer], rank[kCountPer];
        histL[threadIdx.x] = 0;
#pragma unroll
        for (int k = 0; k < kCountPer; ++k) {  // loads first
            const uint32_t j = base + k * kBlock + threadIdx.x;
            key[k]           = (j < n) ? keys[j] : 0ull;
            cnt[k]           = (j < n) ? counts[j] : 0u;
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < kCountPer; ++k) {
            const uint32_t j = base + k * kBlock + threadIdx.x;
            rank[k]          = kInvalid;
            if (j < n) {
                hash[k]   = mix64(key[k] ^ kMergeSalt);
                rank[k]   = atomicAdd(&histL[(uint32_t)(hash[k] >> 56)], 1u);
                ids_at[j] = kInvalid;
                cnt_at[j] = 0;
            }
        }
        __syncthreads();
        {
            uint32_t       tot;
            const uint32_t h  = histL[threadIdx.x];
            offL[threadIdx.x] = block_exclusive_scan(h, &tot);
            uint32_t g        = 0;
            if (h) {
                const uint32_t slot = (blockIdx.x & (uint32_t)(kSub - 1)) * kBins + threadIdx.x;
                const uint32_t at   = atomicAdd(&bs->curA[slot], h);
                if (at + h > region) st->radix_overflow = 1;
                g = slot * region + min(at, region - min(region, h));
            }
            gbaseL[threadIdx.x] = g;
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < kCountPer; ++k) {
            if (rank[k] != kInvalid) {
                const uint32_t hb = (uint32_t)(hash[k] >> 48);
                Rec            r;
                r.key  = key[k];
                r.pos  = base + k * kBlock + threadIdx.x;
                r.meta = (hb << 16) | min(cnt[k], 0xFFFFu);
                recL[offL[hb >> 8] + rank[k]] = r;
            }
        }
        __syncthreads();
        const uint32_t nrec_tile = offL[kBins - 1] + histL[kBins - 1];
        for (uint32_t j = threadIdx.x; j < nrec_tile; j += kBlock) {
            const Rec      x = recL[j];
            const uint32_t a = x.meta >> 24;
            recs[gbaseL[a] + (j - offL[a])] = x;
        }
        __syncthreads();
    }
}
// replies: dense global id = gid_base + (survivors in the bins before the candidate's bin) + its key's rank inside the bin
__global__ __launch_bounds__(kBlock) void shard_reply_radix_kernel(const uint32_t* __restrict__ ids_at, const uint32_t* __restrict__ cnt_at, uint32_t n,
                                                                    const BinState* __restrict__ bs, uint32_t gid_base, uint32_t* __restrict__ reply_gid,
                                                                    uint32_t* __restrict__ reply_cnt) {
    for (uint32_t j = blockIdx.x * kBlock + threadIdx.x; j < n; j += gridDim.x * kBlock) {
        const uint32_t e = ids_at[j];
        uint32_t       g = kInvalid;
        if (e != kInvalid) {
            const uint32_t v = e & ~kExportBitR;
            g                = (gid_base + bs->cur2[v >> 11] + (v & 2047u)) | (e & kExportBitR);
        }
        reply_gid[j] = g;
        reply_cnt[j] = cnt_at[j];
    }
}


// per-bin survivor counts (left in cur2 by bin_count) -> dense result offsets, bin by bin; kept = their total. Block a scans A bin a.
// accumulate: this pass is one of several of its order (a sliced order): its survivors follow those of the passes before it
__global__ __launch_bounds__(kBlock) void bin_kept_scan_kernel(DevState* __restrict__ st, BinState* __restrict__ bs, uint32_t res_cap, bool accumulate = false) {
    if (st->done) return;
    const uint32_t a = blockIdx.x;
    uint32_t       before, tot;
    block_exclusive_scan(threadIdx.x < a ? bs->kept_part[threadIdx.x] : 0u, &before);
    const uint32_t h = bs->cur2[a * kBins + threadIdx.x];
    const uint32_t o = block_exclusive_scan(h, &tot);
    bs->cur2[a * kBins + threadIdx.x] = before + o;
    if (a == kBins - 1 && threadIdx.x == 0) {
        const uint32_t kept = before + tot;
        bs->kept_total      = kept;
        bs->res_base        = st->res_total + (accumulate ? st->kept : 0u);
        st->kept            = (accumulate ? st->kept : 0u) + kept;
        if ((uint64_t)bs->res_base + kept > res_cap) st->overflow = 1;
    }
}
// sparse per-bin survivors -> dense result list: one wave copies one bin's run (no atomics, no scan over dead entries)
__global__ __launch_bounds__(kBlock) void compact_bins_kernel(const uint32_t* __restrict__ sp_rep, const uint32_t* __restrict__ sp_cnt, const DevState* __restrict__ st,
                                                               const BinState* __restrict__ bs, uint32_t* __restrict__ res_rep, uint32_t* __restrict__ res_cnt, uint32_t res_cap,
                                                               const uint32_t* __restrict__ list /* item index -> corpus position, or NULL */) {
    if (st->done) return;
    const uint32_t res_base = bs->res_base, lane = threadIdx.x & (kWave - 1);
    const uint32_t nwaves = gridDim.x * (kBlock / kWave);
    const uint32_t nwalk = (uint32_t)kBins * ((uint32_t)kBins >> bs->bshift);  // (the bins beyond hold nothing)
    for (uint32_t g = blockIdx.x * (kBlock / kWave) + threadIdx.x / kWave; g < nwalk; g += nwaves) {
        const uint32_t f   = ((g & (uint32_t)(kBins - 1)) << 8) | (g >> 8);  // transposed walk, as in bin_count_kernel
        const uint32_t off = bs->cur2[f];
        const uint32_t n   = ((f + 1 < (uint32_t)kFinalBins) ? bs->cur2[f + 1] : bs->kept_total) - off;
        if (n == 0) continue;
        const uint32_t src = bs->hist2[f];
        for (uint32_t j = lane; j < n; j += kWave) {
            const uint32_t r = res_base + off + j;
            if (r < res_cap) {
                const uint32_t rp = sp_rep[src + j];
                res_rep[r]        = list != nullptr ? list[rp] : rp;
                res_cnt[r] = sp_cnt[src + j];
            }
        }
    }
}

// bookkeeping between orders on the binned path: found = sum of the partial counters; survivor-id base moves on by nrec
__global__ void bin_advance_prepare_kernel(DevState* __restrict__ st, const BinState* __restrict__ bs, bool accumulate = false) {
    if (st->done) return;
    uint32_t f = accumulate ? st->found : 0u;
    for (int a = 0; a < kBins; ++a) f += bs->found_part[a];
    st->found = f;
    const uint64_t next = (uint64_t)st->id_base + bs->nrec;
    if (next >= 0xFFFFFFF0ull) st->radix_overflow = 3;  // survivor ids would wrap: the host re-runs on the global table
    st->id_base = (uint32_t)next;
}

// ids[i] = survivor id found at the window's representative position; also builds the active list for the next order
// (positions whose window survived), tile by tile with one reservation per 8192 items.
constexpr uint32_t kDecodeBaseOnDevice = 0xFFFFFFFFu;
constexpr int kResPer  = 32;
constexpr int kResTile = kBlock * kResPer;
template <bool LIST>
__global__ __launch_bounds__(kBlock) void bin_resolve_kernel(const uint32_t* __restrict__ rep_of, const uint32_t* __restrict__ ids_at, uint32_t* __restrict__ ids,
                                                              DevState* __restrict__ st, uint32_t npos, const uint32_t* __restrict__ list_in,
                                                              const uint32_t* __restrict__ nlist_in, uint32_t* __restrict__ list_out, uint32_t* __restrict__ nlist_out,
                                                              const uint32_t* __restrict__ remap, uint32_t remap_base, const BinState* __restrict__ decode = nullptr,
                                                              uint32_t decode_base = 0) {
    if (st->done) return;
    __shared__ uint32_t baseL;
    __shared__ uint32_t redL[kBlock / kWave];
    const uint32_t      nitems = LIST ? *nlist_in : npos;
    const uint32_t      ntiles = (nitems + kResTile - 1) / kResTile;
    uint32_t            nvalid = 0;
    for (uint32_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        uint32_t id[kResPer], pos[kResPer], c = 0;
        uint32_t rp[kResPer];
#pragma unroll
        for (int q = 0; q < kResPer; ++q) {  // loads, then the dependent gathers, then the stores: kResPer independent chains per lane
            const uint32_t j = tile * kResTile + q * kBlock + threadIdx.x;
            rp[q]            = (j < nitems) ? rep_of[j] : kInvalid;
            pos[q]           = (j < nitems) ? (LIST ? list_in[j] : j) : 0u;
        }
#pragma unroll
        for (int q = 0; q < kResPer; ++q) id[q] = (rp[q] != kInvalid) ? ids_at[rp[q]] : kInvalid;
        if (remap != nullptr) {
#pragma unroll
            for (int q = 0; q < kResPer; ++q)
                if (id[q] != kInvalid) id[q] = remap[id[q] - remap_base];  // sharded: local sparse id -> global survivor id
        }
        if (decode != nullptr) {
            const uint32_t dbase = decode_base == kDecodeBaseOnDevice ? decode->res_base : decode_base;  // (enqueued id-keeping runs: the pass's first result index is only known on the device)
#pragma unroll
            for (int q = 0; q < kResPer; ++q)
                if (id[q] != kInvalid) id[q] = dbase + decode->cur2[id[q] >> 11] + (id[q] & 2047u);  // (bin, rank) -> result index
        }
#pragma unroll
        for (int q = 0; q < kResPer; ++q) {
            const uint32_t j = tile * kResTile + q * kBlock + threadIdx.x;
            if (j < nitems) {
                // LIST: every listed position is written, valid or not. The next order reads ids at i and i+1 only for i on the NEW
                // list, and a surviving n-gram at i means the (n-1)-gram at i+1 survived, i.e. i+1 is on THIS list: no fill needed
                // for that reader (callers that read ids at arbitrary positions pre-fill it with kInvalid).
                ids[LIST ? pos[q] : j] = id[q];
                c += id[q] != kInvalid;
            }
        }
        nvalid += c;
        if (list_out != nullptr) {
            uint32_t       total;
            const uint32_t excl = block_exclusive_scan(c, &total);
            if (threadIdx.x == 0) baseL = total ? atomicAdd(nlist_out, total) : 0;
            __syncthreads();
            uint32_t o = baseL + excl;
#pragma unroll
            for (int q = 0; q < kResPer; ++q)
                if (id[q] != kInvalid) list_out[o++] = pos[q];
            __syncthreads();
        }
    }
    for (int off = 32; off > 0; off >>= 1) nvalid += __shfl_down(nvalid, off, kWave);
    if ((threadIdx.x & (kWave - 1)) == 0) redL[threadIdx.x / kWave] = nvalid;
    __syncthreads();
    if (threadIdx.x == 0) {
        const uint32_t v = redL[0] + redL[1] + redL[2] + redL[3];
        if (v) atomicAdd(&st->valid, v);
    }
}

// flag mode of an all-positions order (order 2 when three class ids fit one key, see KeyTrigramCls): instead of survivor ids per
// position, one byte "the window starting here survived" per position, plus the active list for the next order.
__global__ __launch_bounds__(kBlock) void bin_resolve_flags_kernel(const uint32_t* __restrict__ rep_of, const uint8_t* __restrict__ flags_at, uint8_t* __restrict__ flags,
                                                                    DevState* __restrict__ st, uint32_t npos, uint32_t* __restrict__ list_out, uint32_t* __restrict__ nlist_out) {
    if (st->done) return;
    __shared__ uint32_t baseL;
    __shared__ uint32_t redL[kBlock / kWave];
    const uint32_t      ntiles = (npos + kResTile - 1) / kResTile;
    uint32_t            nvalid = 0;
    for (uint32_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        uint32_t rp[kResPer], c = 0;
        uint8_t  v[kResPer];
#pragma unroll
        for (int q = 0; q < kResPer; ++q) {
            const uint32_t j = tile * kResTile + q * kBlock + threadIdx.x;
            rp[q]            = (j < npos) ? rep_of[j] : kInvalid;
        }
#pragma unroll
        for (int q = 0; q < kResPer; ++q) v[q] = (rp[q] != kInvalid) ? flags_at[rp[q]] : (uint8_t)0;
#pragma unroll
        for (int q = 0; q < kResPer; ++q) {
            const uint32_t j = tile * kResTile + q * kBlock + threadIdx.x;
            if (j < npos) {
                flags[j] = v[q];
                c += v[q];
            }
        }
        nvalid += c;
        if (list_out != nullptr) {
            uint32_t       total;
            const uint32_t excl = block_exclusive_scan(c, &total);
            if (threadIdx.x == 0) baseL = total ? atomicAdd(nlist_out, total) : 0;
            __syncthreads();
            uint32_t o = baseL + excl;
#pragma unroll
            for (int q = 0; q < kResPer; ++q)
                if (v[q]) list_out[o++] = tile * kResTile + q * kBlock + threadIdx.x;
            __syncthreads();
        }
    }
    for (int off = 32; off > 0; off >>= 1) nvalid += __shfl_down(nvalid, off, kWave);
    if ((threadIdx.x & (kWave - 1)) == 0) redL[threadIdx.x / kWave] = nvalid;
    __syncthreads();
    if (threadIdx.x == 0) {
        const uint32_t t = redL[0] + redL[1] + redL[2] + redL[3];
        if (t) atomicAdd(&st->valid, t);
    }
}

}  // namespace colibri
