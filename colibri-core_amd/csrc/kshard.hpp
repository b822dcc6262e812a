// kshard.hpp — key-sharded counting: the multi-GPU form of the plain n-gram run (gfx950, wave64).
//
// PatternModel::train (reference include/patternmodel.h:880-1345) has one cross-shard dependency when the corpus is cut by sentence over the GPUs of a node:
// the GLOBAL count of a candidate before the prune of its order (:1195-1245). The first sharded protocol (shard_api.inc) counted every shard locally at threshold 1,
// sent the distinct local candidates to an owner rank, merged them there and sent a global id back for EVERY candidate: 4 x the single-device work per token.
// This file is the other way round — the records travel, not the candidates:
//   source rank   scans its sentences exactly as the single-device run does (bi2_emit_kernel / bin_emit_kernel: window -> record, partitioned by the top 8 bits
//                 of the key's mix = the A bin). The top w = log2(world) of those bits name the OWNER of the key. One more partition step ("split", below) cuts every
//                 (sub-region, A bin) slot by the next w mix bits and writes the records dense and grouped by (owner, A' bin) into the send buffer, where
//                 A' = the 8 mix bits below the owner bits = the owner's own A bin. One all-to-all moves them (RCCL send/recv over xGMI, or device copies);
//   owner rank    holds every record of its 1/world of the key space and counts it with the single-device kernels unchanged in substance (level B over the
//                 received (source, A') slots -> one wave / one block per final bin -> threshold on the EXACT GLOBAL count). What goes back is only what survived:
//                 order 2: the positions of the windows whose bigram survived (-> bitmap -> active list of order 3, as on one device);
//                 order >= 3: (record, global survivor id) for the records of surviving keys (-> ids_at -> bin_resolve_kernel, as on one device);
//                 and one (representative, global count) per surviving pattern to the lowest rank that holds an occurrence: that rank exports it.
// Per order: one exchange out (8- or 16-byte records), one back (4 / 8 bytes per surviving window), two host look-ups of buffer sizes. No candidate is ever
// counted twice, no table is merged, and the order-1 pass is an all-reduce of the dense per-class count array (kernels.hpp section 2b).
// world is a power of two <= 8 (one source rank per sub-region slot of the owner's pipelines: kBi2Sub = kSub = 8).
#pragma once
#include "bigram2.hpp"

namespace colibri {

constexpr int kKsWorld   = 8;
constexpr int kKsSlots   = kBins * kKsWorld;  // 2048 = kBi2Sub * kBins = kASlots
constexpr int kKsThreads = 1024;
static_assert(kKsSlots == kASlots, "one source rank per sub-region slot");

// sender side: what one split pass leaves for the host and for the owner
struct KsSplitState {
    uint32_t hcnt[kKsSlots * kKsWorld];  // records of slot s whose split bits are c: [s * 8 + c]
    uint32_t soff[kKsSlots * kKsWorld];  // ... and where that run starts in the send buffer
    uint32_t dtab[kKsWorld * kBins];     // records per (owner, A' bin): travels with the records
    uint32_t dbase[kKsWorld + 1];        // first record of each owner's segment of the send buffer   | read by the host
    uint32_t overflow;                   // a slot outgrew its region                                   | in one copy
    uint32_t admitted;                   // windows this rank counted at this order                     | (12 words from dbase)
    uint32_t pad;
};

// a destination's share of a routed list (feedback / exports), for the host
struct KsRouteState {
    uint32_t dbase[kKsWorld + 1];
    uint32_t pad[3];
};

// ---- one tile of a partition by a small digit (<= 8 bins), ranks by wave ballots, LDS-staged coalesced runs -------------------------------------
template <class T, int PER>
struct KsTileLds {
    T        stg[kKsThreads * PER];
    uint8_t  bin[kKsThreads * PER];
    uint32_t cnt[512], off[512], wsum[8], cur[kKsWorld], gb[kKsWorld], start[kKsWorld + 1];
    uint32_t lim[kKsWorld];  // end of each bin's room in `out` (entries beyond are dropped: the caller notices from its cursors)
};
// every thread of the (1024-thread) block calls it; L.cur[b] = where bin b's next run goes in `out` (advanced here)
template <class T, int PER>
__device__ __forceinline__ void ks_partition_tile(KsTileLds<T, PER>& L, const bool (&valid)[PER], const uint32_t (&c)[PER], const T (&v)[PER], uint32_t nb, T* __restrict__ out) {
    static_assert(PER * (kKsThreads / kWave) * kKsWorld <= 512, "one counter per (bin, row, wave)");
    const uint32_t lane = threadIdx.x & (kWave - 1), wave = threadIdx.x / kWave;
    uint32_t       rk[PER];
    if (threadIdx.x < 512) L.cnt[threadIdx.x] = 0;
    __syncthreads();
#pragma unroll
    for (int k = 0; k < PER; ++k) {
        rk[k] = 0;
        for (uint32_t b = 0; b < nb; ++b) {
            const bool     mine = valid[k] && c[k] == b;
            const uint64_t m    = __ballot(mine);
            if (mine) rk[k] = (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
            if (lane == 0) L.cnt[(b * PER + k) * (kKsThreads / kWave) + wave] = (uint32_t)__popcll(m);
        }
    }
    __syncthreads();
    const uint32_t total = bi2_scan512(L.cnt, L.off, L.wsum);
    if (threadIdx.x <= nb) L.start[threadIdx.x] = threadIdx.x < nb ? L.off[threadIdx.x * PER * (kKsThreads / kWave)] : total;
#pragma unroll
    for (int k = 0; k < PER; ++k) {
        if (valid[k]) {
            const uint32_t p = L.off[(c[k] * PER + k) * (kKsThreads / kWave) + wave] + rk[k];
            L.stg[p]         = v[k];
            L.bin[p]         = (uint8_t)c[k];
        }
    }
    __syncthreads();
    if (threadIdx.x < nb) {
        L.gb[threadIdx.x] = L.cur[threadIdx.x];
        L.cur[threadIdx.x] += L.start[threadIdx.x + 1] - L.start[threadIdx.x];
    }
    __syncthreads();
    for (uint32_t j = threadIdx.x; j < total; j += kKsThreads) {
        const uint32_t b  = L.bin[j];
        const uint32_t at = L.gb[b] + (j - L.start[b]);
        if (at < L.lim[b]) out[(size_t)at] = L.stg[j];
    }
    __syncthreads();
}

// a lane's private counters of up to 8 bins, 16 bits each (a lane sees < 65 536 entries of a slot / list), and their sum over the block into LDS
struct KsPacked {
    unsigned long long a = 0, b = 0;
    __device__ __forceinline__ void add(uint32_t c) {
        if (c < 4)
            a += 1ull << (16 * c);
        else
            b += 1ull << (16 * (c - 4));
    }
    // histL: 8 zeroed LDS words; a barrier must follow before they are read
    __device__ __forceinline__ void flush(uint32_t* histL) {
        unsigned long long x = a, y = b;
        for (int off = 32; off > 0; off >>= 1) {  // 64 lanes x < 1024 entries each: the 16-bit fields hold the wave's sums (slots and lists are < 2^20 entries)
            x += __shfl_down(x, off, kWave);
            y += __shfl_down(y, off, kWave);
        }
        if ((threadIdx.x & (kWave - 1)) == 0) {
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const uint32_t u = (uint32_t)(x >> (16 * c)) & 0xFFFFu, v = (uint32_t)(y >> (16 * c)) & 0xFFFFu;
                if (u) atomicAdd(&histL[c], u);
                if (v) atomicAdd(&histL[4 + c], v);
            }
        }
    }
};

// ---- split: the sender's (sub-region, A bin) slots -> dense runs per (owner, A' bin) ---------------------------------------------------------------
// order 2 (8-byte records of bigram2.hpp, emitted with sbits = 0): record = (mix bits below the A bin) << pb | position. The split bits c are the top w of those
// mix bits; what leaves is the record an emit pass over slice `owner` of a 2^w-sliced order would have written, with the position tagged by the source rank:
//   (mix bits below c) << (pb + w) | source << pb | position      — the owner's kernels see positions of pb + w bits
struct KsSplit8 {
    static constexpr uint32_t kCurPad = kBi2CurPad;  // (the slots' record counts are Bi2State::curA: one per line)
    uint32_t w, cbit, pb, src;  // cbit = pb + K - 8 - w: the lowest split bit
    __device__ __forceinline__ uint32_t cbin(unsigned long long r) const { return (uint32_t)(r >> cbit) & ((1u << w) - 1u); }
    __device__ __forceinline__ unsigned long long out(unsigned long long r) const {
        const unsigned long long pos = r & ((1ull << pb) - 1ull), rem = (r & ((1ull << cbit) - 1ull)) >> pb;
        return (rem << (pb + w)) | ((unsigned long long)src << pb) | pos;
    }
};
// orders >= 3 (16-byte records of binned.hpp, as uint4: key, key, item index, 16 hash bits << 16 | occurrences inside the tile): the bins come from mix64(key);
// the split bits are hash bits [55 : 56 - w], and the record leaves with hash bits [63 - w : 48 - w] in its meta word (the owner's A and B digits) and the source
// rank above the item index
struct KsSplit16 {
    static constexpr uint32_t kCurPad = 1;  // (BinState::curA)
    uint32_t w, src;
    __device__ __forceinline__ uint64_t hash(const uint4& r) const { return mix64((uint64_t)r.x | ((uint64_t)r.y << 32)); }
    __device__ __forceinline__ uint32_t cbin(const uint4& r) const { return (uint32_t)(hash(r) >> (56 - w)) & ((1u << w) - 1u); }
    __device__ __forceinline__ uint4 out(const uint4& r) const {
        uint4 o = r;
        if (w) o.z |= src << (32 - w);
        o.w = ((uint32_t)(hash(r) >> (48 - w)) << 16) | (r.w & 0xFFFFu);
        return o;
    }
};

// one block per slot: how many of its records go to each split bin
template <class RecT, class Split>
__global__ __launch_bounds__(kKsThreads) void ks_split_hist_kernel(const RecT* __restrict__ recs, uint32_t region, const uint32_t* __restrict__ slotcnt, Split sp,
                                                                    KsSplitState* __restrict__ ss) {
    __shared__ uint32_t histL[kKsWorld];
    const uint32_t      slot = blockIdx.x, have = slotcnt[slot * Split::kCurPad], n = min(have, region);
    if (threadIdx.x < kKsWorld) histL[threadIdx.x] = 0;
    if ((have > region || n >= (1u << 20)) && threadIdx.x == 0) ss->overflow = 1;  // (KsPacked's 16-bit fields hold a wave's sums only below 2^20 records per slot: 489 k at 10^9 tokens)
    __syncthreads();
    KsPacked     acc;
    const size_t base = (size_t)slot * region;
    for (uint32_t j0 = 0; j0 < n; j0 += 4 * kKsThreads) {
        RecT r[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const uint32_t j = j0 + k * kKsThreads + threadIdx.x;
            if (j < n) r[k] = recs[base + j];
        }
#pragma unroll
        for (int k = 0; k < 4; ++k)
            if (j0 + k * kKsThreads + threadIdx.x < n) acc.add(sp.cbin(r[k]));
    }
    acc.flush(histL);
    __syncthreads();
    if (threadIdx.x < kKsWorld) ss->hcnt[slot * kKsWorld + threadIdx.x] = histL[threadIdx.x];
}

// one block: the runs' places in the send buffer, in (owner, A' bin, sub-region) order; per-owner tables and segment bases
__global__ __launch_bounds__(kKsThreads) void ks_split_scan_kernel(KsSplitState* __restrict__ ss, uint32_t w, uint32_t world, const DevState* __restrict__ st) {
    __shared__ uint32_t wsumL[kKsThreads / kWave];
    const uint32_t      e0 = threadIdx.x * 16, nent = world * (uint32_t)(kBins * kKsWorld);
    uint32_t            v[16], idx[16], s = 0;
    // (all sixteen loads of a lane are issued before the first is used: no branch between them — entries beyond the run's ranks read slot 0 and count as zero)
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        const uint32_t e = min(e0 + k, nent - 1);
        const uint32_t dap = e >> 3, sub = e & 7u, d = dap >> 8, ap = dap & 255u, a = ap >> w, c = ap & ((1u << w) - 1u);
        const uint32_t A = (d << (8 - w)) | a;  // the sender's A bin: owner bits on top
        idx[k]           = (sub * kBins + A) * kKsWorld + c;
    }
#pragma unroll
    for (int k = 0; k < 16; ++k) v[k] = ss->hcnt[idx[k]];
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        if (e0 + k >= nent) v[k] = 0;
        s += v[k];
    }
    uint32_t       total;
    const uint32_t excl = bi2_block_scan<kKsThreads>(s, &total, wsumL);
    uint32_t       run  = excl;
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        if (e0 + k < nent) ss->soff[idx[k]] = run;
        run += v[k];
    }
    if (e0 < nent) {
        uint32_t t0 = 0, t1 = 0;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            t0 += v[k];
            t1 += v[8 + k];
        }
        ss->dtab[e0 >> 3]       = t0;
        ss->dtab[(e0 >> 3) + 1] = t1;
        if ((e0 & (uint32_t)(kBins * kKsWorld - 1)) == 0) ss->dbase[e0 / (uint32_t)(kBins * kKsWorld)] = excl;
    }
    if (threadIdx.x == 0) {
        ss->dbase[world] = total;
        ss->admitted     = st->admitted;
    }
}

// one block per slot: its records, transformed, to their runs
template <class RecT, class Split, int PER>
__global__ __launch_bounds__(kKsThreads) void ks_split_move_kernel(const RecT* __restrict__ recs, uint32_t region, const uint32_t* __restrict__ slotcnt, Split sp,
                                                                    const KsSplitState* __restrict__ ss, RecT* __restrict__ out) {
    __shared__ KsTileLds<RecT, PER> L;
    constexpr uint32_t              kTile = kKsThreads * PER;
    const uint32_t                  slot = blockIdx.x, n = min(slotcnt[slot * Split::kCurPad], region), nb = 1u << sp.w;
    if (threadIdx.x < kKsWorld) {
        L.cur[threadIdx.x] = ss->soff[slot * kKsWorld + threadIdx.x];
        L.lim[threadIdx.x] = 0xFFFFFFFFu;
    }
    const size_t base = (size_t)slot * region;
    RecT         r[PER];
    auto         load_tile = [&](uint32_t j0) {
#pragma unroll
        for (int k = 0; k < PER; ++k) {
            const uint32_t j = j0 + k * kKsThreads + threadIdx.x;
            if (j < n) r[k] = recs[base + j];
        }
    };
    load_tile(0);
    __syncthreads();
    for (uint32_t j0 = 0; j0 < n; j0 += kTile) {
        bool     valid[PER];
        uint32_t c[PER];
        RecT     v[PER];
#pragma unroll
        for (int k = 0; k < PER; ++k) {
            valid[k] = j0 + k * kKsThreads + threadIdx.x < n;
            c[k]     = 0;
            if (valid[k]) {
                c[k] = sp.cbin(r[k]);
                v[k] = sp.out(r[k]);
            }
        }
        load_tile(j0 + kTile);  // the next tile is in flight while this one is partitioned in LDS
        ks_partition_tile<RecT, PER>(L, valid, c, v, nb, out);
    }
}

// The split without its histogram, for the key slices of one device (colibri_hip.hip: bigram2_order_split / binned_order_split): nothing has to leave dense, so
// run (slot, c) simply gets room for `cap` records at (slot * 8 + c) * cap and the block that owns the slot — the only writer of its eight runs — moves the records
// in one sweep and leaves the counts it reached in the split's tables (hcnt / soff, as the histogram and the scan would have). A run that outgrows its room
// (keys far from uniform) drops the excess and raises the flag: the run repeats with the exact split.
template <class RecT, class Split, int PER>
__global__ __launch_bounds__(kKsThreads) void ks_split_direct_kernel(const RecT* __restrict__ recs, uint32_t region, const uint32_t* __restrict__ slotcnt, Split sp, uint32_t cap,
                                                                      KsSplitState* __restrict__ ss, RecT* __restrict__ out) {
    __shared__ KsTileLds<RecT, PER> L;
    constexpr uint32_t              kTile = kKsThreads * PER;
    const uint32_t                  slot = blockIdx.x, have = slotcnt[slot * Split::kCurPad], n = min(have, region), nb = 1u << sp.w;
    if (threadIdx.x < kKsWorld) {
        L.cur[threadIdx.x] = (slot * kKsWorld + threadIdx.x) * cap;
        L.lim[threadIdx.x] = (slot * kKsWorld + threadIdx.x + 1) * cap;
    }
    const size_t base = (size_t)slot * region;
    RecT         r[PER];
    auto         load_tile = [&](uint32_t j0) {
#pragma unroll
        for (int k = 0; k < PER; ++k) {
            const uint32_t j = j0 + k * kKsThreads + threadIdx.x;
            if (j < n) r[k] = recs[base + j];
        }
    };
    load_tile(0);
    __syncthreads();
    for (uint32_t j0 = 0; j0 < n; j0 += kTile) {
        bool     valid[PER];
        uint32_t c[PER];
        RecT     v[PER];
#pragma unroll
        for (int k = 0; k < PER; ++k) {
            valid[k] = j0 + k * kKsThreads + threadIdx.x < n;
            c[k]     = 0;
            if (valid[k]) {
                c[k] = sp.cbin(r[k]);
                v[k] = sp.out(r[k]);
            }
        }
        load_tile(j0 + kTile);
        ks_partition_tile<RecT, PER>(L, valid, c, v, nb, out);
    }
    __syncthreads();
    if (threadIdx.x < kKsWorld) {
        const uint32_t first = (slot * kKsWorld + threadIdx.x) * cap, got = L.cur[threadIdx.x] - first;
        ss->hcnt[slot * kKsWorld + threadIdx.x] = min(got, cap);
        ss->soff[slot * kKsWorld + threadIdx.x] = first;
        if (got > cap) atomicMax(&ss->overflow, 2u);  // 2: a run outgrew its room (the exact split will do); 1: a slot of the emit kernel overflowed (nothing here will)
        if (threadIdx.x == 0 && have > region) atomicMax(&ss->overflow, 1u);
    }
}

// ---- owner side: the received (source, A' bin) chunks as the slots of the single-device pipelines ---------------------------------------------------
struct KsBases {
    uint32_t rbase[kKsWorld];  // first record of each source's segment of the receive buffer
};
// order 2: Bi2State of the owner's pass (zeroed before): records per slot = source * 256 + A', their places, key / position widths
__global__ __launch_bounds__(kKsThreads) void ks_owner_init2_kernel(Bi2State* __restrict__ obs, uint32_t* __restrict__ slotbase, const uint32_t* __restrict__ tabs, KsBases kb,
                                                                     uint32_t world, uint32_t kbits, uint32_t posbits) {
    __shared__ uint32_t cntL[kKsSlots], offL[kKsSlots], wsumL[4];
    for (uint32_t s = threadIdx.x; s < (uint32_t)kKsSlots; s += kKsThreads) cntL[s] = (s >> 8) < world ? tabs[s] : 0u;
    __syncthreads();
    for (int g = 0; g < kKsWorld; ++g) bi2_scan256(cntL + g * kBins, offL + g * kBins, wsumL);
    for (uint32_t s = threadIdx.x; s < (uint32_t)kKsSlots; s += kKsThreads) {
        obs->curA[bi2_cur(s)] = cntL[s];
        slotbase[s]  = kb.rbase[s >> 8] + offL[s];
    }
    if (threadIdx.x == 0) {
        obs->kbits   = kbits;
        obs->posbits = posbits;
    }
}
// orders >= 3: BinState of the owner's pass (zeroed before), as bin_offsets_kernel leaves it after an emit
__global__ __launch_bounds__(kBlock) void ks_owner_init_kernel(BinState* __restrict__ bs, const uint32_t* __restrict__ tabs, KsBases kb, uint32_t world) {
    uint32_t hsum = 0, tbase = 0;
    for (int g = 0; g < kSub; ++g) {
        const uint32_t s = g * kBins + threadIdx.x;
        const uint32_t h = (uint32_t)g < world ? tabs[s] : 0u;
        const uint32_t t = (h + kScatTile - 1) / kScatTile;
        uint32_t       tt, ht;
        const uint32_t tp = block_exclusive_scan(t, &tt);
        const uint32_t ho = block_exclusive_scan(h, &ht);
        bs->histA[s]      = h;
        bs->offA[s]       = kb.rbase[g] + ho;
        bs->tprefA[s]     = tbase + tp;
        tbase += tt;
        hsum += h;
    }
    uint32_t tot;
    block_exclusive_scan(hsum, &tot);
    bs->histAt[threadIdx.x] = hsum;
    if (threadIdx.x == 0) {
        bs->nrec            = tot;
        bs->offA[kASlots]   = tot;
        bs->tprefA[kASlots] = tbase;
        uint32_t nb = 1;
        while (nb < (uint32_t)kBins && (uint64_t)nb * kBins * 1024u < tot) nb <<= 1;
        uint32_t sh = 0;
        while ((uint32_t)kBins >> sh > nb) ++sh;
        bs->bshift = sh;
    }
}

// One device, a corpus beyond one pass (an order with more than ~110 M records: 10^9 tokens): the same split cuts the order's records ONCE into 2^s key slices — the
// "owners" are the slices, counted one after the other on the same device, and the sub-regions of the one source play the part of the source ranks. The slice's
// slots come straight from the split's own tables: slot (sub, A') of slice v is run (sub, A = v : A' >> s, c = A' & (2^s - 1)).
__global__ __launch_bounds__(kKsThreads) void ks_local_init2_kernel(Bi2State* __restrict__ obs, uint32_t* __restrict__ slotbase, const KsSplitState* __restrict__ ss, uint32_t v,
                                                                     uint32_t s, uint32_t kbits, uint32_t posbits, const uint32_t* __restrict__ nextchunk_keep, uint32_t room,
                                                                     DevState* __restrict__ st) {
    // `room`: records the slice's level-B output has space for. A slice that holds more (keys far from uniform) is not counted: the run repeats on the fallback path
    __shared__ uint32_t sumL;
    if (threadIdx.x == 0) sumL = 0;
    __syncthreads();
    uint32_t mine = 0;
    for (uint32_t slot = threadIdx.x; slot < (uint32_t)kKsSlots; slot += kKsThreads) {
        const uint32_t sub = slot >> 8, ap = slot & 255u, A = (v << (8 - s)) | (ap >> s), c = ap & ((1u << s) - 1u);
        mine += ss->hcnt[(sub * kBins + A) * kKsWorld + c];
    }
    atomicAdd(&sumL, mine);
    __syncthreads();
    const bool fits = sumL <= room && !ss->overflow;
    if (!fits && threadIdx.x == 0) st->radix_overflow = ss->overflow == 2 ? 8 : 4;  // 8: again with the exact split; 4: on the first-generation kernels
    for (uint32_t slot = threadIdx.x; slot < (uint32_t)kKsSlots; slot += kKsThreads) {
        const uint32_t sub = slot >> 8, ap = slot & 255u, A = (v << (8 - s)) | (ap >> s), c = ap & ((1u << s) - 1u);
        const uint32_t idx = (sub * kBins + A) * kKsWorld + c;
        obs->curA[bi2_cur(slot)]    = fits ? ss->hcnt[idx] : 0u;
        slotbase[slot]     = ss->soff[idx];
    }
    if (threadIdx.x == 0) {
        obs->kbits     = kbits;
        obs->posbits   = posbits;
        obs->nextchunk = *nextchunk_keep;  // the position-list pool goes on where the slice before stopped
    }
}
// a slot of the emit kernel outgrew its region: the run repeats on the fallback path (sticky flags of the run's state; orders >= 3: the emit kernel's own check —
// bin_offsets_kernel in the one-pass form — on the cursors it left)
__global__ void ks_split_flag_kernel(const KsSplitState* __restrict__ ss, const BinState* __restrict__ bs, DevState* __restrict__ st) {
    if (ss->overflow) st->radix_overflow = bs != nullptr ? 1 : 4;
}
__global__ void ks_keep_chunk_kernel(const Bi2State* __restrict__ obs, uint32_t* __restrict__ nextchunk_keep) { *nextchunk_keep = obs->nextchunk; }
__global__ __launch_bounds__(kBlock) void ks_local_init_kernel(BinState* __restrict__ bs, const KsSplitState* __restrict__ ss, uint32_t v, uint32_t s, uint32_t room,
                                                                DevState* __restrict__ st) {
    uint32_t hsum = 0, tbase = 0, mine = 0;
    for (int g = 0; g < kSub; ++g) {
        const uint32_t ap = threadIdx.x, A = (v << (8 - s)) | (ap >> s), c = ap & ((1u << s) - 1u);
        mine += ss->hcnt[((uint32_t)g * kBins + A) * kKsWorld + c];
    }
    uint32_t all;
    block_exclusive_scan(mine, &all);
    const bool fits = all <= room && !ss->overflow;  // (`room`: what the slice's level-B output has space for; a split that dropped records is not counted either)
    if (!fits && threadIdx.x == 0) st->radix_overflow = ss->overflow == 2 ? 8 : 1;  // 8: again with the exact split; 1: on the global table
    for (int g = 0; g < kSub; ++g) {
        const uint32_t slot = g * kBins + threadIdx.x, ap = threadIdx.x, A = (v << (8 - s)) | (ap >> s), c = ap & ((1u << s) - 1u);
        const uint32_t idx = ((uint32_t)g * kBins + A) * kKsWorld + c;
        const uint32_t h = fits ? ss->hcnt[idx] : 0u;
        const uint32_t t = (h + kScatTile - 1) / kScatTile;
        uint32_t       tt;
        const uint32_t tp = block_exclusive_scan(t, &tt);
        bs->histA[slot]   = h;
        bs->offA[slot]    = ss->soff[idx];
        bs->tprefA[slot]  = tbase + tp;
        tbase += tt;
        hsum += h;
    }
    uint32_t tot;
    block_exclusive_scan(hsum, &tot);
    bs->histAt[threadIdx.x] = hsum;
    if (threadIdx.x == 0) {
        bs->nrec            = tot;
        bs->offA[kASlots]   = 0;
        bs->tprefA[kASlots] = tbase;
        uint32_t nb = 1;
        while (nb < (uint32_t)kBins && (uint64_t)nb * kBins * 1024u < tot) nb <<= 1;
        uint32_t sh = 0;
        while ((uint32_t)kBins >> sh > nb) ++sh;
        bs->bshift = sh;
    }
}

// order 2, the dense head (both classes < 64: never records): every rank's local histogram -> [0, 4096) counts (all-reduce SUM by the caller),
// [4096, 8192) this rank where it saw the bigram, else 0x7FFFFFFF (all-reduce MIN: the rank that will export it)
__global__ __launch_bounds__(kBlock) void ks_head_pack_kernel(const Bi2State* __restrict__ sbs, uint32_t rank, uint32_t* __restrict__ headg) {
    const uint32_t k = blockIdx.x * kBlock + threadIdx.x;
    if (k < (uint32_t)kBi2HeadN) {
        const uint32_t c      = sbs->headcnt[k];
        headg[k]              = c;
        headg[kBi2HeadN + k] = c ? rank : 0x7FFFFFFFu;
    }
}
// bi2_finish_kernel for an owner: the bins' found / kept are this owner's; the head is global — its survivors are the same bits on every rank (the list kernel of
// every rank needs them), each is exported by the lowest rank that saw it (with that rank's own lowest position), and rank 0 accounts for the distinct head keys
__global__ __launch_bounds__(kBlock) void ks_finish2_kernel(DevState* __restrict__ ost, Bi2State* __restrict__ obs, const uint32_t* __restrict__ headg, uint32_t rank, uint32_t threshold,
                                                             uint32_t res_cap, uint32_t* __restrict__ headsurv_keep) {
    uint32_t htot, ftot, hftot;
    block_exclusive_scan(obs->found_part[threadIdx.x], &ftot);
    uint32_t hk = 0, hf = 0, bits = 0, mine = 0, hw = 0;
#pragma unroll
    for (int q = 0; q < 16; ++q) {
        const uint32_t c = headg[threadIdx.x * 16 + q];
        hf += c != 0;
        if (c >= threshold) {
            bits |= 1u << q;
            hw += c;  // (headg holds the all-reduced counts: every rank computes the same sum)
            if (headg[kBi2HeadN + threadIdx.x * 16 + q] == rank) {
                mine |= 1u << q;
                ++hk;
            }
        }
    }
    reinterpret_cast<uint16_t*>(obs->headsurv)[threadIdx.x] = (uint16_t)mine;
    reinterpret_cast<uint16_t*>(headsurv_keep)[threadIdx.x] = (uint16_t)bits;
    const uint32_t ho          = block_exclusive_scan(hk, &htot);
    obs->headbase[threadIdx.x] = ho;
    block_exclusive_scan(hf, &hftot);
    uint32_t hwtot;
    block_exclusive_scan(hw, &hwtot);
    if (threadIdx.x == 0) {
        const uint32_t tot = obs->kept_bins;
        obs->kept_head     = htot;
        obs->head_windows  = hwtot;
        obs->res_base      = ost->res_total + ost->kept;
        ost->found += ftot + (rank == 0 ? hftot : 0u);
        ost->kept += tot + htot;
        if ((uint64_t)obs->res_base + tot + htot > res_cap) ost->overflow = 1;
        if (obs->overflow) ost->radix_overflow = 4 + obs->overflow;  // 5: a slot, 6: a final bin's LDS table, 7: a wave's position list
    }
}
// bi2_compact_kernel for an owner: representatives are tagged positions already; the head survivors this rank exports get its own lowest position
__global__ __launch_bounds__(kBlock) void ks_compact2_kernel(const uint32_t* __restrict__ sp_rep, const uint32_t* __restrict__ sp_cnt, const Bi2State* __restrict__ obs,
                                                              const Bi2State* __restrict__ sbs, const uint32_t* __restrict__ headg, uint32_t tag, uint32_t* __restrict__ res_rep,
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
                    res_rep[r] = sp_rep[src + j];
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
                    res_rep[r] = tag | ~sbs->headposinv[threadIdx.x * 16 + q];
                    res_cnt[r] = headg[threadIdx.x * 16 + q];
                }
                ++r;
            }
    }
}

// ---- route: what goes back, partitioned by the rank it goes to --------------------------------------------------------------------------------------
// A Spec describes `nlists` lists of entries: count(l), and get(l, j, dst, out) -> false when entry j is not sent.
// order 2: the count kernel's per-wave lists of tagged positions (source << pb | position) -> the position, to its source
struct KsRouteWaveLists {
    const uint32_t* wlist;
    const uint32_t* wcnt;
    uint32_t        wcap, pb;
    typedef uint32_t Out;
    __device__ __forceinline__ uint32_t count(uint32_t l) const { return min(wcnt[l], wcap); }
    __device__ __forceinline__ bool get(uint32_t l, uint32_t j, uint32_t& dst, Out& out) const {
        const uint32_t e = wlist[(size_t)l * wcap + j];
        dst              = e >> pb;
        out              = e & ((1u << pb) - 1u);
        return true;
    }
};
// orders >= 3: the records of surviving keys (reply[j] = tagged item | the key's global survivor id << 32, all-ones otherwise; j = the record's place after
// level B) -> (item index | id << 32), to the record's source
struct KsRouteReplies {
    const unsigned long long* reply;
    const uint32_t*           n_dev;
    uint32_t                  len, tagshift;  // tagshift = 32 - w (32: one rank, no tag)
    typedef unsigned long long Out;
    __device__ __forceinline__ uint32_t count(uint32_t l) const {
        const uint32_t n = *n_dev, b = l * len;
        return b < n ? min(len, n - b) : 0u;
    }
    __device__ __forceinline__ bool get(uint32_t l, uint32_t j, uint32_t& dst, Out& out) const {
        const unsigned long long e = reply[l * len + j];
        if (e == ~0ull) return false;
        const uint32_t p = (uint32_t)e;
        dst              = tagshift < 32 ? p >> tagshift : 0u;
        out              = tagshift < 32 ? (e & ~((unsigned long long)(~0u << tagshift) & 0xFFFFFFFFull)) : e;
        return true;
    }
};
// every order: the owner's survivors (tagged representative, global count) -> (representative | count << 32), to the representative's rank
struct KsRouteExports {
    const uint32_t* rep;
    const uint32_t* cnt;
    const uint32_t* n_dev;
    uint32_t        len, tagshift;
    typedef unsigned long long Out;
    __device__ __forceinline__ uint32_t count(uint32_t l) const {
        const uint32_t n = *n_dev, b = l * len;
        return b < n ? min(len, n - b) : 0u;
    }
    __device__ __forceinline__ bool get(uint32_t l, uint32_t j, uint32_t& dst, Out& out) const {
        const uint32_t at = l * len + j, p = rep[at];
        dst               = tagshift < 32 ? p >> tagshift : 0u;
        out               = (unsigned long long)(tagshift < 32 ? p & ((1u << tagshift) - 1u) : p) | ((unsigned long long)cnt[at] << 32);
        return true;
    }
};

template <class Spec>
__global__ __launch_bounds__(kKsThreads) void ks_route_hist_kernel(Spec sp, uint32_t nlists, uint32_t* __restrict__ lcnt) {
    __shared__ uint32_t histL[kKsWorld];
    for (uint32_t l = blockIdx.x; l < nlists; l += gridDim.x) {
        if (threadIdx.x < kKsWorld) histL[threadIdx.x] = 0;
        __syncthreads();
        const uint32_t n = sp.count(l);
        KsPacked       acc;
        for (uint32_t j = threadIdx.x; j < n; j += kKsThreads) {
            uint32_t           d;
            typename Spec::Out o;
            if (sp.get(l, j, d, o)) acc.add(d);
        }
        acc.flush(histL);
        __syncthreads();
        if (threadIdx.x < kKsWorld) lcnt[l * kKsWorld + threadIdx.x] = histL[threadIdx.x];
        __syncthreads();
    }
}
// one block: lcnt -> loff in (destination, list) order; the destinations' segment bases
__global__ __launch_bounds__(kKsThreads) void ks_route_scan_kernel(const uint32_t* __restrict__ lcnt, uint32_t* __restrict__ loff, uint32_t nlists, uint32_t world,
                                                                    KsRouteState* __restrict__ rs) {
    __shared__ uint32_t wsumL[kKsThreads / kWave], carryL;
    if (threadIdx.x == 0) carryL = 0;
    __syncthreads();
    const uint32_t nent = world * nlists;
    for (uint32_t base = 0; base < nent; base += kKsThreads * 8) {
        uint32_t v[8], s = 0;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const uint32_t e = base + threadIdx.x * 8 + k;
            v[k]             = e < nent ? lcnt[(e % nlists) * kKsWorld + e / nlists] : 0u;
            s += v[k];
        }
        uint32_t       total;
        const uint32_t carry = carryL;
        uint32_t       run   = carry + bi2_block_scan<kKsThreads>(s, &total, wsumL);
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const uint32_t e = base + threadIdx.x * 8 + k;
            if (e < nent) {
                loff[(e % nlists) * kKsWorld + e / nlists] = run;
                if (e % nlists == 0) rs->dbase[e / nlists] = run;
            }
            run += v[k];
        }
        __syncthreads();
        if (threadIdx.x == 0) carryL = carry + total;
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        rs->dbase[world] = carryL;
        if (nlists == 0)
            for (uint32_t d = 0; d < world; ++d) rs->dbase[d] = 0;
    }
}
template <class Spec, int PER>
__global__ __launch_bounds__(kKsThreads) void ks_route_move_kernel(Spec sp, uint32_t nlists, const uint32_t* __restrict__ loff, uint32_t nb, typename Spec::Out* __restrict__ out) {
    typedef typename Spec::Out      Out;
    __shared__ KsTileLds<Out, PER> L;
    constexpr uint32_t              kTile = kKsThreads * PER;
    for (uint32_t l = blockIdx.x; l < nlists; l += gridDim.x) {
        const uint32_t n = sp.count(l);
        if (threadIdx.x < kKsWorld) {
            L.cur[threadIdx.x] = loff[l * kKsWorld + threadIdx.x];
            L.lim[threadIdx.x] = 0xFFFFFFFFu;
        }
        __syncthreads();
        for (uint32_t j0 = 0; j0 < n; j0 += kTile) {
            bool     valid[PER];
            uint32_t c[PER];
            Out      v[PER];
#pragma unroll
            for (int k = 0; k < PER; ++k) {
                const uint32_t j = j0 + k * kKsThreads + threadIdx.x;
                c[k]             = 0;
                v[k]             = 0;
                valid[k]         = j < n && sp.get(l, j, c[k], v[k]);
            }
            ks_partition_tile<Out, PER>(L, valid, c, v, nb, out);
        }
        __syncthreads();
    }
}

// ---- source side: what came back ----------------------------------------------------------------------------------------------------------------------
// orders >= 3: ids_at[item] = the global survivor id of the item's key (bin_emit_kernel reset ids_at at every record's item; bin_resolve_kernel reads it)
__global__ __launch_bounds__(kBlock) void ks_apply_ids_kernel(const unsigned long long* __restrict__ replies, uint32_t n, uint32_t* __restrict__ ids_at) {
    for (uint32_t j = blockIdx.x * kBlock + threadIdx.x; j < n; j += gridDim.x * kBlock) {
        const unsigned long long e = replies[j];
        ids_at[(uint32_t)e]        = (uint32_t)(e >> 32);
    }
}
// the patterns this rank exports: (representative | count << 32) -> the result arrays; `list`: representatives are item indices of that list (orders >= 3)
__global__ __launch_bounds__(kBlock) void ks_append_exports_kernel(const unsigned long long* __restrict__ ex, uint32_t n, const uint32_t* __restrict__ list, uint32_t* __restrict__ rep,
                                                                    uint32_t* __restrict__ cnt) {
    for (uint32_t j = blockIdx.x * kBlock + threadIdx.x; j < n; j += gridDim.x * kBlock) {
        const unsigned long long e = ex[j];
        const uint32_t           p = (uint32_t)e;
        rep[j]                     = list != nullptr ? list[p] : p;
        cnt[j]                     = (uint32_t)(e >> 32);
    }
}
// between two orders of an owner: this order's figures into the per-order table, the counters back to zero; survivor ids of the next order start above this one's
// records (bin_advance_prepare_kernel moved id_base on); they must stay below the rank tag
struct KsStats {
    uint32_t found[COLIBRI_MAX_ORDER], kept[COLIBRI_MAX_ORDER], admitted[COLIBRI_MAX_ORDER];
};
__global__ void ks_idcheck_kernel(DevState* __restrict__ ost, uint32_t idlimit) {
    if (ost->id_base >= idlimit) ost->radix_overflow = 3;
}
__global__ void ks_order_end_kernel(DevState* __restrict__ ost, DevState* __restrict__ st, KsStats* __restrict__ ks, int n, uint32_t idlimit) {
    if (n < COLIBRI_MAX_ORDER) {
        ks->found[n]    = ost->found;
        ks->kept[n]     = ost->kept;
        ks->admitted[n] = st->admitted;
    }
    if (ost->id_base >= idlimit) ost->radix_overflow = 3;
    ost->found = ost->kept = ost->res_total = 0;
    st->admitted = st->valid = 0;
}

}  // namespace colibri
