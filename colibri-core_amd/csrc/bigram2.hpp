// bigram2.hpp — order 2 of the plain run, second generation of the radix path (gfx950, wave64).
//
// Order 2 is the heaviest pass of PatternModel::train on a class-encoded corpus (reference include/patternmodel.h:1078-1178: window loop,
// look-back :1139-1152, add :2059-2073, prune :2107-2128): nearly every window is admissible and half of them are singletons. The first
// generation (binned.hpp) moved every window five times across HBM as a 16-byte record, elected representatives per tile and scattered
// survivor bytes back to representative positions (one 32-byte sector per record). This file does the same pass with
//   * 8-byte records: the key (two class ids, 42 bits) travels as its BIJECTIVE 42-bit mix; 8 bits of it are implied by the level-A bin the
//     record lies in, the other 34 share the word with the 27-bit corpus position. Distinct bigrams can never merge: inside a final bin the
//     31 low bits of the mix identify the key exactly (the bin fixes the rest);
//   * no election and no rep_of array: the Zipf head (both classes < 64; class ids are frequency-ranked, reference src/classencoder.cpp:220-224)
//     is counted in a dense 64 x 64 LDS histogram per block and never becomes a record, so no bin and no region is skewed by a hot key;
//   * level B without histogram pass and without atomics: every (sub-region, A bin) slot is partitioned by ONE block (histogram sweep, then a
//     move sweep that re-reads the slot from L2 / Infinity Cache), so a final bin is the concatenation of nsub short runs;
//   * no scatter back to positions: the count kernel appends the positions of the windows whose bigram survived to per-position-bucket lists
//     through LDS write-combining queues, bi2_bitmap_kernel turns each bucket's list into a bitmap (built in LDS, one bit per position), and
//     bi2_list3_kernel streams over the corpus once to emit the active list of order 3 (positions i with surviving bigrams at i and i + 1).
// The kernels that walk bins or tiles are software-pipelined (the loads of the next bin / tile are in flight while the current one is in LDS):
// a bin is latency-bound, not bandwidth-bound.
// Results (representative position + count per surviving bigram) go to the same result arrays as every other order.
// Anything this path cannot hold (a slot, a final bin's LDS table) raises Bi2State::overflow; the host then re-runs on the first-generation kernels.
#pragma once
#ifndef COLIBRI_BI2_HUGE
#define COLIBRI_BI2_HUGE 16384
#endif
#include "binned.hpp"

namespace colibri {

// ---- the bijective K-bit mix ----------------------------------------------------------------------------------------------------
// x < 2^K, K >= 17: xor-shifts and odd multiplications mod 2^K are bijections of the K-bit integers.
// Bits of the mix, from the top: [s slice bits][8 A-bin bits][9 B-bin bits][the rest]. A record keeps the bits below the A bin next to the corpus
// position: (K - s - 8) + posbits <= 64. The slice bits exist for corpora beyond ~128 M tokens per device: the order is then counted in 2^s passes,
// each over the keys of one slice (every pass scans the corpus and keeps its share of the windows), so that a final bin stays within one wave's LDS
// table and a 30-bit position fits the record.
// Round 6: a four-round Feistel network over the key's two halves instead of two 64-bit multiplications (xor-shift, multiply mod 2^K, twice). SQ counters
// (profiles/r06c_pmc_sq.txt) show bi2_emit_kernel bound by VALU issue: 214 M vector instructions per launch = 137 per window, ~90 % of the SIMDs' cycles once the
// quarter-rate 32-bit multiplies are priced — and seven of those per window were this function's (v_mul_lo_u32 / v_mad_u64_u32 of the two 64 x 64 -> 64 products).
// A round is (L, R) -> (L ^ F(R), R): invertible for ANY F, so the network is a bijection of the K-bit integers by construction; F(R) = the top bits of the low 32
// bits of R x c with 24-bit operands — one full-rate v_mul_u32_u24 — cut to the other half's width. Halves: lo = the low K / 2 bits, hi = the rest (<= 24 bits each for
// K <= 48: every key of this path; a wider half would only lose its top bits inside F, the map stays a bijection). On 1.27 x 10^7 distinct bigrams of a Zipf corpus the
// keys per bin are as Poisson as the old mix's at 8 / 15 / 17 bin bits (std 223.0 / 19.8 / 9.8 against 222.8 / 19.7 / 9.9 expected), likewise (number, class) keys.
constexpr uint32_t kBi2MixC[4] = {0x9E3779u, 0x85EBCBu, 0xC2B2AFu, 0x27D4EBu};  // odd, 24 bits
__host__ __device__ __forceinline__ uint32_t bi2_mix_f(uint32_t r, uint32_t c, uint32_t outbits) {
#if defined(__HIP_DEVICE_COMPILE__)
    const uint32_t p = __umul24(r, c);
#else
    const uint32_t p = (uint32_t)((uint64_t)(r & 0xFFFFFFu) * c);
#endif
    return p >> (32u - outbits);  // (outbits <= 24)
}
__host__ __device__ __forceinline__ uint64_t bi2_mix(uint64_t x, uint32_t K) {
#ifdef COLIBRI_OLD_MIX  // (A/B measurements only: rounds 2-5's mix)
    const uint64_t M = (1ull << K) - 1;
    const uint32_t g = K >> 1;
    x ^= x >> g;
    x = (x * 0xff51afd7ed558ccdULL) & M;
    x ^= x >> (g - 1);
    x = (x * 0xc4ceb9fe1a85ec53ULL) & M;
    x ^= x >> g;
    return x;
#endif
    const uint32_t h = K >> 1, hb = K - h;  // lo: h bits, hi: hb bits (h <= hb)
    const uint32_t fl = h > 24u ? 24u : h, fh = hb > 24u ? 24u : hb;
    uint32_t       lo = (uint32_t)x & ((1u << h) - 1u), hi = (uint32_t)(x >> h);  // (K <= 48 + slack: both halves fit a word — K <= 56)
    hi ^= bi2_mix_f(lo, kBi2MixC[0], fh);
    lo ^= bi2_mix_f(hi, kBi2MixC[1], fl);
    hi ^= bi2_mix_f(lo, kBi2MixC[2], fh);
    lo ^= bi2_mix_f(hi, kBi2MixC[3], fl);
    return ((uint64_t)hi << h) | lo;
}

#ifndef COLIBRI_BI2_CNT16
#define COLIBRI_BI2_CNT16 1
#endif
#ifndef COLIBRI_BI2_WEU
#define COLIBRI_BI2_WEU 4
#endif
// waves per SIMD the wave-per-bin count kernel is compiled for: five where its 16-bit counters leave the LDS for them (96 registers), else COLIBRI_BI2_WEU
constexpr int bi2_count_weu(int slots, bool key4, bool based) { return (COLIBRI_BI2_CNT16 && slots == 1024 && !key4 && !based) ? 5 : COLIBRI_BI2_WEU; }
#ifndef COLIBRI_BI2_WROWS
#define COLIBRI_BI2_WROWS 12
#endif
#ifndef COLIBRI_BI2_WSLOTS
#define COLIBRI_BI2_WSLOTS 1024
#endif
#ifndef COLIBRI_BI2_BINTARGET
#define COLIBRI_BI2_BINTARGET 700
#endif
constexpr int      kBi2WRows = COLIBRI_BI2_WROWS;   // records per lane in registers (bins of up to 64 x this many records are read once)
constexpr int      kBi2WSlots = COLIBRI_BI2_WSLOTS; // the wave kernel's LDS table
constexpr uint32_t kBi2BinTarget = COLIBRI_BI2_BINTARGET;  // records per final bin aimed at
constexpr int      kBi2Threads  = 1024;                   // emit / level-B block (wide blocks, few items per lane: short LDS chains, 32 waves per CU)
constexpr int      kBi2Per      = 4;                      // items per lane
constexpr int      kBi2Tile     = kBi2Threads * kBi2Per;  // 4096 windows (records) per tile
constexpr int      kBi2Head     = 64;                     // classes < 64 on both sides: dense histogram
constexpr int      kBi2HeadN    = kBi2Head * kBi2Head;
constexpr int      kBi2MaxSub   = 32;                     // sub-regions per A bin (a multiple of 8: sub-region = block index mod nsub, XCD = block index mod 8)
constexpr int      kBi2MaxSlots = kBins * kBi2MaxSub;
constexpr uint32_t kBi2MaxPosBits = 30;                   // corpus positions per device on this path: < 2^30
constexpr int      kBi2BBins    = 512;                    // level-B bins per A bin: ~660 records per final bin at 100 M tokens (one wave counts a bin)
constexpr int      kBi2Final    = kBins * kBi2BBins;      // 131 072 final bins
constexpr int      kBi2Buckets  = 1024;                   // position buckets at most (one LDS bitmap each in bi2_bitmap_kernel)
constexpr int      kBi2Shards   = 8;                      // cursor shards per bucket (a single cursor would serialise ~12 ns per reservation)
constexpr int      kBi2Slots    = 1024;                   // LDS table of one final bin: 256 buckets of 4 slots
constexpr uint32_t kBi2MaxLoad  = 900;
constexpr uint32_t kBi2Empty    = 0xFFFFFFFFu;            // keys are 31 bits
constexpr int      kBi2HeadSplit = 32;                    // row groups of the head reduction
constexpr int      kBi2BmThreads = 1024;                  // bitmap kernels: one block per position bucket
constexpr uint32_t kBi2BigBin    = 1536;                  // records from which a final bin counts as big
constexpr uint32_t kBi2HugeBin   = COLIBRI_BI2_HUGE;      // ... and as huge: counted by a workgroup (bi2_count_big_kernel), not by a wave
constexpr int      kBi2HugeCap   = 4096;
constexpr uint32_t kBi2HeadCode  = 0x80000000u;           // a list entry's code with this bit: a head window, the low 12 bits name the class pair (chain.hpp)
constexpr int      kBi2BigCap    = 32768;                 // (a 10^9-token corpus counted in 8 key slices: ~7000 per slice)

#ifndef COLIBRI_BI2_CURPAD
#define COLIBRI_BI2_CURPAD 1
#endif
// The emit kernels reserve a run with one returning atomic per (tile or queue, A bin), on the cursor of slot s at curA[s * kBi2CurPad]. Round 6 measured the cursors
// 64 and 128 bytes apart (order 1's 256 cursors on lines of their own halved uni_onepass_kernel): here it costs — 3.71 -> 3.75 / 3.81 ms per step, the emit kernels
// 5-13 % slower (four sub-regions already spread a bin's reservations; the clears and the last block's scan of the cursors grow). Kept adjacent.
constexpr uint32_t kBi2CurPad = COLIBRI_BI2_CURPAD;
__host__ __device__ __forceinline__ constexpr uint32_t bi2_cur(uint32_t slot) { return slot * kBi2CurPad; }
#ifndef COLIBRI_BI2_PCPAD
#define COLIBRI_BI2_PCPAD 1
#endif
constexpr uint32_t kBi2PcPad = COLIBRI_BI2_PCPAD;  // the position lists' cursors, this many words apart: cursor of list l at pcur[l * kBi2PcPad]. Round 6 measured 4 / 16
// words: bi2_pospart_kernel's three launches 0.57 -> 0.60 / 0.93 ms — a tile's 1024 reservations are adjacent lanes on adjacent words, which the memory side takes a
// line at a time. Kept adjacent (as the emit kernels' slot cursors, above).
__host__ __device__ __forceinline__ constexpr uint32_t bi2_pc(uint32_t l) { return l * kBi2PcPad; }
struct __attribute__((aligned(16))) Bi2State {
    uint32_t curA[kBi2MaxSlots * kBi2CurPad];  // emit cursors = records per slot (beyond `region`: overflow), kBi2CurPad words apart
    uint32_t cntA[kBins];         // records per A bin
    uint32_t offAt[kBins + 1];    // exclusive scan of cntA
    uint32_t found_part[kBins], kept_part[kBins];
    uint32_t binoff[kBi2Final + 1];  // first sparse index of final bin f = a * 512 + b (a bin owns as many sparse entries as it has records)
    uint32_t binkept[kBi2Final];     // survivors of the bin; after bi2_kept_scan_kernel: their dense offsets
    uint32_t headcnt[kBi2HeadN], headposinv[kBi2HeadN];  // the reduced head histogram: count and ~(lowest position) per (c0, c1)
    uint32_t headsurv[kBi2HeadN / 32];                   // bit k = head bigram k survived
    uint32_t headbase[kBlock];                           // first result rank of the head survivors of lane t (16 head keys per lane)
    uint32_t pcur[(kBi2Shards + 1) * kBi2Buckets * kBi2PcPad];       // position-list cursors (shard 8: chain.hpp's lists of the head windows, one per bucket)
    uint32_t nrec, bshift, overflow, kept_bins, kept_head, nbig, nhuge;
    uint32_t kbits;     // key bits below the slice bits (K - s): A bin = bits [kbits-1 : kbits-8], B bin = the nine below
    uint32_t posbits;   // position bits of a record
    uint32_t res_base;  // first result index of this pass's survivors (an order counted in slices appends pass after pass)
    uint32_t cskip;      // key-sharded runs (kshard2.hpp): the w mix bits below the B bin's that complete the owner's final bin (the source cuts every (A, B) bin by them)
    uint32_t bshift_fix; // key-sharded runs: bshift + 1 as every rank agreed on it (0: bi2_offsets_kernel derives it from this pass's record count)
    uint32_t ran;        // set by bi2_finish_kernel: this order was counted (the run had not ended before it) — what bi2_compact_kernel asks when it runs beside the path
    uint32_t hugebin;   // records from which a final bin goes to bi2_count_big_kernel; 0: kBi2HugeBin (written with kbits by the emit kernel of the order)
    uint32_t pdshift;                   // chain.hpp, corpora beyond 2.15 x 10^8 positions: pshift + 1 when a record's position lacks the three bits above pshift (they equal
                                        // the sub-region the record lies in); the count kernels (PDROP) put them back. 0: positions are whole
    uint32_t emit_done;                 // blocks of the order's emit kernel that have finished: the last one runs bi2_offsets_tail
    uint32_t head_windows;              // key-sharded runs, order 2 (ks_finish2_kernel): the windows of the surviving head pairs over ALL ranks — what order 3 adds to the owners' lists
    uint32_t nextchunk;                 // owner passes of key-sharded runs: next free chunk of the position-list pool (bi2_count_kernel<.., BASED>)
    uint32_t nextbin[kBi2Shards * 16];  // work queues of the count kernel (one per shard, 64 bytes apart): next group of bins to hand out
    uint32_t big[kBi2BigCap];           // final bins with more than kBi2BigBin records: counted first (one wave each), so that none of them starts late
    uint32_t huge[kBi2HugeCap];         // ... with more than kBi2HugeBin records: one workgroup each
};

// ---- position lists: one per (shard, position bucket), filled by bi2_pospart_kernel (and, shard 8, by bi2_emit_kernel for chain.hpp) ----------
struct Bi2Lists {
    uint32_t pcap;    // entries per (shard, bucket) list, a multiple of 4
    uint32_t pshift;  // bucket = position >> pshift
    uint32_t hbase;   // chain.hpp: the head windows' lists (shard 8) start at this entry, list b at hbase + (b << pshift) with room for every position of the bucket
};
// first entry and capacity of list (shard x, bucket b)
__device__ __forceinline__ void bi2_list_of(const Bi2Lists& pl, uint32_t x, uint32_t b, uint32_t& first, uint32_t& cap) {
    if (x < (uint32_t)kBi2Shards) {
        first = (x * kBi2Buckets + b) * pl.pcap;  // (fits 32 bits: shards * buckets * pcap <= 8 * positions)
        cap   = pl.pcap;
    } else {
        first = pl.hbase + (b << pl.pshift);
        cap   = 1u << pl.pshift;
    }
}
// exclusive scan of 256 LDS values by the first 256 threads of a block of any size; every thread of the block must call it
__device__ __forceinline__ uint32_t bi2_scan256(const uint32_t* inL, uint32_t* outL, uint32_t* wsumL) {
    const uint32_t lane = threadIdx.x & (kWave - 1), wave = threadIdx.x / kWave;
    uint32_t       v = 0, incl = 0;
    if (threadIdx.x < 256) {
        v    = inL[threadIdx.x];
        incl = v;
        for (int off = 1; off < kWave; off <<= 1) {
            const uint32_t t = __shfl_up(incl, off, kWave);
            if ((int)lane >= off) incl += t;
        }
        if (lane == kWave - 1) wsumL[wave] = incl;
    }
    __syncthreads();
    const uint32_t s0 = wsumL[0], s1 = wsumL[1], s2 = wsumL[2], s3 = wsumL[3];
    if (threadIdx.x < 256) outL[threadIdx.x] = (wave > 0 ? s0 : 0u) + (wave > 1 ? s1 : 0u) + (wave > 2 ? s2 : 0u) + incl - v;
    __syncthreads();
    return s0 + s1 + s2 + s3;
}
// exclusive scan of one value per thread over a block of T threads (T / 64 <= 16 waves); wsumL: T / 64 words
template <int T>
__device__ __forceinline__ uint32_t bi2_block_scan(uint32_t v, uint32_t* total, uint32_t* wsumL) {
    const uint32_t lane = threadIdx.x & (kWave - 1), wave = threadIdx.x / kWave;
    uint32_t       incl = v;
    for (int off = 1; off < kWave; off <<= 1) {
        const uint32_t t = __shfl_up(incl, off, kWave);
        if ((int)lane >= off) incl += t;
    }
    if (lane == kWave - 1) wsumL[wave] = incl;
    __syncthreads();
    uint32_t ws = lane < (uint32_t)(T / kWave) ? wsumL[lane] : 0u, wincl = ws;  // every wave scans the (<= 16) wave sums itself
    for (int off = 1; off < T / kWave; off <<= 1) {
        const uint32_t t = __shfl_up(wincl, off, kWave);
        if ((int)lane >= off) wincl += t;
    }
    const uint32_t base = __shfl(wincl - ws, (int)wave, kWave);
    *total              = __shfl(wincl, T / kWave - 1, kWave);
    __syncthreads();
    return base + incl - v;
}

// the B-bin shift for `tot` records of an order (one thread; bi2_offsets_tail and bi2_offsets_kernel)
__device__ __forceinline__ void bi2_set_bshift(Bi2State* __restrict__ bs, uint32_t tot) {
    // aim at <= ~700 records per final bin; at least 8 B bins (the 31-bit in-bin key needs three mix bits fixed by the B bin)
    uint32_t nb = 8;
    while (nb < (uint32_t)kBi2BBins && (uint64_t)nb * kBins * kBi2BinTarget < tot) nb <<= 1;
    uint32_t sh = 0;
    while ((uint32_t)kBi2BBins >> sh > nb) ++sh;
    // the 31-bit in-bin key must hold every mix bit the bin does not fix: kbits - 17 + bshift <= 31
    const uint32_t K = bs->kbits;
    if (sh + K > 48u) sh = K >= 48u ? 0u : 48u - K;
    bs->bshift = bs->bshift_fix ? bs->bshift_fix - 1u : sh;
}
// ---- records per A bin, their scan, the B-bin shift for this record count: the last block of an emit kernel to finish does it ---------------------------------------
// (round 5; bi2_offsets_kernel — one block after the emit kernel — cost a launch per order and per skipgram pass.) Every block of the emit kernel calls this as its
// last statement with three LDS arrays it no longer needs (histL, offL: kBins words; wsumL: 4) and a flag word. The cursors were advanced by device-scope atomics of
// other blocks: they are read with agent-scope atomic loads (this CU's L1 may hold nothing of them, but must not be trusted to).
// NO __threadfence(): a device-scope release on this multi-XCD part writes the XCD's whole L2 back — 512 blocks doing so at the end of a kernel that has just written
// 0.7 GB of records cost 0.13-0.3 ms per launch (measured). None is needed: the cursors are only ever touched by returning device-scope atomics, which have been
// performed when their value is back, i.e. before the block's barrier and its own ticket; the last block reads them with atomic loads, and what it writes is read by
// the next kernel.
__device__ __forceinline__ void bi2_offsets_tail(Bi2State* __restrict__ bs, uint32_t region, uint32_t nsub, uint32_t* inL, uint32_t* outL, uint32_t* wsumL, uint32_t* flagL) {
    __syncthreads();
    if (threadIdx.x == 0) *flagL = (atomicAdd(&bs->emit_done, 1u) + 1u == gridDim.x) ? 1u : 0u;
    __syncthreads();
    if (*flagL == 0u) return;
    //  the last block only: an acquire fence at agent scope before it reads what the other blocks' atomics left — this CU's caches are invalidated once, no
    // cache is written back. The other side needs no release: the cursors are written by device-scope atomics alone, each performed at the memory side before its value
    // came back to the block that then drew its ticket. (A release in every block is what cost 0.13-0.3 ms per launch.)
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    uint32_t s = 0;
    if (threadIdx.x < (uint32_t)kBins) {
        for (uint32_t g = 0; g < nsub; ++g) {
            const uint32_t h = __hip_atomic_load(&bs->curA[bi2_cur(g * kBins + threadIdx.x)], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (h > region) bs->overflow = 1;
            s += min(h, region);
        }
        inL[threadIdx.x] = s;
    }
    __syncthreads();
    const uint32_t tot = bi2_scan256(inL, outL, wsumL);
    if (threadIdx.x < (uint32_t)kBins) {
        bs->cntA[threadIdx.x]  = s;
        bs->offAt[threadIdx.x] = outL[threadIdx.x];
    }
    if (threadIdx.x == 0) {
        bs->offAt[kBins] = tot;
        bs->nrec         = tot;
        bi2_set_bshift(bs, tot);
    }
}

// ---- emit: windows -> head histogram | 8-byte records partitioned by A bin -------------------------------------------------------------
// grid: a multiple of nsub persistent blocks; head_rows: [gridDim.x][2][kBi2HeadN] (counts, lowest positions).
// Wide blocks with few items per lane: every phase of a tile is a short chain of LDS operations, and 32 waves per CU hide each other's latencies.
constexpr int kBi2SurvLds = 2048;  // words of the order-1 survivor bitmap kept in LDS: the first 65 536 classes (> 80 % of a Zipf corpus' tokens)
// (Round 6 measured a two-part LDS form — 32 768 classes bit by bit + one "whole group survived" bit per 32 of the others, so that hardly any lane reads the bitmap in
// memory —: 0.55 -> 0.58 ms; the gathers of the rarer classes are not what the kernel waits for.)
__global__ __launch_bounds__(kBi2Threads, kBi2Threads / 128) void bi2_emit_kernel(const uint32_t* cls, const uint32_t* __restrict__ surv, uint32_t nsurvwords, uint32_t npos,
                                                                                   uint32_t clsbits, uint32_t sbits, uint32_t slice, uint32_t pb,
                                                                                   unsigned long long* __restrict__ recsA, uint32_t region, uint32_t nsub, Bi2State* __restrict__ bs,
                                                                                   DevState* __restrict__ st, uint32_t* __restrict__ head_rows,
                                                                                   uint8_t* __restrict__ sid = nullptr /* optional (first pass of a sliced order): the key slice of the
                                                                                       window at every position, 0xFF where no record will ever start (not admissible, or a head bigram) */,
                                                                                   uint32_t kmin = 0 /* key-sharded runs (kshard.hpp): at least this many key bits (17 + owner bits) */,
                                                                                   uint32_t* __restrict__ hplist = nullptr, uint32_t* __restrict__ hpcode = nullptr /* optional (chain.hpp):
                                                                                       every head window also joins list (shard 8, bucket) as (position, kBi2HeadCode | head pair) */,
                                                                                   Bi2Lists hpl = Bi2Lists{0u, 0u, 0u}) {
    if (st->done) return;
    const uint32_t K  = max(max(2u * clsbits, 17u + sbits), kmin);  // key = (class at i) << clsbits | class at i + 1
    const uint32_t Kp = K - sbits;                      // ... of which the slice fixes the top sbits
    if (threadIdx.x == 0) {
        bs->kbits   = Kp;
        bs->posbits = pb;
    }
    __shared__ unsigned long long stgL[kBi2Tile];
    __shared__ uint8_t            binL[kBi2Tile];
    __shared__ uint32_t           histL[kBins], offL[kBins], gbaseL[kBins], wsumL[4];
    __shared__ uint32_t           headL[kBi2HeadN], hposL[kBi2HeadN], survL[kBi2SurvLds];
    __shared__ uint32_t           redL[kBi2Threads / kWave], hcntL, hbaseL;
    for (int k = threadIdx.x; k < kBi2HeadN; k += kBi2Threads) {
        headL[k] = 0;
        hposL[k] = 0xFFFFFFFFu;
    }
    for (int k = threadIdx.x; k < kBi2SurvLds; k += kBi2Threads) survL[k] = (uint32_t)k < nsurvwords ? surv[k] : 0u;
    const uint32_t ntiles = (npos + kBi2Tile - 1) / kBi2Tile;
    const uint32_t sub    = blockIdx.x % nsub;
    const uint32_t lane   = threadIdx.x & (kWave - 1);
    uint32_t       nadm   = 0;
    uint32_t       c0[kBi2Per], cx[kBi2Per];  // class at the lane's positions; cx: lane 63 only, the class after its position
    auto           load_tile = [&](uint32_t tile) {
#pragma unroll
        for (int k = 0; k < kBi2Per; ++k) {
            const uint32_t i = tile * kBi2Tile + k * kBi2Threads + threadIdx.x;
            c0[k]            = (tile < ntiles && i < npos) ? cls[i] : 0u;
            cx[k]            = (lane == kWave - 1 && tile < ntiles && i + 1 < npos) ? cls[i + 1] : 0u;
        }
    };
    auto alive = [&](uint32_t c) -> uint32_t {  // class c survived order 1 (c = 0, the delimiter, never does: bit 0 is clear)
        const uint32_t w = c >> 5;
        return ((w < (uint32_t)kBi2SurvLds ? survL[w] : surv[w]) >> (c & 31u)) & 1u;
    };
    load_tile(blockIdx.x);
    __syncthreads();
    KP_INIT(0);
    // (Round 6 measured the hand-over of the prefetched tile — the wait for its loads, the survivor look-ups — moved in front of the previous tile's copy-out, as in
    // uni_onepass_kernel and level B, where it sits in front of the stores the in-order memory counter would otherwise make it wait for: here the twelve more live
    // registers of a 64-register kernel cost more, 0.55 -> 0.66 ms.)
    for (uint32_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const uint32_t base = tile * kBi2Tile;
        uint32_t       d0[kBi2Per], d1[kBi2Per], ok[kBi2Per];
        if (threadIdx.x < kBins) histL[threadIdx.x] = 0;
        if (threadIdx.x == kBins) hcntL = 0;
#pragma unroll
        for (int k = 0; k < kBi2Per; ++k) {
            d0[k]             = c0[k];
            const uint32_t nb = __shfl_down(c0[k], 1, kWave);  // the next position's class lives in the next lane (lane 63 loaded its own)
            d1[k]             = lane == kWave - 1 ? cx[k] : nb;
        }
#pragma unroll
        for (int k = 0; k < kBi2Per; ++k) {
            const uint32_t a0 = d0[k] ? alive(d0[k]) : 0u;
            const uint32_t an = __shfl_down(a0, 1, kWave);
            const uint32_t a1 = lane == kWave - 1 ? (d1[k] ? alive(d1[k]) : 0u) : an;
            ok[k]             = a0 & a1;
        }
        load_tile(tile + gridDim.x);  // the next tile's class ids are in flight while this one is partitioned in LDS
        __syncthreads();
        KP(0);
        unsigned long long rec[kBi2Per];
        uint32_t           rank[kBi2Per];  // [11:0] rank inside the tile's A bin, [23:16] the A bin; kInvalid: no record
        uint32_t           hrk[kBi2Per];   // head windows (lists wanted): rank among the tile's head windows << 12 | head pair
#pragma unroll
        for (int k = 0; k < kBi2Per; ++k) {
            const uint32_t i = base + k * kBi2Threads + threadIdx.x;
            rank[k]          = kInvalid;
            hrk[k]           = kInvalid;
            rec[k]           = 0;
            uint32_t mine    = 0xFFu;
            if (ok[k]) {
                nadm += slice == 0;  // every pass sees every window: the first one counts them
                if (d0[k] < (uint32_t)kBi2Head && d1[k] < (uint32_t)kBi2Head) {
                    if (slice == 0) {  // the head is counted by the first pass only
                        const uint32_t h = d0[k] * kBi2Head + d1[k];
                        atomicAdd(&headL[h], 1u);
                        atomicMin(&hposL[h], i);
                        if (hplist != nullptr) hrk[k] = (atomicAdd(&hcntL, 1u) << 12) | h;
                    }
                } else {
                    const uint64_t m = bi2_mix(((uint64_t)d0[k] << clsbits) | d1[k], K);
                    mine             = (uint32_t)(m >> Kp);
                    if ((uint32_t)(m >> Kp) == slice) {
                        const uint32_t a = (uint32_t)(m >> (Kp - 8)) & 255u;
                        rec[k]           = ((m & ((1ull << (Kp - 8)) - 1)) << pb) | i;
                        rank[k]          = atomicAdd(&histL[a], 1u) | (a << 16);
                    }
                }
            }
            if (sid != nullptr && i < npos) sid[i] = (uint8_t)mine;
        }
        __syncthreads();
        KP(1);
        bi2_scan256(histL, offL, wsumL);
        // one reservation per (tile, A bin), on the cursor of this block's sub-region — a memory-side atomic (~2 us): its answer is first needed by the copy-out, so it
        // travels while the records are staged (round 6)
        uint32_t rs_at = 0, rs_h = 0;
        if (threadIdx.x < kBins) {
            rs_h = histL[threadIdx.x];
            if (rs_h) rs_at = atomicAdd(&bs->curA[bi2_cur(sub * kBins + threadIdx.x)], rs_h);
        } else if (threadIdx.x == kBins && hplist != nullptr) {  // the tile's head windows: one reservation in the bucket's list (a tile lies inside one bucket, and the list holds
            rs_h = hcntL;                                         // every position of it)
            if (rs_h) rs_at = atomicAdd(&bs->pcur[bi2_pc(kBi2Shards * kBi2Buckets + (base >> hpl.pshift))], rs_h);
        }
#pragma unroll
        for (int k = 0; k < kBi2Per; ++k) {
            if (rank[k] != kInvalid) {
                const uint32_t a = rank[k] >> 16, p = offL[a] + (rank[k] & 0xFFFFu);
                stgL[p]          = rec[k];
                binL[p]          = (uint8_t)a;
            }
        }
        if (threadIdx.x < kBins) {
            uint32_t g = 0;
            if (rs_h) {
                const uint32_t slot = sub * kBins + threadIdx.x;
                if (rs_at + rs_h > region) bs->overflow = 1;
                g = slot * region + min(rs_at, region - min(region, rs_h));
            }
            gbaseL[threadIdx.x] = g;
        } else if (threadIdx.x == kBins && hplist != nullptr) {
            const uint32_t b = base >> hpl.pshift;
            hbaseL           = hpl.hbase + (b << hpl.pshift) + rs_at;
        }
        __syncthreads();
        KP(2);
        if (hplist != nullptr) {
            const uint32_t hb = hbaseL;
#pragma unroll
            for (int k = 0; k < kBi2Per; ++k)
                if (hrk[k] != kInvalid) {
                    hplist[hb + (hrk[k] >> 12)] = base + k * kBi2Threads + threadIdx.x;
                    hpcode[hb + (hrk[k] >> 12)] = kBi2HeadCode | (hrk[k] & 0xFFFu);
                }
        }
        const uint32_t n = offL[kBins - 1] + histL[kBins - 1];
        for (uint32_t j = threadIdx.x; j < n; j += kBi2Threads) {
            const uint32_t a                         = binL[j];
            recsA[(size_t)gbaseL[a] + (j - offL[a])] = stgL[j];
        }
        __syncthreads();
        KP(3);
    }
    uint32_t* const row = head_rows + (size_t)blockIdx.x * (2 * kBi2HeadN);
    for (int k = threadIdx.x; k < kBi2HeadN; k += kBi2Threads) {
        row[k]             = headL[k];
        row[kBi2HeadN + k] = hposL[k];
    }
    for (int off = 32; off > 0; off >>= 1) nadm += __shfl_down(nadm, off, kWave);
    if ((threadIdx.x & (kWave - 1)) == 0) redL[threadIdx.x / kWave] = nadm;
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t a = 0;
        for (int w = 0; w < kBi2Threads / kWave; ++w) a += redL[w];
        if (a) atomicAdd(&st->admitted, a);
    }
    bi2_offsets_tail(bs, region, nsub, histL, offL, wsumL, &hcntL);
    KP(4);
    KP_DONE();
}

// The later passes of a sliced order (corpora beyond ~128 M tokens per device): the first pass left every window's key slice in `sid`, so a pass only looks at the
// windows of its slice. nt tiles of 4096 positions feed one partition step: the matching windows' records are appended to an LDS queue (wave-aggregated), then
// ranked by A bin and written out — three barriers per nt tiles instead of four per tile, and no survivor-bitmap look-ups or mixes for the other slices' windows
// (the plain emit kernel spent 3.4 ms per pass on 10^9 positions whatever the slice held).
constexpr int kBi2SlQueue = 6144;  // records per partition step: nt * 16384 positions * (share of non-head admissible windows, ~0.7) / 2^sbits ~ 2900
constexpr int kBi2SlSpan  = 16;    // consecutive positions per lane and step: one 16-byte load of slice ids
__global__ __launch_bounds__(kBi2Threads, kBi2Threads / 128) void bi2_emit_sliced_kernel(const uint32_t* __restrict__ cls, const uint8_t* __restrict__ sid, uint32_t npos, uint32_t clsbits,
                                                                                          uint32_t sbits, uint32_t slice, uint32_t pb, uint32_t nt,
                                                                                          unsigned long long* __restrict__ recsA, uint32_t region, uint32_t nsub,
                                                                                          Bi2State* __restrict__ bs, DevState* __restrict__ st) {
    // sid is readable (0xFF) up to a multiple of 16 beyond npos
    if (st->done) return;
    const uint32_t K = max(2u * clsbits, 17u + sbits), Kp = K - sbits;
    if (threadIdx.x == 0) {
        bs->kbits   = Kp;
        bs->posbits = pb;
    }
    __shared__ unsigned long long qL[kBi2SlQueue];
    __shared__ uint8_t            qaL[kBi2SlQueue];
    __shared__ uint32_t           histL[kBins], offL[kBins], gbaseL[kBins], wsumL[4], qnL;
    const uint32_t lane = threadIdx.x & (kWave - 1), sub = blockIdx.x % nsub;
    constexpr uint32_t kStep = kBi2Threads * kBi2SlSpan;  // 16384 positions
    const uint32_t span = nt * kStep, nsuper = (npos + span - 1) / span;
    const uint32_t sl4  = slice * 0x01010101u;
    auto load_ids = [&](uint32_t p0) -> uint4 {
        return p0 < npos ? *reinterpret_cast<const uint4*>(sid + p0) : make_uint4(0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu);
    };
    auto match4 = [&](uint32_t w) -> uint32_t {  // bit b set iff byte b of w equals the slice
        const uint32_t x = w ^ sl4;
        return ((x & 0xFFu) == 0) | (((x >> 8) & 0xFFu) == 0) << 1 | (((x >> 16) & 0xFFu) == 0) << 2 | ((x >> 24) == 0) << 3;
    };
    for (uint32_t sup = blockIdx.x; sup < nsuper; sup += gridDim.x) {
        if (threadIdx.x < kBins) histL[threadIdx.x] = 0;
        if (threadIdx.x == 0) qnL = 0;
        __syncthreads();
        uint4 ids = load_ids(sup * span + threadIdx.x * kBi2SlSpan);
        for (uint32_t g = 0; g < nt; ++g) {
            const uint32_t p0 = sup * span + g * kStep + threadIdx.x * kBi2SlSpan;
            uint32_t       m  = match4(ids.x) | match4(ids.y) << 4 | match4(ids.z) << 8 | match4(ids.w) << 12;
            if (g + 1 < nt) ids = load_ids(p0 + kStep);  // the next step's slice ids are in flight while this one's class ids are fetched
            // this lane's records go to queue entries [base, base + popc(m)): one reservation per wave
            const uint32_t cnt = (uint32_t)__popc(m);
            uint32_t       inc = cnt;
            for (int off = 1; off < kWave; off <<= 1) {
                const uint32_t t = __shfl_up(inc, off, kWave);
                if ((int)lane >= off) inc += t;
            }
            const uint32_t wtot = __shfl(inc, kWave - 1, kWave);
            uint32_t       base = 0;
            if (wtot) {
                if (lane == 0) base = atomicAdd(&qnL, wtot);
                base = (uint32_t)__builtin_amdgcn_readfirstlane((int)base);
            }
            uint32_t idx = base + inc - cnt;
            while (m) {
                const uint32_t b = (uint32_t)__builtin_ctz(m);
                m &= m - 1;
                const uint32_t i  = p0 + b;
                const uint2    cc = make_uint2(cls[i], cls[i + 1]);
                if (idx < (uint32_t)kBi2SlQueue) {
                    const uint64_t mx = bi2_mix(((uint64_t)cc.x << clsbits) | cc.y, K);
                    qL[idx]           = ((mx & ((1ull << (Kp - 8)) - 1)) << pb) | i;
                    qaL[idx]          = (uint8_t)((uint32_t)(mx >> (Kp - 8)) & 255u);
                }
                ++idx;
            }
        }
        __syncthreads();
        const uint32_t nq = qnL;
        if (nq > (uint32_t)kBi2SlQueue && threadIdx.x == 0) bs->overflow = 1;  // (a slice far denser than a uniform mix makes it: the run repeats on the fallback path)
        const uint32_t n = min(nq, (uint32_t)kBi2SlQueue);
        uint32_t       rk[kBi2SlQueue / kBi2Threads];
#pragma unroll
        for (int q = 0; q < kBi2SlQueue / kBi2Threads; ++q) {
            const uint32_t j = q * kBi2Threads + threadIdx.x;
            rk[q]            = j < n ? atomicAdd(&histL[qaL[j]], 1u) : 0u;
        }
        __syncthreads();
        bi2_scan256(histL, offL, wsumL);
        if (threadIdx.x < kBins) {
            const uint32_t h = histL[threadIdx.x];
            uint32_t       gb = 0;
            if (h) {
                const uint32_t slot = sub * kBins + threadIdx.x;
                const uint32_t at   = atomicAdd(&bs->curA[bi2_cur(slot)], h);
                if (at + h > region) bs->overflow = 1;
                gb = slot * region + min(at, region - min(region, h));
            }
            gbaseL[threadIdx.x] = gb;
        }
        __syncthreads();
#pragma unroll
        for (int q = 0; q < kBi2SlQueue / kBi2Threads; ++q) {
            const uint32_t j = q * kBi2Threads + threadIdx.x;
            if (j < n) recsA[(size_t)gbaseL[qaL[j]] + rk[q]] = qL[j];  // a bin's records of this step: consecutive addresses
        }
        __syncthreads();
    }
    bi2_offsets_tail(bs, region, nsub, histL, offL, wsumL, &qnL);
}

// column sums / minima of the head rows: grid (kBi2HeadN / 256, kBi2HeadSplit), block (x, y) reduces rows y, y + kBi2HeadSplit, ... of 256 head keys
// into the zeroed Bi2State arrays (the lowest position travels inverted so that zero means "none")
__global__ __launch_bounds__(kBlock) void bi2_head_reduce_kernel(const uint32_t* __restrict__ head_rows, uint32_t nrows, Bi2State* __restrict__ bs, const DevState* __restrict__ st) {
    if (st->done) return;
    const uint32_t k = blockIdx.x * kBlock + threadIdx.x;
    uint32_t       s = 0, p = 0xFFFFFFFFu;
    for (uint32_t r = blockIdx.y; r < nrows; r += gridDim.y) {
        s += head_rows[(size_t)r * (2 * kBi2HeadN) + k];
        p = min(p, head_rows[(size_t)r * (2 * kBi2HeadN) + kBi2HeadN + k]);
    }
    if (s) {
        atomicAdd(&bs->headcnt[k], s);
        atomicMax(&bs->headposinv[k], ~p);
    }
}

// one block: records per A bin, their scan, the B-bin shift for this record count
__global__ __launch_bounds__(kBlock) void bi2_offsets_kernel(Bi2State* __restrict__ bs, uint32_t region, uint32_t nsub, const DevState* __restrict__ st) {
    if (st->done) return;
    uint32_t s = 0;
    for (uint32_t g = 0; g < nsub; ++g) {
        const uint32_t h = bs->curA[bi2_cur(g * kBins + threadIdx.x)];
        if (h > region) bs->overflow = 1;
        s += min(h, region);
    }
    uint32_t       tot;
    const uint32_t o       = block_exclusive_scan(s, &tot);
    bs->cntA[threadIdx.x]  = s;
    bs->offAt[threadIdx.x] = o;
    if (threadIdx.x == 0) {
        bs->offAt[kBins] = tot;
        bs->nrec         = tot;
        bi2_set_bshift(bs, tot);
    }
}

// exclusive scan of 512 LDS values by the first 512 threads of a block (>= 512 threads); every thread of the block must call it
__device__ __forceinline__ uint32_t bi2_scan512(const uint32_t* inL, uint32_t* outL, uint32_t* wsumL) {
    const uint32_t lane = threadIdx.x & (kWave - 1), wave = threadIdx.x / kWave;
    uint32_t       v = 0, incl = 0;
    if (threadIdx.x < 512) {
        v    = inL[threadIdx.x];
        incl = v;
        for (int off = 1; off < kWave; off <<= 1) {
            const uint32_t t = __shfl_up(incl, off, kWave);
            if ((int)lane >= off) incl += t;
        }
        if (lane == kWave - 1) wsumL[wave] = incl;
    }
    __syncthreads();
    uint32_t base = 0, tot = 0;
#pragma unroll
    for (int w = 0; w < 8; ++w) {
        const uint32_t t = wsumL[w];
        if (w < (int)wave) base += t;
        tot += t;
    }
    if (threadIdx.x < 512) outL[threadIdx.x] = base + incl - v;
    __syncthreads();
    return tot;
}

#ifndef COLIBRI_LB_PER
#define COLIBRI_LB_PER 8
#endif
// level B's own tile: 8192 records — a tile's run for one of the 512 B bins is then 128 bytes on average instead of 64 (the PMC passes showed 1.46 x the bytes written for
// the records moved: partial lines). One block of 88 KB of LDS per CU instead of two of 44 KB; 4 / 6 / 8 / 12 records per lane: 5.12 / 5.09 / 4.97 / 5.03 ms per 10^8-token step.
constexpr int kBi2LbPer = COLIBRI_LB_PER, kBi2LbTile = kBi2Threads * kBi2LbPer;
// ---- level B: one block partitions one slot by B bin ------------------------------------------------------------------------------
// boff: [nslots][513] exclusive offsets of the slot's B bins inside the slot (same slot layout in recsB as in recsA).
// B bin of a record: the nine mix bits below the A bin, shifted down by bshift when an order has few records.
// slotbase (optional; key-sharded runs, kshard.hpp): the slots are the chunks of a receive buffer — slot s starts at record slotbase[s] instead of s * region
// CB (key-sharded runs only): the (B, C) histogram and its 16 KB of LDS. One block per CU either way (90 / 106 KB). Round 5 measured TWO blocks per CU (74 KB each
// without the staged bins' array, B bins recomputed from the records): 0.86 against 0.74 ms per step over all orders — 512 slots of ~660 KB in flight no longer fit the
// 256 MB Infinity Cache the second sweep reads from.
template <bool CB = false>
__global__ __launch_bounds__(kBi2Threads, kBi2LbPer == 4 ? kBi2Threads / 128 : 1) void bi2_levelB_kernel(const unsigned long long* recsA, unsigned long long* __restrict__ recsB, uint32_t region,
                                                                                     const Bi2State* __restrict__ bs, uint32_t* __restrict__ boff, const DevState* __restrict__ st,
                                                                                     const uint32_t* __restrict__ slotbase = nullptr,
                                                                                     uint32_t* __restrict__ cbhist = nullptr /* key-sharded runs (kshard2.hpp), [nslots][512 x 8]: also the
                                                                                         slot's records per (B, C), C = the cskip mix bits below the B bin's — the sweep reads them anyway */) {
    if (st->done) return;
    __shared__ uint32_t           cbL[CB ? 8 * kBi2BBins : 1];
    __shared__ unsigned long long stgL[kBi2LbTile];
    __shared__ uint16_t           binL[kBi2LbTile];
    __shared__ uint32_t           histL[kBi2BBins], offL[kBi2BBins], curL[kBi2BBins], gbL[kBi2BBins], wsumL[8];
    const uint32_t  slot = blockIdx.x;
    const uint32_t  n    = min(bs->curA[bi2_cur(slot)], region);
    const uint32_t  bsh  = bs->bshift;
    const uint32_t  bbit = bs->posbits + bs->kbits - 17;  // the B bin = the nine mix bits below the A bin = record bits [bbit + 8 : bbit]
    const size_t    base = slotbase != nullptr ? (size_t)slotbase[slot] : (size_t)slot * region;
    uint32_t* const bo   = boff + (size_t)slot * (kBi2BBins + 1);
    if (threadIdx.x < kBi2BBins) histL[threadIdx.x] = 0;
    const uint32_t cmask = (1u << bs->cskip) - 1u;
    if (CB && cbhist != nullptr)
        for (uint32_t e = threadIdx.x; e < (cmask + 1u) * kBi2BBins; e += kBi2Threads) cbL[e] = 0;
    __syncthreads();
    KP_INIT(1);
    // sweep 1: histogram of the slot, two tiles of loads ahead of their LDS atomics
    // (Round 6 measured the sweep on 2-byte copies of the B bins written by the emit kernels beside the records: the sweep fell from 36 % to 12 % of this kernel, but the
    // move sweep then waits for HBM — the histogram sweep is also what brings the slot into the Infinity Cache — and the emit kernels' extra 32-byte runs cost more than was
    // left: 3.71 -> 3.75 ms per step. Not kept.)
    for (uint32_t j0 = 0; j0 < n; j0 += 2 * kBi2LbTile) {
        unsigned long long r[2 * kBi2LbPer];
#pragma unroll
        for (int k = 0; k < 2 * kBi2LbPer; ++k) {
            const uint32_t j = j0 + k * kBi2Threads + threadIdx.x;
            r[k]             = (j < n) ? recsA[base + j] : 0ull;
        }
#pragma unroll
        for (int k = 0; k < 2 * kBi2LbPer; ++k) {
            const uint32_t j = j0 + k * kBi2Threads + threadIdx.x;
            if (j < n) {
                const uint32_t b = ((uint32_t)(r[k] >> bbit) & 511u) >> bsh;
                atomicAdd(&histL[b], 1u);
                if (CB && cbhist != nullptr) atomicAdd(&cbL[(b << bs->cskip) | ((uint32_t)(r[k] >> (bbit + bsh - bs->cskip)) & cmask)], 1u);
            }
        }
    }
    __syncthreads();
    KP(0);
    if (CB && cbhist != nullptr)
        for (uint32_t e = threadIdx.x; e < (cmask + 1u) * kBi2BBins; e += kBi2Threads) cbhist[(size_t)slot * (8 * kBi2BBins) + e] = cbL[e];
    bi2_scan512(histL, offL, wsumL);
    if (threadIdx.x < kBi2BBins) {
        bo[threadIdx.x]   = offL[threadIdx.x];
        curL[threadIdx.x] = offL[threadIdx.x];
    }
    if (threadIdx.x == 0) bo[kBi2BBins] = n;
    // sweep 2: tile-local counting sort, runs appended at the block's own cursors (nobody else writes this slot); the next tile is prefetched
    unsigned long long r[kBi2LbPer];
    auto               load_tile = [&](uint32_t j0) {
#pragma unroll
        for (int k = 0; k < kBi2LbPer; ++k) {
            const uint32_t j = j0 + k * kBi2Threads + threadIdx.x;
            r[k]             = (j < n) ? recsA[base + j] : 0ull;
        }
    };
    // x: the tile being sorted; r: the next one, in flight. The hand-over x = r (a wait for r's loads) sits BEFORE a tile's copy-out: behind it, the in-order memory counter
    // made it a wait for the tile's stores as well (round 6)
    unsigned long long x[kBi2LbPer];
    load_tile(0);
#pragma unroll
    for (int k = 0; k < kBi2LbPer; ++k) x[k] = r[k];
    load_tile(kBi2LbTile);
    for (uint32_t j0 = 0; j0 < n; j0 += kBi2LbTile) {
        uint32_t           rank[kBi2LbPer];
        if (threadIdx.x < kBi2BBins) histL[threadIdx.x] = 0;
        __syncthreads();
        KP(1);
#pragma unroll
        for (int k = 0; k < kBi2LbPer; ++k) {
            const uint32_t j = j0 + k * kBi2Threads + threadIdx.x;
            rank[k]          = kInvalid;
            if (j < n) {
                const uint32_t b = ((uint32_t)(x[k] >> bbit) & 511u) >> bsh;
                rank[k]          = atomicAdd(&histL[b], 1u) | (b << 16);
            }
        }
        __syncthreads();
        KP(2);
        bi2_scan512(histL, offL, wsumL);
        KP(3);
        if (threadIdx.x < kBi2BBins) {
            gbL[threadIdx.x] = curL[threadIdx.x];
            curL[threadIdx.x] += histL[threadIdx.x];
        }
#pragma unroll
        for (int k = 0; k < kBi2LbPer; ++k) {
            if (rank[k] != kInvalid) {
                const uint32_t b = rank[k] >> 16, p = offL[b] + (rank[k] & 0xFFFFu);
                stgL[p]          = x[k];
                binL[p]          = (uint16_t)b;
            }
        }
#pragma unroll
        for (int k = 0; k < kBi2LbPer; ++k) x[k] = r[k];
        load_tile(j0 + 2 * kBi2LbTile);
        __syncthreads();
        KP(4);
        const uint32_t m = min(n - j0, (uint32_t)kBi2LbTile);
        for (uint32_t j = threadIdx.x; j < m; j += kBi2Threads) {
            const uint32_t b                     = binL[j];
            recsB[base + gbL[b] + (j - offL[b])] = stgL[j];
        }
        __syncthreads();
        KP(5);
    }
    KP_DONE();
}

// sparse ranges of the final bins: block a, lane b: records of bin (a, b) over all sub-regions, scanned inside the A bin
__global__ __launch_bounds__(kBi2BBins) void bi2_binoff_kernel(Bi2State* __restrict__ bs, const uint32_t* __restrict__ boff, uint32_t nsub, const DevState* __restrict__ st) {
    if (st->done) return;
    __shared__ uint32_t inL[kBi2BBins], outL[kBi2BBins], wsumL[8];
    const uint32_t      a = blockIdx.x, b = threadIdx.x, nB = (uint32_t)kBi2BBins >> bs->bshift;
    uint32_t            t = 0;
    if (b < nB)
        for (uint32_t s = 0; s < nsub; ++s) {
            const uint32_t* bo = boff + (size_t)(s * kBins + a) * (kBi2BBins + 1);
            t += bo[b + 1] - bo[b];
        }
    inL[b] = t;
    if (t > (bs->hugebin ? bs->hugebin : kBi2HugeBin)) {
        const uint32_t k = atomicAdd(&bs->nhuge, 1u);
        if (k < (uint32_t)kBi2HugeCap) bs->huge[k] = a * kBi2BBins + b;
    }
    if (t > kBi2BigBin) {
        const uint32_t k = atomicAdd(&bs->nbig, 1u);
        if (k < (uint32_t)kBi2BigCap) bs->big[k] = a * kBi2BBins + b;
    }
    __syncthreads();
    bi2_scan512(inL, outL, wsumL);
    bs->binoff[a * kBi2BBins + b] = bs->offAt[a] + outL[b];
    if (a == kBins - 1 && b == 0) bs->binoff[kBi2Final] = bs->offAt[kBins];
}

// ---- count: one WAVE per final bin --------------------------------------------------------------------------------------------------
// A final bin holds ~660 records of ~400 distinct keys. Measured on MI355X (tools/bigram2_bench.hip): a workgroup per bin is bound by control
// code and barriers (16 waves run ~1500 instructions each to insert one or two records per lane: 1.2 ms per 87 M records), and a wave inserting a
// row of 64 records pays the LONGEST probe sequence of its lanes (linear probing: ~2600 cycles per row). Here a wave owns a bin from start to end —
// no barrier, one copy of the control code per bin, the bin's records (up to 12 per lane) in registers — and needs only 9 KB of LDS, so 16 waves
// per CU hide each other's LDS round trips. The table has buckets of 4 slots read as one 16-byte vector; slots of a bucket fill left to right and
// are never freed, so the first slot that holds the key or is empty decides; a full bucket continues in the next one. Two rows are inserted per
// round (their compare-and-swaps are in flight together). Survivors go to the bin's own range of the sparse result arrays; the positions of
// the windows of surviving keys are appended, unsorted, to the wave's private list (ballot-compacted, coalesced): bi2_pospart_kernel sorts them
// into position buckets afterwards. Bins are handed out dynamically, the big ones (a hot key outside the dense head) first; their records
// beyond the register window are streamed four rows at a time with the next four in flight.
constexpr int      kBi2WReps = 256;  // survivors of a bin whose lowest position is tracked in LDS (beyond: device atomics on the result array)
constexpr uint32_t kBi2Kept  = 0x80000000u;
__device__ __forceinline__ uint32_t bi2_wave_excl_scan(uint32_t v, uint32_t* total) {
    const uint32_t lane = threadIdx.x & (kWave - 1);
    uint32_t       incl = v;
    for (int off = 1; off < kWave; off <<= 1) {
        const uint32_t t = __shfl_up(incl, off, kWave);
        if ((int)lane >= off) incl += t;
    }
    *total = __shfl(incl, kWave - 1, kWave);
    return incl - v;
}
__device__ __forceinline__ uint32_t bi2_scan4(const uint4& v, uint32_t key, bool& hit) {  // index (0..3) of the first slot holding `key` or empty; 4: none
    const uint32_t w[4] = {v.x, v.y, v.z, v.w};
    uint32_t       idx  = 4;
    hit                 = false;
#pragma unroll
    for (int i = 3; i >= 0; --i) {
        if (w[i] == key || w[i] == kBi2Empty) {
            idx = (uint32_t)i;
            hit = w[i] == key;
        }
    }
    return idx;
}
__device__ __forceinline__ uint32_t bi2_bucket_of(uint32_t key, int lgb, uint32_t bmask) { return ((key * 0x9E3779B1u) >> (32 - lgb)) & bmask; }
// finds or inserts two keys per lane (A and B, each optional); slots are returned through sa / sb (kInvalid: table full)
// Round 5: ONE look at the bucket (vA / vB, read by the caller), then a walk of compare-and-swaps. The look settles the hits (about half of a bin's records) and names the
// first free slot; a lane that loses that slot to another key (lanes of one row racing for the same bucket: the usual case, 128 inserts into 256 buckets) tries the NEXT
// slot right away — its compare-and-swap returns what the slot holds, which is all a new look would tell — and walks on into the next bucket when the bucket is full.
// Slots fill left to right and are never freed, so a key is in the first slot of its walk that holds it or is empty. Round 4 went back to the look after every lost race:
// ~4 rounds of (16-byte read, 16 compares and selects, compare-and-swap) per pair of rows, 42 % of the kernel.
// ownA / ownB: this lane's compare-and-swap put the key into the table — exactly one record per distinct key of the bin is its key's "owner"
__device__ __forceinline__ void bi2_insert2(uint32_t* keyT, uint32_t bmask, bool actA, uint32_t keyA, uint32_t bkA, uint4 vA, uint32_t& sa, bool actB, uint32_t keyB, uint32_t bkB, uint4 vB,
                                            uint32_t& sb, uint32_t& ownA, uint32_t& ownB) {
    // The walk's state is INTEGER per lane (the slot to try, bit 31 = still walking) and every update a select: as booleans (act / hit / own per row) the state lived in
    // 64-bit scalar masks that each branch merged with three scalar instructions — the kernel executed more scalar than vector instructions.
    constexpr uint32_t kWalk = 0x80000000u;
    const uint32_t smask = bmask * 4u + 3u;
    bool           hitA = false, hitB = false;
    const uint32_t iA = bi2_scan4(vA, keyA, hitA), iB = bi2_scan4(vB, keyB, hitB);
    const uint32_t pA = (bkA * 4u + iA) & smask, pB = (bkB * 4u + iB) & smask;  // (no slot of the bucket is free or holds the key: the next bucket's first)
    sa            = (actA && hitA) ? pA : kInvalid;
    sb            = (actB && hitB) ? pB : kInvalid;
    uint32_t qa   = (actA && !hitA) ? (pA | kWalk) : 0u, qb = (actB && !hitB) ? (pB | kWalk) : 0u;
    ownA = ownB   = 0u;
    uint32_t steps = 0;
    while (__any(((qa | qb) & kWalk) != 0u)) {
        uint32_t oldA = 0u, oldB = 0u;
        if (qa & kWalk) oldA = atomicCAS(&keyT[qa & smask], kBi2Empty, keyA);
        if (qb & kWalk) oldB = atomicCAS(&keyT[qb & smask], kBi2Empty, keyB);
        const bool wA = (qa & kWalk) != 0u, wB = (qb & kWalk) != 0u;
        const bool dA = wA && (oldA == kBi2Empty || oldA == keyA), dB = wB && (oldB == kBi2Empty || oldB == keyB);
        sa   = dA ? (qa & smask) : sa;
        sb   = dB ? (qb & smask) : sb;
        ownA = (dA && oldA == kBi2Empty) ? 1u : ownA;
        ownB = (dB && oldB == kBi2Empty) ? 1u : ownB;
        qa   = dA ? 0u : (wA ? (((qa + 1u) & smask) | kWalk) : qa);
        qb   = dB ? 0u : (wB ? (((qb + 1u) & smask) | kWalk) : qb);
        if (++steps > smask) break;  // (every slot holds another key: the caller reports the overflow)
    }
}
// slot of a key that is known to be in the table
__device__ __forceinline__ uint32_t bi2_find(const uint32_t* keyT, uint32_t key, uint32_t bk, uint32_t bmask) {
    for (;;) {
        const uint4    v = *reinterpret_cast<const uint4*>(keyT + bk * 4);
        bool           hit;
        const uint32_t i = bi2_scan4(v, key, hit);
        if (i < 4 && hit) return bk * 4 + i;
        bk = (bk + 1) & bmask;
    }
}
#ifdef BI2_PROF
__device__ unsigned long long bi2_prof[16];
#endif
// BASED (key-sharded runs, kshard.hpp): slot s starts at record slotbase[s] of recsB instead of s * region, and the position lists are CHUNKS of a pool:
// wlist = [wcap chunks][kBi2Chunk], wcnt = entries per chunk, a wave takes a fresh chunk (one atomic) whenever its current one cannot hold a row. How many
// windows a wave ends up listing depends on how many bins it drew, i.e. on how the waves were scheduled — ranks sharing a device (tests) starve each other's
// late waves —, so a fixed capacity per wave is no bound there; the pool needs room for the survivors plus one partly filled chunk per wave.
constexpr uint32_t kBi2Chunk = 4096;
// ROWS: records per lane held in registers (bins of up to 64 x ROWS records are read once; an owner of a 125 M-token-per-rank run holds ~810 per bin: 16 rows)
// KEY4 (key-sharded runs, kshard2.hpp; with BASED): the owner's form. recsB is an array of 4-BYTE in-bin keys, the sources' streams one after the other, in final-bin
// order (the sources partitioned completely); a record's "position" is its place: source << 28 | index in the source's stream (bs->posbits = 31). Nothing is listed:
// every record's entry of code_at (= wlist) becomes (final bin << 10 | rank of its key among the bin's survivors), or kInvalid — in stream order, so the source finds its
// windows again by counting. A record's "position" is its place in the receive buffer (31 bits; lower place = lower source rank first).
// SLOTS: the LDS table of a final bin (1024: 900 distinct keys, what a pass of ~2 x 10^8 positions of the bench distribution fills; 2048 for the passes beyond — half
// the waves per CU, which is why it is not the default). A bin's survivors are numbered in 10 bits either way (more than 1023 of them: overflow 2, like a full table).
// PDROP (chained orders of corpora beyond 2.15 x 10^8 positions, eight sub-regions): the records' positions lack three bits (Bi2State::pdshift); run s of a bin lies in
// sub-region s, which names them.
template <int NSUB, bool BASED = false, int ROWS = kBi2WRows, bool KEY4 = false, int SLOTS = kBi2WSlots, bool PDROP = false>
__global__ __launch_bounds__(kWave, bi2_count_weu(SLOTS, KEY4, BASED)) void bi2_count_kernel(const unsigned long long* __restrict__ recsB, uint32_t region, const uint32_t* __restrict__ boff, Bi2State* __restrict__ bs,
                                                              DevState* __restrict__ st, uint32_t threshold, uint32_t* __restrict__ sp_rep, uint32_t* __restrict__ sp_cnt,
                                                              uint32_t* __restrict__ wlist, uint32_t* __restrict__ wcnt, uint32_t wcap, bool want_positions,
                                                              uint32_t* __restrict__ wcode = nullptr /* optional, beside wlist: (final bin << 10) | rank of the window's key among the
                                                                                                        bin's survivors — what bi2_pospart_kernel (dense) and chain_ids_kernel turn into the window's RESULT index */,
                                                              const uint32_t* __restrict__ slotbase = nullptr,
                                                              bool big_elsewhere = false /* bi2_count_big_kernel has counted the huge bins */) {
    if (st->done) return;
    static_assert(3 * NSUB + 1 <= kWave, "bound loaders are lanes of the wave");
    static_assert(!KEY4 || (BASED && NSUB == 8), "the owner's form");
    const uint32_t* const keys4   = reinterpret_cast<const uint32_t*>(recsB);
    uint32_t* const       code_at = wlist;
    __shared__ __attribute__((aligned(16))) uint32_t keyT[SLOTS];
    // C16 (the single-device form with 1024-slot tables): 16-bit counters — 7 KB of LDS per wave instead of 9, so that 20 waves are resident per CU instead of 17
    // (with <= 96 registers: bi2_count_weu). Order-2 count 0.548 -> 0.503 ms per 10^8 tokens. A bin of 32 768 records or more cannot be counted then (bit 15 marks a
    // survivor): such bins belong to the workgroup kernel; if its list overflowed, the order falls back. -DCOLIBRI_BI2_CNT16=0: 32-bit counters everywhere.
    constexpr bool C16 = COLIBRI_BI2_CNT16 && SLOTS == 1024 && !KEY4 && !BASED;
    __shared__ __attribute__((aligned(16))) uint32_t cntT[C16 ? SLOTS / 2 : SLOTS];
    uint16_t* const cnt16 = reinterpret_cast<uint16_t*>(cntT);
    constexpr uint32_t kKeptBit = C16 ? 0x8000u : kBi2Kept;
    auto cnt_add = [&](uint32_t sl_) {
        if (C16)
            atomicAdd(&cntT[sl_ >> 1], 1u << ((sl_ & 1u) * 16u));
        else
            atomicAdd(&cntT[sl_], 1u);
    };
    auto cnt_get = [&](uint32_t sl_) -> uint32_t { return C16 ? (uint32_t)cnt16[sl_] : cntT[sl_]; };
    auto cnt_keep = [&](uint32_t sl_, uint32_t r_) {
        if (C16)
            cnt16[sl_] = (uint16_t)(kKeptBit | r_);
        else
            cntT[sl_] = kKeptBit | r_;
    };
    __shared__ uint32_t                              repS[SLOTS / 4];
    constexpr uint32_t kMaxLoad = kBi2MaxLoad * (uint32_t)SLOTS / (uint32_t)kBi2Slots;
    const uint32_t bsh = bs->bshift, nB = (uint32_t)kBi2BBins >> bsh, nfinal = (uint32_t)kBins * nB;
    const uint32_t lane = threadIdx.x, wid = blockIdx.x, nwaves = gridDim.x;
    const uint32_t pb = bs->posbits;
    const unsigned long long pmask = (1ull << pb) - 1;
    const uint32_t pdsh = PDROP ? bs->pdshift : 0u;
    static_assert(!PDROP || (!BASED && !KEY4 && NSUB == 8), "three dropped bits = one of eight sub-regions");
    uint32_t* const mylist = wlist + (size_t)wid * wcap;
    uint32_t* const mycode = wcode != nullptr ? wcode + (size_t)wid * wcap : nullptr;
    static_assert(kBi2MaxLoad < 1024 && kBi2Final <= (1 << 22), "a (bin, rank) code fits 32 bits");
    uint32_t        cursor = (want_positions && !BASED) ? wcnt[wid] : 0u;  // entries in this wave's position list (wave-uniform); the passes of a sliced order append
    uint32_t        chunk  = kInvalid;                                      // BASED: the chunk being filled (cursor = entries in it)
    bool            lost   = false;  // the list ran out of room
    const bool      skip_huge = big_elsewhere && bs->nhuge <= (uint32_t)kBi2HugeCap;  // bi2_count_big_kernel has counted those
    const uint32_t  hugebin   = bs->hugebin ? bs->hugebin : kBi2HugeBin;
#ifdef BI2_PROF
    unsigned long long tacc[12] = {0}, tlast = wall_clock64();
#define BI2_W(k) do { const unsigned long long t_ = wall_clock64(); tacc[k] += t_ - tlast; tlast = t_; } while (0)
#else
#define BI2_W(k) do {} while (0)
#endif
    // one final bin (a, b), start to end; `big_pass`: the bin comes from the list of big bins (the regular walk skips those)
    auto process_bin = [&](const uint32_t a, const uint32_t b, const bool big_pass, const bool skip_big) {
        BI2_W(0);
        // run bounds: 2 NSUB + 1 lanes load, v_readlane broadcasts (scalar registers)
        uint32_t v = 0;
        if (lane < (uint32_t)(2 * NSUB))
            v = boff[(size_t)((lane % NSUB) * kBins + a) * (kBi2BBins + 1) + b + (lane >= (uint32_t)NSUB ? 1u : 0u)];
        else if (lane == (uint32_t)(2 * NSUB))
            v = bs->binoff[a * kBi2BBins + b];
        else if (BASED && lane < (uint32_t)(3 * NSUB + 1))
            v = slotbase[(lane - (uint32_t)(2 * NSUB + 1)) * kBins + a];
        // record j (< total) of the bin lies at index j + adj[s] of recsB, s = the first run with j < end[s] (scalar registers; mod 2^32 — an index into the record
        // arrays is below 2^32: they are sized for < 2^30 positions)
        uint32_t adj[NSUB], end[NSUB], total = 0;
#pragma unroll
        for (int s = 0; s < NSUB; ++s) {
            const uint32_t rs = (uint32_t)__builtin_amdgcn_readlane((int)v, s), rn = (uint32_t)__builtin_amdgcn_readlane((int)v, NSUB + s) - rs;
            const uint32_t sb = BASED ? (uint32_t)__builtin_amdgcn_readlane((int)v, 2 * NSUB + 1 + s) : ((uint32_t)s * kBins + a) * region;
            adj[s]            = sb + rs - total;
            total += rn;
            end[s] = total;
        }
        const uint32_t spo = (uint32_t)__builtin_amdgcn_readlane((int)v, 2 * NSUB);
        BI2_W(1);
        if (total == 0 || (!big_pass && skip_big && total > kBi2BigBin) || (skip_huge && total > hugebin)) return;
        if (C16 && total >= 32768u) {  // (only when the workgroup kernel's list overflowed)
            if (lane == 0) bs->overflow = 2;
            return;
        }
        auto locate = [&](uint32_t j) -> uint32_t {
            // (the values pass through readfirstlane: a select between two reads of a local array is folded into one read at a selected address, which pins the array
            // to scratch memory — every locate() then waits for a scratch load)
            uint32_t o = (uint32_t)__builtin_amdgcn_readfirstlane((int)adj[NSUB - 1]);
#pragma unroll
            for (int s = NSUB - 2; s >= 0; --s) {
                const uint32_t as = (uint32_t)__builtin_amdgcn_readfirstlane((int)adj[s]), es = (uint32_t)__builtin_amdgcn_readfirstlane((int)end[s]);
                o                 = j < es ? as : o;
            }
            return j + o;
        };
        auto fixpos = [&](uint32_t p, uint32_t j) -> uint32_t {  // the position of record j of the bin as the corpus knows it
            if (!PDROP || pdsh == 0u) return p;
            uint32_t sub = 0;
#pragma unroll
            for (int s = 0; s < NSUB - 1; ++s) sub += j >= (uint32_t)__builtin_amdgcn_readfirstlane((int)end[s]) ? 1u : 0u;
            const uint32_t sh = pdsh - 1u;
            return ((p >> sh) << (sh + 3u)) | (sub << sh) | (p & ((1u << sh) - 1u));
        };
        auto load = [&](uint32_t j) -> unsigned long long {  // record j of the bin as (key << pb | position)
            if (!KEY4) return recsB[locate(j)];
            const uint32_t idx = (uint32_t)locate(j);
            return ((unsigned long long)(keys4[idx] & 0x7FFFFFFFu) << 31) | idx;
        };
        unsigned long long x[ROWS];
#pragma unroll
        for (int q = 0; q < ROWS; ++q) {
            const uint32_t j = q * kWave + lane;
            x[q]             = j < total ? load(j) : 0ull;
        }
        BI2_W(2);
        uint32_t nslots = 256;
        while (nslots < (uint32_t)SLOTS && nslots < total + (total >> 1)) nslots <<= 1;
        int lgb = 6;  // log2 of the number of buckets
        while ((4u << lgb) < nslots) ++lgb;
        const uint32_t bmask = (1u << lgb) - 1u;
        for (uint32_t s = lane * 4; s < nslots; s += kWave * 4) {
            *reinterpret_cast<uint4*>(keyT + s) = make_uint4(kBi2Empty, kBi2Empty, kBi2Empty, kBi2Empty);
            if (C16)
                *reinterpret_cast<uint2*>(cnt16 + s) = make_uint2(0u, 0u);
            else
                *reinterpret_cast<uint4*>(cntT + s) = make_uint4(0u, 0u, 0u, 0u);
        }
        BI2_W(3);
        // pass 1: two rows per round
        uint32_t sl[ROWS], own = 0;  // own: bit q = this lane's record of row q put its key into the table (one "owner" per distinct key)
        bool     fail = false;
#pragma unroll
        for (int q = 0; q < ROWS; q += 2) {
            sl[q] = sl[q + 1] = kInvalid;
            if ((uint32_t)(q * kWave) < total) {
                const bool     actA = (uint32_t)(q * kWave) + lane < total, actB = (uint32_t)((q + 1) * kWave) + lane < total;
                const uint32_t keyA = (uint32_t)(x[q] >> pb) & 0x7FFFFFFFu, keyB = (uint32_t)(x[q + 1] >> pb) & 0x7FFFFFFFu;
                const uint32_t bkA = bi2_bucket_of(keyA, lgb, bmask), bkB = bi2_bucket_of(keyB, lgb, bmask);
                const uint4    vA = *reinterpret_cast<const uint4*>(keyT + bkA * 4), vB = *reinterpret_cast<const uint4*>(keyT + bkB * 4);
                uint32_t       oA, oB;
                bi2_insert2(keyT, bmask, actA, keyA, bkA, vA, sl[q], actB, keyB, bkB, vB, sl[q + 1], oA, oB);
                own |= (oA << q) | (oB << (q + 1));
                fail |= (actA && sl[q] == kInvalid) || (actB && sl[q + 1] == kInvalid);  // (an idle lane's slot is kInvalid too)
                if (sl[q] != kInvalid) cnt_add(sl[q]);
                if (sl[q + 1] != kInvalid) cnt_add(sl[q + 1]);
            }
        }
        BI2_W(4);
        if (total > (uint32_t)(ROWS * kWave)) {  // larger bins stream the remainder: four rows per round, the next four already in flight
            unsigned long long y[4];
            auto               load4 = [&](uint32_t j0) {
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const uint32_t j = j0 + k * kWave + lane;
                    y[k]             = j < total ? load(j) : ~0ull;
                }
            };
            load4(ROWS * kWave);
            for (uint32_t j0 = ROWS * kWave; j0 < total; j0 += 4 * kWave) {
                unsigned long long z[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) z[k] = y[k];
                load4(j0 + 4 * kWave);
#pragma unroll
                for (int k = 0; k < 4; k += 2) {
                    const bool     actA = z[k] != ~0ull, actB = z[k + 1] != ~0ull;
                    const uint32_t keyA = (uint32_t)(z[k] >> pb) & 0x7FFFFFFFu, keyB = (uint32_t)(z[k + 1] >> pb) & 0x7FFFFFFFu;
                    const uint32_t bkA = bi2_bucket_of(keyA, lgb, bmask), bkB = bi2_bucket_of(keyB, lgb, bmask);
                    uint32_t       tA, tB;
                    uint32_t       oA, oB;
                    bi2_insert2(keyT, bmask, actA, keyA, bkA, *reinterpret_cast<const uint4*>(keyT + bkA * 4), tA, actB, keyB, bkB, *reinterpret_cast<const uint4*>(keyT + bkB * 4), tB, oA, oB);
                    fail |= (actA && tA == kInvalid) || (actB && tB == kInvalid);
                    if (tA != kInvalid) cnt_add(tA);
                    if (tB != kInvalid) cnt_add(tB);
                }
            }
        }
        BI2_W(5);
        if (__any(fail)) {
            if (lane == 0) bs->overflow = 2;
            return;
        }
        // survivors. Every distinct key has exactly one owner among the records, so a bin whose records are all in registers is ranked from the records' side: per row one
        // look at the owners' counters, two ballots — instead of two sweeps over the 1024 slots (16 per lane, a branch per slot: 15 % of the kernel in round 4).
        // Survivor r of the bin keeps its lowest position in LDS while r < SLOTS / 4, in the result array (device atomics) beyond.
        constexpr uint32_t kReps = (uint32_t)(SLOTS / 4);
        uint32_t           distinct = 0, ktotal = 0;
        if (total <= (uint32_t)(ROWS * kWave)) {
#pragma unroll
            for (int q = 0; q < ROWS; ++q) {
                if ((uint32_t)(q * kWave) < total) {
                    const bool     owner = (own >> q) & 1u;
                    const uint32_t c     = owner ? cnt_get(sl[q]) : 0u;
                    const bool     kept  = owner && c >= threshold;
                    const uint64_t mk    = __ballot(kept);
                    distinct += (uint32_t)__popcll(__ballot(owner));
                    if (kept) {
                        const uint32_t r = ktotal + (uint32_t)__popcll(mk & ((1ull << lane) - 1ull));
                        sp_cnt[spo + r]  = c;
                        cnt_keep(sl[q], r);  // from here on: the key survived, and which survivor of the bin it is
                        if (r < kReps)
                            repS[r] = 0xFFFFFFFFu;
                        else
                            sp_rep[spo + r] = 0xFFFFFFFFu;
                    }
                    ktotal += (uint32_t)__popcll(mk);
                }
            }
            if (distinct > kMaxLoad || ktotal > 1023u) {
                if (lane == 0) bs->overflow = 2;
                return;
            }
        } else {  // a bin beyond the register window: lane l sweeps the nslots / 64 consecutive slots from l * (nslots / 64), four at a time
            const uint32_t per  = nslots / kWave;
            uint32_t       used = 0, keep = 0;
            for (uint32_t k = 0; k < per; k += 4) {
                const uint32_t s  = lane * per + k;
                const uint4    kk = *reinterpret_cast<const uint4*>(keyT + s), cc = make_uint4(cnt_get(s), cnt_get(s + 1), cnt_get(s + 2), cnt_get(s + 3));
                used += (kk.x != kBi2Empty) + (kk.y != kBi2Empty) + (kk.z != kBi2Empty) + (kk.w != kBi2Empty);
                keep += (cc.x >= threshold) + (cc.y >= threshold) + (cc.z >= threshold) + (cc.w >= threshold);  // an empty slot counts 0 (threshold >= 1)
            }
            const uint32_t excl = bi2_wave_excl_scan(keep, &ktotal);
            bi2_wave_excl_scan(used, &distinct);
            if (distinct > kMaxLoad || ktotal > 1023u) {
                if (lane == 0) bs->overflow = 2;
                return;
            }
            uint32_t r = excl;
            for (uint32_t k = 0; k < per; k += 4) {
                const uint32_t s0 = lane * per + k;
                const uint4    cc = make_uint4(cnt_get(s0), cnt_get(s0 + 1), cnt_get(s0 + 2), cnt_get(s0 + 3));
                const uint32_t c4[4] = {cc.x, cc.y, cc.z, cc.w};
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    if (c4[i] >= threshold) {
                        sp_cnt[spo + r] = c4[i];
                        cnt_keep(s0 + i, r);
                        if (r < kReps)
                            repS[r] = 0xFFFFFFFFu;
                        else
                            sp_rep[spo + r] = 0xFFFFFFFFu;
                        ++r;
                    }
                }
            }
        }
        if (lane == 0) {
            atomicAdd(&bs->found_part[a], distinct);
            bs->binkept[a * kBi2BBins + b] = ktotal;
            if (ktotal) atomicAdd(&bs->kept_part[a], ktotal);
        }
        BI2_W(6);
        if (ktotal == 0) {
            if (KEY4) {  // nothing of this bin survived: its records say so
#pragma unroll
                for (int q = 0; q < ROWS; ++q)
                    if ((uint32_t)(q * kWave) + lane < total) code_at[(uint32_t)(x[q] & pmask)] = kInvalid;
                for (uint32_t j = ROWS * kWave + lane; j < total; j += kWave) code_at[locate(j)] = kInvalid;
            }
            return;
        }
        BI2_W(7);
        if (ktotal > kReps) __threadfence();  // the initial values of the result entries precede the atomics below
        // pass 2: every window of a surviving key: lowest position of the key; the position joins the wave's list
        const uint32_t fcode = (a * (uint32_t)kBi2BBins + b) << 10;
        // c: the record's table entry after the ranking (kBi2Kept | rank of a surviving key); direct: the bin's records are all in registers (no hot key: the lowest
        // position goes straight into an LDS minimum — a hot key's windows, which all aim at one word, look first)
        auto settle = [&](bool valid, uint32_t pos, uint32_t c, bool direct) {
            const bool     kept = (c & kKeptBit) != 0;
            const uint32_t r    = c & ~kKeptBit;
            if (KEY4 && valid) code_at[pos] = kept ? (fcode | r) : kInvalid;  // (a run's records lie one after the other: the stores of a row are coalesced)
            if (kept) {  // (read first: a hot key's windows all aim at one word, and after the first rows hardly any of them lowers it)
                if (r < kReps) {
                    if (direct || pos < repS[r]) atomicMin(&repS[r], pos);
                } else if (pos < __hip_atomic_load(&sp_rep[spo + r], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) {
                    atomicMin(&sp_rep[spo + r], pos);
                }
            }
            if (want_positions && !KEY4) {
                const uint64_t m = __ballot(kept);
                const uint32_t n = (uint32_t)__popcll(m);
                if (BASED) {
                    if (n && (chunk == kInvalid || cursor + n > kBi2Chunk)) {  // (wave-uniform) this row does not fit: close the chunk, take the next one of the pool
                        if (chunk != kInvalid && lane == 0) wcnt[chunk] = cursor;
                        uint32_t c = 0;
                        if (lane == 0) c = atomicAdd(&bs->nextchunk, 1u);
                        c      = (uint32_t)__builtin_amdgcn_readfirstlane((int)c);
                        chunk  = c < wcap ? c : kInvalid;
                        cursor = 0;
                        if (chunk == kInvalid) lost = true;
                    }
                    if (kept && chunk != kInvalid) wlist[(size_t)chunk * kBi2Chunk + cursor + (uint32_t)__popcll(m & ((1ull << lane) - 1ull))] = pos;
                    if (chunk != kInvalid) cursor += n;
                } else {
                    if (kept) {
                        const uint32_t at = cursor + (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
                        if (at < wcap) {
                            mylist[at] = pos;
                            if (mycode != nullptr) mycode[at] = fcode | r;
                        } else {
                            lost = true;
                        }
                    }
                    cursor += n;
                }
            }
        };
        {
            uint32_t cq[ROWS];  // all rows' entries are requested before the first is looked at (one LDS round trip per bin instead of one per row)
#pragma unroll
            for (int q = 0; q < ROWS; ++q) cq[q] = ((uint32_t)(q * kWave) < total && sl[q] != kInvalid) ? cnt_get(sl[q]) : 0u;
            const bool direct = total <= (uint32_t)(ROWS * kWave);
#pragma unroll
            for (int q = 0; q < ROWS; ++q)
                if ((uint32_t)(q * kWave) < total) settle(sl[q] != kInvalid, fixpos((uint32_t)(x[q] & pmask), (uint32_t)(q * kWave) + lane), cq[q], direct);
        }
        BI2_W(8);
        if (total > (uint32_t)(ROWS * kWave)) {
            unsigned long long y[4];
            auto               load4 = [&](uint32_t j0) {
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const uint32_t j = j0 + k * kWave + lane;
                    y[k]             = j < total ? load(j) : ~0ull;
                }
            };
            load4(ROWS * kWave);
            for (uint32_t j0 = ROWS * kWave; j0 < total; j0 += 4 * kWave) {
                unsigned long long z[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) z[k] = y[k];
                load4(j0 + 4 * kWave);
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    if ((uint32_t)(j0 + k * kWave) < total) {
                        const bool     valid = z[k] != ~0ull;
                        const uint32_t key   = (uint32_t)(z[k] >> pb) & 0x7FFFFFFFu;
                        settle(valid, fixpos((uint32_t)(z[k] & pmask), j0 + (uint32_t)(k * kWave) + lane), valid ? cnt_get(bi2_find(keyT, key, bi2_bucket_of(key, lgb, bmask), bmask)) : 0u, false);
                    }
                }
            }
        }
        BI2_W(9);
        for (uint32_t r = lane; r < min(ktotal, kReps); r += kWave) sp_rep[spo + r] = repS[r];
        BI2_W(10);
    };
    // the big bins first, one per wave, so that none of them starts when the others are about to finish; then the regular walk: bins are handed out four at a time
    // from 8 queues (queue q = the bins g with g mod 8 == q; a single counter would serialise ~12 ns per request): a wave that drew a big bin simply takes fewer of the
    // others. ONE call site of process_bin (round 5): inlined twice it made 60 KB of code against 64 KB of instruction cache per two CUs; now 29 KB.
    const uint32_t nbig = bs->nbig;
    const bool     skip_big = nbig <= (uint32_t)kBi2BigCap;  // (more big bins than the list holds: the regular walk takes them all)
    const uint32_t q = wid & (uint32_t)(kBi2Shards - 1);
    uint32_t       kb = skip_big ? wid : kInvalid, tk = 0, tleft = 0;
#pragma unroll 1
    for (;;) {
        uint32_t a, b;
        bool     bigp;
        if (kb < nbig) {
            const uint32_t f = bs->big[kb];
            kb += nwaves;
            a    = f / kBi2BBins;
            b    = f % kBi2BBins;
            bigp = true;
        } else {
            if (tleft == 0) {
                uint32_t t = 0;
                if (lane == 0) t = atomicAdd(&bs->nextbin[q * 16], 4u);
                t = (uint32_t)__builtin_amdgcn_readfirstlane((int)t);
                if (t * kBi2Shards + q >= nfinal) break;
                tk    = t;
                tleft = 4;
            }
            const uint32_t g = tk * kBi2Shards + q;
            ++tk;
            --tleft;
            if (g >= nfinal) continue;
            a    = g & (uint32_t)(kBins - 1);
            b    = g >> 8;
            bigp = false;
        }
        process_bin(a, b, bigp, skip_big);
    }
#ifdef BI2_PROF
    if (lane == 0)
        for (int q = 0; q < 12; ++q) atomicAdd(&bi2_prof[q], tacc[q]);
#endif
    if (BASED) {
        if (want_positions && chunk != kInvalid && lane == 0) wcnt[chunk] = cursor;
    } else if (want_positions && lane == 0) {
        wcnt[wid] = min(cursor, wcap);
    }
    if (__any(lost) && lane == 0) bs->overflow = 3;
}

// The huge final bins (more than kBi2HugeBin records: a hot key outside the dense head), one WORKGROUP each. A wave that owns such a bin alone streams it four
// rows at a time with four more in flight — 2 KB per memory round trip: the hottest bigram of a 10^9-token corpus (~76 000 windows) kept one wave busy for ~2 ms
// while the kernel's other 4095 waves needed 0.9 ms for everything else (measured: 16.5 ms of count kernels per step with the big bins, 7.0 ms without). Here
// eight waves share the bin's LDS table (same buckets, same compare-and-swap insert, same ranks) and keep 32 rows in flight. Runs BEFORE bi2_count_kernel
// (launched with big_elsewhere = true): the position lists of this kernel's waves (list g = block * 8 + wave, or chunks of the pool) are simply continued there.
constexpr int kBi2BigThreads = 512, kBi2BigRows = 8;  // (rows of 512 records a block keeps in flight)
// A big final bin is big because of ONE hot key, and a row of 64 records of that key is 64 atomics on one LDS word, executed one after the other. The lanes whose
// key equals the key of the row's first active lane stand back (act = false) and that lane counts for all of them (weight): one atomic per row for the hot key.
__device__ __forceinline__ void bi2_merge_leader(bool& act, uint32_t key, uint32_t& weight, uint32_t lane) {
    weight                 = 1u;
    const uint64_t actives = __ballot(act);
    if (actives == 0) return;
    const uint32_t lead = (uint32_t)__builtin_amdgcn_readlane((int)key, __builtin_amdgcn_readfirstlane(__ffsll((long long)actives) - 1));
    const bool     same = act && key == lead;
    const uint64_t m    = __ballot(same);
    const uint32_t n    = (uint32_t)__popcll(m);
    if (n >= 4u && same) {
        if (lane == (uint32_t)(__ffsll((long long)m) - 1))
            weight = n;
        else
            act = false;
    }
}
template <int NSUB, bool BASED = false, bool KEY4 = false, int SLOTS = kBi2Slots, bool PDROP = false>
__global__ __launch_bounds__(kBi2BigThreads) void bi2_count_big_kernel(const unsigned long long* __restrict__ recsB, uint32_t region, const uint32_t* __restrict__ boff, Bi2State* __restrict__ bs,
                                                                       DevState* __restrict__ st, uint32_t threshold, uint32_t* __restrict__ sp_rep, uint32_t* __restrict__ sp_cnt,
                                                                       uint32_t* __restrict__ wlist, uint32_t* __restrict__ wcnt, uint32_t wcap, bool want_positions,
                                                                       uint32_t* __restrict__ wcode, const uint32_t* __restrict__ slotbase, uint32_t pool_first = 0,
                                                                       uint32_t pool_n = 0) {
    if (st->done) return;
    const uint32_t nbig = bs->nhuge;
    if (nbig == 0 || nbig > (uint32_t)kBi2HugeCap) return;  // (more than the list holds: bi2_count_kernel walks every bin itself)
    const uint32_t* const keys4   = reinterpret_cast<const uint32_t*>(recsB);
    uint32_t* const       code_at = wlist;
    constexpr int                                    kW = kBi2BigThreads / kWave;
    __shared__ __attribute__((aligned(16))) uint32_t keyT[SLOTS];
    __shared__ __attribute__((aligned(16))) uint32_t cntT[SLOTS];
    __shared__ uint32_t                              repS[kBi2WReps];
    __shared__ uint32_t                              rsL[NSUB], rnL[NSUB], sbL[NSUB], wsumL[kW], failL;
    constexpr uint32_t kMaxLoad = kBi2MaxLoad * (uint32_t)(SLOTS / kBi2Slots);
    const uint32_t tid = threadIdx.x, lane = tid & (kWave - 1);
    const uint32_t pb = bs->posbits;
    const unsigned long long pmask = (1ull << pb) - 1;
    const uint32_t pdsh = PDROP ? bs->pdshift : 0u;
    // position lists: a wave of this kernel may list tens of thousands of windows of one bin, far beyond a fixed per-wave capacity — in both forms a wave takes lists
    // from a POOL (one atomic whenever its current list cannot hold a row). BASED: the chunks of bi2_count_kernel<.., BASED> (wlist = [wcap chunks][kBi2Chunk]).
    // Otherwise: lists of the wave kernel's own size (wcap entries) behind its kBi2Waves private ones: lists pool_first .. pool_first + pool_n - 1
    const uint32_t csize = BASED ? kBi2Chunk : wcap, pfirst = BASED ? 0u : pool_first, pn = BASED ? wcap : pool_n;
    uint32_t       cursor = 0, chunk = kInvalid;
    bool           lost   = false;
    constexpr int      lgb   = SLOTS == 2048 ? 9 : 8;  // SLOTS / 4 buckets: a big bin always takes the whole table
    constexpr uint32_t bmask = (1u << lgb) - 1u;
    static_assert((SLOTS == 1024 || SLOTS == 2048) && kBi2Slots == 1024 && kBi2BigBin + (kBi2BigBin >> 1) >= kBi2Slots, "big bins use all buckets, as they do in bi2_count_kernel");
    for (uint32_t k = blockIdx.x; k < nbig; k += gridDim.x) {
        const uint32_t f = bs->huge[k], a = f / kBi2BBins, b = f % kBi2BBins;
        __syncthreads();  // the bin before is done with the tables
        if (tid < (uint32_t)NSUB) {
            const uint32_t* bo = boff + (size_t)(tid * kBins + a) * (kBi2BBins + 1) + b;
            rsL[tid]           = bo[0];
            rnL[tid]           = bo[1] - bo[0];
            sbL[tid]           = BASED ? slotbase[tid * kBins + a] : 0u;
        }
        if (tid == 0) failL = 0;
        for (uint32_t s = tid; s < (uint32_t)SLOTS; s += kBi2BigThreads) {
            keyT[s] = kBi2Empty;
            cntT[s] = 0u;
        }
        __syncthreads();
        uint32_t rs[NSUB], rn[NSUB], sb[NSUB], total = 0;
#pragma unroll
        for (int s = 0; s < NSUB; ++s) {
            rs[s] = rsL[s];
            rn[s] = rnL[s];
            sb[s] = sbL[s];
            total += rn[s];
        }
        const uint32_t spo    = bs->binoff[a * kBi2BBins + b];
        auto           locate = [&](uint32_t j) -> size_t {
            uint32_t off = 0, slot = 0, sbase = 0;
            bool     ok  = false;
#pragma unroll
            for (int s = 0; s < NSUB; ++s) {
                if (!ok && j < rn[s]) {
                    ok    = true;
                    off   = rs[s] + j;
                    slot  = (uint32_t)s * kBins + a;
                    sbase = sb[s];
                }
                if (!ok) j -= rn[s];
            }
            return BASED ? (size_t)sbase + off : (size_t)slot * region + off;
        };
        unsigned long long y[kBi2BigRows];
        auto               load4 = [&](uint32_t j0) {
#pragma unroll
            for (int q = 0; q < kBi2BigRows; ++q) {
                const uint32_t j = j0 + q * kBi2BigThreads + tid;
                if (KEY4) {
                    const uint32_t idx = j < total ? (uint32_t)locate(j) : 0u;
                    y[q]               = j < total ? (((unsigned long long)(keys4[idx] & 0x7FFFFFFFu) << 31) | idx) : ~0ull;
                } else {
                    y[q] = j < total ? recsB[locate(j)] : ~0ull;
                }
            }
        };
        // pass 1: find or insert, count
        bool fail = false;
        for (uint32_t j0 = 0; j0 < total; j0 += kBi2BigRows * kBi2BigThreads) {
            load4(j0);
#pragma unroll
            for (int q = 0; q < kBi2BigRows; q += 2) {
                bool           actA = y[q] != ~0ull, actB = y[q + 1] != ~0ull;
                const uint32_t keyA = (uint32_t)(y[q] >> pb) & 0x7FFFFFFFu, keyB = (uint32_t)(y[q + 1] >> pb) & 0x7FFFFFFFu;
                uint32_t       wA, wB;
                bi2_merge_leader(actA, keyA, wA, lane);
                bi2_merge_leader(actB, keyB, wB, lane);
                const uint32_t bkA = bi2_bucket_of(keyA, lgb, bmask), bkB = bi2_bucket_of(keyB, lgb, bmask);
                uint32_t       tA, tB;
                uint32_t       oA, oB;
                bi2_insert2(keyT, bmask, actA, keyA, bkA, *reinterpret_cast<const uint4*>(keyT + bkA * 4), tA, actB, keyB, bkB, *reinterpret_cast<const uint4*>(keyT + bkB * 4), tB, oA, oB);
                if (actA) {
                    if (tA == kInvalid)
                        fail = true;
                    else
                        atomicAdd(&cntT[tA], wA);
                }
                if (actB) {
                    if (tB == kInvalid)
                        fail = true;
                    else
                        atomicAdd(&cntT[tB], wB);
                }
            }
        }
        if (fail) failL = 1;
        __syncthreads();
        if (failL) {
            if (tid == 0) bs->overflow = 2;
            continue;
        }
        // survivors: thread t looks at the two slots from 2 t — ranks in slot order, as in bi2_count_kernel
        constexpr uint32_t per = SLOTS / kBi2BigThreads;  // (two or four slots per thread)
        const uint32_t s0 = tid * per;
        uint32_t       cS[per], nk = 0, nu = 0;
#pragma unroll
        for (uint32_t i = 0; i < per; ++i) {
            cS[i] = cntT[s0 + i];
            nk += cS[i] >= threshold;
            nu += keyT[s0 + i] != kBi2Empty;
        }
        uint32_t       distinct, ktotal;
        const uint32_t excl = bi2_block_scan<kBi2BigThreads>(nk, &ktotal, wsumL);
        bi2_block_scan<kBi2BigThreads>(nu, &distinct, wsumL);
        if (distinct > kMaxLoad || ktotal > 1023u) {
            if (tid == 0) bs->overflow = 2;
            continue;
        }
        if (tid == 0) {
            atomicAdd(&bs->found_part[a], distinct);
            bs->binkept[a * kBi2BBins + b] = ktotal;
            if (ktotal) atomicAdd(&bs->kept_part[a], ktotal);
        }
        if (ktotal == 0) {
            if (KEY4)
                for (uint32_t j = tid; j < total; j += kBi2BigThreads) code_at[locate(j)] = kInvalid;
            continue;
        }
        const bool reps_lds = ktotal <= (uint32_t)kBi2WReps;
        {
            uint32_t r = excl;
#pragma unroll
            for (uint32_t i = 0; i < per; ++i) {
                if (cS[i] >= threshold) {
                    sp_cnt[spo + r] = cS[i];
                    cntT[s0 + i]    = kBi2Kept | r;
                    if (reps_lds)
                        repS[r] = 0xFFFFFFFFu;
                    else
                        sp_rep[spo + r] = 0xFFFFFFFFu;
                    ++r;
                }
            }
        }
        if (!reps_lds) __threadfence();
        __syncthreads();
        // pass 2: every window of a surviving key: lowest position of the key; the position joins the wave's list
        const uint32_t fcode = (a * (uint32_t)kBi2BBins + b) << 10;
        for (uint32_t j0 = 0; j0 < total; j0 += kBi2BigRows * kBi2BigThreads) {
            load4(j0);
#pragma unroll
            for (int q = 0; q < kBi2BigRows; ++q) {
                if (j0 + q * kBi2BigThreads + (tid & ~(uint32_t)(kWave - 1)) >= total) continue;  // (wave-uniform)
                const bool     valid = y[q] != ~0ull;
                const uint32_t key = (uint32_t)(y[q] >> pb) & 0x7FFFFFFFu;
                uint32_t       pos = (uint32_t)(y[q] & pmask);
                if (PDROP && pdsh != 0u) {  // (see bi2_count_kernel: the run the record lies in names the three bits its position lacks)
                    uint32_t jj = j0 + q * kBi2BigThreads + tid, sub = 0;
#pragma unroll
                    for (int s2 = 0; s2 < NSUB - 1; ++s2) {
                        const bool beyond = jj >= rn[s2] && sub == (uint32_t)s2;
                        jj -= beyond ? rn[s2] : 0u;
                        sub += beyond ? 1u : 0u;
                    }
                    const uint32_t sh = pdsh - 1u;
                    pos               = ((pos >> sh) << (sh + 3u)) | (sub << sh) | (pos & ((1u << sh) - 1u));
                }
                uint32_t       c = 0;
                if (valid) c = cntT[bi2_find(keyT, key, bi2_bucket_of(key, lgb, bmask), bmask)];
                const bool     kept = (c & kBi2Kept) != 0;
                const uint32_t r    = c & ~kBi2Kept;
                if (KEY4 && valid) code_at[pos] = kept ? (fcode | r) : kInvalid;
                if (kept) {
                    if (reps_lds) {
                        if (pos < repS[r]) atomicMin(&repS[r], pos);
                    } else if (pos < __hip_atomic_load(&sp_rep[spo + r], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) {
                        atomicMin(&sp_rep[spo + r], pos);
                    }
                }
                if (want_positions && !KEY4) {
                    const uint64_t m = __ballot(kept);
                    const uint32_t n = (uint32_t)__popcll(m);
                    if (n && (chunk == kInvalid || cursor + n > csize)) {  // (wave-uniform) this row does not fit: close the list, take the next one of the pool
                        if (chunk != kInvalid && lane == 0) wcnt[chunk] = cursor;
                        uint32_t cc = 0;
                        if (lane == 0) cc = atomicAdd(&bs->nextchunk, 1u);
                        cc     = (uint32_t)__builtin_amdgcn_readfirstlane((int)cc);
                        chunk  = cc < pn ? pfirst + cc : kInvalid;
                        cursor = 0;
                        if (chunk == kInvalid) lost = true;
                    }
                    if (kept && chunk != kInvalid) {
                        const size_t at = (size_t)chunk * csize + cursor + (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
                        wlist[at]       = pos;
                        if (wcode != nullptr) wcode[at] = fcode | r;
                    }
                    if (chunk != kInvalid) cursor += n;
                }
            }
        }
        __syncthreads();
        if (reps_lds)
            for (uint32_t r = tid; r < ktotal; r += kBi2BigThreads) sp_rep[spo + r] = repS[r];
    }
    if (want_positions && chunk != kInvalid && lane == 0) wcnt[chunk] = cursor;
    if (__any(lost) && lane == 0) bs->overflow = 3;
}

// the list pool's cursor across the passes of a sliced order (Bi2State is zeroed per pass)
__global__ void bi2_chunk_cursor_kernel(Bi2State* __restrict__ bs, uint32_t* __restrict__ keep, bool restore) {
    if (restore)
        bs->nextchunk = *keep;
    else
        *keep = bs->nextchunk;
}
// per-bin survivor counts -> dense result offsets, bin by bin; block a scans A bin a
__global__ __launch_bounds__(kBi2BBins) void bi2_kept_scan_kernel(Bi2State* __restrict__ bs, const DevState* __restrict__ st) {
    if (st->done) return;
    __shared__ uint32_t inL[kBi2BBins], outL[kBi2BBins], wsumL[8], beforeL;
    const uint32_t      a = blockIdx.x;
    inL[threadIdx.x]      = threadIdx.x < a ? bs->kept_part[threadIdx.x] : 0u;  // (kBins <= kBi2BBins: entries beyond 255 are zero)
    __syncthreads();
    const uint32_t before = bi2_scan512(inL, outL, wsumL);
    if (threadIdx.x == 0) beforeL = before;
    __syncthreads();
    inL[threadIdx.x] = bs->binkept[a * kBi2BBins + threadIdx.x];
    __syncthreads();
    const uint32_t tot                        = bi2_scan512(inL, outL, wsumL);
    bs->binkept[a * kBi2BBins + threadIdx.x] = beforeL + outL[threadIdx.x];
    if (a == kBins - 1 && threadIdx.x == 0) bs->kept_bins = beforeL + tot;
}
// head bigrams -> survivors; found / kept of the order
__global__ __launch_bounds__(kBlock) void bi2_finish_kernel(DevState* __restrict__ st, Bi2State* __restrict__ bs, uint32_t threshold, uint32_t res_cap,
                                                             uint32_t* __restrict__ headsurv_keep /* first pass: the head survivor bits, kept for the list kernel; else NULL */,
                                                             uint32_t ovf_code = 4 /* what st->radix_overflow becomes when this order could not be held (chain.hpp's orders: 5) */) {
    if (st->done) return;
    uint32_t htot, ftot, hftot;
    block_exclusive_scan(bs->found_part[threadIdx.x], &ftot);
    static_assert(kBi2HeadN == kBlock * 16, "16 head keys per lane");
    uint32_t hk = 0, hf = 0, bits = 0;
#pragma unroll
    for (int q = 0; q < 16; ++q) {
        const uint32_t c = bs->headcnt[threadIdx.x * 16 + q];
        hf += c != 0;
        if (c >= threshold) {
            ++hk;
            bits |= 1u << q;
        }
    }
    reinterpret_cast<uint16_t*>(bs->headsurv)[threadIdx.x] = (uint16_t)bits;
    if (headsurv_keep != nullptr) reinterpret_cast<uint16_t*>(headsurv_keep)[threadIdx.x] = (uint16_t)bits;
    const uint32_t ho = block_exclusive_scan(hk, &htot);
    bs->headbase[threadIdx.x] = ho;
    block_exclusive_scan(hf, &hftot);
    if (threadIdx.x == 0) {
        const uint32_t tot = bs->kept_bins;
        bs->kept_head      = htot;
        bs->ran            = 1;
        bs->res_base       = st->res_total + st->kept;  // (st->found / st->kept are zero at the start of an order: the passes of a sliced order add up)
        st->found += ftot + hftot;
        st->kept += tot + htot;
        if ((uint64_t)bs->res_base + tot + htot > res_cap) st->overflow = 1;
        if (bs->overflow && st->radix_overflow != 4) st->radix_overflow = ovf_code;  // the host re-runs on the first-generation kernels
        const uint64_t next = (uint64_t)st->id_base + bs->nrec;  // keeps the id space of the later orders disjoint, as bin_advance_prepare_kernel does
        if (next >= 0xFFFFFFF0ull) st->radix_overflow = 3;
        st->id_base = (uint32_t)next;
    }
}
// bi2_kept_scan_kernel and bi2_finish_kernel in ONE launch (round 5: a launch of a few microseconds of work still costs ~5 us of the step, and every order and every
// skipgram pass had both): blocks 0 .. kBins - 1 scan their A bin, block kBins does the order's bookkeeping — it needs the per-A-bin totals only (found_part,
// kept_part: complete when the count kernels are), not the scans. grid kBins + 1, kBi2BBins threads.
__global__ __launch_bounds__(kBi2BBins) void bi2_kept_finish_kernel(DevState* __restrict__ st, Bi2State* __restrict__ bs, uint32_t threshold, uint32_t res_cap,
                                                                     uint32_t* __restrict__ headsurv_keep, uint32_t ovf_code) {
    if (st->done) return;
    __shared__ uint32_t inL[kBi2BBins], outL[kBi2BBins], wsumL[8], beforeL;
    if (blockIdx.x < (uint32_t)kBins) {
        const uint32_t a = blockIdx.x;
        inL[threadIdx.x] = threadIdx.x < a ? bs->kept_part[threadIdx.x] : 0u;  // (kBins <= kBi2BBins: entries beyond 255 are zero)
        __syncthreads();
        const uint32_t before = bi2_scan512(inL, outL, wsumL);
        if (threadIdx.x == 0) beforeL = before;
        __syncthreads();
        inL[threadIdx.x] = bs->binkept[a * kBi2BBins + threadIdx.x];
        __syncthreads();
        const uint32_t tot                        = bi2_scan512(inL, outL, wsumL);
        bs->binkept[a * kBi2BBins + threadIdx.x] = beforeL + outL[threadIdx.x];
        if (a == kBins - 1 && threadIdx.x == 0) bs->kept_bins = beforeL + tot;
        return;
    }
    static_assert(kBi2HeadN == kBlock * 16 && kBins == kBlock && kBi2BBins >= kBlock, "16 head keys per lane of the first 256");
    const bool low = threadIdx.x < (uint32_t)kBlock;
    uint32_t   htot, ftot, hftot, tot;
    bi2_block_scan<kBi2BBins>(low ? bs->found_part[threadIdx.x] : 0u, &ftot, wsumL);
    bi2_block_scan<kBi2BBins>(low ? bs->kept_part[threadIdx.x] : 0u, &tot, wsumL);
    uint32_t hk = 0, hf = 0, bits = 0;
    if (low) {
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            const uint32_t c = bs->headcnt[threadIdx.x * 16 + q];
            hf += c != 0;
            if (c >= threshold) {
                ++hk;
                bits |= 1u << q;
            }
        }
        reinterpret_cast<uint16_t*>(bs->headsurv)[threadIdx.x] = (uint16_t)bits;
        if (headsurv_keep != nullptr) reinterpret_cast<uint16_t*>(headsurv_keep)[threadIdx.x] = (uint16_t)bits;
    }
    const uint32_t ho = bi2_block_scan<kBi2BBins>(hk, &htot, wsumL);
    if (low) bs->headbase[threadIdx.x] = ho;
    bi2_block_scan<kBi2BBins>(hf, &hftot, wsumL);
    if (threadIdx.x == 0) {
        bs->kept_head = htot;
        bs->ran       = 1;
        bs->res_base  = st->res_total + st->kept;  // (st->found / st->kept are zero at the start of an order: the passes of a sliced order add up)
        st->found += ftot + hftot;
        st->kept += tot + htot;
        if ((uint64_t)bs->res_base + tot + htot > res_cap) st->overflow = 1;
        if (bs->overflow && st->pad[0] == 0) st->pad[0] = bs->overflow | (bs->kbits << 8) | (bs->posbits << 16) | (bs->bshift << 24);  // (COLIBRI_DEBUG_OVERFLOW: what gave up first)
        if (bs->overflow && st->radix_overflow != 4) st->radix_overflow = ovf_code;  // the host re-runs on the first-generation kernels
        const uint64_t next = (uint64_t)st->id_base + bs->nrec;  // keeps the id space of the later orders disjoint, as bin_advance_prepare_kernel does
        if (next >= 0xFFFFFFF0ull) st->radix_overflow = 3;
        st->id_base = (uint32_t)next;
    }
}
// sparse per-bin survivors -> dense result list: one wave copies one bin's run; the last block appends the head survivors
__global__ __launch_bounds__(kBlock) void bi2_compact_kernel(const uint32_t* __restrict__ sp_rep, const uint32_t* __restrict__ sp_cnt, const DevState* __restrict__ st,
                                                              const Bi2State* __restrict__ bs, uint32_t* __restrict__ res_rep, uint32_t* __restrict__ res_cnt, uint32_t res_cap,
                                                              bool beside = false /* on a second stream, behind the order's finish kernel: the run's state may already say "done"
                                                                                     because THIS order ended it — the order's own flag decides */) {
    if (beside ? !bs->ran : st->done != 0) return;
    const uint32_t res_base = bs->res_base, lane = threadIdx.x & (kWave - 1);
    if (blockIdx.x + 1 < gridDim.x) {
        const uint32_t nwaves = (gridDim.x - 1) * (kBlock / kWave);
        for (uint32_t g = blockIdx.x * (kBlock / kWave) + threadIdx.x / kWave; g < (uint32_t)kBi2Final; g += nwaves) {
            const uint32_t f   = ((g & (uint32_t)(kBins - 1)) * kBi2BBins) + (g >> 8);  // a fastest: the non-empty bins of a small order spread over all waves
            const uint32_t off = bs->binkept[f];
            const uint32_t n   = ((f + 1 < (uint32_t)kBi2Final) ? bs->binkept[f + 1] : bs->kept_bins) - off;
            if (n == 0) continue;
            const uint32_t src = bs->binoff[f];
            for (uint32_t j = lane; j < n; j += kWave) {
                const uint32_t r = res_base + off + j;
                if (r < res_cap) {
                    res_rep[r] = sp_rep[src + j];
                    res_cnt[r] = sp_cnt[src + j];
                }
            }
        }
    } else {
        uint32_t       r    = res_base + bs->kept_bins + bs->headbase[threadIdx.x];
        const uint32_t bits = reinterpret_cast<const uint16_t*>(bs->headsurv)[threadIdx.x];
#pragma unroll
        for (int q = 0; q < 16; ++q)
            if (bits & (1u << q)) {
                if (r < res_cap) {
                    res_rep[r] = ~bs->headposinv[threadIdx.x * 16 + q];
                    res_cnt[r] = bs->headcnt[threadIdx.x * 16 + q];
                }
                ++r;
            }
    }
}

// ---- positions: the waves' unsorted lists -> one list per (shard, position bucket) (Bi2Lists: above) ------------------------------------------
// tile-local counting sort by bucket in LDS, one reserved run per (tile, bucket). The LDS of one tile:
template <int PER>
struct Bi2PospartLdsT {
    uint32_t stgL[kBi2Threads * PER], stgC[kBi2Threads * PER];
    uint16_t binL[kBi2Threads * PER];
    uint32_t histL[kBi2Buckets], offL[kBi2Buckets], gbaseL[kBi2Buckets], wsumL[kBi2Threads / kWave];
};
using Bi2PospartLds = Bi2PospartLdsT<kBi2Per>;
#ifndef COLIBRI_PP_PER
#define COLIBRI_PP_PER 4
#endif
constexpr int kBi2PpPer = COLIBRI_PP_PER;  // entries per lane and tile of bi2_pospart_kernel
// one tile: every lane brings up to kBi2Per (position, code) entries (position 0xFFFFFFFF: none); every thread of the block calls it
template <int PER>
__device__ __forceinline__ void bi2_pospart_tile(Bi2PospartLdsT<PER>& L, const uint32_t (&p)[PER], const uint32_t (&code)[PER], uint32_t shard, Bi2State* __restrict__ bs,
                                                 DevState* __restrict__ st, uint32_t* __restrict__ plist, Bi2Lists pl, uint32_t* __restrict__ pcode) {
    static_assert(kBi2Buckets == kBi2Threads, "one lane per position bucket");
    uint32_t rank[PER];
    KP_INIT(3);  // (per call: what lies between two calls — the loads of the next tile — is the kernel's time minus these sections)
    L.histL[threadIdx.x] = 0;
    __syncthreads();
    KP(0);
#pragma unroll
    for (int k = 0; k < PER; ++k) {
        rank[k] = kInvalid;
        if (p[k] != 0xFFFFFFFFu) {
            const uint32_t b = p[k] >> pl.pshift;
            rank[k]          = atomicAdd(&L.histL[b], 1u) | (b << 16);
        }
    }
    __syncthreads();
    KP(1);
    uint32_t tot;
    L.offL[threadIdx.x] = bi2_block_scan<kBi2Threads>(L.histL[threadIdx.x], &tot, L.wsumL);
    __syncthreads();  // every bucket's offset is written
    KP(2);
    // (the reservation's answer is first needed by the copy-out: the memory-side atomic travels while the tile is staged — round 6)
    const uint32_t rs_h = L.histL[threadIdx.x], rs_l = shard * kBi2Buckets + threadIdx.x;
    uint32_t       rs_at = 0;
    if (rs_h) rs_at = atomicAdd(&bs->pcur[bi2_pc(rs_l)], rs_h);
#pragma unroll
    for (int k = 0; k < PER; ++k) {
        if (rank[k] != kInvalid) {
            const uint32_t b = rank[k] >> 16, q = L.offL[b] + (rank[k] & 0xFFFFu);
            L.stgL[q]        = p[k];
            L.stgC[q]        = code[k];
            L.binL[q]        = (uint16_t)b;
        }
    }
    {
        uint32_t g = 0;
        if (rs_h) {
            if (rs_at + rs_h > pl.pcap) st->radix_overflow = 4;  // (the order's finish kernel has run: the flag goes straight to the run's state)
            g = rs_l * pl.pcap + min(rs_at, pl.pcap - min(pl.pcap, rs_h));  // (fits 32 bits: shards * buckets * pcap <= 8 * positions)
        }
        L.gbaseL[threadIdx.x] = g;
    }
    __syncthreads();
    KP(3);
    for (uint32_t j = threadIdx.x; j < tot; j += kBi2Threads) {
        const uint32_t b                           = L.binL[j];
        plist[(size_t)L.gbaseL[b] + (j - L.offL[b])] = L.stgL[j];
        if (pcode != nullptr) pcode[(size_t)L.gbaseL[b] + (j - L.offL[b])] = L.stgC[j];
    }
    __syncthreads();
    KP(4);
    KP_DONE();
}
// the waves' lists: block x takes the lists x, x + gridDim.x, ...
__global__ __launch_bounds__(kBi2Threads, kBi2PpPer <= 4 ? kBi2Threads / 128 : 4) void bi2_pospart_kernel(const uint32_t* __restrict__ wlist, const uint32_t* __restrict__ wcnt, uint32_t nlists, uint32_t wcap,
                                                                                      Bi2State* __restrict__ bs, DevState* __restrict__ st, uint32_t* __restrict__ plist, Bi2Lists pl,
                                                                                      const uint32_t* __restrict__ wcode = nullptr, uint32_t* __restrict__ pcode = nullptr,
                                                                                      uint32_t flat_n = 0, bool dense = false) {
    // wcode / pcode (optional): the (bin, rank) codes travel with their positions
    // dense (chain.hpp; after bi2_kept_scan_kernel): the codes leave as dense survivor numbers (per-bin offset + rank). A wave's list holds long runs of one bin's
    // windows, so the offset look-up is nearly a broadcast here — after the partition it would be a random gather per entry
    // flat_n (key-sharded runs: the positions the owners sent back, one array): wlist holds flat_n entries, cut into nlists pieces of wcap; wcnt is not read
    if (st->done) return;
    __shared__ Bi2PospartLdsT<kBi2PpPer> L;
    const uint32_t           shard = blockIdx.x & (uint32_t)(kBi2Shards - 1);
    for (uint32_t w = blockIdx.x; w < nlists; w += gridDim.x) {
        const uint32_t        n   = flat_n ? min(wcap, flat_n - min(flat_n, w * wcap)) : min(wcnt[w], wcap);
        const uint32_t* const src  = wlist + (size_t)w * wcap;
        const uint32_t* const csrc = wcode != nullptr ? wcode + (size_t)w * wcap : nullptr;
        for (uint32_t j0 = 0; j0 < n; j0 += kBi2Threads * kBi2PpPer) {
            uint32_t p[kBi2PpPer], code[kBi2PpPer];
#pragma unroll
            for (int k = 0; k < kBi2PpPer; ++k) {
                const uint32_t j = j0 + k * kBi2Threads + threadIdx.x;
                p[k]             = j < n ? src[j] : 0xFFFFFFFFu;
                code[k]          = (csrc != nullptr && j < n) ? csrc[j] : 0u;
            }
            if (dense) {
#pragma unroll
                for (int k = 0; k < kBi2PpPer; ++k)
                    if (p[k] != 0xFFFFFFFFu) code[k] = bs->binkept[code[k] >> 10] + (code[k] & 1023u);
            }
            bi2_pospart_tile(L, p, code, shard, bs, st, plist, pl, pcode);
        }
    }
}

// ---- bitmap: per position bucket, the listed positions -> one bit each (built in LDS, written as whole words) -----------------------------
__global__ __launch_bounds__(kBi2BmThreads) void bi2_bitmap_kernel(uint32_t npos, const Bi2State* __restrict__ bs, const uint32_t* __restrict__ plist, Bi2Lists pl,
                                                                    const DevState* __restrict__ st, uint32_t* __restrict__ bitmap) {
    if (st->done) return;
    extern __shared__ uint32_t bmL[];  // (1 << pshift) / 32 words
    const uint32_t b = blockIdx.x, start = b << pl.pshift;
    if (start >= npos) return;
    const uint32_t size   = min(1u << pl.pshift, npos - start);
    const uint32_t nwords = (size + 31) / 32;
    for (uint32_t w = threadIdx.x; w < nwords; w += kBi2BmThreads) bmL[w] = 0;
    __syncthreads();
    for (uint32_t x = 0; x < (uint32_t)kBi2Shards; ++x) {
        const uint32_t     l = x * kBi2Buckets + b, n = min(bs->pcur[bi2_pc(l)], pl.pcap);
        const uint32_t*    p = plist + (size_t)l * pl.pcap;  // 16-byte aligned: pcap is a multiple of 4
        const uint4* const v = reinterpret_cast<const uint4*>(p);
        const uint32_t     nv = n >> 2;
        for (uint32_t j = threadIdx.x; j < nv; j += kBi2BmThreads) {
            const uint4    e  = v[j];
            const uint32_t o0 = e.x - start, o1 = e.y - start, o2 = e.z - start, o3 = e.w - start;
            atomicOr(&bmL[o0 >> 5], 1u << (o0 & 31u));
            atomicOr(&bmL[o1 >> 5], 1u << (o1 & 31u));
            atomicOr(&bmL[o2 >> 5], 1u << (o2 & 31u));
            atomicOr(&bmL[o3 >> 5], 1u << (o3 & 31u));
        }
        if (threadIdx.x < (n & 3u)) {
            const uint32_t o = p[(nv << 2) + threadIdx.x] - start;
            atomicOr(&bmL[o >> 5], 1u << (o & 31u));
        }
    }
    __syncthreads();
    for (uint32_t w = threadIdx.x; w < nwords; w += kBi2BmThreads) bitmap[(start >> 5) + w] = bmL[w];
}

// ---- result indices per position (the modes that keep every order's ids: forward index, skipgram passes): chain.hpp's chain_ids_kernel (rounds 2-3 scattered a bucket's ids
// with one block per bucket — bi2_ids_kernel, 1.33 GB written for 0.42 GB of ids). The head bigrams' result indices: -----------------------------------------------------------
__global__ __launch_bounds__(kBlock) void bi2_headids_kernel(const Bi2State* __restrict__ bs, const DevState* __restrict__ st, uint32_t* __restrict__ headid /* [kBi2HeadN] */) {
    if (st->done) return;
    uint32_t       r    = bs->res_base + bs->kept_bins + bs->headbase[threadIdx.x];  // as bi2_compact_kernel numbers them
    const uint32_t bits = reinterpret_cast<const uint16_t*>(bs->headsurv)[threadIdx.x];
#pragma unroll
    for (int q = 0; q < 16; ++q) headid[threadIdx.x * 16 + q] = (bits & (1u << q)) ? r++ : kInvalid;
}

// ---- list3: one streaming pass over the corpus: bitmap | head survivors -> active list of order 3 ----------------------------------------
// entry i of the list: bigrams at i and at i + 1 both survived. st->valid += positions with a surviving bigram.
// cls must be readable (zeros) up to index npos + 63 rounded up to a multiple of 32; bitmap up to word npos / 32 + 1 (zeros beyond the corpus).
constexpr int kBi2L3Tile = kBlock * 32;  // one bitmap word per lane
__global__ __launch_bounds__(kBlock) void bi2_list3_kernel(const uint32_t* __restrict__ cls, const uint32_t* __restrict__ surv, uint32_t npos, const uint32_t* __restrict__ headsurv,
                                                            const uint32_t* __restrict__ bitmap, DevState* __restrict__ st, uint32_t* __restrict__ list_out,
                                                            uint32_t* __restrict__ nlist_out, uint32_t* __restrict__ ids = nullptr /* optional: result index per position */,
                                                            const uint32_t* __restrict__ headid = nullptr /* ... of the head bigrams (bi2_headids_kernel) */) {
    if (st->done) return;
    __shared__ uint32_t hsL[kBi2HeadN / 32], stageL[kBi2L3Tile], baseL, redL[kBlock / kWave];
    if (threadIdx.x < kBi2HeadN / 32) hsL[threadIdx.x] = headsurv[threadIdx.x];
    __syncthreads();
    const uint32_t sw0 = surv[0], sw1 = surv[1];
    const uint32_t ntiles = (npos + kBi2L3Tile - 1) / kBi2L3Tile;
    uint32_t       nvalid = 0;
    for (uint32_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const uint32_t w  = tile * kBlock + threadIdx.x;  // this lane's bitmap word = positions [32 w, 32 w + 32)
        const uint32_t p0 = w * 32;
        uint32_t       x = 0, nx = 0;
        uint32_t       c[36];
        if (p0 < npos) {
            x  = bitmap[w];
            nx = bitmap[w + 1];
            const uint4* const v = reinterpret_cast<const uint4*>(cls + p0);
#pragma unroll
            for (int k = 0; k < 9; ++k) {
                const uint4 e = v[k];
                c[4 * k]     = e.x;
                c[4 * k + 1] = e.y;
                c[4 * k + 2] = e.z;
                c[4 * k + 3] = e.w;
            }
            // head bigrams never became records: their windows are evaluated here, against the head survivor bits
            uint32_t hb = 0, hnx = 0;
#pragma unroll
            for (int k = 0; k < 33; ++k) {
                const uint32_t c0 = c[k], c1 = c[k + 1];
                if (c0 - 1u < (uint32_t)(kBi2Head - 1) && c1 - 1u < (uint32_t)(kBi2Head - 1)) {
                    const uint32_t s0 = (c0 < 32 ? sw0 >> c0 : sw1 >> (c0 - 32)) & 1u, s1 = (c1 < 32 ? sw0 >> c1 : sw1 >> (c1 - 32)) & 1u;
                    const uint32_t h  = c0 * kBi2Head + c1;
                    const uint32_t on = s0 & s1 & (hsL[h >> 5] >> (h & 31u));
                    if (k < 32) {
                        hb |= on << k;
                        if (ids != nullptr && on && p0 + (uint32_t)k < npos) ids[p0 + (uint32_t)k] = headid[h];  // (a 16 KB table: cache-resident)
                    } else {
                        hnx = on;
                    }
                }
            }
            x |= hb;
            nx |= hnx;
        }
        const uint32_t y = x & ((x >> 1) | (nx << 31));
        const uint32_t n = __popc(y);
        nvalid += __popc(x);
        uint32_t       total;
        const uint32_t excl = block_exclusive_scan(n, &total);
        if (threadIdx.x == 0) baseL = total ? atomicAdd(nlist_out, total) : 0u;
        uint32_t o = excl, yy = y;
        while (yy) {
            const int bit = __builtin_ctz(yy);
            yy &= yy - 1;
            stageL[o++] = p0 + (uint32_t)bit;
        }
        __syncthreads();
        const uint32_t gb = baseL;
        for (uint32_t j = threadIdx.x; j < total; j += kBlock) list_out[gb + j] = stageL[j];
        __syncthreads();
    }
    for (int off = 32; off > 0; off >>= 1) nvalid += __shfl_down(nvalid, off, kWave);
    if ((threadIdx.x & (kWave - 1)) == 0) redL[threadIdx.x / kWave] = nvalid;
    __syncthreads();
    if (threadIdx.x == 0) {
        const uint32_t v = redL[0] + redL[1] + redL[2] + redL[3];
        if (v) atomicAdd(&st->valid, v);
    }
}

// order 3 over the list bi2_list3_kernel leaves: every listed window is admissible (both bigrams survived), the key is its three class ids
struct KeyTrigramClsListed {
    const uint32_t* cls;
    __device__ __forceinline__ bool operator()(uint32_t i, uint32_t npos, uint64_t& key, uint64_t& hash) const {
        if (i + 2 >= npos) return false;
        const uint32_t c0 = cls[i], c1 = cls[i + 1], c2 = cls[i + 2];
        key  = (uint64_t)c0 | ((uint64_t)c1 << 21) | ((uint64_t)c2 << 42);
        hash = mix64(key);
        return true;
    }
};

}  // namespace colibri
