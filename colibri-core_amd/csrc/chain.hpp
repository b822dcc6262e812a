// chain.hpp — orders >= 3 of the plain run on the second-generation engine (gfx950, wave64).
//
// The window loop of PatternModel::train (reference include/patternmodel.h:1078-1178) admits window i of order n iff the (n-1)-grams at i and at i + 1 both survived the
// prune of order n - 1 (look-back :1139-1152). Rounds 1-3 ran orders >= 3 on the first-generation kernels (binned.hpp): 16-byte records {64-bit key, item, tile count,
// hash bits}, a representative election per tile, a workgroup per final bin, and a 4-byte survivor id scattered to every record position and gathered again. This file
// runs them on the order-2 engine of bigram2.hpp instead:
//   * the EXACT identity of an n-gram is (dense survivor number of the (n-1)-gram at i, class id of token i + n - 1) — idbits + clsbits <= 48 bits — and travels as its
//     bijective mix in an 8-byte record next to the position, like an order-2 record; level B, the wave-per-bin count, the compaction are bigram2.hpp's kernels unchanged;
//   * no per-position id array exists: the count kernel of order n - 1 lists (position, (bin, rank) code) of every surviving window, bi2_pospart_kernel sorts the pairs
//     into position buckets, chain_bitmap_kernel turns the positions into one bit each, and chain_emit_kernel walks the PAIRS: pair (i, code) becomes a record of order n
//     iff bit i + 1 is set. The code names the (n-1)-gram (final bin, rank among the bin's survivors -> dense number through the per-bin offsets the order's scan left);
//     the class id at i + n - 1 is one gather inside the bucket's window. Nothing is scattered by position and nothing but list entries is read;
//   * order 2's dense head (class pairs below 64 x 64, counted in LDS, never records) joins through one streaming pass over the class ids that appends the windows of
//     surviving head pairs to the lists as ordinary pairs (chain_head_pairs_kernel): nothing downstream knows about the head.
// add / prune semantics (reference :2059-2073, :2107-2128) are the count kernel's: exact keys, threshold on the exact count, lowest position as representative.
#pragma once
#include "bigram2.hpp"

namespace colibri {

#ifndef COLIBRI_CH_THREADS
#define COLIBRI_CH_THREADS 512
#endif
constexpr int kChThreads = COLIBRI_CH_THREADS;    // emit blocks: several per CU — a step is a chain of dependent round trips (pairs, gathers, a reservation), other blocks fill them
constexpr int kChPer     = 4;                     // candidates per lane and step
constexpr int kChStep    = kChThreads * kChPer;   // candidates per step
constexpr int kChQueue   = 2 * kChStep;           // records waiting for a partition step (a step starts with fewer than kChStep of them)
constexpr int kChQPer    = kChQueue / kChThreads;

#ifndef COLIBRI_CH_HUGE
#define COLIBRI_CH_HUGE 4096
#endif
// Orders >= 3 have no dense head: the hottest n-grams (tens of thousands of windows of one key at 10^8 tokens) are ordinary records, one final bin each. A bin beyond this
// many records goes to the workgroup kernel (order 2: 16 384): a single wave streams 4 rows per round trip, and the hottest bin below the limit is the count kernel's tail.
constexpr uint32_t kChHugeBin = COLIBRI_CH_HUGE;
constexpr uint32_t kChLists = kBi2Shards + 1;  // lists per bucket: the eight shards bi2_pospart_kernel fills, and (order 2) the head windows' list

// ---- order 2's head windows are pairs like all others ---------------------------------------------------------------------------------------------------------------
// The 64 x 64 most frequent class pairs are counted in LDS and never become records, so the count kernel lists none of their windows. bi2_emit_kernel, which sees every
// window anyway, appends (position, kBi2HeadCode | pair) for every head window to list (shard 8, bucket) — room for every position of the bucket —; whether the pair
// survived is known after the count: bi2_headids_kernel's table gives a surviving pair its result index, kInvalid otherwise, and the two readers of the lists look it up
// (16 KB, cache-resident). Rounds 2-3 evaluated the head again, from the class ids, in every kernel that needed "did the bigram at i survive".
__device__ __forceinline__ bool chain_head_alive(uint32_t code, const uint32_t* __restrict__ headid) { return headid[code & 0xFFFu] != kInvalid; }

// ---- one launch clears what an order starts from: its Bi2State and the position lists' counts (two fills of ~5 us each before) ----------------------------------------
// thread t of `step` threads clears its share of an order's state and of the lists' counts (chain_reset_kernel, and chain_begin_kernel's clearing blocks)
__device__ __forceinline__ void chain_clear_state(Bi2State* __restrict__ bs, uint32_t* __restrict__ wcnt, uint32_t nwcnt, uint32_t t, uint32_t step) {
    static_assert(sizeof(Bi2State) % 16 == 0, "cleared with 16-byte stores");
    uint4* const   p = reinterpret_cast<uint4*>(bs);
    const uint32_t n = (uint32_t)(sizeof(Bi2State) / 16);
    for (uint32_t i = t; i < n; i += step) p[i] = make_uint4(0u, 0u, 0u, 0u);
    for (uint32_t i = t; i < nwcnt; i += step) wcnt[i] = 0u;
}
__global__ __launch_bounds__(kBlock) void chain_reset_kernel(Bi2State* __restrict__ bs, uint32_t* __restrict__ wcnt, uint32_t nwcnt) {
    chain_clear_state(bs, wcnt, nwcnt, blockIdx.x * kBlock + threadIdx.x, gridDim.x * kBlock);
}

// ---- bitmap: per position bucket, the listed positions -> one bit each; st->valid += set bits ---------------------------------------------------------------------------
// The 16 words beyond the corpus read zero (the last bucket's block clears them: chain_emit_kernel looks at bit i + 1).
__global__ __launch_bounds__(kBi2BmThreads) void chain_bitmap_kernel(uint32_t npos, const Bi2State* __restrict__ bs, const uint32_t* __restrict__ plist, Bi2Lists pl, DevState* __restrict__ st,
                                                                      uint32_t* __restrict__ bitmap, const uint32_t* __restrict__ pcode = nullptr /* order 2: the head windows' codes ... */,
                                                                      const uint32_t* __restrict__ headid = nullptr /* ... and who of them survived */,
                                                                      uint32_t* __restrict__ wpre = nullptr /* optional, per bitmap word: set bits of the bucket before the word ... */,
                                                                      uint32_t* __restrict__ btot = nullptr /* ... and per bucket: its set bits (chain_pairs_kernel ranks with them) */) {
    if (st->done) return;
    extern __shared__ uint32_t bmL[];  // (1 << pshift) / 32 words
    __shared__ uint32_t        redL[kBi2BmThreads / kWave];
    const uint32_t b = blockIdx.x, start = b << pl.pshift;
    if (start >= npos) return;
    const uint32_t size   = min(1u << pl.pshift, npos - start);
    const uint32_t nwords = (size + 31) / 32;
    for (uint32_t w = threadIdx.x; w < nwords; w += kBi2BmThreads) bmL[w] = 0;
    __syncthreads();
    if (headid != nullptr) {  // the head windows' list: only the windows of surviving pairs count
        uint32_t first, cap;
        bi2_list_of(pl, kBi2Shards, b, first, cap);
        const uint32_t n = min(bs->pcur[bi2_pc(kBi2Shards * kBi2Buckets + b)], cap);
        for (uint32_t j = threadIdx.x; j < n; j += kBi2BmThreads) {
            const uint32_t o = plist[first + j] - start;
            if (chain_head_alive(pcode[first + j], headid)) atomicOr(&bmL[o >> 5], 1u << (o & 31u));
        }
    }
    for (uint32_t x = 0; x < (uint32_t)kBi2Shards; ++x) {
        uint32_t first, cap;
        bi2_list_of(pl, x, b, first, cap);
        const uint32_t     n = min(bs->pcur[bi2_pc(x * kBi2Buckets + b)], cap);
        const uint32_t*    p = plist + first;  // 16-byte aligned: pcap and hbase are multiples of 4
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
    uint32_t nset = 0;
    for (uint32_t w = threadIdx.x; w < nwords; w += kBi2BmThreads) {
        const uint32_t x           = bmL[w];
        bitmap[(start >> 5) + w] = x;
        nset += (uint32_t)__popc(x);
    }
    if (start + size == npos && threadIdx.x < 16) bitmap[(start >> 5) + nwords + threadIdx.x] = 0u;
    if (wpre != nullptr) {  // a lane takes a run of consecutive words: exclusive prefix of their popcounts inside the bucket
        const uint32_t per = (nwords + kBi2BmThreads - 1) / kBi2BmThreads, w0 = threadIdx.x * per, w1 = min(nwords, w0 + per);
        uint32_t       mine = 0;
        for (uint32_t w = w0; w < w1; ++w) mine += (uint32_t)__popc(bmL[w]);
        uint32_t total;
        uint32_t run = bi2_block_scan<kBi2BmThreads>(mine, &total, redL);
        for (uint32_t w = w0; w < w1; ++w) {
            wpre[(start >> 5) + w] = run;
            run += (uint32_t)__popc(bmL[w]);
        }
        if (threadIdx.x == 0) btot[b] = total;
    }
    for (int off = 32; off > 0; off >>= 1) nset += __shfl_down(nset, off, kWave);
    if ((threadIdx.x & (kWave - 1)) == 0) redL[threadIdx.x / kWave] = nset;
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t v = 0;
        for (int w = 0; w < kBi2BmThreads / kWave; ++w) v += redL[w];
        if (v) atomicAdd(&st->valid, v);
    }
}

// ---- emit: (position, code) pairs of order n - 1 -> 8-byte records of order n, partitioned by A bin ---------------------------------------------------------------------
// grid: a multiple of nsub persistent blocks. Candidates are appended to an LDS queue step by step (a block-wide scan of the lanes' counts: no atomics, a deterministic
// order); whenever the queue holds a step's worth it is counting-sorted by A bin in place and leaves as one run per (queue, A bin) into the block's sub-region, exactly as
// bi2_emit_kernel's tiles do. Key bits: idbits (for the survivors order n - 1 kept) + clsbits; what does not fit the record or the count kernel's 31-bit in-bin key raises
// Bi2State::overflow (the host repeats the run on the first-generation kernels for these orders).
struct ChainQueue {
    unsigned long long* qL;   // [kChQueue]
    uint8_t*            qaL;  // [kChQueue]
    uint32_t *          histL, *offL, *gbaseL, *wsumL;  // [kBins] x 3, [kChThreads / kWave]
    uint32_t            qn;   // records in the queue (block-uniform)
    uint32_t            nadm; // records appended by this block (block-uniform)
    // the queue's records leave: counting sort by A bin in place (every lane holds its entries in registers across the barrier), one reservation per (queue, A bin)
    __device__ __forceinline__ void flush(unsigned long long* __restrict__ recsA, uint32_t region, uint32_t sub, Bi2State* __restrict__ bs) {
        __syncthreads();  // the appends are visible
        if (threadIdx.x < kBins) histL[threadIdx.x] = 0;
        __syncthreads();
        unsigned long long r[kChQPer];
        uint32_t           rk[kChQPer];
#pragma unroll
        for (int q = 0; q < kChQPer; ++q) {
            const uint32_t j = q * kChThreads + threadIdx.x;
            r[q]             = 0;
            rk[q]            = kInvalid;
            if (j < qn) {
                r[q]             = qL[j];
                const uint32_t a = qaL[j];
                rk[q]            = atomicAdd(&histL[a], 1u) | (a << 16);
            }
        }
        __syncthreads();
        bi2_scan256(histL, offL, wsumL);
        // (the reservation's answer is first needed by the copy-out: the memory-side atomic travels while the queue is sorted in place — round 6)
        uint32_t rs_at = 0, rs_h = 0;
        if (threadIdx.x < kBins) {
            rs_h = histL[threadIdx.x];
            if (rs_h) rs_at = atomicAdd(&bs->curA[bi2_cur(sub * kBins + threadIdx.x)], rs_h);
        }
#pragma unroll
        for (int q = 0; q < kChQPer; ++q) {
            if (rk[q] != kInvalid) {
                const uint32_t a = rk[q] >> 16, p = offL[a] + (rk[q] & 0xFFFFu);
                qL[p]            = r[q];
                qaL[p]           = (uint8_t)a;
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
        }
        __syncthreads();
        for (uint32_t j = threadIdx.x; j < qn; j += kChThreads) {
            const uint32_t a                         = qaL[j];
            recsA[(size_t)gbaseL[a] + (j - offL[a])] = qL[j];
        }
        __syncthreads();
        qn = 0;
    }
};
static_assert(kChStep / kChThreads == kChPer && kChThreads >= kBins && kChThreads % kWave == 0 && kChThreads / kWave <= 16, "bi2_scan256 / bi2_block_scan");

struct ChainKey {
    uint32_t           K, clsbits, pb;
    uint32_t           pshift, pdrop;  // key-sharded runs whose key and position exceed 64 bits (kshard2.hpp): the record's position lacks the three bits above pshift
    unsigned long long kmask;
    __device__ __forceinline__ void put(ChainQueue& q, uint32_t at, uint32_t dn, uint32_t cn, uint32_t pos) const {
        const uint64_t m = bi2_mix(((uint64_t)dn << clsbits) | cn, K);
        if (pdrop) pos = ((pos >> (pshift + 3)) << pshift) | (pos & ((1u << pshift) - 1u));
        q.qL[at]  = ((m & kmask) << pb) | pos;
        q.qaL[at] = (uint8_t)(m >> (K - 8));
    }
};
// key bits of order n from what order n - 1 kept; false (and Bi2State::overflow) when the engine cannot hold them
// kfix (key-sharded runs): the key bits every rank agreed on (the numbers are global there); pb is then the width of the record's position field
__device__ __forceinline__ bool chain_key_bits(const Bi2State* __restrict__ prev, uint32_t clsbits, uint32_t pb, Bi2State* __restrict__ bs, ChainKey& ck, uint32_t hugebin = 0,
                                               uint32_t kfix = 0, uint32_t pshift = 0, uint32_t pdrop = 0) {
    const uint32_t kept_prev = prev->kept_bins + prev->kept_head;
    uint32_t       idbits    = 1;
    while (idbits < 32 && (1ull << idbits) < (uint64_t)kept_prev + 1) ++idbits;
    const uint32_t K = kfix ? kfix : max(idbits + clsbits, 17u);
    if (threadIdx.x == 0) {
        bs->kbits   = K;
        bs->posbits = pb;
        bs->hugebin = hugebin ? hugebin : kChHugeBin;
        bs->pdshift = pdrop ? pshift + 1u : 0u;
    }
    ck.pshift = pshift;
    ck.pdrop  = pdrop;
    if ((!kfix && K > 48u) || K - 8u + pb > 64u) {  // (the count kernel's in-bin key holds K - 17 + bshift <= 31 bits; the record K - 8 bits beside the position)
        if (threadIdx.x == 0) bs->overflow = 1;
        return false;
    }
    ck.K       = K;
    ck.clsbits = clsbits;
    ck.pb      = pb;
    ck.kmask   = (1ull << (K - 8)) - 1ull;
    return true;
}

#define CHAIN_QUEUE_LDS(q)                                                                       \
    __shared__ unsigned long long q##_qL[kChQueue];                                              \
    __shared__ uint8_t            q##_qaL[kChQueue];                                             \
    __shared__ uint32_t           q##_histL[kBins], q##_offL[kBins], q##_gbaseL[kBins], q##_wsumL[kChThreads / kWave]; \
    ChainQueue q{q##_qL, q##_qaL, q##_histL, q##_offL, q##_gbaseL, q##_wsumL, 0u, 0u}

// (a) the listed windows.
// Where a step's gathers land decides this kernel: a pair's class id at i + n - 1 is a 4-byte read inside its bucket's 512 KB window of the class-id array, and block b
// runs on XCD b % 8 with its own 4 MB L2. (Round 6 measured a four-stage software pipeline of this loop — every stage's loads issued an iteration before their use, in
// waiting order, unconditional; tools/notes/chain_emit_pipelined.hpp.txt —: the blocks' cycles fell from 1038 M to 672 M per step, but at 104 registers two blocks are resident
// per CU instead of three and the kernel's time rose from 0.65 to 0.69 ms: it is bound by the gathers' issue rate — one line per lane and clock in the texture unit —, not
// by their latency.) The first version gave every block whole (shard, bucket) lists, eight consecutive blocks the eight shards of one bucket: every
// XCD fetched every window, ~100 windows were open per XCD at a time, and each gather went to HBM for a whole line (45 M pairs: 0.75 ms, ~6 GB of fetches for 0.18 GB of
// class ids). Now the buckets are dealt to the XCDs (bucket mod 8), an XCD's steps — pieces of kChStep pairs of its lists, bucket by bucket — are numbered by
// chain_steps_kernel, and the XCD's blocks take them round-robin: at any time an XCD works on ~3 adjacent buckets, whose windows stay in its L2.
constexpr uint32_t kChXcds = 8;
// the step table of XCD x (block-wide: every thread of a kBi2Threads block calls it): the XCD's lists in (bucket, shard) order, cut into pieces of kChStep pairs
__device__ __forceinline__ void chain_number_steps(const Bi2State* __restrict__ prev, const Bi2Lists& pl, uint32_t nbuckets, uint2* __restrict__ table, uint32_t cap,
                                                   uint32_t* __restrict__ nsteps, uint32_t x, uint32_t* wsumL) {
    constexpr uint32_t kPer = (kBi2Buckets / kChXcds * kChLists + kBi2Threads - 1) / kBi2Threads;  // lists of an XCD per lane (2), in (bucket, shard) order
    uint32_t           first[kPer], n[kPer], ns[kPer], sum = 0;
#pragma unroll
    for (uint32_t q = 0; q < kPer; ++q) {
        const uint32_t idx = threadIdx.x * kPer + q, bucket = x + kChXcds * (idx / kChLists), shard = idx % kChLists;
        uint32_t       lcap = 0;
        first[q] = n[q] = 0;
        if (bucket < nbuckets) {
            bi2_list_of(pl, shard, bucket, first[q], lcap);
            n[q] = min(prev->pcur[bi2_pc(shard * kBi2Buckets + bucket)], lcap);
        }
        ns[q] = (n[q] + kChStep - 1) / kChStep;
        sum += ns[q];
    }
    uint32_t total;
    uint32_t base = bi2_block_scan<kBi2Threads>(sum, &total, wsumL);
#pragma unroll
    for (uint32_t q = 0; q < kPer; ++q)
        for (uint32_t k = 0; k < ns[q]; ++k, ++base)
            if (base < cap) table[(size_t)x * cap + base] = make_uint2(first[q] + k * (uint32_t)kChStep, min((uint32_t)kChStep, n[q] - k * (uint32_t)kChStep));
    if (threadIdx.x == 0) nsteps[x] = min(total, cap);
}
// table: [kChXcds][cap] entries {index of the step's first pair in plist / pcode, pairs of the step}; nsteps: [kChXcds]
__global__ __launch_bounds__(kBi2Threads) void chain_steps_kernel(const Bi2State* __restrict__ prev, Bi2Lists pl, uint32_t nbuckets, uint2* __restrict__ table, uint32_t cap,
                                                                   uint32_t* __restrict__ nsteps, const DevState* __restrict__ st) {
    if (st->done) return;
    __shared__ uint32_t wsumL[kBi2Threads / kWave];
    chain_number_steps(prev, pl, nbuckets, table, cap, nsteps, blockIdx.x, wsumL);
}
// chain_reset_kernel and chain_steps_kernel in ONE launch (what an order starts with; they touch different states — the new order's, and the lists of the one before):
// blocks 0 .. 7 number the XCDs' steps, the others clear. grid kChXcds + kChResetBlocks, kBi2Threads threads.
constexpr uint32_t kChResetBlocks = 64;
__global__ __launch_bounds__(kBi2Threads) void chain_begin_kernel(const Bi2State* __restrict__ prev, Bi2Lists pl, uint32_t nbuckets, uint2* __restrict__ table, uint32_t cap,
                                                                   uint32_t* __restrict__ nsteps, const DevState* __restrict__ st, Bi2State* __restrict__ bs, uint32_t* __restrict__ wcnt,
                                                                   uint32_t nwcnt) {
    if (blockIdx.x >= kChXcds) {  // (cleared whether or not the run has ended, as chain_reset_kernel does)
        chain_clear_state(bs, wcnt, nwcnt, (blockIdx.x - kChXcds) * kBi2Threads + threadIdx.x, (gridDim.x - kChXcds) * kBi2Threads);
        return;
    }
    if (st->done) return;
    __shared__ uint32_t wsumL[kBi2Threads / kWave];
    chain_number_steps(prev, pl, nbuckets, table, cap, nsteps, blockIdx.x, wsumL);
}
// (steps of an XCD at most: every list of its buckets filled + one partial step each)
inline uint32_t chain_steps_cap(const Bi2Lists& pl) { return (kBi2Buckets / kChXcds) * (kBi2Shards * (pl.pcap / kChStep + 1) + ((1u << pl.pshift) / kChStep + 1)); }

__global__ __launch_bounds__(kChThreads, 6) void chain_emit_kernel(const uint32_t* __restrict__ cls, uint32_t npos, uint32_t n, uint32_t clsbits, uint32_t pb, const Bi2State* __restrict__ prev,
                                                                    const uint32_t* __restrict__ plist, const uint32_t* __restrict__ pcode, Bi2Lists pl,
                                                                    const uint2* __restrict__ table, uint32_t cap, const uint32_t* __restrict__ nsteps,
                                                                    const uint32_t* __restrict__ bitmap, unsigned long long* __restrict__ recsA, uint32_t region, uint32_t nsub,
                                                                    Bi2State* __restrict__ bs, DevState* __restrict__ st, const uint32_t* __restrict__ headid /* order 3 */, uint32_t dbg = 0,
                                                                    uint32_t kfix = 0, uint32_t pdrop = 0 /* key-sharded runs: agreed key bits; pb then excludes the dropped bits */) {
    if (st->done) return;
    ChainKey ck;
    if (!chain_key_bits(prev, clsbits, pb, bs, ck, dbg >> 8, kfix, pl.pshift, pdrop)) return;
    CHAIN_QUEUE_LDS(Q);
    const uint32_t        sub = blockIdx.x % nsub;
    const uint32_t        x = blockIdx.x % kChXcds, nper = gridDim.x / kChXcds, ns = nsteps[x];  // (grid: a multiple of 8)
    const uint2* const    tab = table + (size_t)x * cap;
    const uint32_t        rbase = kfix ? 0u : prev->res_base;  // (key-sharded runs: the head table holds global numbers)
    uint32_t              k = blockIdx.x / kChXcds;
    // software pipeline, two steps deep: the table entry of step k + 2 nper and the pairs of step k + nper are in flight while step k's gathers and partition run
    // (every one of these is a memory round trip; issued one after the other they were the step's time)
    uint32_t ps[kChPer], code[kChPer];
    bool     ok[kChPer];
    uint2    e1 = k < ns ? tab[k] : make_uint2(0u, 0u);  // the step whose pairs are loaded next
    auto     load_pairs = [&]() {
#pragma unroll
        for (int q = 0; q < kChPer; ++q) {
            const uint32_t j = q * kChThreads + threadIdx.x;
            ok[q]            = j < e1.y;
            ps[q]            = ok[q] ? plist[(size_t)e1.x + j] : 0u;
            code[q]          = ok[q] ? pcode[(size_t)e1.x + j] : 0u;
        }
    };
    load_pairs();
    e1 = k + nper < ns ? tab[k + nper] : make_uint2(0u, 0u);
    KP_INIT(4);
    while (k < ns) {
        uint32_t p2[kChPer], c2[kChPer], w[kChPer], cn[kChPer];
        bool     k2[kChPer];
#pragma unroll
        for (int q = 0; q < kChPer; ++q) {
            p2[q] = ps[q];
            c2[q] = code[q];
            k2[q] = ok[q];
        }
#pragma unroll
        for (int q = 0; q < kChPer; ++q)  // first gather: did the (n-1)-gram at i + 1 survive (the bucket's 16 KB of the bitmap: mostly the CU's own cache)
            w[q] = (k2[q] && !(dbg & 1u)) ? bitmap[(p2[q] + 1u) >> 5] : 0x55555555u;
        load_pairs();  // of step k + nper (its entry arrived a step ago)
        k += nper;
        e1 = k + nper < ns ? tab[k + nper] : make_uint2(0u, 0u);
        uint32_t cnt = 0;
#pragma unroll
        for (int q = 0; q < kChPer; ++q) {
            k2[q] = k2[q] && ((w[q] >> ((p2[q] + 1u) & 31u)) & 1u);
            if (k2[q] && (c2[q] & kBi2HeadCode)) {  // a head window of order 2: the dense number of its pair, if the pair survived
                const uint32_t r = headid[c2[q] & 0xFFFu];
                k2[q]            = r != kInvalid;
                c2[q]            = r - rbase;
            }
            cnt += k2[q] ? 1u : 0u;
        }
        KP(0);  // (pairs arrived, bitmap words gathered)
#pragma unroll
        for (int q = 0; q < kChPer; ++q)  // second gather, for the admitted windows only: the class id at i + n - 1, anywhere in the bucket's 512 KB of class ids — a line from
            cn[q] = (k2[q] && !(dbg & 2u)) ? cls[p2[q] + n - 1u] : (p2[q] & 1023u);  // L2 per window (~120 G/s on MI355X): the kernel's time
        uint32_t total;
        uint32_t at = Q.qn + bi2_block_scan<kChThreads>(cnt, &total, Q.wsumL);
        KP(1);  // (the block's scan: barriers — every wave's bitmap words are in)
#pragma unroll
        for (int q = 0; q < kChPer; ++q)
            if (k2[q]) ck.put(Q, at++, c2[q] /* bi2_pospart_kernel left dense survivor numbers */, cn[q], p2[q]);
        Q.qn += total;
        Q.nadm += total;
        KP(2);  // (class ids gathered, records mixed and queued: thread 0's own)
        if (Q.qn >= (uint32_t)kChStep) {
            Q.flush(recsA, region, sub, bs);
            KP(3);
        }
    }
    KP_DONE();
    if (Q.qn) Q.flush(recsA, region, sub, bs);
    if (threadIdx.x == 0 && Q.nadm) atomicAdd(&st->admitted, Q.nadm);
    bi2_offsets_tail(bs, region, nsub, Q.histL, Q.offL, Q.wsumL, Q.gbaseL);
}

// ---- a skipgram pass of TWO parts on the same engine (exhaustive skipgrams of a chained run: reference patternmodel.h:1163-1171 -> computeskipgrams :1370-1527) -----------
// The masked form of an admitted window is named by its two parts' ids (a class id for a one-token part, the result index of the part's n-gram otherwise): a key of
// lbits + rbits bits, mixed and cut into 8-byte records exactly like an order's (number, class) keys — so the pass runs level B, the wave-per-bin count and the
// compaction of the n-gram orders instead of round 1's 12-byte records (bin_emit / bin_hist2 / bin_scatter / bin_count). list: the admitted windows of the order
// (chain_alist_kernel); a key that does not fit (two result indices of 24 bits beside 27 position bits) raises Bi2State::overflow like an order's: the run repeats on
// round 3's passes.
// HEAD (both parts one token): the frames of two frequent words ("the _ of") are as hot as the bigrams of order 2 — without its dense head the pass spends 0.84 ms in
// the workgroup kernel for hot bins — and are counted like them: pairs of classes below kBi2Head in an LDS histogram per block (count, lowest position), rows reduced
// by bi2_head_reduce_kernel, survivors appended by bi2_finish / bi2_compact. head_rows: [gridDim.x][2][kBi2HeadN] (gridDim.x <= kBi2EmitGrid).
template <bool HEAD>
__global__ __launch_bounds__(kChThreads, 6) void skip_emit_kernel(const uint32_t* __restrict__ list, const uint32_t* __restrict__ nlist, const uint32_t* __restrict__ left, uint32_t offl,
                                                                   const uint32_t* __restrict__ right, uint32_t offr, uint32_t l_is_cls, uint32_t r_is_cls, uint32_t clsbits, uint32_t pb,
                                                                   unsigned long long* __restrict__ recsA, uint32_t region, uint32_t nsub, Bi2State* __restrict__ bs, DevState* __restrict__ st,
                                                                   uint32_t* __restrict__ head_rows = nullptr) {
    if (st->done) return;
    __shared__ uint32_t headL[HEAD ? kBi2HeadN : 1], hposL[HEAD ? kBi2HeadN : 1];
    if (HEAD) {
        for (int k = threadIdx.x; k < kBi2HeadN; k += kChThreads) {
            headL[k] = 0;
            hposL[k] = 0xFFFFFFFFu;
        }
        __syncthreads();
    }
    auto flush_head = [&]() {
        if (HEAD) {
            __syncthreads();
            uint32_t* const row = head_rows + (size_t)blockIdx.x * (2 * kBi2HeadN);
            for (int k = threadIdx.x; k < kBi2HeadN; k += kChThreads) {
                row[k]             = headL[k];
                row[kBi2HeadN + k] = hposL[k];
            }
        }
    };
    uint32_t idb = 1;
    while (idb < 32 && (1ull << idb) < (uint64_t)st->res_total + 1) ++idb;
    const uint32_t lbits = l_is_cls ? clsbits : idb, rbits = r_is_cls ? clsbits : idb;
    ChainKey       ck;
    if (lbits + rbits > 48u) {
        if (threadIdx.x == 0) bs->overflow = 1;
        flush_head();
        return;
    }
    if (!chain_key_bits(bs, rbits, pb, bs, ck, 0, max(lbits + rbits, 17u))) {
        flush_head();
        return;
    }
    CHAIN_QUEUE_LDS(Q);
    const uint32_t sub = blockIdx.x % nsub, n = *nlist;
    for (uint32_t t0 = blockIdx.x * (uint32_t)kChStep; t0 < n; t0 += gridDim.x * (uint32_t)kChStep) {
        uint32_t p[kChPer], l[kChPer], r[kChPer], cnt = 0;
        bool     ok[kChPer];
#pragma unroll
        for (int q = 0; q < kChPer; ++q) {
            const uint32_t j = t0 + q * kChThreads + threadIdx.x;
            ok[q]            = j < n;
            p[q]             = ok[q] ? list[j] : 0u;
        }
#pragma unroll
        for (int q = 0; q < kChPer; ++q) {
            l[q] = ok[q] ? left[p[q] + offl] : 0u;
            r[q] = ok[q] ? right[p[q] + offr] : 0u;
        }
#pragma unroll
        for (int q = 0; q < kChPer; ++q) {
            ok[q] = ok[q] && l[q] != kInvalid && r[q] != kInvalid;  // (cannot fail for an admitted window; kept as a guard)
            if (HEAD && ok[q] && l[q] < (uint32_t)kBi2Head && r[q] < (uint32_t)kBi2Head) {
                const uint32_t h = l[q] * kBi2Head + r[q];
                atomicAdd(&headL[h], 1u);
                atomicMin(&hposL[h], p[q]);
                ok[q] = false;
            }
            cnt += ok[q] ? 1u : 0u;
        }
        uint32_t total;
        uint32_t at = Q.qn + bi2_block_scan<kChThreads>(cnt, &total, Q.wsumL);
#pragma unroll
        for (int q = 0; q < kChPer; ++q)
            if (ok[q]) ck.put(Q, at++, l[q], r[q], p[q]);
        Q.qn += total;
        Q.nadm += total;
        if (Q.qn >= (uint32_t)kChStep) Q.flush(recsA, region, sub, bs);
    }
    if (Q.qn) Q.flush(recsA, region, sub, bs);
    flush_head();
    if (threadIdx.x == 0 && Q.nadm) atomicAdd(&st->admitted, Q.nadm);
    bi2_offsets_tail(bs, region, nsub, Q.histL, Q.offL, Q.wsumL, Q.gbaseL);
}

// ---- result indices per position (the modes that keep every order's ids), with the step order of chain_emit_kernel ------------------------------------------------------
// Rounds 2-3 (bi2_ids_kernel) scattered a bucket's 4-byte ids into its 512 KB window with one block per bucket: all ~800 windows are open at once, a line leaves L2 before its other
// ids arrive, and 1.33 GB are written to store 0.42 GB (rounds 2-3). Here an XCD's blocks walk the XCD's buckets together, piece by piece (chain_steps_kernel's tables):
// ~3 windows are open per L2 and a line is written once it is whole. pcode: dense survivor numbers (bi2_pospart_kernel, dense = true).
__global__ __launch_bounds__(kChThreads) void chain_ids_kernel(const uint32_t* __restrict__ plist, const uint32_t* __restrict__ pcode, const uint2* __restrict__ table, uint32_t cap,
                                                                const uint32_t* __restrict__ nsteps, const Bi2State* __restrict__ bs, const DevState* __restrict__ st,
                                                                uint32_t* __restrict__ ids, const uint32_t* __restrict__ headid = nullptr /* order 2 of a chained run: the head
                                                                windows are listed too (kBi2HeadCode | pair); their result indices, kInvalid where the pair did not survive */) {
    if (st->done) return;
    const uint32_t     x = blockIdx.x % kChXcds, nper = gridDim.x / kChXcds, ns = nsteps[x], res_base = bs->res_base;
    const uint2* const tab = table + (size_t)x * cap;
    for (uint32_t k = blockIdx.x / kChXcds; k < ns; k += nper) {
        const uint2 e = tab[k];
        uint32_t    ps[kChPer], cd[kChPer];
#pragma unroll
        for (int q = 0; q < kChPer; ++q) {
            const uint32_t j = q * kChThreads + threadIdx.x;
            ps[q]            = j < e.y ? plist[(size_t)e.x + j] : kInvalid;
            cd[q]            = j < e.y ? pcode[(size_t)e.x + j] : 0u;
        }
#pragma unroll
        for (int q = 0; q < kChPer; ++q)
            if (ps[q] != kInvalid) ids[ps[q]] = (headid != nullptr && (cd[q] & kBi2HeadCode)) ? headid[cd[q] & 0xFFFu] : res_base + cd[q];
    }
}

// Round 5: the same array WITHOUT a scatter to memory. A block owns one part (a quarter or an eighth: at most 32 768 positions = 128 KB of ids) of one bucket's window,
// builds it in LDS from the bucket's lists — it reads all of them and keeps the pairs of its part; the blocks of a bucket's parts run on one XCD (blockIdx % 8) one
// after the other in dispatch order, so the lists come from HBM once — and writes the window out as whole lines, kInvalid where no window survived: no fill of the array
// beforehand, no partial-line writes (chain_ids_kernel: 1.59 GB written for 0.42 GB of ids, 0.4-0.66 ms per order; the fill: 0.1 ms).
// grid: ceil(nbuckets / 8) * parts * 8 blocks of kBi2Threads; dynamic LDS: 4 << (pshift - plog) bytes.
// PairsOut (indexed models, order 2): the part's ids ARE the forward index's pairs of these positions, in position order — pair k of the part lands at
// (pairs before the bucket) + (the bitmap's prefix at the part's first word: chain_bitmap_kernel's wpre / btot) + k: no array of ids in memory, no counting and no
// writing sweep over it (emit_count / emit_write_kernel: 0.13 + 0.49 ms per 10^8 tokens). Same content and order as emit_write_kernel's pairs.
struct ChainPairsOut {
    const uint32_t*           wpre;    // per bitmap word: set bits of the bucket before the word; nullptr: no pairs
    const uint32_t*           btot;    // per bucket: its set bits
    const uint4*              blocks;  // PosBlock per 64 positions
    const unsigned long long* chain;   // pair counters: chain[which] = pairs written so far
    unsigned long long*       pairs;
    uint32_t*                 pay;     // split pairs: ids as u32 in `pairs`, references here; nullptr: packed 64-bit pairs
    unsigned long long        pcap;
    int                       which;
    uint32_t                  sb, tb;
};
__global__ __launch_bounds__(kBi2Threads) void chain_ids_full_kernel(const uint32_t* __restrict__ plist, const uint32_t* __restrict__ pcode, Bi2Lists pl, uint32_t nbuckets, uint32_t plog,
                                                                      uint32_t npos, const Bi2State* __restrict__ bs, const DevState* __restrict__ st, uint32_t* __restrict__ ids /* or nullptr */,
                                                                      const uint32_t* __restrict__ headid, ChainPairsOut po) {
    if (st->done) return;
    extern __shared__ uint32_t idsL[];
    __shared__ uint32_t        wsumL[kBi2Threads / kWave];
    const uint32_t xcd = blockIdx.x % kChXcds, part = (blockIdx.x / kChXcds) & ((1u << plog) - 1u), bucket = (blockIdx.x / (kChXcds << plog)) * kChXcds + xcd;
    if (bucket >= nbuckets) return;
    const uint32_t wlen = (1u << pl.pshift) >> plog, start = (bucket << pl.pshift) + part * wlen;
    if (start >= npos) return;
    for (uint32_t k = threadIdx.x; k < wlen; k += kBi2Threads) idsL[k] = kInvalid;
    __syncthreads();
    const uint32_t res_base = bs->res_base, nl = headid != nullptr ? kChLists : (uint32_t)kBi2Shards;
    // all lists of the bucket at once: their lengths, then one 16-byte load per list and round in flight together (list after list, each round trip waited for the one
    // before: 0.51 ms per order, a block per CU has nothing else to run)
    uint32_t first[kChLists], nn[kChLists], nmax = 0;
#pragma unroll
    for (uint32_t x = 0; x < kChLists; ++x) {
        uint32_t cap = 0;
        first[x] = nn[x] = 0;
        if (x < nl) {
            bi2_list_of(pl, x, bucket, first[x], cap);
            nn[x] = min(bs->pcur[bi2_pc(x * kBi2Buckets + bucket)], cap);
        }
        nmax = max(nmax, nn[x]);
    }
    auto put = [&](uint32_t pos, uint32_t at) {
        const uint32_t o = pos - start;
        if (o < wlen) {
            const uint32_t cd = pcode[at];
            idsL[o]           = (headid != nullptr && (cd & kBi2HeadCode)) ? headid[cd & 0xFFFu] : res_base + cd;
        }
    };
    for (uint32_t j = threadIdx.x; j * 4u < nmax; j += kBi2Threads) {
        uint4 e[kChLists];
#pragma unroll
        for (uint32_t x = 0; x < kChLists; ++x)  // (16-byte aligned: pcap and hbase are multiples of 4; a list's last, partial vector is read whole — the lists have room —
            e[x] = j * 4u < nn[x] ? reinterpret_cast<const uint4*>(plist + first[x])[j] : make_uint4(0u, 0u, 0u, 0u);  // and cut by the length below)
#pragma unroll
        for (uint32_t x = 0; x < kChLists; ++x) {
            const uint32_t at = first[x] + 4u * j, left = nn[x] > 4u * j ? nn[x] - 4u * j : 0u;
            if (left > 0u) put(e[x].x, at);
            if (left > 1u) put(e[x].y, at + 1u);
            if (left > 2u) put(e[x].z, at + 2u);
            if (left > 3u) put(e[x].w, at + 3u);
        }
    }
    __syncthreads();
    const uint32_t m = min(wlen, npos - start);
    if (ids != nullptr)
        for (uint32_t k = threadIdx.x; k < m; k += kBi2Threads) ids[start + k] = idsL[k];
    if (po.wpre == nullptr) return;
    // pairs: a wave takes rows of 64 consecutive positions (one entry of the position table per row), ballots give every pair its place: coalesced stores.
    // (A lane per 32 consecutive positions, each writing its own run of ~14 pairs, made every store a partial line: 0.9 ms slower than the sweeps it replaces.)
    uint32_t before = 0;  // set bits of the buckets before this one
    for (uint32_t k = threadIdx.x; k < bucket; k += kBi2Threads) before += po.btot[k];
    uint32_t btotal;
    bi2_block_scan<kBi2Threads>(before, &btotal, wsumL);
    constexpr uint32_t kW   = kBi2Threads / kWave;
    const uint32_t     lane = threadIdx.x & (kWave - 1), wave = threadIdx.x / kWave;
    const uint32_t     rows = (m + kWave - 1) / kWave, rper = (rows + kW - 1) / kW, r0 = wave * rper, r1 = min(rows, r0 + rper);  // this wave's rows
    uint32_t           mine = 0;
    for (uint32_t r = r0; r < r1; ++r) {
        const uint32_t k = r * kWave + lane;
        mine += (uint32_t)__popcll(__ballot(k < m && idsL[k] != kInvalid));
    }
    uint32_t       ptotal;
    const uint32_t wexcl = bi2_block_scan<kBi2Threads>(lane == 0 ? mine : 0u, &ptotal, wsumL);  // pairs of the waves before (the value of the wave's lane 0 counts)
    const uint32_t wbase = (uint32_t)__shfl((int)wexcl, 0, kWave);
    const uint32_t tmask = po.tb >= 32 ? 0xFFFFFFFFu : (1u << po.tb) - 1u;
    uint64_t       o     = po.chain[po.which] + btotal + po.wpre[start >> 5] + wbase;
    for (uint32_t r = r0; r < r1; ++r) {
        const uint32_t k = r * kWave + lane, p = start + k;
        const uint32_t id = k < m ? idsL[k] : kInvalid;
        const uint64_t mk = __ballot(id != kInvalid);
        if (mk == 0) continue;
        const uint4 rb = po.blocks[p >> 6];  // (start is a multiple of 64: the row is one block of the table, the same address for the 64 lanes)
        if (id != kInvalid) {
            const uint64_t at = o + (uint32_t)__popcll(mk & ((1ull << lane) - 1ull));
            if (at < po.pcap) {  // (pairs_advance_kernel flags the overflow)
                const uint32_t bit   = p & 63u;
                const uint64_t below = ((((uint64_t)rb.w << 32) | rb.z)) & ((1ull << bit) - 1ull);
                const uint32_t sent = rb.x + (uint32_t)__popcll(below), tok = below ? bit - (64u - (uint32_t)__clzll(below)) : p - rb.y;
                if (po.pay != nullptr) {
                    reinterpret_cast<uint32_t*>(po.pairs)[at] = id;
                    po.pay[at]                                = (sent << po.tb) | (tok & tmask);
                } else {
                    po.pairs[at] = ((unsigned long long)id << (po.sb + po.tb)) | ((unsigned long long)sent << po.tb) | (tok & tmask);
                }
            }
        }
        o += (uint32_t)__popcll(mk);
    }
}

// ---- the forward index's (pattern, reference) pairs of an order, straight from the order's position lists (indexed models without skipgram passes) ---------------------
// emit_count / emit_write compact an order's ids per position into pairs in position order: two sweeps over an array of npos ids that chain_ids_kernel first has to
// scatter (and a fill before it). The bitmap of the listed positions IS that compaction's index: the pair of position p lands at
// (set bits before p) = btot's prefix up to p's bucket + wpre[word of p] + popcount of the word's bits below p — a perfect rank, no sweep, no ids. Same step order as
// chain_ids_kernel (the bucket's words of bitmap / wpre / the position-block table stay in the XCD's L2). Output order and content equal emit_write_kernel's.
__global__ __launch_bounds__(kChThreads) void chain_pairs_kernel(const uint32_t* __restrict__ plist, const uint32_t* __restrict__ pcode, const uint2* __restrict__ table, uint32_t cap,
                                                                  const uint32_t* __restrict__ nsteps, const Bi2State* __restrict__ bs, const DevState* __restrict__ st,
                                                                  const uint32_t* __restrict__ bitmap, const uint32_t* __restrict__ wpre, const uint32_t* __restrict__ btot,
                                                                  uint32_t nbuckets, uint32_t pshift, const uint4* __restrict__ blocks, const unsigned long long* __restrict__ chain,
                                                                  int which, uint64_t pcap, unsigned long long* __restrict__ pairs, uint32_t* __restrict__ pay, uint32_t sb, uint32_t tb,
                                                                  const uint32_t* __restrict__ headid) {
    if (st->done) return;
    __shared__ uint32_t bbaseL[kBi2Buckets], wsumL[kChThreads / kWave];
    static_assert(kBi2Buckets == 2 * kChThreads, "two buckets per lane");
    {
        const uint32_t b0 = 2 * threadIdx.x, t0 = b0 < nbuckets ? btot[b0] : 0u, t1 = b0 + 1 < nbuckets ? btot[b0 + 1] : 0u;
        uint32_t       total;
        const uint32_t excl = bi2_block_scan<kChThreads>(t0 + t1, &total, wsumL);
        bbaseL[b0]     = excl;
        bbaseL[b0 + 1] = excl + t0;
    }
    __syncthreads();
    const uint32_t     x = blockIdx.x % kChXcds, nper = gridDim.x / kChXcds, ns = nsteps[x], res_base = bs->res_base;
    const uint2* const tab   = table + (size_t)x * cap;
    const uint64_t     obase = chain[which];
    const uint32_t     tmask = tb >= 32 ? 0xFFFFFFFFu : (1u << tb) - 1u;
    for (uint32_t k = blockIdx.x / kChXcds; k < ns; k += nper) {
        const uint2 e = tab[k];
        uint32_t    ps[kChPer], cd[kChPer], bw[kChPer], wp[kChPer];
        uint4       r[kChPer];
#pragma unroll
        for (int q = 0; q < kChPer; ++q) {
            const uint32_t j = q * kChThreads + threadIdx.x;
            ps[q]            = j < e.y ? plist[(size_t)e.x + j] : kInvalid;
            cd[q]            = j < e.y ? pcode[(size_t)e.x + j] : 0u;
        }
#pragma unroll
        for (int q = 0; q < kChPer; ++q) {  // all gathers of the lane in flight together
            const uint32_t p = ps[q] != kInvalid ? ps[q] : 0u;
            bw[q]            = bitmap[p >> 5];
            wp[q]            = wpre[p >> 5];
            r[q]             = blocks[p >> 6];
        }
#pragma unroll
        for (int q = 0; q < kChPer; ++q) {
            if (ps[q] == kInvalid) continue;
            const uint32_t p = ps[q];
            if (!((bw[q] >> (p & 31u)) & 1u)) continue;  // (a head window whose pair did not survive)
            const uint32_t id  = (headid != nullptr && (cd[q] & kBi2HeadCode)) ? headid[cd[q] & 0xFFFu] : res_base + cd[q];
            const uint64_t o   = obase + bbaseL[p >> pshift] + wp[q] + (uint32_t)__popc(bw[q] & ((1u << (p & 31u)) - 1u));
            if (o >= pcap) continue;  // (pairs_advance_kernel flags the overflow)
            const uint32_t bit = p & 63u;
            const uint64_t below = ((((uint64_t)r[q].w << 32) | r[q].z)) & ((1ull << bit) - 1ull);
            const uint32_t sent = r[q].x + (uint32_t)__popcll(below), tok = below ? bit - (64u - (uint32_t)__clzll(below)) : p - r[q].y;
            if (pay != nullptr) {
                reinterpret_cast<uint32_t*>(pairs)[o] = id;
                pay[o]                                = (sent << tb) | (tok & tmask);
            } else
                pairs[o] = ((unsigned long long)id << (sb + tb)) | ((unsigned long long)sent << tb) | (tok & tmask);
        }
    }
}

// ---- the windows order n admits, as a list (the exhaustive skipgram passes of order n walk it: reference patternmodel.h:1163-1171, every admitted window) -------------------
// bitmap: the surviving (n-1)-grams (chain_bitmap_kernel); window i is admitted when bits i and i + 1 are set. Tiles take their room with one atomic each: the list is
// ascending inside a tile only, which is all its readers ask for (bi2_list3_kernel's list of order 3 is the same).
__global__ __launch_bounds__(kBlock) void chain_alist_kernel(const uint32_t* __restrict__ bitmap, uint32_t npos, const DevState* __restrict__ st, uint32_t* __restrict__ list_out,
                                                              uint32_t* __restrict__ nlist_out) {
    if (st->done) return;
    __shared__ uint32_t baseL, wsumL[kBlock / kWave];
    constexpr uint32_t  kPerLane = 8;  // consecutive bitmap words per lane: a lane's windows leave as one run (no staging; one scan and one reservation per 65 536 positions)
    const uint32_t      nwords = (npos + 31) / 32, ntiles = (nwords + kBlock * kPerLane - 1) / (kBlock * kPerLane);
    for (uint32_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const uint32_t w0 = (tile * kBlock + threadIdx.x) * kPerLane;
        uint32_t       y[kPerLane], cnt = 0;
        uint32_t       x = w0 < nwords ? bitmap[w0] : 0u;
#pragma unroll
        for (uint32_t k = 0; k < kPerLane; ++k) {
            const uint32_t nx = w0 + k + 1 <= nwords ? bitmap[w0 + k + 1] : 0u;  // (the words beyond the corpus read zero)
            y[k]              = w0 + k < nwords ? x & ((x >> 1) | (nx << 31)) : 0u;
            cnt += (uint32_t)__popc(y[k]);
            x = nx;
        }
        uint32_t       total;
        const uint32_t excl = bi2_block_scan<kBlock>(cnt, &total, wsumL);
        if (threadIdx.x == 0) baseL = total ? atomicAdd(nlist_out, total) : 0u;
        __syncthreads();
        uint32_t o = baseL + excl;
#pragma unroll
        for (uint32_t k = 0; k < kPerLane; ++k) {
            uint32_t yy = y[k];
            while (yy) {
                list_out[o++] = (w0 + k) * 32 + (uint32_t)__builtin_ctz(yy);
                yy &= yy - 1;
            }
        }
        __syncthreads();
    }
}

}  // namespace colibri
