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
//   * order 2's dense head (class pairs below 64 x 64, counted in LDS, never records) joins through a streaming pass over the class ids: head windows carry the dense
//     number bi2_headids_kernel gives their pair.
// add / prune semantics (reference :2059-2073, :2107-2128) are the count kernel's: exact keys, threshold on the exact count, lowest position as representative.
#pragma once
#include "bigram2.hpp"

namespace colibri {

constexpr int kChPer   = 4;                       // candidates per lane and step
constexpr int kChStep  = kBi2Threads * kChPer;    // 4096 candidates per step
constexpr int kChQueue = 2 * kChStep;             // records waiting for a partition step (a step starts with fewer than kChStep of them)
constexpr int kChQPer  = kChQueue / kBi2Threads;  // 8

// ---- bitmap: per position bucket, the listed positions -> one bit each; order 2 also ORs the surviving head windows in; st->valid += set bits -------------------------
// (bi2_bitmap_kernel + the head evaluation bi2_list3_kernel did while streaming.) bitmap words beyond the corpus must read zero (the caller clears 16 of them).
__global__ __launch_bounds__(kBi2BmThreads) void chain_bitmap_kernel(uint32_t npos, const Bi2State* __restrict__ bs, const uint32_t* __restrict__ plist, Bi2Lists pl, DevState* __restrict__ st,
                                                                      uint32_t* __restrict__ bitmap, const uint32_t* __restrict__ cls /* order 2: the head windows; else nullptr */,
                                                                      const uint32_t* __restrict__ surv, const uint32_t* __restrict__ headsurv) {
    if (st->done) return;
    extern __shared__ uint32_t bmL[];  // (1 << pshift) / 32 words
    __shared__ uint32_t        hsL[kBi2HeadN / 32], redL[kBi2BmThreads / kWave];
    const uint32_t b = blockIdx.x, start = b << pl.pshift;
    if (start >= npos) return;
    const uint32_t size   = min(1u << pl.pshift, npos - start);
    const uint32_t nwords = (size + 31) / 32;
    for (uint32_t w = threadIdx.x; w < nwords; w += kBi2BmThreads) bmL[w] = 0;
    if (cls != nullptr && threadIdx.x < kBi2HeadN / 32) hsL[threadIdx.x] = headsurv[threadIdx.x];
    __syncthreads();
    for (uint32_t x = 0; x < (uint32_t)kBi2Shards; ++x) {
        const uint32_t     l = x * kBi2Buckets + b, n = min(bs->pcur[l], pl.pcap);
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
    if (cls != nullptr) {  // head bigrams never became records: their windows are evaluated here, against the head survivor bits (cls is readable, zeros, beyond npos)
        const uint32_t sw0 = surv[0], sw1 = surv[1];
        for (uint32_t i0 = 0; i0 < size; i0 += 4 * kBi2BmThreads) {
            uint32_t c0[4], c1[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const uint32_t i = i0 + k * kBi2BmThreads + threadIdx.x;
                c0[k]            = i < size ? cls[start + i] : 0u;
                c1[k]            = i < size ? cls[start + i + 1] : 0u;
            }
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const uint32_t i = i0 + k * kBi2BmThreads + threadIdx.x;
                if (c0[k] - 1u < (uint32_t)(kBi2Head - 1) && c1[k] - 1u < (uint32_t)(kBi2Head - 1)) {
                    const uint32_t s0 = (c0[k] < 32 ? sw0 >> c0[k] : sw1 >> (c0[k] - 32)) & 1u, s1 = (c1[k] < 32 ? sw0 >> c1[k] : sw1 >> (c1[k] - 32)) & 1u;
                    const uint32_t h  = c0[k] * kBi2Head + c1[k];
                    if (s0 & s1 & (hsL[h >> 5] >> (h & 31u))) atomicOr(&bmL[i >> 5], 1u << (i & 31u));
                }
            }
        }
    }
    __syncthreads();
    uint32_t nset = 0;
    for (uint32_t w = threadIdx.x; w < nwords; w += kBi2BmThreads) {
        const uint32_t x           = bmL[w];
        bitmap[(start >> 5) + w] = x;
        nset += (uint32_t)__popc(x);
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
// order); whenever the queue holds a tile's worth it is counting-sorted by A bin in place and leaves as one run per (queue, A bin) into the block's sub-region, exactly as
// bi2_emit_kernel's tiles do. Key bits: idbits (for the survivors order n - 1 kept) + clsbits; what does not fit the record or the count kernel's 31-bit in-bin key raises
// Bi2State::overflow (the host repeats the run on the first-generation kernels for these orders).
template <bool HEAD>
__global__ __launch_bounds__(kBi2Threads, kBi2Threads / 128) void chain_emit_kernel(const uint32_t* __restrict__ cls, uint32_t npos, uint32_t n, uint32_t clsbits, uint32_t pb,
                                                                                     const Bi2State* __restrict__ prev, const uint32_t* __restrict__ plist, const uint32_t* __restrict__ pcode,
                                                                                     Bi2Lists pl, uint32_t nbuckets, const uint32_t* __restrict__ bitmap, const uint32_t* __restrict__ surv,
                                                                                     const uint32_t* __restrict__ headsurv, const uint32_t* __restrict__ headid,
                                                                                     unsigned long long* __restrict__ recsA, uint32_t region, uint32_t nsub, Bi2State* __restrict__ bs,
                                                                                     DevState* __restrict__ st) {
    if (st->done) return;
    const uint32_t kept_prev = prev->kept_bins + prev->kept_head;
    uint32_t       idbits    = 1;
    while (idbits < 32 && (1ull << idbits) < (uint64_t)kept_prev + 1) ++idbits;
    const uint32_t K = max(idbits + clsbits, 17u);
    if (threadIdx.x == 0) {
        bs->kbits   = K;
        bs->posbits = pb;
    }
    if (K > 48u || K - 8u + pb > 64u) {  // (the count kernel's in-bin key holds K - 17 + bshift <= 31 bits; the record K - 8 bits beside the position)
        if (threadIdx.x == 0) bs->overflow = 1;
        return;
    }
    __shared__ unsigned long long qL[kChQueue];
    __shared__ uint8_t            qaL[kChQueue];
    __shared__ uint32_t           histL[kBins], offL[kBins], gbaseL[kBins], wsumL[kBi2Threads / kWave];
    __shared__ uint32_t           hsL[HEAD ? kBi2HeadN / 32 : 1];
    if (HEAD) {
        if (threadIdx.x < kBi2HeadN / 32) hsL[threadIdx.x] = headsurv[threadIdx.x];
        __syncthreads();
    }
    const uint32_t           sub   = blockIdx.x % nsub;
    const uint32_t           rbase = prev->res_base;
    const unsigned long long kmask = (1ull << (K - 8)) - 1ull;
    uint32_t                 qn    = 0;  // records in the queue (block-uniform)
    uint32_t                 nadm  = 0;  // thread 0: records appended by this block
    // the queue's records leave: counting sort by A bin in place (every lane holds its entries in registers across the barrier), one reservation per (queue, A bin)
    auto flush = [&]() {
        __syncthreads();  // the appends are visible
        if (threadIdx.x < kBins) histL[threadIdx.x] = 0;
        __syncthreads();
        unsigned long long r[kChQPer];
        uint32_t           rk[kChQPer];
#pragma unroll
        for (int q = 0; q < kChQPer; ++q) {
            const uint32_t j = q * kBi2Threads + threadIdx.x;
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
        if (threadIdx.x < kBins) {
            const uint32_t h = histL[threadIdx.x];
            uint32_t       g = 0;
            if (h) {
                const uint32_t slot = sub * kBins + threadIdx.x;
                const uint32_t at   = atomicAdd(&bs->curA[slot], h);
                if (at + h > region) bs->overflow = 1;
                g = slot * region + min(at, region - min(region, h));
            }
            gbaseL[threadIdx.x] = g;
        }
#pragma unroll
        for (int q = 0; q < kChQPer; ++q) {
            if (rk[q] != kInvalid) {
                const uint32_t a = rk[q] >> 16, p = offL[a] + (rk[q] & 0xFFFFu);
                qL[p]            = r[q];
                qaL[p]           = (uint8_t)a;
            }
        }
        __syncthreads();
        for (uint32_t j = threadIdx.x; j < qn; j += kBi2Threads) {
            const uint32_t a                         = qaL[j];
            recsA[(size_t)gbaseL[a] + (j - offL[a])] = qL[j];
        }
        __syncthreads();
        qn = 0;
    };
    // appends this lane's candidates (ok[k]: key material in dn[k] / cn[k], position in ps[k]) to the queue
    auto append = [&](const bool* ok, const uint32_t* dn, const uint32_t* cn, const uint32_t* ps) {
        uint32_t cnt = 0;
#pragma unroll
        for (int k = 0; k < kChPer; ++k) cnt += ok[k] ? 1u : 0u;
        uint32_t total;
        uint32_t at = qn + bi2_block_scan<kBi2Threads>(cnt, &total, wsumL);
#pragma unroll
        for (int k = 0; k < kChPer; ++k) {
            if (ok[k]) {
                const uint64_t m = bi2_mix(((uint64_t)dn[k] << clsbits) | cn[k], K);
                qL[at]           = ((m & kmask) << pb) | ps[k];
                qaL[at]          = (uint8_t)(m >> (K - 8));
                ++at;
            }
        }
        qn += total;
        if (threadIdx.x == 0) nadm += total;
        if (qn >= (uint32_t)kChStep) flush();
    };
    // (a) the listed windows: one unit = one (shard, bucket) list of order n - 1
    const uint32_t nunits = nbuckets * (uint32_t)kBi2Shards;
    for (uint32_t u = blockIdx.x; u < nunits; u += gridDim.x) {
        const uint32_t l  = (u & (uint32_t)(kBi2Shards - 1)) * kBi2Buckets + (u >> 3);
        const uint32_t nl = min(prev->pcur[l], pl.pcap);
        const size_t   o  = (size_t)l * pl.pcap;
        static_assert(kBi2Shards == 8, "unit -> (shard, bucket)");
        for (uint32_t j0 = 0; j0 < nl; j0 += kChStep) {
            uint32_t ps[kChPer], code[kChPer], w[kChPer], cn[kChPer], dn[kChPer];
            bool     ok[kChPer];
#pragma unroll
            for (int k = 0; k < kChPer; ++k) {
                const uint32_t j = j0 + k * kBi2Threads + threadIdx.x;
                ok[k]            = j < nl;
                ps[k]            = ok[k] ? plist[o + j] : 0u;
                code[k]          = ok[k] ? pcode[o + j] : 0u;
            }
#pragma unroll
            for (int k = 0; k < kChPer; ++k) {  // the gathers: all inside the bucket's window (bitmap 16 KB, class ids 512 KB) or a cache-resident table
                w[k]  = ok[k] ? bitmap[(ps[k] + 1u) >> 5] : 0u;
                cn[k] = ok[k] ? cls[ps[k] + n - 1u] : 0u;
                dn[k] = ok[k] ? prev->binkept[code[k] >> 10] : 0u;
            }
#pragma unroll
            for (int k = 0; k < kChPer; ++k) {
                ok[k] = ok[k] && ((w[k] >> ((ps[k] + 1u) & 31u)) & 1u);
                dn[k] += code[k] & 1023u;
            }
            append(ok, dn, cn, ps);
        }
    }
    // (b) order 3: the windows whose bigram is a surviving head pair (they are on no list)
    if (HEAD) {
        const uint32_t sw0 = surv[0], sw1 = surv[1];
        const uint32_t ntiles = (npos + kChStep - 1) / kChStep;
        for (uint32_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
            uint32_t ps[kChPer], c0[kChPer], c1[kChPer], w[kChPer], cn[kChPer], dn[kChPer];
            bool     ok[kChPer];
#pragma unroll
            for (int k = 0; k < kChPer; ++k) {
                ps[k] = tile * kChStep + k * kBi2Threads + threadIdx.x;
                c0[k] = ps[k] < npos ? cls[ps[k]] : 0u;
                c1[k] = ps[k] < npos ? cls[ps[k] + 1u] : 0u;
            }
#pragma unroll
            for (int k = 0; k < kChPer; ++k) {
                ok[k] = false;
                w[k] = cn[k] = dn[k] = 0;
                if (c0[k] - 1u < (uint32_t)(kBi2Head - 1) && c1[k] - 1u < (uint32_t)(kBi2Head - 1)) {
                    const uint32_t s0 = (c0[k] < 32 ? sw0 >> c0[k] : sw1 >> (c0[k] - 32)) & 1u, s1 = (c1[k] < 32 ? sw0 >> c1[k] : sw1 >> (c1[k] - 32)) & 1u;
                    const uint32_t h  = c0[k] * kBi2Head + c1[k];
                    if (s0 & s1 & (hsL[h >> 5] >> (h & 31u))) {
                        ok[k] = true;
                        w[k]  = bitmap[(ps[k] + 1u) >> 5];
                        cn[k] = cls[ps[k] + 2u];
                        dn[k] = headid[h] - rbase;
                    }
                }
            }
#pragma unroll
            for (int k = 0; k < kChPer; ++k) ok[k] = ok[k] && ((w[k] >> ((ps[k] + 1u) & 31u)) & 1u);
            append(ok, dn, cn, ps);
        }
    }
    if (qn) flush();
    if (threadIdx.x == 0 && nadm) atomicAdd(&st->admitted, nadm);
}

}  // namespace colibri
