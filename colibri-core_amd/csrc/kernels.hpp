// kernels.hpp — the HIP kernels of the hot path (gfx950, wave64). See DESIGN.md §3 for the data layout
// and §4 for the roofline of each kernel. Host orchestration and the C ABI are in colibri_hip.hip.
#pragma once
#include "device_common.hpp"
#include "spooky_device.hpp"

namespace colibri {

// =================================================================================================
// block-level helpers (256-thread blocks = 4 waves)
// =================================================================================================
constexpr int kBlock = 256;

// exclusive scan of one u32 per thread over a 256-thread block; returns the exclusive prefix, *total = block sum
__device__ __forceinline__ uint32_t block_exclusive_scan(uint32_t v, uint32_t* total) {
    __shared__ uint32_t wave_sum[kBlock / kWave];
    const uint32_t      lane = threadIdx.x & (kWave - 1), wave = threadIdx.x / kWave;
    uint32_t            incl = v;
    for (int off = 1; off < kWave; off <<= 1) {
        const uint32_t t = __shfl_up(incl, off, kWave);
        if ((int)lane >= off) incl += t;
    }
    if (lane == kWave - 1) wave_sum[wave] = incl;
    __syncthreads();
    uint32_t base = 0, tot = 0;
    for (int w = 0; w < kBlock / kWave; ++w) {
        const uint32_t s = wave_sum[w];
        if (w < (int)wave) base += s;
        tot += s;
    }
    __syncthreads();
    *total = tot;
    return base + incl - v;
}

// =================================================================================================
// 1. tokenise: class-encoded byte stream -> token-start vector
//    replaces the byte-at-a-time line reader (reference src/pattern.cpp:483-587) and the sentence index
//    scan of IndexedCorpus::load (:1942-1958). A byte < 128 ends a position (token or delimiter); every
//    lane classifies 16 bytes with bit tricks, wave/block prefix sums give each terminator its position
//    index, and the start offset of position k+1 is written at tokstart[k+1]. Two passes over the bytes
//    (count, write) with a small scan of per-block counts in between: 2*B read + 4*(T+S) written.
// =================================================================================================
constexpr int kTokBytesPerThread = 16;
constexpr int kTokBytesPerBlock  = kBlock * kTokBytesPerThread;

// bit j set <=> byte g0+j is < 128 (a terminator) and lies inside the corpus
__device__ __forceinline__ uint32_t terminator_mask16(const uint8_t* bytes, uint64_t g0, uint64_t nbytes) {
    if (g0 >= nbytes) return 0;
    const uint4    v = *reinterpret_cast<const uint4*>(bytes + g0);  // buffer is padded to a multiple of 16
    const uint32_t w[4] = {v.x, v.y, v.z, v.w};
    uint32_t       mask = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const uint32_t m = ((~w[k]) & 0x80808080u) >> 7;  // bits 0,8,16,24
        mask |= ((m | (m >> 7) | (m >> 14) | (m >> 21)) & 0xFu) << (4 * k);
    }
    const uint64_t left = nbytes - g0;
    if (left < 16) mask &= (1u << left) - 1u;
    return mask;
}

__global__ __launch_bounds__(kBlock) void tokenise_count_kernel(const uint8_t* __restrict__ bytes, uint64_t nbytes, uint32_t* __restrict__ blockcnt) {
    const uint64_t g0 = (uint64_t)blockIdx.x * kTokBytesPerBlock + (uint64_t)threadIdx.x * kTokBytesPerThread;
    uint32_t       total;
    block_exclusive_scan((uint32_t)__popc(terminator_mask16(bytes, g0, nbytes)), &total);
    if (threadIdx.x == 0) blockcnt[blockIdx.x] = total;
}

__global__ __launch_bounds__(kBlock) void tokenise_write_kernel(const uint8_t* __restrict__ bytes, uint64_t nbytes, const uint32_t* __restrict__ blockoff,
                                                                 uint32_t* __restrict__ tokstart) {
    const uint64_t g0   = (uint64_t)blockIdx.x * kTokBytesPerBlock + (uint64_t)threadIdx.x * kTokBytesPerThread;
    uint32_t       mask = terminator_mask16(bytes, g0, nbytes);
    uint32_t       total;
    uint32_t       k = blockoff[blockIdx.x] + block_exclusive_scan((uint32_t)__popc(mask), &total);
    while (mask) {
        const int j = __builtin_ctz(mask);
        mask &= mask - 1;
        tokstart[++k] = (uint32_t)(g0 + j + 1);  // position k ends at byte g0+j, so position k+1 starts after it
    }
}

// in-place exclusive scan of n u32 values by ONE block (n = number of tokenise blocks: small)
__global__ __launch_bounds__(kBlock) void scan_small_kernel(uint32_t* __restrict__ data, uint32_t n, uint32_t* __restrict__ total_out) {
    uint32_t carry = 0;
    for (uint32_t base = 0; base < n; base += kBlock * 4) {
        uint32_t v[4], s = 0;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const uint32_t i = base + threadIdx.x * 4 + k;
            v[k]             = i < n ? data[i] : 0;
            s += v[k];
        }
        uint32_t tot;
        uint32_t ex = carry + block_exclusive_scan(s, &tot);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const uint32_t i = base + threadIdx.x * 4 + k;
            if (i < n) data[i] = ex;
            ex += v[k];
        }
        carry += tot;
    }
    if (threadIdx.x == 0 && total_out) *total_out = carry;
}

// corpus facts gathered once per upload
struct CorpusInfo {
    uint32_t npos;      // positions = tokens + delimiters
    uint32_t ndelim;    // delimiter positions
    uint32_t maxclass;  // highest class id (tokens of <= 5 bytes)
    uint32_t flags;     // kFlag*
};
constexpr uint32_t kFlagTokenTooLong = 1u;  // a token longer than 8 bytes (cannot be a 64-bit order-1 key)
constexpr uint32_t kFlagFlexClass    = 2u;  // literal class 4 {**}: the reference collapses runs of them (pattern.cpp:1043-1065)
constexpr uint32_t kFlagSkipClass    = 4u;  // literal class 3 {*}: changes skipgram validity (patternmodel.h:1440-1486)
constexpr uint32_t kFlagNonCanonical = 8u;  // a multi-byte token whose last byte is 0 (class id not canonical)

__device__ __forceinline__ bool is_delimiter(const uint8_t* bytes, const uint32_t* tokstart, uint32_t i) {
    const uint32_t a = tokstart[i];
    return (tokstart[i + 1] - a == 1u) && bytes[a] == 0;
}

// pass over positions: validation flags, max class, delimiter count per block
__global__ __launch_bounds__(kBlock) void position_info_kernel(const uint8_t* __restrict__ bytes, const uint32_t* __restrict__ tokstart, uint32_t npos,
                                                                CorpusInfo* __restrict__ info, uint32_t* __restrict__ blockcnt, uint32_t* __restrict__ cls_out) {
    const uint32_t i     = blockIdx.x * kBlock + threadIdx.x;
    uint32_t       flags = 0, cls = 0, delim = 0;
    if (i < npos) {
        const uint32_t a = tokstart[i], len = tokstart[i + 1] - a;
        const uint64_t raw = keep_bytes(ld64u(bytes + a), len);
        if (len > 8) flags |= kFlagTokenTooLong;
        if (len == 1) {
            const uint32_t b = (uint32_t)raw;
            delim            = b == 0;
            if (b == 4) flags |= kFlagFlexClass;
            if (b == 3) flags |= kFlagSkipClass;
        } else if (len <= 8 && ((raw >> (8 * (len - 1))) & 0xFF) == 0) {
            flags |= kFlagNonCanonical;
        }
        if (len <= 5) {  // little-endian base-128 (reference src/classdecoder.cpp:20-43), integer shifts instead of pow()
            uint64_t c = 0;
            for (uint32_t k = 0; k < len; ++k) c |= ((raw >> (8 * k)) & 0x7F) << (7 * k);
            cls = c > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)c;
        } else {
            cls = 0xFFFFFFFFu;
        }
    }
    if (i < npos) cls_out[i] = delim ? 0u : cls;  // class id per position (0 = sentence delimiter); used by the order-1 fast path
    uint32_t total;
    block_exclusive_scan(delim, &total);
    if (threadIdx.x == 0) blockcnt[blockIdx.x] = total;
    // wave reductions, then one atomic per wave
    uint32_t f = flags, c = cls;
    for (int off = 32; off > 0; off >>= 1) {
        f |= __shfl_down(f, off, kWave);
        c = max(c, __shfl_down(c, off, kWave));
    }
    // one atomic per wave only when it would change something: a hot word takes ~83 M atomics/s, and 1.6 M wave-level atomicMax
    // calls on one address used to make this kernel 18 ms for 100 M tokens
    if ((threadIdx.x & (kWave - 1)) == 0) {
        if (f && (__hip_atomic_load(&info->flags, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & f) != f) atomicOr(&info->flags, f);
        if (c > __hip_atomic_load(&info->maxclass, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(&info->maxclass, c);
    }
}

// ordered compaction of delimiter positions: delimpos[s] = position index of the delimiter closing sentence s
__global__ __launch_bounds__(kBlock) void delimiter_write_kernel(const uint8_t* __restrict__ bytes, const uint32_t* __restrict__ tokstart, uint32_t npos,
                                                                  const uint32_t* __restrict__ blockoff, uint32_t* __restrict__ delimpos) {
    const uint32_t i = blockIdx.x * kBlock + threadIdx.x;
    const uint32_t d = (i < npos) && is_delimiter(bytes, tokstart, i);
    uint32_t       total;
    const uint32_t k = blockoff[blockIdx.x] + block_exclusive_scan(d, &total);
    if (d) delimpos[k] = i;
}

// histogram of sentence lengths (in tokens): W_n = sum_len hist[len]*max(0,len-n+1) is finished on the host
// bins 0..65535 by length; longer sentences (the reference's token offsets, u16, wrap there; its counts do not): [65536] = how many, [65537] = the sum of their lengths,
// which is all W_n needs of them (every such sentence holds len - n + 1 windows of n < 128 tokens)
constexpr int kLenHistBins = 65538;
constexpr int kLenHistLds  = 2048;
__global__ __launch_bounds__(kBlock) void sentence_length_kernel(const uint32_t* __restrict__ delimpos, uint32_t ndelim, uint32_t npos,
                                                                  unsigned long long* __restrict__ hist) {
    __shared__ uint32_t lh[kLenHistLds];
    for (int k = threadIdx.x; k < kLenHistLds; k += kBlock) lh[k] = 0;
    __syncthreads();
    const uint32_t nsent = ndelim + ((ndelim == 0 ? npos : npos - (delimpos[ndelim - 1] + 1)) > 0 ? 1u : 0u);
    for (uint32_t s = blockIdx.x * kBlock + threadIdx.x; s < nsent; s += gridDim.x * kBlock) {
        const uint32_t begin = s == 0 ? 0 : delimpos[s - 1] + 1;
        const uint32_t end   = s < ndelim ? delimpos[s] : npos;
        uint32_t       len   = end - begin;
        if (len < (uint32_t)kLenHistLds) {
            atomicAdd(&lh[len], 1u);
        } else if (len < 65536u) {
            atomicAdd(&hist[len], 1ull);
        } else {
            atomicAdd(&hist[65536], 1ull);
            atomicAdd(&hist[65537], (unsigned long long)len);
        }
    }
    __syncthreads();
    for (int k = threadIdx.x; k < kLenHistLds; k += kBlock)
        if (lh[k]) atomicAdd(&hist[k], (unsigned long long)lh[k]);
}

// =================================================================================================
// 2. the order-n pass: scan + SpookyHash + hash-table build (THE dominant kernel)
//
//    replaces the window loop of PatternModel::train (reference include/patternmodel.h:1078-1178):
//    line.ngrams() (:1063), the look-back has() of both (n-1)-grams (:1139-1152) and add() (:1155-1161 ->
//    :2059-2073 -> PatternMap::operator[] -> valuehandler.add).
//
//    One lane per token position i (delimiters are positions). Window i of order n is admissible iff
//    windows i and i+1 of order n-1 survived the threshold: two coalesced reads of the survivor-id vector
//    replace both hash probes of the reference. Its exact 64-bit identity is the pair
//    of those two survivor ids (order 1: the token bytes), so distinct patterns can never merge; the slot
//    is chosen by the 64-bit SpookyHash of the window's bytes — the same value std::hash<Pattern> feeds the
//    reference's unordered_map (include/pattern.h:563-597) — read straight from the corpus bytes.
//
//    Heavy hitters: a block first reduces its 1024 windows in LDS (one representative lane per distinct key
//    collects the block's occurrences with LDS atomics), so the Zipf head costs one device-scope atomic per
//    block instead of one per occurrence (same-address device atomics serialise at ~11 ns each).
//    Representatives then find-or-insert: plain 16-byte probe load; CAS only on an empty slot; one
//    no-return atomicAdd of the block-local count. slot_of[i] records the slot for the resolve pass.
// =================================================================================================
#ifndef COLIBRI_COUNT_PER
#define COLIBRI_COUNT_PER 8
#endif
constexpr int kCountPer   = COLIBRI_COUNT_PER;    // positions per lane
constexpr int kCountTile  = kBlock * kCountPer;   // 1024 positions per block
constexpr int kCountLSlot = 2 * kCountTile;       // LDS election slots

__device__ __forceinline__ uint32_t table_find_or_insert(Slot* __restrict__ table, uint32_t cap, uint64_t key, uint64_t h, uint32_t pos, uint32_t add,
                                                         uint32_t* inserted, DevState* __restrict__ st) {
    uint32_t idx   = slot_of_hash(h, cap);
    uint32_t probe = 0;
    bool     won   = false;
    for (; probe < cap; ++probe) {
        const uint64_t cur = table[idx].key;
        if (cur == key) break;
        if (cur == kEmptyKey) {
            const uint64_t old = atomicCAS(reinterpret_cast<unsigned long long*>(&table[idx].key), (unsigned long long)kEmptyKey, (unsigned long long)key);
            if (old == kEmptyKey) {
                won = true;
                break;
            }
            if (old == key) break;
        }
        idx = (idx + 1 == cap) ? 0 : idx + 1;
    }
    if (probe == cap) {  // table exhausted: cannot happen while cap >= admitted windows; reported, never silent
        st->overflow = 1;
        return kInvalid;
    }
    // {count, rep} is one aligned 64-bit word: the CAS winner adds its position into the (zeroed) rep half in the same
    // no-return atomic that adds its count; everybody else adds to the count half only (counts stay < 2^32).
    unsigned long long inc = add;
    if (won) {
        inc |= (unsigned long long)pos << 32;
        *inserted = 1;
    }
    atomicAdd(reinterpret_cast<unsigned long long*>(&table[idx].count), inc);
    return idx;
}

// ---- key functors: what a position contributes at a given pass --------------------------------------------------
// order 1: the token's own bytes (<= 8, validated at upload) are the exact key
struct KeyUnigram {
    const uint8_t*  bytes;
    const uint32_t* tokstart;
    __device__ __forceinline__ bool operator()(uint32_t i, uint32_t npos, uint64_t& key, uint64_t& hash) const {
        (void)npos;
        const uint32_t a = tokstart[i], len = tokstart[i + 1] - a;
        const uint64_t raw = keep_bytes(ld64u(bytes + a), len);
        if (len == 1 && raw == 0) return false;  // delimiter
        key  = raw;
        hash = spooky64_short(bytes + a, len);
        return true;
    }
};
// 64-bit finaliser used wherever a table slot / radix bin has to be chosen from an exact 64-bit key
// owner rank of a key in a sharded pass: contiguous blocks of the 256 top-byte values of its mix, so that the candidates of one
// owner are a contiguous run of radix bins (A bin = the same top byte) and leave the sparse per-bin arrays already grouped by owner
__device__ __forceinline__ uint64_t mix64(uint64_t x);
__device__ __forceinline__ uint32_t owner_of(uint64_t key, uint32_t world) { return (uint32_t)(((mix64(key) >> 56) * world) >> 8); }
__device__ __forceinline__ uint64_t mix64(uint64_t x) {
    x ^= x >> 33;
    x *= 0xff51afd7ed558ccdULL;
    x ^= x >> 33;
    x *= 0xc4ceb9fe1a85ec53ULL;
    x ^= x >> 33;
    return x;
}
// order n >= 2: admissible iff both (n-1)-grams survived; exact key = their two survivor ids. The bin / slot comes from a mix of
// the exact key: the reference's SpookyHash of the window bytes only places a node in its unordered_map and is not observable
// in any output, while computing it per window costs two more index loads, unaligned byte loads and ~100 integer
// ops — 0.7 ms of a 12.8 ms Z100M step when it was measured (gpurun_out/mixhash.log). The bit-exact device SpookyHash stays in
// the order-1 key functor and behind colibri_hash_windows / colibri_hash_keys (known-answer tested).
struct KeyNgram {
    const uint32_t* id_prev;
    int             n;
    __device__ __forceinline__ bool operator()(uint32_t i, uint32_t npos, uint64_t& key, uint64_t& hash) const {
        if (i + 1 >= npos) return false;
        const uint32_t l = id_prev[i], r = id_prev[i + 1];
        if (l == kInvalid || r == kInvalid) return false;
        key  = ((uint64_t)l << 32) | r;
        hash = mix64(key);
        return true;
    }
};
// order 2 straight from the class ids and the survivor bitmap of order 1 (class-indexed order 1: the survivor id of a unigram IS its
// class): the same key as KeyNgram over the order-1 ids, without materialising them per position
struct KeyBigramCls {
    const uint32_t* cls;
    const uint32_t* surv;  // bit c = class c survived order 1
    __device__ __forceinline__ bool operator()(uint32_t i, uint32_t npos, uint64_t& key, uint64_t& hash) const {
        if (i + 1 >= npos) return false;
        const uint32_t c0 = cls[i], c1 = cls[i + 1];
        if (c0 == 0 || c1 == 0) return false;
        const uint32_t w0 = surv[c0 >> 5], w1 = surv[c1 >> 5];
        if (!((w0 >> (c0 & 31u)) & (w1 >> (c1 & 31u)) & 1u)) return false;
        key  = ((uint64_t)c0 << 32) | c1;
        hash = mix64(key);
        return true;
    }
};
// order 3 straight from the class ids, when three of them fit one 64-bit key (maxclass < 2^21) and order 1 ran class-indexed
// (the survivor id of a unigram IS its class): the key is exact without any order-2 survivor id, so order 2 only has to leave one
// byte per position ("the bigram starting here survived") instead of scattering 4-byte ids into a 420 MB array.
struct KeyTrigramCls {
    const uint32_t* cls;
    const uint8_t*  flag2;
    __device__ __forceinline__ bool operator()(uint32_t i, uint32_t npos, uint64_t& key, uint64_t& hash) const {
        if (i + 2 >= npos) return false;
        const uint8_t  f0 = flag2[i], f1 = flag2[i + 1];
        const uint32_t c0 = cls[i], c1 = cls[i + 1], c2 = cls[i + 2];
        if (!(f0 && f1)) return false;
        key  = (uint64_t)c0 | ((uint64_t)c1 << 21) | ((uint64_t)c2 << 42);
        hash = mix64(key);
        return true;
    }
};
// skipgram passes: the key is a pair of ids taken at two offsets from the window start. `gate`/`gate2` say which windows take
// part (exhaustive: both (n-1)-grams survived = gate[i], gate2[i+1]; indexed: the n-gram itself survived = gate[i]).
// Level >= 2 of a multi-part skipgram pairs the previous level's slot index (left, offset 0) with the next part's id.
struct KeyPair {
    const uint32_t* gate;   // may be NULL
    const uint32_t* gate2;  // may be NULL
    const uint32_t* left;
    uint32_t        offl;
    const uint32_t* right;
    uint32_t        offr;
    __device__ __forceinline__ bool operator()(uint32_t i, uint32_t npos, uint64_t& key, uint64_t& hash) const {
        if (gate != nullptr && gate[i] == kInvalid) return false;  // (NULL: the caller's list holds admitted windows only — chain_alist_kernel's)
        if (gate2 != nullptr && (i + 1 >= npos || gate2[i + 1] == kInvalid)) return false;
        const uint32_t l = left[i + offl], r = right[i + offr];
        if (l == kInvalid || r == kInvalid) return false;  // cannot happen for an admissible window; kept as a guard
        key  = ((uint64_t)l << 32) | r;
        hash = mix64(key);  // the reference's hash of a materialised skipgram is never observable; slots need any good 64-bit mix
        return true;
    }
};

// positions whose entry of `ids` is valid (one reservation per 4096 positions)
__global__ __launch_bounds__(kBlock) void list_from_ids_kernel(const uint32_t* __restrict__ ids, uint32_t npos, uint32_t* __restrict__ list, uint32_t* __restrict__ nlist) {
    __shared__ uint32_t baseL;
    constexpr int       kPer = 16;
    const uint32_t      ntiles = (npos + kBlock * kPer - 1) / (kBlock * kPer);
    // ascending inside a tile; the order of the tiles in the list is whatever the reservations give (no consumer relies on it)
    for (uint32_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const uint32_t i0 = tile * kBlock * kPer + threadIdx.x * kPer;
        uint32_t       m = 0, c = 0;
#pragma unroll
        for (int q = 0; q < kPer; ++q)
            if (i0 + q < npos && ids[i0 + q] != kInvalid) {
                m |= 1u << q;
                ++c;
            }
        uint32_t       total;
        const uint32_t excl = block_exclusive_scan(c, &total);
        if (threadIdx.x == 0) baseL = total ? atomicAdd(nlist, total) : 0;
        __syncthreads();
        uint32_t o = baseL + excl;
#pragma unroll
        for (int q = 0; q < kPer; ++q)
            if (m & (1u << q)) list[o++] = i0 + q;
        __syncthreads();
    }
}
template <class KeyFn>
__global__ __launch_bounds__(kBlock) void count_kernel(KeyFn keyfn, uint32_t* __restrict__ slot_of, Slot* __restrict__ table, DevState* __restrict__ st, uint32_t npos,
                                                        int track,  // track: bit 0 = add to st->admitted, bit 1 = add CAS wins to st->found
                                                        const uint32_t* __restrict__ list = nullptr, const uint32_t* __restrict__ nlist = nullptr) {
    // list / nlist (optional): visit only these positions instead of all npos — the passes of the higher orders, where few
    // positions can start a window. slot_of stays indexed by POSITION; the caller pre-fills it with kInvalid when a list is used.
    if (st->done) return;
    __shared__ uint64_t keyL[kCountTile];
    __shared__ uint32_t winL[kCountLSlot];
    __shared__ uint32_t cntL[kCountTile];
    __shared__ uint32_t slotL[kCountTile];
    __shared__ uint32_t redL[2][kBlock / kWave];

    const uint32_t cap    = st->cap;
    const uint32_t nitems = list != nullptr ? *nlist : npos;
    const uint32_t ntiles = (nitems + kCountTile - 1) / kCountTile;
    uint32_t       nadm = 0, nins = 0;

    for (uint32_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const uint32_t base = tile * kCountTile;
        uint64_t       key[kCountPer], hash[kCountPer];
        uint32_t       posn[kCountPer];
        bool           adm[kCountPer];
#pragma unroll
        for (int k = 0; k < kCountPer; ++k) {
            const uint32_t j = base + k * kBlock + threadIdx.x;
            posn[k]          = list != nullptr ? (j < nitems ? list[j] : 0u) : j;
        }
#pragma unroll
        for (int k = 0; k < kCountPer; ++k) {
            const uint32_t e = k * kBlock + threadIdx.x, i = posn[k];
            key[k]  = 0;
            hash[k] = 0;
            adm[k]  = (base + e < nitems) && keyfn(i, npos, key[k], hash[k]);
            cntL[e] = 0;
            if (adm[k]) {
                keyL[e]                                     = key[k];
                winL[(uint32_t)hash[k] & (kCountLSlot - 1)] = e;  // racing plain stores: any one writer wins the election
            }
        }
        __syncthreads();
        // a window that finds a DIFFERENT key in its election slot gets a second round in another slot: two frequent keys sharing a slot would
        // otherwise send every occurrence of the loser to the table on its own (same-address atomics, ~12 ns each)
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
        if (__syncthreads_or(anylost)) {
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
#pragma unroll
        for (int k = 0; k < kCountPer; ++k) {
            const uint32_t e = k * kBlock + threadIdx.x;
            if (adm[k]) {
                ++nadm;
                if (rep[k] == e) {
                    uint32_t ins = 0;
                    slotL[e]     = table_find_or_insert(table, cap, key[k], hash[k], posn[k], 1u + cntL[e], &ins, st);
                    nins += ins;
                }
            }
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < kCountPer; ++k) {
            const uint32_t e = k * kBlock + threadIdx.x;
            if (base + e < nitems) slot_of[posn[k]] = adm[k] ? slotL[rep[k]] : kInvalid;
        }
        // no barrier needed here: the next tile does not touch slotL before its second barrier
    }
    // one pair of counter atomics per BLOCK (a single address takes ~88 M device-scope atomics/s)
    for (int off = 32; off > 0; off >>= 1) {
        nadm += __shfl_down(nadm, off, kWave);
        nins += __shfl_down(nins, off, kWave);
    }
    if ((threadIdx.x & (kWave - 1)) == 0) {
        redL[0][threadIdx.x / kWave] = nadm;
        redL[1][threadIdx.x / kWave] = nins;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t a = 0, f = 0;
        for (int w = 0; w < kBlock / kWave; ++w) {
            a += redL[0][w];
            f += redL[1][w];
        }
        if (a && (track & 1)) atomicAdd(&st->admitted, a);
        if (f && (track & 2)) atomicAdd(&st->found, f);
    }
}

// indexed skipgrams: number of distinct surviving source n-grams per skipgram (= distinct skip contents, reference
// patternmodel.h:3029-3059): every surviving n-gram has exactly one representative position; that position bumps the count.
__global__ __launch_bounds__(kBlock) void skip_sources_kernel(const uint32_t* __restrict__ ids_n, const uint32_t* __restrict__ res_rep, const uint32_t* __restrict__ slot_of,
                                                               uint32_t* __restrict__ nsrc, const DevState* __restrict__ st, uint32_t npos) {
    if (st->done) return;
    for (uint32_t i = blockIdx.x * kBlock + threadIdx.x; i < npos; i += gridDim.x * kBlock) {
        const uint32_t r = ids_n[i], s = slot_of[i];
        if (r != kInvalid && s != kInvalid && res_rep[r] == i) atomicAdd(&nsrc[s], 1u);
    }
}

// the same on the radix path, where a finished pass has left the RESULT index of every window's skipgram in skip_id: one item per surviving n-gram
// of the order (results [first, first + count)); its representative position names the skipgram it fills. Then the skipgrams with too few distinct
// fillers leave the results again (flags -> scan -> gather / store) and the per-position indices follow (remap).
__global__ __launch_bounds__(kBlock) void skip_sources_results_kernel(const uint32_t* __restrict__ res_rep, uint32_t first, uint32_t count, const uint32_t* __restrict__ skip_id,
                                                                       uint32_t base, uint32_t* __restrict__ nsrc) {
    for (uint32_t r = blockIdx.x * kBlock + threadIdx.x; r < count; r += gridDim.x * kBlock) {
        const uint32_t s = skip_id[res_rep[first + r]];
        if (s != kInvalid) atomicAdd(&nsrc[s - base], 1u);
    }
}
__global__ __launch_bounds__(kBlock) void skip_keep_flags_kernel(uint32_t* __restrict__ nsrc, uint32_t n, uint32_t minsrc) {
    for (uint32_t j = blockIdx.x * kBlock + threadIdx.x; j < n; j += gridDim.x * kBlock) nsrc[j] = nsrc[j] >= minsrc ? 1u : 0u;
}
__global__ __launch_bounds__(kBlock) void skip_filter_gather_kernel(const uint32_t* __restrict__ flag, const unsigned long long* __restrict__ off, uint32_t n,
                                                                     const uint32_t* __restrict__ res_rep, const uint32_t* __restrict__ res_cnt, uint32_t base, uint32_t* __restrict__ tmp) {
    for (uint32_t j = blockIdx.x * kBlock + threadIdx.x; j < n; j += gridDim.x * kBlock)
        if (flag[j]) {
            tmp[off[j]]     = res_rep[base + j];
            tmp[n + off[j]] = res_cnt[base + j];
        }
}
__global__ __launch_bounds__(kBlock) void skip_filter_store_kernel(const uint32_t* __restrict__ tmp, uint32_t n, uint32_t kept, uint32_t* __restrict__ res_rep, uint32_t* __restrict__ res_cnt,
                                                                    uint32_t base) {
    for (uint32_t j = blockIdx.x * kBlock + threadIdx.x; j < kept; j += gridDim.x * kBlock) {
        res_rep[base + j] = tmp[j];
        res_cnt[base + j] = tmp[n + j];
    }
}
__global__ __launch_bounds__(kBlock) void skip_remap_ids_kernel(const uint32_t* __restrict__ list /* NULL: every position */, const uint32_t* __restrict__ nlist, uint32_t npos,
                                                                 uint32_t* __restrict__ ids, const uint32_t* __restrict__ flag, const unsigned long long* __restrict__ off, uint32_t base) {
    const uint32_t n = list != nullptr ? *nlist : npos;
    for (uint32_t j = blockIdx.x * kBlock + threadIdx.x; j < n; j += gridDim.x * kBlock) {
        const uint32_t p = list != nullptr ? list[j] : j, s = ids[p];
        if (s != kInvalid) ids[p] = flag[s - base] ? base + (uint32_t)off[s - base] : kInvalid;
    }
}

constexpr uint32_t kSegLogCap = 4096, kSegLogCapConst = kSegLogCap;
// The same filter with nothing of the pass known to the host (enqueued passes of an indexed model): the pass's kept skipgrams k1 = st->kept and its first result
// st->res_total are read on the device; `bound` >= k1 sizes the arrays (flags beyond k1 are zero, so a scan over `bound` entries gives the same offsets and total).
__global__ __launch_bounds__(kBlock) void skip_sources_results_dev_kernel(const uint32_t* __restrict__ res_rep, uint32_t first, uint32_t count, const uint32_t* __restrict__ skip_id,
                                                                           const DevState* __restrict__ st, uint32_t* __restrict__ nsrc, uint32_t bound) {
    if (st->done) return;
    const uint32_t base = st->res_total, lane = threadIdx.x & (kWave - 1);
    for (uint32_t r0 = blockIdx.x * kBlock; r0 < count; r0 += gridDim.x * kBlock) {  // (a uniform trip count: the rounds below vote with the whole wave)
        const uint32_t r = r0 + threadIdx.x;
        const uint32_t s = r < count ? skip_id[res_rep[first + r]] : kInvalid;
        bool           todo = s != kInvalid && s - base < bound;
        // a frequent frame ("the _ of") is filled by thousands of the order's n-grams, and consecutive results come from one bin of keys: same-address global atomics
        // serialise at ~12 ns each (0.37 ms for the 2 M trigrams of the bench corpus). Two rounds give the skipgram of the first lane still waiting one atomic for its group
#pragma unroll
        for (int round = 0; round < 2; ++round) {
            const uint64_t waiting = __ballot(todo);
            if (!waiting) break;
            const uint32_t lead = (uint32_t)__builtin_ctzll(waiting);
            const uint32_t sl   = (uint32_t)__shfl((int)s, (int)lead, kWave);
            const uint64_t grp  = __ballot(todo && s == sl);
            if (lane == lead) atomicAdd(&nsrc[sl - base], (uint32_t)__popcll(grp));
            todo = todo && s != sl;
        }
        if (todo) atomicAdd(&nsrc[s - base], 1u);
    }
}
__global__ __launch_bounds__(kBlock) void skip_keep_flags_dev_kernel(uint32_t* __restrict__ nsrc, const DevState* __restrict__ st, uint32_t minsrc, uint32_t bound) {
    if (st->done) return;
    const uint32_t n = min(st->kept, bound);
    for (uint32_t j = blockIdx.x * kBlock + threadIdx.x; j < bound; j += gridDim.x * kBlock) nsrc[j] = (j < n && nsrc[j] >= minsrc) ? 1u : 0u;
}
__global__ __launch_bounds__(kBlock) void skip_filter_gather_dev_kernel(const uint32_t* __restrict__ flag, const unsigned long long* __restrict__ off, DevState* __restrict__ st,
                                                                         const uint32_t* __restrict__ res_rep, const uint32_t* __restrict__ res_cnt, uint32_t* __restrict__ tmp, uint32_t bound) {
    if (st->done) return;
    const uint32_t n = st->kept, base = st->res_total;
    if (blockIdx.x == 0 && threadIdx.x == 0 && n > bound) st->overflow = 1;  // (the host's bound did not hold: never silently)
    for (uint32_t j = blockIdx.x * kBlock + threadIdx.x; j < min(n, bound); j += gridDim.x * kBlock)
        if (flag[j]) {
            tmp[off[j]]         = res_rep[base + j];
            tmp[bound + off[j]] = res_cnt[base + j];
        }
}
// ... and the pass's kept count becomes the filtered one (the last block to finish would do; a launch of its own keeps the order obvious)
__global__ __launch_bounds__(kBlock) void skip_filter_store_dev_kernel(const uint32_t* __restrict__ tmp, uint32_t bound, const unsigned long long* __restrict__ total, uint32_t* __restrict__ res_rep,
                                                                        uint32_t* __restrict__ res_cnt, const DevState* __restrict__ st) {
    if (st->done) return;
    const uint32_t kept = (uint32_t)*total, base = st->res_total;
    for (uint32_t j = blockIdx.x * kBlock + threadIdx.x; j < kept; j += gridDim.x * kBlock) {
        res_rep[base + j] = tmp[j];
        res_cnt[base + j] = tmp[bound + j];
    }
}
__global__ __launch_bounds__(kBlock) void skip_remap_ids_dev_kernel(const uint32_t* __restrict__ list, const uint32_t* __restrict__ nlist, uint32_t* __restrict__ ids,
                                                                     const uint32_t* __restrict__ flag, const unsigned long long* __restrict__ off, const DevState* __restrict__ st,
                                                                     uint32_t bound) {
    if (st->done) return;
    const uint32_t n = *nlist, base = st->res_total;
    for (uint32_t j = blockIdx.x * kBlock + threadIdx.x; j < n; j += gridDim.x * kBlock) {
        const uint32_t p = list[j], s = ids[p];
        if (s != kInvalid) ids[p] = (s - base < bound && flag[s - base]) ? base + (uint32_t)off[s - base] : kInvalid;
    }
}
__global__ void skip_set_kept_kernel(DevState* __restrict__ st, const unsigned long long* __restrict__ total) {
    if (st->done) return;
    st->kept = (uint32_t)*total;
}
// after the passes of one order of IndexedPatternModel::trainskipgrams: no skipgram found at this length ends the run (" None found", reference patternmodel.h:2992-2994)
__global__ void skip_order_end_kernel(DevState* __restrict__ st, const uint32_t* __restrict__ log, uint32_t first_entry, uint32_t nentries) {
    if (st->done) return;
    uint32_t found = 0;
    for (uint32_t e = first_entry; e < first_entry + nentries && e < kSegLogCapConst; ++e) found += log[4 + 5 * (size_t)e + 4];
    if (!found) st->done = 1;
}

// Skipgram passes without a host round trip each (unindexed models): the pass's counters are reset and, at its end, logged and folded into the run's state on
// the device; the host reads the log once per order. log[0] = entries so far; entry e = log[4 + 5 e ..]: first result, results, order, gap mask, distinct found.
__global__ void skip_pass_begin_kernel(DevState* __restrict__ st) {
    if (st->done) return;
    st->found = st->kept = st->admitted = st->valid = 0;
}
__global__ void skip_pass_end_kernel(DevState* __restrict__ st, uint32_t* __restrict__ log, uint32_t n, uint32_t mask) {
    if (st->done) return;
    const uint32_t e = log[0];
    if (e < kSegLogCap) {
        uint32_t* const x = log + 4 + 5 * (size_t)e;
        x[0] = st->res_total, x[1] = st->kept, x[2] = n, x[3] = mask, x[4] = st->found;
    }
    log[0] = e + 1;
    st->res_total += st->kept;
}

// table reset for the capacity the current order uses
__global__ __launch_bounds__(kBlock) void clear_table_kernel(Slot* __restrict__ table, const DevState* __restrict__ st) {
    if (st->done) return;
    const uint32_t cap = st->cap;
    const uint4    e   = {0xFFFFFFFFu, 0xFFFFFFFFu, 0u, 0u};
    for (uint32_t i = blockIdx.x * kBlock + threadIdx.x; i < cap; i += gridDim.x * kBlock) reinterpret_cast<uint4*>(table)[i] = e;
}

// =================================================================================================
// 2b. order 1, class-indexed: for a canonical class encoding the class id IS an exact 32-bit identity of a unigram, so order 1
//     needs neither hashing nor a table: a dense count array indexed by class (RCCL-all-reducible across shards), with the Zipf
//     head absorbed by a per-block LDS histogram — class ids are frequency-ranked by construction of the class encoding
//     (reference src/classencoder.cpp:220-224: most frequent word = lowest id), so "class < 8192" is the heavy-hitter set.
//     The survivor id of a unigram is its class id.
// =================================================================================================
constexpr int kPrunePer  = 16;
constexpr int kPruneTile = kBlock * kPrunePer;  // 4096 slots / classes per block iteration (prune, uni_finish, shard kernels)
constexpr int kUniHead = 8192;  // classes counted in LDS
constexpr int kUniPer  = 8;
__global__ __launch_bounds__(kBlock) void uni_count_kernel(const uint32_t* __restrict__ cls, uint32_t npos, uint32_t* __restrict__ cnt1, uint32_t* __restrict__ rep1,
                                                            DevState* __restrict__ st) {
    if (st->done) return;
    __shared__ uint32_t histL[kUniHead], fposL[kUniHead];
    __shared__ uint32_t redL[kBlock / kWave];
    for (int k = threadIdx.x; k < kUniHead; k += kBlock) {
        histL[k] = 0;
        fposL[k] = 0xFFFFFFFFu;
    }
    __syncthreads();
    // contiguous chunk per block (so that a head class gets ONE flush per block)
    const uint32_t per   = (npos + gridDim.x - 1) / gridDim.x;
    const uint32_t begin = blockIdx.x * per, end = min(npos, begin + per);
    uint32_t       nadm  = 0;
    for (uint32_t i0 = begin; i0 < end; i0 += kBlock * kUniPer) {
#pragma unroll
        for (int q = 0; q < kUniPer; ++q) {
            const uint32_t i = i0 + q * kBlock + threadIdx.x;
            if (i < end) {
                const uint32_t c = cls[i];
                if (c != 0) {
                    ++nadm;
                    if (c < (uint32_t)kUniHead) {
                        atomicAdd(&histL[c], 1u);
                        atomicMin(&fposL[c], i);
                    } else if (atomicAdd(&cnt1[c], 1u) == 0u) {
                        rep1[c] = i;  // exactly one increment sees 0: it names the representative position
                    }
                }
            }
        }
    }
    __syncthreads();
    for (int k = threadIdx.x; k < kUniHead; k += kBlock) {
        const uint32_t h = histL[k];
        if (h && atomicAdd(&cnt1[k], h) == 0u) rep1[k] = fposL[k];
    }
    for (int off = 32; off > 0; off >>= 1) nadm += __shfl_down(nadm, off, kWave);
    if ((threadIdx.x & (kWave - 1)) == 0) redL[threadIdx.x / kWave] = nadm;
    __syncthreads();
    if (threadIdx.x == 0) {
        const uint32_t a = redL[0] + redL[1] + redL[2] + redL[3];
        if (a) atomicAdd(&st->admitted, a);
    }
}
// ---- order 1 without per-token global atomics (plain single-device runs) -------------------------------------------------------------
// Global atomics execute memory-side at a fixed ~27 G/s on MI355X whatever their locality (tools/atomics_scope_bench.hip), so the
// third of a Zipf corpus that lies outside the LDS head costs uni_count_kernel 1.2 ms per 100 M tokens. Here the tail is instead
// partitioned by class range (256 bins of 2^shift consecutive classes, shift <= 14 so that a bin's counters fit in LDS) as 2-byte
// in-bin offsets, and every bin is then counted privately in LDS:
//   uni_head_kernel       head histogram (classes < kUniHead) in LDS + tokens per tail bin           reads cls
//   uni_offsets_kernel    exclusive scan of the bin sizes
//   uni_partition_kernel  tile-local counting sort of the tail tokens -> one run per (tile, bin)      reads cls, writes 2 B per tail token
//   uni_tail_count_kernel one LDS histogram per (bin, slice)                                         reads 2 B per tail token
// No representative positions are recorded: the key bytes of a unigram of a canonical encoding are the varint of its class
// (export_*_kernel with kMaskFromClass).
constexpr int kUniBins     = 256;
constexpr int kUniTile     = 16384;  // tokens per partition tile
constexpr int kUniTilePer  = kUniTile / kBlock;
constexpr int kUniSlices   = 8;      // blocks per tail bin (a bin is streamed by up to 8 blocks)
constexpr uint32_t kUniSliceMin = 65536;  // ... but a slice is never smaller than this
#ifndef COLIBRI_UNI_SUB
#define COLIBRI_UNI_SUB 4
#endif
#ifndef COLIBRI_UNI_CURPAD
#define COLIBRI_UNI_CURPAD 32
#endif
// The one-pass form reserves room with one returning atomic per (tile, bin). Atomics are executed memory-side and a LINE of cursors serialises like one cursor
// (docs/history.md: ~12 ns each): 256 adjacent cursors = 16 lines took 12 207 tiles x 256 reservations — 43 % of the kernel's time by its phase clocks (round 6). Hence
// kUniSub runs per bin (run = block index mod kUniSub) and every cursor on a line of its own (kUniCurPad words apart).
constexpr int kUniSub = COLIBRI_UNI_SUB, kUniCurPad = COLIBRI_UNI_CURPAD;
struct UniState {
    uint32_t hist[kUniBins];     // tail tokens per bin
    uint32_t off[kUniBins + 1];  // exclusive scan
    uint32_t cur[kUniBins];      // partition cursors of the two-pass form
    uint32_t overflow;           // the one-pass form: a run outgrew its room
    uint32_t pad_[kUniCurPad];
    uint32_t cur1[kUniBins * kUniSub * kUniCurPad];  // the one-pass form: tokens of run (bin, sub) at [(bin * kUniSub + sub) * kUniCurPad]
};
// exclusive scan of 256 LDS values by the first 256 threads of a block of any size (>= 256); every thread of the block must call it
__device__ __forceinline__ uint32_t bi2_uni_scan256(const uint32_t* inL, uint32_t* outL, uint32_t* wsumL) {
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
template <bool BINS = true>  // BINS = false: the head histogram and the token count only (the one-pass tail of round 5 needs no bin sizes)
__global__ __launch_bounds__(kBlock) void uni_head_kernel(const uint32_t* __restrict__ cls, uint32_t npos, uint32_t shift, uint32_t* __restrict__ head_rows /*[gridDim.x][kUniHead]*/,
                                                           UniState* __restrict__ us, DevState* __restrict__ st) {
    if (st->done) return;
    __shared__ uint32_t histL[kUniHead], binL[kUniBins];
    __shared__ uint32_t redL[kBlock / kWave];
    for (int k = threadIdx.x; k < kUniHead; k += kBlock) histL[k] = 0;
    binL[threadIdx.x] = 0;
    __syncthreads();
    const uint32_t per   = (npos + gridDim.x - 1) / gridDim.x;
    const uint32_t begin = blockIdx.x * per, end = min(npos, begin + per);
    uint32_t       nadm  = 0;
    for (uint32_t i0 = begin; i0 < end; i0 += kBlock * kUniPer) {
        uint32_t c[kUniPer];  // all loads of the batch are issued before the first LDS atomic (the compiler will not hoist them itself)
#pragma unroll
        for (int q = 0; q < kUniPer; ++q) {
            const uint32_t i = i0 + q * kBlock + threadIdx.x;
            c[q]             = (i < end) ? cls[i] : 0u;
        }
#pragma unroll
        for (int q = 0; q < kUniPer; ++q) {
            if (c[q] != 0) {
                ++nadm;
                if (c[q] < (uint32_t)kUniHead)
                    atomicAdd(&histL[c[q]], 1u);
                else if (BINS)
                    atomicAdd(&binL[c[q] >> shift], 1u);
            }
        }
    }
    __syncthreads();
    // the block's head histogram leaves as one plain row (512 blocks x 8192 flush atomics would cost 0.16 ms at the memory-side atomic rate)
    for (int k = threadIdx.x; k < kUniHead; k += kBlock) head_rows[(size_t)blockIdx.x * kUniHead + k] = histL[k];
    if (BINS && binL[threadIdx.x]) atomicAdd(&us->hist[threadIdx.x], binL[threadIdx.x]);
    for (int off = 32; off > 0; off >>= 1) nadm += __shfl_down(nadm, off, kWave);
    if ((threadIdx.x & (kWave - 1)) == 0) redL[threadIdx.x / kWave] = nadm;
    __syncthreads();
    if (threadIdx.x == 0) {
        const uint32_t a = redL[0] + redL[1] + redL[2] + redL[3];
        if (a) atomicAdd(&st->admitted, a);
    }
}
// column sums of the head rows -> cnt1[0 .. kUniHead): block (x, y) adds rows y, y + gridDim.y, ... of classes [x * 256, x * 256 + 256)
__global__ __launch_bounds__(kBlock) void uni_head_reduce_kernel(const uint32_t* __restrict__ head_rows, uint32_t nrows, uint32_t* __restrict__ cnt1, uint32_t nclasses,
                                                                  const DevState* __restrict__ st) {
    if (st->done) return;
    const uint32_t k = blockIdx.x * kBlock + threadIdx.x;
    uint32_t       s = 0;
    for (uint32_t r = blockIdx.y; r < nrows; r += gridDim.y) s += head_rows[(size_t)r * kUniHead + k];
    if (s && k < nclasses) atomicAdd(&cnt1[k], s);
}
__global__ __launch_bounds__(kBlock) void uni_offsets_kernel(UniState* __restrict__ us) {
    uint32_t       tot;
    const uint32_t o     = block_exclusive_scan(us->hist[threadIdx.x], &tot);
    us->off[threadIdx.x] = o;
    us->cur[threadIdx.x] = 0;
    if (threadIdx.x == 0) us->off[kUniBins] = tot;
}
__global__ __launch_bounds__(kBlock) void uni_partition_kernel(const uint32_t* __restrict__ cls, uint32_t npos, uint32_t shift, UniState* __restrict__ us,
                                                                uint16_t* __restrict__ tail, const DevState* __restrict__ st) {
    if (st->done) return;
    __shared__ uint16_t stageL[kUniTile];
    __shared__ uint8_t  sbinL[kUniTile];
    __shared__ uint32_t cntL[kUniBins], offL[kUniBins + 1], curL[kUniBins], gbaseL[kUniBins];
    const uint32_t lowmask = (1u << shift) - 1u;
    const uint32_t ntiles  = (npos + kUniTile - 1) / kUniTile;
    for (uint32_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const uint32_t base = tile * kUniTile;
        uint32_t       c[kUniTilePer];
        cntL[threadIdx.x] = 0;
        curL[threadIdx.x] = 0;
        __syncthreads();
#pragma unroll
        for (int q = 0; q < kUniTilePer; ++q) {
            const uint32_t i = base + q * kBlock + threadIdx.x;
            c[q]             = (i < npos) ? cls[i] : 0u;
        }
#pragma unroll
        for (int q = 0; q < kUniTilePer; ++q)
            if (c[q] >= (uint32_t)kUniHead) atomicAdd(&cntL[c[q] >> shift], 1u);
        __syncthreads();
        {
            uint32_t       tot;
            const uint32_t h  = cntL[threadIdx.x];
            offL[threadIdx.x] = block_exclusive_scan(h, &tot);
            if (threadIdx.x == 0) offL[kUniBins] = tot;
            gbaseL[threadIdx.x] = h ? us->off[threadIdx.x] + atomicAdd(&us->cur[threadIdx.x], h) : 0u;  // one reservation per (tile, bin)
        }
        __syncthreads();
#pragma unroll
        for (int q = 0; q < kUniTilePer; ++q) {
            if (c[q] >= (uint32_t)kUniHead) {
                const uint32_t b = c[q] >> shift, slot = offL[b] + atomicAdd(&curL[b], 1u);
                stageL[slot]     = (uint16_t)(c[q] & lowmask);
                sbinL[slot]      = (uint8_t)b;
            }
        }
        __syncthreads();
        const uint32_t total = offL[kUniBins];
        for (uint32_t j = threadIdx.x; j < total; j += kBlock) {
            const uint32_t b               = sbinL[j];
            tail[gbaseL[b] + (j - offL[b])] = stageL[j];
        }
        __syncthreads();
    }
}
// block (bin, slice): LDS histogram of 2^shift classes (dynamic LDS: 4 << shift bytes)
__global__ __launch_bounds__(kBlock) void uni_tail_count_kernel(const uint16_t* __restrict__ tail, const UniState* __restrict__ us, uint32_t shift, uint32_t* __restrict__ cnt1,
                                                                 uint32_t nclasses, const DevState* __restrict__ st) {
    if (st->done) return;
    extern __shared__ uint32_t uniHistL[];
    const uint32_t bin = blockIdx.x / kUniSlices, slice = blockIdx.x % kUniSlices;
    const uint32_t b0 = us->off[bin], n = us->off[bin + 1] - b0;
    if (n == 0) return;
    uint32_t nsl = (n + kUniSliceMin - 1) / kUniSliceMin;
    if (nsl > (uint32_t)kUniSlices) nsl = kUniSlices;
    if (slice >= nsl) return;
    const uint32_t per = (n + nsl - 1) / nsl, begin = b0 + slice * per, end = min(b0 + n, begin + per);
    const uint32_t width = 1u << shift;
    for (uint32_t k = threadIdx.x; k < width; k += kBlock) uniHistL[k] = 0;
    __syncthreads();
    // 8 tokens (16 bytes) per load: the body of the slice is read as uint4, the unaligned ends token by token
    const uint32_t vbegin = min(end, (begin + 7u) & ~7u), vend = max(vbegin, end & ~7u);
    if (threadIdx.x < vbegin - begin) atomicAdd(&uniHistL[tail[begin + threadIdx.x]], 1u);
    if (threadIdx.x < end - vend) atomicAdd(&uniHistL[tail[vend + threadIdx.x]], 1u);
    const uint4* const v = reinterpret_cast<const uint4*>(tail + vbegin);
    const uint32_t     nv = (vend - vbegin) >> 3;
    for (uint32_t j = threadIdx.x; j < nv; j += kBlock) {
        const uint4 x = v[j];
        atomicAdd(&uniHistL[x.x & 0xFFFFu], 1u);
        atomicAdd(&uniHistL[x.x >> 16], 1u);
        atomicAdd(&uniHistL[x.y & 0xFFFFu], 1u);
        atomicAdd(&uniHistL[x.y >> 16], 1u);
        atomicAdd(&uniHistL[x.z & 0xFFFFu], 1u);
        atomicAdd(&uniHistL[x.z >> 16], 1u);
        atomicAdd(&uniHistL[x.w & 0xFFFFu], 1u);
        atomicAdd(&uniHistL[x.w >> 16], 1u);
    }
    __syncthreads();
    const uint32_t cbase = bin << shift;
    for (uint32_t k = threadIdx.x; k < width; k += kBlock) {
        const uint32_t h = uniHistL[k];
        if (h && cbase + k < nclasses) atomicAdd(&cnt1[cbase + k], h);
    }
}
// ---- order 1 in ONE pass over the class ids (round 5) ---------------------------------------------------------------------------------------------------------------
// The two-pass form above reads the corpus twice: once for the head histogram and the sizes of the 256 tail bins (a bin's place in the tail array needs every bin's size),
// once to partition. Here a tail bin is not a class RANGE but every 256th GROUP of 16 consecutive classes — bin = (c >> 4) & 255, 2-byte offset inside it
// ((c >> 12) << 4) | (c & 15) —: under any class distribution that is not built against it the bins fill evenly (the frequent classes, which a range puts into one
// bin, are dealt round-robin), so every bin gets the same fixed room (`cap`, 1.5 x its even share + slack) and nothing has to be known before the one pass that counts the
// head in LDS, sorts a tile's tail tokens by bin and appends the runs. A bin's counters (16 x (classes >> 12) <= 16 384) fit LDS for class ids below 2^22, and a group of 16
// classes is one 64-byte line of the count array. A bin that does outgrow its room raises UniState::overflow: uni_tail_count1_kernel then adds nothing and
// uni_tail_atomics_kernel (one more launch, ~5 us when idle) counts the tail with global atomics — slow, exact, and only for corpora built to fill one bin.
constexpr int kUni1Threads = 1024, kUni1Per = 8, kUni1Tile = kUni1Threads * kUni1Per;  // 8192 positions per tile: 60 KB of LDS, two blocks per CU
// HEAD = false: the tail only (no head histogram). Measured and not used: uni_head_kernel<false> on a second stream beside it — 0.48 ms per 10^8 tokens against 0.43
// fused and 0.45 for round 4's two passes.
template <bool HEAD>
__global__ __launch_bounds__(kUni1Threads, kUni1Threads / 128) void uni_onepass_kernel(const uint32_t* __restrict__ cls, uint32_t npos, uint32_t cap /* per run */,
                                                                                        uint32_t* __restrict__ head_rows /*[gridDim.x][kUniHead]*/, UniState* __restrict__ us,
                                                                                        uint16_t* __restrict__ tail /*[kUniBins][kUniSub][cap] (+ one tile of slack) */, DevState* __restrict__ st) {
    if (st->done) return;
    __shared__ uint32_t histL[HEAD ? kUniHead : 1];
    __shared__ uint16_t stageL[kUni1Tile];
    __shared__ uint8_t  sbinL[kUni1Tile];
    __shared__ uint32_t cntL[kUniBins], offL[kUniBins], curL[kUniBins], gbaseL[kUniBins], wsumL[4], redL[kUni1Threads / kWave], totL;
    if (HEAD)
        for (int k = threadIdx.x; k < kUniHead; k += kUni1Threads) histL[k] = 0;
    const uint32_t ntiles = (npos + kUni1Tile - 1) / kUni1Tile;
    uint32_t       nadm   = 0;
    uint32_t       c[kUni1Per];
    auto           load_tile = [&](uint32_t tile) {
#pragma unroll
        for (int q = 0; q < kUni1Per; ++q) {
            const uint32_t i = tile * kUni1Tile + q * kUni1Threads + threadIdx.x;
            c[q]             = (tile < ntiles && i < npos) ? cls[i] : 0u;
        }
    };
    // d: the tile being counted; c: the next one, in flight. The hand-over d = c (a wait for c's loads) sits BEFORE a tile's write-out, not behind it: the memory counter
    // is in order, so behind the write-out it also waited for the tile's 2-byte stores to drain (a third of the kernel by its phase clocks, round 6)
    uint32_t d[kUni1Per];
    load_tile(blockIdx.x);
#pragma unroll
    for (int q = 0; q < kUni1Per; ++q) d[q] = c[q];
    load_tile(blockIdx.x + gridDim.x);
    __syncthreads();
    KP_INIT(2);
    for (uint32_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        if (threadIdx.x < (uint32_t)kUniBins) {
            cntL[threadIdx.x] = 0;
            curL[threadIdx.x] = 0;
        }
        __syncthreads();
        KP(0);
#pragma unroll
        for (int q = 0; q < kUni1Per; ++q) {
            if (d[q] != 0) {
                ++nadm;
                if (d[q] < (uint32_t)kUniHead) {
                    if (HEAD) atomicAdd(&histL[d[q]], 1u);
                } else {
                    atomicAdd(&cntL[(d[q] >> 4) & 255u], 1u);
                }
            }
        }
        __syncthreads();
        KP(1);
        // one reservation per (tile, bin), on the cursor of this block's run. The atomic runs memory-side (~2 us): its answer is first needed by the write-out, so it
        // travels while the tile is staged (round 6: the wait sat before the staging and was a third of the kernel)
        uint32_t rs_at = 0, rs_h = 0;
        const uint32_t rs_run = threadIdx.x * (uint32_t)kUniSub + blockIdx.x % (uint32_t)kUniSub;
        {
            const uint32_t tot = bi2_uni_scan256(cntL, offL, wsumL);
            if (threadIdx.x == 0) totL = tot;
            if (threadIdx.x < (uint32_t)kUniBins) {
                rs_h = cntL[threadIdx.x];
                if (rs_h) rs_at = atomicAdd(&us->cur1[rs_run * (uint32_t)kUniCurPad], rs_h);
            }
        }
        KP(2);
#pragma unroll
        for (int q = 0; q < kUni1Per; ++q) {
            if (d[q] >= (uint32_t)kUniHead) {
                const uint32_t b = (d[q] >> 4) & 255u, slot = offL[b] + atomicAdd(&curL[b], 1u);
                stageL[slot]     = (uint16_t)(((d[q] >> 12) << 4) | (d[q] & 15u));
                sbinL[slot]      = (uint8_t)b;
            }
        }
#pragma unroll
        for (int q = 0; q < kUni1Per; ++q) d[q] = c[q];  // the next tile's class ids (asked for a tile ago) ...
        load_tile(tile + 2 * gridDim.x);                  // ... and the one after it travels while that one is counted
        if (threadIdx.x < (uint32_t)kUniBins) {
            uint32_t g = 0;
            if (rs_h) {
                if (rs_at + rs_h > cap) us->overflow = 1;
                g = rs_run * cap + min(rs_at, cap - min(cap, rs_h));  // (h > cap: the run's h entries reach into the next run's room, or — the last run — into the slack
            }                                                          // behind the array; the counts are done again with atomics either way)
            gbaseL[threadIdx.x] = g;
        }
        __syncthreads();
        KP(3);
        const uint32_t total = totL;
        for (uint32_t j = threadIdx.x; j < total; j += kUni1Threads) {
            const uint32_t b                          = sbinL[j];
            tail[(size_t)gbaseL[b] + (j - offL[b])] = stageL[j];
        }
        __syncthreads();
        KP(4);
    }
    KP_DONE();
    if (!HEAD) return;
    // the block's head histogram leaves as one plain row (flush atomics would run at the memory-side atomic rate)
    for (int k = threadIdx.x; k < kUniHead; k += kUni1Threads) head_rows[(size_t)blockIdx.x * kUniHead + k] = histL[k];
    for (int off = 32; off > 0; off >>= 1) nadm += __shfl_down(nadm, off, kWave);
    if ((threadIdx.x & (kWave - 1)) == 0) redL[threadIdx.x / kWave] = nadm;
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t a = 0;
        for (int w = 0; w < kUni1Threads / kWave; ++w) a += redL[w];
        if (a) atomicAdd(&st->admitted, a);
    }
}
// block (bin, slice): LDS histogram of the bin's 16 x nrows classes (dynamic LDS: 64 x nrows bytes); class of offset k: ((k >> 4) << 12) | (bin << 4) | (k & 15)
__global__ __launch_bounds__(kBlock) void uni_tail_count1_kernel(const uint16_t* __restrict__ tail, const UniState* __restrict__ us, uint32_t cap, uint32_t nrows,
                                                                  uint32_t* __restrict__ cnt1, uint32_t nclasses, const DevState* __restrict__ st) {
    if (st->done || us->overflow) return;
    extern __shared__ uint32_t uniHistL[];
    static_assert(kUniSlices % kUniSub == 0, "a run is read by kUniSlices / kUniSub blocks at most");
    const uint32_t bin = blockIdx.x / kUniSlices, slice = blockIdx.x % kUniSlices, run = bin * (uint32_t)kUniSub + slice % (uint32_t)kUniSub, piece = slice / (uint32_t)kUniSub;
    const uint32_t n   = min(us->cur1[run * (uint32_t)kUniCurPad], cap);
    if (n == 0) return;
    uint32_t nsl = (n + kUniSliceMin - 1) / kUniSliceMin;
    if (nsl > (uint32_t)(kUniSlices / kUniSub)) nsl = kUniSlices / kUniSub;
    if (piece >= nsl) return;
    const uint32_t b0 = run * cap, per = (((n + nsl - 1) / nsl) + 7u) & ~7u, begin = min(b0 + n, b0 + piece * per), end = min(b0 + n, begin + per);
    if (begin >= end) return;
    const uint32_t width = nrows << 4;
    for (uint32_t k = threadIdx.x; k < width; k += kBlock) uniHistL[k] = 0;
    __syncthreads();
    // 8 tokens (16 bytes) per load: the body of the slice is read as uint4, the unaligned ends token by token (cap is a multiple of 8)
    const uint32_t vbegin = min(end, (begin + 7u) & ~7u), vend = max(vbegin, end & ~7u);
    if (threadIdx.x < vbegin - begin) atomicAdd(&uniHistL[tail[begin + threadIdx.x]], 1u);
    if (threadIdx.x < end - vend) atomicAdd(&uniHistL[tail[vend + threadIdx.x]], 1u);
    const uint4* const v  = reinterpret_cast<const uint4*>(tail + vbegin);
    const uint32_t     nv = (vend - vbegin) >> 3;
    for (uint32_t j = threadIdx.x; j < nv; j += kBlock) {
        const uint4 x = v[j];
        atomicAdd(&uniHistL[x.x & 0xFFFFu], 1u);
        atomicAdd(&uniHistL[x.x >> 16], 1u);
        atomicAdd(&uniHistL[x.y & 0xFFFFu], 1u);
        atomicAdd(&uniHistL[x.y >> 16], 1u);
        atomicAdd(&uniHistL[x.z & 0xFFFFu], 1u);
        atomicAdd(&uniHistL[x.z >> 16], 1u);
        atomicAdd(&uniHistL[x.w & 0xFFFFu], 1u);
        atomicAdd(&uniHistL[x.w >> 16], 1u);
    }
    __syncthreads();
    for (uint32_t k = threadIdx.x; k < width; k += kBlock) {
        const uint32_t h = uniHistL[k], cl = ((k >> 4) << 12) | (bin << 4) | (k & 15u);
        if (h && cl < nclasses) atomicAdd(&cnt1[cl], h);
    }
}
// a tail bin outgrew its room (a corpus built for it): the tail classes once more, with one global atomic per token
__global__ __launch_bounds__(kBlock) void uni_tail_atomics_kernel(const uint32_t* __restrict__ cls, uint32_t npos, const UniState* __restrict__ us, uint32_t* __restrict__ cnt1,
                                                                   const DevState* __restrict__ st) {
    if (st->done || !us->overflow) return;
    for (uint32_t i = blockIdx.x * kBlock + threadIdx.x; i < npos; i += gridDim.x * kBlock) {
        const uint32_t c = cls[i];
        if (c >= (uint32_t)kUniHead) atomicAdd(&cnt1[c], 1u);
    }
}
// classes -> result list (threshold), found / kept
__global__ __launch_bounds__(kBlock) void uni_finish_kernel(const uint32_t* __restrict__ cnt1, const uint32_t* __restrict__ rep1, uint32_t nclasses, uint32_t threshold,
                                                             DevState* __restrict__ st, uint32_t* __restrict__ res_rep, uint32_t* __restrict__ res_cnt, uint32_t res_cap,
                                                             uint16_t* __restrict__ surv16 = nullptr /* optional: bit c = class c survives (16 classes per lane = one half word) */,
                                                             bool count_valid = false /* also st->valid += occurrences of the surviving classes (when no id pass follows) */,
                                                             uint32_t* __restrict__ resid = nullptr /* optional: result index of every class (kInvalid if it did not survive) */,
                                                             uint32_t thr_ids = 0 /* MINTOKENS_UNIGRAMS (>= threshold): what a word needs to be part of longer patterns (bitmap, resid, valid) */) {
    if (thr_ids < threshold) thr_ids = threshold;
    if (st->done) return;
    __shared__ uint32_t baseL, redL[kBlock / kWave];
    const uint32_t      res_base = st->res_total;
    const uint32_t      ntiles   = (nclasses + kPruneTile - 1) / kPruneTile;
    uint32_t            nfound   = 0;
    unsigned long long  nvalid   = 0;
    for (uint32_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const uint32_t c0 = tile * kPruneTile + threadIdx.x * kPrunePer;
        uint32_t       v[kPrunePer], k = 0;
#pragma unroll
        for (int q = 0; q < kPrunePer; ++q) {
            v[q] = (c0 + q < nclasses) ? cnt1[c0 + q] : 0u;
            nfound += v[q] != 0;
            k += v[q] >= threshold;
            if (v[q] >= thr_ids) nvalid += v[q];
        }
        if (surv16 != nullptr && c0 < nclasses) {
            static_assert(kPrunePer == 16, "one 16-bit store per lane");
            uint32_t bits = 0;
#pragma unroll
            for (int q = 0; q < kPrunePer; ++q) bits |= (uint32_t)(v[q] >= thr_ids) << q;
            surv16[c0 >> 4] = (uint16_t)bits;
        }
        uint32_t       total;
        const uint32_t excl = block_exclusive_scan(k, &total);
        if (threadIdx.x == 0) baseL = total ? atomicAdd(&st->kept, total) : 0;
        __syncthreads();
        uint32_t r = res_base + baseL + excl;
#pragma unroll
        for (int q = 0; q < kPrunePer; ++q) {
            if (resid != nullptr && c0 + q < nclasses) resid[c0 + q] = (v[q] >= threshold && v[q] >= thr_ids) ? r : kInvalid;
            if (v[q] >= threshold) {
                if (r < res_cap) {
                    res_rep[r] = rep1 != nullptr ? rep1[c0 + q] : c0 + q;  // no representative recorded: the class itself (kMaskFromClass export)
                    res_cnt[r] = v[q];
                } else {
                    st->overflow = 1;
                }
                ++r;
            }
        }
        __syncthreads();
    }
    for (int off = 32; off > 0; off >>= 1) nfound += __shfl_down(nfound, off, kWave);
    if ((threadIdx.x & (kWave - 1)) == 0) redL[threadIdx.x / kWave] = nfound;
    __syncthreads();
    if (threadIdx.x == 0) {
        const uint32_t f = redL[0] + redL[1] + redL[2] + redL[3];
        if (f) atomicAdd(&st->found, f);
    }
    if (count_valid) {
        for (int off = 32; off > 0; off >>= 1) nvalid += __shfl_down(nvalid, off, kWave);
        if ((threadIdx.x & (kWave - 1)) == 0 && nvalid) atomicAdd(&st->valid, (uint32_t)nvalid);
    }
}
// survivor id per position = the class id of a surviving unigram
__global__ __launch_bounds__(kBlock) void uni_ids_kernel(const uint32_t* __restrict__ cls, const uint32_t* __restrict__ cnt1, uint32_t threshold, uint32_t* __restrict__ ids,
                                                          DevState* __restrict__ st, uint32_t npos) {
    if (st->done) return;
    uint32_t nvalid = 0;
    constexpr int kPer = 8;  // loads, then gathers, then stores: eight independent chains per lane
    for (uint32_t i0 = blockIdx.x * (kBlock * kPer); i0 < npos; i0 += gridDim.x * (kBlock * kPer)) {
        uint32_t c[kPer], n1[kPer];
#pragma unroll
        for (int q = 0; q < kPer; ++q) {
            const uint32_t i = i0 + q * kBlock + threadIdx.x;
            c[q]             = (i < npos) ? cls[i] : 0u;
        }
#pragma unroll
        for (int q = 0; q < kPer; ++q) n1[q] = c[q] ? cnt1[c[q]] : 0u;
#pragma unroll
        for (int q = 0; q < kPer; ++q) {
            const uint32_t i = i0 + q * kBlock + threadIdx.x;
            if (i < npos) {
                const uint32_t id = n1[q] >= threshold ? c[q] : kInvalid;
                nvalid += id != kInvalid;
                ids[i] = id;
            }
        }
    }
    __shared__ uint32_t redL[kBlock / kWave];
    for (int off = 32; off > 0; off >>= 1) nvalid += __shfl_down(nvalid, off, kWave);
    if ((threadIdx.x & (kWave - 1)) == 0) redL[threadIdx.x / kWave] = nvalid;
    __syncthreads();
    if (threadIdx.x == 0) {
        const uint32_t v = redL[0] + redL[1] + redL[2] + redL[3];
        if (v) atomicAdd(&st->valid, v);
    }
}

// result index per position (the per-pass modes use result indices as survivor ids: they double as the pattern number of the forward index)
__global__ __launch_bounds__(kBlock) void uni_resid_ids_kernel(const uint32_t* __restrict__ cls, const uint32_t* __restrict__ resid, uint32_t* __restrict__ ids,
                                                                DevState* __restrict__ st, uint32_t npos) {
    if (st->done) return;
    uint32_t      nvalid = 0;
    constexpr int kPer   = 8;
    for (uint32_t i0 = blockIdx.x * (kBlock * kPer); i0 < npos; i0 += gridDim.x * (kBlock * kPer)) {
        uint32_t c[kPer], r[kPer];
#pragma unroll
        for (int q = 0; q < kPer; ++q) {
            const uint32_t i = i0 + q * kBlock + threadIdx.x;
            c[q]             = (i < npos) ? cls[i] : 0u;
        }
#pragma unroll
        for (int q = 0; q < kPer; ++q) r[q] = c[q] ? resid[c[q]] : kInvalid;
#pragma unroll
        for (int q = 0; q < kPer; ++q) {
            const uint32_t i = i0 + q * kBlock + threadIdx.x;
            if (i < npos) {
                ids[i] = r[q];
                nvalid += r[q] != kInvalid;
            }
        }
    }
    __shared__ uint32_t redL[kBlock / kWave];
    for (int off = 32; off > 0; off >>= 1) nvalid += __shfl_down(nvalid, off, kWave);
    if ((threadIdx.x & (kWave - 1)) == 0) redL[threadIdx.x / kWave] = nvalid;
    __syncthreads();
    if (threadIdx.x == 0) {
        const uint32_t v = redL[0] + redL[1] + redL[2] + redL[3];
        if (v) atomicAdd(&st->valid, v);
    }
}
// MINTOKENS_UNIGRAMS > MINTOKENS on the table path: the unigrams below the word threshold stay in the model but take no part in longer
// patterns (reference patternmodel.h:1093-1104) — their ids (result indices) are withdrawn from the per-position array
__global__ __launch_bounds__(kBlock) void ids_min_count_kernel(uint32_t* __restrict__ ids, const uint32_t* __restrict__ res_cnt, uint32_t wthr, uint32_t npos) {
    for (uint32_t i = blockIdx.x * kBlock + threadIdx.x; i < npos; i += gridDim.x * kBlock) {
        const uint32_t r = ids[i];
        if (r != kInvalid && res_cnt[r] < wthr) ids[i] = kInvalid;
    }
}
// the same from the survivor bitmap uni_finish_kernel left behind: the first 262 144 classes (> 90 % of a Zipf corpus' tokens) are
// answered from a 32 KB LDS copy, the rest by a gather into a bitmap 32x smaller than the count array
constexpr uint32_t kSurvLdsWords = 8192;
__global__ __launch_bounds__(kBlock) void uni_ids_bitmap_kernel(const uint32_t* __restrict__ cls, const uint32_t* __restrict__ surv, uint32_t nwords, uint32_t* __restrict__ ids,
                                                                 DevState* __restrict__ st, uint32_t npos) {
    if (st->done) return;
    __shared__ uint32_t survL[kSurvLdsWords];
    for (uint32_t w = threadIdx.x; w < kSurvLdsWords; w += kBlock) survL[w] = w < nwords ? surv[w] : 0u;
    __syncthreads();
    uint32_t      nvalid = 0;
    constexpr int kPer   = 8;
    for (uint32_t i0 = blockIdx.x * (kBlock * kPer); i0 < npos; i0 += gridDim.x * (kBlock * kPer)) {
        uint32_t c[kPer], w[kPer];
#pragma unroll
        for (int q = 0; q < kPer; ++q) {
            const uint32_t i = i0 + q * kBlock + threadIdx.x;
            c[q]             = (i < npos) ? cls[i] : 0u;
        }
#pragma unroll
        for (int q = 0; q < kPer; ++q) {
            const uint32_t word = c[q] >> 5;
            w[q]                = word < kSurvLdsWords ? survL[word] : surv[word];
        }
#pragma unroll
        for (int q = 0; q < kPer; ++q) {
            const uint32_t i = i0 + q * kBlock + threadIdx.x;
            if (i < npos) {
                const uint32_t id = (c[q] != 0 && ((w[q] >> (c[q] & 31u)) & 1u)) ? c[q] : kInvalid;
                nvalid += id != kInvalid;
                ids[i] = id;
            }
        }
    }
    __shared__ uint32_t redL[kBlock / kWave];
    for (int off = 32; off > 0; off >>= 1) nvalid += __shfl_down(nvalid, off, kWave);
    if ((threadIdx.x & (kWave - 1)) == 0) redL[threadIdx.x / kWave] = nvalid;
    __syncthreads();
    if (threadIdx.x == 0) {
        const uint32_t v = redL[0] + redL[1] + redL[2] + redL[3];
        if (v) atomicAdd(&st->valid, v);
    }
}

// =================================================================================================
// 3. prune + survivor compaction — replaces PatternModel::prune(threshold, n) (patternmodel.h:2107-2128)
//    One streaming pass over the table: survivors (count >= threshold) are appended to the result arrays
//    (wave-aggregated reservation) and their slot is tagged with the result index, which is the survivor id
//    the next order's keys are built from.
// =================================================================================================
__global__ __launch_bounds__(kBlock) void prune_kernel(Slot* __restrict__ table, DevState* __restrict__ st, uint32_t threshold, uint32_t* __restrict__ res_rep,
                                                        uint32_t* __restrict__ res_cnt, const uint32_t* __restrict__ nsrc, uint32_t minsrc, uint32_t res_cap) {
    if (st->done) return;
    __shared__ uint32_t baseL;
    const uint32_t      cap = st->cap, res_base = st->res_total;
    const uint32_t      ntiles = (cap + kPruneTile - 1) / kPruneTile;
    for (uint32_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const uint32_t t0 = tile * kPruneTile;
        uint32_t       cnt[kPrunePer], rep[kPrunePer];
        uint32_t       keepmask = 0, usedmask = 0;
#pragma unroll
        for (int k = 0; k < kPrunePer; ++k) {
            const uint32_t i = t0 + k * kBlock + threadIdx.x;
            cnt[k] = rep[k] = 0;
            if (i < cap) {
                const Slot s = table[i];
                if (s.key != kEmptyKey) {
                    usedmask |= 1u << k;
                    cnt[k] = s.count;
                    rep[k] = s.rep;
                    if (s.count >= threshold && (nsrc == nullptr || nsrc[i] >= minsrc)) keepmask |= 1u << k;
                }
            }
        }
        uint32_t       total;
        const uint32_t excl = block_exclusive_scan((uint32_t)__popc(keepmask), &total);
        if (threadIdx.x == 0) baseL = total ? atomicAdd(&st->kept, total) : 0;  // one reservation per 4096 slots
        __syncthreads();
        uint32_t r = res_base + baseL + excl;
#pragma unroll
        for (int k = 0; k < kPrunePer; ++k) {
            if (usedmask & (1u << k)) {
                const uint32_t i   = t0 + k * kBlock + threadIdx.x;
                uint32_t       tag = 0;
                if (keepmask & (1u << k)) {
                    if (r < res_cap) {
                        res_rep[r] = rep[k];
                        res_cnt[r] = cnt[k];
                        tag        = kKeptFlag | r;
                    } else {
                        st->overflow = 1;
                    }
                    ++r;
                }
                table[i].count = tag;
            }
        }
        __syncthreads();  // baseL is rewritten by the next tile
    }
}

// =================================================================================================
// 4. resolve: slot index per position -> survivor id per position (in place)
// =================================================================================================
__global__ __launch_bounds__(kBlock) void resolve_kernel(uint32_t* __restrict__ ids, const Slot* __restrict__ table, DevState* __restrict__ st, uint32_t npos) {
    if (st->done) return;
    uint32_t nvalid = 0;
    for (uint32_t i = blockIdx.x * kBlock + threadIdx.x; i < npos; i += gridDim.x * kBlock) {
        const uint32_t s  = ids[i];
        uint32_t       id = kInvalid;
        if (s != kInvalid) {
            const uint32_t tag = table[s].count;
            if (tag & kKeptFlag) {
                id = tag & ~kKeptFlag;
                ++nvalid;
            }
        }
        ids[i] = id;
    }
    __shared__ uint32_t redL[kBlock / kWave];
    for (int off = 32; off > 0; off >>= 1) nvalid += __shfl_down(nvalid, off, kWave);
    if ((threadIdx.x & (kWave - 1)) == 0) redL[threadIdx.x / kWave] = nvalid;
    __syncthreads();
    if (threadIdx.x == 0) {
        const uint32_t v = redL[0] + redL[1] + redL[2] + redL[3];
        if (v) atomicAdd(&st->valid, v);
    }
}

// skipgram passes (the host has synchronised and knows how many windows take part): set capacity, reset per-pass counters
__global__ void begin_pass_kernel(DevState* __restrict__ st, uint32_t cap) {
    st->cap  = cap;
    st->kept = 0;
}
// fold the survivors of a finished skipgram pass into the result total
__global__ void end_pass_kernel(DevState* __restrict__ st) {
    st->res_total += st->kept;
    st->kept = 0;
}

// single-lane bookkeeping between orders: statistics, result offsets, next capacity, termination
__global__ void advance_kernel(DevState* __restrict__ st, int n, uint32_t table_slots) {
    if (st->done) return;
    if (n < COLIBRI_MAX_ORDER) {
        st->s_found[n]    = st->found;
        st->s_kept[n]     = st->kept;
        st->s_admitted[n] = st->admitted;
        st->s_valid[n]    = st->valid;
    }
    if (st->found == 0) {  // reference: "None found" -> break (patternmodel.h:1189-1194)
        st->done = 1;
    } else {
        st->maxn = (uint32_t)n;
    }
    st->res_total += st->kept;
    if (n + 1 <= COLIBRI_MAX_ORDER) st->res_off[n + 1] = st->res_total;
    // next order: at most `valid` windows can be admitted, so 1.5x that many slots keeps the load factor <= 2/3
    uint64_t want = (uint64_t)st->valid + (st->valid >> 1) + 1024u;
    if (want > table_slots) want = table_slots;
    st->cap = (uint32_t)want;
    if (st->valid == 0) st->done = 1;  // nothing can be admitted at n+1: the next order would find nothing
    st->found = st->kept = st->admitted = st->valid = 0;
}

// The id-keeping modes (forward index, skipgram passes) with their order loop enqueued like the plain mode's: idm_ngram_end closes the n-gram pass of order n
// (figures, result range, "None found"), the skipgram passes of the order follow (skip_pass_begin / _end), idm_order_end decides whether order n + 1 can admit
// anything and clears the counters.
__global__ void idm_ngram_end_kernel(DevState* __restrict__ st, int n) {
    if (st->done) return;
    if (n < COLIBRI_MAX_ORDER) {
        st->s_found[n]    = st->found;
        st->s_kept[n]     = st->kept;
        st->s_admitted[n] = st->admitted;
        st->s_valid[n]    = st->valid;
        st->res_off[n]    = st->res_total;
    }
    if (st->found == 0) {  // reference: "None found" -> break (patternmodel.h:1189-1194)
        st->done = 1;
        return;
    }
    st->maxn = (uint32_t)n;
    st->res_total += st->kept;
    if (n + 1 <= COLIBRI_MAX_ORDER) st->res_off[n + 1] = st->res_total;
}
__global__ void idm_order_end_kernel(DevState* __restrict__ st, int n) {
    if (st->done) return;
    if (n < COLIBRI_MAX_ORDER && st->s_valid[n] == 0) st->done = 1;  // nothing can be admitted at n + 1
    st->found = st->kept = st->admitted = st->valid = 0;
}

// =================================================================================================
// 5. export: survivors (representative position, order, count) -> key bytes
//    replaces Pattern::write / BaseValueHandler::write over the map (pattern.cpp:268-277, datatypes.h:219-221)
// =================================================================================================
constexpr uint32_t kMaskFromClass = 0xFFFFFFFFu;  // segment of unigrams whose res_rep holds the CLASS: key bytes = its canonical varint (reference classencoder.cpp:22-42)
__device__ __forceinline__ uint32_t varint_len(uint32_t c) { return c < (1u << 7) ? 1u : c < (1u << 14) ? 2u : c < (1u << 21) ? 3u : c < (1u << 28) ? 4u : 5u; }
__global__ __launch_bounds__(kBlock) void export_len_kernel(const uint32_t* __restrict__ tokstart, const uint32_t* __restrict__ res_rep, uint32_t first, uint32_t count,
                                                             int n, uint32_t mask, uint32_t* __restrict__ keylen) {
    const uint32_t j = blockIdx.x * kBlock + threadIdx.x;
    if (j < count) {
        const uint32_t p = res_rep[first + j];
        uint32_t       len = 0;
        if (mask == kMaskFromClass) {
            len = varint_len(p);
        } else if (mask == 0) {
            len = tokstart[p + n] - tokstart[p];
        } else {  // a gapped token is the single byte 03 (reference src/pattern.cpp:886-908)
            for (int k = 0; k < n; ++k) len += ((mask >> k) & 1u) ? 1u : (tokstart[p + k + 1] - tokstart[p + k]);
        }
        keylen[first + j] = len;
    }
}

// two-level exclusive scan u32 -> u64 (block sums, then scan_sums_kernel, then apply)
__global__ __launch_bounds__(kBlock) void scan_reduce_kernel(const uint32_t* __restrict__ in, uint32_t n, unsigned long long* __restrict__ blocksum) {
    unsigned long long s = 0;
    const uint32_t     base = blockIdx.x * kBlock * 4;
    for (int k = 0; k < 4; ++k) {
        const uint32_t i = base + k * kBlock + threadIdx.x;
        if (i < n) s += in[i];
    }
    __shared__ unsigned long long ws[kBlock / kWave];
    for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off, kWave);
    if ((threadIdx.x & (kWave - 1)) == 0) ws[threadIdx.x / kWave] = s;
    __syncthreads();
    if (threadIdx.x == 0) blocksum[blockIdx.x] = ws[0] + ws[1] + ws[2] + ws[3];
}
__global__ __launch_bounds__(kBlock) void scan_sums_kernel(unsigned long long* __restrict__ blocksum, uint32_t nblocks, unsigned long long* __restrict__ total) {
    // ONE block scans the block sums (one per 1024 inputs) in chunks of 256 with a running carry
    __shared__ unsigned long long ws[kBlock / kWave];
    __shared__ unsigned long long carryL;
    if (threadIdx.x == 0) carryL = 0;
    __syncthreads();
    for (uint32_t base = 0; base < nblocks; base += kBlock) {
        const uint32_t     b = base + threadIdx.x;
        unsigned long long v = b < nblocks ? blocksum[b] : 0ull, incl = v;
        const uint32_t     lane = threadIdx.x & (kWave - 1), wave = threadIdx.x / kWave;
        for (int off = 1; off < kWave; off <<= 1) {
            const unsigned long long t = __shfl_up(incl, off, kWave);
            if ((int)lane >= off) incl += t;
        }
        if (lane == kWave - 1) ws[wave] = incl;
        __syncthreads();
        unsigned long long wbase = 0, tot = 0;
        for (int w = 0; w < kBlock / kWave; ++w) {
            if (w < (int)wave) wbase += ws[w];
            tot += ws[w];
        }
        const unsigned long long carry = carryL;
        if (b < nblocks) blocksum[b] = carry + wbase + incl - v;
        __syncthreads();
        if (threadIdx.x == 0) carryL = carry + tot;
        __syncthreads();
    }
    if (threadIdx.x == 0) *total = carryL;
}
__global__ __launch_bounds__(kBlock) void scan_apply_kernel(const uint32_t* __restrict__ in, uint32_t n, const unsigned long long* __restrict__ blocksum,
                                                             unsigned long long* __restrict__ out) {
    // each block re-scans its 1024 inputs: thread t owns 4 consecutive inputs
    const uint32_t base = blockIdx.x * kBlock * 4 + threadIdx.x * 4;
    uint32_t       v[4], s = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        v[k] = (base + k < n) ? in[base + k] : 0;
        s += v[k];
    }
    uint32_t           tot;
    unsigned long long ex = blocksum[blockIdx.x] + block_exclusive_scan(s, &tot);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        if (base + k < n) out[base + k] = ex;
        ex += v[k];
    }
}

__global__ __launch_bounds__(kBlock) void export_bytes_kernel(const uint8_t* __restrict__ bytes, const uint32_t* __restrict__ tokstart, const uint32_t* __restrict__ res_rep,
                                                               const uint32_t* __restrict__ keylen, const unsigned long long* __restrict__ keyoff, uint32_t first,
                                                               uint32_t count, int n, uint32_t mask, uint8_t* __restrict__ out) {
    const uint32_t j = blockIdx.x * kBlock + threadIdx.x;
    if (j < count) {
        const uint32_t p = res_rep[first + j];
        uint8_t*       dst = out + keyoff[first + j];
        if (mask == kMaskFromClass) {  // little-endian base 128, bit 7 set on every byte but the last
            uint32_t c = p, o = 0;
            while (c >= 128u) {
                dst[o++] = (uint8_t)((c & 127u) | 128u);
                c >>= 7;
            }
            dst[o] = (uint8_t)c;
        } else if (mask == 0) {
            const uint8_t* src = bytes + tokstart[p];
            const uint32_t len = keylen[first + j];
            for (uint32_t b = 0; b < len; ++b) dst[b] = src[b];
        } else {
            uint32_t o = 0;
            for (int k = 0; k < n; ++k) {
                if ((mask >> k) & 1u) {
                    dst[o++] = 3;
                } else {
                    for (uint32_t b = tokstart[p + k]; b < tokstart[p + k + 1]; ++b) dst[o++] = bytes[b];
                }
            }
        }
    }
}

// =================================================================================================
// 6. forward index (IndexedPatternModel): pattern -> sorted [(sentence, token)]
//    replaces IndexedDataHandler::add / push_back per occurrence and the final per-pattern sorts
//    (reference include/datatypes.h:283-289, include/patternmodel.h:2699-2705, :2789-2800).
//    Every pass emits (result id, position) pairs in position order (ordered compaction); one stable LSD radix sort by
//    result id then groups them, positions staying ascending inside a group; a last kernel turns positions into
//    (sentence, token) with a binary search in the delimiter table.
// =================================================================================================
// result id per position of a finished skipgram pass (its slot array + the tags the prune kernel left in the table)
__global__ __launch_bounds__(kBlock) void skip_result_ids_kernel(const uint32_t* __restrict__ slot_of, const Slot* __restrict__ table, uint32_t* __restrict__ out, uint32_t npos) {
    for (uint32_t i = blockIdx.x * kBlock + threadIdx.x; i < npos; i += gridDim.x * kBlock) {
        const uint32_t s  = slot_of[i];
        uint32_t       id = kInvalid;
        if (s != kInvalid) {
            const uint32_t tag = table[s].count;
            if (tag & kKeptFlag) id = tag & ~kKeptFlag;
        }
        out[i] = id;
    }
}
constexpr int kEmitPer  = 4;
constexpr int kEmitTile = kBlock * kEmitPer;
constexpr int kPairThreads = 1024, kPairPer = 8, kPairTile = kPairThreads * kPairPer;  // 8192 positions per block: 12.8 K block counts for 10^8 positions (one short scan)
__device__ __forceinline__ void pair_load(const uint32_t* __restrict__ ids, uint32_t npos, uint32_t base, uint32_t (&v)[kPairPer]) {
    if (base + kPairPer <= npos) {
        const uint4 a = *reinterpret_cast<const uint4*>(ids + base), b = *reinterpret_cast<const uint4*>(ids + base + 4);
        v[0] = a.x, v[1] = a.y, v[2] = a.z, v[3] = a.w, v[4] = b.x, v[5] = b.y, v[6] = b.z, v[7] = b.w;
    } else {
#pragma unroll
        for (int k = 0; k < kPairPer; ++k) v[k] = (base + k < npos) ? ids[base + k] : kInvalid;
    }
}
__device__ __forceinline__ uint32_t pair_block_scan(uint32_t c, uint32_t* total) {  // exclusive scan over the block's 1024 threads
    __shared__ uint32_t wsum[kPairThreads / kWave];
    const uint32_t      lane = threadIdx.x & (kWave - 1), w = threadIdx.x / kWave;
    uint32_t            inc = c;
#pragma unroll
    for (int d = 1; d < kWave; d <<= 1) {
        const uint32_t t = __shfl_up(inc, d);
        if ((int)lane >= d) inc += t;
    }
    if (lane == kWave - 1) wsum[w] = inc;
    __syncthreads();
    uint32_t before = 0, all = 0;
#pragma unroll
    for (int q = 0; q < kPairThreads / kWave; ++q) {
        const uint32_t t = wsum[q];
        before += (uint32_t)q < w ? t : 0u;
        all += t;
    }
    *total = all;
    return before + inc - c;
}
// in-place exclusive scan of the tiles' counts by one block of 1024 threads, 16 values a thread: the 12.2 K tiles of 10^8 positions in one round
// (scan_small_kernel's 256 threads x 4 values went through twelve dependent rounds for them: 90 us between the two sweeps)
__global__ __launch_bounds__(kPairThreads) void scan_tiles_kernel(uint32_t* data, uint32_t n, uint32_t* total_out) {
    constexpr int kPer = 16;
    uint32_t      carry = 0;
    for (uint32_t base = 0; base < n; base += kPairThreads * kPer) {
        const uint32_t i0 = base + threadIdx.x * kPer;
        uint32_t       v[kPer], s = 0;
#pragma unroll
        for (int k = 0; k < kPer; ++k) {
            v[k] = i0 + k < n ? data[i0 + k] : 0u;
            s += v[k];
        }
        uint32_t tot;
        uint32_t ex = carry + pair_block_scan(s, &tot);
#pragma unroll
        for (int k = 0; k < kPer; ++k) {
            if (i0 + k < n) data[i0 + k] = ex;
            ex += v[k];
        }
        carry += tot;
        __syncthreads();  // (pair_block_scan's wave sums are read until here)
    }
    if (threadIdx.x == 0 && total_out != nullptr) *total_out = carry;
}
// Two sweeps over the ids (count per tile, short scan, write); the number of pairs lives on the device. chain[0], chain[1]: pairs so far, before / after
// this pass (which = the one to read); chain[2]: set when the pairs outgrew `cap` (the count goes on, the writes stop: the host then knows how much room
// the model needs). (A one-sweep version with a decoupled look-back over the 12.8 K tiles was 2 x slower: too few tiles in flight to hide the chain.)
constexpr int kChainHead = 3;
__global__ __launch_bounds__(kPairThreads) void emit_count_kernel(const uint32_t* __restrict__ ids, uint32_t npos, uint32_t* __restrict__ blockcnt,
                                                                   const DevState* __restrict__ st = nullptr /* optional: an enqueued run that is over (st->done) emits nothing */,
                                                                   const uint32_t* __restrict__ surv = nullptr /* order 1 of an indexed model without skipgram passes: `ids` is the
                                                                   CLASS per position, a position counts when its class survived (one bit per class); no ids[1] is built ... */,
                                                                   uint32_t* __restrict__ valid_out = nullptr /* ... and the positions with a surviving unigram are counted here */) {
    if (st != nullptr && st->done) {
        if (threadIdx.x == 0) blockcnt[blockIdx.x] = 0;
        return;
    }
    uint32_t v[kPairPer], c = 0;
    pair_load(ids, npos, blockIdx.x * kPairTile + threadIdx.x * kPairPer, v);
    if (surv != nullptr) {
#pragma unroll
        for (int k = 0; k < kPairPer; ++k) v[k] = (v[k] != kInvalid && v[k] != 0u && ((surv[v[k] >> 5] >> (v[k] & 31u)) & 1u)) ? 0u : kInvalid;
    }
#pragma unroll
    for (int k = 0; k < kPairPer; ++k) c += v[k] != kInvalid;
    uint32_t total;
    pair_block_scan(c, &total);
    if (threadIdx.x == 0) {
        blockcnt[blockIdx.x] = total;
        if (valid_out != nullptr && total) atomicAdd(valid_out, total);
    }
}
__global__ void pairs_advance_kernel(unsigned long long* __restrict__ chain, int which, const uint32_t* __restrict__ total, uint64_t cap) {
    const unsigned long long after = chain[which] + *total;
    chain[which ^ 1]               = after;
    if (after > cap) chain[2] = 1ull;
}
// position -> (sentence, token) through a table 30 x smaller than one entry per position (16 bytes per 64 positions instead of 8 bytes per position): { sentences that end before the block, position where
// the sentence running at the block's first position starts, one bit per position of the block that is a delimiter }. A model's 1.6 x 10^8 references
// are sorted by pattern, i.e. their positions are random: gathers from the 0.8 GB per-position table went to HBM (3.5 ms), the 26 MB one stays in cache.
struct __attribute__((aligned(16))) PosBlock {
    uint32_t           sent_before, sent_start;
    unsigned long long delim;
};
// `blocks` (the usual case: sentence count and longest sentence leave room beside a 31-bit id): the pair carries the reference itself instead of the position —
// id << (sb + tb) | (sentences before the position) << tb | token — looked up HERE, where the positions of a tile are consecutive (one 16-byte table entry per lane,
// coalesced), not after the sort, where they are random (161 M gathers: 1.7 ms of the last sort pass). Without: id << 32 | position.
__global__ __launch_bounds__(kPairThreads) void emit_write_kernel(const uint32_t* __restrict__ ids, uint32_t npos, const uint32_t* __restrict__ blockoff,
                                                                   const unsigned long long* __restrict__ chain, int which, uint64_t cap,
                                                                   unsigned long long* __restrict__ pairs, const PosBlock* __restrict__ blocks = nullptr, uint32_t sb = 0,
                                                                   uint32_t tb = 0, uint32_t* __restrict__ pay = nullptr /* split pairs (isort_*): the id goes to pairs as u32[],
                                                                   sentence << tb | token here */,
                                                                   const uint32_t* __restrict__ resid = nullptr /* `ids` is the class per position: the id is resid[class] */) {
    if (blockoff[blockIdx.x + 1] == blockoff[blockIdx.x]) return;  // nothing in this tile (the passes of the high orders are sparse); [ntiles] holds the total
    // Row-wise (round 5): the tile is eight rows of 1024 consecutive positions, a lane has one position of each row. A row's valid entries are ranked with the
    // waves' ballots, so neighbouring lanes write neighbouring pairs (whole lines; the lane-owns-eight-positions form wrote eight words per lane, 32 bytes apart
    // across the wave: 0.72 ms for 10^8 pairs), and a wave's 64 positions of a row are one PosBlock, read once
    constexpr int       kWaves = kPairThreads / kWave;
    __shared__ uint32_t rowcnt[kPairPer * kWaves];  // valid entries per (row, wave), then their exclusive prefix in tile order
    const uint32_t      lane = threadIdx.x & (kWave - 1), w = threadIdx.x / kWave, tile = blockIdx.x * kPairTile;
    uint32_t            v[kPairPer];
#pragma unroll
    for (int k = 0; k < kPairPer; ++k) {
        const uint32_t p = tile + (uint32_t)k * kPairThreads + threadIdx.x;
        v[k]             = p < npos ? ids[p] : kInvalid;
    }
    if (resid != nullptr) {
#pragma unroll
        for (int k = 0; k < kPairPer; ++k) v[k] = (v[k] != kInvalid && v[k] != 0u) ? resid[v[k]] : kInvalid;
    }
    unsigned long long bal[kPairPer];
#pragma unroll
    for (int k = 0; k < kPairPer; ++k) {
        bal[k] = __ballot(v[k] != kInvalid);
        if (lane == 0) rowcnt[k * kWaves + (int)w] = (uint32_t)__popcll(bal[k]);
    }
    __syncthreads();
    static_assert(kPairPer * kWaves == 2 * kWave, "one wave scans the tile's (row, wave) counts, two per lane");
    if (w == 0) {
        const uint32_t a = rowcnt[2 * lane], b2 = rowcnt[2 * lane + 1];
        uint32_t       inc = a + b2;
#pragma unroll
        for (int d = 1; d < kWave; d <<= 1) {
            const uint32_t t = __shfl_up(inc, d);
            if ((int)lane >= d) inc += t;
        }
        rowcnt[2 * lane]     = inc - a - b2;
        rowcnt[2 * lane + 1] = inc - b2;
    }
    __syncthreads();
    const uint64_t first = chain[which] + blockoff[blockIdx.x];
    const uint32_t tmask = tb >= 32 ? 0xFFFFFFFFu : (1u << tb) - 1u;
    const uint64_t lower = (1ull << lane) - 1ull;
#pragma unroll
    for (int k = 0; k < kPairPer; ++k) {
        if (bal[k] == 0ull) continue;  // (wave-uniform)
        const uint32_t p = tile + (uint32_t)k * kPairThreads + threadIdx.x;  // p & 63 == lane
        const uint64_t o = first + rowcnt[k * kWaves + (int)w] + (uint32_t)__popcll(bal[k] & lower);
        if (v[k] == kInvalid || o >= cap) continue;
        if (blocks != nullptr) {
            const uint4    r     = *reinterpret_cast<const uint4*>(blocks + (p >> 6));  // (one entry per wave and row)
            const uint64_t delim = ((uint64_t)r.w << 32) | r.z, below = delim & lower;
            const uint32_t sent = r.x + (uint32_t)__popcll(below), tok = below ? lane - (64u - (uint32_t)__clzll(below)) : p - r.y;
            if (pay != nullptr) {
                reinterpret_cast<uint32_t*>(pairs)[o] = v[k];
                pay[o]                                = (sent << tb) | (tok & tmask);
            } else
                pairs[o] = ((unsigned long long)v[k] << (sb + tb)) | ((unsigned long long)sent << tb) | (tok & tmask);
        } else {
            pairs[o] = ((unsigned long long)v[k] << 32) | p;
        }
    }
}

// ---- order 1 of an indexed model: the references of the HOT unigrams never enter the sort (round 6) ----------------------------------------------------------------------
// The stable sort by result id exists because a pattern's occurrences are spread over the corpus. For the most frequent words they are not spread thinly: a tile of
// 8192 positions holds hundreds of "the" and still a handful of the 256th word, and these 256 words own ~40 % of the unigram references of a Zipf corpus (a quarter of ALL
// references of a model with n <= 5) — each of them sorted three times. Here a tile counts its occurrences of every hot id (LDS histogram), one scan per hot id over
// the tiles gives every tile its place in the id's list, and the write sweep puts the (sentence, token) of a hot occurrence straight at its final place: base of the id's
// list + occurrences in the tiles before + rank inside the tile (waves own 512 consecutive positions; rank = the waves before + the wave's rows before + the lanes before,
// as in isort_scatter_kernel). Stable by construction, nothing to sort.
// Which ids: uni_finish_kernel numbers the surviving classes of a tile of kPruneTile classes consecutively in class order (the tiles among each other in the order their
// blocks arrived), so the first kHotIds survivors of class tile 0 — classes are frequency-ranked, reference src/classencoder.cpp:220-224 — have the ids base0 .. base0 + hn - 1.
// In the sorted order of the OTHER references, those of ids below base0 come first (`below` of them: the sum of their counts), so the hot block lies at [below, below + nhot)
// and the last sort pass moves everything behind `below` up by nhot. Correct for any class numbering; it pays when the low classes are the frequent ones.
constexpr int kHotIds = 256;
struct HotInfo {
    uint32_t           base0, hn;      // hot ids: [base0, base0 + hn)
    unsigned long long below, nhot;    // references that sort before the hot block; references of the hot ids
    uint32_t           total[kHotIds], base[kHotIds];  // occurrences per hot id; their exclusive scan
    uint8_t            hotmap[kPruneTile];             // class c < kPruneTile -> its hot id - base0, 0xFF: not hot (the sweeps over CLASSES look here, not in the 4 MB of resid)
};
// one block: the hot ids of this run and the references sorting before them
__global__ __launch_bounds__(kPairThreads) void hot_setup_kernel(const uint32_t* __restrict__ resid, uint32_t nclasses, const uint32_t* __restrict__ res_cnt, const DevState* __restrict__ st,
                                                                  HotInfo* __restrict__ hi, uint32_t closed /* idm_ngram_end_kernel has closed order 1: its ids start at res_off[1] */) {
    __shared__ uint32_t           redA[kPairThreads / kWave], redB[kPairThreads / kWave];
    __shared__ unsigned long long redC[kPairThreads / kWave];
    __shared__ uint32_t           base0L;
    const uint32_t lane = threadIdx.x & (kWave - 1), w = threadIdx.x / kWave;
    uint32_t       lo = kInvalid, nv = 0;
    if (!st->done)
        for (uint32_t cidx = threadIdx.x; cidx < min(nclasses, (uint32_t)kPruneTile); cidx += kPairThreads) {
            const uint32_t r = resid[cidx];
            if (r != kInvalid) {
                lo = min(lo, r);
                ++nv;
            }
        }
    for (int off = 32; off > 0; off >>= 1) {
        lo = min(lo, (uint32_t)__shfl_down((int)lo, off, kWave));
        nv += __shfl_down(nv, off, kWave);
    }
    if (lane == 0) redA[w] = lo, redB[w] = nv;
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t a = kInvalid, b = 0;
        for (int q = 0; q < kPairThreads / kWave; ++q) a = min(a, redA[q]), b += redB[q];
        base0L    = a;
        hi->base0 = a == kInvalid ? 0u : a;
        hi->hn    = a == kInvalid ? 0u : min(b, (uint32_t)kHotIds - 1u);  // (0xFF is the map's "not hot")
    }
    __syncthreads();
    const uint32_t     base0 = base0L == kInvalid ? 0u : base0L;
    {
        const uint32_t hn = hi->hn;  // (thread 0 wrote it before the barrier)
        for (uint32_t cidx = threadIdx.x; cidx < (uint32_t)kPruneTile; cidx += kPairThreads) {
            const uint32_t r = (cidx < nclasses && !st->done) ? resid[cidx] : kInvalid;
            hi->hotmap[cidx] = (r != kInvalid && r - base0 < hn) ? (uint8_t)(r - base0) : (uint8_t)0xFF;
        }
    }
    unsigned long long sum = 0;
    for (uint32_t r0 = (closed ? st->res_off[1] : st->res_total) + threadIdx.x; r0 < base0; r0 += 8 * kPairThreads) {  // (from order 1's first id; eight loads in flight)
        uint32_t x[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) x[q] = r0 + q * kPairThreads < base0 ? res_cnt[r0 + q * kPairThreads] : 0u;
#pragma unroll
        for (int q = 0; q < 8; ++q) sum += x[q];
    }
    for (int off = 32; off > 0; off >>= 1) sum += __shfl_down(sum, off, kWave);
    if (lane == 0) redC[w] = sum;
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned long long t = 0;
        for (int q = 0; q < kPairThreads / kWave; ++q) t += redC[q];
        hi->below = t;
        hi->nhot  = 0;
    }
}
// count sweep: per tile the references that will travel as pairs (blockcnt) and the occurrences of every hot id (hotcnt[h * ntiles + tile])
__global__ __launch_bounds__(kPairThreads) void emit_hot_count_kernel(const uint32_t* __restrict__ ids, uint32_t npos, uint32_t* __restrict__ blockcnt, const DevState* __restrict__ st,
                                                                       const uint32_t* __restrict__ surv, const uint32_t* __restrict__ resid, uint32_t* __restrict__ valid_out,
                                                                       const HotInfo* __restrict__ hi, uint32_t* __restrict__ hotcnt, uint32_t ntiles) {
    __shared__ uint32_t histL[kHotIds], nhL;
    __shared__ uint32_t mapL[kPruneTile / 4];
    if (threadIdx.x < kHotIds) histL[threadIdx.x] = 0;
    if (threadIdx.x == 0) nhL = 0;
    if (resid != nullptr) mapL[threadIdx.x] = reinterpret_cast<const uint32_t*>(hi->hotmap)[threadIdx.x];
    static_assert(kPruneTile / 4 == kPairThreads, "one word of the map per thread");
    __syncthreads();
    uint32_t c = 0, nh = 0;
    if (!st->done) {
        const uint32_t base0 = hi->base0, hn = hi->hn;
        uint32_t       v[kPairPer];
        pair_load(ids, npos, blockIdx.x * kPairTile + threadIdx.x * kPairPer, v);
#pragma unroll
        for (int k = 0; k < kPairPer; ++k) {
            uint32_t h = 0xFFu;
            if (resid != nullptr) {  // classes: a survivor bit says whether the position counts, the map whether its id is hot
                if (v[k] == kInvalid || v[k] == 0u || !((surv[v[k] >> 5] >> (v[k] & 31u)) & 1u)) continue;
                if (v[k] < (uint32_t)kPruneTile) h = reinterpret_cast<const uint8_t*>(mapL)[v[k]];
            } else {
                if (v[k] == kInvalid) continue;
                if (v[k] - base0 < hn) h = v[k] - base0;
            }
            if (h != 0xFFu) {
                atomicAdd(&histL[h], 1u);
                ++nh;
            } else {
                ++c;
            }
        }
    }
    for (int off = 32; off > 0; off >>= 1) nh += __shfl_down(nh, off, kWave);
    if ((threadIdx.x & (kWave - 1)) == 0 && nh) atomicAdd(&nhL, nh);
    uint32_t total;
    pair_block_scan(c, &total);
    if (threadIdx.x < kHotIds) hotcnt[(size_t)threadIdx.x * ntiles + blockIdx.x] = histL[threadIdx.x];  // (pair_block_scan's barrier is behind every atomic)
    if (threadIdx.x == 0) {
        blockcnt[blockIdx.x] = total;
        if (valid_out != nullptr && total + nhL) atomicAdd(valid_out, total + nhL);  // (ONE atomic per block on this word: a single address takes ~12 ns each)
    }
}
// block h: the tiles' counts of hot id h -> their exclusive scan in place, the id's total to hi->total[h]
__global__ __launch_bounds__(kPairThreads) void hot_scan_kernel(uint32_t* __restrict__ hotcnt, uint32_t ntiles, HotInfo* __restrict__ hi) {
    constexpr int   kPer = 16;
    uint32_t* const data = hotcnt + (size_t)blockIdx.x * ntiles;
    uint32_t        carry = 0;
    for (uint32_t base = 0; base < ntiles; base += kPairThreads * kPer) {
        const uint32_t i0 = base + threadIdx.x * kPer;
        uint32_t       v[kPer], s = 0;
#pragma unroll
        for (int k = 0; k < kPer; ++k) {
            v[k] = i0 + k < ntiles ? data[i0 + k] : 0u;
            s += v[k];
        }
        uint32_t tot;
        uint32_t ex = carry + pair_block_scan(s, &tot);
#pragma unroll
        for (int k = 0; k < kPer; ++k) {
            if (i0 + k < ntiles) data[i0 + k] = ex;
            ex += v[k];
        }
        carry += tot;
        __syncthreads();
    }
    if (threadIdx.x == 0) hi->total[blockIdx.x] = carry;
}
// one block of kHotIds threads: the hot ids' lists one after the other; chain[kChainHot] / [kChainBelow]: what finalize_index reads with the pair count
constexpr int kChainHot = 3, kChainBelow = 4, kChainDisorder = 5, kChainWords = 6;
__global__ __launch_bounds__(kHotIds) void hot_base_kernel(HotInfo* __restrict__ hi, unsigned long long* __restrict__ chain) {
    __shared__ uint32_t wsum[kHotIds / kWave];
    const uint32_t      lane = threadIdx.x & (kWave - 1), w = threadIdx.x / kWave, v = threadIdx.x < hi->hn ? hi->total[threadIdx.x] : 0u;
    uint32_t            inc = v;
    for (int d = 1; d < kWave; d <<= 1) {
        const uint32_t t = __shfl_up(inc, d);
        if ((int)lane >= d) inc += t;
    }
    if (lane == kWave - 1) wsum[w] = inc;
    __syncthreads();
    uint32_t before = 0, all = 0;
    for (int q = 0; q < kHotIds / kWave; ++q) {
        before += (uint32_t)q < w ? wsum[q] : 0u;
        all += wsum[q];
    }
    hi->base[threadIdx.x] = before + inc - v;
    if (threadIdx.x == 0) {
        hi->nhot           = all;
        chain[kChainHot]   = all;
        chain[kChainBelow] = hi->below;
    }
}
// write sweep (split, packed pairs only): the other references leave as pairs in position order, exactly as emit_write_kernel's; a hot occurrence goes to its final place.
// Tiles are dealt to the XCDs in contiguous ranges (block b works on tile (b % 8) * per + b / 8): consecutive tiles append to the same lines of a hot list, and
// should meet in one L2.
__global__ __launch_bounds__(kPairThreads) void emit_hot_write_kernel(const uint32_t* __restrict__ ids, uint32_t npos, const uint32_t* __restrict__ blockoff,
                                                                       unsigned long long* chain /* read: [which]; written: [kChainDisorder] */, int which, uint64_t cap,
                                                                       uint32_t* __restrict__ pair_id, uint32_t* __restrict__ pay, const PosBlock* __restrict__ blocks, uint32_t tb,
                                                                       const uint32_t* __restrict__ surv,
                                                                       const uint32_t* __restrict__ resid, const HotInfo* __restrict__ hi, const uint32_t* __restrict__ hotoff,
                                                                       uint32_t ntiles, uint32_t first_sentence, uint32_t* __restrict__ ref_sentence, uint16_t* __restrict__ ref_token,
                                                                       const DevState* __restrict__ st) {
    if (st->done) return;
    constexpr int  kWaves = kPairThreads / kWave, kRows = kPairPer;
    const uint32_t per = (ntiles + 7u) / 8u, t = (blockIdx.x % 8u) * per + blockIdx.x / 8u;
    if (blockIdx.x / 8u >= per || t >= ntiles) return;
    __shared__ uint32_t rowcnt[kRows * kWaves];
    __shared__ uint32_t wcntL[kWaves][kHotIds];  // occurrences of hot id h wave w has seen; then: those of the waves before it
    __shared__ uint32_t toffL[kHotIds], runL[kHotIds], thL[kHotIds], hsumL[kHotIds / kWave];
    // the tile's hot references, staged in (hot id, position) order: a run per id leaves as whole lines (straight from the lanes, a row of 64 positions wrote to ~15 lists)
    // (sentence << tb | token in one word, as a pair's payload: 57 KB of LDS in all, two blocks per CU — with a 16-bit token array beside it, 73 KB, only one was resident)
    __shared__ uint32_t stgS[kPairTile];
    __shared__ uint8_t  stgH[kPairTile];
    const uint32_t      lane = threadIdx.x & (kWave - 1), w = threadIdx.x / kWave, tile = t * kPairTile;
    const uint32_t      base0 = hi->base0, hn = hi->hn;
    for (uint32_t k = threadIdx.x; k < (uint32_t)(kWaves * kHotIds); k += kPairThreads) (&wcntL[0][0])[k] = 0;
    if (threadIdx.x < kHotIds) toffL[threadIdx.x] = threadIdx.x < hn ? hi->base[threadIdx.x] + hotoff[(size_t)threadIdx.x * ntiles + t] : 0u;
    __shared__ uint32_t mapL[kPruneTile / 4];
    if (resid != nullptr) mapL[threadIdx.x] = reinterpret_cast<const uint32_t*>(hi->hotmap)[threadIdx.x];
    uint32_t v[kRows], rank[kRows];
#pragma unroll
    for (int r = 0; r < kRows; ++r) {  // the wave's 512 consecutive positions, row by row
        const uint32_t p = tile + w * (kWave * kRows) + (uint32_t)r * kWave + lane;
        v[r]             = p < npos ? ids[p] : kInvalid;
    }
    KP_INIT(5);
    __syncthreads();
    KP(0);  // (ids / classes, the tile's offsets and the map have arrived)
    // what a position is: kHqCold (a reference that travels as a pair), kHqNone, or its hot id - base0. Classes: the map says which; the cold classes' result indices are a
    // gather each (as emit_write_kernel's) whose answer only the last loop needs — the ranking must not wait for it
    constexpr uint32_t kHqCold = 0x100u, kHqNone = 0x1FFu;
    uint32_t           hq[kRows], sw[kRows];
#pragma unroll
    for (int r = 0; r < kRows; ++r)  // (all rows' survivor words first: a look-up between two rows' gathers would wait for the gather before it — loads return in order)
        sw[r] = (resid != nullptr && v[r] != kInvalid && v[r] != 0u) ? surv[v[r] >> 5] : 0u;
#pragma unroll
    for (int r = 0; r < kRows; ++r) {
        const uint32_t cc = v[r];
        hq[r]             = kHqNone;
        if (resid != nullptr) {
            if ((sw[r] >> (cc & 31u)) & 1u) {
                const uint32_t h = cc < (uint32_t)kPruneTile ? (uint32_t)reinterpret_cast<const uint8_t*>(mapL)[cc] : 0xFFu;
                hq[r]            = h != 0xFFu ? h : kHqCold;
                v[r]             = h != 0xFFu ? 0u : resid[cc];
            }
        } else if (cc != kInvalid) {
            hq[r] = cc - base0 < hn ? cc - base0 : kHqCold;
        }
    }
    KP(5);  // (this wave's classes / ids have arrived, survivor bits looked up)
    unsigned long long bal[kRows];
#pragma unroll
    for (int r = 0; r < kRows; ++r) {
        const bool hot = hq[r] < kHqCold;
        bal[r]         = __ballot(hq[r] == kHqCold);
        if (lane == 0) rowcnt[(int)w * kRows + r] = (uint32_t)__popcll(bal[r]);
        // rank among the wave's earlier occurrences of the id: one returning LDS add per hot lane. A wave's DS instructions execute in order, so the rows are; lanes of ONE
        // instruction that hit the same counter are served in lane order on this hardware, which no manual promises — the copy-out checks every run for ascending
        // references and raises chain[kChainDisorder] otherwise (the run is then repeated with every reference through the sort). Matching the lanes by id with ballots,
        // as isort_scatter_kernel does, is exact by construction and cost 0.48 of this kernel's 1.02 ms (eight ballots and 64-bit selects per row of 64 positions).
        rank[r] = hot ? atomicAdd(&wcntL[w][hq[r]], 1u) : 0u;
    }
    __syncthreads();
    KP(1);  // (ranks)
    static_assert(kRows * kWaves == 2 * kWave, "one wave scans the tile's (wave, row) counts, two per lane");
    if (w == 0) {
        const uint32_t a = rowcnt[2 * lane], b2 = rowcnt[2 * lane + 1];
        uint32_t       inc = a + b2;
#pragma unroll
        for (int d = 1; d < kWave; d <<= 1) {
            const uint32_t x = __shfl_up(inc, d);
            if ((int)lane >= d) inc += x;
        }
        rowcnt[2 * lane]     = inc - a - b2;
        rowcnt[2 * lane + 1] = inc - b2;
    } else if (threadIdx.x - kWave < (uint32_t)kHotIds) {  // hot id q: the waves' counts -> those of the waves before (waves 1 .. 4)
        const uint32_t q = threadIdx.x - kWave;
        uint32_t       run = 0;
#pragma unroll
        for (int x = 0; x < kWaves; ++x) {
            const uint32_t cc = wcntL[x][q];
            wcntL[x][q]       = run;
            run += cc;
        }
        runL[q] = run;
    }
    __syncthreads();
    {  // exclusive scan of the hot ids' tile totals: where an id's run starts in the staging area
        uint32_t x = 0, inc = 0;
        if (threadIdx.x < (uint32_t)kHotIds) {
            x   = runL[threadIdx.x];
            inc = x;
#pragma unroll
            for (int d = 1; d < kWave; d <<= 1) {
                const uint32_t y = __shfl_up(inc, d);
                if ((int)lane >= d) inc += y;
            }
            if (lane == kWave - 1) hsumL[w] = inc;
        }
        __syncthreads();
        if (threadIdx.x < (uint32_t)kHotIds) {
            uint32_t before = 0;
            for (uint32_t y = 0; y < w; ++y) before += hsumL[y];
            thL[threadIdx.x] = before + inc - x;
        }
        __syncthreads();
    }
    KP(2);  // (scans)
    unsigned long long* const flags = chain;
    const uint64_t first = chain[which] + blockoff[t], hot0 = hi->below;
    const uint32_t tmask = tb >= 32 ? 0xFFFFFFFFu : (1u << tb) - 1u;
    const uint64_t lower = (1ull << lane) - 1ull;
#pragma unroll
    for (int r = 0; r < kRows; ++r) {
        if (hq[r] == kHqNone) continue;
        const uint32_t p     = tile + w * (kWave * kRows) + (uint32_t)r * kWave + lane;  // p & 63 == lane
        const uint4    pb    = *reinterpret_cast<const uint4*>(blocks + (p >> 6));        // (one entry per wave and row)
        const uint64_t delim = ((uint64_t)pb.w << 32) | pb.z, under = delim & lower;
        const uint32_t sent = pb.x + (uint32_t)__popcll(under), tok = under ? lane - (64u - (uint32_t)__clzll(under)) : p - pb.y;
        if (hq[r] < kHqCold) {
            const uint32_t h = hq[r], at = thL[h] + wcntL[w][h] + rank[r];
            stgS[at]         = (sent << tb) | (tok & tmask);
            stgH[at]         = (uint8_t)h;
        } else {
            const uint64_t o = first + rowcnt[(int)w * kRows + r] + (uint32_t)__popcll(bal[r] & lower);
            if (o < cap) {
                pair_id[o] = v[r];
                pay[o]     = (sent << tb) | (tok & tmask);
            }
        }
    }
    __syncthreads();
    KP(3);  // (position blocks read, cold pairs written, hot references staged)
    const uint32_t nstaged = thL[kHotIds - 1] + runL[kHotIds - 1];
    bool disorder = false;
    for (uint32_t j = threadIdx.x; j < nstaged; j += kPairThreads) {
        const uint32_t h   = stgH[j];
        const uint64_t dst = hot0 + toffL[h] + (j - thL[h]);
        const uint32_t yy  = stgS[j];
        disorder           = disorder || (j > thL[h] && stgS[j - 1] >= yy);  // (inside a run (sentence, token) ascends with the position)
        ref_sentence[dst]  = first_sentence + (tb >= 32 ? 0u : yy >> tb);
        ref_token[dst]     = (uint16_t)(yy & tmask);
    }
    if (disorder) flags[kChainDisorder] = 1ull;
    KP(4);
    KP_DONE();
}

// The positions whose entry of `ids` is valid, in ascending order (emit_count_kernel's tile counts, scanned: blockoff[ntiles] = their number): the list the
// skipgram passes of an order walk. In position order, so that references emitted list entry by list entry (below) arrive in corpus order, as the stable sort of
// the forward index needs them.
__global__ __launch_bounds__(kPairThreads) void list_write_kernel(const uint32_t* __restrict__ ids, uint32_t npos, const uint32_t* __restrict__ blockoff, uint32_t* __restrict__ list,
                                                                   uint32_t* __restrict__ nlist) {
    if (blockIdx.x == 0 && threadIdx.x == 0) *nlist = blockoff[gridDim.x];
    if (blockoff[blockIdx.x + 1] == blockoff[blockIdx.x]) return;
    const uint32_t base = blockIdx.x * kPairTile + threadIdx.x * kPairPer;
    uint32_t       v[kPairPer], c = 0;
    pair_load(ids, npos, base, v);
#pragma unroll
    for (int k = 0; k < kPairPer; ++k) c += v[k] != kInvalid;
    uint32_t total;
    uint32_t o = blockoff[blockIdx.x] + pair_block_scan(c, &total);
#pragma unroll
    for (int k = 0; k < kPairPer; ++k)
        if (v[k] != kInvalid) list[o++] = base + k;
}
// emit_count / emit_write over the ENTRIES of such a list instead of over every position: the skipgram passes of an indexed model leave ids at the few
// positions of the order's list (everything else in `ids` is stale), and a pass over the 10^8 positions of the corpus per gap mask (two sweeps, 0.24 ms) cost more
// than the pass's counting. `bound`: what the host knows the list cannot exceed (the grid); entries beyond *nlist count as invalid.
__global__ __launch_bounds__(kPairThreads) void emit_count_list_kernel(const uint32_t* __restrict__ ids, const uint32_t* __restrict__ list, const uint32_t* __restrict__ nlist,
                                                                        uint32_t* __restrict__ blockcnt, DevState* __restrict__ st) {
    const uint32_t n = st->done ? 0u : *nlist;
    if (blockIdx.x == 0 && threadIdx.x == 0 && (unsigned long long)n > (unsigned long long)gridDim.x * kPairTile) st->overflow = 1;  // (the list outgrew the host's bound: never silently)
    uint32_t       c = 0;
    const uint32_t base = blockIdx.x * kPairTile + threadIdx.x * kPairPer;
#pragma unroll
    for (int k = 0; k < kPairPer; ++k)
        if (base + k < n) c += ids[list[base + k]] != kInvalid;
    uint32_t total;
    pair_block_scan(c, &total);
    if (threadIdx.x == 0) blockcnt[blockIdx.x] = total;
}
__global__ __launch_bounds__(kPairThreads) void emit_write_list_kernel(const uint32_t* __restrict__ ids, const uint32_t* __restrict__ list, const uint32_t* __restrict__ nlist,
                                                                        const uint32_t* __restrict__ blockoff, const unsigned long long* __restrict__ chain, int which, uint64_t cap,
                                                                        unsigned long long* __restrict__ pairs, const PosBlock* __restrict__ blocks, uint32_t sb, uint32_t tb,
                                                                        uint32_t* __restrict__ pay = nullptr) {
    if (blockoff[blockIdx.x + 1] == blockoff[blockIdx.x]) return;
    const uint32_t n = *nlist, base = blockIdx.x * kPairTile + threadIdx.x * kPairPer;
    uint32_t       v[kPairPer], p[kPairPer], c = 0;
#pragma unroll
    for (int k = 0; k < kPairPer; ++k) {
        p[k] = base + k < n ? list[base + k] : 0u;
        v[k] = base + k < n ? ids[p[k]] : kInvalid;
        c += v[k] != kInvalid;
    }
    uint32_t       total;
    uint64_t       o     = chain[which] + blockoff[blockIdx.x] + pair_block_scan(c, &total);
    const uint32_t tmask = tb >= 32 ? 0xFFFFFFFFu : (1u << tb) - 1u;
#pragma unroll
    for (int k = 0; k < kPairPer; ++k) {
        if (v[k] != kInvalid) {
            if (o < cap) {
                if (blocks != nullptr) {
                    const uint4    r     = *reinterpret_cast<const uint4*>(blocks + (p[k] >> 6));
                    const uint64_t delim = ((uint64_t)r.w << 32) | r.z;
                    const uint32_t bit   = p[k] & 63u;
                    const uint64_t below = delim & ((1ull << bit) - 1ull);
                    const uint32_t sent = r.x + (uint32_t)__popcll(below), tok = below ? bit - (64u - (uint32_t)__clzll(below)) : p[k] - r.y;
                    if (pay != nullptr) {
                        reinterpret_cast<uint32_t*>(pairs)[o] = v[k];
                        pay[o]                                = (sent << tb) | (tok & tmask);
                    } else
                        pairs[o] = ((unsigned long long)v[k] << (sb + tb)) | ((unsigned long long)sent << tb) | (tok & tmask);
                } else {
                    pairs[o] = ((unsigned long long)v[k] << 32) | p[k];
                }
            }
            ++o;
        }
    }
}

// ---- stable LSD radix sort of (key, value) u32 pairs, 8 bits per pass ----
constexpr int kSortTile = 4096;  // elements per block per pass
__global__ __launch_bounds__(kBlock) void sort_hist_kernel(const uint32_t* __restrict__ keys, uint64_t n, int shift, uint32_t nblocks, uint32_t* __restrict__ ghist) {
    __shared__ uint32_t h[256];
    h[threadIdx.x] = 0;
    __syncthreads();
    const uint64_t t0 = (uint64_t)blockIdx.x * kSortTile;
    for (int r = 0; r < kSortTile / kBlock; ++r) {
        const uint64_t i = t0 + (uint64_t)r * kBlock + threadIdx.x;
        if (i < n) atomicAdd(&h[(keys[i] >> shift) & 255u], 1u);
    }
    __syncthreads();
    ghist[(uint64_t)threadIdx.x * nblocks + blockIdx.x] = h[threadIdx.x];  // digit-major so that one scan yields global offsets
}
__global__ __launch_bounds__(kBlock) void sort_scatter_kernel(const uint32_t* __restrict__ keys, const uint32_t* __restrict__ vals, uint64_t n, int shift, uint32_t nblocks,
                                                               const unsigned long long* __restrict__ goff, uint32_t* __restrict__ okeys, uint32_t* __restrict__ ovals) {
    __shared__ unsigned long long baseL[256];
    __shared__ uint32_t           wcnt[kBlock / kWave][256];
    baseL[threadIdx.x] = goff[(uint64_t)threadIdx.x * nblocks + blockIdx.x];
    for (int w = 0; w < kBlock / kWave; ++w) wcnt[w][threadIdx.x] = 0;
    __syncthreads();
    const uint32_t lane = threadIdx.x & (kWave - 1), wave = threadIdx.x / kWave;
    const uint64_t t0   = (uint64_t)blockIdx.x * kSortTile;
    for (int r = 0; r < kSortTile / kBlock; ++r) {
        const uint64_t i     = t0 + (uint64_t)r * kBlock + threadIdx.x;
        const bool     valid = i < n;
        const uint32_t k = valid ? keys[i] : 0u, v = valid ? vals[i] : 0u;
        const uint32_t d = (k >> shift) & 255u;
        // lanes of this wave with the same digit, in lane order (stable)
        uint64_t peers = __ballot(valid);
#pragma unroll
        for (int b = 0; b < 8; ++b) {
            const uint64_t m = __ballot((d >> b) & 1u);
            peers &= ((d >> b) & 1u) ? m : ~m;
        }
        const uint32_t rank = (uint32_t)__popcll(peers & ((1ull << lane) - 1ull));
        if (valid && rank == 0) wcnt[wave][d] = (uint32_t)__popcll(peers);
        __syncthreads();
        if (valid) {
            unsigned long long o = baseL[d] + rank;
            for (uint32_t w = 0; w < wave; ++w) o += wcnt[w][d];
            okeys[o] = k;
            ovals[o] = v;
        }
        __syncthreads();
        uint32_t add = 0;
        for (int w = 0; w < kBlock / kWave; ++w) {
            add += wcnt[w][threadIdx.x];
            wcnt[w][threadIdx.x] = 0;
        }
        baseL[threadIdx.x] += add;
        __syncthreads();
    }
}

// ---- stable LSD radix sort of packed 64-bit (result id << 32 | position) pairs by 8 bits of the id per pass ----
// The forward index groups 1.6 x 10^8 pairs by pattern with positions ascending: three stable passes over the id. sort_scatter_kernel above writes every
// element to its own address (two 4-byte arrays: 3 x 10^8 uncoalesced stores per pass, 4.7 ms); here a 4096-element tile is ranked stably — lanes of a
// wave with the same digit by ballots, waves and rows by an LDS table of their digit counts —, staged in LDS in output order and written as one
// coalesced run per digit.
constexpr int kS64Threads = 1024, kS64Per = 4, kS64Tile = kS64Threads * kS64Per, kS64Groups = kS64Per * (kS64Threads / kWave);  // 64 (row, wave) groups per tile
__global__ __launch_bounds__(kS64Threads) void sort64_hist_kernel(const unsigned long long* __restrict__ in, uint64_t n, int shift, uint32_t nblocks, uint32_t* __restrict__ ghist) {
    __shared__ uint32_t h[256];
    if (threadIdx.x < 256) h[threadIdx.x] = 0;
    __syncthreads();
    const uint64_t     t0 = (uint64_t)blockIdx.x * kS64Tile;
    unsigned long long x[kS64Per];
#pragma unroll
    for (int r = 0; r < kS64Per; ++r) {
        const uint64_t i = t0 + (uint64_t)r * kS64Threads + threadIdx.x;
        x[r]             = i < n ? in[i] : 0ull;
    }
#pragma unroll
    for (int r = 0; r < kS64Per; ++r)
        if (t0 + (uint64_t)r * kS64Threads + threadIdx.x < n) atomicAdd(&h[(uint32_t)(x[r] >> shift) & 255u], 1u);
    __syncthreads();
    if (threadIdx.x < 256) ghist[(uint64_t)threadIdx.x * nblocks + blockIdx.x] = h[threadIdx.x];  // digit-major so that one scan yields global offsets
}
// FINAL (the last pass): the elements do not leave as pairs again but as what the forward index is made of — (sentence, token) of the position, looked up in the
// position-block table, and optionally the id (sharded runs cut the references into runs by global id) — which saves one read and one write of all pairs.
template <bool FINAL>
__global__ __launch_bounds__(kS64Threads, kS64Threads / 128) void sort64_scatter_kernel(const unsigned long long* __restrict__ in, uint64_t n, int shift, uint32_t nblocks,
                                                                                         const unsigned long long* __restrict__ goff, unsigned long long* __restrict__ out,
                                                                                         const PosBlock* __restrict__ blocks = nullptr, uint32_t first_sentence = 0,
                                                                                         uint32_t* __restrict__ ref_sentence = nullptr, uint16_t* __restrict__ ref_token = nullptr,
                                                                                         uint32_t* __restrict__ sorted_id = nullptr, uint32_t sb = 0, uint32_t tb = 0) {
    // FINAL with blocks == NULL: the pairs are packed (emit_write_kernel): id << (sb + tb) | sentence << tb | token — unpack, no look-up
    __shared__ unsigned long long stgL[kS64Tile];
    __shared__ uint16_t           gcntL[kS64Groups][256];  // elements of digit d in group (row, wave); then: elements of digit d in the groups before it
    __shared__ uint32_t           histL[256], offL[256], wsumL[4];
    __shared__ unsigned long long gbaseL[256];
    for (uint32_t k = threadIdx.x; k < (uint32_t)(kS64Groups * 256 / 2); k += kS64Threads) reinterpret_cast<uint32_t*>(&gcntL[0][0])[k] = 0;
    if (threadIdx.x < 256) gbaseL[threadIdx.x] = goff[(uint64_t)threadIdx.x * nblocks + blockIdx.x];
    const uint32_t     lane = threadIdx.x & (kWave - 1), wave = threadIdx.x / kWave;
    const uint64_t     t0   = (uint64_t)blockIdx.x * kS64Tile;
    unsigned long long x[kS64Per];
    uint32_t           rank[kS64Per];
#pragma unroll
    for (int r = 0; r < kS64Per; ++r) {
        const uint64_t i = t0 + (uint64_t)r * kS64Threads + threadIdx.x;
        x[r]             = i < n ? in[i] : ~0ull;
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < kS64Per; ++r) {
        const bool     valid = t0 + (uint64_t)r * kS64Threads + threadIdx.x < n;
        const uint32_t d     = (uint32_t)(x[r] >> shift) & 255u;
        uint64_t       peers = __ballot(valid);  // lanes of this wave with the same digit, in lane order (stable)
#pragma unroll
        for (int b = 0; b < 8; ++b) {
            const uint64_t m = __ballot((d >> b) & 1u);
            peers &= ((d >> b) & 1u) ? m : ~m;
        }
        rank[r] = (uint32_t)__popcll(peers & ((1ull << lane) - 1ull));
        if (valid && rank[r] == 0) gcntL[r * (kS64Threads / kWave) + wave][d] = (uint16_t)__popcll(peers);
    }
    __syncthreads();
    if (threadIdx.x < 256) {  // digit t: counts of the 64 groups -> exclusive prefixes, in element order (row-major, then wave)
        uint32_t run = 0;
#pragma unroll 8
        for (int g = 0; g < kS64Groups; ++g) {
            const uint32_t c      = gcntL[g][threadIdx.x];
            gcntL[g][threadIdx.x] = (uint16_t)run;
            run += c;
        }
        histL[threadIdx.x] = run;
    }
    __syncthreads();
    {  // exclusive scan of the 256 digit totals (first four waves)
        uint32_t v = 0, incl = 0;
        if (threadIdx.x < 256) {
            v    = histL[threadIdx.x];
            incl = v;
            for (int off = 1; off < kWave; off <<= 1) {
                const uint32_t t = __shfl_up(incl, off, kWave);
                if ((int)lane >= off) incl += t;
            }
            if (lane == kWave - 1) wsumL[wave] = incl;
        }
        __syncthreads();
        if (threadIdx.x < 256) offL[threadIdx.x] = (wave > 0 ? wsumL[0] : 0u) + (wave > 1 ? wsumL[1] : 0u) + (wave > 2 ? wsumL[2] : 0u) + incl - v;
        __syncthreads();
    }
#pragma unroll
    for (int r = 0; r < kS64Per; ++r) {
        if (t0 + (uint64_t)r * kS64Threads + threadIdx.x < n) {
            const uint32_t d = (uint32_t)(x[r] >> shift) & 255u;
            stgL[offL[d] + gcntL[r * (kS64Threads / kWave) + wave][d] + rank[r]] = x[r];
        }
    }
    __syncthreads();
    const uint32_t cnt = (uint32_t)min((uint64_t)kS64Tile, n - t0);
    if (!FINAL) {
        for (uint32_t j = threadIdx.x; j < cnt; j += kS64Threads) {
            const unsigned long long y = stgL[j];
            const uint32_t           d = (uint32_t)(y >> shift) & 255u;
            out[gbaseL[d] + (j - offL[d])] = y;
        }
    } else if (blocks == nullptr) {
        const unsigned long long smask = (1ull << sb) - 1ull, tmask = (1ull << tb) - 1ull;
        for (uint32_t j = threadIdx.x; j < cnt; j += kS64Threads) {
            const unsigned long long y   = stgL[j];
            const uint32_t           d   = (uint32_t)(y >> shift) & 255u;
            const uint64_t           dst = gbaseL[d] + (j - offL[d]);
            ref_sentence[dst]            = first_sentence + (uint32_t)((y >> tb) & smask);
            ref_token[dst]               = (uint16_t)(y & tmask);
            if (sorted_id != nullptr) sorted_id[dst] = (uint32_t)(y >> (sb + tb));
        }
    } else {
        unsigned long long y[kS64Per];
        uint4              r[kS64Per];
#pragma unroll
        for (int q = 0; q < kS64Per; ++q) {  // all look-ups of the lane in flight together
            const uint32_t j = q * kS64Threads + threadIdx.x;
            y[q]             = j < cnt ? stgL[j] : 0ull;
            r[q]             = *reinterpret_cast<const uint4*>(blocks + ((uint32_t)y[q] >> 6));
        }
#pragma unroll
        for (int q = 0; q < kS64Per; ++q) {
            const uint32_t j = q * kS64Threads + threadIdx.x;
            if (j < cnt) {
                const uint32_t d = (uint32_t)(y[q] >> shift) & 255u, p = (uint32_t)y[q], bit = p & 63u;
                const uint64_t dst   = gbaseL[d] + (j - offL[d]);
                const uint64_t below = (((uint64_t)r[q].w << 32) | r[q].z) & ((1ull << bit) - 1ull);
                ref_sentence[dst]    = first_sentence + r[q].x + (uint32_t)__popcll(below);
                ref_token[dst]       = (uint16_t)(below ? bit - (64u - (uint32_t)__clzll(below)) : p - r[q].y);
                if (sorted_id != nullptr) sorted_id[dst] = (uint32_t)(y[q] >> 32);
            }
        }
    }
}

// ---- the same sort over SPLIT pairs: the id (what is left of it) and the reference travel in two arrays, and the id loses the byte a pass has sorted by ----
// The index needs the references grouped by id, not the ids: after the pass over id bits 0-7 nobody reads those bits again. Pass k reads TIN per element (u32, then
// u16 / u8 as the remaining id bits allow), sorts by its low byte and writes the rest (>> 8) as TOUT beside the 4-byte reference (sentence << tb | token); the histogram
// of a pass reads only the id array. Per element and three passes: 4 + 8 + 6, 2 + 6 + 5, 1 + 5 + 6 = 43 bytes against 3 x 24 of the packed form (161 M references: the
// sort's 3.6 ms of the indexed model's step).
// Tiles: a block takes kISuper consecutive tiles of kITile elements (one table column per block instead of one per 4096 elements: the tables of the packed sort are
// 10 M entries per pass, written and read as scattered words); inside a tile every WAVE owns 512 consecutive elements (kIRows rows of 64), so an element's rank among
// its digit is (count in the waves before) + (count in this wave's earlier rows: a running per-wave counter in LDS) + (rank among the row's lanes: ballots) — 16 counters
// per digit to scan instead of 64, and runs of 32 elements per digit and tile leaving for the output.
#ifndef COLIBRI_ISUPER
#define COLIBRI_ISUPER 4
#endif
constexpr int kIRows = 8, kITile = kS64Threads * kIRows, kISuper = COLIBRI_ISUPER, kIWaves = kS64Threads / kWave;
__device__ __forceinline__ uint64_t isort_peers(uint32_t d, bool valid) {
    uint64_t peers = __ballot(valid);
#pragma unroll
    for (int b = 0; b < 8; ++b) {
        const uint64_t m = __ballot((d >> b) & 1u);
        peers &= ((d >> b) & 1u) ? m : ~m;
    }
    return peers;
}
// ---- does the LDS serve the lanes of one returning add in lane order? (round 6) ------------------------------------------------------------------------------------
// emit_hot_write_kernel and isort_scatter_kernel take an element's rank among its wave's earlier elements of the same id / digit from ONE returning LDS add per lane.
// Rows follow each other in program order; inside a row the ranks are right iff lanes that hit the same counter are served in ascending lane order — true on this
// hardware, promised by no manual. Both kernels check their own results (every hot run; one row of eight of the sort); this kernel asks the question once per context,
// before the first forward index is built, with the patterns the two kernels produce (plain counters and two 16-bit counters per word; every lane on one counter, two
// counters, a few, a skewed and a uniform spread over 256): the rank from the add against the rank from matching the row's lanes with ballots. *bad != 0: the context
// matches every rank with ballots from the start (colibri_ctx::hot_off).
__device__ __forceinline__ uint64_t isort_peers(uint32_t d, bool valid);
__global__ __launch_bounds__(kS64Threads) void lds_order_selftest_kernel(uint32_t* __restrict__ bad) {
    constexpr int       kW = kS64Threads / kWave;
    __shared__ uint32_t plainL[kW][256], packedL[kW][128], seenL[kW][256];
    const uint32_t      lane = threadIdx.x & (kWave - 1), w = threadIdx.x / kWave;
    for (uint32_t k = threadIdx.x; k < (uint32_t)(kW * 256); k += kS64Threads) (&plainL[0][0])[k] = (&seenL[0][0])[k] = 0;
    for (uint32_t k = threadIdx.x; k < (uint32_t)(kW * 128); k += kS64Threads) (&packedL[0][0])[k] = 0;
    __syncthreads();
    bool wrong = false;
    for (uint32_t row = 0; row < 40; ++row) {
        uint32_t h = (lane * 0x9E3779B1u) ^ ((row + 17u * w + 131u * blockIdx.x) * 0x85EBCA6Bu);
        h ^= h >> 15;
        uint32_t d;
        switch (row % 5u) {
            case 0: d = row & 255u; break;                        // every lane on one counter
            case 1: d = (lane >> (row & 3u)) & 1u; break;         // two counters, in runs of 1 / 2 / 4 / 8 lanes
            case 2: d = h % 7u; break;                            // a few
            case 3: d = (h & 3u) ? (h >> 8) & 3u : (h >> 8) & 255u; break;  // skewed
            default: d = h & 255u;                                // uniform
        }
        const bool     valid = ((h >> 20) & 15u) != 0u || (row & 1u) == 0u;  // (some rows with idle lanes)
        const uint64_t peers = isort_peers(d, valid);
        const uint32_t below = (uint32_t)__popcll(peers & ((1ull << lane) - 1ull));
        const uint32_t seen  = seenL[w][d];
        __builtin_amdgcn_wave_barrier();
        if (valid && below == 0) seenL[w][d] = seen + (uint32_t)__popcll(peers);
        __builtin_amdgcn_wave_barrier();
        const uint32_t sh = 16u * (d & 1u);
        const uint32_t a  = valid ? atomicAdd(&plainL[w][d], 1u) : 0u;
        const uint32_t b  = valid ? (atomicAdd(&packedL[w][d >> 1], 1u << sh) >> sh) & 0xFFFFu : 0u;
        wrong             = wrong || (valid && (a != seen + below || b != seen + below));
    }
    if (wrong) *bad = 1u;
}

template <typename TIN>
__global__ __launch_bounds__(kS64Threads) void isort_hist_kernel(const TIN* __restrict__ dig, uint64_t n, uint32_t nblocks, uint32_t* __restrict__ ghist) {
    __shared__ uint32_t h[256];
    if (threadIdx.x < 256) h[threadIdx.x] = 0;
    __syncthreads();
    const uint64_t t0 = (uint64_t)blockIdx.x * kITile * kISuper, t1 = min(n, t0 + (uint64_t)kITile * kISuper);
    // (plain LDS atomics: matching the lanes of a row by digit with ballots first — one atomic per distinct digit — cost 240 cycles per row of 64 and made this pass
    // instruction-bound at 0.29 ms whatever the element size; the conflicts of the skewed high byte cost less)
    for (uint64_t i0 = t0; i0 < t1; i0 += (uint64_t)kS64Threads * 8) {  // eight rows of the block in flight
        uint32_t x[8];
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            const uint64_t i = i0 + (uint64_t)r * kS64Threads + threadIdx.x;
            x[r]             = i < t1 ? (uint32_t)dig[i] : 0xFFFFFFFFu;
        }
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            // dense ids follow the classes' frequency ranks: the high byte is 0 for half of all references, and 34 lanes of a row adding to one counter serialise. Two
            // rounds peel the digit of the first lane still waiting (one atomic for the whole group); what is left adds for itself
            bool           todo = i0 + (uint64_t)r * kS64Threads + threadIdx.x < t1;
            const uint32_t d    = x[r] & 255u;
#pragma unroll
            for (int round = 0; round < 2; ++round) {
                const uint64_t waiting = __ballot(todo);
                if (!waiting) break;
                const uint32_t lead = (uint32_t)__builtin_ctzll(waiting);
                const uint32_t dl   = (uint32_t)__shfl((int)d, (int)lead, kWave);
                const uint64_t grp  = __ballot(todo && d == dl);
                if ((threadIdx.x & (kWave - 1)) == lead) atomicAdd(&h[dl], (uint32_t)__popcll(grp));
                todo = todo && d != dl;
            }
            if (todo) atomicAdd(&h[d], 1u);
        }
    }
    __syncthreads();
    if (threadIdx.x < 256) ghist[(uint64_t)threadIdx.x * nblocks + blockIdx.x] = h[threadIdx.x];
}
template <typename TIN, typename TOUT, bool FINAL>
__global__ __launch_bounds__(kS64Threads, kS64Threads / 128) void isort_scatter_kernel(const TIN* __restrict__ dig, const uint32_t* __restrict__ pay, uint64_t n, uint32_t nblocks,
                                                                                        const unsigned long long* __restrict__ goff, uint32_t* __restrict__ out_pay,
                                                                                        TOUT* __restrict__ out_dig, uint32_t first_sentence, uint32_t* __restrict__ ref_sentence,
                                                                                        uint16_t* __restrict__ ref_token, uint32_t tb,
                                                                                        uint64_t hot_below = 0, uint64_t hot_n = 0 /* FINAL: the references of the hot unigrams already
                                                                                            lie at [hot_below, hot_below + hot_n) (emit_hot_write_kernel): what sorts behind moves up */,
                                                                                        uint32_t exact = 1 /* 0: ranks from returning LDS adds, spot-checked */,
                                                                                        unsigned long long* flags = nullptr /* the pair chain: [kChainDisorder] */) {
    static_assert((kIRows & (kIRows - 1)) == 0, "the checked row is the tile number modulo the rows");
    bool disorder = false;
    __shared__ uint32_t           stgP[kITile], stgD[kITile];
    __shared__ uint16_t           wcntL[kIWaves][256];  // elements of digit d wave w has seen in the tile so far; then: those of the waves before it
    __shared__ uint32_t           histL[256], offL[256], wsumL[4];
    __shared__ unsigned long long gbaseL[256];
    if (threadIdx.x < 256) gbaseL[threadIdx.x] = goff[(uint64_t)threadIdx.x * nblocks + blockIdx.x];
    const uint32_t lane = threadIdx.x & (kWave - 1), wave = threadIdx.x / kWave;
    const uint32_t tmask = tb >= 32 ? 0xFFFFFFFFu : (1u << tb) - 1u;
    const uint64_t s0 = (uint64_t)blockIdx.x * kITile * kISuper, s1 = min(n, s0 + (uint64_t)kITile * kISuper);
    for (uint64_t t0 = s0; t0 < s1; t0 += kITile) {
        for (uint32_t k = threadIdx.x; k < (uint32_t)(kIWaves * 256 / 2); k += kS64Threads) reinterpret_cast<uint32_t*>(&wcntL[0][0])[k] = 0;
        const uint32_t tileno = (uint32_t)(t0 / kITile);
        uint32_t       x[kIRows], y[kIRows], rank[kIRows];
#pragma unroll
        for (int r = 0; r < kIRows; ++r) {  // the wave's 512 consecutive elements, row by row
            const uint64_t i = t0 + (uint64_t)wave * (kWave * kIRows) + (uint64_t)r * kWave + lane;
            x[r]             = i < s1 ? (uint32_t)dig[i] : 0u;
            y[r]             = i < s1 ? pay[i] : 0u;
        }
        __syncthreads();  // (the counters are clear; the staging area of the tile before has left)
#pragma unroll
        for (int r = 0; r < kIRows; ++r) {
            const bool     valid = t0 + (uint64_t)wave * (kWave * kIRows) + (uint64_t)r * kWave + lane < s1;
            const uint32_t d     = x[r] & 255u;
            if (exact) {  // ranks by matching the row's lanes (eight ballots): exact by construction
                const uint64_t peers = isort_peers(d, valid);
                const uint32_t below = (uint32_t)__popcll(peers & ((1ull << lane) - 1ull));
                const uint32_t seen  = wcntL[wave][d];
                rank[r]              = seen + below;
                __builtin_amdgcn_wave_barrier();  // (every lane of the row has read its counter before the row's leaders advance them)
                if (valid && below == 0) wcntL[wave][d] = (uint16_t)(seen + (uint32_t)__popcll(peers));
                __builtin_amdgcn_wave_barrier();
            } else {
                // ranks from ONE returning LDS add per lane (two 16-bit counters share a word): the rows are served in program order; lanes of one instruction that
                // hit the same counter are served in lane order on this hardware (see emit_hot_write_kernel) — the three scatter passes 1.80 -> 1.53 ms. What no manual
                // promises is checked: one row of every eight (a different one tile by tile) is also matched with ballots and must agree, and the hot unigrams' runs
                // are checked reference by reference; any disagreement raises chain[kChainDisorder] and the run repeats with exact == 1
                const uint32_t sh  = 16u * (d & 1u);
                const uint32_t old = valid ? atomicAdd(reinterpret_cast<uint32_t*>(&wcntL[wave][0]) + (d >> 1), 1u << sh) : 0u;
                rank[r]            = (old >> sh) & 0xFFFFu;
                if ((uint32_t)r == (tileno & (uint32_t)(kIRows - 1))) {
                    const uint64_t peers  = isort_peers(d, valid);
                    const uint32_t below  = (uint32_t)__popcll(peers & ((1ull << lane) - 1ull));
                    const uint32_t leader = peers ? (uint32_t)__builtin_ctzll(peers) : lane;
                    const uint32_t lr     = (uint32_t)__shfl((int)rank[r], (int)leader, kWave);
                    disorder              = disorder || (valid && rank[r] != lr + below);
                }
            }
        }
        __syncthreads();
        if (threadIdx.x < 256) {  // digit t: the waves' counts -> exclusive prefixes over the waves
            uint32_t run = 0;
#pragma unroll
            for (int w = 0; w < kIWaves; ++w) {
                const uint32_t c        = wcntL[w][threadIdx.x];
                wcntL[w][threadIdx.x] = (uint16_t)run;
                run += c;
            }
            histL[threadIdx.x] = run;
        }
        __syncthreads();
        {  // exclusive scan of the 256 digit totals (first four waves)
            uint32_t v = 0, incl = 0;
            if (threadIdx.x < 256) {
                v    = histL[threadIdx.x];
                incl = v;
                for (int off = 1; off < kWave; off <<= 1) {
                    const uint32_t t = __shfl_up(incl, off, kWave);
                    if ((int)lane >= off) incl += t;
                }
                if (lane == kWave - 1) wsumL[wave] = incl;
            }
            __syncthreads();
            if (threadIdx.x < 256) offL[threadIdx.x] = (wave > 0 ? wsumL[0] : 0u) + (wave > 1 ? wsumL[1] : 0u) + (wave > 2 ? wsumL[2] : 0u) + incl - v;
            __syncthreads();
        }
#pragma unroll
        for (int r = 0; r < kIRows; ++r) {
            if (t0 + (uint64_t)wave * (kWave * kIRows) + (uint64_t)r * kWave + lane < s1) {
                const uint32_t d = x[r] & 255u, o = offL[d] + wcntL[wave][d] + rank[r];
                stgP[o] = y[r];
                stgD[o] = x[r];
            }
        }
        __syncthreads();
        const uint32_t cnt = (uint32_t)min((uint64_t)kITile, s1 - t0);
        for (uint32_t j = threadIdx.x; j < cnt; j += kS64Threads) {
            const uint32_t xx = stgD[j], yy = stgP[j], d = xx & 255u;
            const uint64_t dst = gbaseL[d] + (j - offL[d]);
            if (FINAL) {
                const uint64_t fin = dst + (dst >= hot_below ? hot_n : 0ull);
                ref_sentence[fin]  = first_sentence + (tb >= 32 ? 0u : yy >> tb);
                ref_token[fin]     = (uint16_t)(yy & tmask);
            } else {
                out_pay[dst] = yy;
                out_dig[dst] = (TOUT)(xx >> 8);
            }
        }
        __syncthreads();
        if (threadIdx.x < 256) gbaseL[threadIdx.x] += histL[threadIdx.x];  // the next tile of the block continues every digit's run
    }
    if (disorder && flags != nullptr) flags[kChainDisorder] = 1ull;
}

// position -> (sentence, token): sentence = first_sentence + #delimiters before the position (empty sentences are numbered,
// reference src/pattern.cpp:1947-1958); token = offset inside the sentence, truncated to u16 like IndexReference (datatypes.h:36)
__global__ __launch_bounds__(kBlock) void refs_kernel(const uint32_t* __restrict__ pos, uint64_t n, const uint32_t* __restrict__ delimpos, uint32_t ndelim, uint32_t first_sentence,
                                                       uint32_t* __restrict__ ref_sentence, uint16_t* __restrict__ ref_token) {
    for (uint64_t j = (uint64_t)blockIdx.x * kBlock + threadIdx.x; j < n; j += (uint64_t)gridDim.x * kBlock) {
        const uint32_t p  = pos[j];
        uint32_t       lo = 0, hi = ndelim;  // first delimiter position > p
        while (lo < hi) {
            const uint32_t mid = (lo + hi) >> 1;
            if (delimpos[mid] < p)
                lo = mid + 1;
            else
                hi = mid;
        }
        const uint32_t begin = lo ? delimpos[lo - 1] + 1 : 0;
        ref_sentence[j]      = first_sentence + lo;
        ref_token[j]         = (uint16_t)(p - begin);
    }
}

__global__ __launch_bounds__(kBlock) void position_blocks_kernel(const uint32_t* __restrict__ cls, const uint32_t* __restrict__ delimpos, uint32_t ndelim, uint32_t npos,
                                                                  PosBlock* __restrict__ blocks) {
    const uint32_t nblk = (npos + 63) / 64, lane = threadIdx.x & (kWave - 1);
    for (uint32_t b = blockIdx.x * (kBlock / kWave) + threadIdx.x / kWave; b < nblk; b += gridDim.x * (kBlock / kWave)) {
        const uint32_t p = b * 64 + lane;
        const uint64_t m = __ballot(p < npos && cls[p] == 0u);
        if (lane == 0) {
            uint32_t lo = 0, hi = ndelim;  // delimiters before the block's first position
            while (lo < hi) {
                const uint32_t mid = (lo + hi) >> 1;
                if (delimpos[mid] < b * 64)
                    lo = mid + 1;
                else
                    hi = mid;
            }
            PosBlock x;
            x.sent_before = lo;
            x.sent_start  = lo ? delimpos[lo - 1] + 1 : 0u;
            x.delim       = m;
            blocks[b]     = x;
        }
    }
}

// =================================================================================================
// 7. sentence-sharded multi-GPU: the per-order exchange step
//    Every rank counts its own shard; because survivor ids are GLOBAL (assigned by the owner rank of each key), the 64-bit
//    keys of all ranks are directly comparable. Candidates (key, local count) travel to owner = owner_of(key) (blocks of the top byte of mix64(key)), the
//    owner sums exact global counts, applies the threshold, hands out global survivor ids and names ONE exporting rank
//    (the lowest rank that saw the pattern: it has the bytes); the replies travel back and are applied to the local table.
// =================================================================================================
// pass 1: occupied slots of the local table -> unpartitioned (key, count, slot) + per-owner histogram
__global__ __launch_bounds__(kBlock) void shard_extract_kernel(const Slot* __restrict__ table, uint32_t cap, uint32_t world, unsigned long long* __restrict__ keys,
                                                                uint32_t* __restrict__ counts, uint32_t* __restrict__ slots, uint32_t* __restrict__ ncand,
                                                                uint32_t* __restrict__ owner_hist, const uint32_t* __restrict__ nsrc, uint32_t* __restrict__ aux) {
    __shared__ uint32_t histL[64];
    __shared__ uint32_t baseL;
    if (threadIdx.x < 64) histL[threadIdx.x] = 0;
    __syncthreads();
    const uint32_t ntiles = (cap + kPruneTile - 1) / kPruneTile;
    for (uint32_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const uint32_t t0 = tile * kPruneTile;
        uint32_t       used = 0;
        Slot           sl[kPrunePer];
#pragma unroll
        for (int k = 0; k < kPrunePer; ++k) {
            const uint32_t i = t0 + k * kBlock + threadIdx.x;
            if (i < cap) {
                sl[k] = table[i];
                if (sl[k].key != kEmptyKey) used |= 1u << k;
            }
        }
        uint32_t       total;
        const uint32_t excl = block_exclusive_scan((uint32_t)__popc(used), &total);
        if (threadIdx.x == 0) baseL = total ? atomicAdd(ncand, total) : 0;
        __syncthreads();
        uint32_t o = baseL + excl;
#pragma unroll
        for (int k = 0; k < kPrunePer; ++k) {
            if (used & (1u << k)) {
                keys[o]   = sl[k].key;
                counts[o] = sl[k].count;
                slots[o]  = t0 + k * kBlock + threadIdx.x;
                if (aux != nullptr) aux[o] = nsrc[t0 + k * kBlock + threadIdx.x];
                atomicAdd(&histL[owner_of(sl[k].key, world)], 1u);
                ++o;
            }
        }
        __syncthreads();
    }
    if (threadIdx.x < world && histL[threadIdx.x]) atomicAdd(&owner_hist[threadIdx.x], histL[threadIdx.x]);
}
// pass 2: scatter by owner (owner_off = exclusive scan of the histogram; cursor = zeroed)
__global__ __launch_bounds__(kBlock) void shard_partition_kernel(const unsigned long long* __restrict__ keys, const uint32_t* __restrict__ counts, const uint32_t* __restrict__ slots,
                                                                  uint32_t n, uint32_t world, const uint32_t* __restrict__ owner_off, uint32_t* __restrict__ cursor,
                                                                  unsigned long long* __restrict__ okeys, uint32_t* __restrict__ ocounts, uint32_t* __restrict__ oslots,
                                                                  const uint32_t* __restrict__ aux, uint32_t* __restrict__ oaux) {
    // a block ranks its 1024 records per owner in LDS and reserves ONE range per owner (a single cursor word takes ~88 M atomics/s)
    __shared__ uint32_t histL[64], baseL[64];
    const uint32_t      ntiles = (n + kEmitTile - 1) / kEmitTile;
    for (uint32_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        if (threadIdx.x < 64) histL[threadIdx.x] = 0;
        __syncthreads();
        unsigned long long k[kEmitPer];
        uint32_t           own[kEmitPer], rank[kEmitPer];
#pragma unroll
        for (int q = 0; q < kEmitPer; ++q) {
            const uint32_t j = tile * kEmitTile + q * kBlock + threadIdx.x;
            own[q]           = kInvalid;
            if (j < n) {
                k[q]    = keys[j];
                own[q]  = owner_of(k[q], world);
                rank[q] = atomicAdd(&histL[own[q]], 1u);
            }
        }
        __syncthreads();
        if (threadIdx.x < world) baseL[threadIdx.x] = histL[threadIdx.x] ? owner_off[threadIdx.x] + atomicAdd(&cursor[threadIdx.x], histL[threadIdx.x]) : 0;
        __syncthreads();
#pragma unroll
        for (int q = 0; q < kEmitPer; ++q) {
            const uint32_t j = tile * kEmitTile + q * kBlock + threadIdx.x;
            if (own[q] != kInvalid) {
                const uint32_t d = baseL[own[q]] + rank[q];
                okeys[d]         = k[q];
                ocounts[d]       = counts[j];
                oslots[d]        = slots[j];
                if (aux != nullptr) oaux[d] = aux[j];
            }
        }
        __syncthreads();
    }
}
// owner: sum the received records per key; remember the lowest contributing rank (it will export the pattern).
// The slot comes from a SALTED mix of the key: every key this rank owns has the top byte of its plain mix inside the rank's block of
// owner_of(), so the plain mix would send all of them to 1/world of the table (measured: 75 ms per call at 10 M tokens per rank, world 2).
constexpr uint64_t kOwnerTableSalt = 0xD6E8FEB86659FD93ull;
__global__ __launch_bounds__(kBlock) void shard_merge_kernel(const unsigned long long* __restrict__ keys, const uint32_t* __restrict__ counts, uint32_t n, uint32_t world,
                                                              const uint32_t* __restrict__ src_off /*[world+1]*/, Slot* __restrict__ table, uint32_t* __restrict__ minrank,
                                                              uint32_t* __restrict__ slot_out, DevState* __restrict__ st, const uint32_t* __restrict__ aux,
                                                              uint32_t* __restrict__ onsrc) {
    const uint32_t cap = st->cap;
    uint32_t       ins = 0;
    for (uint32_t j = blockIdx.x * kBlock + threadIdx.x; j < n; j += gridDim.x * kBlock) {
        uint32_t src = 0;
        while (src + 1 < world && j >= src_off[src + 1]) ++src;
        uint32_t       won = 0;
        const uint64_t k   = keys[j];
        const uint32_t s   = table_find_or_insert(table, cap, k, mix64(k ^ kOwnerTableSalt), 0u, counts[j], &won, st);
        ins += won;
        slot_out[j] = s;
        if (s != kInvalid) {
            atomicMin(&minrank[s], src);
            if (aux != nullptr && aux[j]) atomicAdd(&onsrc[s], aux[j]);
        }
    }
    __shared__ uint32_t redL[kBlock / kWave];
    for (int off = 32; off > 0; off >>= 1) ins += __shfl_down(ins, off, kWave);
    if ((threadIdx.x & (kWave - 1)) == 0) redL[threadIdx.x / kWave] = ins;
    __syncthreads();
    if (threadIdx.x == 0) {
        const uint32_t f = redL[0] + redL[1] + redL[2] + redL[3];
        if (f) atomicAdd(&st->found, f);
    }
}
// owner: how many keys reach the threshold
__global__ __launch_bounds__(kBlock) void shard_owner_count_kernel(const Slot* __restrict__ table, DevState* __restrict__ st, uint32_t threshold,
                                                                    const uint32_t* __restrict__ onsrc, uint32_t minsrc) {
    const uint32_t cap = st->cap;
    uint32_t       k   = 0;
    for (uint32_t i = blockIdx.x * kBlock + threadIdx.x; i < cap; i += gridDim.x * kBlock) {
        const Slot s = table[i];
        k += (s.key != kEmptyKey && s.count >= threshold && (onsrc == nullptr || onsrc[i] >= minsrc));
    }
    for (int off = 32; off > 0; off >>= 1) k += __shfl_down(k, off, kWave);
    if ((threadIdx.x & (kWave - 1)) == 0 && k) atomicAdd(&st->kept, k);
}
// owner: hand out global survivor ids gid_base .. gid_base+kept-1 (stored in the slot's rep half; kInvalid = pruned)
__global__ __launch_bounds__(kBlock) void shard_owner_assign_kernel(Slot* __restrict__ table, DevState* __restrict__ st, uint32_t threshold, uint32_t gid_base,
                                                                     const uint32_t* __restrict__ onsrc, uint32_t minsrc) {
    __shared__ uint32_t baseL;
    const uint32_t      cap    = st->cap;
    const uint32_t      ntiles = (cap + kPruneTile - 1) / kPruneTile;
    for (uint32_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const uint32_t t0 = tile * kPruneTile;
        uint32_t       keep = 0, used = 0;
#pragma unroll
        for (int k = 0; k < kPrunePer; ++k) {
            const uint32_t i = t0 + k * kBlock + threadIdx.x;
            if (i < cap) {
                const Slot s = table[i];
                if (s.key != kEmptyKey) {
                    used |= 1u << k;
                    if (s.count >= threshold && (onsrc == nullptr || onsrc[i] >= minsrc)) keep |= 1u << k;
                }
            }
        }
        uint32_t       total;
        const uint32_t excl = block_exclusive_scan((uint32_t)__popc(keep), &total);
        if (threadIdx.x == 0) baseL = total ? atomicAdd(&st->valid, total) : 0;  // st->valid doubles as the owner's id cursor
        __syncthreads();
        uint32_t g = gid_base + baseL + excl;
#pragma unroll
        for (int k = 0; k < kPrunePer; ++k) {
            if (used & (1u << k)) {
                const uint32_t i = t0 + k * kBlock + threadIdx.x;
                table[i].rep     = (keep & (1u << k)) ? g++ : kInvalid;
            }
        }
        __syncthreads();
    }
}
constexpr uint32_t kExportBit = 0x80000000u;  // in a reply: this rank exports the pattern (global ids stay below 2^31)
__global__ __launch_bounds__(kBlock) void shard_reply_kernel(const uint32_t* __restrict__ slot_out, uint32_t n, uint32_t world, const uint32_t* __restrict__ src_off,
                                                              const Slot* __restrict__ table, const uint32_t* __restrict__ minrank, uint32_t* __restrict__ reply_gid,
                                                              uint32_t* __restrict__ reply_cnt) {
    for (uint32_t j = blockIdx.x * kBlock + threadIdx.x; j < n; j += gridDim.x * kBlock) {
        uint32_t src = 0;
        while (src + 1 < world && j >= src_off[src + 1]) ++src;
        const uint32_t s = slot_out[j];
        uint32_t       g = kInvalid, cnt = 0;
        if (s != kInvalid) {
            const Slot sl = table[s];
            cnt           = sl.count;
            if (sl.rep != kInvalid) g = sl.rep | (minrank[s] == src ? kExportBit : 0u);
        }
        reply_gid[j] = g;
        reply_cnt[j] = cnt;
    }
}
// contributor: tag the local slots with the global survivor id (the resolve kernel then writes ids per position) and append
// the patterns this rank exports (local representative position + GLOBAL count) to its result arrays
__global__ __launch_bounds__(kBlock) void shard_apply_kernel(const uint32_t* __restrict__ slots, const uint32_t* __restrict__ reply_gid, const uint32_t* __restrict__ reply_cnt,
                                                              uint32_t n, Slot* __restrict__ table, DevState* __restrict__ st, uint32_t* __restrict__ res_rep,
                                                              uint32_t* __restrict__ res_cnt, uint32_t* __restrict__ res_gid, uint32_t res_cap, uint32_t* __restrict__ mark,
                                                              uint32_t markbit) {
    __shared__ uint32_t baseL;
    const uint32_t      res_base = st->res_total;
    const uint32_t      ntiles   = (n + kEmitTile - 1) / kEmitTile;
    for (uint32_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const uint32_t j0 = tile * kEmitTile + threadIdx.x * kEmitPer;
        uint32_t       g[kEmitPer], nexp = 0;
#pragma unroll
        for (int k = 0; k < kEmitPer; ++k) {
            g[k] = (j0 + k < n) ? reply_gid[j0 + k] : kInvalid;
            nexp += (g[k] != kInvalid) && (g[k] & kExportBit);
        }
        uint32_t       total;
        const uint32_t excl = block_exclusive_scan(nexp, &total);
        if (threadIdx.x == 0) baseL = total ? atomicAdd(&st->kept, total) : 0;  // one reservation per 1024 replies
        __syncthreads();
        uint32_t r = res_base + baseL + excl;
#pragma unroll
        for (int k = 0; k < kEmitPer; ++k) {
            if (j0 + k < n) {
                const uint32_t slot = slots[j0 + k];
                if (g[k] != kInvalid && (g[k] & kExportBit)) {
                    if (r < res_cap) {
                        const uint32_t rp = table[slot].rep;
                        res_rep[r]        = rp;
                        res_cnt[r]        = reply_cnt[j0 + k];
                        res_gid[r]        = g[k] & ~kExportBit;
                        if (mark != nullptr) atomicOr(&mark[rp], markbit);  // this position represents a globally-unique surviving pattern
                    } else {
                        st->overflow = 1;
                    }
                    ++r;
                }
                table[slot].count = (g[k] == kInvalid) ? 0u : (kKeptFlag | (g[k] & ~kExportBit));
            }
        }
        __syncthreads();
    }
}
// sharded radix path: the local candidates are the non-empty entries of the sparse per-bin arrays (bin_count with threshold 1);
// the handle that travels with a candidate is its sparse index
// sharded radix path: replies -> global id per sparse index (+ exports on the exporting rank)
__global__ __launch_bounds__(kBlock) void shard_apply_sparse_kernel(const uint32_t* __restrict__ handles, const uint32_t* __restrict__ reply_gid,
                                                                     const uint32_t* __restrict__ reply_cnt, uint32_t n, const uint32_t* __restrict__ sp_rep,
                                                                     uint32_t* __restrict__ gid_of_sparse, DevState* __restrict__ st, uint32_t* __restrict__ res_rep,
                                                                     uint32_t* __restrict__ res_cnt, uint32_t* __restrict__ res_gid, uint32_t res_cap, uint32_t* __restrict__ mark,
                                                                     uint32_t markbit, const uint32_t* __restrict__ list /* item index -> corpus position, or NULL */) {
    __shared__ uint32_t baseL;
    const uint32_t      res_base = st->res_total;
    const uint32_t      ntiles   = (n + kEmitTile - 1) / kEmitTile;
    for (uint32_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const uint32_t j0 = tile * kEmitTile + threadIdx.x * kEmitPer;
        uint32_t       g[kEmitPer], nexp = 0;
#pragma unroll
        for (int k = 0; k < kEmitPer; ++k) {
            g[k] = (j0 + k < n) ? reply_gid[j0 + k] : kInvalid;
            nexp += (g[k] != kInvalid) && (g[k] & kExportBit);
        }
        uint32_t       total;
        const uint32_t excl = block_exclusive_scan(nexp, &total);
        if (threadIdx.x == 0) baseL = total ? atomicAdd(&st->kept, total) : 0;
        __syncthreads();
        uint32_t r = res_base + baseL + excl;
#pragma unroll
        for (int k = 0; k < kEmitPer; ++k) {
            if (j0 + k < n) {
                const uint32_t h = handles[j0 + k];
                if (g[k] != kInvalid && (g[k] & kExportBit)) {
                    if (r < res_cap) {
                        const uint32_t rp = list != nullptr ? list[sp_rep[h]] : sp_rep[h];
                        res_rep[r]        = rp;
                        res_cnt[r]        = reply_cnt[j0 + k];
                        res_gid[r]        = g[k] & ~kExportBit;
                        if (mark != nullptr) atomicOr(&mark[rp], markbit);
                    } else {
                        st->overflow = 1;
                    }
                    ++r;
                }
                gid_of_sparse[h] = (g[k] == kInvalid) ? kInvalid : (g[k] & ~kExportBit);
            }
        }
        __syncthreads();
    }
}
// sharded order 1 on the class-indexed arrays: after the all-reduce, every rank knows the global count of every class and the
// lowest rank that saw it (that rank exports the unigram; its representative position also counts as the distinct source)
__global__ __launch_bounds__(kBlock) void shard_uni_minrank_kernel(const uint32_t* __restrict__ cnt_local, uint32_t nclasses, uint32_t rank, uint32_t* __restrict__ minrank) {
    for (uint32_t c = blockIdx.x * kBlock + threadIdx.x; c < nclasses; c += gridDim.x * kBlock) minrank[c] = cnt_local[c] ? rank : 0x7FFFFFFFu;
}
__global__ __launch_bounds__(kBlock) void shard_uni_finish_kernel(const uint32_t* __restrict__ cnt_global, const uint32_t* __restrict__ minrank, const uint32_t* __restrict__ rep1,
                                                                   uint32_t nclasses, uint32_t rank, uint32_t threshold, DevState* __restrict__ st,
                                                                   uint32_t* __restrict__ res_rep, uint32_t* __restrict__ res_cnt, uint32_t* __restrict__ res_gid, uint32_t res_cap,
                                                                   uint32_t* __restrict__ mark) {
    __shared__ uint32_t baseL, redL[2][kBlock / kWave];
    const uint32_t      res_base = st->res_total;
    const uint32_t      ntiles   = (nclasses + kPruneTile - 1) / kPruneTile;
    uint32_t            nfound = 0, nkept = 0;
    for (uint32_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const uint32_t c0 = tile * kPruneTile + threadIdx.x * kPrunePer;
        uint32_t       v[kPrunePer], k = 0;
        bool           mine[kPrunePer];
#pragma unroll
        for (int q = 0; q < kPrunePer; ++q) {
            v[q]    = (c0 + q < nclasses) ? cnt_global[c0 + q] : 0u;
            mine[q] = v[q] >= threshold && minrank[c0 + q] == rank;
            nfound += v[q] != 0;
            nkept += v[q] >= threshold;
            k += mine[q];
        }
        uint32_t       total;
        const uint32_t excl = block_exclusive_scan(k, &total);
        if (threadIdx.x == 0) baseL = total ? atomicAdd(&st->kept, total) : 0;
        __syncthreads();
        uint32_t r = res_base + baseL + excl;
#pragma unroll
        for (int q = 0; q < kPrunePer; ++q) {
            if (mine[q]) {
                if (r < res_cap) {
                    res_rep[r] = rep1 != nullptr ? rep1[c0 + q] : c0 + q;  // no representative recorded: the class itself (kMaskFromClass export)
                    res_cnt[r] = v[q];
                    res_gid[r] = c0 + q;  // the global id of a unigram is its class id
                    if (mark != nullptr) atomicOr(&mark[rep1[c0 + q]], 2u);  // bit 1 = order 1
                } else {
                    st->overflow = 1;
                }
                ++r;
            }
        }
        __syncthreads();
    }
    for (int off = 32; off > 0; off >>= 1) {
        nfound += __shfl_down(nfound, off, kWave);
        nkept += __shfl_down(nkept, off, kWave);
    }
    if ((threadIdx.x & (kWave - 1)) == 0) {
        redL[0][threadIdx.x / kWave] = nfound;
        redL[1][threadIdx.x / kWave] = nkept;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        const uint32_t f = redL[0][0] + redL[0][1] + redL[0][2] + redL[0][3], kk = redL[1][0] + redL[1][1] + redL[1][2] + redL[1][3];
        if (f) atomicAdd(&st->found, f);
        if (kk) atomicAdd(&st->admitted, kk);  // (admitted doubles as the GLOBAL kept count of this pass)
    }
}
// intermediate skipgram level: the replies only carry the global id of the interned pair
__global__ __launch_bounds__(kBlock) void shard_tag_kernel(const uint32_t* __restrict__ slots, const uint32_t* __restrict__ reply_gid, uint32_t n, Slot* __restrict__ table) {
    for (uint32_t j = blockIdx.x * kBlock + threadIdx.x; j < n; j += gridDim.x * kBlock) {
        const uint32_t g      = reply_gid[j];
        table[slots[j]].count = (g == kInvalid) ? 0u : (kKeptFlag | (g & ~kExportBit));
    }
}
// sharded indexed skipgrams: a position counts as a distinct source iff it is the (globally unique) exported representative of
// its surviving n-gram (mark bit n, set by shard_apply_kernel on the exporting rank)
__global__ __launch_bounds__(kBlock) void shard_skip_sources_kernel(const uint32_t* __restrict__ mark, uint32_t markbit, const uint32_t* __restrict__ slot_of,
                                                                     uint32_t* __restrict__ nsrc, uint32_t npos) {
    for (uint32_t i = blockIdx.x * kBlock + threadIdx.x; i < npos; i += gridDim.x * kBlock) {
        const uint32_t s = slot_of[i];
        if (s != kInvalid && (mark[i] & markbit)) atomicAdd(&nsrc[s], 1u);
    }
}
// run-length boundaries of the sorted (global id, position) pairs: one entry per distinct id
__global__ __launch_bounds__(kBlock) void rle_count_kernel(const uint32_t* __restrict__ ids, uint64_t n, uint32_t* __restrict__ blockcnt) {
    const uint64_t base = (uint64_t)blockIdx.x * kEmitTile + (uint64_t)threadIdx.x * kEmitPer;
    uint32_t       c    = 0;
#pragma unroll
    for (int k = 0; k < kEmitPer; ++k) {
        const uint64_t j = base + k;
        c += (j < n) && (j == 0 || ids[j] != ids[j - 1]);
    }
    uint32_t total;
    block_exclusive_scan(c, &total);
    if (threadIdx.x == 0) blockcnt[blockIdx.x] = total;
}
__global__ __launch_bounds__(kBlock) void rle_write_kernel(const uint32_t* __restrict__ ids, uint64_t n, const uint32_t* __restrict__ blockoff, uint32_t* __restrict__ ugid,
                                                            unsigned long long* __restrict__ uoff) {
    const uint64_t base = (uint64_t)blockIdx.x * kEmitTile + (uint64_t)threadIdx.x * kEmitPer;
    bool           f[kEmitPer];
    uint32_t       c = 0;
#pragma unroll
    for (int k = 0; k < kEmitPer; ++k) {
        const uint64_t j = base + k;
        f[k]             = (j < n) && (j == 0 || ids[j] != ids[j - 1]);
        c += f[k];
    }
    uint32_t total;
    uint32_t o = blockoff[blockIdx.x] + block_exclusive_scan(c, &total);
#pragma unroll
    for (int k = 0; k < kEmitPer; ++k) {
        if (f[k]) {
            ugid[o] = ids[base + k];
            uoff[o] = base + k;
            ++o;
        }
    }
}
__global__ __launch_bounds__(kBlock) void fill_u32_kernel(uint32_t* __restrict__ p, uint32_t v, uint64_t n) {
    for (uint64_t i = (uint64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (uint64_t)gridDim.x * kBlock) p[i] = v;
}

// =================================================================================================
// parity hooks
// =================================================================================================
__global__ __launch_bounds__(kBlock) void hash_windows_kernel(const uint8_t* __restrict__ bytes, const uint32_t* __restrict__ tokstart, uint32_t npos, int n,
                                                               uint64_t* __restrict__ out) {
    const uint32_t i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= npos) return;
    uint64_t h  = 0;
    bool     ok = (uint64_t)i + n <= npos;
    for (int k = 0; ok && k < n; ++k) ok = !is_delimiter(bytes, tokstart, i + k);
    if (ok) h = spooky64_short(bytes + tokstart[i], tokstart[i + n] - tokstart[i]);
    out[i] = h;
}
__global__ __launch_bounds__(kBlock) void hash_keys_kernel(const uint8_t* __restrict__ bytes, const unsigned long long* __restrict__ off, uint64_t nkeys,
                                                            uint64_t* __restrict__ out) {
    const uint64_t j = (uint64_t)blockIdx.x * kBlock + threadIdx.x;
    if (j < nkeys) {
        const uint32_t len = (uint32_t)(off[j + 1] - off[j]);
        out[j]             = len == 0 ? 0 : spooky64_short(bytes + off[j], len);  // Pattern::hash: empty -> 0 (pattern.cpp:235-236)
    }
}

}  // namespace colibri
