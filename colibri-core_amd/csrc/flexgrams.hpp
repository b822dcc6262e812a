// flexgrams.hpp — flexgrams abstracted from skipgrams: IndexedPatternModel::computeflexgrams_fromskipgrams
// (reference include/patternmodel.h:3724-3744) over Pattern::toflexgram (src/pattern.cpp:145-180). Every skipgram of an indexed
// model hands all its references to the flexgram it abstracts to — the key in which each run of {*} tokens is one {**} — so the
// work is a group-by over the skipgrams (key = the collapsed bytes) followed by a merge of reference lists. On the device:
//   flex_len / flex_write   collapsed key of every skipgram (other categories get length 0 and drop out)
//   flex_insert             open-addressed table on a 64-bit hash of the collapsed bytes; a group's representative is its
//                           lowest-numbered skipgram, its size the sum of its members' reference counts
//   flex_verify             every member compares its collapsed bytes with the representative's: a hash collision is detected
//                           (never seen) and the caller retries with another seed — groups are exact
//   flex_groups             dense group numbers in representative order (deterministic), sizes, key lengths
//   flex_ref_list           one (reference number) entry per skipgram reference; the entries are then radix-sorted by
//                           (group, sentence, token) — LSD: token, sentence, group — which IS the merged, ascending lists
// The reference appends in unordered_map iteration order while inserting into the map it iterates (a rehash makes it skip or
// revisit skipgrams); the specification here is the clean one: all skipgrams, references ascending, duplicates kept
// (IndexedData::insert is a push_back, include/datatypes.h:117-119). gfx950 only.
#pragma once
#include "kernels.hpp"
#include "textenc.hpp"  // text_hash

namespace colibri {

struct FSlot {
    uint64_t hash;  // kEmptyKey = free
    uint32_t rep;   // lowest skipgram number of the group
    uint32_t cnt;   // references of the group
};
struct FlexInfo {
    uint32_t collision;
    uint32_t maxsentence;
    uint32_t pad[2];
};
constexpr uint8_t kSkipByte = 3, kFlexByte = 4;  // ClassDecoder::skipclass / flexclass (reference include/classdecoder.h:50-51)

// collapsed length of pattern p if it is a SKIPGRAM (Pattern::category, src/pattern.cpp:107-127: any {**} makes it a flexgram), else 0
__global__ __launch_bounds__(kBlock) void flex_len_kernel(const uint8_t* __restrict__ kbytes, const unsigned long long* __restrict__ koff, uint32_t np, uint32_t* __restrict__ flen) {
    for (uint32_t p = blockIdx.x * kBlock + threadIdx.x; p < np; p += gridDim.x * kBlock) {
        const uint8_t* k   = kbytes + koff[p];
        const uint32_t len = (uint32_t)(koff[p + 1] - koff[p]);
        uint32_t       out = 0;
        bool           prevhigh = false, gap = false, skip = false, flex = false;
        for (uint32_t i = 0; i < len; ++i) {
            const uint8_t b = k[i];
            if (!prevhigh && b == kSkipByte) {
                out += gap ? 0u : 1u;
                gap  = true;
                skip = true;
            } else {
                flex |= !prevhigh && b == kFlexByte;
                gap = false;
                ++out;
            }
            prevhigh = b >= 128;
        }
        flen[p] = (skip && !flex) ? out : 0u;
    }
}
__global__ __launch_bounds__(kBlock) void flex_write_kernel(const uint8_t* __restrict__ kbytes, const unsigned long long* __restrict__ koff, uint32_t np, const uint32_t* __restrict__ flen,
                                                            const unsigned long long* __restrict__ foff, uint8_t* __restrict__ fbytes) {
    for (uint32_t p = blockIdx.x * kBlock + threadIdx.x; p < np; p += gridDim.x * kBlock) {
        if (!flen[p]) continue;
        const uint8_t* k   = kbytes + koff[p];
        const uint32_t len = (uint32_t)(koff[p + 1] - koff[p]);
        uint8_t*       o   = fbytes + foff[p];
        bool           prevhigh = false, gap = false;
        for (uint32_t i = 0; i < len; ++i) {
            const uint8_t b = k[i];
            if (!prevhigh && b == kSkipByte) {
                if (!gap) *o++ = kFlexByte;
                gap = true;
            } else {
                *o++ = b;
                gap  = false;
            }
            prevhigh = b >= 128;
        }
    }
}
__global__ __launch_bounds__(kBlock) void flex_clear_kernel(FSlot* __restrict__ table, uint32_t cap) {
    for (uint32_t s = blockIdx.x * kBlock + threadIdx.x; s < cap; s += gridDim.x * kBlock) table[s] = FSlot{kEmptyKey, 0xFFFFFFFFu, 0u};
}
__global__ __launch_bounds__(kBlock) void flex_insert_kernel(const uint8_t* __restrict__ fbytes, const unsigned long long* __restrict__ foff, const uint32_t* __restrict__ flen,
                                                             const unsigned long long* __restrict__ ref_off, uint32_t np, uint64_t seed, FSlot* __restrict__ table, uint32_t cap,
                                                             uint32_t* __restrict__ slot_of) {
    for (uint32_t p = blockIdx.x * kBlock + threadIdx.x; p < np; p += gridDim.x * kBlock) {
        if (!flen[p]) continue;
        const uint64_t h = text_hash(fbytes + foff[p], flen[p], seed);
        uint32_t       s = slot_of_hash(mix64(h), cap);
        for (;;) {
            const uint64_t old = atomicCAS(reinterpret_cast<unsigned long long*>(&table[s].hash), (unsigned long long)kEmptyKey, (unsigned long long)h);
            if (old == kEmptyKey || old == h) break;
            s = (s + 1 == cap) ? 0 : s + 1;
        }
        atomicMin(&table[s].rep, p);
        atomicAdd(&table[s].cnt, (uint32_t)(ref_off[p + 1] - ref_off[p]));
        slot_of[p] = s;
    }
}
// byte check against the group's representative; isrep[p] = 1 for the representatives (0 for every other pattern)
__global__ __launch_bounds__(kBlock) void flex_verify_kernel(const uint8_t* __restrict__ fbytes, const unsigned long long* __restrict__ foff, const uint32_t* __restrict__ flen, uint32_t np,
                                                             const FSlot* __restrict__ table, const uint32_t* __restrict__ slot_of, uint32_t* __restrict__ isrep,
                                                             FlexInfo* __restrict__ info) {
    for (uint32_t p = blockIdx.x * kBlock + threadIdx.x; p < np; p += gridDim.x * kBlock) {
        const uint32_t len = flen[p];
        uint32_t       rep = 0;
        if (len) {
            const uint32_t r = table[slot_of[p]].rep;
            rep              = r == p;
            if (!rep) {
                bool same = flen[r] == len;
                const uint8_t *a = fbytes + foff[p], *b = fbytes + foff[r];
                for (uint32_t k = 0; same && k < len; ++k) same = a[k] == b[k];
                if (!same) info->collision = 1;
            }
        }
        isrep[p] = rep;
    }
}
// group g = rank of its representative among the representatives: sizes, key lengths, representative
__global__ __launch_bounds__(kBlock) void flex_groups_kernel(const uint32_t* __restrict__ isrep, const unsigned long long* __restrict__ rank, const uint32_t* __restrict__ flen,
                                                             const FSlot* __restrict__ table, const uint32_t* __restrict__ slot_of, uint32_t np, uint32_t* __restrict__ gcnt,
                                                             uint32_t* __restrict__ glen, uint32_t* __restrict__ grep) {
    for (uint32_t p = blockIdx.x * kBlock + threadIdx.x; p < np; p += gridDim.x * kBlock) {
        if (!isrep[p]) continue;
        const uint32_t g = (uint32_t)rank[p];
        gcnt[g]          = table[slot_of[p]].cnt;
        glen[g]          = flen[p];
        grep[g]          = p;
    }
}
__global__ __launch_bounds__(kBlock) void flex_keys_kernel(const uint8_t* __restrict__ fbytes, const unsigned long long* __restrict__ foff, const uint32_t* __restrict__ grep,
                                                           const uint32_t* __restrict__ glen, const unsigned long long* __restrict__ gkoff, uint32_t ngroups, uint8_t* __restrict__ out) {
    for (uint32_t g = blockIdx.x * kBlock + threadIdx.x; g < ngroups; g += gridDim.x * kBlock) {
        const uint8_t* a = fbytes + foff[grep[g]];
        uint8_t*       o = out + gkoff[g];
        for (uint32_t k = 0; k < glen[g]; ++k) o[k] = a[k];
    }
}
// references contributed per pattern (its own count if it is a skipgram)
__global__ __launch_bounds__(kBlock) void flex_contrib_kernel(const uint32_t* __restrict__ flen, const unsigned long long* __restrict__ ref_off, uint32_t np, uint32_t* __restrict__ contrib) {
    for (uint32_t p = blockIdx.x * kBlock + threadIdx.x; p < np; p += gridDim.x * kBlock) contrib[p] = flen[p] ? (uint32_t)(ref_off[p + 1] - ref_off[p]) : 0u;
}
// per input reference i of a skipgram p: list entry soff[p] + (i - ref_off[p]) = i, sort key = its token; gref[i] = the group of p
__global__ __launch_bounds__(kBlock) void flex_ref_list_kernel(const unsigned long long* __restrict__ ref_off, uint32_t np, uint64_t nrefs, const uint32_t* __restrict__ flen,
                                                               const unsigned long long* __restrict__ soff, const FSlot* __restrict__ table, const uint32_t* __restrict__ slot_of,
                                                               const unsigned long long* __restrict__ rank, const uint32_t* __restrict__ ref_sentence,
                                                               const uint16_t* __restrict__ ref_token, uint32_t* __restrict__ gref, uint32_t* __restrict__ key, uint32_t* __restrict__ val,
                                                               FlexInfo* __restrict__ info) {
    uint32_t maxs = 0;
    for (uint64_t i = (uint64_t)blockIdx.x * kBlock + threadIdx.x; i < nrefs; i += (uint64_t)gridDim.x * kBlock) {
        uint32_t lo = 0, hi = np;  // last pattern whose ref_off <= i (empty patterns share an offset: take the one that owns i)
        while (hi - lo > 1) {
            const uint32_t mid = lo + ((hi - lo) >> 1);
            if (ref_off[mid] <= i)
                lo = mid;
            else
                hi = mid;
        }
        const uint32_t p = lo;
        if (!flen[p]) continue;
        const uint64_t j = soff[p] + (i - ref_off[p]);
        gref[i]          = (uint32_t)rank[table[slot_of[p]].rep];
        key[j]           = ref_token[i];
        val[j]           = (uint32_t)i;
        maxs             = max(maxs, ref_sentence[i]);
    }
    for (int off = 32; off > 0; off >>= 1) maxs = max(maxs, (uint32_t)__shfl_down(maxs, off, kWave));
    if ((threadIdx.x & (kWave - 1)) == 0 && maxs) atomicMax(&info->maxsentence, maxs);
}
__global__ __launch_bounds__(kBlock) void flex_gather_kernel(const uint32_t* __restrict__ field, const uint32_t* __restrict__ val, uint64_t n, uint32_t* __restrict__ key) {
    for (uint64_t j = (uint64_t)blockIdx.x * kBlock + threadIdx.x; j < n; j += (uint64_t)gridDim.x * kBlock) key[j] = field[val[j]];
}
__global__ __launch_bounds__(kBlock) void flex_refs_out_kernel(const uint32_t* __restrict__ val, uint64_t n, const uint32_t* __restrict__ ref_sentence, const uint16_t* __restrict__ ref_token,
                                                               uint32_t* __restrict__ out_sentence, uint16_t* __restrict__ out_token) {
    for (uint64_t j = (uint64_t)blockIdx.x * kBlock + threadIdx.x; j < n; j += (uint64_t)gridDim.x * kBlock) {
        const uint32_t i = val[j];
        out_sentence[j]  = ref_sentence[i];
        out_token[j]     = ref_token[i];
    }
}

}  // namespace colibri
