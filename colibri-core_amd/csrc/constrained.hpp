// constrained.hpp — training constrained by a pattern set: the reference's train(..., constrainbymodel)
// (include/patternmodel.h:1062-1072: every n-gram of every length MINLENGTH..MAXLENGTH of every sentence, in ONE pass;
// :1088-1089: `if (!constrainbymodel->has(pattern)) continue;` — no look-back at shorter patterns; :1209-1217: prune(MINTOKENS)
// regardless of size). On the device that is a membership-filtered scan: the constraint set J lives in an open-addressed table in
// HBM (64-bit hash of the key bytes -> pattern number, the bytes themselves verify a hit), a window is admissible iff its bytes are a
// member, and its exact identity is then J's pattern number — so counting, pruning, survivor ids and the forward index are the
// ordinary count_kernel / prune / resolve / emit_pairs of the table path with another key functor. gfx950 only.
#pragma once
#include "kernels.hpp"
#include "textenc.hpp"  // text_hash

namespace colibri {

struct CSlot {
    uint64_t hash;  // kEmptyKey = free
    uint32_t idx;   // pattern number in J
    uint32_t len;   // key bytes
};
constexpr uint64_t kConstraintSeed = 0x2545F4914F6CDD1Dull;

__global__ __launch_bounds__(kBlock) void constraint_clear_kernel(CSlot* __restrict__ table, uint32_t cap) {
    for (uint32_t s = blockIdx.x * kBlock + threadIdx.x; s < cap; s += gridDim.x * kBlock) table[s].hash = kEmptyKey;
}
// every pattern takes the first free slot of its probe sequence (equal hashes are NOT merged: lookups verify the bytes)
__global__ __launch_bounds__(kBlock) void constraint_insert_kernel(const uint8_t* __restrict__ jbytes, const unsigned long long* __restrict__ joff, uint32_t npatterns,
                                                                    CSlot* __restrict__ table, uint32_t cap) {
    for (uint32_t p = blockIdx.x * kBlock + threadIdx.x; p < npatterns; p += gridDim.x * kBlock) {
        const uint32_t len = (uint32_t)(joff[p + 1] - joff[p]);
        if (len == 0) continue;
        const uint64_t h = text_hash(jbytes + joff[p], len, kConstraintSeed);
        uint32_t       s = slot_of_hash(mix64(h), cap);
        for (;;) {
            const uint64_t old = atomicCAS(reinterpret_cast<unsigned long long*>(&table[s].hash), (unsigned long long)kEmptyKey, (unsigned long long)h);
            if (old == kEmptyKey) {
                table[s].idx = p;
                table[s].len = len;
                break;
            }
            s = (s + 1 == cap) ? 0 : s + 1;
        }
    }
}
// tokens left in the sentence from position i on (0 at a delimiter): a window of n tokens at i exists iff rem[i] >= n
__global__ __launch_bounds__(kBlock) void sentence_rem_kernel(const uint32_t* __restrict__ delimpos, uint32_t ndelim, uint32_t npos, uint32_t* __restrict__ rem) {
    for (uint32_t i = blockIdx.x * kBlock + threadIdx.x; i < npos; i += gridDim.x * kBlock) {
        uint32_t lo = 0, hi = ndelim;  // first delimiter position >= i
        while (lo < hi) {
            const uint32_t mid = (lo + hi) >> 1;
            if (delimpos[mid] < i)
                lo = mid + 1;
            else
                hi = mid;
        }
        rem[i] = (lo < ndelim ? delimpos[lo] : npos) - i;
    }
}
// key functor of order n: admissible iff the window's bytes are a pattern of J; key = its pattern number
struct KeyConstrained {
    const uint8_t*            bytes;
    const uint32_t*           tokstart;
    const uint32_t*           rem;
    const CSlot*              table;
    uint32_t                  cap;
    const uint8_t*            jbytes;
    const unsigned long long* joff;
    int                       n;
    __device__ __forceinline__ bool operator()(uint32_t i, uint32_t /*npos*/, uint64_t& key, uint64_t& hash) const {
        if (rem[i] < (uint32_t)n) return false;
        const uint32_t a = tokstart[i], len = tokstart[i + n] - a;
        const uint64_t h = text_hash(bytes + a, len, kConstraintSeed);
        uint32_t       s = slot_of_hash(mix64(h), cap);
        for (uint32_t probe = 0; probe < cap; ++probe) {
            const CSlot c = table[s];
            if (c.hash == kEmptyKey) return false;
            if (c.hash == h && c.len == len) {
                const uint8_t* j    = jbytes + joff[c.idx];
                bool           same = true;
                uint32_t       k    = 0;
                for (; same && k + 8 <= len; k += 8) same = ld64u(bytes + a + k) == ld64u(j + k);
                for (; same && k < len; ++k) same = bytes[a + k] == j[k];
                if (same) {
                    key  = c.idx;
                    hash = mix64(key ^ 0xC0FFEE123456789ull);
                    return true;
                }
            }
            s = (s + 1 == cap) ? 0 : s + 1;
        }
        return false;
    }
};

}  // namespace colibri
