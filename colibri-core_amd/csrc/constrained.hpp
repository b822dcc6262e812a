// constrained.hpp — training constrained by a pattern set: the reference's train(..., constrainbymodel)
// (include/patternmodel.h:1062-1072: every n-gram of every length MINLENGTH..MAXLENGTH of every sentence, in ONE pass;
// :1088-1089: `if (!constrainbymodel->has(pattern)) continue;` — no look-back at shorter patterns; :1209-1217: prune(MINTOKENS)
// regardless of size). On the device that is a membership-filtered scan: the constraint set J lives in an open-addressed table in
// HBM (64-bit hash of the key -> pattern number, the bytes themselves verify a hit). The hash folds the key token by token, so one
// thread derives the hashes of all window lengths at a position from one walk over its tokens: constraint_probe_kernel answers
// "which pattern of J is the window of n tokens at i" for up to eight lengths per pass, with all first probes of a position in
// flight together (the look-up is a chain of dependent random gathers — latency, not bandwidth). A member window's exact identity
// is then J's pattern number, so counting, pruning, survivor ids and the forward index are the ordinary counting path (radix or
// table) with the key functor KeyMember. gfx950 only.
#pragma once
#include "kernels.hpp"

namespace colibri {

struct CSlot {
    uint64_t hash;  // kEmptyKey = free
    uint32_t idx;   // pattern number in J
    uint32_t len;   // key bytes
};
constexpr uint64_t kConstraintSeed = 0x2545F4914F6CDD1Dull;
constexpr int      kProbeLengths   = 8;  // window lengths answered per probe pass

__device__ __forceinline__ uint64_t fold_chunk(uint64_t h, const uint8_t* __restrict__ p, uint32_t nbytes /*1..8*/) {
    uint64_t v = 0;
    for (uint32_t b = 0; b < nbytes; ++b) v |= (uint64_t)p[b] << (8 * b);
    return mix64(h ^ v ^ ((uint64_t)nbytes << 59));
}
// hash of a key, folded token by token (a token longer than 8 bytes — never a corpus token — continues in 8-byte chunks)
__device__ __forceinline__ uint64_t fold_hash_key(const uint8_t* __restrict__ p, uint32_t len) {
    uint64_t h = kConstraintSeed;
    uint32_t start = 0;
    for (uint32_t i = 0; i < len; ++i) {
        const bool last = p[i] < 128 || i + 1 == len;
        if (last || i + 1 - start == 8) {
            h     = fold_chunk(h, p + start, i + 1 - start);
            start = i + 1;
        }
    }
    return h == kEmptyKey ? h ^ 1ull : h;
}

__global__ __launch_bounds__(kBlock) void constraint_clear_kernel(CSlot* __restrict__ table, uint32_t cap) {
    for (uint32_t s = blockIdx.x * kBlock + threadIdx.x; s < cap; s += gridDim.x * kBlock) table[s].hash = kEmptyKey;
}
// every pattern takes the first free slot of its probe sequence (equal hashes are NOT merged: lookups verify the bytes)
__global__ __launch_bounds__(kBlock) void constraint_insert_kernel(const uint8_t* __restrict__ jbytes, const unsigned long long* __restrict__ joff, uint32_t npatterns,
                                                                    CSlot* __restrict__ table, uint32_t cap) {
    for (uint32_t p = blockIdx.x * kBlock + threadIdx.x; p < npatterns; p += gridDim.x * kBlock) {
        const uint32_t len = (uint32_t)(joff[p + 1] - joff[p]);
        if (len == 0) continue;
        const uint64_t h = fold_hash_key(jbytes + joff[p], len);
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
// pattern number of the key (bytes, len) with hash h, kInvalid if J does not hold it; `c` = the slot at `s`, already loaded
__device__ __forceinline__ uint32_t constraint_lookup(const uint8_t* __restrict__ key, uint32_t len, uint64_t h, CSlot c, uint32_t s, const CSlot* __restrict__ table, uint32_t cap,
                                                      const uint8_t* __restrict__ jbytes, const unsigned long long* __restrict__ joff) {
    for (uint32_t probe = 0; probe < cap; ++probe) {
        if (c.hash == kEmptyKey) return kInvalid;
        if (c.hash == h && c.len == len) {
            const uint8_t* q    = jbytes + joff[c.idx];
            bool           same = true;
            uint32_t       k    = 0;
            for (; same && k + 8 <= len; k += 8) same = ld64u(key + k) == ld64u(q + k);
            for (; same && k < len; ++k) same = key[k] == q[k];
            if (same) return c.idx;
        }
        s = (s + 1 == cap) ? 0 : s + 1;
        c = table[s];
    }
    return kInvalid;
}
// Is J prefix-closed — does every pattern of two or more tokens have its pattern without the last token in J too? A model built with the
// look-back is. Then a window can only be a member if the window one token shorter is, and the probe stops at the first miss.
__global__ __launch_bounds__(kBlock) void constraint_closed_kernel(const uint8_t* __restrict__ jbytes, const unsigned long long* __restrict__ joff, uint32_t npatterns,
                                                                    const CSlot* __restrict__ table, uint32_t cap, uint32_t* __restrict__ open_flag) {
    for (uint32_t p = blockIdx.x * kBlock + threadIdx.x; p < npatterns; p += gridDim.x * kBlock) {
        const uint8_t* key = jbytes + joff[p];
        const uint32_t len = (uint32_t)(joff[p + 1] - joff[p]);
        if (len == 0) continue;
        uint32_t cut = 0;  // start of the last token
        for (uint32_t i = 0; i + 1 < len; ++i)
            if (key[i] < 128) cut = i + 1;
        if (cut == 0) continue;  // one token
        const uint64_t h = fold_hash_key(key, cut);
        const uint32_t s = slot_of_hash(mix64(h), cap);
        if (constraint_lookup(key, cut, h, table[s], s, table, cap, jbytes, joff) == kInvalid) *open_flag = 1;
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
// memb[(n - n0) * stride + i] = pattern number in J of the window of n tokens at position i (n0 <= n < n0 + nlen), kInvalid if it is none
template <bool CLOSED>
__global__ __launch_bounds__(kBlock) void constraint_probe_kernel(const uint8_t* __restrict__ bytes, const uint32_t* __restrict__ tokstart, const uint32_t* __restrict__ rem,
                                                                   const CSlot* __restrict__ table, uint32_t cap, const uint8_t* __restrict__ jbytes,
                                                                   const unsigned long long* __restrict__ joff, uint32_t npos, int n0, int nlen, uint32_t* __restrict__ memb,
                                                                   size_t stride, const uint32_t* __restrict__ alive_in /* CLOSED: the window of n0 - 1 tokens was a member */) {
    for (uint32_t i = blockIdx.x * kBlock + threadIdx.x; i < npos; i += gridDim.x * kBlock) {
        const uint32_t r = rem[i];  // tokens left in the sentence (0 at a delimiter)
        uint64_t       hs[kProbeLengths];
        uint32_t       bl[kProbeLengths], sl[kProbeLengths];
        CSlot          first[kProbeLengths];
        const uint32_t a0 = tokstart[i];
        uint64_t       h  = kConstraintSeed;
        uint32_t       a  = a0;
        for (int k = 1; k < n0; ++k) {  // the tokens before the first length of interest (only when MINLENGTH > 1 or beyond the first eight lengths)
            if ((uint32_t)k > r) break;
            const uint32_t e = tokstart[i + k];
            h                = fold_chunk(h, bytes + a, e - a);
            a                = e;
        }
        // token ends, then token values, then the hash chain: every load of a stage is independent of the others of that stage
        uint32_t te[kProbeLengths];
        uint64_t tv[kProbeLengths];
#pragma unroll
        for (int j = 0; j < kProbeLengths; ++j) te[j] = (j < nlen && (uint32_t)(n0 + j) <= r) ? tokstart[i + n0 + j] : 0u;
#pragma unroll
        for (int j = 0; j < kProbeLengths; ++j) {
            const uint32_t b = j == 0 ? a : te[j - 1];
            tv[j]            = (j < nlen && (uint32_t)(n0 + j) <= r) ? ld64u(bytes + b) : 0ull;  // a corpus token has at most 8 bytes; the buffer is padded
        }
#pragma unroll
        for (int j = 0; j < kProbeLengths; ++j) {
            if (j < nlen && (uint32_t)(n0 + j) <= r) {
                const uint32_t b = j == 0 ? a : te[j - 1], nb = te[j] - b;
                const uint64_t v = nb >= 8 ? tv[j] : (tv[j] & ((1ull << (8 * nb)) - 1ull));
                h                = mix64(h ^ v ^ ((uint64_t)nb << 59));
                hs[j]            = h == kEmptyKey ? h ^ 1ull : h;
                bl[j]            = te[j] - a0;
            }
        }
        if (CLOSED) {
            // prefix-closed J: one look-up after the other, and none after the first miss (fewer look-ups beat having them all in flight:
            // the probe is bound by the number of random 16-byte requests)
            bool alive = alive_in == nullptr || alive_in[i] != kInvalid;
#pragma unroll
            for (int j = 0; j < kProbeLengths; ++j) {
                if (j >= nlen) break;
                uint32_t found = kInvalid;
                if (alive && (uint32_t)(n0 + j) <= r) {
                    const uint32_t s = slot_of_hash(mix64(hs[j]), cap);
                    found            = constraint_lookup(bytes + a0, bl[j], hs[j], table[s], s, table, cap, jbytes, joff);
                }
                alive                        = found != kInvalid;
                memb[(size_t)j * stride + i] = found;
            }
            continue;
        }
        // all first probes of this position are issued before any is looked at
#pragma unroll
        for (int j = 0; j < kProbeLengths; ++j) {
            if (j < nlen && (uint32_t)(n0 + j) <= r) {
                sl[j]    = slot_of_hash(mix64(hs[j]), cap);
                first[j] = table[sl[j]];
            }
        }
#pragma unroll
        for (int j = 0; j < kProbeLengths; ++j) {
            if (j >= nlen) break;
            uint32_t found = kInvalid;
            if ((uint32_t)(n0 + j) <= r) found = constraint_lookup(bytes + a0, bl[j], hs[j], first[j], sl[j], table, cap, jbytes, joff);
            memb[(size_t)j * stride + i] = found;
        }
    }
}
// Skipgrams in a constrained run (reference include/patternmodel.h:1163-1171 -> computeskipgrams :1410-1411: `if (constrainbymodel != NULL &&
// !constrainbymodel->has(skipgram)) continue;`): the masked form of a member window counts iff J holds it. out[i] = pattern number in J of the window of n
// tokens at i with the tokens of `mask` replaced by the skip class (one byte, 03), for the positions whose unmasked window is a member (gate); kInvalid
// elsewhere. The masked key is materialised in a small per-lane buffer (the look-up verifies bytes): at most 13 tokens of at most 8 bytes.
constexpr int kMaskedMaxTokens = 13;
__global__ __launch_bounds__(kBlock) void constraint_probe_masked_kernel(const uint8_t* __restrict__ bytes, const uint32_t* __restrict__ tokstart, const uint32_t* __restrict__ gate,
                                                                          const CSlot* __restrict__ table, uint32_t cap, const uint8_t* __restrict__ jbytes,
                                                                          const unsigned long long* __restrict__ joff, uint32_t npos, int n, uint32_t mask, uint32_t* __restrict__ out) {
    for (uint32_t i = blockIdx.x * kBlock + threadIdx.x; i < npos; i += gridDim.x * kBlock) {
        uint32_t found = kInvalid;
        if (gate[i] != kInvalid) {
            uint8_t  key[kMaskedMaxTokens * 8 + 8];
            uint32_t len = 0;
            uint64_t h   = kConstraintSeed;
            for (int k = 0; k < n; ++k) {
                if ((mask >> k) & 1u) {
                    key[len] = 3;  // ClassDecoder::skipclass (reference include/classencoder.h:61)
                    h        = fold_chunk(h, key + len, 1);
                    len += 1;
                } else {
                    const uint32_t a = tokstart[i + k], nb = tokstart[i + k + 1] - a;
                    for (uint32_t b = 0; b < nb; ++b) key[len + b] = bytes[a + b];
                    h = fold_chunk(h, key + len, nb);
                    len += nb;
                }
            }
            if (h == kEmptyKey) h ^= 1ull;
            const uint32_t s = slot_of_hash(mix64(h), cap);
            found            = constraint_lookup(key, len, h, table[s], s, table, cap, jbytes, joff);
        }
        out[i] = found;
    }
}
// ---- train(..., filter) (reference include/patternmodel.h:1106-1133): a window of n tokens is counted iff one of its sub-n-grams (any length 1..n) is in the
// filter set, or it is an instance of one of the set's skipgrams (same length, the non-gap tokens equal: src/pattern.cpp:1760-1785). Containment follows
// the recurrence "a proper sub-n-gram lies in the first n-1 or in the last n-1 tokens": cont_n[i] = exists_n[i] && (member_n[i] || cont_{n-1}[i] ||
// cont_{n-1}[i+1]); instances are looked up per gap shape with constraint_probe_masked_kernel.
__global__ __launch_bounds__(kBlock) void filter_contains_kernel(const uint32_t* __restrict__ rem, const uint32_t* __restrict__ memb /* pattern numbers of this length, or NULL */,
                                                                  const uint8_t* __restrict__ prev /* cont of the length below, or NULL */, uint32_t npos, uint32_t n,
                                                                  uint8_t* __restrict__ cont, uint32_t* __restrict__ sel, uint32_t* __restrict__ exists) {
    for (uint32_t i = blockIdx.x * kBlock + threadIdx.x; i < npos; i += gridDim.x * kBlock) {
        const bool e = rem[i] >= n;
        bool       m = false;
        if (e) {
            m = memb != nullptr && memb[i] != kInvalid;
            if (!m && prev != nullptr) m = prev[i] != 0 || (i + 1 < npos && prev[i + 1] != 0);
        }
        cont[i]   = m ? 1 : 0;
        sel[i]    = m ? 1u : 0u;
        exists[i] = e ? 0u : kInvalid;  // gate of the masked probes: kInvalid = no window of n tokens starts here
    }
}
__global__ __launch_bounds__(kBlock) void filter_or_kernel(const uint32_t* __restrict__ memb, uint32_t npos, uint32_t* __restrict__ sel) {
    for (uint32_t i = blockIdx.x * kBlock + threadIdx.x; i < npos; i += gridDim.x * kBlock)
        if (memb[i] != kInvalid) sel[i] = 1u;
}
// key functor of one length: admissible iff the probe found the window in J; key = its pattern number
struct KeyMember {
    const uint32_t* memb;
    __device__ __forceinline__ bool operator()(uint32_t i, uint32_t /*npos*/, uint64_t& key, uint64_t& hash) const {
        const uint32_t m = memb[i];
        if (m == kInvalid) return false;
        key  = m;
        hash = mix64(key ^ 0xC0FFEE123456789ull);
        return true;
    }
};

}  // namespace colibri
