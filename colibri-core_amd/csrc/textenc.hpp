// textenc.hpp — the class encoder on the device: plain text -> distinct words with counts (the frequency list of
// ClassEncoder::processcorpus, reference src/classencoder.cpp:156-188) and, once the host has given every distinct word its class,
// the class-encoded corpus (ClassEncoder::encodefile / encodestring, :369-436, :550-600). gfx950 only.
//
// A *segment* is a maximal run of bytes other than ' ' and '\n' (the reference splits each getline() line at ' '). A segment that is
// the single byte \r or \t is no word (with one exception under the frequency-list rules, see text_word_at). The word is the segment RIGHT-trimmed (reference trim() = find_last_not_of + erase,
// src/common.cpp:10-19) of \t \r — and of \b too under the encoder's rules; the frequency-list rules keep an empty result as the
// word "" (e.g. the segment "\t\r"), the encoder's rules drop it. A word's identity on the device is a 64-bit hash of its bytes;
// text_verify_kernel compares every occurrence byte-for-byte with its table slot's representative, so that a hash collision is
// detected (the host then re-runs with another seed) instead of merging two words.
//
// Counting reuses count_kernel<KeyFn> (kernels.hpp §2: block-local election for the Zipf head, open-addressed table in HBM), with
// byte offsets as the "positions". The order in which words first occur — what decides ties between equally frequent words in
// the reference, through the iteration order of its unordered_map — is returned per distinct word (first_start).
#pragma once
#include "kernels.hpp"

namespace colibri {

__device__ __forceinline__ bool text_is_sep(uint8_t b) { return b == (uint8_t)' ' || b == (uint8_t)'\n'; }

// segment starting at i -> trimmed word [i, e); false if i starts no word under `rules` (0 = frequency list, 1 = encoder)
__device__ __forceinline__ bool text_word_at(const uint8_t* __restrict__ text, uint32_t n, uint32_t i, int rules, uint32_t& e) {
    if (text_is_sep(text[i]) || (i != 0 && !text_is_sep(text[i - 1]))) return false;
    uint32_t end = i + 1;
    while (end < n && !text_is_sep(text[end])) ++end;
    if (end - i == 1 && (text[i] == (uint8_t)'\r' || text[i] == (uint8_t)'\t')) {
        // no word — except under the frequency-list rules when the segment is followed by the LAST character of its line and that is
        // a space: the reference then cuts the word as "<segment><space>" (processcorpus :163-167, offset = 1), which passes its
        // one-byte filter and trims to the empty word
        const bool final_space = rules == 0 && end < n && text[end] == (uint8_t)' ' && (end + 1 == n || text[end + 1] == (uint8_t)'\n');
        if (!final_space) return false;
    }
    while (end > i) {
        const uint8_t b = text[end - 1];
        if (b == (uint8_t)'\t' || b == (uint8_t)'\r' || (rules == 1 && b == (uint8_t)'\b'))
            --end;
        else
            break;
    }
    if (rules == 1 && end == i) return false;
    e = end;
    return true;
}
__device__ __forceinline__ uint64_t text_hash(const uint8_t* __restrict__ p, uint32_t len, uint64_t seed) {
    uint64_t h = seed ^ ((uint64_t)len * 0x9E3779B97F4A7C15ull);
    uint32_t k = 0;
    for (; k + 8 <= len; k += 8) h = mix64(h ^ ld64u(p + k));
    uint64_t tail = 0;
    for (uint32_t b = 0; k + b < len; ++b) tail |= (uint64_t)p[k + b] << (8 * b);
    h = mix64(h ^ tail ^ 0xA5A5A5A5ull);
    return h == kEmptyKey ? h ^ 1ull : h;
}
struct KeyWord {
    const uint8_t* text;
    int            rules;
    uint64_t       seed;
    __device__ __forceinline__ bool operator()(uint32_t i, uint32_t npos, uint64_t& key, uint64_t& hash) const {
        uint32_t e;
        if (!text_word_at(text, npos, i, rules, e)) return false;
        key  = text_hash(text + i, e - i, seed);
        hash = mix64(key);
        return true;
    }
};

struct TextInfo {
    uint32_t nsegments;     // segment starts (upper bound of the words)
    uint32_t nlines;        // '\n' bytes
    uint32_t after_last_nl; // first byte after the last '\n' (0 if none): the encoder drops what follows (encodefile :569-570)
    uint32_t collision;     // a word met a different word in its slot
    uint32_t ndistinct;
    uint32_t pad[3];
};
__global__ __launch_bounds__(kBlock) void text_info_kernel(const uint8_t* __restrict__ text, uint32_t n, TextInfo* __restrict__ info) {
    uint32_t seg = 0, nl = 0, last = 0;
    for (uint32_t i = blockIdx.x * kBlock + threadIdx.x; i < n; i += gridDim.x * kBlock) {
        const uint8_t b = text[i];
        if (b == (uint8_t)'\n') {
            ++nl;
            last = i + 1;
        }
        seg += !text_is_sep(b) && (i == 0 || text_is_sep(text[i - 1]));
    }
    __shared__ uint32_t redL[3][kBlock / kWave];
    for (int off = 32; off > 0; off >>= 1) {
        seg += __shfl_down(seg, off, kWave);
        nl += __shfl_down(nl, off, kWave);
        last = max(last, (uint32_t)__shfl_down(last, off, kWave));
    }
    if ((threadIdx.x & (kWave - 1)) == 0) {
        redL[0][threadIdx.x / kWave] = seg;
        redL[1][threadIdx.x / kWave] = nl;
        redL[2][threadIdx.x / kWave] = last;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t a = 0, b = 0, c = 0;
        for (int w = 0; w < kBlock / kWave; ++w) {
            a += redL[0][w];
            b += redL[1][w];
            c = max(c, redL[2][w]);
        }
        if (a) atomicAdd(&info->nsegments, a);
        if (b) atomicAdd(&info->nlines, b);
        if (c) atomicMax(&info->after_last_nl, c);
    }
}
// every occurrence against its slot's representative, byte for byte; first occurrence per slot
__global__ __launch_bounds__(kBlock) void text_verify_kernel(const uint8_t* __restrict__ text, uint32_t n, int rules, const uint32_t* __restrict__ slot_of,
                                                              const Slot* __restrict__ table, uint32_t* __restrict__ first, TextInfo* __restrict__ info) {
    for (uint32_t i = blockIdx.x * kBlock + threadIdx.x; i < n; i += gridDim.x * kBlock) {
        const uint32_t s = slot_of[i];
        if (s == kInvalid) continue;
        uint32_t e = 0, re = 0;
        text_word_at(text, n, i, rules, e);
        const uint32_t r = table[s].rep;
        bool           same = text_word_at(text, n, r, rules, re) && (re - r) == (e - i);
        for (uint32_t k = 0; same && k < e - i; ++k) same = text[i + k] == text[r + k];
        if (!same) info->collision = 1;
        if (i < first[s]) atomicMin(&first[s], i);  // the plain read spares the hot words' atomics after their first few occurrences
    }
}
// table slots -> distinct word list (any order): first occurrence, byte length, count; widx[slot] = index in that list
__global__ __launch_bounds__(kBlock) void text_words_kernel(const uint8_t* __restrict__ text, uint32_t n, int rules, const Slot* __restrict__ table, uint32_t cap,
                                                             const uint32_t* __restrict__ first, uint32_t* __restrict__ widx, uint32_t* __restrict__ wstart,
                                                             uint32_t* __restrict__ wlen, uint32_t* __restrict__ wcount, TextInfo* __restrict__ info) {
    __shared__ uint32_t baseL;
    const uint32_t      ntiles = (cap + kBlock - 1) / kBlock;
    for (uint32_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const uint32_t s    = tile * kBlock + threadIdx.x;
        const bool     used = s < cap && table[s].key != kEmptyKey;
        uint32_t       total;
        const uint32_t excl = block_exclusive_scan(used ? 1u : 0u, &total);
        if (threadIdx.x == 0) baseL = total ? atomicAdd(&info->ndistinct, total) : 0;
        __syncthreads();
        if (used) {
            const uint32_t w = baseL + excl, f = first[s];
            uint32_t       e = f;
            text_word_at(text, n, f, rules, e);
            widx[s]   = w;
            wstart[w] = f;
            wlen[w]   = e - f;
            wcount[w] = table[s].count;
        }
        __syncthreads();
    }
}
// bytes each text position contributes to the encoded stream: a word -> repeat x varint(class), '\n' -> the 00 delimiter
__global__ __launch_bounds__(kBlock) void text_outlen_kernel(const uint8_t* __restrict__ text, uint32_t n, const uint32_t* __restrict__ slot_of, const uint32_t* __restrict__ widx,
                                                              const uint32_t* __restrict__ cls, const uint32_t* __restrict__ repeat, uint32_t limit, uint32_t* __restrict__ outlen,
                                                              unsigned long long* __restrict__ ntokens) {
    unsigned long long tok = 0;
    for (uint32_t i = blockIdx.x * kBlock + threadIdx.x; i < n; i += gridDim.x * kBlock) {
        uint32_t len = 0;
        if (i < limit) {
            const uint32_t s = slot_of[i];
            if (s != kInvalid) {
                const uint32_t w = widx[s], r = repeat[w];
                len = r * varint_len(cls[w]);
                tok += r;
            } else if (text[i] == (uint8_t)'\n') {
                len = 1;
            }
        }
        outlen[i] = len;
    }
    wave_add64(ntokens, tok);
}
__global__ __launch_bounds__(kBlock) void text_write_kernel(const uint8_t* __restrict__ text, uint32_t n, const uint32_t* __restrict__ slot_of, const uint32_t* __restrict__ widx,
                                                             const uint32_t* __restrict__ cls, const uint32_t* __restrict__ repeat, const uint32_t* __restrict__ outlen,
                                                             const unsigned long long* __restrict__ outoff, uint8_t* __restrict__ out) {
    for (uint32_t i = blockIdx.x * kBlock + threadIdx.x; i < n; i += gridDim.x * kBlock) {
        if (outlen[i] == 0) continue;
        uint8_t*       dst = out + outoff[i];
        const uint32_t s   = slot_of[i];
        if (s == kInvalid) {
            dst[0] = 0;  // '\n'
            continue;
        }
        const uint32_t w = widx[s];
        uint32_t       o = 0;
        for (uint32_t r = 0; r < repeat[w]; ++r) {
            uint32_t c = cls[w];
            while (c >= 128u) {
                dst[o++] = (uint8_t)((c & 127u) | 128u);
                c >>= 7;
            }
            dst[o++] = (uint8_t)c;
        }
    }
}

}  // namespace colibri
