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
// All passes walk the *event list* built at upload time — the byte offsets of the segment starts and of the '\n' bytes, in text
// order (one entry per ~4.4 bytes of English-like text) — instead of every byte offset.
// Counting reuses count_kernel<KeyFn> (kernels.hpp §2: block-local election for the Zipf head, open-addressed table in HBM), with
// event indices as the "positions". The order in which words first occur — what decides ties between equally frequent words in
// the reference, through the iteration order of its unordered_map — is returned per distinct word (first_start).
#pragma once
#include "kernels.hpp"

namespace colibri {

__device__ __forceinline__ bool text_is_sep(uint8_t b) { return b == (uint8_t)' ' || b == (uint8_t)'\n'; }

// index of the first ' ' or '\n' byte in the 8 little-endian bytes of w (8 if none): SWAR zero-byte test on w ^ pattern
__device__ __forceinline__ uint32_t text_first_sep8(uint64_t w) {
    const uint64_t lo = 0x0101010101010101ull, hi = 0x8080808080808080ull;
    const uint64_t a = w ^ 0x2020202020202020ull, b = w ^ 0x0A0A0A0A0A0A0A0Aull;
    const uint64_t m = (((a - lo) & ~a) | ((b - lo) & ~b)) & hi;  // bit 7 of every byte at or before the first match is exact; later ones may be false positives
    return m ? (uint32_t)(__builtin_ctzll(m) >> 3) : 8u;
}
// segment starting at i (i is not a separator and follows one or the start of the text) -> trimmed word [i, e); false if the
// segment is no word under `rules` (0 = frequency list, 1 = encoder). The text buffer is readable 16 bytes past n.
__device__ __forceinline__ bool text_word_of_segment(const uint8_t* __restrict__ text, uint32_t n, uint32_t i, int rules, uint32_t& e) {
    // the segment end: 16 bytes in two unaligned loads cover almost every word; longer ones continue 8 bytes at a time
    uint32_t end = i;
    for (;;) {
        const uint32_t k = text_first_sep8(ld64u(text + end));
        end += k;
        if (k < 8 || end >= n) break;
    }
    if (end > n) end = n;
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
// the same for an arbitrary byte offset: false unless i starts a segment
__device__ __forceinline__ bool text_word_at(const uint8_t* __restrict__ text, uint32_t n, uint32_t i, int rules, uint32_t& e) {
    if (text_is_sep(text[i]) || (i != 0 && !text_is_sep(text[i - 1]))) return false;
    return text_word_of_segment(text, n, i, rules, e);
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
    const uint8_t*  text;
    uint32_t        nbytes;
    const uint32_t* events;  // byte offsets of segment starts and newlines, ascending
    int             rules;
    uint64_t        seed;
    __device__ __forceinline__ bool operator()(uint32_t j, uint32_t /*nevents*/, uint64_t& key, uint64_t& hash) const {
        const uint32_t i = events[j];
        uint32_t       e;
        if (text[i] == (uint8_t)'\n' || !text_word_of_segment(text, nbytes, i, rules, e)) return false;  // a newline event, or a segment that is no word
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
// event list = offsets of segment starts and of '\n' bytes, in order: per-block counts, (scan), ordered write
constexpr int kEvBytesPerBlock = kBlock * 16;
__device__ __forceinline__ bool text_is_event(const uint8_t* __restrict__ text, uint32_t i) {
    const uint8_t b = text[i];
    return b == (uint8_t)'\n' || (b != (uint8_t)' ' && (i == 0 || text_is_sep(text[i - 1])));
}
__global__ __launch_bounds__(kBlock) void text_event_count_kernel(const uint8_t* __restrict__ text, uint32_t n, uint32_t* __restrict__ blockcnt) {
    const uint32_t base = blockIdx.x * kEvBytesPerBlock + threadIdx.x * 16;
    uint32_t       c    = 0;
    for (uint32_t k = 0; k < 16; ++k)
        if (base + k < n) c += text_is_event(text, base + k);
    uint32_t total;
    block_exclusive_scan(c, &total);
    if (threadIdx.x == 0) blockcnt[blockIdx.x] = total;
}
__global__ __launch_bounds__(kBlock) void text_event_write_kernel(const uint8_t* __restrict__ text, uint32_t n, const unsigned long long* __restrict__ blockoff,
                                                                   uint32_t* __restrict__ events) {
    const uint32_t base = blockIdx.x * kEvBytesPerBlock + threadIdx.x * 16;
    uint32_t       c = 0, m = 0;
    for (uint32_t k = 0; k < 16; ++k)
        if (base + k < n && text_is_event(text, base + k)) {
            m |= 1u << k;
            ++c;
        }
    uint32_t total;
    uint32_t o = (uint32_t)blockoff[blockIdx.x] + block_exclusive_scan(c, &total);
    for (uint32_t k = 0; k < 16; ++k)
        if (m & (1u << k)) events[o++] = base + k;
}
// every occurrence against its slot's representative, byte for byte; first occurrence per slot
__global__ __launch_bounds__(kBlock) void text_verify_kernel(const uint8_t* __restrict__ text, uint32_t n, const uint32_t* __restrict__ events, uint32_t nevents, int rules,
                                                              const uint32_t* __restrict__ slot_of, const Slot* __restrict__ table, uint32_t* __restrict__ first,
                                                              TextInfo* __restrict__ info) {
    for (uint32_t j = blockIdx.x * kBlock + threadIdx.x; j < nevents; j += gridDim.x * kBlock) {
        const uint32_t s = slot_of[j];
        if (s == kInvalid) continue;
        const uint32_t i = events[j], r = events[table[s].rep];
        uint32_t       e = 0, re = 0;
        text_word_of_segment(text, n, i, rules, e);
        bool           same = text_word_of_segment(text, n, r, rules, re) && (re - r) == (e - i);
        const uint32_t len  = e - i;
        uint32_t       k    = 0;
        for (; same && k + 8 <= len; k += 8) same = ld64u(text + i + k) == ld64u(text + r + k);
        if (same && k < len) {
            const uint64_t keep = ~0ull >> (8 * (8 - (len - k)));
            same                = ((ld64u(text + i + k) ^ ld64u(text + r + k)) & keep) == 0;
        }
        if (!same) info->collision = 1;
        if (j < first[s]) atomicMin(&first[s], j);  // the plain read spares the hot words' atomics after their first few occurrences
    }
}
// table slots -> distinct word list (any order): first occurrence, byte length, count; widx[slot] = index in that list
__global__ __launch_bounds__(kBlock) void text_words_kernel(const uint8_t* __restrict__ text, uint32_t n, int rules, const uint32_t* __restrict__ events, const Slot* __restrict__ table,
                                                             uint32_t cap, const uint32_t* __restrict__ first, uint32_t* __restrict__ widx, uint32_t* __restrict__ wstart,
                                                             uint32_t* __restrict__ wlen, uint32_t* __restrict__ wcount, TextInfo* __restrict__ info) {
    // 4096 slots per tile and reservation (one atomic on a single counter costs ~12 ns whatever else happens)
    __shared__ uint32_t baseL;
    const uint32_t      ntiles = (cap + kPruneTile - 1) / kPruneTile;
    for (uint32_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const uint32_t s0 = tile * kPruneTile + threadIdx.x * kPrunePer;
        uint32_t       used = 0, k = 0;
#pragma unroll
        for (int q = 0; q < kPrunePer; ++q)
            if (s0 + q < cap && table[s0 + q].key != kEmptyKey) {
                used |= 1u << q;
                ++k;
            }
        uint32_t       total;
        const uint32_t excl = block_exclusive_scan(k, &total);
        if (threadIdx.x == 0) baseL = total ? atomicAdd(&info->ndistinct, total) : 0;
        __syncthreads();
        uint32_t w = baseL + excl;
#pragma unroll
        for (int q = 0; q < kPrunePer; ++q) {
            if (!(used & (1u << q))) continue;
            const uint32_t s = s0 + q, f = events[first[s]];
            uint32_t       e = f;
            text_word_of_segment(text, n, f, rules, e);
            widx[s]   = w;
            wstart[w] = f;
            wlen[w]   = e - f;
            wcount[w] = table[s].count;
            ++w;
        }
        __syncthreads();
    }
}
// bytes each event contributes to the encoded stream: a word -> repeat x varint(class), '\n' -> the 00 delimiter
__global__ __launch_bounds__(kBlock) void text_outlen_kernel(const uint8_t* __restrict__ text, const uint32_t* __restrict__ events, uint32_t nevents,
                                                              const uint32_t* __restrict__ slot_of, const uint32_t* __restrict__ widx, const uint32_t* __restrict__ cls,
                                                              const uint32_t* __restrict__ repeat, uint32_t limit, uint32_t* __restrict__ outlen,
                                                              unsigned long long* __restrict__ ntokens) {
    unsigned long long tok = 0;
    for (uint32_t j = blockIdx.x * kBlock + threadIdx.x; j < nevents; j += gridDim.x * kBlock) {
        const uint32_t i   = events[j];
        uint32_t       len = 0;
        if (i < limit) {
            const uint32_t s = slot_of[j];
            if (s != kInvalid) {
                const uint32_t w = widx[s], r = repeat[w];
                len = r * varint_len(cls[w]);
                tok += r;
            } else if (text[i] == (uint8_t)'\n') {
                len = 1;
            }
        }
        outlen[j] = len;
    }
    wave_add64(ntokens, tok);
}
__global__ __launch_bounds__(kBlock) void text_write_kernel(const uint32_t* __restrict__ slot_of, uint32_t nevents, const uint32_t* __restrict__ widx, const uint32_t* __restrict__ cls,
                                                             const uint32_t* __restrict__ repeat, const uint32_t* __restrict__ outlen,
                                                             const unsigned long long* __restrict__ outoff, uint8_t* __restrict__ out) {
    for (uint32_t j = blockIdx.x * kBlock + threadIdx.x; j < nevents; j += gridDim.x * kBlock) {
        if (outlen[j] == 0) continue;
        uint8_t*       dst = out + outoff[j];
        const uint32_t s   = slot_of[j];
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
