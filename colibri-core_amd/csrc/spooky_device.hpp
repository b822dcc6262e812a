// spooky_device.hpp — SpookyHash V2 (short-message form) for gfx950 lanes.
//
// Replaces, on the device, Pattern::hash / PatternPointer::hash (reference src/pattern.cpp:234-266) ->
// SpookyHash::Hash64 (include/SpookyV2.h:59-66) -> Hash128 (src/SpookyV2.cpp:116-120) -> Short
// (src/SpookyV2.cpp:21-113; ShortMix include/SpookyV2.h:277-314, ShortEnd :328-362, sc_const :392).
// Every pattern key on this path is < 192 bytes, which is the only case Hash128 routes to Short.
// Seeds are 0/0 (Hash64(ptr,len) default seed). Bit-exact with the reference: checked on the GPU through
// colibri_hash_keys / colibri_hash_windows against the reference's values (tests/test_gpu_parity.py).
//
// One lane hashes one key. The key bytes are fetched with unaligned 8-byte loads straight from the corpus
// bytes in HBM/L2 (adjacent lanes hash adjacent, overlapping windows, so a wave touches a few hundred
// contiguous bytes); the corpus buffer is padded so that the 16-byte tail fetch never leaves the allocation.
#pragma once
#include <stdint.h>
#ifdef __HIPCC__
#include <hip/hip_runtime.h>
#define COLIBRI_HD __host__ __device__ __forceinline__
#else  // the C++ face (g++) uses the very same routine for Pattern::hash on the host
#define COLIBRI_HD inline
#endif

namespace colibri {

COLIBRI_HD uint64_t rotl64(uint64_t x, int k) { return (x << k) | (x >> (64 - k)); }

// unaligned little-endian 8-byte fetch (gfx950 global loads accept unaligned addresses)
COLIBRI_HD uint64_t ld64u(const uint8_t* p) {
    uint64_t v;
    __builtin_memcpy(&v, p, 8);
    return v;
}
// keep the low `nbytes` (0..8) bytes of v
COLIBRI_HD uint64_t keep_bytes(uint64_t v, uint32_t nbytes) {
    return nbytes >= 8 ? v : (v & ((1ull << (8 * nbytes)) - 1ull));
}

struct Spooky4 {
    uint64_t a, b, c, d;
};

// the 12-step mix applied after every 16-byte group that is followed by more data
COLIBRI_HD void spooky_mix(Spooky4& s) {
#define COLIBRI_MIX(r, ad, x, k) \
    s.r = rotl64(s.r, k);        \
    s.r += s.ad;                 \
    s.x ^= s.r;
    COLIBRI_MIX(c, d, a, 50) COLIBRI_MIX(d, a, b, 52) COLIBRI_MIX(a, b, c, 30) COLIBRI_MIX(b, c, d, 41)
    COLIBRI_MIX(c, d, a, 54) COLIBRI_MIX(d, a, b, 48) COLIBRI_MIX(a, b, c, 38) COLIBRI_MIX(b, c, d, 37)
    COLIBRI_MIX(c, d, a, 62) COLIBRI_MIX(d, a, b, 34) COLIBRI_MIX(a, b, c, 5) COLIBRI_MIX(b, c, d, 36)
#undef COLIBRI_MIX
}

// the 11-step finaliser
COLIBRI_HD void spooky_end(Spooky4& s) {
#define COLIBRI_END(t, u, k) \
    s.t ^= s.u;              \
    s.u = rotl64(s.u, k);    \
    s.t += s.u;
    COLIBRI_END(d, c, 15) COLIBRI_END(a, d, 52) COLIBRI_END(b, a, 26) COLIBRI_END(c, b, 51)
    COLIBRI_END(d, c, 28) COLIBRI_END(a, d, 9) COLIBRI_END(b, a, 47) COLIBRI_END(c, b, 54)
    COLIBRI_END(d, c, 32) COLIBRI_END(a, d, 25) COLIBRI_END(b, a, 63)
#undef COLIBRI_END
}

constexpr uint64_t kSpookyConst = 0xdeadbeefdeadbeefULL;

// Hash64 of p[0..len), len < 192. Reads at most 15 bytes beyond p+len (never used in the result).
COLIBRI_HD uint64_t spooky64_short(const uint8_t* p, uint32_t len) {
    Spooky4  s{0ull, 0ull, kSpookyConst, kSpookyConst};
    uint32_t left = len;
    if (len > 15) {
        while (left >= 32) {
            s.c += ld64u(p);
            s.d += ld64u(p + 8);
            spooky_mix(s);
            s.a += ld64u(p + 16);
            s.b += ld64u(p + 24);
            p += 32;
            left -= 32;
        }
        if (left >= 16) {
            s.c += ld64u(p);
            s.d += ld64u(p + 8);
            spooky_mix(s);
            p += 16;
            left -= 16;
        }
    }
    s.d += (uint64_t)len << 56;
    if (left == 0) {
        s.c += kSpookyConst;
        s.d += kSpookyConst;
    } else {
        // the reference's 15-way switch is a little-endian pack of tail bytes 0..7 into c and 8..14 into d
        s.c += keep_bytes(ld64u(p), left);
        if (left > 8) s.d += keep_bytes(ld64u(p + 8), left - 8);
    }
    spooky_end(s);
    return s.a;
}

}  // namespace colibri
