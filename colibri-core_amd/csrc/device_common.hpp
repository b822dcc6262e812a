// device_common.hpp — shared device-side types and wave64 helpers (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "colibri_hip.h"

namespace colibri {

constexpr uint32_t kInvalid  = 0xFFFFFFFFu;            // "no survivor id / no slot" per position
constexpr uint64_t kEmptyKey = 0xFFFFFFFFFFFFFFFFull;  // never a real key: (kInvalid,kInvalid) is not admissible
constexpr uint32_t kKeptFlag = 0x80000000u;            // after pruning, Slot::count = kKeptFlag | result index
constexpr int      kWave     = 64;

// One hash-table slot: 16 bytes, so one probe = one 16-byte access inside a single 64-byte sector.
//   key   : the exact 64-bit pattern identity (order 1: the token's bytes, little-endian packed;
//           order n>=2: (survivor id of the left (n-1)-gram << 32) | survivor id of the right one)
//   count : occurrences (device-scope atomicAdd); rewritten by the prune kernel (see kKeptFlag)
//   rep   : a token position where the pattern occurs (first CAS winner) — locates the key bytes at export
struct __attribute__((aligned(16))) Slot {
    uint64_t key;
    uint32_t count;
    uint32_t rep;
};

// Training state that lives in HBM so that the order loop needs no host round trip.
struct DevState {
    uint32_t cap;        // table capacity (slots) of the current order
    uint32_t done;       // 1 once an order produced no candidate (reference: "None found", break)
    uint32_t found;      // distinct candidates of the current order (CAS wins)
    uint32_t kept;       // survivors of the current order
    uint32_t admitted;   // windows counted at the current order (P_n)
    uint32_t valid;      // positions holding a survivor id after resolve (upper bound of P_{n+1})
    uint32_t res_total;  // survivors of all finished orders
    uint32_t overflow;   // set when a result buffer or the table was exhausted
    uint32_t maxn;       // last order that found anything
    uint32_t radix_overflow;  // radix path: an A region or a final bin overflowed -> the host re-runs on the global table (sticky)
    uint32_t id_base;         // radix path: survivor ids of the current order start here (ids only need to be unique)
    uint32_t pad[5];
    uint32_t s_found[COLIBRI_MAX_ORDER];
    uint32_t s_kept[COLIBRI_MAX_ORDER];
    uint32_t s_admitted[COLIBRI_MAX_ORDER];
    uint32_t res_off[COLIBRI_MAX_ORDER + 1];  // res_off[n] = first result index of order n
    uint32_t s_valid[COLIBRI_MAX_ORDER];      // positions that carried a survivor id after order n (what order n + 1 can admit at most)
};

__device__ __forceinline__ uint32_t lane_id() { return __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)); }

// wave64-aggregated counter bump: one atomic per wave. Returns this lane's reserved index (valid if pred).
__device__ __forceinline__ uint32_t wave_reserve(uint32_t* counter, bool pred) {
    const uint64_t m = __ballot(pred);
    if (m == 0) return 0;
    const uint32_t lane   = lane_id();
    const uint32_t leader = (uint32_t)__builtin_ctzll(m);
    uint32_t       base   = 0;
    if (lane == leader) base = atomicAdd(counter, (uint32_t)__popcll(m));
    base = __shfl(base, (int)leader, kWave);
    return base + (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
}

// wave64-aggregated add of a per-lane value (no return)
__device__ __forceinline__ void wave_add(uint32_t* counter, uint32_t v) {
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, kWave);
    if (lane_id() == 0 && v) atomicAdd(counter, v);
}
__device__ __forceinline__ void wave_add64(unsigned long long* counter, unsigned long long v) {
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, kWave);
    if (lane_id() == 0 && v) atomicAdd(counter, v);
}

// ---- phase clocks (experimental builds only: -DCOLIBRI_KPROF) ---------------------------------------------------------------------------------------------------
// Where a block-structured kernel spends its time: thread 0 of every block reads the shader clock at the marks (placed behind barriers, so that its view is the block's),
// sums the differences per section and adds them to kprof[kernel][section] when the block ends; colibri_train prints and clears the table. Product builds: no code.
#ifdef COLIBRI_KPROF
__device__ unsigned long long kprof[8][12];
struct KProf {
    unsigned long long t, acc[12];
    int                id;
    __device__ __forceinline__ explicit KProf(int id_) : id(id_) {
#pragma unroll
        for (int s = 0; s < 12; ++s) acc[s] = 0;
        t = clock64();
    }
    __device__ __forceinline__ void mark(int s) {
        if (threadIdx.x == 0) {
            const unsigned long long n = clock64();
            acc[s] += n - t;
            t = n;
        }
    }
    __device__ __forceinline__ void done() {
        if (threadIdx.x == 0) {
#pragma unroll
            for (int s = 0; s < 12; ++s)
                if (acc[s]) atomicAdd(&kprof[id][s], acc[s]);
        }
    }
};
#define KP_INIT(id) KProf kp_(id)
#define KP(s) kp_.mark(s)
#define KP_DONE() kp_.done()
#else
#define KP_INIT(id) do {} while (0)
#define KP(s) do {} while (0)
#define KP_DONE() do {} while (0)
#endif

// table index of a 64-bit hash for a capacity that is not a power of two: high 32 bits scaled into [0,cap)
__device__ __forceinline__ uint32_t slot_of_hash(uint64_t h, uint32_t cap) { return (uint32_t)(((h >> 32) * (uint64_t)cap) >> 32); }

}  // namespace colibri
