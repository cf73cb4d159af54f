// Shared pieces of the two persistent decoder kernels (persist.hip: forward loop, persist_bwd.hip: BPTT): the hand-off primitives.
// THE DATA IS THE FLAG: every ring slot is pre-filled with the bit pattern 0xFFFFFFFF (a NaN no arithmetic here produces); a consumer
// polls its piece with L1-bypassing (sc1) loads until no word reads as that pattern; producers store write-through (sc1) and re-arm a
// slot two steps ahead of its next use.
#pragma once
#include "common.h"

namespace mstts {

typedef float pf32x4 __attribute__((ext_vector_type(4)));
typedef int pi32x4 __attribute__((ext_vector_type(4)));

constexpr int PH = 1024, PM = 768, PA = 128, PT = 128, PROWS = 32, PWG = 256, PTH = 512;
constexpr int PKS = 31;                          // location filter taps (hp.Attention.Conv.Kernel_Size)
constexpr int PRING = 4;
constexpr unsigned PSENT = 0xFFFFFFFFu;
constexpr unsigned long long PERSIST_TIMEOUT_TICKS = 20000000ull;   // 0.2 s of the 100 MHz wall clock per wait

__device__ __forceinline__ bool has_sent(const pf32x4& v) {
    return (__float_as_uint(v[0]) == PSENT) | (__float_as_uint(v[1]) == PSENT) | (__float_as_uint(v[2]) == PSENT) | (__float_as_uint(v[3]) == PSENT);
}
__device__ __forceinline__ pf32x4 xload(__amdgpu_buffer_rsrc_t r, unsigned byte_off) {
    return __builtin_bit_cast(pf32x4, __builtin_amdgcn_raw_buffer_load_b128(r, (int)byte_off, 0, 16));       // aux 16 = sc1
}
__device__ __forceinline__ void xstore(__amdgpu_buffer_rsrc_t r, unsigned byte_off, pf32x4 v) {
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(pi32x4, v), r, (int)byte_off, 0, 16);
}
__device__ __forceinline__ pf32x4 sentv() { const float s = __uint_as_float(PSENT); return (pf32x4){s, s, s, s}; }

// Polls N 16-byte pieces per lane until none holds the sentinel.  Returns false on time-out / abort (wave-uniform).
// (The empty asm with a memory clobber is what makes this a poll: the buffer-load builtin is a plain read to the compiler, which
//  otherwise proves the re-load redundant and deletes the whole loop.)
template <int N>
__device__ __forceinline__ bool gather(__amdgpu_buffer_rsrc_t r, const unsigned (&off)[N], pf32x4 (&v)[N], const unsigned* ctrl) {
    asm volatile("" ::: "memory");
#pragma unroll
    for (int m = 0; m < N; ++m) v[m] = xload(r, off[m]);
    unsigned spins = 0;
    unsigned long long t0 = 0;
    for (;;) {
        asm volatile("" ::: "memory");
        bool miss = false;
#pragma unroll
        for (int m = 0; m < N; ++m) miss |= has_sent(v[m]);
        if (!__builtin_amdgcn_ballot_w64(miss)) return true;
        if ((++spins & 15u) == 0) {
            const unsigned long long now = wall_clock64();
            if (t0 == 0) t0 = now;
            if (now - t0 > PERSIST_TIMEOUT_TICKS || __hip_atomic_load(ctrl + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) return false;
        }
#pragma unroll
        for (int m = 0; m < N; ++m)
            if (has_sent(v[m])) v[m] = xload(r, off[m]);
    }
}

// The same in two parts, so that the round trip of the first request runs under other work: issue<N>() as early as the addresses are
// known, complete<N>() where the data is needed (it polls on as gather<N>() does).
template <int N>
__device__ __forceinline__ void issue(__amdgpu_buffer_rsrc_t r, const unsigned (&off)[N], pf32x4 (&v)[N]) {
    asm volatile("" ::: "memory");
#pragma unroll
    for (int m = 0; m < N; ++m) v[m] = xload(r, off[m]);
    asm volatile("" ::: "memory");                   // ... and they stay HERE: without the second fence the scheduler sinks the requests
    __builtin_amdgcn_sched_barrier(0);               // below the work they are meant to run under
}
template <int N>
__device__ __forceinline__ bool complete(__amdgpu_buffer_rsrc_t r, const unsigned (&off)[N], pf32x4 (&v)[N], const unsigned* ctrl) {
    unsigned spins = 0;
    unsigned long long t0 = 0;
    for (;;) {
        asm volatile("" ::: "memory");
        bool miss = false;
#pragma unroll
        for (int m = 0; m < N; ++m) miss |= has_sent(v[m]);
        if (!__builtin_amdgcn_ballot_w64(miss)) return true;
        if ((++spins & 15u) == 0) {
            const unsigned long long now = wall_clock64();
            if (t0 == 0) t0 = now;
            if (now - t0 > PERSIST_TIMEOUT_TICKS || __hip_atomic_load(ctrl + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) return false;
        }
#pragma unroll
        for (int m = 0; m < N; ++m)
            if (has_sent(v[m])) v[m] = xload(r, off[m]);
    }
}

#define PMFMA(a, b, c) __builtin_amdgcn_mfma_f32_16x16x4f32((a), (b), (c), 0, 0, 0)

}  // namespace mstts
