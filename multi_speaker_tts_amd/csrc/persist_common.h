// Shared pieces of the two persistent decoder kernels (persist.hip: forward loop, persist_bwd.hip: BPTT): the hand-off primitives.
// THE DATA IS THE FLAG: every word that crosses a ring carries the GENERATION of its ring slot in the lowest bit of its mantissa (a ring
// of 4 slots: step k uses slot k & 3, generation (k >> 2) & 1; the rings are pre-filled with all-ones words = "generation 1" before the
// launch, the first pass over the ring is generation 0).  A producer stores its values write-through (sc1) with that bit forced; a
// consumer polls its piece with L1-bypassing (sc1) loads until every word shows the generation it expects - what it last saw in that
// slot, four steps ago, carries the opposite bit.  No flag, no counter, no fence, and no second store to re-arm a slot (the first
// form of these kernels re-armed with a NaN pattern: half of the write-through traffic).  The price is the value's last bit: 2^-24
// relative on exchanged copies only (what BPTT reads back is stored exactly), deterministic.
#pragma once
#include "common.h"

namespace mstts {

typedef float pf32x4 __attribute__((ext_vector_type(4)));
typedef int pi32x4 __attribute__((ext_vector_type(4)));

constexpr int PH = 1024, PM = 768, PA = 128, PT = 128, PROWS = 32, PWG = 256, PTH = 512;
constexpr int PTMAX = 256;                       // encoder positions the persistent decoder kernels admit (instantiations for 128 and 256)
constexpr int PKS = 31;                          // location filter taps (hp.Attention.Conv.Kernel_Size)
constexpr int PRING = 4;
constexpr unsigned PSENT = 0xFFFFFFFFu;
constexpr unsigned long long PERSIST_TIMEOUT_TICKS = 20000000ull;   // 0.2 s of the 100 MHz wall clock per wait inside the loop
// The start rendezvous has its own, much shorter bound: the workgroups of one launch are dispatched together, so either all of them find a
// CU within microseconds or some of them are waiting for CUs that another stream's kernel holds - and then the launch-per-step loop is
// the better use of the next 0.2 s.  2 ms covers a collective's kernel draining off the CUs it took.
constexpr unsigned long long PERSIST_RENDEZVOUS_TICKS = 200000ull;

// true while any word of the piece still shows the other generation
__device__ __forceinline__ bool stale(const pf32x4& v, unsigned gen) {
    return (((__float_as_uint(v[0]) ^ gen) | (__float_as_uint(v[1]) ^ gen) | (__float_as_uint(v[2]) ^ gen) | (__float_as_uint(v[3]) ^ gen)) & 1u) != 0u;
}
__device__ __forceinline__ pf32x4 tagv(pf32x4 v, unsigned gen) {
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = __uint_as_float((__float_as_uint(v[e]) & ~1u) | gen);
    return v;
}
__device__ __forceinline__ pf32x4 xload(__amdgpu_buffer_rsrc_t r, unsigned byte_off) {
    return __builtin_bit_cast(pf32x4, __builtin_amdgcn_raw_buffer_load_b128(r, (int)byte_off, 0, 16));       // aux 16 = sc1
}
__device__ __forceinline__ void xstore(__amdgpu_buffer_rsrc_t r, unsigned byte_off, pf32x4 v) {
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(pi32x4, v), r, (int)byte_off, 0, 16);
}
// publish: the value with its slot's generation in the last bit, write-through
__device__ __forceinline__ void xpublish(__amdgpu_buffer_rsrc_t r, unsigned byte_off, pf32x4 v, unsigned gen) { xstore(r, byte_off, tagv(v, gen)); }
// ... or, when EVERY consumer of the piece sits on the producer's XCD (`local`, established at the start rendezvous from the hardware's
// XCC ids - never assumed from block ids), with a plain store: the line stays in that XCD's L2, where the consumers' L1-bypassing polls
// find it without the trip to the memory side.  Correctness does not depend on the placement: a group whose members report different
// XCC ids publishes write-through.
__device__ __forceinline__ void xpublish_near(__amdgpu_buffer_rsrc_t r, unsigned byte_off, pf32x4 v, unsigned gen, bool local) {
    if (local) __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(pi32x4, tagv(v, gen)), r, (int)byte_off, 0, 0);
    else xstore(r, byte_off, tagv(v, gen));
}
// Before a launch that may publish through an L2 (near_xcd): every workgroup drops its XCD's copies of the rings concerned, by storing
// the rings' idle pattern (all ones, what the host's memset left in memory) write-through over 1/32 of them - sc1 stores do not leave
// the line in the issuing XCD's L2.  The 32 workgroups of a slice group that shares one XCD (the only groups that publish near) cover
// the whole region, so no line a previous launch parked in that L2 can be mistaken for fresh data, whatever the hardware does with L2
// contents between launches.  Call before the start rendezvous.
__device__ __forceinline__ void persist_scrub(__amdgpu_buffer_rsrc_t r, long float_off, long floats, int g, int tid) {
    const long per = floats / 32;                                    // (region sizes are multiples of 32 * 4 floats)
    const unsigned base = (unsigned)((float_off + (long)(g >> 3) * per) * 4);
    const pf32x4 ones = {__uint_as_float(PSENT), __uint_as_float(PSENT), __uint_as_float(PSENT), __uint_as_float(PSENT)};
    for (long x = tid; x < per / 4; x += PTH) xstore(r, base + (unsigned)(16 * x), ones);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}
// XCC id of the CU this wave runs on (HW_REG_XCC_ID, bits 3:0) + 1
__device__ __forceinline__ unsigned xcc_id_plus1() { return (__builtin_amdgcn_s_getreg((3 << 11) | 20) & 0xfu) + 1u; }
constexpr int PCTRL_WORDS = 16 + 256;            // control words: arrivals, abort code, finished count, ...; then one XCC id per workgroup
// Start rendezvous of the 256 workgroups (thread 0 of each): records the XCC id, arrives, waits for the others (bounded).  Returns 0 on
// time-out / abort, 1 when resident, 2 when in addition the 32 workgroups that share this workgroup's slice index (id & 7) all sit on its XCD.
__device__ __forceinline__ int persist_rendezvous(unsigned* ctrl, int g) {
    const unsigned me = xcc_id_plus1();
    __hip_atomic_store(ctrl + 16 + g, me, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    __hip_atomic_fetch_add(ctrl, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const unsigned long long t0 = wall_clock64();
    while (__hip_atomic_load(ctrl, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (unsigned)PWG) {
        __builtin_amdgcn_s_sleep(8);
        if (wall_clock64() - t0 > PERSIST_RENDEZVOUS_TICKS || __hip_atomic_load(ctrl + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) {
            __hip_atomic_store(ctrl + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            return 0;
        }
    }
    bool same = true;
    for (int j = 0; j < 32; ++j) same = same && (__hip_atomic_load(ctrl + 16 + (g & 7) + 8 * j, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == me);
    return same ? 2 : 1;
}

// Polls N 16-byte pieces per lane until each shows its generation gen[m].  Returns false on time-out / abort (wave-uniform).
// (The empty asm with a memory clobber is what makes this a poll: the buffer-load builtin is a plain read to the compiler, which
//  otherwise proves the re-load redundant and deletes the whole loop.)
template <int N>
__device__ __forceinline__ bool gather(__amdgpu_buffer_rsrc_t r, const unsigned (&off)[N], pf32x4 (&v)[N], const unsigned* ctrl, const unsigned (&gen)[N]) {
    asm volatile("" ::: "memory");
#pragma unroll
    for (int m = 0; m < N; ++m) v[m] = xload(r, off[m]);
    unsigned spins = 0;
    unsigned long long t0 = 0;
    for (;;) {
        asm volatile("" ::: "memory");
        bool miss = false;
#pragma unroll
        for (int m = 0; m < N; ++m) miss |= stale(v[m], gen[m]);
        if (!__builtin_amdgcn_ballot_w64(miss)) return true;
        if ((++spins & 15u) == 0) {
            const unsigned long long now = wall_clock64();
            if (t0 == 0) t0 = now;
            if (now - t0 > PERSIST_TIMEOUT_TICKS || __hip_atomic_load(ctrl + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) return false;
        }
#pragma unroll
        for (int m = 0; m < N; ++m)
            if (stale(v[m], gen[m])) v[m] = xload(r, off[m]);
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
__device__ __forceinline__ bool complete(__amdgpu_buffer_rsrc_t r, const unsigned (&off)[N], pf32x4 (&v)[N], const unsigned* ctrl, const unsigned (&gen)[N]) {
    unsigned spins = 0;
    unsigned long long t0 = 0;
    for (;;) {
        asm volatile("" ::: "memory");
        bool miss = false;
#pragma unroll
        for (int m = 0; m < N; ++m) miss |= stale(v[m], gen[m]);
        if (!__builtin_amdgcn_ballot_w64(miss)) return true;
        if ((++spins & 15u) == 0) {
            const unsigned long long now = wall_clock64();
            if (t0 == 0) t0 = now;
            if (now - t0 > PERSIST_TIMEOUT_TICKS || __hip_atomic_load(ctrl + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) return false;
        }
#pragma unroll
        for (int m = 0; m < N; ++m)
            if (stale(v[m], gen[m])) v[m] = xload(r, off[m]);
    }
}

// Packed operands of the BPTT's cell updates, written by the forward loop's cell updates: float4 index of (step, workgroup, cell, half,
// owner thread) - the owner thread tid < 128 is (row 16 (tid >> 6) + (tid & 15), unit 4 g + ((tid & 63) >> 4)); half 0 = the gate
// activations (i, j, f, o), half 1 = (raw cell state, previous zoned cell state, keep-mask bits zc | zh << 1, -).  One workgroup's block
// of a step and cell is 4 KB contiguous: the row-major histories [S, B, 4H] / [S, B, H] cost the owner of 4 units x 32 rows sixteen
// row-strided accesses per step and direction, each touching 16 pages.
__device__ __forceinline__ long opk_index(int s, int g, int cell, int half, int tid) { return ((((long)s * PWG + g) * 2 + cell) * 2 + half) * 128 + tid; }
constexpr long OPK_FLOATS_PER_STEP = (long)PWG * 2 * 2 * 128 * 4;

// Per-device memo of a "can this device take the launch" probe (CU count, occupancy, the kernels' dynamic-LDS attribute): the probe runs
// once for EVERY device a process drives - hipFuncSetAttribute applies to the current device only - not once per process.
constexpr int PERSIST_MAX_DEVICES = 64;
template <typename Probe>
static inline int persist_device_memo(int* memo, Probe probe) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= PERSIST_MAX_DEVICES) { (void)hipGetLastError(); return 0; }
    if (memo[dev] == 0) { memo[dev] = probe(dev) ? 2 : 1; (void)hipGetLastError(); }
    return memo[dev] == 2 ? 1 : 0;
}

#define PMFMA(a, b, c) __builtin_amdgcn_mfma_f32_16x16x4f32((a), (b), (c), 0, 0, 0)

}  // namespace mstts
