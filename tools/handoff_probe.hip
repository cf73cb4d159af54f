// One-way latency of the persistent decoder kernels' hand-off (persist_common.h: the data is the flag, generation bit in the mantissa,
// write-through store, L1-bypassing polls) between two workgroups, as a ping-pong of 1 KB messages (one wave, 16 bytes per lane):
//   * partner on the SAME XCD or on ANOTHER one (XCC ids read from the hardware, printed);
//   * publication write-through (sc1) or - same XCD only - a plain store that stays in the XCD's L2;
//   * polling: one request at a time, re-issued when it comes back stale (what the kernels do), optionally after a back-off of gap x 64
//     clocks; or D requests in flight, staggered, so a stale answer is followed by the next one after RTT / D instead of a whole round trip.
//   hipcc --offload-arch=gfx950 -O3 tools/handoff_probe.hip -o /tmp/handoff_probe && /tmp/handoff_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f4 __attribute__((ext_vector_type(4)));
typedef int i4 __attribute__((ext_vector_type(4)));
constexpr int WG = 256, RING = 4;
__device__ __forceinline__ bool stale(const f4& v, unsigned gen) {
    return (((__float_as_uint(v[0]) ^ gen) | (__float_as_uint(v[1]) ^ gen) | (__float_as_uint(v[2]) ^ gen) | (__float_as_uint(v[3]) ^ gen)) & 1u) != 0u;
}
__device__ __forceinline__ f4 tagv(f4 v, unsigned gen) {
    for (int e = 0; e < 4; ++e) v[e] = __uint_as_float((__float_as_uint(v[e]) & ~1u) | gen);
    return v;
}
__device__ __forceinline__ f4 xload(__amdgpu_buffer_rsrc_t r, unsigned off) { return __builtin_bit_cast(f4, __builtin_amdgcn_raw_buffer_load_b128(r, (int)off, 0, 16)); }
__device__ __forceinline__ void xstore(__amdgpu_buffer_rsrc_t r, unsigned off, f4 v, bool near) {
    if (near) __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(i4, v), r, (int)off, 0, 0);
    else __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(i4, v), r, (int)off, 0, 16);
}
// polls `off` until every lane's piece shows `gen`; DEPTH requests in flight, the later ones issued `gap` x 64 clocks apart
template <int DEPTH>
__device__ __forceinline__ f4 poll(__amdgpu_buffer_rsrc_t r, unsigned off, unsigned gen, int gap, unsigned* polls) {
    f4 v[DEPTH];
    asm volatile("" ::: "memory");
    v[0] = xload(r, off);
#pragma unroll
    for (int d = 1; d < DEPTH; ++d) {
        for (int q = 0; q < gap; ++q) __builtin_amdgcn_s_sleep(1);
        asm volatile("" ::: "memory");
        v[d] = xload(r, off);
    }
    unsigned n = DEPTH;
    for (;;) {
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) {
            asm volatile("" ::: "memory");
            if (!__builtin_amdgcn_ballot_w64(stale(v[d], gen))) { *polls += n; return v[d]; }
            if (DEPTH == 1) for (int q = 0; q < gap; ++q) __builtin_amdgcn_s_sleep(1);        // depth 1: `gap` is the back-off between polls
            v[d] = xload(r, off);
            ++n;
            if (n > 4000000u) { *polls += n; return v[d]; }
        }
    }
}
// workgroup 0 pings workgroup `partner`, `rounds` times; ticks[0] = wall-clock ticks (100 MHz) of the whole exchange, ticks[1] = polls issued by workgroup 0
template <int DEPTH>
__global__ __launch_bounds__(64) void pingpong(float* ring, unsigned* xcc, unsigned long long* ticks, int partner, int rounds, int near, int gap) {
    extern __shared__ float pad[];                   // 160 KB requested: one workgroup per CU
    const int g = blockIdx.x, lane = threadIdx.x;
    if (lane == 0) xcc[g] = __builtin_amdgcn_s_getreg((3 << 11) | 20) & 0xfu;
    if (g != 0 && g != partner) return;
    const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(ring, 0, 2 * RING * 256 * 4, 0x00020000);
    // ring[0]: 0 -> partner, ring[1]: partner -> 0 ; slot k & 3, generation (k >> 2) & 1 ; memset 0xFF = generation 1
    unsigned polls = 0;
    f4 val = {1.0f + lane, 2.0f, 3.0f, 4.0f};
    const unsigned long long t0 = wall_clock64();
    for (int k = 0; k < rounds; ++k) {
        const unsigned slot = k & 3, gen = (k >> 2) & 1;
        const unsigned to = (slot * 256 + lane * 4) * 4, back = ((RING + slot) * 256 + lane * 4) * 4;
        if (g == 0) {
            xstore(r, to, tagv(val, gen), near);
            val = poll<DEPTH>(r, back, gen, gap, &polls);
        } else {
            val = poll<DEPTH>(r, to, gen, gap, &polls);
            val[1] += 1.0f;
            xstore(r, back, tagv(val, gen), near);
        }
    }
    if (g == 0 && lane == 0) { ticks[0] = wall_clock64() - t0; ticks[1] = polls; ticks[2] = (unsigned long long)val[1]; }
}
int main() {
    float* ring; unsigned* xcc; unsigned long long* ticks;
    hipMalloc(&ring, 2 * RING * 256 * 4); hipMalloc(&xcc, WG * 4); hipMalloc(&ticks, 64);
    const int rounds = 4000;
    std::vector<unsigned> hx(WG);
    auto run = [&](int depth, int partner, int near, int gap) {
        hipMemset(ring, 0xFF, 2 * RING * 256 * 4); hipMemset(ticks, 0, 64);
        const size_t lds = 160 * 1024 - 256;
        if (depth == 1) { hipFuncSetAttribute((const void*)pingpong<1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); hipLaunchKernelGGL(pingpong<1>, dim3(WG), dim3(64), lds, 0, ring, xcc, ticks, partner, rounds, near, gap); }
        if (depth == 2) { hipFuncSetAttribute((const void*)pingpong<2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); hipLaunchKernelGGL(pingpong<2>, dim3(WG), dim3(64), lds, 0, ring, xcc, ticks, partner, rounds, near, gap); }
        if (depth == 3) { hipFuncSetAttribute((const void*)pingpong<3>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); hipLaunchKernelGGL(pingpong<3>, dim3(WG), dim3(64), lds, 0, ring, xcc, ticks, partner, rounds, near, gap); }
        if (depth == 4) { hipFuncSetAttribute((const void*)pingpong<4>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); hipLaunchKernelGGL(pingpong<4>, dim3(WG), dim3(64), lds, 0, ring, xcc, ticks, partner, rounds, near, gap); }
        hipDeviceSynchronize();
        unsigned long long h[3];
        hipMemcpy(h, ticks, 24, hipMemcpyDeviceToHost);
        hipMemcpy(hx.data(), xcc, WG * 4, hipMemcpyDeviceToHost);
        printf("  partner %3d (XCC %u vs %u)  %-13s depth %d gap %2d : one way %.3f us, %.2f polls per hand-off%s\n", partner, hx[0], hx[partner],
               near ? "plain store" : "write-through", depth, gap, h[0] * 0.01 / (2.0 * rounds), (double)h[1] / rounds, (h[2] + 8 > (unsigned long long)(2 + rounds) && h[2] < (unsigned long long)(10 + rounds)) ? "" : "  (payload check FAILED)");
    };
    printf("ping-pong of 1 KB messages, %d rounds\n", rounds);
    for (int partner : {1, 4, 8, 16}) {
        for (int gap : {0, 1, 2, 4, 8, 16}) run(1, partner, 0, gap);
        run(2, partner, 0, 2);
        run(3, partner, 0, 2);
        if ((partner & 7) == 0) {
            for (int gap : {0, 1, 2, 4, 8}) run(1, partner, 1, gap);
            run(2, partner, 1, 1);
        }
    }
    return 0;
}
