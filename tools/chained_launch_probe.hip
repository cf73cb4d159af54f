// Can a chain of DEPENDENT cell-step launches hide its fixed costs without becoming one persistent kernel?
// Each launch is one cell-shaped stage as in tools/persistent_cell_probe.hip, but the 128 KB kernel slice of a workgroup is STREAMED
// from memory every launch, as in the product (two 33.5 MB kernels alternate, so the stream comes from the Infinity Cache, not L2).
//   mode 0  today's form: every launch on one stream, the queue's kernel boundary orders producer and consumer (plain loads);
//   mode 1  chained: launches alternate between TWO streams, so launch r+1 starts while launch r still runs.  It requests its whole
//           kernel slice first (it depends on nothing), then waits until all workgroups of launch r have arrived on a counter
//           (hierarchical: per-XCD counters, a global one, per-XCD flags), then reads launch r's outputs with agent-scope loads;
//           outputs are stored write-through and followed by the arrival.  Launch r+2 follows launch r on the same stream, so at most
//           two launches are resident, and the earlier one never waits for the later one.
// Results are checked against a host fp64 recurrence (4 launches).
//   hipcc --offload-arch=gfx950 -O3 tools/chained_launch_probe.hip -o /tmp/clp && /tmp/clp
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int WG = 256, TH = 256, ROWS = 32, K = 2048, HU = 1024;
constexpr int KW = K / 4, NCH = KW / 16;          // 32 chunks of 16 k per wave

__host__ __device__ inline long apos(int row, int k) {
    return ((((long)(k / KW) * NCH + (k % KW) / 16) * 2 + row / 16) * 256) + (((k % 16) / 4) * 16 + row % 16) * 4 + k % 4;
}
// packed kernel: (wg, wave, chunk, lane, e) <- W[wave*KW + 16*chunk + 4*(lane>>4) + e][16*wg + (lane&15)]
__host__ __device__ inline long wpos(int wg, int wave, int c, int lane) { return ((((long)wg * 4 + wave) * NCH + c) * 64 + lane) * 4; }

__device__ __forceinline__ void arrive(unsigned long long* counter, int wg, int r) {
    const int x = wg & 7;
    const unsigned long long a = __hip_atomic_fetch_add(counter + 16 * (1 + x), 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (a + 1 == (unsigned long long)r * (WG / 8)) {
        const unsigned long long g = __hip_atomic_fetch_add(counter, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (g + 1 == (unsigned long long)r * 8)
            for (int y = 0; y < 8; ++y) __hip_atomic_store(counter + 16 * (9 + y), (unsigned long long)r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}
__device__ __forceinline__ bool wait_for(unsigned long long* counter, int wg, int r) {
    const unsigned long long* flag = counter + 16 * (9 + (wg & 7));
    unsigned spins = 0;
    while (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (unsigned long long)r && spins < 4000000u) {
        __builtin_amdgcn_s_sleep(1);
        ++spins;
    }
    return spins < 4000000u;
}

template <bool CHAINED>
__global__ __launch_bounds__(TH) void stage_kernel(float* act, const float* __restrict__ Wp, unsigned long long* counter, int r, int* errors) {
    __shared__ float red[4 * 32 * 17];
    const int wg = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int j = lane & 15, kq = lane >> 4;
    const float* src = act + (long)((r - 1) & 1) * ROWS * K;
    float* dst = act + (long)(r & 1) * ROWS * K;
    // ---- the kernel slice: 32 float4 per lane, all requested before anything else (nothing here depends on the previous launch)
    f32x4 wreg[NCH];
    const float* wp = Wp + wpos(wg, wave, 0, lane);
#pragma unroll
    for (int c = 0; c < NCH; ++c) wreg[c] = *reinterpret_cast<const f32x4*>(wp + 256 * c);
    __builtin_amdgcn_sched_barrier(0);
    if (CHAINED) {
        __shared__ int ok;
        if (threadIdx.x == 0) ok = wait_for(counter, wg, r - 1) ? 1 : 0;
        __syncthreads();
        if (!ok) { if (threadIdx.x == 0) atomicAdd(errors, 1000); return; }
    }
    f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = acc0, acc2 = acc0, acc3 = acc0;
    const float* a0 = src + (long)(wave * NCH) * 512 + lane * 4;
    const float* a1 = a0 + 256;
    f32x4 pa[2][4], pb[2][4];            // two batches of four chunks in flight: with the 128 kernel registers the wave stays under 256 VGPRs,
                                         // so that two workgroups (one of each launch) fit a CU
#define CLP_ISSUE(buf, b)                                                                                                       \
    _Pragma("unroll") for (int c = 0; c < 4; ++c) {                                                                              \
        if (CHAINED) {                                                                                                           \
            asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=v"(pa[buf][c]) : "v"(a0 + 512 * (4 * (b) + c)) : "memory");    \
            asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=v"(pb[buf][c]) : "v"(a1 + 512 * (4 * (b) + c)) : "memory");    \
        } else {                                                                                                                 \
            asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(pa[buf][c]) : "v"(a0 + 512 * (4 * (b) + c)) : "memory");        \
            asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(pb[buf][c]) : "v"(a1 + 512 * (4 * (b) + c)) : "memory");        \
        }                                                                                                                        \
    }
#define CLP_WAIT(buf, n)                                                                                                        \
    asm volatile("s_waitcnt vmcnt(" #n ")" : "+v"(pa[buf][0]), "+v"(pa[buf][1]), "+v"(pa[buf][2]), "+v"(pa[buf][3]),           \
                 "+v"(pb[buf][0]), "+v"(pb[buf][1]), "+v"(pb[buf][2]), "+v"(pb[buf][3]) :: "memory")
#define CLP_MFMA(buf, b)                                                                                                        \
    _Pragma("unroll") for (int c = 0; c < 4; ++c) {                                                                              \
        const f32x4 bv = wreg[4 * (b) + c];                                                                                      \
        acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(pa[buf][c][0], bv[0], acc0, 0, 0, 0);                                        \
        acc2 = __builtin_amdgcn_mfma_f32_16x16x4f32(pa[buf][c][1], bv[1], acc2, 0, 0, 0);                                        \
        acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(pa[buf][c][2], bv[2], acc0, 0, 0, 0);                                        \
        acc2 = __builtin_amdgcn_mfma_f32_16x16x4f32(pa[buf][c][3], bv[3], acc2, 0, 0, 0);                                        \
        acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(pb[buf][c][0], bv[0], acc1, 0, 0, 0);                                        \
        acc3 = __builtin_amdgcn_mfma_f32_16x16x4f32(pb[buf][c][1], bv[1], acc3, 0, 0, 0);                                        \
        acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(pb[buf][c][2], bv[2], acc1, 0, 0, 0);                                        \
        acc3 = __builtin_amdgcn_mfma_f32_16x16x4f32(pb[buf][c][3], bv[3], acc3, 0, 0, 0);                                        \
    }
    CLP_ISSUE(0, 0)
    CLP_ISSUE(1, 1)
    CLP_WAIT(0, 8);
    CLP_MFMA(0, 0)
    CLP_ISSUE(0, 2)
    CLP_WAIT(1, 8);
    CLP_MFMA(1, 1)
    CLP_ISSUE(1, 3)
    CLP_WAIT(0, 8);
    CLP_MFMA(0, 2)
    CLP_ISSUE(0, 4)
    CLP_WAIT(1, 8);
    CLP_MFMA(1, 3)
    CLP_ISSUE(1, 5)
    CLP_WAIT(0, 8);
    CLP_MFMA(0, 4)
    CLP_ISSUE(0, 6)
    CLP_WAIT(1, 8);
    CLP_MFMA(1, 5)
    CLP_ISSUE(1, 7)
    CLP_WAIT(0, 8);
    CLP_MFMA(0, 6)
    CLP_WAIT(1, 0);
    CLP_MFMA(1, 7)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        red[(wave * 32 + kq * 4 + q) * 17 + j] = acc0[q] + acc2[q];
        red[(wave * 32 + 16 + kq * 4 + q) * 17 + j] = acc1[q] + acc3[q];
    }
    __syncthreads();
    if (threadIdx.x < 128) {
        const int row = threadIdx.x >> 2, u = threadIdx.x & 3;
        float g[4];
#pragma unroll
        for (int q = 0; q < 4; ++q)
            g[q] = (red[(0 * 32 + row) * 17 + 4 * q + u] + red[(1 * 32 + row) * 17 + 4 * q + u]) + (red[(2 * 32 + row) * 17 + 4 * q + u] + red[(3 * 32 + row) * 17 + 4 * q + u]);
        const float si = 1.f / (1.f + __expf(-g[0])), so = 1.f / (1.f + __expf(-g[3]));
        const float h = so * tanhf(si * tanhf(g[1]) + 0.5f * g[2]);
        if (CHAINED) __hip_atomic_store(dst + apos(row, 4 * wg + u), h, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        else dst[apos(row, 4 * wg + u)] = h;
    }
    if (CHAINED) {
        __builtin_amdgcn_s_waitcnt(0);
        __syncthreads();
        if (threadIdx.x == 0) arrive(counter, wg, r);
    }
}

int main() {
    const long NW = (long)K * 16 * WG;
    std::vector<float> hW[2] = {std::vector<float>(NW), std::vector<float>(NW)}, hA(2L * ROWS * K);
    unsigned s = 12345u;
    auto rnd = [&]() { s = s * 1664525u + 1013904223u; return ((s >> 8) & 0xffff) / 65536.0f - 0.5f; };
    for (int t = 0; t < 2; ++t) for (auto& v : hW[t]) v = rnd() * 0.08f;
    for (long i = 0; i < (long)ROWS * K; ++i) hA[i] = hA[(long)ROWS * K + i] = rnd();
    float *dW[2], *dA; unsigned long long* counter; int* errors;
    hipMalloc(&dA, 2L * ROWS * K * 4); hipMalloc(&counter, 8 * 16 * 17); hipMalloc(&errors, 4);
    for (int t = 0; t < 2; ++t) {                       // packed kernels
        std::vector<float> pk(NW);
        for (int wg = 0; wg < WG; ++wg) for (int wave = 0; wave < 4; ++wave) for (int c = 0; c < NCH; ++c) for (int lane = 0; lane < 64; ++lane) for (int e = 0; e < 4; ++e)
            pk[wpos(wg, wave, c, lane) + e] = hW[t][(long)(wave * KW + 16 * c + 4 * (lane >> 4) + e) * (16 * WG) + 16 * wg + (lane & 15)];
        hipMalloc(&dW[t], NW * 4);
        hipMemcpy(dW[t], pk.data(), NW * 4, hipMemcpyHostToDevice);
    }
    std::vector<float> pkA(2L * ROWS * K);
    for (int q = 0; q < 2; ++q) for (int row = 0; row < ROWS; ++row) for (int k = 0; k < K; ++k) pkA[q * (long)ROWS * K + apos(row, k)] = hA[(long)row * K + k];
    hipStream_t st[2]; hipStreamCreateWithFlags(&st[0], hipStreamNonBlocking); hipStreamCreateWithFlags(&st[1], hipStreamNonBlocking);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int mode = 0; mode < 2; ++mode)
        for (int rounds : {4, 404, 1604}) {
            hipMemcpy(dA, pkA.data(), 2L * ROWS * K * 4, hipMemcpyHostToDevice);
            hipMemset(counter, 0, 8 * 16 * 17); hipMemset(errors, 0, 4);
            hipDeviceSynchronize();
            hipEventRecord(e0, st[0]);
            for (int r = 1; r <= rounds; ++r) {
                if (mode == 0) hipLaunchKernelGGL(stage_kernel<false>, dim3(WG), dim3(TH), 0, st[0], dA, dW[r & 1], counter, r, errors);
                else hipLaunchKernelGGL(stage_kernel<true>, dim3(WG), dim3(TH), 0, st[r & 1], dA, dW[r & 1], counter, r, errors);
            }
            hipStreamSynchronize(st[1]);
            hipEventRecord(e1, st[0]); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            int err; hipMemcpy(&err, errors, 4, hipMemcpyDeviceToHost);
            double dev = -1.0;
            if (rounds == 4) {
                std::vector<double> a(hA.begin(), hA.begin() + (long)ROWS * K), b(a);
                for (int r = 1; r <= 4; ++r) {
                    std::vector<double>& src = (r & 1) ? a : b; std::vector<double>& dst = (r & 1) ? b : a;
                    const std::vector<float>& W = hW[r & 1];
                    for (int row = 0; row < ROWS; ++row)
                        for (int unit = 0; unit < HU; ++unit) {
                            double g[4];
                            for (int q = 0; q < 4; ++q) {
                                double t = 0;
                                for (int k = 0; k < K; ++k) t += src[(long)row * K + k] * W[(long)k * (16 * WG) + 16 * (unit >> 2) + 4 * q + (unit & 3)];
                                g[q] = t;
                            }
                            const double si = 1 / (1 + exp(-g[0])), so = 1 / (1 + exp(-g[3]));
                            dst[(long)row * K + unit] = so * tanh(si * tanh(g[1]) + 0.5 * g[2]);
                        }
                }
                std::vector<float> out(2L * ROWS * K);
                hipMemcpy(out.data(), dA, 2L * ROWS * K * 4, hipMemcpyDeviceToHost);
                dev = 0;
                for (int row = 0; row < ROWS; ++row)
                    for (int unit = 0; unit < HU; ++unit) dev = fmax(dev, fabs(out[apos(row, unit)] - a[(long)row * K + unit]));     // launch 4 writes buffer 0
            }
            printf("%s: %4d launches %9.1f us total, %6.2f us per launch, time-outs %d", mode ? "two streams, prefetch + arrival counter" : "one stream, kernel boundary            ",
                   rounds, ms * 1e3, ms * 1e3 / rounds, err);
            if (dev >= 0) printf("   max |err| vs host fp64 after 4 launches: %.2e", dev);
            printf("\n");
        }
    return 0;
}
