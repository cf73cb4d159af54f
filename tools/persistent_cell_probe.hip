// What ONE STAGE of a persistent, weights-stationary decoder loop would cost on gfx950 - the number the next design step hangs on.
// Today a decoder cell step is a launch of its own: 12.1-12.7 us, of which ~4.5 us are the kernel boundary and the first-load latency
// and ~4.5 us the 29-34 MB weight stream.  A persistent kernel keeps each workgroup's 128 KB slice of the cell kernel in LDS for all
// 801 steps, and pays per stage: a grid-wide barrier with data visibility (hierarchical form of tools/gridsync_probe.hip: 3.2 us),
// the fetch of the [32, 2048] activation block that the other 255 workgroups have just written (256 KB per workgroup, through the
// CU's 64 B/clk path), the exact-fp32 MFMA work (v_mfma_f32_16x16x4_f32, 256 per wave = 3.4 us) and a small epilogue.
// This probe runs exactly that stage `rounds` times inside one launch: 256 workgroups x 4 waves, workgroup w owns 16 gate columns
// (4 units x 4 gates) of a [2048, 4096] kernel, reads the whole activation block with agent-scope loads, multiplies, applies an
// LSTM-like epilogue and writes its 32 x 4 outputs write-through into the other buffer's columns 4w .. 4w+3 (the h half of the next
// stage's input); columns 1024 .. 2047 stay constant (the x half).  Results are checked against a host fp64 recurrence.
// Two ways to read what the other workgroups wrote: agent-scope loads (past the XCD's L2), or a ring of eight buffers with plain
// loads and an L2 invalidate every eighth round.  -DNWAVES=8 builds the 8-wave form (K split eight ways).
// Measured (MI355X, round 2), per stage, barrier included:
//   activation block row-major (a wave load touches 16 rows 8 KB apart - the same L2 channel):  11.4-12.2 us
//   activation block PACKED (a wave load = one contiguous 1 KB, what the production cells use):   8.4 us  (6.5 without the barrier);
//   ring + plain loads 9.3-10.2 us; 4 or 8 waves make no difference.
// Today's launch per cell step costs 12.1-12.7 us: a persistent, weights-stationary loop would take about 3.8 us off every cell
// stage (and the BPTT's pointwise launches would disappear into its stages).  It needs the attention step as a third stage of the
// same kernel and one cell's slice in registers (246 KB of kernel per CU against 160 KB of LDS) - the next round's first item.
//   hipcc --offload-arch=gfx950 -O3 tools/persistent_cell_probe.hip -o /tmp/pcp && /tmp/pcp
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
#ifndef NWAVES
#define NWAVES 4
#endif
constexpr int WG = 256, NWV = NWAVES, TH = 64 * NWV, ROWS = 32, K = 2048, HU = 1024;      // HU: units = columns rewritten per round
constexpr int KW = K / NWV;                                             // reduction slice of a wave
constexpr int NCH = KW / 16;                                            // 16-deep chunks per wave (32 with 4 waves, 16 with 8)

// element (row, k) of the [32, K] activation block in the packed order
__host__ __device__ inline long apos(int row, int k) {
    return ((((long)(k / KW) * NCH + (k % KW) / 16) * 2 + row / 16) * 256) + (((k % 16) / 4) * 16 + row % 16) * 4 + k % 4;
}

__device__ __forceinline__ void grid_barrier(unsigned long long* counter, int wg, int r) {
    const int x = wg & 7;
    unsigned long long* xc = counter + 16 * (1 + x);
    unsigned long long* flag = counter + 16 * (9 + x);
    const unsigned long long a = __hip_atomic_fetch_add(xc, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (a + 1 == (unsigned long long)r * (WG / 8)) {
        const unsigned long long g = __hip_atomic_fetch_add(counter, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (g + 1 == (unsigned long long)r * 8)
            for (int y = 0; y < 8; ++y) __hip_atomic_store(counter + 16 * (9 + y), (unsigned long long)r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    unsigned spins = 0;
    while (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (unsigned long long)r && spins < 4000000u) {
        __builtin_amdgcn_s_sleep(1);
        ++spins;
    }
}

// act: two buffers [ROWS][K]; W: [K][16 * WG] row-major (column 16 w + 4 q + u = gate q of unit 4 w + u)
__global__ __launch_bounds__(TH) void persistent_kernel(float* act, const float* __restrict__ W, unsigned long long* counter, int rounds, int with_barrier, int inv_mode) {
    extern __shared__ __attribute__((aligned(16))) float smem[];        // [4 waves][NCH][64 lanes][4] weights = 128 KB, then [4][32][17] reduce
    float* red = smem + NWV * NCH * 256;
    const int wg = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int j = lane & 15, kq = lane >> 4;
    // ---- once: this workgroup's kernel slice into LDS in consumption order: (wave, chunk, lane) -> W[wave*KW + 16*chunk + 4*kq + e][16*wg + j]
    for (int c = 0; c < NCH; ++c) {
        f32x4 v;
        for (int e = 0; e < 4; ++e) v[e] = W[(long)(wave * KW + 16 * c + 4 * kq + e) * (16 * WG) + 16 * wg + j];
        *reinterpret_cast<f32x4*>(smem + ((wave * NCH + c) * 64 + lane) * 4) = v;
    }
    __syncthreads();
    for (int r = 1; r <= rounds; ++r) {
        // inv_mode 2: a ring of eight buffers, so that an XCD's L2 has to forget its stale copies only once per eight rounds
        const int ring = inv_mode == 2 ? 7 : 1;
        const float* src = act + (long)((r - 1) & ring) * ROWS * K;
        float* dst = act + (long)(r & ring) * ROWS * K;
        // ---- activation rows j and 16 + j, this wave's K slice, 16 bytes per lane and chunk, in four batches of 8 chunks
        f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = acc0, acc2 = acc0, acc3 = acc0;
        // packed block (as the production cells use): a wave's load instruction covers one contiguous 1 KB
        const float* a0 = src + (long)(wave * NCH) * 512 + lane * 4;
        const float* a1 = a0 + 256;
        // software pipeline: batch b + 1 (8 chunks = 16 loads) is requested before batch b is multiplied; loads return in order, so
        // vmcnt(16) means "everything but the newest batch has arrived".  The waits name the registers they release, so that no MFMA
        // can be scheduled ahead of them.
        f32x4 pa[2][8], pb[2][8];
#define PCP_ISSUE(buf, b)                                                                                                   \
        _Pragma("unroll") for (int c = 0; c < 8; ++c) {                                                                      \
            if (inv_mode) {                                                                                                  \
                asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(pa[buf][c]) : "v"(a0 + 512 * (8 * (b) + c)) : "memory");     \
                asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(pb[buf][c]) : "v"(a1 + 512 * (8 * (b) + c)) : "memory");     \
            } else {                                                                                                         \
                asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=v"(pa[buf][c]) : "v"(a0 + 512 * (8 * (b) + c)) : "memory"); \
                asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=v"(pb[buf][c]) : "v"(a1 + 512 * (8 * (b) + c)) : "memory"); \
            }                                                                                                                \
        }
#define PCP_WAIT(buf, n)                                                                                                    \
        asm volatile("s_waitcnt vmcnt(" #n ")" : "+v"(pa[buf][0]), "+v"(pa[buf][1]), "+v"(pa[buf][2]), "+v"(pa[buf][3]), "+v"(pa[buf][4]), "+v"(pa[buf][5]), \
                     "+v"(pa[buf][6]), "+v"(pa[buf][7]), "+v"(pb[buf][0]), "+v"(pb[buf][1]), "+v"(pb[buf][2]), "+v"(pb[buf][3]), "+v"(pb[buf][4]),         \
                     "+v"(pb[buf][5]), "+v"(pb[buf][6]), "+v"(pb[buf][7]) :: "memory")
#define PCP_MFMA(buf, b)                                                                                                    \
        _Pragma("unroll") for (int c = 0; c < 8; ++c) {                                                                      \
            const f32x4 bv = *reinterpret_cast<const f32x4*>(smem + ((wave * NCH + 8 * (b) + c) * 64 + lane) * 4);          \
            acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(pa[buf][c][0], bv[0], acc0, 0, 0, 0);                               \
            acc2 = __builtin_amdgcn_mfma_f32_16x16x4f32(pa[buf][c][1], bv[1], acc2, 0, 0, 0);                               \
            acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(pa[buf][c][2], bv[2], acc0, 0, 0, 0);                               \
            acc2 = __builtin_amdgcn_mfma_f32_16x16x4f32(pa[buf][c][3], bv[3], acc2, 0, 0, 0);                               \
            acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(pb[buf][c][0], bv[0], acc1, 0, 0, 0);                               \
            acc3 = __builtin_amdgcn_mfma_f32_16x16x4f32(pb[buf][c][1], bv[1], acc3, 0, 0, 0);                               \
            acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(pb[buf][c][2], bv[2], acc1, 0, 0, 0);                               \
            acc3 = __builtin_amdgcn_mfma_f32_16x16x4f32(pb[buf][c][3], bv[3], acc3, 0, 0, 0);                               \
        }
        // inv_mode: drop this CU's L1 and the XCD's stale L2 copies once per round (the writers stored write-through to memory), then
        // read with plain loads: the first CU of an XCD to touch a line brings it into that XCD's L2, the other 31 hit there
        if (inv_mode == 1 || (inv_mode == 2 && (r & 7) == 1)) asm volatile("buffer_inv sc1" ::: "memory");
        else if (inv_mode == 2) asm volatile("buffer_inv sc0" ::: "memory");
        PCP_ISSUE(0, 0)
        PCP_ISSUE(1, 1)
        PCP_WAIT(0, 16);
        PCP_MFMA(0, 0)
        if (NCH > 16) {
            PCP_ISSUE(0, 2)
            PCP_WAIT(1, 16);
            PCP_MFMA(1, 1)
            PCP_ISSUE(1, 3)
            PCP_WAIT(0, 16);
            PCP_MFMA(0, 2)
            PCP_WAIT(1, 0);
            PCP_MFMA(1, 3)
        } else {                      // 8 waves: the wave's whole slice was in flight from the start
            PCP_WAIT(1, 0);
            PCP_MFMA(1, 1)
        }
        // ---- reduce the four K slices, epilogue, write-through stores of the 32 x 4 outputs
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
            for (int q = 0; q < 4; ++q) {
                g[q] = 0.f;
#pragma unroll
                for (int w2 = 0; w2 < NWV; ++w2) g[q] += red[(w2 * 32 + row) * 17 + 4 * q + u];
            }
            const float si = 1.f / (1.f + __expf(-g[0])), so = 1.f / (1.f + __expf(-g[3]));
            const float h = so * tanhf(si * tanhf(g[1]) + 0.5f * g[2]);
            __hip_atomic_store(dst + apos(row, 4 * wg + u), h, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else if (threadIdx.x < 128 + 32 && r == 1) {
            // the constant x half is copied once into the second buffer by workgroup-owned pieces (round 1 only)
        }
        __builtin_amdgcn_s_waitcnt(0);
        __syncthreads();
        if (with_barrier && threadIdx.x == 0) grid_barrier(counter, wg, r);
        __syncthreads();
    }
}

int main() {
    const long NW = (long)K * 16 * WG;
    std::vector<float> hW(NW), hA(8L * ROWS * K);
    unsigned s = 12345u;
    auto rnd = [&]() { s = s * 1664525u + 1013904223u; return ((s >> 8) & 0xffff) / 65536.0f - 0.5f; };
    for (auto& v : hW) v = rnd() * 0.08f;
    for (long i = 0; i < (long)ROWS * K; ++i) { hA[i] = rnd(); for (int q = 1; q < 8; ++q) hA[q * (long)ROWS * K + i] = hA[i]; }     // both buffers: same x half, h half overwritten
    float *dW, *dA; unsigned long long* counter;
    hipMalloc(&dW, NW * 4); hipMalloc(&dA, 8L * ROWS * K * 4); hipMalloc(&counter, 8 * 16 * 17);
    hipMemcpy(dW, hW.data(), NW * 4, hipMemcpyHostToDevice);
    const size_t lds = (size_t)(NWV * NCH * 256 + NWV * 32 * 17) * 4;
    hipFuncSetAttribute((const void*)persistent_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int inv_mode = 0; inv_mode < 3; inv_mode += 2)
    for (int with_barrier = 1; with_barrier >= 0; --with_barrier)
        for (int rounds : {4, 104, 404}) {
            {
                std::vector<float> pk(8L * ROWS * K);
                for (int q = 0; q < 8; ++q) for (int row = 0; row < ROWS; ++row) for (int k = 0; k < K; ++k) pk[q * (long)ROWS * K + apos(row, k)] = hA[q * (long)ROWS * K + (long)row * K + k];
                hipMemcpy(dA, pk.data(), 8L * ROWS * K * 4, hipMemcpyHostToDevice);
            }
            hipMemset(counter, 0, 8 * 16 * 17);
            hipDeviceSynchronize();
            hipEventRecord(e0);
            hipLaunchKernelGGL(persistent_kernel, dim3(WG), dim3(TH), lds, 0, dA, dW, counter, rounds, with_barrier, inv_mode);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            double err = -1.0;
            if (with_barrier && rounds == 4) {                   // host fp64 recurrence of four rounds (buffers are re-read: stale copies would show)
                std::vector<double> a(hA.begin(), hA.begin() + (long)ROWS * K), b(a);
                for (int r = 1; r <= 4; ++r) {
                    std::vector<double>& src = (r & 1) ? a : b; std::vector<double>& dst = (r & 1) ? b : a;
                    for (int row = 0; row < ROWS; ++row)
                        for (int unit = 0; unit < HU; ++unit) {
                            double g[4];
                            for (int q = 0; q < 4; ++q) {
                                double t = 0;
                                for (int k = 0; k < K; ++k) t += src[(long)row * K + k] * hW[(long)k * (16 * WG) + 16 * (unit >> 2) + 4 * q + (unit & 3)];
                                g[q] = t;
                            }
                            const double si = 1 / (1 + exp(-g[0])), so = 1 / (1 + exp(-g[3]));
                            dst[(long)row * K + unit] = so * tanh(si * tanh(g[1]) + 0.5 * g[2]);
                        }
                }
                std::vector<float> out(8L * ROWS * K);
                hipMemcpy(out.data(), dA, 8L * ROWS * K * 4, hipMemcpyDeviceToHost);
                const long fin = (inv_mode == 2 ? 4L : 0L) * ROWS * K;       // round 4 writes buffer 4 of the ring, buffer 0 of the pair
                err = 0;
                for (int row = 0; row < ROWS; ++row)
                    for (int unit = 0; unit < HU; ++unit) err = fmax(err, fabs(out[fin + apos(row, unit)] - a[(long)row * K + unit]));   // round 4 writes buffer 0
            }
            printf("%d waves, %s %s: %3d stages %8.1f us total, %6.2f us per stage", NWV, inv_mode == 2 ? "ring of 8, L2 inv every 8th" : inv_mode ? "L2 invalidate + plain loads" : "agent-scope loads          ", with_barrier ? "with the grid barrier   " : "without (racy, timing only)", rounds, ms * 1e3,
                   ms * 1e3 / rounds);
            if (err >= 0) printf("   max |err| vs host fp64 after 4 stages: %.2e", err);
            printf("\n");
        }
    return 0;
}
