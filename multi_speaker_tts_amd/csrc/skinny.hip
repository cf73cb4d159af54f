// Skinny (M <= 32 rows) weight-streaming contractions for the recurrent steps (gfx950).
//
// One decoder step multiplies a [32, K] activation block with a [K, 4096] LSTM kernel
// (ZoneoutLSTMCell.py:228, matmul of concat([inputs, m_prev]) with the kernel) - 29-34 MB of fp32
// weights per product, 801 dependent steps.  A general tiled GEMM leaves the chip empty at M = 32
// (32 workgroups); here the work is cut so that >= 224 workgroups stream disjoint weight slabs:
//   skinny_fwd : P[ks][M][N]  = X[M, K-slice ks] . W[K-slice ks, N]      (64-column strips x K-splits)
//   skinny_bwd : P[ns][M][R]  = dG[M, N-slice ns] . W[R, N-slice ns]^T   (16-row strips x N-splits)
// Partials are summed by the consumer kernel (fixed order -> deterministic, no atomics, no zeroing).
// Arithmetic is v_mfma_f32_16x16x4_f32 (exact fp32).  Every lane loads float4: the four k's (fwd:
// four columns) of a float4 are fed to four MFMAs with a permuted k (column) assignment, so a wave's
// load instruction covers 4 rows x 256 B (fwd) / 16 rows x 64 B (bwd) of the row-major kernel.
#include "common.h"

namespace mstts {

typedef float f32x4 __attribute__((ext_vector_type(4)));

// ---------------------------------------------------------------------------------------------
// forward: strip of 64 columns, K-slice of KL rows (KL % 16 == 0), 4 waves split the slice
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void skinny_fwd_kernel(const float* __restrict__ X, long ldx, const float* __restrict__ W, long ldw,
                                                         float* __restrict__ P, long pstride, int M, int N, int K, int KL) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int strip = blockIdx.x, ks = blockIdx.y, mb = blockIdx.z;
    const int n0 = strip * 64, kb = ks * KL, m0 = mb * 32;
    const int klen = min(KL, K - kb);                 // multiple of 4 (checked by the host)
    const int lds_ld = KL + 4;                        // == 4 (mod 32) when KL % 32 == 0: <= 2-way on the A reads
    // stage X[m0 .. m0+32, kb .. kb+klen) row-major into LDS (rows >= M are zero)
    const int kl4 = KL / 4;
    for (int i = threadIdx.x; i < 32 * kl4; i += 256) {
        const int b = i / kl4, k4 = i % kl4;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (m0 + b < M && k4 * 4 < klen) v = *reinterpret_cast<const float4*>(X + (long)(m0 + b) * ldx + kb + k4 * 4);
        *reinterpret_cast<float4*>(smem + b * lds_ld + k4 * 4) = v;
    }
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int j = lane & 15, kq = lane >> 4;
    const int wl = (KL / 16) * 4;                     // rows per wave
    const int wk0 = wave * wl;
    f32x4 acc[2][4];
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int m = 0; m < 4; ++m) acc[t][m] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const bool col_ok = n0 + 4 * j + 3 < N;
    const float* wp = W + (long)(kb + wk0 + kq) * ldw + n0 + 4 * j;
    const float* xa = smem + j * lds_ld + wk0 + kq;
    const float* xb = xa + 16 * lds_ld;
#pragma unroll 4
    for (int kk = 0; kk < wl; kk += 4) {
        float4 bv = make_float4(0.f, 0.f, 0.f, 0.f);
        if (col_ok && wk0 + kk + kq < klen) bv = *reinterpret_cast<const float4*>(wp + (long)kk * ldw);
        const float a0 = xa[kk], a1 = xb[kk];
        acc[0][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0, bv.x, acc[0][0], 0, 0, 0);
        acc[1][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1, bv.x, acc[1][0], 0, 0, 0);
        acc[0][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0, bv.y, acc[0][1], 0, 0, 0);
        acc[1][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1, bv.y, acc[1][1], 0, 0, 0);
        acc[0][2] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0, bv.z, acc[0][2], 0, 0, 0);
        acc[1][2] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1, bv.z, acc[1][2], 0, 0, 0);
        acc[0][3] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0, bv.w, acc[0][3], 0, 0, 0);
        acc[1][3] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1, bv.w, acc[1][3], 0, 0, 0);
    }
    // cross-wave reduction through LDS: red[wave][row 32][col 64 (+1)]
    __syncthreads();
    float* red = smem;                                 // 4 * 32 * 65 floats <= 32 * (KL + 4) needs KL >= 256 ... host sizes smem
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
            for (int r = 0; r < 4; ++r)
                red[(wave * 32 + 16 * t + kq * 4 + r) * 65 + 4 * j + m] = acc[t][m][r];
    __syncthreads();
    float* out = P + (long)ks * pstride;
    for (int i = threadIdx.x; i < 32 * 64; i += 256) {
        const int b = i >> 6, c = i & 63;
        if (m0 + b < M && n0 + c < N) {
            const float v = red[b * 65 + c] + red[(32 + b) * 65 + c] + red[(64 + b) * 65 + c] + red[(96 + b) * 65 + c];
            out[(long)(m0 + b) * N + n0 + c] = v;
        }
    }
}

// ---------------------------------------------------------------------------------------------
// backward: strip of 16 kernel rows, N-slice of NL columns (NL % 64 == 0), 4 waves split the slice
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void skinny_bwd_kernel(const float* __restrict__ dG, long ldg, const float* __restrict__ W, long ldw,
                                                         float* __restrict__ P, long pstride, int M, int R, int N, int NL) {
    __shared__ float red[4][32][17];
    const int r0 = blockIdx.x * 16, ns = blockIdx.y, m0 = blockIdx.z * 32;
    const int nb = ns * NL;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int j = lane & 15, kq = lane >> 4;
    const int wl = NL / 4;                             // columns per wave, multiple of 16
    const int c0 = nb + wave * wl + 4 * kq;
    const bool row_ok = r0 + j < R;
    const bool a0_ok = m0 + j < M, a1_ok = m0 + 16 + j < M;
    const float* wp = W + (long)(r0 + j) * ldw + c0;
    const float* g0 = dG + (long)(m0 + j) * ldg + c0;
    const float* g1 = dG + (long)(m0 + 16 + j) * ldg + c0;
    f32x4 acc0 = (f32x4){0.f, 0.f, 0.f, 0.f}, acc1 = acc0, acc2 = acc0, acc3 = acc0;
    const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll 4
    for (int cc = 0; cc < wl; cc += 16) {
        const bool in = c0 + cc + 3 < N;
        const float4 bv = (row_ok && in) ? *reinterpret_cast<const float4*>(wp + cc) : z4;
        const float4 av0 = (a0_ok && in) ? *reinterpret_cast<const float4*>(g0 + cc) : z4;
        const float4 av1 = (a1_ok && in) ? *reinterpret_cast<const float4*>(g1 + cc) : z4;
        acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(av0.x, bv.x, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(av1.x, bv.x, acc1, 0, 0, 0);
        acc2 = __builtin_amdgcn_mfma_f32_16x16x4f32(av0.y, bv.y, acc2, 0, 0, 0);
        acc3 = __builtin_amdgcn_mfma_f32_16x16x4f32(av1.y, bv.y, acc3, 0, 0, 0);
        acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(av0.z, bv.z, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(av1.z, bv.z, acc1, 0, 0, 0);
        acc2 = __builtin_amdgcn_mfma_f32_16x16x4f32(av0.w, bv.w, acc2, 0, 0, 0);
        acc3 = __builtin_amdgcn_mfma_f32_16x16x4f32(av1.w, bv.w, acc3, 0, 0, 0);
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        red[wave][kq * 4 + r][j] = acc0[r] + acc2[r];
        red[wave][16 + kq * 4 + r][j] = acc1[r] + acc3[r];
    }
    __syncthreads();
    float* out = P + (long)ns * pstride;
    for (int i = threadIdx.x; i < 32 * 16; i += 256) {
        const int b = i >> 4, c = i & 15;
        if (m0 + b < M && r0 + c < R)
            out[(long)(m0 + b) * R + r0 + c] = red[0][b][c] + red[1][b][c] + red[2][b][c] + red[3][b][c];
    }
}

}  // namespace mstts
using namespace mstts;

extern "C" int32_t mstts_skinny_fwd_splits(int64_t N, int64_t K) {
    // K-splits so that strips * splits ~ 256 workgroups; each slice a multiple of 32 rows
    if (N <= 0 || K <= 0 || K % 32 != 0) return 0;
    const long strips = (N + 63) / 64;
    long ks = 256 / strips;
    if (ks < 1) ks = 1;
    while (ks > 1 && (K % (ks * 32) != 0)) --ks;
    return (int32_t)ks;
}

extern "C" int mstts_skinny_fwd(const float* X, int64_t ldx, const float* W, int64_t ldw, float* P, int64_t pstride, int64_t M, int64_t N,
                                int64_t K, int32_t ksplit, mstts_stream_t s) {
    MSTTS_REQUIRE(X && W && P && M >= 1 && N >= 1, MSTTS_ERR_SHAPE, "skinny_fwd: bad arguments");
    MSTTS_REQUIRE(ksplit >= 1 && K % (ksplit * 32L) == 0, MSTTS_ERR_SHAPE, "skinny_fwd: K must be a multiple of 32*ksplit");
    MSTTS_REQUIRE(N % 4 == 0 && ldw % 4 == 0 && ldx % 4 == 0 && aligned16(X) && aligned16(W), MSTTS_ERR_ALIGN,
                  "skinny_fwd: float4 alignment (N, ldx, ldw multiples of 4; 16-byte aligned pointers)");
    const int KL = (int)(K / ksplit);
    size_t lds = sizeof(float) * (size_t)32 * (KL + 4);
    const size_t red = sizeof(float) * 4 * 32 * 65;
    if (lds < red) lds = red;
    MSTTS_REQUIRE(lds <= 160 * 1024, MSTTS_ERR_SHAPE, "skinny_fwd: K slice too large for LDS (raise ksplit)");
    dim3 grid((unsigned)((N + 63) / 64), (unsigned)ksplit, (unsigned)((M + 31) / 32));
    static bool lds_attr_set = false;
    if (!lds_attr_set) {
        hipFuncSetAttribute((const void*)skinny_fwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        lds_attr_set = true;
    }
    hipLaunchKernelGGL(skinny_fwd_kernel, grid, dim3(256), lds, (hipStream_t)s, X, (long)ldx, W, (long)ldw, P,
                       (long)(pstride > 0 ? pstride : M * N), (int)M, (int)N, (int)K, KL);
    MSTTS_CHECK_LAUNCH("skinny_fwd");
    return MSTTS_OK;
}

extern "C" int32_t mstts_skinny_bwd_splits(int64_t R, int64_t N) {
    if (R <= 0 || N <= 0 || N % 64 != 0) return 0;
    const long strips = (R + 15) / 16;
    long ns = (256 + strips - 1) / strips;
    if (ns < 1) ns = 1;
    while (ns > 1 && (N % (ns * 64) != 0)) --ns;
    return (int32_t)ns;
}

extern "C" int mstts_skinny_bwd(const float* dG, int64_t ldg, const float* W, int64_t ldw, float* P, int64_t pstride, int64_t M, int64_t R,
                                int64_t N, int32_t nsplit, mstts_stream_t s) {
    MSTTS_REQUIRE(dG && W && P && M >= 1 && R >= 1, MSTTS_ERR_SHAPE, "skinny_bwd: bad arguments");
    MSTTS_REQUIRE(nsplit >= 1 && N % (nsplit * 64L) == 0, MSTTS_ERR_SHAPE, "skinny_bwd: N must be a multiple of 64*nsplit");
    MSTTS_REQUIRE(ldw % 4 == 0 && ldg % 4 == 0 && aligned16(dG) && aligned16(W), MSTTS_ERR_ALIGN, "skinny_bwd: float4 alignment");
    const int NL = (int)(N / nsplit);
    dim3 grid((unsigned)((R + 15) / 16), (unsigned)nsplit, (unsigned)((M + 31) / 32));
    hipLaunchKernelGGL(skinny_bwd_kernel, grid, dim3(256), 0, (hipStream_t)s, dG, (long)ldg, W, (long)ldw, P,
                       (long)(pstride > 0 ? pstride : M * R), (int)M, (int)R, (int)N, NL);
    MSTTS_CHECK_LAUNCH("skinny_bwd");
    return MSTTS_OK;
}
