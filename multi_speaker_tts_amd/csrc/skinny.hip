// Skinny (M <= 32 rows) weight-streaming contractions for the recurrent steps (gfx950).
//
// One decoder step multiplies a [32, K] activation block with a [K, 4096] LSTM kernel
// (ZoneoutLSTMCell.py:228, matmul of concat([inputs, m_prev]) with the kernel) - 29-34 MB of fp32
// weights per product, 801 dependent steps, and the same again transposed in BPTT.  A general tiled
// GEMM leaves the chip empty at M = 32 (32 workgroups); here the work is cut so that ~256
// workgroups stream disjoint weight slabs:
//   skinny_fwd : P[ks][M][N] = X[M, K-slice ks] . W[K-slice ks, N]     (64-column strips x K-splits)
//   skinny_bwd : P[ns][M][R] = dG[M, N-slice ns] . W[R, N-slice ns]^T  (32-row strips x N-splits)
// Partials are summed by the consumer kernel (fixed order -> deterministic, no atomics, no zeroing).
//
// The kernels are latency-bound, not throughput-bound (each wave owns only ~30 KB of weights), so
// every wave issues ALL of its weight loads up front into registers (<= 32 float4 per lane, 1 wave
// per SIMD), the small activation slice goes through LDS once, and only then the MFMA chain runs.
// Arithmetic is v_mfma_f32_16x16x4_f32 (exact fp32).  A lane's float4 feeds four MFMAs with a
// permuted k (fwd: column) assignment, so one load instruction of a wave covers 4 rows x 256 B
// (fwd) or 16 rows x 64 B (bwd) of the row-major kernel.
#include "common.h"
#include <stdlib.h>

namespace mstts {

typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int MAX_IT = 32;        // weight float4 loads per lane held in registers

// ---------------------------------------------------------------------------------------------
// forward: strip of 64 columns, K-slice of KL rows (KL % 16 == 0, KL <= 512), 4 waves split the slice
// ---------------------------------------------------------------------------------------------
template <int NIT, bool TWO>      // NIT > 0: exact trip count (guard-free code); TWO: rows 16..31 of the block exist
__global__ __launch_bounds__(256) void skinny_fwd_kernel(const float* __restrict__ X, long ldx, const float* __restrict__ W, long ldw,
                                                         float* __restrict__ P, long pstride, int M, int N, int K, int KL) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int n0 = blockIdx.x * 64, ks = blockIdx.y, m0 = blockIdx.z * 32;
    const int kb = ks * KL;
    const int lds_ld = KL + 4;                        // == 4 (mod 32) when KL % 32 == 0: <= 2-way on the A reads
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int j = lane & 15, kq = lane >> 4;
    const int wl = KL / 4;                            // rows per wave (multiple of 4)
    constexpr bool EXACT = NIT > 0;
    constexpr int UNROLL = EXACT ? NIT : MAX_IT;
    const int nit = EXACT ? NIT : wl / 4;
    const int wk0 = wave * wl;
    // 1) activation slice first (the MFMA chain needs it before anything else): X[m0 .. m0+32, kb .. kb+KL),
    //    8 threads per row, all loads issued before the first LDS write
    const int kl4 = KL / 4;
    const int sb = threadIdx.x >> 3, sc = threadIdx.x & 7;
    const bool srow_live = m0 + sb < M;
    const float* xr = X + (long)(m0 + sb) * ldx + kb;
    f32x4 st[16];
#pragma unroll
    for (int q = 0; q < 16; ++q) {
        st[q] = (f32x4){0.f, 0.f, 0.f, 0.f};
        const int k4 = sc + 8 * q;
        if (k4 < kl4 && srow_live) st[q] = *reinterpret_cast<const f32x4*>(xr + k4 * 4);
    }
    // 2) all weight loads of this wave: row kb + wk0 + 4*it + kq, columns n0 + 4j .. 4j+3
    const bool col_ok = n0 + 4 * j + 3 < N;
    const float* wp = W + (long)(kb + wk0 + kq) * ldw + n0 + 4 * j;
    f32x4 wreg[UNROLL];
#pragma unroll
    for (int it = 0; it < UNROLL; ++it) {
        wreg[it] = (f32x4){0.f, 0.f, 0.f, 0.f};
#ifndef SKINNY_PROBE_NO_WLOAD
        if ((EXACT || it < nit) && col_ok) wreg[it] = *reinterpret_cast<const f32x4*>(wp + (long)(4 * it) * ldw);
#else
        wreg[it] = (f32x4){(float)it, 1.f, 2.f, (float)lane};
#endif
    }
    // 3) LDS write of the staged slice (waits only for the slice: the weight loads stay in flight)
#pragma unroll
    for (int q = 0; q < 16; ++q) {
        const int k4 = sc + 8 * q;
        if (k4 < kl4) *reinterpret_cast<f32x4*>(smem + sb * lds_ld + k4 * 4) = st[q];
    }
    __syncthreads();
    // 4) MFMA chain (iteration `it` waits only for weight load `it`)
    f32x4 acc[2][4];
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int m = 0; m < 4; ++m) acc[t][m] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const float* xa = smem + j * lds_ld + wk0 + kq;
    const float* xb = xa + 16 * lds_ld;
#pragma unroll
    for (int it = 0; it < UNROLL; ++it) {
        if (EXACT || it < nit) {
            const float a0 = xa[4 * it], a1 = TWO ? xb[4 * it] : 0.f;
            const f32x4 bv = wreg[it];
#ifdef SKINNY_PROBE_NO_MFMA
            acc[0][0] += a0 * bv; acc[1][0] += a1 * bv;
            continue;
#endif
            acc[0][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0, bv[0], acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0, bv[1], acc[0][1], 0, 0, 0);
            acc[0][2] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0, bv[2], acc[0][2], 0, 0, 0);
            acc[0][3] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0, bv[3], acc[0][3], 0, 0, 0);
            if (TWO) {
                acc[1][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1, bv[0], acc[1][0], 0, 0, 0);
                acc[1][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1, bv[1], acc[1][1], 0, 0, 0);
                acc[1][2] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1, bv[2], acc[1][2], 0, 0, 0);
                acc[1][3] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1, bv[3], acc[1][3], 0, 0, 0);
            }
        }
    }
    // 4) cross-wave reduction through LDS: red[wave][row 32][col 64 (+1)]
    __syncthreads();
    float* red = smem;
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
// backward: strip of 32 kernel rows, N-slice of NL columns (NL % 32 == 0, NL <= 1024).
// wave w: row tile (w & 1) of 16 rows, column half (w >> 1) of the slice.
// ---------------------------------------------------------------------------------------------
// PACKED: W is a derived copy in the lanes' consumption order (mstts_pack_skinny_bwd): wave w / iteration it / lane l of
// workgroup (strip, slice) finds its float4 at ((((strip * nsplit + slice) * 4 + w) * nit + it) * 64 + l) * 4 - one contiguous
// 1 KB per wave load.  From the row-major kernel the same load touches 16 rows x 64 B, four cache lines per 4-lane group.
// pair form (the two directions of a BiLSTM step in one launch): blockIdx.z >= nmb selects a second, independent problem of the same
// shape whose operands sit pair_dg / pair_w / pair_p floats behind the first one's
template <int NIT, bool TWO, bool PACKED>
__global__ __launch_bounds__(256) void skinny_bwd_kernel(const float* __restrict__ dG, long ldg, const float* __restrict__ W, long ldw,
                                                         float* __restrict__ P, long pstride, int M, int R, int N, int NL,
                                                         int nmb, long pair_dg, long pair_w, long pair_p) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    if ((int)blockIdx.z >= nmb) { dG += pair_dg; W += pair_w; P += pair_p; }
    const int r0 = blockIdx.x * 32, ns = blockIdx.y, m0 = ((int)blockIdx.z % nmb) * 32;
    const int nb = ns * NL;
    const int lds_ld = NL + 4;                         // row stride == 1 (mod 16) in 16-byte slots: b128 reads spread
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int j = lane & 15, kq = lane >> 4;
    const int rt = wave & 1, half = wave >> 1;
    const int hl = NL / 2;                             // columns per wave, multiple of 16
    constexpr bool EXACT = NIT > 0;
    constexpr int UNROLL = EXACT ? NIT : MAX_IT;
    const int nit = EXACT ? NIT : hl / 16;
    const int c0 = half * hl + 4 * kq;                 // slice-relative first column of this lane
    // 1) dG slice first: dG[m0 .. m0+32, nb .. nb+NL), 8 threads per row, <= 2 rounds of 16 float4 (NL <= 1024);
    //    the first round's loads are issued before the weight loads, its LDS writes after them
    const int nl4 = NL / 4;
    const int sb = threadIdx.x >> 3, sc = threadIdx.x & 7;
    const bool srow_live = m0 + sb < M;
    const float* gr = dG + (long)(m0 + sb) * ldg + nb;
    f32x4 st[16];
#pragma unroll
    for (int q = 0; q < 16; ++q) {
        st[q] = (f32x4){0.f, 0.f, 0.f, 0.f};
        const int c4 = sc + 8 * q;
        if (c4 < nl4 && srow_live) st[q] = *reinterpret_cast<const f32x4*>(gr + c4 * 4);
    }
    // 2) all weight loads: row r0 + 16*rt + j, columns nb + c0 + 16*it .. +3
    const bool row_ok = PACKED || r0 + 16 * rt + j < R;
    const float* wp = PACKED ? W + ((((long)blockIdx.x * gridDim.y + ns) * 4 + wave) * nit) * 256 + lane * 4
                             : W + (long)(r0 + 16 * rt + j) * ldw + nb + c0;
    constexpr int WSTEP = PACKED ? 256 : 16;
    f32x4 wreg[UNROLL];
#pragma unroll
    for (int it = 0; it < UNROLL; ++it) {
        wreg[it] = (f32x4){0.f, 0.f, 0.f, 0.f};
        if ((EXACT || it < nit) && row_ok) wreg[it] = *reinterpret_cast<const f32x4*>(wp + WSTEP * it);
    }
    // 3) LDS writes of the slice (second round only when NL > 512)
#pragma unroll
    for (int q = 0; q < 16; ++q) {
        const int c4 = sc + 8 * q;
        if (c4 < nl4) *reinterpret_cast<f32x4*>(smem + sb * lds_ld + c4 * 4) = st[q];
    }
    if (nl4 > 128) {
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            st[q] = (f32x4){0.f, 0.f, 0.f, 0.f};
            const int c4 = sc + 8 * (q + 16);
            if (c4 < nl4 && srow_live) st[q] = *reinterpret_cast<const f32x4*>(gr + c4 * 4);
        }
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            const int c4 = sc + 8 * (q + 16);
            if (c4 < nl4) *reinterpret_cast<f32x4*>(smem + sb * lds_ld + c4 * 4) = st[q];
        }
    }
    __syncthreads();
    // 3) MFMA chain: two row tiles of dG (rows j and 16+j) against this wave's 16 kernel rows
    f32x4 acc0 = (f32x4){0.f, 0.f, 0.f, 0.f}, acc1 = acc0, acc2 = acc0, acc3 = acc0;
    const float* a0p = smem + j * lds_ld + c0;
    const float* a1p = a0p + 16 * lds_ld;
#pragma unroll
    for (int it = 0; it < UNROLL; ++it) {
        if (EXACT || it < nit) {
            const float4 av0 = *reinterpret_cast<const float4*>(a0p + 16 * it);
            const float4 av1 = TWO ? *reinterpret_cast<const float4*>(a1p + 16 * it) : make_float4(0.f, 0.f, 0.f, 0.f);
            const f32x4 bv = wreg[it];
            acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(av0.x, bv[0], acc0, 0, 0, 0);
            acc2 = __builtin_amdgcn_mfma_f32_16x16x4f32(av0.y, bv[1], acc2, 0, 0, 0);
            acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(av0.z, bv[2], acc0, 0, 0, 0);
            acc2 = __builtin_amdgcn_mfma_f32_16x16x4f32(av0.w, bv[3], acc2, 0, 0, 0);
            if (TWO) {
                acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(av1.x, bv[0], acc1, 0, 0, 0);
                acc3 = __builtin_amdgcn_mfma_f32_16x16x4f32(av1.y, bv[1], acc3, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(av1.z, bv[2], acc1, 0, 0, 0);
                acc3 = __builtin_amdgcn_mfma_f32_16x16x4f32(av1.w, bv[3], acc3, 0, 0, 0);
            }
        }
    }
    // 4) reduce the two column halves, write the 32 x 32 tile
    __syncthreads();
    float* red = smem;                                  // [4 waves][32 rows][17]
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        red[(wave * 32 + kq * 4 + r) * 17 + j] = acc0[r] + acc2[r];
        red[(wave * 32 + 16 + kq * 4 + r) * 17 + j] = acc1[r] + acc3[r];
    }
    __syncthreads();
    float* out = P + (long)ns * pstride;
    for (int i = threadIdx.x; i < 32 * 32; i += 256) {
        const int b = i >> 5, c = i & 31;
        const int t = c >> 4, cj = c & 15;
        if (m0 + b < M && r0 + c < R)
            out[(long)(m0 + b) * R + r0 + c] = red[(t * 32 + b) * 17 + cj] + red[((t + 2) * 32 + b) * 17 + cj];
    }
}

// row-major W[R, N] (row stride ldw) -> consumption order of skinny_bwd_kernel<.., PACKED> for `nsplit` column slices
__global__ void pack_skinny_bwd_kernel(const float* __restrict__ W, long ldw, float* __restrict__ Wp, int R, int N, int nsplit) {
    const int NL = N / nsplit, nit = NL / 32, hl = NL / 2;
    const long n = (long)R * N;
    for (long p = blockIdx.x * (long)blockDim.x + threadIdx.x; p < n; p += (long)gridDim.x * blockDim.x) {
        const int e = (int)(p & 3), lane = (int)((p >> 2) & 63);
        long r = p >> 8;
        const int it = (int)(r % nit); r /= nit;
        const int wave = (int)(r & 3); r >>= 2;
        const int ns = (int)(r % nsplit), strip = (int)(r / nsplit);
        const int j = lane & 15, kq = lane >> 4, rt = wave & 1, half = wave >> 1;
        const int row = strip * 32 + 16 * rt + j, col = ns * NL + half * hl + 4 * kq + 16 * it + e;
        Wp[p] = W[(long)row * ldw + col];
    }
}

static bool g_attr_set = false;
static void set_lds_attr() {
    if (g_attr_set) return;
#define MSTTS_SK_ATTR(N, T)                                                                                                   \
    hipFuncSetAttribute((const void*)skinny_fwd_kernel<N, T>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);       \
    hipFuncSetAttribute((const void*)skinny_bwd_kernel<N, T, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); \
    hipFuncSetAttribute((const void*)skinny_bwd_kernel<N, T, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    MSTTS_SK_ATTR(0, true) MSTTS_SK_ATTR(0, false) MSTTS_SK_ATTR(2, true) MSTTS_SK_ATTR(2, false) MSTTS_SK_ATTR(4, true) MSTTS_SK_ATTR(4, false)
    MSTTS_SK_ATTR(8, true) MSTTS_SK_ATTR(8, false) MSTTS_SK_ATTR(14, true) MSTTS_SK_ATTR(14, false) MSTTS_SK_ATTR(16, true) MSTTS_SK_ATTR(16, false)
    MSTTS_SK_ATTR(28, true) MSTTS_SK_ATTR(28, false) MSTTS_SK_ATTR(32, true) MSTTS_SK_ATTR(32, false)
#undef MSTTS_SK_ATTR
    g_attr_set = true;
}

// pick the guard-free instantiation when the trip count is one of the shapes the model produces
#define SK_FWD(N, T) skinny_fwd_kernel<N, T>
#define SK_BWD(N, T) skinny_bwd_kernel<N, T, false>
#define SK_BWD_PACKED(N, T) skinny_bwd_kernel<N, T, true>
#define MSTTS_SK_DISPATCH(KERNEL, nit, two, ...)                                                       \
    do {                                                                                               \
        if (two) {                                                                                     \
            if (nit == 32) hipLaunchKernelGGL((KERNEL(32, true)), __VA_ARGS__);                        \
            else if (nit == 28) hipLaunchKernelGGL((KERNEL(28, true)), __VA_ARGS__);                   \
            else if (nit == 16) hipLaunchKernelGGL((KERNEL(16, true)), __VA_ARGS__);                   \
            else if (nit == 14) hipLaunchKernelGGL((KERNEL(14, true)), __VA_ARGS__);                   \
            else if (nit == 8) hipLaunchKernelGGL((KERNEL(8, true)), __VA_ARGS__);                     \
            else if (nit == 4) hipLaunchKernelGGL((KERNEL(4, true)), __VA_ARGS__);                     \
            else if (nit == 2) hipLaunchKernelGGL((KERNEL(2, true)), __VA_ARGS__);                     \
            else hipLaunchKernelGGL((KERNEL(0, true)), __VA_ARGS__);                                   \
        } else {                                                                                       \
            if (nit == 32) hipLaunchKernelGGL((KERNEL(32, false)), __VA_ARGS__);                       \
            else if (nit == 28) hipLaunchKernelGGL((KERNEL(28, false)), __VA_ARGS__);                  \
            else if (nit == 16) hipLaunchKernelGGL((KERNEL(16, false)), __VA_ARGS__);                  \
            else if (nit == 14) hipLaunchKernelGGL((KERNEL(14, false)), __VA_ARGS__);                  \
            else if (nit == 8) hipLaunchKernelGGL((KERNEL(8, false)), __VA_ARGS__);                    \
            else if (nit == 4) hipLaunchKernelGGL((KERNEL(4, false)), __VA_ARGS__);                    \
            else if (nit == 2) hipLaunchKernelGGL((KERNEL(2, false)), __VA_ARGS__);                    \
            else hipLaunchKernelGGL((KERNEL(0, false)), __VA_ARGS__);                                  \
        }                                                                                              \
    } while (0)

}  // namespace mstts
using namespace mstts;

// workgroups a skinny launch aims for: two per CU (one computes while the other waits on its loads)
static long target_wgs() { return 512; }

extern "C" int32_t mstts_skinny_fwd_splits(int64_t N, int64_t K) {
    // K-splits so that strips * splits ~ 256 workgroups; each slice a multiple of 32 rows, <= 512 rows
    if (N <= 0 || K <= 0 || K % 32 != 0) return 0;
    const long strips = (N + 63) / 64;
    long ks = target_wgs() / strips;            // default: two workgroups per CU (one computes while the other waits on loads)
    if (ks < 1) ks = 1;
    if (ks > K / 32) ks = K / 32;
    if (ks > 16) ks = 16;                       // consumers sum at most 16 slabs (MSTTS_MAX_PARTS)
    while (ks > 1 && (K % (ks * 32) != 0)) --ks;
    while (K / ks > 512) {                      // slice too long for the register-resident weight loads
        ++ks;
        while (ks < K / 32 && (K % (ks * 32) != 0)) ++ks;
        if (K % (ks * 32) != 0) return 0;
    }
    return (int32_t)ks;
}

extern "C" int mstts_skinny_fwd(const float* X, int64_t ldx, const float* W, int64_t ldw, float* P, int64_t pstride, int64_t M, int64_t N,
                                int64_t K, int32_t ksplit, mstts_stream_t s) {
    MSTTS_REQUIRE(X && W && P && M >= 1 && N >= 1, MSTTS_ERR_SHAPE, "skinny_fwd: bad arguments");
    MSTTS_REQUIRE(ksplit >= 1 && K % (ksplit * 32L) == 0 && K / ksplit <= 512, MSTTS_ERR_SHAPE,
                  "skinny_fwd: K must be a multiple of 32*ksplit with slices of at most 512 rows");
    MSTTS_REQUIRE(N % 4 == 0 && ldw % 4 == 0 && ldx % 4 == 0 && aligned16(X) && aligned16(W), MSTTS_ERR_ALIGN,
                  "skinny_fwd: float4 alignment (N, ldx, ldw multiples of 4; 16-byte aligned pointers)");
    const int KL = (int)(K / ksplit);
    size_t lds = sizeof(float) * (size_t)32 * (KL + 4);
    const size_t red = sizeof(float) * 4 * 32 * 65;
    if (lds < red) lds = red;
    dim3 grid((unsigned)((N + 63) / 64), (unsigned)ksplit, (unsigned)((M + 31) / 32));
    set_lds_attr();
    const int nit = KL / 16;
    const bool two = M > 16;      // blocks of 32 rows; with M <= 16 the second 16-row MFMA tile is all padding
    MSTTS_SK_DISPATCH(SK_FWD, nit, two, grid, dim3(256), lds, (hipStream_t)s, X, (long)ldx, W, (long)ldw, P,
                      (long)(pstride > 0 ? pstride : M * N), (int)M, (int)N, (int)K, KL);
    MSTTS_CHECK_LAUNCH("skinny_fwd");
    return MSTTS_OK;
}

extern "C" int32_t mstts_skinny_bwd_splits(int64_t R, int64_t N) {
    // N-splits so that strips * splits ~ 256 workgroups; slices are multiples of 32 columns,
    // at least 128 (when N allows) and at most 1024 columns
    if (R <= 0 || N <= 0 || N % 32 != 0) return 0;
    const long strips = (R + 31) / 32;
    long ns = target_wgs() / strips;
    if (ns < 1) ns = 1;
    const long max_ns = N >= 128 ? N / 128 : 1;
    if (ns > max_ns) ns = max_ns;
    if (ns > 8) ns = 8;                         // consumers sum at most 8 slabs
    while (ns > 1 && (N % (ns * 32) != 0)) --ns;
    while (N / ns > 1024) {
        ++ns;
        while (ns < N / 32 && (N % (ns * 32) != 0)) ++ns;
        if (N % (ns * 32) != 0) return 0;
    }
    return (int32_t)ns;
}

extern "C" int mstts_skinny_bwd(const float* dG, int64_t ldg, const float* W, int64_t ldw, float* P, int64_t pstride, int64_t M, int64_t R,
                                int64_t N, int32_t nsplit, mstts_stream_t s) {
    MSTTS_REQUIRE(dG && W && P && M >= 1 && R >= 1, MSTTS_ERR_SHAPE, "skinny_bwd: bad arguments");
    MSTTS_REQUIRE(nsplit >= 1 && N % (nsplit * 32L) == 0 && N / nsplit <= 1024, MSTTS_ERR_SHAPE,
                  "skinny_bwd: N must be a multiple of 32*nsplit with slices of at most 1024 columns");
    MSTTS_REQUIRE(ldw % 4 == 0 && ldg % 4 == 0 && aligned16(dG) && aligned16(W), MSTTS_ERR_ALIGN, "skinny_bwd: float4 alignment");
    const int NL = (int)(N / nsplit);
    size_t lds = sizeof(float) * (size_t)32 * (NL + 4);
    const size_t red = sizeof(float) * 4 * 32 * 17;
    if (lds < red) lds = red;
    dim3 grid((unsigned)((R + 31) / 32), (unsigned)nsplit, (unsigned)((M + 31) / 32));
    set_lds_attr();
    const int nit = NL / 32;
    const bool two = M > 16;
    MSTTS_SK_DISPATCH(SK_BWD, nit, two, grid, dim3(256), lds, (hipStream_t)s, dG, (long)ldg, W, (long)ldw, P,
                      (long)(pstride > 0 ? pstride : M * R), (int)M, (int)R, (int)N, NL, (int)grid.z, 0L, 0L, 0L);
    MSTTS_CHECK_LAUNCH("skinny_bwd");
    return MSTTS_OK;
}

extern "C" int mstts_pack_skinny_bwd(const float* W, int64_t ldw, float* Wp, int64_t R, int64_t N, int32_t nsplit, mstts_stream_t s) {
    MSTTS_REQUIRE(W && Wp && R >= 32 && R % 32 == 0 && nsplit >= 1 && N % (nsplit * 32L) == 0 && N / nsplit <= 1024 && R * N < (1LL << 31), MSTTS_ERR_SHAPE,
                  "pack_skinny_bwd: R %% 32 == 0 and N a multiple of 32*nsplit with slices of at most 1024 columns required");
    const long n = R * N;
    hipLaunchKernelGGL(pack_skinny_bwd_kernel, dim3((unsigned)((n + 255) / 256 < 4096 ? (n + 255) / 256 : 4096)), dim3(256), 0, (hipStream_t)s, W, (long)ldw, Wp,
                       (int)R, (int)N, (int)nsplit);
    MSTTS_CHECK_LAUNCH("pack_skinny_bwd");
    return MSTTS_OK;
}

extern "C" int mstts_skinny_bwd_packed(const float* dG, int64_t ldg, const float* Wp, float* P, int64_t pstride, int64_t M, int64_t R,
                                       int64_t N, int32_t nsplit, mstts_stream_t s) {
    MSTTS_REQUIRE(dG && Wp && P && M >= 1 && R >= 32 && R % 32 == 0, MSTTS_ERR_SHAPE, "skinny_bwd_packed: bad arguments (R %% 32 == 0)");
    MSTTS_REQUIRE(nsplit >= 1 && N % (nsplit * 32L) == 0 && N / nsplit <= 1024, MSTTS_ERR_SHAPE,
                  "skinny_bwd_packed: N must be a multiple of 32*nsplit with slices of at most 1024 columns");
    MSTTS_REQUIRE(ldg % 4 == 0 && aligned16(dG) && aligned16(Wp), MSTTS_ERR_ALIGN, "skinny_bwd_packed: float4 alignment");
    const int NL = (int)(N / nsplit);
    size_t lds = sizeof(float) * (size_t)32 * (NL + 4);
    const size_t red = sizeof(float) * 4 * 32 * 17;
    if (lds < red) lds = red;
    dim3 grid((unsigned)(R / 32), (unsigned)nsplit, (unsigned)((M + 31) / 32));
    set_lds_attr();
    const int nit = NL / 32;
    const bool two = M > 16;
    MSTTS_SK_DISPATCH(SK_BWD_PACKED, nit, two, grid, dim3(256), lds, (hipStream_t)s, dG, (long)ldg, Wp, 0L, P,
                      (long)(pstride > 0 ? pstride : M * R), (int)M, (int)R, (int)N, NL, (int)grid.z, 0L, 0L, 0L);
    MSTTS_CHECK_LAUNCH("skinny_bwd_packed");
    return MSTTS_OK;
}

/* two products of identical shape in one launch (the two directions of a BiLSTM backward step): problem 2's operands are given by
 * their own pointers; both must satisfy mstts_skinny_bwd's requirements */
extern "C" int mstts_skinny_bwd_pair(const float* dG, const float* dG2, int64_t ldg, const float* W, const float* W2, int64_t ldw, float* P, float* P2,
                                     int64_t pstride, int64_t M, int64_t R, int64_t N, int32_t nsplit, mstts_stream_t s) {
    MSTTS_REQUIRE(dG && W && P && dG2 && W2 && P2 && M >= 1 && R >= 1, MSTTS_ERR_SHAPE, "skinny_bwd_pair: bad arguments");
    MSTTS_REQUIRE(nsplit >= 1 && N % (nsplit * 32L) == 0 && N / nsplit <= 1024, MSTTS_ERR_SHAPE,
                  "skinny_bwd_pair: N must be a multiple of 32*nsplit with slices of at most 1024 columns");
    MSTTS_REQUIRE(ldw % 4 == 0 && ldg % 4 == 0 && aligned16(dG) && aligned16(W) && aligned16(dG2) && aligned16(W2), MSTTS_ERR_ALIGN, "skinny_bwd_pair: float4 alignment");
    const int NL = (int)(N / nsplit);
    size_t lds = sizeof(float) * (size_t)32 * (NL + 4);
    const size_t red = sizeof(float) * 4 * 32 * 17;
    if (lds < red) lds = red;
    const unsigned nmb = (unsigned)((M + 31) / 32);
    dim3 grid((unsigned)((R + 31) / 32), (unsigned)nsplit, 2 * nmb);
    set_lds_attr();
    const int nit = NL / 32;
    const bool two = M > 16;
    MSTTS_SK_DISPATCH(SK_BWD, nit, two, grid, dim3(256), lds, (hipStream_t)s, dG, (long)ldg, W, (long)ldw, P,
                      (long)(pstride > 0 ? pstride : M * R), (int)M, (int)R, (int)N, NL, (int)nmb, (long)(dG2 - dG), (long)(W2 - W), (long)(P2 - P));
    MSTTS_CHECK_LAUNCH("skinny_bwd_pair");
    return MSTTS_OK;
}
