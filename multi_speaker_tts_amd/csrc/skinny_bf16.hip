// bf16 variants of the skinny weight-streaming products (BASELINE config 3: "bf16 with fp32 master"): the recurrent kernels are
// kept as packed bf16 copies of the fp32 master weights, the [<=32, K] activation block is rounded to bf16 while it is staged
// into LDS, products accumulate in fp32 (v_mfma_f32_16x16x16_bf16), partial slabs stay fp32.  Half the weight bytes and a
// negligible MFMA phase against the fp32 kernels of skinny.hip.  Rounding is round-to-nearest-even, the same as
// torch.Tensor.to(torch.bfloat16), so the oracle can emulate the mode exactly (bf(X) . bf(W) in high precision).
//
// Because the bf16 copies are derived data, they are stored in exactly the order the lanes consume them:
//   forward  Wp[strip][ks][wave][kstep][lane][tile 0..3][4] : W[kb + wave*wl + 16*kstep + 4*(lane>>4) + e][64*strip + 16*tile + (lane&15)]
//   backward Wq[rstrip][ns][wave][pair][lane][2][4]         : W[32*rstrip + 16*(wave&1) + (lane&15)][nb + (wave>>1)*hl + 16*(2*pair+s2) + 4*(lane>>4) + e]
// so every wave-level load is one contiguous 2 KB / 1 KB stream.
#include "common.h"

namespace mstts {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef short bf16x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ unsigned bf16_rne(float f) {
    unsigned u = __float_as_uint(f);
    u += 0x7FFFu + ((u >> 16) & 1u);
    return u >> 16;
}
__device__ __forceinline__ u32x2 pack4(f32x4 v) {
    return (u32x2){bf16_rne(v[0]) | (bf16_rne(v[1]) << 16), bf16_rne(v[2]) | (bf16_rne(v[3]) << 16)};
}
__device__ __forceinline__ bf16x4 as_bf4(unsigned lo, unsigned hi) {
    union { u32x2 u; bf16x4 b; } c;
    c.u = (u32x2){lo, hi};
    return c.b;
}

constexpr int BF_MAX_KS = 8;        // k-steps (16 rows) per wave: KL = 64 * nks <= 512

// ---- forward: P[ks][M][N] = bf(X[M, K-slice]) . bf(W[K-slice, N]) ----------------------------------------------------------
template <int NKS, bool TWO>
__global__ __launch_bounds__(256) void skinny_fwd_bf16_kernel(const float* __restrict__ X, long ldx, const u32x4* __restrict__ Wp,
                                                              float* __restrict__ P, long pstride, int M, int N, int KL) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    unsigned short* sx = reinterpret_cast<unsigned short*>(smem_raw);
    const int strip = blockIdx.x, ks = blockIdx.y, KS = gridDim.y, m0 = blockIdx.z * 32;
    const int n0 = strip * 64, kb = ks * KL;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int j = lane & 15, kq = lane >> 4;
    constexpr bool EXACT = NKS > 0;
    constexpr int UN = EXACT ? NKS : BF_MAX_KS;
    const int nks = EXACT ? NKS : KL / 64;
    const int wl = KL / 4, ld = KL + 8;                           // LDS row stride in bf16 elements
    // 1) activation slice, fp32
    const int kl4 = KL / 4;
    const int sb = threadIdx.x >> 3, sc = threadIdx.x & 7;
    const bool live = m0 + sb < M;
    const float* xr = X + (long)(m0 + sb) * ldx + kb;
    f32x4 st[16];
#pragma unroll
    for (int q = 0; q < 16; ++q) {
        st[q] = (f32x4){0.f, 0.f, 0.f, 0.f};
        const int k4 = sc + 8 * q;
        if (k4 < kl4 && live) st[q] = *reinterpret_cast<const f32x4*>(xr + k4 * 4);
    }
    // 2) this wave's packed weight fragments: 32 B per lane per k-step
    const u32x4* wp = Wp + ((((long)strip * KS + ks) * 4 + wave) * nks * 64 + lane) * 2;
    u32x4 wlo[UN], whi[UN];
#pragma unroll
    for (int s = 0; s < UN; ++s) {
        wlo[s] = (u32x4){0u, 0u, 0u, 0u}; whi[s] = wlo[s];
        if (EXACT || s < nks) { wlo[s] = wp[(long)s * 128]; whi[s] = wp[(long)s * 128 + 1]; }
    }
    // 3) round the slice to bf16 into LDS
#pragma unroll
    for (int q = 0; q < 16; ++q) {
        const int k4 = sc + 8 * q;
        if (k4 < kl4) *reinterpret_cast<u32x2*>(sx + sb * ld + k4 * 4) = pack4(st[q]);
    }
    __syncthreads();
    // 4) MFMA: 2 row tiles x 4 column tiles
    f32x4 acc[2][4];
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int m = 0; m < 4; ++m) acc[t][m] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const unsigned short* xa = sx + j * ld + wave * wl + 4 * kq;
#pragma unroll
    for (int s = 0; s < UN; ++s) {
        if (EXACT || s < nks) {
            const u32x2 a0u = *reinterpret_cast<const u32x2*>(xa + 16 * s);
            const bf16x4 a0 = as_bf4(a0u[0], a0u[1]);
            const bf16x4 b0 = as_bf4(wlo[s][0], wlo[s][1]), b1 = as_bf4(wlo[s][2], wlo[s][3]);
            const bf16x4 b2 = as_bf4(whi[s][0], whi[s][1]), b3 = as_bf4(whi[s][2], whi[s][3]);
            acc[0][0] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a0, b0, acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a0, b1, acc[0][1], 0, 0, 0);
            acc[0][2] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a0, b2, acc[0][2], 0, 0, 0);
            acc[0][3] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a0, b3, acc[0][3], 0, 0, 0);
            if (TWO) {
                const u32x2 a1u = *reinterpret_cast<const u32x2*>(xa + 16 * ld + 16 * s);
                const bf16x4 a1 = as_bf4(a1u[0], a1u[1]);
                acc[1][0] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a1, b0, acc[1][0], 0, 0, 0);
                acc[1][1] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a1, b1, acc[1][1], 0, 0, 0);
                acc[1][2] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a1, b2, acc[1][2], 0, 0, 0);
                acc[1][3] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a1, b3, acc[1][3], 0, 0, 0);
            }
        }
    }
    // 5) cross-wave reduction, fp32 partial slab
    __syncthreads();
    float* red = reinterpret_cast<float*>(smem_raw);             // [4][32][65]
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
            for (int r = 0; r < 4; ++r)
                red[(wave * 32 + 16 * t + kq * 4 + r) * 65 + 16 * m + j] = acc[t][m][r];
    __syncthreads();
    float* out = P + (long)ks * pstride;
    for (int i = threadIdx.x; i < 32 * 64; i += 256) {
        const int b = i >> 6, c = i & 63;
        if (m0 + b < M) out[(long)(m0 + b) * N + n0 + c] = red[b * 65 + c] + red[(32 + b) * 65 + c] + red[(64 + b) * 65 + c] + red[(96 + b) * 65 + c];
    }
}

// ---- backward: P[ns][M][R] = bf(dG[M, N-slice]) . bf(W[R, N-slice])^T -------------------------------------------------------
constexpr int BF_MAX_PAIRS = 16;    // hl = 32 * pairs <= 512
template <int NP, bool TWO>
__global__ __launch_bounds__(256) void skinny_bwd_bf16_kernel(const float* __restrict__ dG, long ldg, const u32x4* __restrict__ Wq,
                                                              float* __restrict__ P, long pstride, int M, int R, int NL) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    unsigned short* sg = reinterpret_cast<unsigned short*>(smem_raw);
    const int rstrip = blockIdx.x, ns = blockIdx.y, NS = gridDim.y, m0 = blockIdx.z * 32;
    const int r0 = rstrip * 32, nb = ns * NL;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int j = lane & 15, kq = lane >> 4;
    const int rt = wave & 1, half = wave >> 1;
    const int hl = NL / 2, ld = NL + 8;
    constexpr bool EXACT = NP > 0;
    constexpr int UN = EXACT ? NP : BF_MAX_PAIRS;
    const int npairs = EXACT ? NP : hl / 32;
    // 1) dG slice (fp32), two rounds of 16 float4 per thread cover NL <= 1024
    const int nl4 = NL / 4;
    const int sb = threadIdx.x >> 3, sc = threadIdx.x & 7;
    const bool live = m0 + sb < M;
    const float* gr = dG + (long)(m0 + sb) * ldg + nb;
    f32x4 st[16];
#pragma unroll
    for (int q = 0; q < 16; ++q) {
        st[q] = (f32x4){0.f, 0.f, 0.f, 0.f};
        const int c4 = sc + 8 * q;
        if (c4 < nl4 && live) st[q] = *reinterpret_cast<const f32x4*>(gr + c4 * 4);
    }
    // 2) packed weight fragments: 16 B per lane per pair of k-steps
    const u32x4* wq = Wq + ((((long)rstrip * NS + ns) * 4 + wave) * npairs) * 64 + lane;
    u32x4 wreg[UN];
#pragma unroll
    for (int p = 0; p < UN; ++p) {
        wreg[p] = (u32x4){0u, 0u, 0u, 0u};
        if (EXACT || p < npairs) wreg[p] = wq[(long)p * 64];
    }
#pragma unroll
    for (int q = 0; q < 16; ++q) {
        const int c4 = sc + 8 * q;
        if (c4 < nl4) *reinterpret_cast<u32x2*>(sg + sb * ld + c4 * 4) = pack4(st[q]);
    }
    if (nl4 > 128) {
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            st[q] = (f32x4){0.f, 0.f, 0.f, 0.f};
            const int c4 = sc + 8 * (q + 16);
            if (c4 < nl4 && live) st[q] = *reinterpret_cast<const f32x4*>(gr + c4 * 4);
        }
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            const int c4 = sc + 8 * (q + 16);
            if (c4 < nl4) *reinterpret_cast<u32x2*>(sg + sb * ld + c4 * 4) = pack4(st[q]);
        }
    }
    __syncthreads();
    f32x4 acc0 = (f32x4){0.f, 0.f, 0.f, 0.f}, acc1 = acc0, acc2 = acc0, acc3 = acc0;
    const unsigned short* a0p = sg + j * ld + half * hl + 4 * kq;
#pragma unroll
    for (int p = 0; p < UN; ++p) {
        if (EXACT || p < npairs) {
            const bf16x4 b0 = as_bf4(wreg[p][0], wreg[p][1]), b1 = as_bf4(wreg[p][2], wreg[p][3]);
            const u32x2 x0 = *reinterpret_cast<const u32x2*>(a0p + 32 * p), x1 = *reinterpret_cast<const u32x2*>(a0p + 32 * p + 16);
            acc0 = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(as_bf4(x0[0], x0[1]), b0, acc0, 0, 0, 0);
            acc2 = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(as_bf4(x1[0], x1[1]), b1, acc2, 0, 0, 0);
            if (TWO) {
                const u32x2 y0 = *reinterpret_cast<const u32x2*>(a0p + 16 * ld + 32 * p), y1 = *reinterpret_cast<const u32x2*>(a0p + 16 * ld + 32 * p + 16);
                acc1 = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(as_bf4(y0[0], y0[1]), b0, acc1, 0, 0, 0);
                acc3 = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(as_bf4(y1[0], y1[1]), b1, acc3, 0, 0, 0);
            }
        }
    }
    __syncthreads();
    float* red = reinterpret_cast<float*>(smem_raw);             // [4 waves][32 rows][17]
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
        if (m0 + b < M) out[(long)(m0 + b) * R + r0 + c] = red[(t * 32 + b) * 17 + cj] + red[((t + 2) * 32 + b) * 17 + cj];
    }
}

__global__ void pack_bf16_fwd_kernel(const float* __restrict__ W, long ldw, unsigned short* __restrict__ Wp, long K, long N, int KS) {
    const long total = K * N, KL = K / KS, wl = KL / 4, nks = KL / 64;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        long r = i;
        const int e = r & 3; r >>= 2;
        const int t = r & 3; r >>= 2;
        const int l = r & 63; r >>= 6;
        const long s = r % nks; r /= nks;
        const int wave = r & 3; r >>= 2;
        const long ks = r % KS, strip = r / KS;
        const long k = ks * KL + wave * wl + 16 * s + 4 * (l >> 4) + e, n = 64 * strip + 16 * t + (l & 15);
        Wp[i] = (unsigned short)bf16_rne(W[k * ldw + n]);
    }
}
__global__ void pack_bf16_bwd_kernel(const float* __restrict__ W, long ldw, unsigned short* __restrict__ Wq, long R, long N, int NS) {
    const long total = R * N, NL = N / NS, hl = NL / 2, npairs = hl / 32;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        long r = i;
        const int e = r & 3; r >>= 2;
        const int s2 = r & 1; r >>= 1;
        const int l = r & 63; r >>= 6;
        const long p = r % npairs; r /= npairs;
        const int wave = r & 3; r >>= 2;
        const long ns = r % NS, rstrip = r / NS;
        const long row = 32 * rstrip + 16 * (wave & 1) + (l & 15);
        const long n = ns * NL + (wave >> 1) * hl + 16 * (2 * p + s2) + 4 * (l >> 4) + e;
        Wq[i] = (unsigned short)bf16_rne(W[row * ldw + n]);
    }
}

static bool g_bf_attr = false;
static void bf_attr() {
    if (g_bf_attr) return;
#define A_(K) hipFuncSetAttribute((const void*)K, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    A_((skinny_fwd_bf16_kernel<0, true>)) A_((skinny_fwd_bf16_kernel<0, false>)) A_((skinny_fwd_bf16_kernel<4, true>)) A_((skinny_fwd_bf16_kernel<4, false>))
    A_((skinny_fwd_bf16_kernel<1, true>)) A_((skinny_fwd_bf16_kernel<1, false>))
    A_((skinny_bwd_bf16_kernel<0, true>)) A_((skinny_bwd_bf16_kernel<0, false>)) A_((skinny_bwd_bf16_kernel<8, true>)) A_((skinny_bwd_bf16_kernel<8, false>))
    A_((skinny_bwd_bf16_kernel<1, true>)) A_((skinny_bwd_bf16_kernel<1, false>))
#undef A_
    g_bf_attr = true;
}

}  // namespace mstts
using namespace mstts;

extern "C" int32_t mstts_skinny_bf16_fwd_splits(int64_t N, int64_t K) {
    if (N <= 0 || K <= 0 || N % 64 != 0 || K % 64 != 0) return 0;
    const long strips = N / 64, units = K / 64;
    long best = 0, best_d = 1L << 40;
    for (long ks = 1; ks <= 16 && ks <= units; ++ks) {
        if (units % ks != 0 || K / ks > 64 * BF_MAX_KS) continue;
        const long d = labs(strips * ks - 512);
        if (d < best_d) { best_d = d; best = ks; }
    }
    return (int32_t)best;
}
extern "C" int32_t mstts_skinny_bf16_bwd_splits(int64_t R, int64_t N) {
    if (R <= 0 || N <= 0 || R % 32 != 0 || N % 64 != 0) return 0;
    const long strips = R / 32, units = N / 64;
    long best = 0, best_d = 1L << 40;
    for (long ns = 1; ns <= 8 && ns <= units; ++ns) {
        if (units % ns != 0 || N / ns > 64 * BF_MAX_PAIRS) continue;
        const long d = labs(strips * ns - 512);
        if (d < best_d) { best_d = d; best = ns; }
    }
    return (int32_t)best;
}
static unsigned pk_grid(long n) { long b = (n + 255) / 256; if (b > 16384) b = 16384; return (unsigned)(b < 1 ? 1 : b); }

extern "C" int mstts_pack_bf16_fwd(const float* W, int64_t ldw, void* Wp, int64_t K, int64_t N, int32_t ksplit, mstts_stream_t s) {
    MSTTS_REQUIRE(W && Wp && ksplit >= 1 && N % 64 == 0 && K % (64L * ksplit) == 0 && K / ksplit <= 64 * BF_MAX_KS, MSTTS_ERR_SHAPE,
                  "pack_bf16_fwd: N %% 64, K %% (64*ksplit) and K/ksplit <= 512 required");
    hipLaunchKernelGGL(pack_bf16_fwd_kernel, dim3(pk_grid(K * N)), dim3(256), 0, (hipStream_t)s, W, (long)ldw, (unsigned short*)Wp, (long)K, (long)N, (int)ksplit);
    MSTTS_CHECK_LAUNCH("pack_bf16_fwd");
    return MSTTS_OK;
}
extern "C" int mstts_pack_bf16_bwd(const float* W, int64_t ldw, void* Wq, int64_t R, int64_t N, int32_t nsplit, mstts_stream_t s) {
    MSTTS_REQUIRE(W && Wq && nsplit >= 1 && R % 32 == 0 && N % (64L * nsplit) == 0 && N / nsplit <= 64 * BF_MAX_PAIRS, MSTTS_ERR_SHAPE,
                  "pack_bf16_bwd: R %% 32, N %% (64*nsplit) and N/nsplit <= 1024 required");
    hipLaunchKernelGGL(pack_bf16_bwd_kernel, dim3(pk_grid(R * N)), dim3(256), 0, (hipStream_t)s, W, (long)ldw, (unsigned short*)Wq, (long)R, (long)N, (int)nsplit);
    MSTTS_CHECK_LAUNCH("pack_bf16_bwd");
    return MSTTS_OK;
}

extern "C" int mstts_skinny_fwd_bf16(const float* X, int64_t ldx, const void* Wp, float* P, int64_t pstride, int64_t M, int64_t N, int64_t K,
                                     int32_t ksplit, mstts_stream_t s) {
    MSTTS_REQUIRE(X && Wp && P && M >= 1 && ksplit >= 1 && N % 64 == 0 && K % (64L * ksplit) == 0 && K / ksplit <= 64 * BF_MAX_KS, MSTTS_ERR_SHAPE,
                  "skinny_fwd_bf16: N %% 64, K %% (64*ksplit), K/ksplit <= 512 required");
    MSTTS_REQUIRE(ldx % 4 == 0 && aligned16(X) && aligned16(Wp), MSTTS_ERR_ALIGN, "skinny_fwd_bf16: 16-byte alignment");
    bf_attr();
    const int KL = (int)(K / ksplit), nks = KL / 64;
    size_t lds = (size_t)32 * (KL + 8) * 2;
    if (lds < sizeof(float) * 4 * 32 * 65) lds = sizeof(float) * 4 * 32 * 65;
    dim3 grid((unsigned)(N / 64), (unsigned)ksplit, (unsigned)((M + 31) / 32));
    const long ps = pstride ? pstride : M * N;
    const bool two = M > 16;
#define L_(NK, T) hipLaunchKernelGGL((skinny_fwd_bf16_kernel<NK, T>), grid, dim3(256), lds, (hipStream_t)s, X, (long)ldx, (const u32x4*)Wp, P, ps, (int)M, (int)N, KL)
    if (nks == 4) { if (two) L_(4, true); else L_(4, false); }
    else if (nks == 1) { if (two) L_(1, true); else L_(1, false); }
    else { if (two) L_(0, true); else L_(0, false); }
#undef L_
    MSTTS_CHECK_LAUNCH("skinny_fwd_bf16");
    return MSTTS_OK;
}
extern "C" int mstts_skinny_bwd_bf16(const float* dG, int64_t ldg, const void* Wq, float* P, int64_t pstride, int64_t M, int64_t R, int64_t N,
                                     int32_t nsplit, mstts_stream_t s) {
    MSTTS_REQUIRE(dG && Wq && P && M >= 1 && nsplit >= 1 && R % 32 == 0 && N % (64L * nsplit) == 0 && N / nsplit <= 64 * BF_MAX_PAIRS, MSTTS_ERR_SHAPE,
                  "skinny_bwd_bf16: R %% 32, N %% (64*nsplit), N/nsplit <= 1024 required");
    MSTTS_REQUIRE(ldg % 4 == 0 && aligned16(dG) && aligned16(Wq), MSTTS_ERR_ALIGN, "skinny_bwd_bf16: 16-byte alignment");
    bf_attr();
    const int NL = (int)(N / nsplit), np = NL / 64;
    size_t lds = (size_t)32 * (NL + 8) * 2;
    if (lds < sizeof(float) * 4 * 32 * 17) lds = sizeof(float) * 4 * 32 * 17;
    dim3 grid((unsigned)(R / 32), (unsigned)nsplit, (unsigned)((M + 31) / 32));
    const long ps = pstride ? pstride : M * R;
    const bool two = M > 16;
#define L_(NP_, T) hipLaunchKernelGGL((skinny_bwd_bf16_kernel<NP_, T>), grid, dim3(256), lds, (hipStream_t)s, dG, (long)ldg, (const u32x4*)Wq, P, ps, (int)M, (int)R, NL)
    if (np == 8) { if (two) L_(8, true); else L_(8, false); }
    else if (np == 1) { if (two) L_(1, true); else L_(1, false); }
    else { if (two) L_(0, true); else L_(0, false); }
#undef L_
    MSTTS_CHECK_LAUNCH("skinny_bwd_bf16");
    return MSTTS_OK;
}
