// bf16-operand GEMM on the gfx950 matrix cores (v_mfma_f32_32x32x16_bf16, fp32 accumulate) - BASELINE config 3
// ("bf16 with fp32 master"): the SAME contractions mstts_gemm_f32 covers (dense, conv1d 'same' as implicit im2col, data and
// weight gradients, split-K, batches), same descriptor, but both operands are rounded to bf16 (round-to-nearest-even) on their
// way into LDS.  Inputs and outputs stay fp32 in memory - master weights, activations, gradients and the accumulators never
// leave fp32; only the multiplicands are 8-bit-mantissa.  16x the fp32 MFMA rate, so these launches become memory/LDS-bound.
//
// Tile 128 x 128 x 64 per 256-thread workgroup (4 wave64 as 2 x 2, each 64 x 64 = 2 x 2 MFMA tiles).  LDS image of an operand:
// [128 rows][64 k] bf16, row stride 144 B, i.e. k CONTIGUOUS per row whatever the operand's memory layout - an MFMA fragment
// (lane l: row l & 31, k = 8 (l >> 5) .. + 7) is then one conflict-free ds_read_b128.  The transposition this needs for operands
// that are contiguous along their M/N index is done in registers: a thread loads a 4 (k) x 4 (rows) block as four float4 and
// writes four 8-byte k-runs.  The next K-tile is prefetched into registers while the current one is multiplied.
#include "common.h"
#include <type_traits>

namespace mstts {

typedef float gb_f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 gb_bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 gb_bf16x4 __attribute__((ext_vector_type(4)));

constexpr int GB_BK = 64, GB_BM = 128, GB_BN = 128;
constexpr int GB_LD = 72;                 // LDS row stride in bf16 elements (144 B = 36 words, 36 mod 32 = 4): 16-byte aligned rows, b128 reads conflict-free

struct GemmBfArgs {
    const float* A; const float* B; float* C; const float* bias;
    int M, N, K;
    long lda, ldb, ldc;
    int win_T, win_C, win_pad, win_dil;
    int act, accumulate, split_k;
    long stride_a, stride_b, stride_c;
    float alpha;
    int k_per_split;
};

__device__ __forceinline__ float gb_act(float v, int act) {
    if (act == MSTTS_ACT_RELU) return fmaxf(v, 0.f);
    if (act == MSTTS_ACT_TANH) return tanhf_(v);
    if (act == MSTTS_ACT_SIGMOID) return sigmoidf_(v);
    return v;
}
__device__ __forceinline__ gb_bf16x4 gb_round4(float a, float b, float c, float d) {
    return (gb_bf16x4){(__bf16)a, (__bf16)b, (__bf16)c, (__bf16)d};
}

// operand contiguous along k in memory (A row-major, or B given as [N,K]): thread (k4 = tid & 15, r = tid >> 4) takes the float4
// k-run k4 of rows r, r + 16, .., r + 112.  Window arithmetic is hoisted like in gemm.hip: prepare() once, then load() for
// k0, k0 + GB_BK, ... (a thread's rows are fixed, its k advances by GB_BK per call).
template <bool VEC>
struct GbLoaderKC {
    float4 reg[8];
    int t_row[8], tap, kc;
    __device__ __forceinline__ void prepare(int row0, int k0, int wT, int wC) {
        if (wT > 0) {
            const int k4 = threadIdx.x & 15, r = threadIdx.x >> 4;
#pragma unroll
            for (int i = 0; i < 8; ++i) t_row[i] = (row0 + r + i * 16) % wT;
            const int k = k0 + k4 * 4;
            tap = k / wC; kc = k - tap * wC;
        }
    }
    __device__ __forceinline__ void load(const float* __restrict__ base, long ld, int row0, int k0, int rows, int kmax,
                                         int wT, int wC, int wpad, int wdil) {
        const int k4 = threadIdx.x & 15, r = threadIdx.x >> 4;
        const int k = k0 + k4 * 4;
        const int sh = wT > 0 ? (tap - wpad) * wdil : 0;     // tap j = k / C of a dilated 'same' conv: source row = row + (j - pad) * dil
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int row = row0 + r + i * 16;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            bool ok = row < rows && k < kmax;
            long off = (long)row * ld + k;
            if (wT > 0) {
                const int t = t_row[i] + sh;
                ok = ok && t >= 0 && t < wT;
                off = ((long)row + sh) * ld + kc;
            }
            if (ok) {
                if (VEC) v = *reinterpret_cast<const float4*>(base + off);
                else {
                    v.x = base[off];
                    if (k + 1 < kmax) v.y = base[off + 1];
                    if (k + 2 < kmax) v.z = base[off + 2];
                    if (k + 3 < kmax) v.w = base[off + 3];
                }
            }
            reg[i] = v;
        }
        if (wT > 0) {
            kc += GB_BK;
            while (kc >= wC) { kc -= wC; ++tap; }
        }
    }
    __device__ __forceinline__ void store(__bf16* __restrict__ s) const {
        const int k4 = threadIdx.x & 15, r = threadIdx.x >> 4;
#pragma unroll
        for (int i = 0; i < 8; ++i)
            *reinterpret_cast<gb_bf16x4*>(s + (r + i * 16) * GB_LD + k4 * 4) = gb_round4(reg[i].x, reg[i].y, reg[i].z, reg[i].w);
    }
};

// operand contiguous along its M/N index (B row-major [K,N], or A given as [K,M]): thread (c4 = tid & 31, kq = tid >> 5) takes, in
// each of the two 32-k halves, the 4 x 4 block of k rows 4 kq .. 4 kq + 3 x columns 4 c4 .. 4 c4 + 3 and writes it transposed
template <bool VEC>
struct GbLoaderMC {
    float4 reg[8];
    int sh, cm, t_k[8];
    __device__ __forceinline__ void prepare(int col0, int k0, int wT, int wC, int wpad, int wdil) {
        if (wT > 0) {
            const int c4 = threadIdx.x & 31, kq = threadIdx.x >> 5;
            const int col = col0 + c4 * 4;
            const int tp = col / wC;
            sh = (tp - wpad) * wdil; cm = col - tp * wC;
#pragma unroll
            for (int i = 0; i < 8; ++i) t_k[i] = (k0 + (i >> 2) * 32 + kq * 4 + (i & 3)) % wT;
        }
    }
    __device__ __forceinline__ void load(const float* __restrict__ base, long ld, int col0, int k0, int cols, int kmax,
                                         int wT, int wC, int wpad, int wdil) {
        const int c4 = threadIdx.x & 31, kq = threadIdx.x >> 5;
        const int col = col0 + c4 * 4;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int k = k0 + (i >> 2) * 32 + kq * 4 + (i & 3);
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            bool ok = k < kmax && col < cols;
            long off = (long)k * ld + col;
            if (wT > 0) {                  // window (A only): element (m, kk) with kk = (b, t) row index, m = (tap, c)
                const int t = t_k[i] + sh;
                ok = ok && t >= 0 && t < wT;
                off = ((long)k + sh) * ld + cm;
                t_k[i] += GB_BK;
                while (t_k[i] >= wT) t_k[i] -= wT;
            }
            if (ok) {
                if (VEC) v = *reinterpret_cast<const float4*>(base + off);
                else {
                    v.x = base[off];
                    if (col + 1 < cols) v.y = base[off + 1];
                    if (col + 2 < cols) v.z = base[off + 2];
                    if (col + 3 < cols) v.w = base[off + 3];
                }
            }
            reg[i] = v;
        }
    }
    __device__ __forceinline__ void store(__bf16* __restrict__ s) const {
        const int c4 = threadIdx.x & 31, kq = threadIdx.x >> 5;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            __bf16* p = s + (c4 * 4) * GB_LD + h * 32 + kq * 4;
            const float4 *q = reg + 4 * h;
            *reinterpret_cast<gb_bf16x4*>(p) = gb_round4(q[0].x, q[1].x, q[2].x, q[3].x);
            *reinterpret_cast<gb_bf16x4*>(p + GB_LD) = gb_round4(q[0].y, q[1].y, q[2].y, q[3].y);
            *reinterpret_cast<gb_bf16x4*>(p + 2 * GB_LD) = gb_round4(q[0].z, q[1].z, q[2].z, q[3].z);
            *reinterpret_cast<gb_bf16x4*>(p + 3 * GB_LD) = gb_round4(q[0].w, q[1].w, q[2].w, q[3].w);
        }
    }
};

template <bool TA, bool TB, bool VEC>
__global__ __launch_bounds__(256) void gemm_bf16_kernel(GemmBfArgs g) {
    __shared__ __attribute__((aligned(16))) __bf16 As[GB_BM * GB_LD];
    __shared__ __attribute__((aligned(16))) __bf16 Bs[GB_BN * GB_LD];

    const int tiles_n = (g.N + GB_BN - 1) / GB_BN;
    int tile = blockIdx.x;                   // XCD-aware tile order (see gemm.hip): each XCD works a contiguous band of tiles
    {
        const int nb = gridDim.x, q = nb >> 3, r = nb & 7, xcd = tile & 7, idx = tile >> 3;
        if (nb >= 64) tile = xcd * q + (xcd < r ? xcd : r) + idx;
    }
    const int tile_m = tile / tiles_n, tile_n = tile % tiles_n;
    const int batch = blockIdx.z / g.split_k, split = blockIdx.z % g.split_k;
    const float* A = g.A + (long)batch * g.stride_a;
    const float* B = g.B + (long)batch * g.stride_b;
    float* C = g.C + (long)batch * g.stride_c;
    const int m0 = tile_m * GB_BM, n0 = tile_n * GB_BN;
    const int kbeg = split * g.k_per_split;
    const int kend = min(g.K, kbeg + g.k_per_split);

    using LA = typename std::conditional<TA, GbLoaderMC<VEC>, GbLoaderKC<VEC>>::type;
    using LB = typename std::conditional<TB, GbLoaderKC<VEC>, GbLoaderMC<VEC>>::type;
    LA la; LB lb;

    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wrow = (wave >> 1) * 64, wcol = (wave & 1) * 64;
    const int l31 = lane & 31, kg = lane >> 5;

    gb_f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    if constexpr (TA) la.prepare(m0, kbeg, g.win_T, g.win_C, g.win_pad, g.win_dil);
    else la.prepare(m0, kbeg, g.win_T, g.win_C);
    if (kbeg < kend) {
        la.load(A, g.lda, m0, kbeg, g.M, kend, g.win_T, g.win_C, g.win_pad, g.win_dil);
        lb.load(B, g.ldb, n0, kbeg, g.N, kend, 0, 1, 0, 1);
    }
    for (int k0 = kbeg; k0 < kend; k0 += GB_BK) {
        __syncthreads();                       // previous tile fully consumed
        la.store(As);
        lb.store(Bs);
        __syncthreads();
        if (k0 + GB_BK < kend) {               // prefetch next tile while this one is multiplied
            la.load(A, g.lda, m0, k0 + GB_BK, g.M, kend, g.win_T, g.win_C, g.win_pad, g.win_dil);
            lb.load(B, g.ldb, n0, k0 + GB_BK, g.N, kend, 0, 1, 0, 1);
        }
#pragma unroll
        for (int ks = 0; ks < GB_BK / 16; ++ks) {
            gb_bf16x8 a[2], b[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) a[i] = *reinterpret_cast<const gb_bf16x8*>(As + (wrow + i * 32 + l31) * GB_LD + ks * 16 + kg * 8);
#pragma unroll
            for (int j = 0; j < 2; ++j) b[j] = *reinterpret_cast<const gb_bf16x8*>(Bs + (wcol + j * 32 + l31) * GB_LD + ks * 16 + kg * 8);
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
        }
    }

    // epilogue: C/D layout col = lane & 31, row = (r & 3) + 8 (r >> 2) + 4 (lane >> 5)
    const bool first_split = (split == 0);
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int col = n0 + wcol + j * 32 + l31;
            if (col >= g.N) continue;
            const float bv = (g.bias != nullptr && first_split) ? g.bias[col] : 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = m0 + wrow + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * kg;
                if (row >= g.M) continue;
                float v = g.alpha * acc[i][j][r] + bv;
                float* dst = C + (long)row * g.ldc + col;
                if (g.split_k > 1) {
                    atomicAdd(dst, v);
                } else {
                    v = gb_act(v, g.act);
                    if (g.accumulate) v += *dst;
                    *dst = v;
                }
            }
        }
}

template <bool TA, bool TB>
static void launch_gemm_bf16(const GemmBfArgs& g, bool vec, dim3 grid, hipStream_t st) {
    if (vec) hipLaunchKernelGGL((gemm_bf16_kernel<TA, TB, true>), grid, dim3(256), 0, st, g);
    else     hipLaunchKernelGGL((gemm_bf16_kernel<TA, TB, false>), grid, dim3(256), 0, st, g);
}

}  // namespace mstts

using namespace mstts;

extern "C" int mstts_gemm_bf16(const mstts_gemm_desc* d, mstts_stream_t stream) {
    MSTTS_REQUIRE(d != nullptr, MSTTS_ERR_SHAPE, "gemm_bf16: null descriptor");
    MSTTS_REQUIRE(d->M >= 0 && d->N >= 0 && d->K >= 0, MSTTS_ERR_SHAPE, "gemm_bf16: negative dims");
    if (d->M == 0 || d->N == 0) return MSTTS_OK;
    MSTTS_REQUIRE(d->A && d->B && d->C, MSTTS_ERR_SHAPE, "gemm_bf16: null operand");
    MSTTS_REQUIRE(d->M < (1LL << 31) && d->N < (1LL << 31) && d->K < (1LL << 31), MSTTS_ERR_SHAPE, "gemm_bf16: dims exceed int32");
    const int batch = d->batch > 0 ? (int)d->batch : 1;
    int split = d->split_k > 1 ? d->split_k : 1;
    MSTTS_REQUIRE(split == 1 || (d->act == MSTTS_ACT_NONE), MSTTS_ERR_SHAPE,
                  "gemm_bf16: split_k needs act=none (output must be pre-zeroed or accumulated into)");
    if (d->win_T > 0) {
        MSTTS_REQUIRE(d->win_C > 0 && d->lda == d->win_C, MSTTS_ERR_SHAPE, "gemm_bf16: window mode needs lda == win_C");
        MSTTS_REQUIRE(d->win_C % 4 == 0, MSTTS_ERR_SHAPE, "gemm_bf16: window mode needs win_C %% 4 == 0");
    }
    GemmBfArgs g;
    g.A = d->A; g.B = d->B; g.C = d->C; g.bias = d->bias;
    g.M = (int)d->M; g.N = (int)d->N; g.K = (int)d->K;
    g.lda = d->lda; g.ldb = d->ldb; g.ldc = d->ldc;
    g.win_T = d->win_T; g.win_C = d->win_C > 0 ? d->win_C : 1; g.win_pad = d->win_pad; g.win_dil = d->win_dil > 0 ? d->win_dil : 1;
    g.act = d->act; g.accumulate = d->accumulate; g.split_k = split;
    g.stride_a = d->stride_a; g.stride_b = d->stride_b; g.stride_c = d->stride_c;
    g.alpha = d->alpha;
    int kps = ((g.K + split - 1) / split + GB_BK - 1) / GB_BK * GB_BK;
    if (kps < GB_BK) kps = GB_BK;
    g.k_per_split = kps;
    bool vec = aligned16(d->A) && aligned16(d->B) && (d->lda % 4 == 0) && (d->ldb % 4 == 0) &&
               (d->stride_a % 4 == 0) && (d->stride_b % 4 == 0);
    vec = vec && (d->trans_a ? (d->M % 4 == 0) : (d->K % 4 == 0));
    vec = vec && (d->trans_b ? (d->K % 4 == 0) : (d->N % 4 == 0));
    if (d->win_T > 0) vec = vec && (d->win_C % 4 == 0);
    dim3 grid(cdiv(d->M, GB_BM) * cdiv(d->N, GB_BN), 1, batch * split);
    hipStream_t st = (hipStream_t)stream;
    const bool ta = d->trans_a != 0, tb = d->trans_b != 0;
    if (!ta && !tb) launch_gemm_bf16<false, false>(g, vec, grid, st);
    else if (!ta && tb) launch_gemm_bf16<false, true>(g, vec, grid, st);
    else if (ta && !tb) launch_gemm_bf16<true, false>(g, vec, grid, st);
    else launch_gemm_bf16<true, true>(g, vec, grid, st);
    MSTTS_CHECK_LAUNCH("gemm_bf16");
    return MSTTS_OK;
}
