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
#include <cstdlib>

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


// ---------------------------------------------------------------------------------------------------------------------------------------
// Big-tile form (round 5): 256 x 256 x 32 per 512-thread workgroup, one workgroup per CU.  The 128 x 128 kernel above moves 32 KB of fp32
// operands per 1.05 MFLOP through the CU's vector-memory path; with one bf16 product per element (not six, as in the fp32 split kernel) the
// matrix cores need 512 cycles for what the loads need ~1 000 for, and the kernel sat at 11-16 % of the bf16 peak.  A 256 x 256 tile
// halves the bytes per flop (64 KB per 4.2 MFLOP).  Eight waves as 2 x 4, each 128 x 64 = 4 x 2 MFMA tiles (128 accumulator registers); every
// wave loads, converts, stages and multiplies; two LDS buffers (2 x 40 KB), ONE raw barrier per K-tile:
//     k-step 0 of tile t   |  registers of tile t + 1 -> bf16 -> buffer (t + 1) & 1  |  loads of tile t + 2 issued  |  k-step 1 of tile t  |  barrier
// (buffer (t + 1) & 1 was last read as tile t - 1, i.e. before the previous barrier).  Loaders as in gemm_split.inc: uniform base + 32-bit
// per-thread offsets fixed at prepare(), unconditional loads (out-of-range lanes read a zero block), branch-free conv-window bookkeeping
// (template flag; needs win_C >= 32 and win_T >= 32 - anything smaller stays on the kernel above).
constexpr int GX_BM = 256, GX_BN = 256, GX_BK = 32, GX_LD = 40, GX_THREADS = 512;     // LDS row stride 80 B: conflict-free b128 fragment reads
constexpr int GX_PLANE = 256 * GX_LD;                                                   // bf16 elements of one operand's tile
constexpr size_t GX_LDS_BYTES = 2 * 2 * GX_PLANE * sizeof(__bf16);
__device__ __attribute__((aligned(16))) const float gx_zero16[4] = {0.f, 0.f, 0.f, 0.f};
typedef float gx_f32x4 __attribute__((ext_vector_type(4)));
typedef const gx_f32x4 __attribute__((address_space(1)))* gx_gptr4;
__device__ __forceinline__ float4 gx_ld4(const float* p) {       // explicit global address space: a flat load would count in lgkmcnt too
    const gx_f32x4 v = *(gx_gptr4)p;
    return make_float4(v[0], v[1], v[2], v[3]);
}
__device__ __forceinline__ void gx_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }   // (not __syncthreads(): its fence drains the loads in flight)

// operand contiguous along k: thread (k4 = tid & 7, r = tid >> 3 in 0..63) takes the float4 k-run k4 of rows r, r + 64, r + 128, r + 192
template <bool VEC, bool WIN>
struct GxLoaderKC {
    const float* ubase;
    unsigned voff[4], rmask;
    int t_row[4], tap, kc, ld_;
    __device__ __forceinline__ void prepare(int tid, const float* __restrict__ base, long ld, int row0, int k0, int rows, int wT, int wC, int wpad, int wdil) {
        const int k4 = tid & 7, r = tid >> 3;
        rmask = 0; ld_ = (int)ld;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int row = row0 + r + i * 64;
            if (row < rows) rmask |= 1u << i;
            voff[i] = (unsigned)((r + i * 64) * (int)ld) + (WIN ? 0u : (unsigned)(k4 * 4));
            if (WIN) t_row[i] = row % wT;
        }
        if (WIN) {
            const int k = k0 + k4 * 4;
            tap = k / wC; kc = k - tap * wC;
            ubase = base + ((long)row0 - (long)wpad * wdil) * ld;
        } else {
            ubase = base + (long)row0 * ld + k0;
        }
    }
    __device__ __forceinline__ void load(int tid, float4 (&reg)[4], int k0, int kmax, int wT, int wC, int wpad, int wdil) {
        const int k4 = tid & 7;
        const int k = k0 + k4 * 4;
        const bool kok = k < kmax;
        const int sh = WIN ? (tap - wpad) * wdil : 0;
        const unsigned wadd = WIN ? (unsigned)(tap * wdil * ld_ + kc) : 0u;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            bool ok = kok && ((rmask >> i) & 1u);
            if (WIN) {
                const int t = t_row[i] + sh;
                ok = ok && t >= 0 && t < wT;
            }
            if (VEC) {
                v = gx_ld4(ok ? ubase + (voff[i] + wadd) : gx_zero16);
            } else if (ok) {
                const float* p = ubase + (voff[i] + wadd);
                v.x = p[0];
                if (k + 1 < kmax) v.y = p[1];
                if (k + 2 < kmax) v.z = p[2];
                if (k + 3 < kmax) v.w = p[3];
            }
            reg[i] = v;
        }
        if (WIN) {
            kc += GX_BK;
            const bool wrap = kc >= wC;
            kc -= wrap ? wC : 0; tap += wrap ? 1 : 0;
        } else {
            ubase += GX_BK;
        }
    }
    __device__ __forceinline__ void store(int tid, const float4 (&reg)[4], __bf16* __restrict__ s) const {
        const int k4 = tid & 7, r = tid >> 3;
#pragma unroll
        for (int i = 0; i < 4; ++i) *reinterpret_cast<gb_bf16x4*>(s + (r + i * 64) * GX_LD + k4 * 4) = gb_round4(reg[i].x, reg[i].y, reg[i].z, reg[i].w);
    }
};

// operand contiguous along its M/N index: thread (kq = tid & 7, c4 = tid >> 3 in 0..63) takes the 4 x 4 block of k rows 4 kq .. + 3 x columns
// 4 c4 .. + 3 as four float4 and writes it transposed (four 8-byte k-runs); kq in the low lane bits (see gemm_split.inc: bank conflicts)
template <bool VEC, bool WIN>
struct GxLoaderMC {
    const float* ubase;
    unsigned voff[4];
    int sh, t_k[4], cols_left;
    __device__ __forceinline__ void prepare(int tid, const float* __restrict__ base, long ld, int col0, int k0, int cols, int wT, int wC, int wpad, int wdil) {
        const int kq = tid & 7, c4 = tid >> 3;
        const int col = col0 + c4 * 4;
        cols_left = cols - col;
        if (WIN) {
            const int tp = col / wC, cm = col - tp * wC;
            sh = (tp - wpad) * wdil;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                t_k[i] = (k0 + kq * 4 + i) % wT;
                voff[i] = (unsigned)((kq * 4 + i + tp * wdil) * (int)ld + cm);
            }
            ubase = base + ((long)k0 - (long)wpad * wdil) * ld;
        } else {
            sh = 0;
#pragma unroll
            for (int i = 0; i < 4; ++i) voff[i] = (unsigned)((kq * 4 + i) * (int)ld + c4 * 4);
            ubase = base + (long)k0 * ld + col0;
        }
    }
    __device__ __forceinline__ void load(int tid, float4 (&reg)[4], int k0, int kmax, long ld, int wT) {
        const int kq = tid & 7;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int k = k0 + kq * 4 + i;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            bool ok = k < kmax && cols_left > 0;
            if (WIN) {
                const int t = t_k[i] + sh;
                ok = ok && t >= 0 && t < wT;
                t_k[i] += GX_BK;
                t_k[i] -= (t_k[i] >= wT) ? wT : 0;
            }
            if (VEC) {
                v = gx_ld4(ok ? ubase + voff[i] : gx_zero16);
            } else if (ok) {
                const float* p = ubase + voff[i];
                v.x = p[0];
                if (cols_left > 1) v.y = p[1];
                if (cols_left > 2) v.z = p[2];
                if (cols_left > 3) v.w = p[3];
            }
            reg[i] = v;
        }
        ubase += GX_BK * ld;
    }
    __device__ __forceinline__ void store(int tid, const float4 (&reg)[4], __bf16* __restrict__ s) const {
        const int kq = tid & 7, c4 = tid >> 3;
        __bf16* p = s + (c4 * 4) * GX_LD + kq * 4;
        *reinterpret_cast<gb_bf16x4*>(p) = gb_round4(reg[0].x, reg[1].x, reg[2].x, reg[3].x);
        *reinterpret_cast<gb_bf16x4*>(p + GX_LD) = gb_round4(reg[0].y, reg[1].y, reg[2].y, reg[3].y);
        *reinterpret_cast<gb_bf16x4*>(p + 2 * GX_LD) = gb_round4(reg[0].z, reg[1].z, reg[2].z, reg[3].z);
        *reinterpret_cast<gb_bf16x4*>(p + 3 * GX_LD) = gb_round4(reg[0].w, reg[1].w, reg[2].w, reg[3].w);
    }
};

template <bool TA, bool TB, bool VEC, bool WIN>
__global__ __launch_bounds__(GX_THREADS) void gemm_bf16_big_kernel(GemmBfArgs g) {
    extern __shared__ __attribute__((aligned(16))) __bf16 gx_lds[];           // [buffer 2][A | B][256 rows][GX_LD]
    const int tiles_n = (g.N + GX_BN - 1) / GX_BN;
    int tile = blockIdx.x;
    {
        const int nb = gridDim.x, q = nb >> 3, r = nb & 7, xcd = tile & 7, idx = tile >> 3;
        if (nb >= 64) tile = xcd * q + (xcd < r ? xcd : r) + idx;
    }
    const int tile_m = tile / tiles_n, tile_n = tile % tiles_n;
    const int batch = blockIdx.z / g.split_k, split = blockIdx.z % g.split_k;
    float* C = g.C + (long)batch * g.stride_c;
    const int m0 = tile_m * GX_BM, n0 = tile_n * GX_BN;
    const int kbeg = split * g.k_per_split;
    const int kend = min(g.K, kbeg + g.k_per_split);
    const int nt = kend > kbeg ? (kend - kbeg + GX_BK - 1) / GX_BK : 0;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wrow = (wave >> 2) * 128, wcol = (wave & 3) * 64;
    const int l31 = lane & 31, kg = lane >> 5;

    using LA = typename std::conditional<TA, GxLoaderMC<VEC, WIN>, GxLoaderKC<VEC, WIN>>::type;
    using LB = typename std::conditional<TB, GxLoaderKC<VEC, false>, GxLoaderMC<VEC, false>>::type;
    LA la; LB lb;
    la.prepare(tid, g.A + (long)batch * g.stride_a, g.lda, m0, kbeg, g.M, g.win_T, g.win_C, g.win_pad, g.win_dil);
    lb.prepare(tid, g.B + (long)batch * g.stride_b, g.ldb, n0, kbeg, g.N, 0, 1, 0, 1);
    float4 ra[4], rb[4];
    auto load = [&](int k) {
        if constexpr (TA) la.load(tid, ra, k, kend, g.lda, g.win_T);
        else la.load(tid, ra, k, kend, g.win_T, g.win_C, g.win_pad, g.win_dil);
        if constexpr (TB) lb.load(tid, rb, k, kend, 0, 1, 0, 1);
        else lb.load(tid, rb, k, kend, g.ldb, 0);
    };
    auto store = [&](int buf) {
        la.store(tid, ra, gx_lds + buf * 2 * GX_PLANE);
        lb.store(tid, rb, gx_lds + buf * 2 * GX_PLANE + GX_PLANE);
    };

    gb_f32x16 acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int aoff = (wrow + l31) * GX_LD + kg * 8, boff = GX_PLANE + (wcol + l31) * GX_LD + kg * 8;
    auto kstep = [&](const __bf16* buf, int ks) {
        gb_bf16x8 a[4], b[2];
#pragma unroll
        for (int i = 0; i < 4; ++i) a[i] = *reinterpret_cast<const gb_bf16x8*>(buf + aoff + i * 32 * GX_LD + ks * 16);
#pragma unroll
        for (int j = 0; j < 2; ++j) b[j] = *reinterpret_cast<const gb_bf16x8*>(buf + boff + j * 32 * GX_LD + ks * 16);
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
    };
    // (Measured and not kept: two fragment register sets with the barrier between a tile's two k-steps, as in gemm_split.inc - 554 instead of
    //  714 TFLOP/s on 8192^3, 416 instead of 500 on the 25 632-row weight gradient; and two operand register sets, i.e. loads issued two
    //  iterations ahead - no change.  With two waves per SIMD the hardware interleaves one wave's fragment reads / staging with the other's MFMAs.)
    if (nt > 0) {
        load(kbeg);
        store(0);
        load(kbeg + GX_BK);                 // (past the end: the loaders read the zero block)
        gx_barrier();
        for (int t = 0; t < nt; ++t) {
            const __bf16* buf = gx_lds + (t & 1) * 2 * GX_PLANE;
            kstep(buf, 0);
            store((t + 1) & 1);             // tile t + 1 (loaded one iteration ago) -> the buffer tile t - 1 was read from
            load(kbeg + (t + 2) * GX_BK);
            kstep(buf, 1);
            gx_barrier();
        }
    }

    // epilogue: C/D layout col = lane & 31, row = (r & 3) + 8 (r >> 2) + 4 (lane >> 5)
    const bool first_split = (split == 0);
#pragma unroll
    for (int i = 0; i < 4; ++i)
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

static int gemm_bf16_big_ready() {
    static int memo[64];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 0;
    if (!memo[dev]) {
        bool ok = true;
#define GX_ATTR1(TA_, TB_, V_, W_) ok = ok && hipFuncSetAttribute((const void*)gemm_bf16_big_kernel<TA_, TB_, V_, W_>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)GX_LDS_BYTES) == hipSuccess
#define GX_ATTR(TA_, TB_) GX_ATTR1(TA_, TB_, true, true); GX_ATTR1(TA_, TB_, true, false); GX_ATTR1(TA_, TB_, false, true); GX_ATTR1(TA_, TB_, false, false)
        GX_ATTR(false, false); GX_ATTR(false, true); GX_ATTR(true, false); GX_ATTR(true, true);
#undef GX_ATTR
#undef GX_ATTR1
        if (!ok) (void)hipGetLastError();
        memo[dev] = ok ? 2 : 1;
    }
    return memo[dev] == 2;
}

template <bool TA, bool TB>
static void launch_gemm_bf16_big(const GemmBfArgs& g, bool vec, dim3 grid, hipStream_t st) {
    const bool win = g.win_T > 0;
    if (vec && win)  hipLaunchKernelGGL((gemm_bf16_big_kernel<TA, TB, true, true>), grid, dim3(GX_THREADS), GX_LDS_BYTES, st, g);
    else if (vec)    hipLaunchKernelGGL((gemm_bf16_big_kernel<TA, TB, true, false>), grid, dim3(GX_THREADS), GX_LDS_BYTES, st, g);
    else if (win)    hipLaunchKernelGGL((gemm_bf16_big_kernel<TA, TB, false, true>), grid, dim3(GX_THREADS), GX_LDS_BYTES, st, g);
    else             hipLaunchKernelGGL((gemm_bf16_big_kernel<TA, TB, false, false>), grid, dim3(GX_THREADS), GX_LDS_BYTES, st, g);
}

}  // namespace mstts

using namespace mstts;

// Process-global development switches (A/B runs, tests), set through the entry points below - the library reads no environment variable
// (multi_speaker_tts_amd/lib.py maps its MSTTS_GEMM_* variables onto these setters when it loads the library).
static int g_bf16_big = 1;
extern "C" int mstts_gemm_bf16_big(int32_t on) { g_bf16_big = on != 0; return MSTTS_OK; }
static int g_bf16_big_min = 160;         // (measured: 180 workgroups of the 256 x 256 kernel beat 720 of the small one by 1.3 x, 101 lose to 404 by 1.2 x)
namespace mstts { void gemm_bf16_set_big_min(int n) { if (n > 0) g_bf16_big_min = n; } }       // (mstts_gemm_big_min_workgroups, csrc/gemm.hip)
static int g_bf16_autocut = 1;           // 0: no K-cut of the library's own (A/B)
extern "C" int mstts_gemm_bf16_autocut(int32_t on) { g_bf16_autocut = on != 0; return MSTTS_OK; }

namespace mstts { int gemm_deterministic_now(); }
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
    hipStream_t st = (hipStream_t)stream;
    const bool ta = d->trans_a != 0, tb = d->trans_b != 0;
    // the 256 x 256 tile where the operand is large enough to fill the chip with such tiles (at least ~3/4 of a round of 256 workgroups) and the
    // strides fit its 32-bit tile-relative offsets; MSTTS_GEMM_BF16_BIG=0 / mstts_gemm_bf16_big(0) keeps the 128 x 128 kernel (A/B, tests)
    const long big_wgs = (long)cdiv(d->M, GX_BM) * cdiv(d->N, GX_BN) * batch * split;
    const bool big = g_bf16_big && d->M >= 192 && d->N >= 192 && big_wgs >= g_bf16_big_min && d->lda < (1 << 22) && d->ldb < (1 << 22) &&
                     (d->win_T <= 0 || (d->win_T >= GX_BK && d->win_C >= GX_BK)) && gemm_bf16_big_ready();
    if (big) {
        int kpsb = ((g.K + split - 1) / split + GX_BK - 1) / GX_BK * GX_BK;
        if (kpsb < GX_BK) kpsb = GX_BK;
        g.k_per_split = kpsb;
        dim3 gridb(cdiv(d->M, GX_BM) * cdiv(d->N, GX_BN), 1, batch * split);
        if (!ta && !tb) launch_gemm_bf16_big<false, false>(g, vec, gridb, st);
        else if (!ta && tb) launch_gemm_bf16_big<false, true>(g, vec, gridb, st);
        else if (ta && !tb) launch_gemm_bf16_big<true, false>(g, vec, gridb, st);
        else launch_gemm_bf16_big<true, true>(g, vec, gridb, st);
        MSTTS_CHECK_LAUNCH("gemm_bf16 (256 x 256 tile)");
        return MSTTS_OK;
    }
    // The 128 x 128 kernel runs two workgroups per CU: 512 slots.  A contraction whose tile list is far from a multiple of that - the encoder's
    // 4 096-row convolutions (128 tiles), the 80 / 84-column products (14 - 201 tiles) - and whose caller asked for NO cut (split_k <= 1) is cut
    // along K so that it fills one round: pieces of at least 128 contraction steps, cost model rounds x (K per piece + 300) x (1 + 2 % per piece
    // for its atomics), fitted to a sweep of every such call of a train step (tools/gemm_split_sweep.py --config3: 0.31 ms per step against
    // cuts chosen for the fp32 kernel's tiles).  Only without a fused activation, for one batch, and not under mstts_gemm_deterministic; without
    // `accumulate` the output's N columns (not its row pitch: C may be a column band of a wider matrix) are cleared here first.  A caller's own
    // split_k > 1 is honoured exactly - as mstts_gemm_f32 does - so a caller that reasons about the number of pieces (0 + p + q) gets that number.
    if (g_bf16_autocut && split == 1 && d->act == MSTTS_ACT_NONE && batch == 1 && !gemm_deterministic_now()) {
        const long tiles = (long)cdiv(d->M, GB_BM) * cdiv(d->N, GB_BN);
        int best = 1;
        double best_cost = 0.0;
        for (int sk = 1; sk <= 64; ++sk) {
            if (sk > 1 && g.K / sk < 128) break;
            const double cost = (double)cdiv(tiles * sk, 512) * ((double)g.K / sk + 300.0) * (1.0 + 0.02 * sk);
            if (sk == 1 || cost < best_cost * 0.97) { best = sk; best_cost = cost; }      // (a cut has to win by 3 %)
        }
        if (best > 1) {
            if (!d->accumulate) {
                if (hipMemset2DAsync(d->C, (size_t)d->ldc * 4, 0, (size_t)d->N * 4, (size_t)d->M, st) != hipSuccess)
                    MSTTS_REQUIRE(false, MSTTS_ERR_LAUNCH, "gemm_bf16: clearing the output of a K-cut contraction failed");
            }
            split = best;
            g.split_k = split;
            kps = ((g.K + split - 1) / split + GB_BK - 1) / GB_BK * GB_BK;
            if (kps < GB_BK) kps = GB_BK;
            g.k_per_split = kps;
        }
    }
    dim3 grid(cdiv(d->M, GB_BM) * cdiv(d->N, GB_BN), 1, batch * split);
    if (!ta && !tb) launch_gemm_bf16<false, false>(g, vec, grid, st);
    else if (!ta && tb) launch_gemm_bf16<false, true>(g, vec, grid, st);
    else if (ta && !tb) launch_gemm_bf16<true, false>(g, vec, grid, st);
    else launch_gemm_bf16<true, true>(g, vec, grid, st);
    MSTTS_CHECK_LAUNCH("gemm_bf16");
    return MSTTS_OK;
}
