// fp32 GEMM on the gfx950 matrix cores (v_mfma_f32_32x32x2_f32: exact fp32, fp32 accumulate).
//
// One kernel family covers every dense contraction of the Tacotron2 hot path:
//   * dense / prenet / projection / memory-layer forward          (A row-major, B [K,N])
//   * conv1d 'same' as an implicit-im2col GEMM in NWC layout      (A = overlapping row windows)
//   * data gradients  dX = dY . W^T                               (B read transposed)
//   * weight gradients dW = X^T . dY, also over conv windows      (A read transposed, split-K)
// Tile: BM x 128 x 32 per 256-thread workgroup (4 wave64), LDS tiles are k-major so an MFMA
// operand is one conflict-free ds_read_b32 per lane; the next K-tile is prefetched into registers
// while the current one is multiplied.  Replaces tf.layers.conv1d / tf.layers.dense /
// tf.matmul call sites of the reference (Modules.py:29-36,125-132,243-247,311-314;
// ZoneoutLSTMCell.py:228) and their autodiff gradients.
#include "common.h"
#include <type_traits>
#include <cmath>
#include <cstdlib>

namespace mstts {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int BK = 32;
constexpr int BN = 128;

struct GemmArgs {
    const float* A; const float* B; float* C; const float* bias;
    int M, N, K;
    long lda, ldb, ldc;
    int win_T, win_C, win_pad, win_dil;
    int act, accumulate, split_k;
    long stride_a, stride_b, stride_c;
    float alpha;
    int k_per_split;
    // body + tail (see mstts_gemm_f32): blocks [0, body) own one whole output tile each; the last tiles of the list are cut along K
    // into tail_s pieces of tail_kps each, blocks body + (tile - body) * tail_s + piece, accumulated with atomics.  body = all tiles
    // when the split is off.
    int body, tail_s, tail_kps;
    int band;                     // gemm_split_kernel: tile list in column bands of this many tiles (0: row-major)
};

__device__ __forceinline__ float apply_act(float v, int act) {
    if (act == MSTTS_ACT_RELU) return fmaxf(v, 0.f);
    if (act == MSTTS_ACT_TANH) return tanhf_(v);
    if (act == MSTTS_ACT_SIGMOID) return sigmoidf_(v);
    return v;
}

// ---- tile loaders -------------------------------------------------------------------------
// "KC": operand contiguous along k in memory (A row-major, or B given as [N,K]).
//   rows = the M (or N) index, R rows per tile.  reg[] holds R*BK/256/4 float4 per thread.
// "MC": operand contiguous along its M/N index (B row-major [K,N], or A given as [K,M]).
// Out-of-range elements are read from this zero block instead of being skipped: the loads stay unconditional (no exec-mask
// branches in the K loop, the scheduler can hide them behind the MFMAs) and the padding is zero without a select on the data.
__device__ __attribute__((aligned(16))) const float gemm_zero16[4] = {0.f, 0.f, 0.f, 0.f};
// ... and they go through an explicit GLOBAL-address-space pointer: the select between the operand and the zero block otherwise degrades to
// a flat pointer, flat loads count in lgkmcnt as well as vmcnt, and every s_waitcnt lgkmcnt(0) in front of an MFMA group (placed for the
// LDS fragment reads) would then also wait for the K-tile prefetch issued just before - the whole global latency exposed once per K-tile.
typedef float gemm_f32x4 __attribute__((ext_vector_type(4)));
typedef const gemm_f32x4 __attribute__((address_space(1)))* gemm_gptr4;
__device__ __forceinline__ float4 gemm_ld4(const float* p) {
    const gemm_f32x4 v = *(gemm_gptr4)p;
    return make_float4(v[0], v[1], v[2], v[3]);
}

// Addressing: every load is  uniform base (SGPRs, advanced once per K-tile)  +  a 32-bit per-thread offset fixed at prepare()  -
// no 64-bit address arithmetic and no bounds arithmetic beyond one compare in the K loop (it cost 9 % of the matrix-core time).
template <int R, bool VEC>
struct LoaderKC {
    static constexpr int NV = R * BK / 4 / 256;      // float4 per thread
    float4 reg[NV];
    // window: element (row, k) is tap j = k / C of a dilated 'same' conv: source row = row + (j - pad) * dil, valid iff it
    // stays inside the row's length-T sequence.  A thread's rows never change (t_row = row % T once) and k advances by BK per
    // load (tap / kc kept incrementally): prepare() once, then load() for k0, k0 + BK, ...
    const float* ubase;                               // non-window: base + row0 ld + k0 ; window: base + (row0 - pad dil) ld (for the
                                                      // first tile that is in front of the operand: only ever added to offsets of valid taps)
    unsigned voff[NV], rmask;                         // (r + 32 i) ld (+ 4 k4 without window); bit i: row inside the operand
    int t_row[NV], tap, kc, ld_;
    __device__ __forceinline__ void prepare(const float* __restrict__ base, long ld, int row0, int k0, int rows,
                                            int wT, int wC, int wpad, int wdil) {
        const int k4 = threadIdx.x & 7, r = threadIdx.x >> 3;
        rmask = 0; ld_ = (int)ld;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int row = row0 + r + i * 32;
            if (row < rows) rmask |= 1u << i;
            voff[i] = (unsigned)((r + i * 32) * (int)ld) + (wT > 0 ? 0u : (unsigned)(k4 * 4));
            if (wT > 0) t_row[i] = row % wT;
        }
        if (wT > 0) {
            const int k = k0 + k4 * 4;
            tap = k / wC; kc = k - tap * wC;
            ubase = base + ((long)row0 - (long)wpad * wdil) * ld;
        } else {
            ubase = base + (long)row0 * ld + k0;
        }
    }
    __device__ __forceinline__ void load(int k0, int kmax, int wT, int wC, int wpad, int wdil) {
        const int k4 = threadIdx.x & 7;
        const int k = k0 + k4 * 4;
        const bool kok = k < kmax;
        const int sh = wT > 0 ? (tap - wpad) * wdil : 0;
        const unsigned wadd = wT > 0 ? (unsigned)(tap * wdil * ld_ + kc) : 0u;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            bool ok = kok && ((rmask >> i) & 1u);
            if (wT > 0) {
                const int t = t_row[i] + sh;
                ok = ok && t >= 0 && t < wT;
            }
            if (VEC) {
                v = gemm_ld4(ok ? ubase + (voff[i] + wadd) : gemm_zero16);
            } else if (ok) {
                const float* p = ubase + (voff[i] + wadd);
                v.x = p[0];
                if (k + 1 < kmax) v.y = p[1];
                if (k + 2 < kmax) v.z = p[2];
                if (k + 3 < kmax) v.w = p[3];
            }
            reg[i] = v;
        }
        if (wT > 0) {                                  // next call is for k0 + BK
            kc += BK;
            while (kc >= wC) { kc -= wC; ++tap; }
        } else {
            ubase += BK;
        }
    }
    __device__ __forceinline__ void store(float* __restrict__ s, int ldS) const {
        const int k4 = threadIdx.x & 7, r = threadIdx.x >> 3;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            float* p = s + (k4 * 4) * ldS + r + i * 32;
            p[0] = reg[i].x; p[ldS] = reg[i].y; p[2 * ldS] = reg[i].z; p[3 * ldS] = reg[i].w;
        }
    }
};

template <int R, bool VEC, int BKT = BK>
struct LoaderMC {
    static constexpr int C4 = R / 4;                  // float4 per k-row
    static constexpr int KR = 256 / C4;               // k-rows per pass
    static constexpr int NV = BKT / KR;
    float4 reg[NV];
    // window (A only): element (m, kk) with kk=(b,t) row index, m=(tap, c):
    //   valid iff 0 <= (kk % T) + m / C - pad < T; address = base[(kk + (tap - pad) dil) * ld + c]
    // a thread's column (tap, c) is fixed, its k-rows advance by BK per load.
    const float* ubase;                               // base + k0 ld + col0 (window: base + (k0 - pad dil) ld), advanced by BK ld per load
    unsigned voff[NV];
    int sh, t_k[NV], cols_left;                       // cols_left: columns of the operand from this thread's first one (<= 0: none)
    __device__ __forceinline__ void prepare(const float* __restrict__ base, long ld, int col0, int k0, int cols,
                                            int wT, int wC, int wpad, int wdil) {
        const int c4 = threadIdx.x % C4, kr = threadIdx.x / C4;
        const int col = col0 + c4 * 4;
        cols_left = cols - col;
        if (wT > 0) {
            const int tp = col / wC, cm = col - tp * wC;
            sh = (tp - wpad) * wdil;
#pragma unroll
            for (int i = 0; i < NV; ++i) {
                t_k[i] = (k0 + kr + i * KR) % wT;
                voff[i] = (unsigned)((kr + i * KR + tp * wdil) * (int)ld + cm);
            }
            ubase = base + ((long)k0 - (long)wpad * wdil) * ld;
        } else {
            sh = 0;
#pragma unroll
            for (int i = 0; i < NV; ++i) voff[i] = (unsigned)((kr + i * KR) * (int)ld + c4 * 4);
            ubase = base + (long)k0 * ld + col0;
        }
    }
    __device__ __forceinline__ void load(int k0, int kmax, long ld, int wT) {
        const int kr = threadIdx.x / C4;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int k = k0 + kr + i * KR;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            bool ok = k < kmax && cols_left > 0;
            if (wT > 0) {
                const int t = t_k[i] + sh;
                ok = ok && t >= 0 && t < wT;
                t_k[i] += BKT;                         // next call is for k0 + BKT
                if (wT >= BKT) t_k[i] -= (t_k[i] >= wT) ? wT : 0;     // one wrap at most (every real sequence is longer than a K-tile)
                else while (t_k[i] >= wT) t_k[i] -= wT;
            }
            if (VEC) {
                v = gemm_ld4(ok ? ubase + voff[i] : gemm_zero16);
            } else if (ok) {
                const float* p = ubase + voff[i];
                v.x = p[0];
                if (cols_left > 1) v.y = p[1];
                if (cols_left > 2) v.z = p[2];
                if (cols_left > 3) v.w = p[3];
            }
            reg[i] = v;
        }
        ubase += BKT * ld;
    }
    __device__ __forceinline__ void store(float* __restrict__ s, int ldS) const {
        const int c4 = threadIdx.x % C4, kr = threadIdx.x / C4;
#pragma unroll
        for (int i = 0; i < NV; ++i)
            *reinterpret_cast<float4*>(s + (kr + i * KR) * ldS + c4 * 4) = reg[i];
    }
};

// epilogue of one wave's WM x WN grid of 32 x 32 MFMA tiles: C/D layout col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)
template <int WM, int WN>
__device__ __forceinline__ void gemm_store_tile(const GemmArgs& g, const f32x16 (&acc)[WM][WN], float* __restrict__ C, int row0, int col0,
                                                int lane, bool with_bias, bool atomic) {
#pragma unroll
    for (int i = 0; i < WM; ++i)
#pragma unroll
        for (int j = 0; j < WN; ++j) {
            const int col = col0 + j * 32 + (lane & 31);
            if (col >= g.N) continue;
            const float bv = (g.bias != nullptr && with_bias) ? g.bias[col] : 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = row0 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                if (row >= g.M) continue;
                float v = g.alpha * acc[i][j][r] + bv;
                float* dst = C + (long)row * g.ldc + col;
                if (atomic) {
                    atomicAdd(dst, v);
                } else {
                    v = apply_act(v, g.act);
                    if (g.accumulate) v += *dst;
                    *dst = v;
                }
            }
        }
}

template <int BM, bool TA, bool TB, bool VEC>
__global__ __launch_bounds__(256) void gemm_kernel(GemmArgs g) {
    // LDS tiles, k-major.  Row stride: +4 floats where the tile is filled with float4 stores (keeps
    // them 16-byte aligned), +2 where it is filled by the transposing loader (4*stride == 8 mod 32
    // banks -> its scattered ds_write_b32 are at most 2-way conflicted, which is free).
    constexpr int LDA_S = TA ? BM + 4 : BM + 2, LDB_S = TB ? BN + 2 : BN + 4;
    constexpr int A_FLOATS = (BK * LDA_S + 3) / 4 * 4;
    __shared__ __attribute__((aligned(16))) float smem[A_FLOATS + BK * LDB_S];
    float* As = smem;
    float* Bs = smem + A_FLOATS;

    const int tiles_n = (g.N + BN - 1) / BN;
    // XCD-aware tile order: block b runs on XCD b % 8 and every XCD has its own L2, so give each XCD a contiguous range
    // of the (tile_m-major) tile list - a band of A rows it re-reads from its own L2 - instead of every eighth tile.
    int tile = blockIdx.x, piece = -1;
    if (tile < g.body) {
        const int nb = g.body, q = nb >> 3, r = nb & 7, xcd = tile & 7, idx = tile >> 3;
        if (nb >= 64) tile = xcd * q + (xcd < r ? xcd : r) + idx;
    } else {
        const int u = tile - g.body;
        tile = g.body + u / g.tail_s;
        piece = u % g.tail_s;
    }
    const int tile_m = tile / tiles_n, tile_n = tile % tiles_n;
    const int batch = blockIdx.z / g.split_k, split = blockIdx.z % g.split_k;
    const float* A = g.A + (long)batch * g.stride_a;
    const float* B = g.B + (long)batch * g.stride_b;
    float* C = g.C + (long)batch * g.stride_c;
    const int m0 = tile_m * BM, n0 = tile_n * BN;
    const int kbeg = piece >= 0 ? piece * g.tail_kps : split * g.k_per_split;
    const int kend = min(g.K, kbeg + (piece >= 0 ? g.tail_kps : g.k_per_split));

    using LA = typename std::conditional<TA, LoaderMC<BM, VEC>, LoaderKC<BM, VEC>>::type;
    using LB = typename std::conditional<TB, LoaderKC<BN, VEC>, LoaderMC<BN, VEC>>::type;
    LA la; LB lb;

    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    // wave tiling: BM=128 -> 2x2 waves of 64x64 ; BM=64 -> 1x4 waves of 64x32 ; BM=32 -> 1x4 waves of 32x32
    constexpr int WM = (BM >= 64) ? 2 : 1;            // 32-row MFMA tiles per wave
    constexpr int WN = (BM == 128) ? 2 : 1;
    const int wm = (BM == 128) ? (wave >> 1) : 0;
    const int wn = (BM == 128) ? (wave & 1) : wave;
    const int wrow = wm * WM * 32, wcol = wn * WN * 32;

    f32x16 acc[WM][WN];
#pragma unroll
    for (int i = 0; i < WM; ++i)
#pragma unroll
        for (int j = 0; j < WN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    auto load_a = [&](int k) {
        if constexpr (TA) la.load(k, kend, g.lda, g.win_T);
        else la.load(k, kend, g.win_T, g.win_C, g.win_pad, g.win_dil);
    };
    auto load_b = [&](int k) {
        if constexpr (TB) lb.load(k, kend, 0, 1, 0, 1);
        else lb.load(k, kend, g.ldb, 0);
    };
    la.prepare(A, g.lda, m0, kbeg, g.M, g.win_T, g.win_C, g.win_pad, g.win_dil);
    lb.prepare(B, g.ldb, n0, kbeg, g.N, 0, 1, 0, 1);
    if (kbeg < kend) {
        load_a(kbeg);
        load_b(kbeg);
    }
    for (int k0 = kbeg; k0 < kend; k0 += BK) {
        __syncthreads();                       // previous tile fully consumed
        la.store(As, LDA_S);
        lb.store(Bs, LDB_S);
        __syncthreads();
        if (k0 + BK < kend) {                  // prefetch next tile while this one is multiplied
            load_a(k0 + BK);
            load_b(k0 + BK);
        }
        const int kh = lane >> 5, l31 = lane & 31;
        // operands of k-step kk + 2 are read from LDS BEFORE the MFMAs of k-step kk issue (two register sets, the scheduler is told
        // to keep that order): left to itself the compiler reuses one register pair for every k-step and waits for each LDS read
        // behind the previous MFMA group - an LDS round trip of dead matrix-core time per k-step (128 -> 14x TFLOP/s on dW shapes)
        float a[2][WM], b[2][WN];
#pragma unroll
        for (int i = 0; i < WM; ++i) a[0][i] = As[kh * LDA_S + wrow + i * 32 + l31];
#pragma unroll
        for (int j = 0; j < WN; ++j) b[0][j] = Bs[kh * LDB_S + wcol + j * 32 + l31];
#pragma unroll
        for (int kk = 0; kk < BK; kk += 2) {
            const int cur = (kk >> 1) & 1, nxt = cur ^ 1;
            if (kk + 2 < BK) {
#pragma unroll
                for (int i = 0; i < WM; ++i) a[nxt][i] = As[(kk + 2 + kh) * LDA_S + wrow + i * 32 + l31];
#pragma unroll
                for (int j = 0; j < WN; ++j) b[nxt][j] = Bs[(kk + 2 + kh) * LDB_S + wcol + j * 32 + l31];
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < WM; ++i)
#pragma unroll
                for (int j = 0; j < WN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[cur][i], b[cur][j], acc[i][j], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
    }

    gemm_store_tile<WM, WN>(g, acc, C, m0 + wrow, n0 + wcol, lane, split == 0 && piece <= 0, g.split_k > 1 || piece >= 0);
}

__global__ void gemm_tail_act_kernel(float* __restrict__ C, long ldc, int N, long n, int act) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    float* p = C + (i / N) * ldc + (i % N);
    *p = apply_act(*p, act);
}

template <int BM, bool TA, bool TB>
static void launch_gemm(const GemmArgs& g, bool vec, dim3 grid, hipStream_t st) {
    if (vec) hipLaunchKernelGGL((gemm_kernel<BM, TA, TB, true>), grid, dim3(256), 0, st, g);
    else     hipLaunchKernelGGL((gemm_kernel<BM, TA, TB, false>), grid, dim3(256), 0, st, g);
}

#include "gemm_split.inc"

}  // namespace mstts

using namespace mstts;

// Process-global development switches (A/B runs, tests).  The library reads NO environment variable: multi_speaker_tts_amd/lib.py maps its
// MSTTS_GEMM_* variables onto these setters once, when it loads the library.
static int g_tail_split = 1;
extern "C" int mstts_gemm_tail_split(int32_t on) { g_tail_split = on != 0; return MSTTS_OK; }
// 1 (default): every contraction with more than 32 rows runs as the six-product bf16 split (gemm_split.inc: fp32 accuracy at 6/16 of
// the f32-input MFMA time); 0: all of them on v_mfma_f32_32x32x2_f32 (bitwise an fmaf chain).  Tests and the bench's A/B leg switch it.
static int g_split3 = 1;
extern "C" int mstts_gemm_split3(int32_t on) { g_split3 = on != 0; return MSTTS_OK; }
// 1 (default): split contractions that fill the chip with 256 x 256 tiles take gemm_split_big_kernel; 0: the 128 x 128 x 32 producer / consumer kernel
static int g_split_big = 1;
extern "C" int mstts_gemm_split_big(int32_t on) { g_split_big = on != 0; return MSTTS_OK; }
// from how many 256 x 256 workgroups on the big-tile kernels are taken (fp32 split kernel, bf16 kernel; <= 0 keeps the current value)
static int g_split_big_min = 160;
namespace mstts { void gemm_bf16_set_big_min(int n); }
extern "C" int mstts_gemm_big_min_workgroups(int32_t f32_split, int32_t bf16) {
    if (f32_split > 0) g_split_big_min = f32_split;
    gemm_bf16_set_big_min(bf16);
    return MSTTS_OK;
}
// Per calling thread: 1 = no K-cuts that the caller did not ask for (neither the tail of a long tile list nor a short list cut entirely), so a
// contraction without split_k adds its K range in one fixed order and its result is bit-reproducible from run to run.  The inference engines
// set it around their forward passes (a vocoder is a long chain of contractions; at random weights it amplifies last-bit differences).
static thread_local int t_deterministic = 0;
extern "C" int mstts_gemm_deterministic(int32_t on) { t_deterministic = on != 0; return MSTTS_OK; }
namespace mstts { int gemm_deterministic_now() { return t_deterministic; } }     // (csrc/gemm_bf16.hip asks before it cuts a contraction along K on its own)

extern "C" int mstts_gemm_f32(const mstts_gemm_desc* d, mstts_stream_t stream) {
    MSTTS_REQUIRE(d != nullptr, MSTTS_ERR_SHAPE, "gemm: null descriptor");
    MSTTS_REQUIRE(d->M >= 0 && d->N >= 0 && d->K >= 0, MSTTS_ERR_SHAPE, "gemm: negative dims");
    if (d->M == 0 || d->N == 0) return MSTTS_OK;
    MSTTS_REQUIRE(d->A && d->B && d->C, MSTTS_ERR_SHAPE, "gemm: null operand");
    MSTTS_REQUIRE(d->M < (1LL << 31) && d->N < (1LL << 31) && d->K < (1LL << 31), MSTTS_ERR_SHAPE, "gemm: dims exceed int32");
    MSTTS_REQUIRE(d->lda >= 0 && d->ldb >= 0 && d->lda < (1 << 24) && d->ldb < (1 << 24), MSTTS_ERR_SHAPE,
                  "gemm: row strides must be below 2^24 elements (tile-relative offsets are 32-bit)");
    const int batch = d->batch > 0 ? (int)d->batch : 1;
    int split = d->split_k > 1 ? d->split_k : 1;
    MSTTS_REQUIRE(split == 1 || (d->act == MSTTS_ACT_NONE), MSTTS_ERR_SHAPE,
                  "gemm: split_k needs act=none (output must be pre-zeroed or accumulated into)");
    if (d->win_T > 0) {
        MSTTS_REQUIRE(d->win_C > 0 && d->lda == d->win_C, MSTTS_ERR_SHAPE, "gemm: window mode needs lda == win_C");
        MSTTS_REQUIRE(d->win_C % 4 == 0, MSTTS_ERR_SHAPE, "gemm: window mode needs win_C %% 4 == 0");
    }
    GemmArgs g;
    g.A = d->A; g.B = d->B; g.C = d->C; g.bias = d->bias;
    g.M = (int)d->M; g.N = (int)d->N; g.K = (int)d->K;
    g.lda = d->lda; g.ldb = d->ldb; g.ldc = d->ldc;
    g.win_T = d->win_T; g.win_C = d->win_C > 0 ? d->win_C : 1; g.win_pad = d->win_pad; g.win_dil = d->win_dil > 0 ? d->win_dil : 1;
    g.act = d->act; g.accumulate = d->accumulate; g.split_k = split;
    g.stride_a = d->stride_a; g.stride_b = d->stride_b; g.stride_c = d->stride_c;
    g.alpha = d->alpha;
    int kps = ((g.K + split - 1) / split + BK - 1) / BK * BK;
    if (kps < BK) kps = BK;
    g.k_per_split = kps;
    // vector path: every float4 the loaders form must be 16-byte aligned and must not straddle
    // a conv tap or the end of a row
    bool vec = aligned16(d->A) && aligned16(d->B) && (d->lda % 4 == 0) && (d->ldb % 4 == 0) &&
               (d->stride_a % 4 == 0) && (d->stride_b % 4 == 0);
    vec = vec && (d->trans_a ? (d->M % 4 == 0) : (d->K % 4 == 0));
    vec = vec && (d->trans_b ? (d->K % 4 == 0) : (d->N % 4 == 0));
    if (d->win_T > 0) vec = vec && (d->win_C % 4 == 0);
    const bool skinny = d->M <= 32;
    // 128-row tiles unless 64-row tiles fill the 256 CUs' rounds visibly better (25 632 x 512: 804 tiles = 3.14 rounds -> 4, but
    // 1 608 half tiles = 6.28 -> 7; 4 096 x 512: 128 tiles use half the chip, 256 half tiles all of it); the half tile re-reads
    // the B operand twice as often, so it has to win by more than 5 %
    int bm = skinny ? 32 : 128, tail_s = 1, tail_rem = 0;
    // (K < 160: the pipeline's prologue and padding tile outweigh the matrix-core time saved; the split kernel has 128-row tiles only)
    const bool split3 = g_split3 && !skinny && d->K >= 160 && (d->win_T <= 0 || (d->win_T >= BK && d->win_C >= BK)) && gemm_split_lds_ready();
    if (!skinny) {
        const double t128 = (double)cdiv(d->M, 128) * cdiv(d->N, BN) * batch * split, t64 = (double)cdiv(d->M, 64) * cdiv(d->N, BN) * batch * split;
        const double e128 = t128 / (ceil(t128 / 256.0) * 256.0), e64 = t64 / (ceil(t64 / 256.0) * 256.0);
        if (e64 * 0.95 > e128 && !split3) bm = 64;
        // Body + tail: when a list of more than 256 tiles ends in a small fraction of a round (25 632 x 512: 804 tiles = 3 rounds + 36),
        // the last round runs 36 tiles on 256 CUs.  Cut those tiles along K into floor(256 / rem) pieces each instead, so the
        // remainder is one short round of the whole chip: 3 + 1/7 rounds instead of 4 (or 7 rounds of half tiles).  Costs in units of
        // one round of 128-row tiles; a half tile is 0.54 (it re-reads B twice as often), a piece pays its atomics and a short K loop.
        if (g_tail_split && !t_deterministic && batch == 1 && split == 1 && (d->act == MSTTS_ACT_NONE || !d->accumulate)) {
            double best = (bm == 128 ? ceil(t128 / 256.0) : ceil(t64 / 256.0) * 0.54);
            const int ktiles = cdiv(d->K, BK);
            for (int cand = 128; cand >= (split3 ? 128 : 64); cand -= 64) {
                const long t = (long)cdiv(d->M, cand) * cdiv(d->N, BN);
                int rem = (int)(t % 256);
                // with an activation the tail is made of whole tile rows (the activation runs over those rows once the pieces are summed)
                if (d->act != MSTTS_ACT_NONE) rem = (int)(t - (t - rem) / cdiv(d->N, BN) * cdiv(d->N, BN));
                // (the split kernel has one workgroup per CU and only 128-row tiles: a list of at most 128 tiles - the encoder's 4 096-row
                //  convolutions - leaves half the chip idle, so there EVERY tile is cut along K)
                if (split3 && t <= 128) rem = (int)t;
                else if (t <= 256 || rem == 0 || rem > 128) continue;
                int s = 256 / rem;
                if (s > 16) s = 16;
                if (s > ktiles / 8) s = ktiles / 8;              // a piece keeps at least 8 K-tiles: below that its prologue and atomics cost more than the round it saves
                if (s < 2) continue;
                const double c = ((double)(t / 256) + 1.25 / s + 0.06) * (cand == 128 ? 1.0 : 0.54);
                if (c < best) { best = c; bm = cand; tail_s = s; tail_rem = rem; }
            }
        }
    }
    hipStream_t st = (hipStream_t)stream;
    {   // the 256 x 256 x 16 form of the split where such tiles fill the chip (from 160 workgroups on; one per CU)
        const long big_wgs = (long)cdiv(d->M, GSB_BM) * cdiv(d->N, GSB_BN) * batch * split;
        if (split3 && g_split_big && d->M >= 192 && d->N >= 192 && big_wgs >= g_split_big_min && d->lda < (1 << 22) && d->ldb < (1 << 22) &&
            (d->win_T <= 0 || (d->win_T >= GSB_BK && d->win_C >= GSB_BK)) && (d->act == MSTTS_ACT_NONE || split == 1) && gemm_split_big_ready()) {
            int kpsb = ((g.K + split - 1) / split + GSB_BK - 1) / GSB_BK * GSB_BK;
            if (kpsb < GSB_BK) kpsb = GSB_BK;
            g.k_per_split = kpsb;
            g.body = 0; g.tail_s = 1; g.tail_kps = kpsb; g.band = 0;
            dim3 gridb(cdiv(d->M, GSB_BM) * cdiv(d->N, GSB_BN), 1, batch * split);
            const bool ta_ = d->trans_a != 0, tb_ = d->trans_b != 0;
            if (!ta_ && !tb_) launch_gemm_split_big<false, false>(g, vec, gridb, st);
            else if (!ta_ && tb_) launch_gemm_split_big<false, true>(g, vec, gridb, st);
            else if (ta_ && !tb_) launch_gemm_split_big<true, false>(g, vec, gridb, st);
            else launch_gemm_split_big<true, true>(g, vec, gridb, st);
            MSTTS_CHECK_LAUNCH("gemm_f32 (split, 256 x 256 tile)");
            return MSTTS_OK;
        }
    }
    const int tiles = cdiv(d->M, bm) * cdiv(d->N, BN);
    g.body = tiles; g.tail_s = 1; g.tail_kps = kps;
    g.band = (split3 && tail_s == 1 && cdiv(d->N, BN) > 8) ? 8 : 0;
    if (tail_s > 1) {
        // the last `rem` tiles of the list as rem x tail_s blocks behind the body (highest block ids: they start as body tiles retire)
        const int rem = tail_rem, tiles_n = cdiv(d->N, BN);
        g.body = tiles - rem; g.tail_s = tail_s;
        g.tail_kps = (cdiv(g.K, BK) + tail_s - 1) / tail_s * BK;
        if (!d->accumulate) {           // the pieces add onto zeros: clear every tile row that holds a tail tile (body tiles of a mixed row overwrite)
            const long r0 = (long)(g.body / tiles_n) * bm;
            if (hipMemset2DAsync(d->C + r0 * d->ldc, (size_t)d->ldc * 4, 0, (size_t)d->N * 4, (size_t)(d->M - r0), st) != hipSuccess)
                MSTTS_REQUIRE(false, MSTTS_ERR_LAUNCH, "gemm: clearing the tail tiles failed");
        }
    }
    dim3 grid(g.body + (tiles - g.body) * g.tail_s, 1, batch * split);
    const bool ta = d->trans_a != 0, tb = d->trans_b != 0;
    if (split3) {
        if (!ta && !tb) launch_gemm_split<false, false>(g, vec, grid, st);
        else if (!ta && tb) launch_gemm_split<false, true>(g, vec, grid, st);
        else if (ta && !tb) launch_gemm_split<true, false>(g, vec, grid, st);
        else launch_gemm_split<true, true>(g, vec, grid, st);
    } else if (skinny) {
        if (!ta && !tb) launch_gemm<32, false, false>(g, vec, grid, st);
        else if (!ta && tb) launch_gemm<32, false, true>(g, vec, grid, st);
        else if (ta && !tb) launch_gemm<32, true, false>(g, vec, grid, st);
        else launch_gemm<32, true, true>(g, vec, grid, st);
    } else if (bm == 64) {
        if (!ta && !tb) launch_gemm<64, false, false>(g, vec, grid, st);
        else if (!ta && tb) launch_gemm<64, false, true>(g, vec, grid, st);
        else if (ta && !tb) launch_gemm<64, true, false>(g, vec, grid, st);
        else launch_gemm<64, true, true>(g, vec, grid, st);
    } else {
        if (!ta && !tb) launch_gemm<128, false, false>(g, vec, grid, st);
        else if (!ta && tb) launch_gemm<128, false, true>(g, vec, grid, st);
        else if (ta && !tb) launch_gemm<128, true, false>(g, vec, grid, st);
        else launch_gemm<128, true, true>(g, vec, grid, st);
    }
    MSTTS_CHECK_LAUNCH("gemm_f32");
    if (tail_s > 1 && d->act != MSTTS_ACT_NONE) {        // the tail's tile rows hold bias + the summed pieces: apply the activation there
        const long r0 = (long)(g.body / cdiv(d->N, BN)) * bm;
        const long n = (d->M - r0) * d->N;
        hipLaunchKernelGGL(gemm_tail_act_kernel, dim3((unsigned)cdiv(n, 256)), dim3(256), 0, st, d->C + r0 * d->ldc, d->ldc, (int)d->N, n, d->act);
        MSTTS_CHECK_LAUNCH("gemm_f32 tail activation");
    }
    return MSTTS_OK;
}
