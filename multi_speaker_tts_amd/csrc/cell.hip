// Fused zoneout-LSTM cell step for the decoder time loops (gfx950): the gates product AND the cell
// update of ZoneoutLSTMCell.call (ZoneoutLSTMCell.py:228-264) in ONE launch, no partial slabs.
//
//   gates[b, :] = X[b, :K] . W[K, 4H] (+ xw[b, :] | + bias)            i, j, f, o   (ZoneoutLSTMCell.py:228-230)
//   c = sigmoid(f + 1) c_prev + sigmoid(i) tanh(j) ; m = sigmoid(o) tanh(c)           (:237-248)
//   c' = keep . zc . (c - c_prev) + c_prev ; h' = keep . zh . (m - h_prev) + h_prev   (:259-271)
//
// Work cut: workgroup g owns the FOUR hidden units 4g .. 4g+3, i.e. the 16 gate columns {gate*H + 4g + u}, for all
// (<= 32) batch rows and the whole reduction - H/4 = 256 workgroups at the reference width, one per CU, each with
// the complete pre-activations of its units, so the cell update is an epilogue and nothing is exchanged.
// The kernel W is a per-optimizer-step derived copy (like the folded cell-0 kernel), stored in exactly the order
// the lanes consume it (mstts_pack_cell_fwd): the four waves of a workgroup split K, wave w / iteration it / lane l
// reads one float4 = W[w K/4 + 16 it + 4 (l >> 4) + {0..3}][column (l & 15)] - a contiguous 1 KB per wave load.
// The activations need no staging either: the same lane needs X[row (l & 15)][the same four k] as one float4 (the k order
// inside a v_mfma_f32_16x16x4_f32 reduction is free as long as A and B agree).  Read from a row-major block that is 16
// different cache lines per 4-lane group - measured 17.6 us per cell, slower than product + pointwise (16.2) - so the
// activation block is kept in the lanes' order as well (cell_act_offset): its producers (the previous cell's epilogue,
// the attention step's context) write their few elements per thread to both layouts, the row-major history BPTT needs and
// the packed block the next cell reads with one contiguous 1 KB per wave load.  Every operand goes global -> register in one
// round trip, all loads issued before the first MFMA, in consumption order.
// Exact fp32 arithmetic (v_mfma_f32_16x16x4_f32); deterministic (fixed reduction order).
#include "common.h"

namespace mstts {

typedef float cf32x4 __attribute__((ext_vector_type(4)));
constexpr int CELL_MAX_NIT = 32;          // 16-row k-steps per wave: K/4 <= 512

typedef __bf16 cbf16x8 __attribute__((ext_vector_type(8)));
constexpr int CELL_MAX_NIT_BF = 16;       // bf16 form: 32-k steps per wave, K/4 <= 512

struct CellFwd {
    const float* Xp;                            // packed activation block [ceil(B/32)*32, K] (cell_act_offset); bf16 form: __bf16 data
    const float* Wp;                            // packed kernel (mstts_pack_cell_fwd / _bf16)
    const float* xw; int xw_ld;                 // optional additive pre-activations [B][4H] (gate-major), or null
    const float* bias;                          // [4H] gate-major, or null
    const float* c_prev; const float* h_prev; int h_prev_ld;
    const uint8_t* zc; const uint8_t* zh; float keep;
    float* out; int out_ld;                     // un-zoned m (the cell output, ZoneoutLSTMCell.py:264)
    float* c_next; float* h_next; int h_next_ld;
    float* acts; float* c_raw;                  // [B][4H] gate activations, [B][H] raw cell state (BPTT); may be null
    PackedDst out_p, h_next_p;                  // optional packed copies of m / h' for the cells that consume them next
    const int32_t* lengths; int step, reverse;  // SEQ: tf.nn.dynamic_rnn semantics (row b live while step < lengths[b]; reversed direction)
    int xw_st, out_st;                          // SEQ: position strides of xw / out (row b, position pos at b * ld + pos * st)
    int B, H, K;
};
struct CellFwdPair { CellFwd d[2]; };           // two independent cells in one launch (blockIdx.z): the two BiLSTM directions

// SEQ = dynamic_rnn semantics (Modules.py:49-73): past its length a row's output is zero and its state is carried through
// unchanged; the reversed direction reads / writes position len - 1 - step.  PAIR = two cells per launch.
// BF16 (config 3): operands are bf16 copies (packed kernel, packed activations written by their producers already rounded), products
// on v_mfma_f32_16x16x32_bf16 with fp32 accumulation; NIT then counts 32-k steps (K / 128).  Everything after the product is fp32.
template <int NIT, bool TWO, bool SEQ, bool PAIR, bool BF16 = false>
__device__ __forceinline__ void cell_fwd_body(const CellFwd& d) {
    __shared__ float red[4][32][17];
    const int g = blockIdx.x, m0 = blockIdx.y * 32;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int j = lane & 15, kq = lane >> 4;
    constexpr bool EXACT = NIT > 0;
    constexpr int UNROLL = EXACT ? NIT : (BF16 ? CELL_MAX_NIT_BF : CELL_MAX_NIT);
    const int nit = EXACT ? NIT : (BF16 ? d.K >> 7 : d.K >> 6);
    const int H = d.H;
    // ---- epilogue operands of this thread (row er, unit eu), requested first: they are tiny and must not cost a round trip later
    const int er = threadIdx.x >> 2, eu = 4 * g + (threadIdx.x & 3);
    const bool elive = threadIdx.x < 128 && m0 + er < d.B;
    const int eb = m0 + er;
    // (raw values only: anything computed from them here would make the compiler wait for the round trip before the main loads go out)
    float cp = 0.f, hp = 0.f, xwv[4] = {0.f, 0.f, 0.f, 0.f}, bv[4] = {0.f, 0.f, 0.f, 0.f};
    uint8_t zcv = 1, zhv = 1;
    int pos = 0;
    bool rlive = true;                           // SEQ: this row still inside its sequence
    if (elive) {
        if (SEQ) {
            const int len = d.lengths ? d.lengths[eb] : 0x7fffffff;
            rlive = d.step < len;
            pos = (d.reverse && rlive) ? len - 1 - d.step : d.step;
        }
        cp = d.c_prev[eb * H + eu];
        hp = d.h_prev[eb * d.h_prev_ld + eu];
        if (d.zc) zcv = d.zc[eb * H + eu];
        if (d.zh) zhv = d.zh[eb * H + eu];
        if (d.xw) {
#pragma unroll
            for (int q = 0; q < 4; ++q) xwv[q] = d.xw[eb * d.xw_ld + (SEQ ? pos * d.xw_st : 0) + q * H + eu];
        }
        if (d.bias) {
#pragma unroll
            for (int q = 0; q < 4; ++q) bv[q] = d.bias[q * H + eu];
        }
    }
    // ---- every operand load of the product, in consumption order (activations of k-step `it` for both row tiles, then its
    //      weights): a wave's loads return in issue order, so the MFMA chain starts with the first triple and runs under the
    //      weight stream instead of behind it
    cf32x4 c00 = (cf32x4){0.f, 0.f, 0.f, 0.f}, c01 = c00, c10 = c00, c11 = c00;
    if constexpr (BF16) {
        // 16 B per lane per step for the kernel and for each row tile: half the bytes of the fp32 form, 1/4 of its k-steps
        const uint4* wp = reinterpret_cast<const uint4*>(d.Wp) + ((long)(g * 4 + wave) * nit) * 64 + lane;
        const uint4* xp = reinterpret_cast<const uint4*>(d.Xp) + (long)blockIdx.y * (512 * nit) + (wave * nit * 2) * 64 + lane;
        uint4 wreg[UNROLL], a0[UNROLL], a1[TWO ? UNROLL : 1];
#pragma unroll
        for (int it = 0; it < UNROLL; ++it) {
            a0[it] = make_uint4(0, 0, 0, 0); wreg[it] = a0[it];
            if (TWO) a1[it] = a0[it];
            if (EXACT || it < nit) {
                a0[it] = xp[it * 128];
                if (TWO) a1[it] = xp[it * 128 + 64];
                wreg[it] = wp[it * 64];
            }
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int it = 0; it < UNROLL; ++it) {
            if (EXACT || it < nit) {
                union { uint4 u; cbf16x8 v; } w_, x_, y_;
                w_.u = wreg[it]; x_.u = a0[it];
                if (it & 1) c01 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(x_.v, w_.v, c01, 0, 0, 0);
                else c00 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(x_.v, w_.v, c00, 0, 0, 0);
                if (TWO) {
                    y_.u = a1[it];
                    if (it & 1) c11 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(y_.v, w_.v, c11, 0, 0, 0);
                    else c10 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(y_.v, w_.v, c10, 0, 0, 0);
                }
            }
        }
    } else {
    const float* wp = d.Wp + ((long)(g * 4 + wave) * nit) * 256 + lane * 4;
        const float* xp = d.Xp + (long)blockIdx.y * (2048 * nit) + (wave * nit * 2) * 256 + lane * 4;
        cf32x4 wreg[UNROLL], a0[UNROLL], a1[TWO ? UNROLL : 1];
#pragma unroll
        for (int it = 0; it < UNROLL; ++it) {
            a0[it] = (cf32x4){0.f, 0.f, 0.f, 0.f};
            wreg[it] = (cf32x4){0.f, 0.f, 0.f, 0.f};
            if (TWO) a1[it] = (cf32x4){0.f, 0.f, 0.f, 0.f};
            if (EXACT || it < nit) {
                a0[it] = *reinterpret_cast<const cf32x4*>(xp + it * 512);
                if (TWO) a1[it] = *reinterpret_cast<const cf32x4*>(xp + it * 512 + 256);
                wreg[it] = *reinterpret_cast<const cf32x4*>(wp + it * 256);
            }
        }
        // every load above stays above: without this the scheduler sinks each load to its MFMA and runs the stream three loads deep
        __builtin_amdgcn_sched_barrier(0);
        // ---- MFMA chain: two independent accumulators per row tile
#pragma unroll
        for (int it = 0; it < UNROLL; ++it) {
            if (EXACT || it < nit) {
                const cf32x4 w = wreg[it], x = a0[it];
                c00 = __builtin_amdgcn_mfma_f32_16x16x4f32(x[0], w[0], c00, 0, 0, 0);
                c01 = __builtin_amdgcn_mfma_f32_16x16x4f32(x[1], w[1], c01, 0, 0, 0);
                if (TWO) {
                    const cf32x4 y = a1[it];
                    c10 = __builtin_amdgcn_mfma_f32_16x16x4f32(y[0], w[0], c10, 0, 0, 0);
                    c11 = __builtin_amdgcn_mfma_f32_16x16x4f32(y[1], w[1], c11, 0, 0, 0);
                }
                c00 = __builtin_amdgcn_mfma_f32_16x16x4f32(x[2], w[2], c00, 0, 0, 0);
                c01 = __builtin_amdgcn_mfma_f32_16x16x4f32(x[3], w[3], c01, 0, 0, 0);
                if (TWO) {
                    const cf32x4 y = a1[it];
                    c10 = __builtin_amdgcn_mfma_f32_16x16x4f32(y[2], w[2], c10, 0, 0, 0);
                    c11 = __builtin_amdgcn_mfma_f32_16x16x4f32(y[3], w[3], c11, 0, 0, 0);
                }
            }
        }
    }
    // ---- the four K-quarters meet in LDS: red[wave][row][column]
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        red[wave][kq * 4 + r][j] = c00[r] + c01[r];
        red[wave][16 + kq * 4 + r][j] = TWO ? c10[r] + c11[r] : 0.f;
    }
    __syncthreads();
    if (!elive) return;
    // ---- cell update of (row er, unit eu): column of gate q is 4 q + (unit & 3)
    const int ec = threadIdx.x & 3;
    float g4[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) g4[q] = ((red[0][er][4 * q + ec] + red[1][er][4 * q + ec]) + (red[2][er][4 * q + ec] + red[3][er][4 * q + ec])) + (xwv[q] + bv[q]);
    const float kc = zcv ? d.keep : 0.f, kh = zhv ? d.keep : 0.f;
    // activations through one hardware exp each (v_exp_f32, ~1e-7 relative): the library forms are 3-4x the instructions on the
    // tail of a latency-bound kernel
    float si = sigmoidf_(g4[0]), tj = tanhf_(g4[1]), sf = sigmoidf_(g4[2] + 1.0f), so = sigmoidf_(g4[3]);
    float c = sf * cp + si * tj;
    float m = so * tanhf_(c);
    float hn = kh * (m - hp) + hp, cn = kc * (c - cp) + cp;
    if (SEQ && !rlive) { m = 0.f; hn = hp; cn = cp; si = 0.f; tj = 0.f; sf = 0.f; so = 0.f; c = cp; }
    d.c_next[eb * H + eu] = cn;
    d.h_next[eb * d.h_next_ld + eu] = hn;
    d.out[eb * d.out_ld + (SEQ ? pos * d.out_st : 0) + eu] = m;
    if (d.out_p.base) packed_store(d.out_p, eb, eu, m);
    if (d.h_next_p.base) packed_store(d.h_next_p, eb, eu, hn);
    if (d.acts) { float* a = d.acts + eb * 4 * H + eu; a[0] = si; a[H] = tj; a[2 * H] = sf; a[3 * H] = so; }
    if (d.c_raw) d.c_raw[eb * H + eu] = c;
}

template <int NIT, bool TWO, bool SEQ>
__global__ __launch_bounds__(256) void cell_fwd_kernel(CellFwd d) { cell_fwd_body<NIT, TWO, SEQ, false>(d); }
template <int NIT, bool TWO, bool SEQ>
__global__ __launch_bounds__(256) void cell_fwd_pair_kernel(CellFwdPair p) { cell_fwd_body<NIT, TWO, SEQ, true>(p.d[blockIdx.z]); }
template <int NIT, bool TWO>
__global__ __launch_bounds__(256) void cell_fwd_bf16_kernel(CellFwd d) { cell_fwd_body<NIT, TWO, false, false, true>(d); }

// bf16 packers: W[K, 4H] / X[B, K] fp32 -> the bf16 consumption order (round to nearest even)
__global__ void pack_cell_fwd_bf16_kernel(const float* __restrict__ W, long ldw, __bf16* __restrict__ Wp, int K, int H) {
    const long n = (long)K * 4 * H;
    const int nit = K >> 7;
    for (long p = blockIdx.x * (long)blockDim.x + threadIdx.x; p < n; p += (long)gridDim.x * blockDim.x) {
        const int e = (int)(p & 7), lane = (int)((p >> 3) & 63);
        long r = p >> 9;
        const int it = (int)(r % nit); r /= nit;
        const int wave = (int)(r & 3), g = (int)(r >> 2);
        const int j = lane & 15, kq = lane >> 4;
        const int k = wave * (nit * 32) + 32 * it + 8 * kq + e;
        const int col = (j >> 2) * H + 4 * g + (j & 3);
        Wp[p] = (__bf16)W[(long)k * ldw + col];
    }
}
__global__ void pack_cell_act_bf16_kernel(const float* __restrict__ X, long ldx, __bf16* __restrict__ Xp, int B, int K) {
    const int nit = K >> 7;
    const long n = (long)((B + 31) / 32 * 32) * K;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const int r = (int)(i / K), k = (int)(i - (long)r * K);
        Xp[cell_act_offset_bf16(r, k, nit)] = (__bf16)(r < B ? X[(long)r * ldx + k] : 0.f);
    }
}

// row-major X[B, K] (row stride ldx) -> packed activation block (rows B .. 32*ceil(B/32)-1 are written as zeros)
__global__ void pack_cell_act_kernel(const float* __restrict__ X, long ldx, float* __restrict__ Xp, int B, int K) {
    const int nit = K >> 6;
    const long n = (long)((B + 31) / 32 * 32) * K;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const int r = (int)(i / K), k = (int)(i - (long)r * K);
        Xp[cell_act_offset(r, k, nit)] = r < B ? X[(long)r * ldx + k] : 0.f;
    }
}

// W[K, 4H] (row stride ldw, gate-major columns i | j | f | o) -> the consumption order of cell_fwd_kernel
__global__ void pack_cell_fwd_kernel(const float* __restrict__ W, long ldw, float* __restrict__ Wp, int K, int H) {
    const long n = (long)K * 4 * H;
    const int nit = K >> 6;
    for (long p = blockIdx.x * (long)blockDim.x + threadIdx.x; p < n; p += (long)gridDim.x * blockDim.x) {
        const int e = (int)(p & 3), lane = (int)((p >> 2) & 63);
        long r = p >> 8;
        const int it = (int)(r % nit); r /= nit;
        const int wave = (int)(r & 3), g = (int)(r >> 2);
        const int j = lane & 15, kq = lane >> 4;
        const int k = wave * (nit * 16) + 16 * it + 4 * kq + e;
        const int col = (j >> 2) * H + 4 * g + (j & 3);
        Wp[p] = W[(long)k * ldw + col];
    }
}

}  // namespace mstts
using namespace mstts;

/* 1 when the fused cell step can run a [B, K] x [K, 4H] cell: whole 4-unit groups, K split over four waves in 16-row steps */
extern "C" int32_t mstts_cell_fwd_supported(int64_t H, int64_t K) {
    return (H >= 4 && H % 4 == 0 && K >= 64 && K % 64 == 0 && K / 64 <= CELL_MAX_NIT && 4 * H * K < (1LL << 31)) ? 1 : 0;
}

extern "C" int mstts_pack_cell_fwd(const float* W, int64_t ldw, float* Wp, int64_t K, int64_t H, mstts_stream_t s) {
    MSTTS_REQUIRE(W && Wp && mstts_cell_fwd_supported(H, K), MSTTS_ERR_SHAPE, "pack_cell_fwd: unsupported shape (H %% 4, K %% 64, K <= 2048)");
    const long n = K * 4 * H;
    hipLaunchKernelGGL(pack_cell_fwd_kernel, dim3((unsigned)((n + 255) / 256 < 4096 ? (n + 255) / 256 : 4096)), dim3(256), 0, (hipStream_t)s, W, (long)ldw, Wp, (int)K, (int)H);
    MSTTS_CHECK_LAUNCH("pack_cell_fwd");
    return MSTTS_OK;
}

extern "C" int64_t mstts_cell_act_floats(int64_t B, int64_t K) { return (B + 31) / 32 * 32 * K; }

extern "C" int mstts_pack_cell_act(const float* X, int64_t ldx, float* Xp, int64_t B, int64_t K, mstts_stream_t s) {
    MSTTS_REQUIRE(X && Xp && B >= 1 && K >= 64 && K % 64 == 0 && K / 64 <= CELL_MAX_NIT, MSTTS_ERR_SHAPE, "pack_cell_act: K %% 64, K <= 2048 required");
    const long n = mstts_cell_act_floats(B, K);
    hipLaunchKernelGGL(pack_cell_act_kernel, dim3((unsigned)((n + 255) / 256 < 2048 ? (n + 255) / 256 : 2048)), dim3(256), 0, (hipStream_t)s, X, (long)ldx, Xp, (int)B, (int)K);
    MSTTS_CHECK_LAUNCH("pack_cell_act");
    return MSTTS_OK;
}

namespace mstts {
int packed_dst_from(const mstts_cell_packed_dst* p, int64_t width, PackedDst* o, const char* what) {
    o->base = p ? p->base : nullptr; o->nit = 0; o->col0 = 0; o->bf = 0;
    if (!o->base) return MSTTS_OK;
    if (p->bf16) {
        MSTTS_REQUIRE(p->K >= 128 && p->K % 128 == 0 && p->K / 128 <= 16 && p->col0 >= 0 && p->col0 + width <= p->K, MSTTS_ERR_SHAPE,
                      "bad packed bf16 destination %s (K %% 128, K <= 2048, col0 + width <= K)", what);
        o->nit = (int)(p->K / 128); o->col0 = (int)p->col0; o->bf = 1;
        return MSTTS_OK;
    }
    MSTTS_REQUIRE(p->K >= 64 && p->K % 64 == 0 && p->K / 64 <= CELL_MAX_NIT && p->col0 >= 0 && p->col0 + width <= p->K, MSTTS_ERR_SHAPE,
                  "bad packed destination %s (K %% 64, K <= 2048, col0 + width <= K)", what);
    o->nit = (int)(p->K / 64); o->col0 = (int)p->col0;
    return MSTTS_OK;
}
}  // namespace mstts

static int cell_args(const mstts_cell_fwd_desc* q, CellFwd* out) {
    MSTTS_REQUIRE(q && q->Xp && q->Wp && q->c_prev && q->h_prev && q->out && q->c_next && q->h_next, MSTTS_ERR_SHAPE, "cell_fwd: null pointer");
    MSTTS_REQUIRE(q->bf16 ? mstts_cell_fwd_bf16_supported(q->H, q->K) : mstts_cell_fwd_supported(q->H, q->K), MSTTS_ERR_SHAPE,
                  "cell_fwd: unsupported shape (H %% 4; K %% 64 (bf16: %% 128), K <= 2048)");
    MSTTS_REQUIRE(q->B >= 1 && aligned16(q->Xp) && aligned16(q->Wp), MSTTS_ERR_ALIGN, "cell_fwd: 16-byte aligned Xp / Wp required");
    const int64_t widest = q->K > 4 * q->H ? q->K : 4 * q->H;
    const int64_t span = (q->xw_ld > q->out_ld ? q->xw_ld : q->out_ld) > widest ? (q->xw_ld > q->out_ld ? q->xw_ld : q->out_ld) : widest;
    MSTTS_REQUIRE((q->B + 32) * span < (1LL << 31), MSTTS_ERR_SHAPE, "cell_fwd: block too large for 32-bit indexing");
    CellFwd& d = *out;
    d.Xp = q->Xp; d.Wp = q->Wp; d.xw = q->xw; d.xw_ld = (int)q->xw_ld; d.bias = q->bias;
    d.c_prev = q->c_prev; d.h_prev = q->h_prev; d.h_prev_ld = (int)(q->h_prev_ld ? q->h_prev_ld : q->H);
    d.zc = q->zc; d.zh = q->zh; d.keep = 1.f - q->zoneout;
    d.out = q->out; d.out_ld = (int)(q->out_ld ? q->out_ld : q->H);
    d.c_next = q->c_next; d.h_next = q->h_next; d.h_next_ld = (int)(q->h_next_ld ? q->h_next_ld : q->H);
    d.acts = q->acts; d.c_raw = q->c_raw; d.B = (int)q->B; d.H = (int)q->H; d.K = (int)q->K;
    d.lengths = q->lengths; d.step = q->step; d.reverse = q->reverse; d.xw_st = (int)q->xw_st; d.out_st = (int)q->out_st;
    int rc = packed_dst_from(&q->out_p, q->H, &d.out_p, "out_p"); if (rc) return rc;
    return packed_dst_from(&q->h_next_p, q->H, &d.h_next_p, "h_next_p");
}
static bool cell_is_seq(const mstts_cell_fwd_desc* q) { return q->lengths || q->reverse || q->xw_st != 0 || q->out_st != 0; }

#define MSTTS_CF_LAUNCH(KERN, N, ARG)                                                                                   \
    do {                                                                                                                \
        if (two && seq) hipLaunchKernelGGL((KERN<N, true, true>), grid, dim3(256), 0, (hipStream_t)s, ARG);             \
        else if (two) hipLaunchKernelGGL((KERN<N, true, false>), grid, dim3(256), 0, (hipStream_t)s, ARG);              \
        else if (seq) hipLaunchKernelGGL((KERN<N, false, true>), grid, dim3(256), 0, (hipStream_t)s, ARG);              \
        else hipLaunchKernelGGL((KERN<N, false, false>), grid, dim3(256), 0, (hipStream_t)s, ARG);                      \
    } while (0)

extern "C" int32_t mstts_cell_fwd_bf16_supported(int64_t H, int64_t K) {
    return (H >= 4 && H % 4 == 0 && K >= 128 && K % 128 == 0 && K / 128 <= CELL_MAX_NIT_BF && 4 * H * K < (1LL << 31)) ? 1 : 0;
}
extern "C" int mstts_pack_cell_fwd_bf16(const float* W, int64_t ldw, void* Wp16, int64_t K, int64_t H, mstts_stream_t s) {
    MSTTS_REQUIRE(W && Wp16 && mstts_cell_fwd_bf16_supported(H, K), MSTTS_ERR_SHAPE, "pack_cell_fwd_bf16: unsupported shape (H %% 4, K %% 128, K <= 2048)");
    const long n = K * 4 * H;
    hipLaunchKernelGGL(pack_cell_fwd_bf16_kernel, dim3((unsigned)((n + 255) / 256 < 4096 ? (n + 255) / 256 : 4096)), dim3(256), 0, (hipStream_t)s, W, (long)ldw,
                       (__bf16*)Wp16, (int)K, (int)H);
    MSTTS_CHECK_LAUNCH("pack_cell_fwd_bf16");
    return MSTTS_OK;
}
extern "C" int mstts_pack_cell_act_bf16(const float* X, int64_t ldx, void* Xp16, int64_t B, int64_t K, mstts_stream_t s) {
    MSTTS_REQUIRE(X && Xp16 && B >= 1 && K >= 128 && K % 128 == 0 && K / 128 <= CELL_MAX_NIT_BF, MSTTS_ERR_SHAPE, "pack_cell_act_bf16: K %% 128, K <= 2048 required");
    const long n = mstts_cell_act_floats(B, K);
    hipLaunchKernelGGL(pack_cell_act_bf16_kernel, dim3((unsigned)((n + 255) / 256 < 2048 ? (n + 255) / 256 : 2048)), dim3(256), 0, (hipStream_t)s, X, (long)ldx,
                       (__bf16*)Xp16, (int)B, (int)K);
    MSTTS_CHECK_LAUNCH("pack_cell_act_bf16");
    return MSTTS_OK;
}

extern "C" int mstts_cell_fwd(const mstts_cell_fwd_desc* q, mstts_stream_t s) {
    CellFwd d;
    int rc = cell_args(q, &d); if (rc) return rc;
    const dim3 grid((unsigned)(q->H / 4), (unsigned)((q->B + 31) / 32));
    if (q->bf16) {
        MSTTS_REQUIRE(!cell_is_seq(q), MSTTS_ERR_SHAPE, "cell_fwd: the bf16 form has no sequence mode");
        const int nitb = (int)(q->K / 128);
        const bool two_ = q->B > 16;
#define MSTTS_CFB(N) do { if (two_) hipLaunchKernelGGL((cell_fwd_bf16_kernel<N, true>), grid, dim3(256), 0, (hipStream_t)s, d);                   \
                          else hipLaunchKernelGGL((cell_fwd_bf16_kernel<N, false>), grid, dim3(256), 0, (hipStream_t)s, d); } while (0)
        if (nitb == 16) MSTTS_CFB(16); else if (nitb == 14) MSTTS_CFB(14); else MSTTS_CFB(0);
#undef MSTTS_CFB
        MSTTS_CHECK_LAUNCH("cell_fwd_bf16");
        return MSTTS_OK;
    }
    const int nit = (int)(q->K / 64);
    const bool two = q->B > 16, seq = cell_is_seq(q);
    if (nit == 32) MSTTS_CF_LAUNCH(cell_fwd_kernel, 32, d); else if (nit == 28) MSTTS_CF_LAUNCH(cell_fwd_kernel, 28, d); else MSTTS_CF_LAUNCH(cell_fwd_kernel, 0, d);
    MSTTS_CHECK_LAUNCH("cell_fwd");
    return MSTTS_OK;
}

/* two independent cells of the same shape (B, H, K) in ONE launch - the forward and backward direction of a BiLSTM step */
extern "C" int mstts_cell_fwd_pair(const mstts_cell_fwd_desc* a, const mstts_cell_fwd_desc* b, mstts_stream_t s) {
    CellFwdPair p;
    int rc = cell_args(a, &p.d[0]); if (rc) return rc;
    rc = cell_args(b, &p.d[1]); if (rc) return rc;
    MSTTS_REQUIRE(a->B == b->B && a->H == b->H && a->K == b->K && !a->bf16 && !b->bf16, MSTTS_ERR_SHAPE, "cell_fwd_pair: the two cells must have the same B, H, K (fp32 form)");
    const dim3 grid((unsigned)(a->H / 4), (unsigned)((a->B + 31) / 32), 2);
    const int nit = (int)(a->K / 64);
    const bool two = a->B > 16, seq = true;            // (the sequence form also covers plain cells: lengths NULL, strides 0)
    if (nit == 32) MSTTS_CF_LAUNCH(cell_fwd_pair_kernel, 32, p); else if (nit == 28) MSTTS_CF_LAUNCH(cell_fwd_pair_kernel, 28, p); else MSTTS_CF_LAUNCH(cell_fwd_pair_kernel, 0, p);
    MSTTS_CHECK_LAUNCH("cell_fwd_pair");
    return MSTTS_OK;
}
#undef MSTTS_CF_LAUNCH
