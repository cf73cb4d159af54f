// Location-sensitive attention step for gfx950 (Location_Sensitive_Attention.py:43-85, plus the
// TF BahdanauAttention score mask / softmax and the AttentionWrapper context it inherits).
//
// The decoder calls this once per mel frame, strictly in sequence, so one step has to spread over the whole chip and cost
// as few dependent round trips as possible.
//   forward, one launch  (lsa_step_kernel, grid (T/16 | M/96, B)): every workgroup computes the energies of 16 encoder
//       positions (folded 31-tap location filter -> tanh -> wave shuffle reduction), the workgroups of a row exchange
//       their slices inside the launch as 8-byte {epoch, value} granules, then each runs the row softmax and streams its
//       96-column slice of values[b].
//   backward, one launch (lsa_step_bwd_kernel, B x T/8 workgroups): d_align = G + values . d_ctx, the row-wide dot(a, d_a) formed
//       locally as dot(a, G) + ctx . d_ctx (no exchange), d_energy, d_query (atomics), the filter-transpose operand h.
//   The two-launch forms (lsa_energy + lsa_context, lsa_dalign + lsa_denergy) are kept: they need no granule buffer, are
//   what the single-launch kernels are tested against, and serve sequences that do not fit the single-launch geometry.
//   Parameter gradients are hoisted out of the time loop into one batched recompute kernel (lsa_param_bwd), so the
//   sequential path carries no read-modify-write traffic.
#include "common.h"

namespace mstts {

constexpr int TS = 8;       // encoder positions per workgroup (T/8 x B workgroups: 512 at T=128, B=32 -> 2 waves per SIMD)
constexpr int A_ = 128;     // attention units  (hp.Attention.Memory_Size)
constexpr int CH_ = 32;     // location conv channels (hp.Attention.Conv.Channel)
constexpr int FLD = CH_ + 4; // LDS row stride of the location features: rows stay 16-byte aligned for b128 broadcast reads
constexpr int KS_MAX = 31;
constexpr int LKT_LD = 36;   // row stride of the by-unit filter copy: 144 B - 16-byte aligned rows that do not all start in the same cache set (128 B did: +0.3 us)  // location conv taps upper bound (= the reference's hp.Attention.Conv.Kernel_Size)

__device__ __forceinline__ float fast_tanh(float x) {
    // tanh via one exp; relative error ~1e-6 over the energy pre-activation range
    const float ax = fabsf(x);
    const float e = __expf(-2.0f * ax);
    return copysignf((1.0f - e) / (1.0f + e), x);
}

// The per-step kernels below are latency-bound (a few KB per workgroup, 801 dependent steps), so each
// one issues ALL of its global loads first - one memory round trip - and only then touches LDS.
//
// Folded location filter: the reference applies conv1d(31 taps, 1 -> 32 ch, +bias) and then a
// bias-free dense 32 -> 128 to the cumulative alignment (Location_Sensitive_Attention.py:48-61) with
// nothing in between, so the pair is ONE 31-tap filter into 128 channels:
//     loc[t,k] = loc_b[k] + sum_j cum[t+j-pad] * loc_k[j,k],   loc_k = conv_k . dense_k,  loc_b = conv_b . dense_k
// The host refreshes loc_k/loc_b after every optimizer step (like the folded cell-0 kernel); the
// gradient comes back as d_loc_k and is unfolded into d_conv_k / d_conv_b / d_dense_k by three tiny GEMMs.

// ---------------------------------------------------------------------------------------------
// forward: energies   grid (B, T/TS), 256 threads = (k = tid & 127, grp = tid >> 7)
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void lsa_energy_kernel(mstts_lsa_const c, const float* __restrict__ q, int q_parts, long q_pstride,
                                                         float* __restrict__ q_sum, const float* __restrict__ cum,
                                                         float* __restrict__ energy) {
    __shared__ float s_cum[TS + KS_MAX - 1];
    __shared__ float s_red[TS][2];
    const int b = blockIdx.x, t0 = blockIdx.y * TS, T = (int)c.T, KS = (int)c.KS, pad = (KS - 1) / 2;
    const int k = threadIdx.x & (A_ - 1), grp = threadIdx.x >> 7;
    // ---- every global load of this workgroup, issued back to back
    float cwin = 0.f;
    if (threadIdx.x < TS + KS - 1) {
        const int t = t0 - pad + threadIdx.x;
        if (t >= 0 && t < T) cwin = cum[(long)b * T + t];
    }
    float lk[KS_MAX];
#pragma unroll
    for (int j = 0; j < KS_MAX; ++j) lk[j] = (j < KS) ? c.loc_k[j * A_ + k] : 0.f;
    const float* keys = c.keys + ((long)b * T + t0) * A_ + k;
    float kv[TS / 2];
#pragma unroll
    for (int i = 0; i < TS / 2; ++i) kv[i] = (t0 + grp + 2 * i < T) ? keys[(long)(grp + 2 * i) * A_] : 0.f;
    const float qv = sum_parts<MSTTS_MAX_PARTS>(q, q_parts, q_pstride, (long)b * A_ + k);
    const float sb = c.score_b[k] + c.loc_b[k], wk = c.score_w[k];
    // ---- LDS phase
    if (threadIdx.x < TS + KS_MAX - 1) s_cum[threadIdx.x] = cwin;      // entries past the KS-tap window are zero (cwin == 0 there)
    if (q_sum && blockIdx.y == 0 && grp == 0) q_sum[(long)b * A_ + k] = qv;
    __syncthreads();
    const float qk = qv + sb;
#pragma unroll
    for (int i = 0; i < TS / 2; ++i) {
        const int tt = grp + 2 * i;
        float e = 0.f;
        if (t0 + tt < T) {
            float pre = kv[i] + qk;
#pragma unroll
            for (int j = 0; j < KS_MAX; ++j) pre += s_cum[tt + j] * lk[j];
            e = wk * fast_tanh(pre);
        }
        e = wave_sum(e);
        if ((threadIdx.x & 63) == 0) s_red[tt][(threadIdx.x >> 6) & 1] = e;
    }
    __syncthreads();
    if (threadIdx.x < TS && t0 + threadIdx.x < T)
        energy[(long)b * T + t0 + threadIdx.x] = s_red[threadIdx.x][0] + s_red[threadIdx.x][1];
}

// ---------------------------------------------------------------------------------------------
// forward: softmax + cumulative alignment + context
// ---------------------------------------------------------------------------------------------
constexpr int DS = 64;          // memory columns per workgroup
constexpr int T_MAX = 1024;
constexpr int VPRE = 8;         // value rows per thread preloaded before the softmax (covers T <= 128)

__global__ __launch_bounds__(256) void lsa_context_kernel(mstts_lsa_const c, const float* __restrict__ energy,
                                                          const float* __restrict__ cum, float* __restrict__ align,
                                                          float* __restrict__ cum_next, float* __restrict__ ctx, long ctx_ld,
                                                          float* __restrict__ ctx2, long ctx2_ld) {
    __shared__ float s_a[T_MAX];
    __shared__ float scratch[16];
    __shared__ __attribute__((aligned(16))) float s_part[16][DS];
    const int b = blockIdx.x, d0 = blockIdx.y * DS, T = (int)c.T, M = (int)c.M;
    const int len = c.lengths ? c.lengths[b] : T;
    const int c4 = threadIdx.x & 15, tg = threadIdx.x >> 4;
    const int col = d0 + c4 * 4;
    // ---- loads first: energies (+ cum for the writer block) and the first VPRE value rows of this thread
    float ev[T_MAX / 256], cv[T_MAX / 256];
#pragma unroll
    for (int i = 0; i < T_MAX / 256; ++i) {
        const int t = threadIdx.x + 256 * i;
        ev[i] = (t < len) ? energy[(long)b * T + t] : -INFINITY;
        cv[i] = (blockIdx.y == 0 && t < T) ? cum[(long)b * T + t] : 0.f;
    }
    const float* v = c.values + (long)b * T * M + col;
    float4 vv[VPRE];
#pragma unroll
    for (int i = 0; i < VPRE; ++i) {
        const int t = tg + 16 * i;
        vv[i] = (col < M && t < len) ? *reinterpret_cast<const float4*>(v + (long)t * M) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    // ---- softmax over the row
    float mx = -INFINITY;
#pragma unroll
    for (int i = 0; i < T_MAX / 256; ++i) mx = fmaxf(mx, ev[i]);
    mx = block_max(mx, scratch);
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < T_MAX / 256; ++i) {
        const int t = threadIdx.x + 256 * i;
        ev[i] = (t < len) ? __expf(ev[i] - mx) : 0.f;
        sum += ev[i];
    }
    sum = block_sum(sum, scratch);
    const float inv = 1.f / sum;
#pragma unroll
    for (int i = 0; i < T_MAX / 256; ++i) {
        const int t = threadIdx.x + 256 * i;
        if (t < T) {
            const float a = ev[i] * inv;
            s_a[t] = a;
            if (blockIdx.y == 0) {
                align[(long)b * T + t] = a;
                cum_next[(long)b * T + t] = cv[i] + a;
            }
        }
    }
    __syncthreads();
    // ---- context slice
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int i = 0; i < VPRE; ++i) {
        const int t = tg + 16 * i;
        if (t < len) {
            const float a = s_a[t];
            acc.x += a * vv[i].x; acc.y += a * vv[i].y; acc.z += a * vv[i].z; acc.w += a * vv[i].w;
        }
    }
    if (col < M) {
        for (int t = tg + 16 * VPRE; t < len; t += 16) {          // only for T > 128
            const float a = s_a[t];
            const float4 x = *reinterpret_cast<const float4*>(v + (long)t * M);
            acc.x += a * x.x; acc.y += a * x.y; acc.z += a * x.z; acc.w += a * x.w;
        }
    }
    *reinterpret_cast<float4*>(&s_part[tg][c4 * 4]) = acc;
    __syncthreads();
    if (threadIdx.x < DS && d0 + threadIdx.x < M) {
        float r = 0.f;
#pragma unroll
        for (int g = 0; g < 16; ++g) r += s_part[g][threadIdx.x];
        ctx[(long)b * ctx_ld + d0 + threadIdx.x] = r;
        if (ctx2) ctx2[(long)b * ctx2_ld + d0 + threadIdx.x] = r;
    }
}

// ---------------------------------------------------------------------------------------------
// forward, single launch: energies -> (in-launch exchange) -> softmax -> cumulative alignment -> context.
//
// 1-D grid of CS x B workgroups, all co-resident (<= 2 per CU at the reference shape), the workgroups of a row placed on one XCD
// (row_slice_of_block).  Workgroup
// cs computes the energies of its own slice of <= 16 encoder positions, publishes them as 8-byte {epoch, value}
// granules with relaxed agent-scope (write-through, sc1) stores, and gathers the rest of the row with relaxed
// agent-scope loads until every tag equals this launch's epoch - the data is the flag, no fence, no counter
// (MI355X hand-off form R2).  Then every workgroup runs the same T-float softmax and streams its own
// column slice of values[b].  The caller zeroes the granule buffer once per sequence and passes epoch =
// step + 1.  Every spin is bounded: a workgroup that gives up recomputes the missing energy itself
// (serially, slow but correct) and counts the event in the word after the last granule, so a launch can
// neither hang nor return a stale value.
// ---------------------------------------------------------------------------------------------
typedef __attribute__((address_space(1))) unsigned long long gu64;

// 1-D grid of B x nsl workgroups -> (row b, slice sl).  Block i runs on XCD i % 8 (observed, not guaranteed - used for speed only):
// with B % 8 == 0 every workgroup of row b gets an id == b (mod 8), so a row's keys / values / gradient slabs are fetched into ONE
// XCD's L2 (not eight) and its in-launch exchange stays inside that XCD; the XCD holds B/8 rows (1.8 MB at B = 32, T = 128).
__device__ __forceinline__ void row_slice_of_block(int id, int nsl, int B, int* b, int* sl) {
    if ((B & 7) == 0) {
        const int x = id & 7, r = id >> 3;
        *sl = r % nsl;
        *b = (r / nsl) * 8 + x;
    } else {
        *b = id / nsl;
        *sl = id - *b * nsl;
    }
}
constexpr int FS_THREADS = 512;
constexpr int FS_TSL = 16;        // encoder positions per workgroup (energy slice)
constexpr int FS_DSL = 96;        // memory columns per workgroup (context slice), at most
constexpr int FS_VPRE = 7;        // value rows per thread preloaded ahead of the exchange
constexpr unsigned FS_MAX_SPINS = 200000;

// (qb: row index into q - 0 when q is the workgroup's own LDS copy of the row's query)
__device__ __forceinline__ float lsa_energy_serial(const mstts_lsa_const& c, const float* q, int q_parts, long q_pstride,
                                                const float* cum, int qb, int t, int b) {
    const int T = (int)c.T, KS = (int)c.KS, pad = (KS - 1) / 2;
    float e = 0.f;
    for (int k = 0; k < A_; ++k) {
        float pre = c.keys[((long)b * T + t) * A_ + k] + c.score_b[k] + c.loc_b[k];
        for (int pp = 0; pp < q_parts; ++pp) pre += q[pp * q_pstride + (long)qb * A_ + k];
        for (int j = 0; j < KS; ++j) {
            const int tau = t + j - pad;
            if (tau >= 0 && tau < T) pre += cum[(long)b * T + tau] * c.loc_k[j * A_ + k];
        }
        e += c.score_w[k] * fast_tanh(pre);
    }
    return e;
}

// q[b, a] = m1[b, :] . Wq[:, a] recomputed by one thread (time-out path of the in-launch query projection)
__device__ __forceinline__ float lsa_q_serial(const float* m1, long m1_ld, const float* wq, int H, int b, int a, int q_bf16) {
    float q = 0.f;
    for (int j = 0; j < H; ++j) {
        float x = m1[(long)b * m1_ld + j], w = wq[(long)j * A_ + a];
        if (q_bf16) { x = (float)(__bf16)x; w = (float)(__bf16)w; }
        q += x * w;
    }
    return q;
}
constexpr int QJ = 8;             // hidden units per thread of the in-launch query projection: 128 chunks x 8 = H = 1024

// QIN: the query projection q = m1 . Wq runs inside this launch (it was a launch of its own, 4.9 us of almost pure fixed cost): the
// workgroup of slice cs computes the 16 units 16 cs .. 16 cs + 15 of its row's query (1/8 of the product: 64 KB of the kernel), the
// eight slices exchange them through a second granule array exactly like the energies, and the energy phase reads q from LDS.
// SELFTEST instantiation (mstts_lsa_step_fwd_selftest only): the workgroup of slice `skip` leaves at once, so the rest of its row must take
// the time-out path - the only way to exercise it, since on a healthy chip no workgroup ever times out
struct LsaQIn { const float* m1; long m1_ld; const float* wq; int H, bf16; };      // QIN operands: m1 rows [B, >= H], Wq [H, A] row-major
// PROJ (free-running decoder): the output projection [m1 | ctx] . Wp + bias also comes out of this launch, with no exchange of its own:
//   ctx . Wp_c = sum_t a[t] (values[t] . Wp_c) = sum_t a[t] vp[t]   - the projected values vp [B, T, NP] are loop invariants like the keys,
//   so once the row's energies are known slice s < 8 forms outputs 11 s .. 11 s + 10 from its own softmax weights (128 x 11 products);
//   m1 . Wp_m for those 11 outputs rides on the query projection (same m1 registers, a by-owner packed copy wp_own of the kernel rows,
//   mstts_lsa_proj_pack), reduced the same way while the query granules travel.
constexpr int PJ_OWN = 11, PJ_OW = 16, PJ_TG = 32, PJ_VPRE = 4;      // outputs per owner slice (16 slots: one 16-byte load per lane), position groups, prefetched positions per thread
struct LsaProj { const float* wp_own; const float* vp; const float* bias; int NP, NM; float* linear; float* stop; };
// PRE (with PROJ): the prenet of the NEXT decoder step (Modules.py:239-255, dropout always on) on the frame this step produces, in the
// same launch and with no hand-off beyond the two the attention has anyway.  Every slice knows the row's complete alignment after the
// energy exchange, so the context part of the WHOLE frame, sum_t a[t] vp[t, 0..NM], is 128 x 81 products any slice can do itself; what it
// lacks is m1 . Wp_m + b for the outputs of the other owners, and those 11 values per owner exist at the very start of the launch: they
// are published and gathered together with the query units.  Each owner slice then holds the row's frame, writes its own 11 outputs,
// computes the whole first prenet layer (NM x 256, its kernel rows requested during the softmax) and 32 of the 256 columns of the second.
constexpr int PR_P = 256, PR_K0 = 10, PR_K1 = 4, PR_GLD = 96;        // prenet width; first / second layer kernel rows per thread (16-byte loads); granules per row
constexpr int PF_Q = 21, PF_G = 24, PF_IT = 6;                        // whole-frame form: 16-byte quads of a projected-value row (NP == 84) x position groups x prefetched positions
// (every operand of these stages is fetched 16 bytes per lane: a wave's load costs the address unit 16 cycles whatever its width, and
// with one-word loads - 110 per wave in the first version of this kernel - that alone was 6 us of the launch)
struct LsaPre { const float* w0; const float* b0; const float* w1; const float* b1; const uint8_t* m0; const uint8_t* m1; float inv_keep;
                float* out; long out_ld; PackedDst out_p; unsigned long long* gf; };
// (forceinline like the other serial paths: a call would take the address of the kernel's by-value argument blocks and put them - and
// every later use of them - into scratch memory)
// m1 . Wp_m[:, o] + bias[o] by one thread (time-out path of the whole-frame form)
__device__ __forceinline__ float lsa_pm_serial(const LsaQIn& qi, const LsaProj& pj, int b, int o) {
    const int so = o / PJ_OWN, i = o % PJ_OWN;
    float acc = (pj.bias && o <= pj.NM) ? pj.bias[o] : 0.f;
    for (int j = 0; j < qi.H; ++j) acc += qi.m1[(long)b * qi.m1_ld + j] * pj.wp_own[((long)(so * QJ + (j & (QJ - 1))) * 128 + (j >> 3)) * PJ_OW + i];
    return acc;
}
template <bool SELFTEST, bool LKT = false, bool QIN = false, bool PROJ = false, bool PRE = false>
__global__ __launch_bounds__(FS_THREADS) void lsa_step_kernel(mstts_lsa_const c, const float* __restrict__ q, int q_parts, long q_pstride,
                                                              float* __restrict__ q_sum, const float* cum,
                                                              float* __restrict__ align, float* __restrict__ cum_next,
                                                              float* __restrict__ ctx, long ctx_ld, float* __restrict__ ctx2, long ctx2_ld, PackedDst ctx_p,
                                                              unsigned long long* gran, unsigned epoch, int tsl, int dsl, int ncs, int skip, LsaQIn qi,
                                                              LsaProj pjx, LsaPre prx) {
    int cs, b;
    row_slice_of_block(blockIdx.x, ncs, (int)c.B, &b, &cs);
    if (SELFTEST && cs == skip) return;
    __shared__ __attribute__((aligned(16))) float s_pq[PROJ ? 32 * PJ_OW : 4];                               // PROJ: per-row-of-16-lanes partial sums of m1 . Wp_m (own outputs)
    __shared__ float s_pm[PROJ ? PJ_OW : 1];                                    //       m1 . Wp_m of the own outputs
    __shared__ float s_pv[PROJ ? (PJ_TG + 1) * PJ_OW : 1];                      //       per-position-group partial sums of sum_t a[t] vp[t]
    __shared__ float s_fr[PRE ? PR_GLD : 1];                                    // PRE: the row's frame
    __shared__ float s_m[PRE ? PR_GLD : 1];                                     //      m1 . Wp_m + b of every output of the row
    __shared__ __attribute__((aligned(16))) float s_pvf[PRE ? PF_G * 4 * PF_Q : 4];   //      per-position-group partial sums of sum_t a[t] vp[t, :]
    __shared__ __attribute__((aligned(16))) float s_qp[QIN ? 32 * 16 : 4];     // QIN: per-row-of-16-lanes partial sums of the 16 own units
    __shared__ float s_q[QIN ? A_ : 1];                                         // QIN: the row's query
    __shared__ __attribute__((aligned(16))) float s_cum[FS_TSL + KS_MAX - 1 + 2];
    __shared__ float s_red[FS_TSL][2];
    __shared__ float s_e[T_MAX];
    __shared__ __attribute__((aligned(16))) float s_part[FS_THREADS * 4];
    const int T = (int)c.T, M = (int)c.M, KS = (int)c.KS, pad = (KS - 1) / 2;
    const int t0 = cs * tsl, d0 = cs * dsl;
    const int tid = threadIdx.x, lane = tid & 63;
    const int k = tid & (A_ - 1), tg = tid >> 7;                // energy role: unit k, positions t0 + 4*tg + {0..3}
    const int nc4 = dsl >> 2, ng = FS_THREADS / nc4;             // context role: float4 column c4, row group vg
    const int c4 = tid % nc4, vg = tid / nc4;
    const int col = d0 + c4 * 4;
    const bool vlive = vg < ng && col < M && c4 * 4 < dsl;
    const int len = c.lengths ? c.lengths[b] : T;
    // ---- every global load of the first phase, issued back to back
    float cwin = 0.f;
    if (tid < tsl + KS - 1) {
        const int t = t0 - pad + tid;
        if (t >= 0 && t < T) cwin = cum[(long)b * T + t];
    }
    float lk[KS_MAX];
    if constexpr (LKT) {                // by-unit copy of the filter: 8 float4 per lane instead of 31 single words (-0.3 us)
        float4 t4[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) t4[j] = reinterpret_cast<const float4*>(c.loc_kt + k * LKT_LD)[j];
#pragma unroll
        for (int j = 0; j < KS_MAX; ++j) lk[j] = reinterpret_cast<const float*>(t4)[j];
    } else {
#pragma unroll
        for (int j = 0; j < KS_MAX; ++j) lk[j] = (j < KS) ? c.loc_k[j * A_ + k] : 0.f;
    }
    float kv[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int tl = 4 * tg + i;
        // (unconditional, from a clamped position: the energy of a position outside the slice is computed and never read.  A predicated
        // load costs a branch, a zeroed default and - when the default's register still has an earlier load in flight - a wait in the
        // middle of the request phase)
        kv[i] = c.keys[((long)b * T + min(t0 + tl, T - 1)) * A_ + k];
    }
    float qv = 0.f, pre0[4];
    float4 qw[QIN ? QJ : 1], qm[QIN ? QJ / 4 : 1];
    if constexpr (QIN) {                 // thread (a4 = tid & 3, chunk = tid >> 2): units 16 cs + 4 a4 .. + 3, hidden units 8 chunk .. + 7
        const int a4 = tid & 3, ch = tid >> 2;
        const int qs = cs < 8 ? cs : 0;      // slices 0..7 own 16 query units each; further slices (T > 128) own none and only gather
#pragma unroll
        for (int jj = 0; jj < QJ; ++jj) qw[jj] = *reinterpret_cast<const float4*>(qi.wq + (long)(ch * QJ + jj) * A_ + 16 * qs + 4 * a4);
#pragma unroll
        for (int jj = 0; jj < QJ / 4; ++jj) qm[jj] = *reinterpret_cast<const float4*>(qi.m1 + (long)b * qi.m1_ld + ch * QJ + 4 * jj);
    } else {
        qv = sum_parts<MSTTS_MAX_PARTS>(q, q_parts, q_pstride, (long)b * A_ + k);
    }
    // PROJ operands, requested with everything else (owners only): the packed kernel rows of this thread's 8 hidden units x 3 outputs,
    // its share of the projected values, the bias
    float4 pw[PROJ ? QJ : 1], pvq[PRE ? PF_IT : 1];
    float pv[(PROJ && !PRE) ? PJ_VPRE : 1], pbias = 0.f;
    const int pol = PROJ ? (PRE ? tid % PF_Q : tid % PJ_OW) : 0, ptg = PROJ ? (PRE ? tid / PF_Q : tid / PJ_OW) : 0;
    const bool pv_live = PRE ? (cs < 8 && ptg < PF_G) : (PROJ && cs < 8 && ptg < PJ_TG && pol < PJ_OWN && PJ_OWN * cs + pol < pjx.NP);
    if constexpr (PROJ) {
        const int a4 = tid & 3, ch = tid >> 2;
#pragma unroll
        for (int jj = 0; jj < QJ; ++jj)
            pw[jj] = *reinterpret_cast<const float4*>(pjx.wp_own + ((long)((cs & 7) * QJ + jj) * 128 + ch) * PJ_OW + 4 * a4);   // (used under cs < 8)
        if constexpr (PRE) {                // whole frame: output pol of positions ptg + PF_G i
#pragma unroll
            for (int i = 0; i < PF_IT; ++i) {
                const int t = ptg + PF_G * i;
                pvq[i] = *reinterpret_cast<const float4*>(pjx.vp + ((long)b * T + min(t, T - 1)) * (4 * PF_Q) + 4 * pol);   // (used under pv_live && t < len)
            }
            const int ob = PJ_OWN * cs + tid - 32;                 // (threads 32..42 finish m1 . Wp_m + b of the own outputs)
            if (cs < 8 && tid >= 32 && tid < 32 + PJ_OWN && ob <= pjx.NM && pjx.bias) pbias = pjx.bias[ob];
        } else {
#pragma unroll
            for (int i = 0; i < PJ_VPRE; ++i) {
                const int t = ptg + PJ_TG * i;
                pv[i] = (pv_live && t < T) ? pjx.vp[((long)b * T + t) * pjx.NP + PJ_OWN * cs + pol] : 0.f;
            }
            const int ob = PJ_OWN * cs + tid - (FS_THREADS - PJ_OW);
            if (cs < 8 && tid >= FS_THREADS - PJ_OW && tid < FS_THREADS - PJ_OW + PJ_OWN && ob <= pjx.NM && pjx.bias) pbias = pjx.bias[ob];
        }
    }
    const float sb = c.score_b[k] + c.loc_b[k], wk = c.score_w[k];
    const float* v = c.values + (long)b * T * M + (vlive ? col : 0);
    float4 vv[FS_VPRE];
#pragma unroll
    for (int i = 0; i < FS_VPRE; ++i) {
        const int t = vg + ng * i;
        // (bounded by T, not by the row's length: `len` is itself a load, and the 12.6 MB of values must not wait for it - rows
        // between len and T are real memory and get a zero alignment below)
        vv[i] = *reinterpret_cast<const float4*>(v + (long)min(t, T - 1) * M);       // (used only under vlive && t < len)
    }
    // ---- own energy slice
    if (tid < FS_TSL + KS_MAX - 1 + 2) s_cum[tid] = cwin;          // entries past the window are zero
    if constexpr (QIN) {
        // own 16 units of the query: 8 hidden units per thread, then the 16 chunks of each 16-lane row through DPP row shifts, the 32
        // rows of the workgroup through LDS
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int jj = 0; jj < QJ; ++jj) {
            float x = reinterpret_cast<const float*>(qm)[jj];
            float4 w4 = qw[jj];
            if (qi.bf16) {
                x = (float)(__bf16)x;
                w4 = make_float4((float)(__bf16)w4.x, (float)(__bf16)w4.y, (float)(__bf16)w4.z, (float)(__bf16)w4.w);
            }
            acc.x += x * w4.x; acc.y += x * w4.y; acc.z += x * w4.z; acc.w += x * w4.w;
        }
        acc.x += dpp_mov<0x114, 0xf>(0.f, acc.x); acc.y += dpp_mov<0x114, 0xf>(0.f, acc.y);       // row_shr:4
        acc.z += dpp_mov<0x114, 0xf>(0.f, acc.z); acc.w += dpp_mov<0x114, 0xf>(0.f, acc.w);
        acc.x += dpp_mov<0x118, 0xf>(0.f, acc.x); acc.y += dpp_mov<0x118, 0xf>(0.f, acc.y);       // row_shr:8
        acc.z += dpp_mov<0x118, 0xf>(0.f, acc.z); acc.w += dpp_mov<0x118, 0xf>(0.f, acc.w);
        if ((tid & 15) >= 12) *reinterpret_cast<float4*>(&s_qp[(tid >> 4) * 16 + 4 * (tid & 3)]) = acc;   // lanes 12..15 of a row hold its sums
        if constexpr (PRE) {                                     // whole-frame form: m1 . Wp_m of the own outputs is handed over WITH the query
            if (cs < 8) {
                float a3[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int jj = 0; jj < QJ; ++jj) {
                    const float x = reinterpret_cast<const float*>(qm)[jj];
                    a3[0] += x * pw[jj].x; a3[1] += x * pw[jj].y; a3[2] += x * pw[jj].z; a3[3] += x * pw[jj].w;
                }
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    a3[i] += dpp_mov<0x114, 0xf>(0.f, a3[i]);
                    a3[i] += dpp_mov<0x118, 0xf>(0.f, a3[i]);
                }
                if ((tid & 15) >= 12) *reinterpret_cast<float4*>(&s_pq[(tid >> 4) * PJ_OW + 4 * (tid & 3)]) = make_float4(a3[0], a3[1], a3[2], a3[3]);
            }
        }
        __syncthreads();
        gu64* gq = (gu64*)(gran + (long)c.B * T + 1 + (long)b * A_);
        if (tid < 16 && cs < 8) {
            float qa = 0.f;
#pragma unroll
            for (int r = 0; r < 32; ++r) qa += s_qp[r * 16 + tid];
            s_q[16 * cs + tid] = qa;
            __hip_atomic_store(gq + 16 * cs + tid, ((unsigned long long)epoch << 32) | __float_as_uint(qa), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        // the part of the pre-activation that does not need the query (keys + filter . cumulative alignment) while the granules travel
        {
            float cw[4 + KS_MAX - 1];
#pragma unroll
            for (int i = 0; i < 4 + KS_MAX - 1; ++i) cw[i] = s_cum[4 * tg + i];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                float pre = kv[i] + sb;
#pragma unroll
                for (int j = 0; j < KS_MAX; ++j) pre += cw[i + j] * lk[j];
                pre0[i] = pre;
            }
        }
        if constexpr (PROJ && !PRE) {                            // m1 . Wp_m for the own outputs, reduced like the query units
            if (cs < 8) {
                float a3[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int jj = 0; jj < QJ; ++jj) {
                    const float x = reinterpret_cast<const float*>(qm)[jj];
                    a3[0] += x * pw[jj].x; a3[1] += x * pw[jj].y; a3[2] += x * pw[jj].z; a3[3] += x * pw[jj].w;
                }
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    a3[i] += dpp_mov<0x114, 0xf>(0.f, a3[i]);
                    a3[i] += dpp_mov<0x118, 0xf>(0.f, a3[i]);
                }
                if ((tid & 15) >= 12) *reinterpret_cast<float4*>(&s_pq[(tid >> 4) * PJ_OW + 4 * (tid & 3)]) = make_float4(a3[0], a3[1], a3[2], a3[3]);
            }
        }
        if constexpr (PRE) {
            gu64* gm = (gu64*)(prx.gf + (long)b * PR_GLD);
            if (cs < 8 && tid >= 32 && tid < 32 + PJ_OWN && PJ_OWN * cs + tid - 32 <= pjx.NM) {
                float a = pbias;
#pragma unroll
                for (int r = 0; r < 32; ++r) a += s_pq[r * PJ_OW + tid - 32];
                s_m[PJ_OWN * cs + tid - 32] = a;
                __hip_atomic_store(gm + PJ_OWN * cs + tid - 32, ((unsigned long long)epoch << 32) | __float_as_uint(a), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            const int o = tid - A_;
            if (cs < 8 && o >= 0 && o <= pjx.NM && o / PJ_OWN != cs) {
                unsigned long long x = __hip_atomic_load(gm + o, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                unsigned spins = 0;
                while ((unsigned)(x >> 32) != epoch && spins < FS_MAX_SPINS) {
                    __builtin_amdgcn_s_sleep(1);
                    x = __hip_atomic_load(gm + o, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    ++spins;
                }
                float a;
                if ((unsigned)(x >> 32) == epoch) a = __uint_as_float((unsigned)x);
                else {
                    a = lsa_pm_serial(qi, pjx, b, o);
                    atomicAdd(gran + (long)c.B * T, 1ull);
                }
                s_m[o] = a;
            }
        }
        if (tid < A_ && (tid >> 4) != cs) {                      // the other slices' units (the data is the flag)
            unsigned long long x = __hip_atomic_load(gq + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            unsigned spins = 0;
            while ((unsigned)(x >> 32) != epoch && spins < FS_MAX_SPINS) {
                __builtin_amdgcn_s_sleep(1);
                x = __hip_atomic_load(gq + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                ++spins;
            }
            float qa;
            if ((unsigned)(x >> 32) == epoch) qa = __uint_as_float((unsigned)x);
            else {
                qa = lsa_q_serial(qi.m1, qi.m1_ld, qi.wq, qi.H, b, tid, qi.bf16);
                atomicAdd(gran + (long)c.B * T, 1ull);
            }
            s_q[tid] = qa;
        }
        __syncthreads();
        qv = s_q[k];
        if constexpr (PROJ && !PRE) {
            if (cs < 8 && tid >= FS_THREADS - PJ_OW) {
                float a = 0.f;
#pragma unroll
                for (int r = 0; r < 32; ++r) a += s_pq[r * PJ_OW + tid - (FS_THREADS - PJ_OW)];
                s_pm[tid - (FS_THREADS - PJ_OW)] = a;              // (read back by the same thread in the tail)
            }
        }
    }
    if (q_sum && cs == 0 && tg == 0) q_sum[(long)b * A_ + k] = qv;
    if constexpr (!QIN) {
        __syncthreads();
        float cw[4 + KS_MAX - 1];
#pragma unroll
        for (int i = 0; i < 4 + KS_MAX - 1; ++i) cw[i] = s_cum[4 * tg + i];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            float pre = kv[i] + sb;
#pragma unroll
            for (int j = 0; j < KS_MAX; ++j) pre += cw[i + j] * lk[j];
            pre0[i] = pre;
        }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        float e = wave_sum(wk * fast_tanh(pre0[i] + qv));
        if (lane == 0) s_red[4 * tg + i][(tid >> 6) & 1] = e;
    }
    __syncthreads();
    gu64* g = (gu64*)(gran + (long)b * T);
    if (tid < tsl && t0 + tid < T) {
        const float e = s_red[tid][0] + s_red[tid][1];
        s_e[t0 + tid] = e;
        __hip_atomic_store(g + t0 + tid, ((unsigned long long)epoch << 32) | __float_as_uint(e), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    // ---- gather the other slices of the row (the data is the flag)
    for (int t = tid; t < T; t += FS_THREADS) {
        if (t >= t0 && t < t0 + tsl) continue;
        unsigned long long x = __hip_atomic_load(g + t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        unsigned spins = 0;
        while ((unsigned)(x >> 32) != epoch && spins < FS_MAX_SPINS) {
            __builtin_amdgcn_s_sleep(1);
            x = __hip_atomic_load(g + t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            ++spins;
        }
        float e;
        if ((unsigned)(x >> 32) == epoch) e = __uint_as_float((unsigned)x);
        else {
            e = QIN ? lsa_energy_serial(c, s_q, 1, 0, cum, 0, t, b) : lsa_energy_serial(c, q, q_parts, q_pstride, cum, b, t, b);
            atomicAdd(gran + (long)c.B * T, 1ull);
        }
        s_e[t] = e;
    }
    __syncthreads();
    // ---- softmax statistics, redundantly per wave (no further block-wide reduction)
    float mx = -INFINITY;
    for (int t = lane; t < len; t += 64) mx = fmaxf(mx, s_e[t]);
    mx = wave_max(mx);
    float sum = 0.f;
    for (int t = lane; t < len; t += 64) sum += __expf(s_e[t] - mx);
    sum = wave_sum(sum);
    const float inv = 1.f / sum;
    if (tid < tsl && t0 + tid < T) {
        const int t = t0 + tid;
        const float a = (t < len) ? __expf(s_e[t] - mx) * inv : 0.f;
        align[(long)b * T + t] = a;
        cum_next[(long)b * T + t] = s_cum[pad + tid] + a;
    }
    // PROJ: the frame before the context phase (it does not depend on the context)
    if constexpr (PRE) {                 // whole-frame form: every owner slice forms all NM + 1 outputs of the row
        if (pv_live) {
            float4 pa = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
            for (int i = 0; i < PF_IT; ++i) {
                const int t = ptg + PF_G * i;
                if (t < len) {
                    const float a = __expf(s_e[t] - mx) * inv;
                    pa.x += a * pvq[i].x; pa.y += a * pvq[i].y; pa.z += a * pvq[i].z; pa.w += a * pvq[i].w;
                }
            }
            for (int t = ptg + PF_G * PF_IT; t < len; t += PF_G) {
                const float a = __expf(s_e[t] - mx) * inv;
                const float4 x = *reinterpret_cast<const float4*>(pjx.vp + ((long)b * T + t) * (4 * PF_Q) + 4 * pol);
                pa.x += a * x.x; pa.y += a * x.y; pa.z += a * x.z; pa.w += a * x.w;
            }
            *reinterpret_cast<float4*>(&s_pvf[(ptg * PF_Q + pol) * 4]) = pa;
        }
        __syncthreads();
        if (cs < 8 && tid <= pjx.NM) {
            float f = s_m[tid];
#pragma unroll
            for (int gq = 0; gq < PF_G; ++gq) f += s_pvf[gq * 4 * PF_Q + tid];
            s_fr[tid] = f;                                       // (identical in every slice of the row: same operands, same order)
            if (tid / PJ_OWN == cs) {
                if (tid < pjx.NM) pjx.linear[(long)b * pjx.NM + tid] = f;
                else pjx.stop[b] = f;
            }
        }
    } else if constexpr (PROJ) {
        if (cs < 8 && ptg < PJ_TG) {
            float pa = 0.f;
#pragma unroll
            for (int i = 0; i < PJ_VPRE; ++i) {
                const int t = ptg + PJ_TG * i;
                if (t < len) pa += __expf(s_e[t] - mx) * inv * pv[i];
            }
            if (pv_live)
                for (int t = ptg + PJ_TG * PJ_VPRE; t < len; t += PJ_TG)
                    pa += __expf(s_e[t] - mx) * inv * pjx.vp[((long)b * T + t) * pjx.NP + PJ_OWN * cs + pol];
            s_pv[ptg * PJ_OW + pol] = pa;
        }
        __syncthreads();
        const int ol = tid - (FS_THREADS - PJ_OW), oo = PJ_OWN * cs + ol;
        if (cs < 8 && ol >= 0 && ol < PJ_OWN && oo < pjx.NP) {
            float v2 = pbias + s_pm[ol];
            for (int gq = 0; gq < PJ_TG; ++gq) v2 += s_pv[gq * PJ_OW + ol];
            if (oo < pjx.NM) pjx.linear[(long)b * pjx.NM + oo] = v2;
            else if (oo == pjx.NM) pjx.stop[b] = v2;
        }
    }
    // PRE: operands of the next step's prenet, requested here so that they arrive under the context phase
    float4 w0r[PRE ? PR_K0 : 1], w1r[PRE ? PR_K1 : 1];
    float b0r = 0.f, b1r = 0.f, m0r = 0.f, m1r = 0.f;
    if constexpr (PRE) {
        if (cs < 8) {
            // first layer: thread (column quad cq0 = tid & 63, row group kg0 = tid >> 6) takes rows 10 kg0 .. + 9; second layer (this slice's
            // 32 columns): thread (column quad cq1 = tid & 7, row group kg1 = tid >> 3) takes rows 4 kg1 .. + 3
            const int cq0 = tid & 63, kg0 = tid >> 6, cq1 = tid & 7, kg1 = tid >> 3;
#pragma unroll
            for (int i = 0; i < PR_K0; ++i)
                w0r[i] = *reinterpret_cast<const float4*>(prx.w0 + (long)min(kg0 * PR_K0 + i, pjx.NM - 1) * PR_P + 4 * cq0);   // (rows past NM meet a zero factor)
#pragma unroll
            for (int i = 0; i < PR_K1; ++i) w1r[i] = *reinterpret_cast<const float4*>(prx.w1 + (long)(kg1 * PR_K1 + i) * PR_P + 32 * cs + 4 * cq1);
            if (tid < PR_P) { b0r = prx.b0[tid]; m0r = (float)prx.m0[(long)b * PR_P + tid]; }
            if (tid < 32) { b1r = prx.b1[32 * cs + tid]; m1r = (float)prx.m1[(long)b * PR_P + 32 * cs + tid]; }
        }
        __builtin_amdgcn_sched_barrier(0);       // (keeps the requests here: sunk to their uses they would cost the tail a round trip)
    }
    // ---- context slice
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int i = 0; i < FS_VPRE; ++i) {
        const int t = vg + ng * i;
        if (vlive && t < len) {
            const float a = __expf(s_e[t] - mx) * inv;
            acc.x += a * vv[i].x; acc.y += a * vv[i].y; acc.z += a * vv[i].z; acc.w += a * vv[i].w;
        }
    }
    if (vlive) {
        for (int t = vg + ng * FS_VPRE; t < len; t += ng) {
            const float a = __expf(s_e[t] - mx) * inv;
            const float4 x = *reinterpret_cast<const float4*>(v + (long)t * M);
            acc.x += a * x.x; acc.y += a * x.y; acc.z += a * x.z; acc.w += a * x.w;
        }
    }
    if (vg < ng) *reinterpret_cast<float4*>(&s_part[(vg * nc4 + c4) * 4]) = acc;
    __syncthreads();
    if (tid < dsl && d0 + tid < M) {
        float r = 0.f;
        for (int gq = 0; gq < ng; ++gq) r += s_part[gq * dsl + tid];
        ctx[(long)b * ctx_ld + d0 + tid] = r;
        if (ctx2) ctx2[(long)b * ctx2_ld + d0 + tid] = r;
        if (ctx_p.base) packed_store(ctx_p, b, d0 + tid, r);
    }
    if constexpr (PRE) {
        if (cs >= 8) return;
        __syncthreads();                 // (every context reduction above has read s_part, which the prenet now reuses; s_fr is long complete)
        float* s_l0 = s_part;            // [8][256] first-layer partial sums by row group
        float* s_h = s_pq;               // [256]   (the query-phase scratch is long free)
        float* s_l1 = s_part;            // [64][32] second-layer partial sums by row group (after the first layer has been reduced)
        {
            const int cq0 = tid & 63, kg0 = tid >> 6;
            float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
            for (int i = 0; i < PR_K0; ++i) {
                const float x = (kg0 * PR_K0 + i < pjx.NM) ? s_fr[kg0 * PR_K0 + i] : 0.f;
                a.x += x * w0r[i].x; a.y += x * w0r[i].y; a.z += x * w0r[i].z; a.w += x * w0r[i].w;
            }
            *reinterpret_cast<float4*>(&s_l0[kg0 * PR_P + 4 * cq0]) = a;
        }
        __syncthreads();
        if (tid < PR_P) {
            float a = b0r;
#pragma unroll
            for (int kg = 0; kg < FS_THREADS / 64; ++kg) a += s_l0[kg * PR_P + tid];
            s_h[tid] = fmaxf(a, 0.f) * (fminf(m0r, 1.f) * prx.inv_keep);
        }
        __syncthreads();
        {
            const int cq1 = tid & 7, kg1 = tid >> 3;
            float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
            for (int i = 0; i < PR_K1; ++i) {
                const float x = s_h[kg1 * PR_K1 + i];
                a.x += x * w1r[i].x; a.y += x * w1r[i].y; a.z += x * w1r[i].z; a.w += x * w1r[i].w;
            }
            *reinterpret_cast<float4*>(&s_l1[kg1 * 32 + 4 * cq1]) = a;
        }
        __syncthreads();
        if (tid < 32) {
            float a = b1r;
#pragma unroll
            for (int kg = 0; kg < FS_THREADS / 8; ++kg) a += s_l1[kg * 32 + tid];
            const float y = fmaxf(a, 0.f) * (fminf(m1r, 1.f) * prx.inv_keep);
            prx.out[(long)b * prx.out_ld + 32 * cs + tid] = y;
            if (prx.out_p.base) packed_store(prx.out_p, b, 32 * cs + tid, y);
        }
    }
}

// ---------------------------------------------------------------------------------------------
// backward: d_align.  G = dL/d cum_{s+1} = G_next + filter^T applied to the next step's energy gradient,
// which the denergy kernel of step s+1 left as h_next[b,t,j] = sum_k g[t,k] loc_k[j,k]:
//     G[t] = G_next[t] + sum_j h_next[t + pad - j][j]
// ---------------------------------------------------------------------------------------------
constexpr int HLD = 32;        // row stride of h (KS_MAX = 31 taps, padded)
constexpr int MROW = 4;        // float4 per lane per value row held in registers (M <= 1024)

__global__ __launch_bounds__(256) void lsa_dalign_kernel(mstts_lsa_const c, const float* __restrict__ d_ctx, long d_ctx_ld,
                                                         const float* __restrict__ d_ctx2, long d_ctx2_ld, int d_ctx2_parts, long d_ctx2_pstride,
                                                         const float* __restrict__ G_next, const float* __restrict__ h_next,
                                                         float* __restrict__ G, float* __restrict__ d_align) {
    __shared__ float s_g[TS];
    const int b = blockIdx.x, t0 = blockIdx.y * TS, T = (int)c.T, M = (int)c.M, KS = (int)c.KS, pad = (KS - 1) / 2;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int len = c.lengths ? c.lengths[b] : T;
    // ---- loads first
    constexpr int LPR = 256 / TS;                    // lanes per row (32): lane j of row tt takes tap j
    const int tt_h = threadIdx.x / LPR, jh = threadIdx.x % LPR;
    float hv = 0.f;
    if (h_next && jh < KS) {
        const int tau = t0 + tt_h + pad - jh;
        if (tau >= 0 && tau < T) hv = h_next[((long)b * T + tau) * HLD + jh];
    }
    float gn = 0.f;
    if (threadIdx.x < TS && G_next && t0 + threadIdx.x < T) gn = G_next[(long)b * T + t0 + threadIdx.x];
    float4 dcv[MROW], val[TS / 4][MROW];
    const float* dc = d_ctx + (long)b * d_ctx_ld;
#pragma unroll
    for (int m = 0; m < MROW; ++m) {
        const int i = lane * 4 + 256 * m;
        float4 y = make_float4(0.f, 0.f, 0.f, 0.f);
        if (i < M) {
            y = *reinterpret_cast<const float4*>(dc + i);
            if (d_ctx2) {
                float4 y2[8];
#pragma unroll
                for (int pp = 0; pp < 8; ++pp)
                    y2[pp] = (pp == 0 || pp < d_ctx2_parts) ? *reinterpret_cast<const float4*>(d_ctx2 + pp * d_ctx2_pstride + (long)b * d_ctx2_ld + i)
                                                            : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
                for (int pp = 0; pp < 8; ++pp) { y.x += y2[pp].x; y.y += y2[pp].y; y.z += y2[pp].z; y.w += y2[pp].w; }
            }
        }
        dcv[m] = y;
    }
#pragma unroll
    for (int r = 0; r < TS / 4; ++r) {
        const int t = t0 + w + 4 * r;
#pragma unroll
        for (int m = 0; m < MROW; ++m) {
            const int i = lane * 4 + 256 * m;
            val[r][m] = (t < len && i < M) ? *reinterpret_cast<const float4*>(c.values + ((long)b * T + t) * M + i)
                                          : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    }
    // ---- G for the TS rows: 32-lane sums of the diagonal taps
    static_assert(LPR == 32, "half_sum_in_last_lane sums 32-lane groups");
    hv = half_sum_in_last_lane(hv);
    if (jh == LPR - 1) s_g[tt_h] = hv;
    __syncthreads();
    if (threadIdx.x < TS) s_g[threadIdx.x] += gn;
    __syncthreads();
    // ---- values . d_ctx for this wave's rows
#pragma unroll
    for (int r = 0; r < TS / 4; ++r) {
        const int tt = w + 4 * r, t = t0 + tt;
        float acc = 0.f;
#pragma unroll
        for (int m = 0; m < MROW; ++m)
            acc += val[r][m].x * dcv[m].x + val[r][m].y * dcv[m].y + val[r][m].z * dcv[m].z + val[r][m].w * dcv[m].w;
        if (t < len) {                                  // M > 1024: remaining columns
            for (int i = lane * 4 + 256 * MROW; i < M; i += 256) {
                const float4 x = *reinterpret_cast<const float4*>(c.values + ((long)b * T + t) * M + i);
                float4 y = *reinterpret_cast<const float4*>(dc + i);
                if (d_ctx2)
                    for (int pp = 0; pp < max(d_ctx2_parts, 1); ++pp) {
                        const float4 y2 = *reinterpret_cast<const float4*>(d_ctx2 + pp * d_ctx2_pstride + (long)b * d_ctx2_ld + i);
                        y.x += y2.x; y.y += y2.y; y.z += y2.z; y.w += y2.w;
                    }
                acc += x.x * y.x + x.y * y.y + x.z * y.z + x.w * y.w;
            }
        }
        acc = wave_sum(acc);
        if (lane == 0 && t < T) {
            const float g = s_g[tt];
            G[(long)b * T + t] = g;
            d_align[(long)b * T + t] = g + acc;
        }
    }
}

// ---------------------------------------------------------------------------------------------
// backward: d_energy, d_query, and h[t,j] = sum_k g[t,k] loc_k[j,k] for the previous step's d_align
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void lsa_denergy_kernel(mstts_lsa_const c, const float* __restrict__ align,
                                                          const float* __restrict__ d_align, const float* __restrict__ q,
                                                          const float* __restrict__ cum, float* __restrict__ d_e_out,
                                                          float* __restrict__ dq, float* __restrict__ h) {
    __shared__ float s_cum[TS + KS_MAX - 1];
    __shared__ __attribute__((aligned(16))) float s_g[TS][A_];
    __shared__ __attribute__((aligned(16))) float s_lk[KS_MAX + 1][A_ + 4];   // +4: rows land on different 16-byte slots for the b128 reads
    __shared__ float s_de[TS];
    __shared__ float s_dq[A_];
    __shared__ float scratch[16];
    const int b = blockIdx.x, t0 = blockIdx.y * TS, T = (int)c.T, KS = (int)c.KS, pad = (KS - 1) / 2;
    const int k = threadIdx.x & (A_ - 1), grp = threadIdx.x >> 7;
    // ---- loads first
    float av[T_MAX / 256], dav[T_MAX / 256];
#pragma unroll
    for (int i = 0; i < T_MAX / 256; ++i) {
        const int t = threadIdx.x + 256 * i;
        av[i] = (t < T) ? align[(long)b * T + t] : 0.f;
        dav[i] = (t < T) ? d_align[(long)b * T + t] : 0.f;
    }
    float a_own = 0.f, da_own = 0.f;
    if (threadIdx.x < TS && t0 + threadIdx.x < T) {
        a_own = align[(long)b * T + t0 + threadIdx.x];
        da_own = d_align[(long)b * T + t0 + threadIdx.x];
    }
    float cwin = 0.f;
    if (threadIdx.x < TS + KS - 1) {
        const int t = t0 - pad + threadIdx.x;
        if (t >= 0 && t < T) cwin = cum[(long)b * T + t];
    }
    constexpr int NLK = (KS_MAX * A_ + 255) / 256;      // loc_k staged for the h product (and read back per k)
    float lkv[NLK];
#pragma unroll
    for (int i = 0; i < NLK; ++i) {
        const int e = threadIdx.x + 256 * i;
        lkv[i] = (e < KS * A_) ? c.loc_k[e] : 0.f;
    }
    const float* keys = c.keys + ((long)b * T + t0) * A_ + k;
    float kv[TS / 2];
#pragma unroll
    for (int i = 0; i < TS / 2; ++i) kv[i] = (t0 + grp + 2 * i < T) ? keys[(long)(grp + 2 * i) * A_] : 0.f;
    const float qk = q[(long)b * A_ + k] + c.score_b[k] + c.loc_b[k];
    const float wk = c.score_w[k];
    // ---- softmax backward needs the whole row's dot(a, d_a)
    float dot = 0.f;
#pragma unroll
    for (int i = 0; i < T_MAX / 256; ++i) dot += av[i] * dav[i];
    dot = block_sum(dot, scratch);
    if (threadIdx.x < TS) {
        const int t = t0 + threadIdx.x;
        const float de = (t < T) ? a_own * (da_own - dot) : 0.f;
        if (t < T) d_e_out[(long)b * T + t] = de;
        s_de[threadIdx.x] = de;
    }
#pragma unroll
    for (int i = 0; i < NLK; ++i) {
        const int e = threadIdx.x + 256 * i;
        if (e < KS_MAX * A_) s_lk[e / A_][e % A_] = lkv[i];
    }
    if (threadIdx.x < TS + KS_MAX - 1) s_cum[threadIdx.x] = cwin;      // entries past the KS-tap window are zero (cwin == 0 there)
    __syncthreads();
    float lk[KS_MAX];
#pragma unroll
    for (int j = 0; j < KS_MAX; ++j) lk[j] = s_lk[j][k];
    float dq_acc = 0.f;
#pragma unroll
    for (int i = 0; i < TS / 2; ++i) {
        const int tt = grp + 2 * i;
        float g = 0.f;
        if (t0 + tt < T) {
            float pre = kv[i] + qk;
#pragma unroll
            for (int j = 0; j < KS_MAX; ++j) pre += s_cum[tt + j] * lk[j];
            const float u = fast_tanh(pre);
            g = s_de[tt] * wk * (1.f - u * u);
        }
        s_g[tt][k] = g;
        dq_acc += g;
    }
    if (grp == 1) s_dq[k] = dq_acc;
    __syncthreads();
    if (grp == 0) atomicAdd(dq + (long)b * A_ + k, dq_acc + s_dq[k]);
    // h[tt][j] = sum_k g[tt][k] * loc_k[j][k] : thread (tt = tid / 32, j = tid % 32)
    {
        const int tt = threadIdx.x >> 5, j = threadIdx.x & 31;
        float acc = 0.f;
        if (j < KS) {
#pragma unroll 8
            for (int kk = 0; kk < A_; kk += 4) {
                const float4 g4 = *reinterpret_cast<const float4*>(&s_g[tt][kk]);
                const float4 l4 = *reinterpret_cast<const float4*>(&s_lk[j][kk]);
                acc += g4.x * l4.x + g4.y * l4.y + g4.z * l4.z + g4.w * l4.w;
            }
        }
        if (t0 + tt < T) h[((long)b * T + t0 + tt) * HLD + j] = acc;
    }
}

// ---------------------------------------------------------------------------------------------
// backward, single launch: the two kernels above in one, grid (T/TS, B) so that the workgroups of a row have adjacent
// block ids.  The only quantity a workgroup needs from the rest of its row is the softmax-backward scalar
// dot(a, d_a); each workgroup publishes its slice's partial as ONE {epoch,value} granule right after the d_align
// phase, recomputes its tanh terms while the granules travel, then gathers the T/TS partials (relaxed agent-scope
// loads, bounded spin).  d_align never goes to memory.  A workgroup that times out recomputes the missing partial itself
// (serially: slow but correct, like the forward kernel) and counts the event in the word after the last granule, so a launch can
// neither hang nor poison the gradients.
// ---------------------------------------------------------------------------------------------
// ---------------------------------------------------------------------------------------------
// backward, single launch, NO exchange: the two kernels above in one, grid B x T/TS workgroups.  The only row-wide quantity of the
// softmax backward is dot(a, d_a), and with d_a = G + values . d_ctx it splits into
//     dot(a, d_a) = dot(a, G) + (sum_t a[t] values[t]) . d_ctx = dot(a, G) + ctx . d_ctx
// where ctx is the FORWARD context of this step (kept in the projection history) - so every workgroup forms the scalar itself from
// 128 + 768 floats it can read directly (plus the row's G, a 128 x 31 filter-transpose sum it recomputes redundantly), and nothing
// has to cross workgroups inside the launch.  d_align never goes to memory.  (The exchanged form of round 1 cost 13.5 us per step.)
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void lsa_step_bwd_kernel(mstts_lsa_const c, const float* __restrict__ d_ctx, long d_ctx_ld,
                                                           const float* __restrict__ d_ctx2, long d_ctx2_ld, int d_ctx2_parts, long d_ctx2_pstride,
                                                           const float* __restrict__ G_next, const float* __restrict__ h_next, float* __restrict__ G,
                                                           const float* __restrict__ align, const float* __restrict__ q, const float* __restrict__ cum,
                                                           const float* __restrict__ ctx_fwd, long ctx_fwd_ld,
                                                           float* __restrict__ d_e_out, float* __restrict__ dq, float* __restrict__ h, int nsl) {
    int sl, b;
    row_slice_of_block(blockIdx.x, nsl, (int)c.B, &b, &sl);
    __shared__ float s_G[TS];                                 // G of this slice's positions
    __shared__ float s_a[T_MAX];                              // the row's alignments
    __shared__ __attribute__((aligned(16))) float s_dc[256 * MROW];  // the row's total d_ctx (first 1024 columns)
    __shared__ float s_da[TS];
    __shared__ float s_cum[TS + KS_MAX - 1];
    __shared__ __attribute__((aligned(16))) float s_g[TS][A_];
    __shared__ __attribute__((aligned(16))) float s_lk[KS_MAX + 1][A_ + 4];
    __shared__ float s_de[TS];
    __shared__ float s_dq[A_];
    __shared__ float scratch[16];
    __shared__ float s_dot2;
    __shared__ float s_gn[TS];
    const int t0 = sl * TS, T = (int)c.T, M = (int)c.M, KS = (int)c.KS, pad = (KS - 1) / 2;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int k = threadIdx.x & (A_ - 1), grp = threadIdx.x >> 7;
    const int len = c.lengths ? c.lengths[b] : T;
    // ---- every global load, back to back.  First the small operands of the tanh recompute (window, folded filter, this slice's
    // keys, query): loads return in order, so with these ahead of the 12.6 MB of value rows the recompute below runs while the
    // values are still in flight ...
    float cwin = 0.f;
    if (threadIdx.x < TS + KS - 1) {
        const int t = t0 - pad + threadIdx.x;
        if (t >= 0 && t < T) cwin = cum[(long)b * T + t];
    }
    constexpr int NLK4 = (KS_MAX * A_ / 4 + 255) / 256;   // the filter as float4: 4 loads per thread instead of 16 words
    float4 lkv[NLK4];
#pragma unroll
    for (int i = 0; i < NLK4; ++i) {
        const int e4 = threadIdx.x + 256 * i;
        lkv[i] = (e4 * 4 < KS * A_) ? reinterpret_cast<const float4*>(c.loc_k)[e4] : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    const float* keys = c.keys + ((long)b * T + t0) * A_ + k;
    float kv[TS / 2];
#pragma unroll
    for (int i = 0; i < TS / 2; ++i) kv[i] = (t0 + grp + 2 * i < T) ? keys[(long)(grp + 2 * i) * A_] : 0.f;
    const float qk = q[(long)b * A_ + k] + c.score_b[k] + c.loc_b[k];
    const float wk = c.score_w[k];
    // ... the row's G operands ...
    constexpr int LPR = 256 / TS;                             // 32 lanes per position: lane j of position tt takes tap j
    const int tt_h = threadIdx.x / LPR, jh = threadIdx.x % LPR;
    float hv = 0.f;                                           // this slice's G: one diagonal tap per lane
    if (h_next && jh < KS) {
        const int tau = t0 + tt_h + pad - jh;
        if (tau >= 0 && tau < T) hv = h_next[((long)b * T + tau) * HLD + jh];
    }
    // the whole h_next[b] tile, coalesced (for dot(a, G) only): rows of HLD = 32 floats, 8 float4 per row, HT4 float4 per thread
    constexpr int HT4 = 4;                                    // covers T <= 128; longer rows loop below
    float4 ht[HT4];
#pragma unroll
    for (int i = 0; i < HT4; ++i) {
        const int e = threadIdx.x + 256 * i;                  // float4 index in the tile: row e / 8, taps 4 (e % 8) ..
        ht[i] = (h_next && (e >> 3) < T) ? reinterpret_cast<const float4*>(h_next + (long)b * T * HLD)[e] : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    float av[T_MAX / 256], gnv[T_MAX / 256];
#pragma unroll
    for (int i = 0; i < T_MAX / 256; ++i) {
        const int t = threadIdx.x + 256 * i;
        av[i] = (t < T) ? align[(long)b * T + t] : 0.f;
        gnv[i] = (t < T && G_next) ? G_next[(long)b * T + t] : 0.f;
    }
    const float a_own = (threadIdx.x < TS && t0 + threadIdx.x < T) ? align[(long)b * T + t0 + threadIdx.x] : 0.f;
    const float gn_own = (threadIdx.x < TS && t0 + threadIdx.x < T && G_next) ? G_next[(long)b * T + t0 + threadIdx.x] : 0.f;
    // ... d_ctx (+ its partial slabs: wave m sums chunk m of the row once for the whole workgroup, shared through LDS below - every
    // wave loading all nine vectors itself was 36 float4 per lane and most of this kernel's L2 traffic), the forward context, this
    // slice's value rows ...
    float4 dcv[MROW], cxv[MROW], val[TS / 4][MROW];
    const float* dc = d_ctx + (long)b * d_ctx_ld;
    float4 dsum = make_float4(0.f, 0.f, 0.f, 0.f);
    {
        const int i = lane * 4 + 256 * w;                     // MROW == 4 == waves per workgroup: wave w owns columns [256 w, 256 w + 256)
        if (i < M) {
            dsum = *reinterpret_cast<const float4*>(dc + i);
            if (d_ctx2) {
                float4 y2[8];
#pragma unroll
                for (int pp = 0; pp < 8; ++pp)
                    y2[pp] = (pp == 0 || pp < d_ctx2_parts) ? *reinterpret_cast<const float4*>(d_ctx2 + pp * d_ctx2_pstride + (long)b * d_ctx2_ld + i)
                                                            : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
                for (int pp = 0; pp < 8; ++pp) { dsum.x += y2[pp].x; dsum.y += y2[pp].y; dsum.z += y2[pp].z; dsum.w += y2[pp].w; }
            }
        }
    }
#pragma unroll
    for (int m = 0; m < MROW; ++m) {
        const int i = lane * 4 + 256 * m;
        cxv[m] = (w == 0 && i < M) ? *reinterpret_cast<const float4*>(ctx_fwd + (long)b * ctx_fwd_ld + i) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
#pragma unroll
    for (int r = 0; r < TS / 4; ++r) {
        const int t = t0 + w + 4 * r;
#pragma unroll
        for (int m = 0; m < MROW; ++m) {
            const int i = lane * 4 + 256 * m;
            val[r][m] = (t < T && i < M) ? *reinterpret_cast<const float4*>(c.values + ((long)b * T + t) * M + i)     // (T, not len:
                                        : make_float4(0.f, 0.f, 0.f, 0.f);                                                // a[t] = 0 there)
        }
    }
    // ---- this slice's G (32-lane sums of the diagonal taps) and the row's alignments in LDS
    {
        static_assert(LPR == 32, "half_sum_in_last_lane sums 32-lane groups");
        const float x = half_sum_in_last_lane(hv);
        if (jh == LPR - 1) s_G[tt_h] = x;
    }
#pragma unroll
    for (int i = 0; i < T_MAX / 256; ++i) {
        const int t = threadIdx.x + 256 * i;
        if (t < T) s_a[t] = av[i];
    }
    *reinterpret_cast<float4*>(&s_dc[lane * 4 + 256 * w]) = dsum;
    if (threadIdx.x < TS) s_gn[threadIdx.x] = gn_own;
    // the filter / window of the tanh recompute
#pragma unroll
    for (int i = 0; i < NLK4; ++i) {
        const int e = (threadIdx.x + 256 * i) * 4;
        if (e < KS_MAX * A_) *reinterpret_cast<float4*>(&s_lk[e / A_][e % A_]) = lkv[i];
    }
    if (threadIdx.x < TS + KS_MAX - 1) s_cum[threadIdx.x] = cwin;
    __syncthreads();
    // tanh terms of this slice (independent of everything row-wide): u = tanh(keys + q + location filter), fac = w (1 - u^2)
    float fac[TS / 2];
    {
        float lk[KS_MAX];
#pragma unroll
        for (int j = 0; j < KS_MAX; ++j) lk[j] = s_lk[j][k];
#pragma unroll
        for (int i = 0; i < TS / 2; ++i) {
            const int tt = grp + 2 * i;
            float pre = kv[i] + qk;
#pragma unroll
            for (int j = 0; j < KS_MAX; ++j) pre += s_cum[tt + j] * lk[j];
            const float u = fast_tanh(pre);
            fac[i] = (t0 + tt < T) ? wk * (1.f - u * u) : 0.f;
        }
    }
#pragma unroll
    for (int m = 0; m < MROW; ++m) dcv[m] = *reinterpret_cast<const float4*>(&s_dc[lane * 4 + 256 * m]);
    // dot(a, G) = dot(a, G_next) + sum_{tau, j} h_next[tau][j] a[tau - pad + j]   (G[t] = G_next[t] + sum_j h_next[t + pad - j][j])
    float part = 0.f;
#pragma unroll
    for (int i = 0; i < T_MAX / 256; ++i) part += av[i] * gnv[i];
    auto tile_dot = [&](const float4& hq, int e) {
        const int tau = e >> 3, j0 = (e & 7) * 4, t = tau - pad + j0;
        float r = 0.f;
        if (j0 < KS && t >= 0 && t < T) r += hq.x * s_a[t];
        if (j0 + 1 < KS && t + 1 >= 0 && t + 1 < T) r += hq.y * s_a[t + 1];
        if (j0 + 2 < KS && t + 2 >= 0 && t + 2 < T) r += hq.z * s_a[t + 2];
        if (j0 + 3 < KS && t + 3 >= 0 && t + 3 < T) r += hq.w * s_a[t + 3];
        return r;
    };
#pragma unroll
    for (int i = 0; i < HT4; ++i) part += tile_dot(ht[i], threadIdx.x + 256 * i);
    if (h_next)
        for (int e = threadIdx.x + 256 * HT4; (e >> 3) < T; e += 256)            // T > 128 only
            part += tile_dot(reinterpret_cast<const float4*>(h_next + (long)b * T * HLD)[e], e);
    // ctx . d_ctx on wave 0 (its lanes hold both vectors in the same layout)
    if (w == 0) {
        float p2 = 0.f;
#pragma unroll
        for (int m = 0; m < MROW; ++m) p2 += cxv[m].x * dcv[m].x + cxv[m].y * dcv[m].y + cxv[m].z * dcv[m].z + cxv[m].w * dcv[m].w;
        for (int i = lane * 4 + 256 * MROW; i < M; i += 256) {            // M > 1024 only
            const float4 x = *reinterpret_cast<const float4*>(ctx_fwd + (long)b * ctx_fwd_ld + i);
            float4 y = *reinterpret_cast<const float4*>(dc + i);
            if (d_ctx2)
                for (int pp = 0; pp < max(d_ctx2_parts, 1); ++pp) {
                    const float4 y2 = *reinterpret_cast<const float4*>(d_ctx2 + pp * d_ctx2_pstride + (long)b * d_ctx2_ld + i);
                    y.x += y2.x; y.y += y2.y; y.z += y2.z; y.w += y2.w;
                }
            p2 += x.x * y.x + x.y * y.y + x.z * y.z + x.w * y.w;
        }
        p2 = wave_sum(p2);
        if (lane == 0) s_dot2 = p2;
    }
    // (no barrier here: the per-wave partial sums of dot(a, G), ctx . d_ctx and the d_align rows all meet behind the ONE barrier below -
    // the kernel had six barriers, each an LDS round trip on its critical path; it has three)
    part = wave_sum(part);
    if (lane == 0) scratch[w] = part;
    // ---- d_align = G + values . d_ctx for this slice's rows (kept in LDS)
#pragma unroll
    for (int r = 0; r < TS / 4; ++r) {
        const int tt = w + 4 * r, t = t0 + tt;
        float acc = 0.f;
#pragma unroll
        for (int m = 0; m < MROW; ++m)
            acc += val[r][m].x * dcv[m].x + val[r][m].y * dcv[m].y + val[r][m].z * dcv[m].z + val[r][m].w * dcv[m].w;
        if (t < len) {
            for (int i = lane * 4 + 256 * MROW; i < M; i += 256) {
                const float4 x = *reinterpret_cast<const float4*>(c.values + ((long)b * T + t) * M + i);
                float4 y = *reinterpret_cast<const float4*>(dc + i);
                if (d_ctx2)
                    for (int pp = 0; pp < max(d_ctx2_parts, 1); ++pp) {
                        const float4 y2 = *reinterpret_cast<const float4*>(d_ctx2 + pp * d_ctx2_pstride + (long)b * d_ctx2_ld + i);
                        y.x += y2.x; y.y += y2.y; y.z += y2.z; y.w += y2.w;
                    }
                acc += x.x * y.x + x.y * y.y + x.z * y.z + x.w * y.w;
            }
        }
        acc = wave_sum(acc);
        if (lane == 0) {
            const float g = (t < T) ? s_G[tt] + s_gn[tt] : 0.f;
            if (t < T) G[(long)b * T + t] = g;
            s_da[tt] = (t < T) ? g + acc : 0.f;
        }
    }
    __syncthreads();
    const float dot = ((scratch[0] + scratch[1]) + (scratch[2] + scratch[3])) + s_dot2;
    if (threadIdx.x < TS) {
        const int t = t0 + threadIdx.x;
        if (t < T) d_e_out[(long)b * T + t] = a_own * (s_da[threadIdx.x] - dot);
    }
    float dq_acc = 0.f;
#pragma unroll
    for (int i = 0; i < TS / 2; ++i) {
        const int tt = grp + 2 * i;
        const float de = (t0 + tt < T) ? s_a[t0 + tt] * (s_da[tt] - dot) : 0.f;     // every thread forms the d_e of its own positions
        const float g = de * fac[i];
        s_g[tt][k] = g;
        dq_acc += g;
    }
    if (grp == 1) s_dq[k] = dq_acc;
    __syncthreads();
    if (grp == 0) atomicAdd(dq + (long)b * A_ + k, dq_acc + s_dq[k]);
    {
        const int tt = threadIdx.x >> 5, j = threadIdx.x & 31;
        float acc = 0.f;
        if (j < KS) {
#pragma unroll 8
            for (int kk = 0; kk < A_; kk += 4) {
                const float4 g4 = *reinterpret_cast<const float4*>(&s_g[tt][kk]);
                const float4 l4 = *reinterpret_cast<const float4*>(&s_lk[j][kk]);
                acc += g4.x * l4.x + g4.y * l4.y + g4.z * l4.z + g4.w * l4.w;
            }
        }
        if (t0 + tt < T) h[((long)b * T + t0 + tt) * HLD + j] = acc;
    }
}

// ---------------------------------------------------------------------------------------------
// post-loop parameter gradients (recompute g per step from saved d_e, q, cum):
//   d_keys[b,t,k] += g ; d_loc_k[j,k] += cum[t+j-pad] g ; d_score_w[k] += d_e u ; d_score_b[k] += g
// (d_loc_b == d_score_b: both biases add to the same pre-activation)
// ---------------------------------------------------------------------------------------------
// Both contractions with the 31-tap window are matrix products: for one (b, s) and a tile of 32 encoder positions, with
// W[t][j] = cum[t + j - pad] (a 32 x 32 Toeplitz tile read straight from the LDS window),
//     L  = W . loc_k          [32 t x 32 j] . [32 j x 128 a]      -> pre-activation -> u, g (VALU, in the MFMA C layout)
//     dK = W^T . g            [32 j x 32 t] . [32 t x 128 a]      -> d_loc_k
// They run on v_mfma_f32_32x32x16_bf16 as exact three-way bf16 splits (x = hi + mid + lo, the six products down to 2^-24 of |a||b|, fp32
// accumulators - the arithmetic of csrc/gemm_split.inc): 24 matrix instructions of 8 passes per step and wave where the f32-input form needs 32
// of 16 passes.  The window is split once per step when it is written to LDS, as EIGHT copies per plane, copy c shifted by c elements, so that
// every lane's run of 8 consecutive window elements is one aligned 16-byte read from copy (start & 7); the filter taps are split once per launch;
// g is split in registers.  The row index of the first product is permuted (bits 2 and 3 swapped) so that the C layout hands each lane, per
// k-step of the second product, 8 CONSECUTIVE positions - g never leaves its registers and the second product's window runs are contiguous too.
// Wave w owns attention units 32w .. 32w+31; a workgroup walks a chunk of steps for one (row, tile); accumulators (d_keys tile, d_loc_k,
// d_score_w/b) stay in registers over the chunk.
constexpr int PT = 32;                      // encoder positions per workgroup
constexpr int LP_ROWS = 34, LP_GROUPS = 16; // partial block rows (32 filter taps incl. padding | score w | score b); block groups of the fp64 reduction
constexpr int LP_WN = 80;                   // bf16 elements per window copy: 8 of padding in front (lanes below the copy's shift write there: no branch), 64 + 8 behind
typedef float lp_f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 lp_bf16x8 __attribute__((ext_vector_type(8)));
__device__ __forceinline__ uint32_t lp_cvt_pk(float a, float b) {
    uint32_t r;
    asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
// (x0, x1) -> the three bf16 planes, packed pairs; remainders exact
__device__ __forceinline__ void lp_split2(float x0, float x1, uint32_t& hi, uint32_t& mid, uint32_t& lo) {
    hi = lp_cvt_pk(x0, x1);
    x0 -= __uint_as_float(hi << 16); x1 -= __uint_as_float(hi & 0xffff0000u);
    mid = lp_cvt_pk(x0, x1);
    x0 -= __uint_as_float(mid << 16); x1 -= __uint_as_float(mid & 0xffff0000u);
    lo = lp_cvt_pk(x0, x1);
}
__device__ __forceinline__ void lp_split8(const float* x, lp_bf16x8& hi, lp_bf16x8& mid, lp_bf16x8& lo) {
    uint4 h, m, l;
    lp_split2(x[0], x[1], h.x, m.x, l.x); lp_split2(x[2], x[3], h.y, m.y, l.y);
    lp_split2(x[4], x[5], h.z, m.z, l.z); lp_split2(x[6], x[7], h.w, m.w, l.w);
    hi = __builtin_bit_cast(lp_bf16x8, h); mid = __builtin_bit_cast(lp_bf16x8, m); lo = __builtin_bit_cast(lp_bf16x8, l);
}
#define LP_SIX(acc, ah, am, al, bh, bm, bl) do { \
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, acc, 0, 0, 0); acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, acc, 0, 0, 0); \
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bm, acc, 0, 0, 0); acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bh, acc, 0, 0, 0); \
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bm, acc, 0, 0, 0); acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, acc, 0, 0, 0); } while (0)
#ifndef LSA_PARAM_WAVES
#define LSA_PARAM_WAVES 3        // waves per SIMD the register budget is cut for (168 registers: 15 spilled, still 428 us against 466 at 2 waves - same-box A/B)
#endif
__global__ __launch_bounds__(256, LSA_PARAM_WAVES) void lsa_param_bwd_kernel(mstts_lsa_const c, int S, int steps_per_block,
                                                            const float* __restrict__ q_hist, const float* __restrict__ cum_hist,
                                                            const float* __restrict__ de_hist, float* __restrict__ d_keys,
                                                            float* __restrict__ d_loc_k, float* __restrict__ d_score_w,
                                                            float* __restrict__ d_score_b, float* __restrict__ part) {
    __shared__ __attribute__((aligned(16))) uint16_t s_wp[2][3][8][LP_WN];   // [buffer][plane][copy c][8 + i]: plane of cum[t0 - pad + i + c] (zero outside the sequence / window)
    __shared__ __attribute__((aligned(16))) float s_de[2][PT];
    const int b = blockIdx.x, t0 = blockIdx.y * PT, T = (int)c.T, B = (int)c.B, KS = (int)c.KS, pad = (KS - 1) / 2;
    const int s_beg = blockIdx.z * steps_per_block, s_end = min(S, s_beg + steps_per_block);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, l31 = lane & 31, kh = lane >> 5;
    const int a = wave * 32 + l31;
    const int pl = (l31 & 0x13) | ((l31 & 4) << 1) | ((l31 & 8) >> 1);      // the position (within the tile) of row l31 of the first product
    const int o1 = (pl & 7) * LP_WN + 8 + (pl & ~7) + 8 * kh, o2 = (l31 & 7) * LP_WN + 8 + (l31 & ~7) + 8 * kh;    // copy | start of the lane's window runs
    // B operand of the first product: filter taps j = 16 ks + 8 kh + i of this wave's 32 units (zero rows beyond the KS taps), split once
    lp_bf16x8 fh[2], fm[2], fl[2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
        float f[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) { const int j = 16 * ks + 8 * kh + i; f[i] = j < KS ? c.loc_k[j * A_ + a] : 0.f; }
        lp_split8(f, fh[ks], fm[ks], fl[ks]);
    }
    const float sb = c.score_b[a] + c.loc_b[a], wk = c.score_w[a];
    float key_v[16];
    lp_f32x16 acc_keys, acc_lk;
#pragma unroll
    for (int r = 0; r < 16; ++r) {                 // register r of the first product's C layout: position (r & 7) + 8 kh + 16 (r >> 3)
        const int t = t0 + (r & 7) + 8 * kh + 16 * (r >> 3);
        key_v[r] = t < T ? c.keys[((long)b * T + t) * A_ + a] : 0.f;
        acc_keys[r] = 0.f; acc_lk[r] = 0.f;
    }
    float acc_w = 0.f, acc_b = 0.f;
    auto load_win = [&](int s) -> float {          // waves 0 .. 2: the window (each wave writes one plane); wave 3: d_e
        float v = 0.f;
        if (wave < 3) {
            const int t = t0 - pad + lane;
            if (lane < PT + KS - 1 && t >= 0 && t < T) v = cum_hist[((long)s * B + b) * T + t];
        } else if (lane < PT) {
            if (t0 + lane < T) v = de_hist[((long)s * B + b) * T + t0 + lane];
        }
        return v;
    };
    float nv = 0.f, nq = 0.f;
    if (s_beg < s_end) { nv = load_win(s_beg); nq = q_hist[((long)s_beg * B + b) * A_ + a]; }
    int buf = 0;
    for (int s = s_beg; s < s_end; ++s, buf ^= 1) {
        if (wave < 3) {
            uint32_t p0, p1, p2;
            lp_split2(nv, 0.f, p0, p1, p2);
            const uint16_t bits = (uint16_t)(wave == 0 ? p0 : wave == 1 ? p1 : p2);
            uint16_t* w = &s_wp[buf][wave][0][8 + lane];
#pragma unroll
            for (int cc = 0; cc < 8; ++cc) w[cc * LP_WN - cc] = bits;
        } else if (lane < PT) s_de[buf][lane] = nv;
        const float qk = nq + sb;
        if (s + 1 < s_end) { nv = load_win(s + 1); nq = q_hist[((long)(s + 1) * B + b) * A_ + a]; }
        __syncthreads();                    // buf written; the other buffer is free to be rewritten next iteration
        const uint16_t* wp = &s_wp[buf][0][0][0];
        lp_f32x16 L;
#pragma unroll
        for (int r = 0; r < 16; ++r) L[r] = 0.f;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {     // A[row l31 = position pl][j = 16 ks + 8 kh + i] = win[pl + j]
            const lp_bf16x8 ah = *reinterpret_cast<const lp_bf16x8*>(wp + o1 + 16 * ks), am = *reinterpret_cast<const lp_bf16x8*>(wp + 8 * LP_WN + o1 + 16 * ks),
                            al = *reinterpret_cast<const lp_bf16x8*>(wp + 16 * LP_WN + o1 + 16 * ks);
            LP_SIX(L, ah, am, al, fh[ks], fm[ks], fl[ks]);
        }
        float g[16];
#pragma unroll
        for (int r8 = 0; r8 < 2; ++r8) {
            const float4 d0 = *reinterpret_cast<const float4*>(&s_de[buf][8 * kh + 16 * r8]), d1 = *reinterpret_cast<const float4*>(&s_de[buf][8 * kh + 16 * r8 + 4]);
            const float de8[8] = {d0.x, d0.y, d0.z, d0.w, d1.x, d1.y, d1.z, d1.w};           // 0 beyond T
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int r = 8 * r8 + i;
                const float u = tanhf_(L[r] + key_v[r] + qk);       // (the persistent forward's form of the same pre-activation)
                g[r] = de8[i] * wk * (1.f - u * u);
                acc_w += de8[i] * u;
                acc_b += g[r];
                acc_keys[r] += g[r];
            }
        }
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {     // k = (kh, i) of k-step ks: position 16 ks + 8 kh + i = register 8 ks + i; A[tap l31][k] = win[position + l31]
            lp_bf16x8 gh, gm, gl;
            lp_split8(g + 8 * ks, gh, gm, gl);
            const lp_bf16x8 ah = *reinterpret_cast<const lp_bf16x8*>(wp + o2 + 16 * ks), am = *reinterpret_cast<const lp_bf16x8*>(wp + 8 * LP_WN + o2 + 16 * ks),
                            al = *reinterpret_cast<const lp_bf16x8*>(wp + 16 * LP_WN + o2 + 16 * ks);
            LP_SIX(acc_lk, ah, am, al, gh, gm, gl);
        }
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int row = (r & 7) + 8 * kh + 16 * (r >> 3);
        if (t0 + row < T) atomicAdd(d_keys + ((long)b * T + t0 + row) * A_ + a, acc_keys[r]);
    }
    acc_w += __shfl_xor(acc_w, 32);            // (once per workgroup, off any per-step path)
    acc_b += __shfl_xor(acc_b, 32);
    if (part) {
        // The filter / score-layer gradients are sums over EVERY (row, step, position) of the batch: thousands of workgroup partials per element
        // with heavy cancellation.  Summed with fp32 atomics they come out 5e-3 off (of the gradient's maximum) at batch 32 x 801 steps and
        // differ run to run; each workgroup therefore writes its partial block [34][128] (31 taps | - | score w | score b) and
        // lsa_param_reduce_kernel adds the blocks in fp64 in a fixed order.
        float* pb = part + ((long)(blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x) * (LP_ROWS * A_);
#pragma unroll
        for (int r = 0; r < 16; ++r) pb[((r & 3) + 8 * (r >> 2) + 4 * kh) * A_ + a] = acc_lk[r];
        if (kh == 0) { pb[32 * A_ + a] = acc_w; pb[33 * A_ + a] = acc_b; }
        return;
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int row = (r & 3) + 8 * (r >> 2) + 4 * kh;
        if (row < KS) atomicAdd(d_loc_k + row * A_ + a, acc_lk[r]);                 // row = tap j here
    }
    if (kh == 0) {
        atomicAdd(d_score_w + a, acc_w);
        atomicAdd(d_score_b + a, acc_b);
    }
}
// stage 1: block group gz sums its share of the partial blocks for 64 elements in fp64 (fixed order); stage 2 adds the LP_GROUPS group sums in order
__global__ __launch_bounds__(256) void lsa_param_reduce1_kernel(const float* __restrict__ part, int nblk, double* __restrict__ gsum) {
    __shared__ double red[4][64];
    const int e = blockIdx.x * 64 + (threadIdx.x & 63), sub = threadIdx.x >> 6, gz = blockIdx.y;
    const int per = (nblk + LP_GROUPS - 1) / LP_GROUPS, b0 = gz * per, b1 = min(nblk, b0 + per);
    double acc = 0.0;
#pragma unroll 8
    for (int bl = b0 + sub; bl < b1; bl += 4) acc += (double)part[(long)bl * (LP_ROWS * A_) + e];
    red[sub][threadIdx.x & 63] = acc;
    __syncthreads();
    if (sub == 0) gsum[(long)gz * (LP_ROWS * A_) + e] = (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
}
__global__ void lsa_param_reduce2_kernel(const double* __restrict__ gsum, int KS, float* __restrict__ d_loc_k, float* __restrict__ d_score_w, float* __restrict__ d_score_b) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= LP_ROWS * A_) return;
    double v = 0.0;
#pragma unroll
    for (int gz = 0; gz < LP_GROUPS; ++gz) v += gsum[(long)gz * (LP_ROWS * A_) + e];
    const int row = e / A_, a = e - row * A_;
    if (row < KS) d_loc_k[row * A_ + a] += (float)v;
    else if (row == 32) d_score_w[a] += (float)v;
    else if (row == 33) d_score_b[a] += (float)v;
}

static int check_const(const mstts_lsa_const* c) {
    MSTTS_REQUIRE(c != nullptr, MSTTS_ERR_SHAPE, "lsa: null const block");
    MSTTS_REQUIRE(c->A == A_ && c->CH == CH_, MSTTS_ERR_SHAPE, "lsa: built for A=%d CH=%d, got A=%ld CH=%ld", A_, CH_, (long)c->A, (long)c->CH);
    MSTTS_REQUIRE(c->loc_k && c->loc_b, MSTTS_ERR_SHAPE, "lsa: folded location filter (loc_k/loc_b) missing - call mstts_lsa_fold_location first");
    MSTTS_REQUIRE(c->KS >= 1 && c->KS <= KS_MAX && (c->KS & 1), MSTTS_ERR_SHAPE, "lsa: conv taps must be odd and <= %d", KS_MAX);
    MSTTS_REQUIRE(c->T >= 1 && c->T <= T_MAX, MSTTS_ERR_SHAPE, "lsa: T must be in [1,%d]", T_MAX);
    MSTTS_REQUIRE(c->M % 4 == 0 && aligned16(c->values), MSTTS_ERR_ALIGN, "lsa: memory width %% 4 and 16-byte aligned values required");
    MSTTS_REQUIRE(c->B >= 1, MSTTS_ERR_SHAPE, "lsa: B must be >= 1");
    return MSTTS_OK;
}

}  // namespace mstts

using namespace mstts;
#define ST(s) ((hipStream_t)(s))

extern "C" int mstts_lsa_energy_fwd(const mstts_lsa_const* c, const float* q, int32_t q_parts, int64_t q_pstride, float* q_sum,
                                    const float* cum, float* energy, mstts_stream_t s) {
    int rc = check_const(c); if (rc) return rc;
    hipLaunchKernelGGL(lsa_energy_kernel, dim3((unsigned)c->B, cdiv(c->T, TS)), dim3(256), 0, ST(s), *c, q, (int)q_parts, (long)q_pstride, q_sum, cum, energy);
    MSTTS_CHECK_LAUNCH("lsa_energy_fwd");
    return MSTTS_OK;
}
extern "C" int mstts_lsa_context_fwd(const mstts_lsa_const* c, const float* energy, const float* cum, float* align, float* cum_next,
                                     float* ctx, int64_t ctx_ld, float* ctx2, int64_t ctx2_ld, mstts_stream_t s) {
    int rc = check_const(c); if (rc) return rc;
    hipLaunchKernelGGL(lsa_context_kernel, dim3((unsigned)c->B, cdiv(c->M, DS)), dim3(256), 0, ST(s), *c, energy, cum, align, cum_next,
                       ctx, (long)ctx_ld, ctx2, (long)ctx2_ld);
    MSTTS_CHECK_LAUNCH("lsa_context_fwd");
    return MSTTS_OK;
}
static void lsa_step_geometry(long T, long M, int* cs, int* tsl, int* dsl) {
    long n = cdiv(T, FS_TSL);
    const long nm = cdiv(M, FS_DSL);
    if (nm > n) n = nm;
    *cs = (int)n;
    *tsl = (int)cdiv(T, n);
    *dsl = (int)(cdiv(cdiv(M, n), 4) * 4);
}
extern "C" int64_t mstts_lsa_step_ws_bytes(int64_t B, int64_t T) { return (B * T + 1) * 8; }
static int lsa_step_fwd_launch(const mstts_lsa_const* c, const float* q, int32_t q_parts, int64_t q_pstride, float* q_sum,
                               const float* cum, float* align, float* cum_next, float* ctx, int64_t ctx_ld, float* ctx2, int64_t ctx2_ld,
                               const mstts_cell_packed_dst* ctx_p, void* granules, uint32_t epoch, int skip, mstts_stream_t s,
                               const LsaQIn* qin = nullptr, const LsaProj* proj = nullptr, const mstts_lsa_prenet* pre = nullptr) {
    int rc = check_const(c); if (rc) return rc;
    MSTTS_REQUIRE(granules && epoch != 0 && ((uintptr_t)granules & 7) == 0, MSTTS_ERR_SHAPE, "lsa_step_fwd: granule buffer (8-byte aligned) and a non-zero epoch required");
    PackedDst cp;
    rc = packed_dst_from(ctx_p, c->M, &cp, "ctx_p"); if (rc) return rc;
    int cs, tsl, dsl;
    lsa_step_geometry(c->T, c->M, &cs, &tsl, &dsl);
    LsaQIn qi;
    memset(&qi, 0, sizeof(qi));
    LsaProj pz;
    memset(&pz, 0, sizeof(pz));
    LsaPre px;
    memset(&px, 0, sizeof(px));
    const bool lkt = c->loc_kt && aligned16(c->loc_kt);
    if (qin && proj) {
        qi = *qin; pz = *proj;
        MSTTS_REQUIRE(cs >= 8 && qi.H == 128 * QJ && lkt, MSTTS_ERR_SHAPE, "lsa_step_fwd_qp: needs >= 8 slices, H == %d and the by-unit filter", 128 * QJ);
        MSTTS_REQUIRE(qi.m1 && qi.wq && aligned16(qi.m1) && aligned16(qi.wq) && qi.m1_ld % 4 == 0, MSTTS_ERR_ALIGN, "lsa_step_fwd_qp: m1 / wq must be 16-byte aligned");
        MSTTS_REQUIRE(pz.wp_own && aligned16(pz.wp_own) && pz.vp && pz.linear && pz.stop && pz.NP >= 2 && pz.NP <= 8 * PJ_OWN && pz.NM < pz.NP, MSTTS_ERR_SHAPE,
                      "lsa_step_fwd_qp: projection width must be 2..%d columns", 8 * PJ_OWN);
        if (pre) {
            MSTTS_REQUIRE(pre->w0 && pre->b0 && pre->w1 && pre->b1 && pre->m0 && pre->m1 && pre->out && pre->P == PR_P && pz.NM <= 8 * PR_K0 &&
                          pz.NM < PR_GLD && pz.NP == 4 * PF_Q && aligned16(pre->w0) && aligned16(pre->w1) && aligned16(pz.vp), MSTTS_ERR_SHAPE,
                          "lsa_step_fwd_qp: in-launch prenet needs P == %d, n_mel <= %d, NP == %d and 16-byte aligned w0 / w1 / vp", PR_P, 8 * PR_K0, 4 * PF_Q);
            px.w0 = pre->w0; px.b0 = pre->b0; px.w1 = pre->w1; px.b1 = pre->b1; px.m0 = pre->m0; px.m1 = pre->m1; px.inv_keep = pre->inv_keep;
            px.out = pre->out; px.out_ld = (long)pre->out_ld;
            rc = packed_dst_from(pre->out_p.base ? &pre->out_p : nullptr, PR_P, &px.out_p, "prenet out_p"); if (rc) return rc;
            px.gf = (unsigned long long*)granules + (c->B * c->T + 1 + c->B * A_);
            if (skip >= 0)
                hipLaunchKernelGGL((lsa_step_kernel<true, true, true, true, true>), dim3((unsigned)(cs * c->B)), dim3(FS_THREADS), 0, ST(s), *c, q, 0, 0L, q_sum, cum,
                                   align, cum_next, ctx, (long)ctx_ld, ctx2, (long)ctx2_ld, cp, (unsigned long long*)granules, (unsigned)epoch, tsl, dsl, cs, skip, qi, pz, px);
            else
                hipLaunchKernelGGL((lsa_step_kernel<false, true, true, true, true>), dim3((unsigned)(cs * c->B)), dim3(FS_THREADS), 0, ST(s), *c, q, 0, 0L, q_sum, cum,
                                   align, cum_next, ctx, (long)ctx_ld, ctx2, (long)ctx2_ld, cp, (unsigned long long*)granules, (unsigned)epoch, tsl, dsl, cs, -1, qi, pz, px);
        } else if (skip >= 0)
            hipLaunchKernelGGL((lsa_step_kernel<true, true, true, true>), dim3((unsigned)(cs * c->B)), dim3(FS_THREADS), 0, ST(s), *c, q, 0, 0L, q_sum, cum,
                               align, cum_next, ctx, (long)ctx_ld, ctx2, (long)ctx2_ld, cp, (unsigned long long*)granules, (unsigned)epoch, tsl, dsl, cs, skip, qi, pz, px);
        else
            hipLaunchKernelGGL((lsa_step_kernel<false, true, true, true>), dim3((unsigned)(cs * c->B)), dim3(FS_THREADS), 0, ST(s), *c, q, 0, 0L, q_sum, cum,
                               align, cum_next, ctx, (long)ctx_ld, ctx2, (long)ctx2_ld, cp, (unsigned long long*)granules, (unsigned)epoch, tsl, dsl, cs, -1, qi, pz, px);
    } else if (qin) {
        qi = *qin;
        MSTTS_REQUIRE(cs >= 8 && qi.H == 128 * QJ && lkt, MSTTS_ERR_SHAPE, "lsa_step_fwd_q: needs at least 8 slices (T > 112 or M > 672), H == %d and the by-unit filter", 128 * QJ);
        MSTTS_REQUIRE(qi.m1 && qi.wq && aligned16(qi.m1) && aligned16(qi.wq) && qi.m1_ld % 4 == 0, MSTTS_ERR_ALIGN, "lsa_step_fwd_q: m1 / wq must be 16-byte aligned");
        if (skip >= 0)
            hipLaunchKernelGGL((lsa_step_kernel<true, true, true>), dim3((unsigned)(cs * c->B)), dim3(FS_THREADS), 0, ST(s), *c, q, 0, 0L, q_sum, cum,
                               align, cum_next, ctx, (long)ctx_ld, ctx2, (long)ctx2_ld, cp, (unsigned long long*)granules, (unsigned)epoch, tsl, dsl, cs, skip, qi, pz, px);
        else
            hipLaunchKernelGGL((lsa_step_kernel<false, true, true>), dim3((unsigned)(cs * c->B)), dim3(FS_THREADS), 0, ST(s), *c, q, 0, 0L, q_sum, cum,
                               align, cum_next, ctx, (long)ctx_ld, ctx2, (long)ctx2_ld, cp, (unsigned long long*)granules, (unsigned)epoch, tsl, dsl, cs, -1, qi, pz, px);
    } else if (skip >= 0)
        hipLaunchKernelGGL(lsa_step_kernel<true>, dim3((unsigned)(cs * c->B)), dim3(FS_THREADS), 0, ST(s), *c, q, (int)q_parts, (long)q_pstride, q_sum, cum,
                           align, cum_next, ctx, (long)ctx_ld, ctx2, (long)ctx2_ld, cp, (unsigned long long*)granules, (unsigned)epoch, tsl, dsl, cs, skip, qi, pz, px);
    else if (lkt)
        hipLaunchKernelGGL((lsa_step_kernel<false, true>), dim3((unsigned)(cs * c->B)), dim3(FS_THREADS), 0, ST(s), *c, q, (int)q_parts, (long)q_pstride, q_sum, cum,
                           align, cum_next, ctx, (long)ctx_ld, ctx2, (long)ctx2_ld, cp, (unsigned long long*)granules, (unsigned)epoch, tsl, dsl, cs, -1, qi, pz, px);
    else
        hipLaunchKernelGGL(lsa_step_kernel<false>, dim3((unsigned)(cs * c->B)), dim3(FS_THREADS), 0, ST(s), *c, q, (int)q_parts, (long)q_pstride, q_sum, cum,
                           align, cum_next, ctx, (long)ctx_ld, ctx2, (long)ctx2_ld, cp, (unsigned long long*)granules, (unsigned)epoch, tsl, dsl, cs, -1, qi, pz, px);
    MSTTS_CHECK_LAUNCH("lsa_step_fwd");
    return MSTTS_OK;
}
extern "C" int mstts_lsa_step_fwd(const mstts_lsa_const* c, const float* q, int32_t q_parts, int64_t q_pstride, float* q_sum,
                                  const float* cum, float* align, float* cum_next, float* ctx, int64_t ctx_ld, float* ctx2, int64_t ctx2_ld,
                                  const mstts_cell_packed_dst* ctx_p, void* granules, uint32_t epoch, mstts_stream_t s) {
    return lsa_step_fwd_launch(c, q, q_parts, q_pstride, q_sum, cum, align, cum_next, ctx, ctx_ld, ctx2, ctx2_ld, ctx_p, granules, epoch, -1, s);
}
/* The same step with the query projection inside the launch: q = m1 . Wq (m1 rows [B, H] with row stride m1_ld, Wq [H, A] row-major;
 * q_bf16 != 0 rounds both operands to bf16 first - BASELINE config 3), one launch less per decoder step.  Needs at least 8 slices
 * (T > 112 or M > 672: the first eight own 16 query units each), H == 1024 and c->loc_kt; granules = mstts_lsa_step_q_ws_bytes(B, T) bytes (the energy granules, the time-out counter,
 * then B * A query granules), zeroed before the first step.  skip_slice >= 0: the self-test form (see below), -1 otherwise. */
extern "C" int32_t mstts_lsa_step_q_supported(int64_t T, int64_t M, int64_t H) {
    int cs, tsl, dsl;
    if (T < 1 || M < 4) return 0;
    lsa_step_geometry(T, M, &cs, &tsl, &dsl);
    return cs >= 8 && H == 128 * QJ;
}
extern "C" int64_t mstts_lsa_step_q_ws_bytes(int64_t B, int64_t T) { return (B * T + 1 + B * A_) * 8; }
extern "C" int mstts_lsa_step_fwd_q(const mstts_lsa_const* c, const float* m1, int64_t m1_ld, const float* wq, int64_t H, int32_t q_bf16,
                                    float* q_sum, const float* cum, float* align, float* cum_next, float* ctx, int64_t ctx_ld, float* ctx2,
                                    int64_t ctx2_ld, const mstts_cell_packed_dst* ctx_p, void* granules, uint32_t epoch, int32_t skip_slice,
                                    mstts_stream_t s) {
    LsaQIn qi;
    qi.m1 = m1; qi.m1_ld = (long)m1_ld; qi.wq = wq; qi.H = (int)H; qi.bf16 = q_bf16 ? 1 : 0;
    return lsa_step_fwd_launch(c, nullptr, 0, 0, q_sum, cum, align, cum_next, ctx, ctx_ld, ctx2, ctx2_ld, ctx_p, granules, epoch,
                               skip_slice >= 0 ? skip_slice : -1, s, &qi);
}
/* ... and with the output projection [m1 | ctx] . Wp + bias out of the same launch (free-running decoder): no further exchange -
 *   vp [B, T, NP] = values . Wp[H:, :] (the projected values: once per utterance, any GEMM), wp_own = mstts_lsa_proj_pack(Wp[:H, :]);
 *   bias [NM + 1] or NULL; linear [B, NM] <- columns 0..NM-1, stop [B] <- column NM.  NP <= 88.
 * Same availability as mstts_lsa_step_fwd_q; granules = mstts_lsa_step_qp_ws_bytes(B, T) bytes.
 * pre != NULL (mstts_lsa_step_prenet_supported(P, NM)): the NEXT step's prenet on this frame in the same launch - see LsaPre above. */
extern "C" int32_t mstts_lsa_step_qp_supported(int64_t T, int64_t M, int64_t H, int64_t NP) {
    return mstts_lsa_step_q_supported(T, M, H) && NP >= 2 && NP <= 8 * PJ_OWN;
}
extern "C" int64_t mstts_lsa_step_qp_ws_bytes(int64_t B, int64_t T) { return (B * T + 1 + B * A_ + B * PR_GLD) * 8; }
extern "C" int32_t mstts_lsa_step_prenet_supported(int64_t P, int64_t NM) { return P == PR_P && NM >= 1 && NM <= 8 * PR_K0 && (NM + 1 + 3) / 4 == PF_Q; }
extern "C" int64_t mstts_lsa_proj_pack_floats(void) { return 8L * QJ * 128 * PJ_OW; }
namespace mstts {
__global__ void lsa_proj_pack_kernel(const float* __restrict__ wp, long ld, int NP, float* __restrict__ out) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= 8 * QJ * 128 * PJ_OW) return;
    const int i = idx % PJ_OW, ch = (idx / PJ_OW) % 128, jj = (idx / (PJ_OW * 128)) % QJ, cs = idx / (PJ_OW * 128 * QJ);
    const int o = PJ_OWN * cs + i;
    out[idx] = (i < PJ_OWN && o < NP) ? wp[(long)(QJ * ch + jj) * ld + o] : 0.f;
}
}  // namespace mstts
/* wp [1024, >= NP] (row stride ld) -> wp_own [8 owners][8][128][16]: owner s holds columns 11 s .. 11 s + 10 (5 zero slots) in the order its lanes read them */
extern "C" int mstts_lsa_proj_pack(const float* wp, int64_t ld, int64_t H, int64_t NP, float* wp_own, mstts_stream_t s) {
    MSTTS_REQUIRE(wp && wp_own && H == 128 * QJ && NP >= 1 && NP <= 8 * PJ_OWN && ld >= NP, MSTTS_ERR_SHAPE, "lsa_proj_pack: needs H == %d and NP <= %d", 128 * QJ, 8 * PJ_OWN);
    const int n = 8 * QJ * 128 * PJ_OW;
    hipLaunchKernelGGL(mstts::lsa_proj_pack_kernel, dim3((n + 255) / 256), dim3(256), 0, ST(s), wp, (long)ld, (int)NP, wp_own);
    MSTTS_CHECK_LAUNCH("lsa_proj_pack");
    return MSTTS_OK;
}
extern "C" int mstts_lsa_step_fwd_qp(const mstts_lsa_const* c, const float* m1, int64_t m1_ld, const float* wq, int64_t H, const float* wp_own,
                                     const float* vp, const float* bias, int64_t NP, int64_t NM, float* linear, float* stop, const float* cum,
                                     float* align, float* cum_next, float* ctx, int64_t ctx_ld, float* ctx2, int64_t ctx2_ld,
                                     const mstts_cell_packed_dst* ctx_p, const mstts_lsa_prenet* pre, void* granules, uint32_t epoch,
                                     int32_t skip_slice, mstts_stream_t s) {
    MSTTS_REQUIRE(c && granules, MSTTS_ERR_SHAPE, "lsa_step_fwd_qp: null pointer");
    LsaQIn qi;
    qi.m1 = m1; qi.m1_ld = (long)m1_ld; qi.wq = wq; qi.H = (int)H; qi.bf16 = 0;
    LsaProj pj;
    pj.wp_own = wp_own; pj.vp = vp; pj.bias = bias; pj.NP = (int)NP; pj.NM = (int)NM; pj.linear = linear; pj.stop = stop;
    return lsa_step_fwd_launch(c, nullptr, 0, 0, nullptr, cum, align, cum_next, ctx, ctx_ld, ctx2, ctx2_ld, ctx_p, granules, epoch,
                               skip_slice >= 0 ? skip_slice : -1, s, &qi, &pj, pre);
}
/* test entry: same launch with the workgroups of slice `skip_slice` (0 .. slices-1) removed, which forces every other workgroup of each row
 * through its time-out path (takes milliseconds); the skipped slice's own outputs are not written */
extern "C" int mstts_lsa_step_fwd_selftest(const mstts_lsa_const* c, const float* q, int32_t q_parts, int64_t q_pstride, float* q_sum,
                                           const float* cum, float* align, float* cum_next, float* ctx, int64_t ctx_ld, void* granules,
                                           uint32_t epoch, int32_t skip_slice, mstts_stream_t s) {
    MSTTS_REQUIRE(skip_slice >= 0, MSTTS_ERR_SHAPE, "lsa_step_fwd_selftest: skip_slice must be >= 0");
    return lsa_step_fwd_launch(c, q, q_parts, q_pstride, q_sum, cum, align, cum_next, ctx, ctx_ld, nullptr, 0, nullptr, granules, epoch, skip_slice, s);
}
extern "C" int mstts_lsa_dalign_bwd(const mstts_lsa_const* c, const float* d_ctx, int64_t d_ctx_ld, const float* d_ctx2, int64_t d_ctx2_ld,
                                    int32_t d_ctx2_parts, int64_t d_ctx2_pstride, const float* G_next, const float* d_f_next, float* G, float* d_align, mstts_stream_t s) {
    int rc = check_const(c); if (rc) return rc;
    MSTTS_REQUIRE(aligned16(d_ctx) && aligned16(d_ctx2) && d_ctx_ld % 4 == 0 && d_ctx2_ld % 4 == 0, MSTTS_ERR_ALIGN,
                  "lsa_dalign: d_ctx rows must be 16-byte aligned");
    hipLaunchKernelGGL(lsa_dalign_kernel, dim3((unsigned)c->B, cdiv(c->T, TS)), dim3(256), 0, ST(s), *c, d_ctx, (long)d_ctx_ld, d_ctx2,
                       (long)d_ctx2_ld, (int)d_ctx2_parts, (long)d_ctx2_pstride, G_next, d_f_next, G, d_align);
    MSTTS_CHECK_LAUNCH("lsa_dalign_bwd");
    return MSTTS_OK;
}
extern "C" int mstts_lsa_denergy_bwd(const mstts_lsa_const* c, const float* align, const float* d_align, const float* q, const float* cum,
                                     float* d_e, float* dq, float* d_f, mstts_stream_t s) {
    int rc = check_const(c); if (rc) return rc;
    hipLaunchKernelGGL(lsa_denergy_kernel, dim3((unsigned)c->B, cdiv(c->T, TS)), dim3(256), 0, ST(s), *c, align, d_align, q, cum, d_e, dq, d_f);
    MSTTS_CHECK_LAUNCH("lsa_denergy_bwd");
    return MSTTS_OK;
}
extern "C" int mstts_lsa_step_bwd(const mstts_lsa_const* c, const float* d_ctx, int64_t d_ctx_ld, const float* d_ctx2, int64_t d_ctx2_ld,
                                  int32_t d_ctx2_parts, int64_t d_ctx2_pstride, const float* G_next, const float* d_f_next, float* G,
                                  const float* align, const float* q, const float* cum, const float* ctx_fwd, int64_t ctx_fwd_ld,
                                  float* d_e, float* dq, float* d_f, mstts_stream_t s) {
    int rc = check_const(c); if (rc) return rc;
    MSTTS_REQUIRE(aligned16(d_ctx) && aligned16(d_ctx2) && d_ctx_ld % 4 == 0 && d_ctx2_ld % 4 == 0, MSTTS_ERR_ALIGN,
                  "lsa_step_bwd: d_ctx rows must be 16-byte aligned");
    MSTTS_REQUIRE(ctx_fwd && aligned16(ctx_fwd) && ctx_fwd_ld % 4 == 0, MSTTS_ERR_ALIGN, "lsa_step_bwd: the forward context rows (16-byte aligned) are required");
    MSTTS_REQUIRE(aligned16(c->loc_k), MSTTS_ERR_ALIGN, "lsa_step_bwd: the folded filter loc_k must be 16-byte aligned");
    hipLaunchKernelGGL(lsa_step_bwd_kernel, dim3((unsigned)(cdiv(c->T, TS) * c->B)), dim3(256), 0, ST(s), *c, d_ctx, (long)d_ctx_ld, d_ctx2,
                       (long)d_ctx2_ld, (int)d_ctx2_parts, (long)d_ctx2_pstride, G_next, d_f_next, G, align, q, cum, ctx_fwd, (long)ctx_fwd_ld, d_e, dq, d_f,
                       cdiv(c->T, TS));
    MSTTS_CHECK_LAUNCH("lsa_step_bwd");
    return MSTTS_OK;
}
namespace mstts { int gemm_deterministic_now(); }       // csrc/gemm.hip
static void lsa_param_geometry(long B, long T, long S, int* nt, int* chunks, int* spb, bool fixed_order) {
    *nt = cdiv(T, PT);
    int ch = (int)(2048 / (B * *nt));
    if (ch < 1) ch = 1;
    if (fixed_order) ch = 1;                     // fixed summation order: one workgroup walks ALL the steps of its (row, tile), so d_keys receives one add per element
    if (ch > S) ch = (int)S;
    *spb = cdiv(S, ch);
    *chunks = cdiv(S, *spb);
}
extern "C" int64_t mstts_lsa_param_bwd_ws_floats(int64_t B, int64_t T, int64_t S) {
    if (B < 1 || T < 1 || S < 1) return 0;
    int nt, chunks, spb;
    lsa_param_geometry(B, T, S, &nt, &chunks, &spb, false);          // (the larger of the two geometries)
    return (int64_t)B * nt * chunks * (LP_ROWS * A_) + 2 * (int64_t)LP_GROUPS * (LP_ROWS * A_);      // partial blocks, then LP_GROUPS blocks of doubles
}
extern "C" int mstts_lsa_param_bwd(const mstts_lsa_const* c, int64_t S, const float* q_hist, const float* cum_hist, const float* de_hist,
                                   float* d_keys, float* d_loc_k, float* d_score_w, float* d_score_b, float* ws, mstts_stream_t s) {
    int rc = check_const(c); if (rc) return rc;
    if (S <= 0) return MSTTS_OK;
    MSTTS_REQUIRE(!ws || (reinterpret_cast<uintptr_t>(ws) & 7u) == 0, MSTTS_ERR_ALIGN, "lsa_param_bwd: the workspace must be 8-byte aligned");
    int nt, chunks, spb;
    lsa_param_geometry(c->B, c->T, S, &nt, &chunks, &spb, gemm_deterministic_now() != 0);
    hipLaunchKernelGGL(lsa_param_bwd_kernel, dim3((unsigned)c->B, nt, chunks), dim3(256), 0, ST(s), *c, (int)S, spb, q_hist, cum_hist,
                       de_hist, d_keys, d_loc_k, d_score_w, d_score_b, ws);
    MSTTS_CHECK_LAUNCH("lsa_param_bwd");
    if (ws) {
        const int nblk = (int)c->B * nt * chunks;
        long nb = (long)nblk * (LP_ROWS * A_);
        nb += nb & 1;                                                    // (doubles behind the blocks: keep them 8-byte aligned)
        double* gsum = reinterpret_cast<double*>(ws + nb);
        hipLaunchKernelGGL(lsa_param_reduce1_kernel, dim3(LP_ROWS * A_ / 64, LP_GROUPS), dim3(256), 0, ST(s), ws, nblk, gsum);
        MSTTS_CHECK_LAUNCH("lsa_param_reduce1");
        hipLaunchKernelGGL(lsa_param_reduce2_kernel, dim3(cdiv(LP_ROWS * A_, 256)), dim3(256), 0, ST(s), gsum, (int)c->KS, d_loc_k, d_score_w, d_score_b);
        MSTTS_CHECK_LAUNCH("lsa_param_reduce2");
    }
    return MSTTS_OK;
}

namespace mstts {
__global__ void lsa_filter_by_unit_kernel(const float* __restrict__ loc_k, float* __restrict__ loc_kt, int KS, int A) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= A * LKT_LD) return;
    const int a = i / LKT_LD, j = i - a * LKT_LD;
    loc_kt[i] = j < KS ? loc_k[j * A + a] : 0.f;
}
}  // namespace mstts
extern "C" int mstts_lsa_filter_by_unit(const float* loc_k, float* loc_kt, int64_t KS, int64_t A, mstts_stream_t s) {
    MSTTS_REQUIRE(loc_k && loc_kt && KS >= 1 && KS <= KS_MAX && A >= 1, MSTTS_ERR_SHAPE, "lsa_filter_by_unit: bad arguments (KS <= %d)", KS_MAX);
    MSTTS_REQUIRE(aligned16(loc_kt), MSTTS_ERR_ALIGN, "lsa_filter_by_unit: loc_kt must be 16-byte aligned");
    hipLaunchKernelGGL(lsa_filter_by_unit_kernel, dim3((unsigned)((A * LKT_LD + 255) / 256)), dim3(256), 0, ST(s), loc_k, loc_kt, (int)KS, (int)A);
    MSTTS_CHECK_LAUNCH("lsa_filter_by_unit");
    return MSTTS_OK;
}

/* loc_k[KS,A] = conv_k[KS,CH] . dense_k[CH,A] ; loc_b[A] = conv_b[CH] . dense_k */
extern "C" int mstts_lsa_fold_location(const float* conv_k, const float* conv_b, const float* dense_k, float* loc_k, float* loc_b,
                                       int64_t KS, int64_t CH, int64_t A, mstts_stream_t s) {
    mstts_gemm_desc g;
    memset(&g, 0, sizeof(g));
    g.A = conv_k; g.B = dense_k; g.C = loc_k; g.M = KS; g.N = A; g.K = CH; g.lda = CH; g.ldb = A; g.ldc = A; g.alpha = 1.f; g.split_k = 1; g.batch = 1;
    int rc = mstts_gemm_f32(&g, s);
    if (rc) return rc;
    g.A = conv_b; g.C = loc_b; g.M = 1;
    return mstts_gemm_f32(&g, s);
}

/* unfold d_loc_k[KS,A] (+ d_loc_b == d_score_b[A]) into the gradients of the three reference variables (accumulating):
 *   d_conv_k += d_loc_k . dense_k^T ; d_dense_k += conv_k^T . d_loc_k + conv_b^T (x) d_loc_b ; d_conv_b += d_loc_b . dense_k^T */
extern "C" int mstts_lsa_unfold_location_grad(const float* conv_k, const float* conv_b, const float* dense_k, const float* d_loc_k,
                                              const float* d_loc_b, float* d_conv_k, float* d_conv_b, float* d_dense_k,
                                              int64_t KS, int64_t CH, int64_t A, mstts_stream_t s) {
    mstts_gemm_desc g;
    memset(&g, 0, sizeof(g));
    g.alpha = 1.f; g.split_k = 1; g.batch = 1; g.accumulate = 1;
    g.A = d_loc_k; g.lda = A; g.B = dense_k; g.ldb = A; g.trans_b = 1; g.C = d_conv_k; g.ldc = CH; g.M = KS; g.N = CH; g.K = A;
    int rc = mstts_gemm_f32(&g, s); if (rc) return rc;
    g.A = d_loc_b; g.C = d_conv_b; g.M = 1;
    rc = mstts_gemm_f32(&g, s); if (rc) return rc;
    memset(&g, 0, sizeof(g));
    g.alpha = 1.f; g.split_k = 1; g.batch = 1; g.accumulate = 1;
    g.A = conv_k; g.lda = CH; g.trans_a = 1; g.B = d_loc_k; g.ldb = A; g.C = d_dense_k; g.ldc = A; g.M = CH; g.N = A; g.K = KS;
    rc = mstts_gemm_f32(&g, s); if (rc) return rc;
    g.A = conv_b; g.lda = CH; g.B = d_loc_b; g.K = 1;
    return mstts_gemm_f32(&g, s);
}
