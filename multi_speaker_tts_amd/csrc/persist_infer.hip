// Persistent FREE-RUNNING decoder loop for gfx950: every step of Decoder_Dynamic_Decode.body in inference mode (Modules.py:397-443 with
// Decoder_Helper.next_inputs' inference branch :212-237 - the step's own output frame goes through the prenet, dropout always on
// :239-255, and becomes the next input; a row is finished once its stop logit is >= 0 or time reaches Max_Inference_Length, finished
// flags are OR-accumulated :409 and the loop ends when every row is finished :395) in ONE launch of 256 co-resident workgroups.
//
// Same geometry and hand-off protocol as the teacher-forced loop (persist.hip, persist_common.h): the two cell kernels stay in
// registers as MFMA A operands (workgroup (gi, gj) = reduction slice gi x gate-column slice gj), row gj's keys / values for attention
// slice gi stay in registers / LDS, the data is the flag.  What the free-running loop adds is the frame's way back into cell 0:
//
//   m1_s  --(near)-->  STAGE Q, all 256 workgroups: one more small product on the matrix cores, m1 . [ Wq | Wp_m | Wp_m[:, :80] W1p ]
//                      (query layer, the m1 half of the output projection, and the m1 half of the FIRST PRENET LAYER's pre-activation with
//                      the projection folded in), 15 columns per workgroup, partial over its reduction slice
//         --(far)--->  attention workgroup (row, slice): 16 query units, 12 projection outputs, 32 prenet-1 units, summed over the 8 slices
//   energies, softmax, context slice as in training, and from the SAME alignment row
//         prenet-1 slice  = relu(G + sum_t a[t] U[t, slice] + b) . mask          U = (values Wp_c)[:, :80] W1p, a loop invariant like the keys
//         frame outputs   = m1 part + sum_t a[t] vp[t, 12 of 84] + b             (the step's output; off the chain)
//         --(far, inside the row's 8 workgroups)-->  prenet-1 row (256)  -->  prenet-2 slice (32 units, its kernel slice in LDS)
//         --(near)-->  the cell-0 product workgroups of reduction slice gi, as the prenet rows of step s + 1
//
// so the frame itself never travels: layer 1 of the prenet is a linear function of (m1, alignment) and is evaluated where those already
// are.  (frame W1p = (m1 Wp_m + ctx Wp_c + bp)[:80] W1p re-associated: differences of fp32 rounding order, ~1e-6 relative.)
// The query kernel slice (64 KB of LDS in the training kernel) is gone from the attention workgroups - stage Q holds it as 8 registers
// per lane across the chip - which is what makes room for the prenet operands.
//
// End of the loop: the workgroup that forms a row's stop logit counts the rows that have finished (ctrl[4]); the one that finishes the
// last row stores n + 1 into ctrl[5] (n = that step).  Every workgroup reads the word at the end of each step and leaves at the top of
// step n + 2: by then its own step n + 1 depended on data published after the word (see DESIGN), so all 256 take the same decision; the
// host keeps steps 0 .. n.  Zoneout is deterministic at inference: state' = (1 - z)(new - old) + old (ZoneoutLSTMCell.py:259-264).
#include "persist_fwd_parts.h"

namespace mstts {

// ring sizes in floats per slot.  Near rings (an XCD's slice group may keep them in its L2) first.
constexpr long IXCTX = 8L * 128 * 24, IXACT = 8L * 128 * 32, IXPRE = 8L * 128 * 8, IXPART = 256L * 8 * 2 * 256, IXQP = 32L * 8 * 512, IXEN = 32L * 8 * PTMAX,
               IXL1 = 32L * 256;
constexpr long IOFF_CTX = 0, IOFF_M0 = IOFF_CTX + PRING * IXCTX, IOFF_H0 = IOFF_M0 + PRING * IXACT, IOFF_H1 = IOFF_H0 + PRING * IXACT,
               IOFF_MQ = IOFF_H1 + PRING * IXACT, IOFF_PRE = IOFF_MQ + PRING * IXACT, IOFF_NEAR_END = IOFF_PRE + PRING * IXPRE,
               IOFF_QP = IOFF_NEAR_END, IOFF_EN = IOFF_QP + PRING * IXQP, IOFF_L1 = IOFF_EN + PRING * IXEN, IOFF_P0 = IOFF_L1 + PRING * IXL1,
               IOFF_P1 = IOFF_P0 + PRING * IXPART, IXCH_FLOATS = IOFF_P1 + PRING * IXPART;
constexpr int INP = 84;                          // projection outputs padded to a multiple of 4 (n_mel + 1 = 81)
constexpr int IPC = 12;                          // projection outputs per attention slice (8 x 12 = 96 >= 84)
// LDS layout (floats); small hot arrays first (DS immediate offsets reach 64 KB).  TT = 128 or 256 encoder positions: the value / projected-value /
// prenet-1 slices in LDS always cover positions 0 .. 127; with TT = 256 the rows from 128 on are read from memory every step (L2 hits)
template <int TT> struct IL {
    static constexpr int I_STG = 0, I_RED = I_STG + 128 * LA, I_TR = I_RED + 4 * 2 * 256, I_QP = I_TR + 2 * 128, I_QS = I_QP + 512, I_EN = I_QS + 64,
              I_CUM = I_EN + 8 * TT, I_A = I_CUM + TT + 48, I_CO = I_A + TT, I_CO2 = I_CO + 4 * 96, I_L1 = I_CO2 + 4 * 32, I_PR = I_L1 + 256,
              I_FR = I_PR + 8 * 32, I_BS = I_FR + 64, I_LK = I_BS + 80, I_FLAG = I_LK + 32 * 16, I_STAMP = I_FLAG + 8, I_VAL = I_STAMP + 2 * 24,
              I_VP = I_VAL + PT * 96, I_U = I_VP + PT * IPC, I_W2 = I_U + PT * 32, I_FLOATS = I_W2 + 256 * 32;
    static_assert(I_FLOATS * 4 <= 160 * 1024, "LDS budget");
};
constexpr int INSTAMP = 24;

struct PersistInfer {
    const float* w0pk; const float* w1pk; const float* wqppk;
    const float* b0; const float* b1;
    const float* pre0;                           // [B, 256] prenet output of the all-zero start frame (Modules.py:178-185), masks of step 0 applied
    const uint8_t* pm0; const uint8_t* pm1; float inv_keep;      // prenet keep-masks [S, B, 256] (step s holds the masks of the prenet that FEEDS step s)
    const float* bf;                             // [256] bias of the folded first layer: bp[:80] . W1p + b1p
    const float* w2; const float* b2;            // second prenet layer [256, 256], [256]
    const float* u;                              // [B, T, 256] = (values . Wp_c)[:, :, :80] . W1p
    const float* vp;                             // [B, T, 84]  = values . Wp_c (padded columns zero)
    const float* bp;                             // [84] projection bias, padded
    float keep;                                  // 1 - zoneout rate
    const float* keys; const float* values; const int32_t* lengths;
    const float* loc_k; const float* loc_b; const float* score_w; const float* score_b;
    int B, S, T, NM;
    float* linear; float* stop; float* align_hist;
    float* xch; unsigned* ctrl;                  // ctrl[0] arrivals, [1] abort code, [2] workgroups that left in order, [3] near groups, [4] finished rows, [5] n + 1
    unsigned long long* stamps;
    int fail_step; int near_xcd;
};

// NT = row tiles of 16 in the batch: a batch of at most 16 rows (BASELINE configs[3]) runs every product, partial-sum exchange and stage-Q
// product on ONE tile - half the matrix-core work and half the bytes of the two partial-sum exchanges, the largest transfers of a step
template <bool PROF, int NT, int TT>
__global__ __launch_bounds__(PTH) void persist_infer_kernel(PersistInfer d) {
    typedef IL<TT> Y;
    constexpr int I_STG = Y::I_STG, I_RED = Y::I_RED, I_TR = Y::I_TR, I_QP = Y::I_QP, I_QS = Y::I_QS, I_EN = Y::I_EN, I_CUM = Y::I_CUM, I_A = Y::I_A, I_CO = Y::I_CO,
                  I_CO2 = Y::I_CO2, I_L1 = Y::I_L1, I_PR = Y::I_PR, I_FR = Y::I_FR, I_BS = Y::I_BS, I_LK = Y::I_LK, I_FLAG = Y::I_FLAG, I_STAMP = Y::I_STAMP,
                  I_VAL = Y::I_VAL, I_VP = Y::I_VP, I_U = Y::I_U, I_W2 = Y::I_W2;
    constexpr int NH = TT / 128;                  // halves of 128 encoder positions
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int g0 = blockIdx.x, tid0 = threadIdx.x, wave0 = __builtin_amdgcn_readfirstlane(tid0 >> 6);
    int g = g0, gi = g & 7, gj = g >> 3;
    int tid = tid0, lane = tid & 63, wave = wave0;
    const int B = d.B, S = d.S, T = d.T;
    const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc(d.xch, 0, (int)(IXCH_FLOATS * 4), 0x00020000);
    unsigned* sflag = reinterpret_cast<unsigned*>(sm + I_FLAG);     // [0] abort seen, [1] near group, [2] leave (loop ended), [3] this row has finished
    float* stg = sm + I_STG;

    // ---------------- start rendezvous
    if (d.near_xcd) persist_scrub(xr, IOFF_CTX, IOFF_NEAR_END - IOFF_CTX, g0, tid);
    __syncthreads();
    if (tid == 0) {
        const int rz = persist_rendezvous(d.ctrl, g0);
        sflag[0] = rz == 0 ? 1u : 0u;
        sflag[1] = (rz == 2 && d.near_xcd) ? 1u : 0u;
        sflag[2] = 0u; sflag[3] = 0u;
        if (rz == 2 && g0 < 8) __hip_atomic_fetch_add(d.ctrl + 3, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __syncthreads();
    if (sflag[0]) return;
    const bool near = sflag[1] != 0;

    // ---------------- once: this workgroup's constants
    //   w0[ks]: k-steps 0..23 context rows of reduction slice gi, 24..31 its prenet rows, 32..63 its h0 rows; w1[ks]: 0..31 m0 rows, 32..63 h1 rows
    //   wq[r]:  stage Q, k-steps 8 (wave >> 1) + r of the m1 slice, 16 columns = 4 query units | 3 projection outputs | 8 prenet-1 units | 0
    float w0[64], w1[64], wq[8];
    {
        const float* p0 = d.w0pk + ((long)(g * 8 + wave) * 64) * 64 + lane;
#pragma unroll
        for (int r = 0; r < 64; ++r) w0[r] = p0[r * 64];
        const float* p1 = d.w1pk + ((long)(g * 8 + wave) * 64) * 64 + lane;
#pragma unroll
        for (int r = 0; r < 64; ++r) w1[r] = p1[r * 64];
        const float* pq = d.wqppk + ((long)(g * 8 + wave) * 8) * 64 + lane;
#pragma unroll
        for (int r = 0; r < 8; ++r) wq[r] = pq[r * 64];
    }
    // attention role: row ab = gj, slice gi: query units / key columns 16 gi .., value columns 96 gi .., projection outputs 12 gi .., prenet units 32 gi ..
    int ab = gj;
    const bool arow = ab < B;
    const int alen = arow ? (d.lengths ? d.lengths[ab] : T) : 0;
    int ak = tid & 15, atg = tid >> 4;
    float kreg[4 * NH];
    float asb = 0.f, awk = 0.f;
    {
#pragma unroll
        for (int m = 0; m < 4 * NH; ++m) {
            const int t = 128 * (m >> 2) + 4 * atg + (m & 3);
            kreg[m] = (arow && t < T) ? d.keys[((long)ab * T + t) * PA + 16 * gi + ak] : 0.f;
        }
        asb = d.score_b[16 * gi + ak] + d.loc_b[16 * gi + ak];
        awk = d.score_w[16 * gi + ak];
        for (int x = tid; x < 32 * 16; x += PTH) sm[I_LK + x] = (x < PKS * 16) ? d.loc_k[(x >> 4) * PA + 16 * gi + (x & 15)] : 0.f;
        for (int x = tid; x < PT * 96; x += PTH) {
            const int t = x / 96, c = x - t * 96;
            sm[I_VAL + x] = (arow && t < alen && t < T) ? d.values[((long)ab * T + t) * PM + 96 * gi + c] : 0.f;
        }
        for (int x = tid; x < PT * IPC; x += PTH) {
            const int t = x / IPC, o = x - t * IPC;
            sm[I_VP + x] = (arow && t < alen && t < T && IPC * gi + o < INP) ? d.vp[((long)ab * T + t) * INP + IPC * gi + o] : 0.f;
        }
        for (int x = tid; x < PT * 32; x += PTH) {
            const int t = x >> 5, c = x & 31;
            sm[I_U + x] = (arow && t < alen && t < T) ? d.u[((long)ab * T + t) * 256 + 32 * gi + c] : 0.f;
        }
        for (int x = tid; x < 256 * 32; x += PTH) sm[I_W2 + x] = d.w2[(long)(x >> 5) * 256 + 32 * gi + (x & 31)];
        for (int x = tid; x < TT + 48; x += PTH) sm[I_CUM + x] = 0.f;
        for (int x = tid; x < 4 * 2 * 256; x += PTH) sm[I_RED + x] = 0.f;      // (NT = 1: the tile-1 gate sums are never written; zeros keep those rows' states finite)
        // bias slices of this attention slice: folded prenet-1 bias (32), prenet-2 bias (32), projection bias (12)
        if (tid < 32) { sm[I_BS + tid] = d.bf[32 * gi + tid]; sm[I_BS + 32 + tid] = d.b2[32 * gi + tid]; }
        if (tid < 16) sm[I_BS + 64 + tid] = (tid < IPC && IPC * gi + tid < INP) ? d.bp[IPC * gi + tid] : 0.f;
    }
    float lkb[8];
    __syncthreads();
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) lkb[ks] = sm[I_LK + (4 * ks + (lane >> 4)) * 16 + (lane & 15)];
    // cell-update role (waves 0, 1): row er, hidden unit eu = 4 g + (lane >> 4); the states stay in registers for all steps
    int et = wave & 1, er = 16 * et + (lane & 15), ee = lane >> 4, eu = 4 * g + ee;
    bool ew = wave < 2;
    float c0s = 0.f, h0s = 0.f, c1s = 0.f, h1s = 0.f;
    float b1v[4], b0v[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) { b1v[q] = d.b1[q * PH + eu]; b0v[q] = d.b0[q * PH + eu]; }
    pf32x4 acc0[2], acc1[2];
#pragma unroll
    for (int b = 0; b < 2; ++b) { acc0[b] = (pf32x4){0.f, 0.f, 0.f, 0.f}; acc1[b] = acc0[b]; }
    unsigned long long* sstamp = reinterpret_cast<unsigned long long*>(sm + I_STAMP);
    unsigned tprev = 0;
    if (PROF && tid < INSTAMP) sstamp[tid] = 0;
#define PSTAMP(idx) do { if (PROF && tid == 0) { const unsigned n__ = (unsigned)wall_clock64(); sstamp[idx] += (unsigned)(n__ - tprev); tprev = n__; } } while (0)
#define PABORT_CHECK() do { __syncthreads(); if (sflag[0]) return; } while (0)
#define PFAIL() do { sflag[0] = 1; unsigned z__ = 0u; __hip_atomic_compare_exchange_strong(d.ctrl + 1, &z__, 2u, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); } while (0)
#define PUBLISH_PARTIAL(OFFP, ACC)                                                                                              \
    _Pragma("unroll") for (int t = 0; t < NT; ++t) {                                                                             \
        const long piece = ((long)(gj * 8 + wave) * 8 + gi) * 2 + t;                                                            \
        xpublish(xr, (unsigned)(((OFFP) + slot * IXPART + piece * 256) * 4 + 16 * lane), ACC[t], gen);                          \
        ACC[t] = (pf32x4){0.f, 0.f, 0.f, 0.f};                                                                                  \
    }
    // (NT = 1: the tile-1 pieces do not exist - both requests of a source point at its tile-0 piece, the tile-1 sums stay at their initial zero)
#define ISSUE_PARTIALS(OFFP)                                                                                                    \
    if (wave < 4) {                                                                                                             \
        _Pragma("unroll") for (int m = 0; m < 4; ++m)                                                                           \
            poff[m] = (unsigned)(((OFFP) + slot * IXPART + (((long)g * 8 + 2 * wave + (m >> 1)) * 2 + (NT == 2 ? (m & 1) : 0)) * 256) * 4 + 16 * lane); \
        issue<4>(xr, poff, pv);                                                                                                 \
    }
#define COMPLETE_PARTIALS()                                                                                                     \
    if (wave < 4) {                                                                                                             \
        const unsigned gens__[4] = {gen, gen, gen, gen};                                                                        \
        if (!complete<4>(xr, poff, pv, d.ctrl, gens__)) PFAIL();                                                                \
        *reinterpret_cast<pf32x4*>(sm + I_RED + ((wave * 2 + 0) * 64 + lane) * 4) = pv[0] + pv[2];                              \
        if (NT == 2) *reinterpret_cast<pf32x4*>(sm + I_RED + ((wave * 2 + 1) * 64 + lane) * 4) = pv[1] + pv[3];                 \
    }
#define SUM_PARTIALS()                                                                                                          \
    ((*reinterpret_cast<const pf32x4*>(sm + I_RED + ((0 * 2 + et) * 64 + lane) * 4) + *reinterpret_cast<const pf32x4*>(sm + I_RED + ((1 * 2 + et) * 64 + lane) * 4)) + \
     (*reinterpret_cast<const pf32x4*>(sm + I_RED + ((2 * 2 + et) * 64 + lane) * 4) + *reinterpret_cast<const pf32x4*>(sm + I_RED + ((3 * 2 + et) * 64 + lane) * 4)))
    // zero context / prenet slice for the rows past the batch (published every step so that the cells' waits complete)
#define PUBLISH_CTX(VAL)                                                                                                        \
    do {                                                                                                                        \
        const int qq = tid / 6, k4 = tid - qq * 6;                                                                              \
        const int rho = ((qq * 2 + (ab >> 4)) * 16) + (ab & 15);                                                                \
        xpublish_near(xr, (unsigned)((IOFF_CTX + slot * IXCTX + gi * 3072L + rho * 24 + 4 * k4) * 4), (VAL), gen, near);        \
    } while (0)
#define PUBLISH_PRE(VAL)                                                                                                        \
    do {                                                                                                                        \
        const int qq = tid >> 1, hf = tid & 1;                                                                                  \
        const int rho = ((qq * 2 + (ab >> 4)) * 16) + (ab & 15);                                                                \
        xpublish_near(xr, (unsigned)((IOFF_PRE + slot * IXPRE + gi * 1024L + rho * 8 + 4 * hf) * 4), (VAL), gen, near);         \
    } while (0)
    __syncthreads();
    if (PROF && tid == 0) tprev = (unsigned)wall_clock64();

    // step 0's prenet slice (prenet of the zero frame, from the host): staging row rho = tid >> 1, half tid & 1
    pf32x4 prv = {0.f, 0.f, 0.f, 0.f};
    {
        const unsigned rho = ((unsigned)tid0 & 255u) >> 1, pb = 16 * ((rho >> 4) & 1u) + (rho & 15u);
        const pf32x4 v = *reinterpret_cast<const pf32x4*>(d.pre0 + (long)(pb < (unsigned)B ? pb : 0u) * 256 + 32 * (g0 & 7) + 8 * (rho >> 5) + 4 * ((unsigned)tid0 & 1u));
        prv = pb < (unsigned)B ? v : (pf32x4){0.f, 0.f, 0.f, 0.f};
    }
    // keep-masks of the prenet that feeds step s + 1 (HBM-cold: requested one step ahead, by every thread, unconditionally)
    unsigned m0w = 0, m1w = 0;                   // 4 mask bytes each: layer 1 units 32 gi + 4 (tid & 7) .. + 3 (threads 32..39 use them), layer 2 units 32 gi + 4 (tid & 7) ..
#define LOAD_MASKS(ST) do { const long o__ = ((long)(ST) * B + (ab < B ? ab : 0)) * 256 + 32 * (g0 & 7) + 4 * (tid0 & 7);                       \
        m0w = *reinterpret_cast<const unsigned*>(d.pm0 + o__); m1w = *reinterpret_cast<const unsigned*>(d.pm1 + o__); } while (0)
    LOAD_MASKS(S > 1 ? 1 : 0);
    unsigned dv = 0;                             // the loop-end word as read at the end of the previous step

    for (int s = 0; s < S; ++s) {
        const unsigned slot = (unsigned)s & 3u, pslot = (unsigned)(s + 3) & 3u;
        const unsigned gen = ((unsigned)s >> 2) & 1u, pgen = ((unsigned)(s - 1) >> 2) & 1u;
        tid = tid0; g = g0; wave = wave0;
        asm volatile("" : "+v"(tid));
        asm volatile("" : "+s"(g), "+s"(wave));
        lane = tid & 63; gi = g & 7; gj = g >> 3; ab = gj; ak = tid & 15; atg = tid >> 4;
        et = wave & 1; er = 16 * et + (lane & 15); ee = lane >> 4; eu = 4 * g + ee;
        ew = wave < 2;
        if (tid == 0) {
            if (dv != 0u && (unsigned)s > dv) sflag[2] = 1;          // every row finished at step dv - 1 and step dv has been run: leave (see header)
            if (s == d.fail_step && g == 0) {
                __hip_atomic_store(d.ctrl + 1, 3u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                sflag[0] = 1;
            }
        }
        unsigned soff[2], poff[4], roff[1];
        pf32x4 sv[2], pv[4], rv[1];
        // ================= A: cell 0: context rows, then the prenet rows (the prenet of this step's input frame left its producers ~2 us
        // behind the context: it arrives under the context product)
        if (s > 0) {
            __syncthreads();
            if (sflag[2]) break;
            PSTAMP(0);
            slice_issue<6>(xr, IOFF_CTX + pslot * IXCTX + gi * 3072L, tid, soff, sv);
            if (!slice_complete<6, LA>(xr, stg, tid, soff, sv, d.ctrl, pgen)) PFAIL();
            PABORT_CHECK();
            PSTAMP(1);
            mfma_part<0, 6, LA, 0, 64, NT>(w0, stg, lane, acc0);
            PSTAMP(2);
            if (tid < 256) {
                roff[0] = (unsigned)((IOFF_PRE + pslot * IXPRE + gi * 1024L) * 4 + 16 * tid);
                issue<1>(xr, roff, rv);
                { const unsigned g1[1] = {pgen}; if (!complete<1>(xr, roff, rv, d.ctrl, g1)) PFAIL(); }
                *reinterpret_cast<pf32x4*>(stg + (tid >> 1) * LA + 24 + 4 * (tid & 1)) = rv[0];
            }
            PABORT_CHECK();
            PSTAMP(3);
        } else {
            if (tid < 256) *reinterpret_cast<pf32x4*>(stg + (tid >> 1) * LA + 24 + 4 * (tid & 1)) = prv;
            PABORT_CHECK();
        }
        mfma_part<6, 8, LA, 0, 64, NT>(w0, stg, lane, acc0);
        PUBLISH_PARTIAL(IOFF_P0, acc0)
        PSTAMP(4);
        // in the shadow of the partial-gates hand-off: h1_{s-1} . W1[h rows]
        if (s > 0) {
            slice_issue<8>(xr, IOFF_H1 + pslot * IXACT + gi * 4096L, tid, soff, sv);
            __syncthreads();                                         // the context / prenet rows are consumed by every wave
            if (!slice_complete<8, LA>(xr, stg, tid, soff, sv, d.ctrl, pgen)) PFAIL();
            PABORT_CHECK();
            mfma_part<0, 8, LA, 32, 64, NT>(w1, stg, lane, acc1);
        }
        PSTAMP(5);
        // ================= B: sum of the eight partials, cell-0 update
        ISSUE_PARTIALS(IOFF_P0)
        COMPLETE_PARTIALS()
        PABORT_CHECK();
        PSTAMP(6);
        {
            const pf32x4 gs = SUM_PARTIALS();
            const CellOut o = cell_update(gs, b0v, c0s, h0s, d.keep, d.keep);
            if (ew) {
                sm[I_TR + er * 4 + ee] = o.m;
                sm[I_TR + 128 + er * 4 + ee] = h0s;
            }
            __syncthreads();
            if (tid < 64) {
                const int arr = tid >> 5, row = tid & 31;
                const pf32x4 val = *reinterpret_cast<const pf32x4*>(sm + I_TR + arr * 128 + row * 4);
                const int rho = (((gj & 3) * 2 + (row >> 4)) * 16) + (row & 15);
                const long base = arr ? IOFF_H0 : IOFF_M0;
                const long o2 = gi * 4096L + rho * 32 + 4 * (gj >> 2);
                xpublish_near(xr, (unsigned)((base + slot * IXACT + o2) * 4), val, gen, near);
            }
        }
        PSTAMP(7);
        // ================= C: cell 1, input rows (m0_s)
        slice_issue<8>(xr, IOFF_M0 + slot * IXACT + gi * 4096L, tid, soff, sv);
        if (!slice_complete<8, LA>(xr, stg, tid, soff, sv, d.ctrl, gen)) PFAIL();
        PABORT_CHECK();
        PSTAMP(8);
        slice_issue<8>(xr, IOFF_H0 + slot * IXACT + gi * 4096L, tid, soff, sv);     // h0_s left its producers together with m0_s: it arrives under the product
        mfma_part<0, 8, LA, 0, 64, NT>(w1, stg, lane, acc1);
        PUBLISH_PARTIAL(IOFF_P1, acc1)
        PSTAMP(9);
        // in the shadow of the partial-gates hand-off: h0_s staged, the whole of h0_s . W0[h rows] for step s + 1 (the staging buffer has to
        // be free for the m1 slice right behind the cell-1 update)
        __syncthreads();                                             // m0 is consumed by every wave
        if (!slice_complete<8, LA>(xr, stg, tid, soff, sv, d.ctrl, gen)) PFAIL();
        PABORT_CHECK();
        mfma_part<0, 8, LA, 32, 64, NT>(w0, stg, lane, acc0);
        PSTAMP(10);
        // ================= D: sum of the eight partials, cell-1 update
        ISSUE_PARTIALS(IOFF_P1)
        COMPLETE_PARTIALS()
        PABORT_CHECK();
        PSTAMP(11);
        {
            const pf32x4 gs = SUM_PARTIALS();
            const CellOut o = cell_update(gs, b1v, c1s, h1s, d.keep, d.keep);
            if (ew) {
                sm[I_TR + er * 4 + ee] = o.m;
                sm[I_TR + 128 + er * 4 + ee] = h1s;
            }
            __syncthreads();
            if (tid < 64) {         // m1 (the cell stack's output) and h1': both into slice gi of the consumers' rings, like m0 / h0
                const int arr = tid >> 5, row = tid & 31;
                const pf32x4 val = *reinterpret_cast<const pf32x4*>(sm + I_TR + arr * 128 + row * 4);
                const int rho = (((gj & 3) * 2 + (row >> 4)) * 16) + (row & 15);
                const long base = arr ? IOFF_H1 : IOFF_MQ;
                const long o2 = gi * 4096L + rho * 32 + 4 * (gj >> 2);
                xpublish_near(xr, (unsigned)((base + slot * IXACT + o2) * 4), val, gen, near);
            }
        }
        PSTAMP(12);
        // ================= Q: m1 slice . [Wq | Wp_m | Wp_m W1p] columns of this workgroup; wave = (row tile wave & 1, k-quarter wave >> 1)
        slice_issue<8>(xr, IOFF_MQ + slot * IXACT + gi * 4096L, tid, soff, sv);
        if (!slice_complete<8, LA>(xr, stg, tid, soff, sv, d.ctrl, gen)) PFAIL();
        PABORT_CHECK();
        PSTAMP(13);
        {
            pf32x4 qa = {0.f, 0.f, 0.f, 0.f};
            const int row = ((lane >> 4) * 2 + (wave & 1)) * 16 + (lane & 15), kq = wave >> 1;
            if (NT == 2 || (wave & 1) == 0) {
#pragma unroll
                for (int k4 = 0; k4 < 2; ++k4) {
                    const pf32x4 x = *reinterpret_cast<const pf32x4*>(stg + row * LA + 4 * (2 * kq + k4));
#pragma unroll
                    for (int e = 0; e < 4; ++e) qa = PMFMA(wq[4 * k4 + e], x[e], qa);
                }
            }
            *reinterpret_cast<pf32x4*>(sm + I_RED + (wave * 64 + lane) * 4) = qa;
            __syncthreads();
            if (tid < 64 * NT) {        // (row tile t, lane l): columns 4 (l >> 4) .. + 3 of row 16 t + (l & 15), summed over the four k-quarters
                const int t = tid >> 6, l = tid & 63;
                const pf32x4 v = (*reinterpret_cast<const pf32x4*>(sm + I_RED + ((0 * 2 + t) * 64 + l) * 4) + *reinterpret_cast<const pf32x4*>(sm + I_RED + ((1 * 2 + t) * 64 + l) * 4)) +
                                 (*reinterpret_cast<const pf32x4*>(sm + I_RED + ((2 * 2 + t) * 64 + l) * 4) + *reinterpret_cast<const pf32x4*>(sm + I_RED + ((3 * 2 + t) * 64 + l) * 4));
                const int r = 16 * t + (l & 15);
                // granule of 16 floats per (row, attention slice gj >> 2, source slice gi, quarter gj & 3): the reader's 128 pieces are contiguous
                const long o = ((((long)r * 8 + (gj >> 2)) * 8 + gi) * 4 + (gj & 3)) * 16 + 4 * (l >> 4);
                xpublish(xr, (unsigned)((IOFF_QP + slot * IXQP + o) * 4), v, gen);
            }
        }
        PSTAMP(14);
        // ================= E: attention: the row's query units (and projection / prenet partial sums) arrive, partial energies
        if (arow) {
            if (tid < 128) {
                roff[0] = (unsigned)((IOFF_QP + slot * IXQP + ((long)ab * 8 + gi) * 512) * 4 + 16 * tid);
                issue<1>(xr, roff, rv);
            }
            pf32x4 locv[NH];                                         // while the granules are in flight: the location filter over the cumulative alignment (persist.hip)
#pragma unroll
            for (int hh = 0; hh < NH; ++hh) {
                locv[hh] = (pf32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int ks = 0; ks < 8; ++ks)
                    locv[hh] = PMFMA(sm[I_CUM + 128 * hh + 16 * wave + (lane & 15) + 4 * ks + (lane >> 4)], lkb[ks], locv[hh]);
            }
            if (tid < 128) {
                { const unsigned g1[1] = {gen}; if (!complete<1>(xr, roff, rv, d.ctrl, g1)) PFAIL(); }
                *reinterpret_cast<pf32x4*>(sm + I_QP + 4 * tid) = rv[0];
            }
            PABORT_CHECK();
            PSTAMP(15);
            if (tid < 64) {         // (quarter j = tid >> 4, column c = tid & 15): sum over the 8 source slices, fixed order
                float v = 0.f;
#pragma unroll
                for (int k = 0; k < 8; ++k) v += sm[I_QP + (k * 4 + (tid >> 4)) * 16 + (tid & 15)];
                sm[I_QS + tid] = v;
            }
            __syncthreads();
            {
                const float qk = sm[I_QS + (ak >> 2) * 16 + (ak & 3)] + asb;
#pragma unroll
                for (int hh = 0; hh < NH; ++hh) {
                    const pf32x4 loc = locv[hh];
                    pf32x4 e4;
#pragma unroll
                    for (int m = 0; m < 4; ++m) {
                        float e = awk * tanhf_(kreg[4 * hh + m] + qk + loc[m]);
                        e += dpp_mov<0xB1, 0xf>(0.f, e);
                        e += dpp_mov<0x4E, 0xf>(0.f, e);
                        e += dpp_mov<0x141, 0xf>(0.f, e);
                        e += dpp_mov<0x140, 0xf>(0.f, e);
                        e4[m] = e;
                    }
                    if (ak == 0) {
                        const long o = ((long)ab * 8 + gi) * TT + 128 * hh + 4 * atg;
                        xpublish(xr, (unsigned)((IOFF_EN + slot * IXEN + o) * 4), e4, gen);
                    }
                }
            }
        } else {
            PSTAMP(15);
        }
        PSTAMP(16);
        __syncthreads();
        // ================= F: energies of the row, softmax, cumulative alignment; context slice, prenet-1 slice, frame outputs
        if (arow) {
            if (tid < 256 * NH) {
                roff[0] = (unsigned)((IOFF_EN + slot * IXEN + (long)ab * 8 * TT) * 4 + 16 * tid);
                issue<1>(xr, roff, rv);
                { const unsigned g1[1] = {gen}; if (!complete<1>(xr, roff, rv, d.ctrl, g1)) PFAIL(); }
                *reinterpret_cast<pf32x4*>(sm + I_EN + 4 * tid) = rv[0];
            }
            PABORT_CHECK();
            PSTAMP(17);
            if (wave == 0) {        // masked softmax over the row's TT positions: lane holds positions lane + 64 i
                float ev[2 * NH];
                float mx = -INFINITY;
#pragma unroll
                for (int i = 0; i < 2 * NH; ++i) {
                    float e = 0.f;
#pragma unroll
                    for (int k = 0; k < 8; ++k) e += sm[I_EN + k * TT + 64 * i + lane];
                    ev[i] = (lane + 64 * i < alen) ? e : -INFINITY;
                    mx = fmaxf(mx, ev[i]);
                }
                mx = wave_max(mx);
                float ps = 0.f;
#pragma unroll
                for (int i = 0; i < 2 * NH; ++i) { ev[i] = (lane + 64 * i < alen) ? __expf(ev[i] - mx) : 0.f; ps += ev[i]; }
                const float inv = 1.f / wave_sum(ps);
                float* ah = d.align_hist + ((long)s * B + ab) * T;
#pragma unroll
                for (int i = 0; i < 2 * NH; ++i) {
                    const float a = ev[i] * inv;
                    sm[I_A + 64 * i + lane] = a;
                    sm[I_CUM + 15 + 64 * i + lane] += a;
                    if (gi == 0 && lane + 64 * i < T) ah[lane + 64 * i] = a;
                }
            }
            __syncthreads();
            if (tid < 384) {        // context: 96 columns x 4 position quarters
                const int c = tid % 96, th = tid / 96;
                float acc = 0.f;
#pragma unroll
                for (int t4 = 0; t4 < 8; ++t4) {                     // (the four alignment weights of a group in one 16-byte LDS read)
                    const pf32x4 a4 = *reinterpret_cast<const pf32x4*>(sm + I_A + 32 * th + 4 * t4);
#pragma unroll
                    for (int e = 0; e < 4; ++e) acc += a4[e] * sm[I_VAL + (32 * th + 4 * t4 + e) * 96 + c];
                }
                if (NH > 1) {       // positions 128 + 32 th ..: from memory (zero past the row's length there)
                    const float* vg = d.values + ((long)ab * T + 128 + 32 * th) * PM + 96 * gi + c;
                    const int nt = T - (128 + 32 * th) < 32 ? (T - (128 + 32 * th) > 0 ? T - (128 + 32 * th) : 0) : 32;
#pragma unroll 16
                    for (int t = 0; t < 32; ++t) acc += sm[I_A + 128 + 32 * th + t] * (t < nt ? vg[(long)t * PM] : 0.f);
                }
                sm[I_CO + th * 96 + c] = acc;
            } else {                // the alignment's share of the prenet's first layer: 32 units x 4 position quarters
                const int x = tid - 384, c = x & 31, th = x >> 5;
                float acc = 0.f;
#pragma unroll 16
                for (int t = 0; t < 32; ++t) acc += sm[I_A + 32 * th + t] * sm[I_U + (32 * th + t) * 32 + c];
                if (NH > 1) {
                    const float* ug = d.u + ((long)ab * T + 128 + 32 * th) * 256 + 32 * gi + c;
                    const int nt = T - (128 + 32 * th) < 32 ? (T - (128 + 32 * th) > 0 ? T - (128 + 32 * th) : 0) : 32;
#pragma unroll 16
                    for (int t = 0; t < 32; ++t) acc += sm[I_A + 128 + 32 * th + t] * (t < nt ? ug[(long)t * 256] : 0.f);
                }
                sm[I_CO2 + th * 32 + c] = acc;
            }
            __syncthreads();
            if (tid < 24) {
                const pf32x4 val = (*reinterpret_cast<const pf32x4*>(sm + I_CO + 4 * tid) + *reinterpret_cast<const pf32x4*>(sm + I_CO + 96 + 4 * tid)) +
                                   (*reinterpret_cast<const pf32x4*>(sm + I_CO + 192 + 4 * tid) + *reinterpret_cast<const pf32x4*>(sm + I_CO + 288 + 4 * tid));
                PUBLISH_CTX(val);
            } else if (tid >= 32 && tid < 40) {
                // prenet layer 1, units 32 gi + 4 x .. + 3 of this row: relu(m1 part + alignment part + folded bias) * keep-mask / keep (Modules.py:239-255)
                const int x = tid - 32;
                const pf32x4 al = (*reinterpret_cast<const pf32x4*>(sm + I_CO2 + 4 * x) + *reinterpret_cast<const pf32x4*>(sm + I_CO2 + 32 + 4 * x)) +
                                  (*reinterpret_cast<const pf32x4*>(sm + I_CO2 + 64 + 4 * x) + *reinterpret_cast<const pf32x4*>(sm + I_CO2 + 96 + 4 * x));
                pf32x4 p1;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int c = 4 * x + e;
                    const float pre = sm[I_QS + (c >> 3) * 16 + 7 + (c & 7)] + al[e] + sm[I_BS + c];
                    p1[e] = ((m0w >> (8 * e)) & 0xffu) ? fmaxf(pre, 0.f) * d.inv_keep : 0.f;
                }
                xpublish(xr, (unsigned)((IOFF_L1 + slot * IXL1 + (long)ab * 256 + 32 * gi + 4 * x) * 4), p1, gen);
            } else if (tid >= 64 && tid < 64 + 4 * IPC) {
                // the alignment's share of this slice's 12 projection outputs (the step's frame; nothing waits for it)
                const int x = tid - 64, o = x % IPC, th = x / IPC;
                float acc = 0.f;
#pragma unroll 8
                for (int t = 0; t < 32; ++t) acc += sm[I_A + 32 * th + t] * sm[I_VP + (32 * th + t) * IPC + o];
                if (NH > 1 && IPC * gi + o < INP) {
                    const float* pg = d.vp + ((long)ab * T + 128 + 32 * th) * INP + IPC * gi + o;
                    const int nt = T - (128 + 32 * th) < 32 ? (T - (128 + 32 * th) > 0 ? T - (128 + 32 * th) : 0) : 32;
#pragma unroll 8
                    for (int t = 0; t < 32; ++t) acc += sm[I_A + 128 + 32 * th + t] * (t < nt ? pg[(long)t * INP] : 0.f);
                }
                sm[I_FR + th * IPC + o] = acc;
            }
            PSTAMP(18);
            // ================= G: the row's prenet-1 units (8 slices x 32) -> this slice's 32 prenet-2 units -> cell-0 workgroups of slice gi
            if (tid < 64) {
                roff[0] = (unsigned)((IOFF_L1 + slot * IXL1 + (long)ab * 256) * 4 + 16 * tid);
                issue<1>(xr, roff, rv);
                { const unsigned g1[1] = {gen}; if (!complete<1>(xr, roff, rv, d.ctrl, g1)) PFAIL(); }
                *reinterpret_cast<pf32x4*>(sm + I_L1 + 4 * tid) = rv[0];
            }
            PABORT_CHECK();
            PSTAMP(19);
            if (tid >= 64 && tid < 64 + IPC) {
                // frame outputs IPC gi + o: m1 part (stage Q) + alignment part + bias -> linear [S, B, n_mel] / stop [S, B]
                const int o = tid - 64, idx = IPC * gi + o;
                const float f = sm[I_QS + (o / 3) * 16 + 4 + (o % 3)] + ((sm[I_FR + o] + sm[I_FR + IPC + o]) + (sm[I_FR + 2 * IPC + o] + sm[I_FR + 3 * IPC + o])) +
                                sm[I_BS + 64 + o];
                if (idx < d.NM) d.linear[((long)s * B + ab) * d.NM + idx] = f;
                if (idx == d.NM) {
                    d.stop[(long)s * B + ab] = f;
                    // Modules.py:216-219: finished = stop logit >= 0 or time >= Max_Inference_Length (the last step of the launch), OR-accumulated (:409)
                    if ((f >= 0.f || s >= S - 1) && sflag[3] == 0u) {
                        sflag[3] = 1u;
                        const unsigned before = __hip_atomic_fetch_add(d.ctrl + 4, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        if (before + 1u == (unsigned)B) __hip_atomic_store(d.ctrl + 5, (unsigned)s + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // acknowledged before this workgroup publishes anything else (barrier below)
                    }
                }
            }
            {       // prenet layer 2, units 32 gi + c: 16 k-groups of 16 rows, two per wave
                const int c = tid & 31, kg = tid >> 5;
                float part = 0.f;
#pragma unroll
                for (int k = 0; k < 16; ++k) part += sm[I_L1 + 16 * kg + k] * sm[I_W2 + (16 * kg + k) * 32 + c];
                part += __shfl_xor(part, 32);
                if (lane < 32) sm[I_PR + wave * 32 + lane] = part;
            }
            __syncthreads();
            if (tid < 8) {
                pf32x4 p2;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int c = 4 * tid + e;
                    float v = sm[I_BS + 32 + c];
#pragma unroll
                    for (int u = 0; u < 8; ++u) v += sm[I_PR + u * 32 + c];
                    p2[e] = ((m1w >> (8 * e)) & 0xffu) ? fmaxf(v, 0.f) * d.inv_keep : 0.f;
                }
                PUBLISH_PRE(p2);
            }
        } else {
            PSTAMP(17);
            if (tid < 24) PUBLISH_CTX(((pf32x4){0.f, 0.f, 0.f, 0.f}));
            PSTAMP(18);
            PSTAMP(19);
            if (tid < 8) PUBLISH_PRE(((pf32x4){0.f, 0.f, 0.f, 0.f}));
        }
        PSTAMP(20);
        LOAD_MASKS(s + 2 < S ? s + 2 : S - 1);
        if (tid == 0) dv = __hip_atomic_load(d.ctrl + 5, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (tid == 0) {
        __hip_atomic_fetch_add(d.ctrl + 2, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (PROF && d.stamps) {
#pragma unroll
            for (int x = 0; x < INSTAMP; ++x) d.stamps[(long)g * INSTAMP + x] = sstamp[x];
        }
    }
#undef PSTAMP
#undef LOAD_MASKS
#undef PABORT_CHECK
#undef PFAIL
#undef PUBLISH_PARTIAL
#undef ISSUE_PARTIALS
#undef COMPLETE_PARTIALS
#undef SUM_PARTIALS
#undef PUBLISH_CTX
#undef PUBLISH_PRE
}

// stage-Q operands in the lanes' order: [workgroup g][wave][8 k-steps][lane]: A operand of v_mfma_f32_16x16x4_f32, lane = (column m = lane & 15,
// k = lane >> 4); wave = (row tile, k-quarter kq): k-step ks = 8 kq + r is units unit_of_kstep(gi, ks, q) of m1.  Columns of workgroup (gi, gj),
// attention slice a = gj >> 2, quarter j = gj & 3:  m 0..3 query units 16 a + 4 j + m | m 4..6 projection outputs 12 a + 3 j + (m - 4) |
// m 7..14 folded prenet-1 units 32 a + 8 j + (m - 7) | m 15 zero
__global__ void persist_pack_qp_kernel(const float* __restrict__ wq, const float* __restrict__ wp, int wp_ld, const float* __restrict__ wfm, float* __restrict__ out) {
    const long n = 256L * 8 * 8 * 64;
    for (long p = blockIdx.x * (long)blockDim.x + threadIdx.x; p < n; p += (long)gridDim.x * blockDim.x) {
        long r = p;
        const int lane = (int)(r & 63); r >>= 6;
        const int rr = (int)(r & 7); r >>= 3;
        const int wave = (int)(r & 7), g = (int)(r >> 3);
        const int gi = g & 7, gj = g >> 3, a = gj >> 2, j = gj & 3;
        const int m = lane & 15, q = lane >> 4, ks = 8 * (wave >> 1) + rr;
        const long u = unit_of_kstep(gi, ks, q);
        float v = 0.f;
        if (m < 4) v = wq[u * PA + 16 * a + 4 * j + m];
        else if (m < 7) { const int o = IPC * a + 3 * j + (m - 4); v = o < INP ? wp[u * wp_ld + o] : 0.f; }
        else if (m < 15) v = wfm[u * 256 + 32 * a + 8 * j + (m - 7)];
        out[p] = v;
    }
}

}  // namespace mstts
using namespace mstts;

extern "C" int64_t mstts_persist_infer_ws_bytes(void) { return IXCH_FLOATS * 4; }
extern "C" int64_t mstts_persist_infer_pack_floats(void) { return 256L * 8 * 8 * 64; }

/* 1 when the persistent free-running loop covers this shape on the current device: the reference's widths (cells 1024, prenet 256,
 * memory 768, attention 128 with 31 taps, 80 mel bins), at most 32 rows and 256 encoder positions (beyond 128 the value / projected-value rows from 128 on are re-read from the L2 every step), 256 CUs that each take one workgroup */
extern "C" int32_t mstts_persist_infer_supported(int64_t B, int64_t H, int64_t P, int64_t M, int64_t A, int64_t T, int64_t KS, int64_t n_mel) {
    if (!(B >= 1 && B <= PROWS && H == PH && P == 256 && M == PM && A == PA && T >= 1 && T <= PTMAX && KS == PKS && n_mel == 80)) return 0;
    static int memo[PERSIST_MAX_DEVICES];
    return persist_device_memo(memo, [](int dev) {
        int cus = 0, per_cu = 0;
        if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus < PWG) return false;
        bool ok = true;
        int per = 0;
#define PI_SETUP(P_, N_, T_)                                                                                                                            \
        ok = ok && hipFuncSetAttribute((const void*)persist_infer_kernel<P_, N_, T_>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(IL<T_>::I_FLOATS * 4)) == hipSuccess && \
             hipOccupancyMaxActiveBlocksPerMultiprocessor(&per, (const void*)persist_infer_kernel<P_, N_, T_>, PTH, (size_t)IL<T_>::I_FLOATS * 4) == hipSuccess && per >= 1;
        PI_SETUP(false, 1, 128) PI_SETUP(true, 1, 128) PI_SETUP(false, 2, 128) PI_SETUP(true, 2, 128)
        PI_SETUP(false, 1, 256) PI_SETUP(true, 1, 256) PI_SETUP(false, 2, 256) PI_SETUP(true, 2, 256)
#undef PI_SETUP
        (void)per_cu;
        return ok;
    });
}

extern "C" int mstts_persist_infer_pack(const float* wq, const float* wp_pad, int64_t wp_ld, const float* wfm, float* wqppk, mstts_stream_t s) {
    MSTTS_REQUIRE(wq && wp_pad && wfm && wqppk && wp_ld >= INP, MSTTS_ERR_SHAPE, "persist_infer_pack: null pointer or projection row stride below 84");
    hipLaunchKernelGGL(persist_pack_qp_kernel, dim3(1024), dim3(256), 0, (hipStream_t)s, wq, wp_pad, (int)wp_ld, wfm, wqppk);
    MSTTS_CHECK_LAUNCH("persist_pack_qp");
    return MSTTS_OK;
}

extern "C" int mstts_decoder_infer_persistent(const mstts_decoder_infer_desc* d, const mstts_persist_infer_desc* p, mstts_stream_t s) {
    MSTTS_REQUIRE(d && p && p->w0pk && p->w1pk && p->wqppk && p->pre0 && p->bf && p->u && p->vp && p->bp_pad && p->xch && p->ctrl, MSTTS_ERR_SHAPE,
                  "decoder_infer_persistent: null pointer");
    MSTTS_REQUIRE(d->b0 && d->b1 && d->pw1 && d->pb1 && d->pm0 && d->pm1 && d->linear && d->stop && d->align_hist, MSTTS_ERR_SHAPE,
                  "decoder_infer_persistent: descriptor pointers missing");
    const long B = d->B, S = d->Smax, T = d->lsa.T;
    MSTTS_REQUIRE(d->lsa.B == B && S >= 1 && mstts_persist_infer_supported(B, d->H, d->P, d->lsa.M, d->lsa.A, T, d->lsa.KS, d->n_mel), MSTTS_ERR_SHAPE,
                  "decoder_infer_persistent: shape or device not supported (see mstts_persist_infer_supported)");
    MSTTS_REQUIRE(d->lsa.keys && d->lsa.values && d->lsa.loc_k && d->lsa.loc_b && d->lsa.score_w && d->lsa.score_b, MSTTS_ERR_SHAPE,
                  "decoder_infer_persistent: attention constants missing");
    MSTTS_REQUIRE(aligned16(p->xch) && aligned16(p->pre0) && ((uintptr_t)d->pm0 & 3u) == 0 && ((uintptr_t)d->pm1 & 3u) == 0 && d->prenet_keep > 0.f, MSTTS_ERR_ALIGN,
                  "decoder_infer_persistent: alignment (rings and step-0 prenet 16 bytes, masks 4 bytes)");
    hipStream_t hs = (hipStream_t)s;
    hipError_t e = hipMemsetAsync(p->xch, 0xFF, IXCH_FLOATS * 4, hs);                      // every word "generation 1": stale for the first pass
    if (e == hipSuccess) e = hipMemsetAsync(p->ctrl, 0, PCTRL_WORDS * sizeof(unsigned), hs);
    if (e != hipSuccess) return set_err(MSTTS_ERR_LAUNCH, "decoder_infer_persistent: memset: %s", hipGetErrorString(e));
    PersistInfer a;
    a.w0pk = p->w0pk; a.w1pk = p->w1pk; a.wqppk = p->wqppk; a.b0 = d->b0; a.b1 = d->b1; a.pre0 = p->pre0;
    a.pm0 = d->pm0; a.pm1 = d->pm1; a.inv_keep = 1.f / d->prenet_keep; a.bf = p->bf; a.w2 = d->pw1; a.b2 = d->pb1; a.u = p->u; a.vp = p->vp; a.bp = p->bp_pad;
    a.keep = 1.f - d->zoneout;
    a.keys = d->lsa.keys; a.values = d->lsa.values; a.lengths = d->lsa.lengths;
    a.loc_k = d->lsa.loc_k; a.loc_b = d->lsa.loc_b; a.score_w = d->lsa.score_w; a.score_b = d->lsa.score_b;
    a.B = (int)B; a.S = (int)S; a.T = (int)T; a.NM = (int)d->n_mel;
    a.linear = d->linear; a.stop = d->stop; a.align_hist = d->align_hist;
    a.xch = p->xch; a.ctrl = p->ctrl; a.stamps = (unsigned long long*)p->stamps;
    a.fail_step = p->selftest_fail_step > 0 ? p->selftest_fail_step - 1 : -1; a.near_xcd = p->near_xcd;
#define PI_LAUNCH(N_, T_)                                                                                                               \
    {                                                                                                                                   \
        const size_t lds = (size_t)IL<T_>::I_FLOATS * 4;                                                                                \
        if (p->stamps) hipLaunchKernelGGL((persist_infer_kernel<true, N_, T_>), dim3(PWG), dim3(PTH), lds, hs, a);                      \
        else hipLaunchKernelGGL((persist_infer_kernel<false, N_, T_>), dim3(PWG), dim3(PTH), lds, hs, a);                               \
    }
    // one row tile for batches of at most 16 rows (half the matrix-core work and partial-sum bytes); 128 or 256 encoder positions
    if (B <= 16) { if (T <= 128) PI_LAUNCH(1, 128) else PI_LAUNCH(1, 256) }
    else { if (T <= 128) PI_LAUNCH(2, 128) else PI_LAUNCH(2, 256) }
#undef PI_LAUNCH
    MSTTS_CHECK_LAUNCH("persist_infer");
    return MSTTS_OK;
}
