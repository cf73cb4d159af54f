// Persistent BPTT of the teacher-forced decoder loop for gfx950: the backward pass through ALL S steps of
// Decoder_Dynamic_Decode.body (Modules.py:397-443; ZoneoutLSTMCell.py:228-271; Location_Sensitive_Attention.py:43-85) in ONE
// launch of 256 co-resident workgroups - the mirror image of csrc/persist.hip and built from the same hand-off primitives
// (persist_common.h: write-through stores, L1-bypassing polls, the data is the flag).
//
// Per step (descending) the chain of mstts_decoder_train_bwd is
//     attention backward -> cell-1 update backward -> d[g1] . W1^T -> cell-0 update backward -> d[g0] . W0f^T -> (next step)
// and only half of each product is on it: the rows that give d[m0] / d[ctx]; the rows that give the gradients of the recurrent
// states h1 / h0 are consumed one step later and run in the shadow of the hand-offs.
//
// Work cut.  Workgroup (i, j) = (id & 7, id >> 3):
//   * products: contraction slice i (the 512 gate columns of the 128 hidden units {32 j' + 4 i + e}) x output slice j (cell 1: m0 / h1
//     units 32 j .. 32 j + 31; cell 0: context columns 24 j .. 24 j + 23 and h0 units 32 j ..) of BOTH transposed kernels, in registers
//     for the whole sequence (128 per lane, MFMA A operands).  Each of the 8 waves takes one eighth of the contraction and fetches
//     exactly its own slice of the gate gradients straight into MFMA B-operand registers - no staging, no barrier in front of
//     the product; the eight partial tiles meet in LDS and leave as one [32 k x 32 rows] tile for the 8 workgroups of column j.
//   * cell updates: the 4 hidden units 4 id .. 4 id + 3 of both cells for all 32 rows (same owner as in the forward kernel); the
//     carried state gradients stay in registers.
//   * attention: row j, unit / column slice i as in the forward kernel: 96 value columns (LDS), 16 key / query units, its slice of
//     the query kernel (LDS), and the part of G = dL/d(cumulative alignment) that its own 16 units contribute - the row's
//     d_alignment = sum over the 8 slices of (values_i . d_ctx_i + G_i), so ONE exchange per step carries both.
// Exact fp32, fixed summation order (deterministic).  Bounded waits / abort word / start rendezvous as in the forward kernel; on
// abort the host re-runs mstts_decoder_train_bwd.
#include "persist_fwd_parts.h"        // (the bf16 operand types and the 16x16x32 bf16 MFMA macro; persist_common.h through it)

#ifndef BG0_LATE
#define BG0_LATE 10             // ticks (10 ns) between the cell-0 update's publications and the gather of the gate gradients for the d[g0] product
#endif
// Round 6 (profiles/r06_ab_bptt_*.txt): the fp32 instantiations run the on-chain half of the cell-1 data-gradient product as the exact
// three-way bf16 split (BSPLIT_C1), with the registers for its third plane found by moving the query layer's data-gradient product to the cell-1 owners
// (BPTT_QOWN: frees the 64 KB query-kernel slice in LDS) and the recurrent-state half of the cell-0 kernel into that LDS (BPTT_W0LDS).  Each is a build
// option so that the A/B stays reproducible: -DBPTT_QOWN=0 is round 5's kernel.
#ifndef BSPLIT_C1
#define BSPLIT_C1 1             // 1 = the on-chain half of the cell-1 data-gradient product as an exact three-way bf16 split, six products on v_mfma_f32_16x16x32_bf16 (needs BPTT_QOWN + BPTT_W0LDS for its registers: 14 - 44 spilled without)
#endif
#ifndef BPTT_QOWN
#define BPTT_QOWN 1             // 1 = (fp32 instantiations) the query-layer data-gradient product at the cell-1 OWNERS (each attention workgroup publishes its 16 dq values once, all owners
                                // read all of them) instead of at the attention workgroups (a personal 16-byte piece per owner): frees the 64 KB query-kernel slice in LDS
#endif
#ifndef BPTT_W0LDS
#define BPTT_W0LDS 1            // with BPTT_QOWN, fp32: the recurrent-state half of the cell-0 kernel (the shadow product's A operands) in that LDS instead of 32 registers
#endif
#ifndef BPTT_KEYS_PER_STEP
#define BPTT_KEYS_PER_STEP 0    // 1 = the attention slice's key / score constants (6 registers) requested again at the end of every step (L2 hits) instead of
                                // living in registers across the two products of the step
#endif
#ifndef BPTT_LATE_OP0
#define BPTT_LATE_OP0 0         // 1 = the cell-0 update's packed operands requested behind the cell-1 shadow product (in front of the d_m0 wait) instead of in front of it
#endif
#ifndef BSPLIT_C0
#define BSPLIT_C0 0             // like BSPLIT_C1 for the on-chain half of the cell-0 product (context columns)
#endif
namespace mstts {

// x[0..7] -> the three bf16 planes (exact: both remainders are representable in fp32)
__device__ __forceinline__ void bsp_split8(const float (&x)[8], pbf16x8& hi, pbf16x8& mid, pbf16x8& lo) {
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        hi[e] = (__bf16)x[e]; const float r1 = x[e] - (float)hi[e];
        mid[e] = (__bf16)r1; lo[e] = (__bf16)(r1 - (float)mid[e]);
    }
}

// ring sizes in floats per slot
constexpr long BDG = 8L * 8 * 2 * 4 * 256;       // gate gradients of one cell: [slice 8][eighth 8][row tile 2][group 4][lane 64][4 units]
constexpr long BPART = 256L * 8 * 32 * 4;        // partial product tiles: [reducer 256][source 8][row 32][4 units]
constexpr long BPCTX = 32L * 192 * 8 * 4;        // partial d_ctx: [row 32][4-column block 192][source 8][4]
constexpr long BDM1 = 32L * 8 * PH;              // query-layer data gradient: [owner 256][slice 8][row 32][4 units]
constexpr long BDA = 32L * 8 * PTMAX;            // partial d_alignment: [row 32][slice 8][TT]
constexpr long BO_DG1 = 0, BO_DG0 = BO_DG1 + PRING * BDG, BO_PM0 = BO_DG0 + PRING * BDG, BO_PH1 = BO_PM0 + PRING * BPART,
               BO_PH0 = BO_PH1 + PRING * BPART, BO_PCTX = BO_PH0 + PRING * BPART, BO_DM1 = BO_PCTX + PRING * BPCTX,
               BO_DA = BO_DM1 + PRING * BDM1, BXCH_FLOATS = BO_DA + PRING * BDA;
// LDS layout (floats); small arrays first (DS immediate offsets reach 64 KB).  TT = 128 or 256 encoder positions (the kernel's instantiations);
// the transposed value slice in LDS always covers positions 0 .. 127, with TT = 256 the positions from 128 on are read from memory (L2 hits).
template <int TT> struct BL {
    static constexpr int B_RED = 0,                  // [8 waves][4 tiles][64 lanes][4]: product partials; the attention phases use it as scratch
              B_G = B_RED,                           //   [TT + 32 padded positions][16 units] energy gradients of this slice (attention only)
              B_PC = B_G + (TT + 32) * 16,           //   [192 pieces][4] partial d_ctx as fetched
              B_DA = B_PC + 192 * 4,                 //   [8 slices][TT] partial d_alignment as fetched
              B_PD = B_DA + 8 * TT,                  //   [4 column quarters][TT] values . d_ctx
              B_SCR_END = B_PD + 4 * TT,
              B_TR = B_SCR_END > B_RED + 8 * 4 * 256 ? B_SCR_END : B_RED + 8 * 4 * 256,   // [3][128] transposes between the (unit, row) and (row, 4 units) thread layouts + [512] gate transpose
              B_GP = B_TR + 3 * 128 + 512,           // [TT] this slice's part of G
              B_A = B_GP + TT, B_CUM = B_A + TT, B_DE = B_CUM + TT + 48, B_DC = B_DE + TT, B_DPJ = B_DC + 96, B_QF = B_DPJ + 96,
              B_DQ = B_QF + 16, B_DQF = B_DQ + 512, B_LK = B_DQF + 16, B_FLAG = B_LK + 32 * 16, B_STAMP = B_FLAG + 4,
              B_VALT = B_STAMP + 2 * 16,             // [96 columns][128 positions] values slice, transposed
              B_WQT = B_VALT + 96 * PT,              // [(k4 * 4 + e) * 256 + l][4]: Wq[4 l + e][16 i + 4 k4 ..]   (BPTT_QOWN: [32 registers][512 threads] cell-0 kernel, half 1)
              B_WQO = B_WQT + 16 * 256 * 4,          // BPTT_QOWN: [4 units e][128 k] Wq rows of this owner's units
              B_FLOATS = B_WQO + (BPTT_QOWN ? 4 * 128 : 0);
    static_assert(B_FLOATS * 4 <= 160 * 1024, "LDS budget");
};
constexpr int NBSTAMP = 16;

struct PersistBwd {
    const float* w1t; const float* w0t; const float* wqt;
    const float* acts0; const float* acts1; const float* craw0; const float* craw1; const float* c0; const float* c1;
    const uint8_t* zc0; const uint8_t* zh0; const uint8_t* zc1; const uint8_t* zh1; float keep;
    const float* align_hist; const float* cum_hist; const float* q_hist;
    const float* keys; const float* values; const int32_t* lengths;
    const float* loc_k; const float* loc_b; const float* score_w; const float* score_b;
    const float* d_pj;
    const float* opk;                            // packed cell-update operands written by the persistent forward (persist_common.h)
    int B, S, T;
    float* dg0; float* dg1; float* dq_hist; float* de_hist; float* d_in0;
    float* xch; unsigned* ctrl; unsigned long long* stamps; int fail_step; int near_xcd;
};

// zoneout-LSTM cell backward for one (row, unit) (the pointwise part of mstts_lstm_point_bwd): dm = gradient of the cell output m
// (without the zoned-state path), dhs / dcs = gradients of the zoned states h' / c'.  Returns the four gate gradients, updates the carried
// state gradients to those of the PREVIOUS step's states (direct zoneout paths only; the product part is added by the caller).
__device__ __forceinline__ pf32x4 cell_bwd(float dm, float& dhs, float& dcs, float si, float tj, float sf, float so, float c, float cp,
                                           float mh, float mc) {
    dm += mh * dhs;
    const float tc = tanhf_(c);
    const float dc = dm * so * (1.f - tc * tc) + mc * dcs;
    pf32x4 dg;
    dg[0] = dc * tj * si * (1.f - si);
    dg[1] = dc * si * (1.f - tj * tj);
    dg[2] = dc * cp * sf * (1.f - sf);
    dg[3] = dm * tc * so * (1.f - so);
    dcs = dcs * (1.f - mc) + dc * sf;
    dhs = dhs * (1.f - mh);
    return dg;
}

// BF16 (BASELINE config 3): the data-gradient products d[g] . W^T of both cells and of the query layer take their operands rounded to bf16
// (the transposed kernels once, when they are loaded into registers; the gate / query gradients when they are fetched) and run on
// v_mfma_f32_16x16x32_bf16 with fp32 accumulators; the cell-update backward, the attention backward and everything exchanged stay fp32, and
// the gate gradients written for the hoisted weight-gradient products are the unrounded ones (those products round for themselves).
template <bool PROF, int TT, bool BF16 = false>
__global__ __launch_bounds__(PTH) void persist_bwd_kernel(PersistBwd d) {
    typedef BL<TT> Y;
    constexpr int B_RED = Y::B_RED, B_G = Y::B_G, B_PC = Y::B_PC, B_DA = Y::B_DA, B_PD = Y::B_PD, B_TR = Y::B_TR, B_GP = Y::B_GP, B_A = Y::B_A, B_CUM = Y::B_CUM,
                  B_DE = Y::B_DE, B_DC = Y::B_DC, B_DPJ = Y::B_DPJ, B_QF = Y::B_QF, B_DQ = Y::B_DQ, B_DQF = Y::B_DQF, B_LK = Y::B_LK, B_FLAG = Y::B_FLAG,
                  B_STAMP = Y::B_STAMP, B_VALT = Y::B_VALT, B_WQT = Y::B_WQT, B_WQO = Y::B_WQO;
    constexpr bool QOWN = BPTT_QOWN != 0 && !BF16;                // (the bf16 instantiations have no f32-input product to replace: the re-cut only costs them its longer gather)
    constexpr bool W0L = QOWN && BPTT_W0LDS != 0 && !BF16;
    constexpr int NH = TT / 128;                  // halves of 128 encoder positions
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int g0 = blockIdx.x, tid0 = threadIdx.x, wave0 = __builtin_amdgcn_readfirstlane(tid0 >> 6);
    int g = g0, gi = g & 7, gj = g >> 3;
    int tid = tid0, lane = tid & 63, wave = wave0;
    const int B = d.B, S = d.S, T = d.T;
    const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc(d.xch, 0, (int)(BXCH_FLOATS * 4), 0x00020000);
    unsigned* sflag = reinterpret_cast<unsigned*>(sm + B_FLAG);

    // ---------------- start rendezvous
    if (d.near_xcd) persist_scrub(xr, BO_DG1, BO_PM0 - BO_DG1, g0, tid);        // the two gate-gradient rings: the ones a slice group may keep in its L2
    __syncthreads();
    if (tid == 0) {
        const int rz = persist_rendezvous(d.ctrl, g0);
        sflag[0] = rz == 0 ? 1u : 0u;
        sflag[1] = (rz == 2 && d.near_xcd) ? 1u : 0u;
        if (rz == 2 && g0 < 8) __hip_atomic_fetch_add(d.ctrl + 3, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);     // (reported: groups publishing through their L2)
    }
    __syncthreads();
    if (sflag[0]) return;
    const bool near = sflag[1] != 0;          // this workgroup's slice group shares one XCD: intra-group pieces may stay in its L2

    // ---------------- once: the transposed kernels of this workgroup / wave -> registers.  w1t[h * 32 + kt * 16 + ks]: half h (0 = the
    // rows on the chain: m0 units, 1 = h1 units), output tile kt, contraction step ks of this wave's eighth; w0t likewise (context / h0)
    constexpr bool BS1 = BSPLIT_C1 != 0 && !BF16 && BPTT_QOWN != 0 && BPTT_W0LDS != 0, BS0 = BSPLIT_C0 != 0 && !BF16;
    // BS1 / BS0: half 0 lives as planes (w1s / w0s), the fp32 array holds half 1 only; W0L: half 1 of cell 0 lives in LDS
    constexpr int NW0 = BF16 ? 1 : ((BS0 ? 0 : 32) + (W0L ? 0 : 32)) > 0 ? ((BS0 ? 0 : 32) + (W0L ? 0 : 32)) : 1;
    constexpr int W0H1 = BS0 ? 0 : 32;                           // where half 1 starts in w0t (unused with W0L)
    float w1t[BF16 ? 1 : (BS1 ? 32 : 64)], w0t[NW0];
    pbf16x8 wb1t[BF16 ? 8 : 1], wb0t[BF16 ? 8 : 1];              // BF16: the same values as packed octets (contraction steps 8 j .. 8 j + 7 of a (half, tile))
    pbf16x8 w1s[3][BS1 ? 4 : 1];                                 // BS1: half 0 of the cell-1 kernel as three planes, [plane][kt * 2 + j]
    pbf16x8 w0s[3][BS0 ? 4 : 1];
    {
        const float* p1 = d.w1t + ((long)(g * 8 + wave) * 64) * 64 + lane;
        const float* p0 = d.w0t + ((long)(g * 8 + wave) * 64) * 64 + lane;
        if constexpr (BF16) {
#pragma unroll
            for (int r = 0; r < 64; ++r) { wb1t[r >> 3][r & 7] = (__bf16)p1[r * 64]; wb0t[r >> 3][r & 7] = (__bf16)p0[r * 64]; }
        } else {
            if constexpr (BS1) {
#pragma unroll
                for (int o = 0; o < 4; ++o) {          // (kt, j) = (o >> 1, o & 1): registers kt * 16 + 8 j + e of half 0
                    float x[8];
#pragma unroll
                    for (int e = 0; e < 8; ++e) x[e] = p1[((o >> 1) * 16 + 8 * (o & 1) + e) * 64];
                    bsp_split8(x, w1s[0][o], w1s[1][o], w1s[2][o]);
                }
#pragma unroll
                for (int r = 0; r < 32; ++r) w1t[r] = p1[(32 + r) * 64];
            } else {
#pragma unroll
                for (int r = 0; r < 64; ++r) w1t[r] = p1[r * 64];
            }
            if constexpr (BS0) {
#pragma unroll
                for (int o = 0; o < 4; ++o) {
                    float x[8];
#pragma unroll
                    for (int e = 0; e < 8; ++e) x[e] = p0[((o >> 1) * 16 + 8 * (o & 1) + e) * 64];
                    bsp_split8(x, w0s[0][o], w0s[1][o], w0s[2][o]);
                }
            } else {
#pragma unroll
                for (int r = 0; r < 32; ++r) w0t[r] = p0[r * 64];
            }
            if constexpr (W0L) {
#pragma unroll
                for (int r = 0; r < 32; ++r) sm[B_WQT + r * PTH + tid] = p0[(32 + r) * 64];
            } else {
#pragma unroll
                for (int r = 0; r < 32; ++r) w0t[W0H1 + r] = p0[(32 + r) * 64];
            }
        }
    }
    int ab = gj;
    const bool arow = ab < B;
    const int alen = arow ? (d.lengths ? d.lengths[ab] : T) : 0;
    int ak = tid & 15, atg = tid >> 4;
    float kreg[4 * NH];
    float asb = 0.f, awk = 0.f, asb2 = 0.f;
    // (with the per-iteration copies of the indices when the macro stands at the bottom of the loop body: the loads are then this step's, not hoisted)
#define LOAD_KEYS() do { const int gi__ = g & 7, ab__ = g >> 3, ak__ = tid & 15, atg__ = tid >> 4; const bool ar__ = ab__ < B;              \
        _Pragma("unroll") for (int m = 0; m < 4 * NH; ++m) {                                                                              \
            const int t = 128 * (m >> 2) + 4 * atg__ + (m & 3);                                                                            \
            kreg[m] = d.keys[((long)(ar__ ? ab__ : 0) * T + (t < T ? t : 0)) * PA + 16 * gi__ + ak__];   /* (raw: masked where it is used - a use here would wait for the load) */ \
        }                                                                                                                                  \
        asb = d.score_b[16 * gi__ + ak__]; asb2 = d.loc_b[16 * gi__ + ak__]; awk = d.score_w[16 * gi__ + ak__]; } while (0)
    {
        LOAD_KEYS();
#if !BPTT_KEYS_PER_STEP
        asb += asb2;
#endif
        for (int x = tid; x < 32 * 16; x += PTH) sm[B_LK + x] = (x < PKS * 16) ? d.loc_k[(x >> 4) * PA + 16 * gi + (x & 15)] : 0.f;
        for (int x = tid; x < 96 * PT; x += PTH) {
            const int c = x >> 7, t = x & 127;
            sm[B_VALT + x] = (arow && t < alen && t < T) ? d.values[((long)ab * T + t) * PM + 96 * gi + c] : 0.f;
        }
        for (int x = tid; x < TT + 48; x += PTH) sm[B_CUM + x] = 0.f;
        for (int x = tid; x < TT; x += PTH) { sm[B_GP + x] = 0.f; sm[B_A + x] = 0.f; sm[B_DE + x] = 0.f; }
        if constexpr (QOWN) {
            // Wq rows of THIS owner's four units, all 128 k: packed float4 ((slice * 16 + k4 * 4 + e) * 256 + g) holds Wq[4 g + e][16 slice + 4 k4 ..]
            if (tid < 128) {
                const int sl = tid >> 4, k4 = (tid >> 2) & 3, e = tid & 3;
                pf32x4 v = reinterpret_cast<const pf32x4*>(d.wqt)[((long)(sl * 16 + k4 * 4 + e)) * 256 + g];
                if constexpr (BF16) { v[0] = bf16_round(v[0]); v[1] = bf16_round(v[1]); v[2] = bf16_round(v[2]); v[3] = bf16_round(v[3]); }
                *reinterpret_cast<pf32x4*>(sm + B_WQO + e * 128 + 16 * sl + 4 * k4) = v;
            }
        } else {
            const pf32x4* wqs = reinterpret_cast<const pf32x4*>(d.wqt) + (long)gi * 16 * 256;
            for (int x = tid; x < 16 * 256; x += PTH) {
                pf32x4 v = wqs[x];
                if constexpr (BF16) { v[0] = bf16_round(v[0]); v[1] = bf16_round(v[1]); v[2] = bf16_round(v[2]); v[3] = bf16_round(v[3]); }
                reinterpret_cast<pf32x4*>(sm + B_WQT)[x] = v;
            }
        }
    }
    int et = wave & 1, er = 16 * et + (lane & 15), ee = lane >> 4, eu = 4 * g + ee;
    bool ew = wave < 2, elive = ew && er < B;
    float dc0s = 0.f, dh0s = 0.f, dc1s = 0.f, dh1s = 0.f;       // carried gradients of the zoned states c' / h' (direct paths)
    unsigned long long* sstamp = reinterpret_cast<unsigned long long*>(sm + B_STAMP);
    unsigned tprev = 0;
    if (PROF && tid < NBSTAMP) sstamp[tid] = 0;
#define PSTAMP(idx) do { if (PROF && tid == 0) { const unsigned n__ = (unsigned)wall_clock64(); sstamp[idx] += (unsigned)(n__ - tprev); tprev = n__; } } while (0)
#define PABORT_CHECK() do { __syncthreads(); if (sflag[0]) return; } while (0)
#define PFAIL() do { sflag[0] = 1; unsigned z__ = 0u; __hip_atomic_compare_exchange_strong(d.ctrl + 1, &z__, 2u, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); } while (0)
    // the forward rows of a step that its attention backward needs (alignment, cumulative alignment BEFORE the step, this slice's
    // query units and the projection's part of d_ctx): requested one step ahead (end of the attention phase of step s + 1), parked
    // in LDS once the cell-1 wait of that step has drained the load queue anyway - they cost HBM latency, not bandwidth
    float rav = 0.f, rcv = 0.f, rqv = 0.f, rdv = 0.f;
    const int abc = arow ? ab : 0;
#define LOAD_ROWS(ST) do { const long r__ = (long)(ST) * B + abc; const int t__ = tid & (TT - 1);                                   \
        rav = (d.align_hist + r__ * T)[t__ < T ? t__ : 0]; rcv = (d.cum_hist + r__ * T)[t__ < T ? t__ : 0];                     \
        rqv = (d.q_hist + r__ * PA + 16 * (g & 7))[tid & 15]; rdv = (d.d_pj + r__ * (PH + PM) + PH + 96 * (g & 7))[tid < 96 ? tid : 0]; } while (0)
#define STORE_ROWS() do { const int t__ = tid & (TT - 1);                                                                           \
        sm[B_A + t__] = t__ < T ? rav : 0.f; sm[B_CUM + 15 + t__] = t__ < T ? rcv : 0.f;                                        \
        sm[B_QF + (tid & 15)] = rqv; sm[B_DPJ + (tid < 96 ? tid : 0)] = rdv; } while (0)
    // operands of the cell-1 update backward (saved activations, cell states, keep-masks, the projection's d_m1): HBM-cold, requested ONE STEP
    // AHEAD (here for the first step, at the bottom of the loop body for the next) by every wave outside any condition - see persist.hip
    float a1v[4], cr1, cp1, dpm1;
    uint8_t zc1v, zh1v;
#define LOAD_OPERANDS1(ST) do { const long b__ = (long)(ST) * B; const unsigned r__ = (16 * (wave & 1) + (tid & 15)) < (unsigned)B ? 16 * (wave & 1) + (tid & 15) : 0u; \
        const unsigned u__ = 4 * g + ((tid & 63) >> 4);                                                                                   \
        /* (waves 2..7 only repeat the update: they read what wave 0 / 1 read - tid & 127 - from the packed block) */                      \
        const pf32x4* ob__ = reinterpret_cast<const pf32x4*>(d.opk) + opk_index((ST), g, 1, 0, tid & 127);                                \
        const pf32x4 A__ = ob__[0], B__ = ob__[128];                                                                                        \
        a1v[0] = A__[0]; a1v[1] = A__[1]; a1v[2] = A__[2]; a1v[3] = A__[3]; cr1 = B__[0]; cp1 = B__[1];                                      \
        zc1v = (uint8_t)(__float_as_uint(B__[2]) & 1u); zh1v = (uint8_t)((__float_as_uint(B__[2]) >> 1) & 1u);                               \
        dpm1 = (d.d_pj + b__ * (PH + PM))[wave < 2 ? r__ * (PH + PM) + u__ : 0u]; } while (0)
    LOAD_OPERANDS1(S - 1);
    LOAD_ROWS(S - 1);
    __syncthreads();          // the zero fills of B_CUM / B_A above are other threads' stores to the words STORE_ROWS writes: order them (round 6: with the
                              // query-kernel load gone from half of the waves the fill of a slow wave could land behind a fast wave's row - one workgroup's
                              // first step then ran on a zeroed cumulative-alignment window, 4e-4 in dq_hist of the first launch of a process)
    STORE_ROWS();
    __syncthreads();
    if (PROF && tid == 0) tprev = (unsigned)wall_clock64();

    for (int s = S - 1; s >= 0; --s) {
        const bool first = s == S - 1;
        const unsigned kk = (unsigned)(S - 1 - s);                                           // steps counted from the start of the launch
        const unsigned slot = kk & 3u, nslot = (kk + 3u) & 3u;                                 // nslot: the slot of step s + 1 (the one before in time)
        const unsigned gen = (kk >> 2) & 1u, ngen = ((kk - 1u) >> 2) & 1u;                     // ... and their generations
        tid = tid0; g = g0; wave = wave0;
        asm volatile("" : "+v"(tid));
        asm volatile("" : "+s"(g), "+s"(wave));
        lane = tid & 63; gi = g & 7; gj = g >> 3; ab = gj; ak = tid & 15; atg = tid >> 4;
        et = wave & 1; er = 16 * et + (lane & 15); ee = lane >> 4; eu = 4 * g + ee;
        ew = wave < 2; elive = ew && er < B;
        if (s == d.fail_step && g == 0 && tid == 0) {
            __hip_atomic_store(d.ctrl + 1, 3u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            sflag[0] = 1;
        }
        const long sB = (long)s * B, sB1 = sB + B;
        const bool rowok = er < B;
        const unsigned erc = rowok ? (unsigned)er : 0u;
        const unsigned oH = erc * PH + eu, o4H = erc * 4 * PH + eu;
        // requests of the two cell-update waits, issued as soon as this workgroup's own contribution has left
        unsigned uoff[2], qoff[QOWN ? 5 : 1];
        pf32x4 uv[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
        pf32x4 qv5[QOWN ? 5 : 1];
#define ISSUE_UPDATE1() do { if (tid < 256) { const int row = tid >> 3, sl = tid & 7;                                                  \
            if constexpr (QOWN) {   /* the 16 dq values of (row, slice sl) + this piece of the recurrent-state product */                \
                _Pragma("unroll") for (int p = 0; p < 4; ++p) qoff[p] = (unsigned)((BO_DM1 + slot * BDM1 + ((long)row * 8 + sl) * 16 + 4 * p) * 4); \
                qoff[4] = first ? qoff[0] : (unsigned)((BO_PH1 + nslot * BPART + (((long)g * 8 + sl) * 32 + row) * 4) * 4);           \
                issue<5>(xr, qoff, qv5);                                                                                              \
            } else {                                                                                                                  \
            uoff[0] = (unsigned)((BO_DM1 + slot * BDM1 + (((long)g * 8 + sl) * 32 + row) * 4) * 4);                                   \
            uoff[1] = first ? uoff[0] : (unsigned)((BO_PH1 + nslot * BPART + (((long)g * 8 + sl) * 32 + row) * 4) * 4);               \
            issue<2>(xr, uoff, uv); } } } while (0)
#define ISSUE_UPDATE0() do { if (tid < 256) { const int row = tid >> 3, src = tid & 7;                                                 \
            uoff[0] = (unsigned)((BO_PM0 + slot * BPART + (((long)g * 8 + src) * 32 + row) * 4) * 4);                                 \
            uoff[1] = first ? uoff[0] : (unsigned)((BO_PH0 + nslot * BPART + (((long)g * 8 + src) * 32 + row) * 4) * 4);              \
            issue<2>(xr, uoff, uv); } } while (0)
        // ================= attention backward of row ab, slice gi
        float fac[4 * NH];
#pragma unroll
        for (int m = 0; m < 4 * NH; ++m) fac[m] = 0.f;
        if (arow) {
            // (this step's forward rows are in LDS already: LOAD_ROWS / STORE_ROWS)
            // zero padding of the energy-gradient window (positions -15 .. -1 and TT .. TT + 16): the products of the step before wrote here
            sm[B_G + (tid < 240 ? tid : (15 + TT) * 16 + (tid - 240))] = 0.f;
            pf32x4 pc[1] = {{0.f, 0.f, 0.f, 0.f}};
            unsigned pcoff[1];
            pcoff[0] = (unsigned)((BO_PCTX + nslot * BPCTX + ((long)ab * 192 + 24 * gi) * 32) * 4 + 16 * (tid < 192 ? tid : 0));
            if (!first && tid < 192) issue<1>(xr, pcoff, pc);       // requested now, it travels under the tanh terms
            __syncthreads();
            PSTAMP(15);
            // tanh terms of this slice (independent of everything that arrives): fac = w_k (1 - tanh^2(keys + q + location filter))
            {
                const float qk = sm[B_QF + ak] + (BPTT_KEYS_PER_STEP ? asb + asb2 : asb);
                // location filter as a Toeplitz product on the matrix core (see persist.hip): A = the cumulative-alignment window, B = the filter slice
#pragma unroll
                for (int hh = 0; hh < NH; ++hh) {
                    pf32x4 loc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int ks = 0; ks < 8; ++ks)
                        loc = PMFMA(sm[B_CUM + 128 * hh + 16 * wave + (lane & 15) + 4 * ks + (lane >> 4)], sm[B_LK + (4 * ks + (lane >> 4)) * 16 + (lane & 15)], loc);
#pragma unroll
                    for (int m = 0; m < 4; ++m) {
                        const float kv = (128 * hh + 4 * atg + m < T) ? kreg[4 * hh + m] : 0.f;      // (arow holds here; positions past T read as zero keys)
                        const float u = tanhf_(kv + qk + loc[m]); fac[4 * hh + m] = awk * (1.f - u * u);
                    }
                }
            }
            PSTAMP(0);
            // d_ctx of this slice's 96 columns: projection part + the 8 partial tiles of the next step's cell-0 product
            if (!first) {
                if (tid < 192) {
                    { const unsigned g1[1] = {ngen}; if (!complete<1>(xr, pcoff, pc, d.ctrl, g1)) PFAIL(); }
                    *reinterpret_cast<pf32x4*>(sm + B_PC + 4 * tid) = pc[0];
                }
            }
            PABORT_CHECK();
            PSTAMP(1);
            if (tid < 24) {
                pf32x4 acc = {0.f, 0.f, 0.f, 0.f};
                if (!first) {
#pragma unroll
                    for (int src = 0; src < 8; ++src) acc += *reinterpret_cast<const pf32x4*>(sm + B_PC + (tid * 8 + src) * 4);
                    reinterpret_cast<pf32x4*>(d.d_in0 + (sB1 + ab) * (PM + PH) + 96 * gi)[tid] = acc;       // gradient of ctx_s from cell 0 of step s+1 (d_values)
                }
                *reinterpret_cast<pf32x4*>(sm + B_DC + 4 * tid) = acc + *reinterpret_cast<const pf32x4*>(sm + B_DPJ + 4 * tid);
            }
            __syncthreads();
            {   // values_i . d_ctx_i: thread (position t, column quarter cq)
                const int t = tid & 127, cq = tid >> 7;
                pf32x4 vg[6];
                if (NH > 1) {       // position 128 + t: this thread's 24 value columns from memory (zero past the row's length there; L2 hits after the first step)
                    const int tt = 128 + t < T ? 128 + t : 0;
                    const pf32x4* vp4 = reinterpret_cast<const pf32x4*>(d.values + ((long)ab * T + tt) * PM + 96 * gi + 24 * cq);
#pragma unroll
                    for (int c4 = 0; c4 < 6; ++c4) vg[c4] = vp4[c4];
                }
                float acc = 0.f, acc2 = 0.f;
#pragma unroll 2
                for (int c4 = 0; c4 < 6; ++c4) {
                    const pf32x4 dcv = *reinterpret_cast<const pf32x4*>(sm + B_DC + 24 * cq + 4 * c4);
#pragma unroll
                    for (int e = 0; e < 4; ++e) acc += sm[B_VALT + (24 * cq + 4 * c4 + e) * PT + t] * dcv[e];
                }
                sm[B_PD + cq * TT + t] = acc;
                if (NH > 1) {
#pragma unroll
                    for (int c4 = 0; c4 < 6; ++c4) {
                        const pf32x4 dcv = *reinterpret_cast<const pf32x4*>(sm + B_DC + 24 * cq + 4 * c4);
#pragma unroll
                        for (int e = 0; e < 4; ++e) acc2 += vg[c4][e] * dcv[e];
                    }
                    sm[B_PD + cq * TT + 128 + t] = 128 + t < T ? acc2 : 0.f;
                }
            }
            __syncthreads();
            if (tid < 32 * NH) {    // + this slice's part of G; the row's d_alignment is the sum of the eight published vectors
                pf32x4 pd = (*reinterpret_cast<const pf32x4*>(sm + B_PD + 4 * tid) + *reinterpret_cast<const pf32x4*>(sm + B_PD + TT + 4 * tid)) +
                            (*reinterpret_cast<const pf32x4*>(sm + B_PD + 2 * TT + 4 * tid) + *reinterpret_cast<const pf32x4*>(sm + B_PD + 3 * TT + 4 * tid));
                pd += *reinterpret_cast<const pf32x4*>(sm + B_GP + 4 * tid);
                const long o = ((long)ab * 8 + gi) * TT + 4 * tid;
                xpublish(xr, (unsigned)((BO_DA + slot * BDA + o) * 4), pd, gen);
            }
            PSTAMP(2);
        }
        if (arow) {
            if (tid < 256 * NH) {
                unsigned off[1]; pf32x4 v[1];
                off[0] = (unsigned)((BO_DA + slot * BDA + (long)ab * 8 * TT) * 4 + 16 * tid);
                const unsigned g1[1] = {gen};
                if (!gather<1>(xr, off, v, d.ctrl, g1)) PFAIL();
                *reinterpret_cast<pf32x4*>(sm + B_DA + 4 * tid) = v[0];
            }
            PABORT_CHECK();
            PSTAMP(3);
            if (wave == 0) {        // softmax backward: d_e = a (d_a - dot(a, d_a)); lane holds positions lane + 64 i
                float da[2 * NH], av[2 * NH];
                float dp = 0.f;
#pragma unroll
                for (int i = 0; i < 2 * NH; ++i) {
                    float x = 0.f;
#pragma unroll
                    for (int k = 0; k < 8; ++k) x += sm[B_DA + k * TT + 64 * i + lane];
                    da[i] = x; av[i] = sm[B_A + 64 * i + lane];
                }
                if (NH == 1) dp = av[0] * da[0] + av[1] * da[1];
                else {
#pragma unroll
                    for (int i = 0; i < 2 * NH; ++i) dp += av[i] * da[i];
                }
                const float dot = wave_sum(dp);
                float* de = d.de_hist + (sB + ab) * T;
#pragma unroll
                for (int i = 0; i < 2 * NH; ++i) {
                    const float e = av[i] * (da[i] - dot);
                    sm[B_DE + 64 * i + lane] = e;
                    if (gi == 0 && lane + 64 * i < T) de[lane + 64 * i] = e;
                }
            }
            __syncthreads();
            {   // energy gradients g[t][k] = d_e[t] fac[t][k]; dq[k] = sum_t g
                float dqp = 0.f;
#pragma unroll
                for (int m = 0; m < 4 * NH; ++m) {
                    const int t = 128 * (m >> 2) + 4 * atg + (m & 3);
                    const float gv = sm[B_DE + t] * fac[m];
                    sm[B_G + (15 + t) * 16 + ak] = gv;
                    dqp += gv;
                }
                sm[B_DQ + atg * 16 + ak] = dqp;
            }
            __syncthreads();
            if (tid < 16) {
                float qv = 0.f;
#pragma unroll
                for (int u = 0; u < 32; ++u) qv += sm[B_DQ + u * 16 + tid];
                sm[B_DQF + tid] = BF16 ? bf16_round(qv) : qv;       // (BF16: the data-gradient product's operand; the history keeps the unrounded value)
                (d.dq_hist + (sB + ab) * PA + 16 * gi)[tid] = qv;
            }
            __syncthreads();
            if constexpr (QOWN) {
                // the 16 dq values of (row ab, slice gi), once, for all 256 owners (tools/broadcast_probe.hip: a 256-way read of 16 KB costs
                // 0.17 us more per exchange than 256 personal pieces)
                if (tid < 4) xpublish(xr, (unsigned)((BO_DM1 + slot * BDM1 + ((long)ab * 8 + gi) * 16 + 4 * tid) * 4), *reinterpret_cast<const pf32x4*>(sm + B_DQF + 4 * tid), gen);
            } else
            if (tid < 256) {        // query layer, data gradient of this slice's 16 units: d_m1[4 l + e] += sum_k dq[k] Wq[4 l + e][16 i + k]
                pf32x4 out = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 1
                for (int k4 = 0; k4 < 4; ++k4) {
                    const pf32x4 dq4 = *reinterpret_cast<const pf32x4*>(sm + B_DQF + 4 * k4);
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const pf32x4 wv = *reinterpret_cast<const pf32x4*>(sm + B_WQT + ((k4 * 4 + e) * 256 + tid) * 4);
                        out[e] += wv[0] * dq4[0] + wv[1] * dq4[1] + wv[2] * dq4[2] + wv[3] * dq4[3];
                    }
                }
                const long o = (((long)tid * 8 + gi) * 32 + ab) * 4;           // owner-major: the owner of units 4 tid .. polls ONE contiguous 4 KB block (row-major, its 256 pieces were 4 KB apart: 0.24 us of address processing per poll instruction)
                xpublish(xr, (unsigned)((BO_DM1 + slot * BDM1 + o) * 4), out, gen);
            }
            PSTAMP(4);
#pragma unroll 1
            for (int hh = 0; hh < NH; ++hh) {   // in the shadow of that hand-off: this slice's part of G for the step before: G[t] += sum_j sum_k g[t + 15 - j][k] loc_k[j][k]
                const int p0 = 128 * hh + 4 * atg;
                float acc[4] = {0.f, 0.f, 0.f, 0.f};
                float w0 = sm[B_G + p0 * 16 + ak], w1 = sm[B_G + (p0 + 1) * 16 + ak], w2 = sm[B_G + (p0 + 2) * 16 + ak],
                      w3 = sm[B_G + (p0 + 3) * 16 + ak];
#pragma unroll 4
                for (int x = 0; x < PKS; ++x) {         // x = 30 - j: padded window position p0 + m + x
                    const float lk = sm[B_LK + (PKS - 1 - x) * 16 + ak];
                    acc[0] += w0 * lk; acc[1] += w1 * lk; acc[2] += w2 * lk; acc[3] += w3 * lk;
                    w0 = w1; w1 = w2; w2 = w3; w3 = sm[B_G + (p0 + x + 4) * 16 + ak];
                }
#pragma unroll
                for (int m = 0; m < 4; ++m) {
                    float v = acc[m];
                    v += dpp_mov<0xB1, 0xf>(0.f, v);
                    v += dpp_mov<0x4E, 0xf>(0.f, v);
                    v += dpp_mov<0x141, 0xf>(0.f, v);
                    v += dpp_mov<0x140, 0xf>(0.f, v);
                    acc[m] = v;
                }
                if (ak == 0) {
#pragma unroll
                    for (int m = 0; m < 4; ++m) sm[B_GP + p0 + m] += acc[m];
                }
            }
        } else {
            PSTAMP(0); PSTAMP(1); PSTAMP(2); PSTAMP(3);
            if constexpr (QOWN) {
                if (tid < 4) xpublish(xr, (unsigned)((BO_DM1 + slot * BDM1 + ((long)ab * 8 + gi) * 16 + 4 * tid) * 4), (pf32x4){0.f, 0.f, 0.f, 0.f}, gen);
            } else
            if (tid < 256) {        // rows past the batch: zeros, so that the cell owners' waits complete
                const long o = (((long)tid * 8 + gi) * 32 + ab) * 4;
                xpublish(xr, (unsigned)((BO_DM1 + slot * BDM1 + o) * 4), (pf32x4){0.f, 0.f, 0.f, 0.f}, gen);
            }
            PSTAMP(4);
        }
        PSTAMP(5);
        LOAD_ROWS(s > 0 ? s - 1 : 0);
        // ================= cell 1, update backward (units 4 g .., all rows)
        __syncthreads();                                             // the attention phases are done with the scratch
        ISSUE_UPDATE1();
        if (tid < 256) {            // piece (row, slice / source): 8 consecutive lanes hold the 8 partial vectors of one (row, 4 units)
            const int row = tid >> 3, sl = tid & 7;
            if constexpr (QOWN) {
                { const unsigned g5[5] = {gen, gen, gen, gen, first ? gen : ngen}; if (!complete<5>(xr, qoff, qv5, d.ctrl, g5)) PFAIL(); }
                uv[1] = qv5[4];
                // this slice's share of the query layer's data gradient for the owner's four units: d_m1[4 g + e] += sum_k dq[row][16 sl + k] Wq[4 g + e][16 sl + k]
                uv[0] = (pf32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int p = 0; p < 4; ++p) {          // (one piece at a time, fenced: left alone the scheduler hoists all sixteen kernel reads - 64 registers)
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const pf32x4 wv = *reinterpret_cast<const pf32x4*>(sm + B_WQO + e * 128 + 16 * sl + 4 * p);
                        uv[0][e] += wv[0] * qv5[p][0] + wv[1] * qv5[p][1] + wv[2] * qv5[p][2] + wv[3] * qv5[p][3];
                    }
                    asm volatile("" ::: "memory");
                }
            } else {
                const unsigned g2[2] = {gen, first ? gen : ngen}; if (!complete<2>(xr, uoff, uv, d.ctrl, g2)) PFAIL();
            }
            if (first) uv[1] = (pf32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float x = uv[0][e], y = uv[1][e];
                x += dpp_mov<0xB1, 0xf>(0.f, x); x += dpp_mov<0x4E, 0xf>(0.f, x); x += dpp_mov<0x141, 0xf>(0.f, x);
                y += dpp_mov<0xB1, 0xf>(0.f, y); y += dpp_mov<0x4E, 0xf>(0.f, y); y += dpp_mov<0x141, 0xf>(0.f, y);
                uv[0][e] = x; uv[1][e] = y;
            }
            if (sl == 0) {
                *reinterpret_cast<pf32x4*>(sm + B_TR + row * 4) = uv[0];
                *reinterpret_cast<pf32x4*>(sm + B_TR + 128 + row * 4) = uv[1];
            }
        }
        STORE_ROWS();                                                // (the wait above drained the queue: the rows of step s - 1 have arrived)
        PABORT_CHECK();
        PSTAMP(6);
        {
            const float dm = sm[B_TR + er * 4 + ee] + dpm1;
            float dhs = dh1s + sm[B_TR + 128 + er * 4 + ee];
            const float mh = zh1v ? d.keep : 0.f, mc = zc1v ? d.keep : 0.f;
            const pf32x4 dgv = cell_bwd(dm, dhs, dc1s, a1v[0], a1v[1], a1v[2], a1v[3], cr1, cp1, mh, mc);
            dh1s = dhs;
            if (ew) {
#pragma unroll
                for (int q = 0; q < 4; ++q) sm[B_TR + 384 + (q * 32 + er) * 4 + ee] = dgv[q];
            }
            __syncthreads();
            if (tid < 128) {        // (gate, row): the 4 units' gradients of one gate, 16 bytes into the B-operand order of the product waves
                const int gate = tid >> 5, row = tid & 31;
                const pf32x4 val = *reinterpret_cast<const pf32x4*>(sm + B_TR + 384 + tid * 4);
                const long o = ((((long)(gi * 8 + (gj >> 2)) * 2 + (row >> 4)) * 4 + (gj & 3)) * 64 + gate * 16 + (row & 15)) * 4;
                xpublish_near(xr, (unsigned)((BO_DG1 + slot * BDG + o) * 4), val, gen, near);
            }
            if (elive) { float* o = d.dg1 + sB * 4 * PH; o[o4H] = dgv[0]; o[o4H + PH] = dgv[1]; o[o4H + 2 * PH] = dgv[2]; o[o4H + 3 * PH] = dgv[3]; }
        }
        PSTAMP(7);
        // ================= products: a wave's eighth of the gate gradients -> B-operand registers, 64 MFMAs on the chain, the 8 partial
        // tiles meet in LDS, leave as one tile; then the 64 MFMAs of the recurrent-state rows in the shadow of that hand-off
#define PROD_GATHER(OFF_DG)                                                                                                           \
        pf32x4 bq[8];                                                                                                                \
        {                                                                                                                            \
            unsigned off[8];                                                                                                         \
            _Pragma("unroll") for (int x = 0; x < 8; ++x)                                                                            \
                off[x] = (unsigned)(((OFF_DG) + slot * BDG + ((((long)(gi * 8 + wave) * 2 + (x >> 2)) * 4 + (x & 3)) * 64 + lane) * 4) * 4); \
            const unsigned g8[8] = {gen, gen, gen, gen, gen, gen, gen, gen};                                                         \
            if (!gather<8>(xr, off, bq, d.ctrl, g8)) PFAIL();                                                                            \
        }                                                                                                                            \
        pbf16x8 bqh[BF16 ? 4 : 1];            /* BF16: [row tile t][octet j] = contraction steps 8 j .. 8 j + 7 of tile t, rounded */  \
        if constexpr (BF16) {                                                                                                        \
            _Pragma("unroll") for (int x = 0; x < 4; ++x)                                                                            \
                _Pragma("unroll") for (int e = 0; e < 8; ++e) bqh[x][e] = (__bf16)bq[(x >> 1) * 4 + 2 * (x & 1) + (e >> 2)][e & 3];  \
        }
// WT / WBASE: the fp32 kernel registers of this half; WB: the bf16 instantiation's; WS + SPL: this half as three planes, evaluated as the
// six-product split; ALDS: the fp32 kernel values of this half come from LDS ([register][thread], B_WQT region) instead of registers
#define PROD_HALF(half, WT, WBASE, WB, WS, OFF_OUT, IS_CTX, SPL, ALDS)                                                                \
        {                                                                                                                            \
            pf32x4 acc[2][2];                                                                                                        \
            _Pragma("unroll") for (int kt = 0; kt < 2; ++kt)                                                                         \
                _Pragma("unroll") for (int t = 0; t < 2; ++t) acc[kt][t] = (pf32x4){0.f, 0.f, 0.f, 0.f};                             \
            if constexpr (BF16) {                                                                                                    \
                _Pragma("unroll") for (int j = 0; j < 2; ++j)                                                                        \
                    _Pragma("unroll") for (int kt = 0; kt < 2; ++kt) {                                                               \
                        acc[kt][0] = PMFMA_BF16(WB[(half) * 4 + kt * 2 + j], bqh[j], acc[kt][0]);                                     \
                        acc[kt][1] = PMFMA_BF16(WB[(half) * 4 + kt * 2 + j], bqh[2 + j], acc[kt][1]);                                 \
                    }                                                                                                                \
            } else if constexpr (SPL) {                                                                                              \
                /* one row tile at a time: its six gate-gradient plane octets (24 registers) live only over its 24 matrix instructions, and its \
                   two accumulator tiles go to the reduction scratch at once (free since the attention phases' last barrier) - by (row tile,     \
                   octet), 12 registers at a time, the register allocator does worse: 44 spilled against 14 */                                   \
                _Pragma("unroll") for (int t = 0; t < 2; ++t) {                                                                      \
                    pbf16x8 bp__[3][2];                                                                                              \
                    _Pragma("unroll") for (int j = 0; j < 2; ++j) {                                                                  \
                        float f__[8];                                                                                                \
                        _Pragma("unroll") for (int e = 0; e < 8; ++e) f__[e] = bq[t * 4 + 2 * j + (e >> 2)][e & 3];                  \
                        bsp_split8(f__, bp__[0][j], bp__[1][j], bp__[2][j]);                                                         \
                    }                                                                                                                \
                    _Pragma("unroll") for (int kt = 0; kt < 2; ++kt) {                                                               \
                        pf32x4 a__ = {0.f, 0.f, 0.f, 0.f};                                                                           \
                        _Pragma("unroll") for (int j = 0; j < 2; ++j) {                                                              \
                            a__ = PMFMA_BF16(WS[2][kt * 2 + j], bp__[0][j], a__);                                                    \
                            a__ = PMFMA_BF16(WS[0][kt * 2 + j], bp__[2][j], a__);                                                    \
                            a__ = PMFMA_BF16(WS[1][kt * 2 + j], bp__[1][j], a__);                                                    \
                            a__ = PMFMA_BF16(WS[1][kt * 2 + j], bp__[0][j], a__);                                                    \
                            a__ = PMFMA_BF16(WS[0][kt * 2 + j], bp__[1][j], a__);                                                    \
                            a__ = PMFMA_BF16(WS[0][kt * 2 + j], bp__[0][j], a__);                                                    \
                        }                                                                                                            \
                        *reinterpret_cast<pf32x4*>(sm + B_RED + ((wave * 4 + kt * 2 + t) * 64 + lane) * 4) = a__;                    \
                    }                                                                                                                \
                    __builtin_amdgcn_sched_barrier(0);                                                                               \
                }                                                                                                                    \
            } else {                                                                                                                 \
                _Pragma("unroll") for (int ks = 0; ks < 16; ++ks)                                                                    \
                    _Pragma("unroll") for (int kt = 0; kt < 2; ++kt) {                                                               \
                        float a__;                                                                                                   \
                        if constexpr (ALDS) a__ = sm[B_WQT + (kt * 16 + ks) * PTH + tid]; else a__ = WT[(WBASE) + kt * 16 + ks];     \
                        acc[kt][0] = PMFMA(a__, bq[ks >> 2][ks & 3], acc[kt][0]);                                                    \
                        acc[kt][1] = PMFMA(a__, bq[4 + (ks >> 2)][ks & 3], acc[kt][1]);                                              \
                    }                                                                                                                \
            }                                                                                                                        \
            if ((half) == 1) __syncthreads();                        /* the first tile's readers are done */                         \
            if constexpr (BF16 || !(SPL)) {                          /* (the split form has stored its tiles already; it is used for half 0 only) */ \
                _Pragma("unroll") for (int kt = 0; kt < 2; ++kt)                                                                     \
                    _Pragma("unroll") for (int t = 0; t < 2; ++t)                                                                    \
                        *reinterpret_cast<pf32x4*>(sm + B_RED + ((wave * 4 + kt * 2 + t) * 64 + lane) * 4) = acc[kt][t];             \
            }                                                                                                                        \
            PABORT_CHECK();                                                                                                          \
            if (tid < 256) {                                                                                                         \
                const int tile = tid >> 6, l = tid & 63, kt = tile >> 1, t = tile & 1;                                               \
                pf32x4 r = {0.f, 0.f, 0.f, 0.f};                                                                                     \
                _Pragma("unroll") for (int w = 0; w < 8; ++w) r += *reinterpret_cast<const pf32x4*>(sm + B_RED + ((w * 4 + tile) * 64 + l) * 4); \
                const int blk = 4 * kt + (l >> 4), row = 16 * t + (l & 15);                                                          \
                if (IS_CTX) {                                                                                                        \
                    if (blk < 6) {                                                                                                   \
                        const long o = (((long)row * 192 + 6 * gj + blk) * 8 + gi) * 4;                                              \
                        xpublish(xr, (unsigned)((BO_PCTX + slot * BPCTX + o) * 4), r, gen);                                                 \
                    }                                                                                                                \
                } else {                                                                                                             \
                    const long o = (((long)(8 * gj + blk) * 8 + gi) * 32 + row) * 4;                                                 \
                    xpublish(xr, (unsigned)(((OFF_OUT) + slot * BPART + o) * 4), r, gen);                                                   \
                }                                                                                                                    \
            }                                                                                                                        \
        }
        pf32x4 A0_, B0_;                      // cell-0 operands as loaded: split into activations / states / mask bits only where the update uses them
        {
            PROD_GATHER(BO_DG1)
            PROD_HALF(0, w1t, 0, wb1t, w1s, BO_PM0, false, BS1, false)
            PSTAMP(8);
#define LOAD_OPERANDS0() do { /* operands of the cell-0 update backward (HBM-cold): they arrive under the product / the hand-off that follows */ \
                const pf32x4* ob = reinterpret_cast<const pf32x4*>(d.opk) + opk_index(s, g, 0, 0, tid & 127);       /* (the per-iteration copies: the address is formed again every step instead of living in two registers across the loop) */ \
                A0_ = ob[0]; B0_ = ob[128];                                                                                           \
                /* (no arithmetic on them here: the first use makes the compiler wait for the loads, and its vmcnt(0) - the publication just \
                    above sits in a conditional block - also waits for that write-through store to be acknowledged: 0.5 us in this stage) */ \
            } while (0)
#if !BPTT_LATE_OP0
            LOAD_OPERANDS0();
#endif
            PROD_HALF(1, w1t, (BS1 ? 0 : 32), wb1t, w1s, BO_PH1, false, false, false)
        }
        PSTAMP(9);
        // ================= cell 0, update backward
        __syncthreads();
#if BPTT_LATE_OP0
        LOAD_OPERANDS0();
#endif
        ISSUE_UPDATE0();
        if (tid < 256) {
            const int row = tid >> 3, src = tid & 7;
            { const unsigned g2[2] = {gen, first ? gen : ngen}; if (!complete<2>(xr, uoff, uv, d.ctrl, g2)) PFAIL(); }
            if (first) uv[1] = (pf32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float x = uv[0][e], y = uv[1][e];
                x += dpp_mov<0xB1, 0xf>(0.f, x); x += dpp_mov<0x4E, 0xf>(0.f, x); x += dpp_mov<0x141, 0xf>(0.f, x);
                y += dpp_mov<0xB1, 0xf>(0.f, y); y += dpp_mov<0x4E, 0xf>(0.f, y); y += dpp_mov<0x141, 0xf>(0.f, y);
                uv[0][e] = x; uv[1][e] = y;
            }
            if (src == 0) {
                *reinterpret_cast<pf32x4*>(sm + B_TR + row * 4) = uv[0];
                *reinterpret_cast<pf32x4*>(sm + B_TR + 128 + row * 4) = uv[1];
            }
        }
        PABORT_CHECK();
        PSTAMP(10);
        {
            const float dm = sm[B_TR + er * 4 + ee];
            float dhs = dh0s + sm[B_TR + 128 + er * 4 + ee];
            const unsigned bits0 = __float_as_uint(B0_[2]);
            const float mh = (bits0 & 2u) ? d.keep : 0.f, mc = (bits0 & 1u) ? d.keep : 0.f;
            const pf32x4 dgv = cell_bwd(dm, dhs, dc0s, A0_[0], A0_[1], A0_[2], A0_[3], B0_[0], B0_[1], mh, mc);
            dh0s = dhs;
            if (ew) {
#pragma unroll
                for (int q = 0; q < 4; ++q) sm[B_TR + 384 + (q * 32 + er) * 4 + ee] = dgv[q];
            }
            __syncthreads();
            if (tid < 128) {
                const int gate = tid >> 5, row = tid & 31;
                const pf32x4 val = *reinterpret_cast<const pf32x4*>(sm + B_TR + 384 + tid * 4);
                const long o = ((((long)(gi * 8 + (gj >> 2)) * 2 + (row >> 4)) * 4 + (gj & 3)) * 64 + gate * 16 + (row & 15)) * 4;
                xpublish_near(xr, (unsigned)((BO_DG0 + slot * BDG + o) * 4), val, gen, near);
            }
            if (elive) { float* o = d.dg0 + sB * 4 * PH; o[o4H] = dgv[0]; o[o4H + PH] = dgv[1]; o[o4H + 2 * PH] = dgv[2]; o[o4H + 3 * PH] = dgv[3]; }
        }
        PSTAMP(11);
        {
            // A short timed pause between this workgroup's own publication of its gate gradients and the gather of everybody's: with the requests issued
            // right behind the write-through stores the stage in front (`cell-0 update backward + publish`) is 0.25 us longer.  Swept on one box (0 / 1 / 5 /
            // 15 / 30 / 45 / 60 ticks: frame 19.23 / 19.02 / 18.90 / 18.93 / 19.05 / 19.2 / 19.24 us; bf16: 12.86 / - / 12.70 / 12.68 / 12.77); the same pause
            // in front of the d[g1] gather moves 0.18 us from one stage into the next and gains nothing, and in front of the forward loop's requests it loses.
            if constexpr (BG0_LATE > 0) { const unsigned long long t__ = wall_clock64(); while (wall_clock64() - t__ < (unsigned long long)BG0_LATE) __builtin_amdgcn_s_sleep(1); }
            PROD_GATHER(BO_DG0)
            PROD_HALF(0, w0t, 0, wb0t, w0s, BO_PM0, true, BS0, false)
            PSTAMP(12);
            PROD_HALF(1, w0t, W0H1, wb0t, w0s, BO_PH0, false, false, W0L)
        }
        PSTAMP(13);
#ifndef EXP_NO_OPLOAD
        LOAD_OPERANDS1(s > 0 ? s - 1 : 0);
#endif
#if BPTT_KEYS_PER_STEP
        LOAD_KEYS();
#endif
        __syncthreads();                                             // the product's readers are done before the attention scratch is written
        PSTAMP(14);
    }
    if (tid == 0) {
        __hip_atomic_fetch_add(d.ctrl + 2, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (PROF && d.stamps) {
#pragma unroll
            for (int x = 0; x < NBSTAMP; ++x) d.stamps[(long)g * NBSTAMP + x] = sstamp[x];
        }
    }
#undef PSTAMP
#undef PABORT_CHECK
#undef PFAIL
#undef PROD_GATHER
#undef ISSUE_UPDATE1
#undef ISSUE_UPDATE0
#undef PROD_HALF
#undef LOAD_OPERANDS0
#undef LOAD_KEYS
#undef LOAD_ROWS
#undef LOAD_OPERANDS1
#undef STORE_ROWS
}

// ---- packers.  Contraction step ks of wave w's eighth: group mm = ks >> 2 -> producer column slice j' = 4 w + mm, unit 32 j' + 4 i + (ks & 3); the
// MFMA's inner index (lane >> 4) is the gate.  Output tile kt, row m = lane & 15 of the A operand: cell 1 half 0 = m0 unit 32 j + 16 kt + m
// (kernel row = that), half 1 = h1 unit (row H + ...); cell 0 half 0 = context column 24 j + 16 kt + m when 16 kt + m < 24 (else a zero
// row), half 1 = h0 unit (row M + ...).
__global__ void persist_pack_bwd_kernel(const float* __restrict__ w0f, const float* __restrict__ w1, float* __restrict__ w0t, float* __restrict__ w1t) {
    const long n = 256L * 8 * 64 * 64;
    for (long p = blockIdx.x * (long)blockDim.x + threadIdx.x; p < 2 * n; p += (long)gridDim.x * blockDim.x) {
        const bool c1 = p >= n;
        long r = c1 ? p - n : p;
        const int lane = (int)(r & 63); r >>= 6;
        const int reg = (int)(r & 63); r >>= 6;
        const int wave = (int)(r & 7), g = (int)(r >> 3);
        const int gi = g & 7, gj = g >> 3;
        const int half = reg >> 5, kt = (reg >> 4) & 1, ks = reg & 15;
        const int m = lane & 15, gate = lane >> 4;
        const int unit_c = 32 * (4 * wave + (ks >> 2)) + 4 * gi + (ks & 3);
        const long col = (long)gate * PH + unit_c;
        const int ok = 16 * kt + m;
        float v;
        if (c1) {
            const long row = (half == 0 ? 0 : PH) + 32 * gj + ok;
            v = w1[row * 4 * PH + col];
        } else if (half == 0) {
            v = ok < 24 ? w0f[(long)(24 * gj + ok) * 4 * PH + col] : 0.f;
        } else {
            v = w0f[(long)(PM + 32 * gj + ok) * 4 * PH + col];
        }
        if (c1) w1t[p - n] = v; else w0t[p] = v;
    }
}
// query kernel for the data gradient, by unit slice gi: float4 index (k4 * 4 + e) * 256 + l holds Wq[4 l + e][16 gi + 4 k4 .. + 3]
__global__ void persist_pack_wqt_kernel(const float* __restrict__ wq, float* __restrict__ wqt) {
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= 8 * 16 * 256 * 4) return;
    const int kq = p & 3, l = (p >> 2) & 255, ke = (p >> 10) & 15, gi = p >> 14;
    const int k4 = ke >> 2, e = ke & 3;
    wqt[p] = wq[(long)(4 * l + e) * PA + 16 * gi + 4 * k4 + kq];
}

// packed operand blocks -> the row-major histories mstts_decoder_train_bwd reads (fallback path; also what the tests compare)
__global__ void persist_unpack_history_kernel(const float* __restrict__ opk, int S, int B, float* acts0, float* acts1, float* craw0, float* craw1,
                                              float* c0, float* c1) {
    const long n = (long)S * PWG * 2 * 128;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const int tid = (int)(i & 127), cell = (int)((i >> 7) & 1), g = (int)((i >> 8) & 255), s = (int)(i >> 16);
        const int row = 16 * (tid >> 6) + (tid & 15), u = 4 * g + ((tid & 63) >> 4);
        if (row >= B) continue;
        const pf32x4* ob = reinterpret_cast<const pf32x4*>(opk) + opk_index(s, g, cell, 0, tid);
        const pf32x4 A = ob[0], Bv = ob[128];
        float* acts = (cell ? acts1 : acts0) + ((long)s * B + row) * 4 * PH + u;
        acts[0] = A[0]; acts[PH] = A[1]; acts[2 * PH] = A[2]; acts[3 * PH] = A[3];
        (cell ? craw1 : craw0)[((long)s * B + row) * PH + u] = Bv[0];
        (cell ? c1 : c0)[((long)s * B + row) * PH + u] = Bv[1];                // the zoned state BEFORE step s = slot s of the [S + 1, B, H] history
    }
}

}  // namespace mstts
using namespace mstts;

extern "C" int64_t mstts_persist_opk_floats(int64_t S) { return S * OPK_FLOATS_PER_STEP; }

extern "C" int mstts_persist_unpack_history(const float* opk, const mstts_decoder_train_desc* d, mstts_stream_t s) {
    MSTTS_REQUIRE(opk && d && d->acts0 && d->acts1 && d->craw0 && d->craw1 && d->c0 && d->c1 && d->H == PH && d->B >= 1 && d->B <= PROWS, MSTTS_ERR_SHAPE,
                  "persist_unpack_history: null pointer or unsupported shape");
    hipLaunchKernelGGL(persist_unpack_history_kernel, dim3(4096), dim3(256), 0, (hipStream_t)s, opk, (int)d->S, (int)d->B, d->acts0, d->acts1, d->craw0,
                       d->craw1, d->c0, d->c1);
    MSTTS_CHECK_LAUNCH("persist_unpack_history");
    return MSTTS_OK;
}

extern "C" int64_t mstts_persist_bwd_ws_bytes(void) { return BXCH_FLOATS * 4; }
extern "C" int64_t mstts_persist_bwd_pack_floats(int32_t which) { return which < 2 ? 256L * 8 * 64 * 64 : 8L * 16 * 256 * 4; }

extern "C" int32_t mstts_persist_bwd_supported(int64_t B, int64_t H, int64_t M, int64_t A, int64_t T, int64_t KS) {
    if (!(B >= 1 && B <= PROWS && H == PH && M == PM && A == PA && T >= 1 && T <= PTMAX && KS == PKS)) return 0;
    static int memo[PERSIST_MAX_DEVICES];
    return persist_device_memo(memo, [](int dev) {
        int cus = 0, per_cu = 0;
        if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus < PWG) return false;
        bool ok = true;
        int per = 0;
#define PBW_SETUP(P_, T_)                                                                                                                              \
        ok = ok && hipFuncSetAttribute((const void*)persist_bwd_kernel<P_, T_>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(BL<T_>::B_FLOATS * 4)) == hipSuccess && \
             hipOccupancyMaxActiveBlocksPerMultiprocessor(&per, (const void*)persist_bwd_kernel<P_, T_>, PTH, (size_t)BL<T_>::B_FLOATS * 4) == hipSuccess && per >= 1;
        PBW_SETUP(false, 128) PBW_SETUP(true, 128) PBW_SETUP(false, 256) PBW_SETUP(true, 256)
#undef PBW_SETUP
#define PBW_SETUP16(P_, T_)                                                                                                                            \
        ok = ok && hipFuncSetAttribute((const void*)persist_bwd_kernel<P_, T_, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(BL<T_>::B_FLOATS * 4)) == hipSuccess && \
             hipOccupancyMaxActiveBlocksPerMultiprocessor(&per, (const void*)persist_bwd_kernel<P_, T_, true>, PTH, (size_t)BL<T_>::B_FLOATS * 4) == hipSuccess && per >= 1;
        PBW_SETUP16(false, 128) PBW_SETUP16(true, 128) PBW_SETUP16(false, 256) PBW_SETUP16(true, 256)
#undef PBW_SETUP16
        (void)per_cu;
        return ok;
    });
}

extern "C" int mstts_persist_bwd_pack(const float* w0f, const float* w1, const float* wq, float* w0t, float* w1t, float* wqt, mstts_stream_t s) {
    MSTTS_REQUIRE(w0f && w1 && wq && w0t && w1t && wqt, MSTTS_ERR_SHAPE, "persist_bwd_pack: null pointer");
    hipLaunchKernelGGL(persist_pack_bwd_kernel, dim3(4096), dim3(256), 0, (hipStream_t)s, w0f, w1, w0t, w1t);
    MSTTS_CHECK_LAUNCH("persist_pack_bwd");
    hipLaunchKernelGGL(persist_pack_wqt_kernel, dim3(8 * 16 * 4), dim3(256), 0, (hipStream_t)s, wq, wqt);
    MSTTS_CHECK_LAUNCH("persist_pack_wqt");
    return MSTTS_OK;
}

extern "C" int mstts_decoder_train_bwd_persistent(const mstts_decoder_train_bwd_desc* bd, const mstts_persist_desc* p, mstts_stream_t s) {
    MSTTS_REQUIRE(bd && bd->fwd && p && bd->d_pj && bd->dg0 && bd->dg1 && bd->dq_hist && bd->de_hist && bd->d_in0 && p->w0pk && p->w1pk && p->wqpk &&
                  p->xch && p->ctrl, MSTTS_ERR_SHAPE, "decoder_train_bwd_persistent: null pointer");
    const mstts_decoder_train_desc* d = bd->fwd;
    const long B = d->B, S = d->S, H = d->H, M = d->lsa.M, A = d->lsa.A, T = d->lsa.T;
    MSTTS_REQUIRE(d->lsa.B == B && mstts_persist_bwd_supported(B, H, M, A, T, d->lsa.KS), MSTTS_ERR_SHAPE,
                  "decoder_train_bwd_persistent: shape or device not supported (see mstts_persist_bwd_supported)");
    MSTTS_REQUIRE(d->zc0 && d->zh0 && d->zc1 && d->zh1, MSTTS_ERR_SHAPE, "decoder_train_bwd_persistent: the four zoneout keep-masks are required");
    MSTTS_REQUIRE(p->opk, MSTTS_ERR_SHAPE, "decoder_train_bwd_persistent: the packed operand blocks of the persistent forward (opk) are required");
    MSTTS_REQUIRE(d->align_hist && d->cum_hist && d->q_hist && d->lsa.keys && d->lsa.values &&
                  d->lsa.loc_k && d->lsa.loc_b && d->lsa.score_w && d->lsa.score_b, MSTTS_ERR_SHAPE, "decoder_train_bwd_persistent: forward state missing");
    MSTTS_REQUIRE(aligned16(p->xch) && aligned16(bd->d_in0) && (M + H) % 4 == 0, MSTTS_ERR_ALIGN, "decoder_train_bwd_persistent: 16-byte alignment");
    hipStream_t hs = (hipStream_t)s;
    hipError_t e = hipMemsetAsync(p->xch, 0xFF, BXCH_FLOATS * 4, hs);
    if (e == hipSuccess) e = hipMemsetAsync(p->ctrl, 0, PCTRL_WORDS * sizeof(unsigned), hs);
    if (e != hipSuccess) return set_err(MSTTS_ERR_LAUNCH, "decoder_train_bwd_persistent: memset: %s", hipGetErrorString(e));
    PersistBwd a;
    a.w1t = p->w1pk; a.w0t = p->w0pk; a.wqt = p->wqpk;
    a.acts0 = d->acts0; a.acts1 = d->acts1; a.craw0 = d->craw0; a.craw1 = d->craw1; a.c0 = d->c0; a.c1 = d->c1;
    a.zc0 = d->zc0; a.zh0 = d->zh0; a.zc1 = d->zc1; a.zh1 = d->zh1; a.keep = 1.f - d->zoneout;
    a.align_hist = d->align_hist; a.cum_hist = d->cum_hist; a.q_hist = d->q_hist;
    a.keys = d->lsa.keys; a.values = d->lsa.values; a.lengths = d->lsa.lengths;
    a.loc_k = d->lsa.loc_k; a.loc_b = d->lsa.loc_b; a.score_w = d->lsa.score_w; a.score_b = d->lsa.score_b;
    a.d_pj = bd->d_pj; a.opk = p->opk; a.B = (int)B; a.S = (int)S; a.T = (int)T;
    a.dg0 = bd->dg0; a.dg1 = bd->dg1; a.dq_hist = bd->dq_hist; a.de_hist = bd->de_hist; a.d_in0 = bd->d_in0;
    a.xch = p->xch; a.ctrl = p->ctrl; a.stamps = (unsigned long long*)p->stamps; a.fail_step = p->selftest_fail_step > 0 ? p->selftest_fail_step - 1 : -1; a.near_xcd = p->near_xcd;
#define PBW_LAUNCH(T_)                                                                                                                  \
    {                                                                                                                                   \
        const size_t lds = (size_t)BL<T_>::B_FLOATS * 4;                                                                                \
        if (p->recurrent_bf16) {                                                                                                        \
            if (p->stamps) hipLaunchKernelGGL((persist_bwd_kernel<true, T_, true>), dim3(PWG), dim3(PTH), lds, hs, a);                  \
            else hipLaunchKernelGGL((persist_bwd_kernel<false, T_, true>), dim3(PWG), dim3(PTH), lds, hs, a);                           \
        } else {                                                                                                                        \
            if (p->stamps) hipLaunchKernelGGL((persist_bwd_kernel<true, T_>), dim3(PWG), dim3(PTH), lds, hs, a);                        \
            else hipLaunchKernelGGL((persist_bwd_kernel<false, T_>), dim3(PWG), dim3(PTH), lds, hs, a);                                 \
        }                                                                                                                               \
    }
    if (a.T <= 128) PBW_LAUNCH(128) else PBW_LAUNCH(256)      // (the 128-position instantiation; beyond that the positions from 128 on are read from memory)
#undef PBW_LAUNCH
    MSTTS_CHECK_LAUNCH("persist_bwd");
    return MSTTS_OK;
}
