// Persistent teacher-forced decoder loop for gfx950: ALL S steps of Decoder_Dynamic_Decode.body (Modules.py:397-443 - two
// ZoneoutLSTMCells, ZoneoutLSTMCell.py:228-271, and the Location_Sensitive_Attention step, Location_Sensitive_Attention.py:43-85)
// in ONE launch of 256 co-resident workgroups, one per CU.  Nothing recurrent is re-read from memory between steps:
//
//   * the two cell kernels (63 MB fp32) live in REGISTERS for the whole sequence: workgroup (i, j) = (id & 7, id >> 3) owns the
//     [K/8, 128]-tile "reduction slice i x gate-column slice j" of both kernels; each of its 8 waves (two per SIMD) keeps 16 gate
//     columns of that tile, 120 registers per lane, as MFMA A-operands;
//   * its 16 query units' slice of the query kernel (64 KB of LDS), its row's key slice (4 registers per lane) and its 96-column
//     slice of the row's values (48 KB of LDS) stay on chip as well.
//
// Per step the only traffic is the recurrent data itself, handed from CU to CU through write-through (sc1) stores and L1-bypassing
// (sc1) loads on small rings in memory, where THE DATA IS THE FLAG (persist_common.h): every exchanged word carries the generation
// of its ring slot in its last mantissa bit, and a consumer polls its piece until every word shows the generation it expects.  No
// barrier, no counter, no fence, no second store on the step path.  Six hand-offs per step:
//
//   ctx_{s-1} -> [cell-0 product, context rows]  -> partial gates -> (sum of 8, cell-0 update)  -> m0, h0
//   m0        -> [cell-1 product, input rows]    -> partial gates -> (sum of 8, cell-1 update)  -> m1, h1
//   m1 row    -> [16 query units, partial energies over those 16 units for all 128 positions]   -> partial energies
//   energies  -> (sum of 8, softmax, cumulative alignment, 96 context columns)                   -> ctx_s
//
// The recurrent halves of both products (h0_{s-1} . W0[h rows], h1_{s-1} . W1[h rows]) do not depend on the step's own chain and run
// in the shadow of the hand-offs.  Exact fp32 (v_mfma_f32_16x16x4_f32), fixed summation order (deterministic run to run).
//
// Every wait is bounded (wall clock); a workgroup that gives up raises an abort word that all others poll, the launch ends, and the
// host re-runs the sequence on the launch-per-step path (mstts_decoder_train_fwd).  The same happens when the 256 workgroups do not
// become co-resident (start rendezvous).
#include "persist_fwd_parts.h"

namespace mstts {

#ifndef M1_LATE
#define M1_LATE 60              // BF16 instantiation: 0 = the m1 row's request leaves right behind the h0 product; 1 = behind the location product; n > 1 = and n ticks (10 ns) into the stage
#endif
#ifndef PRE_EARLY
#define PRE_EARLY 1             // fp32 instantiation with the split on-chain products: the prenet rows' product of step s + 1 runs in the flight of step s' energies instead of at the loop top
#endif
#ifndef SPLIT_M0
#define SPLIT_M0 1              // 0: the on-chain cell-1 product on the f32-input MFMA like every other product of the loop (A/B builds)
#endif
#ifndef SPLIT_C0
#define SPLIT_C0 1              // 0: the on-chain cell-0 product (context + prenet rows) on the f32-input MFMA
#endif

// ring sizes in floats per slot
constexpr long XCTX = 8L * 128 * 24, XACT = 8L * 128 * 32, XPART = 256L * 8 * 2 * 256, XM1 = 32L * PH, XEN = 32L * 8 * PTMAX;
constexpr long OFF_CTX = 0, OFF_M0 = OFF_CTX + PRING * XCTX, OFF_H0 = OFF_M0 + PRING * XACT, OFF_H1 = OFF_H0 + PRING * XACT,
               OFF_M1 = OFF_H1 + PRING * XACT, OFF_EN = OFF_M1 + PRING * XM1, OFF_P0 = OFF_EN + PRING * XEN, OFF_P1 = OFF_P0 + PRING * XPART,
               XCH_FLOATS = OFF_P1 + PRING * XPART;
// LDS layout (floats)
// ONE staging buffer serves the four slices a step consumes, in turn: ctx_{s-1} -> m0_s -> h0_s -> h1_s (each is dead before the next arrives)
// (small arrays first: a DS instruction's immediate offset reaches 64 KB, and every access beyond that needs an address register of its
//  own, which the compiler hoists out of the step loop - with the two big flat arrays in front the kernel spilled 119 registers)
// TT = 128 or 256 encoder positions (the kernel's instantiations).  The value slice in LDS always covers positions 0 .. 127; with TT = 256 the
// positions from 128 on are read from memory (an XCD's 32 attention workgroups read 1.5 MB of them per step: they stay in its L2), the
// energies of a row (8 x 256) share the staging buffer, which is idle between the last product of a step and the first of the next.
template <int TT> struct FL {
    static constexpr int S_STG = 0, S_RED = S_STG + (TT == 128 ? 3 * 128 * 32 / 2 : 128 * LA),     // (TT = 128: room for a slice staged as three swizzled bf16 planes, SM0)
                         S_TR = S_RED + 4 * 2 * 256, S_M1 = S_TR + 2 * 128,
                         S_EN = TT > 128 ? S_STG : S_M1 + PH, S_CUM = TT > 128 ? S_M1 + PH : S_EN + 8 * PT,
                         S_A = S_CUM + TT + 48, S_Q = S_A + TT, S_QF = S_Q + 512, S_CO = S_QF + 16, S_LK = S_CO + 4 * 96,
                         S_FLAG = S_LK + 32 * 16, S_STAMP = S_FLAG + 4, S_VAL = S_STAMP + 2 * 16, S_WQ = S_VAL + PT * 96, S_FLOATS = S_WQ + 8 * 512 * 4;
    static_assert(8 * TT <= 128 * LA && S_FLOATS * 4 <= 160 * 1024, "LDS budget");
};
constexpr int NSTAMP = 16;

struct PersistFwd {
    const float* w0pk; const float* w1pk; const float* wqpk;
    const float* xw0; const float* b1;
    const float* pre; const float* b0;           // FOLD: prenet output [S, B, 256] and the cell-0 bias - the hoisted product xw0 is then formed inside the loop
    const uint8_t* zc0; const uint8_t* zh0; const uint8_t* zc1; const uint8_t* zh1; float keep;
    const float* keys; const float* values; const int32_t* lengths;
    const float* loc_k; const float* loc_b; const float* score_w; const float* score_b;
    int B, S, T;
    float* in0; float* in1; float* pj; float* c0; float* c1; float* acts0; float* acts1; float* craw0; float* craw1;
    float* q_hist; float* align_hist; float* cum_hist;
    float* opk;                                  // packed BPTT operands (persist_common.h), or null: the standard histories acts / craw / c are written instead
    float* xch; unsigned* ctrl;                  // ctrl[0] arrivals, ctrl[1] abort code, ctrl[2] workgroups that finished all S steps
    unsigned long long* stamps;                  // PROF: [256][NSTAMP] summed interval ticks
    int fail_step; int near_xcd;                               // self-test: workgroup 0 raises the abort word at this step (-1 = never)
};


// FOLD: the prenet rows of the cell-0 kernel ride along with the context rows (8 more k-steps per wave on the matrix cores) instead of
// arriving as a hoisted [S B, 4096] product: no 420 MB tensor written by a GEMM and read back by row-strided loads in every step.
// BF16 (BASELINE config 3, "bf16 with fp32 master"; FOLD only): both cell products and the query product take their operands rounded to
// bf16 - the kernels once, when they are loaded into registers, the activations when they are staged - and run on
// v_mfma_f32_16x16x32_bf16 with fp32 accumulators (persist_fwd_parts.h); cell states, gates, softmax, cumulative alignments, the history
// written for BPTT and everything exchanged between workgroups stay fp32.  Same geometry, same hand-offs, half the kernel registers.
template <bool PROF, bool FOLD, int TT, bool BF16 = false>
__global__ __launch_bounds__(PTH) void persist_fwd_kernel(PersistFwd d) {
    static_assert(!BF16 || FOLD, "the bf16 instantiation forms the prenet rows' product itself");
    // SM0: the on-chain cell-1 product m0 . W1[m0 rows] as the exact three-way bf16 split of BOTH operands (six products on
    // v_mfma_f32_16x16x32_bf16, fp32 accumulate: fp32 accuracy at 6/16 of the f32-input MFMA's matrix-core time, as gemm_split.inc) - the one
    // product of the loop whose kernel half fits as three planes (32 -> 48 registers per lane)
    constexpr bool SM0 = SPLIT_M0 && !BF16 && FOLD && TT == 128;
    constexpr bool PE = PRE_EARLY && SPLIT_M0 && SPLIT_C0 && !BF16 && FOLD && TT == 128;      // (= PRE_EARLY && SC0)
    // SC0: the same for the on-chain half of cell 0 (context rows + prenet rows: k-steps 0 .. 31 of w0).  Its 16 registers come from the
    // owner's biases, the score constants (both to a spare corner of S_Q, read where they are used) and SM0's location filter
    constexpr bool SC0 = SPLIT_C0 && SM0;
    constexpr int S_BIA = FL<TT>::S_Q + 128;          // [2 cells][4 gates][4 units] biases, [16] score bias + location bias, [16] score weights (S_Q itself uses 128 of its 512 floats)
    typedef FL<TT> Y;
    constexpr int S_STG = Y::S_STG, S_RED = Y::S_RED, S_TR = Y::S_TR, S_M1 = Y::S_M1, S_EN = Y::S_EN, S_CUM = Y::S_CUM, S_A = Y::S_A, S_Q = Y::S_Q,
                  S_CO = Y::S_CO, S_LK = Y::S_LK, S_FLAG = Y::S_FLAG, S_STAMP = Y::S_STAMP, S_VAL = Y::S_VAL, S_WQ = Y::S_WQ;
    constexpr int NH = TT / 128;                  // halves of 128 encoder positions
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int g0 = blockIdx.x, tid0 = threadIdx.x, wave0 = __builtin_amdgcn_readfirstlane(tid0 >> 6);
    int g = g0, gi = g & 7, gj = g >> 3;
    int tid = tid0, lane = tid & 63, wave = wave0;
    const int B = d.B, S = d.S, T = d.T;
    const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc(d.xch, 0, (int)(XCH_FLOATS * 4), 0x00020000);
    unsigned* sflag = reinterpret_cast<unsigned*>(sm + S_FLAG);
    float* stg = sm + S_STG;
    __bf16* stg16 = reinterpret_cast<__bf16*>(sm + S_STG);       // BF16: the staged slices as bf16, row stride LB16

    // ---------------- start rendezvous: all 256 workgroups must be resident before anyone waits for data
    if (d.near_xcd) persist_scrub(xr, OFF_CTX, OFF_M1 - OFF_CTX, g0, tid);      // context, m0, h0, h1 rings: the ones a slice group may keep in its L2
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

    // ---------------- once: this workgroup's constants.  Cell kernels -> registers (MFMA A operands), wave v = gate-column group v of the tile:
    //   w0[ks]: k-steps 0..23 = context rows of reduction slice gi, 24..31 = its prenet rows (FOLD), 32..63 = its h0 rows; w1[ks]: 0..31 = m0 rows, 32..63 = h1 rows
    float w0[BF16 ? 1 : 64], w1[BF16 ? 1 : 64];
    pbf16x8 wb0[BF16 ? 8 : 1], wb1[BF16 ? 8 : 1];                // BF16: the same 64 + 64 values as 8 + 8 packed octets (k-steps 8 j .. 8 j + 7)
    {
        const float* p0 = d.w0pk + ((long)(g * 8 + wave) * 64) * 64 + lane;
        const float* p1 = d.w1pk + ((long)(g * 8 + wave) * 64) * 64 + lane;
        if constexpr (BF16) {
#pragma unroll
            for (int r = 0; r < 64; ++r) { wb0[r >> 3][r & 7] = (__bf16)p0[r * 64]; wb1[r >> 3][r & 7] = (__bf16)p1[r * 64]; }
        } else {
#pragma unroll
            for (int r = 0; r < 64; ++r) w0[r] = (SC0 && r < 32) ? 0.f : (FOLD || r < 24 || r >= 32) ? p0[r * 64] : 0.f;
#pragma unroll
            for (int r = 0; r < 64; ++r) w1[r] = (SM0 && r < 32) ? 0.f : p1[r * 64];
        }
    }
    pbf16x8 w1s[SM0 ? 3 : 1][SM0 ? 4 : 1];                        // SM0: planes hi / mid / lo of the m0 rows, octet j = k-steps 8 j .. 8 j + 7
    pbf16x8 w0s[SC0 ? 3 : 1][SC0 ? 4 : 1];                        // SC0: ... of the context rows (octets 0 .. 2) and the prenet rows (octet 3)
    if constexpr (SM0) {
        const float* p1 = d.w1pk + ((long)(g * 8 + wave) * 64) * 64 + lane;
        const float* p0 = d.w0pk + ((long)(g * 8 + wave) * 64) * 64 + lane;
#pragma unroll
        for (int r = 0; r < 32; ++r) {
            {
                const float x = p1[r * 64];
                const __bf16 hi = (__bf16)x; const float r1 = x - (float)hi;
                const __bf16 mid = (__bf16)r1; const float r2 = r1 - (float)mid;
                w1s[0][r >> 3][r & 7] = hi; w1s[1][r >> 3][r & 7] = mid; w1s[2][r >> 3][r & 7] = (__bf16)r2;
            }
            if constexpr (SC0) {
                const float x = p0[r * 64];
                const __bf16 hi = (__bf16)x; const float r1 = x - (float)hi;
                const __bf16 mid = (__bf16)r1; const float r2 = r1 - (float)mid;
                w0s[0][r >> 3][r & 7] = hi; w0s[1][r >> 3][r & 7] = mid; w0s[2][r >> 3][r & 7] = (__bf16)r2;
            }
        }
    }
    // attention role: row ab = gj, unit slice gi (query units / key columns 16 gi ..), value columns 96 gi ..
    int ab = gj;
    const bool arow = ab < B;
    const int alen = arow ? (d.lengths ? d.lengths[ab] : T) : 0;
    int ak = tid & 15, atg = tid >> 4;                           // energy phase: attention unit 16 gi + ak, positions 4 atg .. 4 atg + 3
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
        if constexpr (SC0) {
            if (tid < 16) { sm[S_BIA + 32 + tid] = asb; sm[S_BIA + 48 + tid] = awk; }
            if (tid < 32) {      // cell tid >> 4, gate (tid >> 2) & 3, unit 4 g + (tid & 3)
                const int q = (tid >> 2) & 3, u = 4 * g + (tid & 3);
                sm[S_BIA + tid] = (tid < 16) ? d.b0[q * PH + u] : d.b1[q * PH + u];
            }
        }
        for (int x = tid; x < 32 * 16; x += PTH) sm[S_LK + x] = (x < PKS * 16) ? d.loc_k[(x >> 4) * PA + 16 * gi + (x & 15)] : 0.f;
        for (int x = tid; x < PT * 96; x += PTH) {
            const int t = x / 96, c = x - t * 96;
            sm[S_VAL + x] = (arow && t < alen && t < T) ? d.values[((long)ab * T + t) * PM + 96 * gi + c] : 0.f;
        }
        for (int x = tid; x < TT + 48; x += PTH) sm[S_CUM + x] = 0.f;
        const pf32x4* wqs = reinterpret_cast<const pf32x4*>(d.wqpk) + (long)gi * 8 * 512;      // query kernel slice: [8][512 threads] float4
        for (int x = tid; x < 8 * 512; x += PTH) {
            pf32x4 v = wqs[x];
            if constexpr (BF16) { v[0] = bf16_round(v[0]); v[1] = bf16_round(v[1]); v[2] = bf16_round(v[2]); v[3] = bf16_round(v[3]); }
            reinterpret_cast<pf32x4*>(sm + S_WQ)[x] = v;
        }
    }
    float lkb[SM0 ? 1 : 8];                                      // filter slice as MFMA B operand: lk[4 ks + (lane >> 4)][unit lane & 15]; tap 31 is zero
    __syncthreads();                                             // (SM0: read from LDS where it is used - its 8 registers are part of what the split planes cost)
    if constexpr (!SM0) {
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) lkb[ks] = sm[S_LK + (4 * ks + (lane >> 4)) * 16 + (lane & 15)];
    }
    // cell-update role (waves 0, 1): row er, hidden unit eu = 4 g + (lane >> 4); the states stay in registers for all S steps
    int et = wave & 1, er = 16 * et + (lane & 15), ee = lane >> 4, eu = 4 * g + ee;
    bool ew = wave < 2, elive = ew && er < B;
    float c0s = 0.f, h0s = 0.f, c1s = 0.f, h1s = 0.f;
    float b1v[4], b0v[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) { b1v[q] = SC0 ? 0.f : d.b1[q * PH + eu]; b0v[q] = (FOLD && !SC0) ? d.b0[q * PH + eu] : 0.f; }
    pf32x4 acc0[2], acc1[2];
#pragma unroll
    for (int b = 0; b < 2; ++b) { acc0[b] = (pf32x4){0.f, 0.f, 0.f, 0.f}; acc1[b] = acc0[b]; }
    // PROF: per-stage wall-clock ticks summed in LDS by thread 0 (no registers held across the loop)
    unsigned long long* sstamp = reinterpret_cast<unsigned long long*>(sm + S_STAMP);
    unsigned tprev = 0;
    if (PROF && tid < NSTAMP) sstamp[tid] = 0;
#define PSTAMP(idx) do { if (PROF && tid == 0) { const unsigned n__ = (unsigned)wall_clock64(); sstamp[idx] += (unsigned)(n__ - tprev); tprev = n__; } } while (0)
#define PABORT_CHECK() do { __syncthreads(); if (sflag[0]) return; } while (0)
    // (the first code raised stays: a workgroup that merely found the abort word while waiting does not overwrite it)
#define PFAIL() do { sflag[0] = 1; unsigned z__ = 0u; __hip_atomic_compare_exchange_strong(d.ctrl + 1, &z__, 2u, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); } while (0)
    // this wave's partial gates -> reducer (gate-column group `wave` of column slice gj), as source gi
#define PUBLISH_PARTIAL(OFFP, ACC)                                                                                              \
    _Pragma("unroll") for (int t = 0; t < 2; ++t) {                                                                             \
        const long piece = ((long)(gj * 8 + wave) * 8 + gi) * 2 + t;                                                            \
        xpublish(xr, (unsigned)(((OFFP) + slot * XPART + piece * 256) * 4 + 16 * lane), ACC[t], gen);                                  \
        ACC[t] = (pf32x4){0.f, 0.f, 0.f, 0.f};                                                                                  \
    }
    // waves 0..3 fetch the 8 x 2 partial tiles of this workgroup's 16 gate columns (sources 2 wave, 2 wave + 1) and pre-add the pair;
    // requested right behind the publication of this workgroup's own partials, completed where the sums are needed
#define ISSUE_PARTIALS(OFFP)                                                                                                    \
    if (wave < 4) {                                                                                                             \
        _Pragma("unroll") for (int m = 0; m < 4; ++m)                                                                           \
            poff[m] = (unsigned)(((OFFP) + slot * XPART + (((long)g * 8 + 2 * wave + (m >> 1)) * 2 + (m & 1)) * 256) * 4 + 16 * lane); \
        issue<4>(xr, poff, pv);                                                                                                 \
    }
#define COMPLETE_PARTIALS()                                                                                                     \
    if (wave < 4) {                                                                                                             \
        const unsigned gens__[4] = {gen, gen, gen, gen};                                                                        \
        if (!complete<4>(xr, poff, pv, d.ctrl, gens__)) PFAIL();                                                                        \
        *reinterpret_cast<pf32x4*>(sm + S_RED + ((wave * 2 + 0) * 64 + lane) * 4) = pv[0] + pv[2];                              \
        *reinterpret_cast<pf32x4*>(sm + S_RED + ((wave * 2 + 1) * 64 + lane) * 4) = pv[1] + pv[3];                              \
    }
#define SUM_PARTIALS()                                                                                                          \
    ((*reinterpret_cast<const pf32x4*>(sm + S_RED + ((0 * 2 + et) * 64 + lane) * 4) + *reinterpret_cast<const pf32x4*>(sm + S_RED + ((1 * 2 + et) * 64 + lane) * 4)) + \
     (*reinterpret_cast<const pf32x4*>(sm + S_RED + ((2 * 2 + et) * 64 + lane) * 4) + *reinterpret_cast<const pf32x4*>(sm + S_RED + ((3 * 2 + et) * 64 + lane) * 4)))
    __syncthreads();
    if (PROF && tid == 0) tprev = (unsigned)wall_clock64();

    // ---- cell-update operands of both cells (plain loads of loop-invariant inputs: hoisted prenet product, zoneout keep-masks).  They are
    // HBM-cold, so they are requested ONE STEP AHEAD - here for step 0, at the bottom of the loop body for step s + 1 - by EVERY wave
    // outside any condition, and consumed by every wave (waves 2..7 repeat the update of waves 0 / 1 and drop the result): a load that is
    // issued or consumed under a condition stays "pending" for the compiler's wait-count pass, and the s_waitcnt vmcnt it then places
    // also waits for write-through stores that have nothing to do with it.
    float xwv[4] = {0.f, 0.f, 0.f, 0.f};
    pf32x4 prv = {0.f, 0.f, 0.f, 0.f};           // FOLD: this thread's 16 bytes of the step's prenet slice (staging row rho = (tid & 255) >> 1, half tid & 1)
    uint8_t zc0v, zh0v, zc1v, zh1v;
#define LOAD_PRE(ST) do { const long bp__ = (long)(ST) * B;                                                                                  \
            const unsigned rho__ = ((unsigned)tid0 & 255u) >> 1, pb__ = 16 * ((rho__ >> 4) & 1u) + (rho__ & 15u);                                     \
            const pf32x4 v__ = *reinterpret_cast<const pf32x4*>(d.pre + (bp__ + (pb__ < (unsigned)B ? pb__ : 0u)) * 256 + 32 * (g0 & 7) + 8 * (rho__ >> 5) + 4 * ((unsigned)tid0 & 1u)); \
            prv = pb__ < (unsigned)B ? v__ : (pf32x4){0.f, 0.f, 0.f, 0.f}; } while (0)
#define LOAD_OPERANDS(ST) do { const long b__ = (long)(ST) * B; const unsigned r__ = (16 * (wave0 & 1) + (tid0 & 15)) < (unsigned)B ? 16 * (wave0 & 1) + (tid0 & 15) : 0u;  \
        /* (waves 2..7 only repeat the update: their lanes all read ONE address, a single cache-line request instead of 16 scattered ones) */ \
        const unsigned u__ = 4 * g0 + ((tid0 & 63) >> 4), h__ = wave0 < 2 ? r__ * PH + u__ : 0u, q__ = wave0 < 2 ? r__ * 4 * PH + u__ : 0u;                                     \
        if (FOLD) {                                                                                                                         \
            if (!PE) LOAD_PRE(ST);                                                                                                          \
        } else {                                                                                                                            \
            const float* xw__ = d.xw0 + b__ * 4 * PH;                                                                                       \
            _Pragma("unroll") for (int q = 0; q < 4; ++q) xwv[q] = xw__[q__ + q * PH];                                                       \
        }                                                                                                                                   \
        zc0v = (d.zc0 + b__ * PH)[h__]; zh0v = (d.zh0 + b__ * PH)[h__]; zc1v = (d.zc1 + b__ * PH)[h__]; zh1v = (d.zh1 + b__ * PH)[h__]; } while (0)
    LOAD_OPERANDS(0);
    if (PE) LOAD_PRE(0);

    for (int s = 0; s < S; ++s) {
        const unsigned slot = (unsigned)s & 3u, pslot = (unsigned)(s + 3) & 3u;                    // pslot: the slot of step s - 1
        const unsigned gen = ((unsigned)s >> 2) & 1u, pgen = ((unsigned)(s - 1) >> 2) & 1u;       // ... and the generations of the two
        // Every index below is re-derived from the thread / workgroup id behind an opaque asm once per step: left alone, the compiler
        // hoists some ninety loop-invariant ring offsets and LDS addresses out of the step loop, keeps them in VGPRs next to the 120
        // kernel registers, and spills into scratch inside the hand-off paths.
        tid = tid0; g = g0; wave = wave0;
        asm volatile("" : "+v"(tid));
        asm volatile("" : "+s"(g), "+s"(wave));
        lane = tid & 63; gi = g & 7; gj = g >> 3; ab = gj; ak = tid & 15; atg = tid >> 4;
        et = wave & 1; er = 16 * et + (lane & 15); ee = lane >> 4; eu = 4 * g + ee;
        ew = wave < 2; elive = ew && er < B;
        if (s == d.fail_step && g == 0 && tid == 0) {      // self-test: this workgroup leaves at its next check, the others find the abort word while they wait for its data
            __hip_atomic_store(d.ctrl + 1, 3u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            sflag[0] = 1;
        }
        const long sB = (long)s * B, sB1 = sB + B;
        const bool rowok = er < B;
        const unsigned erc = rowok ? (unsigned)er : 0u;
        const unsigned oH = erc * PH + eu, o4H = erc * 4 * PH + eu;
        unsigned soff[2], poff[4], roff[1];                          // requests in flight: a slice, the partial tiles, a row piece
        pf32x4 sv[2], pv[4], rv[1];
        // ================= A: cell 0, context rows.  In the shadow of the ctx_{s-1} hand-off: second half of h1_{s-1} . W1[h rows]
        if (FOLD) {
            // prenet slice of this step (requested one step ahead) -> columns 24..31 of the staging rows, multiplied at once: this runs in the
            // shadow of the context hand-off, which has nothing else to hide behind; then the context slice (columns 0..23) when it arrives
            if constexpr (BF16) {
                if (tid < 256) *reinterpret_cast<pbf16x4*>(stg16 + (tid >> 1) * LB16 + 24 + 4 * (tid & 1)) = to_bf16x4(prv);
                __syncthreads();
                mfma_part_bf16<3, 4, 0>(wb0, stg16, lane, acc0);
            } else if constexpr (SC0) {
                if (!PE || s == 0) {          // PE: step s + 1's rows are multiplied in the flight of step s' energies (stage F)
                    if (tid < 256) sp3_put_rk(stg16, tid >> 1, 6 + (tid & 1), prv);
                    __syncthreads();
                    mfma_part_split3<3, 4>(w0s, stg16, lane, acc0);
                }
            } else {
                if (tid < 256) *reinterpret_cast<pf32x4*>(stg + (tid >> 1) * LA + 24 + 4 * (tid & 1)) = prv;
                __syncthreads();
                mfma_part<6, 8, LA, 0, 64>(w0, stg, lane, acc0);
            }
            if (s > 0) {
                PSTAMP(0);
                slice_issue<6>(xr, OFF_CTX + pslot * XCTX + gi * 3072L, tid, soff, sv);
                if constexpr (BF16) { if (!slice_complete_bf16<6>(xr, stg16, tid, soff, sv, d.ctrl, pgen)) PFAIL(); }
                else if constexpr (SC0) { if (!slice_complete_split3_k6(xr, stg16, tid, soff, sv, d.ctrl, pgen)) PFAIL(); }
                else { if (!slice_complete<6, LA>(xr, stg, tid, soff, sv, d.ctrl, pgen)) PFAIL(); }
                PABORT_CHECK();
                PSTAMP(1);
                if constexpr (BF16) mfma_part_bf16<0, 3, 0>(wb0, stg16, lane, acc0);
                else if constexpr (SC0) mfma_part_split3<0, 3>(w0s, stg16, lane, acc0);
                else mfma_part<0, 6, LA, 0, 64>(w0, stg, lane, acc0);
            }
        } else {
            if (s > 0) {
                PSTAMP(0);
                slice_issue<6>(xr, OFF_CTX + pslot * XCTX + gi * 3072L, tid, soff, sv);
                if (!slice_complete<6, LC>(xr, stg, tid, soff, sv, d.ctrl, pgen)) PFAIL();
                PABORT_CHECK();
                PSTAMP(1);
            }
            if constexpr (!BF16) { if (s > 0) mfma_part<0, 6, LC, 0, 64>(w0, stg, lane, acc0); }
        }
        PUBLISH_PARTIAL(OFF_P0, acc0)
        PSTAMP(2);
        // in the shadow of the partial-gates hand-off (the longest wait of the step, and the staging buffer is free): h1_{s-1} . W1[h rows]
        if (s > 0) {
            slice_issue<8>(xr, OFF_H1 + pslot * XACT + gi * 4096L, tid, soff, sv);
            __syncthreads();                                         // the context rows are consumed by every wave
            if constexpr (BF16) { if (!slice_complete_bf16<8>(xr, stg16, tid, soff, sv, d.ctrl, pgen)) PFAIL(); }
            else { if (!slice_complete<8, LA>(xr, stg, tid, soff, sv, d.ctrl, pgen)) PFAIL(); }
            PABORT_CHECK();
            if constexpr (BF16) mfma_part_bf16<0, 4, 4>(wb1, stg16, lane, acc1);
            else mfma_part<0, 8, LA, 32, 64>(w1, stg, lane, acc1);
        }
        PSTAMP(13);
        // ================= B: sum of the eight partials, cell-0 update
        ISSUE_PARTIALS(OFF_P0)
        COMPLETE_PARTIALS()
        PABORT_CHECK();
        PSTAMP(3);
        {
            const pf32x4 gs = SUM_PARTIALS();
            float add0[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) add0[q] = SC0 ? sm[S_BIA + q * 4 + ee] : FOLD ? b0v[q] : (rowok ? xwv[q] : 0.f);
            const float cprev0 = c0s;
            const CellOut o = cell_update(gs, add0, c0s, h0s, (zc0v || !rowok) ? d.keep : 0.f, (zh0v || !rowok) ? d.keep : 0.f);
            if (ew) {
                sm[S_TR + er * 4 + ee] = o.m;
                sm[S_TR + 128 + er * 4 + ee] = h0s;
            }
            __syncthreads();
            if (tid < 64) {         // m0 / h0' of the 4 units for one row: 16 bytes into slice gi of the consumers' rings
                const int arr = tid >> 5, row = tid & 31;
                const pf32x4 val = *reinterpret_cast<const pf32x4*>(sm + S_TR + arr * 128 + row * 4);
                const int rho = (((gj & 3) * 2 + (row >> 4)) * 16) + (row & 15);
                const long base = arr ? OFF_H0 : OFF_M0;
                const long o2 = gi * 4096L + rho * 32 + 4 * (gj >> 2);
                xpublish_near(xr, (unsigned)((base + slot * XACT + o2) * 4), val, gen, near);
            }
            // what BPTT reads, behind the hand-off stores (ahead of them they delayed the publication by their issue time)
            if (ew && d.opk) {      // the cell-update operands as ONE contiguous block per workgroup, step and cell: 2 KB instead of 6 row-strided stores
                pf32x4* ob = reinterpret_cast<pf32x4*>(d.opk) + opk_index(s, g, 0, 0, tid);
                ob[0] = (pf32x4){o.si, o.tj, o.sf, o.so};
                ob[128] = (pf32x4){o.c, cprev0, __uint_as_float((zc0v ? 1u : 0u) | (zh0v ? 2u : 0u)), 0.f};
            }
            if (elive) {
                if (!d.opk) {
                    float* a = d.acts0 + sB * 4 * PH; a[o4H] = o.si; a[o4H + PH] = o.tj; a[o4H + 2 * PH] = o.sf; a[o4H + 3 * PH] = o.so;
                    (d.craw0 + sB * PH)[oH] = o.c;
                    (d.c0 + sB1 * PH)[oH] = c0s;
                }
                (d.in1 + sB * 2 * PH)[(unsigned)er * 2 * PH + eu] = o.m;
                (d.in0 + sB1 * (PM + PH))[(unsigned)er * (PM + PH) + PM + eu] = h0s;
            }
        }
        PSTAMP(4);
        // ================= C: cell 1, input rows (m0_s)   (every wave passed the barriers of B since it read the staged h1)
        slice_issue<8>(xr, OFF_M0 + slot * XACT + gi * 4096L, tid, soff, sv);
        if constexpr (BF16) { if (!slice_complete_bf16<8>(xr, stg16, tid, soff, sv, d.ctrl, gen)) PFAIL(); }
        else if constexpr (SM0) { if (!slice_complete_split3(xr, stg16, tid, soff, sv, d.ctrl, gen)) PFAIL(); }
        else { if (!slice_complete<8, LA>(xr, stg, tid, soff, sv, d.ctrl, gen)) PFAIL(); }
        PABORT_CHECK();
        PSTAMP(5);
        slice_issue<8>(xr, OFF_H0 + slot * XACT + gi * 4096L, tid, soff, sv);     // h0_s left its producers together with m0_s: it arrives under the product
        if constexpr (BF16) mfma_part_bf16<0, 4, 0>(wb1, stg16, lane, acc1);
        else if constexpr (SM0) mfma_part_split3(w1s, stg16, lane, acc1);
        else mfma_part<0, 8, LA, 0, 64>(w1, stg, lane, acc1);
        PUBLISH_PARTIAL(OFF_P1, acc1)
        PSTAMP(6);
        // in the shadow of the partial-gates hand-off: h0_s staged, first half of h0_s . W0[h rows] for step s+1
        __syncthreads();                                             // m0 is consumed by every wave
        if constexpr (BF16) { if (!slice_complete_bf16<8>(xr, stg16, tid, soff, sv, d.ctrl, gen)) PFAIL(); }
        else { if (!slice_complete<8, LA>(xr, stg, tid, soff, sv, d.ctrl, gen)) PFAIL(); }
        PABORT_CHECK();
        if constexpr (BF16) mfma_part_bf16<0, 2, 4>(wb0, stg16, lane, acc0);
        else mfma_part<0, 4, LA, 32, 64>(w0, stg, lane, acc0);
        PSTAMP(7);
        // ================= D: sum of the eight partials, cell-1 update
        ISSUE_PARTIALS(OFF_P1)
        COMPLETE_PARTIALS()
        PABORT_CHECK();
        PSTAMP(8);
        {
            const pf32x4 gs = SUM_PARTIALS();
            const float cprev1 = c1s;
            float add1[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) add1[q] = SC0 ? sm[S_BIA + 16 + q * 4 + ee] : b1v[q];
            const CellOut o = cell_update(gs, add1, c1s, h1s, (zc1v || !rowok) ? d.keep : 0.f, (zh1v || !rowok) ? d.keep : 0.f);
            if (ew) {
                sm[S_TR + er * 4 + ee] = o.m;
                sm[S_TR + 128 + er * 4 + ee] = h1s;
            }
            __syncthreads();
            if (tid < 64) {
                const int arr = tid >> 5, row = tid & 31;
                const pf32x4 val = *reinterpret_cast<const pf32x4*>(sm + S_TR + arr * 128 + row * 4);
                if (arr == 0) {     // m1: row-major [32][1024] for the attention workgroups of the row
                    const long o2 = (long)row * PH + 4 * g;
                    xpublish(xr, (unsigned)((OFF_M1 + slot * XM1 + o2) * 4), val, gen);
                } else {
                    const int rho = (((gj & 3) * 2 + (row >> 4)) * 16) + (row & 15);
                    const long o2 = gi * 4096L + rho * 32 + 4 * (gj >> 2);
                    xpublish_near(xr, (unsigned)((OFF_H1 + slot * XACT + o2) * 4), val, gen, near);
                }
            }
            if (ew && d.opk) {
                pf32x4* ob = reinterpret_cast<pf32x4*>(d.opk) + opk_index(s, g, 1, 0, tid);
                ob[0] = (pf32x4){o.si, o.tj, o.sf, o.so};
                ob[128] = (pf32x4){o.c, cprev1, __uint_as_float((zc1v ? 1u : 0u) | (zh1v ? 2u : 0u)), 0.f};
            }
            if (elive) {
                if (!d.opk) {
                    float* a = d.acts1 + sB * 4 * PH; a[o4H] = o.si; a[o4H + PH] = o.tj; a[o4H + 2 * PH] = o.sf; a[o4H + 3 * PH] = o.so;
                    (d.craw1 + sB * PH)[oH] = o.c;
                    (d.c1 + sB1 * PH)[oH] = c1s;
                }
                (d.pj + sB * (PH + PM))[(unsigned)er * (PH + PM) + eu] = o.m;
                (d.in1 + sB1 * 2 * PH)[(unsigned)er * 2 * PH + PH + eu] = h1s;
            }
        }
        PSTAMP(9);
        // in the shadow of the m1 hand-off: second half of h0_s . W0[h rows]
        if constexpr (BF16) mfma_part_bf16<2, 4, 4>(wb0, stg16, lane, acc0);
        else mfma_part<4, 8, LA, 32, 64>(w0, stg, lane, acc0);
        PSTAMP(10);
        // ================= E: attention, query units and partial energies of row ab
        if (PE) LOAD_PRE(s + 1 < S ? s + 1 : s);                      // (every wave, outside any condition: see LOAD_OPERANDS)
        unsigned long long t_e0 = 0;
        if constexpr (BF16 && M1_LATE > 1) t_e0 = wall_clock64();
        if (arow) {
            if (tid < 256) roff[0] = (unsigned)((OFF_M1 + slot * XM1 + (long)ab * PH) * 4 + 16 * tid);
            if constexpr (!(BF16 && M1_LATE)) { if (tid < 256) issue<1>(xr, roff, rv); }
            // while the m1 row is in flight: the location filter over the cumulative alignment (known since the last step's softmax) as a
            // Toeplitz product on the matrix core: loc[t][k] = sum_j cum[t + j - 15] lk[j][k] = A . B with A[t][j] = cum window (one LDS word per
            // lane and k-step), B[j][k] = the filter slice (8 registers, loaded once); wave w takes positions 16 w .. 16 w + 15, and the D layout
            // (position 4 (lane >> 4) + r, unit lane & 15) is exactly this thread's four positions - 8 MFMAs replace 62 LDS reads + 124 FMAs
            pf32x4 locv[NH];
#pragma unroll
            for (int hh = 0; hh < NH; ++hh) {                        // (TT = 256: the same for positions 128 + ...)
                locv[hh] = (pf32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int ks = 0; ks < 8; ++ks)
                    locv[hh] = PMFMA(sm[S_CUM + 128 * hh + 16 * wave + (lane & 15) + 4 * ks + (lane >> 4)],
                                     SM0 ? sm[S_LK + (4 * ks + (lane >> 4)) * 16 + (lane & 15)] : lkb[SM0 ? 0 : ks], locv[hh]);
            }
            if constexpr (BF16 && M1_LATE) {
                // BF16: the second half of the h0 product is 0.1 us, so a request right behind it reaches the memory side before the row does and comes
                // back stale - a second round trip (2.1 us of waiting where the fp32 instantiation, 0.5 us of product in between, waits 0.5).  Here the
                // request leaves behind the location product and, if that was not enough, M1_LATE ticks (10 ns) after the stage began
                if (M1_LATE > 1) { while (wall_clock64() - t_e0 < (unsigned long long)M1_LATE) __builtin_amdgcn_s_sleep(1); }
                if (tid < 256) issue<1>(xr, roff, rv);
            }
            if (tid < 256) {
                { const unsigned g1[1] = {gen}; if (!complete<1>(xr, roff, rv, d.ctrl, g1)) PFAIL(); }
                if constexpr (BF16) { rv[0][0] = bf16_round(rv[0][0]); rv[0][1] = bf16_round(rv[0][1]); rv[0][2] = bf16_round(rv[0][2]); rv[0][3] = bf16_round(rv[0][3]); }
                *reinterpret_cast<pf32x4*>(sm + S_M1 + 4 * tid) = rv[0];       // (BF16: the query product's operands are bf16 values; their products are exact in fp32)
            }
            PABORT_CHECK();
            PSTAMP(11);
            {       // thread (unit ak, hidden-unit group atg): rows 32 atg .. 32 atg + 31 of column 16 gi + ak of the query kernel
                float qp = 0.f;
#pragma unroll
                for (int x4 = 0; x4 < 8; ++x4) {
                    const pf32x4 mv = *reinterpret_cast<const pf32x4*>(sm + S_M1 + 32 * atg + 4 * x4);
                    const pf32x4 wv = *reinterpret_cast<const pf32x4*>(sm + S_WQ + (x4 * PTH + tid) * 4);
                    qp += wv[0] * mv[0]; qp += wv[1] * mv[1]; qp += wv[2] * mv[2]; qp += wv[3] * mv[3];
                }
                // the wave's four hidden-unit groups (lanes ak, ak + 16, ak + 32, ak + 48) meet through the LDS crossbar, the eight waves through
                // 128 floats of LDS behind ONE barrier; every thread then sums the eight wave partials of its unit itself (fixed order)
                qp += __shfl_xor(qp, 16);
                qp += __shfl_xor(qp, 32);
                if (lane < 16) sm[S_Q + wave * 16 + lane] = qp;
            }
            __syncthreads();
            float qsum = 0.f;
#pragma unroll
            for (int u = 0; u < 8; ++u) qsum += sm[S_Q + u * 16 + ak];
            if (tid < 16) (d.q_hist + (sB + ab) * PA + 16 * gi)[tid] = qsum;
            {
                const float qk = qsum + (SC0 ? sm[S_BIA + 32 + ak] : asb);
#pragma unroll
                for (int hh = 0; hh < NH; ++hh) {
                    const pf32x4 loc = locv[hh];
                    float pre[4];
#pragma unroll
                    for (int m = 0; m < 4; ++m) pre[m] = kreg[4 * hh + m] + qk + loc[m];
                    pf32x4 e4;
#pragma unroll
                    for (int m = 0; m < 4; ++m) {
                        float e = (SC0 ? sm[S_BIA + 48 + ak] : awk) * tanhf_(pre[m]);
                        e += dpp_mov<0xB1, 0xf>(0.f, e);
                        e += dpp_mov<0x4E, 0xf>(0.f, e);
                        e += dpp_mov<0x141, 0xf>(0.f, e);
                        e += dpp_mov<0x140, 0xf>(0.f, e);
                        e4[m] = e;
                    }
                    if (ak == 0) {
                        const long o = ((long)ab * 8 + gi) * TT + 128 * hh + 4 * atg;
                        xpublish(xr, (unsigned)((OFF_EN + slot * XEN + o) * 4), e4, gen);
                    }
                }
            }
        } else {
            PSTAMP(11);
        }
        PSTAMP(12);
        __syncthreads();                                             // (h0 is consumed by every wave before the next step stages the context)
        if constexpr (PE) if (s + 1 < S) {
            // The partial energies were published a barrier ago and need ~1 us to reach the eight workgroups of the row: a request sent now or
            // 0.5 us from now completes at the same time (swept, notes).  The prenet rows of the NEXT step's cell-0 product (teacher-forced input:
            // independent of this step's attention) are staged and multiplied here instead of at the loop top, where they stood between the
            // context's publication and its request with more work than that flight hides
            if (tid < 256) sp3_put_rk(stg16, tid >> 1, 6 + (tid & 1), prv);
            __syncthreads();
            mfma_part_split3<3, 4>(w0s, stg16, lane, acc0);
        }
        // ================= F: energies of the row, softmax, cumulative alignment, context columns 96 gi ..
        if (arow) {
            if (tid < 256 * NH) {                                    // 8 slices x TT energies = 256 NH pieces
                roff[0] = (unsigned)((OFF_EN + slot * XEN + (long)ab * 8 * TT) * 4 + 16 * tid);
                issue<1>(xr, roff, rv);
                { const unsigned g1[1] = {gen}; if (!complete<1>(xr, roff, rv, d.ctrl, g1)) PFAIL(); }
                *reinterpret_cast<pf32x4*>(sm + S_EN + 4 * tid) = rv[0];
            }
            PABORT_CHECK();
            PSTAMP(14);
            if (wave == 0) {        // masked softmax over the row's TT positions: lane holds positions lane + 64 i
                float ev[2 * NH];
                float mx = -INFINITY;
#pragma unroll
                for (int i = 0; i < 2 * NH; ++i) {
                    float e = 0.f;
#pragma unroll
                    for (int k = 0; k < 8; ++k) e += sm[S_EN + k * TT + 64 * i + lane];
                    ev[i] = (lane + 64 * i < alen) ? e : -INFINITY;
                    mx = fmaxf(mx, ev[i]);
                }
                mx = wave_max(mx);
                float ps = 0.f;
#pragma unroll
                for (int i = 0; i < 2 * NH; ++i) { ev[i] = (lane + 64 * i < alen) ? __expf(ev[i] - mx) : 0.f; ps += ev[i]; }
                const float inv = 1.f / wave_sum(ps);
                float* ah = d.align_hist + (sB + ab) * T; float* ch = d.cum_hist + (sB1 + ab) * T;
#pragma unroll
                for (int i = 0; i < 2 * NH; ++i) {
                    const float a = ev[i] * inv, n = sm[S_CUM + 15 + 64 * i + lane] + a;
                    sm[S_A + 64 * i + lane] = a;
                    sm[S_CUM + 15 + 64 * i + lane] = n;
                    if (gi == 0 && lane + 64 * i < T) { ah[lane + 64 * i] = a; ch[lane + 64 * i] = n; }
                }
            }
            __syncthreads();
            if (tid < 384) {
                const int c = tid % 96, th = tid / 96;
                float acc = 0.f;
#pragma unroll
                for (int t4 = 0; t4 < 8; ++t4) {                     // (the four alignment weights of a group in one 16-byte LDS read)
                    const pf32x4 a4 = *reinterpret_cast<const pf32x4*>(sm + S_A + 32 * th + 4 * t4);
#pragma unroll
                    for (int e = 0; e < 4; ++e) acc += a4[e] * sm[S_VAL + (32 * th + 4 * t4 + e) * 96 + c];
                }
                if (NH > 1) {       // positions 128 + 32 th ..: the values from memory (zero past the row's length there, Modules.py:87-93), L2 hits after the first step
                    const float* vg = d.values + ((long)ab * T + 128 + 32 * th) * PM + 96 * gi + c;
                    const int nt = T - (128 + 32 * th) < 32 ? (T - (128 + 32 * th) > 0 ? T - (128 + 32 * th) : 0) : 32;
#pragma unroll 16
                    for (int t = 0; t < 32; ++t) acc += sm[S_A + 128 + 32 * th + t] * (t < nt ? vg[(long)t * PM] : 0.f);
                }
                sm[S_CO + th * 96 + c] = acc;
            }
            __syncthreads();
            if (tid < 24) {
                const pf32x4 val = (*reinterpret_cast<const pf32x4*>(sm + S_CO + 4 * tid) + *reinterpret_cast<const pf32x4*>(sm + S_CO + 96 + 4 * tid)) +
                                   (*reinterpret_cast<const pf32x4*>(sm + S_CO + 192 + 4 * tid) + *reinterpret_cast<const pf32x4*>(sm + S_CO + 288 + 4 * tid));
                const int qq = tid / 6, k4 = tid - qq * 6;
                const int rho = ((qq * 2 + (ab >> 4)) * 16) + (ab & 15);
                const long o = gi * 3072L + rho * 24 + 4 * k4;
                xpublish_near(xr, (unsigned)((OFF_CTX + slot * XCTX + o) * 4), val, gen, near);
                reinterpret_cast<pf32x4*>(d.in0 + (sB1 + ab) * (PM + PH) + 96 * gi)[tid] = val;
                reinterpret_cast<pf32x4*>(d.pj + (sB + ab) * (PH + PM) + PH + 96 * gi)[tid] = val;
            }
        } else {
            PSTAMP(14);
            if (tid < 24) {         // rows past the batch: zero context, so that the cells' waits complete
                const int qq = tid / 6, k4 = tid - qq * 6;
                const int rho = ((qq * 2 + (ab >> 4)) * 16) + (ab & 15);
                const long o = gi * 3072L + rho * 24 + 4 * k4;
                xpublish_near(xr, (unsigned)((OFF_CTX + slot * XCTX + o) * 4), (pf32x4){0.f, 0.f, 0.f, 0.f}, gen, near);
            }
        }
        PSTAMP(15);
        LOAD_OPERANDS(s + 1 < S ? s + 1 : s);
    }
    if (tid == 0) {
        __hip_atomic_fetch_add(d.ctrl + 2, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (PROF && d.stamps) {
#pragma unroll
            for (int x = 0; x < NSTAMP; ++x) d.stamps[(long)g * NSTAMP + x] = sstamp[x];
        }
    }
#undef PSTAMP
#undef LOAD_OPERANDS
#undef LOAD_PRE
#undef PABORT_CHECK
#undef PFAIL
#undef PUBLISH_PARTIAL
#undef ISSUE_PARTIALS
#undef COMPLETE_PARTIALS
#undef SUM_PARTIALS
}

// ---- packers: the kernels in the order the lanes keep them (see the header comment)
// (unit_of_kstep: persist_fwd_parts.h)

__global__ void persist_pack_cells_kernel(const float* __restrict__ w0f, const float* __restrict__ w1, const float* __restrict__ wx0, float* __restrict__ w0pk, float* __restrict__ w1pk) {
    const long n0 = 256L * 8 * 64 * 64, n1 = 256L * 8 * 64 * 64;
    for (long p = blockIdx.x * (long)blockDim.x + threadIdx.x; p < n0 + n1; p += (long)gridDim.x * blockDim.x) {
        const bool c1 = p >= n0;
        long r = c1 ? p - n0 : p;
        const int lane = (int)(r & 63); r >>= 6;
        const int ks = (int)(r & 63); r >>= 6;
        const int wave = (int)(r & 7), g = (int)(r >> 3);
        const int gi = g & 7, gj = g >> 3;
        const int q = lane >> 4, mcol = lane & 15, ue = mcol >> 2, gate = mcol & 3;
        const int u = 32 * gj + 4 * wave + ue;
        const long col = (long)gate * PH + u;
        long row;
        // cell 0: k-steps 0..23 context rows 96 gi + 24 q + ks, 24..31 prenet rows 32 gi + 8 q + (ks - 24) of wx0 (zeros without it), 32..63 h0 rows
        if (!c1) row = ks < 24 ? 96 * gi + 24 * q + ks : ks < 32 ? 32 * gi + 8 * q + (ks - 24) : PM + unit_of_kstep(gi, ks - 32, q);
        else row = ks < 32 ? unit_of_kstep(gi, ks, q) : PH + unit_of_kstep(gi, ks - 32, q);
        if (c1) w1pk[p - n0] = w1[row * 4 * PH + col];
        else w0pk[p] = (ks >= 24 && ks < 32) ? (wx0 ? wx0[row * 4 * PH + col] : 0.f) : w0f[row * 4 * PH + col];
    }
}
// query kernel by unit slice gi: [8][8 x4][512 threads][4] - thread (unit ak = tid & 15, hidden-unit group atg = tid >> 4) finds rows
// 32 atg + 4 x4 .. + 3 of column 16 gi + ak in its x4-th float4 (a straight copy into LDS, conflict-free b128 reads)
__global__ void persist_pack_wq_kernel(const float* __restrict__ wq, float* __restrict__ wqpk) {
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= 8 * 64 * 256) return;
    const int e = p & 3, tid = (p >> 2) & 511, x4 = (p >> 11) & 7, gi = p >> 14;
    wqpk[p] = wq[(long)(32 * (tid >> 4) + 4 * x4 + e) * PA + 16 * gi + (tid & 15)];
}

}  // namespace mstts
using namespace mstts;

extern "C" int64_t mstts_persist_fwd_ws_bytes(void) { return XCH_FLOATS * 4; }
extern "C" int64_t mstts_persist_pack_floats(int32_t which) { return which == 0 ? 256L * 8 * 64 * 64 : which == 1 ? 256L * 8 * 64 * 64 : 8L * 64 * 256; }

/* 1 when the persistent loop can run this shape on the current device: reference widths, at most 32 rows and 256 encoder
 * positions (two instantiations: up to 128 positions everything of the attention stage is on chip, beyond that the value rows from 128 on are
 * re-read from the L2 every step), 31 filter taps, and a device that takes all 256 workgroups at once (one per CU - the occupancy query must admit the
 * kernel's LDS and registers, and the device must have at least 256 CUs) */
extern "C" int32_t mstts_persist_fwd_supported(int64_t B, int64_t H, int64_t M, int64_t A, int64_t T, int64_t KS) {
    if (!(B >= 1 && B <= PROWS && H == PH && M == PM && A == PA && T >= 1 && T <= PTMAX && KS == PKS)) return 0;
    static int memo[PERSIST_MAX_DEVICES];
    return persist_device_memo(memo, [](int dev) {
        int cus = 0, per_cu = 0;
        if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus < PWG) return false;
        bool ok = true;
        int per = 0;
#define PFW_SETUP(P_, F_, T_)                                                                                                                          \
        ok = ok && hipFuncSetAttribute((const void*)persist_fwd_kernel<P_, F_, T_>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(FL<T_>::S_FLOATS * 4)) == hipSuccess && \
             hipOccupancyMaxActiveBlocksPerMultiprocessor(&per, (const void*)persist_fwd_kernel<P_, F_, T_>, PTH, (size_t)FL<T_>::S_FLOATS * 4) == hipSuccess && per >= 1;
        PFW_SETUP(false, false, 128) PFW_SETUP(true, false, 128) PFW_SETUP(false, true, 128) PFW_SETUP(true, true, 128)
        PFW_SETUP(false, false, 256) PFW_SETUP(true, false, 256) PFW_SETUP(false, true, 256) PFW_SETUP(true, true, 256)
#undef PFW_SETUP
#define PFW_SETUP16(P_, T_)                                                                                                                            \
        ok = ok && hipFuncSetAttribute((const void*)persist_fwd_kernel<P_, true, T_, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(FL<T_>::S_FLOATS * 4)) == hipSuccess && \
             hipOccupancyMaxActiveBlocksPerMultiprocessor(&per, (const void*)persist_fwd_kernel<P_, true, T_, true>, PTH, (size_t)FL<T_>::S_FLOATS * 4) == hipSuccess && per >= 1;
        PFW_SETUP16(false, 128) PFW_SETUP16(true, 128) PFW_SETUP16(false, 256) PFW_SETUP16(true, 256)
#undef PFW_SETUP16
        (void)per_cu;
        return ok;
    });
}

extern "C" int mstts_persist_pack(const float* w0f, const float* w1, const float* wq, const float* wx0, float* w0pk, float* w1pk, float* wqpk, mstts_stream_t s) {
    MSTTS_REQUIRE(w0f && w1 && wq && w0pk && w1pk && wqpk, MSTTS_ERR_SHAPE, "persist_pack: null pointer");
    hipLaunchKernelGGL(persist_pack_cells_kernel, dim3(4096), dim3(256), 0, (hipStream_t)s, w0f, w1, wx0, w0pk, w1pk);
    MSTTS_CHECK_LAUNCH("persist_pack_cells");
    hipLaunchKernelGGL(persist_pack_wq_kernel, dim3(8 * 64), dim3(256), 0, (hipStream_t)s, wq, wqpk);
    MSTTS_CHECK_LAUNCH("persist_pack_wq");
    return MSTTS_OK;
}

extern "C" int mstts_decoder_train_fwd_persistent(const mstts_decoder_train_desc* d, const mstts_persist_desc* p, mstts_stream_t s) {
    const bool fold = p && p->pre != nullptr;
    MSTTS_REQUIRE(d && p && (fold ? p->b0 != nullptr : d->xw0 != nullptr) && d->b1 && d->in0 && d->in1 && d->pj && d->c0 && d->c1 && d->acts0 && d->acts1 && d->craw0 && d->craw1 &&
                  d->q_hist && d->align_hist && d->cum_hist && p->w0pk && p->w1pk && p->wqpk && p->xch && p->ctrl, MSTTS_ERR_SHAPE,
                  "decoder_train_fwd_persistent: null pointer");
    const long B = d->B, S = d->S, H = d->H, M = d->lsa.M, A = d->lsa.A, T = d->lsa.T;
    MSTTS_REQUIRE(d->lsa.B == B && mstts_persist_fwd_supported(B, H, M, A, T, d->lsa.KS), MSTTS_ERR_SHAPE,
                  "decoder_train_fwd_persistent: shape or device not supported (see mstts_persist_fwd_supported)");
    MSTTS_REQUIRE(d->zc0 && d->zh0 && d->zc1 && d->zh1, MSTTS_ERR_SHAPE, "decoder_train_fwd_persistent: the four zoneout keep-masks are required (all ones = no zoneout)");
    MSTTS_REQUIRE(d->lsa.keys && d->lsa.values && d->lsa.loc_k && d->lsa.loc_b && d->lsa.score_w && d->lsa.score_b, MSTTS_ERR_SHAPE,
                  "decoder_train_fwd_persistent: attention constants missing");
    MSTTS_REQUIRE(aligned16(p->xch) && aligned16(d->in0) && aligned16(d->pj) && (M + H) % 4 == 0, MSTTS_ERR_ALIGN, "decoder_train_fwd_persistent: 16-byte alignment");
    hipStream_t hs = (hipStream_t)s;
    // step-0 state (zero context / hidden / cell states / cumulative alignment), armed rings, cleared control words
    hipError_t e = hipMemsetAsync(d->in0, 0, B * (M + H) * sizeof(float), hs);
    if (e == hipSuccess) e = hipMemsetAsync(d->in1, 0, B * 2 * H * sizeof(float), hs);
    if (e == hipSuccess) e = hipMemsetAsync(d->c0, 0, B * H * sizeof(float), hs);
    if (e == hipSuccess) e = hipMemsetAsync(d->c1, 0, B * H * sizeof(float), hs);
    if (e == hipSuccess) e = hipMemsetAsync(d->cum_hist, 0, B * T * sizeof(float), hs);
    if (e == hipSuccess) e = hipMemsetAsync(p->xch, 0xFF, XCH_FLOATS * 4, hs);                  // every word "generation 1": stale for the first pass
    if (e == hipSuccess) e = hipMemsetAsync(p->ctrl, 0, PCTRL_WORDS * sizeof(unsigned), hs);
    if (e != hipSuccess) return set_err(MSTTS_ERR_LAUNCH, "decoder_train_fwd_persistent: memset: %s", hipGetErrorString(e));
    PersistFwd a;
    MSTTS_REQUIRE(!fold || (d->P == 256 && aligned16(p->pre)), MSTTS_ERR_SHAPE, "decoder_train_fwd_persistent: the folded prenet product needs a 256-wide prenet, 16-byte aligned");
    a.w0pk = p->w0pk; a.w1pk = p->w1pk; a.wqpk = p->wqpk; a.xw0 = d->xw0; a.b1 = d->b1; a.pre = p->pre; a.b0 = p->b0;
    a.zc0 = d->zc0; a.zh0 = d->zh0; a.zc1 = d->zc1; a.zh1 = d->zh1; a.keep = 1.f - d->zoneout;
    a.keys = d->lsa.keys; a.values = d->lsa.values; a.lengths = d->lsa.lengths;
    a.loc_k = d->lsa.loc_k; a.loc_b = d->lsa.loc_b; a.score_w = d->lsa.score_w; a.score_b = d->lsa.score_b;
    a.B = (int)B; a.S = (int)S; a.T = (int)T;
    a.in0 = d->in0; a.in1 = d->in1; a.pj = d->pj; a.c0 = d->c0; a.c1 = d->c1; a.acts0 = d->acts0; a.acts1 = d->acts1;
    a.craw0 = d->craw0; a.craw1 = d->craw1; a.q_hist = d->q_hist; a.align_hist = d->align_hist; a.cum_hist = d->cum_hist;
    a.opk = p->opk; a.xch = p->xch; a.ctrl = p->ctrl; a.stamps = (unsigned long long*)p->stamps; a.fail_step = p->selftest_fail_step > 0 ? p->selftest_fail_step - 1 : -1; a.near_xcd = p->near_xcd;
    MSTTS_REQUIRE(!p->recurrent_bf16 || fold, MSTTS_ERR_SHAPE, "decoder_train_fwd_persistent: the bf16 form needs the folded prenet product (pre / b0)");
#define PFW_LAUNCH(T_)                                                                                                                  \
    {                                                                                                                                   \
        const size_t lds = (size_t)FL<T_>::S_FLOATS * 4;                                                                                \
        if (p->recurrent_bf16) {                                                                                                        \
            if (p->stamps) hipLaunchKernelGGL((persist_fwd_kernel<true, true, T_, true>), dim3(PWG), dim3(PTH), lds, hs, a);            \
            else hipLaunchKernelGGL((persist_fwd_kernel<false, true, T_, true>), dim3(PWG), dim3(PTH), lds, hs, a);                     \
        } else if (fold) {                                                                                                              \
            if (p->stamps) hipLaunchKernelGGL((persist_fwd_kernel<true, true, T_>), dim3(PWG), dim3(PTH), lds, hs, a);                  \
            else hipLaunchKernelGGL((persist_fwd_kernel<false, true, T_>), dim3(PWG), dim3(PTH), lds, hs, a);                           \
        } else {                                                                                                                        \
            if (p->stamps) hipLaunchKernelGGL((persist_fwd_kernel<true, false, T_>), dim3(PWG), dim3(PTH), lds, hs, a);                 \
            else hipLaunchKernelGGL((persist_fwd_kernel<false, false, T_>), dim3(PWG), dim3(PTH), lds, hs, a);                          \
        }                                                                                                                               \
    }
    if (T <= 128) PFW_LAUNCH(128) else PFW_LAUNCH(256)   // (the 128-position instantiation keeps the whole value slice in LDS)
#undef PFW_LAUNCH
    MSTTS_CHECK_LAUNCH("persist_fwd");
    return MSTTS_OK;
}
