// tf.nn.dynamic_rnn over the two ZoneoutLSTMCells of Encoder_BiLSTM (Modules.py:49-73; cell: ZoneoutLSTMCell.py:188-271), all T steps of
// BOTH directions in ONE launch each way (forward pass, BPTT), for H = 256 and at most 32 rows: the encoder's recurrence was 128 steps x
// (1 + 2) dependent launches of 5-7 us.  Same scheme as the persistent decoder loops (persist.hip, persist_common.h), much smaller:
//   * forward: 32 workgroups per direction; workgroup gl owns hidden units 8 gl .. 8 gl + 7 and keeps their 32 gate columns of the
//     recurrent kernel [256, 1024] in registers as MFMA A operands (v_mfma_f32_16x16x4_f32, transposed orientation: a lane's
//     accumulator is the four gates of one (unit, row)); its 8 waves are 2 row tiles x 4 quarters of the contraction; the hidden state
//     of a step travels through a ring in global memory in exactly the order the consumers' lanes load it as B operands (one 16-byte
//     piece per lane and 16 k), tagged with the slot generation in the last mantissa bit: the data is the flag;
//   * BPTT: 16 workgroups per direction; workgroup gl owns 16 units, keeps their rows of the transposed kernel in registers, fetches
//     the step's gate gradients [32, 1024] (unit-major columns: a piece = the four gates of one (unit, row), written by its owner lane),
//     multiplies, reduces its four contraction quarters through LDS and applies the zoneout-cell backward for its 16 x 32 elements.
// One barrier and one hand-off per step.  Every wait is bounded; a time-out raises the abort word in ctrl[1] and the caller re-runs the
// launch-per-step form (mstts_lstm_seq_fwd_pair / mstts_lstm_seq_bwd_pair), which reads and writes the same buffers.
#include "persist_common.h"

namespace mstts {

constexpr int EH = 256, EFWG = 32, EBWG = 16, ETH = 512;
constexpr long EF_SLOT = 2L * 16 * 64 * 4;        // forward ring slot (one direction): h of one step as [row tile 2][j 16][lane 64][4]
constexpr long EB_SLOT = 2L * 64 * 64 * 4;        // backward ring slot: gate gradients of one step as [row tile 2][J 64][lane 64][4]
// Everything a step reads or writes besides the ring is PACKED by owner lane, so that a wave's access is one contiguous kilobyte: a
// wave-wide access to the row-major tensors touches 16 rows (an owner wave holds 4 units x 16 rows) and costs the wave 0.2 us of
// address processing, fifteen of them per step (measured: 3.8 us per forward step with them).  Small streaming kernels
// in front of / behind the two loops convert: forward inputs ipx (float4: hoisted gate inputs) + ipm (keep-mask bits), forward
// history epk (2 float4 per owner lane and step: gate activations | output, h, c, bits zc | zh << 1 | live << 2), BPTT input dop
// (upstream gradient), BPTT output dpk (float4 gate gradients).  Owner-lane index of the forward kernel: ((t * 64 + g) * 4 + wave) *
// 64 + lane with g = group * 32 + gl; of the BPTT kernel: ((k * G2 + g) * 8 + wave) * 64 + lane with g = group * 16 + gl, k = T - 1 - t.
// ROW GROUPS: rows are independent recurrences, so a sequence of more than 32 rows (the speaker-encoder trainer: 320) is cut into groups of 32
// rows, each with its own 32 (forward) / 16 (BPTT) workgroups and its own ring; group index = direction * groups_per_direction + row block.
constexpr long EP_GROUP = 32L * 4 * 64;           // owner lanes per step and row group: 8 192, forward and backward alike
// The FORWARD loop also exists for H = 128 (the Taco1 vocoder's BiRNN, Taco1_Mel_to_Spect/Modules.py:75-99; inference only - the BPTT kernel
// stays at 256): same scheme, 8 hidden units per workgroup, so H / 8 workgroups per row group, H / 16 k-steps per wave quarter.
template <int HH> struct LF {
    static constexpr int WG = HH / 8;                    // workgroups per row group
    static constexpr int KS = HH / 16;                   // k-steps of one wave's quarter of the contraction
    static constexpr int NJ = HH / 16;                   // 16-unit pieces per row tile of a ring slot
    static constexpr long SLOT = 2L * NJ * 64 * 4;       // ring slot in floats
    static constexpr long PGROUP = (long)WG * 4 * 64;    // owner lanes per step and row group
};
static_assert(LF<256>::WG == EFWG && LF<256>::SLOT == EF_SLOT && LF<256>::PGROUP == EP_GROUP, "H = 256 geometry");
// float4 index into epk: [t][forward workgroup G][owner wave 4][half 2][lane 64]
__host__ __device__ __forceinline__ long epk_index(int t, int G, int g, int ow, int half, int lane) { return ((((long)t * G + g) * 4 + ow) * 2 + half) * 64 + lane; }

struct EncFwdDir {
    const float* xw; const float* wpk; const int32_t* lengths; const uint8_t* zc; const uint8_t* zh;
    float* out; long out_sb, out_st;
    float* c_hist; float* h_hist; float* acts; float* c_raw;
    int reverse;
};
struct EncFwd { EncFwdDir d[2]; int ndir, gpd, B, T; float keep; float* xch; unsigned* ctrl; const pf32x4* ipx; const unsigned* ipm; pf32x4* epk; };     // gpd: row groups per direction
struct EncBwdDir {
    const float* wtpk; const int32_t* lengths; const uint8_t* zc; const uint8_t* zh;
    const float* d_out; long dout_sb, dout_st;
    const float* c_hist; const float* acts; const float* c_raw;
    float* dgates_step; float* dgates_pos;
    int reverse;
};
struct EncBwd { EncBwdDir d[2]; int ndir, gpd, B, T; float keep; float* xch; unsigned* ctrl; const pf32x4* epk; const float* dop; pf32x4* dpk; };

// start rendezvous of n workgroups (thread 0 of each); false on time-out / abort
__device__ __forceinline__ bool lstm_rendezvous(unsigned* ctrl, unsigned n) {
    __hip_atomic_fetch_add(ctrl, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const unsigned long long t0 = wall_clock64();
    while (__hip_atomic_load(ctrl, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < n) {
        __builtin_amdgcn_s_sleep(8);
        if (wall_clock64() - t0 > PERSIST_RENDEZVOUS_TICKS || __hip_atomic_load(ctrl + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) {
            __hip_atomic_store(ctrl + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            return false;
        }
    }
    return true;
}
#define LFAIL(ctrl) do { sflag[0] = 1; unsigned z__ = 0u; __hip_atomic_compare_exchange_strong((ctrl) + 1, &z__, 2u, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); } while (0)

template <int HH>
__global__ __launch_bounds__(ETH) void persist_lstm_fwd_kernel(EncFwd p) {
    typedef LF<HH> L;
    constexpr int KS = L::KS, NPC = L::KS / 4;
    __shared__ __attribute__((aligned(16))) float red[2 * 4 * 2 * 2 * 64 * 4];      // [buffer][quarter][row tile][unit tile][lane][4]
    __shared__ unsigned sflag[2];
    const int g = blockIdx.x, grp = g / L::WG, gl = g % L::WG, dir = grp / p.gpd, row0 = 32 * (grp % p.gpd);
    const int G = p.ndir * p.gpd * L::WG;
    const long ep = (long)p.ndir * p.gpd * L::PGROUP;
    const EncFwdDir& d = p.d[dir];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int rt = wave & 1, kq4 = wave >> 1, n = lane & 15, q = lane >> 4;
    const int B = p.B, T = p.T;
    const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc(p.xch + (long)grp * PRING * L::SLOT, 0, (int)(PRING * L::SLOT * 4), 0x00020000);
    if (tid == 0) { sflag[0] = lstm_rendezvous(p.ctrl, (unsigned)G) ? 0u : 1u; }
    __syncthreads();
    if (sflag[0]) return;
    // this wave's slice of the recurrent kernel: 2 unit tiles x KS k-steps (its quarter of the H hidden units)
    float wa[2][KS];
#pragma unroll
    for (int ut = 0; ut < 2; ++ut)
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) wa[ut][ks] = d.wpk[((((long)gl * 4 + kq4) * 2 + ut) * KS + ks) * 64 + lane];
    // cell-update role (waves 0..3): row tile o_rt, unit tile o_ut; lane = (unit within the tile q, row n)
    const bool owner = wave < 4;
    const int o_rt = wave & 1, o_ut = (wave >> 1) & 1;
    const int b = row0 + 16 * o_rt + n, u = 8 * gl + 4 * o_ut + q;
    const bool brow = owner && b < B;
    const int len = brow ? (d.lengths ? d.lengths[b] : T) : 0;
    float cs = 0.f, hs = 0.f;
    // The update's operands (hoisted input product, keep-mask bits; packed: one contiguous KB per wave) are requested one step ahead, right
    // BEHIND a completed gather: vector memory returns in order, so a request in front of a poll holds the poll's answer back until it
    // has come in itself.
    pf32x4 xa = {0.f, 0.f, 0.f, 0.f}, xb = xa;
    unsigned zca = 3, zcb = 3;
    const long own = (long)g * 256 + (wave & 3) * 64 + lane;     // + t * ep
    // (unconditional, in every wave, clamped at the last step: a conditional request leaves the compiler's wait-count pass with a "maybe
    //  pending" load at the loop's back edge, and the s_waitcnt vmcnt(0) it then places there also waits for the write-through
    //  publication to be acknowledged - 0.5 us per step)
#define LSTM_FWD_OPERANDS(TT, XV, ZM)                                                                                              \
    { const long t__ = (TT) < T ? (TT) : T - 1; XV = p.ipx[t__ * ep + own]; ZM = p.ipm[t__ * ep + own]; }
    LSTM_FWD_OPERANDS(0, xa, zca)
#ifdef LSTM_PROF
    unsigned long long st_[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    unsigned tp_ = (unsigned)wall_clock64();
#define LSTAMP(i) do { if (tid == 0) { const unsigned n__ = (unsigned)wall_clock64(); st_[i] += (unsigned)(n__ - tp_); tp_ = n__; } } while (0)
#else
#define LSTAMP(i) do { } while (0)
#endif
    for (int t = 0; t < T; ++t) {
        const unsigned slot = t & 3, gen = (t >> 2) & 1, pslot = (t + 3) & 3, pgen = ((t - 1) >> 2) & 1;
        const bool live = brow && t < len;
        const int pos = (d.reverse && live) ? len - 1 - t : t;
        LSTAMP(0);
        const pf32x4 xv = xa;
        const unsigned zcv = zca & 1u, zhv = zca & 2u;
        pf32x4 acc[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
        if (t > 0) {
            unsigned off[NPC], gens[NPC];
            pf32x4 hv[NPC];
#pragma unroll
            for (int jj = 0; jj < NPC; ++jj) { off[jj] = (unsigned)((pslot * L::SLOT + ((rt * L::NJ + NPC * kq4 + jj) * 64 + lane) * 4) * 4); gens[jj] = pgen; }
            if (!gather<NPC>(xr, off, hv, p.ctrl, gens)) LFAIL(p.ctrl);
            LSTAMP(1);
            LSTM_FWD_OPERANDS(t + 1, xb, zcb)
#pragma unroll
            for (int jj = 0; jj < NPC; ++jj)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    acc[0] = PMFMA(wa[0][4 * jj + e], hv[jj][e], acc[0]);
                    acc[1] = PMFMA(wa[1][4 * jj + e], hv[jj][e], acc[1]);
                }
        } else {
            LSTM_FWD_OPERANDS(t + 1, xb, zcb)
        }
        float* rb = red + (t & 1) * (4 * 2 * 2 * 64 * 4);
#pragma unroll
        for (int ut = 0; ut < 2; ++ut) *reinterpret_cast<pf32x4*>(rb + ((((kq4 * 2 + rt) * 2 + ut) * 64) + lane) * 4) = acc[ut];
        LSTAMP(2);
        __syncthreads();
        LSTAMP(3);
        if (sflag[0]) return;
        // next step's operands must have LANDED here, in front of the update's stores: left to the end of the loop body the register
        // hand-over waits with vmcnt(0) behind the write-through publication (the stores sit in a conditional block, the wait-count pass
        // cannot count them) - 0.5 us per step
        xa = xb; zca = zcb;
        asm volatile("" : "+v"(xa), "+v"(zca));
        if (owner) {
            pf32x4 gs = xv;
#pragma unroll
            for (int k4 = 0; k4 < 4; ++k4) gs += *reinterpret_cast<const pf32x4*>(rb + ((((k4 * 2 + o_rt) * 2 + o_ut) * 64) + lane) * 4);
            // ZoneoutLSTMCell.py:228-271 (gates i, j, f, o; forget bias 1; training-mode zoneout) under dynamic_rnn's length rule
            float si = sigmoidf_(gs[0]), tj = tanhf_(gs[1]), sf = sigmoidf_(gs[2] + 1.0f), so = sigmoidf_(gs[3]);
            float c = sf * cs + si * tj;
            float m = so * tanhf_(c);
            const float kc = zcv ? p.keep : 0.f, kh = zhv ? p.keep : 0.f;
            float hn = kh * (m - hs) + hs, cn = kc * (c - cs) + cs;
            if (!live) { m = 0.f; hn = hs; cn = cs; si = 0.f; tj = 0.f; sf = 0.f; so = 0.f; c = cs; }
            hs = hn; cs = cn;
            // publish h_t: the piece of (row, 4 consecutive units) is the four unit lanes of this row
            pf32x4 pv;
#pragma unroll
            for (int e = 0; e < 4; ++e) pv[e] = __shfl(hn, 16 * e + n, 64);
            if (q == 0)
                xpublish(xr, (unsigned)((slot * L::SLOT + ((o_rt * L::NJ + (gl >> 1)) * 64 + ((2 * (gl & 1) + o_ut) * 16 + n)) * 4) * 4), pv, gen);
            LSTAMP(4);
            // history of this (row, unit) for the BPTT and for the row-major tensors (persist_lstm_unpack_fwd_kernel): two contiguous KB per wave
            pf32x4* eo = p.epk + epk_index(t, G, g, wave, 0, lane);
            eo[0] = (pf32x4){si, tj, sf, so};
            eo[64] = (pf32x4){m, hn, cn, __uint_as_float((zcv ? 1u : 0u) | (zhv ? 2u : 0u) | (live ? 4u : 0u))};
            LSTAMP(5);
        }
    }
#ifdef LSTM_PROF
    if (tid == 0 && g == 0) for (int i = 0; i < 6; ++i) reinterpret_cast<unsigned long long*>(p.xch + 2 * PRING * L::SLOT)[i] = st_[i];
#endif
    if (tid == 0) __hip_atomic_fetch_add(p.ctrl + 2, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

__global__ __launch_bounds__(ETH) void persist_lstm_bwd_kernel(EncBwd p) {
    __shared__ __attribute__((aligned(16))) float red[2 * 4 * 2 * 64 * 4];          // [buffer][quarter][row tile][lane][4]
    __shared__ unsigned sflag[2];
    const int g = blockIdx.x, grp = g / EBWG, gl = g % EBWG, dir = grp / p.gpd, row0 = 32 * (grp % p.gpd);
    const int G = p.ndir * p.gpd * EFWG;              // forward workgroups (the packed history's geometry)
    const long ep = (long)p.ndir * p.gpd * EP_GROUP;
    const EncBwdDir& d = p.d[dir];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int rt = wave & 1, kq4 = wave >> 1, n = lane & 15, q = lane >> 4;
    const int B = p.B, T = p.T;
    const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc(p.xch + (long)grp * PRING * EB_SLOT, 0, (int)(PRING * EB_SLOT * 4), 0x00020000);
    if (tid == 0) { sflag[0] = lstm_rendezvous(p.ctrl, (unsigned)(p.ndir * p.gpd * EBWG)) ? 0u : 1u; }
    __syncthreads();
    if (sflag[0]) return;
    // rows 16 gl .. + 15 of the recurrent kernel (the hidden units this workgroup owns), this wave's quarter of the 1024 gate columns
    float wt[64];
#pragma unroll
    for (int ks = 0; ks < 64; ++ks) wt[ks] = d.wtpk[(((long)gl * 4 + kq4) * 64 + ks) * 64 + lane];
    // cell-backward role (all 8 waves): row tile o_rt = rt, accumulator component r = kq4: unit 4 q + r of the tile, row n
    const int r = kq4;
    const int b = row0 + 16 * rt + n, u = 16 * gl + 4 * q + r;
    const bool brow = b < B;
    const int len = brow ? (d.lengths ? d.lengths[b] : T) : 0;
    float dcs = 0.f, dhc = 0.f;                       // gradients of the carried cell / hidden state (from the later steps)
    // operands of the cell backward two steps ahead, requested behind a completed gather (see the forward kernel)
    // this lane's element (unit u, row b) in the forward kernel's owner order: workgroup u >> 3, owner wave rt + 2 ((u >> 2) & 1), lane 16 (u & 3) + n
    const int fg = grp * EFWG + (u >> 3), fw = rt + 2 * ((u >> 2) & 1), fl = 16 * (u & 3) + n;
    const long bown = (long)g * 512 + wave * 64 + lane;      // + k * ep
    struct BwdOps { pf32x4 act, st, prev; float dout; };
    BwdOps oa, ob;
    oa.act = oa.st = oa.prev = (pf32x4){0.f, 0.f, 0.f, 0.f}; oa.dout = 0.f;
    ob = oa;
#define LSTM_BWD_OPERANDS(KK, O)                                                                                                   \
    {                                                                                                                              \
        const int kk__ = (KK) < T ? (KK) : T - 1, tt__ = T - 1 - kk__;                                                             \
        O.act = p.epk[epk_index(tt__, G, fg, fw, 0, fl)];                                                                             \
        O.st = p.epk[epk_index(tt__, G, fg, fw, 1, fl)];                                                                             \
        O.prev = p.epk[epk_index(tt__ > 0 ? tt__ - 1 : 0, G, fg, fw, 1, fl)];                                                         \
        O.dout = p.dop[(long)kk__ * ep + bown];                                                                                    \
    }
    LSTM_BWD_OPERANDS(0, oa)
    for (int k = 0; k < T; ++k) {
        const int t = T - 1 - k;
        const unsigned slot = k & 3, gen = (k >> 2) & 1, pslot = (k + 3) & 3, pgen = ((k - 1) >> 2) & 1;
        const unsigned bits = __float_as_uint(oa.st[3]);
        const bool live = (bits & 4u) != 0u;
        const pf32x4 av = oa.act;
        const float cp = t > 0 ? oa.prev[2] : 0.f;
        float dm = oa.dout;
        const unsigned zcv = bits & 1u, zhv = bits & 2u;
#define LSTM_BWD_LAND() do { oa = ob; asm volatile("" : "+v"(oa.act), "+v"(oa.st), "+v"(oa.prev), "+v"(oa.dout)); } while (0)
        float dh_rec = 0.f;
        if (k > 0) {
            unsigned off[16];
            pf32x4 gv[16];
            unsigned gens[16];
#pragma unroll
            for (int jj = 0; jj < 16; ++jj) {
                off[jj] = (unsigned)((pslot * EB_SLOT + ((rt * 64 + 16 * kq4 + jj) * 64 + lane) * 4) * 4);
                gens[jj] = pgen;
            }
            if (!gather<16>(xr, off, gv, p.ctrl, gens)) LFAIL(p.ctrl);
            LSTM_BWD_OPERANDS(k + 1, ob)
            pf32x4 acc = {0.f, 0.f, 0.f, 0.f}, acc2 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int jj = 0; jj < 16; ++jj)
#pragma unroll
                for (int e = 0; e < 4; e += 2) {
                    acc = PMFMA(wt[4 * jj + e], gv[jj][e], acc);
                    acc2 = PMFMA(wt[4 * jj + e + 1], gv[jj][e + 1], acc2);
                }
            acc += acc2;
            float* rb = red + (k & 1) * (4 * 2 * 64 * 4);
            *reinterpret_cast<pf32x4*>(rb + (((kq4 * 2 + rt) * 64) + lane) * 4) = acc;
            __syncthreads();
            if (sflag[0]) return;
#pragma unroll
            for (int k4 = 0; k4 < 4; ++k4) dh_rec += rb[(((k4 * 2 + rt) * 64) + lane) * 4 + r];
        } else {
            LSTM_BWD_OPERANDS(k + 1, ob)
        }
        // zoneout-cell backward of (row b, unit u) at step t (elementwise.hip: lstm_point_bwd_kernel)
        const float dhs = dhc + dh_rec;
        pf32x4 dg = {0.f, 0.f, 0.f, 0.f};
        if (live) {
            const float mh = zhv ? p.keep : 0.f, mc = zcv ? p.keep : 0.f;
            dm += mh * dhs;
            const float si = av[0], tj = av[1], sf = av[2], so = av[3];
            const float tc = tanhf_(sf * cp + si * tj);            // the raw cell state of the step, as the forward kernel formed it
            const float dc = dm * so * (1.f - tc * tc) + mc * dcs;
            dg[3] = dm * tc * so * (1.f - so);
            dg[0] = dc * tj * si * (1.f - si);
            dg[1] = dc * si * (1.f - tj * tj);
            dg[2] = dc * cp * sf * (1.f - sf);
            dcs = dcs * (1.f - mc) + dc * sf;
            dhc = dhs * (1.f - mh);
        } else {
            dhc = dhs;
        }
        LSTM_BWD_LAND();                                    // next step's operands have landed before this step's stores go out (see the forward kernel)
        // the piece of (row b, unit u) = its four gate gradients: ring column 4 u + gate
        const int uu = 16 * gl + 4 * q + r;
        xpublish(xr, (unsigned)((slot * EB_SLOT + ((rt * 64 + (uu >> 2)) * 64 + ((uu & 3) * 16 + n)) * 4) * 4), dg, gen);
        p.dpk[(long)k * ep + bown] = dg;               // -> dgates_step / dgates_pos by persist_lstm_unpack_bwd_kernel
    }
    if (tid == 0) __hip_atomic_fetch_add(p.ctrl + 2, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// ---- the streaming kernels around the two loops: owner-lane order <-> row-major tensors
// (ngr = row groups in all, gpd = per direction; a group is 32 forward / 16 BPTT workgroups.)  Owner lanes hold (row, 4 gates of one unit):
// a wave-wide access to a row-major tensor from that order is 64 scattered words, so every kernel here goes through LDS - whole rows
// (16-byte pieces, a kilobyte per wave) on the row-major side, the owner lanes' contiguous pieces on the packed side.  One block handles 8
// rows of one (step, row group): rows row0 .. row0 + 7 with row0 = 32 (group % gpd) + 8 o; in the forward order (persist_lstm_fwd_kernel)
//     owner lane ((t * G + group * WG + gl) * 4 + ow) * 64 + lane  holds row 32 (..) + 16 (ow & 1) + (lane & 15), unit 8 gl + 4 (ow >> 1) + (lane >> 4)
// and in the BPTT order (persist_lstm_bwd_kernel)
//     owner lane ((k * G2 + group * 16 + gl) * 8 + wave) * 64 + lane  holds row 32 (..) + 16 (wave & 1) + (lane & 15), unit 16 gl + 4 (lane >> 4) + (wave >> 1).
template <int HH> struct LP { static constexpr int RS = 4 * HH + 16, US = HH + 4, NIT = HH / 32; };   // LDS row strides (gate rows / unit rows); items per thread
__device__ __forceinline__ void block_of(int bid, int ngr, int gpd, int& o, int& grp, int& step, int& dir, int& row0) {
    o = bid & 3; grp = (bid >> 2) % ngr; step = (bid >> 2) / ngr; dir = grp / gpd; row0 = 32 * (grp % gpd) + 8 * o;
}
// item q of a forward block -> (row r of the 8, unit u, owner-lane index)
template <int HH>
__device__ __forceinline__ void fwd_item(int q, int o, int grp, int t, int G, int& r, int& u, int& g, int& ow, int& lane) {
    r = q & 7;
    const int kq = (q >> 3) & 3, w2 = (q >> 5) & 1, gl = q >> 6;
    u = 8 * gl + 4 * w2 + kq; g = grp * LF<HH>::WG + gl; ow = 2 * w2 + (o >> 1); lane = 16 * kq + 8 * (o & 1) + r;
}
// hoisted gate inputs xw [B, T, 4H] (at the row's position of step t) and the keep-mask bytes -> ipx / ipm
template <int HH>
__global__ __launch_bounds__(256) void persist_lstm_pack_in_kernel(EncFwd p, pf32x4* __restrict__ ipx, unsigned* __restrict__ ipm) {
    constexpr int RS = LP<HH>::RS;
    __shared__ __attribute__((aligned(16))) float rows[8 * RS];
    __shared__ uint8_t mk[8 * HH];
    const int ngr = p.ndir * p.gpd, G = ngr * LF<HH>::WG, tid = threadIdx.x;
    int o, grp, t, dir, row0;
    block_of(blockIdx.x, ngr, p.gpd, o, grp, t, dir, row0);
    const EncFwdDir& d = p.d[dir];
    for (int q = tid; q < 8 * HH; q += 256) {
        const int r = q / HH, c = q % HH, b = row0 + r;
        pf32x4 v = {0.f, 0.f, 0.f, 0.f};
        unsigned m = 3u;
        if (b < p.B) {
            const int len = d.lengths ? d.lengths[b] : p.T;
            const int pos = (d.reverse && t < len) ? len - 1 - t : t;
            v = *reinterpret_cast<const pf32x4*>(d.xw + ((long)b * p.T + pos) * 4 * HH + 4 * c);
            m = (d.zc ? (d.zc[((long)t * p.B + b) * HH + c] ? 1u : 0u) : 1u) | (d.zh ? (d.zh[((long)t * p.B + b) * HH + c] ? 2u : 0u) : 2u);
        }
        *reinterpret_cast<pf32x4*>(rows + r * RS + 4 * c) = v;
        mk[q] = (uint8_t)m;
    }
    __syncthreads();
#pragma unroll
    for (int it = 0; it < LP<HH>::NIT; ++it) {
        int r, u, g, ow, lane;
        fwd_item<HH>(it * 256 + tid, o, grp, t, G, r, u, g, ow, lane);
        const float* w = rows + r * RS + u;
        const pf32x4 x = {w[0], w[HH], w[2 * HH], w[3 * HH]};
        const long idx = (((long)t * G + g) * 4 + ow) * 64 + lane;
        ipx[idx] = x;
        ipm[idx] = mk[r * HH + u];
    }
}
// epk -> the row-major tensors of mstts_lstm_seq_fwd_desc: out (at the row's position), h_hist / c_hist [T + 1, B, H], acts, c_raw
template <int HH>
__global__ __launch_bounds__(256) void persist_lstm_unpack_fwd_kernel(EncFwd p) {
    constexpr int RS = LP<HH>::RS, US = LP<HH>::US, NIT = LP<HH>::NIT;
    __shared__ __attribute__((aligned(16))) float rows[8 * RS];          // the gate rows, then [out | h | c | c_raw][8 rows][US]
    __shared__ int s_live[8];
    static_assert(4 * US <= RS, "the unit rows reuse the gate rows' space");
    const int ngr = p.ndir * p.gpd, G = ngr * LF<HH>::WG, tid = threadIdx.x, B = p.B;
    int o, grp, t, dir, row0;
    block_of(blockIdx.x, ngr, p.gpd, o, grp, t, dir, row0);
    const EncFwdDir& d = p.d[dir];
    pf32x4 a[NIT], s[NIT];
    float cp[NIT];
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
        int r, u, g, ow, lane;
        fwd_item<HH>(it * 256 + tid, o, grp, t, G, r, u, g, ow, lane);
        a[it] = p.epk[epk_index(t, G, g, ow, 0, lane)];
        s[it] = p.epk[epk_index(t, G, g, ow, 1, lane)];
        cp[it] = (t > 0 && d.c_raw) ? p.epk[epk_index(t - 1, G, g, ow, 1, lane)][2] : 0.f;
        if (u == 0) s_live[r] = (__float_as_uint(s[it][3]) & 4u) != 0u;
    }
    if (d.acts) {
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            int r, u, g, ow, lane;
            fwd_item<HH>(it * 256 + tid, o, grp, t, G, r, u, g, ow, lane);
            float* w = rows + r * RS + u;
            w[0] = a[it][0]; w[HH] = a[it][1]; w[2 * HH] = a[it][2]; w[3 * HH] = a[it][3];
        }
        __syncthreads();
        for (int q = tid; q < 8 * HH; q += 256) {
            const int r = q / HH, c = q % HH, b = row0 + r;
            if (b < B) *reinterpret_cast<pf32x4*>(d.acts + ((long)t * B + b) * 4 * HH + 4 * c) = *reinterpret_cast<const pf32x4*>(rows + r * RS + 4 * c);
        }
    }
    __syncthreads();
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
        int r, u, g, ow, lane;
        fwd_item<HH>(it * 256 + tid, o, grp, t, G, r, u, g, ow, lane);
        const bool live = (__float_as_uint(s[it][3]) & 4u) != 0u;
        float* w = rows + r * US + u;
        w[0] = s[it][0]; w[8 * US] = s[it][1]; w[16 * US] = s[it][2];
        w[24 * US] = live ? a[it][2] * cp[it] + a[it][0] * a[it][1] : cp[it];
    }
    __syncthreads();
    for (int q = tid; q < 8 * HH; q += 256) {           // [array 4][row 8][HH / 4 pieces]
        const int c = q % (HH / 4), r = (q / (HH / 4)) & 7, arr = q / (2 * HH), b = row0 + r;
        if (b >= B) continue;
        const pf32x4 v = *reinterpret_cast<const pf32x4*>(rows + (arr * 8 + r) * US + 4 * c);
        if (arr == 0) {
            const int len = d.lengths ? d.lengths[b] : p.T;
            const int pos = (d.reverse && s_live[r]) ? len - 1 - t : t;
            *reinterpret_cast<pf32x4*>(d.out + (long)b * d.out_sb + (long)pos * d.out_st + 4 * c) = v;
        } else if (arr == 1) *reinterpret_cast<pf32x4*>(d.h_hist + ((long)(t + 1) * B + b) * HH + 4 * c) = v;
        else if (arr == 2) *reinterpret_cast<pf32x4*>(d.c_hist + ((long)(t + 1) * B + b) * HH + 4 * c) = v;
        else if (d.c_raw) *reinterpret_cast<pf32x4*>(d.c_raw + ((long)t * B + b) * HH + 4 * c) = v;
    }
}
// upstream gradient d_out (at the row's position of step t = T - 1 - k, 0 for rows past their length) -> dop.  One block per (step, row group,
// 16-row half): a wave of the BPTT order is those 16 rows x 4 units
__global__ __launch_bounds__(256) void persist_lstm_pack_dout_kernel(EncBwd p, float* __restrict__ dop) {
    constexpr int US = EH + 4;
    __shared__ __attribute__((aligned(16))) float rows[16 * US];
    const int ngr = p.ndir * p.gpd, G2 = ngr * EBWG, tid = threadIdx.x;
    const int h = blockIdx.x & 1, grp = (blockIdx.x >> 1) % ngr, k = (blockIdx.x >> 1) / ngr, dir = grp / p.gpd, row0 = 32 * (grp % p.gpd) + 16 * h;
    const EncBwdDir& d = p.d[dir];
    const int t = p.T - 1 - k;
#pragma unroll
    for (int it = 0; it < 4; ++it) {
        const int q = it * 256 + tid, r = q >> 6, c = q & 63, b = row0 + r;
        pf32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (b < p.B) {
            const int len = d.lengths ? d.lengths[b] : p.T;
            if (t < len) v = *reinterpret_cast<const pf32x4*>(d.d_out + (long)b * d.dout_sb + (long)(d.reverse ? len - 1 - t : t) * d.dout_st + 4 * c);
        }
        *reinterpret_cast<pf32x4*>(rows + r * US + 4 * c) = v;
    }
    __syncthreads();
#pragma unroll
    for (int it = 0; it < 16; ++it) {
        const int q = it * 256 + tid, lane = q & 63, wq = (q >> 6) & 3, g = q >> 8;
        dop[(((long)k * G2 + grp * EBWG + g) * 8 + 2 * wq + h) * 64 + lane] = rows[(lane & 15) * US + 16 * g + 4 * (lane >> 4) + wq];
    }
}
// dpk -> dgates_step [T, B, 4H] (processing order) and dgates_pos [B, T, 4H] (position order).  One block per (step, row group, 8 rows): the
// owner lanes' 16-byte pieces (128 contiguous bytes per 8 rows in dpk) are turned in LDS into whole 4 KB gate rows, written as 16-byte pieces
constexpr int UB_RS = 4 * EH + 8;                   // LDS row stride in floats (16-byte aligned rows, the 8 rows of a piece 8 banks apart)
__global__ __launch_bounds__(256) void persist_lstm_unpack_bwd_kernel(EncBwd p) {
    __shared__ __attribute__((aligned(16))) float rows[8 * UB_RS];
    const int ngr = p.ndir * p.gpd, G2 = ngr * EBWG;
    const int o = blockIdx.x & 3, grp = (blockIdx.x >> 2) % ngr, k = (blockIdx.x >> 2) / ngr;      // o = (16-row half, 8-row octet)
    const int dir = grp / p.gpd, row0 = 32 * (grp % p.gpd) + 8 * o;
    const int tid = threadIdx.x;
#pragma unroll
    for (int it = 0; it < 8; ++it) {
        const int q = it * 256 + tid, r = q & 7, kq = (q >> 3) & 3, wq = (q >> 5) & 3, g = q >> 7;
        const long idx = (((long)k * G2 + grp * EBWG + g) * 8 + 2 * wq + (o >> 1)) * 64 + 16 * kq + 8 * (o & 1) + r;    // bwd_owner_of, inverted
        const pf32x4 dg = p.dpk[idx];
        float* w = rows + r * UB_RS + 16 * g + 4 * kq + wq;
        w[0] = dg[0]; w[EH] = dg[1]; w[2 * EH] = dg[2]; w[3 * EH] = dg[3];
    }
    __syncthreads();
    const EncBwdDir& d = p.d[dir];
    const int t = p.T - 1 - k;
#pragma unroll
    for (int it = 0; it < 8; ++it) {
        const int b = row0 + it;
        if (b >= p.B) break;
        const int len = d.lengths ? d.lengths[b] : p.T;
        const int pos = d.reverse ? (t < len ? len - 1 - t : t) : t;
        const pf32x4 v = *reinterpret_cast<const pf32x4*>(rows + it * UB_RS + 4 * tid);
        *reinterpret_cast<pf32x4*>(d.dgates_step + ((long)t * p.B + b) * 4 * EH + 4 * tid) = v;
        *reinterpret_cast<pf32x4*>(d.dgates_pos + ((long)b * p.T + pos) * 4 * EH + 4 * tid) = v;
    }
}

// recurrent kernel Wh [256, 1024] (row stride ld, gate-major columns i | j | f | o) -> the forward kernel's register order
// [workgroup 32][quarter 4][unit tile 2][k-step 16][lane 64]: A operand of k-step ks = 4 jj + e, lane (kq, m): row k = 64 quarter + 16 jj +
// 4 kq + e, column = gate (m & 3) of unit 8 gl + 4 ut + (m >> 2)
template <int HH>
__global__ void persist_lstm_pack_fwd_kernel(const float* __restrict__ wh, long ld, float* __restrict__ pk) {
    constexpr int EH = HH, KS = LF<HH>::KS;
    const int pidx = blockIdx.x * blockDim.x + threadIdx.x;
    if (pidx >= EH * 4 * EH) return;
    const int lane = pidx & 63, ks = (pidx >> 6) % KS, ut = (pidx / (64 * KS)) & 1, kq4 = (pidx / (128 * KS)) & 3, gl = pidx / (512 * KS);
    const int m = lane & 15, kq = lane >> 4, jj = ks >> 2, e = ks & 3;
    const int k = (EH / 4) * kq4 + 16 * jj + 4 * kq + e, col = (m & 3) * EH + 8 * gl + 4 * ut + (m >> 2);
    pk[pidx] = wh[(long)k * ld + col];
}
// ... -> the BPTT kernel's order [workgroup 16][quarter 4][k-step 64][lane 64]: lane (kq, m) of k-step ks = 4 jj + e holds Wh[unit 16 gl + m]
// [column of ring index c = 256 quarter + 16 jj + 4 kq + e], ring index c = 4 unit + gate
__global__ void persist_lstm_pack_bwd_kernel(const float* __restrict__ wh, long ld, float* __restrict__ pk) {
    const int pidx = blockIdx.x * blockDim.x + threadIdx.x;
    if (pidx >= EH * 4 * EH) return;
    const int lane = pidx & 63, ks = (pidx >> 6) & 63, kq4 = (pidx >> 12) & 3, gl = pidx >> 14;
    const int m = lane & 15, kq = lane >> 4, jj = ks >> 2, e = ks & 3;
    const int c = 256 * kq4 + 16 * jj + 4 * kq + e;
    pk[pidx] = wh[(long)(16 * gl + m) * ld + (c & 3) * EH + (c >> 2)];
}

}  // namespace mstts
using namespace mstts;

static int lstm_groups(long B) { return (int)((B + 31) / 32); }
/* 1 when the persistent LSTM launches cover `ndir` sequences (1, or the 2 directions of a bidirectional layer) of B rows and H units each on
 * the current device: H == 256, at most 16 row groups of 32 rows in all (their workgroups must be co-resident) */
extern "C" int32_t mstts_persist_lstm_supported_n(int64_t B, int64_t H, int32_t ndir) {
    if (!(B >= 1 && H == EH && (ndir == 1 || ndir == 2) && ndir * lstm_groups(B) <= 16)) return 0;
    static int memo[PERSIST_MAX_DEVICES];
    return persist_device_memo(memo, [](int dev) {
        int cus = 0;
        return hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && cus >= 256;
    });
}
extern "C" int32_t mstts_persist_lstm_supported(int64_t B, int64_t H) { return B <= 32 ? mstts_persist_lstm_supported_n(B, H, 2) : 0; }
/* the FORWARD launches alone also take H == 128 (inference: the Taco1 vocoder's BiRNN) */
extern "C" int32_t mstts_persist_lstm_fwd_supported_n(int64_t B, int64_t H, int32_t ndir) {
    if (H == EH) return mstts_persist_lstm_supported_n(B, H, ndir);
    return H == 128 ? mstts_persist_lstm_supported_n(B, EH, ndir) : 0;
}
extern "C" int mstts_persist_lstm_pack_fwd(const float* wh, int64_t wh_ld, int64_t H, float* fwd_pk, mstts_stream_t s) {
    MSTTS_REQUIRE(wh && fwd_pk && (H == 256 || H == 128) && wh_ld >= 4 * H, MSTTS_ERR_SHAPE, "persist_lstm_pack_fwd: null pointer, H not 256 / 128, or row stride below 4 H");
    if (H == 256) hipLaunchKernelGGL(persist_lstm_pack_fwd_kernel<256>, dim3(256 * 4 * 256 / 256), dim3(256), 0, (hipStream_t)s, wh, (long)wh_ld, fwd_pk);
    else hipLaunchKernelGGL(persist_lstm_pack_fwd_kernel<128>, dim3(128 * 4 * 128 / 256), dim3(256), 0, (hipStream_t)s, wh, (long)wh_ld, fwd_pk);
    MSTTS_CHECK_LAUNCH("persist_lstm_pack_fwd");
    return MSTTS_OK;
}
extern "C" int64_t mstts_persist_lstm_pack_floats(void) { return (int64_t)EH * 4 * EH; }
/* ring bytes, packed forward inputs + history floats (float4 ipx | uint32 ipm | 2 float4 epk per owner lane and step) and packed BPTT input /
 * output floats (float dop | float4 dpk) for ndir sequences of B rows and T steps */
extern "C" int64_t mstts_persist_lstm_ws_bytes_n(int64_t B, int32_t ndir) { return (int64_t)ndir * lstm_groups(B) * PRING * (EB_SLOT > EF_SLOT ? EB_SLOT : EF_SLOT) * 4; }
extern "C" int64_t mstts_persist_lstm_hist_floats_n(int64_t T, int64_t B, int32_t ndir) { return T * ndir * lstm_groups(B) * EP_GROUP * (4 + 1 + 8); }
extern "C" int64_t mstts_persist_lstm_bwd_floats_n(int64_t T, int64_t B, int32_t ndir) { return T * ndir * lstm_groups(B) * EP_GROUP * (1 + 4); }
/* the bidirectional pair of at most 32 rows (the encoder) */
extern "C" int64_t mstts_persist_lstm_ws_bytes(void) { return mstts_persist_lstm_ws_bytes_n(32, 2); }
extern "C" int64_t mstts_persist_lstm_hist_floats(int64_t T) { return mstts_persist_lstm_hist_floats_n(T, 32, 2); }
extern "C" int64_t mstts_persist_lstm_bwd_floats(int64_t T) { return mstts_persist_lstm_bwd_floats_n(T, 32, 2); }

extern "C" int mstts_persist_lstm_pack(const float* wh, int64_t wh_ld, float* fwd_pk, float* bwd_pk, mstts_stream_t s) {
    MSTTS_REQUIRE(wh && fwd_pk && bwd_pk && wh_ld >= 4 * EH, MSTTS_ERR_SHAPE, "persist_lstm_pack: null pointer or row stride below 4 H");
    hipLaunchKernelGGL(persist_lstm_pack_fwd_kernel<EH>, dim3(EH * 4 * EH / 256), dim3(256), 0, (hipStream_t)s, wh, (long)wh_ld, fwd_pk);
    MSTTS_CHECK_LAUNCH("persist_lstm_pack_fwd");
    hipLaunchKernelGGL(persist_lstm_pack_bwd_kernel, dim3(EH * 4 * EH / 256), dim3(256), 0, (hipStream_t)s, wh, (long)wh_ld, bwd_pk);
    MSTTS_CHECK_LAUNCH("persist_lstm_pack_bwd");
    return MSTTS_OK;
}

template <int HH>
static int lstm_fwd_launch_h(const mstts_lstm_seq_fwd_desc* const* dd, const float* const* pk, int ndir, float* xch, uint32_t* ctrl, float* hist, mstts_stream_t s) {
    typedef LF<HH> L;
    const mstts_lstm_seq_fwd_desc* a = dd[0];
    MSTTS_REQUIRE(xch && ctrl && hist, MSTTS_ERR_SHAPE, "lstm_seq_fwd_persistent: null pointer");
    MSTTS_REQUIRE(aligned16(hist) && aligned16(xch), MSTTS_ERR_ALIGN, "lstm_seq_fwd_persistent: hist / xch must be 16-byte aligned");
    MSTTS_REQUIRE(a->H == HH && mstts_persist_lstm_fwd_supported_n(a->B, a->H, ndir) && a->T >= 1, MSTTS_ERR_SHAPE,
                  "lstm_seq_fwd_persistent: shape or device not supported (see mstts_persist_lstm_fwd_supported_n)");
    EncFwd p;
    memset(&p, 0, sizeof(p));
    hipStream_t hs = (hipStream_t)s;
    const long BH = a->B * a->H;
    for (int k = 0; k < ndir; ++k) {
        const mstts_lstm_seq_fwd_desc* d = dd[k];
        MSTTS_REQUIRE(d && pk[k] && d->B == a->B && d->T == a->T && d->H == a->H, MSTTS_ERR_SHAPE, "lstm_seq_fwd_persistent: null descriptor / the sequences must have one shape");
        MSTTS_REQUIRE(d->xw && d->c_hist && d->h_hist && d->out && !d->residual, MSTTS_ERR_SHAPE, "lstm_seq_fwd_persistent: null pointer / residual input not covered");
        MSTTS_REQUIRE(!(d->reverse && !d->lengths), MSTTS_ERR_SHAPE, "lstm_seq_fwd_persistent: reverse needs a lengths array (pass T for every row)");
        MSTTS_REQUIRE(aligned16(d->xw) && aligned16(d->out) && d->out_sb % 4 == 0 && d->out_st % 4 == 0 && aligned16(d->c_hist) && aligned16(d->h_hist) &&
                      aligned16(d->acts) && aligned16(d->c_raw), MSTTS_ERR_ALIGN, "lstm_seq_fwd_persistent: xw / out / histories must be 16-byte aligned, out strides multiples of 4");
        MSTTS_REQUIRE(d->zoneout == a->zoneout, MSTTS_ERR_SHAPE, "lstm_seq_fwd_persistent: one zoneout rate for both directions");
        if (hipMemsetAsync(d->c_hist, 0, BH * sizeof(float), hs) != hipSuccess || hipMemsetAsync(d->h_hist, 0, BH * sizeof(float), hs) != hipSuccess)
            return set_err(MSTTS_ERR_LAUNCH, "lstm_seq_fwd_persistent: memset failed");
        EncFwdDir& e = p.d[k];
        e.xw = d->xw; e.wpk = pk[k]; e.lengths = d->lengths; e.zc = d->zc; e.zh = d->zh; e.out = d->out; e.out_sb = d->out_sb; e.out_st = d->out_st;
        e.c_hist = d->c_hist; e.h_hist = d->h_hist; e.acts = d->acts; e.c_raw = d->c_raw; e.reverse = d->reverse;
    }
    p.ndir = ndir; p.gpd = lstm_groups(a->B); p.B = (int)a->B; p.T = (int)a->T; p.keep = 1.f - a->zoneout; p.xch = xch; p.ctrl = ctrl;
    const int ngr = ndir * p.gpd;
    if (hipMemsetAsync(xch, 0xFF, (size_t)ngr * PRING * L::SLOT * 4, hs) != hipSuccess || hipMemsetAsync(ctrl, 0, 16 * sizeof(unsigned), hs) != hipSuccess)
        return set_err(MSTTS_ERR_LAUNCH, "lstm_seq_fwd_persistent: memset failed");
    const long nl = (long)p.T * ngr * L::PGROUP;
    pf32x4* ipx = reinterpret_cast<pf32x4*>(hist);
    unsigned* ipm = reinterpret_cast<unsigned*>(hist + nl * 4);
    p.ipx = ipx; p.ipm = ipm; p.epk = reinterpret_cast<pf32x4*>(hist + nl * 5);
    static_assert(L::PGROUP % 4 == 0, "epk stays 16-byte aligned behind ipx | ipm");
    hipLaunchKernelGGL(persist_lstm_pack_in_kernel<HH>, dim3((unsigned)(4 * ngr * p.T)), dim3(256), 0, hs, p, ipx, ipm);
    MSTTS_CHECK_LAUNCH("persist_lstm_pack_in");
    hipLaunchKernelGGL(persist_lstm_fwd_kernel<HH>, dim3(ngr * L::WG), dim3(ETH), 0, hs, p);
    MSTTS_CHECK_LAUNCH("persist_lstm_fwd");
    hipLaunchKernelGGL(persist_lstm_unpack_fwd_kernel<HH>, dim3((unsigned)(4 * ngr * p.T)), dim3(256), 0, hs, p);
    MSTTS_CHECK_LAUNCH("persist_lstm_unpack_fwd");
    return MSTTS_OK;
}
static int lstm_fwd_launch(const mstts_lstm_seq_fwd_desc* const* dd, const float* const* pk, int ndir, float* xch, uint32_t* ctrl, float* hist, mstts_stream_t s) {
    MSTTS_REQUIRE(dd[0], MSTTS_ERR_SHAPE, "lstm_seq_fwd_persistent: null descriptor");
    return dd[0]->H == 128 ? lstm_fwd_launch_h<128>(dd, pk, ndir, xch, ctrl, hist, s) : lstm_fwd_launch_h<256>(dd, pk, ndir, xch, ctrl, hist, s);
}

static int lstm_bwd_launch(const mstts_lstm_seq_bwd_desc* const* dd, const float* const* pk, int ndir, float* xch, uint32_t* ctrl, const float* hist, float* bws,
                           mstts_stream_t s) {
    const mstts_lstm_seq_bwd_desc* a = dd[0];
    MSTTS_REQUIRE(xch && ctrl && hist && bws, MSTTS_ERR_SHAPE, "lstm_seq_bwd_persistent: null pointer");
    MSTTS_REQUIRE(aligned16(hist) && aligned16(xch) && aligned16(bws), MSTTS_ERR_ALIGN, "lstm_seq_bwd_persistent: hist / xch / bws must be 16-byte aligned");
    MSTTS_REQUIRE(mstts_persist_lstm_supported_n(a->B, a->H, ndir) && a->T >= 1, MSTTS_ERR_SHAPE,
                  "lstm_seq_bwd_persistent: shape or device not supported (see mstts_persist_lstm_supported_n)");
    EncBwd p;
    memset(&p, 0, sizeof(p));
    hipStream_t hs = (hipStream_t)s;
    for (int k = 0; k < ndir; ++k) {
        const mstts_lstm_seq_bwd_desc* d = dd[k];
        MSTTS_REQUIRE(d && pk[k] && d->B == a->B && d->T == a->T && d->H == a->H, MSTTS_ERR_SHAPE, "lstm_seq_bwd_persistent: null descriptor / the sequences must have one shape");
        MSTTS_REQUIRE(d->d_out && d->dgates_step && d->dgates_pos, MSTTS_ERR_SHAPE, "lstm_seq_bwd_persistent: null pointer");
        MSTTS_REQUIRE(aligned16(d->dgates_step) && aligned16(d->dgates_pos) && aligned16(d->d_out) && d->dout_sb % 4 == 0 && d->dout_st % 4 == 0, MSTTS_ERR_ALIGN,
                      "lstm_seq_bwd_persistent: dgates_step / dgates_pos / d_out must be 16-byte aligned, d_out strides multiples of 4");
        MSTTS_REQUIRE(!(d->reverse && !d->lengths), MSTTS_ERR_SHAPE, "lstm_seq_bwd_persistent: reverse needs a lengths array");
        MSTTS_REQUIRE(d->zoneout == a->zoneout, MSTTS_ERR_SHAPE, "lstm_seq_bwd_persistent: one zoneout rate for both directions");
        EncBwdDir& e = p.d[k];
        e.wtpk = pk[k]; e.lengths = d->lengths; e.zc = d->zc; e.zh = d->zh; e.d_out = d->d_out; e.dout_sb = d->dout_sb; e.dout_st = d->dout_st;
        e.c_hist = d->c_hist; e.acts = d->acts; e.c_raw = d->c_raw; e.dgates_step = d->dgates_step; e.dgates_pos = d->dgates_pos; e.reverse = d->reverse;
    }
    p.ndir = ndir; p.gpd = lstm_groups(a->B); p.B = (int)a->B; p.T = (int)a->T; p.keep = 1.f - a->zoneout; p.xch = xch; p.ctrl = ctrl;
    const int ngr = ndir * p.gpd;
    if (hipMemsetAsync(xch, 0xFF, (size_t)ngr * PRING * EB_SLOT * 4, hs) != hipSuccess || hipMemsetAsync(ctrl, 0, 16 * sizeof(unsigned), hs) != hipSuccess)
        return set_err(MSTTS_ERR_LAUNCH, "lstm_seq_bwd_persistent: memset failed");
    const long nl = (long)p.T * ngr * EP_GROUP;
    p.epk = reinterpret_cast<const pf32x4*>(hist + nl * 5);
    p.dop = bws; p.dpk = reinterpret_cast<pf32x4*>(bws + nl);
    hipLaunchKernelGGL(persist_lstm_pack_dout_kernel, dim3((unsigned)(2 * ngr * p.T)), dim3(256), 0, hs, p, bws);
    MSTTS_CHECK_LAUNCH("persist_lstm_pack_dout");
    hipLaunchKernelGGL(persist_lstm_bwd_kernel, dim3(ngr * EBWG), dim3(ETH), 0, hs, p);
    MSTTS_CHECK_LAUNCH("persist_lstm_bwd");
    hipLaunchKernelGGL(persist_lstm_unpack_bwd_kernel, dim3((unsigned)(4 * ngr * p.T)), dim3(256), 0, hs, p);
    MSTTS_CHECK_LAUNCH("persist_lstm_unpack_bwd");
    return MSTTS_OK;
}

extern "C" int mstts_lstm_seq_fwd_pair_persistent(const mstts_lstm_seq_fwd_desc* a, const mstts_lstm_seq_fwd_desc* b, const float* pk_a, const float* pk_b,
                                                  float* xch, uint32_t* ctrl, float* hist, mstts_stream_t s) {
    MSTTS_REQUIRE(a && b && pk_a && pk_b, MSTTS_ERR_SHAPE, "lstm_seq_fwd_pair_persistent: null pointer");
    const mstts_lstm_seq_fwd_desc* dd[2] = {a, b};
    const float* pk[2] = {pk_a, pk_b};
    return lstm_fwd_launch(dd, pk, 2, xch, ctrl, hist, s);
}
extern "C" int mstts_lstm_seq_bwd_pair_persistent(const mstts_lstm_seq_bwd_desc* a, const mstts_lstm_seq_bwd_desc* b, const float* pkt_a, const float* pkt_b,
                                                  float* xch, uint32_t* ctrl, const float* hist, float* bws, mstts_stream_t s) {
    MSTTS_REQUIRE(a && b && pkt_a && pkt_b, MSTTS_ERR_SHAPE, "lstm_seq_bwd_pair_persistent: null pointer");
    const mstts_lstm_seq_bwd_desc* dd[2] = {a, b};
    const float* pk[2] = {pkt_a, pkt_b};
    return lstm_bwd_launch(dd, pk, 2, xch, ctrl, hist, bws, s);
}
/* one sequence (a unidirectional layer) of up to 512 rows */
extern "C" int mstts_lstm_seq_fwd_persistent(const mstts_lstm_seq_fwd_desc* a, const float* pk, float* xch, uint32_t* ctrl, float* hist, mstts_stream_t s) {
    MSTTS_REQUIRE(a && pk, MSTTS_ERR_SHAPE, "lstm_seq_fwd_persistent: null pointer");
    const mstts_lstm_seq_fwd_desc* dd[1] = {a};
    const float* pks[1] = {pk};
    return lstm_fwd_launch(dd, pks, 1, xch, ctrl, hist, s);
}
extern "C" int mstts_lstm_seq_bwd_persistent(const mstts_lstm_seq_bwd_desc* a, const float* pkt, float* xch, uint32_t* ctrl, const float* hist, float* bws, mstts_stream_t s) {
    MSTTS_REQUIRE(a && pkt, MSTTS_ERR_SHAPE, "lstm_seq_bwd_persistent: null pointer");
    const mstts_lstm_seq_bwd_desc* dd[1] = {a};
    const float* pks[1] = {pkt};
    return lstm_bwd_launch(dd, pks, 1, xch, ctrl, hist, bws, s);
}
