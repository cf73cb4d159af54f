// Native time-loop drivers: they enqueue the dependent per-step kernels of the recurrent parts of
// the graph on one HIP stream (no host round trips, capturable into a hipGraph by the caller).
//   mstts_lstm_seq_fwd/bwd        - tf.nn.dynamic_rnn over a ZoneoutLSTMCell (encoder BiLSTM, Taco1
//                                   BiRNN, speaker encoder)          [Modules.py:49-73]
//   mstts_decoder_train_fwd/bwd   - the teacher-forced attention decoder loop and its BPTT
//                                   [Modules.py:76-119,323-472 + TF AttentionWrapper]
//   mstts_decoder_infer_steps     - the free-running loop            [Modules.py:212-237]
#include "common.h"
#include "prenet_body.h"

using namespace mstts;

#define RC(call) do { int rc__ = (call); if (rc__ != MSTTS_OK) return rc__; } while (0)

// ---------------------------------------------------------------------------------------------
// Profiling probes (bench.py only): HIP events bracketing every launch of one selected kernel kind
// inside the loop drivers, on the stream the kernel is launched on.  Process-global and not
// thread-safe by design - never armed on the product path.
// ---------------------------------------------------------------------------------------------
#include <vector>
namespace {
struct Probe {
    int kind = 0;
    std::vector<hipEvent_t> ev;
    size_t used = 0;
} g_probe;
inline bool probe_on(int kind) { return g_probe.kind == kind && g_probe.used + 3 <= g_probe.ev.size(); }
inline void probe_mark(mstts_stream_t s) { hipEventRecord(g_probe.ev[g_probe.used++], (hipStream_t)s); }
}  // namespace
// three events per launch: e0 | launch | e1 | e2 - (e1 - e0) is the bracketed launch, (e2 - e1) an empty bracket on the same stream
// in the same place, i.e. what the event pair itself costs
#define PROBED(kind, s, call) do { const bool pr__ = probe_on(kind); if (pr__) probe_mark(s); RC(call); if (pr__) { probe_mark(s); probe_mark(s); } } while (0)

extern "C" int mstts_probe_begin(int32_t kind, int64_t max_launches) {
    for (hipEvent_t e : g_probe.ev) hipEventDestroy(e);
    g_probe.ev.clear();
    g_probe.used = 0;
    g_probe.kind = kind;
    if (kind == 0) return MSTTS_OK;
    g_probe.ev.resize((size_t)max_launches * 3);
    for (auto& e : g_probe.ev)
        if (hipEventCreate(&e) != hipSuccess) return set_err(MSTTS_ERR_LAUNCH, "probe: hipEventCreate failed");
    return MSTTS_OK;
}
/* after a stream synchronise: number of bracketed launches and their summed duration (ms) */
extern "C" int64_t mstts_probe_result(double* total_ms, double* empty_total_ms) {
    double tot = 0.0, emp = 0.0;
    for (size_t i = 0; i + 2 < g_probe.used; i += 3) {
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, g_probe.ev[i], g_probe.ev[i + 1]) == hipSuccess) tot += ms;
        if (hipEventElapsedTime(&ms, g_probe.ev[i + 1], g_probe.ev[i + 2]) == hipSuccess) emp += ms;
    }
    if (total_ms) *total_ms = tot;
    if (empty_total_ms) *empty_total_ms = emp;
    return (int64_t)(g_probe.used / 3);
}

static int gemm(const float* A, long lda, const float* B, long ldb, int trans_b, float* C, long ldc, long M, long N, long K,
                const float* bias, int act, int accumulate, mstts_stream_t s) {
    mstts_gemm_desc g;
    memset(&g, 0, sizeof(g));
    g.A = A; g.B = B; g.C = C; g.bias = bias;
    g.M = M; g.N = N; g.K = K; g.lda = lda; g.ldb = ldb; g.ldc = ldc;
    g.trans_a = 0; g.trans_b = trans_b; g.act = act; g.accumulate = accumulate; g.split_k = 1; g.batch = 1; g.alpha = 1.f;
    return mstts_gemm_f32(&g, s);
}

static int zero(float* p, long n, mstts_stream_t s) {
    hipError_t e = hipMemsetAsync(p, 0, n * sizeof(float), (hipStream_t)s);
    if (e != hipSuccess) return set_err(MSTTS_ERR_LAUNCH, "memset: %s", hipGetErrorString(e));
    return MSTTS_OK;
}

// one fused cell step (product + cell update, cell.hip); out_p / hn_p: packed blocks of the cells that consume m / h' next
static int cell_step(const float* Xp, const float* Wp, long K, const float* xw, long xw_ld, const float* bias,
                     const float* c_prev, const float* h_prev, long h_prev_ld, const uint8_t* zc, const uint8_t* zh, float zoneout,
                     float* out, long out_ld, float* c_next, float* h_next, long h_next_ld, float* acts, float* c_raw, long B, long H,
                     float* out_p, long out_p_K, long out_p_col0, float* hn_p, long hn_p_K, long hn_p_col0, mstts_stream_t s, int bf16 = 0) {
    mstts_cell_fwd_desc q;
    memset(&q, 0, sizeof(q));
    q.bf16 = bf16; q.out_p.bf16 = bf16; q.h_next_p.bf16 = bf16;
    q.B = B; q.H = H; q.K = K; q.Xp = Xp; q.Wp = Wp; q.xw = xw; q.xw_ld = xw_ld; q.bias = bias;
    q.c_prev = c_prev; q.h_prev = h_prev; q.h_prev_ld = h_prev_ld; q.zc = zc; q.zh = zh; q.zoneout = zoneout;
    q.out = out; q.out_ld = out_ld; q.c_next = c_next; q.h_next = h_next; q.h_next_ld = h_next_ld; q.acts = acts; q.c_raw = c_raw;
    q.out_p.base = out_p; q.out_p.K = out_p_K; q.out_p.col0 = out_p_col0;
    q.h_next_p.base = hn_p; q.h_next_p.K = hn_p_K; q.h_next_p.col0 = hn_p_col0;
    return mstts_cell_fwd(&q, s);
}

// X[M,K] . W[K,N]: skinny K-split kernel when the shape fits (parts slabs in P), else the tiled GEMM
static int xw_fwd(const float* X, long ldx, const float* W, long ldw, float* P, long M, long N, long K, int splits, int* parts,
                  mstts_stream_t s) {
    if (splits > 0 && ldx % 4 == 0 && ldw % 4 == 0 && aligned16(X) && aligned16(W)) {
        *parts = splits;
        return mstts_skinny_fwd(X, ldx, W, ldw, P, 0, M, N, K, splits, s);
    }
    *parts = 1;
    return gemm(X, ldx, W, ldw, 0, P, N, M, N, K, nullptr, 0, 0, s);
}
// dG[M,N] . W[R,N]^T   (Wp: optional packed copy of W for exactly `splits` slices)
static int xw_bwd(const float* dG, long ldg, const float* W, long ldw, float* P, long pstride, long M, long R, long N, int splits,
                  int* parts, mstts_stream_t s, const float* Wp = nullptr) {
    if (Wp && splits > 0 && R % 32 == 0 && ldg % 4 == 0 && aligned16(dG) && aligned16(Wp)) {
        *parts = splits;
        return mstts_skinny_bwd_packed(dG, ldg, Wp, P, pstride, M, R, N, splits, s);
    }
    if (splits > 0 && ldg % 4 == 0 && ldw % 4 == 0 && aligned16(dG) && aligned16(W)) {
        *parts = splits;
        return mstts_skinny_bwd(dG, ldg, W, ldw, P, pstride, M, R, N, splits, s);
    }
    *parts = 1;
    return gemm(dG, ldg, W, ldw, 1, P, R, M, R, N, nullptr, 0, 0, s);
}

// ---- bf16 recurrent products (BASELINE config 3): split counts capped by the fp32 path's, so every workspace / slab count the
// caller sized for fp32 also holds the bf16 path; the counts are baked into the packed kernels, hence exported.
static int bf_fwd_split(long N, long K, int cap) {
    if (N % 64 != 0 || K % 64 != 0 || cap < 1) return 0;
    const long strips = N / 64, units = K / 64;
    long best = 0, best_d = 1L << 40;
    for (long ks = 1; ks <= cap && ks <= units; ++ks) {
        if (units % ks != 0 || K / ks > 512) continue;
        const long dd = labs(strips * ks - 512);
        if (dd < best_d) { best_d = dd; best = ks; }
    }
    return (int)best;
}
static int bf_bwd_split(long R, long N, int cap) {
    if (R % 32 != 0 || N % 64 != 0 || cap < 1) return 0;
    const long strips = R / 32, units = N / 64;
    long best = 0, best_d = 1L << 40;
    for (long ns = 1; ns <= cap && ns <= units; ++ns) {
        if (units % ns != 0 || N / ns > 1024) continue;
        const long dd = labs(strips * ns - 512);
        if (dd < best_d) { best_d = dd; best = ns; }
    }
    return (int)best;
}
/* out = {fwd cell0, fwd cell1, fwd query, bwd cell0, bwd cell1, bwd query}; returns 1 when the bf16 decoder path can run */
extern "C" int32_t mstts_decoder_bf16_splits(int64_t H, int64_t M, int64_t A, int32_t* out) {
    const long W0 = M + H, W1 = 2 * H;
    const int c0 = mstts_skinny_fwd_splits(4 * H, W0), c1 = mstts_skinny_fwd_splits(4 * H, W1), cq = mstts_skinny_fwd_splits(A, H);
    const int d0 = mstts_skinny_bwd_splits(W0, 4 * H), d1 = mstts_skinny_bwd_splits(W1, 4 * H), dq = mstts_skinny_bwd_splits(H, A);
    int v[6] = {bf_fwd_split(4 * H, W0, c0), bf_fwd_split(4 * H, W1, c1), bf_fwd_split(A, H, cq),
                d0 > 0 ? bf_bwd_split(W0, 4 * H, d0) : 0, bf_bwd_split(W1, 4 * H, d1), bf_bwd_split(H, A, dq > 1 ? dq : 1)};
    // the d_in0 slabs are counted by the caller (mstts_decoder_train_bwd_parts): the bf16 product must write exactly as many
    if (v[3] != d0) v[3] = (d0 > 0 && (4 * H) % (64L * d0) == 0 && 4 * H / d0 <= 1024 && W0 % 32 == 0) ? d0 : 0;
    int ok = 1;
    for (int i = 0; i < 6; ++i) { if (out) out[i] = v[i]; if (v[i] < 1) ok = 0; }
    return ok;
}

static bool seq_fused_ok(const mstts_lstm_seq_fwd_desc* d);
static void seq_cell_desc(const mstts_lstm_seq_fwd_desc* d, long t, mstts_cell_fwd_desc* q);
static int seq_fused_begin(const mstts_lstm_seq_fwd_desc* d, mstts_stream_t s);

extern "C" int64_t mstts_lstm_seq_ws_floats(int64_t B, int64_t H, int32_t backward) {
    int p = backward ? mstts_skinny_bwd_splits(H, 4 * H) : mstts_skinny_fwd_splits(4 * H, H);
    if (p < 1) p = 1;
    return backward ? 4 * B * H + (int64_t)p * B * H : (int64_t)p * B * 4 * H;
}

extern "C" int mstts_lstm_seq_fwd(const mstts_lstm_seq_fwd_desc* d, mstts_stream_t s) {
    MSTTS_REQUIRE(d && d->xw && d->wh && d->c_hist && d->h_hist && d->gates_ws, MSTTS_ERR_SHAPE, "lstm_seq_fwd: null pointer");
    MSTTS_REQUIRE(!(d->reverse && !d->lengths), MSTTS_ERR_SHAPE, "lstm_seq_fwd: reverse needs a lengths array (pass T for every row)");
    const long B = d->B, T = d->T, H = d->H, BH = B * H;
    if (seq_fused_ok(d)) {
        RC(seq_fused_begin(d, s));
        for (long t = 0; t < T; ++t) {
            mstts_cell_fwd_desc q;
            seq_cell_desc(d, t, &q);
            RC(mstts_cell_fwd(&q, s));
        }
        return MSTTS_OK;
    }
    const int sp = mstts_skinny_fwd_splits(4 * H, H);
    RC(zero(d->c_hist, BH, s));
    RC(zero(d->h_hist, BH, s));
    for (long t = 0; t < T; ++t) {
        int parts = 1;
        RC(xw_fwd(d->h_hist + t * BH, H, d->wh, d->wh_ld, d->gates_ws, B, 4 * H, H, sp, &parts, s));
        mstts_lstm_point_fwd_desc p;
        memset(&p, 0, sizeof(p));
        p.B = B; p.H = H; p.gates_h = d->gates_ws; p.gates_parts = parts; p.gates_pstride = 4 * BH;
        p.xw = d->xw; p.xw_sb = T * 4 * H; p.xw_st = 4 * H;
        p.c_prev = d->c_hist + t * BH; p.h_prev = d->h_hist + t * BH;
        p.zc = d->zc ? d->zc + t * BH : nullptr; p.zh = d->zh ? d->zh + t * BH : nullptr;
        p.zoneout = d->zoneout; p.lengths = d->lengths; p.step = (int)t; p.reverse = d->reverse;
        p.residual = d->residual; p.res_sb = T * H; p.res_st = H;
        p.out = d->out; p.out_sb = d->out_sb; p.out_st = d->out_st;
        p.c_next = d->c_hist + (t + 1) * BH; p.h_next = d->h_hist + (t + 1) * BH;
        p.acts_out = d->acts ? d->acts + t * 4 * BH : nullptr;
        p.c_raw = d->c_raw ? d->c_raw + t * BH : nullptr;
        RC(mstts_lstm_point_fwd(&p, s));
    }
    return MSTTS_OK;
}

// ---- fused sequence steps: one mstts_cell_fwd launch per step (recurrent product + cell update), h carried in packed blocks
static bool seq_fused_ok(const mstts_lstm_seq_fwd_desc* d) {
    return d->wh_p && d->h_p && !d->residual && mstts_cell_fwd_supported(d->H, d->H);
}
static void seq_cell_desc(const mstts_lstm_seq_fwd_desc* d, long t, mstts_cell_fwd_desc* q) {
    const long B = d->B, T = d->T, H = d->H, BH = B * H, blk = mstts_cell_act_floats(B, H);
    memset(q, 0, sizeof(*q));
    q->B = B; q->H = H; q->K = H;
    q->Xp = d->h_p + (t & 1) * blk; q->Wp = d->wh_p;
    q->xw = d->xw; q->xw_ld = T * 4 * H; q->xw_st = 4 * H;
    q->c_prev = d->c_hist + t * BH; q->h_prev = d->h_hist + t * BH; q->h_prev_ld = H;
    q->zc = d->zc ? d->zc + t * BH : nullptr; q->zh = d->zh ? d->zh + t * BH : nullptr; q->zoneout = d->zoneout;
    q->out = d->out; q->out_ld = d->out_sb; q->out_st = d->out_st;
    q->c_next = d->c_hist + (t + 1) * BH; q->h_next = d->h_hist + (t + 1) * BH; q->h_next_ld = H;
    q->acts = d->acts ? d->acts + t * 4 * BH : nullptr; q->c_raw = d->c_raw ? d->c_raw + t * BH : nullptr;
    q->h_next_p.base = d->h_p + ((t + 1) & 1) * blk; q->h_next_p.K = H; q->h_next_p.col0 = 0;
    q->lengths = d->lengths; q->step = (int)t; q->reverse = d->reverse;
}
static int seq_fused_begin(const mstts_lstm_seq_fwd_desc* d, mstts_stream_t s) {
    MSTTS_REQUIRE(d->xw && d->c_hist && d->h_hist && d->out, MSTTS_ERR_SHAPE, "lstm_seq_fwd: null pointer");
    MSTTS_REQUIRE(!(d->reverse && !d->lengths), MSTTS_ERR_SHAPE, "lstm_seq_fwd: reverse needs a lengths array (pass T for every row)");
    RC(zero(d->c_hist, d->B * d->H, s));
    RC(zero(d->h_hist, d->B * d->H, s));
    return zero(d->h_p, 2 * mstts_cell_act_floats(d->B, d->H), s);
}

extern "C" int mstts_lstm_seq_fwd_pair(const mstts_lstm_seq_fwd_desc* a, const mstts_lstm_seq_fwd_desc* b, mstts_stream_t s) {
    MSTTS_REQUIRE(a && b, MSTTS_ERR_SHAPE, "lstm_seq_fwd_pair: null descriptor");
    if (!(a->B == b->B && a->T == b->T && a->H == b->H && seq_fused_ok(a) && seq_fused_ok(b))) {
        RC(mstts_lstm_seq_fwd(a, s));
        return mstts_lstm_seq_fwd(b, s);
    }
    RC(seq_fused_begin(a, s));
    RC(seq_fused_begin(b, s));
    for (long t = 0; t < a->T; ++t) {
        mstts_cell_fwd_desc qa, qb;
        seq_cell_desc(a, t, &qa);
        seq_cell_desc(b, t, &qb);
        RC(mstts_cell_fwd_pair(&qa, &qb, s));
    }
    return MSTTS_OK;
}

extern "C" int mstts_lstm_seq_bwd_pair(const mstts_lstm_seq_bwd_desc* a, const mstts_lstm_seq_bwd_desc* b, mstts_stream_t s) {
    MSTTS_REQUIRE(a && b, MSTTS_ERR_SHAPE, "lstm_seq_bwd_pair: null descriptor");
    const long B = a->B, T = a->T, H = a->H, BH = B * H;
    const int sp = mstts_skinny_bwd_splits(H, 4 * H);
    const bool pair_ok = a->B == b->B && a->T == b->T && a->H == b->H && sp > 0 && a->wh_ld == b->wh_ld && a->wh_ld % 4 == 0 && aligned16(a->wh) &&
                         aligned16(b->wh) && (sp == 1 || sp == 2 || sp == 4 || sp == 8) && a->dgates_step && b->dgates_step;
    if (!pair_ok) {
        RC(mstts_lstm_seq_bwd(a, s));
        return mstts_lstm_seq_bwd(b, s);
    }
    const mstts_lstm_seq_bwd_desc* dd[2] = {a, b};
    for (int k = 0; k < 2; ++k) {
        MSTTS_REQUIRE(dd[k]->wh && dd[k]->d_out && dd[k]->c_hist && dd[k]->acts && dd[k]->c_raw && dd[k]->ws, MSTTS_ERR_SHAPE, "lstm_seq_bwd_pair: null pointer");
        MSTTS_REQUIRE(!(dd[k]->reverse && !dd[k]->lengths), MSTTS_ERR_SHAPE, "lstm_seq_bwd_pair: reverse needs a lengths array");
        RC(zero(dd[k]->ws, 4 * BH, s));
    }
    int cur = 0;
    for (long t = T - 1; t >= 0; --t) {
        const int nxt = cur ^ 1;
        mstts_lstm_point_bwd_desc p[2];
        for (int k = 0; k < 2; ++k) {
            const mstts_lstm_seq_bwd_desc* d = dd[k];
            float* dc[2] = {d->ws, d->ws + BH};
            float* dh[2] = {d->ws + 2 * BH, d->ws + 3 * BH};
            float* dhg = d->ws + 4 * BH;
            memset(&p[k], 0, sizeof(p[k]));
            p[k].B = B; p[k].H = H;
            p[k].d_out = d->d_out; p[k].dout_sb = d->dout_sb; p[k].dout_st = d->dout_st;
            p[k].d_c_state = dc[cur]; p[k].d_h_state = dh[cur];
            p[k].d_h_state2 = (t == T - 1) ? nullptr : dhg; p[k].dhs2_ld = H; p[k].dhs2_parts = sp; p[k].dhs2_pstride = BH;
            p[k].acts = d->acts + t * 4 * BH; p[k].c_raw = d->c_raw + t * BH; p[k].c_prev = d->c_hist + t * BH;
            p[k].zc = d->zc ? d->zc + t * BH : nullptr; p[k].zh = d->zh ? d->zh + t * BH : nullptr;
            p[k].zoneout = d->zoneout; p[k].lengths = d->lengths; p[k].step = (int)t; p[k].reverse = d->reverse;
            p[k].dgates = d->dgates_step + t * 4 * BH;
            p[k].dgates_pos = d->dgates_pos; p[k].dgp_sb = T * 4 * H; p[k].dgp_st = 4 * H;
            p[k].d_c_prev = dc[nxt]; p[k].d_h_prev = dh[nxt];
        }
        RC(mstts_lstm_point_bwd_pair(&p[0], &p[1], s));
        RC(mstts_skinny_bwd_pair(p[0].dgates, p[1].dgates, 4 * H, a->wh, b->wh, a->wh_ld, a->ws + 4 * BH, b->ws + 4 * BH, 0, B, H, 4 * H, sp, s));
        cur = nxt;
    }
    return MSTTS_OK;
}

extern "C" int mstts_lstm_seq_bwd(const mstts_lstm_seq_bwd_desc* d, mstts_stream_t s) {
    MSTTS_REQUIRE(d && d->wh && d->d_out && d->c_hist && d->acts && d->c_raw && d->dgates_step && d->ws, MSTTS_ERR_SHAPE,
                  "lstm_seq_bwd: null pointer");
    MSTTS_REQUIRE(!(d->reverse && !d->lengths), MSTTS_ERR_SHAPE, "lstm_seq_bwd: reverse needs a lengths array");
    const long B = d->B, T = d->T, H = d->H, BH = B * H;
    float* dc[2] = {d->ws, d->ws + BH};
    float* dh[2] = {d->ws + 2 * BH, d->ws + 3 * BH};
    float* dhg = d->ws + 4 * BH;                        // [parts][B][H]: dgates . Wh^T of the later step
    const int sp = mstts_skinny_bwd_splits(H, 4 * H);
    RC(zero(d->ws, 4 * BH, s));
    int cur = 0, parts = 1;
    for (long t = T - 1; t >= 0; --t) {
        const int nxt = cur ^ 1;
        mstts_lstm_point_bwd_desc p;
        memset(&p, 0, sizeof(p));
        p.B = B; p.H = H;
        p.d_out = d->d_out; p.dout_sb = d->dout_sb; p.dout_st = d->dout_st;
        p.d_c_state = dc[cur]; p.d_h_state = dh[cur];
        p.d_h_state2 = (t == T - 1) ? nullptr : dhg; p.dhs2_ld = H; p.dhs2_parts = parts; p.dhs2_pstride = BH;
        p.acts = d->acts + t * 4 * BH; p.c_raw = d->c_raw + t * BH; p.c_prev = d->c_hist + t * BH;
        p.zc = d->zc ? d->zc + t * BH : nullptr; p.zh = d->zh ? d->zh + t * BH : nullptr;
        p.zoneout = d->zoneout; p.lengths = d->lengths; p.step = (int)t; p.reverse = d->reverse;
        p.dgates = d->dgates_step + t * 4 * BH;
        p.dgates_pos = d->dgates_pos; p.dgp_sb = T * 4 * H; p.dgp_st = 4 * H;
        p.d_c_prev = dc[nxt]; p.d_h_prev = dh[nxt];
        RC(mstts_lstm_point_bwd(&p, s));
        // recurrent part of d_h_prev = dgates . Wh^T (slabs, consumed by the next iteration)
        RC(xw_bwd(p.dgates, 4 * H, d->wh, d->wh_ld, dhg, 0, B, H, 4 * H, sp, &parts, s));
        cur = nxt;
    }
    return MSTTS_OK;
}

// ---------------------------------------------------------------------------------------------
// teacher-forced decoder loop
// ---------------------------------------------------------------------------------------------
extern "C" int mstts_decoder_train_ws_floats(int64_t B, int64_t H, int64_t M, int64_t A, int64_t* gates, int64_t* q) {
    int p0 = mstts_skinny_fwd_splits(4 * H, M + H), p1 = mstts_skinny_fwd_splits(4 * H, 2 * H), pq = mstts_skinny_fwd_splits(A, H);
    int pg = p0 > p1 ? p0 : p1;
    if (pg < 1) pg = 1;
    if (pq < 1) pq = 1;
    if (gates) *gates = (int64_t)pg * B * 4 * H;
    if (q) *q = (int64_t)pq * B * A;
    return MSTTS_OK;
}

// (Row chains - splitting the batch rows into groups whose kernel chains run on separate HIP streams - were built and measured
// in round 1: correct, but slower (eager: host-bound; graph: branches serialised).  Removed; `chains` in the descriptor is ignored.)
constexpr int MAX_CHAINS = 1;

extern "C" int mstts_decoder_train_fwd(const mstts_decoder_train_desc* d, mstts_stream_t s) {
    MSTTS_REQUIRE(d && d->xw0 && d->w0f && d->w1 && d->b1 && d->wq && d->in0 && d->in1 && d->pj && d->c0 && d->c1 &&
                  d->acts0 && d->acts1 && d->craw0 && d->craw1 && d->q_hist && d->align_hist && d->cum_hist && d->gates_ws &&
                  d->energy_ws && d->q_ws, MSTTS_ERR_SHAPE, "decoder_train_fwd: null pointer");
    const long B = d->B, S = d->S, H = d->H, M = d->lsa.M, A = d->lsa.A, T = d->lsa.T;
    MSTTS_REQUIRE(d->lsa.B == B, MSTTS_ERR_SHAPE, "decoder_train_fwd: lsa.B != B");
    const long BH = B * H, W0 = M + H, W1 = 2 * H, WP = H + M;
    const int sp0 = mstts_skinny_fwd_splits(4 * H, W0), sp1 = mstts_skinny_fwd_splits(4 * H, W1), spq = mstts_skinny_fwd_splits(A, H);
    int32_t bfs[6];
    const bool bf = d->bf_w0f_f && d->bf_w1_f && d->bf_wq_f && mstts_decoder_bf16_splits(H, M, A, bfs);
    const int pg = (sp0 > sp1 ? sp0 : sp1) > 0 ? (sp0 > sp1 ? sp0 : sp1) : 1, pq = spq > 0 ? spq : 1;
    const int chains = 1;
    const bool fused_lsa = true;         // (the time-out counter sits after the last row's granules)
    // fused cell steps need the single-launch attention step (it writes the context into cell 0's packed block)
    const bool fused_cells = fused_lsa && d->act_p && M % 4 == 0 &&
                             (bf ? (d->w0p16 && d->w1p16 && mstts_cell_fwd_bf16_supported(H, W0) && mstts_cell_fwd_bf16_supported(H, W1))
                                 : (d->w0p && d->w1p && mstts_cell_fwd_supported(H, W0) && mstts_cell_fwd_supported(H, W1)));
    const float* w0pk = bf ? (const float*)d->w0p16 : d->w0p;
    const float* w1pk = bf ? (const float*)d->w1p16 : d->w1p;
    const long p0n = mstts_cell_act_floats(B, W0), p1n = mstts_cell_act_floats(B, W1);
    if (fused_cells) RC(zero(d->act_p, 2 * (p0n + p1n), s));          // step-0 state: zero context / hidden states
    RC(zero(d->in0, B * W0, s));
    RC(zero(d->in1, B * W1, s));
    RC(zero(d->c0, BH, s));
    RC(zero(d->c1, BH, s));
    RC(zero(d->cum_hist, B * T, s));
    // query projection inside the attention launch: geometry supported, by-unit filter given, granule buffer large enough
    const bool fused_q = fused_lsa && chains == 1 && mstts_lsa_step_q_supported(T, M, H) && d->lsa.loc_kt && A == 128 &&
                         d->energy_ws_floats >= mstts_lsa_step_q_ws_bytes(B, T) / 4 && WP % 4 == 0;
    if (fused_lsa) RC(zero(d->energy_ws, fused_q ? mstts_lsa_step_q_ws_bytes(B, T) / 4 : 2 * B * T + 2, s));   // granules + time-out counter
    const long Bc = B / chains;
    mstts_stream_t cs[MAX_CHAINS] = {s};
    for (long st = 0; st < S; ++st) {
        for (int c = 0; c < chains; ++c) {
            const long b0 = c * Bc;
            mstts_stream_t q_s = cs[c];
            float* gates = d->gates_ws + (long)pg * b0 * 4 * H;         // [parts][Bc][4H]
            float* qws = d->q_ws + (long)pq * b0 * A;                   // [parts][Bc][A]
            float* energy = d->energy_ws + b0 * T;
            mstts_lsa_const lc = d->lsa;
            lc.B = Bc; lc.keys += b0 * T * A; lc.values += b0 * T * M;
            if (lc.lengths) lc.lengths += b0;
            mstts_lstm_point_fwd_desc p;
            mstts_cell_packed_dst ctx_p = {nullptr, 0, 0, 0};
            int parts = 1;
            // ---- cell 0: gates = [ctx | h0] . w0f + xw0[st]
            const float* in0 = d->in0 + (st * B + b0) * W0;
            float* in0n = d->in0 + ((st + 1) * B + b0) * W0;
            const float* in1 = d->in1 + (st * B + b0) * W1;
            float* in1w = d->in1 + (st * B + b0) * W1;
            float* in1n = d->in1 + ((st + 1) * B + b0) * W1;
            float* pj = d->pj + (st * B + b0) * WP;
            if (fused_cells) {
                // packed activation blocks, ping-pong by step parity: P0 = [ctx | h0] of cell 0, P1 = [m0 | h1] of cell 1.
                // cell 0 (step st) reads P0[st&1], writes m0 -> P1[st&1] and h0' -> P0[~st&1]; cell 1 reads P1[st&1], writes
                // h1' -> P1[~st&1]; the attention step writes ctx -> P0[~st&1].
                float* P0c = d->act_p + (st & 1) * p0n; float* P0n = d->act_p + ((st + 1) & 1) * p0n;
                float* P1c = d->act_p + 2 * p0n + (st & 1) * p1n; float* P1n = d->act_p + 2 * p0n + ((st + 1) & 1) * p1n;
                PROBED(MSTTS_PROBE_CELL0_GEMM, q_s, cell_step(P0c, w0pk, W0, d->xw0 + (st * B + b0) * 4 * H, 4 * H, nullptr,
                       d->c0 + (st * B + b0) * H, in0 + M, W0, d->zc0 ? d->zc0 + (st * B + b0) * H : nullptr, d->zh0 ? d->zh0 + (st * B + b0) * H : nullptr,
                       d->zoneout, in1w, W1, d->c0 + ((st + 1) * B + b0) * H, in0n + M, W0, d->acts0 + (st * B + b0) * 4 * H,
                       d->craw0 + (st * B + b0) * H, Bc, H, P1c, W1, 0, P0n, W0, M, q_s, bf ? 1 : 0));
                PROBED(MSTTS_PROBE_CELL1_GEMM, q_s, cell_step(P1c, w1pk, W1, nullptr, 0, d->b1,
                       d->c1 + (st * B + b0) * H, in1 + H, W1, d->zc1 ? d->zc1 + (st * B + b0) * H : nullptr, d->zh1 ? d->zh1 + (st * B + b0) * H : nullptr,
                       d->zoneout, pj, WP, d->c1 + ((st + 1) * B + b0) * H, in1n + H, W1, d->acts1 + (st * B + b0) * 4 * H,
                       d->craw1 + (st * B + b0) * H, Bc, H, nullptr, 0, 0, P1n, W1, H, q_s, bf ? 1 : 0));
                ctx_p.base = P0n; ctx_p.K = W0; ctx_p.col0 = 0; ctx_p.bf16 = bf ? 1 : 0;
            } else {
            if (bf) { parts = bfs[0]; PROBED(MSTTS_PROBE_CELL0_GEMM, q_s, mstts_skinny_fwd_bf16(in0, W0, d->bf_w0f_f, gates, 0, Bc, 4 * H, W0, bfs[0], q_s)); }
            else PROBED(MSTTS_PROBE_CELL0_GEMM, q_s, xw_fwd(in0, W0, d->w0f, 4 * H, gates, Bc, 4 * H, W0, sp0, &parts, q_s));
            memset(&p, 0, sizeof(p));
            p.B = Bc; p.H = H; p.gates_h = gates; p.gates_parts = parts; p.gates_pstride = 4 * Bc * H;
            p.xw = d->xw0 + (st * B + b0) * 4 * H; p.xw_sb = 4 * H; p.xw_st = 0;
            p.c_prev = d->c0 + (st * B + b0) * H; p.h_prev = in0 + M; p.h_prev_ld = W0;
            p.zc = d->zc0 ? d->zc0 + (st * B + b0) * H : nullptr; p.zh = d->zh0 ? d->zh0 + (st * B + b0) * H : nullptr;
            p.zoneout = d->zoneout;
            p.out = in1w; p.out_sb = W1; p.out_st = 0;
            p.c_next = d->c0 + ((st + 1) * B + b0) * H; p.h_next = in0n + M; p.h_next_ld = W0;
            p.acts_out = d->acts0 + (st * B + b0) * 4 * H; p.c_raw = d->craw0 + (st * B + b0) * H;
            RC(mstts_lstm_point_fwd(&p, q_s));
            // ---- cell 1: gates = [m0 | h1] . w1 + b1
            if (bf) { parts = bfs[1]; PROBED(MSTTS_PROBE_CELL1_GEMM, q_s, mstts_skinny_fwd_bf16(in1, W1, d->bf_w1_f, gates, 0, Bc, 4 * H, W1, bfs[1], q_s)); }
            else PROBED(MSTTS_PROBE_CELL1_GEMM, q_s, xw_fwd(in1, W1, d->w1, 4 * H, gates, Bc, 4 * H, W1, sp1, &parts, q_s));
            memset(&p, 0, sizeof(p));
            p.B = Bc; p.H = H; p.gates_h = gates; p.gates_parts = parts; p.gates_pstride = 4 * Bc * H; p.bias = d->b1;
            p.c_prev = d->c1 + (st * B + b0) * H; p.h_prev = in1 + H; p.h_prev_ld = W1;
            p.zc = d->zc1 ? d->zc1 + (st * B + b0) * H : nullptr; p.zh = d->zh1 ? d->zh1 + (st * B + b0) * H : nullptr;
            p.zoneout = d->zoneout;
            p.out = pj; p.out_sb = WP; p.out_st = 0;
            p.c_next = d->c1 + ((st + 1) * B + b0) * H; p.h_next = in1n + H; p.h_next_ld = W1;
            p.acts_out = d->acts1 + (st * B + b0) * 4 * H; p.c_raw = d->craw1 + (st * B + b0) * H;
            RC(mstts_lstm_point_fwd(&p, q_s));
            }
            // ---- query (partials summed inside the energy kernel, which also saves q) + attention
            const float* cum = d->cum_hist + (st * B + b0) * T;
            if (fused_q) {          // one launch: the eight slices of a row compute and exchange the query themselves
                PROBED(MSTTS_PROBE_LSA_ENERGY, q_s, mstts_lsa_step_fwd_q(&lc, pj, WP, d->wq, H, bf ? 1 : 0, d->q_hist + (st * B + b0) * A, cum,
                                         d->align_hist + (st * B + b0) * T, d->cum_hist + ((st + 1) * B + b0) * T, in0n, W0, pj + H, WP,
                                         fused_cells ? &ctx_p : nullptr, (unsigned long long*)d->energy_ws + b0 * T, (uint32_t)(st + 1), -1, q_s));
                continue;
            }
            if (bf) { parts = bfs[2]; RC(mstts_skinny_fwd_bf16(pj, WP, d->bf_wq_f, qws, 0, Bc, A, H, bfs[2], q_s)); }
            else RC(xw_fwd(pj, WP, d->wq, A, qws, Bc, A, H, spq, &parts, q_s));
            if (fused_lsa) {
                PROBED(MSTTS_PROBE_LSA_ENERGY, q_s, mstts_lsa_step_fwd(&lc, qws, parts, Bc * A, d->q_hist + (st * B + b0) * A, cum,
                                         d->align_hist + (st * B + b0) * T, d->cum_hist + ((st + 1) * B + b0) * T, in0n, W0, pj + H, WP,
                                         fused_cells ? &ctx_p : nullptr, (unsigned long long*)d->energy_ws + b0 * T, (uint32_t)(st + 1), q_s));
            } else {
                PROBED(MSTTS_PROBE_LSA_ENERGY, q_s, mstts_lsa_energy_fwd(&lc, qws, parts, Bc * A, d->q_hist + (st * B + b0) * A, cum, energy, q_s));
                PROBED(MSTTS_PROBE_LSA_CONTEXT, q_s, mstts_lsa_context_fwd(&lc, energy, cum, d->align_hist + (st * B + b0) * T,
                                         d->cum_hist + ((st + 1) * B + b0) * T, in0n, W0, pj + H, WP, q_s));
            }
        }
    }
    return MSTTS_OK;
}

extern "C" int32_t mstts_decoder_train_bwd_parts(int64_t H, int64_t M) {
    const int p = mstts_skinny_bwd_splits(M + H, 4 * H);
    return p > 0 ? p : 1;
}

extern "C" int64_t mstts_decoder_train_bwd_ws_floats(int64_t B, int64_t H, int64_t M, int64_t A, int64_t T, int64_t CH) {
    int p1 = mstts_skinny_bwd_splits(2 * H, 4 * H), pq = mstts_skinny_bwd_splits(H, A);
    if (p1 < 1) p1 = 1;
    if (pq < 1) pq = 1;
    return 8 * B * H + 2 * B * T + 2 * B * T * CH + B * T + (int64_t)p1 * B * 2 * H + (int64_t)pq * B * H;
}

extern "C" int mstts_decoder_train_bwd(const mstts_decoder_train_bwd_desc* bd, mstts_stream_t s) {
    MSTTS_REQUIRE(bd && bd->fwd && bd->d_pj && bd->dg0 && bd->dg1 && bd->dq_hist && bd->de_hist && bd->d_in0 && bd->ws,
                  MSTTS_ERR_SHAPE, "decoder_train_bwd: null pointer");
    const mstts_decoder_train_desc* d = bd->fwd;
    const long B = d->B, S = d->S, H = d->H, M = d->lsa.M, A = d->lsa.A, T = d->lsa.T, CH = d->lsa.CH;
    const long W0 = M + H, W1 = 2 * H, WP = H + M;
    const int sp1 = mstts_skinny_bwd_splits(W1, 4 * H), sp0 = mstts_skinny_bwd_splits(W0, 4 * H), spq = mstts_skinny_bwd_splits(H, A);
    const int np1 = sp1 > 0 ? sp1 : 1, npq = spq > 0 ? spq : 1;
    int32_t bfs[6];
    const bool bf = d->bf_w0f_b && d->bf_w1_b && d->bf_wq_b && mstts_decoder_bf16_splits(H, M, A, bfs);
    const long d_in0_slab = S * B * W0;
    const int chains = 1;
    const long Bc = B / chains, BcH = Bc * H, BcT = Bc * T;
    const long ws_per_row = 8 * H + 2 * T + 2 * T * CH + T + (long)np1 * W1 + (long)npq * H;
    // single-launch attention backward: its granules (B*ceil(T/8)+1 8-byte words) live in the d_align block (B*T floats)
    const bool fused_lsa = WP % 4 == 0 && H % 4 == 0;            // (the single-launch backward reads the forward context rows as float4)
    // query-layer data gradient inside cell 1's pointwise kernel: fp32 mode, A == 128, slab counts the lean kernel is built for
    // (bf16 mode: the kernel rounds both operands to bf16 first - the same products as the bf16 product launch it replaces)
    const int np1_eff = bf ? bfs[4] : np1;
    const bool fuse_q = d->wq_t && A == 128 && H % 128 == 0 && (np1_eff == 8 || np1_eff == 4 || np1_eff == 2 || np1_eff == 1) && B * H * 4 < (1LL << 30);
    RC(zero(bd->ws, ws_per_row * B, s));
    mstts_stream_t cs[MAX_CHAINS] = {s};
    struct ChainWs { float *dc0[2], *dh0[2], *dc1[2], *dh1[2], *G[2], *df[2], *d_align, *tmp1, *dqm; int parts0, parts1, partsq; } cw[MAX_CHAINS];
    for (int c = 0; c < chains; ++c) {
        float* w = bd->ws + ws_per_row * (c * Bc);
        ChainWs& k = cw[c];
        k.dc0[0] = w; k.dc0[1] = w + BcH; w += 2 * BcH;
        k.dh0[0] = w; k.dh0[1] = w + BcH; w += 2 * BcH;
        k.dc1[0] = w; k.dc1[1] = w + BcH; w += 2 * BcH;
        k.dh1[0] = w; k.dh1[1] = w + BcH; w += 2 * BcH;
        k.G[0] = w; k.G[1] = w + BcT; w += 2 * BcT;
        k.df[0] = w; k.df[1] = w + BcT * CH; w += 2 * BcT * CH;
        k.d_align = w; w += BcT;
        k.tmp1 = w; w += (long)np1 * Bc * W1;
        k.dqm = w; w += (long)npq * BcH;
        k.parts0 = k.parts1 = k.partsq = 1;
    }
    int cur = 0;
    for (long st = S - 1; st >= 0; --st) {
        const int nxt = cur ^ 1;
        const bool last = (st == S - 1);
        for (int c = 0; c < chains; ++c) {
            const long b0 = c * Bc;
            mstts_stream_t q_s = cs[c];
            ChainWs& k = cw[c];
            mstts_lsa_const lc = d->lsa;
            lc.B = Bc; lc.keys += b0 * T * A; lc.values += b0 * T * M;
            if (lc.lengths) lc.lengths += b0;
            float* dpj = bd->d_pj + (st * B + b0) * WP;
            const float* d_in0_next = last ? nullptr : bd->d_in0 + ((st + 1) * B + b0) * W0;
            // ---- attention backward
            if (fused_lsa) {        // d_align stays on chip
                PROBED(MSTTS_PROBE_LSA_DALIGN, q_s, mstts_lsa_step_bwd(&lc, dpj + H, WP, d_in0_next, W0, k.parts0, d_in0_slab,
                                        last ? nullptr : k.G[cur], last ? nullptr : k.df[cur], k.G[nxt], d->align_hist + (st * B + b0) * T,
                                        d->q_hist + (st * B + b0) * A, d->cum_hist + (st * B + b0) * T, d->pj + (st * B + b0) * WP + H, WP,
                                        bd->de_hist + (st * B + b0) * T, bd->dq_hist + (st * B + b0) * A, k.df[nxt], q_s));
            } else {
                PROBED(MSTTS_PROBE_LSA_DALIGN, q_s, mstts_lsa_dalign_bwd(&lc, dpj + H, WP, d_in0_next, W0, k.parts0, d_in0_slab,
                                        last ? nullptr : k.G[cur], last ? nullptr : k.df[cur], k.G[nxt], k.d_align, q_s));
                PROBED(MSTTS_PROBE_LSA_DENERGY, q_s, mstts_lsa_denergy_bwd(&lc, d->align_hist + (st * B + b0) * T, k.d_align, d->q_hist + (st * B + b0) * A,
                                         d->cum_hist + (st * B + b0) * T, bd->de_hist + (st * B + b0) * T, bd->dq_hist + (st * B + b0) * A, k.df[nxt], q_s));
            }
            // d_m1 (query path) = dq . Wq^T  -> slabs consumed by the cell-1 pointwise kernel
            if (fuse_q) {}          // folded into the cell-1 pointwise kernel below
            else if (bf) { k.partsq = bfs[5]; RC(mstts_skinny_bwd_bf16(bd->dq_hist + (st * B + b0) * A, A, d->bf_wq_b, k.dqm, 0, Bc, H, A, bfs[5], q_s)); }
            else RC(xw_bwd(bd->dq_hist + (st * B + b0) * A, A, d->wq, A, k.dqm, 0, Bc, H, A, spq, &k.partsq, q_s, d->wq_bp));
            // ---- cell 1 backward
            mstts_lstm_point_bwd_desc p;
            memset(&p, 0, sizeof(p));
            p.B = Bc; p.H = H;
            p.d_out = dpj; p.dout_sb = WP; p.dout_st = 0;
            if (fuse_q) { p.dq = bd->dq_hist + (st * B + b0) * A; p.wq_t = d->wq_t; p.A = A; p.dq_bf16 = bf ? 1 : 0; }
            else { p.d_out2 = k.dqm; p.dout2_parts = k.partsq; p.dout2_pstride = BcH; }
            p.d_c_state = k.dc1[cur]; p.d_h_state = k.dh1[cur];
            p.d_h_state2 = last ? nullptr : k.tmp1 + H; p.dhs2_ld = W1; p.dhs2_parts = k.parts1; p.dhs2_pstride = Bc * W1;
            p.acts = d->acts1 + (st * B + b0) * 4 * H; p.c_raw = d->craw1 + (st * B + b0) * H; p.c_prev = d->c1 + (st * B + b0) * H;
            p.zc = d->zc1 ? d->zc1 + (st * B + b0) * H : nullptr; p.zh = d->zh1 ? d->zh1 + (st * B + b0) * H : nullptr;
            p.zoneout = d->zoneout;
            p.dgates = bd->dg1 + (st * B + b0) * 4 * H;
            p.d_c_prev = k.dc1[nxt]; p.d_h_prev = k.dh1[nxt];
            RC(mstts_lstm_point_bwd(&p, q_s));
            // [d_m0 | d_h1 state] = dg1 . w1^T
            if (bf) { k.parts1 = bfs[4]; PROBED(MSTTS_PROBE_CELL1_DGEMM, q_s, mstts_skinny_bwd_bf16(p.dgates, 4 * H, d->bf_w1_b, k.tmp1, 0, Bc, W1, 4 * H, bfs[4], q_s)); }
            else PROBED(MSTTS_PROBE_CELL1_DGEMM, q_s, xw_bwd(p.dgates, 4 * H, d->w1, 4 * H, k.tmp1, 0, Bc, W1, 4 * H, sp1, &k.parts1, q_s, d->w1_bp));
            // ---- cell 0 backward
            memset(&p, 0, sizeof(p));
            p.B = Bc; p.H = H;
            p.d_out = k.tmp1; p.dout_sb = W1; p.dout_st = 0; p.dout_parts = k.parts1; p.dout_pstride = Bc * W1;
            p.d_c_state = k.dc0[cur]; p.d_h_state = k.dh0[cur];
            p.d_h_state2 = last ? nullptr : d_in0_next + M; p.dhs2_ld = W0; p.dhs2_parts = k.parts0; p.dhs2_pstride = d_in0_slab;
            p.acts = d->acts0 + (st * B + b0) * 4 * H; p.c_raw = d->craw0 + (st * B + b0) * H; p.c_prev = d->c0 + (st * B + b0) * H;
            p.zc = d->zc0 ? d->zc0 + (st * B + b0) * H : nullptr; p.zh = d->zh0 ? d->zh0 + (st * B + b0) * H : nullptr;
            p.zoneout = d->zoneout;
            p.dgates = bd->dg0 + (st * B + b0) * 4 * H;
            p.d_c_prev = k.dc0[nxt]; p.d_h_prev = k.dh0[nxt];
            RC(mstts_lstm_point_bwd(&p, q_s));
            // [d_ctx_{st-1} | d_h0 state] = dg0 . w0f^T   (slabs at stride S*B*W0)
            if (bf) { k.parts0 = bfs[3]; PROBED(MSTTS_PROBE_CELL0_DGEMM, q_s, mstts_skinny_bwd_bf16(p.dgates, 4 * H, d->bf_w0f_b, bd->d_in0 + (st * B + b0) * W0,
                                                                                                   d_in0_slab, Bc, W0, 4 * H, bfs[3], q_s)); }
            else PROBED(MSTTS_PROBE_CELL0_DGEMM, q_s, xw_bwd(p.dgates, 4 * H, d->w0f, 4 * H, bd->d_in0 + (st * B + b0) * W0, d_in0_slab, Bc, W0, 4 * H,
                                                            sp0, &k.parts0, q_s, d->w0f_bp));
        }
        cur = nxt;
    }
    return MSTTS_OK;
}

// ---------------------------------------------------------------------------------------------
// free-running decoder steps
// ---------------------------------------------------------------------------------------------
namespace mstts {
// Two-layer prenet of one decoder step for B <= 32 rows in ONE launch (prenet_body.h), the frame read from memory
__global__ __launch_bounds__(256) void prenet_step_kernel(const float* __restrict__ frame, int NM, const float* __restrict__ w0,
                                                          const float* __restrict__ b0, const float* __restrict__ w1, const float* __restrict__ b1,
                                                          const uint8_t* __restrict__ m0, const uint8_t* __restrict__ m1, float inv_keep,
                                                          int B, int P, float* __restrict__ out, long out_ld, PackedDst out_p) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    PnFramePlain src{frame};
    prenet_body(src, (int)blockIdx.x * PN_COLS, NM, w0, b0, w1, b1, m0, m1, inv_keep, B, P, out, out_ld, out_p, sm);
}
// [parts][B][NP] projection partial slabs + bias -> linear[B][NM], stop[B] (column NM)
__global__ void proj_finish_kernel(const float* __restrict__ P_, int parts, long pstride, const float* __restrict__ bias, int B, int NP, int NM,
                                   float* __restrict__ linear, float* __restrict__ stop) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B * (NM + 1)) return;
    const int b = i / (NM + 1), c = i % (NM + 1);
    float v = bias ? bias[c] : 0.f;
    for (int pp = 0; pp < parts; ++pp) v += P_[pp * pstride + (long)b * NP + c];
    if (c < NM) linear[(long)b * NM + c] = v; else stop[b] = v;
}
}  // namespace mstts

/* 1 when mstts_decoder_infer_steps can take its weight-streaming path (stacked cell-0 kernel w0s = [wx0 ; w0f] and the padded
 * projection kernel wp_pad must then be supplied in the descriptor). */
extern "C" int32_t mstts_decoder_infer_fast(int64_t B, int64_t H, int64_t P, int64_t M, int64_t A, int64_t n_mel) {
    const long NP = (n_mel + 1 + 3) / 4 * 4;
    if (B < 1 || B > PN_MAXB || M % 4 != 0) return 0;
    if (P % 64 != 0 || P / 64 > PN_MAXTPW || P / 16 > PN_MAXK1 || n_mel % 4 != 0 || n_mel / 4 > PN_MAXNM4) return 0;     // prenet_step_kernel tiles
    return mstts_skinny_fwd_splits(4 * H, P + M + H) > 0 && mstts_skinny_fwd_splits(4 * H, 2 * H) > 0 &&
           mstts_skinny_fwd_splits(A, H) > 0 && mstts_skinny_fwd_splits(NP, H + M) > 0;
}

static int infer_steps_fast(const mstts_decoder_infer_desc* d, int64_t step0, int64_t n, mstts_stream_t s) {
    const long B = d->B, H = d->H, P = d->P, NM = d->n_mel, M = d->lsa.M, A = d->lsa.A, T = d->lsa.T;
    const long BH = B * H, W0 = P + M + H, W1 = 2 * H, WP = H + M, BT = B * T, NP = (NM + 1 + 3) / 4 * 4;
    const int sp0 = mstts_skinny_fwd_splits(4 * H, W0), sp1 = mstts_skinny_fwd_splits(4 * H, W1), spq = mstts_skinny_fwd_splits(A, H),
              spp = mstts_skinny_fwd_splits(NP, WP);
    float* w = d->pre_ws;
    float* gates = w;       w += (long)MSTTS_MAX_PARTS * 4 * BH;
    const long gran_n = mstts_lsa_step_qp_ws_bytes(B, T) / 4;       // energy granules + counter, then the query and the frame granules
    float* gran = w;        w += gran_n;
    float* q = w;           w += (long)MSTTS_MAX_PARTS * B * A;
    float* pp = w;          w += (long)MSTTS_MAX_PARTS * B * NP;
    float* zero_frame = w;  w += B * NM;
    if (step0 == 0) {
        RC(zero(d->in0, 2 * B * W0, s));
        RC(zero(d->in1, 2 * B * W1, s));
        RC(zero(d->c0, 2 * BH, s));
        RC(zero(d->c1, 2 * BH, s));
        RC(zero(d->cum, 2 * BT, s));
        RC(zero(zero_frame, B * NM, s));
        RC(zero(gran, gran_n, s));
    }
    // query projection inside the attention launch, and the output projection too where the slice count allows
    const bool fused_q = mstts_lsa_step_q_supported(T, M, H) && d->lsa.loc_kt && A == 128 && WP % 4 == 0;
    const bool fused_qp = fused_q && d->wp_own && d->vp && mstts_lsa_step_qp_supported(T, M, H, NP);
    // ... and the next step's prenet too (the frame leaves its owners before their context phase): 3 launches per frame
    const bool fused_pre = fused_qp && mstts_lsa_step_prenet_supported(P, NM);
    const size_t pn_lds = sizeof(float) * (size_t)(PN_MAXB * (NM + 1) + PN_MAXB * (P + 1) + 4 * 32 * 17);
    // fused cell steps (cell.hip): packed kernels given and shapes covered -> 7 launches per frame instead of 9
    const bool fused = d->w0sp && d->w1p && d->act_p && mstts_cell_fwd_supported(H, W0) && mstts_cell_fwd_supported(H, W1);
    const long p0n = mstts_cell_act_floats(B, W0), p1n = mstts_cell_act_floats(B, W1);
    if (fused && step0 == 0) RC(zero(d->act_p, 2 * (p0n + p1n), s));
    for (long st = step0; st < step0 + n; ++st) {
        const int par = (int)(st & 1), nx = par ^ 1;
        const float* frame = (st == 0) ? zero_frame : d->linear + (st - 1) * B * NM;
        float* in0c = d->in0 + par * B * W0; float* in0n = d->in0 + nx * B * W0;      // rows [prenet P | ctx M | h0 H]
        float* in1c = d->in1 + par * B * W1; float* in1n = d->in1 + nx * B * W1;      // rows [m0 H | h1 H]
        float* P0c = fused ? d->act_p + par * p0n : nullptr; float* P0n = fused ? d->act_p + nx * p0n : nullptr;
        float* P1c = fused ? d->act_p + 2 * p0n + par * p1n : nullptr; float* P1n = fused ? d->act_p + 2 * p0n + nx * p1n : nullptr;
        PackedDst pre_p;
        pre_p.base = P0c; pre_p.nit = (int)(W0 / 64); pre_p.col0 = 0; pre_p.bf = 0;
        const bool pre_here = fused && fused_pre;        // the previous step's attention launch has left this step's prenet in in0c / P0c
        if (!pre_here || st == 0) {
            hipLaunchKernelGGL(prenet_step_kernel, dim3((unsigned)((P + PN_COLS - 1) / PN_COLS)), dim3(256), pn_lds, (hipStream_t)s, frame, (int)NM,
                               d->pw0, d->pb0, d->pw1, d->pb1, d->pm0 + st * B * P, d->pm1 + st * B * P, 1.f / d->prenet_keep, (int)B, (int)P, in0c, W0, pre_p);
            MSTTS_CHECK_LAUNCH("prenet_step");
        }
        mstts_lstm_point_fwd_desc p;
        int parts = 1;
        if (fused) {
            RC(cell_step(P0c, d->w0sp, W0, nullptr, 0, d->b0, d->c0 + par * BH, in0c + P + M, W0, nullptr, nullptr, d->zoneout,
                         in1c, W1, d->c0 + nx * BH, in0n + P + M, W0, nullptr, nullptr, B, H, P1c, W1, 0, P0n, W0, P + M, s));
            RC(cell_step(P1c, d->w1p, W1, nullptr, 0, d->b1, d->c1 + par * BH, in1c + H, W1, nullptr, nullptr, d->zoneout,
                         d->pj, WP, d->c1 + nx * BH, in1n + H, W1, nullptr, nullptr, B, H, nullptr, 0, 0, P1n, W1, H, s));
            mstts_cell_packed_dst ctx_p = {P0n, W0, P, 0};
            if (fused_qp) {
                mstts_lsa_prenet pn;
                memset(&pn, 0, sizeof(pn));
                const bool pre_next = fused_pre && st + 1 < d->Smax;           // (the masks hold Smax steps)
                if (pre_next) {
                    pn.w0 = d->pw0; pn.b0 = d->pb0; pn.w1 = d->pw1; pn.b1 = d->pb1; pn.m0 = d->pm0 + (st + 1) * B * P; pn.m1 = d->pm1 + (st + 1) * B * P;
                    pn.inv_keep = 1.f / d->prenet_keep; pn.P = (int32_t)P; pn.out = in0n; pn.out_ld = W0;
                    pn.out_p.base = P0n; pn.out_p.K = W0; pn.out_p.col0 = 0; pn.out_p.bf16 = 0;
                }
                RC(mstts_lsa_step_fwd_qp(&d->lsa, d->pj, WP, d->wq, H, d->wp_own, d->vp, d->bproj, NP, NM, d->linear + st * B * NM, d->stop + st * B,
                                         d->cum + par * BT, d->align_hist + st * BT, d->cum + nx * BT, in0n + P, W0, d->pj + H, WP, &ctx_p,
                                         pre_next ? &pn : nullptr, gran, (uint32_t)(st + 1), -1, s));
                continue;
            }
            if (fused_q) {
                RC(mstts_lsa_step_fwd_q(&d->lsa, d->pj, WP, d->wq, H, 0, nullptr, d->cum + par * BT, d->align_hist + st * BT, d->cum + nx * BT,
                                        in0n + P, W0, d->pj + H, WP, &ctx_p, gran, (uint32_t)(st + 1), -1, s));
            } else {
                RC(xw_fwd(d->pj, WP, d->wq, A, q, B, A, H, spq, &parts, s));
                RC(mstts_lsa_step_fwd(&d->lsa, q, parts, B * A, nullptr, d->cum + par * BT, d->align_hist + st * BT, d->cum + nx * BT,
                                      in0n + P, W0, d->pj + H, WP, &ctx_p, gran, (uint32_t)(st + 1), s));
            }
            RC(xw_fwd(d->pj, WP, d->wp_pad, NP, pp, B, NP, WP, spp, &parts, s));
            hipLaunchKernelGGL(proj_finish_kernel, dim3((unsigned)((B * (NM + 1) + 255) / 256)), dim3(256), 0, (hipStream_t)s, pp, parts, B * NP, d->bproj,
                               (int)B, (int)NP, (int)NM, d->linear + st * B * NM, d->stop + st * B);
            MSTTS_CHECK_LAUNCH("proj_finish");
            continue;
        }
        RC(xw_fwd(in0c, W0, d->w0s, 4 * H, gates, B, 4 * H, W0, sp0, &parts, s));
        memset(&p, 0, sizeof(p));
        p.B = B; p.H = H; p.gates_h = gates; p.gates_parts = parts; p.gates_pstride = 4 * BH; p.bias = d->b0;
        p.c_prev = d->c0 + par * BH; p.h_prev = in0c + P + M; p.h_prev_ld = W0; p.zoneout = d->zoneout;
        p.out = in1c; p.out_sb = W1; p.c_next = d->c0 + nx * BH; p.h_next = in0n + P + M; p.h_next_ld = W0;
        RC(mstts_lstm_point_fwd(&p, s));
        RC(xw_fwd(in1c, W1, d->w1, 4 * H, gates, B, 4 * H, W1, sp1, &parts, s));
        memset(&p, 0, sizeof(p));
        p.B = B; p.H = H; p.gates_h = gates; p.gates_parts = parts; p.gates_pstride = 4 * BH; p.bias = d->b1;
        p.c_prev = d->c1 + par * BH; p.h_prev = in1c + H; p.h_prev_ld = W1; p.zoneout = d->zoneout;
        p.out = d->pj; p.out_sb = WP; p.c_next = d->c1 + nx * BH; p.h_next = in1n + H; p.h_next_ld = W1;
        RC(mstts_lstm_point_fwd(&p, s));
        RC(xw_fwd(d->pj, WP, d->wq, A, q, B, A, H, spq, &parts, s));
        RC(mstts_lsa_step_fwd(&d->lsa, q, parts, B * A, nullptr, d->cum + par * BT, d->align_hist + st * BT, d->cum + nx * BT,
                              in0n + P, W0, d->pj + H, WP, nullptr, gran, (uint32_t)(st + 1), s));
        RC(xw_fwd(d->pj, WP, d->wp_pad, NP, pp, B, NP, WP, spp, &parts, s));
        hipLaunchKernelGGL(proj_finish_kernel, dim3((unsigned)((B * (NM + 1) + 255) / 256)), dim3(256), 0, (hipStream_t)s, pp, parts, B * NP, d->bproj,
                           (int)B, (int)NP, (int)NM, d->linear + st * B * NM, d->stop + st * B);
        MSTTS_CHECK_LAUNCH("proj_finish");
    }
    return MSTTS_OK;
}

extern "C" int64_t mstts_decoder_infer_ws_floats(int64_t B, int64_t H, int64_t P, int64_t T, int64_t A, int64_t n_mel) {
    const long np = (n_mel + 1 + 3) / 4 * 4;
    const long slow = 2 * B * P + 8 * B * H + (2 * B * T + 2) + B * A + B * n_mel;
    const long fast = (long)MSTTS_MAX_PARTS * (4 * B * H + B * A + B * np) + mstts_lsa_step_qp_ws_bytes(B, T) / 4 + B * n_mel;
    return slow > fast ? slow : fast;
}

extern "C" int mstts_decoder_infer_steps(const mstts_decoder_infer_desc* d, int64_t step0, int64_t n, mstts_stream_t s) {
    MSTTS_REQUIRE(d && d->pw0 && d->pw1 && d->wx0 && d->w0f && d->w1 && d->wq && d->wproj && d->in0 && d->in1 && d->pj &&
                  d->c0 && d->c1 && d->cum && d->pre_ws && d->linear && d->stop && d->align_hist && d->pm0 && d->pm1,
                  MSTTS_ERR_SHAPE, "decoder_infer_steps: null pointer");
    MSTTS_REQUIRE(step0 >= 0 && step0 + n <= d->Smax, MSTTS_ERR_SHAPE, "decoder_infer_steps: step range exceeds Smax");
    if (d->w0s && d->wp_pad && mstts_decoder_infer_fast(d->B, d->H, d->P, d->lsa.M, d->lsa.A, d->n_mel)) return infer_steps_fast(d, step0, n, s);
    const long B = d->B, H = d->H, P = d->P, NM = d->n_mel, M = d->lsa.M, A = d->lsa.A, T = d->lsa.T;
    const long BH = B * H, W0 = M + H, W1 = 2 * H, WP = H + M, BT = B * T;
    const bool fused_lsa = true;
    float* w = d->pre_ws;
    float* pa = w;          w += B * P;
    float* pb = w;          w += B * P;
    float* xw = w;          w += 4 * BH;
    float* gates = w;       w += 4 * BH;
    float* energy = w;      w += 2 * BT + 2;          // energies, or the 8-byte granules (+ counter) of the single-launch step
    float* q = w;           w += B * A;
    float* zero_frame = w;  w += B * NM;
    if (step0 == 0) {
        RC(zero(d->in0, B * W0, s));
        RC(zero(d->in1, B * W1, s));
        RC(zero(d->c0, BH, s));
        RC(zero(d->c1, BH, s));
        RC(zero(d->cum, BT, s));
        RC(zero(zero_frame, B * NM, s));
        if (fused_lsa) RC(zero(energy, 2 * BT + 2, s));
    }
    for (long st = step0; st < step0 + n; ++st) {
        const int par = (int)(st & 1), nx = par ^ 1;
        const float* frame = (st == 0) ? zero_frame : d->linear + (st - 1) * B * NM;
        // prenet (dropout always on, Modules.py:248-253)
        RC(gemm(frame, NM, d->pw0, P, 0, pa, P, B, P, NM, d->pb0, MSTTS_ACT_RELU, 0, s));
        RC(mstts_dropout(pa, d->pm0 + st * B * P, d->prenet_keep, pb, B * P, s));
        RC(gemm(pb, P, d->pw1, P, 0, pa, P, B, P, P, d->pb1, MSTTS_ACT_RELU, 0, s));
        RC(mstts_dropout(pa, d->pm1 + st * B * P, d->prenet_keep, pb, B * P, s));
        RC(gemm(pb, P, d->wx0, 4 * H, 0, xw, 4 * H, B, 4 * H, P, d->b0, 0, 0, s));
        mstts_lstm_point_fwd_desc p;
        // cell 0
        float* in0c = d->in0 + par * B * W0; float* in0n = d->in0 + nx * B * W0;
        float* in1c = d->in1 + par * B * W1; float* in1n = d->in1 + nx * B * W1;
        RC(gemm(in0c, W0, d->w0f, 4 * H, 0, gates, 4 * H, B, 4 * H, W0, nullptr, 0, 0, s));
        memset(&p, 0, sizeof(p));
        p.B = B; p.H = H; p.gates_h = gates; p.xw = xw; p.xw_sb = 4 * H; p.xw_st = 0;
        p.c_prev = d->c0 + par * BH; p.h_prev = in0c + M; p.h_prev_ld = W0; p.zoneout = d->zoneout;
        p.out = in1c; p.out_sb = W1; p.c_next = d->c0 + nx * BH; p.h_next = in0n + M; p.h_next_ld = W0;
        RC(mstts_lstm_point_fwd(&p, s));
        // cell 1
        RC(gemm(in1c, W1, d->w1, 4 * H, 0, gates, 4 * H, B, 4 * H, W1, nullptr, 0, 0, s));
        memset(&p, 0, sizeof(p));
        p.B = B; p.H = H; p.gates_h = gates; p.bias = d->b1;
        p.c_prev = d->c1 + par * BH; p.h_prev = in1c + H; p.h_prev_ld = W1; p.zoneout = d->zoneout;
        p.out = d->pj; p.out_sb = WP; p.c_next = d->c1 + nx * BH; p.h_next = in1n + H; p.h_next_ld = W1;
        RC(mstts_lstm_point_fwd(&p, s));
        // attention
        RC(gemm(d->pj, WP, d->wq, A, 0, q, A, B, A, H, nullptr, 0, 0, s));
        if (fused_lsa) {
            RC(mstts_lsa_step_fwd(&d->lsa, q, 1, 0, nullptr, d->cum + par * BT, d->align_hist + st * BT, d->cum + nx * BT,
                                  in0n, W0, d->pj + H, WP, nullptr, energy, (uint32_t)(st + 1), s));
        } else {
            RC(mstts_lsa_energy_fwd(&d->lsa, q, 1, 0, nullptr, d->cum + par * BT, energy, s));
            RC(mstts_lsa_context_fwd(&d->lsa, energy, d->cum + par * BT, d->align_hist + st * BT, d->cum + nx * BT,
                                     in0n, W0, d->pj + H, WP, s));
        }
        // projection: [m1 | ctx] . Wp + b -> linear (n_mel) and stop (1)
        RC(gemm(d->pj, WP, d->wproj, NM + 1, 0, d->linear + st * B * NM, NM, B, NM, WP, d->bproj, 0, 0, s));
        RC(gemm(d->pj, WP, d->wproj + NM, NM + 1, 0, d->stop + st * B, 1, B, 1, WP, d->bproj ? d->bproj + NM : nullptr, 0, 0, s));
    }
    return MSTTS_OK;
}
