/* libmstts_hip.so - C ABI of the MI355X (gfx950) Tacotron2 hot path.
 *
 * The reference (CODEJIN/multi_speaker_tts) has no FFI: its device seam is
 * tf.Session.run (MSTTS_SV.py:270-273,305-308) over graph functions in Modules.py,
 * ZoneoutLSTMCell.py, Location_Sensitive_Attention.py, Taco1_Mel_to_Spect/Modules.py,
 * Speaker_Embedding/Modules.py and Audio.py.  Each entry point below replaces one of those
 * graph functions (cited per function) or the TF library op it is built from.
 *
 * Conventions:
 *   - plain pointers + sizes; every pointer is DEVICE memory unless marked host;
 *   - activations [B,T,C] row-major (NWC), conv kernels [K,Cin,Cout], dense [in,out],
 *     LSTM kernels [in+H,4H] gate order i,j,f,o - the reference's layouts;
 *   - the caller owns every buffer (workspaces included); the library allocates no device memory and the compute entry points
 *     keep no mutable state, so they are re-entrant across streams/threads.  Two exceptions, both named where they are declared:
 *     the GEMM scheduling switches mstts_gemm_tail_split / _split3 / _split_big / _big_min_workgroups / _bf16_big / _bf16_autocut
 *     (process-global development switches for A/B runs and tests; they select WHICH kernel evaluates a contraction, never what it
 *     means; mstts_gemm_deterministic is per calling thread) and the profiling facility mstts_probe_* at the end of this header
 *     (process-global event list, armed only by bench.py, not thread-safe);
 *   - the library reads NO environment variable (tests/test_cpu_abi.py greps csrc/ for getenv): the MSTTS_GEMM_* variables of the
 *     Python binding are mapped onto the setters above by multi_speaker_tts_amd/lib.py when it loads the library;
 *   - asynchronous on the given hipStream_t (passed as void*), no implicit synchronisation;
 *   - returns 0 or a negative MSTTS_ERR_* code; mstts_last_error() gives thread-local text.
 */
#ifndef MSTTS_H_
#define MSTTS_H_
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef void* mstts_stream_t; /* hipStream_t */

enum { MSTTS_OK = 0, MSTTS_ERR_SHAPE = -1, MSTTS_ERR_DTYPE = -2, MSTTS_ERR_ALIGN = -3, MSTTS_ERR_LAUNCH = -4 };
enum { MSTTS_ACT_NONE = 0, MSTTS_ACT_RELU = 1, MSTTS_ACT_TANH = 2, MSTTS_ACT_SIGMOID = 3 };

const char* mstts_last_error(void);
/* Bumped whenever a descriptor struct or an entry point's signature changes (4: round 4; 5: round 5 - mstts_persist_desc.recurrent_bf16 in the slot of round 4's schedule selector); the Python binding checks it at load. */
int mstts_abi_version(void);
/* Diagnostic (tests of the persistent launches' co-residency handling; no reference counterpart - MSTTS_SV.py:24 is a single session on one
 * device): n_workgroups workgroups that each hold 96 KB of LDS - a whole CU as far as a persistent workgroup is concerned - for
 * `microseconds`, then add 1 to *done_count (may be null). */
int mstts_debug_park_cus(int32_t n_workgroups, int64_t microseconds, uint32_t* done_count, mstts_stream_t s);

/* ---- dense contraction (tf.matmul / tf.layers.dense / tf.layers.conv1d and their gradients) ---
 * C[M,N] = act(alpha * op(A) . op(B) + bias)   (accumulate: C += ...; split_k > 1: atomic add)
 * trans_a = 0: A(m,k) = A[m*lda + k]; 1: A(m,k) = A[k*lda + m]
 * trans_b = 0: B(k,n) = B[k*ldb + n]; 1: B(k,n) = B[n*ldb + k]
 * win_T > 0 turns A into the implicit im2col view of X[rows, win_C] (conv1d 'same', NWC):
 *   trans_a = 0: A(m,k) = X[m - win_pad + k / win_C][k % win_C], zero outside the row's
 *                length-win_T sequence (m % win_T + k / win_C - win_pad must be in [0, win_T));
 *   trans_a = 1: the same view transposed (weight gradient).   lda must equal win_C.
 *   win_dil > 1 dilates the taps (tap j reads row m + (j - win_pad) * win_dil; WaveGlow/Modules.py:267-275); 0 means 1. */
typedef struct {
    const float* A; const float* B; float* C; const float* bias;
    int64_t M, N, K;
    int64_t lda, ldb, ldc;
    int32_t trans_a, trans_b;
    int32_t win_T, win_C, win_pad;
    int32_t act, accumulate, split_k;
    int64_t batch, stride_a, stride_b, stride_c;
    float alpha;
    int32_t win_dil;
} mstts_gemm_desc;
int mstts_gemm_f32(const mstts_gemm_desc* d, mstts_stream_t s);
/* Scheduling of mstts_gemm_f32, process-wide, default on: a tile list that ends in a small fraction of a round of the 256 CUs
 * (25 632 x 512 outputs = 3 rounds + 36 tiles) has its last tiles cut along K into pieces that fill one short round, accumulated with
 * atomics onto a cleared (or, with accumulate, the existing) C; bias from piece 0; an activation (relu / tanh) is applied to the cut
 * tiles' rows by a second small kernel once the pieces have landed.  Only for batch == 1, split_k == 1.  The fp32 atomicAdd order of
 * the pieces is not fixed, so the cut tiles of a FORWARD product (e.g. the last rows of a postnet convolution) are reproducible run to
 * run only to the last bit or two; 0 switches the schedule off (every tile whole: the summation order of every output element fixed -
 * what the bit-reproducibility tests select). */
int mstts_gemm_tail_split(int32_t on);
/* 1 (default): contractions with more than 32 rows run on the bf16 matrix cores as an EXACT three-way split of every fp32 operand element and
 * six bf16 products per fp32 product (csrc/gemm_split.inc: dropped terms <= 2^-26 relative, fp32 accumulate - fp32 accuracy, 6/16 of the
 * f32-input MFMA time); 0: v_mfma_f32_32x32x2_f32 for everything (bitwise an fmaf chain).  Process-wide switch for tests and A/B runs.
 * Edge semantics of the split form (tests/test_gpu_ops.py::test_gemm_split_edge_semantics), where it differs from an IEEE fp32 contraction:
 *   - finite operands of any magnitude mix: none (same error against fp64 as the f32-input MFMA);
 *   - an operand element that is +-inf or NaN, or finite with |x| >= 3.3961e38 (it rounds to bf16 infinity; FLT_MAX is 3.4028e38): hi = +-inf,
 *     x - hi = NaN, so every output element of that row of A / column of B is NaN, where IEEE gives +-inf (or NaN).  The SET of non-finite
 *     outputs is the same; only "which non-finite value" differs.  A training step that reaches it has diverged either way;
 *   - fp32 denormal operands, and the mid / lo planes of operands below ~1e-33, may be flushed to zero by the bf16 matrix cores: an absolute
 *     error of at most 2^-8 |a| |b| per such product - visible only in an output made of such products alone.
 * Callers that need IEEE behaviour at those edges select 0. */
int mstts_gemm_split3(int32_t on);
/* 1 (default): split contractions whose output fills the chip with 256 x 256 tiles (from 160 workgroups on) run on the big-tile form of the same
 * six products (gemm_split_big_kernel: 256 x 256 x 16 per 512-thread workgroup, every wave loads, splits, stages and multiplies); 0: the
 * 128 x 128 x 32 producer / consumer kernel for all of them.  Same arithmetic, another summation order.  Process-wide (A/B runs, tests). */
int mstts_gemm_split_big(int32_t on);
/* From how many 256 x 256 workgroups on the big-tile kernels are taken: first value for mstts_gemm_f32's split kernel, second for mstts_gemm_bf16's
 * (default 160 each; a value <= 0 keeps the current setting).  Process-wide development switch (tools/gemm_split_sweep.py). */
int mstts_gemm_big_min_workgroups(int32_t f32_split, int32_t bf16);
/* Per calling thread.  1: mstts_gemm_f32 makes no K-cut the caller did not ask for with split_k (body + tail schedule and the full cut of
 * short tile lists off): every output element is one fixed-order sum, bit-reproducible run to run.  0 (default): the schedules of DESIGN 4.7,
 * whose cut tiles are summed with atomics (reproducible to the last bit or two).  The inference engines set it around their forward passes.
 * The same switch selects the fixed-order forms of the other reductions a train step contains: the column sums behind mstts_bn_train_fwd / _bwd /
 * mstts_colsum (one workgroup per 64 columns instead of atomics across row chunks), mstts_embedding_bwd (one thread per table column) and
 * mstts_lsa_param_bwd's d_keys (one workgroup per (row, tile) over all steps).  With split_k = 1 in every descriptor a train step is then
 * bit-reproducible (TrainEngine(deterministic=True)); the loss scalars of mstts_tts_loss_fwd_bwd / mstts_l2_loss_acc stay atomic sums (they feed
 * nothing).  Slow: a debugging mode. */
int mstts_gemm_deterministic(int32_t on);
/* The same contraction with both operands rounded to bf16 (round-to-nearest-even) on their way into LDS, fp32 accumulation on
 * v_mfma_f32_32x32x16_bf16, fp32 A / B / C in memory (BASELINE config 3: "bf16 with fp32 master").  Same descriptor, same modes.
 * split_k > 1 is honoured exactly, as by mstts_gemm_f32.  A call WITHOUT a cut of its own (split_k <= 1, no fused activation, batch 1) whose tile
 * list is far from a round of the chip is cut along K by the library: the pieces are added with atomics, and without `accumulate` the output's
 * M x N elements (N columns of each row, not the row pitch ldc) are cleared first - not under mstts_gemm_deterministic(1), and not at all after
 * mstts_gemm_bf16_autocut(0). */
int mstts_gemm_bf16(const mstts_gemm_desc* d, mstts_stream_t s);
/* 1 (default): the library's own K-cuts described above; 0: every contraction cut exactly as its caller asked.  Process-wide (A/B runs). */
int mstts_gemm_bf16_autocut(int32_t on);
/* 1 (default): contractions large enough to fill the chip with 256 x 256 tiles run on the big-tile kernel (csrc/gemm_bf16.hip: half the operand
 * bytes per flop of the 128 x 128 kernel); 0: the 128 x 128 kernel for everything (A/B runs, tests).  Process-wide. */
int mstts_gemm_bf16_big(int32_t on);

/* ---- randomness: Philox4x32-10 keep-masks (replaces tf.random_uniform inside
 * tf.layers.dropout, Modules.py:41-45,137-141,248-253, and ZoneoutLSTMCell.py:266-271) ---- */
int mstts_philox_keep_mask(uint8_t* out, int64_t n, uint64_t seed, uint32_t stream_id, float keep_prob, mstts_stream_t s);
/* sample-keyed form for a mask laid out [outer, B, inner] (outer = 1: batch-major): sample b draws from its own stream,
 * Philox counter (block, sample0 + b, stream_id, 0), element (o, c) = draw o*inner + c.  sample0 = global index of the first
 * local sample, so data-parallel runs of any width draw the same mask for the same sample. */
int mstts_philox_keep_mask_rows(uint8_t* out, int64_t outer, int64_t B, int64_t inner, uint64_t seed, uint32_t stream_id,
                                uint64_t sample0, float keep_prob, mstts_stream_t s);

/* ---- Encoder_Embedding (Modules.py:15-23): out[i,:] = table[token[i],:] ; bit-exact gather.
 * bwd: dtable[token[i],:] += dout[i,:] (atomic scatter-add). */
int mstts_embedding_fwd(const int32_t* token, const float* table, float* out, int64_t n, int64_t vocab, int64_t width, mstts_stream_t s);
int mstts_embedding_bwd(const int32_t* token, const float* dout, float* dtable, int64_t n, int64_t vocab, int64_t width, mstts_stream_t s);

/* ---- tf.layers.batch_normalization (+ the tf.layers.dropout that follows it) on [rows, C]
 * (Modules.py:37-45,133-141; Taco1_Mel_to_Spect/Modules.py:22-50).
 * train fwd: batch moments (biased variance), y = ((x-mean)*rstd*gamma+beta) * mask/keep,
 *            moving = moving*momentum + batch*(1-momentum).  keep_mask may be NULL.
 * ws: 2*C floats of scratch.  bwd also applies the derivative of the activation that produced
 * x (act: relu/tanh/none, evaluated from x itself) and emits the column sums the conv bias needs:
 *   dz = dBN(dy*mask/keep) * act'(x);  dgamma += ..., dbeta += ..., dbias += colsum(dz). */
int mstts_bn_train_fwd(const float* x, const float* gamma, const float* beta, float* moving_mean, float* moving_var,
                       float* y, float* save_mean, float* save_rstd, const uint8_t* keep_mask, float keep_prob,
                       float momentum, float eps, int64_t rows, int64_t C, float* ws, mstts_stream_t s);
int mstts_bn_infer_fwd(const float* x, const float* gamma, const float* beta, const float* moving_mean,
                       const float* moving_var, float* y, float eps, int64_t rows, int64_t C, mstts_stream_t s);
int mstts_bn_train_bwd(const float* dy, const float* x, const float* gamma, const float* save_mean, const float* save_rstd,
                       const uint8_t* keep_mask, float keep_prob, int32_t act, float* dz, float* dgamma, float* dbeta,
                       float* dbias, int64_t rows, int64_t C, float* ws, mstts_stream_t s);

/* ---- small fused elementwise pieces ---------------------------------------------------------
 * dropout fwd/bwd on flat arrays: y = x * mask / keep  (prenet, Modules.py:248-253)
 * relu_dropout_bwd: dx = dy * mask/keep * (y_saved > 0)  where y_saved is the dropped relu output */
int mstts_dropout(const float* x, const uint8_t* keep_mask, float keep_prob, float* y, int64_t n, mstts_stream_t s);
int mstts_relu_dropout_bwd(const float* dy, const float* y_saved, const uint8_t* keep_mask, float keep_prob, float* dx, int64_t n, mstts_stream_t s);
/* out[c] (+)= sum over rows of x[r*ld + c] */
int mstts_colsum(const float* x, int64_t rows, int64_t C, int64_t ld, float* out, int32_t accumulate, mstts_stream_t s);
/* y = a + b ; y = a*alpha ; fill */
int mstts_add(const float* a, const float* b, float* y, int64_t n, mstts_stream_t s);
int mstts_fill(float* y, float v, int64_t n, mstts_stream_t s);
/* *flag = 1 when every given persistent launch ran to its end (control words [1] == 0: no abort code, [2] == done_x: every workgroup finished),
 * else 0; a null ctrl pointer is skipped.  The device-side form of the host's check of those words, so that a data-parallel job can take the
 * MINIMUM over its ranks (dist.GradAllReduce.agree_async) without a host round trip.  No reference counterpart (MSTTS_SV.py:24: one session). */
int mstts_persist_status(const uint32_t* ctrl_a, int32_t done_a, const uint32_t* ctrl_b, int32_t done_b, int32_t* flag, mstts_stream_t s);
/* strided 2-D copy / accumulate: dst[r*ldd + c] (+)= src[r*lds + c] */
int mstts_copy2d(const float* src, int64_t lds, float* dst, int64_t ldd, int64_t rows, int64_t cols, int32_t accumulate, mstts_stream_t s);
/* tf max_pooling1d(2,1,'same') on [B,T,C]: y[t] = max(x[t], x[t+1]) (Taco1 Modules.py:28-33) */
int mstts_maxpool2_same(const float* x, float* y, int64_t B, int64_t T, int64_t C, mstts_stream_t s);
/* highway combine (Taco1 Modules.py:54-72): y = H*T + x*(1-T) with H=relu(h_pre), T=sigmoid(t_pre) */
int mstts_highway_combine(const float* h_pre, const float* t_pre, const float* x, float* y, int64_t n, mstts_stream_t s);
/* backward of the two above and the Taco1 trainer's loss (Taco1_Mel_to_Spect/Modules.py:28-33,54-72,107-108; training path of
 * Taco1_Mel_to_Spect.py:24-100): max-pool gradient goes to the first maximum of each window; l1: loss = mean |pred - target|
 * (tf.losses.absolute_difference), d_pred = sign(pred - target) / n (may be NULL) */
int mstts_maxpool2_same_bwd(const float* x, const float* dy, float* dx, int64_t B, int64_t T, int64_t C, mstts_stream_t s);
int mstts_highway_combine_bwd(const float* h_pre, const float* t_pre, const float* x, const float* dy, float* dh_pre, float* dt_pre,
                              float* dx, int64_t n, mstts_stream_t s);
int mstts_l1_loss_fwd_bwd(const float* pred, const float* target, int64_t n, float* loss, float* d_pred, mstts_stream_t s);

/* ---- ZoneoutLSTMCell (ZoneoutLSTMCell.py:188-271), one time step for all rows -----------------
 * gates_pre = gates_h[B,4H] (+ xw row) (+ bias);  i,j,f,o = split;  c = sig(f+1)*c_prev + sig(i)*tanh(j);
 * m = sig(o)*tanh(c);  c' = (1-z)*zc*(c-c_prev)+c_prev;  h' = (1-z)*zh*(m-h_prev)+h_prev  (zc/zh NULL
 * at inference).  dynamic_rnn masking: rows with step >= lengths[b] keep their state and emit 0.
 * reverse: the row's sequence position is lengths[b]-1-step (bidirectional backward direction).
 * Row b reads xw at xw + b*xw_sb + pos*xw_st, writes its output m (+ residual) at
 * out + b*out_sb + pos*out_st.  acts_out[B,4H] (sig i, tanh j, sig f, sig o) and c_raw[B,H] are the
 * BPTT saves (NULL to skip). */
typedef struct {
    int64_t B, H;
    const float* gates_h;            /* [B,4H] recurrent (and per-step input) product, no bias */
    int32_t gates_parts; int64_t gates_pstride;   /* gates_h = sum of `gates_parts` slabs (0/1 = single) */
    const float* xw; int64_t xw_sb, xw_st;   /* hoisted input product incl. bias, or NULL */
    const float* bias;               /* [4H] or NULL (when already folded into xw) */
    const float* c_prev; const float* h_prev; int64_t h_prev_ld;   /* h_prev row stride (0 -> H) */
    const uint8_t* zc; const uint8_t* zh;     /* [B,H] keep masks or NULL */
    float zoneout;
    const int32_t* lengths;          /* [B] or NULL */
    int32_t step, reverse;
    const float* residual; int64_t res_sb, res_st;  /* ResidualWrapper input or NULL */
    float* out; int64_t out_sb, out_st;
    float* c_next; float* h_next; int64_t h_next_ld;   /* [B,H]; h_next row stride (0 -> H) */
    float* acts_out; float* c_raw;   /* saves or NULL */
} mstts_lstm_point_fwd_desc;
int mstts_lstm_point_fwd(const mstts_lstm_point_fwd_desc* d, mstts_stream_t s);

typedef struct {
    int64_t B, H;
    const float* d_out; int64_t dout_sb, dout_st;   /* grad wrt the cell output m (row b at pos) or NULL */
    int32_t dout_parts; int64_t dout_pstride;       /* d_out = sum of slabs (0/1 = single) */
    const float* d_out2;             /* second [B,H] addend to the output grad or NULL */
    int32_t dout2_parts; int64_t dout2_pstride;     /* ... itself a sum of slabs (0/1 = single) */
    const float* d_c_state; const float* d_h_state; /* [B,H] grads wrt c', h' from step+1 */
    const float* d_h_state2; int64_t dhs2_ld;       /* optional second addend of d_h_state (row stride) or NULL */
    int32_t dhs2_parts; int64_t dhs2_pstride;       /* ... itself a sum of slabs (0/1 = single) */
    const float* acts; const float* c_raw; const float* c_prev;
    const uint8_t* zc; const uint8_t* zh;
    float zoneout;
    const int32_t* lengths; int32_t step, reverse;
    float* dgates;                   /* [B,4H] step-major */
    float* dgates_pos; int64_t dgp_sb, dgp_st;     /* optional second copy at (b,pos) or NULL */
    float* d_c_prev; float* d_h_prev;               /* [B,H]; d_h_prev gets only the direct (zoneout bypass) part */
    /* optional third addend of the output grad, computed in the launch: dq[B,A] . Wq[H,A]^T - the attention query layer's data
     * gradient (q = m . Wq, Location_Sensitive_Attention.py:46) folded into the cell update.  wq_t = a derived copy of Wq laid out
     * [A/4][H][4] (mstts_transpose01(Wq, wq_t, H, A/4, 4)); A must be 128 */
    const float* dq; const float* wq_t; int64_t A;
    int32_t dq_bf16;        /* multiply bf16-rounded dq / Wq values (config 3 arithmetic) */
} mstts_lstm_point_bwd_desc;
int mstts_lstm_point_bwd(const mstts_lstm_point_bwd_desc* d, mstts_stream_t s);
/* two independent cells of the same shape in one launch (sequence form without d_out slabs / d_out2 / dq); MSTTS_ERR_SHAPE when the
 * geometry is not covered */
int mstts_lstm_point_bwd_pair(const mstts_lstm_point_bwd_desc* a, const mstts_lstm_point_bwd_desc* b, mstts_stream_t s);

/* ---- fused zoneout-LSTM cell step for [B, K] x [K, 4H] cells (ZoneoutLSTMCell.py:228-271 in ONE launch: the gates
 * product and the cell update; no partial slabs).  Needs mstts_cell_fwd_supported(H, K) == 1 (H % 4 == 0, K % 64 == 0,
 * K <= 2048).  Both operands are derived copies in the kernel's lane order:
 *   Wp = the cell kernel W[K, 4H] (row stride ldw) packed by mstts_pack_cell_fwd (refresh after every optimizer step);
 *   Xp = the activation block [B, K] in the packed layout of mstts_pack_cell_act, mstts_cell_act_floats(B, K) floats.
 *        In a time loop its producers write it directly: this kernel stores packed copies of m and h' for the next cells
 *        (out_p / h_next_p) and mstts_lsa_step_fwd a packed copy of the context (ctx_p), beside the row-major history.
 * Semantics and field meanings as mstts_lstm_point_fwd_desc with the product folded in:
 *   gates = X . W + xw + bias ; out = m ; c_next / h_next = zoned state ; acts / c_raw = BPTT saves (may be NULL). */
typedef struct {
    float* base;            /* packed activation block of the consuming cell (NULL = none) */
    int64_t K, col0;        /* its reduction width and the first of the H columns this producer owns in it */
    int32_t bf16;           /* the block is in the bf16 form (values stored rounded to bf16): consumer is a bf16 cell */
} mstts_cell_packed_dst;
typedef struct {
    int64_t B, H, K;
    const float* Xp;
    const float* Wp;
    const float* xw; int64_t xw_ld;      /* [B, 4H] rows (stride xw_ld) added to the gates, or NULL */
    const float* bias;                   /* [4H] or NULL */
    const float* c_prev; const float* h_prev; int64_t h_prev_ld;
    const uint8_t* zc; const uint8_t* zh; float zoneout;
    float* out; int64_t out_ld;
    float* c_next; float* h_next; int64_t h_next_ld;
    float* acts; float* c_raw;
    mstts_cell_packed_dst out_p, h_next_p;
    /* tf.nn.dynamic_rnn form (all zero / NULL = plain cell): row b is live while step < lengths[b] (past it: output 0, state carried
     * through); reverse reads / writes position lengths[b] - 1 - step; xw and out rows are indexed (b, pos) with strides
     * (xw_ld, xw_st) and (out_ld, out_st) */
    const int32_t* lengths; int32_t step, reverse; int64_t xw_st, out_st;
    /* bf16 != 0 (BASELINE config 3): Xp / Wp are bf16 copies (mstts_pack_cell_act_bf16 / mstts_pack_cell_fwd_bf16; K % 128 == 0),
     * the product runs on v_mfma_f32_16x16x32_bf16 with fp32 accumulation; everything after it is fp32 as above.  No sequence form. */
    int32_t bf16;
} mstts_cell_fwd_desc;
int32_t mstts_cell_fwd_supported(int64_t H, int64_t K);
int mstts_pack_cell_fwd(const float* W, int64_t ldw, float* Wp, int64_t K, int64_t H, mstts_stream_t s);
int64_t mstts_cell_act_floats(int64_t B, int64_t K);
int mstts_pack_cell_act(const float* X, int64_t ldx, float* Xp, int64_t B, int64_t K, mstts_stream_t s);
int32_t mstts_cell_fwd_bf16_supported(int64_t H, int64_t K);
int mstts_pack_cell_fwd_bf16(const float* W, int64_t ldw, void* Wp16, int64_t K, int64_t H, mstts_stream_t s);    /* K*4H bf16 */
int mstts_pack_cell_act_bf16(const float* X, int64_t ldx, void* Xp16, int64_t B, int64_t K, mstts_stream_t s);    /* mstts_cell_act_floats(B,K) bf16 */
int mstts_cell_fwd(const mstts_cell_fwd_desc* d, mstts_stream_t s);
/* two independent cells of identical (B, H, K) in one launch: the two directions of a BiLSTM step (Modules.py:49-73) */
int mstts_cell_fwd_pair(const mstts_cell_fwd_desc* a, const mstts_cell_fwd_desc* b, mstts_stream_t s);

/* ---- Location_Sensitive_Attention step (Location_Sensitive_Attention.py:43-85 + TF
 * BahdanauAttention masking/softmax + AttentionWrapper context).  Two launches:
 *   energy : e[b,t] = sum_k w_k tanh(keys[b,t,k] + q[b,k] + (conv31(cum)[b,t,:] . Wd)[k] + b_k)
 *   context: a = softmax(mask(e)); cum_next = cum + a; ctx[b,:] = sum_t a[b,t] values[b,t,:]
 * T <= 512, A == 128, att conv channels == 32. */
typedef struct {
    int64_t B, T, A, M, KS, CH;      /* batch, encoder steps, attention units, memory width, conv taps, conv channels */
    const float* keys; const float* values; const int32_t* lengths;
    const float* conv_k; const float* conv_b; const float* dense_k; const float* score_w; const float* score_b;
    const float* loc_k; const float* loc_b;   /* folded location filter [KS,A], [A] (mstts_lsa_fold_location) */
    const float* loc_kt;                      /* optional: the same filter by unit, [A,36] (taps >= KS zero; mstts_lsa_filter_by_unit) - the
                                                 single-launch forward step then reads a unit's taps as 8 float4 instead of 31 words */
} mstts_lsa_const;
/* The location conv (KS taps, 1 -> CH, +bias) and the bias-free dense CH -> A that follows it are one linear map;
 * the step kernels use it folded: loc_k = conv_k . dense_k, loc_b = conv_b . dense_k.  Refresh after the variables change. */
int mstts_lsa_fold_location(const float* conv_k, const float* conv_b, const float* dense_k, float* loc_k, float* loc_b,
                            int64_t KS, int64_t CH, int64_t A, mstts_stream_t s);
/* loc_kt[a*36 + j] = loc_k[j*A + a] for j < KS, 0 for KS <= j < 36 (KS <= 31; 36 * A floats, 16-byte aligned): the by-unit copy
 * mstts_lsa_const.loc_kt points to */
int mstts_lsa_filter_by_unit(const float* loc_k, float* loc_kt, int64_t KS, int64_t A, mstts_stream_t s);
/* accumulate the gradients of conv_k / conv_b / dense_k from d_loc_k [KS,A] and d_loc_b [A] (= d_score_b) */
int mstts_lsa_unfold_location_grad(const float* conv_k, const float* conv_b, const float* dense_k, const float* d_loc_k,
                                   const float* d_loc_b, float* d_conv_k, float* d_conv_b, float* d_dense_k,
                                   int64_t KS, int64_t CH, int64_t A, mstts_stream_t s);
/* q = sum of q_parts slabs of [B,A] (slab stride q_pstride; parts 0/1 = single); when q_sum != NULL the
 * summed query is also stored there (the BPTT save). */
int mstts_lsa_energy_fwd(const mstts_lsa_const* c, const float* q, int32_t q_parts, int64_t q_pstride, float* q_sum,
                         const float* cum, float* energy, mstts_stream_t s);
/* ctx row b is written at ctx + b*ctx_ld (and, when ctx2 != NULL, also at ctx2 + b*ctx2_ld) so the
 * decoder can place it straight into the next step's GEMM input rows. */
int mstts_lsa_context_fwd(const mstts_lsa_const* c, const float* energy, const float* cum, float* align, float* cum_next,
                          float* ctx, int64_t ctx_ld, float* ctx2, int64_t ctx2_ld, mstts_stream_t s);
/* Single-launch form of the two calls above (energies -> softmax -> cumulative alignment -> context): the
 * workgroups of a row exchange their energy slices inside the launch through 8-byte {epoch,value} words in
 * `granules` (mstts_lsa_step_ws_bytes(B,T) bytes, 8-byte aligned).  Zero that buffer (hipMemsetAsync) before the
 * first step of a sequence and pass a distinct non-zero epoch per call (step + 1).  The word after the last
 * granule counts workgroups that timed out waiting and recomputed an energy themselves (0 in normal operation). */
int64_t mstts_lsa_step_ws_bytes(int64_t B, int64_t T);
int mstts_lsa_step_fwd(const mstts_lsa_const* c, const float* q, int32_t q_parts, int64_t q_pstride, float* q_sum,
                       const float* cum, float* align, float* cum_next, float* ctx, int64_t ctx_ld, float* ctx2, int64_t ctx2_ld,
                       const mstts_cell_packed_dst* ctx_p,   /* optional third copy of the context, in a fused cell's packed block (or NULL) */
                       void* granules, uint32_t epoch, mstts_stream_t s);
/* The same step with the query projection q = m1 . Wq inside the launch (one launch less per decoder step): m1 rows [B, H] with row
 * stride m1_ld, wq [H, A] row-major; q_bf16 != 0 rounds both operands to bf16 first (BASELINE config 3).  Available when
 * mstts_lsa_step_q_supported(T, M, H) (at least 8 slices: T > 112 or M > 672; H == 1024) and c->loc_kt is set.  granules =
 * mstts_lsa_step_q_ws_bytes(B, T) bytes (energy granules, time-out counter, B * A query granules), zeroed before the first step.
 * q_sum (may be NULL) receives the query.  skip_slice >= 0 = self-test form (the workgroups of that slice leave at once, the others
 * time out on its query units and energies and recompute them), -1 = normal operation. */
int32_t mstts_lsa_step_q_supported(int64_t T, int64_t M, int64_t H);
int64_t mstts_lsa_step_q_ws_bytes(int64_t B, int64_t T);
int mstts_lsa_step_fwd_q(const mstts_lsa_const* c, const float* m1, int64_t m1_ld, const float* wq, int64_t H, int32_t q_bf16, float* q_sum,
                         const float* cum, float* align, float* cum_next, float* ctx, int64_t ctx_ld, float* ctx2, int64_t ctx2_ld,
                         const mstts_cell_packed_dst* ctx_p, void* granules, uint32_t epoch, int32_t skip_slice, mstts_stream_t s);
/* ... and with the output projection [m1 | ctx] . Wp + bias out of the same launch (free-running decoder), with no exchange of its own:
 * ctx . Wp[H:, :] = sum_t a[t] vp[t] with vp [B, T, NP] = values . Wp[H:, :] (loop invariant like the keys: one GEMM per utterance), formed
 * by slice s < 8 for outputs 11 s .. 11 s + 10 from its own softmax weights; m1 . Wp[:H, :] for those outputs rides on the query
 * projection through wp_own = mstts_lsa_proj_pack(Wp[:H, :]) (mstts_lsa_proj_pack_floats() floats).  bias [NM + 1] or NULL; columns
 * 0..NM-1 -> linear [B, NM], column NM -> stop [B]; NP <= 88.  Availability and skip_slice as mstts_lsa_step_fwd_q; granules = mstts_lsa_step_qp_ws_bytes(B, T) bytes, zeroed before the first step. */
int32_t mstts_lsa_step_qp_supported(int64_t T, int64_t M, int64_t H, int64_t NP);
int64_t mstts_lsa_step_qp_ws_bytes(int64_t B, int64_t T);
int64_t mstts_lsa_proj_pack_floats(void);
int mstts_lsa_proj_pack(const float* wp, int64_t ld, int64_t H, int64_t NP, float* wp_own, mstts_stream_t s);
/* optional last stage of mstts_lsa_step_fwd_qp: the prenet of the NEXT decoder step (two dense layers, relu, dropout always on -
 * Modules.py:239-255) applied to the frame this step produces, in the same launch and without a hand-off of its own: every slice
 * knows the row's whole alignment, so each owner slice forms ALL n_mel + 1 outputs (sum_t a[t] vp[t, :] plus the m1 . Wp + b parts of
 * the other owners, which travel with the query units), writes its own 11, then computes the whole first prenet layer and 32 of
 * the 256 columns of the second.  w0 [n_mel, P], w1 [P, P] row-major and 16-byte aligned, P == 256, n_mel == 80 (NP == 84:
 * mstts_lsa_step_prenet_supported); m0 / m1 [B, P] = the NEXT step's masks; out rows [B, >= P] (stride out_ld) receive the result,
 * out_p (base NULL = none) a copy in a fused cell's packed block. */
typedef struct {
    const float* w0; const float* b0; const float* w1; const float* b1;
    const uint8_t* m0; const uint8_t* m1; float inv_keep; int32_t P;
    float* out; int64_t out_ld; mstts_cell_packed_dst out_p;
} mstts_lsa_prenet;
int32_t mstts_lsa_step_prenet_supported(int64_t P, int64_t n_mel);
int mstts_lsa_step_fwd_qp(const mstts_lsa_const* c, const float* m1, int64_t m1_ld, const float* wq, int64_t H, const float* wp_own,
                          const float* vp, const float* bias, int64_t NP, int64_t NM, float* linear, float* stop, const float* cum,
                          float* align, float* cum_next, float* ctx, int64_t ctx_ld, float* ctx2, int64_t ctx2_ld,
                          const mstts_cell_packed_dst* ctx_p, const mstts_lsa_prenet* pre /* or NULL */, void* granules, uint32_t epoch,
                          int32_t skip_slice, mstts_stream_t s);
/* backward of one step, two launches:
 *  dalign : G[t] = G_next[t] + sum_j h_next[t+pad-j][j] ; d_a[b,t] = G[b,t] + values[b,t,:] . d_ctx[b,:]
 *  denergy: d_e = a*(d_a - sum a d_a); g = d_e*w*(1-u^2); dq[b,:] += sum_t g (atomic); h[t,j] = sum_k g[t,k] loc_k[j,k]
 *           (h is [B,T,32], the filter-transpose operand of the previous step's dalign); saves d_e */
/* d_ctx row b = d_ctx[b*d_ctx_ld ..] (+ d_ctx2[b*d_ctx2_ld ..] when d_ctx2 != NULL) */
int mstts_lsa_dalign_bwd(const mstts_lsa_const* c, const float* d_ctx, int64_t d_ctx_ld, const float* d_ctx2, int64_t d_ctx2_ld,
                         int32_t d_ctx2_parts, int64_t d_ctx2_pstride, const float* G_next, const float* d_f_next, float* G, float* d_align, mstts_stream_t s);
int mstts_lsa_denergy_bwd(const mstts_lsa_const* c, const float* align, const float* d_align, const float* q, const float* cum,
                          float* d_e, float* dq, float* d_f, mstts_stream_t s);
/* Single-launch form of dalign + denergy (d_align stays on chip).  The row-wide softmax-backward scalar dot(a, d_a) needs NO exchange:
 * with d_a = G + values . d_ctx it equals dot(a, G) + ctx . d_ctx, ctx = this step's FORWARD context [B, M] (row stride ctx_fwd_ld), which
 * the caller kept from the forward pass - every workgroup forms it from data it can read directly. */
int mstts_lsa_step_bwd(const mstts_lsa_const* c, const float* d_ctx, int64_t d_ctx_ld, const float* d_ctx2, int64_t d_ctx2_ld,
                       int32_t d_ctx2_parts, int64_t d_ctx2_pstride, const float* G_next, const float* d_f_next, float* G,
                       const float* align, const float* q, const float* cum, const float* ctx_fwd, int64_t ctx_fwd_ld,
                       float* d_e, float* dq, float* d_f, mstts_stream_t s);
/* Test entry for the time-out path of the single-launch forward kernel: the same launch without the workgroups of one slice, which
 * forces the rest of each row to time out (milliseconds) and fall back to its serial recompute; the counter behind the granules
 * then reads > 0.  The skipped slice's own outputs are not written. */
int mstts_lsa_step_fwd_selftest(const mstts_lsa_const* c, const float* q, int32_t q_parts, int64_t q_pstride, float* q_sum,
                                const float* cum, float* align, float* cum_next, float* ctx, int64_t ctx_ld, void* granules,
                                uint32_t epoch, int32_t skip_slice, mstts_stream_t s);

/* post-loop parameter gradients over all S steps (recomputes tanh tiles from the saved d_e):
 * hist pointers are [S,B,*]; outputs accumulate: d_keys[B,T,A] (atomic, at most a few adds per element), d_loc_k[KS,A], d_score_w[A], d_score_b[A]
 * (d_loc_b equals d_score_b); unfold d_loc_k with mstts_lsa_unfold_location_grad.
 * ws: mstts_lsa_param_bwd_ws_floats(B, T, S) floats, 8-byte aligned: every workgroup writes its partial block of the filter / score-layer
 * gradients there and two small kernels add the blocks in fp64 in a fixed order - these gradients are sums over every (row, step, position) of
 * the batch with heavy cancellation (with fp32 atomics: 5e-3 of the gradient's maximum off at batch 32 x 801 steps, and different run to run).
 * NULL: fp32 atomics. */
int64_t mstts_lsa_param_bwd_ws_floats(int64_t B, int64_t T, int64_t S);
int mstts_lsa_param_bwd(const mstts_lsa_const* c, int64_t S, const float* q_hist, const float* cum_hist, const float* de_hist,
                        float* d_keys, float* d_loc_k, float* d_score_w, float* d_score_b, float* ws, mstts_stream_t s);

/* ---- losses (MSTTS_SV.py:127-144) forward + gradient in one pass -------------------------------
 * linear/post [B,S,n_mel] with S = L+1, mel [B,L,n_mel], stop_logit [B,S], mel_length [B].
 * scalars[0..2] = linear_loss, postnet_loss, stop_loss (accumulated atomically: zero them first).
 * d_linear/d_post [B,S,n_mel] (last step gets 0), d_stop [B,S]; grad_scale multiplies every gradient. */
int mstts_tts_loss_fwd_bwd(const float* linear, const float* post, const float* mel, const float* stop_logit,
                           const int32_t* mel_length, int64_t B, int64_t S, int64_t n_mel, int32_t use_l1,
                           float grad_scale, float* scalars, float* d_linear, float* d_post, float* d_stop, mstts_stream_t s);
/* 0.5 * sum(mask ? x^2 : 0) added atomically into *out (tf.nn.l2_loss over the regularised
 * variables, MSTTS_SV.py:145-159); mask NULL = all elements */
int mstts_l2_loss_acc(const float* x, const uint8_t* mask, int64_t n, float* out, mstts_stream_t s);

/* ---- layout glue of the decoder (Decoder_Helper teacher forcing, Modules.py:178-185,224-228) ----
 * shift_frames: frames[s,b,:] = (s == 0) ? 0 : mel[b,s-1,:]   (mel [B,L,C] -> frames [L+1,B,C])
 * unpack_proj : proj[S,B,ldp] (cols 0..C-1 = linear, col C = stop) -> linear[B,S,C], stop[B,S]
 * pack_dproj  : the inverse for gradients (pad columns zeroed)
 * speaker_tile: values[b,t,off+j] = (t < lengths[b]) ? spk[b,j] : 0      (MSTTS_SV.py:70-71 + memory mask)
 * conv_kernel_flip: wt[K-1-k][o][c] = w[k][c][o]  (data-gradient form of a conv1d kernel) */
int mstts_shift_frames(const float* mel, float* frames, int64_t B, int64_t L, int64_t C, mstts_stream_t s);
int mstts_unpack_proj(const float* proj, int64_t ldp, float* linear, float* stop, int64_t B, int64_t S, int64_t C, mstts_stream_t s);
int mstts_pack_dproj(const float* d_linear, const float* d_stop, float* d_proj, int64_t ldp, int64_t B, int64_t S, int64_t C, mstts_stream_t s);
int mstts_speaker_tile(const float* spk, const int32_t* lengths, float* values, int64_t B, int64_t T, int64_t M, int64_t off, int64_t width, mstts_stream_t s);
int mstts_conv_kernel_flip(const float* w, float* wt, int64_t K, int64_t Cin, int64_t Cout, mstts_stream_t s);
/* dst[d1][d0][:] = src[d0][d1][:]  (step-major <-> batch-major histories) */
int mstts_transpose01(const float* src, float* dst, int64_t D0, int64_t D1, int64_t C, mstts_stream_t s);
/* Speaker_Embedding Modules.Inference (Speaker_Embedding/Modules.py:127-137): x [B*samples, T, E] ->
 * out[b,:] = mean_k x[b*samples+k, T-1, :], then divided by the L2 norm of the WHOLE [B,E] tensor
 * (tf.nn.l2_normalize with axis=None, epsilon 1e-12).  Single workgroup; B*E <= 65536. */
int mstts_speaker_finalize(const float* x, float* out, int64_t B, int64_t samples, int64_t T, int64_t E, mstts_stream_t s);

/* ---- tf.train.AdamOptimizer step on a flat slab (MSTTS_SV.py:171-176; epsilon outside the bias
 * correction).  g_total = grad*grad_scale + wd[i]*p ; wd_mask (uint8, may be NULL) selects the
 * weight-regularised elements with coefficient wd. */
int mstts_adam_tf(float* p, const float* grad, float* m, float* v, const uint8_t* wd_mask, float wd, float grad_scale,
                  float lr_t, float beta1, float beta2, float eps, int64_t n, mstts_stream_t s);

/* ---- Audio.melspectrogram (Audio.py:12-13,29-32,42-48,62-96): wav[n] -> normalised mel
 * [frames, n_mel] with frames = 1 + n/hop (already [T,80] as Feeder.py:214-222 transposes it).
 * Steps: preemphasis (lfilter [1,-coef]) + centre reflect pad by n_fft/2; windowed DFT as one MFMA
 * GEMM frames[frames,win] . dft_basis[win, 2*NB] (only the `win` non-zero taps of the zero-padded
 * Hann window enter; NB = n_fft/2+1 rounded up to a multiple of 4; columns [0,NB) = w*cos,
 * [NB,2NB) = -w*sin; pad columns zero); magnitude; mel GEMM with mel_basis_t[NB, n_mel];
 * 20*log10(max(1e-5,.)) and symmetric normalisation to [-max_abs, max_abs].
 * ws floats: (n + n_fft) + frames*2*NB + frames*NB. */
int mstts_stft_mel(const float* wav, int64_t n, float preemph, const float* dft_basis, const float* mel_basis_t,
                   int32_t n_fft, int32_t hop, int32_t win, int32_t n_mel, float max_abs, float* ws, float* mel_out,
                   int64_t frames, mstts_stream_t s);
int64_t mstts_stft_mel_ws_floats(int64_t n, int32_t n_fft, int64_t frames);

/* The same transform (Audio.py:19-22,29-40,42-48,62-96) as ONE launch for nw waveforms: a workgroup per frame, real FFT in LDS
 * (n_fft a power of two in [512, 4096]; mstts_stft_fft_supported).  wav = the waveforms back to back; wav_off[nw+1] / frame_off[nw+1] =
 * DEVICE arrays of sample / frame offsets (frames of waveform w = 1 + len_w / hop, len_w > n_fft / 2); window[win] = the periodic Hann
 * window; twiddle[n_fft] = (cos, -sin)(2 pi k / n_fft) pairs, k < n_fft; mel_basis[n_mel, n_fft/2+1] row-major with mel_rng[n_mel][2] = each
 * filter's [first, last+1) non-zero bin.  mel_out [total_frames, n_mel] = Audio.melspectrogram's symmetric normalisation (flags & 1:
 * its [0, 1] normalisation, max_abs_value None), spec_out [total_frames, n_fft/2+1] = Audio.spectrogram's [0, 1] normalisation with
 * ref_level_db (flags & 2: the raw magnitudes instead); either may be NULL.  mag_in != NULL: no transform - the magnitudes are read
 * from mag_in [total_frames, n_fft/2+1] and sub[n_fft/2+1] * sub_scale (sub may be NULL) is subtracted, clipped at 0: the second pass
 * of spectral_subtract (Audio.py:45-46; sub = the per-bin sum over the waveform's frames from mstts_colsum, sub_scale = 0.1 / frames). */
int mstts_stft_fft_supported(int32_t n_fft, int32_t win);
int mstts_stft_fft(const float* wav, const int64_t* wav_off, const int64_t* frame_off, int32_t nw, float preemph, const float* window,
                   const float* twiddle, const float* mel_basis, const int32_t* mel_rng, int32_t n_fft, int32_t hop, int32_t win,
                   int32_t n_mel, float max_abs, float ref_level_db, float* mel_out, float* spec_out, int64_t total_frames,
                   const float* mag_in, const float* sub, float sub_scale, int32_t flags, mstts_stream_t s);

/* ---- skinny (M <= 32 rows per block) weight-streaming products of the recurrent steps -------------
 * fwd: P[ks][M][N] = X[M, K-slice ks] . W[K-slice ks, N]   (W row-major [K,N], ld ldw); ksplit from
 *      mstts_skinny_fwd_splits (0 = shape not supported -> use mstts_gemm_f32).
 * bwd: P[ns][M][R] = dG[M, N-slice ns] . W[R, N-slice ns]^T (W row-major [R,N]).
 * The consumer sums the partial slabs. */
int32_t mstts_skinny_fwd_splits(int64_t N, int64_t K);
int mstts_skinny_fwd(const float* X, int64_t ldx, const float* W, int64_t ldw, float* P, int64_t pstride, int64_t M, int64_t N,
                     int64_t K, int32_t ksplit, mstts_stream_t s);   /* pstride: floats between slabs (0 -> M*N) */
int32_t mstts_skinny_bwd_splits(int64_t R, int64_t N);
int mstts_skinny_bwd(const float* dG, int64_t ldg, const float* W, int64_t ldw, float* P, int64_t pstride, int64_t M, int64_t R,
                     int64_t N, int32_t nsplit, mstts_stream_t s);
/* the same product against a derived copy of W in the kernel's lane order (R % 32 == 0; made by mstts_pack_skinny_bwd for the same
 * nsplit, R*N floats; refresh after every optimizer step): every wave load is one contiguous 1 KB instead of 16 rows x 64 B */
int mstts_pack_skinny_bwd(const float* W, int64_t ldw, float* Wp, int64_t R, int64_t N, int32_t nsplit, mstts_stream_t s);
int mstts_skinny_bwd_pair(const float* dG, const float* dG2, int64_t ldg, const float* W, const float* W2, int64_t ldw, float* P, float* P2,
                          int64_t pstride, int64_t M, int64_t R, int64_t N, int32_t nsplit, mstts_stream_t s);   /* two same-shape products, one launch */
int mstts_skinny_bwd_packed(const float* dG, int64_t ldg, const float* Wp, float* P, int64_t pstride, int64_t M, int64_t R,
                            int64_t N, int32_t nsplit, mstts_stream_t s);

/* ---- WaveGlow vocoder, inference direction (WaveGlow/Modules.py:177-208,210-327,354-371; Inv1x1.py:9-41).  The contractions run
 * on mstts_gemm_f32 (win_dil for the dilated K=3 convs); these are the remaining pieces, all [rows, channels] row-major.
 *  overlap_add : Y[N,T,K,C] tap products of conv2d_transpose((1,K), stride (1,S), VALID) -> out[N,(T-1)*S+K,C] + bias
 *  gate        : z[rows,C] = tanh(a[:, :C]) * sigmoid(a[:, C:2C]),  a rows are lda floats apart
 *  res_skip    : !last: x = z + rs[:, :C], skip = rs[:, C:] (rs [rows,2C]); last: skip = rs [rows,C]; out = first ? skip : out + skip
 *  coupling_inv: a1 = (audio[:, c/2:] - b) * exp(-log_s) with log_s_b = [log_s | b] [rows,c]; out[:, c_early:] = [a0 | a1] . w_inv[c,c];
 *                out[:, :c_early] = early * sigma (the re-injected latent, Glow_Inference :362-369); out is [rows, c + c_early]
 *  philox_normal: out[i] ~ N(0, sigma^2), Box-Muller over Philox4x32-10 stream (seed, stream_id) (tf.random.normal stand-in) */
int mstts_wg_overlap_add(const float* Y, const float* bias, float* out, int64_t N, int64_t T, int64_t K, int64_t S, int64_t C, mstts_stream_t s);
int mstts_wg_gate(const float* a, int64_t lda, float* z, int64_t rows, int64_t C, mstts_stream_t s);
/* The same gate with the dilated convolution's output kept in its own buffer b [rows, 2C]: z = tanh(a[:, :C] + b[:, :C]) * sigmoid(a[:, C:2C] + b[:, C:2C]).
 * WaveGlowEngine runs that convolution as TWO reduction pieces onto a zeroed b (0 + p + q is the same float whichever piece lands first, so the
 * flow stays bit-reproducible per latent seed) and so reaches the 256 x 256-tile contraction kernel at batch 4 x 40 frames. */
int mstts_wg_gate_add(const float* a, int64_t lda, const float* b, float* z, int64_t rows, int64_t C, mstts_stream_t s);
int mstts_wg_res_skip(const float* z, const float* rs, float* x, float* out, int64_t rows, int64_t C, int32_t last, int32_t first, mstts_stream_t s);
int mstts_wg_coupling_inv(const float* audio, const float* log_s_b, const float* w_inv, const float* early, float sigma, float* out,
                          int64_t rows, int64_t c, int64_t c_early, mstts_stream_t s);
int mstts_philox_normal(float* out, int64_t n, uint64_t seed, uint32_t stream_id, float sigma, mstts_stream_t s);

/* ---- GE2E loss of the speaker-encoder trainer, forward + backward (Speaker_Embedding/Modules.py:39-98, "Softmax" method):
 * x [N = S*P, D] = last-frame outputs of the LSTM stack, speaker-major (P consecutive rows per speaker), rows ldx apart;
 * wb = {weight, bias} of the scaled cosine similarity.  out[0] = loss, out[1] = d/d weight, out[2] = d/d bias (identically 0);
 * dx [N, D] rows lddx apart = d loss / d x (through tf.nn.l2_normalize(axis=1)).  ws: mstts_ge2e_ws_floats(N, D, S) floats. */
int64_t mstts_ge2e_ws_floats(int64_t N, int64_t D, int64_t S);
int mstts_ge2e_loss_fwd_bwd(const float* x, int64_t ldx, int64_t S, int64_t P, int64_t D, const float* wb, float* out,
                            float* dx, int64_t lddx, float* ws, mstts_stream_t s);

/* ---- bf16 variants of the skinny products (BASELINE config 3, "bf16 with fp32 master"): the kernel W is a packed bf16 copy made by
 * mstts_pack_bf16_fwd / _bwd from the fp32 master (round to nearest even; lane-consumption order, opaque), the activation block is
 * rounded to bf16 on the way into LDS, accumulation and the partial slabs are fp32:
 *   fwd: P[ks][M][N] = bf(X[M, K-slice ks]) . bf(W[K-slice ks, N])      N % 64 == 0, K % (64*ksplit) == 0, K/ksplit <= 512
 *   bwd: P[ns][M][R] = bf(dG[M, N-slice ns]) . bf(W[R, N-slice ns])^T   R % 32 == 0, N % (64*nsplit) == 0, N/nsplit <= 1024
 * The split counts are baked into the packed layout: pack and multiply with the same value (mstts_skinny_bf16_*_splits). */
int32_t mstts_skinny_bf16_fwd_splits(int64_t N, int64_t K);
int32_t mstts_skinny_bf16_bwd_splits(int64_t R, int64_t N);
int mstts_pack_bf16_fwd(const float* W, int64_t ldw, void* Wp, int64_t K, int64_t N, int32_t ksplit, mstts_stream_t s);
int mstts_pack_bf16_bwd(const float* W, int64_t ldw, void* Wq, int64_t R, int64_t N, int32_t nsplit, mstts_stream_t s);
int mstts_skinny_fwd_bf16(const float* X, int64_t ldx, const void* Wp, float* P, int64_t pstride, int64_t M, int64_t N, int64_t K,
                          int32_t ksplit, mstts_stream_t s);
int mstts_skinny_bwd_bf16(const float* dG, int64_t ldg, const void* Wq, float* P, int64_t pstride, int64_t M, int64_t R, int64_t N,
                          int32_t nsplit, mstts_stream_t s);

/* ---- LSTM weight utilities --------------------------------------------------------------------
 * fold_rows: dst[r,:] = src[r,:] for r<r0 ; dst[r0+i,:] = src[r0+i,:] + src[r0+n+i,:] (i<n) ; rest shifted up.
 * Used for the decoder cell-0 kernel whose context rows appear twice. */
int mstts_fold_rows(const float* src, float* dst, int64_t rows, int64_t cols, int64_t r0, int64_t n, mstts_stream_t s);

/* ---- tf.nn.dynamic_rnn over one ZoneoutLSTMCell (Encoder_BiLSTM Modules.py:49-73, Taco1 BiRNN
 * Taco1_Mel_to_Spect/Modules.py:75-99, speaker Stack_LSTM Speaker_Embedding/Modules.py:12-35).
 * The input product xw = x.Wx + bias is hoisted by the caller into one big GEMM; this driver
 * enqueues the T dependent steps (recurrent GEMM + fused cell) natively.
 * hist buffers are step-major: c_hist/h_hist [T+1,B,H] (slot 0 is zeroed here), acts [T,B,4H],
 * c_raw [T,B,H] (acts/c_raw may be NULL at inference).  zc/zh [T,B,H] in processing order. */
typedef struct {
    int64_t B, T, H;
    const float* xw;                 /* [B,T,4H], batch-major, bias included */
    const float* wh; int64_t wh_ld;  /* recurrent rows of the cell kernel: [H,4H] view, row stride wh_ld */
    const int32_t* lengths;          /* [B] or NULL */
    int32_t reverse;
    float zoneout;
    const uint8_t* zc; const uint8_t* zh;
    const float* residual;           /* [B,T,H] or NULL */
    float* out; int64_t out_sb, out_st;
    float* c_hist; float* h_hist; float* acts; float* c_raw;
    float* gates_ws;                 /* mstts_lstm_seq_ws_floats(B, H, 0) floats */
    /* optional fused steps (mstts_cell_fwd: recurrent product + cell update in one launch): wh_p = the [H,4H] recurrent rows packed by
     * mstts_pack_cell_fwd(wh, wh_ld, wh_p, H, H); h_p = 2 * mstts_cell_act_floats(B, H) floats of scratch.  Used when both are
     * non-NULL, mstts_cell_fwd_supported(H, H) and there is no residual input; otherwise product + pointwise launches. */
    const float* wh_p; float* h_p;
} mstts_lstm_seq_fwd_desc;
int mstts_lstm_seq_fwd(const mstts_lstm_seq_fwd_desc* d, mstts_stream_t s);
/* the two directions of a bidirectional layer (same B, T, H) advanced together: one launch per step for both when the fused form is
 * available (else the two sequences run one after the other) */
int mstts_lstm_seq_fwd_pair(const mstts_lstm_seq_fwd_desc* a, const mstts_lstm_seq_fwd_desc* b, mstts_stream_t s);
/* floats needed for gates_ws (backward = 0) or for the BPTT ws (backward = 1) */
int64_t mstts_lstm_seq_ws_floats(int64_t B, int64_t H, int32_t backward);

/* BPTT over the same sequence.  d_out is the gradient of `out` (same strides).  Produces
 * dgates_step [T,B,4H] (pairs with h_hist[0:T] for dWh) and dgates_pos [B,T,4H] (pairs with x for
 * dWx / dX); the weight/bias/input gradients are then plain GEMMs/colsums done by the caller.
 * ws: mstts_lstm_seq_ws_floats(B, H, 1) floats. */
typedef struct {
    int64_t B, T, H;
    const float* wh; int64_t wh_ld;
    const int32_t* lengths; int32_t reverse;
    float zoneout;
    const uint8_t* zc; const uint8_t* zh;
    const float* d_out; int64_t dout_sb, dout_st;
    const float* c_hist; const float* acts; const float* c_raw;
    float* dgates_step; float* dgates_pos;
    float* ws;
} mstts_lstm_seq_bwd_desc;
int mstts_lstm_seq_bwd(const mstts_lstm_seq_bwd_desc* d, mstts_stream_t s);
/* BPTT of the two directions together: two launches per step (pointwise pair + product pair) instead of four */
int mstts_lstm_seq_bwd_pair(const mstts_lstm_seq_bwd_desc* a, const mstts_lstm_seq_bwd_desc* b, mstts_stream_t s);

/* Both directions of a bidirectional layer, ALL T steps in ONE launch each way (csrc/persist_lstm.hip; H == 256, B <= 32, no residual
 * input): the recurrent kernels stay in registers, the hidden state (forward) / the gate gradients (BPTT) travel between the workgroups
 * through a small ring in `xch`.  Same descriptors, same buffers and same results (up to the order of the partial sums) as
 * mstts_lstm_seq_fwd_pair / mstts_lstm_seq_bwd_pair (the packed-kernel / scratch fields wh, wh_p, h_p, gates_ws, ws are not used).
 *   pk_* / pkt_*: mstts_persist_lstm_pack(wh, wh_ld, fwd_pk, bwd_pk) copies of each direction's recurrent rows,
 *                 mstts_persist_lstm_pack_floats() floats each, refreshed when the variables change;
 *   xch: mstts_persist_lstm_ws_bytes() bytes, 16-byte aligned;  ctrl: 16 uint32;
 *   hist: mstts_persist_lstm_hist_floats(T) floats, 16-byte aligned: the forward call packs its per-step inputs there, the loop writes its
 *         history there (one contiguous kilobyte per wave access instead of 16 row-strided pieces) and a streaming kernel of the same call
 *         then fills the descriptor's row-major tensors; the BPTT call reads the SAME buffer (so it must follow a persistent forward
 *         call on it) and packs / unpacks its own input / output through bws (mstts_persist_lstm_bwd_floats(T) floats).
 * After the launch ctrl[1] == 0 and ctrl[2] == 64 (forward) / 32 (BPTT) <=> it ran to its end; anything else (a bounded wait expired)
 * means the outputs are incomplete and the caller re-runs the launch-per-step entry point. */
int32_t mstts_persist_lstm_supported(int64_t B, int64_t H);
int64_t mstts_persist_lstm_pack_floats(void);
int64_t mstts_persist_lstm_ws_bytes(void);
int mstts_persist_lstm_pack(const float* wh, int64_t wh_ld, float* fwd_pk, float* bwd_pk, mstts_stream_t s);
int64_t mstts_persist_lstm_hist_floats(int64_t T);
int64_t mstts_persist_lstm_bwd_floats(int64_t T);
int mstts_lstm_seq_fwd_pair_persistent(const mstts_lstm_seq_fwd_desc* a, const mstts_lstm_seq_fwd_desc* b, const float* pk_a, const float* pk_b,
                                       float* xch, uint32_t* ctrl, float* hist, mstts_stream_t s);
int mstts_lstm_seq_bwd_pair_persistent(const mstts_lstm_seq_bwd_desc* a, const mstts_lstm_seq_bwd_desc* b, const float* pkt_a, const float* pkt_b,
                                       float* xch, uint32_t* ctrl, const float* hist, float* bws, mstts_stream_t s);
/* The same launches for MORE THAN 32 ROWS and for a single (unidirectional) sequence: rows are independent recurrences, so B rows run as
 * ceil(B / 32) row groups, each with its own workgroups and ring (ndir x groups <= 16: up to 512 rows of one sequence, 256 of a pair; the
 * speaker-encoder trainer's 320 utterances are 10 groups).  Buffer sizes for ndir sequences (1, or 2 = a bidirectional pair) of B rows:
 * mstts_persist_lstm_ws_bytes_n / _hist_floats_n / _bwd_floats_n; ctrl: 16 uint32, ctrl[2] == 32 x groups (forward) / 16 x groups (BPTT)
 * after a complete run.  The pair entry points above accept any B that mstts_persist_lstm_supported_n(B, H, 2) admits. */
int32_t mstts_persist_lstm_supported_n(int64_t B, int64_t H, int32_t ndir);
/* The FORWARD launches (mstts_lstm_seq_fwd_persistent / _pair_persistent) also cover H == 128 - the Taco1 vocoder's BiRNN at inference
 * (Taco1_Mel_to_Spect/Modules.py:75-99): H / 8 workgroups per row group; pack that kernel with mstts_persist_lstm_pack_fwd (H = 256 or 128,
 * H * 4H floats).  Buffers sized by the same *_n functions (they are sized for H = 256).  After a complete run ctrl[2] == (H / 8) x row groups. */
int32_t mstts_persist_lstm_fwd_supported_n(int64_t B, int64_t H, int32_t ndir);
int mstts_persist_lstm_pack_fwd(const float* wh, int64_t wh_ld, int64_t H, float* fwd_pk, mstts_stream_t s);
int64_t mstts_persist_lstm_ws_bytes_n(int64_t B, int32_t ndir);
int64_t mstts_persist_lstm_hist_floats_n(int64_t T, int64_t B, int32_t ndir);
int64_t mstts_persist_lstm_bwd_floats_n(int64_t T, int64_t B, int32_t ndir);
int mstts_lstm_seq_fwd_persistent(const mstts_lstm_seq_fwd_desc* a, const float* pk, float* xch, uint32_t* ctrl, float* hist, mstts_stream_t s);
int mstts_lstm_seq_bwd_persistent(const mstts_lstm_seq_bwd_desc* a, const float* pkt, float* xch, uint32_t* ctrl, const float* hist, float* bws, mstts_stream_t s);

/* ---- Decoder_LSTM / Decoder_Dynamic_Decode in teacher-forcing mode (Modules.py:76-119,323-472
 * with the TF AttentionWrapper step).  Everything that does not depend on the
 * recurrence is hoisted by the caller: xw0 = prenet(frames).Wx0 + b0 for all S steps, and the
 * output projection / losses afterwards.  This driver enqueues the S dependent steps:
 *   LSTM0([ctx_{s-1}|h0_{s-1}]) -> LSTM1([m0|h1_{s-1}]) -> query -> energy -> softmax/context.
 * Row-block layouts (step-major, so each step's GEMM input is one contiguous [B,*] matrix):
 *   in0 [S+1,B,M+H] = [ctx | h0 state]   (slot 0 zeroed here; slot s+1 written by step s)
 *   in1 [S+1,B,2H]  = [m0  | h1 state]   (m0 of step s in slot s, h1 state of step s in slot s+1)
 *   pj  [S,B,H+M]   = [m1  | ctx]        (the projection's input rows)
 *   c0/c1 [S+1,B,H], acts0/acts1 [S,B,4H], craw0/craw1 [S,B,H], q_hist [S,B,A],
 *   align_hist [S,B,T], cum_hist [S+1,B,T] (slot 0 zeroed here). */
typedef struct {
    int64_t B, S, H, P;
    mstts_lsa_const lsa;             /* B,T,A,M,KS,CH + keys/values/lengths + attention weights */
    const float* xw0;                /* [S,B,4H] */
    const float* w0f;                /* [M+H,4H]: folded context rows, then recurrent rows */
    const float* w1; const float* b1;   /* [2H,4H], [4H] */
    const float* wq;                 /* [H,A] */
    const uint8_t* zc0; const uint8_t* zh0; const uint8_t* zc1; const uint8_t* zh1;   /* [S,B,H] or NULL */
    float zoneout;
    float* in0; float* in1; float* pj;
    float* c0; float* c1; float* acts0; float* acts1; float* craw0; float* craw1;
    float* q_hist; float* align_hist; float* cum_hist;
    float* gates_ws; float* energy_ws;      /* [parts,B,4H] (parts = max skinny K-splits, see mstts_decoder_train_ws_floats), 2*B*T+2 floats (8-byte aligned) */
    float* q_ws;                            /* [parts,B,A] query partials */
    int32_t chains;                         /* independent row groups run on separate HIP streams (0/1 = one; must divide B) */
    /* optional bf16 mode of the recurrent products (BASELINE config 3): packed bf16 copies of w0f / w1 / wq made with
     * mstts_pack_bf16_fwd (first three) and mstts_pack_bf16_bwd (last three) using the split counts of
     * mstts_decoder_bf16_splits(H, M, A, out[6]); all six non-NULL -> cell / query products and their data gradients run
     * as bf(X).bf(W) with fp32 accumulation, everything else stays fp32 */
    const void* bf_w0f_f; const void* bf_w1_f; const void* bf_wq_f; const void* bf_w0f_b; const void* bf_w1_b; const void* bf_wq_b;
    /* optional fused cell steps (fp32): w0f / w1 packed by mstts_pack_cell_fwd; both non-NULL and
     * mstts_cell_fwd_supported(H, M+H) && (H, 2H) -> each cell is one mstts_cell_fwd launch instead of product + pointwise */
    const float* w0p; const float* w1p;
    const void* w0p16; const void* w1p16;   /* ... or, in the bf16 mode, their mstts_pack_cell_fwd_bf16 copies (fused bf16 cell steps) */
    /* optional packed kernels of the BPTT data-gradient products (mstts_pack_skinny_bwd with the split counts of
     * mstts_skinny_bwd_splits(M+H, 4H) / (2H, 4H) / (H, A)); NULL -> the row-major kernels are streamed */
    const float* w0f_bp; const float* w1_bp; const float* wq_bp;
    const float* wq_t;  /* optional [A/4,H,4] re-layout of wq (mstts_transpose01(wq, wq_t, H, A/4, 4)): the query layer's data gradient is folded into cell 1's pointwise backward */
    float* act_p;      /* ... and their packed activation blocks: 2 * (mstts_cell_act_floats(B, M+H) + mstts_cell_act_floats(B, 2H)) floats */
    /* floats available behind energy_ws.  >= mstts_lsa_step_q_ws_bytes(B, T) / 4 (and mstts_lsa_step_q_supported(T, M, H), lsa.loc_kt
     * set): the query projection runs inside the attention launch (mstts_lsa_step_fwd_q) - 3 launches per forward step instead of 4.
     * 0 = the 2*B*T+2 floats of the plain form. */
    int64_t energy_ws_floats;
} mstts_decoder_train_desc;
int32_t mstts_decoder_bf16_splits(int64_t H, int64_t M, int64_t A, int32_t* out6);
/* floats needed for gates_ws (*gates) and q_ws (*q) */
int mstts_decoder_train_ws_floats(int64_t B, int64_t H, int64_t M, int64_t A, int64_t* gates, int64_t* q);
int mstts_decoder_train_fwd(const mstts_decoder_train_desc* d, mstts_stream_t s);

/* ---- The same S steps as ONE persistent launch (csrc/persist.hip; Modules.py:397-443 with ZoneoutLSTMCell.py:228-271 and
 * Location_Sensitive_Attention.py:43-85 inside): 256 co-resident workgroups keep both cell kernels, the query kernel and the row's
 * keys / values on chip for the whole sequence and hand the recurrent data from CU to CU through small rings in `xch`.  It reads and
 * writes exactly the buffers of mstts_decoder_train_desc that mstts_decoder_train_fwd does (in0, in1, pj, c0, c1, acts*, craw*, q_hist,
 * align_hist, cum_hist; the packed blocks and workspaces of the launch-per-step path are not touched), so mstts_decoder_train_bwd
 * runs behind either.  Reference widths only: mstts_persist_fwd_supported(B <= 32, H == 1024, M == 768, A == 128, T <= 128, KS == 31)
 * and a device that admits all 256 workgroups at once (occupancy query, >= 256 CUs).
 *   w0pk / w1pk / wqpk: mstts_persist_pack(w0f, w1, wq) copies (mstts_persist_pack_floats(0 / 1 / 2) floats), refreshed when the
 *                       variables change;  xch: mstts_persist_fwd_ws_bytes() bytes, 16-byte aligned;  ctrl: 272 uint32.
 * After the launch: ctrl[1] == 0 and ctrl[2] == 256  <=>  the sequence ran to its end.  Anything else (the workgroups were not
 * co-resident within the start window, e.g. because another kernel held CUs; or a bounded wait expired) means the outputs are
 * incomplete: the caller re-runs mstts_decoder_train_fwd, which recomputes every step (abort codes: 1 = start rendezvous timed out,
 * 2 = a hand-off wait expired, 3 = self-test).  stamps (NULL in production): 256 x 16
 * uint64 of summed 100 MHz wall-clock ticks per stage and workgroup (bench.py's in-kernel stage timing). */
typedef struct {
    const float* w0pk; const float* w1pk; const float* wqpk;
    float* xch; uint32_t* ctrl; uint64_t* stamps;
    float* opk;                   /* packed operands of the BPTT's cell updates, mstts_persist_opk_floats(S) floats: written by the persistent forward
                                     INSTEAD of the row-major histories acts0/1, craw0/1, c0/1 (NULL: those are written), read by the persistent
                                     BPTT (required there).  mstts_persist_unpack_history() converts for mstts_decoder_train_bwd. */
    int32_t selftest_fail_step;   /* 0 in production; k > 0: workgroup 0 raises the abort word at step k - 1 (exercises the fallback) */
    int32_t near_xcd;             /* != 0: hand-offs whose producer and all consumers report the same hardware XCC id at the start rendezvous are
                                     published with plain stores and stay in that XCD's L2 (-3 ms per step); every launch first drops its
                                     XCDs' copies of those rings.  0: every hand-off write-through (placement never matters for correctness) */
    const float* pre; const float* b0;
                                  /* forward only, optional: the prenet output [S, B, 256] (step-major, 16-byte aligned) and the cell-0 bias [4H].
                                     When given, the launch forms the prenet rows' share of the cell-0 gates itself (8 more k-steps per wave,
                                     kernel rows from the wx0 argument of mstts_persist_pack) and ignores mstts_decoder_train_desc.xw0: the
                                     caller skips that [S B, 256] x [256, 4H] product and its 16 KB-per-row tensor.  NULL: xw0 is read.
                                     ARITHMETIC: with pre given (and recurrent_bf16 == 0, at most 128 encoder positions) the two products that sit on
                                     the step's chain - context + prenet rows into cell 0, the cell-0 output into cell 1 - are evaluated as the EXACT
                                     three-way bf16 split of both operands, six products on v_mfma_f32_16x16x32_bf16 with fp32 accumulators: fp32 accuracy
                                     (dropped terms <= 2^-26 relative) and the edge semantics stated at mstts_gemm_split3 (an operand that is inf / NaN or
                                     rounds to bf16 infinity gives NaN where IEEE gives inf; denormal planes may be flushed).  With pre == NULL every
                                     product of the loop runs on the f32-input MFMA (bitwise an fmaf chain): that is the form to select together
                                     with mstts_gemm_split3(0) when IEEE behaviour at those edges is wanted (TrainEngine.exact_f32_products; bench.py's
                                     f32_input_mfma_everywhere leg runs it). */
    int32_t recurrent_bf16;       /* != 0 (BASELINE config 3, "bf16 with fp32 master"): both cell products, the query product and - in the BPTT launch - the
                                     data-gradient products take their operands rounded to bf16 (round to nearest even; the kernels when they are loaded into
                                     registers, the activations / gate gradients when they are staged) and run on v_mfma_f32_16x16x32_bf16 with fp32 accumulators;
                                     states, gates, softmax, histories and everything exchanged stay fp32.  The forward launch needs pre / b0 in this mode.
                                     (Round 4 had a software-pipelined schedule selector in this slot: removed, tools/persist_pipe.inc.) */
} mstts_persist_desc;
int32_t mstts_persist_fwd_supported(int64_t B, int64_t H, int64_t M, int64_t A, int64_t T, int64_t KS);
int64_t mstts_persist_fwd_ws_bytes(void);
int64_t mstts_persist_pack_floats(int32_t which);
/* wx0: the prenet rows [256, 4H] of the cell-0 kernel (row stride 4H), or NULL (zeros are packed: the launch then needs xw0) */
int mstts_persist_pack(const float* w0f, const float* w1, const float* wq, const float* wx0, float* w0pk, float* w1pk, float* wqpk, mstts_stream_t s);
int mstts_decoder_train_fwd_persistent(const mstts_decoder_train_desc* d, const mstts_persist_desc* p, mstts_stream_t s);
int64_t mstts_persist_opk_floats(int64_t S);
int mstts_persist_unpack_history(const float* opk, const mstts_decoder_train_desc* d, mstts_stream_t s);

/* BPTT through the same S steps.  d_pj [S,B,H+M] holds the projection's input gradient on entry
 * (d_m1 | d_ctx) and is updated in place.  Outputs for the hoisted gradient GEMMs:
 *   dg0/dg1 [S,B,4H], dq_hist [S,B,A] (must be zeroed by the caller), de_hist [S,B,T],
 *   d_in0 [parts0,S,B,M+H] = partial slabs of dg0.w0f^T per step (rows of step s are the gradient of in0
 *   slot s; parts0 = mstts_decoder_train_bwd_parts(...); the slabs must be summed by the consumer).
 * ws: mstts_decoder_train_bwd_ws_floats(...) floats. */
typedef struct {
    const mstts_decoder_train_desc* fwd;
    float* d_pj; float* dg0; float* dg1; float* dq_hist; float* de_hist; float* d_in0;
    float* ws;
} mstts_decoder_train_bwd_desc;
int mstts_decoder_train_bwd(const mstts_decoder_train_bwd_desc* d, mstts_stream_t s);
int64_t mstts_decoder_train_bwd_ws_floats(int64_t B, int64_t H, int64_t M, int64_t A, int64_t T, int64_t CH);
int32_t mstts_decoder_train_bwd_parts(int64_t H, int64_t M);   /* number of d_in0 slabs */

/* ---- BPTT through the same S steps as ONE persistent launch (csrc/persist_bwd.hip), the counterpart of
 * mstts_decoder_train_fwd_persistent: same geometry and support rule (mstts_persist_bwd_supported), same descriptor type with
 *   w0pk / w1pk / wqpk = the mstts_persist_bwd_pack(w0f, w1, wq) copies (transposed kernels in the lanes' order,
 *                        mstts_persist_bwd_pack_floats(0 / 1 / 2) floats),  xch = mstts_persist_bwd_ws_bytes() bytes,  ctrl = 272 uint32.
 * It reads the forward history of bd->fwd and bd->d_pj and writes what mstts_decoder_train_bwd writes - dg0, dg1 [S,B,4H], dq_hist
 * (no pre-zeroing needed), de_hist - with ONE difference: d_in0 receives, in its FIRST slab only, the complete gradient of the
 * context rows (columns 0..M-1 of slots 1..S-1; the h0 columns are not written), so a consumer sums 1 slab instead of
 * mstts_decoder_train_bwd_parts().  ctrl[1] == 0 and ctrl[2] == 256 after the launch <=> complete; otherwise re-run
 * mstts_decoder_train_bwd (it overwrites everything). */
int32_t mstts_persist_bwd_supported(int64_t B, int64_t H, int64_t M, int64_t A, int64_t T, int64_t KS);
int64_t mstts_persist_bwd_ws_bytes(void);
int64_t mstts_persist_bwd_pack_floats(int32_t which);
int mstts_persist_bwd_pack(const float* w0f, const float* w1, const float* wq, float* w0t, float* w1t, float* wqt, mstts_stream_t s);
int mstts_decoder_train_bwd_persistent(const mstts_decoder_train_bwd_desc* bd, const mstts_persist_desc* p, mstts_stream_t s);

/* ---- free-running decoder steps (inference branch of Decoder_Helper.next_inputs,
 * Modules.py:212-237): enqueues steps [step0, step0+n).  frame feedback: step s reads its input
 * frame from linear rows of step s-1 (zeros for s == 0).  Buffers as in the train descriptor but
 * only two state slots are kept (ping-pong by step parity); outputs: linear [Smax,B,n_mel]
 * and stop [Smax,B] step-major, align_hist [Smax,B,T].  prenet keep-masks pm0/pm1 [Smax,B,P]. */
typedef struct {
    int64_t B, H, P, n_mel, Smax;
    mstts_lsa_const lsa;
    const float* pw0; const float* pb0; const float* pw1; const float* pb1;   /* prenet dense kernels */
    const uint8_t* pm0; const uint8_t* pm1; float prenet_keep;
    const float* wx0; const float* b0;      /* [P,4H] input rows of cell 0, [4H] */
    const float* w0f; const float* w1; const float* b1; const float* wq;
    const float* wproj; const float* bproj; /* [H+M, n_mel+1], [n_mel+1] */
    float zoneout;
    float* in0; float* in1; float* pj;      /* [2,B,P+M+H] (the generic path uses [2,B,M+H] of it), [2,B,2H], [B,H+M] */
    float* c0; float* c1;                   /* [2,B,H] */
    float* cum;                             /* [2,B,T] */
    float* pre_ws;                          /* mstts_decoder_infer_ws_floats(...) floats, 8-byte aligned */
    float* linear; float* stop; float* align_hist;
    /* optional, enable the weight-streaming path when mstts_decoder_infer_fast(...) == 1:
     * w0s = [wx0 ; w0f] stacked [P+M+H, 4H]; wp_pad = wproj zero-padded to [H+M, 4*ceil((n_mel+1)/4)] columns */
    const float* w0s; const float* wp_pad;
    /* optional fused cell steps on that path: w0sp / w1p = w0s / w1 packed by mstts_pack_cell_fwd, act_p = 2 * (mstts_cell_act_floats(B, P+M+H)
     * + mstts_cell_act_floats(B, 2H)) floats of scratch: each cell becomes one launch (7 launches per frame instead of 9) */
    const float* w0sp; const float* w1p; float* act_p;
    /* optional, with the fused cell steps: the output projection inside the attention launch (mstts_lsa_step_fwd_qp) -
     * wp_own = mstts_lsa_proj_pack(wproj rows 0..H-1), vp [B, T, 4*ceil((n_mel+1)/4)] = values . wp_pad[H:, :] for this utterance batch */
    const float* wp_own; const float* vp;
} mstts_decoder_infer_desc;
int32_t mstts_decoder_infer_fast(int64_t B, int64_t H, int64_t P, int64_t M, int64_t A, int64_t n_mel);
int mstts_decoder_infer_steps(const mstts_decoder_infer_desc* d, int64_t step0, int64_t n, mstts_stream_t s);
int64_t mstts_decoder_infer_ws_floats(int64_t B, int64_t H, int64_t P, int64_t T, int64_t A, int64_t n_mel);

/* ---- the same free-running loop as ONE persistent launch (csrc/persist_infer.hip): every step of Decoder_Dynamic_Decode.body in
 * inference mode (Modules.py:397-443; Decoder_Helper.next_inputs :212-237: the step's own frame -> prenet with dropout always on
 * :239-255 -> next input; stop gating :216-219, finished OR-accumulated :409, loop condition :395; projection :309-321) on 256 co-resident
 * workgroups, cell kernels in registers, keys / values / prenet operands on chip, at most Smax = Hyper_Parameters.py:53 + 1 steps.
 * Reads from mstts_decoder_infer_desc: B, H, P, n_mel, Smax, lsa (keys, values, lengths, loc_k, loc_b, score_w, score_b), b0, b1, pw1, pb1 (second prenet
 * layer), pm0 / pm1 (keep-masks [Smax, B, P]: row s = the prenet that FEEDS step s), prenet_keep, zoneout; writes linear [Smax, B, n_mel],
 * stop [Smax, B], align_hist [Smax, B, T] for the steps it ran.
 * Loop invariants the caller prepares once per batch (plain mstts_gemm_f32 products; W1p / b1p = first prenet layer, Wp = [Wp_m ; Wp_c] the
 * projection kernel padded to 84 columns, bp its padded bias):
 *   vp  [B T, 84]  = values . Wp_c                 u   [B T, 256] = vp[:, :80] . W1p
 *   wfm [H, 256]   = Wp_m[:, :80] . W1p            bf  [256]      = bp[:80] . W1p + b1p
 *   pre0 [B, 256]  = prenet of the all-zero start frame with the masks of step 0 (Modules.py:178-185)
 *   w0pk / w1pk    = mstts_persist_pack(w0f, w1, wq, wx0) copies of the cell kernels (prenet rows folded in),
 *   wqppk          = mstts_persist_infer_pack(wq, wp_pad, 84, wfm): mstts_persist_infer_pack_floats() floats.
 * xch: mstts_persist_infer_ws_bytes() bytes; ctrl: 272 uint32, afterwards [1] = abort code (0 = none: 1 start rendezvous, 2 a bounded wait
 * expired, 3 self-test), [2] = workgroups that left in order (256), [5] = number of valid steps (n + 1 where n is the step at which the last row
 * finished; the launch runs one more step than that before every workgroup has seen the word - its rows are to be ignored).  Non-zero [1] or
 * [2] != 256: run mstts_decoder_infer_steps instead (it rewrites every output). */
typedef struct {
    const float* w0pk; const float* w1pk; const float* wqppk;
    const float* pre0; const float* bf; const float* u; const float* vp; const float* bp_pad;
    float* xch; uint32_t* ctrl;
    void* stamps;                    /* NULL, or 256 x 24 uint64: per-workgroup stage ticks (100 MHz) of the profiling instantiation */
    int32_t selftest_fail_step;      /* k > 0: workgroup 0 raises the abort word at step k - 1 (tests) */
    int32_t near_xcd;
} mstts_persist_infer_desc;
int32_t mstts_persist_infer_supported(int64_t B, int64_t H, int64_t P, int64_t M, int64_t A, int64_t T, int64_t KS, int64_t n_mel);
int64_t mstts_persist_infer_ws_bytes(void);
int64_t mstts_persist_infer_pack_floats(void);
int mstts_persist_infer_pack(const float* wq, const float* wp_pad, int64_t wp_ld, const float* wfm, float* wqppk, mstts_stream_t s);
int mstts_decoder_infer_persistent(const mstts_decoder_infer_desc* d, const mstts_persist_infer_desc* p, mstts_stream_t s);

/* ---- bf16 gradient exchange (BASELINE config 3, dist.GradAllReduce(comm_dtype="bf16")): the message is bf16, the sum is fp32.
 * f32_to_bf16 rounds to nearest even; bf16_chunks_sum: out[i] = bf16(sum_r float(chunks[r*stride + i])), r = 0..nchunks-1 in order. */
int mstts_f32_to_bf16(const float* x, void* y_bf16, int64_t n, mstts_stream_t s);
int mstts_bf16_to_f32(const void* x_bf16, float* y, int64_t n, mstts_stream_t s);
int mstts_bf16_chunks_sum(const void* chunks_bf16, int32_t nchunks, int64_t stride, int64_t n, void* out_bf16, mstts_stream_t s);

/* ---- profiling probes (bench.py only; process-global, not thread-safe, never armed on the product
 * path): HIP events bracket every launch of one kernel kind inside the decoder loop drivers, on the
 * launch stream.  begin(kind, max_launches) arms (kind 0 disarms); after a stream synchronise,
 * result() returns the number of bracketed launches and their summed duration in ms. */
enum { MSTTS_PROBE_OFF = 0, MSTTS_PROBE_LSA_ENERGY = 1, MSTTS_PROBE_LSA_CONTEXT = 2, MSTTS_PROBE_CELL0_GEMM = 3,
       MSTTS_PROBE_CELL1_GEMM = 4, MSTTS_PROBE_LSA_DALIGN = 5, MSTTS_PROBE_LSA_DENERGY = 6, MSTTS_PROBE_CELL0_DGEMM = 7,
       MSTTS_PROBE_CELL1_DGEMM = 8 };
int mstts_probe_begin(int32_t kind, int64_t max_launches);
/* after a stream synchronise: launches bracketed, their summed event-to-event time (ms) and the summed time of an empty
 * event bracket recorded right behind each one (what the event pair itself costs on that stream) */
int64_t mstts_probe_result(double* total_ms, double* empty_total_ms);

#ifdef __cplusplus
}
#endif
#endif /* MSTTS_H_ */
