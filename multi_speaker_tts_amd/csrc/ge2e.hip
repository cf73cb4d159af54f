// GE2E speaker-verification loss of the reference's speaker-encoder trainer, forward + backward in one launch
// (Speaker_Embedding/Modules.py:39-98 with Embedding_Generate :39-40; "Softmax" method).
//   e_i   = x_i * rsqrt(max(|x_i|^2, 1e-12))                         (tf.nn.l2_normalize(inputs[:, -1, :], axis=1))
//   Sum_k = sum of the P embeddings of speaker k;  cw_i = (Sum_s(i) - e_i)/(P-1);  cb_k = Sum_k / P
//   cosw_i   = e_i.cw_i / (|e_i| |cw_i|)                              (Cosine_Similarity, no epsilon)
//   cosb_i,k = e_i.cb_k / (|cb_k| |e_i| + 1e-8),  k != s(i)            (Cosine_Similarity2D)
//   loss = mean_i  -log softmax([w cosw_i - b, {w cosb_i,k - b}])[0]   (tf.losses.sparse_softmax_cross_entropy, label 0)
// Everything is a function of G = E.Sum^T [N,S], n_k = |Sum_k|^2 and q_i = |e_i|^2, so the backward is
//   dE = dG.Sum + 2 dq e + (dG^T.E + 2 dn Sum)[s(i)],   dx = (dE - e (e.dE)) / |x|.
// The problem is tiny (N = 320, D = 256, S = 32 at the reference sizes): ONE workgroup of 1024 threads walks the phases with
// block barriers; intermediates live in the caller's workspace.
#include "common.h"

namespace mstts {

__global__ __launch_bounds__(1024) void ge2e_loss_kernel(const float* __restrict__ x, long ldx, int N, int D, int S, int P,
                                                         const float* __restrict__ wb, float* __restrict__ out /* loss, dw, db */,
                                                         float* __restrict__ dx, long lddx, float* __restrict__ ws) {
    __shared__ float scratch[16];
    __shared__ float s_n[256], s_dn[256];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nw = blockDim.x >> 6;
    float* E = ws;                       // [N,D]
    float* Sum = E + (long)N * D;        // [S,D]
    float* G = Sum + (long)S * D;        // [N,S]
    float* dG = G + (long)N * S;         // [N,S]
    float* dSum = dG + (long)N * S;      // [S,D]
    float* inv = dSum + (long)S * D;     // [N]  1/|x|
    float* q = inv + N;                  // [N]  |e|^2
    float* dq = q + N;                   // [N]
    const float w = wb[0], b = wb[1];
    // 1) normalise rows
    for (int i = wave; i < N; i += nw) {
        float ss = 0.f;
        for (int d = lane; d < D; d += 64) { const float v = x[(long)i * ldx + d]; ss += v * v; }
        ss = wave_sum(ss);
        const float r = rsqrtf(fmaxf(ss, 1e-12f));
        for (int d = lane; d < D; d += 64) E[(long)i * D + d] = x[(long)i * ldx + d] * r;
        if (lane == 0) { inv[i] = r; q[i] = ss * r * r; }
    }
    if (tid < 256) s_dn[tid] = 0.f;
    __syncthreads();
    // 2) speaker sums and their squared norms
    for (int e = tid; e < S * D; e += blockDim.x) {
        const int k = e / D, d = e - k * D;
        float v = 0.f;
        for (int p = 0; p < P; ++p) v += E[(long)(k * P + p) * D + d];
        Sum[e] = v;
    }
    __syncthreads();
    for (int k = wave; k < S; k += nw) {
        float ss = 0.f;
        for (int d = lane; d < D; d += 64) { const float v = Sum[(long)k * D + d]; ss += v * v; }
        ss = wave_sum(ss);
        if (lane == 0) s_n[k] = ss;
    }
    // 3) G = E . Sum^T
    for (int e = wave; e < N * S; e += nw) {
        const int i = e / S, k = e - i * S;
        float v = 0.f;
        for (int d = lane; d < D; d += 64) v += E[(long)i * D + d] * Sum[(long)k * D + d];
        v = wave_sum(v);
        if (lane == 0) G[e] = v;
    }
    __syncthreads();
    // 4) per sample: logits, softmax cross-entropy, gradients w.r.t. G, q, n, w
    float loss_acc = 0.f, dw_acc = 0.f;
    const float invN = 1.f / (float)N, pm1 = (float)(P - 1), fP = (float)P;
    for (int i = tid; i < N; i += blockDim.x) {
        const int s = i / P;
        const float a = G[(long)i * S + s], qi = q[i], sq = sqrtf(qi), ns = s_n[s];
        const float dotw = (a - qi) / pm1, nw2 = (ns - 2.f * a + qi) / (pm1 * pm1);
        const float cosw = dotw / (sq * sqrtf(nw2));
        const float l0 = w * cosw - b;
        float mx = l0;
        for (int k = 0; k < S; ++k) {
            if (k == s) continue;
            const float cb = (G[(long)i * S + k] / fP) / (sqrtf(s_n[k]) / fP * sq + 1e-8f);
            mx = fmaxf(mx, w * cb - b);
        }
        float den = expf(l0 - mx);
        for (int k = 0; k < S; ++k) {
            if (k == s) continue;
            const float cb = (G[(long)i * S + k] / fP) / (sqrtf(s_n[k]) / fP * sq + 1e-8f);
            den += expf(w * cb - b - mx);
        }
        loss_acc += (logf(den) - (l0 - mx)) * invN;
        const float p0 = expf(l0 - mx) / den, g0 = (p0 - 1.f) * invN;
        float dqi = 0.f;
        // within term
        {
            const float dc = g0 * w;
            dw_acc += g0 * cosw;
            const float ddot = dc / (sq * sqrtf(nw2));
            const float dnw2 = dc * cosw * (-0.5f) / nw2;
            dqi += dc * cosw * (-0.5f) / qi - ddot / pm1 + dnw2 / (pm1 * pm1);
            dG[(long)i * S + s] = ddot / pm1 - 2.f * dnw2 / (pm1 * pm1);
            atomicAdd(&s_dn[s], dnw2 / (pm1 * pm1));
        }
        for (int k = 0; k < S; ++k) {
            if (k == s) continue;
            const float snk = sqrtf(s_n[k]);
            const float nb = snk / fP, dn_ = nb * sq + 1e-8f;
            const float cb = (G[(long)i * S + k] / fP) / dn_;
            const float gk = expf(w * cb - b - mx) / den * invN;
            dw_acc += gk * cb;
            const float dc = gk * w;
            dG[(long)i * S + k] = dc / dn_ / fP;
            const float dden = -dc * cb / dn_;
            dqi += dden * nb * 0.5f / sq;
            atomicAdd(&s_dn[k], dden * sq * 0.5f / (snk * fP));
        }
        dq[i] = dqi;
    }
    loss_acc = block_sum(loss_acc, scratch);
    dw_acc = block_sum(dw_acc, scratch);
    if (tid == 0) { out[0] = loss_acc; out[1] = dw_acc; out[2] = 0.f; }     // d/db: the softmax gradients sum to zero
    __syncthreads();
    // 5) dSum = dG^T . E + 2 dn Sum
    for (int e = tid; e < S * D; e += blockDim.x) {
        const int k = e / D, d = e - k * D;
        float v = 2.f * s_dn[k] * Sum[e];
        for (int i = 0; i < N; ++i) v += dG[(long)i * S + k] * E[(long)i * D + d];
        dSum[e] = v;
    }
    __syncthreads();
    // 6) dE and the normalisation backward, one wave per row
    for (int i = wave; i < N; i += nw) {
        const int s = i / P;
        float de[16];                                      // D <= 1024
        float dot = 0.f;
#pragma unroll
        for (int c = 0; c < 16; ++c) {
            const int d = lane + 64 * c;
            float v = 0.f;
            if (d < D) {
                v = 2.f * dq[i] * E[(long)i * D + d] + dSum[(long)s * D + d];
                for (int k = 0; k < S; ++k) v += dG[(long)i * S + k] * Sum[(long)k * D + d];
                dot += v * E[(long)i * D + d];
            }
            de[c] = v;
        }
        dot = wave_sum(dot);
#pragma unroll
        for (int c = 0; c < 16; ++c) {
            const int d = lane + 64 * c;
            if (d < D) dx[(long)i * lddx + d] = (de[c] - E[(long)i * D + d] * dot) * inv[i];
        }
    }
}

}  // namespace mstts
using namespace mstts;

extern "C" int64_t mstts_ge2e_ws_floats(int64_t N, int64_t D, int64_t S) { return N * D + 2 * S * D + 2 * N * S + 3 * N + 16; }

/* x: [N = S*P, D] rows ldx apart (the last-frame outputs of the speaker LSTM stack), wb = {weight, bias} of the similarity;
 * out[0] = loss, out[1] = d loss / d weight, out[2] = d loss / d bias; dx rows lddx apart */
extern "C" int mstts_ge2e_loss_fwd_bwd(const float* x, int64_t ldx, int64_t S, int64_t P, int64_t D, const float* wb, float* out,
                                       float* dx, int64_t lddx, float* ws, mstts_stream_t s) {
    MSTTS_REQUIRE(x && wb && out && dx && ws, MSTTS_ERR_SHAPE, "ge2e_loss: null pointer");
    MSTTS_REQUIRE(S >= 2 && S <= 256 && P >= 2 && D >= 1 && D <= 1024, MSTTS_ERR_SHAPE, "ge2e_loss: need 2 <= speakers <= 256, >= 2 utterances each, D <= 1024");
    hipLaunchKernelGGL(ge2e_loss_kernel, dim3(1), dim3(1024), 0, (hipStream_t)s, x, (long)ldx, (int)(S * P), (int)D, (int)S, (int)P, wb, out, dx, (long)lddx, ws);
    MSTTS_CHECK_LAUNCH("ge2e_loss_fwd_bwd");
    return MSTTS_OK;
}
