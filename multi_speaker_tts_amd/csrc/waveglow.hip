// WaveGlow vocoder, inference direction (WaveGlow/Modules.py:177-208,210-327,354-371; WaveGlow/Inv1x1.py:9-41) - the
// pieces that are not plain contractions.  The contractions themselves (transposed-conv taps, dilated K=3 convs, the
// conditioning and res/skip 1x1 convs) run on mstts_gemm_f32 (window mode with win_dil for the dilated taps).
//   overlap_add   : conv2d_transpose(kernel (1,K), stride (1,S), VALID) epilogue: out[n, t*S + k, c] = sum of the tap
//                   products Y[n, t, k, c] that land on that sample, + bias          (Upsample_Mel :198-208)
//   gate          : z = tanh(a[:, :C]) * sigmoid(a[:, C:])                           (:286-291)
//   res_skip      : x = z + rs[:, :C] ; out (+)= rs[:, C:]   (last layer: out (+)= rs)   (:293-311; the residual is added to
//                   the GATED activation - the reference's quirk)
//   coupling_inv  : a1 = (a1 - b) * exp(-log_s) ; audio = [a0 | a1] . inv(W) ; optional early-noise prepend (:239-249,354-369)
//   philox_normal : N(0, sigma^2) draws (Box-Muller on Philox4x32-10 words) for tf.random.normal (:189-193,362-366)
#include "common.h"

namespace mstts {

__global__ void wg_overlap_add_kernel(const float* __restrict__ Y, const float* __restrict__ bias, float* __restrict__ out,
                                      int N, int T, int K, int S, int C) {
    const long L = (long)(T - 1) * S + K;
    const long n_el = (long)N * L * C;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n_el; i += (long)gridDim.x * blockDim.x) {
        const int c = (int)(i % C);
        const long l = (i / C) % L;
        const int n = (int)(i / (C * L));
        // taps: l = t*S + k, 0 <= k < K  ->  t from ceil((l-K+1)/S) to floor(l/S)
        long t_hi = l / S; if (t_hi > T - 1) t_hi = T - 1;
        long t_lo = (l - K + 1 + S - 1) / S; if (l - K + 1 <= 0) t_lo = 0;
        float v = bias ? bias[c] : 0.f;
        for (long t = t_lo; t <= t_hi; ++t) {
            const long k = l - t * S;
            v += Y[(((long)n * T + t) * K + k) * C + c];
        }
        out[i] = v;
    }
}

__global__ void wg_gate_kernel(const float* __restrict__ a, long lda, float* __restrict__ z, long rows, int C) {
    const long n = rows * C;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const long r = i / C; const int c = (int)(i - r * C);
        z[i] = tanhf(a[r * lda + c]) * sigmoid_acc(a[r * lda + C + c]);
    }
}

// ... with the dilated convolution's output in its own buffer b [rows, 2C] (see mstts_wg_gate_add): pre-activation = a + b
__global__ void wg_gate_add_kernel(const float* __restrict__ a, long lda, const float* __restrict__ b, float* __restrict__ z, long rows, int C) {
    const long n4 = rows * (C >> 2);
    const int c4n = C >> 2;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
        const long r = i / c4n; const int c = (int)(i - r * c4n) * 4;
        const float4 at = *reinterpret_cast<const float4*>(a + r * lda + c), as = *reinterpret_cast<const float4*>(a + r * lda + C + c);
        const float4 bt = *reinterpret_cast<const float4*>(b + r * 2 * C + c), bs = *reinterpret_cast<const float4*>(b + r * 2 * C + C + c);
        float4 o;
        o.x = tanhf(at.x + bt.x) * sigmoid_acc(as.x + bs.x); o.y = tanhf(at.y + bt.y) * sigmoid_acc(as.y + bs.y);
        o.z = tanhf(at.z + bt.z) * sigmoid_acc(as.z + bs.z); o.w = tanhf(at.w + bt.w) * sigmoid_acc(as.w + bs.w);
        *reinterpret_cast<float4*>(z + r * C + c) = o;
    }
}

__global__ void wg_res_skip_kernel(const float* __restrict__ z, const float* __restrict__ rs, float* __restrict__ x, float* __restrict__ out,
                                   long rows, int C, int last, int first) {
    const long n = rows * C;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const long r = i / C; const int c = (int)(i - r * C);
        float skip;
        if (last) skip = rs[r * C + c];
        else { x[i] = z[i] + rs[r * 2 * C + c]; skip = rs[r * 2 * C + C + c]; }
        out[i] = first ? skip : out[i] + skip;
    }
}

// one thread per row: c <= 16 channels
__global__ void wg_coupling_inv_kernel(const float* __restrict__ audio, const float* __restrict__ ls_b, const float* __restrict__ winv,
                                       const float* __restrict__ early, float sigma, float* __restrict__ out, long rows, int c, int ce) {
    const long r = blockIdx.x * (long)blockDim.x + threadIdx.x;
    if (r >= rows) return;
    const int h = c / 2;
    float x[16];
    for (int i = 0; i < h; ++i) x[i] = audio[r * c + i];
    for (int i = 0; i < h; ++i) x[h + i] = (audio[r * c + h + i] - ls_b[r * c + h + i]) * expf(-ls_b[r * c + i]);
    float* o = out + r * (c + ce);
    for (int e = 0; e < ce; ++e) o[e] = early[r * ce + e] * sigma;
    for (int j = 0; j < c; ++j) {
        float v = 0.f;
        for (int i = 0; i < c; ++i) v += x[i] * winv[i * c + j];
        o[ce + j] = v;
    }
}

__device__ __forceinline__ void philox4(unsigned c0, unsigned c1, unsigned c2, unsigned c3, unsigned k0, unsigned k1, unsigned (&r)[4]) {
#pragma unroll
    for (int i = 0; i < 10; ++i) {
        const unsigned long long p0 = 0xD2511F53ull * c0, p1 = 0xCD9E8D57ull * c2;
        const unsigned n0 = (unsigned)(p1 >> 32) ^ c1 ^ k0, n1 = (unsigned)p1, n2 = (unsigned)(p0 >> 32) ^ c3 ^ k1, n3 = (unsigned)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    r[0] = c0; r[1] = c1; r[2] = c2; r[3] = c3;
}
// draw i: words (2*(i&1), 2*(i&1)+1) of philox block i>>1 of stream (seed, stream): Box-Muller cosine branch
__global__ void philox_normal_kernel(float* __restrict__ out, long n, unsigned k0, unsigned k1, unsigned stream, float sigma) {
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const unsigned long long blk = (unsigned long long)(i >> 1);
        unsigned w[4];
        philox4((unsigned)blk, (unsigned)(blk >> 32), stream, 0u, k0, k1, w);
        const unsigned a = w[2 * (i & 1)], b = w[2 * (i & 1) + 1];
        const float u1 = ((float)(a >> 8) + 0.5f) * 5.9604644775390625e-8f;      // (0,1)
        const float u2 = (float)(b >> 8) * 5.9604644775390625e-8f;               // [0,1)
        out[i] = sigma * sqrtf(-2.f * logf(u1)) * cosf(6.283185307179586f * u2);
    }
}

}  // namespace mstts
using namespace mstts;
#define ST(s) ((hipStream_t)(s))
static unsigned wg_grid(long n) { long b = (n + 255) / 256; if (b > 8192) b = 8192; if (b < 1) b = 1; return (unsigned)b; }

extern "C" int mstts_wg_overlap_add(const float* Y, const float* bias, float* out, int64_t N, int64_t T, int64_t K, int64_t S, int64_t C, mstts_stream_t s) {
    MSTTS_REQUIRE(Y && out && N >= 1 && T >= 1 && K >= 1 && S >= 1 && C >= 1, MSTTS_ERR_SHAPE, "wg_overlap_add: bad arguments");
    hipLaunchKernelGGL(wg_overlap_add_kernel, dim3(wg_grid(N * ((T - 1) * S + K) * C)), dim3(256), 0, ST(s), Y, bias, out, (int)N, (int)T, (int)K, (int)S, (int)C);
    MSTTS_CHECK_LAUNCH("wg_overlap_add");
    return MSTTS_OK;
}
extern "C" int mstts_wg_gate(const float* a, int64_t lda, float* z, int64_t rows, int64_t C, mstts_stream_t s) {
    MSTTS_REQUIRE(a && z && rows >= 0 && C >= 1 && lda >= 2 * C, MSTTS_ERR_SHAPE, "wg_gate: bad arguments");
    if (rows == 0) return MSTTS_OK;
    hipLaunchKernelGGL(wg_gate_kernel, dim3(wg_grid(rows * C)), dim3(256), 0, ST(s), a, (long)lda, z, (long)rows, (int)C);
    MSTTS_CHECK_LAUNCH("wg_gate");
    return MSTTS_OK;
}
extern "C" int mstts_wg_gate_add(const float* a, int64_t lda, const float* b, float* z, int64_t rows, int64_t C, mstts_stream_t s) {
    MSTTS_REQUIRE(a && b && z && rows >= 0 && C >= 4 && C % 4 == 0 && lda >= 2 * C && lda % 4 == 0, MSTTS_ERR_SHAPE, "wg_gate_add: bad arguments");
    MSTTS_REQUIRE(aligned16(a) && aligned16(b) && aligned16(z), MSTTS_ERR_ALIGN, "wg_gate_add: 16-byte aligned operands required");
    if (rows == 0) return MSTTS_OK;
    hipLaunchKernelGGL(wg_gate_add_kernel, dim3(wg_grid(rows * C / 4)), dim3(256), 0, ST(s), a, (long)lda, b, z, (long)rows, (int)C);
    MSTTS_CHECK_LAUNCH("wg_gate_add");
    return MSTTS_OK;
}
extern "C" int mstts_wg_res_skip(const float* z, const float* rs, float* x, float* out, int64_t rows, int64_t C, int32_t last, int32_t first, mstts_stream_t s) {
    MSTTS_REQUIRE(z && rs && out && (last || x) && rows >= 0 && C >= 1, MSTTS_ERR_SHAPE, "wg_res_skip: bad arguments");
    if (rows == 0) return MSTTS_OK;
    hipLaunchKernelGGL(wg_res_skip_kernel, dim3(wg_grid(rows * C)), dim3(256), 0, ST(s), z, rs, x, out, (long)rows, (int)C, (int)last, (int)first);
    MSTTS_CHECK_LAUNCH("wg_res_skip");
    return MSTTS_OK;
}
extern "C" int mstts_wg_coupling_inv(const float* audio, const float* log_s_b, const float* w_inv, const float* early, float sigma, float* out,
                                     int64_t rows, int64_t c, int64_t c_early, mstts_stream_t s) {
    MSTTS_REQUIRE(audio && log_s_b && w_inv && out && c >= 2 && c <= 16 && c % 2 == 0 && c_early >= 0 && (c_early == 0 || early), MSTTS_ERR_SHAPE,
                  "wg_coupling_inv: 2 <= c <= 16 (even) and an early-noise block when c_early > 0");
    if (rows == 0) return MSTTS_OK;
    hipLaunchKernelGGL(wg_coupling_inv_kernel, dim3((unsigned)((rows + 255) / 256)), dim3(256), 0, ST(s), audio, log_s_b, w_inv, early, sigma, out,
                       (long)rows, (int)c, (int)c_early);
    MSTTS_CHECK_LAUNCH("wg_coupling_inv");
    return MSTTS_OK;
}
extern "C" int mstts_philox_normal(float* out, int64_t n, uint64_t seed, uint32_t stream_id, float sigma, mstts_stream_t s) {
    MSTTS_REQUIRE(out && n >= 0, MSTTS_ERR_SHAPE, "philox_normal: bad arguments");
    if (n == 0) return MSTTS_OK;
    hipLaunchKernelGGL(philox_normal_kernel, dim3(wg_grid(n)), dim3(256), 0, ST(s), out, (long)n, (unsigned)(seed & 0xffffffffu), (unsigned)(seed >> 32), (unsigned)stream_id, sigma);
    MSTTS_CHECK_LAUNCH("philox_normal");
    return MSTTS_OK;
}
