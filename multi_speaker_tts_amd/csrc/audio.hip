// Audio.melspectrogram on gfx950 (Audio.py:12-13,29-32,42-48,62-96).
// The reference zero-pads an 800-tap Hann window to n_fft = 2048, so each STFT frame touches only
// 800 samples: the windowed DFT is a [frames, 800] x [800, 2*1025] contraction and goes through the
// fp32 MFMA GEMM (overlapping frames are just rows with stride hop), followed by the 1025 -> 80 mel
// filterbank GEMM and a fused dB / normalise pass.
#include "common.h"

namespace mstts {

__global__ void preemph_pad_kernel(const float* __restrict__ x, long n, float coef, int pad, float* __restrict__ out) {
    const long total = n + 2L * pad;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        long j = i - pad;
        if (j < 0) j = -j;                       // np.pad(mode='reflect')
        if (j >= n) j = 2 * (n - 1) - j;
        if (j < 0) j = 0;
        const float prev = j > 0 ? x[j - 1] : 0.f;
        out[i] = x[j] - coef * prev;
    }
}

__global__ void magnitude_kernel(const float* __restrict__ spec, float* __restrict__ mag, long frames, int NB) {
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < frames * NB; i += (long)gridDim.x * blockDim.x) {
        const long f = i / NB; const int k = (int)(i % NB);
        const float re = spec[f * 2 * NB + k], im = spec[f * 2 * NB + NB + k];
        mag[i] = sqrtf(re * re + im * im);
    }
}

__global__ void db_normalize_kernel(float* __restrict__ mel, long n, float max_abs) {
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const float db = 20.f * log10f(fmaxf(1e-5f, mel[i]));
        const float v = (2.f * max_abs) * ((db + 100.f) / 100.f) - max_abs;
        mel[i] = fminf(fmaxf(v, -max_abs), max_abs);
    }
}

}  // namespace mstts
using namespace mstts;

static inline long nb_of(int n_fft) { return ((n_fft / 2 + 1) + 3) / 4 * 4; }

extern "C" int64_t mstts_stft_mel_ws_floats(int64_t n, int32_t n_fft, int64_t frames) {
    const long NB = nb_of(n_fft);
    return ((n + n_fft + 3) / 4 * 4) + frames * 2 * NB + frames * NB;
}

extern "C" int mstts_stft_mel(const float* wav, int64_t n, float preemph, const float* dft_basis, const float* mel_basis_t,
                              int32_t n_fft, int32_t hop, int32_t win, int32_t n_mel, float max_abs, float* ws, float* mel_out,
                              int64_t frames, mstts_stream_t s) {
    MSTTS_REQUIRE(wav && dft_basis && mel_basis_t && ws && mel_out, MSTTS_ERR_SHAPE, "stft_mel: null pointer");
    MSTTS_REQUIRE(n >= 2 && frames == 1 + n / hop, MSTTS_ERR_SHAPE, "stft_mel: frames must be 1 + n / hop");
    MSTTS_REQUIRE(win <= n_fft && n > n_fft / 2, MSTTS_ERR_SHAPE, "stft_mel: signal shorter than the reflect pad");
    hipStream_t st = (hipStream_t)s;
    const long NB = nb_of(n_fft);
    const int pad = n_fft / 2;
    float* padded = ws;
    float* spec = ws + ((n + n_fft + 3) / 4 * 4);
    float* mag = spec + frames * 2 * NB;
    long tot = n + 2L * pad;
    hipLaunchKernelGGL(preemph_pad_kernel, dim3((unsigned)((tot + 255) / 256 > 2048 ? 2048 : (tot + 255) / 256)), dim3(256), 0, st,
                       wav, (long)n, preemph, pad, padded);
    MSTTS_CHECK_LAUNCH("preemph_pad");
    // frame f, tap i  ->  padded[f*hop + (n_fft - win)/2 + i]
    mstts_gemm_desc g;
    memset(&g, 0, sizeof(g));
    g.A = padded + (n_fft - win) / 2; g.lda = hop;
    g.B = dft_basis; g.ldb = 2 * NB; g.C = spec; g.ldc = 2 * NB;
    g.M = frames; g.N = 2 * NB; g.K = win; g.alpha = 1.f; g.split_k = 1; g.batch = 1;
    int rc = mstts_gemm_f32(&g, s);
    if (rc) return rc;
    long nm = frames * NB;
    hipLaunchKernelGGL(magnitude_kernel, dim3((unsigned)((nm + 255) / 256 > 2048 ? 2048 : (nm + 255) / 256)), dim3(256), 0, st,
                       (const float*)spec, mag, (long)frames, (int)NB);
    MSTTS_CHECK_LAUNCH("magnitude");
    memset(&g, 0, sizeof(g));
    g.A = mag; g.lda = NB; g.B = mel_basis_t; g.ldb = n_mel; g.C = mel_out; g.ldc = n_mel;
    g.M = frames; g.N = n_mel; g.K = NB; g.alpha = 1.f; g.split_k = 1; g.batch = 1;
    rc = mstts_gemm_f32(&g, s);
    if (rc) return rc;
    long ne = frames * n_mel;
    hipLaunchKernelGGL(db_normalize_kernel, dim3((unsigned)((ne + 255) / 256 > 2048 ? 2048 : (ne + 255) / 256)), dim3(256), 0, st,
                       mel_out, ne, max_abs);
    MSTTS_CHECK_LAUNCH("db_normalize");
    return MSTTS_OK;
}
