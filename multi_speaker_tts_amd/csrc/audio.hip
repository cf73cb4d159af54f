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

// ---------------------------------------------------------------------------------------------
// The same transform as ONE launch: a workgroup per frame, real FFT in LDS.
//   x[i] = window[i - off] * preemph(reflect(f*hop + i - n_fft/2))   (0 outside the window)       Audio.py:42-48,62-64
//   z[m] = x[2m] + i x[2m+1]  ->  Stockham radix-4 FFT of n_fft/2 complex points (ping-pong LDS buffers, twiddles from a table
//   built in fp64 on the host)  ->  X[k] = E[k] + e^{-2 pi i k / n_fft} O[k]  ->  |X[k]|, k = 0 .. n_fft/2
//   spec_out = clip((20 log10(max(1e-5, |X|)) - ref_db + 100) / 100, 0, 1)                          Audio.py:19-22,91-92
//   mel_out  = symmetric-normalised dB of mel_basis . |X| over each filter's non-zero bin range     Audio.py:29-32,78-80,94-96
// Several waveforms per launch: frame g belongs to waveform w with frame_off[w] <= g < frame_off[w + 1].
// Algorithmic bytes per frame: hop new samples in, n_mel (+ n_fft/2 + 1) floats out; 5 N log2 N flops - the kernel is bound by
// LDS round trips (log4(N/2) stages), a few microseconds per workgroup, 4+ workgroups resident per CU.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void stft_fft_kernel(const float* __restrict__ wav, const long* __restrict__ wav_off,
                                                       const long* __restrict__ frame_off, int nw, float coef,
                                                       const float* __restrict__ window, const float2* __restrict__ tw,
                                                       const float* __restrict__ mel_basis, const int* __restrict__ mel_rng, int n_fft, int hop,
                                                       int win, int n_mel, float max_abs, float ref_db, float* __restrict__ mel_out,
                                                       float* __restrict__ spec_out, const float* __restrict__ mag_in,
                                                       const float* __restrict__ sub, float sub_scale, int flags) {
    extern __shared__ __attribute__((aligned(16))) float2 fft_lds[];
    const int N2 = n_fft >> 1, tid = threadIdx.x;
    float2* bufa = fft_lds;
    float2* bufb = fft_lds + N2;
    const long g = blockIdx.x;
    int w = 0;
    for (int lo = 0, hi = nw; hi - lo > 1;) {            // uniform binary search: frame_off[w] <= g
        const int mid = (lo + hi) >> 1;
        if (frame_off[mid] <= g) lo = mid; else hi = mid;
        w = lo;
    }
    const long f = g - frame_off[w], n = wav_off[w + 1] - wav_off[w];
    const float* x = wav + wav_off[w];
    const int off = (n_fft - win) >> 1, pad = n_fft >> 1;
    auto sample = [&](int i) -> float {                  // windowed, pre-emphasised, reflect-padded sample i of this frame
        if (i < off || i >= off + win) return 0.f;
        long j = f * hop + i - pad;
        if (j < 0) j = -j;
        if (j >= n) j = 2 * (n - 1) - j;
        if (j < 0) j = 0;
        const float prev = j > 0 ? x[j - 1] : 0.f;
        return window[i - off] * (x[j] - coef * prev);
    };
    const int NB = N2 + 1;
    float* mag = reinterpret_cast<float*>(bufb);
    if (mag_in) {
        // second pass of the spectral-subtraction form (Audio.py:45-46): magnitudes of the first pass minus sub[k], clipped at 0
        for (int k = tid; k < NB; k += 256) {
            const float m = fmaxf(mag_in[g * NB + k] - (sub ? sub[k] * sub_scale : 0.f), 0.f);
            mag[k] = m;
            if (spec_out) {
                const float db = 20.f * log10f(fmaxf(1e-5f, m)) - ref_db;
                spec_out[g * NB + k] = fminf(fmaxf((db + 100.f) * 0.01f, 0.f), 1.f);
            }
        }
        __syncthreads();
    } else {
    for (int m = tid; m < N2; m += 256) bufa[m] = make_float2(sample(2 * m), sample(2 * m + 1));
    __syncthreads();
    // Stockham autosort, radix 4 while it fits and one radix-2 stage for the odd power (N2 = 512, 2048).  Stage with sub-transform
    // size p: thread i takes x[i + m t], m < radix, t = N2 / radix, twiddles e^{-2 pi i m k / (radix p)} = tw[m k n_fft / (radix p)]
    // (tw has n_fft entries; 3 m k n_fft / (4 p) < 3 n_fft / 4).
    int p = 1;
    for (; p * 4 <= N2; p <<= 2) {
        const int t4 = N2 >> 2, tmul = n_fft / (4 * p);
        for (int i = tid; i < t4; i += 256) {
            const int k = i & (p - 1), m = k * tmul;         // e^{-2 pi i k / (4 p)} = tw[k n_fft / (4 p)]
            const float2 w1 = tw[m], w2 = tw[2 * m], w3 = tw[3 * m];
            const float2 x0 = bufa[i], x1 = bufa[i + t4], x2 = bufa[i + 2 * t4], x3 = bufa[i + 3 * t4];
            const float2 u1 = make_float2(x1.x * w1.x - x1.y * w1.y, x1.x * w1.y + x1.y * w1.x);
            const float2 u2 = make_float2(x2.x * w2.x - x2.y * w2.y, x2.x * w2.y + x2.y * w2.x);
            const float2 u3 = make_float2(x3.x * w3.x - x3.y * w3.y, x3.x * w3.y + x3.y * w3.x);
            const float2 v0 = make_float2(x0.x + u2.x, x0.y + u2.y), v1 = make_float2(x0.x - u2.x, x0.y - u2.y);
            const float2 v2 = make_float2(u1.x + u3.x, u1.y + u3.y), v3 = make_float2(u1.y - u3.y, u3.x - u1.x);   // (u1 - u3) * (-i)
            const int j0 = ((i - k) << 2) + k;
            bufb[j0] = make_float2(v0.x + v2.x, v0.y + v2.y);
            bufb[j0 + p] = make_float2(v1.x + v3.x, v1.y + v3.y);
            bufb[j0 + 2 * p] = make_float2(v0.x - v2.x, v0.y - v2.y);
            bufb[j0 + 3 * p] = make_float2(v1.x - v3.x, v1.y - v3.y);
        }
        __syncthreads();
        float2* t_ = bufa; bufa = bufb; bufb = t_;
    }
    if (p < N2) {                                            // p == N2 / 2: the last radix-2 stage
        const int tmul = n_fft / (2 * p);
        for (int i = tid; i < (N2 >> 1); i += 256) {
            const int k = i & (p - 1);
            const float2 a = bufa[i], b = bufa[i + (N2 >> 1)], t = tw[k * tmul];
            const float2 bt = make_float2(b.x * t.x - b.y * t.y, b.x * t.y + b.y * t.x);
            const int j0 = ((i - k) << 1) + k;
            bufb[j0] = make_float2(a.x + bt.x, a.y + bt.y);
            bufb[j0 + p] = make_float2(a.x - bt.x, a.y - bt.y);
        }
        __syncthreads();
        float2* t_ = bufa; bufa = bufb; bufb = t_;
    }
    // real-input post-processing -> magnitudes in LDS (bufb is free)
    mag = reinterpret_cast<float*>(bufb);
    for (int k = tid; k < NB; k += 256) {
        const float2 zk = bufa[k & (N2 - 1)], zc = bufa[(N2 - k) & (N2 - 1)];
        const float er = 0.5f * (zk.x + zc.x), ei = 0.5f * (zk.y - zc.y);           // E = (Z[k] + conj(Z[N2-k])) / 2
        const float orr = 0.5f * (zk.y + zc.y), oi = -0.5f * (zk.x - zc.x);         // O = -i (Z[k] - conj(Z[N2-k])) / 2
        const float2 t = k < N2 ? tw[k] : make_float2(-1.f, 0.f);
        const float re = er + orr * t.x - oi * t.y, im = ei + orr * t.y + oi * t.x;
        const float m = sqrtf(re * re + im * im);
        mag[k] = m;
        if (spec_out) {
            if (flags & 2) spec_out[g * NB + k] = m;                    // raw magnitudes (first pass of the spectral-subtraction form)
            else {
                const float db = 20.f * log10f(fmaxf(1e-5f, m)) - ref_db;
                spec_out[g * NB + k] = fminf(fmaxf((db + 100.f) * 0.01f, 0.f), 1.f);
            }
        }
    }
    __syncthreads();
    }
    if (!mel_out) return;
    // mel: three threads per filter, each sums every third bin of the filter's non-zero range; the three partial sums are
    // combined in fixed order (bit-reproducible).  No wave reductions: twenty dependent shuffle chains per wave cost more than
    // the transform itself.
    float* part = reinterpret_cast<float*>(bufa);             // the spectrum has been consumed
    for (int c0 = 0; c0 < n_mel; c0 += 85) {                   // 85 filters x 3 threads per pass
        const int c = c0 + tid / 3, q = tid - (tid / 3) * 3;
        if (tid < 255 && c < n_mel) {
            const int lo = mel_rng[2 * c], hi = mel_rng[2 * c + 1];
            const float* row = mel_basis + (long)c * NB;
            float acc = 0.f;
            for (int b = lo + q; b < hi; b += 3) acc = fmaf(row[b], mag[b], acc);
            part[tid] = acc;
        }
        __syncthreads();
        if (tid < 85 && c0 + tid < n_mel) {
            const int cc = c0 + tid;
            const float m = (part[3 * tid] + part[3 * tid + 1]) + part[3 * tid + 2];
            const float db = 20.f * log10f(fmaxf(1e-5f, m));
            if (flags & 1) mel_out[g * n_mel + cc] = fminf(fmaxf((db + 100.f) * 0.01f, 0.f), 1.f);          // Audio._normalize
            else {
                const float v = (2.f * max_abs) * ((db + 100.f) * 0.01f) - max_abs;                        // Audio._symmetric_normalize
                mel_out[g * n_mel + cc] = fminf(fmaxf(v, -max_abs), max_abs);
            }
        }
        __syncthreads();
    }
}

}  // namespace mstts
using namespace mstts;

extern "C" int mstts_stft_fft_supported(int32_t n_fft, int32_t win) {
    return n_fft >= 512 && n_fft <= 4096 && (n_fft & (n_fft - 1)) == 0 && win >= 2 && win <= n_fft;
}

extern "C" int mstts_stft_fft(const float* wav, const int64_t* wav_off, const int64_t* frame_off, int32_t nw, float preemph,
                              const float* window, const float* twiddle, const float* mel_basis, const int32_t* mel_rng, int32_t n_fft,
                              int32_t hop, int32_t win, int32_t n_mel, float max_abs, float ref_level_db, float* mel_out, float* spec_out,
                              int64_t total_frames, const float* mag_in, const float* sub, float sub_scale, int32_t flags, mstts_stream_t s) {
    MSTTS_REQUIRE(wav && wav_off && frame_off && window && twiddle && (mel_out || spec_out), MSTTS_ERR_SHAPE, "stft_fft: null pointer");
    MSTTS_REQUIRE(!mel_out || (mel_basis && mel_rng && n_mel >= 1), MSTTS_ERR_SHAPE, "stft_fft: mel output needs the filterbank and its ranges");
    MSTTS_REQUIRE(mstts_stft_fft_supported(n_fft, win) && hop >= 1 && nw >= 1, MSTTS_ERR_SHAPE, "stft_fft: n_fft must be a power of two in [512, 4096]");
    MSTTS_REQUIRE(total_frames >= 0 && total_frames < (1LL << 31), MSTTS_ERR_SHAPE, "stft_fft: frame count");
    if (total_frames == 0) return MSTTS_OK;
    hipLaunchKernelGGL(stft_fft_kernel, dim3((unsigned)total_frames), dim3(256), sizeof(float2) * (size_t)n_fft, (hipStream_t)s, wav,
                       (const long*)wav_off, (const long*)frame_off, (int)nw, preemph, window, (const float2*)twiddle, mel_basis,
                       (const int*)mel_rng, (int)n_fft, (int)hop, (int)win, (int)n_mel, max_abs, ref_level_db, mel_out, spec_out, mag_in, sub, sub_scale, (int)flags);
    MSTTS_CHECK_LAUNCH("stft_fft");
    return MSTTS_OK;
}

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
