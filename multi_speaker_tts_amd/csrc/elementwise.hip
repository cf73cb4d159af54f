// HBM-bound elementwise / reduction kernels of the Tacotron2 hot path (gfx950).
// Everything here is streaming work: float4 accesses, grid-stride loops capped at 2048 blocks,
// wave64 shuffles for reductions.
#include "common.h"
#include <stdarg.h>

namespace mstts {

char* err_buf() {
    static thread_local char buf[512] = {0};
    return buf;
}
int set_err(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(err_buf(), 512, fmt, ap);
    va_end(ap);
    return code;
}

static inline int grid_for(long n, int per_block) {
    long b = (n + per_block - 1) / per_block;
    if (b < 1) b = 1;
    if (b > 2048) b = 2048;
    return (int)b;
}

// ---------------------------------------------------------------------------------------------
// Philox4x32-10 keep masks: draw i of (seed, stream) = word i&3 of philox(ctr=(i>>2,0,stream,0))
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ void philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3,
                                              uint32_t k0, uint32_t k1, uint32_t out[4]) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint64_t p0 = (uint64_t)0xD2511F53u * c0;
        const uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
        const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
        const uint32_t n1 = (uint32_t)p1;
        const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
        const uint32_t n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

__global__ void philox_mask_kernel(uint8_t* __restrict__ out, long n, uint32_t k0, uint32_t k1, uint32_t stream, float keep) {
    const long nblk = (n + 3) >> 2;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < nblk; i += (long)gridDim.x * blockDim.x) {
        uint32_t w[4];
        philox4x32_10((uint32_t)i, (uint32_t)((uint64_t)i >> 32), stream, 0u, k0, k1, w);
        uint8_t m[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) m[j] = ((float)(w[j] >> 8) * 5.9604644775390625e-08f < keep) ? 1 : 0;
        const long base = i << 2;
        if (base + 3 < n) {
            *reinterpret_cast<uchar4*>(out + base) = make_uchar4(m[0], m[1], m[2], m[3]);
        } else {
            for (int j = 0; j < 4 && base + j < n; ++j) out[base + j] = m[j];
        }
    }
}

// sample-keyed form: out is [outer, B, inner]; the mask of sample b is its own stream, counter = (block, sample0 + b, stream, 0),
// element (o, c) = draw o * inner + c of it - so a sample's mask does not depend on how the global batch is sharded over ranks
__global__ void philox_mask_rows_kernel(uint8_t* __restrict__ out, long outer, int B, long inner, uint32_t k0, uint32_t k1, uint32_t stream,
                                        uint32_t sample0, float keep) {
    const long n = outer * inner, nblk = (n + 3) >> 2, total = nblk * B;
    const bool vec = (inner & 3) == 0;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int b = (int)(i / nblk);
        const long blk = i - (long)b * nblk;
        uint32_t w[4];
        philox4x32_10((uint32_t)blk, sample0 + (uint32_t)b, stream, 0u, k0, k1, w);
        uint8_t m[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) m[j] = ((float)(w[j] >> 8) * 5.9604644775390625e-08f < keep) ? 1 : 0;
        const long e = blk << 2;
        if (vec) {                      // the four draws stay inside one [inner] run (and n % 4 == 0)
            const long o = e / inner, c = e - o * inner;
            *reinterpret_cast<uchar4*>(out + (o * B + b) * inner + c) = make_uchar4(m[0], m[1], m[2], m[3]);
        } else {
            for (int j = 0; j < 4 && e + j < n; ++j) {
                const long o = (e + j) / inner, c = (e + j) - o * inner;
                out[(o * B + b) * inner + c] = m[j];
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------
// embedding
// ---------------------------------------------------------------------------------------------
__global__ void embedding_fwd_kernel(const int32_t* __restrict__ tok, const float* __restrict__ table,
                                     float* __restrict__ out, long n, int vocab, int width) {
    const int w4 = width >> 2;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n * w4; i += (long)gridDim.x * blockDim.x) {
        const long row = i / w4; const int c = (int)(i % w4);
        int t = tok[row];
        t = t < 0 ? 0 : (t >= vocab ? vocab - 1 : t);
        reinterpret_cast<float4*>(out)[i] = reinterpret_cast<const float4*>(table + (long)t * width)[c];
    }
}
__global__ void embedding_bwd_kernel(const int32_t* __restrict__ tok, const float* __restrict__ dout,
                                     float* __restrict__ dtable, long n, int vocab, int width) {
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n * width; i += (long)gridDim.x * blockDim.x) {
        const long row = i / width; const int c = (int)(i % width);
        int t = tok[row];
        t = t < 0 ? 0 : (t >= vocab ? vocab - 1 : t);
        atomicAdd(dtable + (long)t * width + c, dout[i]);
    }
}

// fixed-order form (mstts_gemm_deterministic(1)): a thread owns one column of the table and walks the rows in order - no atomics
__global__ void embedding_bwd_det_kernel(const int32_t* __restrict__ tok, const float* __restrict__ dout,
                                         float* __restrict__ dtable, long n, int vocab, int width) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= width) return;
    for (long row = 0; row < n; ++row) {
        int t = tok[row];
        t = t < 0 ? 0 : (t >= vocab ? vocab - 1 : t);
        dtable[(long)t * width + c] += dout[row * width + c];
    }
}

// ---------------------------------------------------------------------------------------------
// column reductions over [rows, C]: each block owns 64 columns x a row chunk; threads (cx, ry) =
// (64 columns, 4 row lanes); partial sums go out with one atomic per column per block.
// ---------------------------------------------------------------------------------------------
// mode 0: out0 += sum x, out1 += sum x^2
// mode 1 (bn bwd): xhat = (x-mean)*rstd, dyn = dy*mask/keep: out0 += sum dyn, out1 += sum dyn*xhat
__global__ __launch_bounds__(256) void col_stats_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                        const uint8_t* __restrict__ mask, float inv_keep,
                                                        const float* __restrict__ mean, const float* __restrict__ rstd,
                                                        long rows, int C, long ld, int rows_per_block, int mode,
                                                        float* __restrict__ out0, float* __restrict__ out1) {
    __shared__ float s0[4][64], s1[4][64];
    const int cx = threadIdx.x & 63, ry = threadIdx.x >> 6;
    const int col = blockIdx.x * 64 + cx;
    const long r0 = (long)blockIdx.y * rows_per_block;
    const long r1 = min(rows, r0 + rows_per_block);
    float a0 = 0.f, a1 = 0.f;
    if (col < C) {
        float mu = 0.f, rs = 1.f;
        if (mode == 1) { mu = mean[col]; rs = rstd[col]; }
        for (long r = r0 + ry; r < r1; r += 4) {
            const float xv = x[r * ld + col];
            if (mode == 0) {
                a0 += xv; a1 += xv * xv;
            } else {
                float d = dy[r * (long)C + col];
                if (mask) d *= mask[r * (long)C + col] ? inv_keep : 0.f;
                a0 += d; a1 += d * (xv - mu) * rs;
            }
        }
    }
    s0[ry][cx] = a0; s1[ry][cx] = a1;
    __syncthreads();
    if (ry == 0 && col < C) {
        a0 = s0[0][cx] + s0[1][cx] + s0[2][cx] + s0[3][cx];
        a1 = s1[0][cx] + s1[1][cx] + s1[2][cx] + s1[3][cx];
        atomicAdd(out0 + col, a0);
        if (out1) atomicAdd(out1 + col, a1);
    }
}

// The same sums for C % 4 == 0 with float4 traffic: a lane owns 4 adjacent columns (a wave reads 1 KB of a row per load instead of
// 256 B), the 4 waves of a workgroup take rows r, r+4, .. of the chunk, UNR rows in flight per wave.
template <int MODE, int UNR>
__global__ __launch_bounds__(256) void col_stats4_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                         const uint8_t* __restrict__ mask, float inv_keep,
                                                         const float* __restrict__ mean, const float* __restrict__ rstd,
                                                         long rows, int C, long ld, int rows_per_block,
                                                         float* __restrict__ out0, float* __restrict__ out1) {
    __shared__ float4 s0[4][64], s1[4][64];
    const int cx = threadIdx.x & 63, ry = threadIdx.x >> 6;
    const int col = (blockIdx.x * 64 + cx) * 4;
    const long r0 = (long)blockIdx.y * rows_per_block;
    const long r1 = min(rows, r0 + rows_per_block);
    float4 a0 = make_float4(0.f, 0.f, 0.f, 0.f), a1 = a0;
    if (col < C) {
        float4 mu = a0, rs = make_float4(1.f, 1.f, 1.f, 1.f);
        if (MODE == 1) { mu = *reinterpret_cast<const float4*>(mean + col); rs = *reinterpret_cast<const float4*>(rstd + col); }
        for (long r = r0 + ry; r < r1; r += 4 * UNR) {
            float4 xv[UNR], dv[UNR];
            uint32_t mv[UNR];
#pragma unroll
            for (int u = 0; u < UNR; ++u) {
                const long rr = r + 4 * u;
                const bool ok = rr < r1;
                xv[u] = ok ? *reinterpret_cast<const float4*>(x + rr * ld + col) : make_float4(0.f, 0.f, 0.f, 0.f);
                if (MODE == 1) {
                    dv[u] = ok ? *reinterpret_cast<const float4*>(dy + rr * (long)C + col) : make_float4(0.f, 0.f, 0.f, 0.f);
                    mv[u] = (mask && ok) ? *reinterpret_cast<const uint32_t*>(mask + rr * (long)C + col) : 0x01010101u;
                    if (!ok) xv[u] = mu;
                }
            }
#pragma unroll
            for (int u = 0; u < UNR; ++u) {
                if (MODE == 0) {
                    a0.x += xv[u].x; a0.y += xv[u].y; a0.z += xv[u].z; a0.w += xv[u].w;
                    a1.x += xv[u].x * xv[u].x; a1.y += xv[u].y * xv[u].y; a1.z += xv[u].z * xv[u].z; a1.w += xv[u].w * xv[u].w;
                } else {
                    float4 d = dv[u];
                    if (mask) {
                        d.x *= (mv[u] & 0xffu) ? inv_keep : 0.f; d.y *= (mv[u] & 0xff00u) ? inv_keep : 0.f;
                        d.z *= (mv[u] & 0xff0000u) ? inv_keep : 0.f; d.w *= (mv[u] & 0xff000000u) ? inv_keep : 0.f;
                    }
                    a0.x += d.x; a0.y += d.y; a0.z += d.z; a0.w += d.w;
                    a1.x += d.x * (xv[u].x - mu.x) * rs.x; a1.y += d.y * (xv[u].y - mu.y) * rs.y;
                    a1.z += d.z * (xv[u].z - mu.z) * rs.z; a1.w += d.w * (xv[u].w - mu.w) * rs.w;
                }
            }
        }
    }
    s0[ry][cx] = a0; s1[ry][cx] = a1;
    __syncthreads();
    // 256 threads finish 64 x 4 columns: thread t -> column quad t >> 2, component t & 3
    const int q = threadIdx.x >> 2, comp = threadIdx.x & 3;
    const int oc = (blockIdx.x * 64 + q) * 4 + comp;
    if (oc < C) {
        const float* p0 = reinterpret_cast<const float*>(&s0[0][q]) + comp;
        const float* p1 = reinterpret_cast<const float*>(&s1[0][q]) + comp;
        const float v0 = p0[0] + p0[256] + p0[512] + p0[768];
        atomicAdd(out0 + oc, v0);
        if (out1) atomicAdd(out1 + oc, p1[0] + p1[256] + p1[512] + p1[768]);
    }
}

__global__ void bn_finalize_kernel(const float* __restrict__ ws, float inv_rows, float eps, float momentum,
                                   float* __restrict__ moving_mean, float* __restrict__ moving_var,
                                   float* __restrict__ save_mean, float* __restrict__ save_rstd, int C) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    const float mean = ws[c] * inv_rows;
    float var = ws[C + c] * inv_rows - mean * mean;
    var = fmaxf(var, 0.f);
    save_mean[c] = mean;
    save_rstd[c] = rsqrtf(var + eps);
    if (moving_mean) {
        moving_mean[c] = moving_mean[c] * momentum + mean * (1.f - momentum);
        moving_var[c] = moving_var[c] * momentum + var * (1.f - momentum);
    }
}

// y = ((x - mean) * rstd * gamma + beta) * mask/keep     (rstd_is_var: second stat is a variance)
__global__ void bn_apply_kernel(const float* __restrict__ x, const float* __restrict__ gamma, const float* __restrict__ beta,
                                const float* __restrict__ mean, const float* __restrict__ stat2, int stat2_is_var, float eps,
                                const uint8_t* __restrict__ mask, float inv_keep, float* __restrict__ y, long rows, int C) {
    const int c4n = C >> 2;
    const long n4 = rows * c4n;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
        const int c = (int)(i % c4n) * 4;
        const float4 xv = reinterpret_cast<const float4*>(x)[i];
        float xs[4] = {xv.x, xv.y, xv.z, xv.w}, o[4];
        uchar4 mk = make_uchar4(1, 1, 1, 1);
        if (mask) mk = reinterpret_cast<const uchar4*>(mask)[i];
        const uint8_t ms[4] = {mk.x, mk.y, mk.z, mk.w};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float rs = stat2[c + j];
            if (stat2_is_var) rs = rsqrtf(rs + eps);
            float v = (xs[j] - mean[c + j]) * rs * gamma[c + j] + beta[c + j];
            if (mask) v *= ms[j] ? inv_keep : 0.f;
            o[j] = v;
        }
        reinterpret_cast<float4*>(y)[i] = make_float4(o[0], o[1], o[2], o[3]);
    }
}

// dz = gamma*rstd*(dyn - dbeta/N - xhat*dgamma/N) * act'(x); ws = [sum dyn | sum dyn*xhat]
__global__ void bn_bwd_apply_kernel(const float* __restrict__ dy, const float* __restrict__ x, const float* __restrict__ gamma,
                                    const float* __restrict__ mean, const float* __restrict__ rstd,
                                    const uint8_t* __restrict__ mask, float inv_keep, int act, const float* __restrict__ ws,
                                    float inv_rows, float* __restrict__ dz, long rows, int C) {
    const int c4n = C >> 2;
    const long n4 = rows * c4n;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
        const int c = (int)(i % c4n) * 4;
        const float4 xv = reinterpret_cast<const float4*>(x)[i];
        const float4 dv = reinterpret_cast<const float4*>(dy)[i];
        float xs[4] = {xv.x, xv.y, xv.z, xv.w}, ds[4] = {dv.x, dv.y, dv.z, dv.w}, o[4];
        uchar4 mk = make_uchar4(1, 1, 1, 1);
        if (mask) mk = reinterpret_cast<const uchar4*>(mask)[i];
        const uint8_t ms[4] = {mk.x, mk.y, mk.z, mk.w};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float d = ds[j];
            if (mask) d *= ms[j] ? inv_keep : 0.f;
            const float xhat = (xs[j] - mean[c + j]) * rstd[c + j];
            float g = gamma[c + j] * rstd[c + j] * (d - ws[c + j] * inv_rows - xhat * ws[C + c + j] * inv_rows);
            if (act == MSTTS_ACT_RELU) g = xs[j] > 0.f ? g : 0.f;
            else if (act == MSTTS_ACT_TANH) g *= (1.f - xs[j] * xs[j]);
            else if (act == MSTTS_ACT_SIGMOID) g *= xs[j] * (1.f - xs[j]);
            o[j] = g;
        }
        reinterpret_cast<float4*>(dz)[i] = make_float4(o[0], o[1], o[2], o[3]);
    }
}

__global__ void vec_acc_kernel(float* __restrict__ dst, const float* __restrict__ src, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dst[i] += src[i];
}

// ---------------------------------------------------------------------------------------------
// flat elementwise
// ---------------------------------------------------------------------------------------------
__global__ void dropout_kernel(const float* __restrict__ x, const uint8_t* __restrict__ mask, float inv_keep,
                               float* __restrict__ y, long n) {
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x)
        y[i] = mask[i] ? x[i] * inv_keep : 0.f;
}
__global__ void relu_dropout_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ ys, const uint8_t* __restrict__ mask,
                                        float inv_keep, float* __restrict__ dx, long n) {
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x)
        dx[i] = (mask[i] && ys[i] > 0.f) ? dy[i] * inv_keep : 0.f;
}
__global__ void add_kernel(const float* __restrict__ a, const float* __restrict__ b, float* __restrict__ y, long n) {
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) y[i] = a[i] + b[i];
}
__global__ void fill_kernel(float* __restrict__ y, float v, long n) {
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) y[i] = v;
}
// 16-byte stores (y 16-byte aligned, n4 = n / 4 whole quads; the caller's tail goes through fill_kernel)
__global__ void fill4_kernel(float4* __restrict__ y, float v, long n4) {
    const float4 q = make_float4(v, v, v, v);
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) y[i] = q;
}
// One word that says whether up to two persistent launches ran to their end: control words [1] = abort code (0 = none), [2] = workgroups that
// finished (persist_common.h / persist_lstm.hip keep the same layout).  Null pointers are skipped.
__global__ void persist_status_kernel(const unsigned* a, unsigned done_a, const unsigned* b, unsigned done_b, int* flag) {
    bool ok = true;
    if (a) ok = ok && a[1] == 0u && a[2] == done_a;
    if (b) ok = ok && b[1] == 0u && b[2] == done_b;
    *flag = ok ? 1 : 0;
}
__global__ void copy2d_kernel(const float* __restrict__ src, long lds, float* __restrict__ dst, long ldd, long rows, long cols, int acc) {
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < rows * cols; i += (long)gridDim.x * blockDim.x) {
        const long r = i / cols, c = i % cols;
        const float v = src[r * lds + c];
        if (acc) dst[r * ldd + c] += v; else dst[r * ldd + c] = v;
    }
}
__global__ void maxpool2_kernel(const float* __restrict__ x, float* __restrict__ y, long B, long T, long C) {
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < B * T * C; i += (long)gridDim.x * blockDim.x) {
        const long t = (i / C) % T;
        float v = x[i];
        if (t + 1 < T) v = fmaxf(v, x[i + C]);
        y[i] = v;
    }
}
__global__ void highway_kernel(const float* __restrict__ hp, const float* __restrict__ tp, const float* __restrict__ x,
                               float* __restrict__ y, long n) {
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const float H = fmaxf(hp[i], 0.f), T = sigmoidf_(tp[i]);
        y[i] = H * T + x[i] * (1.f - T);
    }
}
// max_pooling1d(2, stride 1, 'same') backward: y[t] = max(x[t], x[t+1]); the gradient goes to the first maximum (x[t] on ties)
__global__ void maxpool2_bwd_kernel(const float* __restrict__ x, const float* __restrict__ dy, float* __restrict__ dx, long B, long T, long C) {
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < B * T * C; i += (long)gridDim.x * blockDim.x) {
        const long t = (i / C) % T;
        const float xv = x[i];
        float g = 0.f;
        if (t + 1 >= T || xv >= x[i + C]) g += dy[i];              // window t = {t, t+1}: x[t] wins (or is alone)
        if (t > 0 && xv > x[i - C]) g += dy[i - C];                 // window t-1 = {t-1, t}: x[t] wins strictly
        dx[i] = g;
    }
}
// highway combine backward: y = H*T + x*(1-T), H = relu(h), T = sigmoid(t)
__global__ void highway_bwd_kernel(const float* __restrict__ hp, const float* __restrict__ tp, const float* __restrict__ x,
                                   const float* __restrict__ dy, float* __restrict__ dh, float* __restrict__ dt, float* __restrict__ dx, long n) {
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const float h = hp[i], H = fmaxf(h, 0.f), T = sigmoid_acc(tp[i]), g = dy[i];
        dh[i] = h > 0.f ? g * T : 0.f;
        dt[i] = g * (H - x[i]) * T * (1.f - T);
        dx[i] = g * (1.f - T);
    }
}
// tf.losses.absolute_difference: mean |p - t| over all elements; d/dp = sign(p - t) / n
__global__ void l1_loss_kernel(const float* __restrict__ p, const float* __restrict__ t, long n, float inv_n, float* __restrict__ loss,
                               float* __restrict__ dp) {
    __shared__ float scratch[16];
    float acc = 0.f;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const float d = p[i] - t[i];
        acc += fabsf(d);
        if (dp) dp[i] = (d > 0.f ? inv_n : (d < 0.f ? -inv_n : 0.f));
    }
    acc = block_sum(acc, scratch);
    if (threadIdx.x == 0) atomicAdd(loss, acc * inv_n);
}
__global__ void fold_rows_kernel(const float* __restrict__ src, float* __restrict__ dst, long rows, long cols, long r0, long n) {
    const long out_rows = rows - n;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < out_rows * cols; i += (long)gridDim.x * blockDim.x) {
        const long r = i / cols, c = i % cols;
        float v;
        if (r < r0) v = src[r * cols + c];
        else if (r < r0 + n) v = src[r * cols + c] + src[(r + n) * cols + c];
        else v = src[(r + n) * cols + c];
        dst[i] = v;
    }
}

// ---------------------------------------------------------------------------------------------
// zoneout LSTM cell: pointwise part, forward and backward
// ---------------------------------------------------------------------------------------------
__global__ void lstm_point_fwd_kernel(mstts_lstm_point_fwd_desc d) {
    const long n = d.B * d.H;
    const int H = (int)d.H;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const int b = (int)(i / H), u = (int)(i % H);
        int len = d.lengths ? d.lengths[b] : 0x7fffffff;
        const bool live = d.step < len;
        const int pos = d.reverse ? (live ? len - 1 - d.step : d.step) : d.step;
        const long hpl = d.h_prev_ld ? d.h_prev_ld : H, hnl = d.h_next_ld ? d.h_next_ld : H;
        const float cp = d.c_prev[i], hp = d.h_prev[b * hpl + u];
        float* outp = d.out ? d.out + b * d.out_sb + pos * d.out_st + u : nullptr;
        if (!live) {
            if (outp) *outp = 0.f;
            d.c_next[i] = cp; d.h_next[b * hnl + u] = hp;
            if (d.acts_out) { for (int g = 0; g < 4; ++g) d.acts_out[(long)b * 4 * H + g * H + u] = 0.f; }
            if (d.c_raw) d.c_raw[i] = cp;
            continue;
        }
        float g4[4];
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            float v = sum_parts<MSTTS_MAX_PARTS>(d.gates_h, d.gates_parts, d.gates_pstride, (long)b * 4 * H + g * H + u);
            if (d.xw) v += d.xw[b * d.xw_sb + pos * d.xw_st + g * H + u];
            if (d.bias) v += d.bias[g * H + u];
            g4[g] = v;
        }
        const float si = sigmoidf_(g4[0]), tj = tanhf_(g4[1]), sf = sigmoidf_(g4[2] + 1.0f), so = sigmoidf_(g4[3]);
        const float c = sf * cp + si * tj;
        const float m = so * tanhf_(c);
        float dc = c - cp, dm = m - hp;
        if (d.zc) dc = d.zc[i] ? dc : 0.f;
        if (d.zh) dm = d.zh[i] ? dm : 0.f;
        d.c_next[i] = (1.f - d.zoneout) * dc + cp;
        d.h_next[b * hnl + u] = (1.f - d.zoneout) * dm + hp;
        if (outp) {
            float o = m;
            if (d.residual) o += d.residual[b * d.res_sb + pos * d.res_st + u];
            *outp = o;
        }
        if (d.acts_out) {
            float* a = d.acts_out + (long)b * 4 * H + u;
            a[0] = si; a[H] = tj; a[2 * H] = sf; a[3 * H] = so;
        }
        if (d.c_raw) d.c_raw[i] = c;
    }
}

__global__ void lstm_point_bwd_kernel(mstts_lstm_point_bwd_desc d) {
    const long n = d.B * d.H;
    const int H = (int)d.H;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const int b = (int)(i / H), u = (int)(i % H);
        int len = d.lengths ? d.lengths[b] : 0x7fffffff;
        const bool live = d.step < len;
        const int pos = d.reverse ? (live ? len - 1 - d.step : d.step) : d.step;
        float dhs = d.d_h_state[i];
        if (d.d_h_state2) {
            dhs += sum_parts<MSTTS_MAX_PARTS>(d.d_h_state2, d.dhs2_parts, d.dhs2_pstride, b * d.dhs2_ld + u);
        }
        const float dcs = d.d_c_state[i];
        float* dg = d.dgates + (long)b * 4 * H + u;
        float* dgp = d.dgates_pos ? d.dgates_pos + b * d.dgp_sb + pos * d.dgp_st + u : nullptr;
        if (!live) {
            dg[0] = dg[H] = dg[2 * H] = dg[3 * H] = 0.f;
            if (dgp) { dgp[0] = dgp[H] = dgp[2 * H] = dgp[3 * H] = 0.f; }
            d.d_c_prev[i] = dcs; d.d_h_prev[i] = dhs;
            continue;
        }
        float dm = 0.f;
        if (d.d_out) {
            dm += sum_parts<MSTTS_MAX_PARTS>(d.d_out, d.dout_parts, d.dout_pstride, b * d.dout_sb + pos * d.dout_st + u);
        }
        if (d.d_out2) {
            dm += sum_parts<MSTTS_MAX_PARTS>(d.d_out2, d.dout2_parts, d.dout2_pstride, i);
        }
        const float kz = 1.f - d.zoneout;
        const float mh = d.zh ? (d.zh[i] ? kz : 0.f) : kz;     // d h'/d m
        const float mc = d.zc ? (d.zc[i] ? kz : 0.f) : kz;     // d c'/d c
        dm += mh * dhs;
        const float* a = d.acts + (long)b * 4 * H + u;
        const float si = a[0], tj = a[H], sf = a[2 * H], so = a[3 * H];
        const float c = d.c_raw[i], cp = d.c_prev[i];
        const float tc = tanhf_(c);
        const float dc = dm * so * (1.f - tc * tc) + mc * dcs;
        const float d_o = dm * tc * so * (1.f - so);
        const float d_i = dc * tj * si * (1.f - si);
        const float d_j = dc * si * (1.f - tj * tj);
        const float d_f = dc * cp * sf * (1.f - sf);
        dg[0] = d_i; dg[H] = d_j; dg[2 * H] = d_f; dg[3 * H] = d_o;
        if (dgp) { dgp[0] = d_i; dgp[H] = d_j; dgp[2 * H] = d_f; dgp[3 * H] = d_o; }
        d.d_c_prev[i] = dcs * (1.f - mc) + dc * sf;
        d.d_h_prev[i] = dhs * (1.f - mh);
    }
}

// ---------------------------------------------------------------------------------------------
// lean decoder-path variants of the two kernels above (no sequence-length masking, no residual, no
// position indirection, 32-bit indexing, compile-time slab count): the generic kernels cost ~1900
// instructions per thread in address arithmetic alone, which is what a 32K-element step pays for.
// ---------------------------------------------------------------------------------------------
struct PointFwdFast {
    const float* gates; int pstride;            // [PARTS][B][4H]
    const float* xw; int xw_ld, xw_st;          // row b, position pos at xw + b*xw_ld + pos*xw_st (or null)
    const float* bias;                          // [4H] or null
    const float* c_prev; const float* h_prev; int h_prev_ld;
    const uint8_t* zc; const uint8_t* zh; float keep;
    float* out; int out_ld, out_st;
    float* c_next; float* h_next; int h_next_ld;
    float* acts; float* c_raw;                  // may be null when SEQ
    const int32_t* lengths; int step, reverse;  // SEQ only
    const float* residual; int res_ld, res_st;  // SEQ only
    int B, H;
};

// SEQ = dynamic_rnn semantics (length masking, reversed direction, residual wrapper); the decoder uses SEQ = false
template <int PARTS, bool SEQ>
__global__ __launch_bounds__(128) void lstm_point_fwd_fast_kernel(PointFwdFast d) {
    const int i = blockIdx.x * 128 + threadIdx.x;
    const int H = d.H;
    if (i >= d.B * H) return;
    const int b = i / H, u = i - b * H;
    const int g0 = b * 4 * H + u;
    int pos = 0;
    bool live = true;
    if (SEQ) {
        const int len = d.lengths ? d.lengths[b] : 0x7fffffff;
        live = d.step < len;
        pos = (d.reverse && live) ? len - 1 - d.step : d.step;
    }
    const float cp = d.c_prev[i], hp = d.h_prev[b * d.h_prev_ld + u];
    if (SEQ && !live) {
        d.out[b * d.out_ld + pos * d.out_st + u] = 0.f;
        d.c_next[i] = cp; d.h_next[b * d.h_next_ld + u] = hp;
        if (d.acts) { float* a = d.acts + g0; a[0] = 0.f; a[H] = 0.f; a[2 * H] = 0.f; a[3 * H] = 0.f; }
        if (d.c_raw) d.c_raw[i] = cp;
        return;
    }
    float g4[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int pp = 0; pp < PARTS; ++pp) {
#pragma unroll
        for (int g = 0; g < 4; ++g) g4[g] += d.gates[pp * d.pstride + g0 + g * H];
    }
    if (d.xw) {
#pragma unroll
        for (int g = 0; g < 4; ++g) g4[g] += d.xw[b * d.xw_ld + pos * d.xw_st + g * H + u];
    }
    if (d.bias) {
#pragma unroll
        for (int g = 0; g < 4; ++g) g4[g] += d.bias[g * H + u];
    }
    const float kc = d.zc ? (d.zc[i] ? d.keep : 0.f) : d.keep;
    const float kh = d.zh ? (d.zh[i] ? d.keep : 0.f) : d.keep;
    const float si = sigmoidf_(g4[0]), tj = tanhf_(g4[1]), sf = sigmoidf_(g4[2] + 1.0f), so = sigmoidf_(g4[3]);
    const float c = sf * cp + si * tj;
    const float m = so * tanhf_(c);
    d.c_next[i] = kc * (c - cp) + cp;
    d.h_next[b * d.h_next_ld + u] = kh * (m - hp) + hp;
    float o = m;
    if (SEQ && d.residual) o += d.residual[b * d.res_ld + pos * d.res_st + u];
    d.out[b * d.out_ld + pos * d.out_st + u] = o;
    if (!SEQ || d.acts) { float* a = d.acts + g0; a[0] = si; a[H] = tj; a[2 * H] = sf; a[3 * H] = so; }
    if (!SEQ || d.c_raw) d.c_raw[i] = c;
}

struct PointBwdFast {
    const float* d_out; int dout_ld, dout_st, dout_parts, dout_pstride;     // may be null
    const float* d_out2; int dout2_parts, dout2_pstride;           // [parts][B][H] or null
    const float* d_c_state; const float* d_h_state;
    const float* dhs2; int dhs2_ld, dhs2_parts; long dhs2_pstride; // or null
    const float* acts; const float* c_raw; const float* c_prev;
    const uint8_t* zc; const uint8_t* zh; float keep;
    float* dgates; float* d_c_prev; float* d_h_prev;
    const int32_t* lengths; int step, reverse;                     // SEQ only
    float* dgates_pos; int dgp_ld, dgp_st;                         // SEQ only (or null)
    const float* dq; const float* wq_t;                            // FUSE_Q only: [B][QA], [QA/4][H][4]
    int B, H;
    int row2d;                                                     // grid (H / 128, B): the row is blockIdx.y - no division in front of the loads
};
constexpr int QA = 128;        // attention units of the fused query-layer gradient (hp.Attention.Memory_Size)
__device__ const float pb_zero[1] = {0.f};        // stand-ins for optional operands of the pointwise kernels (see the load block below)
__device__ const uint8_t pb_one[1] = {1};

// FUSE_Q: the output gradient also gets dq[b,:] . Wq[u,:] (the attention query layer's data gradient, q = m1 . Wq), computed here from
// the transposed kernel instead of by a product launch of its own; all QA loads of a thread are issued before the first FMA
struct PointBwdFastPair { PointBwdFast d[2]; };      // two independent cells per launch (blockIdx.y): the two BiLSTM directions

template <int P_OUT, int P_OUT2, int P_DHS, bool SEQ, int FUSE_Q = 0>        // FUSE_Q: 0 off, 1 fp32, 2 bf16-rounded operands
__device__ __forceinline__ void lstm_point_bwd_fast_body(const PointBwdFast& d) {
    const int H = d.H;
    // fused form: H % 128 == 0 (checked by the host), so the 128 threads of a block share one row - a block-uniform index lets the dq
    // row come through scalar loads
    int i, b, u;
    if (d.row2d) { b = blockIdx.y; u = blockIdx.x * 128 + threadIdx.x; i = b * H + u; }
    else {
        i = blockIdx.x * 128 + threadIdx.x;
        if (i >= d.B * H) return;
        b = FUSE_Q ? (int)(blockIdx.x * 128) / H : i / H; u = i - b * H;
    }
    int pos = 0;
    bool live = true;
    if (SEQ) {
        const int len = d.lengths ? d.lengths[b] : 0x7fffffff;
        live = d.step < len;
        pos = (d.reverse && live) ? len - 1 - d.step : d.step;
    }
    float4 wqv[FUSE_Q ? QA / 4 : 1], dqv[1];
    __shared__ float4 s_dq[QA / 4];
    if (FUSE_Q) {                 // wq_t is [QA/4][H][4]: consecutive units are consecutive float4 - 1 KB per wave load
#pragma unroll
        for (int a = 0; a < QA / 4; ++a) wqv[a] = reinterpret_cast<const float4*>(d.wq_t)[a * H + u];
        // the row's dq (block-uniform, 128 floats) goes through LDS: as scalar loads it needs 128 SGPRs at once - 79 of them spilled
        // to VGPR lanes (writelane / readlane pairs) with a wait per batch
        if (threadIdx.x < QA / 4) s_dq[threadIdx.x] = reinterpret_cast<const float4*>(d.dq)[b * (QA / 4) + threadIdx.x];
    }
    // ---- every operand of the element, requested back to back.  Optional operands (absent slabs, absent masks) are read from a
    // constant block through a selected pointer rather than skipped: a null check is a branch, and the kernel used to be seven basic
    // blocks each ending in its own wait - seven dependent memory round trips for 30 floats.
    const float* p_dhs2 = d.dhs2 ? d.dhs2 + b * d.dhs2_ld + u : pb_zero;
    const long s_dhs2 = d.dhs2 ? d.dhs2_pstride : 0;
    const float* p_do = d.d_out ? d.d_out + b * d.dout_ld + pos * d.dout_st + u : pb_zero;
    const long s_do = d.d_out ? d.dout_pstride : 0;
    const float* p_do2 = d.d_out2 ? d.d_out2 + i : pb_zero;
    const long s_do2 = d.d_out2 ? d.dout2_pstride : 0;
    const uint8_t* p_zh = d.zh ? d.zh + i : pb_one;
    const uint8_t* p_zc = d.zc ? d.zc + i : pb_one;
    const float r_dhs = d.d_h_state[i], dcs = d.d_c_state[i];
    float r_dhs2[P_DHS], r_do[P_OUT], r_do2[P_OUT2];
#pragma unroll
    for (int pp = 0; pp < P_DHS; ++pp) r_dhs2[pp] = p_dhs2[pp * s_dhs2];
#pragma unroll
    for (int pp = 0; pp < P_OUT; ++pp) r_do[pp] = p_do[pp * s_do];
#pragma unroll
    for (int pp = 0; pp < P_OUT2; ++pp) r_do2[pp] = p_do2[pp * s_do2];
    const uint8_t r_zh = *p_zh, r_zc = *p_zc;
    const float* a = d.acts + b * 4 * H + u;
    const float si = a[0], tj = a[H], sf = a[2 * H], so = a[3 * H];
    const float c = d.c_raw[i], cp = d.c_prev[i];
    __builtin_amdgcn_sched_barrier(0);
    float dhs = r_dhs;
#pragma unroll
    for (int pp = 0; pp < P_DHS; ++pp) dhs += r_dhs2[pp];
    float* dg = d.dgates + b * 4 * H + u;
    float* dgp = (SEQ && d.dgates_pos) ? d.dgates_pos + b * d.dgp_ld + pos * d.dgp_st + u : nullptr;
    if (SEQ && !live) {
        dg[0] = 0.f; dg[H] = 0.f; dg[2 * H] = 0.f; dg[3 * H] = 0.f;
        if (dgp) { dgp[0] = 0.f; dgp[H] = 0.f; dgp[2 * H] = 0.f; dgp[3 * H] = 0.f; }
        d.d_c_prev[i] = dcs; d.d_h_prev[i] = dhs;
        return;
    }
    float dm = 0.f;
#pragma unroll
    for (int pp = 0; pp < P_OUT; ++pp) dm += r_do[pp];
#pragma unroll
    for (int pp = 0; pp < P_OUT2; ++pp) dm += r_do2[pp];
    if (FUSE_Q) {
        __syncthreads();
        float q0 = 0.f, q1 = 0.f, q2 = 0.f, q3 = 0.f;
#pragma unroll
        for (int a_ = 0; a_ < QA / 4; ++a_) {
            dqv[0] = s_dq[a_];
            if (FUSE_Q == 2) {
                auto r = [](float x) { return (float)(__bf16)x; };
                q0 += r(dqv[0].x) * r(wqv[a_].x); q1 += r(dqv[0].y) * r(wqv[a_].y); q2 += r(dqv[0].z) * r(wqv[a_].z); q3 += r(dqv[0].w) * r(wqv[a_].w);
            } else { q0 += dqv[0].x * wqv[a_].x; q1 += dqv[0].y * wqv[a_].y; q2 += dqv[0].z * wqv[a_].z; q3 += dqv[0].w * wqv[a_].w; }
        }
        dm += (q0 + q1) + (q2 + q3);
    }
    const float mh = r_zh ? d.keep : 0.f;
    const float mc = r_zc ? d.keep : 0.f;
    dm += mh * dhs;
    const float tc = tanhf_(c);
    const float dc = dm * so * (1.f - tc * tc) + mc * dcs;
    const float d_i = dc * tj * si * (1.f - si), d_j = dc * si * (1.f - tj * tj), d_f = dc * cp * sf * (1.f - sf), d_o = dm * tc * so * (1.f - so);
    dg[0] = d_i; dg[H] = d_j; dg[2 * H] = d_f; dg[3 * H] = d_o;
    if (dgp) { dgp[0] = d_i; dgp[H] = d_j; dgp[2 * H] = d_f; dgp[3 * H] = d_o; }
    d.d_c_prev[i] = dcs * (1.f - mc) + dc * sf;
    d.d_h_prev[i] = dhs * (1.f - mh);
}
template <int P_OUT, int P_OUT2, int P_DHS, bool SEQ, int FUSE_Q = 0>
__global__ __launch_bounds__(128) void lstm_point_bwd_fast_kernel(PointBwdFast d) { lstm_point_bwd_fast_body<P_OUT, P_OUT2, P_DHS, SEQ, FUSE_Q>(d); }
template <int P_DHS>
__global__ __launch_bounds__(128) void lstm_point_bwd_fast_pair_kernel(PointBwdFastPair p) { lstm_point_bwd_fast_body<1, 1, P_DHS, true, 0>(p.d[blockIdx.y]); }

// ---------------------------------------------------------------------------------------------
// losses
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void tts_loss_kernel(const float* __restrict__ lin, const float* __restrict__ post,
                                                       const float* __restrict__ mel, const float* __restrict__ stop,
                                                       const int32_t* __restrict__ mlen, long B, long S, long nm, int use_l1,
                                                       float gs, float* __restrict__ scal, float* __restrict__ dlin,
                                                       float* __restrict__ dpost, float* __restrict__ dstop) {
    __shared__ float scratch[16];
    const long L = S - 1;
    const long n_mel_el = B * S * nm;
    const float inv_n = 1.f / (float)(B * L * nm);
    float a_lin = 0.f, a_post = 0.f, a_stop = 0.f;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n_mel_el; i += (long)gridDim.x * blockDim.x) {
        const long c = i % nm, t = (i / nm) % S, b = i / (nm * S);
        float gl = 0.f, gp = 0.f;
        if (t < L) {
            const float tgt = mel[(b * L + t) * nm + c];
            const float e1 = lin[i] - tgt, e2 = post[i] - tgt;
            a_lin += e1 * e1; a_post += e2 * e2;
            gl = 2.f * e1 * inv_n; gp = 2.f * e2 * inv_n;
            if (use_l1) {
                a_lin += fabsf(e1); a_post += fabsf(e2);
                gl += (e1 > 0.f ? 1.f : (e1 < 0.f ? -1.f : 0.f)) * inv_n;
                gp += (e2 > 0.f ? 1.f : (e2 < 0.f ? -1.f : 0.f)) * inv_n;
            }
        }
        dlin[i] = gl * gs; dpost[i] = gp * gs;
    }
    const float inv_s = 1.f / (float)(B * S);
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < B * S; i += (long)gridDim.x * blockDim.x) {
        const long t = i % S, b = i / S;
        const float z = stop[i];
        const float y = (t >= mlen[b]) ? 1.f : 0.f;
        a_stop += fmaxf(z, 0.f) - z * y + log1pf(__expf(-fabsf(z)));
        dstop[i] = (sigmoidf_(z) - y) * inv_s * gs;
    }
    a_lin = block_sum(a_lin, scratch);
    a_post = block_sum(a_post, scratch);
    a_stop = block_sum(a_stop, scratch);
    if (threadIdx.x == 0) {
        atomicAdd(scal + 0, a_lin * inv_n);
        atomicAdd(scal + 1, a_post * inv_n);
        atomicAdd(scal + 2, a_stop * inv_s);
    }
}

__global__ void shift_frames_kernel(const float* __restrict__ mel, float* __restrict__ fr, long B, long L, long C) {
    const long n = (L + 1) * B * C;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const long c = i % C, b = (i / C) % B, s = i / (C * B);
        fr[i] = (s == 0) ? 0.f : mel[(b * L + (s - 1)) * C + c];
    }
}
__global__ void unpack_proj_kernel(const float* __restrict__ proj, long ldp, float* __restrict__ lin, float* __restrict__ stop,
                                   long B, long S, long C) {
    const long n = B * S * (C + 1);
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const long c = i % (C + 1), s = (i / (C + 1)) % S, b = i / ((C + 1) * S);
        const float v = proj[(s * B + b) * ldp + c];
        if (c < C) lin[(b * S + s) * C + c] = v; else stop[b * S + s] = v;
    }
}
__global__ void pack_dproj_kernel(const float* __restrict__ dlin, const float* __restrict__ dstop, float* __restrict__ dproj,
                                  long ldp, long B, long S, long C) {
    const long n = S * B * ldp;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const long c = i % ldp, b = (i / ldp) % B, s = i / (ldp * B);
        float v = 0.f;
        if (c < C) v = dlin[(b * S + s) * C + c]; else if (c == C) v = dstop[b * S + s];
        dproj[i] = v;
    }
}
__global__ void speaker_tile_kernel(const float* __restrict__ spk, const int32_t* __restrict__ len, float* __restrict__ values,
                                    long B, long T, long M, long off, long width) {
    const long n = B * T * width;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const long j = i % width, t = (i / width) % T, b = i / (width * T);
        const bool live = len ? (t < len[b]) : true;
        values[(b * T + t) * M + off + j] = live ? spk[b * width + j] : 0.f;
    }
}
__global__ void conv_kernel_flip_kernel(const float* __restrict__ w, float* __restrict__ wt, long K, long Cin, long Cout) {
    const long n = K * Cin * Cout;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const long c = i % Cin, o = (i / Cin) % Cout, kk = i / (Cin * Cout);
        wt[i] = w[((K - 1 - kk) * Cin + c) * Cout + o];
    }
}

__global__ void transpose01_kernel(const float* __restrict__ src, float* __restrict__ dst, long D0, long D1, long C) {
    const long n = D0 * D1 * C;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const long c = i % C, d1 = (i / C) % D1, d0 = i / (C * D1);
        dst[(d1 * D0 + d0) * C + c] = src[i];
    }
}
__global__ __launch_bounds__(1024) void speaker_finalize_kernel(const float* __restrict__ x, float* __restrict__ out, int B, int S, long T, int E) {
    __shared__ float scratch[16];
    const int n = B * E;
    float ss = 0.f;
    for (int i = threadIdx.x; i < n; i += blockDim.x) {
        const int b = i / E, e = i % E;
        float a = 0.f;
        for (int k = 0; k < S; ++k) a += x[(((long)b * S + k) * T + (T - 1)) * E + e];
        a /= (float)S;
        out[i] = a;
        ss += a * a;
    }
    ss = block_sum(ss, scratch);
    const float inv = rsqrtf(fmaxf(ss, 1e-12f));
    __syncthreads();
    for (int i = threadIdx.x; i < n; i += blockDim.x) out[i] *= inv;
}

__global__ __launch_bounds__(256) void l2_loss_kernel(const float* __restrict__ x, const uint8_t* __restrict__ mask, long n, float* __restrict__ out) {
    __shared__ float scratch[16];
    float a = 0.f;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x)
        if (!mask || mask[i]) a += x[i] * x[i];
    a = block_sum(a, scratch);
    if (threadIdx.x == 0) atomicAdd(out, 0.5f * a);
}

__global__ void adam_tf_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v,
                               const uint8_t* __restrict__ wdm, float wd, float gs, float lr_t, float b1, float b2, float eps, long n) {
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const float pv = p[i];
        float gv = g[i] * gs;
        if (wdm && wdm[i]) gv += wd * pv;
        const float mv = b1 * m[i] + (1.f - b1) * gv;
        const float vv = b2 * v[i] + (1.f - b2) * gv * gv;
        m[i] = mv; v[i] = vv;
        p[i] = pv - lr_t * mv / (sqrtf(vv) + eps);
    }
}

// Diagnostic: `n` workgroups that each claim a whole CU's worth of LDS (96 KB: no persistent workgroup fits beside one) and sit on it for
// `ticks` of the 100 MHz wall clock - a stand-in for another tenant of the chip (a collective that outlives its slot, a second process)
// in the tests of the persistent launches' co-residency rendezvous.
__global__ __launch_bounds__(64) void park_cus_kernel(unsigned long long ticks, unsigned* out) {
    extern __shared__ float park_lds[];
    park_lds[threadIdx.x] = 1.f;
    const unsigned long long t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(32);
    if (out && threadIdx.x == 0 && park_lds[0] > 0.f) atomicAdd(out, 1u);
}

}  // namespace mstts

using namespace mstts;
#define ST(s) ((hipStream_t)(s))

extern "C" const char* mstts_last_error(void) { return err_buf(); }
extern "C" int mstts_abi_version(void) { return 5; }

extern "C" int mstts_debug_park_cus(int32_t n_workgroups, int64_t microseconds, uint32_t* done_count, mstts_stream_t s) {
    MSTTS_REQUIRE(n_workgroups >= 1 && n_workgroups <= 1024 && microseconds >= 0 && microseconds <= 1000000, MSTTS_ERR_SHAPE, "debug_park_cus: 1..1024 workgroups, at most 1 s");
    const size_t lds = 96 * 1024;
    static int memo[64];
    int dev = 0;
    if (hipGetDevice(&dev) == hipSuccess && dev >= 0 && dev < 64 && !memo[dev]) {
        if (hipFuncSetAttribute((const void*)park_cus_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
            return set_err(MSTTS_ERR_LAUNCH, "debug_park_cus: the device does not take a 96 KB workgroup");
        memo[dev] = 1;
    }
    hipLaunchKernelGGL(park_cus_kernel, dim3((unsigned)n_workgroups), dim3(64), lds, ST(s), (unsigned long long)microseconds * 100ull, done_count);
    MSTTS_CHECK_LAUNCH("park_cus");
    return MSTTS_OK;
}

extern "C" int mstts_philox_keep_mask(uint8_t* out, int64_t n, uint64_t seed, uint32_t stream_id, float keep_prob, mstts_stream_t s) {
    MSTTS_REQUIRE(n >= 0 && (n == 0 || out), MSTTS_ERR_SHAPE, "philox: bad args");
    if (n == 0) return MSTTS_OK;
    MSTTS_REQUIRE((reinterpret_cast<uintptr_t>(out) & 3u) == 0, MSTTS_ERR_ALIGN, "philox: out must be 4-byte aligned");
    hipLaunchKernelGGL(philox_mask_kernel, dim3(grid_for((n + 3) / 4, 256)), dim3(256), 0, ST(s), out, (long)n,
                       (uint32_t)(seed & 0xFFFFFFFFu), (uint32_t)(seed >> 32), stream_id, keep_prob);
    MSTTS_CHECK_LAUNCH("philox_keep_mask");
    return MSTTS_OK;
}

extern "C" int mstts_philox_keep_mask_rows(uint8_t* out, int64_t outer, int64_t B, int64_t inner, uint64_t seed, uint32_t stream_id,
                                           uint64_t sample0, float keep_prob, mstts_stream_t s) {
    MSTTS_REQUIRE(outer >= 0 && B >= 0 && inner >= 0 && (outer * B * inner == 0 || out) && B < (1LL << 31), MSTTS_ERR_SHAPE, "philox rows: bad args");
    MSTTS_REQUIRE(inner % 4 != 0 || ((uintptr_t)out & 3) == 0, MSTTS_ERR_ALIGN, "philox rows: 4-byte aligned buffer required");
    if (outer * B * inner == 0) return MSTTS_OK;
    const long total = ((outer * inner + 3) / 4) * B;
    hipLaunchKernelGGL(philox_mask_rows_kernel, dim3(grid_for(total, 256)), dim3(256), 0, ST(s), out, (long)outer, (int)B, (long)inner,
                       (uint32_t)seed, (uint32_t)(seed >> 32), stream_id, (uint32_t)sample0, keep_prob);
    MSTTS_CHECK_LAUNCH("philox_keep_mask_rows");
    return MSTTS_OK;
}

extern "C" int mstts_embedding_fwd(const int32_t* token, const float* table, float* out, int64_t n, int64_t vocab, int64_t width, mstts_stream_t s) {
    MSTTS_REQUIRE(width % 4 == 0 && aligned16(table) && aligned16(out), MSTTS_ERR_ALIGN, "embedding: width %% 4 and 16-byte alignment required");
    if (n == 0) return MSTTS_OK;
    hipLaunchKernelGGL(embedding_fwd_kernel, dim3(grid_for(n * width / 4, 256)), dim3(256), 0, ST(s), token, table, out, (long)n, (int)vocab, (int)width);
    MSTTS_CHECK_LAUNCH("embedding_fwd");
    return MSTTS_OK;
}
namespace mstts { int gemm_deterministic_now(); }       // csrc/gemm.hip: the calling thread asked for fixed summation orders (mstts_gemm_deterministic)
extern "C" int mstts_embedding_bwd(const int32_t* token, const float* dout, float* dtable, int64_t n, int64_t vocab, int64_t width, mstts_stream_t s) {
    if (n == 0) return MSTTS_OK;
    if (gemm_deterministic_now()) {
        hipLaunchKernelGGL(embedding_bwd_det_kernel, dim3(cdiv(width, 64)), dim3(64), 0, ST(s), token, dout, dtable, (long)n, (int)vocab, (int)width);
        MSTTS_CHECK_LAUNCH("embedding_bwd (fixed order)");
        return MSTTS_OK;
    }
    hipLaunchKernelGGL(embedding_bwd_kernel, dim3(grid_for(n * width, 256)), dim3(256), 0, ST(s), token, dout, dtable, (long)n, (int)vocab, (int)width);
    MSTTS_CHECK_LAUNCH("embedding_bwd");
    return MSTTS_OK;
}

// x[rows, C] (row stride ld) column sums: float4 form when the layout allows, else the scalar kernel
static void launch_col_stats(const float* x, const float* dy, const uint8_t* mask, float inv_keep, const float* mean, const float* rstd,
                             long rows, int C, long ld, int mode, float* out0, float* out1, hipStream_t st);

static int rows_per_block_for(int64_t rows, int64_t C) {
    // aim for ~1024 blocks
    long colb = (C + 63) / 64;
    long want = 1024 / (colb > 0 ? colb : 1);
    if (want < 1) want = 1;
    long rpb = (rows + want - 1) / want;
    if (rpb < 32) rpb = 32;
    return (int)rpb;
}

static void launch_col_stats(const float* x, const float* dy, const uint8_t* mask, float inv_keep, const float* mean, const float* rstd,
                             long rows, int C, long ld, int mode, float* out0, float* out1, hipStream_t st) {
    if (gemm_deterministic_now()) {
        // fixed summation order: ONE workgroup per 64 columns walks all the rows (4 row lanes, combined in a fixed order), so every output
        // element receives exactly one add - slow (C / 64 workgroups), reproducible bit for bit
        dim3 grid(cdiv(C, 64), 1);
        hipLaunchKernelGGL(col_stats_kernel, grid, dim3(256), 0, st, x, dy, mask, inv_keep, mean, rstd, rows, C, ld, (int)(rows < (1L << 30) ? rows : (1L << 30)), mode, out0, out1);
        return;
    }
    const bool v4 = C % 4 == 0 && ld % 4 == 0 && aligned16(x) && (mode == 0 || (aligned16(dy) && aligned16(mean) && aligned16(rstd) &&
                                                                                 (!mask || ((uintptr_t)mask & 3) == 0)));
    if (v4) {
        const long colb = (C + 255) / 256;
        long want = 512 / colb;                      // ~512 workgroups: 2 per CU, 8 rows x 1 KB in flight per wave
        if (want < 1) want = 1;
        long rpb = (rows + want - 1) / want;
        if (rpb < 32) rpb = 32;
        dim3 grid((unsigned)colb, (unsigned)cdiv(rows, rpb));
        if (mode == 0) hipLaunchKernelGGL((col_stats4_kernel<0, 8>), grid, dim3(256), 0, st, x, dy, mask, inv_keep, mean, rstd, rows, C, ld, (int)rpb, out0, out1);
        else hipLaunchKernelGGL((col_stats4_kernel<1, 4>), grid, dim3(256), 0, st, x, dy, mask, inv_keep, mean, rstd, rows, C, ld, (int)rpb, out0, out1);
        return;
    }
    const int rpb = rows_per_block_for(rows, C);
    dim3 grid(cdiv(C, 64), cdiv(rows, rpb));
    hipLaunchKernelGGL(col_stats_kernel, grid, dim3(256), 0, st, x, dy, mask, inv_keep, mean, rstd, rows, C, ld, rpb, mode, out0, out1);
}

extern "C" int mstts_colsum(const float* x, int64_t rows, int64_t C, int64_t ld, float* out, int32_t accumulate, mstts_stream_t s) {
    if (!accumulate) hipMemsetAsync(out, 0, C * sizeof(float), ST(s));
    if (rows == 0 || C == 0) return MSTTS_OK;
    launch_col_stats(x, nullptr, nullptr, 1.f, nullptr, nullptr, (long)rows, (int)C, (long)ld, 0, out, nullptr, ST(s));
    MSTTS_CHECK_LAUNCH("colsum");
    return MSTTS_OK;
}

extern "C" int mstts_bn_train_fwd(const float* x, const float* gamma, const float* beta, float* moving_mean, float* moving_var,
                                  float* y, float* save_mean, float* save_rstd, const uint8_t* keep_mask, float keep_prob,
                                  float momentum, float eps, int64_t rows, int64_t C, float* ws, mstts_stream_t s) {
    MSTTS_REQUIRE(C % 4 == 0 && aligned16(x) && aligned16(y), MSTTS_ERR_ALIGN, "bn: C %% 4 and 16-byte alignment required");
    MSTTS_REQUIRE(rows > 0, MSTTS_ERR_SHAPE, "bn: rows must be > 0");
    hipMemsetAsync(ws, 0, 2 * C * sizeof(float), ST(s));
    launch_col_stats(x, nullptr, nullptr, 1.f, nullptr, nullptr, (long)rows, (int)C, (long)C, 0, ws, ws + C, ST(s));
    hipLaunchKernelGGL(bn_finalize_kernel, dim3(cdiv(C, 256)), dim3(256), 0, ST(s), ws, 1.f / (float)rows, eps, momentum,
                       moving_mean, moving_var, save_mean, save_rstd, (int)C);
    hipLaunchKernelGGL(bn_apply_kernel, dim3(grid_for(rows * C / 4, 256)), dim3(256), 0, ST(s), x, gamma, beta,
                       (const float*)save_mean, (const float*)save_rstd, 0, eps, keep_mask, 1.f / keep_prob, y, (long)rows, (int)C);
    MSTTS_CHECK_LAUNCH("bn_train_fwd");
    return MSTTS_OK;
}

extern "C" int mstts_bn_infer_fwd(const float* x, const float* gamma, const float* beta, const float* moving_mean,
                                  const float* moving_var, float* y, float eps, int64_t rows, int64_t C, mstts_stream_t s) {
    MSTTS_REQUIRE(C % 4 == 0 && aligned16(x) && aligned16(y), MSTTS_ERR_ALIGN, "bn: C %% 4 and 16-byte alignment required");
    if (rows == 0) return MSTTS_OK;
    hipLaunchKernelGGL(bn_apply_kernel, dim3(grid_for(rows * C / 4, 256)), dim3(256), 0, ST(s), x, gamma, beta, moving_mean,
                       moving_var, 1, eps, (const uint8_t*)nullptr, 1.f, y, (long)rows, (int)C);
    MSTTS_CHECK_LAUNCH("bn_infer_fwd");
    return MSTTS_OK;
}

extern "C" int mstts_bn_train_bwd(const float* dy, const float* x, const float* gamma, const float* save_mean, const float* save_rstd,
                                  const uint8_t* keep_mask, float keep_prob, int32_t act, float* dz, float* dgamma, float* dbeta,
                                  float* dbias, int64_t rows, int64_t C, float* ws, mstts_stream_t s) {
    MSTTS_REQUIRE(C % 4 == 0 && aligned16(x) && aligned16(dy) && aligned16(dz), MSTTS_ERR_ALIGN, "bn: C %% 4 and 16-byte alignment required");
    MSTTS_REQUIRE(rows > 0, MSTTS_ERR_SHAPE, "bn: rows must be > 0");
    hipMemsetAsync(ws, 0, 2 * C * sizeof(float), ST(s));
    launch_col_stats(x, dy, keep_mask, 1.f / keep_prob, save_mean, save_rstd, (long)rows, (int)C, (long)C, 1, ws, ws + C, ST(s));
    hipLaunchKernelGGL(bn_bwd_apply_kernel, dim3(grid_for(rows * C / 4, 256)), dim3(256), 0, ST(s), dy, x, gamma, save_mean, save_rstd,
                       keep_mask, 1.f / keep_prob, (int)act, (const float*)ws, 1.f / (float)rows, dz, (long)rows, (int)C);
    if (dbeta) hipLaunchKernelGGL(vec_acc_kernel, dim3(cdiv(C, 256)), dim3(256), 0, ST(s), dbeta, (const float*)ws, (int)C);
    if (dgamma) hipLaunchKernelGGL(vec_acc_kernel, dim3(cdiv(C, 256)), dim3(256), 0, ST(s), dgamma, (const float*)(ws + C), (int)C);
    if (dbias) {
        launch_col_stats(dz, nullptr, nullptr, 1.f, nullptr, nullptr, (long)rows, (int)C, (long)C, 0, dbias, nullptr, ST(s));
    }
    MSTTS_CHECK_LAUNCH("bn_train_bwd");
    return MSTTS_OK;
}

extern "C" int mstts_dropout(const float* x, const uint8_t* keep_mask, float keep_prob, float* y, int64_t n, mstts_stream_t s) {
    if (n == 0) return MSTTS_OK;
    hipLaunchKernelGGL(dropout_kernel, dim3(grid_for(n, 256)), dim3(256), 0, ST(s), x, keep_mask, 1.f / keep_prob, y, (long)n);
    MSTTS_CHECK_LAUNCH("dropout");
    return MSTTS_OK;
}
extern "C" int mstts_relu_dropout_bwd(const float* dy, const float* y_saved, const uint8_t* keep_mask, float keep_prob, float* dx, int64_t n, mstts_stream_t s) {
    if (n == 0) return MSTTS_OK;
    hipLaunchKernelGGL(relu_dropout_bwd_kernel, dim3(grid_for(n, 256)), dim3(256), 0, ST(s), dy, y_saved, keep_mask, 1.f / keep_prob, dx, (long)n);
    MSTTS_CHECK_LAUNCH("relu_dropout_bwd");
    return MSTTS_OK;
}
// ---- bf16 gradient exchange (BASELINE config 3: bf16 message, fp32 accumulation on receipt) ----------------------------------
namespace mstts {
__global__ void f32_to_bf16_kernel(const float* __restrict__ x, __bf16* __restrict__ y, long n) {
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) y[i] = (__bf16)x[i];   // RNE
}
__global__ void bf16_to_f32_kernel(const __bf16* __restrict__ x, float* __restrict__ y, long n) {
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) y[i] = (float)x[i];
}
// out[i] = bf16( sum_r float(chunks[r * stride + i]) ), the sum in fp32 in rank order (deterministic)
__global__ void bf16_chunks_sum_kernel(const __bf16* __restrict__ chunks, int nchunks, long stride, long n, __bf16* __restrict__ out) {
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        float a = 0.f;
        for (int r = 0; r < nchunks; ++r) a += (float)chunks[r * stride + i];
        out[i] = (__bf16)a;
    }
}
}  // namespace mstts
extern "C" int mstts_f32_to_bf16(const float* x, void* y, int64_t n, mstts_stream_t s) {
    if (n == 0) return MSTTS_OK;
    MSTTS_REQUIRE(x && y, MSTTS_ERR_SHAPE, "f32_to_bf16: null pointer");
    hipLaunchKernelGGL(f32_to_bf16_kernel, dim3(grid_for(n, 256)), dim3(256), 0, ST(s), x, (__bf16*)y, (long)n);
    MSTTS_CHECK_LAUNCH("f32_to_bf16");
    return MSTTS_OK;
}
extern "C" int mstts_bf16_to_f32(const void* x, float* y, int64_t n, mstts_stream_t s) {
    if (n == 0) return MSTTS_OK;
    MSTTS_REQUIRE(x && y, MSTTS_ERR_SHAPE, "bf16_to_f32: null pointer");
    hipLaunchKernelGGL(bf16_to_f32_kernel, dim3(grid_for(n, 256)), dim3(256), 0, ST(s), (const __bf16*)x, y, (long)n);
    MSTTS_CHECK_LAUNCH("bf16_to_f32");
    return MSTTS_OK;
}
extern "C" int mstts_bf16_chunks_sum(const void* chunks, int32_t nchunks, int64_t stride, int64_t n, void* out, mstts_stream_t s) {
    if (n == 0) return MSTTS_OK;
    MSTTS_REQUIRE(chunks && out && nchunks >= 1 && stride >= n, MSTTS_ERR_SHAPE, "bf16_chunks_sum: bad arguments");
    hipLaunchKernelGGL(bf16_chunks_sum_kernel, dim3(grid_for(n, 256)), dim3(256), 0, ST(s), (const __bf16*)chunks, (int)nchunks, (long)stride, (long)n,
                       (__bf16*)out);
    MSTTS_CHECK_LAUNCH("bf16_chunks_sum");
    return MSTTS_OK;
}

extern "C" int mstts_add(const float* a, const float* b, float* y, int64_t n, mstts_stream_t s) {
    if (n == 0) return MSTTS_OK;
    hipLaunchKernelGGL(add_kernel, dim3(grid_for(n, 256)), dim3(256), 0, ST(s), a, b, y, (long)n);
    MSTTS_CHECK_LAUNCH("add");
    return MSTTS_OK;
}
extern "C" int mstts_fill(float* y, float v, int64_t n, mstts_stream_t s) {
    if (n == 0) return MSTTS_OK;
    const long n4 = aligned16(y) ? n / 4 : 0;
    if (n4 > 0) hipLaunchKernelGGL(fill4_kernel, dim3(grid_for(n4, 256)), dim3(256), 0, ST(s), reinterpret_cast<float4*>(y), v, n4);
    if (n - 4 * n4 > 0) hipLaunchKernelGGL(fill_kernel, dim3(grid_for(n - 4 * n4, 256)), dim3(256), 0, ST(s), y + 4 * n4, v, (long)(n - 4 * n4));
    MSTTS_CHECK_LAUNCH("fill");
    return MSTTS_OK;
}
extern "C" int mstts_persist_status(const uint32_t* ctrl_a, int32_t done_a, const uint32_t* ctrl_b, int32_t done_b, int32_t* flag, mstts_stream_t s) {
    MSTTS_REQUIRE(flag != nullptr, MSTTS_ERR_SHAPE, "persist_status: null flag");
    hipLaunchKernelGGL(persist_status_kernel, dim3(1), dim3(1), 0, ST(s), ctrl_a, (unsigned)done_a, ctrl_b, (unsigned)done_b, flag);
    MSTTS_CHECK_LAUNCH("persist_status");
    return MSTTS_OK;
}
extern "C" int mstts_copy2d(const float* src, int64_t lds, float* dst, int64_t ldd, int64_t rows, int64_t cols, int32_t accumulate, mstts_stream_t s) {
    if (rows * cols == 0) return MSTTS_OK;
    hipLaunchKernelGGL(copy2d_kernel, dim3(grid_for(rows * cols, 256)), dim3(256), 0, ST(s), src, (long)lds, dst, (long)ldd, (long)rows, (long)cols, (int)accumulate);
    MSTTS_CHECK_LAUNCH("copy2d");
    return MSTTS_OK;
}
extern "C" int mstts_maxpool2_same(const float* x, float* y, int64_t B, int64_t T, int64_t C, mstts_stream_t s) {
    if (B * T * C == 0) return MSTTS_OK;
    hipLaunchKernelGGL(maxpool2_kernel, dim3(grid_for(B * T * C, 256)), dim3(256), 0, ST(s), x, y, (long)B, (long)T, (long)C);
    MSTTS_CHECK_LAUNCH("maxpool2_same");
    return MSTTS_OK;
}
extern "C" int mstts_highway_combine(const float* h_pre, const float* t_pre, const float* x, float* y, int64_t n, mstts_stream_t s) {
    if (n == 0) return MSTTS_OK;
    hipLaunchKernelGGL(highway_kernel, dim3(grid_for(n, 256)), dim3(256), 0, ST(s), h_pre, t_pre, x, y, (long)n);
    MSTTS_CHECK_LAUNCH("highway_combine");
    return MSTTS_OK;
}
extern "C" int mstts_maxpool2_same_bwd(const float* x, const float* dy, float* dx, int64_t B, int64_t T, int64_t C, mstts_stream_t s) {
    if (B * T * C == 0) return MSTTS_OK;
    hipLaunchKernelGGL(maxpool2_bwd_kernel, dim3(grid_for(B * T * C, 256)), dim3(256), 0, ST(s), x, dy, dx, (long)B, (long)T, (long)C);
    MSTTS_CHECK_LAUNCH("maxpool2_same_bwd");
    return MSTTS_OK;
}
extern "C" int mstts_highway_combine_bwd(const float* h_pre, const float* t_pre, const float* x, const float* dy, float* dh_pre, float* dt_pre,
                                         float* dx, int64_t n, mstts_stream_t s) {
    if (n == 0) return MSTTS_OK;
    hipLaunchKernelGGL(highway_bwd_kernel, dim3(grid_for(n, 256)), dim3(256), 0, ST(s), h_pre, t_pre, x, dy, dh_pre, dt_pre, dx, (long)n);
    MSTTS_CHECK_LAUNCH("highway_combine_bwd");
    return MSTTS_OK;
}
extern "C" int mstts_l1_loss_fwd_bwd(const float* pred, const float* target, int64_t n, float* loss, float* d_pred, mstts_stream_t s) {
    MSTTS_REQUIRE(pred && target && loss && n >= 1, MSTTS_ERR_SHAPE, "l1_loss: bad arguments");
    hipError_t e = hipMemsetAsync(loss, 0, sizeof(float), ST(s));
    if (e != hipSuccess) return set_err(MSTTS_ERR_LAUNCH, "l1_loss: memset failed");
    hipLaunchKernelGGL(l1_loss_kernel, dim3(grid_for(n, 256)), dim3(256), 0, ST(s), pred, target, (long)n, 1.f / (float)n, loss, d_pred);
    MSTTS_CHECK_LAUNCH("l1_loss_fwd_bwd");
    return MSTTS_OK;
}
extern "C" int mstts_fold_rows(const float* src, float* dst, int64_t rows, int64_t cols, int64_t r0, int64_t n, mstts_stream_t s) {
    MSTTS_REQUIRE(r0 >= 0 && n >= 0 && r0 + 2 * n <= rows, MSTTS_ERR_SHAPE, "fold_rows: bad row ranges");
    hipLaunchKernelGGL(fold_rows_kernel, dim3(grid_for((rows - n) * cols, 256)), dim3(256), 0, ST(s), src, dst, (long)rows, (long)cols, (long)r0, (long)n);
    MSTTS_CHECK_LAUNCH("fold_rows");
    return MSTTS_OK;
}

extern "C" int mstts_lstm_point_fwd(const mstts_lstm_point_fwd_desc* d, mstts_stream_t s) {
    MSTTS_REQUIRE(d && d->gates_h && d->c_prev && d->h_prev && d->c_next && d->h_next, MSTTS_ERR_SHAPE, "lstm_point_fwd: null pointer");
    if (d->B * d->H == 0) return MSTTS_OK;
    const int parts = d->gates_parts > 1 ? d->gates_parts : 1;
    const bool seq = d->lengths || d->residual || d->reverse || d->xw_st != 0 || d->out_st != 0 || !d->acts_out || !d->c_raw;
    const long span = (d->B + 1) * (d->out_sb > d->xw_sb ? d->out_sb : d->xw_sb);
    const bool fast = d->out && d->B * d->H * 4 < (1LL << 30) && (parts == 1 || parts == 2 || parts == 4 || parts == 7 || parts == 8 || parts == 16) &&
                      (long)parts * d->gates_pstride < (1LL << 30) && span < (1LL << 30);
    if (fast) {
        PointFwdFast f;
        f.gates = d->gates_h; f.pstride = (int)d->gates_pstride; f.xw = d->xw; f.xw_ld = (int)d->xw_sb; f.xw_st = (int)d->xw_st; f.bias = d->bias;
        f.c_prev = d->c_prev; f.h_prev = d->h_prev; f.h_prev_ld = (int)(d->h_prev_ld ? d->h_prev_ld : d->H);
        f.zc = d->zc; f.zh = d->zh; f.keep = 1.f - d->zoneout;
        f.out = d->out; f.out_ld = (int)d->out_sb; f.out_st = (int)d->out_st; f.c_next = d->c_next; f.h_next = d->h_next;
        f.h_next_ld = (int)(d->h_next_ld ? d->h_next_ld : d->H); f.acts = d->acts_out; f.c_raw = d->c_raw;
        f.lengths = d->lengths; f.step = d->step; f.reverse = d->reverse;
        f.residual = d->residual; f.res_ld = (int)d->res_sb; f.res_st = (int)d->res_st;
        f.B = (int)d->B; f.H = (int)d->H;
        dim3 grid((unsigned)((d->B * d->H + 127) / 128));
#define MSTTS_PF(P)                                                                                        \
        if (seq) hipLaunchKernelGGL((lstm_point_fwd_fast_kernel<P, true>), grid, dim3(128), 0, ST(s), f);    \
        else hipLaunchKernelGGL((lstm_point_fwd_fast_kernel<P, false>), grid, dim3(128), 0, ST(s), f)
        if (parts == 16) { MSTTS_PF(16); } else if (parts == 8) { MSTTS_PF(8); } else if (parts == 7) { MSTTS_PF(7); } else if (parts == 4) { MSTTS_PF(4); }
        else if (parts == 2) { MSTTS_PF(2); } else { MSTTS_PF(1); }
#undef MSTTS_PF
        MSTTS_CHECK_LAUNCH("lstm_point_fwd_fast");
        return MSTTS_OK;
    }
    hipLaunchKernelGGL(lstm_point_fwd_kernel, dim3(grid_for(d->B * d->H, 256)), dim3(256), 0, ST(s), *d);
    MSTTS_CHECK_LAUNCH("lstm_point_fwd");
    return MSTTS_OK;
}
static void point_bwd_fill(const mstts_lstm_point_bwd_desc* d, PointBwdFast& f) {
    const int po = d->dout_parts > 1 ? d->dout_parts : 1, po2 = d->dout2_parts > 1 ? d->dout2_parts : 1, ph = d->dhs2_parts > 1 ? d->dhs2_parts : 1;
    f.dq = d->dq; f.wq_t = d->wq_t;
    f.d_out = d->d_out; f.dout_ld = (int)d->dout_sb; f.dout_st = (int)d->dout_st; f.dout_parts = po; f.dout_pstride = (int)d->dout_pstride;
    f.d_out2 = d->d_out2; f.dout2_parts = po2; f.dout2_pstride = (int)d->dout2_pstride;
    f.d_c_state = d->d_c_state; f.d_h_state = d->d_h_state;
    f.dhs2 = d->d_h_state2; f.dhs2_ld = (int)d->dhs2_ld; f.dhs2_parts = ph; f.dhs2_pstride = (long)d->dhs2_pstride;
    f.acts = d->acts; f.c_raw = d->c_raw; f.c_prev = d->c_prev; f.zc = d->zc; f.zh = d->zh; f.keep = 1.f - d->zoneout;
    f.dgates = d->dgates; f.d_c_prev = d->d_c_prev; f.d_h_prev = d->d_h_prev;
    f.lengths = d->lengths; f.step = d->step; f.reverse = d->reverse;
    f.dgates_pos = d->dgates_pos; f.dgp_ld = (int)d->dgp_sb; f.dgp_st = (int)d->dgp_st;
    f.B = (int)d->B; f.H = (int)d->H; f.row2d = 0;
}

/* the pointwise backward of two independent cells of the same shape in ONE launch (the two directions of a BiLSTM step, sequence
 * form: lengths / reverse / dgates_pos as in mstts_lstm_point_bwd; no d_out slabs, no d_out2; dhs2 slabs 1, 2, 4 or 8).  Returns
 * MSTTS_ERR_SHAPE when the pair form is not available for the geometry - the caller then issues two single calls. */
extern "C" int mstts_lstm_point_bwd_pair(const mstts_lstm_point_bwd_desc* a, const mstts_lstm_point_bwd_desc* b, mstts_stream_t s) {
    MSTTS_REQUIRE(a && b && a->B == b->B && a->H == b->H, MSTTS_ERR_SHAPE, "lstm_point_bwd_pair: the two cells must have the same shape");
    const int ph = a->dhs2_parts > 1 ? a->dhs2_parts : 1, phb = b->dhs2_parts > 1 ? b->dhs2_parts : 1;
    const bool ok = ph == phb && (ph == 1 || ph == 2 || ph == 4 || ph == 8) && a->dout_parts <= 1 && b->dout_parts <= 1 && !a->d_out2 && !b->d_out2 &&
                    !a->dq && !b->dq && a->B * a->H * 4 < (1LL << 30) && (a->B + 1) * a->dout_sb < (1LL << 30) && (a->B + 1) * a->dgp_sb < (1LL << 30) &&
                    (!a->d_h_state2) == (!b->d_h_state2);
    MSTTS_REQUIRE(ok, MSTTS_ERR_SHAPE, "lstm_point_bwd_pair: geometry not covered by the pair kernel");
    PointBwdFastPair p;
    point_bwd_fill(a, p.d[0]);
    point_bwd_fill(b, p.d[1]);
    dim3 grid((unsigned)((a->B * a->H + 127) / 128), 2);
    if (ph == 8) hipLaunchKernelGGL((lstm_point_bwd_fast_pair_kernel<8>), grid, dim3(128), 0, ST(s), p);
    else if (ph == 4) hipLaunchKernelGGL((lstm_point_bwd_fast_pair_kernel<4>), grid, dim3(128), 0, ST(s), p);
    else if (ph == 2) hipLaunchKernelGGL((lstm_point_bwd_fast_pair_kernel<2>), grid, dim3(128), 0, ST(s), p);
    else hipLaunchKernelGGL((lstm_point_bwd_fast_pair_kernel<1>), grid, dim3(128), 0, ST(s), p);
    MSTTS_CHECK_LAUNCH("lstm_point_bwd_pair");
    return MSTTS_OK;
}

extern "C" int mstts_lstm_point_bwd(const mstts_lstm_point_bwd_desc* d, mstts_stream_t s) {
    MSTTS_REQUIRE(d && d->d_c_state && d->d_h_state && d->acts && d->c_raw && d->c_prev && d->dgates && d->d_c_prev && d->d_h_prev,
                  MSTTS_ERR_SHAPE, "lstm_point_bwd: null pointer");
    if (d->B * d->H == 0) return MSTTS_OK;
    {
        const int po = d->dout_parts > 1 ? d->dout_parts : 1, po2 = d->dout2_parts > 1 ? d->dout2_parts : 1;
        const int ph = d->dhs2_parts > 1 ? d->dhs2_parts : 1;
        const bool seq = d->lengths || d->reverse || d->dgates_pos || d->dout_st != 0;
        const bool small = d->B * d->H * 4 < (1LL << 30) && (long)po * d->dout_pstride < (1LL << 30) && (d->B + 1) * d->dout_sb < (1LL << 30) &&
                           (d->B + 1) * d->dgp_sb < (1LL << 30);
        const bool fuse_q = d->dq && d->wq_t;
        MSTTS_REQUIRE(!fuse_q || (aligned16(d->dq) && aligned16(d->wq_t)), MSTTS_ERR_ALIGN, "lstm_point_bwd: dq / wq_t must be 16-byte aligned");
        MSTTS_REQUIRE(!fuse_q || d->H % 128 == 0, MSTTS_ERR_SHAPE, "lstm_point_bwd: the fused query-layer gradient needs H %% 128 == 0");
        MSTTS_REQUIRE(!fuse_q || (d->A == QA && !d->d_out2 && !d->lengths && !d->reverse && !d->dgates_pos && d->dout_st == 0), MSTTS_ERR_SHAPE,
                      "lstm_point_bwd: the fused query-layer gradient needs A == %d, no d_out2 and the plain (non-sequence) form", QA);
        int shape = -1;      // (P_OUT, P_OUT2, P_DHS)
        if (fuse_q) shape = (po == 1 && ph == 8) ? 10 : (po == 1 && ph == 1) ? 11 : (po == 1 && ph == 4) ? 12 : (po == 1 && ph == 2) ? 13 : -2;
        else if (po == 1 && po2 == 1 && ph == 1) shape = 0;
        else if (po == 1 && po2 == 1 && ph == 4) shape = 1;
        else if (po == 4 && po2 == 1 && ph == 4) shape = 2;
        else if (po == 1 && po2 == 1 && ph == 8) shape = 3;
        else if (po == 8 && po2 == 1 && ph == 8) shape = 4;
        else if (po == 8 && po2 == 1 && ph == 1) shape = 5;
        else if (po == 1 && po2 == 1 && ph == 2) shape = 6;
        MSTTS_REQUIRE(!fuse_q || (small && shape >= 10), MSTTS_ERR_SHAPE, "lstm_point_bwd: fused query-layer gradient not available for this slab geometry");
        if (small && shape >= 0) {
            PointBwdFast f;
            f.dq = d->dq; f.wq_t = d->wq_t;
            f.d_out = d->d_out; f.dout_ld = (int)d->dout_sb; f.dout_st = (int)d->dout_st; f.dout_parts = po; f.dout_pstride = (int)d->dout_pstride;
            f.d_out2 = d->d_out2; f.dout2_parts = po2; f.dout2_pstride = (int)d->dout2_pstride;
            f.d_c_state = d->d_c_state; f.d_h_state = d->d_h_state;
            f.dhs2 = d->d_h_state2; f.dhs2_ld = (int)d->dhs2_ld; f.dhs2_parts = ph; f.dhs2_pstride = (long)d->dhs2_pstride;
            f.acts = d->acts; f.c_raw = d->c_raw; f.c_prev = d->c_prev; f.zc = d->zc; f.zh = d->zh; f.keep = 1.f - d->zoneout;
            f.dgates = d->dgates; f.d_c_prev = d->d_c_prev; f.d_h_prev = d->d_h_prev;
            f.lengths = d->lengths; f.step = d->step; f.reverse = d->reverse;
            f.dgates_pos = d->dgates_pos; f.dgp_ld = (int)d->dgp_sb; f.dgp_st = (int)d->dgp_st;
            f.B = (int)d->B; f.H = (int)d->H;
            f.row2d = (d->H % 128 == 0 && d->B <= 65535) ? 1 : 0;
            dim3 grid((unsigned)((d->B * d->H + 127) / 128));
            if (f.row2d) grid = dim3((unsigned)(d->H / 128), (unsigned)d->B);
#define MSTTS_PB(A1, A2, A3)                                                                                          \
            if (seq) hipLaunchKernelGGL((lstm_point_bwd_fast_kernel<A1, A2, A3, true>), grid, dim3(128), 0, ST(s), f);   \
            else hipLaunchKernelGGL((lstm_point_bwd_fast_kernel<A1, A2, A3, false>), grid, dim3(128), 0, ST(s), f)
            switch (shape) {
                case 0: MSTTS_PB(1, 1, 1); break;
                case 1: MSTTS_PB(1, 1, 4); break;
                case 2: MSTTS_PB(4, 1, 4); break;
                case 3: MSTTS_PB(1, 1, 8); break;
                case 4: MSTTS_PB(8, 1, 8); break;
                case 5: MSTTS_PB(8, 1, 1); break;
#define MSTTS_PBQ(PH) do { if (d->dq_bf16) hipLaunchKernelGGL((lstm_point_bwd_fast_kernel<1, 1, PH, false, 2>), grid, dim3(128), 0, ST(s), f);     \
                           else hipLaunchKernelGGL((lstm_point_bwd_fast_kernel<1, 1, PH, false, 1>), grid, dim3(128), 0, ST(s), f); } while (0)
                case 10: MSTTS_PBQ(8); break;
                case 11: MSTTS_PBQ(1); break;
                case 12: MSTTS_PBQ(4); break;
                case 13: MSTTS_PBQ(2); break;
#undef MSTTS_PBQ
                default: MSTTS_PB(1, 1, 2); break;
            }
#undef MSTTS_PB
            MSTTS_CHECK_LAUNCH("lstm_point_bwd_fast");
            return MSTTS_OK;
        }
    }
    hipLaunchKernelGGL(lstm_point_bwd_kernel, dim3(grid_for(d->B * d->H, 256)), dim3(256), 0, ST(s), *d);
    MSTTS_CHECK_LAUNCH("lstm_point_bwd");
    return MSTTS_OK;
}

extern "C" int mstts_tts_loss_fwd_bwd(const float* linear, const float* post, const float* mel, const float* stop_logit,
                                      const int32_t* mel_length, int64_t B, int64_t S, int64_t n_mel, int32_t use_l1,
                                      float grad_scale, float* scalars, float* d_linear, float* d_post, float* d_stop, mstts_stream_t s) {
    MSTTS_REQUIRE(S >= 2 && B > 0 && n_mel > 0, MSTTS_ERR_SHAPE, "tts_loss: need S >= 2");
    hipLaunchKernelGGL(tts_loss_kernel, dim3(grid_for(B * S * n_mel, 256)), dim3(256), 0, ST(s), linear, post, mel, stop_logit, mel_length,
                       (long)B, (long)S, (long)n_mel, (int)use_l1, grad_scale, scalars, d_linear, d_post, d_stop);
    MSTTS_CHECK_LAUNCH("tts_loss_fwd_bwd");
    return MSTTS_OK;
}
extern "C" int mstts_shift_frames(const float* mel, float* frames, int64_t B, int64_t L, int64_t C, mstts_stream_t s) {
    hipLaunchKernelGGL(shift_frames_kernel, dim3(grid_for((L + 1) * B * C, 256)), dim3(256), 0, ST(s), mel, frames, (long)B, (long)L, (long)C);
    MSTTS_CHECK_LAUNCH("shift_frames");
    return MSTTS_OK;
}
extern "C" int mstts_unpack_proj(const float* proj, int64_t ldp, float* linear, float* stop, int64_t B, int64_t S, int64_t C, mstts_stream_t s) {
    hipLaunchKernelGGL(unpack_proj_kernel, dim3(grid_for(B * S * (C + 1), 256)), dim3(256), 0, ST(s), proj, (long)ldp, linear, stop, (long)B, (long)S, (long)C);
    MSTTS_CHECK_LAUNCH("unpack_proj");
    return MSTTS_OK;
}
extern "C" int mstts_pack_dproj(const float* d_linear, const float* d_stop, float* d_proj, int64_t ldp, int64_t B, int64_t S, int64_t C, mstts_stream_t s) {
    hipLaunchKernelGGL(pack_dproj_kernel, dim3(grid_for(B * S * ldp, 256)), dim3(256), 0, ST(s), d_linear, d_stop, d_proj, (long)ldp, (long)B, (long)S, (long)C);
    MSTTS_CHECK_LAUNCH("pack_dproj");
    return MSTTS_OK;
}
extern "C" int mstts_speaker_tile(const float* spk, const int32_t* lengths, float* values, int64_t B, int64_t T, int64_t M, int64_t off, int64_t width, mstts_stream_t s) {
    hipLaunchKernelGGL(speaker_tile_kernel, dim3(grid_for(B * T * width, 256)), dim3(256), 0, ST(s), spk, lengths, values, (long)B, (long)T, (long)M, (long)off, (long)width);
    MSTTS_CHECK_LAUNCH("speaker_tile");
    return MSTTS_OK;
}
extern "C" int mstts_conv_kernel_flip(const float* w, float* wt, int64_t K, int64_t Cin, int64_t Cout, mstts_stream_t s) {
    hipLaunchKernelGGL(conv_kernel_flip_kernel, dim3(grid_for(K * Cin * Cout, 256)), dim3(256), 0, ST(s), w, wt, (long)K, (long)Cin, (long)Cout);
    MSTTS_CHECK_LAUNCH("conv_kernel_flip");
    return MSTTS_OK;
}
extern "C" int mstts_transpose01(const float* src, float* dst, int64_t D0, int64_t D1, int64_t C, mstts_stream_t s) {
    if (D0 * D1 * C == 0) return MSTTS_OK;
    hipLaunchKernelGGL(transpose01_kernel, dim3(grid_for(D0 * D1 * C, 256)), dim3(256), 0, ST(s), src, dst, (long)D0, (long)D1, (long)C);
    MSTTS_CHECK_LAUNCH("transpose01");
    return MSTTS_OK;
}
extern "C" int mstts_speaker_finalize(const float* x, float* out, int64_t B, int64_t samples, int64_t T, int64_t E, mstts_stream_t s) {
    MSTTS_REQUIRE(B * E <= 65536 && B >= 1 && samples >= 1 && T >= 1, MSTTS_ERR_SHAPE, "speaker_finalize: bad shape");
    hipLaunchKernelGGL(speaker_finalize_kernel, dim3(1), dim3(1024), 0, ST(s), x, out, (int)B, (int)samples, (long)T, (int)E);
    MSTTS_CHECK_LAUNCH("speaker_finalize");
    return MSTTS_OK;
}
extern "C" int mstts_l2_loss_acc(const float* x, const uint8_t* mask, int64_t n, float* out, mstts_stream_t s) {
    if (n == 0) return MSTTS_OK;
    hipLaunchKernelGGL(l2_loss_kernel, dim3(grid_for(n, 256 * 8)), dim3(256), 0, ST(s), x, mask, (long)n, out);
    MSTTS_CHECK_LAUNCH("l2_loss_acc");
    return MSTTS_OK;
}
extern "C" int mstts_adam_tf(float* p, const float* grad, float* m, float* v, const uint8_t* wd_mask, float wd, float grad_scale,
                             float lr_t, float beta1, float beta2, float eps, int64_t n, mstts_stream_t s) {
    if (n == 0) return MSTTS_OK;
    hipLaunchKernelGGL(adam_tf_kernel, dim3(grid_for(n, 256)), dim3(256), 0, ST(s), p, grad, m, v, wd_mask, wd, grad_scale, lr_t, beta1, beta2, eps, (long)n);
    MSTTS_CHECK_LAUNCH("adam_tf");
    return MSTTS_OK;
}
