// Two-layer prenet of one decoder step for all rows of the batch on MFMA (the stand-alone launch of decoder.hip), written against a
// frame source so that the frame can come from memory or from an in-flight hand-off.
#pragma once
#include "common.h"
namespace mstts {
// Two-layer prenet of one decoder step for B <= 32 rows in ONE launch (Modules.py:239-255; dropout always on):
//   h1 = drop0(relu(frame . W0 + b0)),  out[:, c0:c0+16] = drop1(relu(h1 . W1[:, c0:c0+16] + b1)).
// P/16 workgroups of 4 waves; every workgroup recomputes the small first layer (wave w owns column tiles w, w+4, ..)
// and owns one 16-column tile of the second (its reduction split over the 4 waves).  fp32 MFMA 16x16x4, the kernel
// operands go global -> registers in one round trip issued before anything else.
constexpr int PN_MAXB = 32, PN_COLS = 16, PN_MAXNM4 = 20, PN_MAXTPW = 4, PN_MAXK1 = 16;
typedef float pn_f32x4 __attribute__((ext_vector_type(4)));
// frame source of the stand-alone launch: the previous step's linear output in memory
struct PnFramePlain {
    const float* frame;
    __device__ __forceinline__ unsigned long long issue(int e, int, int) const { return (unsigned long long)__float_as_uint(frame[e]); }
    __device__ __forceinline__ float value(unsigned long long t, int, int) const { return __uint_as_float((unsigned)t); }
    __device__ __forceinline__ void repair(float*, int) const {}
};
inline size_t prenet_lds_bytes(long NM, long P) { return sizeof(float) * (size_t)(PN_MAXB * (NM + 1) + PN_MAXB * (P + 1) + 4 * 32 * 17); }
// c0: first of the workgroup's 16 output columns; sm: PN_MAXB (NM + 1) + PN_MAXB (P + 1) + 4 * 32 * 17 floats of LDS; 256 threads
template <class Src>
__device__ __forceinline__ void prenet_body(Src& src, int c0, int NM, const float* __restrict__ w0, const float* __restrict__ b0,
                                            const float* __restrict__ w1, const float* __restrict__ b1, const uint8_t* __restrict__ m0,
                                            const uint8_t* __restrict__ m1, float inv_keep, int B, int P, float* __restrict__ out, long out_ld,
                                            PackedDst out_p, float* sm) {
    __shared__ __attribute__((aligned(16))) uint8_t s_m0[PN_MAXB * 64 * PN_MAXTPW];       // first dropout mask, rows >= B zero
    const int ldx = NM + 1, ldh = P + 1;
    float* s_x = sm;                               // [32][NM + 1]
    float* s_h = s_x + PN_MAXB * ldx;              // [32][P + 1]
    float* s_r = s_h + PN_MAXB * ldh;              // [4][32][17]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int j = lane & 15, kq = lane >> 4;
    const int nit0 = NM / 4, tpw = P / 64, nit1 = P / 16;          // k-steps of layer 0; column tiles per wave; k-steps per wave of layer 1
    const bool two = B > 16;
    // ---- loads first
    constexpr int NX = (PN_MAXB * 4 * PN_MAXNM4 + 255) / 256;
    const float inv_nm = 1.f / (float)NM;
    unsigned long long xt[NX];                    // frame elements as the source's tokens (plain: the value)
#pragma unroll
    for (int i = 0; i < NX; ++i) {
        const int e = tid + 256 * i;
        const int row = (int)(((float)e + 0.5f) * inv_nm);                   // e / NM without the integer division (e < 2560: exact)
        xt[i] = (e < B * NM) ? src.issue(e, row, e - row * NM) : 0ull;
    }
    // layer-0 kernel: lane j of wave w holds columns (16 w + j) tpw + t of its tpw column tiles t, so each k row is one 16-byte load
    // (tpw == 4) instead of four words 64 bytes apart
    float w0r[PN_MAXTPW][PN_MAXNM4];
    const int colb = (16 * wave + j) * tpw;
    if (tpw == 4) {
#pragma unroll
        for (int it = 0; it < PN_MAXNM4; ++it) {
            const float4 x = (it < nit0) ? *reinterpret_cast<const float4*>(w0 + (long)(4 * it + kq) * P + colb) : make_float4(0.f, 0.f, 0.f, 0.f);
            w0r[0][it] = x.x; w0r[1][it] = x.y; w0r[2][it] = x.z; w0r[3][it] = x.w;
        }
    } else {
#pragma unroll
        for (int t = 0; t < PN_MAXTPW; ++t)
#pragma unroll
            for (int it = 0; it < PN_MAXNM4; ++it)
                w0r[t][it] = (t < tpw && it < nit0) ? w0[(long)(4 * it + kq) * P + colb + t] : 0.f;
    }
    float w1r[PN_MAXK1];
#pragma unroll
    for (int it = 0; it < PN_MAXK1; ++it)
        w1r[it] = (it < nit1 && c0 + j < P) ? w1[(long)(wave * (P / 4) + 4 * it + kq) * P + c0 + j] : 0.f;
    // (the dropout masks and biases of both layers too: they do not depend on anything computed here, and fetched where they are
    // used each was a memory round trip of its own on the step's critical path)
    // the first mask goes to LDS through two 16-byte loads per thread (fetched byte by byte where it is used - 32 loads of one byte per
    // lane - it cost 4.6 us of the kernel's 13.8)
    float b0r[PN_MAXTPW];
#pragma unroll
    for (int t = 0; t < PN_MAXTPW; ++t) b0r[t] = (t < tpw) ? b0[colb + t] : 0.f;
    uint4 m0q[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int e = 16 * (tid + 256 * i);
        m0q[i] = (e < B * P) ? *reinterpret_cast<const uint4*>(m0 + e) : make_uint4(0u, 0u, 0u, 0u);
    }
    float b1r[2];
    float m1r[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int e = tid + 256 * i, b = e / PN_COLS, cc = e % PN_COLS;
        const bool live = b < B && c0 + cc < P;
        b1r[i] = live ? b1[c0 + cc] : 0.f;
        m1r[i] = live ? (float)m1[(long)b * P + c0 + cc] : 0.f;
    }
    {
#pragma unroll
        for (int i = 0; i < 2; ++i) reinterpret_cast<uint4*>(s_m0)[tid + 256 * i] = m0q[i];
#pragma unroll
        for (int i = 0; i < NX; ++i) {
            const int e = tid + 256 * i;
            const int row = (int)(((float)e + 0.5f) * inv_nm);
            if (e < PN_MAXB * NM) s_x[row * ldx + e - row * NM] = (e < B * NM) ? src.value(xt[i], row, e - row * NM) : 0.f;     // rows >= B are zero
        }
    }
    __syncthreads();
    src.repair(s_x, ldx);                         // (hook for a source whose values may fail to arrive; workgroup-uniform)
    // ---- layer 0
    float a0[PN_MAXNM4], a1[PN_MAXNM4];
#pragma unroll
    for (int it = 0; it < PN_MAXNM4; ++it) {
        a0[it] = (it < nit0) ? s_x[j * ldx + 4 * it + kq] : 0.f;
        a1[it] = (it < nit0) ? s_x[(16 + j) * ldx + 4 * it + kq] : 0.f;
    }
#pragma unroll
    for (int t = 0; t < PN_MAXTPW; ++t) {
        if (t < tpw) {
            pn_f32x4 acc0 = (pn_f32x4){0.f, 0.f, 0.f, 0.f}, acc1 = acc0;
#pragma unroll
            for (int it = 0; it < PN_MAXNM4; ++it) {
                if (it < nit0) {
                    acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[it], w0r[t][it], acc0, 0, 0, 0);
                    if (two) acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1[it], w0r[t][it], acc1, 0, 0, 0);
                }
            }
            const int col = colb + t;
            const float bj = b0r[t];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = kq * 4 + r;
                s_h[row * ldh + col] = s_m0[row * P + col] ? fmaxf(acc0[r] + bj, 0.f) * inv_keep : 0.f;
                s_h[(16 + row) * ldh + col] = s_m0[(16 + row) * P + col] ? fmaxf(acc1[r] + bj, 0.f) * inv_keep : 0.f;
            }
        }
    }
    __syncthreads();
    // ---- layer 1: this wave's quarter of the reduction
    {
        pn_f32x4 acc0 = (pn_f32x4){0.f, 0.f, 0.f, 0.f}, acc1 = acc0;
        const float* ha = s_h + j * ldh + wave * (P / 4) + kq;
#pragma unroll
        for (int it = 0; it < PN_MAXK1; ++it) {
            if (it < nit1) {
                acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(ha[4 * it], w1r[it], acc0, 0, 0, 0);
                if (two) acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(ha[16 * ldh + 4 * it], w1r[it], acc1, 0, 0, 0);
            }
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            s_r[(wave * 32 + kq * 4 + r) * 17 + j] = acc0[r];
            s_r[(wave * 32 + 16 + kq * 4 + r) * 17 + j] = acc1[r];
        }
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int e = tid + 256 * i, b = e / PN_COLS, cc = e % PN_COLS;
        if (b >= B || c0 + cc >= P) continue;
        const float v = s_r[b * 17 + cc] + s_r[(32 + b) * 17 + cc] + s_r[(64 + b) * 17 + cc] + s_r[(96 + b) * 17 + cc] + b1r[i];
        const float y = fmaxf(v, 0.f) * (fminf(m1r[i], 1.f) * inv_keep);
        out[(long)b * out_ld + c0 + cc] = y;
        if (out_p.base) packed_store(out_p, b, c0 + cc, y);          // the fused cell-0 step reads its input row from the packed block
    }
}
}  // namespace mstts
