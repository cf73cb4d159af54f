// Pieces shared by the two persistent forward loops of the decoder (persist.hip: teacher-forced train loop, persist_infer.hip: free-running
// inference loop): the matrix-core product over a staged activation slice, the slice hand-off, the zoneout cell update.
#pragma once
#include "persist_common.h"

namespace mstts {

constexpr int LC = 28, LA = 36;                  // padded row strides of the staged activation slices (conflict-free b128 reads)

// acc[t] += W[k-steps 4 K4A .. 4 K4B) . X[t], X read from the staged LDS slice (row stride LD); the wave's registers w[WOFF + ks]
// NT = 1: row tile 0 only (a batch of at most 16 rows: the second tile's rows do not exist)
template <int K4A, int K4B, int LD, int WOFF, int NW, int NT = 2>
__device__ __forceinline__ void mfma_part(const float (&w)[NW], const float* sx, int lane, pf32x4 (&acc)[2]) {
    const int row0 = ((lane >> 4) * 2) * 16 + (lane & 15);          // rho of row tile 0; tile 1 is 16 rows further
#pragma unroll
    for (int k4 = K4A; k4 < K4B; ++k4) {
        const pf32x4 x0 = *reinterpret_cast<const pf32x4*>(sx + row0 * LD + 4 * k4);
        if (NT == 2) {
            const pf32x4 x1 = *reinterpret_cast<const pf32x4*>(sx + (row0 + 16) * LD + 4 * k4);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                acc[0] = PMFMA(w[WOFF + 4 * k4 + e], x0[e], acc[0]);
                acc[1] = PMFMA(w[WOFF + 4 * k4 + e], x1[e], acc[1]);
            }
        } else {
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[0] = PMFMA(w[WOFF + 4 * k4 + e], x0[e], acc[0]);
        }
    }
}

// one slice of a ring slot (128 rows x 4 K4 floats, contiguous) -> the staging buffer (row stride LD), in two parts: slice_issue()
// requests the thread's two 16-byte pieces, slice_complete() polls them in and writes them to LDS; false on time-out
template <int K4>
__device__ __forceinline__ void slice_issue(__amdgpu_buffer_rsrc_t xr, long slice_float_off, int tid, unsigned (&off)[2], pf32x4 (&v)[2]) {
    constexpr int NPC = 128 * K4;                                    // 16-byte pieces of the slice: 768 or 1024 for 512 threads
    off[0] = (unsigned)(slice_float_off * 4 + 16 * tid);
    off[1] = (tid + PTH < NPC) ? off[0] + 16 * PTH : off[0];
    issue<2>(xr, off, v);
}
template <int K4, int LD>
__device__ __forceinline__ bool slice_complete(__amdgpu_buffer_rsrc_t xr, float* stg, int tid, const unsigned (&off)[2], pf32x4 (&v)[2], const unsigned* ctrl, unsigned gen) {
    constexpr int NPC = 128 * K4;
    const unsigned gens[2] = {gen, gen};
    const bool ok = complete<2>(xr, off, v, ctrl, gens);
    {
        const int rho = tid / K4, k4 = tid - rho * K4;
        *reinterpret_cast<pf32x4*>(stg + rho * LD + 4 * k4) = v[0];
    }
    if (tid + PTH < NPC) {
        const int p = tid + PTH, rho = p / K4, k4 = p - rho * K4;
        *reinterpret_cast<pf32x4*>(stg + rho * LD + 4 * k4) = v[1];
    }
    return ok;
}

// ---- BASELINE config 3 ("bf16 with fp32 master"): the same products with both operands rounded to bf16 (round to nearest even) and fp32
// accumulation on v_mfma_f32_16x16x32_bf16 - 16 times the f32-input MFMA's rate.  One instruction covers EIGHT k-steps of the fp32 form: lane
// (q = lane >> 4, m) holds, as A operand, the 8 kernel values of k-steps 8 j .. 8 j + 7 it would have used one at a time (same registers,
// packed in pairs), and as B operand 8 consecutive activations of ITS staging row (the staging rows are per q already) - the sum over the
// instruction's inner index 8 q + e is the sum over (q, k-step), i.e. the same contraction.  The staged slices are kept as bf16 (converted
// once, by the thread that fetched the piece, instead of once per consuming wave): row stride 80 B, conflict-free 16-byte reads.
typedef __bf16 pbf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 pbf16x4 __attribute__((ext_vector_type(4)));
constexpr int LB16 = 40;
#define PMFMA_BF16(a, b, c) __builtin_amdgcn_mfma_f32_16x16x32_bf16((a), (b), (c), 0, 0, 0)
__device__ __forceinline__ pbf16x4 to_bf16x4(const pf32x4& v) {
    pbf16x4 r;
#pragma unroll
    for (int e = 0; e < 4; ++e) r[e] = (__bf16)v[e];
    return r;
}
__device__ __forceinline__ float bf16_round(float x) { return (float)(__bf16)x; }
// acc[t] += W[k-steps 8 K8A .. 8 K8B) . X[t]; kernel registers wb[WOFF8 + j] = k-steps 8 j .. 8 j + 7
template <int K8A, int K8B, int WOFF8, int NW, int NT = 2>
__device__ __forceinline__ void mfma_part_bf16(const pbf16x8 (&wb)[NW], const __bf16* sx, int lane, pf32x4 (&acc)[2]) {
    const int row0 = ((lane >> 4) * 2) * 16 + (lane & 15);
#pragma unroll
    for (int j = K8A; j < K8B; ++j) {
        const pbf16x8 x0 = *reinterpret_cast<const pbf16x8*>(sx + row0 * LB16 + 8 * j);
        acc[0] = PMFMA_BF16(wb[WOFF8 + j], x0, acc[0]);
        if (NT == 2) {
            const pbf16x8 x1 = *reinterpret_cast<const pbf16x8*>(sx + (row0 + 16) * LB16 + 8 * j);
            acc[1] = PMFMA_BF16(wb[WOFF8 + j], x1, acc[1]);
        }
    }
}
// slice_complete with the pieces rounded to bf16 on their way into the staging buffer
template <int K4>
__device__ __forceinline__ bool slice_complete_bf16(__amdgpu_buffer_rsrc_t xr, __bf16* stg, int tid, const unsigned (&off)[2], pf32x4 (&v)[2], const unsigned* ctrl, unsigned gen) {
    constexpr int NPC = 128 * K4;
    const unsigned gens[2] = {gen, gen};
    const bool ok = complete<2>(xr, off, v, ctrl, gens);
    {
        const int rho = tid / K4, k4 = tid - rho * K4;
        *reinterpret_cast<pbf16x4*>(stg + rho * LB16 + 4 * k4) = to_bf16x4(v[0]);
    }
    if (tid + PTH < NPC) {
        const int p = tid + PTH, rho = p / K4, k4 = p - rho * K4;
        *reinterpret_cast<pbf16x4*>(stg + rho * LB16 + 4 * k4) = to_bf16x4(v[1]);
    }
    return ok;
}

// ---- fp32 at 6/16 of the f32-input MFMA's time, for ONE product of the fp32 loop (the on-chain m0 . W1 of cell 1): both operands as exact
// three-way bf16 splits (x = hi + mid + lo, differences exact in fp32), six bf16 products per fp32 product, fp32 accumulate - the arithmetic of
// gemm_split.inc.  The staged slice is three planes [128 rows][32 k] of bf16 without padding (the LDS has no room for it), the four 16-byte
// octets of a row XOR-swizzled by (row >> 2) & 3 so that the 16 lanes a ds_read_b128 serves together hit 16 different bank quads.
constexpr int SP3_PLANE = 128 * 32;
__device__ __forceinline__ int sp3_off(int rho, int j) { return rho * 32 + ((j ^ ((rho >> 2) & 3)) << 3); }
__device__ __forceinline__ void sp3_put_rk(__bf16* stg, int rho, int k4, const pf32x4& v);
__device__ __forceinline__ void sp3_put(__bf16* stg, int p, const pf32x4& v) { sp3_put_rk(stg, p >> 3, p & 7, v); }       // piece p = (row rho, float4 k4) of a 128 x 32 slice
__device__ __forceinline__ void sp3_put_rk(__bf16* stg, int rho, int k4, const pf32x4& v) {
    pbf16x4 hi, mid, lo;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const float x = v[e];
        hi[e] = (__bf16)x; const float r1 = x - (float)hi[e];
        mid[e] = (__bf16)r1; lo[e] = (__bf16)(r1 - (float)mid[e]);
    }
    __bf16* q = stg + sp3_off(rho, k4 >> 1) + 4 * (k4 & 1);
    *reinterpret_cast<pbf16x4*>(q) = hi;
    *reinterpret_cast<pbf16x4*>(q + SP3_PLANE) = mid;
    *reinterpret_cast<pbf16x4*>(q + 2 * SP3_PLANE) = lo;
}
__device__ __forceinline__ bool slice_complete_split3(__amdgpu_buffer_rsrc_t xr, __bf16* stg, int tid, const unsigned (&off)[2], pf32x4 (&v)[2], const unsigned* ctrl, unsigned gen) {
    const unsigned gens[2] = {gen, gen};
    const bool ok = complete<2>(xr, off, v, ctrl, gens);
    sp3_put(stg, tid, v[0]);
    sp3_put(stg, tid + PTH, v[1]);
    return ok;
}
// a 128 x 24 slice (K4 = 6 pieces per row: the context rows) into octets 0 .. 2 of the planes
__device__ __forceinline__ bool slice_complete_split3_k6(__amdgpu_buffer_rsrc_t xr, __bf16* stg, int tid, const unsigned (&off)[2], pf32x4 (&v)[2], const unsigned* ctrl, unsigned gen) {
    const unsigned gens[2] = {gen, gen};
    const bool ok = complete<2>(xr, off, v, ctrl, gens);
    { const int rho = tid / 6; sp3_put_rk(stg, rho, tid - rho * 6, v[0]); }
    if (tid + PTH < 128 * 6) { const int p = tid + PTH, rho = p / 6; sp3_put_rk(stg, rho, p - rho * 6, v[1]); }
    return ok;
}
// acc[t] += W[k-steps 8 JA .. 8 JB) . X[t] with wsp[plane][octet] and the staged planes: hi.hi + mid.hi + hi.mid + mid.mid + lo.hi + hi.lo
template <int JA = 0, int JB = 4>
__device__ __forceinline__ void mfma_part_split3(const pbf16x8 (&wsp)[3][4], const __bf16* sx, int lane, pf32x4 (&acc)[2]) {
    const int row0 = ((lane >> 4) * 2) * 16 + (lane & 15);
#pragma unroll
    for (int j = JA; j < JB; ++j) {
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const __bf16* p = sx + sp3_off(row0 + 16 * t, j);
            const pbf16x8 xh = *reinterpret_cast<const pbf16x8*>(p), xm = *reinterpret_cast<const pbf16x8*>(p + SP3_PLANE),
                          xl = *reinterpret_cast<const pbf16x8*>(p + 2 * SP3_PLANE);
            acc[t] = PMFMA_BF16(wsp[0][j], xh, acc[t]);
            acc[t] = PMFMA_BF16(wsp[1][j], xh, acc[t]);
            acc[t] = PMFMA_BF16(wsp[0][j], xm, acc[t]);
            acc[t] = PMFMA_BF16(wsp[1][j], xm, acc[t]);
            acc[t] = PMFMA_BF16(wsp[2][j], xh, acc[t]);
            acc[t] = PMFMA_BF16(wsp[0][j], xl, acc[t]);
        }
    }
}

struct CellOut { float si, tj, sf, so, c, m; };
// ZoneoutLSTMCell.py:228-271 for one (row, unit): gates i, j, f, o (forget bias 1.0 added here); zoneout as state' = k (new - old) + old with
// k = (1 - z) * keep-mask in training (:266-271) and k = 1 - z at inference (:259-264)
__device__ __forceinline__ CellOut cell_update(const pf32x4& gs, const float (&add)[4], float& cs, float& hs, float kc, float kh) {
    CellOut o;
    o.si = sigmoidf_(gs[0] + add[0]); o.tj = tanhf_(gs[1] + add[1]); o.sf = sigmoidf_(gs[2] + add[2] + 1.0f); o.so = sigmoidf_(gs[3] + add[3]);
    o.c = o.sf * cs + o.si * o.tj;
    o.m = o.so * tanhf_(o.c);
    hs = kh * (o.m - hs) + hs;
    cs = kc * (o.c - cs) + cs;
    return o;
}

// k-step ks of a 128-unit recurrent slice: q = lane >> 4 -> producer column slice j' = 4 (ks / 4) + q, unit 32 j' + 4 i + ks % 4
__device__ __forceinline__ int unit_of_kstep(int gi, int ks, int q) { return 32 * (4 * (ks >> 2) + q) + 4 * gi + (ks & 3); }

}  // namespace mstts
