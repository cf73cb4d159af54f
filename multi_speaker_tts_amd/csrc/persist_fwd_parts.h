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
