// Shared helpers for the gfx950 kernels of libmstts_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include "../../include/mstts.h"

namespace mstts {

// thread-local last-error text, returned by mstts_last_error()
char* err_buf();
int set_err(int code, const char* fmt, ...);

#define MSTTS_CHECK_LAUNCH(name)                                                      \
    do {                                                                              \
        hipError_t e__ = hipGetLastError();                                           \
        if (e__ != hipSuccess)                                                        \
            return mstts::set_err(MSTTS_ERR_LAUNCH, "%s: %s", name, hipGetErrorString(e__)); \
    } while (0)

#define MSTTS_REQUIRE(cond, code, ...)                       \
    do {                                                     \
        if (!(cond)) return mstts::set_err(code, __VA_ARGS__); \
    } while (0)

static inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }
static inline int cdiv(long a, long b) { return (int)((a + b - 1) / b); }

// (v_rcp_f32, 1 ulp, instead of the IEEE division's ten-instruction sequence: these sit on the serial tails of the recurrent kernels)
__device__ __forceinline__ float sigmoidf_(float x) { return __builtin_amdgcn_rcpf(1.0f + __expf(-x)); }
__device__ __forceinline__ float sigmoid_acc(float x) { return 1.0f / (1.0f + expf(-x)); }
// tanh through exp: |err| ~ 1e-7 rel, saturates correctly for large |x|
__device__ __forceinline__ float tanhf_(float x) {
    float ax = fabsf(x);
    float e = __expf(-2.0f * ax);
    float t = (1.0f - e) * __builtin_amdgcn_rcpf(1.0f + e);
    return copysignf(t, x);
}

// Packed activation block of a fused cell with reduction width K = 64 nit (what cell_fwd_kernel reads, csrc/cell.hip): element
// (row r, column k) lives where lane (k % 16 / 4) * 16 + r % 16 of wave k / (16 nit) finds it in its float4 number
// (k % (16 nit)) / 16 for row tile (r % 32) / 16 - every wave load of the consumer is one contiguous 1 KB.  Blocks of 32 rows
// follow each other.
__device__ __forceinline__ int cell_act_offset(int r, int k, int nit) {
    const int q = 16 * nit, blk = r >> 5, rr = r & 31;
    const int w = (k >= q) + (k >= 2 * q) + (k >= 3 * q), kk = k - w * q;    // k / q without a division (k < 4 q; this sits on producers' tails)
    return blk * (2048 * nit) + (((w * nit + (kk >> 4)) * 2 + (rr >> 4)) << 8) + ((((kk & 15) >> 2) * 16 + (rr & 15)) << 2) + (kk & 3);
}
// bf16 form (config 3): 16-byte lane fragments of v_mfma_f32_16x16x32_bf16 - wave k / (K/4), 32-k step (k % (K/4)) / 32, lane
// ((k % 32) / 8) * 16 + r % 16 holds 8 consecutive k; nit = K / 128.  Offsets in bf16 elements.
__device__ __forceinline__ int cell_act_offset_bf16(int r, int k, int nit) {
    const int q = 32 * nit, blk = r >> 5, rr = r & 31;
    const int w = (k >= q) + (k >= 2 * q) + (k >= 3 * q), kk = k - w * q;
    return blk * (4096 * nit) + (((w * nit + (kk >> 5)) * 2 + (rr >> 4)) << 9) + ((((kk & 31) >> 3) * 16 + (rr & 15)) << 3) + (kk & 7);
}
struct PackedDst { float* base; int nit, col0, bf; };  // packed block of a consumer cell (bf: bf16 form); the producer owns columns col0 ...
__device__ __forceinline__ void packed_store(const PackedDst& p, int r, int c, float v) {
    if (p.bf) reinterpret_cast<__bf16*>(p.base)[cell_act_offset_bf16(r, p.col0 + c, p.nit)] = (__bf16)v;
    else p.base[cell_act_offset(r, p.col0 + c, p.nit)] = v;
}
// validates a mstts_cell_packed_dst whose producer writes `width` columns
int packed_dst_from(const mstts_cell_packed_dst* p, int64_t width, PackedDst* o, const char* what);

// sum of up to MAXP partial slabs p[pp * stride + idx]; all loads are issued before the first add
// (a runtime-length `for` would serialize one memory round trip per slab)
template <int MAXP>
__device__ __forceinline__ float sum_parts(const float* __restrict__ p, int parts, long stride, long idx) {
    float t[MAXP];
#pragma unroll
    for (int pp = 0; pp < MAXP; ++pp) t[pp] = (pp == 0 || pp < parts) ? p[pp * stride + idx] : 0.f;
    float v = t[0];
#pragma unroll
    for (int pp = 1; pp < MAXP; ++pp) v += t[pp];
    return v;
}
constexpr int MSTTS_MAX_PARTS = 16;

// Wave-wide reductions on the DPP path (row operations inside the VALU, a few cycles each) instead of ds_bpermute shuffles (an
// LDS round trip per step: six dependent steps were ~0.25 us on the critical path of the latency-bound per-step kernels).
//   quad_perm 1,0,3,2 / 2,3,0,1 -> quads ; row_half_mirror -> 8 ; row_mirror -> rows of 16 ; row_bcast15 (rows 1, 3) -> halves of 32
//   in lanes 16-31 / 48-63 ; row_bcast31 (rows 2, 3) -> the wave's total in row 3 ; readlane 63 broadcasts it.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_mov(float old, float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(old), __float_as_int(v), CTRL, ROW_MASK, 0xf, false));
}
// sums of the two 32-lane halves: valid in lanes 31 (lanes 0-31) and 63 (lanes 32-63)
__device__ __forceinline__ float half_sum_in_last_lane(float v) {
    v += dpp_mov<0xB1, 0xf>(0.f, v);
    v += dpp_mov<0x4E, 0xf>(0.f, v);
    v += dpp_mov<0x141, 0xf>(0.f, v);
    v += dpp_mov<0x140, 0xf>(0.f, v);
    v += dpp_mov<0x142, 0xa>(0.f, v);
    return v;
}
__device__ __forceinline__ float wave_sum(float v) {
    v = half_sum_in_last_lane(v);
    v += dpp_mov<0x143, 0xc>(0.f, v);
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}
__device__ __forceinline__ float wave_max(float v) {
    v = fmaxf(v, dpp_mov<0xB1, 0xf>(v, v));
    v = fmaxf(v, dpp_mov<0x4E, 0xf>(v, v));
    v = fmaxf(v, dpp_mov<0x141, 0xf>(v, v));
    v = fmaxf(v, dpp_mov<0x140, 0xf>(v, v));
    v = fmaxf(v, dpp_mov<0x142, 0xa>(v, v));
    v = fmaxf(v, dpp_mov<0x143, 0xc>(v, v));
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}

// block-wide sum for blockDim.x <= 1024 (multiple of 64); scratch needs 16 floats of LDS
__device__ __forceinline__ float block_sum(float v, float* scratch) {
    v = wave_sum(v);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
    __syncthreads();
    if (lane == 0) scratch[w] = v;
    __syncthreads();
    float r = 0.f;
    for (int i = 0; i < nw; ++i) r += scratch[i];
    return r;
}
__device__ __forceinline__ float block_max(float v, float* scratch) {
    v = wave_max(v);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
    __syncthreads();
    if (lane == 0) scratch[w] = v;
    __syncthreads();
    float r = scratch[0];
    for (int i = 1; i < nw; ++i) r = fmaxf(r, scratch[i]);
    return r;
}

}  // namespace mstts
