// Round 6 probe for DESIGN 6b: what would it cost the persistent BPTT to move the query-layer data-gradient product from the attention workgroups
// (each publishes a PERSONAL 16-byte piece to each of the 256 cell owners: 4 KB out, 4 KB in per workgroup and step - today's form, which needs
// the 64 KB query-kernel slice in LDS) to the owners (each attention workgroup publishes its 64-byte dq slice ONCE and all 256 owners read all
// 256 slices: a 256-way broadcast of 16 KB)?  256 co-resident workgroups, one per CU, step after step: publish, gather everything, barrier.
//   mode 0: personal pieces (writer g -> reader r at [r][g], 16 B)               - the shipped pattern
//   mode 1: broadcast, the 256 slices contiguous (16 KB)
//   mode 2: broadcast, slice g at a stride of 4 160 bytes (other channel / bank every slice)
//   hipcc --offload-arch=gfx950 -O3 tools/broadcast_probe.hip -o /tmp/broadcast_probe && /tmp/broadcast_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f4 __attribute__((ext_vector_type(4)));
typedef int i4 __attribute__((ext_vector_type(4)));
constexpr int WG = 256, RING = 4, TH = 256;
__device__ __forceinline__ bool stale(const f4& v, unsigned gen) {
    return (((__float_as_uint(v[0]) ^ gen) | (__float_as_uint(v[1]) ^ gen) | (__float_as_uint(v[2]) ^ gen) | (__float_as_uint(v[3]) ^ gen)) & 1u) != 0u;
}
__device__ __forceinline__ f4 tagv(f4 v, unsigned gen) {
    for (int e = 0; e < 4; ++e) v[e] = __uint_as_float((__float_as_uint(v[e]) & ~1u) | gen);
    return v;
}
__device__ __forceinline__ f4 xload(__amdgpu_buffer_rsrc_t r, unsigned off) { return __builtin_bit_cast(f4, __builtin_amdgcn_raw_buffer_load_b128(r, (int)off, 0, 16)); }
__device__ __forceinline__ void xstore(__amdgpu_buffer_rsrc_t r, unsigned off, f4 v) { __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(i4, v), r, (int)off, 0, 16); }
constexpr long SLOT_BYTES = 256L * 4160;     // covers every mode (mode 0 needs 256 * 256 * 16 = 1 MB)
__global__ __launch_bounds__(TH) void probe(float* ring, unsigned long long* ticks, int rounds, int mode, unsigned* arrive) {
    extern __shared__ float pad[];
    const int g = blockIdx.x, tid = threadIdx.x;
    const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(ring, 0, (int)(RING * (mode == 0 ? 1048576L : SLOT_BYTES)), 0x00020000);
    const long slot_bytes = mode == 0 ? 1048576L : SLOT_BYTES;
    // start rendezvous
    if (tid == 0) { atomicAdd(arrive, 1u); while (atomicAdd(arrive, 0u) < (unsigned)WG) __builtin_amdgcn_s_sleep(4); }
    __syncthreads();
    float chk = 0.f;
    const unsigned long long t0 = wall_clock64();
    for (int k = 0; k < rounds; ++k) {
        const unsigned slot = k & 3, gen = (k >> 2) & 1;
        const unsigned base = (unsigned)(slot * slot_bytes);
        const f4 val = tagv((f4){(float)(g + 1), (float)k, 3.f, 4.f}, gen);
        if (mode == 0) {
            xstore(r, base + (unsigned)(((long)tid * WG + g) * 16), val);                       // my piece for reader `tid`
        } else if (tid < 4) {
            const unsigned off = mode == 1 ? (unsigned)(g * 64 + tid * 16) : (unsigned)(g * 4160 + tid * 16);
            xstore(r, base + off, val);
        }
        // gather: thread t takes writer t's data
        f4 v[4];
        unsigned off[4];
        const int np = mode == 0 ? 1 : 4;
        for (int p = 0; p < np; ++p)
            off[p] = base + (mode == 0 ? (unsigned)(((long)g * WG + tid) * 16) : mode == 1 ? (unsigned)(tid * 64 + p * 16) : (unsigned)(tid * 4160 + p * 16));
        asm volatile("" ::: "memory");
        for (int p = 0; p < np; ++p) v[p] = xload(r, off[p]);
        for (unsigned spins = 0; spins < 4000000u; ++spins) {
            asm volatile("" ::: "memory");
            bool miss = false;
            for (int p = 0; p < np; ++p) miss |= stale(v[p], gen);
            if (!__builtin_amdgcn_ballot_w64(miss)) break;
            for (int p = 0; p < np; ++p) if (stale(v[p], gen)) v[p] = xload(r, off[p]);
        }
        chk += v[0][0];
        __syncthreads();
    }
    if (tid == 0 && g == 0) { ticks[0] = wall_clock64() - t0; ticks[1] = (unsigned long long)chk; }
}
int main() {
    float* ring; unsigned long long* ticks; unsigned* arrive;
    hipMalloc(&ring, RING * SLOT_BYTES); hipMalloc(&ticks, 64); hipMalloc(&arrive, 4);
    const int rounds = 2000;
    const size_t lds = 150 * 1024;
    hipFuncSetAttribute((const void*)probe, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    const char* names[3] = {"personal 16-byte pieces (shipped pattern: 4 KB out + 4 KB in per workgroup)", "broadcast of 64-byte slices, contiguous 16 KB", "broadcast of 64-byte slices, 4 160-byte stride"};
    for (int rep = 0; rep < 2; ++rep)
        for (int mode = 0; mode < 3; ++mode) {
            hipMemset(ring, 0xFF, RING * SLOT_BYTES); hipMemset(ticks, 0, 64); hipMemset(arrive, 0, 4);
            hipLaunchKernelGGL(probe, dim3(WG), dim3(TH), lds, 0, ring, ticks, rounds, mode, arrive);
            hipDeviceSynchronize();
            unsigned long long h[2];
            hipMemcpy(h, ticks, 16, hipMemcpyDeviceToHost);
            printf("mode %d  %-82s : %.3f us per all-to-all step (check %llu)\n", mode, names[mode], h[0] * 0.01 / rounds, h[1]);
        }
    return 0;
}
