// Cost of a grid-wide barrier carried by memory on gfx950 (256 workgroups, one per CU), with the data visibility a two-stage kernel
// needs: every workgroup writes a slab, release, arrives, waits for all, acquire, reads the slabs of the others and checks them.
//   hipcc --offload-arch=gfx950 -O3 tools/gridsync_probe.hip -o /tmp/gridsync_probe && /tmp/gridsync_probe
#include <hip/hip_runtime.h>
#include <cstdio>
constexpr int WG = 256, TH = 256, SLAB = 1024;          // floats per workgroup per round (4 KB)
__global__ __launch_bounds__(TH) void rounds_kernel(float* data, unsigned long long* counter, int rounds, int* errors, int mode) {
    const int tid = threadIdx.x, wg = blockIdx.x;
    int bad = 0;
    for (int r = 1; r <= rounds; ++r) {
        float* mine = data + (long)(r & 1) * WG * SLAB + (long)wg * SLAB;
        for (int i = tid; i < SLAB; i += TH) {
            if (mode >= 1) __hip_atomic_store(mine + i, (float)(r * 1000 + wg), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            else mine[i] = (float)(r * 1000 + wg);
        }
        if (mode == 0) __threadfence();                 // release: the slab must be visible to every XCD before the arrival
        else __builtin_amdgcn_s_waitcnt(0);             // (mode 1: write-through stores, only their completion is awaited)
        __syncthreads();
        if (mode == 2) {
            // hierarchical: the 32 workgroups of an XCD (blockIdx % 8) arrive on their XCD's counter; the last of them arrives on the
            // global counter; the last of those eight publishes the round in eight per-XCD flags (one 128-byte line each) that the
            // XCD's workgroups poll
            if (tid == 0) {
                const int x = wg & 7;
                unsigned long long* xc = counter + 16 * (1 + x);
                unsigned long long* flag = counter + 16 * (9 + x);
                const unsigned long long a = __hip_atomic_fetch_add(xc, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (a + 1 == (unsigned long long)r * (WG / 8)) {
                    const unsigned long long g = __hip_atomic_fetch_add(counter, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if (g + 1 == (unsigned long long)r * 8)
                        for (int y = 0; y < 8; ++y) __hip_atomic_store(counter + 16 * (9 + y), (unsigned long long)r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
                unsigned spins = 0;
                while (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (unsigned long long)r && spins < 2000000u) {
                    __builtin_amdgcn_s_sleep(1);
                    ++spins;
                }
                if (spins >= 2000000u) bad += 1000;
            }
        } else if (tid == 0) {
            __hip_atomic_fetch_add(counter, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            unsigned spins = 0;
            while (__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (unsigned long long)r * WG && spins < 2000000u) {
                __builtin_amdgcn_s_sleep(1);
                ++spins;
            }
            if (spins >= 2000000u) bad += 1000;
        }
        __syncthreads();
        if (mode == 0) __threadfence();                 // acquire
        // read 16 floats of each of 16 other workgroups' slabs
        const float* other = data + (long)(r & 1) * WG * SLAB + (long)((wg + 1 + (tid >> 4)) % WG) * SLAB + (tid & 15) * 61 % SLAB;
        float v = (mode >= 1) ? __hip_atomic_load(other, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : *other;
        if (v != (float)(r * 1000 + (wg + 1 + (tid >> 4)) % WG)) ++bad;
    }
    if (bad) atomicAdd(errors, bad);
}
int main() {
    float* data; unsigned long long* counter; int* errors;
    hipMalloc(&data, 2L * WG * SLAB * 4); hipMalloc(&counter, 8 * 16 * 17); hipMalloc(&errors, 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int mode = 0; mode < 3; ++mode)
        for (int rounds : {1, 101, 401}) {
            hipMemset(counter, 0, 8 * 16 * 17); hipMemset(errors, 0, 4); hipMemset(data, 0, 2L * WG * SLAB * 4);
            hipDeviceSynchronize();
            hipEventRecord(e0);
            hipLaunchKernelGGL(rounds_kernel, dim3(WG), dim3(TH), 0, 0, data, counter, rounds, errors, mode);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            int err; hipMemcpy(&err, errors, 4, hipMemcpyDeviceToHost);
            printf("%s: %3d rounds %8.1f us total, %6.2f us per round, errors %d\n", mode == 2 ? "write-through, per-XCD counters + flags" : mode ? "write-through stores / atomic loads   " : "fence release / acquire               ",
                   rounds, ms * 1e3, ms * 1e3 / rounds, err);
        }
    return 0;
}
