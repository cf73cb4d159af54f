// Does read-only data survive in the per-XCD L2 from one kernel launch to the next?  Streams a buffer of N MB once per launch
// (each workgroup always the same slice; optionally in alternating direction) and reports the rate over back-to-back launches.
//   hipcc --offload-arch=gfx950 -O3 tools/l2_probe.hip -o /tmp/l2_probe && /tmp/l2_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f4 __attribute__((ext_vector_type(4)));
// workgroups >= keep_wgs read with the non-temporal hint: does the L2 then keep the others' slices from launch to launch?
__global__ __launch_bounds__(256) void stream_nt_kernel(const f4* __restrict__ buf, long n4_per_wg, float* out, int keep_wgs) {
    const f4* p = buf + (long)blockIdx.x * n4_per_wg;
    f4 acc = {0.f, 0.f, 0.f, 0.f};
    const long iters = n4_per_wg / 256;
    if ((int)blockIdx.x < keep_wgs) {
        for (long i = 0; i < iters; ++i) acc += p[i * 256 + threadIdx.x];
    } else {
        for (long i = 0; i < iters; ++i) acc += __builtin_nontemporal_load(p + i * 256 + threadIdx.x);
    }
    if (acc.x + acc.y + acc.z + acc.w == 12345.678f) out[blockIdx.x] = acc.x;
}
__global__ __launch_bounds__(256) void stream_kernel(const float4* __restrict__ buf, long n4_per_wg, float* out, int reverse) {
    const float4* p = buf + (long)blockIdx.x * n4_per_wg;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    const long iters = n4_per_wg / 256;
    for (long i = 0; i < iters; ++i) {
        const long ii = reverse ? iters - 1 - i : i;
        const float4 x = p[ii * 256 + threadIdx.x];
        acc.x += x.x; acc.y += x.y; acc.z += x.z; acc.w += x.w;
    }
    if (acc.x + acc.y + acc.z + acc.w == 12345.678f) out[blockIdx.x] = acc.x;
}
int main() {
    const int WG = 2048;
    float* out; hipMalloc(&out, WG * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int alt = 0; alt < 2; ++alt)
    for (long mb : {4L, 8L, 16L, 24L, 32L, 48L, 64L, 96L, 128L, 192L, 256L, 384L, 1024L}) {
        const long bytes = mb << 20, n4 = bytes / 16, per = n4 / WG;
        float4* buf; hipMalloc(&buf, bytes); hipMemset(buf, 0, bytes);
        for (int w = 0; w < 3; ++w) hipLaunchKernelGGL(stream_kernel, dim3(WG), dim3(256), 0, 0, buf, per, out, alt ? (w & 1) : 0);
        hipDeviceSynchronize();
        const int R = 40;
        hipEventRecord(e0);
        for (int r = 0; r < R; ++r) hipLaunchKernelGGL(stream_kernel, dim3(WG), dim3(256), 0, 0, buf, per, out, alt ? ((r + 1) & 1) : 0);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("%s %5ld MB: %7.2f us per launch, %6.2f TB/s\n", alt ? "alternating" : "same-order ", mb, ms * 1e3 / R, bytes / (ms * 1e-3 / R) / 1e12);
        hipFree(buf);
    }
    for (long mb : {64L, 96L, 128L})
        for (long keep_mb : {0L, 8L, 16L, 24L, 28L, 32L}) {
            const long bytes = mb << 20, n4 = bytes / 16, per = n4 / WG;
            const int keep = (int)(WG * keep_mb / mb);
            f4* buf; hipMalloc(&buf, bytes); hipMemset(buf, 0, bytes);
            for (int w = 0; w < 3; ++w) hipLaunchKernelGGL(stream_nt_kernel, dim3(WG), dim3(256), 0, 0, buf, per, out, keep);
            hipDeviceSynchronize();
            const int R = 40;
            hipEventRecord(e0);
            for (int r = 0; r < R; ++r) hipLaunchKernelGGL(stream_nt_kernel, dim3(WG), dim3(256), 0, 0, buf, per, out, keep);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            printf("nt beyond %3ld of %4ld MB: %7.2f us per launch, %6.2f TB/s\n", keep_mb, mb, ms * 1e3 / R, bytes / (ms * 1e-3 / R) / 1e12);
            hipFree(buf);
        }
    return 0;
}
