// What a dependent launch costs on gfx950 as a function of HOW ITS ARGUMENTS ARRIVE: a chain of N kernels on one stream, each reading
// what the previous one wrote (256 workgroups x 256 threads, one load + one store per thread), with
//   mode 0  no kernel arguments at all: pointers come from a __device__ table (PC-relative s_load, a line that stays cached)
//   mode 1  64 bytes of kernel arguments   (a fresh kernarg block per launch, written by the host into device memory)
//   mode 2  512 bytes of kernel arguments  (what the decoder loop's by-value argument structs look like)
//   mode 3  64 bytes of arguments, of which the kernel only reads a pointer to a device-resident copy of the 512-byte block
// `HIP_FORCE_DEV_KERNARG=0` moves the kernarg blocks back to host memory (the bench loses 12.7 ms per step that way).
//   hipcc --offload-arch=gfx950 -O3 tools/kernarg_probe.hip -o /tmp/kernarg_probe && /tmp/kernarg_probe
#include <hip/hip_runtime.h>
#include <cstdio>
constexpr int WG = 256, TH = 256;
struct Small { float* a; float* b; long n; int step; int pad[9]; };                  // 64 B
struct Big { float* a; float* b; long n; int step; int pad[9]; long more[56]; };      // 512 B
static_assert(sizeof(Small) == 64 && sizeof(Big) == 512, "sizes");
__device__ float* g_tab[2];
__device__ Big g_big;

// every kernel also idles ~4 us (constant 100 MHz clock) so that the chain is bound by the GPU, not by the host's launch rate
__device__ __forceinline__ void idle() {
    const unsigned long long t0 = wall_clock64();
    while (wall_clock64() - t0 < 400ull) __builtin_amdgcn_s_sleep(2);
}
__device__ __forceinline__ void body(float* a, float* b, int step, long extra) {
    const long i = (long)blockIdx.x * TH + threadIdx.x;
    float* src = (step & 1) ? b : a;
    float* dst = (step & 1) ? a : b;
    const float v = src[i] + 1.0f + (float)extra;      // the arguments are needed BEFORE the idle phase: their latency is not hidden
    if (v < 0.f) return;
    idle();
    dst[i] = v;
}
__global__ __launch_bounds__(TH) void k_noarg() {
    // the step parity comes from the data itself: no argument, no counter
    const long i = (long)blockIdx.x * TH + threadIdx.x;
    float* a = g_tab[0]; float* b = g_tab[1];
    const float va = a[i], vb = b[i];
    if (va < 0.f) return;
    idle();
    if (va >= vb) b[i] = va + 1.0f; else a[i] = vb + 1.0f;
}
// mode 4: no kernel arguments; a device-side launch counter selects this launch's own 512-byte block out of a table uploaded once
// (a block nobody has touched before: cold like a kernarg block, but not written by the host just now); mode 5: the same, and every
// launch touches the NEXT launch's block from one wave per XCD so that it waits in that XCD's L2
__device__ const Big* g_blocks;
__device__ unsigned g_count;
template <bool PREFETCH>
__global__ __launch_bounds__(TH) void k_table() {
    const unsigned n = g_count;
    const Big* p = g_blocks + n;
    float* a = p->a; float* b = p->b;
    const int step = p->step;
    const long extra = p->more[55];
    const long i = (long)blockIdx.x * TH + threadIdx.x;
    float* src = (step & 1) ? b : a;
    float* dst = (step & 1) ? a : b;
    const float v = src[i] + 1.0f + (float)extra;
    if (v < 0.f) return;
    if (PREFETCH && blockIdx.x < 8 && threadIdx.x < 8) {        // 8 x 64 B = the next block, once per XCD
        const float t = reinterpret_cast<const float*>(p + 1)[threadIdx.x * 16];
        if (t == 123456.f) dst[0] = t;
    }
    idle();
    dst[i] = v;
    if (blockIdx.x == 0 && threadIdx.x == 0) g_count = n + 1;    // visible to the next launch: the kernel boundary orders it
}
__global__ __launch_bounds__(TH) void k_small(Small s) { body(s.a, s.b, s.step, 0); }
__global__ __launch_bounds__(TH) void k_big(Big s) { body(s.a, s.b, s.step, s.more[55]); }
__global__ __launch_bounds__(TH) void k_indirect(Small s, const Big* p) { body(p->a, p->b, s.step, p->more[55]); }

int main() {
    float *a, *b;
    hipMalloc(&a, WG * TH * 4); hipMalloc(&b, WG * TH * 4);
    hipMemset(a, 0, WG * TH * 4); hipMemset(b, 0, WG * TH * 4);
    float* tab[2] = {a, b};
    hipMemcpyToSymbol(HIP_SYMBOL(g_tab), tab, sizeof(tab));
    Big big = {}; big.a = a; big.b = b; big.n = WG * TH;
    hipMemcpyToSymbol(HIP_SYMBOL(g_big), &big, sizeof(big));
    Big* dbig; hipGetSymbolAddress((void**)&dbig, HIP_SYMBOL(g_big));
    Small sm = {}; sm.a = a; sm.b = b; sm.n = WG * TH;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int N = 4000;
    Big* hblocks = new Big[N + 1];
    for (int s = 0; s <= N; ++s) { hblocks[s] = big; hblocks[s].step = s; }
    Big* dblocks; hipMalloc(&dblocks, sizeof(Big) * (N + 1));
    hipMemcpy(dblocks, hblocks, sizeof(Big) * (N + 1), hipMemcpyHostToDevice);
    hipMemcpyToSymbol(HIP_SYMBOL(g_blocks), &dblocks, sizeof(dblocks));
    const char* names[6] = {"no arguments (__device__ table)", "64 B kernarg", "512 B kernarg", "64 B kernarg + device-resident block",
                            "no kernarg, per-launch block from a table", "  ... and the next block touched ahead"};
    for (int rep = 0; rep < 2; ++rep)
        for (int mode = 0; mode < 6; ++mode) {
            unsigned zero = 0; hipMemcpyToSymbol(HIP_SYMBOL(g_count), &zero, sizeof(zero));
            hipMemset(a, 0, WG * TH * 4); hipMemset(b, 0, WG * TH * 4);
            hipDeviceSynchronize();
            hipEventRecord(e0, 0);
            for (int s = 0; s < N; ++s) {
                sm.step = s; big.step = s;
                if (mode == 0) hipLaunchKernelGGL(k_noarg, dim3(WG), dim3(TH), 0, 0);
                else if (mode == 1) hipLaunchKernelGGL(k_small, dim3(WG), dim3(TH), 0, 0, sm);
                else if (mode == 2) hipLaunchKernelGGL(k_big, dim3(WG), dim3(TH), 0, 0, big);
                else if (mode == 3) hipLaunchKernelGGL(k_indirect, dim3(WG), dim3(TH), 0, 0, sm, (const Big*)dbig);
                else if (mode == 4) hipLaunchKernelGGL(k_table<false>, dim3(WG), dim3(TH), 0, 0);
                else hipLaunchKernelGGL(k_table<true>, dim3(WG), dim3(TH), 0, 0);
            }
            hipEventRecord(e1, 0);
            hipEventSynchronize(e1);
            float ms = 0; hipEventElapsedTime(&ms, e0, e1);
            float h[2]; hipMemcpy(&h[0], a, 4, hipMemcpyDeviceToHost); hipMemcpy(&h[1], b, 4, hipMemcpyDeviceToHost);
            if (rep) printf("%-40s %.3f us per dependent launch   (check %g)\n", names[mode], 1e3 * ms / N, h[0] > h[1] ? h[0] : h[1]);
        }
    return 0;
}
