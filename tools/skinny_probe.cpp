// Standalone decomposition probe for skinny_fwd_kernel (development aid): hipcc -DSKINNY_PROBE_* variants.
#include "../multi_speaker_tts_amd/csrc/skinny.hip"
#include <vector>
namespace mstts { char* err_buf() { static char b[512]; return b; } int set_err(int c, const char*, ...) { return c; } }
int main() {
    const int M = 32, N = 4096, K = 1792, KS = 8, KL = K / KS;
    float *X, *W, *P;
    hipMalloc(&X, sizeof(float) * M * K); hipMalloc(&W, sizeof(float) * (size_t)K * N); hipMalloc(&P, sizeof(float) * KS * M * N);
    hipMemset(X, 0, sizeof(float) * M * K); hipMemset(W, 0, sizeof(float) * (size_t)K * N);
    size_t lds = sizeof(float) * 32 * (KL + 4); if (lds < sizeof(float) * 4 * 32 * 65) lds = sizeof(float) * 4 * 32 * 65;
    hipFuncSetAttribute((const void*)mstts::skinny_fwd_kernel<14, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    dim3 grid(N / 64, KS, 1);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0, 0);
        for (int i = 0; i < 1000; ++i)
            hipLaunchKernelGGL((mstts::skinny_fwd_kernel<14, true>), grid, dim3(256), lds, 0, X, (long)K, W, (long)N, P, (long)M * N, M, N, K, KL);
        hipEventRecord(e1, 0); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("%s: %.2f us per launch\n", VARIANT, ms);
    }
    return 0;
}
