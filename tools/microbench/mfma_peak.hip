// Sustained v_mfma_f32_32x32x2_f32 rate on gfx950 (evidence for the streamed kernel's roofline;
// not part of the library).  hipcc --offload-arch=gfx950 -O3 tools/microbench/mfma_peak.hip -o tools/microbench/mfma_peak
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
#define ITER 4096
template <int CHAINS> __global__ __launch_bounds__(256) void k(float *out, float x)
{
    f32x16 acc[CHAINS];
    for (int c = 0; c < CHAINS; ++c)
        for (int r = 0; r < 16; ++r) acc[c][r] = 0.f;
    float a = x + threadIdx.x, b = x * 0.5f + threadIdx.x;
    for (int it = 0; it < ITER; ++it) {
#pragma unroll
        for (int u = 0; u < 16; ++u)
#pragma unroll
            for (int c = 0; c < CHAINS; ++c) acc[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[c], 0, 0, 0);
    }
    float s = 0;
    for (int c = 0; c < CHAINS; ++c)
        for (int r = 0; r < 16; ++r) s += acc[c][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int CHAINS> void run(int blocks, float *d)
{
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    k<CHAINS><<<blocks, 256>>>(d, 1.f);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    k<CHAINS><<<blocks, 256>>>(d, 2.f);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms;
    (void)hipEventElapsedTime(&ms, e0, e1);
    double flops = (double)blocks * 4 * ITER * 16 * CHAINS * (32.0 * 32 * 2 * 2);
    printf("chains %d blocks %5d (%d waves/SIMD): %.2f ms  %.1f TFLOP/s  (%.1f cycles/MFMA/SIMD at 2.4 GHz)\n", CHAINS, blocks,
           blocks / 256, ms, flops / ms / 1e9, ms * 1e-3 * 2.4e9 / ((double)ITER * 16 * CHAINS * (blocks / 256.0)));
}
int main()
{
    float *d;
    (void)hipMalloc(&d, 4096 * 256 * 4);
    for (int blocks : {256, 512, 1024}) { run<1>(blocks, d); run<2>(blocks, d); }
    return 0;
}
