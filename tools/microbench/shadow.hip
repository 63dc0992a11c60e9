// Does VALU work issue in the shadow of an MFMA on gfx950?  One wave per SIMD (256-thread workgroups, one per CU): a chain
// of MFMAs with exactly K independent v_fma_f32 (inline asm, pinned) after each one.  Cycles per MFMA by K.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
template <int BF, int K, int NACC> __global__ __launch_bounds__(256) void k(float *out, long long *cyc)
{
    f32x16 acc[NACC];
    for (int q = 0; q < NACC; ++q)
        for (int r = 0; r < 16; ++r) acc[q][r] = 0.f;
    float a = threadIdx.x * 0.001f, b = 1.0001f;
    bf16x8 ab, bb;
    for (int j = 0; j < 8; ++j) { ab[j] = (__bf16)(a + j); bb[j] = (__bf16)(b * j); }
    float y0 = 1, y1 = 2, y2 = 3, y3 = 4, y4 = 5, y5 = 6, y6 = 7, y7 = 8;
    long long t0 = clock64();
    for (int rep = 0; rep < 100; ++rep) {
#pragma unroll
        for (int u = 0; u < 64; ++u) {
            f32x16 &c = acc[u % NACC];
            if (BF) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(c) : "v"(ab), "v"(bb));
            else asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+v"(c) : "v"(a), "v"(b));
            if (K >= 1) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(y0) : "v"(b));
            if (K >= 2) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(y1) : "v"(b));
            if (K >= 3) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(y2) : "v"(b));
            if (K >= 4) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(y3) : "v"(b));
            if (K >= 5) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(y4) : "v"(b));
            if (K >= 6) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(y5) : "v"(b));
            if (K >= 7) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(y6) : "v"(b));
            if (K >= 8) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(y7) : "v"(b));
            if (K >= 12) { asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(y0) : "v"(b)); asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(y1) : "v"(b));
                           asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(y2) : "v"(b)); asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(y3) : "v"(b)); }
            if (K >= 16) { asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(y4) : "v"(b)); asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(y5) : "v"(b));
                           asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(y6) : "v"(b)); asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(y7) : "v"(b)); }
        }
    }
    long long t1 = clock64();
    float s = y0 + y1 + y2 + y3 + y4 + y5 + y6 + y7;
    for (int q = 0; q < NACC; ++q)
        for (int r = 0; r < 16; ++r) s += acc[q][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
template <int BF, int K, int NACC> void run()
{
    float *out; long long *cyc, h[256];
    (void)hipMalloc(&out, 256 * 256 * 4); (void)hipMalloc(&cyc, sizeof(h));
    k<BF, K, NACC><<<256, 256>>>(out, cyc);
    (void)hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
    double a = 0;
    for (int i = 0; i < 256; ++i) a += h[i];
    printf("%s MFMA, %d accumulator(s), %2d v_fma after each: %6.1f cycles per MFMA\n", BF ? "bf16 32x32x16" : "f32 32x32x2  ", NACC, K, a / 256 / 6400.0);
    (void)hipFree(out); (void)hipFree(cyc);
}
int main()
{
    run<0, 0, 1>(); run<0, 2, 1>(); run<0, 4, 1>(); run<0, 8, 1>(); run<0, 12, 1>(); run<0, 16, 1>();
    run<0, 0, 2>(); run<0, 4, 2>(); run<0, 8, 2>(); run<0, 16, 2>();
    run<1, 0, 1>(); run<1, 2, 1>(); run<1, 4, 1>(); run<1, 8, 1>();
    run<1, 0, 2>(); run<1, 2, 2>(); run<1, 4, 2>(); run<1, 8, 2>();
    run<1, 0, 4>(); run<1, 4, 4>(); run<1, 8, 4>();
    return 0;
}
