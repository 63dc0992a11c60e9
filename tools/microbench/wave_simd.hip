// Which SIMD does wave w of a 512-thread workgroup run on?  (HW_REG_HW_ID: wave_id [3:0], simd_id [5:4], cu_id [11:8])
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ __launch_bounds__(512) void k(unsigned *out)
{
    extern __shared__ char big[];
    unsigned id;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(id));
    if ((threadIdx.x & 63) == 0) out[blockIdx.x * 8 + (threadIdx.x >> 6)] = id;
    if (threadIdx.x == 9999) big[0] = 1;
}
int main()
{
    unsigned *d, h[64 * 8];
    hipMalloc(&d, sizeof(h));
    hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, 120 * 1024);
    k<<<64, 512, 120 * 1024>>>(d);
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    for (int b = 0; b < 12; ++b) {
        printf("block %2d cu %2u: simd of waves 0..7 =", b, (h[b * 8] >> 8) & 15);
        for (int w = 0; w < 8; ++w) printf(" %u", (h[b * 8 + w] >> 4) & 3);
        printf("   wave slots =");
        for (int w = 0; w < 8; ++w) printf(" %u", h[b * 8 + w] & 15);
        printf("\n");
    }
    return 0;
}
