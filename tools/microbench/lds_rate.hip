// LDS read rate per CU on gfx950 for the tile kernels' operand pattern: NW waves of one workgroup per CU each read a 16 KB slab
// (64 columns x 256 B, 16-byte units XOR-swizzled by the column as in knnh.hip) REP times.
//   mode 0: 16 x ds_read_b128 per slab and wave (the kernel's form)      mode 1: 32 x ds_read_b64      mode 2: b128, linear addresses
//   mode 3: 16 x ds_read_b128, the two half-waves reading the SAME columns' two k-halves (the MFMA B layout) -- as mode 0
// Prints cycles per slab-read of a wave and bytes per cycle and CU.
//   hipcc --offload-arch=gfx950 -O3 -o lds_rate lds_rate.hip && ./lds_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define REP 2000
template <int MODE> __global__ __launch_bounds__(512) void k(float *out, long long *cyc)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float4 *ring = reinterpret_cast<float4 *>(smem);
    for (int i = threadIdx.x; i < 3 * 1024; i += blockDim.x) ring[i] = float4{1.f, 2.f, 3.f, (float)i};
    __syncthreads();
    const int lane = threadIdx.x & 63, col = lane & 31, half = lane >> 5;
    const int gsw = half ^ (col & 15);
    float4 acc = float4{0.f, 0.f, 0.f, 0.f};
    float2 acc2 = float2{0.f, 0.f};
    const long long t0 = clock64();
    for (int rep = 0; rep < REP; ++rep) {
        const int slot = rep % 3;
        const float4 *base0 = ring + slot * 1024 + col * 16;
        const float4 *base1 = base0 + 32 * 16;
        if (MODE == 0 || MODE == 3) {
            float4 b[16];
#pragma unroll
            for (int g = 0; g < 8; ++g) b[g] = base0[(2 * g) ^ gsw];
#pragma unroll
            for (int g = 0; g < 8; ++g) b[8 + g] = base1[(2 * g) ^ gsw];
#pragma unroll
            for (int g = 0; g < 16; ++g) { acc.x += b[g].x; acc.y += b[g].y; acc.z += b[g].z; acc.w += b[g].w; }
        } else if (MODE == 1) {
            float2 b[32];
#pragma unroll
            for (int g = 0; g < 8; ++g) {
                const float2 *p = reinterpret_cast<const float2 *>(base0 + ((2 * g) ^ gsw));
                b[2 * g] = p[0]; b[2 * g + 1] = p[1];
            }
#pragma unroll
            for (int g = 0; g < 8; ++g) {
                const float2 *p = reinterpret_cast<const float2 *>(base1 + ((2 * g) ^ gsw));
                b[16 + 2 * g] = p[0]; b[16 + 2 * g + 1] = p[1];
            }
#pragma unroll
            for (int g = 0; g < 32; ++g) { acc2.x += b[g].x; acc2.y += b[g].y; }
        } else {
            float4 b[16];
#pragma unroll
            for (int g = 0; g < 16; ++g) b[g] = ring[slot * 1024 + g * 64 + lane];
#pragma unroll
            for (int g = 0; g < 16; ++g) { acc.x += b[g].x; acc.y += b[g].y; acc.z += b[g].z; acc.w += b[g].w; }
        }
    }
    const long long t1 = clock64();
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc.x + acc.y + acc.z + acc.w + acc2.x + acc2.y;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
template <int MODE> static void run(const char *name, int nw, float *out, long long *cyc)
{
    hipFuncSetAttribute((const void *)k<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, 49152);
    k<MODE><<<256, 64 * nw, 49152>>>(out, cyc);
    hipDeviceSynchronize();
    long long h[256];
    hipMemcpy(h, cyc, sizeof h, hipMemcpyDeviceToHost);
    double s = 0;
    for (int i = 0; i < 256; ++i) s += (double)h[i];
    s /= 256;
    // clock64 = s_memtime: a constant 100 MHz-class counter on gfx9?  report raw ticks per slab too
    printf("%-28s waves %d: %8.1f ticks per slab-read and wave, %6.1f B per tick and CU\n", name, nw, s / REP, (double)nw * 16384.0 * REP / s);
}
int main()
{
    float *out; long long *cyc;
    hipMalloc(&out, 256 * 512 * 4); hipMalloc(&cyc, 256 * 8);
    for (int nw : {1, 2, 4, 8}) {
        run<0>("b128 swizzled (kernel)", nw, out, cyc);
        run<1>("b64 x 2 swizzled", nw, out, cyc);
        run<2>("b128 linear", nw, out, cyc);
    }
    return 0;
}
