// Measured VALU issue rates on gfx950 (tools/microbench: evidence for the roofline peak used
// for the Levenshtein kernel; not part of the library).
//   hipcc --offload-arch=gfx950 -O3 tools/microbench/valu_peak.hip -o tools/microbench/valu_peak
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

#define ITER 16384
template <int OP> __global__ __launch_bounds__(256) void k(uint32_t *out, uint32_t seed)
{
    uint32_t x[8];
    float f[8];
    for (int i = 0; i < 8; ++i) { x[i] = threadIdx.x * 2654435761u + i + seed; f[i] = (float)x[i]; }
    uint32_t y = seed | 1u;
    float g = 1.0001f;
    for (int it = 0; it < ITER; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            if (OP == 0) x[i] = x[i] + y;                                  // v_add_u32
            if (OP == 1) x[i] = x[i] ^ y;                                  // v_xor_b32
            if (OP == 2) x[i] = (x[i] & y) | (x[(i + 1) & 7]);             // v_bitop3 / v_and_or
            if (OP == 3) f[i] = __builtin_fmaf(f[i], g, 0.5f);             // v_fma_f32
            if (OP == 4) x[i] = __builtin_amdgcn_alignbit(x[i], y, 31);    // v_alignbit_b32
            if (OP == 5) x[i] = (x[i] + y) ^ x[i];                         // dependent add -> xor pair
            if (OP == 6) x[i] = x[(i + 1) & 7] + y;                        // independent, dst != src
            if (OP == 7) x[i] = (i & 1) ? (x[i] ^ y) : (x[i] + y);         // in place, alternating opcodes
            if (OP == 8) x[i] = (x[i] + y) ^ y;                            // pair through a temp, 1 VGPR source each
            if (OP == 10) x[i] = x[i] + x[(i + 1) & 7];                    // two VGPR sources, in place
        }
        if (OP == 9) {   // one 16-deep dependent chain per iteration (latency)
#pragma unroll
            for (int j = 0; j < 8; ++j) x[0] = (x[0] + y) ^ x[1];
        }
        if (OP != 3) asm volatile("" : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]), "+v"(x[4]), "+v"(x[5]), "+v"(x[6]), "+v"(x[7]));
        else asm volatile("" : "+v"(f[0]), "+v"(f[1]), "+v"(f[2]), "+v"(f[3]), "+v"(f[4]), "+v"(f[5]), "+v"(f[6]), "+v"(f[7]));
    }
    uint32_t s = 0;
    for (int i = 0; i < 8; ++i) s += x[i] + (uint32_t)f[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int OP> void run(const char *name, int blocks, int ops_per_inner, uint32_t *d)
{
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    k<OP><<<blocks, 256>>>(d, 1);
    hipDeviceSynchronize();
    hipEventRecord(a);
    k<OP><<<blocks, 256>>>(d, 2);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms = 0;
    hipEventElapsedTime(&ms, a, b);
    double wave_instr = (double)blocks * 4 * ITER * 8 * ops_per_inner;
    printf("%-28s blocks %6d  %.3f ms  %.1f G wave-instr/s  %.2f T lane-ops/s\n", name, blocks, ms, wave_instr / ms / 1e6,
           wave_instr * 64 / ms / 1e9);
}

int main()
{
    uint32_t *d;
    hipMalloc(&d, 256 * 64 * 256 * 4);
    for (int blocks : {256, 256 * 2, 256 * 8}) {   // 1, 2, 8 waves per SIMD
        run<0>("v_add_u32", blocks, 1, d);
        run<1>("v_xor_b32", blocks, 1, d);
        run<2>("v_and_or / bitop3", blocks, 1, d);
        run<4>("v_alignbit_b32", blocks, 1, d);
        run<5>("add->xor dependent pair", blocks, 2, d);
        run<3>("v_fma_f32", blocks, 1, d);
        run<6>("add, dst != src", blocks, 1, d);
        run<7>("add/xor alternating in place", blocks, 1, d);
        run<8>("add->xor via temp (1 vgpr src)", blocks, 2, d);
        run<10>("add two vgpr sources", blocks, 1, d);
        run<9>("16-deep dependent chain", blocks, 2, d);
    }
    return 0;
}
