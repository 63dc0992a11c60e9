// The stage-1 slab of k_st_knnh (knnh.hip: stream) in isolation: NW waves per workgroup (one or two workgroups per CU), the ring
// prefilled, no LDS-DMA, no barrier -- what does the instruction sequence itself cost?
//   parts (bit mask): 1 operand reads (16 x ds_read_b128)   2 seeds (32 v_sub)   4 MFMAs (16)   8 running maximum + compare + branch
//   hipcc --offload-arch=gfx950 -O3 -o stream_seq stream_seq.hip && ./stream_seq
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8h __attribute__((ext_vector_type(8)));
#define REP 1000
template <int PARTS, bool BARRIER, int DMA = 0> __global__ __launch_bounds__(256, 2) __attribute__((amdgpu_waves_per_eu(2, 2))) void k(float *out, long long *cyc, float thr, const char *src = nullptr, long long nslabs = 0)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float *ring = reinterpret_cast<float *>(smem);
    for (int i = threadIdx.x; i < 3 * 4096; i += blockDim.x) ring[i] = 1e-3f * (float)(i & 255);
    __syncthreads();
    const int lane = threadIdx.x & 63, col = lane & 31, half = lane >> 5;
    constexpr int G = 8;
    f16x8h ah[G];
    for (int g = 0; g < G; ++g)
        for (int j = 0; j < 8; ++j) ah[g][j] = (_Float16)(0.01f * (float)(lane + g + j));
    float hqr[16];
    for (int r = 0; r < 16; ++r) hqr[r] = thr + (float)r;
    f32x16 a0, a1, b0a, b1a;
    for (int r = 0; r < 16; ++r) { a0[r] = 0.f; a1[r] = 0.f; b0a[r] = 0.f; b1a[r] = 0.f; }
    int hits = 0;
    auto stream = [&](int slot, f32x16 &c0, f32x16 &c1, const f32x16 &p0, const f32x16 &p1) __attribute__((always_inline)) {
        const float4 *base0 = reinterpret_cast<const float4 *>(&ring[slot * 4096]) + col * 16;
        const float4 *base1 = base0 + 32 * 16;
        const int gsw = half ^ (col & 15);
        float4 b0[G], b1[G];
        if (PARTS & 1) {
#pragma unroll
            for (int g = 0; g < G; ++g) b0[g] = base0[(2 * g) ^ gsw];
#pragma unroll
            for (int g = 0; g < G; ++g) b1[g] = base1[(2 * g) ^ gsw];
        } else {
#pragma unroll
            for (int g = 0; g < G; ++g) { b0[g] = float4{1.f, 2.f, 3.f, (float)g}; b1[g] = b0[g]; asm volatile("" : "+v"(b0[g].x), "+v"(b1[g].y)); }
        }
        const float n0 = -0.5f * b0[0].x, n1 = -0.5f * b1[0].y;
        float mx0 = 0.f, mx1 = 0.f;
        if (PARTS & 2) {
#pragma unroll
            for (int r = 0; r < 16; ++r) c0[r] = n0 - hqr[r];
        }
#pragma unroll
        for (int m = 0; m < 2 * G; m += 2) {
#pragma unroll
            for (int h2 = 0; h2 < 2; ++h2) {
                const int mm = m + h2;
                if (PARTS & 4) {
                    if (mm < G) c0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[mm], __builtin_bit_cast(f16x8h, b0[mm]), c0, 0, 0, 0);
                    else c1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[mm - G], __builtin_bit_cast(f16x8h, b1[mm - G]), c1, 0, 0, 0);
                } else {
                    if (mm < G) c0[mm] += b0[mm].x; else c1[mm - G] += b1[mm - G].x;
                }
                if (mm == 0 && (PARTS & 2)) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) c1[r] = n1 - hqr[r];
                }
            }
            if (PARTS & 8) {
                const int kq = m >> 1;
                if (kq == 0) { mx0 = __builtin_fmaxf(__builtin_fmaxf(p0[0], p0[1]), p0[2]); mx1 = __builtin_fmaxf(__builtin_fmaxf(p1[0], p1[1]), p1[2]); }
                else if (kq < 7) { mx0 = __builtin_fmaxf(__builtin_fmaxf(mx0, p0[2 * kq + 1]), p0[2 * kq + 2]); mx1 = __builtin_fmaxf(__builtin_fmaxf(mx1, p1[2 * kq + 1]), p1[2 * kq + 2]); }
                else mx0 = __builtin_fmaxf(__builtin_fmaxf(mx0, p0[15]), __builtin_fmaxf(mx1, p1[15]));
            }
            if (m < G) asm volatile("" : "+v"(c0));
            else asm volatile("" : "+v"(c1));
        }
        if ((PARTS & 8) && __builtin_expect(__ballot(mx0 > 1e30f) != 0, 0)) ++hits;
    };
    // DMA: after every slab the wave waits for everything it has outstanding, then requests its quarter of the slab after the next
    // (4 x global_load_lds_dwordx4, 1 KB each, into the ring slot before the current one) -- DMA 1: sources walk a 1 GB array (HBM),
    // DMA 2: one 16 KB slab again and again (L2)
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    uint32_t loff[4];
    for (int i = 0; i < 4; ++i) {
        const int u = (wave * 4 + i) * 64 + lane;
        const int c = u >> 4, x = u & 15;
        loff[i] = (uint32_t)(c * 512 + ((x ^ (c & 15)) << 4)) - 1024u * i;
    }
    const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) unsigned char *)smem;
    unsigned long long pos = (unsigned long long)blockIdx.x * 7919ull;
    auto dma = [&](int slotv) __attribute__((always_inline)) {
        pos = DMA == 2 ? (unsigned long long)blockIdx.x : (unsigned long long)((uint32_t)pos * 1664525u + 1013904223u);   // (32-bit arithmetic: a 64-bit modulo is hundreds of cycles)
        const unsigned long long slab = DMA == 2 ? (pos & 1023) : ((pos >> 9) & (unsigned long long)(nslabs - 1));
        const unsigned long long v = (unsigned long long)(uintptr_t)(src + slab * 32768ull);
        const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)v), hi = __builtin_amdgcn_readfirstlane((uint32_t)(v >> 32));
        const char *sp = reinterpret_cast<const char *>((uintptr_t)(((unsigned long long)hi << 32) | lo));
        const uint32_t dst = lds0 + (uint32_t)(__builtin_amdgcn_readfirstlane(slotv) * 16384 + wave * 4096);
#ifndef DMA_MOD
#define DMA_MOD ""
#endif
        asm volatile("s_mov_b32 m0, %5\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %4" DMA_MOD "\n\t"
                     "global_load_lds_dwordx4 %1, %4 offset:1024" DMA_MOD "\n\t"
                     "global_load_lds_dwordx4 %2, %4 offset:2048" DMA_MOD "\n\t"
                     "global_load_lds_dwordx4 %3, %4 offset:3072" DMA_MOD
                     : : "v"(loff[0]), "v"(loff[1]), "v"(loff[2]), "v"(loff[3]), "s"(sp), "s"(dst) : "memory");
    };
    long long t_wait = 0, t_issue = 0;
    const long long t0 = clock64();
    int slot = 0;
    for (int rep = 0; rep < REP; ++rep) {
        if (BARRIER) { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
        stream(slot, a0, a1, b0a, b1a);
        if (DMA) { const long long w0 = clock64(); __builtin_amdgcn_s_waitcnt(0x0F70); const long long w1 = clock64(); dma(slot == 0 ? 2 : slot - 1); const long long w2 = clock64(); t_wait += w1 - w0; t_issue += w2 - w1; }
        slot = slot == 2 ? 0 : slot + 1;
        if (BARRIER) { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
        stream(slot, b0a, b1a, a0, a1);
        if (DMA) { __builtin_amdgcn_s_waitcnt(0x0F70); dma(slot == 0 ? 2 : slot - 1); }
        slot = slot == 2 ? 0 : slot + 1;
    }
    if (DMA) __builtin_amdgcn_s_waitcnt(0x0F70);
    const long long t1 = clock64();
    float s = (float)hits;
    for (int r = 0; r < 16; ++r) s += a0[r] + a1[r] + b0a[r] + b1a[r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) { cyc[blockIdx.x] = t1 - t0; cyc[512 + blockIdx.x] = t_wait; cyc[1024 + blockIdx.x] = t_issue; }
}
static char *g_src = nullptr;
template <int PARTS, bool BARRIER, int DMA = 0> static void run(const char *name, int blocks, float *out, long long *cyc)
{
    hipFuncSetAttribute((const void *)k<PARTS, BARRIER, DMA>, hipFuncAttributeMaxDynamicSharedMemorySize, 70000);
    k<PARTS, BARRIER, DMA><<<blocks, 256, 70000>>>(out, cyc, 1e30f, g_src, (1ll << 30) / 32768);
    if (hipDeviceSynchronize() != hipSuccess) { printf("launch failed\n"); return; }
    long long h[512];
    hipMemcpy(h, cyc, sizeof(long long) * blocks, hipMemcpyDeviceToHost);
    double s = 0;
    for (int i = 0; i < blocks; ++i) s += (double)h[i];
    s /= blocks;
    printf("%-52s %d workgroup(s) per CU: %7.1f cycles per slab", name, blocks / 256, s / (2 * REP));
    if (DMA) {
        long long hw[512], hi[512];
        hipMemcpy(hw, cyc + 512, sizeof(long long) * blocks, hipMemcpyDeviceToHost); hipMemcpy(hi, cyc + 1024, sizeof(long long) * blocks, hipMemcpyDeviceToHost);
        double a = 0, b = 0;
        for (int i = 0; i < blocks; ++i) { a += (double)hw[i]; b += (double)hi[i]; }
        printf("   (of every second slab: wait %.0f, requests %.0f)", a / blocks / REP, b / blocks / REP);
    }
    printf("\n");
}
int main()
{
    float *out; long long *cyc;
    hipMalloc(&out, 512 * 256 * 4); hipMalloc(&cyc, 3 * 512 * 8);
    hipMalloc(&g_src, (1ull << 30) + 65536); hipMemset(g_src, 0, (1ull << 30) + 65536);
    for (int blocks : {256, 512}) {
        run<15, false>("reads + seeds + MFMAs + test", blocks, out, cyc);
        run<15, true>("reads + seeds + MFMAs + test, barrier per slab", blocks, out, cyc);
        run<15, true, 1>("... + wait + 4 LDS-DMA requests per slab (HBM)", blocks, out, cyc);
        run<15, true, 2>("... + wait + 4 LDS-DMA requests per slab (L2)", blocks, out, cyc);
        run<15, false, 1>("no barrier, wait + 4 LDS-DMA requests (HBM)", blocks, out, cyc);
        run<7, false>("reads + seeds + MFMAs", blocks, out, cyc);
        run<5, false>("reads + MFMAs", blocks, out, cyc);
        run<4, false>("MFMAs alone", blocks, out, cyc);
        run<1, false>("reads alone (+ 16 adds)", blocks, out, cyc);
        run<6, false>("seeds + MFMAs", blocks, out, cyc);
        run<12, false>("MFMAs + test", blocks, out, cyc);
    }
    return 0;
}
