// What does one column step of the Levenshtein kernel cost, piece by piece?  (evidence for
// DESIGN.md's kernel notes; not part of the library)
//   hipcc --offload-arch=gfx950 -O3 tools/microbench/lev_body.hip -o tools/microbench/lev_body
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

#define ITER 8192
__device__ __forceinline__ uint32_t shr1(uint32_t v) { return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x138, 0xf, 0xf, false); }
__device__ __forceinline__ uint32_t row_shr1(uint32_t v) { return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, false); }

// V bit0: DPP wave_shr carries; bit1: validity selects; bit2: LDS match-mask reads; bit3: row_shr instead of wave_shr
template <int R, int V> __global__ __launch_bounds__(64) void k(uint32_t *out, uint32_t seed, int n)
{
    __shared__ uint32_t pm[32 * 64];
    __shared__ uint16_t txt[ITER + 64];
    for (int i = threadIdx.x; i < 32 * 64; i += 64) pm[i] = i * 2654435761u + seed;
    for (int i = threadIdx.x; i < ITER + 64; i += 64) txt[i] = (uint16_t)(((i * 7 + seed) & 31) * 256);
    __syncthreads();
    const int w = threadIdx.x & 15;
    uint32_t vp[R], vn[R], eq[R];
    for (int r = 0; r < R; ++r) { vp[r] = ~0u; vn[r] = 0; eq[r] = seed * (r + 3) + threadIdx.x; }
    uint32_t out_hp = 0, out_hn = 0;
    const unsigned char *pmw = (const unsigned char *)(pm + (threadIdx.x & 31) * R % 32);
    const uint16_t *tp = txt + 32 - w;
    uint32_t c1 = tp[1];
    typedef uint32_t vecR __attribute__((ext_vector_type(R)));
    vecR eqv = *(const vecR *)(pmw + tp[0]);
    for (int k = 0; k < ITER; ++k) {
        uint32_t c2 = 0;
        vecR eqn = eqv;
        if (V & 4) { c2 = tp[k + 2]; eqn = *(const vecR *)(pmw + c1); }
        uint32_t hp_up = out_hp, hn_up = out_hn;
        if (V & 1) { hp_up = (V & 8) ? row_shr1(out_hp) : shr1(out_hp); hn_up = (V & 8) ? row_shr1(out_hn) : shr1(out_hn); }
        __builtin_amdgcn_sched_barrier(0);
        if (V & 1) { hp_up = (w == 0) ? 0x80000000u : hp_up; hn_up = (w == 0) ? 0u : hn_up; }
        const bool valid = (uint32_t)(k - w) < (uint32_t)n;
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const uint32_t e = (V & 4) ? eqv[r] : eq[r];
            const uint32_t x = e | (hn_up >> 31);
            const uint32_t d0 = (((x & vp[r]) + vp[r]) ^ vp[r]) | x | vn[r];
            const uint32_t hp = vn[r] | ~(d0 | vp[r]);
            const uint32_t hn = d0 & vp[r];
            const uint32_t hps = __builtin_amdgcn_alignbit(hp, hp_up, 31);
            const uint32_t hns = __builtin_amdgcn_alignbit(hn, hn_up, 31);
            const uint32_t nvp = hns | ~(d0 | hps), nvn = hps & d0;
            if (V & 2) { vp[r] = valid ? nvp : vp[r]; vn[r] = valid ? nvn : vn[r]; }
            else { vp[r] = nvp; vn[r] = nvn; }
            hp_up = hp; hn_up = hn;
            if (!(V & 4)) eq[r] = __builtin_amdgcn_alignbit(eq[r], eq[r], 7);
        }
        out_hp = hp_up; out_hn = hn_up;
        __builtin_amdgcn_sched_barrier(0);
        eqv = eqn; c1 = c2;
    }
    uint32_t s = 0;
    for (int r = 0; r < R; ++r) s += vp[r] ^ vn[r];
    out[blockIdx.x * 64 + threadIdx.x] = s + out_hp;
}

template <int R, int V> void run(const char *name, uint32_t *d)
{
    for (int wps : {1, 2, 4}) {
        int blocks = 1024 * wps;
        hipEvent_t a, b;
        (void)hipEventCreate(&a); (void)hipEventCreate(&b);
        k<R, V><<<blocks, 64>>>(d, 1, 1 << 30);
        (void)hipDeviceSynchronize();
        (void)hipEventRecord(a);
        k<R, V><<<blocks, 64>>>(d, 2, 1 << 30);
        (void)hipEventRecord(b);
        (void)hipEventSynchronize(b);
        float ms = 0;
        (void)hipEventElapsedTime(&ms, a, b);
        // cycles per column step per SIMD at 2.4 GHz, and per (word x column)
        double cyc = ms * 1e-3 * 2.4e9 / ((double)ITER * wps);
        printf("R=%d %-44s waves/SIMD %d  %.3f ms  %.1f cyc/col/wave  %.1f cyc per word-col\n", R, name, wps, ms, cyc, cyc / R);
    }
}

int main()
{
    uint32_t *d;
    (void)hipMalloc(&d, 1024 * 8 * 64 * 4);
    run<1, 0>("recurrence only", d);
    run<1, 1>("+ wave_shr carries", d);
    run<1, 9>("+ row_shr carries", d);
    run<1, 3>("+ wave_shr + validity", d);
    run<1, 7>("+ wave_shr + validity + LDS", d);
    run<2, 0>("recurrence only", d);
    run<2, 1>("+ wave_shr carries", d);
    run<2, 9>("+ row_shr carries", d);
    run<2, 3>("+ wave_shr + validity", d);
    run<2, 7>("+ wave_shr + validity + LDS", d);
    run<4, 0>("recurrence only", d);
    run<4, 7>("+ wave_shr + validity + LDS", d);
    return 0;
}
