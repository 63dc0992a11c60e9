// What does a SIMD's second wave cost an f32 MFMA stream?  512-thread workgroups, one per CU; waves 0-3 issue REP x 64
// dependent v_mfma_f32_32x32x2_f32 (one chain), waves 4-7 (same SIMDs) do `mode` meanwhile:
//   0 nothing (exit)   1 independent v_fma_f32   2 dependent v_fma_f32 chain   3 ds_read_b128   4 v_cmp/v_cndmask mix
//   5 s_sleep loop   6 MFMA stream too (two streams on one pipe)
// Prints cycles per MFMA of wave 0 and partner instructions per MFMA time.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
#define REP 200
template <int MODE, int YIELD = 0> __global__ __launch_bounds__(512, 2) void k(float *out, long long *cyc, int n_partner, int reps)
{
    __shared__ float4 lds[1024];
    const int wave = threadIdx.x >> 6;
    if (threadIdx.x < 1024) lds[threadIdx.x] = float4{1.f, 2.f, 3.f, 4.f};
    __syncthreads();
    if (wave < 4) {
        if (YIELD == 5 || YIELD == 600) __builtin_amdgcn_s_setprio(0);
        f32x16 acc, acc1, acc2, acc3;
        for (int r = 0; r < 16; ++r) { acc[r] = 0.f; acc1[r] = 0.f; acc2[r] = 0.f; acc3[r] = 0.f; }
        float a = threadIdx.x * 0.001f, b = 1.0001f;
        float y[8] = {1, 2, 3, 4, 5, 6, 7, 8};
        bf16x8 ab, bb;
        for (int j = 0; j < 8; ++j) { ab[j] = (__bf16)(a + j); bb[j] = (__bf16)(b * j); }
        long long t0 = clock64();
        for (int rep = 0; rep < reps; ++rep) {
#pragma unroll
            for (int u = 0; u < 64; ++u) {
                if (YIELD >= 200 && YIELD < 600) {
                    const int own = YIELD % 100;
                    const bool bf = (YIELD / 100) & 1;          // 3xx, 5xx: bf16
                    const int nacc = YIELD >= 400 ? 4 : 2;      // 4xx, 5xx: four accumulators
                    f32x16 &c = (u % nacc) == 0 ? acc : (u % nacc) == 1 ? acc1 : (u % nacc) == 2 ? acc2 : acc3;
                    if (bf) c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ab, bb, c, 0, 0, 0);
                    else c = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
#pragma unroll
                    for (int q = 0; q < own; ++q) y[q & 7] = fmaf(y[q & 7], 1.0001f, 0.25f);
                    continue;
                }
                if (YIELD == 9 || YIELD >= 100) { acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ab, bb, acc, 0, 0, 0); if (YIELD == 600) continue; }
                else acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
                if (YIELD >= 100) {   // YIELD - 100 independent VALU ops of the SAME wave beside the bf16 MFMA
#pragma unroll
                    for (int q = 0; q < YIELD - 100; ++q) y[q & 7] = fmaf(y[q & 7], 1.0001f, 0.25f);
                }
                if (YIELD >= 10 && YIELD < 100) {   // YIELD - 10 independent VALU ops of the SAME wave in the MFMA's shadow
#pragma unroll
                    for (int q = 0; q < YIELD - 10; ++q) y[q & 7] = fmaf(y[q & 7], 1.0001f, 0.25f);
                }
                if (YIELD == 1) __builtin_amdgcn_s_sleep(1);
                if (YIELD == 2 && (u & 1)) __builtin_amdgcn_s_sleep(1);
                if (YIELD == 3 && (u & 3) == 3) __builtin_amdgcn_s_sleep(2);
                if (YIELD == 4 && (u & 3) == 3) __builtin_amdgcn_s_sleep(3);
            }
        }
        float s = 0;
        for (int r = 0; r < 16; ++r) s += acc[r] + acc1[r] + acc2[r] + acc3[r];
        for (int q = 0; q < 8; ++q) s += y[q];
        long long t1 = clock64();
        out[blockIdx.x * 512 + threadIdx.x] = s;
        if (threadIdx.x == 0) cyc[blockIdx.x * 2] = t1 - t0;
    } else {
        if (YIELD == 5 || YIELD == 600) __builtin_amdgcn_s_setprio(3);
        long long t0 = clock64();
        float x0 = threadIdx.x, x1 = 1.5f, x2 = 2.5f, x3 = 3.5f, s = 0;
        int cnt = 0;
        if (MODE == 1)
            for (int i = 0; i < n_partner; ++i) {
#pragma unroll
                for (int u = 0; u < 16; ++u) { x0 = fmaf(x0, 1.0001f, 0.5f); x1 = fmaf(x1, 1.0001f, 0.5f); x2 = fmaf(x2, 1.0001f, 0.5f); x3 = fmaf(x3, 1.0001f, 0.5f); }
                cnt += 64;
            }
        if (MODE == 2)
            for (int i = 0; i < n_partner; ++i) {
#pragma unroll
                for (int u = 0; u < 64; ++u) x0 = fmaf(x0, 1.0001f, 0.5f);
                cnt += 64;
            }
        if (MODE == 3)
            for (int i = 0; i < n_partner; ++i) {
#pragma unroll
                for (int u = 0; u < 16; ++u) { float4 v = lds[(threadIdx.x * 7 + u * 64 + i) & 1023]; s += v.x + v.w; }
                cnt += 16;
            }
        if (MODE == 4)
            for (int i = 0; i < n_partner; ++i) {
#pragma unroll
                for (int u = 0; u < 16; ++u) { x0 = x0 < x1 ? x2 : x0 + 1.f; x1 = x1 < x2 ? x3 : x1 + 1.f; x2 = fmaxf(x2, x0); x3 = x3 + x1; }
                cnt += 64;
            }
        if (MODE == 5)
            for (int i = 0; i < n_partner; ++i) { __builtin_amdgcn_s_sleep(8); cnt += 1; }
        if (MODE == 6) {
            f32x16 acc;
            for (int r = 0; r < 16; ++r) acc[r] = 0.f;
            for (int i = 0; i < n_partner; ++i) {
#pragma unroll
                for (int u = 0; u < 64; ++u) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(x0, x1, acc, 0, 0, 0);
                cnt += 64;
            }
            for (int r = 0; r < 16; ++r) s += acc[r];
        }
        long long t1 = clock64();
        out[blockIdx.x * 512 + threadIdx.x] = x0 + x1 + x2 + x3 + s;
        if (threadIdx.x == 256) { cyc[blockIdx.x * 2 + 1] = t1 - t0; out[0] = (float)cnt; }
    }
}
template <int MODE, int YIELD = 0> void run(const char *name, int n_partner, int reps = REP)
{
    float *out; long long *cyc, h[512];
    hipMalloc(&out, 256 * 512 * 4); hipMalloc(&cyc, sizeof(h)); hipMemset(cyc, 0, sizeof(h));
    k<MODE, YIELD><<<256, 512>>>(out, cyc, n_partner, reps);
    hipDeviceSynchronize();
    hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
    double a = 0, b = 0;
    for (int i = 0; i < 256; ++i) { a += h[2 * i]; b += h[2 * i + 1]; }
    printf("%-28s MFMA wave (%4d x 64): %7.1f cycles per MFMA, %9.0f total   partner: %9.0f cycles for %d x its body = %.1f per body\n", name, reps, reps ? a / 256 / (reps * 64.0) : 0.0, a / 256, b / 256, n_partner, n_partner ? b / 256 / n_partner : 0.0);
    hipFree(out); hipFree(cyc);
}
int main()
{
    run<0>("partner exits", 0);
    // body = 64 VALU (modes 1, 2, 4) / 16 ds_read_b128 (mode 3); first alone (MFMA waves exit at once), then under a long MFMA stream
    run<1>("indep fma x4, alone", 300, 0);
    run<1>("indep fma x4, beside MFMA", 300, 400);
    run<2>("dependent fma, alone", 300, 0);
    run<2>("dependent fma, beside MFMA", 300, 400);
    run<3>("ds_read_b128, alone", 300, 0);
    run<3>("ds_read_b128, beside MFMA", 300, 400);
    run<4>("cmp/cndmask mix, alone", 300, 0);
    run<4>("cmp/cndmask mix, beside MFMA", 300, 400);
    run<6>("MFMA stream too", 200, 200);
    // the streaming wave yields the issue port: s_sleep between its MFMAs / priorities
    run<4, 1>("mix beside MFMA+sleep1 each", 300, 400);
    run<4, 2>("mix beside MFMA+sleep1 per 2", 300, 400);
    run<4, 3>("mix beside MFMA+sleep2 per 4", 300, 400);
    run<4, 4>("mix beside MFMA+sleep3 per 4", 300, 400);
    run<4, 5>("mix beside MFMA, setprio 3/0", 300, 400);
    run<3, 1>("ds_read beside MFMA+sleep1", 300, 400);
    run<0, 14>("MFMA + 4 own fma each, alone", 0, 400);
    run<0, 18>("MFMA + 8 own fma each, alone", 0, 400);
    run<0, 22>("MFMA + 12 own fma each, alone", 0, 400);
    run<0, 26>("MFMA + 16 own fma each, alone", 0, 400);
    run<0, 42>("MFMA + 32 own fma each, alone", 0, 400);
    run<4, 18>("mix beside MFMA + 8 own fma", 300, 400);
    run<0, 9>("bf16 MFMA chain alone", 0, 400);
    run<4, 9>("mix beside bf16 MFMA chain", 300, 400);
    run<3, 9>("ds_read beside bf16 MFMA", 300, 400);
    run<1, 9>("indep fma beside bf16 MFMA", 300, 400);
    run<0, 104>("bf16 MFMA + 4 own fma each", 0, 400);
    run<0, 108>("bf16 MFMA + 8 own fma each", 0, 400);
    run<0, 200>("f32 MFMA 2 acc alone", 0, 400);
    run<0, 204>("f32 MFMA 2 acc + 4 own fma", 0, 400);
    run<0, 208>("f32 MFMA 2 acc + 8 own fma", 0, 400);
    run<4, 200>("mix beside f32 MFMA 2 acc", 300, 400);
    run<0, 300>("bf16 MFMA 2 acc alone", 0, 400);
    run<0, 304>("bf16 MFMA 2 acc + 4 own fma", 0, 400);
    run<0, 308>("bf16 MFMA 2 acc + 8 own fma", 0, 400);
    run<4, 300>("mix beside bf16 MFMA 2 acc", 300, 400);
    run<0, 504>("bf16 MFMA 4 acc + 4 own fma", 0, 400);
    run<4, 500>("mix beside bf16 MFMA 4 acc", 300, 400);
    run<4, 400>("mix beside f32 MFMA 4 acc", 300, 400);
    run<4, 600>("mix(prio 3) beside bf16 chain(prio 0)", 300, 400);
    run<3, 600>("ds_read(prio 3) beside bf16 chain", 300, 400);
    run<1, 600>("indep fma(prio 3) beside bf16 chain", 300, 400);
    run<0, 1>("MFMA+sleep1 alone", 0, 400);
    run<0, 4>("MFMA+sleep3 per 4 alone", 0, 400);
    return 0;
}
