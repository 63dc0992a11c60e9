// Split-bf16 dot products on the matrix cores: x = hi + lo (two bf16), x.y ~ hi.hi + hi.lo + lo.hi with
// v_mfma_f32_32x32x16_bf16 -- operand layout check (asymmetric data), error against float64, rate against the
// exact v_mfma_f32_32x32x2_f32 stream.   hipcc --offload-arch=gfx950 -O3 bf16_split.hip -o bf16_split
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
#define DIM 128

__device__ __forceinline__ void split(float x, __bf16 &hi, __bf16 &lo)
{
    hi = (__bf16)x;
    lo = (__bf16)(x - (float)hi);
}

// one wave: C[32][32] = A[32][DIM] . B[32][DIM]^T  (rows of A = "rows", rows of B = "columns")
__global__ void k_split(const float *A, const float *B, float *C, float *Cf, int reps, long long *cyc)
{
    const int lane = threadIdx.x & 63;
    bf16x8 ah[DIM / 16], al[DIM / 16], bh[DIM / 16], bl[DIM / 16];
    for (int g = 0; g < DIM / 16; ++g)
        for (int j = 0; j < 8; ++j) {
            const int k = 16 * g + 8 * (lane >> 5) + j;
            __bf16 h, l;
            split(A[(lane & 31) * DIM + k], h, l); ah[g][j] = h; al[g][j] = l;
            split(B[(lane & 31) * DIM + k], h, l); bh[g][j] = h; bl[g][j] = l;
        }
    f32x16 acc;
    long long t0 = clock64();
    for (int r = 0; r < reps; ++r) {
        for (int q = 0; q < 16; ++q) acc[q] = 0.f;
#pragma unroll
        for (int g = 0; g < DIM / 16; ++g) {
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[g], bh[g], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[g], bl[g], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[g], bh[g], acc, 0, 0, 0);
        }
        asm volatile("" : "+v"(acc));
    }
    long long t1 = clock64();
    for (int q = 0; q < 16; ++q) C[((q & 3) + 8 * (q >> 2) + 4 * (lane >> 5)) * 32 + (lane & 31)] = acc[q];
    // exact f32 stream
    float af[DIM / 2], bf[DIM / 2];
    for (int s = 0; s < DIM / 2; ++s) { af[s] = A[(lane & 31) * DIM + 2 * s + (lane >> 5)]; bf[s] = B[(lane & 31) * DIM + 2 * s + (lane >> 5)]; }
    long long t2 = clock64();
    f32x16 acc2;
    for (int r = 0; r < reps; ++r) {
        for (int q = 0; q < 16; ++q) acc2[q] = 0.f;
#pragma unroll
        for (int s = 0; s < DIM / 2; ++s) acc2 = __builtin_amdgcn_mfma_f32_32x32x2f32(af[s], bf[s], acc2, 0, 0, 0);
        asm volatile("" : "+v"(acc2));
    }
    long long t3 = clock64();
    for (int q = 0; q < 16; ++q) Cf[((q & 3) + 8 * (q >> 2) + 4 * (lane >> 5)) * 32 + (lane & 31)] = acc2[q];
    if (lane == 0) { cyc[0] = t1 - t0; cyc[1] = t3 - t2; }
}

int main()
{
    std::vector<float> A(32 * DIM), B(32 * DIM);
    srand(7);
    for (auto &v : A) v = (float)(rand() % 2001 - 1000) / 37.f;
    for (auto &v : B) v = (float)(rand() % 2001 - 1000) / 91.f + 3.f;   // asymmetric
    float *dA, *dB, *dC, *dCf; long long *dcy;
    hipMalloc(&dA, A.size() * 4); hipMalloc(&dB, B.size() * 4); hipMalloc(&dC, 4096); hipMalloc(&dCf, 4096); hipMalloc(&dcy, 16);
    hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice); hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice);
    const int reps = 2000;
    k_split<<<1, 64>>>(dA, dB, dC, dCf, reps, dcy);
    std::vector<float> C(1024), Cf(1024); long long cy[2];
    hipMemcpy(C.data(), dC, 4096, hipMemcpyDeviceToHost); hipMemcpy(Cf.data(), dCf, 4096, hipMemcpyDeviceToHost); hipMemcpy(cy, dcy, 16, hipMemcpyDeviceToHost);
    double worst = 0, worstf = 0;
    for (int i = 0; i < 32; ++i)
        for (int j = 0; j < 32; ++j) {
            double ref = 0, na = 0, nb = 0;
            for (int k = 0; k < DIM; ++k) { ref += (double)A[i * DIM + k] * B[j * DIM + k]; na += (double)A[i * DIM + k] * A[i * DIM + k]; nb += (double)B[j * DIM + k] * B[j * DIM + k]; }
            worst = fmax(worst, fabs(C[i * 32 + j] - ref) / sqrt(na * nb));
            worstf = fmax(worstf, fabs(Cf[i * 32 + j] - ref) / sqrt(na * nb));
        }
    printf("split-bf16: max |err| / (|x||y|) = %.3g (2^%.1f)   f32 MFMA: %.3g\n", worst, log2(worst), worstf);
    printf("cycles per 32x32x%d block: split-bf16 (24 MFMA) %.0f, f32 (64 MFMA) %.0f\n", DIM, (double)cy[0] / reps, (double)cy[1] / reps);
    return 0;
}
