// Split-fp16 against split-bf16 dot products on the matrix cores: x = hi + lo, x.y ~ hi.hi + hi.lo + lo.hi with
// v_mfma_f32_32x32x16_f16 / _bf16; error against float64 on data of the streamed form's shape (latent 8-d in 128-d, centred,
// scaled so that max |x| = 2^13) and on data with a wide dynamic range.
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
#define DIM 128
__global__ void k(const float *A, const float *B, float *Cb, float *Ch, float *Cf)
{
    const int lane = threadIdx.x & 63;
    bf16x8 ah[DIM / 16], al[DIM / 16], bh[DIM / 16], bl[DIM / 16];
    f16x8 fah[DIM / 16], fal[DIM / 16], fbh[DIM / 16], fbl[DIM / 16];
    for (int g = 0; g < DIM / 16; ++g)
        for (int j = 0; j < 8; ++j) {
            const int kk = 16 * g + 8 * (lane >> 5) + j;
            const float x = A[(lane & 31) * DIM + kk], y = B[(lane & 31) * DIM + kk];
            __bf16 h = (__bf16)x; ah[g][j] = h; al[g][j] = (__bf16)(x - (float)h);
            h = (__bf16)y; bh[g][j] = h; bl[g][j] = (__bf16)(y - (float)h);
            _Float16 f = (_Float16)x; fah[g][j] = f; fal[g][j] = (_Float16)(x - (float)f);
            f = (_Float16)y; fbh[g][j] = f; fbl[g][j] = (_Float16)(y - (float)f);
        }
    f32x16 acc, acch;
    for (int q = 0; q < 16; ++q) { acc[q] = 0.f; acch[q] = 0.f; }
    for (int g = 0; g < DIM / 16; ++g) {
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[g], bh[g], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[g], bl[g], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[g], bh[g], acc, 0, 0, 0);
        acch = __builtin_amdgcn_mfma_f32_32x32x16_f16(fal[g], fbh[g], acch, 0, 0, 0);
        acch = __builtin_amdgcn_mfma_f32_32x32x16_f16(fah[g], fbl[g], acch, 0, 0, 0);
        acch = __builtin_amdgcn_mfma_f32_32x32x16_f16(fah[g], fbh[g], acch, 0, 0, 0);
    }
    float af[DIM / 2], bf[DIM / 2];
    for (int s = 0; s < DIM / 2; ++s) { af[s] = A[(lane & 31) * DIM + 2 * s + (lane >> 5)]; bf[s] = B[(lane & 31) * DIM + 2 * s + (lane >> 5)]; }
    f32x16 acc2;
    for (int q = 0; q < 16; ++q) acc2[q] = 0.f;
    for (int s = 0; s < DIM / 2; ++s) acc2 = __builtin_amdgcn_mfma_f32_32x32x2f32(af[s], bf[s], acc2, 0, 0, 0);
    for (int q = 0; q < 16; ++q) {
        const int o = ((q & 3) + 8 * (q >> 2) + 4 * (lane >> 5)) * 32 + (lane & 31);
        Cb[o] = acc[q]; Ch[o] = acch[q]; Cf[o] = acc2[q];
    }
}
static void run(const char *name, std::vector<float> &A, std::vector<float> &B)
{
    float *dA, *dB, *dC[3];
    (void)hipMalloc(&dA, A.size() * 4); (void)hipMalloc(&dB, B.size() * 4);
    for (auto &p : dC) (void)hipMalloc(&p, 4096);
    (void)hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice); (void)hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice);
    k<<<1, 64>>>(dA, dB, dC[0], dC[1], dC[2]);
    std::vector<float> C[3];
    for (int v = 0; v < 3; ++v) { C[v].resize(1024); (void)hipMemcpy(C[v].data(), dC[v], 4096, hipMemcpyDeviceToHost); }
    double worst[3] = {0, 0, 0}, rms[3] = {0, 0, 0};
    for (int i = 0; i < 32; ++i)
        for (int j = 0; j < 32; ++j) {
            double ref = 0, na = 0, nb = 0;
            for (int kk = 0; kk < DIM; ++kk) { ref += (double)A[i * DIM + kk] * B[j * DIM + kk]; na += (double)A[i * DIM + kk] * A[i * DIM + kk]; nb += (double)B[j * DIM + kk] * B[j * DIM + kk]; }
            for (int v = 0; v < 3; ++v) { const double e = fabs(C[v][i * 32 + j] - ref) / sqrt(na * nb); worst[v] = fmax(worst[v], e); rms[v] += e * e; }
        }
    printf("%-34s max |err| / (|x||y|): split-bf16 2^%.1f  split-fp16 2^%.1f  f32 MFMA 2^%.1f   (rms 2^%.1f / 2^%.1f / 2^%.1f)\n", name,
           log2(worst[0]), log2(worst[1]), log2(worst[2]), log2(sqrt(rms[0] / 1024)), log2(sqrt(rms[1] / 1024)), log2(sqrt(rms[2] / 1024)));
}
int main()
{
    std::vector<float> A(32 * DIM), B(32 * DIM);
    srand(7);
    auto gauss = []() { double s = 0; for (int i = 0; i < 12; ++i) s += rand() / (double)RAND_MAX; return s - 6.0; };
    // latent 8-d in 128-d, scaled to max |x| ~ 2^13
    std::vector<double> W(8 * DIM);
    for (auto &w : W) w = gauss();
    auto fill = [&](std::vector<float> &M, double scale) {
        for (int i = 0; i < 32; ++i) {
            double z[8]; for (auto &t : z) t = gauss();
            for (int kk = 0; kk < DIM; ++kk) { double v = 0.05 * gauss(); for (int q = 0; q < 8; ++q) v += z[q] * W[q * DIM + kk]; M[i * DIM + kk] = (float)(v * scale); }
        }
    };
    fill(A, 1.0); fill(B, 1.0); run("latent data, scale 1", A, B);
    fill(A, 512.0); fill(B, 512.0); run("latent data, scale 512 (max ~2^13)", A, B);
    for (auto &v : A) v = (float)(gauss() * pow(2.0, (rand() % 24) - 10)); for (auto &v : B) v = (float)(gauss() * pow(2.0, (rand() % 24) - 10));
    run("24 octaves of dynamic range", A, B);
    return 0;
}
