// Where does bounds_dad_features (features.hip: k_features) spend its time at 127 M pairs?
// Variants on synthetic data of the same shape (N = 16000, all pairs, 24 anchors).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include <cmath>
#pragma clang fp contract(off)
template <int MODE> __global__ __launch_bounds__(256) void k(const int2 *__restrict__ ij, int64_t n, const double *__restrict__ Dt,
                                                            int64_t nx, int na, const int32_t *__restrict__ cA,
                                                            const int32_t *__restrict__ ar, double *__restrict__ lb,
                                                            double *__restrict__ ub, double *__restrict__ dad,
                                                            uint8_t *__restrict__ anc, uint8_t *__restrict__ ncm)
{
    int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n) return;
    const int2 q = ij[p];
    const int i = q.x, j = q.y;
    double l = 0.0, u = INFINITY;
    if (MODE >= 1) {
        for (int a0 = 0; a0 < na; a0 += 8) {
            double di[8], dj[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const size_t row = (size_t)min(a0 + e, na - 1) * nx;
                di[e] = MODE == 3 ? 1.0 : Dt[row + i];
                dj[e] = Dt[row + j];
            }
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                l = fmax(l, fabs(di[e] - dj[e]));
                u = fmin(u, di[e] + dj[e]);
            }
        }
    }
    lb[p] = l;
    ub[p] = u;
    if (MODE == 1) dad[p] = (Dt[(size_t)cA[j] * nx + i] + Dt[(size_t)cA[i] * nx + j]) / 2;
    else dad[p] = l + u;
    const uint8_t isa = MODE == 1 ? ((ar[i] >= 0) | (ar[j] >= 0)) : (uint8_t)(i == 0);
    anc[p] = isa;
    ncm[p] = !isa;
}
// anchors interleaved in pairs: D2[a/2][j] = {D[a][j], D[a+1][j]}: 12 loads of 16 B per pair instead of 24 of 8 B
__global__ __launch_bounds__(256) void k_pairs(const int2 *__restrict__ ij, int64_t n, const double2 *__restrict__ D2, int64_t nx,
                                              int na2, double *__restrict__ lb, double *__restrict__ ub, double *__restrict__ dad,
                                              uint8_t *__restrict__ anc, uint8_t *__restrict__ ncm)
{
    int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n) return;
    const int2 q = ij[p];
    const int i = q.x, j = q.y;
    const int i0 = __builtin_amdgcn_readfirstlane(i);
    double l = 0.0, u = INFINITY;
    for (int a0 = 0; a0 < na2; a0 += 4) {
        double2 di[4], dj[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const size_t row = (size_t)min(a0 + e, na2 - 1) * nx;
            di[e] = D2[row + i0];
            dj[e] = D2[row + j];
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            l = fmax(l, fabs(di[e].x - dj[e].x)); u = fmin(u, di[e].x + dj[e].x);
            l = fmax(l, fabs(di[e].y - dj[e].y)); u = fmin(u, di[e].y + dj[e].y);
        }
    }
    lb[p] = l; ub[p] = u; dad[p] = l + u;
    const uint8_t isa = (uint8_t)(i == 0);
    anc[p] = isa; ncm[p] = !isa;
}
// MODE 4: a workgroup owns 256 consecutive pairs of ONE row i (rows are long): D[.][i] in
// scalar registers via LDS broadcast, D[.][j] staged once through LDS as [a][256]
int main()
{
    const int64_t nx = 16000; const int na = 24;
    const int64_t n = nx * (nx - 1) / 2;
    std::vector<int2> h((size_t)n);
    { int64_t p = 0; for (int i = 0; i < nx; ++i) for (int j = i + 1; j < nx; ++j) h[p++] = make_int2(i, j); }
    std::vector<double> D((size_t)na * nx); for (auto &x : D) x = rand() / (double)RAND_MAX;
    std::vector<int32_t> cA((size_t)nx), ar((size_t)nx, -1); for (auto &x : cA) x = rand() % na;
    int2 *ij; double *Dt, *lb, *ub, *dad; int32_t *dcA, *dar; uint8_t *anc, *ncm;
    hipMalloc(&ij, n * 8); hipMalloc(&Dt, na * nx * 8); hipMalloc(&lb, n * 8); hipMalloc(&ub, n * 8); hipMalloc(&dad, n * 8);
    hipMalloc(&dcA, nx * 4); hipMalloc(&dar, nx * 4); hipMalloc(&anc, n); hipMalloc(&ncm, n);
    hipMemcpy(ij, h.data(), n * 8, hipMemcpyHostToDevice); hipMemcpy(Dt, D.data(), na * nx * 8, hipMemcpyHostToDevice);
    hipMemcpy(dcA, cA.data(), nx * 4, hipMemcpyHostToDevice); hipMemcpy(dar, ar.data(), nx * 4, hipMemcpyHostToDevice);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    auto run = [&](auto kern, const char *name) {
        const int blocks = (int)((n + 255) / 256);
        for (int r = 0; r < 2; ++r) kern<<<blocks, 256>>>(ij, n, Dt, nx, na, dcA, dar, lb, ub, dad, anc, ncm);
        hipEventRecord(a);
        for (int r = 0; r < 5; ++r) kern<<<blocks, 256>>>(ij, n, Dt, nx, na, dcA, dar, lb, ub, dad, anc, ncm);
        hipEventRecord(b); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b); ms /= 5;
        printf("%-58s %.3f ms  %.0f GB/s algorithmic (34 B/pair)\n", name, ms, n * 34.0 / ms / 1e6);
    };
    run(k<0>, "stream only (read ij, write lb ub dad + 2 masks)");
    run(k<2>, "+ 24-anchor bounds (no cA gathers)");
    run(k<3>, "+ 24-anchor bounds, j side only");
    run(k<1>, "full kernel");
    {
        std::vector<double> D2h((size_t)(na / 2) * nx * 2);
        for (int a = 0; a < na; ++a) for (int64_t j = 0; j < nx; ++j) D2h[((size_t)(a / 2) * nx + j) * 2 + (a & 1)] = D[(size_t)a * nx + j];
        double2 *D2; hipMalloc(&D2, D2h.size() * 8); hipMemcpy(D2, D2h.data(), D2h.size() * 8, hipMemcpyHostToDevice);
        const int blocks = (int)((n + 255) / 256);
        for (int r = 0; r < 2; ++r) k_pairs<<<blocks, 256>>>(ij, n, D2, nx, na / 2, lb, ub, dad, anc, ncm);
        hipEventRecord(a);
        for (int r = 0; r < 5; ++r) k_pairs<<<blocks, 256>>>(ij, n, D2, nx, na / 2, lb, ub, dad, anc, ncm);
        hipEventRecord(b); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b); ms /= 5;
        printf("%-58s %.3f ms  %.0f GB/s algorithmic (34 B/pair)\n", "anchors interleaved in pairs (16-byte gathers), scalar i side", ms, n * 34.0 / ms / 1e6);
    }
    return 0;
}
