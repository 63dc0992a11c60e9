// Cost of global atomics and of short streaming kernels on MI355X (8 XCDs, one L2 each):
// how long a kernel takes when every workgroup ends with atomics on the same address(es),
// returning or not, versus plain stores -- evidence for the histogram / counter layouts in
// scan.hip and features.hip.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
__global__ void k_empty(int) {}
// mode 0: one no-return atomicAdd per block on ONE address; 1: returning; 2: per-block address
// 3: K no-return atomics per block on the same K addresses; 4: 16 replicas (b & 15)
__global__ void k_atom(unsigned long long *p, int mode, int K, unsigned long long *sink)
{
    if (threadIdx.x >= (unsigned)K && mode == 3) return;
    if (threadIdx.x != 0 && mode != 3) return;
    if (mode == 0) atomicAdd(p, 1ull);
    else if (mode == 1) sink[blockIdx.x] = atomicAdd(p, 1ull);
    else if (mode == 2) atomicAdd(p + blockIdx.x * 8, 1ull);
    else if (mode == 3) atomicAdd(p + threadIdx.x * 8, 1ull);
    else if (mode == 4) atomicAdd(p + (blockIdx.x & 15) * 8, 1ull);
    else if (mode == 5) p[blockIdx.x * 8] = 1ull;
}
// streaming read of n doubles: ITEMS independent loads per thread, one block per tile
template <int ITEMS> __global__ void k_read(const double *v, int64_t n, double *out)
{
    double s = 0;
    const int64_t base = (int64_t)blockIdx.x * blockDim.x * ITEMS;
#pragma unroll
    for (int j = 0; j < ITEMS; ++j) { int64_t t = base + j * blockDim.x + threadIdx.x; if (t < n) s += v[t]; }
    if (s == 12345.678) out[0] = s;
}
int main()
{
    hipStream_t st; (void)hipStreamCreate(&st);
    unsigned long long *p, *sink; (void)hipMalloc(&p, 1 << 22); (void)hipMalloc(&sink, 1 << 22);
    (void)hipMemset(p, 0, 1 << 22);
    hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    auto timeit = [&](auto launch, const char *name) {
        for (int r = 0; r < 5; ++r) launch();
        (void)hipEventRecord(a, st);
        for (int r = 0; r < 50; ++r) launch();
        (void)hipEventRecord(b, st); (void)hipEventSynchronize(b);
        float ms; (void)hipEventElapsedTime(&ms, a, b);
        printf("%-60s %.2f us / launch\n", name, ms * 1000 / 50);
    };
    timeit([&] { k_empty<<<1, 64, 0, st>>>(0); }, "empty kernel, 1 block");
    timeit([&] { k_empty<<<625, 256, 0, st>>>(0); }, "empty kernel, 625 blocks");
    char nm[128];
    for (int blocks : {128, 256, 625, 1024, 4096}) {
        for (int mode : {0, 1, 2, 4, 5}) {
            static const char *mn[] = {"1 addr no-return", "1 addr returning", "own addr", "", "16 replicas", "plain store"};
            snprintf(nm, sizeof nm, "%5d blocks, 1 atomic/block, %s", blocks, mn[mode]);
            timeit([&] { k_atom<<<blocks, 64, 0, st>>>(p, mode, 1, sink); }, nm);
        }
        for (int K : {8, 64}) {
            snprintf(nm, sizeof nm, "%5d blocks, %d atomics/block on the same %d addrs", blocks, K, K);
            timeit([&] { k_atom<<<blocks, 64, 0, st>>>(p, 3, K, sink); }, nm);
        }
    }
    double *v, *o; const int64_t n = 1280000; (void)hipMalloc(&v, n * 8); (void)hipMalloc(&o, 64); (void)hipMemset(v, 0, n * 8);
    timeit([&] { k_read<8><<<(n + 2047) / 2048, 256, 0, st>>>(v, n, o); }, "read 10 MB: 625 blocks x 256 thr x 8 loads");
    timeit([&] { k_read<4><<<(n + 1023) / 1024, 256, 0, st>>>(v, n, o); }, "read 10 MB: 1250 blocks x 256 thr x 4 loads");
    timeit([&] { k_read<2><<<(n + 511) / 512, 256, 0, st>>>(v, n, o); }, "read 10 MB: 2500 blocks x 256 thr x 2 loads");
    timeit([&] { k_read<1><<<(n + 255) / 256, 256, 0, st>>>(v, n, o); }, "read 10 MB: 5000 blocks x 256 thr x 1 load");
    timeit([&] { k_read<8><<<(n + 8191) / 8192, 1024, 0, st>>>(v, n, o); }, "read 10 MB: 157 blocks x 1024 thr x 8 loads");
    return 0;
}
