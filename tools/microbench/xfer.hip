// Small-transfer latency on the box: pageable vs pinned staging (evidence for ann_d2h / ann_h2d).
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <vector>
__global__ void k_touch(double *p) { p[threadIdx.x] += 1.0; }
int main()
{
    hipStream_t st;
    (void)hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
    double *d;
    (void)hipMalloc(&d, 1 << 22);
    void *pin;
    (void)hipHostMalloc(&pin, 1 << 22, hipHostMallocDefault);
    std::vector<char> page(1 << 22);
    auto now = [] { return std::chrono::steady_clock::now(); };
    for (size_t bytes : {64ul, 4096ul, 40000ul, 160000ul, 640000ul, 4000000ul}) {
        for (int mode = 0; mode < 4; ++mode) {
            const int reps = 200;
            double tot = 0;
            for (int r = -20; r < reps; ++r) {
                k_touch<<<1, 64, 0, st>>>(d);
                auto t0 = now();
                if (mode == 0) { (void)hipMemcpyAsync(page.data(), d, bytes, hipMemcpyDeviceToHost, st); (void)hipStreamSynchronize(st); }
                if (mode == 1) { (void)hipMemcpyAsync(pin, d, bytes, hipMemcpyDeviceToHost, st); (void)hipStreamSynchronize(st); memcpy(page.data(), pin, bytes); }
                if (mode == 2) { (void)hipMemcpyAsync(d, page.data(), bytes, hipMemcpyHostToDevice, st); (void)hipStreamSynchronize(st); }
                if (mode == 3) { memcpy(pin, page.data(), bytes); (void)hipMemcpyAsync(d, pin, bytes, hipMemcpyHostToDevice, st); (void)hipStreamSynchronize(st); }
                if (r >= 0) tot += std::chrono::duration<double, std::micro>(now() - t0).count();
            }
            static const char *names[4] = {"D2H pageable", "D2H pinned+memcpy", "H2D pageable", "H2D memcpy+pinned"};
            printf("%8zu B  %-20s %.1f us\n", bytes, names[mode], tot / reps);
        }
    }
    // kernel launch + sync only
    double tot = 0;
    for (int r = 0; r < 200; ++r) { auto t0 = now(); k_touch<<<1, 64, 0, st>>>(d); (void)hipStreamSynchronize(st); tot += std::chrono::duration<double, std::micro>(now() - t0).count(); }
    printf("launch + sync: %.1f us\n", tot / 200);
    return 0;
}
