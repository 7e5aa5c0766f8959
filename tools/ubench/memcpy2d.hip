// D2H of the first K sections of every column: hipMemcpy2DAsync against one flat copy (pinned destination).
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
int main() {
    const size_t C = 257, S = 200, SEC = 32;
    char *d, *h;
    CK(hipMalloc(&d, C * S * SEC));
    CK(hipHostMalloc(&h, C * S * SEC));
    hipStream_t st; CK(hipStreamCreate(&st));
    auto time = [&](auto f, const char* name) {
        for (int i = 0; i < 20; i++) f();
        (void)hipStreamSynchronize(st);
        auto t0 = std::chrono::steady_clock::now();
        const int n = 200;
        for (int i = 0; i < n; i++) { f(); (void)hipStreamSynchronize(st); }
        double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / n;
        printf("%-40s %8.1f us per copy + sync\n", name, us);
    };
    time([&] { (void)hipMemcpyAsync(h, d, C * S * SEC, hipMemcpyDeviceToHost, st); }, "flat 1.64 MB");
    for (size_t K : {32, 64, 100}) {
        char name[64]; snprintf(name, sizeof name, "2D %zu rows x %zu B (pitch %zu)", C, K * SEC, S * SEC);
        time([&] { (void)hipMemcpy2DAsync(h, K * SEC, d, S * SEC, K * SEC, C, hipMemcpyDeviceToHost, st); }, name);
    }
    time([&] { (void)hipMemcpyAsync(h, d, C * 64 * SEC, hipMemcpyDeviceToHost, st); }, "flat 0.53 MB");
    time([&] { (void)hipMemcpyAsync(h, d, 64, hipMemcpyDeviceToHost, st); }, "flat 64 B");
    return 0;
}
