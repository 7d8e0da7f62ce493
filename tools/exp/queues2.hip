// Do streams made with hipExtStreamCreateWithCUMask (all CUs enabled) get a hardware queue of their own?  Same probe as queues.hip.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
__global__ void spin(unsigned long long ticks) { const unsigned long long t0 = __builtin_amdgcn_s_memrealtime(); while (__builtin_amdgcn_s_memrealtime() - t0 < ticks) __builtin_amdgcn_s_sleep(8); }
__global__ void nop(int *p) { if (p && threadIdx.x == 12345) *p = 1; }
int main()
{
    struct S { const char *name; int kind; hipStream_t s; };
    std::vector<S> st = {{"plain0", 0, 0}, {"plain1", 0, 0}, {"plain2", 0, 0}, {"plain3", 0, 0}, {"plain4", 0, 0},
                         {"mask0", 1, 0}, {"mask1", 1, 0}, {"mask2", 1, 0}, {"mask3", 1, 0}, {"mask4", 1, 0}, {"mask5", 1, 0}};
    std::vector<uint32_t> all(8, 0xffffffffu);
    for (auto &x : st) {
        if (x.kind == 0) CK(hipStreamCreateWithFlags(&x.s, hipStreamNonBlocking));
        else CK(hipExtStreamCreateWithCUMask(&x.s, 8, all.data()));
    }
    for (auto &x : st) hipLaunchKernelGGL(nop, dim3(1), dim3(64), 0, x.s, nullptr);
    CK(hipDeviceSynchronize());
    // launch latency: nop submit -> complete, per stream kind
    for (auto &x : st) {
        double best = 1e9, sum = 0;
        for (int r = 0; r < 50; ++r) {
            const auto t0 = std::chrono::steady_clock::now();
            hipLaunchKernelGGL(nop, dim3(1), dim3(64), 0, x.s, nullptr);
            CK(hipStreamSynchronize(x.s));
            const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
            best = us < best ? us : best; sum += us;
        }
        printf("%-8s launch+sync: min %.1f us avg %.1f us\n", x.name, best, sum / 50);
    }
    printf("%-10s", "A \\ B");
    for (auto &b : st) printf("%8s", b.name);
    printf("\n");
    for (auto &a : st) {
        printf("%-10s", a.name);
        for (auto &b : st) {
            if (a.s == b.s) { printf("%8s", "-"); continue; }
            hipLaunchKernelGGL(spin, dim3(1), dim3(64), 0, a.s, 30000ull);
            const auto t0 = std::chrono::steady_clock::now();
            hipLaunchKernelGGL(nop, dim3(1), dim3(64), 0, b.s, nullptr);
            CK(hipStreamSynchronize(b.s));
            const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
            CK(hipDeviceSynchronize());
            printf("%8.0f", us);
        }
        printf("\n");
    }
    // a 256-workgroup kernel on a masked stream vs a plain one: dispatch rate
    for (int k : {0, 5}) {
        hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        CK(hipEventRecord(e0, st[k].s));
        for (int r = 0; r < 20; ++r) hipLaunchKernelGGL(spin, dim3(256), dim3(1024), 0, st[k].s, 2000ull); // 20 us each
        CK(hipEventRecord(e1, st[k].s));
        CK(hipStreamSynchronize(st[k].s));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        printf("%s: 20 x (256 workgroups x 1024 threads spinning 20 us): %.1f us per launch\n", st[k].name, ms * 1000 / 20);
    }
    return 0;
}
