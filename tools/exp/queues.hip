// Which streams of a process share a hardware queue (MI355X, ROCm 7.2)?  A 300 us spin kernel goes on stream A, a trivial kernel
// on stream B right behind it: B's kernel completes at once when the two have queues of their own and after the spin when
// they share one.  Streams are made in the order the mapper makes them: copy (greatest), icp0 (mid), grid (least), [icp1 (mid)],
// build (least), plus candidates for the build's stream.
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
    int least, greatest;
    CK(hipDeviceGetStreamPriorityRange(&least, &greatest));
    printf("priority range: least %d greatest %d; GPU_MAX_HW_QUEUES=%s\n", least, greatest, getenv("GPU_MAX_HW_QUEUES") ? getenv("GPU_MAX_HW_QUEUES") : "(unset)");
    const int mid = (least + greatest) / 2;
    struct S { const char *name; int prio; hipStream_t s; };
    std::vector<S> st = {{"copy(greatest)", greatest, 0}, {"icp0(mid)", mid, 0}, {"grid(least)", least, 0}, {"build(least)", least, 0},
                         {"extra(greatest)", greatest, 0}, {"extra(mid)", mid, 0}, {"extra2(least)", least, 0}, {"flags-only", 99, 0}};
    for (auto &x : st) {
        if (x.prio == 99) CK(hipStreamCreateWithFlags(&x.s, hipStreamNonBlocking));
        else CK(hipStreamCreateWithPriority(&x.s, hipStreamNonBlocking, x.prio));
    }
    for (auto &x : st) { hipLaunchKernelGGL(nop, dim3(1), dim3(64), 0, x.s, nullptr); }
    CK(hipDeviceSynchronize());
    printf("%-18s", "A \\ B");
    for (auto &b : st) printf("%16s", b.name);
    printf("\n");
    for (auto &a : st) {
        printf("%-18s", a.name);
        for (auto &b : st) {
            if (a.s == b.s) { printf("%16s", "-"); continue; }
            hipLaunchKernelGGL(spin, dim3(1), dim3(64), 0, a.s, 30000ull); // 300 us
            const auto t0 = std::chrono::steady_clock::now();
            hipLaunchKernelGGL(nop, dim3(1), dim3(64), 0, b.s, nullptr);
            CK(hipStreamSynchronize(b.s));
            const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
            CK(hipDeviceSynchronize());
            printf("%13.0f us", us);
        }
        printf("\n");
    }
    return 0;
}
