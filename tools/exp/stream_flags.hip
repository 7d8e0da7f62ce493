// does a stream made with hipExtStreamCreateWithCUMask synchronise with the legacy null stream (hipStreamDefault) or not
// (hipStreamNonBlocking, what every other stream of the library is)?   hipcc --offload-arch=gfx950 tools/exp/stream_flags.hip -o tools/exp/stream_flags
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <vector>
__global__ void spin(long long ticks, int *out)
{
    const long long t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) {}
    if (out) *out = 1;
}
int main()
{
    int n_cu = 0;
    hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, 0);
    std::vector<uint32_t> all((n_cu + 31) / 32, 0u);
    for (int i = 0; i < n_cu; ++i) all[i / 32] |= 1u << (i % 32);
    hipStream_t masked, plain;
    hipExtStreamCreateWithCUMask(&masked, (uint32_t)all.size(), all.data());
    hipStreamCreateWithFlags(&plain, hipStreamNonBlocking);
    unsigned fm = 99, fp = 99;
    hipStreamGetFlags(masked, &fm);
    hipStreamGetFlags(plain, &fp);
    printf("flags: masked %u, non-blocking stream %u (hipStreamDefault = %u, hipStreamNonBlocking = %u)\n", fm, fp, hipStreamDefault, hipStreamNonBlocking);
    // behaviour: a 20 ms kernel on the null stream, then a trivial kernel on each stream: does it wait?
    for (int which = 0; which < 2; ++which) {
        hipStream_t s = which ? plain : masked;
        hipDeviceSynchronize();
        hipLaunchKernelGGL(spin, dim3(1), dim3(1), 0, nullptr, 2000000LL, nullptr); // 100 MHz clock: 20 ms
        const auto t0 = std::chrono::steady_clock::now();
        hipLaunchKernelGGL(spin, dim3(1), dim3(1), 0, s, 100LL, nullptr);
        hipStreamSynchronize(s);
        const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
        printf("%s stream: a kernel enqueued behind a 20 ms null-stream kernel finished after %.2f ms (%s)\n", which ? "non-blocking" : "CU-masked", ms,
               ms > 10 ? "it WAITED for the null stream" : "it did not wait");
        hipDeviceSynchronize();
    }
    return 0;
}
