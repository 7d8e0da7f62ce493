// Dependent-load latency as a latency-bound kernel of this library sees it (tools/exp, measurement only): one wavefront chases a
// pointer chain through a buffer of `bytes` (stride 192 B, a permutation), first right after launch (cold: TLB, L2 as the launch
// found them), then again (warm); wall clock by s_memrealtime (100 MHz).  Also: an LDS chase, and an exchange-like sc1 round trip.
//   hipcc --offload-arch=gfx950 -O3 -o latency_chase latency_chase.hip && ./latency_chase
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <numeric>
#include <algorithm>
#include <random>

__global__ void chase(const unsigned *buf, int steps, unsigned long long *out, int passes)
{
    unsigned p = threadIdx.x == 0 ? 0u : 0u;
    for (int k = 0; k < passes; ++k) {
        const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
        for (int i = 0; i < steps; ++i) p = buf[p];
        asm volatile("" ::"v"(p));
        const unsigned long long t1 = __builtin_amdgcn_s_memrealtime();
        if (threadIdx.x == 0) out[blockIdx.x * passes + k] = t1 - t0;
    }
    if (p == 0xffffffffu) out[0] = 0;
}

__global__ void lds_chase(int steps, unsigned long long *out)
{
    __shared__ unsigned l[4096];
    for (int i = threadIdx.x; i < 4096; i += blockDim.x) l[i] = (i * 97 + 13) & 4095;
    __syncthreads();
    unsigned                 p = threadIdx.x;
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    for (int i = 0; i < steps; ++i) p = l[p];
    asm volatile("" ::"v"(p));
    const unsigned long long t1 = __builtin_amdgcn_s_memrealtime();
    if (threadIdx.x == 0) out[0] = t1 - t0;
    if (p == 0xffffffffu) out[1] = 0;
}

int main()
{
    for (size_t bytes : {size_t(64) << 10, size_t(512) << 10, size_t(4) << 20, size_t(64) << 20}) {
        const size_t          n = bytes / 4, stride = 48; // 192 B
        const size_t          nodes = n / stride;
        std::vector<unsigned> order(nodes);
        std::iota(order.begin(), order.end(), 0u);
        std::mt19937 rng(1);
        std::shuffle(order.begin() + 1, order.end(), rng);
        std::vector<unsigned> h(n, 0);
        for (size_t i = 0; i < nodes; ++i) h[order[i] * stride] = order[(i + 1) % nodes] * stride;
        unsigned           *d;
        unsigned long long *o;
        hipMalloc(&d, bytes);
        hipMalloc(&o, 4096 * sizeof(unsigned long long));
        hipMemcpy(d, h.data(), bytes, hipMemcpyHostToDevice);
        const int steps = 64, passes = 3;
        for (int rep = 0; rep < 3; ++rep) {
            chase<<<1, 64>>>(d, steps, o, passes);
            hipDeviceSynchronize();
            unsigned long long r[3];
            hipMemcpy(r, o, sizeof(r), hipMemcpyDeviceToHost);
            printf("global chase %6zu KB rep %d: ns per load, pass 0 %.0f  pass 1 %.0f  pass 2 %.0f\n", bytes >> 10, rep, r[0] * 10.0 / steps, r[1] * 10.0 / steps,
                   r[2] * 10.0 / steps);
        }
        // the same from many workgroups at once (each its own start is the same chain: they share lines)
        chase<<<86, 64>>>(d, steps, o, passes);
        hipDeviceSynchronize();
        std::vector<unsigned long long> r(86 * 3);
        hipMemcpy(r.data(), o, r.size() * 8, hipMemcpyDeviceToHost);
        double a = 0, b = 0;
        for (int w = 0; w < 86; ++w) a += r[w * 3], b += r[w * 3 + 2];
        printf("   86 workgroups: pass 0 %.0f ns, pass 2 %.0f ns per load (mean)\n", a * 10.0 / steps / 86, b * 10.0 / steps / 86);
        hipFree(d);
        hipFree(o);
    }
    unsigned long long *o;
    hipMalloc(&o, 64);
    lds_chase<<<1, 64>>>(256, o);
    hipDeviceSynchronize();
    unsigned long long r;
    hipMemcpy(&r, o, 8, hipMemcpyDeviceToHost);
    printf("LDS chase: %.1f ns per dependent ds_read\n", r * 10.0 / 256);
    return 0;
}
