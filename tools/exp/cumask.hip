// Which CUs does a stream made with hipExtStreamCreateWithCUMask run on (MI355X, SPX: 8 XCDs x 32 CUs)?
// Launches 4096 one-wavefront workgroups that each spin ~20 us and record (XCC_ID, SE_ID, SH_ID, CU_ID).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <set>
#include <vector>
#include <map>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
__global__ void where(unsigned *out)
{
    unsigned hw, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    while (__builtin_amdgcn_s_memrealtime() - t0 < 2000) __builtin_amdgcn_s_sleep(4); // 20 us at 100 MHz
    if (threadIdx.x == 0) out[blockIdx.x] = (hw & 0xffff) | ((xcc & 0xf) << 16);
}
int run(const char *name, hipStream_t st, unsigned *d, int n)
{
    hipLaunchKernelGGL(where, dim3(n), dim3(64), 0, st, d);
    CK(hipStreamSynchronize(st));
    std::vector<unsigned> h(n);
    CK(hipMemcpy(h.data(), d, 4 * n, hipMemcpyDeviceToHost));
    std::set<unsigned> cus;
    std::map<unsigned, std::set<unsigned>> per_xcc;
    for (unsigned v : h) {
        const unsigned cu = (v >> 8) & 0xf, sh = (v >> 12) & 1, se = (v >> 13) & 7, xcc = (v >> 16) & 0xf;
        const unsigned id = (xcc << 12) | (se << 8) | (sh << 4) | cu;
        cus.insert(id);
        per_xcc[xcc].insert(id & 0xfff);
    }
    printf("%-28s distinct CUs %3zu  per XCC:", name, cus.size());
    for (auto &kv : per_xcc) printf(" %u:%zu", kv.first, kv.second.size());
    printf("\n");
    return 0;
}
int main()
{
    unsigned *d;
    const int n = 8192;
    CK(hipMalloc(&d, 4 * n));
    hipStream_t s0;
    CK(hipStreamCreate(&s0));
    run("unmasked", s0, d, n);
    struct { const char *name; std::vector<uint32_t> mask; } cases[] = {
        {"first 128 bits", {0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu, 0, 0, 0, 0}},
        {"first 32 bits", {0xffffffffu, 0, 0, 0, 0, 0, 0, 0}},
        {"all but low 8 bits", {0xffffff00u, 0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu}},
        {"all but top 8 bits", {0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu, 0x00ffffffu}},
        {"every word minus bit 0", {0xfffffffeu, 0xfffffffeu, 0xfffffffeu, 0xfffffffeu, 0xfffffffeu, 0xfffffffeu, 0xfffffffeu, 0xfffffffeu}},
        {"one word only (32 bits)", {0xffffffffu}},
        {"even bits", {0x55555555u, 0x55555555u, 0x55555555u, 0x55555555u, 0x55555555u, 0x55555555u, 0x55555555u, 0x55555555u}},
    };
    for (auto &c : cases) {
        hipStream_t s;
        hipError_t  e = hipExtStreamCreateWithCUMask(&s, (uint32_t)c.mask.size(), c.mask.data());
        if (e != hipSuccess) { printf("%-28s create failed: %s\n", c.name, hipGetErrorString(e)); (void)hipGetLastError(); continue; }
        run(c.name, s, d, n);
        hipStreamDestroy(s);
    }
    return 0;
}
