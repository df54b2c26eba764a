// How do hipExtStreamCreateWithCUMask bits map onto (XCD, SE, CU) on MI355X?  Prints the set of physical CUs that ran
// blocks for a few masks.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <set>
#include <vector>
__global__ void k(unsigned* out)
{
    if (threadIdx.x == 0) {
        unsigned xcc, hwid;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
        out[blockIdx.x] = ((xcc & 0xf) << 16) | ((hwid >> 8) & 0xff);  // cu_id[11:8] sh_id[12] se_id[15:13]
    }
    // keep the CU busy a little so blocks spread over everything that is allowed
    long long t0 = __builtin_amdgcn_s_memtime();
    while (__builtin_amdgcn_s_memtime() - t0 < 20000) {}
}
static void run(const char* name, std::vector<uint32_t> mask)
{
    hipStream_t s;
    hipError_t e = hipExtStreamCreateWithCUMask(&s, (uint32_t)mask.size(), mask.data());
    if (e != hipSuccess) { printf("%s: create failed: %s\n", name, hipGetErrorString(e)); return; }
    const int nb = 2048;
    unsigned* d;
    (void)hipMalloc(&d, nb * 4);
    hipLaunchKernelGGL(k, dim3(nb), dim3(256), 0, s, d);
    (void)hipStreamSynchronize(s);
    std::vector<unsigned> h(nb);
    (void)hipMemcpy(h.data(), d, nb * 4, hipMemcpyDeviceToHost);
    std::set<unsigned> cus;
    for (auto v : h) cus.insert(v);
    printf("%s: %zu distinct (xcc,se,sh,cu):", name, cus.size());
    int cnt = 0;
    for (auto v : cus) if (cnt++ < 12) printf(" x%u:%03x", v >> 16, v & 0xffff);
    printf("\n");
    (void)hipFree(d);
    (void)hipStreamDestroy(s);
}
int main()
{
    run("all-256", std::vector<uint32_t>(8, 0xffffffffu));
    run("bit0", {1u, 0, 0, 0, 0, 0, 0, 0});
    run("bits0-7", {0xffu, 0, 0, 0, 0, 0, 0, 0});
    run("word0", {0xffffffffu, 0, 0, 0, 0, 0, 0, 0});
    run("all-but-bit0", {0xfffffffeu, 0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu});
    run("one-word-only(32bits)", {0xffffffffu});
    return 0;
}
