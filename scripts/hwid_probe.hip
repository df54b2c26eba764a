// Which bits of HW_REG_HW_ID identify a CU?  512 co-resident blocks (2 per CU expected), per block (XCC_ID, HW_ID).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <map>
#include <set>
__global__ __launch_bounds__(256, 2) void k(unsigned* out)
{
    __shared__ double pad[9000];  // 72 KiB: two blocks per CU
    if (threadIdx.x == 0) {
        unsigned xcc, hw;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
        out[2 * blockIdx.x] = xcc;
        out[2 * blockIdx.x + 1] = hw;
        pad[0] = 1.0;
    }
    // stay resident long enough for every block to start
    unsigned long long t0 = wall_clock64();
    while (wall_clock64() - t0 < 200000) __builtin_amdgcn_s_sleep(10);
    if (pad[threadIdx.x] == 3.0) out[0] = 0;
}
int main()
{
    const int nb = 512;
    unsigned* d;
    (void)hipMalloc(&d, 2 * nb * sizeof(unsigned));
    hipLaunchKernelGGL(k, dim3(nb), dim3(256), 0, 0, d);
    static unsigned h[2 * nb];
    (void)hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    for (int i = 0; i < 8; ++i) printf("block %d: xcc %u hw_id 0x%08x\n", i, h[2 * i] & 0xf, h[2 * i + 1]);
    const unsigned masks[] = {0x0000ff00u, 0x0000ffc0u, 0x00001f00u, 0x0000e000u, 0x00000f00u, 0x0000ff30u, 0xffffffffu};
    for (unsigned m : masks) {
        std::map<unsigned long long, int> cnt;
        for (int i = 0; i < nb; ++i) cnt[((unsigned long long)(h[2 * i] & 0xf) << 32) | (h[2 * i + 1] & m)]++;
        int mx = 0;
        for (auto& kv : cnt) mx = kv.second > mx ? kv.second : mx;
        printf("mask 0x%08x: %zu distinct keys, max blocks per key %d\n", m, cnt.size(), mx);
    }
    unsigned orv = 0, andv = ~0u;
    for (int i = 0; i < nb; ++i) {
        orv |= h[2 * i + 1];
        andv &= h[2 * i + 1];
    }
    printf("bits that vary: 0x%08x\n", orv & ~andv);
    return 0;
}
