// Which (SE, SH, CU) ids exist on each XCD of this MI355X, and how does the dispatcher deal the workgroups of one launch to
// them (HW_REG_HW_ID: cu_id[11:8] sh_id[12] se_id[15:13])?  Prints, per XCD, the CU ids seen per SE and how many of a launch's
// workgroups ran on each SE for a launch that fits in one round and one that needs several.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <map>
#include <set>
#include <vector>
__global__ void k(unsigned* out, int spin)
{
    if (threadIdx.x == 0) {
        unsigned xcc, hwid;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
        out[blockIdx.x] = ((xcc & 0xf) << 16) | (hwid & 0xffff);
    }
    long long t0 = __builtin_amdgcn_s_memtime();
    while (__builtin_amdgcn_s_memtime() - t0 < spin) {}
}
int main()
{
    for (int nb : {256, 2048}) {
        unsigned* d;
        (void)hipMalloc(&d, nb * 4);
        hipLaunchKernelGGL(k, dim3(nb), dim3(256), 0, 0, d, 40000);
        (void)hipDeviceSynchronize();
        std::vector<unsigned> h(nb);
        (void)hipMemcpy(h.data(), d, nb * 4, hipMemcpyDeviceToHost);
        std::map<unsigned, std::map<unsigned, std::set<unsigned>>> cus;  // xcc -> se -> {sh:cu}
        std::map<unsigned, std::map<unsigned, int>> cnt;
        for (int b = 0; b < nb; ++b) {
            const unsigned v = h[b], xcc = v >> 16, cu = (v >> 8) & 0xf, sh = (v >> 12) & 1, se = (v >> 13) & 7;
            cus[xcc][se].insert(sh * 16 + cu);
            cnt[xcc][se]++;
        }
        printf("launch of %d workgroups:\n", nb);
        for (auto& x : cus) {
            printf("  XCD %u:", x.first);
            for (auto& s : x.second) {
                printf("  SE%u[%d wgs]:", s.first, cnt[x.first][s.first]);
                for (auto c : s.second) printf(" %s%u", c >= 16 ? "h" : "", c & 15);
            }
            printf("\n");
        }
        if (nb == 256) {
            printf("  first 32 workgroups (b: xcd/se/cu):");
            for (int b = 0; b < 32; ++b) printf(" %d:%u/%u/%u", b, h[b] >> 16, (h[b] >> 13) & 7, (h[b] >> 8) & 0xf);
            printf("\n");
        }
        (void)hipFree(d);
    }
    return 0;
}
