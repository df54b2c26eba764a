// Why is the FIRST product of a panel slow next to a trailing update that has just started (T0: 100 us where its siblings take
// 27, N = 8192)?  The panel stream's launches carry 8 / R times the workgroups they need -- workgroup b runs on XCD (X + b) % 8
// and only those on the R reserved XCDs work -- and the idle ones still need a SLOT on their (busy) XCD before they can exit.
//   hog:   workgroups with the trailing update's footprint (240 VGPRs, 73 KB LDS: two per CU), exit at once on the reserved
//          XCDs, spin `hog_us` elsewhere; enough of them for `rounds` rounds                                  (main stream)
//   panel: G workgroups with the 32-row tile's footprint (176 VGPRs, 49 KB LDS), work 10 us on the reserved XCDs, exit at once
//          elsewhere                                                                                          (priority stream)
// Device-side stamps (s_memrealtime, 100 MHz): first hog start, first / last panel workgroup start, last panel end.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <cstring>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
__device__ __forceinline__ unsigned xcc_id()
{
    unsigned x;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(x));
    return x & 7u;
}
__device__ __forceinline__ void spin_us(double us)
{
    const long long t0 = __builtin_amdgcn_s_memrealtime();
    while ((double)(__builtin_amdgcn_s_memrealtime() - t0) < us * 100.0) __builtin_amdgcn_s_sleep(8);
}
struct Stamps {  // per XCD: the XCDs' s_memrealtime counters are NOT synchronised with each other (offsets of milliseconds)
    unsigned long long hog_first[8], hog_last_start[8], panel_first[8], panel_last_start[8], panel_end[8];
    unsigned seen[8];
};
__global__ __launch_bounds__(256) void hog(unsigned x0, unsigned nres, double us, Stamps* st)
{
    extern __shared__ double lds[];
    asm volatile("v_mov_b32 v239, 0" ::: "v239");
    if (((xcc_id() - x0) & 7u) < nres) return;
    if (threadIdx.x == 0) {
        const unsigned long long t = __builtin_amdgcn_s_memrealtime();
        atomicMin(&st->hog_first[xcc_id()], t);
        atomicMax(&st->hog_last_start[xcc_id()], t);
    }
    spin_us(us);
    if (us < 0) lds[threadIdx.x] = 1.0;
}
__global__ __launch_bounds__(256) void panel(unsigned x0, unsigned nres, double us, Stamps* st)
{
    extern __shared__ double lds[];
    asm volatile("v_mov_b32 v175, 0" ::: "v175");
    const unsigned x = xcc_id();
    const bool work = ((x - x0) & 7u) < nres;
    if (threadIdx.x == 0) {
        const unsigned long long t = __builtin_amdgcn_s_memrealtime();
        atomicMin(&st->panel_first[x], t);
        atomicMax(&st->panel_last_start[x], t);
        atomicAdd(st->seen + x, 1u);
    }
    if (work) spin_us(us);
    if (threadIdx.x == 0) {
        const unsigned long long t = __builtin_amdgcn_s_memrealtime();
        atomicMax(&st->panel_end[x], t);
    }
    if (us < 0) lds[threadIdx.x] = 1.0;
}
__global__ void where(unsigned* out)
{
    if (threadIdx.x == 0) out[blockIdx.x] = xcc_id();
}
int main()
{
    hipStream_t sm, sp;
    int lo, hi;
    CK(hipDeviceGetStreamPriorityRange(&lo, &hi));
    CK(hipStreamCreateWithFlags(&sm, hipStreamNonBlocking));
    CK(hipStreamCreateWithPriority(&sp, hipStreamNonBlocking, hi));
    CK(hipFuncSetAttribute((const void*)hog, hipFuncAttributeMaxDynamicSharedMemorySize, 73736));
    CK(hipFuncSetAttribute((const void*)panel, hipFuncAttributeMaxDynamicSharedMemorySize, 49160));
    unsigned* d;
    CK(hipMalloc(&d, 4096));
    Stamps* st;   // device memory (atomics on pinned host memory cross PCIe: a microsecond each)
    CK(hipMalloc(&st, sizeof(Stamps)));
    Stamps hs, *sth = &hs;
    hipLaunchKernelGGL(where, dim3(1), dim3(64), 0, sp, d);
    CK(hipStreamSynchronize(sp));
    unsigned x0 = 0;
    CK(hipMemcpy(&x0, d, 4, hipMemcpyDeviceToHost));
    printf("priority stream: single-workgroup launches run on XCD %u\n", x0);
    for (unsigned nres : {2u}) {
        const int need = 156, grid = need * 8 / (int)nres;
        for (int rounds : {1, 3}) {
            for (double hog_us : {110.0}) {
                for (int delay : {-1, 0, 5, 15, 25, 40, 80, 130}) {
                    memset(sth, 0, sizeof(Stamps));
                    for (int i = 0; i < 8; ++i) sth->hog_first[i] = sth->panel_first[i] = ~0ull;
                    CK(hipMemcpy(st, sth, sizeof(Stamps), hipMemcpyHostToDevice));
                    const int slots = (8 - (int)nres) * 32 * 2;
                    const int hog_grid = (slots * rounds - (rounds > 1 ? 100 : 0)) * 8 / (8 - (int)nres);
                    if (delay >= 0) hipLaunchKernelGGL(hog, dim3(hog_grid), dim3(256), 73736, sm, x0, nres, hog_us, st);
                    if (delay > 0) hipLaunchKernelGGL(panel, dim3(1), dim3(256), 49160, sp, x0, 8u, (double)delay, (Stamps*)(d + 512));
                    hipLaunchKernelGGL(panel, dim3(grid), dim3(256), 49160, sp, x0, nres, 10.0, st);
                    CK(hipStreamSynchronize(sp));
                    CK(hipStreamSynchronize(sm));
                    CK(hipMemcpy(sth, st, sizeof(Stamps), hipMemcpyDeviceToHost));
                    printf("R=%u hog %4d workgroups (%d round%s of %g us) panel asked to arrive %3d us in:", nres, hog_grid, rounds, rounds > 1 ? "s" : "", hog_us, delay);
                    // reserved XCD x0: the working workgroups (first start -> last end); a busy XCD: the idle ones against the hog's first start there
                    const unsigned xb = (x0 + nres) & 7u;
                    printf("  working XCD: %5.1f us first start -> last end |", ((double)sth->panel_end[x0] - (double)sth->panel_first[x0]) / 100.0);
                    if (delay >= 0)
                        printf(" busy XCD %u: hog starts 0 .. %+.1f, idle panel workgroups start %+7.1f .. %+7.1f us after the hog's first\n", xb,
                               ((double)sth->hog_last_start[xb] - (double)sth->hog_first[xb]) / 100.0, ((double)sth->panel_first[xb] - (double)sth->hog_first[xb]) / 100.0,
                               ((double)sth->panel_last_start[xb] - (double)sth->hog_first[xb]) / 100.0);
                    else
                        printf(" no hog: idle workgroups of a non-reserved XCD start over %.1f us\n", ((double)sth->panel_last_start[xb] - (double)sth->panel_first[xb]) / 100.0);
                }
            }
        }
    }
    return 0;
}
