// Micro-benchmark: sustained v_mfma_f64_16x16x4_f64 rate on gfx950 (the FP64 matrix-core ceiling the
// Cholesky trailing update is priced against).  Reports wall-clock TFLOP/s, shader cycles per MFMA
// (s_memtime) and the effective clock, at several occupancies, to separate the issue rate from DVFS.
// Build: hipcc --offload-arch=gfx950 -O3 -o mfma_f64_peak mfma_f64_peak.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4 __attribute__((ext_vector_type(4)));
template <int NACC>
__global__ __launch_bounds__(256) void k(double* out, long long* cyc, int iters)
{
    d4 acc[NACC];
    for (int i = 0; i < NACC; ++i) acc[i] = d4{0, 0, 0, 0};
    double a = threadIdx.x * 1e-3, b = threadIdx.x * 2e-3 + 1.0;
    const long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) asm volatile("v_mfma_f64_16x16x4_f64 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(a), "v"(b));
    }
    double s = 0;
    for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    const long long t1 = __builtin_amdgcn_s_memtime();
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
template <int NACC>
void run(int nblocks, int threads)
{
    double* out;
    long long* cyc;
    (void)hipMalloc(&out, sizeof(double) * nblocks * threads);
    (void)hipMalloc(&cyc, sizeof(long long) * nblocks);
    const int iters = 20000;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(k<NACC>, dim3(nblocks), dim3(threads), 0, 0, out, cyc, 100);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(k<NACC>, dim3(nblocks), dim3(threads), 0, 0, out, cyc, iters);
    (void)hipEventRecord(e1);
    (void)hipDeviceSynchronize();
    float ms;
    (void)hipEventElapsedTime(&ms, e0, e1);
    long long c0 = 0;
    (void)hipMemcpy(&c0, cyc, sizeof(c0), hipMemcpyDeviceToHost);
    const double nm = (double)NACC * iters;
    const double flops = 2.0 * 16 * 16 * 4 * nm * (threads / 64) * nblocks;
    printf("NACC=%2d blocks=%4d waves/block=%d: %7.2f TFLOP/s  %.3f ms  memtime ticks/MFMA=%.1f  ticks/s=%.3e\n", NACC, nblocks,
           threads / 64, flops / ms / 1e9, ms, (double)c0 / nm, (double)c0 / (ms * 1e-3));
    (void)hipFree(out);
    (void)hipFree(cyc);
}
int main()
{
    run<16>(1, 64);
    run<16>(256, 64);
    run<16>(256, 128);
    run<16>(256, 256);
    run<16>(512, 256);
    run<4>(256, 256);
    run<1>(256, 256);
    return 0;
}
