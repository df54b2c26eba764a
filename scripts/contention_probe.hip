// Developer probe: what does a single latency-bound wave pay for sharing a CU with the library's FP64 GEMM?
// One workgroup (resources of the diagonal-block kernel: 8 waves, 80 KiB LDS) runs dependent chains of one instruction
// class on wave 0 and reports cycles per operation, alone and while a stream of fr_gemm launches fills the chip.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <unistd.h>
#include "friedrich_amd.h"

__global__ __launch_bounds__(512, 4) void probe(double* out, long long* ts, double seed)
{
    extern __shared__ double lds[];
    const int t = threadIdx.x, lane = t & 63, w = t >> 6;
    if (w != 0) return;
    if (seed > 0.2) __builtin_amdgcn_s_setprio(3);
    else if (seed > 0.1) __builtin_amdgcn_s_setprio(1);
    double x = seed + lane * 1e-9, y = 1.0 + seed;
    float f = (float)seed + lane;
    long long t0, t1;
    // 1) dependent f64 FMA chain
    t0 = __builtin_amdgcn_s_memtime();
#pragma unroll 1
    for (int i = 0; i < 64; ++i) {
#pragma unroll
        for (int j = 0; j < 16; ++j) x = __builtin_fma(x, y, 1e-9);
    }
    t1 = __builtin_amdgcn_s_memtime();
    if (lane == 0) ts[0] = t1 - t0;
    // 2) independent f64 FMAs (16 accumulators)
    double acc[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) acc[j] = x + j;
    t0 = __builtin_amdgcn_s_memtime();
#pragma unroll 1
    for (int i = 0; i < 64; ++i) {
#pragma unroll
        for (int j = 0; j < 16; ++j) acc[j] = __builtin_fma(acc[j], y, 1e-9);
    }
    t1 = __builtin_amdgcn_s_memtime();
    if (lane == 0) ts[1] = t1 - t0;
#pragma unroll
    for (int j = 0; j < 16; ++j) x += acc[j];
    // 3) dependent f32 FMA chain
    t0 = __builtin_amdgcn_s_memtime();
#pragma unroll 1
    for (int i = 0; i < 64; ++i) {
#pragma unroll
        for (int j = 0; j < 16; ++j) f = __builtin_fmaf(f, 1.0001f, 1e-6f);
    }
    t1 = __builtin_amdgcn_s_memtime();
    if (lane == 0) ts[2] = t1 - t0;
    // 4) LDS round trips: write, read back (dependent)
    double v = x;
    t0 = __builtin_amdgcn_s_memtime();
#pragma unroll 1
    for (int i = 0; i < 256; ++i) {
        lds[lane + 64 * (i & 7)] = v;
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        v = lds[((lane + 1) & 63) + 64 * (i & 7)] + 1.0;
    }
    t1 = __builtin_amdgcn_s_memtime();
    if (lane == 0) ts[3] = t1 - t0;
    // 5) readlane chain (VALU -> SGPR -> VALU)
    int q = lane;
    t0 = __builtin_amdgcn_s_memtime();
#pragma unroll 1
    for (int i = 0; i < 64; ++i) {
#pragma unroll
        for (int j = 0; j < 16; ++j) q = q + __builtin_amdgcn_readlane(q, 5);
    }
    t1 = __builtin_amdgcn_s_memtime();
    if (lane == 0) ts[4] = t1 - t0;
    // 6) dependent v_rsq_f64 chain
    double z = 1.5 + seed;
    t0 = __builtin_amdgcn_s_memtime();
#pragma unroll 1
    for (int i = 0; i < 64; ++i) {
#pragma unroll
        for (int j = 0; j < 4; ++j) z = __builtin_amdgcn_rsq(z) + 1.0;
    }
    t1 = __builtin_amdgcn_s_memtime();
    if (lane == 0) ts[5] = t1 - t0;
    // 7) dependent f64 FMA chain with an independent v_rsq_f64 between the links
    double dz = 2.5 + seed;
    t0 = __builtin_amdgcn_s_memtime();
#pragma unroll 1
    for (int i = 0; i < 64; ++i) {
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            x = __builtin_fma(x, y, 1e-9);
            asm volatile("v_rsq_f64 %0, %0" : "+v"(dz));
        }
    }
    t1 = __builtin_amdgcn_s_memtime();
    if (lane == 0) ts[6] = t1 - t0;
    // 8) dependent f64 FMA chain, two interleaved independent chains
    double x2 = x + 1.0;
    t0 = __builtin_amdgcn_s_memtime();
#pragma unroll 1
    for (int i = 0; i < 64; ++i) {
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            x = __builtin_fma(x, y, 1e-9);
            x2 = __builtin_fma(x2, y, 1e-9);
        }
    }
    t1 = __builtin_amdgcn_s_memtime();
    if (lane == 0) ts[7] = t1 - t0;
    out[lane] = x + f + v + q + z + dz + x2;
}

int main(int argc, char** argv)
{
    const bool noise = argc > 1;
    fr_ctx* ctx = nullptr;
    double *NA = nullptr, *NC = nullptr;
    const int64_t NM = 16384, NK = 512;
    if (noise) {
        if (fr_ctx_create(&ctx, 0) != FR_OK) return 1;
        (void)hipMalloc(&NA, NM * NK * 8);
        (void)hipMalloc(&NC, NM * NM * 8);
        (void)hipMemset(NA, 0, NM * NK * 8);
        (void)hipMemset(NC, 0, NM * NM * 8);
    }
    hipStream_t hs;
    int lo_p, hi_p;
    (void)hipDeviceGetStreamPriorityRange(&lo_p, &hi_p);
    (void)hipStreamCreateWithPriority(&hs, hipStreamNonBlocking, hi_p);
    double* out;
    long long* ts;
    (void)hipMalloc(&out, 8 * 64);
    (void)hipMalloc(&ts, 8 * 16);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(probe), hipFuncAttributeMaxDynamicSharedMemorySize, 81936);
    for (int rep = 0; rep < 4; ++rep) {
        if (noise) {
            for (int g = 0; g < 6; ++g) fr_gemm(ctx, 0, 1, NM, NM, NK, -1.0, NA, NM, NA, NM, 1.0, NC, NM);
            usleep(6000);
        }
        hipLaunchKernelGGL(probe, dim3(1), dim3(512), 81936, hs, out, ts, (rep == 0) ? 0.25 : ((rep == 1) ? 0.125 : 0.0625));
        (void)hipDeviceSynchronize();
        long long h[16];
        (void)hipMemcpy(h, ts, 128, hipMemcpyDeviceToHost);
        printf("%s rep %d: cycles/op  dep-fma64 %.1f | indep-fma64 %.1f | dep-fma32 %.1f | lds round trip %.1f | readlane+add %.1f | dep-rsq64+add %.1f | dep-fma64+indep-rsq %.1f | 2 dep-fma64 chains (per pair) %.1f\n",
               noise ? "noise" : "alone", rep, h[0] / 1024.0, h[1] / 1024.0, h[2] / 1024.0, h[3] / 256.0, h[4] / 1024.0, h[5] / 256.0, h[6] / 1024.0, h[7] / 1024.0);
    }
    return 0;
}
