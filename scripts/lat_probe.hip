// Developer probe: dependent-issue latencies of the instructions on the pivot chain of the diagonal-block kernel (K4), one
// wave alone on a CU: cycles per instruction of a chain of N dependent operations (s_memtime around 256 of them).
//   hipcc --offload-arch=gfx950 -O3 -o lat_probe lat_probe.hip && ./lat_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

#define REP16(x) x x x x x x x x x x x x x x x x
#define REP256(x) REP16(REP16(x))

__global__ void probe(double* out, long long* t, double seed, int* ldsidx)
{
    __shared__ double lds[1024];
    const int lane = threadIdx.x & 63;
    lds[threadIdx.x] = seed + lane;
    __syncthreads();
    double x = seed, y = seed * 0.5 + 1.0;
    long long t0, t1;
    int k = 0;
    // 0: dependent v_fma_f64
    t0 = __builtin_amdgcn_s_memtime();
    REP256(asm volatile("v_fma_f64 %0, %0, %1, %1" : "+v"(x) : "v"(y));)
    t1 = __builtin_amdgcn_s_memtime();
    if (lane == 0) t[k] = t1 - t0; ++k;
    // 1: independent v_fma_f64 (4 accumulators)
    double a0 = x, a1 = y, a2 = x + 1, a3 = y + 1;
    t0 = __builtin_amdgcn_s_memtime();
    REP16(REP16(asm volatile("v_fma_f64 %0, %0, %4, %4\n\tv_fma_f64 %1, %1, %4, %4\n\tv_fma_f64 %2, %2, %4, %4\n\tv_fma_f64 %3, %3, %4, %4" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(y));))
    t1 = __builtin_amdgcn_s_memtime();
    if (lane == 0) t[k] = (t1 - t0) / 4; ++k;
    x += a0 + a1 + a2 + a3;
    // 2: dependent v_mul_f64
    t0 = __builtin_amdgcn_s_memtime();
    REP256(asm volatile("v_mul_f64 %0, %0, %1" : "+v"(x) : "v"(y));)
    t1 = __builtin_amdgcn_s_memtime();
    if (lane == 0) t[k] = t1 - t0; ++k;
    // 3: dependent v_rsq_f64
    x = 1.5;
    t0 = __builtin_amdgcn_s_memtime();
    REP256(asm volatile("v_rsq_f64 %0, %0" : "+v"(x));)
    t1 = __builtin_amdgcn_s_memtime();
    if (lane == 0) t[k] = t1 - t0; ++k;
    // 4: v_readlane_b32 x2 -> v_mul_f64 with the SGPR pair (the pivot publication), dependent
    t0 = __builtin_amdgcn_s_memtime();
    REP256(asm volatile("v_readlane_b32 s20, %0, 3\n\tv_readlane_b32 s21, %1, 3\n\ts_nop 1\n\tv_mul_f64 %2, s[20:21], %3"
                        : "+v"(reinterpret_cast<int*>(&x)[0]), "+v"(reinterpret_cast<int*>(&x)[1]), "=v"(x) : "v"(y) : "s20", "s21");)
    t1 = __builtin_amdgcn_s_memtime();
    if (lane == 0) t[k] = t1 - t0; ++k;
    // 5: dependent v_cndmask_b32 pair + v_mul_f64
    t0 = __builtin_amdgcn_s_memtime();
    REP256(asm volatile("v_cndmask_b32 %0, %0, %2, vcc\n\tv_cndmask_b32 %1, %1, %2, vcc\n\t" : "+v"(reinterpret_cast<int*>(&x)[0]), "+v"(reinterpret_cast<int*>(&x)[1]) : "v"(lane) : "vcc");)
    t1 = __builtin_amdgcn_s_memtime();
    if (lane == 0) t[k] = t1 - t0; ++k;
    // 6: permlane32_swap pair (dependent)
    {
        int lo = __double2loint(x), hi = __double2hiint(x), l2 = lo, h2 = hi;
        t0 = __builtin_amdgcn_s_memtime();
        REP256(asm volatile("v_permlane32_swap_b32 %0, %1\n\tv_permlane32_swap_b32 %2, %3" : "+v"(lo), "+v"(l2), "+v"(hi), "+v"(h2));)
        t1 = __builtin_amdgcn_s_memtime();
        if (lane == 0) t[k] = t1 - t0; ++k;
        x += lo + hi + l2 + h2;
    }
    // 7: LDS round trip: ds_write_b64 -> ds_read_b64 (same address, dependent through the data)
    {
        double* p = lds + lane;
        t0 = __builtin_amdgcn_s_memtime();
        REP256(asm volatile("ds_write_b64 %1, %0\n\tds_read_b64 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "+v"(x) : "v"((unsigned)(uintptr_t)p) : "memory");)
        t1 = __builtin_amdgcn_s_memtime();
        if (lane == 0) t[k] = t1 - t0; ++k;
    }
    // 8: LDS: ds_write_b64 + 8 x ds_read_b128 broadcast + wait (the column round trip of the pivot step)
    {
        double* p = lds + lane;
        double2 r0, r1, r2, r3, r4, r5, r6, r7;
        unsigned base = (unsigned)(uintptr_t)(lds + 64);
        t0 = __builtin_amdgcn_s_memtime();
        REP16(REP16(asm volatile("ds_write_b64 %9, %8\n\tds_read_b128 %0, %10\n\tds_read_b128 %1, %10 offset:16\n\tds_read_b128 %2, %10 offset:32\n\tds_read_b128 %3, %10 offset:48\n\t"
                                 "ds_read_b128 %4, %10 offset:64\n\tds_read_b128 %5, %10 offset:80\n\tds_read_b128 %6, %10 offset:96\n\tds_read_b128 %7, %10 offset:112\n\ts_waitcnt lgkmcnt(0)"
                                 : "=&v"(r0), "=&v"(r1), "=&v"(r2), "=&v"(r3), "=&v"(r4), "=&v"(r5), "=&v"(r6), "=&v"(r7), "+v"(x) : "v"((unsigned)(uintptr_t)p), "v"(base) : "memory"); x += r0.x;))
        t1 = __builtin_amdgcn_s_memtime();
        if (lane == 0) t[k] = t1 - t0; ++k;
        x += r1.x + r2.x + r3.x + r4.x + r5.x + r6.x + r7.x;
    }
    // 9: independent 32-bit VALU (v_mov) issue rate
    {
        int m0 = lane, m1 = lane + 1, m2 = lane + 2, m3 = lane + 3;
        t0 = __builtin_amdgcn_s_memtime();
        REP16(REP16(asm volatile("v_add_u32 %0, %0, %1\n\tv_add_u32 %1, %1, %2\n\tv_add_u32 %2, %2, %3\n\tv_add_u32 %3, %3, %0" : "+v"(m0), "+v"(m1), "+v"(m2), "+v"(m3));))
        t1 = __builtin_amdgcn_s_memtime();
        if (lane == 0) t[k] = (t1 - t0) / 4; ++k;
        x += m0 + m1 + m2 + m3;
    }
    // 10: dependent v_mfma_f64_16x16x4 (same accumulator)
    {
        typedef double d4 __attribute__((ext_vector_type(4)));
        d4 acc = {x, y, x, y};
        t0 = __builtin_amdgcn_s_memtime();
        REP16(REP16(acc = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, acc, 0, 0, 0);))
        t1 = __builtin_amdgcn_s_memtime();
        if (lane == 0) t[k] = t1 - t0; ++k;
        // 11: independent MFMAs (4 accumulators)
        d4 b0 = acc, b1 = acc, b2 = acc, b3 = acc;
        t0 = __builtin_amdgcn_s_memtime();
        REP16(REP16(b0 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, b0, 0, 0, 0); b1 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, b1, 0, 0, 0);
                    b2 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, b2, 0, 0, 0); b3 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, b3, 0, 0, 0);))
        t1 = __builtin_amdgcn_s_memtime();
        if (lane == 0) t[k] = (t1 - t0) / 4; ++k;
        x += acc[0] + b0[0] + b1[1] + b2[2] + b3[3];
    }
    out[threadIdx.x] = x;
}

int main()
{
    double* out; long long* t; int* idx;
    (void)hipMalloc(&out, 8 * 64); (void)hipMalloc(&t, 8 * 32); (void)hipMalloc(&idx, 4);
    const char* names[] = {"dependent v_fma_f64", "independent v_fma_f64 (issue)", "dependent v_mul_f64", "dependent v_rsq_f64",
                           "readlane x2 + s_nop + v_mul_f64 (SGPR operand), dependent", "dependent v_cndmask_b32 pair", "permlane32_swap pair (dependent)",
                           "LDS ds_write_b64 -> ds_read_b64 -> wait", "LDS ds_write_b64 + 8 ds_read_b128 + wait + add", "independent 32-bit VALU (issue)",
                           "dependent v_mfma_f64_16x16x4", "independent v_mfma_f64_16x16x4 (issue)"};
    for (int rep = 0; rep < 2; ++rep) {
        hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, out, t, 1.25, idx);
        (void)hipDeviceSynchronize();
    }
    long long h[32]; (void)hipMemcpy(h, t, 8 * 32, hipMemcpyDeviceToHost);
    for (int k = 0; k < 12; ++k) printf("%-62s %7.1f cycles per instruction (group)\n", names[k], h[k] / 256.0);
    return 0;
}
