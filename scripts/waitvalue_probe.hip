// Developer probe (round 6): can a stream wait for a value a RUNNING kernel of another stream writes (hipStreamWaitValue32 on signal
// memory), and what does the hand-over cost?  Kernel A (stream 1) sleeps ~100 us, writes the value, sleeps another ~100 us; stream 2
// waits for the value and runs kernel B, which stamps the 100 MHz clock.  Reported: B's stamp minus A's write stamp.
// Build: hipcc --offload-arch=gfx950 -O2 -o waitvalue_probe waitvalue_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void ka(unsigned* sig, unsigned v, unsigned long long* ts)
{
    const unsigned long long t0 = wall_clock64();
    while (wall_clock64() - t0 < 10000) __builtin_amdgcn_s_sleep(8);
    __hip_atomic_store(sig, v, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    ts[0] = wall_clock64();
    while (wall_clock64() - t0 < 20000) __builtin_amdgcn_s_sleep(8);
    ts[1] = wall_clock64();
}
__global__ void kb(unsigned long long* ts) { ts[2] = wall_clock64(); }
int main()
{
    unsigned* sig = nullptr;
    hipError_t e = hipExtMallocWithFlags((void**)&sig, 8, hipMallocSignalMemory);
    printf("hipExtMallocWithFlags(signal): %s\n", hipGetErrorString(e));
    if (e != hipSuccess) return 1;
    (void)hipMemset(sig, 0, 8);
    unsigned long long* ts;
    (void)hipHostMalloc((void**)&ts, 64, hipHostMallocMapped);
    hipStream_t s1, s2;
    (void)hipStreamCreate(&s1);
    (void)hipStreamCreate(&s2);
    for (unsigned it = 1; it <= 5; ++it) {
        ts[0] = ts[1] = ts[2] = 0;
        hipLaunchKernelGGL(ka, dim3(1), dim3(64), 0, s1, sig, it, ts);
        e = hipStreamWaitValue32(s2, sig, it, hipStreamWaitValueGte, 0xffffffffu);
        if (e != hipSuccess) { printf("hipStreamWaitValue32: %s\n", hipGetErrorString(e)); return 1; }
        hipLaunchKernelGGL(kb, dim3(1), dim3(64), 0, s2, ts);
        (void)hipDeviceSynchronize();
        printf("it %u: value written -> waiting stream's kernel started: %.1f us (kernel A ended %.1f us after the write)\n", it,
               ((long long)ts[2] - (long long)ts[0]) / 100.0, ((long long)ts[1] - (long long)ts[0]) / 100.0);
    }
    return 0;
}
