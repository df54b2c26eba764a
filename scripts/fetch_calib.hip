// Developer probe: calibrates rocprofv3's FETCH_SIZE / WRITE_SIZE on gfx950 against known byte counts for the access
// widths this library uses (8 B / lane coalesced loads and stores; 16 B / lane for comparison).  1 GiB buffers (past the
// 256 MiB Infinity Cache).  Run under:  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d out -- fetch_calib
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void read8(const double* __restrict__ p, size_t n, double* out)
{
    double s = 0.0;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) s += p[i];
    if (s == 12345.678) out[0] = s;
}
__global__ void read16(const double2* __restrict__ p, size_t n, double* out)
{
    double s = 0.0;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const double2 v = p[i];
        s += v.x + v.y;
    }
    if (s == 12345.678) out[0] = s;
}
__global__ void write8(double* __restrict__ p, size_t n)
{
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = 1.0;
}
int main()
{
    const size_t n = (size_t)1 << 27;  // 1 GiB of doubles
    double *a, *o;
    (void)hipMalloc(&a, n * 8);
    (void)hipMalloc(&o, 8);
    (void)hipMemset(a, 0, n * 8);
    for (int rep = 0; rep < 2; ++rep) {
        hipLaunchKernelGGL(read8, dim3(4096), dim3(256), 0, 0, a, n, o);
        hipLaunchKernelGGL(read16, dim3(4096), dim3(256), 0, 0, (const double2*)a, n / 2, o);
        hipLaunchKernelGGL(write8, dim3(4096), dim3(256), 0, 0, a, n);
    }
    (void)hipDeviceSynchronize();
    printf("bytes per kernel: %zu\n", n * 8);
    return 0;
}
