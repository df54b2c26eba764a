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
// the store pattern of a column-major output tile (Gram assembly, GEMM epilogue): a wave writes 64 consecutive rows
// (512 B) of one column, then the same rows of the next column (ld * 8 bytes further)
__global__ void write_tile8(double* __restrict__ p, size_t n, size_t ld)
{
    const int t = threadIdx.x, r = t & 63, g = t >> 6;
    const size_t tiles_m = n / 128;
    const size_t i0 = (blockIdx.x % tiles_m) * 128, j0 = (blockIdx.x / tiles_m) * 64;
    for (int h = 0; h < 2; ++h)
        for (int b = 0; b < 16; ++b) p[(i0 + r + 64 * h) + (j0 + g * 16 + b) * ld] = 1.0;
}
// same tile, a wave writes 128 consecutive rows (1 KiB) of one column per instruction (16 B per lane)
__global__ void write_tile16(double* __restrict__ p, size_t n, size_t ld)
{
    const int t = threadIdx.x, r = t & 63, g = t >> 6;
    const size_t tiles_m = n / 128;
    const size_t i0 = (blockIdx.x % tiles_m) * 128, j0 = (blockIdx.x / tiles_m) * 64;
    for (int b = 0; b < 16; ++b) {
        double2 v;
        v.x = 1.0;
        v.y = 2.0;
        *reinterpret_cast<double2*>(p + (i0 + 2 * r) + (j0 + g * 16 + b) * ld) = v;
    }
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
    {
        const size_t m = 8192;  // 8192 x 8192 doubles = 512 MiB, as (8192/128) * (8192/64) tiles
        for (int rep = 0; rep < 2; ++rep) {
            hipLaunchKernelGGL(write_tile8, dim3((unsigned)((m / 128) * (m / 64))), dim3(256), 0, 0, a, m, m);
            hipLaunchKernelGGL(write_tile16, dim3((unsigned)((m / 128) * (m / 64))), dim3(256), 0, 0, a, m, m);
            hipLaunchKernelGGL(write_tile8, dim3((unsigned)((m / 128) * (m / 64))), dim3(256), 0, 0, a, m, m + 64);
        }
    }
    (void)hipDeviceSynchronize();
    printf("bytes per kernel: %zu (streams), %zu (tiles)\n", n * 8, (size_t)8192 * 8192 * 8);
    return 0;
}
