// Where does the single workgroup of a one-block launch run?  (a) a sequence of one-block launches on one stream,
// (b) the same while a second stream keeps the chip full, (c) 3-block and 8-block launches in between.
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void one(int* out, int slot)
{
    if (threadIdx.x == 0) {
        int id;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(id));
        out[slot + blockIdx.x] = id & 0xf;
    }
}
__global__ void busy(double* x, int iters)
{
    double a = x[threadIdx.x];
    for (int i = 0; i < iters; ++i) a = a * 1.0000001 + 1e-9;
    x[blockIdx.x * 256 + threadIdx.x] = a;
}
int main()
{
    int* d;
    (void)hipMalloc(&d, 4096 * sizeof(int));
    double* x;
    (void)hipMalloc(&x, 8192 * 256 * sizeof(double));
    hipStream_t s1, s2;
    (void)hipStreamCreateWithFlags(&s1, hipStreamNonBlocking);
    (void)hipStreamCreateWithFlags(&s2, hipStreamNonBlocking);
    static int h[4096];
    for (int phase = 0; phase < 3; ++phase) {
        if (phase >= 1) hipLaunchKernelGGL(busy, dim3(8192), dim3(256), 0, s2, x, 200000);
        int slot = 0;
        for (int i = 0; i < 24; ++i) {
            const int nb = (phase == 2) ? ((i % 3 == 0) ? 1 : (i % 3 == 1 ? 3 : 8)) : 1;
            hipLaunchKernelGGL(one, dim3(nb), dim3(256), 0, s1, d, slot);
            slot += nb;
        }
        (void)hipStreamSynchronize(s1);
        (void)hipMemcpy(h, d, sizeof(int) * slot, hipMemcpyDeviceToHost);
        printf("phase %d (%s): ", phase, phase == 0 ? "idle, 1-block launches" : phase == 1 ? "busy chip, 1-block launches" : "busy chip, 1/3/8-block launches");
        for (int i = 0; i < slot; ++i) printf("%d ", h[i]);
        printf("\n");
        (void)hipDeviceSynchronize();
    }
    return 0;
}
