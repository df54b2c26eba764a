// Which XCD does block b run on?  (dispatch rule check: expected b % 8)
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(int* out)
{
    if (threadIdx.x == 0) {
        int id;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(id));
        out[blockIdx.x] = id & 0xf;
    }
}
int main()
{
    const int nb = 4096;
    int* d;
    (void)hipMalloc(&d, nb * sizeof(int));
    hipLaunchKernelGGL(k, dim3(nb), dim3(256), 0, 0, d);
    static int h[nb];
    (void)hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    int bad = 0;
    for (int i = 0; i < nb; ++i) bad += (h[i] != (i % 8));
    printf("first 32: ");
    for (int i = 0; i < 32; ++i) printf("%d ", h[i]);
    printf("\nblocks not on XCD b%%8: %d of %d\n", bad, nb);
    return 0;
}
