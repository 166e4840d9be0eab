#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(int* out) {
    unsigned x;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(x));
    if (threadIdx.x == 0) out[blockIdx.x] = (int)(x & 0xf);
}
int main() {
    int n = 128; int* d; hipMalloc(&d, n * 4); int h[128];
    for (int rep = 0; rep < 3; ++rep) {
        hipLaunchKernelGGL(k, dim3(n), dim3(64), 0, 0, d);
        hipMemcpy(h, d, n * 4, hipMemcpyDeviceToHost);
        for (int i = 0; i < 32; ++i) printf("%d ", h[i]); printf("| workers(b%%8==0): ");
        for (int i = 0; i < n; i += 8) printf("%d ", h[i]); printf("\n");
    }
    return 0;
}
