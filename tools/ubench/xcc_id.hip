#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(int* out) {
    // HW_REG_XCC_ID = 20, bits [3:0]
    const unsigned x = __builtin_amdgcn_s_getreg(20 | (0 << 6) | ((4 - 1) << 11));
    if (threadIdx.x == 0) out[blockIdx.x] = (int)x;
}
int main() {
    int* d; hipMalloc(&d, 64 * 4);
    k<<<64, 64>>>(d);
    int h[64]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    for (int i = 0; i < 64; i++) printf("%d ", h[i]);
    printf("\n");
    return 0;
}
