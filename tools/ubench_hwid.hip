// Diagnostic #6: which SIMD does each wavefront of a 512-thread workgroup land on?  (HW_REG_HW_ID: wave_id[3:0], simd_id[5:4], cu_id[11:8])
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(unsigned *out)
{
    unsigned id;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(id));
    if ((threadIdx.x & 63) == 0) out[blockIdx.x * 8 + (threadIdx.x >> 6)] = id;
}
int main()
{
    unsigned *d, h[8 * 16];
    hipMalloc(&d, sizeof h);
    k<<<16, 512>>>(d);
    hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
    for (int b = 0; b < 16; b++) {
        printf("wg %2d (cu %2u): simd of waves 0..7 =", b, (h[b * 8] >> 8) & 15);
        for (int w = 0; w < 8; w++) printf(" %u", (h[b * 8 + w] >> 4) & 3);
        printf("   wave slots =");
        for (int w = 0; w < 8; w++) printf(" %u", h[b * 8 + w] & 15);
        printf("\n");
    }
    return 0;
}
