// How the dispatcher places single-wavefront workgroups: W wavefronts of issue-bound fp64 work, launched as W
// workgroups of one wavefront and as W / 4 workgroups of four -- plain, with 4 KB of LDS and a barrier, with a 136-register footprint, with both.
//   * On an idle chip the two forms take the same time at every W and every duration: single-wavefront workgroups ARE spread one per SIMD.
//   * Launched right behind a kernel of 1024 workgroups x 512 threads x 72 KB of LDS (the shape of the register-light pitch kernel, which is
//     what precedes the vocoder pipeline's autocorrelation when both processes run), the one-wavefront form takes 41.5 us where the
//     four-wavefront form takes 25.5 us (W = 896 on 1024 SIMDs): some SIMDs get two wavefronts, others none.  A workgroup's own wavefronts
//     are always dealt to the CU's four SIMDs in turn, so the four-wavefront form is immune.
//   hipcc -O3 --offload-arch=gfx950 -w tools/ubench_placement.hip -o /tmp/ubench_placement && /tmp/ubench_placement
#include <hip/hip_runtime.h>
#include <cstdio>
template <int LDSUSE, int BIGREGS>
__global__ void k(double *out, int iters)
{
    extern __shared__ double sm[];
    if (LDSUSE) { for (int i = threadIdx.x; i < 528; i += blockDim.x) sm[i] = i; __syncthreads(); }
    if (BIGREGS) asm volatile("v_mov_b32 v135, 0" ::: "v135");          // a register footprint like the autocorrelation kernel's (136 VGPRs)
    double a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    const double c = 1.0000001, d = 1e-9;
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int r = 0; r < 32; r++)
            asm volatile("v_fma_f64 %0, %0, %8, %9\n v_fma_f64 %1, %1, %8, %9\n v_fma_f64 %2, %2, %8, %9\n v_fma_f64 %3, %3, %8, %9\n"
                         "v_fma_f64 %4, %4, %8, %9\n v_fma_f64 %5, %5, %8, %9\n v_fma_f64 %6, %6, %8, %9\n v_fma_f64 %7, %7, %8, %9"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c), "v"(d));
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + (LDSUSE ? sm[threadIdx.x & 63] : 0.0);
}
// a predecessor like the register-light pitch kernel: 1024 workgroups of 512 threads, 72 KB of LDS each (two per CU), short
__global__ __launch_bounds__(512) void pred(double *out, int iters)
{
    extern __shared__ double sm[];
    for (int i = threadIdx.x; i < 9216; i += 512) sm[i] = i;
    __syncthreads();
    double a = sm[threadIdx.x];
    for (int it = 0; it < iters; it++) a = a * 1.0000001 + sm[(threadIdx.x + it) & 8191];
    out[blockIdx.x * 512 + threadIdx.x] = a;
}
static int ITERS = 400, PRED = 0;
template <int LDSUSE, int BIGREGS>
static float run(int waves, int wavesPerWg, double *d)
{
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const size_t lds = LDSUSE ? 528 * 8 : 0;
    k<LDSUSE, BIGREGS><<<waves / wavesPerWg, 64 * wavesPerWg, lds>>>(d, 10);
    if (PRED) pred<<<1024, 512, 9216 * 8>>>(d + 1024 * 1024, 200);
    hipEventRecord(e0);
    k<LDSUSE, BIGREGS><<<waves / wavesPerWg, 64 * wavesPerWg, lds>>>(d, ITERS);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    return ms * 1e3f;
}
int main()
{
    double *d; hipMalloc(&d, (4096 * 64 + 1024 * 1024 + 1024 * 512) * 8);
    const int ws[] = {256, 512, 768, 896, 1024, 1536, 2048};
    hipFuncSetAttribute((const void *)pred, hipFuncAttributeMaxDynamicSharedMemorySize, 9216 * 8);
    for (int iters : {400, 40, 10, -40}) {
    PRED = iters < 0; ITERS = iters < 0 ? -iters : iters;
    printf("-- %d iterations of 256 fp64 multiply-adds per wavefront%s\n", ITERS, PRED ? ", launched right behind 1024 workgroups of 512 threads with 72 KB of LDS each (the events bracket the second kernel only)" : "");
    printf("%8s %22s %22s   (us; plain | with 4 KB of LDS and a barrier | with a 136-register footprint | both)\n", "waves", "1 wavefront / workgroup", "4 wavefronts / workgroup");
    for (int w : ws)
        printf("%8d   %6.1f %6.1f %6.1f %6.1f      %6.1f %6.1f %6.1f %6.1f\n", w, run<0, 0>(w, 1, d), run<1, 0>(w, 1, d), run<0, 1>(w, 1, d), run<1, 1>(w, 1, d),
               run<0, 0>(w, 4, d), run<1, 0>(w, 4, d), run<0, 1>(w, 4, d), run<1, 1>(w, 4, d));
    }
    return 0;
}
