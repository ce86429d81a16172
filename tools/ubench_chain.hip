// Microbenchmark (diagnostic): cost of the exact IIR inner step on gfx950 for ONE active lane.
//   A: dependent pair   t = h*a ; acc = acc - t        (what the compiler emitted)
//   B: products first   t[k] = h[k]*a[k] (independent), then the dependent subtract chain
//   C: subtract chain only (dependent v_add_f64)
//   D: v_mov_b64 copies (history shift)
// Each kernel runs `iters` x 32 steps; time per step is reported for 1 wave per SIMD and 2 waves per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("hip error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

template <int MODE>
__global__ void k(double *out, const double *in, int iters)
{
    double a[32], h[32];
#pragma unroll
    for (int i = 0; i < 32; i++) { a[i] = in[i]; h[i] = in[32 + i]; }
    double acc = in[64];
    if ((threadIdx.x & 63) == 0) {
        for (int it = 0; it < iters; it++) {
            if (MODE == 0) {
#pragma unroll
                for (int k = 0; k < 32; k++) { double t = h[k] * a[k]; acc = acc - t; }
            } else if (MODE == 1) {
                double t[32];
#pragma unroll
                for (int k = 0; k < 32; k++) t[k] = h[k] * a[k];
#pragma unroll
                for (int k = 0; k < 32; k++) acc = acc - t[k];
            } else if (MODE == 2) {
#pragma unroll
                for (int k = 0; k < 32; k++) acc = acc - a[k];
            } else {
#pragma unroll
                for (int k = 31; k > 0; k--) h[k] = h[k - 1];
                h[0] = acc; acc = acc - h[31];
            }
            h[it & 31] = acc;   // keep the products loop-variant
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc + h[5];
}

template <int MODE>
void run(const char *name, int threads)
{
    double *in, *out;
    CHK(hipMalloc(&in, 80 * 8)); CHK(hipMalloc(&out, 1024 * 1024 * 8));
    double hin[80];
    for (int i = 0; i < 80; i++) hin[i] = 1.0 + 1e-9 * i;
    CHK(hipMemcpy(in, hin, sizeof hin, hipMemcpyHostToDevice));
    const int iters = 20000;
    hipEvent_t e0, e1; CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
    k<MODE><<<256, threads>>>(out, in, 100);
    CHK(hipDeviceSynchronize());
    CHK(hipEventRecord(e0));
    k<MODE><<<256, threads>>>(out, in, iters);
    CHK(hipEventRecord(e1)); CHK(hipDeviceSynchronize());
    float ms; CHK(hipEventElapsedTime(&ms, e0, e1));
    printf("%-28s threads/WG=%4d : %.2f ns per step (32 steps/iter)\n", name, threads, ms * 1e6 / (iters * 32.0));
    CHK(hipFree(in)); CHK(hipFree(out));
}

int main()
{
    for (int threads : {64, 256, 512}) {
        run<0>("A dependent mul->sub", threads);
        run<1>("B muls first, then sub chain", threads);
        run<2>("C sub chain only", threads);
        run<3>("D 31 v_mov_b64 + 1 sub", threads);
    }
    return 0;
}
