// Microbenchmark (diagnostic) #2: fp64 VALU issue/latency on gfx950, one wave per workgroup unless noted.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("hip error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

template <int MODE>
__global__ void k(double *out, const double *in, int iters)
{
    double a[16];
#pragma unroll
    for (int i = 0; i < 16; i++) a[i] = in[i];
    double c0 = in[64], c1 = in[65], c2 = in[66], c3 = in[67], c4 = in[68], c5 = in[69], c6 = in[70], c7 = in[71];
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int k = 0; k < 16; k++) {
            if (MODE == 0) { c0 = c0 - a[k]; }                                            // 1 dependent add chain
            if (MODE == 1) { c0 = c0 - a[k]; c1 = c1 - a[k]; }                            // 2 chains
            if (MODE == 2) { c0 = c0 - a[k]; c1 = c1 - a[k]; c2 = c2 - a[k]; c3 = c3 - a[k]; }   // 4 chains
            if (MODE == 3) { c0 -= a[k]; c1 -= a[k]; c2 -= a[k]; c3 -= a[k]; c4 -= a[k]; c5 -= a[k]; c6 -= a[k]; c7 -= a[k]; } // 8
            if (MODE == 4) { c0 = c0 * a[k]; }                                            // dependent mul chain
            if (MODE == 5) { c0 *= a[k]; c1 *= a[k]; c2 *= a[k]; c3 *= a[k]; }            // 4 mul chains
            if (MODE == 6) { c0 = __builtin_fma(c0, a[k], a[k]); }                        // dependent fma chain
            if (MODE == 7) { c0 = __builtin_fma(c0, a[k], a[k]); c1 = __builtin_fma(c1, a[k], a[k]); c2 = __builtin_fma(c2, a[k], a[k]); c3 = __builtin_fma(c3, a[k], a[k]); }
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = c0 + c1 + c2 + c3 + c4 + c5 + c6 + c7;
}

template <int MODE>
void run(const char *name, int threads, int per_iter_ops)
{
    double *in, *out;
    CHK(hipMalloc(&in, 80 * 8)); CHK(hipMalloc(&out, 1024 * 1024 * 8));
    double hin[80];
    for (int i = 0; i < 80; i++) hin[i] = 1.0 + 1e-9 * i;
    CHK(hipMemcpy(in, hin, sizeof hin, hipMemcpyHostToDevice));
    const int iters = 40000;
    hipEvent_t e0, e1; CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
    k<MODE><<<256, threads>>>(out, in, 100);
    CHK(hipDeviceSynchronize());
    CHK(hipEventRecord(e0));
    k<MODE><<<256, threads>>>(out, in, iters);
    CHK(hipEventRecord(e1)); CHK(hipDeviceSynchronize());
    float ms; CHK(hipEventElapsedTime(&ms, e0, e1));
    printf("%-26s threads/WG=%4d : %.2f ns per 16-step group /16 = %.3f ns per step; %.3f ns per VALU op per wave\n", name, threads,
           ms * 1e6 / iters, ms * 1e6 / iters / 16.0, ms * 1e6 / iters / per_iter_ops);
    CHK(hipFree(in)); CHK(hipFree(out));
}

int main()
{
    for (int threads : {64, 256}) {
        run<0>("add x1 chain", threads, 16);
        run<1>("add x2 chains", threads, 32);
        run<2>("add x4 chains", threads, 64);
        run<3>("add x8 chains", threads, 128);
        run<4>("mul x1 chain", threads, 16);
        run<5>("mul x4 chains", threads, 64);
        run<6>("fma x1 chain", threads, 16);
        run<7>("fma x4 chains", threads, 64);
    }
    return 0;
}
