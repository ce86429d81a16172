// Diagnostic #4: same patterns as #3 but WITHOUT inline asm: plain C++ ops pinned with
// __builtin_amdgcn_sched_barrier(0) (does hipcc's hazard padding around asm statements cost an issue slot?)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("hip error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
#define SB() __builtin_amdgcn_sched_barrier(0)

template <int MODE>
__global__ void k(double *out, const double *in, int iters)
{
    double a[16], h[16], t0 = in[70], t1 = in[71], t2 = in[72];
#pragma unroll
    for (int i = 0; i < 16; i++) { a[i] = in[i]; h[i] = in[16 + i]; }
    double acc = in[64], vv = in[65];
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int k = 0; k < 16; k++) {
            if (MODE == 0) { acc = acc - a[k]; SB(); }
            if (MODE == 1) { acc = acc - t0; SB(); t0 = h[k] * vv; SB(); }                 // product used next step
            if (MODE == 2) {                                                                   // product used 3 steps later
                if (k % 3 == 0) { acc = acc - t0; SB(); t0 = h[k] * vv; SB(); }
                if (k % 3 == 1) { acc = acc - t1; SB(); t1 = h[k] * vv; SB(); }
                if (k % 3 == 2) { acc = acc - t2; SB(); t2 = h[k] * vv; SB(); }
            }
            if (MODE == 3) { acc = acc - a[k]; SB(); t0 = h[k] * vv + t0 * 0; SB(); }
        }
        vv += 1e-9;
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc + t0 + t1 + t2;
}

template <int MODE>
void run(const char *name, int threads)
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
    printf("%-44s threads/WG=%4d : %.3f ns per step\n", name, threads, ms * 1e6 / iters / 16.0);
    CHK(hipFree(in)); CHK(hipFree(out));
}

int main()
{
    for (int threads : {64}) {
        run<0>("0 sub chain (C++)", threads);
        run<1>("1 sub chain + mul feeding next step (C++)", threads);
        run<2>("2 sub chain + mul feeding 3 steps later", threads);
    }
    return 0;
}
