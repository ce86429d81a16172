// Diagnostic #3: what does an independent fp64 op cost next to a dependent fp64 chain (one wave)?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("hip error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
#define SUB(ACC, T) asm volatile("v_add_f64 %0, %0, -%1" : "+v"(ACC) : "v"(T))
#define MUL(D, A, B) asm volatile("v_mul_f64 %0, %1, %2" : "=v"(D) : "v"(A), "v"(B))
#define ADD(D, A, B) asm volatile("v_add_f64 %0, %1, %2" : "=v"(D) : "v"(A), "v"(B))
#define FMA(D, A, B) asm volatile("v_fma_f64 %0, %1, %2, %2" : "=v"(D) : "v"(A), "v"(B))
#define MULF(D, A, B) asm volatile("v_mul_f32 %0, %1, %2" : "=v"(D) : "v"(A), "v"(B))
#define NOP() asm volatile("s_nop 0")

template <int MODE>
__global__ void k(double *out, const double *in, int iters)
{
    double a[16], h[16], t0 = in[70], t1 = in[71], t2 = in[72];
    float f0 = (float)in[73], f1 = (float)in[74];
#pragma unroll
    for (int i = 0; i < 16; i++) { a[i] = in[i]; h[i] = in[16 + i]; }
    double acc = in[64];
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int k = 0; k < 16; k++) {
            if (MODE == 0) { SUB(acc, a[k]); }
            if (MODE == 1) { SUB(acc, t0); MUL(t0, h[k], a[k]); }                       // consumer uses product of previous step
            if (MODE == 2) { SUB(acc, (k % 3 == 0 ? t0 : k % 3 == 1 ? t1 : t2)); if (k % 3 == 0) MUL(t0, h[k], a[k]); else if (k % 3 == 1) MUL(t1, h[k], a[k]); else MUL(t2, h[k], a[k]); }
            if (MODE == 3) { SUB(acc, a[k]); ADD(t0, h[k], a[k]); }                     // independent add beside the chain
            if (MODE == 4) { SUB(acc, a[k]); MUL(t0, h[k], a[k]); }                     // independent mul, result unused by chain
            if (MODE == 5) { SUB(acc, a[k]); MULF(f0, f1, f1); }                        // independent f32 op
            if (MODE == 6) { SUB(acc, a[k]); MUL(t0, h[k], a[k]); MUL(t1, h[k], a[k]); }
            if (MODE == 7) { NOP(); SUB(acc, t0); MUL(t0, h[k], a[k]); }
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc + t0 + t1 + t2 + f0;
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
    for (int threads : {64, 256, 512}) {
        run<0>("0 sub chain", threads);
        run<1>("1 sub chain + mul feeding next step", threads);
        run<2>("2 sub chain + mul feeding 3 steps later", threads);
        run<3>("3 sub chain + independent add", threads);
        run<4>("4 sub chain + independent mul", threads);
        run<5>("5 sub chain + independent f32 mul", threads);
        run<6>("6 sub chain + 2 independent mul", threads);
        run<7>("7 nop + sub chain + mul feeding next", threads);
    }
    return 0;
}
