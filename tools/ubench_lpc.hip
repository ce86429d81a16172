// Diagnostic #9 (round 3): the LPC chain of a large-order pitch frame in isolation -- autocorr_rows_fast and levinson_row48 as the
// pitch kernel calls them (one wavefront of a 512-thread workgroup, the others idle), microseconds per call.
// hipcc -O3 --offload-arch=gfx950 -ffp-contract=off -fdenormal-fp-math=preserve-sign -I include -I vocoderproject_amd/csrc tools/ubench_lpc.hip -o tools/ubench_lpc
#define VP_TU 99
#include "../vocoderproject_amd/csrc/vp_kernels.hip"
#include <cstdio>
#include <cstdlib>
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("hip error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

template <int V>
__global__ __launch_bounds__(512) void k(double *out, const double *in, int F, int order, int reps, unsigned long long *ticks)
{
    extern __shared__ double sm[];
    lds_f64 *x = (lds_f64 *)sm, *r = x + F + 256, *a = r + 128;
    for (int i = threadIdx.x; i < F + 256; i += blockDim.x) sm[i] = (i < F) ? in[i] : 0.0;
    __syncthreads();
    const int wv = threadIdx.x >> 6;
    if (wv == 7) autocorr_rows_fast<false>(x, F, order, r);
    __syncthreads();
    unsigned long long t0 = wall_clock64();
    if (wv == 7)
        for (int q = 0; q < reps; q++) {
            if (V == 0) autocorr_rows_fast<false>(x, F, order, r);
            if (V == 4) autocorr_rows_fast<true, true>(x, F, order, r);
            if (V == 1) levinson_row48(r, a, order, 101, 1e-9);
            if (V == 2) levinson_fast64(r, a, order, 101, 1e-9);
            if (V == 3) levinson_row16(r, a, min(order, 15), 101, 1e-9);
        }
    unsigned long long t1 = wall_clock64();
    if (threadIdx.x == 511) { ticks[blockIdx.x] = t1 - t0; }
    __syncthreads();
    if (threadIdx.x <= order) out[blockIdx.x * 128 + threadIdx.x] = (V == 0 || V == 4) ? r[threadIdx.x] : a[threadIdx.x];
}

int main(int argc, char **argv)
{
    const int F = argc > 1 ? atoi(argv[1]) : 2048, order = argc > 2 ? atoi(argv[2]) : 48, reps = 20, nb = 256;
    double *in, *out; unsigned long long *ticks;
    CHK(hipMalloc(&in, (F + 256) * 8)); CHK(hipMalloc(&out, nb * 128 * 8)); CHK(hipMalloc(&ticks, nb * 8));
    double *h = (double *)malloc((F + 256) * 8);
    for (int i = 0; i < F + 256; i++) h[i] = sin(0.02 * i) + 0.5 * sin(0.33 * i + 1) + 0.01 * ((i * 7919) % 101 - 50) / 50.0;
    CHK(hipMemcpy(in, h, (F + 256) * 8, hipMemcpyHostToDevice));
    const size_t lds = (F + 256 + 128 + 128) * 8;
    const char *names[5] = {"autocorr_rows_fast<false>", "levinson_row48", "levinson_fast64", "levinson_row16 (order 15)", "autocorr_rows_fast<true,true>"};
    for (int v = 0; v < 5; v++) {
        for (int it = 0; it < 2; it++) {
            if (v == 0) hipLaunchKernelGGL(k<0>, dim3(nb), dim3(512), lds, 0, out, in, F, order, reps, ticks);
            if (v == 1) hipLaunchKernelGGL(k<1>, dim3(nb), dim3(512), lds, 0, out, in, F, order, reps, ticks);
            if (v == 2) hipLaunchKernelGGL(k<2>, dim3(nb), dim3(512), lds, 0, out, in, F, order, reps, ticks);
            if (v == 4) hipLaunchKernelGGL(k<4>, dim3(nb), dim3(512), lds, 0, out, in, F, order, reps, ticks);
            if (v == 3) hipLaunchKernelGGL(k<3>, dim3(nb), dim3(512), lds, 0, out, in, F, order, reps, ticks);
            CHK(hipDeviceSynchronize());
        }
        unsigned long long t[256]; double o[128];
        CHK(hipMemcpy(t, ticks, nb * 8, hipMemcpyDeviceToHost));
        CHK(hipMemcpy(o, out, 128 * 8, hipMemcpyDeviceToHost));
        double mx = 0; for (int i = 0; i < nb; i++) mx = t[i] > mx ? t[i] : mx;
        printf("%-28s F=%d order=%d: %.2f us per call   (out[1]=%.12g out[%d]=%.12g)\n", names[v], F, order, mx / 100.0 / reps, o[1], order, o[order]);
    }
    return 0;
}
