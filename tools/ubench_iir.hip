// Diagnostic: time the product's exact IIR routine in isolation (same code: includes vp_kernels.hip).
#include "../vocoderproject_amd/csrc/vp_kernels.hip"
#include <cstdio>
#include <cstdlib>
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("hip error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

// mode 0: lane 0 of wave 0 only (others wait at the barrier); 1: lane 0 of EVERY wave; 2: all lanes of wave 0
__global__ __launch_bounds__(512) void k_iir(double *out, const double *in, int order, int reps, int mode)
{
    __shared__ double xs[8][256], ys[8][256], as[128];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    for (int i = tid; i < 8 * 256; i += blockDim.x) { (&xs[0][0])[i] = in[i % 128] * 1e-3; (&ys[0][0])[i] = 0; }
    if (tid < 128) as[tid] = (tid == 0) ? 1.0 : in[tid] * ((tid & 1) ? -0.01 : 0.01);
    __syncthreads();
    bool doit = (mode == 0) ? (tid == 0) : (mode == 1) ? (lane == 0) : (mode == 2) ? (wave == 0) : true;
    for (int r = 0; r < reps; r++) {
        if (doit) iir_exact((const lds_f64 *)xs[wave], (lds_f64 *)ys[wave], 256, (const lds_f64 *)as, order, (const lds_f64 *)nullptr, 0, 1.0);
        __syncthreads();
    }
    if (tid == 0) out[blockIdx.x] = ys[0][255];
}

int main()
{
    double *in, *out;
    CHK(hipMalloc(&in, 128 * 8)); CHK(hipMalloc(&out, 4096 * 8));
    double hin[128];
    for (int i = 0; i < 128; i++) hin[i] = 1.0 + 0.01 * i;
    CHK(hipMemcpy(in, hin, sizeof hin, hipMemcpyHostToDevice));
    hipEvent_t e0, e1; CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
    for (int order : {15, 40}) for (int mode = 1; mode < 4; mode += 2) for (int threads : {64, 128, 256, 512}) for (int grid : {256}) {
        const int reps = 50;
        k_iir<<<grid, threads>>>(out, in, order, 2, mode);
        CHK(hipDeviceSynchronize());
        CHK(hipEventRecord(e0));
        k_iir<<<grid, threads>>>(out, in, order, reps, mode);
        CHK(hipEventRecord(e1)); CHK(hipDeviceSynchronize());
        float ms; CHK(hipEventElapsedTime(&ms, e0, e1));
        double h0; CHK(hipMemcpy(&h0, out, 8, hipMemcpyDeviceToHost));
        printf("order %2d mode %d threads %3d grid %3d : %.2f ns per tap (%.1f ns per sample) y=%g\n", order, mode, threads, grid,
               ms * 1e6 / (reps * 256.0 * order), ms * 1e6 / (reps * 256.0), h0);
    }
    return 0;
}
