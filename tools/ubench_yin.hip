// Diagnostic #7: the two-lag YIN difference loop in isolation (4 waves = one per SIMD, F = 1024), to try
// instruction orders without rebuilding the whole kernel.  ns per element per wave.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("hip error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
typedef __attribute__((address_space(3))) double lds_f64;
typedef double d2 __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) d2 lds_d2;
#define SB() __builtin_amdgcn_sched_barrier(0)

template <int V>
__global__ void k(double *out, const double *in, int F, int reps, unsigned long long *ticks)
{
    extern __shared__ double sm[];
    lds_f64 *xs = (lds_f64 *)sm;
    for (int i = threadIdx.x; i < 2 * F + 64; i += blockDim.x) sm[i] = in[i & 1023];
    __syncthreads();
    const int l = threadIdx.x % 221;
    const lds_f64 *xa = xs, *xw = xs + 2 * l;
    double tot = 0.0;
    unsigned long long t0 = wall_clock64();
    for (int r = 0; r < reps; r++) {
        double accA = 0.0, accB = 0.0;
        double w0 = xw[0], w1 = xw[1];
        d2 a0[4], v0[4], a1[4], v1[4];
#define LOAD(A, Vv, I) _Pragma("unroll") for (int u = 0; u < 4; u++) { A[u] = *(const lds_d2 *)(xa + (I) + 2 * u); Vv[u] = *(const lds_d2 *)(xw + (I) + 2 + 2 * u); }
#define COMP(A, Vv) { const double e_[8] = {A[0].x, A[0].y, A[1].x, A[1].y, A[2].x, A[2].y, A[3].x, A[3].y}; \
        const double w_[10] = {w0, w1, Vv[0].x, Vv[0].y, Vv[1].x, Vv[1].y, Vv[2].x, Vv[2].y, Vv[3].x, Vv[3].y}; \
        double dA_[8], dB_[8]; \
        if (V == 0) { _Pragma("unroll") for (int u = 0; u < 8; u++) { double dA = e_[u] - w_[u], dB = e_[u] - w_[u + 1]; accA += dA * dA; accB += dB * dB; } } \
        if (V >= 1) { \
        _Pragma("unroll") for (int u = 0; u < 8; u++) { dA_[u] = e_[u] - w_[u]; dB_[u] = e_[u] - w_[u + 1]; } \
        SB(); \
        _Pragma("unroll") for (int u = 0; u < 8; u++) { dA_[u] = dA_[u] * dA_[u]; dB_[u] = dB_[u] * dB_[u]; } \
        SB(); \
        _Pragma("unroll") for (int u = 0; u < 8; u++) { accA += dA_[u]; accB += dB_[u]; } \
        SB(); } \
        w0 = w_[8]; w1 = w_[9]; }
        if (V <= 1) {
            LOAD(a0, v0, 0)
            for (int i = 0; i < F; i += 16) {
                const bool more1 = i + 8 < F;
                if (more1) { LOAD(a1, v1, i + 8) }
                COMP(a0, v0)
                if (more1) {
                    if (i + 16 < F) { LOAD(a0, v0, i + 16) }
                    COMP(a1, v1)
                }
            }
        }
        if (V == 2) {            // arithmetic only (operands stay in registers): the VALU floor
            LOAD(a0, v0, 0)
            for (int i = 0; i < F; i += 8) { COMP(a0, v0) }
        }
        if (V == 3) {            // loads only
            for (int i = 0; i < F; i += 8) { LOAD(a0, v0, i) accA += a0[0].x + a0[1].x + a0[2].x + a0[3].x; accB += v0[0].x + v0[1].x + v0[2].x + v0[3].x; }
        }
        tot += accA + accB;
    }
    unsigned long long t1 = wall_clock64();
    out[blockIdx.x * blockDim.x + threadIdx.x] = tot;
    if (threadIdx.x == 0 && blockIdx.x == 0) ticks[0] = t1 - t0;
}

template <int V>
void run(const char *name, int threads, int grid)
{
    const int F = 1024, reps = 200;
    double *in, *out; unsigned long long *tk;
    CHK(hipMalloc(&in, 1024 * 8)); CHK(hipMalloc(&out, 1024 * 1024 * 8)); CHK(hipMalloc(&tk, 8));
    double hin[1024];
    for (int i = 0; i < 1024; i++) hin[i] = 1.0 + 1e-3 * (i % 97);
    CHK(hipMemcpy(in, hin, sizeof hin, hipMemcpyHostToDevice));
    k<V><<<grid, threads, (2 * F + 64) * 8>>>(out, in, F, 2, tk);
    CHK(hipDeviceSynchronize());
    k<V><<<grid, threads, (2 * F + 64) * 8>>>(out, in, F, reps, tk);
    CHK(hipDeviceSynchronize());
    unsigned long long t; CHK(hipMemcpy(&t, tk, 8, hipMemcpyDeviceToHost));
    printf("%-64s waves/WG=%d grid=%3d : %.2f ns per element per wave (%.1f us per 1024)\n", name, threads / 64, grid, t * 10.0 / reps / F, t * 10.0 / reps / 1000.0);
}

int main()
{
    for (int threads : {64, 256, 320}) {
        run<0>("0 compiler order (sub,mul,add per element)", threads, 256);
        run<1>("1 batched: 16 subs, 16 muls, 16 adds", threads, 256);
        run<2>("2 arithmetic only (no LDS in the loop)", threads, 256);
        run<3>("3 LDS reads only", threads, 256);
    }
    return 0;
}
