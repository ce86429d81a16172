// Diagnostic #8 (round 3): inner loops for the cross-correlation form of the YIN difference function,
//     R_k = sum_{i<F} w_i w_{i+k},   k < tauMax = 441,  F = 1024,
// in isolation (512-thread workgroups, waves 1..4 work, one workgroup per CU), to choose the register blocking
// before rebuilding the pitch kernel.  Prints us per frame (= the slowest of the four waves).
//   V0  the round-2 loop: two adjacent lags per lane, every wave all 1024 elements, uniform factor through the DPP row broadcast
//   V1  eight lags per lane, every wave a quarter of the elements, uniform factor through the DPP row broadcast
//   V2  eight lags per lane, quarter of the elements, uniform factor as an LDS broadcast read (plain v_fma_f64)
//   V3  four lags per lane, two waves per half of the elements, DPP
// hipcc -O3 --offload-arch=gfx950 -ffp-contract=off tools/ubench_xcorr.hip -o tools/ubench_xcorr
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("hip error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
typedef __attribute__((address_space(3))) double lds_f64;
typedef double d2 __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) d2 lds_d2;
#define FB(RUN, V, M, U) asm volatile("v_fmac_f64_dpp %0, %1, %2 row_newbcast:" #U " row_mask:0xf bank_mask:0xf" : "+v"(RUN) : "v"(V), "v"(M))

// acc[j] += E(lane U of the row) * W[U + j], j < 8
#define T8(U, W0, W1, W2, W3, W4, W5, W6, W7) FB(a0, E, W0, U); FB(a1, E, W1, U); FB(a2, E, W2, U); FB(a3, E, W3, U); FB(a4, E, W4, U); FB(a5, E, W5, U); FB(a6, E, W6, U); FB(a7, E, W7, U);

template <int V>
__global__ __launch_bounds__(512) void k(double *out, const double *in, int F, int tauMax, int reps, unsigned long long *ticks)
{
    extern __shared__ double sm[];
    lds_f64 *xs = (lds_f64 *)sm;
    for (int i = threadIdx.x; i < 2 * F + 128; i += blockDim.x) sm[i] = in[i & 1023];
    __syncthreads();
    const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63, l16 = lane & 15;
    double tot = 0.0;
    unsigned long long t0 = wall_clock64();
    if (wv >= 1 && wv <= 4)
    for (int r = 0; r < reps; r++) {
        if (V == 0) {
            const int nPairs = (tauMax + 1) >> 1;
            const int l = min((int)threadIdx.x - 64, nPairs - 1);
            const lds_f64 *xa = xs, *xw = xs + 2 * l;
            double accA = 0.0, accB = 0.0;
            double w0 = xw[0], w1 = xw[1];
            double E = xa[l16], En = 0.0;
            d2 v0[8], v1[8];
#define XLOAD(Vv, I) _Pragma("unroll") for (int u = 0; u < 8; u++) Vv[u] = *(const lds_d2 *)(xw + (I) + 2 + 2 * u);
#define XT(U, WA, WB) FB(accA, E, WA, U); FB(accB, E, WB, U);
#define XCOMP(Vv) { XT(0, w0, w1) XT(1, w1, Vv[0].x) XT(2, Vv[0].x, Vv[0].y) XT(3, Vv[0].y, Vv[1].x) XT(4, Vv[1].x, Vv[1].y) \
        XT(5, Vv[1].y, Vv[2].x) XT(6, Vv[2].x, Vv[2].y) XT(7, Vv[2].y, Vv[3].x) XT(8, Vv[3].x, Vv[3].y) XT(9, Vv[3].y, Vv[4].x) \
        XT(10, Vv[4].x, Vv[4].y) XT(11, Vv[4].y, Vv[5].x) XT(12, Vv[5].x, Vv[5].y) XT(13, Vv[5].y, Vv[6].x) XT(14, Vv[6].x, Vv[6].y) \
        XT(15, Vv[6].y, Vv[7].x) w0 = Vv[7].x; w1 = Vv[7].y; }
            XLOAD(v0, 0)
            for (int i = 0; i < F; i += 32) {
                XLOAD(v1, i + 16) En = xa[i + 16 + l16];
                XCOMP(v0)
                E = En;
                if (i + 32 < F) { XLOAD(v0, i + 32) En = xa[i + 32 + l16]; }
                XCOMP(v1)
                E = En;
            }
            tot += accA + accB;
        }
        if (V == 1 || V == 2 || V == 4) {
            const int nl = (tauMax + 7) >> 3;
            const int l = min(lane, nl - 1);
            const int q = wv - 1, i0 = q * (F >> 2), i1 = i0 + (F >> 2);
            const lds_f64 *xa = xs, *xw = xs + 8 * l;
            double a0 = 0, a1 = 0, a2 = 0, a3 = 0, a4 = 0, a5 = 0, a6 = 0, a7 = 0;
            d2 c[4], v0[8], v1[8];                        // carry: xw[i .. i+7]; new: xw[i+8 .. i+23]
#define CLOAD(I) _Pragma("unroll") for (int u = 0; u < 4; u++) c[u] = *(const lds_d2 *)(xw + (I) + 2 * u);
#define NLOAD(Vv, I) _Pragma("unroll") for (int u = 0; u < 8; u++) Vv[u] = *(const lds_d2 *)(xw + (I) + 8 + 2 * u);
            CLOAD(i0)
            NLOAD(v0, i0)
            double E = xa[i0 + l16], En = 0.0;
#define COMP8(Vv) { \
            T8(0, c[0].x, c[0].y, c[1].x, c[1].y, c[2].x, c[2].y, c[3].x, c[3].y) \
            T8(1, c[0].y, c[1].x, c[1].y, c[2].x, c[2].y, c[3].x, c[3].y, Vv[0].x) \
            T8(2, c[1].x, c[1].y, c[2].x, c[2].y, c[3].x, c[3].y, Vv[0].x, Vv[0].y) \
            T8(3, c[1].y, c[2].x, c[2].y, c[3].x, c[3].y, Vv[0].x, Vv[0].y, Vv[1].x) \
            T8(4, c[2].x, c[2].y, c[3].x, c[3].y, Vv[0].x, Vv[0].y, Vv[1].x, Vv[1].y) \
            T8(5, c[2].y, c[3].x, c[3].y, Vv[0].x, Vv[0].y, Vv[1].x, Vv[1].y, Vv[2].x) \
            T8(6, c[3].x, c[3].y, Vv[0].x, Vv[0].y, Vv[1].x, Vv[1].y, Vv[2].x, Vv[2].y) \
            T8(7, c[3].y, Vv[0].x, Vv[0].y, Vv[1].x, Vv[1].y, Vv[2].x, Vv[2].y, Vv[3].x) \
            T8(8, Vv[0].x, Vv[0].y, Vv[1].x, Vv[1].y, Vv[2].x, Vv[2].y, Vv[3].x, Vv[3].y) \
            T8(9, Vv[0].y, Vv[1].x, Vv[1].y, Vv[2].x, Vv[2].y, Vv[3].x, Vv[3].y, Vv[4].x) \
            T8(10, Vv[1].x, Vv[1].y, Vv[2].x, Vv[2].y, Vv[3].x, Vv[3].y, Vv[4].x, Vv[4].y) \
            T8(11, Vv[1].y, Vv[2].x, Vv[2].y, Vv[3].x, Vv[3].y, Vv[4].x, Vv[4].y, Vv[5].x) \
            T8(12, Vv[2].x, Vv[2].y, Vv[3].x, Vv[3].y, Vv[4].x, Vv[4].y, Vv[5].x, Vv[5].y) \
            T8(13, Vv[2].y, Vv[3].x, Vv[3].y, Vv[4].x, Vv[4].y, Vv[5].x, Vv[5].y, Vv[6].x) \
            T8(14, Vv[3].x, Vv[3].y, Vv[4].x, Vv[4].y, Vv[5].x, Vv[5].y, Vv[6].x, Vv[6].y) \
            T8(15, Vv[3].y, Vv[4].x, Vv[4].y, Vv[5].x, Vv[5].y, Vv[6].x, Vv[6].y, Vv[7].x) \
            c[0] = Vv[4]; c[1] = Vv[5]; c[2] = Vv[6]; c[3] = Vv[7]; }
            if (V == 1) {
                for (int i = i0; i < i1; i += 32) {
                    NLOAD(v1, i + 16) En = xa[i + 16 + l16];
                    COMP8(v0)
                    E = En;
                    if (i + 32 < i1) { NLOAD(v0, i + 32) En = xa[i + 32 + l16]; }
                    COMP8(v1)
                    E = En;
                }
            } else if (V == 4) {
                // uniform factor through v_readlane into scalar registers (one LDS read per 16 elements), eight plain fused multiply-adds per element
                double ev = xa[i0 + l16], evn = 0.0;
                for (int i = i0; i < i1; i += 16) {
                    double w[24];
#pragma unroll
                    for (int u = 0; u < 4; u++) { w[2 * u] = c[u].x; w[2 * u + 1] = c[u].y; }
#pragma unroll
                    for (int u = 0; u < 8; u++) { w[8 + 2 * u] = v0[u].x; w[9 + 2 * u] = v0[u].y; }
                    if (i + 16 < i1) { NLOAD(v1, i + 16) evn = xa[i + 16 + l16]; }
#pragma unroll
                    for (int u = 0; u < 16; u++) {
                        const double eu = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(ev), u), __builtin_amdgcn_readlane(__double2loint(ev), u));
                        a0 = __builtin_fma(eu, w[u], a0); a1 = __builtin_fma(eu, w[u + 1], a1); a2 = __builtin_fma(eu, w[u + 2], a2);
                        a3 = __builtin_fma(eu, w[u + 3], a3); a4 = __builtin_fma(eu, w[u + 4], a4); a5 = __builtin_fma(eu, w[u + 5], a5);
                        a6 = __builtin_fma(eu, w[u + 6], a6); a7 = __builtin_fma(eu, w[u + 7], a7);
                    }
                    c[0] = v0[4]; c[1] = v0[5]; c[2] = v0[6]; c[3] = v0[7];
#pragma unroll
                    for (int u = 0; u < 8; u++) v0[u] = v1[u];
                    ev = evn;
                }
            } else {
                // uniform factor as an LDS broadcast read, eight plain fused multiply-adds per element
                for (int i = i0; i < i1; i += 16) {
                    double e[16];
#pragma unroll
                    for (int u = 0; u < 16; u++) e[u] = xa[i + u];
                    double w[24];
#pragma unroll
                    for (int u = 0; u < 4; u++) { w[2 * u] = c[u].x; w[2 * u + 1] = c[u].y; }
#pragma unroll
                    for (int u = 0; u < 8; u++) { w[8 + 2 * u] = v0[u].x; w[9 + 2 * u] = v0[u].y; }
                    if (i + 16 < i1) { NLOAD(v1, i + 16) }
#pragma unroll
                    for (int u = 0; u < 16; u++) {
                        a0 = __builtin_fma(e[u], w[u], a0); a1 = __builtin_fma(e[u], w[u + 1], a1); a2 = __builtin_fma(e[u], w[u + 2], a2);
                        a3 = __builtin_fma(e[u], w[u + 3], a3); a4 = __builtin_fma(e[u], w[u + 4], a4); a5 = __builtin_fma(e[u], w[u + 5], a5);
                        a6 = __builtin_fma(e[u], w[u + 6], a6); a7 = __builtin_fma(e[u], w[u + 7], a7);
                    }
                    c[0] = v0[4]; c[1] = v0[5]; c[2] = v0[6]; c[3] = v0[7];
#pragma unroll
                    for (int u = 0; u < 8; u++) v0[u] = v1[u];
                }
            }
            tot += ((a0 + a1) + (a2 + a3)) + ((a4 + a5) + (a6 + a7));
        }
        if (V == 3) {
            // four lags per lane: 111 lanes = two waves per half of the elements
            const int nl = (tauMax + 3) >> 2;
            const int q = (wv - 1) >> 1, i0 = q * (F >> 1), i1 = i0 + (F >> 1);
            const int l = min(((wv - 1) & 1) * 64 + lane, nl - 1);
            const lds_f64 *xa = xs, *xw = xs + 4 * l;
            double a0 = 0, a1 = 0, a2 = 0, a3 = 0;
            d2 c[2], v0[8], v1[8];                        // carry: xw[i .. i+3]; new: xw[i+4 .. i+19]
            c[0] = *(const lds_d2 *)(xw + i0); c[1] = *(const lds_d2 *)(xw + i0 + 2);
#define N4LOAD(Vv, I) _Pragma("unroll") for (int u = 0; u < 8; u++) Vv[u] = *(const lds_d2 *)(xw + (I) + 4 + 2 * u);
#define T4(U, W0, W1, W2, W3) FB(a0, E, W0, U); FB(a1, E, W1, U); FB(a2, E, W2, U); FB(a3, E, W3, U);
#define COMP4(Vv) { \
            T4(0, c[0].x, c[0].y, c[1].x, c[1].y) T4(1, c[0].y, c[1].x, c[1].y, Vv[0].x) T4(2, c[1].x, c[1].y, Vv[0].x, Vv[0].y) T4(3, c[1].y, Vv[0].x, Vv[0].y, Vv[1].x) \
            T4(4, Vv[0].x, Vv[0].y, Vv[1].x, Vv[1].y) T4(5, Vv[0].y, Vv[1].x, Vv[1].y, Vv[2].x) T4(6, Vv[1].x, Vv[1].y, Vv[2].x, Vv[2].y) T4(7, Vv[1].y, Vv[2].x, Vv[2].y, Vv[3].x) \
            T4(8, Vv[2].x, Vv[2].y, Vv[3].x, Vv[3].y) T4(9, Vv[2].y, Vv[3].x, Vv[3].y, Vv[4].x) T4(10, Vv[3].x, Vv[3].y, Vv[4].x, Vv[4].y) T4(11, Vv[3].y, Vv[4].x, Vv[4].y, Vv[5].x) \
            T4(12, Vv[4].x, Vv[4].y, Vv[5].x, Vv[5].y) T4(13, Vv[4].y, Vv[5].x, Vv[5].y, Vv[6].x) T4(14, Vv[5].x, Vv[5].y, Vv[6].x, Vv[6].y) T4(15, Vv[5].y, Vv[6].x, Vv[6].y, Vv[7].x) \
            c[0] = Vv[6]; c[1] = Vv[7]; }
            N4LOAD(v0, i0)
            double E = xa[i0 + l16], En = 0.0;
            for (int i = i0; i < i1; i += 32) {
                N4LOAD(v1, i + 16) En = xa[i + 16 + l16];
                COMP4(v0)
                E = En;
                if (i + 32 < i1) { N4LOAD(v0, i + 32) En = xa[i + 32 + l16]; }
                COMP4(v1)
                E = En;
            }
            tot += (a0 + a1) + (a2 + a3);
        }
    }
    unsigned long long t1 = wall_clock64();
    out[blockIdx.x * blockDim.x + threadIdx.x] = tot;
    if (lane == 0 && blockIdx.x == 0 && wv >= 1 && wv <= 4) ticks[wv] = t1 - t0;
}

template <int V>
void run(const char *name)
{
    const int F = 1024, reps = 200;
    double *in, *out; unsigned long long *tk;
    CHK(hipMalloc(&in, 1024 * 8)); CHK(hipMalloc(&out, 1024 * 1024 * 8)); CHK(hipMalloc(&tk, 64));
    double hin[1024];
    for (int i = 0; i < 1024; i++) hin[i] = 1.0 + 1e-3 * (i % 97);
    CHK(hipMemcpy(in, hin, sizeof hin, hipMemcpyHostToDevice));
    CHK(hipMemset(tk, 0, 64));
    k<V><<<256, 512, (2 * F + 128) * 8>>>(out, in, F, 441, 2, tk);
    CHK(hipDeviceSynchronize());
    k<V><<<256, 512, (2 * F + 128) * 8>>>(out, in, F, 441, reps, tk);
    CHK(hipDeviceSynchronize());
    unsigned long long t[8]; CHK(hipMemcpy(t, tk, 64, hipMemcpyDeviceToHost));
    unsigned long long mx = 0;
    for (int w = 1; w <= 4; w++) mx = t[w] > mx ? t[w] : mx;
    double o0; CHK(hipMemcpy(&o0, out + 64, 8, hipMemcpyDeviceToHost));
    printf("%-72s %.2f us per frame (waves: %.2f %.2f %.2f %.2f) chk %.6g\n", name, mx * 0.01 / reps, t[1] * 0.01 / reps, t[2] * 0.01 / reps, t[3] * 0.01 / reps,
           t[4] * 0.01 / reps, o0);
}

int main()
{
    run<0>("V0 two lags per lane, all elements per wave, DPP (round 2)");
    run<1>("V1 eight lags per lane, a quarter of the elements per wave, DPP");
    run<2>("V2 eight lags per lane, quarter, LDS broadcast + plain fma");
    run<3>("V3 four lags per lane, two waves per half, DPP");
    run<4>("V4 eight lags per lane, quarter, v_readlane -> SGPR factor + plain fma");
    return 0;
}
