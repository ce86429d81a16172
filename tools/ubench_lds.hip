// Diagnostic #5: what does a serial chain pay for its LDS traffic?  One wave per workgroup.
//   0  fp64 add chain, operands in registers (floor)
//   1  chain + one ds_write_b64 per add, ALL lanes to the SAME address
//   2  chain + one ds_write_b64 per add, lane 0 only
//   3  chain + one ds_write_b64 per add, lane-distinct addresses
//   4  chain whose addend is a wave-uniform ds_read_b64 issued 8 elements ahead
//   5  chain whose addend comes from ONE coalesced read per 16 elements + DPP row broadcast (v_fmac_f64_dpp)
//   6  LDS read latency: dependent pointer chase
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("hip error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
#define SB() __builtin_amdgcn_sched_barrier(0)
typedef __attribute__((address_space(3))) double lds_f64;
typedef __attribute__((address_space(3))) int lds_i32;
#define FM(U) asm volatile("v_fmac_f64_dpp %0, %1, %2 row_newbcast:" #U " row_mask:0xf bank_mask:0xf" : "+v"(run) : "v"(v), "v"(one))

template <int MODE>
__global__ void k(double *out, const double *in, int iters)
{
    __shared__ double sm[2048];
    __shared__ int ch[1024];
    lds_f64 *L = (lds_f64 *)sm;
    lds_i32 *C = (lds_i32 *)ch;
    const int lane = threadIdx.x & 63;
    for (int i = lane; i < 2048; i += 64) sm[i] = in[i & 63] * 1e-3;
    for (int i = lane; i < 1024; i += 64) ch[i] = (i * 17 + 5) & 1023;
    __syncthreads();
    double run = in[64];
    const double one = in[66];        // 1.0
    int p = 0;
    for (int it = 0; it < iters; it++) {
        if (MODE <= 3) {
#pragma unroll
            for (int u = 0; u < 16; u++) {
                run += one; SB();
                if (MODE == 1) L[1024 + u] = run;
                if (MODE == 2) { if (lane == 0) L[1024 + u] = run; }
                if (MODE == 3) L[1024 + u * 64 + lane] = run;
                SB();
            }
        }
        if (MODE == 4) {
            double va[8], vb[8];
#pragma unroll
            for (int u = 0; u < 8; u++) va[u] = L[(it & 63) * 16 + u];
#pragma unroll
            for (int u = 0; u < 8; u++) vb[u] = L[(it & 63) * 16 + 8 + u];
#pragma unroll
            for (int u = 0; u < 8; u++) { run += va[u]; SB(); }
#pragma unroll
            for (int u = 0; u < 8; u++) { run += vb[u]; SB(); }
        }
        if (MODE == 5) {
            double v = L[(it & 63) * 16 + (lane & 15)];
            asm volatile("s_nop 1");
            FM(0); FM(1); FM(2); FM(3); FM(4); FM(5); FM(6); FM(7); FM(8); FM(9); FM(10); FM(11); FM(12); FM(13); FM(14); FM(15);
        }
        if (MODE == 7 || MODE == 8) {
            if (threadIdx.x < 64) {
                double va[8], vb[8], cs[8];
                const int b = (it & 31) * 16;
#pragma unroll
                for (int u = 0; u < 8; u++) va[u] = L[b + u];
#pragma unroll
                for (int u = 0; u < 8; u++) vb[u] = L[b + 8 + u];
#pragma unroll
                for (int u = 0; u < 8; u++) { run += va[u]; cs[u] = run; }
#pragma unroll
                for (int u = 0; u < 8; u++) L[1024 + b + u] = cs[u];
#pragma unroll
                for (int u = 0; u < 8; u++) { run += vb[u]; cs[u] = run; }
#pragma unroll
                for (int u = 0; u < 8; u++) L[1024 + b + 8 + u] = cs[u];
            }
            if (MODE == 8 && (it & 31) == 31) __syncthreads();
        }
        if (MODE == 9) {
            if (threadIdx.x < 64) {
#pragma unroll
                for (int u = 0; u < 16; u++) { run += one; SB(); }
            }
            if ((it & 31) == 31) __syncthreads();
        }
        if (MODE == 6) {
#pragma unroll
            for (int u = 0; u < 16; u++) p = C[p];
        }
    }
    if (threadIdx.x < 64) out[blockIdx.x * 64 + lane] = run + p + L[1024 + lane];
}

template <int MODE>
void run(const char *name, int grid, int threads = 64)
{
    double *in, *out;
    CHK(hipMalloc(&in, 80 * 8)); CHK(hipMalloc(&out, 1024 * 64 * 8));
    double hin[80];
    for (int i = 0; i < 80; i++) hin[i] = 1.0 + 1e-9 * i;
    hin[66] = 1.0;
    CHK(hipMemcpy(in, hin, sizeof hin, hipMemcpyHostToDevice));
    const int iters = 20000;
    hipEvent_t e0, e1; CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
    k<MODE><<<grid, threads>>>(out, in, 100);
    CHK(hipDeviceSynchronize());
    CHK(hipEventRecord(e0));
    k<MODE><<<grid, threads>>>(out, in, iters);
    CHK(hipEventRecord(e1)); CHK(hipDeviceSynchronize());
    float ms; CHK(hipEventElapsedTime(&ms, e0, e1));
    printf("%-78s grid=%4d : %.3f ns per element\n", name, grid, ms * 1e6 / iters / 16.0);
    CHK(hipFree(in)); CHK(hipFree(out));
}

int main()
{
    for (int grid : {1, 256}) {
        run<0>("0 add chain, registers", grid);
        run<1>("1 chain + ds_write_b64 per add, all lanes same address", grid);
        run<2>("2 chain + ds_write_b64 per add, lane 0 only", grid);
        run<3>("3 chain + ds_write_b64 per add, lane-distinct addresses", grid);
        run<4>("4 chain fed by uniform ds_read_b64, 16 reads issued per 16 adds (no prefetch)", grid);
        run<5>("5 chain fed by one coalesced read per 16 + DPP row broadcast fmac", grid);
        run<6>("6 dependent LDS read (latency)", grid);
        run<7>("7 cum-loop replica (8 adds, 8 stores of distinct regs, reads ahead), 1 wave", grid);
        run<8>("8 same, 512 threads: 7 waves parked at s_barrier (512 adds per barrier)", grid, 512);
        run<9>("9 register add chain in wave 0, 7 waves parked at s_barrier", grid, 512);
    }
    return 0;
}
