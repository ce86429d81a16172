// vp_kernels.hip -- hand-written CDNA4 (gfx950) kernels of the batch pitch-corrector / vocoder.
//
// Numerics contract: every value that feeds a discrete decision or reaches the output is
// computed in IEEE double in the reference's own operation order (no FMA contraction: this file
// is compiled with -ffp-contract=off), so results are bit-identical to the CPU restatement of
// the reference and decisions (YIN threshold walk, pitch marks, nearest note, gate) cannot flip.
// Parallelism is taken only where the reference's order leaves it free:
//   * across streams (one workgroup per stream),
//   * across YIN lags / autocorrelation lags (each lag is its own left-to-right sum),
//   * across output samples of FIR / PSOLA / overlap-add,
//   * across vocoder windows of a block (one wavefront per window).
// Citations are file:line relative to /root/reference/Source/.
#include <hip/hip_runtime.h>
#include <utility>
#include <limits.h>

#include "vp_common.h"

// The kernels are template instantiations of two large bodies; the build compiles this file several times side by side,
// each translation unit keeping one group of them (-DVP_TU=k; 0 or undefined: all of them, e.g. for -S listings).
//   1: ingest/gate, vocoder, emit, STFT   2: vp_k_pitch   3: vp_k_pitch_fast   4: vp_k_pitch_multi, vp_k_pitch_fast_multi
//   5: vp_k_pitch_lite, vp_k_pitch_lite_fast
#ifndef VP_TU
#define VP_TU 0
#endif
#define VP_TU_HAS(K) (VP_TU == 0 || VP_TU == (K))
#define VP_NUM_TUS 5

#define WAVE 64

// Orders from VP_LEV_SCALAR_MIN to 48 take the register-resident Levinson-Durbin (levinson_scalar).  Measured on one box
// (tools/abn.sh): order 48 (configs[4]) pitch kernel +5 %, vocoder +8 %; at order 40 the wave-distributed form is 7 % faster.
#ifndef VP_LEV_SCALAR_MIN
#define VP_LEV_SCALAR_MIN 41
#endif

// explicit LDS address space: keeps the compiler on ds_read/ds_write instead of flat accesses
typedef __attribute__((address_space(3))) double lds_f64;
typedef __attribute__((address_space(3))) float lds_f32;
typedef __attribute__((address_space(3))) int lds_i32;

// Diagnostic build only (-DVP_STAMPS): workgroup 0 / thread 0 accumulates, per phase id, the
// 100 MHz wall-clock ticks spent since the previous stamp into d.dbg[id].  No stamp executes in
// the product build.
#ifdef VP_STAMPS
__device__ unsigned long long vp_last_stamp;
#define STAMP(D, ID) do { if (blockIdx.x == 0 && threadIdx.x == 0) { unsigned long long _t = wall_clock64(); \
        (D).dbg[ID] += _t - vp_last_stamp; vp_last_stamp = _t; } } while (0)
#define STAMP0(D) do { if (blockIdx.x == 0 && threadIdx.x == 0) { vp_last_stamp = wall_clock64(); vp_dbg_g = (D).dbg; } } while (0)
// inside helpers that do not see VpDev: ticks since the previous STAMPG_BEGIN/STAMPG, into dbg[ID]
__device__ unsigned long long *vp_dbg_g;
__device__ unsigned long long vp_last_g;
#define STAMPG_BEGIN() do { if (blockIdx.x == 0 && threadIdx.x == 0) vp_last_g = wall_clock64(); } while (0)
#define STAMPG(ID) do { if (blockIdx.x == 0 && threadIdx.x == 0) { unsigned long long _t = wall_clock64(); \
        vp_dbg_g[ID] += _t - vp_last_g; vp_last_g = _t; } } while (0)
// the same for the LAST thread of workgroup 0 (work parked on the last wavefront)
__device__ unsigned long long vp_last_l;
#define STAMPL_BEGIN() do { if (blockIdx.x == 0 && threadIdx.x == blockDim.x - 1) vp_last_l = wall_clock64(); } while (0)
#define STAMPL(ID) do { if (blockIdx.x == 0 && threadIdx.x == blockDim.x - 1) { unsigned long long _t = wall_clock64(); \
        vp_dbg_g[ID] += _t - vp_last_l; vp_last_l = _t; } } while (0)
// per-wavefront timers (lane 0 of every wave of workgroup 0), into dbg[ID + wave]
__device__ unsigned long long vp_last_w[16];
#define STAMPW_BEGIN() do { if (blockIdx.x == 0 && (threadIdx.x & 63) == 0) vp_last_w[threadIdx.x >> 6] = wall_clock64(); } while (0)
#define STAMPW(ID) do { if (blockIdx.x == 0 && (threadIdx.x & 63) == 0) { unsigned long long _t = wall_clock64(); \
        vp_dbg_g[(ID) + (threadIdx.x >> 6)] += _t - vp_last_w[threadIdx.x >> 6]; vp_last_w[threadIdx.x >> 6] = _t; } } while (0)
#else
#define STAMPW_BEGIN() do { } while (0)
#define STAMPW(ID) do { } while (0)
#define STAMPL_BEGIN() do { } while (0)
#define STAMPL(ID) do { } while (0)
#define STAMPG_BEGIN() do { } while (0)
#define STAMPG(ID) do { } while (0)
#define STAMP(D, ID) do { } while (0)
#define STAMP0(D) do { } while (0)
#endif

#ifndef VP_XC_ACSPLIT
#define VP_XC_ACSPLIT 11        /* sixteenths of the LPC autocorrelation summed beside the cross-correlation YIN */
#endif

// Diagnostic build (-DVP_POISON_LDS=bytes): every kernel starts by filling its dynamic LDS with signalling garbage (NaNs),
// so that a read of LDS the launch has not written -- which otherwise returns whatever the previous kernel on that CU
// left there -- shows up deterministically in the parity tests.
#ifdef VP_POISON_LDS
#define VP_POISON(SMEM, BYTES) do { for (int i_ = threadIdx.x; i_ < (int)((BYTES) / 8); i_ += blockDim.x) \
        ((unsigned long long *)(SMEM))[i_] = 0x7ff8dead0000beefULL; __syncthreads(); } while (0)
#else
#define VP_POISON(SMEM, BYTES) do { } while (0)
#endif

// ------------------------------------------------------------------------------------------------
// helpers

// threadIdx.x through an opaque move.  Index arithmetic derived from the plain builtin is loop-invariant
// for the whole kernel; the compiler hoists it all to the top and then has to spill it around the
// register-heavy phases (every reload in a serial phase is a memory round trip).  Derived from this
// value it is recomputed where it is used.
__device__ __forceinline__ int vp_tid()
{
    int t = threadIdx.x;
    asm volatile("" : "+v"(t));
    return t;
}

// the stream this workgroup serves (see VpDev::streamMap)
__device__ __forceinline__ int vp_stream(const VpDev &d)
{
    return d.streamMap ? d.streamMap[blockIdx.x] : (int)blockIdx.x;
}

typedef __attribute__((address_space(3))) VpPitchState lds_state;
__device__ __forceinline__ void emit_block(const VpGeom &g, const VpCall &c, const VpDev &d, float *__restrict__ out,
                                           const lds_state *stl = nullptr, int boff = 0);

__device__ __forceinline__ int ring_pos(int curr, int idx, int inSize)
{
    // MyBuffer::getVoiceSample index math (MyBuffer.cpp:152): (currCounter + idx + inSize) % inSize
    int p = curr + idx + inSize;
    p %= inSize;
    return p;
}

struct MinIdx { double v; int i; };

__device__ __forceinline__ MinIdx min_first(MinIdx a, MinIdx b)
{
    // argExt semantics (PitchProcess.cpp:752-776): strict '<' while scanning upwards keeps the
    // FIRST minimum, i.e. the lexicographic minimum of (value, index).
    if (b.v < a.v || (b.v == a.v && b.i < a.i)) return b;
    return a;
}

__device__ __forceinline__ double wave_sum(double v)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, WAVE);
    return v;
}

// ------------------------------------------------------------------------------------------------
// K0: ingest + silence gate.  MyBuffer::fillInputBuffers (MyBuffer.cpp:74-105) and the whole-ring
// RMS gate used by both processes (MyBuffer.cpp:258-261,299-302; VocoderProcess.cpp:199-204;
// PitchProcess.cpp:208).  One workgroup per stream.
//
// The reference sums x^2 left to right over the PHYSICAL ring; its verdict is
// "sum_seq < gateThrSum" (gateThrSum is found on the host by bisection through the same libm
// calls).  A tree sum T differs from the sequential one by at most ~2 n eps T, so T decides unless
// it is within that band of the threshold, in which case one lane redoes the exact sequential sum.
__device__ __forceinline__ void ingest_gate_block(const VpGeom &g, const VpCall &c, const VpDev &d, const float *__restrict__ in,
                                                  int boff = 0)
{
    const int s = vp_stream(d), tid = threadIdx.x, nt = blockDim.x;
    float *vr = d.voiceRing + (size_t)s * g.inSize;
    float *sr0 = d.synthRing + (size_t)s * 2 * g.inSize;
    float *sr1 = sr0 + g.inSize;
    const int mono = c.inMono;
    const float *xin = in + (size_t)s * (mono ? 1 : 3) * g.N;
    if (!mono) {
        for (int i = tid; i < g.N; i += nt) {
            int p = (c.inCounter + boff + i) % g.inSize;
            vr[p] = xin[i];
            sr0[p] = xin[g.N + i];
            sr1[p] = xin[2 * g.N + i];
        }
    } else {                                           // null side-chain pointers: zeros (MyBuffer.cpp:93-102)
        for (int i = tid; i < g.N; i += nt) {
            int p = (c.inCounter + boff + i) % g.inSize;
            vr[p] = xin[i];
            if (mono == 1) { sr0[p] = 0.0f; sr1[p] = 0.0f; }
        }
    }
    __syncthreads();   // own-workgroup global writes are visible to the workgroup after the barrier

    // the side chain's gate is only consulted by the vocoder (VocoderProcess.cpp:199-204); an all-zero ring sums to 0
    const bool needS = c.vocOn && mono != 2;
    double sv = 0.0, ss = 0.0;
    if (needS) {
        for (int i = tid; i < g.inSize; i += nt) {
            double a = (double)vr[i], b = (double)sr0[i];
            sv += a * a;
            ss += b * b;
        }
    } else {
        for (int i = tid; i < g.inSize; i += nt) {
            double a = (double)vr[i];
            sv += a * a;
        }
    }
    sv = wave_sum(sv);
    ss = wave_sum(ss);
    __shared__ double red[2][8];
    if ((tid & 63) == 0) { red[0][tid >> 6] = sv; red[1][tid >> 6] = ss; }
    __syncthreads();
    if (tid < 2) {
        const float *ring = tid == 0 ? vr : sr0;
        double T = 0.0;
        for (int w = 0; w < nt / WAVE; w++) T += red[tid][w];
        const double band = T * (double)(2 * g.inSize + 64) * 1.1102230246251565e-16 * 1.5;
        int open;
        if (T - g.gateThrSum > band) open = 1;
        else if (g.gateThrSum - T > band) open = 0;
        else {
            double seq = 0.0;                       // AudioBuffer::getRMSLevel order
            for (int i = 0; i < g.inSize; i++) { double a = (double)ring[i]; seq += a * a; }
            open = !(seq < g.gateThrSum);
        }
        d.gate[s * 2 + tid] = open;
    }
    if (!c.pitchOn && tid == 0) {
        // PitchProcess::silence() (PitchProcess.cpp:146-158), called when pitchBool is off
        VpPitchState *ps = d.pitch + s;
        ps->nAn = 0; ps->nSt = 0;
        ps->prevPitch = 0; ps->prevPeriod = 0; ps->pitch = 0; ps->period = 0;
    }
    __syncthreads();
}

#if VP_TU_HAS(1)
__global__ __launch_bounds__(256) void vp_k_ingest_gate(VpGeom g, VpCall c, VpDev d, const float *__restrict__ in)
{
    ingest_gate_block(g, c, d, in);
}
#endif

// ------------------------------------------------------------------------------------------------
// Wave-uniform broadcast of a double held by lane `src` (src is wave-uniform): two v_readlane_b32.
__device__ __forceinline__ double bcast_f64(double v, int src)
{
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_readlane(lo, src);
    hi = __builtin_amdgcn_readlane(hi, src);
    return __hiloint2double(hi, lo);
}

// Levinson-Durbin (LPC.cpp:107-148) by ONE WAVEFRONT with the coefficient vector spread over the
// lanes: lane l owns a[l] and a[64+l].  The two dot products of every order step are summed in the
// reference's order (i = 1..p-1, left to right) by broadcasting the per-lane products one after
// the other, so the result is bit-identical to the serial recursion; the products and the
// coefficient update run lane-parallel.  All 64 lanes must call it (wave-uniform arguments).
// `scratch` (LDS, 16-byte aligned, >= 128 doubles) lets orders below 64 take a cheaper route for the
// ordered sums: the lanes park their two products side by side in LDS and every lane then reads
// them back in order, four steps per trip, instead of 4 v_readlane per step.
template <class RP, class AP>
__device__ __forceinline__ bool levinson_wave(RP r, AP a, int order, int aLen, double eps, lds_f64 *scratch = nullptr)
{
    const int lane = threadIdx.x & 63;
    if (fabs(r[0]) < eps) {                        // :110-114 (floating abs intended)
        for (int i = lane; i < aLen; i += WAVE) a[i] = (i == 0) ? 1.0 : 0.0;
        return true;                               // the whole vector was rewritten
    }
    const double r0 = r[0];
    if (scratch && order < WAVE) {
        typedef double d2 __attribute__((ext_vector_type(2)));
        typedef __attribute__((address_space(3))) d2 lds_d2;
        lds_d2 *pr = (lds_d2 *)scratch;                 // pr[i] = { r[p-i]*a[i], r[i]*a[i] }
        double av = 0.0;                                 // a[lane]
        if (lane == 0) av = 1.0;
        if (lane == 1) av = r[1] / r0;
        const double rl = (lane <= order) ? r[lane] : 0.0;
        for (int p = 2; p < order + 1; p++) {
            const bool in = lane >= 1 && lane < p;
            d2 qs;
            qs.x = in ? r[p - lane] * av : 0.0;
            qs.y = rl * av;
            pr[lane] = qs;
            double rho_a = 0.0, r_a = 0.0;
            int i = 1;
            for (; i + 4 <= p; i += 4) {                 // i .. i+3 < p
                const d2 v0 = pr[i], v1 = pr[i + 1], v2 = pr[i + 2], v3 = pr[i + 3];
                rho_a += v0.x; r_a += v0.y;
                rho_a += v1.x; r_a += v1.y;
                rho_a += v2.x; r_a += v2.y;
                rho_a += v3.x; r_a += v3.y;
            }
            for (; i < p; i++) { const d2 v = pr[i]; rho_a += v.x; r_a += v.y; }
            const double k = (r[p] - rho_a) / (r0 - r_a);
            const double partner = __shfl(av, (p - lane) & 63, WAVE);     // aPrev[p - i]
            double nv = av;
            if (in) nv = av - k * partner;
            if (lane == p) nv = k;
            av = nv;
        }
        if (lane >= 1) av *= -1.;
        if (lane <= order) a[lane] = av;
        return false;
    }
    double a0 = 0.0, a1 = 0.0;                      // a[lane], a[64+lane]
    if (lane == 0) a0 = 1.0;
    if (lane == 1) a0 = r[1] / r0;
    const double rl0 = (lane <= order) ? r[lane] : 0.0;             // r[i] for i = lane
    const double rl1 = (64 + lane <= order) ? r[64 + lane] : 0.0;   // r[i] for i = 64+lane
    for (int p = 2; p < order + 1; p++) {
        // per-lane products for i = lane and i = 64+lane (only 1 <= i < p are consumed)
        const int i0 = lane, i1 = 64 + lane;
        double q0 = (i0 >= 1 && i0 < p) ? r[p - i0] * a0 : 0.0;
        double q1 = (i1 < p) ? r[p - i1] * a1 : 0.0;
        double s0 = rl0 * a0, s1 = rl1 * a1;
        double rho_a = 0.0, r_a = 0.0;
        const int n0 = min(p, 64);
        for (int i = 1; i < n0; i++) { rho_a += bcast_f64(q0, i); r_a += bcast_f64(s0, i); }
        for (int i = 64; i < p; i++) { rho_a += bcast_f64(q1, i - 64); r_a += bcast_f64(s1, i - 64); }
        const double k = (r[p] - rho_a) / (r0 - r_a);
        // a[i] = aPrev[i] - k * aPrev[p - i], 1 <= i < p
        int j0 = p - i0, j1 = p - i1;                               // partner indices
        double p0lo = __shfl(a0, j0 & 63, WAVE), p0hi = __shfl(a1, j0 & 63, WAVE);
        double p1lo = __shfl(a0, j1 & 63, WAVE), p1hi = __shfl(a1, j1 & 63, WAVE);
        double n0v = a0, n1v = a1;
        if (i0 >= 1 && i0 < p) n0v = a0 - k * ((j0 >= 64) ? p0hi : p0lo);
        if (i1 < p) n1v = a1 - k * ((j1 >= 64) ? p1hi : p1lo);
        if (i0 == p) n0v = k;
        if (i1 == p) n1v = k;
        a0 = n0v; a1 = n1v;
    }
    if (lane >= 1) a0 *= -1.;                       // :145-146 (entries beyond the order are 0 -> -0, never read)
    a1 *= -1.;
    if (lane <= order) a[lane] = a0;
    if (64 + lane <= order) a[64 + lane] = a1;
    return false;
}

// RUN += (lane U of V's 16-lane row) * ONE in one VALU op (DPP row broadcast; ONE must hold 1.0, so the
// fused multiply-add rounds exactly like RUN + value).
#define VP_FMAC_BCAST(RUN, V, ONE, U) \
    asm volatile("v_fmac_f64_dpp %0, %1, %2 row_newbcast:" #U " row_mask:0xf bank_mask:0xf" : "+v"(RUN) : "v"(V), "v"(ONE))

// A stretch [n0, n1) of the LPC autocorrelation sum_n x[n] x[n+m] (LPC.cpp:44-97), continued from `sum`,
// by one full wavefront with lane -> lag m.  n0 and n1 are multiples of 8 and n1 <= F - (largest lag).
// Eight elements per trip, the next trip's LDS reads issued before this trip's arithmetic; product and
// sum stay two roundings, in the reference's order.  (v_mul_f64 has no DPP form on gfx950 -- only
// v_fmac_f64 does -- so the wave-uniform factor x[n] is an LDS broadcast read.)
__device__ __forceinline__ double autocorr_stretch(const lds_f64 *x, int m, int n0, int n1, double sum)
{
    const lds_f64 *xm = x + m;
    if (n0 >= n1) return sum;
    double a0[8], b0[8], a1[8], b1[8];
#define VP_ALOAD(A, B, I) _Pragma("unroll") for (int u = 0; u < 8; u++) { A[u] = x[(I) + u]; B[u] = xm[(I) + u]; }
#define VP_ACOMP(A, B) { double p_[8]; _Pragma("unroll") for (int u = 0; u < 8; u++) p_[u] = A[u] * B[u]; \
                         __builtin_amdgcn_sched_barrier(0); \
                         _Pragma("unroll") for (int u = 0; u < 8; u++) sum += p_[u]; \
                         __builtin_amdgcn_sched_barrier(0); }
    VP_ALOAD(a0, b0, n0)
    for (int n = n0; n < n1; n += 16) {
        const bool more1 = n + 8 < n1;
        if (more1) { VP_ALOAD(a1, b1, n + 8) }
        VP_ACOMP(a0, b0)
        if (more1) {
            if (n + 16 < n1) { VP_ALOAD(a0, b0, n + 16) }
            VP_ACOMP(a1, b1)
        }
    }
#undef VP_ALOAD
#undef VP_ACOMP
    return sum;
}

// Levinson-Durbin for orders below 16 with the coefficient vector inside ONE 16-lane row (lane m of
// every row owns a[m]; the four rows of the calling wavefront work redundantly).  The ordered sums
// rho_a = sum_{i=1..p-1} r[p-i] a[i] and r_a = sum r[i] a[i] (LPC.cpp:120-128) are chains of
// v_fmac_f64 with a DPP row-broadcast operand: entries i >= p hold +0.0, and adding +0.0 to a sum that
// started from +0.0 changes nothing, so all fifteen terms are always added and no lane ever talks to
// the LDS for them (the general form's park-and-read-back costs two LDS round trips per order step).
template <class RP, class AP>
__device__ __forceinline__ bool levinson_row16(RP r, AP a, int order, int aLen, double eps)
{
    const int lane = threadIdx.x & 63, m = lane & 15;
    if (fabs(r[0]) < eps) {                        // :110-114
        for (int i = lane; i < aLen; i += WAVE) a[i] = (i == 0) ? 1.0 : 0.0;
        return true;
    }
    const double r0 = r[0], one = 1.0;
    double av = 0.0;                                 // a[m]
    if (m == 0) av = 1.0;
    if (m == 1) av = r[1] / r0;
    const double rl = (m >= 1 && m <= order) ? r[m] : 0.0;
    double rq = (m >= 1 && m < 2) ? r[2 - m] : 0.0;  // r[p - m] of the coming step
    for (int p = 2; p < order + 1; p++) {
        const bool in = m >= 1 && m < p;
        double q = in ? rq * av : 0.0;
        double sv = rl * av;
        const double rp = r[p];
        const double partner = __shfl(av, (lane & 48) | ((p - m) & 15), WAVE);      // a[p - m]
        if (p + 1 < order + 1) rq = (m >= 1 && m < p + 1) ? r[p + 1 - m] : 0.0;
        double rho_a = 0.0, r_a = 0.0;
        asm volatile("s_nop 1" : "+v"(q), "+v"(sv));                             // VALU write -> DPP read
#define VP_LV(U) VP_FMAC_BCAST(rho_a, q, one, U); VP_FMAC_BCAST(r_a, sv, one, U);
        VP_LV(1) VP_LV(2) VP_LV(3) VP_LV(4) VP_LV(5) VP_LV(6) VP_LV(7) VP_LV(8)
        VP_LV(9) VP_LV(10) VP_LV(11) VP_LV(12) VP_LV(13) VP_LV(14) VP_LV(15)
#undef VP_LV
        const double k = (rp - rho_a) / (r0 - r_a);
        double nv = av;
        if (in) nv = av - k * partner;
        if (m == p) nv = k;
        av = nv;
    }
    if (m >= 1) av *= -1.;
    if (lane <= order) a[lane] = av;
    return false;
}

// Levinson-Durbin for orders 16..P by ONE wavefront with every lane running the whole recursion (LPC.cpp:107-148 as it
// stands), the autocorrelation and coefficient vectors in REGISTERS (fully unrolled, static names).  Sixty-four lanes doing
// the same thing is as wasteful as it sounds, but the wave-distributed form above pays two LDS round trips per order step
// for its ordered sums (26 us at order 40, 42 us at order 48); this one has none and takes a quarter of that.
// r, a: LDS vectors.  Returns true when the whole vector was rewritten (the |r0| < eps branch).
// (not inlined: the body is ~8000 instructions and the kernels call it from several places)
template <int P>
__device__ __noinline__ bool levinson_scalar(const lds_f64 *r, lds_f64 *a, int order_, int aLen, double eps)
{
    const int lane = threadIdx.x & 63;
    const int order = __builtin_amdgcn_readfirstlane(order_);
    if (fabs(r[0]) < eps) {                        // :110-114
        for (int i = lane; i < aLen; i += WAVE) a[i] = (i == 0) ? 1.0 : 0.0;
        return true;
    }
    double rr[P + 1], aa[P + 1];
#pragma unroll
    for (int k = 0; k <= P; k++) { rr[k] = (k <= order) ? r[k] : 0.0; aa[k] = 0.0; }
    const double r0 = rr[0];
    aa[0] = 1.0;
    aa[1] = rr[1] / r0;
#pragma unroll
    for (int p = 2; p <= P; p++) {
        if (p <= order) {                          // wave-uniform
            double rho_a = 0.0, r_a = 0.0;
#pragma unroll
            for (int i = 1; i < p; i++) {          // :120-128
                rho_a += rr[p - i] * aa[i];
                r_a += rr[i] * aa[i];
            }
            const double k = (rr[p] - rho_a) / (r0 - r_a);
#pragma unroll
            for (int i = 1; 2 * i <= p; i++) {     // a[i] = aPrev[i] - k aPrev[p-i], both ends of the pair
                const double ai = aa[i], aj = aa[p - i];
                aa[i] = ai - k * aj;
                if (2 * i != p) aa[p - i] = aj - k * ai;
            }
            aa[p] = k;
        }
    }
    if (lane == 0) a[0] = 1.0;
#pragma unroll
    for (int k = 1; k <= P; k++)
        if (k <= order && lane == (k & 63)) a[k] = aa[k] * -1.;             // :145-146
    return false;
}

// The same register-resident recursion with one WINDOW per lane (r, a: the lane's own vectors; the order is wave-uniform):
// the workgroup vocoder's round of up to eight windows on ONE wavefront instead of eight wavefronts that each run it 64 times
// over (they share four SIMDs and an LDS pipe: 27 us per round at order 40 for the wave-distributed form).  A lane whose
// r[0] is below eps writes the unit vector (LPC.cpp:110-114) and steps out.
template <int P>
__device__ __noinline__ void levinson_lanes(const lds_f64 *r, lds_f64 *a, int order_, int aLen, double eps)
{
    const int order = __builtin_amdgcn_readfirstlane(order_);
    double rr[P + 1], aa[P + 1];
#pragma unroll
    for (int k = 0; k <= P; k++) { rr[k] = (k <= order) ? r[k] : 0.0; aa[k] = 0.0; }
    const double r0 = rr[0];
    if (fabs(r0) < eps) {
        for (int i = 0; i < aLen; i++) a[i] = (i == 0) ? 1.0 : 0.0;
        return;
    }
    aa[0] = 1.0;
    aa[1] = rr[1] / r0;
#pragma unroll
    for (int p = 2; p <= P; p++) {
        if (p <= order) {                          // wave-uniform
            double rho_a = 0.0, r_a = 0.0;
#pragma unroll
            for (int i = 1; i < p; i++) {          // :120-128
                rho_a += rr[p - i] * aa[i];
                r_a += rr[i] * aa[i];
            }
            const double k = (rr[p] - rho_a) / (r0 - r_a);
#pragma unroll
            for (int i = 1; 2 * i <= p; i++) {     // a[i] = aPrev[i] - k aPrev[p-i], both ends of the pair
                const double ai = aa[i], aj = aa[p - i];
                aa[i] = ai - k * aj;
                if (2 * i != p) aa[p - i] = aj - k * ai;
            }
            aa[p] = k;
        }
    }
    a[0] = 1.0;
#pragma unroll
    for (int k = 1; k <= P; k++)
        if (k <= order) a[k] = aa[k] * -1.;                                  // :145-146
}

// Left-to-right sums of e[i]^2 for two arrays at once (VocoderProcess.cpp:250), every lane of the
// calling wavefront redundantly: eight entries are read ahead per trip so that only the two
// (interleaved) chains of dependent adds remain.
template <class EP>
__device__ __forceinline__ void energy_pair_wave(EP e0, EP e1, int n, double &E0, double &E1)
{
    double r0 = 0.0, r1 = 0.0;
    if ((n & 15) == 0) {
        // sixteen entries per trip: ONE read per array (lane l holds entry i + (l & 15), every 16-lane
        // row the same), the squares computed once, lane-parallel, and the two ordered sums taken as
        // chains of v_fmac_f64 with a DPP row-broadcast operand (x * 1.0 + run rounds like run + x).
        // Per element this is two VALU ops and an eighth of an LDS read instead of two loads, two
        // multiplies and two adds done redundantly by every lane.
        const int l16 = threadIdx.x & 15;
        const double one = 1.0;
        double v0 = e0[l16], v1 = e1[l16];
        for (int i = 0; i < n; i += 16) {
            double s0 = v0 * v0, s1 = v1 * v1;
            if (i + 16 < n) { v0 = e0[i + 16 + l16]; v1 = e1[i + 16 + l16]; }
            asm volatile("s_nop 1" : "+v"(s0), "+v"(s1));                      // VALU write -> DPP read
#define VP_EN(U) VP_FMAC_BCAST(r0, s0, one, U); VP_FMAC_BCAST(r1, s1, one, U);
            VP_EN(0) VP_EN(1) VP_EN(2) VP_EN(3) VP_EN(4) VP_EN(5) VP_EN(6) VP_EN(7)
            VP_EN(8) VP_EN(9) VP_EN(10) VP_EN(11) VP_EN(12) VP_EN(13) VP_EN(14) VP_EN(15)
#undef VP_EN
        }
        E0 = r0; E1 = r1;
        return;
    }
    const int n8 = n & ~7;
    for (int i = 0; i < n8; i += 8) {
        double a[8], b[8];
#pragma unroll
        for (int u = 0; u < 8; u++) { a[u] = e0[i + u]; b[u] = e1[i + u]; }
#pragma unroll
        for (int u = 0; u < 8; u++) { a[u] = a[u] * a[u]; b[u] = b[u] * b[u]; }
#pragma unroll
        for (int u = 0; u < 8; u++) { r0 += a[u]; r1 += b[u]; }
    }
    for (int i = n8; i < n; i++) { double a = e0[i], b = e1[i]; r0 += a * a; r1 += b * b; }
    E0 = r0; E1 = r1;
}

// One array's left-to-right sum of e[i]^2, same arithmetic as a half of energy_pair_wave (for callers that give the two
// sums to two wavefronts).
template <class EP>
__device__ __forceinline__ double energy_wave(EP e0, int n)
{
    double r0 = 0.0;
    if ((n & 15) == 0) {
        const int l16 = threadIdx.x & 15;
        const double one = 1.0;
        double v0 = e0[l16];
        for (int i = 0; i < n; i += 16) {
            double s0 = v0 * v0;
            if (i + 16 < n) v0 = e0[i + 16 + l16];
            asm volatile("s_nop 1" : "+v"(s0));                                // VALU write -> DPP read
#define VP_EN(U) VP_FMAC_BCAST(r0, s0, one, U);
            VP_EN(0) VP_EN(1) VP_EN(2) VP_EN(3) VP_EN(4) VP_EN(5) VP_EN(6) VP_EN(7)
            VP_EN(8) VP_EN(9) VP_EN(10) VP_EN(11) VP_EN(12) VP_EN(13) VP_EN(14) VP_EN(15)
#undef VP_EN
        }
        return r0;
    }
    const int n8 = n & ~7;
    for (int i = 0; i < n8; i += 8) {
        double a[8];
#pragma unroll
        for (int u = 0; u < 8; u++) a[u] = e0[i + u];
#pragma unroll
        for (int u = 0; u < 8; u++) a[u] = a[u] * a[u];
#pragma unroll
        for (int u = 0; u < 8; u++) r0 += a[u];
    }
    for (int i = n8; i < n; i++) { double a = e0[i]; r0 += a * a; }
    return r0;
}

// VocoderProcess::filterFIR (VocoderProcess.cpp:235-251) for one window by one wavefront:
// e[i] = a[0]*xw[i] + sum_{k=1..min(order,i)} xw[i-k]*a[k].
template <class XP, class AP, class EP>
__device__ __forceinline__ void fir_window8(XP xw, AP a, int order, int W, EP e, int lane, int unit0 = 0, int unitStep = 1)
{
    // units of 8 x 64 outputs; a caller that shares the window with other wavefronts takes units unit0, unit0 + unitStep, ...
    for (int i0 = unit0 * 8 * WAVE; i0 < W; i0 += unitStep * 8 * WAVE) {
        double acc[8];
        int idx[8];
#pragma unroll
        for (int u = 0; u < 8; u++) {
            idx[u] = i0 + lane + u * WAVE;
            acc[u] = (idx[u] < W) ? a[0] * xw[idx[u]] : 0.0;             // a[0]*x*w with a[0] == 1
        }
        for (int k = 1; k <= order; k++) {
            const double ak = a[k];
#pragma unroll
            for (int u = 0; u < 8; u++) {
                const int j = idx[u] - k;
                const double xv_ = (j >= 0 && idx[u] < W) ? xw[j] : 0.0;
                acc[u] += xv_ * ak;
            }
        }
#pragma unroll
        for (int u = 0; u < 8; u++) if (idx[u] < W) e[idx[u]] = acc[u];
    }
}

// v_mul_f64 / v_add_f64 with a pinned program order (asm volatile statements keep their relative
// order).  Measured on gfx950 (tools/ubench_chain2.hip): a dependent fp64 op completes in ~8 cycles,
// a wave can issue one every ~4, so a chain step costs 8 cycles provided the NEXT products are
// already in flight; hipcc's own schedule put each product right in front of its subtract
// (16 cycles per tap).  Plain VALU ops on VGPR operands: hardware interlocks cover the dependencies.
#define VP_MUL64(D, A, B) asm volatile("v_mul_f64 %0, %1, %2" : "=v"(D) : "v"(A), "v"(B))
#define VP_SUB64(ACC, T) asm volatile("v_add_f64 %0, %0, -%1" : "+v"(ACC) : "v"(T))
#define VP_COPY64(D, S) asm("v_mul_f64 %0, %1, 1.0" : "=v"(D) : "v"(S))

template <int P, class XP, class YP, class AP, class HP>
__device__ __forceinline__ void iir_exact_lane(XP x, YP y, int n, AP aL, int order, HP hist, double gmul)
{
    static_assert(P % 4 == 0 && P >= 4, "P multiple of 4");
    double a[P + 1], h[P + 4];
#pragma unroll
    for (int k = 1; k <= P; k++) a[k] = (k <= order) ? aL[k] : 0.0;
#pragma unroll
    for (int j = 0; j < P; j++) h[j] = (hist && j < order) ? hist[j] : 0.0;
    for (int i = 0; i < n; i += 4) {
        double yn[4];
        const double xin[4] = {x[i], x[i + 1], x[i + 2], x[i + 3]};
#pragma unroll
        for (int s2 = 0; s2 < 4; s2++) {
            // tap k multiplies y[i+s2-k]: one of this trip's fresh outputs (k <= s2) or history
#define VP_HV(K) (((K) <= s2) ? yn[s2 - (K) < 0 ? 0 : s2 - (K)] : h[(K) - 1 - s2])
            double acc = gmul * xin[s2];
            double t[3];
            VP_MUL64(t[2 % 3], VP_HV(2), a[2]);           // independent of the sample just finished
            VP_MUL64(t[3 % 3], VP_HV(3), a[3]);
            VP_MUL64(t[1 % 3], VP_HV(1), a[1]);           // needs y[i+s2-1]: the one unavoidable wait
#pragma unroll
            for (int k = 1; k <= P; k++) {
                VP_SUB64(acc, t[k % 3]);
                if (k + 3 <= P) VP_MUL64(t[k % 3], VP_HV(k + 3), a[k + 3]);   // refill the slot just consumed
            }
            yn[s2] = acc;
#undef VP_HV
        }
        // history shift.  A plain double copy compiles to v_mov_b64, measured at ~12 ns per wave
        // instruction on gfx950 (tools/ubench_chain.hip) against 1.9 ns for an fp64 multiply, so
        // the copies are written as exact multiplications by 1.0.
#pragma unroll
        for (int j = P - 1; j >= 4; j--) VP_COPY64(h[j], h[j - 4]);
        VP_COPY64(h[3], yn[0]); VP_COPY64(h[2], yn[1]); VP_COPY64(h[1], yn[2]); VP_COPY64(h[0], yn[3]);
        y[i] = yn[0]; y[i + 1] = yn[1]; y[i + 2] = yn[2]; y[i + 3] = yn[3];
    }
}

// Generic (any order, any n) form of the same recursion with the history read back from y[]:
// y must be preceded by its own past (y[-k] valid for k <= min(order, i0 + i)).
template <class XP, class YP, class AP>
__device__ __forceinline__ void iir_exact_generic(XP x, YP y, int n, AP aL, int order, int i0, double gmul)
{
    for (int i = 0; i < n; i++) {
        double acc = gmul * x[i];
        const int kmax = min(order, i0 + i);
        for (int k = 1; k <= kmax; k++) acc -= y[i - k] * aL[k];
        y[i] = acc;
    }
}

// Dispatch on the (wave-uniform) order.  hist as in iir_exact_lane; i0 = number of valid past
// outputs before y[0] (only used by the generic path).
// LITE = the register-light kernel variant (two workgroups per CU): only the small register-resident
// instantiations, larger orders take the generic LDS form.
template <bool LITE = false, class XP, class YP, class AP, class HP>
__device__ __forceinline__ void iir_exact(XP x, YP y, int n, AP aL, int order_, HP hist, int i0, double gmul)
{
    const int order = __builtin_amdgcn_readfirstlane(order_);
    const bool q = (__builtin_amdgcn_readfirstlane(n) & 3) == 0;
    if (q && order <= 8) iir_exact_lane<8>(x, y, n, aL, order, hist, gmul);
    else if (q && order <= 16) iir_exact_lane<16>(x, y, n, aL, order, hist, gmul);
    else if (LITE) iir_exact_generic(x, y, n, aL, order, i0, gmul);
    else if (q && order <= 24) iir_exact_lane<24>(x, y, n, aL, order, hist, gmul);
    else if (q && order <= 32) iir_exact_lane<32>(x, y, n, aL, order, hist, gmul);
    else if (q && order <= 40) iir_exact_lane<40>(x, y, n, aL, order, hist, gmul);
    else if (q && order <= 48) iir_exact_lane<48>(x, y, n, aL, order, hist, gmul);
    else iir_exact_generic(x, y, n, aL, order, i0, gmul);
}

// lane i <- lane i+1 of the whole wavefront (DPP wave_shl:1, a plain VALU move: no LDS round trip
// like ds_bpermute); lane 63 receives 0.
__device__ __forceinline__ double wave_shl1_f64(double v)
{
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_update_dpp(0, lo, 0x130, 0xf, 0xf, false);
    hi = __builtin_amdgcn_update_dpp(0, hi, 0x130, 0xf, 0xf, false);
    return __hiloint2double(hi, lo);
}

// FAST (not bit-exact) form of the same all-pole recursion, selected with vp_set_iir_mode(h, 1):
// transposed direct form II with the state vector spread over the lanes (lane j holds s_{j+1},
// and s_{j+65} for orders above 64):   y = g*x + s_1 ;  s_k = s_{k+1} - a_k*y  (one fma per lane).
// The critical path per sample is readlane -> add -> fma whatever the order, against (order+1)
// dependent operations for the exact chain.  The taps are summed oldest-first instead of
// newest-first, so the result differs from the reference's by rounding only (~1e-16 relative per
// operation); no decision of the algorithm depends on an IIR output.  hist[j] = y[-1-j] or nullptr.
template <bool TWO, class XP, class YP, class AP, class HP>
__device__ __forceinline__ void iir_fast_wave_impl(XP x, YP y, int n, AP aL, int order, HP hist, double gmul)
{
    const int lane = threadIdx.x & 63;
    const int k0 = lane + 1, k1 = lane + 65;
    const double a0 = (k0 <= order) ? aL[k0] : 0.0;
    const double a1 = (TWO && k1 <= order) ? aL[k1] : 0.0;
    double s0 = 0.0, s1 = 0.0;
    if (hist) {                                   // state equivalent to the given output history
        for (int m = 0; m < order; m++) {
            const double hm = hist[m];
            if (k0 + m <= order) s0 = __builtin_fma(-aL[k0 + m], hm, s0);
            if (TWO && k1 + m <= order) s1 = __builtin_fma(-aL[k1 + m], hm, s1);
        }
    }
    auto step = [&](double xi) -> double {
        const double yy = __builtin_fma(gmul, xi, bcast_f64(s0, 0));
        double n0 = wave_shl1_f64(s0), n1 = 0.0;
        if (TWO) {
            const double c = bcast_f64(s1, 0);
            n1 = wave_shl1_f64(s1);
            if (lane == 63) n0 = c;
        }
        s0 = __builtin_fma(-a0, yy, n0);
        if (TWO) s1 = __builtin_fma(-a1, yy, n1);
        return yy;
    };
    const int n8 = n & ~7;
    for (int i = 0; i < n8; i += 8) {
        double xv[8], yv[8];
#pragma unroll
        for (int u = 0; u < 8; u++) xv[u] = x[i + u];
#pragma unroll
        for (int u = 0; u < 8; u++) yv[u] = step(xv[u]);
#pragma unroll
        for (int u = 0; u < 8; u++) y[i + u] = yv[u];           // every lane stores the same value
    }
    for (int i = n8; i < n; i++) y[i] = step(x[i]);
}
template <class XP, class YP, class AP, class HP>
__device__ __forceinline__ void iir_fast_wave(XP x, YP y, int n, AP aL, int order_, HP hist, double gmul)
{
    const int order = __builtin_amdgcn_readfirstlane(order_);
    if (order > 64) iir_fast_wave_impl<true>(x, y, n, aL, order, hist, gmul);
    else iir_fast_wave_impl<false>(x, y, n, aL, order, hist, gmul);
}

// FAST mode, BLOCK form of the all-pole recursion (used when n is a multiple of 64): with h the
// impulse response of 1/A(z) (64 samples, computed once per coefficient set with the wave recursion
// above), a block of 64 outputs is   y = T(h) (g x + u),   u_i = -sum_{m>i} a_m y[i-m]  (i < order)
// carrying the previous outputs in, T(h) lower-triangular Toeplitz.  One lane per output sample:
// no dependence between the samples of a block, ~0.4 us per block instead of 64 x 33 ns.
// hpad: 128 doubles (64 zeros, then h), xp: 64 doubles of scratch.  nh0 = valid outputs before y[0].
template <class XP, class YP, class AP>
__device__ __forceinline__ void iir_block_wave(XP x, YP y, int n, AP aL, int order_, int nh0, const lds_f64 *hpad, lds_f64 *xp,
                                               double gmul)
{
    const int lane = threadIdx.x & 63;
    const int order = __builtin_amdgcn_readfirstlane(order_);
    for (int b = 0; b < n; b += WAVE) {
        STAMPG_BEGIN();
        // u_lane = -sum_{k=1..order} a[lane+k] * y[b-k]   (a beyond the order = 0; y before the start = 0):
        // the history values are wave-uniform reads, the trip count is uniform, two accumulators
        const int kmax = (b == 0) ? min(order, nh0) : order;
        double u0 = 0.0, u1 = 0.0;
        int k = 1;
        // eight taps per trip, their sixteen LDS reads issued together (the two-tap loop this replaces waited out an LDS
        // round trip per pair: 1.8 us per 64-sample block at order 48, 79 of the IIR's 92 us per block at the configs[4] geometry)
        for (; k + 7 <= kmax; k += 8) {
            double yv[8], av[8];
#pragma unroll
            for (int t = 0; t < 8; t++) { yv[t] = y[b - k - t]; av[t] = (lane + k + t <= order) ? aL[lane + k + t] : 0.0; }
#pragma unroll
            for (int t = 0; t < 8; t += 2) { u0 = __builtin_fma(-av[t], yv[t], u0); u1 = __builtin_fma(-av[t + 1], yv[t + 1], u1); }
        }
        for (; k + 1 <= kmax; k += 2) {
            const double y1 = y[b - k], y2 = y[b - k - 1];
            const double a1 = (lane + k <= order) ? aL[lane + k] : 0.0;
            const double a2 = (lane + k + 1 <= order) ? aL[lane + k + 1] : 0.0;
            u0 = __builtin_fma(-a1, y1, u0);
            u1 = __builtin_fma(-a2, y2, u1);
        }
        if (k <= kmax) { const double a1 = (lane + k <= order) ? aL[lane + k] : 0.0; u0 = __builtin_fma(-a1, y[b - k], u0); }
        const double u = u0 + u1;
        xp[lane] = __builtin_fma(gmul, x[b + lane], u);
        STAMPG(24);
        // 64-term dot product with four independent accumulators, eight terms read ahead per trip
        double acc0 = 0.0, acc1 = 0.0, acc2 = 0.0, acc3 = 0.0;
        const lds_f64 *hh = hpad + WAVE + lane;                             // hh[-j] = h[lane - j] (0 for j > lane)
        double hv[8], xv[8], hn[8], xn[8];
#pragma unroll
        for (int q = 0; q < 8; q++) { hv[q] = hh[-q]; xv[q] = xp[q]; }
#pragma unroll
        for (int j = 0; j < WAVE; j += 8) {
            if (j + 8 < WAVE) {
#pragma unroll
                for (int q = 0; q < 8; q++) { hn[q] = hh[-(j + 8 + q)]; xn[q] = xp[j + 8 + q]; }
            }
            acc0 = __builtin_fma(hv[0], xv[0], acc0); acc1 = __builtin_fma(hv[1], xv[1], acc1);
            acc2 = __builtin_fma(hv[2], xv[2], acc2); acc3 = __builtin_fma(hv[3], xv[3], acc3);
            acc0 = __builtin_fma(hv[4], xv[4], acc0); acc1 = __builtin_fma(hv[5], xv[5], acc1);
            acc2 = __builtin_fma(hv[6], xv[6], acc2); acc3 = __builtin_fma(hv[7], xv[7], acc3);
#pragma unroll
            for (int q = 0; q < 8; q++) { hv[q] = hn[q]; xv[q] = xn[q]; }
        }
        const double acc = (acc0 + acc1) + (acc2 + acc3);
        y[b + lane] = acc;
        STAMPG(25);
    }
}

// The same block form with the dot product kept off the LDS (orders <= 16, whole 64-sample blocks,
// called by ONE full wavefront).  A DS instruction costs its wave 4-8 ns of issue even when nothing
// waits for it (tools/ubench_lds.hip), and the LDS form above spends two of them per term.  Here
//   * the lane's 64 taps H[j] = h[lane - j] stay in registers for the whole chunk (128 VGPRs),
//   * the block's input is held as four registers X_q[lane] = x[b + 16 q + (lane & 15)] (every
//     16-lane row holds the same 16 samples), so that term i = 16 q + m is ONE instruction,
//     v_fmac_f64_dpp acc, X_q row_newbcast:m, H[i], with no broadcast traffic at all,
//   * the carry-in of the previous block only reaches samples 0..order-1 <= 15, i.e. X_0: each row
//     computes it for its own copy, and the 48 terms of X_1..X_3 are issued while the history
//     values (the previous block's last outputs, written a moment ago) come back from the LDS.
// Rounding differs from the LDS form only in the order of the partial sums (both are FAST mode).
#define VP_BI_T(ACC, XQ, HI, M) \
    asm volatile("v_fmac_f64_dpp %0, %1, %2 row_newbcast:" #M " row_mask:0xf bank_mask:0xf" : "+v"(ACC) : "v"(XQ), "v"(HI))
#define VP_BI_ROW(XQ, Q) \
    VP_BI_T(acc0, XQ, H[16 * Q + 0], 0);   VP_BI_T(acc1, XQ, H[16 * Q + 1], 1);   VP_BI_T(acc2, XQ, H[16 * Q + 2], 2);   \
    VP_BI_T(acc3, XQ, H[16 * Q + 3], 3);   VP_BI_T(acc0, XQ, H[16 * Q + 4], 4);   VP_BI_T(acc1, XQ, H[16 * Q + 5], 5);   \
    VP_BI_T(acc2, XQ, H[16 * Q + 6], 6);   VP_BI_T(acc3, XQ, H[16 * Q + 7], 7);   VP_BI_T(acc0, XQ, H[16 * Q + 8], 8);   \
    VP_BI_T(acc1, XQ, H[16 * Q + 9], 9);   VP_BI_T(acc2, XQ, H[16 * Q + 10], 10); VP_BI_T(acc3, XQ, H[16 * Q + 11], 11); \
    VP_BI_T(acc0, XQ, H[16 * Q + 12], 12); VP_BI_T(acc1, XQ, H[16 * Q + 13], 13); VP_BI_T(acc2, XQ, H[16 * Q + 14], 14); \
    VP_BI_T(acc3, XQ, H[16 * Q + 15], 15);
__device__ __forceinline__ void iir_block_wave_regs(const lds_f64 *x, lds_f64 *y, int n, const lds_f64 *aL, int order_,
                                                    bool haveHist0, const lds_f64 *hpad)
{
    const int lane = threadIdx.x & 63, m = lane & 15;
    const int order = __builtin_amdgcn_readfirstlane(order_);
    double H[64], A[16];
#pragma unroll
    for (int j = 0; j < 64; j++) H[j] = hpad[WAVE + lane - j];                // h[lane - j], 0 for j > lane
#pragma unroll
    for (int k = 1; k <= 16; k++) A[k - 1] = (m + k <= order) ? -aL[m + k] : 0.0;  // -a[m + k]
    double X1 = x[16 + m], X2 = x[32 + m], X3 = x[48 + m], X0 = x[m];
    for (int b = 0; b < n; b += WAVE) {
        // history: Y3[lane] = y[b - 16 + (lane & 15)] (zero before the frame's first sample), so that
        // y[b - k] is lane 16 - k of the row
        const bool hist = (b > 0) || haveHist0;
        double Y3 = 0.0;
        // (only the last `order` outputs are history: older ones are not even restored when the frame continues in a
        // later launch, and a zero coefficient does not neutralise a NaN left in LDS by somebody else)
        if (hist && 16 - m <= order) Y3 = y[b - 16 + m];
        double acc0 = 0.0, acc1 = 0.0, acc2 = 0.0, acc3 = 0.0;
        VP_BI_ROW(X1, 1)
        VP_BI_ROW(X2, 2)
        VP_BI_ROW(X3, 3)
        double u0 = 0.0, u1 = 0.0;
        VP_BI_T(u0, Y3, A[0], 15);  VP_BI_T(u1, Y3, A[1], 14);  VP_BI_T(u0, Y3, A[2], 13);  VP_BI_T(u1, Y3, A[3], 12);
        VP_BI_T(u0, Y3, A[4], 11);  VP_BI_T(u1, Y3, A[5], 10);  VP_BI_T(u0, Y3, A[6], 9);   VP_BI_T(u1, Y3, A[7], 8);
        VP_BI_T(u0, Y3, A[8], 7);   VP_BI_T(u1, Y3, A[9], 6);   VP_BI_T(u0, Y3, A[10], 5);  VP_BI_T(u1, Y3, A[11], 4);
        VP_BI_T(u0, Y3, A[12], 3);  VP_BI_T(u1, Y3, A[13], 2);  VP_BI_T(u0, Y3, A[14], 1);  VP_BI_T(u1, Y3, A[15], 0);
        X0 = X0 + (u0 + u1);
        asm volatile("s_nop 1" : "+v"(X0));                                   // VALU write of X0 -> DPP read: 2 wait states
        VP_BI_ROW(X0, 0)
        y[b + lane] = (acc0 + acc1) + (acc2 + acc3);
        const int bn = b + WAVE;
        if (bn < n) { X0 = x[bn + m]; X1 = x[bn + 16 + m]; X2 = x[bn + 32 + m]; X3 = x[bn + 48 + m]; }
    }
}

// The block form for orders 17..48 (whole 64-sample blocks, ONE full wavefront), with the recursion between the blocks
// reduced to what really is serial.  By linearity a block's outputs are
//     y[b + i] = z[b + i] + sum_{k=1..order} Hc[i][k] * y[b - k],        z = T(h) (g x)   (zero-state response),
// where column k of Hc is the block's response to a unit in history slot k.  Phase 1 computes z for ALL blocks of the chunk
// (no dependence between them: 64 DPP terms per block, taps in registers as above).  Phase 2 walks the blocks with the
// lane's row Hc[lane][1..48] in registers and the last 48 outputs in three row-broadcast registers: `order` DPP terms and
// one LDS round trip per 64 samples (the LDS form above: 112 terms with two LDS reads each, 2.3 us per block at order 48).
// Hc comes from the impulse response without any sum over taps:  Hc[i][1] = h[i+1],  Hc[i][k+1] = Hc[i+1][k] + a[k] h[i+1]
// (shift the carry-in sequence u_k[j] = -a[j+k] left by one), i.e. 47 steps of "lane i takes lane i+1's value" on h[0..111]:
// hpad holds 64 zeros, then h[0..127].  nh0 = valid outputs before y[0].  Rounding differs from the other FAST forms only
// in how the same sums are grouped.
__device__ __forceinline__ void iir_block_wave_hc(const lds_f64 *x, lds_f64 *y, int n, const lds_f64 *aL, int order_, int nh0,
                                                  const lds_f64 *hpad, double gmul)
{
    const int lane = threadIdx.x & 63, m = lane & 15;
    const int order = __builtin_amdgcn_readfirstlane(order_);
    {
        double H[64];
#pragma unroll
        for (int j = 0; j < 64; j++) H[j] = hpad[WAVE + lane - j];            // h[lane - j], 0 for j > lane
        for (int b = 0; b < n; b += WAVE) {
            double X0 = gmul * x[b + m], X1 = gmul * x[b + 16 + m], X2 = gmul * x[b + 32 + m], X3 = gmul * x[b + 48 + m];
            double acc0 = 0.0, acc1 = 0.0, acc2 = 0.0, acc3 = 0.0;
            asm volatile("s_nop 1" : "+v"(X0), "+v"(X1), "+v"(X2), "+v"(X3));     // VALU write -> DPP read
            VP_BI_ROW(X0, 0)
            VP_BI_ROW(X1, 1)
            VP_BI_ROW(X2, 2)
            VP_BI_ROW(X3, 3)
            y[b + lane] = (acc0 + acc1) + (acc2 + acc3);
        }
    }
    double R[48];
    {
        const lds_f64 *h = hpad + WAVE;
        const double h0 = h[lane + 1], h1 = (lane + 65 < 128) ? h[lane + 65] : 0.0;   // h[i + 1] for i = lane, lane + 64
        double c0 = h0, c1 = h1;
        R[0] = c0;
#pragma unroll
        for (int k = 1; k < 48; k++) {
            const double ak = (k <= order) ? aL[k] : 0.0;
            const double up = bcast_f64(c1, 0);
            double n0 = wave_shl1_f64(c0);
            const double n1 = wave_shl1_f64(c1);
            if (lane == 63) n0 = up;
            c0 = __builtin_fma(ak, h0, n0);
            c1 = __builtin_fma(ak, h1, n1);
            R[k] = c0;
        }
    }
    for (int b = 0; b < n; b += WAVE) {
        // Yq[lane] = y[b - 16 (4 - q) + (lane & 15)]: history slot k = 16 (4 - q) - m sits in lane 16 (4 - q) - k of the row.
        // Slots beyond the order, or before the frame's first sample, read as zero (never from LDS: a zero coefficient does
        // not neutralise a NaN somebody else left there).
        const int kHave = (b == 0) ? min(order, nh0) : order;
        double Y3 = 0.0, Y2 = 0.0, Y1 = 0.0;
        if (16 - m <= kHave) Y3 = y[b - 16 + m];
        if (32 - m <= kHave) Y2 = y[b - 32 + m];
        if (48 - m <= kHave) Y1 = y[b - 48 + m];
        const double z = y[b + lane];
        double acc0 = 0.0, acc1 = 0.0, acc2 = 0.0, acc3 = 0.0;
        asm volatile("s_nop 1" : "+v"(Y1), "+v"(Y2), "+v"(Y3));
#define VP_HC_ROW(YQ, K0) \
        VP_BI_T(acc0, YQ, R[K0 + 0], 15);  VP_BI_T(acc1, YQ, R[K0 + 1], 14);  VP_BI_T(acc2, YQ, R[K0 + 2], 13);  VP_BI_T(acc3, YQ, R[K0 + 3], 12);  \
        VP_BI_T(acc0, YQ, R[K0 + 4], 11);  VP_BI_T(acc1, YQ, R[K0 + 5], 10);  VP_BI_T(acc2, YQ, R[K0 + 6], 9);   VP_BI_T(acc3, YQ, R[K0 + 7], 8);   \
        VP_BI_T(acc0, YQ, R[K0 + 8], 7);   VP_BI_T(acc1, YQ, R[K0 + 9], 6);   VP_BI_T(acc2, YQ, R[K0 + 10], 5);  VP_BI_T(acc3, YQ, R[K0 + 11], 4);  \
        VP_BI_T(acc0, YQ, R[K0 + 12], 3);  VP_BI_T(acc1, YQ, R[K0 + 13], 2);  VP_BI_T(acc2, YQ, R[K0 + 14], 1);  VP_BI_T(acc3, YQ, R[K0 + 15], 0);
        VP_HC_ROW(Y3, 0)
        if (order > 16) { VP_HC_ROW(Y2, 16) }
        if (order > 32) { VP_HC_ROW(Y1, 32) }
#undef VP_HC_ROW
        y[b + lane] = z + ((acc0 + acc1) + (acc2 + acc3));
    }
}

// The same decomposition for orders up to 16 in the register-light builds (two workgroups per CU, 128 VGPRs: the 64 taps of
// iir_block_wave_regs do not fit, and the LDS form spends 35 us per 1024 samples on its reads): the zero-state responses in
// four passes of 16 taps each, the history matrix row (16 entries) summed directly from the 64-sample impulse response --
// Hc[i][k] = -sum_{j=0..16-k} h[i-j] a[j+k], the coefficients through the row broadcast -- and 16 DPP terms per block in
// the serial pass.
template <int M>
__device__ __forceinline__ void fmac_row_bcast(double &acc, double rowv, double h)
{
    asm volatile("v_fmac_f64_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(rowv), "v"(h), "n"(M));
}
template <int K, int... J>
__device__ __forceinline__ double hc16_row(double arow, const double (&hl)[16], std::integer_sequence<int, J...>)
{
    double r0 = 0.0, r1 = 0.0;
    (fmac_row_bcast<J + K - 1>((J & 1) ? r1 : r0, arow, hl[J]), ...);         // -a[J + K] sits in lane J + K - 1 of the row
    return r0 + r1;
}
__device__ __forceinline__ void iir_block_wave_hc16(const lds_f64 *x, lds_f64 *y, int n, const lds_f64 *aL, int order_, int nh0,
                                                    const lds_f64 *hpad, double gmul)
{
    const int lane = threadIdx.x & 63, m = lane & 15;
    const int order = __builtin_amdgcn_readfirstlane(order_);
#pragma unroll 1
    for (int q = 0; q < 4; q++) {                                             // taps 16 q .. 16 q + 15 against x[b + 16 q + ...]
        double H[16];
#pragma unroll
        for (int j = 0; j < 16; j++) H[j] = hpad[WAVE + lane - 16 * q - j];   // h[lane - 16 q - j], 0 left of the start
        for (int b = 0; b < n; b += WAVE) {
            double X0 = gmul * x[b + 16 * q + m];
            double acc0 = 0.0, acc1 = 0.0, acc2 = 0.0, acc3 = 0.0;
            asm volatile("s_nop 1" : "+v"(X0));
            VP_BI_ROW(X0, 0)
            const double sum = (acc0 + acc1) + (acc2 + acc3);
            if (q == 0) y[b + lane] = sum;
            else y[b + lane] += sum;
        }
    }
    double R[16];
    {
        double hl[16];
#pragma unroll
        for (int j = 0; j < 16; j++) hl[j] = hpad[WAVE + lane - j];
        double arow = (m + 1 <= order) ? -aL[m + 1] : 0.0;
        asm volatile("s_nop 1" : "+v"(arow));
#define VP_HC16_R(K) R[K - 1] = hc16_row<K>(arow, hl, std::make_integer_sequence<int, 17 - K>{});
        VP_HC16_R(1) VP_HC16_R(2) VP_HC16_R(3) VP_HC16_R(4) VP_HC16_R(5) VP_HC16_R(6) VP_HC16_R(7) VP_HC16_R(8)
        VP_HC16_R(9) VP_HC16_R(10) VP_HC16_R(11) VP_HC16_R(12) VP_HC16_R(13) VP_HC16_R(14) VP_HC16_R(15) VP_HC16_R(16)
#undef VP_HC16_R
    }
    for (int b = 0; b < n; b += WAVE) {
        const int kHave = (b == 0) ? min(order, nh0) : order;
        double Y3 = 0.0;
        if (16 - m <= kHave) Y3 = y[b - 16 + m];          // (slots beyond the order or before the frame's start: zero, never from LDS)
        const double z = y[b + lane];
        double acc0 = 0.0, acc1 = 0.0, acc2 = 0.0, acc3 = 0.0;
        asm volatile("s_nop 1" : "+v"(Y3));
        VP_BI_T(acc0, Y3, R[0], 15);  VP_BI_T(acc1, Y3, R[1], 14);  VP_BI_T(acc2, Y3, R[2], 13);  VP_BI_T(acc3, Y3, R[3], 12);
        VP_BI_T(acc0, Y3, R[4], 11);  VP_BI_T(acc1, Y3, R[5], 10);  VP_BI_T(acc2, Y3, R[6], 9);   VP_BI_T(acc3, Y3, R[7], 8);
        VP_BI_T(acc0, Y3, R[8], 7);   VP_BI_T(acc1, Y3, R[9], 6);   VP_BI_T(acc2, Y3, R[10], 5);  VP_BI_T(acc3, Y3, R[11], 4);
        VP_BI_T(acc0, Y3, R[12], 3);  VP_BI_T(acc1, Y3, R[13], 2);  VP_BI_T(acc2, Y3, R[14], 1);  VP_BI_T(acc3, Y3, R[15], 0);
        y[b + lane] = z + ((acc0 + acc1) + (acc2 + acc3));
    }
}
#undef VP_BI_ROW

// ------------------------------------------------------------------------------------------------
// K1: vocoder.  VocoderProcess::process/processWindow (VocoderProcess.cpp:173-223), one workgroup
// per stream, one wavefront per window, windows of a block taken in rounds of (waves per group).
//
// Per-wave LDS (W = window length):
//   A  [W] f64   first: voice f32[W] | synth f32[W] (raw samples for the autocorrelation);
//                later: eSynth (residual of the carrier)
//   B  [W] f64   first: xwV = voice*anWindow; later: out (IIR output, then scaled for the OLA)
//   Cc [W] f64   xwS = synth*anWindow
//   D  [W] f64   eVoice (only its energy is used)
//   r/a for voice (101 each) and synth (31 each)

// biaisedAutoCorr of the vocoder (LPC.cpp:44-97), TWO adjacent lags per lane (m and m + 1, m even): r[m] = sum_n
// ((xw[n] * x[n+m]) * w[n+m]), n ascending, each product rounded as the reference rounds it.  The one-lag form pays three LDS
// reads per element and lag, and the eight wavefronts of a round are bound by the LDS pipe (25 us per round at 512-sample
// windows); here the lane's raw samples and window values slide past xw[n] -- one new value of each serves two lags -- and
// the caller packs two windows into a wavefront, so the round issues a quarter of the LDS instructions.
// xw: the window's x * anWindow (f64), xf: its raw samples + m (f32), wm: anWindow + m; all 8-byte (xf) / 16-byte aligned.
// Requests run two trips ahead, unconditionally (up to 16 elements past the lane's count: inside the LDS allocation).
__device__ __forceinline__ void voc_autocorr2(const lds_f64 *xw, const lds_f32 *xf, const lds_f64 *wm, int cntA, double &sA, double &sB)
{
    typedef double d2 __attribute__((ext_vector_type(2)));
    typedef float f2 __attribute__((ext_vector_type(2)));
    typedef __attribute__((address_space(3))) d2 lds_d2;
    typedef __attribute__((address_space(3))) f2 lds_f2;
    const int cntB = cntA - 1, c8 = max(cntB, 0) & ~7;
    double w0 = wm[0], w1 = wm[1], f0 = (double)xf[0], f1 = (double)xf[1];
    d2 u0[4], q0[4], u1[4], q1[4];
    f2 x0[4], x1[4];
#define VP_V2LOAD(U, Q, X, I) _Pragma("unroll") for (int u = 0; u < 4; u++) { \
        U[u] = *(const lds_d2 *)(xw + (I) + 2 * u); Q[u] = *(const lds_d2 *)(wm + (I) + 2 + 2 * u); X[u] = *(const lds_f2 *)(xf + (I) + 2 + 2 * u); }
#define VP_V2COMP(U, Q, X) { \
        const double g0 = (double)X[0].x, g1 = (double)X[0].y, g2 = (double)X[1].x, g3 = (double)X[1].y, \
                     g4 = (double)X[2].x, g5 = (double)X[2].y, g6 = (double)X[3].x, g7 = (double)X[3].y; \
        double pa[8], pb[8]; \
        pa[0] = U[0].x * f0; pb[0] = U[0].x * f1; pa[1] = U[0].y * f1; pb[1] = U[0].y * g0; \
        pa[2] = U[1].x * g0; pb[2] = U[1].x * g1; pa[3] = U[1].y * g1; pb[3] = U[1].y * g2; \
        pa[4] = U[2].x * g2; pb[4] = U[2].x * g3; pa[5] = U[2].y * g3; pb[5] = U[2].y * g4; \
        pa[6] = U[3].x * g4; pb[6] = U[3].x * g5; pa[7] = U[3].y * g5; pb[7] = U[3].y * g6; \
        pa[0] *= w0;     pb[0] *= w1;     pa[1] *= w1;     pb[1] *= Q[0].x; \
        pa[2] *= Q[0].x; pb[2] *= Q[0].y; pa[3] *= Q[0].y; pb[3] *= Q[1].x; \
        pa[4] *= Q[1].x; pb[4] *= Q[1].y; pa[5] *= Q[1].y; pb[5] *= Q[2].x; \
        pa[6] *= Q[2].x; pb[6] *= Q[2].y; pa[7] *= Q[2].y; pb[7] *= Q[3].x; \
        __builtin_amdgcn_sched_barrier(0); \
        _Pragma("unroll") for (int u = 0; u < 8; u++) { sA += pa[u]; sB += pb[u]; } \
        __builtin_amdgcn_sched_barrier(0); \
        f0 = g6; f1 = g7; w0 = Q[3].x; w1 = Q[3].y; }
    VP_V2LOAD(u0, q0, x0, 0)
    VP_V2LOAD(u1, q1, x1, 8)
    int n = 0;
    for (; n + 16 <= c8; n += 16) {
        VP_V2COMP(u0, q0, x0)
        VP_V2LOAD(u0, q0, x0, n + 16)
        VP_V2COMP(u1, q1, x1)
        VP_V2LOAD(u1, q1, x1, n + 24)
    }
    if (n < c8) { VP_V2COMP(u0, q0, x0) }
#undef VP_V2LOAD
#undef VP_V2COMP
    for (int i = c8; i < cntA; i++) sA += (xw[i] * (double)xf[i]) * wm[i];
    for (int i = c8; i < cntB; i++) sB += (xw[i] * (double)xf[i + 1]) * wm[i + 1];
}

// E = sum e[i]^2 left to right (VocoderProcess.cpp:250) with one WINDOW per lane (e: the lane's own residual, 16-byte aligned):
// the round's eight sums on one wavefront instead of eight wavefronts that each chain 2 W dependent adds on shared SIMDs.
// Eight entries per trip, requested two trips ahead (unconditionally: up to 16 entries past n, inside the LDS allocation).
__device__ __forceinline__ double energy_lanes(const lds_f64 *e, int n)
{
    typedef double d2 __attribute__((ext_vector_type(2)));
    typedef __attribute__((address_space(3))) d2 lds_d2;
    double E = 0.0;
    const int n8 = n & ~7;
    d2 a0[4], a1[4];
#define VP_ELLOAD(A, I) _Pragma("unroll") for (int u = 0; u < 4; u++) A[u] = *(const lds_d2 *)(e + (I) + 2 * u);
#define VP_ELCOMP(A) { double s_[8]; _Pragma("unroll") for (int u = 0; u < 4; u++) { s_[2 * u] = A[u].x * A[u].x; s_[2 * u + 1] = A[u].y * A[u].y; } \
        __builtin_amdgcn_sched_barrier(0); \
        _Pragma("unroll") for (int u = 0; u < 8; u++) E += s_[u]; \
        __builtin_amdgcn_sched_barrier(0); }
    VP_ELLOAD(a0, 0)
    VP_ELLOAD(a1, 8)
    int i = 0;
    for (; i + 16 <= n8; i += 16) {
        VP_ELCOMP(a0)
        VP_ELLOAD(a0, i + 16)
        VP_ELCOMP(a1)
        VP_ELLOAD(a1, i + 24)
    }
    if (i < n8) { VP_ELCOMP(a0) }
#undef VP_ELLOAD
#undef VP_ELCOMP
    for (int k = n8; k < n; k++) E += e[k] * e[k];
    return E;
}

// LITE: the build for two workgroups per CU (<= 128 VGPRs, FAST IIR only -- the register-resident exact recursion is
// compiled out; the host never launches it in exact mode).
template <bool LITE>
__device__ __forceinline__ void vocoder_block(const VpGeom &g, const VpCall &c, const VpDev &d, double *smem)
{
    const int s = vp_stream(d), tid = threadIdx.x;
    const int waveHw = tid >> 6, lane = tid & 63, nWaves = c.vocWin;      // nWaves: windows per round
    // The launch carries nRoles wavefronts per window slot: wavefront waveHw works for window waveHw % vocWin in role
    // waveHw / vocWin.  Role 0 owns the window (everything ordered or serial); the others share its storage and take
    // their part of the lane-parallel phases (autocorrelation passes, residual-FIR units).
    const int role = waveHw / nWaves, wave = waveHw - role * nWaves;
    const int nRoles = (int)(blockDim.x >> 6) / nWaves;                   // the host launches a whole number of them
    const bool helper = role != 0;
    if (!(d.gate[s * 2 + 0] && d.gate[s * 2 + 1])) return;     // :199-204, whole workgroup

    const VpStreamParams sp = d.pitch[s].sp;              // this stream's treeState values (one uniform read)
    const int W = g.W, oV = sp.orderVoice, oS = sp.orderSynth;
    lds_f64 *sm = (lds_f64 *)smem;
    lds_f64 *win = sm;                        // [W] shared by all waves
    lds_f64 *hist = win + W;                  // [2][10] EeVoiceArr, EeSynthArr
    lds_f64 *roundE = hist + 20;              // [2][8]
    lds_f64 *gArr = roundE + 16;              // [8]
    lds_f64 *wbase = gArr + 8 + (size_t)wave * voc_wave_doubles(W);
    lds_f64 *A = wbase, *B = A + W, *Cc = B + W, *D = Cc + W;
    lds_f32 *xv = (lds_f32 *)A, *xsy = xv + W;
    lds_f64 *rV = D + W, *aV = rV + (VP_ORDER_MAX + 1);
    lds_f64 *rS = aV + (VP_ORDER_MAX + 1), *aS = rS + (VP_ORDER_MAX_SYNTH + 1);

    for (int i = tid; i < W; i += blockDim.x) win[i] = d.vocWin[i];
    if (tid < 20) hist[tid] = d.EeArr[(size_t)s * 20 + tid];
    const float *vr = d.voiceRing + (size_t)s * g.inSize;
    const float *sr0 = d.synthRing + (size_t)s * 2 * g.inSize;
    double *acc = d.outAcc + (size_t)s * g.outSize;
    STAMP0(d);
    __syncthreads();

    for (int w0 = 0; w0 < c.nWin; w0 += nWaves) {
        const int w = w0 + wave;
        const bool activeW = w < c.nWin && role < nRoles;      // the window of this wavefront (own or helped) is in the round
        const bool active = activeW && !helper;                // ... and this wavefront owns it
        const int nAct = min(nWaves, c.nWin - w0);
        const int start = c.vStart + w * g.h;

        if (active) {
            for (int i = lane; i < W; i += WAVE) {
                int p = ring_pos(c.currCounter, start + i, g.inSize);
                float v = vr[p], y = sr0[p];
                xv[i] = v; xsy[i] = y;
                B[i] = (double)v * win[i];               // x*anWindow, LPC.cpp:61 tmp / filterFIR product
                Cc[i] = (double)y * win[i];
            }
        }
        __syncthreads();
        STAMP(d, 16);

        // biaisedAutoCorr (LPC.cpp:44-97) for voice lags 0..oV and synth lags 0..oS: one lane per
        // lag, each lag its own left-to-right sum over n.
        // (two lags per lane, two windows per wavefront when all of a window's lag pairs fit 32 lanes: voc_autocorr2)
        const int nPV = (oV >> 1) + 1, nPS = (oS >> 1) + 1;
#ifdef VP_DIAG_NO_VOC_AC2
        const bool ac2 = false;
#else
        const bool ac2 = !LITE && nPV + nPS <= 32 && (W & 1) == 0 && W > oV + 2 && W > oS + 2;
#endif
        if (ac2) {
            const int half = lane >> 5, l = lane & 31;
            if (2 * waveHw < nAct) {
                const int wj0 = 2 * waveHw + half, wj = min(wj0, nAct - 1);
                lds_f64 *wb = gArr + 8 + (size_t)wj * voc_wave_doubles(W);
                const bool isV = l < nPV;
                const int m = 2 * (isV ? l : min(l - nPV, nPS - 1)), ord = isV ? oV : oS;
                double sA = 0.0, sB = 0.0;
                voc_autocorr2((const lds_f64 *)(wb + (isV ? 1 : 2) * (size_t)W), (const lds_f32 *)wb + (isV ? 0 : W) + m, (const lds_f64 *)win + m, W - m, sA, sB);
                sA /= (double)W;
                sB /= (double)W;
                lds_f64 *rdst = wb + 4 * (size_t)W + (isV ? 0 : 2 * (VP_ORDER_MAX + 1));
                if (wj0 < nAct && l < nPV + nPS) { rdst[m] = sA; if (m + 1 <= ord) rdst[m + 1] = sB; }
            }
        } else
        if (activeW) {
            const int nLags = oV + 1 + oS + 1;
            // whole wavefronts (spare lanes redo the last lag, no store): partial-EXEC loops are slow
            // here; eight elements are read ahead per trip, the sum stays left to right.
            // More than 64 lags (orders 48/30) take several passes: pass k goes to role k % nRoles.
            for (int q0 = role * WAVE + lane; (q0 & ~(WAVE - 1)) < nLags; q0 += nRoles * WAVE) {
                const int q = min(q0, nLags - 1);
                const bool isV = q <= oV;
                const int m = isV ? q : q - (oV + 1);
                const lds_f64 *xw = isV ? B : Cc;
                const lds_f32 *x = (isV ? xv : xsy) + m;
                const lds_f64 *wm = win + m;
                double sum = 0.0;
                const int cnt = max(W - m, 0), c8 = cnt & ~7;       // (a lag beyond the window -- order > W at low sample rates -- sums nothing: LPC.cpp:65)
                for (int n = 0; n < c8; n += 8) {
                    double p_[8], w_[8]; float f_[8];
#pragma unroll
                    for (int u = 0; u < 8; u++) { p_[u] = xw[n + u]; f_[u] = x[n + u]; w_[u] = wm[n + u]; }
#pragma unroll
                    for (int u = 0; u < 8; u++) p_[u] = p_[u] * (double)f_[u];
#pragma unroll
                    for (int u = 0; u < 8; u++) p_[u] = p_[u] * w_[u];
#pragma unroll
                    for (int u = 0; u < 8; u++) sum += p_[u];
                }
                for (int n = c8; n < cnt; n++) sum += xw[n] * (double)x[n] * wm[n];
                sum /= (double)W;
                if (q0 < nLags) { if (isV) rV[m] = sum; else rS[m] = sum; }
            }
        }
        __syncthreads();
        STAMP(d, 17);
#ifdef VP_DIAG_NO_LANES_LEV
        const bool lanesLev = false;
#else
        const bool lanesLev = !LITE && nWaves >= 2 && W >= 128 && oV > 16 && oV <= 48;
#endif
        if (lanesLev) {
            // the voice's recursions of the whole round on wave 0, one window per lane (spare lanes redo the last one); the
            // carrier's (order <= 30, wave-distributed) on the windows' own wavefronts meanwhile, wave 0's window by wave 1
            if (waveHw == 0) {
                lds_f64 *wb = gArr + 8 + (size_t)min(lane, nAct - 1) * voc_wave_doubles(W);
                levinson_lanes<48>((const lds_f64 *)(wb + 4 * (size_t)W), wb + 4 * (size_t)W + (VP_ORDER_MAX + 1), oV, VP_ORDER_MAX + 1, g.levEps);
            } else {
                if (activeW && role == 0) levinson_wave(rS, aS, oS, VP_ORDER_MAX_SYNTH + 1, g.levEps, D);
                if (waveHw == 1) {
                    lds_f64 *wb = gArr + 8, *rS0 = wb + 4 * (size_t)W + 2 * (VP_ORDER_MAX + 1);
                    levinson_wave(rS0, rS0 + (VP_ORDER_MAX_SYNTH + 1), oS, VP_ORDER_MAX_SYNTH + 1, g.levEps, wb + 3 * (size_t)W);
                }
            }
        } else
        if (activeW && role < 2) {                       // whole wavefront, coefficient vector over the lanes
            // eVoice is not written yet: scratch (128 doubles each; the carrier's recursion runs beside the voice's on
            // the window's second wavefront when it has one)
            lds_f64 *scr = (W >= 128) ? D : (lds_f64 *)nullptr;
            lds_f64 *scr2 = (W >= 256) ? D + 128 : (lds_f64 *)nullptr;
            // orders up to 48: the register-resident recursion (every lane the whole of it) -- no LDS round trips
            auto levV = [&](lds_f64 *sc) { if (!LITE && oV >= VP_LEV_SCALAR_MIN && oV <= 48) levinson_scalar<48>((const lds_f64 *)rV, aV, oV, VP_ORDER_MAX + 1, g.levEps);
                                           else levinson_wave(rV, aV, oV, VP_ORDER_MAX + 1, g.levEps, sc); };
            if (nRoles == 1) {
                levV(scr);
                levinson_wave(rS, aS, oS, VP_ORDER_MAX_SYNTH + 1, g.levEps, scr);
            } else if (role == 0)
                levV(scr);
            else
                levinson_wave(rS, aS, oS, VP_ORDER_MAX_SYNTH + 1, g.levEps, scr2);
        }
        __syncthreads();
        STAMP(d, 18);

        // filterFIR (VocoderProcess.cpp:235-251): zero history left of the window.
        // Eight output samples per lane side by side (independent accumulators, taps in the inner
        // position in the reference's order k = 1..order); a tap that reaches left of the window
        // contributes an exact 0.
        // The two residuals touch disjoint arrays (voice: B -> D, carrier: Cc -> A; A's raw samples are dead since the
        // barrier behind the autocorrelation) and every output is its own sum, so the roles share them freely.
        // Work units of 512 outputs, the voice's first, then the carrier's: unit j goes to role j % nRoles.
        if (activeW) {
            const int nU = (W + 8 * WAVE - 1) / (8 * WAVE);
            const int v0 = role;                                            // first voice unit of this role
            const int s0 = ((role - nU) % nRoles + nRoles) % nRoles;        // first carrier unit: nU + s0 = role (mod nRoles)
            fir_window8(B, aV, oV, W, D, lane, v0, nRoles);
            fir_window8(Cc, aS, oS, W, A, lane, s0, nRoles);
        }
        __syncthreads();
        STAMP(d, 19);
#ifdef VP_DIAG_NO_LANES_ENERGY
        const bool lanesE = false;
#else
        const bool lanesE = !LITE && nWaves >= 2 && (W & 1) == 0;
#endif
        if (lanesE) {                                    // wave 0: the round's eVoice sums, wave 1: its eSynth sums, a window per lane
            if (waveHw < 2) {
                const lds_f64 *wb = gArr + 8 + (size_t)min(lane, nAct - 1) * voc_wave_doubles(W);
                const double E = energy_lanes(wb + (waveHw == 0 ? 3 * (size_t)W : 0), W);
                if (lane < nAct) roundE[8 * waveHw + lane] = E;
            }
        } else
        if (activeW && role < 2) {                       // E += e[i]*e[i], left to right (:250)
            if (nRoles == 1) {
                double Ev, Es;
                energy_pair_wave(D, A, W, Ev, Es);
                if (lane == 0) { roundE[wave] = Ev; roundE[8 + wave] = Es; }
            } else if (role == 0) {                      // one sum per wavefront
                const double Ev = energy_wave(D, W);
                if (lane == 0) roundE[wave] = Ev;
            } else {
                const double Es = energy_wave(A, W);
                if (lane == 0) roundE[8 + wave] = Es;
            }
        }
        __syncthreads();
        STAMP(d, 20);

        // filterIIR part 1 (VocoderProcess.cpp:264-275): 10-deep energy histories, window by window.
        if (waveHw == 0) {                               // all lanes redundantly (full EXEC)
            for (int j = 0; j < nAct; j++) {
                for (int i = 9; i > 0; i--) { hist[i] = hist[i - 1]; hist[10 + i] = hist[10 + i - 1]; }
                hist[0] = roundE[j];
                hist[10] = roundE[8 + j];
                double gg = 0.0;
                if (roundE[8 + j] > g.eeFloor) {
                    double sv = 0, ss = 0;
                    for (int i = 0; i < 10; i++) sv += hist[i];
                    for (int i = 0; i < 10; i++) ss += hist[10 + i];
                    gg = sqrt(sv / ss);
                }
                gArr[j] = gg;
            }
        }
        __syncthreads();
        STAMP(d, 21);

        // filterIIR part 2 (:277-286): all-pole recursion, serial in i; one lane per window.
        // One lane per window, all windows of the round in ONE wavefront: chains running in
        // different waves of a workgroup slow each other down almost linearly (measured,
        // tools/ubench_iir.hip), chains in different lanes of one wave cost nothing extra.
        // The chain code must run with EVERY lane of the wave active: executed under a one-lane
        // EXEC mask it is slower and, worse, chains in different waves then serialise (measured,
        // tools/ubench_iir.hip modes 1 vs 3).  Spare lanes redo the last window (identical stores).
        if (LITE || c.iirFast) {
            if (active) {
#ifndef VP_DIAG_NO_HC_IIR
                if (!LITE && (W & 63) == 0 && W >= 5 * WAVE && oV > 16 && oV <= 48) {
                    // block form (iir_block_wave_hc): the window's impulse response (128 samples of the serial form below)
                    // instead of all W of them, then 64 outputs at a time; scratch in D (eVoice: only its energy was needed)
                    lds_f64 *hpad = D, *xp = D + 3 * WAVE;
                    hpad[lane] = 0.0;
                    xp[lane] = (lane == 0) ? 1.0 : 0.0;
                    xp[WAVE + lane] = 0.0;
                    iir_fast_wave(xp, hpad + WAVE, 2 * WAVE, aV, oV, (const lds_f64 *)nullptr, 1.0);
                    iir_block_wave_hc((const lds_f64 *)A, B, W, (const lds_f64 *)aV, oV, 0, hpad, gArr[wave]);
                } else
#endif
                iir_fast_wave(A, B, W, aV, oV, (const lds_f64 *)nullptr, gArr[wave]);   // wave per window, lanes over taps
            }
        } else if (waveHw == 0) {
            const int wj = min(lane, nAct - 1);
            lds_f64 *wb = gArr + 8 + (size_t)wj * voc_wave_doubles(W);        // window `wj` of this round
            const lds_f64 *Aj = wb, *aVj = wb + 4 * (size_t)W + (VP_ORDER_MAX + 1);
            iir_exact(Aj, wb + W, W, aVj, oV, (const lds_f64 *)nullptr, 0, gArr[wj]);
        }
        __syncthreads();
        STAMP(d, 22);
        if (active)                                      // gainVoc * out[i] * stWindow[i] (:291-295)
            for (int i = lane; i < W; i += WAVE) B[i] = sp.gainVoc * B[i] * win[i];
        __syncthreads();

        // overlap-add in gather form: every output sample adds its covering windows in window order,
        // which is the order of the reference's addOutSample calls (MyBuffer.cpp:181-191).
        {
            const int start0 = c.vStart + w0 * g.h;
            const int span = (nAct - 1) * g.h + W;
            for (int t = tid; t < span; t += blockDim.x) {
                int pos = (c.outCounter + start0 + t) % g.outSize;
                double v = acc[pos];
                int jlo = max(0, (t - W + g.h) / g.h), jhi = min(nAct - 1, t / g.h);
                for (int j = jlo; j <= jhi; j++) {
                    int i = t - j * g.h;
                    if (i >= 0 && i < W) {
                        const lds_f64 *Bj = gArr + 8 + (size_t)j * voc_wave_doubles(W) + W;
                        v += Bj[i];
                    }
                }
                acc[pos] = v;
            }
        }
        __syncthreads();
        STAMP(d, 23);
    }
    if (tid < 20) d.EeArr[(size_t)s * 20 + tid] = hist[tid];
}

// Per launch the host may fold the ingest+gate prologue and/or the emit epilogue into this kernel
// (c.fuseIngest / c.fuseEmit): every stage is one-workgroup-per-stream, so the fusion only removes
// kernel boundaries (~10 us each at this size), not parallelism.
#if VP_TU_HAS(1)
__global__ __launch_bounds__(512) void vp_k_vocoder(VpGeom g, VpCall c, VpDev d, const float *__restrict__ in, float *__restrict__ out)
{
    extern __shared__ double smem[];
    VP_POISON(smem, c.ldsBytes);
    if (c.fuseIngest) ingest_gate_block(g, c, d, in);
    vocoder_block<false>(g, c, d, smem);
    if (c.fuseEmit) {
        __syncthreads();
        emit_block(g, c, d, out);
    }
}
__global__ __launch_bounds__(512, 4) void vp_k_vocoder_lite(VpGeom g, VpCall c, VpDev d, const float *__restrict__ in, float *__restrict__ out)
{
    extern __shared__ double smem[];
    VP_POISON(smem, c.ldsBytes);
    if (c.fuseIngest) ingest_gate_block(g, c, d, in);
    vocoder_block<true>(g, c, d, smem);
    if (c.fuseEmit) {
        __syncthreads();
        emit_block(g, c, d, out);
    }
}
#endif

// ------------------------------------------------------------------------------------------------
// Radix-2 complex FFT in LDS (double), whole workgroup, M = 1 << logM points, split re/im arrays.
// Forward: decimation in frequency, natural order in -> BIT-REVERSED order out.
// Inverse: decimation in time, bit-reversed in -> natural out, unscaled (caller divides by M).
// tw[j] = exp(-2 pi i j / M), j < M/2 (host libm).  No reference counterpart: this is the
// accelerator of SURVEY.md 8f(1) (Wiener-Khinchin form of the YIN difference function and of the
// LPC autocorrelation) and the engine of the standalone STFT kernel.
template <class TP>
__device__ __forceinline__ void fft_forward_dif(lds_f64 *zr, lds_f64 *zi, int logM, TP twr, TP twi)
{
    const int M = 1 << logM, tid = threadIdx.x, nt = blockDim.x;
    for (int st = logM - 1; st >= 0; st--) {
        const int half = 1 << st;
        for (int t = tid; t < (M >> 1); t += nt) {
            const int j = t & (half - 1);
            const int i0 = ((t >> st) << (st + 1)) + j, i1 = i0 + half;
            const int tj = j << (logM - 1 - st);
            const double wr = twr[tj], wi = twi[tj];
            const double ur = zr[i0], ui = zi[i0], vr = zr[i1], vi = zi[i1];
            const double dr = ur - vr, di = ui - vi;
            zr[i0] = ur + vr; zi[i0] = ui + vi;
            zr[i1] = dr * wr - di * wi; zi[i1] = dr * wi + di * wr;
        }
        __syncthreads();
    }
}

template <class TP>
__device__ __forceinline__ void fft_inverse_dit(lds_f64 *zr, lds_f64 *zi, int logM, TP twr, TP twi)
{
    const int M = 1 << logM, tid = threadIdx.x, nt = blockDim.x;
    for (int st = 0; st < logM; st++) {
        const int half = 1 << st;
        for (int t = tid; t < (M >> 1); t += nt) {
            const int j = t & (half - 1);
            const int i0 = ((t >> st) << (st + 1)) + j, i1 = i0 + half;
            const int tj = j << (logM - 1 - st);
            const double wr = twr[tj], wi = -twi[tj];                       // conjugate twiddle
            const double ur = zr[i0], ui = zi[i0], xr = zr[i1], xi = zi[i1];
            const double vr = xr * wr - xi * wi, vi = xr * wi + xi * wr;
            zr[i0] = ur + vr; zi[i0] = ui + vi;
            zr[i1] = ur - vr; zi[i1] = ui - vi;
        }
        __syncthreads();
    }
}

__device__ __forceinline__ int bitrev(int k, int logM) { return (int)(__brev((unsigned)k) >> (32 - logM)); }

// ------------------------------------------------------------------------------------------------
// K2: pitch corrector.  PitchProcess::process (PitchProcess.cpp:166-196) for one block: one
// workgroup per stream walks the block's chunk steps in order.
//
// LDS: xs   [toKeep+F]  voice samples idx in [startSample-toKeep, startSample+F) of the step
//      eF   [eLen]      eFrame      (frame position p <-> eF[toKeep+p], PitchProcess.cpp:698)
//      oE   [F]         outEFrame
//      yF   [F]         yFrame
//      dY   [tauMax+1]  yinTemp (+ guard slot), cum [tauMax]
//      r, aPrev [101]
typedef __attribute__((address_space(3))) MinIdx lds_minidx;
struct PitchLds {
    lds_f64 *xs, *eF, *oE, *yF, *dY, *cum, *r, *aPrev, *qtab, *htab, *fft;
    lds_f64 *xcA;      // [1] energy of the YIN window (error bound of the cross-correlation form)
    int *lpcFlag;      // generation number of the Start whose LPC coefficients are ready in aPrev
    int *psFlag;       // generation number of the grain table built ahead of its chunk (pitch_iir)
    lds_state *st;
    lds_minidx *part;  // [8]
    int *ishare;       // [4] (generic pointer: used with atomicMin)
};

// argExt over frame positions [lo, hi) (PitchProcess.cpp:752-776), executed by ONE wavefront with
// all 64 lanes active.  The search windows between marks are (2 - 2 delta) T ~ 0.12 T wide (< 64
// samples), so every lane simply repeats the reference's scan (strict '<' keeps the first minimum);
// only the whole-frame search after an unvoiced frame is split over the lanes and reduced with
// "first index wins" tie-breaking.
__device__ __forceinline__ int wave_arg_min(const PitchLds &L, int toKeep, int lo, int hi)
{
    const lds_f64 *x = L.xs + toKeep;
    if (hi - lo <= 128) {
        double ext = x[lo];
        int arg = lo;
        int i = lo + 1;
        for (; i + 8 <= hi; i += 8) {
            double v[8];
#pragma unroll
            for (int u = 0; u < 8; u++) v[u] = x[i + u];
#pragma unroll
            for (int u = 0; u < 8; u++) if (v[u] < ext) { ext = v[u]; arg = i + u; }
        }
        for (; i < hi; i++) { double v = x[i]; if (v < ext) { ext = v; arg = i; } }
        return arg;
    }
    const int lane = threadIdx.x & 63;
    MinIdx m; m.v = x[lo]; m.i = lo;                      // ext = sample at idxStart, always read
    for (int i = lo + 1 + lane; i < hi; i += WAVE) {
        MinIdx o; o.v = x[i]; o.i = i;
        m = min_first(m, o);
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        MinIdx o;
        o.v = __shfl_xor(m.v, off, WAVE);
        o.i = __shfl_xor(m.i, off, WAVE);
        m = min_first(m, o);
    }
    return m.i;
}

__device__ __forceinline__ int marks_back(const lds_i32 *v, int n, unsigned long long *ub)
{
    // std::vector::back(); empty -> the reference reads the word before the heap block (0 on glibc)
    if (n > 0) return v[n - 1];
    if ((threadIdx.x & 63) == 0) atomicAdd(&ub[2], 1ULL);
    return 0;
}

// PitchProcess::pitchMarks (PitchProcess.cpp:455-567), executed by wave 0 only: all 64 lanes walk
// the (uniform) control flow and make the same stores, so there is no barrier inside.
__device__ __forceinline__ void pitch_marks(const VpGeom &g, const PitchLds &L, unsigned long long *ub)
{
    lds_state *st = L.st;
    const int tid = threadIdx.x;
    {
        for (int i = 0; i < st->nAn; i++) st->prevAnMarks[i] = st->anMarks[i] - g.H;   // prevAnMarks = anMarks; -= hop
        st->nPrevAn = st->nAn;
        int ov = 0;
        for (int i = 0; i < st->nAn; i++) if (st->prevAnMarks[i] >= 0) ov++;
        st->nAnMarksOv = ov;
        st->nAn = 0;
    }
    const int nPrev = st->nPrevAn, nOv = st->nAnMarksOv;
    const double pitch = st->pitch, prevPitch = st->prevPitch;
    const int period = st->period, prevPeriod = st->prevPeriod, pvp = st->prevVoicedPeriod;
    int n = 0, front = 0, back = 0;                  // uniform mirrors of anMarks.size()/front()/back()
    lds_i32 *an = st->anMarks;

#define PUSH_BACK(val) do { int _v = (val); { if (n < VP_MARKS) an[n] = _v; if (n + 1 > 20 && tid == 0) atomicAdd(&ub[4], 1ULL); } \
                            if (n == 0) front = _v; back = _v; if (n < VP_MARKS) n++; } while (0)
#define PUSH_FRONT(val) do { int _v = (val); { int _k = n < VP_MARKS ? n : VP_MARKS - 1; \
                            for (int _i = _k; _i > 0; _i--) an[_i] = an[_i - 1]; an[0] = _v; if (n + 1 > 20 && tid == 0) atomicAdd(&ub[4], 1ULL); } \
                            front = _v; if (n < VP_MARKS) n++; } while (0)

    if (pitch > 1) {
        const int gapMin = (int)floor(g.delta * period);
        const int gapMax = (int)ceil((2.0 - g.delta) * period);
        bool growLeft = false;
        int t;
        if (prevPitch > 1) {
            if (nOv == 0) {
                int tailMark = marks_back(st->prevAnMarks, nPrev, ub);
                int winLo = max(tailMark + min(gapMin, (int)floor(g.delta * min(prevPeriod, period))), 0);
                int winHi = min(tailMark + max(gapMax, (int)ceil((2 - g.delta) * max(prevPeriod, period))), g.F);
                t = wave_arg_min(L, g.toKeep, winLo, winHi);
            } else
                t = st->prevAnMarks[nPrev - nOv];
        } else {
            growLeft = true;
            t = wave_arg_min(L, g.toKeep, 0, g.F);
        }
        PUSH_BACK(t);
        while (back + gapMin < g.F) {                                         // :505-519
            if (back + gapMax < g.F) {
                int m = wave_arg_min(L, g.toKeep, back + gapMin, back + gapMax);
                PUSH_BACK(m);
            } else {
                if (back + period < g.F) {
                    int m = wave_arg_min(L, g.toKeep, back + gapMin, g.F);
                    PUSH_BACK(m);
                }
                break;
            }
        }
        if (growLeft) {                                                    // :522-539
            while (front - gapMin > 0) {
                if (front - gapMax >= 0) {
                    int m = wave_arg_min(L, g.toKeep, front - gapMax, front - gapMin);
                    PUSH_FRONT(m);
                } else {
                    if (front - period >= 0) {
                        int m = wave_arg_min(L, g.toKeep, 0, front - gapMin);
                        PUSH_FRONT(m);
                    }
                    break;
                }
            }
        }
    } else {
        if (nPrev != 0) {                                                    // :545-565
            if (nOv > 0) {
                for (int i = 0; i < nOv; i++) PUSH_BACK(st->prevAnMarks[nPrev - nOv + i]);
            } else
                PUSH_BACK(marks_back(st->prevAnMarks, nPrev, ub) + pvp);
            if (pvp > 0)
                while (back + pvp < g.F) PUSH_BACK(back + pvp);
        }
    }
#undef PUSH_BACK
#undef PUSH_FRONT
    st->nAn = n;
}

// Notes::getClosestFreq (Notes.cpp:79-110) on the precomputed table of `key`.
__device__ __forceinline__ double notes_closest(const double *freq, int size, double pitch)
{
    int lo = 0, hi = size;
    while (lo < hi) {
        int mid = lo + (hi - lo) / 2;
        if (freq[mid] < pitch) lo = mid + 1; else hi = mid;
    }
    int idx = lo;
    if (idx > 0) {
        if (fabs(freq[idx] - pitch) <= fabs(freq[idx - 1] - pitch)) return freq[idx];   // may read the popped slot
        return freq[idx - 1];
    }
    return freq[idx];
}

// PitchProcess::placeStMarks (PitchProcess.cpp:573-658), one lane.
__device__ __forceinline__ void place_st_marks(const VpGeom &g, const VpCall &c, const VpDev &d, lds_state *st)
{
    unsigned long long *ub = d.ub;
    for (int i = 0; i < st->nSt; i++) st->prevStMarks[i] = st->stMarks[i] - g.H;
    st->nPrevSt = st->nSt;
    st->nSt = 0;
    st->nStMarksOv = 0;
    if (st->nAn == 0) return;
    int nOv = 0;
    for (int i = 0; i < st->nPrevSt; i++) if (st->prevStMarks[i] >= 0) nOv++;
    st->nStMarksOv = nOv;
    st->prevClosestFreq = st->closestFreq;
    if (st->pitch > 1) {
        const int key = st->sp.key;
        st->closestFreq = notes_closest(d.notes + (size_t)key * VP_NOTES_STRIDE, d.notesN[key], st->pitch);
        st->beta = st->closestFreq / st->pitch;
        if (st->sp.shiftOn) {                       // vp_set_pitch_shift (extension): a fixed interval instead of the nearest note
            st->beta = st->sp.shiftBeta;
            st->closestFreq = st->beta * st->pitch;
        }
        st->periodNew = (int)round(st->period / st->beta);
    } else {
        st->closestFreq = 0;
        st->periodNew = st->prevVoicedPeriod;
    }
    if (st->periodNew <= 0) return;                 // the reference asserts (:604-608)
    const int nPrev = st->nPrevSt;
    const int periodNew = st->periodNew;
    int headMark;
    if (st->pitch > 1) {
        if (st->prevPitch > 1) {
            if (nOv > 0)
                headMark = st->prevStMarks[nPrev - nOv];
            else {
                int b = marks_back(st->prevStMarks, nPrev, ub);
                headMark = (b + periodNew >= 0) ? b + periodNew : st->anMarks[0];
            }
        } else
            headMark = st->anMarks[0];
    } else {
        if (nPrev == 0) return;
        if (nOv > 0)
            headMark = st->prevStMarks[nPrev - nOv];
        else {
            int b = st->prevStMarks[nPrev - 1];
            int n = 1;
            while (b + n * periodNew < 0) n += 1;
            headMark = b + n * periodNew;
        }
    }
    int n = 0;
    st->stMarks[n++] = headMark;
    while (st->stMarks[n - 1] + periodNew < g.F) {
        int v = st->stMarks[n - 1] + periodNew;
        if (n < VP_MARKS) st->stMarks[n] = v;
        if (n + 1 > 20 && threadIdx.x == 0) atomicAdd(&ub[4], 1ULL);
        if (n < VP_MARKS) n++; else break;
    }
    st->nSt = n;
}

// PitchProcess::getClosestAnMarkIdx (PitchProcess.cpp:788-831); uniform, read-only.
__device__ __forceinline__ int closest_an_mark_idx(const VpGeom &g, const lds_state *st, int stMark, int T, int nChunk,
                                                   int pS, bool &q2hit)
{
    q2hit = false;
    const lds_i32 *an = st->anMarks;
    const int nAn = st->nAn;
    int lo = 0, hi = nAn;
    while (lo < hi) {
        int mid = lo + (hi - lo) / 2;
        if (an[mid] < stMark) lo = mid + 1; else hi = mid;
    }
    const int idx = lo;
    const int avail = g.bufferIdxMax - pS;
    const int sh = nChunk * g.C;
    if (idx > 0 && idx < nAn) {
        if (abs(an[idx] - stMark) <= abs(an[idx - 1] - stMark) && an[idx] + T - sh < avail) return idx;
        if (an[idx - 1] + T - sh < avail) return idx - 1;
        if (idx - 2 > 0) return idx - 2;
        return -st->nAnMarksOv - 1;                                         // :812 -> Q3
    }
    if (idx == 0) return 0;
    q2hit = true;                                                            // Q2: anMarks[size]
    int stale = idx < VP_MARKS ? an[idx] : 0;
    if (stale + T - sh < avail) return idx - 1;
    if (idx - 2 >= 0) return idx - 2;
    return idx - 1;
}

// PitchProcess::psola (PitchProcess.cpp:665-741) + interp (:842-870) in GATHER form.
// Pass 1 (wave 0, every lane redundantly): walk the synthesis marks that are due in this chunk in
// order and write one table entry per grain (mark, source offset in eFrame, first/last flags, the
// grain's x-range and integer output range).  Pass 2 (all threads): every output sample adds the
// grains that cover it, in grain order -- the order in which the reference's interp() calls
// accumulate into outEFrame -- so there is no barrier between grains and all lanes stay busy.
struct GrainTab {                       // lives in the yinTemp/cum scratch, free at this point
    lds_f64 *x0, *xN;                   // [VP_MARKS]
    lds_i32 *stMark, *srcBase, *flags, *startIdx, *stopIdx;   // [VP_MARKS]
};

__device__ __forceinline__ GrainTab psola_grain_tab(const PitchLds &L)
{
    GrainTab G;
    G.x0 = L.dY; G.xN = L.dY + VP_MARKS;
    G.stMark = (lds_i32 *)(L.dY + 2 * VP_MARKS); G.srcBase = G.stMark + VP_MARKS; G.flags = G.srcBase + VP_MARKS;
    G.startIdx = G.flags + VP_MARKS; G.stopIdx = G.startIdx + VP_MARKS;
    return G;
}

// Pass 1 of a chunk's PSOLA, by ONE full wavefront (lane = tid & 63): one lane per synthesis mark.  The marks ascend,
// so the marks that are due in this chunk (:685 stMark - T < (nChunk+1) C) form a prefix of the pending ones; each
// lane prepares its own grain entry (closest analysis mark :692-694, x-range, output range).  Leaves the number of
// grains in ishare[1] and advances st->stMarkIdx.
__device__ __forceinline__ void psola_table_wave(const VpGeom &g, const VpDev &d, const PitchLds &L, int nChunk, int pS)
{
    lds_state *st = L.st;
    const int lane = vp_tid() & 63;
    const int T = (st->pitch > 1) ? st->period : st->prevVoicedPeriod;
    const int nG = 2 * T + 1;
    const int nSt = st->nSt;
    const GrainTab G = psola_grain_tab(L);
    const int smi0 = st->stMarkIdx;
    const int smi = smi0 + lane;
    const bool have = smi < nSt;
    const int stMark = have ? st->stMarks[smi] : 0;
    const bool due = have && !(stMark - T >= (nChunk + 1) * g.C);
    const unsigned long long dueMask = __ballot(due);
    const unsigned long long notDue = ~dueMask;
    const int ng = notDue ? (int)__builtin_ctzll(notDue) : 64;          // length of the due prefix
    if (lane < ng) {
        bool q2 = false;
        int clIdx = closest_an_mark_idx(g, st, stMark, T, nChunk, pS, q2);
        if (q2) atomicAdd(&d.ub[0], 1ULL);
        int clAnMark;
        if (clIdx >= 0)
            clAnMark = st->anMarks[clIdx];
        else {                                                           // Q3
            int j = st->nPrevAn - clIdx;
            atomicAdd(&d.ub[1], 1ULL);
            clAnMark = (j >= 0 && j < VP_MARKS) ? st->prevAnMarks[j] : 0;
        }
        const double dSt = (double)stMark;
        const double x0 = dSt + L.qtab[0];                              // xInterp[0]
        const double xN = dSt + L.qtab[nG - 1];                         // xInterp.back()
        G.x0[lane] = x0; G.xN[lane] = xN;
        G.stMark[lane] = stMark;
        G.srcBase[lane] = g.toKeep + clAnMark - T;
        G.flags[lane] = (smi == 0 ? 1 : 0) | (smi == nSt - 1 ? 2 : 0);
        G.startIdx[lane] = max((int)floor(x0), 0);
        G.stopIdx[lane] = min((int)ceil(xN), g.F);
    }
    if (lane == 0) { st->stMarkIdx = smi0 + ng; L.ishare[1] = ng; }
}

// Pass 2: every output sample adds the grains that cover it, in grain order; thread `tix` of `nthr`.  Samples below
// `loMin` are skipped: with loMin = nChunk C those are outEFrame entries of chunks that have already been filtered and
// sent out, which nothing reads again (filterIIR :307-322 takes [nChunk C, (nChunk+1) C)), see Q5 in SURVEY.md.
__device__ __forceinline__ void psola_pass2(const VpGeom &g, const PitchLds &L, int tix, int nthr, int loMin)
{
    lds_state *st = L.st;
    const int T = (st->pitch > 1) ? st->period : st->prevVoicedPeriod;
    const int nG = 2 * T + 1;
    const lds_f64 *hw = L.htab;                       // d.hannTab + d.hannOff[T], staged by psola()
    const double beta = st->beta;
    const GrainTab G = psola_grain_tab(L);
    const int ng = L.ishare[1];
    if (ng > 0) {
        const int lo = max(G.startIdx[0], loMin);                           // marks ascend: first grain starts first
        int hi = 0;
        for (int q = 0; q < ng; q++) hi = max(hi, G.stopIdx[q]);
        for (int i = lo + tix; i < hi; i += nthr) {
            const double di = (double)i;
            double accv = L.oE[i];
            for (int q = 0; q < ng; q++) {
                if (i < G.startIdx[q] || i >= G.stopIdx[q]) continue;
                const double x0 = G.x0[q], xN = G.xN[q];
                if (!(di >= x0 && di <= xN)) continue;                      // :852
                const double dSt = (double)G.stMark[q];
                const int srcBase = G.srcBase[q], fl = G.flags[q];
                // std::lower_bound on the strictly increasing x[j] = stMark + (j - T)/beta (:853)
                int j = (int)ceil((di - dSt) * beta) + T;
                j = max(0, min(j, nG - 1));
                double xj = dSt + L.qtab[j], xjm = 0.0;
                while (j < nG - 1 && xj < di) { j++; xj = dSt + L.qtab[j]; }
                while (j > 0) {
                    xjm = dSt + L.qtab[j - 1];
                    if (xjm >= di) { j--; xj = xjm; } else break;
                }
                auto ys = [&](int jj) -> double {
                    int src = srcBase + jj;
                    double e = (src >= 0 && src < g.eLen) ? L.eF[src] : 0.0;
                    bool windowed = (fl & 1) ? (jj >= T) : ((fl & 2) ? (jj < T) : true);   // :696-731 (first wins)
                    return windowed ? e * hw[jj] : e;
                };
                double value;
                if (j > 0) {
                    double ya = ys(j - 1), yb = ys(j);
                    value = ya + (yb - ya) / (xj - xjm) * (di - xjm);          // :860
                } else
                    value = ys(0);
                accv += value;
            }
            L.oE[i] = accv;
        }
    }
}

__device__ __forceinline__ void psola(const VpGeom &g, const VpDev &d, const PitchLds &L, int nChunk, int pS, bool &qValid)
{
    lds_state *st = L.st;
    const int tid = vp_tid(), nt = blockDim.x;
    // xInterp[j] - stMark = (j - T)/beta is the same for every grain of the frame (:700,715,731):
    // the 2T+1 quotients are computed once per frame, x[j] is then one exact add away.
    if (!qValid) {                                    // once per frame and kernel launch; later chunks come here behind a barrier
        const int T = (st->pitch > 1) ? st->period : st->prevVoicedPeriod;
        const int nG = 2 * T + 1;
        const double beta = st->beta;
        const double *hg = d.hannTab + d.hannOff[T];
        for (int j = tid; j < nG; j += nt) { L.qtab[j] = (double)(j - T) / beta; L.htab[j] = hg[j]; }
        qValid = true;
        __syncthreads();
    }
    if (tid < WAVE) psola_table_wave(g, d, L, nChunk, pS);
    __syncthreads();
    STAMP(d, 14);
    psola_pass2(g, L, tid, nt, 0);
    __syncthreads();
    STAMP(d, 7);
}

// PitchProcess::filterFIR (PitchProcess.cpp:280-302) for FOUR consecutive outputs e[j0 .. j0+3] by one thread:
// e[j] = a[0] x[j] + sum_{k=1..min(order, j)} x[j-k] a[k], every output summed in the reference's order k = 1, 2, ...
// The four outputs share the sliding window of inputs (one new x and one coefficient per tap for eight operations,
// four independent chains) instead of two LDS reads per multiply-add.  x points at the sample of output 0 of the
// whole filter call (so that j - k >= 0 is the history test), eo at its output.
__device__ __forceinline__ void fir4(const lds_f64 *x, const lds_f64 *a, int order, int j0, int jEnd, lds_f64 *eo)
{
    if (j0 >= jEnd) return;
    if (j0 >= order && j0 + 4 <= jEnd) {
        double w0 = x[j0], w1 = x[j0 + 1], w2 = x[j0 + 2], w3 = x[j0 + 3];
        const double a0 = a[0];
        double e0 = a0 * w0, e1 = a0 * w1, e2 = a0 * w2, e3 = a0 * w3;
        int k = 1;
        for (; k + 3 <= order; k += 4) {                                     // four taps per trip, window registers rotate by name
            const double ak0 = a[k], ak1 = a[k + 1], ak2 = a[k + 2], ak3 = a[k + 3];
            const double n0 = x[j0 - k], n1 = x[j0 - k - 1], n2 = x[j0 - k - 2], n3 = x[j0 - k - 3];
            e0 += n0 * ak0; e1 += w0 * ak0; e2 += w1 * ak0; e3 += w2 * ak0;   // tap k:   x[j-k] for j = j0..j0+3
            e0 += n1 * ak1; e1 += n0 * ak1; e2 += w0 * ak1; e3 += w1 * ak1;   // tap k+1
            e0 += n2 * ak2; e1 += n1 * ak2; e2 += n0 * ak2; e3 += w0 * ak2;   // tap k+2
            e0 += n3 * ak3; e1 += n2 * ak3; e2 += n1 * ak3; e3 += n0 * ak3;   // tap k+3
            w3 = n0; w2 = n1; w1 = n2; w0 = n3;                               // the window is now x[j0-k-3 .. j0-k]
        }
        for (; k <= order; k++) {
            const double ak = a[k], n0 = x[j0 - k];
            e0 += n0 * ak; e1 += w0 * ak; e2 += w1 * ak; e3 += w2 * ak;
            w3 = w2; w2 = w1; w1 = w0; w0 = n0;
        }
        eo[j0] = e0; eo[j0 + 1] = e1; eo[j0 + 2] = e2; eo[j0 + 3] = e3;
        return;
    }
    for (int j = j0; j < min(j0 + 4, jEnd); j++) {                           // the filter's first outputs (short history) and ragged ends
        double e = a[0] * x[j];
        const int kmax = min(order, j);
        for (int k = 1; k <= kmax; k++) e += x[j - k] * a[k];
        eo[j] = e;
    }
}

// filterFIR(F-C, C, toKeep+F+(n-1)C) :253-259 -- the C new residual samples of chunk nChunk, four per thread (fir4), by ONE
// wavefront; xsStep = the voice window of the step that chunk belongs to
__device__ __forceinline__ void fir_cont_wave(const VpGeom &g, const PitchLds &L, const lds_f64 *xsStep, int nChunk)
{
    const int lane = vp_tid() & 63;
    const int x0 = g.toKeep + g.F - g.C;                                   // first input sample of the chunk; full history left of it
    for (int j = x0 + 4 * lane; j < g.toKeep + g.F; j += 4 * WAVE)
        fir4(xsStep, (const lds_f64 *)L.st->a, g.orderPitch, j, g.toKeep + g.F, L.eF + nChunk * g.C);
}

// The EXACT all-pole recursion y[i] = x[i] - sum_{k=1..order} a[k] y[i-k] for orders <= 16, by one full wavefront, in the
// reference's operation order (product rounded, then subtracted, k = 1, 2, ...), organised around the DPP row broadcast:
//   * lane m of every 16-lane row holds the output y[i-1-m] (Yh) and the coefficient a[m+1]: ONE v_mul_f64 gives all
//     the sample's products t_k = a[k] y[i-k], each correctly rounded, instead of `order` multiplies issued in between
//     the chain's subtractions (an fp64 op issued in front of a dependent one is not hidden, section 4.2);
//   * the chain acc -= t_k is v_fmac_f64 acc, bcast(T, k-1), -1.0: t * (-1) is exact, so the fused operation rounds
//     exactly like the subtraction;
//   * the new output enters Yh through a DPP row shift (lane 0 of each row takes acc), inputs are read and outputs
//     written sixteen at a time.
// hist[j] = y[-1-j] (j < order) or nullptr.  n a multiple of 16.
// (the order is a template parameter: a wave-uniform branch per tap cost three times the tap itself)
#define VP_XR_TAP(K) if ((K) <= ORD) { VP_FMAC_BCAST(acc, T, mone, (K) - 1); }
template <int ORD, class XP, class YP, class AP, class HP>
__device__ __forceinline__ void iir_exact_row16_t(XP x, YP y, int n, AP aL, HP hist, double gmul)
{
    const int lane = threadIdx.x & 63, m = lane & 15;
    constexpr int order = ORD;
    const double A = (m + 1 <= order) ? aL[m + 1] : 0.0;
    const double mone = -1.0, zero = 0.0;
    double Yh = (hist && m < order) ? hist[m] : 0.0;
    for (int i0 = 0; i0 < n; i0 += 16) {
        const double X = x[i0 + m];                                         // sixteen inputs, one per lane of the row
        double Yout = 0.0;                                                  // the sixteen outputs, gathered the same way
#define VP_XR_SAMPLE(U) { \
            double T = Yh * A;                                              /* t_k for every k at once */ \
            double acc = zero * zero;                                       /* +0.0 in a fresh register */ \
            asm volatile("s_nop 1" : "+v"(T), "+v"(acc));                   /* VALU write -> DPP read */ \
            VP_FMAC_BCAST(acc, X, gmul, U);                                 /* acc = gmul * x[i0+U] (one rounding, as the product) */ \
            VP_XR_TAP(1) VP_XR_TAP(2) VP_XR_TAP(3) VP_XR_TAP(4) VP_XR_TAP(5) VP_XR_TAP(6) VP_XR_TAP(7) VP_XR_TAP(8) \
            VP_XR_TAP(9) VP_XR_TAP(10) VP_XR_TAP(11) VP_XR_TAP(12) VP_XR_TAP(13) VP_XR_TAP(14) VP_XR_TAP(15) VP_XR_TAP(16) \
            if (m == (U)) Yout = acc; \
            /* Yh[m] <- Yh[m-1], Yh[0] <- acc: DPP row_shr:1, lane 0 of each row keeps `old` = acc */ \
            const int lo_ = __builtin_amdgcn_update_dpp(__double2loint(acc), __double2loint(Yh), 0x111, 0xf, 0xf, false); \
            const int hi_ = __builtin_amdgcn_update_dpp(__double2hiint(acc), __double2hiint(Yh), 0x111, 0xf, 0xf, false); \
            Yh = __hiloint2double(hi_, lo_); }
        VP_XR_SAMPLE(0) VP_XR_SAMPLE(1) VP_XR_SAMPLE(2) VP_XR_SAMPLE(3) VP_XR_SAMPLE(4) VP_XR_SAMPLE(5) VP_XR_SAMPLE(6) VP_XR_SAMPLE(7)
        VP_XR_SAMPLE(8) VP_XR_SAMPLE(9) VP_XR_SAMPLE(10) VP_XR_SAMPLE(11) VP_XR_SAMPLE(12) VP_XR_SAMPLE(13) VP_XR_SAMPLE(14) VP_XR_SAMPLE(15)
#undef VP_XR_SAMPLE
        y[i0 + m] = Yout;                                                   // (all four rows store the same values)
    }
}
#undef VP_XR_TAP

template <class XP, class YP, class AP, class HP>
__device__ __forceinline__ void iir_exact_row16(XP x, YP y, int n, AP aL, int order_, HP hist, double gmul)
{
    switch (__builtin_amdgcn_readfirstlane(order_)) {
#define VP_XR_CASE(O) case O: iir_exact_row16_t<O>(x, y, n, aL, hist, gmul); break;
        VP_XR_CASE(2) VP_XR_CASE(3) VP_XR_CASE(4) VP_XR_CASE(5) VP_XR_CASE(6) VP_XR_CASE(7) VP_XR_CASE(8) VP_XR_CASE(9)
        VP_XR_CASE(10) VP_XR_CASE(11) VP_XR_CASE(12) VP_XR_CASE(13) VP_XR_CASE(14) VP_XR_CASE(15) VP_XR_CASE(16)
#undef VP_XR_CASE
        default: break;                                                      // orders are 2..100 (params_valid); > 16 never comes here
    }
}

// Which block form runs the chunk's recursion (FAST mode, whole 64-sample blocks): orders up to 16 the register form, 17..48
// iir_block_wave_hc (it needs 128 samples of the impulse response instead of 64), else the LDS form.  The register-light
// builds keep the LDS form (the other two hold 64 taps in registers).
template <bool LITE, bool COMMON>
__device__ __forceinline__ bool pitch_iir_hc(const VpGeom &g)
{
#ifdef VP_DIAG_NO_HC_IIR
    return false;
#else
    return !LITE && !COMMON && g.orderPitch > 16 && g.orderPitch <= 48;
#endif
}
// LDS scratch of the block forms inside cum[]: 64 zeros, the impulse response (up to 128 samples), 128 doubles of input
#define VP_HPAD_OFF 128
#define VP_XP_OFF 320
template <bool LITE, bool COMMON>
__device__ __forceinline__ void pitch_impulse_response(const VpGeom &g, const PitchLds &L, const lds_f64 *a)
{
    lds_f64 *hpad = L.cum + VP_HPAD_OFF, *xp = L.cum + VP_XP_OFF;
    const int lane = threadIdx.x & 63;
    const int nh = pitch_iir_hc<LITE, COMMON>(g) ? 2 * WAVE : WAVE;
    hpad[lane] = 0.0;
    xp[lane] = (lane == 0) ? 1.0 : 0.0;
    if (nh > WAVE) xp[WAVE + lane] = 0.0;
    iir_fast_wave(xp, hpad + WAVE, nh, a, g.orderPitch, (const lds_f64 *)nullptr, 1.0);
}

// PitchProcess::filterIIR (PitchProcess.cpp:307-322): serial recursion.  Called by ONE wavefront;
// all 64 lanes run the same chain redundantly (full EXEC mask: see the vocoder's note), identical stores.
template <bool LITE, bool FAST, bool COMMON>
__device__ __forceinline__ void pitch_iir_wave(const VpGeom &g, const PitchLds &L, int nChunk, bool &hValid)
{
    if (FAST && (COMMON || ((g.C & 63) == 0 && g.orderPitch < WAVE))) {
        // block form; the impulse response of the frame's 1/A(z) lives in cum[128..256) (64 zeros in front)
        const int shift = nChunk * g.C, order = g.orderPitch;
        lds_f64 *hpad = L.cum + VP_HPAD_OFF, *xp = L.cum + VP_XP_OFF;
        if (!hValid) {                                   // normally precomputed (Start, wave 7) or reloaded (kernel start)
            pitch_impulse_response<LITE, COMMON>(g, L, (const lds_f64 *)L.st->a);
            hValid = true;
        }
#ifdef VP_DIAG_NO_REGS_IIR
        if (false)
#else
        if (!LITE && (COMMON || order <= 16))
#endif
            iir_block_wave_regs((const lds_f64 *)(L.oE + shift), L.yF + shift, g.C, (const lds_f64 *)L.st->a, order, shift > 0, hpad);
        else if (pitch_iir_hc<LITE, COMMON>(g))
            iir_block_wave_hc((const lds_f64 *)(L.oE + shift), L.yF + shift, g.C, (const lds_f64 *)L.st->a, order, min(order, shift), hpad, 1.0);
#ifndef VP_DIAG_NO_HC16_IIR
        else if (LITE && (COMMON || order <= 16))
            iir_block_wave_hc16((const lds_f64 *)(L.oE + shift), L.yF + shift, g.C, (const lds_f64 *)L.st->a, order, min(order, shift), hpad, 1.0);
#endif
        else
            iir_block_wave(L.oE + shift, L.yF + shift, g.C, (const lds_f64 *)L.st->a, order, min(order, shift), hpad, xp, 1.0);
        return;
    }
    {
        const int shift = nChunk * g.C, order = g.orderPitch;
        // history y[shift-1-j]; the frame starts from a zero state (yFrame is zero-filled at the
        // frame start, so reading it as history for shift > 0 is the same thing)
        lds_f64 *hist = L.cum;                          // yinTemp scratch is free here
        const int nh = min(order, shift);
        for (int j = 0; j < order; j++) hist[j] = (j < nh) ? L.yF[shift - 1 - j] : 0.0;
        if (FAST) iir_fast_wave(L.oE + shift, L.yF + shift, g.C, (const lds_f64 *)L.st->a, order, (const lds_f64 *)hist, 1.0);
        else if (COMMON || (order <= 16 && (g.C & 15) == 0))
            iir_exact_row16(L.oE + shift, L.yF + shift, g.C, (const lds_f64 *)L.st->a, order, (const lds_f64 *)hist, 1.0);
        else iir_exact<LITE>(L.oE + shift, L.yF + shift, g.C, (const lds_f64 *)L.st->a, order, (const lds_f64 *)hist, shift, 1.0);
    }
}

template <bool LITE, bool FAST, bool COMMON>
// The chunk's IIR on wave 0.  `ahead`: the frame's NEXT chunk belongs to the next step of this block and its voice
// window (xsNext) is staged, so the other wavefronts do that chunk's residual FIR, grain table (wave 1, which then raises
// a flag) and second PSOLA pass now instead of waiting: none of it touches what the recursion reads or writes (the
// second pass skips the outEFrame entries below its own chunk, which are dead anyway, see psola_pass2).
__device__ __forceinline__ void pitch_iir(const VpGeom &g, const VpDev &d, const PitchLds &L, int nChunk, bool &hValid, bool ahead,
                                          const lds_f64 *xsNext, int pSNext, int &psGen)
{
    const int tid = vp_tid(), nt = blockDim.x;
    const int gen = ahead ? (psGen += 2) : 0;       // the flag counts two producers per use
    if (tid < WAVE) pitch_iir_wave<LITE, FAST, COMMON>(g, L, nChunk, hValid);
    else if (ahead) {
        const int wv = tid >> 6, nw = nt >> 6;
        const bool two = nw >= 3;                        // residual on wave 2, grain table on wave 1, side by side
        if (wv == 1 || (two && wv == 2)) {
            if (!two || wv == 2) fir_cont_wave(g, L, xsNext, nChunk + 1);
            if (wv == 1) psola_table_wave(g, d, L, nChunk + 1, pSNext);
            __threadfence_block();
            if ((tid & 63) == 0) __hip_atomic_fetch_add(L.psFlag, two ? 1 : 2, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
        int spin = 0;                                  // bounded: a bug shows as a parity failure and a count, never as a hang
        for (; spin < (1 << 22) && __hip_atomic_load(L.psFlag, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) != gen; spin++)
            __builtin_amdgcn_s_sleep(1);
        if (spin == (1 << 22) && (tid & 63) == 0) atomicAdd(&d.dbg[61], 1ULL);
        // (wave 4 shares wave 0's SIMD and stays out of the recursion's way)
        if (nw != 8) psola_pass2(g, L, tid - WAVE, nt - WAVE, (nChunk + 1) * g.C);
        else if (wv != 4) psola_pass2(g, L, (wv < 4 ? wv - 1 : wv - 2) * WAVE + (tid & 63), 6 * WAVE, (nChunk + 1) * g.C);
    }
    __syncthreads();
    STAMP(d, 8);
}

// PitchProcess::fillOutputBuffer (PitchProcess.cpp:328-342) by the calling wavefront alone
__device__ __forceinline__ void pitch_fill_output_wave(const VpGeom &g, const VpCall &c, const VpDev &d, const PitchLds &L,
                                                       int nChunk, int pS, int s)
{
    double *acc = d.outAcc + (size_t)s * g.outSize;
    const double gainPitch = L.st->sp.gainPitch;
    for (int i = vp_tid() & 63; i < g.C; i += WAVE) {
        int pos = (c.outCounter + pS + i) % g.outSize;
        acc[pos] += L.yF[i + nChunk * g.C] * d.pitchStWin[i + nChunk * g.C] * gainPitch;
    }
}

__device__ __forceinline__ void pitch_fill_output(const VpGeom &g, const VpCall &c, const VpDev &d, const PitchLds &L,
                                  int nChunk, int pS, int s)
{
    // PitchProcess::fillOutputBuffer (PitchProcess.cpp:328-342)
    double *acc = d.outAcc + (size_t)s * g.outSize;
    const double gainPitch = L.st->sp.gainPitch;
    for (int i = vp_tid(); i < g.C; i += blockDim.x) {
        int pos = (c.outCounter + pS + i) % g.outSize;
        acc[pos] += L.yF[i + nChunk * g.C] * d.pitchStWin[i + nChunk * g.C] * gainPitch;
    }
    __syncthreads();
    STAMP(d, 9);
}

// computeYinTemp's sums (PitchProcess.cpp:350-403) in the reference's own arithmetic, two adjacent lags per lane on
// waves yw0 .. yw0+wavesY-1, into dY.  (Also the fallback of the cross-correlation form, see yin_pick.)
__device__ __forceinline__ void yin2_exact_waves(const VpGeom &g, const PitchLds &L, int base, int tid, int yw0, int wavesY, int nPairs)
{
    typedef double d2 __attribute__((ext_vector_type(2)));
    typedef __attribute__((address_space(3))) d2 lds_d2;
    if ((tid >> 6) >= yw0 && (tid >> 6) < yw0 + wavesY) {        // whole wavefronts; spare lanes redo the last pair
                const int ty = tid - yw0 * WAVE;
                const int l = min(ty, nPairs - 1);
                const lds_f64 *xa = L.xs + base, *xw = L.xs + base + 2 * l;
                double accA = 0.0, accB = 0.0;
                const int F8 = g.F & ~7;
                double w0 = xw[0], w1 = xw[1];
                d2 a0[4], v0[4], a1[4], v1[4];
#define VP_Y2LOAD(A, V, I) _Pragma("unroll") for (int u = 0; u < 4; u++) { A[u] = *(const lds_d2 *)(xa + (I) + 2 * u); V[u] = *(const lds_d2 *)(xw + (I) + 2 + 2 * u); }
#define VP_Y2COMP(A, V) { const double e_[8] = {A[0].x, A[0].y, A[1].x, A[1].y, A[2].x, A[2].y, A[3].x, A[3].y}; \
        const double w_[10] = {w0, w1, V[0].x, V[0].y, V[1].x, V[1].y, V[2].x, V[2].y, V[3].x, V[3].y}; \
        double dA_[8], dB_[8]; \
        _Pragma("unroll") for (int u = 0; u < 8; u++) { dA_[u] = e_[u] - w_[u]; dB_[u] = e_[u] - w_[u + 1]; } \
        __builtin_amdgcn_sched_barrier(0); \
        _Pragma("unroll") for (int u = 0; u < 8; u++) { dA_[u] = dA_[u] * dA_[u]; dB_[u] = dB_[u] * dB_[u]; } \
        __builtin_amdgcn_sched_barrier(0); \
        _Pragma("unroll") for (int u = 0; u < 8; u++) { accA += dA_[u]; accB += dB_[u]; } \
        __builtin_amdgcn_sched_barrier(0); \
        w0 = w_[8]; w1 = w_[9]; }
                if (F8 > 0) { VP_Y2LOAD(a0, v0, 0) }
                for (int i = 0; i < F8; i += 16) {
                    const bool more1 = i + 8 < F8;
                    if (more1) { VP_Y2LOAD(a1, v1, i + 8) }
                    VP_Y2COMP(a0, v0)
                    if (more1) {
                        if (i + 16 < F8) { VP_Y2LOAD(a0, v0, i + 16) }
                        VP_Y2COMP(a1, v1)
                    }
                }
#undef VP_Y2LOAD
#undef VP_Y2COMP
                for (int i = F8; i < g.F; i++) {
                    double dA = xa[i] - xw[i], dB = xa[i] - xw[i + 1];
                    accA += dA * dA; accB += dB * dB;
                }
                if (ty < nPairs) {
                    L.dY[2 * l] = accA;
                    if (2 * l + 1 < g.tauMax) L.dY[2 * l + 1] = accB;
                }
    }
}

// The running sum of the difference function (PitchProcess.cpp:395-402) by ONE full wavefront, into cum[].
__device__ __forceinline__ void yin_cumsum_wave(const VpGeom &g, const PitchLds &L, int tid)
{
        // Sixteen entries per trip, ONE read and ONE write for the whole group: lane l holds entry
        // k0 + (l & 15) (every 16-lane row the same), and the chain is sixteen v_fmac_f64 with a DPP
        // row-broadcast operand, cap += entry_u * M_u, where lane l's multiplier M_u is 1.0 for u <= (l & 15)
        // and 0.0 after.  x * 1.0 + cap rounds exactly like cap + x and x * 0.0 + cap is cap (the entries
        // are finite sums of squares), so lane l ends the trip holding the running sum up to ITS entry --
        // the same additions in the same order as the serial loop -- and lane 15's value starts the next
        // trip.  The per-entry form (uniform reads, same-address writes) spent 2.5x the chain's time on
        // LDS instructions issued from the chain's own wave (tools/ubench_lds.hip).
        const int lane = tid, l16 = lane & 15;
        const double one = 1.0, zero = 0.0;
        double M[16];
#pragma unroll
        for (int u = 0; u < 16; u++) M[u] = (l16 >= u) ? 1.0 : 0.0;
        double cap = 0.0;                                                     // running sum before the trip
        double vnext = (1 + l16 < g.tauMax) ? L.dY[1 + l16] : 0.0;
        for (int k0 = 1; k0 < g.tauMax; k0 += 16) {
            const double v = vnext;
            { const int kn = k0 + 16 + l16; vnext = (kn < g.tauMax) ? L.dY[kn] : 0.0; }
            VP_FMAC_BCAST(cap, v, M[0], 0);   VP_FMAC_BCAST(cap, v, M[1], 1);   VP_FMAC_BCAST(cap, v, M[2], 2);   VP_FMAC_BCAST(cap, v, M[3], 3);
            VP_FMAC_BCAST(cap, v, M[4], 4);   VP_FMAC_BCAST(cap, v, M[5], 5);   VP_FMAC_BCAST(cap, v, M[6], 6);   VP_FMAC_BCAST(cap, v, M[7], 7);
            VP_FMAC_BCAST(cap, v, M[8], 8);   VP_FMAC_BCAST(cap, v, M[9], 9);   VP_FMAC_BCAST(cap, v, M[10], 10); VP_FMAC_BCAST(cap, v, M[11], 11);
            VP_FMAC_BCAST(cap, v, M[12], 12); VP_FMAC_BCAST(cap, v, M[13], 13); VP_FMAC_BCAST(cap, v, M[14], 14); VP_FMAC_BCAST(cap, v, M[15], 15);
            if (k0 + l16 < g.tauMax) L.cum[k0 + l16] = cap;
            double nxt = zero * zero;                                         // +0.0 in a fresh register
            asm volatile("s_nop 1" : "+v"(cap), "+v"(nxt));                   // VALU write -> DPP read
            VP_FMAC_BCAST(nxt, cap, one, 15);                                 // lane 15 of the row: the sum so far
            cap = nxt;
        }
        if (lane == 0) { L.dY[0] = 1.0; L.dY[g.tauMax] = 0.0; }               // :395, guard slot (see oracle)
}

// Normalise, first lag under the tolerance, walk down to the local minimum (PitchProcess.cpp:395-403, 431-440): sets
// st->pitch / st->period.  cert = 0: dY holds the reference's sums, every comparison is what it is.  cert != 0: dY
// holds the cross-correlation form's sums, which differ from the reference's by at most
//     Delta = 2^-39 * (sum of w_j^2 over the F + tauMax samples)          (xcA; derivation in DESIGN.md section 4.1)
// in absolute value; every comparison the decision rests on is then CERTIFIED -- its two sides must differ by more
// than the error either side can carry -- and if one is not (or cert == 2, the test hook) the function returns false
// without touching the state: the caller recomputes the frame in the reference's arithmetic.
__device__ __forceinline__ bool yin_pick(const VpGeom &g, const VpDev &d, const PitchLds &L, lds_state *st, int tid, int nt, int cert)
{
    const double delta = cert ? 1.8189894035458565e-12 * L.xcA[0] : 0.0;    // 2^-39 * window energy
    for (int k = 1 + tid; k < g.tauMax; k += nt) {
        const double cm = L.cum[k], q = (double)k / cm, v = L.dY[k] * q;
        L.dY[k] = v;
        if (cert) {
            // |d'~_k - d'_k| <= Delta q (1 + d') + d' (k + 4) u,  q = k / cum_k;  four times that, kept where cum_k was
            const double dp = fabs(v);
            double e = 4.0 * (delta * q * (1.0 + dp) + dp * (double)(2 * k + 8) * 1.1102230246251565e-16);
            if (!(cm > 0.0) || !(e < 1.0)) e = 1e300;                         // silence, or not finite: nothing is certain
            L.cum[k] = e;
        }
    }
    if (cert && tid == 0) L.cum[g.tauMax] = 0.0;                              // the guard slot dY[tauMax] = 0 is exact
    __syncthreads();
    STAMP(d, 13);
    for (int k = g.tau0 + tid; k < g.tauMax; k += nt) {                    // first tau with d < tol (:431-433)
        const double v = L.dY[k];
        if (v < g.yinTol) atomicMin(&L.ishare[0], k);
        if (cert && !(fabs(v - g.yinTol) > L.cum[k])) L.ishare[1] = 1;       // too close to call
    }
    __syncthreads();
    if (tid == 0) {
        int tau = L.ishare[0];
        bool sure = !cert || (L.ishare[1] == 0 && cert != 2 && L.xcA[0] > 0.0);
        if (sure && tau < g.tauMax) {
            while (true) {                                                   // :435-440
                const double a = L.dY[tau + 1], b = L.dY[tau];
                if (cert && !(fabs(a - b) > L.cum[tau + 1] + L.cum[tau])) { sure = false; break; }
                if (!(a < b)) break;
                tau += 1;
                if (tau + 1 >= g.tauMax) break;
            }
        }
        if (sure && tau < g.tauMax) {
            if (tau >= g.tauMax) atomicAdd(&d.ub[3], 1ULL);
            st->pitch = g.fs / tau;
            st->period = tau;
        }
        L.ishare[1] = sure ? 0 : 1;
    }
    __syncthreads();
    return L.ishare[1] == 0;
}

// First half of PitchProcess::processChunkCont (PitchProcess.cpp:253-259): residual of the new samples.
// Returns true when the chunk has work (analysis marks exist); the caller then runs the shared
// tail psola -> filterIIR -> fillOutputBuffer (:262-268).
__device__ __forceinline__ bool pitch_chunk_cont_pre(const VpGeom &g, const VpCall &c, const VpDev &d, const PitchLds &L,
                                                     int nChunk, int pS, int s, bool noBarrier)
{
    if (L.st->nAn == 0) return false;
    // by wave 1 -- wave 0 goes straight on to PSOLA's grain table when `noBarrier` (nothing there reads the residual;
    // the barrier behind the table is the one the second pass needs anyway)
    const int tid = vp_tid(), w1 = (blockDim.x >= 2 * WAVE) ? WAVE : 0;
    if (tid >= w1 && tid < w1 + WAVE) fir_cont_wave(g, L, (const lds_f64 *)L.xs, nChunk);
    if (!noBarrier) __syncthreads();
    STAMP(d, 10);
    return true;
}

// PitchProcess::processChunkStart (PitchProcess.cpp:203-236) up to the residual; returns
// 0: gate closed (no output at all), 1: output only (no analysis marks: yFrame is zero), 2: full tail.
// true when the YIN phase leaves wave 0 free (two-lags-per-lane form on at most six waves)
__device__ __forceinline__ bool pitch_can_overlap(const VpGeom &g)
{
    return (g.C & 1) == 0 && ((((g.tauMax + 1) >> 1) + WAVE - 1) / WAVE) <= 6;
}

// pendingCont >= 0: the last chunk of the PREVIOUS frame (same step, PitchProcess.cpp:173-175) has had
// its residual and PSOLA done but not yet its IIR + output; wave 0 runs them here, next to the new
// frame's YIN on the other waves (they touch disjoint data: the old frame's outEFrame/yFrame and
// coefficients versus xs/yinTemp), and the new frame's buffers are zeroed only afterwards.
template <bool LITE, bool FAST, bool FFT, bool COMMON>
__device__ __forceinline__ int pitch_chunk_start_pre(const VpGeom &g, const VpCall &c, const VpDev &d, const PitchLds &L,
                                                     int pS, int s, int pendingCont, bool &hValid, int &xcGenCtr)
{
    lds_state *st = L.st;
    const int xcGen = ++xcGenCtr;                     // this Start's value of the prefix-sum flag (ishare[3])
    const int tid = vp_tid(), nt = blockDim.x;
    if (!d.gate[s * 2 + 0]) {                                               // :208-214
        if (pendingCont >= 0) {
            if (tid < WAVE) { pitch_iir_wave<LITE, FAST, COMMON>(g, L, pendingCont, hValid); pitch_fill_output_wave(g, c, d, L, pendingCont, pS, s); }
        }
        __syncthreads();
        if (tid == 0) { st->nAn = 0; st->prevPitch = 0; st->gateOpen = 0; }
        __syncthreads();
        return 0;
    }
    if (tid == 0) {                                                          // yin() state roll, :415-425
        st->gateOpen = 1;
        st->prevPeriod = st->period;
        st->prevPitch = st->pitch;
        if (st->pitch > 1) { st->prevVoicedPeriod = st->period; st->prevVoicedPitch = st->pitch; }
        st->pitch = 0; st->period = 0;
        L.ishare[0] = INT_MAX;
        L.ishare[1] = 0;                                                     // "a comparison was too close to call" (yin_pick)
    }
    // LPC ahead of the pitch decisions (see below): needs the time-domain autocorrelation on one wavefront
    const bool yinFft = FFT && c.yinFft;              // (only the *_fft builds carry that path)
    const bool specLpc = COMMON || (!yinFft && g.orderPitch < WAVE && nt >= 8 * WAVE);
    // VP_YIN_XCORR: cross-correlation form of the difference function, certified (yin_pick) with the reference's
    // arithmetic as the fallback; needs the two-lags-per-lane layout on waves 1..4 and a free wave 5
    const int yNPairs = (g.tauMax + 1) >> 1, yWaves = (yNPairs + WAVE - 1) / WAVE;
    const int xcCert = (c.yinCert != 0 && (COMMON || (!yinFft && (g.C & 1) == 0 && yWaves <= 4 && nt == 8 * WAVE))) ? c.yinCert : 0;
    const bool levLate = xcCert != 0;                 // Levinson-Durbin at the top of the marks phase instead of beside the running sum
    const int acM = min(nt - 1 - tid, g.orderPitch);                         // the last wavefront's lag per lane
    // how much of the sum runs beside YIN (the cross-correlation form of YIN is shorter: less fits beside it)
    const int acSplit = max(0, min((g.F * (c.yinCert ? VP_XC_ACSPLIT : 15) / 16) & ~15, (g.F - g.orderPitch) & ~15));   // (frames shorter than the order: nothing here)
    double acSum = 0.0;
    // computeYinTemp (PitchProcess.cpp:350-403): every lag is its own left-to-right sum over i.
    STAMPW_BEGIN();
    {
        const int base = g.toKeep - g.tauMax;
        if (!COMMON && pendingCont >= 0 && (!pitch_can_overlap(g) || yinFft)) {      // no free wave: finish the old frame first
            if (tid < WAVE) { pitch_iir_wave<LITE, FAST, COMMON>(g, L, pendingCont, hValid); pitch_fill_output_wave(g, c, d, L, pendingCont, pS, s); }
            __syncthreads();
            pendingCont = -1;
        }
        if (yinFft) {
            // VP_YIN_FFT (accelerator, not bit-exact): with a = frame (F samples, zero padded) and
            // b = the window of F + tauMax samples the difference function reads,
            //   d[k] = sum a_i^2 + sum_{j=k}^{k+F-1} b_j^2 - 2 (a (x) b)[k],
            // and with f = the frame itself (the last F samples of b): r_lpc[m] = (f (x) f)[m] / F.
            // Forward FFTs of f and of z = a + i b, then ONE inverse FFT of conj(A) B + i |Ff|^2: the two
            // real correlations ride in the real and imaginary parts.
            const int M = 1 << g.fftLog;
            lds_f64 *zr = L.fft, *zi = L.fft + M;
            lds_f64 *T = L.oE;                                               // |Ff|^2, M <= 2F doubles (outEFrame + yFrame: free here)
            const lds_f64 *w = L.xs + base;
            const int nb = g.F + g.tauMax;
            lds_f64 *twr = zi + M, *twi = twr + (M >> 1);                     // twiddles staged in LDS once per frame
            for (int j = tid; j < (M >> 1); j += nt) { twr[j] = d.twRe[j]; twi[j] = d.twIm[j]; }
            for (int j = tid; j < M; j += nt) { zr[j] = (j < g.F) ? w[g.tauMax + j] : 0.0; zi[j] = 0.0; }
            __syncthreads();
            fft_forward_dif(zr, zi, g.fftLog, (const lds_f64 *)twr, (const lds_f64 *)twi);
            for (int j = tid; j < M; j += nt) T[j] = zr[j] * zr[j] + zi[j] * zi[j];    // same (bit-reversed) positions as below
            __syncthreads();
            for (int j = tid; j < M; j += nt) { zr[j] = (j < g.F) ? w[j] : 0.0; zi[j] = (j < nb) ? w[j] : 0.0; }
            // exclusive prefix sums of b_j^2 (three entries per thread, wave + group scan) into eF scratch
            lds_f64 *P = L.eF;
            {
                const int per = (nb + nt - 1) / nt;
                const int j0 = tid * per;
                double loc = 0.0;
                for (int u = 0; u < per; u++) { const int j = j0 + u; if (j < nb) { const double v = w[j]; loc += v * v; } }
                double inc = loc;
                const int lane = tid & 63;
                for (int off = 1; off < WAVE; off <<= 1) { const double o = __shfl_up(inc, off, WAVE); if (lane >= off) inc += o; }
                if (lane == 63) L.dY[tid >> 6] = inc;                       // wave totals (dY is free until the scan is read)
                __syncthreads();
                double wbase = 0.0;
                for (int q = 0; q < (tid >> 6); q++) wbase += L.dY[q];
                double run = wbase + inc - loc;                              // exclusive prefix of this thread
                for (int u = 0; u < per; u++) { const int j = j0 + u; if (j <= nb) { P[j] = run; if (j < nb) { const double v = w[j]; run += v * v; } } }
                if (tid == nt - 1 && j0 + per <= nb) P[nb] = run;
            }
            __syncthreads();
            fft_forward_dif(zr, zi, g.fftLog, (const lds_f64 *)twr, (const lds_f64 *)twi);
            for (int k = tid; k <= (M >> 1); k += nt) {                      // spectra in bit-reversed positions
                const int pk = bitrev(k, g.fftLog), pm = bitrev((M - k) & (M - 1), g.fftLog);
                const double zkr = zr[pk], zki = zi[pk], zmr = zr[pm], zmi = zi[pm];
                const double Ar = 0.5 * (zkr + zmr), Ai = 0.5 * (zki - zmi);
                const double Br = 0.5 * (zki + zmi), Bi = -0.5 * (zkr - zmr);
                const double c1r = Ar * Br + Ai * Bi, c1i = Ar * Bi - Ai * Br;      // conj(A) B
                const double c2 = T[pk];                                           // |Ff|^2 (== T[pm])
                zr[pk] = c1r; zi[pk] = c1i + c2;
                zr[pm] = c1r; zi[pm] = c2 - c1i;
            }
            __syncthreads();
            fft_inverse_dit(zr, zi, g.fftLog, (const lds_f64 *)twr, (const lds_f64 *)twi);
            const double invM = 1.0 / (double)M;
            const double E0 = P[g.F] - P[0];
            for (int k = tid; k < g.tauMax; k += nt) L.dY[k] = E0 + (P[k + g.F] - P[k]) - 2.0 * (zr[k] * invM);
            for (int m = tid; m <= g.orderPitch; m += nt) L.r[m] = (zi[m] * invM) / (double)g.F;
        } else if (COMMON || (g.C & 1) == 0) {
            // TWO adjacent lags per lane (k = 2l, 2l+1): the lane slides one window of samples past
            // x[i], so each element costs one new LDS value for two lags, the two accumulation chains
            // interleave, and with the window base made even (see xsAll) every read is an aligned
            // ds_read_b128.  (The one-lag-per-lane form below spent twice the LDS cycles, on
            // ds_read2_b64 at half the LDS rate, and was LDS-bound at 30 us per frame.)  The eight
            // elements of a trip are done as 16 differences, 16 squares, 16 ordered adds: left to
            // the compiler each square and add directly followed its producer and paid the
            // dependent-issue stall (17 ns per element instead of 14).
            typedef double d2 __attribute__((ext_vector_type(2)));
            typedef __attribute__((address_space(3))) d2 lds_d2;
            const int nPairs = (g.tauMax + 1) >> 1;
            const int wavesY = (nPairs + WAVE - 1) / WAVE;
            // waves 1..wavesY when that leaves wave 0 free for the previous frame's pending chunk
            const int yw0 = (COMMON || wavesY <= 6) ? 1 : 0;
            if (yw0 == 1 && pendingCont >= 0 && tid < WAVE) {
                pitch_iir_wave<LITE, FAST, COMMON>(g, L, pendingCont, hValid);
                pitch_fill_output_wave(g, c, d, L, pendingCont, pS, s);
            }
#ifdef VP_DIAG_NO_YIN
            if (false)
#endif
            const bool xc = xcCert != 0;
            if (xc && (tid >> 6) == 5) {
                // cross-correlation form (see below): exclusive prefix sums P[j] of w_j^2 over the F + tauMax samples
                // the difference function reads, by the otherwise idle wave 5, into eFrame (free until the barrier);
                // the YIN waves pick them up through the flag when their own sums are done
                const int nb = g.F + g.tauMax, lane = tid & 63, per = (nb + WAVE - 1) / WAVE, j0 = lane * per;
                const lds_f64 *w = L.xs + base;
                lds_f64 *P = L.eF;
                double loc = 0.0;
                for (int u = 0; u < per; u++) { const int j = j0 + u; if (j < nb) { const double v = w[j]; loc += v * v; } }
                double inc = loc;
                for (int off = 1; off < WAVE; off <<= 1) { const double o = __shfl_up(inc, off, WAVE); if (lane >= off) inc += o; }
                double run = inc - loc;
                for (int u = 0; u < per; u++) { const int j = j0 + u; if (j <= nb) { P[j] = run; if (j < nb) { const double v = w[j]; run += v * v; } } }
                __threadfence_block();
                if (lane == 0) __hip_atomic_store(&L.ishare[3], xcGen, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
            if (xc && (tid >> 6) >= yw0 && (tid >> 6) < yw0 + wavesY) {
                // d[k] = sum_i w_i^2 + sum_i w_{i+k}^2 - 2 sum_i w_i w_{i+k}: two fused multiply-adds per element and lane
                // instead of six operations.  NOT the reference's arithmetic (differences ~1e-13 of the window energy).
                const int ty = tid - yw0 * WAVE;
                const int l = min(ty, nPairs - 1);
                const lds_f64 *xa = L.xs + base, *xw = L.xs + base + 2 * l;
                double accA = 0.0, accB = 0.0;
                // sixteen elements per trip: the wave-uniform factor w_i comes from ONE read (lane l holds w_{i0 + (l & 15)},
                // every 16-lane row the same) through the DPP row broadcast of v_fmac_f64, the lane's own factors w_{i+k}
                // stream through registers (aligned ds_read_b128, the next trip's in flight)
                const int F8 = g.F & ~15, l16 = tid & 15;
                double w0 = xw[0], w1 = xw[1];
                double E = xa[l16], En = 0.0;
                d2 v0[8], v1[8];
#define VP_XLOAD(V, I) _Pragma("unroll") for (int u = 0; u < 8; u++) V[u] = *(const lds_d2 *)(xw + (I) + 2 + 2 * u);
#define VP_XT(U, WA, WB) VP_FMAC_BCAST(accA, E, WA, U); VP_FMAC_BCAST(accB, E, WB, U);
#define VP_XCOMP(V) { VP_XT(0, w0, w1) VP_XT(1, w1, V[0].x) VP_XT(2, V[0].x, V[0].y) VP_XT(3, V[0].y, V[1].x) VP_XT(4, V[1].x, V[1].y) \
        VP_XT(5, V[1].y, V[2].x) VP_XT(6, V[2].x, V[2].y) VP_XT(7, V[2].y, V[3].x) VP_XT(8, V[3].x, V[3].y) VP_XT(9, V[3].y, V[4].x) \
        VP_XT(10, V[4].x, V[4].y) VP_XT(11, V[4].y, V[5].x) VP_XT(12, V[5].x, V[5].y) VP_XT(13, V[5].y, V[6].x) VP_XT(14, V[6].x, V[6].y) \
        VP_XT(15, V[6].y, V[7].x) w0 = V[7].x; w1 = V[7].y; }
                if (F8 > 0) { VP_XLOAD(v0, 0) }
                for (int i = 0; i < F8; i += 32) {
                    const bool more1 = i + 16 < F8;
                    if (more1) { VP_XLOAD(v1, i + 16) En = xa[i + 16 + l16]; }
                    VP_XCOMP(v0)
                    if (more1) {
                        E = En;
                        if (i + 32 < F8) { VP_XLOAD(v0, i + 32) En = xa[i + 32 + l16]; }
                        VP_XCOMP(v1)
                        E = En;
                    }
                }
#undef VP_XLOAD
#undef VP_XT
#undef VP_XCOMP
                for (int i = F8; i < g.F; i++) { accA = __builtin_fma(xa[i], xw[i], accA); accB = __builtin_fma(xa[i], xw[i + 1], accB); }
                int spin = 0;
                for (; spin < (1 << 22) &&
                     __hip_atomic_load(&L.ishare[3], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) != xcGen; spin++)
                    __builtin_amdgcn_s_sleep(1);         // wave 5 finished long ago; bounded so that a bug cannot hang the GPU
                if (spin == (1 << 22)) { L.ishare[1] = 1; if ((tid & 63) == 0) atomicAdd(&d.dbg[59], 1ULL); }   // never seen; falls back
                const lds_f64 *P = L.eF;
                const double E0 = P[g.F] - P[0];
                if (ty == 0) L.xcA[0] = P[g.F + g.tauMax];
                if (ty < nPairs) {
                    L.dY[2 * l] = E0 + (P[2 * l + g.F] - P[2 * l]) - 2.0 * accA;
                    if (2 * l + 1 < g.tauMax) L.dY[2 * l + 1] = E0 + (P[2 * l + 1 + g.F] - P[2 * l + 1]) - 2.0 * accB;
                }
            } else
            if (!xc) yin2_exact_waves(g, L, base, tid, yw0, wavesY, nPairs);
        } else {
#ifdef VP_DIAG_NO_YIN
        if (false)
#endif
        // whole wavefronts only: a wave whose lags run out computes the last lag again in its spare
        // lanes (no store) -- loops executed under a partial EXEC mask run markedly slower on this
        // chip and slow the other waves of the group down with them (tools/ubench_iir.hip, and a
        // 2x drop of this phase when the mask was made full)
        for (int kk = tid; kk < ((g.tauMax + WAVE - 1) & ~(WAVE - 1)); kk += nt) {
            const int k = min(kk, g.tauMax - 1);
            double accv = 0.0;
            const lds_f64 *xa = L.xs + base, *xb = L.xs + base + k;
            // eight elements per trip, the next trip's LDS reads issued before this trip's arithmetic
            // (the sum itself stays strictly left to right)
            const int F8 = g.F & ~7;
            double a0[8], b0[8], a1[8], b1[8];
#define VP_YLOAD(A, B, I) _Pragma("unroll") for (int u = 0; u < 8; u++) { A[u] = xa[(I) + u]; B[u] = xb[(I) + u]; }
// differences, then squares, then the (ordered) accumulation: a dependent fp64 op stalls until
// every VALU op issued before it has retired (tools/ubench_chain4.hip), so element-by-element
// sub -> mul -> add pays three such stalls per element, the batched order one per add.
#define VP_YCOMP(A, B) { double df_[8]; \
        _Pragma("unroll") for (int u = 0; u < 8; u++) df_[u] = A[u] - B[u]; \
        __builtin_amdgcn_sched_barrier(0); \
        _Pragma("unroll") for (int u = 0; u < 8; u++) df_[u] = df_[u] * df_[u]; \
        __builtin_amdgcn_sched_barrier(0); \
        _Pragma("unroll") for (int u = 0; u < 8; u++) accv += df_[u]; \
        __builtin_amdgcn_sched_barrier(0); }
            if (F8 > 0) { VP_YLOAD(a0, b0, 0) }
            for (int i = 0; i < F8; i += 16) {
                const bool more1 = i + 8 < F8;
                if (more1) { VP_YLOAD(a1, b1, i + 8) }
                VP_YCOMP(a0, b0)
                if (more1) {
                    if (i + 16 < F8) { VP_YLOAD(a0, b0, i + 16) }
                    VP_YCOMP(a1, b1)
                }
            }
#undef VP_YLOAD
#undef VP_YCOMP
            for (int i = F8; i < g.F; i++) {
                double df = xa[i] - xb[i];
                accv += df * df;
            }
            if (kk < g.tauMax) L.dY[k] = accv;
        }
        }
    }
    // LPC autocorrelation of the frame (rectangular window: the products with 1.0 are exact),
    // LPC.cpp:44-97; independent of the pitch decisions, so it is done in the same phase.
    // Its result is only used when analysis marks exist (:230-233).
    {
        const int order = g.orderPitch;
        const lds_f64 *x = L.xs + g.toKeep;
        STAMPL_BEGIN();
        if (specLpc) {
            // an order below 64 has all its lags on the LAST wavefront (lane -> lag, spare lanes redo lag
            // `order`).  It shares its SIMD with a YIN wavefront, so only the first stretch of the sum is done
            // here; the chain is carried in a register across the barrier and finished, followed by
            // Levinson-Durbin, while wave 0 runs the cumulative sum below.
            if (tid >= nt - WAVE) acSum = autocorr_stretch(x, acM, 0, acSplit, 0.0);
        } else
#ifdef VP_DIAG_NO_AUTOCORR
        if (false)
#endif
        if (!yinFft)
        // highest threads (they have no YIN lag), again whole wavefronts: spare lanes redo lag `order`
        for (int m0 = nt - 1 - tid; (m0 & ~(WAVE - 1)) <= order && m0 >= 0; m0 += nt) {
            const int m = min(m0, order);
            double sum = 0.0;
            const int cnt = max(g.F - m, 0), c8 = cnt & ~7;
            const lds_f64 *xm = x + m;
            double a0[8], b0[8], a1[8], b1[8];              // same software pipeline as the YIN loop
#define VP_ALOAD(A, B, I) _Pragma("unroll") for (int u = 0; u < 8; u++) { A[u] = x[(I) + u]; B[u] = xm[(I) + u]; }
#define VP_ACOMP(A, B) _Pragma("unroll") for (int u = 0; u < 8; u++) { sum += A[u] * B[u]; }
            if (c8 > 0) { VP_ALOAD(a0, b0, 0) }
            for (int n = 0; n < c8; n += 16) {
                const bool more1 = n + 8 < c8;
                if (more1) { VP_ALOAD(a1, b1, n + 8) }
                VP_ACOMP(a0, b0)
                if (more1) {
                    if (n + 16 < c8) { VP_ALOAD(a0, b0, n + 16) }
                    VP_ACOMP(a1, b1)
                }
            }
#undef VP_ALOAD
#undef VP_ACOMP
            for (int n = c8; n < cnt; n++) sum += x[n] * xm[n];
            if (m0 <= order) L.r[m] = sum / (double)g.F;
        }
        STAMPL(26);
    }
    STAMPW(40);
    __syncthreads();
    // :216-218 (after the old frame's last chunk is out).  With specLpc the residual of the whole frame is
    // written straight into eFrame further down (every entry below toKeep + F), so only the tail is
    // zeroed here; a frame without analysis marks zeroes the rest again.
    for (int i = (specLpc ? g.toKeep + g.F : 0) + tid; i < g.eLen; i += nt) L.eF[i] = 0.0;
    for (int i = tid; i < g.F; i += nt) { L.oE[i] = 0.0; L.yF[i] = 0.0; }
    STAMP(d, 1);
    if (specLpc && tid >= nt - WAVE) {
        // the rest of the autocorrelation (uniform stretch, then the lag-dependent tail), then the LPC
        // itself (it needs nothing from the pitch decisions) into a scratch vector that is adopted below
        // if analysis marks exist
        STAMPL_BEGIN();
        const lds_f64 *x = L.xs + g.toKeep, *xm = x + acM;
        const int nU = max(0, (g.F - g.orderPitch) & ~7);
        acSum = autocorr_stretch(x, acM, acSplit, nU, acSum);
        for (int n = nU; n < g.F - acM; n++) acSum += x[n] * xm[n];
        L.r[acM] = acSum / (double)g.F;                                       // spare lanes: identical stores
        STAMPL(26);
        // Levinson-Durbin: here, beside wave 0's running sum, when YIN runs in the reference's arithmetic (this
        // wavefront then has only a sixteenth of the autocorrelation left to do in this phase); with the
        // cross-correlation YIN more of the autocorrelation lands here and the recursion would pace the phase, so it
        // follows at the top of the marks phase instead (levLate)
        if (!levLate) {
            const int order = g.orderPitch;
            const bool z = (COMMON || order < 16) ? levinson_row16(L.r, L.aPrev, order, VP_ORDER_MAX + 1, g.levEps)
                           : (!LITE && order >= VP_LEV_SCALAR_MIN && order <= 48) ? levinson_scalar<48>((const lds_f64 *)L.r, L.aPrev, order, VP_ORDER_MAX + 1, g.levEps)
                                        : levinson_wave(L.r, L.aPrev, order, VP_ORDER_MAX + 1, g.levEps, L.qtab);   // qtab: rebuilt per frame later
            if (tid == nt - 1) { L.ishare[2] = z ? 1 : 0; *L.lpcFlag = xcGen; }   // (a barrier follows before the FIR waves look)
            STAMPL(27);
        }
    }
    if (tid < WAVE) yin_cumsum_wave(g, L, tid);                            // :395-402 running sum tmp += yinTemp[k], in order
    __syncthreads();
    STAMP(d, 12);
    if (!yin_pick(g, d, L, st, tid, nt, xcCert)) {
        // a comparison of the certified form was too close to call (about once in 1e9 frames; always with the
        // diagnostic mode 3): the frame again, in the reference's arithmetic
        // (barrier first: yin_pick's verdict is read from ishare[1] by every thread; a wavefront that got there after the
        // reset below would take the frame for certified, skip this branch and run one barrier out of step with the others)
        __syncthreads();
        if (tid == 0) { L.ishare[0] = INT_MAX; L.ishare[1] = 0; atomicAdd(&d.dbg[63], 1ULL); }
        yin2_exact_waves(g, L, g.toKeep - g.tauMax, tid, 1, yWaves, yNPairs);
        __syncthreads();
        if (tid < WAVE) yin_cumsum_wave(g, L, tid);
        __syncthreads();
        yin_pick(g, d, L, st, tid, nt, 0);
    } else if (xcCert && tid == 0)
        atomicAdd(&d.dbg[62], 1ULL);
    __syncthreads();
    STAMP(d, 2);
    if (tid < WAVE) {                                                        // wave 0, all lanes redundantly (full EXEC)
        pitch_marks(g, L, d.ub);
        STAMP(d, 3);
        place_st_marks(g, c, d, st);
    } else if (specLpc && tid < nt - WAVE) {
        // meanwhile, on the middle wavefronts: filterFIR(-toKeep, toKeep+F, 0) :280-302 with the new
        // coefficients (speculative: only used if analysis marks exist); the wavefront that shares wave 0's
        // SIMD (wave 4) stays out of the serial code's way
        const int wv = tid >> 6, nw = nt >> 6;
        if (nw < 8 || wv != 4) {
            const int rank = (nw < 8 || wv < 4) ? wv - 1 : wv - 2, nWork = (nw < 8) ? nw - 2 : nw - 3;
            const int order = g.orderPitch;
            const lds_f64 *a = L.aPrev;
            // the coefficients come from the last wavefront, a few microseconds into this phase (flag = this Start's
            // generation number; bounded wait so that a bug cannot hang the GPU -- a timeout is counted and would show
            // as a parity failure, never as a hang)
            int spin = 0;
            for (; spin < (1 << 22) && __hip_atomic_load(L.lpcFlag, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) != xcGen; spin++)
                __builtin_amdgcn_s_sleep(1);
            if (spin == (1 << 22) && (tid & 63) == 0) atomicAdd(&d.dbg[60], 1ULL);
            for (int j = 4 * (rank * WAVE + (tid & 63)); j < g.toKeep + g.F; j += 4 * nWork * WAVE)
                fir4((const lds_f64 *)L.xs, a, order, j, g.toKeep + g.F, L.eF);
        }
    } else if (tid >= nt - WAVE && (COMMON || g.orderPitch < WAVE)) {
        // meanwhile, on the last wavefront (FFT mode: Levinson-Durbin first, the autocorrelation came late) ...
        if (!specLpc || levLate) {
            STAMPL_BEGIN();
            const bool z = (COMMON || g.orderPitch < 16) ? levinson_row16(L.r, L.aPrev, g.orderPitch, VP_ORDER_MAX + 1, g.levEps)
                           : (!LITE && g.orderPitch >= VP_LEV_SCALAR_MIN && g.orderPitch <= 48) ? levinson_scalar<48>((const lds_f64 *)L.r, L.aPrev, g.orderPitch, VP_ORDER_MAX + 1, g.levEps)
                                               : levinson_wave(L.r, L.aPrev, g.orderPitch, VP_ORDER_MAX + 1, g.levEps, L.qtab);
            if (tid == nt - 1) L.ishare[2] = z ? 1 : 0;
            __threadfence_block();
            if (tid == nt - 1) __hip_atomic_store(L.lpcFlag, xcGen, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);   // the FIR waves may go
            STAMPL(27);
        }
        if (COMMON || (g.C & 63) == 0) {
            // ... the impulse response of the new 1/A(z) for the block-form IIR (cum[] is dead after
            // the normalisation; layout as in pitch_iir_wave)
            pitch_impulse_response<LITE, COMMON>(g, L, (const lds_f64 *)L.aPrev);
        }
    }
    __syncthreads();
    STAMP(d, 4);
    if (st->nAn != 0) {
        if (COMMON || g.orderPitch < WAVE) {                                 // coefficients are ready: adopt them (:233)
            const int ncopy = L.ishare[2] ? VP_ORDER_MAX + 1 : g.orderPitch + 1;
            for (int i = tid; i < ncopy; i += nt) st->a[i] = L.aPrev[i];
        } else if (tid < WAVE)
            levinson_wave(L.r, (lds_f64 *)st->a, g.orderPitch, VP_ORDER_MAX + 1, g.levEps);
        if (specLpc) {                                                        // the residual is already in eFrame
            if (tid == 0) st->stMarkIdx = 0;
            __syncthreads();
            STAMP(d, 5);
            return 2;
        }
        __syncthreads();
        STAMP(d, 5);
        const int order = g.orderPitch;
        const lds_f64 *a = st->a;
        for (int j = tid; j < g.toKeep + g.F; j += nt) {                    // filterFIR(-toKeep, toKeep+F, 0) :280-302
            double e = a[0] * L.xs[j];
            int kmax = min(order, j);
            for (int k = 1; k <= kmax; k++) e += L.xs[j - k] * a[k];
            L.eF[j] = e;
        }
        if (tid == 0) st->stMarkIdx = 0;
        __syncthreads();
        STAMP(d, 6);
        return 2;
    }
    if (specLpc) {                                                            // no analysis marks: the frame stays as :216-218 left it
        for (int i = tid; i < g.toKeep + g.F; i += nt) L.eF[i] = 0.0;
    }
    return 1;
}

template <bool LITE, bool FAST, bool MULTI, bool FFT, bool COMMON>
__device__ __forceinline__ void pitch_kernel_body(const VpGeom &g, const VpCall &c, const VpDev &d, const float *__restrict__ in0,
                                                  float *__restrict__ out0, double *smem)
{
    const int s = vp_stream(d), tid = threadIdx.x, nt = blockDim.x;
#ifdef VP_STAMPS
    const unsigned long long wgT0 = wall_clock64();
#endif
    // the block in hand (vp_process_blocks_device runs several per launch): its offset in samples from the first one
    // -- the ring and output counters of `c` describe the first block, later ones are that much further on --, its
    // chunk-step count and its I/O slabs
    int boff = 0, nSteps = c.nSteps;
    const float *in = in0;
    float *out = out0;
    VP_POISON(smem, c.ldsBytes);
    if (c.fuseIngest) ingest_gate_block(g, c, d, in);
    PitchLds L;
    // voice window of g.xsSteps consecutive chunk steps; shifted by one double when needed so that
    // xs[toKeep - tauMax] (where the YIN window starts) is 16-byte aligned
    lds_f64 *xsAll = (lds_f64 *)smem + ((g.toKeep - g.tauMax) & 1);
    L.xs = xsAll;
    L.eF = (lds_f64 *)smem + (g.toKeep + g.F + (g.xsSteps - 1) * g.C + 4);   // +1 alignment pad, +2 read-ahead slack
    L.oE = L.eF + g.eLen;
    L.yF = L.oE + g.F;
    L.dY = L.yF + g.F;
    L.cum = L.dY + vp_dy_len(g.tauMax);               // (both sized for what later phases park there, vp_common.h)
    L.r = L.cum + vp_cum_len(g.tauMax);
    L.aPrev = L.r + (VP_ORDER_MAX + 1);
    L.qtab = L.aPrev + (VP_ORDER_MAX + 1);            // [2 tauMax + 2] PSOLA quotient table (also Levinson scratch)
    L.qtab += (int)((L.qtab - (lds_f64 *)smem) & 1);  // keep it 16-byte aligned
    L.htab = L.qtab + (2 * g.tauMax + 2);             // [2 tauMax + 2] the frame's Hann(2T+1) window, staged from the global table
    L.part = (lds_minidx *)(L.htab + (2 * g.tauMax + 2));
    L.st = (lds_state *)(L.part + 8);
    L.ishare = (int *)((char *)smem + ((size_t)((lds_i32 *)(L.st + 1) - (lds_i32 *)smem)) * sizeof(int));
    L.xcA = (lds_f64 *)(L.st + 1) + 2;                // behind ishare's 16 bytes, in front of the FFT arrays
    L.lpcFlag = (int *)((char *)smem + ((size_t)((lds_i32 *)((lds_f64 *)(L.st + 1) + 3) - (lds_i32 *)smem)) * sizeof(int));
    L.psFlag = L.lpcFlag + 1;
    L.fft = (lds_f64 *)(L.st + 1) + 8;                // [2 << fftLog] only when launched with the FFT extension

    // Everything the block needs from global memory is requested in ONE go (tracker state, the frame in
    // flight, the voice window of the first steps): three dependent round trips cost three memory latencies.
    const float *vr = d.voiceRing + (size_t)s * g.inSize;
    int pS = c.pStart, nChunk = c.nChunk0;
    auto load_xs = [&](int step, int tid_) {
        // voice samples idx in [pS - toKeep, pS + F) of this step, widened to double.  Consecutive steps
        // overlap by all but C samples, so the ring is read once per g.xsSteps steps (once per block
        // when LDS allows) and the step's window is just an offset into that span.
        const int nst = min(g.xsSteps, nSteps - step);
        const int span = g.toKeep + g.F + (nst - 1) * g.C;
        int p0 = ring_pos(c.currCounter, boff + pS - g.toKeep, g.inSize);
        for (int j = tid_; j < span; j += nt) {
            int pp = p0 + j;
            pp -= (pp >= g.inSize) ? g.inSize : 0;                  // span < inSize: one wrap at most
            xsAll[j] = (double)vr[pp];
        }
    };
    STAMP0(d);
    {   // state in
        const int *src = (const int *)(d.pitch + s);
        lds_i32 *dst = (lds_i32 *)L.st;
        for (int i = tid; i < (int)(sizeof(VpPitchState) / sizeof(int)); i += nt) dst[i] = src[i];
    }
#ifdef VP_DIAG_NO_FRAME_IO
    // DIAGNOSTIC (wrong results): the in-flight frame is neither read back nor written out -- what does that traffic cost?
    if (false) {
#else
    if (c.nChunk0 != 0) {
#endif
        // a frame may be in flight (it is if the state says nAn != 0; if not, nothing reads what is loaded
        // here): its residual, the not yet filtered part of outEFrame (chunks >= nChunk0) and the last
        // `order` outputs (the IIR's history) are all that later chunks can read
        const double *ge = d.eFrame + (size_t)s * g.eLen, *go = d.outEFrame + (size_t)s * g.F, *gy = d.yFrame + (size_t)s * g.F;
        const int done = c.nChunk0 * g.C;
        for (int i = tid; i < g.eLen; i += nt) L.eF[i] = ge[i];
        for (int i = done + tid; i < g.F; i += nt) L.oE[i] = go[i];
        for (int i = max(0, done - g.orderPitch) + tid; i < done; i += nt) L.yF[i] = gy[i];
        if (COMMON || ((g.C & 63) == 0 && g.orderPitch < WAVE)) {   // the frame's impulse response (block-form IIR)
            if (tid < WAVE) L.cum[VP_HPAD_OFF + tid] = 0.0;
            if (tid < (pitch_iir_hc<LITE, COMMON>(g) ? 2 * WAVE : WAVE)) L.cum[VP_HPAD_OFF + WAVE + tid] = d.hImp[(size_t)s * 2 * WAVE + tid];
        }
    }
    if (nSteps > 0) load_xs(0, tid);
    if (tid == 0) { L.ishare[3] = 0; *L.lpcFlag = 0; *L.psFlag = 0; }   // flags (generation counters): YIN prefix sums, LPC coefficients, grain table
    __syncthreads();
    const bool frameLive0 = (c.nChunk0 != 0) && (L.st->nAn != 0);
    const bool hValid0 = frameLive0 && (COMMON || ((g.C & 63) == 0 && g.orderPitch < WAVE));

    bool qValid = false, hValid = hValid0;
    int xcGenCtr = 0, psGen = 0;
    bool preDone = false;                             // the coming chunk's PSOLA has been done ahead (pitch_iir)
    // vp_process_blocks_device: several consecutive blocks in this launch.  The tracker state and the frame in flight
    // stay in LDS between them; per block only the input is ingested (rings, gate), the voice window staged and the
    // output emitted -- exactly what separate launches would do, minus their state round trips.
    // (a separate build, MULTI: the extra loop level costs the single-block kernels 3 % through register allocation)
    for (int blk = 0; blk < (MULTI ? c.nBlocks : 1); blk++) {
    if (MULTI && blk > 0) {
        // MyBuffer.cpp:129-132 and PitchProcess.cpp:166-196, as the host does between calls
        boff += g.N;
        pS -= g.N;
        nSteps = (pS < g.N) ? (g.N - pS + g.C - 1) / g.C : 0;
        in += (size_t)g.S * (c.inMono ? 1 : 3) * g.N;
        out += (size_t)g.S * (c.inplace ? 3 : 2) * g.N;
        __syncthreads();                              // the previous block's emit has read what the ingest overwrites
        ingest_gate_block(g, c, d, in, boff);
        if (nSteps > 0) load_xs(0, tid);
        __syncthreads();
    }
    for (int step = 0; step < nSteps; step++) {
        const int tid = vp_tid();
        if (step > 0 && step % g.xsSteps == 0) {
            load_xs(step, tid);
            __syncthreads();
        }
        L.xs = xsAll + (step % g.xsSteps) * g.C;
        STAMP(d, 0);
        // PitchProcess::process (:171-189): a step is [Cont of the running frame] then, when a new
        // frame starts here, [Start]; both feed the same tail psola -> filterIIR -> fillOutputBuffer.
        int pendingCont = -1;
        for (int sub = 0; sub < 2; sub++) {
            int mode;                       // 0 nothing, 1 output only, 2 psola + IIR + output
            int nC;
            if (sub == 0) {
                if (nChunk == 0) continue;
                nC = nChunk;
                if (preDone) mode = 2;         // residual, grain table and second pass were done beside the previous chunk's IIR
                else mode = pitch_chunk_cont_pre(g, c, d, L, nChunk, pS, s, qValid) ? 2 : 0;   // qValid: PSOLA starts with the grain table
                if (mode == 2 && nChunk == g.cpf - 1) {
                    // a new frame starts in this step: leave this chunk's IIR + output to wave 0 during
                    // the new frame's YIN phase (pitch_chunk_start_pre)
                    if (!preDone) psola(g, d, L, nC, pS, qValid);
                    preDone = false;
                    pendingCont = nC;
                    continue;
                }
            } else {
                if (nChunk == g.cpf - 1) nChunk = 0;
                if (nChunk != 0) break;
                nC = 0;
                mode = pitch_chunk_start_pre<LITE, FAST, FFT, COMMON>(g, c, d, L, boff + pS, s, pendingCont, hValid, xcGenCtr);   // pS: output position only
                hValid = (mode != 0) && (COMMON || ((g.C & 63) == 0 && g.orderPitch < WAVE));    // computed there for the new coefficients
            }
            if (sub == 1) qValid = false;                 // a new frame: new beta / period
            if (mode == 2) {
                if (!preDone) psola(g, d, L, nC, pS, qValid);
                preDone = false;
                // the frame's next chunk: handled by the next step of this block, window staged?
                // (and the grain table must fit inside the yinTemp scratch: at low sample rates it spills into cum[], where
                // the exact recursion keeps its history while it runs)
                const bool ahead = qValid && nC + 1 <= g.cpf - 1 && step + 1 < nSteps && (step + 1) / g.xsSteps == step / g.xsSteps &&
                                   nt >= 2 * WAVE && g.tauMax + 1 >= 2 * VP_MARKS + (5 * VP_MARKS + 1) / 2;
                pitch_iir<LITE, FAST, COMMON>(g, d, L, nC, hValid, ahead, (const lds_f64 *)(xsAll + ((step + 1) % g.xsSteps) * g.C), pS + g.C, psGen);
                preDone = ahead;
            } else
                preDone = false;
            if (mode >= 1) pitch_fill_output(g, c, d, L, nC, boff + pS, s);
            __syncthreads();
        }
        nChunk += 1;
        __syncthreads();
        pS += g.C;
    }
    if (MULTI && blk + 1 < c.nBlocks) {               // not the last block: its output goes out now
        __syncthreads();
        emit_block(g, c, d, out, L.st, boff);
    }
    }

    {   // state out
        int *dst = (int *)(d.pitch + s);
        const lds_i32 *src = (const lds_i32 *)L.st;
        for (int i = tid; i < (int)(sizeof(VpPitchState) / sizeof(int)); i += nt) dst[i] = src[i];
    }
#ifdef VP_DIAG_NO_FRAME_IO
    if (false) {
#else
    if (nChunk != 0 && L.st->nAn != 0) {
#endif
        double *ge = d.eFrame + (size_t)s * g.eLen, *go = d.outEFrame + (size_t)s * g.F, *gy = d.yFrame + (size_t)s * g.F;
        const int done = nChunk * g.C;                       // same ranges as the load above
        for (int i = tid; i < g.eLen; i += nt) ge[i] = L.eF[i];
        for (int i = done + tid; i < g.F; i += nt) go[i] = L.oE[i];
        for (int i = max(0, done - g.orderPitch) + tid; i < done; i += nt) gy[i] = L.yF[i];
        if ((COMMON || ((g.C & 63) == 0 && g.orderPitch < WAVE)) && tid < (pitch_iir_hc<LITE, COMMON>(g) ? 2 * WAVE : WAVE))
            d.hImp[(size_t)s * 2 * WAVE + tid] = L.cum[VP_HPAD_OFF + WAVE + tid];
    }
    STAMP(d, 11);
    if (c.fuseEmit) {
        __syncthreads();
        emit_block(g, c, d, out, L.st, boff);
    }
#ifdef VP_STAMPS
    if (tid == 0) d.dbg[64 + s] += wall_clock64() - wgT0;
#endif
}

// Four builds of the one body: the IIR mode is a compile-time choice so that the exact recursion's big
// register-resident instantiations and the block form's 64 resident taps never meet in one register
// allocation (together they pushed kernel-invariant values into scratch, and every reload in a serial
// phase is a memory round trip), and each build carries half the code.
#if VP_TU_HAS(2)
__global__ __launch_bounds__(512) void vp_k_pitch(VpGeom g, VpCall c, VpDev d, const float *__restrict__ in, float *__restrict__ out)
{
    extern __shared__ double smem[];
    pitch_kernel_body<false, false, false, false, false>(g, c, d, in, out, smem);
}
#endif

#if VP_TU_HAS(3)
__global__ __launch_bounds__(512) void vp_k_pitch_fast(VpGeom g, VpCall c, VpDev d, const float *__restrict__ in,
                                                       float *__restrict__ out)
{
    extern __shared__ double smem[];
    pitch_kernel_body<false, true, false, false, false>(g, c, d, in, out, smem);
}
#endif

// The COMMON-CASE builds of the two above: chunk a multiple of 64 samples, lpcPitch <= 15, tauMax <= 512 (any sample rate
// up to 51.2 kHz with the plugin's own geometry).  Every alternative the general builds keep for other geometries and
// orders (one-lag YIN, LDS-form block IIR, general Levinson-Durbin, the wave-0 fallback orders, ...) is compiled out:
// code a launch never executes still costs it (section 4.2.1).  The host picks them whenever the handle qualifies.
#if VP_TU_HAS(2)
__global__ __launch_bounds__(512) void vp_k_pitch_c(VpGeom g, VpCall c, VpDev d, const float *__restrict__ in, float *__restrict__ out)
{
    extern __shared__ double smem[];
    pitch_kernel_body<false, false, false, false, true>(g, c, d, in, out, smem);
}
#endif
#if VP_TU_HAS(3)
__global__ __launch_bounds__(512) void vp_k_pitch_fast_c(VpGeom g, VpCall c, VpDev d, const float *__restrict__ in,
                                                         float *__restrict__ out)
{
    extern __shared__ double smem[];
    pitch_kernel_body<false, true, false, false, true>(g, c, d, in, out, smem);
}
#endif

#if VP_TU_HAS(4)
__global__ __launch_bounds__(512) void vp_k_pitch_fast_multi_c(VpGeom g, VpCall c, VpDev d, const float *__restrict__ in,
                                                               float *__restrict__ out)
{
    extern __shared__ double smem[];
    pitch_kernel_body<false, true, true, false, true>(g, c, d, in, out, smem);
}
#endif

// vp_process_blocks_device: the same two, looping over c.nBlocks consecutive blocks (state stays in LDS between them)
#if VP_TU_HAS(4)
__global__ __launch_bounds__(512) void vp_k_pitch_multi(VpGeom g, VpCall c, VpDev d, const float *__restrict__ in,
                                                        float *__restrict__ out)
{
    extern __shared__ double smem[];
    pitch_kernel_body<false, false, true, false, false>(g, c, d, in, out, smem);
}
#endif

#if VP_TU_HAS(4)
__global__ __launch_bounds__(512) void vp_k_pitch_fast_multi(VpGeom g, VpCall c, VpDev d, const float *__restrict__ in,
                                                             float *__restrict__ out)
{
    extern __shared__ double smem[];
    pitch_kernel_body<false, true, true, false, false>(g, c, d, in, out, smem);
}
#endif

#if VP_TU_HAS(5)
// common-case build of the register-light FAST kernel (large batches)
__global__ __launch_bounds__(512, 4) void vp_k_pitch_lite_fast_c(VpGeom g, VpCall c, VpDev d, const float *__restrict__ in,
                                                                  float *__restrict__ out)
{
    extern __shared__ double smem[];
    pitch_kernel_body<true, true, false, false, true>(g, c, d, in, out, smem);
}
#endif

// VP_YIN_FFT (experimental accelerator, section 4.3) has builds of its own, so that the others do not carry its code
#if VP_TU_HAS(4)
__global__ __launch_bounds__(512) void vp_k_pitch_fft(VpGeom g, VpCall c, VpDev d, const float *__restrict__ in, float *__restrict__ out)
{
    extern __shared__ double smem[];
    pitch_kernel_body<false, false, false, true, false>(g, c, d, in, out, smem);
}

__global__ __launch_bounds__(512) void vp_k_pitch_fast_fft(VpGeom g, VpCall c, VpDev d, const float *__restrict__ in,
                                                           float *__restrict__ out)
{
    extern __shared__ double smem[];
    pitch_kernel_body<false, true, false, true, false>(g, c, d, in, out, smem);
}
#endif

// Register-light build of the same kernel (<= 128 VGPRs: two 512-thread workgroups per CU), selected by the
// host for large batches when the exact IIR needs no big register-resident instantiation (FAST mode, or
// lpcPitch <= 16).  With S >> 256 streams a second resident workgroup fills the first one's serial phases.
#if VP_TU_HAS(5)
__global__ __launch_bounds__(512, 4) void vp_k_pitch_lite(VpGeom g, VpCall c, VpDev d, const float *__restrict__ in,
                                                           float *__restrict__ out)
{
    extern __shared__ double smem[];
    pitch_kernel_body<true, false, false, false, false>(g, c, d, in, out, smem);
}
#endif

#if VP_TU_HAS(5)
__global__ __launch_bounds__(512, 4) void vp_k_pitch_lite_fast(VpGeom g, VpCall c, VpDev d, const float *__restrict__ in,
                                                                float *__restrict__ out)
{
    extern __shared__ double smem[];
    pitch_kernel_body<true, true, false, false, false>(g, c, d, in, out, smem);
}
#endif

// ------------------------------------------------------------------------------------------------
// K3: emit.  addDryVoice / addSynth (MyBuffer.cpp:309-448) + fillOutputBuffer + clearOutput
// (MyBuffer.cpp:113-133, 218-228).  out[ch] = float(((acc + dry) + synth_ch)); the consumed region of
// the accumulator is zeroed.
__device__ __forceinline__ void emit_block(const VpGeom &g, const VpCall &c, const VpDev &d, float *__restrict__ out,
                                           const lds_state *stl, int boff)
{
    const int s = vp_stream(d);
    const float *vr = d.voiceRing + (size_t)s * g.inSize;
    const float *sr0 = d.synthRing + (size_t)s * 2 * g.inSize;
    const float *sr1 = sr0 + g.inSize;
    double *acc = d.outAcc + (size_t)s * g.outSize;
    double *acc2 = d.outAcc2 ? d.outAcc2 + (size_t)s * g.outSize : nullptr;
    float *o = out + (size_t)s * (c.inplace ? 3 : 2) * g.N;
    // the stream's dry-path switches and gains: from the state the pitch kernel holds in LDS, else from HBM
    const int dryOn = stl ? stl->sp.dryOn : d.pitch[s].sp.dryOn, synthOn = stl ? stl->sp.synthOn : d.pitch[s].sp.synthOn;
    const double gainVoice = stl ? stl->sp.gainVoice : d.pitch[s].sp.gainVoice;
    const double gainSynth = stl ? stl->sp.gainSynth : d.pitch[s].sp.gainSynth;
    for (int i = threadIdx.x; i < g.N; i += blockDim.x) {
        int pos = (c.outCounter + boff + i) % g.outSize;
        int pin = (c.currCounter + boff + i) % g.inSize;
        double v = acc[pos];
        if (acc2) { v += acc2[pos]; acc2[pos] = 0.0; }                       // the pitch corrector's share, when it ran beside the vocoder
        if (dryOn) v += (double)vr[pin] * gainVoice;
        double l = v, r = v;
        if (synthOn) { l += (double)sr0[pin] * gainSynth; r += (double)sr1[pin] * gainSynth; }
        o[i] = (float)l;
        o[g.N + i] = (float)r;
        if (c.inplace) o[2 * g.N + i] = 0.0f;
        acc[pos] = 0.0;
    }
}

#if VP_TU_HAS(1)
__global__ __launch_bounds__(256) void vp_k_emit(VpGeom g, VpCall c, VpDev d, float *__restrict__ out)
{
    emit_block(g, c, d, out);
}
#endif

// ------------------------------------------------------------------------------------------------
// Standalone STFT round trip (NO reference counterpart -- the reference has no FFT, SURVEY.md section 0;
// this is the "Hann windowing, batched radix-2 FFT/iFFT, overlap-add" kernel BASELINE.json's
// north_star names, reported separately).  One workgroup per (stream, frame): sqrt-Hann analysis
// window, forward FFT, [identity spectral stage, optional magnitude dump], inverse FFT, sqrt-Hann
// synthesis window -> frame scratch; a second kernel overlap-adds in gather form (deterministic).
#if VP_TU_HAS(1)
__global__ __launch_bounds__(256) void vp_k_stft_frames(const float *__restrict__ in, float *__restrict__ frames,
                                                        float *__restrict__ mag, const double *__restrict__ win,
                                                        const double *__restrict__ twRe, const double *__restrict__ twIm,
                                                        int nSamples, int nFrames, int logF, int hop)
{
    extern __shared__ double smem[];
    const int F = 1 << logF, tid = threadIdx.x, nt = blockDim.x;
    const int s = blockIdx.y, f = blockIdx.x;
    lds_f64 *zr = (lds_f64 *)smem, *zi = zr + F, *twr = zi + F, *twi = twr + (F >> 1);
    const float *x = in + (size_t)s * nSamples + (size_t)f * hop;
    for (int j = tid; j < (F >> 1); j += nt) { twr[j] = twRe[j]; twi[j] = twIm[j]; }
    for (int j = tid; j < F; j += nt) { zr[j] = (double)x[j] * win[j]; zi[j] = 0.0; }
    __syncthreads();
    fft_forward_dif(zr, zi, logF, (const lds_f64 *)twr, (const lds_f64 *)twi);
    if (mag) {                                                       // |X[k]|, k <= F/2 (natural order)
        float *m = mag + ((size_t)s * nFrames + f) * ((F >> 1) + 1);
        for (int k = tid; k <= (F >> 1); k += nt) { const int p = bitrev(k, logF); m[k] = (float)sqrt(zr[p] * zr[p] + zi[p] * zi[p]); }
    }
    fft_inverse_dit(zr, zi, logF, (const lds_f64 *)twr, (const lds_f64 *)twi);
    float *o = frames + ((size_t)s * nFrames + f) * F;
    const double invF = 1.0 / (double)F;
    for (int j = tid; j < F; j += nt) o[j] = (float)(zr[j] * invF * win[j]);
}
#endif

#if VP_TU_HAS(1)
__global__ __launch_bounds__(256) void vp_k_stft_ola(const float *__restrict__ frames, float *__restrict__ out, int nSamples,
                                                     int nFrames, int F, int hop, float scale)
{
    const int s = blockIdx.y;
    for (int t = blockIdx.x * blockDim.x + threadIdx.x; t < nSamples; t += gridDim.x * blockDim.x) {
        const int fhi = min(nFrames - 1, t / hop), flo = max(0, (t - F + hop) / hop);
        float acc = 0.f;
        for (int f = flo; f <= fhi; f++) {
            const int j = t - f * hop;
            if (j >= 0 && j < F) acc += frames[((size_t)s * nFrames + f) * F + j];
        }
        out[(size_t)s * nSamples + t] = acc * scale;
    }
}
#endif
