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
//   1: ingest/gate, vocoder, emit   2: vp_k_pitch   3: vp_k_pitch_fast   4: vp_k_pitch_multi, vp_k_pitch_fast_multi
//   5: vp_k_pitch_lite, vp_k_pitch_lite_fast   6: vp_k_pitch_ws, vp_k_pitch_ws_x (vp_pitch_ws.inc)
#ifndef VP_TU
#define VP_TU 0
#endif
#define VP_TU_HAS(K) (VP_TU == 0 || VP_TU == (K))
#define VP_NUM_TUS 6

#define WAVE 64

// The workgroup vocoder's orders from VP_LEV_SCALAR_MIN to 48 take the register-resident Levinson-Durbin (levinson_scalar).  Measured
// on one box (tools/abn.sh): order 48 (configs[4]) +8 %; at order 40 the wave-distributed form is 7 % faster.  (The pitch kernel's
// orders 16..48 take the three-register row forms levinson_row48 / levinson_fast64.)
#ifndef VP_LEV_SCALAR_MIN
#define VP_LEV_SCALAR_MIN 41
#endif

// explicit LDS address space: keeps the compiler on ds_read/ds_write instead of flat accesses
typedef __attribute__((address_space(3))) double lds_f64;
typedef __attribute__((address_space(3))) float lds_f32;
typedef __attribute__((address_space(3))) int lds_i32;

// A bounded inter-wavefront wait ran out (a lost signal, or a wavefront held up for ~a second by a debugger / preemption): the data
// it waited for is not there, so whatever the launch produces is INVALID.  Counted in dbg[slot] (vp_debug_read_stamps) and raised in
// the handle's fault word in pinned host memory, where the host finds it at its next look (vp_capi.hip check_fault): the call fails
// with VP_ERR_TIMEOUT and the handle is poisoned.  The reference asserts on impossible state (PitchProcess.cpp:824,828); this is
// the batch's counterpart.  One lane calls.
struct VpTmo { unsigned long long *ctr; unsigned int *fault; int limit; };
__device__ __forceinline__ VpTmo vp_tmo(const VpDev &d, int slot) { VpTmo t; t.ctr = &d.dbg[slot]; t.fault = d.fault; t.limit = d.spinLimit; return t; }
__device__ __forceinline__ void vp_timeout(const VpTmo &t, int slot = 61)
{
    atomicAdd(t.ctr, 1ULL);
    if (t.fault) __hip_atomic_store(t.fault, (unsigned)(0x100 | slot), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

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

#ifndef VP_LPC_FAST
#define VP_LPC_FAST 1           /* VP_IIR_FAST: LPC autocorrelation with the sum over n split across the lanes (autocorr_rows_fast); 0 = the ordered sums */
#endif
#ifndef VP_ACR_AHEAD
#define VP_ACR_AHEAD 1          /* ... its operands requested a trip ahead (two register sets; not in the 128-register builds) */
#endif
#ifndef VP_VOC_AC_FAST
#define VP_VOC_AC_FAST 1        /* workgroup vocoder, VP_IIR_FAST: autocorrelations with split sums on the windowed samples (0 = the ordered sums) */
#endif
#ifndef VP_VOC_FIR4
#define VP_VOC_FIR4 1           /* workgroup vocoder: residual FIRs as four consecutive outputs per lane (fir4); 0 = eight outputs 64 apart (fir_window8) */
#endif
#ifndef VP_VOC_FIR_FS
#define VP_VOC_FIR_FS 1         /* ... and the residual FIRs (fir_window8) */
#endif
#ifndef VP_VOC_LEV_FS
#define VP_VOC_LEV_FS 1         /* ... and the lane-per-window Levinson-Durbin with fused multiply-adds (0 = the reference's two roundings per term) */
#endif
#ifndef VP_VOC_E_FAST
#define VP_VOC_E_FAST 1         /* ... and the residual energies as lane-parallel partial sums (0 = the ordered sums) */
#endif
#ifndef VP_FIR4_SELECT
#define VP_FIR4_SELECT 1        /* fir4: the filter's first outputs as the four-chain select form (0 = the scalar loop) */
#endif
#ifndef VP_HC_WAVE
#define VP_HC_WAVE 5
#endif
#ifndef VP_FIR_TWO
#define VP_FIR_TWO 1            /* the next chunk's residual beside the recursion: on two wavefronts (full workgroups) */
#endif
#ifndef VP_HC_TWO
#define VP_HC_TWO 1             /* FAST block recursion of orders 17..48: zero-state responses on a second wavefront beside the history matrix */
#endif
#ifndef VP_ACR_FUSE
#define VP_ACR_FUSE 1           /* ... orders above 32: three groups of sixteen lags per pass, operands shared between the groups (autocorr_rows_fast3) */
#endif
#ifndef VP_LEV_TREE
#define VP_LEV_TREE 1           /* ... and its Levinson-Durbin (orders 16..48) with rotation-tree sums (levinson_row48<true>); 0 = the ordered chains */
#endif
#ifndef VP_LPC_LATE
#define VP_LPC_LATE 2           /* ... orders >= 16: lag groups on the two wavefronts the YIN phase leaves idle; 2 = the last one then runs Levinson-Durbin in
                                   that phase (it waits for the other's groups), 1 = the recursion opens the marks phase, 0 = one wavefront does it all */
#endif
#ifndef VP_AC_ROWS
#define VP_AC_ROWS 1            /* exact modes, orders below 32: LPC autocorrelation with lane-parallel products and DPP-ordered sums (autocorr_rows_exact8) */
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

// (double)v of a wave-uniform integer, converted where it is used: the compiler otherwise converts once at the top of the kernel and
// carries the pair of vector registers through every phase (the register-light builds spilled it)
__device__ __forceinline__ double vp_f64_here(int v)
{
    asm volatile("" : "+s"(v));
    return (double)v;
}

// 1.0 in a vector register pair for the DPP forms (their operands cannot be inline constants), made where the routine starts: a plain
// `const double one = 1.0` is hoisted to the top of the kernel and kept alive -- or spilled -- across every phase
__device__ __forceinline__ double vp_one()
{
    double o;
    asm volatile("v_mov_b64 %0, 1.0" : "=v"(o));
    return o;
}

// the stream this workgroup serves (see VpDev::streamMap)
__device__ __forceinline__ int vp_stream(const VpDev &d)
{
    // (through readfirstlane: the map's entry is a loaded value, which the compiler takes for lane-varying -- and with it every row
    // pointer derived from the stream index: 64-bit vector registers carried, or spilled, across the whole kernel)
    return __builtin_amdgcn_readfirstlane(d.streamMap ? d.streamMap[blockIdx.x] : (int)blockIdx.x);
}

typedef __attribute__((address_space(3))) VpPitchState lds_state;
__device__ __forceinline__ void emit_block(const VpGeom &g, const VpCall &c, const VpDev &d, float *__restrict__ out,
                                           const lds_state *stl = nullptr, int boff = 0, const lds_f64 *oA = nullptr);

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
    const int s = vp_stream(d), tid = vp_tid(), nt = blockDim.x;
    float *vr = d.voiceRing + (size_t)s * g.inSize;
    float *sr0 = d.synthRing + (size_t)s * 2 * g.inSize;
    float *sr1 = sr0 + g.inSize;
    const int mono = c.inMono;
    const float *xin = in + (size_t)s * (mono ? 1 : 3) * g.N;
    // Four samples per lane and request where everything is a multiple of four (round 5: these loops are bound by the NUMBER of memory
    // instructions, and the ring position by a run-time modulo is forty vector instructions per sample)
    typedef float f4v __attribute__((ext_vector_type(4)));
    const int base = (c.inCounter + boff) % g.inSize;                        // (uniform: one modulo per launch)
    const bool vec = ((g.N | g.inSize | base) & 3) == 0 && (((size_t)xin | (size_t)vr | (size_t)sr0) & 15) == 0;
    if (vec) {
        for (int i = 4 * tid; i < g.N; i += 4 * nt) {
            int p = base + i;
            p -= (p >= g.inSize) ? g.inSize : 0;                              // (N <= inSize: one wrap at most, never inside the four)
            *(f4v *)(vr + p) = *(const f4v *)(xin + i);
            if (!mono) { *(f4v *)(sr0 + p) = *(const f4v *)(xin + g.N + i); *(f4v *)(sr1 + p) = *(const f4v *)(xin + 2 * g.N + i); }
            else if (mono == 1) { *(f4v *)(sr0 + p) = f4v{0.f, 0.f, 0.f, 0.f}; *(f4v *)(sr1 + p) = f4v{0.f, 0.f, 0.f, 0.f}; }
        }
    } else if (!mono) {
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
    if (vec) {
        // (the sums' order differs from the one-sample loops': any order is inside the rounding band the verdict below allows for)
        for (int i = 4 * tid; i < g.inSize; i += 4 * nt) {
            const f4v a = *(const f4v *)(vr + i);
            { const double a0 = (double)a.x, a1 = (double)a.y, a2 = (double)a.z, a3 = (double)a.w; sv += a0 * a0; sv += a1 * a1; sv += a2 * a2; sv += a3 * a3; }
            if (needS) {
                const f4v b = *(const f4v *)(sr0 + i);
                const double b0 = (double)b.x, b1 = (double)b.y, b2 = (double)b.z, b3 = (double)b.w;
                ss += b0 * b0; ss += b1 * b1; ss += b2 * b2; ss += b3 * b3;
            }
        }
    } else if (needS) {
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
    __shared__ double red[2][16];
    if ((tid & 63) == 0) { red[0][tid >> 6] = sv; red[1][tid >> 6] = ss; }
    __syncthreads();
    if (tid < 2) {
        const float *ring = tid == 0 ? vr : sr0;
        double T = 0.0;
        for (int w = 0; w < nt / WAVE; w++) T += red[tid][w];
        const double band = T * vp_f64_here(2 * g.inSize + 64) * 1.1102230246251565e-16 * 1.5;
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

// The kernels proper, in three parts (one translation unit: they share the helpers above and each other's routines):
#include "vp_fft.inc"            // the wavefront-level FFT (512 complex points in one wavefront's registers) and the real-input split
#include "vp_filters.inc"        // Levinson-Durbin, autocorrelation, FIR, energies, the forms of the all-pole recursion
#include "vp_vocoder_wg.inc"     // K1: workgroup-per-stream vocoder
#include "vp_pitch.inc"          // LDS FFT + K2: pitch corrector

// ------------------------------------------------------------------------------------------------
// K3: emit.  addDryVoice / addSynth (MyBuffer.cpp:309-448) + fillOutputBuffer + clearOutput
// (MyBuffer.cpp:113-133, 218-228).  out[ch] = float(((acc + dry) + synth_ch)); the consumed region of
// the accumulator is zeroed.
// oA: the pitch kernel's LDS copy of the block's accumulator slice (logical positions from outCounter), read instead of HBM
__device__ __forceinline__ void emit_block(const VpGeom &g, const VpCall &c, const VpDev &d, float *__restrict__ out,
                                           const lds_state *stl, int boff, const lds_f64 *oA)
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
    for (int i = vp_tid(); i < g.N; i += blockDim.x) {
        int pos = (c.outCounter + boff + i) % g.outSize;
        int pin = (c.currCounter + boff + i) % g.inSize;
        double v = oA ? oA[i] : acc[pos];
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

#include "vp_pitch_ws.inc"       // K2w: the wave-specialised pitch corrector (round 5)
