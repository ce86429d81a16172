// vp_voc2.hip -- the LPC vocoder (VocoderProcess.cpp:173-297) for LARGE batches, as a pipeline of small kernels in
// which one LANE owns one window instead of one wavefront (or workgroup) owning it.
//
// Why: with one workgroup per stream (vp_k_vocoder) every serial stretch of a window -- Levinson-Durbin, the two
// ordered energy sums, the all-pole recursion -- runs on 64 lanes redundantly, and the one lane-parallel stretch that
// dominates (the autocorrelation, lane = lag) is bound by LDS traffic.  Counters at 1024 streams (profiles/, round 2):
// the vector ALUs are busy half of the kernel's time, i.e. the kernel is throughput-bound on instructions that are
// 3-60x more numerous than the arithmetic needs.  A batch of S streams x nWin windows per block has thousands of
// independent windows; giving each its own lane makes every instruction do 64 windows' worth of the reference's
// arithmetic, in the reference's own order (each lane simply runs the reference's loops), with no LDS and no barriers:
// operands stream from HBM/L2 into registers, uniform ones (the window function) through scalar registers.
//
//   stage      (fused behind ingest+gate) the samples the block's windows cover, linear per stream   [S][2][span] f32
//   autocorr   lane = window, wave = 64 windows x L lags (biaisedAutoCorr, LPC.cpp:44-97)            -> r  [NW][.]
//   levinson   lane = window (levinsonDurbin, LPC.cpp:107-148)                                       -> a  [NW][.]
//   fir        lane = window, wave = 64 windows x 64 outputs (filterFIR, VocoderProcess.cpp:235-251) -> e  [NW][W]
//   energy     EXACT: lane = window: sum e^2 in order (:250); FAST: the slices' sums, added up by the recursion kernel's rows
//   iir        10-deep histories, gain (:264-275); EXACT: lane = window, the reference's chain (:277-286); FAST: wave = window, block form -> out [NW][W]
//   ola        workgroup = stream: gainVoc * out * stWindow added in window order (:291-295) [+ emit]
//
// Every kernel is bit-identical to vp_k_vocoder in VP_IIR_EXACT mode (tests/test_gpu_round2.py); in VP_IIR_FAST mode the
// recursion is the block form the pitch kernel uses (tolerance-tested).  Citations: file:line under /root/reference/Source/.
#define VP_TU 99                 // vp_kernels.hip's device helpers without any of its kernels
#include "vp_kernels.hip"

#include <algorithm>
#include <cstdlib>

#include "vp_voc2.h"

// ------------------------------------------------------------------------------------------------
// window w of the launch -> (cohort-local stream b, window j of the block), stream id, gate
// (one 16-byte record per window, written by the stage kernel: a lane learns everything about its window in ONE round
// trip instead of the chain stream map -> gate -> parameters)
struct V2Win { int b, j, s, oV, oS; bool live; };
__device__ __forceinline__ V2Win v2_window(const VpCall &c, const VpDev &d, const VpV2 &v, int w)
{
    V2Win q;
    const int NW = v.nStreams * c.nWin;
    const int wc = min(w, NW - 1);
    const int4 m = v.meta[wc];
    q.b = wc / c.nWin;
    q.j = wc - q.b * c.nWin;
    q.s = m.w; q.oV = m.y; q.oS = m.z;
    q.live = w < NW && m.x != 0;                                          // VocoderProcess.cpp:199-204
    return q;
}

// r / a vectors of one window: column `lane` of the tile [G][k][64] (coalesced across the lanes)
struct V2Col { double *base; __device__ __forceinline__ double &operator[](int k) const { return base[(size_t)k * 64]; } };
__device__ __forceinline__ V2Col v2_col(double *arr, int K, int w) { V2Col r; r.base = arr + ((size_t)(w >> 6) * K) * 64 + (w & 63); return r; }

// ------------------------------------------------------------------------------------------------
// Layout of what the lane-per-window kernels stream: TRANSPOSED tiles, so that the 64 lanes of a wavefront -- 64 different
// windows -- read one contiguous kilobyte per load instead of 64 separate cache lines:
//   xT[ch][G][i / 4][lane][4]  f32   sample i of window 64 G + lane (every window materialised, overlap and all)
//   eT[ch][G][i / 2][lane][2]  f64   residuals
// (ch 0 voice, 1 side chain.)  The all-pole output goes to out[w][i], window-major, which is what the overlap-add reads.
// VP_IIR_FAST (round 4) keeps the two intermediates that only its own kernels read in f32, in the same buffers: the side chain's residual
// as eF[G][i / 4][lane][4] (the voice's is not stored at all in that mode) and the all-pole output as outF[w][i] -- half the bytes written
// by the residual and recursion kernels and read by the recursion and the overlap-add (the arithmetic stays double; tolerance mode).
struct V2X {                                                                // samples of one window, by index
    const float *base; int stride;                                         // base: (G, 0, lane, 0)
    __device__ __forceinline__ float operator[](int i) const { return base[(size_t)(i >> 2) * 256 + (i & 3)]; }
};
__device__ __forceinline__ V2X v2_x(const VpV2 &v, int ch, int w)
{
    V2X r;
    r.base = v.xT + (((size_t)ch * v.nGroupsMax + (w >> 6)) * v.W4p * 64 + (w & 63)) * 4;
    r.stride = 0;
    return r;
}
template <class T> struct V2ET {                                            // residuals of one window, by index
    T *base;
    __device__ __forceinline__ T &operator[](int i) const { return base[(size_t)(i >> 1) * 128 + (i & 1)]; }
};
__device__ __forceinline__ V2ET<double> v2_e(const VpV2 &v, int ch, int w)
{
    V2ET<double> r;
    r.base = v.eT + (((size_t)ch * v.nGroupsMax + (w >> 6)) * v.W2p * 64 + (w & 63)) * 2;
    return r;
}

struct V2EF {                                                               // f32 residuals of one window (VP_IIR_FAST), by index
    float *base;
    __device__ __forceinline__ float &operator[](int i) const { return base[(size_t)(i >> 2) * 256 + (i & 3)]; }
};
__device__ __forceinline__ V2EF v2_ef(const VpV2 &v, int w)
{
    V2EF r;
    r.base = (float *)v.eT + (((size_t)(w >> 6)) * v.W4p * 64 + (w & 63)) * 4;
    return r;
}

// stage: every window of the block, sample by sample (ring wrap resolved here), voice and side-chain channel 0.
// spanLds > 0 (round 4; the launch grants 2 x spanLds floats of dynamic LDS): the samples the block's windows cover -- one contiguous
// stretch of the ring, (nWin - 1) hop + W samples -- are first read into LDS with coalesced loads, and the tiles are built from there.
// Read straight from the ring, a wavefront's 64 lanes (eight windows a hop apart x eight positions) touched 64 cache lines per load
// instruction, eight instructions per tile entry: the kernel was bound by that gather (21.5 us for 86 MB at 1024 streams).
// floats of padding per hop in the staged stretch (host and device): (hop + pad) = 33 mod 64
__host__ __device__ static inline int v2_stage_pad(int hop) { return ((33 - hop) % 64 + 64) % 64; }
__device__ __forceinline__ void v2_stage_block(const VpGeom &g, const VpCall &c, const VpDev &d, const VpV2 &v, int spanLds)
{
    extern __shared__ float v2_stage_lds[];
    const int s = vp_stream(d), b = blockIdx.x;
    const float *vr = d.voiceRing + (size_t)s * g.inSize;
    const float *sr0 = d.synthRing + (size_t)s * 2 * g.inSize;
    const int W4 = (g.W + 3) >> 2;
    const int span = (c.nWin - 1) * g.h + g.W;
    const bool viaLds = spanLds >= span && c.nWin > 0;
    // (round 6) the staged stretch is PADDED per hop: sample t sits at t + (t / hop) pad, pad chosen so that consecutive windows -- a hop
    // apart, i.e. a whole number of bank rows apart for the plugin's hops of 128 / 256 samples: the tile builder's eight windows x eight
    // positions per wavefront met in eight banks, an 8-way conflict on every read (SQ_LDS_BANK_CONFLICT 3.9x the LDS-active cycles) --
    // start 33 banks apart: at most two lanes of a read meet
    const int hpad = viaLds ? v2_stage_pad(g.h) : 0;
    const int hsh = (g.h & (g.h - 1)) == 0 ? __builtin_ctz(g.h) : -1;
    auto lidx = [&](int t) -> int { return t + (hsh >= 0 ? t >> hsh : t / g.h) * hpad; };
    if (viaLds) {
        int p = ring_pos(c.currCounter, c.vStart + (int)threadIdx.x, g.inSize);
        const int step = blockDim.x % g.inSize;
        for (int t = threadIdx.x; t < span; t += blockDim.x) {
            const int li = lidx(t);
            v2_stage_lds[li] = vr[p];
            v2_stage_lds[spanLds + li] = sr0[p];
            p += step;
            p -= (p >= g.inSize) ? g.inSize : 0;
        }
        __syncthreads();
    }
    if ((int)threadIdx.x < c.nWin) {
        const VpStreamParams sp = d.pitch[s].sp;
        const int live = d.gate[s * 2 + 0] && d.gate[s * 2 + 1];
        v.meta[b * c.nWin + threadIdx.x] = make_int4(live, sp.orderVoice, sp.orderSynth, s);
        v.rank[b * c.nWin + threadIdx.x] = live ? (int)threadIdx.x + 1 : 0;
        v.liveList[b * c.nWin + threadIdx.x] = threadIdx.x;
    }
    for (int t = threadIdx.x; t < c.nWin * W4; t += blockDim.x) {
        const int i4 = t / c.nWin, j = t - i4 * c.nWin;                       // window fastest: the stream's windows are adjacent lanes of the tile
        const int w = b * c.nWin + j;
        float a4[4], b4[4];
        if (viaLds) {
            const int q0 = j * g.h + 4 * i4;
#pragma unroll
            for (int u = 0; u < 4; u++) {
                const bool in = 4 * i4 + u < g.W;
                const int li = lidx(min(q0 + u, span - 1));
                a4[u] = in ? v2_stage_lds[li] : 0.0f;
                b4[u] = in ? v2_stage_lds[spanLds + li] : 0.0f;
            }
        } else {
        int p = ring_pos(c.currCounter, c.vStart + j * g.h + 4 * i4, g.inSize);
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const bool in = 4 * i4 + u < g.W;
            a4[u] = in ? vr[p] : 0.0f;
            b4[u] = in ? sr0[p] : 0.0f;
            p = (p + 1 == g.inSize) ? 0 : p + 1;
        }
        }
        float *xv = v.xT + ((((size_t)(w >> 6)) * v.W4p + i4) * 64 + (w & 63)) * 4;
        float *xs = xv + (size_t)v.nGroupsMax * v.W4p * 256;
        *(float4 *)xv = make_float4(a4[0], a4[1], a4[2], a4[3]);
        *(float4 *)xs = make_float4(b4[0], b4[1], b4[2], b4[3]);
    }
}

__global__ __launch_bounds__(256) void vp_k_v2_ingest_stage(VpGeom g, VpCall c, VpDev d, VpV2 v, const float *__restrict__ in, int spanLds)
{
    if (c.fuseIngest) ingest_gate_block(g, c, d, in);                      // (ends with a barrier: the ring is visible)
    v2_stage_block(g, c, d, v, spanLds);
}

// ------------------------------------------------------------------------------------------------
// autocorr: r[m] = (1/W) sum_{n < W-m} ((x[n] w[n]) * x[n+m]) * w[n+m], each lag its own left-to-right sum
// (LPC.cpp:58-96: outer n, inner m).  Lane = window; the wave takes L consecutive lags [m0, m0+L) of the voice
// (blockIdx.y < gV) or of the side chain.  The lane's L+7 samples x[n+m0 ..] slide through registers (eight steps per
// trip, names rotate statically), the window function comes through scalar registers (uniform index).
// FS (VP_IIR_FAST, round 3): the lane's ring holds the WINDOWED samples x[n+m] w[n+m] (one multiply per new sample) and a term is
// ONE fused multiply-add tmp[n] * (x w)[n+m] instead of two multiplies and an add -- 2.2x fewer vector instructions per trip.  The
// reference's association ((x[n] w[n]) x[n+m]) w[n+m] and its two roundings per term are given up: tolerance-mode arithmetic.
// V2_AC_WAVES wavefronts (consecutive 64-window groups, the same lags) per workgroup (round 4).  A workgroup's wavefronts are dealt to
// the CU's four SIMDs in turn.  Single-wavefront workgroups are spread just as evenly on an idle chip -- but not right behind a kernel of
// big workgroups (tools/ubench_placement.hip reproduces it: 41.5 against 25.5 us), and with both processes on this kernel follows the pitch
// kernel: the 896 one-wavefront workgroups it is at 1024 streams left some of the 1024 SIMDs with two of these issue-bound wavefronts
// and others with none, and the kernel lasted as long as the doubly loaded SIMDs -- 37.9 us against 26.1 us for the same wavefronts
// four to a workgroup (vocoder alone, i.e. behind the stage kernel: 28.2 against 26.7 us; with its loads replaced by constants the old
// form still took 30 us: that is how it was found).  The pipeline's other lane-per-window kernels were measured the same way and stay one wavefront
// per workgroup: they are latency-bound (two of their wavefronts on a SIMD overlap) or have several wavefronts per SIMD anyway --
// vp_k_v2_iir_fast 31.1 -> 30.8 us, vp_k_v2_fir2 16.8 -> 17.0, vp_k_v2_levinson2 12.3 -> 12.4, vp_k_v2_iir_exact<40> 157 -> 162.
#define V2_AC_WAVES 4
// Round 6, VP_IIR_FAST: the workgroup's four wavefronts are four STRETCHES of n of ONE (window group, lag group) instead of four window
// groups: stretch p takes the main loop's samples [p S8, (p + 1) S8), S8 = W / 4 rounded up to whole 8-sample trips, the last one also
// the tail; the four partial sums of a lag meet in LDS and are added in stretch order.  Few, long windows (configs[4]: 2048 windows of
// 2048 samples) gave this kernel 672 wavefronts of 256 trips each -- 63 us on 1024 SIMDs; this way it has four times the wavefronts, a
// quarter as long.  Which terms meet in which partial sum depends on the window length alone (not on the orders or on how many
// windows a launch carries), so a stream's tolerance-mode bits do not depend on its neighbours.
// (Measured on the way, same box, rocprofv3: stretches as a grid dimension whose partial sums meet in HBM -- the LAST workgroup of a cell,
// found by a counter behind __threadfence(), adds them -- cost 60 us PER FENCE LEVEL: an agent-scope release/acquire on this chip writes
// back and invalidates an XCD's whole L2.  Twenty-four lags per wavefront (a 32-entry register ring, a third of the tile traffic):
// 197 registers, 99 us against 27 -- fewer, longer wavefronts wait longer for the same loads.  The lag groups of one window group as
// the four wavefronts of a workgroup, for L1 hits on the shared tile rows: 35 against 33 us.)
// VP_IIR_EXACT keeps the single left-to-right sum per lag: four window groups per workgroup.
template <int L, bool FS = false>
__global__ __launch_bounds__(64 * V2_AC_WAVES) void vp_k_v2_autocorr(VpGeom g, VpCall c, VpDev d, VpV2 v)
{
    static_assert(L == 4 || L == 8, "lag groups of 4 or 8");
    extern __shared__ double smem[];
    const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));   // (uniform FOR THE COMPILER too: the stretch's bounds index scalar loads)
    constexpr int nP = FS ? V2_AC_WAVES : 1;                               // stretches (FS: one per wavefront of the workgroup)
    const int pz = FS ? wv : 0;
    const int lane = threadIdx.x & 63, w = (FS ? (int)blockIdx.x : (int)blockIdx.x * V2_AC_WAVES + wv) * WAVE + lane;
    const int gV = (v.oVmax + L) / L;                                     // lag groups of the voice: ceil((oVmax + 1) / L)
    const bool isS = (int)blockIdx.y >= gV;
    const int m0 = (isS ? (int)blockIdx.y - gV : (int)blockIdx.y) * L;
    const V2Win q = v2_window(c, d, v, w);
    const int W = g.W;
    // the window function in LDS (uniform reads = broadcasts).  Scalar loads would do, but they return out of order, so each
    // one has to be waited for on its own (lgkmcnt(0)) and the loop stalls three times per trip; LDS reads pipeline.
    lds_f64 *wl = (lds_f64 *)smem;
    for (int i = threadIdx.x; i < W + 16; i += 64 * V2_AC_WAVES) wl[i] = (i < W) ? d.vocWin[i] : 0.0;
    __syncthreads();
    if (!__any(q.live)) return;                                              // (FS: the four wavefronts see the same windows -- all leave or none)
    const V2X x = v2_x(v, isS ? 1 : 0, q.b * c.nWin + q.j);
    double sum[L];
#pragma unroll
    for (int j = 0; j < L; j++) sum[j] = 0.0;
    // all L lags are inside the window while n + m0 + L - 1 < W
    const int nAll = max(0, W - m0 - (L - 1));
    const int nMain = nAll & ~7;
    // this wavefront's stretch of the main loop: fixed sample positions (a function of W alone)
    const int S8 = ((W / nP) + 7) & ~7;
    const int nLo = FS ? min(nMain, pz * S8) : 0, nHi = (!FS || pz == nP - 1) ? nMain : min(nMain, (pz + 1) * S8);
    if (nHi > nLo) {
        // The lane's samples x[n + m0 + q] slide through a RING of sixteen registers (static names: two trips of eight steps per
        // loop iteration, the second one half a ring further on), eight new ones per trip; those and the eight samples of
        // tmp = x[n] w[n] are requested one trip ahead (aligned 16-byte loads, contiguous across the lanes; the last trip
        // reads a few samples past the window: inside the tile's padding, never used).
        // (round 3) the window function of the main loop through SCALAR loads: every index is wave-uniform and inside [0, W), so the
        // values arrive in SGPRs and feed the multiplies as their one scalar operand -- the loop then holds no LDS instruction at
        // all (19 broadcast reads per 8-element trip at ~8 ns of issue each were 40 % of the trip; the LDS copy below serves the tail).
        const double *__restrict__ wg = d.vocWin;
#define V2_AC_WIN(I) wg[I]
        double R[16];
#pragma unroll
        for (int t = 0; t < L; t++) R[t] = FS ? (double)x[nLo + m0 + t] * V2_AC_WIN(nLo + m0 + t) : (double)x[nLo + m0 + t];
        float4 au0, au1, ax0, ax1, bu0, bu1, bx0, bx1, cu0, cu1, cx0, cx1, du0, du1, dx0, dx1;
#define V2_AC_LOAD(U0, U1, X0, X1, N) { U0 = *(const float4 *)&x.base[(size_t)((N) >> 2) * 256]; U1 = *(const float4 *)&x.base[(size_t)(((N) >> 2) + 1) * 256]; \
        X0 = *(const float4 *)&x.base[(size_t)(((N) + m0 + L) >> 2) * 256]; X1 = *(const float4 *)&x.base[(size_t)((((N) + m0 + L) >> 2) + 1) * 256]; }
        // The window function's values of a trip -- w[N .. N+7] for tmp and, in the FS form, w[N+m0+L .. +7] for the ring's new entries
        // (d.vocWin is zero-padded by 16 entries) -- as TWO wide scalar loads requested one trip ahead (WU/WX: two named sets).
        // Left to itself the compiler fetched them piecemeal where they were used: three or four scalar-cache round trips in a row
        // per trip (scalar loads return out of order, so each batch ended in a wait for all of them) -- more than the trip's
        // arithmetic.
        // (WX: w[N+m0 ..]: the FS form uses entries L .. L+7, the reference form entries t + j = 0 .. L+6)
        constexpr int NWX = FS ? 8 : L + 8;
        double wuA[8], wxA[NWX], wuB[8], wxB[NWX];
#define V2_AC_WLOAD(WU, WX, N) { const double *__restrict__ pu_ = &V2_AC_WIN(N), *__restrict__ px_ = &V2_AC_WIN((N) + m0 + (FS ? L : 0)); \
        _Pragma("unroll") for (int t = 0; t < 8; t++) WU[t] = pu_[t]; \
        _Pragma("unroll") for (int t = 0; t < NWX; t++) WX[t] = px_[t]; }
#define V2_AC_TRIP(PH, U0, U1, X0, X1, N, WU, WX) { \
        const float fx_[8] = {X0.x, X0.y, X0.z, X0.w, X1.x, X1.y, X1.z, X1.w}, fu_[8] = {U0.x, U0.y, U0.z, U0.w, U1.x, U1.y, U1.z, U1.w}; \
        _Pragma("unroll") for (int t = 0; t < 8; t++) R[(L + t + PH) & 15] = FS ? (double)fx_[t] * WX[t] : (double)fx_[t]; \
        double u[8]; \
        _Pragma("unroll") for (int t = 0; t < 8; t++) u[t] = (double)fu_[t] * WU[t];                        /* tmp, LPC.cpp:61 */ \
        _Pragma("unroll") for (int t = 0; t < 8; t++) { \
            _Pragma("unroll") for (int j = 0; j < L; j++) { \
                if (FS) sum[j] = __builtin_fma(u[t], R[(t + j + PH) & 15], sum[j]); \
                else { double p = u[t] * R[(t + j + PH) & 15]; p = p * WX[t + j]; sum[j] += p; } } } }
        // (requests run TWO trips ahead -- four named buffer sets, four trips per loop iteration: with one or two wavefronts
        // per SIMD a trip of 0.2-0.4 us does not cover a memory round trip)
        V2_AC_LOAD(au0, au1, ax0, ax1, nLo)
        V2_AC_LOAD(bu0, bu1, bx0, bx1, nLo + 8)
        V2_AC_WLOAD(wuA, wxA, nLo)
        for (int n = nLo; n < nHi; n += 32) {
            V2_AC_LOAD(cu0, cu1, cx0, cx1, n + 16)
            V2_AC_WLOAD(wuB, wxB, n + 8)
            V2_AC_TRIP(0, au0, au1, ax0, ax1, n, wuA, wxA)
            if (n + 8 < nHi) {
                V2_AC_LOAD(du0, du1, dx0, dx1, n + 24)
                V2_AC_WLOAD(wuA, wxA, n + 16)
                V2_AC_TRIP(8, bu0, bu1, bx0, bx1, n + 8, wuB, wxB)
            }
            if (n + 16 < nHi) {
                V2_AC_LOAD(au0, au1, ax0, ax1, n + 32)
                V2_AC_WLOAD(wuB, wxB, n + 24)
                V2_AC_TRIP(0, cu0, cu1, cx0, cx1, n + 16, wuA, wxA)
            }
            if (n + 24 < nHi) {
                V2_AC_LOAD(bu0, bu1, bx0, bx1, n + 40)
                V2_AC_WLOAD(wuA, wxA, n + 32)
                V2_AC_TRIP(8, du0, du1, dx0, dx1, n + 24, wuB, wxB)
            }
        }
#undef V2_AC_LOAD
#undef V2_AC_TRIP
#undef V2_AC_WLOAD
#undef V2_AC_WIN
    }
    if (pz == nP - 1)
    for (int n = nMain; n < W - m0; n++) {                                  // the last steps: lags drop out one by one
        const double u = (double)x[n] * wl[n];
#pragma unroll
        for (int j = 0; j < L; j++) {
            if (n < W - m0 - j) {
                if (FS) sum[j] = __builtin_fma(u, (double)x[n + m0 + j] * wl[n + m0 + j], sum[j]);
                else {
                double p = u * (double)x[n + m0 + j];
                p = p * wl[n + m0 + j];
                sum[j] += p;
                }
            }
        }
    }
    const V2Col r = isS ? v2_col(v.rS, V2_RS_STRIDE, w) : v2_col(v.rV, V2_RV_STRIDE, w);
    const int top = isS ? VP_ORDER_MAX_SYNTH : VP_ORDER_MAX;
    if (FS) {
        // the four stretches' partial sums meet in LDS (behind the window function's copy; the tail is done with it) and are added in
        // stretch order: wavefront p adds lags j = p, p + 4, ... (LPC.cpp:93-96's division on the total)
        lds_f64 *ps = (lds_f64 *)smem + ((W + 16 + 1) & ~1);
#pragma unroll
        for (int j = 0; j < L; j++) ps[(pz * L + j) * WAVE + lane] = sum[j];
        __syncthreads();
        if (q.live) {
#pragma unroll
            for (int j = 0; j < L; j++) {
                if ((j & (nP - 1)) == pz && m0 + j <= top) {
                    double s = ps[(0 * L + j) * WAVE + lane];
#pragma unroll
                    for (int p = 1; p < nP; p++) s += ps[(p * L + j) * WAVE + lane];
                    r[m0 + j] = s / (double)W;
                }
            }
        }
    } else if (q.live) {
#pragma unroll
        for (int j = 0; j < L; j++)
            if (m0 + j <= top) r[m0 + j] = sum[j] / (double)W;           // :93-96
    }
}

// ------------------------------------------------------------------------------------------------
// levinson: lane = window, the reference's recursion as it stands (LPC.cpp:107-148), fully unrolled for orders up to P with the
// coefficient and autocorrelation vectors in registers (a first version kept them in per-lane LDS columns and spent its time
// waiting for the round trips inside the two ordered sums: 47 us against 18 us at order 40).  The coefficient update is done
// on the pair (i, p - i) at once -- each new value is old[i] - k * old[p - i], the same operands and operation as the
// reference's pass over a copy.  A lane whose own order is smaller steps out of the remaining order steps (EXEC mask).
// FS (VP_IIR_FAST, round 3): sums and updates as fused multiply-adds -- half the instructions of a recursion that is executed once
// per wavefront and bound by instruction fetch (four partial accumulators per sum, tried before, did not change the count)
template <int P, bool FS = false>
__device__ __forceinline__ void v2_levinson_body(const VpGeom &g, const VpCall &c, const VpDev &d, const VpV2 &v, int isS)
{
    const int lane = threadIdx.x, w = blockIdx.x * WAVE + lane;
    const V2Win q = v2_window(c, d, v, w);
    if (!__any(q.live)) return;
    const int order = isS ? q.oS : q.oV;
    const int wc = q.b * c.nWin + q.j;
    const V2Col rg = isS ? v2_col(v.rS, V2_RS_STRIDE, wc) : v2_col(v.rV, V2_RV_STRIDE, wc);
    const V2Col ag = isS ? v2_col(v.aS, V2_RS_STRIDE, wc) : v2_col(v.aV, V2_RV_STRIDE, wc);
    double r[P + 1], a[P + 1];
#pragma unroll
    for (int k = 0; k <= P; k++) { r[k] = (k <= order) ? rg[k] : 0.0; a[k] = 0.0; }
    if (!q.live) return;
    const double r0 = r[0];
    if (fabs(r0) < g.levEps) {                                              // :110-114
        for (int k = 0; k <= order; k++) ag[k] = (k == 0) ? 1.0 : 0.0;
        return;
    }
    a[0] = 1.0;
    a[1] = r[1] / r0;
#pragma unroll
    for (int p = 2; p <= P; p++) {
        if (p <= order) {
            double rho_a = 0.0, r_a = 0.0;
#pragma unroll
            for (int i = 1; i < p; i++) {                                   // :120-128
                if (FS) { rho_a = __builtin_fma(r[p - i], a[i], rho_a); r_a = __builtin_fma(r[i], a[i], r_a); }
                else {
                rho_a += r[p - i] * a[i];
                r_a += r[i] * a[i];
                }
            }
            const double k = (r[p] - rho_a) / (r0 - r_a);
#pragma unroll
            for (int i = 1; 2 * i <= p; i++) {                              // a[i] = aPrev[i] - k aPrev[p-i], both ends of the pair
                const double ai = a[i], aj = a[p - i];
                if (FS) { a[i] = __builtin_fma(-k, aj, ai); if (2 * i != p) a[p - i] = __builtin_fma(-k, ai, aj); }
                else {
                a[i] = ai - k * aj;
                if (2 * i != p) a[p - i] = aj - k * ai;
                }
            }
            a[p] = k;
        }
    }
    ag[0] = 1.0;
#pragma unroll
    for (int k = 1; k <= P; k++)
        if (k <= order) ag[k] = a[k] * -1.;                                 // :145-146
}

// ------------------------------------------------------------------------------------------------
// fir: e[i] = a[0] xw[i] + sum_{k=1..min(order,i)} xw[i-k] a[k], xw = x * anWindow (VocoderProcess.cpp:235-251), taps in
// the reference's order.  Lane = window, the wave takes outputs [i0, i0 + 64) of 64 windows: coefficients (P + 1, zero
// above the lane's own order) and the last P inputs live in registers; four outputs per trip (the next trip's samples
// requested before this trip's arithmetic), then the history moves by four.  A tap that reaches left of the window, or
// above the order, multiplies an exact zero: the sum is unchanged.
// isS = 0: voice -> eV, 1: side chain -> eS (a launch of its own: its orders are much smaller).
// Also leaves the slice's own sum of e^2 (in order) in EEp: VP_IIR_FAST takes the window energies as the sum of those (the
// reference's single left-to-right sum -- vp_k_v2_energy -- stays the exact mode's), and then eVoice is not stored at all.
// FS (VP_IIR_FAST, round 3): a tap is ONE fused multiply-add instead of a product and an addition (half the vector instructions of
// the kernel; one rounding per tap instead of two: tolerance-mode arithmetic)
template <int P, bool FS = false>
__device__ __forceinline__ void v2_fir_body(const VpGeom &g, const VpCall &c, const VpDev &d, const VpV2 &v, int isS)
{
    static_assert(P % 4 == 0 && P >= 4, "P multiple of 4");
    extern __shared__ double smem[];
    const int lane = threadIdx.x, w = blockIdx.x * WAVE + lane;
    const V2Win q = v2_window(c, d, v, w);
    if (!__any(q.live)) return;
    const int W = g.W, i0 = blockIdx.y * V2_FIR_SLICE, i1 = min(W, i0 + V2_FIR_SLICE);
    const int order = isS ? q.oS : q.oV;
    const int wc = q.b * c.nWin + q.j;
    const V2X x = v2_x(v, isS ? 1 : 0, wc);
    const V2Col ag = isS ? v2_col(v.aS, V2_RS_STRIDE, wc) : v2_col(v.aV, V2_RV_STRIDE, wc);
    const V2ET<double> e = v2_e(v, isS ? 1 : 0, wc);
    const V2EF ef = v2_ef(v, wc);
    const bool store = q.live && (isS || !c.iirFast), storeF = c.iirFast;
    // the slice's stretch of the window function in LDS, wl[k] = win[i0 - P + k] (uniform reads: see vp_k_v2_autocorr)
    lds_f64 *wl = (lds_f64 *)smem + P - i0;                                 // wl[i] = win[i] for i in [i0 - P, i1 + 8)
    for (int k = lane; k < P + V2_FIR_SLICE + 8; k += WAVE) { const int i = i0 - P + k; ((lds_f64 *)smem)[k] = (i >= 0 && i < W) ? d.vocWin[i] : 0.0; }
    __syncthreads();
    double a[P + 1], h[P + 4];                                              // h[q] = xw[i - 1 - q]
#pragma unroll
    for (int k = 0; k <= P; k++) a[k] = (k <= order) ? ag[k] : 0.0;
#pragma unroll
    for (int t = 0; t < P; t += 4) {                                        // the P inputs in front of the slice, four per aligned load
        const int ib = i0 - 4 - t;                                          // x[ib .. ib + 3] -> h[t + 3 .. t]
        float4 f_ = make_float4(0.f, 0.f, 0.f, 0.f);
        if (ib >= 0) f_ = *(const float4 *)&x.base[(size_t)(ib >> 2) * 256];
        h[t + 3] = (double)f_.x * wl[ib]; h[t + 2] = (double)f_.y * wl[ib + 1]; h[t + 1] = (double)f_.z * wl[ib + 2]; h[t] = (double)f_.w * wl[ib + 3];
    }
    double Ep = 0.0;
    const int i4 = i0 + ((i1 - i0) & ~3);
    float4 fa, fb;
#define V2_FIR_LOAD(F, I) F = *(const float4 *)&x.base[(size_t)((I) >> 2) * 256];
#define V2_FIR_TRIP(F, I) { double xn[4], en[4]; \
        xn[0] = (double)F.x * wl[(I)]; xn[1] = (double)F.y * wl[(I) + 1]; xn[2] = (double)F.z * wl[(I) + 2]; xn[3] = (double)F.w * wl[(I) + 3]; \
        _Pragma("unroll") for (int t = 0; t < 4; t++) { \
            /* tap k of output i + t reads xw[i + t - k]: one of this trip's new inputs (k <= t) or history h[k - 1 - t] */ \
            double acc = a[0] * xn[t]; \
            _Pragma("unroll") for (int k = 1; k <= P; k++) { \
                const double xv_ = (k <= t) ? xn[t - k < 0 ? 0 : t - k] : h[k - 1 - t < 0 ? 0 : k - 1 - t]; \
                if (FS) acc = __builtin_fma(xv_, a[k], acc); else acc += xv_ * a[k]; } \
            en[t] = acc; } \
        _Pragma("unroll") for (int t = P - 1; t >= 4; t--) h[t] = h[t - 4]; \
        h[3] = xn[0]; h[2] = xn[1]; h[1] = xn[2]; h[0] = xn[3]; \
        _Pragma("unroll") for (int t = 0; t < 4; t++) Ep += en[t] * en[t]; \
        if (store) { if (storeF) *(float4 *)&ef.base[(size_t)((I) >> 2) * 256] = make_float4((float)en[0], (float)en[1], (float)en[2], (float)en[3]); \
                     else { typedef double d2_ __attribute__((ext_vector_type(2))); d2_ p0_, p1_; p0_.x = en[0]; p0_.y = en[1]; p1_.x = en[2]; p1_.y = en[3]; \
                     *(d2_ *)&e.base[(size_t)((I) >> 1) * 128] = p0_; *(d2_ *)&e.base[(size_t)(((I) >> 1) + 1) * 128] = p1_; } } }
    // (loads run one trip ahead and are unconditional: past the window they land in the tile's padding)
    V2_FIR_LOAD(fa, i0)
    int i = i0;
    for (; i < i4; i += 8) {
        V2_FIR_LOAD(fb, i + 4)
        V2_FIR_TRIP(fa, i)
        if (i + 4 < i4) {
            V2_FIR_LOAD(fa, i + 8)
            V2_FIR_TRIP(fb, i + 4)
        }
    }
#undef V2_FIR_LOAD
#undef V2_FIR_TRIP
    for (i = i4; i < i1; i++) {                                             // window lengths that are not a multiple of 4
        const double xn = (double)x[i] * wl[i];
        double acc = a[0] * xn;
#pragma unroll
        for (int k = 1; k <= P; k++) acc += h[k - 1] * a[k];
#pragma unroll
        for (int t = P - 1; t >= 1; t--) h[t] = h[t - 1];
        h[0] = xn;
        Ep += acc * acc;
        if (store) { if (storeF) ef[i] = (float)acc; else e[i] = acc; }
    }
    if (q.live) v.EEp[((size_t)wc * 2 + (isS ? 1 : 0)) * v.nSlices + blockIdx.y] = Ep;
}

// The voice's and the side chain's coefficient / residual kernels are ONE launch each (blockIdx.y resp. blockIdx.z selects;
// the side chain's orders are much smaller -- its own template parameter -- and its wavefronts fill in beside the voice's
// instead of queueing behind them: 34 -> 24 us and 22 -> 18 us at 1024 streams).
// (round 3, measured and dropped: the autocorrelation's sums split into 2 or 4 stretches of n, a wavefront each, in VP_IIR_FAST mode
// -- "more wavefronts for few, long windows": configs[4] geometry 312.7 -> 299 (2 parts) / 322 us (4 parts), 1024 streams 162.5 -> 170 /
// 189 us: every extra wavefront refills its window-function copy and its register ring, and the kernel is not short of wavefronts)
// (round 3, measured and dropped: the two sums of an order step with four partial accumulators each in VP_IIR_FAST mode -- no gain,
// 18.4 us either way at order 40: the fully unrolled recursion is ~8000 instructions executed once per wavefront, bound by
// instruction fetch, not by the sums' dependent chains)
template <int PV, int PS, bool FS = false>
__global__ __launch_bounds__(64) void vp_k_v2_levinson2(VpGeom g, VpCall c, VpDev d, VpV2 v)
{
    if (blockIdx.y == 0) v2_levinson_body<PV, FS>(g, c, d, v, 0);
    else v2_levinson_body<PS, FS>(g, c, d, v, 1);
}
template <int PV, int PS, bool FS = false>
__global__ __launch_bounds__(64) void vp_k_v2_fir2(VpGeom g, VpCall c, VpDev d, VpV2 v)
{
    if (blockIdx.z == 0) v2_fir_body<PV, FS>(g, c, d, v, 0);
    else v2_fir_body<PS, FS>(g, c, d, v, 1);
}

// ------------------------------------------------------------------------------------------------
// energy: E = sum e[i]^2 left to right (VocoderProcess.cpp:250), lane = window, blockIdx.y = 0: eVoice, 1: eSynth.
// Eight entries per trip, the next trip's requested before this trip's squares and ordered adds.  -> EE[w][2]
__global__ __launch_bounds__(64) void vp_k_v2_energy(VpGeom g, VpCall c, VpDev d, VpV2 v)
{
    const int lane = threadIdx.x, w = blockIdx.x * WAVE + lane;
    const V2Win q = v2_window(c, d, v, w);
    if (!__any(q.live)) return;
    const int W = g.W, wc = q.b * c.nWin + q.j;
    const V2ET<double> e = v2_e(v, blockIdx.y ? 1 : 0, wc);
    double E = 0.0;
    const int W8 = W & ~7;
    double a0[8], a1[8];
#define V2_EN_LOAD(A, I) _Pragma("unroll") for (int t = 0; t < 8; t++) A[t] = e[(I) + t];
#define V2_EN_TRIP(A) { _Pragma("unroll") for (int t = 0; t < 8; t++) A[t] = A[t] * A[t]; _Pragma("unroll") for (int t = 0; t < 8; t++) E += A[t]; }
    if (W8 > 0) { V2_EN_LOAD(a0, 0) }
    for (int i = 0; i < W8; i += 16) {
        const bool more1 = i + 8 < W8;
        if (more1) { V2_EN_LOAD(a1, i + 8) }
        V2_EN_TRIP(a0)
        if (more1) {
            if (i + 16 < W8) { V2_EN_LOAD(a0, i + 16) }
            V2_EN_TRIP(a1)
        }
    }
#undef V2_EN_LOAD
#undef V2_EN_TRIP
    for (int i = W8; i < W; i++) { const double a_ = e[i]; E += a_ * a_; }
    if (q.live) v.EE[(size_t)w * 2 + blockIdx.y] = E;
}

// The 10-deep energy histories as they stand right after window j of the block was pushed (VocoderProcess.cpp:264-268):
// its own energies, the block's earlier windows (newest first), then what the stream carried in (EeArr).
// entry t of history `which` (0 voice, 1 side chain)
// (a launch of several blocks can have gated blocks in between: only LIVE windows push, so "t windows back" goes through the
// stream's list of live windows; with one block per launch that list is simply 0, 1, 2, ...)
__device__ __forceinline__ double v2_hist_entry(const VpDev &d, const VpV2 &v, int s, int wBase, int j, int t, int which)
{
    const int rr = v.rank[wBase + j] - 1 - t;                               // rank of the wanted live window, or history entry -rr - 1
    return (rr >= 0) ? v.EE[(size_t)(wBase + v.liveList[wBase + rr]) * 2 + which] : d.EeArr[(size_t)s * 20 + which * 10 - rr - 1];
}
__device__ __forceinline__ void v2_histories(const VpDev &d, const VpV2 &v, int s, int wBase, int j, double hv[10], double hs[10])
{
#pragma unroll
    for (int t = 0; t < 10; t++) { hv[t] = v2_hist_entry(d, v, s, wBase, j, t, 0); hs[t] = v2_hist_entry(d, v, s, wBase, j, t, 1); }
}

// g = sqrt(sum EeVoiceArr / sum EeSynthArr) if EeSynth > 1e-4 else 0 (:270-275), for window j of stream s
__device__ __forceinline__ double v2_gain(const VpGeom &g, const VpDev &d, const VpV2 &v, int s, int wBase, int j)
{
    double hv[10], hs[10];
    v2_histories(d, v, s, wBase, j, hv, hs);
    if (!(hs[0] > g.eeFloor)) return 0.0;
    double sv = 0.0, ss = 0.0;
#pragma unroll
    for (int t = 0; t < 10; t++) sv += hv[t];
#pragma unroll
    for (int t = 0; t < 10; t++) ss += hs[t];
    return sqrt(sv / ss);
}

// The same gain by the sixteen lanes of a ROW that serves one window (vp_k_v2_iir_fast, round 4): lane t < 10 of the row takes history
// entry t of both energies -- a window of this launch as the in-order sum of its slices' partial sums (what vp_k_v2_energy_slices, a
// kernel of its own until then, computed: 5-8 us of launch and one round trip for a few kilobytes), an older one from EeArr -- and
// the two sums over the ten entries run through the row broadcast in the order t = 0 .. 9: the same additions in the same order as
// v2_gain, so the result is bit-identical.  Lane 0's entries are the window's own energies: it leaves them in EE for the overlap-add
// kernel's history write-back.  All 64 lanes must call (DPP row operations).
__device__ __forceinline__ double v2_gain_row(const VpGeom &g, const VpDev &d, const VpV2 &v, int s, int wBase, int j, int m, bool live)
{
    double hv = 0.0, hs = 0.0;
    if (m < 10) {
        const int rr = v.rank[wBase + j] - 1 - m;
        if (rr >= 0) {
            const int nS = (g.W + V2_FIR_SLICE - 1) / V2_FIR_SLICE;
            const double *pv = v.EEp + ((size_t)(wBase + v.liveList[wBase + rr]) * 2) * v.nSlices, *ps = pv + v.nSlices;
            for (int k = 0; k < nS; k++) { hv += pv[k]; hs += ps[k]; }
        } else {
            hv = d.EeArr[(size_t)s * 20 - rr - 1];
            hs = d.EeArr[(size_t)s * 20 + 10 - rr - 1];
        }
        if (m == 0 && live) { v.EE[(size_t)(wBase + j) * 2 + 0] = hv; v.EE[(size_t)(wBase + j) * 2 + 1] = hs; }
    }
    const double one = vp_one();
    double sv = 0.0, ss = 0.0, hs0 = 0.0;
    asm volatile("s_nop 1" : "+v"(hv), "+v"(hs), "+v"(sv), "+v"(ss), "+v"(hs0));      // VALU write -> DPP read
    VP_FMAC_BCAST(hs0, hs, one, 0);
#define V2_GR(U) VP_FMAC_BCAST(sv, hv, one, U); VP_FMAC_BCAST(ss, hs, one, U);
    V2_GR(0) V2_GR(1) V2_GR(2) V2_GR(3) V2_GR(4) V2_GR(5) V2_GR(6) V2_GR(7) V2_GR(8) V2_GR(9)
#undef V2_GR
    return (hs0 > g.eeFloor) ? sqrt(sv / ss) : 0.0;
}

// ------------------------------------------------------------------------------------------------
// iir, EXACT: out[i] = g e[i] - sum_{k=1..min(order,i)} out[i-k] a[k] in the reference's order (VocoderProcess.cpp:277-286),
// lane = window: the register-resident chain of vp_k_vocoder (iir_exact_lane), now with 64 different windows in the lanes.
template <int P>
__global__ __launch_bounds__(64) void vp_k_v2_iir_exact(VpGeom g, VpCall c, VpDev d, VpV2 v)
{
    const int lane = threadIdx.x, w = blockIdx.x * WAVE + lane;
    const V2Win q = v2_window(c, d, v, w);
    if (!__any(q.live)) return;
    const int W = g.W;
    // every lane runs the chain (full EXEC): lanes past the last window redo it (identical stores), windows of gated
    // streams chew on whatever their rows hold -- nothing reads those rows (vp_k_v2_ola looks at the gate)
    const int wc = q.b * c.nWin + q.j;
    const int order = q.oV;
    const V2ET<double> es = v2_e(v, 1, wc);
    const V2Col a = v2_col(v.aV, V2_RV_STRIDE, wc);
    double *out = v.out + (size_t)wc * W;
    const double gg = v2_gain(g, d, v, q.s, q.b * c.nWin, q.j);
    iir_exact_lane<P>(es, out, W & ~3, a, order, (const double *)nullptr, gg);
    for (int i = W & ~3; i < W; i++) {                                     // window lengths that are not a multiple of 4
        double acc = gg * es[i];
        const int kmax = min(order, i);
        for (int k = 1; k <= kmax; k++) acc -= out[i - k] * a[k];
        out[i] = acc;
    }
}

// iir, FAST (tolerance mode): the same filter as transposed direct form II,  y = g x + s_1;  s_k = s_{k+1} - a_k y  (one fused
// multiply-add per tap and no dependence between the taps of a sample), with a window's state spread over a 16-lane ROW:
// lane m of the row owns taps m T + 1 .. m T + T (orders up to 16 T).  s_1 reaches every lane of the row through the DPP
// row broadcast of v_fmac_f64, the state crosses to the lower lane through a DPP row shift; four windows per wavefront.
// With one lane per window a sample costs order + 1 dependent-issue slots and a launch has only NW / 64 wavefronts; this
// way it costs T + 3 and there are sixteen times as many.  Differs from the exact chain by rounding only.
// NI: windows per row, interleaved step by step (two independent recursions in one instruction stream); NI = 2 does not pay
// (the kernel is bound by issue, not by the chain's latency) and is not launched.
template <int T, int NI>
__global__ __launch_bounds__(64) void vp_k_v2_iir_fast(VpGeom g, VpCall c, VpDev d, VpV2 v)
{
    const int lane = threadIdx.x, m = lane & 15;
    const int W = g.W;
    const double one = 1.0, zero = 0.0;
    bool live[NI], any = false;
    double gg[NI], na[NI][T], st[NI][T];
    V2EF es[NI];
    float *out[NI];
#pragma unroll
    for (int k = 0; k < NI; k++) {
        const int w = (blockIdx.x * 4 + (lane >> 4)) * NI + k;
        const V2Win q = v2_window(c, d, v, w);
        const int wc = q.b * c.nWin + q.j;
        live[k] = q.live;
        any = any || q.live;
        es[k] = v2_ef(v, wc);
        out[k] = (float *)v.out + (size_t)wc * W;
        gg[k] = v2_gain_row(g, d, v, q.s, q.b * c.nWin, q.j, m, q.live);
        const V2Col ag = v2_col(v.aV, V2_RV_STRIDE, wc);
#pragma unroll
        for (int j = 0; j < T; j++) { const int kk = m * T + 1 + j; na[k][j] = (kk <= q.oV) ? -ag[kk] : 0.0; st[k][j] = 0.0; }
    }
    if (!__any(any)) return;
    // Sixteen samples per trip: lane m of the row loads sample i + m (ONE load per trip, requested a trip ahead), the row
    // broadcast hands sample t to every lane, lane t keeps output t, one store per trip.
    // (g x for the sixteen samples of a trip is taken first, off the recursion's critical path: per sample the chain is
    // then  s_1 broadcast-add -> state update  and nothing else)
#define V2_IF_GX(X, GX, G) { _Pragma("unroll") for (int u_ = 0; u_ < 16; u_++) GX[u_] = zero * zero; \
        double xx_ = (double)(X); asm volatile("s_nop 1" : "+v"(xx_)); \
        VP_FMAC_BCAST(GX[0], xx_, G, 0); VP_FMAC_BCAST(GX[1], xx_, G, 1); VP_FMAC_BCAST(GX[2], xx_, G, 2); VP_FMAC_BCAST(GX[3], xx_, G, 3); \
        VP_FMAC_BCAST(GX[4], xx_, G, 4); VP_FMAC_BCAST(GX[5], xx_, G, 5); VP_FMAC_BCAST(GX[6], xx_, G, 6); VP_FMAC_BCAST(GX[7], xx_, G, 7); \
        VP_FMAC_BCAST(GX[8], xx_, G, 8); VP_FMAC_BCAST(GX[9], xx_, G, 9); VP_FMAC_BCAST(GX[10], xx_, G, 10); VP_FMAC_BCAST(GX[11], xx_, G, 11); \
        VP_FMAC_BCAST(GX[12], xx_, G, 12); VP_FMAC_BCAST(GX[13], xx_, G, 13); VP_FMAC_BCAST(GX[14], xx_, G, 14); VP_FMAC_BCAST(GX[15], xx_, G, 15); }
#define V2_IF_STEP(U) { _Pragma("unroll") for (int k = 0; k < NI; k++) { \
        double yy = gx_[k][U];                                              /* g x[i + U] */ \
        double s0 = st[k][0]; \
        asm volatile("s_nop 1" : "+v"(s0), "+v"(yy));                       /* VALU write -> DPP read */ \
        VP_FMAC_BCAST(yy, s0, one, 0);                                      /* + s_1, held by lane 0 of the row */ \
        const int lo_ = __builtin_amdgcn_update_dpp(0, __double2loint(s0), 0x101, 0xf, 0xf, true);     /* row_shl:1: lane m <- lane m + 1, */ \
        const int hi_ = __builtin_amdgcn_update_dpp(0, __double2hiint(s0), 0x101, 0xf, 0xf, true);     /* 0 into the row's last lane (bound_ctrl: no pre-zeroed destination, two moves fewer per sample) */ \
        const double in_ = __hiloint2double(hi_, lo_); \
        _Pragma("unroll") for (int j = 0; j + 1 < T; j++) st[k][j] = __builtin_fma(na[k][j], yy, st[k][j + 1]); \
        st[k][T - 1] = __builtin_fma(na[k][T - 1], yy, in_); \
        if (m == (U)) yo_[k] = yy; } }
#define V2_IF_STEPS V2_IF_STEP(0) V2_IF_STEP(1) V2_IF_STEP(2) V2_IF_STEP(3) V2_IF_STEP(4) V2_IF_STEP(5) V2_IF_STEP(6) V2_IF_STEP(7) \
        V2_IF_STEP(8) V2_IF_STEP(9) V2_IF_STEP(10) V2_IF_STEP(11) V2_IF_STEP(12) V2_IF_STEP(13) V2_IF_STEP(14) V2_IF_STEP(15)
#define V2_IF_TRIP(X, I) { double yo_[NI], gx_[NI][16]; \
        _Pragma("unroll") for (int k = 0; k < NI; k++) { yo_[k] = 0.0; V2_IF_GX(X[k], gx_[k], gg[k]) } \
        V2_IF_STEPS \
        _Pragma("unroll") for (int k = 0; k < NI; k++) if (live[k]) out[k][(I) + m] = (float)yo_[k]; }
    const int W16 = W & ~15;
    // (round 3) the trip's one load is requested THREE trips ahead (four named registers): a trip is 16 samples x ~44 issue cycles,
    // 0.3 us -- with one wavefront per SIMD (few, long windows) or two, one trip of lead did not cover a round trip to memory
    // (the residuals of a block are tens of megabytes: they do not come from L2), and the recursion sat out the difference
    // (held as loaded -- f32 -- and widened where the trip consumes them: a conversion right behind the load would wait for it)
    float xa[NI], xb[NI], xc[NI], xd[NI];
#pragma unroll
    for (int k = 0; k < NI; k++) {
        xa[k] = (W16 > 0) ? es[k][m] : 0.0f; xb[k] = (W16 > 16) ? es[k][16 + m] : 0.0f;
        xc[k] = (W16 > 32) ? es[k][32 + m] : 0.0f; xd[k] = (W16 > 48) ? es[k][48 + m] : 0.0f;
    }
#define V2_IF_NEXT(X, I) if ((I) < W16) { _Pragma("unroll") for (int k = 0; k < NI; k++) X[k] = es[k][(I) + m]; }
    for (int i = 0; i < W16; i += 64) {
        V2_IF_TRIP(xa, i)
        V2_IF_NEXT(xa, i + 64)
        if (i + 16 < W16) { V2_IF_TRIP(xb, i + 16) V2_IF_NEXT(xb, i + 80) }
        if (i + 32 < W16) { V2_IF_TRIP(xc, i + 32) V2_IF_NEXT(xc, i + 96) }
        if (i + 48 < W16) { V2_IF_TRIP(xd, i + 48) V2_IF_NEXT(xd, i + 112) }
    }
#undef V2_IF_NEXT
    if (W16 < W) {                                                          // the ragged end: same steps, masked loads and stores
        float xr[NI]; double yo_[NI], gx_[NI][16];
#pragma unroll
        for (int k = 0; k < NI; k++) { xr[k] = (W16 + m < W) ? es[k][W16 + m] : 0.0f; yo_[k] = 0.0; V2_IF_GX(xr[k], gx_[k], gg[k]) }
        V2_IF_STEPS
#pragma unroll
        for (int k = 0; k < NI; k++) if (live[k] && W16 + m < W) out[k][W16 + m] = (float)yo_[k];
    }
#undef V2_IF_GX
#undef V2_IF_STEPS
#undef V2_IF_STEP
#undef V2_IF_TRIP
}

// ------------------------------------------------------------------------------------------------
// ola: every output sample adds its covering windows in window order -- the order of the reference's addOutSample calls
// (VocoderProcess.cpp:291-295, MyBuffer.cpp:181-191) -- each term gainVoc * out[i] * stWindow[i]; optionally the emit
// epilogue.  Workgroup = stream.
// (round 4, measured and dropped: the emitted block's samples handed to the emit epilogue through LDS instead of through the accumulator
// ring -- 16 bytes per output sample less -- 12.0 -> 13.9 us at 1024 streams, 18.1 -> 21.9 us at the configs[4] geometry: the kernel
// is not bound by those bytes)
__global__ __launch_bounds__(256) void vp_k_v2_ola(VpGeom g, VpCall c, VpDev d, VpV2 v, float *__restrict__ out)
{
    const int s = vp_stream(d), b = blockIdx.x, tid = threadIdx.x;
    if (d.gate[s * 2 + 0] && d.gate[s * 2 + 1]) {
        if (tid < 20) {                                                      // the block's last window leaves the histories behind
            // (twenty lanes of ONE wavefront: all of them have read the old entries before any of them stores)
            const double hnew = v2_hist_entry(d, v, s, b * c.nWin, c.nWin - 1, tid % 10, tid / 10);
            d.EeArr[(size_t)s * 20 + tid] = hnew;
        }
        const double gainVoc = d.pitch[s].sp.gainVoc;
        const size_t o0 = (size_t)b * c.nWin * g.W;
        double *acc = d.outAcc + (size_t)s * g.outSize;
        const int W = g.W, span = (c.nWin - 1) * g.h + W;
        // (per output sample: no division by a run-time value -- the ring position is a once-reduced base plus a compare-and-subtract
        // (span <= outSize), the covering windows come from shifts when the hop is a power of two; ~75 vector instructions less per sample)
        const int base = __builtin_amdgcn_readfirstlane((c.outCounter + c.vStart) % g.outSize);
        const int hsh = (g.h & (g.h - 1)) == 0 ? __builtin_ctz(g.h) : -1;
        auto run = [&](auto *o) {                                             // (o: the block's windows, f32 in VP_IIR_FAST mode, else f64)
        for (int t = tid; t < span; t += blockDim.x) {
            int pos = base + t;
            pos -= (pos >= g.outSize) ? g.outSize : 0;
            double a = acc[pos];
            const int jlo = max(0, hsh >= 0 ? (t - W + g.h) >> hsh : (t - W + g.h) / g.h), jhi = min(c.nWin - 1, hsh >= 0 ? t >> hsh : t / g.h);
            for (int j = jlo; j <= jhi; j++) {
                const int i = t - j * g.h;
                if (i >= 0 && i < W) a += gainVoc * (double)o[(size_t)j * W + i] * d.vocWin[i];
            }
            acc[pos] = a;
        } };
        if (c.iirFast) run((const float *)v.out + o0); else run((const double *)v.out + o0);
    }
    if (c.fuseEmit) {
        __syncthreads();
        emit_block(g, c, d, out);
    }
}

// ------------------------------------------------------------------------------------------------
// Several consecutive blocks in one launch of the pipeline (vp_process_blocks_device, vocoder-only plan): the window grid is
// continuous in stream time (startSample carries from block to block, VocoderProcess.cpp:176-182), so B blocks are simply
// B times as many windows per stream -- B times as many lanes for every kernel above -- with three things per BLOCK:
// the ring it is ingested into, its own silence gate (the whole-ring RMS after ITS ingest) and its own output slab.
// Logical index L counts samples from block 0's index 0: L < latency comes from the ring as it stood before this call,
// anything later straight from the input slabs (the ring cannot hold more than a block or two).

// sample of channel ch (0 voice, 1/2 side chain) at logical index L >= 0, before any of this call's blocks was ingested
__device__ __forceinline__ float v2_mb_src(const VpGeom &g, const VpCall &c, const VpDev &d, const float *__restrict__ in, int s, int ch, int L)
{
    if (L < g.latency) {
        const float *ring = (ch == 0) ? d.voiceRing + (size_t)s * g.inSize : d.synthRing + ((size_t)s * 2 + (ch - 1)) * g.inSize;
        return ring[ring_pos(c.currCounter, L, g.inSize)];
    }
    const int bl = (L - g.latency) / g.N, off = (L - g.latency) - bl * g.N;
    return in[(((size_t)bl * g.S + s) * 3 + ch) * g.N + off];
}

// x / d for x >= 0 and a wave-uniform d that is usually a power of two (hop, block size): a shift then, else the division -- a division
// by a run-time value is ~25 vector instructions, and the multi-block kernels did several per sample
struct V2Div {
    int d, sh;
    __device__ __forceinline__ int operator()(int x) const { return sh >= 0 ? x >> sh : x / d; }
};
__device__ __forceinline__ V2Div v2_div(int d) { V2Div r; r.d = d; r.sh = (d > 0 && (d & (d - 1)) == 0) ? __builtin_ctz(d) : -1; return r; }

// voice and side-chain channel 0 at logical index L >= 0 (v2_mb_src for both, one index computation)
__device__ __forceinline__ void v2_mb_src2(const VpGeom &g, const VpCall &c, const VpDev &d, const float *__restrict__ in, int s, const V2Div &divN, int L,
                                           float &a, float &b)
{
    if (L < g.latency) {
        const int p = ring_pos(c.currCounter, L, g.inSize);
        a = d.voiceRing[(size_t)s * g.inSize + p];
        b = d.synthRing[(size_t)s * 2 * g.inSize + p];
    } else {
        const int bl = divN(L - g.latency), off = (L - g.latency) - bl * g.N;
        const float *q = in + (((size_t)bl * g.S + s) * 3) * g.N + off;
        a = q[0];
        b = q[g.N];
    }
}

// chunkWin > 0 (round 4; the launch grants dynamic LDS for two channels of (chunkWin - 1) hop + W samples): the windows are staged
// chunkWin at a time through LDS -- their stretch of the signal read once with coalesced loads, the tiles built from there (see
// v2_stage_block) -- instead of eight gathers with an index division each per tile entry (247 -> 212 us for 8 blocks of 1024 streams).
__global__ __launch_bounds__(256) void vp_k_v2_mb_ingest_stage(VpGeom g, VpCall c, VpDev d, VpV2 v, VpV2MB mb, const float *__restrict__ in, int chunkWin)
{
    extern __shared__ float v2_stage_lds[];
    __shared__ int gl[V2_MB_MAX];
    const int s = vp_stream(d), b = blockIdx.x, tid = threadIdx.x;
    const int NWs = c.nWin;                                                 // windows per stream in this launch
    // a. every window of the launch into the tiles, and the ring-held part of the dry paths' samples
    const int W4 = (g.W + 3) >> 2;
    const V2Div divN = v2_div(g.N);
    if (chunkWin > 0) {
        const int cap = (chunkWin - 1) * g.h + g.W;                         // floats per channel in LDS
        for (int k0 = 0; k0 < NWs; k0 += chunkWin) {
            const int nk = min(chunkWin, NWs - k0), span = (nk - 1) * g.h + g.W, Lb = mb.vStart[0] + k0 * g.h;
            for (int t = tid; t < span; t += blockDim.x) v2_mb_src2(g, c, d, in, s, divN, Lb + t, v2_stage_lds[t], v2_stage_lds[cap + t]);
            __syncthreads();
            // threads as (row ti, window tk): windows fastest, KW = a power of two covering the chunk's windows (at most 64)
            int ksh = 0;
            while ((1 << ksh) < min(nk, 64)) ksh++;
            const int tk = tid & ((1 << ksh) - 1), ti = tid >> ksh, rows = (int)blockDim.x >> ksh;
            for (int i4 = ti; i4 < W4; i4 += rows)
                for (int k = tk; k < nk; k += 1 << ksh) {
                    const int w = b * NWs + k0 + k, q0 = k * g.h + 4 * i4;
                    float a4[4], b4[4];
#pragma unroll
                    for (int u = 0; u < 4; u++) {
                        const bool inw = 4 * i4 + u < g.W;
                        a4[u] = inw ? v2_stage_lds[q0 + u] : 0.0f;
                        b4[u] = inw ? v2_stage_lds[cap + q0 + u] : 0.0f;
                    }
                    float *xv = v.xT + ((((size_t)(w >> 6)) * v.W4p + i4) * 64 + (w & 63)) * 4;
                    float *xs = xv + (size_t)v.nGroupsMax * v.W4p * 256;
                    *(float4 *)xv = make_float4(a4[0], a4[1], a4[2], a4[3]);
                    *(float4 *)xs = make_float4(b4[0], b4[1], b4[2], b4[3]);
                }
            __syncthreads();
        }
    } else
    for (int t = tid; t < NWs * W4; t += blockDim.x) {
        const int i4 = t / NWs, k = t - i4 * NWs;
        const int w = b * NWs + k;
        const int L0 = mb.vStart[0] + k * g.h + 4 * i4;
        float a4[4], b4[4];
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const bool inw = 4 * i4 + u < g.W;
            a4[u] = inw ? v2_mb_src(g, c, d, in, s, 0, L0 + u) : 0.0f;
            b4[u] = inw ? v2_mb_src(g, c, d, in, s, 1, L0 + u) : 0.0f;
        }
        float *xv = v.xT + ((((size_t)(w >> 6)) * v.W4p + i4) * 64 + (w & 63)) * 4;
        float *xs = xv + (size_t)v.nGroupsMax * v.W4p * 256;
        *(float4 *)xv = make_float4(a4[0], a4[1], a4[2], a4[3]);
        *(float4 *)xs = make_float4(b4[0], b4[1], b4[2], b4[3]);
    }
    for (int t = tid; t < 3 * g.latency; t += blockDim.x) {
        const int ch = t / g.latency, L = t - ch * g.latency;
        v.dry[((size_t)b * 3 + ch) * g.latency + L] = v2_mb_src(g, c, d, in, s, ch, L);
    }
    __syncthreads();
    // b. the blocks into the ring one after the other, each with its own gate (MyBuffer.cpp:74-105, VocoderProcess.cpp:199-204)
    if (mb.preIngested) {
        if (tid < mb.nBlocks) gl[tid] = d.gateB[((size_t)tid * g.S + s) * 2 + 0] && d.gateB[((size_t)tid * g.S + s) * 2 + 1];
        __syncthreads();
    } else
    for (int bl = 0; bl < mb.nBlocks; bl++) {
        ingest_gate_block(g, c, d, in + (size_t)bl * g.S * 3 * g.N, bl * g.N);
        if (tid == 0) gl[bl] = d.gate[s * 2 + 0] && d.gate[s * 2 + 1];
        __syncthreads();
    }
    // c. the windows' records: gate of their block, orders, rank among the stream's live windows
    if (tid == 0) {
        const VpStreamParams sp = d.pitch[s].sp;
        int nLive = 0, bl = 0;
        for (int k = 0; k < NWs; k++) {
            while (bl + 1 < mb.nBlocks && k >= mb.first[bl + 1]) bl++;
            const int live = gl[bl];
            v.meta[b * NWs + k] = make_int4(live, sp.orderVoice, sp.orderSynth, s);
            v.rank[b * NWs + k] = live ? nLive + 1 : 0;
            if (live) v.liveList[b * NWs + nLive++] = k;
        }
        v.rank[(size_t)v.nGroupsMax * 64 + b] = nLive;                       // (the stream's live count, behind the per-window ranks)
    }
}

// overlap-add of every window of the launch in window order + emit of every block (VocoderProcess.cpp:291-295,
// MyBuffer.cpp:113-133, 309-448); what reaches beyond the last block is left in the accumulator ring for the next call.
// Dynamic LDS: outSize doubles.
__global__ __launch_bounds__(256) void vp_k_v2_mb_ola_emit(VpGeom g, VpCall c, VpDev d, VpV2 v, VpV2MB mb, const float *__restrict__ in,
                                                            float *__restrict__ out)
{
    extern __shared__ double smem[];
    lds_f64 *tail = (lds_f64 *)smem;
    const int s = vp_stream(d), b = blockIdx.x, tid = threadIdx.x;
    const int NWs = c.nWin, wBase = b * NWs, W = g.W, BN = mb.nBlocks * g.N, v0 = mb.vStart[0];
    const VpStreamParams sp = d.pitch[s].sp;
    const double gainVoc = sp.gainVoc;
    double *acc = d.outAcc + (size_t)s * g.outSize;
    double *acc2 = d.outAcc2 ? d.outAcc2 + (size_t)s * g.outSize : nullptr;
    const bool fastF = c.iirFast != 0;                                      // the all-pole output is f32 in VP_IIR_FAST mode
    const float *oF = (const float *)v.out + (size_t)wBase * W;
    const double *oD = v.out + (size_t)wBase * W;
    double *pl = d.pLin ? d.pLin + (size_t)s * ((size_t)V2_MB_MAX * g.N + g.outSize) : nullptr;
    // (round 4, measured and dropped: the run-time divisions of this loop -- ring position, covering windows, block and slab of a
    // sample -- as shifts / compare-and-subtract: 217.7 -> 229.8 us for 8 blocks of 1024 streams)
    for (int t = tid; t < BN + g.outSize; t += blockDim.x) {
        double val = 0.0;
        if (t < g.outSize) {
            const int pos = (c.outCounter + t) % g.outSize;
            val = acc[pos];
            if (acc2) val += acc2[pos];
        }
        if (NWs > 0 && t >= v0) {
            const int klo = max(0, (t - v0 - W + g.h) / g.h), khi = min(NWs - 1, (t - v0) / g.h);
            for (int k = klo; k <= khi; k++) {
                const int i = t - v0 - k * g.h;
                if (i >= 0 && i < W && v.meta[wBase + k].x) val += gainVoc * (fastF ? (double)oF[(size_t)k * W + i] : oD[(size_t)k * W + i]) * d.vocWin[i];
            }
        }
        if (pl) { val += pl[t]; pl[t] = 0.0; }                               // the pitch corrector's chunks of this call (and the slot cleared for the next)
        if (t < BN) {
            const int bl = t / g.N, i = t - bl * g.N;
            auto smp = [&](int ch) -> double { return (double)((t < g.latency) ? v.dry[((size_t)b * 3 + ch) * g.latency + t] : v2_mb_src(g, c, d, in, s, ch, t)); };
            if (sp.dryOn) val += smp(0) * sp.gainVoice;
            double l = val, r = val;
            if (sp.synthOn) { l += smp(1) * sp.gainSynth; r += smp(2) * sp.gainSynth; }
            float *ob = out + (((size_t)bl * g.S + s) * 2) * g.N;
            ob[i] = (float)l;
            ob[g.N + i] = (float)r;
        } else
            tail[t - BN] = val;
    }
    __syncthreads();
    for (int u = tid; u < g.outSize; u += blockDim.x) {
        const int pos = (c.outCounter + BN + u) % g.outSize;
        acc[pos] = tail[u];
        if (acc2) acc2[pos] = 0.0;
    }
    const int nLive = v.rank[(size_t)v.nGroupsMax * 64 + b];
    if (tid < 20 && nLive > 0) {                                             // the last live window leaves the histories behind
        const double hnew = v2_hist_entry(d, v, s, wBase, v.liveList[wBase + nLive - 1], tid % 10, tid / 10);
        d.EeArr[(size_t)s * 20 + tid] = hnew;
    }
}

// ------------------------------------------------------------------------------------------------
// host side of the pipeline (called from vp_capi.hip's process_device)
#define V2_LAUNCH(K, GRID, BLOCK, LDS, ...) hipLaunchKernelGGL(K, GRID, BLOCK, LDS, st, __VA_ARGS__)
#define V2_LAUNCH_ON(SX, K, GRID, BLOCK, LDS, ...) hipLaunchKernelGGL(K, GRID, BLOCK, LDS, SX, __VA_ARGS__)

static bool v2_lev_fs() { static const bool on = !getenv("VP_V2_NO_LEV_FS"); return on; }   // (diagnostic switch for A/B runs)
static bool v2_fir_fs() { static const bool on = !getenv("VP_V2_NO_FIR_FS"); return on; }   // (diagnostic switch for A/B runs)
// <PV, PS>: orders rounded up to the instantiated sizes (voice 8..48 in steps of 8; side chain 8, 16, 24, 32)
template <int PV> static void v2_launch_lpc_fir_v(int ps, const dim3 &gl, const dim3 &gf, hipStream_t st, const VpGeom &g, const VpCall &c,
                                                   const VpDev &d, const VpV2 &v)
{
#define V2_PAIR(PS) { if (c.iirFast && v2_lev_fs()) V2_LAUNCH((vp_k_v2_levinson2<PV, PS, true>), gl, dim3(64), 0, g, c, d, v); \
                      else V2_LAUNCH((vp_k_v2_levinson2<PV, PS>), gl, dim3(64), 0, g, c, d, v); \
                      if (c.iirFast && v2_fir_fs()) V2_LAUNCH((vp_k_v2_fir2<PV, PS, true>), gf, dim3(64), (size_t)((PV > PS ? PV : PS) + V2_FIR_SLICE + 8) * 8, g, c, d, v); \
                      else V2_LAUNCH((vp_k_v2_fir2<PV, PS>), gf, dim3(64), (size_t)((PV > PS ? PV : PS) + V2_FIR_SLICE + 8) * 8, g, c, d, v); }
    if (ps <= 8) V2_PAIR(8) else if (ps <= 16) V2_PAIR(16) else if (ps <= 24) V2_PAIR(24) else V2_PAIR(32)
#undef V2_PAIR
}
static void v2_launch_lpc_fir(int oV, int oS, int nGroups, int W, hipStream_t st, const VpGeom &g, const VpCall &c, const VpDev &d, const VpV2 &v)
{
    const dim3 gl(nGroups, 2), gf(nGroups, (W + V2_FIR_SLICE - 1) / V2_FIR_SLICE, 2);
    switch ((oV + 7) & ~7) {
    case 8: v2_launch_lpc_fir_v<8>(oS, gl, gf, st, g, c, d, v); break;
    case 16: v2_launch_lpc_fir_v<16>(oS, gl, gf, st, g, c, d, v); break;
    case 24: v2_launch_lpc_fir_v<24>(oS, gl, gf, st, g, c, d, v); break;
    case 32: v2_launch_lpc_fir_v<32>(oS, gl, gf, st, g, c, d, v); break;
    case 40: v2_launch_lpc_fir_v<40>(oS, gl, gf, st, g, c, d, v); break;
    default: v2_launch_lpc_fir_v<48>(oS, gl, gf, st, g, c, d, v); break;
    }
}
template <int P> static void v2_launch_iir_exact(dim3 grid, hipStream_t st, const VpGeom &g, const VpCall &c, const VpDev &d, const VpV2 &v)
{
    V2_LAUNCH(vp_k_v2_iir_exact<P>, grid, dim3(64), 0, g, c, d, v);
}

int vp_v2_init()
{
    if (hipFuncSetAttribute((const void *)vp_k_v2_mb_ola_emit, hipFuncAttributeMaxDynamicSharedMemorySize, VP_V2_MB_LDS_MAX) != hipSuccess) return -1;
    // dynamic-LDS ceilings of the two kernels that use it (process-wide function attributes)
    return 0;
}

static bool v2_ac_fs() { static const bool on = !getenv("VP_V2_NO_AC_FS"); return on; }   // (diagnostic switch for A/B runs)

// autocorrelation ... all-pole output for the NW = nStreams x c.nWin windows the stage kernel has laid out
static void v2_launch_middle(const VpGeom &g, const VpCall &c, const VpDev &d, const VpV2 &v, hipStream_t st, void (*beforeIir)(void *) = nullptr, void *hookArg = nullptr)
{
    const int NW = v.nStreams * c.nWin, nGroups = (NW + 63) / 64;
    if (NW <= 0) { if (beforeIir) beforeIir(hookArg); return; }
    // few, long windows: fewer lags per wave so that there are enough waves (the n loop is serial)
    static const int forceL = getenv("VP_V2_AC_L") ? atoi(getenv("VP_V2_AC_L")) : 0;     // (diagnostic)
    // (round 4, with the wavefronts placed evenly: four lags per wavefront while those workgroups still get a CU each -- the
    // wavefronts are then alone on their SIMDs and half as long --, eight beyond that)
    static const int nCus = [] { int dev = 0; hipDeviceProp_t pr; return (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&pr, dev) == hipSuccess) ? pr.multiProcessorCount : 256; }();
    const int wgs4 = ((nGroups + V2_AC_WAVES - 1) / V2_AC_WAVES) * ((v.oVmax + 4) / 4 + (v.oSmax + 4) / 4);
    if (c.iirFast && v2_ac_fs()) {
        // (round 6) the workgroup's wavefronts are the four stretches of one (window group, lag group): grid = window groups x lag groups,
        // eight lags per wavefront; + LDS for the partial sums [4][L][64]
        const int L = (forceL == 4) ? 4 : 8, gy = (v.oVmax + L) / L + (v.oSmax + L) / L;
        const dim3 ga(nGroups, gy), ba(64 * V2_AC_WAVES);
        const size_t lds = (size_t)(((g.W + 16 + 1) & ~1) + V2_AC_WAVES * L * 64) * 8;
        if (L == 8) V2_LAUNCH((vp_k_v2_autocorr<8, true>), ga, ba, lds, g, c, d, v);
        else V2_LAUNCH((vp_k_v2_autocorr<4, true>), ga, ba, lds, g, c, d, v);
    } else if (forceL ? forceL == 8 : wgs4 > nCus) {
        const int L = 8, gy = (v.oVmax + L) / L + (v.oSmax + L) / L;
        const dim3 ga((nGroups + V2_AC_WAVES - 1) / V2_AC_WAVES, gy), ba(64 * V2_AC_WAVES);
        V2_LAUNCH(vp_k_v2_autocorr<8>, ga, ba, (size_t)(g.W + 16) * 8, g, c, d, v);
    } else {
        const int L = 4, gy = (v.oVmax + L) / L + (v.oSmax + L) / L;
        const dim3 ga((nGroups + V2_AC_WAVES - 1) / V2_AC_WAVES, gy), ba(64 * V2_AC_WAVES);
        V2_LAUNCH(vp_k_v2_autocorr<4>, ga, ba, (size_t)(g.W + 16) * 8, g, c, d, v);
    }
    v2_launch_lpc_fir(v.oVmax, v.oSmax, nGroups, g.W, st, g, c, d, v);
    if (beforeIir) beforeIir(hookArg);                                     // (the caller forks what is to run beside the recursion and the overlap-add)
    if (c.iirFast) {
        // (the window energies and gains are formed inside the recursion kernel: v2_gain_row)
        // (two windows per row interleaved, NI = 2, was tried for few, long windows: 82 -> 133 us at the configs[4] geometry)
        const int Tt = (v.oVmax + 15) / 16;
        const dim3 gi((NW + 3) / 4);
        if (Tt <= 1) V2_LAUNCH((vp_k_v2_iir_fast<1, 1>), gi, dim3(64), 0, g, c, d, v);
        else if (Tt == 2) V2_LAUNCH((vp_k_v2_iir_fast<2, 1>), gi, dim3(64), 0, g, c, d, v);
        else V2_LAUNCH((vp_k_v2_iir_fast<3, 1>), gi, dim3(64), 0, g, c, d, v);
    } else {
        V2_LAUNCH(vp_k_v2_energy, dim3(nGroups, 2), dim3(64), 0, g, c, d, v);
        switch ((v.oVmax + 7) & ~7) {
        case 8: v2_launch_iir_exact<8>(dim3(nGroups), st, g, c, d, v); break;
        case 16: v2_launch_iir_exact<16>(dim3(nGroups), st, g, c, d, v); break;
        case 24: v2_launch_iir_exact<24>(dim3(nGroups), st, g, c, d, v); break;
        case 32: v2_launch_iir_exact<32>(dim3(nGroups), st, g, c, d, v); break;
        case 40: v2_launch_iir_exact<40>(dim3(nGroups), st, g, c, d, v); break;
        default: v2_launch_iir_exact<48>(dim3(nGroups), st, g, c, d, v); break;
        }
    }
}

void vp_v2_launch(const VpGeom &g, const VpCall &c, const VpDev &d, const VpV2 &v_, const float *d_in, float *d_out, hipStream_t st,
                  void (*beforeIir)(void *), void *hookArg)
{
    const VpV2 &v = v_;
    // the staged stretch of the ring through LDS when two channels of it fit beside the kernel's static LDS (else straight from the ring)
    // (per channel: the stretch the block's windows cover, padded per hop -- v2_stage_pad)
    const int span = (c.nWin - 1) * g.h + g.W, spanPad = span + (span / g.h + 1) * v2_stage_pad(g.h);
    const int spanLds = (c.nWin > 0 && (size_t)2 * spanPad * sizeof(float) <= 48 * 1024) ? ((spanPad + 3) & ~3) : 0;
    V2_LAUNCH(vp_k_v2_ingest_stage, dim3(v.nStreams), dim3(256), (size_t)2 * spanLds * sizeof(float), g, c, d, v, d_in, spanLds);
    v2_launch_middle(g, c, d, v, st, beforeIir, hookArg);
    V2_LAUNCH(vp_k_v2_ola, dim3(v.nStreams), dim3(256), 0, g, c, d, v, d_out);
}

void vp_v2_launch_blocks(const VpGeom &g, const VpCall &c, const VpDev &d, const VpV2 &v_, const VpV2MB &mb, const float *d_in, float *d_out,
                         hipStream_t st)
{
    const VpV2 &v = v_;
    // windows per LDS chunk of the stage kernel: two channels of (chunk - 1) hop + W floats within 40 KB (0: straight from memory)
    int chunkWin = ((40 * 1024 / 8) - g.W) / g.h + 1;
    chunkWin = (chunkWin >= 2 && g.W <= 40 * 1024 / 8) ? std::min(chunkWin, 64) : 0;
    const size_t ldsStage = chunkWin ? (size_t)2 * ((chunkWin - 1) * g.h + g.W) * sizeof(float) : 0;
    V2_LAUNCH(vp_k_v2_mb_ingest_stage, dim3(v.nStreams), dim3(256), ldsStage, g, c, d, v, mb, d_in, chunkWin);
    v2_launch_middle(g, c, d, v, st);
    V2_LAUNCH(vp_k_v2_mb_ola_emit, dim3(v.nStreams), dim3(256), (size_t)g.outSize * sizeof(double), g, c, d, v, mb, d_in, d_out);
}

