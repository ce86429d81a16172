// vp_stft.hip -- the fused STFT round trip for gfx950: Hann windowing, FFT, [per-bin spectral stage], inverse FFT and overlap-add
// in ONE kernel, one frame per WAVEFRONT, the transform's butterflies in registers.
//
// NO reference counterpart (the reference contains no FFT, no STFT and no phase vocoder: SURVEY.md section 0).  These are the
// kernels BASELINE.json's north_star lists ("Hann windowing, batched FFT/iFFT, per-bin phase unwrap/accumulate and overlap-add ...
// one frame per wavefront with samples staged in LDS"); they are checked against numpy.fft and a build-authored NumPy restatement
// of the phase-vocoder stage (tests/stft_reference.py) -- parity unpinned by nature -- and reported apart from the metric.
//
// Shape of the computation (F = 1024, hop = 256; F = 128 P in general, P = complex points per lane = 8):
//   * a workgroup = VP_STFT_WAVES (4) wavefronts owns a RUN of consecutive rounds of one stream; in a round wavefront w takes frame
//     4 round + w.  Its 1024 real samples are read straight from HBM/L2 (one aligned float2 per lane and register: consecutive frames
//     overlap by 3/4, the re-reads hit L2), windowed and packed as 512 complex points z[n] = x[2n] + i x[2n+1], n = lane + 64 r.
//   * 512-point complex FFT in registers: three radix-8 steps (dft8: 56 fp64 operations on 8 points held by one lane) with two
//     exchanges through a wavefront-private 8 KB LDS buffer between them -- decimation in time in the "autosort" order, so input
//     AND output are in natural order (lane = index mod 64, register = index div 64): no bit reversal anywhere.  The exchange
//     addresses are skewed so that every ds_write_b128 / ds_read_b128 lane group hits distinct banks (tools/stft_fft_model.py
//     replays the index algebra and the bank model of MI355X_MICROARCH.md: 0 conflicts).  No barrier inside a transform: a
//     wavefront's LDS operations execute in order.
//   * real-input split: X[k] = E[k] + W^k O[k] needs Z[k] and Z[N - k]; lane j owns the pairs k = 64 q + j, q < 4, and fetches the
//     partners from lane 64 - j's upper registers (a half exchange: 4 values per lane).  The spectral stage works on the pair
//     in registers: identity (plus an optional magnitude dump) or the phase-vocoder pitch shift below.  Merge, half exchange back,
//     conjugate, the SAME forward transform again (inverse = conj FFT conj), window * 1/N * overlap-add normalisation.
//   * overlap-add in LDS, deterministic: every wavefront parks its windowed output frame (f32) in its own exchange buffer, one
//     barrier, then the workgroup adds, for every output sample of the round's hops, the frames that cover it IN FRAME ORDER onto
//     a carry of the hops the previous round left incomplete, writes each finished sample to HBM exactly once and keeps the
//     unfinished hops as the new carry.  No frame scratch in HBM, no second kernel, no atomics: HBM traffic = input once (+ L2
//     re-reads) + output once.
//   * runs: with few streams a stream is cut into runs of rounds so that the grid fills the chip; a run recomputes the
//     ceil((O - 1) / 4) rounds in front of it with stores suppressed, which rebuilds exactly the carry the previous run ends with
//     -- the additions happen in the same order, so the output does not depend on the partition (tested).
//
// Phase-vocoder stage (template PV; "per-bin phase unwrap/accumulate", north_star; the classic analysis/synthesis pitch shifter):
// per frame and bin k: magnitude and phase; phase advance against the previous frame minus the bin's nominal advance, wrapped to
// (-pi, pi] (the unwrap) -> true frequency; bins move to round(k ratio), magnitudes of bins that land together add, the frequency
// scales by the ratio; the synthesis phase accumulates the scaled advance frame after frame.  A workgroup then owns a whole
// stream (the accumulator is a recurrence over frames); inside a round the four frames' increments are summed in frame order.
#include <hip/hip_runtime.h>

#include "vp_stft.h"

#define WAVE 64
#define NWV VP_STFT_WAVES

typedef __attribute__((address_space(3))) float lds_f32;
typedef __attribute__((address_space(3))) double lds_f64;
#include "vp_fft.inc"
#include "vp_fft32.inc"

// exchange buffers / output slots (8 KB per wavefront; 4 KB in the single-precision build), overlap-add carry (F - hop floats), rounded up to 16 bytes
__host__ __device__ static inline size_t stft_lds_base(int F, int hop, int f32 = 0)
{
    return (((size_t)NWV * (f32 ? 4096 : 8192) + (size_t)(F - hop) * sizeof(float)) + 15) & ~(size_t)15;
}
size_t vp_stft_lds_bytes(int F, int hop, int f32) { return stft_lds_base(F, hop, f32); }
static size_t stft_pv_lds_bytes(int F)
{
    const size_t nb = (size_t)F / 2 + 1;
    // previous-frame phases [NWV + 1][nb], analysis (magnitude, frequency) [NWV][nb][2], phase increments [NWV][nb], accumulator [nb]
    return ((NWV + 1) * nb + NWV * nb * 2 + NWV * nb + nb) * sizeof(double);
}

int vp_stft_supported(int F, int hop)
{
    return (F == 1024 || F == 2048) && hop > 0 && F % hop == 0 && F / hop >= 2 && F / hop <= 16;
}

// Overlap-add of one round, by the whole workgroup (behind the barrier that follows the wavefronts' slot writes): relative hop u of the
// round (hops rd * NWV + u) takes frames w in [u - O + 1, u] IN FRAME ORDER on top of the carry of the hops the previous round left
// incomplete; finished hops go to HBM (once), the others become the new carry.  slots: wavefront w's output frame at w * SLOT floats.
template <int SLOT = 2048>
__device__ __forceinline__ void stft_overlap_add(const VpStftArgs &A, const lds_f32 *slots, lds_f32 *carry, int s, int rd, bool emit, int tid)
{
    const int hop = A.hop, O = A.O, T = A.T;
    float *o = A.out + (size_t)s * T;
    for (int i = tid; i < hop; i += 64 * NWV) {
        for (int u = 0; u < NWV + O - 1; u++) {
            float v = (u < O - 1) ? carry[u * hop + i] : 0.f;
            const int wlo = max(0, u - O + 1), whi = min(u, NWV - 1);
            for (int w = wlo; w <= whi; w++) v += slots[w * SLOT + (u - w) * hop + i];
            if (u < NWV) {
                const long t = (long)(rd * NWV + u) * hop + i;
                if (emit && t < T) o[t] = v;
            } else carry[(u - NWV) * hop + i] = v;
        }
    }
}

#define VP_TWO_PI 6.283185307179586476925286766559

// ---- the phase-vocoder stage's elementary functions, in TURNS (round 5) -----------------------------------------------------------
// The stage kept its phases in radians and called libm: atan2 per bin, sincos of an accumulated phase of up to thousands of radians
// per bin (the device library's argument reduction for large arguments), nine of each per lane and frame -- 6000 vector instructions
// per lane and round, four fifths of the kernel's time.  In turns (1 turn = 2 pi) the reduction is a subtraction of rint(), exact, and
// what is left are three short polynomials (minimax fits on Chebyshev nodes, tools/stft_pv_fit.py; absolute errors below 4e-16 of a
// turn / of the unit circle).  Same stage, same definition (tests/stft_reference.py keeps radians and numpy's functions: an independent
// check); the outputs differ at the 1e-13 level.
__device__ static const double PV_AT[11] = {0.15915494309189532, -0.05305164769729216, 0.031830988616730685, -0.022736420283186426, 0.017683874894479496,
                                            -0.014468416126674181, 0.012238931495957754, -0.010567969142530591, 0.009050862433965945, -0.006910937165090144,
                                            0.003350241773092173};                          // atan(r) / (2 pi r) in u = r^2 on [0, tan^2(pi/8)]
__device__ static const double PV_SN[7] = {6.283185307179581, -41.341702240399435, 81.60524927567752, -76.70585960744309, 42.05867032464745, -15.092818445192446,
                                           3.7567969372293444};                              // sin(2 pi t) / t in v = t^2 on [0, 1/64]
__device__ static const double PV_CS[7] = {1.0000000000000002, -19.739208802178528, 64.93939402241888, -85.4568171016908, 60.24462112885379, -26.424298367777688,
                                           7.810833486309896};                               // cos(2 pi t) in v = t^2 on [0, 1/64]
// atan2(im, re) / (2 pi), in [-0.5, 0.5]
__device__ __forceinline__ double pv_phase_turns(double im, double re)
{
    const double ax = fabs(re), ay = fabs(im);
    const double mx = fmax(ax, ay), mn = fmin(ax, ay);
    double r = (mx > 0.0) ? mn / mx : 0.0;                                     // in [0, 1]  (atan2(0, 0) = 0)
    double base = 0.0;
    if (r > 0.41421356237309503) { r = (r - 1.0) / (r + 1.0); base = 0.125; }  // atan(r) = pi/4 + atan((r - 1) / (r + 1))
    const double u = r * r;
    double q = PV_AT[10];
#pragma unroll
    for (int i = 9; i >= 0; i--) q = __builtin_fma(q, u, PV_AT[i]);
    double t = base + q * r;                                                   // in [0, 1/8]
    if (ay > ax) t = 0.25 - t;
    if (re < 0.0) t = 0.5 - t;
    return (im < 0.0) ? -t : t;
}
// (sin, cos) of 2 pi t for any finite t
__device__ __forceinline__ void pv_sincos_turns(double t, double &sn, double &cs)
{
    t -= rint(t);                                                              // [-1/2, 1/2], exact
    const double q = rint(t * 4.0);                                            // the quarter turn, -2 .. 2
    const double r = t - q * 0.25;                                             // [-1/8, 1/8], exact
    const double v = r * r;
    double ps = PV_SN[6], pc = PV_CS[6];
#pragma unroll
    for (int i = 5; i >= 0; i--) { ps = __builtin_fma(ps, v, PV_SN[i]); pc = __builtin_fma(pc, v, PV_CS[i]); }
    const double sr = ps * r, cr = pc;
    const int iq = (int)q & 3;
    sn = (iq == 0) ? sr : (iq == 1) ? cr : (iq == 2) ? -sr : -cr;
    cs = (iq == 0) ? cr : (iq == 1) ? -sr : (iq == 2) ? -cr : sr;
}

// the wavefront's phase-vocoder work arrays (PV builds only)
struct PvLds {
    lds_f64 *phPrev;           // [NWV + 1][nb]  slot w + 1: frame of wavefront w this round; slot 0: the previous round's last frame
    lds_d2 *ana;               // [NWV][nb]      (magnitude, true frequency in bins) of this wavefront's frame
    lds_f64 *inc;              // [NWV][nb]      synthesis phase increment of each wavefront's frame
    lds_f64 *sum;              // [nb]           synthesis phase accumulator after the previous round
};

// PV: phase-vocoder stage between the transforms; MAG: magnitude dump (builds of their own: the timed round trip carries neither)
template <bool PV, bool MAG>
__global__ __launch_bounds__(64 * NWV) void vp_k_stft_fused(VpStftArgs A)
{
    extern __shared__ double smem[];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int s = blockIdx.y, run = blockIdx.x;
    constexpr int N = 512;                                                     // complex points = F / 2
    const int F = A.F, hop = A.hop, O = A.O, T = A.T;
    lds_d2 *xb = (lds_d2 *)smem + wv * 512;                                    // this wavefront's exchange buffer ...
    lds_f32 *slots = (lds_f32 *)smem;                                          // ... whose first F floats double as its output slot (slot w at w * 2048)
    lds_f32 *carry = (lds_f32 *)smem + NWV * 2048;                             // [(O - 1) hop]
    PvLds pv;
    if (PV) {
        const int nb = N + 1;
        lds_f64 *p = (lds_f64 *)smem + (stft_lds_base(F, hop) / 8);
        pv.ana = (lds_d2 *)p + (size_t)wv * nb; p += NWV * nb * 2;           // (the 16-byte type first: the base is 16-byte aligned, nb is odd)
        pv.phPrev = p; p += (NWV + 1) * nb;
        pv.inc = p; p += NWV * nb;
        pv.sum = p;
    }

    // per-lane constants, once per wavefront: window values of the lane's 16 samples, transform constants, split twiddles
    FftLane L;
    fft_lane_init(L, lane, A.tw1, A.tw2);
    d2 wa[8];                                                                  // (w[2n], w[2n + 1]), n = lane + 64 r
    d2 ws[4];                                                                  // W_1024^(64 q + lane)
#pragma unroll
    for (int r = 0; r < 8; r++) wa[r] = ((const d2 *)A.win)[lane + 64 * r];
#pragma unroll
    for (int q = 0; q < 4; q++) ws[q] = ((const d2 *)A.tws)[lane * 4 + q];
    const bool lane0 = lane == 0;

    for (int i = tid; i < F - hop; i += 64 * NWV) carry[i] = 0.f;
    if (PV) {
        const int nb = N + 1;
        for (int i = tid; i < nb; i += 64 * NWV) { pv.phPrev[i] = 0.0; pv.sum[i] = 0.0; }
    }
    __syncthreads();

    const int rFirst = run * A.roundsPerRun;                                   // first round whose hops this workgroup stores
    const int r0 = max(0, rFirst - (run > 0 ? A.haloRounds : 0));
    const int r1 = min(rFirst + A.roundsPerRun, A.nRounds);
    const float *xs = A.in + (size_t)s * T;
    // the frame's samples are requested a round ahead (see vp_k_stft_fused32)
    f2 xv[8];
    auto request = [&](int rd_) {
        const int f_ = rd_ * NWV + wv;
        if (rd_ < r1 && f_ < A.nFrames) {
            const float *x = xs + (size_t)f_ * hop;
            if (A.aligned) {
#pragma unroll
                for (int r = 0; r < 8; r++) xv[r] = *(const f2 *)(x + 2 * (lane + 64 * r));
            } else {
#pragma unroll
                for (int r = 0; r < 8; r++) xv[r] = f2{x[2 * (lane + 64 * r)], x[2 * (lane + 64 * r) + 1]};
            }
        }
    };
    request(r0);
    for (int rd = r0; rd < r1; rd++) {
        const int f = rd * NWV + wv;
        const bool live = f < A.nFrames;                                       // (wavefront-uniform)
        C8 z;
        RPairs X;                                                              // X[k], X[N - k] of the lane's pairs (vp_fft.inc)
        if (live) {
#pragma unroll
            for (int r = 0; r < 8; r++) { z.re[r] = (double)xv[r].x * wa[r].x; z.im[r] = (double)xv[r].y * wa[r].y; }
        }
        request(rd + 1);
        if (live) {
            fft512_rx(z, xb, L);
            rfft_split(z, xb, lane, (const d2 *)ws, X);
            if (MAG) {                                                         // |X[k]|, k <= N, natural order
                float *m = A.mag + ((size_t)s * A.nFrames + f) * (N + 1);
#pragma unroll
                for (int q = 0; q < 4; q++) {
                    const int k = 64 * q + lane;
                    m[k] = (float)sqrt(X.kr[q] * X.kr[q] + X.ki[q] * X.ki[q]);
                    m[N - k] = (float)sqrt(X.mr[q] * X.mr[q] + X.mi[q] * X.mi[q]);      // (lane 0, q = 0: X[N] sits in the mirror slot)
                }
                if (lane0) m[N / 2] = (float)sqrt(X.hr * X.hr + X.hi * X.hi);
            }
        }
        if (PV) {
            // ---- phase-vocoder stage, phases in TURNS.  Bins of this lane: k = 64 q + lane and N - k (q < 4); lane 0 also holds 0, N and N / 2.
            const int nb = N + 1;
            const double invO = 1.0 / (double)O;                               // nominal phase advance of bin 1 per hop, in turns (O a power of two: exact)
            const double invRatio = 1.0 / A.pvRatio;
            double ph[9];
            int kb[9];
            double mg[9];
            if (live) {
#pragma unroll
                for (int q = 0; q < 4; q++) {
                    kb[2 * q] = 64 * q + lane; kb[2 * q + 1] = N - kb[2 * q];
                    mg[2 * q] = sqrt(X.kr[q] * X.kr[q] + X.ki[q] * X.ki[q]); ph[2 * q] = pv_phase_turns(X.ki[q], X.kr[q]);
                    mg[2 * q + 1] = sqrt(X.mr[q] * X.mr[q] + X.mi[q] * X.mi[q]); ph[2 * q + 1] = pv_phase_turns(X.mi[q], X.mr[q]);
                }
                kb[8] = N / 2; mg[8] = sqrt(X.hr * X.hr + X.hi * X.hi); ph[8] = pv_phase_turns(X.hi, X.hr);            // lane 0 only
#pragma unroll
                for (int e = 0; e < 9; e++) if (e < 8 || lane0) pv.phPrev[(wv + 1) * nb + kb[e]] = ph[e];
            }
            __syncthreads();
            if (live) {
#pragma unroll
                for (int e = 0; e < 9; e++) {
                    if (e == 8 && !lane0) continue;
                    const int k = kb[e];
                    double d = ph[e] - pv.phPrev[wv * nb + k] - (double)k * invO;
                    d -= rint(d);                                              // the unwrap: deviation from the nominal advance in [-1/2, 1/2] turns
                    pv.ana[k] = d2{mg[e], (double)k + d * (double)O};          // true frequency in bins
                }
                wave_sync();
                // bins move to floor(k ratio + 0.5): synthesis bin kk gathers the analysis bins that land on it, in increasing k
#pragma unroll
                for (int e = 0; e < 9; e++) {
                    if (e == 8 && !lane0) continue;
                    const int kk = kb[e];
                    const int kc = (int)((double)kk * invRatio);
                    double sm = 0.0, sf = 0.0;
                    // (the five candidates are requested at once and chosen from in registers: a conditional read per candidate was a
                    // dependent LDS round trip each, forty-five per frame -- the stage's largest single cost)
                    d2 cand[5];
#pragma unroll
                    for (int c_ = 0; c_ < 5; c_++) cand[c_] = pv.ana[min(max(kc - 2 + c_, 0), N)];
#pragma unroll
                    for (int c_ = 0; c_ < 5; c_++) {
                        const int k = kc - 2 + c_;
                        if (k >= 0 && k <= N && (int)floor((double)k * A.pvRatio + 0.5) == kk) { sm += cand[c_].x; sf = cand[c_].y * A.pvRatio; }
                    }
                    mg[e] = sm;
                    pv.inc[wv * nb + kk] = sf * invO;                           // phase advance of the synthesis bin over one hop, in turns
                }
            }
            __syncthreads();
            if (live) {
#pragma unroll
                for (int e = 0; e < 9; e++) {
                    if (e == 8 && !lane0) continue;
                    const int kk = kb[e];
                    double sp = pv.sum[kk];
                    for (int w = 0; w <= wv; w++) sp += pv.inc[w * nb + kk];   // frames of the round in order (frames beyond the stream's last add nothing: they are never live)
                    ph[e] = sp;
                    double sn, cs;
                    pv_sincos_turns(sp, sn, cs);
                    const double re = mg[e] * cs, im = mg[e] * sn;
                    if (e == 8) { X.hr = re; X.hi = im; }
                    else if (e & 1) { X.mr[e >> 1] = re; X.mi[e >> 1] = im; }
                    else { X.kr[e >> 1] = re; X.ki[e >> 1] = im; }
                }
                if (lane0) { X.ki[0] = 0.0; X.mi[0] = 0.0; }                   // X[0], X[N] of a real frame are real: keep the real parts
            }
            __syncthreads();                                                   // every wavefront has read phPrev / sum / inc of this round
            // the last live wavefront of the round leaves the next round's "previous frame" and accumulator
            const int lastLive = min(NWV - 1, A.nFrames - 1 - rd * NWV);
            if (live && wv == lastLive) {
#pragma unroll
                for (int e = 0; e < 9; e++) {
                    if (e == 8 && !lane0) continue;
                    pv.phPrev[kb[e]] = pv.phPrev[(wv + 1) * nb + kb[e]];
                    pv.sum[kb[e]] = ph[e] - rint(ph[e]);
                }
            }
        }
        lds_f2 *slot = (lds_f2 *)(slots + wv * 2048);
        if (live) {
            // ---- merge (the inverse of the split), scaled by c = overlap-add normalisation / N, and conjugated for the inverse transform
            rfft_merge_conj(z, xb, lane, (const d2 *)ws, X, A.c);
            fft512_rx(z, xb, L);                                                  // y = FFT(conj Z'): x'[2n] = Re y, x'[2n + 1] = -Im y (1/N is in c)
            wave_sync();
#pragma unroll
            for (int r = 0; r < 8; r++) slot[lane + 64 * r] = f2{(float)(z.re[r] * wa[r].x), (float)(-(z.im[r] * wa[r].y))};
        } else {
#pragma unroll
            for (int r = 0; r < 8; r++) slot[lane + 64 * r] = f2{0.f, 0.f};
        }
        __syncthreads();
        stft_overlap_add(A, slots, carry, s, rd, rd >= rFirst, tid);
        __syncthreads();
    }
}

// ---- 2048-point frames (BASELINE configs[4]'s "2048-pt FFT hop 512"): 1024 complex points, SIXTEEN per lane --------------------------
// The same round structure; the transform is a radix-2 step on top of two 512-point ones.  Forward, decimation in time: the lane loads
// z[2m] and z[2m + 1] for m = lane + 64 r (four consecutive samples: one aligned float4), E = FFT512(even), O = FFT512(odd),
// Z[k'] = E + W^k' O, Z[k' + 512] = E - W^k' O (k' = 64 q + lane: in-lane, natural order).  Backward, decimation in frequency on the
// conjugated spectrum: e = lo + hi, o = (lo - hi) W^k', two 512-point transforms, y[2k'] and y[2k' + 1] -- again four consecutive
// samples per lane and register.  The 512-point transforms' twiddles come from LDS copies of the tables (the sixteen points, the eight
// bin pairs and the top step's twiddles use the registers), the window from the global table (L1/L2-resident: 16 KB).
template <bool MAG>
__global__ __launch_bounds__(64 * NWV) void vp_k_stft_fused2k(VpStftArgs A)
{
    extern __shared__ double smem[];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int s = blockIdx.y, run = blockIdx.x;
    constexpr int N = 1024;                                                    // complex points = F / 2
    const int F = A.F, hop = A.hop, T = A.T;
    lds_d2 *xb = (lds_d2 *)smem + wv * 512;
    lds_f32 *slots = (lds_f32 *)smem;                                          // slot w: the whole exchange buffer (2048 floats)
    lds_f32 *carry = (lds_f32 *)smem + NWV * 2048;
    lds_d2 *twL = (lds_d2 *)smem + stft_lds_base(F, hop) / 16;                 // [8][8] W_64 rows | [64][8] W_512
    FftAddr L;
    fft_addr_init(L, lane);
    for (int i = tid; i < 64; i += 64 * NWV) twL[i] = ((const d2 *)A.tw1)[(i >> 3) * 64 + (i & 7)];
    for (int i = tid; i < 512; i += 64 * NWV) twL[64 + i] = ((const d2 *)A.tw2)[i];
    const lds_d2 *tw1p = twL + (lane >> 3) * 8, *tw2p = twL + 64 + lane * 8;
    d2 wtop[8], ws[8];                                                         // W_1024^(64 q + lane), W_2048^(64 q + lane)
#pragma unroll
    for (int q = 0; q < 8; q++) { wtop[q] = ((const d2 *)A.twTop)[lane * 8 + q]; ws[q] = ((const d2 *)A.tws)[lane * 8 + q]; }
    for (int i = tid; i < F - hop; i += 64 * NWV) carry[i] = 0.f;
    __syncthreads();

    const int rFirst = run * A.roundsPerRun;
    const int r0 = max(0, rFirst - (run > 0 ? A.haloRounds : 0));
    const int r1 = min(rFirst + A.roundsPerRun, A.nRounds);
    const float *xs = A.in + (size_t)s * T;
    typedef float f4 __attribute__((ext_vector_type(4)));
    typedef __attribute__((address_space(3))) f4 lds_f4;
    for (int rd = r0; rd < r1; rd++) {
        const int f = rd * NWV + wv;
        const bool live = f < A.nFrames;
        lds_f4 *slot = (lds_f4 *)(slots + wv * 2048);
        if (live) {
            const float *x = xs + (size_t)f * hop;
            C8 e, o;                                                           // even / odd packed points of the lane: z[2m], z[2m + 1], m = lane + 64 r
#pragma unroll
            for (int r = 0; r < 8; r++) {
                const int m = lane + 64 * r;
                f4 v;
                if (A.aligned) v = *(const f4 *)(x + 4 * m);
                else v = f4{x[4 * m], x[4 * m + 1], x[4 * m + 2], x[4 * m + 3]};
                const d2 w0 = ((const d2 *)A.win)[2 * m], w1 = ((const d2 *)A.win)[2 * m + 1];
                e.re[r] = (double)v.x * w0.x; e.im[r] = (double)v.y * w0.y;
                o.re[r] = (double)v.z * w1.x; o.im[r] = (double)v.w * w1.y;
            }
            fft512_rx(e, xb, L, tw1p, tw2p);
            fft512_rx(o, xb, L, tw1p, tw2p);
            double hr[8], hi[8];
#pragma unroll
            for (int q = 0; q < 8; q++) {                                      // radix-2 on top: lo = E + W^k' O (kept in e), hi = E - W^k' O
                const double tr = __builtin_fma(o.re[q], wtop[q].x, -(o.im[q] * wtop[q].y)), ti = __builtin_fma(o.re[q], wtop[q].y, o.im[q] * wtop[q].x);
                hr[q] = e.re[q] - tr; hi[q] = e.im[q] - ti;
                e.re[q] += tr; e.im[q] += ti;
            }
            RPairsN<8> X;
            rfft_split_n<8>(e.re, e.im, hr, hi, xb, lane, (const d2 *)ws, X);
            if (MAG) {                                                         // |X[k]|, k <= N, natural order
                float *mg = A.mag + ((size_t)s * A.nFrames + f) * (N + 1);
#pragma unroll
                for (int q = 0; q < 8; q++) {
                    const int k = 64 * q + lane;
                    mg[k] = (float)sqrt(X.kr[q] * X.kr[q] + X.ki[q] * X.ki[q]);
                    mg[N - k] = (float)sqrt(X.mr[q] * X.mr[q] + X.mi[q] * X.mi[q]);
                }
                if (lane == 0) mg[N / 2] = (float)sqrt(X.hr * X.hr + X.hi * X.hi);
            }
            rfft_merge_conj_n<8>(e.re, e.im, hr, hi, xb, lane, (const d2 *)ws, X, A.c);
#pragma unroll
            for (int q = 0; q < 8; q++) {                                      // radix-2 on top, decimation in frequency: e = lo + hi, o = (lo - hi) W^k'
                const double dr = e.re[q] - hr[q], di = e.im[q] - hi[q];
                e.re[q] += hr[q]; e.im[q] += hi[q];
                o.re[q] = __builtin_fma(dr, wtop[q].x, -(di * wtop[q].y)); o.im[q] = __builtin_fma(dr, wtop[q].y, di * wtop[q].x);
            }
            fft512_rx(e, xb, L, tw1p, tw2p);                                      // y[2k'] ...
            fft512_rx(o, xb, L, tw1p, tw2p);                                      // ... and y[2k' + 1], k' = lane + 64 r: x'[2n] = Re y[n], x'[2n + 1] = -Im y[n]
            wave_sync();
#pragma unroll
            for (int r = 0; r < 8; r++) {
                const int m = lane + 64 * r;
                const d2 w0 = ((const d2 *)A.win)[2 * m], w1 = ((const d2 *)A.win)[2 * m + 1];
                slot[m] = f4{(float)(e.re[r] * w0.x), (float)(-(e.im[r] * w0.y)), (float)(o.re[r] * w1.x), (float)(-(o.im[r] * w1.y))};
            }
        } else {
#pragma unroll
            for (int r = 0; r < 8; r++) slot[lane + 64 * r] = f4{0.f, 0.f, 0.f, 0.f};
        }
        __syncthreads();
        stft_overlap_add(A, slots, carry, s, rd, rd >= rFirst, tid);
        __syncthreads();
    }
}

// ---- single precision (vp_stft_set_precision(t, VP_STFT_F32); 1024-point frames) ------------------------------------------------------
// The same kernel with the transform, split and merge in f32 (vp_fft32.inc).  Why it is a build of its own and not the default: the fp64
// kernel is bound by its instruction stream -- 613 fp64 vector instructions per lane and frame at 4 cycles each, two wavefronts per SIMD
// (223 registers) -- and an f32 vector instruction issues in 2 cycles once two wavefronts share the SIMD (MI355X_MICROARCH.md,
// "vector-instruction ISSUE cost"); it also needs half the registers (four wavefronts per SIMD instead of two) and half the LDS
// bytes per exchange.  The input is f32 and the output is f32 either way; what changes is the rounding inside: ~1e-7 relative per
// frame instead of ~1e-16, two orders below the 1e-4 the north_star allows -- but not the bit pattern of the default build, hence opt-in.
template <bool MAG>
__global__ __launch_bounds__(64 * NWV, 4) void vp_k_stft_fused32(VpStftArgs A)
{
    extern __shared__ double smem[];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int s = blockIdx.y, run = blockIdx.x;
    constexpr int N = 512;
    const int F = A.F, hop = A.hop, T = A.T;
    lds_f2 *xb = (lds_f2 *)smem + wv * 512;                                    // the wavefront's exchange buffer (4 KB) = its output slot (F floats)
    lds_f32 *slots = (lds_f32 *)smem;
    lds_f32 *carry = (lds_f32 *)smem + NWV * 1024;
    Fft32Addr L;
    fft32_addr_init(L, lane);
    // the double-precision tables, rounded once: the second step's twiddles (a row of eight per lane >> 3: 64 values) in LDS, the rest in registers
    lds_f2 *tw1L = (lds_f2 *)((lds_f32 *)smem + stft_lds_base(F, hop, 1) / 4);
    if (tid < 64) { const d2 t = ((const d2 *)A.tw1)[(tid >> 3) * 64 + (tid & 7)]; tw1L[tid] = f2{(float)t.x, (float)t.y}; }
    const lds_f2 *tw1 = tw1L + (lane >> 3) * 8;
    f2 tw2[8], wa[8], ws[4];
#pragma unroll
    for (int r = 0; r < 8; r++) {
        const d2 t2 = ((const d2 *)A.tw2)[lane * 8 + r], w = ((const d2 *)A.win)[lane + 64 * r];
        tw2[r] = f2{(float)t2.x, (float)t2.y}; wa[r] = f2{(float)w.x, (float)w.y};
    }
#pragma unroll
    for (int q = 0; q < 4; q++) { const d2 t = ((const d2 *)A.tws)[lane * 4 + q]; ws[q] = f2{(float)t.x, (float)t.y}; }
    const float c = (float)A.c;
    for (int i = tid; i < F - hop; i += 64 * NWV) carry[i] = 0.f;
    __syncthreads();

    const int rFirst = run * A.roundsPerRun;
    const int r0 = max(0, rFirst - (run > 0 ? A.haloRounds : 0));
    const int r1 = min(rFirst + A.roundsPerRun, A.nRounds);
    const float *xs = A.in + (size_t)s * T;
    // the frame's samples are requested a round ahead (16 registers this build can afford; +3 to +8 % on one box: 570 -> 590 M frames/s at
    // 256 streams x 65 536 samples, 675 -> 730 M at 4096 x 32 768)
    f2 xv[8];
    auto request = [&](int rd_) {
        const int f_ = rd_ * NWV + wv;
        if (rd_ < r1 && f_ < A.nFrames) {
            const float *x = xs + (size_t)f_ * hop;
            if (A.aligned) {
#pragma unroll
                for (int r = 0; r < 8; r++) xv[r] = *(const f2 *)(x + 2 * (lane + 64 * r));
            } else {
#pragma unroll
                for (int r = 0; r < 8; r++) xv[r] = f2{x[2 * (lane + 64 * r)], x[2 * (lane + 64 * r) + 1]};
            }
        }
    };
    request(r0);
    for (int rd = r0; rd < r1; rd++) {
        const int f = rd * NWV + wv;
        const bool live = f < A.nFrames;
        C8f z;
        if (live) {
#pragma unroll
            for (int r = 0; r < 8; r++) { z.re[r] = xv[r].x * wa[r].x; z.im[r] = xv[r].y * wa[r].y; }
        }
        request(rd + 1);
        if (live) {
            fft512f(z, xb, L, tw1, (const f2 *)tw2);
            RPairsT<float, 4> X;
            rfft_split_n<4>(z.re, z.im, z.re + 4, z.im + 4, xb, lane, (const f2 *)ws, X);
            if (MAG) {
                float *m = A.mag + ((size_t)s * A.nFrames + f) * (N + 1);
#pragma unroll
                for (int q = 0; q < 4; q++) {
                    const int k = 64 * q + lane;
                    m[k] = sqrtf(X.kr[q] * X.kr[q] + X.ki[q] * X.ki[q]);
                    m[N - k] = sqrtf(X.mr[q] * X.mr[q] + X.mi[q] * X.mi[q]);
                }
                if (lane == 0) m[N / 2] = sqrtf(X.hr * X.hr + X.hi * X.hi);
            }
            rfft_merge_conj_n<4>(z.re, z.im, z.re + 4, z.im + 4, xb, lane, (const f2 *)ws, X, c);
            fft512f(z, xb, L, tw1, (const f2 *)tw2);
            wave_sync();
#pragma unroll
            for (int r = 0; r < 8; r++) lds_put(xb, lane + 64 * r, z.re[r] * wa[r].x, -(z.im[r] * wa[r].y));
        } else {
#pragma unroll
            for (int r = 0; r < 8; r++) xb[lane + 64 * r] = f2{0.f, 0.f};
        }
        __syncthreads();
        stft_overlap_add<1024>(A, slots, carry, s, rd, rd >= rFirst, tid);
        __syncthreads();
    }
}

// single precision, 2048-point frames: vp_k_stft_fused2k's radix-2 step over two 512-point transforms, in f32.  The wavefront's output slot
// (8 KB) starts with its exchange buffer (4 KB); the window (f32, 8 KB) and the second step's twiddle rows are LDS copies: 47 KB per
// workgroup, three workgroups per CU.
template <bool MAG>
__global__ __launch_bounds__(64 * NWV, 3) void vp_k_stft_fused2k32(VpStftArgs A)
{
    extern __shared__ double smem[];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int s = blockIdx.y, run = blockIdx.x;
    constexpr int N = 1024;
    const int F = A.F, hop = A.hop, T = A.T;
    typedef float f4 __attribute__((ext_vector_type(4)));
    typedef __attribute__((address_space(3))) f4 lds_f4;
    lds_f32 *slots = (lds_f32 *)smem;                                          // slot w: 2048 floats at w * 2048
    lds_f32 *carry = slots + NWV * 2048;
    lds_f2 *xb = (lds_f2 *)(slots + wv * 2048);
    lds_f4 *winL = (lds_f4 *)((lds_f32 *)smem + stft_lds_base(F, hop, 0) / 4);              // [512] the window, four consecutive samples per entry
    lds_f2 *twL = (lds_f2 *)(winL + 512);                                                    // [8][8] W_64 rows (| [64][8] W_512 in the magnitude-dump build)
    Fft32Addr L;
    fft32_addr_init(L, lane);
    for (int i = tid; i < 64; i += 64 * NWV) { const d2 t = ((const d2 *)A.tw1)[(i >> 3) * 64 + (i & 7)]; twL[i] = f2{(float)t.x, (float)t.y}; }
    const lds_f2 *tw1p = twL + (lane >> 3) * 8;
    for (int i = tid; i < 512; i += 64 * NWV) {
        const d2 w0 = ((const d2 *)A.win)[2 * i], w1 = ((const d2 *)A.win)[2 * i + 1];
        winL[i] = f4{(float)w0.x, (float)w0.y, (float)w1.x, (float)w1.y};
    }
    // the third step's twiddles are the lane's own: registers (the magnitude-dump build, which is not the timed one, has none to spare
    // and reads them from an LDS copy)
    f2 wtop[8], ws[8], tw2r[8];
#pragma unroll
    for (int q = 0; q < 8; q++) {
        const d2 t = ((const d2 *)A.twTop)[lane * 8 + q], u = ((const d2 *)A.tws)[lane * 8 + q], v = ((const d2 *)A.tw2)[lane * 8 + q];
        wtop[q] = f2{(float)t.x, (float)t.y}; ws[q] = f2{(float)u.x, (float)u.y};
        if (MAG) twL[64 + lane * 8 + q] = f2{(float)v.x, (float)v.y}; else tw2r[q] = f2{(float)v.x, (float)v.y};
    }
    auto fft = [&](C8f &z_) {
        if (MAG) fft512f(z_, xb, L, tw1p, (const lds_f2 *)(twL + 64 + lane * 8)); else fft512f(z_, xb, L, tw1p, (const f2 *)tw2r);
    };
    const float c = (float)A.c;
    for (int i = tid; i < F - hop; i += 64 * NWV) carry[i] = 0.f;
    __syncthreads();

    const int rFirst = run * A.roundsPerRun;
    const int r0 = max(0, rFirst - (run > 0 ? A.haloRounds : 0));
    const int r1 = min(rFirst + A.roundsPerRun, A.nRounds);
    const float *xs = A.in + (size_t)s * T;
    for (int rd = r0; rd < r1; rd++) {
        const int f = rd * NWV + wv;
        const bool live = f < A.nFrames;
        lds_f4 *slot = (lds_f4 *)(slots + wv * 2048);
        if (live) {
            const float *x = xs + (size_t)f * hop;
            C8f e, o;
#pragma unroll
            for (int r = 0; r < 8; r++) {
                const int m = lane + 64 * r;
                f4 v;
                if (A.aligned) v = *(const f4 *)(x + 4 * m);
                else v = f4{x[4 * m], x[4 * m + 1], x[4 * m + 2], x[4 * m + 3]};
                const f4 w = winL[m];
                e.re[r] = v.x * w.x; e.im[r] = v.y * w.y;
                o.re[r] = v.z * w.z; o.im[r] = v.w * w.w;
            }
            fft(e);
            fft(o);
            float hr[8], hi[8];
#pragma unroll
            for (int q = 0; q < 8; q++) {
                const float tr = vp_fma(o.re[q], wtop[q].x, -(o.im[q] * wtop[q].y)), ti = vp_fma(o.re[q], wtop[q].y, o.im[q] * wtop[q].x);
                hr[q] = e.re[q] - tr; hi[q] = e.im[q] - ti;
                e.re[q] += tr; e.im[q] += ti;
            }
            RPairsT<float, 8> X;
            rfft_split_n<8>(e.re, e.im, hr, hi, xb, lane, (const f2 *)ws, X);
            if (MAG) {
                float *mg = A.mag + ((size_t)s * A.nFrames + f) * (N + 1);
#pragma unroll
                for (int q = 0; q < 8; q++) {
                    const int k = 64 * q + lane;
                    mg[k] = sqrtf(X.kr[q] * X.kr[q] + X.ki[q] * X.ki[q]);
                    mg[N - k] = sqrtf(X.mr[q] * X.mr[q] + X.mi[q] * X.mi[q]);
                }
                if (lane == 0) mg[N / 2] = sqrtf(X.hr * X.hr + X.hi * X.hi);
            }
            rfft_merge_conj_n<8>(e.re, e.im, hr, hi, xb, lane, (const f2 *)ws, X, c);
#pragma unroll
            for (int q = 0; q < 8; q++) {
                const float dr = e.re[q] - hr[q], di = e.im[q] - hi[q];
                e.re[q] += hr[q]; e.im[q] += hi[q];
                o.re[q] = vp_fma(dr, wtop[q].x, -(di * wtop[q].y)); o.im[q] = vp_fma(dr, wtop[q].y, di * wtop[q].x);
            }
            fft(e);
            fft(o);
            wave_sync();                                                       // (the slot starts with the exchange buffer)
#pragma unroll
            for (int r = 0; r < 8; r++) {
                const f4 w = winL[lane + 64 * r];
                slot[lane + 64 * r] = f4{e.re[r] * w.x, -(e.im[r] * w.y), o.re[r] * w.z, -(o.im[r] * w.w)};
            }
        } else {
#pragma unroll
            for (int r = 0; r < 8; r++) slot[lane + 64 * r] = f4{0.f, 0.f, 0.f, 0.f};
        }
        __syncthreads();
        stft_overlap_add(A, slots, carry, s, rd, rd >= rFirst, tid);
        __syncthreads();
    }
}

// hipFuncSetAttribute acts on the CURRENT device: every handle raises the phase-vocoder build's dynamic-LDS ceiling on its own device
// at create (vp_stft_create, behind hipSetDevice), and a failure is the caller's VP_ERR_HIP -- not a process-wide flag set once.
hipError_t vp_stft_prepare_device()
{
    return hipFuncSetAttribute((const void *)vp_k_stft_fused<true, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 512);
}

hipError_t vp_stft_launch(const VpStftArgs &a, int nStreams, int nRuns, hipStream_t st)
{
    const size_t lds = vp_stft_lds_bytes(a.F, a.hop, a.f32);
    const dim3 grid(nRuns, nStreams), block(64 * NWV);
    if (a.f32 && a.F == 2048) {
        // slots + carry as in the double-precision build, then the f32 copies of the window and of the second step's twiddle rows
        const size_t lds2k = vp_stft_lds_bytes(a.F, a.hop, 0) + 2048 * 4 + 64 * 8;
        if (a.mag) hipLaunchKernelGGL((vp_k_stft_fused2k32<true>), grid, block, lds2k + 512 * 8, st, a);
        else hipLaunchKernelGGL((vp_k_stft_fused2k32<false>), grid, block, lds2k, st, a);
    } else if (a.f32) {
        const size_t lds32 = lds + 64 * 8;                                      // + the second step's twiddle rows
        if (a.mag) hipLaunchKernelGGL((vp_k_stft_fused32<true>), grid, block, lds32, st, a);
        else hipLaunchKernelGGL((vp_k_stft_fused32<false>), grid, block, lds32, st, a);
    } else if (a.F == 2048) {
        const size_t lds2 = lds + (64 + 512) * 16;                             // + the LDS copies of the 512-point transform's twiddle tables
        if (a.mag) hipLaunchKernelGGL((vp_k_stft_fused2k<true>), grid, block, lds2, st, a);
        else hipLaunchKernelGGL((vp_k_stft_fused2k<false>), grid, block, lds2, st, a);
    } else if (a.pv)
        hipLaunchKernelGGL((vp_k_stft_fused<true, false>), grid, block, lds + stft_pv_lds_bytes(a.F), st, a);
    else if (a.mag)
        hipLaunchKernelGGL((vp_k_stft_fused<false, true>), grid, block, lds, st, a);
    else
        hipLaunchKernelGGL((vp_k_stft_fused<false, false>), grid, block, lds, st, a);
    return hipGetLastError();
}
