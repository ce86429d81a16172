// vp_common.h -- host/device shared layout of the batch plugin state.
//
// Data layout in HBM (S = streams of the handle):
//   voiceRing  f32 [S][inSize]      MyBuffer::mInputVoice, same PHYSICAL ring positions as the
//                                   reference (MyBuffer.cpp:34-65); float storage is exact because
//                                   the ring only ever holds float inputs widened to double (:74-105)
//   synthRing  f32 [S][2][inSize]   MyBuffer::mInputSynth
//   outAcc     f64 [S][outSize]     MyBuffer::mOutput; ONE accumulator for both channels: the
//                                   vocoder and the pitch corrector add the same value to every
//                                   channel (VocoderProcess.cpp:291-295, PitchProcess.cpp:332);
//                                   the channels only diverge in addSynth, applied at emit time
//   pitch      VpPitchState [S]     PitchProcess private members (PitchProcess.h:108-138)
//   eFrame/outEFrame/yFrame f64     per-stream frame buffers that live across blocks
//   EeArr      f64 [S][2][10]       VocoderProcess::EeVoiceArr / EeSynthArr
// Counters (inCounter/outCounter/currCounter, startSample, nChunk) advance identically for all
// streams, so they live on the host and travel as kernel arguments (VpCall).
#pragma once
#include <stddef.h>

#define VP_MARKS 64
#define VP_ORDER_MAX 100
#define VP_ORDER_MAX_SYNTH 30
#define VP_NOTES_STRIDE 89     // Notes.cpp:27 reserve(88) + the popped element

struct VpGeom {
    int S, N, F, H, C, cpf, W, h, toKeep, latency, inSize, outSize, tauMax;
    int eLen;            // toKeep + F + (cpf-1)*C : the part of eFrame the reference ever writes
    int orderPitch;      // lpcPitch as read at prepare (PitchProcess.cpp:70)
    int tau0;            // floor(fS/fMax) (PitchProcess.cpp:429)
    int bufferIdxMax;    // latency + N (PitchProcess.cpp:138)
    int xsSteps;         // chunk steps whose voice window is staged in LDS at once (pitch kernel)
    int htabGlobal;      // pitch kernel: the frame's Hann(2T+1) window is read from the global table instead of an LDS copy (batches of
                         // more than 256 streams: the 7 KB buy the staging of the block's later chunk steps within half a CU's LDS)
    double fs, delta, yinTol;
    double gateThrSum;   // smallest sum(x^2) over the ring for which 20log10(rms) >= -60 dB
    double levEps;       // pow(10,-9)  LPC.cpp:110
    double eeFloor;      // pow(10,-4)  VocoderProcess.cpp:270
};

struct VpCall {
    int inCounter, outCounter, currCounter;
    int vStart, nWin;            // vocoder windows start at vStart + j*h, j < nWin
    int pStart, nChunk0, nSteps; // pitch chunk steps start at pStart + j*C
    int pitchOn, vocOn, inplace;
    int fuseIngest, fuseEmit;    // this launch also runs the ingest+gate prologue / the emit epilogue
    int fftOff, fftWaves;        // VP_YIN_FFT: byte offset, in the launch's dynamic LDS, of [64-byte flag block | fftWaves x 8 KB exchange buffers]
                                 // for the certified form's cross-correlations by FFT (xcorr_fft_wave), 0 = the fused-multiply-add form;
                                 // wavefronts 1..fftWaves of the workgroup transform
    int yinCert;                 // 1: cross-correlation form of the difference function (fused multiply-adds)
    int iirFast;                 // 0: exact (reference summation order), 1: transposed-form fast IIR
    int ldsBytes;                // dynamic LDS of this launch (used by the -DVP_POISON_LDS diagnostic build only)
    int vocWin;                  // vocoder kernel: windows per round (= window slots in LDS).  The launch carries a whole
                                 // number of wavefronts per slot: wave r * vocWin + j works for window j in role r (role 0 owns
                                 // it; the others take their share of the autocorrelation passes and residual-FIR units)
    int inMono;                  // 0: input [S][3][N].  1: input [S][N], side-chain bus absent -> the synth ring takes zeros
                                 // (MyBuffer.cpp:93-102).  2: same, and the host knows the synth ring holds nothing but zeros
                                 // already (nothing to write, nothing to sum for its gate)
    int nBlocks;                 // pitch kernel with both fusions: consecutive blocks handled by this launch (>= 1);
                                 // the counters above describe the first, the kernel advances them itself
    int pitchLin;                // pitch kernel, combined multi-block plan: gates to VpDev::gateB, no emit (see VpDev::pLin)
    int ldsAcc;                  // pitch kernel: the launch carries vp_pitch_acc_lds_bytes() more dynamic LDS, in which the block's
                                 // slice of the output accumulator lives while the chunks add to it (one read and one write
                                 // of HBM per block instead of a read-modify-write per chunk)
};

// The per-block parameters of ONE stream (each stream is a plugin instance with its own treeState).  They live in
// the stream's device state -- the pitch kernel stages that in LDS anyway -- and are rewritten by the host whenever
// vp_set_params / vp_set_stream_params changed them (stream-ordered, in front of the block's kernels).
struct VpStreamParams {
    int orderVoice, orderSynth, key, dryOn, synthOn;
    int shiftOn;                                       // extension (no reference counterpart): fixed shift factor instead of the key's note
    double gainPitch, gainVoc, gainVoice, gainSynth;   // (double) of the float gains
    double shiftBeta;                                  // 2^(semitones/12), host libm
};

struct VpPitchState {
    int period, prevPeriod, prevVoicedPeriod, periodNew;
    int stMarkIdx, nAnMarksOv, nStMarksOv, gateOpen;
    double pitch, prevPitch, prevVoicedPitch, closestFreq, prevClosestFreq, beta;
    int nAn, nSt, nPrevAn, nPrevSt;
    int anMarks[VP_MARKS], stMarks[VP_MARKS], prevAnMarks[VP_MARKS], prevStMarks[VP_MARKS];
    double a[VP_ORDER_MAX + 1];
    VpStreamParams sp;
};

#define VP_FFT_TW_D2 (64 + 512 + 256)   // complex doubles of the wavefront FFT's twiddle tables (vp_fft.inc)

struct VpDev {
    float *voiceRing;
    float *synthRing;
    double *outAcc;
    int *gate;               // [S][2] voice open, synth(ch0) open
    VpPitchState *pitch;
    double *eFrame, *outEFrame, *yFrame;
    double *EeArr;
    double *hImp;            // [S][128] impulse response of the in-flight pitch frame's 1/A(z) (block-form IIR)
    const double *vocWin;    // [W]  anWindow == stWindow ("sine", VocoderProcess.cpp:125-129)
    const double *pitchStWin;// [F]
    const double *hannTab;   // hann(2T+1) for T = 1..tauMax, concatenated
    const int *hannOff;      // [tauMax+1]
    const double *notes;     // [13][VP_NOTES_STRIDE]
    const int *notesN;       // [13]
    const double *fftTw1, *fftTw2, *fftTws;   // per-lane twiddles of the wavefront FFT (vp_fft.inc): [64][8][2], [64][8][2], [64][4][2]
    unsigned long long *ub;  // [5]
    unsigned long long *dbg; // [64] phase timers of the -DVP_STAMPS diagnostic build
    double *outAcc2;         // [S][outSize] second accumulator, non-null in the emit stage while it may hold anything: in
                             // VP_IIR_FAST mode the pitch corrector can run BESIDE the vocoder pipeline (another HIP stream) and
                             // then adds into this one; emit merges (and clears) both
    // combined multi-block plan (pitch corrector AND vocoder, several blocks per call, VP_IIR_FAST): the pitch kernel runs first, ingests
    // the blocks, leaves each block's two gates in gateB and adds its chunks into the LINEAR accumulator pLin (sample t of the call at
    // pLin[s][t]) instead of the accumulator ring; the vocoder pipeline then works from a snapshot of the rings as they stood before
    // the call and its overlap-add/emit kernel takes pLin in
    int *gateB;              // [V2_MB_MAX][S][2]
    double *pLin;            // [S][pLinLen]
    unsigned int *fault;     // ONE word of pinned host memory (device address): a bounded inter-wavefront wait that ran out raises it
                             // (vp_timeout, vp_kernels.hip); the host looks at it on every process call and after every
                             // synchronisation of its own, fails the call with VP_ERR_TIMEOUT and poisons the handle
    int spinLimit;           // polls a bounded wait makes before it gives up (2^22 ~ a second; vp_debug_set_spin_limit shortens it for the test)
    const int *streamMap;    // launch of a cohort (streams whose pitchBool/vocBool histories differ from the others'):
                             // workgroup b serves stream streamMap[b]; nullptr (the normal case): stream b
};

#if defined(__HIPCC__)
#define VP_HD __host__ __device__
#else
#define VP_HD
#endif

// Pitch kernel, LDS: length (doubles) of the yinTemp and running-sum regions.  They double as scratch once the pitch is
// picked -- the PSOLA grain table (2 x VP_MARKS doubles + 5 x VP_MARKS ints from dY[0]), the exact recursion's history
// (cum[0 .. order)), the block-form IIR's impulse response and input (cum[128 .. 448)) -- so at low sample rates, where
// tauMax + 1 is smaller than that scratch, the regions are sized for the scratch instead.
VP_HD static inline int vp_dy_len(int tauMax) { const int need = 2 * VP_MARKS + (5 * VP_MARKS + 1) / 2; return tauMax + 1 > need ? tauMax + 1 : need; }
VP_HD static inline int vp_cum_len(int tauMax) { return tauMax + 1 > 448 ? tauMax + 1 : 448; }

// bytes of dynamic LDS vp_k_pitch needs for a geometry
VP_HD static inline size_t vp_pitch_lds_bytes(const VpGeom &g)
{
    size_t dbl = (size_t)(g.toKeep + g.F) + 12 + (size_t)(g.xsSteps - 1) * g.C + g.eLen + 2 * (size_t)g.F + (size_t)vp_dy_len(g.tauMax) + (size_t)vp_cum_len(g.tauMax) + 2 * (VP_ORDER_MAX + 1) + (2 * (size_t)g.tauMax + 4) + (g.htabGlobal ? 0 : 2 * (size_t)g.tauMax + 2);
    return dbl * sizeof(double) + 8 * 16 + sizeof(VpPitchState) + 64 + 64;
}
// extra dynamic LDS for the block's slice of the output accumulator (VpCall::ldsAcc) and, behind it, a deferred chunk's input
// and history (PitchLds::pend), placed behind vp_pitch_lds_bytes()
VP_HD static inline size_t vp_pitch_acc_lds_bytes(const VpGeom &g) { return ((size_t)g.N + g.C + 1 + g.C + 128) * sizeof(double) + 16; }

// ---- the wave-specialised pitch kernel (vp_pitch_ws.inc): control block and LDS carve, shared with the host's plan ----
#define WS_MAXI 24                      // chunk instances per block (steps + frame starts)
#define WS_MAXS 8                       // frame starts per block
#define WS_NBG 4                        // background wavefronts
#define WS_ORDER_MAX 24                 // lpcPitch the kernel serves (orders 16 .. 24: the _o24 builds, round 6)
// doubles per parity (the orders up to 15 keep round 5's carve to the byte: blocks of five chunk steps still fit the CU's LDS):
#define WS_AF(g) ((g).orderPitch > 15 ? 32 : 16)      // the frame's coefficients where the residual and the recursions read them
#define WS_HP(g) ((g).orderPitch > 16 ? 192 : 128)    // 64 zeros + the impulse response of 1/A(z) (64 samples; 128 for the orders above 16)
#define WS_HIST(g) ((g).orderPitch > 15 ? 32 : 16)    // the exact recursion's history

struct WsCtl {                          // ints in LDS, zeroed by thread 0 in the prologue
    int psDone;                         // instances whose producer work is complete
    int fillTurn;                       // instances whose windowed add is complete (the reference's order of additions)
    int bgPub, bgAdopt;                 // frame starts published by the background / adopted by the producers
    int prodBar, bgBar;                 // software barriers of the two groups (monotone counters)
    int xcDone, zeroDone, pDone, acDone, lpcDone, pickState, pickDone;   // background-internal, generation = start index + 1
    int tDone;                          // starts whose grain half-length T is final (pickT: per parity), for the producers' early Hann staging
    int pickT[2];
    int iirDone;                        // FAST: instances whose recursion is complete (the windowed add is another wavefront's)
    int pubMode;                        // of the published start: 0 gate closed, 1 no analysis marks, 2 full
    int lpcZ[2];                        // per parity: levinsonDurbin took the |r0| < 1e-9 branch
    int instMode[WS_MAXI];              // per instance, decided by the producers: 0 nothing, 1 output only, 2 recursion + output
    int ishareBG[4], ishareP[4];
    int fftFlag[4];
    int gNext[WS_MAXS + 1];             // per segment: the gather pass's next trip (taken by whichever wavefront comes for one)
    int gDone[WS_MAXI];                 // per instance: finished trips among those that start in its chunk's stretch of outEFrame
    int gNT[WS_MAXS + 1], gLo[WS_MAXS + 1], gHi[WS_MAXS + 1], gNg[WS_MAXS + 1];   // per segment, from its grain table: trips, first sample, end, grains
};

// The block's SCHEDULE (PitchProcess.cpp:171-189): per chunk step [Cont of the running frame], then, when a frame starts there,
// [Start].  It follows from the cohort's counters alone, so the host builds it and hands it over as a kernel argument (scalar
// loads; thread 0 used to build it in LDS while the workgroup waited at the prologue's barrier).  Frames alternate between the two
// parity buffers; the frame in flight at the block's entry has parity 0.  A SEGMENT is a maximal run of instances of one frame
// inside the block: the producers take a segment's residual, grain table and gather pass in one go.
struct VpWsSched {
    int nInst, nStart, nSeg;
    int instStep[WS_MAXI], instK[WS_MAXI], instPar[WS_MAXI], instStart[WS_MAXI];   // instStart: the start's index, or -1 (a Cont chunk)
    int startStep[WS_MAXS], startPar[WS_MAXS], startNeed[WS_MAXS];                 // startNeed: instances that must have been added to the
                                                                                    // output before the start's parity buffers are free
    int segA[WS_MAXS + 1], segB[WS_MAXS + 1];                                       // first and last instance of each segment
};
VP_HD static inline bool ws_build_sched(const VpGeom &g, int nChunk0, int nSteps, VpWsSched &sc)
{
    int nCh = nChunk0, par = 0, nI = 0, nS = 0, nG = 0;
    int lastUse[2] = {0, 0};                                                  // instances up to which a parity's buffers are in use
    for (int t = 0; t < nSteps; t++) {
        if (nCh != 0) {
            if (nI >= WS_MAXI) return false;
            if (nI == 0) { if (nG > WS_MAXS) return false; sc.segA[nG] = 0; sc.segB[nG] = 0; nG++; }   // the frame in flight opens the block
            sc.instStep[nI] = t; sc.instK[nI] = nCh; sc.instPar[nI] = par; sc.instStart[nI] = -1;
            sc.segB[nG - 1] = nI;
            nI++;
            lastUse[par] = nI;
        }
        if (nCh == g.cpf - 1) nCh = 0;
        if (nCh == 0) {
            if (nI >= WS_MAXI || nS >= WS_MAXS || nG > WS_MAXS) return false;
            par ^= 1;
            sc.instStep[nI] = t; sc.instK[nI] = 0; sc.instPar[nI] = par; sc.instStart[nI] = nS;
            sc.startStep[nS] = t; sc.startPar[nS] = par; sc.startNeed[nS] = lastUse[par];
            sc.segA[nG] = nI; sc.segB[nG] = nI; nG++;
            nI++; nS++;
            lastUse[par] = nI;
        }
        nCh += 1;
    }
    sc.nInst = nI; sc.nStart = nS; sc.nSeg = nG;
    return nI > 0;
}

// Several queued blocks in ONE launch of the wave-specialised kernel (round 6, vp_k_pitch_ws_mb): per block of the call its counters and
// which of the (few, cyclically recurring) schedules it runs -- all of it follows from the cohort's counters, so the host builds it.
#define WS_MB_MAX 16
#define WS_MB_SCHEDS 4
struct VpWsMb {
    int nBlocks;
    int schedOf[WS_MB_MAX], nChunk0[WS_MB_MAX], inCtr[WS_MB_MAX], outCtr[WS_MB_MAX], currCtr[WS_MB_MAX];
    VpWsSched sc[WS_MB_SCHEDS];
};

// dynamic LDS of the kernel: see the carve in pitch_ws_body (the host's vp_pitch_ws_lds_bytes mirrors it)
struct WsCarve {
    int xs, eF, fr, qtab, htab, P, dY, gtab, r, aPrev, hp, aF, xp, hist, tw, oA, st, ctl, end;   // offsets in doubles
};
VP_HD static inline int ws_even(int v) { return (v + 1) & ~1; }
VP_HD static inline WsCarve ws_carve(const VpGeom &g, int nSteps)
{
    WsCarve c;
    int o = 0;
    c.xs = o;    o += ws_even(g.toKeep + g.F + (nSteps - 1) * g.C + 4);
    c.eF = o;    o += ws_even(g.eLen);
    c.fr = o;    o += 4 * g.F + 2048;                                          // oE[0] yF[0] | 2 exchange buffers | yF[1] oE[1]
    c.qtab = o;  o += ws_even(2 * g.tauMax + 4);
    c.htab = o;  o += ws_even(2 * g.tauMax + 2);
    c.P = o;     o += ws_even(g.F + g.tauMax + 2 > 450 ? g.F + g.tauMax + 2 : 450);   // prefix sums; the running sum of the fallback
    c.dY = o;    o += ws_even(vp_dy_len(g.tauMax) + 1);
    c.gtab = o;  o += 2 * VP_MARKS + (6 * VP_MARKS + 1) / 2 + 1;          // the segment's grain table: 2 double + 6 int arrays
    o = ws_even(o);
    c.r = o;     o += 2 * 64;                                                  // per parity: r[0..15], the other half's sums at r[32..47]
    c.aPrev = o; o += 2 * (VP_ORDER_MAX + 1 + 1);                              // per parity: Levinson-Durbin's output
    c.hp = o;    o += 2 * WS_HP(g);
    c.aF = o;    o += 2 * WS_AF(g);
    c.xp = o;    o += (g.orderPitch > 15 ? 128 : 64);
    c.hist = o;  o += 2 * WS_HIST(g);
    c.tw = o;    o += 2 * VP_FFT_TW_D2;
    c.oA = o;    o += ws_even(g.N + g.C + 1);
    c.st = o;    o += 2 * ((int)(sizeof(VpPitchState) + 15) / 16 * 2);      // the producers' copy of the tracker state, the background's
    c.ctl = o;   o += ((int)sizeof(WsCtl) + 7) / 8;
    c.end = ws_even(o);
    return c;
}
VP_HD static inline size_t vp_pitch_ws_lds_bytes(const VpGeom &g, int nSteps) { return (size_t)ws_carve(g, nSteps).end * sizeof(double); }

// doubles of LDS one vocoder wavefront needs for a window of length W (see vp_k_vocoder)
VP_HD static inline size_t voc_wave_doubles(int W)
{
    return (size_t)4 * W + 2 * (VP_ORDER_MAX + 1) + 2 * (VP_ORDER_MAX_SYNTH + 1) + 2;    // A B Cc D | rV aV | rS aS | pad (even: 16-byte alignment)
}
#define VP_VOC_SHARED_DOUBLES(W) ((size_t)(W) + 20 + 16 + 8)
