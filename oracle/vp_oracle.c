/*
 * vp_oracle.c -- CPU restatement of the DamRsn/VocoderProject hot path (see vp_oracle.h for the
 * parity status: partially pinned, end-to-end PARITY UNPINNED).  TEST INFRASTRUCTURE ONLY.
 *
 * Plain C99, IEEE double, no FMA contraction (build with -ffp-contract=off), one object per
 * audio stream exactly like one plugin instance.  All file:line citations are relative to
 * /root/reference/Source/.
 *
 * Restated, not copied: the reference's std::vector/JUCE AudioBuffer objects become flat
 * arrays with explicit lengths; the places where the reference reads outside a vector's size
 *  are given explicit, documented behaviour and counted in ub[].
 */
#include "vp_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

#if defined(__x86_64__) || defined(__i386__)
#include <xmmintrin.h>
#define VPO_HAVE_MXCSR 1
#endif

/* ------------------------------------------------------------------------------------------ */
/* small fixed-capacity int vector standing in for std::vector<int> with reserve()d storage.     */
/* Contents beyond n are never cleared: clear() keeps them, copy-assign copies only n entries,    */
/* insert-at-begin shifts right.                                                    */
typedef struct {
    int v[VPO_MARK_CAP];
    int n;
} marks_t;

static void marks_clear(marks_t *m) { m->n = 0; }
static void marks_push(marks_t *m, int x, long *ub)
{
    if (m->n < VPO_MARK_CAP) m->v[m->n] = x;
    m->n++;
    if (m->n > 20 && ub) ub[4]++;      /* beyond the reference's reserve(): realloc there */
    if (m->n > VPO_MARK_CAP) m->n = VPO_MARK_CAP;
}
static void marks_insert_front(marks_t *m, int x, long *ub)
{
    int k = m->n < VPO_MARK_CAP ? m->n : VPO_MARK_CAP - 1;
    memmove(m->v + 1, m->v, (size_t)k * sizeof(int));
    m->v[0] = x;
    m->n = k + 1;
    if (m->n > 20 && ub) ub[4]++;
}
static void marks_assign(marks_t *dst, const marks_t *src)
{
    memcpy(dst->v, src->v, (size_t)src->n * sizeof(int));
    dst->n = src->n;
}
/* std::vector::back(); on an empty vector the reference reads the int just before the heap
 * block (undefined behaviour; on glibc x86-64 that is the high half of the malloc size word = 0).
 * The restatement defines it as 0 and counts the event. */
static int marks_back(const marks_t *m, long *ub)
{
    if (m->n > 0) return m->v[m->n - 1];
    if (ub) ub[2]++;
    return 0;
}

struct vpo {
    /* ---- AudioProcessorValueTreeState raw values (std::atomic<float>), PluginProcessor.cpp:37-73 */
    float gainPitch, gainVoice, gainSynth, gainVoc;
    float lpcVoice, lpcPitch, lpcSynth, keyPitch, pitchBool, vocBool;
    int prepared, ftz;
    long ub[5];

    /* ---- MyBuffer (MyBuffer.h:60-86) */
    int N, toKeep, latency, inSize, outSize;
    int inCounter, outCounter, currCounter;
    double fs;
    double *voice;      /* mInputVoice [1][inSize] */
    double *synth[2];   /* mInputSynth [2][inSize] */
    double *out[2];     /* mOutput     [2][outSize] */

    /* ---- VocoderProcess (VocoderProcess.h:36-107) */
    int W, h, vStart, orderVoice, orderSynth;
    double silenceDb;
    double *vAn, *vSt;
    double rV[VPO_ORDER_MAX + 1], aV[VPO_ORDER_MAX + 1], aPV[VPO_ORDER_MAX + 1];
    double rS[VPO_ORDER_MAX_SYNTH + 1], aS[VPO_ORDER_MAX_SYNTH + 1], aPS[VPO_ORDER_MAX_SYNTH + 1];
    double *eV, *eS, *vOut;
    double EeV, EeS, g;
    double EeVArr[10], EeSArr[10];

    /* ---- PitchProcess (PitchProcess.h:79-157) */
    int F, H, C, chunksPerFrame, pStart, nChunk, bufferIdxMax, tauMax, order;
    double fMin, fMax, delta, yinTol, overlap;
    int key;
    int period, prevPeriod, prevVoicedPeriod, periodNew;
    int shiftOn; double shiftBeta;              /* extension: fixed shift factor 2^(semitones/12) instead of the key's note */
    double pitch, prevPitch, prevVoicedPitch, closestFreq, prevClosestFreq, beta;
    int stMarkIdx, nAnMarksOv, nStMarksOv;
    double *yinTemp;               /* tauMax (+1 guard slot, see yin()) */
    marks_t anMarks, stMarks, prevAnMarks, prevStMarks;
    double a[VPO_ORDER_MAX + 1], aPrev[VPO_ORDER_MAX + 1], r[VPO_ORDER_MAX + 1];
    double *eFrame; int eFrameLen;
    double *pAn, *pSt;             /* anWindow (ones), stWindow */
    double *psolaWindow, *periodSamples, *xInterp;
    double *lin; int linCap;          /* linear copy of the YIN window */
    double *outEFrame, *yFrame;
    /* Notes (Notes.h:20-31) */
    double freq[VPO_NOTES_CAP + 1]; int nFreq; int notesKey;

    /* ---- trace */
    vpo_pitch_frame *trace; int traceCap, traceN;
};

/* ------------------------------------------------------------------------------------------ */
/* JUCE arithmetic surface (restated from JUCE 5.4.x semantics; parity unpinned)                */

/* Decibels::gainToDecibels<double>(gain, -100.0) */
double vpo_gain_to_db(double gain)
{
    const double minusInf = -100.0;
    if (gain > 0.0) {
        double d = log10(gain) * 20.0;
        return d > minusInf ? d : minusInf;
    }
    return minusInf;
}

/* Decibels::decibelsToGain<float>(dB, -59.0f): computed in float at every call site that feeds
 * a gain (PluginProcessor.cpp:227,230; VocoderProcess.cpp:294; PitchProcess.cpp:339). */
float vpo_db_to_gain_f(float dB)
{
    return dB > -59.0f ? powf(10.0f, dB * 0.05f) : 0.0f;
}

/* dsp::WindowingFunction<double>::fillWindowingTables(w, n, hann, normalise=false):
 * w[i] = 0.5 - 0.5 cos(2 i pi / (n-1)) -- JUCE's symmetric Hann. */
void vpo_hann(double *w, int n)
{
    for (int i = 0; i < n; i++) {
        double c = cos((double)(2 * (long)i) * 3.141592653589793238 / (double)(n - 1));
        w[i] = 0.5 - 0.5 * c;
    }
}

/* AudioBuffer<double>::getRMSLevel(ch, 0, n): sqrt(sum x^2 / n), sequential double sum. */
static double rms_level(const double *x, int n)
{
    double sum = 0.0;
    for (int i = 0; i < n; i++) {
        double s = x[i];
        sum += s * s;
    }
    return sqrt(sum / n);
}

/* ------------------------------------------------------------------------------------------ */
/* L0: LPC.cpp                                                                                   */

/* LPC.cpp:44-97 biaisedAutoCorr.  r[m] = (1/L) sum_{n<L-m} (x[n]w[n]) * x[n+m] * w[n+m]; the
 * accumulation order is outer n, inner m, so each r[m] is a left-to-right sum over n. */
void vpo_biased_autocorr(const double *ring, int inSize, int currCounter, int startSample,
                         int order, int wlen, const double *anWindow, double *r)
{
    for (int m = 0; m <= order; m++) r[m] = 0.0;
    for (int n = 0; n < wlen; n++) {
        int startIdx = (currCounter + startSample + n + inSize) % inSize;
        double tmp = ring[startIdx] * anWindow[n];                          /* :61 */
        for (int m = 0; m <= order; m++) {
            if (n < wlen - m) {
                int idx = startIdx + m;                                      /* :66-88 two-part wrap */
                if (idx >= inSize) idx -= inSize;
                r[m] += tmp * ring[idx] * anWindow[m + n];
            }
        }
    }
    for (int m = 0; m <= order; m++) r[m] /= (double)wlen;                  /* :93-96 */
}

/* LPC.cpp:107-148 levinsonDurbin. fabs() is the intended overload. */
void vpo_levinson_durbin(const double *r, double *a, double *aPrev, int order, int aLen)
{
    if (fabs(r[0]) < pow(10, -9)) {                                          /* :110-114 */
        for (int i = 0; i < aLen; i++) a[i] = 0.0;
        a[0] = 1.0;
        return;
    }
    a[0] = 1.0;
    a[1] = r[1] / r[0];
    for (int p = 2; p < order + 1; p++) {
        for (int j = 1; j < p; j++) aPrev[j] = a[j];
        double rho_a = 0.0, r_a = 0.0;
        for (int i = 1; i < p; i++) {
            rho_a += r[p - i] * a[i];
            r_a += r[i] * a[i];
        }
        double k = (r[p] - rho_a) / (r[0] - r_a);
        for (int i = 1; i < p; i++) a[i] = aPrev[i] - k * aPrev[p - i];
        a[p] = k;
    }
    for (int i = 1; i < order + 1; i++) a[i] *= -1.;                        /* :145-146 */
}

/* ------------------------------------------------------------------------------------------ */
/* L0: Notes.cpp                                                                                 */

/* Notes.cpp:43-70 buildFreqVect.  freq must hold VPO_NOTES_CAP+1 doubles.  The element removed
 * by pop_back() (:69) stays in memory at freq[size]; getClosestFreq reads it when the pitch is
 * above every table entry (:99 with idx == size), so it is kept. */
int vpo_notes_build(int key, double fMin, double fMax, double *freq)
{
    static const int intervals[7] = {2, 2, 1, 2, 2, 2, 1};
    int n = 0, i = 0;
    double f = 27.5;
    f = f * pow(2, (double)key / 12.0);
    double factorSemiTone = pow(2, 1.0 / 12);
    while (n == 0 || freq[n - 1] < fMax) {
        if (key != 12)
            f = f * pow(factorSemiTone, intervals[i % 7]);
        else
            f = f * factorSemiTone;
        if (f > fMin) {
            if (n < VPO_NOTES_CAP + 1) freq[n] = f;
            n++;
        }
        i += 1;
    }
    return n - 1;
}

/* Notes.cpp:79-110 getClosestFreq (table lookup part). */
double vpo_notes_closest(const double *freq, int size, double pitch)
{
    int lo = 0, hi = size;                      /* std::lower_bound: first element >= pitch */
    while (lo < hi) {
        int mid = lo + (hi - lo) / 2;
        if (freq[mid] < pitch) lo = mid + 1; else hi = mid;
    }
    int idx = lo;
    if (idx > 0) {
        if (fabs(freq[idx] - pitch) <= fabs(freq[idx - 1] - pitch))          /* :99, may read freq[size] */
            return freq[idx];
        return freq[idx - 1];
    }
    return freq[idx];
}

static double notes_get_closest(vpo *o, double pitch, int key)
{
    if (key != o->notesKey) {                                               /* :83-88 */
        o->notesKey = key;
        o->nFreq = vpo_notes_build(key, o->fMin, o->fMax, o->freq);
    }
    return vpo_notes_closest(o->freq, o->nFreq, pitch);
}

/* ------------------------------------------------------------------------------------------ */
/* L2: MyBuffer.cpp                                                                              */

static double voice_sample(const vpo *o, int idx)                           /* MyBuffer.cpp:141-153 */
{
    return o->voice[(o->currCounter + idx + o->inSize) % o->inSize];
}
static double synth_sample(const vpo *o, int ch, int idx)                   /* :161-172 */
{
    return o->synth[ch][(o->currCounter + idx + o->inSize) % o->inSize];
}
static void add_out_sample(vpo *o, int ch, int idx, double v)               /* :181-191 */
{
    o->out[ch][(o->outCounter + idx) % o->outSize] += v;
}

static int mybuffer_prepare(vpo *o, int N, int toKeep, int latency, double fs)  /* :34-65 */
{
    o->N = N; o->toKeep = toKeep; o->latency = latency; o->fs = fs;
    o->inSize = toKeep + N + latency;
    o->outSize = N + latency;
    free(o->voice); free(o->synth[0]); free(o->synth[1]); free(o->out[0]); free(o->out[1]);
    o->voice = calloc((size_t)o->inSize, sizeof(double));
    o->synth[0] = calloc((size_t)o->inSize, sizeof(double));
    o->synth[1] = calloc((size_t)o->inSize, sizeof(double));
    o->out[0] = calloc((size_t)o->outSize, sizeof(double));
    o->out[1] = calloc((size_t)o->outSize, sizeof(double));
    if (!o->voice || !o->synth[0] || !o->synth[1] || !o->out[0] || !o->out[1]) return -2;
    o->inCounter = toKeep + latency;
    o->outCounter = 0;
    o->currCounter = toKeep;
    return 0;
}

static void fill_input_buffers(vpo *o, const float *v, const float *s0, const float *s1) /* :74-105 */
{
    for (int i = 0; i < o->N; i++) o->voice[(o->inCounter + i) % o->inSize] = v[i];
    const float *s[2] = {s0, s1};
    for (int ch = 0; ch < 2; ch++)
        for (int i = 0; i < o->N; i++)
            o->synth[ch][(o->inCounter + i) % o->inSize] = s[ch] ? (double)s[ch][i] : 0.0;
}

/* MyBuffer.cpp:309-373 / :380-448.  All the wrap case analysis reduces to
 * out[ch][(outCounter+i)%outSize] += gain * in[(currCounter+i)%inSize], i = 0..N-1
 * (JUCE addFrom: d[i] += s[i] * gain). */
static void add_dry_voice(vpo *o, double gain)
{
    for (int ch = 0; ch < 2; ch++)
        for (int i = 0; i < o->N; i++)
            o->out[ch][(o->outCounter + i) % o->outSize] += o->voice[(o->currCounter + i) % o->inSize] * gain;
}
static void add_synth(vpo *o, double gain)
{
    for (int ch = 0; ch < 2; ch++)
        for (int i = 0; i < o->N; i++)
            o->out[ch][(o->outCounter + i) % o->outSize] += o->synth[ch][(o->currCounter + i) % o->inSize] * gain;
}

/* MyBuffer.cpp:113-133 fillOutputBuffer + clearOutput :218-228.  buffer.clear() zeroes all three
 * channels first, so ch2 returns 0. */
static void fill_output_buffer(vpo *o, float *ch0, float *ch1, float *ch2)
{
    float *dst[3] = {ch0, ch1, ch2};
    for (int c = 0; c < 3; c++)
        if (dst[c]) memset(dst[c], 0, (size_t)o->N * sizeof(float));
    for (int ch = 0; ch < 2; ch++) {
        for (int i = 0; i < o->N; i++) {
            int p = (o->outCounter + i) % o->outSize;
            if (dst[ch]) dst[ch][i] = (float)o->out[ch][p];
            o->out[ch][p] = 0.0;
        }
    }
    o->outCounter = (o->outCounter + o->N) % o->outSize;
    o->inCounter = (o->inCounter + o->N) % o->inSize;
    o->currCounter = (o->currCounter + o->N) % o->inSize;
}

/* ------------------------------------------------------------------------------------------ */
/* L1: VocoderProcess.cpp                                                                        */

#define VOC_PI 3.14159265      /* VocoderProcess.cpp:13 -- truncated literal, behaviour (Q9) */

/* VocoderProcess.cpp:95-135 setWindows("sine") */
int vpo_vocoder_windows(int wlen, int hop, double *anWindow, double *stWindow)
{
    double overlap = (double)(wlen - hop) / (double)wlen;
    double overlapFactor = 1.0;
    if (fabs(overlap - 0.75) < pow(10, -10)) overlapFactor = 1.0 / sqrt(2);
    if (fabs(overlap - 0.75) > pow(10, -10) && fabs(overlap - 0.5) > pow(10, -10)) return -1;
    for (int i = 0; i < wlen; i++) {
        anWindow[i] = overlapFactor * sin((i + 0.5) * VOC_PI / (double)wlen);
        stWindow[i] = overlapFactor * sin((i + 0.5) * VOC_PI / (double)wlen);
    }
    return 0;
}

static int vocoder_prepare(vpo *o, int wlen, int hop, double silenceDb)     /* :35-71 */
{
    o->W = wlen; o->h = hop; o->vStart = 0;
    o->orderVoice = (int)o->lpcVoice;
    o->orderSynth = (int)o->lpcSynth;
    o->silenceDb = silenceDb;
    for (int i = 0; i <= VPO_ORDER_MAX; i++) o->rV[i] = o->aV[i] = o->aPV[i] = 1.0;
    for (int i = 0; i <= VPO_ORDER_MAX_SYNTH; i++) o->rS[i] = o->aS[i] = o->aPS[i] = 1.0;
    free(o->eV); free(o->eS); free(o->vOut); free(o->vAn); free(o->vSt);
    o->eV = calloc((size_t)wlen, sizeof(double));
    o->eS = calloc((size_t)wlen, sizeof(double));
    o->vOut = calloc((size_t)wlen, sizeof(double));
    o->vAn = calloc((size_t)wlen, sizeof(double));
    o->vSt = calloc((size_t)wlen, sizeof(double));
    o->g = 0.0; o->EeS = 1.0; o->EeV = 0.0;
    memset(o->EeVArr, 0, sizeof o->EeVArr);
    memset(o->EeSArr, 0, sizeof o->EeSArr);
    return vpo_vocoder_windows(wlen, hop, o->vAn, o->vSt);
}

/* VocoderProcess.cpp:235-251 filterFIR: zero history to the left of the window. */
static void voc_filter_fir(vpo *o, int synth, double *e, const double *a, int order, double *E)
{
    *E = 0.0;
    for (int i = 0; i < o->W; i++) {
        double x = synth ? synth_sample(o, 0, o->vStart + i) : voice_sample(o, o->vStart + i);
        e[i] = a[0] * x * o->vAn[i];
        for (int k = 1; k < order + 1; k++) {
            if (i - k >= 0) {
                double xk = synth ? synth_sample(o, 0, o->vStart + i - k) : voice_sample(o, o->vStart + i - k);
                e[i] += xk * o->vAn[i - k] * a[k];
            } else
                break;
        }
        *E += e[i] * e[i];
    }
}

/* VocoderProcess.cpp:260-297 filterIIR + shift/sum :301-327 */
static void voc_filter_iir(vpo *o, const double *a, int order)
{
    for (int i = 9; i > 0; i--) { o->EeVArr[i] = o->EeVArr[i - 1]; o->EeSArr[i] = o->EeSArr[i - 1]; }
    o->EeVArr[0] = o->EeV;
    o->EeSArr[0] = o->EeS;
    if (o->EeS > pow(10, -4)) {
        double sv = 0, ss = 0;
        for (int i = 0; i < 10; i++) sv += o->EeVArr[i];
        for (int i = 0; i < 10; i++) ss += o->EeSArr[i];
        o->g = sqrt(sv / ss);
    } else
        o->g = 0.0;
    for (int i = 0; i < o->W; i++) {
        o->vOut[i] = o->g * o->eS[i];
        for (int k = 1; k < order + 1; k++) {
            if (i - k >= 0)
                o->vOut[i] -= o->vOut[i - k] * a[k];
            else
                break;
        }
    }
    for (int ch = 0; ch < 2; ch++)
        for (int i = 0; i < o->W; i++)
            add_out_sample(o, ch, o->vStart + i, vpo_db_to_gain_f(o->gainVoc) * o->vOut[i] * o->vSt[i]); /* :291-295 */
}

static void voc_process_window(vpo *o)                                      /* :190-223 */
{
    o->orderSynth = (int)o->lpcSynth;                                       /* setOrderSynth :156-165 */
    o->orderVoice = (int)o->lpcVoice;                                       /* setOrderVoice :141-150 */
    double rmsVoiceDb = vpo_gain_to_db(rms_level(o->voice, o->inSize));     /* MyBuffer.cpp:258-261 */
    double rmsSynthDb = vpo_gain_to_db(rms_level(o->synth[0], o->inSize));  /* :299-302 (ch0 only) */
    if (rmsVoiceDb < o->silenceDb || rmsSynthDb < o->silenceDb) return;
    vpo_biased_autocorr(o->voice, o->inSize, o->currCounter, o->vStart, o->orderVoice, o->W, o->vAn, o->rV);
    vpo_levinson_durbin(o->rV, o->aV, o->aPV, o->orderVoice, VPO_ORDER_MAX + 1);
    vpo_biased_autocorr(o->synth[0], o->inSize, o->currCounter, o->vStart, o->orderSynth, o->W, o->vAn, o->rS);
    vpo_levinson_durbin(o->rS, o->aS, o->aPS, o->orderSynth, VPO_ORDER_MAX_SYNTH + 1);
    voc_filter_fir(o, 0, o->eV, o->aV, o->orderVoice, &o->EeV);
    voc_filter_fir(o, 1, o->eS, o->aS, o->orderSynth, &o->EeS);
    voc_filter_iir(o, o->aV, o->orderVoice);
}

static void vocoder_process(vpo *o)                                         /* :173-183 */
{
    while (o->vStart < o->N) {
        voc_process_window(o);
        o->vStart += o->h;
    }
    o->vStart -= o->N;
}

/* ------------------------------------------------------------------------------------------ */
/* L1: PitchProcess.cpp                                                                          */

/* PitchProcess.cpp:889-905 buildWindows: half-Hann | ones | half-Hann (JUCE symmetric Hann). */
int vpo_pitch_st_window(int frameLen, int hop, double *stWindow)
{
    double overlap = ((double)(frameLen - hop)) / ((double)frameLen);
    int ov = (int)round(overlap * frameLen);
    int nh = 2 * ov;
    if (nh > frameLen || nh < 2) return -1;
    double *hw = malloc((size_t)nh * sizeof(double));
    if (!hw) return -2;
    vpo_hann(hw, nh);
    int ones = frameLen - nh;
    for (int i = 0; i < ov; i++) stWindow[i] = hw[i];
    for (int i = 0; i < ones; i++) stWindow[ov + i] = 1.0;
    for (int i = 0; i < ov; i++) stWindow[ov + ones + i] = hw[ov + i];
    free(hw);
    return 0;
}

static int pitch_prepare(vpo *o, double fS, double fMin, double fMax, int frameLen, int hop,
                         double silenceDb)                                  /* :62-128 */
{
    o->fMin = fMin; o->fMax = fMax; o->F = frameLen; o->H = hop;
    o->order = (int)o->lpcPitch;                                            /* read once, Q7 */
    o->overlap = ((double)(frameLen - hop)) / ((double)frameLen);
    o->delta = 0.94; o->pitch = 0; o->prevPitch = 0; o->period = 0; o->periodNew = 0;
    o->prevVoicedPeriod = 0; o->prevPeriod = 0; o->beta = 1; o->yinTol = 0.25;
    o->shiftOn = 0; o->shiftBeta = 1;                                      /* extension: off after every prepare */
    o->prevVoicedPitch = 0; o->closestFreq = 0; o->prevClosestFreq = 0;   /* uninitialised in the reference */
    o->stMarkIdx = 0; o->nAnMarksOv = 0; o->nStMarksOv = 0;
    o->pStart = 0; o->bufferIdxMax = 0; o->silenceDb = silenceDb;
    o->nChunk = 0;
    o->C = frameLen - hop;
    if (o->C <= 0) return -3;
    o->chunksPerFrame = frameLen / o->C;
    if (frameLen % o->C != 0) return -3;                                   /* SURVEY.md section 8 constraint */
    o->key = (int)o->keyPitch;
    o->notesKey = o->key;
    o->nFreq = vpo_notes_build(o->key, fMin, fMax, o->freq);                /* notes.prepare :98 */
    o->tauMax = (int)ceil(fS / fMin);
    free(o->yinTemp); free(o->outEFrame); free(o->yFrame); free(o->pAn); free(o->pSt);
    free(o->psolaWindow); free(o->periodSamples); free(o->xInterp);
    o->yinTemp = calloc((size_t)o->tauMax + 1, sizeof(double));
    o->outEFrame = calloc((size_t)frameLen, sizeof(double));
    o->yFrame = calloc((size_t)frameLen, sizeof(double));
    o->pAn = malloc((size_t)frameLen * sizeof(double));
    o->pSt = malloc((size_t)frameLen * sizeof(double));
    o->psolaWindow = calloc((size_t)2 * o->tauMax + 3, sizeof(double));
    o->periodSamples = calloc((size_t)2 * o->tauMax + 3, sizeof(double));
    o->xInterp = calloc((size_t)2 * o->tauMax + 3, sizeof(double));
    memset(&o->anMarks, 0, sizeof o->anMarks); memset(&o->stMarks, 0, sizeof o->stMarks);
    memset(&o->prevAnMarks, 0, sizeof o->prevAnMarks); memset(&o->prevStMarks, 0, sizeof o->prevStMarks);
    memset(o->a, 0, sizeof o->a); memset(o->aPrev, 0, sizeof o->aPrev); memset(o->r, 0, sizeof o->r);
    for (int i = 0; i < frameLen; i++) o->pAn[i] = 1.0;
    return vpo_pitch_st_window(frameLen, hop, o->pSt);
}

static void pitch_prepare2(vpo *o)                                          /* :134-141 */
{
    o->eFrameLen = o->inSize + (o->chunksPerFrame - 1) * o->C;
    free(o->eFrame);
    o->eFrame = calloc((size_t)o->eFrameLen, sizeof(double));
    o->bufferIdxMax = o->latency + o->N;
}

static void pitch_silence(vpo *o)                                           /* :146-158 */
{
    marks_clear(&o->anMarks);
    marks_clear(&o->stMarks);
    o->prevPitch = 0; o->prevPeriod = 0; o->pitch = 0; o->period = 0;
}

/* PitchProcess.cpp:350-403 computeYinTemp on a linear signal.  d[k] accumulates over i = 0..F-1
 * in order for every k; pow(x,2) == x*x. */
void vpo_yin_temp_linear(const double *x, int frameLen, int tauMax, double *yinTemp)
{
    for (int k = 0; k < tauMax; k++) yinTemp[k] = 0.0;
    for (int i = 0; i < frameLen; i++) {
        double value_i = x[i];
        for (int k = 0; k < tauMax; k++) {
            double d = value_i - x[i + k];
            yinTemp[k] += d * d;
        }
    }
    yinTemp[0] = 1.0;
    double tmp = 0;
    for (int k = 1; k < tauMax; k++) {
        tmp += yinTemp[k];
        yinTemp[k] *= k / tmp;
    }
}

static void compute_yin_temp(vpo *o)
{
    /* The reference walks the ring with a two-part wrap (:369-386); the window it reads,
     * idx in [startSample - tauMax, startSample + F + tauMax - 1), is first copied to a linear
     * scratch here so that the inner loop over k vectorises like the reference's does
     * (#pragma clang loop vectorize, :371).  Each yinTemp[k] is still its own left-to-right sum. */
    const int n = o->F + o->tauMax;
    double *lin;
    if (n > o->linCap) {
        free(o->lin);
        o->lin = malloc((size_t)n * sizeof(double));
        o->linCap = n;
    }
    lin = o->lin;
    for (int j = 0; j < n; j++) lin[j] = voice_sample(o, o->pStart - o->tauMax + j);
    double *restrict y = o->yinTemp;
    const int tauMax = o->tauMax;
    for (int k = 0; k < tauMax; k++) y[k] = 0.0;
    for (int i = 0; i < o->F; i++) {
        const double value_i = lin[i];                                       /* :364 */
        const double *restrict w = lin + i;
        for (int k = 0; k < tauMax; k++) {
            double d = value_i - w[k];
            y[k] += d * d;
        }
    }
    o->yinTemp[0] = 1.0;                                                     /* :395 */
    double tmp = 0;
    for (int k = 1; k < o->tauMax; k++) {
        tmp += o->yinTemp[k];
        o->yinTemp[k] *= k / tmp;
    }
}

/* PitchProcess.cpp:429-447 threshold walk.  yinTemp[tauMax] can be read by the inner while when
 * tau == tauMax-1 (one past the vector; on glibc that is the next chunk's size word, a tiny
 * positive double, ~0 under DAZ): callers provide a guard slot holding 0.0. */
int vpo_yin_pick(const double *yinTemp, int tauMax, double fS, double fMax, double yinTol)
{
    int tau = (int)floor(fS / fMax);
    while (tau < tauMax) {
        if (yinTemp[tau] < yinTol) {
            while (yinTemp[tau + 1] < yinTemp[tau]) {
                tau += 1;
                if (tau + 1 >= tauMax) break;
            }
            return tau;
        }
        tau += 1;
    }
    return 0;
}

static void pitch_yin(vpo *o)                                               /* :411-448 */
{
    o->prevPeriod = o->period;
    o->prevPitch = o->pitch;
    if (o->pitch > 1) {
        o->prevVoicedPeriod = o->period;
        o->prevVoicedPitch = o->pitch;
    }
    o->pitch = 0; o->period = 0;
    compute_yin_temp(o);
    o->yinTemp[o->tauMax] = 0.0;                                             /* guard slot, see vpo_yin_pick */
    int tau = vpo_yin_pick(o->yinTemp, o->tauMax, o->fs, o->fMax, o->yinTol);
    if (tau > 0) {
        if (tau >= o->tauMax) o->ub[3]++;
        o->pitch = o->fs / tau;
        o->period = tau;
    }
}

/* PitchProcess.cpp:752-776 argExt: first strict minimum (or maximum) of the raw voice. */
static int arg_ext(const vpo *o, int idxStart, int idxEnd, int min)
{
    double ext = voice_sample(o, o->pStart + idxStart);
    int arg = idxStart;
    for (int i = idxStart + 1; i < idxEnd; i++) {
        double v = voice_sample(o, o->pStart + i);
        if (min ? (v < ext) : (v > ext)) { ext = v; arg = i; }
    }
    return arg;
}

static int imin(int a, int b) { return a < b ? a : b; }
static int imax(int a, int b) { return a > b ? a : b; }

static void pitch_marks(vpo *o)                                             /* :455-567 */
{
    const int valley = 1;                                                    /* :124 */
    marks_assign(&o->prevAnMarks, &o->anMarks);
    marks_clear(&o->anMarks);
    o->nAnMarksOv = 0;
    for (int i = 0; i < o->prevAnMarks.n; i++) o->prevAnMarks.v[i] -= o->H;
    for (int i = 0; i < o->prevAnMarks.n; i++)
        if (o->prevAnMarks.v[i] >= 0) o->nAnMarksOv += 1;
    int searchLeft = 0;
    int t, l_lim, r_lim, lastMark, sw_c, sw_f;
    marks_t *an = &o->anMarks;
    if (o->pitch > 1) {
        sw_c = (int)floor(o->delta * o->period);
        sw_f = (int)ceil((2.0 - o->delta) * o->period);
        if (o->prevPitch > 1) {
            if (o->nAnMarksOv == 0) {
                lastMark = marks_back(&o->prevAnMarks, o->ub);
                l_lim = imax(lastMark + imin(sw_c, (int)floor(o->delta * imin(o->prevPeriod, o->period))), 0);
                r_lim = imin(lastMark + imax(sw_f, (int)ceil((2 - o->delta) * imax(o->prevPeriod, o->period))), o->F);
                t = arg_ext(o, l_lim, r_lim, valley);
            } else
                t = o->prevAnMarks.v[o->prevAnMarks.n - o->nAnMarksOv];
        } else {
            searchLeft = 1;
            t = arg_ext(o, 0, o->F, valley);
        }
        marks_push(an, t, o->ub);
        while (an->v[an->n - 1] + sw_c < o->F) {                            /* search right :505-519 */
            int back = an->v[an->n - 1];
            if (back + sw_f < o->F)
                marks_push(an, arg_ext(o, back + sw_c, back + sw_f, valley), o->ub);
            else {
                if (back + o->period < o->F)
                    marks_push(an, arg_ext(o, back + sw_c, o->F, valley), o->ub);
                break;
            }
        }
        if (searchLeft) {                                                    /* :522-539 */
            while (an->v[0] - sw_c > 0) {
                int front = an->v[0];
                if (front - sw_f >= 0)
                    marks_insert_front(an, arg_ext(o, front - sw_f, front - sw_c, valley), o->ub);
                else {
                    if (front - o->period >= 0)
                        marks_insert_front(an, arg_ext(o, 0, front - sw_c, valley), o->ub);
                    break;
                }
            }
        }
    } else {
        if (o->prevAnMarks.n != 0) {                                         /* :545-565 */
            if (o->nAnMarksOv > 0) {
                for (int i = 0; i < o->nAnMarksOv; i++)
                    marks_push(an, o->prevAnMarks.v[o->prevAnMarks.n - o->nAnMarksOv + i], o->ub);
            } else
                marks_push(an, marks_back(&o->prevAnMarks, o->ub) + o->prevVoicedPeriod, o->ub);
            /* the reference asserts prevVoicedPeriod > 0 here (:555-559); with 0 it would loop forever */
            if (o->prevVoicedPeriod > 0)
                while (an->v[an->n - 1] + o->prevVoicedPeriod < o->F)
                    marks_push(an, an->v[an->n - 1] + o->prevVoicedPeriod, o->ub);
        }
    }
}

static void place_st_marks(vpo *o)                                          /* :573-658 */
{
    marks_assign(&o->prevStMarks, &o->stMarks);
    marks_clear(&o->stMarks);
    o->nStMarksOv = 0;
    for (int i = 0; i < o->prevStMarks.n; i++) o->prevStMarks.v[i] -= o->H;
    if (o->anMarks.n == 0) return;
    for (int i = 0; i < o->prevStMarks.n; i++)
        if (o->prevStMarks.v[i] >= 0) o->nStMarksOv += 1;
    int firstMark;
    o->prevClosestFreq = o->closestFreq;
    if (o->pitch > 1) {
        o->closestFreq = notes_get_closest(o, o->pitch, o->key);
        o->beta = o->closestFreq / o->pitch;
        if (o->shiftOn) {                      /* extension, no reference counterpart: fixed interval (vpo_set_pitch_shift) */
            o->beta = o->shiftBeta;
            o->closestFreq = o->beta * o->pitch;
        }
        o->periodNew = (int)round(o->period / o->beta);
    } else {
        o->closestFreq = 0;
        o->periodNew = o->prevVoicedPeriod;
    }
    if (o->periodNew <= 0) return;                                          /* reference asserts :604-608 */
    if (o->pitch > 1) {
        if (o->prevPitch > 1) {
            if (o->nStMarksOv > 0)
                firstMark = o->prevStMarks.v[o->prevStMarks.n - o->nStMarksOv];
            else if (marks_back(&o->prevStMarks, o->ub) + o->periodNew >= 0)
                firstMark = marks_back(&o->prevStMarks, NULL) + o->periodNew;
            else
                firstMark = o->anMarks.v[0];
        } else
            firstMark = o->anMarks.v[0];
    } else {
        if (o->prevStMarks.n == 0) return;
        if (o->nStMarksOv > 0)
            firstMark = o->prevStMarks.v[o->prevStMarks.n - o->nStMarksOv];
        else {
            int n = 1;
            while (marks_back(&o->prevStMarks, NULL) + n * o->periodNew < 0) n += 1;
            firstMark = marks_back(&o->prevStMarks, NULL) + n * o->periodNew;
        }
    }
    marks_push(&o->stMarks, firstMark, o->ub);
    while (o->stMarks.v[o->stMarks.n - 1] + o->periodNew < o->F)
        marks_push(&o->stMarks, o->stMarks.v[o->stMarks.n - 1] + o->periodNew, o->ub);
}

/* PitchProcess.cpp:280-302 filterFIR: FIR with the ring's real history, stops at -samplesToKeep. */
static void pitch_filter_fir(vpo *o, int startIdxBuf, int samplesToFilter, int startIdxE)
{
    int minBufIdx = -o->toKeep;
    for (int i = 0; i < samplesToFilter; i++) {
        double e = o->a[0] * voice_sample(o, o->pStart + startIdxBuf + i);
        for (int k = 1; k < o->order + 1; k++) {
            if (o->pStart + startIdxBuf + i - k >= minBufIdx)
                e += voice_sample(o, o->pStart + startIdxBuf + i - k) * o->a[k];
            else
                break;
        }
        o->eFrame[startIdxE + i] = e;
    }
}

static void pitch_filter_iir(vpo *o)                                        /* :307-322 */
{
    int shift = o->nChunk * o->C;
    for (int i = 0; i < o->C; i++) {
        double y = o->outEFrame[i + shift];
        for (int k = 1; k < o->order + 1; k++) {
            if (i + shift - k >= 0)
                y -= o->yFrame[i + shift - k] * o->a[k];
            else
                break;
        }
        o->yFrame[i + shift] = y;
    }
}

static void pitch_fill_output(vpo *o)                                       /* :328-342 */
{
    for (int ch = 0; ch < 2; ch++)
        for (int i = 0; i < o->C; i++) {
            float gain = o->gainPitch;
            add_out_sample(o, ch, o->pStart + i,
                           o->yFrame[i + o->nChunk * o->C] * o->pSt[i + o->nChunk * o->C] * vpo_db_to_gain_f(gain));
        }
}

/* PitchProcess.cpp:788-831 getClosestAnMarkIdx.  abs() here is on ints. */
static int closest_an_mark_idx(vpo *o, int stMark, int periodPsola)
{
    const marks_t *an = &o->anMarks;
    int lo = 0, hi = an->n;                                                  /* std::lower_bound */
    while (lo < hi) {
        int mid = lo + (hi - lo) / 2;
        if (an->v[mid] < stMark) lo = mid + 1; else hi = mid;
    }
    int idx = lo;
    int avail = o->bufferIdxMax - o->pStart;
    int sh = o->nChunk * o->C;
    if (idx > 0 && idx < an->n) {
        if (abs(an->v[idx] - stMark) <= abs(an->v[idx - 1] - stMark) && an->v[idx] + periodPsola - sh < avail)
            return idx;
        if (an->v[idx - 1] + periodPsola - sh < avail)
            return idx - 1;
        if (idx - 2 > 0)
            return idx - 2;
        return -o->nAnMarksOv - 1;                                           /* :812, leads to Q3 */
    }
    if (idx == 0) return 0;
    /* idx == size: anMarks[idx] reads the stale slot past the end (Q2) */
    o->ub[0]++;
    {
        int stale = idx < VPO_MARK_CAP ? an->v[idx] : 0;
        if (stale + periodPsola - sh < avail) return idx - 1;
        if (idx - 2 >= 0) return idx - 2;
        return idx - 1;                                                      /* reference asserts :824 */
    }
}

/* PitchProcess.cpp:842-870 interp: linear interpolation of (x,y) onto the integer grid
 * [startIdx, stopIdx), accumulated into outEFrame.  x is strictly increasing, so the restricted
 * lower_bound (:853-856) equals the global one (Q4 is benign). */
static void psola_interp(vpo *o, int n, int startIdx, int stopIdx)
{
    const double *x = o->xInterp, *y = o->periodSamples;
    int lb = 0;
    for (int i = startIdx; i < stopIdx; i++) {
        if (i >= x[0] && i <= x[n - 1]) {
            while (lb < n && x[lb] < i) lb++;
            double value;
            if (lb > 0)
                value = y[lb - 1] + (y[lb] - y[lb - 1]) / (x[lb] - x[lb - 1]) * (i - x[lb - 1]);
            else
                value = y[lb];
            o->outEFrame[i] += value;
            if (lb > 0) lb -= 1;                                             /* startSearchIt = lb - 1 */
        } else if (i > x[n - 1])
            break;
    }
}

static void pitch_psola(vpo *o)                                             /* :665-741 */
{
    int T = (o->pitch > 1) ? o->period : o->prevVoicedPeriod;
    int n = 2 * T + 1;
    vpo_hann(o->psolaWindow, n);                                             /* fillPsolaWindow :878-882 */
    while (o->stMarkIdx < o->stMarks.n) {
        int stMark = o->stMarks.v[o->stMarkIdx];
        if (stMark - T >= (o->nChunk + 1) * o->C) break;                    /* :685 */
        int clIdx = closest_an_mark_idx(o, stMark, T);
        int clAnMark;
        if (clIdx >= 0)
            clAnMark = o->anMarks.v[clIdx];
        else {                                                               /* Q3: size - clIdx, past the end */
            int j = o->prevAnMarks.n - clIdx;
            o->ub[1]++;
            clAnMark = (j >= 0 && j < VPO_MARK_CAP) ? o->prevAnMarks.v[j] : 0;
        }
        int first = (o->stMarkIdx == 0);
        int last = (o->stMarkIdx == o->stMarks.n - 1);
        for (int j = 0; j < n; j++) {
            int src = o->toKeep + clAnMark - T + j;
            double e = (src >= 0 && src < o->eFrameLen) ? o->eFrame[src] : 0.0;
            double w = o->psolaWindow[j];
            if (first)                                                       /* :707-716 (first wins when size==1) */
                o->periodSamples[j] = (j < T) ? e : e * w;
            else if (last)                                                   /* :721-731 */
                o->periodSamples[j] = (j < T) ? e * w : e;
            else                                                             /* :696-700 */
                o->periodSamples[j] = e * w;
            o->xInterp[j] = stMark + (-T + j) / o->beta;
        }
        int startIdx = imax((int)floor(o->xInterp[0]), 0);
        int stopIdx = imin((int)ceil(o->xInterp[n - 1]), o->F);
        psola_interp(o, n, startIdx, stopIdx);
        o->stMarkIdx += 1;
    }
}

static void trace_frame(vpo *o, int gated)
{
    if (o->traceN >= o->traceCap) return;
    vpo_pitch_frame *t = &o->trace[o->traceN++];
    memset(t, 0, sizeof *t);
    t->gated = gated;
    t->period = o->period; t->prevPeriod = o->prevPeriod; t->prevVoicedPeriod = o->prevVoicedPeriod;
    t->periodNew = o->periodNew; t->pitch = o->pitch; t->prevPitch = o->prevPitch; t->beta = o->beta;
    t->closestFreq = o->closestFreq;
    t->nAn = o->anMarks.n; t->nSt = o->stMarks.n;
    memcpy(t->anMarks, o->anMarks.v, sizeof t->anMarks);
    memcpy(t->stMarks, o->stMarks.v, sizeof t->stMarks);
    memcpy(t->a, o->a, sizeof t->a);
}

static void pitch_chunk_start(vpo *o)                                       /* :203-247 */
{
    o->key = (int)o->keyPitch;
    if (vpo_gain_to_db(rms_level(o->voice, o->inSize)) < o->silenceDb) {
        marks_clear(&o->anMarks);
        o->prevPitch = 0;
        trace_frame(o, 1);
        return;
    }
    memset(o->eFrame, 0, (size_t)o->eFrameLen * sizeof(double));
    memset(o->outEFrame, 0, (size_t)o->F * sizeof(double));
    memset(o->yFrame, 0, (size_t)o->F * sizeof(double));
    pitch_yin(o);
    pitch_marks(o);
    place_st_marks(o);
    if (o->anMarks.n != 0) {
        vpo_biased_autocorr(o->voice, o->inSize, o->currCounter, o->pStart, o->order, o->F, o->pAn, o->r);
        vpo_levinson_durbin(o->r, o->a, o->aPrev, o->order, VPO_ORDER_MAX + 1);
        pitch_filter_fir(o, -o->toKeep, o->toKeep + o->F, 0);
        o->stMarkIdx = 0;
        pitch_psola(o);
        pitch_filter_iir(o);
    }
    pitch_fill_output(o);
    trace_frame(o, 0);
}

static void pitch_chunk_cont(vpo *o)                                        /* :253-271 */
{
    if (o->anMarks.n != 0) {
        pitch_filter_fir(o, o->F - o->C, o->C, o->toKeep + o->F + (o->nChunk - 1) * o->C);
        pitch_psola(o);
        pitch_filter_iir(o);
        pitch_fill_output(o);
    }
}

static void pitch_process(vpo *o)                                           /* :166-196 */
{
    while (o->pStart < o->N) {
        if (o->nChunk % o->chunksPerFrame == o->chunksPerFrame - 1) {
            pitch_chunk_cont(o);
            o->nChunk = 0;
            pitch_chunk_start(o);
            o->nChunk += 1;
            o->nChunk %= o->chunksPerFrame;
        } else if (o->nChunk == 0) {
            pitch_chunk_start(o);
            o->nChunk += 1;
        } else {
            pitch_chunk_cont(o);
            o->nChunk += 1;
        }
        o->pStart += o->C;
    }
    o->pStart -= o->N;
}

/* ------------------------------------------------------------------------------------------ */
/* L3: PluginProcessor.cpp                                                                       */

vpo *vpo_create(void)
{
    vpo *o = calloc(1, sizeof *o);
    if (!o) return NULL;
    o->gainPitch = 0.0f; o->gainVoice = -60.0f; o->gainSynth = -60.0f; o->gainVoc = 0.0f;   /* :41-51 */
    o->lpcVoice = 40; o->lpcPitch = 15; o->lpcSynth = 5; o->keyPitch = 12;                   /* :53-64 */
    o->pitchBool = 1; o->vocBool = 1;                                                         /* :68-69 */
    o->ftz = 1;
    return o;
}

void vpo_destroy(vpo *o)
{
    if (!o) return;
    free(o->voice); free(o->synth[0]); free(o->synth[1]); free(o->out[0]); free(o->out[1]);
    free(o->vAn); free(o->vSt); free(o->eV); free(o->eS); free(o->vOut);
    free(o->yinTemp); free(o->eFrame); free(o->pAn); free(o->pSt);
    free(o->psolaWindow); free(o->periodSamples); free(o->xInterp); free(o->outEFrame); free(o->yFrame);
    free(o->trace); free(o->lin);
    free(o);
}

static float *param_slot(vpo *o, const char *id, float *lo, float *hi)
{
    struct { const char *id; float *p; float lo, hi; } t[] = {
        {"gainPitch", &o->gainPitch, -60.f, 6.f}, {"gainVoice", &o->gainVoice, -60.f, 6.f},
        {"gainSynth", &o->gainSynth, -60.f, 6.f}, {"gainVoc", &o->gainVoc, -60.f, 6.f},
        {"lpcVoice", &o->lpcVoice, 2.f, 100.f},   {"lpcPitch", &o->lpcPitch, 2.f, 100.f},
        {"lpcSynth", &o->lpcSynth, 2.f, 30.f},    {"keyPitch", &o->keyPitch, 0.f, 12.f},
        {"pitchBool", &o->pitchBool, 0.f, 1.f},   {"vocBool", &o->vocBool, 0.f, 1.f},
    };
    for (unsigned i = 0; i < sizeof t / sizeof t[0]; i++)
        if (strcmp(t[i].id, id) == 0) { *lo = t[i].lo; *hi = t[i].hi; return t[i].p; }
    return NULL;
}

/* Extension used by BASELINE configs[1] ("+-12-semitone pitch shift"): placeStMarks takes beta = 2^(semitones/12)
 * instead of closestFreq/pitch (:596-598).  The plugin has no such parameter: parity for it is GPU <-> this file only. */
int vpo_set_pitch_shift(vpo *o, int on, double semitones)
{
    if (!(semitones >= -12.0 && semitones <= 12.0)) return -1;
    o->shiftOn = on ? 1 : 0;
    o->shiftBeta = pow(2.0, semitones / 12.0);
    return 0;
}

int vpo_set_param(vpo *o, const char *id, float value)
{
    float lo, hi;
    float *p = param_slot(o, id, &lo, &hi);
    if (!p || !(value >= lo && value <= hi)) return -1;
    *p = value;
    return 0;
}

float vpo_get_param(const vpo *o, const char *id)
{
    float lo, hi;
    float *p = param_slot((vpo *)o, id, &lo, &hi);
    return p ? *p : NAN;
}

int vpo_prepare_explicit(vpo *o, double fs, int N, int F, int H, int W, int h)
{
    const double silenceDb = -60.0;                                          /* :148 */
    if (N <= 0 || F <= 0 || H <= 0 || W <= 0 || h <= 0) return -4;
    o->fs = fs;
    int rc = pitch_prepare(o, fs, 100, 800, F, H, silenceDb);               /* :172 */
    if (rc) return rc;
    rc = vocoder_prepare(o, W, h, silenceDb);                                /* :173 */
    if (rc) return rc;
    int latency = F > W ? F : W;                                             /* :175 (both getLatency return the frame length) */
    if (o->tauMax > F) return -5;                                            /* YIN reads back to startSample - tauMax: needs toKeep >= tauMax */
    rc = mybuffer_prepare(o, N, F, latency, fs);                             /* :176-179 samplesToKeep = frameLenPitch */
    if (rc) return rc;
    pitch_prepare2(o);                                                       /* :181 */
    free(o->trace);
    o->traceCap = N / (H > 0 ? H : 1) + 4;
    o->trace = calloc((size_t)o->traceCap, sizeof *o->trace);
    o->traceN = 0;
    memset(o->ub, 0, sizeof o->ub);
    o->prepared = 1;
    return 0;
}

int vpo_prepare_to_play(vpo *o, double sampleRate, int samplesPerBlock)     /* :144-184 */
{
    double ratioSR = sampleRate / 44100.0;
    int hopVoc = (int)floor(128.0 * ratioSR);
    int wlenVoc = 4 * hopVoc;
    int corres_256 = (int)floor(256.0 * ratioSR);
    int hopPitch = 3 * corres_256;
    int frameLenPitch = 4 * corres_256;
    return vpo_prepare_explicit(o, sampleRate, samplesPerBlock, frameLenPitch, hopPitch, wlenVoc, hopVoc);
}

static int process_block_io(vpo *o, const float *in0, const float *in1, const float *in2, float *ch0, float *ch1, float *ch2);

int vpo_process_block(vpo *o, float *ch0, float *ch1, float *ch2)          /* :203-234 */
{
    return process_block_io(o, ch0, ch1, ch2, ch0, ch1, ch2);
}

/* processBlock on a buffer whose side-chain bus is absent: MyBuffer::fillInputBuffers gets null side-chain pointers
 * and fills the synth ring with zeros (MyBuffer.cpp:93-102). */
int vpo_process_block_mono(vpo *o, const float *voice, float *outL, float *outR)
{
    return process_block_io(o, voice, NULL, NULL, outL, outR, NULL);
}

static int process_block_io(vpo *o, const float *in0, const float *in1, const float *in2, float *ch0, float *ch1, float *ch2)
{
    if (!o->prepared) return -1;
#ifdef VPO_HAVE_MXCSR
    unsigned int saved = _mm_getcsr();
    if (o->ftz) _mm_setcsr(saved | 0x8040);                                  /* ScopedNoDenormals: FTZ | DAZ */
#endif
    o->traceN = 0;
    fill_input_buffers(o, in0, in1, in2);
    if (o->vocBool) vocoder_process(o);
    if (o->pitchBool) pitch_process(o); else pitch_silence(o);
    if (o->gainVoice > -59.0) add_dry_voice(o, vpo_db_to_gain_f(o->gainVoice));
    if (o->gainSynth > -59.0) add_synth(o, vpo_db_to_gain_f(o->gainSynth));
    fill_output_buffer(o, ch0, ch1, ch2);
#ifdef VPO_HAVE_MXCSR
    _mm_setcsr(saved);
#endif
    return 0;
}

int vpo_get_latency(const vpo *o) { return o->latency; }

/* KAT hook: analysis pitch marks of one frame x[0..F) that follows an unvoiced frame
 * (PitchProcess.cpp:455-567 with prevPitch = 0, empty previous marks), for a given period. */
int vpo_kat_pitch_marks(const double *x, int F, int H, double fs, int period, int *marksOut)
{
    vpo *o = vpo_create();
    if (!o) return -1;
    int rc = vpo_prepare_explicit(o, fs, F, F, H, 512, 128);
    if (rc) { vpo_destroy(o); return rc; }
    for (int i = 0; i < F; i++) o->voice[(o->currCounter + i) % o->inSize] = x[i];
    o->pStart = 0;
    o->period = period; o->pitch = fs / period; o->prevPitch = 0; o->prevPeriod = 0;
    pitch_marks(o);
    int n = o->anMarks.n;
    memcpy(marksOut, o->anMarks.v, (size_t)n * sizeof(int));
    vpo_destroy(o);
    return n;
}

/* A run of consecutive pitch frames through pitchMarks (:455-567) with the tracker state rolled the way yin()
 * rolls it (:413-420): periods[f] > 0 = voiced frame with that period, 0 = unvoiced.  x holds F + (nFrames-1)*H
 * samples; frame f is x[f*H .. f*H+F).  marksOut is [nFrames][VPO_MARK_CAP], countsOut [nFrames]. */
int vpo_kat_marks_seq(const double *x, int nFrames, const int *periods, int F, int H, double fs, int *marksOut, int *countsOut,
                      int *stMarksOut, int *stCountsOut, int *periodNewOut, double *betaOut);
int vpo_kat_pitch_marks_seq(const double *x, int nFrames, const int *periods, int F, int H, double fs, int *marksOut, int *countsOut)
{
    return vpo_kat_marks_seq(x, nFrames, periods, F, H, fs, marksOut, countsOut, NULL, NULL, NULL, NULL);
}

/* ... and, when stMarksOut is given, placeStMarks (:573-658) after every pitchMarks, key = the default (chromatic). */
int vpo_kat_marks_seq(const double *x, int nFrames, const int *periods, int F, int H, double fs, int *marksOut, int *countsOut,
                      int *stMarksOut, int *stCountsOut, int *periodNewOut, double *betaOut)
{
    vpo *o = vpo_create();
    if (!o) return -1;
    int rc = vpo_prepare_explicit(o, fs, F, F, H, 512, 128);
    if (rc) { vpo_destroy(o); return rc; }
    o->pStart = 0;
    o->period = 0; o->pitch = 0; o->prevPitch = 0; o->prevPeriod = 0;
    for (int f = 0; f < nFrames; f++) {
        for (int i = 0; i < F; i++) o->voice[(o->currCounter + i) % o->inSize] = x[(size_t)f * H + i];
        o->prevPeriod = o->period; o->prevPitch = o->pitch;                  /* :413-414 */
        if (o->pitch > 1) { o->prevVoicedPeriod = o->period; o->prevVoicedPitch = o->pitch; }   /* :416-420 */
        o->period = periods[f];
        o->pitch = periods[f] > 0 ? fs / periods[f] : 0.0;
        pitch_marks(o);
        int n = o->anMarks.n;
        countsOut[f] = n;
        memcpy(marksOut + (size_t)f * VPO_MARK_CAP, o->anMarks.v, (size_t)(n < VPO_MARK_CAP ? n : VPO_MARK_CAP) * sizeof(int));
        if (stMarksOut) {
            place_st_marks(o);
            n = o->stMarks.n;
            stCountsOut[f] = n;
            memcpy(stMarksOut + (size_t)f * VPO_MARK_CAP, o->stMarks.v, (size_t)(n < VPO_MARK_CAP ? n : VPO_MARK_CAP) * sizeof(int));
            periodNewOut[f] = o->periodNew;
            betaOut[f] = o->beta;
        }
    }
    vpo_destroy(o);
    return 0;
}

/* KAT hook: PitchProcess::psola (:665-741) + interp (:842-870) + getClosestAnMarkIdx (:788-831) for ONE frame, every
 * synthesis mark in one go (nChunk = chunksPerFrame - 1, so that the scheduling test :685 passes all of them; with
 * N = F the completeness tests :800-818 hold for every mark of the frame).  e = eFrame[0 .. toKeep + F) (frame position
 * p <-> e[toKeep + p], :698), voiced frame of period T, shift factor beta.  outE [F] receives outEFrame. */
int vpo_kat_psola(const double *e, int F, int H, double fs, int T, double beta, const int *an, int nAn, const int *st, int nSt,
                  double *outE)
{
    if (nAn > VPO_MARK_CAP || nSt > VPO_MARK_CAP || T < 1) return -1;
    vpo *o = vpo_create();
    if (!o) return -1;
    int rc = vpo_prepare_explicit(o, fs, F, F, H, 512, 128);
    if (rc) { vpo_destroy(o); return rc; }
    if (T > o->tauMax) { vpo_destroy(o); return -1; }
    memset(o->eFrame, 0, (size_t)o->eFrameLen * sizeof(double));
    memcpy(o->eFrame, e, (size_t)(o->toKeep + F) * sizeof(double));
    memset(o->outEFrame, 0, (size_t)F * sizeof(double));
    o->pStart = 0; o->nChunk = o->chunksPerFrame - 1;
    o->period = T; o->pitch = fs / T; o->beta = beta;
    o->nAnMarksOv = 0;
    for (int i = 0; i < nAn; i++) o->anMarks.v[i] = an[i];
    o->anMarks.n = nAn;
    for (int i = 0; i < nSt; i++) o->stMarks.v[i] = st[i];
    o->stMarks.n = nSt;
    o->stMarkIdx = 0;
    pitch_psola(o);
    memcpy(outE, o->outEFrame, (size_t)F * sizeof(double));
    rc = (int)(o->ub[0] + o->ub[1]);                                          /* > 0: a Q2/Q3 path was taken */
    vpo_destroy(o);
    return rc;
}

/* KAT hook: the pitch path's two filters with GIVEN coefficients a[0..order].  x [toKeep + F] = voice samples
 * idx -toKeep .. F-1; eOut [toKeep + F] <- filterFIR(-toKeep, toKeep + F, 0) (:280-302, as processChunkStart :235
 * calls it); yOut [F] <- filterIIR (:307-322) of outE [F], chunk after chunk with the state carried through yFrame. */
int vpo_kat_pitch_filters(const double *x, int F, int H, double fs, const double *a, int order, const double *outE,
                          double *eOut, double *yOut)
{
    vpo *o = vpo_create();
    if (!o) return -1;
    if (vpo_set_param(o, "lpcPitch", (float)order)) { vpo_destroy(o); return -1; }
    int rc = vpo_prepare_explicit(o, fs, F, F, H, 512, 128);
    if (rc) { vpo_destroy(o); return rc; }
    for (int i = 0; i < o->toKeep + F; i++)
        o->voice[(o->currCounter + (i - o->toKeep) + o->inSize) % o->inSize] = x[i];
    o->pStart = 0;
    memset(o->a, 0, sizeof o->a);
    memcpy(o->a, a, (size_t)(order + 1) * sizeof(double));
    memset(o->eFrame, 0, (size_t)o->eFrameLen * sizeof(double));
    pitch_filter_fir(o, -o->toKeep, o->toKeep + F, 0);
    memcpy(eOut, o->eFrame, (size_t)(o->toKeep + F) * sizeof(double));
    memcpy(o->outEFrame, outE, (size_t)F * sizeof(double));
    memset(o->yFrame, 0, (size_t)F * sizeof(double));
    for (o->nChunk = 0; o->nChunk < o->chunksPerFrame; o->nChunk++) pitch_filter_iir(o);
    memcpy(yOut, o->yFrame, (size_t)F * sizeof(double));
    vpo_destroy(o);
    return 0;
}

/* KAT hook: one vocoder window with GIVEN coefficient vectors, from a fresh state (empty energy histories): the two
 * residuals and their energies (filterFIR :235-251), the gain (:264-275) and the all-pole output before the synthesis
 * window (:277-286).  voice/synth [W] are the raw samples of the window (the analysis window is applied inside). */
int vpo_kat_voc_window(const double *voice, const double *synth, int W, int hop, const double *aV, int orderV, const double *aS,
                       int orderS, double *eV, double *eS, double *EE, double *g, double *out)
{
    vpo *o = vpo_create();
    if (!o) return -1;
    int rc = vpo_prepare_explicit(o, 44100.0, W, 1024, 768, W, hop);
    if (rc) { vpo_destroy(o); return rc; }
    for (int i = 0; i < W; i++) {
        o->voice[(o->currCounter + i) % o->inSize] = voice[i];
        o->synth[0][(o->currCounter + i) % o->inSize] = synth[i];
    }
    o->vStart = 0;
    voc_filter_fir(o, 0, o->eV, aV, orderV, &o->EeV);
    voc_filter_fir(o, 1, o->eS, aS, orderS, &o->EeS);
    voc_filter_iir(o, aV, orderV);
    memcpy(eV, o->eV, (size_t)W * sizeof(double));
    memcpy(eS, o->eS, (size_t)W * sizeof(double));
    memcpy(out, o->vOut, (size_t)W * sizeof(double));
    EE[0] = o->EeV; EE[1] = o->EeS; *g = o->g;
    vpo_destroy(o);
    return 0;
}

int vpo_get_geometry(const vpo *o, int out[12])
{
    out[0] = o->N; out[1] = o->F; out[2] = o->H; out[3] = o->C; out[4] = o->W; out[5] = o->h;
    out[6] = o->toKeep; out[7] = o->latency; out[8] = o->inSize; out[9] = o->outSize;
    out[10] = o->tauMax; out[11] = o->chunksPerFrame;
    return 0;
}

int vpo_trace_count(const vpo *o) { return o->traceN; }
int vpo_trace_get(const vpo *o, int i, vpo_pitch_frame *out)
{
    if (i < 0 || i >= o->traceN) return -1;
    *out = o->trace[i];
    return 0;
}
void vpo_ub_counters(const vpo *o, long out[5]) { memcpy(out, o->ub, sizeof o->ub); }
void vpo_set_ftz(vpo *o, int on) { o->ftz = on; }
