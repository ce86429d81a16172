// vp_capi.hip -- host side of libvp_amd.so: the C ABI declared in include/vp_amd.h.
//
// Mirrors VocoderAudioProcessor (PluginProcessor.cpp) for a batch of streams: parameter storage
// (:37-73), prepareToPlay geometry + allocation (:144-184) and the processBlock orchestration
// (:203-234: fill -> vocoder -> pitch (or silence) -> dry voice -> dry synth -> out), with the
// per-block counters of MyBuffer / VocoderProcess / PitchProcess kept on the host because they
// advance identically for every stream.  There is no CPU fallback: without a HIP device every
// entry point that needs one fails with VP_ERR_NO_DEVICE.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <type_traits>
#include <vector>

#include "../../include/vp_amd.h"
#include "vp_kernels.h"
#include "vp_voc2.h"
#include "vp_stft.h"

struct vp_handle {
    int device = 0;
    bool prepared = false;
    vp_params params;
    VpGeom g;
    VpDev d;
    // MyBuffer counters (MyBuffer.h:72-79): every instance advances them by N per block whatever its switches
    int inCounter = 0, outCounter = 0, currCounter = 0;
    // VocoderProcess::startSample, PitchProcess::startSample/nChunk only advance while the instance's vocBool / pitchBool
    // is on (PluginProcessor.cpp:214-221), so streams whose switches differ(ed) carry different values: the batch is
    // kept as COHORTS of streams that share switches and counters.  One cohort holding every stream (dMap == nullptr)
    // is the normal case and costs nothing; otherwise each cohort is launched on its own with a stream-index map.
    // ordHi: the cohort's streams ask for a voice LPC order above what the lane-per-window pipeline is instantiated for (they
    // take the workgroup kernel; the others keep the pipeline: one order-64 stream no longer demotes a thousand others)
    struct Cohort { int pitchOn, vocOn, vStart, pStart, nChunk, n; int *dMap; std::vector<int> ids; int ordHi = 0, oVmax = 0, oSmax = 0; };
    std::vector<Cohort> cohorts;
    bool cohortsDirty = false;                  // a set call changed some stream's pitchBool / vocBool
    int *dMapAll = nullptr;                     // [S] device storage of the cohorts' maps, back to back
    std::vector<int> mapHost;                   // staging for its upload
    bool poisoned = false;                      // a HIP call failed inside a process call: host counters and device state may disagree
    int poisonCode = VP_ERR_HIP;                // ... or (VP_ERR_TIMEOUT) a kernel's bounded inter-wavefront wait ran out: its output is void
    // the fault word: pinned host memory the kernels can write (VpDev::fault).  Looked at -- a plain host read, no synchronisation --
    // at the top of every process call and behind every synchronisation this library makes itself (check_fault).
    unsigned int *faultHost = nullptr, *faultDev = nullptr;
    int spinLimit = 1 << 22;                    // VpDev::spinLimit (vp_debug_set_spin_limit)
    // the batched lane-per-window vocoder pipeline (vp_voc2.hip): scratch, the orders it must cover, and who picks it
    VpV2 v2;
    int vocPath = VP_VOC_AUTO, oVmax = 0, oSmax = 0, nWinMax = 0;
    VpV2 v2mb;                                  // the same scratch sized for several blocks per launch (allocated on first use)
    int v2mbWin = 0;                            // windows per stream it holds
    std::vector<void *> mbAllocs;
    // combined multi-block plan (process_both_blocks): ring snapshots, per-block gates, the pitch corrector's linear accumulator
    float *snapV = nullptr, *snapS = nullptr; int *gateB = nullptr; double *pLin = nullptr;
    std::vector<void *> bothAllocs;
    // vp_set_overlap(h, 1), VP_IIR_FAST, both processes on, batched vocoder: the pitch kernel runs beside the vocoder pipeline on
    // auxStream and adds into its own accumulator (acc2), which emit merges.  acc2Live: blocks for which acc2 may still hold
    // something.  VP_OVERLAP_AUTO (default): where the pitch build leaves registers beside it (process_device).
    hipStream_t auxStream = nullptr;
    hipEvent_t evFork = nullptr, evJoin = nullptr;
    double *acc2 = nullptr;
    int overlap = VP_OVERLAP_AUTO, acc2Live = 0;
    int waveSpec = 1;                           // vp_set_wave_specialised: vp_k_pitch_ws* where they apply (pitch_ws_ok)
    int timeParallel = 0;                       // vp_set_time_parallel: stored and returned, without effect (the analysis front end it selected was removed in round 6)
    std::vector<void *> allocs;
    float *stageIn = nullptr, *stageOut = nullptr;
    int synthNonZero = 0;                       // samples of the synth rings not known to be zero (mono entry points)
    float *stageInB = nullptr, *stageOutB = nullptr; int stageBlocks = 0;   // vp_process_blocks: [B][S][3|2][N], sized by vp_reserve_blocks
    int reservedBlocks = 0;                     // vp_reserve_blocks: blocks per call the multi-block scratch is sized for
    long nAllocs = 0;                           // device allocations made for this handle since vp_create (vp_debug_alloc_count)
    hipStream_t ownStream = nullptr;
    int vocWaves = 8;
    size_t vocLds = 0, pitchLds = 0, ldsMax = 0;
    int prof = 0;                               // 0 off, k: every k-th launch is bracketed with events
    unsigned profTick = 0;
    bool profThis = false;                      // the call in hand is a sampled one
    int iirMode = 0, yinMode = 0;
    // per-stream parameter overrides (vp_set_stream_params): host copy, device copy, upload pending
    std::vector<vp_params> sparams;             // [S] what each stream's treeState holds
    std::vector<VpStreamParams> spHost;         // [S] the same in kernel form (float gains applied), staging for the upload
    std::vector<int> shiftOn;                   // [S] vp_set_pitch_shift: fixed interval instead of the key's nearest note
    std::vector<double> shiftSemi, shiftBeta;   // [S]
    bool perStream = false, spDirty = false;    // any stream differs from `params` / device copy out of date
    struct EvPair { hipEvent_t a, b; int slot; };
    std::vector<EvPair> pending;
    std::vector<hipEvent_t> evPool;
    double profMs[VP_NUM_KERNEL_SLOTS] = {0, 0, 0, 0};
    long profN[VP_NUM_KERNEL_SLOTS] = {0, 0, 0, 0};
    std::string lastError;
};

static int fail_hip(vp_handle *h, hipError_t e, const char *what)
{
    char buf[256];
    snprintf(buf, sizeof buf, "%s: %s", what, hipGetErrorString(e));
    if (h) h->lastError = buf;
    return VP_ERR_HIP;
}
#define HIPCHK(h, call) do { hipError_t _e = (call); if (_e != hipSuccess) return fail_hip((h), _e, #call); } while (0)

// A poisoned handle answers every process call with the code that poisoned it until it is prepared again.
static int poisoned_rc(vp_handle *h)
{
    h->lastError = h->poisonCode == VP_ERR_TIMEOUT
        ? "an earlier launch timed out on an inter-wavefront wait (its output and the device state are void): prepare again"
        : "an earlier HIP failure left the handle out of step with its device state: prepare again";
    return h->poisonCode;
}
// The kernels' fault word (vp_timeout, vp_kernels.hip).  A plain read of pinned host memory: what it shows is whatever launches have
// COMPLETED so far, so a timeout is reported by the call that synchronises behind the launch (the host-pointer entry points,
// vp_synchronize, the vp_read_* calls) or, for the device-pointer entry points, by the next call on the handle.
static int check_fault(vp_handle *h)
{
    if (!h->faultHost) return VP_OK;
    const unsigned f = *(volatile unsigned int *)h->faultHost;
    if (f == 0) return VP_OK;
    h->poisoned = true; h->poisonCode = VP_ERR_TIMEOUT;
    char buf[200];
    snprintf(buf, sizeof buf, "a kernel's bounded wait ran out (counter dbg[%u], see vp_debug_read_stamps): the block's output is invalid; prepare again", f & 0xff);
    h->lastError = buf;
    return VP_ERR_TIMEOUT;
}

// every device allocation of a handle goes through here: prepare and vp_reserve_blocks allocate, vp_process_*() never do
// (include/vp_amd.h; tests/test_gpu_round4.py watches the count across every process entry point)
static hipError_t vp_malloc(vp_handle *h, void **q, size_t bytes)
{
    h->nAllocs += 1;
    return hipMalloc(q, bytes);
}

extern "C" int vp_abi_version(void) { return VP_ABI_VERSION; }

extern "C" const char *vp_error_string(int code)
{
    switch (code) {
    case VP_OK: return "ok";
    case VP_ERR_INVALID_ARG: return "invalid argument or parameter out of range";
    case VP_ERR_NOT_PREPARED: return "process called before prepare";
    case VP_ERR_INVALID_OVERLAP: return "Invalid overlap";                    // VocoderProcess.cpp:112
    case VP_ERR_GEOMETRY: return "invalid frame geometry";
    case VP_ERR_ORDER: return "LPC order is larger than OrderMax";            // VocoderProcess.cpp:146,162
    case VP_ERR_NO_DEVICE: return "no HIP device (this library has no CPU fallback)";
    case VP_ERR_HIP: return "HIP runtime error";
    case VP_ERR_OOM: return "out of device memory";
    case VP_ERR_TIMEOUT: return "a kernel's bounded inter-wavefront wait ran out: the output is invalid, prepare again";
    default: return "unknown error";
    }
}

extern "C" const char *vp_last_error(const vp_handle *h) { return h ? h->lastError.c_str() : ""; }

// large batches: the register-light build lets two workgroups share a CU (needs <= 80 KB LDS each and no big
// register-resident exact-IIR instantiation)
static bool pitch_lite(const vp_handle *h, bool iirFast)
{
    // (htabGlobal: the register-light builds are compiled for the global Hann table only; prepare sets it for exactly these batches)
    return h->g.S > 256 && (iirFast || h->g.orderPitch <= 16) && h->pitchLds <= 80 * 1024 && h->g.htabGlobal;
}

// geometry and order for which the common-case builds (vp_k_pitch*_c) are valid: the geometric predicate proper -- all the register-light
// common-case builds need, they carry the eight-lag form of the certified YIN --
static bool pitch_common_geom(const vp_handle *h)
{
    // (the last term: room behind the YIN window's prefix sums in eFrame for the quarter sums of xcorr8_quarter, vp_pitch.inc xc8_ok)
    return (h->g.C & 63) == 0 && h->g.orderPitch <= 15 && h->g.tauMax <= 512 && h->g.eLen >= h->g.F + 4 * h->g.tauMax + 1;
}
// ... and, for the full-register ones (round 4): frames of one or two whole 512-sample segments -- 22.05 and 44.1 kHz with the plugin's
// geometry -- and room for the FFT cross-correlation's tables and exchange buffers: they carry no other form of the certified YIN
static bool pitch_common(const vp_handle *h)
{
    return pitch_common_geom(h) && (h->g.F == 512 || h->g.F == 1024) && h->pitchLds + 16 + vp_pitch_fft_lds_bytes(2 * (h->g.F >> 9)) <= h->ldsMax;
}

// THE selection of the pitch-kernel build for a launch (used by the launch site and by vp_pitch_kernel_name alike)
typedef void (*vp_dsp_kernel)(VpGeom, VpCall, VpDev, const float *, float *);
struct PitchPlan { vp_dsp_kernel fn; const char *name; size_t lds; };
#define VP_PLAN(K, LDS) PitchPlan{K, #K, LDS}
static PitchPlan pitch_plan(const vp_handle *h, bool fast, int nBlocks)
{
    const bool lite = pitch_lite(h, fast), com = lite ? pitch_common_geom(h) : pitch_common(h);
    const size_t lds = h->pitchLds;
    if (nBlocks > 1) {                                    // (`lite` multi-block launches exist for the FAST recursion only: process_blocks_device)
        if (lite) return com ? VP_PLAN(vp_k_pitch_lite_fast_multi_c, lds) : VP_PLAN(vp_k_pitch_lite_fast_multi, lds);
        return fast ? (com ? VP_PLAN(vp_k_pitch_fast_multi_c, lds) : VP_PLAN(vp_k_pitch_fast_multi, lds)) : VP_PLAN(vp_k_pitch_multi, lds);
    }
    if (lite) return fast ? (com ? VP_PLAN(vp_k_pitch_lite_fast_c, lds) : VP_PLAN(vp_k_pitch_lite_fast, lds)) : VP_PLAN(vp_k_pitch_lite, lds);
    if (com) return fast ? VP_PLAN(vp_k_pitch_fast_c, lds) : VP_PLAN(vp_k_pitch_c, lds);
    return fast ? VP_PLAN(vp_k_pitch_fast, lds) : VP_PLAN(vp_k_pitch, lds);
}
// Round 5: the wave-specialised kernel (vp_pitch_ws.inc) serves single-block launches of the common-case geometry with 1024-sample
// frames (the plugin's own at 44.1 kHz) whenever its carve -- the whole block's voice window, two frames' buffers, the background's
// scratch -- fits the CU's LDS; one workgroup per CU, so batches that the register-light builds would pack two to a CU keep those.
// VP_NO_WS=1 (diagnostic): the phase kernels everywhere, for same-box A/B runs.
static bool pitch_ws_ok(const vp_handle *h, bool fast, int nBlocks, int nSteps)
{
    static const bool off = getenv("VP_NO_WS") != nullptr;
    const VpGeom &g = h->g;
    // (its own geometric conditions: chunks of whole wavefronts up to 512 samples, lpcPitch <= 24 -- the _o24 builds from 16 on, round 6 --,
    // every lag on one wavefront, two 512-sample segments per frame for the FFT cross-correlation)
    if (off || !h->waveSpec || nBlocks != 1 || nSteps <= 0 || pitch_lite(h, fast) || g.F != 1024 || (g.C & 63) != 0 || g.C > 512 || g.cpf < 2 ||
        g.orderPitch > WS_ORDER_MAX || g.tauMax > 512) return false;
    if ((size_t)(g.toKeep + g.F + (nSteps - 1) * g.C) >= (size_t)g.inSize) return false;
    if (nSteps + (nSteps + g.cpf - 2) / (g.cpf - 1) + 1 > WS_MAXI || (nSteps + g.cpf - 2) / (g.cpf - 1) + 1 > WS_MAXS) return false;   // (ws_build_sched's limits, whatever nChunk)
    return vp_pitch_ws_lds_bytes(g, nSteps) + 256 <= h->ldsMax;                 // (+ the kernels' static reduction slots: 384 bytes against the 256 ldsMax leaves)
}

// The certified cross-correlation YIN of the full-register common-case builds (vp_k_pitch_c, vp_k_pitch_fast_c, vp_k_pitch_fast_multi_c)
// evaluates its cross-correlations by FFT (vp_pitch.inc xcorr_fft_wave; they carry no other form): one wavefront and one exchange
// buffer per forward transform, two per 512-sample segment of the frame.  0: the launch does not use it.
static int pitch_xfft_waves(const vp_handle *h, bool fast, int nBlocks)
{
    if (!pitch_common(h) || pitch_lite(h, fast) || (nBlocks > 1 && !fast)) return 0;
    int fw = 2 * (h->g.F >> 9);
    static const int envWaves = [] { const char *e = getenv("VP_XFFT_WAVES"); return e ? atoi(e) : 0; }();   // (read once: this runs on every process call)
    if (envWaves == fw / 2) fw = envWaves;                                    // diagnostic: one SEGMENT per wavefront (measured 2 % slower)
    return fw;
}
static size_t pitch_xfft_lds(const vp_handle *h) { return vp_pitch_fft_lds_bytes(2 * (h->g.F >> 9)); }                  // (a buffer per TRANSFORM)
#undef VP_PLAN

extern "C" const char *vp_pitch_kernel_name(const vp_handle *h)
{
    if (!h || !h->prepared) return "";
    const bool fast = h->iirMode == VP_IIR_FAST;
    if (pitch_ws_ok(h, fast, 1, (h->g.N + h->g.C - 1) / h->g.C))
        return h->g.orderPitch > 15 ? (fast ? "vp_k_pitch_ws_o24" : "vp_k_pitch_ws_x_o24") : (fast ? "vp_k_pitch_ws" : "vp_k_pitch_ws_x");
    return pitch_plan(h, fast, 1).name;
}

// vp_k_vocoder_lite (FAST IIR only, <= 128 VGPRs): above 256 streams, with at most four window slots so that two
// workgroups share a CU (<= 80 KB of LDS each).  Returns the slots to use, 0 = the regular build.
static int voc_lite_slots(const vp_handle *h, bool iirFast, int nw)
{
    if (!(h->g.S > 256 && iirFast) || getenv("VP_VOC_NO_LITE")) return 0;
    int nl = std::max(1, std::min(nw, 4));
    while (nl > 1 && vp_voc_lds_bytes(h->g.W, nl) > (size_t)(80 * 1024 - 512)) nl--;
    return vp_voc_lds_bytes(h->g.W, nl) <= (size_t)(80 * 1024 - 512) ? nl : 0;
}

// The batched pipeline pays as soon as the batch no longer fits one workgroup per CU (measured, vocoder alone, default
// geometry: 256 streams 110 us workgroup / 133 us batched; 320 streams 198 / 137; 512: 203 / 142; 1024: 389 / 184) and the
// launch has a thousand windows or more (a lane each); it covers LPC orders up to V2_ORDER_MAX and blocks of up to 64
// windows per stream.  VP_VOC_BATCHED forces it wherever it is able to run.
#define VP_V2_MIN_STREAMS 257
#define VP_V2_MIN_WINDOWS 1024
// VP_VOC_AUTO is decided ONCE per prepare from the handle's batch (streams x windows per block), not per block and cohort: a
// cohort split (one stream's pitchBool toggled) or a block with one window fewer (N not a multiple of the hop) must not flip
// the other streams between the two implementations, whose FAST-mode roundings differ.
static bool voc_auto_batched(const vp_handle *h)
{
    return h->g.S >= VP_V2_MIN_STREAMS && (size_t)h->g.S * h->nWinMax >= VP_V2_MIN_WINDOWS;
}
static bool voc_pipeline_wanted(const vp_handle *h)
{
    return h->v2.xT && h->vocPath != VP_VOC_WORKGROUP && (h->vocPath == VP_VOC_BATCHED || voc_auto_batched(h));
}
// per block only what the pipeline is ABLE to run: orders it is instantiated for, 1..64 windows per stream
static bool voc_batched_for(const vp_handle *h, int nWin, int oV, int oS)
{
    if (!voc_pipeline_wanted(h) || nWin < 1 || nWin > 64) return false;
    return oV <= V2_ORDER_MAX && oS <= VP_ORDER_MAX_SYNTH && oV >= 2 && oS >= 2;
}

// the workgroup vocoder's two full-register builds: vp_k_vocoder (orders up to 32, and above 48: no scratch) and vp_k_vocoder_o48
static bool voc_o48(int oVmax) { return oVmax > 32 && oVmax <= 48; }

extern "C" int vp_set_vocoder_path(vp_handle *h, int path)
{
    if (!h || path < VP_VOC_AUTO || path > VP_VOC_BATCHED) return VP_ERR_INVALID_ARG;
    if (path != h->vocPath) h->cohortsDirty = true;        // the cohorts are keyed by the order class the pipeline can take (rebuild_cohorts)
    h->vocPath = path;
    return VP_OK;
}
extern "C" int vp_get_vocoder_path(const vp_handle *h) { return h ? h->vocPath : VP_ERR_INVALID_ARG; }
extern "C" int vp_set_overlap(vp_handle *h, int on)
{
    if (!h || on < 0 || on > VP_OVERLAP_AUTO) return VP_ERR_INVALID_ARG;
    h->overlap = on;
    return VP_OK;
}
extern "C" int vp_get_overlap(const vp_handle *h) { return h ? h->overlap : VP_ERR_INVALID_ARG; }
extern "C" int vp_set_time_parallel(vp_handle *h, int on)
{
    if (!h) return VP_ERR_INVALID_ARG;
    h->timeParallel = on ? 1 : 0;
    return VP_OK;
}
extern "C" int vp_get_time_parallel(const vp_handle *h) { return h ? h->timeParallel : VP_ERR_INVALID_ARG; }
extern "C" int vp_set_wave_specialised(vp_handle *h, int on)
{
    if (!h) return VP_ERR_INVALID_ARG;
    h->waveSpec = on ? 1 : 0;
    return VP_OK;
}
extern "C" int vp_get_wave_specialised(const vp_handle *h) { return h ? h->waveSpec : VP_ERR_INVALID_ARG; }

extern "C" const char *vp_vocoder_kernel_name(const vp_handle *h)
{
    if (!h || !h->prepared) return "";
    {
        // (orders as the next block will see them; streams above the pipeline's orders form a cohort of their own on the workgroup
        // kernel, the others keep the pipeline)
        int oV = 0, oS = 0, nLow = 0;
        for (const auto &q : h->sparams) if (q.lpcVoice <= V2_ORDER_MAX) { oV = std::max(oV, q.lpcVoice); oS = std::max(oS, q.lpcSynth); nLow++; }
        if (nLow > 0 && voc_batched_for(h, h->nWinMax, oV, oS)) return "vp_k_v2_pipeline";
    }
    if (voc_lite_slots(h, h->iirMode == VP_IIR_FAST, h->vocWaves)) return "vp_k_vocoder_lite";
    int oV = 0;
    for (const auto &q : h->sparams) oV = std::max(oV, q.lpcVoice);
    return (oV > 32 && oV <= 48) ? "vp_k_vocoder_o48" : "vp_k_vocoder";
}

extern "C" const char *vp_kernel_slot_name(int slot)
{
    static const char *n[VP_NUM_KERNEL_SLOTS] = {"vp_k_ingest_gate", "vp_k_vocoder", "vp_k_pitch", "vp_k_emit"};
    return (slot >= 0 && slot < VP_NUM_KERNEL_SLOTS) ? n[slot] : "";
}

extern "C" void vp_default_params(vp_params *p)
{
    // createParameterLayout defaults, PluginProcessor.cpp:41-69
    p->gainPitch = 0.0f; p->gainVoice = -60.0f; p->gainSynth = -60.0f; p->gainVoc = 0.0f;
    p->lpcVoice = 40; p->lpcPitch = 15; p->lpcSynth = 5; p->keyPitch = 12;
    p->pitchBool = 1; p->vocBool = 1;
}

static bool params_valid(const vp_params *p)
{
    auto gain_ok = [](float v) { return v >= -60.0f && v <= 6.0f; };
    return gain_ok(p->gainPitch) && gain_ok(p->gainVoice) && gain_ok(p->gainSynth) && gain_ok(p->gainVoc) &&
           p->lpcVoice >= 2 && p->lpcVoice <= VP_ORDER_MAX && p->lpcPitch >= 2 && p->lpcPitch <= VP_ORDER_MAX &&
           p->lpcSynth >= 2 && p->lpcSynth <= VP_ORDER_MAX_SYNTH && p->keyPitch >= 0 && p->keyPitch <= 12 &&
           (p->pitchBool == 0 || p->pitchBool == 1) && (p->vocBool == 0 || p->vocBool == 1);
}

extern "C" int vp_create(int device, vp_handle **out)
{
    if (!out) return VP_ERR_INVALID_ARG;
    *out = nullptr;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) return VP_ERR_NO_DEVICE;
    if (device < 0 || device >= n) return VP_ERR_INVALID_ARG;
    vp_handle *h = new vp_handle();
    h->device = device;
    vp_default_params(&h->params);
    memset(&h->g, 0, sizeof h->g);
    memset(&h->d, 0, sizeof h->d);
    if (hipSetDevice(device) != hipSuccess || hipStreamCreateWithFlags(&h->ownStream, hipStreamNonBlocking) != hipSuccess ||
        hipStreamCreateWithFlags(&h->auxStream, hipStreamNonBlocking) != hipSuccess ||
        hipEventCreateWithFlags(&h->evFork, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&h->evJoin, hipEventDisableTiming) != hipSuccess) {
        delete h;
        return VP_ERR_NO_DEVICE;
    }
    void *fh = nullptr, *fd = nullptr;
    if (hipHostMalloc(&fh, 64, hipHostMallocMapped | hipHostMallocCoherent) != hipSuccess || hipHostGetDevicePointer(&fd, fh, 0) != hipSuccess) {
        if (fh) (void)hipHostFree(fh);
        (void)hipGetLastError();
        (void)hipStreamDestroy(h->ownStream); (void)hipStreamDestroy(h->auxStream); (void)hipEventDestroy(h->evFork); (void)hipEventDestroy(h->evJoin);
        delete h;
        return VP_ERR_OOM;
    }
    memset(fh, 0, 64);
    h->faultHost = (unsigned int *)fh; h->faultDev = (unsigned int *)fd;
    *out = h;
    return VP_OK;
}

static void free_all(vp_handle *h)
{
    for (void *p : h->allocs) (void)hipFree(p);
    h->allocs.clear();
    h->stageIn = h->stageOut = nullptr; h->dMapAll = nullptr; h->cohorts.clear(); h->acc2 = nullptr;
    for (void *p : h->mbAllocs) (void)hipFree(p);
    h->mbAllocs.clear(); h->v2mbWin = 0; memset(&h->v2mb, 0, sizeof h->v2mb); memset(&h->v2, 0, sizeof h->v2);
    for (void *p : h->bothAllocs) (void)hipFree(p);
    h->bothAllocs.clear(); h->snapV = h->snapS = nullptr; h->gateB = nullptr; h->pLin = nullptr;
    if (h->stageInB) (void)hipFree(h->stageInB);
    if (h->stageOutB) (void)hipFree(h->stageOutB);
    h->stageInB = h->stageOutB = nullptr; h->stageBlocks = 0; h->reservedBlocks = 0;
    h->prepared = false;
}

extern "C" int vp_destroy(vp_handle *h)
{
    if (!h) return VP_ERR_INVALID_ARG;
    (void)hipSetDevice(h->device);
    (void)hipDeviceSynchronize();
    free_all(h);
    for (auto &p : h->pending) { h->evPool.push_back(p.a); h->evPool.push_back(p.b); }
    for (hipEvent_t e : h->evPool) (void)hipEventDestroy(e);
    if (h->ownStream) (void)hipStreamDestroy(h->ownStream);
    if (h->auxStream) (void)hipStreamDestroy(h->auxStream);
    if (h->evFork) (void)hipEventDestroy(h->evFork);
    if (h->evJoin) (void)hipEventDestroy(h->evJoin);
    if (h->faultHost) (void)hipHostFree(h->faultHost);
    delete h;
    return VP_OK;
}

extern "C" int vp_set_params(vp_handle *h, const vp_params *p)
{
    if (!h || !p || !params_valid(p)) return VP_ERR_INVALID_ARG;
    const bool sw = p->pitchBool != h->params.pitchBool || p->vocBool != h->params.vocBool || h->perStream;
    h->params = *p;
    h->perStream = false;                       // one set for every stream again
    for (auto &q : h->sparams) q = *p;
    h->spDirty = true;
    if (sw) h->cohortsDirty = true;             // (the processes' counters stay where each stream left them)
    return VP_OK;
}

static void fill_stream_params(VpStreamParams &o, const vp_params &P);

extern "C" int vp_set_stream_params(vp_handle *h, int stream, const vp_params *p)
{
    if (!h || !p || !params_valid(p)) return VP_ERR_INVALID_ARG;
    if (!h->prepared) return VP_ERR_NOT_PREPARED;
    if (stream < 0 || stream >= h->g.S) { h->lastError = "stream index out of range"; return VP_ERR_INVALID_ARG; }
    // the prepare-time order is per handle (it selects the pitch kernel's build)
    if (p->lpcPitch != h->params.lpcPitch) {
        h->lastError = "lpcPitch is per handle (vp_set_params, read at prepare); per-stream sets must repeat it";
        return VP_ERR_INVALID_ARG;
    }
    if (p->pitchBool != h->sparams[stream].pitchBool || p->vocBool != h->sparams[stream].vocBool) h->cohortsDirty = true;
    h->sparams[stream] = *p;
    h->perStream = true;
    h->spDirty = true;
    return VP_OK;
}

// Extension (BASELINE configs[1] "+-12-semitone pitch shift"; the plugin only corrects to the key's nearest note):
// placeStMarks takes beta = 2^(semitones/12) instead of closestFreq/pitch (PitchProcess.cpp:596-598).
extern "C" int vp_set_pitch_shift(vp_handle *h, int stream, int on, double semitones)
{
    if (!h) return VP_ERR_INVALID_ARG;
    if (!h->prepared) return VP_ERR_NOT_PREPARED;
    if (stream < -1 || stream >= h->g.S) { h->lastError = "stream index out of range"; return VP_ERR_INVALID_ARG; }
    if (!(semitones >= -12.0 && semitones <= 12.0)) { h->lastError = "pitch shift outside +-12 semitones"; return VP_ERR_INVALID_ARG; }
    const double beta = pow(2.0, semitones / 12.0);
    if (on) {
        // the synthesis marks of a frame must fit the mark arrays: shortest period floor(fs/fMax), new period round(period/beta)
        const int pNew = (int)round(h->g.tau0 / beta);
        if (pNew < 1 || h->g.F / pNew + 2 > VP_MARKS) {
            h->lastError = "pitch shift needs more synthesis marks per frame than the mark arrays hold";
            return VP_ERR_GEOMETRY;
        }
    }
    const int lo = stream < 0 ? 0 : stream, hi = stream < 0 ? h->g.S : stream + 1;
    for (int i = lo; i < hi; i++) { h->shiftOn[i] = on ? 1 : 0; h->shiftSemi[i] = semitones; h->shiftBeta[i] = beta; }
    h->spDirty = true;
    return VP_OK;
}

extern "C" int vp_get_pitch_shift(const vp_handle *h, int stream, int *on, double *semitones)
{
    if (!h || !on || !semitones) return VP_ERR_INVALID_ARG;
    if (!h->prepared) return VP_ERR_NOT_PREPARED;
    if (stream < 0 || stream >= h->g.S) return VP_ERR_INVALID_ARG;
    *on = h->shiftOn[stream]; *semitones = h->shiftSemi[stream];
    return VP_OK;
}

extern "C" int vp_get_stream_params(const vp_handle *h, int stream, vp_params *p)
{
    if (!h || !p) return VP_ERR_INVALID_ARG;
    if (!h->prepared) return VP_ERR_NOT_PREPARED;
    if (stream < 0 || stream >= h->g.S) return VP_ERR_INVALID_ARG;
    *p = h->perStream ? h->sparams[stream] : h->params;
    return VP_OK;
}

extern "C" int vp_get_params(const vp_handle *h, vp_params *p)
{
    if (!h || !p) return VP_ERR_INVALID_ARG;
    *p = h->params;
    return VP_OK;
}

// ---- prepare-time tables (host libm, the same calls the reference makes) -------------------------

// JUCE dsp::WindowingFunction<double>::fillWindowingTables(w, n, hann, false): symmetric Hann
static void fill_hann(double *w, int n)
{
    for (int i = 0; i < n; i++)
        w[i] = 0.5 - 0.5 * std::cos((double)(2 * (long)i) * 3.141592653589793238 / (double)(n - 1));
}

// VocoderProcess::setWindows("sine") (VocoderProcess.cpp:95-135); PI is the truncated literal (:13)
static int fill_voc_window(std::vector<double> &w, int wlen, int hop)
{
    const double PI_REF = 3.14159265;
    double overlap = double(wlen - hop) / double(wlen);
    double overlapFactor = 1.0;
    if (std::fabs(overlap - 0.75) < std::pow(10, -10)) overlapFactor = 1.0 / std::sqrt(2);
    if (std::fabs(overlap - 0.75) > std::pow(10, -10) && std::fabs(overlap - 0.5) > std::pow(10, -10))
        return VP_ERR_INVALID_OVERLAP;
    w.resize(wlen);
    for (int i = 0; i < wlen; i++) w[i] = overlapFactor * std::sin((i + 0.5) * PI_REF / double(wlen));
    return VP_OK;
}

// PitchProcess::buildWindows (PitchProcess.cpp:889-905): half-Hann | ones | half-Hann
static int fill_pitch_st_window(std::vector<double> &w, int frameLen, int hop)
{
    double overlap = ((double)(frameLen - hop)) / ((double)(frameLen));
    int ov = (int)std::round(overlap * frameLen);
    int nh = 2 * ov;
    if (nh > frameLen || nh < 2) return VP_ERR_GEOMETRY;
    std::vector<double> hw(nh);
    fill_hann(hw.data(), nh);
    w.assign(frameLen, 1.0);
    for (int i = 0; i < ov; i++) w[i] = hw[i];
    for (int i = 0; i < ov; i++) w[frameLen - ov + i] = hw[ov + i];
    return VP_OK;
}

// Notes::buildFreqVect (Notes.cpp:43-70); tab[size] keeps the popped element (:69, read at :99)
static int build_notes(int key, double fMin, double fMax, double *tab)
{
    static const int intervals[7] = {2, 2, 1, 2, 2, 2, 1};
    int n = 0, i = 0;
    double f = 27.5;
    f = f * std::pow(2, double(key) / 12.0);
    double factorSemiTone = std::pow(2, 1.0 / 12);
    while (n == 0 || tab[n - 1] < fMax) {
        if (key != 12) f = f * std::pow(factorSemiTone, intervals[i % 7]);
        else f = f * factorSemiTone;
        if (f > fMin) { if (n < VP_NOTES_STRIDE) tab[n] = f; n++; }
        i += 1;
    }
    return n - 1;
}

// Decibels::gainToDecibels(getRMSLevel(...)) < silenceThresholdDb as a function of sum(x^2)
static bool gate_closed(double sum, int n, double silenceDb)
{
    double rms = std::sqrt(sum / n);
    double db = rms > 0.0 ? std::max(-100.0, std::log10(rms) * 20.0) : -100.0;
    return db < silenceDb;
}
static double gate_threshold_sum(int n, double silenceDb)
{
    // the verdict is monotone in sum: bisect over the bit patterns of the positive doubles
    uint64_t lo = 0, hi;
    double big = 1e300;
    memcpy(&hi, &big, 8);
    while (lo + 1 < hi) {
        uint64_t mid = lo + (hi - lo) / 2;
        double v;
        memcpy(&v, &mid, 8);
        if (gate_closed(v, n, silenceDb)) lo = mid; else hi = mid;
    }
    double v;
    memcpy(&v, &hi, 8);
    return v;      // smallest sum for which the gate is open
}

// Decibels::decibelsToGain<float>(dB, -59.0f)
static float db_to_gain_f(float dB) { return dB > -59.0f ? std::pow(10.0f, dB * 0.05f) : 0.0f; }

template <typename T>
static int dev_alloc(vp_handle *h, T **p, size_t count, bool zero = true)
{
    void *q = nullptr;
    hipError_t e = vp_malloc(h, &q, std::max<size_t>(count, 1) * sizeof(T));
    if (e != hipSuccess) { h->lastError = "hipMalloc failed"; return VP_ERR_OOM; }
    h->allocs.push_back(q);
    if (zero) HIPCHK(h, hipMemset(q, 0, std::max<size_t>(count, 1) * sizeof(T)));
    *p = (T *)q;
    return VP_OK;
}
template <typename T>
static int dev_upload(vp_handle *h, const T **p, const std::vector<T> &v)
{
    T *q = nullptr;
    int rc = dev_alloc(h, &q, v.size(), false);
    if (rc) return rc;
    HIPCHK(h, hipMemcpy(q, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice));
    *p = q;
    return VP_OK;
}
#define RC(x) do { int _rc = (x); if (_rc) { free_all(h); return _rc; } } while (0)

extern "C" int vp_prepare_explicit(vp_handle *h, double fs, int N, int S, int F, int H, int W, int hop)
{
    if (!h || N <= 0 || S <= 0 || F <= 0 || H <= 0 || W <= 0 || hop <= 0 || !(fs > 0)) return VP_ERR_INVALID_ARG;
    if (hipSetDevice(h->device) != hipSuccess) return VP_ERR_NO_DEVICE;
    (void)hipDeviceSynchronize();
    free_all(h);

    const double fMin = 100, fMax = 800, silenceDb = -60.0;                  // PluginProcessor.cpp:148,172
    VpGeom g;
    memset(&g, 0, sizeof g);
    g.S = S; g.N = N; g.F = F; g.H = H; g.W = W; g.h = hop; g.fs = fs;
    g.C = F - H;                                                             // PitchProcess.cpp:91-92
    if (g.C <= 0 || F % g.C != 0) return VP_ERR_GEOMETRY;
    g.cpf = F / g.C;
    if (g.cpf < 2) return VP_ERR_GEOMETRY;
    g.toKeep = F;                                                            // PluginProcessor.cpp:176
    g.latency = std::max(F, W);                                              // :175
    g.inSize = g.toKeep + N + g.latency;                                     // MyBuffer.cpp:46-48
    g.outSize = N + g.latency;
    g.tauMax = (int)std::ceil(fs / fMin);                                    // PitchProcess.cpp:100
    if (g.tauMax > g.toKeep) return VP_ERR_GEOMETRY;                         // YIN reads back to startSample - tauMax (:364)
    g.tau0 = (int)std::floor(fs / fMax);
    if (g.tau0 < 1) return VP_ERR_GEOMETRY;
    g.eLen = g.toKeep + F + (g.cpf - 1) * g.C;
    g.orderPitch = h->params.lpcPitch;                                       // read once (PitchProcess.cpp:70)
    g.bufferIdxMax = g.latency + N;                                          // PitchProcess.cpp:138
    g.delta = 0.94; g.yinTol = 0.25;                                         // :76,84
    g.gateThrSum = gate_threshold_sum(g.inSize, silenceDb);
    g.levEps = std::pow(10, -9);
    g.eeFloor = std::pow(10, -4);

    std::vector<double> vocWin, pitchSt;
    int rc = fill_voc_window(vocWin, W, hop);
    if (rc) return rc;
    rc = fill_pitch_st_window(pitchSt, F, H);
    if (rc) return rc;

    // Hann(2T+1) for every possible PSOLA period (PitchProcess.cpp:878-882), T = 1..tauMax
    std::vector<int> hannOff(g.tauMax + 1, 0);
    size_t tot = 0;
    for (int T = 1; T <= g.tauMax; T++) { hannOff[T] = (int)tot; tot += 2 * (size_t)T + 1; }
    std::vector<double> hannTab(tot);
    for (int T = 1; T <= g.tauMax; T++) fill_hann(hannTab.data() + hannOff[T], 2 * T + 1);

    std::vector<double> notes(13 * VP_NOTES_STRIDE, 0.0);
    std::vector<int> notesN(13);
    for (int k = 0; k < 13; k++) notesN[k] = build_notes(k, fMin, fMax, notes.data() + (size_t)k * VP_NOTES_STRIDE);

    // LDS budgets
    hipDeviceProp_t prop;
    HIPCHK(h, hipGetDeviceProperties(&prop, h->device));
    // (minus the kernels' few static reduction slots, which come out of the same 160 KB)
    const size_t ldsMax = std::min<size_t>((size_t)prop.maxSharedMemoryPerMultiProcessor ? prop.maxSharedMemoryPerMultiProcessor : 65536, 160 * 1024) - 256;
    // stage the voice window of as many consecutive chunk steps as fit comfortably (the whole block if possible)
    g.xsSteps = 1;
    // batches that can use the two-workgroups-per-CU builds: no LDS copy of the frame's Hann window (read from the global table,
    // same values), which lets the block's later chunk steps be staged inside the 80 KB -- and with them the overlap of the next
    // chunk's residual / PSOLA with the current chunk's recursion
    // (exactly the batches the register-light builds serve -- pitch_lite(): more than 256 streams and a frame that fits half a CU
    // with one step staged; they are compiled for the global table only)
    g.htabGlobal = 0;
    if (g.S > 256) {
        VpGeom t = g;
        t.xsSteps = 1;
        if (vp_pitch_lds_bytes(t) <= (size_t)80 * 1024) g.htabGlobal = 1;
    }
    {
        const int stepsMax = (N + g.C - 1) / g.C;
        // batches that can use the two-workgroups-per-CU build keep the frame within half a CU's LDS
        size_t budget = std::min<size_t>(ldsMax, (g.S > 256 ? 80 : 112) * 1024);
        {   // a frame that is beyond that budget even with one step staged has a CU to itself whatever is done here: use it
            VpGeom t = g;
            t.xsSteps = 1;
            if (vp_pitch_lds_bytes(t) > budget) budget = ldsMax;
        }
        for (int k = stepsMax; k >= 1; k--) {
            VpGeom t = g;
            t.xsSteps = k;
            if ((size_t)(g.toKeep + F + (k - 1) * g.C) < (size_t)g.inSize && vp_pitch_lds_bytes(t) <= budget) { g.xsSteps = k; break; }
        }
    }
    h->pitchLds = vp_pitch_lds_bytes(g);
    h->ldsMax = ldsMax;
    if (h->pitchLds > ldsMax) { h->lastError = "pitch frame does not fit LDS"; return VP_ERR_GEOMETRY; }
    int nw = 8;
    while (nw > 1 && vp_voc_lds_bytes(W, nw) > ldsMax) nw--;
    if (const char *e = getenv("VP_VOC_WAVES")) { int v = atoi(e); if (v >= 1 && v < nw) nw = v; }    // diagnostic override
    if (vp_voc_lds_bytes(W, nw) > ldsMax) { h->lastError = "vocoder window does not fit LDS"; return VP_ERR_GEOMETRY; }
    h->vocWaves = nw;
    h->vocLds = vp_voc_lds_bytes(W, nw);
    // the dynamic-LDS ceiling is a property of the FUNCTION, shared by every handle of the process: always the device's
    // maximum, never this handle's own need (a later prepare of a smaller geometry must not lower it under another handle)
    {
        const void *fns[] = {(const void *)vp_k_pitch, (const void *)vp_k_pitch_fast, (const void *)vp_k_pitch_multi,
                             (const void *)vp_k_pitch_fast_multi, (const void *)vp_k_pitch_lite, (const void *)vp_k_pitch_lite_fast,
                             (const void *)vp_k_pitch_c, (const void *)vp_k_pitch_fast_c, (const void *)vp_k_pitch_fast_multi_c,
                             (const void *)vp_k_pitch_lite_fast_c, (const void *)vp_k_pitch_lite_fast_multi, (const void *)vp_k_pitch_lite_fast_multi_c,
                             (const void *)vp_k_pitch_ws, (const void *)vp_k_pitch_ws_x, (const void *)vp_k_pitch_ws_mb, (const void *)vp_k_pitch_ws_x_mb, (const void *)vp_k_pitch_ws_mb_o24, (const void *)vp_k_pitch_ws_x_mb_o24, (const void *)vp_k_pitch_ws_o24, (const void *)vp_k_pitch_ws_x_o24,
                             (const void *)vp_k_vocoder, (const void *)vp_k_vocoder_o48, (const void *)vp_k_vocoder_lite};
        for (const void *f : fns) {
            hipFuncAttributes fa;
            HIPCHK(h, hipFuncGetAttributes(&fa, f));                        // the static part (a few reduction slots) comes off the top
            HIPCHK(h, hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(ldsMax + 256 - fa.sharedSizeBytes)));
        }
    }

    VpDev d;
    memset(&d, 0, sizeof d);
    RC(dev_alloc(h, &d.voiceRing, (size_t)S * g.inSize));
    RC(dev_alloc(h, &d.synthRing, (size_t)S * 2 * g.inSize));
    RC(dev_alloc(h, &d.outAcc, (size_t)S * g.outSize));
    RC(dev_alloc(h, &h->acc2, (size_t)S * g.outSize));
    h->acc2Live = 0;
    RC(dev_alloc(h, &d.gate, (size_t)S * 2));
    RC(dev_alloc(h, &d.pitch, (size_t)S));
    RC(dev_alloc(h, &d.eFrame, (size_t)S * g.eLen));
    RC(dev_alloc(h, &d.outEFrame, (size_t)S * F));
    RC(dev_alloc(h, &d.yFrame, (size_t)S * F));
    RC(dev_alloc(h, &d.EeArr, (size_t)S * 20));
    RC(dev_alloc(h, &d.hImp, (size_t)S * 128));
    RC(dev_alloc(h, &d.ub, (size_t)5));
    RC(dev_alloc(h, &d.dbg, (size_t)64 + (size_t)S + 512));  // [64] phase timers / counters, then (diagnostic build) per-stream kernel ticks, then the
                                                             // wave-specialised pitch kernel's per-wavefront timers [4 block types][16 wavefronts][8]
    vocWin.resize((size_t)W + 16, 0.0);                                      // (zero padding: vp_k_v2_autocorr fetches whole 8-entry stretches)
    RC(dev_upload(h, &d.vocWin, vocWin));
    RC(dev_upload(h, &d.pitchStWin, pitchSt));
    RC(dev_upload(h, &d.hannTab, hannTab));
    RC(dev_upload(h, &d.hannOff, hannOff));
    RC(dev_upload(h, &d.notes, notes));
    RC(dev_upload(h, &d.notesN, notesN));
    {   // per-lane twiddles of the wavefront FFT (vp_fft.inc fft512) and of the real-input split, host libm
        const double PI = 3.141592653589793238;
        std::vector<double> t1(64 * 8 * 2), t2(64 * 8 * 2), ts(64 * 4 * 2);
        for (int lane = 0; lane < 64; lane++) {
            for (int r = 0; r < 8; r++) {
                const double a1 = -2.0 * PI * (double)(r * (lane >> 3)) / 64.0, a2 = -2.0 * PI * (double)(r * lane) / 512.0;
                t1[(lane * 8 + r) * 2] = std::cos(a1); t1[(lane * 8 + r) * 2 + 1] = std::sin(a1);
                t2[(lane * 8 + r) * 2] = std::cos(a2); t2[(lane * 8 + r) * 2 + 1] = std::sin(a2);
            }
            for (int q = 0; q < 4; q++) {
                const double a = -2.0 * PI * (double)(64 * q + lane) / 1024.0;
                ts[(lane * 4 + q) * 2] = std::cos(a); ts[(lane * 4 + q) * 2 + 1] = std::sin(a);
            }
        }
        RC(dev_upload(h, &d.fftTw1, t1));
        RC(dev_upload(h, &d.fftTw2, t2));
        RC(dev_upload(h, &d.fftTws, ts));
    }
    RC(dev_alloc(h, &h->dMapAll, (size_t)S));
    {   // scratch of the batched vocoder pipeline (a block has at most ceil(N / hop) windows per stream)
        memset(&h->v2, 0, sizeof h->v2);
        h->nWinMax = (N + hop - 1) / hop;
        const size_t NW = (size_t)S * h->nWinMax;
        if (h->nWinMax <= 64 && vp_v2_init() == 0) {
            h->v2.nGroupsMax = (int)((NW + 63) / 64);
            h->v2.W4p = (W + 3) / 4 + 16;                                 // (+ padding: the autocorrelation requests its samples two trips ahead)
            h->v2.W2p = (W + 1) / 2 + 2;
            RC(dev_alloc(h, &h->v2.xT, (size_t)2 * h->v2.nGroupsMax * h->v2.W4p * 256));
            RC(dev_alloc(h, &h->v2.eT, (size_t)2 * h->v2.nGroupsMax * h->v2.W2p * 128));
            RC(dev_alloc(h, &h->v2.out, NW * W));
            const size_t NWp = (size_t)h->v2.nGroupsMax * 64;
            RC(dev_alloc(h, &h->v2.rV, NWp * V2_RV_STRIDE));
            RC(dev_alloc(h, &h->v2.aV, NWp * V2_RV_STRIDE));
            RC(dev_alloc(h, &h->v2.rS, NWp * V2_RS_STRIDE));
            RC(dev_alloc(h, &h->v2.aS, NWp * V2_RS_STRIDE));
            RC(dev_alloc(h, &h->v2.meta, NWp));
            RC(dev_alloc(h, &h->v2.rank, NWp + S));
            RC(dev_alloc(h, &h->v2.liveList, NWp));
            RC(dev_alloc(h, &h->v2.EE, NW * 2));
            h->v2.nSlices = (W + V2_FIR_SLICE - 1) / V2_FIR_SLICE;
            RC(dev_alloc(h, &h->v2.EEp, NW * 2 * h->v2.nSlices));
        }
    }
    RC(dev_alloc(h, &h->stageIn, (size_t)S * 3 * N, false));
    RC(dev_alloc(h, &h->stageOut, (size_t)S * 3 * N, false));
    {   // PitchProcess::prepare initial members (:76-85): everything 0 except beta = 1
        std::vector<VpPitchState> init(S);
        memset(init.data(), 0, init.size() * sizeof(VpPitchState));
        for (auto &p : init) p.beta = 1;
        if (hipMemcpy(d.pitch, init.data(), init.size() * sizeof(VpPitchState), hipMemcpyHostToDevice) != hipSuccess) {
            free_all(h);
            return VP_ERR_HIP;
        }
    }
    d.fault = h->faultDev; d.spinLimit = h->spinLimit;
    *(volatile unsigned int *)h->faultHost = 0;                              // (the device is idle: prepare synchronised it)
    h->g = g;
    h->d = d;
    h->inCounter = g.toKeep + g.latency;                                     // MyBuffer.cpp:60-62
    h->outCounter = 0;
    h->currCounter = g.toKeep;
    // VocoderProcess.cpp:39, PitchProcess.cpp:85,90: startSample = 0, nChunk = 0 for every instance
    h->cohorts.assign(1, vp_handle::Cohort{h->params.pitchBool, h->params.vocBool, 0, 0, 0, S, nullptr, {}});
    h->cohortsDirty = false; h->poisoned = false; h->poisonCode = VP_ERR_HIP;
    h->sparams.assign((size_t)S, h->params);                                 // prepare starts every stream from the handle's set
    h->spHost.assign((size_t)S, VpStreamParams{});
    h->synthNonZero = 0;                                                     // the rings start zeroed
    h->shiftOn.assign((size_t)S, 0); h->shiftSemi.assign((size_t)S, 0.0); h->shiftBeta.assign((size_t)S, 1.0);
    h->perStream = false; h->spDirty = true;                                 // the fresh (zeroed) state needs them
    h->prepared = true;
    (void)hipDeviceSynchronize();
    return VP_OK;
}

extern "C" int vp_prepare_to_play(vp_handle *h, double sampleRate, int samplesPerBlock, int nStreams)
{
    if (!h || !(sampleRate > 0)) return VP_ERR_INVALID_ARG;
    // PluginProcessor.cpp:159-170
    double ratioSR = sampleRate / 44100.0;
    int hopVoc = (int)std::floor(128.0 * ratioSR);
    int wlenVoc = 4 * hopVoc;
    int corres_256 = (int)std::floor(256.0 * ratioSR);
    int hopPitch = 3 * corres_256;
    int frameLenPitch = 4 * corres_256;
    return vp_prepare_explicit(h, sampleRate, samplesPerBlock, nStreams, frameLenPitch, hopPitch, wlenVoc, hopVoc);
}

static hipEvent_t get_event(vp_handle *h)
{
    if (!h->evPool.empty()) { hipEvent_t e = h->evPool.back(); h->evPool.pop_back(); return e; }
    hipEvent_t e;
    (void)hipEventCreate(&e);
    return e;
}

struct ProfScope {
    vp_handle *h; hipStream_t st; int slot; hipEvent_t a, b; bool on;
    ProfScope(vp_handle *h_, hipStream_t st_, int slot_) : h(h_), st(st_), slot(slot_)
    {
        on = h->profThis;
        if (on) { a = get_event(h); b = get_event(h); (void)hipEventRecord(a, st); }
    }
    ~ProfScope()
    {
        if (on) { (void)hipEventRecord(b, st); h->pending.push_back({a, b, slot}); }
    }
};

// float gains via decibelsToGain computed in float, switches of the dry paths (PluginProcessor.cpp:214-230)
static void fill_stream_params(VpStreamParams &o, const vp_params &P)
{
    o.orderVoice = P.lpcVoice; o.orderSynth = P.lpcSynth; o.key = P.keyPitch;
    o.dryOn = ((double)P.gainVoice > -59.0);                                 // PluginProcessor.cpp:226
    o.synthOn = ((double)P.gainSynth > -59.0);                               // :229
    o.gainPitch = (double)db_to_gain_f(P.gainPitch);
    o.gainVoc = (double)db_to_gain_f(P.gainVoc);
    o.gainVoice = (double)db_to_gain_f(P.gainVoice);
    o.gainSynth = (double)db_to_gain_f(P.gainSynth);
}

// Streams are grouped by (pitchBool, vocBool, VocoderProcess::startSample, PitchProcess::startSample, nChunk): the
// switches of a stream decide whether its two processes advance their counters in a block (PluginProcessor.cpp:214-221).
// Called at the top of a process call when a set call touched a switch.
static int rebuild_cohorts(vp_handle *h, hipStream_t st)
{
    const int S = h->g.S;
    struct Key { int pitchOn, vocOn, vStart, pStart, nChunk, ordHi; };
    std::vector<Key> cur((size_t)S);
    for (const auto &co : h->cohorts) {
        const Key k{co.pitchOn, co.vocOn, co.vStart, co.pStart, co.nChunk, co.ordHi};
        if (!co.dMap) for (int i = 0; i < S; i++) cur[i] = k;
        else for (int i : co.ids) cur[i] = k;
    }
    std::vector<vp_handle::Cohort> out;
    for (int i = 0; i < S; i++) {
        Key k = cur[i];
        k.pitchOn = h->sparams[i].pitchBool; k.vocOn = h->sparams[i].vocBool;
        k.ordHi = (voc_pipeline_wanted(h) && k.vocOn && h->sparams[i].lpcVoice > V2_ORDER_MAX) ? 1 : 0;
        vp_handle::Cohort *hit = nullptr;
        for (auto &co : out)
            if (co.pitchOn == k.pitchOn && co.vocOn == k.vocOn && co.vStart == k.vStart && co.pStart == k.pStart && co.nChunk == k.nChunk && co.ordHi == k.ordHi) { hit = &co; break; }
        if (!hit) { out.push_back(vp_handle::Cohort{k.pitchOn, k.vocOn, k.vStart, k.pStart, k.nChunk, 0, nullptr, {}}); out.back().ordHi = k.ordHi; hit = &out.back(); }
        hit->ids.push_back(i);
        hit->oVmax = std::max(hit->oVmax, h->sparams[i].lpcVoice); hit->oSmax = std::max(hit->oSmax, h->sparams[i].lpcSynth);
    }
    if (out.size() == 1) { out[0].ids.clear(); out[0].n = S; out[0].dMap = nullptr; }
    else {
        h->mapHost.clear();
        for (auto &co : out) {
            co.n = (int)co.ids.size();
            co.dMap = h->dMapAll + h->mapHost.size();
            h->mapHost.insert(h->mapHost.end(), co.ids.begin(), co.ids.end());
        }
        // (pageable source: the copy has left the host buffer when the call returns; stream-ordered in front of the launches)
        HIPCHK(h, hipMemcpyAsync(h->dMapAll, h->mapHost.data(), (size_t)S * sizeof(int), hipMemcpyHostToDevice, st));
    }
    h->cohorts.swap(out);
    h->cohortsDirty = false;
    return VP_OK;
}

// What process_device does first: parameters that travel in device state, and the cohorts
static int sync_stream_state(vp_handle *h, hipStream_t st)
{
    const VpGeom &g = h->g;
    if (h->spDirty) {
        // orders, key, gains and the dry-path switches travel in each stream's device state (VpPitchState::sp);
        // rewritten here, stream-ordered in front of this block's kernels, whenever a set call changed them
        h->oVmax = h->oSmax = 0;
        for (int i = 0; i < g.S; i++) {
            fill_stream_params(h->spHost[i], h->sparams[i]);
            h->spHost[i].shiftOn = h->shiftOn[i]; h->spHost[i].shiftBeta = h->shiftBeta[i];
            h->oVmax = std::max(h->oVmax, h->sparams[i].lpcVoice); h->oSmax = std::max(h->oSmax, h->sparams[i].lpcSynth);
        }
        HIPCHK(h, hipMemcpy2DAsync(&h->d.pitch[0].sp, sizeof(VpPitchState), h->spHost.data(), sizeof(VpStreamParams),
                                   sizeof(VpStreamParams), (size_t)g.S, hipMemcpyHostToDevice, st));
        h->spDirty = false;
        // the cohorts carry their own order maxima (and are keyed by the order class when the pipeline is in play)
        if (h->cohorts.size() == 1 && !h->perStream) { h->cohorts[0].oVmax = h->oVmax; h->cohorts[0].oSmax = h->oSmax; if (h->cohorts[0].ordHi) h->cohortsDirty = true; }
        else h->cohortsDirty = true;
    }
    if (h->cohortsDirty) return rebuild_cohorts(h, st);
    return VP_OK;
}

static int process_device(vp_handle *h, const float *d_in, float *d_out, hipStream_t st, int inplace, int nBlocks = 1, bool mono = false)
{
    const VpGeom &g = h->g;
    if (h->poisoned) return poisoned_rc(h);
    if (int frc = check_fault(h)) return frc;                                 // (an earlier launch of this handle timed out)
    const vp_params P = h->params;                                           // snapshot at call entry
    h->profThis = h->prof > 0 && (h->profTick++ % (unsigned)h->prof) == 0;   // all kernels of every k-th call
    if (P.lpcVoice > VP_ORDER_MAX || P.lpcSynth > VP_ORDER_MAX_SYNTH) return VP_ERR_ORDER;
    struct Poison { vp_handle *h; bool armed; ~Poison() { if (armed) h->poisoned = true; } } guard{h, true};   // disarmed on success
    int rc = sync_stream_state(h, st);
    if (rc) return rc;
    VpCall c0;
    memset(&c0, 0, sizeof c0);
    c0.inCounter = h->inCounter; c0.outCounter = h->outCounter; c0.currCounter = h->currCounter;
    c0.inplace = inplace;
    c0.iirFast = h->iirMode;
    c0.nBlocks = 1;
    // side-chain bus present or absent; the synth ring is written N samples per call, so after inSize samples of zeros
    // (or straight after prepare) it holds nothing else and the mono path need not touch it
    if (mono) {
        c0.inMono = h->synthNonZero > 0 ? 1 : 2;
        h->synthNonZero = std::max(0, h->synthNonZero - g.N * nBlocks);
    } else {
        c0.inMono = 0;
        h->synthNonZero = g.inSize;
    }
    c0.yinCert = (h->yinMode == VP_YIN_XCORR || h->yinMode == VP_YIN_FFT) ? 1 : (h->yinMode == VP_YIN_XCORR_FORCE_FALLBACK) ? 2 : 0;

    for (auto &co : h->cohorts) {
        VpCall c = c0;
        VpDev d = h->d;
        d.streamMap = co.dMap;                                               // nullptr: workgroup b serves stream b
        d.outAcc2 = h->acc2Live > 0 ? h->acc2 : nullptr;                     // emit merges the second accumulator while it may hold anything
        c.pitchOn = co.pitchOn; c.vocOn = co.vocOn;
        // VocoderProcess::process (VocoderProcess.cpp:173-183): windows while startSample < N
        c.vStart = co.vStart;
        c.nWin = 0;
        if (c.vocOn && co.vStart < g.N) c.nWin = (g.N - co.vStart + g.h - 1) / g.h;
        // PitchProcess::process (PitchProcess.cpp:166-196): chunk steps while startSample < N
        c.pStart = co.pStart; c.nChunk0 = co.nChunk;
        c.nSteps = 0;
        if (c.pitchOn && co.pStart < g.N) c.nSteps = (g.N - co.pStart + g.C - 1) / g.C;

        // Kernel plan: ingest+gate runs as the prologue of the first DSP kernel and emit as the epilogue of
        // the last one (all stages are one-workgroup-per-stream); they only stand alone when no DSP kernel runs.
        // A launch that carries several blocks (pitch-only plan, process_blocks_device) runs the pitch kernel even when
        // its FIRST block has no chunk step to do (N smaller than the chunk): later blocks of the launch do.
        const bool runVoc = c.nWin > 0, runPitch = c.nSteps > 0 || (nBlocks > 1 && c.pitchOn);
        bool runPitchDone = false;
        if (!runVoc && !runPitch) {
            { ProfScope ps(h, st, 0); hipLaunchKernelGGL(vp_k_ingest_gate, dim3(co.n), dim3(256), 0, st, g, c, d, d_in); }
            { ProfScope ps(h, st, 3); hipLaunchKernelGGL(vp_k_emit, dim3(co.n), dim3(256), 0, st, g, c, d, d_out); }
        } else {
            const bool batched = runVoc && voc_batched_for(h, c.nWin, co.oVmax, co.oSmax);
            if (batched && runPitch && c.iirFast && nBlocks == 1 && (h->overlap == 1 || (h->overlap == VP_OVERLAP_AUTO && !pitch_lite(h, true)))) {
                // VP_IIR_FAST (tolerance mode), both processes, batched vocoder: the pitch kernel runs on a second HIP stream BESIDE
                // the pipeline's last two kernels and adds into an accumulator of its own.  (Round 6) the fork sits behind the
                // residual kernel: autocorrelation, Levinson-Durbin and the residual FIRs need 130-300 registers per wavefront and
                // would only take turns with the pitch workgroups; vp_k_v2_iir_fast (64 registers, no LDS, a serial chain per window
                // that leaves half the chip idle at the configs[4] geometry) and vp_k_v2_ola (26) fit into what the pitch
                // full-register builds leave of every SIMD's register file (vp_k_pitch_fast: 2 x 224 of 512), so they run in its shadow:
                // configs[4] geometry 419 -> 376 us per block (same box).  The register-light builds fill the file (4 x 128): a 112-register
                // build of vp_k_pitch_lite_fast_c (13 spills) did hide the two kernels, but ran 10 us longer itself, and the plan's own
                // costs -- the separate emit kernel, two cross-stream waits -- ate the rest: 276 against 274 us at 1024 streams.  So
                // VP_OVERLAP_AUTO (the default) takes this plan for the full-register builds only.
                // What is given up is the ORDER of the additions into the output accumulator (windows, then chunks, per
                // block: PluginProcessor.cpp:214-221) -- rounding-level; VP_IIR_EXACT keeps the sequential plan below.
                struct Fork { vp_handle *h; hipStream_t st; VpGeom g; VpCall cp; VpDev dp; const float *in; float *out; int n; hipError_t err; } fk;
                fk.h = h; fk.st = st; fk.g = g; fk.cp = c; fk.dp = d; fk.in = d_in; fk.out = d_out; fk.n = co.n; fk.err = hipSuccess;
                fk.cp.fuseIngest = 0; fk.cp.fuseEmit = 0; fk.cp.nBlocks = 1;
                fk.dp.outAcc = h->acc2; fk.dp.outAcc2 = nullptr;
                auto fork = [](void *a) {
                    Fork *f = (Fork *)a;
                    vp_handle *h = f->h;
                    if ((f->err = hipEventRecord(h->evFork, f->st)) != hipSuccess) return;
                    if ((f->err = hipStreamWaitEvent(h->auxStream, h->evFork, 0)) != hipSuccess) return;
                    ProfScope ps(h, h->auxStream, 2);
                    PitchPlan plan = pitch_plan(h, true, 1);
                    if (const int fw = f->cp.yinCert ? pitch_xfft_waves(h, true, 1) : 0) {
                        const size_t off = (plan.lds + 15) / 16 * 16;
                        f->cp.fftOff = (int)off; f->cp.fftWaves = fw; plan.lds = off + pitch_xfft_lds(h);
                    }
                    f->cp.ldsBytes = (int)plan.lds;
                    hipLaunchKernelGGL(plan.fn, dim3(f->n), dim3(512), plan.lds, h->auxStream, f->g, f->cp, f->dp, f->in, f->out);
                };
                VpCall cv = c;
                cv.fuseIngest = 1; cv.fuseEmit = 0;
                VpV2 v = h->v2;
                v.nStreams = co.n; v.oVmax = co.oVmax; v.oSmax = co.oSmax;
                { ProfScope ps(h, st, 1); vp_v2_launch(g, cv, d, v, d_in, d_out, st, fork, &fk); }
                if (fk.err != hipSuccess) return fail_hip(h, fk.err, "fork of the pitch kernel");
                HIPCHK(h, hipEventRecord(h->evJoin, h->auxStream));
                HIPCHK(h, hipStreamWaitEvent(st, h->evJoin, 0));
                VpDev de = d;
                de.outAcc2 = h->acc2;
                { ProfScope ps(h, st, 3); hipLaunchKernelGGL(vp_k_emit, dim3(co.n), dim3(256), 0, st, g, c, de, d_out); }
                h->acc2Live = (g.outSize + g.N - 1) / g.N + 1;               // (+1: this block's own decrement below)
                runPitchDone = true;
            } else if (batched) {
                // large batches: the pipeline of lane-per-window kernels (vp_voc2.hip), ingest+gate in front, emit behind
                VpCall cv = c;
                cv.fuseIngest = 1; cv.fuseEmit = runPitch ? 0 : 1;
                ProfScope ps(h, st, 1);
                VpV2 v = h->v2;
                v.nStreams = co.n; v.oVmax = co.oVmax; v.oSmax = co.oSmax;
                vp_v2_launch(g, cv, d, v, d_in, d_out, st);
            } else if (runVoc) {
                VpCall cv = c;
                cv.fuseIngest = 1; cv.fuseEmit = runPitch ? 0 : 1;
                ProfScope ps(h, st, 1);
                int nw = std::min(h->vocWaves, c.nWin);
                if (nw < 4) nw = std::min(4, h->vocWaves);        // the fused ingest/emit want a few waves
                // large batches, FAST IIR: the register-light build with half the window slots, so that two workgroups share a
                // CU (each slot then has two wavefronts; needs <= 80 KB of LDS per workgroup)
                const int nl = voc_lite_slots(h, cv.iirFast != 0, nw);
                const bool lite = nl > 0;
                if (lite) nw = nl;
                cv.ldsBytes = (int)vp_voc_lds_bytes(g.W, nw);
                cv.vocWin = nw;                                   // window slots per round; spare wavefronts (up to as many again) help
                const int nThreads = 64 * nw * std::max(1, 8 / nw);           // a whole number of wavefronts per window slot, at most 8
                // (orders 33..48: the build that carries the big register-resident instantiations, see vp_vocoder_wg.inc)
                hipLaunchKernelGGL(lite ? vp_k_vocoder_lite : voc_o48(co.oVmax) ? vp_k_vocoder_o48 : vp_k_vocoder, dim3(co.n), dim3(nThreads), vp_voc_lds_bytes(g.W, nw), st,
                                   g, cv, d, d_in, d_out);
            }
            if (runPitch && !runPitchDone) {
                VpCall cp = c;
                cp.fuseIngest = runVoc ? 0 : 1; cp.fuseEmit = 1;
                cp.nBlocks = nBlocks;                          // > 1 only from process_blocks_device, pitch-only plan
                ProfScope ps(h, st, 2);
                PitchPlan plan = pitch_plan(h, cp.iirFast != 0, nBlocks);
                // one workgroup per CU anyway (S <= 256 or a frame beyond half a CU's LDS): the block's accumulator slice rides in LDS
                // the block's accumulator slice in LDS when it fits beside the FFT cross-correlation's minimum (tables + one buffer)
                const int fw = cp.yinCert ? pitch_xfft_waves(h, cp.iirFast != 0, nBlocks) : 0;
                const size_t fftMin = fw ? pitch_xfft_lds(h) + 16 : 0;
                if (nBlocks == 1 && !pitch_lite(h, cp.iirFast != 0) && plan.lds + vp_pitch_acc_lds_bytes(g) + 16 + fftMin <= h->ldsMax) {
                    cp.ldsAcc = 1;
                    plan.lds = ((plan.lds + 15) / 16) * 16 + vp_pitch_acc_lds_bytes(g);
                }
                // certified YIN of the full-register common-case builds: tables and exchange buffers behind everything else
                if (fw) {
                    const size_t off = (plan.lds + 15) / 16 * 16;
                    cp.fftOff = (int)off; cp.fftWaves = fw; plan.lds = off + pitch_xfft_lds(h);
                }
                cp.ldsBytes = (int)plan.lds;
                if (pitch_ws_ok(h, cp.iirFast != 0, nBlocks, cp.nSteps)) {
                    // the wave-specialised kernel: its own carve (the accumulator slice and the FFT's tables are part of it)
                    cp.ldsAcc = 1; cp.fftOff = 0; cp.fftWaves = 0;
                    cp.ldsBytes = (int)vp_pitch_ws_lds_bytes(g, cp.nSteps);
                    static const int wsWaves = [] { const char *e = getenv("VP_WS_WAVES"); const int v = e ? atoi(e) : 0; return (v >= 8 && v <= 12) ? v : 12; }();   // (diagnostic: 8..12)
                    VpWsSched sc;
                    memset(&sc, 0, sizeof sc);
                    if (!ws_build_sched(g, cp.nChunk0, cp.nSteps, sc)) return fail_hip(h, hipErrorInvalidValue, "pitch schedule");   // (pitch_ws_ok's bounds rule this out)
                    const auto wsk = g.orderPitch > 15 ? (cp.iirFast ? vp_k_pitch_ws_o24 : vp_k_pitch_ws_x_o24) : (cp.iirFast ? vp_k_pitch_ws : vp_k_pitch_ws_x);
                    hipLaunchKernelGGL(wsk, dim3(co.n), dim3(64 * wsWaves), (size_t)cp.ldsBytes, st, g, cp, d, sc, d_in, d_out);
                } else
                hipLaunchKernelGGL(plan.fn, dim3(co.n), dim3(512), plan.lds, st, g, cp, d, d_in, d_out);
            }
        }
        hipError_t e = hipGetLastError();
        if (e != hipSuccess) return fail_hip(h, e, "kernel launch");

        // counters: VocoderProcess.cpp:176-182, PitchProcess.cpp:169-195 (once per block handled)
        for (int b = 0; b < nBlocks; b++) {
            if (co.vocOn) {
                const int nWin = (co.vStart < g.N) ? (g.N - co.vStart + g.h - 1) / g.h : 0;
                co.vStart = co.vStart + nWin * g.h - g.N;
            }
            if (co.pitchOn) {
                const int nSteps = (co.pStart < g.N) ? (g.N - co.pStart + g.C - 1) / g.C : 0;
                int nChunk = co.nChunk;
                for (int i = 0; i < nSteps; i++) {
                    if (nChunk % g.cpf == g.cpf - 1) nChunk = 1 % g.cpf;
                    else nChunk += 1;
                }
                co.nChunk = nChunk;
                co.pStart = co.pStart + nSteps * g.C - g.N;
            }
        }
    }
    h->acc2Live = std::max(0, h->acc2Live - nBlocks);
    for (int b = 0; b < nBlocks; b++) {                                      // MyBuffer.cpp:129-132
        h->outCounter = (h->outCounter + g.N) % g.outSize;
        h->inCounter = (h->inCounter + g.N) % g.inSize;
        h->currCounter = (h->currCounter + g.N) % g.inSize;
    }
    guard.armed = false;
    return VP_OK;
}

extern "C" int vp_process_block_device(vp_handle *h, const float *d_in, float *d_out, void *hip_stream)
{
    if (!h || !d_in || !d_out) return VP_ERR_INVALID_ARG;
    if (!h->prepared) return VP_ERR_NOT_PREPARED;
    if (hipSetDevice(h->device) != hipSuccess) return VP_ERR_NO_DEVICE;
    return process_device(h, d_in, d_out, (hipStream_t)hip_stream, 0);
}

// scratch of the batched vocoder pipeline for `nWin` windows per stream (multi-block launches); called by vp_reserve_blocks only
static int ensure_v2mb(vp_handle *h, int nWin)
{
    if (nWin <= h->v2mbWin) return VP_OK;
    for (void *p : h->mbAllocs) (void)hipFree(p);
    h->mbAllocs.clear(); h->v2mbWin = 0;
    const VpGeom &g = h->g;
    VpV2 v;
    memset(&v, 0, sizeof v);
    const size_t NW = (size_t)g.S * nWin;
    v.nGroupsMax = (int)((NW + 63) / 64);
    v.W4p = h->v2.W4p; v.W2p = h->v2.W2p; v.nSlices = h->v2.nSlices;
    const size_t NWp = (size_t)v.nGroupsMax * 64;
    auto get = [&](auto **p, size_t count) -> bool {
        void *q = nullptr;
        if (vp_malloc(h, &q, std::max<size_t>(count, 1) * sizeof(**p)) != hipSuccess) return false;
        h->mbAllocs.push_back(q);
        (void)hipMemset(q, 0, std::max<size_t>(count, 1) * sizeof(**p));
        *p = (std::remove_reference_t<decltype(**p)> *)q;
        return true;
    };
    const bool ok = get(&v.xT, (size_t)2 * v.nGroupsMax * v.W4p * 256) && get(&v.eT, (size_t)2 * v.nGroupsMax * v.W2p * 128) &&
                    get(&v.out, NW * g.W) && get(&v.rV, NWp * V2_RV_STRIDE) && get(&v.aV, NWp * V2_RV_STRIDE) &&
                    get(&v.rS, NWp * V2_RS_STRIDE) && get(&v.aS, NWp * V2_RS_STRIDE) && get(&v.meta, NWp) && get(&v.rank, NWp + g.S) &&
                    get(&v.liveList, NWp) && get(&v.EE, NW * 2) && get(&v.EEp, NW * 2 * v.nSlices) && get(&v.dry, (size_t)g.S * 3 * g.latency);
    if (!ok) {
        for (void *p : h->mbAllocs) (void)hipFree(p);
        h->mbAllocs.clear();
        h->lastError = "out of device memory for the multi-block vocoder scratch";
        return VP_ERR_OOM;
    }
    h->v2mb = v;
    h->v2mbWin = nWin;
    return VP_OK;
}

// vp_process_blocks_device, vocoder-only plan on the batched pipeline: up to V2_MB_MAX consecutive blocks as ONE launch of the
// pipeline (B times the windows = B times the lanes; vp_voc2.hip).  Returns VP_OK + *done = false when the plan does not apply.
// Smallest group of blocks the multi-block plans of the lane-per-window pipeline take (round 6, measured: profiles/r06_blocks_per_call.txt;
// 0: never).  The pipeline's kernels launch a lane per WINDOW: with 8192 windows per block (1024 streams at 512 / 128) one block fills the
// chip, and more blocks per launch only add the plans' fixed costs (ring snapshots, dry-path staging) and push the tiles out of the
// Infinity Cache -- the vocoder-only plan then loses at every group size (118 us per block alone; 134 / 131 / 133 with 2 / 4 / 8), the
// combined plan breaks even at eight (307 / 284 / 279 against 275-277).  With fewer windows the launches are short of wavefronts and the
// plans pay from two blocks on (512 streams: vocoder 87.7 -> 75.8 / 71.4 / 68.0 us per block, both 170.9 -> 162.5 / 153.0 / 144.9; the
// configs[4] geometry, 2048 windows of 2048 samples: vocoder 198 -> 184 / 162 / 168) -- except that the combined plan has no pitch kernel
// beside the pipeline's tail, so where single-block calls have one (VP_OVERLAP_AUTO, the full-register pitch builds: configs[4]) it
// too needs eight blocks to break even (416 / 377 / 374 against 378).  VP_BOTH_MB_MIN overrides (the test suite: 2).
static int v2_mb_min_blocks(const vp_handle *h, const vp_handle::Cohort &co, bool both)
{
    if (const char *e = getenv("VP_BOTH_MB_MIN")) { const int v = atoi(e); if (v >= 2) return v; }
    const bool full = (size_t)co.n * (size_t)h->nWinMax >= 8192;
    if (!both) return full ? 0 : 2;
    const bool tailOverlap = h->overlap == 1 || (h->overlap == VP_OVERLAP_AUTO && !pitch_lite(h, true));
    return (full || tailOverlap) ? 8 : 2;
}

static int process_voc_blocks(vp_handle *h, const float *d_in, float *d_out, int nb, hipStream_t st, bool *done)
{
    *done = false;
    const VpGeom &g = h->g;
    if (h->cohorts.size() != 1 || nb < 2 || nb > V2_MB_MAX) return VP_OK;
    auto &co = h->cohorts[0];
    if (!co.vocOn || co.pitchOn || !h->v2.xT || h->vocPath == VP_VOC_WORKGROUP) return VP_OK;
    { const int mn = v2_mb_min_blocks(h, co, false); if (mn == 0 || nb < mn) return VP_OK; }
    // VP_IIR_FAST: the pipeline's and the workgroup kernel's tolerance-mode roundings differ, so the plan only runs where single-block
    // calls take the pipeline too (the output must not depend on how the caller groups blocks); VP_IIR_EXACT: both give the same bits
    if (h->iirMode == VP_IIR_FAST && !voc_pipeline_wanted(h)) return VP_OK;
    if (co.oVmax > V2_ORDER_MAX || co.oSmax > VP_ORDER_MAX_SYNTH || co.oVmax < 2 || co.oSmax < 2) return VP_OK;
    // vp_k_v2_mb_ola_emit keeps outSize doubles in dynamic LDS: a geometry beyond its ceiling takes the block-by-block plan
    if ((size_t)g.outSize * sizeof(double) > (size_t)VP_V2_MB_LDS_MAX) return VP_OK;
    VpV2MB mb;
    memset(&mb, 0, sizeof mb);
    mb.nBlocks = nb;
    int vs = co.vStart, NWs = 0;
    for (int b = 0; b < nb; b++) {                                           // VocoderProcess.cpp:173-183 block after block
        const int nWin = (vs < g.N) ? (g.N - vs + g.h - 1) / g.h : 0;
        mb.vStart[b] = vs; mb.nWin[b] = nWin; mb.first[b] = NWs;
        NWs += nWin;
        vs = vs + nWin * g.h - g.N;
    }
    if (NWs < 1 || mb.nWin[0] < 1) return VP_OK;                             // (the launch's coordinates hang on block 0's first window)
    // the grid is continuous: window k starts at vStart[0] + k h (in block-0 coordinates)
    for (int b = 1; b < nb; b++)
        if (mb.nWin[b] > 0 && b * g.N + mb.vStart[b] != mb.vStart[0] + mb.first[b] * g.h) return VP_OK;
    if (h->poisoned) return poisoned_rc(h);
    if (int frc = check_fault(h)) return frc;
    if (NWs > h->v2mbWin) return VP_OK;                                      // beyond what vp_reserve_blocks sized: no allocation here
    h->profThis = h->prof > 0 && (h->profTick++ % (unsigned)h->prof) == 0;
    VpCall c;
    memset(&c, 0, sizeof c);
    c.inCounter = h->inCounter; c.outCounter = h->outCounter; c.currCounter = h->currCounter;
    c.iirFast = h->iirMode; c.nBlocks = nb; c.inMono = 0; c.vocOn = 1; c.pitchOn = 0;
    c.vStart = mb.vStart[0]; c.nWin = NWs; c.fuseIngest = 1; c.fuseEmit = 1;
    VpDev d = h->d;
    d.streamMap = co.dMap;
    d.outAcc2 = h->acc2Live > 0 ? h->acc2 : nullptr;
    VpV2 v = h->v2mb;
    v.nStreams = co.n; v.oVmax = co.oVmax; v.oSmax = co.oSmax;
    { ProfScope ps(h, st, 1); vp_v2_launch_blocks(g, c, d, v, mb, d_in, d_out, st); }
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { h->poisoned = true; return fail_hip(h, e, "kernel launch"); }
    co.vStart = vs;
    h->synthNonZero = g.inSize;
    h->acc2Live = std::max(0, h->acc2Live - nb);
    for (int b = 0; b < nb; b++) {                                           // MyBuffer.cpp:129-132
        h->outCounter = (h->outCounter + g.N) % g.outSize;
        h->inCounter = (h->inCounter + g.N) % g.inSize;
        h->currCounter = (h->currCounter + g.N) % g.inSize;
    }
    *done = true;
    return VP_OK;
}

static int ensure_both(vp_handle *h)
{
    if (h->pLin) return VP_OK;
    const VpGeom &g = h->g;
    auto get = [&](auto **p, size_t count) -> bool {
        void *q = nullptr;
        if (vp_malloc(h, &q, count * sizeof(**p)) != hipSuccess) return false;
        h->bothAllocs.push_back(q);
        (void)hipMemset(q, 0, count * sizeof(**p));
        *p = (std::remove_reference_t<decltype(**p)> *)q;
        return true;
    };
    const bool ok = get(&h->snapV, (size_t)g.S * g.inSize) && get(&h->snapS, (size_t)g.S * 2 * g.inSize) &&
                    get(&h->gateB, (size_t)V2_MB_MAX * g.S * 2) && get(&h->pLin, (size_t)g.S * ((size_t)V2_MB_MAX * g.N + g.outSize));
    if (!ok) {
        for (void *p : h->bothAllocs) (void)hipFree(p);
        h->bothAllocs.clear(); h->snapV = h->snapS = nullptr; h->gateB = nullptr; h->pLin = nullptr;
        h->lastError = "out of device memory for the combined multi-block plan";
        return VP_ERR_OOM;
    }
    return VP_OK;
}

// vp_process_blocks_device, BOTH processes on, VP_IIR_FAST, batched vocoder pipeline: up to V2_MB_MAX consecutive blocks as one launch
// of the serial pitch kernel (state on chip across the blocks) followed by one launch of the vocoder pipeline over all the blocks'
// windows.  The pitch kernel goes first: it ingests the blocks (both gates per block -> gateB) and adds its chunks into a linear
// accumulator of the call (pLin); the vocoder kernels read the rings as they stood BEFORE the call from a snapshot, and their
// overlap-add/emit kernel folds pLin in.  What is given up against the block-by-block plan is the order of the additions into the
// output accumulator (chunks before windows instead of windows before chunks, PluginProcessor.cpp:214-221): rounding-level, which is
// why the plan exists in the tolerance mode only.  Returns VP_OK + *done = false when the plan does not apply.
static int process_both_blocks(vp_handle *h, const float *d_in, float *d_out, int nb, hipStream_t st, bool *done)
{
    *done = false;
    const VpGeom &g = h->g;
    if (h->cohorts.size() != 1 || nb < 2 || nb > V2_MB_MAX || h->iirMode != VP_IIR_FAST) return VP_OK;
    { const int mn = v2_mb_min_blocks(h, h->cohorts[0], true); if (mn == 0 || nb < mn) return VP_OK; }
    auto &co = h->cohorts[0];
    // only where single-block calls run the pipeline too (VP_VOC_AUTO: batches above 256 streams; VP_VOC_BATCHED): the vocoder's
    // arithmetic must not depend on how the caller groups blocks (voc_auto_batched: "decided ONCE per prepare")
    if (!co.vocOn || !co.pitchOn || !voc_pipeline_wanted(h)) return VP_OK;
    if (co.oVmax > V2_ORDER_MAX || co.oSmax > VP_ORDER_MAX_SYNTH || co.oVmax < 2 || co.oSmax < 2) return VP_OK;
    if ((size_t)g.outSize * sizeof(double) > (size_t)VP_V2_MB_LDS_MAX) return VP_OK;
    VpV2MB mb;
    memset(&mb, 0, sizeof mb);
    mb.nBlocks = nb; mb.preIngested = 1;
    int vs = co.vStart, NWs = 0;
    for (int b = 0; b < nb; b++) {                                           // VocoderProcess.cpp:173-183 block after block
        const int nWin = (vs < g.N) ? (g.N - vs + g.h - 1) / g.h : 0;
        mb.vStart[b] = vs; mb.nWin[b] = nWin; mb.first[b] = NWs;
        NWs += nWin;
        vs = vs + nWin * g.h - g.N;
    }
    if (NWs < 1 || mb.nWin[0] < 1) return VP_OK;
    for (int b = 1; b < nb; b++)
        if (mb.nWin[b] > 0 && b * g.N + mb.vStart[b] != mb.vStart[0] + mb.first[b] * g.h) return VP_OK;
    if (h->poisoned) return poisoned_rc(h);
    if (int frc = check_fault(h)) return frc;
    if (NWs > h->v2mbWin || !h->pLin) return VP_OK;                          // not reserved (vp_reserve_blocks): block by block, no allocation here
    h->profThis = h->prof > 0 && (h->profTick++ % (unsigned)h->prof) == 0;
    struct Poison { vp_handle *h; bool armed; ~Poison() { if (armed) h->poisoned = true; } } guard{h, true};
    HIPCHK(h, hipMemcpyAsync(h->snapV, h->d.voiceRing, (size_t)g.S * g.inSize * sizeof(float), hipMemcpyDeviceToDevice, st));
    HIPCHK(h, hipMemcpyAsync(h->snapS, h->d.synthRing, (size_t)g.S * 2 * g.inSize * sizeof(float), hipMemcpyDeviceToDevice, st));
    VpCall c;
    memset(&c, 0, sizeof c);
    c.inCounter = h->inCounter; c.outCounter = h->outCounter; c.currCounter = h->currCounter;
    c.iirFast = 1; c.nBlocks = nb; c.inMono = 0; c.vocOn = 1; c.pitchOn = 1;
    c.yinCert = (h->yinMode == VP_YIN_XCORR || h->yinMode == VP_YIN_FFT) ? 1 : (h->yinMode == VP_YIN_XCORR_FORCE_FALLBACK) ? 2 : 0;
    VpDev d = h->d;
    d.streamMap = co.dMap;
    d.gateB = h->gateB;
    {   // the pitch corrector: every block of the call, chunks into the linear accumulator
        VpGeom gp = g;
        gp.outSize = V2_MB_MAX * g.N + g.outSize;
        VpCall cp = c;
        cp.outCounter = 0;
        cp.pStart = co.pStart; cp.nChunk0 = co.nChunk;
        cp.nSteps = (co.pStart < g.N) ? (g.N - co.pStart + g.C - 1) / g.C : 0;
        cp.fuseIngest = 1; cp.fuseEmit = 0; cp.pitchLin = 1;
        VpDev dp = d;
        dp.outAcc = h->pLin; dp.outAcc2 = nullptr;
        PitchPlan plan = pitch_plan(h, true, nb);
        if (const int fw = cp.yinCert ? pitch_xfft_waves(h, true, nb) : 0) {
            const size_t off = (plan.lds + 15) / 16 * 16;
            cp.fftOff = (int)off; cp.fftWaves = fw; plan.lds = off + pitch_xfft_lds(h);
        }
        cp.ldsBytes = (int)plan.lds;
        ProfScope ps(h, st, 2);
        hipLaunchKernelGGL(plan.fn, dim3(co.n), dim3(512), plan.lds, st, gp, cp, dp, d_in, d_out);
    }
    {   // the vocoder: all the windows of the call, from the snapshot and the call's input; its last kernel emits
        VpCall cv = c;
        cv.vStart = mb.vStart[0]; cv.nWin = NWs; cv.fuseIngest = 1; cv.fuseEmit = 1;
        VpDev dv = d;
        dv.voiceRing = h->snapV; dv.synthRing = h->snapS;
        dv.pLin = h->pLin;
        dv.outAcc2 = h->acc2Live > 0 ? h->acc2 : nullptr;
        VpV2 v = h->v2mb;
        v.nStreams = co.n; v.oVmax = co.oVmax; v.oSmax = co.oSmax;
        ProfScope ps(h, st, 1);
        vp_v2_launch_blocks(g, cv, dv, v, mb, d_in, d_out, st);
    }
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail_hip(h, e, "kernel launch");
    co.vStart = vs;
    for (int b = 0; b < nb; b++) {                                           // PitchProcess.cpp:169-195, once per block
        const int nSteps = (co.pStart < g.N) ? (g.N - co.pStart + g.C - 1) / g.C : 0;
        int nChunk = co.nChunk;
        for (int i = 0; i < nSteps; i++) nChunk = (nChunk % g.cpf == g.cpf - 1) ? 1 % g.cpf : nChunk + 1;
        co.nChunk = nChunk;
        co.pStart = co.pStart + nSteps * g.C - g.N;
    }
    h->synthNonZero = g.inSize;
    h->acc2Live = std::max(0, h->acc2Live - nb);
    for (int b = 0; b < nb; b++) {                                           // MyBuffer.cpp:129-132
        h->outCounter = (h->outCounter + g.N) % g.outSize;
        h->inCounter = (h->inCounter + g.N) % g.inSize;
        h->currCounter = (h->currCounter + g.N) % g.inSize;
    }
    guard.armed = false;
    *done = true;
    return VP_OK;
}

// vp_process_blocks*_device, pitch corrector alone, where the wave-specialised kernel serves the geometry: up to WS_MB_MAX queued blocks
// in ONE launch of vp_k_pitch_ws_mb / _x_mb / _mb_o24 / _x_mb_o24 (round 6) -- tracker state, frame in flight, voice window and output
// accumulator stay in LDS from block to block.  Conditions beyond pitch_ws_ok: host blocks of whole chunks that start on the chunk grid
// (every block then runs N / C steps), at most WS_MB_SCHEDS distinct schedules in the call (the plugin's geometry cycles through three).
// Returns VP_OK + *done = false when the plan does not apply.
static int process_ws_blocks(vp_handle *h, const float *d_in, float *d_out, int nb, hipStream_t st, bool mono, bool *done)
{
    *done = false;
    static const bool off = getenv("VP_NO_WS_MB") != nullptr;
    const VpGeom &g = h->g;
    const bool fast = h->iirMode == VP_IIR_FAST;
    if (off || h->cohorts.size() != 1 || nb < 2 || nb > WS_MB_MAX) return VP_OK;
    auto &co = h->cohorts[0];
    if (!co.pitchOn || co.vocOn || co.pStart != 0 || g.N % g.C != 0 || g.N < g.C) return VP_OK;
    const int nSteps = g.N / g.C;
    if (!pitch_ws_ok(h, fast, 1, nSteps)) return VP_OK;
    {   // what the kernel's block boundary is written for (ws_mb_boundary, vp_pitch_ws.inc): fixed trip counts of its 768 threads
        const int nt = 768, span = g.toKeep + g.F + (nSteps - 1) * g.C;
        if (g.N > 2 * nt || g.C > nt || span - g.N > 3 * nt || span < g.N || g.latency > g.F + (nSteps - 1) * g.C || g.latency < g.N) return VP_OK;
    }
    if (h->poisoned) return poisoned_rc(h);
    if (int frc = check_fault(h)) return frc;
    VpWsMb mb;
    memset(&mb, 0, sizeof mb);
    mb.nBlocks = nb;
    int nSched = 0, schedChunk[WS_MB_SCHEDS];
    int nChunk = co.nChunk, inC = h->inCounter, outC = h->outCounter, curC = h->currCounter;
    for (int b = 0; b < nb; b++) {
        int k = 0;
        while (k < nSched && schedChunk[k] != nChunk) k++;
        if (k == nSched) {
            if (nSched == WS_MB_SCHEDS) return VP_OK;
            if (!ws_build_sched(g, nChunk, nSteps, mb.sc[nSched])) return VP_OK;
            schedChunk[nSched++] = nChunk;
        }
        mb.schedOf[b] = k; mb.nChunk0[b] = nChunk; mb.inCtr[b] = inC; mb.outCtr[b] = outC; mb.currCtr[b] = curC;
        for (int i = 0; i < nSteps; i++) nChunk = (nChunk % g.cpf == g.cpf - 1) ? 1 % g.cpf : nChunk + 1;
        inC = (inC + g.N) % g.inSize; outC = (outC + g.N) % g.outSize; curC = (curC + g.N) % g.inSize;
    }
    h->profThis = h->prof > 0 && (h->profTick++ % (unsigned)h->prof) == 0;
    VpCall c;
    memset(&c, 0, sizeof c);
    c.inCounter = h->inCounter; c.outCounter = h->outCounter; c.currCounter = h->currCounter;
    c.iirFast = fast ? 1 : 0; c.nBlocks = 1; c.pitchOn = 1; c.vocOn = 0;
    c.pStart = 0; c.nChunk0 = co.nChunk; c.nSteps = nSteps;
    c.fuseIngest = 1; c.fuseEmit = 1; c.ldsAcc = 1;
    if (mono) {
        c.inMono = h->synthNonZero > 0 ? 1 : 2;                               // (per launch: zeros are written while the ring may still hold anything else)
        h->synthNonZero = std::max(0, h->synthNonZero - g.N * nb);
    } else {
        c.inMono = 0;
        h->synthNonZero = g.inSize;
    }
    c.yinCert = (h->yinMode == VP_YIN_XCORR || h->yinMode == VP_YIN_FFT) ? 1 : (h->yinMode == VP_YIN_XCORR_FORCE_FALLBACK) ? 2 : 0;
    c.ldsBytes = (int)vp_pitch_ws_lds_bytes(g, nSteps);
    VpDev d = h->d;
    d.streamMap = co.dMap;
    d.outAcc2 = nullptr;
    {
        ProfScope ps(h, st, 2);
        const auto wsk = g.orderPitch > 15 ? (fast ? vp_k_pitch_ws_mb_o24 : vp_k_pitch_ws_x_mb_o24) : (fast ? vp_k_pitch_ws_mb : vp_k_pitch_ws_x_mb);
        hipLaunchKernelGGL(wsk, dim3(co.n), dim3(64 * 12), (size_t)c.ldsBytes, st, g, c, d, mb, d_in, d_out);
    }
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { h->poisoned = true; h->poisonCode = VP_ERR_HIP; return fail_hip(h, e, "kernel launch"); }
    co.nChunk = nChunk;
    h->inCounter = inC; h->outCounter = outC; h->currCounter = curC;
    h->acc2Live = std::max(0, h->acc2Live - nb);
    *done = true;
    return VP_OK;
}

static int process_blocks_device(vp_handle *h, const float *d_in, float *d_out, int n_blocks, void *hip_stream, bool mono)
{
    if (!h || !d_in || !d_out || n_blocks < 1) return VP_ERR_INVALID_ARG;
    if (!h->prepared) return VP_ERR_NOT_PREPARED;
    if (hipSetDevice(h->device) != hipSuccess) return VP_ERR_NO_DEVICE;
    if (h->poisoned) return process_device(h, d_in, d_out, (hipStream_t)hip_stream, 0, 1, mono);       // reports it
    int rc = sync_stream_state(h, (hipStream_t)hip_stream);                   // the plan below depends on the cohorts
    if (rc) { h->poisoned = true; return rc; }
    const bool fast = h->iirMode == VP_IIR_FAST;
    const bool pitchOnly = h->cohorts.size() == 1 && h->cohorts[0].pitchOn && !h->cohorts[0].vocOn;
    // one launch of the serial kernel for all the blocks (state stays on chip between them); above 256 streams the register-light builds
    // exist for the FAST recursion only.  Where the wave-specialised kernel serves the geometry: groups of up to sixteen blocks per launch
    // of vp_k_pitch_ws_mb (round 6, process_ws_blocks), else a launch of the single-block kernel per block (44.8
    // against 53.6 us per block for the one-launch phase kernel at 256 streams, round 5)
    const bool wsBlocks = pitchOnly && pitch_ws_ok(h, fast, 1, (h->g.N + h->g.C - 1) / h->g.C);
    if (wsBlocks && n_blocks > 1 && h->acc2Live == 0) {
        // groups of up to WS_MB_MAX blocks per launch of the wave-specialised kernel; what the plan does not take goes block by block below
        const size_t nIn_ = (size_t)h->g.S * (mono ? 1 : 3) * h->g.N, nOut_ = (size_t)h->g.S * 2 * h->g.N;
        while (n_blocks > 1) {
            const int nb = std::min(n_blocks, WS_MB_MAX);
            bool done = false;
            rc = process_ws_blocks(h, d_in, d_out, nb, (hipStream_t)hip_stream, mono, &done);
            if (rc) return rc;
            if (!done) break;
            d_in += nb * nIn_; d_out += nb * nOut_; n_blocks -= nb;
        }
        if (n_blocks == 0) return VP_OK;
    }
    if (pitchOnly && n_blocks > 1 && !wsBlocks && (!pitch_lite(h, fast) || fast))
        return process_device(h, d_in, d_out, (hipStream_t)hip_stream, 0, n_blocks, mono);
    const size_t nIn = (size_t)h->g.S * (mono ? 1 : 3) * h->g.N, nOut = (size_t)h->g.S * 2 * h->g.N;
    if (!mono && n_blocks > 1 && !wsBlocks) {                  // vocoder-only plan on the batched pipeline: groups of blocks per launch
        int b = 0;
        while (b < n_blocks) {
            const int nb = std::min(n_blocks - b, std::min(V2_MB_MAX, std::max(h->reservedBlocks, 1)));   // groups the reserved scratch holds
            bool done = false;
            rc = process_voc_blocks(h, d_in + b * nIn, d_out + b * nOut, nb, (hipStream_t)hip_stream, &done);
            if (rc) return rc;
            if (!done) rc = process_both_blocks(h, d_in + b * nIn, d_out + b * nOut, nb, (hipStream_t)hip_stream, &done);
            if (rc) return rc;
            if (!done) break;
            b += nb;
        }
        if (b == n_blocks) return VP_OK;
        d_in += b * nIn; d_out += b * nOut; n_blocks -= b;     // (a remainder of one block, or a plan that does not apply)
    }
    for (int b = 0; b < n_blocks; b++) {                       // other plans: block by block
        rc = process_device(h, d_in + b * nIn, d_out + b * nOut, (hipStream_t)hip_stream, 0, 1, mono);
        if (rc) return rc;
    }
    return VP_OK;
}

extern "C" int vp_process_blocks_device(vp_handle *h, const float *d_in, float *d_out, int n_blocks, void *hip_stream)
{
    return process_blocks_device(h, d_in, d_out, n_blocks, hip_stream, false);
}

// voice [n_blocks][S][N] -> [n_blocks][S][2][N]: the mono form of the above
extern "C" int vp_process_blocks_mono_device(vp_handle *h, const float *d_voice, float *d_out, int n_blocks, void *hip_stream)
{
    return process_blocks_device(h, d_voice, d_out, n_blocks, hip_stream, true);
}

// processBlock() on buffers WITHOUT the side-chain bus: voice [S][N] only.  The reference's fillInputBuffers takes null
// side-chain pointers and fills the synth ring with zeros (MyBuffer.cpp:93-102); same here, without reading or (once
// the ring is known to be all zero) writing anything for it.
extern "C" int vp_process_block_mono_device(vp_handle *h, const float *d_voice, float *d_out, void *hip_stream)
{
    if (!h || !d_voice || !d_out) return VP_ERR_INVALID_ARG;
    if (!h->prepared) return VP_ERR_NOT_PREPARED;
    if (hipSetDevice(h->device) != hipSuccess) return VP_ERR_NO_DEVICE;
    return process_device(h, d_voice, d_out, (hipStream_t)hip_stream, 0, 1, true);
}

extern "C" int vp_process_block_mono(vp_handle *h, const float *voice, float *out)
{
    if (!h || !voice || !out) return VP_ERR_INVALID_ARG;
    if (!h->prepared) return VP_ERR_NOT_PREPARED;
    if (hipSetDevice(h->device) != hipSuccess) return VP_ERR_NO_DEVICE;
    const size_t nIn = (size_t)h->g.S * h->g.N, nOut = (size_t)h->g.S * 2 * h->g.N;
    HIPCHK(h, hipMemcpyAsync(h->stageIn, voice, nIn * sizeof(float), hipMemcpyHostToDevice, h->ownStream));
    int rc = process_device(h, h->stageIn, h->stageOut, h->ownStream, 0, 1, true);
    if (rc) return rc;
    HIPCHK(h, hipMemcpyAsync(out, h->stageOut, nOut * sizeof(float), hipMemcpyDeviceToHost, h->ownStream));
    HIPCHK(h, hipStreamSynchronize(h->ownStream));
    return check_fault(h);                                                    // (behind the synchronisation: this call's own launches included)
}

// Host-pointer form of vp_process_blocks_device (offline rendering from a host program without device allocations of
// its own): one upload, n_blocks blocks, one download, one synchronisation.
extern "C" int vp_process_blocks(vp_handle *h, const float *in, float *out, int n_blocks)
{
    if (!h || !in || !out || n_blocks < 1) return VP_ERR_INVALID_ARG;
    if (!h->prepared) return VP_ERR_NOT_PREPARED;
    if (hipSetDevice(h->device) != hipSuccess) return VP_ERR_NO_DEVICE;
    const size_t nIn = (size_t)h->g.S * 3 * h->g.N, nOut = (size_t)h->g.S * 2 * h->g.N;
    // groups of as many blocks as the staging buffers hold (vp_reserve_blocks); none reserved: block by block through the
    // single-block staging of prepare.  No allocation here.
    for (int b = 0; b < n_blocks; ) {
        const int nb = std::min(n_blocks - b, std::max(h->stageBlocks, 1));
        float *sIn = h->stageBlocks ? h->stageInB : h->stageIn, *sOut = h->stageBlocks ? h->stageOutB : h->stageOut;
        HIPCHK(h, hipMemcpyAsync(sIn, in + b * nIn, nIn * nb * sizeof(float), hipMemcpyHostToDevice, h->ownStream));
        int rc = vp_process_blocks_device(h, sIn, sOut, nb, (void *)h->ownStream);
        if (rc) return rc;
        HIPCHK(h, hipMemcpyAsync(out + b * nOut, sOut, nOut * nb * sizeof(float), hipMemcpyDeviceToHost, h->ownStream));
        b += nb;
    }
    HIPCHK(h, hipStreamSynchronize(h->ownStream));
    return check_fault(h);                                                    // (behind the synchronisation: this call's own launches included)
}

// Sizes everything the multi-block entry points need for calls of up to n_blocks blocks.  The ONLY allocation site besides prepare.
extern "C" int vp_reserve_blocks(vp_handle *h, int n_blocks)
{
    if (!h || n_blocks < 1) return VP_ERR_INVALID_ARG;
    if (!h->prepared) return VP_ERR_NOT_PREPARED;
    if (hipSetDevice(h->device) != hipSuccess) return VP_ERR_NO_DEVICE;
    HIPCHK(h, hipDeviceSynchronize());                                        // nothing in flight may still use what is replaced below
    const VpGeom &g = h->g;
    const int nb = std::min(n_blocks, V2_MB_MAX);
    if (h->v2.xT && nb >= 2 && (size_t)g.outSize * sizeof(double) <= (size_t)VP_V2_MB_LDS_MAX) {
        // the lane-per-window pipeline over the windows of nb blocks (vocoder-only plan), and the combined plan's ring snapshots,
        // per-block gates and linear accumulator (both processes, VP_IIR_FAST) -- whatever the switches are now: they may change later
        int rc = ensure_v2mb(h, nb * h->nWinMax);
        if (rc) return rc;
        if ((rc = ensure_both(h)) != VP_OK) return rc;
    }
    if (n_blocks > h->stageBlocks) {                                          // host staging of vp_process_blocks
        const size_t nIn = (size_t)g.S * 3 * g.N, nOut = (size_t)g.S * 2 * g.N;
        if (h->stageInB) (void)hipFree(h->stageInB);
        if (h->stageOutB) (void)hipFree(h->stageOutB);
        h->stageInB = h->stageOutB = nullptr; h->stageBlocks = 0;
        if (vp_malloc(h, (void **)&h->stageInB, nIn * n_blocks * sizeof(float)) != hipSuccess ||
            vp_malloc(h, (void **)&h->stageOutB, nOut * n_blocks * sizeof(float)) != hipSuccess) {
            if (h->stageInB) (void)hipFree(h->stageInB);
            h->stageInB = nullptr;
            h->lastError = "out of device memory for the block staging buffers";
            return VP_ERR_OOM;
        }
        h->stageBlocks = n_blocks;
    }
    h->reservedBlocks = std::max(h->reservedBlocks, n_blocks);
    HIPCHK(h, hipDeviceSynchronize());                                        // the zero fills (null stream) are complete before any launch
    return VP_OK;
}
extern "C" int vp_get_reserved_blocks(const vp_handle *h) { return (h && h->prepared) ? h->reservedBlocks : VP_ERR_NOT_PREPARED; }
extern "C" long vp_debug_alloc_count(const vp_handle *h) { return h ? h->nAllocs : -1; }

extern "C" int vp_process_block(vp_handle *h, const float *in, float *out)
{
    if (!h || !in || !out) return VP_ERR_INVALID_ARG;
    if (!h->prepared) return VP_ERR_NOT_PREPARED;
    if (hipSetDevice(h->device) != hipSuccess) return VP_ERR_NO_DEVICE;
    const size_t nIn = (size_t)h->g.S * 3 * h->g.N, nOut = (size_t)h->g.S * 2 * h->g.N;
    HIPCHK(h, hipMemcpyAsync(h->stageIn, in, nIn * sizeof(float), hipMemcpyHostToDevice, h->ownStream));
    int rc = process_device(h, h->stageIn, h->stageOut, h->ownStream, 0);
    if (rc) return rc;
    HIPCHK(h, hipMemcpyAsync(out, h->stageOut, nOut * sizeof(float), hipMemcpyDeviceToHost, h->ownStream));
    HIPCHK(h, hipStreamSynchronize(h->ownStream));
    return check_fault(h);                                                    // (behind the synchronisation: this call's own launches included)
}

extern "C" int vp_process_block_inplace(vp_handle *h, float *io)
{
    if (!h || !io) return VP_ERR_INVALID_ARG;
    if (!h->prepared) return VP_ERR_NOT_PREPARED;
    if (hipSetDevice(h->device) != hipSuccess) return VP_ERR_NO_DEVICE;
    const size_t n = (size_t)h->g.S * 3 * h->g.N;
    HIPCHK(h, hipMemcpyAsync(h->stageIn, io, n * sizeof(float), hipMemcpyHostToDevice, h->ownStream));
    int rc = process_device(h, h->stageIn, h->stageOut, h->ownStream, 1);
    if (rc) return rc;
    HIPCHK(h, hipMemcpyAsync(io, h->stageOut, n * sizeof(float), hipMemcpyDeviceToHost, h->ownStream));
    HIPCHK(h, hipStreamSynchronize(h->ownStream));
    return check_fault(h);                                                    // (behind the synchronisation: this call's own launches included)
}

extern "C" int vp_get_latency(const vp_handle *h) { return (h && h->prepared) ? h->g.latency : VP_ERR_NOT_PREPARED; }

extern "C" int vp_get_geometry(const vp_handle *h, int out[12])
{
    if (!h || !out) return VP_ERR_INVALID_ARG;
    if (!h->prepared) return VP_ERR_NOT_PREPARED;
    const VpGeom &g = h->g;
    int v[12] = {g.N, g.F, g.H, g.C, g.W, g.h, g.toKeep, g.latency, g.inSize, g.outSize, g.tauMax, g.cpf};
    memcpy(out, v, sizeof v);
    return VP_OK;
}

extern "C" int vp_get_num_streams(const vp_handle *h) { return (h && h->prepared) ? h->g.S : VP_ERR_NOT_PREPARED; }

extern "C" int vp_synchronize(vp_handle *h)
{
    if (!h) return VP_ERR_INVALID_ARG;
    if (hipSetDevice(h->device) != hipSuccess) return VP_ERR_NO_DEVICE;
    HIPCHK(h, hipDeviceSynchronize());
    if (h->poisoned) return poisoned_rc(h);
    return check_fault(h);                                                    // every launch so far has completed: the verdict is final
}

// Diagnostic: the number of polls a kernel's bounded inter-wavefront wait makes before it gives up (default 2^22, about a second).
// tests/test_gpu_round6.py sets it to 1 to force the timeout path: the call that synchronises behind such a launch must return
// VP_ERR_TIMEOUT and leave the handle poisoned.
extern "C" int vp_debug_set_spin_limit(vp_handle *h, int polls)
{
    if (!h || polls < 1) return VP_ERR_INVALID_ARG;
    h->spinLimit = polls;
    h->d.spinLimit = polls;
    return VP_OK;
}

extern "C" int vp_read_pitch_state(vp_handle *h, int stream, vp_pitch_state *out)
{
    if (!h || !out) return VP_ERR_INVALID_ARG;
    if (!h->prepared) return VP_ERR_NOT_PREPARED;
    if (stream < 0 || stream >= h->g.S) return VP_ERR_INVALID_ARG;
    if (hipSetDevice(h->device) != hipSuccess) return VP_ERR_NO_DEVICE;
    HIPCHK(h, hipDeviceSynchronize());
    VpPitchState ps;
    int gate[2];
    HIPCHK(h, hipMemcpy(&ps, h->d.pitch + stream, sizeof ps, hipMemcpyDeviceToHost));
    HIPCHK(h, hipMemcpy(gate, h->d.gate + 2 * stream, sizeof gate, hipMemcpyDeviceToHost));
    memset(out, 0, sizeof *out);
    out->period = ps.period; out->prevPeriod = ps.prevPeriod; out->prevVoicedPeriod = ps.prevVoicedPeriod;
    out->periodNew = ps.periodNew; out->nAn = ps.nAn; out->nSt = ps.nSt; out->stMarkIdx = ps.stMarkIdx;
    out->gateOpen = gate[0];
    out->pitch = ps.pitch; out->prevPitch = ps.prevPitch; out->beta = ps.beta; out->closestFreq = ps.closestFreq;
    memcpy(out->anMarks, ps.anMarks, sizeof out->anMarks);
    memcpy(out->stMarks, ps.stMarks, sizeof out->stMarks);
    memcpy(out->a, ps.a, sizeof out->a);
    return VP_OK;
}

extern "C" int vp_read_ub_counters(vp_handle *h, long out[5])
{
    if (!h || !out) return VP_ERR_INVALID_ARG;
    if (!h->prepared) return VP_ERR_NOT_PREPARED;
    if (hipSetDevice(h->device) != hipSuccess) return VP_ERR_NO_DEVICE;
    HIPCHK(h, hipDeviceSynchronize());
    unsigned long long v[5];
    HIPCHK(h, hipMemcpy(v, h->d.ub, sizeof v, hipMemcpyDeviceToHost));
    for (int i = 0; i < 5; i++) out[i] = (long)v[i];
    return VP_OK;
}

extern "C" int vp_profile_enable(vp_handle *h, int on)
{
    if (!h) return VP_ERR_INVALID_ARG;
    h->prof = on < 0 ? 0 : on;
    h->profTick = 0;
    return VP_OK;
}

extern "C" int vp_profile_read(vp_handle *h, double ms[VP_NUM_KERNEL_SLOTS], long launches[VP_NUM_KERNEL_SLOTS], int reset)
{
    if (!h || !ms || !launches) return VP_ERR_INVALID_ARG;
    if (hipSetDevice(h->device) != hipSuccess) return VP_ERR_NO_DEVICE;
    HIPCHK(h, hipDeviceSynchronize());
    for (auto &p : h->pending) {
        float t = 0.f;
        if (hipEventElapsedTime(&t, p.a, p.b) == hipSuccess) { h->profMs[p.slot] += t; h->profN[p.slot] += 1; }
        h->evPool.push_back(p.a);
        h->evPool.push_back(p.b);
    }
    h->pending.clear();
    for (int i = 0; i < VP_NUM_KERNEL_SLOTS; i++) { ms[i] = h->profMs[i]; launches[i] = h->profN[i]; }
    if (reset) for (int i = 0; i < VP_NUM_KERNEL_SLOTS; i++) { h->profMs[i] = 0; h->profN[i] = 0; }
    return VP_OK;
}

// Diagnostic (-DVP_STAMPS build only): per-phase 100 MHz ticks accumulated by workgroup 0.
// Diagnostic build only: accumulated ticks (100 MHz) each stream's workgroup spent inside the pitch kernel.
extern "C" int vp_debug_read_stream_ticks(vp_handle *h, unsigned long long *out, int n, int reset)
{
    if (!h || !out || n < 1) return VP_ERR_INVALID_ARG;
    if (!h->prepared) return VP_ERR_NOT_PREPARED;
    if (hipSetDevice(h->device) != hipSuccess) return VP_ERR_NO_DEVICE;
    n = std::min(n, h->g.S);
    HIPCHK(h, hipDeviceSynchronize());
    HIPCHK(h, hipMemcpy(out, h->d.dbg + 64, (size_t)n * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    if (reset) HIPCHK(h, hipMemset(h->d.dbg + 64, 0, (size_t)h->g.S * sizeof(unsigned long long)));
    return VP_OK;
}

extern "C" int vp_debug_read_ws_stamps(vp_handle *h, unsigned long long out[512], int reset)
{
    if (!h || !out) return VP_ERR_INVALID_ARG;
    if (!h->prepared) return VP_ERR_NOT_PREPARED;
    if (hipSetDevice(h->device) != hipSuccess) return VP_ERR_NO_DEVICE;
    HIPCHK(h, hipDeviceSynchronize());
    HIPCHK(h, hipMemcpy(out, h->d.dbg + 64 + h->g.S, 512 * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    if (reset) HIPCHK(h, hipMemset(h->d.dbg + 64 + h->g.S, 0, 512 * sizeof(unsigned long long)));
    return VP_OK;
}

extern "C" int vp_debug_read_stamps(vp_handle *h, unsigned long long out[64], int reset)
{
    if (!h || !out) return VP_ERR_INVALID_ARG;
    if (!h->prepared) return VP_ERR_NOT_PREPARED;
    if (hipSetDevice(h->device) != hipSuccess) return VP_ERR_NO_DEVICE;
    HIPCHK(h, hipDeviceSynchronize());
    HIPCHK(h, hipMemcpy(out, h->d.dbg, 64 * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    if (reset) HIPCHK(h, hipMemset(h->d.dbg, 0, 64 * sizeof(unsigned long long)));
    return VP_OK;
}

extern "C" int vp_set_iir_mode(vp_handle *h, int mode)
{
    if (!h || (mode != VP_IIR_EXACT && mode != VP_IIR_FAST)) return VP_ERR_INVALID_ARG;
    h->iirMode = mode;
    return VP_OK;
}
extern "C" int vp_get_iir_mode(const vp_handle *h) { return h ? h->iirMode : VP_ERR_INVALID_ARG; }

extern "C" int vp_set_yin_mode(vp_handle *h, int mode)
{
    if (!h || mode < VP_YIN_DIRECT || mode > VP_YIN_XCORR_FORCE_FALLBACK) return VP_ERR_INVALID_ARG;
    h->yinMode = mode;
    return VP_OK;
}
extern "C" int vp_get_yin_mode(const vp_handle *h) { return h ? h->yinMode : VP_ERR_INVALID_ARG; }

// ---- standalone STFT round trip (no reference counterpart): the fused kernel of vp_stft.hip -------------------------------------
// (one frame per wavefront, register FFT, overlap-add in LDS; 1024-point frames, optionally with the phase-vocoder stage, and 2048-point frames)
struct vp_stft {
    int device, F, hop, S, T, nFrames;
    double *win = nullptr, *tw1 = nullptr, *tw2 = nullptr, *tws = nullptr, *twTop = nullptr;
    float scale;
    bool pvOk = false;                // the phase-vocoder build's dynamic-LDS ceiling could be raised on this device (vp_stft_pitch_shift needs it)
    int runsPerStream = 0;            // 0: chosen from the batch so that the grid fills the chip; > 0: vp_stft_set_runs (tests)
    int f32 = 0;                      // vp_stft_set_precision
};

static void stft_free(vp_stft *p)
{
    (void)hipFree(p->win); (void)hipFree(p->tw1); (void)hipFree(p->tw2); (void)hipFree(p->tws); (void)hipFree(p->twTop);
}

extern "C" int vp_stft_create(int device, int n_streams, int n_samples, int frame_len, int hop, vp_stft **out)
{
    if (!out || n_streams <= 0 || frame_len < 8 || hop <= 0 || n_samples < frame_len) return VP_ERR_INVALID_ARG;
    // (hop == frame_len: the sqrt-Hann windows do not overlap and w[0]^2 = 0 cannot be normalised; at least two frames must cover every
    // sample.  Frame lengths other than 1024 and 2048 -- eight / sixteen complex points per lane of one wavefront -- are not built.)
    if (!vp_stft_supported(frame_len, hop)) return VP_ERR_GEOMETRY;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) return VP_ERR_NO_DEVICE;
    if (hipSetDevice(device) != hipSuccess) return VP_ERR_NO_DEVICE;
    vp_stft *p = new vp_stft();
    // The phase-vocoder build needs more than the default 64 KB of dynamic LDS: its ceiling is raised per handle and device
    // (hipFuncSetAttribute acts on the current device).  Only vp_stft_pitch_shift needs it, so a failure is remembered and fails
    // THAT call; the plain and single-precision round trips (<= 64 KB) are served regardless.  The sticky HIP error is cleared so
    // that a later hipGetLastError() behind a launch does not report this one.
    p->pvOk = vp_stft_prepare_device() == hipSuccess;
    if (!p->pvOk) (void)hipGetLastError();
    p->device = device; p->F = frame_len; p->hop = hop; p->S = n_streams; p->T = n_samples;
    p->nFrames = (n_samples - frame_len) / hop + 1;
    const double PI = 3.141592653589793238;
    std::vector<double> w(frame_len);
    double sumsq = 0;                                             // sum over one hop grid of w^2 (constant for periodic Hann)
    for (int i = 0; i < frame_len; i++) w[i] = std::sqrt(0.5 - 0.5 * std::cos(2.0 * PI * i / frame_len));
    for (int i = 0; i < frame_len; i += hop) sumsq += w[i] * w[i];
    p->scale = (float)(1.0 / sumsq);
    auto up = [](double **d, const std::vector<double> &v) {
        return hipMalloc(d, v.size() * 8) == hipSuccess && hipMemcpy(*d, v.data(), v.size() * 8, hipMemcpyHostToDevice) == hipSuccess;
    };
    // per-lane twiddles of the wavefront transform (vp_fft.inc fft512), of the real-input split (NP = F / 256 bin pairs per lane:
    // W_F^(64 q + lane)) and, for 2048-point frames, of the radix-2 step on top of two 512-point transforms (W_1024^(64 q + lane)), host libm
    const int NP = frame_len / 256;
    std::vector<double> t1(64 * 8 * 2), t2(64 * 8 * 2), ts(64 * NP * 2), tt(64 * 8 * 2);
    for (int lane = 0; lane < 64; lane++) {
        for (int r = 0; r < 8; r++) {
            const double a1 = -2.0 * PI * (double)(r * (lane >> 3)) / 64.0, a2 = -2.0 * PI * (double)(r * lane) / 512.0;
            const double a3 = -2.0 * PI * (double)(64 * r + lane) / 1024.0;
            t1[(lane * 8 + r) * 2] = std::cos(a1); t1[(lane * 8 + r) * 2 + 1] = std::sin(a1);
            t2[(lane * 8 + r) * 2] = std::cos(a2); t2[(lane * 8 + r) * 2 + 1] = std::sin(a2);
            tt[(lane * 8 + r) * 2] = std::cos(a3); tt[(lane * 8 + r) * 2 + 1] = std::sin(a3);
        }
        for (int q = 0; q < NP; q++) {
            const double a = -2.0 * PI * (double)(64 * q + lane) / (double)frame_len;
            ts[(lane * NP + q) * 2] = std::cos(a); ts[(lane * NP + q) * 2 + 1] = std::sin(a);
        }
    }
    if (!(up(&p->win, w) && up(&p->tw1, t1) && up(&p->tw2, t2) && up(&p->tws, ts) && (frame_len != 2048 || up(&p->twTop, tt)))) { stft_free(p); delete p; return VP_ERR_OOM; }
    *out = p;
    return VP_OK;
}

extern "C" int vp_stft_destroy(vp_stft *p)
{
    if (!p) return VP_ERR_INVALID_ARG;
    (void)hipSetDevice(p->device);
    (void)hipDeviceSynchronize();
    stft_free(p);
    delete p;
    return VP_OK;
}

extern "C" int vp_stft_num_frames(const vp_stft *p) { return p ? p->nFrames : VP_ERR_INVALID_ARG; }
extern "C" int vp_stft_is_fused(const vp_stft *p) { return p ? 1 : VP_ERR_INVALID_ARG; }
extern "C" int vp_stft_set_runs(vp_stft *p, int runs_per_stream)
{
    if (!p || runs_per_stream < 0) return VP_ERR_INVALID_ARG;
    p->runsPerStream = runs_per_stream;
    return VP_OK;
}

extern "C" int vp_stft_set_precision(vp_stft *p, int precision)
{
    if (!p || (precision != VP_STFT_F64 && precision != VP_STFT_F32)) return VP_ERR_INVALID_ARG;
    p->f32 = precision == VP_STFT_F32;
    return VP_OK;
}
extern "C" int vp_stft_get_precision(const vp_stft *p) { return p ? (p->f32 ? VP_STFT_F32 : VP_STFT_F64) : VP_ERR_INVALID_ARG; }

static int stft_fused(vp_stft *p, const float *d_in, float *d_out, float *d_mag, hipStream_t st, bool pv, double ratio)
{
    VpStftArgs a;
    memset(&a, 0, sizeof a);
    a.in = d_in; a.out = d_out; a.mag = d_mag; a.win = p->win; a.tw1 = p->tw1; a.tw2 = p->tw2; a.tws = p->tws; a.twTop = p->twTop;
    a.pvRatio = ratio; a.pv = pv ? 1 : 0;
    a.f32 = (p->f32 && !pv) ? 1 : 0;                           // (the phase-vocoder stage keeps double: its phases accumulate over the stream)
    a.c = (double)p->scale / (double)(p->F / 2);
    a.T = p->T; a.nFrames = p->nFrames; a.F = p->F; a.hop = p->hop; a.O = p->F / p->hop;
    a.nHops = (p->T + p->hop - 1) / p->hop;
    a.nRounds = (a.nHops + VP_STFT_WAVES - 1) / VP_STFT_WAVES;
    a.haloRounds = (a.O - 1 + VP_STFT_WAVES - 1) / VP_STFT_WAVES;
    // (float2 loads per lane at 1024 points, float4 at 2048: rows and hops aligned to the load's width)
    const int al = p->F == 2048 ? 4 : 2;
    a.aligned = (p->T % al == 0 && p->hop % al == 0 && ((uintptr_t)d_in & (4 * al - 1)) == 0) ? 1 : 0;
    // runs: enough workgroups for two per CU (the kernel's register budget), but runs of at least four rounds per recomputed one;
    // the phase-vocoder stage carries a recurrence over the frames of a stream: one run
    int runs = 1;
    if (!pv) {
        // (four workgroups per CU in the single-precision build -- half the registers --, three at 2048 points)
        runs = p->runsPerStream > 0 ? p->runsPerStream : ((a.f32 ? (p->F == 2048 ? 3 : 4) : 2) * 256 + p->S - 1) / p->S;
        runs = std::max(1, std::min(runs, a.nRounds / (4 * a.haloRounds)));
    }
    a.roundsPerRun = (a.nRounds + runs - 1) / runs;
    const int nRuns = (a.nRounds + a.roundsPerRun - 1) / a.roundsPerRun;
    return vp_stft_launch(a, p->S, nRuns, st) == hipSuccess ? VP_OK : VP_ERR_HIP;
}

// d_in, d_out: device float32 [S][T]; d_mag: optional device float32 [S][nFrames][F/2+1] (or NULL)
extern "C" int vp_stft_roundtrip(vp_stft *p, const float *d_in, float *d_out, float *d_mag, void *hip_stream)
{
    if (!p || !d_in || !d_out) return VP_ERR_INVALID_ARG;
    if (hipSetDevice(p->device) != hipSuccess) return VP_ERR_NO_DEVICE;
    return stft_fused(p, d_in, d_out, d_mag, (hipStream_t)hip_stream, false, 1.0);
}

// the same round trip with the phase-vocoder pitch shift between the transforms (fused kernel only)
extern "C" int vp_stft_pitch_shift(vp_stft *p, const float *d_in, float *d_out, double semitones, void *hip_stream)
{
    if (!p || !d_in || !d_out || !(semitones >= -12.0 && semitones <= 12.0)) return VP_ERR_INVALID_ARG;
    if (p->F != 1024) return VP_ERR_GEOMETRY;                  // the phase-vocoder stage is built for 1024-point frames
    if (!p->pvOk) return VP_ERR_HIP;                           // (its dynamic-LDS ceiling could not be raised on this device: vp_stft_create)
    if (hipSetDevice(p->device) != hipSuccess) return VP_ERR_NO_DEVICE;
    return stft_fused(p, d_in, d_out, nullptr, (hipStream_t)hip_stream, true, std::pow(2.0, semitones / 12.0));
}
