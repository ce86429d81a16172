// vp_voc2.h -- interface between vp_capi.hip and the batched lane-per-window vocoder pipeline (vp_voc2.hip)
#pragma once
#include <hip/hip_runtime.h>

#include "vp_common.h"

#define V2_RV_STRIDE 104        // doubles per window for r / a of the voice (orders <= VP_ORDER_MAX)
#define V2_RS_STRIDE 32         // ... of the side chain (orders <= VP_ORDER_MAX_SYNTH)
#define V2_FIR_SLICE 64         // outputs of a window per wavefront of the residual kernel
#define V2_ORDER_MAX 48         // largest LPC order the register-resident kernels are instantiated for

// scratch of the pipeline, per handle.  NW = streams x windows per block; window w = (cohort-local stream) x nWin + j.
struct VpV2 {
    float *xT;                  // [2][nGroupsMax][W4p][64][4]  the windows' samples, transposed tiles (see vp_voc2.hip)
    double *rV, *rS;            // [nGroupsMax][V2_RV_STRIDE][64], [nGroupsMax][V2_RS_STRIDE][64]   autocorrelations (lag-major tiles)
    double *aV, *aS;            // same shapes: A(z) coefficients
    int4 *meta;                 // [NW] per window: gate open, lpcVoice, lpcSynth, stream
    int *rank;                  // [NW] 1-based rank of the window among the stream's LIVE windows of this launch (0: gated)
    int *liveList;              // [NW] per stream: window index of its q-th live window
    float *dry;                 // [S][3][latency] multi-block launches: the samples of the dry paths that only the ring held
    double *eT;                 // [2][nGroupsMax][W2p][64][2]  residuals (voice, side chain), transposed tiles
    double *out;                // [NW][W] the all-pole output, window-major
    int nGroupsMax, W4p, W2p;   // 64-window groups the scratch holds; float4 / double2 rows per window (padded)
    double *EE;                 // [NW][2] residual energies (voice, side chain)
    double *EEp;                // [NW][2][nSlices] the same per 64-output slice of the residual kernel
    int nSlices;
    int nStreams;               // streams of this launch (the cohort)
    int oVmax, oSmax;           // largest lpcVoice / lpcSynth over the streams (the orders themselves are per stream)
};

// multi-block launches (vp_process_blocks_device, vocoder-only plan): per block of the launch
#define V2_MB_MAX 16
#define VP_V2_MB_LDS_MAX (160 * 1024 - 1024)      // dynamic-LDS ceiling vp_v2_init() sets on vp_k_v2_mb_ola_emit (it holds outSize doubles)
// preIngested: the blocks are in the ring already and their gates in VpDev::gateB (combined plan: the pitch kernel ran first); the
// samples then come from the ring SNAPSHOT the caller put into VpDev::voiceRing / synthRing
struct VpV2MB { int nBlocks, vStart[V2_MB_MAX], nWin[V2_MB_MAX], first[V2_MB_MAX], preIngested; };

int vp_v2_init();
void vp_v2_launch_blocks(const VpGeom &g, const VpCall &c, const VpDev &d, const VpV2 &v, const VpV2MB &mb, const float *d_in, float *d_out,
                         hipStream_t st);
// `beforeIir` (optional) is called right behind the launch of the residual kernel, in front of the recursion's (what the caller wants to start beside
// the pipeline's register-light tail: vp_k_v2_iir_fast, vp_k_v2_ola)
void vp_v2_launch(const VpGeom &g, const VpCall &c, const VpDev &d, const VpV2 &v, const float *d_in, float *d_out, hipStream_t st,
                  void (*beforeIir)(void *) = nullptr, void *hookArg = nullptr);
