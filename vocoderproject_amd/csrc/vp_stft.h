// vp_stft.h -- the fused STFT round trip (csrc/vp_stft.hip): arguments and launcher.
// No reference counterpart (the reference has no FFT: SURVEY.md section 0); BASELINE.json's north_star names these kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>

#define VP_STFT_WAVES 4                 // wavefronts per workgroup = frames per round

struct VpStftArgs {
    const float *in;                    // [S][T]
    float *out;                         // [S][T]
    float *mag;                         // [S][nFrames][F/2 + 1] or nullptr
    const double *win;                  // [F] sqrt-Hann (periodic), analysis = synthesis
    const double *tw1;                  // [64][8][2]  W_64^(m0 (lane >> 3))            second step of the 512-point transform
    const double *tw2;                  // [64][8][2]  W_512^(a lane)                    third step
    const double *tws;                  // [64][NP][2] W_(2F)^... real-input split / merge: W_F^(64 q + lane), NP = F / 256 pairs per lane
    const double *twTop;                // [64][8][2]  W_1024^(64 q + lane): the radix-2 step on top of two 512-point transforms (F = 2048), else nullptr
    double pvRatio;                     // pitch ratio of the phase-vocoder stage
    double c;                           // scale / (F/2): overlap-add normalisation and the inverse transform's 1/N, folded into the merge
    int T, nFrames, nHops, nRounds;     // samples per stream; frames; hops of output (ceil(T / hop)); rounds of VP_STFT_WAVES frames
    int roundsPerRun, haloRounds;       // a workgroup owns roundsPerRun rounds of one stream (+ haloRounds recomputed in front)
    int F, hop, O;                      // frame, hop, overlap factor F / hop
    int aligned;                        // rows and hops 8-byte aligned: float2 loads
    int pv;                             // 1: phase-vocoder stage between the transforms (one workgroup per stream), 0: identity
    int f32;                            // 1: transform, split and merge in single precision (vp_k_stft_fused32 / vp_k_stft_fused2k32; no phase-vocoder stage)
};

size_t vp_stft_lds_bytes(int F, int hop, int f32 = 0);
// enqueues the fused kernel (grid = runs x streams); returns hipGetLastError()
hipError_t vp_stft_launch(const VpStftArgs &a, int nStreams, int nRuns, hipStream_t st);
int vp_stft_supported(int F, int hop);
// once per handle, on the handle's device (current device): the dynamic-LDS ceiling of the builds that need more than 64 KB
hipError_t vp_stft_prepare_device();
