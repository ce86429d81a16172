// vp_kernels.h -- kernel prototypes shared by vp_kernels.hip and vp_capi.hip
#pragma once
#include <hip/hip_runtime.h>

#include "vp_common.h"

__global__ void vp_k_ingest_gate(VpGeom g, VpCall c, VpDev d, const float *__restrict__ in);
__global__ void vp_k_vocoder(VpGeom g, VpCall c, VpDev d, const float *__restrict__ in, float *__restrict__ out);
__global__ void vp_k_vocoder_o48(VpGeom g, VpCall c, VpDev d, const float *__restrict__ in, float *__restrict__ out);
__global__ void vp_k_vocoder_lite(VpGeom g, VpCall c, VpDev d, const float *__restrict__ in, float *__restrict__ out);
__global__ void vp_k_pitch(VpGeom g, VpCall c, VpDev d, const float *__restrict__ in, float *__restrict__ out);
__global__ void vp_k_pitch_lite(VpGeom g, VpCall c, VpDev d, const float *__restrict__ in, float *__restrict__ out);
__global__ void vp_k_pitch_fast(VpGeom g, VpCall c, VpDev d, const float *__restrict__ in, float *__restrict__ out);
__global__ void vp_k_pitch_lite_fast(VpGeom g, VpCall c, VpDev d, const float *__restrict__ in, float *__restrict__ out);
__global__ void vp_k_pitch_multi(VpGeom g, VpCall c, VpDev d, const float *__restrict__ in, float *__restrict__ out);
__global__ void vp_k_pitch_fast_multi(VpGeom g, VpCall c, VpDev d, const float *__restrict__ in, float *__restrict__ out);
__global__ void vp_k_pitch_c(VpGeom g, VpCall c, VpDev d, const float *__restrict__ in, float *__restrict__ out);
__global__ void vp_k_pitch_fast_c(VpGeom g, VpCall c, VpDev d, const float *__restrict__ in, float *__restrict__ out);
__global__ void vp_k_pitch_fast_multi_c(VpGeom g, VpCall c, VpDev d, const float *__restrict__ in, float *__restrict__ out);
__global__ void vp_k_pitch_lite_fast_c(VpGeom g, VpCall c, VpDev d, const float *__restrict__ in, float *__restrict__ out);
__global__ void vp_k_pitch_lite_fast_multi(VpGeom g, VpCall c, VpDev d, const float *__restrict__ in, float *__restrict__ out);
__global__ void vp_k_pitch_lite_fast_multi_c(VpGeom g, VpCall c, VpDev d, const float *__restrict__ in, float *__restrict__ out);
__global__ void vp_k_pitch_ws(VpGeom g, VpCall c, VpDev d, VpWsSched sc, const float *__restrict__ in, float *__restrict__ out);
__global__ void vp_k_pitch_ws_x(VpGeom g, VpCall c, VpDev d, VpWsSched sc, const float *__restrict__ in, float *__restrict__ out);
__global__ void vp_k_pitch_ws_mb(VpGeom g, VpCall c, VpDev d, VpWsMb mb, const float *__restrict__ in, float *__restrict__ out);
__global__ void vp_k_pitch_ws_x_mb(VpGeom g, VpCall c, VpDev d, VpWsMb mb, const float *__restrict__ in, float *__restrict__ out);
__global__ void vp_k_pitch_ws_mb_o24(VpGeom g, VpCall c, VpDev d, VpWsMb mb, const float *__restrict__ in, float *__restrict__ out);
__global__ void vp_k_pitch_ws_x_mb_o24(VpGeom g, VpCall c, VpDev d, VpWsMb mb, const float *__restrict__ in, float *__restrict__ out);
__global__ void vp_k_pitch_ws_o24(VpGeom g, VpCall c, VpDev d, VpWsSched sc, const float *__restrict__ in, float *__restrict__ out);
__global__ void vp_k_pitch_ws_x_o24(VpGeom g, VpCall c, VpDev d, VpWsSched sc, const float *__restrict__ in, float *__restrict__ out);
__global__ void vp_k_emit(VpGeom g, VpCall c, VpDev d, float *__restrict__ out);


// extra dynamic LDS of VP_YIN_FFT (certified cross-correlations by FFT): a 64-byte flag block and an 8 KB exchange buffer per transforming wavefront
// and the LDS copies of the twiddle tables ([8][8] + [64][8] + [64][4] complex doubles)
static inline size_t vp_pitch_fft_lds_bytes(int waves) { return 64 + (size_t)VP_FFT_TW_D2 * 16 + (size_t)waves * 8192; }
static inline size_t vp_voc_lds_bytes(int W, int nWaves)
{
    return (VP_VOC_SHARED_DOUBLES(W) + (size_t)nWaves * voc_wave_doubles(W)) * sizeof(double);
}
