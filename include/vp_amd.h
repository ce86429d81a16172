/*
 * vp_amd.h -- C ABI of libvp_amd.so: the MI355X (gfx950) batch implementation of the
 * DamRsn/VocoderProject DSP hot path (PitchProcess pitch corrector + LPC/VocoderProcess
 * cross-synthesis behind VocoderAudioProcessor::processBlock()).
 *
 * One handle = a BATCH of S independent plugin instances (one per audio stream) living on one
 * GPU.  Every entry point is what the reference's host code would bind through FFI for this
 * path; the reference interface each one replaces is cited as file:line relative to
 * /root/reference/Source/.  Plain pointers and sizes only; no C++/torch types.
 *
 * Threading: one caller thread per handle (the reference runs processBlock() on one audio
 * thread per instance, PluginProcessor.cpp:203).  Parameters are snapshotted at call entry
 * (the reference reads std::atomic<float> values with load(), :214-230).
 * No allocation happens in any vp_process_*() call (reference: all vectors are sized in
 * prepareToPlay, PitchProcess.cpp:95-121): vp_prepare_*() and vp_reserve_blocks() are the only
 * entry points that allocate device memory (vp_debug_alloc_count() counts the allocations).
 */
#ifndef VP_AMD_H
#define VP_AMD_H

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define VP_ABI_VERSION 3        /* 3 (round 6): + VP_ERR_TIMEOUT, vp_debug_set_spin_limit (additions only: a version-2 caller keeps working); 2 (round 5): + vp_set_wave_specialised, vp_debug_read_ws_stamps; round 4 had added vp_reserve_blocks and the
                                   vp_stft_* precision / pitch-shift entries and narrowed vp_stft_create to 1024- and 2048-point frames */

/* status codes (the reference has none: it prints to std::cerr and assert(false)s) */
enum {
    VP_OK = 0,
    VP_ERR_INVALID_ARG = -1,     /* null pointer, non-positive size, parameter out of its range (PluginProcessor.cpp:41-69) */
    VP_ERR_NOT_PREPARED = -2,    /* processBlock before prepareToPlay */
    VP_ERR_INVALID_OVERLAP = -3, /* VocoderProcess.cpp:110-114 "Invalid overlap": (wlen-hop)/wlen must be 0.5 or 0.75 */
    VP_ERR_GEOMETRY = -4,        /* frameLen % (frameLen-hop) != 0 (PitchProcess.cpp:91-92), tauMax > samplesToKeep (:364), ... */
    VP_ERR_ORDER = -5,           /* LPC order above its parameter maximum (VocoderProcess.cpp:145-148,161-164) */
    VP_ERR_NO_DEVICE = -6,       /* no usable HIP device: this library has NO CPU fallback */
    VP_ERR_HIP = -7,             /* a HIP runtime call failed; see vp_last_error() */
    VP_ERR_OOM = -8,
    VP_ERR_TIMEOUT = -9          /* a kernel's bounded inter-wavefront wait ran out (a lost signal, or a wavefront held up for about a second
                                    by a debugger / preemption): that launch's output and the handle's device state are INVALID.  The handle
                                    is poisoned -- every later process call returns this code -- until it is prepared again.  Reported by the
                                    call that synchronises behind the launch (vp_process_block*, vp_process_blocks with host pointers,
                                    vp_synchronize) or, on the device-pointer entry points, by the next call on the handle; the reference's
                                    counterpart is the assert(false) on impossible state, PitchProcess.cpp:824,828 */
};

/* The ten plugin parameters, same ids, ranges and defaults as
 * VocoderAudioProcessor::createParameterLayout() (PluginProcessor.cpp:37-73).
 * vp_set_params() gives every stream of the handle this set; vp_set_stream_params() gives ONE stream its own (each stream
 * of a batch is a plugin instance with its own treeState), all ten parameters except lpcPitch. */
typedef struct vp_params {
    float gainPitch;   /* dB  [-60, 6]   default   0 */
    float gainVoice;   /* dB  [-60, 6]   default -60 (dry voice off) */
    float gainSynth;   /* dB  [-60, 6]   default -60 (dry carrier off) */
    float gainVoc;     /* dB  [-60, 6]   default   0 */
    int lpcVoice;      /* [2, 100] default 40; re-read every vocoder window (VocoderProcess.cpp:193-194) */
    int lpcPitch;      /* [2, 100] default 15; read ONLY at prepare (PitchProcess.cpp:70) */
    int lpcSynth;      /* [2, 30]  default  5 */
    int keyPitch;      /* 0..11 = major key on A, A#, ... G#; 12 = chromatic (default) (Notes.h:18) */
    int pitchBool;     /* 1: pitch corrector on (default) */
    int vocBool;       /* 1: vocoder on (default) */
} vp_params;

/* Per-stream pitch-tracker state, for tracing/parity tests (the reference keeps these as private
 * members, PitchProcess.h:108-131). */
#define VP_MARK_CAP 64
typedef struct vp_pitch_state {
    int period, prevPeriod, prevVoicedPeriod, periodNew;
    int nAn, nSt, stMarkIdx, gateOpen;
    double pitch, prevPitch, beta, closestFreq;
    int anMarks[VP_MARK_CAP];
    int stMarks[VP_MARK_CAP];
    double a[101];
} vp_pitch_state;

typedef struct vp_handle vp_handle;

/* createPluginFilter() (PluginProcessor.cpp:272-275): a batch of plugin instances on HIP device
 * `device`.  Fails with VP_ERR_NO_DEVICE when no GPU is present. */
int vp_create(int device, vp_handle **out);
int vp_destroy(vp_handle *h);

/* treeState.getRawParameterValue(id)->store(v) for all ten ids at once (PluginProcessor.cpp:37-73).
 * May be called between blocks at any time; lpcPitch only takes effect at the next prepare. */
int vp_set_params(vp_handle *h, const vp_params *p);
int vp_get_params(const vp_handle *h, vp_params *p);
/* One parameter set PER STREAM (each stream of a batch is its own plugin instance with its own treeState).
 * After prepare; takes effect at the next block.  Everything is per stream, pitchBool and vocBool included: a process that
 * is switched off does not advance its startSample / nChunk (PluginProcessor.cpp:214-221), so the library keeps streams
 * whose switches differ(ed) as separate cohorts and launches each with a stream-index map.  Only lpcPitch stays per handle
 * (it is read at prepare and selects the pitch kernel's build): `p` must repeat the handle's value for it
 * (VP_ERR_INVALID_ARG otherwise).  A later vp_set_params() puts every stream back on one common set; so does a new prepare. */
int vp_set_stream_params(vp_handle *h, int stream, const vp_params *p);
int vp_get_stream_params(const vp_handle *h, int stream, vp_params *p);
void vp_default_params(vp_params *p);

/* EXTENSION (no reference counterpart; BASELINE configs[1] "+-12-semitone pitch shift"): a fixed interval instead of
 * the correction to the key's nearest note.  placeStMarks (PitchProcess.cpp:593-598) then takes
 * beta = 2^(semitones/12) (host libm) in place of closestFreq/pitch; everything else on the path is unchanged.
 * stream = -1 sets every stream.  After prepare; takes effect at the next analysed frame; a new prepare switches it
 * off.  |semitones| <= 12 (VP_ERR_INVALID_ARG); VP_ERR_GEOMETRY when an upward shift would need more synthesis marks
 * per frame than VP_MARK_CAP (F / round(floor(fs/fMax) / beta) + 2). */
int vp_set_pitch_shift(vp_handle *h, int stream, int on, double semitones);
int vp_get_pitch_shift(const vp_handle *h, int stream, int *on, double *semitones);

/* VocoderAudioProcessor::prepareToPlay(sampleRate, samplesPerBlock) (PluginProcessor.cpp:144-184)
 * for n_streams instances: derives the vocoder/pitch geometry from the sample rate (:159-170),
 * allocates and zeroes all per-stream state in HBM. */
int vp_prepare_to_play(vp_handle *h, double sample_rate, int samples_per_block, int n_streams);

/* Same with explicit geometry: pitchProcess.prepare(fs,100,800,F,H,N,-60),
 * vocoderProcess.prepare(W,h,"sine",-60), myBuffer.prepare(N,F,max(F,W),fs,1,2,2)
 * (PluginProcessor.cpp:172-181; both prepare()s are public: PitchProcess.h:40, VocoderProcess.h:29). */
int vp_prepare_explicit(vp_handle *h, double sample_rate, int samples_per_block, int n_streams,
                        int frame_len_pitch, int hop_pitch, int wlen_voc, int hop_voc);

/* processBlock(AudioBuffer<float>&, MidiBuffer&) (PluginProcessor.cpp:203-234) for every stream.
 * Host buffers, planar float32:  in  [n_streams][3][N]  (ch0 voice, ch1/ch2 side-chain L/R),
 *                                out [n_streams][2][N]  (L, R).
 * Synchronous: returns when `out` is filled. */
int vp_process_block(vp_handle *h, const float *in, float *out);

/* The reference's exact in-place form: io [n_streams][3][N]; on return ch0/ch1 hold out L/R and
 * ch2 is zero (MyBuffer.cpp:115 buffer.clear()). Host buffer, synchronous. */
int vp_process_block_inplace(vp_handle *h, float *io);

/* Device-resident form (the multi-GPU / throughput path): d_in, d_out are device pointers with
 * the layouts above; kernels are enqueued on `hip_stream` (a hipStream_t, may be NULL = default
 * stream) and the call returns without synchronising. */
int vp_process_block_device(vp_handle *h, const float *d_in, float *d_out, void *hip_stream);
/* processBlock() on buffers without the side-chain bus (mono voice in, stereo out): voice float [n_streams][N].
 * MyBuffer::fillInputBuffers takes null side-chain pointers and fills the synth ring with zeros (MyBuffer.cpp:93-102);
 * results are identical to vp_process_block with zeroed ch1/ch2, at a third of the input traffic.  May be mixed freely
 * with the three-channel calls. */
int vp_process_block_mono(vp_handle *h, const float *voice, float *out);
int vp_process_block_mono_device(vp_handle *h, const float *d_voice, float *d_out, void *hip_stream);
/* n_blocks of them at once: d_voice float [n_blocks][n_streams][N] (the mono form of vp_process_blocks_device below) */
int vp_process_blocks_mono_device(vp_handle *h, const float *d_voice, float *d_out, int n_blocks, void *hip_stream);
/* n_blocks consecutive processBlock() calls at once (offline rendering, servers with audio queued up):
 * d_in float [n_blocks][n_streams][3][N], d_out float [n_blocks][n_streams][2][N], i.e. block b's slabs are what
 * vp_process_block_device would take.  Parameters are read once, at entry.  What runs:
 *  - pitch corrector alone: the blocks run in ONE launch (tracker state and the frame in flight stay on chip between them);
 *    results identical to n_blocks single calls;
 *  - vocoder alone (LPC orders <= 48): groups of up to 16 blocks as ONE launch of the lane-per-window pipeline -- the window grid
 *    carries across blocks, so B blocks are B times the windows, i.e. B times the lanes; every block keeps its own ring ingest,
 *    silence gate and output slab.  In VP_IIR_EXACT always (the pipeline and the workgroup kernel give the same bits: results
 *    identical to single calls; 256 streams, 8 blocks per call: 3x the single-call throughput); in VP_IIR_FAST only where
 *    single-block calls take the pipeline too (VP_VOC_AUTO above 256 streams, or VP_VOC_BATCHED) -- the two implementations'
 *    tolerance-mode roundings differ, and the output must not depend on how the caller groups blocks;
 *  - both enabled, VP_IIR_FAST, and single-block calls on the pipeline (same condition): groups of up to 16 blocks as one launch of
 *    the serial pitch kernel followed by one launch of the pipeline (the pitch kernel ingests the blocks and adds its chunks into a
 *    linear accumulator of the call, the pipeline works from a snapshot of the rings and folds that accumulator in when it emits):
 *    every decision and every filter output is the block-by-block path's; the audio equals it to the rounding of the additions
 *    into the output accumulator, which happen chunks-first instead of windows-first (<= 2e-6 of full scale, tested);
 *  - anything else (VP_IIR_EXACT with both enabled, per-stream switches that split the batch, ...): block by block.
 * The two pipeline plans are only taken where they pay (round 6, measured): a batch whose single block already fills the chip
 * (8192 windows and more per block: e.g. 1024 streams at 512 / 128) runs the vocoder-only plan never and the combined plan from
 * groups of eight blocks on -- as does a batch whose single-block calls overlap the pitch kernel with the pipeline's tail
 * (vp_set_overlap) --; smaller batches take both from two blocks on.  Same results either way (the plans' arithmetic is the
 * single-block pipeline's).
 * The two pipeline plans need scratch for the call's windows: vp_reserve_blocks(h, n) sizes it for calls of up to n blocks; a
 * larger call is carried out in groups of n, and without a reservation these plans fall back to block by block.  Nothing is
 * allocated here. */
int vp_process_blocks_device(vp_handle *h, const float *d_in, float *d_out, int n_blocks, void *hip_stream);
/* The same from HOST memory: in float [n_blocks][n_streams][3][N], out float [n_blocks][n_streams][2][N]; upload, the blocks,
 * download, in groups of as many blocks as vp_reserve_blocks sized the staging buffers for (none reserved: block by block through
 * the single-block staging buffers of prepare); synchronises before returning. */
int vp_process_blocks(vp_handle *h, const float *in, float *out, int n_blocks);
/* Allocation for the multi-block entry points, outside the process calls: scratch of the lane-per-window pipeline for the windows of
 * min(n_blocks, 16) blocks, the combined plan's ring snapshots / per-block gates / linear accumulator, and host staging for
 * n_blocks blocks.  After prepare (a new prepare drops the reservation); grows only; synchronises the device.  VP_ERR_OOM.
 * (The reference sizes everything in prepareToPlay, PitchProcess.cpp:95-121; how many blocks a caller queues per call is not a
 * prepareToPlay argument, hence the separate entry point.) */
int vp_reserve_blocks(vp_handle *h, int n_blocks);
int vp_get_reserved_blocks(const vp_handle *h);
/* Device allocations (hipMalloc calls) made for this handle since vp_create: constant across any sequence of vp_process_*() calls. */
long vp_debug_alloc_count(const vp_handle *h);

/* Arithmetic of the two all-pole synthesis filters (VocoderProcess.cpp:277-286, PitchProcess.cpp:307-322).
 * VP_IIR_EXACT (default): the reference's summation order, output bit-identical to the CPU restatement.
 * VP_IIR_FAST: transposed-form recursion with the taps spread over the lanes; differs from EXACT by
 * rounding only (measured max |diff| ~1e-12 of full scale before the float32 cast); every decision of
 * the algorithm (pitch, marks, gate) is still bit-identical because none depends on a filter output. */
#define VP_IIR_EXACT 0
#define VP_IIR_FAST 1
int vp_set_iir_mode(vp_handle *h, int mode);
int vp_get_iir_mode(const vp_handle *h);

/* Which implementation of the vocoder (VocoderProcess::process) a block runs.  Same results (bit-identical in
 * VP_IIR_EXACT mode), different mapping onto the GPU:
 * VP_VOC_WORKGROUP: one workgroup per stream, one wavefront per window (vp_k_vocoder / vp_k_vocoder_lite);
 * VP_VOC_BATCHED: a pipeline of small kernels in which one LANE owns one window (vp_voc2.hip) -- pays when a block of the
 *   batch has thousands of windows; LPC orders up to 48, at most 64 windows per stream and block, else the call falls back;
 * VP_VOC_AUTO (default): batched for batches of more than 256 streams (one workgroup per CU no longer holds them) with at
 *   least 1024 windows per block. */
#define VP_VOC_AUTO 0
#define VP_VOC_WORKGROUP 1
#define VP_VOC_BATCHED 2
int vp_set_vocoder_path(vp_handle *h, int path);
int vp_get_vocoder_path(const vp_handle *h);

/* VP_IIR_FAST only, both processes on, batched vocoder: run the pitch corrector BESIDE the tail of the vocoder pipeline (its all-pole
 * recursion and overlap-add: register-light kernels that fit into what the pitch kernel leaves of a SIMD's register file; second
 * HIP stream, accumulator of its own, merged at emit) instead of behind it.  0 off, 1 on, VP_OVERLAP_AUTO (the default): on where it
 * pays -- the full-register pitch builds (frames too large for two workgroups per CU: 419 -> 376 us per block at the configs[4]
 * geometry), off for the register-light ones (1024 streams of the plugin's geometry: 276 against 274 us).  What is given up is the
 * order in which the two processes' contributions are added into the output accumulator (PluginProcessor.cpp:214-221), i.e.
 * rounding; the exact mode never does this.  A caller's hip_stream is respected: the work it sees is ordered on that stream. */
#define VP_OVERLAP_AUTO 2
int vp_set_overlap(vp_handle *h, int on);
int vp_get_overlap(const vp_handle *h);
/* SURVEY 8(f2), queued audio.  What serves it (round 6): vp_process_blocks[_mono]_device hands the pitch corrector's blocks to ONE launch
 * of vp_k_pitch_ws_mb (exact arithmetic: _x_mb; lpcPitch 16 .. 24: _mb_o24 / _x_mb_o24) per group of up to sixteen (state, frame in flight, voice window and accumulator slice stay on chip between the
 * blocks; 27 M frames/s against 24 M block by block at 256 streams) -- no switch, same bits.  The earlier attempt, a time-parallel
 * analysis front end in front of the serial kernel (vp_k_pitch_front, rounds 3-5), never paid -- its cross-correlations were the same
 * vector work on the same CUs -- and was removed; the two entry points below are kept so that a version-2 caller links: the value is
 * stored and returned, and changes nothing. */
int vp_set_time_parallel(vp_handle *h, int on);
int vp_get_time_parallel(const vp_handle *h);
/* Round 5: which pitch-corrector kernel serves single-block calls of the plugin's own geometry (1024-sample frames, lpcPitch <= 15,
 * up to 256 streams).  1 (default): vp_k_pitch_ws / vp_k_pitch_ws_x, the WAVE-SPECIALISED kernel (csrc/vp_pitch_ws.inc: every frame
 * start's yin() / LPC / marks on four background wavefronts beside the running frame's chunks, PitchProcess.cpp:203-271).  0: the
 * phase kernels (vp_k_pitch_fast_c / vp_k_pitch_c), which also serve every other geometry.  Same output bits either way
 * (tests/test_gpu_round5.py); the switch exists for that test and for same-box A/B timing (tools/ws_ab.py). */
int vp_set_wave_specialised(vp_handle *h, int on);
int vp_get_wave_specialised(const vp_handle *h);

/* How the YIN difference function (PitchProcess.cpp:350-403) and the pitch frame's LPC autocorrelation
 * (LPC.cpp:44-97) are evaluated.
 * VP_YIN_DIRECT (default): the reference's O(F tau) sums in its own order; decisions bit-identical.
 * VP_YIN_FFT: Wiener-Khinchin accelerator (one forward + one inverse radix-2 FFT of >= F + tauMax points in
 * LDS).  Values differ by ~1e-13 relative, so a threshold decision CAN differ on a near-tie; the measured
 * rate of frames whose period differs is reported by tests/test_gpu_parity.py.
 * VP_YIN_XCORR: the difference function as energies minus a cross-correlation (fused multiply-adds: a third of the
 * arithmetic), CERTIFIED: its values differ from the reference's by at most 2^-39 of the window energy, every
 * comparison the pitch decision rests on is checked against what that can do to it, and a frame with a comparison
 * too close to call is recomputed in the reference's arithmetic.  Decisions -- and therefore the output -- are
 * bit-identical to VP_YIN_DIRECT by construction; only the LPC autocorrelation stays as it is. */
#define VP_YIN_DIRECT 0
#define VP_YIN_FFT 1
#define VP_YIN_XCORR 2
#define VP_YIN_XCORR_FORCE_FALLBACK 3   /* diagnostic: as XCORR, but every frame is treated as "too close to call" */
int vp_set_yin_mode(vp_handle *h, int mode);
int vp_get_yin_mode(const vp_handle *h);

/* AudioProcessor::getLatencySamples() after setLatencySamples(max(F, W)) (PluginProcessor.cpp:175,183). */
int vp_get_latency(const vp_handle *h);
/* N, F, H, C, W, h, samplesToKeep, latency, inSize, outSize, tauMax, chunksPerFrame
 * (MyBuffer.h:37-43 getters + PitchProcess/VocoderProcess geometry). */
int vp_get_geometry(const vp_handle *h, int out[12]);
int vp_get_num_streams(const vp_handle *h);

/* Copies one stream's pitch-tracker state to the host (synchronises). */
int vp_read_pitch_state(vp_handle *h, int stream, vp_pitch_state *out);

/* hipDeviceSynchronize on the handle's device, then the verdict on everything launched so far: VP_OK, or the code that poisoned the
 * handle (VP_ERR_TIMEOUT / VP_ERR_HIP).  A host that drives the device-pointer entry points calls this before it trusts their output. */
int vp_synchronize(vp_handle *h);
/* Diagnostic: polls a kernel's bounded inter-wavefront wait makes before it gives up (default 2^22, about a second).  The test
 * suite sets 1 to force the timeout path (tests/test_gpu_round6.py). */
int vp_debug_set_spin_limit(vp_handle *h, int polls);

/* Kernel timing with HIP events on the launch stream (bench.py's roofline figures).
 * vp_profile_enable(h, k) brackets the kernel launches of every k-th process call with events (k = 1: every call;
 * the records cost the stream 2-3 us per bracketed launch, so a throughput measurement samples); vp_profile_read returns, per
 * kernel slot, the accumulated milliseconds and launch count since the last reset.
 * Slots: 0 ingest+gate, 1 vocoder, 2 pitch, 3 emit. */
#define VP_NUM_KERNEL_SLOTS 4
int vp_profile_enable(vp_handle *h, int on);
int vp_profile_read(vp_handle *h, double ms[VP_NUM_KERNEL_SLOTS], long launches[VP_NUM_KERNEL_SLOTS], int reset);
const char *vp_kernel_slot_name(int slot);
/* Symbol of the pitch-kernel build the handle's current geometry and modes select (slot 2 is one of
 * vp_k_pitch[_fast][_c], vp_k_pitch_lite[_fast], vp_k_pitch[_fast]_fft; _c = the common-case builds); "" before prepare. */
const char *vp_pitch_kernel_name(const vp_handle *h);
/* Likewise for slot 1: vp_k_vocoder (LPC orders up to 32 and above 48) / vp_k_vocoder_o48 (orders 33..48), vp_k_vocoder_lite
 * (FAST IIR, more than 256 streams on the workgroup path: two workgroups per CU), or "vp_k_v2_pipeline" -- the lane-per-window
 * pipeline of kernels (vp_k_v2_ingest_stage, _autocorr, _levinson2, _fir2, _energy[_slices], _iir_exact / _iir_fast, _ola) that
 * VP_VOC_AUTO picks above 256 streams with at least 1024 windows per block. */
const char *vp_vocoder_kernel_name(const vp_handle *h);

/* Counts, over all streams since prepare, how often a kernel reached one of the reference's
 * undefined-behaviour sites (back() of an empty vector, yinTemp[tauMax]):
 * same meaning as the oracle's counters. Synchronises. */
int vp_read_ub_counters(vp_handle *h, long out[5]);

/* Slots 0..58: diagnostic build (-DVP_STAMPS) only, per-phase timers (100 MHz ticks) of workgroup 0, all zero in
 * the product build.  Slots 59 / 60 / 61 (every build): wavefronts whose bounded wait for an in-workgroup flag (YIN prefix
 * sums / LPC coefficients / grain table) ran out -- always 0; anything else has also raised VP_ERR_TIMEOUT and poisoned the handle.  Slots 62 / 63 (every build): frames, over all streams, whose pitch decision VP_YIN_XCORR
 * certified / handed to the reference's arithmetic. */
int vp_debug_read_stamps(vp_handle *h, unsigned long long out[64], int reset);
/* Diagnostic build only: per-wavefront timers of the wave-specialised pitch kernel, [4 block types][16 wavefronts][8 slots] (tools/ws_stamps.py). */
int vp_debug_read_ws_stamps(vp_handle *h, unsigned long long out[512], int reset);
/* Diagnostic build only: ticks each of the first n streams' workgroups spent inside the pitch kernel (which stream paces a launch). */
int vp_debug_read_stream_ticks(vp_handle *h, unsigned long long *out, int n, int reset);

/* Standalone STFT round trip: sqrt-Hann window, batched FFT, [spectral stage], inverse FFT, overlap-add, in ONE fused kernel
 * (csrc/vp_stft.hip: one frame per wavefront, the transform's butterflies in registers, overlap-add in LDS, every input sample
 * read from HBM once and every output sample written once).  frame_len must be 1024 or 2048 (eight / sixteen complex points
 * per lane of a wavefront; VP_ERR_GEOMETRY otherwise), hop a divisor of it with 2 <= frame_len / hop <= 16.  NO reference counterpart (the
 * reference contains no FFT): these are the STFT-shaped kernels BASELINE.json's north_star lists, reported
 * on their own by bench.py and checked against numpy.fft / a build-authored NumPy restatement (tests/stft_reference.py: parity
 * unpinned by nature).
 * d_in/d_out: device float32 [n_streams][n_samples]; d_mag (optional): [n_streams][frames][frame_len/2+1]. */
typedef struct vp_stft vp_stft;
int vp_stft_create(int device, int n_streams, int n_samples, int frame_len, int hop, vp_stft **out);
int vp_stft_destroy(vp_stft *p);
int vp_stft_num_frames(const vp_stft *p);
int vp_stft_roundtrip(vp_stft *p, const float *d_in, float *d_out, float *d_mag, void *hip_stream);
/* The north_star's "per-bin phase unwrap/accumulate" stage between the two transforms: the classic phase-vocoder pitch shift by
 * `semitones` in [-12, 12] (per frame and bin: magnitude and phase; phase advance against the previous frame minus the bin's
 * nominal advance, wrapped to (-pi, pi] -> true frequency; bins move to floor(k ratio + 0.5), magnitudes that land together
 * add; the synthesis phase accumulates the scaled advance).  Each call starts from a zero phase state.  1024-point frames only
 * (VP_ERR_GEOMETRY otherwise).  No reference counterpart; checked against tests/stft_reference.py. */
int vp_stft_pitch_shift(vp_stft *p, const float *d_in, float *d_out, double semitones, void *hip_stream);
int vp_stft_is_fused(const vp_stft *p);                      /* 1 (every handle runs the fused kernel; kept for older callers) */
/* Diagnostic: cut every stream into this many runs of frames (one workgroup each) instead of choosing from the batch size
 * (0 = automatic).  The output does not depend on it (tests). */
int vp_stft_set_runs(vp_stft *p, int runs_per_stream);
/* Arithmetic of vp_stft_roundtrip's transforms.  VP_STFT_F64 (default): double, as everything else in this library.  VP_STFT_F32: the
 * transform, split and merge in single precision (vp_k_stft_fused32: an f32 vector instruction issues in half the cycles of an fp64 one,
 * the kernel needs half the registers and half the LDS bytes) -- input and output are float32 either way; the result differs from the
 * default's by rounding (~2e-7 of the signal's scale; the north_star's bound is 1e-4 RMS).  Both frame lengths
 * (vp_k_stft_fused32, vp_k_stft_fused2k32); vp_stft_pitch_shift always runs in double (its phases accumulate over the whole stream). */
#define VP_STFT_F64 0
#define VP_STFT_F32 1
int vp_stft_set_precision(vp_stft *p, int precision);
int vp_stft_get_precision(const vp_stft *p);

const char *vp_error_string(int code);
const char *vp_last_error(const vp_handle *h);
int vp_abi_version(void);

#ifdef __cplusplus
}
#endif
#endif /* VP_AMD_H */
