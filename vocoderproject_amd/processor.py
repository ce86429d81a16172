"""Host-side mirror of the reference's plugin surface over the C ABI (include/vp_amd.h).

`BatchVocoderProcessor` plays the role of `VocoderAudioProcessor` (PluginProcessor.h:24-80) for a
batch of independent streams on one MI355X: the same ten parameters (PluginProcessor.cpp:37-73),
`prepareToPlay(sampleRate, samplesPerBlock)` (:144) and `processBlock(buffer)` (:203) on planar
float32 buffers `[stream][channel][sample]`.  Everything below the class is ctypes plumbing; the
compute lives in libvp_amd.so (HIP kernels).  There is no CPU fallback: if the shared library is
missing or no GPU is present the constructor raises.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("VP_AMD_LIB") or os.path.join(_HERE, "libvp_amd.so")   # VP_AMD_LIB: diagnostic builds only

PARAM_IDS = ("gainPitch", "gainVoice", "gainSynth", "gainVoc", "lpcVoice", "lpcPitch",
             "lpcSynth", "keyPitch", "pitchBool", "vocBool")
KEYS = ("A", "A#", "B", "C", "C#", "D", "D#", "E", "F", "F#", "G", "G#", "Chrom")   # PluginProcessor.cpp:64
GEOM_KEYS = ("N", "F", "H", "C", "W", "h", "toKeep", "latency", "inSize", "outSize", "tauMax", "chunksPerFrame")
KERNEL_SLOTS = 4
MARK_CAP = 64


class VpParams(C.Structure):
    _fields_ = [("gainPitch", C.c_float), ("gainVoice", C.c_float), ("gainSynth", C.c_float), ("gainVoc", C.c_float),
                ("lpcVoice", C.c_int), ("lpcPitch", C.c_int), ("lpcSynth", C.c_int), ("keyPitch", C.c_int),
                ("pitchBool", C.c_int), ("vocBool", C.c_int)]


class VpPitchState(C.Structure):
    _fields_ = [("period", C.c_int), ("prevPeriod", C.c_int), ("prevVoicedPeriod", C.c_int), ("periodNew", C.c_int),
                ("nAn", C.c_int), ("nSt", C.c_int), ("stMarkIdx", C.c_int), ("gateOpen", C.c_int),
                ("pitch", C.c_double), ("prevPitch", C.c_double), ("beta", C.c_double), ("closestFreq", C.c_double),
                ("anMarks", C.c_int * MARK_CAP), ("stMarks", C.c_int * MARK_CAP), ("a", C.c_double * 101)]


class VpError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"libvp_amd error {code}: {msg}")
        self.code = code


_lib = None


def load_library():
    """dlopen libvp_amd.so (built by vocoderproject_amd.build). Raises if it is missing."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise FileNotFoundError(f"{LIB_PATH} not built: run `python -m vocoderproject_amd.build` "
                                "(needs hipcc); this package has no CPU fallback")
    L = C.CDLL(LIB_PATH)
    vp = C.c_void_p
    fp = C.c_void_p      # float* passed as integer addresses (host numpy or device data_ptr)
    L.vp_abi_version.restype = C.c_int
    L.vp_create.argtypes = [C.c_int, C.POINTER(vp)]
    L.vp_destroy.argtypes = [vp]
    L.vp_set_params.argtypes = [vp, C.POINTER(VpParams)]
    L.vp_get_params.argtypes = [vp, C.POINTER(VpParams)]
    L.vp_default_params.argtypes = [C.POINTER(VpParams)]
    L.vp_default_params.restype = None
    L.vp_prepare_to_play.argtypes = [vp, C.c_double, C.c_int, C.c_int]
    L.vp_prepare_explicit.argtypes = [vp, C.c_double, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]
    L.vp_process_block.argtypes = [vp, fp, fp]
    L.vp_process_block_inplace.argtypes = [vp, fp]
    L.vp_process_block_device.argtypes = [vp, fp, fp, C.c_void_p]
    L.vp_get_latency.argtypes = [vp]
    L.vp_get_geometry.argtypes = [vp, C.POINTER(C.c_int)]
    L.vp_get_num_streams.argtypes = [vp]
    L.vp_read_pitch_state.argtypes = [vp, C.c_int, C.POINTER(VpPitchState)]
    L.vp_synchronize.argtypes = [vp]
    L.vp_profile_enable.argtypes = [vp, C.c_int]
    L.vp_profile_read.argtypes = [vp, C.POINTER(C.c_double), C.POINTER(C.c_long), C.c_int]
    L.vp_kernel_slot_name.argtypes = [C.c_int]
    L.vp_kernel_slot_name.restype = C.c_char_p
    if hasattr(L, "vp_process_blocks_device"):
        L.vp_process_blocks_device.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]
    if hasattr(L, "vp_set_stream_params"):            # absent only from older builds loaded through VP_AMD_LIB (tools/ab.sh)
        L.vp_set_stream_params.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
        L.vp_get_stream_params.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
    if hasattr(L, "vp_process_block_mono"):
        L.vp_process_block_mono.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        L.vp_process_block_mono_device.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.vp_process_blocks_mono_device.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]
    if hasattr(L, "vp_process_blocks"):
        L.vp_process_blocks.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
    if hasattr(L, "vp_set_pitch_shift"):
        L.vp_set_pitch_shift.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_double]
        L.vp_get_pitch_shift.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_double)]
    L.vp_pitch_kernel_name.argtypes = [C.c_void_p]
    L.vp_pitch_kernel_name.restype = C.c_char_p
    if hasattr(L, "vp_vocoder_kernel_name"):
        L.vp_vocoder_kernel_name.argtypes = [C.c_void_p]
        L.vp_vocoder_kernel_name.restype = C.c_char_p
    if hasattr(L, "vp_set_vocoder_path"):
        L.vp_set_vocoder_path.argtypes = [C.c_void_p, C.c_int]
        L.vp_get_vocoder_path.argtypes = [C.c_void_p]
    if hasattr(L, "vp_set_overlap"):
        L.vp_set_overlap.argtypes = [C.c_void_p, C.c_int]
        L.vp_get_overlap.argtypes = [C.c_void_p]
    if hasattr(L, "vp_reserve_blocks"):                  # (an older A/B library handed over through VP_AMD_LIB still loads)
        L.vp_reserve_blocks.argtypes = [C.c_void_p, C.c_int]
        L.vp_get_reserved_blocks.argtypes = [C.c_void_p]
        L.vp_debug_alloc_count.argtypes = [C.c_void_p]
        L.vp_debug_alloc_count.restype = C.c_long
    if hasattr(L, "vp_set_time_parallel"):
        L.vp_set_time_parallel.argtypes = [C.c_void_p, C.c_int]
        L.vp_get_time_parallel.argtypes = [C.c_void_p]
    if hasattr(L, "vp_set_wave_specialised"):
        L.vp_set_wave_specialised.argtypes = [C.c_void_p, C.c_int]
        L.vp_get_wave_specialised.argtypes = [C.c_void_p]
    if hasattr(L, "vp_debug_set_spin_limit"):          # (ABI version 3)
        L.vp_debug_set_spin_limit.argtypes = [C.c_void_p, C.c_int]
    L.vp_read_ub_counters.argtypes = [vp, C.POINTER(C.c_long)]
    L.vp_debug_read_stamps.argtypes = [vp, C.POINTER(C.c_ulonglong), C.c_int]
    L.vp_set_yin_mode.argtypes = [vp, C.c_int]
    L.vp_get_yin_mode.argtypes = [vp]
    L.vp_set_iir_mode.argtypes = [vp, C.c_int]
    L.vp_get_iir_mode.argtypes = [vp]
    L.vp_stft_create.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(vp)]
    L.vp_stft_destroy.argtypes = [vp]
    L.vp_stft_num_frames.argtypes = [vp]
    L.vp_stft_roundtrip.argtypes = [vp, fp, fp, fp, C.c_void_p]
    L.vp_stft_pitch_shift.argtypes = [vp, fp, fp, C.c_double, C.c_void_p]
    L.vp_stft_is_fused.argtypes = [vp]
    L.vp_stft_set_runs.argtypes = [vp, C.c_int]
    L.vp_stft_set_precision.argtypes = [vp, C.c_int]
    L.vp_stft_get_precision.argtypes = [vp]
    L.vp_error_string.argtypes = [C.c_int]
    L.vp_error_string.restype = C.c_char_p
    L.vp_last_error.argtypes = [vp]
    L.vp_last_error.restype = C.c_char_p
    _lib = L
    return L


class BatchVocoderProcessor:
    """A batch of `VocoderAudioProcessor` instances on one GPU (one per stream)."""

    def __init__(self, device=0, **params):
        self.L = load_library()
        h = C.c_void_p()
        rc = self.L.vp_create(int(device), C.byref(h))
        if rc:
            raise VpError(rc, self.L.vp_error_string(rc).decode())
        self.h = h
        self.device = device
        self._p = VpParams()
        self.L.vp_default_params(C.byref(self._p))
        self.n_streams = 0
        self.N = 0
        for k, v in params.items():
            self.setParameter(k, v)

    # ---- lifetime ------------------------------------------------------------------------------
    def close(self):
        if getattr(self, "h", None):
            self.L.vp_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _chk(self, rc):
        if rc:
            raise VpError(rc, f"{self.L.vp_error_string(rc).decode()} ({self.L.vp_last_error(self.h).decode()})")

    # ---- parameters (treeState) --------------------------------------------------------------------
    def setParameter(self, pid, value):
        if pid not in PARAM_IDS:
            raise KeyError(pid)
        old = getattr(self._p, pid)
        setattr(self._p, pid, type(old)(value))
        rc = self.L.vp_set_params(self.h, C.byref(self._p))
        if rc:
            setattr(self._p, pid, old)
            self._chk(rc)

    def getParameter(self, pid):
        return getattr(self._p, pid)

    def setStreamParameter(self, stream, pid, value):
        """One stream's own value of a parameter (each stream is a plugin instance with its own treeState).  After
        prepare; pitchBool, vocBool and lpcPitch stay per handle.  setParameter() puts all streams back on one set."""
        if pid not in PARAM_IDS:
            raise KeyError(pid)
        q = VpParams()
        self._chk(self.L.vp_get_stream_params(self.h, int(stream), C.byref(q)))
        setattr(q, pid, type(getattr(q, pid))(value))
        self._chk(self.L.vp_set_stream_params(self.h, int(stream), C.byref(q)))

    def setPitchShift(self, semitones, on=True, stream=-1):
        """Extension (no reference counterpart): shift by a fixed interval of +-12 semitones instead of correcting to
        the key's nearest note; stream = -1 sets every stream of the batch."""
        self._chk(self.L.vp_set_pitch_shift(self.h, int(stream), int(bool(on)), float(semitones)))

    def getPitchShift(self, stream):
        on, semi = C.c_int(), C.c_double()
        self._chk(self.L.vp_get_pitch_shift(self.h, int(stream), C.byref(on), C.byref(semi)))
        return bool(on.value), semi.value

    def getStreamParameter(self, stream, pid):
        q = VpParams()
        self._chk(self.L.vp_get_stream_params(self.h, int(stream), C.byref(q)))
        return getattr(q, pid)

    def set_iir_mode(self, mode):
        """"exact" (default, bit-identical to the reference's summation order) or "fast" (VP_IIR_FAST)."""
        self._chk(self.L.vp_set_iir_mode(self.h, {"exact": 0, "fast": 1}[mode] if isinstance(mode, str) else int(mode)))

    def set_vocoder_path(self, path):
        """"auto" (default: batched above 256 streams), "workgroup" (one workgroup per stream) or
        "batched" (the lane-per-window pipeline wherever it can run)."""
        self._chk(self.L.vp_set_vocoder_path(self.h, {"auto": 0, "workgroup": 1, "batched": 2}[path] if isinstance(path, str) else int(path)))

    def set_overlap(self, on):
        """FAST mode, both processes, batched vocoder: pitch corrector beside the vocoder pipeline (default) or behind it."""
        self._chk(self.L.vp_set_overlap(self.h, 2 if on == "auto" else int(bool(on))))

    def set_time_parallel(self, on):
        """Kept for older callers: stored and returned, without effect (the analysis front end it selected was removed in round 6)."""
        self._chk(self.L.vp_set_time_parallel(self.h, int(bool(on))))

    def set_wave_specialised(self, on):
        """Single-block calls of the plugin's geometry on the wave-specialised pitch kernel (default) or on the phase kernels (same bits)."""
        self._chk(self.L.vp_set_wave_specialised(self.h, int(bool(on))))

    def set_yin_mode(self, mode):
        """"direct" (default: the reference's sums), "xcorr" (certified cross-correlation form, fused multiply-adds) or "fft" (the same
        certified form with its cross-correlations by FFT where the build carries it); all three give the same output bits."""
        self._chk(self.L.vp_set_yin_mode(self.h, {"direct": 0, "fft": 1, "xcorr": 2, "xcorr_force_fallback": 3}[mode] if isinstance(mode, str) else int(mode)))

    def get_yin_mode(self):
        return {0: "direct", 1: "fft", 2: "xcorr", 3: "xcorr_force_fallback"}[self.L.vp_get_yin_mode(self.h)]

    def get_iir_mode(self):
        return "fast" if self.L.vp_get_iir_mode(self.h) == 1 else "exact"

    # ---- prepareToPlay -----------------------------------------------------------------------------
    def prepareToPlay(self, sampleRate, samplesPerBlock, nStreams=1):
        self._chk(self.L.vp_prepare_to_play(self.h, float(sampleRate), int(samplesPerBlock), int(nStreams)))
        self.n_streams, self.N = int(nStreams), int(samplesPerBlock)

    def prepareExplicit(self, sampleRate, samplesPerBlock, nStreams, frameLenPitch, hopPitch, wlenVoc, hopVoc):
        self._chk(self.L.vp_prepare_explicit(self.h, float(sampleRate), int(samplesPerBlock), int(nStreams),
                                             int(frameLenPitch), int(hopPitch), int(wlenVoc), int(hopVoc)))
        self.n_streams, self.N = int(nStreams), int(samplesPerBlock)

    def getLatencySamples(self):
        rc = self.L.vp_get_latency(self.h)
        if rc < 0:
            self._chk(rc)
        return rc

    @property
    def latency(self):
        return self.getLatencySamples()

    def geometry(self):
        g = (C.c_int * 12)()
        self._chk(self.L.vp_get_geometry(self.h, g))
        return dict(zip(GEOM_KEYS, list(g)))

    # ---- processBlock ---------------------------------------------------------------------------------
    def processBlock(self, buffer):
        """In place, like the reference: float32 numpy [S][3][N]; on return ch0/ch1 = out L/R, ch2 = 0."""
        assert isinstance(buffer, np.ndarray) and buffer.dtype == np.float32 and buffer.flags.c_contiguous
        assert buffer.shape == (self.n_streams, 3, self.N), buffer.shape
        self._chk(self.L.vp_process_block_inplace(self.h, buffer.ctypes.data))

    def process(self, x):
        """Host convenience: float32 numpy [S][3][N] -> new float32 [S][2][N]."""
        assert x.dtype == np.float32 and x.flags.c_contiguous and x.shape == (self.n_streams, 3, self.N)
        out = np.empty((self.n_streams, 2, self.N), np.float32)
        self._chk(self.L.vp_process_block(self.h, x.ctypes.data, out.ctypes.data))
        return out

    def process_mono(self, voice):
        """Buffers without the side-chain bus: float32 numpy [S][N] -> new float32 [S][2][N] (== process() with zeroed ch1/ch2)."""
        assert voice.dtype == np.float32 and voice.flags.c_contiguous and voice.shape == (self.n_streams, self.N)
        out = np.empty((self.n_streams, 2, self.N), np.float32)
        self._chk(self.L.vp_process_block_mono(self.h, voice.ctypes.data, out.ctypes.data))
        return out

    def process_mono_device(self, d_voice, d_out, stream=None):
        """Device-resident mono entry: torch float32 tensors [S][N] -> [S][2][N]."""
        assert d_voice.is_cuda and d_out.is_cuda and d_voice.is_contiguous() and d_out.is_contiguous()
        assert tuple(d_voice.shape) == (self.n_streams, self.N) and tuple(d_out.shape) == (self.n_streams, 2, self.N)
        if stream is None:
            import torch
            stream = torch.cuda.current_stream(d_voice.device).cuda_stream
        self._chk(self.L.vp_process_block_mono_device(self.h, d_voice.data_ptr(), d_out.data_ptr(), C.c_void_p(stream)))

    def process_blocks_mono_device(self, d_voice, d_out, stream=None):
        """B consecutive mono blocks at once: torch float32 [B][S][N] -> [B][S][2][N]."""
        assert d_voice.is_cuda and d_out.is_cuda and d_voice.is_contiguous() and d_out.is_contiguous()
        B = d_voice.shape[0]
        assert tuple(d_voice.shape) == (B, self.n_streams, self.N) and tuple(d_out.shape) == (B, self.n_streams, 2, self.N)
        if stream is None:
            import torch
            stream = torch.cuda.current_stream(d_voice.device).cuda_stream
        self.reserve_blocks(B)
        self._chk(self.L.vp_process_blocks_mono_device(self.h, d_voice.data_ptr(), d_out.data_ptr(), int(B), C.c_void_p(stream)))

    def reserve_blocks(self, n_blocks):
        """vp_reserve_blocks: size the multi-block scratch and staging for calls of up to n_blocks blocks (the C process calls never
        allocate; the process_blocks* methods of this mirror call it for the caller when a call is larger than what is reserved)."""
        if int(n_blocks) > self.L.vp_get_reserved_blocks(self.h):
            self._chk(self.L.vp_reserve_blocks(self.h, int(n_blocks)))

    def alloc_count(self):
        return int(self.L.vp_debug_alloc_count(self.h))

    def process_device(self, d_in, d_out, stream=None):
        """Device-resident, asynchronous: torch CUDA(HIP) float32 tensors [S][3][N] -> [S][2][N]."""
        assert d_in.is_cuda and d_out.is_cuda and d_in.is_contiguous() and d_out.is_contiguous()
        assert tuple(d_in.shape) == (self.n_streams, 3, self.N) and tuple(d_out.shape) == (self.n_streams, 2, self.N)
        if stream is None:
            import torch
            stream = torch.cuda.current_stream(d_in.device).cuda_stream
        self._chk(self.L.vp_process_block_device(self.h, d_in.data_ptr(), d_out.data_ptr(), C.c_void_p(stream)))

    def process_blocks_device(self, d_in, d_out, stream=None):
        """B consecutive blocks at once: torch float32 tensors [B][S][3][N] -> [B][S][2][N]; same results as B calls of
        process_device (pitch corrector alone: ONE launch, state stays on chip between the blocks; vocoder alone, or both in the
        fast IIR mode: groups of up to 16 blocks per launch of the pipeline -- see include/vp_amd.h)."""
        assert d_in.is_cuda and d_out.is_cuda and d_in.is_contiguous() and d_out.is_contiguous()
        B = d_in.shape[0]
        assert tuple(d_in.shape) == (B, self.n_streams, 3, self.N) and tuple(d_out.shape) == (B, self.n_streams, 2, self.N)
        if stream is None:
            import torch
            stream = torch.cuda.current_stream(d_in.device).cuda_stream
        self.reserve_blocks(B)
        self._chk(self.L.vp_process_blocks_device(self.h, d_in.data_ptr(), d_out.data_ptr(), int(B), C.c_void_p(stream)))

    def process_blocks(self, x):
        """vp_process_blocks: float32 numpy [B][S][3][N] -> new float32 [B][S][2][N], same results as B calls of
        process() (pitch corrector alone: one launch; used by the offline front end)."""
        assert x.dtype == np.float32 and x.flags.c_contiguous and x.ndim == 4 and x.shape[1:] == (self.n_streams, 3, self.N), x.shape
        out = np.empty((x.shape[0], self.n_streams, 2, self.N), np.float32)
        self.reserve_blocks(x.shape[0])
        self._chk(self.L.vp_process_blocks(self.h, x.ctypes.data, out.ctypes.data, int(x.shape[0])))
        return out

    def run(self, x):
        """x: float32 numpy [S][3][T], T a multiple of N -> float32 [S][2][T] (block by block)."""
        S, _, T = x.shape
        assert S == self.n_streams and T % self.N == 0
        out = np.empty((S, 2, T), np.float32)
        for b in range(T // self.N):
            blk = np.ascontiguousarray(x[:, :, b * self.N:(b + 1) * self.N])
            out[:, :, b * self.N:(b + 1) * self.N] = self.process(blk)
        return out

    # ---- introspection ------------------------------------------------------------------------------------
    def synchronize(self):
        self._chk(self.L.vp_synchronize(self.h))

    def pitch_state(self, stream):
        st = VpPitchState()
        self._chk(self.L.vp_read_pitch_state(self.h, int(stream), C.byref(st)))
        return dict(period=st.period, prevPeriod=st.prevPeriod, prevVoicedPeriod=st.prevVoicedPeriod,
                    periodNew=st.periodNew, pitch=st.pitch, prevPitch=st.prevPitch, beta=st.beta,
                    closestFreq=st.closestFreq, gateOpen=st.gateOpen, stMarkIdx=st.stMarkIdx,
                    anMarks=list(st.anMarks[:st.nAn]), stMarks=list(st.stMarks[:st.nSt]), a=np.array(st.a[:]))

    def ub_counters(self):
        c = (C.c_long * 5)()
        self._chk(self.L.vp_read_ub_counters(self.h, c))
        return list(c)

    def debug_stamps(self, reset=True):
        """Diagnostic build only: per-phase microseconds accumulated by workgroup 0."""
        v = (C.c_ulonglong * 64)()
        self._chk(self.L.vp_debug_read_stamps(self.h, v, int(bool(reset))))
        return [t / 100.0 for t in v]

    def yin_certified_counts(self, reset=True):
        """(frames whose pitch decision the certified cross-correlation form settled, frames it handed to the reference's
        arithmetic) since the last reset, over all streams; both 0 outside VP_YIN_XCORR."""
        v = (C.c_ulonglong * 64)()
        self._chk(self.L.vp_debug_read_stamps(self.h, v, int(bool(reset))))
        assert int(v[61]) == 0, "the prefix-sum flag of VP_YIN_XCORR timed out (kernel bug)"
        return int(v[62]), int(v[63])

    def profile_enable(self, on=True):
        """True/1: HIP events around every kernel launch; k > 1: around every k-th one; False/0: off."""
        self._chk(self.L.vp_profile_enable(self.h, int(on)))

    def profile_read(self, reset=True):
        ms = (C.c_double * KERNEL_SLOTS)()
        n = (C.c_long * KERNEL_SLOTS)()
        self._chk(self.L.vp_profile_read(self.h, ms, n, int(bool(reset))))
        names = [self.L.vp_kernel_slot_name(i).decode() for i in range(KERNEL_SLOTS)]
        names[2] = self.pitch_kernel_name() or names[2]          # the build actually launched (rocprof shows this symbol)
        return {names[i]: (ms[i], n[i]) for i in range(KERNEL_SLOTS)}

    def pitch_kernel_name(self):
        return self.L.vp_pitch_kernel_name(self.h).decode()

    def debug_set_spin_limit(self, polls):
        """Diagnostic: polls a kernel's bounded inter-wavefront wait makes before it raises VP_ERR_TIMEOUT (default 2^22)."""
        self._chk(self.L.vp_debug_set_spin_limit(self.h, int(polls)))

    def vocoder_kernel_name(self):
        return self.L.vp_vocoder_kernel_name(self.h).decode()


class StftRoundTrip:
    """Standalone batched STFT -> iSTFT (no reference counterpart; see include/vp_amd.h vp_stft_*)."""

    def __init__(self, n_streams, n_samples, frame_len=1024, hop=256, device=0):
        self.L = load_library()
        h = C.c_void_p()
        rc = self.L.vp_stft_create(int(device), int(n_streams), int(n_samples), int(frame_len), int(hop), C.byref(h))
        if rc:
            raise VpError(rc, self.L.vp_error_string(rc).decode())
        self.h, self.S, self.T, self.F, self.hop = h, n_streams, n_samples, frame_len, hop
        self.n_frames = self.L.vp_stft_num_frames(h)

    def __call__(self, d_in, d_out, d_mag=None, stream=None):
        import torch
        assert d_in.is_cuda and d_in.dtype == torch.float32 and tuple(d_in.shape) == (self.S, self.T) and d_in.is_contiguous()
        assert d_out.is_cuda and tuple(d_out.shape) == (self.S, self.T) and d_out.is_contiguous()
        if stream is None:
            stream = torch.cuda.current_stream(d_in.device).cuda_stream
        rc = self.L.vp_stft_roundtrip(self.h, d_in.data_ptr(), d_out.data_ptr(), d_mag.data_ptr() if d_mag is not None else None,
                                      C.c_void_p(stream))
        if rc:
            raise VpError(rc, self.L.vp_error_string(rc).decode())

    @property
    def fused(self):
        """True when the handle runs the fused kernel (csrc/vp_stft.hip: 1024-point frames)."""
        return self.L.vp_stft_is_fused(self.h) == 1

    def set_runs(self, runs_per_stream):
        """Diagnostic: runs of frames (workgroups) per stream, 0 = automatic; the output does not depend on it."""
        rc = self.L.vp_stft_set_runs(self.h, int(runs_per_stream))
        if rc:
            raise VpError(rc, self.L.vp_error_string(rc).decode())

    def set_precision(self, precision):
        """"f64" (default) or "f32": arithmetic of the round trip's transforms (vp_stft_set_precision)."""
        rc = self.L.vp_stft_set_precision(self.h, {"f64": 0, "f32": 1}[precision])
        if rc:
            raise VpError(rc, self.L.vp_error_string(rc).decode())

    @property
    def precision(self):
        return "f32" if self.L.vp_stft_get_precision(self.h) == 1 else "f64"

    def pitch_shift(self, d_in, d_out, semitones, stream=None):
        """Round trip with the phase-vocoder stage (per-bin phase unwrap / accumulate) shifting the pitch by `semitones`."""
        import torch
        assert d_in.is_cuda and d_in.dtype == torch.float32 and tuple(d_in.shape) == (self.S, self.T) and d_in.is_contiguous()
        assert d_out.is_cuda and tuple(d_out.shape) == (self.S, self.T) and d_out.is_contiguous()
        if stream is None:
            stream = torch.cuda.current_stream(d_in.device).cuda_stream
        rc = self.L.vp_stft_pitch_shift(self.h, d_in.data_ptr(), d_out.data_ptr(), float(semitones), C.c_void_p(stream))
        if rc:
            raise VpError(rc, self.L.vp_error_string(rc).decode())

    def close(self):
        if getattr(self, "h", None):
            self.L.vp_stft_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
