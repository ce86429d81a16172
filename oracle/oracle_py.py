"""ctypes loader for the CPU oracle (oracle/vp_oracle.c).

TEST INFRASTRUCTURE ONLY: importable from tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg; never from vocoderproject_amd/.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

MARK_CAP = 64
ORDER_MAX = 100

PARAM_IDS = ("gainPitch", "gainVoice", "gainSynth", "gainVoc", "lpcVoice", "lpcPitch",
             "lpcSynth", "keyPitch", "pitchBool", "vocBool")


class PitchFrame(C.Structure):
    _fields_ = [
        ("gated", C.c_int),
        ("period", C.c_int), ("prevPeriod", C.c_int), ("prevVoicedPeriod", C.c_int), ("periodNew", C.c_int),
        ("pitch", C.c_double), ("prevPitch", C.c_double), ("beta", C.c_double), ("closestFreq", C.c_double),
        ("nAn", C.c_int), ("nSt", C.c_int),
        ("anMarks", C.c_int * MARK_CAP),
        ("stMarks", C.c_int * MARK_CAP),
        ("a", C.c_double * (ORDER_MAX + 1)),
    ]

    def as_dict(self):
        return dict(gated=self.gated, period=self.period, prevPeriod=self.prevPeriod,
                    prevVoicedPeriod=self.prevVoicedPeriod, periodNew=self.periodNew, pitch=self.pitch,
                    prevPitch=self.prevPitch, beta=self.beta, closestFreq=self.closestFreq,
                    anMarks=list(self.anMarks[: self.nAn]), stMarks=list(self.stMarks[: self.nSt]),
                    a=np.array(self.a[:]))


def build(force=False, asan=False):
    """Compile the oracle with gcc (Makefile in this directory)."""
    target = "libvp_oracle_asan.so" if asan else "libvp_oracle.so"
    path = os.path.join(_HERE, target)
    src = os.path.join(_HERE, "vp_oracle.c")
    hdr = os.path.join(_HERE, "vp_oracle.h")
    stale = (not os.path.exists(path)
             or os.path.getmtime(path) < max(os.path.getmtime(src), os.path.getmtime(hdr)))
    if force or stale:
        subprocess.check_call(["make", "-s", "-C", _HERE, target])
    return path


_REF_DIR = "/root/reference"
_REF_NOTES = None


def build_ref(force=False):
    """oracle/_ref/libnotes_ref.so: the REFERENCE's own Source/Notes.cpp (the one translation unit that needs only the standard
    library) compiled where it lies under /root/reference, behind a C binding (ref_notes_shim.cpp).  Only the build container has
    /root/reference; elsewhere the prebuilt file is used if it travelled.  Returns its path, or None."""
    path = os.path.join(_HERE, "_ref", "libnotes_ref.so")
    if os.path.isdir(os.path.join(_REF_DIR, "Source")):
        src = [os.path.join(_REF_DIR, "Source", "Notes.cpp"), os.path.join(_REF_DIR, "Source", "Notes.h"), os.path.join(_HERE, "ref_notes_shim.cpp")]
        if force or not os.path.exists(path) or os.path.getmtime(path) < max(os.path.getmtime(f) for f in src):
            subprocess.check_call(["make", "-s", "-C", _HERE, "-B" if force else "-s", "ref"])
    return path if os.path.exists(path) else None


class RefNotes:
    """The reference's Notes object (Notes.h:20-44) through the binding: prepare (Notes.cpp:24-37), getClosestFreq (:79-110)."""

    def __init__(self, key, f_min, f_max):
        global _REF_NOTES
        if _REF_NOTES is None:
            p = build_ref()
            if p is None:
                raise FileNotFoundError("oracle/_ref/libnotes_ref.so is not built and /root/reference is not here")
            R = C.CDLL(p)
            R.refnotes_new.restype = C.c_void_p
            R.refnotes_free.argtypes = [C.c_void_p]
            R.refnotes_prepare.argtypes = [C.c_void_p, C.c_int, C.c_double, C.c_double]
            R.refnotes_closest.argtypes = [C.c_void_p, C.c_double, C.c_int]
            R.refnotes_closest.restype = C.c_double
            _REF_NOTES = R
        self.R = _REF_NOTES
        self.h = self.R.refnotes_new()
        self.R.refnotes_prepare(self.h, int(key), float(f_min), float(f_max))

    def closest(self, pitch, key):
        return self.R.refnotes_closest(self.h, float(pitch), int(key))

    def __del__(self):
        if getattr(self, "h", None):
            self.R.refnotes_free(self.h)
            self.h = None


def lib():
    global _LIB
    if _LIB is None:
        L = C.CDLL(build())
        dp = C.POINTER(C.c_double)
        fp = C.POINTER(C.c_float)
        L.vpo_create.restype = C.c_void_p
        L.vpo_destroy.argtypes = [C.c_void_p]
        L.vpo_set_param.argtypes = [C.c_void_p, C.c_char_p, C.c_float]
        L.vpo_get_param.argtypes = [C.c_void_p, C.c_char_p]
        L.vpo_set_pitch_shift.argtypes = [C.c_void_p, C.c_int, C.c_double]
        L.vpo_get_param.restype = C.c_float
        L.vpo_prepare_to_play.argtypes = [C.c_void_p, C.c_double, C.c_int]
        L.vpo_prepare_explicit.argtypes = [C.c_void_p, C.c_double] + [C.c_int] * 5
        L.vpo_process_block.argtypes = [C.c_void_p, fp, fp, fp]
        L.vpo_process_block_mono.argtypes = [C.c_void_p, fp, fp, fp]
        L.vpo_get_latency.argtypes = [C.c_void_p]
        L.vpo_get_geometry.argtypes = [C.c_void_p, C.POINTER(C.c_int)]
        L.vpo_trace_count.argtypes = [C.c_void_p]
        L.vpo_trace_get.argtypes = [C.c_void_p, C.c_int, C.POINTER(PitchFrame)]
        L.vpo_ub_counters.argtypes = [C.c_void_p, C.POINTER(C.c_long)]
        L.vpo_set_ftz.argtypes = [C.c_void_p, C.c_int]
        L.vpo_biased_autocorr.argtypes = [dp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, dp, dp]
        L.vpo_levinson_durbin.argtypes = [dp, dp, dp, C.c_int, C.c_int]
        L.vpo_notes_build.argtypes = [C.c_int, C.c_double, C.c_double, dp]
        L.vpo_notes_closest.argtypes = [dp, C.c_int, C.c_double]
        L.vpo_notes_closest.restype = C.c_double
        L.vpo_hann.argtypes = [dp, C.c_int]
        L.vpo_vocoder_windows.argtypes = [C.c_int, C.c_int, dp, dp]
        L.vpo_pitch_st_window.argtypes = [C.c_int, C.c_int, dp]
        L.vpo_yin_temp_linear.argtypes = [dp, C.c_int, C.c_int, dp]
        L.vpo_yin_pick.argtypes = [dp, C.c_int, C.c_double, C.c_double, C.c_double]
        L.vpo_kat_pitch_marks.argtypes = [dp, C.c_int, C.c_int, C.c_double, C.c_int, C.POINTER(C.c_int)]
        L.vpo_kat_pitch_marks_seq.argtypes = [dp, C.c_int, C.POINTER(C.c_int), C.c_int, C.c_int, C.c_double,
                                               C.POINTER(C.c_int), C.POINTER(C.c_int)]
        L.vpo_kat_marks_seq.argtypes = [dp, C.c_int, C.POINTER(C.c_int), C.c_int, C.c_int, C.c_double,
                                         C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int),
                                         C.POINTER(C.c_int), dp]
        ip = C.POINTER(C.c_int)
        L.vpo_kat_psola.argtypes = [dp, C.c_int, C.c_int, C.c_double, C.c_int, C.c_double, ip, C.c_int, ip, C.c_int, dp]
        L.vpo_kat_pitch_filters.argtypes = [dp, C.c_int, C.c_int, C.c_double, dp, C.c_int, dp, dp, dp]
        L.vpo_kat_voc_window.argtypes = [dp, dp, C.c_int, C.c_int, dp, C.c_int, dp, C.c_int, dp, dp, dp, dp, dp]
        L.vpo_gain_to_db.argtypes = [C.c_double]
        L.vpo_gain_to_db.restype = C.c_double
        L.vpo_db_to_gain_f.argtypes = [C.c_float]
        L.vpo_db_to_gain_f.restype = C.c_float
        _LIB = L
    return _LIB


def _dp(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


GEOM_KEYS = ("N", "F", "H", "C", "W", "h", "toKeep", "latency", "inSize", "outSize", "tauMax", "chunksPerFrame")


class OracleStream:
    """One reference plugin instance (VocoderAudioProcessor) restated on the CPU."""

    def __init__(self, **params):
        self.L = lib()
        self.h = C.c_void_p(self.L.vpo_create())
        for k, v in params.items():
            self.set_param(k, v)

    def __del__(self):
        try:
            if self.h:
                self.L.vpo_destroy(self.h)
                self.h = None
        except Exception:
            pass

    def set_param(self, pid, value):
        rc = self.L.vpo_set_param(self.h, pid.encode(), float(value))
        if rc:
            raise ValueError(f"bad parameter {pid}={value}")

    def set_pitch_shift(self, semitones, on=True):
        if self.L.vpo_set_pitch_shift(self.h, int(bool(on)), float(semitones)):
            raise ValueError(f"bad pitch shift {semitones}")

    def get_param(self, pid):
        return self.L.vpo_get_param(self.h, pid.encode())

    def prepare_to_play(self, sample_rate, samples_per_block):
        rc = self.L.vpo_prepare_to_play(self.h, float(sample_rate), int(samples_per_block))
        if rc:
            raise ValueError(f"prepare_to_play failed rc={rc}")
        self._N = samples_per_block

    def prepare_explicit(self, sample_rate, samples_per_block, F, H, W, h):
        rc = self.L.vpo_prepare_explicit(self.h, float(sample_rate), int(samples_per_block), F, H, W, h)
        if rc:
            raise ValueError(f"prepare_explicit failed rc={rc}")
        self._N = samples_per_block

    def geometry(self):
        g = (C.c_int * 12)()
        self.L.vpo_get_geometry(self.h, g)
        return dict(zip(GEOM_KEYS, list(g)))

    @property
    def latency(self):
        return self.L.vpo_get_latency(self.h)

    def process_block(self, io):
        """io: float32 [3][N] C-contiguous, processed in place (ch0,ch1 = out L,R; ch2 = 0)."""
        assert io.dtype == np.float32 and io.shape == (3, self._N) and io.flags.c_contiguous
        fp = C.POINTER(C.c_float)
        rc = self.L.vpo_process_block(self.h, io[0].ctypes.data_as(fp), io[1].ctypes.data_as(fp),
                                      io[2].ctypes.data_as(fp))
        if rc:
            raise RuntimeError(f"process_block rc={rc}")

    def process_block_mono(self, voice):
        """voice: float32 [N]; side chain absent (null pointers -> zeros).  Returns float32 [2][N]."""
        assert voice.dtype == np.float32 and voice.shape == (self._N,) and voice.flags.c_contiguous
        out = np.empty((2, self._N), np.float32)
        fp = C.POINTER(C.c_float)
        rc = self.L.vpo_process_block_mono(self.h, voice.ctypes.data_as(fp), out[0].ctypes.data_as(fp), out[1].ctypes.data_as(fp))
        if rc:
            raise RuntimeError(f"process_block_mono rc={rc}")
        return out

    def run(self, x, trace=False):
        """x: float32 [3][T] with T a multiple of N. Returns float32 [2][T] (and per-frame traces)."""
        N = self._N
        T = x.shape[1]
        assert T % N == 0
        out = np.empty((2, T), np.float32)
        traces = []
        io = np.empty((3, N), np.float32)
        for b in range(T // N):
            io[:] = x[:, b * N:(b + 1) * N]
            self.process_block(io)
            out[:, b * N:(b + 1) * N] = io[:2]
            if trace:
                traces.extend(self.traces())
        return (out, traces) if trace else out

    def traces(self):
        n = self.L.vpo_trace_count(self.h)
        res = []
        for i in range(n):
            f = PitchFrame()
            self.L.vpo_trace_get(self.h, i, C.byref(f))
            res.append(f.as_dict())
        return res

    def ub_counters(self):
        c = (C.c_long * 5)()
        self.L.vpo_ub_counters(self.h, c)
        return list(c)

    def set_ftz(self, on):
        self.L.vpo_set_ftz(self.h, int(on))


# ---- primitive wrappers for KATs -----------------------------------------------------------------

def biased_autocorr(ring, curr_counter, start_sample, order, wlen, window):
    ring = np.ascontiguousarray(ring, np.float64)
    window = np.ascontiguousarray(window, np.float64)
    r = np.zeros(order + 1)
    lib().vpo_biased_autocorr(_dp(ring), len(ring), curr_counter, start_sample, order, wlen, _dp(window), _dp(r))
    return r


def levinson_durbin(r, order, a_len=None):
    r = np.ascontiguousarray(r, np.float64)
    a_len = a_len or order + 1
    a = np.ones(a_len)
    ap = np.ones(a_len)
    lib().vpo_levinson_durbin(_dp(r), _dp(a), _dp(ap), order, a_len)
    return a


def notes_build(key, fmin=100.0, fmax=800.0):
    f = np.zeros(89)
    n = lib().vpo_notes_build(key, fmin, fmax, _dp(f))
    return f, n


def notes_closest(pitch, key, fmin=100.0, fmax=800.0):
    f, n = notes_build(key, fmin, fmax)
    return lib().vpo_notes_closest(_dp(f), n, float(pitch))


def hann(n):
    w = np.zeros(n)
    lib().vpo_hann(_dp(w), n)
    return w


def vocoder_windows(wlen, hop):
    a = np.zeros(wlen)
    s = np.zeros(wlen)
    rc = lib().vpo_vocoder_windows(wlen, hop, _dp(a), _dp(s))
    if rc:
        raise ValueError("Invalid overlap")
    return a, s


def pitch_st_window(F, H):
    w = np.zeros(F)
    rc = lib().vpo_pitch_st_window(F, H, _dp(w))
    if rc:
        raise ValueError("bad geometry")
    return w


def yin_temp_linear(x, frame_len, tau_max):
    x = np.ascontiguousarray(x, np.float64)
    assert len(x) >= frame_len + tau_max
    y = np.zeros(tau_max + 1)
    lib().vpo_yin_temp_linear(_dp(x), frame_len, tau_max, _dp(y))
    return y


def yin_pick(yt, tau_max, fs, fmax=800.0, tol=0.25):
    yt = np.ascontiguousarray(yt, np.float64)
    assert len(yt) >= tau_max + 1
    return lib().vpo_yin_pick(_dp(yt), tau_max, fs, fmax, tol)


def gain_to_db(g):
    return lib().vpo_gain_to_db(float(g))


def db_to_gain_f(db):
    return lib().vpo_db_to_gain_f(float(db))


def kat_pitch_marks(x, period, F=1024, H=768, fs=44100.0):
    x = np.ascontiguousarray(x, np.float64)
    m = (C.c_int * MARK_CAP)()
    n = lib().vpo_kat_pitch_marks(_dp(x), F, H, fs, int(period), m)
    if n < 0:
        raise RuntimeError(f"kat_pitch_marks rc={n}")
    return list(m[:n])


def kat_pitch_marks_seq(x, periods, F=1024, H=768, fs=44100.0):
    """Consecutive frames through pitchMarks with the tracker state rolled as yin() does; periods[f] == 0 = unvoiced."""
    x = np.ascontiguousarray(x, np.float64)
    nf = len(periods)
    assert x.size >= F + (nf - 1) * H
    per = (C.c_int * nf)(*[int(p) for p in periods])
    m = (C.c_int * (MARK_CAP * nf))()
    cnt = (C.c_int * nf)()
    rc = lib().vpo_kat_pitch_marks_seq(_dp(x), nf, per, F, H, fs, m, cnt)
    if rc:
        raise RuntimeError(f"kat_pitch_marks_seq rc={rc}")
    return [list(m[f * MARK_CAP: f * MARK_CAP + cnt[f]]) for f in range(nf)]


def kat_marks_seq(x, periods, F=1024, H=768, fs=44100.0):
    """As kat_pitch_marks_seq plus placeStMarks after every frame (key = chromatic):
    returns (anMarks per frame, stMarks per frame, periodNew per frame, beta per frame)."""
    x = np.ascontiguousarray(x, np.float64)
    nf = len(periods)
    assert x.size >= F + (nf - 1) * H
    per = (C.c_int * nf)(*[int(p) for p in periods])
    m = (C.c_int * (MARK_CAP * nf))()
    cnt = (C.c_int * nf)()
    sm = (C.c_int * (MARK_CAP * nf))()
    scnt = (C.c_int * nf)()
    pn = (C.c_int * nf)()
    beta = np.zeros(nf)
    rc = lib().vpo_kat_marks_seq(_dp(x), nf, per, F, H, fs, m, cnt, sm, scnt, pn, _dp(beta))
    if rc:
        raise RuntimeError(f"kat_marks_seq rc={rc}")
    an = [list(m[f * MARK_CAP: f * MARK_CAP + cnt[f]]) for f in range(nf)]
    st = [list(sm[f * MARK_CAP: f * MARK_CAP + scnt[f]]) for f in range(nf)]
    return an, st, list(pn), beta


def kat_psola(e, T, beta, an_marks, st_marks, F=1024, H=768, fs=44100.0):
    """PitchProcess::psola + interp for one frame, all synthesis marks at once.  e: eFrame[0 .. toKeep + F) with toKeep = F.
    Returns (outEFrame [F], number of Q2/Q3 paths taken)."""
    e = np.ascontiguousarray(e, np.float64)
    assert e.shape == (2 * F,)
    an = np.ascontiguousarray(an_marks, np.int32)
    st = np.ascontiguousarray(st_marks, np.int32)
    out = np.zeros(F)
    ip = C.POINTER(C.c_int)
    rc = lib().vpo_kat_psola(_dp(e), F, H, float(fs), int(T), float(beta), an.ctypes.data_as(ip), len(an), st.ctypes.data_as(ip), len(st), _dp(out))
    if rc < 0:
        raise ValueError(f"kat_psola rc={rc}")
    return out, rc


def kat_pitch_filters(x, a, out_e, F=1024, H=768, fs=44100.0):
    """filterFIR(-toKeep, toKeep + F, 0) of x [toKeep + F] and the chunked filterIIR of out_e [F], coefficients a[0..p]."""
    x = np.ascontiguousarray(x, np.float64)
    a = np.ascontiguousarray(a, np.float64)
    out_e = np.ascontiguousarray(out_e, np.float64)
    assert x.shape == (2 * F,) and out_e.shape == (F,)
    e, y = np.zeros(2 * F), np.zeros(F)
    rc = lib().vpo_kat_pitch_filters(_dp(x), F, H, float(fs), _dp(a), len(a) - 1, _dp(out_e), _dp(e), _dp(y))
    if rc:
        raise ValueError(f"kat_pitch_filters rc={rc}")
    return e, y


def kat_voc_window(voice, synth, hop, a_v, a_s):
    """One vocoder window from a fresh state with given coefficient vectors: (eVoice, eSynth, (EeV, EeS), g, out)."""
    voice = np.ascontiguousarray(voice, np.float64)
    synth = np.ascontiguousarray(synth, np.float64)
    a_v = np.ascontiguousarray(a_v, np.float64)
    a_s = np.ascontiguousarray(a_s, np.float64)
    W = len(voice)
    ev, es, out, EE, g = np.zeros(W), np.zeros(W), np.zeros(W), np.zeros(2), np.zeros(1)
    rc = lib().vpo_kat_voc_window(_dp(voice), _dp(synth), W, int(hop), _dp(a_v), len(a_v) - 1, _dp(a_s), len(a_s) - 1, _dp(ev), _dp(es),
                                  _dp(EE), _dp(g), _dp(out))
    if rc:
        raise ValueError(f"kat_voc_window rc={rc}")
    return ev, es, EE, float(g[0]), out
