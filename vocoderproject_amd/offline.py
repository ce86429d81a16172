"""Offline (file-to-file) front end: the caller one step either side of the hot path.

The reference's notebook renders whole recordings -- `pitch_corrector(x, ...)` and `vocode(x, y, ...)`
(Notebook/"Pitch Corrector and Vocoder.ipynb" cells 9 and 25) on signals read with `wavio.read(...).data[:, 0] / 32767`
(cells 3, 15, 22) and returns an output as long as, and aligned with, its input.  This module does the same with the
plugin's own path (the MI355X kernels behind the C ABI) for a whole BATCH of recordings at once: every recording is one
stream of one `BatchVocoderProcessor`, the batch is padded to a common length, pushed through `processBlock()` block by
block (several blocks per call where the library offers it), and the plugin's latency (`setLatencySamples`,
PluginProcessor.cpp:175,183) is taken off the front so that output sample t belongs to input sample t.

There is no CPU path here either: without a GPU the processor raises.

    python -m vocoderproject_amd.offline pitch  take1.wav take2.wav --out-dir tuned/ [--key 12] [--shift +3]
    python -m vocoderproject_amd.offline vocode voice.wav --carrier synth.wav --out-dir out/
"""
import argparse
import os
import sys
import wave

import numpy as np

PCM_SCALE = 32767.0          # the notebook's convention: int16 / 32767.0 (cell 3)


# ---- WAV files (stdlib `wave`: integer PCM, 8/16/24/32 bit) -------------------------------------------------------------

def read_wav(path):
    """-> (sample_rate, float32 [channels][T]) scaled like the notebook does: int16 / 32767 (other widths likewise by
    their own full scale)."""
    with wave.open(path, "rb") as w:
        nch, width, fs, n = w.getnchannels(), w.getsampwidth(), w.getframerate(), w.getnframes()
        raw = w.readframes(n)
    if width == 1:
        a = (np.frombuffer(raw, np.uint8).astype(np.float64) - 128.0) / 127.0
    elif width == 2:
        a = np.frombuffer(raw, "<i2").astype(np.float64) / PCM_SCALE
    elif width == 3:
        b = np.frombuffer(raw, np.uint8).reshape(-1, 3).astype(np.int32)
        v = b[:, 0] | (b[:, 1] << 8) | (b[:, 2] << 16)
        v = np.where(v >= 1 << 23, v - (1 << 24), v)
        a = v.astype(np.float64) / float((1 << 23) - 1)
    elif width == 4:
        a = np.frombuffer(raw, "<i4").astype(np.float64) / float((1 << 31) - 1)
    else:
        raise ValueError(f"{path}: unsupported sample width {width}")
    a = a.reshape(-1, nch).T
    return int(fs), np.ascontiguousarray(a, np.float32)


def write_wav(path, fs, y, width=2):
    """y: float [channels][T] or [T]; clipped to full scale; 16-bit (default) or 24-bit PCM."""
    y = np.asarray(y, np.float64)
    if y.ndim == 1:
        y = y[None]
    full = {2: PCM_SCALE, 3: float((1 << 23) - 1)}[width]
    v = np.rint(np.clip(y, -1.0, 1.0) * full).astype(np.int32).T          # [T][ch]
    if width == 2:
        raw = v.astype("<i2").tobytes()
    else:
        u = (v & 0xFFFFFF).astype(np.uint32).reshape(-1)
        raw = np.stack([u & 0xFF, (u >> 8) & 0xFF, (u >> 16) & 0xFF], axis=1).astype(np.uint8).tobytes()
    with wave.open(path, "wb") as w:
        w.setnchannels(y.shape[0])
        w.setsampwidth(width)
        w.setframerate(int(fs))
        w.writeframes(raw)


# ---- batching -----------------------------------------------------------------------------------------------------------

def pack_batch(voices, carriers, N, latency):
    """Recordings of different lengths -> one [S][3][T] float32 batch, T a multiple of N that leaves room for the
    plugin's latency behind the longest recording.  carriers[s] may be None (no side chain: zeros, like the null
    pointer of MyBuffer.cpp:93-102), mono [T] (copied to both side-chain channels) or stereo [2][T]."""
    S = len(voices)
    if S == 0:
        raise ValueError("no recordings")
    lens = [int(np.asarray(v).shape[-1]) for v in voices]
    T = max(lens) + int(latency)
    T = ((T + N - 1) // N) * N
    x = np.zeros((S, 3, T), np.float32)
    for s in range(S):
        v = np.asarray(voices[s], np.float32)
        if v.ndim != 1:
            raise ValueError(f"voice {s}: expected a mono signal, got shape {v.shape}")
        x[s, 0, :lens[s]] = v
        c = None if carriers is None else carriers[s]
        if c is not None:
            c = np.asarray(c, np.float32)
            if c.ndim == 1:
                c = np.stack([c, c])
            if c.ndim != 2 or c.shape[0] != 2:
                raise ValueError(f"carrier {s}: expected [T] or [2][T], got shape {c.shape}")
            n = min(c.shape[1], T)
            x[s, 1:3, :n] = c[:, :n]
    return x, lens


def unpack_batch(y, lens, latency):
    """[S][2][T] -> list of [2][len_s]: the plugin's latency taken off the front (output t <-> input t)."""
    return [np.ascontiguousarray(y[s, :, latency:latency + lens[s]]) for s in range(len(lens))]


def render(voices, carriers, fs, *, pitch=True, vocoder=False, params=None, stream_params=None, shift=None,
           N=1024, blocks_per_call=8, device=0, iir_mode="exact", yin_mode="direct", processor=None):
    """Push a batch of recordings through the plugin path.  voices: list of mono float arrays (one stream each);
    carriers: None or a list (entries None / mono / stereo).  params: plugin parameter ids -> values for all streams;
    stream_params: optional list of dicts, one per stream; shift: None, one number, or a list of numbers / None per
    stream (fixed interval in semitones, vp_set_pitch_shift).  Returns a list of float32 [2][len] outputs aligned with
    the inputs.  `processor` (tests): an object with the BatchVocoderProcessor interface to use instead of a new one."""
    S = len(voices)
    if processor is None:
        from . import BatchVocoderProcessor
        kw = dict(params or {})
        kw.update(pitchBool=int(bool(pitch)), vocBool=int(bool(vocoder)))
        processor = BatchVocoderProcessor(device=device, **kw)
    p = processor
    p.prepareToPlay(float(fs), int(N), S)
    p.set_iir_mode(iir_mode)
    p.set_yin_mode(yin_mode)
    for s, sp in enumerate(stream_params or []):
        for k, v in (sp or {}).items():
            p.setStreamParameter(s, k, v)
    if shift is not None:
        per = shift if isinstance(shift, (list, tuple)) else [shift] * S
        if len(per) != S:
            raise ValueError("shift: one value per stream expected")
        for s, v in enumerate(per):
            if v is not None:
                p.setPitchShift(float(v), on=True, stream=s)
    lat = p.latency
    x, lens = pack_batch(voices, carriers, int(N), lat)
    T = x.shape[2]
    nb = T // N
    y = np.empty((S, 2, T), np.float32)
    B = max(1, int(blocks_per_call))
    b = 0
    while b < nb:
        k = min(B, nb - b)
        if k > 1 and hasattr(p, "process_blocks"):
            xb = np.ascontiguousarray(x[:, :, b * N:(b + k) * N].reshape(S, 3, k, N).transpose(2, 0, 1, 3))   # [k][S][3][N]
            yb = p.process_blocks(xb)                                                                       # [k][S][2][N]
            y[:, :, b * N:(b + k) * N] = yb.transpose(1, 2, 0, 3).reshape(S, 2, k * N)
        else:
            for j in range(k):
                y[:, :, (b + j) * N:(b + j + 1) * N] = p.process(np.ascontiguousarray(x[:, :, (b + j) * N:(b + j + 1) * N]))
        b += k
    return unpack_batch(y, lens, lat)


def pitch_corrector(voices, fs, key=12, **kw):
    """The notebook's `pitch_corrector(x, ...)` (cell 9) for a batch: mono in, corrected mono out (the plugin writes the
    same signal to both output channels; channel 0 is returned)."""
    params = dict(kw.pop("params", None) or {})
    params.setdefault("keyPitch", int(key))
    return [o[0] for o in render(voices, None, fs, pitch=True, vocoder=False, params=params, **kw)]


def vocode(voices, carriers, fs, order_lpc=40, order_synth=5, **kw):
    """The notebook's `vocode(x, y, window, window_len, hop, order_lpc)` (cell 25) for a batch: voice + carrier in,
    cross-synthesis out ([2][len] per stream: left/right)."""
    params = dict(kw.pop("params", None) or {})
    params.setdefault("lpcVoice", int(order_lpc))
    params.setdefault("lpcSynth", int(order_synth))
    return render(voices, carriers, fs, pitch=False, vocoder=True, params=params, **kw)


# ---- command line ---------------------------------------------------------------------------------------------------------

def main(argv=None):
    ap = argparse.ArgumentParser(prog="python -m vocoderproject_amd.offline", description=__doc__.split("\n\n")[0])
    ap.add_argument("flow", choices=["pitch", "vocode", "both"])
    ap.add_argument("inputs", nargs="+", help="voice recordings (WAV; channel 0 is used, like the notebook)")
    ap.add_argument("--carrier", action="append", default=None,
                    help="side-chain recording(s) for vocode/both: one for all voices or one per voice")
    ap.add_argument("--out-dir", required=True)
    ap.add_argument("--key", type=int, default=12, help="keyPitch 0..12 (12 = chromatic, the plugin's default)")
    ap.add_argument("--shift", type=float, default=None, help="fixed interval in semitones instead of the key correction")
    ap.add_argument("--lpc-voice", type=int, default=40)
    ap.add_argument("--lpc-synth", type=int, default=5)
    ap.add_argument("--block", type=int, default=1024)
    ap.add_argument("--device", type=int, default=0)
    ap.add_argument("--fast", action="store_true", help="VP_IIR_FAST + certified cross-correlation YIN")
    a = ap.parse_args(argv)

    recs = [read_wav(f) for f in a.inputs]
    fs = recs[0][0]
    if any(r[0] != fs for r in recs):
        raise SystemExit("all recordings of a batch must share one sample rate (one prepareToPlay)")
    voices = [r[1][0] for r in recs]
    carriers = None
    if a.flow != "pitch":
        if not a.carrier:
            raise SystemExit("vocode/both need --carrier")
        cs = [read_wav(f) for f in a.carrier]
        if any(c[0] != fs for c in cs):
            raise SystemExit("carrier sample rate differs from the voices'")
        if len(cs) == 1:
            cs = cs * len(voices)
        if len(cs) != len(voices):
            raise SystemExit("--carrier: give one, or one per voice")
        carriers = [c[1][:2] if c[1].shape[0] >= 2 else c[1][0] for c in cs]
    params = dict(keyPitch=a.key, lpcVoice=a.lpc_voice, lpcSynth=a.lpc_synth)
    outs = render(voices, carriers, fs, pitch=a.flow != "vocode", vocoder=a.flow != "pitch", params=params, shift=a.shift,
                  N=a.block, device=a.device, iir_mode="fast" if a.fast else "exact", yin_mode="xcorr" if a.fast else "direct")
    os.makedirs(a.out_dir, exist_ok=True)
    for f, y in zip(a.inputs, outs):
        out = os.path.join(a.out_dir, os.path.splitext(os.path.basename(f))[0] + f"_{a.flow}.wav")
        write_wav(out, fs, y)
        print(out)
    return 0


if __name__ == "__main__":
    sys.exit(main())
