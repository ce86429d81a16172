#!/usr/bin/env python3
"""Soak: the randomised parity tests of tests/test_gpu_parity.py over many more seeds than the suite carries
(GPU box only; prints a summary line, exits non-zero on the first mismatch).

    python tools/soak_fuzz.py [first_seed] [count]
    VP_SOAK_LITE=1 / VP_SOAK_R2=1 / VP_SOAK_FAST=1 select the large-batch, round-2 or FAST-mode-against-exact cases instead
    VP_SOAK_MB=1: random plans of multi-block calls (vp_k_pitch_ws_mb and its fallbacks) against block-by-block runs (round 6)
"""
import os
import sys
import traceback

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import pytest  # noqa: E402
import test_gpu_parity as T  # noqa: E402


def lite_case(seed):
    """Large-batch builds (two workgroups per CU) against the regular ones: a 260-stream batch in FAST mode must give, for
    the streams looked at, what a small batch gives up to last-bit flips of the float32 cast (same filters; the two builds
    group the recursion's sums differently)."""
    import numpy as np
    from vocoderproject_amd import BatchVocoderProcessor, VpError
    fs, N, params = T._fuzz_case(9000 + seed)
    S = 260
    nb = max(4, int(9000 * fs / 44100.0) // N)
    base = T._streams(13, N * nb, fs=fs)
    x = np.ascontiguousarray(np.tile(base, (20, 1, 1)))

    def run(xs):
        p = BatchVocoderProcessor(**params)
        p.prepareToPlay(fs, N, xs.shape[0])
        p.set_iir_mode("fast")
        p.set_yin_mode("xcorr")
        return p.run(xs)

    try:
        big = run(x)
    except VpError as e:
        assert e.code == -4, e
        raise pytest.skip.Exception("geometry beyond the LDS budget")
    pick = [0, 12, 130, 259]
    small = run(np.ascontiguousarray(x[pick]))
    dlt = np.abs(big[pick].astype(np.float64) - small)
    # (round 4: with the vocoder on, the large batch runs the lane-per-window pipeline, whose tolerance mode keeps two intermediates in
    # f32 -- last-bit differences are then the rule, bounded in size; the pitch corrector alone still only flips the odd last bit)
    scale = max(1.0, float(np.abs(small).max()))
    few = (big[pick] != small).mean() < 0.02 if not params["vocBool"] else float(np.sqrt((dlt ** 2).mean())) < 6e-8 * scale
    assert dlt.max() <= 4e-7 * scale and few, \
        f"lite vs regular, seed {seed}: fs={fs} N={N} {params}: max diff {dlt.max()}, differing {(big[pick] != small).mean():.3f}, rms {np.sqrt((dlt ** 2).mean()):.2e}"
    if params["pitchBool"] or params["vocBool"]:
        assert np.abs(big).max() > 0.01, "vacuous comparison"


def round2_case(seed):
    """Round-2 features under random schedules: per-stream pitchBool / vocBool (cohorts), the lane-per-window vocoder
    pipeline (forced or not), per-stream orders and gains, changed at random blocks -- every block bit-exact against one
    oracle instance per stream."""
    import numpy as np
    from oracle import oracle_py as O
    from vocoderproject_amd import BatchVocoderProcessor, VpError
    rng = np.random.default_rng(31000 + seed)
    fs = float(rng.choice([16000.0, 22050.0, 44100.0, 48000.0]))
    N = int(rng.choice([128, 300, 512, 1024, 2048]))
    S = int(rng.integers(3, 24))
    B = max(8, int(14000 * fs / 44100.0) // N)
    x = T._streams(S, N * B, fs=fs)
    x[0, 0] *= np.where((np.arange(N * B) // 6000) % 2 == 0, 1.0, 2e-5).astype(np.float32)
    base = dict(lpcVoice=int(rng.integers(2, 49)), lpcPitch=int(rng.integers(2, 40)), lpcSynth=int(rng.integers(2, 31)))
    p = BatchVocoderProcessor(**base)
    try:
        p.prepareToPlay(fs, N, S)
    except VpError as e:
        assert e.code == -4, e
        raise pytest.skip.Exception("geometry beyond the LDS budget")
    p.set_vocoder_path(str(rng.choice(["batched", "workgroup", "batched"])))
    os_ = []
    for s_ in range(S):
        o = O.OracleStream(**base)
        o.prepare_to_play(fs, N)
        os_.append(o)
    choices = [("pitchBool", lambda: int(rng.random() < 0.6)), ("vocBool", lambda: int(rng.random() < 0.7)),
               ("lpcVoice", lambda: int(rng.integers(2, 49))), ("lpcSynth", lambda: int(rng.integers(2, 31))),
               ("keyPitch", lambda: int(rng.integers(0, 13))), ("gainVoc", lambda: float(rng.uniform(-20, 6))),
               ("gainVoice", lambda: float(rng.choice([-60.0, -20.0]))), ("gainSynth", lambda: float(rng.choice([-60.0, -15.0])))]
    for b in range(B):
        for _ in range(int(rng.integers(0, 3))):
            s_ = int(rng.integers(0, S)); k, f = choices[int(rng.integers(0, len(choices)))]; val = f()
            p.setStreamParameter(s_, k, val)
            os_[s_].set_param(k, val)
        blk = np.ascontiguousarray(x[:, :, b * N:(b + 1) * N])
        got = p.process(blk)
        for s_ in range(S):
            io = blk[s_].copy()
            os_[s_].process_block(io)
            T._assert_equal(got[s_], io[:2], f"seed {seed}: fs={fs} N={N} S={S} block {b} stream {s_}")
    ub = np.sum([o.ub_counters() for o in os_], axis=0)
    assert list(p.ub_counters()) == list(ub)


def mb_case(seed):
    """Round 6: queued blocks in one launch.  Random geometry among those the wave-specialised kernel serves (and some it does not:
    the call then falls back to the phase kernels' multi-block build or to block by block), random order, keys, dry gains, channel
    form and call plan; output, tracker states and UB-site counters must equal a block-by-block run's bit for bit."""
    import numpy as np
    import torch
    from vocoderproject_amd import BatchVocoderProcessor
    rng = np.random.default_rng(52000 + seed)
    geoms = [None, None, (44100.0, 512, 1024, 768, 512, 128), (44100.0, 256, 1024, 768, 512, 128), (44100.0, 1024, 1024, 512, 512, 256),
             (44100.0, 512, 1024, 896, 512, 128), (44100.0, 768, 1024, 768, 512, 256), (48000.0, 1024, None), (22050.0, 512, None)]
    prepare = geoms[int(rng.integers(0, len(geoms)))]
    fs, N = (prepare[0], prepare[1]) if prepare else (44100.0, 1024)
    order = int(rng.choice([2, 8, 15, 15, 16, 17, 24, 24, 30]))
    mono = bool(rng.random() < 0.5)
    iir = "fast" if rng.random() < 0.6 else "exact"
    S = int(rng.integers(2, 12))
    plan = [int(rng.choice([1, 2, 3, 5, 8, 16, 17, 20])) for _ in range(int(rng.integers(3, 7)))]
    B = sum(plan)
    x = T._streams(S, N * B, fs=fs, first_stream=int(rng.integers(0, 4000)))
    x[0, 0] *= np.where((np.arange(N * B) // 7000) % 2 == 0, 1.0, 2e-5).astype(np.float32)        # gate crossings
    if S > 2:
        x[1, 0, N * (B // 2):] = 0.0                                                                 # silence()
    kw = dict(vocBool=0, lpcPitch=order)
    if rng.random() < 0.4:
        kw.update(gainVoice=-12.0, gainSynth=-20.0)
    keys = [int(rng.integers(0, 13)) for _ in range(S)]

    def make():
        p = BatchVocoderProcessor(**kw)
        if prepare and prepare[2] is not None:
            p.prepareExplicit(fs, N, S, *prepare[2:])
        else:
            p.prepareToPlay(fs, N, S)
        p.set_iir_mode(iir)
        p.set_yin_mode("xcorr")
        for s_ in range(S):
            p.setStreamParameter(s_, "keyPitch", keys[s_])
        return p

    xd = torch.from_numpy(x).cuda()

    def blocks(b0, n):
        t = torch.stack([xd[:, :, (b0 + k) * N:(b0 + k + 1) * N] for k in range(n)])
        return t[:, :, 0, :].contiguous() if mono else t.contiguous()

    d_out = torch.empty((S, 2, N), dtype=torch.float32, device="cuda")

    def one(p, xb):
        (p.process_mono_device if mono else p.process_device)(xb, d_out)
        return d_out.cpu().numpy()

    ref_p = make()
    ref = np.empty((S, 2, N * B), np.float32)
    for b in range(B):
        ref[:, :, b * N:(b + 1) * N] = one(ref_p, blocks(b, 1)[0])
    ref_state = [sorted((k, v.tobytes() if hasattr(v, "tobytes") else v) for k, v in ref_p.pitch_state(s_).items()) for s_ in range(S)]
    ref_ub = ref_p.ub_counters()
    ref_p.close()
    p = make()
    p.reserve_blocks(max(plan))
    out = np.empty_like(ref)
    b = 0
    for n in plan:
        xin = blocks(b, n)
        if n == 1:
            out[:, :, b * N:(b + 1) * N] = one(p, xin[0])
        else:
            yo = torch.empty((n, S, 2, N), dtype=torch.float32, device="cuda")
            (p.process_blocks_mono_device if mono else p.process_blocks_device)(xin, yo)
            o = yo.cpu().numpy()
            for k in range(n):
                out[:, :, (b + k) * N:(b + k + 1) * N] = o[k]
        b += n
    p.synchronize()
    what = f"mb seed {seed}: fs={fs} N={N} prepare={prepare} order={order} iir={iir} mono={mono} S={S} plan={plan} kernel={p.pitch_kernel_name()}"
    T._assert_equal(out, ref, what)
    assert [sorted((k, v.tobytes() if hasattr(v, "tobytes") else v) for k, v in p.pitch_state(s_).items()) for s_ in range(S)] == ref_state, what
    assert p.ub_counters() == ref_ub, what
    v = p.debug_stamps(reset=False)
    assert [round(v[i] * 100.0) for i in (59, 60, 61)] == [0, 0, 0], what
    assert np.abs(ref).max() > 0.01, "vacuous comparison"
    p.close()


def main():
    first = int(sys.argv[1]) if len(sys.argv) > 1 else 100
    count = int(sys.argv[2]) if len(sys.argv) > 2 else 40
    ok = skipped = 0
    for seed in range(first, first + count):
        fns = (mb_case,) if os.environ.get("VP_SOAK_MB") else (lite_case,) if os.environ.get("VP_SOAK_LITE") else (round2_case,) if os.environ.get("VP_SOAK_R2") else \
              (T.test_randomised_configurations_fast_modes_against_exact,) if os.environ.get("VP_SOAK_FAST") else \
              (T.test_randomised_configurations_bit_exact, T.test_randomised_configurations_with_extensions_bit_exact)
        for fn in fns:
            try:
                fn(seed)
                ok += 1
            except pytest.skip.Exception:
                skipped += 1
            except Exception:
                traceback.print_exc()
                print(f"FAILED: {fn.__name__}({seed})")
                return 1
    print(f"soak ok: {ok} cases bit-exact, {skipped} skipped (geometry beyond the LDS budget), seeds {first}..{first + count - 1}")
    return 0


if __name__ == "__main__":
    sys.exit(main())
