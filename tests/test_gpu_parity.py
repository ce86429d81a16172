"""GPU parity tests: libvp_amd.so (through the C ABI) against the CPU oracle on identical inputs.
The kernels reproduce the reference's double arithmetic operation by operation, so the bar is
BIT-EXACT float32 output and identical pitch-tracker state (period, marks, LPC coefficients).
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

FS = 44100.0


def _streams(S, T, **kw):
    from vocoderproject_amd.synth import make_streams
    return np.ascontiguousarray(make_streams(S, T, **kw).numpy())


def _oracle_run(x, N, params, prepare=None, trace=False):
    from oracle import oracle_py as O
    outs, traces = [], []
    for s in range(x.shape[0]):
        o = O.OracleStream(**params)
        if prepare:
            o.prepare_explicit(*prepare)
        else:
            o.prepare_to_play(FS, N)
        if trace:
            y, tr = o.run(x[s], trace=True)
            traces.append(tr)
        else:
            y = o.run(x[s])
        outs.append(y)
    return (np.stack(outs), traces) if trace else np.stack(outs)


def _gpu_run(x, N, params, prepare=None):
    from vocoderproject_amd import BatchVocoderProcessor
    p = BatchVocoderProcessor(**params)
    if prepare:
        fs, n, F, H, W, h = prepare
        p.prepareExplicit(fs, n, x.shape[0], F, H, W, h)
    else:
        p.prepareToPlay(FS, N, x.shape[0])
    y = p.run(x)
    return y, p


@pytest.mark.parametrize("mode", ["pitch", "voc", "both"])
def test_bit_exact_default_geometry(mode):
    S, N, B = 6, 1024, 24
    x = _streams(S, N * B)
    params = dict(pitchBool=int(mode != "voc"), vocBool=int(mode != "pitch"))
    ref = _oracle_run(x, N, params)
    got, p = _gpu_run(x, N, params)
    assert p.getLatencySamples() == 1024
    bad = np.argwhere(got != ref)
    assert bad.size == 0, f"{len(bad)} samples differ, first at {bad[0]}, max abs {np.abs(got - ref).max()}"
    assert np.abs(ref).max() > 0.05       # the comparison is not vacuous
