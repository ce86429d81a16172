"""CPU-side checks of the C-ABI library: it loads, exports every symbol include/vp_amd.h declares,
and FAILS LOUDLY without a GPU (no CPU fallback).  No compute calls here."""
import ctypes as C
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    from vocoderproject_amd import build
    return C.CDLL(build.build())


def _declared_symbols():
    txt = open(os.path.join(ROOT, "include", "vp_amd.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(vp_[a-z_0-9]+)\s*\(", txt)))


def test_every_declared_symbol_is_exported(lib):
    syms = _declared_symbols()
    assert len(syms) >= 20 and "vp_process_block" in syms and "vp_prepare_to_play" in syms
    for s in syms:
        assert hasattr(lib, s), f"{s} declared in include/vp_amd.h but not exported"


def test_abi_version_and_error_strings(lib):
    assert lib.vp_abi_version() == 3
    lib.vp_error_string.restype = C.c_char_p
    assert lib.vp_error_string(0) == b"ok"
    assert lib.vp_error_string(-3) == b"Invalid overlap"          # VocoderProcess.cpp:112
    assert b"no CPU fallback" in lib.vp_error_string(-6)


def test_default_params_match_reference_layout(lib):
    from vocoderproject_amd.processor import VpParams
    p = VpParams()
    lib.vp_default_params(C.byref(p))
    # PluginProcessor.cpp:41-69
    assert (p.gainPitch, p.gainVoice, p.gainSynth, p.gainVoc) == (0.0, -60.0, -60.0, 0.0)
    assert (p.lpcVoice, p.lpcPitch, p.lpcSynth, p.keyPitch, p.pitchBool, p.vocBool) == (40, 15, 5, 12, 1, 1)


def test_create_fails_loudly_without_gpu(lib):
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    h = C.c_void_p()
    rc = lib.vp_create(0, C.byref(h))
    assert rc == -6 and not h.value                                 # VP_ERR_NO_DEVICE
    from vocoderproject_amd import BatchVocoderProcessor, VpError
    with pytest.raises(VpError):
        BatchVocoderProcessor()


def test_product_never_imports_oracle():
    pkg = os.path.join(ROOT, "vocoderproject_amd")
    for dp, _, fs in os.walk(pkg):
        for f in fs:
            if f.endswith((".py", ".hip", ".inc", ".h", ".hpp", ".cpp")):
                t = open(os.path.join(dp, f), errors="ignore").read()
                assert "import oracle" not in t and "from oracle" not in t and "vp_oracle" not in t, f


def test_cpp_adapter_header_compiles_and_links(tmp_path):
    """include/vp_amd.hpp (the C++ mirror of the plugin's call surface) is valid C++17 on its own and every
    entry point it uses resolves against the built library; run without a GPU it must throw vp::Error, not crash."""
    import shutil
    import subprocess
    from vocoderproject_amd import build
    gxx = shutil.which("g++")
    if not gxx:
        pytest.skip("no g++")
    lib = build.build()
    src = tmp_path / "t.cpp"
    src.write_text(r"""
#include "vp_amd.hpp"
#include <cstdio>
#include <vector>
int main() {
    try {
        vp::BatchVocoderProcessor p(0);
        p.setParameter("lpcVoice", 24);
        p.prepareToPlay(44100.0, 256, 2);
        p.setStreamParameter(1, "keyPitch", 3);
        std::vector<float> io(2 * 3 * 256, 0.f);
        p.processBlock(io.data());
        std::printf("latency %d\n", p.getLatencySamples());
        return 0;
    } catch (const vp::Error &e) {
        std::printf("vp::Error %d\n", e.code);
        return e.code == VP_ERR_NO_DEVICE ? 42 : 1;
    }
}
""")
    exe = tmp_path / "t"
    subprocess.check_call([gxx, "-std=c++17", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe), lib,
                           "-Wl,-rpath," + os.path.dirname(lib)])
    import torch
    rc = subprocess.call([str(exe)])
    assert rc == (0 if torch.cuda.is_available() else 42)


def test_source_hash_and_build_dependencies_cover_every_included_kernel_source():
    """bench.py ties profiles/*_counters.json to the kernel sources by a hash, and the build caches objects by one: both must
    see every file the kernel translation units include (a part added to vp_kernels.hip and forgotten there would let a stale
    counter file or a stale object pass for current)."""
    import re
    import sys
    sys.path.insert(0, ROOT)
    import bench
    from vocoderproject_amd import build
    csrc = os.path.join(ROOT, "vocoderproject_amd", "csrc")
    seen, todo = set(), ["vp_kernels.hip", "vp_voc2.hip"]
    while todo:
        f = todo.pop()
        if f in seen:
            continue
        seen.add(f)
        for inc in re.findall(r'^#include "([^"]+)"', open(os.path.join(csrc, f)).read(), re.M):
            if os.path.exists(os.path.join(csrc, inc)):
                todo.append(inc)
    src = open(os.path.join(ROOT, "bench.py")).read()
    hashed = set(re.findall(r'"vocoderproject_amd/csrc/([^"]+)"', src[src.index("def kernel_source_hash"):src.index("def committed_counters")]))
    assert seen <= hashed, seen - hashed
    assert seen <= set(build.DEPS), seen - set(build.DEPS)
    assert len(bench.kernel_source_hash()) == 16
