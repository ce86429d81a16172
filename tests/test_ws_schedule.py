"""Host logic of the wave-specialised pitch kernel (no GPU): the block's SCHEDULE that the host builds and hands to `vp_k_pitch_ws`
as a kernel argument (`ws_build_sched`, csrc/vp_common.h) against an independent restatement of PitchProcess::process's chunk
loop (PitchProcess.cpp:166-196: a Cont chunk of the frame in flight when nChunk != 0, then -- when that was the frame's last
chunk, or none is in flight -- a Start), and the LDS carve both sides share (`ws_carve`): regions ascending, disjoint, 16-byte
aligned where the kernel needs it, within the CU's 160 KB for the plugin's geometry."""
import json
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SRC = r"""
#include <cstdio>
#include <cstring>
#include <initializer_list>
#include "vp_common.h"
int main()
{
    printf("[");
    bool first = true;
    for (int cpf : {2, 4, 8, 16}) {
        VpGeom g;
        memset(&g, 0, sizeof g);
        g.F = 1024; g.cpf = cpf; g.C = g.F / cpf; g.N = 1024; g.tauMax = 512; g.toKeep = 1024;
        g.eLen = g.toKeep + g.F + (cpf - 1) * g.C;
        for (int nChunk0 = 0; nChunk0 < cpf; nChunk0++)
            for (int nSteps = 0; nSteps <= 2 * cpf + 1; nSteps++) {
                VpWsSched sc;
                memset(&sc, 0xff, sizeof sc);
                const bool ok = ws_build_sched(g, nChunk0, nSteps, sc);
                printf("%s{\"cpf\":%d,\"nChunk0\":%d,\"nSteps\":%d,\"ok\":%d", first ? "" : ",", cpf, nChunk0, nSteps, ok ? 1 : 0);
                first = false;
                if (ok) {
                    printf(",\"nInst\":%d,\"nStart\":%d,\"nSeg\":%d,\"inst\":[", sc.nInst, sc.nStart, sc.nSeg);
                    for (int j = 0; j < sc.nInst; j++)
                        printf("%s[%d,%d,%d,%d]", j ? "," : "", sc.instStep[j], sc.instK[j], sc.instPar[j], sc.instStart[j]);
                    printf("],\"start\":[");
                    for (int i = 0; i < sc.nStart; i++) printf("%s[%d,%d,%d]", i ? "," : "", sc.startStep[i], sc.startPar[i], sc.startNeed[i]);
                    printf("],\"seg\":[");
                    for (int q = 0; q < sc.nSeg; q++) printf("%s[%d,%d]", q ? "," : "", sc.segA[q], sc.segB[q]);
                    printf("]");
                }
                printf("}");
            }
    }
    printf("]\n");
    // the carve for the plugin's geometry at 44.1 kHz, blocks of 1024 (four chunk steps; five when a block straddles)
    VpGeom g;
    memset(&g, 0, sizeof g);
    g.F = 1024; g.cpf = 4; g.C = 256; g.N = 1024; g.tauMax = 441; g.toKeep = 1024; g.eLen = g.toKeep + g.F + 3 * g.C;
    for (int nSteps : {1, 4, 5}) {
        const WsCarve c = ws_carve(g, nSteps);
        printf("%d %d %d %d %d %d %d %d %d %d %d %d %d %d %d %d %d %d %d %d %zu %zu\n", nSteps, c.xs, c.eF, c.fr, c.qtab, c.htab, c.P, c.dY, c.gtab, c.r,
               c.aPrev, c.hp, c.aF, c.xp, c.hist, c.tw, c.oA, c.st, c.ctl, c.end, vp_pitch_ws_lds_bytes(g, nSteps), sizeof(WsCtl));
    }
    return 0;
}
"""


@pytest.fixture(scope="module")
def host_dump(tmp_path_factory):
    gxx = shutil.which("g++")
    if gxx is None:
        pytest.skip("no g++")
    d = tmp_path_factory.mktemp("ws_sched")
    src = d / "dump.cpp"
    src.write_text(SRC)
    exe = d / "dump"
    subprocess.run([gxx, "-std=c++17", "-O1", "-I", os.path.join(ROOT, "vocoderproject_amd", "csrc"), "-I", os.path.join(ROOT, "include"),
                    str(src), "-o", str(exe)], check=True)
    out = subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout.splitlines()
    return json.loads(out[0]), [list(map(int, ln.split())) for ln in out[1:]]


def _reference_schedule(cpf, n_chunk0, n_steps):
    """PitchProcess::process's loop over chunk steps, instance by instance: (step, chunk index within its frame, is_start)."""
    inst = []
    n_chunk = n_chunk0
    for t in range(n_steps):
        if n_chunk != 0:
            inst.append((t, n_chunk, False))            # processChunkCont of the frame in flight (:173-175 / :185-187)
        if n_chunk == cpf - 1:
            n_chunk = 0                                 # that was its last chunk
        if n_chunk == 0:
            inst.append((t, 0, True))                   # processChunkStart of the next frame, same step (:176-184)
        n_chunk += 1
    return inst


def test_schedule_matches_the_reference_chunk_loop(host_dump):
    cases, _ = host_dump
    assert len(cases) > 100
    for cse in cases:
        ref = _reference_schedule(cse["cpf"], cse["nChunk0"], cse["nSteps"])
        if not cse["ok"]:
            assert len(ref) == 0 or len(ref) > 24 or sum(1 for r in ref if r[2]) > 8, cse      # empty, or beyond the kernel's tables
            continue
        inst = cse["inst"]
        assert cse["nInst"] == len(inst) == len(ref)
        assert [(i[0], i[1], i[3] >= 0) for i in inst] == ref, cse
        # frames alternate between the two parity buffers; the frame in flight at entry has parity 0
        par = 0
        n_start = 0
        for (step, k, p, si) in inst:
            if si >= 0:
                par ^= 1
                assert si == n_start and k == 0
                n_start += 1
            assert p == par, cse
        assert cse["nStart"] == n_start == len(cse["start"])
        # starts: their step and parity; startNeed = the instances of the previous frame of that parity (all before the start)
        for i, (step, p, need) in enumerate(cse["start"]):
            j = next(j for j, x in enumerate(inst) if x[3] == i)
            assert (step, p) == (inst[j][0], inst[j][2])
            earlier_same_par = [q for q in range(j) if inst[q][2] == p]
            assert need == (max(earlier_same_par) + 1 if earlier_same_par else 0), cse
        # segments: maximal runs of instances of one frame, in order, covering every instance exactly once
        seg = cse["seg"]
        assert cse["nSeg"] == len(seg) and seg[0][0] == 0 and seg[-1][1] == len(inst) - 1
        for q, (a, b) in enumerate(seg):
            assert a <= b and (q == 0 or a == seg[q - 1][1] + 1)
            assert all(inst[j][3] < 0 for j in range(a + 1, b + 1))                      # only a segment's first instance may be a Start
            assert all(inst[j][2] == inst[a][2] for j in range(a, b + 1))                # one frame: one parity
            assert [inst[j][1] for j in range(a, b + 1)] == list(range(inst[a][1], inst[a][1] + b - a + 1))   # consecutive chunks
            if b + 1 < len(inst):
                assert inst[b + 1][3] >= 0                                               # the next segment opens with a Start


def test_lds_carve_is_ordered_and_fits(host_dump):
    _, carves = host_dump
    for row in carves:
        n_steps, *offs, lds_bytes, ctl_bytes = row
        end = offs[-1]
        regions = offs[:-1]
        assert regions == sorted(regions) and len(set(regions)) == len(regions), row      # ascending, no two regions at one offset
        assert all(o % 2 == 0 for o in regions), row                                       # 16-byte aligned (offsets are in doubles)
        assert lds_bytes == end * 8
        assert (end - regions[-1]) * 8 >= ctl_bytes                                        # the control block fits behind its offset
        assert lds_bytes + 512 <= 160 * 1024, row                                          # the CU's LDS, with the kernel's static share
