"""N > 1 path on CPU: two gloo ranks shard a stream batch, each processes its own shard (with the
oracle standing in for the GPU processor: streams are independent, so the per-shard results must
reassemble to exactly the single-process result), then gather on rank 0."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from vocoderproject_amd.dist import shard_range, shard_sizes


def test_shard_range_covers_and_is_ragged_safe():
    for S in (0, 1, 2, 7, 8, 255, 256, 8192):
        for W in (1, 2, 3, 8):
            spans = [shard_range(S, r, W) for r in range(W)]
            assert spans[0][0] == 0 and spans[-1][1] == S
            assert all(spans[i][1] == spans[i + 1][0] for i in range(W - 1))
            sz = shard_sizes(S, W)
            assert max(sz) - min(sz) <= 1 and sum(sz) == S


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, S, N, B, ret, use_async=False):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from oracle import oracle_py as O
        from vocoderproject_amd.dist import gather_streams, scatter_streams
        from vocoderproject_amd.synth import make_streams
        x_root = make_streams(S, N * B) if rank == 0 else None
        if use_async:                                   # the double-buffered exchange's building blocks (bench.py --exchange)
            lo, hi = shard_range(S, rank, world)
            buf = torch.empty((hi - lo, 3, N * B), dtype=torch.float32)
            mine, works = scatter_streams(x_root, S, (3, N * B), torch.float32, "cpu", out=buf, async_op=True)
            assert mine is buf
            for w in works:
                w.wait()
        else:
            mine = scatter_streams(x_root, S, (3, N * B), torch.float32, "cpu")
        lo, hi = shard_range(S, rank, world)
        assert mine.shape[0] == hi - lo
        out = torch.empty((hi - lo, 2, N * B), dtype=torch.float32)
        for s in range(hi - lo):
            o = O.OracleStream()
            o.prepare_to_play(44100.0, N)
            out[s] = torch.from_numpy(o.run(np.ascontiguousarray(mine[s].numpy())))
        if use_async:
            dst = torch.empty((S, 2, N * B), dtype=torch.float32) if rank == 0 else None
            full, works = gather_streams(out, S, out=dst, async_op=True)
            for w in works:
                w.wait()
            assert (full is dst) if rank == 0 else (full is None)
        else:
            full = gather_streams(out, S)
        if rank == 0:
            ret["y"] = full.numpy().copy()
            ret["x"] = x_root.numpy().copy()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("S,use_async", [(3, False), (4, False), (5, True)])
def test_two_rank_shard_process_gather_matches_single_process(S, use_async):
    from oracle import oracle_py as O
    N, B, world = 256, 12, 2
    port = _free_port()
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(world, port, S, N, B, ret, use_async), nprocs=world, join=True)
    x, y = ret["x"], ret["y"]
    ref = np.empty_like(y)
    for s in range(S):
        o = O.OracleStream()
        o.prepare_to_play(44100.0, N)
        ref[s] = o.run(np.ascontiguousarray(x[s]))
    np.testing.assert_array_equal(y, ref)
    assert np.abs(ref).max() > 0.01


def _subgroup_worker(rank, world, port, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from vocoderproject_amd.dist import gather_streams, scatter_streams
        grp = dist.new_group([1, 2])                     # group ranks 0, 1 are GLOBAL ranks 1, 2
        if rank in (1, 2):
            S = 5
            x_root = torch.arange(S * 4, dtype=torch.float32).view(S, 4) if rank == 1 else None
            mine = scatter_streams(x_root, S, (4,), torch.float32, "cpu", src=0, group=grp)
            lo, hi = shard_range(S, dist.get_rank(grp), 2)
            assert torch.equal(mine, torch.arange(S * 4, dtype=torch.float32).view(S, 4)[lo:hi])
            full = gather_streams(mine * 2, S, dst=1, group=grp)      # gather on the OTHER member (global rank 2)
            if rank == 2:
                ret["y"] = full.numpy().copy()
            else:
                assert full is None
        dist.barrier()
    finally:
        dist.destroy_process_group()


def test_scatter_gather_inside_a_subgroup_uses_group_ranks():
    """src/dst are ranks of `group`; the point-to-point calls need global ranks."""
    port = _free_port()
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_subgroup_worker, args=(3, port, ret), nprocs=3, join=True)
    np.testing.assert_array_equal(ret["y"], 2 * np.arange(20, dtype=np.float32).reshape(5, 4))


def test_bench_gpus_n_spawns_itself_and_refuses_loudly_without_the_gpus():
    """`python bench.py --gpus N` needs no launcher: it spawns the ranks itself; on a node with fewer GPUs it says so and
    exits non-zero instead of quietly running fewer ranks (round-1 verdict: the flag used to be ignored)."""
    import subprocess
    import sys
    if torch.cuda.device_count() >= 2:
        pytest.skip("this node has the GPUs")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2"], capture_output=True, text=True, env=env, timeout=300)
    assert r.returncode == 2 and "--gpus 2" in r.stderr and r.stdout.strip() == ""


def _exchange_worker(rank, world, port, S, n_steps, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from vocoderproject_amd.dist import exchange_steps
        xs = [torch.arange(S * 6, dtype=torch.float32).view(S, 3, 2) + 100.0 * i for i in range(n_steps)] if rank == 0 else None
        outs = [torch.zeros((S, 2, 2)) for _ in range(n_steps)] if rank == 0 else None
        calls = []

        def process(i_, o_):                              # stand-in for processBlock: out = (ch0 + ch1, ch0 - ch2)
            calls.append(i_.shape[0])
            o_[:, 0] = i_[:, 0] + i_[:, 1]
            o_[:, 1] = i_[:, 0] - i_[:, 2]

        exchange_steps(n_steps, S, (lambda i: xs[i]), (lambda i: outs[i]), (3, 2), (2, 2), torch.float32, "cpu", process)
        lo, hi = shard_range(S, rank, world)
        assert calls == [hi - lo] * n_steps
        if rank == 0:
            ret["ok"] = all(torch.equal(outs[i][:, 0], xs[i][:, 0] + xs[i][:, 1]) and torch.equal(outs[i][:, 1], xs[i][:, 0] - xs[i][:, 2])
                            for i in range(n_steps))
        dist.barrier()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,S,n_steps", [(2, 5, 7), (3, 4, 1), (3, 8, 6), (2, 1, 3)])
def test_double_buffered_exchange_steps(world, S, n_steps):
    """bench.py's exchange leg (root scatter -> processBlock -> root gather per step, step i+1's scatter and step
    i-1's gather in flight beside step i) with a stand-in process function: every step's gathered output is that step's
    input processed, ragged and empty shards included, and nothing deadlocks."""
    port = _free_port()
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_exchange_worker, args=(world, port, S, n_steps, ret), nprocs=world, join=True)
    assert ret["ok"]
