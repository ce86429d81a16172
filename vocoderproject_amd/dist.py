"""Sharding of the stream batch across ranks (one process per GPU).

Streams are independent plugin instances (PluginProcessor.h:69-73: each instance owns its MyBuffer and
processes), so the batch shards embarrassingly: rank r owns the contiguous block
[shard_range(S, r, W)) for the whole run and keeps its per-stream state resident on its GPU.
No collective is needed on the data path.  When the caller holds the whole batch on one rank,
`scatter_streams` / `gather_streams` move it with torch.distributed point-to-point traffic
(backend "nccl" = RCCL over xGMI on a GPU node, "gloo" in the CPU tests): a root fan-out uses each
peer's direct link once, there is no reduction anywhere.
"""
import torch
import torch.distributed as dist


def shard_range(n_streams, rank, world):
    """Contiguous [lo, hi) of streams owned by `rank`; sizes differ by at most one (ragged batches)."""
    base, rem = divmod(int(n_streams), int(world))
    lo = rank * base + min(rank, rem)
    hi = lo + base + (1 if rank < rem else 0)
    return lo, hi


def shard_sizes(n_streams, world):
    return [shard_range(n_streams, r, world)[1] - shard_range(n_streams, r, world)[0] for r in range(world)]


def scatter_streams(x_root, n_streams, tail_shape, dtype, device, src=0, group=None, out=None, async_op=False):
    """Rank `src` holds x_root [n_streams, *tail_shape]; every rank returns its own shard (into `out` when given).
    async_op: returns (shard, works) without waiting -- `w.wait()` on each work before the shard is read (under RCCL
    that orders the current stream behind the transfer, it does not block the host): double-buffered exchanges.

    Point-to-point (no padding needed for ragged shards), issued as ONE batch_isend_irecv group: under RCCL every
    peer pair otherwise sets its channel up lazily, one after the other.  `src` is a rank OF THE GROUP."""
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    lo, hi = shard_range(n_streams, rank, world)
    mine = out if out is not None else torch.empty((hi - lo, *tail_shape), dtype=dtype, device=device)
    assert tuple(mine.shape) == (hi - lo, *tail_shape)
    ops = []
    if rank == src:
        for r in range(world):
            rlo, rhi = shard_range(n_streams, r, world)
            if r == src:
                mine.copy_(x_root[rlo:rhi])
            elif rhi > rlo:
                ops.append(dist.P2POp(dist.isend, x_root[rlo:rhi].contiguous(), _global_rank(group, r), group))
    elif hi > lo:
        ops.append(dist.P2POp(dist.irecv, mine, _global_rank(group, src), group))
    works = _run(ops, async_op)
    return (mine, works) if async_op else mine


def _global_rank(group, r):
    """isend/irecv/P2POp take GLOBAL ranks; shard ownership is by rank within the group."""
    return r if group is None else dist.get_global_rank(group, r)


def _run(ops, async_op=False):
    works = dist.batch_isend_irecv(ops) if ops else []
    if not async_op:
        for q in works:
            q.wait()
    return works


def gather_streams(y_local, n_streams, dst=0, group=None, out=None, async_op=False):
    """Inverse of scatter_streams: rank `dst` (of the group) returns [n_streams, *tail], the others None.
    `out` (optional, on `dst`): a preallocated [n_streams, *tail] tensor to receive into.
    async_op: returns (result, works) without waiting (see scatter_streams)."""
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    tail = tuple(y_local.shape[1:])
    if rank == dst:
        if out is None:
            out = torch.empty((n_streams, *tail), dtype=y_local.dtype, device=y_local.device)
        ops = []
        for r in range(world):
            rlo, rhi = shard_range(n_streams, r, world)
            if r == dst:
                out[rlo:rhi].copy_(y_local)
            elif rhi > rlo:
                ops.append(dist.P2POp(dist.irecv, out[rlo:rhi], _global_rank(group, r), group))
        works = _run(ops, async_op)
        return (out, works) if async_op else out
    lo, hi = shard_range(n_streams, rank, world)
    works = _run([dist.P2POp(dist.isend, y_local.contiguous(), _global_rank(group, dst), group)], async_op) if hi > lo else []
    return (None, works) if async_op else None


def exchange_steps(n_steps, n_streams, in_root, out_root, in_tail, out_tail, dtype, device, process, root=0, group=None):
    """SURVEY 8(e), double-buffered: per step the root fans block `in_root(i)` ([n_streams, *in_tail], root only) out to the
    ranks, every rank runs `process(in_shard, out_shard)` on its own shard, the root gathers [n_streams, *out_tail] into
    `out_root(i)` (root only; may return None to let the helper allocate).  Step i+1's scatter and step i-1's gather are in
    flight while step i is processed.  Returns the list of the root's output tensors of the last two steps (root) or None.
    Every rank must call it with the same n_steps.

    STREAM CONTRACT: `process` must enqueue its work on torch's CURRENT stream of `device` (pass
    `torch.cuda.current_stream(device).cuda_stream` to process_device): the double buffering is ordered against the kernels only
    through NCCL's implicit synchronisation with the current stream (`w.wait()` makes the current stream wait for the transfer,
    and a transfer is enqueued behind whatever the current stream holds).  A `process` that launches on a stream of its own
    races the scatter of step i + 1 into the other input buffer's twin and the reuse of its output buffer.  To run the whole
    exchange on another stream, call this function under `with torch.cuda.stream(that_stream):`."""
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    lo, hi = shard_range(n_streams, rank, world)
    inb = [torch.empty((hi - lo, *in_tail), dtype=dtype, device=device) for _ in range(2)]
    outb = [torch.empty((hi - lo, *out_tail), dtype=dtype, device=device) for _ in range(2)]
    wsc, wga, res = [None, None], [None, None], [None, None]

    def wait(ws):
        for w in ws or []:
            w.wait()

    if n_steps <= 0:
        return None
    _, wsc[0] = scatter_streams(in_root(0) if rank == root else None, n_streams, in_tail, dtype, device, src=root, group=group,
                                out=inb[0], async_op=True)
    for i in range(n_steps):
        cur = i & 1
        wait(wsc[cur])
        if i + 1 < n_steps:
            _, wsc[1 - cur] = scatter_streams(in_root(i + 1) if rank == root else None, n_streams, in_tail, dtype, device, src=root,
                                              group=group, out=inb[1 - cur], async_op=True)
        wait(wga[cur])                                # step i-2's gather has read outb[cur]
        process(inb[cur], outb[cur])
        res[cur], wga[cur] = gather_streams(outb[cur], n_streams, dst=root, group=group,
                                            out=out_root(i) if rank == root else None, async_op=True)
    wait(wga[0])
    wait(wga[1])
    return res if rank == root else None
