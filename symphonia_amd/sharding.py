"""Batch sharding across GPUs (one process per GPU).

Chains (one channel of one stream) are independent, so the batch is split by chain with no data-path
collective; the channels of one stream stay on one rank so a decoder adapter sees whole frames.
`torch.distributed` (RCCL on the GPU box, gloo in the CPU tests) is used only for barriers, the
max-over-ranks timing and -- when a caller really wants the PCM in one place -- an all_gather.
"""


def shard_streams(n_streams, world_size, rank):
    """Contiguous, balanced [begin, end) range of streams for `rank` (first ranks take the remainder)."""
    if world_size < 1 or not 0 <= rank < world_size:
        raise ValueError("rank %d out of range for world size %d" % (rank, world_size))
    base, extra = divmod(int(n_streams), world_size)
    begin = rank * base + min(rank, extra)
    return begin, begin + base + (1 if rank < extra else 0)


def shard_chains(n_chains, channels_per_stream, world_size, rank):
    """[begin, end) range of chains for `rank`, aligned to whole streams."""
    if n_chains % channels_per_stream:
        raise ValueError("n_chains must be a multiple of channels_per_stream")
    b, e = shard_streams(n_chains // channels_per_stream, world_size, rank)
    return b * channels_per_stream, e * channels_per_stream


def local_pairs(pair_chains, chain_begin, chain_end):
    """The channel pairs (rows of pair_chains, global chain indices) that live on a rank owning chains
    [chain_begin, chain_end), re-indexed to that rank's local chain numbers, plus their row indices in the global
    list (to slice the per-pair descriptors).  shard_chains keeps the channels of a stream together, so a pair is
    either wholly inside or wholly outside the range; anything else is a caller error."""
    rows, local = [], []
    for i, (c0, c1) in enumerate(pair_chains):
        inside = (chain_begin <= c0 < chain_end, chain_begin <= c1 < chain_end)
        if inside[0] != inside[1]:
            raise ValueError("pair (%d, %d) straddles the shard [%d, %d)" % (c0, c1, chain_begin, chain_end))
        if inside[0]:
            rows.append(i)
            local.append((int(c0) - chain_begin, int(c1) - chain_begin))
    return rows, local


def max_over_ranks(seconds, dist=None, device=None):
    """The bench clock: the slowest rank's time (bench.py contract)."""
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return float(seconds)
    import torch
    t = torch.tensor([float(seconds)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def gather_chains(local, n_chains, channels_per_stream, dist):
    """all_gather of chain-major shards (rows = chains) back into the full batch order, on every rank."""
    import torch
    world = dist.get_world_size()
    sizes = [shard_chains(n_chains, channels_per_stream, world, r) for r in range(world)]
    rows = max(e - b for b, e in sizes)
    pad = torch.zeros((rows,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    pad[: local.shape[0]] = local
    out = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(out, pad)
    return torch.cat([o[: e - b] for o, (b, e) in zip(out, sizes)], dim=0)


def timed_all_gather(local, dist, reps=3):
    """Optional collection step of a sharded batch: all_gather of every rank's (equal-shape) PCM shard.  Returns
    (seconds per all_gather, max over ranks; bytes contributed per rank).  This is the only data-path collective the
    backend ever needs, and only when a caller wants every shard in one place (BASELINE config 4's "split / gather")."""
    import time
    import torch
    world = dist.get_world_size()
    out = [torch.empty_like(local) for _ in range(world)]
    dist.all_gather(out, local)  # warm-up (connection set-up)
    if local.is_cuda:
        torch.cuda.synchronize()
    dist.barrier()
    t0 = time.perf_counter()
    for _ in range(reps):
        dist.all_gather(out, local)
    if local.is_cuda:
        torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    return max_over_ranks(dt, dist, device=local.device if local.is_cuda else None), local.numel() * local.element_size()


def timed_exchange(local_in, local_out, step, dist, sync, device=None, reps=3):
    """a batch that starts and ends on rank 0.  Per repetition: rank 0 scatters one input shard to every
    rank (dist.scatter: point-to-point sends over xGMI under RCCL), every rank runs `step` on its shard (in `local_in`,
    result in `local_out`), rank 0 gathers the PCM shards (dist.gather).  Returns seconds per repetition, max over
    ranks, as {"scatter", "step", "gather", "total"}.  One-to-all traffic is bounded by rank 0's links, which is why this
    is never the headline number."""
    import time
    import torch
    world, rank = dist.get_world_size(), dist.get_rank()
    src = [torch.empty_like(local_in).copy_(local_in) for _ in range(world)] if rank == 0 else None
    dst = [torch.empty_like(local_out) for _ in range(world)] if rank == 0 else None

    def once():
        t = [time.perf_counter()]
        dist.scatter(local_in, src, src=0)
        sync()
        t.append(time.perf_counter())
        step()
        sync()
        t.append(time.perf_counter())
        dist.gather(local_out, dst, dst=0)
        sync()
        t.append(time.perf_counter())
        return t

    once()  # warm-up: connection set-up
    acc = [0.0, 0.0, 0.0, 0.0]
    for _ in range(reps):
        dist.barrier()
        sync()
        t = once()
        for i in range(3):
            acc[i] += t[i + 1] - t[i]
        acc[3] += t[3] - t[0]
    names = ("scatter", "step", "gather", "total")
    return {n: max_over_ranks(a / reps, dist, device=device) for n, a in zip(names, acc)}
