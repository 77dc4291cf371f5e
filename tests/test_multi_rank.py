"""N>1 path on CPU: two gloo ranks shard a batch by chain (no data-path collective), each runs its shard
through the C ABI (the TEST-ONLY CPU emulation build of the kernels stands in for the GPU here), and the
gathered PCM must equal the oracle on the whole batch.  Also checks the bench clock (max over ranks)."""
import os
import sys
from pathlib import Path

import numpy as np
import pytest

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "tests"))

from symphonia_amd.sharding import shard_chains, shard_streams  # noqa: E402


def test_shard_arithmetic():
    for n, w in ((64, 8), (7, 2), (3, 4), (0, 2), (65, 8)):
        ranges = [shard_streams(n, w, r) for r in range(w)]
        assert ranges[0][0] == 0 and ranges[-1][1] == n
        assert all(a[1] == b[0] for a, b in zip(ranges, ranges[1:]))
        sizes = [e - b for b, e in ranges]
        assert max(sizes) - min(sizes) <= 1
    assert shard_chains(128, 2, 8, 3) == (48, 64)
    with pytest.raises(ValueError):
        shard_chains(7, 2, 2, 0)
    with pytest.raises(ValueError):
        shard_streams(4, 2, 2)


def _worker(rank, world, port, q):
    try:
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
        import torch
        import torch.distributed as dist
        dist.init_process_group("gloo", rank=rank, world_size=world)
        import oracle
        from emu_lib import emu_library
        from helpers import aac_spectra
        from symphonia_amd import AacDsp, Context
        from symphonia_amd.sharding import gather_chains, max_over_ranks, shard_chains
        n_chains, nfr = 6, 5  # 3 stereo streams over 2 ranks: 2 + 1
        rng = np.random.default_rng(5)  # every rank builds the same global batch, then takes its shard
        coeffs = aac_spectra(rng, (n_chains, nfr))
        side = np.full((n_chains, nfr), oracle.aac_side(0, 1, 1), np.uint8)
        delay = rng.standard_normal((n_chains, 1024)).astype(np.float32)
        b, e = shard_chains(n_chains, 2, world, rank)
        ctx = Context(0, library=emu_library())
        pcm, new_delay = AacDsp(ctx).synth(coeffs[b:e], side[b:e], delay[b:e])
        ctx.close()
        full = gather_chains(torch.from_numpy(pcm), n_chains, 2, dist).numpy()
        full_delay = gather_chains(torch.from_numpy(new_delay), n_chains, 2, dist).numpy()
        want, want_delay = oracle.aac_synth(coeffs, side, delay)
        ok = np.array_equal(full, want) and np.array_equal(full_delay, want_delay)
        from symphonia_amd.sharding import timed_all_gather
        secs, nbytes = timed_all_gather(torch.from_numpy(pcm[:2].copy()), dist, reps=2)
        ok = ok and secs > 0 and nbytes == pcm[:2].nbytes
        slowest = max_over_ranks(1.0 + rank, dist)
        dist.barrier()
        dist.destroy_process_group()
        q.put((rank, bool(ok), slowest, (b, e)))
    except Exception as ex:  # noqa: BLE001
        q.put((rank, False, repr(ex), None))


def test_two_ranks_shard_by_chain_gloo():
    import torch.multiprocessing as mp
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mpctx = mp.get_context("spawn")
    q = mpctx.Queue()
    procs = [mpctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    results = sorted(q.get(timeout=240) for _ in procs)
    for p in procs:
        p.join(timeout=60)
    assert [r[1] for r in results] == [True, True], results
    assert [r[2] for r in results] == [2.0, 2.0]  # max over ranks of (1.0, 2.0)
    assert [r[3] for r in results] == [(0, 4), (4, 6)]
