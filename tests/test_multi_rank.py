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

from symphonia_amd.sharding import local_pairs, shard_chains, shard_streams  # noqa: E402


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


def test_channel_pairs_follow_their_streams():
    """The joint-stereo stages take (left chain, right chain) pairs: sharding by whole streams keeps every pair on one
    rank, and local_pairs re-indexes them (and says which descriptor rows go with them)."""
    pairs = [(2 * s, 2 * s + 1) for s in range(7)]  # 7 stereo streams = 14 chains
    seen = []
    for r in range(3):
        b, e = shard_chains(14, 2, 3, r)
        rows, local = local_pairs(pairs, b, e)
        assert all(0 <= c0 < e - b and c1 == c0 + 1 for c0, c1 in local)
        assert [pairs[i][0] - b for i in rows] == [c0 for c0, _ in local]
        seen += rows
    assert sorted(seen) == list(range(7))
    with pytest.raises(ValueError):
        local_pairs([(3, 4)], 0, 4)  # a pair cut by the shard boundary


def test_two_ranks_shard_the_mp3_front_stage():
    """requantize + stereo on two 'ranks' (shards run one after the other through the CPU emulation): the union of the
    shards equals the whole batch -- no rank needs anything from another."""
    from emu_lib import emu_library
    from symphonia_amd import Context, Mp3Stereo
    from test_mp3_stereo import fused_case
    q, rd, pairs, sd, want, _ = fused_case(5, 0, 3, 4)          # 7 chains, 3 pairs in scrambled order
    flat = pairs.reshape(-1)                                    # lay the batch out stream by stream: (L, R) adjacent
    q2, rd2, want2 = q[flat], rd[flat], want[flat]              # chains 0..5 = pair 0 L, R, pair 1 L, R, ...
    pairs2 = [(2 * p, 2 * p + 1) for p in range(3)]
    with Context(0, library=emu_library()) as ctx:
        for r in range(2):
            b, e = shard_chains(6, 2, 2, r)
            rows, local = local_pairs(pairs2, b, e)
            xr = np.zeros((e - b,) + q2.shape[1:], np.float32)
            Mp3Stereo(ctx, 0).requantize_stereo(np.ascontiguousarray(q2[b:e]), np.ascontiguousarray(rd2[b:e]),
                                                np.array(local, np.int32), np.ascontiguousarray(sd[rows]), xr)
            assert np.array_equal(xr.view(np.uint32), want2[b:e].view(np.uint32)), r


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


def _bench(args, env_extra=None, timeout=900):
    import json
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(env_extra or {})
    r = subprocess.run([sys.executable, str(ROOT / "bench.py")] + args, capture_output=True, text=True, timeout=timeout, env=env,
                       cwd=str(ROOT))
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    return r, (json.loads(lines[-1]) if lines else None)


def test_bench_gpus_2_starts_two_ranks_by_itself():
    """`python bench.py --gpus 2` with no launcher in the environment (how the driver calls it) must start two ranks and
    report a two-rank line.  Runs the real control flow -- self-launch under torch.distributed.run on 127.0.0.1, rank /
    device checks, barrier + max-over-ranks clock, scatter -> step -> gather leg, config-4 line -- on CPU tensors through
    the emulation build with gloo (--emulate, test only)."""
    r, d = _bench(["--gpus", "2", "--emulate", "--steps", "2", "--warmup", "1"])
    assert r.returncode == 0 and d is not None, r.stderr[-3000:]
    assert d["n_gpus"] == 2 and d["steps"] == 2 and d["scaling"] == "weak"
    assert d["ranks"]["world_size"] == 2 and d["ranks"]["backend"] == "gloo"
    assert len(set(d["ranks"]["devices"])) == 2 and len(d["ranks"]["ms_per_step_per_rank"]) == 2
    assert d["ms_per_step"] * 1.001 >= max(d["ranks"]["ms_per_step_per_rank"])  # the clock is the slowest rank's
    # value = units of ALL ranks / max-over-ranks time
    frames_per_rank = d["config"]["channel_frames"] / 2
    assert abs(d["value"] - 2 * frames_per_rank * d["steps"] / (d["ms_per_step"] * d["steps"] / 1e3)) < 1e-6 * d["value"]
    ex = d["exchange"]
    assert set(ex["ms"]) == {"scatter", "step", "gather", "total"} and ex["ms"]["total"] >= ex["ms"]["step"]
    assert ex["value_inclusive"] > 0
    assert d["config4_vorbis"]["value"] > 0 and "Vorbis" in d["config4_vorbis"]["config"]["workload"]
    assert "cpu_baseline" not in d  # rank 0 at N = 1 only


def test_bench_refuses_a_launcher_with_the_wrong_world_size():
    r, d = _bench(["--gpus", "2", "--emulate", "--steps", "1", "--warmup", "0"], {"WORLD_SIZE": "1", "RANK": "0", "LOCAL_RANK": "0"},
                  timeout=300)
    assert r.returncode != 0 and d is None
    assert "one rank per GPU" in r.stderr


def test_bench_line_carries_the_other_workloads():
    """The default N = 1 line reports BASELINE configs 3, 4 (one GPU's shard), 5 and the ALAC predictor beside the headline
    (`other_workloads`, 20 / 8 steps each, each in its own try): control flow through the emulation build at a tiny scale."""
    r, d = _bench(["--emulate", "--scale", "0.0001", "--steps", "2", "--warmup", "1"], timeout=600)
    assert r.returncode == 0 and d is not None, r.stderr[-3000:]
    ow = d["other_workloads"]
    assert set(ow) == {"mp3", "vorbis", "flac", "alac", "flac_padded_rows", "alac_padded_rows", "aac_mix_0.05", "aac_mix_0.25", "mp3_mix_0.06", "mp3_int16_one_kernel", "mp3_int16_two_kernels",
                       "vorbis_posts_byte_plane", "vorbis_posts_f32_spectrum",
                       "aac_joint_stereo_on_load", "aac_joint_stereo_two_kernels", "aac_tns_0.30"}
    for name, line in ow.items():
        assert "error" not in line, (name, line)
        assert line["steps"] == (8 if name in ("flac", "alac", "flac_padded_rows", "alac_padded_rows") else 20) and line["value"] > 0 and line["algorithmic_bytes_per_launch"] > 0
        if name.startswith(("aac", "mp3", "vorbis", "flac", "alac")) and "int16" not in name and "two_kernels" not in name:  # the timed batch itself is compared with the oracle (sampled chains / blocks), mixes included
            assert line["verified"]["mismatches"] == 0 and line["verified"]["samples_compared"] > 0
        if "mix" in name:
            assert abs(sum(v for k, v in line["mix"].items() if k not in ("p_switch", "mixed_share_of_short")) - 1.0) < 1e-9
        assert abs(line["roofline_frac"] - line["algorithmic_bytes_per_launch"] / (line["kernel_ms"] / 1e3) / 1e9 / 8000.0) < 1e-9
    assert d["verified"]["mismatches"] == 0 and d["schema"] == 5 and "protocol" in d and (d.get("repeats") is None or d["repeats"]["regions"] >= 5)
    assert ow["flac"]["kernel"] == "flac_restore_f64_kernel" and ow["mp3"]["kernel"] == "mp3_synth_kernel"
    assert "rows 4608 words apart" in ow["flac_padded_rows"]["workload"] and "rows 4608 words apart" in ow["alac_padded_rows"]["workload"]
