#!/usr/bin/env python3
"""Benchmark of the batched synthesis hot path (BASELINE.json metric).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload aac|mp3|vorbis|flac]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A "step" is one pass of the hot path over one batch of synthetic, HBM-resident input.  Default
workload = BASELINE config 2: AAC-LC 48 kHz stereo, 65 536 long-block frames (128 chains x 1024
frames, KBD windows), 1024-pt IMDCT + window + overlap-add.  Rank 0 prints ONE JSON line.

Multi-GPU: one process per GPU, chains sharded across ranks with NO data-path collective (streams are
independent) -- weak scaling: every rank runs the full per-GPU batch.  RCCL is used only
for the barrier / max-over-ranks timing.
"""
import argparse
import json
import os
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))

T_START = time.perf_counter()
HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6.3 TB/s is the measured copy ceiling


def log(msg):
    print("[bench %.1fs] %s" % (time.perf_counter() - T_START, msg), file=sys.stderr, flush=True)


def make_workload(name, torch, ctx, seed, scale=1.0, mix=0.0):
    """Returns (step_fn, units_per_step, unit_name, algorithmic_bytes_per_step, description, kernel_name, output tensor)."""
    import symphonia_amd as sa
    g = torch.Generator(device="cuda").manual_seed(seed)
    if name == "aac":
        nch, nfr = int(128 * scale), 1024  # 64 stereo streams x 1024 frames = 65 536 frames
        coeffs = torch.randn((nch, nfr, 1024), generator=g, device="cuda", dtype=torch.float32)
        coeffs *= torch.exp2(torch.randint(-8, 13, (nch, nfr, 64), generator=g, device="cuda").float()).repeat_interleave(16, dim=2)
        coeffs[:, :, 672:] = 0.0  # 48 kHz content: band-limited like a real encoder's output
        side = torch.full((nch, nfr), int(sa.aac_side(0, 1, 1)), dtype=torch.uint8, device="cuda")
        if mix > 0.0:  # development: a legal window-sequence walk with block switching (the headline is all ONLY_LONG)
            rng = np.random.default_rng(seed)
            sd = np.empty((nch, nfr), np.uint8)
            for c in range(nch):
                cur, prev_shape = 0, 1
                for t in range(nfr):
                    cur = (1 if rng.random() < mix else 0) if cur in (0, 3) else (2 if rng.random() < 0.5 else 3)
                    shape = int(rng.integers(0, 2))
                    sd[c, t] = int(sa.aac_side(cur, shape, prev_shape))
                    prev_shape = shape
            side = torch.from_numpy(sd).cuda()
        delay = torch.zeros((nch, 1024), device="cuda", dtype=torch.float32)
        pcm = torch.empty_like(coeffs)
        dsp = sa.AacDsp(ctx)

        def step():
            dsp.synth(coeffs, side, delay, pcm)
        frames = nch * nfr // 2
        return step, frames, "frames", nch * nfr * 8192, {
            "workload": "AAC-LC 48 kHz stereo, %d long-block frames (%d chains x %d), 1024-pt IMDCT+window+OLA, KBD"
                        % (frames, nch, nfr), "channel_frames": nch * nfr, "samples_per_frame": 1024}, "aac_synth_kernel", pcm
    if name == "mp3":
        nch, ngr = int(128 * scale), 2048  # 64 stereo streams x 2048 granules = 131 072 granules
        xr = torch.randn((nch, ngr, 576), generator=g, device="cuda", dtype=torch.float32) * 0.05
        bt, mx, rz = np.zeros((nch, ngr), np.uint8), np.zeros((nch, ngr), np.uint8), np.full((nch, ngr), 576)
        if mix > 0.0:  # development: Long -> Start -> Short... -> End walks (a quarter of the short runs mixed), random rzero
            rng = np.random.default_rng(seed)
            for c in range(nch):
                g = 0
                while g < ngr:
                    if rng.random() < mix and g + 3 < ngr:
                        run = int(rng.integers(1, 4))
                        mixed = rng.random() < 0.25
                        bt[c, g] = 1
                        bt[c, g + 1:g + 1 + run] = 2
                        mx[c, g + 1:g + 1 + run] = mixed
                        bt[c, min(g + 1 + run, ngr - 1)] = 3
                        g += run + 2
                    else:
                        g += 1
            rz = 2 * rng.integers(100, 289, (nch, ngr))
        side_np = sa.mp3_side(bt, mx, rz)
        side = torch.from_numpy(side_np.view(np.uint8).reshape(nch, ngr, 4)).cuda()
        st = [torch.zeros((nch, 576), device="cuda"), torch.zeros((nch, 1024), device="cuda"),
              torch.zeros(nch, dtype=torch.int32, device="cuda")]
        pcm = torch.empty_like(xr)
        syn = sa.Mp3Synthesis(ctx, 0)

        def step():
            syn.synth(xr, side, st[0], st[1], st[2], pcm)
        granules = nch * ngr // 2
        return step, granules, "granules", nch * ngr * 4608, {
            "workload": "MP3 Layer III 44.1 kHz stereo, %d long-block granules (%d chains x %d), hybrid synthesis + polyphase"
                        % (granules, nch, ngr), "granule_channels": nch * ngr, "samples_per_granule": 576}, "mp3_synth_kernel", pcm
    if name == "vorbis":
        nch, nb = int(64 * scale), 4096  # one GPU's shard of config 4: 8 streams x 8 ch x 4096 blocks
        rng = np.random.default_rng(seed)
        flags = np.zeros((nch, nb), np.uint8)
        cur = np.ones(nch, bool)
        for b in range(nb):  # Markov: P(long->long) = 0.9, P(short->short) = 0.7
            r = rng.random(nch)
            cur = np.where(cur, r < 0.9, r >= 0.7)
            flags[:, b] = cur
        v = sa.VorbisDsp(ctx, 8, 11)
        so, po = v.layout(flags, np.full(nch, -1))
        spec_stride, pcm_stride = int(so[:, -1].max()), int(po[:, -1].max())
        spectra = torch.randn((nch, spec_stride), generator=g, device="cuda", dtype=torch.float32) * 0.1
        d_flags = torch.from_numpy(flags).cuda()
        prev = torch.full((nch,), -1, dtype=torch.int32, device="cuda")
        overlap = torch.zeros((nch, 1024), device="cuda")
        pcm = torch.zeros((nch, pcm_stride), device="cuda")

        def step():
            prev.fill_(-1)
            v.synth(spectra, d_flags, prev, overlap, pcm_stride, pcm)
        bytes_alg = int(4 * (so[:, -1].sum() + po[:, -1].sum()))
        return step, nch * nb // 8, "frames", bytes_alg, {
            "workload": "Vorbis 2048/256 mixed block sizes, 8 ch, %d blocks (%d chains x %d)" % (nch * nb // 8, nch, nb),
            "channel_blocks": nch * nb}, "vorbis_synth_wave_kernel", pcm
    if name == "flac":
        nb, bs = int(262144 * scale), 4096  # 1/4 of config 5 per step (4 GiB in place; the full 1 M blocks = 16 GiB)
        buf = torch.randint(-(1 << 12), 1 << 12, (nb, bs), generator=g, device="cuda", dtype=torch.int32)
        desc_np = sa.flac_desc(np.full(nb, 2), np.full(nb, 32), np.full(nb, 12), np.zeros(nb))
        desc = torch.from_numpy(desc_np.view(np.uint8).reshape(nb, 4)).cuda()
        co = torch.randint(-40, 40, (nb, 32), generator=g, device="cuda", dtype=torch.int32)
        co[:, 0] = int(1.6 * 4096)
        co[:, 1] = int(-0.7 * 4096)
        fp = sa.FlacPredictor(ctx)

        def step():
            # The restore is in place, so every step after the first predicts over the previous step's
            # output.  That is still one full pass of the recurrence over the batch: the kernel has no
            # data-dependent control flow and the arithmetic wraps, so the time per pass is the same.
            fp.restore(buf, desc, co)
        return step, nb, "blocks", nb * bs * 8, {
            "workload": "FLAC 24-bit, LPC order 32 (15-bit coefficients, shift 12), %d subframe blocks of 4096 samples, "
                        "in place" % nb, "samples": nb * bs}, "flac_restore_f64_kernel", buf
    if name == "alac":
        nb, bs = int(262144 * scale), 4096  # 16-bit ALAC frames of 4096 samples, adaptive predictor of order 8
        buf = torch.randint(-(1 << 9), 1 << 9, (nb, bs), generator=g, device="cuda", dtype=torch.int32)
        desc_np = sa.alac_desc(np.zeros(nb), np.full(nb, 8), np.full(nb, 9), np.full(nb, 16))
        desc = torch.from_numpy(desc_np.view(np.uint8).reshape(nb, 4)).cuda()
        co = torch.randint(-200, 200, (nb, 32), generator=g, device="cuda", dtype=torch.int32)
        ap = sa.AlacPredictor(ctx)

        def step():
            ap.predict(buf, desc, co)  # in place, like the FLAC workload: every pass costs the same
        return step, nb, "blocks", nb * bs * 8, {
            "workload": "ALAC 16-bit, adaptive LPC order 8 (shift 9), %d element-channel blocks of 4096 samples, in place" % nb,
            "samples": nb * bs}, "alac_predict_kernel", buf
    raise ValueError(name)


def usable_cores():
    """Threads the CPU baseline uses: the affinity mask, capped by a cgroup CPU quota if one is set."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = Path("/sys/fs/cgroup/cpu.max").read_text().split()
        if quota != "max":
            n = max(1, min(n, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return n


def cpu_baseline(name, seconds=10.0):
    """The oracle (C restatement of the reference's scalar path) timed on the host cores on a bounded sample.
    Fan-out over cores happens inside oracle/bench_mt.c (pthreads, private outputs per thread)."""
    import oracle
    cores = usable_cores()
    rng = np.random.default_rng(0)
    if name == "aac":
        nch, nfr = 2, 128
        in0 = rng.standard_normal((nch, nfr, 1024)).astype(np.float32)
        in0[:, :, 672:] = 0.0
        in1 = np.full((nch, nfr), oracle.aac_side(0, 1, 1), np.uint8)
        kw = dict(n_chains=nch, per_chain=nfr)
        units, unit = nch * nfr / 2, "frames/s"
        sample = "%d channel-frames (2 chains x %d long blocks)" % (nch * nfr, nfr)
    elif name == "mp3":
        nch, ngr = 2, 128
        in0 = rng.standard_normal((nch, ngr, 576)).astype(np.float32)
        in1 = oracle.mp3_side(np.zeros((nch, ngr)), np.zeros((nch, ngr)), np.full((nch, ngr), 576))
        kw = dict(n_chains=nch, per_chain=ngr, p0=0)
        units, unit = nch * ngr / 2, "granules/s"
        sample = "%d granule-channels (2 chains x %d long granules)" % (nch * ngr, ngr)
    elif name == "vorbis":
        nch, nb = 8, 32
        in1 = np.ones((nch, nb), np.uint8)
        in0 = rng.standard_normal((nch, nb * 1024)).astype(np.float32)
        kw = dict(n_chains=nch, per_chain=nb, stride_in=nb * 1024, stride_out=nb * 1024, p0=8, p1=11)
        units, unit = nb, "frames/s"
        sample = "8 ch x %d long blocks" % nb
    elif name == "alac":
        nb, bs = 8, 4096
        in0 = rng.integers(-512, 512, (nb, bs)).astype(np.int32)
        in1 = oracle.alac_desc(np.zeros(nb), np.full(nb, 8), np.full(nb, 9), np.full(nb, 16))
        co = rng.integers(-200, 200, (nb, 32)).astype(np.int32)
        kw = dict(in2=co, n_chains=nb, per_chain=bs)
        units, unit = nb, "blocks/s"
        sample = "%d order-8 blocks of 4096 samples" % nb
    else:
        nb, bs = 8, 4096
        in0 = rng.integers(-4096, 4096, (nb, bs)).astype(np.int32)
        in1 = oracle.flac_desc(np.full(nb, 2), np.full(nb, 32), np.full(nb, 12), np.zeros(nb))
        co = rng.integers(-40, 40, (nb, 32)).astype(np.int32)
        kw = dict(in2=co, n_chains=nb, per_chain=bs)
        units, unit = nb, "blocks/s"
        sample = "%d order-32 blocks of 4096 samples" % nb
    dt1, reps1 = oracle.bench_mt(name, 1, 2.0, in0, in1, **kw)  # the reference is single-threaded (BENCHMARKS.md:5)
    dt, reps = oracle.bench_mt(name, cores, seconds, in0, in1, **kw)
    return {"value": units * reps / dt, "unit": unit, "cores": cores, "kind": "port",
            "single_thread_value": units * reps1 / dt1,
            "sample": "oracle/symoracle.c (scalar restatement of the reference's non-SIMD path, gcc -O2, no FMA): "
                      + sample + " per task; %d tasks on %d threads in %.1f s" % (reps, cores, dt)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="aac", choices=["aac", "mp3", "vorbis", "flac", "alac"])
    ap.add_argument("--segment", type=int, default=0, help="frames per wavefront segment (0 = library default)")
    ap.add_argument("--scale", type=float, default=1.0, help="batch size multiplier (development only)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--mix", dest="aac_mix", type=float, default=0.0,
                    help="development: probability of a block switch per long frame / granule in the AAC and MP3 workloads (headline: 0)")
    ap.add_argument("--gather", action="store_true",
                    help="N > 1 only: also time an RCCL all_gather of the PCM shards (reported beside, never inside, `value`)")
    args = ap.parse_args()
    import faulthandler
    faulthandler.enable()
    faulthandler.dump_traceback_later(170, repeat=True, file=sys.stderr)  # a hang leaves a stack in the log

    import torch
    import torch.distributed as dist
    import symphonia_amd as sa
    from symphonia_amd.sharding import max_over_ranks

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        sys.exit("bench.py needs an MI355X: symphonia_amd has no CPU path")
    # Development smoke test of the N > 1 control flow on a one-GPU box (SYM_BENCH_ONE_GPU_SMOKE=1): every rank shares
    # cuda:0 and the two tiny reductions go through gloo on the host.  Never set by the driver; its numbers mean nothing.
    one_gpu_smoke = os.environ.get("SYM_BENCH_ONE_GPU_SMOKE") == "1"
    if one_gpu_smoke:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    if world > 1:
        if one_gpu_smoke:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    ctx = sa.Context(local_rank)
    ctx.use_torch_stream()
    if args.segment:
        ctx.set_segment(args.segment)
    step, units, unit_name, alg_bytes, config, kernel, result = make_workload(args.workload, torch, ctx, 1234 + rank, args.scale, args.aac_mix)

    log("workload built: %s" % config["workload"])
    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    log("warmup done")
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    # One HIP event pair around the K launches (on the launch stream; the context uses torch's current stream).
    # Per-step event pairs would put ~40 us of signal traffic between consecutive launches.
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    ev0.record()
    for i in range(args.steps):
        step()
    ev1.record()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    elapsed = max_over_ranks(elapsed, dist if world > 1 else None, device="cpu" if one_gpu_smoke else "cuda")
    launch_s = ev0.elapsed_time(ev1) / 1e3 / args.steps  # mean launch period of the hot-path kernel(s)
    log("timed region done: %.3f ms/step (device), %.3f ms/step (wall)" % (launch_s * 1e3, elapsed / args.steps * 1e3))

    gather = None
    if args.gather and world > 1:
        from symphonia_amd.sharding import timed_all_gather
        secs, nbytes = timed_all_gather(result, dist)
        gather = {"op": "all_gather of every rank's PCM shard (RCCL over xGMI)", "ms": secs * 1e3, "bytes_per_rank": nbytes,
                  "algbw_GBps": nbytes * world / secs / 1e9}

    if rank == 0:
        achieved = alg_bytes / launch_s / 1e9
        out = {
            "metric": "decoded audio frames/sec (batch IMDCT+OLA)" if args.workload == "aac" else
                      "decoded %s/sec (%s synthesis)" % (unit_name, args.workload),
            "value": units * world * args.steps / elapsed,
            "unit": unit_name + "/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "i32/i64" if args.workload == "flac" else ("i32" if args.workload == "alac" else "f32"),
            "data": "synthetic",
            "config": dict(config, parallelism="chains sharded per GPU, no collective", segment=args.segment or "auto"),
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": None, "kernel": kernel,
                         "kernel_ms": launch_s * 1e3, "algorithmic_bytes_per_launch": alg_bytes},
        }
        try:  # HBM bytes per launch from the committed rocprofv3 PMC passes (profiles/hbm_traffic.json), if measured
            tr = json.loads((ROOT / "profiles" / "hbm_traffic.json").read_text()).get(args.workload)
            if tr and tr["algorithmic_bytes_per_launch"] == alg_bytes:
                out["roofline"]["traffic"] = tr["bytes_per_launch"]
                out["roofline"]["traffic_source"] = tr["source"]
        except (OSError, ValueError, KeyError):
            pass
        out["roofline"]["copy_ceiling_note"] = ("a plain copy of the same 1:1 read/write footprint reaches 5.1-5.9 TB/s on "
                                                "this part (profiles/r01_ubench_hbm_copy.txt); peak = HBM3E spec")
        if args.workload == "alac":
            out["roofline"]["note"] = "integer-ALU bound (adaptive predictor, ~13 x order operations per sample), not HBM"
        if args.workload == "flac":
            out["roofline"]["note"] = "FP64-FMA-issue bound (32 exact FMAs per 8 B), not HBM (DESIGN.md 4.5)"
        if gather:
            out["collective"] = gather
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(args.workload)
        print(json.dumps(out))
    ctx.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
