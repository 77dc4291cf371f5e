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


def make_workload(name, torch, ctx, seed, scale=1.0, mix=0.0, emulate=False):
    """Returns (step_fn, units_per_step, unit_name, algorithmic_bytes_per_step, description, kernel_name, output tensor)."""
    def sync_dev():
        if not emulate:
            torch.cuda.synchronize()

    import symphonia_amd as sa
    dev = "cpu" if emulate else "cuda"
    g = torch.Generator(device=dev).manual_seed(seed)
    if name == "aac":
        nch, nfr = max(2, int(128 * scale)), (6 if emulate else 1024)  # 64 stereo streams x 1024 frames = 65 536 frames
        coeffs = torch.randn((nch, nfr, 1024), generator=g, device=dev, dtype=torch.float32)
        coeffs *= torch.exp2(torch.randint(-8, 13, (nch, nfr, 64), generator=g, device=dev).float()).repeat_interleave(16, dim=2)
        coeffs[:, :, 672:] = 0.0  # 48 kHz content: band-limited like a real encoder's output
        side = torch.full((nch, nfr), int(sa.aac_side(0, 1, 1)), dtype=torch.uint8, device=dev)
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
            side = torch.from_numpy(sd).to(dev)
        delay = [torch.zeros((nch, 1024), device=dev, dtype=torch.float32) for _ in range(2)]  # ping-pong state buffers
        pcm = torch.empty_like(coeffs)
        dsp = sa.AacDsp(ctx)

        def step():
            dsp.synth(coeffs, side, delay[0], pcm, delay_out=delay[1])  # one launch; the next call continues from delay[1]
            delay.reverse()
        step.input = coeffs
        step.aac = (coeffs, side)

        def chunk_step(first, count):  # chains [first, first + count) from a zero state (symaccel_exchange_pipelined's callback)
            dsp.synth(coeffs[first:first + count], side[first:first + count], torch.zeros((count, 1024), device=dev), pcm[first:first + count],
                      delay_out=delay[1][first:first + count])
        step.chunk_step = chunk_step

        def verify_step():
            z = torch.zeros_like(delay[0])
            dsp.synth(coeffs, side, z, pcm, delay_out=delay[1])
            return pcm
        step.verify_step = verify_step
        frames = nch * nfr // 2
        cfg = {"workload": "AAC-LC 48 kHz stereo, %d long-block frames (%d chains x %d), 1024-pt IMDCT+window+OLA, KBD"
                           % (frames, nch, nfr), "channel_frames": nch * nfr, "samples_per_frame": 1024}
        if mix > 0.0:
            hist = np.bincount((sd & 3).ravel(), minlength=4) / float(sd.size)
            cfg["workload"] = ("AAC-LC 48 kHz stereo, %d frames (%d chains x %d) on legal window-sequence walks, random window shapes, "
                               "1024-pt / 8 x 128-pt IMDCT+window+OLA" % (frames, nch, nfr))
            cfg["mix"] = {"p_switch": mix, "only_long": hist[0], "long_start": hist[1], "eight_short": hist[2], "long_stop": hist[3]}
        return step, frames, "frames", nch * nfr * 8192, cfg, "aac_synth_quad_kernel", pcm
    if name == "mp3":
        nch, ngr = max(2, int(128 * scale)), (6 if emulate else 2048)  # 64 stereo streams x 2048 granules = 131 072 granules
        xr = torch.randn((nch, ngr, 576), generator=g, device=dev, dtype=torch.float32) * 0.05
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
        side = torch.from_numpy(side_np.view(np.uint8).reshape(nch, ngr, 4)).to(dev)
        st = [[torch.zeros((nch, 576), device=dev), torch.zeros((nch, 1024), device=dev),
               torch.zeros(nch, dtype=torch.int32, device=dev)] for _ in range(2)]  # ping-pong state buffers
        pcm = torch.empty_like(xr)
        syn = sa.Mp3Synthesis(ctx, 0)

        def step():
            syn.synth(xr, side, st[0][0], st[0][1], st[0][2], pcm, state_out=st[1])
            st.reverse()
        step.input = xr
        step.mp3 = (xr, side)

        def verify_step():
            z = [torch.zeros_like(t) for t in st[0]]
            syn.synth(xr, side, z[0], z[1], z[2], pcm, state_out=st[1])
            return pcm
        step.verify_step = verify_step
        granules = nch * ngr // 2
        cfg = {"workload": "MP3 Layer III 44.1 kHz stereo, %d long-block granules (%d chains x %d), hybrid synthesis + polyphase"
                           % (granules, nch, ngr), "granule_channels": nch * ngr, "samples_per_granule": 576}
        if mix > 0.0:
            hist = np.bincount(bt.ravel(), minlength=4) / float(bt.size)
            cfg["workload"] = ("MP3 Layer III 44.1 kHz stereo, %d granules (%d chains x %d) on Long / Start -> Short.. -> End walks, a quarter of "
                               "the short runs mixed, rzero uniform in 200..576, hybrid synthesis + polyphase" % (granules, nch, ngr))
            cfg["mix"] = {"p_switch": mix, "long": hist[0], "start": hist[1], "short": hist[2], "end": hist[3],
                          "mixed_share_of_short": float(mx[bt == 2].mean()) if (bt == 2).any() else 0.0}
        return step, granules, "granules", nch * ngr * 4608, cfg, "mp3_synth_kernel", pcm
    if name in ("aacjs", "aacjs2"):
        # config 2 from what the spectrum decoder produces for mid/side-coded stereo: coded spectra + the pairs' stereo maps -> PCM.
        # aacjs: joint stereo on load inside the synthesis kernel (one launch + the map expansion); aacjs2: aac_joint_stereo_kernel
        # in place, then the synthesis (round 3's two passes; the coded spectra are rewritten every step, their magnitudes drift --
        # timing only).
        nch, nfr = max(2, int(128 * scale)) & ~1, (6 if emulate else 1024)
        coeffs = torch.randn((nch, nfr, 1024), generator=g, device=dev, dtype=torch.float32)
        coeffs *= torch.exp2(torch.randint(-8, 13, (nch, nfr, 64), generator=g, device=dev).float()).repeat_interleave(16, dim=2)
        coeffs[:, :, 672:] = 0.0
        side = torch.full((nch, nfr), int(sa.aac_side(0, 1, 1)), dtype=torch.uint8, device=dev)
        swb_long = [0, 4, 8, 12, 16, 20, 24, 28, 32, 36, 40, 48, 56, 64, 72, 80, 88, 96, 108, 120, 132, 144, 160, 176, 196, 216, 240, 264, 292, 320,
                    352, 384, 416, 448, 480, 512, 544, 576, 608, 640, 672, 704, 736, 768, 800, 832, 864, 896, 928, 1024]  # 44.1 / 48 kHz
        swb_short = [0, 4, 8, 12, 16, 20, 28, 36, 44, 56, 68, 80, 96, 112, 128]
        rng = np.random.default_rng(seed)
        n_pairs = nch // 2
        desc = np.zeros((n_pairs, nfr), sa.AAC_JS_DTYPE)
        desc["num_windows"], desc["max_sfb"] = 1, 40  # bands up to line 672
        desc["mode"] = rng.choice([0, 1, 1, 1, 1, 1, 1, 2, 2, 0], (n_pairs, nfr, 128)).astype(np.uint8)  # 60 % mid/side, 20 % intensity
        desc["scale"] = (rng.standard_normal((n_pairs, nfr, 128)) * 0.5).astype(np.float32)
        d_desc = torch.from_numpy(desc.view(np.uint8).reshape(n_pairs, nfr, 644)).to(dev)
        pairs = np.arange(nch, dtype=np.int32).reshape(n_pairs, 2)
        d_pairs = torch.from_numpy(pairs).to(dev)
        delay = [torch.zeros((nch, 1024), device=dev, dtype=torch.float32) for _ in range(2)]
        pcm = torch.empty_like(coeffs)
        tools, dsp = sa.AacSpectralTools(ctx, swb_long, swb_short), sa.AacDsp(ctx)

        def step():
            if name == "aacjs":
                tools.synth_joint_stereo(coeffs, side, delay[0], d_pairs, d_desc, pcm, delay_out=delay[1])
            else:
                tools.joint_stereo(coeffs, d_pairs, d_desc)
                dsp.synth(coeffs, side, delay[0], pcm, delay_out=delay[1])
            delay.reverse()
        step.input = coeffs
        pristine = coeffs.clone() if name == "aacjs2" else coeffs  # (aacjs2 decodes in place: its check runs on a copy of the coded spectra)

        def verify():
            import oracle
            z = torch.zeros_like(delay[0])
            if name == "aacjs":
                tools.synth_joint_stereo(coeffs, side, z, d_pairs, d_desc, pcm, delay_out=delay[1])
            else:
                x = pristine.clone()
                tools.joint_stereo(x, d_pairs, d_desc)
                dsp.synth(x, side, z, pcm, delay_out=delay[1])
                del x
            sync_dev()
            bad = checked = 0
            for p_ in sorted({0, n_pairs - 1}):
                l, r = int(pairs[p_, 0]), int(pairs[p_, 1])
                fr = slice(0, min(nfr, 24))
                cl, cr = pristine[l, fr].cpu().numpy(), pristine[r, fr].cpu().numpy()
                dl, dr = cl.copy(), cr.copy()
                for f in range(cl.shape[0]):
                    dl[f], dr[f] = oracle.aac_joint_stereo(cl[f], cr[f], 1, 40, swb_long, desc[p_, f]["mode"], desc[p_, f]["scale"])
                want, _ = oracle.aac_synth(np.stack([dl, dr]), side[[l, r], fr].cpu().numpy(), np.zeros((2, 1024), np.float32))
                got = pcm[[l, r], fr].cpu().numpy()
                bad += int((got != want).sum())
                checked += got.size
            if bad:
                raise RuntimeError("bench: the %s batch differs from the oracle in %d of %d sampled samples" % (name, bad, checked))
            return {"checker": "oracle/symoracle.c (joint stereo, then Dsp::synth), outside the timed region" + ("" if name == "aacjs" else ", on a pristine copy of the spectra"),
                    "pairs": sorted({0, n_pairs - 1}), "frame_windows": [[0, min(nfr, 24)]], "samples_compared": checked, "mismatches": bad,
                    "criterion": "bit-identical f32 (value comparison)"}
        step.verify = verify
        frames = nch * nfr // 2
        bytes_alg = nch * nfr * 8192 + n_pairs * nfr * 644
        return step, frames, "frames", bytes_alg, {
            "workload": "AAC-LC 48 kHz stereo, %d long-block frames (%d chains x %d) from mid/side- and intensity-coded spectra (60 %% / 20 %% of the bands): %s"
                        % (frames, nch, nfr, "joint stereo on load in the synthesis kernel" if name == "aacjs" else "joint-stereo kernel in place, then synthesis"),
            "channel_frames": nch * nfr}, "aac_synth_quad_kernel<true>" if name == "aacjs" else "aac_joint_stereo_kernel + aac_synth_quad_kernel", pcm
    if name == "aactns":
        # config 2 as a REAL joint-stereo stream with temporal noise shaping looks: coded spectra + the pairs' stereo maps + TNS filters
        # -> PCM, the device-resident kernel sequence of symaccel_aac_decode_pipelined / SYMACCEL_BATCH_AAC_DECODE (ics/mod.rs:449-468,
        # cpe.rs:110-157, ics/tns.rs:149-199): a LIST pass decodes the joint stereo of the pair frames that carry a filter in place,
        # the filters run (one lane per filter: the recurrence is serial along the spectrum), then ONE walk decodes the joint stereo
        # of every other frame on load and synthesises all of them.  30 % of the channel frames carry one order-12 filter over
        # lines 160..672; 60 % / 20 % of the bands mid/side / intensity.  The list pass and the filters work in place, so the
        # spectra drift from step to step (timing only, like aacjs2); `verify` runs the sequence once on a pristine copy.
        nch, nfr = max(2, int(128 * scale)) & ~1, (6 if emulate else 1024)
        coeffs = torch.randn((nch, nfr, 1024), generator=g, device=dev, dtype=torch.float32)
        coeffs *= torch.exp2(torch.randint(-8, 13, (nch, nfr, 64), generator=g, device=dev).float()).repeat_interleave(16, dim=2)
        coeffs[:, :, 672:] = 0.0
        pristine = coeffs.clone()
        side = torch.full((nch, nfr), int(sa.aac_side(0, 1, 1)), dtype=torch.uint8, device=dev)
        swb_long = [0, 4, 8, 12, 16, 20, 24, 28, 32, 36, 40, 48, 56, 64, 72, 80, 88, 96, 108, 120, 132, 144, 160, 176, 196, 216, 240, 264, 292, 320,
                    352, 384, 416, 448, 480, 512, 544, 576, 608, 640, 672, 704, 736, 768, 800, 832, 864, 896, 928, 1024]
        swb_short = [0, 4, 8, 12, 16, 20, 28, 36, 44, 56, 68, 80, 96, 112, 128]
        rng = np.random.default_rng(seed)
        n_pairs = nch // 2
        desc = np.zeros((n_pairs, nfr), sa.AAC_JS_DTYPE)
        desc["num_windows"], desc["max_sfb"] = 1, 40
        desc["mode"] = rng.choice([0, 1, 1, 1, 1, 1, 1, 2, 2, 0], (n_pairs, nfr, 128)).astype(np.uint8)
        desc["scale"] = (rng.standard_normal((n_pairs, nfr, 128)) * 0.5).astype(np.float32)
        has = rng.random((nch, nfr)) < 0.30
        cf = np.argwhere(has)
        filt = np.zeros(len(cf), sa.AAC_TNS_DTYPE)
        filt["frame"] = (cf[:, 0] * nfr + cf[:, 1]).astype(np.uint32)
        filt["start"], filt["end"], filt["order"] = 160, 672, 12
        if os.environ.get("SYM_BENCH_TNS_RANGE") == "staggered":  # (measurement: 512-line ranges at 32 different offsets -- is the pass bound by every filter
            st = 16 * rng.integers(0, 32, len(cf))                 # touching the same offset of its 4 KiB frame at the same time?  Not verified: the spectra are
            filt["start"], filt["end"] = st, st + 512              # zero from line 672 on and the walk's joint-stereo bands assume it)
        filt["direction"] = rng.integers(0, 2, len(cf)).astype(np.uint8)
        filt["lpc"][:, :12] = (rng.integers(-4, 5, (len(cf), 12)) * 0.05 * 0.8 ** np.arange(12)).astype(np.float32)
        tns_order = os.environ.get("SYM_BENCH_TNS_ORDER", "shuffled")  # "stream": chain-major, frames ascending -- what a decoder hands over
        if tns_order != "stream":
            filt = filt[rng.permutation(len(filt))]
        pair_has = has[0::2] | has[1::2]                                   # pair p = chains (2p, 2p + 1)
        pf = np.flatnonzero(pair_has.ravel()).astype(np.uint32)            # pair * nfr + frame
        desc_walk = desc.copy()
        desc_walk["mode"][pair_has] = 0                                    # what the list pass leaves for the walk (launch_aac_js_consume)
        d_desc = torch.from_numpy(desc.view(np.uint8).reshape(n_pairs, nfr, 644)).to(dev)
        d_desc_walk = torch.from_numpy(desc_walk.view(np.uint8).reshape(n_pairs, nfr, 644)).to(dev)
        d_filt = torch.from_numpy(filt.view(np.uint8).reshape(-1, 92)).to(dev)
        d_pf = torch.from_numpy(pf.view(np.int32)).to(dev)
        pairs = np.arange(nch, dtype=np.int32).reshape(n_pairs, 2)
        d_pairs = torch.from_numpy(pairs).to(dev)
        delay = [torch.zeros((nch, 1024), device=dev, dtype=torch.float32) for _ in range(2)]
        pcm = torch.empty_like(coeffs)
        tools = sa.AacSpectralTools(ctx, swb_long, swb_short)

        def sequence(x, d_in, d_out):
            tools.joint_stereo_list(x, d_pairs, d_desc, d_pf)
            tools.tns(x, d_filt, len(filt))
            tools.synth_joint_stereo(x, side, d_in, d_pairs, d_desc_walk, pcm, delay_out=d_out)

        def step():
            sequence(coeffs, delay[0], delay[1])
            delay.reverse()
        step.input = coeffs

        def verify():
            import oracle
            x = pristine.clone()
            sequence(x, torch.zeros_like(delay[0]), delay[1])
            sync_dev()
            bad = checked = 0
            vr = verify_rng()
            picked = sorted({0, n_pairs - 1, int(vr.integers(0, n_pairs))})
            for p_ in picked:
                l, r = int(pairs[p_, 0]), int(pairs[p_, 1])
                a = int(vr.integers(0, max(1, nfr - 24)))
                for fr in (slice(0, min(nfr, 16)), slice(a, min(nfr, a + 16))):
                    f0 = max(0, fr.start - 1)  # (one frame of halo: the delay line of a frame depends on the previous frame's input alone)
                    cl, cr = pristine[l, f0:fr.stop].cpu().numpy(), pristine[r, f0:fr.stop].cpu().numpy()
                    dl, dr = cl.copy(), cr.copy()
                    for i in range(cl.shape[0]):
                        f = f0 + i
                        dl[i], dr[i] = oracle.aac_joint_stereo(cl[i], cr[i], 1, 40, swb_long, desc[p_, f]["mode"], desc[p_, f]["scale"])
                        for ch, buf in ((l, dl), (r, dr)):
                            for q in filt[filt["frame"] == ch * nfr + f]:
                                buf[i] = oracle.aac_tns_filter(buf[i], int(q["start"]), int(q["end"]), int(q["order"]), int(q["direction"]), q["lpc"][:int(q["order"])])
                    want, _ = oracle.aac_synth(np.stack([dl, dr]), side[[l, r], f0:fr.stop].cpu().numpy(), np.zeros((2, 1024), np.float32))
                    got = pcm[[l, r], fr].cpu().numpy()
                    bad += int((got != want[:, fr.start - f0:]).sum())
                    checked += got.size
            if bad:
                raise RuntimeError("bench: the aactns batch differs from the oracle in %d of %d sampled samples" % (bad, checked))
            return {"checker": "oracle/symoracle.c (joint stereo, TNS, then Dsp::synth), outside the timed region, on a pristine copy of the spectra",
                    "pairs": picked, "seed": VERIFY_SEED, "samples_compared": checked, "mismatches": bad, "criterion": "bit-identical f32 (value comparison)"}
        step.verify = verify
        frames = nch * nfr // 2
        bytes_alg = nch * nfr * 8192 + n_pairs * nfr * 644 + len(filt) * 92
        return step, frames, "frames", bytes_alg, {
            "workload": "AAC-LC 48 kHz stereo, %d long-block frames (%d chains x %d) from mid/side- and intensity-coded spectra (60 %% / 20 %% of the bands) with "
                        "one order-12 TNS filter on 30 %% of the channel frames (%d filters, %d of %d pair frames take the list pass): joint-stereo list pass, "
                        "filters, one walk" % (frames, nch, nfr, len(filt), len(pf), n_pairs * nfr),
            "channel_frames": nch * nfr, "tns_filters": int(len(filt)), "tns_filter_order": tns_order, "tns_pair_frames": int(len(pf))}, \
            "aac_joint_stereo_kernel (list) + aac_tns_kernel<true> + aac_synth_quad_kernel<true>", pcm
    if name in ("mp3q", "mp3q2"):
        # config 3 from what the ENTROPY DECODER produces: int16 Huffman samples + the 52-byte requantize record per granule-channel
        # + one 48-byte joint-stereo record per granule of a pair, every stream a mid/side pair, long blocks.
        #   mp3q : requantize + stereo + synthesis in ONE kernel (symaccel_mp3_decode_pp_device: the front lives in the load path)
        #   mp3q2: the same work as TWO kernels (requantize + stereo -> f32 spectra in HBM -> synthesis), the round-3 pipeline
        from symphonia_amd import backend
        nch, ngr = max(2, int(128 * scale)) & ~1, (6 if emulate else 2048)
        rng = np.random.default_rng(seed)
        q = torch.randint(-40, 41, (nch, ngr, 576), generator=g, device=dev, dtype=torch.int16)
        rq = np.zeros((nch, ngr), backend.MP3_REQUANT_DTYPE)
        rq["global_gain"], rq["rzero"] = 150, 576
        rq["scalefacs"] = rng.integers(0, 4, (nch, ngr, 39))
        st = np.zeros((nch // 2, ngr), backend.MP3_STEREO_DTYPE)
        st["flags"], st["rzero0"], st["rzero1"] = 1 | 4, 576, 576  # mid/side, MPEG-1
        units = np.arange(nch, dtype=np.int32).reshape(-1, 2)
        side_np = sa.mp3_side(np.zeros((nch, ngr), np.uint8), np.zeros((nch, ngr), np.uint8), np.full((nch, ngr), 576))
        as_bytes = lambda a: torch.from_numpy(np.ascontiguousarray(a).view(np.uint8).reshape(a.shape + (-1,))).to(dev)  # noqa: E731
        d_rq, d_st, d_units, side = as_bytes(rq), as_bytes(st), torch.from_numpy(units).to(dev), as_bytes(side_np)
        stt = [[torch.zeros((nch, 576), device=dev), torch.zeros((nch, 1024), device=dev), torch.zeros(nch, dtype=torch.int32, device=dev)]
               for _ in range(2)]
        pcm = torch.empty((nch, ngr, 576), device=dev, dtype=torch.float32)
        syn = sa.Mp3Synthesis(ctx, 0)
        if name == "mp3q":
            def run_from(s0, s1):
                syn.decode(q, d_rq, d_units, d_st, side, s0[0], s0[1], s0[2], pcm, state_out=s1)
            kernel = "mp3_synth_kernel<4, true>"
        else:
            xr = torch.empty((nch, ngr, 576), device=dev, dtype=torch.float32)
            ste = sa.Mp3Stereo(ctx, 0)

            def run_from(s0, s1):
                ste.requantize_stereo(q, d_rq, d_units, d_st, xr)
                syn.synth(xr, side, s0[0], s0[1], s0[2], pcm, state_out=s1)
            kernel = "mp3_stereo_kernel<true> + mp3_synth_kernel<1, false>"

        def step():
            run_from(stt[0], stt[1])
            stt.reverse()

        def verify():
            """the timed batch itself, from a zero state: sampled streams x the first granules against the oracle chain requantize ->
            stereo -> synthesis (layer3/mod.rs:421-477)"""
            import oracle
            run_from([torch.zeros_like(t) for t in stt[0]], stt[1])
            sync_dev()
            bad = checked = 0
            gw = min(ngr, 24)
            picked = sorted({0, nch // 2 - 1})
            for u in picked:
                c0, c1 = int(units[u, 0]), int(units[u, 1])
                qs = q[[c0, c1], :gw].cpu().numpy()
                xr_ = oracle.mp3_requantize(qs.reshape(-1, 576), np.ascontiguousarray(rq[[c0, c1], :gw]).reshape(-1), 0).reshape(2, gw, 576)
                for gi in range(gw):
                    xr_[0, gi], xr_[1, gi] = oracle.mp3_stereo(xr_[0, gi], xr_[1, gi], st[u, gi], 0)
                want = oracle.mp3_synth(xr_, np.ascontiguousarray(side_np[[c0, c1], :gw]).view(np.uint8).reshape(2, gw, 4), 0, np.zeros((2, 576), np.float32),
                                        np.zeros((2, 1024), np.float32), np.zeros(2, np.int32))[0]
                got = pcm[[c0, c1], :gw].cpu().numpy()
                bad += int((got != want).sum())
                checked += got.size
            if bad:
                raise RuntimeError("bench: the %s batch differs from the oracle in %d of %d sampled samples" % (name, bad, checked))
            return {"checker": "oracle/symoracle.c (requantize, stereo, then the synthesis tail), outside the timed region", "streams": picked,
                    "granule_windows": [[0, gw]], "samples_compared": checked, "mismatches": bad, "criterion": "bit-identical f32 (value comparison)"}
        step.verify = verify
        step.input = q
        granules = nch * ngr // 2
        per_gc = 1152 + 52 + 24 + 4 + 2304  # samples + requantize record + half a stereo record + side word in, PCM out
        return step, granules, "granules", nch * ngr * per_gc, {
            "workload": "MP3 Layer III 44.1 kHz stereo from int16 Huffman samples + side records, %d long-block mid/side granules (%d chains x %d): "
                        "requantize + joint stereo + hybrid synthesis + polyphase, %s" % (granules, nch, ngr, "one kernel" if name == "mp3q" else
                                                                                      "two kernels (f32 spectra through HBM)"),
            "granule_channels": nch * ngr, "algorithmic_bytes_per_granule_channel": per_gc}, kernel, pcm
    if name == "vorbis":
        nch, nb = max(1, int(64 * scale)), (16 if emulate else 4096)  # one GPU's shard of config 4: 8 streams x 8 ch x 4096 blocks
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
        spectra = torch.randn((nch, spec_stride), generator=g, device=dev, dtype=torch.float32) * 0.1
        d_flags = torch.from_numpy(flags).to(dev)
        # every step decodes the same packed batch from a fresh stream start (prev flag -1 fixes the packed layout), so the
        # incoming state is constant and the outgoing state goes to the second buffer: one launch per step
        prev = [torch.full((nch,), -1, dtype=torch.int32, device=dev), torch.zeros((nch,), dtype=torch.int32, device=dev)]
        overlap = [torch.zeros((nch, 1024), device=dev) for _ in range(2)]
        pcm = torch.zeros((nch, pcm_stride), device=dev)

        def step():
            v.synth(spectra, d_flags, prev[0], overlap[0], pcm_stride, pcm, state_out=(prev[1], overlap[1]))
        step.input = spectra
        step.vorbis = {"flags": flags, "spectrum": lambda: spectra, "pcm": pcm, "pcm_stride": pcm_stride, "used": po[:, -1]}
        bytes_alg = int(4 * (so[:, -1].sum() + po[:, -1].sum()))
        return step, nch * nb // 8, "frames", bytes_alg, {
            "workload": "Vorbis 2048/256 mixed block sizes, 8 ch, %d blocks (%d chains x %d)" % (nch * nb // 8, nch, nb),
            "channel_blocks": nch * nb}, "vorbis_synth_wave_kernel", pcm
    if name in ("vorbisf", "vorbisf2"):
        # config 4's shard from what the floor-1 and residue decoders produce: posts per channel-block + residue lines -> PCM.
        # vorbisf: floor-1 curve as dB-table indices (1 B / line, one call per block-size class) + synthesis that looks the table up and
        # multiplies as it loads the residue; vorbisf2: curve x residue written as an f32 spectrum, then plain synthesis (round 3).
        nch, nb = max(1, int(64 * scale)), (16 if emulate else 4096)
        rng = np.random.default_rng(seed)
        flags = np.zeros((nch, nb), np.uint8)
        cur = np.ones(nch, bool)
        for b in range(nb):
            r = rng.random(nch)
            cur = np.where(cur, r < 0.9, r >= 0.7)
            flags[:, b] = cur
        v = sa.VorbisDsp(ctx, 8, 11)
        so, po = v.layout(flags, np.full(nch, -1))
        spec_stride, pcm_stride = int(so[:, -1].max()), int(po[:, -1].max())
        spec_stride += (-spec_stride) % 4
        pcm_stride += (-pcm_stride) % 4
        residue = torch.randn((nch, spec_stride), generator=g, device=dev, dtype=torch.float32) * 0.1
        d_flags = torch.from_numpy(flags).to(dev)
        classes, n_post_words, host_ys = [], 0, []
        for flag, n, n_posts, mult in ((1, 1024, 40, 2), (0, 128, 12, 2)):
            xs = [0, n] + rng.permutation(np.arange(1, n))[:n_posts - 2].tolist()
            where = np.argwhere(flags == flag)
            ys = rng.integers(0, 128, size=(len(where), n_posts)).astype(np.uint32)
            ys[rng.random(ys.shape) < 0.2] = 0
            offs = np.array([c * spec_stride + so[c, b] for c, b in where], np.uint32)
            classes.append((xs, mult, torch.from_numpy(ys).to(dev), n, torch.from_numpy(offs).to(dev), len(where)))
            host_ys.append(ys)
            n_post_words += ys.size + offs.size
        plane = torch.zeros((nch, spec_stride), dtype=torch.uint8, device=dev)
        spectrum = torch.zeros((nch, spec_stride), device=dev)
        prev = [torch.full((nch,), -1, dtype=torch.int32, device=dev), torch.zeros((nch,), dtype=torch.int32, device=dev)]
        overlap = [torch.zeros((nch, 1024), device=dev) for _ in range(2)]
        pcm = torch.zeros((nch, pcm_stride), device=dev)

        def step():
            if name == "vorbisf":  # (both block-size classes in one launch)
                v.floor1_y_jobs([(xs, mult, ys, n, offs, cnt) for xs, mult, ys, n, offs, cnt in classes], plane)
            else:
                for xs, mult, ys, n, offs, cnt in classes:
                    v.floor1(xs, mult, ys, n, spectrum, cnt, residue=residue, line_offsets=offs)
            if name == "vorbisf":
                v.synth_floor_y(plane, residue, d_flags, prev[0], overlap[0], pcm_stride, pcm, state_out=(prev[1], overlap[1]))
            else:
                v.synth(spectrum, d_flags, prev[0], overlap[0], pcm_stride, pcm, state_out=(prev[1], overlap[1]))
        step.input = residue
        step.vorbis = {"flags": flags, "pcm": pcm, "pcm_stride": pcm_stride, "used": po[:, -1], "residue": residue, "so": so,
                       "classes": [(xs, mult, hy, n, flag) for (xs, mult, _, n, offs, cnt), hy, flag in zip(classes, host_ys, (1, 0))]}
        bytes_alg = int(4 * (so[:, -1].sum() + po[:, -1].sum()) + 4 * n_post_words)  # residue lines + PCM samples + posts and offsets
        return step, nch * nb // 8, "frames", bytes_alg, {
            "workload": "Vorbis 2048/256, 8 ch, %d blocks (%d chains x %d) from floor-1 posts + residue: %s" % (
                nch * nb // 8, nch, nb, "byte plane of dB-table indices, multiplied in the synthesis load path" if name == "vorbisf" else
                "curve x residue as an f32 spectrum, then synthesis"),
            "channel_blocks": nch * nb}, ("vorbis_floor1_pair_kernel + vorbis_synth_wave_kernel" if name == "vorbisf" else "vorbis_floor1_kernel x2 + vorbis_synth_wave_kernel"), pcm
    if name in ("flac", "flacp"):
        nb, bs = max(2, int(1048576 * scale) & ~1), 4096  # config 5: 1 M subframe blocks of 4096 samples = 16 GiB, in place
        buf, desc, co, pair_mode, _ = flac_config5(torch, nb, bs, seed, dev)
        fp = sa.FlacPredictor(ctx)
        # "flacp": the same batch with its rows at the pitch symaccel_row_stride() recommends (4608 words: rows 18 KiB apart instead
        # of 16 KiB) through symaccel_flac_restore_strided_device -- the layout a caller that owns the plane (the batcher) uses
        padded = name == "flacp"
        if padded:
            pitch = int(os.environ.get("SYM_BENCH_PITCH", "0")) or int(ctx.lib.dll.symaccel_row_stride(bs))  # (development: the sweep of profiles/r06zz33_*)
            wide = torch.full((nb, pitch), 0x5A5A5A5A, dtype=torch.int32, device=dev)
            wide[:, :bs] = buf
            del buf
            buf = wide

        def step():
            # In place, so every step after the first predicts over the previous step's output.  That is still one full
            # pass of the recurrence + decorrelation over the batch: the kernel has no data-dependent control flow (the
            # f64 / i64 path is chosen from the coefficients alone) and the arithmetic wraps, so every pass costs the same.
            if padded:
                fp.restore_strided(buf, desc, co, bs, pair_mode, 8)
            else:
                fp.restore_stereo(buf, desc, co, pair_mode, 8)

        def verify():
            """one more pass over the batch (outside every timed region): sampled channel pairs, their words saved in front of it,
            compared with the oracle's restore + decorrelate + shift of those words (the checker: oracle/, test infrastructure)"""
            import oracle
            pairs = sorted({0, min(1, nb // 2 - 1), max(0, nb // 4 - 1), nb // 4, nb // 2 - 1})
            rows = [r for p_ in pairs for r in (2 * p_, 2 * p_ + 1)]
            before = buf[rows][:, :bs].cpu().numpy()
            step()
            if torch.cuda.is_available():
                torch.cuda.synchronize()
            got = buf[rows][:, :bs].cpu().numpy()
            if padded and not bool((buf[rows][:, bs:] == 0x5A5A5A5A).all()):
                raise RuntimeError("bench: the FLAC kernel wrote into the padding of its rows")
            want = oracle.flac_restore(before, desc[rows].cpu().numpy(), co[rows].cpu().numpy())
            pm = pair_mode[pairs].cpu().numpy()
            for k in range(len(pairs)):
                a, b = oracle.flac_decorrelate(int(pm[k]), want[2 * k], want[2 * k + 1])
                want[2 * k], want[2 * k + 1] = oracle.flac_shl(a, 8), oracle.flac_shl(b, 8)
            bad = int((got != want).sum())
            if bad:
                raise RuntimeError("bench: the FLAC batch differs from the oracle in %d of %d sampled samples" % (bad, got.size))
            return {"checker": "oracle/symoracle.c (CPU restatement), outside the timed region", "channel_pairs": pairs,
                    "samples_compared": int(got.size), "mismatches": bad, "criterion": "bit-identical i32"}
        step.verify = verify
        return step, nb, "blocks", nb * bs * 8, {
            "workload": "FLAC 24-bit 192 kHz, LPC order 32 (15-bit quantised coefficients of random AR models, shift 10..14), "
                        "%d subframe blocks of 4096 samples, half the channel pairs mid/side (side channel 25 bits), "
                        "restore + decorrelate + left-justify fused, in place%s" % (
                            nb, ", rows %d words apart (symaccel_row_stride)" % buf.shape[1] if padded else ""), "samples": nb * bs}, "flac_restore_f64_kernel", buf
    if name in ("alac", "alacp"):
        nb, bs = max(1, int(262144 * scale)), 4096  # 16-bit ALAC frames of 4096 samples, adaptive predictor of order 8
        padded = name == "alacp"  # (as "flacp")
        pitch = (int(os.environ.get("SYM_BENCH_PITCH", "0")) or int(ctx.lib.dll.symaccel_row_stride(bs))) if padded else bs
        buf = torch.full((nb, pitch), 0x5A5A5A5A, dtype=torch.int32, device=dev)
        buf[:, :bs] = torch.randint(-(1 << 9), 1 << 9, (nb, bs), generator=g, device=dev, dtype=torch.int32)
        desc_np = sa.alac_desc(np.zeros(nb), np.full(nb, 8), np.full(nb, 9), np.full(nb, 16))
        desc = torch.from_numpy(desc_np.view(np.uint8).reshape(nb, 4)).to(dev)
        co = torch.randint(-200, 200, (nb, 32), generator=g, device=dev, dtype=torch.int32)
        ap = sa.AlacPredictor(ctx)

        def step():
            if padded:
                ap.predict_strided(buf, desc, co, bs)
            else:
                ap.predict(buf, desc, co)  # in place, like the FLAC workload: every pass costs the same

        def verify():
            """as for FLAC: sampled blocks of one more pass against the oracle's predictor on the words saved in front of it"""
            import oracle
            rows = sorted({0, min(1, nb - 1), min(63, nb - 1), min(64, nb - 1), nb // 2, nb - 1})
            before = buf[rows][:, :bs].cpu().numpy()
            step()
            if torch.cuda.is_available():
                torch.cuda.synchronize()
            got = buf[rows][:, :bs].cpu().numpy()
            if padded and not bool((buf[rows][:, bs:] == 0x5A5A5A5A).all()):
                raise RuntimeError("bench: the ALAC kernel wrote into the padding of its rows")
            want = oracle.alac_predict(before, desc_np[rows], co[rows].cpu().numpy())
            bad = int((got != want).sum())
            if bad:
                raise RuntimeError("bench: the ALAC batch differs from the oracle in %d of %d sampled samples" % (bad, got.size))
            return {"checker": "oracle/symoracle.c (CPU restatement), outside the timed region", "blocks": rows,
                    "samples_compared": int(got.size), "mismatches": bad, "criterion": "bit-identical i32"}
        step.verify = verify
        return step, nb, "blocks", nb * bs * 8, {
            "workload": "ALAC 16-bit, adaptive LPC order 8 (shift 9), %d element-channel blocks of 4096 samples, in place%s" % (
                nb, ", rows %d words apart (symaccel_row_stride)" % pitch if padded else ""),
            "samples": nb * bs}, "alac_predict_kernel", buf
    raise ValueError(name)


def flac_config5(torch, nb, bs, seed, dev, chunk=32768):
    """SURVEY 8d config 5 data, generated on the device: per block a 24-bit signal (25-bit for the side channel of a
    mid/side pair), the 15-bit quantised coefficients of a random stable AR(32) model with shift 10..14, and the
    residual of the forward predictor (FIR over the signal: decoder.rs:716-752 run backwards), so that restoring the
    block must give the signal back exactly.  Blocks 2p / 2p+1 are the channels of pair p; odd pairs are mid/side.
    Returns (buf[nb, bs] i32 = 32 warm-up samples then residuals, desc[nb, 4] u8, coeffs[nb, 32] i32, pair_mode[nb/2] u8,
    expect(b0, b1) -> the i32 samples blocks [b0, b1) must decode to after decorrelation and `<< 8`)."""
    g = torch.Generator(device=dev).manual_seed(seed)
    order = 32
    shift = torch.randint(10, 15, (nb,), generator=g, device=dev, dtype=torch.int64)
    # random stable AR(32): reflection coefficients shrinking with the order, stepped up to direct form (Levinson)
    k = (torch.rand((nb, order), generator=g, device=dev, dtype=torch.float64) * 2 - 1) * 0.9 * (0.8 ** torch.arange(order, device=dev, dtype=torch.float64))
    a = torch.zeros((nb, order), device=dev, dtype=torch.float64)
    for i in range(order):
        prev = a[:, :i].clone()
        a[:, :i] = prev - k[:, i:i + 1] * prev.flip(1)
        a[:, i] = k[:, i]
    # the AR model x[i] = -sum a[j] x[i-1-j] + e[i] predicts with coefficients -a
    co = torch.clamp(torch.round(-a * torch.exp2(shift.double())[:, None]), -16383, 16383).to(torch.int64)  # qlp precision 15
    pair_mode = torch.zeros(nb // 2, dtype=torch.uint8, device=dev)
    pair_mode[1::2] = 2  # decorrelate_mid_side
    t = torch.arange(bs, device=dev, dtype=torch.float32)
    buf = torch.empty((nb, bs), dtype=torch.int32, device=dev)

    def signal(b0, b1):
        """the PCM of blocks [b0, b1) as the subframes carry it (mid / side for the odd pairs), int64"""
        gs = torch.Generator(device=dev).manual_seed(seed * 1000003 + b0)
        n = b1 - b0
        f = torch.rand((n, 3), generator=gs, device=dev) * 0.2 + 0.001
        ph = torch.rand((n, 3), generator=gs, device=dev) * 6.2831853
        amp = torch.tensor([0.55, 0.3, 0.1], device=dev) * float(1 << 23)
        x = sum(amp[j] * torch.sin(f[:, j:j + 1] * t[None, :] + ph[:, j:j + 1]) for j in range(3))
        x = x + torch.randn((n, bs), generator=gs, device=dev) * 3000.0
        x = torch.clamp(torch.round(x), -(1 << 23), (1 << 23) - 1).to(torch.int64)  # left / right, 24 bit
        ms = (torch.arange(b0, b1, device=dev) // 2) % 2 == 1
        left, right = x[0::2].clone(), x[1::2].clone()
        mid, side = (left + right) >> 1, left - right
        msp = ms[0::2]
        x[0::2] = torch.where(msp[:, None], mid, left)
        x[1::2] = torch.where(msp[:, None], side, right)
        return x, left, right

    assert chunk % 2 == 0
    for b0 in range(0, nb, chunk):
        b1 = min(nb, b0 + chunk)
        x, _, _ = signal(b0, b1)
        pred = torch.zeros_like(x)
        for j in range(order):  # pred[i] = sum_j co[j] * x[i - 1 - j]
            pred[:, order:] += co[b0:b1, j:j + 1] * x[:, order - 1 - j:bs - 1 - j]
        res = x - (pred >> shift[b0:b1, None])
        res[:, :order] = x[:, :order]  # warm-up samples are stored verbatim
        buf[b0:b1] = ((res + (1 << 31)) % (1 << 32) - (1 << 31)).to(torch.int32)
    desc = torch.zeros((nb, 4), dtype=torch.uint8, device=dev)
    desc[:, 0], desc[:, 1], desc[:, 2] = 2, order, shift.to(torch.uint8)

    def expect(b0, b1):
        assert b0 % chunk == 0 and b1 <= min(nb, b0 + chunk)  # signal() is seeded per generation chunk
        _, left, right = signal(b0, min(nb, b0 + chunk))
        out = torch.empty((2 * left.shape[0], bs), dtype=torch.int64, device=dev)
        out[0::2], out[1::2] = left, right
        return ((out[: b1 - b0] << 8) + (1 << 31)) % (1 << 32) - (1 << 31)

    return buf, desc, co.to(torch.int32), pair_mode, expect


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
    """The CPU restatement of the reference's path timed on the host cores, on a bounded sample (about 2 + 2 + 10 s).
    Fan-out over cores happens inside oracle/bench_mt.c (pthreads, private outputs per thread).  Two builds of the SAME
    operation DAG are timed (tests/test_cpu_baseline.py asserts they produce identical bits):
      * the scalar -O2 restatement on one thread -- what the reference's non-SIMD, single-threaded design looks like;
      * the strongest CPU schedule we have, on every usable core -- for the headline workload the across-chains SIMD
        schedule of oracle/cpu_simd.c (16 chains per vector, -O3 -march=native), otherwise the -O3 -march=native build.
    `value` is the second figure.  Still a port, not rustc output: the reference's default build runs its FFT through
    rustfft's SIMD kernels (BENCHMARKS.md:15), which could be faster again."""
    import oracle
    cores = usable_cores()
    rng = np.random.default_rng(0)
    fast_kind = name
    if name == "aac":
        nch, nfr = 16, 32
        in0 = rng.standard_normal((nch, nfr, 1024)).astype(np.float32)
        in0[:, :, 672:] = 0.0
        in1 = np.full((nch, nfr), oracle.aac_side(0, 1, 1), np.uint8)
        kw = dict(n_chains=nch, per_chain=nfr)
        units, unit = nch * nfr / 2, "frames/s"
        sample = "%d channel-frames (16 chains x %d long blocks)" % (nch * nfr, nfr)
        fast_kind = "aac_simd"
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
    dtn, repsn = oracle.bench_mt(fast_kind, 1, 2.0, in0, in1, native=True, **kw)
    dt, reps = oracle.bench_mt(fast_kind, cores, seconds, in0, in1, native=True, **kw)
    build = ("oracle/cpu_simd.c: the scalar DAG with 16 chains per vector, gcc -O3 -march=native, no FMA" if fast_kind == "aac_simd"
             else "oracle/symoracle.c, gcc -O3 -march=native, no FMA")
    return {"value": units * reps / dt, "unit": unit, "cores": cores, "kind": "port",
            "single_thread_value": units * repsn / dtn,
            "scalar_O2_single_thread_value": units * reps1 / dt1,
            "cpu": "%s, %d logical CPUs online, %d usable" % (oracle.cpu_model(), os.cpu_count() or 0, cores),
            "sample": build + ": " + sample + " per task; %d tasks on %d threads in %.1f s (scalar -O2 restatement on one thread: "
                      "%.3g %s)" % (reps, cores, dt, units * reps1 / dt1, unit)}


def sample_clocks(step, sync):
    """What the board's clocks are WHILE the workload runs (boards of this pool differ in the clock they hold under load, which
    moves the issue-bound lines by 20-30 %: profiles/HISTORY.md, "the final tree on two boards"): about half a second of steps is
    queued, `rocm-smi --showclocks` is asked while they execute.  Outside every timed region; None if rocm-smi is not there."""
    import re
    import shutil
    import subprocess
    exe = shutil.which("rocm-smi") or "/opt/rocm/bin/rocm-smi"
    try:
        t0 = time.perf_counter()
        step()
        sync()
        per = max(time.perf_counter() - t0, 1e-5)
        for _ in range(int(min(20000, max(10, 0.6 / per)))):
            step()
        r = subprocess.run([exe, "--showclocks"], capture_output=True, text=True, timeout=20)
        sync()
        found = {}
        for name in ("sclk", "mclk", "fclk", "socclk"):
            m = re.search(r"%s clock level:?\s*\S*\s*\((\d+)\s*Mhz\)" % name, r.stdout, re.I)
            if m:
                found[name + "_mhz"] = int(m.group(1))
        if not found:
            return None
        found["source"] = "rocm-smi --showclocks while ~0.6 s of steps execute, outside the timed region"
        return found
    except Exception:  # noqa: BLE001
        return None


VERIFY_SEED = None  # per-run seed of the verification samples (main() draws it; printed in the line as `verified.seed`)


def verify_rng():
    return np.random.default_rng(VERIFY_SEED if VERIFY_SEED is not None else 0)


def verify_sampled_chains(name, step, torch, sync):
    """Tie the timed batch to a verified result: one more step of THE SAME batch (outside every timed region) from a zero
    carried state, then sampled chains x frame windows of its output compared bit for bit with the oracle (the checker,
    oracle/ -- test infrastructure).  A window that does not start at frame 0 is recomputed from `halo` frames earlier with
    a zero state: the carried state of these codecs depends on the previous frame's (MP3: two granules') INPUT alone
    (SURVEY 8e).  Windows: the chain's start, the 256-frame workgroup-walk boundary, a segment boundary, the chain's end."""
    import oracle
    if name in ("vorbis", "vorbisf", "vorbisf2"):
        return verify_vorbis_chains(name, step, torch, sync)
    if name == "aac":
        (coeffs, side), halo = step.aac, 1
    elif name == "mp3":
        (coeffs, side), halo = step.mp3, 2
    else:
        return None
    nch, nfr = int(coeffs.shape[0]), int(coeffs.shape[1])
    pcm = step.verify_step()
    sync()
    # the fixed samples (the chain's start, the 256-frame workgroup-walk boundary, a segment boundary, the chain's end of the first,
    # the middle two and the last chain) + chains and windows nobody chose: drawn from the run's seed, which the line prints
    rng = verify_rng()
    chains = sorted({0, nch // 2 - 1, nch // 2, nch - 1} | {int(c) for c in rng.integers(0, nch, 3)})
    wins = [(a, min(b, nfr)) for a, b in ((0, 24), (60, 70), (250, 262), (nfr - 12, nfr)) if a < nfr and a >= 0]
    for a in rng.integers(0, max(1, nfr - 16), 3):
        wins.append((int(a), min(int(a) + 12, nfr)))
    checked, bad = 0, 0
    for c in chains:
        for a, b in wins:
            a0 = max(0, a - halo)
            x = coeffs[c:c + 1, a0:b].cpu().numpy()
            sd = side[c:c + 1, a0:b].cpu().numpy()
            if name == "aac":
                want, _ = oracle.aac_synth(x, sd, np.zeros((1, 1024), np.float32))
            else:
                want, _, _, _ = oracle.mp3_synth(x, sd, 0, np.zeros((1, 576), np.float32), np.zeros((1, 1024), np.float32), np.zeros(1, np.int32))
            got = pcm[c, a:b].cpu().numpy()
            w = want[0, a - a0:]
            bad += int((got != w).sum())  # (value comparison: the sign of a zero is not part of the contract)
            checked += got.size
    if bad:
        raise RuntimeError("bench: the timed %s batch differs from the oracle in %d of %d sampled samples" % (name, bad, checked))
    return {"checker": "oracle/symoracle.c (CPU restatement), outside the timed region", "chains": chains, "frame_windows": wins,
            "seed": VERIFY_SEED, "samples_compared": checked, "mismatches": bad, "criterion": "bit-identical f32 (value comparison)"}


def verify_vorbis_chains(name, step, torch, sync):
    """The Vorbis workloads: every step decodes the batch from a fresh stream start, so the PCM the timed steps left IS the result to
    check -- whole sampled chains against the oracle (floor-1 curve x residue for the posts pipelines, lib.rs:289-291, then
    DspChannel::synth).  Packed layouts make frame windows awkward; a chain is 4096 blocks and takes the oracle a fraction of a second."""
    import oracle
    v = step.vorbis
    step()
    sync()
    flags = v["flags"]
    nch = flags.shape[0]
    chains = sorted({0, nch - 1} | {int(verify_rng().integers(0, nch))})  # first, last, and one the run's seed picks
    checked, bad = 0, 0
    for c in chains:
        if "classes" in v:
            res = v["residue"][c].cpu().numpy()
            spec = np.zeros_like(res)
            so = v["so"][c]
            for xs, mult, ys, n, flag in v["classes"]:
                where = np.argwhere(flags == flag)
                mine = np.nonzero(where[:, 0] == c)[0]
                ys_c = ys[mine]
                for y, (_, b) in zip(ys_c, where[mine]):
                    o = int(so[b])
                    spec[o:o + n] = oracle.vorbis_floor1(xs, y, mult, n) * res[o:o + n]
        else:
            spec = v["spectrum"]()[c].cpu().numpy()
        want = oracle.vorbis_synth(8, 11, spec[None], flags[c:c + 1], np.full(1, -1, np.int32), np.zeros((1, 1024), np.float32), v["pcm_stride"])[0][0]
        used = int(v["used"][c])
        first = (2048 if flags[c, 0] else 256) // 2  # a stream's first block owns n / 2 slots and leaves them untouched (lib.rs:298-303)
        got = v["pcm"][c, first:used].cpu().numpy()
        bad += int((got != want[first:used]).sum())
        checked += got.size
    if bad:
        raise RuntimeError("bench: the timed %s batch differs from the oracle in %d of %d sampled samples" % (name, bad, checked))
    return {"checker": "oracle/symoracle.c (CPU restatement), outside the timed region", "chains": chains, "frame_windows": "whole chains",
            "seed": VERIFY_SEED, "samples_compared": checked, "mismatches": bad, "criterion": "bit-identical f32 (value comparison)"}


def workload_input(name, step):
    """The spectra tensor a step consumes (what a one-to-all scatter would have to move)."""
    return step.input


def host_to_host_aac(sa, ctx, torch, pcm, coeffs, frames, reps=3):
    """The trait-adapter shape of the same batch: spectra start in (page-locked) HOST memory and the PCM must end there.
    symaccel_aac_synth_pipelined cuts the batch into chunks and overlaps H2D, kernels and D2H; reported beside -- never
    inside -- `value`, which is the resident-in-HBM figure."""
    nch, nfr = int(coeffs.shape[0]), int(coeffs.shape[1])
    h_in, h_out = sa.PinnedBuffer((nch, nfr, 1024), np.float32), sa.PinnedBuffer((nch, nfr, 1024), np.float32)
    torch.cuda.synchronize()
    h_in.array[:] = coeffs.cpu().numpy()
    side = np.full((nch, nfr), int(sa.aac_side(0, 1, 1)), np.uint8)
    delay = np.zeros((nch, 1024), np.float32)
    d = ctx.lib.dll
    times = []
    for r in range(reps + 1):
        delay[:] = 0.0
        t0 = time.perf_counter()
        ctx._call(d.symaccel_aac_synth_pipelined, h_in.array.ctypes.data, side.ctypes.data, delay.ctypes.data, h_out.array.ctypes.data,
                  nch, nfr, 0)
        times.append(time.perf_counter() - t0)
    best = min(times[1:])
    # (frame 0 depends on the incoming delay line, which the resident steps carry on from step to step: compare from frame 1)
    same = bool(np.array_equal(h_out.array[:2, 1:64].view(np.uint32), pcm[:2, 1:64].cpu().numpy().view(np.uint32)))
    nbytes = h_in.array.nbytes
    # the same batch as the spectrum decoder leaves it -- mid/side- and intensity-coded spectra + one 644-byte descriptor per pair-frame
    # -- through symaccel_aac_decode_pipelined (joint stereo decoded on load by the pair walk, no TNS here): what the second-generation
    # AAC seam feeds
    coded = None
    try:
        rng = np.random.default_rng(11)
        n_pairs = nch // 2
        desc = np.zeros((n_pairs, nfr), sa.AAC_JS_DTYPE)
        desc["num_windows"], desc["max_sfb"] = 1, 40
        desc["mode"] = rng.choice([0, 1, 1, 1, 1, 1, 1, 2, 2, 0], (n_pairs, nfr, 128)).astype(np.uint8)
        desc["scale"] = (rng.standard_normal((n_pairs, nfr, 128)) * 0.5).astype(np.float32)
        pairs = np.arange(nch, dtype=np.int32).reshape(n_pairs, 2)
        swb_long = np.array([0, 4, 8, 12, 16, 20, 24, 28, 32, 36, 40, 48, 56, 64, 72, 80, 88, 96, 108, 120, 132, 144, 160, 176, 196, 216, 240, 264, 292,
                             320, 352, 384, 416, 448, 480, 512, 544, 576, 608, 640, 672, 704, 736, 768, 800, 832, 864, 896, 928, 1024], np.uint16)
        swb_short = np.array([0, 4, 8, 12, 16, 20, 28, 36, 44, 56, 68, 80, 96, 112, 128], np.uint16)
        tj = []
        for r in range(reps + 1):
            delay[:] = 0.0
            t0 = time.perf_counter()
            ctx._call(d.symaccel_aac_decode_pipelined, h_in.array.ctypes.data, side.ctypes.data, pairs.ctypes.data, desc.ctypes.data, n_pairs,
                      swb_long.ctypes.data, len(swb_long) - 1, swb_short.ctypes.data, len(swb_short) - 1, None, 0, delay.ctypes.data,
                      h_out.array.ctypes.data, nch, nfr, 0)
            tj.append(time.perf_counter() - t0)
        bj = min(tj[1:])
        coded = {"value": frames / bj, "unit": "frames/s", "ms": bj * 1e3, "bytes_in": nbytes + desc.nbytes, "bytes_out": nbytes,
                 "finite": bool(np.isfinite(h_out.array[:2, :8]).all()),
                 "path": "symaccel_aac_decode_pipelined: coded spectra + joint-stereo descriptors (60 % / 20 % of the bands mid/side / intensity) -> PCM"}
    except Exception as e:  # noqa: BLE001
        coded = {"error": "%s: %s" % (type(e).__name__, e)}
    h_in.free()
    h_out.free()
    return {"value": frames / best, "unit": "frames/s", "ms": best * 1e3, "bytes_each_way": nbytes, "GBps_each_way": nbytes / best / 1e9,
            "matches_resident_result": same, "from_coded_spectra": coded,
            "path": "symaccel_aac_synth_pipelined on page-locked host buffers: chunked H2D || kernel || D2H on three streams"}


def copy_ceiling(ctx, torch, seg_len, reps=10):
    """the copy rates THIS run reaches with the synthesis kernels' traffic shape (1 byte read : 1 byte
    written, config 2's footprint: 512 MiB in, 512 MiB out), through symaccel_probe_copy_device on the launch stream:
    a plain 16 B/lane grid-stride copy, and the copy in which every wavefront streams `seg_len` consecutive 4 KiB frames
    (the wavefront walk's access pattern), and the workgroup walk's (aac_synth_quad_kernel: four wavefronts share consecutive frames round-robin).  HIP events around `reps` launches each, after `reps` untimed ones."""
    nbytes = 512 << 20
    a = torch.empty(nbytes // 4, dtype=torch.float32, device="cuda").normal_()
    b = torch.empty_like(a)
    d = ctx.lib.dll
    out = {"bytes_read": nbytes, "bytes_written": nbytes, "unit": "GB/s", "reps": reps}
    for key, fpw, flags in (("plain_float4", 0, 0), ("plain_float4_nt", 0, 1), ("frames_per_wavefront_1_nt", 1, 1),
                            ("frames_per_wavefront_%d_nt" % seg_len, seg_len, 1), ("frames_per_wavefront_%d" % seg_len, seg_len, 0),
                            ("frames_per_workgroup_walk_%d_nt" % (4 * seg_len), seg_len, 9),
                            ("frames_per_workgroup_walk_%d_window_major_nt" % (4 * seg_len), seg_len, 9 | 32)):
        for _ in range(reps):  # (untimed: the same sustained state as the headline's timed region)
            ctx._call(d.symaccel_probe_copy_device, a.data_ptr(), b.data_ptr(), nbytes, fpw, flags)
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ev0.record()
        for _ in range(reps):
            ctx._call(d.symaccel_probe_copy_device, a.data_ptr(), b.data_ptr(), nbytes, fpw, flags)
        ev1.record()
        torch.cuda.synchronize()
        out[key] = 2.0 * nbytes / (ev0.elapsed_time(ev1) / 1e3 / reps) / 1e9
    if not bool(torch.equal(a, b)):
        raise RuntimeError("probe copy did not copy")
    del a, b
    return out


def exchange_c_api(ctx, torch, dist, world, rank, local_in, local_out, step, units, unit_name, sync, reps=3):
    """The scatter -> synthesis -> gather leg again, through the C ABI a Rust / C++ host would use (csrc/multi.cpp):
    symaccel_comm_unique_id on rank 0, the id handed round with torch.distributed, symaccel_comm_init on every rank (its own
    RCCL communicator, nothing borrowed from torch), then symaccel_scatter_streams / symaccel_gather_streams with every rank's
    shard as one "stream".  Returns the same keys as the torch leg."""
    import ctypes as C
    d = ctx.lib.dll
    uid = (C.c_char * 128)()
    if rank == 0:
        assert d.symaccel_comm_unique_id(C.addressof(uid)) == 0  # (no context argument)
    box = [uid.raw]
    dist.broadcast_object_list(box, src=0)
    uid = C.create_string_buffer(box[0], 128)
    comm = C.c_void_p()
    ctx._call(d.symaccel_comm_init, C.addressof(uid), world, rank, C.byref(comm))
    try:
        in_bytes, out_bytes = local_in.numel() * local_in.element_size(), local_out.numel() * local_out.element_size()
        all_in = torch.empty((world,) + tuple(local_in.shape), dtype=local_in.dtype, device=local_in.device) if rank == 0 else None
        all_out = torch.empty((world,) + tuple(local_out.shape), dtype=local_out.dtype, device=local_out.device) if rank == 0 else None
        if rank == 0:
            all_in[:] = local_in
        p_in, p_out = (all_in.data_ptr(), all_out.data_ptr()) if rank == 0 else (None, None)

        def once():
            ts = [time.perf_counter()]
            ctx._call(d.symaccel_scatter_streams, comm, world, rank, 0, p_in, local_in.data_ptr(), world, in_bytes)
            sync()
            ts.append(time.perf_counter())
            step()
            sync()
            ts.append(time.perf_counter())
            ctx._call(d.symaccel_gather_streams, comm, world, rank, 0, local_out.data_ptr(), p_out, world, out_bytes)
            sync()
            ts.append(time.perf_counter())
            return ts

        once()  # connection set-up
        acc = [0.0, 0.0, 0.0, 0.0]
        for _ in range(reps):
            dist.barrier()
            sync()
            ts = once()
            for i in range(3):
                acc[i] += ts[i + 1] - ts[i]
            acc[3] += ts[3] - ts[0]
        from symphonia_amd.sharding import max_over_ranks
        secs = {k: max_over_ranks(a / reps, dist, device="cuda") for k, a in zip(("scatter", "step", "gather", "total"), acc)}
        same = bool(torch.equal(all_out[0], local_out)) if rank == 0 else None  # (the root's own slice came back through the local copy)
        out = {"op": "symaccel_scatter_streams + synthesis + symaccel_gather_streams on a communicator of its own (RCCL over xGMI, C ABI)",
               "ms": {k: v * 1e3 for k, v in secs.items()}, "bytes_per_rank": {"in": in_bytes, "out": out_bytes},
               "value_inclusive": units * world / secs["total"], "unit": unit_name + "/s", "root_slice_round_trip": same}
        # the same leg chunked and overlapped (symaccel_exchange_pipelined): streams = chains, the step = the synthesis of a chunk of
        # this rank's chains from a zero state; the root's link carries chunk c + 1 out and chunk c - 1's PCM back while chunk c is decoded
        try:
            if not (hasattr(step, "chunk_step") and local_in.dim() == 3):
                return out
            chains = int(local_in.shape[0])
            per_chain_in, per_chain_out = in_bytes // chains, out_bytes // chains
            STEP = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_size_t, C.c_size_t)
            failed = []

            def cb(user, first, count):
                try:
                    step.chunk_step(int(first), int(count))
                    return 0
                except Exception as e:  # noqa: BLE001
                    failed.append(repr(e))
                    return 1
            cbf = STEP(cb)
            n_chunks = 4

            def piped():
                t0 = time.perf_counter()
                ctx._call(d.symaccel_exchange_pipelined, comm, world, rank, 0, p_in, local_in.data_ptr(), per_chain_in, p_out, local_out.data_ptr(),
                          per_chain_out, world * chains, n_chunks, cbf, None)
                sync()
                return time.perf_counter() - t0
            piped()
            tp = 0.0
            for _ in range(reps):
                dist.barrier()
                sync()
                tp += piped()
            tp = max_over_ranks(tp / reps, dist, device="cuda")
            out["pipelined"] = {"op": "symaccel_exchange_pipelined: %d chunks of chains, scatter / synthesis / gather overlapped on two streams" % n_chunks,
                                "ms": tp * 1e3, "value_inclusive": units * world / tp, "errors": failed or None,
                                "root_slice_round_trip": bool(torch.equal(all_out[0], local_out)) if rank == 0 else None}
        except Exception as e:  # noqa: BLE001  (what was measured above stays)
            out["pipelined"] = {"error": "%s: %s" % (type(e).__name__, e)}
        return out
    finally:
        d.symaccel_comm_destroy(comm)


def guarded(fn, seconds):
    """Run an optional leg on a watchdog: (result, hung).  A leg that never returns (a collective that cannot complete) must not
    cost the line: the caller reports the timeout and leaves with os._exit once the line is out."""
    import threading
    box = {}

    def run():
        try:
            box["result"] = fn()
        except Exception as e:  # noqa: BLE001
            box["result"] = {"error": "%s: %s" % (type(e).__name__, e)}
    th = threading.Thread(target=run, daemon=True)
    th.start()
    th.join(seconds)
    if th.is_alive():
        return {"error": "timed out after %d s" % seconds}, True
    return box["result"], False


def host_to_host_mp3(sa, ctx, torch, nch, ngr, granules, reps=3):
    """Config 3 from and to page-locked HOST memory, two ways: the f32 spectra through symaccel_mp3_synth_pipelined (576 MiB in),
    and what the entropy decoder actually produces -- int16 Huffman samples + side records -- through
    symaccel_mp3_decode_pipelined (requantize + joint stereo + synthesis on the device: 288 MiB + records in).  Synthetic long
    blocks, every stream a mid/side pair.  Reported beside -- never inside -- `value`."""
    from symphonia_amd import backend
    d = ctx.lib.dll
    rng = np.random.default_rng(5)
    h_q = sa.PinnedBuffer((nch, ngr, 576), np.int16)
    h_x = sa.PinnedBuffer((nch, ngr, 576), np.float32)
    h_out = sa.PinnedBuffer((nch, ngr, 576), np.float32)
    h_q.array[:] = rng.integers(-40, 41, (nch, ngr, 576), dtype=np.int16)
    h_x.array[:] = (h_q.array * np.float32(0.01))
    rq = np.zeros((nch, ngr), backend.MP3_REQUANT_DTYPE)
    rq["global_gain"], rq["rzero"] = 150, 576
    rq["scalefacs"] = rng.integers(0, 4, (nch, ngr, 39))
    st = np.zeros((nch // 2, ngr), backend.MP3_STEREO_DTYPE)
    st["flags"], st["rzero0"], st["rzero1"] = 1 | 4, 576, 576  # mid/side, MPEG-1
    pairs = np.arange(nch, dtype=np.int32).reshape(-1, 2)
    side = sa.mp3_side(np.zeros((nch, ngr), np.uint8), np.zeros((nch, ngr), np.uint8), np.full((nch, ngr), 576))
    side = np.ascontiguousarray(side)
    out = {}
    for key, call in (("f32_spectra", lambda o, v, f: ctx._call(d.symaccel_mp3_synth_pipelined, h_x.array.ctypes.data, side.ctypes.data, 0, o.ctypes.data,
                                                              v.ctypes.data, f.ctypes.data, h_out.array.ctypes.data, nch, ngr, 0)),
                      ("int16_samples", lambda o, v, f: ctx._call(d.symaccel_mp3_decode_pipelined, h_q.array.ctypes.data, rq.ctypes.data, pairs.ctypes.data,
                                                                st.ctypes.data, nch // 2, side.ctypes.data, 0, o.ctypes.data, v.ctypes.data, f.ctypes.data,
                                                                h_out.array.ctypes.data, nch, ngr, 0))):
        times = []
        for _ in range(reps + 1):
            o, v, f = np.zeros((nch, 576), np.float32), np.zeros((nch, 1024), np.float32), np.zeros(nch, np.int32)
            t0 = time.perf_counter()
            call(o, v, f)
            times.append(time.perf_counter() - t0)
        best = min(times[1:])
        bytes_in = h_x.array.nbytes if key == "f32_spectra" else h_q.array.nbytes + rq.nbytes + st.nbytes
        out[key] = {"value": granules / best, "unit": "granules/s", "ms": best * 1e3, "bytes_in": int(bytes_in) + side.nbytes,
                    "bytes_out": h_out.array.nbytes, "finite": bool(np.isfinite(h_out.array[:2, :8]).all())}
    out["path"] = ("symaccel_mp3_synth_pipelined (f32 spectra) and symaccel_mp3_decode_pipelined (int16 samples + records; requantize + joint "
                   "stereo + synthesis on the device) on page-locked host buffers, chunked H2D || kernels || D2H")
    for b in (h_q, h_x, h_out):
        b.free()
    return out


def decoders_workload(quick=False, seconds_cpu=3.0, lookahead=64):
    """Trait-level throughput (`--workload decoders`, and a short form in the default line's `other_workloads`): S AAC-LC stereo
    streams decoded packet by packet through the compiled C++ twin of the shim's decoders (codecs::LookaheadDecoder,
    include/symaccel.hpp; tools/decoders_bench.cpp) with ONE process-wide cross-stream batcher (symaccel_batcher_*,
    csrc/batcher.cpp) -- host memory in, host memory out, T = min(S, cores) caller threads -- beside the CPU port decoding the same
    packets frame by frame on the same number of threads (oracle/bench_mt.c: one so_aac_synth call per packet and channel, what
    AudioDecoder::decode_ref does per call, codecs/audio.rs:279-297) and beside the decoders batching per stream (no batcher).
    PCIe-inclusive by construction: never the headline `value`."""
    import subprocess
    import oracle
    from symphonia_amd import build as sa_build
    exe = sa_build.build_decoders_bench()
    cores = usable_cores()
    rng = np.random.default_rng(0)
    sweep = [256] if quick else [1, 4, 16, 64, 256, 1024]
    # Timed packets per stream: at least 16 batches.  The harness warms up with two batches per stream and then times `packets`; the
    # look-ahead decoder has its next batch in flight when the timed region starts, so a region of only one or two batches (rounds 4 / 5
    # until the soak of profiles/r05z_soak_lengths.txt: 512 packets at a look-ahead of 256) counts work done before the clock started --
    # 4.8 M packets/s where the sustained rate is 1.8 - 2.3 M.
    def timed_packets(la, streams):
        return max(512, 16 * la)
    packets = timed_packets(lookahead, 256)
    out = {"harness": "tools/decoders_bench.cpp (g++, links libsymaccel.so only)", "decoders_built_by": "CodecRegistry::make_audio_decoder (--via-registry)", "lookahead": lookahead, "cores": cores, "sweep": [],
           "timed_packets_per_stream": {"lookahead_%d" % lookahead: packets, "lookahead_256": timed_packets(256, 256),
                                        "note": "after a warm-up of two batches per stream; >= 16 batches timed"}}

    def run(codec, streams, threads, per_stream=False, pk=packets, reps=3, la=None):
        """the harness `reps` times (a fresh process each: its own context, pool and warm-up); the run with the MEDIAN rate is the one
        reported, every rate is kept beside it (`runs_packets_per_s`: threads meeting a shared pipeline scatter by +-20 %).
        The pooled decoders are built the way the registry builds them -- `--via-registry`: CodecRegistry::make_audio_decoder is handed
        (params, options) and the packet source, no batcher (registry.rs:330-341; the Rust shim's try_registry_new) -- and, where the
        codec's batch layout allows it, parse straight into the batcher's slot (`--direct`)."""
        cmd = [str(exe), "--codec", codec, "--streams", str(streams), "--lookahead", str(la or lookahead), "--packets", str(pk), "--threads", str(threads)]
        if per_stream:
            cmd.append("--per-stream")
        else:
            cmd.append("--via-registry")
            if codec in ("aac", "mp3", "mp3h"):
                cmd.append("--direct")
        lines = []
        for _ in range(reps):
            r = subprocess.run(cmd, capture_output=True, text=True, timeout=300)
            if r.returncode != 0:
                return {"error": (r.stderr or r.stdout)[-400:]}
            lines.append(json.loads(r.stdout.strip().splitlines()[-1]))
        lines.sort(key=lambda d: d["packets_per_s"])
        mid = dict(lines[len(lines) // 2])
        mid["runs_packets_per_s"] = [d["packets_per_s"] for d in lines]
        return mid

    def cpu(threads):
        in0 = rng.standard_normal((2, 1, 1024)).astype(np.float32)
        in0[:, :, 672:] = 0.0
        in1 = np.full((2, 1), oracle.aac_side(0, 1, 1), np.uint8)
        dt, reps = oracle.bench_mt("aac", threads, seconds_cpu, in0, in1, native=True, n_chains=2, per_chain=1)
        return reps / dt
    cpu_cache = {}
    for s_ in sweep:
        t_ = max(1, min(s_, cores))
        if t_ not in cpu_cache:
            cpu_cache[t_] = cpu(t_)
        g = run("aac", s_, t_, pk=packets)
        row = {"streams": s_, "threads": t_, "gpu_batcher": g, "cpu_port_packets_per_s": cpu_cache[t_],
               "gpu_over_cpu": (g.get("packets_per_s", 0.0) / cpu_cache[t_]) if "error" not in g else None}
        if not quick and s_ in (16, 256):
            row["gpu_per_stream_batches"] = run("aac", s_, t_, per_stream=True, pk=packets // 2)
        out["sweep"].append(row)
        log("decoders: S = %d done" % s_)
    if not quick:
        # the same sweep at the Rust shim's default look-ahead (DEFAULT_LOOKAHEAD = 256 packets): a batch is a fixed ~150 us of
        # latency whatever its size, so few streams gain the most from longer batches
        out["sweep_lookahead_256"] = []
        for s_ in (1, 4, 16, 64, 256):
            t_ = max(1, min(s_, cores))
            g = run("aac", s_, t_, pk=timed_packets(256, s_), la=256)
            out["sweep_lookahead_256"].append({"streams": s_, "threads": t_, "gpu_batcher": g, "cpu_port_packets_per_s": cpu_cache[t_],
                                               "gpu_over_cpu": (g.get("packets_per_s", 0.0) / cpu_cache[t_]) if "error" not in g else None})
        out["mp3_int16_S256"] = run("mp3h", 256, max(1, min(256, cores)))
        out["mp3_f32_S256"] = run("mp3", 256, max(1, min(256, cores)))
        out["vorbis_8ch_S64"] = run("vorbis", 64, max(1, min(64, cores)))  # (BASELINE config 4's shape: 8 channels, 2048 / 256)
        # AAC one stage earlier: coded spectra + joint-stereo descriptors + TNS filters (30 % of the frames) -> PCM
        out["aac_coded_S256"] = run("aacd", 256, max(1, min(256, cores)))
        out["flac_S256"] = run("flac", 256, max(1, min(256, cores)))  # (stereo, 24 bit, 4096-sample blocks: one packet = one block)
        in0 = rng.standard_normal((16, 32, 1024)).astype(np.float32)
        in0[:, :, 672:] = 0.0
        in1 = np.full((16, 32), oracle.aac_side(0, 1, 1), np.uint8)
        dt, reps = oracle.bench_mt("aac_simd", cores, seconds_cpu, in0, in1, native=True, n_chains=16, per_chain=32)
        out["cpu_port_across_streams_simd_packets_per_s"] = 8 * 32 * reps / dt
        out["cpu_port_across_streams_simd_note"] = ("oracle/cpu_simd.c on %d threads: 16 chains (8 stereo streams) per vector, 32 frames per call -- "
                                                    "a CPU decoder would need the same cross-stream batcher to run this schedule" % cores)
    best = max((r for r in out["sweep"] if "error" not in r["gpu_batcher"]), key=lambda r: r["gpu_batcher"]["packets_per_s"], default=None)
    if best:
        out["best"] = {"streams": best["streams"], "packets_per_s": best["gpu_batcher"]["packets_per_s"], "gpu_over_cpu": best["gpu_over_cpu"]}
        if not quick:  # (the full sweep only: where the batcher stays ahead of the CPU port on the same number of threads from there on)
            ahead = None
            for r in reversed(out["sweep"]):
                if r["gpu_over_cpu"] and r["gpu_over_cpu"] > 1.0:
                    ahead = r["streams"]
                else:
                    break
            out["gpu_ahead_of_cpu_port_from_streams"] = ahead
            out["gpu_over_cpu_by_streams"] = {str(r["streams"]): r["gpu_over_cpu"] for r in out["sweep"]}
    out["note"] = ("packets/s through decode(): one packet = one AAC-LC stereo frame (2 x 1024 lines in, 2 x 1024 samples out, f32); the CPU port "
                   "runs native (-O3 -march=native) frame by frame on the same number of threads")
    return out


def relaunch_under_torchrun(args):
    """`python bench.py --gpus N` with no launcher in the environment: start N ranks ourselves (one process per GPU) with
    torch.distributed.run on the loopback address and hand its exit status back.  (The driver may also start the ranks
    itself -- `python -m torch.distributed.run ... bench.py --gpus N` -- in which case WORLD_SIZE is set and we are a rank.)"""
    import socket
    import subprocess
    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), str(Path(__file__).resolve())] + sys.argv[1:]
    log("no launcher in the environment: starting %d ranks: %s" % (args.gpus, " ".join(cmd)))
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    return subprocess.call(cmd, env=env)


def rccl_version(torch):
    try:
        return list(torch.cuda.nccl.version())
    except Exception:  # noqa: BLE001
        return None


def device_identity(torch, local_rank, emulate):
    """Something that differs between two physical GPUs (PCI bus id via the UUID when the runtime gives one)."""
    if emulate:
        return "emulated-%d" % local_rank
    props = torch.cuda.get_device_properties(local_rank)
    uuid = getattr(props, "uuid", None)
    return "%s|%s|%s" % (props.name, uuid if uuid is not None else "no-uuid", getattr(props, "pci_bus_id", local_rank))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=256,
                    help="timed steps (default 256: a 0.2 ms step gives a timed region of ~50 ms, long enough for an outside clock to see)")
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="aac", choices=["aac", "mp3", "vorbis", "flac", "alac", "flacp", "alacp", "mp3q", "mp3q2", "vorbisf", "vorbisf2", "aacjs", "aacjs2", "aactns", "decoders"])
    ap.add_argument("--segment", type=int, default=0, help="frames per wavefront segment (0 = library default)")
    ap.add_argument("--scale", type=float, default=1.0, help="batch size multiplier (development only)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--mix", dest="aac_mix", type=float, default=0.0,
                    help="development: probability of a block switch per long frame / granule in the AAC and MP3 workloads (headline: 0)")
    ap.add_argument("--gather", action="store_true",
                    help="N > 1 only: also time an RCCL all_gather of the PCM shards (reported beside, never inside, `value`)")
    ap.add_argument("--no-exchange", action="store_true",
                    help="N > 1: skip the scatter -> synthesis -> gather leg (reported beside, never inside, `value`)")
    ap.add_argument("--no-host-path", action="store_true",
                    help="N = 1, aac: skip the host-to-host line (pinned, chunked, overlapped staging through symaccel_aac_synth_pipelined)")
    ap.add_argument("--no-others", action="store_true",
                    help="N = 1, aac: skip the `other_workloads` object (BASELINE configs 3, 4 (one GPU's shard), 5 and the ALAC "
                         "predictor: 20 / 20 / 8 / 8 steps each with the same event timing)")
    ap.add_argument("--no-copy-ceiling", action="store_true", help="N = 1: skip the same-run copy probes")
    ap.add_argument("--spinup-ms", type=int, default=60, help="milliseconds of back-to-back steps in front of the W warm-up steps (sustained clocks)")
    ap.add_argument("--no-spinup", action="store_true", help="measure W + K from an idle board only (the clock ramp)")
    ap.add_argument("--repeats", type=int, default=5,
                    help="further W + K regions timed right after the line's own (same protocol, back to back): `repeats` reports every "
                         "region's ms per step, the median value and the spread; 0 = none")
    ap.add_argument("--no-config4", action="store_true", help="N > 1: skip the extra BASELINE config-4 (Vorbis shard) line")
    ap.add_argument("--selftest-multi", type=int, nargs="?", const=4, default=0, metavar="WORLD",
                    help="one GPU: exercise the N > 1 C path (RCCL binding at world size 1; scatter -> synthesis -> gather with WORLD "
                         "in-process ranks over a mailbox transport), print its JSON and exit")
    ap.add_argument("--emulate", action="store_true",
                    help="TEST ONLY: run the control flow on CPU tensors through the CPU emulation build of the kernels "
                         "(tests/emu) with gloo; its numbers mean nothing")
    ap.add_argument("--no-verify", action="store_true", help="measurement builds whose results are wrong on purpose (SYMACCEL_TUNE_*_ABLATE): skip the oracle check; the line says so")
    ap.add_argument("--verify-seed", type=int, default=None,
                    help="seed of the chains / windows the `verified` blocks sample beside the fixed ones (default: drawn per run, printed in the line)")
    args = ap.parse_args()
    global VERIFY_SEED
    VERIFY_SEED = args.verify_seed if args.verify_seed is not None else int.from_bytes(os.urandom(4), "little")
    if args.selftest_multi:
        from symphonia_amd.selftest import multi_selftest
        print(json.dumps({"selftest_multi": multi_selftest(args.selftest_multi)}), flush=True)
        return
    if args.workload == "decoders":
        # (host memory in -> host memory out through the C++ twin of the shim's decoders: its own harness, its own JSON)
        d = decoders_workload()
        best = d.get("best") or {}
        print(json.dumps({"metric": "decoded packets/sec through AudioDecoder-shaped decode() (host memory in and out, cross-stream batcher)",
                          "value": best.get("packets_per_s"), "unit": "packets/s", "n_gpus": 1, "higher_is_better": True, "dtype": "f32",
                          "data": "synthetic", "config": {"workload": "S AAC-LC 48 kHz stereo streams x look-ahead %d through codecs::LookaheadDecoder + "
                                                          "symaccel_batcher, S = %s" % (d["lookahead"], best.get("streams"))},
                          "vs_baseline": None, "decoders": d}), flush=True)
        return
    if args.gpus < 1:
        sys.exit("--gpus must be >= 1")
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(relaunch_under_torchrun(args))
    import faulthandler
    faulthandler.enable()
    faulthandler.dump_traceback_later(170, repeat=True, file=sys.stderr)  # a hang leaves a stack in the log

    import torch
    import torch.distributed as dist
    import symphonia_amd as sa
    from symphonia_amd.sharding import max_over_ranks, timed_exchange

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        sys.exit("bench.py --gpus %d was started with WORLD_SIZE=%d: the launcher must start one rank per GPU" % (args.gpus, world))
    emulate = args.emulate
    # Development smoke test of the N > 1 control flow on a one-GPU box (SYM_BENCH_ONE_GPU_SMOKE=1): every rank shares
    # cuda:0 and the collectives go through gloo on the host.  Never set by the driver; its numbers mean nothing.
    one_gpu_smoke = os.environ.get("SYM_BENCH_ONE_GPU_SMOKE") == "1"
    if emulate:
        sys.path.insert(0, str(ROOT / "tests"))
        from emu_lib import emu_library
        library = emu_library()
        args.scale = min(args.scale, 1.0 / 32)
    else:
        library = None
        if not torch.cuda.is_available():
            sys.exit("bench.py needs an MI355X: symphonia_amd has no CPU path")
        if one_gpu_smoke:
            local_rank = 0
        elif torch.cuda.device_count() < world:
            sys.exit("bench.py --gpus %d: only %d GPU(s) visible" % (world, torch.cuda.device_count()))
        torch.cuda.set_device(local_rank)
    host_collectives = emulate or one_gpu_smoke
    if world > 1:
        if host_collectives:
            dist.init_process_group("gloo")
        else:
            # (a collective that cannot complete should end the run in minutes, not in the default ten)
            import datetime
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank), timeout=datetime.timedelta(seconds=180))
        assert dist.get_world_size() == args.gpus, (dist.get_world_size(), args.gpus)
    red_dev = "cpu" if host_collectives else "cuda"

    def sync():
        if not emulate:
            torch.cuda.synchronize()

    # every rank must sit on its own GPU
    ident = device_identity(torch, local_rank, emulate)
    idents = [ident]
    if world > 1:
        idents = [None] * world
        dist.all_gather_object(idents, ident)
        if len(set(idents)) != world and not one_gpu_smoke:
            sys.exit("bench.py: ranks share a device: %r" % (idents,))

    ctx = sa.Context(0 if emulate else local_rank, library=library)  # (the emulation has one 'device')
    if not emulate:
        ctx.use_torch_stream()
    if args.segment:
        ctx.set_segment(args.segment)

    def timed(step, steps, warmup, spinup_s=0.0):
        """(wall seconds for `steps` steps = max over ranks, mean launch period on this rank's launch stream, this rank's wall).
        `spinup_s`: seconds of back-to-back steps BEFORE the W warm-up steps.  The board's clock / power management needs
        ~25 ms of load to leave its idle state (profiles/r03w_step_timeline.txt: mp3 0.43 -> 0.31 ms per step, vorbis 0.59 ->
        0.38 over the first 25 ms, flat for the next second); W + K short steps from idle measure that ramp, not the kernel."""
        if spinup_s > 0.0 and not emulate:
            t_spin = time.perf_counter()
            while time.perf_counter() - t_spin < spinup_s:
                for _ in range(8):
                    step()
                sync()
        for _ in range(warmup):
            step()
        sync()
        if world > 1:
            dist.barrier()
        sync()
        # One HIP event pair around the K launches (on the launch stream; the context uses torch's current stream).
        # Per-step event pairs would put ~40 us of signal traffic between consecutive launches.
        if not emulate:
            ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter()
        if not emulate:
            ev0.record()
        for _ in range(steps):
            step()
        if not emulate:
            ev1.record()
        sync()
        mine = time.perf_counter() - t0
        if world > 1:
            dist.barrier()
        sync()
        elapsed = time.perf_counter() - t0
        elapsed = max_over_ranks(elapsed, dist if world > 1 else None, device=red_dev)
        launch_s = (ev0.elapsed_time(ev1) / 1e3 if not emulate else mine) / steps
        return elapsed, launch_s, mine

    step, units, unit_name, alg_bytes, config, kernel, result = make_workload(args.workload, torch, ctx, 1234 + rank, args.scale,
                                                                              args.aac_mix, emulate)
    log("workload built: %s" % config["workload"])
    # the same W + K twice: from idle (reported as `cold_start`), then at the sustained clocks (the line's `value`)
    spin = 0.0 if (args.no_spinup or emulate) else args.spinup_ms / 1e3
    cold = None
    if spin > 0.0:
        sync()
        time.sleep(0.25)  # (an idle board: the workload's construction has just run kernels of its own)
        e_c, l_c, _ = timed(step, args.steps, args.warmup)
        cold = {"ms_per_step": e_c / args.steps * 1e3, "kernel_ms": l_c * 1e3, "steps": args.steps, "warmup": args.warmup,
                "note": "the same W warm-up + K timed steps started from an idle board (clock ramp of ~25 ms, "
                        "profiles/r03w_step_timeline.txt); `value` is measured after %d ms of back-to-back steps" % args.spinup_ms}
        log("cold-start region done: %.3f ms/step" % (e_c / args.steps * 1e3))
    elapsed, launch_s, mine = timed(step, args.steps, args.warmup, spin)
    log("timed region done: %.3f ms/step (device), %.3f ms/step (wall)" % (launch_s * 1e3, elapsed / args.steps * 1e3))
    # the same W + K region again, `--repeats` times back to back: the line's `value` stays the first region (the contract's "exactly
    # K steps"); the repeats say how far one 4 ms region is from the next on this board (`repeats.value_median`, `spread_frac`)
    repeats = None
    if args.repeats > 0 and not emulate:
        per, dev_ms = [elapsed / args.steps * 1e3], [launch_s * 1e3]
        for _ in range(args.repeats):
            e_r, l_r, _ = timed(step, args.steps, args.warmup)
            per.append(e_r / args.steps * 1e3)
            dev_ms.append(l_r * 1e3)
        med = float(np.median(per))
        repeats = {"regions": len(per), "ms_per_step": per, "kernel_ms": dev_ms, "ms_per_step_median": med,
                   "value_median": units * world / (med / 1e3), "value_min": units * world / (max(per) / 1e3),
                   "value_max": units * world / (min(per) / 1e3), "spread_frac": (max(per) - min(per)) / med,
                   "roofline_frac_median": alg_bytes / (float(np.median(dev_ms)) / 1e3) / 1e9 / HBM_PEAK_GBS,
                   "note": "region 0 is the line's own timed region (`value`); the others follow it back to back, W warm-up + K timed steps each"}
        log("repeats done: median %.4f ms/step, spread %.1f %%" % (med, repeats["spread_frac"] * 100))
    verified = None
    if args.no_verify:
        verified = {"skipped": "--no-verify: NOT a measurement of the product (an ablation build's results are wrong on purpose)"}
    elif rank == 0 and hasattr(step, "verify"):
        verified = step.verify()
    if not args.no_verify and rank == 0 and args.workload in ("aac", "mp3", "vorbis", "vorbisf", "vorbisf2"):
        verified = verify_sampled_chains(args.workload, step, torch, sync)  # raises on a mismatch: no line for a wrong result
        log("timed batch verified against the oracle: %d samples" % verified["samples_compared"])
    per_rank_ms = [mine / args.steps * 1e3]
    if world > 1:
        per_rank_ms = [None] * world
        dist.all_gather_object(per_rank_ms, mine / args.steps * 1e3)

    gather = None
    if args.gather and world > 1:
        from symphonia_amd.sharding import timed_all_gather
        secs, nbytes = timed_all_gather(result, dist)
        gather = {"op": "all_gather of every rank's PCM shard (RCCL over xGMI)", "ms": secs * 1e3, "bytes_per_rank": nbytes,
                  "algbw_GBps": nbytes * world / secs / 1e9}

    # the same batch when it starts and ends on ONE rank -- scatter of the input shards from rank 0, the
    # synthesis step, gather of the PCM shards on rank 0.  Reported beside `value` (which is (i): shards resident per GPU).
    exchange = None
    if world > 1 and not args.no_exchange and args.workload in ("aac", "mp3", "vorbis"):
        try:  # an optional leg must never cost the headline line (an error here is reported in the line instead)
            src = workload_input(args.workload, step)
            secs = timed_exchange(src, result, step, dist, sync, device=red_dev, reps=2 if emulate else 3)
            exchange = {"op": "scatter of input shards from rank 0 + synthesis + gather of PCM shards on rank 0 (RCCL over xGMI)",
                        "ms": {k: v * 1e3 for k, v in secs.items()},
                        "bytes_per_rank": {"in": src.numel() * src.element_size(), "out": result.numel() * result.element_size()},
                        "value_inclusive": units * world / secs["total"], "unit": unit_name + "/s"}
        except Exception as e:  # noqa: BLE001
            exchange = {"error": "%s: %s" % (type(e).__name__, e)}

    # BASELINE config 4 is the one the north star phrases as an 8-GPU job (64 streams x 8 ch sharded 8 streams per GPU):
    # with N > 1 and another headline workload, add its line (same timing discipline, fewer steps) beside the headline.
    config4 = None
    if world > 1 and not args.no_config4 and args.workload != "vorbis":
        try:
            st4, units4, unit4, bytes4, cfg4, kernel4, _ = make_workload("vorbis", torch, ctx, 4321 + rank, args.scale, 0.0, emulate)
            e4, l4, _ = timed(st4, max(2, args.steps // 2), 1 if emulate else 2, spin)
            n4 = max(2, args.steps // 2)
            config4 = {"value": units4 * world * n4 / e4, "unit": unit4 + "/s", "ms_per_step": e4 / n4 * 1e3, "steps": n4,
                       "config": cfg4, "roofline_frac_rank0": bytes4 / l4 / 1e9 / HBM_PEAK_GBS, "kernel": kernel4}
        except Exception as e:  # noqa: BLE001
            config4 = {"error": "%s: %s" % (type(e).__name__, e)}

    # The same run's copy ceilings and the other BASELINE configs under the same clock: optional legs, each in
    # its own try so that nothing here can cost the headline line.
    ceiling = None
    if world == 1 and not emulate and not args.no_copy_ceiling:
        try:
            ceiling = copy_ceiling(ctx, torch, 64)
        except Exception as e:  # noqa: BLE001
            ceiling = {"error": "%s: %s" % (type(e).__name__, e)}
        log("copy probes done")
    clocks = None
    if world == 1 and not emulate and not args.no_copy_ceiling:
        clocks = sample_clocks(step, sync)
    others = None
    if world == 1 and args.workload == "aac" and not args.no_others:
        others = {}
        # (key, workload, block-switch probability): the BASELINE configs at their headline settings, then what a REAL stream
        # costs -- AAC with legal window-sequence walks (5 % / 25 % of the long frames start a LONG_START -> EIGHT_SHORT.. ->
        # LONG_STOP run, random window shapes) and MP3 with Start -> Short.. -> End runs (a quarter of them mixed) and random
        # rzero: SURVEY 8d's correctness mixes, timed
        for key, w, mixw in (("mp3", "mp3", 0.0), ("vorbis", "vorbis", 0.0), ("flac", "flac", 0.0), ("alac", "alac", 0.0),
                             ("flac_padded_rows", "flacp", 0.0), ("alac_padded_rows", "alacp", 0.0),
                             ("aac_mix_0.05", "aac", 0.05), ("aac_mix_0.25", "aac", 0.25), ("mp3_mix_0.06", "mp3", 0.06),
                             ("mp3_int16_one_kernel", "mp3q", 0.0), ("mp3_int16_two_kernels", "mp3q2", 0.0),
                             ("vorbis_posts_byte_plane", "vorbisf", 0.0), ("vorbis_posts_f32_spectrum", "vorbisf2", 0.0),
                             ("aac_joint_stereo_on_load", "aacjs", 0.0), ("aac_joint_stereo_two_kernels", "aacjs2", 0.0),
                             ("aac_tns_0.30", "aactns", 0.0)):
            try:
                stw, unitsw, unitw, bytesw, cfgw, kernelw, resw = make_workload(w, torch, ctx, 4321, args.scale, mixw, emulate)
                nw, ww = (8, 2) if w in ("flac", "alac", "flacp", "alacp") else (20, 3)  # (a few milliseconds each for the short ones)
                ew, lw, _ = timed(stw, nw, ww, spin)
                others[key] = {"value": unitsw * nw / ew, "unit": unitw + "/s", "ms_per_step": ew / nw * 1e3, "steps": nw, "warmup": ww,
                               "kernel": kernelw, "kernel_ms": lw * 1e3, "algorithmic_bytes_per_launch": bytesw,
                               "roofline_frac": bytesw / lw / 1e9 / HBM_PEAK_GBS, "workload": cfgw["workload"]}
                if mixw:
                    others[key]["mix"] = cfgw.get("mix")
                if w in ("aac", "mp3", "vorbis", "vorbisf", "vorbisf2"):
                    others[key]["verified"] = verify_sampled_chains(w, stw, torch, sync)
                elif hasattr(stw, "verify"):
                    others[key]["verified"] = stw.verify()
                del stw, resw
            except Exception as e:  # noqa: BLE001
                others[key] = {"error": "%s: %s" % (type(e).__name__, e)}
            torch.cuda.empty_cache()
            log("other workload %s done" % key)

    # N > 1, real GPUs: (a) the deployment shape DESIGN.md section 7 argues for -- every rank stages ITS shard from its own
    # page-locked host memory (per-GPU producers: no rank-0 bottleneck) --, (b) the exchange leg once more through the C ABI.
    # Both last, (b) on a watchdog: nothing here can cost what was measured above.
    producers, exchange_c, hung = None, None, False
    if world > 1 and not emulate and not one_gpu_smoke and args.workload == "aac" and not args.no_host_path:
        try:
            h2h = host_to_host_aac(sa, ctx, torch, result, step.input, units)
            worst = max_over_ranks(h2h["ms"] / 1e3, dist, device=red_dev)
            producers = {"op": "every rank: its shard from page-locked host memory through symaccel_aac_synth_pipelined and back (no inter-GPU traffic)",
                         "ms": worst * 1e3, "value": units * world / worst, "unit": unit_name + "/s", "GBps_each_way_per_rank": h2h["GBps_each_way"]}
        except Exception as e:  # noqa: BLE001
            producers = {"error": "%s: %s" % (type(e).__name__, e)}
    if world > 1 and not host_collectives and not args.no_exchange and args.workload in ("aac", "mp3", "vorbis"):
        exchange_c, hung = guarded(lambda: exchange_c_api(ctx, torch, dist, world, rank, workload_input(args.workload, step), result, step, units,
                                                          unit_name, sync), 150)

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
            "schema": 5,
            "protocol": ("sustained (spinup %d ms of back-to-back steps, then W warm-up + K timed steps); `cold_start` = the same W + K "
                         "from an idle board, the protocol of BENCH_r01 / r02" % args.spinup_ms) if spin > 0.0 else "cold (W + K from an idle board)",
            "spinup_ms": (args.spinup_ms if spin > 0.0 else 0),
            "cold_start": cold,
            "repeats": repeats,
            "verified": verified,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "i32/i64" if args.workload in ("flac", "flacp") else ("i32" if args.workload in ("alac", "alacp") else "f32"),
            "data": "synthetic",
            "config": dict(config, parallelism="chains sharded per GPU, no data-path collective", segment=args.segment or "auto"),
            "ranks": {"world_size": dist.get_world_size() if world > 1 else 1, "backend": (dist.get_backend() if world > 1 else None),
                      "rccl_version": rccl_version(torch) if (world > 1 and not host_collectives) else None,
                      "devices": idents, "ms_per_step_per_rank": per_rank_ms},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": None, "kernel": kernel,
                         "kernel_ms": launch_s * 1e3, "algorithmic_bytes_per_launch": alg_bytes,
                         "kernel_ms_note": "mean launch period over the timed region, HIP events on the launch stream; "
                                           "one kernel launch per step (ping-pong state buffers)"},
            "library": {"path": str(ctx.lib.path), "build": ctx.lib.build_flags()},
        }
        if emulate:
            out["data"] = "synthetic (EMULATED on CPU: control-flow test, not a measurement)"
        try:  # HBM bytes per launch from the committed rocprofv3 PMC passes (profiles/hbm_traffic.json), if measured
            tr = json.loads((ROOT / "profiles" / "hbm_traffic.json").read_text()).get(args.workload)
            if tr and tr["algorithmic_bytes_per_launch"] == alg_bytes:
                out["roofline"]["traffic"] = tr["bytes_per_launch"]
                out["roofline"]["traffic_source"] = tr["source"] + " (a committed rocprofv3 PMC measurement of this command, not taken in this run)"
                # which library the counters were taken on, against the one this line ran: a kernel changed since then is flagged
                have = (ctx.lib.build_flags() or {}).get("source_sha256")
                out["roofline"]["traffic_source_sha256"] = tr.get("source_sha256")
                out["roofline"]["traffic_is_of_this_library"] = bool(have and tr.get("source_sha256") == have)
        except (OSError, ValueError, KeyError):
            pass
        if clocks:
            out["clocks_under_load"] = clocks
        if ceiling:
            out["roofline"]["copy_ceiling"] = ceiling
            best = max((v for k, v in ceiling.items() if k.startswith(("plain", "frames")) and isinstance(v, float)), default=None)
            same = ceiling.get("frames_per_workgroup_walk_256_nt")
            if best:
                out["roofline"]["frac_of_best_copy"] = achieved / best
            if same:
                out["roofline"]["frac_of_same_pattern_copy"] = achieved / same
            out["roofline"]["copy_ceiling_note"] = ("symaccel_probe_copy_device in THIS run: 512 MiB read + 512 MiB written per launch; "
                                                    "frames_per_wavefront_64 = every wavefront streams 64 consecutive 4 KiB frames (the "
                                                    "wavefront walk's pattern); frames_per_workgroup_walk_256 = the four wavefronts of a "
                                                    "workgroup share 256 consecutive frames round-robin, 16 KiB contiguous per step (the "
                                                    "headline kernel's pattern: `frac_of_same_pattern_copy`); `peak` stays the HBM3E spec")
        if others:
            out["other_workloads"] = others
        if args.workload in ("alac", "alacp"):
            out["roofline"]["note"] = "integer-ALU bound (adaptive predictor, ~13 x order operations per sample), not HBM"
        if args.workload in ("flac", "flacp"):
            out["roofline"]["note"] = ("one lane per block: bound by how the 64 row segments of a wavefront's tile spread over the HBM channels -- rows 16 KiB apart "
                                       "0.57, rows at symaccel_row_stride() 0.66 (--workload flacp) -- with the FP64 work (32 exact FMAs per 8 B) underneath (DESIGN.md 4.5)")
        if gather:
            out["collective"] = gather
        if exchange:
            out["exchange"] = exchange
        if config4:
            out["config4_vorbis"] = config4
        if producers:
            out["per_gpu_producers"] = producers
        if exchange_c:
            out["exchange_c_api"] = exchange_c
        if world == 1 and args.workload == "aac" and not args.no_host_path and not emulate:
            out["host_to_host"] = host_to_host_aac(sa, ctx, torch, result, step.input, units)
            try:  # BASELINE config 3 the same way: f32 spectra against the entropy decoder's int16 samples (half the bytes in)
                out["host_to_host_mp3"] = host_to_host_mp3(sa, ctx, torch, 128, 2048, 131072)
            except Exception as e:  # noqa: BLE001
                out["host_to_host_mp3"] = {"error": "%s: %s" % (type(e).__name__, e)}
            try:  # the trait-level figure: S streams through decode() and the cross-stream batcher (short form: S = 256)
                out["decoders"] = decoders_workload(quick=True, lookahead=256)  # (the Rust shim's DEFAULT_LOOKAHEAD)
            except Exception as e:  # noqa: BLE001
                out["decoders"] = {"error": "%s: %s" % (type(e).__name__, e)}
        if world == 1 and args.workload == "mp3" and not args.no_host_path and not emulate:
            try:
                out["host_to_host"] = host_to_host_mp3(sa, ctx, torch, int(step.input.shape[0]), int(step.input.shape[1]), units)
            except Exception as e:  # noqa: BLE001
                out["host_to_host"] = {"error": "%s: %s" % (type(e).__name__, e)}
        if world == 1 and not args.no_cpu_baseline and not emulate:
            # (the int16 MP3 lines are timed against the same CPU restatement as config 3: its synthesis tail; the CPU side of
            # requantize + stereo is a few per cent of that)
            out["cpu_baseline"] = cpu_baseline({"mp3q": "mp3", "mp3q2": "mp3", "vorbisf": "vorbis", "vorbisf2": "vorbis", "aacjs": "aac", "aacjs2": "aac", "aactns": "aac", "flacp": "flac", "alacp": "alac"}.get(args.workload, args.workload))
        print(json.dumps(out), flush=True)
    if hung:
        os._exit(0)  # a leg is still stuck in a collective: the line is out, do not wait for it
    ctx.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
