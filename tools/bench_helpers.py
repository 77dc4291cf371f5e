#!/usr/bin/env python3
"""Throughput of the streaming helper kernels (SURVEY 8a rows a22, a23, a27 and the ALAC mid/side): GB/s of algorithmic
traffic, HIP events around 10 launches each.  Development tool: python tools/bench_helpers.py (on the GPU box)."""
import sys
from pathlib import Path

import numpy as np
import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import symphonia_amd as sa  # noqa: E402


def timeit(fn, reps=10, spinup_s=0.05):
    import time
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()  # sustained clocks: the board needs ~25 ms of load to leave its idle state
    while time.perf_counter() - t0 < spinup_s:
        for _ in range(4):
            fn()
        torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps / 1e3


def main():
    ctx = sa.Context(0)
    ctx.use_torch_stream()
    v = sa.VorbisDsp(ctx, 8, 11)
    n, ch, blocks = 1024, 8, 32768  # 32768 blocks x 8 channels x 1024 lines = 1 GiB of residue
    res = torch.randn((blocks * ch, n), device="cuda")
    flo = torch.randn(blocks * ch * n, device="cuda")
    out = {}
    t = timeit(lambda: v.dot_product(flo, res, blocks * ch * n))
    out["vorbis dot product (floor *= residue)"] = 3 * flo.numel() * 4 / t
    pairs_m, pairs_a = list(range(0, 64, 2)), list(range(1, 64, 2))
    big = torch.randn((64, 1 << 22), device="cuda")
    t = timeit(lambda: v.inverse_coupling(big, 1 << 22, pairs_m, pairs_a))
    out["vorbis inverse coupling (32 pairs x 4 Mi lines)"] = 4 * big.numel() * 4 / 2 / t * 1.0
    t2 = torch.randn((4096, ch * n), device="cuda")
    planar = torch.empty((4096, ch, n), device="cuda")
    t = timeit(lambda: v.deinterleave2(t2, planar, ch, n, 4096))
    out["vorbis residue type-2 de-interleave"] = 2 * t2.numel() * 4 / t
    xs = [0, n] + list(range(16, 1008, 16))
    ys = torch.randint(0, 128, (262144, len(xs)), device="cuda", dtype=torch.int32)  # same bits as u32 for values < 2^31
    fl = torch.empty((262144, n), device="cuda")
    t = timeit(lambda: v.floor1(xs, 2, ys, n, fl, 262144))
    out["vorbis floor-1 render (64 posts -> 1024 lines)"] = fl.numel() * 4 / t
    rng = np.random.default_rng(5)
    xs_irr = [0, n] + sorted(rng.permutation(np.arange(1, n))[:62].tolist())  # 64 posts at irregular x (the usual case)
    ys_irr = torch.randint(0, 128, (262144, len(xs_irr)), device="cuda", dtype=torch.int32)
    t = timeit(lambda: v.floor1(xs_irr, 2, ys_irr, n, fl, 262144))
    out["vorbis floor-1 render, irregular post positions"] = fl.numel() * 4 / t
    rs = torch.randn((262144, n), device="cuda")
    t = timeit(lambda: v.floor1(xs, 2, ys, n, fl, 262144, residue=rs))
    out["vorbis floor-1 x residue fused (read 4 B + write 4 B per line; %.2f ms)" % (t * 1e3)] = 2 * fl.numel() * 4 / t
    t = timeit(lambda: (v.floor1(xs, 2, ys, n, fl, 262144), v.dot_product(fl.view(-1), rs, fl.numel())))
    out["vorbis floor-1, then dot product: two launches (%.2f ms)" % (t * 1e3)] = 2 * fl.numel() * 4 / t
    fp = sa.FlacPredictor(ctx)
    a = torch.randint(-1000, 1000, (65536, 4096), device="cuda", dtype=torch.int32)
    b = torch.randint(-1000, 1000, (65536, 4096), device="cuda", dtype=torch.int32)
    mode = torch.randint(0, 4, (65536,), device="cuda", dtype=torch.uint8)
    t = timeit(lambda: fp.decorrelate(mode, a, b, 4096, 8))
    out["flac decorrelate + shift (65536 pairs x 4096)"] = 4 * a.numel() * 4 / t
    ap = sa.AlacPredictor(ctx)
    w = torch.randint(1, 4, (65536,), device="cuda", dtype=torch.int32)
    sh = torch.full((65536,), 2, device="cuda", dtype=torch.uint8)
    t = timeit(lambda: ap.mid_side(w, sh, a, b))
    out["alac mid/side (65536 pairs x 4096)"] = 4 * a.numel() * 4 / t
    rq = sa.Mp3Requantize(ctx, 0)
    ngc = 262144  # config 3: 131072 granules x 2 channels
    rng = np.random.default_rng(0)
    quant = torch.from_numpy(np.rint(rng.laplace(0, 6, (ngc, 576))).astype(np.int16)).cuda()
    dnp = np.zeros(ngc, sa.MP3_REQUANT_DTYPE)
    dnp["global_gain"], dnp["rzero"] = rng.integers(120, 200, ngc), rng.integers(300, 577, ngc)
    dnp["scalefacs"] = rng.integers(0, 16, (ngc, 39))
    dnp["block_type"] = rng.choice([0, 0, 0, 0, 0, 0, 0, 1, 2, 3], ngc)
    desc = torch.from_numpy(dnp.view(np.uint8).reshape(ngc, 52)).cuda()
    xr = torch.empty((ngc, 576), device="cuda")
    t = timeit(lambda: rq.requantize(quant, desc, xr))
    out["mp3 requantize (262144 granule-channels, i16 -> f32)"] = ngc * (576 * 6 + 52) / t
    st = sa.Mp3Stereo(ctx, 0)
    xr2 = torch.randn((128, 2048, 576), device="cuda") * 0.1   # config 3: 64 stereo streams x 2048 granules
    xr2[1::2, :, 342:] = 0                                       # channel 1 ends at band 20: the top two bands are intensity coded
    sdn = np.zeros((64, 2048), sa.MP3_STEREO_DTYPE)
    sdn["flags"], sdn["rzero0"], sdn["rzero1"] = 3 | 4, 576, 342
    sdn["scalefacs1"] = rng.integers(0, 7, (64, 2048, 39))
    sdesc = torch.from_numpy(sdn.view(np.uint8).reshape(64, 2048, 48)).cuda()
    spairs = torch.arange(128, dtype=torch.int32, device="cuda").reshape(64, 2)
    t = timeit(lambda: st.stereo(xr2, spairs, sdesc))
    out["mp3 joint stereo (131072 granules, m/s + intensity)"] = 131072 * (4 * 2304 + 48) / t
    quant2 = quant.reshape(128, 2048, 576)
    quant2[1::2, :, 342:] = 0
    t = timeit(lambda: st.requantize_stereo(quant2, desc.reshape(128, 2048, 52), spairs, sdesc, xr2))
    out["mp3 requantize + stereo fused (131072 granules; %.3f ms)" % (t * 1e3)] = 131072 * (2 * (1152 + 2304 + 52) + 48) / t
    syn = sa.Mp3Synthesis(ctx, 0)
    side = torch.from_numpy(sa.mp3_side(np.zeros((128, 2048), np.uint8), np.zeros((128, 2048), np.uint8),
                                        np.full((128, 2048), 576)).view(np.uint8).reshape(128, 2048, 4)).cuda()
    sst = [torch.zeros((128, 576), device="cuda"), torch.zeros((128, 1024), device="cuda"), torch.zeros(128, dtype=torch.int32, device="cuda")]
    pcm2 = torch.empty_like(xr2)
    t3 = timeit(lambda: (st.requantize_stereo(quant2, desc.reshape(128, 2048, 52), spairs, sdesc, xr2),
                         syn.synth(xr2, side, sst[0], sst[1], sst[2], pcm2)))
    out["mp3 requantize+stereo, then synth: two launches (%.3f ms)" % (t3 * 1e3)] = 131072 * (2 * (1152 + 52 + 2304) + 48) / t3
    # AAC spectral tools at config-2 size: 64 pairs x 1024 frames, every band of every frame mid/side coded;
    # one order-12 TNS filter over lines 96..896 in every channel-frame
    swb_long = [0, 4, 8, 12, 16, 20, 24, 28, 32, 36, 40, 48, 56, 64, 72, 80, 88, 96, 108, 120, 132, 144, 160, 176, 196, 216,
                240, 264, 292, 320, 352, 384, 416, 448, 480, 512, 544, 576, 608, 640, 672, 704, 736, 768, 800, 832, 864, 896,
                928, 1024]
    swb_short = [0, 4, 8, 12, 16, 20, 28, 36, 44, 56, 68, 80, 96, 112, 128]
    tools = sa.AacSpectralTools(ctx, swb_long, swb_short)
    coeffs = torch.randn((128, 1024, 1024), device="cuda") * 0.01
    pairs = torch.arange(128, dtype=torch.int32, device="cuda").reshape(64, 2)
    jd = np.zeros((64, 1024), sa.AAC_JS_DTYPE)
    jd["num_windows"], jd["max_sfb"], jd["mode"] = 1, 49, 1
    jdesc = torch.from_numpy(jd.view(np.uint8).reshape(64, 1024, 644)).cuda()
    t = timeit(lambda: tools.joint_stereo(coeffs, pairs, jdesc))
    out["aac mid/side, all bands (64 pairs x 1024 frames)"] = (4 * 65536 * 4096 + jd.nbytes) / t
    tf = np.zeros(131072, sa.AAC_TNS_DTYPE)
    tf["frame"], tf["start"], tf["end"], tf["order"] = np.arange(131072), 96, 896, 12
    tf["lpc"][:, :12] = (rng.standard_normal((131072, 12)) * 0.05).astype(np.float32)
    tfilt = torch.from_numpy(tf.view(np.uint8).reshape(-1, 92)).cuda()
    t = timeit(lambda: tools.tns(coeffs, tfilt), reps=3)
    out["aac TNS order 12 x 800 lines in 131072 frames (%.2f ms)" % (t * 1e3)] = 131072 * 800 * 8 / t
    # the same filters with their ranges staggered (start 0, 32, ... 224 in turn): the lanes of a wavefront are no longer at the
    # same offset of their 4 KiB frames at the same time
    tf["start"] = 32 * (np.arange(131072) % 8)
    tf["end"] = tf["start"] + 800
    tfilt2 = torch.from_numpy(tf.view(np.uint8).reshape(-1, 92)).cuda()
    t = timeit(lambda: tools.tns(coeffs, tfilt2), reps=3)
    out["aac TNS, ranges staggered by 32 lines (%.2f ms)" % (t * 1e3)] = 131072 * 800 * 8 / t
    tf["start"] = 32 * ((np.arange(131072) // 64) % 8)
    tf["end"] = tf["start"] + 800
    tfilt3 = torch.from_numpy(tf.view(np.uint8).reshape(-1, 92)).cuda()
    t = timeit(lambda: tools.tns(coeffs, tfilt3), reps=3)
    out["aac TNS, ranges staggered per wavefront (%.2f ms)" % (t * 1e3)] = 131072 * 800 * 8 / t
    for k, gbps in out.items():
        print("%-52s %8.1f GB/s  (%.1f %% of 8 TB/s)" % (k, gbps / 1e9, gbps / 8e12 * 100))
    ctx.close()


if __name__ == "__main__" and "--core" not in sys.argv and "--vorbis-fused" not in sys.argv and "--flac-fused" not in sys.argv:
    main()


def core_transforms():
    """Generic Imdct / Fft kernels (any power-of-two size): algorithmic GB/s."""
    ctx = sa.Context(0)
    ctx.use_torch_stream()
    for n in (32, 64, 128, 256, 512, 1024, 2048, 4096, 8192):
        count = (1 << 27) // n  # 512 MiB of spectra
        spec = torch.randn((count, n), device="cuda")
        out = torch.empty((count, 2 * n), device="cuda")
        im = sa.Imdct(ctx, n, 1.0 / (2 * n))
        t = timeit(lambda: im.imdct(spec, out))
        print("Imdct n=%-5d x %-8d %8.1f GB/s (in + out)  %.3f ms" % (n, count, 3 * spec.numel() * 4 / t / 1e9, t * 1e3))
    for n in (16, 64, 128, 256, 512, 1024, 2048, 4096):
        count = (1 << 26) // n  # 512 MiB of complex values
        x = torch.randn((count, n, 2), device="cuda")
        y = torch.empty_like(x)
        f = sa.Fft(ctx, n)
        t = timeit(lambda: f.fft(x, y))
        print("Fft   n=%-5d x %-8d %8.1f GB/s (in + out)  %.3f ms" % (n, count, 2 * x.numel() * 4 / t / 1e9, t * 1e3))
    ctx.close()


if __name__ == "__main__" and "--core" in sys.argv:
    core_transforms()


def vorbis_fused():
    """Config-4 shard: dot product kernel + synth versus synth with the dot product fused."""
    ctx = sa.Context(0)
    ctx.use_torch_stream()
    nch, nb = 64, 4096
    rng = np.random.default_rng(4)
    flags = np.zeros((nch, nb), np.uint8)
    cur = np.ones(nch, bool)
    for b in range(nb):
        r = rng.random(nch)
        cur = np.where(cur, r < 0.9, r >= 0.7)
        flags[:, b] = cur
    v = sa.VorbisDsp(ctx, 8, 11)
    so, po = v.layout(flags, np.full(nch, -1))
    spec_stride, pcm_stride = int(so[:, -1].max()), int(po[:, -1].max())
    floor = torch.randn((nch, spec_stride), device="cuda")
    residue = torch.randn((nch, spec_stride), device="cuda")
    work = torch.empty_like(floor)
    d_flags = torch.from_numpy(flags).cuda()
    prev = torch.full((nch,), -1, dtype=torch.int32, device="cuda")
    overlap = torch.zeros((nch, 1024), device="cuda")
    pcm = torch.zeros((nch, pcm_stride), device="cuda")

    def separate():
        work.copy_(floor)  # the dot product is in place on the floor (lib.rs:289-291)
        v.dot_product(work, residue, work.numel())
        prev.fill_(-1)
        v.synth(work, d_flags, prev, overlap, pcm_stride, pcm)

    def separate_no_copy():
        v.dot_product(work, residue, work.numel())
        prev.fill_(-1)
        v.synth(work, d_flags, prev, overlap, pcm_stride, pcm)

    def fused():
        prev.fill_(-1)
        v.synth_floor_residue(floor, residue, d_flags, prev, overlap, pcm_stride, pcm)

    print("vorbis config-4 shard: dot product + synth  %.3f ms   fused  %.3f ms" % (timeit(separate_no_copy) * 1e3, timeit(fused) * 1e3))
    ctx.close()


if __name__ == "__main__" and "--vorbis-fused" in sys.argv:
    vorbis_fused()


def flac_fused():
    """Config-5 step: restore + decorrelate/shift as two passes versus the fused kernel."""
    ctx = sa.Context(0)
    ctx.use_torch_stream()
    nb, bs = 131072, 4096
    buf = torch.randint(-(1 << 12), 1 << 12, (nb, bs), device="cuda", dtype=torch.int32)
    desc = torch.from_numpy(sa.flac_desc(np.full(nb, 2), np.full(nb, 32), np.full(nb, 12), np.zeros(nb)).view(np.uint8).reshape(nb, 4)).cuda()
    co = torch.randint(-40, 40, (nb, 32), device="cuda", dtype=torch.int32)
    co[:, 0] = 6553
    co[:, 1] = -2867
    mode = torch.randint(0, 4, (nb // 2,), device="cuda", dtype=torch.uint8)
    fp = sa.FlacPredictor(ctx)
    pairs = buf.view(nb // 2, 2, bs)

    def separate():
        fp.restore(buf, desc, co)
        # channel planes of a pair are rows 2p / 2p+1: decorrelate wants ch0[pair][bs], ch1[pair][bs] -> two strided views
        # are not contiguous, so the two-pass pipeline is timed on contiguous halves (same byte count)
        fp.decorrelate(mode, buf[: nb // 2], buf[nb // 2:], bs, 8)

    def fused():
        fp.restore_stereo(buf, desc, co, mode, 8)

    print("flac %d blocks: restore + decorrelate  %.3f ms   fused  %.3f ms" % (nb, timeit(separate, 5) * 1e3, timeit(fused, 5) * 1e3))
    ctx.close()


if __name__ == "__main__" and "--flac-fused" in sys.argv:
    flac_fused()
