"""Randomised shape / segment sweep of the chain kernels on the MI355X against the oracle: chains x frames x segment
length combinations no hand-picked test lists (odd counts, one-frame chains, segments longer than the chain, a last
workgroup that is only partly populated ...).  SYM_FUZZ_ITERS scales the number of random cases (default 6 per codec)."""
import os

import numpy as np
import pytest

import oracle
from helpers import bit_equal

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")
ITERS = int(os.environ.get("SYM_FUZZ_ITERS", "6"))


@pytest.fixture(scope="module")
def ctx():
    if not torch.cuda.is_available():
        pytest.fail("no GPU visible")
    from symphonia_amd import Context
    c = Context(0)
    yield c
    c.close()


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def host(t):
    torch.cuda.synchronize()
    return t.cpu().numpy()


def shape(rng):
    nch = int(rng.choice([1, 2, 3, 5, 8, 13, 31]))
    n = int(rng.choice([1, 2, 3, 4, 7, 16, 33, 64, 97]))
    seg = int(rng.choice([0, 1, 2, 3, 5, 8, 31, 64, 1000]))
    return nch, n, seg


@pytest.mark.parametrize("it", range(ITERS))
def test_fuzz_aac(ctx, it):
    from symphonia_amd import AacDsp
    from test_gpu_parity import aac_case
    rng = np.random.default_rng(1000 + it)
    nch, nfr, seg = shape(rng)
    coeffs, side, delay = aac_case(2000 + it, nch, nfr, only_long=bool(it % 3 == 0))
    ctx.set_segment(seg)
    d_delay = dev(delay)
    pcm = host(AacDsp(ctx).synth(dev(coeffs), dev(side), d_delay))
    ctx.set_segment(0)
    wp, wd = oracle.aac_synth(coeffs, side, delay)
    assert bit_equal(pcm, wp) and bit_equal(host(d_delay), wd), (nch, nfr, seg)


@pytest.mark.parametrize("it", range(ITERS))
def test_fuzz_mp3(ctx, it):
    from symphonia_amd import Mp3Synthesis, mp3_side
    from test_gpu_parity import mp3_case
    rng = np.random.default_rng(3000 + it)
    nch, ngr, seg = shape(rng)
    sr = int(rng.integers(0, 9))
    xr, bt, mx, rz, ov, vv, vf = mp3_case(4000 + it, nch, ngr)
    side = mp3_side(bt, mx, rz)
    d_ov, d_vv, d_vf = dev(ov), dev(vv), dev(vf)
    ctx.set_segment(seg)
    pcm = host(Mp3Synthesis(ctx, sr).synth(dev(xr), dev(side.view(np.uint8).reshape(nch, ngr, 4)), d_ov, d_vv, d_vf))
    ctx.set_segment(0)
    want = oracle.mp3_synth(xr, oracle.mp3_side(bt, mx, rz), sr, ov, vv, vf)
    assert bit_equal(pcm, want[0]) and bit_equal(host(d_ov), want[1]) and bit_equal(host(d_vv), want[2]), (nch, ngr, seg, sr)
    assert np.array_equal(host(d_vf), want[3])


@pytest.mark.parametrize("it", range(ITERS))
def test_fuzz_vorbis(ctx, it):
    from symphonia_amd import VorbisDsp
    from test_emu_codecs import vorbis_case, vorbis_wave_case
    rng = np.random.default_rng(5000 + it)
    nch, nb, seg = shape(rng)
    if it % 2 == 0:  # the 256 / 2048 wavefront kernel
        bs0e, bs1e = 8, 11
        flags, prev, spectra, overlap, pcm_stride = vorbis_wave_case(6000 + it, nch, nb, float(rng.choice([0.0, 0.3, 0.75, 1.0])),
                                                                    int(rng.integers(0, nb)))
    else:
        bs0e = int(rng.integers(6, 11))
        bs1e = int(rng.integers(bs0e, 14))  # up to 8192-sample blocks (lib.rs:404-406)
        flags, prev, spectra, overlap, pcm_stride = vorbis_case(rng, bs0e, bs1e, nch, nb)
    d_prev, d_ov = dev(prev), dev(overlap)
    ctx.set_segment(seg)
    pcm = host(VorbisDsp(ctx, bs0e, bs1e).synth(dev(spectra), dev(flags), d_prev, d_ov, pcm_stride))
    ctx.set_segment(0)
    want = oracle.vorbis_synth(bs0e, bs1e, spectra, flags, prev, overlap, pcm_stride)
    assert bit_equal(pcm, want[0]) and bit_equal(host(d_ov), want[1]), (bs0e, bs1e, nch, nb, seg)
    assert np.array_equal(host(d_prev), want[2])


@pytest.mark.parametrize("it", range(ITERS))
def test_fuzz_vorbis_floor_y(ctx, it):
    """symaccel_vorbis_synth_fy_*: random block-size pairs (every kernel: 256 / 2048, the group kernel, 4096- and 8192-sample blocks),
    batch shapes and segment lengths; spectrum = FLOOR1_INVERSE_DB_TABLE[y] * residue (floor.rs:822, lib.rs:289-291)."""
    from symphonia_amd import VorbisDsp
    from test_vorbis_floor_y import db_table, fy_case
    rng = np.random.default_rng(5500 + it)
    nch, nb, seg = shape(rng)
    if it % 3 == 0:
        bs0e, bs1e = 8, 11
    else:
        bs0e = int(rng.integers(6, 12))
        bs1e = int(rng.integers(bs0e, 14))
    if bs1e > 11:
        nb = min(nb, 12)
    flags, prev, residue, overlap, pcm_stride, ypl = fy_case(rng, bs0e, bs1e, nch, nb)
    d_prev, d_ov = dev(prev), dev(overlap)
    pcm = torch.zeros((nch, pcm_stride), device="cuda")
    ctx.set_segment(seg)
    VorbisDsp(ctx, bs0e, bs1e).synth_floor_y(dev(ypl), dev(residue), dev(flags), d_prev, d_ov, pcm_stride, pcm)
    ctx.set_segment(0)
    want = oracle.vorbis_synth(bs0e, bs1e, db_table()[ypl] * residue, flags, prev, overlap, pcm_stride)
    assert bit_equal(host(pcm), want[0]) and bit_equal(host(d_ov), want[1]), (bs0e, bs1e, nch, nb, seg)
    assert np.array_equal(host(d_prev), want[2])


@pytest.mark.parametrize("it", range(ITERS))
def test_fuzz_flac_alac(ctx, it):
    from symphonia_amd import AlacPredictor, FlacPredictor, alac_desc, flac_desc
    rng = np.random.default_rng(7000 + it)
    nb = int(rng.choice([1, 2, 63, 64, 65, 130, 257]))
    bs = int(rng.choice([1, 2, 31, 32, 33, 64, 100, 576, 1152, 4096, 4608]))
    kind = rng.integers(0, 3, nb).astype(np.uint8)
    order = np.where(kind == 1, rng.integers(0, 5, nb), rng.integers(1, 33, nb))
    order = np.minimum(order, bs).astype(np.uint8)
    kind[(kind == 2) & (order == 0)] = 0
    shift = rng.integers(0, 16, nb).astype(np.uint8)
    wasted = np.where(rng.random(nb) < 0.2, rng.integers(1, 8, nb), 0).astype(np.uint8)
    buf = rng.integers(-(1 << 23), 1 << 23, (nb, bs)).astype(np.int32)
    coeffs = rng.integers(-(1 << 14), 1 << 14, (nb, 32)).astype(np.int32)
    d_buf = dev(buf)
    FlacPredictor(ctx).restore(d_buf, dev(flac_desc(kind, order, shift, wasted).view(np.uint8).reshape(nb, 4)), dev(coeffs))
    assert bit_equal(host(d_buf), oracle.flac_restore(buf, oracle.flac_desc(kind, order, shift, wasted), coeffs)), (nb, bs)
    from test_alac import alac_case
    res, mode, aorder, ashift, bps, acoef = alac_case(8000 + it, nb, bs)
    d_res = dev(res)
    AlacPredictor(ctx).predict(d_res, dev(alac_desc(mode, aorder, ashift, bps).view(np.uint8).reshape(nb, 4)), dev(acoef))
    assert bit_equal(host(d_res), oracle.alac_predict(res, oracle.alac_desc(mode, aorder, ashift, bps), acoef)), (nb, bs)
    # the instantiation classes of the ALAC kernel: uniform channel depth (24-bit multiplies below 24 bits) x orders <= 8 / any
    depth = int(rng.choice([16, 20, 23, 24]))
    small = bool(rng.integers(0, 2))
    res2 = rng.integers(-(1 << (depth - 2)), 1 << (depth - 2), (nb, bs)).astype(np.int32)
    res2[::5] = rng.integers(-(1 << 31), 1 << 31, (len(res2[::5]), bs))
    res2[:, 0] = rng.integers(-(1 << 15), 1 << 15, nb)
    order2 = (rng.integers(1, 9, nb) if small else rng.choice([4, 8, 12, 16, 31], nb)).astype(np.uint8)
    d2 = (rng.choice([0, 0, 15], nb).astype(np.uint8), order2, rng.integers(0, 12, nb).astype(np.uint8), np.full(nb, depth, np.uint8))
    coef2 = rng.integers(-(1 << 15), 1 << 15, (nb, 32)).astype(np.int32)
    d_res2 = dev(res2)
    AlacPredictor(ctx).predict(d_res2, dev(alac_desc(*d2).view(np.uint8).reshape(nb, 4)), dev(coef2))
    assert bit_equal(host(d_res2), oracle.alac_predict(res2, oracle.alac_desc(*d2), coef2)), (nb, bs, depth, small)


@pytest.mark.parametrize("it", range(ITERS))
def test_fuzz_flac_alac_strided(ctx, it):
    """the strided entry points (ABI 8) at random pitches -- the recommended one, aligned and unaligned pads, none --, plain and fused stereo:
    rows == the oracle on the compact rows, the padding untouched"""
    from symphonia_amd import AlacPredictor, FlacPredictor, alac_desc, flac_desc
    from test_alac import alac_case
    from test_row_stride import alac_want, check, flac_case, flac_want, padded
    rng = np.random.default_rng(9000 + it)
    nb = int(rng.choice([2, 62, 64, 66, 130, 258]))
    bs = int(rng.choice([1, 31, 32, 33, 100, 576, 1024, 1152, 2048, 4096]))
    stride = int(rng.choice([bs, ctx.lib.dll.symaccel_row_stride(bs), bs + 1, bs + 4, bs + 64, bs + int(rng.integers(1, 200))]))
    fused = bool(rng.integers(0, 2))
    buf, kind, order, shift, wasted, coeffs, mode = flac_case(9100 + it, nb, bs)
    d = dev(padded(buf, stride))
    FlacPredictor(ctx).restore_strided(d, dev(flac_desc(kind, order, shift, wasted).view(np.uint8).reshape(nb, 4)), dev(coeffs), bs,
                                       pair_mode=dev(mode) if fused else None, out_shift=8 if fused else 0)
    check(host(d), flac_want(buf, kind, order, shift, wasted, coeffs, mode if fused else None, 8), bs)
    res, amode, aorder, ashift, bps, acoef = alac_case(9200 + it, nb, bs)
    weight, msh = rng.integers(-3, 4, nb // 2).astype(np.int32), rng.integers(0, 32, nb // 2).astype(np.uint8)
    d = dev(padded(res, stride))
    AlacPredictor(ctx).predict_strided(d, dev(alac_desc(amode, aorder, ashift, bps).view(np.uint8).reshape(nb, 4)), dev(acoef), bs,
                                       dev(weight) if fused else None, dev(msh) if fused else None)
    check(host(d), alac_want(res, amode, aorder, ashift, bps, acoef, weight if fused else None, msh if fused else None), bs)


@pytest.mark.parametrize("it", range(ITERS))
def test_fuzz_mp3_front(ctx, it):
    """requantize (+ stereo, separately and fused) for random shapes / sample rates."""
    from symphonia_amd import Mp3Requantize, Mp3Stereo
    from test_mp3_stereo import fused_case
    rng = np.random.default_rng(9000 + it)
    sr = int(rng.integers(0, 9))
    n_pairs, granules = int(rng.choice([1, 2, 5, 17])), int(rng.choice([1, 2, 3, 16, 33]))
    q, rd, pairs, sd, want, mono = fused_case(9100 + it, sr, n_pairs, granules)
    d_q = dev(q)
    d_rd = dev(rd.view(np.uint8).reshape(rd.shape + (52,)))
    d_sd = dev(sd.view(np.uint8).reshape(sd.shape + (48,)))
    d_pairs = dev(pairs)
    paired = sorted(pairs.reshape(-1))
    # two launches
    xr = Mp3Requantize(ctx, sr).requantize(d_q, d_rd)
    Mp3Stereo(ctx, sr).stereo(xr, d_pairs, d_sd)
    assert bit_equal(host(xr)[paired], want[paired]), (sr, n_pairs, granules)
    # one launch
    xr2 = torch.full(q.shape, 7.0, device="cuda")
    Mp3Stereo(ctx, sr).requantize_stereo(d_q, d_rd, d_pairs, d_sd, xr2)
    got = host(xr2)
    assert bit_equal(got[paired], want[paired]) and (got[mono] == 7.0).all(), (sr, n_pairs, granules)


@pytest.mark.parametrize("it", range(ITERS))
def test_fuzz_aac_tools(ctx, it):
    from symphonia_amd import AacSpectralTools
    import test_aac_tools as T
    rng = np.random.default_rng(9500 + it)
    n_pairs, frames = int(rng.choice([1, 2, 7])), int(rng.choice([1, 3, 20]))
    chains = 2 * n_pairs + 1
    coeffs = rng.standard_normal((chains, frames, 1024)).astype(np.float32)
    order = rng.permutation(chains)
    pairs = np.array([[order[2 * p], order[2 * p + 1]] for p in range(n_pairs)], np.int32)
    desc = np.array([[T.js_frame(rng, short=bool(rng.integers(0, 3) == 0)) for _ in range(frames)] for _ in pairs])
    _, filt = T.tns_case(rng, chains * frames, int(rng.choice([1, 10, 200])))
    want_js = T.js_reference(coeffs, pairs, desc)
    want = T.tns_reference(want_js.reshape(-1, 1024), filt, chains * frames).reshape(coeffs.shape)
    tools = AacSpectralTools(ctx, T.SWB_LONG, T.SWB_SHORT)
    d = dev(coeffs)
    tools.joint_stereo(d, dev(pairs), dev(desc.view(np.uint8).reshape(n_pairs, frames, 644)))
    assert bit_equal(host(d), want_js), (n_pairs, frames)
    tools.tns(d, dev(filt.view(np.uint8).reshape(-1, 92)))
    assert bit_equal(host(d), want), (n_pairs, frames, len(filt))


@pytest.mark.parametrize("it", range(4 * ITERS))
def test_fuzz_vorbis_floor1(ctx, it):
    from symphonia_amd import VorbisDsp
    from helpers import floor1_case
    rng = np.random.default_rng(7000 + it)
    xs, mult, n, ys = floor1_case(rng)
    v = VorbisDsp(ctx, 8, 11)
    out = torch.zeros((len(ys), n), dtype=torch.float32, device="cuda")
    v.floor1(xs, mult, dev(ys), n, out, len(ys))
    curve = np.stack([oracle.vorbis_floor1(xs, y, mult, n) for y in ys])
    assert bit_equal(host(out), curve), (n, len(xs), mult, len(ys))
    res = (rng.standard_normal(curve.shape) * np.exp2(rng.integers(-8, 9, curve.shape))).astype(np.float32)
    d_res = dev(res)
    v.floor1(xs, mult, dev(ys), n, d_res, len(ys), residue=d_res)
    assert bit_equal(host(d_res), curve * res), (n, len(xs), mult, len(ys))
