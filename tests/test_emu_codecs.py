"""Kernel-logic checks (CPU emulation of the HIP sources, see test_emu_core_aac.py) for the MP3,
Vorbis and FLAC paths: bit-exact against the oracle."""
import numpy as np
import pytest

import oracle
from emu_lib import emu_ctx  # noqa: F401
from helpers import bit_equal, flac_carrier_case, flac_extreme_case
from symphonia_amd import FlacPredictor, Mp3Synthesis, VorbisDsp, flac_desc, mp3_side
from symphonia_amd.backend import FLAC_FIXED, FLAC_LPC, FLAC_VERBATIM


def mp3_case(rng, nch, ngr, legal=True):
    xr = (rng.standard_normal((nch, ngr, 576)) * np.exp2(-rng.integers(0, 12, (nch, ngr, 1)))).astype(np.float32)
    bt = np.zeros((nch, ngr), np.uint8)
    mx = np.zeros((nch, ngr), np.uint8)
    for c in range(nch):
        g = 0
        while g < ngr:
            r = rng.random()
            if r < 0.45 or g + 3 > ngr:
                bt[c, g] = 0
                g += 1
            else:  # Start -> Short(s) -> End
                bt[c, g] = 1
                n_short = int(rng.integers(1, 3))
                for k in range(n_short):
                    if g + 1 + k < ngr:
                        bt[c, g + 1 + k] = 2
                        mx[c, g + 1 + k] = rng.random() < 0.3
                if g + 1 + n_short < ngr:
                    bt[c, g + 1 + n_short] = 3
                g += 2 + n_short
    rz = (rng.integers(0, 289, (nch, ngr)) * 2).astype(np.uint16)
    rz[rng.random((nch, ngr)) < 0.2] = 576
    for c in range(nch):
        for g in range(ngr):
            xr[c, g, rz[c, g]:] = 0.0
    return xr, bt, mx, rz


@pytest.mark.parametrize("seg,sr", [(2, 0), (3, 1), (64, 8), (5, 4)])
def test_emu_mp3(emu_ctx, seg, sr):
    rng = np.random.default_rng(10 * seg + sr)
    nch, ngr = 3, 11
    xr, bt, mx, rz = mp3_case(rng, nch, ngr)
    ov = rng.standard_normal((nch, 576)).astype(np.float32)
    # a consistent polyphase state: run the oracle on a warm-up granule from reset
    warm = rng.standard_normal((nch, 2, 576)).astype(np.float32)
    _, _, vv, vf = oracle.mp3_synth(warm, oracle.mp3_side(np.zeros((nch, 2)), np.zeros((nch, 2)), np.full((nch, 2), 576)),
                                    sr, np.zeros((nch, 576), np.float32), np.zeros((nch, 1024), np.float32),
                                    np.array([0, 5, 11], np.int32))
    emu_ctx.set_segment(seg)
    got = Mp3Synthesis(emu_ctx, sr).synth(xr, mp3_side(bt, mx, rz), ov, vv, vf)
    want = oracle.mp3_synth(xr, oracle.mp3_side(bt, mx, rz), sr, ov, vv, vf)
    emu_ctx.set_segment(0)
    assert bit_equal(got[0], want[0]), "pcm"
    assert bit_equal(got[1], want[1]), "overlap"
    assert np.array_equal(got[3], want[3]), "v_front"
    # v_vec rows: identical up to the sign of zero in entry 16 of a row (the reference stores +0.0 there)
    assert bit_equal(got[2], want[2]), "v_vec"


def test_emu_mp3_non_finite_lines(emu_ctx):
    """+-Inf / NaN lines in a few granules: NaN appears in the PCM, the overlap and the polyphase FIFO exactly where the
    reference's operation graph puts it (the FIFO keeps it for 16 time slots), and the halo recompute walks through it."""
    from helpers import equal_mod_nan, sprinkle_specials
    rng = np.random.default_rng(405)
    nch, ngr = 3, 9
    xr, bt, mx, rz = mp3_case(rng, nch, ngr)
    sprinkle_specials(xr, rng, [1, 4, ngr + 2, 2 * ngr + 8])
    z = (np.zeros((nch, 576), np.float32), np.zeros((nch, 1024), np.float32), np.zeros(nch, np.int32))
    want = oracle.mp3_synth(xr, oracle.mp3_side(bt, mx, rz), 0, *z)
    assert np.isnan(want[0]).any() and np.isnan(want[2]).any()
    for seg in (2, 3, 64):
        emu_ctx.set_segment(seg)
        got = Mp3Synthesis(emu_ctx, 0).synth(xr, mp3_side(bt, mx, rz), *z)
        for a, b, what in zip(got[:3], want[:3], ("pcm", "overlap", "v_vec")):
            assert equal_mod_nan(a, np.asarray(b)), (seg, what)
        assert np.array_equal(got[3], want[3])
    emu_ctx.set_segment(0)


def test_emu_mp3_single_granule_and_odd_chain_count(emu_ctx):
    rng = np.random.default_rng(77)
    for nch, ngr in ((1, 1), (1, 2), (5, 3)):
        xr, bt, mx, rz = mp3_case(rng, nch, ngr)
        z = (np.zeros((nch, 576), np.float32), np.zeros((nch, 1024), np.float32), np.zeros(nch, np.int32))
        got = Mp3Synthesis(emu_ctx, 0).synth(xr, mp3_side(bt, mx, rz), *z)
        want = oracle.mp3_synth(xr, oracle.mp3_side(bt, mx, rz), 0, *z)
        for a, b in zip(got, want):
            assert bit_equal(a, np.asarray(b)), (nch, ngr)


def vorbis_case(rng, bs0e, bs1e, nch, nb, p_long=0.6):
    flags = (rng.random((nch, nb)) < p_long).astype(np.uint8)
    prev = rng.integers(-1, 2, nch).astype(np.int32)
    dsp_layout = oracle.vorbis_layout(bs0e, bs1e, flags, prev)
    spec_stride = int(dsp_layout[0][:, -1].max())
    pcm_stride = int(dsp_layout[1][:, -1].max())
    spectra = (rng.standard_normal((nch, spec_stride)) * 0.25).astype(np.float32)
    overlap = rng.standard_normal((nch, (1 << bs1e) // 2)).astype(np.float32)
    return flags, prev, spectra, overlap, pcm_stride


@pytest.mark.parametrize("bs0e,bs1e,seg", [(8, 11, 3), (6, 6, 2), (6, 9, 64), (7, 12, 4), (8, 8, 1)])
def test_emu_vorbis_synth(emu_ctx, bs0e, bs1e, seg):
    rng = np.random.default_rng(bs0e * 100 + bs1e)
    flags, prev, spectra, overlap, pcm_stride = vorbis_case(rng, bs0e, bs1e, 2, 9)
    emu_ctx.set_segment(seg)
    got = VorbisDsp(emu_ctx, bs0e, bs1e).synth(spectra, flags, prev, overlap, pcm_stride)
    emu_ctx.set_segment(0)
    want = oracle.vorbis_synth(bs0e, bs1e, spectra, flags, prev, overlap, pcm_stride)
    assert bit_equal(got[0], want[0]), "pcm"
    assert bit_equal(got[1], want[1]), "overlap"
    assert np.array_equal(got[2], want[2]), "prev flag"


@pytest.mark.parametrize("bs0e,bs1e,nb,p_long,seg", [(7, 10, 150, 0.6, 1000), (6, 9, 140, 0.3, 37), (8, 10, 70, 0.75, 5), (6, 7, 200, 0.5, 64),
                                                      (9, 11, 40, 0.5, 7), (10, 11, 33, 0.2, 1), (11, 11, 20, 0.5, 3), (6, 11, 90, 0.1, 13),
                                                      (7, 7, 77, 0.5, 10), (9, 9, 30, 1.0, 4), (6, 10, 100, 0.0, 33),
                                                      (9, 12, 40, 0.6, 6), (10, 13, 24, 0.5, 5), (12, 12, 12, 0.5, 3), (12, 13, 14, 0.5, 4),
                                                      (13, 13, 7, 0.4, 2), (6, 13, 60, 0.3, 9), (11, 12, 30, 0.7, 1000), (8, 12, 50, 0.2, 3),
                                                      (10, 13, 1, 1.0, 1), (10, 13, 2, 0.5, 1), (9, 13, 3, 0.5, 2), (11, 13, 5, 0.6, 1)])
def test_emu_vorbis_register_pass_kernel_pairs(emu_ctx, bs0e, bs1e, nb, p_long, seg):
    """vorbis_synth_wave2_kernel (every pair except 256 / 2048; blocks of 4096 / 8192 samples on its big-block path): runs longer than a group
    holds (2048 / bs blocks), runs across the 64-block flag masks, every transition, segment halos, equal sizes, chains of one
    flag only, the stale-state fix-up after a short tail, arbitrary incoming overlap."""
    rng = np.random.default_rng(1000 * bs0e + 10 * bs1e + nb)
    nch = 3
    flags = (rng.random((nch, nb)) < p_long).astype(np.uint8)
    flags[0, : min(nb, 45)] = 0                  # a run longer than thirty-two short blocks
    flags[1, nb // 2: nb // 2 + 40] = 1          # ... and a long run of long blocks
    flags[2, nb - min(nb, 50):] = 0              # a short tail longer than most segments (state fix-up)
    prev = rng.integers(-1, 2, nch).astype(np.int32)
    lay = oracle.vorbis_layout(bs0e, bs1e, flags, prev)
    spectra = (rng.standard_normal((nch, int(lay[0][:, -1].max()))) * np.exp2(rng.integers(-4, 4, (nch, 1)))).astype(np.float32)
    overlap = rng.standard_normal((nch, (1 << bs1e) // 2)).astype(np.float32)
    pcm_stride = int(lay[1][:, -1].max())
    emu_ctx.set_segment(seg)
    got = VorbisDsp(emu_ctx, bs0e, bs1e).synth(spectra, flags, prev, overlap, pcm_stride)
    emu_ctx.set_segment(0)
    want = oracle.vorbis_synth(bs0e, bs1e, spectra, flags, prev, overlap, pcm_stride)
    assert bit_equal(got[0], want[0]), "pcm"
    assert bit_equal(got[1], want[1]), "overlap"
    assert np.array_equal(got[2], want[2]), "prev flag"


def vorbis_wave_case(seed, nch, nb, p_long, tail_short=0):
    """256/2048 case with controllable run structure; `tail_short` forces the last blocks short (state fix-up path)."""
    rng = np.random.default_rng(seed)
    flags = (rng.random((nch, nb)) < p_long).astype(np.uint8)
    if tail_short:
        flags[:, nb - tail_short:] = 0
    flags[0, : min(nb, 11)] = 0  # a run longer than the eight-way short transform handles at once
    prev = rng.integers(-1, 2, nch).astype(np.int32)
    lay = oracle.vorbis_layout(8, 11, flags, prev)
    spectra = (rng.standard_normal((nch, int(lay[0][:, -1].max()))) * 0.25).astype(np.float32)
    overlap = rng.standard_normal((nch, 1024)).astype(np.float32)
    return flags, prev, spectra, overlap, int(lay[1][:, -1].max())


@pytest.mark.parametrize("seed,nb,p_long,tail_short,seg", [(1, 14, 0.5, 0, 4), (2, 13, 0.2, 6, 3), (3, 12, 0.9, 1, 5),
                                                            (4, 9, 0.0, 0, 2), (5, 10, 1.0, 0, 1), (6, 16, 0.5, 9, 32),
                                                            (7, 150, 0.6, 0, 1000)])
def test_emu_vorbis_wave_paths(emu_ctx, seed, nb, p_long, tail_short, seg):
    """The 256/2048 wavefront kernel: every transition, runs of short blocks, segment halos, the stale-state fix-up."""
    flags, prev, spectra, overlap, pcm_stride = vorbis_wave_case(seed, 3, nb, p_long, tail_short)
    emu_ctx.set_segment(seg)
    got = VorbisDsp(emu_ctx, 8, 11).synth(spectra, flags, prev, overlap, pcm_stride)
    emu_ctx.set_segment(0)
    want = oracle.vorbis_synth(8, 11, spectra, flags, prev, overlap, pcm_stride)
    assert bit_equal(got[0], want[0]), "pcm"
    assert bit_equal(got[1], want[1]), "overlap"
    assert np.array_equal(got[2], want[2]), "prev flag"


@pytest.mark.parametrize("bs0e,bs1e,seg,tail", [(6, 9, 2, 5), (7, 10, 3, 9), (6, 8, 1, 12)])
def test_emu_vorbis_generic_state_after_short_tail(emu_ctx, bs0e, bs1e, seg, tail):
    """Generic kernel: a chain that ends in more short blocks than a segment holds -- the part of `overlap` the short
    blocks do not rewrite must still be what the last long block left (or the incoming state)."""
    rng = np.random.default_rng(31 + bs1e)
    flags, prev, spectra, overlap, pcm_stride = vorbis_case(rng, bs0e, bs1e, 3, 14)
    flags[:, 14 - tail:] = 0
    flags[2, :] = 0  # a chain with no long block at all keeps the incoming state
    lay = oracle.vorbis_layout(bs0e, bs1e, flags, prev)
    spectra = (rng.standard_normal((3, int(lay[0][:, -1].max()))) * 0.25).astype(np.float32)
    pcm_stride = int(lay[1][:, -1].max())
    emu_ctx.set_segment(seg)
    got = VorbisDsp(emu_ctx, bs0e, bs1e).synth(spectra, flags, prev, overlap, pcm_stride)
    emu_ctx.set_segment(0)
    want = oracle.vorbis_synth(bs0e, bs1e, spectra, flags, prev, overlap, pcm_stride)
    assert bit_equal(got[0], want[0]) and bit_equal(got[1], want[1]) and np.array_equal(got[2], want[2])


@pytest.mark.parametrize("bs0e,bs1e,seg", [(8, 11, 3), (6, 9, 4)])
def test_emu_vorbis_fused_dot_product(emu_ctx, bs0e, bs1e, seg):
    """synth on floor x residue (fused multiply on load) == dot product, then synth -- wave kernel and generic kernel."""
    rng = np.random.default_rng(77 + bs0e)
    flags, prev, floor, overlap, pcm_stride = vorbis_case(rng, bs0e, bs1e, 3, 11)
    residue = rng.standard_normal(floor.shape).astype(np.float32)
    v = VorbisDsp(emu_ctx, bs0e, bs1e)
    emu_ctx.set_segment(seg)
    pf, ov = prev.copy(), overlap.copy()
    pcm = np.zeros((flags.shape[0], pcm_stride), np.float32)
    v.synth_floor_residue(floor, residue, flags, pf, ov, pcm_stride, pcm)
    emu_ctx.set_segment(0)
    want = oracle.vorbis_synth(bs0e, bs1e, floor * residue, flags, prev, overlap, pcm_stride)
    assert bit_equal(pcm, want[0]) and bit_equal(ov, want[1]) and np.array_equal(pf, want[2])


def test_emu_vorbis_helpers(emu_ctx):
    rng = np.random.default_rng(5)
    v = VorbisDsp(emu_ctx, 8, 11)
    n = 1000
    res = rng.standard_normal((4, n)).astype(np.float32)
    res[rng.random((4, n)) < 0.2] = 0.0
    want = res.copy()
    for m, a in ((0, 1), (2, 3), (1, 2)):
        want[m], want[a] = oracle.vorbis_inverse_coupling(want[m], want[a])
    got = res.copy()
    v.inverse_coupling(got, n, [0, 2, 1], [1, 3, 2])
    assert bit_equal(got, want)
    fl = rng.standard_normal(4 * n).astype(np.float32)
    want_f = oracle.vorbis_dot_product(fl, got.ravel())
    v.dot_product(fl, got, 4 * n)
    assert bit_equal(fl, want_f)
    t2 = rng.standard_normal((3, 5 * 128)).astype(np.float32)
    planar = np.empty((3, 5, 128), np.float32)
    v.deinterleave2(t2, planar, 5, 128, 3)
    assert bit_equal(planar, np.stack([oracle.vorbis_deinterleave2(t, 5) for t in t2]))


def test_emu_vorbis_floor1(emu_ctx):
    rng = np.random.default_rng(6)
    v = VorbisDsp(emu_ctx, 8, 11)
    for n, n_posts, mult, count, p_zero in ((1024, 30, 2, 7, 0.3), (128, 9, 1, 7, 0.3), (1024, 65, 4, 7, 0.3), (128, 2, 3, 7, 0.3),
                                             (2048, 65, 1, 70, 0.02), (32, 5, 2, 130, 0.5), (4096, 40, 2, 3, 0.9)):
        xs = [0, n] + rng.permutation(np.arange(1, n))[:n_posts - 2].tolist()
        rr = [256, 128, 86, 64][mult - 1]
        ys = rng.integers(0, rr, size=(count, n_posts)).astype(np.uint32)
        ys[rng.random((count, n_posts)) < p_zero] = 0
        out = np.zeros((count, n), np.float32)
        v.floor1(xs, mult, ys, n, out, count)
        want = np.stack([oracle.vorbis_floor1(xs, y, mult, n) for y in ys])
        assert bit_equal(out, want), (n, n_posts, mult)
        # the dot product of lib.rs:282-292 fused into the curve's store: one rounded multiply per line, out of place and
        # in place on the residue
        res = (rng.standard_normal((count, n)) * np.exp2(rng.integers(-8, 9, (count, n)))).astype(np.float32)
        res[0, ::5] = -0.0
        prod = np.zeros((count, n), np.float32)
        v.floor1(xs, mult, ys, n, prod, count, residue=res)
        assert bit_equal(prod, want * res), (n, n_posts, mult)
        inplace = res.copy()
        v.floor1(xs, mult, ys, n, inplace, count, residue=inplace)
        assert bit_equal(inplace, want * res), (n, n_posts, mult)


def test_emu_vorbis_floor1_random(emu_ctx):
    from helpers import floor1_case
    rng = np.random.default_rng(61)
    v = VorbisDsp(emu_ctx, 8, 11)
    for _ in range(25):
        xs, mult, n, ys = floor1_case(rng)
        out = np.zeros((len(ys), n), np.float32)
        v.floor1(xs, mult, ys, n, out, len(ys))
        want = np.stack([oracle.vorbis_floor1(xs, y, mult, n) for y in ys])
        assert bit_equal(out, want), (n, len(xs), mult, len(ys))


def test_emu_flac_restore(emu_ctx):
    rng = np.random.default_rng(9)
    for blocksize in (1, 31, 64, 100, 192, 4096 // 8):
        nb = 70
        buf = rng.integers(-(1 << 22), 1 << 22, (nb, blocksize)).astype(np.int32)
        kind = rng.integers(0, 3, nb).astype(np.uint8)
        order = np.where(kind == FLAC_FIXED, rng.integers(0, 5, nb), rng.integers(1, 33, nb))
        order = np.minimum(order, blocksize).astype(np.uint8)
        kind[(kind == FLAC_LPC) & (order == 0)] = FLAC_VERBATIM
        shift = rng.integers(0, 16, nb).astype(np.uint8)
        wasted = np.where(rng.random(nb) < 0.2, rng.integers(1, 8, nb), 0).astype(np.uint8)
        coeffs = rng.integers(-(1 << 14), 1 << 14, (nb, 32)).astype(np.int32)
        got = FlacPredictor(emu_ctx).restore(buf, flac_desc(kind, order, shift, wasted), coeffs)
        want = oracle.flac_restore(buf, oracle.flac_desc(kind, order, shift, wasted), coeffs)
        assert np.array_equal(got, want), blocksize


def test_emu_flac_small_orders_only(emu_ctx):
    rng = np.random.default_rng(19)
    nb, blocksize = 64, 130
    buf = rng.integers(-(1 << 15), 1 << 15, (nb, blocksize)).astype(np.int32)
    for hi in (4, 12):
        order = rng.integers(1, hi + 1, nb).astype(np.uint8)
        kind = np.full(nb, FLAC_LPC, np.uint8)
        coeffs = rng.integers(-2000, 2000, (nb, 32)).astype(np.int32)
        shift = np.full(nb, 9, np.uint8)
        got = FlacPredictor(emu_ctx).restore(buf, flac_desc(kind, order, shift, 0 * shift), coeffs)
        want = oracle.flac_restore(buf, oracle.flac_desc(kind, order, shift, 0 * shift), coeffs)
        assert np.array_equal(got, want), hi


@pytest.mark.parametrize("big_coeffs", [False, True])
def test_emu_flac_extreme_ranges(emu_ctx, big_coeffs):
    buf, kind, order, shift, coeffs = flac_extreme_case(23, big_coeffs)
    got = FlacPredictor(emu_ctx).restore(buf, flac_desc(kind, order, shift, 0 * shift), coeffs)
    want = oracle.flac_restore(buf, oracle.flac_desc(kind, order, shift, 0 * shift), coeffs)
    assert np.array_equal(got, want)


@pytest.mark.parametrize("carrier", ["dot2", "dot2x2", "f64"])
def test_emu_flac_carriers_at_their_edges(emu_ctx, carrier):
    """Full-range samples with coefficient sets whose magnitudes sum to about 2^16 (helpers.flac_carrier_case): the reference's i64
    arithmetic bit for bit."""
    buf, kind, order, shift, coeffs = flac_carrier_case(31, carrier)
    got = FlacPredictor(emu_ctx).restore(buf, flac_desc(kind, order, shift, 0 * shift), coeffs)
    want = oracle.flac_restore(buf, oracle.flac_desc(kind, order, shift, 0 * shift), coeffs)
    assert np.array_equal(got, want), np.argwhere(got != want)[:5]


@pytest.mark.parametrize("blocksize,nb", [(64, 64), (100, 70), (192, 6), (31, 130)])
def test_emu_flac_restore_stereo_fused(emu_ctx, blocksize, nb):
    """restore with decorrelation + shift fused into the write-back == restore, then decorrelate, then shift."""
    rng = np.random.default_rng(blocksize + nb)
    buf = rng.integers(-(1 << 20), 1 << 20, (nb, blocksize)).astype(np.int32)
    kind = rng.integers(0, 3, nb).astype(np.uint8)
    order = np.minimum(np.where(kind == FLAC_FIXED, rng.integers(0, 5, nb), rng.integers(1, 33, nb)), blocksize).astype(np.uint8)
    kind[(kind == FLAC_LPC) & (order == 0)] = FLAC_VERBATIM
    shift = rng.integers(0, 16, nb).astype(np.uint8)
    wasted = np.where(rng.random(nb) < 0.2, rng.integers(1, 4, nb), 0).astype(np.uint8)
    coeffs = rng.integers(-(1 << 14), 1 << 14, (nb, 32)).astype(np.int32)
    mode = rng.integers(0, 4, nb // 2).astype(np.uint8)
    got = buf.copy()
    FlacPredictor(emu_ctx).restore_stereo(got, flac_desc(kind, order, shift, wasted), coeffs, mode, 8)
    want = oracle.flac_restore(buf, oracle.flac_desc(kind, order, shift, wasted), coeffs)
    for p in range(nb // 2):
        a, b = oracle.flac_decorrelate(int(mode[p]), want[2 * p], want[2 * p + 1])
        want[2 * p], want[2 * p + 1] = oracle.flac_shl(a, 8), oracle.flac_shl(b, 8)
    assert np.array_equal(got, want)


def test_emu_flac_decorrelate(emu_ctx):
    rng = np.random.default_rng(29)
    n_pairs, bs = 9, 77
    a = rng.integers(-(1 << 24), 1 << 24, (n_pairs, bs)).astype(np.int32)
    b = rng.integers(-(1 << 24), 1 << 24, (n_pairs, bs)).astype(np.int32)
    mode = rng.integers(0, 4, n_pairs).astype(np.uint8)
    ga, gb = a.copy(), b.copy()
    FlacPredictor(emu_ctx).decorrelate(mode, ga, gb, bs, out_shift=8)
    for p in range(n_pairs):
        wa, wb = oracle.flac_decorrelate(int(mode[p]), a[p], b[p])
        assert np.array_equal(ga[p], oracle.flac_shl(wa, 8)) and np.array_equal(gb[p], oracle.flac_shl(wb, 8))
