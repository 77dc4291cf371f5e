"""ALAC predictor (symphonia-codec-alac/src/lib.rs:165-264, 664-671).

The reference has no unit tests or vectors for it -- PARITY UNPINNED BY THE REFERENCE -- so the oracle is pinned by an
independent forward encoder written here (pure Python, textbook form of the adaptive predictor): residuals produced by
running the predictor over known PCM must decode back to that PCM exactly.  The kernel is then checked bit-for-bit
against the oracle: in CPU emulation (logic) and, gpu-marked, on the MI355X."""
import numpy as np
import pytest

import oracle
from emu_lib import emu_ctx  # noqa: F401


def sext(v, bits):
    v &= (1 << bits) - 1
    return v - (1 << bits) if v >> (bits - 1) else v


def w32(v):
    return sext(v, 32)


def alac_encode(pcm, mode, order, shift, bps, coeffs):
    """Forward pass: the residual stream whose decode is `pcm` (values must fit `bps` bits)."""
    n = len(pcm)
    co = [int(c) for c in coeffs[:order]] + [0] * (32 - order)
    out = [int(v) for v in pcm]            # decoder-side reconstruction (== pcm by construction)
    stage = [0] * n                        # what predict() must see after its first pass (or the residuals themselves)
    clip = lambda v: sext(v, bps)
    stage[0] = out[0]
    for i in range(1, min(1 + order, n)):
        stage[i] = clip(out[i] - out[i - 1])
    for i in range(1 + order, n):
        past0 = out[i - order - 1]
        s = 0
        for j in range(order):
            s = w32(s + w32(co[order - 1 - j] * w32(out[i - order + j] - past0)))
        val = w32(s + ((1 << shift) >> 1)) >> shift
        res = clip(out[i] - past0 - val)
        stage[i] = res
        assert clip(w32(w32(res + past0) + val)) == out[i]
        if res > 0:
            for j in range(order):
                v = w32(past0 - out[i - order + j])
                sg = (v > 0) - (v < 0)
                co[order - 1 - j] -= sg
                res -= (1 + j) * (w32(sg * v) >> shift)
                if res <= 0:
                    break
        elif res < 0:
            for j in range(order):
                v = w32(past0 - out[i - order + j])
                sg = (v > 0) - (v < 0)
                co[order - 1 - j] += sg
                res -= (1 + j) * (w32(-sg * v) >> shift)
                if res >= 0:
                    break
    if order == 31 or mode == 15:          # undo the decoder's first pass: stage[i] = clip(r[i] + stage[i-1])
        r = [stage[0]] + [clip(stage[i] - stage[i - 1]) for i in range(1, n)]
        return np.array(r, dtype=np.int64).astype(np.int32)
    return np.array(stage, dtype=np.int64).astype(np.int32)


def smooth_pcm(rng, n, bps):
    t = np.arange(n)
    amp = (1 << (bps - 2))
    x = amp * 0.6 * np.sin(t * rng.uniform(0.01, 0.2)) + rng.standard_normal(n) * amp * 0.01
    return np.clip(np.round(x), -(1 << (bps - 1)), (1 << (bps - 1)) - 1).astype(np.int64)


@pytest.mark.parametrize("mode,order,shift,bps", [(0, 4, 9, 16), (0, 8, 9, 16), (15, 4, 9, 16), (0, 31, 9, 24), (0, 1, 4, 16),
                                                   (0, 2, 0, 20), (0, 12, 9, 32), (0, 0, 9, 16)])
def test_oracle_decodes_what_an_encoder_produced(mode, order, shift, bps):
    rng = np.random.default_rng(order * 31 + bps + mode)
    n = 300
    pcm = smooth_pcm(rng, n, min(bps, 24))
    coeffs = np.zeros(32, np.int32)
    coeffs[:order] = rng.integers(-300, 300, order)
    if order:
        coeffs[0] = 1 << shift if shift < 15 else 1 << 14
    res = alac_encode(pcm, mode, order, shift, bps, coeffs) if order else pcm.astype(np.int32)
    got = oracle.alac_predict(res[None], oracle.alac_desc([mode], [order], [shift], [bps]), coeffs[None])[0]
    assert np.array_equal(got, pcm.astype(np.int32))


def test_oracle_rejects_invalid_modes_like_the_reference():
    buf = np.arange(20, dtype=np.int32)[None]
    out = oracle.alac_predict(buf, oracle.alac_desc([7], [4], [9], [16]), np.ones((1, 32), np.int32))
    assert np.array_equal(out, buf)  # lib.rs:167-169: decode_error, nothing written


def alac_case(seed, nb, blocksize):
    rng = np.random.default_rng(seed)
    bps = rng.choice([16, 20, 24, 32], nb).astype(np.uint8)
    buf = np.stack([rng.integers(-(1 << 10), 1 << 10, blocksize) for _ in range(nb)]).astype(np.int32)
    buf[0] = rng.integers(-(1 << 31), 1 << 31, blocksize)      # wrapping arithmetic everywhere
    mode = rng.choice([0, 0, 0, 15, 7], nb).astype(np.uint8)   # 7 = invalid: block must stay untouched
    order = rng.choice([0, 1, 2, 3, 4, 8, 12, 16, 31], nb).astype(np.uint8)
    order[:9] = [0, 1, 2, 3, 4, 8, 12, 16, 31][:min(9, nb)] if nb >= 9 else order[:9]
    shift = rng.integers(0, 16, nb).astype(np.uint8)
    coeffs = rng.integers(-(1 << 15), 1 << 15, (nb, 32)).astype(np.int32)
    return buf, mode, order, shift, bps, coeffs


@pytest.mark.parametrize("blocksize", [1, 5, 31, 32, 100, 352])
def test_emu_alac_predict(emu_ctx, blocksize):
    from symphonia_amd import AlacPredictor, alac_desc
    buf, mode, order, shift, bps, coeffs = alac_case(blocksize, 70, blocksize)
    got = AlacPredictor(emu_ctx).predict(buf, alac_desc(mode, order, shift, bps), coeffs)
    want = oracle.alac_predict(buf, oracle.alac_desc(mode, order, shift, bps), coeffs)
    assert np.array_equal(got, want)


def test_emu_alac_uniform_small_orders(emu_ctx):
    """Whole wavefronts of order <= 4 / <= 8 take the short-tap specialisations."""
    from symphonia_amd import AlacPredictor, alac_desc
    rng = np.random.default_rng(3)
    for hi in (4, 6, 8, 16):  # (6: the six-tap steady tiles between eight-tap first / last ones)
        nb, bs = 64, 96
        buf = rng.integers(-(1 << 14), 1 << 14, (nb, bs)).astype(np.int32)
        order = rng.integers(1, hi + 1, nb).astype(np.uint8)
        d = (np.zeros(nb, np.uint8), order, np.full(nb, 9, np.uint8), np.full(nb, 16, np.uint8))
        coeffs = rng.integers(-2000, 2000, (nb, 32)).astype(np.int32)
        got = AlacPredictor(emu_ctx).predict(buf, alac_desc(*d), coeffs)
        assert np.array_equal(got, oracle.alac_predict(buf, oracle.alac_desc(*d), coeffs)), hi


@pytest.mark.parametrize("bps", [16, 24])
def test_emu_alac_uniform_order_8(emu_ctx, bps):
    """Every block of the wavefront at order 8 (what Apple's encoder writes): the instantiation without per-tap order
    masks, with 24-bit multiplies (16-bit channels) and with full ones (24-bit channels)."""
    from symphonia_amd import AlacPredictor, alac_desc
    rng = np.random.default_rng(80 + bps)
    nb, bs = 128, 200
    buf = rng.integers(-(1 << (bps - 3)), 1 << (bps - 3), (nb, bs)).astype(np.int32)
    buf[::7] = rng.integers(-(1 << 31), 1 << 31, (len(buf[::7]), bs))
    buf[:, 0] = rng.integers(-(1 << 15), 1 << 15, nb)
    d = (rng.choice([0, 15], nb).astype(np.uint8), np.full(nb, 8, np.uint8), rng.integers(0, 12, nb).astype(np.uint8), np.full(nb, bps, np.uint8))
    coeffs = rng.integers(-(1 << 15), 1 << 15, (nb, 32)).astype(np.int32)
    got = AlacPredictor(emu_ctx).predict(buf, alac_desc(*d), coeffs)
    assert np.array_equal(got, oracle.alac_predict(buf, oracle.alac_desc(*d), coeffs))


def narrow_case(seed, blocksize):
    """Five wavefronts of 64 blocks around the bound that selects the 24-bit-multiply instantiation (alac.hip, "narrow"):
    wavefronts 0 and 4 sit exactly ON the bound (23-bit channels, first samples -2^22 and 2^22 - 1, coefficients
    +-(2^22 - 1), residuals that drive the outputs to the rails), wavefronts 1, 2, 3 each have ONE block just outside it
    (first sample 2^22; a coefficient of 2^22; a 24-bit channel) and must take the full multiply."""
    rng = np.random.default_rng(seed)
    nb = 5 * 64
    bps = np.full(nb, 23, np.uint8)
    bps[64:256] = rng.choice([16, 20, 23], 192)
    buf = rng.integers(-(1 << 21), 1 << 21, (nb, blocksize)).astype(np.int32)
    buf[::3] = rng.integers(-(1 << 31), 1 << 31, (len(buf[::3]), blocksize))   # wrapping residuals
    buf[:, 0] = rng.choice([-(1 << 22), (1 << 22) - 1, 0, 12345], nb)
    mode = rng.choice([0, 0, 15], nb).astype(np.uint8)
    order = rng.choice([1, 2, 3, 4, 8, 16, 31], nb).astype(np.uint8)
    shift = rng.integers(0, 12, nb).astype(np.uint8)
    coeffs = rng.integers(-(1 << 15), 1 << 15, (nb, 32)).astype(np.int32)
    coeffs[::5] = rng.choice([-(1 << 22) + 1, (1 << 22) - 1], (len(coeffs[::5]), 32))
    buf[64 + 7, 0] = 1 << 22
    order[128 + 9] = 8
    coeffs[128 + 9, 3] = 1 << 22
    bps[192 + 11] = 24
    return buf, mode, order, shift, bps, coeffs


@pytest.mark.parametrize("blocksize", [40, 352])
def test_emu_alac_24_bit_multiply_bound(emu_ctx, blocksize):
    from symphonia_amd import AlacPredictor, alac_desc
    buf, mode, order, shift, bps, coeffs = narrow_case(blocksize, blocksize)
    want = oracle.alac_predict(buf, oracle.alac_desc(mode, order, shift, bps), coeffs)
    got = AlacPredictor(emu_ctx).predict(buf, alac_desc(mode, order, shift, bps), coeffs)
    assert np.array_equal(got, want)
    rng = np.random.default_rng(1)
    weight, msh = rng.integers(-3, 4, 160).astype(np.int32), rng.integers(0, 32, 160).astype(np.uint8)
    got = buf.copy()
    AlacPredictor(emu_ctx).predict_stereo(got, alac_desc(mode, order, shift, bps), coeffs, weight, msh)
    for p in range(160):
        if weight[p]:
            want[2 * p], want[2 * p + 1] = oracle.alac_decorrelate_mid_side(want[2 * p], want[2 * p + 1], int(weight[p]), int(msh[p]))
    assert np.array_equal(got, want)


def narrow_update_case(seed, nb, bs, hi_order):
    """The narrow form's sign-LMS update at its edges (alac.hip, alac_step: the residual carried as -|res|): every shift 0 .. 31 (the
    ceiling of the negative case adds 2^shift - 1), residuals of 0, +-1, i32::MIN / MAX, small ones that cross zero after a tap or two,
    outputs on both ends of the channel's range so that |val| reaches 2^bps - 1."""
    rng = np.random.default_rng(seed)
    bps = rng.integers(16, 24, nb).astype(np.uint8)
    buf = rng.integers(-40, 41, (nb, bs)).astype(np.int32)
    kind = rng.integers(0, 6, (nb, bs))
    buf[kind == 0] = 0
    buf[kind == 1] = rng.choice([-(1 << 31), (1 << 31) - 1, -(1 << 31) + 1, 1, -1], int((kind == 1).sum()))
    big = rng.integers(-(1 << 22), 1 << 22, (nb, bs))
    buf[kind == 2] = big[kind == 2]
    buf[:, 0] = rng.integers(-(1 << 22), 1 << 22, nb)
    order = rng.integers(1, hi_order + 1, nb).astype(np.uint8)
    shift = (np.arange(nb) % 32).astype(np.uint8)
    mode = rng.choice([0, 15], nb).astype(np.uint8)
    coeffs = rng.integers(-(1 << 12), 1 << 12, (nb, 32)).astype(np.int32)
    coeffs[::3] = rng.integers(-(1 << 22) + 1, 1 << 22, (len(coeffs[::3]), 32))
    return buf, mode, order, shift, bps, coeffs


@pytest.mark.parametrize("hi_order,uniform", [(8, 1), (8, 2), (8, 0), (6, 0), (6, 2), (4, 0), (31, 0)])
def test_emu_alac_narrow_update_edges(emu_ctx, hi_order, uniform):
    from symphonia_amd import AlacPredictor, alac_desc
    buf, mode, order, shift, bps, coeffs = narrow_update_case(40 + hi_order, 192, 150, hi_order)
    if uniform:
        order[:] = hi_order
    if uniform == 2:  # no block runs the double predictor: the instantiation without that pass
        mode[:] = 0
    want = oracle.alac_predict(buf, oracle.alac_desc(mode, order, shift, bps), coeffs)
    got = AlacPredictor(emu_ctx).predict(buf, alac_desc(mode, order, shift, bps), coeffs)
    assert np.array_equal(got, want)


def wide_update_case(seed, nb, bs):
    """The wide form's carried-residual update (alac_step, QF): wavefronts of order-8 blocks without the double predictor whose channels have
    24 .. 26 bits ("mid": the instantiation under test), next to wavefronts just outside the bound (27 .. 32 bits, or a first sample beyond
    +-2^25), which must take the general form.  Every shift, residuals at the i32 limits, outputs at both ends of the range."""
    rng = np.random.default_rng(seed)
    group = np.arange(nb) // 64
    bps = np.where(group % 3 == 2, rng.integers(27, 33, nb), rng.integers(24, 27, nb)).astype(np.uint8)
    buf = rng.integers(-40, 41, (nb, bs)).astype(np.int32)
    kind = rng.integers(0, 6, (nb, bs))
    buf[kind == 0] = 0
    buf[kind == 1] = rng.choice([-(1 << 31), (1 << 31) - 1, -(1 << 31) + 1, 1, -1], int((kind == 1).sum()))
    big = rng.integers(-(1 << 25), 1 << 25, (nb, bs))
    buf[kind == 2] = big[kind == 2]
    buf[:, 0] = rng.integers(-(1 << 25), 1 << 25, nb)
    buf[group % 3 == 1, 0] = rng.choice([-(1 << 25), (1 << 25) - 1, -(1 << 25) - 1, 1 << 25, -(1 << 31), (1 << 31) - 1], int((group % 3 == 1).sum()))
    order = np.full(nb, 8, np.uint8)
    shift = (np.arange(nb) % 32).astype(np.uint8)
    mode = np.zeros(nb, np.uint8)
    coeffs = rng.integers(-(1 << 15), 1 << 15, (nb, 32)).astype(np.int32)
    coeffs[::3] = rng.integers(-(1 << 31), (1 << 31) - 1, (len(coeffs[::3]), 32))
    return buf, mode, order, shift, bps, coeffs


def test_emu_alac_wide_update_edges(emu_ctx):
    from symphonia_amd import AlacPredictor, alac_desc
    buf, mode, order, shift, bps, coeffs = wide_update_case(77, 384, 170)
    want = oracle.alac_predict(buf, oracle.alac_desc(mode, order, shift, bps), coeffs)
    got = AlacPredictor(emu_ctx).predict(buf, alac_desc(mode, order, shift, bps), coeffs)
    assert np.array_equal(got, want)


@pytest.mark.parametrize("blocksize,nb", [(64, 64), (100, 70), (31, 130)])
def test_emu_alac_predict_stereo_fused(emu_ctx, blocksize, nb):
    """predict with decorrelate_mid_side fused into the write-back == predict, then decorrelate_mid_side."""
    from symphonia_amd import AlacPredictor, alac_desc
    buf, mode, order, shift, bps, coeffs = alac_case(7 * blocksize, nb, blocksize)
    rng = np.random.default_rng(blocksize)
    weight = rng.integers(-3, 4, nb // 2).astype(np.int32)
    msh = rng.integers(0, 32, nb // 2).astype(np.uint8)
    got = buf.copy()
    AlacPredictor(emu_ctx).predict_stereo(got, alac_desc(mode, order, shift, bps), coeffs, weight, msh)
    want = oracle.alac_predict(buf, oracle.alac_desc(mode, order, shift, bps), coeffs)
    for p in range(nb // 2):
        if weight[p]:
            want[2 * p], want[2 * p + 1] = oracle.alac_decorrelate_mid_side(want[2 * p], want[2 * p + 1], int(weight[p]), int(msh[p]))
    assert np.array_equal(got, want)


def test_emu_alac_mid_side(emu_ctx):
    from symphonia_amd import AlacPredictor
    rng = np.random.default_rng(4)
    a = rng.integers(-(1 << 31), 1 << 31, (5, 77)).astype(np.int32)
    b = rng.integers(-(1 << 31), 1 << 31, (5, 77)).astype(np.int32)
    weight = np.array([0, 1, 2, 3, -5], np.int32)
    shift = np.array([0, 1, 2, 31, 4], np.uint8)
    ga, gb = AlacPredictor(emu_ctx).mid_side(weight, shift, a, b)
    for p in range(5):
        wa, wb = (a[p], b[p]) if weight[p] == 0 else oracle.alac_decorrelate_mid_side(a[p], b[p], int(weight[p]), int(shift[p]))
        assert np.array_equal(ga[p], wa) and np.array_equal(gb[p], wb)


@pytest.mark.gpu
@pytest.mark.parametrize("blocksize", [1, 33, 352, 4096])
def test_gpu_alac_predict(blocksize):
    import torch
    from symphonia_amd import AlacPredictor, Context, alac_desc
    if not torch.cuda.is_available():
        pytest.fail("no GPU visible")
    buf, mode, order, shift, bps, coeffs = alac_case(100 + blocksize, 200, blocksize)
    with Context(0) as ctx:
        ctx.use_torch_stream()
        d = torch.from_numpy(buf.copy()).cuda()
        desc = torch.from_numpy(alac_desc(mode, order, shift, bps).view(np.uint8).reshape(-1, 4)).cuda()
        AlacPredictor(ctx).predict(d, desc, torch.from_numpy(coeffs).cuda())
        torch.cuda.synchronize()
        got = d.cpu().numpy()
        d2 = torch.from_numpy(buf.copy()).cuda()
        AlacPredictor(ctx).predict_stereo(d2, desc, torch.from_numpy(coeffs).cuda(), torch.arange(-50, 50, dtype=torch.int32).cuda(),
                                          (torch.arange(100) % 32).to(torch.uint8).cuda())
        torch.cuda.synchronize()
        got_fused = d2.cpu().numpy()
        a = torch.from_numpy(buf[:100].copy()).cuda()
        b = torch.from_numpy(buf[100:].copy()).cuda()
        w = torch.arange(-50, 50, dtype=torch.int32).cuda()
        s = (torch.arange(100) % 32).to(torch.uint8).cuda()
        AlacPredictor(ctx).mid_side(w, s, a, b)
        torch.cuda.synchronize()
        ga, gb = a.cpu().numpy(), b.cpu().numpy()
    want = oracle.alac_predict(buf, oracle.alac_desc(mode, order, shift, bps), coeffs)
    assert np.array_equal(got, want)
    wq = np.arange(-50, 50, dtype=np.int32)
    sq = (np.arange(100) % 32).astype(np.uint8)
    for p in range(100):
        if wq[p]:
            want[2 * p], want[2 * p + 1] = oracle.alac_decorrelate_mid_side(want[2 * p], want[2 * p + 1], int(wq[p]), int(sq[p]))
    assert np.array_equal(got_fused, want)
    for p in range(100):
        wgt, sh = p - 50, p % 32
        wa, wb = (buf[p], buf[100 + p]) if wgt == 0 else oracle.alac_decorrelate_mid_side(buf[p], buf[100 + p], wgt, sh)
        assert np.array_equal(ga[p], wa) and np.array_equal(gb[p], wb), p


@pytest.mark.gpu
@pytest.mark.parametrize("bps", [16, 24])
def test_gpu_alac_uniform_order_8(bps):
    import torch
    from symphonia_amd import AlacPredictor, Context, alac_desc
    if not torch.cuda.is_available():
        pytest.fail("no GPU visible")
    rng = np.random.default_rng(180 + bps)
    nb, bs = 640, 4096
    buf = rng.integers(-(1 << (bps - 3)), 1 << (bps - 3), (nb, bs)).astype(np.int32)
    buf[::7] = rng.integers(-(1 << 31), 1 << 31, (len(buf[::7]), bs))
    buf[:, 0] = rng.integers(-(1 << 15), 1 << 15, nb)
    d = (rng.choice([0, 15], nb).astype(np.uint8), np.full(nb, 8, np.uint8), rng.integers(0, 12, nb).astype(np.uint8), np.full(nb, bps, np.uint8))
    coeffs = rng.integers(-(1 << 15), 1 << 15, (nb, 32)).astype(np.int32)
    want = oracle.alac_predict(buf, oracle.alac_desc(*d), coeffs)
    with Context(0) as ctx:
        ctx.use_torch_stream()
        t = torch.from_numpy(buf.copy()).cuda()
        AlacPredictor(ctx).predict(t, torch.from_numpy(alac_desc(*d).view(np.uint8).reshape(-1, 4)).cuda(), torch.from_numpy(coeffs).cuda())
        torch.cuda.synchronize()
        assert np.array_equal(t.cpu().numpy(), want)


@pytest.mark.gpu
@pytest.mark.parametrize("blocksize", [40, 4096])
def test_gpu_alac_24_bit_multiply_bound(blocksize):
    """On the bound and just outside it (narrow_case), long blocks included: the coefficient drift of 4096 updates."""
    import torch
    from symphonia_amd import AlacPredictor, Context, alac_desc
    if not torch.cuda.is_available():
        pytest.fail("no GPU visible")
    buf, mode, order, shift, bps, coeffs = narrow_case(1000 + blocksize, blocksize)
    want = oracle.alac_predict(buf, oracle.alac_desc(mode, order, shift, bps), coeffs)
    with Context(0) as ctx:
        ctx.use_torch_stream()
        d = torch.from_numpy(buf.copy()).cuda()
        desc = torch.from_numpy(alac_desc(mode, order, shift, bps).view(np.uint8).reshape(-1, 4)).cuda()
        AlacPredictor(ctx).predict(d, desc, torch.from_numpy(coeffs).cuda())
        torch.cuda.synchronize()
        assert np.array_equal(d.cpu().numpy(), want)


@pytest.mark.gpu
@pytest.mark.parametrize("hi_order,uniform", [(8, 1), (8, 2), (8, 0), (6, 0), (6, 2), (4, 0), (31, 0)])
def test_gpu_alac_narrow_update_edges(hi_order, uniform):
    import torch
    from symphonia_amd import AlacPredictor, Context, alac_desc
    if not torch.cuda.is_available():
        pytest.fail("no GPU visible")
    buf, mode, order, shift, bps, coeffs = narrow_update_case(140 + hi_order, 1024, 1000, hi_order)
    if uniform:
        order[:] = hi_order
    if uniform == 2:
        mode[:] = 0
    want = oracle.alac_predict(buf, oracle.alac_desc(mode, order, shift, bps), coeffs)
    with Context(0) as ctx:
        ctx.use_torch_stream()
        d = torch.from_numpy(buf.copy()).cuda()
        desc = torch.from_numpy(alac_desc(mode, order, shift, bps).view(np.uint8).reshape(-1, 4)).cuda()
        AlacPredictor(ctx).predict(d, desc, torch.from_numpy(coeffs).cuda())
        torch.cuda.synchronize()
        assert np.array_equal(d.cpu().numpy(), want)


@pytest.mark.gpu
def test_gpu_alac_wide_update_edges():
    import torch
    from symphonia_amd import AlacPredictor, Context, alac_desc
    if not torch.cuda.is_available():
        pytest.fail("no GPU visible")
    buf, mode, order, shift, bps, coeffs = wide_update_case(177, 1536, 1000)
    want = oracle.alac_predict(buf, oracle.alac_desc(mode, order, shift, bps), coeffs)
    with Context(0) as ctx:
        ctx.use_torch_stream()
        d = torch.from_numpy(buf.copy()).cuda()
        desc = torch.from_numpy(alac_desc(mode, order, shift, bps).view(np.uint8).reshape(-1, 4)).cuda()
        AlacPredictor(ctx).predict(d, desc, torch.from_numpy(coeffs).cuda())
        torch.cuda.synchronize()
        assert np.array_equal(d.cpu().numpy(), want)
