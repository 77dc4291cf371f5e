"""AAC spectral tools in front of Dsp::synth: joint-stereo decoding (aac/cpe.rs:110-157) and the
filtering loops of Tns::synth (aac/ics/tns.rs:180-195).

The reference has no test for either, so the oracle's restatement is "parity unpinned by the reference"; it is pinned here
by the defining arithmetic (sum / difference / one scalar multiply in f32) and, for TNS, by scipy's all-pole filter in f64
plus the start-up structure of the reference's loops.  The kernels are then compared bit for bit with the oracle: in
CPU emulation here, on the MI355X under -m gpu."""
import numpy as np
import pytest
from scipy.signal import lfilter

import oracle
from emu_lib import emu_ctx  # noqa: F401
from helpers import bit_equal, equal_mod_nan

# swb offsets of the 44.1 / 48 kHz tables (ISO/IEC 14496-3 Tables 4.138, 4.139): what ICS get_bands() returns there
SWB_LONG = [0, 4, 8, 12, 16, 20, 24, 28, 32, 36, 40, 48, 56, 64, 72, 80, 88, 96, 108, 120, 132, 144, 160, 176, 196, 216,
            240, 264, 292, 320, 352, 384, 416, 448, 480, 512, 544, 576, 608, 640, 672, 704, 736, 768, 800, 832, 864, 896,
            928, 1024]
SWB_SHORT = [0, 4, 8, 12, 16, 20, 28, 36, 44, 56, 68, 80, 96, 112, 128]


def tns_lpc(rng, order, coef_res=True, limit=None):
    """TnsCoeffs::read's arithmetic (tns.rs:70-103) on random transmitted values: realistic, stable filters."""
    bits, fac = (4, 8.0) if coef_res else (3, 4.0)
    iqfac, iqfac_m = np.float32((fac - 0.5) / (np.pi / 2)), np.float32((fac + 0.5) / (np.pi / 2))
    lo, hi = -(1 << (bits - 1)), 1 << (bits - 1)
    if limit:
        lo, hi = -limit, limit + 1
    c = rng.integers(lo, hi, order).astype(np.float32)
    tmp = np.sin(np.where(c >= 0, c / iqfac, c / iqfac_m).astype(np.float32)).astype(np.float32)
    coef, b = np.zeros(21, np.float32), np.zeros(21, np.float32)
    for m in range(1, order + 1):
        for i in range(1, m):
            b[i] = coef[i - 1] + tmp[m - 1] * coef[m - i - 1]
        coef[:m - 1] = b[1:m]
        coef[m - 1] = tmp[m - 1]
    return coef[:20].copy()


def js_frame(rng, short):
    d = np.zeros((), oracle_dtype_js())
    nb = len(SWB_SHORT if short else SWB_LONG) - 1
    d["num_windows"] = 8 if short else 1
    d["max_sfb"] = rng.integers(0, nb + 1)
    d["mode"] = rng.choice([0, 0, 1, 1, 2], 128)
    d["scale"] = (rng.standard_normal(128) * 2).astype(np.float32)
    return d


def oracle_dtype_js():
    from symphonia_amd import AAC_JS_DTYPE
    return AAC_JS_DTYPE


def js_reference(coeffs, pairs, desc):
    out = coeffs.copy()
    for p, (cl, cr) in enumerate(pairs):
        for f in range(coeffs.shape[1]):
            d = desc[p, f]
            bands = SWB_LONG if d["num_windows"] == 1 else SWB_SHORT
            out[cl, f], out[cr, f] = oracle.aac_joint_stereo(coeffs[cl, f], coeffs[cr, f], d["num_windows"], d["max_sfb"], bands,
                                                             d["mode"], d["scale"])
    return out


def tns_case(rng, n_frames, n_filters):
    from symphonia_amd import AAC_TNS_DTYPE
    coeffs = (rng.standard_normal((n_frames, 1024)) * np.exp2(rng.integers(-6, 8, (n_frames, 1)))).astype(np.float32)
    filt = np.zeros(n_filters, AAC_TNS_DTYPE)
    used = {}  # frame -> list of taken ranges (filters of one frame are disjoint, tns.rs:163-166)
    for k in range(n_filters):
        f = int(rng.integers(0, n_frames))
        if rng.random() < 0.3:  # a short window's filter
            w = int(rng.integers(0, 8))
            lo, hi = sorted(rng.choice(SWB_SHORT, 2, replace=False))
            lo, hi, order = w * 128 + lo, w * 128 + hi, int(rng.integers(1, 8))
        else:
            lo, hi = sorted(rng.choice(SWB_LONG, 2, replace=False))
            order = int(rng.integers(1, 21))
        if any(lo < b and a < hi for a, b in used.get(f, [])):
            hi = lo  # overlaps an earlier filter of the same frame: make it empty (skipped)
        used.setdefault(f, []).append((lo, hi))
        filt[k] = (f, lo, hi, order, int(rng.integers(0, 2)), 0, tns_lpc(rng, order, coef_res=bool(rng.integers(0, 2))))
    return coeffs, filt


def tns_reference(coeffs, filt, n_frames):
    out = coeffs.copy()
    for f in filt:
        if f["frame"] < n_frames and f["start"] < f["end"] <= 1024 and 1 <= f["order"] <= 20:
            out[f["frame"]] = oracle.aac_tns_filter(out[f["frame"]], f["start"], f["end"], f["order"], f["direction"], f["lpc"])
    return out


# ---------------------------------------------------------------- oracle pins

def test_oracle_joint_stereo_arithmetic():
    rng = np.random.default_rng(1)
    l, r = rng.standard_normal(1024).astype(np.float32), rng.standard_normal(1024).astype(np.float32)
    mode, scale = np.zeros(128, np.uint8), np.zeros(128, np.float32)
    mode[3], mode[5], scale[5], mode[7] = 1, 2, -1.75, 1
    ol, orr = oracle.aac_joint_stereo(l, r, 1, 7, SWB_LONG, mode, scale)  # max_sfb 7: band 7 is out of reach
    wl, wr = l.copy(), r.copy()
    wl[12:16], wr[12:16] = l[12:16] + r[12:16], l[12:16] - r[12:16]
    wr[20:24] = np.float32(-1.75) * l[20:24]
    assert bit_equal(ol, wl) and bit_equal(orr, wr)
    # eight short windows: slot = w * 16 + sfb, lines w * 128 + band
    mode[:] = 0
    mode[2 * 16 + 6] = 1
    ol, orr = oracle.aac_joint_stereo(l, r, 8, 14, SWB_SHORT, mode, scale)
    a, b = 2 * 128 + 28, 2 * 128 + 36
    wl, wr = l.copy(), r.copy()
    wl[a:b], wr[a:b] = l[a:b] + r[a:b], l[a:b] - r[a:b]
    assert bit_equal(ol, wl) and bit_equal(orr, wr)


@pytest.mark.parametrize("order,direction", [(1, 0), (7, 1), (12, 0), (12, 1), (20, 0)])
def test_oracle_tns_is_the_all_pole_filter(order, direction):
    rng = np.random.default_rng(order * 2 + direction)
    x = rng.standard_normal(1024).astype(np.float32)
    lpc = tns_lpc(rng, order, limit=3)  # mild resonances: the f32 recurrence stays close to the f64 one
    y = oracle.aac_tns_filter(x, 96, 896, order, direction, lpc)
    seg = x[96:896].astype(np.float64)
    a = np.concatenate([[1.0], lpc[:order].astype(np.float64)])
    f = lfilter([1.0], a, seg[::-1] if direction else seg)
    ref = x.astype(np.float64)
    ref[96:896] = f[::-1] if direction else f
    assert np.abs(y - ref).max() <= 2e-5 * np.abs(ref).max()
    assert bit_equal(y[:96], x[:96]) and bit_equal(y[896:], x[896:])


def test_oracle_tns_start_up():
    # tns.rs:184, 191: line m of the range only sees min(order, m) earlier lines of the range, nothing outside it
    x = np.arange(1, 1025, dtype=np.float32)
    lpc = np.zeros(20, np.float32)
    lpc[:3] = (0.5, 0.25, 0.125)
    y = oracle.aac_tns_filter(x, 8, 16, 3, 0, lpc)
    f = np.float32
    assert y[8] == x[8]
    assert y[9] == x[9] - y[8] * f(0.5)
    assert y[10] == f(x[10] - y[9] * f(0.5)) - y[8] * f(0.25)
    assert y[11] == f(f(x[11] - y[10] * f(0.5)) - y[9] * f(0.25)) - y[8] * f(0.125)
    y = oracle.aac_tns_filter(x, 8, 16, 3, 1, lpc)
    assert y[15] == x[15] and y[14] == x[14] - y[15] * f(0.5)
    # an untouched -0.0 stays -0.0 (no "minus zero times coefficient" is ever subtracted)
    z = np.zeros(1024, np.float32)
    z[8] = -0.0
    assert np.signbit(oracle.aac_tns_filter(z, 8, 16, 3, 0, -lpc)[8])


# ---------------------------------------------------------------- kernels vs oracle (CPU emulation)

@pytest.mark.parametrize("seed,frames", [(0, 5), (1, 1), (2, 9)])
def test_emu_joint_stereo(emu_ctx, seed, frames):
    from symphonia_amd import AacSpectralTools
    rng = np.random.default_rng(seed)
    coeffs = rng.standard_normal((6, frames, 1024)).astype(np.float32)
    pairs = np.array([[4, 1], [0, 5]], np.int32)  # chains 2 and 3 are mono / untouched
    desc = np.array([[js_frame(rng, short=bool(rng.integers(0, 2))) for _ in range(frames)] for _ in pairs])
    want = js_reference(coeffs, pairs, desc)
    got = coeffs.copy()
    AacSpectralTools(emu_ctx, SWB_LONG, SWB_SHORT).joint_stereo(got, pairs, desc)
    assert bit_equal(got, want)
    assert bit_equal(got[2:4], coeffs[2:4])


@pytest.mark.parametrize("seed,n_frames,n_filters", [(0, 9, 70), (1, 3, 5), (2, 40, 130)])
def test_emu_tns(emu_ctx, seed, n_frames, n_filters):
    from symphonia_amd import AacSpectralTools
    rng = np.random.default_rng(seed)
    coeffs, filt = tns_case(rng, n_frames, n_filters)
    filt[0]["frame"] = n_frames + 3   # out of range: skipped
    filt[1]["order"] = 0              # skipped
    want = tns_reference(coeffs, filt, n_frames)
    got = coeffs.copy()
    AacSpectralTools(emu_ctx, SWB_LONG, SWB_SHORT).tns(got, filt)
    assert bit_equal(got, want)


def test_emu_tns_unaligned_ranges(emu_ctx):
    """Ranges that are not multiples of four lines (no swb table produces them, the ABI still honours them)."""
    from symphonia_amd import AAC_TNS_DTYPE, AacSpectralTools
    rng = np.random.default_rng(11)
    coeffs = rng.standard_normal((6, 1024)).astype(np.float32)
    filt = np.zeros(6, AAC_TNS_DTYPE)
    for k, (lo, hi, order, direction) in enumerate([(1, 1023, 20, 0), (97, 331, 7, 1), (2, 5, 12, 0), (1021, 1024, 3, 1),
                                                    (510, 511, 1, 0), (0, 1024, 12, 1)]):
        filt[k] = (k, lo, hi, order, direction, 0, tns_lpc(rng, order))
    want = tns_reference(coeffs, filt, 6)
    got = coeffs.copy()
    AacSpectralTools(emu_ctx, SWB_LONG, SWB_SHORT).tns(got, filt)
    assert bit_equal(got, want)


def tns_case_ragged(rng, n_frames, orders=(1, 20)):
    """One filter per frame over an arbitrary range (any start, any length, both directions, every order): quads of
    lanes then mix ranges whose groups are aligned, unaligned, ragged or already finished.  `orders`: the range the orders are
    drawn from -- a block of 128 filters runs in the kernel of its HIGHEST order (two filters per lane up to order 12; with
    (12, 12) the form without masks)."""
    from symphonia_amd import AAC_TNS_DTYPE
    coeffs = (rng.standard_normal((n_frames, 1024)) * np.exp2(rng.integers(-6, 8, (n_frames, 1)))).astype(np.float32)
    filt = np.zeros(n_frames, AAC_TNS_DTYPE)
    for k in range(n_frames):
        kind = rng.integers(0, 4)
        if kind == 0:    # aligned start, any length
            lo = 4 * int(rng.integers(0, 255))
            hi = int(rng.integers(lo + 1, 1025))
        elif kind == 1:  # anything
            lo = int(rng.integers(0, 1023))
            hi = int(rng.integers(lo + 1, 1025))
        elif kind == 2:  # short
            lo = int(rng.integers(0, 1000))
            hi = lo + int(rng.integers(1, 24))
        else:            # whole multiples of sixteen lines
            lo = 16 * int(rng.integers(0, 32))
            hi = lo + 16 * int(rng.integers(1, (1024 - lo) // 16 + 1))
        order = int(rng.integers(orders[0], orders[1] + 1))
        filt[k] = (k, lo, hi, order, int(rng.integers(0, 2)), 0, tns_lpc(rng, order, coef_res=bool(rng.integers(0, 2))))
    return coeffs, filt[rng.permutation(n_frames)]


TNS_ORDER_CLASSES = [(1, 4), (4, 4), (3, 8), (8, 8), (5, 12), (12, 12), (9, 16)]


@pytest.mark.parametrize("direct", ["0", "1"])
@pytest.mark.parametrize("seed,n_frames", [(3, 64), (4, 130), (5, 7), (6, 300)])
def test_emu_tns_ragged_quads(emu_ctx, seed, n_frames, direct, monkeypatch):
    """both ways a group's lines move (SYMACCEL_TNS_DIRECT: the quad exchange of a large pass, lane by lane in a small one)"""
    from symphonia_amd import AacSpectralTools
    monkeypatch.setenv("SYMACCEL_TNS_DIRECT", direct)
    coeffs, filt = tns_case_ragged(np.random.default_rng(seed), n_frames)
    want = tns_reference(coeffs, filt, n_frames)
    got = coeffs.copy()
    AacSpectralTools(emu_ctx, SWB_LONG, SWB_SHORT).tns(got, filt)
    assert bit_equal(got, want)


@pytest.mark.parametrize("direct", ["0", "1"])
@pytest.mark.parametrize("orders", TNS_ORDER_CLASSES)
def test_emu_tns_order_classes(emu_ctx, orders, direct, monkeypatch):
    """every tap class of the pass, with and without orders below the class's (the masked and the unmasked steady state), some
    filters skipped (order 0, a frame outside the batch)"""
    from symphonia_amd import AacSpectralTools
    monkeypatch.setenv("SYMACCEL_TNS_DIRECT", direct)
    n_frames = 200
    coeffs, filt = tns_case_ragged(np.random.default_rng(100 + orders[0] + 20 * orders[1]), n_frames, orders)
    filt[5]["order"] = 0
    filt[77]["frame"] = n_frames + 1
    filt[150]["start"], filt[150]["end"] = 8, 8
    want = tns_reference(coeffs, filt, n_frames)
    got = coeffs.copy()
    AacSpectralTools(emu_ctx, SWB_LONG, SWB_SHORT).tns(got, filt)
    assert bit_equal(got, want)


def test_emu_tools_feed_synth(emu_ctx):
    """joint stereo -> TNS -> Dsp::synth: the order of ChannelPair::decode + Ics::synth_channel (cpe.rs:109-157, ics/mod.rs:449-468)."""
    from symphonia_amd import AacDsp, AacSpectralTools
    rng = np.random.default_rng(5)
    frames = 4
    coeffs = rng.standard_normal((2, frames, 1024)).astype(np.float32)
    pairs = np.array([[0, 1]], np.int32)
    desc = np.array([[js_frame(rng, short=False) for _ in range(frames)]])
    _, filt = tns_case(rng, 2 * frames, 6)
    want = tns_reference(js_reference(coeffs, pairs, desc).reshape(-1, 1024), filt, 2 * frames).reshape(2, frames, 1024)
    tools = AacSpectralTools(emu_ctx, SWB_LONG, SWB_SHORT)
    got = coeffs.copy()
    tools.joint_stereo(got, pairs, desc)
    tools.tns(got, filt)
    assert bit_equal(got, want)
    side = oracle.aac_side(np.zeros((2, frames), int), np.ones((2, frames), int), np.ones((2, frames), int))
    delay = np.zeros((2, 1024), np.float32)
    pcm = AacDsp(emu_ctx).synth(got, side, delay)[0]
    assert bit_equal(pcm, oracle.aac_synth(want, side, delay)[0])


def test_joint_stereo_rejects_bad_band_tables(emu_ctx):
    from symphonia_amd import AacSpectralTools, SymaccelError
    coeffs = np.zeros((2, 1, 1024), np.float32)
    pairs = np.array([[0, 1]], np.int32)
    desc = np.zeros((1, 1), oracle_dtype_js())
    for bad in ([0, 4, 6, 1024], [0, 8, 4], [4, 8], [0, 4, 1028]):
        with pytest.raises(SymaccelError):
            AacSpectralTools(emu_ctx, bad, SWB_SHORT).joint_stereo(coeffs, pairs, desc)


# ---------------------------------------------------------------- MI355X

@pytest.mark.gpu
def test_gpu_aac_tools():
    import torch
    from symphonia_amd import AacSpectralTools, Context
    if not torch.cuda.is_available():
        pytest.fail("no GPU visible")
    rng = np.random.default_rng(77)
    chains, frames = 12, 33
    coeffs = rng.standard_normal((chains, frames, 1024)).astype(np.float32)
    pairs = np.array([[0, 1], [2, 3], [9, 4], [10, 11]], np.int32)
    desc = np.array([[js_frame(rng, short=bool(rng.integers(0, 4) == 0)) for _ in range(frames)] for _ in pairs])
    _, filt = tns_case(rng, chains * frames, 700)
    want_js = js_reference(coeffs, pairs, desc)
    want = tns_reference(want_js.reshape(-1, 1024), filt, chains * frames).reshape(coeffs.shape)
    with Context(0) as ctx:
        tools = AacSpectralTools(ctx, SWB_LONG, SWB_SHORT)
        d = torch.from_numpy(coeffs).cuda()
        tools.joint_stereo(d, torch.from_numpy(pairs).cuda(), torch.from_numpy(desc.view(np.uint8).reshape(len(pairs), frames, 644)).cuda())
        ctx.sync()
        assert bit_equal(d.cpu().numpy(), want_js)
        tools.tns(d, torch.from_numpy(filt.view(np.uint8).reshape(-1, 92)).cuda())
        ctx.sync()
        assert bit_equal(d.cpu().numpy(), want)


@pytest.mark.gpu
@pytest.mark.parametrize("direct", ["0", "1"])
@pytest.mark.parametrize("seed,n_frames", [(13, 64), (14, 1000), (15, 333)])
def test_gpu_tns_ragged_quads(seed, n_frames, direct, monkeypatch):
    import torch
    from symphonia_amd import AacSpectralTools, Context
    if not torch.cuda.is_available():
        pytest.fail("no GPU visible")
    monkeypatch.setenv("SYMACCEL_TNS_DIRECT", direct)
    coeffs, filt = tns_case_ragged(np.random.default_rng(seed), n_frames)
    want = tns_reference(coeffs, filt, n_frames)
    with Context(0) as ctx:
        d = torch.from_numpy(coeffs).cuda()
        AacSpectralTools(ctx, SWB_LONG, SWB_SHORT).tns(d, torch.from_numpy(filt.view(np.uint8).reshape(-1, 92)).cuda())
        ctx.sync()
        assert bit_equal(d.cpu().numpy(), want)


@pytest.mark.gpu
@pytest.mark.parametrize("direct", ["0", "1"])
@pytest.mark.parametrize("orders", TNS_ORDER_CLASSES)
def test_gpu_tns_order_classes(orders, direct, monkeypatch):
    import torch
    from symphonia_amd import AacSpectralTools, Context
    if not torch.cuda.is_available():
        pytest.fail("no GPU visible")
    monkeypatch.setenv("SYMACCEL_TNS_DIRECT", direct)
    n_frames = 1500
    coeffs, filt = tns_case_ragged(np.random.default_rng(200 + orders[0] + 20 * orders[1]), n_frames, orders)
    filt[5]["order"] = 0
    filt[77]["frame"] = n_frames + 1
    # non-finite and signed-zero lines: a tap that does not apply subtracts +0.0, the others are the reference's rounded operations
    coeffs[3, 100:140] = [np.inf, -np.inf, np.nan, -0.0] * 10
    coeffs[9, :64] = -0.0
    want = tns_reference(coeffs, filt, n_frames)
    with Context(0) as ctx:
        d = torch.from_numpy(coeffs).cuda()
        AacSpectralTools(ctx, SWB_LONG, SWB_SHORT).tns(d, torch.from_numpy(filt.view(np.uint8).reshape(-1, 92)).cuda())
        ctx.sync()
        assert equal_mod_nan(d.cpu().numpy(), want)
