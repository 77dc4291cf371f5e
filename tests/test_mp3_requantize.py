"""MP3 Layer III requantisation (layer3/requantize.rs:28-31, 117-147, 239-380).

The reference has no test for this stage, so the oracle's restatement is "parity unpinned by the reference"; it is
pinned here by (a) the band-edge and pre-emphasis tables recorded from the reference (tests/golden/ref_kats.json),
(b) the ISO/IEC 11172-3 2.4.3.4 closed form xr = sign(s) |s|^(4/3) 2^((A - B)/4) evaluated in f64, and (c) the structure
of the reference's band loops (which lines each scale factor reaches, incl. the unscaled lines of a mixed block).
The kernel is then compared bit for bit with the oracle: in CPU emulation here, on the MI355X under -m gpu."""
import json
from pathlib import Path

import numpy as np
import pytest

import oracle
from emu_lib import emu_ctx  # noqa: F401
from helpers import bit_equal

KATS = json.loads((Path(__file__).parent / "golden" / "ref_kats.json").read_text())
SHORT, LONG, START, END = 2, 0, 1, 3


def make_case(seed, n, kinds=("long", "short", "mixed", "start", "end"), big=False):
    rng = np.random.default_rng(seed)
    # Laplacian-ish magnitudes like real spectra, a few linbits-sized ones, exact zeros, count1-style +-1
    q = np.rint(rng.laplace(0, 6, (n, 576))).astype(np.int64)
    q[rng.random((n, 576)) < 0.02] = 0
    sel = rng.random((n, 576)) < 0.01
    q[sel] = rng.integers(-8206, 8207, int(sel.sum()))
    if big:
        q[:, ::97] = 8206
        q[:, 1::97] = -8206
    d = np.zeros(n, oracle.MP3_REQUANT_DTYPE)
    d["global_gain"] = rng.integers(0, 256, n)
    d["flags"] = rng.integers(0, 4, n)
    kind = rng.choice(kinds, n)
    d["block_type"] = [dict(long=LONG, short=SHORT, mixed=SHORT, start=START, end=END)[k] for k in kind]
    d["is_mixed"] = [1 if k == "mixed" else 0 for k in kind]
    d["subblock_gain"] = rng.integers(0, 8, (n, 3))
    d["scalefacs"] = rng.integers(0, 16, (n, 39))
    d["scalefacs"][rng.random(n) < 0.1] = 252  # the largest value the domain allows
    rz = rng.integers(0, 577, n)
    rz[rng.random(n) < 0.3] = 576
    rz[:2] = (0, 576)[:min(n, 2)]
    d["rzero"] = rz
    for i in range(n):
        q[i, rz[i]:] = 0 if i % 2 else q[i, rz[i]:]  # odd cases: the parser's zero fill; even: stale values to be ignored
    return q.astype(np.int16), d


# ---------------------------------------------------------------- oracle pins

def test_band_tables_match_the_reference():
    for sr in range(9):
        assert list(oracle.mp3_sfb_long(sr)) == KATS["mp3_sfb_long"][sr]
        short, mixed, switch = oracle.mp3_sfb_tables(sr)
        assert list(short) == KATS["mp3_sfb_short"][sr]
        assert list(mixed) == KATS["mp3_sfb_mixed"][sr]
        assert switch == KATS["mp3_sfb_mixed_switch"][sr]


def test_pow_tables_closed_form():
    p = oracle.mp3_pow43().astype(np.float64)
    i = np.arange(8207, dtype=np.float64)
    assert p[0] == 0.0 and p[1] == 1.0
    # the exponent is the f32 nearest to 4/3 (requantize.rs:29: both powf arguments are f32), so 8^(4/3) is not 16
    np.testing.assert_allclose(p[1:], i[1:] ** float(np.float32(4.0 / 3.0)), rtol=1.2e-7)
    np.testing.assert_allclose(p[1:], i[1:] ** (4.0 / 3.0), rtol=5e-7)  # ln(8206) * 2^-25 exponent error + rounding
    e = oracle.mp3_pow2ab()
    k = np.arange(oracle.MP3_POW2AB_LEN) + oracle.MP3_POW2AB_MIN_E
    want = (2.0 ** (0.25 * k.astype(np.float64))).astype(np.float32)  # exact powers of two and their 2^(1/4) multiples
    assert bit_equal(e[k % 4 == 0], want[k % 4 == 0])  # 2^integer is exact in every libm
    big = k > -500  # the rest: correctly rounded up to 1 ulp of libm `pow`
    np.testing.assert_allclose(e[big].astype(np.float64), 2.0 ** (0.25 * k[big]), rtol=1.2e-7)


@pytest.mark.parametrize("sr", [0, 1, 2, 3, 4, 8])
def test_oracle_requantize_closed_form(sr):
    q, d = make_case(10 + sr, 60)
    got = oracle.mp3_requantize(q, d, sr).astype(np.float64)
    pre_tab = KATS["mp3_pre_emphasis"]
    pow43, pow2ab = oracle.mp3_pow43(), oracle.mp3_pow2ab()
    for g in range(q.shape[0]):
        ch = d[g]
        rz = int(ch["rzero"])
        shift = 2 if ch["flags"] & 1 else 1
        expo = np.full(576, np.nan)  # (A - B) per line; nan = no band reaches the line
        if ch["block_type"] == SHORT:
            mixed = bool(ch["is_mixed"])
            edges = KATS["mp3_sfb_mixed"][sr] if mixed else KATS["mp3_sfb_short"][sr]
            sw = KATS["mp3_sfb_mixed_switch"][sr] if mixed else 0
            for i in range(sw - 1):  # requantize_long over bands[..switch]: switch edges = switch - 1 bands
                pre = pre_tab[i] if ch["flags"] & 2 else 0
                expo[edges[i]:edges[i + 1]] = int(ch["global_gain"]) - 210 - ((int(ch["scalefacs"][i]) + pre) << shift)
            for i in range(len(edges) - sw - 1):
                a = int(ch["global_gain"]) - 210 - 8 * int(ch["subblock_gain"][i % 3])
                expo[edges[sw + i]:edges[sw + i + 1]] = a - (int(ch["scalefacs"][sw + i]) << shift)
        else:
            edges = KATS["mp3_sfb_long"][sr]
            for i in range(22):
                pre = pre_tab[i] if ch["flags"] & 2 else 0
                expo[edges[i]:edges[i + 1]] = int(ch["global_gain"]) - 210 - ((int(ch["scalefacs"][i]) + pre) << shift)
        # the reference's arithmetic: +-POW43[|s|] (exact sign), then ONE rounded f32 multiply by the band's
        # `2^(0.25 e) as f32` (itself possibly a denormal); the two tables are pinned by test_pow_tables_closed_form
        s = q[g].astype(np.int64)
        v = np.where(s < 0, -pow43[np.abs(s)], pow43[np.abs(s)]).astype(np.float32)
        v[s == 0] = 0.0
        v[rz:] = 0.0
        scale = np.where(np.isnan(expo), np.float32(1.0),
                         pow2ab[np.nan_to_num(expo).astype(np.int64) - oracle.MP3_POW2AB_MIN_E]).astype(np.float32)
        want = v * scale
        want[np.isnan(expo)] = v[np.isnan(expo)]
        assert bit_equal(got[g].astype(np.float32), want), (g, np.flatnonzero(got[g].astype(np.float32) != want)[:5])
        # and the closed form of ISO/IEC 11172-3 2.4.3.4 wherever the scale is a normal f32
        normal = ~np.isnan(expo) & (np.nan_to_num(expo) > -500) & (s != 0) & (np.arange(576) < rz)
        cf = np.sign(s[normal]) * np.abs(s[normal]).astype(np.float64) ** (4.0 / 3.0) * 2.0 ** (0.25 * expo[normal])
        np.testing.assert_allclose(got[g][normal], cf, rtol=8e-7)
        assert not np.signbit(got[g][q[g] == 0]).any() and not np.signbit(got[g][rz:]).any()  # zeros are +0.0


def test_mixed_block_leaves_the_gap_unscaled():
    # requantize.rs:368-372: the long part gets bands[..switch], whose last edge is one band short of the first short band
    q = np.full((1, 576), 3, np.int16)
    d = np.zeros(1, oracle.MP3_REQUANT_DTYPE)
    d["global_gain"], d["block_type"], d["is_mixed"], d["rzero"] = 150, SHORT, 1, 576
    p3 = oracle.mp3_pow43()[3]
    for sr, gap in ((0, (30, 36)), (3, (30, 36)), (8, (24, 36))):
        x = oracle.mp3_requantize(q, d, sr)[0]
        assert (x[gap[0]:gap[1]] == p3).all() and (x[:gap[0]] != p3).all() and (x[gap[1]:] != p3).all()


# ---------------------------------------------------------------- kernel vs oracle

@pytest.mark.parametrize("sr,n", [(0, 37), (1, 16), (4, 33), (8, 5), (2, 1)])
def test_emu_mp3_requantize(emu_ctx, sr, n):
    from symphonia_amd import Mp3Requantize
    q, d = make_case(sr * 7 + n, n, big=True)
    got = Mp3Requantize(emu_ctx, sr).requantize(q, d)
    assert bit_equal(got, oracle.mp3_requantize(q, d, sr))


def test_emu_mp3_requantize_feeds_synth(emu_ctx):
    """requantize -> synthesis tail, mono: what Layer3::decode does per granule (layer3/mod.rs:393-476) without stereo."""
    from symphonia_amd import Mp3Requantize, Mp3Synthesis
    q, d = make_case(99, 6, kinds=("long", "short", "start"))
    xr = Mp3Requantize(emu_ctx, 0).requantize(q, d).reshape(2, 3, 576)
    bt, mixed, rz = d["block_type"].reshape(2, 3), d["is_mixed"].reshape(2, 3), d["rzero"].reshape(2, 3)
    ov, vv, vf = np.zeros((2, 576), np.float32), np.zeros((2, 1024), np.float32), np.zeros(2, np.int32)
    side = oracle.mp3_side(bt, mixed, rz)
    got = Mp3Synthesis(emu_ctx, 0).synth(xr, side, ov, vv, vf)
    want = oracle.mp3_synth(oracle.mp3_requantize(q, d, 0).reshape(2, 3, 576), side, 0, ov, vv, vf)
    assert bit_equal(got[0], want[0])


def test_requantize_argument_errors(emu_ctx):
    from symphonia_amd import Mp3Requantize
    with pytest.raises(ValueError):
        Mp3Requantize(emu_ctx, 9)
    assert Mp3Requantize(emu_ctx, 0).requantize(np.zeros((0, 576), np.int16), np.zeros(0, oracle.MP3_REQUANT_DTYPE)).shape == (0, 576)


@pytest.mark.gpu
@pytest.mark.parametrize("sr,n", [(0, 4099), (3, 1000), (8, 257)])
def test_gpu_mp3_requantize(sr, n):
    import torch
    from symphonia_amd import Context, Mp3Requantize
    if not torch.cuda.is_available():
        pytest.fail("no GPU visible")
    q, d = make_case(1000 + sr + n, n, big=True)
    want = oracle.mp3_requantize(q, d, sr)
    with Context(0) as ctx:
        rq = Mp3Requantize(ctx, sr)
        dq = torch.from_numpy(q).cuda()
        dd = torch.from_numpy(d.view(np.uint8).reshape(n, 52)).cuda()
        got = rq.requantize(dq, dd)
        ctx.sync()
        assert bit_equal(got.cpu().numpy(), want)
        assert bit_equal(rq.requantize(q, d), want)  # host-pointer entry point
