"""Stages the reference has NO test for ("parity unpinned by the reference"):
pin the oracle with independent f64 definitions from the codec specifications and with
invertibility properties (analysis -> oracle synthesis == identity)."""
import math

import numpy as np
import pytest

import oracle
from helpers import (aac_sequence_chain, imdct_analytical, mdct_forward, imdct36_analytical,
                     imdct12_analytical)

# ----------------------------------------------------------------------------- AAC


def test_aac_sine_window_closed_form():
    for size in (1024, 128):
        w = oracle.aac_window(False, 0.0, size)
        n = np.arange(size)
        assert np.abs(w - np.sin((n + 0.5) * np.pi / (2 * size))).max() < 2e-7


def test_aac_kbd_power_complementary():
    # Princen-Bradley: w[n]^2 + w[N-1-n]^2 == 1 for the half-window pair.
    for alpha, size in ((4.0, 1024), (6.0, 128)):
        w = oracle.aac_window(True, alpha, size).astype(np.float64)
        assert np.all(np.diff(w) >= 0)
        assert np.abs(w ** 2 + w[::-1] ** 2 - 1.0).max() < 2e-3  # the reference's "+1" normaliser
        assert abs(w[-1] - 1.0) < 1e-3


def _aac_windows():
    return {
        (1, "long"): oracle.aac_window(True, 4.0, 1024).astype(np.float64),
        (1, "short"): oracle.aac_window(True, 6.0, 128).astype(np.float64),
        (0, "long"): oracle.aac_window(False, 0.0, 1024).astype(np.float64),
        (0, "short"): oracle.aac_window(False, 0.0, 128).astype(np.float64),
    }


def test_aac_only_long_closed_form():
    """out_t[i] = w[i] y_t[i] + w[1023-i] y_{t-1}[1024+i]."""
    rng = np.random.default_rng(11)
    W = _aac_windows()
    nfr = 5
    coeffs = rng.standard_normal((1, nfr, 1024)).astype(np.float32)
    shape = np.array([1, 0, 0, 1, 1], dtype=np.uint8)
    prev = np.array([0, 1, 0, 0, 1], dtype=np.uint8)
    side = oracle.aac_side(np.zeros(nfr, np.uint8), shape, prev)[None, :]
    pcm, delay = oracle.aac_synth(coeffs, side, np.zeros((1, 1024), np.float32))
    y = imdct_analytical(coeffs[0], 1.0 / 2048.0)  # [nfr, 2048] f64
    for t in range(nfr):
        tail = np.zeros(1024) if t == 0 else y[t - 1, 1024:] * W[(shape[t - 1], "long")][::-1]
        exp = tail + y[t, :1024] * W[(prev[t], "long")]
        assert np.abs(pcm[0, t] - exp).max() < 1e-5 * max(1.0, np.abs(exp).max())
    assert np.abs(delay[0] - y[-1, 1024:] * W[(shape[-1], "long")][::-1]).max() < 1e-5


def _aac_analysis(x, seq, shape, prev):
    """ISO/IEC 14496-3 4.6.11 analysis windowing + MDCT of frame blocks (2048 samples each)."""
    W = _aac_windows()
    nfr = len(seq)
    coeffs = np.zeros((nfr, 1024))
    for t in range(nfr):
        blk = x[t * 1024:t * 1024 + 2048]
        lw, ls = W[(prev[t], "long")], W[(prev[t], "short")]
        rw, rs = W[(shape[t], "long")][::-1], W[(shape[t], "short")][::-1]
        if seq[t] == 2:
            for w in range(8):
                sub = blk[448 + 128 * w:448 + 128 * w + 256]
                left = ls if w == 0 else W[(shape[t], "short")]
                win = np.concatenate((left, rs))
                coeffs[t, 128 * w:128 * w + 128] = mdct_forward(sub * win)
            continue
        win = np.empty(2048)
        if seq[t] in (0, 1):
            win[:1024] = lw
        else:  # LONG_STOP
            win[:448] = 0.0
            win[448:576] = ls
            win[576:1024] = 1.0
        if seq[t] in (0, 3):
            win[1024:] = rw
        else:  # LONG_START
            win[1024:1472] = 1.0
            win[1472:1600] = rs
            win[1600:] = 0.0
        coeffs[t] = mdct_forward(blk * win)
    return coeffs


@pytest.mark.parametrize("seed", [1, 2, 3])
def test_aac_tdac_perfect_reconstruction_all_sequences(seed):
    """analysis (spec windows, all four window sequences, both shapes) -> Dsp::synth == 0.25 * x  (scale * N/2 = 1024/2/2048)."""
    rng = np.random.default_rng(seed)
    nfr = 24
    seq, shape, prev = aac_sequence_chain(rng, nfr, p_switch=0.4)
    x = rng.standard_normal(1024 * (nfr + 1))
    x[:1024] = 0.0  # decoder starts from a zero delay line
    coeffs = _aac_analysis(x, seq, shape, prev).astype(np.float32)[None]
    side = oracle.aac_side(seq, shape, prev)[None]
    pcm, _ = oracle.aac_synth(coeffs, side, np.zeros((1, 1024), np.float32))
    got = pcm[0].reshape(-1)[1024:]
    want = 0.25 * x[1024:1024 * nfr]
    # KBD windows in the reference are normalised with the "+1" term -> ~1e-3 relative leakage
    assert np.abs(got - want).max() < 4e-3 * np.abs(x).max()
    sine_only = (shape == 0).all() and (prev == 0).all()
    if sine_only:
        assert np.abs(got - want).max() < 2e-5 * np.abs(x).max()


def test_aac_tdac_sine_windows_tight():
    rng = np.random.default_rng(5)
    nfr = 20
    seq, _, _ = aac_sequence_chain(rng, nfr, p_switch=0.5)
    z = np.zeros(nfr, np.uint8)
    x = rng.standard_normal(1024 * (nfr + 1))
    x[:1024] = 0.0
    coeffs = _aac_analysis(x, seq, z, z).astype(np.float32)[None]
    pcm, _ = oracle.aac_synth(coeffs, oracle.aac_side(seq, z, z)[None], np.zeros((1, 1024), np.float32))
    got = pcm[0].reshape(-1)[1024:]
    assert set(seq.tolist()) == {0, 1, 2, 3}
    assert np.abs(got - 0.25 * x[1024:1024 * nfr]).max() < 3e-5 * np.abs(x).max()


# ----------------------------------------------------------------------------- MP3

def _iso_polyphase(slots):
    """ISO/IEC 11172-3 figure A.2: slots[t][32] -> pcm[t][32], f64, zero initial V."""
    D = oracle.mp3_synthesis_window().astype(np.float64)
    i = np.arange(64)[:, None]
    k = np.arange(32)[None, :]
    N = np.cos((16 + i) * (2 * k + 1) * np.pi / 64)
    V = np.zeros(1024)
    out = []
    for s in slots:
        V = np.concatenate((N @ s, V[:-64]))
        U = np.zeros(512)
        for j in range(8):
            U[64 * j:64 * j + 32] = V[128 * j:128 * j + 32]
            U[64 * j + 32:64 * j + 64] = V[128 * j + 96:128 * j + 128]
        out.append((U * D).reshape(16, 32).sum(axis=0))
    return np.array(out)


def test_mp3_polyphase_matches_iso_definition():
    rng = np.random.default_rng(7)
    n_slots = 18 * 3
    slots = rng.standard_normal((n_slots, 32))
    want = _iso_polyphase(slots)
    v = np.zeros((16, 64), np.float32)
    vf = 0
    got = []
    for g in range(3):
        blk = slots[18 * g:18 * g + 18].T.astype(np.float32).copy()  # [32][18]: in[18*i + b]
        out, v, vf = oracle.mp3_polyphase(v, vf, 18, blk.reshape(-1))
        got.append(out.reshape(18, 32))
    got = np.concatenate(got)
    assert np.abs(got - want).max() < 2e-5
    assert vf == (15 * n_slots) % 16


def test_mp3_imdct36_windows_and_overlap():
    rng = np.random.default_rng(8)
    W = oracle.mp3_imdct_windows()
    n = np.arange(36)
    assert np.abs(W[0] - np.sin(np.pi / 36 * (n + 0.5))).max() < 1e-7
    assert np.all(W[1][30:] == 0) and np.all(W[1][18:24] == 1) and np.all(W[3][:6] == 0)
    assert np.abs(W[2][:12] - np.sin(np.pi / 12 * (np.arange(12) + 0.5))).max() < 1e-7
    for bt in (0, 1, 3):
        x = rng.standard_normal(18).astype(np.float32)
        ov = rng.standard_normal(18).astype(np.float32)
        out, ov2 = oracle.mp3_imdct36(x, W[bt], ov)
        y = imdct36_analytical(x) * W[bt]
        assert np.abs(out - (y[:18] + ov)).max() < 2e-5
        assert np.abs(ov2 - y[18:]).max() < 2e-5


def _np_hybrid(buf, overlap, bt, mixed, rzero, sr):
    """Independent f64 model of reorder+antialias+hybrid+freq-inversion from ISO 11172-3 2.4.3.4."""
    W = oracle.mp3_imdct_windows().astype(np.float64)
    short, mixed_b, switch = oracle.mp3_sfb_tables(sr)
    x = buf.astype(np.float64).copy()
    x[rzero:] = 0.0
    if bt == 2:
        bands = mixed_b[switch:] if mixed else short
        y = x.copy()
        for b in range(0, len(bands) - 3, 3):
            s0, s1, s2, s3 = bands[b:b + 4]
            wl = s1 - s0
            for k in range(wl):
                y[s0 + 3 * k + 0] = x[s0 + k]
                y[s0 + 3 * k + 1] = x[s1 + k]
                y[s0 + 3 * k + 2] = x[s2 + k]
        x = y
    c = np.array([-0.6, -0.535, -0.33, -0.185, -0.095, -0.041, -0.0142, -0.0037])
    cs, ca = 1 / np.sqrt(1 + c * c), c / np.sqrt(1 + c * c)
    n_aa = 0 if (bt == 2 and not mixed) else (1 if bt == 2 else 31)
    for sb in range(1, n_aa + 1):
        for i in range(8):
            lo, up = x[18 * sb - 1 - i], x[18 * sb + i]
            x[18 * sb - 1 - i] = lo * cs[i] - up * ca[i]
            x[18 * sb + i] = up * cs[i] + lo * ca[i]
    out = np.zeros(576)
    new_ov = np.zeros(576)
    for sb in range(32):
        xs = x[18 * sb:18 * sb + 18]
        if bt == 2 and mixed and sb >= 2:
            # REFERENCE QUIRK kept for parity: antialias() clamps rzero to 36 for mixed blocks
            # (hybrid_synthesis.rs:233-239 with sb_limit = 2), so hybrid_synthesis() treats every
            # short sub-band of a mixed block as zeroed: output = overlap, overlap = 0.
            out[18 * sb:18 * sb + 18] = overlap[18 * sb:18 * sb + 18]
            continue
        is_long = not (bt == 2 and (not mixed or sb >= 2))
        if is_long:
            y = imdct36_analytical(xs) * W[{0: 0, 1: 1, 3: 3, 2: 0}[bt]]
        else:
            y = np.zeros(36)
            for w in range(3):
                y[6 + 6 * w:18 + 6 * w] += imdct12_analytical(xs[w::3]) * W[2][:12]
        out[18 * sb:18 * sb + 18] = y[:18] + overlap[18 * sb:18 * sb + 18]
        new_ov[18 * sb:18 * sb + 18] = y[18:]
    for sb in range(1, 32, 2):
        out[18 * sb + 1:18 * sb + 18:2] *= -1
    return out, new_ov


@pytest.mark.parametrize("bt,mixed", [(0, 0), (1, 0), (3, 0), (2, 0), (2, 1)])
@pytest.mark.parametrize("sr", [0, 1, 4, 8])
def test_mp3_hybrid_chain_matches_spec_model(bt, mixed, sr):
    rng = np.random.default_rng(100 * bt + 10 * mixed + sr)
    for rzero in (576, 400, 342, 36, 20, 0):
        buf = rng.standard_normal(576).astype(np.float32)
        buf[rzero:] = 0.0
        ov = rng.standard_normal(576).astype(np.float32)
        b, rz = oracle.mp3_reorder(buf, bt, mixed, sr, rzero)
        b, rz = oracle.mp3_antialias(b, bt, mixed, rz)
        b, ov2 = oracle.mp3_hybrid(b, ov, bt, mixed, rz)
        b = oracle.mp3_frequency_inversion(b)
        want, want_ov = _np_hybrid(buf, ov, bt, mixed, rzero, sr)
        assert np.abs(b - want).max() < 5e-5, (bt, mixed, sr, rzero)
        assert np.abs(ov2 - want_ov).max() < 5e-5, (bt, mixed, sr, rzero)


def test_mp3_batch_equals_stagewise():
    rng = np.random.default_rng(9)
    ngr = 6
    xr = rng.standard_normal((2, ngr, 576)).astype(np.float32)
    bt = np.array([[0, 1, 2, 2, 3, 0], [0, 0, 1, 2, 3, 0]], np.uint8)
    mx = np.array([[0, 0, 1, 0, 0, 0], [0, 0, 0, 0, 0, 0]], np.uint8)
    rz = np.array([[576, 500, 300, 576, 100, 0], [576, 18, 36, 54, 574, 576]], np.uint16)
    for c in range(2):
        for g in range(ngr):
            xr[c, g, rz[c, g]:] = 0
    side = oracle.mp3_side(bt, mx, rz)
    pcm, ov, vv, vf = oracle.mp3_synth(xr, side, 0, np.zeros((2, 576), np.float32),
                                       np.zeros((2, 1024), np.float32), np.zeros(2, np.int32))
    for c in range(2):
        o = np.zeros(576, np.float32)
        v = np.zeros(1024, np.float32)
        f = 0
        for g in range(ngr):
            b, r = oracle.mp3_reorder(xr[c, g], bt[c, g], mx[c, g], 0, int(rz[c, g]))
            b, r = oracle.mp3_antialias(b, bt[c, g], mx[c, g], r)
            b, o = oracle.mp3_hybrid(b, o, bt[c, g], mx[c, g], r)
            b = oracle.mp3_frequency_inversion(b)
            out, v, f = oracle.mp3_polyphase(v, f, 18, b)
            assert np.array_equal(out.view(np.uint32), pcm[c, g].view(np.uint32))
        assert np.array_equal(o, ov[c]) and np.array_equal(v, vv[c]) and f == vf[c]


# ----------------------------------------------------------------------------- Vorbis

def test_vorbis_window_closed_form():
    for bs in (256, 2048, 64, 8192):
        w = oracle.vorbis_window(bs).astype(np.float64)
        i = np.arange(bs // 2)
        want = np.sin(np.pi / 2 * np.sin(np.pi / 2 * (i + 0.5) / (bs // 2)) ** 2)
        assert np.abs(w - want).max() < 1e-7
        assert np.abs(w ** 2 + w[::-1] ** 2 - 1).max() < 1e-6


def _vorbis_analysis(x, flags, bs0, bs1):
    """Vorbis I 4.3.1 window shapes + forward MDCT, block centres advancing by (prev+cur)/4."""
    W = {bs0: oracle.vorbis_window(bs0).astype(np.float64), bs1: oracle.vorbis_window(bs1).astype(np.float64)}
    spectra, centre = [], bs1 // 2  # centre of block 0 inside x
    for b, f in enumerate(flags):
        n = bs1 if f else bs0
        pf = flags[b - 1] if b > 0 else f
        nf = flags[b + 1] if b + 1 < len(flags) else f
        if b > 0:
            centre += ((bs1 if pf else bs0) + n) // 4
        win = np.zeros(n)
        ln = (bs1 if pf else bs0) if f else bs0
        rn = (bs1 if nf else bs0) if f else bs0
        ln, rn = min(ln, n), min(rn, n)
        ls, rs = n // 4 - ln // 4, 3 * n // 4 - rn // 4
        win[ls:ls + ln // 2] = W[ln]
        win[ls + ln // 2:rs] = 1.0
        win[rs:rs + rn // 2] = W[rn][::-1]
        blk = x[centre - n // 2:centre + n // 2]
        spectra.append(mdct_forward(blk * win))
    return spectra


@pytest.mark.parametrize("seed", [21, 22])
def test_vorbis_tdac_mixed_blocksizes(seed):
    """analysis with spec windows (long/short transitions) -> DspChannel::synth == (bs/4)-scaled x."""
    rng = np.random.default_rng(seed)
    bs0e, bs1e = 6, 9
    bs0, bs1 = 1 << bs0e, 1 << bs1e
    flags = [1, 1, 0, 0, 1, 0, 1, 1, 0, 0, 0, 1, 1]
    total = bs1 * (len(flags) + 2)
    x = rng.standard_normal(total)
    spectra = _vorbis_analysis(x, flags, bs0, bs1)
    # IMDCT(unscaled) of an unscaled MDCT returns (N/2)*signal for N spectral lines; normalise
    packed = np.concatenate([s / (len(s) / 2) for s in spectra]).astype(np.float32)[None]
    bf = np.array(flags, np.uint8)[None]
    spec_off, pcm_off = oracle.vorbis_layout(bs0e, bs1e, bf, np.array([-1]))
    pcm, ov, pf = oracle.vorbis_synth(bs0e, bs1e, packed, bf, np.array([-1], np.int32),
                                      np.zeros((1, bs1 // 2), np.float32), int(pcm_off[0, -1]))
    # output of block b (b >= 1) covers x[centre_{b-1} .. centre_b)
    centre = bs1 // 2
    for b in range(1, len(flags)):
        n_prev = bs1 if flags[b - 1] else bs0
        n = bs1 if flags[b] else bs0
        ln = (n_prev + n) // 4
        got = pcm[0, pcm_off[0, b]:pcm_off[0, b + 1]]
        want = x[centre:centre + ln]
        assert np.abs(got - want).max() < 2e-4 * np.abs(x).max(), b
        centre += ln
    assert pf[0] == flags[-1]


def test_vorbis_inverse_coupling_table():
    m = np.array([1.0, 1.0, -1.0, -1.0, 0.0, 0.0, 2.5, -2.5], np.float32)
    a = np.array([0.5, -0.5, 0.5, -0.5, 1.0, -1.0, 0.0, 0.0], np.float32)
    nm, na = oracle.vorbis_inverse_coupling(m, a)
    # Vorbis I spec 4.3.5
    assert nm.tolist() == [1.0, 0.5, -1.0, -0.5, 0.0, 1.0, 2.5, -2.5]
    assert na.tolist() == [0.5, 1.0, -0.5, -1.0, 1.0, 0.0, 2.5, -2.5]


def test_vorbis_floor1_render_against_spec_model():
    """Independent Python model of Vorbis I 7.2.4 (integer render_line) with the dB table."""
    rng = np.random.default_rng(31)
    table = oracle.vorbis_floor1_table()
    for trial in range(20):
        n = int(rng.choice([128, 1024]))
        n_posts = int(rng.integers(2, 30))
        xs = [0, n] + sorted(rng.choice(np.arange(1, n), size=n_posts - 2, replace=False).tolist())
        rng.shuffle(xs[2:])
        xs = [0, n] + rng.permutation(xs[2:]).tolist()
        mult = int(rng.integers(1, 5))
        rng_range = [256, 128, 86, 64][mult - 1]
        ys = rng.integers(0, rng_range, size=n_posts)
        ys[rng.random(n_posts) < 0.3] = 0
        got = oracle.vorbis_floor1(xs, ys, mult, n)
        # spec model
        def low_high(i):
            lo = max((j for j in range(i) if xs[j] < xs[i]), key=lambda j: xs[j])
            hi = min((j for j in range(i) if xs[j] > xs[i]), key=lambda j: xs[j])
            return lo, hi
        def render_point(x0, y0, x1, y1, x):
            dy, adx = y1 - y0, x1 - x0
            off = abs(dy) * (x - x0) // adx
            return y0 - off if dy < 0 else y0 + off
        fy, fl = [int(ys[0]), int(ys[1])], [True, True]
        for i in range(2, n_posts):
            lo, hi = low_high(i)
            pred = render_point(xs[lo], fy[lo], xs[hi], fy[hi], xs[i])
            val, hr, lr = int(ys[i]), rng_range - pred, pred
            room = 2 * min(hr, lr)
            if val:
                fl[lo] = fl[hi] = True
                fl.append(True)
                if val >= room:
                    fy.append(val - lr + pred if hr > lr else pred - val + hr - 1)
                else:
                    fy.append(pred - (val + 1) // 2 if val & 1 else pred + val // 2)
            else:
                fl.append(False)
                fy.append(pred)
        order = sorted(range(n_posts), key=lambda i: xs[i])
        curve = np.zeros(n, dtype=np.int64)
        lx, ly = 0, min(max(fy[order[0]] * mult, 0), 255)
        hx = hy = 0
        for i in order[1:]:
            if fl[i]:
                hy, hx = min(max(fy[i] * mult, 0), 255), xs[i]
                # spec render_line (integer Bresenham)
                dy, adx = hy - ly, hx - lx
                base = abs(dy) // adx * (1 if dy >= 0 else -1)
                sy = base - 1 if dy < 0 else base + 1
                ady = abs(dy) - abs(base) * adx
                y, err = ly, 0
                if lx < n:
                    curve[lx] = y
                for x in range(lx + 1, min(hx, n)):
                    err += ady
                    if err >= adx:
                        err -= adx
                        y += sy
                    else:
                        y += base
                    curve[x] = y
                lx, ly = hx, hy
        if hx < n:
            curve[hx:] = hy
        assert np.array_equal(got, table[curve]), trial


# ----------------------------------------------------------------------------- FLAC

def _lpc_encode(x, coeffs, shift, order):
    res = x.astype(np.int64).copy()
    for i in range(order, len(x)):
        pred = sum(int(coeffs[j]) * int(x[i - 1 - j]) for j in range(order)) >> shift
        res[i] = x[i] - pred
    return res


@pytest.mark.parametrize("order", [1, 2, 4, 5, 8, 11, 12, 13, 32])
def test_flac_lpc_inverts_forward_encoder(order):
    rng = np.random.default_rng(order)
    n = 300
    x = np.round(np.cumsum(rng.standard_normal(n)) * 50000).astype(np.int64)
    x = np.clip(x, -(1 << 23), (1 << 23) - 1)
    coeffs = rng.integers(-(1 << 11), 1 << 11, size=order)
    shift = int(rng.integers(8, 15))
    res = _lpc_encode(x, coeffs, shift, order)
    res32 = ((res + (1 << 31)) % (1 << 32) - (1 << 31)).astype(np.int32)  # residual may wrap: decoder wraps back
    got = oracle.flac_lpc_predict(order, coeffs.astype(np.int32), shift, res32)
    assert np.array_equal(got, x.astype(np.int32))


def test_flac_lpc_short_blocks_and_prefill():
    rng = np.random.default_rng(3)
    for order, n in ((3, 3), (3, 4), (9, 9), (9, 10), (20, 25), (32, 32), (32, 33), (1, 1)):
        x = rng.integers(-(1 << 15), 1 << 15, size=n).astype(np.int64)
        coeffs = rng.integers(-500, 500, size=order)
        res = _lpc_encode(x, coeffs, 9, order).astype(np.int32)
        assert np.array_equal(oracle.flac_lpc_predict(order, coeffs.astype(np.int32), 9, res), x.astype(np.int32))


@pytest.mark.parametrize("order", [0, 1, 2, 3, 4])
def test_flac_fixed_inverts_finite_differences(order):
    rng = np.random.default_rng(40 + order)
    x = rng.integers(-(1 << 23), 1 << 23, size=257).astype(np.int64)
    res = x.copy()
    taps = {0: [], 1: [1], 2: [2, -1], 3: [3, -3, 1], 4: [4, -6, 4, -1]}[order]
    for i in range(order, len(x)):
        res[i] = x[i] - sum(t * x[i - 1 - j] for j, t in enumerate(taps))
    res32 = ((res + (1 << 31)) % (1 << 32) - (1 << 31)).astype(np.int32)
    assert np.array_equal(oracle.flac_fixed_predict(order, res32), x.astype(np.int32))


def test_flac_decorrelation_inverts_encoder():
    rng = np.random.default_rng(50)
    l = rng.integers(-(1 << 23), 1 << 23, size=1000).astype(np.int64)
    r = rng.integers(-(1 << 23), 1 << 23, size=1000).astype(np.int64)
    side = l - r
    a, b = oracle.flac_decorrelate(1, l.astype(np.int32), side.astype(np.int32))
    assert np.array_equal(a, l) and np.array_equal(b, r)
    a, b = oracle.flac_decorrelate(2, ((l + r) >> 1).astype(np.int32), side.astype(np.int32))
    assert np.array_equal(a, l) and np.array_equal(b, r)
    a, b = oracle.flac_decorrelate(3, side.astype(np.int32), r.astype(np.int32))
    assert np.array_equal(a, l) and np.array_equal(b, r)
    assert np.array_equal(oracle.flac_shl(np.array([1, -1, 0x7FFFFF], np.int32), 8),
                          np.array([256, -256, 0x7FFFFF00], np.int32))
