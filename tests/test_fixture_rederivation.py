"""An interpreter-independent guard on the integer fixture groups.

tests/golden/rs_fixtures/{flac,alac}.npz were produced by running the reference's Rust text under tools/rsinterp, and
the oracle is pinned to them.  Here the same numbers are derived a THIRD way: from the published FLAC and ALAC decoding
rules, written in Python's unbounded integers with explicit wrap-to-i32 -- no numpy arithmetic, no oracle, no
interpreter.  If the interpreter mis-executed an overflow, a shift or an iterator adaptor, this is where it shows.

  FLAC (format specification; reference call sites symphonia-bundle-flac/src/decoder.rs:32-82, 403-409, 632-644, 663-752)
  ALAC (Apple's reference decoder's dynamic predictor; symphonia-codec-alac/src/lib.rs:165-264, 659-671)"""
from pathlib import Path

import numpy as np

FIX = Path(__file__).resolve().parent / "golden" / "rs_fixtures"


def i32(v):
    return ((v + (1 << 31)) & 0xFFFFFFFF) - (1 << 31)


def ints(a):
    return [int(v) for v in a]


# ---------------------------------------------------------------------------------------------------------- FLAC
BINOMIAL = {0: [], 1: [1], 2: [2, -1], 3: [3, -3, 1], 4: [4, -6, 4, -1]}  # the fixed predictors of the format


def flac_fixed(order, res):
    s = list(res)
    for i in range(order, len(s)):
        s[i] = i32(res[i] + sum(c * s[i - 1 - j] for j, c in enumerate(BINOMIAL[order])))
    return s


def flac_lpc(order, coeffs, shift, res):
    """coeffs in bitstream order: coeffs[0] multiplies the most recent sample; the sum is 64-bit, the shift arithmetic"""
    s = list(res)
    for i in range(order, len(s)):
        acc = sum(coeffs[j] * s[i - 1 - j] for j in range(order))
        s[i] = i32(res[i] + (acc >> shift))
    return s


def flac_decorrelate(mode, a, b):
    if mode == 1:  # left / side: right = left - side
        return a, [i32(x - y) for x, y in zip(a, b)]
    if mode == 2:  # mid / side
        out0, out1 = [], []
        for m, s in zip(a, b):
            m = (m << 1) | (s & 1)
            out0.append(i32((m + s) >> 1))
            out1.append(i32((m - s) >> 1))
        return out0, out1
    return [i32(x + y) for x, y in zip(a, b)], b  # side / right: left = side + right


def test_flac_fixture_group_from_python_integers():
    f = np.load(FIX / "flac.npz")
    a, b = ints(f["decor_ch0"]), ints(f["decor_ch1"])
    for mode in (1, 2, 3):
        i0, i1 = ([v >> 1 for v in a[4:]], [v >> 2 for v in b[4:]]) if mode == 2 else (a[4:], b[4:])
        w0, w1 = flac_decorrelate(mode, i0, i1)
        assert w0 == ints(f["decor_out0_m%d" % mode]) and w1 == ints(f["decor_out1_m%d" % mode]), mode
    for sh in (0, 1, 8, 31):
        assert [i32(v << sh) for v in a] == ints(f["shl_%d" % sh]), sh
    assert [i32((w >> 1) ^ -(w & 1)) for w in ints(f["rice_in"])] == ints(f["rice_out"])  # zig-zag folding
    for order in range(5):
        assert flac_fixed(order, ints(f["fixed_in_%d" % order])) == ints(f["fixed_out_%d" % order]), order
    assert flac_fixed(4, ints(f["fixed_in_wrap"])) == ints(f["fixed_out_wrap"])
    n = 0
    while "lpc_%d_in" % n in f:
        order, shift = ints(f["lpc_%d_order_shift" % n])
        got = flac_lpc(order, ints(f["lpc_%d_coeffs" % n]), shift, ints(f["lpc_%d_in" % n]))
        assert got == ints(f["lpc_%d_out" % n]), (n, order, shift)
        n += 1
    assert n == 16


# ---------------------------------------------------------------------------------------------------------- ALAC
def clip(v, bits):
    """keep the low (32 - bits) bits of an i32, sign-extended"""
    w = 32 - bits
    v &= (1 << w) - 1
    return v - (1 << w) if v >> (w - 1) else v


def sgn(v):
    return (v > 0) - (v < 0)


def alac_predict(mode, order, shift, bps, coeffs, res):
    """the adaptive FIR predictor: warm-up by first differences, then per sample a prediction relative to the sample in
    front of the window, and a sign-LMS update of the coefficients that stops as soon as the residual is used up"""
    out, co = list(res), list(coeffs)
    if order == 0 or not out:
        return out, co
    nclip = 32 - bps
    if order == 31 or mode == 15:
        for i in range(1, len(out)):
            out[i] = clip(out[i] + out[i - 1], nclip)
    for i in range(1, min(1 + order, len(out))):
        out[i] = clip(out[i] + out[i - 1], nclip)
    for i in range(1 + order, len(out)):
        r = out[i]
        past0 = out[i - order - 1]
        window = out[i - order:i]  # oldest first; coefficient k pairs with window[order - 1 - k]
        acc = 0
        for k in range(order):
            acc = i32(acc + i32(co[k] * (window[order - 1 - k] - past0)))
        val = i32(acc + ((1 << shift) >> 1)) >> shift
        out[i] = clip(out[i] + past0 + val, nclip)
        if r > 0:
            for j in range(order):  # j walks the coefficients from the last to the first, the window oldest first
                v = past0 - window[j]
                s = sgn(v)
                co[order - 1 - j] -= s
                r -= (1 + j) * ((s * v) >> shift)
                if r <= 0:
                    break
        elif r < 0:
            for j in range(order):
                v = past0 - window[j]
                s = sgn(v)
                co[order - 1 - j] += s
                r -= (1 + j) * ((-s * v) >> shift)
                if r >= 0:
                    break
    return out, co


def test_alac_fixture_group_from_python_integers():
    f = np.load(FIX / "alac.npz")
    n = 0
    while "predict_%d_in" % n in f:
        mode, order, shift, bps = ints(f["predict_%d_params" % n])
        out, co = alac_predict(mode, order, shift, bps, ints(f["predict_%d_coeffs" % n]), ints(f["predict_%d_in" % n]))
        assert out == ints(f["predict_%d_out" % n]), (n, mode, order, shift, bps)
        assert co == ints(f["predict_%d_coeffs_after" % n]), n
        n += 1
    assert n == 11
    n = 0
    while "ms_%d_in0" % n in f:
        w, s = ints(f["ms_%d_weight_shift" % n])
        o0, o1 = [], []
        for x, y in zip(ints(f["ms_%d_in0" % n]), ints(f["ms_%d_in1" % n])):
            left = x + y - ((y * w) >> s)
            o0.append(left)
            o1.append(left - y)
        assert o0 == ints(f["ms_%d_out0" % n]) and o1 == ints(f["ms_%d_out1" % n]), n
        n += 1
    assert n == 5
