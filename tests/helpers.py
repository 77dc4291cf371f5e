"""Shared test helpers: ULP distance, f64 closed forms, input generators."""
import json
from pathlib import Path

import numpy as np

GOLDEN = Path(__file__).resolve().parent / "golden"


def kats():
    return json.loads((GOLDEN / "ref_kats.json").read_text())


def bits_to_f32(bits):
    return np.array(bits, dtype=np.uint32).view(np.float32)


def ulp_diff(a, b):
    """Element-wise ULP distance between two f32 arrays; +0 == -0; NaN == NaN."""
    a = np.ascontiguousarray(a, dtype=np.float32).ravel()
    b = np.ascontiguousarray(b, dtype=np.float32).ravel()
    ai = a.view(np.int32).astype(np.int64)
    bi = b.view(np.int32).astype(np.int64)
    ai = np.where(ai < 0, -(ai & 0x7FFFFFFF), ai)
    bi = np.where(bi < 0, -(bi & 0x7FFFFFFF), bi)
    d = np.abs(ai - bi)
    both_nan = np.isnan(a) & np.isnan(b)
    return np.where(both_nan, 0, d)


def assert_ulp(actual, expected, max_ulp, what=""):
    d = ulp_diff(actual, expected)
    worst = int(d.max()) if d.size else 0
    assert worst <= max_ulp, "%s: max ULP distance %d > %d (at flat index %d)" % (
        what, worst, max_ulp, int(d.argmax()))


def bit_equal(a, b):
    a = np.ascontiguousarray(a)
    b = np.ascontiguousarray(b)
    return a.shape == b.shape and a.dtype == b.dtype and np.array_equal(
        a.view(np.uint8), b.view(np.uint8))


def equal_mod_nan(a, b):
    """Bit-equal wherever neither side is NaN (so +-Inf, signed zeros and denormals must match exactly), NaN exactly
    where the other is NaN.  (The NaN a GPU and an x86 produce for inf - inf differ in sign and payload; WHERE a NaN
    appears is a property of the operation graph and must agree.)"""
    a = np.ascontiguousarray(a, dtype=np.float32)
    b = np.ascontiguousarray(b, dtype=np.float32)
    if a.shape != b.shape:
        return False
    na, nb = np.isnan(a), np.isnan(b)
    return bool(np.array_equal(na, nb) and np.array_equal(a.view(np.uint32)[~na], b.view(np.uint32)[~nb]))


def sprinkle_specials(x, rng, units, per_unit=3):
    """+Inf, -Inf and NaN lines in the given units (frames / granules: indices into x reshaped to [-1, x.shape[-1]]):
    the non-finite inputs a corrupt stream can hand the synthesis stage (the reference does no range check)."""
    flat = x.reshape(-1, x.shape[-1])
    vals = np.array([np.inf, -np.inf, np.nan], np.float32)
    for u in units:
        pos = rng.integers(0, x.shape[-1], size=per_unit)
        flat[u, pos] = vals[np.arange(per_unit) % 3]
    return x


# ---- closed forms (f64) ------------------------------------------------------

def imdct_analytical(x, scale):
    """mdct.rs:154-175: y[i] = scale * sum_j x[j] cos(pi/(4N) (2i+1+N)(2j+1))."""
    x = np.asarray(x, dtype=np.float64)
    n = x.shape[-1]
    i = np.arange(2 * n)[:, None]
    j = np.arange(n)[None, :]
    c = np.cos(np.pi / (4 * n) * ((2 * i + 1 + n) * (2 * j + 1)))
    return scale * (x @ c.T)


def dft_naive(x):
    x = np.asarray(x, dtype=np.complex128)
    n = x.size
    k = np.arange(n)
    w = np.exp(-2j * np.pi * ((k[:, None] * k[None, :]) % n) / n)
    return w @ x


def imdct36_analytical(x):
    x = np.asarray(x, dtype=np.float64)
    i = np.arange(36)[:, None]
    j = np.arange(18)[None, :]
    return (np.cos(np.pi / 72 * ((2 * i + 19) * (2 * j + 1))) * x).sum(axis=1)


def imdct12_analytical(x6):
    x6 = np.asarray(x6, dtype=np.float64)
    i = np.arange(12)[:, None]
    k = np.arange(6)[None, :]
    return (np.cos(np.pi / 24 * ((2 * i + 7) * (2 * k + 1))) * x6).sum(axis=1)


def dct32_analytical(x):
    x = np.asarray(x, dtype=np.float64)
    i = np.arange(32)[:, None]
    j = np.arange(32)[None, :]
    return (np.cos(np.pi / 32 * i * (j + 0.5)) * x).sum(axis=1)


def mdct_forward(x2n):
    """Forward MDCT (2N -> N) matching imdct_analytical's kernel (for TDAC tests)."""
    x = np.asarray(x2n, dtype=np.float64)
    n = x.shape[-1] // 2
    i = np.arange(2 * n)[None, :]
    j = np.arange(n)[:, None]
    c = np.cos(np.pi / (4 * n) * ((2 * i + 1 + n) * (2 * j + 1)))
    return x @ c.T


# ---- input generators --------------------------------------------------------

def aac_spectra(rng, shape_prefix, band_limit=672):
    """SURVEY 8d config 2 spectra: N(0,1) * 2^u, u ~ U{-8..12} per 16-line band,
    zero above band_limit, with a few denormals / signed zeros sprinkled in."""
    shape = tuple(shape_prefix) + (1024,)
    x = rng.standard_normal(shape).astype(np.float32)
    u = rng.integers(-8, 13, size=tuple(shape_prefix) + (64,))
    x *= np.repeat(np.exp2(u).astype(np.float32), 16, axis=-1)
    x[..., band_limit:] = 0.0
    flat = x.reshape(-1)
    idx = rng.integers(0, flat.size, size=max(4, flat.size // 4096))
    flat[idx[0::4]] = np.float32(1e-41)
    flat[idx[1::4]] = np.float32(-0.0)
    flat[idx[2::4]] = np.float32(-3e-39)
    return x


def aac_sequence_chain(rng, n_frames, p_switch=0.25):
    """Legal window-sequence state machine {0->0/1, 1->2/3, 2->2/3, 3->0/1} plus
    per-frame window shapes; prev_shape[t] = shape[t-1] (ics/mod.rs:119, 172-177)."""
    seq = np.zeros(n_frames, dtype=np.uint8)
    cur = 0
    for t in range(n_frames):
        if cur in (0, 3):
            nxt = 1 if rng.random() < p_switch else 0
        else:
            nxt = 2 if rng.random() < 0.5 else 3
        seq[t] = nxt
        cur = nxt
    shape = rng.integers(0, 2, size=n_frames).astype(np.uint8)
    prev = np.concatenate(([rng.integers(0, 2)], shape[:-1])).astype(np.uint8)
    return seq, shape, prev


def flac_extreme_case(seed, big_coeffs):
    """Full-range i32 samples (wrapping adds) with coefficients at the edge of the FP64-exact path (the magnitudes of a
    block's coefficients sum to at most 2^20 - 1, big_coeffs=False) or far beyond it (integer path, big_coeffs=True)."""
    rng = np.random.default_rng(seed)
    nb, blocksize = 64, 97
    buf = rng.integers(-(1 << 31), 1 << 31, (nb, blocksize)).astype(np.int32)
    buf[0, :40] = -(1 << 31)
    buf[1, :40] = (1 << 31) - 1
    kind = np.full(nb, 2, np.uint8)
    order = rng.integers(1, 33, nb).astype(np.uint8)
    order[:4] = 32
    shift = rng.integers(0, 32, nb).astype(np.uint8)
    shift[:4] = (0, 31, 19, 20)
    lim = (1 << 30) if big_coeffs else 32767
    coeffs = rng.integers(-lim, lim + 1, (nb, 32)).astype(np.int32)
    coeffs[0] = -lim
    coeffs[1] = lim
    coeffs[2] = np.where(np.arange(32) % 2 == 0, lim, -lim)
    if not big_coeffs:  # 32 * 32767 + 31 = 2^20 - 1: the largest sum of magnitudes the FP64 path takes
        coeffs[0, 5] -= 31
        coeffs[1, 9] += 31
        coeffs[2, 0] += 31
        coeffs[3] = 0
        coeffs[3, 7] = (1 << 20) - 1  # ... as one coefficient
    return buf, kind, order, shift, coeffs


def flac_carrier_case(seed, carrier):
    """64 blocks (one wavefront) of full-range i32 samples whose coefficient magnitudes sum to values around 2^16, with blocks 0..3
    exactly on the bounds an exact 16-bit integer carrier of the sum has (`dot2`: every |c| <= 32767, sum <= 65533; `dot2x2`: the
    tap pairs of even and of odd index each within 65533; `f64`: one step outside both).  Round 5 built that carrier on
    v_dot2_i32_i16 and measured it slower than the FP64 FMA path (the instruction issues at half rate on gfx950:
    profiles/r05e_*, HISTORY); the cases stay as extra coverage of the exact-sum arithmetic at awkward magnitudes."""
    rng = np.random.default_rng(seed)
    nb, blocksize = 64, 131
    buf = rng.integers(-(1 << 31), 1 << 31, (nb, blocksize)).astype(np.int32)
    buf[0, :50] = -(1 << 31)           # hi = -32768, lo' = -32768 throughout the history
    buf[1, :50] = (1 << 31) - 1        # hi = 32767, lo' = 32767
    buf[2, :50] = 0xffff               # hi = 0, lo' = 32767
    buf[3, :50] = -65536               # hi = -1, lo' = -32768
    kind = np.full(nb, 2, np.uint8)
    order = rng.integers(1, 33, nb).astype(np.uint8)
    order[:6] = 32
    shift = rng.integers(0, 32, nb).astype(np.uint8)
    shift[:6] = (0, 31, 15, 16, 1, 30)
    coeffs = np.zeros((nb, 32), np.int64)
    for b in range(nb):  # random signs, magnitudes scaled to a random total below the carrier's bound
        mag = rng.random(32) ** 3
        total = int(rng.integers(1, 65534)) if carrier == "dot2" else int(rng.integers(65534, 131067))
        if carrier == "dot2":
            m = np.floor(mag / mag.sum() * total)
        else:  # each half of the tap pairs gets its own budget
            m = np.zeros(32)
            for h in (0, 1):
                idx = [j for j in range(32) if (j >> 1) & 1 == h]
                part = mag[idx]
                m[idx] = np.floor(part / part.sum() * min(65533, total // 2 + (0 if h else total % 2)))
        m = np.minimum(m, 32767)
        coeffs[b] = (m * rng.choice([-1, 1], 32)).astype(np.int64)
    if carrier == "dot2":
        coeffs[0] = 0; coeffs[0, :2] = (32767, 32766)                      # sum 65533, all positive: the accumulators' largest value
        coeffs[1] = 0; coeffs[1, :2] = (-32767, -32766)                    # ... and smallest
        coeffs[2] = 0; coeffs[2, [0, 1, 31]] = (32767, -32765, 1)          # an odd coefficient sum (the folded constant's half)
        coeffs[3] = np.where(np.arange(32) % 2 == 0, 2047, -2047); coeffs[3, 0] = 2047 + 29  # 65533 spread over all 32 taps
    elif carrier == "dot2x2":
        coeffs[0] = 0; coeffs[0, [0, 1, 2, 3]] = (32767, 32766, 32767, 32766)        # both halves at 65533
        coeffs[1] = 0; coeffs[1, [0, 1, 2, 3]] = (-32767, -32766, -32767, -32766)
        coeffs[2] = 0; coeffs[2, [4, 5, 6, 7]] = (32767, -32766, -32767, 32765)
        coeffs[3] = np.where(np.arange(32) % 2 == 0, 4095, -4095); coeffs[3, [0, 2]] += 13  # 16 * 4095 + 13 = 65533 per half
    else:  # "f64": one step outside the dot2 carriers
        coeffs[0] = 0; coeffs[0, 0] = 32768                                 # does not fit a signed half
        coeffs[1] = 0; coeffs[1, [0, 1]] = (32767, 32767)                   # a half at 65534
        coeffs[2] = 0; coeffs[2, 3] = -32768
        coeffs[3] = 0; coeffs[3, [2, 3, 6, 7]] = (32767, 32767, 1, 0)       # the odd-index half at 65535
    return buf, kind, order, shift, coeffs.astype(np.int32)


def floor1_case(rng):
    """A random floor-1 configuration and y rows: (x_list, multiplier, n, ys).  Post positions are spread or clustered
    (runs of adjacent x: one-line segments), y values up to 255 whatever the range (what a non-conforming but accepted
    stream can carry, symaccel_vorbis_floor1_status_device), zero densities from almost none to almost all."""
    import numpy as np
    n = int(rng.choice([32, 64, 128, 256, 512, 1024, 2048, 4096]))
    n_posts = int(rng.integers(2, min(65, n) + 1))
    mult = int(rng.integers(1, 5))
    count = int(rng.choice([1, 3, 17, 64, 65, 130]))
    if rng.random() < 0.4:
        base = int(rng.integers(1, max(2, n - n_posts)))
        rest = rng.permutation(np.arange(base, min(n, base + n_posts + 5)))[:n_posts - 2].tolist()
        while len(rest) < n_posts - 2:
            x = int(rng.integers(1, n))
            if x not in rest:
                rest.append(x)
    else:
        rest = rng.permutation(np.arange(1, n))[:n_posts - 2].tolist()
    xs = [0, n] + rest
    if rng.random() < 0.25:
        # a floor whose range reaches past the block (rangebits up to 15 are legal, floor.rs:494-497): posts beyond n and
        # neighbour spans above 4096, where render_point runs in integers (vorbis.hip `wide`) and render_line's segments
        # are longer than the block
        top_x = 1 << int(rng.integers(max(6, int(np.log2(n)) + 1), 16))
        inside = sorted(set(int(v) for v in rng.integers(1, n, size=int(rng.integers(0, n_posts - 1)))))
        rest = inside[:n_posts - 2]
        while len(rest) < n_posts - 2:
            x = int(rng.integers(1, top_x))
            if x not in rest:
                rest.append(x)
        xs = [0, top_x] + [int(v) for v in rng.permutation(rest)]
    # (values beyond the range are legal codebook entries: up to 511 the kernel reproduces the reference's i32 arithmetic)
    top = 512 if rng.random() < 0.2 else (256 if rng.random() < 0.3 else [256, 128, 86, 64][mult - 1])
    ys = rng.integers(0, top, size=(count, n_posts)).astype(np.uint32)
    ys[rng.random((count, n_posts)) < rng.choice([0.02, 0.3, 0.9])] = 0
    return xs, mult, n, ys
