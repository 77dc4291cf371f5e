#!/usr/bin/env python3
"""Design-validation model of the wave-per-frame AAC long-block kernel ("aac_synth").

Emulates the 64 lanes of one wavefront with numpy float32 arithmetic (IEEE, no FMA), using the
exact lane <-> data mapping, twiddle forms and LDS transposes the HIP kernel uses, and checks
the result bit-for-bit against the CPU oracle.  Also feeds the LDS address functions to
tests/models/lds_sim.py to count bank-conflict cycles of the chosen layout.

Test infrastructure (tests/test_models.py runs it); not shipped, not imported by the product.
"""
import sys
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parent.parent.parent))
sys.path.insert(0, str(Path(__file__).resolve().parent))
import oracle  # noqa: E402  (test infrastructure: the oracle is the checker)
import lds_sim  # noqa: E402

F = np.float32
LANES = np.arange(64)
C = F(0.70710678118654752440)


def rev(x, bits):
    r = 0
    for b in range(bits):
        r |= ((x >> b) & 1) << (bits - 1 - b)
    return r


REV3 = np.array([rev(i, 3) for i in range(8)])


def cmul(wr, wi, xr, xi):
    return wr * xr - wi * xi, wr * xi + wi * xr


# --------------------------------------------------------------------------- in-register pieces

def fft8_regs(u):
    """u: [lanes, 8] complex as (re, im) float32 arrays, bit-reversed input order (no_simd.rs:405-454)."""
    re, im = u
    re, im = re.copy(), im.copy()

    def bf(a, b, form):
        xr, xi = re[:, b], im[:, b]
        if form == 0:
            qr, qi = xr, xi
        elif form == 1:  # -i
            qr, qi = xi, -xr
        elif form == 2:  # (1-i)/sqrt2
            aa, bb = C * xr, C * xi
            qr, qi = aa + bb, bb - aa
        else:  # (-1-i)/sqrt2
            aa, bb = -C * xr, -C * xi
            qr, qi = aa - bb, aa + bb
        er, ei = re[:, a].copy(), im[:, a].copy()
        re[:, a], im[:, a] = er + qr, ei + qi
        re[:, b], im[:, b] = er - qr, ei - qi

    for a in (0, 2, 4, 6):  # fft2
        bf(a, a + 1, 0)
    for base in (0, 4):  # fft4 combine: k=0 id, k=1 -i
        bf(base, base + 2, 0)
        bf(base + 1, base + 3, 1)
    bf(0, 4, 0)  # fft8 combine: k=0 id, 1 -> form2, 2 -> -i, 3 -> form3
    bf(1, 5, 2)
    bf(2, 6, 1)
    bf(3, 7, 3)
    return re, im


def twiddled(xr, xi, wr, wi, form):
    """Per-lane twiddle of a small-fft combine with lane-dependent form (0 general, 1 id, 2 -i)."""
    gr, gi = cmul(wr, wi, xr, xi)
    qr = np.where(form == 1, xr, np.where(form == 2, xi, gr))
    qi = np.where(form == 1, xi, np.where(form == 2, -xr, gi))
    return qr, qi


def small_tw(n, k):
    """Exact constants of the fft16/fft32 combine for element k, as (wr, wi, form).
    k == n/8 and 3n/8 use w = (+-c, -c) with the general formula (bit-identical to the
    reference's (a+b, b-a) / (a-b, a+b) forms)."""
    wr = np.empty(k.shape, F)
    wi = np.empty(k.shape, F)
    form = np.zeros(k.shape, np.int32)
    for idx, kk in np.ndenumerate(k):
        if kk == 0:
            wr[idx], wi[idx], form[idx] = 1, 0, 1
        elif 4 * kk == n:
            wr[idx], wi[idx], form[idx] = 0, -1, 2
        elif 8 * kk == n:
            wr[idx], wi[idx] = C, -C
        elif 8 * kk == 3 * n:
            wr[idx], wi[idx] = -C, -C
        else:
            wr[idx] = F(np.cos(2.0 * np.pi * kk / n))
            wi[idx] = F(-np.sin(2.0 * np.pi * kk / n))
    return wr, wi, form


def butterfly(re, im, a, b, qr, qi):
    er, ei = re[:, a].copy(), im[:, a].copy()
    re[:, a], im[:, a] = er + qr, ei + qi
    re[:, b], im[:, b] = er - qr, ei - qi


# --------------------------------------------------------------------------- LDS layouts

def lds_T1(B, j, k):
    """complex index of element (B, j, k) for the pass1 -> pass2 transpose: separable
    (lane base + per-instruction immediate on both sides) and conflict-free."""
    return B + 8 * (j >> 2) + 36 * k + 288 * (j & 3)


def lds_T2(B, j, k):
    """pass2 -> pass3: writer lane (B,k) holds j=0..7; reader lane k'=8j+k holds B=0..7."""
    return lds_T2_impl(B, j, k)


def lds_T2_impl(B, j, k):
    return k + 8 * ((B & 1) + j) + 64 * (B >> 1) + 256 * (B & 1)


class Lds:
    def __init__(self, n):
        self.re = np.zeros(n, F)
        self.im = np.zeros(n, F)


def run_frame(spec, tw_pre, W16, W32, W64, W128, W256, W512, conflict_log=None):
    """One 1024-line long block -> pcm[2048] (raw IMDCT output), wave-emulated."""
    m = LANES
    # ---- load + pre-twiddle: lane m, s = 0..7 -> z[m + 64 s]
    zr = np.empty((64, 8), F)
    zi = np.empty((64, 8), F)
    for s in range(8):
        i = m + 64 * s
        even = spec[2 * i]
        odd = -spec[1023 - 2 * i]
        wr, wi = tw_pre[i].real.astype(F), tw_pre[i].imag.astype(F)
        zr[:, s] = odd * wi - even * wr
        zi[:, s] = odd * wr + even * wi
    # ---- pass 1: fft8 on u[r] = z[m + 64 rev3(r)]
    re, im = fft8_regs((zr[:, REV3], zi[:, REV3]))
    Bw, jw = REV3[m & 7], REV3[m >> 3]
    # ---- T1 through LDS
    lds = Lds(1536)
    for r in range(8):
        a = lds_T1(Bw, jw, r)
        lds.re[a], lds.im[a] = re[:, r], im[:, r]
        if conflict_log is not None:
            conflict_log("T1 write", "write_b64", a * 8, r)
    B2, k2 = m >> 3, m & 7
    re = np.empty((64, 8), F)
    im = np.empty((64, 8), F)
    for j in range(8):
        a = lds_T1(B2, j, k2)
        re[:, j], im[:, j] = lds.re[a], lds.im[a]
        if conflict_log is not None:
            conflict_log("T1 read", "read_b64", a * 8, j)
    # ---- pass 2: stages 4, 5, 6 (u[j] = a[64B + 8j + k])
    wr, wi, form = small_tw(16, k2)
    for j in (0, 2, 4, 6):
        qr, qi = twiddled(re[:, j + 1], im[:, j + 1], wr, wi, form)
        butterfly(re, im, j, j + 1, qr, qi)
    for jj in (0, 1):
        wr, wi, form = small_tw(32, 8 * jj + k2)
        for h in (0, 4):
            qr, qi = twiddled(re[:, h + jj + 2], im[:, h + jj + 2], wr, wi, form)
            butterfly(re, im, h + jj, h + jj + 2, qr, qi)
    for j in range(4):
        w = W64[8 * j + k2]
        qr, qi = cmul(re[:, j + 4], im[:, j + 4], w.real.astype(F), w.imag.astype(F))
        butterfly(re, im, j, j + 4, qr, qi)
    # ---- T2 through LDS: position 64B + 8j + k
    for j in range(8):
        a = lds_T2(B2, j, k2)
        lds.re[a], lds.im[a] = re[:, j], im[:, j]
        if conflict_log is not None:
            conflict_log("T2 write", "write_b64", a * 8, j)
    kp = m
    re = np.empty((64, 8), F)
    im = np.empty((64, 8), F)
    for B in range(8):
        a = lds_T2(np.full(64, B), kp >> 3, kp & 7)
        re[:, B], im[:, B] = lds.re[a], lds.im[a]
        if conflict_log is not None:
            conflict_log("T2 read", "read_b64", a * 8, B)
    # ---- pass 3: stages 7, 8, 9 (v[B] = a[64B + k'])
    w = W128[kp]
    for B in (0, 2, 4, 6):
        qr, qi = cmul(re[:, B + 1], im[:, B + 1], w.real.astype(F), w.imag.astype(F))
        butterfly(re, im, B, B + 1, qr, qi)
    for b in (0, 1):
        w = W256[64 * b + kp]
        for h in (0, 4):
            qr, qi = cmul(re[:, h + b + 2], im[:, h + b + 2], w.real.astype(F), w.imag.astype(F))
            butterfly(re, im, h + b, h + b + 2, qr, qi)
    for B in range(4):
        w = W512[64 * B + kp]
        qr, qi = cmul(re[:, B + 4], im[:, B + 4], w.real.astype(F), w.imag.astype(F))
        butterfly(re, im, B, B + 4, qr, qi)
    # ---- T3: natural order Z[64B + k'] -> slots
    Z = np.empty(512, np.complex64)
    for B in range(8):
        Z[64 * B + kp] = re[:, B] + 1j * im[:, B]
        if conflict_log is not None:
            conflict_log("T3 write", "write_b64", (64 * B + kp) * 8, B)
    pcm = np.empty(2048, F)
    for half in range(2):
        m2 = m + 64 * half
        if conflict_log is not None:
            conflict_log("T3 read", "read_b128", (254 - 2 * m2) * 8, 2 * half)
            conflict_log("T3 read", "read_b128", (256 + 2 * m2) * 8, 2 * half + 1)

        def val(i):
            x, w = Z[i], tw_pre[i]
            return cmul(w.real.astype(F), w.imag.astype(F), x.real.astype(F), -x.imag.astype(F))

        vA, vB, vC, vD = val(255 - 2 * m2), val(254 - 2 * m2), val(256 + 2 * m2), val(257 + 2 * m2)
        for q, v in enumerate((-vC[0], -vA[1], -vD[0], -vB[1])):
            pcm[4 * m2 + q] = v
        for q, v in enumerate((vB[1], vD[0], vA[1], vC[0])):
            pcm[512 + 508 - 4 * m2 + q] = v
        for q, v in enumerate((vC[1], vA[0], vD[1], vB[0])):
            pcm[1024 + 4 * m2 + q] = v
        for q, v in enumerate((vB[0], vD[1], vA[0], vC[1])):
            pcm[1536 + 508 - 4 * m2 + q] = v
    return pcm


def main():
    rng = np.random.default_rng(0)
    tw_pre = oracle.imdct_twiddles(1024, 1.0 / 2048.0)
    W = {n: oracle.fft_twiddles(n) for n in (64, 128, 256, 512)}
    W16, W32 = oracle.fft_small_twiddles(16), oracle.fft_small_twiddles(32)
    totals = {}

    def log(name, instr, byte_addrs, i):
        c = lds_sim.cycles(instr, list(byte_addrs))
        key = (name, instr)
        t = totals.setdefault(key, [0, 0])
        t[0] += c
        t[1] += lds_sim.ideal(instr)

    ok = True
    for trial in range(3):
        spec = (rng.standard_normal(1024) * np.exp2(rng.integers(-8, 12, 1024))).astype(F)
        got = run_frame(spec, tw_pre, W16, W32, W[64], W[128], W[256], W[512],
                        conflict_log=log if trial == 0 else None)
        want = oracle.imdct(spec, 1.0 / 2048.0)
        same = np.array_equal(got.view(np.uint32), want.view(np.uint32))
        ok &= same
        print("trial", trial, "bit-exact vs oracle:", same)
    for (name, instr), (c, ideal) in totals.items():
        print("%-10s %-10s LDS cycles %3d (ideal %3d)" % (name, instr, c, ideal))
    return ok, totals


if __name__ == "__main__":
    sys.exit(0 if main()[0] else 1)
