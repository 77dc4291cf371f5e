#!/usr/bin/env python3
"""LDS cycles of one group of vorbis_synth_wave2_kernel per block size (tests/models/lds_sim.py rules): which accesses conflict.
    python tools/vorbis_lds_model.py"""
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent / "tests" / "models"))
from lds_sim import cycles, ideal  # noqa: E402


def rev_bits(x, n):
    r = 0
    for i in range(n):
        r |= ((x >> i) & 1) << (n - 1 - i)
    return r


def t1_lane_w(B, j): return B + 8 * (j >> 2) + 288 * (j & 3)
def t1_inst_w(k): return 36 * k
def t1_lane_r(B, k): return B + 36 * k
def t1_inst_r(j): return 8 * (j >> 2) + 288 * (j & 3)
def t2_lane_w(B, k): return k + 8 * (B & 1) + 64 * (B >> 1) + 256 * (B & 1)
def t2_inst_w(j): return 8 * j
def t2_lane_r(j, k): return k + 8 * j
def t2_inst_r(B): return 8 * (B & 1) + 64 * (B >> 1) + 256 * (B & 1)


def group(logp, post=None, verbose=True):
    P, gbits = 1 << logp, logp - 3
    G = 1 << gbits
    rows = []

    def add(name, instr, fn, n):
        tot = sum(cycles(instr, [fn(l, i) for l in range(64)]) for i in range(n))
        rows.append((name, instr, n, tot, ideal(instr) * n))
    add("stage", "write_b128", lambda l, q: 16 * (l + 64 * q), 4)
    T = lambda l: l >> gbits
    u = lambda l: l & (G - 1)
    add("pre fwd", "read_b64", lambda l, s: 4 * ((T(l) << (logp + 1)) + 2 * u(l) + 2 * (s << gbits)), 8)
    add("pre bwd", "read_b32", lambda l, s: 4 * ((T(l) << (logp + 1)) + 2 * P - 1 - 2 * u(l) - 2 * (s << gbits)), 8)
    add("pre tw", "read_b64", lambda l, s: 8 * (u(l) + (s << gbits)), 8)
    g = lambda l: (T(l) << gbits) + rev_bits(u(l), gbits)
    add("fft w1", "write_b64", lambda l, r: 8 * (t1_lane_w(g(l) >> 3, g(l) & 7) + t1_inst_w(r)), 8)
    add("fft r1", "read_b64", lambda l, j: 8 * (t1_lane_r(l >> 3, l & 7) + t1_inst_r(j)), 8)
    if logp >= 7:
        add("fft w2", "write_b64", lambda l, j: 8 * (t2_lane_w(l >> 3, l & 7) + t2_inst_w(j)), 8)
        add("fft r2", "read_b64", lambda l, B: 8 * (t2_lane_r(l >> 3, l & 7) + t2_inst_r(B)), 8)
    # post twiddle
    n4 = P >> 1
    gq = 8 if logp <= 6 else 64
    kl = (lambda l: l & 7) if logp <= 6 else (lambda l: l)
    Tl = (lambda l: (l >> 3) << (6 - logp)) if logp <= 6 else (lambda l: 0)
    add("post tw", "read_b64", lambda l, q: 8 * (kl(l) + (((8 if logp <= 6 else 64) * q) & (P - 1))), 8)
    stores = []
    for q in range(8):
        pq = (8 if logp <= 6 else 64) * q
        kq, Tq = pq & (P - 1), pq >> logp
        base = Tq << (logp + 2)
        for v in range(4):
            lo = kq < n4
            kk = kq if lo else kq - n4
            rising = (v in (1, 3)) if lo else (v in (0, 2))
            if rising:
                stores.append(lambda l, base=base, v=v, kk=kk: 4 * ((Tl(l) << (logp + 2)) + base + v * P + 2 * (kk + kl(l))))
            else:
                stores.append(lambda l, base=base, v=v, kk=kk: 4 * ((Tl(l) << (logp + 2)) + base + v * P + P - 1 - 2 * (kk + kl(l))))
    tot = sum(cycles("write_b32", [f(l) for l in range(64)]) for f in stores)
    rows.append(("post scatter", "write_b32", 32, tot, 64))
    add("ola reads (a, y, wf, wr)", "read_b128", lambda l, i: 16 * l, 16)
    add("overlap copy r", "read_b128", lambda l, i: 16 * l, 2)
    add("overlap copy w", "write_b128", lambda l, i: 16 * l, 2)
    tot = sum(r[3] for r in rows)
    idl = sum(r[4] for r in rows)
    if verbose:
        print("P = %d  (blocks of %d samples): %d LDS cycles per group (conflict-free %d)" % (P, 4 * P, tot, idl))
        for r in rows:
            print("    %-26s %-10s x%-2d %5d  (ideal %4d)" % r)
    return tot, idl


if __name__ == "__main__":
    for logp in range(4, 10):
        group(logp)
