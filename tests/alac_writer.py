"""TEST ONLY.  A small Apple Lossless packet writer for tests/test_alac_packets.py: the inverse of the reference's parse stage
(symphonia-codec-alac/src/lib.rs: element header 471-505, channel header 83-110, the adaptive Rice code 112-163 and 606-656, the
dynamic predictor 165-264 run forwards, mid-side 664-671 inverted) -- compressed and uncompressed elements, SCE and CPE, both
predictor modes, separately coded low bits, partial frames.  The writer is lossless, so the reference's decoder giving the PCM back
validates it (the same check tests/flac_writer.py gets)."""
import numpy as np

from flac_writer import BitWriter

PB, MB, KB = 40, 10, 14  # the encoder defaults every ALAC file carries in its cookie


def cookie(frame_length, depth, nch, rate=44100):
    import struct
    return struct.pack(">IBBBBBBHIII", frame_length, 0, depth, PB, MB, KB, nch, 255, 0, 0, rate)


def sext(v, bits):
    v &= (1 << bits) - 1
    return v - (1 << bits) if v >> (bits - 1) else v


def sign(v):
    return (v > 0) - (v < 0)


def residuals(samples, mode, order, qshift, bps, coeffs):
    """what the decoder's predict() must be given to restore `samples` (lib.rs:165-264 solved for its input)"""
    s = [int(v) for v in samples]
    n = len(s)
    co = [int(c) for c in coeffs[:order]]
    e = list(s)
    if order > 0 and n > 0:
        for i in range(1, min(1 + order, n)):
            e[i] = sext(s[i] - s[i - 1], bps)
        for i in range(1 + order, n):
            past0 = s[i - order - 1]
            hist = s[i - order:i]
            acc = sum(c * (h - past0) for c, h in zip(reversed(co), hist))  # (the decoder sums in wrapping 32-bit arithmetic)
            val = sext(acc + ((1 << qshift) >> 1), 32) >> qshift
            res = e[i] = sext(s[i] - past0 - val, bps)
            assert sext(res + past0 + val, bps) == s[i]
            if res != 0:
                sg = 1 if res > 0 else -1
                for j in range(order):
                    v = past0 - hist[j]
                    sn = sign(v)
                    co[order - 1 - j] -= sg * sn
                    res -= (1 + j) * ((sg * sn * v) >> qshift)
                    if sg * res <= 0:
                        break
        if order == 31 or mode == 15:  # the decoder integrates once more in front (lib.rs:187-191)
            e = [e[0]] + [sext(e[i] - e[i - 1], bps) for i in range(1, n)]
    return e


def put_rice(bw, value, k, bps):
    m = (1 << k) - 1
    prefix, rem = divmod(value, m)
    if prefix > 8:
        bw.put(0x1FF, 9)
        bw.put(value, bps)
        return
    bw.put(((1 << prefix) - 1) << 1, prefix + 1)
    if k > 1:
        if rem == 0:
            bw.put(0, k - 1)
        else:
            bw.put(rem + 1, k)


def lg3a(v):
    return ((v >> 9) + 3).bit_length() - 1


def put_residuals(bw, res, bps, pb_factor):
    mb, toggle, i, n = MB, 0, 0, len(res)
    while i < n:
        r = res[i]
        val = 2 * r if r >= 0 else -2 * r - 1
        assert val >= toggle and val < (1 << bps)
        put_rice(bw, val - toggle, min(lg3a(mb), KB), bps)
        mb = 0xFFFF if val > 0xFFFF else (mb + pb_factor * val - (((pb_factor * mb) & 0xFFFFFFFF) >> 9)) & 0xFFFFFFFF
        toggle = 0
        i += 1
        if mb < 128 and i < n:
            k = (32 - mb.bit_length()) - 24 + ((mb + 16) >> 6)
            zeros = 0
            while i + zeros < n and res[i + zeros] == 0:
                zeros += 1
            assert zeros < 0xFFFF
            put_rice(bw, zeros, min(k, KB), 16)
            toggle, mb = 1, 0
            i += zeros


def put_element(bw, planes, depth, frame_length, kind, rng):
    """one SCE (one plane) or CPE (two); kind in {"lpc", "lpc15", "verbatim", "raw"}; returns nothing, writes the element"""
    is_cpe = len(planes) == 2
    n = len(planes[0])
    bw.put(1 if is_cpe else 0, 3)
    bw.put(int(rng.integers(0, 16)), 4)
    bw.put(0, 12)
    partial = n != frame_length
    tail_bytes = int(rng.integers(0, 3)) if (depth > 16 and kind != "raw") else 0
    tail = 8 * tail_bytes
    bw.put(int(partial), 1)
    bw.put(tail_bytes, 2)
    bw.put(int(kind == "raw"), 1)
    if partial:
        bw.put(n, 32)
    if kind == "raw":
        for i in range(n):
            for p in planes:
                bw.put_signed(int(p[i]), depth)
        return
    bps = depth - tail + int(is_cpe)
    low = [[int(v) & ((1 << tail) - 1) for v in p] for p in planes]
    hi = [[int(v) >> tail for v in p] for p in planes]
    ms_shift, ms_weight = 0, 0
    if is_cpe and rng.random() < 0.7:
        ms_shift = int(rng.integers(1, 4))
        ms_weight = int(rng.integers(1, (1 << ms_shift) + 1))  # a weight of at most one keeps the mid channel inside bps bits
        # lib.rs:664-671 inverted: the difference channel, then the weighted sum that gives the left channel back
        diff = [a - b for a, b in zip(hi[0], hi[1])]
        mid = [b + ((d * ms_weight) >> ms_shift) for b, d in zip(hi[1], diff)]
        hi = [mid, diff]
    bw.put(ms_shift, 8)
    bw.put_signed(ms_weight, 8)
    chans = []
    for p in hi:
        if kind == "verbatim":
            mode, order, qshift, co = 0, 0, 0, []
        else:
            mode = 15 if kind == "lpc15" else 0
            order = int(rng.choice([1, 4, 8, 12, 31]))
            qshift = 9
            co = [0] * order
            co[-1] = int(rng.integers(300, 500))  # roughly "repeat the previous sample"; the adaptation does the rest
            if order > 1:
                co[-2] = -int(rng.integers(0, 120))
        pb3 = int(rng.integers(2, 5))
        bw.put(mode, 4)
        bw.put(qshift, 4)
        bw.put(pb3, 3)
        bw.put(order, 5)
        for c in co:
            bw.put_signed(c, 16)
        chans.append((residuals(p, mode, order, qshift, bps, co), (pb3 * PB) >> 2))
    if tail:
        for i in range(n):
            for lw in low:
                bw.put(lw[i], tail)
    for res, pbf in chans:
        put_residuals(bw, res, bps, pbf)


def packet(planes, depth, frame_length, kinds, rng, fill=False):
    """`planes` [nch][n] in ALAC element order; kinds: one entry per element ("sce:<kind>" or "cpe:<kind>")"""
    bw = BitWriter()
    c = 0
    for k in kinds:
        el, kind = k.split(":")
        take = 2 if el == "cpe" else 1
        if fill:  # a fill and a data-stream element in front: the decoder skips them (lib.rs:364-389)
            bw.put(6, 3)
            bw.put(3, 4)
            bw.put(0xABCDEF, 24)
            bw.put(4, 3)
            bw.put(0, 4)
            bw.put(1, 1)
            bw.put(2, 8)
            bw.align()
            bw.put(0x1234, 16)
        put_element(bw, planes[c:c + take], depth, frame_length, kind, rng)
        c += take
    bw.put(7, 3)
    bw.align()
    return bw.bytes()


def smooth_pcm(rng, nch, n, depth):
    t = np.arange(n)
    amp = (1 << (depth - 2)) * 0.8
    out = []
    for c in range(nch):
        f = rng.uniform(0.003, 0.05)
        sig = amp * np.sin(2 * np.pi * f * t + rng.uniform(0, 6)) + rng.normal(0, amp / 400, n)
        if rng.random() < 0.3:
            sig[n // 3: n // 3 + 40] = 0  # digital silence: runs of zero residuals
        out.append(np.round(sig).astype(np.int64))
    return np.stack(out)
