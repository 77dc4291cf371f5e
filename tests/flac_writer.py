"""TEST ONLY.  A small FLAC frame writer (RFC 9639 / the format symphonia-bundle-flac/src/{frame,decoder}.rs read): frame header with
its CRC-8, subframes of every type (constant, verbatim, fixed order 0..4, LPC order 1..32 with given quantised coefficients), wasted
bits, Rice / Rice2 partitions including escaped (binary) partitions, every channel assignment, frame CRC-16.  Produces the packet bytes
a FLAC demuxer hands to `AudioDecoder::decode_ref`; used to drive the reference's decoder (executed by tools/rsinterp) and the
accelerated decoder with the same bytes (tests/test_flac_packets.py)."""
import numpy as np


class BitWriter:
    def __init__(self):
        self.acc, self.n, self.out = 0, 0, bytearray()

    def put(self, value, bits):
        if bits == 0:
            return
        assert 0 <= value < (1 << bits), (value, bits)
        self.acc = (self.acc << bits) | value
        self.n += bits
        while self.n >= 8:
            self.n -= 8
            self.out.append((self.acc >> self.n) & 0xFF)
        self.acc &= (1 << self.n) - 1

    def put_signed(self, value, bits):
        self.put(value & ((1 << bits) - 1), bits)

    def unary_zeros(self, q):  # q zeros then a one
        while q >= 32:
            self.put(0, 32)
            q -= 32
        self.put(1, q + 1)

    def align(self):
        if self.n:
            self.put(0, 8 - self.n)

    def bytes(self):
        assert self.n == 0
        return bytes(self.out)


def crc8(data):
    c = 0
    for b in data:
        c ^= b
        for _ in range(8):
            c = ((c << 1) ^ 0x07) & 0xFF if c & 0x80 else (c << 1) & 0xFF
    return c


def crc16(data):
    c = 0
    for b in data:
        c ^= b << 8
        for _ in range(8):
            c = ((c << 1) ^ 0x8005) & 0xFFFF if c & 0x8000 else (c << 1) & 0xFFFF
    return c


def utf8_number(v):
    """the "extended UTF-8" coding of the frame / sample number (frame.rs utf8_decode_be_u64)"""
    if v < 0x80:
        return bytes([v])
    n = 1  # continuation bytes
    while v >= (1 << (6 * n + (6 - n))):
        n += 1
    lead = (0xFF << (7 - n)) & 0xFF
    out = [lead | (v >> (6 * n))]
    for k in range(n - 1, -1, -1):
        out.append(0x80 | ((v >> (6 * k)) & 0x3F))
    return bytes(out)


BPS_CODE = {8: 1, 12: 2, 16: 4, 20: 5, 24: 6, 32: 7}
ASSIGN_CODE = {"left_side": 8, "right_side": 9, "mid_side": 10}
FIXED_COEFFS = {0: [], 1: [1], 2: [2, -1], 3: [3, -3, 1], 4: [4, -6, 4, -1]}


def predict_residual(x, coeffs, shift):
    """x: python ints; residual[i] = x[i] - ((sum_j coeffs[j] * x[i-1-j]) >> shift) for i >= order, x[i] below"""
    order = len(coeffs)
    res = list(x[:order])
    for i in range(order, len(x)):
        acc = sum(coeffs[j] * x[i - 1 - j] for j in range(order))
        res.append(x[i] - (acc >> shift))
    return res


def write_residual(bw, res, order, blocksize, part_order, rice2=False, escape_partitions=()):
    bw.put(1 if rice2 else 0, 2)
    bw.put(part_order, 4)
    nparts = 1 << part_order
    plen = blocksize >> part_order
    assert plen << part_order == blocksize and order <= plen
    pbits = 5 if rice2 else 4
    pos = order
    for p in range(nparts):
        n = plen - order if p == 0 else plen
        vals = res[pos:pos + n]
        pos += n
        if p in escape_partitions:
            bw.put((1 << pbits) - 1, pbits)
            width = max([1] + [(v if v >= 0 else ~v).bit_length() + 1 for v in vals])
            bw.put(width, 5)
            for v in vals:
                bw.put_signed(v, width)
            continue
        folded = [(v << 1) if v >= 0 else ((-v << 1) - 1) for v in vals]
        mean = (sum(folded) // max(1, len(folded))) if folded else 0
        k = min(max(mean.bit_length() - 1, 0), (1 << pbits) - 2)
        bw.put(k, pbits)
        for u in folded:
            bw.unary_zeros(u >> k)
            bw.put(u & ((1 << k) - 1), k)
    assert pos == blocksize


def write_subframe(bw, x, bps, spec):
    """x: the subframe's samples (python ints, `bps` bits); spec: dict(kind='constant'|'verbatim'|'fixed'|'lpc', order, coeffs,
    shift, precision, wasted, part_order, rice2, escape)"""
    wasted = spec.get("wasted", 0)
    if wasted:
        assert all(v % (1 << wasted) == 0 for v in x)
        x = [v >> wasted for v in x]
    bps -= wasted
    kind = spec["kind"]
    order = spec.get("order", 0)
    code = {"constant": 0, "verbatim": 1, "fixed": 8 + order, "lpc": 32 + order - 1}[kind]
    bw.put(0, 1)
    bw.put(code, 6)
    if wasted:
        bw.put(1, 1)
        bw.unary_zeros(wasted - 1)
    else:
        bw.put(0, 1)
    n = len(x)
    if kind == "constant":
        assert all(v == x[0] for v in x)
        bw.put_signed(x[0], bps)
    elif kind == "verbatim":
        for v in x:
            bw.put_signed(v, bps)
    else:
        coeffs = FIXED_COEFFS[order] if kind == "fixed" else [int(c) for c in spec["coeffs"]]
        shift = 0 if kind == "fixed" else spec["shift"]
        assert len(coeffs) == order
        for v in x[:order]:
            bw.put_signed(v, bps)
        if kind == "lpc":
            prec = spec.get("precision", 15)
            bw.put(prec - 1, 4)
            bw.put_signed(shift, 5)
            for c in coeffs:
                bw.put_signed(c, prec)
        res = predict_residual(x, coeffs, shift)
        write_residual(bw, res, order, n, spec.get("part_order", 0), spec.get("rice2", False), spec.get("escape", ()))


def write_frame(channels, bps, frame_number, assignment="independent", specs=None, sample_rate_code=9, streaminfo_bps=False):
    """channels: list of per-channel sample lists (left/right/... PCM, python ints in `bps` bits).  assignment: 'independent',
    'left_side', 'right_side', 'mid_side' (stereo only: the side channel gets bps + 1 bits).  specs: per SUBFRAME dicts (see
    write_subframe).  Returns the frame's bytes."""
    nch, n = len(channels), len(channels[0])
    planes = [[int(v) for v in ch] for ch in channels]
    sub_bps = [bps] * nch
    if assignment != "independent":
        left, right = planes
        side = [a - b for a, b in zip(left, right)]
        if assignment == "left_side":
            planes, sub_bps = [left, side], [bps, bps + 1]
        elif assignment == "right_side":
            planes, sub_bps = [side, right], [bps + 1, bps]
        else:
            planes, sub_bps = [[(a + b) >> 1 for a, b in zip(left, right)], side], [bps, bps + 1]
    specs = specs or [dict(kind="verbatim")] * nch
    hdr = bytearray([0xFF, 0xF8])  # sync, fixed block size strategy
    table = {192: 1, 576: 2, 1152: 3, 2304: 4, 4608: 5, 256: 8, 512: 9, 1024: 10, 2048: 11, 4096: 12, 8192: 13, 16384: 14, 32768: 15}
    bs_code = table.get(n, 6 if n <= 256 else 7)
    hdr.append((bs_code << 4) | sample_rate_code)
    ch_code = nch - 1 if assignment == "independent" else ASSIGN_CODE[assignment]
    hdr.append((ch_code << 4) | ((0 if streaminfo_bps else BPS_CODE[bps]) << 1))
    hdr += utf8_number(frame_number)
    if bs_code == 6:
        hdr.append(n - 1)
    elif bs_code == 7:
        hdr += bytes([(n - 1) >> 8, (n - 1) & 0xFF])
    hdr.append(crc8(hdr))
    bw = BitWriter()
    for x, b, spec in zip(planes, sub_bps, specs):
        write_subframe(bw, x, b, spec)
    bw.align()
    body = bytes(hdr) + bw.bytes()
    c = crc16(body)
    return body + bytes([c >> 8, c & 0xFF])


def random_stream(seed, n_frames, nch=2, bps=16, blocksize=192):
    """A reproducible little stream: every subframe type, wasted bits, escaped partitions, all channel assignments.
    Returns (frames: list of bytes, pcm: int64 array [n_frames][nch][blocksize])."""
    rng = np.random.default_rng(seed)
    frames, pcm = [], []
    t = 0
    for f in range(n_frames):
        tt = np.arange(t, t + blocksize)
        t += blocksize
        amp = (1 << (bps - 2))
        chans = []
        for c in range(nch):
            x = amp * 0.6 * np.sin(tt * (0.02 + 0.013 * c) + c) + amp * 0.2 * np.sin(tt * 0.31 + 2 * c) + rng.standard_normal(blocksize) * amp * 0.01
            chans.append(np.clip(np.round(x), -(1 << (bps - 1)), (1 << (bps - 1)) - 1).astype(np.int64))
        assignment = ["independent", "left_side", "mid_side", "right_side"][f % 4] if nch == 2 else "independent"
        specs = []
        for c in range(nch):
            pick = (f + 3 * c) % 7
            if pick == 0:
                spec = dict(kind="verbatim")
            elif pick == 1 and assignment == "independent":
                chans[c][:] = chans[c][0]
                spec = dict(kind="constant")
            elif pick in (1, 2, 3):
                spec = dict(kind="fixed", order=int(rng.integers(0, 5)), part_order=int(rng.integers(0, 3)))
            else:
                order = int(rng.choice([1, 2, 4, 6, 8, 12, 13, 32][: 8 if blocksize >= 128 else 5]))
                shift = int(rng.integers(8, 14))
                co = rng.standard_normal(order) * (0.6 ** np.arange(order))
                co[0] += 1.2
                coeffs = np.clip(np.round(co * (1 << shift)), -(1 << 14), (1 << 14) - 1).astype(np.int64)
                spec = dict(kind="lpc", order=order, coeffs=[int(v) for v in coeffs], shift=shift, precision=15,
                            part_order=int(rng.integers(0, 3)) if order <= blocksize // 4 else 0, rice2=bool(f % 2))
            if pick == 5:
                spec["escape"] = (0,)
            if pick == 6 and assignment == "independent":
                w = 2
                chans[c] = (chans[c] >> w) << w
                spec["wasted"] = w
            specs.append(spec)
        if assignment != "independent":
            for s in specs:
                s.pop("wasted", None)
        frames.append(write_frame([list(map(int, ch)) for ch in chans], bps, f, assignment, specs))
        pcm.append(np.stack(chans))
    return frames, np.stack(pcm)
