"""TEST ONLY.  A syntax-level AAC-LC raw_data_block writer for tests/test_aac_packets.py: the inverse of the reference's parse stage
(symphonia-codec-aac/src/aac/mod.rs:125-212 elements, cpe.rs:55-107 channel pair, ics/mod.rs:113-165 ics_info, 228-273 section data,
305-357 scale factors, 365-409 + 480-622 spectral data, ics/pulse.rs:41-60, ics/tns.rs:39-147).  It is not an encoder: quantised spectra,
scale factors, sections, window sequences and tool data are drawn at random inside what the syntax allows, so that every codebook,
both stereo tools, pulse data, TNS, PNS, grouping and all four window sequences occur.  The Huffman tables and the scale-factor-band
offsets are READ from the reference's source text at run time (the tests that use this are `localref`), not kept in the repository."""
import re
from pathlib import Path

import numpy as np

from flac_writer import BitWriter

ONLY_LONG, LONG_START, EIGHT_SHORT, LONG_STOP = 0, 1, 2, 3
NOISE_HCB, INTENSITY_HCB2, INTENSITY_HCB = 13, 14, 15
LAV = {1: 1, 2: 1, 3: 2, 4: 2, 5: 4, 6: 4, 7: 7, 8: 7, 9: 12, 10: 12, 11: 16}


class Tables:
    def __init__(self, ref_root):
        src = (Path(ref_root) / "symphonia-codec-aac/src/aac/codebooks.rs").read_text()
        com = (Path(ref_root) / "symphonia-codec-aac/src/aac/common.rs").read_text()

        def arr(text, name):
            m = re.search(r"\b%s: \[\w+; [\d +]+\] =\s*\[(.*?)\];" % name, text, re.S)
            assert m, name
            return [int(x, 0) for x in re.findall(r"0x[0-9a-fA-F]+|\d+", m.group(1))]

        self.spec = {k: (arr(src, "SPECTRUM_CODEBOOK%d_CODES" % k), arr(src, "SPECTRUM_CODEBOOK%d_LENS" % k)) for k in range(1, 12)}
        self.scf = (arr(src, "SCF_CODEBOOK_CODES"), arr(src, "SCF_CODEBOOK_LENS"))
        self.long_bands = arr(com, "SWB_OFFSET_48K_LONG")    # 44.1 kHz shares the 48 kHz tables (common.rs:145-149)
        self.short_bands = arr(com, "SWB_OFFSET_48K_SHORT")
        assert len(self.scf[0]) == 121 and self.long_bands[-1] == 1024 and self.short_bands[-1] == 128


def next_window_sequence(rng, prev):
    if prev in (ONLY_LONG, LONG_STOP):
        return LONG_START if rng.random() < 0.35 else ONLY_LONG
    return EIGHT_SHORT if rng.random() < 0.5 else LONG_STOP


class IcsInfo:
    def __init__(self, rng, T, prev_seq):
        self.seq = next_window_sequence(rng, prev_seq)
        self.shape = int(rng.integers(0, 2))
        self.long = self.seq != EIGHT_SHORT
        if self.long:
            self.max_sfb = int(rng.integers(0, len(T.long_bands)))  # 0 ..= num_swb
            self.grouping, self.groups = [], [[0]]
            self.bands = T.long_bands
        else:
            self.max_sfb = int(rng.integers(0, len(T.short_bands)))
            self.grouping = [int(rng.random() < 0.6) for _ in range(7)]
            self.groups = [[0]]
            for w in range(1, 8):
                if self.grouping[w - 1]:
                    self.groups[-1].append(w)
                else:
                    self.groups.append([w])
            self.bands = T.short_bands

    def write(self, bw):
        bw.put(0, 1)
        bw.put(self.seq, 2)
        bw.put(self.shape, 1)
        if self.long:
            bw.put(self.max_sfb, 6)
            bw.put(0, 1)  # predictor_data_present
        else:
            bw.put(self.max_sfb, 4)
            for g in self.grouping:
                bw.put(g, 1)


def put_code(bw, table, idx):
    bw.put(table[0][idx], table[1][idx])


def put_spectral(bw, T, cb, vals):
    """one sfb of one window: quantised integers, cb in 1..11"""
    step = 4 if cb <= 4 else 2
    assert len(vals) % step == 0
    for i in range(0, len(vals), step):
        v = [int(x) for x in vals[i:i + step]]
        if cb in (1, 2):
            put_code(bw, T.spec[cb], 27 * (v[0] + 1) + 9 * (v[1] + 1) + 3 * (v[2] + 1) + (v[3] + 1))
        elif cb in (3, 4):
            a = [abs(x) for x in v]
            put_code(bw, T.spec[cb], 27 * a[0] + 9 * a[1] + 3 * a[2] + a[3])
            for x in v:
                if x != 0:
                    bw.put(int(x < 0), 1)
        elif cb in (5, 6):
            put_code(bw, T.spec[cb], 9 * (v[0] + 4) + (v[1] + 4))
        else:
            mod = {7: 8, 8: 8, 9: 13, 10: 13, 11: 17}[cb]
            a = [min(abs(x), 16) if cb == 11 else abs(x) for x in v]
            put_code(bw, T.spec[cb], mod * a[0] + a[1])
            for x in v:
                if x != 0:
                    bw.put(int(x < 0), 1)
            if cb == 11:
                for x in v:
                    if abs(x) >= 16:
                        n = abs(x).bit_length() - 5
                        bw.put(((1 << n) - 1) << 1, n + 1)
                        bw.put(abs(x) - (1 << (n + 4)), n + 4)


def draw_values(rng, cb, count):
    lav = LAV[cb]
    if cb == 11:  # 0..15 directly, 16 and above as an escape sequence (up to 13 bits)
        v = rng.integers(0, 16, count)
        big = np.maximum(rng.integers(16, 8192, count) >> rng.integers(0, 9, count), 16)
        v = np.where(rng.random(count) < 0.08, big, v)
    else:
        v = rng.integers(0, lav + 1, count)
    sign = np.where(rng.random(count) < 0.5, -1, 1)
    return v * sign


def write_ics(bw, rng, T, info, common_window, right_of_common_pair=False, tools=True):
    """individual_channel_stream (ics/mod.rs:411-451)"""
    global_gain = int(rng.integers(110, 190))
    bw.put(global_gain, 8)
    if not common_window:
        info.write(bw)
    sect_bits = 5 if info.long else 3
    esc = (1 << sect_bits) - 1
    books = list(range(0, 12)) + [0, 0, NOISE_HCB] + ([INTENSITY_HCB, INTENSITY_HCB2, INTENSITY_HCB] if right_of_common_pair else [])
    sfb_cb = []
    for g in info.groups:
        row, k = [], 0
        while k < info.max_sfb:
            ln = int(min(info.max_sfb - k, 1 + rng.integers(0, 1 + (12 if info.long else 5))))
            if rng.random() < 0.05 and info.long:
                ln = info.max_sfb - k  # a long section: the length needs escape increments
            cb = int(rng.choice(books))
            bw.put(cb, 4)
            rest = ln
            while rest >= esc:
                bw.put(esc, sect_bits)
                rest -= esc
            bw.put(rest, sect_bits)
            row += [cb] * ln
            k += ln
        sfb_cb.append(row)
    # scale factors (ics/mod.rs:305-357): dpcm per class, the first noise energy as 9-bit pcm
    sf_normal, sf_intensity, sf_noise, noise_pcm = global_gain, 0, global_gain - 90, True
    for row in sfb_cb:
        for cb in row:
            if cb == 0:
                continue
            if cb in (INTENSITY_HCB, INTENSITY_HCB2):
                d = int(np.clip(rng.integers(-8, 9), -60 - sf_intensity, 60 - sf_intensity))  # keep the position inside +-60
                sf_intensity += d
                put_code(bw, T.scf, d + 60)
            elif cb == NOISE_HCB:
                if noise_pcm:
                    noise_pcm = False
                    d = int(rng.integers(-40, 20))
                    bw.put(d + 256, 9)
                else:
                    d = int(np.clip(rng.integers(-10, 11), 20 - sf_noise, 140 - sf_noise))
                    put_code(bw, T.scf, d + 60)
                sf_noise += d
            else:
                d = int(np.clip(rng.integers(-12, 13), 90 - sf_normal, 200 - sf_normal))
                d = int(np.clip(d, -60, 60))
                sf_normal += d
                put_code(bw, T.scf, d + 60)
    # pulse data (long windows only), TNS, no gain control
    if tools and info.long and info.max_sfb > 0 and rng.random() < 0.3:
        bw.put(1, 1)
        n = int(rng.integers(1, 5))
        bw.put(n - 1, 2)
        bw.put(int(rng.integers(0, info.max_sfb)), 6)
        for _ in range(n):
            bw.put(int(rng.integers(0, 32)), 5)
            bw.put(int(rng.integers(0, 16)), 4)
    else:
        bw.put(0, 1)
    if tools and rng.random() < 0.45:
        bw.put(1, 1)
        for w in range(1 if info.long else 8):
            n_filt = int(rng.integers(0, 4 if info.long else 2))
            bw.put(n_filt, 2 if info.long else 1)
            if n_filt:
                coef_res = int(rng.integers(0, 2))
                bw.put(coef_res, 1)
            for _ in range(n_filt):
                bw.put(int(rng.integers(0, 25 if info.long else 10)), 6 if info.long else 4)
                order = int(rng.integers(0, 13 if info.long else 8))
                bw.put(order, 5 if info.long else 3)
                if order:
                    bw.put(int(rng.integers(0, 2)), 1)
                    compress = int(rng.integers(0, 2))
                    bw.put(compress, 1)
                    bits = (4 if coef_res else 3) - compress
                    for _ in range(order):
                        bw.put(int(rng.integers(0, 1 << bits)), bits)
    else:
        bw.put(0, 1)
    bw.put(0, 1)  # gain_control_data_present
    # spectral data: per group, per band, per window of the group (ics/mod.rs:371-406)
    for g, row in zip(info.groups, sfb_cb):
        for sfb, cb in enumerate(row):
            if cb == 0 or cb >= NOISE_HCB:
                continue
            width = info.bands[sfb + 1] - info.bands[sfb]
            for _ in g:
                put_spectral(bw, T, cb, draw_values(rng, cb, width))
    return sfb_cb


class Stream:
    """raw_data_blocks of one track: `layout` is a list of "sce" / "cpe" elements (mod.rs:125-212)"""

    def __init__(self, ref_root, seed, layout):
        self.T = Tables(ref_root)
        self.rng = np.random.default_rng(seed)
        self.layout = layout
        self.prev = [[ONLY_LONG, ONLY_LONG] for _ in layout]

    def packet(self, filler=False):
        """(bytes, what was written per channel: window sequence, shape, max_sfb, the codebook of every band of every group)"""
        bw, rng, T = BitWriter(), self.rng, self.T
        meta = []

        def note(info, sfb_cb):
            meta.append({"seq": info.seq, "shape": info.shape, "max_sfb": info.max_sfb, "sfb_cb": sfb_cb})
        for e, el in enumerate(self.layout):
            if filler and e == 0:
                bw.put(6, 3)  # ID_FIL: 3 bytes of EXT_FILL
                bw.put(3, 4)
                bw.put(0, 4)
                bw.put(0, 4)
                bw.put(0xA5A5, 16)
                bw.put(4, 3)  # ID_DSE, byte-aligned, 2 bytes
                bw.put(1, 4)
                bw.put(1, 1)
                bw.put(2, 8)
                bw.align()
                bw.put(0xBEEF, 16)
            if el == "sce":
                bw.put(0, 3)
                bw.put(int(rng.integers(0, 16)), 4)
                info = IcsInfo(rng, T, self.prev[e][0])
                self.prev[e][0] = info.seq
                note(info, write_ics(bw, rng, T, info, False))
            else:
                bw.put(1, 3)
                bw.put(int(rng.integers(0, 16)), 4)
                common = int(rng.random() < 0.7)
                bw.put(common, 1)
                if common:
                    info = IcsInfo(rng, T, self.prev[e][0])
                    self.prev[e] = [info.seq, info.seq]
                    info.write(bw)
                    ms = int(rng.integers(0, 3))
                    bw.put(ms, 2)
                    if ms == 1:
                        for _ in info.groups:
                            for _ in range(info.max_sfb):
                                bw.put(int(rng.integers(0, 2)), 1)
                    note(info, write_ics(bw, rng, T, info, True))
                    note(info, write_ics(bw, rng, T, info, True, right_of_common_pair=True))
                else:
                    for c in range(2):
                        info = IcsInfo(rng, T, self.prev[e][c])
                        self.prev[e][c] = info.seq
                        note(info, write_ics(bw, rng, T, info, False))
        bw.put(7, 3)
        bw.align()
        return bw.bytes(), meta
