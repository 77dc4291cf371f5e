"""TEST ONLY.  A syntax-level MPEG-1 / MPEG-2 Layer III frame writer for tests/test_mp3_packets.py: the inverse of the reference's
parse stage (symphonia-bundle-mp3/src/header.rs:94-229 frame header, layer3/bitstream.rs:48-174 side information, 176-237 MPEG-1 and
239-343 MPEG-2 scale factors, layer3/requantize.rs:47-224 Huffman-coded samples, layer3/mod.rs:42-108 the bit reservoir).  Not an
encoder: quantised samples, tables, region splits, gains, scale factors, block types and stereo modes are drawn at random inside what
the syntax allows -- every Huffman table with and without linbits, both count1 tables, all block types incl. mixed blocks, scfsi,
mono / stereo / dual / joint stereo (mid-side and intensity), variable bit rate, main data reaching back into earlier frames.  The
Huffman tables and the scale-factor-band offsets are READ from the reference's source text at run time (the tests using this are
`localref`), not kept in the repository."""
import re
from pathlib import Path

import numpy as np

from flac_writer import BitWriter

BIT_RATES_V1 = [0, 32, 40, 48, 56, 64, 80, 96, 112, 128, 160, 192, 224, 256, 320]
BIT_RATES_V2 = [0, 8, 16, 24, 32, 40, 48, 56, 64, 80, 96, 112, 128, 144, 160]
WRAP = {1: 2, 2: 3, 3: 3, 5: 4, 6: 4, 7: 6, 8: 6, 9: 6, 10: 8, 11: 8, 12: 8, 13: 16, 15: 16}
LINBITS = [0] * 16 + [1, 2, 3, 4, 6, 8, 10, 13, 4, 5, 6, 7, 8, 9, 11, 13]
SLEN = [(0, 0), (0, 1), (0, 2), (0, 3), (3, 0), (1, 1), (1, 2), (1, 3), (2, 1), (2, 2), (2, 3), (3, 1), (3, 2), (3, 3), (4, 2), (4, 3)]
NSFB_V2 = [[[7, 7, 7, 0], [12, 12, 12, 0], [6, 15, 12, 0]], [[6, 6, 6, 3], [12, 9, 9, 6], [6, 12, 9, 6]], [[8, 8, 5, 0], [15, 12, 9, 0], [6, 18, 9, 0]],
           [[6, 5, 5, 5], [9, 9, 9, 9], [6, 9, 9, 9]], [[6, 5, 7, 3], [9, 9, 12, 6], [6, 9, 12, 6]], [[11, 10, 0, 0], [18, 18, 0, 0], [15, 18, 0, 0]]]
LONG, START, SHORT, END = 0, 1, 2, 3


class Tables:
    def __init__(self, ref_root):
        src = (Path(ref_root) / "symphonia-bundle-mp3/src/layer3/codebooks.rs").read_text()
        com = (Path(ref_root) / "symphonia-bundle-mp3/src/layer3/common.rs").read_text()

        def arr(name):
            m = re.search(r"\b%s: \[\w+; \d+\] =\s*\[(.*?)\];" % name, src, re.S)
            assert m, name
            return [int(x, 0) for x in re.findall(r"0x[0-9a-fA-F]+|\d+", re.sub(r"//[^\n]*", "", m.group(1)))]

        self.big = {t: (arr("MPEG_CODES_%d" % t), arr("MPEG_BITS_%d" % t)) for t in list(WRAP) + [16, 24]}
        self.quads = [(arr("MPEG_QUADS_CODES_" + s), arr("MPEG_QUADS_BITS_" + s)) for s in "AB"]
        m = re.search(r"SFB_LONG_BANDS: \[\[usize; 23\]; 9\] = \[(.*?)\n\];", com, re.S)
        nums = [int(x) for x in re.findall(r"\d+", re.sub(r"//[^\n]*", "", m.group(1)))]
        assert len(nums) == 9 * 23
        self.sfb_long = [nums[23 * i:23 * i + 23] for i in range(9)]
        assert all(len(self.big[t][0]) == WRAP[t] ** 2 for t in WRAP) and len(self.big[16][0]) == 256 and self.sfb_long[0][22] == 576

    def table(self, t):
        """(codes, lens, wrap, linbits, largest value) of table_select t"""
        if t >= 24:
            c, wrap = self.big[24], 16
        elif t >= 16:
            c, wrap = self.big[16], 16
        else:
            c, wrap = self.big[t], WRAP[t]
        lb = LINBITS[t]
        return c[0], c[1], wrap, lb, (wrap - 1) + ((1 << lb) - 1 if lb else 0)


USABLE = [t for t in range(1, 32) if t not in (4, 14)]


def put_pair(bw, tab, x, y):
    codes, lens, wrap, lb, _ = tab
    ax, ay = abs(x), abs(y)
    cx, cy = min(ax, 15) if lb else ax, min(ay, 15) if lb else ay
    i = cx * wrap + cy
    bw.put(codes[i], lens[i])
    n = lens[i]
    for a, c, v in ((ax, cx, x), (ay, cy, y)):
        if a:
            if c == 15 and lb:
                bw.put(a - 15, lb)
                n += lb
            bw.put(int(v < 0), 1)
            n += 1
    return n


def put_quad(bw, tab, v):
    i = (abs(v[0]) << 3) | (abs(v[1]) << 2) | (abs(v[2]) << 1) | abs(v[3])
    bw.put(tab[0][i], tab[1][i])
    n = tab[1][i]
    for x in v:
        if x:
            bw.put(int(x < 0), 1)
            n += 1
    return n


class Stream:
    def __init__(self, ref_root, seed, mode, mpeg1=True, sr_code=0):
        """mode: "mono" | "stereo" | "dual" | "joint" (the mode extension then varies per frame)"""
        self.T = Tables(ref_root)
        self.rng = np.random.default_rng(seed)
        self.mode, self.mpeg1, self.sr_code = mode, mpeg1, sr_code
        self.nch = 1 if mode == "mono" else 2
        self.sr_idx = sr_code if mpeg1 else 3 + sr_code
        self.rate = [44100, 48000, 32000][sr_code] // (1 if mpeg1 else 2)
        self.block = LONG
        self.frames = []      # (header word, side info bytes, slot size, record)
        self.stream = bytearray()
        self.slot_start = 0   # S_n: where this frame's own main-data slot starts in the main-data stream

    def next_block_type(self):
        r = self.rng.random()
        if self.block in (LONG, END):
            self.block = START if r < 0.3 else LONG
        elif self.block == START:
            self.block = SHORT
        else:
            self.block = SHORT if r < 0.5 else END
        return self.block

    def granule_channel(self, bw, budget_bits, block, mixed, scfsi, gr, is_right_intensity):
        """writes part 2 + part 3 into bw; returns (side-info fields, quantised samples[576])"""
        rng, T = self.rng, self.T
        ws = block != LONG
        f = {"block": block, "mixed": mixed, "global_gain": int(rng.integers(130, 215)), "scalefac_scale": int(rng.integers(0, 2)),
             "preflag": int(rng.integers(0, 2)), "count1table": int(rng.integers(0, 2)), "subblock_gain": [int(x) for x in rng.integers(0, 8, 3)]}
        bits = 0
        # part 2: scale factors
        if self.mpeg1:
            f["sfc"] = int(rng.integers(0, 16))
            s1, s2 = SLEN[f["sfc"]]
            if block == SHORT:
                n1 = 17 if mixed else 18
                groups = [(n1, s1), (18, s2)]
            else:
                groups = [(n, s) for i, (n, s) in enumerate(((6, s1), (5, s1), (5, s2), (5, s2))) if not (gr > 0 and scfsi[i])]
        else:
            f["sfc"] = int(rng.integers(0, 512))
            bi = (2 if mixed else 1) if block == SHORT else 0
            if is_right_intensity:
                s = f["sfc"] >> 1
                if s < 180:
                    sl, row = [s // 36, (s % 36) // 6, (s % 36) % 6, 0], 0
                elif s < 244:
                    sl, row = [((s - 180) % 64) >> 4, ((s - 180) % 16) >> 2, (s - 180) % 4, 0], 1
                else:
                    sl, row = [(s - 244) // 3, (s - 244) % 3, 0, 0], 2
            else:
                s = f["sfc"]
                if s < 400:
                    sl, row = [(s >> 4) // 5, (s >> 4) % 5, (s % 16) >> 2, s % 4], 3
                elif s < 500:
                    sl, row = [((s - 400) >> 2) // 5, ((s - 400) >> 2) % 5, (s - 400) % 4, 0], 4
                else:
                    sl, row = [(s - 500) // 3, (s - 500) % 3, 0, 0], 5
            groups = list(zip(NSFB_V2[row][bi], sl))
        for n, s in groups:
            if s:
                for _ in range(n):
                    bw.put(int(rng.integers(0, 1 << s)), s)
                bits += n * s
        part2_bits = bits
        # part 3: regions, tables, samples
        if ws:
            r1 = 36 if (self.mpeg1 or block == SHORT) else 54
            r2 = 576
            f["tables"] = [int(rng.choice(USABLE)), int(rng.choice(USABLE))]
        else:
            f["region0"], f["region1"] = int(rng.integers(0, 16)), int(rng.integers(0, 8))
            r1 = T.sfb_long[self.sr_idx][f["region0"] + 1]
            k = f["region0"] + f["region1"] + 2
            r2 = T.sfb_long[self.sr_idx][k] if k <= 22 else 576
            f["tables"] = [int(rng.choice(USABLE)) for _ in range(3)]
            if rng.random() < 0.15:
                f["tables"][int(rng.integers(0, 3))] = 0  # table 0: a region of zeros that costs no bits
        want_pairs = int(rng.integers(0, 289) * rng.random() ** 0.5)
        q = np.zeros(576, np.int64)
        i = 0
        ends = [min(r1, 2 * want_pairs), min(r2, 2 * want_pairs), 2 * want_pairs]
        for region, end in enumerate(ends[:len(f["tables"])] if ws else ends):
            t = f["tables"][region]
            if t == 0:
                i = max(i, end)
                continue
            tab = T.table(t)
            top = tab[4]
            while i < end and bits < budget_bits - 60:
                if rng.random() < 0.1:
                    x, y = int(rng.integers(0, top + 1)), int(rng.integers(0, top + 1))  # anywhere up to the table's largest value
                else:
                    x, y = int(min(top, rng.geometric(0.35) - 1)), int(min(top, rng.geometric(0.35) - 1))
                x, y = x * int(rng.choice([-1, 1])), y * int(rng.choice([-1, 1]))
                bits += put_pair(bw, tab, x, y)
                q[i], q[i + 1] = x, y
                i += 2
            if i < end:  # out of budget: the big-value area ends here
                break
        f["big_values"] = i // 2
        want_quads = int(rng.integers(0, 1 + (576 - i) // 4) * rng.random())
        for _ in range(want_quads):
            if i > 572 or bits >= budget_bits - 12:
                break
            v = [int(x) for x in rng.integers(-1, 2, 4)]
            bits += put_quad(bw, T.quads[f["count1table"]], v)
            q[i:i + 4] = v
            i += 4
        f["rzero"], f["part2_3_length"] = (i if bits > part2_bits else 0), bits  # (no part 3 at all: requantize.rs:54-57)
        assert bits < 4096
        return f, q

    def frame(self):
        rng = self.rng
        rates = BIT_RATES_V1 if self.mpeg1 else BIT_RATES_V2
        br = int(rng.integers(9 if self.mpeg1 else 8, 15))
        padding = int(rng.integers(0, 2))
        slots = (144 if self.mpeg1 else 72) * rates[br] * 1000 // self.rate + padding
        ngr = 2 if self.mpeg1 else 1
        side_len = (17 if self.nch == 1 else 32) if self.mpeg1 else (9 if self.nch == 1 else 17)
        cap = slots - 4 - side_len
        mode_bits = {"stereo": 0, "joint": 1, "dual": 2, "mono": 3}[self.mode]
        mode_ext = int(rng.integers(0, 4)) if self.mode == "joint" else 0
        intensity = self.mode == "joint" and (mode_ext & 1) != 0
        header = (0x7FF << 21) | ((3 if self.mpeg1 else 2) << 19) | (1 << 17) | (1 << 16) | (br << 12) | (self.sr_code << 10) | (padding << 9) \
            | (mode_bits << 6) | (mode_ext << 4) | (int(rng.integers(0, 2)) << 3) | (int(rng.integers(0, 2)) << 2)
        # where this frame's main data starts: right behind the previous frame's, at most 511 (255) bytes before its own slot
        reach = 511 if self.mpeg1 else 255
        if self.slot_start - len(self.stream) > reach:
            self.stream += bytes(self.slot_start - len(self.stream) - reach)  # stuffing nobody reads
        begin = self.slot_start - len(self.stream)
        avail_bits = 8 * (begin + cap) - 200  # (scale factors are written before the budget is looked at)
        blocks = [self.next_block_type() for _ in range(ngr)]
        mixed = [int(rng.integers(0, 2)) if b == SHORT else 0 for b in blocks]
        scfsi = [[int(rng.integers(0, 2)) if SHORT not in blocks else 0 for _ in range(4)] for _ in range(self.nch)]
        md = BitWriter()
        rec, used = [], 0
        share = avail_bits * float(rng.uniform(0.35, 0.98)) / (ngr * self.nch)
        for gr in range(ngr):
            for ch in range(self.nch):
                budget = int(min(4000, share * rng.uniform(0.6, 1.3), avail_bits - used - 16))
                f, q = self.granule_channel(md, max(budget, 0), blocks[gr], mixed[gr], scfsi[ch], gr, intensity and ch == 1)
                used += f["part2_3_length"]
                rec.append((f, q))
        md.align()
        data = md.bytes()
        assert len(data) <= begin + cap
        # side information
        si = BitWriter()
        if self.mpeg1:
            si.put(begin, 9)
            si.put(0, 5 if self.nch == 1 else 3)
            for ch in range(self.nch):
                for b in scfsi[ch]:
                    si.put(b, 1)
        else:
            si.put(begin, 8)
            si.put(0, 1 if self.nch == 1 else 2)
        for f, _ in rec:
            si.put(f["part2_3_length"], 12)
            si.put(f["big_values"], 9)
            si.put(f["global_gain"], 8)
            si.put(f["sfc"], 4 if self.mpeg1 else 9)
            if f["block"] != LONG:
                si.put(1, 1)
                si.put(f["block"], 2)
                si.put(f["mixed"], 1)
                for t in f["tables"]:
                    si.put(t, 5)
                for g in f["subblock_gain"]:
                    si.put(g, 3)
            else:
                si.put(0, 1)
                for t in f["tables"]:
                    si.put(t, 5)
                si.put(f["region0"], 4)
                si.put(f["region1"], 3)
            if self.mpeg1:
                si.put(f["preflag"], 1)
            si.put(f["scalefac_scale"], 1)
            si.put(f["count1table"], 1)
        si.align()
        side = si.bytes()
        assert len(side) == side_len, (len(side), side_len)
        self.stream += data
        self.frames.append((header, side, cap, {"mode_ext": mode_ext, "granules": rec, "main_data_begin": begin, "bitrate": rates[br]}))
        self.slot_start += cap

    def packets(self, n):
        """n frames as packets (header + side information + this frame's slot of the main-data stream) and their records"""
        for _ in range(n):
            self.frame()
        total = self.slot_start
        stream = bytes(self.stream) + bytes(max(0, total - len(self.stream)))
        out, pos = [], 0
        for header, side, cap, rec in self.frames:
            out.append((header.to_bytes(4, "big") + side + stream[pos:pos + cap], rec))
            pos += cap
        return out
