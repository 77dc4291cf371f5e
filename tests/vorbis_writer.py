"""TEST ONLY.  A syntax-level Vorbis I stream writer for tests/test_vorbis_packets.py: the inverse of the reference's parse stage
(symphonia-codec-vorbis/src/lib.rs:374-470 identification header, 479-520 + 529-760 setup header, 128-190 audio packet;
codebook.rs:181-330 codebooks, 113-175 codeword assignment; floor.rs:441-565 floor 1 set-up, 657-722 floor 1 packet data;
residue.rs:57-123 residue set-up, 185-400 residue packet data).  Not an encoder: the set-up (Huffman trees, VQ lattices, floor
partitions, residue classes, couplings, submaps) and the packets (modes, floor posts, classifications, VQ entries) are drawn at
random inside what the syntax allows -- plain, length-ordered and sparse codebooks, lookup types 1 and 2, sequence_p, floor 1 with
and without subclasses, residue types 0 / 1 / 2 with several passes, channel coupling, two submaps, unused floors, short and long
blocks in every order.  Bits are packed least significant first (Vorbis I, section 2)."""
import numpy as np


class BitWriterRtl:
    def __init__(self):
        self.acc, self.n, self.out = 0, 0, bytearray()

    def put(self, value, bits):
        assert 0 <= value < (1 << bits) or bits == 0, (value, bits)
        self.acc |= value << self.n
        self.n += bits
        while self.n >= 8:
            self.out.append(self.acc & 0xFF)
            self.acc >>= 8
            self.n -= 8

    def bytes(self):
        return bytes(self.out) + (bytes([self.acc & 0xFF]) if self.n else b"")


def ilog(x):
    return int(x).bit_length()


def float32_pack(mantissa, exponent):
    """the Vorbis float format (codebook.rs:11-18): value = mantissa * 2^exponent, |mantissa| < 2^21"""
    sign = 0x80000000 if mantissa < 0 else 0
    return sign | ((exponent + 788) << 21) | abs(mantissa)


def codewords(lengths):
    """Vorbis I section 3.2.1: every entry takes the lowest codeword still free at its length, in tree order
    (the reference's own test vector for this: [2,4,4,4,4,2,3,3] -> [0,4,5,6,7,2,6,7], codebook.rs `verify_synthesize_codewords`)"""
    free = {0: [0]}  # length -> sorted list of free prefixes of that length
    out = []
    for ln in lengths:
        # the lowest free node at a depth <= ln, taken leftmost in tree order
        best = None
        for d, nodes in free.items():
            if d <= ln and nodes:
                cand = nodes[0] << (ln - d)
                if best is None or cand < best[0]:
                    best = (cand, d)
        assert best is not None, "over-specified tree"
        cand, d = best
        node = free[d].pop(0)
        # descending to the leaf frees the right siblings along the way
        for depth in range(d + 1, ln + 1):
            node <<= 1
            free.setdefault(depth, []).append(node | 1)
            free[depth].sort()
        out.append(node)
    assert not any(free.values()), "under-specified tree"
    return out


assert codewords([2, 4, 4, 4, 4, 2, 3, 3]) == [0, 4, 5, 6, 7, 2, 6, 7]


def complete_lengths(rng, n, shuffle=True):
    """n code lengths whose Kraft sum is exactly one"""
    if n == 1:
        return [1]  # (a single-entry book: the reference doubles it, codebook.rs:262-265)
    k = (n - 1).bit_length() - 1 if n & (n - 1) else n.bit_length() - 1
    short = (1 << (k + 1)) - n
    lens = [k] * short + [k + 1] * (n - short)
    if shuffle:
        rng.shuffle(lens)
    return [int(x) for x in lens]


class Codebook:
    def __init__(self, rng, dims, entries, kind="plain", lookup=0, sparse_unused=0):
        """kind: plain | ordered | sparse; lookup: 0 (scalar), 1 (lattice), 2 (listed)"""
        self.dims, self.entries, self.kind, self.lookup = dims, entries, kind, lookup
        used = entries - sparse_unused
        lens = complete_lengths(rng, used, shuffle=(kind != "ordered"))
        if kind == "ordered":
            lens.sort()
        self.used = list(range(entries))
        if kind == "sparse":
            self.used = sorted(int(x) for x in rng.choice(entries, used, replace=False))
        self.lens = dict(zip(self.used, lens))
        self.codes = dict(zip(self.used, codewords(lens)))
        if lookup:
            self.value_bits = int(rng.integers(2, 6))
            self.sequence_p = int(rng.random() < 0.3)
            self.min = (int(rng.integers(-12, 1)), int(rng.integers(-3, 1)))     # (mantissa, exponent)
            self.delta = (int(rng.integers(1, 6)), int(rng.integers(-3, 1)))
            if lookup == 1:
                self.lookup_values = int(np.floor(np.float32(entries) ** (np.float32(1.0) / np.float32(dims))))
                while (self.lookup_values + 1) ** dims <= entries:
                    self.lookup_values += 1
                while self.lookup_values ** dims > entries:
                    self.lookup_values -= 1
            else:
                self.lookup_values = entries * dims
            self.mult = [int(x) for x in rng.integers(0, 1 << self.value_bits, self.lookup_values)]

    def write(self, bw):
        bw.put(0x564342, 24)
        bw.put(self.dims, 16)
        bw.put(self.entries, 24)
        if self.kind == "ordered":
            bw.put(1, 1)
            lens = [self.lens[e] for e in range(self.entries)]
            cur, length = 0, lens[0]
            bw.put(length - 1, 5)
            while cur < self.entries:
                num = sum(1 for x in lens if x == length)
                bw.put(num, ilog(self.entries - cur))
                cur += num
                length += 1
        else:
            bw.put(0, 1)
            bw.put(int(self.kind == "sparse"), 1)
            for e in range(self.entries):
                if self.kind == "sparse":
                    bw.put(int(e in self.lens), 1)
                    if e not in self.lens:
                        continue
                bw.put(self.lens[e] - 1, 5)
        bw.put(self.lookup, 4)
        if self.lookup:
            bw.put(float32_pack(*self.min), 32)
            bw.put(float32_pack(*self.delta), 32)
            bw.put(self.value_bits - 1, 4)
            bw.put(self.sequence_p, 1)
            for m in self.mult:
                bw.put(m, self.value_bits)

    def put(self, bw, entry):
        """one codeword, most significant bit first"""
        code, ln = self.codes[entry], self.lens[entry]
        for b in range(ln - 1, -1, -1):
            bw.put((code >> b) & 1, 1)

    def vector(self, entry):
        """the VQ vector of an entry in f32 arithmetic (codebook.rs:33-92)"""
        f = np.float32
        mn = f(self.min[0]) * f(2.0) ** f(self.min[1])
        dl = f(self.delta[0]) * f(2.0) ** f(self.delta[1])
        out, last, div = [], f(0.0), 1
        for d in range(self.dims):
            m = self.mult[(entry // div) % self.lookup_values] if self.lookup == 1 else self.mult[entry * self.dims + d]
            v = f(f(f(m) * dl) + mn) + last
            v = f(v)
            if self.sequence_p:
                last = v
            div *= self.lookup_values
            out.append(v)
        return np.array(out, np.float32)


class Stream:
    def __init__(self, seed, nch, bs0_exp, bs1_exp, residue_types=(2, 1), couple=True):
        self.rng = rng = np.random.default_rng(seed)
        self.nch, self.bs0_exp, self.bs1_exp = nch, bs0_exp, bs1_exp
        B = self.books = []

        def book(*a, **k):
            B.append(Codebook(rng, *a, **k))
            return len(B) - 1

        book(1, 4)  # book 0 is never a residue value book (residue.rs:105: book number 0 is invalid there)
        self.floors, self.residues, self.mappings = [], [], []
        for which, exp in enumerate((bs0_exp, bs1_exp)):
            n2 = (1 << exp) >> 1
            # ---- floor 1: classes 0 (with subclasses) and 1 (without), an unused subclass book in class 0
            ybook_a = book(1, 64 if which == 0 else 128, kind="plain")
            ybook_b = book(1, 96, kind="sparse", sparse_unused=20)
            ybook_c = book(1, 32, kind="ordered")
            master = book(1, 8)  # class 0: 3 dimensions x 1 subclass bit
            classes = [{"dim": 3, "bits": 1, "master": master, "sub": [ybook_a, ybook_b]},
                       {"dim": 2, "bits": 0, "master": None, "sub": [ybook_c]},
                       {"dim": 1, "bits": 1, "master": book(1, 2), "sub": [None, ybook_a]}]
            parts = [int(x) for x in rng.integers(0, 3, int(rng.integers(2, 7)))]
            parts[0] = 0
            rangebits = exp - 1
            count = sum(classes[c]["dim"] for c in parts)
            xs = sorted(int(x) for x in rng.choice(np.arange(1, n2), count, replace=False))
            rng.shuffle(xs)
            self.floors.append({"parts": parts, "classes": classes, "mult": int(rng.integers(1, 5)), "rangebits": rangebits, "xs": [int(x) for x in xs]})
            # ---- residue: three classes -- nothing, one pass, three passes (the middle pass unused)
            rtype = residue_types[which]
            dims_a, dims_b = (2, 4) if which == 0 else (4, 8)
            vq_a = book(dims_a, 3 ** dims_a + 3, lookup=1)  # (not a perfect power: the lattice size does not hang on how a root rounds)
            vq_b = book(dims_b, 25, lookup=2)
            vq_c = book(2, 16, kind="sparse", sparse_unused=3, lookup=2)
            ppc = 2
            classbook = book(ppc, 3 ** ppc)
            psize = 16 if which == 0 else 32
            span = n2 * (nch if rtype == 2 else 1)
            end = span if rng.random() < 0.5 else span - 2 * psize   # (an end short of the block: the top stays zero)
            self.residues.append({"type": rtype, "begin": 0, "end": end, "psize": psize, "classbook": classbook, "ppc": ppc,
                                  "classes": [{}, {0: vq_a}, {0: vq_b, 2: vq_c}]})
        # ---- mappings: the short mode's with one submap, the long mode's with two when there are three or more channels
        for which in range(2):
            coupling = [(0, 1)] if (couple and nch >= 2) else []
            if nch >= 3 and which == 1:
                mux = [0, 0] + [1] * (nch - 2)
                submaps = [(which, which), (which, which)]
            else:
                mux, submaps = [0] * nch, [(which, which)]
            self.mappings.append({"coupling": coupling, "mux": mux, "submaps": submaps})
        self.prev_flag = None

    # ------------------------------------------------------------------ headers
    def ident(self):
        import struct
        return b"\x01vorbis" + struct.pack("<IBIiiiBB", 0, self.nch, 44100, 0, 128000, 0, (self.bs1_exp << 4) | self.bs0_exp, 1)

    def setup(self):
        bw = BitWriterRtl()
        bw.put(len(self.books) - 1, 8)
        for b in self.books:
            b.write(bw)
        bw.put(0, 6)
        bw.put(0, 16)  # one time-domain transform placeholder
        bw.put(len(self.floors) - 1, 6)
        for f in self.floors:
            bw.put(1, 16)
            bw.put(len(f["parts"]), 5)
            for c in f["parts"]:
                bw.put(c, 4)
            for c in f["classes"][:max(f["parts"]) + 1]:
                bw.put(c["dim"] - 1, 3)
                bw.put(c["bits"], 2)
                if c["bits"]:
                    bw.put(c["master"], 8)
                for s in c["sub"]:
                    bw.put(0 if s is None else s + 1, 8)
            bw.put(f["mult"] - 1, 2)
            bw.put(f["rangebits"], 4)
            for x in f["xs"]:
                bw.put(x, f["rangebits"])
        bw.put(len(self.residues) - 1, 6)
        for r in self.residues:
            bw.put(r["type"], 16)
            bw.put(r["begin"], 24)
            bw.put(r["end"], 24)
            bw.put(r["psize"] - 1, 24)
            bw.put(len(r["classes"]) - 1, 6)
            bw.put(r["classbook"], 8)
            for c in r["classes"]:
                mask = sum(1 << p for p in c)
                bw.put(mask & 7, 3)
                bw.put(int(mask >> 3 != 0), 1)
                if mask >> 3:
                    bw.put(mask >> 3, 5)
            for c in r["classes"]:
                for p in sorted(c):
                    bw.put(c[p], 8)
        bw.put(len(self.mappings) - 1, 6)
        for m in self.mappings:
            bw.put(0, 16)
            bw.put(int(len(m["submaps"]) > 1), 1)
            if len(m["submaps"]) > 1:
                bw.put(len(m["submaps"]) - 1, 4)
            bw.put(int(bool(m["coupling"])), 1)
            if m["coupling"]:
                bw.put(len(m["coupling"]) - 1, 8)
                for mag, ang in m["coupling"]:
                    bw.put(mag, ilog(self.nch - 1))
                    bw.put(ang, ilog(self.nch - 1))
            bw.put(0, 2)
            if len(m["submaps"]) > 1:
                for x in m["mux"]:
                    bw.put(x, 4)
            for fl, rs in m["submaps"]:
                bw.put(0, 8)
                bw.put(fl, 8)
                bw.put(rs, 8)
        bw.put(1, 6)  # two modes
        for which in range(2):
            bw.put(which, 1)
            bw.put(0, 16)
            bw.put(0, 16)
            bw.put(which, 8)
        bw.put(1, 1)  # framing
        return b"\x05vorbis" + bw.bytes()

    def extra_data(self):
        return self.ident() + self.setup()

    # ------------------------------------------------------------------ audio packets
    def packet(self, long_block=None):
        """(bytes, record): record = {"long", "floor_y": per channel list or None, "residue": per channel expected f32 vector or None}"""
        rng = self.rng
        which = int(rng.integers(0, 2)) if long_block is None else int(long_block)
        exp = self.bs1_exp if which else self.bs0_exp
        n2 = (1 << exp) >> 1
        m = self.mappings[which]
        bw = BitWriterRtl()
        bw.put(0, 1)
        bw.put(which, 1)
        if which:
            bw.put(int(rng.integers(0, 2)), 1)
            bw.put(int(rng.integers(0, 2)), 1)
        floor_y, unused = [], []
        for ch in range(self.nch):
            f = self.floors[m["submaps"][m["mux"][ch]][0]]
            if rng.random() < 0.15:
                bw.put(0, 1)
                floor_y.append(None)
                unused.append(True)
                continue
            bw.put(1, 1)
            rng_bits = ilog([256, 128, 86, 64][f["mult"] - 1] - 1)
            ys = [int(rng.integers(0, 1 << rng_bits)), int(rng.integers(0, 1 << rng_bits))]
            bw.put(ys[0], rng_bits)
            bw.put(ys[1], rng_bits)
            for c in f["parts"]:
                cl = f["classes"][c]
                cval = int(rng.integers(0, 1 << (cl["bits"] * cl["dim"]))) if cl["bits"] else 0
                if cl["bits"]:
                    self.books[cl["master"]].put(bw, cval)
                for _ in range(cl["dim"]):
                    sb = cl["sub"][cval & ((1 << cl["bits"]) - 1)]
                    cval >>= cl["bits"]
                    if sb is None:
                        ys.append(0)
                    else:
                        b = self.books[sb]
                        e = int(rng.choice(b.used)) if rng.random() < 0.7 else b.used[0]
                        b.put(bw, e)
                        ys.append(e)
            floor_y.append(ys)
            unused.append(False)
        # a coupled pair is decoded if either channel's floor is in use (lib.rs:165-177)
        decode = [not u for u in unused]
        for mag, ang in m["coupling"]:
            if decode[mag] != decode[ang]:
                decode[mag] = decode[ang] = True
        expect = [np.zeros(n2, np.float32) for _ in range(self.nch)]
        for si, (_, ri) in enumerate(m["submaps"]):
            chans = [c for c in range(self.nch) if m["mux"][c] == si]
            self.write_residue(bw, self.residues[ri], chans, decode, n2, expect)
        rec = {"long": bool(which), "floor_y": floor_y, "residue": expect, "decoded": decode, "coupled": bool(m["coupling"])}
        return bw.bytes(), rec

    def write_residue(self, bw, r, chans, decode, n2, expect):
        rng = self.rng
        rtype, psize, ppc = r["type"], r["psize"], r["ppc"]
        ncls = len(r["classes"])
        if rtype == 2:
            full = n2 * len(chans)
            end = min(r["end"], full)
            parts = (end - min(r["begin"], full)) // psize
            if not any(decode[c] for c in chans):
                return
            vectors = [np.zeros(full, np.float32)]
            lanes = [0]
        else:
            end = min(r["end"], n2)
            parts = (end - min(r["begin"], n2)) // psize
            live = [c for c in chans if decode[c]]
            if not live:
                return
            vectors = [expect[c] for c in live]
            lanes = list(range(len(live)))
        assert parts % ppc == 0
        classes = [[int(x) for x in rng.choice(ncls, parts, p=[0.3, 0.4, 0.3])] for _ in lanes]
        max_pass = max(max(c) if c else 0 for c in r["classes"])
        for p in range(max_pass + 1):
            for first in range(0, parts, ppc):
                if p == 0:
                    for ln in lanes:
                        code = 0
                        for k in range(ppc):
                            code = code * ncls + classes[ln][first + k]
                        self.books[r["classbook"]].put(bw, code)
                for part in range(first, first + ppc):
                    for ln in lanes:
                        cl = r["classes"][classes[ln][part]]
                        if p not in cl:
                            continue
                        b = self.books[cl[p]]
                        seg = vectors[ln][r["begin"] + psize * part: r["begin"] + psize * (part + 1)]
                        if rtype == 0:
                            step = psize // b.dims
                            for i in range(step):
                                e = int(rng.choice(b.used))
                                b.put(bw, e)
                                seg[i::step][:b.dims] += b.vector(e)
                        else:
                            for i in range(0, psize, b.dims):
                                e = int(rng.choice(b.used))
                                b.put(bw, e)
                                seg[i:i + b.dims] += b.vector(e)
        if rtype == 2:
            for i, c in enumerate(chans):
                expect[c][:] = vectors[0][i::len(chans)]
