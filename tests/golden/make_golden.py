#!/usr/bin/env python3
"""Extract the known-answer INPUT vectors the reference's own in-source tests use
for the hot path, and the literal constants the oracle regenerates by closed
form, into tests/golden/ref_kats.json.

Runs only where /root/reference exists (the build container).  The JSON is
committed; tests read the JSON, never the reference tree.

Sources (relative to the Symphonia tree):
  symphonia-core/src/dsp/mdct.rs:180-195          Imdct N=32 ramp, scale sqrt(2/64)
  symphonia-core/src/dsp/fft/mod.rs:88-153        64-point complex TEST_VECTOR
  symphonia-bundle-mp3/src/layer3/hybrid_synthesis.rs:512-516, 804-808   18 literals
  symphonia-bundle-mp3/src/synthesis.rs:868-873   32 literals for dct32
  symphonia-bundle-flac/src/decoder.rs:646-661    rice sign map cases
  symphonia-bundle-mp3/src/layer3/common.rs:9-172, requantize.rs:256-257   band-edge / pre-emphasis tables
and the literal twiddle / scale constants (as IEEE-754 bit patterns) of
  dsp/fft/no_simd.rs:307-324, 374-383; hybrid_synthesis.rs:611-630, 668-678,
  722-730; synthesis.rs:13-142, 354-396; vorbis floor.rs:21-112.
"""
import json
import re
import sys
from pathlib import Path

import numpy as np

REF = Path("/root/reference")
OUT = Path(__file__).resolve().parent / "ref_kats.json"


def floats_in(block):
    block = re.sub(r"//[^\n]*", "", block)
    return [float(s.replace("_", "")) for s in re.findall(r"-?\d+\.\d+(?:_\d+)*(?:e-?\d+)?", block)]


def bits(vals):
    return [int(np.float32(v).view(np.uint32)) for v in vals]


def main():
    if not REF.exists():
        sys.exit("reference tree not present; committed ref_kats.json is authoritative")
    kat = {}

    src = (REF / "symphonia-core/src/dsp/mdct.rs").read_text()
    m = re.search(r"const TEST_VECTOR: \[f32; 32\] = \[(.*?)\];", src, re.S)
    kat["imdct32_input"] = floats_in(m.group(1))
    kat["imdct32_scale"] = "sqrt(2/64)"
    kat["imdct_tolerance"] = 0.00001

    src = (REF / "symphonia-core/src/dsp/fft/mod.rs").read_text()
    pairs = re.findall(r"Complex \{ re: (-?[\d.]+), im: (-?[\d.]+) \}", src)
    assert len(pairs) == 64
    kat["fft64_input"] = [[float(a), float(b)] for a, b in pairs]
    kat["fft_tolerance"] = 0.00001

    src = (REF / "symphonia-bundle-mp3/src/layer3/hybrid_synthesis.rs").read_text()
    vs = re.findall(r"const TEST_VECTOR: \[f32; 18\] = \[(.*?)\];", src, re.S)
    assert len(vs) == 2 and floats_in(vs[0]) == floats_in(vs[1])
    kat["mp3_imdct_input18"] = floats_in(vs[0])
    kat["mp3_tolerance"] = 0.00001
    lit = {}
    m = re.search(r"const SCALE: \[f32; 18\] = \[(.*?)\];", src, re.S)
    lit["dct_iv_scale"] = bits(floats_in(m.group(1)))
    m = re.search(r"const SCALE: \[f32; 9\] = \[(.*?)\];", src, re.S)
    v = floats_in(m.group(1))
    assert len(v) == 8  # SQRT_2 is symbolic
    lit["sdct18_scale_without_m4"] = bits(v)
    m = re.search(r"const D: \[f32; 7\] = \[(.*?)\];", src, re.S)
    lit["sdct9_d"] = bits(floats_in(m.group(1)))

    src = (REF / "symphonia-bundle-mp3/src/synthesis.rs").read_text()
    m = re.search(r"const TEST_VECTOR: \[f32; 32\] = \[(.*?)\];", src, re.S)
    kat["mp3_dct32_input"] = floats_in(m.group(1))
    for name in ("COS_16", "COS_8", "COS_4", "COS_2"):
        m = re.search(r"const %s: \[f32; \d+\] = \[(.*?)\];" % name, src, re.S)
        lit[name.lower()] = bits(floats_in(m.group(1)))
    m = re.search(r"const COS_1: f32 = ([\d._]+);", src)
    lit["cos_1"] = bits([float(m.group(1).replace("_", ""))])
    m = re.search(r"static SYNTHESIS_D: \[f32; 512\] = \[(.*?)\];", src, re.S)
    lit["synthesis_d"] = bits(floats_in(m.group(1)))
    assert len(lit["synthesis_d"]) == 512

    src = (REF / "symphonia-core/src/dsp/fft/no_simd.rs").read_text()
    tw = re.findall(r"complex!\((-?[0-9.]+), (-?[0-9.]+)\)", src)
    assert len(tw) == 16
    lit["fft32_general_twiddles"] = [bits([float(a), float(b)]) for a, b in tw[:12]]
    lit["fft16_general_twiddles"] = [bits([float(a), float(b)]) for a, b in tw[12:]]

    src = (REF / "symphonia-codec-vorbis/src/floor.rs").read_text()
    m = re.search(r"FLOOR1_INVERSE_DB_TABLE: \[f32; 256\] = \[(.*?)\];", src, re.S)
    body = re.sub(r"//[^\n]*", "", m.group(1))
    lit["floor1_inverse_db"] = bits([float(s) for s in body.split(",") if s.strip()])
    assert len(lit["floor1_inverse_db"]) == 256

    kat["rice_cases"] = [[0, 0], [1, -1], [2, 1], [3, -2], [4, 2], [5, -3], [6, 3], [7, -4],
                         [8, 4], [9, -5], [10, 5], [4294967295, -2147483648]]
    src = (REF / "symphonia-bundle-flac/src/decoder.rs").read_text()
    for w, e in kat["rice_cases"]:
        ws = "u32::max_value()" if w == 4294967295 else str(w)
        es = "-2_147_483_648" if e == -2147483648 else str(e)
        assert "rice_signed_to_i32(%s), %s)" % (ws, es) in src, (w, e)
    # scale-factor band edge tables and the pre-emphasis table (standards data; the repo states them as band widths)
    src = (REF / "symphonia-bundle-mp3/src/layer3/common.rs").read_text()
    def int_rows(block):
        block = re.sub(r"//[^\n]*", "", block)
        return [[int(v) for v in re.findall(r"\d+", row)] for row in re.findall(r"\[([^\[\]]*)\]", block)]
    m = re.search(r"pub static SFB_LONG_BANDS: \[\[usize; 23\]; 9\] = \[(.*?)\n\];", src, re.S)
    kat["mp3_sfb_long"] = int_rows(m.group(1))
    m = re.search(r"pub static SFB_SHORT_BANDS: \[\[usize; 40\]; 9\] = \[(.*?)\n\];", src, re.S)
    kat["mp3_sfb_short"] = int_rows(m.group(1))
    m = re.search(r"pub const SFB_MIXED_BANDS: \[&\[usize\]; 9\] = \[(.*?)\n\];", src, re.S)
    kat["mp3_sfb_mixed"] = int_rows(m.group(1))
    m = re.search(r"pub const SFB_MIXED_SWITCH_POINT: \[usize; 9\] = \[(.*?)\];", src)
    kat["mp3_sfb_mixed_switch"] = [int(v) for v in m.group(1).split(",")]
    assert [len(r) for r in kat["mp3_sfb_long"]] == [23] * 9 and [len(r) for r in kat["mp3_sfb_short"]] == [40] * 9
    assert len(kat["mp3_sfb_mixed"]) == 9
    src = (REF / "symphonia-bundle-mp3/src/layer3/requantize.rs").read_text()
    m = re.search(r"const PRE_EMPHASIS: \[u8; 22\] =\s*\[(.*?)\];", src, re.S)
    kat["mp3_pre_emphasis"] = [int(v) for v in m.group(1).split(",")]
    assert len(kat["mp3_pre_emphasis"]) == 22
    kat["literal_bits"] = lit
    OUT.write_text(json.dumps(kat, indent=1))
    print("wrote", OUT)


if __name__ == "__main__":
    main()
