"""The oracle (and, gpu-marked, the HIP kernels through the C ABI) against fixtures produced by EXECUTING the reference's
Rust source text (tools/rs2fixtures.py + tools/rsinterp: a Rust-subset interpreter with Rust's arithmetic -- one rounding
per f32 operation, wrapping integers, glibc libm for the table formulas).  Bit patterns, not tolerances: this pins the
oracle to the reference itself, not to a reading of it.  tests/golden/rs_fixtures/manifest.json lists every case with the
reference file:lines it ran and the sha256 of the source files."""
import json
from pathlib import Path

import numpy as np
import pytest

import oracle
from helpers import bit_equal

FIX = Path(__file__).resolve().parent / "golden" / "rs_fixtures"
MANIFEST = json.loads((FIX / "manifest.json").read_text())


def load(group):
    z = np.load(FIX / (group + ".npz"))
    out = {}
    for k in z.files:
        dt, _ = MANIFEST[group]["arrays"][k]
        a = z[k]
        out[k] = a.view(np.float32) if dt == "float32" else a.view(np.float64) if dt == "float64" else a
    return out


def same(a, b):
    """bit-identical; for floats the sign of zero is not part of the parity contract"""
    a, b = np.asarray(a), np.asarray(b)
    assert a.shape == b.shape, (a.shape, b.shape)
    if a.dtype.kind == "f":
        return bit_equal(a, b)
    return np.array_equal(a, b)


def test_manifest_covers_the_hot_path():
    assert {"fft", "imdct", "aac", "mp3", "vorbis", "flac", "alac", "mp3_front", "aac_tools"} <= set(MANIFEST)
    for g, m in MANIFEST.items():
        assert m["cases"] and m["sources"], g
    # plain `+ - *` of the reference never wrapped while the fixtures were generated (a debug build would have run the
    # same way); the ALAC adaptive predictor is the documented exception domain
    for g in ("fft", "imdct", "aac", "mp3", "vorbis", "flac"):
        assert MANIFEST[g]["implicit_integer_wraps"] == 0, g


# ------------------------------------------------------------------------------------------ oracle == reference text

def test_oracle_fft():
    f = load("fft")
    for n in (2, 4, 8, 16, 32, 64, 128, 256, 512, 1024, 2048, 4096, 8192):
        x = f["in_%d" % n]
        want = f["out_%d" % n]
        got = oracle.fft((x[:, 0] + 1j * x[:, 1]).astype(np.complex64))
        assert same(np.stack([got.real, got.imag], 1), want), n
    for n in (2, 8, 16, 32, 64, 256, 1024, 4096):
        x = f["iin_%d" % n]
        got = oracle.ifft((x[:, 0] + 1j * x[:, 1]).astype(np.complex64))
        assert same(np.stack([got.real, got.imag], 1), f["iout_%d" % n]), n
    for n in (64, 128, 256, 512, 1024, 2048):
        w = oracle.fft_twiddles(n)
        assert same(np.stack([w.real, w.imag], 1), f["twiddle_%d" % n]), n


def test_oracle_imdct():
    f = load("imdct")
    for c in MANIFEST["imdct"]["cases"]:
        key = c["key"]
        got = oracle.imdct(f["spec_" + key][None], c["scale"])[0]
        assert same(got, f["out_" + key]), c
    for n in (4, 32, 128, 1024, 4096):
        tw = oracle.imdct_twiddles(n, 1.0)
        assert same(np.stack([tw.real, tw.imag], 1), f["twiddle_%d" % n]), n


def aac_side_of(f, walk, lanes):
    sd = f["side_" + walk]
    return np.tile(np.array([oracle.aac_side(s, sh, pv) for s, sh, pv in sd], np.uint8), (lanes, 1))


def test_oracle_aac():
    f = load("aac")
    for nm, (kbd, alpha, size) in {"kbd_long_win": (1, 4.0, 1024), "kbd_short_win": (1, 6.0, 128), "sine_long_win": (0, 0, 1024),
                                   "sine_short_win": (0, 0, 128)}.items():
        assert same(oracle.aac_window(kbd, alpha, size), f[nm]), nm
    for walk in ("a", "b", "c", "d"):
        coeffs = f["coeffs_" + walk]
        pcm, delay = oracle.aac_synth(coeffs, aac_side_of(f, walk, coeffs.shape[0]), f["delay_in_" + walk])
        assert same(pcm, f["pcm_" + walk]) and same(delay, f["delay_out_" + walk]), walk
        assert set(f["side_" + walk][:, 0].tolist()) == {0, 1, 2, 3}


def test_oracle_mp3():
    f = load("mp3")
    assert same(oracle.mp3_imdct_windows(), f["IMDCT_WINDOWS"])
    assert same(oracle.mp3_synthesis_window(), f["SYNTHESIS_D"])
    c = oracle.mp3_constants()
    assert same(c["half_cos_12"], f["IMDCT_HALF_COS_12"]) and same(c["cs"], f["ANTIALIAS_CS"]) and same(c["ca"], f["ANTIALIAS_CA"])
    lanes = f["dct32_in"].shape[0]
    for l in range(lanes):
        assert same(oracle.mp3_dct32(f["dct32_in"][l]), f["dct32_out"][l])
        for wi in range(4):
            o, oo = oracle.mp3_imdct36(f["imdct36_x"][l], f["IMDCT_WINDOWS"][wi], f["imdct36_overlap"][l])
            assert same(o, f["imdct36_out_w%d" % wi][l]) and same(oo, f["imdct36_ovout_w%d" % wi][l]), (l, wi)
        o, oo = oracle.mp3_imdct12_win(f["imdct36_x"][l], f["IMDCT_WINDOWS"][2], f["imdct36_overlap"][l])
        assert same(o, f["imdct12_out"][l]) and same(oo, f["imdct12_ovout"][l]), l
    for chain in ("long", "switch", "sr3", "sr8") + tuple("mix%d" % i for i in range(9)):
        k = "chain_%s_" % chain
        xr = f[k + "xr"]
        g = f[k + "side"]
        n = xr.shape[0]
        side = oracle.mp3_side(np.tile(g[:, 0], (n, 1)), np.tile(g[:, 1], (n, 1)), np.tile(g[:, 2], (n, 1)))
        pcm, ov, vv, vf = oracle.mp3_synth(xr, side, int(f[k + "sr"][0]), f[k + "overlap_in"].reshape(n, 576), f[k + "vvec_in"].reshape(n, 1024),
                                           np.full(n, f[k + "vfront"][0], np.int32))
        assert same(pcm, f[k + "pcm"]), chain
        assert same(ov, f[k + "overlap_out"].reshape(n, 576)) and same(vv, f[k + "vvec_out"].reshape(n, 1024)), chain
        assert (vf == f[k + "vfront"][1]).all(), chain
    for nf in (12, 36):
        x = f["poly%d_in" % nf]
        for l in range(x.shape[0]):
            v, fr = np.zeros(1024, np.float32), 0
            for p in range(x.shape[1]):
                o, v, fr = oracle.mp3_polyphase(v, fr, nf, x[l, p])
                assert same(o, f["poly%d_out" % nf][l, p]), (nf, l, p)
            assert same(v, f["poly%d_vvec_out" % nf][l].reshape(-1)) and fr == f["poly%d_vfront_out" % nf][0]


def test_oracle_vorbis():
    f = load("vorbis")
    for bs in (64, 128, 256, 512, 1024, 2048, 4096, 8192):
        assert same(oracle.vorbis_window(bs), f["window_%d" % bs]), bs
    assert same(oracle.vorbis_floor1_table(), f["FLOOR1_INVERSE_DB_TABLE"])
    for c in MANIFEST["vorbis"]["cases"]:
        if c["fn"] != "DspChannel::synth":
            continue
        b0, b1 = c["bs0_exp"], c["bs1_exp"]
        k = "synth_%d_%d_" % (b0, b1)
        sp = f[k + "spectra"]
        lanes = sp.shape[0]
        flags = np.tile(f[k + "flags"], (lanes, 1))
        prev = np.full(lanes, int(f[k + "prev_flag"][0]))
        _, po = oracle.vorbis_layout(b0, b1, flags, prev)
        pcm, ov, pf = oracle.vorbis_synth(b0, b1, sp, flags, prev, f[k + "overlap_in"], int(po[0, -1]))
        assert same(pcm, f[k + "pcm"]) and same(ov, f[k + "overlap_out"]), (b0, b1)
        assert (pf == f[k + "flags"][-1]).all()
    r = f["coupling_in"].copy()
    for m, a in f["coupling_pairs"]:
        r[m], r[a] = oracle.vorbis_inverse_coupling(r[m], r[a])
    assert same(r, f["coupling_out"])
    for c in range(4):
        want = f["dot_floor_out"][c]
        got = f["dot_floor_in"][c] if f["do_not_decode_out"][c] else oracle.vorbis_dot_product(f["dot_floor_in"][c], f["coupling_out"][c])
        assert same(got, want), c
    for c in MANIFEST["vorbis"]["cases"]:
        if not c["fn"].startswith("Floor1"):
            continue
        ci = c["case"]
        mult, n = f["floor1_%d_mult_n" % ci]
        for y, o in zip(f["floor1_%d_y" % ci], f["floor1_%d_out" % ci]):
            assert same(oracle.vorbis_floor1(f["floor1_%d_x" % ci], y, int(mult), int(n)), o), ci


def test_oracle_flac():
    f = load("flac")
    a, b = f["decor_ch0"], f["decor_ch1"]
    for mode in (1, 2, 3):
        i0, i1 = (a[4:] >> 1, b[4:] >> 2) if mode == 2 else (a[4:], b[4:])
        wa, wb = oracle.flac_decorrelate(mode, i0, i1)
        assert same(wa, f["decor_out0_m%d" % mode]) and same(wb, f["decor_out1_m%d" % mode]), mode
    for sh in (0, 1, 8, 31):
        assert same(oracle.flac_shl(a, sh), f["shl_%d" % sh]), sh
    for w, want in zip(f["rice_in"], f["rice_out"]):
        assert oracle.flac_rice_signed_to_i32(int(w)) == want
    for order in range(5):
        assert same(oracle.flac_fixed_predict(order, f["fixed_in_%d" % order]), f["fixed_out_%d" % order]), order
    assert same(oracle.flac_fixed_predict(4, f["fixed_in_wrap"]), f["fixed_out_wrap"])
    ci = 0
    while "lpc_%d_in" % ci in f:
        order, shift = f["lpc_%d_order_shift" % ci]
        got = oracle.flac_lpc_predict(int(order), f["lpc_%d_coeffs" % ci], int(shift), f["lpc_%d_in" % ci])
        assert same(got, f["lpc_%d_out" % ci]), (ci, order, shift)
        ci += 1
    assert ci >= 12


def test_oracle_alac():
    f = load("alac")
    ci = 0
    while "predict_%d_in" % ci in f:
        mode, order, shift, bps = f["predict_%d_params" % ci]
        d = oracle.alac_desc(np.array([mode]), np.array([order]), np.array([shift]), np.array([bps]))
        got = oracle.alac_predict(f["predict_%d_in" % ci][None], d, f["predict_%d_coeffs" % ci][None])[0]
        assert same(got, f["predict_%d_out" % ci]), (ci, mode, order, shift, bps)
        ci += 1
    assert ci >= 10
    ci = 0
    while "ms_%d_in0" % ci in f:
        w, s = f["ms_%d_weight_shift" % ci]
        o0, o1 = oracle.alac_decorrelate_mid_side(f["ms_%d_in0" % ci], f["ms_%d_in1" % ci], int(w), int(s))
        assert same(o0, f["ms_%d_out0" % ci]) and same(o1, f["ms_%d_out1" % ci]), ci
        ci += 1


def rq_desc(row):
    d = np.zeros(1, oracle.MP3_REQUANT_DTYPE)
    d["global_gain"], d["flags"], d["block_type"], d["is_mixed"] = row[0], row[1], row[2], row[3]
    d["subblock_gain"], d["rzero"], d["scalefacs"] = row[4:7], row[7] | (row[8] << 8), row[9:48]
    return d


def st_desc(r):
    d = np.zeros(1, oracle.MP3_STEREO_DTYPE)
    d["flags"] = (1 if r[3] else 0) | (2 if r[4] else 0) | (4 if r[0] else 0) | (8 if r[8] else 0)
    d["block_type"], d["is_mixed"], d["rzero0"], d["rzero1"], d["scalefacs1"] = r[1], r[2], r[6], r[7], r[9:48]
    return d


def test_oracle_mp3_front():
    f = load("mp3_front")
    assert same(oracle.mp3_pow43(), f["POW43"])
    m1, m2 = oracle.mp3_intensity_ratios()
    assert same(m1, f["IS_RATIOS_MPEG1"]) and same(m2, f["IS_RATIOS_MPEG2"])
    for sr in range(9):
        assert np.array_equal(oracle.mp3_sfb_long(sr), f["SFB_LONG_BANDS"][sr])
        short, mixed, switch = oracle.mp3_sfb_tables(sr)
        assert np.array_equal(short, f["SFB_SHORT_BANDS"][sr]) and switch == f["SFB_MIXED_SWITCH_POINT"][sr]
        row = f["SFB_MIXED_BANDS"][sr]
        assert np.array_equal(mixed, row[row >= 0])
    for ci in range(f["rq_quant"].shape[0]):
        got = oracle.mp3_requantize(f["rq_quant"][ci][None], rq_desc(f["rq_desc"][ci]), int(f["rq_sr"][ci]))[0]
        assert same(got, f["rq_out"][ci]), ci
    for ci in range(f["st_in"].shape[0]):
        r = f["st_desc"][ci]
        a, b = oracle.mp3_stereo(f["st_in"][ci, 0], f["st_in"][ci, 1], st_desc(r)[0], int(r[5]))
        assert same(np.stack([a, b]), f["st_out"][ci]), ci
        if r[3] or r[4]:
            assert (f["st_rzero_out"][ci] == max(r[6], r[7])).all()


def tns_filters_of(f, ci):
    """(start, end, order, direction, lpc) per filter, by the band arithmetic of tns.rs:149-175 (what the host keeps)."""
    k = "tns_%d_" % ci
    long_win, nwin, max_sfb = (int(v) for v in f[k + "info"])
    bands = f["swb_long_48k"] if long_win else f["swb_short_48k"]
    rate_idx = int(f["rate_idx_48k"][0])
    tmax = min(int((f["TNS_MAX_LONG_BANDS"] if long_win else f["TNS_MAX_SHORT_BANDS"])[rate_idx]), max_sfb)
    out = []
    for w in range(nwin):
        bottom = len(bands) - 1
        for fi in range(int(f[k + "n_filt"][w])):
            ln, order, direction = (int(v) for v in f[k + "filters"][w, fi, :3])
            top = bottom
            bottom = max(0, top - ln)
            if order == 0:
                continue
            out.append((w * 128 + int(bands[min(bottom, tmax)]), w * 128 + int(bands[min(top, tmax)]), order, direction,
                        f[k + "filters"][w, fi, 3:3 + order]))
    return out


def js_modes_of(f, ci):
    """per-window-band tool + scale, by the group -> window expansion of cpe.rs:110-143 (what the host keeps)."""
    k = "js_%d_" % ci
    long_win, nwin, max_sfb, ms_mask_present = (int(v) for v in f[k + "info"])
    mode, scale = np.zeros(128, np.uint8), np.zeros(128, np.float32)
    g = 0
    for w in range(nwin):
        if w > 0 and not f[k + "grouping"][w - 1]:
            g += 1
        for sfb in range(max_sfb):
            idx = sfb if long_win else w * 16 + sfb
            cb0, cb1 = int(f[k + "sfb_cb0"][g, sfb]), int(f[k + "sfb_cb1"][g, sfb])
            if cb1 in (14, 15):
                invert = ms_mask_present == 1 and f[k + "ms_used"][g, sfb]
                mode[idx] = 2
                scale[idx] = np.float32(1.0 if cb1 == 15 else -1.0) * np.float32(-1.0 if invert else 1.0) * f[k + "scales1"][g, sfb]
            elif cb0 == 13 or cb1 == 13:
                pass
            elif f[k + "ms_used"][g, sfb]:
                mode[idx] = 1
    return long_win, nwin, max_sfb, mode, scale


def test_oracle_aac_tools():
    f = load("aac_tools")
    for ci in range(6):
        c = f["tns_%d_in" % ci].copy()
        for start, end, order, direction, lpc in tns_filters_of(f, ci):
            c = oracle.aac_tns_filter(c, start, end, order, direction, lpc)
        assert same(c, f["tns_%d_out" % ci]), ci
    for ci in range(4):
        long_win, nwin, max_sfb, mode, scale = js_modes_of(f, ci)
        bands = (f["swb_long_48k"] if long_win else f["swb_short_48k"]).astype(np.uint16)
        l, r = oracle.aac_joint_stereo(f["js_%d_left_in" % ci], f["js_%d_right_in" % ci], nwin, max_sfb, bands, mode, scale)
        assert same(l, f["js_%d_left_out" % ci]) and same(r, f["js_%d_right_out" % ci]), ci
    for ci in range(5):
        p = f["pulse_%d_params" % ci]
        got = oracle.aac_pulse(f["pulse_%d_in" % ci], f["swb_long_48k"], f["pulse_%d_scales0" % ci], int(p[0]), int(p[1]), p[2:6], p[6:10])
        assert same(got, f["pulse_%d_out" % ci]), ci


def test_oracle_vorbis_floor0():
    f = load("vorbis")
    n_cases = 0
    for c in MANIFEST["vorbis"]["cases"]:
        if "Floor0" not in c["fn"]:
            continue
        n_cases += 1
        k = "floor0_%d_" % c["case"]
        b0, b1, order, rate, map_size, amp_bits, amp_off, amp_s, amp_l = (int(v) for v in f[k + "params"])
        assert np.array_equal(oracle.vorbis_bark_map(1 << (b0 - 1), rate, map_size), f[k + "map_short"])
        assert np.array_equal(oracle.vorbis_bark_map(1 << (b1 - 1), rate, map_size), f[k + "map_long"])
        assert same(oracle.vorbis_floor0_coeffs(f[k + "angles"]), f[k + "coeffs"])
        for which, bexp, amp in (("short", b0, amp_s), ("long", b1, amp_l)):
            got = oracle.vorbis_floor0(f[k + "coeffs"], f[k + "map_" + which], map_size, amp_bits, amp_off, amp)
            assert same(got, f[k + "out_" + which]), (c["case"], which)
    assert n_cases >= 4
