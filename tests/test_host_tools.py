"""Host-side tools of the C ABI (AAC pulse, Vorbis floor 0: libm-dependent stages that stay on the CPU by design) against
the reference-text fixtures, and the per-record status arrays (what the reference would have answered per block)."""
import numpy as np
import pytest

import symphonia_amd as sa
from emu_lib import emu_ctx  # noqa: F401
from test_rs_fixtures import MANIFEST, load, same


def test_pulse_matches_the_reference_text():
    f = load("aac_tools")
    for ci in range(5):
        p = f["pulse_%d_params" % ci]
        rec = np.zeros(1, sa.AAC_PULSE_DTYPE)
        rec["frame"], rec["number_pulse"], rec["pulse_start_sfb"] = 1, p[0], p[1]
        rec["pulse_offset"], rec["pulse_amp"], rec["scales0"] = p[2:6], p[6:10], f["pulse_%d_scales0" % ci]
        coeffs = np.zeros((3, 1024), np.float32)
        coeffs[1] = f["pulse_%d_in" % ci]
        coeffs[2] = 7.0
        sa.aac_pulse(coeffs, rec, f["swb_long_48k"])
        assert same(coeffs[1], f["pulse_%d_out" % ci]), ci
        assert not coeffs[0].any() and (coeffs[2] == 7.0).all()   # other frames untouched


def test_pulse_argument_checks():
    lib = sa.default_library()
    rec = np.zeros(1, sa.AAC_PULSE_DTYPE)
    rec["frame"], rec["number_pulse"] = 5, 1
    swb = np.arange(0, 1028, 4, dtype=np.uint16)[:50]
    c = np.zeros((2, 1024), np.float32)
    with pytest.raises(sa.SymaccelError):
        sa.aac_pulse(c, rec, swb)            # frame out of range
    rec["frame"], rec["number_pulse"] = 0, 5
    with pytest.raises(sa.SymaccelError):
        sa.aac_pulse(c, rec, swb)            # more than four pulses
    assert lib.dll.symaccel_host_aac_pulse(None, 0, None, 0, None, 0) == 0  # nothing to do


def test_floor0_matches_the_reference_text():
    f = load("vorbis")
    n_cases = 0
    for c in MANIFEST["vorbis"]["cases"]:
        if "Floor0" not in c["fn"]:
            continue
        n_cases += 1
        k = "floor0_%d_" % c["case"]
        b0, b1, order, rate, map_size, amp_bits, amp_off, amp_s, amp_l = (int(v) for v in f[k + "params"])
        assert np.array_equal(sa.vorbis_bark_map(1 << (b0 - 1), rate, map_size), f[k + "map_short"])
        assert np.array_equal(sa.vorbis_bark_map(1 << (b1 - 1), rate, map_size), f[k + "map_long"])
        co = sa.vorbis_floor0_coeffs(f[k + "angles"])
        assert same(co, f[k + "coeffs"])
        for which, amp in (("short", amp_s), ("long", amp_l)):
            got = sa.vorbis_floor0(co, f[k + "map_" + which], map_size, amp_bits, amp_off, amp)
            assert same(got, f[k + "out_" + which]), (c["case"], which)
    assert n_cases >= 4


def test_floor0_reports_the_reference_s_decode_error():
    # p + q == 0: order 0 makes p = q = 1 * ((1 -+ cos) / 2); with cos(omega) = 1 (map value 0) p = 0, q = 1 -- fine; force
    # the error with an odd order whose single coefficient equals 2 cos(omega): q = 0 and p = 1 - cos^2 = 0 at omega = 0.
    co = np.array([2.0], np.float32)
    with pytest.raises(sa.SymaccelError) as e:
        sa.vorbis_floor0(co, np.zeros(8, np.int32), 64, 6, 100, 5)
    assert e.value.status == sa._ffi.ERR_DECODE


def check_status_arrays(ctx, dev, host):
    """The per-block status kernels (ABI v2) against what the reference would have answered; `dev` / `host` move arrays."""
    desc = sa.flac_desc(np.array([0, 1, 1, 2, 2, 2, 3, 2, 2]), np.array([0, 4, 5, 32, 33, 0, 0, 9, 8]), np.array([0, 0, 0, 14, 3, 3, 0, 32, 31]),
                        np.zeros(9))
    st = np.full(9, 99, np.int8)
    st = host(sa.flac_block_status(ctx, dev(desc), 8, dev(st)))
    D, U = sa._ffi.ERR_DECODE, sa._ffi.ERR_UNSUPPORTED
    #           verbatim fixed4 fixed5 lpc32>8 lpc33 lpc0 kind3 order9>8 ok
    assert st.tolist() == [0, 0, D, D, D, D, D, D, 0]
    st2 = np.full(2, 99, np.int8)
    st2 = host(sa.flac_block_status(ctx, dev(sa.flac_desc(np.array([2, 2]), np.array([4, 4]), np.array([32, 15]), np.zeros(2))), 4096, dev(st2)))
    assert st2.tolist() == [U, 0]
    ad = sa.alac_desc(np.array([0, 1, 14, 15]), np.array([4, 4, 4, 4]), np.array([9, 9, 9, 9]), np.array([16, 16, 16, 16]))
    st3 = np.full(4, 99, np.int8)
    st3 = host(sa.alac_block_status(ctx, dev(ad), dev(st3)))
    assert st3.tolist() == [0, D, D, 0]
    ys = np.zeros((5, 65), np.uint32)
    ys[1, 64], ys[2, 0], ys[3, 30], ys[4, 63] = 512, 511, 70000, 255
    st5 = np.full(5, 99, np.int8)
    st5 = host(sa.vorbis_floor1_status(ctx, 65, dev(ys), 5, dev(st5)))
    assert st5.tolist() == [0, U, 0, U, 0]
    st6 = np.full(3, 99, np.int8)
    st6 = host(sa.vorbis_floor1_status(ctx, 7, dev(np.array([[0, 1, 2, 3, 4, 5, 600], [511] * 7, [0, 0, 0, 0, 0, 0, 0]], np.uint32)), 3, dev(st6)))
    assert st6.tolist() == [U, 0, 0]
    filt = np.zeros(5, sa.AAC_TNS_DTYPE)
    filt["frame"], filt["start"], filt["end"], filt["order"] = [0, 9, 0, 0, 1], [0, 0, 8, 0, 4], [16, 16, 8, 1028, 1024], [3, 3, 3, 3, 21]
    filt[0]["order"] = 20
    st4 = np.full(5, 99, np.int8)
    st4 = host(sa.aac_tns_status(ctx, 2, dev(filt), dev(st4)))
    assert st4.tolist() == [0, -1, -1, -1, -1]


def test_status_arrays(emu_ctx):
    check_status_arrays(emu_ctx, lambda a: a, lambda a: a)
