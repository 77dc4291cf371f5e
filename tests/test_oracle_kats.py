"""Pin the CPU oracle with the reference's OWN known-answer tests, at the reference's
own tolerances (1e-5 absolute against the f64 closed form), and check that every
constant the oracle regenerates by closed form is bit-identical to the reference's
literal (bit patterns extracted into tests/golden/ref_kats.json)."""
import math

import numpy as np
import pytest

import oracle
from helpers import (bits_to_f32, dct32_analytical, dft_naive, imdct12_analytical,
                     imdct36_analytical, imdct_analytical, kats)

K = kats()


def test_verify_imdct_n32():  # mdct.rs:177-201
    x = np.array(K["imdct32_input"], dtype=np.float32)
    scale = math.sqrt(2.0 / 64.0)
    actual = oracle.imdct(x, scale)
    expected = imdct_analytical(x, scale).astype(np.float32)
    assert np.abs(actual.astype(np.float64) - expected.astype(np.float64)).max() < K["imdct_tolerance"]


def test_verify_fft_64():  # dsp/fft/mod.rs:155-186
    v = np.array(K["fft64_input"], dtype=np.float32)
    x = (v[:, 0] + 1j * v[:, 1]).astype(np.complex64)
    expected = dft_naive(x).astype(np.complex64)
    for actual in (oracle.fft(x), oracle.fft_inplace(x)):
        assert np.abs(actual.real - expected.real).max() < K["fft_tolerance"]
        assert np.abs(actual.imag - expected.imag).max() < K["fft_tolerance"]
    assert np.array_equal(oracle.fft(x).view(np.uint32), oracle.fft_inplace(x).view(np.uint32))


def test_verify_imdct36():  # hybrid_synthesis.rs:802-822
    x = np.array(K["mp3_imdct_input18"], dtype=np.float32)
    out, ov = oracle.mp3_imdct36(x, np.ones(36, np.float32), np.zeros(18, np.float32))
    exp = imdct36_analytical(x).astype(np.float32)
    assert np.abs(exp[:18] - out).max() < K["mp3_tolerance"]
    assert np.abs(exp[18:] - ov).max() < K["mp3_tolerance"]


def test_verify_imdct12_win():  # hybrid_synthesis.rs:510-556
    x = np.array(K["mp3_imdct_input18"], dtype=np.float32)
    window = oracle.mp3_imdct_windows()[2]
    out, ov = oracle.mp3_imdct12_win(x, window, np.zeros(18, np.float32))
    exp = np.zeros(36, dtype=np.float32)
    for w in range(3):
        y = imdct12_analytical(x[w::3]).astype(np.float32)
        exp[6 + 6 * w:18 + 6 * w] += y * window[:12]
    assert np.abs(exp[:18] - out).max() < K["mp3_tolerance"]
    assert np.abs(exp[18:] - ov).max() < K["mp3_tolerance"]


def test_verify_dct32():  # synthesis.rs:866-882
    x = np.array(K["mp3_dct32_input"], dtype=np.float32)
    assert np.abs(dct32_analytical(x).astype(np.float32) - oracle.mp3_dct32(x)).max() < K["mp3_tolerance"]


def test_verify_rice_signed_to_i32():  # flac/decoder.rs:646-661
    for word, expected in K["rice_cases"]:
        assert oracle.flac_rice_signed_to_i32(word) == expected


# ---- literal parity (bit-exact) ------------------------------------------------

def _eq_bits(a, bits):
    return np.array_equal(np.ascontiguousarray(a, np.float32).view(np.uint32),
                          np.array(bits, dtype=np.uint32))


def test_fft_small_twiddle_literals():  # no_simd.rs:307-324, 374-383
    L = K["literal_bits"]
    w32 = oracle.fft_small_twiddles(32).view(np.float32).reshape(-1, 2)
    general32 = [k for k in range(16) if k not in (0, 4, 8, 12)]
    assert _eq_bits(w32[general32].ravel(), np.array(L["fft32_general_twiddles"]).ravel())
    w16 = oracle.fft_small_twiddles(16).view(np.float32).reshape(-1, 2)
    assert _eq_bits(w16[[1, 3, 5, 7]].ravel(), np.array(L["fft16_general_twiddles"]).ravel())


def test_mp3_literals():  # hybrid_synthesis.rs:611-630, 668-678, 722-730; synthesis.rs:13-142, 354-396
    L = K["literal_bits"]
    c = oracle.mp3_constants()
    assert _eq_bits(c["dct_iv_scale"], L["dct_iv_scale"])
    assert _eq_bits(np.delete(c["sdct18_scale"], 4), L["sdct18_scale_without_m4"])
    assert c["sdct18_scale"][4] == np.float32(math.sqrt(2.0))
    assert _eq_bits(c["sdct9_d"], L["sdct9_d"])
    for name in ("cos_16", "cos_8", "cos_4", "cos_2", "cos_1"):
        assert _eq_bits(c[name], L[name]), name
    assert _eq_bits(oracle.mp3_synthesis_window(), L["synthesis_d"])


def test_vorbis_floor1_table_literals():  # vorbis floor.rs:21-112
    assert _eq_bits(oracle.vorbis_floor1_table(), K["literal_bits"]["floor1_inverse_db"])
