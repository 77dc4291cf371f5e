"""The CPU-baseline builds (bench.py's `cpu_baseline` leg) compute the same bits as the -O2 oracle the parity tests use:
the -O3 -march=native build of the scalar restatement, and the across-chains SIMD schedule of the headline workload
(oracle/cpu_simd.c).  A faster baseline must not be a different computation."""
import numpy as np

import oracle
from helpers import aac_spectra, bit_equal


def test_simd_schedule_matches_the_scalar_oracle():
    rng = np.random.default_rng(0)
    nch, nfr = 32, 5
    coeffs = aac_spectra(rng, (nch, nfr))
    coeffs[3, 2, :40] = np.float32(1e-41)
    delay = rng.standard_normal((nch, 1024)).astype(np.float32)
    side = np.full((nch, nfr), oracle.aac_side(0, 1, 1), np.uint8)
    want_pcm, want_delay = oracle.aac_synth(coeffs, side, delay)
    for native in (False, True):
        pcm, d = oracle.aac_long_kbd_batch_simd(coeffs, delay, native=native)
        assert np.array_equal(pcm.view(np.uint32), want_pcm.view(np.uint32)), native
        assert np.array_equal(d.view(np.uint32), want_delay.view(np.uint32)), native


def test_native_build_matches_the_o2_build():
    rng = np.random.default_rng(1)
    nch, nfr = 2, 4
    coeffs = aac_spectra(rng, (nch, nfr))
    side = np.full((nch, nfr), oracle.aac_side(0, 1, 1), np.uint8)
    xr = rng.standard_normal((2, 3, 576)).astype(np.float32)
    mside = oracle.mp3_side(np.zeros((2, 3)), np.zeros((2, 3)), np.full((2, 3), 576))
    import ctypes as C
    n = oracle.native_lib()
    # AAC and MP3 through the native library's batch entry points
    pcm = np.empty_like(coeffs)
    delay = np.zeros((nch, 1024), np.float32)
    n.so_aac_synth_batch(coeffs.ctypes.data_as(C.c_void_p), side.ctypes.data_as(C.c_void_p), delay.ctypes.data_as(C.c_void_p),
                         pcm.ctypes.data_as(C.c_void_p), C.c_size_t(nch), C.c_size_t(nfr))
    want, wd = oracle.aac_synth(coeffs, side, np.zeros((nch, 1024), np.float32))
    assert bit_equal(pcm, want) and bit_equal(delay, wd)
    out = np.empty_like(xr)
    ov, vv, vf = np.zeros((2, 576), np.float32), np.zeros((2, 1024), np.float32), np.zeros(2, np.int32)
    n.so_mp3_synth_batch(xr.ctypes.data_as(C.c_void_p), mside.ctypes.data_as(C.c_void_p), C.c_int(0), ov.ctypes.data_as(C.c_void_p),
                         vv.ctypes.data_as(C.c_void_p), vf.ctypes.data_as(C.c_void_p), out.ctypes.data_as(C.c_void_p), C.c_size_t(2), C.c_size_t(3))
    w = oracle.mp3_synth(xr, mside, 0, np.zeros((2, 576), np.float32), np.zeros((2, 1024), np.float32), np.zeros(2, np.int32))
    assert bit_equal(out, w[0]) and bit_equal(ov, w[1]) and bit_equal(vv, w[2])
