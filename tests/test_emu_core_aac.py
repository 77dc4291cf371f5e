"""Kernel LOGIC check without a GPU: the product's .hip sources compiled by g++ against the
stand-in HIP runtime (tests/emu), driven through the real C ABI and ctypes binding, compared with
the oracle bit for bit.  (The GPU parity tests proper are the `-m gpu` tests.)"""
import numpy as np
import pytest

import oracle
from emu_lib import emu_ctx  # noqa: F401
from helpers import aac_sequence_chain, aac_spectra, bit_equal
from symphonia_amd import AacDsp, Fft, Imdct, aac_side


@pytest.mark.parametrize("n", [4, 8, 16, 32, 64, 128, 256, 512, 1024, 2048, 4096])
def test_emu_imdct_bit_exact(emu_ctx, n):
    rng = np.random.default_rng(n)
    count = 5 if n <= 1024 else 2
    spec = (rng.standard_normal((count, n)) * np.exp2(rng.integers(-6, 8, (count, n)))).astype(np.float32)
    for scale in (1.0, 1.0 / (2 * n), -1.0 / 3):
        got = Imdct(emu_ctx, n, scale).imdct(spec)
        assert bit_equal(got, oracle.imdct(spec, scale)), (n, scale)


@pytest.mark.parametrize("n", [2, 4, 8, 16, 32, 64, 128, 256, 512, 1024, 2048, 4096])
def test_emu_fft_bit_exact(emu_ctx, n):
    rng = np.random.default_rng(100 + n)
    count = 3
    x = (rng.standard_normal((count, n)) + 1j * rng.standard_normal((count, n))).astype(np.complex64)
    y = np.empty_like(x)
    Fft(emu_ctx, n).fft(x, y)  # the emulated "device" memory is host memory
    want = np.stack([oracle.fft(v) for v in x])
    assert bit_equal(y, want)
    z = x.copy()
    Fft(emu_ctx, n).fft_inplace(z)
    assert bit_equal(z, want)


@pytest.mark.parametrize("n", [1024, 2048, 4096])
def test_emu_big_register_pass_kernels(emu_ctx, n):
    """1024 / 2048 points: two / four 512-point sub-transforms per wavefront, the last stages in registers; 4096 points: the four
    wavefronts of a workgroup take a quarter each and meet in LDS for the last two stages; Fft, Ifft (in place), Imdct of 2 n
    lines, several transforms per wavefront / workgroup."""
    from symphonia_amd import Ifft
    rng = np.random.default_rng(400 + n)
    for count in (1, 3, 9):
        x = (rng.standard_normal((count, n)) * np.exp2(rng.integers(-6, 7, (count, n))) + 1j * rng.standard_normal((count, n))).astype(np.complex64)
        y = np.empty_like(x)
        Fft(emu_ctx, n).fft(x, y)
        assert bit_equal(y, np.stack([oracle.fft(v) for v in x])), (n, count)
        z = x.copy()
        Ifft(emu_ctx, n).ifft_inplace(z)
        assert bit_equal(z, np.stack([oracle.ifft(v) for v in x])), (n, count)
        spec = (rng.standard_normal((count, 2 * n)) * np.exp2(rng.integers(-6, 8, (count, 2 * n)))).astype(np.float32)
        assert bit_equal(Imdct(emu_ctx, 2 * n, -1.0 / 3).imdct(spec), oracle.imdct(spec, -1.0 / 3)), (n, count)


@pytest.mark.parametrize("n", [16, 32, 64, 128, 256, 512])
def test_emu_register_pass_kernels_over_several_groups(emu_ctx, n):
    """Transforms of 16 .. 512 points run 512 / n at a time per wavefront pass (fft_wave_multi): whole groups, a ragged last
    group, several groups per wavefront, Fft / Ifft in and out of place, Imdct of 2 n lines."""
    from symphonia_amd import Ifft
    rng = np.random.default_rng(300 + n)
    per_group = 512 // n
    for count in (1, per_group, per_group + 1, 5 * per_group + max(1, per_group // 2)):
        x = (rng.standard_normal((count, n)) * np.exp2(rng.integers(-6, 7, (count, n))) + 1j * rng.standard_normal((count, n))).astype(np.complex64)
        y = np.empty_like(x)
        Fft(emu_ctx, n).fft(x, y)
        assert bit_equal(y, np.stack([oracle.fft(v) for v in x])), (n, count)
        z = x.copy()
        Ifft(emu_ctx, n).ifft_inplace(z)
        assert bit_equal(z, np.stack([oracle.ifft(v) for v in x])), (n, count)
        spec = (rng.standard_normal((count, 2 * n)) * np.exp2(rng.integers(-6, 8, (count, 2 * n)))).astype(np.float32)
        if 2 * n not in (128, 1024):  # (those two sizes have kernels of their own)
            assert bit_equal(Imdct(emu_ctx, 2 * n, 1.0 / n).imdct(spec), oracle.imdct(spec, 1.0 / n)), (n, count)


@pytest.mark.parametrize("n", [8192, 32768])
def test_emu_large_fft_and_ifft(emu_ctx, n):
    """More than 4096 points (legal in the reference up to 65 536, used by none of its codecs): 4096-point tiles in LDS,
    then the remaining merge stages in global memory -- the same butterflies, in and out of place."""
    from symphonia_amd import Ifft
    rng = np.random.default_rng(200 + n)
    x = (rng.standard_normal((2, n)) * np.exp2(rng.integers(-4, 5, (2, n))) + 1j * rng.standard_normal((2, n))).astype(np.complex64)
    y = np.empty_like(x)
    Fft(emu_ctx, n).fft(x, y)
    assert bit_equal(y, np.stack([oracle.fft(v) for v in x]))
    z = x.copy()
    Fft(emu_ctx, n).fft_inplace(z)
    assert bit_equal(z, y)
    Ifft(emu_ctx, n).ifft(x, y)
    assert bit_equal(y, np.stack([oracle.ifft(v) for v in x]))
    z = x.copy()
    Ifft(emu_ctx, n).ifft_inplace(z)
    assert bit_equal(z, y)


@pytest.mark.parametrize("n", [16384, 65536])
def test_emu_large_imdct(emu_ctx, n):
    rng = np.random.default_rng(300 + n)
    spec = (rng.standard_normal((2, n)) * np.exp2(rng.integers(-6, 8, (2, n)))).astype(np.float32)
    for scale in (1.0, -1.0 / 3):
        assert bit_equal(Imdct(emu_ctx, n, scale).imdct(spec), oracle.imdct(spec, scale)), (n, scale)


def test_emu_aac_only_long(emu_ctx):
    rng = np.random.default_rng(1)
    coeffs = aac_spectra(rng, (2, 5))
    side = np.full((2, 5), aac_side(0, 1, 1), np.uint8)
    delay = rng.standard_normal((2, 1024)).astype(np.float32)
    emu_ctx.set_segment(2)  # forces halo recompute at frames 2 and 4
    pcm, nd = AacDsp(emu_ctx).synth(coeffs, side, delay)
    wp, wd = oracle.aac_synth(coeffs, side, delay)
    assert bit_equal(pcm, wp)
    assert bit_equal(nd, wd)


@pytest.mark.parametrize("seg", [1, 3, 64])
def test_emu_aac_all_sequences(emu_ctx, seg):
    rng = np.random.default_rng(2 + seg)
    nch, nfr = 2, 9
    coeffs = aac_spectra(rng, (nch, nfr))
    side = np.empty((nch, nfr), np.uint8)
    for c in range(nch):
        s, sh, pv = aac_sequence_chain(rng, nfr, p_switch=0.5)
        side[c] = aac_side(s, sh, pv)
    delay = rng.standard_normal((nch, 1024)).astype(np.float32)
    emu_ctx.set_segment(seg)
    pcm, nd = AacDsp(emu_ctx).synth(coeffs, side, delay)
    wp, wd = oracle.aac_synth(coeffs, side, delay)
    assert set((side & 3).ravel().tolist()) >= {0, 1, 2, 3} or seg != 3
    assert bit_equal(pcm, wp)
    assert bit_equal(nd, wd)
    emu_ctx.set_segment(0)


def test_emu_aac_non_finite_lines(emu_ctx):
    """+-Inf / NaN spectral lines (a corrupt stream): the frame and the delay line it feeds turn NaN exactly where the
    reference's operation graph makes them NaN, the frames after recover, and the halo recompute walks through it."""
    from helpers import equal_mod_nan, sprinkle_specials
    rng = np.random.default_rng(404)
    nch, nfr = 2, 10
    coeffs = aac_spectra(rng, (nch, nfr))
    sprinkle_specials(coeffs, rng, [2, 5, nfr + 3, nfr + 4])
    side = np.empty((nch, nfr), np.uint8)
    for c in range(nch):
        s, sh, pv = aac_sequence_chain(rng, nfr, p_switch=0.5)
        side[c] = aac_side(s, sh, pv)
    delay = rng.standard_normal((nch, 1024)).astype(np.float32)
    wp, wd = oracle.aac_synth(coeffs, side, delay)
    assert np.isnan(wp).any() and np.isfinite(wp[0, 8]).all()
    for seg in (3, 64):
        emu_ctx.set_segment(seg)
        pcm, nd = AacDsp(emu_ctx).synth(coeffs, side, delay)
        assert equal_mod_nan(pcm, wp) and equal_mod_nan(nd, wd), seg
    emu_ctx.set_segment(0)
