"""GPU tests for the corners of SURVEY 8(a) the parity suite did not reach (VERDICT r3 "what's weak"), and for row a5.

 * FLAC: block sizes above 4096 up to the frame header's u16 limit (symphonia-bundle-flac/src/frame.rs:58), 32-bit samples
   through every channel assignment (decoder.rs:199-242 with bits_per_sample = 32: no left-justification shift, the
   side channel's arithmetic wraps in i32 exactly like the reference's release build).
 * Vorbis: 32 channels of one stream as 32 chains, with a pcm_stride that is odd.
 * AAC: chains of one, two and three frames through the workgroup walk with segments of one, two and three frames.
 * a5 -- the reference's DEFAULT build runs `Fft` through rustfft (symphonia-core/src/dsp/fft/simd.rs:19-62), whose roundings
   differ from the in-tree path the product reproduces bit for bit.  Against that build the only criterion the reference
   itself states is 1e-5 against the f64 closed forms (dsp/fft/mod.rs:155-249, mdct.rs:177-201): swept here over EVERY
   size `Fft` / `Imdct` accept and over one AAC, one MP3 and one Vorbis chain.  The reference's vectors are O(1) at n = 64
   / N = 32; for the larger sizes the criterion is applied relative to the largest output (1e-5 * max|y|) -- an absolute
   1e-5 is below f32 resolution of the outputs of a 65 536-point transform.
"""
import numpy as np
import pytest

import oracle
from helpers import aac_sequence_chain, aac_spectra

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")


@pytest.fixture(scope="module")
def ctx():
    if not torch.cuda.is_available():
        pytest.fail("no GPU visible: the gpu-marked tests must run on an MI355X (there is no CPU path)")
    from symphonia_amd import Context
    c = Context(0)
    c.use_torch_stream()
    yield c
    c.close()


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def host(t):
    torch.cuda.synchronize()
    return t.cpu().numpy()


def same_bits(got, want):
    """bit-identical, the sign of a zero aside"""
    got, want = np.asarray(got), np.asarray(want)
    return got.shape == want.shape and bool(np.array_equal(got, want)) and not np.isnan(got).any()


# ------------------------------------------------------------------------------------------ FLAC

@pytest.mark.parametrize("blocksize", [4608, 16384, 65535])
def test_flac_block_sizes_above_4096(ctx, blocksize):
    from symphonia_amd import FlacPredictor, flac_desc
    rng = np.random.default_rng(blocksize)
    nb = 66 if blocksize < 60000 else 34
    buf = rng.integers(-(1 << 15), 1 << 15, (nb, blocksize)).astype(np.int32)
    kind = rng.integers(0, 3, nb).astype(np.uint8)
    kind[:4] = [2, 1, 0, 2]
    order = np.where(kind == 1, rng.integers(0, 5, nb), rng.integers(1, 33, nb)).astype(np.uint8)
    order[0], order[3] = 32, 12
    kind[(kind == 2) & (order == 0)] = 0
    shift = rng.integers(8, 15, nb).astype(np.uint8)
    wasted = np.where(rng.random(nb) < 0.2, rng.integers(1, 4, nb), 0).astype(np.uint8)
    # decaying coefficients: the recurrence stays bounded over 65 535 samples (an exploding one only wraps, which both
    # sides do identically, but then every sample is noise from the first few hundred on)
    coeffs = (rng.integers(-(1 << 14), 1 << 14, (nb, 32)) * (0.55 ** np.arange(32))[None, ::-1] * 0.5).astype(np.int32)
    mode = rng.integers(0, 4, nb // 2).astype(np.uint8)
    desc = flac_desc(kind, order, shift, wasted)
    want = oracle.flac_restore(buf, oracle.flac_desc(kind, order, shift, wasted), coeffs)
    d = dev(buf)
    FlacPredictor(ctx).restore(d, dev(desc.view(np.uint8).reshape(nb, 4)), dev(coeffs))
    assert np.array_equal(host(d), want)
    d = dev(buf)
    FlacPredictor(ctx).restore_stereo(d, dev(desc.view(np.uint8).reshape(nb, 4)), dev(coeffs), dev(mode), 8)
    for p in range(nb // 2):
        a, b = oracle.flac_decorrelate(int(mode[p]), want[2 * p], want[2 * p + 1])
        want[2 * p], want[2 * p + 1] = oracle.flac_shl(a, 8), oracle.flac_shl(b, 8)
    assert np.array_equal(host(d), want)


@pytest.mark.parametrize("blocksize", [192, 4096, 5000])
def test_flac_32_bit_samples_every_channel_assignment(ctx, blocksize):
    """bits_per_sample = 32: the predictors run on full-range i32 words, the side channel of left/side, mid/side and
    right/side would need 33 bits and wraps (decoder.rs:32-82 in a release build), `32 - bps` = 0 so nothing is shifted."""
    from symphonia_amd import FlacPredictor, flac_desc
    rng = np.random.default_rng(32 + blocksize)
    nb = 64
    pcm = rng.integers(-(1 << 31), 1 << 31, (nb, blocksize), dtype=np.int64)
    pcm[:, :8] = [(1 << 31) - 1, -(1 << 31), -1, 0, 1, -(1 << 31), (1 << 31) - 1, 12345]
    # the subframes of a 32-bit stream: order-2 / order-1 fixed predictors and verbatim, residuals = wrapped differences
    kind = np.array([0, 1, 1, 2] * (nb // 4), np.uint8)
    order = np.array([0, 2, 1, 3] * (nb // 4), np.uint8)
    shift = np.zeros(nb, np.uint8)
    coeffs = np.zeros((nb, 32), np.int32)
    coeffs[:, 29:] = [1, -3, 3]  # order-3 LPC with shift 0 == the order-3 fixed predictor; reference layout: the first coefficient last
    buf = (((pcm + (1 << 31)) % (1 << 32)) - (1 << 31)).astype(np.int32)
    desc = flac_desc(kind, order, shift, 0 * shift)
    want = oracle.flac_restore(buf, oracle.flac_desc(kind, order, shift, 0 * shift), coeffs)
    mode = np.array([0, 1, 2, 3] * (nb // 8), np.uint8)
    d = dev(buf)
    FlacPredictor(ctx).restore_stereo(d, dev(desc.view(np.uint8).reshape(nb, 4)), dev(coeffs), dev(mode), 0)
    for p in range(nb // 2):
        want[2 * p], want[2 * p + 1] = oracle.flac_decorrelate(int(mode[p]), want[2 * p], want[2 * p + 1])
    assert np.array_equal(host(d), want)
    da, db = dev(buf[0::2]), dev(buf[1::2])
    FlacPredictor(ctx).decorrelate(dev(mode), da, db, blocksize, out_shift=0)
    ga, gb = host(da), host(db)
    for p in range(nb // 2):
        wa, wb = oracle.flac_decorrelate(int(mode[p]), buf[2 * p], buf[2 * p + 1])
        assert np.array_equal(ga[p], wa) and np.array_equal(gb[p], wb)


# ------------------------------------------------------------------------------------------ Vorbis

@pytest.mark.parametrize("bs0e,bs1e", [(8, 11), (7, 10), (9, 13)])
def test_vorbis_32_channels_odd_pcm_stride(ctx, bs0e, bs1e):
    from test_emu_codecs import vorbis_case
    from symphonia_amd import VorbisDsp
    rng = np.random.default_rng(320 + bs0e)
    flags, prev, spectra, overlap, pcm_stride = vorbis_case(rng, bs0e, bs1e, 32, 19)
    flags[:] = flags[:1]  # the channels of one stream share the block flags
    prev[:] = prev[0]
    pcm_stride = int(oracle.vorbis_layout(bs0e, bs1e, flags, prev)[1][:, -1].max())
    pcm_stride += 1 if pcm_stride % 2 == 0 else 2  # odd, and not the exact length
    d_prev, d_ov = dev(prev), dev(overlap)
    ctx.set_segment(5)
    pcm = host(VorbisDsp(ctx, bs0e, bs1e).synth(dev(spectra), dev(flags), d_prev, d_ov, pcm_stride))
    ctx.set_segment(0)
    want = oracle.vorbis_synth(bs0e, bs1e, spectra, flags, prev, overlap, pcm_stride)
    used = int(oracle.vorbis_layout(bs0e, bs1e, flags, prev)[1][0, -1])
    assert same_bits(pcm[:, :used], want[0][:, :used])
    assert same_bits(host(d_ov), want[1]) and np.array_equal(host(d_prev), want[2])


# ------------------------------------------------------------------------------------------ AAC

@pytest.mark.parametrize("nfr", [1, 2, 3])
@pytest.mark.parametrize("seg", [1, 2, 3])
def test_aac_chains_of_one_to_three_frames(ctx, nfr, seg):
    from symphonia_amd import AacDsp
    rng = np.random.default_rng(10 * nfr + seg)
    nch = 7
    coeffs = aac_spectra(rng, (nch, nfr))
    side = np.empty((nch, nfr), np.uint8)
    starts = [(0, 0), (1, 0), (2, 1), (3, 1), (0, 1), (2, 0), (1, 1)]  # every window sequence opens a chain
    for c in range(nch):
        s, sh, pv = aac_sequence_chain(rng, nfr + 4, p_switch=0.6)
        k = next((i for i in range(4) if s[i] == starts[c][0]), 0)
        side[c] = oracle.aac_side(s[k:k + nfr], sh[k:k + nfr], pv[k:k + nfr])
    delay = rng.standard_normal((nch, 1024)).astype(np.float32)
    ctx.set_segment(seg)
    d_delay = dev(delay)
    pcm = host(AacDsp(ctx).synth(dev(coeffs), dev(side), d_delay))
    ctx.set_segment(0)
    wp, wd = oracle.aac_synth(coeffs, side, delay)
    assert same_bits(pcm, wp) and same_bits(host(d_delay), wd)


# ------------------------------------------------------------------------------------------ a5: the 1e-5 criterion

def _dft_bins(x, bins):
    """f64 closed form (dsp/fft/mod.rs:38-64) of the selected output bins."""
    n = x.size
    k = np.arange(n)
    return np.array([(x.astype(np.complex128) * np.exp(-2j * np.pi * ((b * k) % n) / n)).sum() for b in bins])


def _imdct_samples(x, scale, idx):
    """f64 closed form (mdct.rs:154-175) of the selected output samples."""
    n = x.size
    j = np.arange(n)
    return np.array([scale * (x.astype(np.float64) * np.cos(np.pi / (4 * n) * (((2 * i + 1 + n) * (2 * j + 1)) % (8 * n)))).sum() for i in idx])


@pytest.mark.parametrize("e", range(1, 17))
def test_a5_every_fft_size_within_1e5_of_the_closed_form(ctx, e):
    from symphonia_amd import Fft, Ifft
    n = 1 << e
    rng = np.random.default_rng(500 + e)
    x = (rng.standard_normal((3, n)) + 1j * rng.standard_normal((3, n))).astype(np.complex64)
    xd = torch.view_as_real(dev(x)).contiguous()
    yd = torch.empty_like(xd)
    Fft(ctx, n).fft(xd, yd)
    got = host(yd).reshape(3, n, 2)
    got = got[..., 0] + 1j * got[..., 1]
    bins = np.arange(n) if n <= 512 else np.unique(np.concatenate(([0, 1, n // 2, n - 1], rng.integers(0, n, 60))))
    for r in range(3):
        want = _dft_bins(x[r], bins)
        full = np.fft.fft(x[r].astype(np.complex128))
        assert np.abs(full[bins] - want).max() < 1e-9 * max(1.0, np.abs(full).max())  # (the sampled closed form is the DFT)
        assert np.abs(got[r] - full).max() <= 1e-5 * max(1.0, np.abs(full).max()), n
    if n < 32:
        return  # the reference's Ifft runs no butterflies below 32 points (`transform`, no_simd.rs:221-281, has no case for them:
        # its output there is NOT the inverse DFT -- the product reproduces that bit for bit, test_fft_parity / fixtures `ifft_*`)
    zi = xd.clone()
    Ifft(ctx, n).ifft_inplace(zi)
    back = host(zi).reshape(3, n, 2)
    back = back[..., 0] + 1j * back[..., 1]
    full = np.fft.ifft(x.astype(np.complex128), axis=1)
    assert np.abs(back - full).max() <= 1e-5 * max(1.0, np.abs(full).max()), n


@pytest.mark.parametrize("e", range(2, 18))
def test_a5_every_imdct_size_within_1e5_of_the_closed_form(ctx, e):
    from symphonia_amd import Imdct
    n = 1 << e
    rng = np.random.default_rng(600 + e)
    spec = rng.standard_normal((2, n)).astype(np.float32)
    for scale in (1.0 / n, float(np.sqrt(2.0 / (2 * n)))):
        got = host(Imdct(ctx, n, scale).imdct(dev(spec)))
        idx = np.arange(2 * n) if n <= 512 else np.unique(np.concatenate(([0, 1, n - 1, n, 2 * n - 1], rng.integers(0, 2 * n, 60))))
        for r in range(2):
            want = _imdct_samples(spec[r], scale, idx)
            assert np.abs(got[r, idx] - want).max() <= 1e-5 * max(1.0, np.abs(want).max()), (n, scale)


def test_a5_aac_chain_reconstructs_the_signal(ctx):
    """analysis (ISO/IEC 14496-3 windows, all four window sequences, sine windows: power-complementary to f64 accuracy) ->
    the GPU's Dsp::synth must give 0.25 x back: the criterion that holds whichever FFT the build uses."""
    from test_oracle_unpinned import _aac_analysis
    from symphonia_amd import AacDsp
    rng = np.random.default_rng(15)
    nfr = 28
    seq, _, _ = aac_sequence_chain(rng, nfr, p_switch=0.5)
    z = np.zeros(nfr, np.uint8)
    x = rng.standard_normal(1024 * (nfr + 1))
    x[:1024] = 0.0
    coeffs = _aac_analysis(x, seq, z, z).astype(np.float32)[None]
    pcm = host(AacDsp(ctx).synth(dev(coeffs), dev(oracle.aac_side(seq, z, z)[None]), dev(np.zeros((1, 1024), np.float32))))
    assert set(seq.tolist()) == {0, 1, 2, 3}
    got = pcm[0].reshape(-1)[1024:]
    assert np.abs(got - 0.25 * x[1024:1024 * nfr]).max() < 1e-5  # absolute, on an O(1) signal: the reference's own criterion


def test_a5_mp3_chain_against_the_iso_model(ctx):
    """ISO 11172-3 in f64 (alias reduction, IMDCT-36 / 3 x IMDCT-12 + windows + overlap, frequency inversion, the
    matrixing + 512-tap window of figure A.2) against the GPU's granule chain, every block type."""
    from test_oracle_unpinned import _iso_polyphase, _np_hybrid
    from symphonia_amd import Mp3Synthesis, mp3_side
    rng = np.random.default_rng(16)
    bt = np.array([0, 0, 1, 2, 2, 3, 0, 1, 2, 3, 0, 0], np.uint8)
    mx = np.array([0, 0, 0, 0, 1, 0, 0, 0, 0, 0, 0, 0], np.uint8)
    ngr = bt.size
    rz = np.array([576, 400, 576, 576, 342, 576, 36, 576, 300, 576, 0, 576])
    xr = rng.standard_normal((1, ngr, 576)).astype(np.float32)
    for g in range(ngr):
        xr[0, g, rz[g]:] = 0.0
    side = mp3_side(bt[None], mx[None], rz[None])
    st = [dev(np.zeros((1, 576), np.float32)), dev(np.zeros((1, 1024), np.float32)), dev(np.zeros(1, np.int32))]
    pcm = host(Mp3Synthesis(ctx, 0).synth(dev(xr), dev(side.view(np.uint8).reshape(1, ngr, 4)), *st))
    ov = np.zeros(576)
    slots = []
    for g in range(ngr):
        out, ov = _np_hybrid(xr[0, g], ov, int(bt[g]), int(mx[g]), int(rz[g]), 0)
        slots.append(out.reshape(32, 18).T)  # [time slot][sub-band]
    want = _iso_polyphase(np.concatenate(slots)).reshape(ngr, 576)
    assert np.abs(pcm[0] - want).max() <= 1e-5 * max(1.0, np.abs(want).max())


def test_a5_vorbis_chain_reconstructs_the_signal(ctx):
    from test_oracle_unpinned import _vorbis_analysis
    from symphonia_amd import VorbisDsp
    rng = np.random.default_rng(17)
    bs0e, bs1e = 8, 11
    bs0, bs1 = 1 << bs0e, 1 << bs1e
    flags = [1, 1, 0, 0, 1, 0, 1, 1, 0, 0, 0, 1, 1]
    x = rng.standard_normal(bs1 * (len(flags) + 2))
    spectra = _vorbis_analysis(x, flags, bs0, bs1)
    packed = np.concatenate([s / (len(s) / 2) for s in spectra]).astype(np.float32)[None]
    bf = np.array(flags, np.uint8)[None]
    _, pcm_off = oracle.vorbis_layout(bs0e, bs1e, bf, np.array([-1]))
    pcm = host(VorbisDsp(ctx, bs0e, bs1e).synth(dev(packed), dev(bf), dev(np.array([-1], np.int32)), dev(np.zeros((1, bs1 // 2), np.float32)),
                                                int(pcm_off[0, -1])))
    centre = bs1 // 2
    for b in range(1, len(flags)):
        ln = ((bs1 if flags[b - 1] else bs0) + (bs1 if flags[b] else bs0)) // 4
        got = pcm[0, pcm_off[0, b]:pcm_off[0, b + 1]]
        assert np.abs(got - x[centre:centre + ln]).max() < 1e-5, b  # absolute, on an O(1) signal
        centre += ln
