"""GPU parity tests (run on a real MI355X: `pytest -m gpu`).  Every call goes through the C ABI of
the hipcc-built symphonia_amd/libsymaccel.so; the oracle is only the checker.

Bar (BASELINE.json north_star): PCM within +-1 ULP f32 of the reference CPU path, bit-exact for the
FLAC integer path.  The kernels reproduce the reference's operation DAG, so we assert the stronger
property -- bit-exact (signed zeros and NaN payloads aside) -- and report the ULP bound as the
contract.  Full BASELINE sizes are checked through size-independent properties (segmentation
invariance, sampled-chain parity against the oracle, analysis->synthesis reconstruction)."""
import numpy as np
import pytest

import oracle
from helpers import aac_sequence_chain, aac_spectra, assert_ulp, mdct_forward, ulp_diff

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


def assert_parity(got, want, what):
    """Contract: <= 1 ULP.  Expectation: identical bits (modulo the sign of zero)."""
    assert_ulp(got, want, 1, what)
    d = ulp_diff(got, want)
    assert int(d.max()) == 0, "%s: within 1 ULP but not bit-identical (%d elements differ)" % (what, int((d > 0).sum()))


# ------------------------------------------------------------------------------------------ core

@pytest.mark.parametrize("n", [4, 16, 32, 64, 128, 256, 1024, 2048, 4096, 8192, 16384, 131072])
def test_imdct_parity(ctx, n):
    from symphonia_amd import Imdct
    rng = np.random.default_rng(n)
    count = 37 if n <= 2048 else (5 if n <= 8192 else 3)
    spec = (rng.standard_normal((count, n)) * np.exp2(rng.integers(-10, 12, (count, n)))).astype(np.float32)
    spec[0, : n // 2] = np.float32(1e-41)  # denormals must survive (no flush-to-zero)
    spec[1, ::3] = -0.0
    for scale in (1.0, 1.0 / (2 * n), -2.0):
        got = host(Imdct(ctx, n, scale).imdct(dev(spec)))
        assert_parity(got, oracle.imdct(spec, scale), "imdct n=%d scale=%g" % (n, scale))


def test_imdct_reference_kat_on_gpu(ctx):
    """mdct.rs:177-201 (N=32 ramp vs the f64 closed form, 1e-5) through the GPU path."""
    import math
    from helpers import imdct_analytical, kats
    from symphonia_amd import Imdct
    x = np.array(kats()["imdct32_input"], dtype=np.float32)
    scale = math.sqrt(2.0 / 64.0)
    got = host(Imdct(ctx, 32, scale).imdct(dev(x[None])))[0]
    assert np.abs(got - imdct_analytical(x, scale)).max() < 1e-5


def test_config1_imdct1024_one_frame_host_path(ctx):
    """BASELINE config 1: Imdct::new_scaled(1024, 1/2048), one frame, through the host-pointer entry point
    (symaccel_imdct_f32: stage to HBM, transform, copy back): bit-identical to the oracle, 1e-5 from the closed form."""
    from helpers import imdct_analytical
    from symphonia_amd import Imdct
    rng = np.random.default_rng(0)
    spec = rng.standard_normal((1, 1024)).astype(np.float32)
    got = Imdct(ctx, 1024, 1.0 / 2048.0).imdct(spec)  # numpy in -> numpy out
    assert got.shape == (1, 2048)
    assert_parity(got, oracle.imdct(spec, 1.0 / 2048.0), "config 1")
    ref = imdct_analytical(spec[0], 1.0 / 2048.0)
    assert np.abs(got[0] - ref).max() <= 1e-5 * max(1.0, np.abs(ref).max())


@pytest.mark.parametrize("n", [2, 8, 16, 32, 64, 512, 1024, 4096, 8192, 65536])
def test_fft_parity(ctx, n):
    from helpers import dft_naive, kats
    from symphonia_amd import Fft
    rng = np.random.default_rng(n + 1)
    x = (rng.standard_normal((9, n)) + 1j * rng.standard_normal((9, n))).astype(np.complex64)
    if n == 64:  # dsp/fft/mod.rs:88-186
        v = np.array(kats()["fft64_input"], dtype=np.float32)
        x[0] = v[:, 0] + 1j * v[:, 1]
    xd = torch.view_as_real(dev(x)).contiguous()
    guard = torch.full((2 * xd.numel(),), 12345.0, device="cuda")  # the output sits in front of a sentinel region
    yd = guard[: xd.numel()].view(xd.shape)
    Fft(ctx, n).fft(xd, yd)
    torch.cuda.synchronize()
    assert bool((guard[xd.numel():] == 12345.0).all()), "Fft wrote past the end of its output"
    got = host(yd).view(np.complex64).reshape(9, n)
    want = np.stack([oracle.fft(r) for r in x])
    assert_parity(got.view(np.float32), want.view(np.float32), "fft n=%d" % n)
    if n == 64:
        assert np.abs(got[0] - dft_naive(x[0])).max() < 1e-5
    xi = xd.clone()
    Fft(ctx, n).fft_inplace(xd)
    assert_parity(host(xd).view(np.complex64).reshape(9, n).view(np.float32), want.view(np.float32), "fft_inplace")
    from symphonia_amd import Ifft
    Ifft(ctx, n).ifft_inplace(xi)  # (in place: the large sizes go through a scratch copy)
    got = host(xi).view(np.complex64).reshape(9, n)
    want = np.stack([oracle.ifft(r) for r in x])
    assert_parity(got.view(np.float32), want.view(np.float32), "ifft n=%d" % n)


# ------------------------------------------------------------------------------------------ AAC

def aac_case(seed, nch, nfr, only_long=False):
    rng = np.random.default_rng(seed)
    coeffs = aac_spectra(rng, (nch, nfr))
    side = np.empty((nch, nfr), np.uint8)
    for c in range(nch):
        if only_long:
            side[c] = oracle.aac_side(0, 1, 1)
        else:
            s, sh, pv = aac_sequence_chain(rng, nfr, p_switch=0.35)
            side[c] = oracle.aac_side(s, sh, pv)
    delay = rng.standard_normal((nch, 1024)).astype(np.float32)
    return coeffs, side, delay


@pytest.mark.parametrize("seg", [1, 5, 32, 1000])
@pytest.mark.parametrize("only_long", [True, False])
def test_aac_parity(ctx, seg, only_long):
    from symphonia_amd import AacDsp
    coeffs, side, delay = aac_case(10 + seg, 6, 41, only_long)
    if not only_long:
        assert set((side & 3).ravel().tolist()) == {0, 1, 2, 3}
    ctx.set_segment(seg)
    d_delay = dev(delay)
    pcm = host(AacDsp(ctx).synth(dev(coeffs), dev(side), d_delay))
    ctx.set_segment(0)
    wp, wd = oracle.aac_synth(coeffs, side, delay)
    assert_parity(pcm, wp, "aac pcm")
    assert_parity(host(d_delay), wd, "aac delay")


@pytest.mark.parametrize("seg", [3, 64])
def test_aac_non_finite_lines(ctx, seg):
    """+-Inf / NaN spectral lines: NaN exactly where the reference's operation graph makes it, everything else bit-equal."""
    from helpers import equal_mod_nan, sprinkle_specials
    from symphonia_amd import AacDsp
    coeffs, side, delay = aac_case(77, 4, 21, False)
    sprinkle_specials(coeffs, np.random.default_rng(78), [2, 9, 21 + 3, 21 + 4, 3 * 21 + 20])
    wp, wd = oracle.aac_synth(coeffs, side, delay)
    assert np.isnan(wp).any() and np.isnan(wd[3]).any() and np.isfinite(wp[0, 15]).all()
    ctx.set_segment(seg)
    d_delay = dev(delay)
    pcm = host(AacDsp(ctx).synth(dev(coeffs), dev(side), d_delay))
    ctx.set_segment(0)
    assert equal_mod_nan(pcm, wp) and equal_mod_nan(host(d_delay), wd)


def test_aac_streaming_state(ctx):
    """Two consecutive calls continue the stream exactly like one call (delay_io contract)."""
    from symphonia_amd import AacDsp
    coeffs, side, delay = aac_case(3, 3, 20)
    d = dev(delay)
    a = host(AacDsp(ctx).synth(dev(coeffs[:, :9]), dev(side[:, :9]), d))
    b = host(AacDsp(ctx).synth(dev(coeffs[:, 9:]), dev(side[:, 9:]), d))
    wp, wd = oracle.aac_synth(coeffs, side, delay)
    assert_parity(np.concatenate((a, b), axis=1), wp, "aac two calls")
    assert_parity(host(d), wd, "aac delay after two calls")


def test_aac_config2_full_size_properties(ctx):
    """BASELINE config 2: 65 536 stereo long-block frames = 128 chains x 1024 frames (1 GiB of traffic)."""
    from symphonia_amd import AacDsp
    nch, nfr = 128, 1024
    g = torch.Generator(device="cuda").manual_seed(2)
    coeffs = torch.randn((nch, nfr, 1024), generator=g, device="cuda", dtype=torch.float32)
    coeffs *= torch.exp2(torch.randint(-8, 13, (nch, nfr, 1), generator=g, device="cuda").float())
    coeffs[:, :, 672:] = 0.0
    side = torch.full((nch, nfr), int(oracle.aac_side(0, 1, 1)), dtype=torch.uint8, device="cuda")
    delay0 = torch.randn((nch, 1024), generator=g, device="cuda", dtype=torch.float32)
    outs = []
    for seg in (16, 64):  # segmentation (halo recompute) must not change a single bit
        ctx.set_segment(seg)
        d = delay0.clone()
        outs.append((AacDsp(ctx).synth(coeffs, side, d), d))
    ctx.set_segment(0)
    torch.cuda.synchronize()
    assert torch.equal(outs[0][0].view(torch.int32), outs[1][0].view(torch.int32))
    assert torch.equal(outs[0][1].view(torch.int32), outs[1][1].view(torch.int32))
    # sampled chains against the oracle (the oracle finishes 6 x 1024 frames in seconds)
    pick = [0, 1, 63, 64, 126, 127]
    wp, wd = oracle.aac_synth(host(coeffs[pick]), host(side[pick]), host(delay0[pick]))
    assert_parity(host(outs[0][0][pick]), wp, "config-2 sampled chains")
    assert_parity(host(outs[0][1][pick]), wd, "config-2 sampled delay")


def test_aac_tdac_reconstruction_on_gpu(ctx):
    """analysis (sine windows, long blocks) -> GPU Dsp::synth == 0.25 * x."""
    from symphonia_amd import AacDsp
    rng = np.random.default_rng(5)
    nfr = 12
    x = rng.standard_normal(1024 * (nfr + 1))
    x[:1024] = 0
    w = oracle.aac_window(False, 0.0, 1024).astype(np.float64)
    win = np.concatenate((w, w[::-1]))
    coeffs = np.stack([mdct_forward(x[t * 1024:t * 1024 + 2048] * win) for t in range(nfr)]).astype(np.float32)[None]
    side = np.zeros((1, nfr), np.uint8)
    pcm = host(AacDsp(ctx).synth(dev(coeffs), dev(side), dev(np.zeros((1, 1024), np.float32))))
    assert np.abs(pcm[0].reshape(-1)[1024:] - 0.25 * x[1024:1024 * nfr]).max() < 3e-5 * np.abs(x).max()


# ------------------------------------------------------------------------------------------ MP3

def mp3_case(seed, nch, ngr):
    from test_emu_codecs import mp3_case as gen
    rng = np.random.default_rng(seed)
    xr, bt, mx, rz = gen(rng, nch, ngr)
    ov = rng.standard_normal((nch, 576)).astype(np.float32)
    warm = rng.standard_normal((nch, 2, 576)).astype(np.float32)
    z = np.zeros((nch, 2))
    _, _, vv, vf = oracle.mp3_synth(warm, oracle.mp3_side(z, z, z + 576), 0, np.zeros((nch, 576), np.float32),
                                    np.zeros((nch, 1024), np.float32), rng.integers(0, 16, nch).astype(np.int32))
    return xr, bt, mx, rz, ov, vv, vf


@pytest.mark.parametrize("seg,sr", [(2, 0), (7, 1), (32, 8), (1000, 3)])
def test_mp3_parity(ctx, seg, sr):
    from symphonia_amd import Mp3Synthesis, mp3_side
    xr, bt, mx, rz, ov, vv, vf = mp3_case(seg + sr, 5, 37)
    side = mp3_side(bt, mx, rz)
    d_ov, d_vv, d_vf = dev(ov), dev(vv), dev(vf)
    ctx.set_segment(seg)
    pcm = host(Mp3Synthesis(ctx, sr).synth(dev(xr), dev(side.view(np.uint8).reshape(5, 37, 4)), d_ov, d_vv, d_vf))
    ctx.set_segment(0)
    want = oracle.mp3_synth(xr, oracle.mp3_side(bt, mx, rz), sr, ov, vv, vf)
    assert set(bt.ravel().tolist()) == {0, 1, 2, 3}
    assert_parity(pcm, want[0], "mp3 pcm")
    assert_parity(host(d_ov), want[1], "mp3 overlap")
    assert_parity(host(d_vv), want[2], "mp3 v_vec")
    assert np.array_equal(host(d_vf), want[3])


@pytest.mark.parametrize("seg", [2, 5, 64])
def test_mp3_non_finite_lines(ctx, seg):
    from helpers import equal_mod_nan, sprinkle_specials
    from symphonia_amd import Mp3Synthesis, mp3_side
    xr, bt, mx, rz, ov, vv, vf = mp3_case(90, 5, 23)
    sprinkle_specials(xr, np.random.default_rng(91), [1, 7, 23 + 2, 2 * 23 + 22, 4 * 23 + 11])
    want = oracle.mp3_synth(xr, oracle.mp3_side(bt, mx, rz), 0, ov, vv, vf)
    assert np.isnan(want[0]).any() and np.isnan(want[2]).any()
    side = mp3_side(bt, mx, rz)
    d_ov, d_vv, d_vf = dev(ov), dev(vv), dev(vf)
    ctx.set_segment(seg)
    pcm = host(Mp3Synthesis(ctx, 0).synth(dev(xr), dev(side.view(np.uint8).reshape(5, 23, 4)), d_ov, d_vv, d_vf))
    ctx.set_segment(0)
    assert equal_mod_nan(pcm, want[0]) and equal_mod_nan(host(d_ov), want[1]) and equal_mod_nan(host(d_vv), want[2])
    assert np.array_equal(host(d_vf), want[3])


def test_mp3_config3_sampled(ctx):
    """BASELINE config 3: 131 072 stereo granules = 128 chains x 2048 granules, long blocks.  Segmentation (halo
    recompute) must not change a bit; sampled chains are checked against the oracle."""
    from symphonia_amd import Mp3Synthesis, mp3_side
    nch, ngr = 128, 2048
    g = torch.Generator(device="cuda").manual_seed(3)
    d_xr = torch.randn((nch, ngr, 576), generator=g, device="cuda", dtype=torch.float32) * 0.1
    side = mp3_side(np.zeros((nch, ngr)), np.zeros((nch, ngr)), np.full((nch, ngr), 576))
    d_side = dev(side.view(np.uint8).reshape(nch, ngr, 4))
    outs = []
    for seg in (16, 128):
        ctx.set_segment(seg)
        st = [dev(np.zeros((nch, 576), np.float32)), dev(np.zeros((nch, 1024), np.float32)), dev(np.zeros(nch, np.int32))]
        outs.append(Mp3Synthesis(ctx, 0).synth(d_xr, d_side, *st))
    ctx.set_segment(0)
    torch.cuda.synchronize()
    assert torch.equal(outs[0].view(torch.int32), outs[1].view(torch.int32))
    pick = [0, 77, 127]
    xr = host(d_xr[pick])
    z = (np.zeros((3, 576), np.float32), np.zeros((3, 1024), np.float32), np.zeros(3, np.int32))
    want = oracle.mp3_synth(xr, oracle.mp3_side(np.zeros((3, ngr)), np.zeros((3, ngr)), np.full((3, ngr), 576)), 0, *z)
    assert_parity(host(outs[0][pick]), want[0], "mp3 config-3 sampled chains")


# ------------------------------------------------------------------------------------------ Vorbis

@pytest.mark.parametrize("bs0e,bs1e,seg", [(8, 11, 5), (6, 6, 2), (6, 13, 3), (7, 12, 1000), (11, 11, 4), (12, 13, 3), (12, 13, 1000)])
def test_vorbis_parity(ctx, bs0e, bs1e, seg):
    from test_emu_codecs import vorbis_case
    from symphonia_amd import VorbisDsp
    rng = np.random.default_rng(bs0e * 31 + bs1e)
    flags, prev, spectra, overlap, pcm_stride = vorbis_case(rng, bs0e, bs1e, 4, 23)
    d_prev, d_ov = dev(prev), dev(overlap)
    ctx.set_segment(seg)
    pcm = host(VorbisDsp(ctx, bs0e, bs1e).synth(dev(spectra), dev(flags), d_prev, d_ov, pcm_stride))
    ctx.set_segment(0)
    want = oracle.vorbis_synth(bs0e, bs1e, spectra, flags, prev, overlap, pcm_stride)
    assert_parity(pcm, want[0], "vorbis pcm")
    assert_parity(host(d_ov), want[1], "vorbis overlap")
    assert np.array_equal(host(d_prev), want[2])


@pytest.mark.parametrize("seed,nb,p_long,tail_short,seg", [(11, 67, 0.75, 0, 32), (12, 40, 0.3, 13, 7), (13, 33, 0.95, 2, 1),
                                                            (14, 29, 0.0, 0, 5), (15, 31, 1.0, 0, 1000), (16, 50, 0.5, 20, 16),
                                                            (17, 300, 0.7, 0, 1000), (18, 200, 0.1, 0, 150)])
def test_vorbis_wave_paths(ctx, seed, nb, p_long, tail_short, seg):
    """The 256/2048 wavefront kernel: transitions, short runs > 8, segment halos, the stale-state fix-up."""
    from test_emu_codecs import vorbis_wave_case
    from symphonia_amd import VorbisDsp
    flags, prev, spectra, overlap, pcm_stride = vorbis_wave_case(seed, 9, nb, p_long, tail_short)
    d_prev, d_ov = dev(prev), dev(overlap)
    ctx.set_segment(seg)
    pcm = host(VorbisDsp(ctx, 8, 11).synth(dev(spectra), dev(flags), d_prev, d_ov, pcm_stride))
    ctx.set_segment(0)
    want = oracle.vorbis_synth(8, 11, spectra, flags, prev, overlap, pcm_stride)
    assert_parity(pcm, want[0], "vorbis pcm")
    assert_parity(host(d_ov), want[1], "vorbis overlap")
    assert np.array_equal(host(d_prev), want[2])


@pytest.mark.parametrize("bs0e,bs1e,nb,p_long", [(10, 13, 36, 0.7), (6, 13, 60, 0.4), (13, 13, 20, 1.0), (9, 12, 48, 0.7), (12, 13, 40, 0.5)])
def test_vorbis_big_blocks_under_load(ctx, bs0e, bs1e, nb, p_long):
    """Long blocks of 8192 samples (the workgroup-cooperative kernel: four wavefronts per block, LDS exchange between them) and 4096
    samples with the whole chip busy: 256 chains = 4 distinct chains x 64 copies, several segments each.  Every copy must equal its
    original bit for bit (chains are independent: anything else is a race or a stray address), the originals are checked against
    the oracle."""
    from symphonia_amd import VorbisDsp
    rng = np.random.default_rng(17 * bs0e + bs1e)
    kinds, copies = 4, 64
    flags0 = (rng.random((kinds, nb)) < p_long).astype(np.uint8)
    flags0[0, : nb // 3] = 0   # a long run of short blocks (several groups at once)
    flags0[1, nb // 2:] = 1    # ... and of long ones
    prev0 = rng.integers(-1, 2, kinds).astype(np.int32)
    lay = oracle.vorbis_layout(bs0e, bs1e, flags0, prev0)
    spec_stride, pcm_stride = int(lay[0][:, -1].max()), int(lay[1][:, -1].max())
    spec_stride += (-spec_stride) % 4
    pcm_stride += (-pcm_stride) % 4
    spectra0 = (rng.standard_normal((kinds, spec_stride)) * 0.25).astype(np.float32)
    overlap0 = rng.standard_normal((kinds, (1 << bs1e) // 2)).astype(np.float32)
    rep = lambda a: np.ascontiguousarray(np.tile(a, (copies,) + (1,) * (a.ndim - 1)))  # chain c = original c % kinds
    d_prev, d_ov = dev(rep(prev0)), dev(rep(overlap0))
    ctx.set_segment(7)
    pcm = VorbisDsp(ctx, bs0e, bs1e).synth(dev(rep(spectra0)), dev(rep(flags0)), d_prev, d_ov, pcm_stride)
    ctx.set_segment(0)
    torch.cuda.synchronize()
    pcm_k = pcm.view(copies, kinds, -1)
    ov_k = d_ov.view(copies, kinds, -1)
    assert torch.equal(pcm_k.view(torch.int32), pcm_k[:1].expand_as(pcm_k).contiguous().view(torch.int32)), "copies of a chain differ"
    assert torch.equal(ov_k.view(torch.int32), ov_k[:1].expand_as(ov_k).contiguous().view(torch.int32)), "copies of a chain's state differ"
    want = oracle.vorbis_synth(bs0e, bs1e, spectra0, flags0, prev0, overlap0, pcm_stride)
    assert_parity(host(pcm[:kinds]), want[0], "vorbis pcm")
    assert_parity(host(d_ov[:kinds]), want[1], "vorbis overlap")
    assert np.array_equal(host(d_prev[:kinds]), want[2])


@pytest.mark.parametrize("n", [4096, 8192])
def test_workgroup_transforms_under_load(ctx, n):
    """Fft of 4096 points / Imdct of 8192 lines (four wavefronts per transform) with every CU busy and several transforms per
    workgroup: 4096 transforms = 16 distinct ones x 256 copies; copies bit-equal, the originals against the oracle."""
    from symphonia_amd import Fft, Imdct
    rng = np.random.default_rng(n)
    kinds, copies = 16, 256
    if n == 4096:
        x0 = (rng.standard_normal((kinds, n)) + 1j * rng.standard_normal((kinds, n))).astype(np.complex64)
        xd = torch.view_as_real(dev(np.ascontiguousarray(np.tile(x0, (copies, 1))))).contiguous()
        yd = torch.empty_like(xd)
        Fft(ctx, n).fft(xd, yd)
        torch.cuda.synchronize()
        y = yd.view(copies, kinds, -1)
        assert torch.equal(y.view(torch.int32), y[:1].expand_as(y).contiguous().view(torch.int32))
        got = host(yd[:kinds].contiguous()).reshape(kinds, -1)  # (re, im) pairs as f32
        want = np.ascontiguousarray(np.stack([oracle.fft(v) for v in x0]).astype(np.complex64)).view(np.float32).reshape(kinds, -1)
        assert_parity(got, want, "fft 4096")
    else:
        spec0 = (rng.standard_normal((kinds, n)) * np.exp2(rng.integers(-8, 8, (kinds, n)))).astype(np.float32)
        out = Imdct(ctx, n, -1.0 / 7).imdct(dev(np.ascontiguousarray(np.tile(spec0, (copies, 1)))))
        torch.cuda.synchronize()
        o = out.view(copies, kinds, -1)
        assert torch.equal(o.view(torch.int32), o[:1].expand_as(o).contiguous().view(torch.int32))
        assert_parity(host(out[:kinds]), oracle.imdct(spec0, -1.0 / 7), "imdct 8192")


def test_vorbis_config4_shard_properties(ctx):
    """BASELINE config 4, one GPU's shard: 8 streams x 8 ch = 64 chains x 4096 blocks, 2048/256 Markov block flags.
    Segmentation must not change a bit; sampled chains are checked against the oracle."""
    from symphonia_amd import VorbisDsp
    nch, nb = 64, 4096
    rng = np.random.default_rng(4)
    flags = np.zeros((nch, nb), np.uint8)
    cur = np.ones(nch, bool)
    for b in range(nb):  # P(long -> long) = 0.9, P(short -> short) = 0.7
        r = rng.random(nch)
        cur = np.where(cur, r < 0.9, r >= 0.7)
        flags[:, b] = cur
    prev = np.full(nch, -1, np.int32)
    v = VorbisDsp(ctx, 8, 11)
    so, po = v.layout(flags, prev)
    spec_stride, pcm_stride = int(so[:, -1].max()), int(po[:, -1].max())
    g = torch.Generator(device="cuda").manual_seed(4)
    spectra = torch.randn((nch, spec_stride), generator=g, device="cuda", dtype=torch.float32) * 0.1
    d_flags = dev(flags)
    outs = []
    for seg in (24, 200):
        ctx.set_segment(seg)
        d_prev, d_ov = dev(prev), torch.zeros((nch, 1024), device="cuda")
        pcm = torch.zeros((nch, pcm_stride), device="cuda")
        v.synth(spectra, d_flags, d_prev, d_ov, pcm_stride, pcm)
        outs.append((pcm, d_ov, d_prev))
    ctx.set_segment(0)
    torch.cuda.synchronize()
    for a, b in zip(outs[0], outs[1]):
        assert torch.equal(a.view(torch.int32), b.view(torch.int32))
    pick = [0, 31, 63]
    want = oracle.vorbis_synth(8, 11, host(spectra[pick]), flags[pick], prev[pick], np.zeros((3, 1024), np.float32), pcm_stride)
    used = po[pick, -1]
    got = host(outs[0][0][pick])
    for i in range(3):
        assert_parity(got[i, : used[i]], want[0][i, : used[i]], "config-4 sampled chain pcm")
    assert_parity(host(outs[0][1][pick]), want[1], "config-4 sampled overlap")
    assert np.array_equal(host(outs[0][2][pick]), want[2])


@pytest.mark.parametrize("bs0e,bs1e,seg", [(8, 11, 0), (7, 10, 0)])
def test_vorbis_fused_dot_product(ctx, bs0e, bs1e, seg):
    """symaccel_vorbis_synth_fr_device: floor x residue multiplied on load == dot product kernel + synth."""
    from test_emu_codecs import vorbis_case
    from symphonia_amd import VorbisDsp
    rng = np.random.default_rng(91 + bs0e)
    flags, prev, floor, overlap, pcm_stride = vorbis_case(rng, bs0e, bs1e, 10, 70)
    residue = rng.standard_normal(floor.shape).astype(np.float32)
    v = VorbisDsp(ctx, bs0e, bs1e)
    d_prev, d_ov = dev(prev), dev(overlap)
    pcm = torch.zeros((flags.shape[0], pcm_stride), device="cuda")
    v.synth_floor_residue(dev(floor), dev(residue), dev(flags), d_prev, d_ov, pcm_stride, pcm)
    want = oracle.vorbis_synth(bs0e, bs1e, floor * residue, flags, prev, overlap, pcm_stride)
    assert_parity(host(pcm), want[0], "fused pcm")
    assert_parity(host(d_ov), want[1], "fused overlap")
    assert np.array_equal(host(d_prev), want[2])


def test_vorbis_helpers_parity(ctx):
    from symphonia_amd import VorbisDsp
    rng = np.random.default_rng(8)
    v = VorbisDsp(ctx, 8, 11)
    n = 1024
    res = rng.standard_normal((8, n)).astype(np.float32)
    res[rng.random((8, n)) < 0.2] = 0.0
    want = res.copy()
    pairs = [(0, 1), (2, 3), (1, 2), (6, 7)]
    for m, a in pairs:
        want[m], want[a] = oracle.vorbis_inverse_coupling(want[m], want[a])
    d = dev(res)
    v.inverse_coupling(d, n, [p[0] for p in pairs], [p[1] for p in pairs])
    assert_parity(host(d), want, "coupling")
    fl = rng.standard_normal(8 * n).astype(np.float32)
    dfl = dev(fl)
    v.dot_product(dfl, d, 8 * n)
    assert_parity(host(dfl), oracle.vorbis_dot_product(fl, want.ravel()), "dot product")
    t2 = rng.standard_normal((3, 6 * 1024)).astype(np.float32)
    planar = torch.empty((3, 6, 1024), dtype=torch.float32, device="cuda")
    v.deinterleave2(dev(t2), planar, 6, 1024, 3)
    assert_parity(host(planar), np.stack([oracle.vorbis_deinterleave2(t, 6) for t in t2]), "deinterleave")
    for nn, n_posts, mult, cnt, p_zero in ((1024, 40, 2, 50, 0.3), (128, 12, 1, 50, 0.3), (1024, 65, 4, 50, 0.3),
                                           (2048, 65, 1, 333, 0.02), (32, 5, 2, 1000, 0.5), (4096, 33, 3, 65, 0.9)):
        xs = [0, nn] + rng.permutation(np.arange(1, nn))[:n_posts - 2].tolist()
        rr = [256, 128, 86, 64][mult - 1]
        ys = rng.integers(0, rr, size=(cnt, n_posts)).astype(np.uint32)
        ys[rng.random((cnt, n_posts)) < p_zero] = 0
        out = torch.zeros((cnt, nn), dtype=torch.float32, device="cuda")
        v.floor1(xs, mult, dev(ys), nn, out, cnt)
        curve = np.stack([oracle.vorbis_floor1(xs, y, mult, nn) for y in ys])
        assert_parity(host(out), curve, "floor1")
        res = (rng.standard_normal((cnt, nn)) * np.exp2(rng.integers(-8, 9, (cnt, nn)))).astype(np.float32)
        d_res = dev(res)
        v.floor1(xs, mult, dev(ys), nn, d_res, cnt, residue=d_res)  # floor x residue, in place on the residue
        assert np.array_equal(host(d_res).view(np.uint32), (curve * res).view(np.uint32)), "floor1 x residue"


# ------------------------------------------------------------------------------------------ FLAC

def test_flac_restore_parity(ctx):
    from symphonia_amd import FlacPredictor, flac_desc
    rng = np.random.default_rng(4)
    for blocksize in (1, 33, 192, 1000, 4096):
        nb = 200
        buf = rng.integers(-(1 << 23), 1 << 23, (nb, blocksize)).astype(np.int32)
        kind = rng.integers(0, 3, nb).astype(np.uint8)
        order = np.where(kind == 1, rng.integers(0, 5, nb), rng.integers(1, 33, nb))
        order = np.minimum(order, blocksize).astype(np.uint8)
        kind[(kind == 2) & (order == 0)] = 0
        shift = rng.integers(0, 16, nb).astype(np.uint8)
        wasted = np.where(rng.random(nb) < 0.2, rng.integers(1, 8, nb), 0).astype(np.uint8)
        coeffs = rng.integers(-(1 << 14), 1 << 14, (nb, 32)).astype(np.int32)
        d = dev(buf)
        FlacPredictor(ctx).restore(d, dev(flac_desc(kind, order, shift, wasted).view(np.uint8).reshape(nb, 4)), dev(coeffs))
        want = oracle.flac_restore(buf, oracle.flac_desc(kind, order, shift, wasted), coeffs)
        assert np.array_equal(host(d), want), blocksize


def test_flac_config5_encoder_identity(ctx):
    """BASELINE config 5 shape (order-32 LPC, 4096-sample blocks, 24-bit), reduced count: a forward
    encoder's residuals must decode back to the PCM exactly."""
    from symphonia_amd import FlacPredictor, flac_desc
    rng = np.random.default_rng(5)
    nb, bs, order, shift = 512, 4096, 32, 12
    t = np.arange(bs)
    pcm = (np.sin(t[None] * rng.uniform(0.01, 0.3, (nb, 1))) * 3e6 + rng.standard_normal((nb, bs)) * 2e4).astype(np.int64)
    co = np.zeros((nb, 32), np.int64)
    co[:, 0] = int(1.6 * (1 << shift))
    co[:, 1] = int(-0.7 * (1 << shift))
    co[:, 2:] = rng.integers(-60, 60, (nb, 30))
    res = pcm.copy()
    hist = np.stack([pcm[:, order - 1 - j: bs - 1 - j] for j in range(order)], axis=2)  # [nb, bs-order, order]
    pred = (hist * co[:, None, :]).sum(axis=2) >> shift
    res[:, order:] = pcm[:, order:] - pred
    assert np.abs(res).max() < (1 << 31)
    d = dev(res.astype(np.int32))
    kind = np.full(nb, 2, np.uint8)
    FlacPredictor(ctx).restore(d, dev(flac_desc(kind, kind * 0 + order, kind * 0 + shift, kind * 0).view(np.uint8).reshape(nb, 4)),
                               dev(co.astype(np.int32)))
    assert np.array_equal(host(d), pcm.astype(np.int32))


def test_flac_config5_step_sampled(ctx):
    """BASELINE config 5 at the size of one bench step and beyond what the oracle could redo in full: 65 536 order-32
    subframes of 4096 samples (1 GiB, 24-bit residual range, 15-bit coefficients, mixed shifts) restored on the GPU;
    128 sampled subframes, including the first and last wavefront's, are checked bit-for-bit against the oracle."""
    from symphonia_amd import FlacPredictor, flac_desc
    nb, bs = 65536, 4096
    g = torch.Generator(device="cuda").manual_seed(5)
    buf = torch.randint(-(1 << 14), 1 << 14, (nb, bs), generator=g, device="cuda", dtype=torch.int32)
    co = torch.randint(-900, 900, (nb, 32), generator=g, device="cuda", dtype=torch.int32)
    co[:, 0] = 6553
    co[:, 1] = -2867
    rng = np.random.default_rng(5)
    shift = rng.integers(10, 15, nb).astype(np.uint8)
    desc = flac_desc(np.full(nb, 2, np.uint8), np.full(nb, 32, np.uint8), shift, np.zeros(nb, np.uint8))
    pick = np.unique(np.concatenate(([0, 1, 63, 64, nb - 64, nb - 1], rng.integers(0, nb, 122))))
    before = host(buf[pick])
    FlacPredictor(ctx).restore(buf, dev(desc.view(np.uint8).reshape(nb, 4)), co)
    want = oracle.flac_restore(before, oracle.flac_desc(np.full(pick.size, 2), np.full(pick.size, 32), shift[pick],
                                                        np.zeros(pick.size)), host(co[pick]))
    assert np.array_equal(host(buf[pick]), want)


def test_flac_config5_full_size(ctx):
    """BASELINE config 5 at its full size -- 1 048 576 order-32 subframe blocks of 4096 24-bit samples (16 GiB in place),
    SURVEY 8d's generator (bench.flac_config5: quantised random AR(32) models, forward-predictor residuals, half the pairs
    mid/side) -- restored, decorrelated and left-justified (<< 8) by the fused stereo entry point in one call.  Checked
    two ways: the encoder identity (three whole 32 768-block generation chunks, first / middle / last, must come back as
    the PCM the residuals were made from) and the oracle on 96 sampled pairs spread over the batch."""
    import sys
    from pathlib import Path
    sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
    import bench
    from symphonia_amd import FlacPredictor
    free, _ = torch.cuda.mem_get_info()
    nb, bs, chunk = 1048576, 4096, 32768
    if free < 40 << 30:
        pytest.skip("needs 40 GiB of free HBM")
    buf, desc, co, pair_mode, expect = bench.flac_config5(torch, nb, bs, 5, "cuda", chunk)
    rng = np.random.default_rng(55)
    pairs = np.unique(np.concatenate(([0, 1, nb // 2 - 1], rng.integers(0, nb // 2, 93))))
    rows = np.stack([2 * pairs, 2 * pairs + 1], axis=1).ravel()
    before, d_h, c_h, pm_h = host(buf[rows]), host(desc[rows]), host(co[rows]), host(pair_mode[pairs])
    FlacPredictor(ctx).restore_stereo(buf, desc, co, pair_mode, 8)
    torch.cuda.synchronize()
    for b0 in (0, nb // 2, nb - chunk):
        assert torch.equal(buf[b0:b0 + chunk].to(torch.int64), expect(b0, b0 + chunk)), b0
    want = oracle.flac_restore(before, d_h, c_h)
    got = host(buf[rows])
    for i, p in enumerate(pairs):
        left, right = oracle.flac_decorrelate(int(pm_h[i]), want[2 * i], want[2 * i + 1])  # decoder.rs:557-571
        assert np.array_equal(got[2 * i], oracle.flac_shl(left, 8)) and np.array_equal(got[2 * i + 1], oracle.flac_shl(right, 8)), p
    del buf


@pytest.mark.parametrize("big_coeffs", [False, True])
def test_flac_extreme_ranges(ctx, big_coeffs):
    """Full-range i32 samples with |c| < 2^16 (FP64-exact dot product path) and |c| up to 2^30 (i64 path)."""
    from helpers import flac_extreme_case
    from symphonia_amd import FlacPredictor, flac_desc
    buf, kind, order, shift, coeffs = flac_extreme_case(29, big_coeffs)
    d = dev(buf)
    FlacPredictor(ctx).restore(d, dev(flac_desc(kind, order, shift, 0 * shift).view(np.uint8).reshape(-1, 4)), dev(coeffs))
    want = oracle.flac_restore(buf, oracle.flac_desc(kind, order, shift, 0 * shift), coeffs)
    assert np.array_equal(host(d), want)


@pytest.mark.parametrize("carrier", ["dot2", "dot2x2", "f64"])
def test_flac_carriers_at_their_edges(ctx, carrier):
    """Coefficient sets whose magnitudes sum to about 2^16, full-range i32 samples (helpers.flac_carrier_case): the reference's
    exact i64 sum, bit for bit."""
    from helpers import flac_carrier_case
    from symphonia_amd import FlacPredictor, flac_desc
    for seed in (41, 42):
        buf, kind, order, shift, coeffs = flac_carrier_case(seed, carrier)
        d = dev(buf)
        FlacPredictor(ctx).restore(d, dev(flac_desc(kind, order, shift, 0 * shift).view(np.uint8).reshape(-1, 4)), dev(coeffs))
        want = oracle.flac_restore(buf, oracle.flac_desc(kind, order, shift, 0 * shift), coeffs)
        assert np.array_equal(host(d), want), (seed, np.argwhere(host(d) != want)[:5])


@pytest.mark.parametrize("blocksize,nb", [(4096, 256), (1000, 130), (33, 64)])
def test_flac_restore_stereo_fused(ctx, blocksize, nb):
    """symaccel_flac_restore_stereo_device == restore, decorrelate, shift (decoder.rs:199-242)."""
    from symphonia_amd import FlacPredictor, flac_desc
    rng = np.random.default_rng(blocksize)
    buf = rng.integers(-(1 << 20), 1 << 20, (nb, blocksize)).astype(np.int32)
    kind = rng.integers(0, 3, nb).astype(np.uint8)
    order = np.minimum(np.where(kind == 1, rng.integers(0, 5, nb), rng.integers(1, 33, nb)), blocksize).astype(np.uint8)
    kind[(kind == 2) & (order == 0)] = 0
    shift = rng.integers(0, 16, nb).astype(np.uint8)
    wasted = np.where(rng.random(nb) < 0.2, rng.integers(1, 4, nb), 0).astype(np.uint8)
    coeffs = rng.integers(-(1 << 14), 1 << 14, (nb, 32)).astype(np.int32)
    mode = rng.integers(0, 4, nb // 2).astype(np.uint8)
    d = dev(buf)
    FlacPredictor(ctx).restore_stereo(d, dev(flac_desc(kind, order, shift, wasted).view(np.uint8).reshape(nb, 4)), dev(coeffs),
                                      dev(mode), 8)
    want = oracle.flac_restore(buf, oracle.flac_desc(kind, order, shift, wasted), coeffs)
    for p in range(nb // 2):
        a, b = oracle.flac_decorrelate(int(mode[p]), want[2 * p], want[2 * p + 1])
        want[2 * p], want[2 * p + 1] = oracle.flac_shl(a, 8), oracle.flac_shl(b, 8)
    assert np.array_equal(host(d), want)


def test_flac_decorrelate_parity(ctx):
    from symphonia_amd import FlacPredictor
    rng = np.random.default_rng(6)
    n_pairs, bs = 40, 1152
    a = rng.integers(-(1 << 24), 1 << 24, (n_pairs, bs)).astype(np.int32)
    b = rng.integers(-(1 << 24), 1 << 24, (n_pairs, bs)).astype(np.int32)
    mode = rng.integers(0, 4, n_pairs).astype(np.uint8)
    da, db = dev(a), dev(b)
    FlacPredictor(ctx).decorrelate(dev(mode), da, db, bs, out_shift=8)
    ga, gb = host(da), host(db)
    for p in range(n_pairs):
        wa, wb = oracle.flac_decorrelate(int(mode[p]), a[p], b[p])
        assert np.array_equal(ga[p], oracle.flac_shl(wa, 8)) and np.array_equal(gb[p], oracle.flac_shl(wb, 8))


def test_status_arrays_on_gpu(ctx):
    """symaccel_{flac,alac}_block_status_device, symaccel_aac_tns_status_device, symaccel_vorbis_floor1_status_device on
    the device (tests/test_host_tools.py runs the same cases through the CPU emulation)."""
    from test_host_tools import check_status_arrays

    def to_dev(a):
        a = np.ascontiguousarray(a)
        return torch.from_numpy(a.view(np.uint8).reshape(-1) if a.dtype.fields else a).cuda()

    def to_host(t):
        torch.cuda.synchronize()
        return t.cpu().numpy()

    check_status_arrays(ctx, to_dev, to_host)


def test_no_cpu_fallback(ctx):
    """The product library refuses to exist without HIP: creating a context on a bogus device fails."""
    from symphonia_amd import Context, SymaccelError
    with pytest.raises(SymaccelError):
        Context(99)
