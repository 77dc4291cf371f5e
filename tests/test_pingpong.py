"""The `_pp_device` entry points (state in and state out as two distinct buffers, ONE kernel launch) give exactly what the
`_io` entry points give, across consecutive calls that swap the buffers; aliasing is rejected.  CPU: through the
emulation build; GPU: the real library."""
import numpy as np
import pytest

import oracle
from emu_lib import emu_ctx  # noqa: F401
from helpers import aac_sequence_chain, aac_spectra, bit_equal
from symphonia_amd import AacDsp, Mp3Synthesis, MpaPolyphase, VorbisDsp, _ffi, aac_side, mp3_side


class HostTensor:
    """A numpy array with the two members backend.py uses of a device tensor (the emulation treats host memory as
    device memory), so the device-pointer code paths run on CPU."""
    __module__ = "torch.emulated"

    def __init__(self, a):
        self.a = np.ascontiguousarray(a)
        self.shape, self.dtype = self.a.shape, _TorchLike(self.a.dtype)

    def data_ptr(self):
        return self.a.ctypes.data

    def is_contiguous(self):
        return True

    def numel(self):
        return self.a.size

    def element_size(self):
        return self.a.itemsize


class _TorchLike:
    def __init__(self, dt):
        self.dt = dt

    def __eq__(self, other):
        return True


def run_aac(ctx, wrap, unwrap, seg):
    rng = np.random.default_rng(41 + seg)
    nch, nfr = 3, 7
    coeffs = [aac_spectra(rng, (nch, nfr)) for _ in range(3)]
    sides = []
    for _ in range(3):
        side = np.empty((nch, nfr), np.uint8)
        for c in range(nch):
            s, sh, pv = aac_sequence_chain(rng, nfr, p_switch=0.4)
            side[c] = aac_side(s, sh, pv)
        sides.append(side)
    delay0 = rng.standard_normal((nch, 1024)).astype(np.float32)
    ctx.set_segment(seg)
    dsp = AacDsp(ctx)
    st = [wrap(delay0.copy()), wrap(np.full((nch, 1024), np.nan, np.float32))]
    want_delay = delay0.copy()
    for k in range(3):  # three consecutive batches of one stream set: the state ping-pongs
        pcm = wrap(np.zeros((nch, nfr, 1024), np.float32))
        dsp.synth(wrap(coeffs[k]), wrap(sides[k]), st[0], pcm, delay_out=st[1])
        st.reverse()
        want_pcm, want_delay = oracle.aac_synth(coeffs[k], sides[k], want_delay)
        assert bit_equal(unwrap(pcm), want_pcm), k
        assert bit_equal(unwrap(st[0]), want_delay), k
    ctx.set_segment(0)


@pytest.mark.parametrize("seg", [1, 3, 64])
def test_emu_aac_pingpong(emu_ctx, seg):
    run_aac(emu_ctx, HostTensor, lambda t: t.a, seg)


def run_mp3(ctx, wrap, unwrap, seg):
    from test_emu_codecs import mp3_case
    rng = np.random.default_rng(7 + seg)
    nch, ngr = 3, 6
    ctx.set_segment(seg)
    syn = Mp3Synthesis(ctx, 2)
    ov, vv, vf = (rng.standard_normal((nch, 576)).astype(np.float32), rng.standard_normal((nch, 1024)).astype(np.float32),
                  rng.integers(0, 16, nch).astype(np.int32))
    st = [[wrap(ov.copy()), wrap(vv.copy()), wrap(vf.copy())],
          [wrap(np.zeros_like(ov)), wrap(np.zeros_like(vv)), wrap(np.zeros_like(vf))]]
    for k in range(3):
        xr, bt, mx, rz = mp3_case(rng, nch, ngr)
        side = mp3_side(bt, mx, rz)
        pcm = wrap(np.zeros((nch, ngr, 576), np.float32))
        syn.synth(wrap(xr), wrap(side.view(np.uint8).reshape(nch, ngr, 4)), st[0][0], st[0][1], st[0][2], pcm, state_out=st[1])
        st.reverse()
        want, ov, vv, vf = oracle.mp3_synth(xr, oracle.mp3_side(bt, mx, rz), 2, ov, vv, vf)
        assert bit_equal(unwrap(pcm), want), k
        assert bit_equal(unwrap(st[0][0]).reshape(ov.shape), ov) and bit_equal(unwrap(st[0][1]).reshape(vv.shape), vv), k
        assert np.array_equal(unwrap(st[0][2]), vf), k
    ctx.set_segment(0)


@pytest.mark.parametrize("seg", [2, 5])
def test_emu_mp3_pingpong(emu_ctx, seg):
    run_mp3(emu_ctx, HostTensor, lambda t: t.a, seg)


def run_vorbis(ctx, wrap, unwrap, bs0, bs1, fused):
    rng = np.random.default_rng(bs0 * 16 + bs1)
    nch, nb = 3, 9
    v = VorbisDsp(ctx, bs0, bs1)
    half1 = (1 << bs1) // 2
    prev = np.array([-1, 0, 1], np.int32)
    overlap = rng.standard_normal((nch, half1)).astype(np.float32)
    overlap[0] = 0.0
    st = [[wrap(prev.copy()), wrap(overlap.copy())], [wrap(np.zeros_like(prev)), wrap(np.zeros_like(overlap))]]
    for k in range(2):
        flags = rng.integers(0, 2, (nch, nb)).astype(np.uint8)
        so, po = v.layout(flags, prev)
        ss, ps = int(so[:, -1].max()), int(po[:, -1].max())
        floor = rng.standard_normal((nch, ss)).astype(np.float32)
        res = rng.standard_normal((nch, ss)).astype(np.float32)
        spectra = (floor * res).astype(np.float32)
        pcm = wrap(np.zeros((nch, ps), np.float32))
        if fused:
            v.synth(wrap(floor), wrap(flags), st[0][0], st[0][1], ps, pcm, state_out=st[1], residue=wrap(res))
        else:
            v.synth(wrap(spectra), wrap(flags), st[0][0], st[0][1], ps, pcm, state_out=st[1])
        st.reverse()
        want, overlap, prev = oracle.vorbis_synth(bs0, bs1, spectra, flags, prev, overlap, ps)
        got = unwrap(pcm)
        for c in range(nch):
            assert bit_equal(got[c, : po[c, -1]], want[c, : po[c, -1]]), (k, c)
        assert bit_equal(unwrap(st[0][1]), overlap) and np.array_equal(unwrap(st[0][0]), prev), k


@pytest.mark.parametrize("bs0,bs1,fused", [(8, 11, False), (8, 11, True), (6, 9, False), (7, 7, True)])
def test_emu_vorbis_pingpong(emu_ctx, bs0, bs1, fused):
    run_vorbis(emu_ctx, HostTensor, lambda t: t.a, bs0, bs1, fused)


def run_polyphase(ctx, wrap, unwrap, n_frames):
    rng = np.random.default_rng(n_frames)
    nch, npk = 2, 3
    x = rng.standard_normal((nch, npk, 32 * n_frames)).astype(np.float32)
    vv, vf = rng.standard_normal((nch, 1024)).astype(np.float32), rng.integers(0, 16, nch).astype(np.int32)
    st_out = (wrap(np.zeros_like(vv)), wrap(np.zeros_like(vf)))
    pcm = wrap(np.zeros_like(x))
    MpaPolyphase(ctx, n_frames).synth(wrap(x), wrap(vv.copy()), wrap(vf.copy()), pcm, state_out=st_out)
    want, wv, wf = np.empty_like(x), vv.copy(), vf.copy()
    for c in range(nch):
        v, f = vv[c], int(vf[c])
        for k in range(npk):
            want[c, k], v, f = oracle.mp3_polyphase(v, f, n_frames, x[c, k])
        wv[c], wf[c] = v, f
    assert bit_equal(unwrap(pcm), want) and bit_equal(unwrap(st_out[0]).reshape(wv.shape), wv) and np.array_equal(unwrap(st_out[1]), wf)


@pytest.mark.parametrize("n_frames", [12, 36])
def test_emu_polyphase_pingpong(emu_ctx, n_frames):
    run_polyphase(emu_ctx, HostTensor, lambda t: t.a, n_frames)


def test_pingpong_rejects_aliased_state(emu_ctx):
    d = emu_ctx.lib.dll
    f = np.zeros(4096, np.float32)
    b = np.zeros(64, np.uint8)
    i = np.zeros(4, np.int32)
    i2 = np.zeros(4, np.int32)
    p = lambda a: a.ctypes.data  # noqa: E731
    h = emu_ctx.handle
    assert d.symaccel_aac_synth_pp_device(h, p(f), p(b), p(f), p(f), p(f), 1, 1) == _ffi.ERR_INVALID_ARG
    assert d.symaccel_mp3_synth_pp_device(h, p(f), p(b), 0, p(f), p(f), p(i), p(f), p(f), p(i), p(f), 1, 1) == _ffi.ERR_INVALID_ARG
    assert d.symaccel_vorbis_synth_pp_device(h, 8, 11, p(f), None, 1024, p(b), p(i), p(i), p(f), p(f), p(f), 1024, 1, 1) == _ffi.ERR_INVALID_ARG
    assert d.symaccel_mpa_polyphase_pp_device(h, 12, p(f), p(f), p(i), p(f), p(i2), p(f), 1, 1) == _ffi.ERR_INVALID_ARG
    assert d.symaccel_mpa_polyphase_pp_device(h, 18, p(f), p(f), p(i), p(f), p(i2), p(f), 1, 1) == _ffi.ERR_UNSUPPORTED
    assert d.symaccel_aac_synth_pp_device(h, None, None, None, None, None, 0, 4) == _ffi.OK  # empty batch


# ---------------------------------------------------------------------------------------------- GPU

@pytest.fixture(scope="module")
def gpu_ctx():
    torch = pytest.importorskip("torch")
    if not torch.cuda.is_available():
        pytest.fail("no GPU visible: the gpu-marked tests must run on an MI355X (there is no CPU path)")
    from symphonia_amd import Context
    c = Context(0)
    c.use_torch_stream()
    yield c
    c.close()


def _gpu_wrap(a):
    import torch
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def _gpu_unwrap(t):
    import torch
    torch.cuda.synchronize()
    return t.cpu().numpy()


@pytest.mark.gpu
@pytest.mark.parametrize("seg", [0, 1, 3])
def test_gpu_aac_pingpong(gpu_ctx, seg):
    run_aac(gpu_ctx, _gpu_wrap, _gpu_unwrap, seg)


@pytest.mark.gpu
@pytest.mark.parametrize("seg", [0, 2, 5])
def test_gpu_mp3_pingpong(gpu_ctx, seg):
    run_mp3(gpu_ctx, _gpu_wrap, _gpu_unwrap, seg)


@pytest.mark.gpu
@pytest.mark.parametrize("bs0,bs1,fused", [(8, 11, False), (8, 11, True), (6, 9, False), (9, 12, True)])
def test_gpu_vorbis_pingpong(gpu_ctx, bs0, bs1, fused):
    run_vorbis(gpu_ctx, _gpu_wrap, _gpu_unwrap, bs0, bs1, fused)


@pytest.mark.gpu
@pytest.mark.parametrize("n_frames", [12, 36])
def test_gpu_polyphase_pingpong(gpu_ctx, n_frames):
    run_polyphase(gpu_ctx, _gpu_wrap, _gpu_unwrap, n_frames)


@pytest.mark.gpu
def test_entry_points_leave_the_current_device_alone(gpu_ctx):
    """a library call must not move the calling thread's current HIP device."""
    import torch
    before = torch.cuda.current_device()
    x = torch.zeros((2, 64), device="cuda")
    from symphonia_amd import Imdct
    Imdct(gpu_ctx, 64, 1.0).imdct(x)
    assert torch.cuda.current_device() == before
