"""The PRODUCT (HIP kernels through the C ABI) against the fixtures produced by executing the reference's Rust text
(tests/golden/rs_fixtures, tools/rs2fixtures.py) -- no oracle in between.  `-m gpu`: the hipcc-built library on the
MI355X.  Without a GPU the same checks run through the CPU emulation build of the same kernel sources (logic check)."""
import numpy as np
import pytest

from emu_lib import emu_ctx  # noqa: F401
from test_rs_fixtures import js_modes_of, load, rq_desc, same, st_desc, tns_filters_of


class Emu:
    """numpy arrays stand for device memory (the emulation library treats host memory as device memory)"""

    def __init__(self, ctx):
        self.ctx = ctx

    @staticmethod
    def dev(a):
        return np.ascontiguousarray(a).copy()

    @staticmethod
    def host(a):
        return a


class Gpu:
    def __init__(self, ctx):
        self.ctx = ctx

    @staticmethod
    def dev(a):
        import torch
        a = np.ascontiguousarray(a)
        if a.dtype.fields is not None:  # record arrays travel as bytes
            a = a.view(np.uint8).reshape(a.shape + (a.dtype.itemsize,))
        if a.dtype == np.uint16:
            return torch.from_numpy(a.view(np.int16)).cuda()
        if a.dtype == np.uint32:
            return torch.from_numpy(a.view(np.int32)).cuda()
        return torch.from_numpy(a).cuda()

    @staticmethod
    def host(t):
        import torch
        torch.cuda.synchronize()
        return t.cpu().numpy()


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


# ------------------------------------------------------------------------------------------ the checks

def check_fft(r):
    from symphonia_amd import Fft
    f = load("fft")
    for n in (2, 4, 8, 16, 32, 64, 128, 256, 512, 1024, 2048, 4096, 8192):
        x = r.dev(f["in_%d" % n][None])          # [1, n, 2] interleaved (re, im)
        y = r.dev(np.zeros((1, n, 2), np.float32))
        Fft(r.ctx, n).fft(x, y)
        assert same(r.host(y)[0], f["out_%d" % n]), n
        Fft(r.ctx, n).fft_inplace(x)
        assert same(r.host(x)[0], f["out_%d" % n]), n
    from symphonia_amd import Ifft
    for n in (2, 8, 16, 32, 64, 256, 1024, 4096):  # (below 32 points the reference's Ifft only permutes, swaps and scales)
        x = r.dev(np.stack([f["iin_%d" % n]] * 3))
        y = r.dev(np.zeros((3, n, 2), np.float32))
        Ifft(r.ctx, n).ifft(x, y)
        assert all(same(row, f["iout_%d" % n]) for row in r.host(y)), n
        Ifft(r.ctx, n).ifft_inplace(x)
        assert all(same(row, f["iout_%d" % n]) for row in r.host(x)), n


def check_imdct(r, manifest_cases):
    from symphonia_amd import Imdct
    f = load("imdct")
    for c in manifest_cases:
        key = c["key"]
        out = r.dev(np.zeros((1, 2 * c["n"]), np.float32))
        if isinstance(r, Emu):
            got = Imdct(r.ctx, c["n"], c["scale"]).imdct(f["spec_" + key][None])[0]
        else:
            got = r.host(Imdct(r.ctx, c["n"], c["scale"]).imdct(r.dev(f["spec_" + key][None]), out))[0]
        assert same(got, f["out_" + key]), c


def check_aac(r, seg):
    import oracle  # (only for the side-byte packing helper: a bit field, not arithmetic)
    from symphonia_amd import AacDsp
    f = load("aac")
    r.ctx.set_segment(seg)
    for walk in ("a", "b", "c", "d"):
        coeffs = f["coeffs_" + walk]
        side = np.tile(np.array([oracle.aac_side(s, sh, pv) for s, sh, pv in f["side_" + walk]], np.uint8), (coeffs.shape[0], 1))
        if isinstance(r, Emu):
            pcm, delay = AacDsp(r.ctx).synth(coeffs, side, f["delay_in_" + walk])
        else:
            d_delay = r.dev(f["delay_in_" + walk])
            pcm = r.host(AacDsp(r.ctx).synth(r.dev(coeffs), r.dev(side), d_delay))
            delay = r.host(d_delay)
        assert same(pcm, f["pcm_" + walk]) and same(delay, f["delay_out_" + walk]), (walk, seg)
    r.ctx.set_segment(0)


def check_mp3(r, seg):
    from symphonia_amd import Mp3Synthesis, MpaPolyphase, mp3_side
    f = load("mp3")
    r.ctx.set_segment(seg)
    for chain in ("long", "switch", "sr3", "sr8") + tuple("mix%d" % i for i in range(9)):
        k = "chain_%s_" % chain
        xr, g = f[k + "xr"], f[k + "side"]
        n = xr.shape[0]
        side = mp3_side(np.tile(g[:, 0], (n, 1)), np.tile(g[:, 1], (n, 1)), np.tile(g[:, 2], (n, 1)))
        ov, vv, vf = f[k + "overlap_in"].reshape(n, 576), f[k + "vvec_in"].reshape(n, 1024), np.full(n, f[k + "vfront"][0], np.int32)
        syn = Mp3Synthesis(r.ctx, int(f[k + "sr"][0]))
        if isinstance(r, Emu):
            pcm, ov2, vv2, vf2 = syn.synth(xr, side, ov, vv, vf)
        else:
            d = [r.dev(ov), r.dev(vv), r.dev(vf)]
            pcm = r.host(syn.synth(r.dev(xr), r.dev(side), *d))
            ov2, vv2, vf2 = (r.host(t) for t in d)
        assert same(pcm, f[k + "pcm"]), (chain, seg)
        assert same(ov2, f[k + "overlap_out"].reshape(n, 576)) and same(vv2, f[k + "vvec_out"].reshape(n, 1024)), (chain, seg)
        assert (vf2 == f[k + "vfront"][1]).all()
    r.ctx.set_segment(0)
    for nf in (12, 36):
        x = f["poly%d_in" % nf]
        n = x.shape[0]
        z = (np.zeros((n, 1024), np.float32), np.zeros(n, np.int32))
        if isinstance(r, Emu):
            pcm, vv, vf = MpaPolyphase(r.ctx, nf).synth(x, *z)
        else:
            d = [r.dev(z[0]), r.dev(z[1])]
            pcm = r.host(MpaPolyphase(r.ctx, nf).synth(r.dev(x), *d))
            vv, vf = r.host(d[0]), r.host(d[1])
        assert same(pcm, f["poly%d_out" % nf]), nf
        assert same(vv, f["poly%d_vvec_out" % nf].reshape(n, 1024)) and (vf == f["poly%d_vfront_out" % nf][0]).all()


def check_vorbis(r, cases, seg):
    from symphonia_amd import VorbisDsp
    f = load("vorbis")
    r.ctx.set_segment(seg)
    for c in cases:
        if c["fn"] != "DspChannel::synth":
            continue
        b0, b1 = c["bs0_exp"], c["bs1_exp"]
        k = "synth_%d_%d_" % (b0, b1)
        sp = f[k + "spectra"]
        n = sp.shape[0]
        flags = np.tile(f[k + "flags"], (n, 1))
        prev = np.full(n, int(f[k + "prev_flag"][0]), np.int32)
        v = VorbisDsp(r.ctx, b0, b1)
        stride = f[k + "pcm"].shape[1]
        if isinstance(r, Emu):
            pcm, ov, pf = v.synth(sp, flags, prev, f[k + "overlap_in"], stride)
        else:
            d_prev, d_ov = r.dev(prev), r.dev(f[k + "overlap_in"])
            pcm = r.host(v.synth(r.dev(sp), r.dev(flags), d_prev, d_ov, stride))
            ov, pf = r.host(d_ov), r.host(d_prev)
        assert same(pcm, f[k + "pcm"]) and same(ov, f[k + "overlap_out"]), (b0, b1, seg)
        assert (pf == f[k + "flags"][-1]).all()
    r.ctx.set_segment(0)
    v = VorbisDsp(r.ctx, 6, 7)
    res = r.dev(f["coupling_in"])
    pairs = f["coupling_pairs"]
    v.inverse_coupling(res, 64, pairs[:, 0], pairs[:, 1])
    coupled = r.host(res)
    assert same(coupled, f["coupling_out"])
    for c in range(4):
        if f["do_not_decode_out"][c]:
            continue
        fl = r.dev(f["dot_floor_in"][c])
        v.dot_product(fl, r.dev(f["coupling_out"][c]), 64)
        assert same(r.host(fl), f["dot_floor_out"][c]), c
    for c in cases:
        if not c["fn"].startswith("Floor1"):
            continue
        ci = c["case"]
        mult, n = (int(x) for x in f["floor1_%d_mult_n" % ci])
        y = f["floor1_%d_y" % ci]
        out = r.dev(np.zeros((y.shape[0], n), np.float32))
        v.floor1(f["floor1_%d_x" % ci], mult, r.dev(y), n, out, y.shape[0])
        assert same(r.host(out), f["floor1_%d_out" % ci]), ci


def check_flac(r):
    from symphonia_amd import FlacPredictor, flac_desc
    f = load("flac")
    fp = FlacPredictor(r.ctx)
    a, b = f["decor_ch0"], f["decor_ch1"]
    for mode in (1, 2, 3):
        i0, i1 = (a[4:] >> 1, b[4:] >> 2) if mode == 2 else (a[4:], b[4:])
        d0, d1 = r.dev(i0[None]), r.dev(i1[None])
        fp.decorrelate(r.dev(np.array([mode], np.uint8)), d0, d1, i0.size)
        assert same(r.host(d0)[0], f["decor_out0_m%d" % mode]) and same(r.host(d1)[0], f["decor_out1_m%d" % mode]), mode

    def restore(kind, order, shift, wasted, coeffs, buf):
        co = np.zeros((1, 32), np.int32)
        co[0, :len(coeffs)] = coeffs
        desc = flac_desc(np.array([kind]), np.array([order]), np.array([shift]), np.array([wasted]))
        if isinstance(r, Emu):
            return fp.restore(buf[None], desc, co)[0]
        d = r.dev(buf[None])
        fp.restore(d, r.dev(desc), r.dev(co))
        return r.host(d)[0]
    for sh in (0, 1, 8, 31):
        assert same(restore(0, 0, 0, sh, [], a), f["shl_%d" % sh]), sh
    for order in range(5):
        assert same(restore(1, order, 0, 0, [], f["fixed_in_%d" % order]), f["fixed_out_%d" % order]), order
    assert same(restore(1, 4, 0, 0, [], f["fixed_in_wrap"]), f["fixed_out_wrap"])
    ci = 0
    while "lpc_%d_in" % ci in f:
        order, shift = (int(v) for v in f["lpc_%d_order_shift" % ci])
        assert same(restore(2, order, shift, 0, f["lpc_%d_coeffs" % ci], f["lpc_%d_in" % ci]), f["lpc_%d_out" % ci]), (ci, order, shift)
        ci += 1


def check_alac(r):
    from symphonia_amd import AlacPredictor, alac_desc
    f = load("alac")
    ap = AlacPredictor(r.ctx)
    ci = 0
    while "predict_%d_in" % ci in f:
        mode, order, shift, bps = (int(v) for v in f["predict_%d_params" % ci])
        desc = alac_desc(np.array([mode]), np.array([order]), np.array([shift]), np.array([bps]))
        buf, co = f["predict_%d_in" % ci][None], f["predict_%d_coeffs" % ci][None]
        if isinstance(r, Emu):
            got = ap.predict(buf, desc, co)[0]
        else:
            d = r.dev(buf)
            ap.predict(d, r.dev(desc), r.dev(co))
            got = r.host(d)[0]
        assert same(got, f["predict_%d_out" % ci]), (ci, mode, order, shift, bps)
        ci += 1
    ci = 0
    while "ms_%d_in0" % ci in f:
        w, s = (int(v) for v in f["ms_%d_weight_shift" % ci])
        a, b = f["ms_%d_in0" % ci][None], f["ms_%d_in1" % ci][None]
        if w == 0:  # the ABI gives weight 0 the CALLER's meaning (decode_element skips the call, lib.rs:552): pair left alone
            ci += 1
            continue
        if isinstance(r, Emu):
            o0, o1 = ap.mid_side(np.array([w], np.int32), np.array([s], np.uint8), a, b)
        else:
            d0, d1 = r.dev(a), r.dev(b)
            ap.mid_side(r.dev(np.array([w], np.int32)), r.dev(np.array([s], np.uint8)), d0, d1)
            o0, o1 = r.host(d0), r.host(d1)
        assert same(o0[0], f["ms_%d_out0" % ci]) and same(o1[0], f["ms_%d_out1" % ci]), ci
        ci += 1


def check_mp3_front(r):
    from symphonia_amd import MP3_REQUANT_DTYPE, MP3_STEREO_DTYPE, Mp3Requantize, Mp3Stereo
    f = load("mp3_front")
    for ci in range(f["rq_quant"].shape[0]):
        d = np.zeros(1, MP3_REQUANT_DTYPE)
        o = rq_desc(f["rq_desc"][ci])
        for name in ("global_gain", "flags", "block_type", "is_mixed", "subblock_gain", "rzero", "scalefacs"):
            d[name] = o[name]
        rq = Mp3Requantize(r.ctx, int(f["rq_sr"][ci]))
        if isinstance(r, Emu):
            got = rq.requantize(f["rq_quant"][ci][None], d)[0]
        else:
            got = r.host(rq.requantize(r.dev(f["rq_quant"][ci][None]), r.dev(d)))[0]
        assert same(got, f["rq_out"][ci]), ci
    for ci in range(f["st_in"].shape[0]):
        row = f["st_desc"][ci]
        d = np.zeros((1, 1), MP3_STEREO_DTYPE)
        o = st_desc(row)
        for name in ("flags", "block_type", "is_mixed", "rzero0", "rzero1", "scalefacs1"):
            d[name] = o[name]
        xr = r.dev(f["st_in"][ci][:, None, :])   # two chains x one granule
        Mp3Stereo(r.ctx, int(row[5])).stereo(xr, r.dev(np.array([[0, 1]], np.int32)), r.dev(d))
        assert same(r.host(xr)[:, 0, :], f["st_out"][ci]), ci


def check_aac_tools(r):
    from symphonia_amd import AAC_JS_DTYPE, AAC_TNS_DTYPE, AacSpectralTools
    f = load("aac_tools")
    tools = AacSpectralTools(r.ctx, f["swb_long_48k"], f["swb_short_48k"])
    for ci in range(6):
        fl = tns_filters_of(f, ci)
        filt = np.zeros(len(fl), AAC_TNS_DTYPE)
        for k, (start, end, order, direction, lpc) in enumerate(fl):
            coef = np.zeros(20, np.float32)
            coef[:order] = lpc
            filt[k] = (0, start, end, order, direction, 0, coef)
        c = r.dev(f["tns_%d_in" % ci][None])
        tools.tns(c, r.dev(filt), len(fl))
        assert same(r.host(c)[0], f["tns_%d_out" % ci]), ci
    for ci in range(4):
        long_win, nwin, max_sfb, mode, scale = js_modes_of(f, ci)
        d = np.zeros((1, 1), AAC_JS_DTYPE)
        d["num_windows"], d["max_sfb"], d["mode"], d["scale"] = nwin, max_sfb, mode, scale
        c = r.dev(np.stack([f["js_%d_left_in" % ci], f["js_%d_right_in" % ci]])[:, None, :])
        tools.joint_stereo(c, r.dev(np.array([[0, 1]], np.int32)), r.dev(d))
        got = r.host(c)
        assert same(got[0, 0], f["js_%d_left_out" % ci]) and same(got[1, 0], f["js_%d_right_out" % ci]), ci


def manifest(group):
    from test_rs_fixtures import MANIFEST
    return MANIFEST[group]["cases"]


# ------------------------------------------------------------------------------------------ CPU: emulation build

def test_emu_core(emu_ctx):
    check_fft(Emu(emu_ctx))
    check_imdct(Emu(emu_ctx), manifest("imdct"))


@pytest.mark.parametrize("seg", [0, 1, 3])
def test_emu_aac(emu_ctx, seg):
    check_aac(Emu(emu_ctx), seg)


@pytest.mark.parametrize("seg", [0, 2, 3])
def test_emu_mp3(emu_ctx, seg):
    check_mp3(Emu(emu_ctx), seg)


@pytest.mark.parametrize("seg", [0, 1, 3])
def test_emu_vorbis(emu_ctx, seg):
    check_vorbis(Emu(emu_ctx), manifest("vorbis"), seg)


def test_emu_flac_alac(emu_ctx):
    check_flac(Emu(emu_ctx))
    check_alac(Emu(emu_ctx))


def test_emu_front_stages(emu_ctx):
    check_mp3_front(Emu(emu_ctx))
    check_aac_tools(Emu(emu_ctx))


# ------------------------------------------------------------------------------------------ GPU: the product

@pytest.mark.gpu
def test_gpu_core(gpu_ctx):
    check_fft(Gpu(gpu_ctx))
    check_imdct(Gpu(gpu_ctx), manifest("imdct"))


@pytest.mark.gpu
@pytest.mark.parametrize("seg", [0, 1, 3])
def test_gpu_aac(gpu_ctx, seg):
    check_aac(Gpu(gpu_ctx), seg)


@pytest.mark.gpu
@pytest.mark.parametrize("seg", [0, 2, 3])
def test_gpu_mp3(gpu_ctx, seg):
    check_mp3(Gpu(gpu_ctx), seg)


@pytest.mark.gpu
@pytest.mark.parametrize("seg", [0, 1, 3])
def test_gpu_vorbis(gpu_ctx, seg):
    check_vorbis(Gpu(gpu_ctx), manifest("vorbis"), seg)


@pytest.mark.gpu
def test_gpu_flac_alac(gpu_ctx):
    check_flac(Gpu(gpu_ctx))
    check_alac(Gpu(gpu_ctx))


@pytest.mark.gpu
def test_gpu_front_stages(gpu_ctx):
    check_mp3_front(Gpu(gpu_ctx))
    check_aac_tools(Gpu(gpu_ctx))
