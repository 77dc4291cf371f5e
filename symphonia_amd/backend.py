"""Host-side mirror of the reference's DSP interfaces, batched, on top of the symaccel C ABI.

Names follow the reference (symphonia-core / codec crates, 0.6.1):

    Imdct        symphonia_core::dsp::mdct::Imdct           (mdct.rs:16-146)
    Fft          symphonia_core::dsp::fft::Fft              (fft/no_simd.rs:70-141)
    AacDsp       symphonia-codec-aac  aac::dsp::Dsp         (aac/dsp.rs:22-158)
    Mp3Synthesis symphonia-bundle-mp3 layer3 hybrid synthesis + synthesis::synthesis
    VorbisDsp    symphonia-codec-vorbis dsp::Dsp / DspChannel (dsp.rs:12-145)
    FlacPredictor symphonia-bundle-flac decoder.rs predictors + decorrelation

Arguments may be numpy arrays (host entry points: staged through HBM, results returned as new
numpy arrays) or torch CUDA tensors (`*_device` entry points: zero-copy, enqueued on the current
torch stream, outputs written into the tensors you pass).  PyTorch is used only for device memory
and streams.  Error behaviour mirrors the reference: argument errors raise (the reference
asserts / panics), device problems raise SymaccelError (Error::IoError class).
"""
import ctypes as C

import numpy as np

from . import _ffi
from ._ffi import SymaccelError  # noqa: F401


def _is_torch(x):
    return type(x).__module__.startswith("torch")


def _ptr(x):
    if x is None:
        return None
    if _is_torch(x):
        return x.data_ptr()
    return x.ctypes.data


def _np(a, dtype):
    return np.ascontiguousarray(a, dtype=dtype)


class Context:
    """One symaccel context = one HIP device + stream + device-resident constant tables."""

    def __init__(self, device=0, library=None):
        self.lib = library if library is not None else _ffi.default_library()
        h = C.c_void_p()
        self.lib.check(self.lib.dll.symaccel_ctx_create(int(device), C.byref(h)))
        self.handle = h
        self.device = device
        self._batchers = []  # weak references: a batcher must go before its context does

    def close(self):
        for ref in getattr(self, "_batchers", []):
            b = ref()
            if b is not None:
                b.close()
        self._batchers = []
        if getattr(self, "handle", None):
            self.lib.dll.symaccel_ctx_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def _call(self, fn, *args):
        return self.lib.check(fn(self.handle, *args), self.handle)

    def set_stream(self, hip_stream_ptr):
        self._call(self.lib.dll.symaccel_ctx_set_stream, hip_stream_ptr)

    def use_torch_stream(self):
        import torch
        self.set_stream(torch.cuda.current_stream(self.device).cuda_stream)

    def set_segment(self, frames):
        self._call(self.lib.dll.symaccel_ctx_set_segment, int(frames))

    def sync(self):
        self._call(self.lib.dll.symaccel_sync)


class Imdct:
    """Imdct::new_scaled(n, scale) (mdct.rs:35-60)."""

    def __init__(self, ctx, n, scale=1.0):
        if n < 4 or n & (n - 1):
            raise ValueError("n must be a power of two")  # mdct.rs:37
        self.ctx, self.n, self.scale = ctx, int(n), float(scale)

    def imdct(self, spec, out=None):
        """spec[..., n] -> out[..., 2n] (mdct.rs:67-146), batched over the leading dims."""
        d = self.ctx.lib.dll
        if _is_torch(spec):
            import torch
            assert spec.is_cuda and spec.dtype == torch.float32 and spec.is_contiguous()
            assert spec.shape[-1] == self.n  # mdct.rs:76
            if out is None:
                out = torch.empty(spec.shape[:-1] + (2 * self.n,), dtype=torch.float32, device=spec.device)
            assert out.is_contiguous() and out.numel() == 2 * spec.numel()  # mdct.rs:78
            self.ctx._call(d.symaccel_imdct_f32_device, self.n, self.scale, _ptr(spec), _ptr(out),
                           spec.numel() // self.n)
            return out
        spec = _np(spec, np.float32)
        assert spec.shape[-1] == self.n
        res = np.empty(spec.shape[:-1] + (2 * self.n,), dtype=np.float32)
        self.ctx._call(d.symaccel_imdct_f32, self.n, self.scale, _ptr(spec), _ptr(res), spec.size // self.n)
        return res


class Fft:
    """Fft::new(n) (no_simd.rs:75-88); device tensors only (complex64, interleaved)."""

    _entry = "symaccel_fft_c32_device"

    def __init__(self, ctx, n):
        if n < 2 or n & (n - 1) or n > 65536:
            raise ValueError("n must be a power of two <= 65536")  # no_simd.rs:77-80
        self.ctx, self.n = ctx, int(n)

    def fft(self, x, y):
        """x, y: device buffers holding count * n complex values -- complex64 [..., n], or float32 [..., n, 2]
        (torch.view_as_real layout).  (numpy arrays are accepted for the CPU-emulated test library, whose "device"
        memory is host memory.)"""
        total = x.numel() if _is_torch(x) else x.size
        if str(x.dtype).endswith("complex64"):
            values = total
        else:
            assert str(x.dtype).endswith("float32") and x.shape[-1] == 2, "interleaved (re, im) float32 expected"
            values = total // 2
        assert (y.numel() if _is_torch(y) else y.size) == total and y.dtype == x.dtype
        assert values % self.n == 0  # no_simd.rs:97, 122-123: slice lengths must equal the transform size
        self.ctx._call(getattr(self.ctx.lib.dll, self._entry), self.n, _ptr(x), _ptr(y), values // self.n)
        return y

    def fft_inplace(self, x):
        return self.fft(x, x)


class Ifft(Fft):
    """Ifft::new(n) (no_simd.rs:143-158): ifft / ifft_inplace, same buffers as Fft."""

    _entry = "symaccel_ifft_c32_device"

    def ifft(self, x, y):
        return self.fft(x, y)

    def ifft_inplace(self, x):
        return self.fft(x, x)


def aac_side(seq, window_shape, prev_window_shape):
    """SYMACCEL_AAC_SIDE for arrays."""
    return ((np.asarray(seq, np.uint8) & 3) | ((np.asarray(window_shape, np.uint8) & 1) << 2)
            | ((np.asarray(prev_window_shape, np.uint8) & 1) << 3)).astype(np.uint8)


class AacDsp:
    """aac::dsp::Dsp (aac/dsp.rs:22-158), batched over chains x frames."""

    def __init__(self, ctx):
        self.ctx = ctx

    def synth(self, coeffs, side, delay, pcm=None, delay_out=None, chunk_frames=None):
        """coeffs[chains, frames, 1024], side[chains, frames] u8, delay[chains, 1024] (updated).
        numpy: returns (pcm, new_delay).  torch: writes pcm / delay in place, returns pcm.
        delay_out (torch, a second state buffer): the ping-pong entry point -- `delay` is only read, the new delay
        lines go to `delay_out`, one kernel launch; the caller swaps the two buffers for the next call."""
        d = self.ctx.lib.dll
        nch, nfr = int(coeffs.shape[0]), int(coeffs.shape[1])
        assert coeffs.shape[2] == 1024
        if _is_torch(coeffs):
            import torch
            assert coeffs.is_contiguous() and side.is_contiguous() and delay.is_contiguous()
            assert side.dtype == torch.uint8 and tuple(side.shape) == (nch, nfr) and tuple(delay.shape) == (nch, 1024)
            if pcm is None:
                pcm = torch.empty_like(coeffs)
            if delay_out is not None:
                assert delay_out.is_contiguous() and tuple(delay_out.shape) == (nch, 1024)
                self.ctx._call(d.symaccel_aac_synth_pp_device, _ptr(coeffs), _ptr(side), _ptr(delay), _ptr(delay_out), _ptr(pcm),
                               nch, nfr)
                return pcm
            self.ctx._call(d.symaccel_aac_synth_device, _ptr(coeffs), _ptr(side), _ptr(delay), _ptr(pcm), nch, nfr)
            return pcm
        coeffs = _np(coeffs, np.float32)
        side = _np(side, np.uint8)
        new_delay = np.array(delay, dtype=np.float32, copy=True, order="C")
        assert side.shape == (nch, nfr) and new_delay.shape == (nch, 1024)
        res = np.empty((nch, nfr, 1024), dtype=np.float32)
        if chunk_frames is not None:  # the staged (chunked, overlapped) host path, explicitly
            self.ctx._call(d.symaccel_aac_synth_pipelined, _ptr(coeffs), _ptr(side), _ptr(new_delay), _ptr(res), nch, nfr, int(chunk_frames))
        else:
            self.ctx._call(d.symaccel_aac_synth, _ptr(coeffs), _ptr(side), _ptr(new_delay), _ptr(res), nch, nfr)
        return res, new_delay


MP3_SIDE_DTYPE = np.dtype([("block_type", np.uint8), ("is_mixed", np.uint8), ("rzero", "<u2")])


def mp3_side(block_type, is_mixed, rzero):
    bt = np.asarray(block_type)
    s = np.zeros(bt.shape, dtype=MP3_SIDE_DTYPE)
    s["block_type"], s["is_mixed"], s["rzero"] = bt, np.asarray(is_mixed), np.asarray(rzero)
    return s


VORBIS_FLOOR1_DTYPE = np.dtype([("multiplier", np.uint8), ("n_posts", np.uint8), ("pad", np.uint8, (2,)), ("x_list", np.uint32, (65,))])
AAC_JS_DTYPE = np.dtype([("num_windows", np.uint8), ("max_sfb", np.uint8), ("pad", np.uint8, (2,)), ("mode", np.uint8, (128,)),
                         ("scale", np.float32, (128,))])
AAC_TNS_DTYPE = np.dtype([("frame", np.uint32), ("start", np.uint16), ("end", np.uint16), ("order", np.uint8),
                          ("direction", np.uint8), ("pad", np.uint8, (2,)), ("lpc", np.float32, (20,))])
AAC_JS_MS, AAC_JS_INTENSITY = 1, 2


class AacSpectralTools:
    """Joint-stereo decoding (aac/cpe.rs:110-157) and the TNS filters (aac/ics/tns.rs:180-195), in place on the
    chain-major coefficient array AacDsp.synth consumes.  Device-pointer entry points only: pass torch CUDA tensors
    (descriptors as uint8 views of AAC_JS_DTYPE / AAC_TNS_DTYPE records)."""

    def __init__(self, ctx, swb_long, swb_short):
        self.ctx = ctx
        self.swb_long, self.swb_short = _np(swb_long, np.uint16), _np(swb_short, np.uint16)

    def joint_stereo(self, coeffs, pair_chains, desc):
        """coeffs[chains, frames, 1024]; pair_chains[pairs, 2] i32; desc[pairs, frames] AAC_JS_DTYPE."""
        frames = int(coeffs.shape[1])
        n_pairs = int(pair_chains.shape[0])
        self.ctx._call(self.ctx.lib.dll.symaccel_aac_joint_stereo_device, _ptr(coeffs), frames, _ptr(pair_chains), _ptr(desc),
                       n_pairs, _ptr(self.swb_long), self.swb_long.size - 1, _ptr(self.swb_short), self.swb_short.size - 1)
        return coeffs

    def synth_joint_stereo(self, coeffs, side, delay, pair_chains, desc, pcm, delay_out=None):
        """Dsp::synth with the joint-stereo decoding of `pair_chains` done as the lines are loaded (one kernel; the decoded spectra never
        go to memory).  coeffs[chains, frames, 1024] as the spectrum decoder left them, side / delay / pcm as AacDsp.synth takes them;
        delay_out: the ping-pong entry point (delay is then only read)."""
        d = self.ctx.lib.dll
        nch, frames = int(coeffs.shape[0]), int(coeffs.shape[1])
        n_pairs = int(pair_chains.shape[0]) if pair_chains is not None else 0
        swb = (_ptr(self.swb_long), self.swb_long.size - 1, _ptr(self.swb_short), self.swb_short.size - 1)
        if delay_out is not None:
            self.ctx._call(d.symaccel_aac_synth_js_pp_device, _ptr(coeffs), _ptr(side), _ptr(pair_chains) if n_pairs else None,
                           _ptr(desc) if n_pairs else None, n_pairs, *swb, _ptr(delay), _ptr(delay_out), _ptr(pcm), nch, frames)
        else:
            self.ctx._call(d.symaccel_aac_synth_js_device, _ptr(coeffs), _ptr(side), _ptr(pair_chains) if n_pairs else None,
                           _ptr(desc) if n_pairs else None, n_pairs, *swb, _ptr(delay), _ptr(pcm), nch, frames)
        return pcm

    def joint_stereo_list(self, coeffs, pair_chains, desc, pair_frames):
        """joint_stereo for the listed channel-pair frames only (pair * frames + frame each): the frames that carry TNS filters."""
        frames = int(coeffs.shape[1])
        self.ctx._call(self.ctx.lib.dll.symaccel_aac_joint_stereo_list_device, _ptr(coeffs), frames, _ptr(pair_chains), _ptr(desc),
                       int(pair_chains.shape[0]), _ptr(self.swb_long), self.swb_long.size - 1, _ptr(self.swb_short), self.swb_short.size - 1,
                       _ptr(pair_frames), int(pair_frames.shape[0]))
        return coeffs

    def decode(self, coeffs, side, delay, pair_chains, desc, tns_filters, pcm, chunk_frames=0):
        """The whole AAC-LC tail HOST to HOST (symaccel_aac_decode_pipelined): coded spectra + joint-stereo descriptors + TNS filters
        (frame = chain * frames + frame) in numpy arrays -> pcm; delay is updated in place."""
        nch, frames = int(coeffs.shape[0]), int(coeffs.shape[1])
        n_pairs = int(pair_chains.shape[0]) if pair_chains is not None else 0
        n_tns = int(tns_filters.shape[0]) if tns_filters is not None else 0
        self.ctx._call(self.ctx.lib.dll.symaccel_aac_decode_pipelined, _ptr(coeffs), _ptr(side), _ptr(pair_chains) if n_pairs else None,
                       _ptr(desc) if n_pairs else None, n_pairs, _ptr(self.swb_long), self.swb_long.size - 1, _ptr(self.swb_short),
                       self.swb_short.size - 1, _ptr(tns_filters) if n_tns else None, n_tns, _ptr(delay), _ptr(pcm), nch, frames,
                       int(chunk_frames))
        return pcm

    def tns(self, coeffs, filters, n_filters=None):
        """coeffs[..., 1024] (any leading shape = n_frames); filters[n] AAC_TNS_DTYPE."""
        n_frames = 1
        for d in coeffs.shape[:-1]:
            n_frames *= int(d)
        if n_filters is None:
            n_filters = filters.shape[0]
        self.ctx._call(self.ctx.lib.dll.symaccel_aac_tns_device, _ptr(coeffs), n_frames, _ptr(filters), int(n_filters))
        return coeffs


class Mp3Synthesis:
    """Layer III synthesis tail (layer3/mod.rs:440-476): reorder, antialias, hybrid_synthesis,
    frequency_inversion, synthesis::synthesis."""

    def __init__(self, ctx, sample_rate_idx=0):
        if not 0 <= sample_rate_idx <= 8:
            raise ValueError("sample_rate_idx")
        self.ctx, self.sr = ctx, int(sample_rate_idx)

    def synth(self, xr, side, overlap, v_vec, v_front, pcm=None, state_out=None, chunk_granules=None):
        """state_out = (overlap_out, v_vec_out, v_front_out): the ping-pong entry point (device buffers only)."""
        d = self.ctx.lib.dll
        nch, ngr = int(xr.shape[0]), int(xr.shape[1])
        assert xr.shape[2] == 576
        if _is_torch(xr):
            import torch
            if pcm is None:
                pcm = torch.empty_like(xr)
            if state_out is not None:
                self.ctx._call(d.symaccel_mp3_synth_pp_device, _ptr(xr), _ptr(side), self.sr, _ptr(overlap), _ptr(v_vec),
                               _ptr(v_front), _ptr(state_out[0]), _ptr(state_out[1]), _ptr(state_out[2]), _ptr(pcm), nch, ngr)
                return pcm
            self.ctx._call(d.symaccel_mp3_synth_device, _ptr(xr), _ptr(side), self.sr, _ptr(overlap), _ptr(v_vec),
                           _ptr(v_front), _ptr(pcm), nch, ngr)
            return pcm
        xr = _np(xr, np.float32)
        side = np.ascontiguousarray(side)
        assert side.nbytes == nch * ngr * 4
        ov = np.array(overlap, dtype=np.float32, copy=True, order="C")
        vv = np.array(v_vec, dtype=np.float32, copy=True, order="C")
        vf = np.array(v_front, dtype=np.int32, copy=True, order="C")
        assert ov.size == nch * 576 and vv.size == nch * 1024 and vf.size == nch
        res = np.empty((nch, ngr, 576), dtype=np.float32)
        if chunk_granules is not None:
            self.ctx._call(d.symaccel_mp3_synth_pipelined, _ptr(xr), _ptr(side), self.sr, _ptr(ov), _ptr(vv), _ptr(vf), _ptr(res),
                           nch, ngr, int(chunk_granules))
        else:
            self.ctx._call(d.symaccel_mp3_synth, _ptr(xr), _ptr(side), self.sr, _ptr(ov), _ptr(vv), _ptr(vf), _ptr(res),
                           nch, ngr)
        return res, ov, vv, vf


    def decode(self, quant, rq_desc, unit_chains, st_desc, side, overlap, v_vec, v_front, pcm=None, state_out=None):
        """The whole tail from the entropy decoder's output in ONE kernel (symaccel_mp3_decode_*_device; device buffers):
        quant[chains, granules, 576] i16, rq_desc[chains, granules] (52-byte records), unit_chains[units, 2] i32 (second
        -1 for a mono stream), st_desc[units, granules] (48-byte records), side as synth() takes it."""
        d = self.ctx.lib.dll
        nch, ngr = int(quant.shape[0]), int(quant.shape[1])
        nu = int(unit_chains.shape[0])
        if pcm is None:
            import torch
            pcm = torch.empty((nch, ngr, 576), dtype=torch.float32, device=quant.device)
        if state_out is not None:
            self.ctx._call(d.symaccel_mp3_decode_pp_device, _ptr(quant), _ptr(rq_desc), _ptr(unit_chains), _ptr(st_desc), nu, _ptr(side),
                           self.sr, _ptr(overlap), _ptr(v_vec), _ptr(v_front), _ptr(state_out[0]), _ptr(state_out[1]), _ptr(state_out[2]),
                           _ptr(pcm), nch, ngr)
            return pcm
        self.ctx._call(d.symaccel_mp3_decode_device, _ptr(quant), _ptr(rq_desc), _ptr(unit_chains), _ptr(st_desc), nu, _ptr(side), self.sr,
                       _ptr(overlap), _ptr(v_vec), _ptr(v_front), _ptr(pcm), nch, ngr)
        return pcm


class MpaPolyphase:
    """synthesis::synthesis for Layer I (n_frames 12) / Layer II (n_frames 36) (synthesis.rs:158-336)."""

    def __init__(self, ctx, n_frames):
        if n_frames not in (12, 36):
            raise ValueError("n_frames must be 12 (Layer I) or 36 (Layer II)")
        self.ctx, self.n_frames = ctx, int(n_frames)

    def synth(self, samples, v_vec, v_front, pcm=None, state_out=None):
        """samples[chains, packets, 32 * n_frames] sub-band-major; state v_vec[chains, 1024], v_front[chains] i32.
        numpy: returns (pcm, v_vec, v_front); torch: state updated in place, returns pcm.
        state_out = (v_vec_out, v_front_out): the ping-pong entry point (device buffers only)."""
        d = self.ctx.lib.dll
        nch, npk = int(samples.shape[0]), int(samples.shape[1])
        assert samples.shape[2] == 32 * self.n_frames
        if _is_torch(samples):
            import torch
            if pcm is None:
                pcm = torch.empty_like(samples)
            if state_out is not None:
                self.ctx._call(d.symaccel_mpa_polyphase_pp_device, self.n_frames, _ptr(samples), _ptr(v_vec), _ptr(v_front),
                               _ptr(state_out[0]), _ptr(state_out[1]), _ptr(pcm), nch, npk)
                return pcm
            self.ctx._call(d.symaccel_mpa_polyphase_device, self.n_frames, _ptr(samples), _ptr(v_vec), _ptr(v_front), _ptr(pcm),
                           nch, npk)
            return pcm
        x = _np(samples, np.float32)
        vv = np.array(v_vec, dtype=np.float32, copy=True, order="C")
        vf = np.array(v_front, dtype=np.int32, copy=True, order="C")
        res = np.empty_like(x)
        self.ctx._call(d.symaccel_mpa_polyphase, self.n_frames, _ptr(x), _ptr(vv), _ptr(vf), _ptr(res), nch, npk)
        return res, vv, vf


MP3_REQUANT_DTYPE = np.dtype([("global_gain", np.uint8), ("flags", np.uint8), ("block_type", np.uint8),
                              ("is_mixed", np.uint8), ("subblock_gain", np.uint8, (3,)), ("reserved", np.uint8),
                              ("rzero", np.uint16), ("scalefacs", np.uint8, (39,)), ("pad", np.uint8, (3,))])
MP3_RQ_SCALEFAC_SCALE, MP3_RQ_PREFLAG = 1, 2


class Mp3Requantize:
    """read_huffman_samples' sample mapping + requantize (layer3/requantize.rs:28-31, 117-147, 239-380)."""

    def __init__(self, ctx, sample_rate_idx):
        if not 0 <= int(sample_rate_idx) <= 8:
            raise ValueError("sample_rate_idx")
        self.ctx, self.sr = ctx, int(sample_rate_idx)

    def requantize(self, quant, desc, xr=None):
        """quant[..., 576] int16, desc[...] MP3_REQUANT_DTYPE (torch: uint8[..., 52]); returns xr[..., 576] f32."""
        d = self.ctx.lib.dll
        if _is_torch(quant):
            import torch
            n = quant.numel() // 576
            assert desc.numel() * desc.element_size() == 52 * n
            if xr is None:
                xr = torch.empty(quant.shape, dtype=torch.float32, device=quant.device)
            self.ctx._call(d.symaccel_mp3_requantize_device, _ptr(quant), _ptr(desc), self.sr, _ptr(xr), n)
            return xr
        q = _np(quant, np.int16)
        dd = np.ascontiguousarray(desc, dtype=MP3_REQUANT_DTYPE)
        assert dd.size * 576 == q.size
        res = np.empty(q.shape, np.float32)
        self.ctx._call(d.symaccel_mp3_requantize, _ptr(q), _ptr(dd), self.sr, _ptr(res), dd.size)
        return res


MP3_STEREO_DTYPE = np.dtype([("flags", np.uint8), ("block_type", np.uint8), ("is_mixed", np.uint8), ("reserved", np.uint8),
                             ("rzero0", np.uint16), ("rzero1", np.uint16), ("scalefacs1", np.uint8, (39,)), ("pad", np.uint8)])
MP3_ST_MID_SIDE, MP3_ST_INTENSITY, MP3_ST_MPEG1, MP3_ST_IS_SCALE = 1, 2, 4, 8


class Mp3Stereo:
    """stereo() (layer3/stereo.rs:485-556), in place on xr[chains, granules, 576] (device-pointer entry point)."""

    def __init__(self, ctx, sample_rate_idx):
        if not 0 <= int(sample_rate_idx) <= 8:
            raise ValueError("sample_rate_idx")
        self.ctx, self.sr = ctx, int(sample_rate_idx)

    def stereo(self, xr, pair_chains, desc):
        """pair_chains[pairs, 2] i32; desc[pairs, granules] MP3_STEREO_DTYPE (torch: uint8[pairs, granules, 48])."""
        self.ctx._call(self.ctx.lib.dll.symaccel_mp3_stereo_device, _ptr(xr), int(xr.shape[1]), _ptr(pair_chains), _ptr(desc),
                       self.sr, int(pair_chains.shape[0]))
        return xr

    def requantize_stereo(self, quant, rq_desc, pair_chains, desc, xr):
        """Mp3Requantize.requantize + stereo for the paired chains in one pass: quant[chains, granules, 576] i16,
        rq_desc[chains, granules] MP3_REQUANT_DTYPE, xr[chains, granules, 576] f32 (only the paired chains are written)."""
        self.ctx._call(self.ctx.lib.dll.symaccel_mp3_requantize_stereo_device, _ptr(quant), _ptr(rq_desc), int(quant.shape[1]),
                       _ptr(pair_chains), _ptr(desc), self.sr, _ptr(xr), int(pair_chains.shape[0]))
        return xr


class VorbisDsp:
    """dsp::Dsp / DspChannel::synth (vorbis/dsp.rs:12-145) for chains of mixed-size blocks."""

    def __init__(self, ctx, bs0_exp, bs1_exp):
        if not (6 <= bs0_exp <= bs1_exp <= 13):
            raise ValueError("block size exponents")  # vorbis/lib.rs:404-406, 461-470
        self.ctx, self.bs0_exp, self.bs1_exp = ctx, int(bs0_exp), int(bs1_exp)

    def layout(self, block_flag, prev_flag):
        """Packed offsets: (spec_off[chains, blocks+1], pcm_off[chains, blocks+1])."""
        bf = np.asarray(block_flag).astype(np.int64)
        nch, nb = bf.shape
        bs = np.where(bf > 0, 1 << self.bs1_exp, 1 << self.bs0_exp)
        pf = np.empty_like(bf)
        first = np.asarray(prev_flag).astype(np.int64)
        pf[:, 0] = np.where(first < 0, bf[:, 0], first)
        pf[:, 1:] = bf[:, :-1]
        prev_n = np.where(pf > 0, 1 << self.bs1_exp, 1 << self.bs0_exp)
        so = np.zeros((nch, nb + 1), np.int64)
        po = np.zeros((nch, nb + 1), np.int64)
        so[:, 1:] = np.cumsum(bs // 2, axis=1)
        po[:, 1:] = np.cumsum((prev_n + bs) // 4, axis=1)
        return so, po

    def synth(self, spectra, block_flag, prev_flag, overlap, pcm_stride, pcm=None, state_out=None, residue=None):
        """state_out = (prev_flag_out, overlap_out): the ping-pong entry point (device buffers only; `residue` fuses the dot
        product there: spectra is then the floor)."""
        d = self.ctx.lib.dll
        nch, nb = int(block_flag.shape[0]), int(block_flag.shape[1])
        spec_stride = int(spectra.shape[1])
        if _is_torch(spectra):
            import torch
            if pcm is None:
                pcm = torch.zeros((nch, pcm_stride), dtype=torch.float32, device=spectra.device)
            if state_out is not None:
                self.ctx._call(d.symaccel_vorbis_synth_pp_device, self.bs0_exp, self.bs1_exp, _ptr(spectra),
                               _ptr(residue) if residue is not None else None, spec_stride, _ptr(block_flag), _ptr(prev_flag),
                               _ptr(state_out[0]), _ptr(overlap), _ptr(state_out[1]), _ptr(pcm), int(pcm_stride), nch, nb)
                return pcm
            assert residue is None
            self.ctx._call(d.symaccel_vorbis_synth_device, self.bs0_exp, self.bs1_exp, _ptr(spectra), spec_stride,
                           _ptr(block_flag), _ptr(prev_flag), _ptr(overlap), _ptr(pcm), int(pcm_stride), nch, nb)
            return pcm
        sp = _np(spectra, np.float32)
        bf = _np(block_flag, np.uint8)
        pf = np.array(prev_flag, dtype=np.int32, copy=True, order="C")
        ov = np.array(overlap, dtype=np.float32, copy=True, order="C")
        res = np.zeros((nch, int(pcm_stride)), dtype=np.float32)
        self.ctx._call(d.symaccel_vorbis_synth, self.bs0_exp, self.bs1_exp, _ptr(sp), spec_stride, _ptr(bf), _ptr(pf),
                       _ptr(ov), _ptr(res), int(pcm_stride), nch, nb)
        return res, ov, pf

    def synth_floor_residue(self, floor, residue, block_flag, prev_flag, overlap, pcm_stride, pcm):
        """synth() with the dot product of lib.rs:282-292 fused: spectrum = floor * residue, multiplied on load.
        Device buffers (torch tensors, or raw arrays with the CPU-emulated test library); state updated in place."""
        nch, nb = int(block_flag.shape[0]), int(block_flag.shape[1])
        self.ctx._call(self.ctx.lib.dll.symaccel_vorbis_synth_fr_device, self.bs0_exp, self.bs1_exp, _ptr(floor), _ptr(residue),
                       int(floor.shape[1]), _ptr(block_flag), _ptr(prev_flag), _ptr(overlap), _ptr(pcm), int(pcm_stride), nch, nb)
        return pcm

    def synth_floor_y(self, floor_y, residue, block_flag, prev_flag, overlap, pcm_stride, pcm, state_out=None):
        """synth() from the floor curve as dB-table indices (one byte per line, floor1(..., y_plane=)) and the residue: the table
        look-up and the dot product happen in the synthesis kernel's load path.  state_out = (prev_flag_out, overlap_out): the
        ping-pong entry point; otherwise the state is updated in place."""
        d = self.ctx.lib.dll
        nch, nb = int(block_flag.shape[0]), int(block_flag.shape[1])
        if state_out is not None:
            self.ctx._call(d.symaccel_vorbis_synth_fy_pp_device, self.bs0_exp, self.bs1_exp, _ptr(floor_y), _ptr(residue),
                           int(residue.shape[1]), _ptr(block_flag), _ptr(prev_flag), _ptr(state_out[0]), _ptr(overlap),
                           _ptr(state_out[1]), _ptr(pcm), int(pcm_stride), nch, nb)
        else:
            self.ctx._call(d.symaccel_vorbis_synth_fy_device, self.bs0_exp, self.bs1_exp, _ptr(floor_y), _ptr(residue),
                           int(residue.shape[1]), _ptr(block_flag), _ptr(prev_flag), _ptr(overlap), _ptr(pcm), int(pcm_stride), nch, nb)
        return pcm

    def decode(self, residue, block_flag, floor, posts, floors, channels_per_stream, coupling, coupling_first, prev_flag, overlap,
               pcm_stride, pcm):
        """The Vorbis tail HOST to HOST (symaccel_vorbis_decode): residue[chains, spec_stride] f32, block_flag / floor[chains, blocks] u8
        (floor: index into `floors` or 255 = unused), posts[chains, blocks, posts_stride] u32, floors[n] VORBIS_FLOOR1_DTYPE,
        coupling[steps, 2] u8 + coupling_first[streams * blocks + 1] u32; prev_flag / overlap are updated in place; numpy arrays."""
        nch, nb = int(block_flag.shape[0]), int(block_flag.shape[1])
        self.ctx._call(self.ctx.lib.dll.symaccel_vorbis_decode, self.bs0_exp, self.bs1_exp, _ptr(residue), int(residue.shape[1]),
                       _ptr(block_flag), _ptr(floor), _ptr(posts), int(posts.shape[2]), _ptr(floors), int(floors.shape[0]),
                       int(channels_per_stream), _ptr(coupling) if coupling.size else None, _ptr(coupling_first), _ptr(prev_flag),
                       _ptr(overlap), _ptr(pcm), int(pcm_stride), nch, nb)
        return pcm

    # device-pointer helpers (torch tensors, or raw arrays when the library treats host memory as device)
    def inverse_coupling(self, residue, n, mag_index, ang_index):
        mi = _np(mag_index, np.uint32)
        ai = _np(ang_index, np.uint32)
        self.ctx._call(self.ctx.lib.dll.symaccel_vorbis_inverse_coupling_device, _ptr(residue), int(n), _ptr(mi),
                       _ptr(ai), mi.size)

    def dot_product(self, floor, residue, total):
        self.ctx._call(self.ctx.lib.dll.symaccel_vorbis_dot_product_device, _ptr(floor), _ptr(residue), int(total))

    def deinterleave2(self, type2, planar, n_ch, n2, count):
        self.ctx._call(self.ctx.lib.dll.symaccel_vorbis_deinterleave2_device, _ptr(type2), _ptr(planar), int(n_ch),
                       int(n2), int(count))

    def floor1(self, x_list, multiplier, y, n, floor, count, residue=None, y_plane=None, line_offsets=None):
        """floor[count][n] = the floor-1 curve; with `residue`, floor = curve * residue (the dot product fused into the
        curve's store; `floor` may be `residue`); with `y_plane` (uint8), the curve's dB-table indices, one byte per line, block b
        at byte offset line_offsets[b] (None: b * n) -- what synth_floor_y() reads."""
        xl = _np(x_list, np.uint32)
        if y_plane is not None:
            self.ctx._call(self.ctx.lib.dll.symaccel_vorbis_floor1_y_device, _ptr(xl), xl.size, int(multiplier), _ptr(y), int(n),
                           _ptr(line_offsets) if line_offsets is not None else None, _ptr(y_plane), int(count))
        elif residue is None:
            self.ctx._call(self.ctx.lib.dll.symaccel_vorbis_floor1_device, _ptr(xl), xl.size, int(multiplier), _ptr(y),
                           int(n), _ptr(floor), int(count))
        elif line_offsets is not None:
            self.ctx._call(self.ctx.lib.dll.symaccel_vorbis_floor1_dot_at_device, _ptr(xl), xl.size, int(multiplier), _ptr(y),
                           int(n), _ptr(line_offsets), _ptr(residue), _ptr(floor), int(count))
        else:
            self.ctx._call(self.ctx.lib.dll.symaccel_vorbis_floor1_dot_device, _ptr(xl), xl.size, int(multiplier), _ptr(y),
                           int(n), _ptr(residue), _ptr(floor), int(count))


    def floor1_y_jobs(self, jobs, y_plane):
        """several floor1(..., y_plane=...) renders into ONE plane, two per launch (symaccel_vorbis_floor1_y_jobs_device):
        jobs = [(x_list, multiplier, y, n, line_offsets or None, count)]; the jobs' lines must not overlap."""
        from ._ffi import VorbisFloor1Job
        arr = (VorbisFloor1Job * max(1, len(jobs)))()
        keep = []
        for k, (x_list, multiplier, y, n, line_offsets, count) in enumerate(jobs):
            xl = _np(x_list, np.uint32)
            keep.append(xl)
            arr[k] = VorbisFloor1Job(_ptr(xl), xl.size, int(multiplier), _ptr(y), int(n),
                                     _ptr(line_offsets) if line_offsets is not None else None, int(count))
        self.ctx._call(self.ctx.lib.dll.symaccel_vorbis_floor1_y_jobs_device, arr, len(jobs), _ptr(y_plane))


FLAC_DESC_DTYPE = np.dtype([("kind", np.uint8), ("order", np.uint8), ("shift", np.uint8), ("wasted_bits", np.uint8)])
FLAC_VERBATIM, FLAC_FIXED, FLAC_LPC = 0, 1, 2


def flac_desc(kind, order, shift, wasted_bits):
    k = np.asarray(kind)
    dsc = np.zeros(k.shape, dtype=FLAC_DESC_DTYPE)
    dsc["kind"], dsc["order"], dsc["shift"], dsc["wasted_bits"] = k, np.asarray(order), np.asarray(shift), np.asarray(wasted_bits)
    return dsc


class FlacPredictor:
    """fixed_predict / lpc_predict / decorrelate_* of symphonia-bundle-flac/src/decoder.rs."""

    def __init__(self, ctx):
        self.ctx = ctx

    def restore(self, buf, desc, coeffs, chunk_blocks=None):
        """buf[n_blocks, blocksize] i32 (warm-up + residuals) -> samples.  numpy: returns a new array;
        torch: in place."""
        d = self.ctx.lib.dll
        nb, bs = int(buf.shape[0]), int(buf.shape[1])
        if _is_torch(buf):
            self.ctx._call(d.symaccel_flac_restore_device, _ptr(buf), _ptr(desc), _ptr(coeffs), nb, bs)
            return buf
        res = np.array(buf, dtype=np.int32, copy=True, order="C")
        dsc = np.ascontiguousarray(desc)
        co = _np(coeffs, np.int32)
        assert dsc.nbytes == nb * 4 and co.shape == (nb, 32)
        if chunk_blocks is not None:
            self.ctx._call(d.symaccel_flac_restore_pipelined, _ptr(res), _ptr(dsc), _ptr(co), nb, bs, int(chunk_blocks))
        else:
            self.ctx._call(d.symaccel_flac_restore, _ptr(res), _ptr(dsc), _ptr(co), nb, bs)
        return res

    def restore_stereo(self, buf, desc, coeffs, pair_mode, out_shift=0):
        """restore() with decorrelate + `<< out_shift` fused into the write-back: blocks 2p / 2p+1 = channels of pair p.
        Device buffers, in place."""
        nb, bs = int(buf.shape[0]), int(buf.shape[1])
        self.ctx._call(self.ctx.lib.dll.symaccel_flac_restore_stereo_device, _ptr(buf), _ptr(desc), _ptr(coeffs),
                       _ptr(pair_mode), int(out_shift), nb, bs)
        return buf

    def restore_strided(self, buf, desc, coeffs, blocksize, pair_mode=None, out_shift=0):
        """restore() / restore_stereo() over padded rows: buf[n_blocks, stride] i32 on the device, the first `blocksize` words of a
        row are the subframe (symaccel_flac_restore_strided_device).  In place; the padding is neither read nor written."""
        nb, stride = int(buf.shape[0]), int(buf.shape[1])
        self.ctx._call(self.ctx.lib.dll.symaccel_flac_restore_strided_device, _ptr(buf), _ptr(desc), _ptr(coeffs),
                       _ptr(pair_mode) if pair_mode is not None else None, int(out_shift), nb, int(blocksize), stride)
        return buf

    def decorrelate(self, mode, ch0, ch1, blocksize, out_shift=0):
        n_pairs = (ch0.numel() if _is_torch(ch0) else ch0.size) // int(blocksize)
        self.ctx._call(self.ctx.lib.dll.symaccel_flac_decorrelate_device, _ptr(mode), _ptr(ch0), _ptr(ch1), n_pairs,
                       int(blocksize), int(out_shift))


ALAC_DESC_DTYPE = np.dtype([("mode", np.uint8), ("lpc_order", np.uint8), ("shift", np.uint8), ("bps", np.uint8)])


def alac_desc(mode, lpc_order, shift, bps):
    m = np.asarray(mode)
    d = np.zeros(m.shape, dtype=ALAC_DESC_DTYPE)
    d["mode"], d["lpc_order"], d["shift"], d["bps"] = m, np.asarray(lpc_order), np.asarray(shift), np.asarray(bps)
    return d


class AlacPredictor:
    """ElementChannel::predict and decorrelate_mid_side of symphonia-codec-alac/src/lib.rs (165-264, 664-671)."""

    def __init__(self, ctx):
        self.ctx = ctx

    def predict(self, buf, desc, coeffs):
        """buf[n_blocks, blocksize] i32 residuals -> samples.  numpy: returns a new array; torch: in place."""
        d = self.ctx.lib.dll
        nb, bs = int(buf.shape[0]), int(buf.shape[1])
        if _is_torch(buf):
            self.ctx._call(d.symaccel_alac_predict_device, _ptr(buf), _ptr(desc), _ptr(coeffs), nb, bs)
            return buf
        res = np.array(buf, dtype=np.int32, copy=True, order="C")
        dsc = np.ascontiguousarray(desc)
        co = _np(coeffs, np.int32)
        assert dsc.nbytes == nb * 4 and co.shape == (nb, 32)
        self.ctx._call(d.symaccel_alac_predict, _ptr(res), _ptr(dsc), _ptr(co), nb, bs)
        return res

    def predict_stereo(self, buf, desc, coeffs, pair_weight, pair_shift):
        """predict() with decorrelate_mid_side fused into the write-back: blocks 2p / 2p+1 = channels of pair p.
        Device buffers, in place."""
        nb, bs = int(buf.shape[0]), int(buf.shape[1])
        self.ctx._call(self.ctx.lib.dll.symaccel_alac_predict_stereo_device, _ptr(buf), _ptr(desc), _ptr(coeffs),
                       _ptr(pair_weight), _ptr(pair_shift), nb, bs)
        return buf

    def predict_strided(self, buf, desc, coeffs, blocksize, pair_weight=None, pair_shift=None):
        """predict() / predict_stereo() over padded rows: buf[n_blocks, stride] i32 on the device
        (symaccel_alac_predict_strided_device).  In place."""
        nb, stride = int(buf.shape[0]), int(buf.shape[1])
        self.ctx._call(self.ctx.lib.dll.symaccel_alac_predict_strided_device, _ptr(buf), _ptr(desc), _ptr(coeffs),
                       _ptr(pair_weight) if pair_weight is not None else None,
                       _ptr(pair_shift) if pair_shift is not None else None, nb, int(blocksize), stride)
        return buf

    def mid_side(self, weight, shift, ch0, ch1):
        """ch0/ch1[n_pairs, blocksize]; weight[n_pairs] i32, shift[n_pairs] u8.  numpy: returns new arrays; torch: in place."""
        d = self.ctx.lib.dll
        n_pairs, bs = int(ch0.shape[0]), int(ch0.shape[1])
        if _is_torch(ch0):
            self.ctx._call(d.symaccel_alac_mid_side_device, _ptr(weight), _ptr(shift), _ptr(ch0), _ptr(ch1), n_pairs, bs)
            return ch0, ch1
        a = np.array(ch0, dtype=np.int32, copy=True, order="C")
        b = np.array(ch1, dtype=np.int32, copy=True, order="C")
        self.ctx._call(d.symaccel_alac_mid_side, _ptr(_np(weight, np.int32)), _ptr(_np(shift, np.uint8)), _ptr(a), _ptr(b),
                       n_pairs, bs)
        return a, b


# ---- host-side tools (libm-dependent stages that stay on the CPU) and per-record status arrays ----------------------

AAC_PULSE_DTYPE = np.dtype([("frame", np.uint32), ("number_pulse", np.uint8), ("pulse_start_sfb", np.uint8), ("pulse_offset", np.uint8, (4,)),
                            ("pulse_amp", np.uint8, (4,)), ("pad", np.uint8, (2,)), ("scales0", np.float32, (64,))])
assert AAC_PULSE_DTYPE.itemsize == 272


def aac_pulse(coeffs, pulses, swb_long, library=None):
    """Pulse::synth (aac/ics/pulse.rs:64-105) on HOST spectra coeffs[n_frames, 1024] (in place; numpy float32, C order).
    pulses: AAC_PULSE_DTYPE records; swb_long: n_swb + 1 offsets."""
    lib = library if library is not None else _ffi.default_library()
    assert isinstance(coeffs, np.ndarray) and coeffs.dtype == np.float32 and coeffs.flags.c_contiguous and coeffs.shape[-1] == 1024
    p = np.ascontiguousarray(pulses, dtype=AAC_PULSE_DTYPE)
    swb = _np(swb_long, np.uint16)
    lib.check(lib.dll.symaccel_host_aac_pulse(_ptr(coeffs), coeffs.size // 1024, _ptr(p), p.size, _ptr(swb), swb.size - 1))
    return coeffs


def vorbis_bark_map(n, rate, bark_map_size, library=None):
    """bark_map (vorbis floor.rs:358-376), n = blocksize / 2."""
    lib = library if library is not None else _ffi.default_library()
    out = np.zeros(int(n), np.int32)
    lib.check(lib.dll.symaccel_host_vorbis_bark_map(int(n), int(rate), int(bark_map_size), _ptr(out)))
    return out


def vorbis_floor0_coeffs(angles, library=None):
    """coeff = 2 cos(coeff) (the end of Floor0::read_channel, floor.rs:246-248)."""
    lib = library if library is not None else _ffi.default_library()
    c = np.array(angles, dtype=np.float32, copy=True)
    lib.check(lib.dll.symaccel_host_vorbis_floor0_coeffs(_ptr(c), c.size))
    return c


def vorbis_floor0(coeffs, bark_map, bark_map_size, amplitude_bits, amplitude_offset, amplitude, library=None):
    """Floor0::synthesis (floor.rs:262-340) for one channel-block on the host; raises SymaccelError (status ERR_DECODE)
    where the reference returns decode_error("vorbis: invalid floor0 coefficients")."""
    lib = library if library is not None else _ffi.default_library()
    c = _np(coeffs, np.float32)
    m = _np(bark_map, np.int32)
    out = np.zeros(m.size, np.float32)
    lib.check(lib.dll.symaccel_host_vorbis_floor0(_ptr(c), c.size, _ptr(m), m.size, int(bark_map_size), int(amplitude_bits),
                                                  int(amplitude_offset), int(amplitude), _ptr(out)))
    return out


def flac_block_status(ctx, desc, blocksize, status):
    """desc[n] FLAC_DESC_DTYPE (device), status[n] int8 (device): what the reference would have answered per subframe."""
    n = (desc.numel() * desc.element_size() if _is_torch(desc) else desc.nbytes) // 4
    ctx._call(ctx.lib.dll.symaccel_flac_block_status_device, _ptr(desc), n, int(blocksize), _ptr(status))
    return status


def vorbis_floor1_status(ctx, n_posts, y, count, status):
    """y[count][n_posts] uint32 (device), status[count] int8 (device): 0, or ERR_UNSUPPORTED for a block with a y value above
    255 (outside the floor-1 kernels' arithmetic; the reference's i32 arithmetic still covers it)."""
    ctx._call(ctx.lib.dll.symaccel_vorbis_floor1_status_device, int(n_posts), _ptr(y), int(count), _ptr(status))
    return status


def alac_block_status(ctx, desc, status):
    n = (desc.numel() * desc.element_size() if _is_torch(desc) else desc.nbytes) // 4
    ctx._call(ctx.lib.dll.symaccel_alac_block_status_device, _ptr(desc), n, _ptr(status))
    return status


def aac_tns_status(ctx, n_frames, filters, status):
    n = (filters.numel() * filters.element_size() if _is_torch(filters) else filters.nbytes) // 92
    ctx._call(ctx.lib.dll.symaccel_aac_tns_status_device, int(n_frames), _ptr(filters), n, _ptr(status))
    return status


class PinnedBuffer:
    """Page-locked host memory from symaccel_host_alloc, exposed as a numpy array (for the staged host entry points)."""

    def __init__(self, shape, dtype, library=None):
        self.lib = library if library is not None else _ffi.default_library()
        self.shape, self.dtype = tuple(int(v) for v in shape), np.dtype(dtype)
        nbytes = int(np.prod(self.shape)) * self.dtype.itemsize
        p = C.c_void_p()
        self.lib.check(self.lib.dll.symaccel_host_alloc(max(nbytes, 16), C.byref(p)))
        self.ptr = p
        self.array = np.frombuffer((C.c_char * max(nbytes, 1)).from_address(p.value), dtype=self.dtype, count=int(np.prod(self.shape))).reshape(self.shape)

    def free(self):
        if self.ptr is not None:
            self.array = None
            self.lib.dll.symaccel_host_free(self.ptr)
            self.ptr = None


BATCH_AAC_SYNTH, BATCH_MP3_SYNTH, BATCH_MP3_DECODE, BATCH_VORBIS_SYNTH, BATCH_AAC_DECODE = 1, 2, 3, 4, 5
BATCH_VORBIS_DECODE, BATCH_FLAC_RESTORE, BATCH_ALAC_PREDICT = 6, 7, 8
BATCH_MAX_INPUTS = 6


class BatchSlot(C.Structure):
    """symaccel_batch_slot (include/symaccel.h)"""
    _fields_ = [("input", C.c_void_p * 6), ("state", C.c_void_p * 3), ("out", C.c_void_p), ("input_bytes", C.c_size_t * 6),
                ("state_bytes", C.c_size_t * 3), ("out_bytes", C.c_size_t)]


class BatcherStats(C.Structure):
    _fields_ = [(n, C.c_uint64) for n in ("submissions", "launches", "chunks", "chains_launched", "max_chains_per_launch", "staging_bytes",
                                           "pending", "failed_tickets", "lanes", "mutex_wait_ns", "mutex_contended", "launch_host_ns",
                                           "lane_wait_ns", "launch_api_ns", "group_allocs", "flag_wait_ns", "slots_peak", "blocks", "commit_to_launch_ns", "waits", "waits_blocked", "launch_to_done_ns", "launches_timed")]


def _slot_view(ptr, nbytes, dtype, shape):
    if not ptr or not nbytes:
        return None
    return np.frombuffer((C.c_char * nbytes).from_address(ptr), dtype=dtype).reshape(shape)


class Batcher:
    """The cross-stream batcher (csrc/batcher.cpp, `symaccel_batcher_*`): decoders of one process submit their look-ahead
    batches, one launch per (kind, units per chain) group serves all of them.  `submit` / `collect` copy from and into the
    caller's arrays; `reserve` / `commit` / `wait` / `release` expose the page-locked slot itself."""

    _PLANES = {  # kind -> (input dtypes/shapes per unit, state dtypes/shapes, out shape per unit)
        BATCH_AAC_SYNTH: ([(np.float32, (1024,)), (np.uint8, ())], [(np.float32, (1024,))], (np.float32, (1024,))),
        BATCH_MP3_SYNTH: ([(np.float32, (576,)), (np.dtype("u1,u1,u2"), ())], [(np.float32, (576,)), (np.float32, (1024,)), (np.int32, ())],
                          (np.float32, (576,))),
    }

    def __init__(self, ctx, flush_bytes=0):
        self.ctx = ctx
        self.dll = ctx.lib.dll
        h = C.c_void_p()
        ctx.lib.check(self.dll.symaccel_batcher_create(ctx.handle, int(flush_bytes), C.byref(h)), ctx.handle)
        self.handle = h
        import weakref
        ctx._batchers.append(weakref.ref(self))

    def close(self):
        if getattr(self, "handle", None):
            self.dll.symaccel_batcher_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:  # noqa: BLE001
            pass

    def _check(self, st):
        return self.ctx.lib.check(st, self.ctx.handle)

    def submit(self, kind, param, inputs, states, out):
        """inputs / states / out: C-contiguous numpy arrays ([chain][unit]... / [chain]...); the states are updated and `out` is
        filled by collect()."""
        n_chains, units = int(out.shape[0]), int(out.shape[1])
        ins = (C.c_void_p * 6)(*[a.ctypes.data if a is not None else None for a in list(inputs) + [None] * (6 - len(inputs))])
        sts = (C.c_void_p * 3)(*[a.ctypes.data for a in list(states)] + [None] * (3 - len(states)))
        for a in list(inputs) + list(states) + [out]:
            assert a is None or a.flags["C_CONTIGUOUS"]
        t = C.c_uint64()
        self._check(self.dll.symaccel_batcher_submit(self.handle, int(kind), int(param), n_chains, units, ins, sts, out.ctypes.data, C.byref(t)))
        return int(t.value)

    def collect(self, ticket):
        self._check(self.dll.symaccel_batcher_collect(self.handle, int(ticket)))

    def aac_bands(self, swb_long, swb_short):
        """register a stream's scale-factor-band offset tables (n + 1 offsets each): the `bands` index of submit_aac_decode"""
        lo, sh = np.ascontiguousarray(swb_long, np.uint16), np.ascontiguousarray(swb_short, np.uint16)
        out = C.c_int(-1)
        self._check(self.dll.symaccel_batcher_aac_bands(self.handle, lo.ctypes.data, len(lo) - 1, sh.ctypes.data, len(sh) - 1, C.byref(out)))
        return int(out.value)

    def submit_aac_decode(self, bands, coeffs, side, pairs, js_desc, tns, delay_io, pcm):
        """symaccel_batcher_submit_aac_decode: one stream's batch as symaccel_aac_decode_pipelined takes it (pairs [n][2] i32 or None,
        js_desc [pair][frame] records or None, tns filters or None); delay_io / pcm are written by collect()"""
        n_chains, frames = int(coeffs.shape[0]), int(coeffs.shape[1])
        n_pairs = 0 if pairs is None else len(pairs)
        n_tns = 0 if tns is None else len(tns)
        keep = [np.ascontiguousarray(pairs, np.int32) if n_pairs else None, np.ascontiguousarray(js_desc) if n_pairs else None,
                np.ascontiguousarray(tns) if n_tns else None]
        t = C.c_uint64()
        self._check(self.dll.symaccel_batcher_submit_aac_decode(
            self.handle, int(bands), coeffs.ctypes.data, side.ctypes.data, keep[0].ctypes.data if n_pairs else None,
            keep[1].ctypes.data if n_pairs else None, n_pairs, keep[2].ctypes.data if n_tns else None, n_tns, delay_io.ctypes.data,
            pcm.ctypes.data, n_chains, frames, C.byref(t)))
        return int(t.value)

    def configure(self, lanes=0, hint_bytes=0):
        self._check(self.dll.symaccel_batcher_configure(self.handle, int(lanes), int(hint_bytes)))

    def last_error(self):
        buf = C.create_string_buffer(512)
        self._check(self.dll.symaccel_batcher_last_error(self.handle, buf, 512))
        return buf.value.decode(errors="replace")

    def vorbis_floor(self, multiplier, x_list):
        """register a floor-1 configuration (floor.rs:510-555): the index VORBIS_DECODE submissions put in their `floor` plane"""
        cfg = np.zeros(1, VORBIS_FLOOR1_DTYPE)
        cfg["multiplier"], cfg["n_posts"] = int(multiplier), len(x_list)
        cfg["x_list"][0, :len(x_list)] = np.asarray(x_list, np.uint32)
        out = C.c_int(-1)
        self._check(self.dll.symaccel_batcher_vorbis_floor(self.handle, cfg.ctypes.data, C.byref(out)))
        return int(out.value)

    def submit_vorbis_decode(self, bs0_exp, bs1_exp, residue, flags, floor, posts, coupling, coupling_first, prev_flag_io, overlap_io, pcm):
        """symaccel_batcher_submit_vorbis_decode: ONE stream's batch as symaccel_vorbis_decode takes it -- residue / pcm
        [chain][blocks * bs1 / 2] f32 (packed at the front), flags / floor [chain][blocks] u8, posts [chain][blocks][65] u32,
        coupling [steps][2] u8 (or None), coupling_first [blocks + 1] u32; prev_flag_io / overlap_io / pcm are written by collect()"""
        n_chains, blocks = int(flags.shape[0]), int(flags.shape[1])
        for a in (residue, flags, floor, posts, coupling_first, prev_flag_io, overlap_io, pcm):
            assert a.flags["C_CONTIGUOUS"]
        keep = np.ascontiguousarray(coupling, np.uint8) if coupling is not None and len(coupling) else None
        t = C.c_uint64()
        self._check(self.dll.symaccel_batcher_submit_vorbis_decode(
            self.handle, int(bs0_exp), int(bs1_exp), residue.ctypes.data, flags.ctypes.data, floor.ctypes.data, posts.ctypes.data,
            keep.ctypes.data if keep is not None else None, coupling_first.ctypes.data, prev_flag_io.ctypes.data, overlap_io.ctypes.data,
            pcm.ctypes.data, n_chains, blocks, C.byref(t)))
        return int(t.value)

    def submit_flac_restore(self, buf_io, desc, coeffs, pair_mode=None, out_shift=0):
        """symaccel_batcher_submit_flac_restore: buf_io [block][blocksize] i32 (in place at collect()), desc [block], coeffs [block][32];
        pair_mode [block / 2] u8 selects the fused stereo form"""
        n_blocks, blocksize = int(buf_io.shape[0]), int(buf_io.shape[1])
        for a in (buf_io, desc, coeffs):
            assert a.flags["C_CONTIGUOUS"]
        t = C.c_uint64()
        self._check(self.dll.symaccel_batcher_submit_flac_restore(
            self.handle, buf_io.ctypes.data, desc.ctypes.data, coeffs.ctypes.data, pair_mode.ctypes.data if pair_mode is not None else None,
            int(out_shift), n_blocks, blocksize, C.byref(t)))
        return int(t.value)

    def submit_alac_predict(self, buf_io, desc, coeffs, pair_weight=None, pair_shift=None):
        """symaccel_batcher_submit_alac_predict: as submit_flac_restore; pair_weight [block / 2] i32 + pair_shift [block / 2] u8 select the
        fused mid/side form"""
        n_blocks, blocksize = int(buf_io.shape[0]), int(buf_io.shape[1])
        for a in (buf_io, desc, coeffs):
            assert a.flags["C_CONTIGUOUS"]
        t = C.c_uint64()
        self._check(self.dll.symaccel_batcher_submit_alac_predict(
            self.handle, buf_io.ctypes.data, desc.ctypes.data, coeffs.ctypes.data, pair_weight.ctypes.data if pair_weight is not None else None,
            pair_shift.ctypes.data if pair_shift is not None else None, n_blocks, blocksize, C.byref(t)))
        return int(t.value)

    def reserve(self, kind, param, n_chains, units):
        slot, t = BatchSlot(), C.c_uint64()
        self._check(self.dll.symaccel_batcher_reserve(self.handle, int(kind), int(param), int(n_chains), int(units), C.byref(slot), C.byref(t)))
        return int(t.value), slot

    def commit(self, ticket):
        self._check(self.dll.symaccel_batcher_commit(self.handle, int(ticket)))

    def wait(self, ticket):
        slot = BatchSlot()
        self._check(self.dll.symaccel_batcher_wait(self.handle, int(ticket), C.byref(slot)))
        return slot

    def release(self, ticket):
        self._check(self.dll.symaccel_batcher_release(self.handle, int(ticket)))

    def flush(self):
        self._check(self.dll.symaccel_batcher_flush(self.handle))

    def stats(self):
        s = BatcherStats()
        self._check(self.dll.symaccel_batcher_get_stats(self.handle, C.byref(s)))
        return {n: int(getattr(s, n)) for n, _ in BatcherStats._fields_}

    @staticmethod
    def slot_arrays(slot, kind, n_chains, units):
        """numpy views of a slot's planes (no copies): (inputs, states, out)"""
        if kind == BATCH_MP3_DECODE:
            ins = [_slot_view(slot.input[0], slot.input_bytes[0], np.int16, (n_chains, units, 576)),
                   _slot_view(slot.input[1], slot.input_bytes[1], MP3_REQUANT_DTYPE, (n_chains, units)),
                   _slot_view(slot.input[2], slot.input_bytes[2], np.dtype("u1,u1,u2"), (n_chains, units)),
                   _slot_view(slot.input[3], slot.input_bytes[3], MP3_STEREO_DTYPE, (units,))]
            sts_spec, out_spec = Batcher._PLANES[BATCH_MP3_SYNTH][1], Batcher._PLANES[BATCH_MP3_SYNTH][2]
        else:
            in_spec, sts_spec, out_spec = Batcher._PLANES[kind]
            ins = [_slot_view(slot.input[i], slot.input_bytes[i], dt, (n_chains, units) + sh) for i, (dt, sh) in enumerate(in_spec)]
        sts = [_slot_view(slot.state[i], slot.state_bytes[i], dt, (n_chains,) + sh) for i, (dt, sh) in enumerate(sts_spec)]
        out = _slot_view(slot.out, slot.out_bytes, out_spec[0], (n_chains, units) + out_spec[1])
        return ins, sts, out
