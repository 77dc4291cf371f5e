"""ctypes front-end of the CPU oracle (oracle/symoracle.c).

TEST INFRASTRUCTURE ONLY.  Importable from tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg; the product package (symphonia_amd) never imports
this module (tests/test_layout.py enforces that).
"""
import ctypes as C
import os
import subprocess
from pathlib import Path

import numpy as np

_HERE = Path(__file__).resolve().parent
_SO = _HERE / "libsymoracle.so"


_SO_NATIVE = _HERE / "libsymoracle_native.so"


def build(force=False):
    src_m = max((_HERE / f).stat().st_mtime for f in ("symoracle.c", "bench_mt.c", "cpu_simd.c", "symoracle.h", "spec_tables.h", "Makefile"))
    if force or not _SO.exists() or _SO.stat().st_mtime < src_m or not _SO_NATIVE.exists() or _SO_NATIVE.stat().st_mtime < src_m:
        subprocess.run(["make", "-C", str(_HERE)] + (["-B"] if force else []), check=True, stdout=subprocess.DEVNULL)
    return _SO


_lib = None


def _bind(path):
    l = C.CDLL(str(path))
    l.so_bench_mt.restype = C.c_double
    l.so_bench_mt.argtypes = [C.c_int, C.c_int, C.c_double, C.POINTER(C.c_long), C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t,
                              C.c_size_t, C.c_size_t, C.c_size_t, C.c_int, C.c_int]
    return l


_native = None


def native_lib():
    """The -O3 -march=native build of the same sources (CPU baseline only).  Built ON the machine it runs on: a library
    built for another CPU model is rebuilt (the GPU box is not the build container)."""
    global _native
    if _native is None:
        build()
        stamp = _HERE / "libsymoracle_native.cpu"
        model = cpu_model()
        if not stamp.exists() or stamp.read_text() != model:
            subprocess.run(["make", "-C", str(_HERE), "-B", "libsymoracle_native.so"], check=True, stdout=subprocess.DEVNULL)
            stamp.write_text(model)
        _native = _bind(_SO_NATIVE)
    return _native


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown CPU"


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = C.CDLL(str(_SO))
        _lib.so_imdct_new.restype = C.c_void_p
        _lib.so_imdct_new.argtypes = [C.c_int, C.c_double]
        _lib.so_flac_rice_signed_to_i32.restype = C.c_int32
        _lib.so_flac_rice_signed_to_i32.argtypes = [C.c_uint32]
        _lib.so_bench_mt.restype = C.c_double
        _lib.so_bench_mt.argtypes = [C.c_int, C.c_int, C.c_double, C.POINTER(C.c_long), C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t,
                                     C.c_size_t, C.c_size_t, C.c_size_t, C.c_int, C.c_int]
        _lib.so_mp3_reorder.restype = C.c_int
        _lib.so_mp3_antialias.restype = C.c_int
    return _lib


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


# ---- core ------------------------------------------------------------------

def fft(x):
    """Fft::fft on a complex64 vector (power-of-two length)."""
    x = np.ascontiguousarray(x, dtype=np.complex64)
    y = np.empty_like(x)
    lib().so_fft(_p(x), _p(y), C.c_int(x.size))
    return y


def ifft(x):
    """Ifft::ifft on a complex64 vector (power-of-two length; below 32 points the reference only permutes and scales)."""
    x = np.ascontiguousarray(x, dtype=np.complex64)
    y = np.empty_like(x)
    lib().so_ifft(_p(x), _p(y), C.c_int(x.size))
    return y


def fft_inplace(x):
    y = np.array(x, dtype=np.complex64, copy=True)
    lib().so_fft_inplace(_p(y), C.c_int(y.size))
    return y


def fft_twiddles(n):
    w = np.empty(n // 2, dtype=np.complex64)
    lib().so_fft_twiddles(C.c_int(n), _p(w))
    return w


def imdct(spec, scale=1.0):
    """Imdct::new_scaled(n, scale).imdct over spec[..., n] -> [..., 2n]."""
    spec = _f32(spec)
    n = spec.shape[-1]
    count = spec.size // n
    out = np.empty(spec.shape[:-1] + (2 * n,), dtype=np.float32)
    lib().so_imdct_batch(C.c_int(n), C.c_double(scale), _p(spec), _p(out), C.c_size_t(count))
    return out


def imdct_twiddles(n, scale=1.0):
    w = np.empty(n // 2, dtype=np.complex64)
    lib().so_imdct_twiddles(C.c_int(n), C.c_double(scale), _p(w))
    return w


def fft_small_twiddles(n):
    w = np.empty(n // 2, dtype=np.complex64)
    lib().so_fft_small_twiddles(C.c_int(n), _p(w))
    return w


def mp3_constants():
    w = np.empty(117, dtype=np.float32)
    lib().so_mp3_constants(_p(w))
    return {"dct_iv_scale": w[0:18], "sdct18_scale": w[18:27], "sdct9_d": w[27:34],
            "cos_16": w[34:50], "cos_8": w[50:58], "cos_4": w[58:62], "cos_2": w[62:64],
            "cos_1": w[64:65], "half_cos_12": w[65:101].reshape(6, 6), "cs": w[101:109],
            "ca": w[109:117]}


def mp3_sfb_tables(sr):
    t = np.empty(81, dtype=np.int32)
    lib().so_mp3_sfb_tables(C.c_int(sr), _p(t))
    mixed = t[40:80]
    return t[:40].copy(), mixed[mixed >= 0].copy(), int(t[80])


def vorbis_floor1_table():
    w = np.empty(256, dtype=np.float32)
    lib().so_vorbis_floor1_table(_p(w))
    return w


# ---- AAC -------------------------------------------------------------------

def aac_window(kbd, alpha, size):
    w = np.empty(size, dtype=np.float32)
    lib().so_aac_window(C.c_int(int(kbd)), C.c_float(alpha), C.c_int(size), _p(w))
    return w


def aac_side(seq, shape, prev_shape):
    return (np.asarray(seq, np.uint8) & 3) | (np.asarray(shape, np.uint8) << 2) | (
        np.asarray(prev_shape, np.uint8) << 3)


def aac_synth(coeffs, side, delay):
    """coeffs[chains, frames, 1024], side[chains, frames] u8, delay[chains, 1024].
    Returns (pcm[chains, frames, 1024], new_delay)."""
    coeffs = _f32(coeffs)
    side = np.ascontiguousarray(side, dtype=np.uint8)
    delay = np.array(delay, dtype=np.float32, copy=True, order="C")
    nch, nfr = coeffs.shape[0], coeffs.shape[1]
    pcm = np.empty((nch, nfr, 1024), dtype=np.float32)
    lib().so_aac_synth_batch(_p(coeffs), _p(side), _p(delay), _p(pcm), C.c_size_t(nch),
                             C.c_size_t(nfr))
    return pcm, delay


# ---- MP3 -------------------------------------------------------------------

def mp3_imdct_windows():
    w = np.empty((4, 36), dtype=np.float32)
    lib().so_mp3_imdct_windows(_p(w))
    return w


def mp3_synthesis_window():
    w = np.empty(512, dtype=np.float32)
    lib().so_mp3_synthesis_window(_p(w))
    return w


def mp3_imdct36(x18, window36, overlap18):
    x = np.array(x18, dtype=np.float32, copy=True)
    ov = np.array(overlap18, dtype=np.float32, copy=True)
    w = _f32(window36)
    lib().so_mp3_imdct36(_p(x), _p(w), _p(ov))
    return x, ov


def mp3_imdct12_win(x18, window36, overlap18):
    x = np.array(x18, dtype=np.float32, copy=True)
    ov = np.array(overlap18, dtype=np.float32, copy=True)
    w = _f32(window36)
    lib().so_mp3_imdct12_win(_p(x), _p(w), _p(ov))
    return x, ov


def mp3_dct32(x32):
    x = _f32(x32)
    y = np.empty(32, dtype=np.float32)
    lib().so_mp3_dct32(_p(x), _p(y))
    return y


def mp3_reorder(buf, block_type, is_mixed, sr_idx, rzero):
    b = np.array(buf, dtype=np.float32, copy=True)
    rz = lib().so_mp3_reorder(_p(b), C.c_int(block_type), C.c_int(int(is_mixed)), C.c_int(sr_idx),
                              C.c_int(rzero))
    return b, rz


def mp3_antialias(buf, block_type, is_mixed, rzero):
    b = np.array(buf, dtype=np.float32, copy=True)
    rz = lib().so_mp3_antialias(_p(b), C.c_int(block_type), C.c_int(int(is_mixed)), C.c_int(rzero))
    return b, rz


def mp3_hybrid(buf, overlap, block_type, is_mixed, rzero):
    b = np.array(buf, dtype=np.float32, copy=True)
    ov = np.array(overlap, dtype=np.float32, copy=True)
    lib().so_mp3_hybrid(_p(b), _p(ov), C.c_int(block_type), C.c_int(int(is_mixed)), C.c_int(rzero))
    return b, ov


def mp3_frequency_inversion(buf):
    b = np.array(buf, dtype=np.float32, copy=True)
    lib().so_mp3_frequency_inversion(_p(b))
    return b


def mp3_polyphase(v_vec, v_front, n_frames, samples):
    v = np.array(v_vec, dtype=np.float32, copy=True)
    vf = C.c_int(v_front)
    s = _f32(samples)
    out = np.empty(32 * n_frames, dtype=np.float32)
    lib().so_mp3_polyphase(_p(v), C.byref(vf), C.c_int(n_frames), _p(s), _p(out))
    return out, v, vf.value


def mp3_side(block_type, is_mixed, rzero):
    bt = np.asarray(block_type, np.uint8)
    side = np.zeros(bt.shape + (4,), dtype=np.uint8)
    side[..., 0] = bt
    side[..., 1] = np.asarray(is_mixed, np.uint8)
    rz = np.asarray(rzero, np.uint16)
    side[..., 2] = rz & 0xFF
    side[..., 3] = rz >> 8
    return side


def mp3_synth(xr, side, sr_idx, overlap, v_vec, v_front):
    """xr[chains, granules, 576], side[chains, granules, 4] u8.
    Returns (pcm, overlap', v_vec', v_front')."""
    xr = _f32(xr)
    side = np.ascontiguousarray(side, dtype=np.uint8)
    nch, ngr = xr.shape[0], xr.shape[1]
    ov = np.array(overlap, dtype=np.float32, copy=True, order="C")
    vv = np.array(v_vec, dtype=np.float32, copy=True, order="C")
    vf = np.array(v_front, dtype=np.int32, copy=True, order="C")
    pcm = np.empty((nch, ngr, 576), dtype=np.float32)
    lib().so_mp3_synth_batch(_p(xr), _p(side), C.c_int(sr_idx), _p(ov), _p(vv), _p(vf), _p(pcm),
                             C.c_size_t(nch), C.c_size_t(ngr))
    return pcm, ov, vv, vf


# ---- Vorbis ----------------------------------------------------------------

def vorbis_window(bs):
    w = np.empty(bs // 2, dtype=np.float32)
    lib().so_vorbis_window(C.c_int(bs), _p(w))
    return w


def vorbis_inverse_coupling(mag, ang):
    m = np.array(mag, dtype=np.float32, copy=True)
    a = np.array(ang, dtype=np.float32, copy=True)
    lib().so_vorbis_inverse_coupling(_p(m), _p(a), C.c_size_t(m.size))
    return m, a


def vorbis_dot_product(floor, residue):
    f = np.array(floor, dtype=np.float32, copy=True)
    r = _f32(residue)
    lib().so_vorbis_dot_product(_p(f), _p(r), C.c_size_t(f.size))
    return f


def vorbis_deinterleave2(type2, n_ch):
    t = _f32(type2)
    n2 = t.size // n_ch
    out = np.empty((n_ch, n2), dtype=np.float32)
    lib().so_vorbis_deinterleave2(_p(t), _p(out), C.c_int(n_ch), C.c_size_t(n2))
    return out


def vorbis_floor1(x_list, y, multiplier, n):
    x = np.ascontiguousarray(x_list, dtype=np.uint32)
    yy = np.ascontiguousarray(y, dtype=np.uint32)
    out = np.zeros(n, dtype=np.float32)
    lib().so_vorbis_floor1(_p(x), _p(yy), C.c_int(x.size), C.c_int(multiplier), C.c_uint32(n), _p(out))
    return out


def vorbis_layout(bs0_exp, bs1_exp, block_flag, prev_flag):
    """Per-chain packed offsets. block_flag[chains, blocks] (0/1), prev_flag[chains]
    (-1 none). Returns (spec_off[chains, blocks+1], pcm_off[chains, blocks+1])."""
    bf = np.asarray(block_flag).astype(np.int64)
    nch, nb = bf.shape
    bs = np.where(bf > 0, 1 << bs1_exp, 1 << bs0_exp)
    pf = np.empty_like(bf)
    first = np.asarray(prev_flag).astype(np.int64)
    pf[:, 0] = np.where(first < 0, bf[:, 0], first)
    pf[:, 1:] = bf[:, :-1]
    prev_n = np.where(pf > 0, 1 << bs1_exp, 1 << bs0_exp)
    spec_off = np.zeros((nch, nb + 1), dtype=np.int64)
    pcm_off = np.zeros((nch, nb + 1), dtype=np.int64)
    spec_off[:, 1:] = np.cumsum(bs // 2, axis=1)
    pcm_off[:, 1:] = np.cumsum((prev_n + bs) // 4, axis=1)
    return spec_off, pcm_off


def vorbis_synth(bs0_exp, bs1_exp, spectra, block_flag, prev_flag, overlap, pcm_stride):
    """spectra[chains, spec_stride] packed; returns (pcm[chains, pcm_stride], overlap', prev_flag')."""
    sp = _f32(spectra)
    bf = np.ascontiguousarray(block_flag, dtype=np.uint8)
    nch, nb = bf.shape
    pf = np.array(prev_flag, dtype=np.int32, copy=True, order="C")
    ov = np.array(overlap, dtype=np.float32, copy=True, order="C")
    pcm = np.zeros((nch, pcm_stride), dtype=np.float32)
    lib().so_vorbis_synth_batch(C.c_int(bs0_exp), C.c_int(bs1_exp), _p(sp), C.c_size_t(sp.shape[1]),
                                _p(bf), _p(pf), _p(ov), _p(pcm), C.c_size_t(pcm_stride),
                                C.c_size_t(nch), C.c_size_t(nb))
    return pcm, ov, pf


# ---- FLAC ------------------------------------------------------------------

def flac_fixed_predict(order, buf):
    b = np.array(buf, dtype=np.int32, copy=True)
    lib().so_flac_fixed_predict(C.c_int(order), _p(b), C.c_size_t(b.size))
    return b


def flac_lpc_predict(order, coeffs, shift, buf):
    b = np.array(buf, dtype=np.int32, copy=True)
    c = np.ascontiguousarray(coeffs, dtype=np.int32)
    lib().so_flac_lpc_predict(C.c_int(order), _p(c), C.c_uint32(shift), _p(b), C.c_size_t(b.size))
    return b


def flac_decorrelate(mode, ch0, ch1):
    a = np.array(ch0, dtype=np.int32, copy=True)
    b = np.array(ch1, dtype=np.int32, copy=True)
    lib().so_flac_decorrelate(C.c_int(mode), _p(a), _p(b), C.c_size_t(a.size))
    return a, b


def flac_shl(buf, shift):
    b = np.array(buf, dtype=np.int32, copy=True)
    lib().so_flac_shl(_p(b), C.c_size_t(b.size), C.c_uint32(shift))
    return b


def flac_rice_signed_to_i32(word):
    return lib().so_flac_rice_signed_to_i32(C.c_uint32(word))


def flac_desc(kind, order, shift, wasted):
    k = np.asarray(kind, np.uint8)
    d = np.zeros(k.shape + (4,), dtype=np.uint8)
    d[..., 0] = k
    d[..., 1] = np.asarray(order, np.uint8)
    d[..., 2] = np.asarray(shift, np.uint8)
    d[..., 3] = np.asarray(wasted, np.uint8)
    return d


def flac_restore(buf, desc, coeffs):
    """buf[n_blocks, blocksize] i32, desc[n_blocks, 4] u8, coeffs[n_blocks, 32] i32."""
    b = np.array(buf, dtype=np.int32, copy=True, order="C")
    d = np.ascontiguousarray(desc, dtype=np.uint8)
    c = np.ascontiguousarray(coeffs, dtype=np.int32)
    lib().so_flac_restore_batch(_p(b), _p(d), _p(c), C.c_size_t(b.shape[0]), C.c_size_t(b.shape[1]))
    return b


# ---- ALAC ------------------------------------------------------------------

ALAC_DESC_DTYPE = np.dtype([("mode", np.uint8), ("order", np.uint8), ("shift", np.uint8), ("bps", np.uint8)])


def alac_desc(mode, order, shift, bps):
    m = np.asarray(mode)
    d = np.zeros(m.shape, dtype=ALAC_DESC_DTYPE)
    d["mode"], d["order"], d["shift"], d["bps"] = m, np.asarray(order), np.asarray(shift), np.asarray(bps)
    return d


def alac_predict(buf, desc, coeffs):
    """ElementChannel::predict for buf[blocks, blocksize] (copy); desc from alac_desc; coeffs[blocks, 32] i32."""
    out = np.array(buf, dtype=np.int32, copy=True, order="C")
    desc = np.ascontiguousarray(desc)
    co = np.ascontiguousarray(coeffs, dtype=np.int32)
    lib().so_alac_predict_batch(_p(out), _p(desc), _p(co), C.c_size_t(out.shape[0]), C.c_size_t(out.shape[1]))
    return out


def alac_decorrelate_mid_side(out0, out1, weight, shift):
    a = np.array(out0, dtype=np.int32, copy=True)
    b = np.array(out1, dtype=np.int32, copy=True)
    lib().so_alac_decorrelate_mid_side(_p(a), _p(b), C.c_size_t(a.size), C.c_int32(weight), C.c_uint32(shift))
    return a, b


# ---- AAC spectral tools (aac/cpe.rs:110-157, aac/ics/tns.rs:180-195) -----------

def aac_joint_stereo(left, right, num_windows, max_sfb, bands, mode, scale):
    """One channel-pair frame; mode[128] u8 (0 none, 1 M/S, 2 intensity), scale[128] f32. Returns (left', right')."""
    l = np.array(left, dtype=np.float32, copy=True)
    r = np.array(right, dtype=np.float32, copy=True)
    b = np.ascontiguousarray(bands, dtype=np.uint16)
    m = np.ascontiguousarray(mode, dtype=np.uint8)
    sc = np.ascontiguousarray(scale, dtype=np.float32)
    assert l.size == 1024 and r.size == 1024 and m.size == 128 and sc.size == 128 and b.size > max_sfb
    lib().so_aac_joint_stereo(_p(l), _p(r), int(num_windows), int(max_sfb), _p(b), _p(m), _p(sc))
    return l, r


def aac_tns_filter(coeffs, start, end, order, direction, lpc):
    c = np.array(coeffs, dtype=np.float32, copy=True)
    lp = np.zeros(20, np.float32)
    lp[:len(lpc)] = lpc
    lib().so_aac_tns_filter(_p(c), int(start), int(end), int(order), int(direction), _p(lp))
    return c


def aac_iquant_requant(val, scale):
    """iquant(val) and requant(val, scale) of aac/ics/pulse.rs:19-33 for an array of values."""
    v = _f32(val)
    iq, rq = np.empty_like(v), np.empty_like(v)
    lib().so_aac_iquant_requant(_p(v), C.c_float(scale), _p(iq), _p(rq), C.c_size_t(v.size))
    return iq, rq


def aac_pulse(coeffs, bands, scales0, number_pulse, pulse_start_sfb, pulse_offset, pulse_amp):
    """Pulse::synth (aac/ics/pulse.rs:64-105) on one channel-frame; bands = swb offsets (n_swb + 1), scales0 = scales[0]."""
    c = np.array(coeffs, dtype=np.float32, copy=True)
    b = np.ascontiguousarray(bands, dtype=np.int32)
    sc = _f32(scales0)
    off = np.ascontiguousarray(pulse_offset, dtype=np.int32)
    amp = np.ascontiguousarray(pulse_amp, dtype=np.int32)
    lib().so_aac_pulse(_p(c), _p(b), C.c_int(b.size), _p(sc), C.c_int(int(number_pulse)), C.c_int(int(pulse_start_sfb)), _p(off), _p(amp))
    return c


# ---- Vorbis floor 0 (floor.rs:246-248, 262-340, 353-390) --------------------------

def vorbis_bark_map(n, rate, map_size):
    out = np.zeros(int(n), np.int32)
    lib().so_vorbis_bark_map(C.c_uint32(int(n)), C.c_uint32(int(rate)), C.c_uint32(int(map_size)), _p(out))
    return out


def vorbis_floor0_coeffs(angles):
    c = np.array(angles, dtype=np.float32, copy=True)
    lib().so_vorbis_floor0_coeffs(_p(c), C.c_int(c.size))
    return c


def vorbis_floor0(coeffs, bark_map, map_size, amplitude_bits, amplitude_offset, amplitude):
    """Floor0::synthesis for one channel-block: coeffs = 2 cos(lsp) values, bark_map = the map of this block size."""
    c = _f32(coeffs)
    m = np.ascontiguousarray(bark_map, dtype=np.int32)
    out = np.zeros(m.size, np.float32)
    lib().so_vorbis_floor0.restype = C.c_int
    st = lib().so_vorbis_floor0(_p(c), C.c_int(c.size), _p(m), C.c_uint32(m.size), C.c_uint32(int(map_size)), C.c_uint32(int(amplitude_bits)),
                                C.c_uint32(int(amplitude_offset)), C.c_uint64(int(amplitude)), _p(out))
    if st != 0:
        raise ValueError("vorbis: invalid floor0 coefficients")
    return out


# ---- MP3 requantisation (layer3/requantize.rs) ---------------------------------

MP3_REQUANT_DTYPE = np.dtype([("global_gain", np.uint8), ("flags", np.uint8), ("block_type", np.uint8),
                              ("is_mixed", np.uint8), ("subblock_gain", np.uint8, (3,)), ("reserved", np.uint8),
                              ("rzero", np.uint16), ("scalefacs", np.uint8, (39,)), ("pad", np.uint8, (3,))])
assert MP3_REQUANT_DTYPE.itemsize == 52
MP3_RQ_SCALEFAC_SCALE, MP3_RQ_PREFLAG = 1, 2
MP3_POW2AB_MIN_E, MP3_POW2AB_LEN = -1300, 1346


def mp3_sfb_long(sr):
    out = np.zeros(23, np.int32)
    lib().so_mp3_sfb_long(int(sr), _p(out))
    return out


def mp3_pow43():
    out = np.zeros(8207, np.float32)
    lib().so_mp3_pow43(_p(out))
    return out


def mp3_pow2ab():
    out = np.zeros(MP3_POW2AB_LEN, np.float32)
    lib().so_mp3_pow2ab(_p(out))
    return out


def mp3_requantize(quant, desc, sr):
    """read_huffman_samples' value mapping + requantize for quant[n, 576] int16, desc[n] MP3_REQUANT_DTYPE."""
    q = np.ascontiguousarray(quant, dtype=np.int16).reshape(-1, 576)
    d = np.ascontiguousarray(desc, dtype=MP3_REQUANT_DTYPE).reshape(-1)
    assert d.shape[0] == q.shape[0]
    out = np.zeros(q.shape, np.float32)
    lib().so_mp3_requantize_batch(_p(q), _p(d), int(sr), _p(out), C.c_size_t(q.shape[0]))
    return out


MP3_STEREO_DTYPE = np.dtype([("flags", np.uint8), ("block_type", np.uint8), ("is_mixed", np.uint8), ("reserved", np.uint8),
                             ("rzero0", np.uint16), ("rzero1", np.uint16), ("scalefacs1", np.uint8, (39,)), ("pad", np.uint8)])
assert MP3_STEREO_DTYPE.itemsize == 48
MP3_ST_MID_SIDE, MP3_ST_INTENSITY, MP3_ST_MPEG1, MP3_ST_IS_SCALE = 1, 2, 4, 8


def mp3_intensity_ratios():
    m1, m2 = np.zeros((7, 2), np.float32), np.zeros((2, 32, 2), np.float32)
    lib().so_mp3_intensity_ratios(_p(m1), _p(m2))
    return m1, m2


def mp3_stereo(ch0, ch1, desc, sr):
    """stereo() for one granule: ch0/ch1[576], desc: MP3_STEREO_DTYPE scalar.  Returns (ch0', ch1')."""
    a = np.array(ch0, dtype=np.float32, copy=True)
    b = np.array(ch1, dtype=np.float32, copy=True)
    d = np.ascontiguousarray(desc, dtype=MP3_STEREO_DTYPE).reshape(1)
    lib().so_mp3_stereo(_p(a), _p(b), _p(d), int(sr))
    return a, b


# ---- timing driver (bench.py cpu_baseline) ------------------------------------

def aac_long_kbd_batch_simd(coeffs, delay, native=False):
    """cpu_simd.c: ONLY_LONG / KBD frames, 16 chains per vector.  coeffs[chains, frames, 1024], chains % 16 == 0."""
    c = _f32(coeffs)
    d = np.array(delay, dtype=np.float32, copy=True, order="C")
    pcm = np.empty_like(c)
    l = native_lib() if native else lib()
    l.so_aac_long_kbd_batch_simd(_p(c), _p(d), _p(pcm), C.c_size_t(c.shape[0]), C.c_size_t(c.shape[1]))
    return pcm, d


def bench_mt(kind, threads, seconds, in0, in1, in2=None, n_chains=0, per_chain=0, stride_in=0, stride_out=0, p0=0, p1=0, native=False):
    """oracle/bench_mt.c: `threads` pthreads each run the batch on private outputs for `seconds`.
    Returns (elapsed seconds, batches completed by all threads)."""
    k = {"aac": 0, "mp3": 1, "vorbis": 2, "flac": 3, "alac": 4, "aac_simd": 5}[kind]
    reps = C.c_long(0)
    dt = float((native_lib() if native else lib()).so_bench_mt(k, int(threads), float(seconds), C.byref(reps), _p(in0), _p(in1),
                                 _p(in2) if in2 is not None else None, n_chains, per_chain, stride_in, stride_out, p0, p1))
    return dt, int(reps.value)
