"""Error behaviour of the C ABI (include/symaccel.h conventions): what the reference asserts / panics on comes back as
SYMACCEL_ERR_INVALID_ARG, what it would report as Error::Unsupported as SYMACCEL_ERR_UNSUPPORTED, and empty batches are
no-ops.  Exercised through the CPU-emulated build of the library (argument checking is host code, identical in both)."""
import ctypes as C

import numpy as np
import pytest

from emu_lib import emu_ctx, emu_library  # noqa: F401
from symphonia_amd import _ffi


def call(ctx, name, *args):
    return getattr(ctx.lib.dll, name)(ctx.handle, *args)


def buf(n, dtype=np.float32):
    return np.zeros(n, dtype=dtype)


def test_imdct_and_fft_size_checks(emu_ctx):
    x, y = buf(4096), buf(8192)
    for n in (3, 24, 1000, 0, -4):  # mdct.rs:37-40: n must be a power of two
        assert call(emu_ctx, "symaccel_imdct_f32_device", n, 1.0, x.ctypes.data, y.ctypes.data, 1) == _ffi.ERR_INVALID_ARG
    # legal in the reference (mdct.rs:40, no_simd.rs:80) but above what the kernels hold: Unsupported, not a panic-class error
    # (sizes above 4096 / 8192 points used to be SYMACCEL_ERR_UNSUPPORTED; they run the global-memory path now, up to the
    #  reference's own limits of 65 536 / 131 072)
    assert call(emu_ctx, "symaccel_imdct_f32_device", 1 << 18, 1.0, x.ctypes.data, y.ctypes.data, 1) == _ffi.ERR_INVALID_ARG
    assert call(emu_ctx, "symaccel_fft_c32_device", 1 << 17, x.ctypes.data, y.ctypes.data, 1) == _ffi.ERR_INVALID_ARG
    assert call(emu_ctx, "symaccel_fft_c32_device", 6, x.ctypes.data, y.ctypes.data, 1) == _ffi.ERR_INVALID_ARG
    assert call(emu_ctx, "symaccel_imdct_f32_device", 64, 1.0, None, y.ctypes.data, 1) == _ffi.ERR_INVALID_ARG
    assert call(emu_ctx, "symaccel_imdct_f32_device", 64, 1.0, None, None, 0) == _ffi.OK  # empty batch: nothing to do
    # ... also for the sizes that run the workgroup-cooperative kernels
    assert call(emu_ctx, "symaccel_imdct_f32_device", 8192, 1.0, None, None, 0) == _ffi.OK
    assert call(emu_ctx, "symaccel_fft_c32_device", 4096, None, None, 0) == _ffi.OK


def test_codec_entry_points_reject_bad_arguments(emu_ctx):
    f, i8, i32 = buf(4096), buf(64, np.uint8), buf(64, np.int32)
    p = lambda a: a.ctypes.data  # noqa: E731
    assert call(emu_ctx, "symaccel_aac_synth_device", None, p(i8), p(f), p(f), 1, 1) == _ffi.ERR_INVALID_ARG
    assert call(emu_ctx, "symaccel_aac_synth_device", None, None, None, None, 0, 5) == _ffi.OK
    assert call(emu_ctx, "symaccel_mp3_synth_device", p(f), p(i8), 9, p(f), p(f), p(i32), p(f), 1, 1) == _ffi.ERR_INVALID_ARG
    assert call(emu_ctx, "symaccel_mp3_synth_device", p(f), p(i8), -1, p(f), p(f), p(i32), p(f), 1, 1) == _ffi.ERR_INVALID_ARG
    assert call(emu_ctx, "symaccel_mpa_polyphase_device", 18, p(f), p(f), p(i32), p(f), 1, 1) == _ffi.ERR_UNSUPPORTED
    assert call(emu_ctx, "symaccel_mpa_polyphase_device", 12, None, p(f), p(i32), p(f), 1, 1) == _ffi.ERR_INVALID_ARG
    # vorbis/lib.rs:404-406, 461-470: 2^6 <= bs0 <= bs1 <= 2^13
    for bs0, bs1 in ((5, 11), (8, 14), (11, 8)):
        assert call(emu_ctx, "symaccel_vorbis_synth_device", bs0, bs1, p(f), 1024, p(i8), p(i32), p(f), p(f), 1024, 1, 1) == _ffi.ERR_INVALID_ARG
    # empty batches (no chains / no blocks), also for the pair with 8192-sample long blocks (its own kernel)
    for bs0, bs1 in ((8, 11), (10, 13), (7, 10)):
        assert call(emu_ctx, "symaccel_vorbis_synth_device", bs0, bs1, None, 0, None, None, None, None, 0, 0, 5) == _ffi.OK
        assert call(emu_ctx, "symaccel_vorbis_synth_device", bs0, bs1, None, 0, None, None, None, None, 0, 3, 0) == _ffi.OK
    assert call(emu_ctx, "symaccel_vorbis_synth_fr_device", 8, 11, p(f), None, 1024, p(i8), p(i32), p(f), p(f), 1024, 1, 1) == _ffi.ERR_INVALID_ARG
    xs = np.array([0, 128, 64, 64], np.uint32)  # duplicate x: render_line would divide by zero in the reference
    assert call(emu_ctx, "symaccel_vorbis_floor1_device", p(xs), 4, 1, p(i32), 128, p(f), 1) == _ffi.ERR_INVALID_ARG
    xs = np.array([0, 128, 64], np.uint32)
    assert call(emu_ctx, "symaccel_vorbis_floor1_device", p(xs), 3, 5, p(i32), 128, p(f), 1) == _ffi.ERR_INVALID_ARG  # multiplier
    assert call(emu_ctx, "symaccel_vorbis_floor1_device", p(xs), 3, 1, p(i32), 100, p(f), 1) == _ffi.ERR_INVALID_ARG  # n
    assert call(emu_ctx, "symaccel_flac_decorrelate_device", p(i8), p(i32), p(i32), 1, 8, 32) == _ffi.ERR_INVALID_ARG
    assert call(emu_ctx, "symaccel_flac_restore_stereo_device", p(i32), p(i8), p(i32), p(i8), 0, 3, 4) == _ffi.ERR_INVALID_ARG
    assert call(emu_ctx, "symaccel_alac_predict_stereo_device", p(i32), p(i8), p(i32), p(i32), p(i8), 3, 4) == _ffi.ERR_INVALID_ARG
    assert call(emu_ctx, "symaccel_ctx_set_segment", -1) == _ffi.ERR_INVALID_ARG


def test_front_stage_entry_points_reject_bad_arguments(emu_ctx):
    """The rank-1 stages: requantize / stereo (MP3), joint stereo / TNS (AAC)."""
    f, i8, i16, i32 = buf(4096), buf(4096, np.uint8), buf(4096, np.int16), buf(64, np.int32)
    p = lambda a: a.ctypes.data  # noqa: E731
    for sr in (-1, 9):
        assert call(emu_ctx, "symaccel_mp3_requantize_device", p(i16), p(i8), sr, p(f), 1) == _ffi.ERR_INVALID_ARG
        assert call(emu_ctx, "symaccel_mp3_stereo_device", p(f), 1, p(i32), p(i8), sr, 1) == _ffi.ERR_INVALID_ARG
        assert call(emu_ctx, "symaccel_mp3_requantize_stereo_device", p(i16), p(i8), 1, p(i32), p(i8), sr, p(f), 1) == _ffi.ERR_INVALID_ARG
    assert call(emu_ctx, "symaccel_mp3_requantize_device", None, p(i8), 0, p(f), 1) == _ffi.ERR_INVALID_ARG
    assert call(emu_ctx, "symaccel_mp3_requantize_device", None, None, 0, None, 0) == _ffi.OK
    assert call(emu_ctx, "symaccel_mp3_requantize", p(i16), None, 0, p(f), 1) == _ffi.ERR_INVALID_ARG
    assert call(emu_ctx, "symaccel_mp3_stereo_device", None, 1, p(i32), p(i8), 0, 1) == _ffi.ERR_INVALID_ARG
    assert call(emu_ctx, "symaccel_mp3_stereo_device", None, 1, None, None, 0, 0) == _ffi.OK
    assert call(emu_ctx, "symaccel_mp3_requantize_stereo_device", p(i16), p(i8), 1, p(i32), p(i8), 0, None, 1) == _ffi.ERR_INVALID_ARG
    swb_l, swb_s = np.array([0, 4, 1024], np.uint16), np.array([0, 4, 128], np.uint16)
    assert call(emu_ctx, "symaccel_aac_joint_stereo_device", p(f), 1, p(i32), p(i8), 1, None, 2, p(swb_s), 2) == _ffi.ERR_INVALID_ARG
    assert call(emu_ctx, "symaccel_aac_joint_stereo_device", p(f), 1, p(i32), p(i8), 1, p(swb_l), 2, p(swb_s), 17) == _ffi.ERR_INVALID_ARG
    assert call(emu_ctx, "symaccel_aac_joint_stereo_device", None, 1, p(i32), p(i8), 1, p(swb_l), 2, p(swb_s), 2) == _ffi.ERR_INVALID_ARG
    assert call(emu_ctx, "symaccel_aac_joint_stereo_device", None, 1, None, None, 0, p(swb_l), 2, p(swb_s), 2) == _ffi.OK
    assert call(emu_ctx, "symaccel_aac_tns_device", None, 4, p(i8), 1) == _ffi.ERR_INVALID_ARG
    assert call(emu_ctx, "symaccel_aac_tns_device", p(f), 4, None, 1) == _ffi.ERR_INVALID_ARG
    assert call(emu_ctx, "symaccel_aac_tns_device", None, 4, None, 0) == _ffi.OK


def test_null_context_and_error_strings():
    lib = emu_library()
    assert lib.dll.symaccel_sync(None) == _ffi.ERR_INVALID_ARG
    assert lib.dll.symaccel_ctx_create(0, None) == _ffi.ERR_INVALID_ARG
    h = C.c_void_p()
    assert lib.dll.symaccel_ctx_create(-1, C.byref(h)) == _ffi.ERR_INVALID_ARG and not h.value
    with pytest.raises(_ffi.SymaccelError) as e:
        lib.check(_ffi.ERR_UNSUPPORTED)
    assert e.value.status == _ffi.ERR_UNSUPPORTED and "unsupported" in str(e.value).lower()
    assert lib.dll.symaccel_table_f32(None, 99, None, 0) == _ffi.ERR_INVALID_ARG


def test_vorbis_strides_are_checked_against_the_block_flags(emu_ctx):
    """Host-pointer entry point: the packed layout implied by the flags must fit the strides (a stride that is too small
    would let one chain's blocks overwrite the next chain's); device-pointer entry points can only apply the
    all-short-blocks lower bound, the flags being on the device."""
    p = lambda a: a.ctypes.data  # noqa: E731
    flags = np.array([[1, 0, 1, 1]], np.uint8)            # long short long long, 256 / 2048
    prev = np.array([1], np.int32)
    lines = 1024 + 128 + 1024 + 1024
    samples = (2048 + 2048) // 4 + (2048 + 256) // 4 + (256 + 2048) // 4 + (2048 + 2048) // 4
    spec, ov, pcm = buf(lines), buf(1024), buf(samples)
    args = lambda ss, ps: (8, 11, p(spec), ss, p(flags), p(prev), p(ov), p(pcm), ps, 1, 4)  # noqa: E731
    assert call(emu_ctx, "symaccel_vorbis_synth", *args(lines, samples)) == _ffi.OK
    assert call(emu_ctx, "symaccel_vorbis_synth", *args(lines - 1, samples)) == _ffi.ERR_INVALID_ARG
    assert call(emu_ctx, "symaccel_vorbis_synth", *args(lines, samples - 1)) == _ffi.ERR_INVALID_ARG
    prev[0] = -1  # no previous block: the first block still owns (n + n) / 4 slots of the packed layout (symaccel.h)
    assert call(emu_ctx, "symaccel_vorbis_synth", *args(lines, samples - 1)) == _ffi.ERR_INVALID_ARG
    assert call(emu_ctx, "symaccel_vorbis_synth", *args(lines, samples)) == _ffi.OK
    assert call(emu_ctx, "symaccel_vorbis_synth_device", 8, 11, p(spec), 4 * 128 - 1, p(flags), p(prev), p(ov), p(pcm), samples, 1, 4) == _ffi.ERR_INVALID_ARG
