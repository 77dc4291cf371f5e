"""ctypes binding of the symaccel C ABI (include/symaccel.h).

`Library(path)` binds one shared object; `default_library()` binds the hipcc-built
symphonia_amd/libsymaccel.so and raises if it is missing -- there is no CPU path.
"""
import ctypes as C
import json
import os
import sys
from pathlib import Path

import numpy as np

HERE = Path(__file__).resolve().parent
DEFAULT_SO = HERE / "libsymaccel.so"

OK = 0
ERR_INVALID_ARG = -1
ERR_UNSUPPORTED = -2
ERR_DEVICE = -3
ERR_OOM = -4
ERR_DECODE = -5

TABLE_AAC_KBD_LONG, TABLE_AAC_KBD_SHORT, TABLE_AAC_SINE_LONG, TABLE_AAC_SINE_SHORT = 0, 1, 2, 3
TABLE_MP3_SYNTH_D, TABLE_MP3_IMDCT_WIN, TABLE_VORBIS_FLOOR1_DB, TABLE_MP3_CONSTS = 4, 5, 6, 7
TABLE_MP3_POW43, TABLE_MP3_POW2AB = 8, 9

# every symbol include/symaccel.h declares (tests/test_abi.py checks the built library exports all)
ABI_SYMBOLS = [
    "symaccel_abi_version", "symaccel_strerror", "symaccel_last_error", "symaccel_ctx_create",
    "symaccel_ctx_destroy", "symaccel_ctx_set_stream", "symaccel_sync", "symaccel_ctx_set_segment",
    "symaccel_fft_c32_device", "symaccel_fft_c32", "symaccel_ifft_c32_device", "symaccel_ifft_c32", "symaccel_imdct_f32_device", "symaccel_imdct_f32",
    "symaccel_aac_synth_device", "symaccel_aac_synth", "symaccel_aac_joint_stereo_device", "symaccel_aac_tns_device", "symaccel_mp3_synth_device", "symaccel_mp3_synth", "symaccel_mpa_polyphase_device", "symaccel_mpa_polyphase",
    "symaccel_mp3_requantize_device", "symaccel_mp3_requantize", "symaccel_mp3_stereo_device", "symaccel_mp3_requantize_stereo_device",
    "symaccel_vorbis_synth_device", "symaccel_vorbis_synth_fr_device", "symaccel_vorbis_synth", "symaccel_vorbis_inverse_coupling_device",
    "symaccel_vorbis_dot_product_device", "symaccel_vorbis_deinterleave2_device",
    "symaccel_vorbis_floor1_device", "symaccel_vorbis_floor1_dot_device", "symaccel_vorbis_floor1_y_device", "symaccel_vorbis_floor1_dot_at_device", "symaccel_vorbis_floor1_y_jobs_device",
    "symaccel_vorbis_synth_fy_pp_device", "symaccel_vorbis_synth_fy_device", "symaccel_flac_restore_device", "symaccel_flac_restore", "symaccel_flac_restore_stereo_device",
    "symaccel_flac_decorrelate_device", "symaccel_flac_decorrelate", "symaccel_alac_predict_device",
    "symaccel_alac_predict", "symaccel_alac_predict_stereo_device", "symaccel_alac_mid_side_device", "symaccel_alac_mid_side", "symaccel_table_f32", "symaccel_imdct_twiddles",
    "symaccel_fft_twiddles",
    "symaccel_host_alloc", "symaccel_host_free", "symaccel_host_register", "symaccel_host_unregister",
    "symaccel_aac_synth_pipelined", "symaccel_mp3_synth_pipelined", "symaccel_flac_restore_pipelined",
    "symaccel_host_aac_pulse", "symaccel_host_vorbis_bark_map", "symaccel_host_vorbis_floor0_coeffs", "symaccel_host_vorbis_floor0",
    "symaccel_flac_block_status_device", "symaccel_alac_block_status_device", "symaccel_vorbis_floor1_status_device", "symaccel_aac_tns_status_device",
    "symaccel_aac_synth_pp_device", "symaccel_aac_synth_js_pp_device", "symaccel_aac_synth_js_device", "symaccel_mp3_synth_pp_device", "symaccel_vorbis_synth_pp_device", "symaccel_mpa_polyphase_pp_device",
    "symaccel_probe_copy_device",
    "symaccel_shard_range", "symaccel_scatter_streams", "symaccel_gather_streams", "symaccel_exchange_pipelined", "symaccel_comm_unique_id", "symaccel_comm_init",
    "symaccel_comm_destroy", "symaccel_multi_set_transport", "symaccel_mp3_decode_pipelined",
    "symaccel_mp3_decode_pp_device", "symaccel_mp3_decode_device",
    "symaccel_aac_joint_stereo_list_device", "symaccel_aac_decode_pipelined", "symaccel_vorbis_decode",
    "symaccel_batcher_create", "symaccel_batcher_destroy", "symaccel_batcher_reserve", "symaccel_batcher_commit", "symaccel_batcher_wait",
    "symaccel_batcher_release", "symaccel_batcher_submit", "symaccel_batcher_collect", "symaccel_batcher_abandon",
    "symaccel_batcher_submit_aac_synth", "symaccel_batcher_submit_mp3_synth", "symaccel_batcher_submit_mp3_decode", "symaccel_batcher_submit_vorbis_synth", "symaccel_batcher_aac_bands", "symaccel_batcher_submit_aac_decode", "symaccel_batcher_flush", "symaccel_batcher_hint", "symaccel_batcher_plane_bytes",
    "symaccel_batcher_get_stats", "symaccel_batcher_configure", "symaccel_batcher_last_error", "symaccel_batcher_vorbis_floor",
    "symaccel_batcher_submit_vorbis_decode", "symaccel_batcher_submit_flac_restore", "symaccel_batcher_submit_alac_predict",
    "symaccel_flac_restore_strided_device", "symaccel_alac_predict_strided_device", "symaccel_row_stride",
]

_vp, _sz, _i, _d, _u32 = C.c_void_p, C.c_size_t, C.c_int, C.c_double, C.c_uint32


class VorbisFloor1Job(C.Structure):  # symaccel_vorbis_floor1_job
    _fields_ = [("x_list", C.c_void_p), ("n_posts", C.c_int), ("multiplier", C.c_int), ("d_y", C.c_void_p), ("n", C.c_uint32),
                ("d_line_offsets", C.c_void_p), ("count", C.c_size_t)]


class SymaccelError(RuntimeError):
    def __init__(self, status, message):
        super().__init__("%s (status %d)" % (message, status))
        self.status = status


def _links_hip(path):
    """does this shared object name the HIP runtime among its dependencies? (a scan of the file for the DT_NEEDED string)"""
    try:
        return b"libamdhip64" in Path(path).read_bytes()
    except OSError:
        return False


class Library:
    def __init__(self, path=None):
        path = Path(path) if path is not None else DEFAULT_SO
        if not path.exists():
            raise FileNotFoundError(
                "%s is missing: build it with `python -m symphonia_amd.build` (hipcc, gfx950). "
                "symphonia_amd has no CPU fallback." % path)
        self.path = path
        # One HIP runtime per process: libsymaccel.so is linked against the system's libamdhip64, PyTorch-ROCm ships its own.  When
        # torch is loaded first the library binds to torch's copy and the two share the device; the other way round both runtimes
        # are live and symaccel_ctx_create fails with a device error (seen on the MI355X box: build() followed by smoke() in one
        # process).  The harness (tests, bench.py, smoke) uses torch for device memory, so a library that links the HIP runtime
        # pulls torch in first when torch is installed but not yet imported -- and SAYS so; a host that never loads torch sets
        # SYMACCEL_NO_TORCH_PRELOAD=1.  (The CPU-emulation library of the tests does not link HIP: nothing to preload.)
        self.torch_preload = None
        if "torch" not in sys.modules and not os.environ.get("SYMACCEL_NO_TORCH_PRELOAD") and _links_hip(path):
            import warnings
            try:
                import torch  # noqa: F401
                self.torch_preload = "imported torch before %s (one HIP runtime per process)" % path.name
            except Exception as e:  # noqa: BLE001
                self.torch_preload = "torch could not be imported before %s: %s: %s" % (path.name, type(e).__name__, e)
                warnings.warn("symphonia_amd: " + self.torch_preload)
            if os.environ.get("SYMACCEL_VERBOSE"):
                print("symphonia_amd: " + self.torch_preload, file=sys.stderr)
        self.dll = C.CDLL(str(path))
        d = self.dll
        d.symaccel_strerror.restype = C.c_char_p
        d.symaccel_strerror.argtypes = [_i]
        d.symaccel_last_error.restype = C.c_char_p
        d.symaccel_last_error.argtypes = [_vp]
        d.symaccel_ctx_create.argtypes = [_i, C.POINTER(_vp)]
        d.symaccel_ctx_destroy.argtypes = [_vp]
        d.symaccel_ctx_destroy.restype = None
        d.symaccel_ctx_set_stream.argtypes = [_vp, _vp]
        d.symaccel_sync.argtypes = [_vp]
        d.symaccel_ctx_set_segment.argtypes = [_vp, _i]
        d.symaccel_fft_c32_device.argtypes = [_vp, _i, _vp, _vp, _sz]
        d.symaccel_fft_c32.argtypes = [_vp, _i, _vp, _vp, _sz]
        d.symaccel_ifft_c32_device.argtypes = [_vp, _i, _vp, _vp, _sz]
        d.symaccel_ifft_c32.argtypes = [_vp, _i, _vp, _vp, _sz]
        d.symaccel_flac_decorrelate.argtypes = [_vp, _vp, _vp, _vp, _sz, _sz, _u32]
        d.symaccel_imdct_f32_device.argtypes = [_vp, _i, _d, _vp, _vp, _sz]
        d.symaccel_imdct_f32.argtypes = [_vp, _i, _d, _vp, _vp, _sz]
        d.symaccel_aac_synth_device.argtypes = [_vp, _vp, _vp, _vp, _vp, _sz, _sz]
        d.symaccel_aac_synth.argtypes = [_vp, _vp, _vp, _vp, _vp, _sz, _sz]
        d.symaccel_mp3_synth_device.argtypes = [_vp, _vp, _vp, _i, _vp, _vp, _vp, _vp, _sz, _sz]
        d.symaccel_mp3_synth.argtypes = [_vp, _vp, _vp, _i, _vp, _vp, _vp, _vp, _sz, _sz]
        d.symaccel_mpa_polyphase_device.argtypes = [_vp, _i, _vp, _vp, _vp, _vp, _sz, _sz]
        d.symaccel_mpa_polyphase.argtypes = [_vp, _i, _vp, _vp, _vp, _vp, _sz, _sz]
        d.symaccel_aac_joint_stereo_device.argtypes = [_vp, _vp, _sz, _vp, _vp, _sz, _vp, _i, _vp, _i]
        d.symaccel_aac_tns_device.argtypes = [_vp, _vp, _sz, _vp, _sz]
        d.symaccel_aac_joint_stereo_list_device.argtypes = [_vp, _vp, _sz, _vp, _vp, _sz, _vp, _i, _vp, _i, _vp, _sz]
        d.symaccel_batcher_create.argtypes = [_vp, _sz, C.POINTER(_vp)]
        d.symaccel_batcher_destroy.argtypes = [_vp]
        d.symaccel_batcher_reserve.argtypes = [_vp, _i, _i, _sz, _sz, _vp, C.POINTER(C.c_uint64)]
        d.symaccel_batcher_commit.argtypes = [_vp, C.c_uint64]
        d.symaccel_batcher_wait.argtypes = [_vp, C.c_uint64, _vp]
        d.symaccel_batcher_release.argtypes = [_vp, C.c_uint64]
        d.symaccel_batcher_submit.argtypes = [_vp, _i, _i, _sz, _sz, _vp, _vp, _vp, C.POINTER(C.c_uint64)]
        d.symaccel_batcher_collect.argtypes = [_vp, C.c_uint64]
        d.symaccel_batcher_abandon.argtypes = [_vp, C.c_uint64]
        d.symaccel_batcher_submit_aac_synth.argtypes = [_vp, _vp, _vp, _vp, _vp, _sz, _sz, C.POINTER(C.c_uint64)]
        d.symaccel_batcher_submit_mp3_synth.argtypes = [_vp, _vp, _vp, _i, _vp, _vp, _vp, _vp, _sz, _sz, C.POINTER(C.c_uint64)]
        d.symaccel_batcher_submit_mp3_decode.argtypes = [_vp, _vp, _vp, _vp, _vp, _i, _vp, _vp, _vp, _vp, _sz, _sz, C.POINTER(C.c_uint64)]
        d.symaccel_batcher_flush.argtypes = [_vp]
        d.symaccel_batcher_hint.argtypes = [_vp]
        d.symaccel_batcher_plane_bytes.argtypes = [_i, _i, _sz, _vp, _vp, _vp]
        d.symaccel_batcher_aac_bands.argtypes = [_vp, _vp, _i, _vp, _i, C.POINTER(_i)]
        d.symaccel_batcher_submit_aac_decode.argtypes = [_vp, _i, _vp, _vp, _vp, _vp, _sz, _vp, _sz, _vp, _vp, _sz, _sz, C.POINTER(C.c_uint64)]
        d.symaccel_batcher_submit_vorbis_synth.argtypes = [_vp, _i, _i, _vp, _vp, _vp, _vp, _vp, _sz, _sz, C.POINTER(C.c_uint64)]
        d.symaccel_batcher_get_stats.argtypes = [_vp, _vp]
        d.symaccel_batcher_configure.argtypes = [_vp, _i, _sz]
        d.symaccel_batcher_last_error.argtypes = [_vp, _vp, _sz]
        d.symaccel_batcher_vorbis_floor.argtypes = [_vp, _vp, C.POINTER(_i)]
        d.symaccel_batcher_submit_vorbis_decode.argtypes = [_vp, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _sz, _sz, C.POINTER(C.c_uint64)]
        d.symaccel_batcher_submit_flac_restore.argtypes = [_vp, _vp, _vp, _vp, _vp, _u32, _sz, _sz, C.POINTER(C.c_uint64)]
        d.symaccel_batcher_submit_alac_predict.argtypes = [_vp, _vp, _vp, _vp, _vp, _vp, _sz, _sz, C.POINTER(C.c_uint64)]
        d.symaccel_aac_decode_pipelined.argtypes = [_vp, _vp, _vp, _vp, _vp, _sz, _vp, _i, _vp, _i, _vp, _sz, _vp, _vp, _sz, _sz, _sz]
        d.symaccel_mp3_stereo_device.argtypes = [_vp, _vp, _sz, _vp, _vp, _i, _sz]
        d.symaccel_mp3_requantize_stereo_device.argtypes = [_vp, _vp, _vp, _sz, _vp, _vp, _i, _vp, _sz]
        d.symaccel_mp3_requantize_device.argtypes = [_vp, _vp, _vp, _i, _vp, _sz]
        d.symaccel_mp3_requantize.argtypes = [_vp, _vp, _vp, _i, _vp, _sz]
        d.symaccel_vorbis_synth_device.argtypes = [_vp, _i, _i, _vp, _sz, _vp, _vp, _vp, _vp, _sz, _sz, _sz]
        d.symaccel_vorbis_synth_fr_device.argtypes = [_vp, _i, _i, _vp, _vp, _sz, _vp, _vp, _vp, _vp, _sz, _sz, _sz]
        d.symaccel_vorbis_synth.argtypes = [_vp, _i, _i, _vp, _sz, _vp, _vp, _vp, _vp, _sz, _sz, _sz]
        d.symaccel_vorbis_decode.argtypes = [_vp, _i, _i, _vp, _sz, _vp, _vp, _vp, _sz, _vp, _sz, _sz, _vp, _vp, _vp, _vp, _vp, _sz, _sz, _sz]
        d.symaccel_vorbis_inverse_coupling_device.argtypes = [_vp, _vp, _sz, _vp, _vp, _sz]
        d.symaccel_vorbis_dot_product_device.argtypes = [_vp, _vp, _vp, _sz]
        d.symaccel_vorbis_deinterleave2_device.argtypes = [_vp, _vp, _vp, _i, _sz, _sz]
        d.symaccel_vorbis_floor1_device.argtypes = [_vp, _vp, _i, _i, _vp, _u32, _vp, _sz]
        d.symaccel_vorbis_floor1_dot_device.argtypes = [_vp, _vp, _i, _i, _vp, C.c_uint32, _vp, _vp, _sz]
        d.symaccel_vorbis_floor1_y_device.argtypes = [_vp, _vp, _i, _i, _vp, C.c_uint32, _vp, _vp, _sz]
        d.symaccel_vorbis_floor1_y_jobs_device.argtypes = [_vp, _vp, _sz, _vp]
        d.symaccel_vorbis_floor1_dot_at_device.argtypes = [_vp, _vp, _i, _i, _vp, C.c_uint32, _vp, _vp, _vp, _sz]
        d.symaccel_vorbis_synth_fy_pp_device.argtypes = [_vp, _i, _i, _vp, _vp, _sz, _vp, _vp, _vp, _vp, _vp, _vp, _sz, _sz, _sz]
        d.symaccel_vorbis_synth_fy_device.argtypes = [_vp, _i, _i, _vp, _vp, _sz, _vp, _vp, _vp, _vp, _sz, _sz, _sz]
        d.symaccel_flac_restore_device.argtypes = [_vp, _vp, _vp, _vp, _sz, _sz]
        d.symaccel_flac_restore_stereo_device.argtypes = [_vp, _vp, _vp, _vp, _vp, _u32, _sz, _sz]
        d.symaccel_flac_restore_strided_device.argtypes = [_vp, _vp, _vp, _vp, _vp, _u32, _sz, _sz, _sz]
        d.symaccel_alac_predict_strided_device.argtypes = [_vp, _vp, _vp, _vp, _vp, _vp, _sz, _sz, _sz]
        d.symaccel_row_stride.argtypes = [_sz]
        d.symaccel_row_stride.restype = _sz
        d.symaccel_flac_restore.argtypes = [_vp, _vp, _vp, _vp, _sz, _sz]
        d.symaccel_flac_decorrelate_device.argtypes = [_vp, _vp, _vp, _vp, _sz, _sz, _u32]
        d.symaccel_alac_predict_device.argtypes = [_vp, _vp, _vp, _vp, _sz, _sz]
        d.symaccel_alac_predict_stereo_device.argtypes = [_vp, _vp, _vp, _vp, _vp, _vp, _sz, _sz]
        d.symaccel_alac_predict.argtypes = [_vp, _vp, _vp, _vp, _sz, _sz]
        d.symaccel_mp3_decode_pipelined.argtypes = [_vp, _vp, _vp, _vp, _vp, _sz, _vp, _i, _vp, _vp, _vp, _vp, _sz, _sz, _sz]
        d.symaccel_mp3_decode_pp_device.argtypes = [_vp, _vp, _vp, _vp, _vp, _sz, _vp, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _sz, _sz]
        d.symaccel_mp3_decode_device.argtypes = [_vp, _vp, _vp, _vp, _vp, _sz, _vp, _i, _vp, _vp, _vp, _vp, _sz, _sz]
        d.symaccel_shard_range.argtypes = [_sz, _i, _i, C.POINTER(_sz), C.POINTER(_sz)]
        d.symaccel_scatter_streams.argtypes = [_vp, _vp, _i, _i, _i, _vp, _vp, _sz, _sz]
        d.symaccel_gather_streams.argtypes = [_vp, _vp, _i, _i, _i, _vp, _vp, _sz, _sz]
        d.symaccel_exchange_pipelined.argtypes = [_vp, _vp, _i, _i, _i, _vp, _vp, _sz, _vp, _vp, _sz, _sz, _i, _vp, _vp]
        d.symaccel_comm_unique_id.argtypes = [_vp]
        d.symaccel_comm_init.argtypes = [_vp, _vp, _i, _i, C.POINTER(_vp)]
        d.symaccel_comm_destroy.argtypes = [_vp]
        d.symaccel_multi_set_transport.argtypes = [_vp]
        d.symaccel_probe_copy_device.argtypes = [_vp, _vp, _vp, _sz, _u32, _u32]
        d.symaccel_alac_mid_side_device.argtypes = [_vp, _vp, _vp, _vp, _vp, _sz, _sz]
        d.symaccel_alac_mid_side.argtypes = [_vp, _vp, _vp, _vp, _vp, _sz, _sz]
        d.symaccel_aac_synth_pp_device.argtypes = [_vp, _vp, _vp, _vp, _vp, _vp, _sz, _sz]
        d.symaccel_aac_synth_js_pp_device.argtypes = [_vp, _vp, _vp, _vp, _vp, _sz, _vp, _i, _vp, _i, _vp, _vp, _vp, _sz, _sz]
        d.symaccel_aac_synth_js_device.argtypes = [_vp, _vp, _vp, _vp, _vp, _sz, _vp, _i, _vp, _i, _vp, _vp, _sz, _sz]
        d.symaccel_mp3_synth_pp_device.argtypes = [_vp, _vp, _vp, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _sz, _sz]
        d.symaccel_vorbis_synth_pp_device.argtypes = [_vp, _i, _i, _vp, _vp, _sz, _vp, _vp, _vp, _vp, _vp, _vp, _sz, _sz, _sz]
        d.symaccel_mpa_polyphase_pp_device.argtypes = [_vp, _i, _vp, _vp, _vp, _vp, _vp, _vp, _sz, _sz]
        d.symaccel_host_alloc.argtypes = [_sz, C.POINTER(_vp)]
        d.symaccel_host_free.argtypes = [_vp]
        d.symaccel_host_register.argtypes = [_vp, _sz]
        d.symaccel_host_unregister.argtypes = [_vp]
        d.symaccel_aac_synth_pipelined.argtypes = [_vp, _vp, _vp, _vp, _vp, _sz, _sz, _sz]
        d.symaccel_mp3_synth_pipelined.argtypes = [_vp, _vp, _vp, _i, _vp, _vp, _vp, _vp, _sz, _sz, _sz]
        d.symaccel_flac_restore_pipelined.argtypes = [_vp, _vp, _vp, _vp, _sz, _sz, _sz]
        d.symaccel_host_aac_pulse.argtypes = [_vp, _sz, _vp, _sz, _vp, _i]
        d.symaccel_host_vorbis_bark_map.argtypes = [_u32, C.c_uint16, C.c_uint16, _vp]
        d.symaccel_host_vorbis_floor0_coeffs.argtypes = [_vp, _i]
        d.symaccel_host_vorbis_floor0.argtypes = [_vp, _i, _vp, _u32, C.c_uint16, C.c_uint8, C.c_uint8, C.c_uint64, _vp]
        d.symaccel_flac_block_status_device.argtypes = [_vp, _vp, _sz, _sz, _vp]
        d.symaccel_alac_block_status_device.argtypes = [_vp, _vp, _sz, _vp]
        d.symaccel_vorbis_floor1_status_device.argtypes = [_vp, _i, _vp, _sz, _vp]
        d.symaccel_aac_tns_status_device.argtypes = [_vp, _sz, _vp, _sz, _vp]
        d.symaccel_table_f32.argtypes = [_vp, _i, _vp, _sz]
        d.symaccel_imdct_twiddles.argtypes = [_i, _d, _vp]
        d.symaccel_fft_twiddles.argtypes = [_i, _vp]

    def build_flags(self):
        """The flags stamp symphonia_amd.build wrote beside this library (None if it has none)."""
        try:
            return json.loads(Path(str(self.path) + ".flags.json").read_text())
        except (OSError, ValueError):
            return None

    def check(self, status, ctx=None):
        if status < 0:
            msg = self.dll.symaccel_strerror(status).decode()
            if ctx:
                extra = self.dll.symaccel_last_error(ctx).decode()
                if extra:
                    msg += ": " + extra
            raise SymaccelError(status, msg)
        return status

    # table read-back (host copies; no device needed)
    def table(self, which):
        buf = np.empty(8207, dtype=np.float32)  # the largest table (TABLE_MP3_POW43)
        n = self.check(self.dll.symaccel_table_f32(None, which, buf.ctypes.data, buf.size))
        return buf[:n].copy()

    def imdct_twiddles(self, n, scale):
        buf = np.empty(n // 2, dtype=np.complex64)
        self.check(self.dll.symaccel_imdct_twiddles(n, float(scale), buf.ctypes.data))
        return buf

    def fft_twiddles(self, n):
        buf = np.empty(n // 2, dtype=np.complex64)
        self.check(self.dll.symaccel_fft_twiddles(n, buf.ctypes.data))
        return buf


_default = None


def default_library():
    global _default
    if _default is None:
        # SYMACCEL_LIB: bind another build of the same ABI (a tuned side build, symphonia_amd/build.py); never a fallback
        _default = Library(os.environ.get("SYMACCEL_LIB") or None)
    return _default
