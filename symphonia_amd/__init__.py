"""symphonia_amd -- MI355X-native batched synthesis backend for Symphonia's DSP hot path.

The compute lives in symphonia_amd/libsymaccel.so (hand-written HIP for gfx950 behind the C ABI
of include/symaccel.h).  This package is the thin host-side mirror of the reference's interfaces.
There is no CPU fallback: importing works anywhere, creating a Context needs the library + a GPU.
"""
from ._ffi import Library, SymaccelError, default_library  # noqa: F401
from .backend import (PinnedBuffer, Batcher, BatchSlot, BATCH_AAC_SYNTH, BATCH_MP3_SYNTH, BATCH_MP3_DECODE, BATCH_VORBIS_SYNTH, BATCH_AAC_DECODE, BATCH_VORBIS_DECODE, BATCH_FLAC_RESTORE, BATCH_ALAC_PREDICT, AAC_PULSE_DTYPE, aac_pulse, vorbis_bark_map, vorbis_floor0_coeffs, vorbis_floor0, flac_block_status, alac_block_status, vorbis_floor1_status, aac_tns_status,  # noqa: F401
                      AacDsp, AacSpectralTools, VORBIS_FLOOR1_DTYPE, AAC_JS_DTYPE, AAC_TNS_DTYPE, AAC_JS_MS, AAC_JS_INTENSITY, AlacPredictor, Context, Fft, FlacPredictor, Ifft, Imdct, Mp3Requantize, Mp3Stereo, Mp3Synthesis, MpaPolyphase, VorbisDsp,  # noqa: F401
                      MP3_REQUANT_DTYPE, MP3_RQ_PREFLAG, MP3_RQ_SCALEFAC_SCALE, MP3_STEREO_DTYPE, MP3_ST_MID_SIDE, MP3_ST_INTENSITY,
                      MP3_ST_MPEG1, MP3_ST_IS_SCALE,
                      aac_side, alac_desc, flac_desc, mp3_side, FLAC_FIXED, FLAC_LPC, FLAC_VERBATIM)

__all__ = ["Library", "SymaccelError", "default_library", "Context", "Imdct", "Fft", "Ifft", "AacDsp", "AacSpectralTools", "Mp3Synthesis", "Mp3Requantize", "Mp3Stereo", "MpaPolyphase",
           "VorbisDsp", "FlacPredictor", "AlacPredictor", "aac_side", "mp3_side", "flac_desc", "alac_desc"]
