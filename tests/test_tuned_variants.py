"""The build variants kept behind tuning knobs (SYMACCEL_TUNE_*, symphonia_amd/build.py) are bit-exact too: each one is
compiled into its own CPU-emulation library (tests/emu/build_emu.py honours the same knobs) and runs the emulation
parity tests of its kernel in a subprocess.  (The GPU A/B of the same variants is in profiles/r02*_ab.txt.)"""
import os
import subprocess
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent

CASES = [
    ({"SYMACCEL_TUNE_AAC_VARIANT": "1"}, "tests/test_emu_core_aac.py", "aac_all_sequences or imdct_bit_exact"),
    ({"SYMACCEL_TUNE_AAC_QUAD": "0"}, "tests/test_emu_core_aac.py", "aac"),
    ({"SYMACCEL_TUNE_MP3_VARIANT": "2", "SYMACCEL_TUNE_MP3_PACKED": "0"}, "tests/test_emu_codecs.py", "emu_mp3"),
    ({"SYMACCEL_TUNE_MP3_VARIANT": "3", "SYMACCEL_TUNE_MP3_PACKED": "0"}, "tests/test_emu_codecs.py", "emu_mp3"),
    ({"SYMACCEL_TUNE_MP3_VARIANT": "0", "SYMACCEL_TUNE_MP3_PACKED": "0"}, "tests/test_emu_codecs.py", "emu_mp3"),
    ({"SYMACCEL_TUNE_MP3_VARIANT": "0"}, "tests/test_emu_codecs.py", "emu_mp3"),
    ({"SYMACCEL_TUNE_MP3_PACKED": "0", "SYMACCEL_TUNE_MP3_PAIR_GROUP": "1"}, "tests/test_emu_codecs.py", "emu_mp3"),
    ({"SYMACCEL_TUNE_MP3_SLOT_GROUP": "1"}, "tests/test_emu_codecs.py", "emu_mp3"),
    ({"SYMACCEL_TUNE_MP3_SLOT_GROUP": "18"}, "tests/test_emu_codecs.py", "emu_mp3"),
    ({"SYMACCEL_TUNE_FLAC_PARTS": "4", "SYMACCEL_TUNE_FLAC_STORE_SWITCH": "0"}, "tests/test_emu_codecs.py tests/test_row_stride.py", "emu_flac"),
    # round 6: the knobs of the last session's A/Bs (profiles/r06zz3 .. r06zz17), several per build (one emulation library each)
    ({"SYMACCEL_TUNE_FLAC_GROUP": "4", "SYMACCEL_TUNE_FLAC_OLDEST_FIRST": "1", "SYMACCEL_TUNE_ALAC_UPDATE": "0", "SYMACCEL_TUNE_MP3_FRONT": "3",
      "SYMACCEL_TUNE_F1_LANE16": "0"}, "tests/test_emu_codecs.py tests/test_alac.py tests/test_mp3_stereo.py tests/test_vorbis_floor_y.py",
     "emu_flac or emu_alac or emu_mp3_decode_device or emu_requantize_stereo_fused or emu_floor or emu_synth_floor_y"),
    ({"SYMACCEL_TUNE_ALAC_UPDATE": "2", "SYMACCEL_TUNE_ALAC_UNROLL": "1"}, "tests/test_alac.py", "emu"),
    ({"SYMACCEL_TUNE_VORBIS_WG": "0"}, "tests/test_emu_codecs.py", "register_pass_kernel_pairs"),
    ({"SYMACCEL_TUNE_VORBIS_WG": "2"}, "tests/test_emu_codecs.py", "register_pass_kernel_pairs or emu_vorbis_synth"),
]


@pytest.mark.parametrize("knobs,module,select", CASES, ids=lambda v: "_".join("%s%s" % kv for kv in v.items()) if isinstance(v, dict) else None)
def test_tuned_variant_is_bit_exact_in_emulation(knobs, module, select):
    env = dict(os.environ, **knobs)
    r = subprocess.run([sys.executable, "-m", "pytest", *module.split(), "-x", "-q", "-k", select, "-p", "no:cacheprovider"],
                       cwd=str(ROOT), env=env, capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert " passed" in r.stdout
