"""codecs::LookaheadDecoder (include/symaccel.hpp): the reference's AudioDecoder method set (codecs/audio.rs:251-298) over
batched calls.  tests/cpp/lookahead_test.cpp decodes synthetic AAC, MP3 and Vorbis tracks packet by packet and compares every
returned buffer with a frame-by-frame decoder (the oracle), across batch boundaries, a reset() and a discontinuity.
CPU: linked against the emulation build of the kernels; GPU: against libsymaccel.so."""
import subprocess
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
BUILD = ROOT / "tests" / "cpp" / "build"
sys.path.insert(0, str(ROOT / "tests" / "emu"))


def build(against_emu):
    import oracle
    oracle.build()
    if against_emu:
        import build_emu
        so = build_emu.build()
        libname = "symaccel_emu"
    else:
        from symphonia_amd import build as sa_build
        so = sa_build.build()
        libname = "symaccel"
    BUILD.mkdir(exist_ok=True)
    exe = BUILD / ("lookahead_test_" + ("emu" if against_emu else "gpu"))
    cmd = ["g++", "-std=c++17", "-O1", "-Wall", "-I", str(ROOT / "include"), "-I", str(ROOT / "oracle"),
           str(ROOT / "tests" / "cpp" / "lookahead_test.cpp"), "-o", str(exe), "-L", str(so.parent), "-l" + libname,
           "-L", str(ROOT / "oracle"), "-lsymoracle", "-Wl,-rpath," + str(so.parent), "-Wl,-rpath," + str(ROOT / "oracle"), "-lm", "-pthread"]
    subprocess.run(cmd, check=True)
    return exe


def test_lookahead_decoder_logic_on_the_emulation_build():
    exe = build(against_emu=True)
    r = subprocess.run([str(exe)], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and "all checks passed" in r.stdout, r.stdout[-3000:] + r.stderr[-2000:]


@pytest.mark.gpu
def test_lookahead_decoder_on_the_gpu():
    exe = build(against_emu=False)
    r = subprocess.run([str(exe)], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "all checks passed" in r.stdout, r.stdout[-3000:] + r.stderr[-2000:]


def test_decoders_bench_harness_on_the_emulation_build():
    """tools/decoders_bench.cpp (what `bench.py --workload decoders` runs on the GPU): S streams through LookaheadDecoder + one
    Batcher on several caller threads, and the same decoders batching per stream -- control flow and bookkeeping on the emulation
    library (its numbers mean nothing): no failed decode, every stream's batches accounted for, the batcher coalesced."""
    import json
    import build_emu
    so = build_emu.build()
    BUILD.mkdir(exist_ok=True)
    exe = BUILD / "decoders_bench_emu"
    subprocess.run(["g++", "-std=c++17", "-O2", "-Wall", "-I", str(ROOT / "include"), str(ROOT / "tools" / "decoders_bench.cpp"), "-o", str(exe),
                    "-L", str(so.parent), "-lsymaccel_emu", "-Wl,-rpath," + str(so.parent), "-pthread"], check=True)
    for codec, extra in (("aac", []), ("aacd", []), ("mp3h", []), ("vorbis", []), ("mp3", ["--per-stream"]), ("aac", ["--direct"]), ("mp3h", ["--direct", "--in-phase"]),
                         ("mp3", ["--direct"]), ("flac", [])):
        r = subprocess.run([str(exe), "--codec", codec, "--streams", "6", "--lookahead", "4", "--packets", "12", "--threads", "3"] + extra,
                           capture_output=True, text=True, timeout=900)
        assert r.returncode == 0, r.stderr[-2000:]
        d = json.loads(r.stdout.strip().splitlines()[-1])
        assert d["failures"] == 0 and d["packets"] == 72 and d["streams"] == 6 and d["threads"] == 3
        if "--per-stream" in extra:
            assert d["mode"] == "per-stream" and d["launches"] == 0
        else:
            assert d["mode"] == "batcher" and 0 < d["launches"] < d["decoder_batches"]
