"""The host C++ of the product -- csrc/{ctx,stage,batcher,multi,host_tools,tables}.cpp: everything that holds a lock, owns memory or
talks to the runtime -- under AddressSanitizer + UndefinedBehaviorSanitizer and, in a separate build, ThreadSanitizer (SURVEY section 5,
"Race detection / sanitizers"; GPU AddressSanitizer is not available on this pool, so the sanitizers run on the CPU emulation build:
tests/emu/build_emu.py build_sanitized).  Driven three ways: the compiled C++ twin's look-ahead / cross-stream scripts
(tests/cpp/lookahead_test.cpp), the trait-level harness with several caller threads (tools/decoders_bench.cpp), and the Python batcher
suites (concurrent submitters, abandoned tickets, random traffic) with the sanitizer's runtime pre-loaded.  `pytest -m sanitize`;
clean logs of a run are kept under profiles/."""
import os
import subprocess
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
BUILD = ROOT / "tests" / "cpp" / "build"
sys.path.insert(0, str(ROOT / "tests" / "emu"))

pytestmark = pytest.mark.sanitize

REPORTS = ("ERROR: AddressSanitizer", "runtime error:", "WARNING: ThreadSanitizer", "ERROR: LeakSanitizer", "SUMMARY: UndefinedBehaviorSanitizer")
ENV = {"asan": {"ASAN_OPTIONS": "detect_leaks=0:abort_on_error=0:halt_on_error=1", "UBSAN_OPTIONS": "print_stacktrace=1:halt_on_error=1"},
       "tsan": {"TSAN_OPTIONS": "halt_on_error=0:second_deadlock_stack=1:report_signal_unsafe=0"}}
RUNTIME = {"asan": "libasan.so", "tsan": "libtsan.so"}


def sanitized(kind):
    import build_emu
    return build_emu.build_sanitized(kind)


def run(cmd, kind, extra_env=None, timeout=1500):
    env = dict(os.environ, **ENV[kind], **(extra_env or {}))
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=timeout, env=env, cwd=str(ROOT))
    text = r.stdout + r.stderr
    hits = [ln for ln in text.splitlines() if any(tag in ln for tag in REPORTS)]
    return r.returncode, text, hits


def compile_against(so, src, exe, kind, includes, libs=()):
    import build_emu
    BUILD.mkdir(exist_ok=True)
    cmd = ["g++", "-std=c++17", "-O1", "-g", *build_emu.SANITIZERS[kind], *sum((["-I", str(i)] for i in includes), []), str(src), "-o", str(exe),
           "-L", str(so.parent), "-l" + so.stem[3:], "-Wl,-rpath," + str(so.parent), *libs, "-pthread"]
    subprocess.run(cmd, check=True)
    return exe


@pytest.mark.parametrize("kind", ["asan", "tsan"])
def test_cpp_twin_scripts_under_the_sanitizer(kind):
    """codecs::LookaheadDecoder over every codec, with and without the batcher, S streams round-robin, the zero-copy parse"""
    import oracle
    oracle.build()
    so = sanitized(kind)
    exe = compile_against(so, ROOT / "tests" / "cpp" / "lookahead_test.cpp", BUILD / ("lookahead_test_" + kind), kind, [ROOT / "include", ROOT / "oracle"],
                          ["-L", str(ROOT / "oracle"), "-lsymoracle", "-Wl,-rpath," + str(ROOT / "oracle"), "-lm"])
    rc, text, hits = run([str(exe)], kind)
    assert not hits, "\n".join(hits[:10]) + text[-3000:]
    assert rc == 0 and "all checks passed" in text, text[-3000:]


@pytest.mark.parametrize("kind", ["asan", "tsan"])
@pytest.mark.parametrize("args", [["--codec", "aac", "--direct"], ["--codec", "mp3h"], ["--codec", "aacd", "--in-phase"], ["--codec", "vorbis"], ["--codec", "flac"]])
def test_trait_level_harness_with_caller_threads_under_the_sanitizer(kind, args):
    """4 caller threads x 12 streams through ONE batcher (lanes, enqueue outside the mutex, completion flags, slot pool)"""
    import json
    so = sanitized(kind)
    exe = compile_against(so, ROOT / "tools" / "decoders_bench.cpp", BUILD / ("decoders_bench_" + kind), kind, [ROOT / "include"])
    rc, text, hits = run([str(exe), *args, "--streams", "12", "--lookahead", "4", "--packets", "24", "--threads", "4", "--lanes", "3"], kind)
    assert not hits, "\n".join(hits[:10]) + text[-3000:]
    assert rc == 0, text[-3000:]
    d = json.loads([ln for ln in text.splitlines() if ln.startswith("{")][-1])
    assert d["failures"] == 0 and d["launches"] > 0


@pytest.mark.parametrize("kind", ["asan", "tsan"])
def test_python_batcher_suites_under_the_sanitizer(kind):
    """tests/test_batcher.py + tests/test_batcher_kinds.py (ragged groups, flush_bytes, zero-copy slots, abandoned and uncommitted
    tickets, per-ticket failures, concurrent submitters, random traffic) against the sanitized library, runtime pre-loaded"""
    so = sanitized(kind)
    rt = subprocess.run(["gcc", "-print-file-name=" + RUNTIME[kind]], capture_output=True, text=True, check=True).stdout.strip()
    rc, text, hits = run([sys.executable, "-m", "pytest", "tests/test_batcher.py", "tests/test_batcher_kinds.py", "-q", "-x", "-m", "not gpu", "-p", "no:cacheprovider"],
                         kind, {"LD_PRELOAD": rt, "SYMACCEL_EMU_SANITIZED": kind}, timeout=2400)
    assert so.exists()
    assert not hits, "\n".join(hits[:10]) + text[-3000:]
    assert rc == 0 and " passed" in text, text[-3000:]
