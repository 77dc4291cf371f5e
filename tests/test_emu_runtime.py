"""The CPU-emulation runtime itself (tests/emu/emu_rt.cpp: work-items as fibers of the launching thread): barrier semantics, ended
work-items, cross-lane builtins, concurrent launches, 1024-work-item workgroups -- tests/cpp/emu_runtime_test.cpp, built with g++
against the stand-in hip_runtime.h.  Test infrastructure checking test infrastructure: none of this is in the product."""
import subprocess
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
BUILD = ROOT / "tests" / "cpp" / "build"


def test_fiber_runtime_semantics():
    BUILD.mkdir(exist_ok=True)
    exe = BUILD / "emu_runtime_test"
    subprocess.run(["g++", "-x", "c++", "-std=c++17", "-O1", "-w", "-pthread", "-I", str(ROOT / "tests" / "emu" / "include"),
                    str(ROOT / "tests" / "cpp" / "emu_runtime_test.cpp"), str(ROOT / "tests" / "emu" / "emu_rt.cpp"), "-o", str(exe)], check=True)
    r = subprocess.run([str(exe)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "all checks passed" in r.stdout, r.stdout[-3000:] + r.stderr[-2000:]
