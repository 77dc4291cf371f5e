"""The C++ host mirror of the reference's DSP interfaces (include/symaccel.hpp): it compiles against the C ABI with a
plain C++17 compiler, refuses to run without a GPU, and (on the MI355X) passes the reference-style checks of
tests/cpp/host_mirror_test.cpp through the per-packet calls."""
import json
import subprocess
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
BUILD = ROOT / "tests" / "cpp" / "build"


def build_binary():
    from symphonia_amd import build
    import oracle
    so = build.build()
    oracle.build()
    BUILD.mkdir(exist_ok=True)
    kats = json.loads((ROOT / "tests" / "golden" / "ref_kats.json").read_text())
    inc = ["static const float kImdct32Input[32] = {%s};" % ", ".join("(float)%.9g" % v for v in kats["imdct32_input"]),
           "static const float kFft64Input[64][2] = {%s};" % ", ".join("{(float)%.9g, (float)%.9g}" % (a, b) for a, b in kats["fft64_input"])]
    (BUILD / "kats.inc").write_text("\n".join(inc) + "\n")
    exe = BUILD / "host_mirror_test"
    cmd = ["g++", "-std=c++17", "-O1", "-Wall", "-I", str(ROOT / "include"), "-I", str(ROOT / "oracle"), "-I", str(BUILD),
           str(ROOT / "tests" / "cpp" / "host_mirror_test.cpp"), "-o", str(exe),
           "-L", str(so.parent), "-lsymaccel", "-L", str(ROOT / "oracle"), "-lsymoracle",
           "-Wl,-rpath," + str(so.parent), "-Wl,-rpath," + str(ROOT / "oracle"), "-lm"]
    subprocess.run(cmd, check=True)
    return exe


def test_host_mirror_compiles_and_refuses_to_run_without_gpu():
    import torch
    exe = build_binary()
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    r = subprocess.run([str(exe), "--expect-no-device"], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stdout + r.stderr


@pytest.mark.gpu
def test_host_mirror_reference_style_checks():
    exe = build_binary()
    r = subprocess.run([str(exe)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert "all checks passed" in r.stdout
