"""The floor-1 render kernel replaces render_line's integer DDA (vorbis floor.rs:785-825) by a closed form in f32; this
walks every (adx, |dy|, t) the ABI admits and checks that the two agree (tests/cpp/floor1_division_check.c)."""
import os
import subprocess
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
BUILD = ROOT / "tests" / "cpp" / "build"


def test_floor1_closed_form_is_exact_for_every_segment():
    BUILD.mkdir(exist_ok=True)
    exe = BUILD / "floor1_division_check"
    subprocess.run(["gcc", "-O2", "-fopenmp", "-ffp-contract=off", "-o", str(exe), str(ROOT / "tests" / "cpp" / "floor1_division_check.c")],
                   check=True)
    # segments longer than the block (4096 < adx <= 65535, first 4096 lines): every 16th adx here, every adx with
    # SYM_SLOW_TESTS=1 (run once per change of the closed form: profiles/r03_floor1_wide_check.txt)
    stride = "1" if os.environ.get("SYM_SLOW_TESTS") == "1" else "16"
    out = subprocess.run([str(exe), "4096", stride], capture_output=True, text=True, timeout=3600)
    assert out.returncode == 0, out.stdout + out.stderr
    assert out.stdout.count("bad 0") == 3, out.stdout
