"""The floor-1 render kernel replaces render_line's integer DDA (vorbis floor.rs:785-825) by a closed form in f32; this
walks every (adx, |dy|, t) the ABI admits and checks that the two agree (tests/cpp/floor1_division_check.c)."""
import subprocess
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
BUILD = ROOT / "tests" / "cpp" / "build"


def test_floor1_closed_form_is_exact_for_every_segment():
    BUILD.mkdir(exist_ok=True)
    exe = BUILD / "floor1_division_check"
    subprocess.run(["gcc", "-O2", "-ffp-contract=off", "-o", str(exe), str(ROOT / "tests" / "cpp" / "floor1_division_check.c")],
                   check=True)
    out = subprocess.run([str(exe), "4096"], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "bad 0" in out.stdout
