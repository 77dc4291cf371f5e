"""Build libsymaccel.so (the C-ABI product library) with hipcc for gfx950.

    python -m symphonia_amd.build            # or __graft_entry__.build()

hipcc cross-compiles without a GPU.  Flags that matter for parity (DESIGN.md "Arithmetic contract"):
  -ffp-contract=off   no FMA contraction anywhere (Rust never contracts)
  (no fast-math, no -fgpu-flush-denormals-to-zero: f32 denormals stay enabled)
"""
import os
import shutil
import subprocess
import sys
from pathlib import Path

HERE = Path(__file__).resolve().parent
CSRC = HERE / "csrc"
OUT = HERE / "libsymaccel.so"
SOURCES = ["tables.cpp", "ctx.cpp", "imdct_generic.hip", "aac.hip", "aac_tools.hip", "mp3.hip", "mpa_polyphase.hip", "mp3_requant.hip", "mp3_stereo.hip", "vorbis.hip", "vorbis_wave.hip", "flac.hip", "alac.hip", "state_copy.hip"]
ARCH = "gfx950"
FLAGS = ["-O3", "-std=c++17", "-fPIC", "-I", str(CSRC), "-ffp-contract=off", "-fno-fast-math", "-fno-slp-vectorize", "-Wall", "-Wno-unused-function",
         "-Wno-unused-variable", "-Wno-unused-but-set-variable"]


def hipcc():
    exe = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not Path(exe).exists():
        raise RuntimeError("hipcc not found: libsymaccel has no CPU build")
    return exe


def needs_build():
    if not OUT.exists():
        return True
    t = OUT.stat().st_mtime
    deps = list(CSRC.glob("*")) + [HERE.parent / "include" / "symaccel.h", Path(__file__)]
    return any(p.stat().st_mtime > t for p in deps)


def tuning_defines():
    """Build-time tuning knobs (development): SYM_<NAME>=<int> in the environment becomes -DSYM_<NAME>=<int>."""
    return ["-D%s=%d" % (k, int(v)) for k, v in sorted(os.environ.items()) if k.startswith("SYM_") and v.lstrip("-").isdigit()]


def build(force=False, verbose=False, save_temps=False):
    if not force and not needs_build() and not tuning_defines():
        return OUT
    objdir = HERE / "build"
    objdir.mkdir(exist_ok=True)
    objs = []
    procs = []
    for src in SOURCES:
        obj = objdir / (src.replace(".", "_") + ".o")
        cmd = [hipcc(), "--offload-arch=" + ARCH, "-x", "hip", *FLAGS, *tuning_defines(), "-c", str(CSRC / src), "-o", str(obj)]
        if save_temps:
            cmd += ["-save-temps=obj"]
        if verbose:
            print(" ".join(cmd))
        procs.append((src, subprocess.Popen(cmd, cwd=str(objdir), stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
        objs.append(str(obj))
    failed = False
    for src, p in procs:
        out, _ = p.communicate()
        text = out.decode(errors="replace")
        if p.returncode != 0:
            failed = True
            sys.stderr.write("==== %s failed ====\n%s\n" % (src, text))
        elif verbose and text.strip():
            print(text)
    if failed:
        raise RuntimeError("hipcc failed")
    cmd = [hipcc(), "--offload-arch=" + ARCH, "-shared", "-fPIC", "-o", str(OUT), *objs]
    subprocess.run(cmd, check=True)
    return OUT


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True, save_temps="--save-temps" in sys.argv))
