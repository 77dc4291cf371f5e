"""TEST-ONLY: compile the product's .hip/.cpp sources with g++ against the stand-in HIP header
(tests/emu/include) into tests/emu/libsymaccel_emu.so, so kernel logic can run on CPU threads.
The product library (symphonia_amd/libsymaccel.so) is never built this way."""
import subprocess
import sys
from pathlib import Path

HERE = Path(__file__).resolve().parent
ROOT = HERE.parent.parent
CSRC = ROOT / "symphonia_amd" / "csrc"
OUT = HERE / "libsymaccel_emu.so"
SOURCES = ["tables.cpp", "ctx.cpp", "host_tools.cpp", "stage.cpp", "imdct_generic.hip", "aac.hip", "aac_tools.hip", "mp3.hip", "mpa_polyphase.hip", "mp3_requant.hip", "mp3_stereo.hip", "vorbis.hip", "vorbis_wave.hip", "flac.hip", "alac.hip", "state_copy.hip"]
# same parity-critical flags as the GPU build: no contraction, no fast-math
FLAGS = ["-O1", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-fast-math", "-pthread", "-w"]


def needs_build():
    if not OUT.exists():
        return True
    t = OUT.stat().st_mtime
    deps = list(CSRC.glob("*")) + list(HERE.glob("*.cpp")) + list((HERE / "include" / "hip").glob("*")) + [
        ROOT / "include" / "symaccel.h", Path(__file__)]
    return any(p.stat().st_mtime > t for p in deps)


def build(force=False):
    if not force and not needs_build():
        return OUT
    objdir = HERE / "build"
    objdir.mkdir(exist_ok=True)
    procs, objs = [], []
    for src in SOURCES + ["../../tests/emu/emu_rt.cpp"]:
        path = (CSRC / src).resolve()
        obj = objdir / (path.name.replace(".", "_") + ".o")
        cmd = ["g++", "-x", "c++", *FLAGS, "-I", str(HERE / "include"), "-I", str(CSRC), "-c", str(path), "-o", str(obj)]
        procs.append((src, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
        objs.append(str(obj))
    bad = False
    for src, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            bad = True
            sys.stderr.write("==== %s ====\n%s\n" % (src, out.decode(errors="replace")[-6000:]))
    if bad:
        raise RuntimeError("emulation build failed")
    subprocess.run(["g++", "-shared", "-pthread", "-o", str(OUT), *objs], check=True)
    return OUT


if __name__ == "__main__":
    print(build(force=True))
