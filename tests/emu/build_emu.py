"""TEST-ONLY: compile the product's .hip/.cpp sources with g++ against the stand-in HIP header
(tests/emu/include) into tests/emu/libsymaccel_emu.so, so kernel logic can run on the CPU (emu_rt.cpp).
The product library (symphonia_amd/libsymaccel.so) is never built this way."""
import os
import subprocess
import sys
from pathlib import Path

HERE = Path(__file__).resolve().parent
ROOT = HERE.parent.parent
CSRC = ROOT / "symphonia_amd" / "csrc"
OUT = HERE / "libsymaccel_emu.so"
SOURCES = ["tables.cpp", "ctx.cpp", "host_tools.cpp", "stage.cpp", "batcher.cpp", "multi.cpp", "imdct_generic.hip", "imdct_big.hip", "aac.hip", "aac_tools.hip", "mp3.hip", "mpa_polyphase.hip", "mp3_requant.hip", "mp3_stereo.hip", "vorbis.hip", "vorbis_wave.hip", "vorbis_wave2.hip", "vorbis_wg.hip", "flac.hip", "alac.hip", "state_copy.hip", "batch_copy.hip", "probe.hip"]
# same parity-critical flags as the GPU build: no contraction, no fast-math
FLAGS = ["-O1", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-fast-math", "-pthread", "-w"]


def needs_build(out=None):
    out = OUT if out is None else out
    if not out.exists():
        return True
    t = out.stat().st_mtime
    deps = list(CSRC.glob("*")) + list(HERE.glob("*.cpp")) + list((HERE / "include" / "hip").glob("*")) + [
        ROOT / "include" / "symaccel.h", Path(__file__)]
    return any(p.stat().st_mtime > t for p in deps)


def tuned():
    """The allow-listed SYMACCEL_TUNE_* knobs of the product build (symphonia_amd/build.py), so a kernel variant can be
    parity-tested on the CPU before it is taken to the GPU.  A tuned emulation build has its own output path."""
    sys.path.insert(0, str(ROOT))
    from symphonia_amd.build import tuning_defines
    return list(tuning_defines())


def build(force=False):
    """Build if needed; safe to call from several processes at once (pytest -n: the workers share the object directory)."""
    import fcntl
    with open(HERE / ".build.lock", "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        return _build(force)


def _build(force=False):
    defines = tuned()
    if defines:
        tag = "_".join(d[2:].replace("=", "") for d in defines)
        out = HERE / ("libsymaccel_emu_%s.so" % tag)
        objdir = HERE / "build" / tag
        if not force and out.exists() and not needs_build(out):
            return out
    else:
        out, objdir = OUT, HERE / "build"
        if not force and not needs_build():
            return OUT
    objdir.mkdir(exist_ok=True, parents=True)
    procs, objs = [], []
    for src in SOURCES + ["../../tests/emu/emu_rt.cpp"]:
        path = (CSRC / src).resolve()
        obj = objdir / (path.name.replace(".", "_") + ".o")
        cmd = ["g++", "-x", "c++", *FLAGS, *defines, "-I", str(HERE / "include"), "-I", str(CSRC), "-c", str(path), "-o", str(obj)]
        procs.append((src, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
        objs.append(str(obj))
    bad = False
    for src, p in procs:
        log, _ = p.communicate()
        if p.returncode != 0:
            bad = True
            sys.stderr.write("==== %s ====\n%s\n" % (src, log.decode(errors="replace")[-6000:]))
    if bad:
        raise RuntimeError("emulation build failed")
    tmp = out.with_suffix(".so.%d.tmp" % os.getpid())  # linked aside and renamed: a process that races this one never maps half a file
    subprocess.run(["g++", "-shared", "-pthread", "-o", str(tmp), *objs], check=True)
    os.replace(tmp, out)
    return out


HOST_SOURCES = ["tables.cpp", "ctx.cpp", "host_tools.cpp", "stage.cpp", "batcher.cpp", "multi.cpp"]
SANITIZERS = {"asan": ["-fsanitize=address,undefined", "-fno-sanitize-recover=undefined", "-fno-omit-frame-pointer"],
              "tsan": ["-fsanitize=thread", "-fno-omit-frame-pointer"]}


def build_sanitized(kind):
    """The emulation library with the HOST C++ of the product (ctx / stage / batcher / multi / host_tools / tables) compiled under a
    sanitizer (`asan`: AddressSanitizer + UndefinedBehaviorSanitizer; `tsan`: ThreadSanitizer).  The kernels and the emulation runtime
    switch stacks by hand (emu_rt.cpp: work-items are fibers) and stay uninstrumented: what is checked is the code that holds locks,
    owns memory and talks to the runtime.  -> tests/emu/build/<kind>/libsymaccel_emu_<kind>.so"""
    import fcntl
    flags = SANITIZERS[kind]
    with open(HERE / ".build.lock", "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        objdir = HERE / "build" / kind
        out = objdir / ("libsymaccel_emu_%s.so" % kind)
        if out.exists() and not needs_build(out):
            return out
        objdir.mkdir(exist_ok=True, parents=True)
        procs, objs = [], []
        for src in SOURCES + ["../../tests/emu/emu_rt.cpp"]:
            path = (CSRC / src).resolve()
            obj = objdir / (path.name.replace(".", "_") + ".o")
            extra = flags + ["-g"] if src in HOST_SOURCES else []
            cmd = ["g++", "-x", "c++", *FLAGS, *extra, "-I", str(HERE / "include"), "-I", str(CSRC), "-c", str(path), "-o", str(obj)]
            procs.append((src, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
            objs.append(str(obj))
        bad = False
        for src, p in procs:
            log, _ = p.communicate()
            if p.returncode != 0:
                bad = True
                sys.stderr.write("==== %s ====\n%s\n" % (src, log.decode(errors="replace")[-6000:]))
        if bad:
            raise RuntimeError("sanitized emulation build failed")
        tmp = out.with_suffix(".so.%d.tmp" % os.getpid())
        subprocess.run(["g++", "-shared", "-pthread", *flags, "-o", str(tmp), *objs], check=True)
        os.replace(tmp, out)
        return out


if __name__ == "__main__":
    print(build_sanitized(sys.argv[1]) if len(sys.argv) > 1 else build(force=True))
