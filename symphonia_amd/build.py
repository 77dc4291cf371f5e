"""Build libsymaccel.so (the C-ABI product library) with hipcc for gfx950.

    python -m symphonia_amd.build            # or __graft_entry__.build()

hipcc cross-compiles without a GPU.  Flags that matter for parity (DESIGN.md "Arithmetic contract"):
  -ffp-contract=off   no FMA contraction anywhere (Rust never contracts)
  (no fast-math, no -fgpu-flush-denormals-to-zero: f32 denormals stay enabled)

Tuning knobs (development only).  SYMACCEL_TUNE_<NAME>=<int> in the environment becomes -DSYM_<NAME>=<int> for the
names in TUNING_KNOBS, nothing else in the environment is looked at.  A tuned build never replaces the product library:
it is written to symphonia_amd/build/tuned/libsymaccel.so (bind it with SYMACCEL_LIB=<path>; bench.py reports the
flags of whatever library it loaded).  The flags of every build are stamped beside the .so (<so>.flags.json) and are
part of needs_build().
"""
import hashlib
import json
import os
import shutil
import subprocess
import sys
from pathlib import Path

HERE = Path(__file__).resolve().parent
CSRC = HERE / "csrc"
OUT = HERE / "libsymaccel.so"
TUNED_OUT = HERE / "build" / "tuned" / "libsymaccel.so"
SOURCES = ["tables.cpp", "ctx.cpp", "host_tools.cpp", "stage.cpp", "batcher.cpp", "multi.cpp", "imdct_generic.hip", "imdct_big.hip", "aac.hip", "aac_tools.hip", "mp3.hip", "mpa_polyphase.hip", "mp3_requant.hip",
           "mp3_stereo.hip", "vorbis.hip", "vorbis_wave.hip", "vorbis_wave2.hip", "vorbis_wg.hip", "flac.hip", "alac.hip", "state_copy.hip", "batch_copy.hip", "probe.hip"]
ARCH = "gfx950"
FLAGS = ["-O3", "-std=c++17", "-fPIC", "-I", str(CSRC), "-ffp-contract=off", "-fno-fast-math", "-fno-slp-vectorize", "-Wall", "-Wno-unused-function",
         "-Wno-unused-variable", "-Wno-unused-but-set-variable", "-Wno-missing-braces"]
# Per-source additions.  aac_tools.hip: the TNS filters are chains of dependent packed instructions, and a packed instruction that reads the
# result of the instruction right before it costs a wait state (s_nop): the default scheduler puts every multiply directly in front of its
# subtraction (mul, nop, sub per tap), the ILP strategy interleaves the next tap's multiply (1239 -> 150 s_nop in aac_tns_pair_kernel<12, true>).
# mp3.hip / alac.hip / flac.hip: measured on the GPU against the default strategy (profiles/r06z13_ilp_ab.txt): MP3 int16 -> PCM 0.428 -> 0.403 ms,
# ALAC 3.04 -> 3.00, FLAC 7.75 -> 7.71, MP3 config 3 equal; vorbis_wave.hip spills under it (68 B of scratch) and vorbis.hip gains nothing: not these.
_ILP = ["-mllvm", "-amdgpu-sched-strategy=iterative-ilp"]
SOURCE_FLAGS = {"aac_tools.hip": _ILP, "mp3.hip": _ILP, "alac.hip": _ILP, "flac.hip": _ILP}  # (an AMDGPU option: the host pass ignores it; -misched=gcn-iterative-ilp crashes the x86 side)
TUNING_KNOBS = ("AAC_MIN_WAVES", "AAC_PREFETCH", "AAC_VARIANT", "NT", "MP3_WAVES", "MP3_VARIANT", "MP3_WG_WAVES", "MP3_FUSED_WAVES", "MP3_FUSED_WG_WAVES", "VORBIS_WAVES", "ALAC_SMALL_WAVES", "FLAC_PARTS", "MP3_SLOT_GROUP", "MP3_PACKED", "MP3_PAIR_GROUP", "AAC_ABLATE", "AAC_SINK", "AAC_CLOCK", "AAC_QUAD", "MULTI_WAVE", "VORBIS_WAVE2", "VORBIS_WG", "VORBIS_WG_SHARED", "ST_POLICY", "PACKED_C32", "WG4096_ABLATE", "FLAC_STORE_SWITCH", "ALAC_ONE_LAUNCH", "MP3_FRONT", "LDS_ABLATE", "F1_ABLATE", "F1_WAVES", "TNS_AHEAD", "TNS_ABLATE", "ALAC_UPDATE", "FLAC_OLDEST_FIRST", "FLAC_GROUP", "ALAC_UNROLL", "F1_LANE16", "FLAC_WAVES")
TUNE_PREFIX = "SYMACCEL_TUNE_"


def hipcc():
    exe = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not Path(exe).exists():
        raise RuntimeError("hipcc not found: libsymaccel has no CPU build")
    return exe


def tuned_source_flags(env=None):
    """SYMACCEL_TUNE_ILP=<a.hip,b.hip>: the ILP scheduling strategy for more sources than SOURCE_FLAGS names (development A/B: a tuned side build)."""
    env = os.environ if env is None else env
    names = [n for n in env.get(TUNE_PREFIX + "ILP", "").split(",") if n]
    for n in names:
        if n not in SOURCES:
            raise ValueError("%sILP names %r, not one of the sources" % (TUNE_PREFIX, n))
    out = {k: list(v) for k, v in SOURCE_FLAGS.items()}
    for n in names:
        out[n] = ["-mllvm", "-amdgpu-sched-strategy=" + env.get(TUNE_PREFIX + "SCHED", "iterative-ilp")]  # (SYMACCEL_TUNE_SCHED: another strategy for the A/B)
    return out


def tuning_defines(env=None):
    """-DSYM_<NAME>=<int> for every allow-listed SYMACCEL_TUNE_<NAME> in the environment."""
    env = os.environ if env is None else env
    out = []
    if env.get(TUNE_PREFIX + "ILP"):
        out.append("-DSYM_TUNED_ILP_SOURCES=1")  # (names no macro of the sources: it only marks the build as a tuned one, see tuned_source_flags)
    for name in TUNING_KNOBS:
        v = env.get(TUNE_PREFIX + name)
        if v is None:
            continue
        if not v.lstrip("-").isdigit():
            raise ValueError("%s%s must be an integer, got %r" % (TUNE_PREFIX, name, v))
        out.append("-DSYM_%s=%d" % (name, int(v)))
    return out


def source_files():
    """what the product library is built from (csrc/experiments/ is only reachable through a tuning knob: not part of it)"""
    return sorted(p for p in CSRC.glob("*") if p.is_file()) + [HERE.parent / "include" / "symaccel.h"]


def flags_record(defines):
    h = hashlib.sha256()
    for p in source_files():
        h.update(p.name.encode())
        h.update(p.read_bytes())
    return {"arch": ARCH, "flags": [f for f in FLAGS if f != str(CSRC)], "source_flags": tuned_source_flags(), "tuning": list(defines), "sources": SOURCES,
            "source_sha256": h.hexdigest()}


def stamp_path(so):
    return Path(str(so) + ".flags.json")


def read_stamp(so):
    try:
        return json.loads(stamp_path(so).read_text())
    except (OSError, ValueError):
        return None


def needs_build(out=OUT, defines=()):
    if not Path(out).exists():
        return True
    have = read_stamp(out)
    if have is None:
        # a library without a stamp (built before stamps existed, or shipped alone): fall back to modification times
        t = Path(out).stat().st_mtime
        return any(p.stat().st_mtime > t for p in source_files() + [Path(__file__)])
    return have != flags_record(defines)


def build(force=False, verbose=False, save_temps=False):
    """Build (if needed) and return the path of the library: the product library, or the tuned side build when
    SYMACCEL_TUNE_* knobs are set."""
    defines = tuning_defines()
    out = TUNED_OUT if defines else OUT
    if not force and not needs_build(out, defines):
        return out
    objdir = (HERE / "build" / "tuned" / "obj") if defines else (HERE / "build")
    objdir.mkdir(parents=True, exist_ok=True)
    objs = []
    procs = []
    # A tuned build only recompiles the sources a knob can reach (the knob's macro appears in the file, or in a header -- then
    # everything); the rest links the product build's objects, which are current when the product library is.
    reuse = set()
    if defines and not force and not needs_build(OUT, ()):
        macros = [d[2:].split("=")[0] for d in defines]
        if not any(m in h.read_text() for h in CSRC.glob("*.h") for m in macros):
            reuse = {s for s in SOURCES if not any(m in (CSRC / s).read_text() for m in macros)
                     and (HERE / "build" / (s.replace(".", "_") + ".o")).exists()}
    for src in SOURCES:
        if src in reuse:
            objs.append(str(HERE / "build" / (src.replace(".", "_") + ".o")))
            continue
        obj = objdir / (src.replace(".", "_") + ".o")
        cmd = [hipcc(), "--offload-arch=" + ARCH, "-x", "hip", *FLAGS, *tuned_source_flags().get(src, []), *defines, "-c", str(CSRC / src), "-o", str(obj)]
        if save_temps:
            cmd += ["-save-temps=obj"]
        if verbose:
            print(" ".join(cmd))
        procs.append((src, subprocess.Popen(cmd, cwd=str(objdir), stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
        objs.append(str(obj))
    failed = False
    for src, p in procs:
        text = p.communicate()[0].decode(errors="replace")
        if p.returncode != 0:
            failed = True
            sys.stderr.write("==== %s failed ====\n%s\n" % (src, text))
        elif verbose and text.strip():
            print(text)
    if failed:
        raise RuntimeError("hipcc failed")
    tmp = Path(str(out) + ".tmp")  # link beside the target, then rename: a process that has the old file mapped keeps it
    subprocess.run([hipcc(), "--offload-arch=" + ARCH, "-shared", "-fPIC", "-o", str(tmp), *objs], check=True)
    os.replace(tmp, out)
    stamp_path(out).write_text(json.dumps(flags_record(defines), indent=1) + "\n")
    return out


DECODERS_BENCH = HERE / "build" / "decoders_bench"


def build_decoders_bench(force=False):
    """tools/decoders_bench.cpp: S streams through codecs::LookaheadDecoder and the cross-stream batcher, host memory in -> host
    memory out (`bench.py --workload decoders` runs it).  g++ against the built libsymaccel.so; the binary sits beside the objects
    (it travels to the GPU box with them) and finds the library through an $ORIGIN-relative rpath."""
    so = build()
    root = HERE.parent
    srcs = [root / "tools" / "decoders_bench.cpp", root / "include" / "symaccel.hpp", root / "include" / "symaccel.h"]
    if not force and DECODERS_BENCH.exists() and all(DECODERS_BENCH.stat().st_mtime >= p.stat().st_mtime for p in srcs + [so]):
        return DECODERS_BENCH
    DECODERS_BENCH.parent.mkdir(parents=True, exist_ok=True)
    cxx = shutil.which("g++") or "g++"
    subprocess.run([cxx, "-std=c++17", "-O2", "-Wall", "-I", str(root / "include"), str(srcs[0]), "-o", str(DECODERS_BENCH), "-L", str(so.parent),
                    "-lsymaccel", "-Wl,-rpath,$ORIGIN/..", "-Wl,-rpath-link," + "/opt/rocm/lib", "-pthread"], check=True)
    return DECODERS_BENCH


if __name__ == "__main__":
    # the library first, with the flags asked for; the trait-level harness is a g++ program beside it: a box without g++ (or a link
    # failure there) must not fail a build whose library is fine
    print(build(force="--force" in sys.argv, verbose=True, save_temps="--save-temps" in sys.argv))
    try:
        print(build_decoders_bench(force="--force" in sys.argv))
    except (OSError, subprocess.CalledProcessError) as e:
        sys.stderr.write("decoders_bench not built: %s\n" % e)
