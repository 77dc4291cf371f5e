"""Arithmetic contract at the ISA level: the f32 synthesis kernels must not contain a single fused
multiply-add -- the reference (Rust) never contracts a*b+c, and one FMA changes the rounding of a PCM sample.  The
device assembly of every f32 kernel file is checked for FMA / MAC / MAD mnemonics; flac.hip is allowed its FP64 FMAs
(exact by construction) but no f32 ones either."""
import re
import shutil
import subprocess
from concurrent.futures import ThreadPoolExecutor
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
CSRC = ROOT / "symphonia_amd" / "csrc"
F32_FUSED = re.compile(r"\b(v_fma_f32|v_fmac_f32|v_mac_f32|v_mad_f32|v_pk_fma_f32|v_fma_mix\w*|v_mad_mix\w*|v_fma_legacy_f32|"
                       r"v_mad_legacy_f32|v_mac_legacy_f32|v_dot2c?_f32\w*)\b")
F64_FUSED = re.compile(r"\b(v_fma_f64|v_fmac_f64)\b")
SOURCES = ["aac.hip", "aac_tools.hip", "mp3.hip", "mp3_requant.hip", "mp3_stereo.hip", "mpa_polyphase.hip", "vorbis.hip", "vorbis_wave.hip", "vorbis_wave2.hip", "vorbis_wg.hip",
           "imdct_generic.hip", "imdct_big.hip", "flac.hip", "alac.hip", "state_copy.hip", "batch_copy.hip", "probe.hip"]


import sys
sys.path.insert(0, str(ROOT))
from tools.kernel_resources import device_asm, kernel_resources  # noqa: E402


@pytest.fixture(scope="module")
def asm():
    if not (shutil.which("hipcc") or Path("/opt/rocm/bin/hipcc").exists()):
        pytest.skip("hipcc not available")
    with ThreadPoolExecutor(8) as ex:
        return dict(zip(SOURCES, ex.map(device_asm, SOURCES)))


def kernels(text):
    """{mangled kernel name: body} from a device assembly listing."""
    out = {}
    for m in re.finditer(r"^(_Z\w+):.*?\n(.*?)^\.Lfunc_end\d+:", text, flags=re.S | re.M):
        out[m.group(1)] = m.group(2)
    return out


# Kernels with no f32 signal arithmetic at all: the only FMAs they may contain belong to the compiler's expansion of
# integer and IEEE float DIVISION (index math, the floor-1 DDA reciprocal), which is correctly rounded by construction.
INDEX_MATH_ONLY = re.compile(r"floor1|offsets|deinterleave|flac_|alac_|state_copy|batch_copy|mp3_requantize_kernel")


def test_no_f32_fma_in_any_synthesis_kernel(asm):
    seen = 0
    for src, text in asm.items():
        for name, body in kernels(text).items():
            if INDEX_MATH_ONLY.search(name):
                continue
            seen += 1
            hits = sorted(set(F32_FUSED.findall(body)))
            assert not hits, "%s: %s contains fused f32 arithmetic: %s" % (src, name, hits)
    assert seen >= 8  # aac, mp3, mpa x2, vorbis synth x2 + wave, imdct / fft x4, coupling, dot


def test_every_source_of_the_library_is_checked():
    from symphonia_amd import build
    assert sorted(SOURCES) == sorted(s for s in build.SOURCES if s.endswith(".hip"))


def test_no_kernel_spills(asm):
    """A spilled register is a scratch (HBM-backed) access in the middle of a latency-bound loop: every kernel must
    fit its register budget.  ScratchSize is the compiler's own figure (.amdhsa_private_segment_fixed_size)."""
    seen = 0
    for src, text in asm.items():
        for name, r in kernel_resources(text).items():
            seen += 1
            assert r.get("ScratchSize") == 0, "%s: %s uses %s bytes of scratch (VGPRs %s)" % (src, name, r.get("ScratchSize"), r.get("NumVgprs"))
    assert seen >= 30


def test_aac_three_wave_variant_fits_its_register_budget():
    """SYMACCEL_TUNE_AAC_VARIANT=1 (six wavefronts per workgroup, lane twiddles in LDS, peeled halo) is sized for three
    wavefronts per SIMD: <= 168 VGPRs, no scratch, two 71.6 KiB workgroups per CU -- and contains no fused multiply-add."""
    text = device_asm("aac.hip", ["-DSYM_AAC_VARIANT=1", "-DSYM_AAC_QUAD=0"])  # (the wavefront walk: csrc/experiments/aac_wave_walk.h)
    (r,) = [v for k, v in kernel_resources(text).items() if "quad" not in k and "aac_js_" not in k]  # (aac.hip also holds the product's kernels)
    assert r["ScratchSize"] == 0 and r["NumVgprs"] <= 168 and r["Occupancy"] == 3, r
    assert 2 * r["LDSByteSize"] <= 160 * 1024
    assert not F32_FUSED.search(text)


@pytest.mark.parametrize("variant", [2, 3])
def test_mp3_build_variants_fit_three_waves_per_simd(variant):
    """The measured-and-documented MP3 variants (float4 PCM stores through an LDS tile; two granules in
    flight) stay buildable at the same occupancy, without scratch or fused multiply-adds."""
    # (round-2 variants, as measured then: the default scheduling strategy -- under the product's ILP strategy they spill 28 bytes)
    text = device_asm("mp3.hip", ["-DSYM_MP3_VARIANT=%d" % variant, "-DSYM_MP3_PACKED=0"], source_flags=[])
    (r,) = kernel_resources(text).values()
    assert r["ScratchSize"] == 0 and r["NumVgprs"] <= 168 and r["Occupancy"] == 3, r
    assert 3 * 4 * r["LDSByteSize"] <= 160 * 1024 * (4 if variant == 2 else 1)  # variant 2: four wavefronts per workgroup
    assert not F32_FUSED.search(text)


def test_vorbis_wg_kernel_fits_three_workgroups_per_cu():
    """vorbis_synth_wg_kernel (8192-sample blocks, the 4096 / 8192 pair with BOTH cooperative block routines included): exchange and
    group work areas inside the staging area -- at most 168 VGPRs and a third of the CU's 160 KiB of LDS, no scratch; the round-3
    layout (SYMACCEL_TUNE_VORBIS_WG_SHARED=0) stays buildable."""
    res = kernel_resources(device_asm("vorbis_wg.hip", []))
    assert len(res) == 9  # FUSED 0 / 1 / 2 x short blocks <= 2048 / 4096 / 8192 samples
    for name, r in res.items():
        assert r["ScratchSize"] == 0 and r["NumVgprs"] <= 168 and 3 * r["LDSByteSize"] <= 160 * 1024, (name, r)
    old = kernel_resources(device_asm("vorbis_wg.hip", ["-DSYM_VORBIS_WG_SHARED=0"]))
    assert all(r["ScratchSize"] == 0 and 2 * r["LDSByteSize"] <= 160 * 1024 for r in old.values()), old


def test_alac_has_both_multiply_forms(asm):
    """alac_predict_kernel<., true, .> multiplies with v_mul_i32_i24 / v_mad_i32_i24 (full rate), <., false, .> with
    v_mul_lo_u32; which one a wavefront runs is decided by alac_narrow_kernel from a proven operand bound
    (tests/test_alac.py, narrow_case).  The orders-<=-8 instantiations fit three wavefronts per SIMD."""
    text = asm["alac.hip"]
    bodies = {}
    for chunk in re.split(r"^\s*\.globl\s+", text, flags=re.M)[1:]:
        name = chunk.split(None, 1)[0]
        if "alac_predict_kernel" in name:
            bodies[name] = chunk
    assert len(bodies) == 8
    fast = 0
    for name, body in bodies.items():
        n24, nlo = len(re.findall(r"v_m(?:ul|ad)_i32_i24", body)), len(re.findall(r"v_mul_lo_u32", body))
        # (the few v_mul_lo_u32 of the 24-bit form are address arithmetic and the mid/side weight of the fused store)
        # (and the few 24-bit multiply-adds of the wide form are the coefficient moves of its carried-residual update: signum x direction mask, alac_step QF)
        assert (n24 > 100 and nlo < 60) or (nlo > 100 and n24 < nlo // 3), (name, n24, nlo)
        fast += nlo < 60
    assert fast == 4
    res = kernel_resources(text)
    small = [r for n, r in res.items() if "alac_predict_kernel" in n and n.split("alac_predict_kernel")[1].startswith("ILb") and "ELb1EEE" in n]
    assert len(small) == 4 and all(r["Occupancy"] >= 3 and r["LDSByteSize"] < 10240 for r in small), small


def test_fp64_fma_only_in_the_flac_kernel(asm):
    for src, text in asm.items():
        if src == "flac.hip":
            assert F64_FUSED.search(text), "flac.hip lost its exact FP64 dot product"
        else:
            assert not F64_FUSED.search(text), src


def test_build_flags_pin_the_contract():
    from symphonia_amd import build
    assert "-ffp-contract=off" in build.FLAGS and "-fno-fast-math" in build.FLAGS
    assert not any("flush-denormals" in f or f == "-ffast-math" for f in build.FLAGS)


def test_the_product_translation_units_carry_no_measurement_only_code():
    """VERDICT r3 hygiene: the ablation / cycle-counter / alternative-walk code of the AAC kernel lives in csrc/experiments/ and is
    only compiled into TUNED builds (symphonia_amd/build/tuned/): the default build neither includes the file nor defines its knobs."""
    from symphonia_amd import build
    aac = (build.CSRC / "aac.hip").read_text()
    assert "SYM_AAC_ABLATE" not in aac and "SYM_AAC_CLOCK" not in aac and "SYM_AAC_SINK" not in aac
    assert '#if !SYM_AAC_QUAD\n#include "experiments/aac_wave_walk.h"' in aac
    assert "__global__" in (build.CSRC / "experiments" / "aac_wave_walk.h").read_text()
    assert all(p.parent == build.CSRC or p.name == "symaccel.h" for p in build.source_files())  # the product's sources: csrc/* only
    text = device_asm("aac.hip", [])
    kernels = [k for k in kernel_resources(text)]  # the walk (plain / a channel pair with joint stereo on load) and the pair index, nothing else
    assert len(kernels) == 3 and sum("aac_synth_quad_kernel" in k for k in kernels) == 2 and sum("aac_js_index_kernel" in k for k in kernels) == 1, kernels


def test_rccl_constants_the_c_glue_hard_codes():
    """csrc/multi.cpp binds RCCL through dlsym without its header: the two facts it hard-codes are checked against rccl.h here --
    ncclUniqueId is 128 opaque bytes passed BY VALUE to ncclCommInitRank (symaccel_unique_id), and ncclInt8 == 0 (the dtype of
    every ncclSend / ncclRecv: counts are bytes)."""
    hdr = Path("/opt/rocm/include/rccl/rccl.h")
    if not hdr.exists():
        pytest.skip("no rccl.h in this image")
    text = hdr.read_text()
    assert re.search(r"#define\s+NCCL_UNIQUE_ID_BYTES\s+128\b", text)
    assert re.search(r"typedef struct \{ char internal\[NCCL_UNIQUE_ID_BYTES\];", text)
    assert re.search(r"ncclInt8\s*=\s*0\b", text)
    assert re.search(r"ncclCommInitRank\(ncclComm_t\*\s*comm,\s*int nranks,\s*ncclUniqueId commId,\s*int rank\)", text)
    sym = (ROOT / "include" / "symaccel.h").read_text()
    assert re.search(r"typedef struct symaccel_unique_id \{\s*char internal\[128\];", sym)
    multi = (CSRC / "multi.cpp").read_text()
    assert "int (*CommInitRank)(void **comm, int nranks, symaccel_unique_id id, int rank)" in multi and "rccl()->Send(buf, bytes, 0, peer" in multi
