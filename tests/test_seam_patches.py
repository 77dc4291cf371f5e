"""bindings/rust/patches/*.diff -- the `pub trait SynthBackend` seam each of the reference's four codec crates needs so that a
crate OUTSIDE it (bindings/rust/symphonia-accel-hip) can take over the DSP without vendoring the parse stage.

Checked here (needs /root/reference, `localref`):
  * every patch applies cleanly to the reference tree with `patch -p1`;
  * every file a patch touches or adds still parses, down to every function body (tools/rsinterp/parser.py);
  * each patch defines `pub trait SynthBackend: Send + Sync`, a default (CPU) implementation that calls the crate's own code, and a
    public `try_new_with_backend` constructor, while `try_new` keeps its signature (behaviour unchanged: tests/test_flac_packets.py
    executes the patched FLAC decoder against the unpatched one);
  * the recording backends of the shim crate implement exactly the patched traits (method names, receivers, parameter and return
    type shapes), so that `frontends.rs` -- no longer a stub -- builds real front ends, and `register()` registers all four decoders;
  * the stand-in container types the executed tests use (tests/rust/audio_stubs.rs) have the method names and arities of the reference's.
"""
import re
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "tests"))
sys.path.insert(0, str(ROOT / "tools"))

from rs_harness import CODEC_CRATES, CRATE, PATCHES, REF, patched_tree  # noqa: E402
from rsinterp import parser as P  # noqa: E402
from test_rust_shim import expanded_items, type_shape, walk_items  # noqa: E402

pytestmark = pytest.mark.localref

SEAM = {  # crate -> (file defining the trait, shim file with the recording backend, the decoder type, file with the constructor)
    "symphonia-bundle-flac": ("src/backend.rs", "flac.rs", "FlacDecoder", "src/decoder.rs"),
    "symphonia-codec-aac": ("src/aac/backend.rs", "aac.rs", "AacDecoder", "src/aac/mod.rs"),
    "symphonia-bundle-mp3": ("src/backend.rs", "mpa.rs", "MpaDecoder", "src/decoder.rs"),
    "symphonia-codec-vorbis": ("src/backend.rs", "vorbis.rs", "VorbisDecoder", "src/lib.rs"),
    "symphonia-codec-alac": ("src/backend.rs", "alac.rs", "AlacDecoder", "src/lib.rs"),
}


@pytest.fixture(scope="module")
def tree():
    return patched_tree()


def touched_files(crate):
    text = (PATCHES / (crate + ".diff")).read_text()
    return sorted(set(re.findall(r"^\+\+\+ b/(\S+)", text, flags=re.M)))


def test_every_patch_applies_and_the_patched_files_parse(tree):
    n_fns = 0
    for crate in CODEC_CRATES:
        files = touched_files(crate)
        assert files and all(f.startswith(crate + "/src/") for f in files), files
        for f in files:
            items = P.parse_source((tree / f).read_text(), f)
            for it in walk_items(items):
                assert it[0] != "unparsed", "%s: %s" % (f, it[1])
                if it[0] == "fn" and it[6] is not None:
                    it[8].parse_body(it[6])
                    n_fns += 1
    assert n_fns > 100


def trait_of(path, name="SynthBackend"):
    for it in P.parse_source(path.read_text(), str(path)):
        if it[0] == "trait" and it[1] == name:
            return {m[1]: m for m in it[2] if m[0] == "fn"}
    raise AssertionError("no trait %s in %s" % (name, path))


@pytest.mark.parametrize("crate", CODEC_CRATES)
def test_the_seam_is_a_public_trait_with_the_cpu_code_as_default(tree, crate):
    trait_file, _, decoder, ctor_file = SEAM[crate]
    text = (tree / crate / trait_file).read_text()
    assert re.search(r"pub trait SynthBackend\s*:\s*Send \+ Sync", text)
    methods = trait_of(tree / crate / trait_file)
    # every operation is `&mut self`; the queries of the second-generation seams (`takes_*`: does the backend also want the stage in
    # front of synthesis?) are `&self`
    assert methods and all(m[4] == "ref_mut" or (name.startswith("takes_") and m[4] == "ref") for name, m in methods.items())
    ctor = (tree / crate / ctor_file).read_text()
    assert re.search(r"pub fn try_new_with_backend\(", ctor), "no public constructor taking a backend"
    before = (REF / crate / ctor_file).read_text()
    sig = re.search(r"pub fn try_new\(([^)]*)\)\s*->\s*Result<Self>", before).group(1)
    assert re.sub(r"\b_opts\b", "opts", sig) in re.sub(r"\b_opts\b", "opts", ctor), "try_new changed its signature"
    # the default path is the crate's own code: either a CPU impl of the trait that calls it, or `None` = the old code path
    diff = (PATCHES / (crate + ".diff")).read_text()
    if crate == "symphonia-bundle-flac":
        assert "impl SynthBackend for CpuBackend" in text and "decoder::fixed_predict(order, buf)" in text and "decoder::lpc_dispatch(" in text
    elif crate == "symphonia-codec-aac":
        assert "impl SynthBackend for Dsp" in text and "Dsp::synth(self, coeffs, delay" in text
    elif crate == "symphonia-codec-alac":
        assert "impl SynthBackend for CpuBackend" in text and "crate::predict(pred, out)" in text and "crate::decorrelate_mid_side(" in text
    else:
        assert "if let Some(backend) = self.backend.as_mut()" in diff and "continue;" in diff
    # the seam is reachable from outside the crate
    lib = (tree / crate / "src/lib.rs").read_text()
    assert re.search(r"pub (mod backend|use [\w:]*backend::)", lib) or "pub use aac::backend::" in lib


@pytest.mark.parametrize("crate", CODEC_CRATES)
def test_the_shim_recorders_implement_the_patched_traits(tree, crate):
    trait_file, shim_file, _, _ = SEAM[crate]
    want = trait_of(tree / crate / trait_file)
    found = False
    for it in expanded_items(CRATE / shim_file):
        if it[0] != "impl" or it[2] is None or it[2][1][-1] != "SynthBackend":
            continue
        found = True
        have = {m[1]: m for m in it[3] if m[0] == "fn"}
        assert set(have) == set(want), (crate, sorted(have), sorted(want))
        for n, m in have.items():
            w = want[n]
            assert m[4] == w[4] and len(m[3]) == len(w[3]), (crate, n)
            for (_, ta), (_, tb) in zip(m[3], w[3]):
                assert type_shape(ta) == type_shape(tb), (crate, n, type_shape(ta), type_shape(tb))
            assert type_shape(m[5]) == type_shape(w[5]), (crate, n)
    assert found, "no `impl SynthBackend for ..` in %s" % shim_file
    shim = (CRATE / shim_file).read_text()
    assert "try_new_with_backend(" in shim and "struct SeamFrontEnd" in shim


def test_register_registers_all_five_decoders():
    fe = (CRATE / "frontends.rs").read_text()
    assert "Available { aac: true, mpa: true, vorbis: true, flac: true, alac: true }" in fe
    assert "unsupported_error" not in fe  # no stub front end is left
    for codec in ("aac", "mpa", "vorbis", "flac", "alac"):
        assert re.search(r"pub fn %s_front_end\(.*\) -> Result<Box<dyn \w+FrontEnd>> \{\s*Ok\(Box::new\(crate::%s::SeamFrontEnd::try_new\(" % (codec, codec), fe), codec
    lib = (CRATE / "lib.rs").read_text()
    for dec in ("HipAacDecoder", "HipMpaDecoder", "HipVorbisDecoder", "HipFlacDecoder", "HipAlacDecoder"):
        assert "register_one::<%s>(registry, frontends::AVAILABLE." % dec in lib


def test_the_audio_stand_ins_have_the_shape_of_the_reference_types():
    core = REF / "symphonia-core" / "src"
    stubs = {it[1]: it for it in P.parse_source((ROOT / "tests/rust/audio_stubs.rs").read_text(), "audio_stubs.rs") if it[0] in ("struct", "enum")}
    impls = {}
    for it in P.parse_source((ROOT / "tests/rust/audio_stubs.rs").read_text(), "audio_stubs.rs"):
        if it[0] == "impl":
            impls.setdefault(it[1][1][-1], {}).update({m[1]: m for m in it[3] if m[0] == "fn"})
    ref_fns = {}
    for f in ("audio/buf.rs", "audio/mod.rs", "audio/channels.rs", "codecs/audio.rs"):
        for it in walk_items(P.parse_source((core / f).read_text(), f)):
            if it[0] == "fn":
                ref_fns.setdefault(it[1], []).append(it)
    for ty, methods in impls.items():
        if ty == "Validator":
            continue
        for name, m in methods.items():
            assert name in ref_fns, "%s::%s is not a function of the reference" % (ty, name)
            assert any(len(r[3]) == len(m[3]) and r[4] == m[4] for r in ref_fns[name]), "%s::%s: arity / receiver differ from the reference" % (ty, name)
    ref_params = next(it for it in P.parse_source((core / "codecs/audio.rs").read_text(), "audio.rs") if it[0] == "struct" and it[1] == "AudioCodecParameters")
    assert {f for f, _ in stubs["AudioCodecParameters"][3]} <= {f for f, _ in ref_params[3]}
    gen = (core / "audio/generic.rs").read_text()
    assert "S32(&'a AudioBuffer<i32>)" in gen and "F32(&'a AudioBuffer<f32>)" in gen
