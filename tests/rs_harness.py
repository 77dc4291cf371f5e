"""TEST ONLY.  Builds a tools/rsinterp interpreter in which the Rust shim crate (bindings/rust/symphonia-accel-hip) can be EXECUTED:
its sources, the generated FFI declarations bound to a libsymaccel through ctypes (tools/rsinterp/ffi.py: the CPU-emulation build
in the CPU suite, the hipcc-built library in the `-m gpu` suite), stand-ins for symphonia-core's container types
(tests/rust/core_stubs.rs, audio_stubs.rs) -- or, with `reference=True` (needs /root/reference), the reference's own io / checksum /
packet modules and a codec crate with the repository's seam patch applied."""
import os
import re
import shutil
import subprocess
import sys
import tempfile
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "tools"))

from rsinterp import Interp  # noqa: E402
from rsinterp import ffi as F  # noqa: E402
from rsinterp import interp as I  # noqa: E402
from rsinterp import parser as P  # noqa: E402

CRATE = ROOT / "bindings" / "rust" / "symphonia-accel-hip" / "src"
PATCHES = ROOT / "bindings" / "rust" / "patches"
REF = Path("/root/reference")
CODEC_CRATES = ("symphonia-bundle-flac", "symphonia-codec-aac", "symphonia-bundle-mp3", "symphonia-codec-vorbis", "symphonia-codec-alac")
CORE_IO = ("errors.rs", "util.rs", "io/mod.rs", "io/buf_reader.rs", "io/monitor_stream.rs", "checksum/crc8.rs", "checksum/crc16.rs",
           "units.rs", "packet.rs")


# The packet-level tests run whole decoders under the interpreter (1-2 s per frame and decoder).  By default they run a subset sized
# for the every-round CPU suite; SYMACCEL_PACKET_TESTS=full runs every stream (profiles/r04_packet_tests_full.log is such a run).
FULL_PACKET_TESTS = os.environ.get("SYMACCEL_PACKET_TESTS", "") == "full"


def sized(full, quick):
    return full if FULL_PACKET_TESTS else quick


def usize(v):
    return I.Int(int(v), "usize")


def u8_vec(data):
    return I.Arr([I.Int(int(b), "u8") for b in data], True)


def f32_vec(a):
    return I.Arr([I.F32(x) for x in np.asarray(a, np.float32).ravel()], True)


def i32_vec(a):
    return I.Arr([I.Int(int(x), "i32") for x in np.asarray(a).ravel()], True)


def patched_tree(crates=CODEC_CRATES):
    """A temporary copy of the reference's codec crates with bindings/rust/patches/*.diff applied; returns its root."""
    tmp = Path(tempfile.mkdtemp(prefix="seam_"))
    for c in crates:
        shutil.copytree(REF / c / "src", tmp / c / "src")
        r = subprocess.run(["patch", "-p1", "-s", "-i", str(PATCHES / (c + ".diff"))], cwd=tmp, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("patch %s does not apply: %s%s" % (c, r.stdout, r.stderr))
    return tmp


def expand_vlc_entries(src):
    """`decl_entry!(/// doc ...)` matches its `#[doc = $expr]` arm through the doc comment, which the lexer drops with the other
    comments: write the nine invocations out as the items the macro produces (io/bit.rs:52-66)"""
    def item(m):
        name, value_type, index_type = [s.strip() for s in re.sub(r"///[^\n]*\n", "", m.group(1)).strip().rstrip(",").split(",")]
        return "pub struct %s;\nimpl CodebookEntry for %s { type IndexType = %s; type ValueType = %s; }\n" % (name, name, index_type, value_type)
    return re.sub(r"decl_entry!\(((?:.|\n)*?)\);\n", item, src)


def aac_codebooks_for_the_interpreter(src):
    """symphonia-codec-aac/src/aac/codebooks.rs with two spellings changed, neither of which changes what it computes:
      * `make_basic_codebook` / `make_value_codebook` are generic in their RETURN type, which rustc infers from the declared types
        of the statics they initialise (codebooks.rs:627-650); a dynamically typed interpreter needs the type named at the call;
      * the private `iquant(usize)` shares its name with ics/pulse.rs's private `iquant(f32)`; the interpreter has one namespace."""
    src = src.replace("make_basic_codebook(&SPECTRUM_TABLES", "make_basic_codebook::<QuadsCodebook>(&SPECTRUM_TABLES")
    src = re.sub(r"make_value_codebook\(&SPECTRUM_TABLES\[(\d)\]", r"make_value_codebook::<PairsCodebook, _>(&SPECTRUM_TABLES[\1]", src)
    src = src.replace("make_value_codebook(&SPECTRUM_TABLES[10]", "make_value_codebook::<EscapeCodebook, _>(&SPECTRUM_TABLES[10]")
    return re.sub(r"\biquant\(", "codebook_iquant(", src)


def cfg_features(src, enabled):
    """what rustc's cfg-stripping does to a file whose items, enum variants, match arms and array elements carry
    `#[cfg(feature = "...")]` on a line of their own (symphonia-bundle-mp3/src/decoder.rs): the attribute line goes, and with a
    feature that is off so does the element behind it (the parser reads attributes in those places and drops them)"""
    out, lines, i = [], src.split("\n"), 0
    while i < len(lines):
        m = re.match(r'\s*#\[cfg\(feature = "(\w+)"\)\]\s*$', lines[i])
        if not m:
            out.append(lines[i])
            i += 1
            continue
        i += 1
        if m.group(1) in enabled:
            continue
        depth = 0
        while True:
            ln = lines[i]
            i += 1
            depth += sum(ln.count(c) for c in "{([") - sum(ln.count(c) for c in "})]")
            if depth <= 0 and re.search(r"[,;}]\s*$", ln):
                break
    return "\n".join(out)


MP3_FILES = ("common.rs", "header.rs", "layer3/common.rs", "layer3/codebooks.rs", "layer3/bitstream.rs", "layer3/requantize.rs",
             "layer3/stereo.rs", "layer3/hybrid_synthesis.rs", "synthesis.rs", "layer3/mod.rs")
AAC_FILES = ("common.rs", "window.rs", "dsp.rs", "ics/gain.rs", "ics/ltp.rs", "ics/pulse.rs", "ics/tns.rs", "ics/mod.rs", "cpe.rs", "mod.rs")
CORE_DSP = ("dsp/fft/mod.rs", "dsp/fft/no_simd.rs", "dsp/mdct.rs")  # the in-tree transform: what the product reproduces


_SHARED_STATICS = {}  # the reference's lazily built tables (VLC codebooks, pow43, windows): built once per process and source text


class Harness:
    def __init__(self, dll, reference=False, flac_tree=None, alac_tree=None, aac_tree=None, mp3_tree=None, vorbis_tree=None, sample=None):
        self.it = it = Interp()
        # the silence value of new AudioBuffers: untyped unless the decoder under test hands plane slices to functions declared
        # `&mut [f32]` (the reference's AAC and MP3 decoders; sample="f32")
        zero = {None: lambda: I.Int(0, None), "i32": lambda: I.Int(0, "i32"), "f32": lambda: I.F32(0.0)}["f32" if (aac_tree or mp3_tree or vorbis_tree) and sample is None else sample]
        it.globals["audio_stub_sample_mid"] = I.Builtin(zero, "audio_stub_sample_mid")
        self.dll = dll
        self.reference = reference
        if reference:
            it.shared_statics = _SHARED_STATICS
        if reference:
            for f in CORE_IO:
                it.load_file(REF / "symphonia-core" / "src" / f)
            it.load_source(expand_vlc_entries((REF / "symphonia-core/src/io/bit.rs").read_text()), "io/bit.rs")
        else:
            it.load_file(ROOT / "tests" / "rust" / "core_stubs.rs")
        it.load_file(ROOT / "tests" / "rust" / "audio_stubs.rs")
        it.native_methods[("AudioBuffer", "as_generic_audio_buffer_ref")] = self._generic_ref
        if flac_tree is not None:  # the FLAC crate (patched or not) + STREAMINFO
            for f in ("frame.rs", "decoder.rs") + (("backend.rs",) if (flac_tree / "backend.rs").exists() else ()):
                it.load_file(flac_tree / f)
            it.load_file(REF / "symphonia-common/src/xiph/audio/flac/mod.rs")
            # (symphonia-core's Position bit flags are outside the stand-in Channels)
            it.load_source("pub fn flac_channels_to_channels(channels: u32) -> Channels { Channels::Discrete(channels as u16) }", "stub")
        if alac_tree is not None:  # the ALAC crate (patched or not) + the magic cookie
            for f in (("backend.rs",) if (alac_tree / "backend.rs").exists() else ()) + ("lib.rs",):
                it.load_file(alac_tree / f)
            it.load_file(REF / "symphonia-common/src/apple/audio/alac.rs")
        if aac_tree is not None:  # the AAC crate's `aac` module (patched or not), the transforms it calls, the AudioSpecificConfig
            for f in CORE_DSP:
                it.load_file(REF / "symphonia-core" / "src" / f)
            it.load_file(REF / "symphonia-common/src/mpeg/audio/mod.rs")
            it.load_source(aac_codebooks_for_the_interpreter((aac_tree / "aac" / "codebooks.rs").read_text()), "aac/codebooks.rs")
            for f in AAC_FILES + (("backend.rs",) if (aac_tree / "aac" / "backend.rs").exists() else ()):
                it.load_file(aac_tree / "aac" / f)
        if mp3_tree is not None:  # the MP3 crate (patched or not) built with `features = ["mp3"]`, as the shim's Cargo.toml asks
            for f in MP3_FILES + (("backend.rs",) if (mp3_tree / "backend.rs").exists() else ()):
                it.load_file(mp3_tree / f)
            it.load_source(cfg_features((mp3_tree / "decoder.rs").read_text(), {"mp3"}), "decoder.rs")
        if vorbis_tree is not None:  # the Vorbis crate (patched or not), the transform it calls, the Xiph helpers
            for f in CORE_DSP:
                it.load_file(REF / "symphonia-core" / "src" / f)
            it.load_file(REF / "symphonia-common/src/xiph/audio/vorbis/mod.rs")
            # (symphonia-core's Position bit flags are outside the stand-in Channels)
            it.load_source("pub fn vorbis_channels_to_channels(num_channels: u8) -> Option<Channels> { Some(Channels::Discrete(num_channels as u16)) }", "stub")
            for f in ("common.rs", "window.rs", "codebook.rs", "floor.rs", "residue.rs", "dsp.rs") + (("backend.rs",) if (vorbis_tree / "backend.rs").exists() else ()) + ("lib.rs",):
                it.load_file(vorbis_tree / f)
        self.bridge = F.Bridge(it, (ROOT / "bindings" / "rust" / "symaccel_sys.rs").read_text(), dll) if dll is not None else None
        bad = it.globals.get("__unparsed__")
        assert not bad, bad

    @staticmethod
    def _generic_ref(buf, args):
        planes = buf.f["planes"].a
        first = planes[0].a[0] if planes and planes[0].a else None
        kind = "S32" if isinstance(first, I.Int) and first.t == "i32" else "F32"
        return I.Enum("GenericAudioBufferRef", kind, {"0": buf})

    def load_shim(self, *names):
        """crate files with `hip_decoder!` invocations expanded (decoder.rs defines the macro and must come first)"""
        for name in names:
            path = CRATE / name
            out = []
            for item in P.parse_source(path.read_text(), str(path)):
                if item[0] == "macro_item" and item[1] == "hip_decoder":
                    toks = self.it.expand_macro("hip_decoder", item[2])
                    toks = [tk for i, tk in enumerate(toks) if not (tk.s == "$" and i + 1 < len(toks) and toks[i + 1].s == "crate")]
                    out.extend(P.parse_tokens_as_items(toks, str(path)))
                else:
                    out.append(item)
            self.it.register_items(out, str(path))
        bad = self.it.globals.get("__unparsed__")
        assert not bad, bad

    # ---- values
    def params(self, codec, rate=None, nch=None, extra=None, bps=None):
        it = self.it
        return I.Struct("AudioCodecParameters", {
            "codec": it.resolve_value([codec], I.Env(), None),
            "sample_rate": I.some(I.Int(rate, "u32")) if rate else I.NONE,
            "bits_per_sample": I.some(I.Int(bps, "u32")) if bps else I.NONE,
            "channels": I.some(I.Enum("Channels", "Discrete", {"0": I.Int(nch, "u16")})) if nch else I.NONE,
            "max_frames_per_packet": I.NONE,
            "extra_data": I.some(u8_vec(extra)) if extra is not None else I.NONE})

    def opts(self, gapless=True, verify=False):
        return I.Struct("AudioDecoderOptions", {"verify": bool(verify), "gapless": bool(gapless)})

    def packet(self, data, pts, track=0, owned=False):
        """a PacketRef (or, owned=True, a Packet a MockReader can hand out)"""
        arr = u8_vec(data)
        ts = I.Struct("Timestamp", {"0": I.Int(int(pts), "i64")})
        if owned and self.reference:  # symphonia-core's own Packet (packet.rs:50-89)
            zero = I.Struct("Duration", {"0": I.Int(0, "u64")})
            return I.Struct("Packet", {"track_id": I.Int(track, "u32"), "pts": ts, "dts": ts, "dur": zero, "trim_start": zero, "trim_end": zero,
                                       "data": arr})
        if owned:
            return self.it.call("Packet::new", I.Int(track, "u32"), ts, arr)
        if self.reference:  # symphonia-core's own PacketRef (packet.rs:146-171)
            zero = I.Struct("Duration", {"0": I.Int(0, "u64")})
            return I.Struct("PacketRef", {"track_id": I.Int(track, "u32"), "pts": ts, "dts": ts, "dur": zero, "trim_start": zero, "trim_end": zero,
                                          "data": I.Slice(arr.a, 0, len(arr.a), False)})
        return I.Struct("PacketRef", {"track_id": I.Int(track, "u32"), "pts": ts, "data": I.Slice(arr.a, 0, len(arr.a), False)})

    @staticmethod
    def f32_buffers(value):
        """the stand-in AudioBuffer<S>::new cannot know S and fills its planes with untyped zeros; a decoder that hands plane slices
        to functions declared `&mut [f32]` (the reference's AAC decoder) needs them typed: walk `value`, retype every AudioBuffer"""
        from rsinterp import stdext
        seen, todo = set(), [value]
        while todo:
            v = I.deref(todo.pop())
            if id(v) in seen:
                continue
            seen.add(id(v))
            if isinstance(v, I.Struct) and v.name == "AudioBuffer":
                for plane in v.f["planes"].a:
                    plane.a[:] = [I.F32(0.0)] * len(plane.a)
            elif isinstance(v, (I.Struct, I.Enum)):
                todo.extend((v.f or {}).values())
            elif isinstance(v, I.Arr):
                todo.extend(x for x in v.a if not isinstance(x, (I.Int, float, bool, str)))
            elif isinstance(v, stdext.Cell):
                todo.append(v.v)
            elif isinstance(v, tuple):
                todo.extend(v)
        return value

    def decode(self, type_name, dec, pkt):
        """decode_ref: ('ok', planes as a numpy array [channel][frames]) or ('err', variant name); checks the buffer is cleared on error"""
        r = self.it.call_method(type_name, "decode_ref", dec, pkt)
        last = self.it.call_method(type_name, "last_decoded", dec)
        if r.variant == "Err":
            assert last.f["0"].f["num_frames"].v == 0, "the buffer must be cleared on error (codecs/audio.rs:278)"
            return "err", r.f["0"].variant
        buf = r.f["0"].f["0"]
        assert buf is last.f["0"], "last_decoded() must be the buffer decode_ref returned (codecs/audio.rs:291-297)"
        return "ok", self.planes(buf)

    @staticmethod
    def planes(buf):
        n = buf.f["num_frames"].v
        rows = []
        for p in buf.f["planes"].a:
            vals = p.a[:n]
            if vals and isinstance(vals[0], I.Int):
                rows.append(np.array([x.v for x in vals], np.int64))
            else:
                rows.append(np.array([np.float32(x) if not isinstance(x, I.Int) else np.float32(x.v) for x in vals], np.float32))
        return np.stack(rows) if rows else np.zeros((0, 0))


def registry_round_trip(h, decoder_type, params_list, opts=None):
    """`register` -> `make_audio_decoder` for every entry of params_list, the way an application gets its decoders
    (symphonia-core/src/codecs/registry.rs:252-269, 330-341): `register_one::<decoder_type>` (what lib.rs `register()` does for each
    of the five decoder types) enters the type at Tier::Preferred, the registry's factory -- `try_registry_new`, from (params, opts)
    alone -- builds the decoders.  Returns the decoders; the harness must have lib.rs, fallback.rs and the codec's adapter loaded."""
    it = h.it
    it.load_file(ROOT / "tests" / "rust" / "registry_generic.rs")
    it.load_source("pub fn register_under_test(registry: &mut CodecRegistry) { register_one::<%s>(registry, true); }" % decoder_type, "register_under_test.rs")
    reg = it.call("CodecRegistry::new")
    it.call("register_under_test", reg)
    decs = []
    for p in params_list:
        r = it.call_method("CodecRegistry", "make_registered_audio_decoder", reg, p, opts if opts is not None else h.opts())
        assert r.variant == "Ok", r
        dec = I.deref(r.f["0"])
        assert isinstance(dec, I.Struct) and dec.name == decoder_type, dec
        decs.append(dec)
    return decs


def pool_stats(h):
    """symaccel_batcher_get_stats of the process-wide Pool the interpreted crate created (ctx.rs `Pool::shared` / `Pool::stats`)"""
    pool = h.it.call("Pool::shared")
    assert pool.variant == "Ok", pool
    arc = pool.f["0"]
    r = h.it.call_method("Pool", "stats", arc.v if hasattr(arc, "v") else arc)
    assert r.variant == "Ok", r
    return {k: int(v.v) for k, v in r.f["0"].f.items()}
