"""The trait-side Rust shim (bindings/rust/symphonia-accel-hip) cannot be compiled here (no Rust toolchain).  What can be
checked without one, is:

  * syntax: every file of the crate parses (tools/rsinterp/parser.py), down to every function body;
  * imports: every `use symphonia_core::...` names an item the reference tree defines in that module      [localref]
  * traits: every `impl AudioDecoder / RegisterableAudioDecoder / FormatReader for ..` -- the ones the `hip_decoder!` macro
    expands to included -- has the methods of the trait TEXT in the reference (names, receivers, parameter count, return
    type), implements every method without a default body and nothing the trait does not have         [localref]
  * calls: every method the crate calls on a symphonia-core value exists somewhere in symphonia-core        [localref]
  * behaviour: lookahead.rs and fallback.rs are EXECUTED under the repository's Rust interpreter with a mock codec, a
    mock demuxer and a mock registry (tests/rust/), on the packet script of the compiled C++ twin
    (tests/cpp/lookahead_test.cpp): sequential decode across batch boundaries, reset() after a seek, a discontinuity
    without reset(), corrupt packets, device errors, two containers with equal track ids, cloned packets, and the
    registry's no-fall-through rule (codecs/registry.rs:152-154, 330-341).
"""
import re
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "tools"))

from rsinterp import Interp  # noqa: E402
from rsinterp import interp as I  # noqa: E402
from rsinterp import parser as P  # noqa: E402

CRATE = ROOT / "bindings" / "rust" / "symphonia-accel-hip" / "src"
REF = Path("/root/reference")
CORE = REF / "symphonia-core" / "src"
localref = pytest.mark.localref


# ------------------------------------------------------------------------------------------------ syntax

def walk_items(items):
    for it in items:
        yield it
        if it[0] == "impl":
            yield from walk_items(it[3])
        elif it[0] in ("mod", "trait"):
            yield from walk_items(it[2])


def crate_files():
    return sorted(CRATE.glob("*.rs"))


def expanded_items(path):
    """The items of a crate file with `hip_decoder!` invocations expanded (decoder.rs defines the macro)."""
    items = P.parse_source(path.read_text(), str(path))
    out = []
    mac = Interp()
    mac.load_file(CRATE / "decoder.rs")
    for it in items:
        if it[0] == "macro_item" and it[1] == "hip_decoder":
            toks = mac.expand_macro("hip_decoder", it[2])
            toks = [tk for i, tk in enumerate(toks) if not (tk.s == "$" and i + 1 < len(toks) and toks[i + 1].s == "crate")]  # $crate -> crate
            out.extend(P.parse_tokens_as_items(toks, str(path)))
        else:
            out.append(it)
    return out


def test_every_file_of_the_crate_parses_down_to_every_function_body():
    n_fns = 0
    for path in crate_files():
        for it in walk_items(expanded_items(path)):
            assert it[0] != "unparsed", "%s: %s" % (path.name, it[1])
            if it[0] == "fn" and it[6] is not None:
                it[8].parse_body(it[6])  # raises ParseError on a syntax error
                n_fns += 1
    assert n_fns >= 60
    # the macro body itself, with its metavariables bound like mpa.rs binds them
    assert any(it[0] == "impl" for it in expanded_items(CRATE / "mpa.rs"))


def test_generated_ffi_include_parses():
    text = (ROOT / "bindings" / "rust" / "symaccel_sys.rs").read_text()
    # extern blocks are outside the interpreter's subset: check the declarations line by line instead
    decls = re.findall(r"pub fn (symaccel_\w+)\(([^)]*)\)\s*(->\s*[\w:*<> ]+)?;", text)
    assert len(decls) >= 60
    for name, params, _ in decls:
        for prm in [x for x in params.split(",") if x.strip()]:
            assert re.fullmatch(r"\s*\w+:\s*(\*(const|mut)\s+)*[\w:]+\s*", prm), (name, prm)


# ------------------------------------------------------------------------------------------------ imports

def use_paths(text):
    """symphonia_core paths a file imports: [('codecs', 'audio', 'AudioDecoder'), ...]"""
    out = []
    for m in re.finditer(r"^\s*(?:pub\s+)?use\s+(symphonia_core::[^;]+);", re.sub(r"//.*", "", text), flags=re.M):
        def expand(s):
            s = s.strip()
            mm = re.fullmatch(r"(.*?)::\{(.*)\}", s, flags=re.S)
            if not mm:
                return [s]
            depth, cur, parts = 0, "", []
            for ch in mm.group(2):
                if ch == "{":
                    depth += 1
                elif ch == "}":
                    depth -= 1
                if ch == "," and depth == 0:
                    parts.append(cur)
                    cur = ""
                else:
                    cur += ch
            parts.append(cur)
            return [x for p in parts if p.strip() for x in expand(mm.group(1) + "::" + p.strip())]
        for full in expand(m.group(1)):
            out.append(tuple(s.strip().split(" as ")[0].strip() for s in full.split("::"))[1:])
    return out


def module_file(segs):
    p = CORE
    for s in segs:
        p = p / s
    for cand in (p.with_suffix(".rs"), p / "mod.rs"):
        if cand.exists():
            return cand
    return None


def defines(path, name, seen=None):
    seen = seen or set()
    if path in seen:
        return False
    seen.add(path)
    text = path.read_text()
    if re.search(r"\bpub\s+(?:unsafe\s+)?(?:const\s+)?(?:struct|enum|trait|fn|type|const|static|mod|union)\s+%s\b" % re.escape(name), text):
        return True
    if re.search(r"macro_rules!\s+%s\b" % re.escape(name), text):
        return True
    for m in re.finditer(r"pub use ([^;]+);", text):
        if re.search(r"\b%s\b" % re.escape(name), m.group(1)):
            return True
        g = re.fullmatch(r"\s*(?:self::|super::)?(\w+)::\*\s*", m.group(1))  # glob re-export of a child module
        if g:
            base = path.parent if path.name == "mod.rs" else path.parent / path.stem
            for cand in (base / (g.group(1) + ".rs"), base / g.group(1) / "mod.rs"):
                if cand.exists() and defines(cand, name, seen):
                    return True
    return False


@localref
def test_every_symphonia_core_import_names_an_item_the_reference_defines():
    checked = 0
    for path in crate_files():
        for segs in use_paths(path.read_text()):
            *mod, name = segs
            if not mod:  # `symphonia_core::support_audio_codec`: a #[macro_export] macro lives at the crate root
                hits = [p for p in CORE.rglob("*.rs") if re.search(r"macro_rules!\s+%s\b" % name, p.read_text())]
                assert hits, "%s: symphonia_core::%s is not an exported macro of the reference" % (path.name, name)
                checked += 1
                continue
            f = module_file(mod)
            # `well_known` style inline modules: look in the parent file for `pub mod <last> { ... }`
            if f is None and module_file(mod[:-1]) is not None:
                parent = module_file(mod[:-1]).read_text()
                assert re.search(r"pub mod %s\b" % mod[-1], parent), "%s: no module %s" % (path.name, "::".join(mod))
                assert re.search(r"\b%s\b" % re.escape(name), parent), "%s: %s not found in %s" % (path.name, name, "::".join(mod))
                checked += 1
                continue
            assert f is not None, "%s: symphonia_core::%s is not a module of the reference" % (path.name, "::".join(mod))
            assert defines(f, name), "%s: %s does not define or re-export `%s`" % (path.name, f.relative_to(REF), name)
            checked += 1
    assert checked >= 40


@localref
def test_the_alac_cookie_import_names_what_symphonia_common_exports():
    """alac.rs reads the frame length and channel count from symphonia_common::apple::audio::alac::MagicCookie"""
    src = (CRATE / "alac.rs").read_text()
    assert "use symphonia_common::apple::audio::alac::MagicCookie;" in src
    ref = (REF / "symphonia-common/src/apple/audio/alac.rs").read_text()
    assert re.search(r"pub struct MagicCookie\b", ref) and re.search(r"pub fn read\(mut buf: &\[u8\]\) -> Result<MagicCookie>", ref)
    for field in ("frame_length: u32", "num_channels: u8"):
        assert "pub " + field in ref
    for mod in ("apple", "audio", "alac"):
        assert re.search(r"pub mod %s\b" % mod, "\n".join(q.read_text() for q in (REF / "symphonia-common/src").rglob("*.rs") if q.name in ("lib.rs", "mod.rs")))


# ------------------------------------------------------------------------------------------------ traits

TRAIT_FILES = {"AudioDecoder": "codecs/audio.rs", "RegisterableAudioDecoder": "codecs/registry.rs", "FormatReader": "formats/mod.rs"}


def trait_methods(name):
    items = P.parse_source((CORE / TRAIT_FILES[name]).read_text(), TRAIT_FILES[name])
    for it in items:
        if it[0] == "trait" and it[1] == name:
            return {m[1]: m for m in it[2] if m[0] == "fn"}
    raise AssertionError("trait %s not found" % name)


def type_shape(ty):
    """A type reduced to what must agree between a trait method and its implementation: last path segments, reference
    and slice structure, generic arguments; lifetimes and module prefixes dropped, `Self` kept."""
    if ty is None:
        return "()"
    k = ty[0]
    if k == "tpath":
        args = [type_shape(g[1]) for g in ty[2] if g[0] == "gtype"]
        return ty[1][-1] + ("<" + ",".join(args) + ">" if args else "")
    if k == "tref":
        return "&" + ("mut " if ty[1] else "") + type_shape(ty[2])
    if k == "tslice":
        return "[" + type_shape(ty[1]) + "]"
    if k == "ttuple":
        return "(" + ",".join(type_shape(t) for t in ty[1]) + ")"
    return k


@localref
def test_trait_impls_match_the_trait_text_of_the_reference():
    seen = {}
    for path in crate_files():
        for it in expanded_items(path):
            if it[0] != "impl" or it[2] is None:
                continue
            tname = it[2][1][-1] if it[2][0] == "tpath" else None
            if tname not in TRAIT_FILES:
                continue
            want = trait_methods(tname)
            have = {m[1]: m for m in it[3] if m[0] == "fn"}
            who = "%s: impl %s for %s" % (path.name, tname, type_shape(it[1]))
            extra = set(have) - set(want)
            assert not extra, "%s has methods the trait does not: %s" % (who, sorted(extra))
            missing = {n for n, m in want.items() if m[6] is None} - set(have)
            assert not missing, "%s lacks required methods: %s" % (who, sorted(missing))
            for n, m in have.items():
                w = want[n]
                assert m[4] == w[4], "%s::%s receiver %s, the trait has %s" % (who, n, m[4], w[4])
                assert len(m[3]) == len(w[3]), "%s::%s takes %d parameters, the trait %d" % (who, n, len(m[3]), len(w[3]))
                for (_, ta), (_, tb) in zip(m[3], w[3]):
                    assert type_shape(ta) == type_shape(tb), "%s::%s parameter %s vs %s" % (who, n, type_shape(ta), type_shape(tb))
                assert type_shape(m[5]) == type_shape(w[5]), "%s::%s returns %s, the trait %s" % (who, n, type_shape(m[5]), type_shape(w[5]))
            seen.setdefault(tname, []).append(type_shape(it[1]))
    assert sorted(seen["AudioDecoder"]) == ["HipAacDecoder", "HipAlacDecoder", "HipFlacDecoder", "HipMpaDecoder", "HipVorbisDecoder"]
    assert sorted(seen["RegisterableAudioDecoder"]) == sorted(seen["AudioDecoder"])
    assert seen["FormatReader"] == ["LookaheadReader"]


# ------------------------------------------------------------------------------------------------ calls

STD_METHODS = set("""
    len is_empty iter iter_mut into_iter enumerate take map collect clone copied cloned unwrap unwrap_or unwrap_or_default expect
    lock push push_back pop_front pop_back front back clear fill max min as_ptr as_mut_ptr as_slice as_mut_slice as_ref as_mut
    copy_from_slice to_str to_string_lossy into_owned is_null add write get get_mut insert remove entry or_default or_insert
    or_insert_with values_mut values retain upgrade strong_count is_none is_some and_then or_else ok_or first last to_vec
    with_capacity extend extend_from_slice contains_key contains sort get_or_insert_with get_or_init last_mut chunks chunks_exact zip rev sum fold any all
    position find filter resize truncate drain swap split_at split_at_mut reserve capacity saturating_sub checked_sub
    wrapping_add wrapping_sub wrapping_shl wrapping_shr wrapping_mul leading_zeros trailing_zeros abs signum min_by max_by count skip step_by flat_map keys
    ok err map_err unwrap_or_else is_ok is_err then then_some into try_into from iter_chunks other pow
""".split())


@localref
def test_every_method_called_on_a_reference_type_exists_in_the_reference():
    core_fns = set()
    for p in CORE.rglob("*.rs"):
        core_fns.update(re.findall(r"\bfn\s+(\w+)", p.read_text()))
    crate_fns = set()
    for path in crate_files():
        crate_fns.update(re.findall(r"\bfn\s+(\w+)", path.read_text()))
    unknown = {}
    for path in crate_files():
        text = re.sub(r"//.*", "", path.read_text())
        text = re.sub(r'"(?:[^"\\]|\\.)*"', '""', text)
        for name in re.findall(r"\.\s*(\w+)\s*(?:::<[^>]*>)?\(", text):
            if name not in STD_METHODS and name not in crate_fns and name not in core_fns:
                unknown.setdefault(path.name, set()).add(name)
    assert not unknown, "methods that neither std, the crate nor symphonia-core define: %r" % unknown
    # the ones the previous revision of the shim got wrong stay wrong in the reference:
    assert "to_packet" not in core_fns and "as_packet_ref" in core_fns


@localref
def test_test_stubs_have_the_shape_of_the_reference_items():
    stubs = P.parse_source((ROOT / "tests" / "rust" / "core_stubs.rs").read_text(), "core_stubs.rs")
    ref_packet = P.parse_source((CORE / "packet.rs").read_text(), "packet.rs")
    ref_structs = {it[1]: {f for f, _ in it[3]} for it in ref_packet if it[0] == "struct"}
    for it in stubs:
        if it[0] == "struct" and it[1] in ("Packet", "PacketRef"):
            assert {f for f, _ in it[3]} <= ref_structs[it[1]], it[1]
    units = (CORE / "units.rs").read_text()
    assert re.search(r"pub const fn get\(self\) -> i64", units) and re.search(r"pub const fn new\(ts: i64\) -> Self", units)
    reg = (CORE / "codecs" / "registry.rs").read_text()
    for name in ("get_audio_decoder", "get_audio_decoder_at_tier", "make_audio_decoder", "register_audio_decoder_at_tier"):
        assert re.search(r"pub fn %s\b" % name, reg), name
    assert "self.preferred.get(id).or_else(|| self.standard.get(id)).or_else(|| self.fallback.get(id))" in reg  # no fall-through on error
    assert re.search(r"pub factory: AudioDecoderFactoryFn", reg)


# ------------------------------------------------------------------------------------------------ behaviour (executed)

def u32(v):
    return I.Int(v, "u32")


def i64(v):
    return I.Int(v, "i64")


def usize(v):
    return I.Int(v, "usize")


class Rig:
    """The shim's lookahead.rs + fallback.rs under the interpreter, with the mocks of tests/rust/."""

    def __init__(self):
        it = Interp()
        for f in (ROOT / "tests" / "rust" / "core_stubs.rs", CRATE / "lookahead.rs", CRATE / "fallback.rs", ROOT / "tests" / "rust" / "mocks.rs"):
            it.load_file(f)
        assert not it.globals.get("__unparsed__")
        self.it = it

    def packets(self, values, track=1, pts0=0):
        return [self.it.call("make_packet", u32(track), i64(pts0 + 10 * i), i64(v)) for i, v in enumerate(values)]

    def reader(self, packets, depth):
        inner = self.it.call("MockReader::new", I.Arr(list(packets), True))
        reader = self.it.call("LookaheadReader::new", inner, usize(depth))
        return reader, reader.f["inner"]  # (`inner` was moved into the reader: the interpreter's move is a copy)

    def decoder(self, max_batch, pooled=False):
        return self.it.call("MockCodec::new_pooled" if pooled else "MockCodec::new"), self.it.call("Lookahead::new", usize(max_batch))

    def next_packet(self, reader):
        r = self.it.call_method("LookaheadReader", "next_packet", reader)
        if r.variant == "Err":
            return "err", r.f["0"]
        o = r.f["0"]
        return ("eof", None) if o.variant == "None" else ("ok", o.f["0"])

    def decode(self, la, codec, packet):
        """decode_ref(&packet.as_packet_ref()): the buffer's content, or the error's variant name"""
        pr = self.it.call_method("Packet", "as_packet_ref", packet)
        r = self.it.call_method("Lookahead", "decode", la, codec, pr)
        buf = codec.f["buffer"]
        if r.variant == "Err":
            assert buf.variant == "None", "the buffer must be cleared on error (codecs/audio.rs:278)"
            return r.f["0"].variant
        assert buf.variant == "Some"
        return buf.f["0"].v

    @staticmethod
    def batch_sizes(codec):
        return [x.v for x in codec.f["batch_sizes"].a]


class FrameByFrame:
    """What a decoder without look-ahead gives for MockCodec: the model every script is compared with."""

    def __init__(self):
        self.state = 0

    def reset(self):
        self.state = 0

    def decode(self, value):
        out = value * 100000 + self.state
        self.state = value
        return out


@pytest.fixture()
def rig():
    return Rig()


def test_sequential_decode_is_frame_by_frame_across_batch_boundaries(rig):
    values = [3, 1, 4, 1, 5, 9, 2, 6, 5, 3, 5]
    reader, inner = rig.reader(rig.packets(values), depth=6)
    codec, la = rig.decoder(4)
    ref = FrameByFrame()
    for v in values:
        st, p = rig.next_packet(reader)
        assert st == "ok"
        assert rig.decode(la, codec, p) == ref.decode(v)
    assert rig.next_packet(reader)[0] == "eof"
    assert rig.batch_sizes(codec) == [4, 4, 3]      # one transform per max_batch packets, not one per packet
    assert codec.f["parses"].v == len(values)       # every packet parsed exactly once
    assert inner.f["reads"].v == len(values) + 1    # the demuxer was read ahead, each packet once (+ the end)


def test_reset_after_a_seek(rig):
    values = list(range(1, 21))
    reader, inner = rig.reader(rig.packets(values), depth=8)
    codec, la = rig.decoder(5)
    ref = FrameByFrame()
    for _ in range(7):
        st, p = rig.next_packet(reader)
        assert rig.decode(la, codec, p) == ref.decode(p.f["data"].a[0].v)
    # seek to packet 12: the reader drops what it read ahead, the application resets the decoder (audio.rs:252-257)
    rig.it.call_method("LookaheadReader", "seek", reader, i64(0), usize(12))
    rig.it.call_method("MockCodec", "reset_state", codec)
    rig.it.call_method("Lookahead", "reset", la)
    ref.reset()
    got = []
    while True:
        st, p = rig.next_packet(reader)
        if st == "eof":
            break
        got.append(p.f["pts"].f["0"].v)
        assert rig.decode(la, codec, p) == ref.decode(p.f["data"].a[0].v)
    assert got == [10 * i for i in range(12, 20)]


def test_discontinuity_without_reset_continues_from_the_last_returned_packet(rig):
    """Packets dropped by the application without reset(): a frame-by-frame decoder carries on from the state the last
    RETURNED packet left -- not from the end of the batch that was pre-computed (the C++ twin replays that packet too)."""
    values = [7, 8, 9, 10, 11, 12, 13, 14, 15, 16]
    reader, _ = rig.reader(rig.packets(values), depth=8)
    codec, la = rig.decoder(6)
    ref = FrameByFrame()
    taken = []
    for _ in range(10):
        st, p = rig.next_packet(reader)
        taken.append(p)
    for i in (0, 1, 2):
        assert rig.decode(la, codec, taken[i]) == ref.decode(values[i])
    assert rig.batch_sizes(codec) == [1, 1, 1]  # (all ten were handed out before the first decode: the queue does not follow packet 0)
    # now with a real look-ahead: a fresh rig state
    reader, _ = rig.reader(rig.packets(values, track=2), depth=8)
    codec, la = rig.decoder(6)
    ref = FrameByFrame()
    st, p0 = rig.next_packet(reader)
    assert rig.decode(la, codec, p0) == ref.decode(values[0])       # batch of 6: packets 0..5 transformed
    st, p1 = rig.next_packet(reader)
    assert rig.decode(la, codec, p1) == ref.decode(values[1])
    for _ in range(3):                                              # packets 2, 3, 4 are read and dropped
        rig.next_packet(reader)
    st, p5 = rig.next_packet(reader)
    # packet 5 IS in the pre-computed batch, but computed from packet 4's state: the frame-by-frame answer uses packet 1's
    assert rig.decode(la, codec, p5) == ref.decode(values[5])
    st, p6 = rig.next_packet(reader)
    assert rig.decode(la, codec, p6) == ref.decode(values[6])
    assert rig.batch_sizes(codec)[:3] == [6, 1, 5]                  # batch, replay of packet 1, new batch: packet 5 + the four queued


def test_a_corrupt_packet_ends_the_batch_and_fails_at_its_own_decode(rig):
    pk = rig.packets([1, 2, 3, 0, 5, 6, 7])
    pk[3] = rig.it.call("corrupt_packet", u32(1), i64(30))
    reader, _ = rig.reader(pk, depth=7)
    codec, la = rig.decoder(7)
    ref = FrameByFrame()
    out = []
    for i in range(7):
        st, p = rig.next_packet(reader)
        out.append(rig.decode(la, codec, p))
    want = [ref.decode(1), ref.decode(2), ref.decode(3), "DecodeError", ref.decode(5), ref.decode(6), ref.decode(7)]
    assert out == want
    assert rig.batch_sizes(codec) == [3, 3]  # the look-ahead stopped in front of the corrupt packet, then resumed behind it


def test_a_device_error_clears_the_buffer_and_the_next_call_starts_over(rig):
    values = [4, 5, 6, 7]
    reader, _ = rig.reader(rig.packets(values), depth=4)
    codec, la = rig.decoder(2)
    ref = FrameByFrame()
    st, p = rig.next_packet(reader)
    assert rig.decode(la, codec, p) == ref.decode(4)
    st, p = rig.next_packet(reader)
    assert rig.decode(la, codec, p) == ref.decode(5)
    codec.f["fail_transform_in"] = i64(0)
    st, p = rig.next_packet(reader)
    assert rig.decode(la, codec, p) == "IoError"
    assert rig.it.call_method("Lookahead", "precomputed", la).v == 0
    st, p = rig.next_packet(reader)
    assert rig.decode(la, codec, p) == ref.decode(7)  # (packet 6 was lost to the error, like with any decoder)


def test_two_containers_with_equal_track_ids_do_not_share_a_queue(rig):
    a_vals, b_vals = [1, 2, 3, 4, 5, 6], [101, 102, 103, 104, 105, 106]
    ra, _ = rig.reader(rig.packets(a_vals, track=1), depth=6)
    rb, _ = rig.reader(rig.packets(b_vals, track=1), depth=6)   # same track id, same pts
    ca, la = rig.decoder(4)
    cb, lb = rig.decoder(4)
    fa, fb = FrameByFrame(), FrameByFrame()
    for i in range(6):
        st, pa = rig.next_packet(ra)
        st, pb = rig.next_packet(rb)
        assert rig.decode(la, ca, pa) == fa.decode(a_vals[i])
        assert rig.decode(lb, cb, pb) == fb.decode(b_vals[i])
    assert rig.batch_sizes(ca) == [4, 2] and rig.batch_sizes(cb) == [4, 2]


def test_tracks_of_one_container_are_queued_separately(rig):
    a, b = rig.packets([1, 2, 3, 4], track=1), rig.packets([11, 12, 13, 14], track=2)
    inter = [x for pair in zip(a, b) for x in pair]
    reader, _ = rig.reader(inter, depth=8)
    c1, l1 = rig.decoder(8)
    c2, l2 = rig.decoder(8)
    f1, f2 = FrameByFrame(), FrameByFrame()
    for i in range(8):
        st, p = rig.next_packet(reader)
        if p.f["track_id"].v == 1:
            assert rig.decode(l1, c1, p) == f1.decode(p.f["data"].a[0].v)
        else:
            assert rig.decode(l2, c2, p) == f2.decode(p.f["data"].a[0].v)
    assert rig.batch_sizes(c1) == [4] and rig.batch_sizes(c2) == [4]


def test_without_a_reader_or_with_cloned_packets_every_call_is_a_batch_of_one(rig):
    values = [9, 8, 7, 6]
    codec, la = rig.decoder(4)
    ref = FrameByFrame()
    for p in rig.packets(values, track=7):  # no LookaheadReader at all
        assert rig.decode(la, codec, p) == ref.decode(p.f["data"].a[0].v)
    assert rig.batch_sizes(codec) == [1, 1, 1, 1]
    reader, _ = rig.reader(rig.packets(values, track=8), depth=4)
    codec, la = rig.decoder(4)
    ref = FrameByFrame()
    for v in values:
        st, p = rig.next_packet(reader)
        clone = rig.it.builtin_method(p, "clone", [], None, None)  # the application decodes a copy: no identity, no look-ahead
        assert rig.decode(la, codec, clone) == ref.decode(v)
    assert rig.batch_sizes(codec) == [1, 1, 1, 1]


def test_a_read_error_of_the_inner_reader_is_reported_and_reading_continues(rig):
    reader, inner = rig.reader(rig.packets([1, 2, 3, 4, 5]), depth=2)
    inner.f["fail_at"] = i64(3)
    seen = []
    for _ in range(8):
        st, p = rig.next_packet(reader)
        seen.append(st if st != "ok" else p.f["data"].a[0].v)
        if st == "eof":
            break
    # the inner reader fails in front of its fourth packet: the three packets read before that are all delivered, THEN the error
    # (where the inner reader itself would have returned it), then reading goes on -- ADVICE r3: an error met while reading
    # AHEAD must not swallow the packets already read
    assert seen == [1, 2, 3, "err", 4, 5, "eof"]


def test_a_failing_inner_reader_loses_no_packet_at_any_depth(rig):
    for depth in (1, 2, 3, 8):
        reader, inner = rig.reader(rig.packets(list(range(1, 11))), depth=depth)
        inner.f["fail_at"] = i64(6)
        seen = []
        for _ in range(14):
            st, p = rig.next_packet(reader)
            seen.append(st if st != "ok" else p.f["data"].a[0].v)
            if st == "eof":
                break
        assert seen == [1, 2, 3, 4, 5, 6, "err", 7, 8, 9, 10, "eof"], (depth, seen)


# ---- the cross-stream batcher form of the same decoder (BatchCodec::pooled / submit / collect / hint / abandon): the next batch is
# parsed and submitted while the current one is still being handed out; what the application sees does not change

def test_pooled_decode_is_frame_by_frame_and_submits_ahead(rig):
    values = [3, 1, 4, 1, 5, 9, 2, 6, 5, 3, 5, 8, 9, 7, 9, 3, 2, 3, 8, 4, 6]
    reader, inner = rig.reader(rig.packets(values), depth=12)
    codec, la = rig.decoder(4, pooled=True)
    ref = FrameByFrame()
    for i, v in enumerate(values):
        st, p = rig.next_packet(reader)
        assert st == "ok"
        assert rig.decode(la, codec, p) == ref.decode(v)
        if i == 1:  # half of the first batch of four handed out: the next four are with the batcher already
            assert codec.f["submits"].v == 1 and codec.f["in_flight"] is True and codec.f["parses"].v == 8
        if i == 2:  # a quarter left: the hint
            assert codec.f["hints"].v == 1
    assert rig.next_packet(reader)[0] == "eof"
    assert codec.f["parses"].v == len(values)                       # every packet parsed exactly once, in order
    assert rig.batch_sizes(codec) == [4, 4, 4, 4, 4, 1]              # the first by transform, the others collected
    assert codec.f["submits"].v == 5 and codec.f["collects"].v == 5 and codec.f["abandons"].v == 0 and codec.f["in_flight"] is False


def test_pooled_reset_after_a_seek_gives_up_the_submitted_batch(rig):
    values = list(range(1, 25))
    reader, inner = rig.reader(rig.packets(values), depth=12)
    codec, la = rig.decoder(4, pooled=True)
    ref = FrameByFrame()
    for _ in range(7):
        st, p = rig.next_packet(reader)
        assert rig.decode(la, codec, p) == ref.decode(p.f["data"].a[0].v)
    assert codec.f["in_flight"] is True
    # seek: the application seeks the reader and resets the decoder (hip_decoder!'s reset: reset_with, then reset_state)
    rig.it.call_method("LookaheadReader", "seek", reader, i64(0), usize(15))
    rig.it.call_method("Lookahead", "reset_with", la, codec)
    rig.it.call_method("MockCodec", "reset_state", codec)
    ref.reset()
    assert codec.f["abandons"].v == 1 and codec.f["in_flight"] is False
    while True:
        st, p = rig.next_packet(reader)
        if st == "eof":
            break
        assert rig.decode(la, codec, p) == ref.decode(p.f["data"].a[0].v)


def test_pooled_discontinuity_and_corrupt_packets(rig):
    values = list(range(1, 31))
    packets = rig.packets(values)
    packets[13] = rig.it.call("corrupt_packet", u32(1), i64(130))  # inside a batch that is parsed AHEAD
    reader, inner = rig.reader(packets, depth=12)
    codec, la = rig.decoder(4, pooled=True)
    ref = FrameByFrame()
    got = []
    for i in range(len(values)):
        st, p = rig.next_packet(reader)
        assert st == "ok"
        if i in (5, 6):  # the application drops two packets without reset(): both decoders carry their state on
            continue
        r = rig.decode(la, codec, p)
        if i == 13:
            assert r == "DecodeError"  # fails at its own decode_ref, nothing before it was lost
            continue
        assert r == ref.decode(values[i]), i
        got.append(i)
    assert got == [i for i in range(30) if i not in (5, 6, 13)]


def test_pooled_submit_failure_falls_back_to_the_synchronous_transform(rig):
    values = list(range(1, 14))
    reader, inner = rig.reader(rig.packets(values), depth=12)
    codec, la = rig.decoder(4, pooled=True)
    codec.f["fail_submit_in"] = I.Int(1, "i64")  # the second submit fails (a device error at submit time)
    ref = FrameByFrame()
    for v in values:
        st, p = rig.next_packet(reader)
        assert rig.decode(la, codec, p) == ref.decode(v)
    assert codec.f["parses"].v == len(values)  # the batch that could not be submitted was kept and transformed when needed


def test_registry_has_no_fall_through_and_the_shim_delegates(rig):
    it = rig.it
    reg = it.call("CodecRegistry::new")
    aac, mp3 = I.Struct("AudioCodecId", {"0": u32(1)}), I.Struct("AudioCodecId", {"0": u32(2)})
    params = lambda cid: I.Struct("AudioCodecParameters", {"codec": cid})  # noqa: E731
    opts = I.Struct("AudioDecoderOptions", {"gapless": True})
    std, pref = I.Enum("Tier", "Standard"), I.Enum("Tier", "Preferred")
    cpu = it.resolve_value(["cpu_factory"], I.Env(), None)
    hip = it.resolve_value(["hip_factory"], I.Env(), None)
    it.call_method("CodecRegistry", "register_at_tier", reg, std, aac, cpu)
    it.call_method("CodecRegistry", "register_at_tier", reg, std, mp3, cpu)

    def make(cid):
        r = it.call_method("CodecRegistry", "make_audio_decoder", reg, params(cid), opts)
        return r.f["0"].v if r.variant == "Ok" else r.f["0"].variant

    assert make(aac) == 1001
    # the previous revision's register(): a preferred decoder whose factory fails SHADOWS the CPU decoder
    it.call_method("CodecRegistry", "register_at_tier", reg, pref, mp3, hip)
    assert make(mp3) == "Unsupported"
    # this revision: remember what was in force, then register above it; the factory delegates when it cannot build
    it.call("remember", reg, I.Arr([aac], True))
    it.call_method("CodecRegistry", "register_at_tier", reg, pref, aac, hip)
    assert make(aac) == 1001
    # a second register() call must not record the shim's own factory (it would recurse)
    it.call("remember", reg, I.Arr([aac], True))
    assert make(aac) == 1001
    fb = it.call("factory_below", aac)
    assert fb.variant == "Some"
    assert it.call("factory_below", mp3).variant == "None"
    # a codec with NOTHING below: the first lookup (nothing) wins, so a second register() cannot record the shim's own factory
    flac = I.Struct("AudioCodecId", {"0": u32(3)})
    it.call("remember", reg, I.Arr([flac], True))
    it.call_method("CodecRegistry", "register_at_tier", reg, pref, flac, hip)
    it.call("remember", reg, I.Arr([flac], True))
    assert it.call("factory_below", flac).variant == "None"
    assert make(flac) == "Unsupported"  # the reason the accelerated decoder could not be built, not a recursion
