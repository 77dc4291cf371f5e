"""Packet bytes -> PCM for Vorbis I, symphonia-check style (symphonia-check/src/main.rs:289-295), through the WHOLE decoder:

  identification + setup headers and audio packets written by tests/vorbis_writer.py (plain / length-ordered / sparse codebooks,
  VQ lookup types 1 and 2 with and without sequence_p, floor 1 with and without subclasses, residue types 0 / 1 / 2 with several
  passes and a skipped pass, channel coupling, two submaps, unused floors, short and long blocks in every order)
     |
     +--> the REFERENCE: symphonia-codec-vorbis's VorbisDecoder (lib.rs, codebook.rs, floor.rs, residue.rs, dsp.rs, window.rs) on
     |    symphonia-core's own BitReaderRtl, VLC codebook builder, Imdct and in-tree Fft -- all EXECUTED from /root/reference by
     |    tools/rsinterp  ........................................................................................  PCM_ref (f32)
     |
     +--> the same decoder with bindings/rust/patches/symphonia-codec-vorbis.diff applied, default (CPU) backend  .  == PCM_ref, bit for bit
     |
     +--> HipVorbisDecoder (frontends.rs -> vorbis.rs SeamFrontEnd = the patched decoder with the recording backend: header and
          codebook parsing, floor and residue decoding, coupling, floor x residue stay the reference's code; VorbisBatch,
          decoder.rs, lookahead.rs, ctx.rs) with its extern "C" calls bound to libsymaccel (the CPU-emulation build of the
          kernels): symaccel_vorbis_synth does the mixed-block-size IMDCT, the windows and the overlap-add  ......  == PCM_ref, bit for bit

What the writer wrote is compared with what the reference's parser holds after each packet: the floor posts, and every channel's
residue vector (the VQ sums re-done in numpy f32, the inverse coupling applied).  Needs /root/reference (`localref`); the `-m gpu`
twin of the accelerated path is tests/test_rust_adapters.py::test_vorbis_adapter_* (same adapter code, hipcc-built library)."""
import sys
from pathlib import Path

import numpy as np
import pytest

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "tests"))

import vorbis_writer as W  # noqa: E402
from rs_harness import REF, Harness, patched_tree, sized, usize  # noqa: E402
from rsinterp import interp as I  # noqa: E402

pytestmark = pytest.mark.localref

CRATE = "symphonia-codec-vorbis"
# (seed, packets, channels, bs0_exp, bs1_exp, residue types of the short / long mode, channel coupling)
ALL_STREAMS = [(1, 9, 2, 6, 9, (2, 1), True), (2, 9, 1, 6, 8, (0, 2), False), (3, 8, 3, 7, 10, (1, 0), True), (4, 7, 2, 8, 11, (2, 2), True)]
STREAMS = sized(ALL_STREAMS, [(1, 7, 2, 6, 9, (2, 1), True), (3, 6, 3, 7, 9, (1, 0), True)])  # (SYMACCEL_PACKET_TESTS=full: all of them)
BATCH_OF_THE_PLAIN_TESTS = sized(None, 2)  # None: built as the registry builds it (see tests/test_aac_packets.py)


@pytest.fixture(scope="module")
def trees():
    return REF / CRATE / "src", patched_tree((CRATE,)) / CRATE / "src"


def stream(seed, n, nch, bs0, bs1, rtypes, couple):
    s = W.Stream(seed, nch, bs0, bs1, residue_types=rtypes, couple=couple)
    flags = [True, False, False, True, True, False, True, False, False, True, True, False][:n]  # every transition, both ways
    return s, [s.packet(long_block=f) for f in flags]


def cpu_decoder(h, s):
    r = h.it.call("VorbisDecoder::try_new", h.params("CODEC_ID_VORBIS", 44100, s.nch, extra=s.extra_data()), h.opts())
    assert r.variant == "Ok", r
    return r.f["0"]


def bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


def uncouple(m, a):
    """lib.rs:224-246 on two vectors"""
    m, a = m.copy(), a.copy()
    nm = np.where(m > 0, np.where(a > 0, m, m + a), np.where(a > 0, m, m - a)).astype(np.float32)
    na = np.where(m > 0, np.where(a > 0, m - a, m), np.where(a > 0, m + a, m)).astype(np.float32)
    return nm, na


def check_parse(dec, s, rec):
    n2 = (1 << (s.bs1_exp if rec["long"] else s.bs0_exp)) >> 1
    which = int(rec["long"])
    m = s.mappings[which]
    # the floor posts of the last channel that used each floor (the Floor object is shared by a submap's channels)
    for fi in set(fl for fl, _ in m["submaps"]):
        users = [c for c in range(s.nch) if m["submaps"][m["mux"][c]][0] == fi and rec["floor_y"][c] is not None]
        if users:
            floor = I.deref(dec.f["floors"].a[fi])
            assert [y.v for y in floor.f["floor_y"].a] == rec["floor_y"][users[-1]]
    want = [v.copy() for v in rec["residue"]]
    for mag, ang in m["coupling"]:
        want[mag], want[ang] = uncouple(want[mag], want[ang])
    for c in range(s.nch):
        ch = dec.f["dsp"].f["channels"].a[c]
        got = np.array([np.float32(x) for x in ch.f["residue"].a[:n2]], np.float32)
        assert np.array_equal(bits(got), bits(want[c])), (c, rec["long"])
        assert bool(ch.f["do_not_decode"]) == (not rec["decoded"][c])


@pytest.mark.parametrize("seed,n,nch,bs0,bs1,rtypes,couple", STREAMS)
def test_the_reference_decoder_reads_the_packets_as_written_and_its_patched_twin_agrees(trees, seed, n, nch, bs0, bs1, rtypes, couple):
    s, packets = stream(seed, n, nch, bs0, bs1, rtypes, couple)
    outs = []
    for k, tree in enumerate(trees):
        h = Harness(None, reference=True, vorbis_tree=tree)
        dec = cpu_decoder(h, s)
        got = []
        for i, (pk, rec) in enumerate(packets):
            st, planes = h.decode("VorbisDecoder", dec, h.packet(pk, 0))
            assert st == "ok", (i, planes)
            if k == 0:
                check_parse(dec, s, rec)
            got.append(planes)
        outs.append(got)
    assert outs[0][0].shape[1] == 0 and sum(p.shape[1] for p in outs[0]) > 0  # (the first packet only primes the overlap: lib.rs:333-336)
    assert max(float(np.abs(p).max()) for p in outs[0][1:]) > 1e-3
    for a, b in zip(*outs):
        assert a.shape == b.shape and np.array_equal(bits(a), bits(b)), "the seam patch changed what the decoder computes"


def hip_decoder(tree, s, max_batch=None):
    from emu_lib import emu_library
    h = Harness(emu_library().dll, reference=True, vorbis_tree=tree)
    h.it.load_file(ROOT / "tests" / "rust" / "registry_stubs.rs")
    h.load_shim("lib.rs", "ctx.rs", "decoder.rs", "lookahead.rs", "fallback.rs", "vorbis.rs", "frontends.rs")
    p = h.params("CODEC_ID_VORBIS", 44100, s.nch, extra=s.extra_data())
    if max_batch is None:
        r = h.it.call("HipVorbisDecoder::try_registry_new", p, h.opts())
    else:
        front = h.it.call("vorbis_front_end", p, h.opts())
        assert front.variant == "Ok", front
        r = h.it.call("HipVorbisDecoder::try_new", p, h.opts(), front.f["0"], usize(max_batch))
    assert r.variant == "Ok", r
    return h, r.f["0"]


@pytest.mark.parametrize("seed,n,nch,bs0,bs1,rtypes,couple", sized(ALL_STREAMS, [(1, 5, 2, 6, 9, (2, 1), True)]))
def test_the_accelerated_decoder_equals_the_reference_on_packet_bytes(trees, seed, n, nch, bs0, bs1, rtypes, couple):
    s, packets = stream(seed, n, nch, bs0, bs1, rtypes, couple)
    ref = Harness(None, reference=True, vorbis_tree=trees[0])
    ref_dec = cpu_decoder(ref, s)
    h, dec = hip_decoder(trees[1], s, max_batch=BATCH_OF_THE_PLAIN_TESTS)
    for i, (pk, _) in enumerate(packets):
        st_r, want = ref.decode("VorbisDecoder", ref_dec, ref.packet(pk, 0))
        st, got = h.decode("HipVorbisDecoder", dec, h.packet(pk, i))
        assert st == st_r == "ok"
        assert got.shape == want.shape and np.array_equal(bits(got), bits(want)), (i, got.shape, want.shape)
    # no look-ahead reader: batches of one, each through the FUSED entry point (residue vectors + floor posts + coupling steps in,
    # PCM out) -- the reference's inverse coupling, floor synthesis and dot product did not run
    assert h.bridge.calls.count("symaccel_vorbis_decode") == n and h.bridge.calls.count("symaccel_vorbis_synth") == 0
    fused = [a for name, a in h.bridge.scalars if name == "symaccel_vorbis_decode"]
    assert all(a["n_floors"] >= 1 and a["channels_per_stream"] == nch for a in fused), fused


def test_damaged_packets_fail_like_the_reference_and_the_stream_goes_on(trees):
    s, packets = stream(5, sized(8, 6), 2, 6, 8, (1, 2), True)
    data = [p for p, _ in packets]
    data[2] = bytes([data[2][0] | 1]) + data[2][1:]   # the packet type bit: not an audio packet
    data[4] = b""                                      # an empty packet: the bit reader has nothing to give
    ref = Harness(None, reference=True, vorbis_tree=trees[0])
    ref_dec = cpu_decoder(ref, s)
    h, dec = hip_decoder(trees[1], s, max_batch=BATCH_OF_THE_PLAIN_TESTS)
    outcomes = []
    for i, pk in enumerate(data):
        st_r, want = ref.decode("VorbisDecoder", ref_dec, ref.packet(pk, 0))
        st, got = h.decode("HipVorbisDecoder", dec, h.packet(pk, i))
        assert st == st_r, (i, st, st_r, got, want)
        if st == "ok":
            assert got.shape == want.shape and np.array_equal(bits(got), bits(want)), i
        else:
            assert got == want, (i, got, want)
        outcomes.append(st)
    assert outcomes.count("err") == 2 and outcomes[3] == "ok" and outcomes[-1] == "ok"


def test_look_ahead_batches_and_reset(trees):
    n, batch = sized((10, 4), (6, 6))
    s, packets = stream(6, n, 2, 6, 9, (2, 1), True)
    data = [p for p, _ in packets]
    ref = Harness(None, reference=True, vorbis_tree=trees[0])
    ref_dec = cpu_decoder(ref, s)
    want = [ref.decode("VorbisDecoder", ref_dec, ref.packet(pk, 0))[1] for pk in data]
    ref.it.call_method("VorbisDecoder", "reset", ref_dec)
    again = [ref.decode("VorbisDecoder", ref_dec, ref.packet(pk, 0))[1] for pk in data[:3]]
    h, dec = hip_decoder(trees[1], s, max_batch=batch)
    h.it.load_file(ROOT / "tests" / "rust" / "mocks.rs")
    pk = I.Arr([h.packet(d, i, track=1, owned=True) for i, d in enumerate(data)], True)
    reader = h.it.call("LookaheadReader::new", h.it.call("MockReader::new", pk), usize(8))

    def run(first, count):
        out = []
        for i in range(first, first + count):
            r = h.it.call_method("LookaheadReader", "next_packet", reader)
            p = r.f["0"].f["0"]
            assert p.f["pts"].f["0"].v == i
            out.append(h.decode("HipVorbisDecoder", dec, h.it.call_method("Packet", "as_packet_ref", p)))
        return out

    n0 = h.bridge.calls.count("symaccel_vorbis_decode")
    for i, (st, got) in enumerate(run(0, n)):
        assert st == "ok" and got.shape == want[i].shape and np.array_equal(bits(got), bits(want[i])), i
    assert h.bridge.calls.count("symaccel_vorbis_decode") - n0 == -(-n // batch)  # the overlap halves carry across the batches
    assert h.bridge.calls.count("symaccel_vorbis_synth") == 0
    # seek back + reset: the next packet primes the overlap again and gives no samples (dsp.rs:13-19, lib.rs:333-336)
    h.it.call_method("LookaheadReader", "seek", reader, I.Int(0, "i64"), usize(0))
    h.it.call_method("HipVorbisDecoder", "reset", dec)
    for i, ((st, got), want_r) in enumerate(zip(run(0, 3), again)):
        assert st == "ok" and got.shape == want_r.shape and np.array_equal(bits(got), bits(want_r)), i
    assert again[0].shape[1] == 0


def test_decoders_built_by_the_registry_share_the_cross_stream_batcher(trees):
    """What an application gets: `register()` enters HipVorbisDecoder at Tier::Preferred, `make_audio_decoder(params, opts)` builds every decoder
    from (params, opts) alone (codecs/registry.rs:34-44, 252-269, 330-341) -- and the decoders so built find each other in the
    process-wide `Pool`: two streams behind look-ahead readers, decoded alternately, every packet's PCM the reference decoder's bit for
    bit, their batches in common launches (symaccel_batcher_get_stats)."""
    from emu_lib import emu_library
    from rs_harness import pool_stats, registry_round_trip
    n, depth = sized((12, 6), (8, 4))
    # two streams of one shape (block sizes, channels) from different seeds: their own codebooks, floors and packets
    streams = [stream(11, n, 2, 6, 9, (2, 1), True), stream(12, n, 2, 6, 9, (1, 2), True)]
    want = []
    for s, packets in streams:
        ref = Harness(None, reference=True, vorbis_tree=trees[0])
        ref_dec = cpu_decoder(ref, s)
        want.append([ref.decode("VorbisDecoder", ref_dec, ref.packet(pk, 0))[1] for pk, _ in packets])
    h = Harness(emu_library().dll, reference=True, vorbis_tree=trees[1])
    h.it.load_file(ROOT / "tests" / "rust" / "registry_stubs.rs")
    h.it.load_file(ROOT / "tests" / "rust" / "mocks.rs")
    h.load_shim("lib.rs", "ctx.rs", "decoder.rs", "lookahead.rs", "fallback.rs", "vorbis.rs", "frontends.rs")
    decs = registry_round_trip(h, "HipVorbisDecoder", [h.params("CODEC_ID_VORBIS", 44100, s.nch, extra=s.extra_data()) for s, _ in streams])
    readers = []
    for k, (s, packets) in enumerate(streams):
        pk = I.Arr([h.packet(d, i, track=1 + k, owned=True) for i, (d, _) in enumerate(packets)], True)
        readers.append(h.it.call("LookaheadReader::new", h.it.call("MockReader::new", pk), usize(depth)))
    for i in range(n):
        for k in range(2):
            r = h.it.call_method("LookaheadReader", "next_packet", readers[k])
            st, got = h.decode("HipVorbisDecoder", decs[k], h.it.call_method("Packet", "as_packet_ref", r.f["0"].f["0"]))
            assert st == "ok" and got.shape == want[k][i].shape and np.array_equal(bits(got), bits(want[k][i])), (k, i)
    calls = h.bridge.calls
    assert calls.count("symaccel_batcher_create") == 1 and calls.count("symaccel_batcher_reserve") >= 2
    assert calls.count("symaccel_batcher_vorbis_floor") >= 2           # every stream's floor configurations are the batcher's
    assert calls.count("symaccel_vorbis_decode") == 2                   # each stream's cold start only: residue + posts in, PCM out
    kinds = [a["kind"] for name, a in h.bridge.scalars if name == "symaccel_batcher_reserve"]
    assert kinds and all(kd == 6 for kd in kinds), kinds               # SYMACCEL_BATCH_VORBIS_DECODE: coupling, floors and synthesis on the device
    stats = pool_stats(h)
    assert stats["submissions"] >= 2 and stats["launches"] < stats["submissions"] and stats["failed_tickets"] == 0, stats
