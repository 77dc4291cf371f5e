"""Packet bytes -> PCM for AAC-LC, symphonia-check style (symphonia-check/src/main.rs:289-295) -- the codec of the headline
benchmark, through the WHOLE decoder:

  raw_data_blocks written by tests/aac_writer.py (SCE / CPE with and without a common window, all four window sequences and both
  shapes, grouped short windows, every spectral codebook incl. escapes, mid/side masks, intensity stereo, PNS, pulse data, TNS,
  fill / data-stream elements)
     |
     +--> the REFERENCE: symphonia-codec-aac's AacDecoder (aac/mod.rs, cpe.rs, ics/*.rs, codebooks.rs, dsp.rs, window.rs) on
     |    symphonia-core's own BitReaderLtr, VLC codebook builder, Imdct and in-tree Fft -- all EXECUTED from /root/reference by
     |    tools/rsinterp  .......................................................................................  PCM_ref (f32)
     |
     +--> the same decoder with bindings/rust/patches/symphonia-codec-aac.diff applied, default (CPU) backend  ...  == PCM_ref, bit for bit
     |
     +--> HipAacDecoder (frontends.rs -> aac.rs SeamFrontEnd = the patched decoder with the recording backend: section data,
          scale factors, Huffman decoding, pulse, TNS, PNS, joint stereo stay the reference's code; AacBatch, decoder.rs,
          lookahead.rs, ctx.rs) with its extern "C" calls bound to libsymaccel (the CPU-emulation build of the kernels):
          symaccel_aac_synth does the IMDCT, the windowing and the overlap-add  .................................  == PCM_ref, bit for bit

The writer's own bookkeeping (window sequence, shape, max_sfb, the codebook of every band) is compared with what the reference's
parser reads back, so a packet that decodes "ok" was also understood as written.  Needs /root/reference (`localref`); the `-m gpu`
twin of the accelerated path is tests/test_rust_adapters.py::test_aac_adapter_* (same adapter code, hipcc-built library)."""
import sys
from pathlib import Path

import numpy as np
import pytest

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "tests"))

import aac_writer as W  # noqa: E402
from rs_harness import REF, Harness, patched_tree, sized, usize  # noqa: E402
from rsinterp import interp as I  # noqa: E402

pytestmark = pytest.mark.localref

CRATE = "symphonia-codec-aac"
LAYOUTS = {1: ["sce"], 2: ["cpe"]}

# None: the decoder is built as the registry builds it (try_registry_new: the default batch, whose pinned buffers the interpreter
# takes 20 s to zero); the default set builds it through try_new with a small batch instead
BATCH_OF_THE_PLAIN_TESTS = sized(None, 2)


@pytest.fixture(scope="module")
def trees():
    return REF / CRATE / "src", patched_tree((CRATE,)) / CRATE / "src"


def stream(seed, n_packets, nch):
    s = W.Stream(REF, seed, LAYOUTS[nch])
    return [s.packet(filler=(t % 4 == 2)) for t in range(n_packets)]


def cpu_decoder(h, nch):
    r = h.it.call("AacDecoder::try_new", h.params("CODEC_ID_AAC", 44100, nch), h.opts())
    assert r.variant == "Ok", r
    return h.f32_buffers(r.f["0"])


def parsed_state(dec, nch):
    """what the reference's parser holds after a packet, per channel (cpe.rs:24-32, ics/mod.rs:88-100, 194-209)"""
    pair = dec.f["pairs"].a[0]
    out = []
    for c in range(nch):
        ics = pair.f["ics%d" % c]
        info = ics.f["info"]
        groups, max_sfb = info.f["window_groups"].v, info.f["max_sfb"].v
        out.append({"seq": info.f["window_sequence"].v, "shape": int(bool(info.f["window_shape"])), "max_sfb": max_sfb,
                    "sfb_cb": [[ics.f["sfb_cb"].a[g].a[s].v for s in range(max_sfb)] for g in range(groups)]})
    return out


def bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


ALL_STREAMS = [(1, 7, 1), (2, 8, 2), (3, 6, 2)]
STREAMS = sized(ALL_STREAMS, [(2, 5, 2)])  # (SYMACCEL_PACKET_TESTS=full: all of them)


@pytest.mark.parametrize("seed,n_packets,nch", STREAMS)
def test_the_reference_decoder_reads_the_packets_as_written_and_its_patched_twin_agrees(trees, seed, n_packets, nch):
    packets = stream(seed, n_packets, nch)
    outs = []
    for tree in trees:
        h = Harness(None, reference=True, aac_tree=tree)
        dec = cpu_decoder(h, nch)
        got = []
        for i, (pk, meta) in enumerate(packets):
            st, planes = h.decode("AacDecoder", dec, h.packet(pk, i * 1024))
            assert st == "ok", (i, planes)
            assert planes.shape == (nch, 1024) and np.isfinite(planes).all()
            assert parsed_state(dec, nch) == meta, i
            got.append(planes)
        outs.append(np.stack(got))
    assert np.abs(outs[0]).max() > 1.0  # (not a stream of silence)
    assert np.array_equal(bits(outs[0]), bits(outs[1])), "the seam patch changed what the decoder computes"


def test_the_streams_cover_the_syntax():
    from collections import Counter
    books, sequences, shapes = Counter(), Counter(), Counter()
    for seed, n_packets, nch in ALL_STREAMS:  # (writing is cheap: always the full set)
        for _, meta in stream(seed, n_packets, nch):
            for m in meta:
                sequences[m["seq"]] += 1
                shapes[m["shape"]] += 1
                for row in m["sfb_cb"]:
                    books.update(row)
    assert set(books) == set(range(0, 12)) | {13, 14, 15}  # zero, every spectral codebook, noise, both intensity directions
    assert set(sequences) == {0, 1, 2, 3} and set(shapes) == {0, 1}


def hip_decoder(tree, nch, max_batch=None):
    from emu_lib import emu_library
    h = Harness(emu_library().dll, reference=True, aac_tree=tree)
    h.it.load_file(ROOT / "tests" / "rust" / "registry_stubs.rs")
    h.load_shim("lib.rs", "ctx.rs", "decoder.rs", "lookahead.rs", "fallback.rs", "aac.rs", "frontends.rs")
    p = h.params("CODEC_ID_AAC", 44100, nch)
    if max_batch is None:
        r = h.it.call("HipAacDecoder::try_registry_new", p, h.opts())  # what the registry calls (registry.rs:34-44)
    else:
        front = h.it.call("aac_front_end", p, h.opts())
        assert front.variant == "Ok", front
        r = h.it.call("HipAacDecoder::try_new", p, h.opts(), front.f["0"], usize(max_batch))
    assert r.variant == "Ok", r
    return h, h.f32_buffers(r.f["0"])


# (the quick set reaches packet 6 of stream 2: the first one whose TNS filters cover a non-empty range of lines)
@pytest.mark.parametrize("seed,n_packets,nch", sized(ALL_STREAMS[:2], [(2, 7, 2)]))
def test_the_accelerated_decoder_equals_the_reference_on_packet_bytes(trees, seed, n_packets, nch):
    packets = stream(seed, n_packets, nch)
    ref = Harness(None, reference=True, aac_tree=trees[0])
    ref_dec = cpu_decoder(ref, nch)
    h, dec = hip_decoder(trees[1], nch, max_batch=BATCH_OF_THE_PLAIN_TESTS)
    for i, (pk, _) in enumerate(packets):
        st_r, want = ref.decode("AacDecoder", ref_dec, ref.packet(pk, i * 1024))
        st, got = h.decode("HipAacDecoder", dec, h.packet(pk, i * 1024))
        assert st == st_r == "ok"
        assert np.array_equal(bits(got), bits(want)), (i, np.abs(got - want).max())
    # no look-ahead reader: batches of one, each through the FUSED entry point (coefficients as decoded + joint-stereo descriptors +
    # TNS filters in, PCM out) -- the reference's M/S, intensity and TNS code did not run
    assert h.bridge.calls.count("symaccel_aac_decode_pipelined") == n_packets and h.bridge.calls.count("symaccel_aac_synth") == 0
    fused = [a for name, a in h.bridge.scalars if name == "symaccel_aac_decode_pipelined"]
    if nch == 2:  # the writer's streams use M/S, intensity stereo and TNS: they reached the device as descriptors, not as arithmetic
        assert any(a["n_pairs"] == 1 for a in fused) and any(a["n_tns"] > 0 for a in fused), fused
    cp = I.deref(h.it.call_method("HipAacDecoder", "codec_params", dec))
    assert cp.f["sample_rate"].f["0"].v == 44100 and cp.f["channels"].f["0"].f["0"].v == nch


def test_damaged_packets_fail_like_the_reference_and_the_stream_goes_on(trees):
    nch = 2
    packets = [p for p, _ in stream(5, sized(7, 5), nch)]
    bad = {1: bytearray(packets[1]), 3: bytearray(packets[3])}
    bad[1] = bad[1][: max(6, len(bad[1]) // 3)]  # truncated: the bit reader runs dry inside the element
    bad[3][0] = (2 << 5) | (bad[3][0] & 0x1F)     # a coupling channel element: unsupported
    data = [bytes(bad.get(i, p)) for i, p in enumerate(packets)]
    ref = Harness(None, reference=True, aac_tree=trees[0])
    ref_dec = cpu_decoder(ref, nch)
    h, dec = hip_decoder(trees[1], nch, max_batch=BATCH_OF_THE_PLAIN_TESTS)
    outcomes = []
    for i, pk in enumerate(data):
        st_r, want = ref.decode("AacDecoder", ref_dec, ref.packet(pk, i * 1024))
        st, got = h.decode("HipAacDecoder", dec, h.packet(pk, i * 1024))
        assert st == st_r, (i, st, st_r, got, want)
        if st == "ok":
            # after a lost packet both decoders overlap-add onto the delay line the last GOOD packet left (the failed packet never
            # reached the synthesis on either side)
            assert np.array_equal(bits(got), bits(want)), i
        else:
            assert got == want, (i, got, want)
        outcomes.append(st)
    assert outcomes.count("err") == 2 and outcomes[2] == "ok" and outcomes[-1] == "ok"


def test_look_ahead_batches_and_reset(trees):
    nch = 2
    n, batch = sized((9, 4), (4, 4))
    packets = [p for p, _ in stream(6, n, nch)]
    ref = Harness(None, reference=True, aac_tree=trees[0])
    ref_dec = cpu_decoder(ref, nch)
    want = [ref.decode("AacDecoder", ref_dec, ref.packet(pk, i * 1024))[1] for i, pk in enumerate(packets)]
    # ... and what the reference gives after the same seek + reset: its noise generator is NOT reset (cpe.rs:49-52 resets the two
    # channel streams, not `lcg`), so the noise-substituted bands of packet 0 differ from the first pass
    ref.it.call_method("AacDecoder", "reset", ref_dec)
    again = [ref.decode("AacDecoder", ref_dec, ref.packet(pk, i * 1024))[1] for i, pk in enumerate(packets[:3])]
    assert not np.array_equal(bits(again[0]), bits(want[0]))
    h, dec = hip_decoder(trees[1], nch, max_batch=batch)
    h.it.load_file(ROOT / "tests" / "rust" / "mocks.rs")
    pk = I.Arr([h.packet(d, i * 1024, track=1, owned=True) for i, d in enumerate(packets)], True)
    reader = h.it.call("LookaheadReader::new", h.it.call("MockReader::new", pk), usize(8))

    def run(first, count):
        out = []
        for i in range(first, first + count):
            r = h.it.call_method("LookaheadReader", "next_packet", reader)
            p = r.f["0"].f["0"]
            assert p.f["pts"].f["0"].v == i * 1024
            out.append(h.decode("HipAacDecoder", dec, h.it.call_method("Packet", "as_packet_ref", p)))
        return out

    n0 = h.bridge.calls.count("symaccel_aac_decode_pipelined")
    for i, (st, got) in enumerate(run(0, n)):
        assert st == "ok" and np.array_equal(bits(got), bits(want[i])), i
    assert h.bridge.calls.count("symaccel_aac_decode_pipelined") - n0 == -(-n // batch)  # the delay lines carry across the batches
    assert h.bridge.calls.count("symaccel_aac_synth") == 0
    # seek back to the start and reset both: the overlap state is cleared (aac/mod.rs:244-248, ics/mod.rs:223-226)
    h.it.call_method("LookaheadReader", "seek", reader, I.Int(0, "i64"), usize(0))
    h.it.call_method("HipAacDecoder", "reset", dec)
    for i, (st, got) in zip(range(0, 3), run(0, 3)):
        assert st == "ok" and np.array_equal(bits(got), bits(again[i])), i


def test_two_pooled_decoders_share_the_cross_stream_batcher(trees):
    """`HipAacDecoder::try_new_pooled`: two AAC-LC streams (joint stereo, intensity, TNS, all window sequences) behind look-ahead readers,
    decoded alternately.  The batches after each stream's first go through symaccel_batcher_submit_aac_decode / _collect of ONE
    process-wide pool -- joint stereo, TNS and synthesis of both streams in the same launches -- and every packet's PCM equals the
    reference decoder's bit for bit."""
    nch = 2
    n, batch = sized((10, 4), (6, 2))
    streams = [[p for p, _ in stream(6, n, nch)], [p for p, _ in stream(2, n, nch)]]
    want = []
    for packets in streams:
        ref = Harness(None, reference=True, aac_tree=trees[0])
        ref_dec = cpu_decoder(ref, nch)
        want.append([ref.decode("AacDecoder", ref_dec, ref.packet(pk, i * 1024))[1] for i, pk in enumerate(packets)])
    from emu_lib import emu_library
    h = Harness(emu_library().dll, reference=True, aac_tree=trees[1])
    h.it.load_file(ROOT / "tests" / "rust" / "registry_stubs.rs")
    h.load_shim("lib.rs", "ctx.rs", "decoder.rs", "lookahead.rs", "fallback.rs", "aac.rs", "frontends.rs")
    h.it.load_file(ROOT / "tests" / "rust" / "mocks.rs")
    decs, readers = [], []
    for k, packets in enumerate(streams):
        p = h.params("CODEC_ID_AAC", 44100, nch)
        front = h.it.call("aac_front_end", p, h.opts())
        assert front.variant == "Ok", front
        r = h.it.call("HipAacDecoder::try_new_pooled", p, h.opts(), front.f["0"], usize(batch))
        assert r.variant == "Ok", r
        decs.append(h.f32_buffers(r.f["0"]))
        pk = I.Arr([h.packet(d, i * 1024, track=1 + k, owned=True) for i, d in enumerate(packets)], True)
        readers.append(h.it.call("LookaheadReader::new", h.it.call("MockReader::new", pk), usize(3 * batch)))
    for i in range(n):
        for k in range(2):
            r = h.it.call_method("LookaheadReader", "next_packet", readers[k])
            st, got = h.decode("HipAacDecoder", decs[k], h.it.call_method("Packet", "as_packet_ref", r.f["0"].f["0"]))
            assert st == "ok" and np.array_equal(bits(got), bits(want[k][i])), (k, i)
    calls = h.bridge.calls
    assert calls.count("symaccel_batcher_create") == 1
    assert calls.count("symaccel_batcher_submit_aac_decode") >= 2 * ((n - batch) // batch)
    assert calls.count("symaccel_aac_decode_pipelined") == 2  # each stream's cold start only


def test_seek_into_the_middle_of_the_stream_and_reset(trees):
    """reset() after a seek must reach the reference decoder inside the front end (`AacFrontEnd::reset`): its pairs remember the
    window shape of the frame parsed last (ics/mod.rs:229-232 forgets it) -- and the front end parsed AHEAD of what the caller had been
    given when the seek came."""
    nch, n, batch, k = 2, sized(8, 5), 4, 2
    written = stream(10, n, nch)
    packets = [p for p, _ in written]
    # the last packet the look-ahead parses before the seek carries KBD windows: a front end that is not reset hands packet k on with
    # prev_window_shape = KBD where the reference, reset, says sine
    assert all(m["shape"] == 1 for m in written[batch - 1][1])
    ref = Harness(None, reference=True, aac_tree=trees[0])
    ref_dec = cpu_decoder(ref, nch)
    for i, pk in enumerate(packets[:batch]):  # the reference plays on past the target, like the accelerated decoder's look-ahead does
        ref.decode("AacDecoder", ref_dec, ref.packet(pk, i * 1024))
    ref.it.call_method("AacDecoder", "reset", ref_dec)
    want = [ref.decode("AacDecoder", ref_dec, ref.packet(pk, (k + i) * 1024)) for i, pk in enumerate(packets[k:])]
    h, dec = hip_decoder(trees[1], nch, max_batch=batch)
    h.it.load_file(ROOT / "tests" / "rust" / "mocks.rs")
    pk = I.Arr([h.packet(d, i * 1024, track=1, owned=True) for i, d in enumerate(packets)], True)
    reader = h.it.call("LookaheadReader::new", h.it.call("MockReader::new", pk), usize(8))

    def run(first, count):
        out = []
        for i in range(first, first + count):
            p = h.it.call_method("LookaheadReader", "next_packet", reader).f["0"].f["0"]
            assert p.f["pts"].f["0"].v == i * 1024
            out.append(h.decode("HipAacDecoder", dec, h.it.call_method("Packet", "as_packet_ref", p)))
        return out

    assert all(st == "ok" for st, _ in run(0, 1))  # (the first batch has parsed packets 0 .. batch - 1 by now)
    h.it.call_method("LookaheadReader", "seek", reader, I.Int(0, "i64"), usize(k))
    h.it.call_method("HipAacDecoder", "reset", dec)
    got = run(k, n - k)
    assert [st for st, _ in got] == [st for st, _ in want]
    for i, ((st, a), (_, b)) in enumerate(zip(got, want)):
        assert (np.array_equal(bits(a), bits(b)) if st == "ok" else a == b), (k + i, st)


def test_decoders_built_by_the_registry_share_the_cross_stream_batcher(trees):
    """What an application gets: `register()` enters HipAacDecoder at Tier::Preferred, `make_audio_decoder(params, opts)` builds every decoder
    from (params, opts) alone (codecs/registry.rs:34-44, 252-269, 330-341) -- and the decoders so built find each other in the
    process-wide `Pool`: two streams behind look-ahead readers, decoded alternately, every packet's PCM the reference decoder's bit for
    bit, their batches in common launches (symaccel_batcher_get_stats)."""
    from emu_lib import emu_library
    from rs_harness import pool_stats, registry_round_trip
    nch = 2
    n, depth = sized((10, 6), (6, 4))
    streams = [[p for p, _ in stream(6, n, nch)], [p for p, _ in stream(2, n, nch)]]
    want = []
    for packets in streams:
        ref = Harness(None, reference=True, aac_tree=trees[0])
        ref_dec = cpu_decoder(ref, nch)
        want.append([ref.decode("AacDecoder", ref_dec, ref.packet(pk, i * 1024))[1] for i, pk in enumerate(packets)])
    h = Harness(emu_library().dll, reference=True, aac_tree=trees[1])
    h.it.load_file(ROOT / "tests" / "rust" / "registry_stubs.rs")
    h.load_shim("lib.rs", "ctx.rs", "decoder.rs", "lookahead.rs", "fallback.rs", "aac.rs", "frontends.rs")
    h.it.load_file(ROOT / "tests" / "rust" / "mocks.rs")
    p = h.params("CODEC_ID_AAC", 44100, nch)
    decs = [h.f32_buffers(d) for d in registry_round_trip(h, "HipAacDecoder", [p, p])]
    readers = []
    for k, packets in enumerate(streams):
        pk = I.Arr([h.packet(d, i * 1024, track=1 + k, owned=True) for i, d in enumerate(packets)], True)
        readers.append(h.it.call("LookaheadReader::new", h.it.call("MockReader::new", pk), usize(depth)))
    for i in range(n):
        for k in range(2):
            r = h.it.call_method("LookaheadReader", "next_packet", readers[k])
            st, got = h.decode("HipAacDecoder", decs[k], h.it.call_method("Packet", "as_packet_ref", r.f["0"].f["0"]))
            assert st == "ok" and np.array_equal(bits(got), bits(want[k][i])), (k, i)
    calls = h.bridge.calls
    assert calls.count("symaccel_batcher_create") == 1 and calls.count("symaccel_batcher_submit_aac_decode") >= 2
    assert calls.count("symaccel_aac_decode_pipelined") == 2  # each stream's cold start only
    stats = pool_stats(h)
    assert stats["submissions"] >= 2 and stats["launches"] < stats["submissions"] and stats["failed_tickets"] == 0, stats
