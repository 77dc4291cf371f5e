"""Packet bytes -> PCM for MPEG-1 / MPEG-2 Layer III, symphonia-check style (symphonia-check/src/main.rs:289-295), through the WHOLE
decoder:

  frames written by tests/mp3_writer.py (every Huffman table with and without linbits, both count1 tables, all block types incl.
  mixed blocks, scfsi, MPEG-1 and MPEG-2 scale factors, mono / stereo / dual channel / joint stereo with mid-side and intensity,
  variable bit rate, main data reaching up to 511 bytes back into earlier frames)
     |
     +--> the REFERENCE: symphonia-bundle-mp3's MpaDecoder (decoder.rs, header.rs, layer3/*.rs, synthesis.rs; built with
     |    `features = ["mp3"]`) on symphonia-core's own BufReader, BitReaderLtr and VLC codebook builder -- all EXECUTED from
     |    /root/reference by tools/rsinterp  ........................................................................  PCM_ref (f32)
     |
     +--> the same decoder with bindings/rust/patches/symphonia-bundle-mp3.diff applied, default (CPU) backend  ......  == PCM_ref, bit for bit
     |
     +--> HipMpaDecoder (frontends.rs -> mpa.rs SeamFrontEnd = the patched decoder with the recording backend: header, side
          information, bit reservoir, scale factors, Huffman decoding, requantisation and joint stereo stay the reference's code;
          MpaBatch, decoder.rs, lookahead.rs, ctx.rs) with its extern "C" calls bound to libsymaccel (the CPU-emulation build of
          the kernels): symaccel_mp3_synth does reorder, alias reduction, IMDCT, overlap, frequency inversion and the polyphase
          filterbank  .................................................................................................  == PCM_ref, bit for bit

What the writer wrote is compared with what the reference's parser hands on: `rzero` of every granule, and -- where no joint stereo
mixes the channels -- the positions of the non-zero samples.  Needs /root/reference (`localref`); the `-m gpu` twin of the accelerated
path is tests/test_rust_adapters.py::test_mpa_adapter_* (same adapter code, hipcc-built library, scripted front end)."""
import sys
from pathlib import Path

import numpy as np
import pytest

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "tests"))

import mp3_writer as W  # noqa: E402
from rs_harness import REF, Harness, patched_tree, sized, usize  # noqa: E402
from rsinterp import interp as I  # noqa: E402

pytestmark = pytest.mark.localref

CRATE = "symphonia-bundle-mp3"
# (seed, frames, channel mode, MPEG-1?, sample rate code)
ALL_STREAMS = [(1, 6, "stereo", True, 0), (2, 6, "joint", True, 1), (3, 5, "mono", True, 2), (4, 6, "joint", False, 0), (5, 5, "dual", False, 1)]
STREAMS = sized(ALL_STREAMS, [(2, 3, "joint", True, 1), (4, 4, "joint", False, 0)])  # (SYMACCEL_PACKET_TESTS=full: all of them)

# None: the decoder is built as the registry builds it (try_registry_new: the default batch, whose pinned buffers the interpreter
# takes 20 s to zero); the default set builds it through try_new with a small batch instead
BATCH_OF_THE_PLAIN_TESTS = sized(None, 2)


@pytest.fixture(scope="module")
def trees():
    return REF / CRATE / "src", patched_tree((CRATE,)) / CRATE / "src"


def stream(seed, n, mode, mpeg1, sr_code):
    s = W.Stream(REF, seed, mode, mpeg1=mpeg1, sr_code=sr_code)
    return s, s.packets(n)


def cpu_decoder(h, s):
    r = h.it.call("MpaDecoder::try_new", h.params("CODEC_ID_MP3", s.rate, s.nch), h.opts())
    assert r.variant == "Ok", r
    return r.f["0"]


def bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


@pytest.mark.parametrize("seed,n,mode,mpeg1,sr_code", STREAMS)
def test_the_reference_decoder_and_its_patched_twin_decode_the_frames(trees, seed, n, mode, mpeg1, sr_code):
    s, packets = stream(seed, n, mode, mpeg1, sr_code)
    per_frame = 1152 if mpeg1 else 576
    outs = []
    for tree in trees:
        h = Harness(None, reference=True, mp3_tree=tree)
        dec = cpu_decoder(h, s)
        got = []
        for i, (pk, _) in enumerate(packets):
            st, planes = h.decode("MpaDecoder", dec, h.packet(pk, i * per_frame))
            assert st == "ok", (i, planes)
            assert planes.shape == (s.nch, per_frame) and np.isfinite(planes).all()
            got.append(planes)
        outs.append(np.stack(got))
    assert np.abs(outs[0]).max() > 1e-3  # (not a stream of silence)
    assert np.array_equal(bits(outs[0]), bits(outs[1])), "the seam patch changed what the decoder computes"


def shim(tree):
    from emu_lib import emu_library
    h = Harness(emu_library().dll, reference=True, mp3_tree=tree)
    h.it.load_file(ROOT / "tests" / "rust" / "registry_stubs.rs")
    h.load_shim("lib.rs", "ctx.rs", "decoder.rs", "lookahead.rs", "fallback.rs", "mpa.rs", "frontends.rs")
    return h


def hip_decoder(tree, s, max_batch=None):
    h = shim(tree)
    p = h.params("CODEC_ID_MP3", s.rate, s.nch)
    if max_batch is None:
        r = h.it.call("HipMpaDecoder::try_registry_new", p, h.opts())  # what the registry calls (registry.rs:34-44)
    else:
        front = h.it.call("mpa_front_end", p, h.opts())
        assert front.variant == "Ok", front
        r = h.it.call("HipMpaDecoder::try_new", p, h.opts(), front.f["0"], usize(max_batch))
    assert r.variant == "Ok", r
    return h, r.f["0"]


@pytest.mark.parametrize("seed,n,mode,mpeg1,sr_code", sized(ALL_STREAMS[:2] + ALL_STREAMS[3:4], [(1, 3, "stereo", True, 0)]))
def test_the_parser_hands_on_what_was_written(trees, seed, n, mode, mpeg1, sr_code):
    """the shim's front end (the patched decoder with the recording backend) on the writer's frames: side information and spectra"""
    s, packets = stream(seed, n, mode, mpeg1, sr_code)
    h = shim(trees[1])
    front = h.it.call("mpa_front_end", h.params("CODEC_ID_MP3", s.rate, s.nch), h.opts())
    assert front.variant == "Ok", front
    front = front.f["0"]
    ngr = 2 if mpeg1 else 1
    for i, (pk, rec) in enumerate(packets):
        r = h.it.call_method("SeamFrontEnd", "parse", front, h.packet(pk, 0))
        assert r.variant == "Ok", (i, r)
        parsed = r.f["0"]
        assert parsed.f["n_granules"].v == ngr
        # the FUSED front end (the default): what the entropy decoder produced, as integers, in front of requantize and stereo --
        # exactly the samples the writer coded, whatever the stereo mode
        fused = parsed.f["fused"]
        assert fused.variant == "Some" and len(parsed.f["xr"].a) == 0
        fused = fused.f["0"]
        quant = np.array([x.v for x in fused.f["quant"].a], np.int64).reshape(ngr * s.nch, 576)
        assert len(fused.f["rq"].a) == ngr * s.nch and len(fused.f["st"].a) == ngr
        for k, ((f, q), side, rq) in enumerate(zip(rec["granules"], parsed.f["side"].a, fused.f["rq"].a)):
            plain = mode != "joint" or rec["mode_ext"] == 0
            # (joint stereo processing widens the non-zero range of a channel to its partner's: stereo.rs:549-553)
            assert (side.f["rzero"].v == f["rzero"]) if plain else (side.f["rzero"].v >= f["rzero"]), (i, k)
            assert rq.f["rzero"].v == f["rzero"] and rq.f["global_gain"].v == f["global_gain"], (i, k)
            assert side.f["block_type"].v == f["block"] and bool(side.f["is_mixed"].v) == bool(f["mixed"] and f["block"] == W.SHORT), (i, k)
            assert np.array_equal(quant[k], np.asarray(q, np.int64)), (i, k)
        for g, st in enumerate(fused.f["st"].a):
            want_flags = ((rec["mode_ext"] >> 1) & 1) | ((rec["mode_ext"] & 1) << 1) if mode == "joint" else 0
            assert (st.f["flags"].v & 3) == want_flags, (i, g)


def test_the_unfused_front_end_hands_on_requantized_spectra(trees):
    """the first-generation seam (behind requantize + stereo) stays available: SeamFrontEnd::try_new_at(.., false)"""
    seed, n, mode, mpeg1, sr_code = 1, 2, "stereo", True, 0
    s, packets = stream(seed, n, mode, mpeg1, sr_code)
    h = shim(trees[1])
    front = h.it.call("SeamFrontEnd::try_new_at", h.params("CODEC_ID_MP3", s.rate, s.nch), h.opts(), False)
    assert front.variant == "Ok", front
    front = front.f["0"]
    for i, (pk, rec) in enumerate(packets):
        r = h.it.call_method("SeamFrontEnd", "parse", front, h.packet(pk, 0))
        assert r.variant == "Ok", (i, r)
        parsed = r.f["0"]
        assert parsed.f["fused"].variant == "None"
        xr = np.array([np.float32(x) for x in parsed.f["xr"].a], np.float32).reshape(2 * s.nch, 576)
        for k, ((f, q), side) in enumerate(zip(rec["granules"], parsed.f["side"].a)):
            assert side.f["rzero"].v == f["rzero"], (i, k)
            assert np.array_equal(xr[k] != 0, np.asarray(q) != 0), (i, k)


# (a call into the CPU-emulation library costs seconds whatever the batch: the default set keeps the number of calls down)
@pytest.mark.parametrize("seed,n,mode,mpeg1,sr_code", sized(ALL_STREAMS, [(4, 3, "joint", False, 0)]))
def test_the_accelerated_decoder_equals_the_reference_on_packet_bytes(trees, seed, n, mode, mpeg1, sr_code):
    s, packets = stream(seed, n, mode, mpeg1, sr_code)
    per_frame = 1152 if mpeg1 else 576
    ref = Harness(None, reference=True, mp3_tree=trees[0])
    ref_dec = cpu_decoder(ref, s)
    h, dec = hip_decoder(trees[1], s, max_batch=BATCH_OF_THE_PLAIN_TESTS)
    for i, (pk, _) in enumerate(packets):
        st_r, want = ref.decode("MpaDecoder", ref_dec, ref.packet(pk, i * per_frame))
        st, got = h.decode("HipMpaDecoder", dec, h.packet(pk, i * per_frame))
        assert st == st_r == "ok"
        assert np.array_equal(bits(got), bits(want)), (i, float(np.abs(got - want).max()))
    # no look-ahead reader: batches of one, each through the FUSED entry point (int16 Huffman samples + records in, PCM out)
    assert h.bridge.calls.count("symaccel_mp3_decode_pipelined") == n and h.bridge.calls.count("symaccel_mp3_synth") == 0


def test_damaged_packets_fail_like_the_reference_and_the_stream_goes_on(trees):
    mpeg1 = sized(True, False)
    per_frame = 1152 if mpeg1 else 576
    s, packets = stream(7, sized(7, 5), "stereo", mpeg1, 0)
    data = [p for p, _ in packets]
    data[1] = data[1][:-5]                                   # the packet is shorter than the header says
    bad = bytearray(data[3])
    bad[2] = (bad[2] & 0x0F) | 0xF0                          # an invalid bit-rate index: the header does not check out
    data[3] = bytes(bad)
    ref = Harness(None, reference=True, mp3_tree=trees[0])
    ref_dec = cpu_decoder(ref, s)
    h, dec = hip_decoder(trees[1], s, max_batch=BATCH_OF_THE_PLAIN_TESTS)
    outcomes = []
    for i, pk in enumerate(data):
        st_r, want = ref.decode("MpaDecoder", ref_dec, ref.packet(pk, i * per_frame))
        st, got = h.decode("HipMpaDecoder", dec, h.packet(pk, i * per_frame))
        assert st == st_r, (i, st, st_r, got, want)
        if st == "ok":
            assert np.array_equal(bits(got), bits(want)), i
        else:
            assert got == want, (i, got, want)
        outcomes.append(st)
    assert outcomes.count("err") >= 2 and outcomes[-1] == "ok"


def test_look_ahead_batches_and_reset(trees):
    mpeg1, n, batch = sized((True, 9, 4), (True, 4, 4))
    per_frame = 1152 if mpeg1 else 576
    s, packets = stream(8, n, "joint", mpeg1, 0)
    data = [p for p, _ in packets]
    ref = Harness(None, reference=True, mp3_tree=trees[0])
    ref_dec = cpu_decoder(ref, s)
    want = [ref.decode("MpaDecoder", ref_dec, ref.packet(pk, i * per_frame))[1] for i, pk in enumerate(data)]
    ref.it.call_method("MpaDecoder", "reset", ref_dec)
    again = [ref.decode("MpaDecoder", ref_dec, ref.packet(pk, i * per_frame)) for i, pk in enumerate(data[:3])]
    h, dec = hip_decoder(trees[1], s, max_batch=batch)
    h.it.load_file(ROOT / "tests" / "rust" / "mocks.rs")
    pk = I.Arr([h.packet(d, i * per_frame, track=1, owned=True) for i, d in enumerate(data)], True)
    reader = h.it.call("LookaheadReader::new", h.it.call("MockReader::new", pk), usize(8))

    def run(first, count):
        out = []
        for i in range(first, first + count):
            r = h.it.call_method("LookaheadReader", "next_packet", reader)
            p = r.f["0"].f["0"]
            assert p.f["pts"].f["0"].v == i * per_frame
            out.append(h.decode("HipMpaDecoder", dec, h.it.call_method("Packet", "as_packet_ref", p)))
        return out

    n0 = h.bridge.calls.count("symaccel_mp3_decode_pipelined")
    for i, (st, got) in enumerate(run(0, n)):
        assert st == "ok" and np.array_equal(bits(got), bits(want[i])), i
    assert h.bridge.calls.count("symaccel_mp3_decode_pipelined") - n0 == -(-n // batch)  # overlap and the V FIFO carry across the batches
    assert h.bridge.calls.count("symaccel_mp3_synth") == 0
    # seek back to the start + reset: the reference drops overlap, FIFO and the bit reservoir (decoder.rs:149-152); packet 0 does not
    # reach back, packets 1 and 2 do -- into packet 0's slot, which both decoders have again
    h.it.call_method("LookaheadReader", "seek", reader, I.Int(0, "i64"), usize(0))
    h.it.call_method("HipMpaDecoder", "reset", dec)
    for i, ((st, got), (st_r, want_r)) in enumerate(zip(run(0, 3), again)):  # (one batch: the reader is three or more packets ahead)
        assert st == st_r == "ok" and np.array_equal(bits(got), bits(want_r)), i


def test_two_pooled_decoders_share_the_cross_stream_batcher(trees):
    """`HipMpaDecoder::try_new_pooled`: two streams behind look-ahead readers, decoded alternately like a server's worker would.  The
    batches after each stream's first go through `Pool::reserve` / `commit` / `wait` / `release` of the process-wide `Pool` (one
    batcher for both decoders; the Huffman samples are written straight into its page-locked slots), and every packet's PCM still equals the reference decoder's bit for bit."""
    mpeg1, n, batch = sized((True, 10, 4), (True, 7, 2))
    per_frame = 1152 if mpeg1 else 576
    streams = [stream(8, n, "joint", mpeg1, 0), stream(9, n, "stereo", mpeg1, 0)]
    want = []
    for s, packets in streams:
        ref = Harness(None, reference=True, mp3_tree=trees[0])
        ref_dec = cpu_decoder(ref, s)
        want.append([ref.decode("MpaDecoder", ref_dec, ref.packet(pk, i * per_frame))[1] for i, (pk, _) in enumerate(packets)])
    h = shim(trees[1])
    h.it.load_file(ROOT / "tests" / "rust" / "mocks.rs")
    decs, readers = [], []
    for k, (s, packets) in enumerate(streams):
        p = h.params("CODEC_ID_MP3", s.rate, s.nch)
        front = h.it.call("mpa_front_end", p, h.opts())
        r = h.it.call("HipMpaDecoder::try_new_pooled", p, h.opts(), front.f["0"], usize(batch))
        assert r.variant == "Ok", r
        decs.append(r.f["0"])
        pk = I.Arr([h.packet(d, i * per_frame, track=1 + k, owned=True) for i, (d, _) in enumerate(packets)], True)
        readers.append(h.it.call("LookaheadReader::new", h.it.call("MockReader::new", pk), usize(3 * batch)))
    for i in range(n):
        for k in range(2):
            r = h.it.call_method("LookaheadReader", "next_packet", readers[k])
            pkt = r.f["0"].f["0"]
            st, got = h.decode("HipMpaDecoder", decs[k], h.it.call_method("Packet", "as_packet_ref", pkt))
            assert st == "ok" and np.array_equal(bits(got), bits(want[k][i])), (k, i)
    calls = h.bridge.calls
    assert calls.count("symaccel_batcher_create") == 1                                     # one pool for both decoders
    assert calls.count("symaccel_batcher_reserve") >= 2 * ((n - batch) // batch)             # every batch after the first: submitted ahead,
    assert calls.count("symaccel_batcher_commit") == calls.count("symaccel_batcher_reserve")  # written straight into the batcher's slots
    assert calls.count("symaccel_batcher_wait") >= calls.count("symaccel_batcher_reserve") - 2
    assert "symaccel_batcher_submit_mp3_decode" not in calls and "symaccel_batcher_collect" not in calls
    assert calls.count("symaccel_mp3_decode_pipelined") == 2                                 # each stream's cold start only
    # a seek + reset with a batch in flight: given up, then everything again from packet 0
    h.it.call_method("LookaheadReader", "seek", readers[0], I.Int(0, "i64"), usize(0))
    h.it.call_method("HipMpaDecoder", "reset", decs[0])
    for i in range(3):
        r = h.it.call_method("LookaheadReader", "next_packet", readers[0])
        st, got = h.decode("HipMpaDecoder", decs[0], h.it.call_method("Packet", "as_packet_ref", r.f["0"].f["0"]))
        assert st == "ok" and np.array_equal(bits(got), bits(want[0][i])), i


def test_seek_into_the_middle_of_the_stream_and_reset(trees):
    """A seek that lands on a packet whose main data reaches BACK into earlier packets (main_data_begin > 0), followed by reset():
    the reference rebuilds its whole state, bit reservoir included (decoder.rs:149-152), so the granules that would have started
    in the missing bytes are handled as an underflow.  The accelerated decoder's front end parsed AHEAD of what the caller had been
    given when the seek came; `reset` must reach the reference decoder inside the front end (`MpaFrontEnd::reset`), or the stale
    reservoir decodes those granules from the wrong bytes."""
    mpeg1, n, batch = True, sized(8, 5), 4
    per_frame = 1152
    s, packets = stream(11, n, "joint", mpeg1, 0)
    data = [p for p, _ in packets]
    backs = [rec["main_data_begin"] for _, rec in packets]
    k = next(i for i in range(2, n) if backs[i] > 0)  # the seek target reaches back
    ref = Harness(None, reference=True, mp3_tree=trees[0])
    ref_dec = cpu_decoder(ref, s)
    for i, pk in enumerate(data[:k + 2]):  # the reference plays on past the target, like the accelerated decoder's look-ahead does
        ref.decode("MpaDecoder", ref_dec, ref.packet(pk, i * per_frame))
    ref.it.call_method("MpaDecoder", "reset", ref_dec)
    want = [ref.decode("MpaDecoder", ref_dec, ref.packet(pk, (k + i) * per_frame)) for i, pk in enumerate(data[k:])]
    h, dec = hip_decoder(trees[1], s, max_batch=batch)
    h.it.load_file(ROOT / "tests" / "rust" / "mocks.rs")
    pk = I.Arr([h.packet(d, i * per_frame, track=1, owned=True) for i, d in enumerate(data)], True)
    reader = h.it.call("LookaheadReader::new", h.it.call("MockReader::new", pk), usize(8))

    def run(first, count):
        out = []
        for i in range(first, first + count):
            p = h.it.call_method("LookaheadReader", "next_packet", reader).f["0"].f["0"]
            assert p.f["pts"].f["0"].v == i * per_frame
            out.append(h.decode("HipMpaDecoder", dec, h.it.call_method("Packet", "as_packet_ref", p)))
        return out

    assert all(st == "ok" for st, _ in run(0, 2))  # (the first batch has parsed packets 0 .. batch - 1 by now)
    h.it.call_method("LookaheadReader", "seek", reader, I.Int(0, "i64"), usize(k))
    h.it.call_method("HipMpaDecoder", "reset", dec)
    got = run(k, n - k)
    assert [st for st, _ in got] == [st for st, _ in want]
    for i, ((st, a), (_, b)) in enumerate(zip(got, want)):
        assert (np.array_equal(bits(a), bits(b)) if st == "ok" else a == b), (k + i, st)


def test_decoders_built_by_the_registry_share_the_cross_stream_batcher(trees):
    """What an application gets: `register()` enters HipMpaDecoder at Tier::Preferred, `make_audio_decoder(params, opts)` builds every decoder
    from (params, opts) alone (codecs/registry.rs:34-44, 252-269, 330-341) -- and the decoders so built find each other in the
    process-wide `Pool`: two streams behind look-ahead readers, decoded alternately, every packet's PCM the reference decoder's bit for
    bit, their batches in common launches (symaccel_batcher_get_stats)."""
    from rs_harness import pool_stats, registry_round_trip
    mpeg1, n, depth = sized((True, 10, 6), (True, 7, 4))
    per_frame = 1152
    streams = [stream(8, n, "joint", mpeg1, 0), stream(9, n, "stereo", mpeg1, 0)]
    want = []
    for s, packets in streams:
        ref = Harness(None, reference=True, mp3_tree=trees[0])
        ref_dec = cpu_decoder(ref, s)
        want.append([ref.decode("MpaDecoder", ref_dec, ref.packet(pk, i * per_frame))[1] for i, (pk, _) in enumerate(packets)])
    h = shim(trees[1])
    h.it.load_file(ROOT / "tests" / "rust" / "mocks.rs")
    decs = registry_round_trip(h, "HipMpaDecoder", [h.params("CODEC_ID_MP3", s.rate, s.nch) for s, _ in streams])
    readers = []
    for k, (s, packets) in enumerate(streams):
        pk = I.Arr([h.packet(d, i * per_frame, track=1 + k, owned=True) for i, (d, _) in enumerate(packets)], True)
        readers.append(h.it.call("LookaheadReader::new", h.it.call("MockReader::new", pk), usize(depth)))
    for i in range(n):
        for k in range(2):
            r = h.it.call_method("LookaheadReader", "next_packet", readers[k])
            st, got = h.decode("HipMpaDecoder", decs[k], h.it.call_method("Packet", "as_packet_ref", r.f["0"].f["0"]))
            assert st == "ok" and np.array_equal(bits(got), bits(want[k][i])), (k, i)
    calls = h.bridge.calls
    assert calls.count("symaccel_batcher_create") == 1 and calls.count("symaccel_batcher_reserve") >= 2
    assert calls.count("symaccel_mp3_decode_pipelined") == 2  # each stream's cold start only
    stats = pool_stats(h)
    assert stats["submissions"] >= 2 and stats["launches"] < stats["submissions"] and stats["failed_tickets"] == 0, stats
