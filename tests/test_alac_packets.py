"""Packet bytes -> PCM for Apple Lossless, symphonia-check style (symphonia-check/src/main.rs:289-295), the ALAC twin of
tests/test_flac_packets.py:

  packets written by tests/alac_writer.py (SCE / CPE, compressed with both predictor modes and orders up to 31, verbatim (order 0),
  uncompressed, separately coded low bits, partial frames, fill / data-stream elements)
     |
     +--> the REFERENCE: symphonia-codec-alac's AlacDecoder (lib.rs) with symphonia-common's MagicCookie on symphonia-core's own
     |    BitReaderLtr, all EXECUTED from /root/reference by tools/rsinterp  ..................................................  PCM_ref
     |
     +--> the same decoder with bindings/rust/patches/symphonia-codec-alac.diff applied, default (CPU) backend  ...............  == PCM_ref
     |
     +--> HipAlacDecoder (frontends.rs -> alac.rs SeamFrontEnd = the patched decoder with the recording backend, AlacBatch,
          decoder.rs, lookahead.rs, ctx.rs) with its extern "C" calls bound to libsymaccel (the CPU-emulation build of the
          kernels): symaccel_alac_predict runs the dynamic predictor  .........................................................  == PCM_ref

PCM_ref is also the PCM the packets were encoded from.  Needs /root/reference (`localref`); the `-m gpu` twin of the accelerated
path is tests/test_rust_adapters.py::test_alac_adapter_restores_the_pcm (same adapter code, hipcc-built library, scripted front)."""
import sys
from pathlib import Path

import numpy as np
import pytest

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "tests"))

import alac_writer as W  # noqa: E402
from rs_harness import REF, Harness, patched_tree, sized, usize  # noqa: E402
from rsinterp import interp as I  # noqa: E402

pytestmark = pytest.mark.localref

CRATE = "symphonia-codec-alac"
# ALAC element order -> AudioBuffer plane (lib.rs:56-69)
PLANE_OF = {1: [0], 2: [0, 1], 3: [2, 0, 1], 6: [2, 0, 1, 4, 5, 3]}
ELEMENTS = {1: ["sce"], 2: ["cpe"], 3: ["sce", "cpe"], 6: ["sce", "cpe", "cpe", "sce"]}
KINDS = ["lpc", "lpc15", "verbatim", "raw", "lpc", "lpc"]


@pytest.fixture(scope="module")
def trees():
    return REF / CRATE / "src", patched_tree((CRATE,)) / CRATE / "src"


def stream(seed, n_packets, nch, depth, frame_length):
    """packets + the PCM they encode, [packet][plane][frame] in AudioBuffer plane order (None where a packet is partial)"""
    rng = np.random.default_rng(seed)
    packets, pcm = [], []
    for t in range(n_packets):
        n = frame_length if t != n_packets - 1 else frame_length - 29  # the last packet of a stream is a partial frame
        x = W.smooth_pcm(rng, nch, n, depth)
        kinds = ["%s:%s" % (el, KINDS[(t + i) % len(KINDS)]) for i, el in enumerate(ELEMENTS[nch])]
        packets.append(W.packet(x, depth, frame_length, kinds, rng, fill=(t % 3 == 2)))
        planes = np.zeros((nch, n), np.int64)
        for c in range(nch):
            planes[PLANE_OF[nch][c]] = x[c]
        pcm.append(planes)
    return packets, pcm


def left_justified(pcm, depth):
    v = pcm.astype(np.int64) << (32 - depth)
    return ((v + (1 << 31)) % (1 << 32)) - (1 << 31)


def cpu_decoder(h, nch, depth, frame_length):
    p = h.params("CODEC_ID_ALAC", extra=W.cookie(frame_length, depth, nch))
    r = h.it.call("AlacDecoder::try_new", p, h.opts())
    assert r.variant == "Ok", r
    return r.f["0"]


ALL_STREAMS = [(1, 7, 2, 16, 256), (2, 7, 1, 24, 160), (3, 6, 3, 24, 96), (4, 6, 6, 20, 64)]
STREAMS = sized(ALL_STREAMS, [(1, 7, 2, 16, 128), (4, 6, 6, 20, 64)])  # (SYMACCEL_PACKET_TESTS=full: all of them)


@pytest.mark.parametrize("seed,n_packets,nch,depth,frame_length", STREAMS)
def test_the_reference_decoder_and_its_patched_twin_decode_the_packets(trees, seed, n_packets, nch, depth, frame_length):
    packets, pcm = stream(seed, n_packets, nch, depth, frame_length)
    outs = []
    for tree in trees:
        h = Harness(None, reference=True, alac_tree=tree)
        dec = cpu_decoder(h, nch, depth, frame_length)
        got = []
        for i, pk in enumerate(packets):
            st, planes = h.decode("AlacDecoder", dec, h.packet(pk, i * frame_length))
            assert st == "ok", (i, planes)
            got.append(planes)
        outs.append(got)
        assert h.it.overflows == 0
    for i in range(n_packets):
        assert np.array_equal(outs[0][i], left_justified(pcm[i], depth)), "the reference's decoder does not give the encoded PCM back"
        assert np.array_equal(outs[0][i], outs[1][i]), "the seam patch changed what the decoder computes"


def hip_decoder(tree, nch, depth, frame_length, max_batch=None):
    from emu_lib import emu_library
    h = Harness(emu_library().dll, reference=True, alac_tree=tree)
    h.it.load_file(ROOT / "tests" / "rust" / "registry_stubs.rs")
    h.load_shim("lib.rs", "ctx.rs", "decoder.rs", "lookahead.rs", "fallback.rs", "alac.rs", "frontends.rs")
    p = h.params("CODEC_ID_ALAC", extra=W.cookie(frame_length, depth, nch))
    if max_batch is None:
        r = h.it.call("HipAlacDecoder::try_registry_new", p, h.opts())
    else:
        front = h.it.call("alac_front_end", p, h.opts())
        assert front.variant == "Ok", front
        r = h.it.call("HipAlacDecoder::try_new", p, h.opts(), front.f["0"], usize(max_batch))
    assert r.variant == "Ok", r
    return h, r.f["0"]


@pytest.mark.parametrize("seed,n_packets,nch,depth,frame_length", STREAMS)
def test_the_accelerated_decoder_equals_the_reference_on_packet_bytes(trees, seed, n_packets, nch, depth, frame_length):
    packets, pcm = stream(seed, n_packets, nch, depth, frame_length)
    ref = Harness(None, reference=True, alac_tree=trees[0])
    ref_dec = cpu_decoder(ref, nch, depth, frame_length)
    h, dec = hip_decoder(trees[1], nch, depth, frame_length)
    for i, pk in enumerate(packets):
        st_r, want = ref.decode("AlacDecoder", ref_dec, ref.packet(pk, i * frame_length))
        st, got = h.decode("HipAlacDecoder", dec, h.packet(pk, i * frame_length))
        assert st == st_r == "ok"
        assert np.array_equal(got, want), i
    assert h.bridge.calls.count("symaccel_alac_predict") == n_packets  # no look-ahead reader: batches of one
    cp = I.deref(h.it.call_method("HipAlacDecoder", "codec_params", dec))
    assert cp.f["sample_rate"].f["0"].v == 44100 and cp.f["channels"].f["0"].f["0"].v == nch


def test_damaged_packets_fail_like_the_reference_and_the_stream_goes_on(trees):
    nch, depth, frame_length = 2, 16, 128
    packets, pcm = stream(9, 8, nch, depth, frame_length)
    bad = {1: bytearray(packets[1]), 4: bytearray(packets[4]), 6: bytearray(packets[6])}
    bad[1][1] |= 0x08          # one of the twelve unused header bits
    bad[4] = bad[4][:30]       # truncated: the bit reader runs dry inside the residuals
    bad[6][0] = (2 << 5)       # a coupling channel element: unsupported in version 0
    data = [bytes(bad.get(i, p)) for i, p in enumerate(packets)]
    ref = Harness(None, reference=True, alac_tree=trees[0])
    ref_dec = cpu_decoder(ref, nch, depth, frame_length)
    h, dec = hip_decoder(trees[1], nch, depth, frame_length)
    outcomes = []
    for i, pk in enumerate(data):
        st_r, want = ref.decode("AlacDecoder", ref_dec, ref.packet(pk, i * frame_length))
        st, got = h.decode("HipAlacDecoder", dec, h.packet(pk, i * frame_length))
        assert st == st_r, (i, st, st_r, got, want)
        if st == "ok":
            assert np.array_equal(got, want), i
        else:
            assert got == want, (i, got, want)
        outcomes.append(st)
    assert outcomes.count("err") == 3 and outcomes[2] == "ok" and outcomes[7] == "ok"


def test_look_ahead_batches_and_reset(trees):
    nch, depth, frame_length = 2, 24, 96
    packets, pcm = stream(13, 11, nch, depth, frame_length)
    data = list(packets)
    broken = bytearray(data[6])
    broken[1] |= 0x08
    data[6] = bytes(broken)
    h, dec = hip_decoder(trees[1], nch, depth, frame_length, max_batch=4)
    h.it.load_file(ROOT / "tests" / "rust" / "mocks.rs")
    pk = I.Arr([h.packet(d, i * frame_length, track=1, owned=True) for i, d in enumerate(data)], True)
    reader = h.it.call("LookaheadReader::new", h.it.call("MockReader::new", pk), usize(8))

    def run(first, count):
        out = []
        for i in range(first, first + count):
            r = h.it.call_method("LookaheadReader", "next_packet", reader)
            p = r.f["0"].f["0"]
            assert p.f["pts"].f["0"].v == i * frame_length
            out.append(h.decode("HipAlacDecoder", dec, h.it.call_method("Packet", "as_packet_ref", p)))
        return out

    n0 = h.bridge.calls.count("symaccel_alac_predict")
    for i, (st, got) in enumerate(run(0, 11)):
        if i == 6:
            assert (st, got) == ("err", "DecodeError")
        else:
            assert st == "ok" and np.array_equal(got, left_justified(pcm[i], depth)), i
    # packets 0-3 | 4, 5 (the look-ahead stops in front of the corrupt packet 6) | 6 fails alone | 7-10
    assert h.bridge.calls.count("symaccel_alac_predict") - n0 == 3
    h.it.call_method("LookaheadReader", "seek", reader, I.Int(0, "i64"), usize(2))
    h.it.call_method("HipAlacDecoder", "reset", dec)
    for i, (st, got) in zip(range(2, 6), run(2, 4)):
        assert st == "ok" and np.array_equal(got, left_justified(pcm[i], depth)), i


def test_decoders_built_by_the_registry_share_the_cross_stream_batcher(trees):
    """What an application gets: `register()` enters HipAlacDecoder at Tier::Preferred, `make_audio_decoder(params, opts)` builds every decoder
    from (params, opts) alone (codecs/registry.rs:34-44, 252-269, 330-341) -- and the decoders so built find each other in the
    process-wide `Pool`: two streams behind look-ahead readers, decoded alternately, every packet's PCM the reference decoder's bit for
    bit, their batches in common launches (symaccel_batcher_get_stats)."""
    from emu_lib import emu_library
    from rs_harness import pool_stats, registry_round_trip
    nch, depth, frame_length = 2, 16, 128
    n, reader_depth = sized((12, 6), (8, 4))
    streams = [stream(31 + k, n, nch, depth, frame_length) for k in range(2)]
    h = Harness(emu_library().dll, reference=True, alac_tree=trees[1])
    h.it.load_file(ROOT / "tests" / "rust" / "registry_stubs.rs")
    h.it.load_file(ROOT / "tests" / "rust" / "mocks.rs")
    h.load_shim("lib.rs", "ctx.rs", "decoder.rs", "lookahead.rs", "fallback.rs", "alac.rs", "frontends.rs")
    p = h.params("CODEC_ID_ALAC", extra=W.cookie(frame_length, depth, nch))
    decs = registry_round_trip(h, "HipAlacDecoder", [p, p])
    readers = []
    for k, (packets, _) in enumerate(streams):
        pk = I.Arr([h.packet(d, i * frame_length, track=1 + k, owned=True) for i, d in enumerate(packets)], True)
        readers.append(h.it.call("LookaheadReader::new", h.it.call("MockReader::new", pk), usize(reader_depth)))
    for i in range(n):
        for k in range(2):
            r = h.it.call_method("LookaheadReader", "next_packet", readers[k])
            st, got = h.decode("HipAlacDecoder", decs[k], h.it.call_method("Packet", "as_packet_ref", r.f["0"].f["0"]))
            assert st == "ok" and np.array_equal(got, left_justified(streams[k][1][i], depth)), (k, i)
    calls = h.bridge.calls
    assert calls.count("symaccel_batcher_create") == 1 and calls.count("symaccel_batcher_reserve") >= 2
    assert calls.count("symaccel_alac_predict") == 2  # each stream's cold start only
    stats = pool_stats(h)
    assert stats["submissions"] >= 2 and stats["launches"] < stats["submissions"] and stats["failed_tickets"] == 0, stats
