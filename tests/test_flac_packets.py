"""Packet bytes -> PCM, symphonia-check style (symphonia-check/src/main.rs:289-295: decode the same packets two ways, compare the
samples), for FLAC -- the first codec for which the whole chain was run here without a Rust toolchain (the other four: test_{aac,mp3,vorbis,alac}_packets.py):

  frames written by tests/flac_writer.py (every subframe type, Rice / Rice2 / escaped partitions, wasted bits, all four channel
  assignments, 16- and 24-bit, frame CRC-8 / CRC-16)
     |
     +--> the REFERENCE: symphonia-bundle-flac's FlacDecoder (frame.rs, decoder.rs) on symphonia-core's own BufReader, BitReaderLtr,
     |    MonitorStream and CRC code, all EXECUTED from /root/reference by tools/rsinterp  ...................................  PCM_ref
     |
     +--> the same decoder with bindings/rust/patches/symphonia-bundle-flac.diff applied, default (CPU) backend  ..............  == PCM_ref
     |
     +--> HipFlacDecoder (bindings/rust/symphonia-accel-hip: frontends.rs -> flac.rs SeamFrontEnd = the patched decoder with the
          recording backend, FlacBatch, decoder.rs, lookahead.rs, ctx.rs) with its extern "C" calls bound to libsymaccel
          (the CPU-emulation build of the kernels): symaccel_flac_restore does the prediction  ...............................  == PCM_ref

PCM_ref is also the PCM the frames were encoded from (the writer is lossless), so the reference run validates the writer.
Needs /root/reference (`localref`); the `-m gpu` twin of the accelerated path is tests/test_rust_adapters.py::test_flac_adapter_*
(same adapter code, hipcc-built library, scripted front end)."""
import sys
from pathlib import Path

import numpy as np
import pytest

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "tests"))

import flac_writer as W  # noqa: E402
from rs_harness import REF, Harness, patched_tree, sized, u8_vec, usize  # noqa: E402
from rsinterp import interp as I  # noqa: E402

pytestmark = pytest.mark.localref


def streaminfo(blocksize, rate, nch, bps):
    bw = W.BitWriter()
    for v, n in ((blocksize, 16), (blocksize, 16), (0, 24), (0, 24), (rate, 20), (nch - 1, 3), (bps - 1, 5), (0, 36)):
        bw.put(v, n)
    for _ in range(16):
        bw.put(0, 8)
    return bw.bytes()


def left_justified(pcm, bps):
    v = pcm.astype(np.int64) << (32 - bps)
    return ((v + (1 << 31)) % (1 << 32)) - (1 << 31)


@pytest.fixture(scope="module")
def trees():
    return REF / "symphonia-bundle-flac" / "src", patched_tree(("symphonia-bundle-flac",)) / "symphonia-bundle-flac" / "src"


def cpu_decoder(h, nch, bps, blocksize):
    p = h.params("CODEC_ID_FLAC", extra=streaminfo(blocksize, 44100, nch, bps))
    r = h.it.call("FlacDecoder::try_new", p, h.opts())
    assert r.variant == "Ok", r
    return r.f["0"]


STREAMS = [(1, 8, 2, 16, 192), (2, 6, 1, 24, 576), (3, 5, 2, 24, 100), (4, 3, 2, 16, 1152)]


@pytest.mark.parametrize("seed,n_frames,nch,bps,blocksize", STREAMS)
def test_the_reference_decoder_and_its_patched_twin_decode_the_frames(trees, seed, n_frames, nch, bps, blocksize):
    frames, pcm = W.random_stream(seed, n_frames, nch, bps, blocksize)
    outs = []
    for tree in trees:
        h = Harness(None, reference=True, flac_tree=tree)
        dec = cpu_decoder(h, nch, bps, blocksize)
        got = []
        for i, fr in enumerate(frames):
            st, planes = h.decode("FlacDecoder", dec, h.packet(fr, i * blocksize))
            assert st == "ok", (i, planes)
            got.append(planes)
        outs.append(np.stack(got))
        assert h.it.overflows == 0  # no implicit integer wrap: a debug build of the reference would not have panicked either
    assert np.array_equal(outs[0], left_justified(pcm, bps)), "the reference's decoder does not give the encoded PCM back"
    assert np.array_equal(outs[0], outs[1]), "the seam patch changed what the decoder computes"


def hip_decoder(tree, nch, bps, blocksize, max_batch=None):
    from emu_lib import emu_library
    h = Harness(emu_library().dll, reference=True, flac_tree=tree)
    h.it.load_file(ROOT / "tests" / "rust" / "registry_stubs.rs")
    h.load_shim("lib.rs", "ctx.rs", "decoder.rs", "lookahead.rs", "fallback.rs", "flac.rs", "frontends.rs")
    p = h.params("CODEC_ID_FLAC", extra=streaminfo(blocksize, 44100, nch, bps))
    if max_batch is None:
        r = h.it.call("HipFlacDecoder::try_registry_new", p, h.opts())  # what the registry calls (registry.rs:34-44)
    else:
        front = h.it.call("flac_front_end", p, h.opts())
        assert front.variant == "Ok", front
        r = h.it.call("HipFlacDecoder::try_new", p, h.opts(), front.f["0"], usize(max_batch))
    assert r.variant == "Ok", r
    return h, r.f["0"]


@pytest.mark.parametrize("seed,n_frames,nch,bps,blocksize", sized(STREAMS[:3], STREAMS[2:3]))  # (SYMACCEL_PACKET_TESTS=full: three streams)
def test_the_accelerated_decoder_equals_the_reference_on_packet_bytes(trees, seed, n_frames, nch, bps, blocksize):
    frames, pcm = W.random_stream(seed, n_frames, nch, bps, blocksize)
    ref = Harness(None, reference=True, flac_tree=trees[0])
    ref_dec = cpu_decoder(ref, nch, bps, blocksize)
    h, dec = hip_decoder(trees[1], nch, bps, blocksize)
    for i, fr in enumerate(frames):
        st_r, want = ref.decode("FlacDecoder", ref_dec, ref.packet(fr, i * blocksize))
        st, got = h.decode("HipFlacDecoder", dec, h.packet(fr, i * blocksize))
        assert st == st_r == "ok"
        assert np.array_equal(got, want), i
    assert h.bridge.calls.count("symaccel_flac_restore") == n_frames  # no look-ahead reader: batches of one
    # the decoder reports the stream's parameters as the reference amends them from STREAMINFO (decoder.rs:110-118)
    cp = h.it.call_method("HipFlacDecoder", "codec_params", dec)
    cp = I.deref(cp)
    assert cp.f["sample_rate"].f["0"].v == 44100 and cp.f["bits_per_sample"].f["0"].v == bps
    assert cp.f["max_frames_per_packet"].f["0"].v == blocksize


def test_damaged_packets_fail_like_the_reference_and_the_stream_goes_on(trees):
    nch, bps, blocksize = 2, 16, 192
    frames, pcm = W.random_stream(7, 9, nch, bps, blocksize)
    bad = {2: bytearray(frames[2]), 5: bytearray(frames[5]), 6: bytearray(frames[6])}
    bad[2][3] ^= 0x10          # a header bit: CRC-8 mismatch
    bad[5] = bad[5][:40]       # truncated: the bit reader runs dry inside a subframe
    bad[6][4 + 2] = 0x7E       # (first subframe header byte) a reserved subframe type
    stream = [bytes(bad.get(i, f)) for i, f in enumerate(frames)]
    ref = Harness(None, reference=True, flac_tree=trees[0])
    ref_dec = cpu_decoder(ref, nch, bps, blocksize)
    h, dec = hip_decoder(trees[1], nch, bps, blocksize)
    outcomes = []
    for i, fr in enumerate(stream):
        st_r, want = ref.decode("FlacDecoder", ref_dec, ref.packet(fr, i * blocksize))
        st, got = h.decode("HipFlacDecoder", dec, h.packet(fr, i * blocksize))
        assert st == st_r, (i, st, st_r, got, want)
        if st == "ok":
            assert np.array_equal(got, want), i
        else:
            assert got == want, (i, got, want)  # the same Error variant
        outcomes.append(st)
    assert outcomes.count("err") == 3 and outcomes[3] == "ok" and outcomes[8] == "ok"


def test_look_ahead_batches_and_reset(trees):
    """behind a LookaheadReader the packets the demuxer has already read are parsed ahead and restored in ONE device call per
    batch; a corrupt packet inside the look-ahead ends the batch in front of it and fails at its own decode_ref; after a seek the
    application resets the decoder (codecs/audio.rs:252-257) and decoding starts over"""
    nch, bps, blocksize = 2, 16, 192
    frames, pcm = W.random_stream(11, 12, nch, bps, blocksize)
    stream = list(frames)
    broken = bytearray(stream[7])
    broken[3] ^= 0x10
    stream[7] = bytes(broken)
    h, dec = hip_decoder(trees[1], nch, bps, blocksize, max_batch=5)
    h.it.load_file(ROOT / "tests" / "rust" / "mocks.rs")
    packets = I.Arr([h.packet(fr, i * blocksize, track=1, owned=True) for i, fr in enumerate(stream)], True)
    reader = h.it.call("LookaheadReader::new", h.it.call("MockReader::new", packets), usize(8))
    want = left_justified(pcm, bps)

    def run(first, count):
        out = []
        for i in range(first, first + count):
            r = h.it.call_method("LookaheadReader", "next_packet", reader)
            p = r.f["0"].f["0"]
            assert p.f["pts"].f["0"].v == i * blocksize
            out.append(h.decode("HipFlacDecoder", dec, h.it.call_method("Packet", "as_packet_ref", p)))
        return out

    n0 = h.bridge.calls.count("symaccel_flac_restore")
    res = run(0, 12)
    for i, (st, got) in enumerate(res):
        if i == 7:
            assert (st, got) == ("err", "DecodeError")
        else:
            assert st == "ok" and np.array_equal(got, want[i]), i
    # packets 0-4 | 5, 6 (the look-ahead stops in front of the corrupt packet 7) | 7 fails alone | 8-11
    assert h.bridge.calls.count("symaccel_flac_restore") - n0 == 3
    # seek back to packet 3, reset, decode again
    h.it.call_method("LookaheadReader", "seek", reader, I.Int(0, "i64"), usize(3))
    h.it.call_method("HipFlacDecoder", "reset", dec)
    for i, (st, got) in zip(range(3, 7), run(3, 4)):
        assert st == "ok" and np.array_equal(got, want[i]), i


def test_decoders_built_by_the_registry_share_the_cross_stream_batcher(trees):
    """What an application gets: `register()` enters HipFlacDecoder at Tier::Preferred, `make_audio_decoder(params, opts)` builds every
    decoder from (params, opts) alone (codecs/registry.rs:34-44, 252-269, 330-341) -- and the decoders so built find each other in the
    process-wide `Pool`: two streams behind look-ahead readers, decoded alternately, every packet's PCM the reference decoder's, their
    batches in common launches (symaccel_batcher_get_stats), written straight into the batcher's page-locked slots."""
    from emu_lib import emu_library
    from rs_harness import pool_stats, registry_round_trip
    nch, bps, blocksize = 2, 16, 192
    n, depth = sized((12, 6), (8, 4))
    streams = [W.random_stream(21 + k, n, nch, bps, blocksize) for k in range(2)]
    h = Harness(emu_library().dll, reference=True, flac_tree=trees[1])
    h.it.load_file(ROOT / "tests" / "rust" / "registry_stubs.rs")
    h.it.load_file(ROOT / "tests" / "rust" / "mocks.rs")
    h.load_shim("lib.rs", "ctx.rs", "decoder.rs", "lookahead.rs", "fallback.rs", "flac.rs", "frontends.rs")
    p = h.params("CODEC_ID_FLAC", extra=streaminfo(blocksize, 44100, nch, bps))
    decs = registry_round_trip(h, "HipFlacDecoder", [p, p])
    readers = []
    for k, (frames, _) in enumerate(streams):
        pk = I.Arr([h.packet(fr, i * blocksize, track=1 + k, owned=True) for i, fr in enumerate(frames)], True)
        readers.append(h.it.call("LookaheadReader::new", h.it.call("MockReader::new", pk), usize(depth)))
    for i in range(n):
        for k in range(2):
            r = h.it.call_method("LookaheadReader", "next_packet", readers[k])
            st, got = h.decode("HipFlacDecoder", decs[k], h.it.call_method("Packet", "as_packet_ref", r.f["0"].f["0"]))
            assert st == "ok" and np.array_equal(got, left_justified(streams[k][1][i], bps)), (k, i)
    calls = h.bridge.calls
    assert calls.count("symaccel_batcher_create") == 1 and calls.count("symaccel_batcher_reserve") >= 2
    assert calls.count("symaccel_flac_restore") == 2  # each stream's cold start only: every later batch went through the batcher
    stats = pool_stats(h)
    assert stats["submissions"] >= 2 and stats["launches"] < stats["submissions"] and stats["failed_tickets"] == 0, stats
