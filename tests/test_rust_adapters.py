"""The codec adapters of the Rust shim crate -- bindings/rust/symphonia-accel-hip/src/{aac,mpa,vorbis,flac,alac,ctx,decoder,lookahead}.rs --
EXECUTED under the repository's Rust interpreter with their `unsafe extern "C"` calls bound through ctypes to libsymaccel
(tools/rsinterp/ffi.py): the CPU-emulation build of the kernels here, the hipcc-built library in the `-m gpu` twin of every test.

No Rust toolchain exists in the image, so this is how `BatchCodec::transform / publish` -- the index arithmetic `(c * k + i) * 1024`,
the side bytes, `Pinned`, the packed Vorbis spans, the FLAC descriptor slots -- gets run at all ("what's missing" 2).
The CPU front ends are scripted (tests/rust/mock_fronts.rs) and replay the inputs of the reference-text fixtures
(tests/golden/rs_fixtures: the reference's own functions executed on these inputs); what `decode_ref` leaves in the decoder's
`AudioBuffer` must equal the fixtures' PCM bit for bit -- across batch boundaries, across `reset()`, and around a packet that does
not parse (buffer cleared, codecs/audio.rs:273-278).  The first frames of a chain depend on the fixtures' non-zero initial state,
which a freshly built decoder does not have: they are compared with the oracle run from a zero state instead."""
import sys
from pathlib import Path

import numpy as np
import pytest

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "tests"))

import oracle  # noqa: E402
from rs_harness import Harness, f32_vec, i32_vec, u8_vec, usize  # noqa: E402
from rsinterp import interp as I  # noqa: E402

FIX = ROOT / "tests" / "golden" / "rs_fixtures"


def f32(bits):
    return np.ascontiguousarray(bits).view(np.float32)


def emu_dll():
    from emu_lib import emu_library
    return emu_library().dll


def gpu_dll():
    torch = pytest.importorskip("torch")
    if not torch.cuda.is_available():
        pytest.fail("no GPU visible: the gpu-marked tests must run on an MI355X (there is no CPU path)")
    from symphonia_amd._ffi import default_library
    return default_library().dll


LIBS = [pytest.param(emu_dll, id="emulated"), pytest.param(gpu_dll, id="mi355x", marks=pytest.mark.gpu)]


def harness(make_dll, *codec_files):
    h = Harness(make_dll())
    h.load_shim("ctx.rs", "decoder.rs", "lookahead.rs", *codec_files)
    h.it.load_file(ROOT / "tests" / "rust" / "mock_fronts.rs")
    return h


def key(i):
    return bytes([i % 256, i // 256])


def wrap32(v):
    return ((np.asarray(v, np.int64) + (1 << 31)) % (1 << 32)) - (1 << 31)


def same(got, want):
    got, want = np.asarray(got), np.asarray(want)
    return got.shape == want.shape and bool(np.array_equal(got, want)) and not np.isnan(got.astype(np.float64)).any()


def batcher_stats(h):
    """symaccel_batcher_get_stats of the process-wide Pool the interpreted crate created (ctx.rs `Pool::shared` / `Pool::stats`)"""
    pool = h.it.call("Pool::shared")
    assert pool.variant == "Ok", pool
    arc = pool.f["0"]
    r = h.it.call_method("Pool", "stats", arc.v if hasattr(arc, "v") else arc)
    assert r.variant == "Ok", r
    return {k: int(v.v) for k, v in r.f["0"].f.items()}


# ------------------------------------------------------------------------------------------------ AAC

def aac_decoder(h, coeffs, side, max_batch):
    """coeffs[channel][frame][1024], side[frame][channel] -> a HipAacDecoder over a scripted front end"""
    nch, nfr = coeffs.shape[0], coeffs.shape[1]
    script = I.Arr([I.Struct("ParsedAac", {"coeffs": f32_vec(coeffs[:, t]), "side": u8_vec(side[t]), "fused": I.NONE}) for t in range(nfr)], True)
    params = h.params("CODEC_ID_AAC", 48000, nch)
    front = I.Struct("ScriptedAacFront", {"params": params, "nch": usize(nch), "script": script, "parses": usize(0), "resets": usize(0)})
    r = h.it.call("HipAacDecoder::try_new", params, h.opts(), front, usize(max_batch))
    assert r.variant == "Ok", r
    return r.f["0"], r.f["0"].f["batch"].f["front"]  # (the decoder owns the front end: `front` was moved into it)


@pytest.mark.parametrize("make_dll", LIBS)
@pytest.mark.parametrize("walk,max_batch", [("a", 4), ("c", 1), ("d", 16)])
def test_aac_adapter_publishes_the_fixture_pcm(make_dll, walk, max_batch):
    h = harness(make_dll, "aac.rs")
    f = np.load(FIX / "aac.npz")
    coeffs, want = f32(f["coeffs_" + walk]), f32(f["pcm_" + walk])          # [lane][frame][1024]: the lanes are the channels here
    sd = f["side_" + walk]                                                  # [frame][seq, shape, prev_shape]
    nch, nfr = coeffs.shape[0], coeffs.shape[1]
    side = np.repeat(oracle.aac_side(sd[:, 0], sd[:, 1], sd[:, 2])[:, None], nch, axis=1)
    zero_state, _ = oracle.aac_synth(coeffs, side.T.copy(), np.zeros((nch, 1024), np.float32))
    dec, front = aac_decoder(h, coeffs, side, max_batch)
    # every packet once, as an application without a LookaheadReader decodes: batches of one ...
    for t in range(nfr):
        st, got = h.decode("HipAacDecoder", dec, h.packet(key(t), 1024 * t))
        assert st == "ok" and got.shape == (nch, 1024)
        assert same(got, want[:, t] if t >= 1 else zero_state[:, 0]), (walk, t)  # (frame 0 laps with the fixture's delay line)
    assert front.f["parses"].v == nfr
    # ... a packet that does not parse: DecodeError, buffer cleared, and the stream goes on (the delay line is the one the
    # last good packet left: the next frame's left half is lapped with it, like the reference after a discarded packet)
    st, err = h.decode("HipAacDecoder", dec, h.packet(bytes([255, 0]), 0))
    assert (st, err) == ("err", "DecodeError")
    # ... reset(): the delay lines are zeroed (codecs/audio.rs:252-257) -- frame 0 again gives the zero-state answer
    h.it.call_method("HipAacDecoder", "reset", dec)
    st, got = h.decode("HipAacDecoder", dec, h.packet(key(0), 0))
    assert st == "ok" and same(got, zero_state[:, 0])
    assert "symaccel_aac_synth" in h.bridge.calls and "symaccel_ctx_create" in h.bridge.calls


@pytest.mark.parametrize("make_dll", LIBS)
def test_aac_adapter_batches_behind_a_lookahead_reader(make_dll):
    """with the reader-side look-ahead the same packets go through `transform` in batches of max_batch: same PCM, fewer calls"""
    h = harness(make_dll, "aac.rs")
    h.it.load_file(ROOT / "tests" / "rust" / "mocks.rs")
    f = np.load(FIX / "aac.npz")
    coeffs, want, sd = f32(f["coeffs_c"]), f32(f["pcm_c"]), f["side_c"]
    nch, nfr = coeffs.shape[0], coeffs.shape[1]
    side = np.repeat(oracle.aac_side(sd[:, 0], sd[:, 1], sd[:, 2])[:, None], nch, axis=1)
    dec, front = aac_decoder(h, coeffs, side, 6)
    packets = I.Arr([h.packet(key(t), 1024 * t, track=3, owned=True) for t in range(nfr)], True)
    inner = h.it.call("MockReader::new", packets)
    reader = h.it.call("LookaheadReader::new", inner, usize(8))
    n_calls0 = h.bridge.calls.count("symaccel_aac_synth")
    for t in range(nfr):
        r = h.it.call_method("LookaheadReader", "next_packet", reader)
        p = r.f["0"].f["0"]
        st, got = h.decode("HipAacDecoder", dec, h.it.call_method("Packet", "as_packet_ref", p))
        assert st == "ok"
        if t >= 1:
            assert same(got, want[:, t]), t
    assert h.bridge.calls.count("symaccel_aac_synth") - n_calls0 == -(-nfr // 6)  # 15 packets, batches of 6, 6, 3
    assert front.f["parses"].v == nfr


@pytest.mark.parametrize("make_dll", LIBS)
def test_two_pooled_aac_adapters_share_the_batcher(make_dll):
    """`HipAacDecoder::try_new_pooled` over scripted front ends that hand on the spectrum decoder's form (`ParsedAac::fused`): two
    decoders behind look-ahead readers, decoded alternately -- the batches after each stream's first go through
    symaccel_batcher_submit_aac_decode / _collect of ONE `Pool` (here and, in the gpu twin, on the MI355X), same PCM."""
    h = harness(make_dll, "aac.rs")
    h.it.load_file(ROOT / "tests" / "rust" / "mocks.rs")
    f = np.load(FIX / "aac.npz")
    swb_long = I.Arr([I.Int(v, "u16") for v in (0, 4, 8, 16, 32, 64, 128, 256, 512, 1024)], True)
    swb_short = I.Arr([I.Int(v, "u16") for v in (0, 4, 8, 16, 32, 64, 128)], True)
    decs, readers, wants = [], [], []
    for k, walk in enumerate(("c", "d")):
        coeffs, want, sd = f32(f["coeffs_" + walk]), f32(f["pcm_" + walk]), f["side_" + walk]
        nch, nfr = coeffs.shape[0], coeffs.shape[1]
        side = np.repeat(oracle.aac_side(sd[:, 0], sd[:, 1], sd[:, 2])[:, None], nch, axis=1)
        fused = lambda: I.some(I.Struct("FusedAac", {"joint": I.Arr([], True), "tns": I.Arr([], True), "swb_long": swb_long,  # noqa: E731
                                                     "swb_short": swb_short}))
        script = I.Arr([I.Struct("ParsedAac", {"coeffs": f32_vec(coeffs[:, t]), "side": u8_vec(side[t]), "fused": fused()}) for t in range(nfr)], True)
        params = h.params("CODEC_ID_AAC", 48000, nch)
        front = I.Struct("ScriptedAacFront", {"params": params, "nch": usize(nch), "script": script, "parses": usize(0), "resets": usize(0)})
        r = h.it.call("HipAacDecoder::try_new_pooled", params, h.opts(), front, usize(4))
        assert r.variant == "Ok", r
        decs.append(r.f["0"])
        packets = I.Arr([h.packet(key(t), 1024 * t, track=3 + k, owned=True) for t in range(nfr)], True)
        readers.append(h.it.call("LookaheadReader::new", h.it.call("MockReader::new", packets), usize(12)))
        wants.append(want)
    nfr = min(w.shape[1] for w in wants)
    for t in range(nfr):
        for k in range(2):
            r = h.it.call_method("LookaheadReader", "next_packet", readers[k])
            st, got = h.decode("HipAacDecoder", decs[k], h.it.call_method("Packet", "as_packet_ref", r.f["0"].f["0"]))
            assert st == "ok"
            if t >= 1:
                assert same(got, wants[k][:, t]), (k, t)
    calls = h.bridge.calls
    assert calls.count("symaccel_batcher_create") == 1 and calls.count("symaccel_batcher_submit_aac_decode") >= 2 * ((nfr - 4) // 4)
    assert calls.count("symaccel_batcher_collect") >= calls.count("symaccel_batcher_submit_aac_decode") - 2


# ------------------------------------------------------------------------------------------------ MP3

@pytest.mark.parametrize("make_dll", LIBS)
@pytest.mark.parametrize("chain,max_batch", [("long", 2), ("switch", 8), ("sr3", 1), ("mix4", 3)])
def test_mpa_adapter_publishes_the_fixture_pcm(make_dll, chain, max_batch):
    h = harness(make_dll, "mpa.rs")
    f = np.load(FIX / "mp3.npz")
    k = "chain_%s_" % chain
    xr, want = f32(f[k + "xr"]), f32(f[k + "pcm"])                           # [lane][granule][576]
    sd, sr = f[k + "side"], int(f[k + "sr"][0])                             # [granule][block type, mixed, rzero]
    nch, ngr = xr.shape[0], xr.shape[1] & ~1
    gpp = 2                                                                 # granules per packet (an MPEG-1 frame)
    side = oracle.mp3_side(np.tile(sd[:, 0], (nch, 1)), np.tile(sd[:, 1], (nch, 1)), np.tile(sd[:, 2], (nch, 1)))
    z = (np.zeros((nch, 576), np.float32), np.zeros((nch, 1024), np.float32), np.zeros(nch, np.int32))
    zero_state = oracle.mp3_synth(xr, side, sr, *z)[0]

    def parsed(p):
        rows = []
        for g in range(gpp * p, gpp * p + gpp):
            for c in range(nch):
                rows.append(I.Struct("SymaccelMp3Side", {"block_type": I.Int(int(sd[g, 0]), "u8"), "is_mixed": I.Int(int(sd[g, 1]), "u8"),
                                                         "rzero": I.Int(int(sd[g, 2]), "u16")}))
        lines = np.stack([xr[c, g] for g in range(gpp * p, gpp * p + gpp) for c in range(nch)])
        return I.Struct("ParsedMpa", {"trim": (usize(0), usize(0)), "n_granules": usize(gpp), "xr": f32_vec(lines), "side": I.Arr(rows, True), "fused": I.NONE})

    script = I.Arr([parsed(p) for p in range(ngr // gpp)], True)
    front = I.Struct("ScriptedMpaFront", {"nch": usize(nch), "sr_idx": I.Int(sr, "i32"), "script": script, "parses": usize(0), "resets": usize(0)})
    r = h.it.call("HipMpaDecoder::try_new", h.params("CODEC_ID_MP3", 44100, nch), h.opts(), front, usize(max_batch))
    assert r.variant == "Ok", r
    dec = r.f["0"]
    for p in range(ngr // gpp):
        st, got = h.decode("HipMpaDecoder", dec, h.packet(key(p), 1152 * p))
        assert st == "ok" and got.shape == (nch, 1152)
        for j in range(gpp):
            g = gpp * p + j
            ref = want[:, g] if g >= 2 else zero_state[:, g]  # (granules 0 and 1 still see the fixture's overlap and V FIFO)
            assert same(got[:, 576 * j:576 * j + 576], ref), (chain, g)
    h.it.call_method("HipMpaDecoder", "reset", dec)
    st, got = h.decode("HipMpaDecoder", dec, h.packet(key(0), 0))
    assert st == "ok" and same(got[:, :576], zero_state[:, 0]) and same(got[:, 576:], zero_state[:, 1])
    st, err = h.decode("HipMpaDecoder", dec, h.packet(bytes([255, 0]), 0))
    assert (st, err) == ("err", "DecodeError")
    assert "symaccel_mp3_synth" in h.bridge.calls


# ------------------------------------------------------------------------------------------------ Vorbis

@pytest.mark.parametrize("make_dll", LIBS)
@pytest.mark.parametrize("pair,max_batch", [((8, 11), 4), ((6, 8), 1), ((7, 10), 16), ((9, 12), 3)])
def test_vorbis_adapter_publishes_the_fixture_pcm(make_dll, pair, max_batch):
    h = harness(make_dll, "vorbis.rs")
    f = np.load(FIX / "vorbis.npz")
    b0, b1 = pair
    k = "synth_%d_%d_" % pair
    flags, prev0 = f[k + "flags"], int(f[k + "prev_flag"][0])
    spectra, want = f32(f[k + "spectra"]), f32(f[k + "pcm"])               # [lane][packed]
    nch, nb = spectra.shape[0], flags.size
    so, po = oracle.vorbis_layout(b0, b1, np.tile(flags, (nch, 1)), np.full(nch, prev0))
    script = I.Arr([I.Struct("ParsedVorbis", {"trim": (usize(0), usize(0)), "long_block": bool(flags[b]),
                                               "spectra": f32_vec(spectra[:, so[0, b]:so[0, b + 1]]), "fused": I.NONE}) for b in range(nb)], True)
    front = I.Struct("ScriptedVorbisFront", {"nch": usize(nch), "bs0_exp": I.Int(b0, "i32"), "bs1_exp": I.Int(b1, "i32"), "script": script,
                                              "parses": usize(0), "resets": usize(0)})
    r = h.it.call("HipVorbisDecoder::try_new", h.params("CODEC_ID_VORBIS", 44100, nch), h.opts(gapless=True), front, usize(max_batch))
    assert r.variant == "Ok", r
    dec = r.f["0"]
    for b in range(nb):
        st, got = h.decode("HipVorbisDecoder", dec, h.packet(key(b), 1000 * b))
        assert st == "ok"
        if b == 0:
            assert got.size == 0  # the first packet after a reset is silenced (lib.rs:335-338, gapless)
            continue
        ref = want[:, po[0, b]:po[0, b + 1]]  # a block's samples depend on this block's and the previous block's input alone
        assert got.shape == ref.shape and same(got, ref), (pair, b)
    # without gapless support the batched decoder declines (the CPU decoder below it takes the track, fallback.rs)
    r = h.it.call("HipVorbisDecoder::try_new", h.params("CODEC_ID_VORBIS", 44100, nch), h.opts(gapless=False), front, usize(2))
    assert r.variant == "Err" and r.f["0"].variant == "Unsupported"
    # reset, then a packet that does not parse, then the stream again
    h.it.call_method("HipVorbisDecoder", "reset", dec)
    st, err = h.decode("HipVorbisDecoder", dec, h.packet(bytes([255, 0]), 0))
    assert (st, err) == ("err", "DecodeError")
    st, got = h.decode("HipVorbisDecoder", dec, h.packet(key(0), 0))
    assert st == "ok" and got.size == 0
    st, got = h.decode("HipVorbisDecoder", dec, h.packet(key(1), 1000))
    assert st == "ok" and same(got, want[:, po[0, 1]:po[0, 2]])
    assert "symaccel_vorbis_synth" in h.bridge.calls


@pytest.mark.parametrize("make_dll", LIBS)
def test_two_pooled_vorbis_adapters_share_the_batcher(make_dll):
    """`HipVorbisDecoder::try_new_pooled` over scripted front ends that hand on finished spectra (SYMACCEL_BATCH_VORBIS_SYNTH): two streams
    of one block-size pair with DIFFERENT block flags behind look-ahead readers; the batches after each stream's first are written into
    page-locked slots of ONE `Pool` (planes at their largest, packed data at the front) and the PCM is published from the slot."""
    h = harness(make_dll, "vorbis.rs")
    h.it.load_file(ROOT / "tests" / "rust" / "mocks.rs")
    f = np.load(FIX / "vorbis.npz")
    b0, b1 = 8, 11
    k = "synth_%d_%d_" % (b0, b1)
    flags0, prev0 = f[k + "flags"], int(f[k + "prev_flag"][0])
    spectra0 = f32(f[k + "spectra"])
    nch = spectra0.shape[0]
    so0, _ = oracle.vorbis_layout(b0, b1, np.tile(flags0, (nch, 1)), np.full(nch, prev0))
    decs, readers, wants, layouts = [], [], [], []
    for s in range(2):
        # stream 1: the fixture's blocks in reverse order (other flags at every position); expectation = the oracle from a reset state
        order = list(range(flags0.size)) if s == 0 else list(range(flags0.size))[::-1]
        flags = flags0[order]
        blocks = [spectra0[:, so0[0, b]:so0[0, b + 1]] for b in order]
        packed = np.ascontiguousarray(np.concatenate(blocks, axis=1))
        fl = np.tile(flags, (nch, 1))
        prev = np.full(nch, -1, np.int32)
        so, po = oracle.vorbis_layout(b0, b1, fl, prev)
        want = oracle.vorbis_synth(b0, b1, packed, fl, prev, np.zeros((nch, 1 << (b1 - 1)), np.float32), int(po[0, -1] + 3) & ~3)[0]
        script = I.Arr([I.Struct("ParsedVorbis", {"trim": (usize(0), usize(0)), "long_block": bool(flags[b]), "spectra": f32_vec(blocks[b]),
                                                   "fused": I.NONE}) for b in range(flags.size)], True)
        front = I.Struct("ScriptedVorbisFront", {"nch": usize(nch), "bs0_exp": I.Int(b0, "i32"), "bs1_exp": I.Int(b1, "i32"), "script": script,
                                                  "parses": usize(0), "resets": usize(0)})
        r = h.it.call("HipVorbisDecoder::try_new_pooled", h.params("CODEC_ID_VORBIS", 44100, nch), h.opts(gapless=True), front, usize(3))
        assert r.variant == "Ok", r
        decs.append(r.f["0"])
        packets = I.Arr([h.packet(key(t), 1000 * t, track=9 + s, owned=True) for t in range(flags.size)], True)
        readers.append(h.it.call("LookaheadReader::new", h.it.call("MockReader::new", packets), usize(9)))
        wants.append(np.asarray(want))
        layouts.append(po)
    for b in range(flags0.size):
        for s in range(2):
            r = h.it.call_method("LookaheadReader", "next_packet", readers[s])
            st, got = h.decode("HipVorbisDecoder", decs[s], h.it.call_method("Packet", "as_packet_ref", r.f["0"].f["0"]))
            assert st == "ok"
            if b == 0:
                assert got.size == 0
                continue
            po = layouts[s]
            ref = wants[s][:, po[0, b]:po[0, b + 1]]
            assert got.shape == ref.shape and same(got, ref), (s, b)
    calls = h.bridge.calls
    assert calls.count("symaccel_batcher_create") == 1 and calls.count("symaccel_batcher_reserve") >= 4
    assert "symaccel_batcher_collect" not in calls
    stats = batcher_stats(h)
    assert stats["launches"] < stats["submissions"] and stats["failed_tickets"] == 0, stats


# ------------------------------------------------------------------------------------------------ FLAC

def flac_script(rng, bps, blocksize, nfr, nch=2):
    """what the reference's parser would leave behind its seam for `nfr` frames (tests/flac_writer.py computes it while it encodes):
    (ParsedFlac script, the PCM per frame [channel][blocksize])"""
    import flac_writer as W
    script, pcm = [], []
    for t in range(nfr):
        chans = [rng.integers(-(1 << (bps - 2)), 1 << (bps - 2), blocksize) for _ in range(nch)]
        mode = t % 4                                                        # 0 independent, 1 left/side, 2 mid/side, 3 right/side
        left, right = chans
        planes = {0: [left, right], 1: [left, left - right], 2: [(left + right) >> 1, left - right], 3: [left - right, right]}[mode]
        words, descs, coeffs = [], [], np.zeros((nch, 32), np.int64)
        for c in range(nch):
            x = [int(v) for v in planes[c]]
            pick = (t + 2 * c) % 4
            wasted = 0
            if pick == 0:
                kind, order, shift, res = 0, 0, 0, x                          # verbatim
            elif pick in (1, 2):
                order = int(rng.integers(0, 5)) if pick == 1 else 2
                kind, shift, res = 1, 0, W.predict_residual(x, W.FIXED_COEFFS[order], 0)
            else:
                order, shift = int(rng.choice([1, 4, 8, 12, 32])), int(rng.integers(6, 13))
                co = np.clip(np.round((rng.standard_normal(order) * 0.5 ** np.arange(order) + (np.arange(order) == 0)) * (1 << shift)), -16000, 16000)
                kind, res = 2, W.predict_residual(x, [int(v) for v in co], shift)
                coeffs[c, :order] = co                                          # bitstream order (include/symaccel.h)
            if mode == 0 and pick == 0 and t % 2:
                wasted = 3                                                      # samples_shl after the (verbatim) subframe
                res = [v for v in res]
                planes[c] = planes[c] << wasted
                chans[c] = chans[c] << wasted
            words.append(res)
            descs.append(I.Struct("SymaccelFlacDesc", {"kind": I.Int(kind, "u8"), "order": I.Int(order, "u8"), "shift": I.Int(shift, "u8"),
                                                         "wasted_bits": I.Int(wasted, "u8")}))
        script.append(I.Struct("ParsedFlac", {"blocksize": usize(blocksize), "words": i32_vec(np.array(words)), "desc": I.Arr(descs, True),
                                               "coeffs": i32_vec(coeffs), "pair_mode": I.Int(mode, "u8"), "out_shift": I.Int(32 - bps, "u32")}))
        pcm.append(np.stack(chans))
    return script, pcm



@pytest.mark.parametrize("make_dll", LIBS)
@pytest.mark.parametrize("bps,blocksize,max_batch", [(16, 192, 4), (24, 576, 1), (16, 60, 8)])
def test_flac_adapter_restores_the_pcm(make_dll, bps, blocksize, max_batch):
    """every subframe type, wasted bits and channel assignment through FlacBatch::transform / publish: the descriptors and
    residuals are what the reference's parser would leave behind its seam (tests/flac_writer.py computes them while it
    encodes), the PCM must come back exactly -- the encoder identity, the same criterion as config 5's"""
    h = harness(make_dll, "flac.rs")
    rng = np.random.default_rng(bps + blocksize)
    nfr, nch = 9, 2
    script, pcm = flac_script(rng, bps, blocksize, nfr, nch)
    params = h.params("CODEC_ID_FLAC", 44100, nch, bps=bps)
    front = I.Struct("ScriptedFlacFront", {"params": params, "nch": usize(nch), "max_bs": usize(blocksize), "script": I.Arr(script, True),
                                            "parses": usize(0)})
    r = h.it.call("HipFlacDecoder::try_new", params, h.opts(), front, usize(max_batch))
    assert r.variant == "Ok", r
    dec = r.f["0"]
    for t in range(nfr):
        st, got = h.decode("HipFlacDecoder", dec, h.packet(key(t), blocksize * t))
        want = (pcm[t].astype(np.int64) << (32 - bps))
        want = ((want + (1 << 31)) % (1 << 32)) - (1 << 31)
        assert st == "ok" and np.array_equal(got, want), (t, t % 4)
    st, err = h.decode("HipFlacDecoder", dec, h.packet(bytes([255, 0]), 0))
    assert (st, err) == ("err", "DecodeError")
    r = h.it.call("HipFlacDecoder::try_new", params, h.opts(verify=True), front, usize(2))
    assert r.variant == "Err" and r.f["0"].variant == "Unsupported"  # the MD5 check is the caller's with this decoder
    assert "symaccel_flac_restore" in h.bridge.calls


@pytest.mark.parametrize("make_dll", LIBS)
def test_two_pooled_flac_adapters_share_the_batcher(make_dll):
    """`HipFlacDecoder::try_new_pooled`: two streams (16 and 24 bit, the same maximum block size) behind look-ahead readers, decoded
    alternately.  The batches after each stream's first are written straight into page-locked slots of ONE `Pool`
    (`Pool::reserve` / `commit` / `wait` / `release`: SYMACCEL_BATCH_FLAC_RESTORE) and published from there; every packet's PCM is
    the encoder's input, the two streams' subframes went to the device in common launches."""
    h = harness(make_dll, "flac.rs")
    h.it.load_file(ROOT / "tests" / "rust" / "mocks.rs")
    blocksize, nfr = 96, 13
    decs, readers, wants, shifts = [], [], [], []
    for k, bps in enumerate((16, 24)):
        rng = np.random.default_rng(700 + bps)
        script, pcm = flac_script(rng, bps, blocksize, nfr)
        params = h.params("CODEC_ID_FLAC", 44100, 2, bps=bps)
        front = I.Struct("ScriptedFlacFront", {"params": params, "nch": usize(2), "max_bs": usize(blocksize), "script": I.Arr(script, True),
                                                "parses": usize(0)})
        r = h.it.call("HipFlacDecoder::try_new_pooled", params, h.opts(), front, usize(4))
        assert r.variant == "Ok", r
        decs.append(r.f["0"])
        packets = I.Arr([h.packet(key(t), blocksize * t, track=5 + k, owned=True) for t in range(nfr)], True)
        readers.append(h.it.call("LookaheadReader::new", h.it.call("MockReader::new", packets), usize(12)))
        wants.append(pcm)
        shifts.append(32 - bps)
    for t in range(nfr):
        for k in range(2):
            r = h.it.call_method("LookaheadReader", "next_packet", readers[k])
            st, got = h.decode("HipFlacDecoder", decs[k], h.it.call_method("Packet", "as_packet_ref", r.f["0"].f["0"]))
            want = wrap32(wants[k][t].astype(np.int64) << shifts[k])
            assert st == "ok" and np.array_equal(got, want), (k, t)
    calls = h.bridge.calls
    assert calls.count("symaccel_batcher_create") == 1 and calls.count("symaccel_batcher_reserve") >= 4
    assert calls.count("symaccel_batcher_wait") >= calls.count("symaccel_batcher_reserve") - 2
    assert "symaccel_batcher_submit" not in calls and "symaccel_batcher_collect" not in calls  # zero-copy: no staging copies
    stats = batcher_stats(h)
    assert stats["launches"] < stats["submissions"], stats  # the two streams' batches shared launches


# ------------------------------------------------------------------------------------------------ ALAC

def alac_script(rng, depth, frames, npk, nch=2):
    """`npk` packets as the reference's parser leaves them behind its seam (residuals, predictors, pairs, low bits) and the PCM they
    decode to [packet][channel][frames], left-justified"""
    script, want = [], []
    for t in range(npk):
        n = frames if t != npk - 1 else frames - 37  # the last packet of a stream is short
        tail_shift = 8 if (t % 3 == 1 and depth > 16) else 0
        is_pair = t % 4 != 3
        bps = depth - tail_shift + (1 if is_pair else 0)
        uncompressed = t % 5 == 4
        words, descs, coeffs, planes = [], [], np.zeros((nch, 32), np.int64), []
        for c in range(nch):
            if uncompressed:
                mode, order, qshift = 0, 0, 0
                res = rng.integers(-(1 << (depth - 1)), 1 << (depth - 1), n)
            else:
                mode = 15 if (t + c) % 4 == 2 else 0
                order = int(rng.choice([0, 1, 4, 8, 12, 31]))
                qshift = int(rng.integers(4, 10))
                co = (rng.integers(-300, 300, 32) * (0.7 ** np.arange(32))).astype(np.int64)
                co[order:] = 0
                coeffs[c] = co
                res = rng.integers(-200, 200, n)
                res[rng.random(n) < 0.15] = 0
            d = oracle.alac_desc(np.array([mode]), np.array([order]), np.array([qshift]), np.array([bps if not uncompressed else 32]))
            planes.append(oracle.alac_predict(res.astype(np.int32)[None], d, coeffs[c].astype(np.int32)[None])[0].astype(np.int64))
            words.append(res)
            descs.append(I.Struct("SymaccelAlacDesc", {"mode": I.Int(mode, "u8"), "lpc_order": I.Int(order, "u8"), "shift": I.Int(qshift, "u8"),
                                                         "bps": I.Int(bps if not uncompressed else 32, "u8")}))
        pairs, tails = [], []
        if is_pair and not uncompressed and t % 2 == 0:
            w, s = int(rng.integers(1, 4)), int(rng.integers(1, 5))
            pairs.append(I.Struct("AlacPair", {"plane0": usize(0), "plane1": usize(1), "weight": I.Int(w, "i32"), "shift": I.Int(s, "u8")}))
            left = planes[0] + planes[1] - ((planes[1] * w) >> s)
            planes = [left, left - planes[1]]
        if tail_shift and not uncompressed:
            bits = rng.integers(0, 1 << tail_shift, 2 * n if is_pair else n)
            if is_pair:
                tails.append(I.Struct("AlacTail", {"plane0": usize(0), "plane1": I.some(usize(1)), "shift": I.Int(tail_shift, "u8"),
                                                   "bits": I.Arr([I.Int(int(b), "u16") for b in bits], True)}))
                planes = [(planes[0] << tail_shift) | bits[0::2], (planes[1] << tail_shift) | bits[1::2]]
            else:
                tails.append(I.Struct("AlacTail", {"plane0": usize(0), "plane1": I.NONE, "shift": I.Int(tail_shift, "u8"),
                                                   "bits": I.Arr([I.Int(int(b), "u16") for b in bits], True)}))
                planes = [(planes[0] << tail_shift) | bits, planes[1]]
        script.append(I.Struct("ParsedAlac", {"frames": usize(n), "words": i32_vec(np.array(words)), "desc": I.Arr(descs, True),
                                               "coeffs": i32_vec(coeffs), "pairs": I.Arr(pairs, True), "tails": I.Arr(tails, True),
                                               "out_shift": I.Int(32 - depth, "u32")}))
        want.append(wrap32(np.stack([wrap32(p) for p in planes]) << (32 - depth)))
    return script, want


@pytest.mark.parametrize("make_dll", LIBS)
@pytest.mark.parametrize("depth,frames,max_batch", [(16, 160, 4), (24, 352, 1), (20, 64, 8)])
def test_alac_adapter_restores_the_pcm(make_dll, depth, frames, max_batch):
    """compressed and uncompressed elements, both predictor modes, orders up to 31, mid-side pairs, separately coded low bits and the
    final left-justification through AlacBatch::transform / publish: the device predicts (symaccel_alac_predict), the adapter does
    what follows on the copy out, in the decoder's order (symphonia-codec-alac/src/lib.rs:541-598, 409-414)"""
    h = harness(make_dll, "alac.rs")
    rng = np.random.default_rng(depth * 1000 + frames)
    npk, nch = 9, 2
    script, want = alac_script(rng, depth, frames, npk, nch)
    params = h.params("CODEC_ID_ALAC", 44100, nch, bps=depth)
    front = I.Struct("ScriptedAlacFront", {"params": params, "nch": usize(nch), "max_frames": usize(frames), "script": I.Arr(script, True),
                                            "parses": usize(0)})
    r = h.it.call("HipAlacDecoder::try_new", params, h.opts(), front, usize(max_batch))
    assert r.variant == "Ok", r
    dec = r.f["0"]
    for t in range(npk):
        st, got = h.decode("HipAlacDecoder", dec, h.packet(key(t), frames * t))
        assert st == "ok" and np.array_equal(got, want[t]), (t, got.shape, want[t].shape)
    st, err = h.decode("HipAlacDecoder", dec, h.packet(bytes([255, 0]), 0))
    assert (st, err) == ("err", "DecodeError")
    h.it.call_method("HipAlacDecoder", "reset", dec)
    st, got = h.decode("HipAlacDecoder", dec, h.packet(key(2), 0))
    assert st == "ok" and np.array_equal(got, want[2])
    assert "symaccel_alac_predict" in h.bridge.calls


@pytest.mark.parametrize("make_dll", LIBS)
def test_two_pooled_alac_adapters_share_the_batcher(make_dll):
    """`HipAlacDecoder::try_new_pooled`: two streams (16 and 24 bit, the same frame length) behind look-ahead readers, decoded
    alternately; after each stream's first batch the element channels are written into page-locked slots of ONE `Pool`
    (SYMACCEL_BATCH_ALAC_PREDICT), predicted in common launches, and what follows the predictor (mid/side, low bits,
    left-justification) is applied in place in the slot at publish time."""
    h = harness(make_dll, "alac.rs")
    h.it.load_file(ROOT / "tests" / "rust" / "mocks.rs")
    frames, npk = 160, 13
    decs, readers, wants = [], [], []
    for k, depth in enumerate((16, 24)):
        script, want = alac_script(np.random.default_rng(900 + depth), depth, frames, npk)
        params = h.params("CODEC_ID_ALAC", 44100, 2, bps=depth)
        front = I.Struct("ScriptedAlacFront", {"params": params, "nch": usize(2), "max_frames": usize(frames), "script": I.Arr(script, True),
                                                "parses": usize(0)})
        r = h.it.call("HipAlacDecoder::try_new_pooled", params, h.opts(), front, usize(4))
        assert r.variant == "Ok", r
        decs.append(r.f["0"])
        packets = I.Arr([h.packet(key(t), frames * t, track=7 + k, owned=True) for t in range(npk)], True)
        readers.append(h.it.call("LookaheadReader::new", h.it.call("MockReader::new", packets), usize(12)))
        wants.append(want)
    for t in range(npk):
        for k in range(2):
            r = h.it.call_method("LookaheadReader", "next_packet", readers[k])
            st, got = h.decode("HipAlacDecoder", decs[k], h.it.call_method("Packet", "as_packet_ref", r.f["0"].f["0"]))
            assert st == "ok" and np.array_equal(got, wants[k][t]), (k, t)
    calls = h.bridge.calls
    assert calls.count("symaccel_batcher_create") == 1 and calls.count("symaccel_batcher_reserve") >= 4
    assert "symaccel_batcher_submit" not in calls and "symaccel_batcher_collect" not in calls
    stats = batcher_stats(h)
    assert stats["launches"] < stats["submissions"] and stats["failed_tickets"] == 0, stats


def test_pool_set_device_only_before_the_pool_exists():
    """ctx.rs `Pool::set_device`: one process per GPU picks the shared pool's device before the first decoder; the context is created
    on THAT device (symaccel_ctx_create's first argument), and once the pool exists the call is refused"""
    h = harness(emu_dll, "aac.rs")
    assert h.it.call("Pool::set_device", I.Int(0, "i32")) is True
    n0 = h.bridge.calls.count("symaccel_ctx_create")
    assert h.it.call("Pool::shared").variant == "Ok"
    assert h.bridge.calls.count("symaccel_ctx_create") == n0 + 1 and h.bridge.calls.count("symaccel_batcher_create") == 1
    assert h.it.call("Pool::set_device", I.Int(1, "i32")) is False
    assert h.it.call("Pool::shared").variant == "Ok" and h.bridge.calls.count("symaccel_batcher_create") == 1


def test_register_enters_all_five_decoders_at_the_preferred_tier():
    """lib.rs `register()` EXECUTED: every decoder type of the crate is entered for its codec at Tier::Preferred through
    `register_audio_decoder_at_tier::<D>` (codecs/registry.rs:252-269), above whatever the registry held (remembered: fallback.rs)"""
    h = Harness(emu_dll())
    h.load_shim("lib.rs", "ctx.rs", "decoder.rs", "lookahead.rs", "fallback.rs", "aac.rs", "mpa.rs", "vorbis.rs", "flac.rs", "alac.rs", "frontends.rs")
    h.it.load_file(ROOT / "tests" / "rust" / "registry_generic.rs")
    reg = h.it.call("CodecRegistry::new")
    h.it.call("register", reg)
    pref = I.Enum("Tier", "Preferred")
    for cid in ("CODEC_ID_AAC", "CODEC_ID_MP3", "CODEC_ID_VORBIS", "CODEC_ID_FLAC", "CODEC_ID_ALAC"):
        codec = h.it.resolve_value([cid], I.Env(), None)
        r = h.it.call_method("CodecRegistry", "get_audio_decoder_at_tier", reg, pref, codec)
        assert r.variant == "Some", cid
        assert h.it.call("factory_below", codec).variant == "None"  # (nothing was registered below in this registry)
    # a second register() on the same registry changes nothing (and cannot record the crate's own factories: fallback.rs)
    h.it.call("register", reg)
    assert h.it.call("factory_below", h.it.resolve_value(["CODEC_ID_FLAC"], I.Env(), None)).variant == "None"


def test_the_ffi_bridge_leaves_the_bindings_ctypes_declarations_alone():
    """the interpreter's bridge sets restype / argtypes per call; it must do so on a CDLL object of its own -- the Python binding
    declares argtypes on ITS handle of the same library, and a later call through it with undeclared arguments truncates pointers
    (a serial run of the CPU suite crashed that way: a shim test first, a plain emulation test after it)"""
    from emu_lib import emu_library
    lib = emu_library()
    before = {n: getattr(lib.dll, n).argtypes for n in ("symaccel_aac_synth", "symaccel_ctx_create", "symaccel_host_alloc")}
    assert all(v is not None for v in before.values())
    h = harness(emu_dll, "aac.rs")
    coeffs = np.zeros((1, 2, 1024), np.float32)
    dec, _ = aac_decoder(h, coeffs, np.zeros((2, 1), np.uint8), 2)
    st, _ = h.decode("HipAacDecoder", dec, h.packet(key(0), 0))
    assert st == "ok" and "symaccel_aac_synth" in h.bridge.calls
    assert h.bridge.dll is not lib.dll
    assert {n: getattr(lib.dll, n).argtypes for n in before} == before
