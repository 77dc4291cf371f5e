"""Whole-decoder golden vectors (tests/golden/packets/*.npz, tools/make_packet_fixtures.py): `pcm` is what the REFERENCE's own
decoder returns for packets written by tests/aac_writer.py (its whole decoder, executed from /root/reference when the fixture was
made), `coeffs` / `side` what the reference's parse stage hands to its synthesis stage for the same packets.  Here the product does
the synthesis and must give `pcm` back, bit for bit -- no oracle in between, nothing read from /root/reference:

  * through the C ABI (symaccel_aac_synth / symaccel_mp3_synth / symaccel_vorbis_synth, one chain per channel, the packets as its frames / granules), in
    several segmentations;
  * through the Rust shim's HipAacDecoder / HipMpaDecoder / HipVorbisDecoder (run by tools/rsinterp, `extern "C"` bound to the library) with a front
    end that replays the parse results, packet by packet and in look-ahead batches.

FLAC / ALAC (integer paths): 16- and 24-bit FLAC frames of every subframe type and channel assignment, 24-bit stereo and 20-bit 6-channel ALAC
packets (tests/flac_writer.py, tests/alac_writer.py), through the shim decoders.  Vorbis: two stereo streams (64 / 512 and 256 / 2048 -- config 4's block sizes), every block-size transition, residue types 1 and 2,
coupling (tests/vorbis_writer.py).  AAC-LC: mono and stereo streams with every tool (tests/aac_writer.py).  MP3: an MPEG-1 joint-stereo stream at 48 kHz and an MPEG-2
(LSF) joint-stereo stream at 22.05 kHz, variable bit rate, all block types, main data through the bit reservoir (tests/mp3_writer.py).

`-m gpu`: the hipcc-built library on the MI355X.  Without a GPU the same checks run on the CPU emulation build of the kernel sources.
tests/test_aac_packets.py / test_mp3_packets.py (localref) are where the fixtures' two halves are shown to belong together."""
from pathlib import Path

import numpy as np
import pytest

from emu_lib import emu_ctx  # noqa: F401
from test_product_vs_reference_text import Emu, Gpu, gpu_ctx  # noqa: F401
from test_rust_adapters import LIBS, aac_decoder, harness, key

PACKETS = Path(__file__).resolve().parent / "golden" / "packets"
AAC = ["aac_mono", "aac_stereo"]


def bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


def load(name):
    f = np.load(PACKETS / (name + ".npz"))
    coeffs, side, pcm = f["coeffs"], f["side"], f["pcm"]       # [packet][channel][1024], [packet][channel], [packet][channel][1024]
    assert coeffs.dtype == np.float32 and pcm.dtype == np.float32 and side.dtype == np.uint8
    assert int(f["packet_lens"].sum()) == f["packet_bytes"].size and len(f["packet_lens"]) == len(pcm)
    return coeffs, side, pcm


def check_c_abi(r, name, seg):
    from symphonia_amd import AacDsp
    coeffs, side, pcm = load(name)
    chains = np.ascontiguousarray(coeffs.transpose(1, 0, 2))    # [channel][packet][1024]: a channel's packets are one chain
    sides = np.ascontiguousarray(side.T)
    zero = np.zeros((chains.shape[0], 1024), np.float32)        # a fresh decoder's delay lines (ics/mod.rs:209)
    r.ctx.set_segment(seg)
    if isinstance(r, Emu):
        got, _ = AacDsp(r.ctx).synth(chains, sides, zero)
    else:
        got = r.host(AacDsp(r.ctx).synth(r.dev(chains), r.dev(sides), r.dev(zero)))
    want = pcm.transpose(1, 0, 2)
    assert np.array_equal(bits(got), bits(want)), (name, seg, float(np.abs(got - want).max()))
    # the stream in two calls, the delay lines carried by the caller: what a decoder that batches fewer packets does
    cut = 3
    if isinstance(r, Emu):
        a, d = AacDsp(r.ctx).synth(chains[:, :cut], sides[:, :cut], zero)
        b, _ = AacDsp(r.ctx).synth(chains[:, cut:], sides[:, cut:], d)
    else:
        d = r.dev(zero)
        a = r.host(AacDsp(r.ctx).synth(r.dev(chains[:, :cut]), r.dev(sides[:, :cut]), d))
        b = r.host(AacDsp(r.ctx).synth(r.dev(chains[:, cut:]), r.dev(sides[:, cut:]), d))
    assert np.array_equal(bits(np.concatenate([a, b], axis=1)), bits(want)), (name, seg)


@pytest.mark.parametrize("seg", [0, 1, 2, 5])
@pytest.mark.parametrize("name", AAC)
def test_emulated_c_abi_gives_the_reference_decoders_pcm(emu_ctx, name, seg):  # noqa: F811
    check_c_abi(Emu(emu_ctx), name, seg)


@pytest.mark.gpu
@pytest.mark.parametrize("seg", [0, 1, 2, 5])
@pytest.mark.parametrize("name", AAC)
def test_gpu_c_abi_gives_the_reference_decoders_pcm(gpu_ctx, name, seg):  # noqa: F811
    check_c_abi(Gpu(gpu_ctx), name, seg)


@pytest.mark.parametrize("make_dll", LIBS)
@pytest.mark.parametrize("name,max_batch", [("aac_mono", 1), ("aac_stereo", 3), ("aac_stereo", 16)])
def test_the_shim_decoder_gives_the_reference_decoders_pcm(make_dll, name, max_batch):
    from rs_harness import usize
    from rsinterp import interp as I
    coeffs, side, pcm = load(name)
    h = harness(make_dll, "aac.rs")
    h.it.load_file(Path(__file__).resolve().parent / "rust" / "mocks.rs")
    dec, front = aac_decoder(h, np.ascontiguousarray(coeffs.transpose(1, 0, 2)), side, max_batch)
    n = len(pcm)
    packets = I.Arr([h.packet(key(t), 1024 * t, track=2, owned=True) for t in range(n)], True)
    reader = h.it.call("LookaheadReader::new", h.it.call("MockReader::new", packets), usize(16))
    calls0 = h.bridge.calls.count("symaccel_aac_synth")
    for t in range(n):
        p = h.it.call_method("LookaheadReader", "next_packet", reader).f["0"].f["0"]
        st, got = h.decode("HipAacDecoder", dec, h.it.call_method("Packet", "as_packet_ref", p))
        assert st == "ok" and np.array_equal(bits(got), bits(pcm[t])), (name, t)
    assert h.bridge.calls.count("symaccel_aac_synth") - calls0 == -(-n // max_batch)
    assert front.f["parses"].v == n


# ------------------------------------------------------------------------------------------------ MP3

MP3 = ["mp3_joint", "mp3_lsf"]
MP3_SIDE = np.dtype([("block_type", np.uint8), ("is_mixed", np.uint8), ("rzero", "<u2")])  # symaccel_mp3_side


def load_mp3(name):
    f = np.load(PACKETS / (name + ".npz"))
    xr, side, pcm = f["xr"], f["side"], f["pcm"]    # [packet][granule][channel][576], [packet][granule][channel][3], [packet][channel][576 g]
    assert xr.dtype == np.float32 and pcm.dtype == np.float32 and len(xr) == len(pcm) == len(f["packet_lens"])
    return xr, side, pcm, int(f["sample_rate_idx"][0]), int(f["sample_rate"][0])


def check_mp3_c_abi(r, name, seg):
    from symphonia_amd import Mp3Synthesis
    xr, side, pcm, sr_idx, _ = load_mp3(name)
    npk, ngr, nch = xr.shape[:3]
    chains = np.ascontiguousarray(xr.transpose(2, 0, 1, 3).reshape(nch, npk * ngr, 576))   # a channel's granules are one chain
    rec = np.zeros((nch, npk * ngr), MP3_SIDE)
    s = side.transpose(2, 0, 1, 3).reshape(nch, npk * ngr, 3)
    rec["block_type"], rec["is_mixed"], rec["rzero"] = s[..., 0], s[..., 1], s[..., 2]
    want = pcm.reshape(npk, nch, ngr, 576).transpose(1, 0, 2, 3).reshape(nch, npk * ngr, 576)
    zero = (np.zeros((nch, 576), np.float32), np.zeros((nch, 1024), np.float32), np.zeros(nch, np.int32))  # a fresh Layer3 (mod.rs:226-233)
    r.ctx.set_segment(seg)
    if isinstance(r, Emu):
        got = Mp3Synthesis(r.ctx, sr_idx).synth(chains, rec, *zero)[0]
    else:
        got = r.host(Mp3Synthesis(r.ctx, sr_idx).synth(r.dev(chains), r.dev(rec), *[r.dev(z) for z in zero]))
    assert np.array_equal(bits(got), bits(want)), (name, seg, float(np.abs(got - want).max()))


@pytest.mark.parametrize("seg", [0, 1, 3])
@pytest.mark.parametrize("name", MP3)
def test_emulated_c_abi_gives_the_reference_mp3_decoders_pcm(emu_ctx, name, seg):  # noqa: F811
    check_mp3_c_abi(Emu(emu_ctx), name, seg)


@pytest.mark.gpu
@pytest.mark.parametrize("seg", [0, 1, 3])
@pytest.mark.parametrize("name", MP3)
def test_gpu_c_abi_gives_the_reference_mp3_decoders_pcm(gpu_ctx, name, seg):  # noqa: F811
    check_mp3_c_abi(Gpu(gpu_ctx), name, seg)


@pytest.mark.parametrize("make_dll", LIBS)
@pytest.mark.parametrize("name,max_batch", [("mp3_joint", 4), ("mp3_lsf", 1)])
def test_the_shim_mp3_decoder_gives_the_reference_decoders_pcm(make_dll, name, max_batch):
    from rs_harness import f32_vec, usize
    from rsinterp import interp as I
    xr, side, pcm, sr_idx, rate = load_mp3(name)
    npk, ngr, nch = xr.shape[:3]
    h = harness(make_dll, "mpa.rs")
    h.it.load_file(Path(__file__).resolve().parent / "rust" / "mocks.rs")

    def parsed(p):
        rows = [I.Struct("SymaccelMp3Side", {"block_type": I.Int(int(side[p, g, c, 0]), "u8"), "is_mixed": I.Int(int(side[p, g, c, 1]), "u8"),
                                             "rzero": I.Int(int(side[p, g, c, 2]), "u16")}) for g in range(ngr) for c in range(nch)]
        return I.Struct("ParsedMpa", {"trim": (usize(0), usize(0)), "n_granules": usize(ngr), "xr": f32_vec(xr[p]), "side": I.Arr(rows, True), "fused": I.NONE})

    front = I.Struct("ScriptedMpaFront", {"nch": usize(nch), "sr_idx": I.Int(sr_idx, "i32"), "script": I.Arr([parsed(p) for p in range(npk)], True),
                                          "parses": usize(0), "resets": usize(0)})
    r = h.it.call("HipMpaDecoder::try_new", h.params("CODEC_ID_MP3", rate, nch), h.opts(), front, usize(max_batch))
    assert r.variant == "Ok", r
    dec = r.f["0"]
    per = 576 * ngr
    packets = I.Arr([h.packet(key(t), per * t, track=2, owned=True) for t in range(npk)], True)
    reader = h.it.call("LookaheadReader::new", h.it.call("MockReader::new", packets), usize(16))
    calls0 = h.bridge.calls.count("symaccel_mp3_synth")
    for t in range(npk):
        p = h.it.call_method("LookaheadReader", "next_packet", reader).f["0"].f["0"]
        st, got = h.decode("HipMpaDecoder", dec, h.it.call_method("Packet", "as_packet_ref", p))
        assert st == "ok" and np.array_equal(bits(got), bits(pcm[t])), (name, t)
    assert h.bridge.calls.count("symaccel_mp3_synth") - calls0 == -(-npk // max_batch)


# ------------------------------------------------------------------------------------------------ Vorbis

VORBIS = ["vorbis_small", "vorbis_256_2048"]


def load_vorbis(name):
    f = np.load(PACKETS / (name + ".npz"))
    flags, spectra, pcm, frames = f["long_block"], f["spectra"], f["pcm"], f["frames"]   # [packet], [channel][sum n/2], [channel][sum frames]
    b0, b1 = (int(x) for x in f["block_exps"])
    n2 = np.where(flags > 0, 1 << b1, 1 << b0) // 2
    assert spectra.shape[1] == n2.sum() and pcm.shape[1] == frames.sum() and frames[0] == 0
    return flags, spectra, pcm, frames, b0, b1, n2


def check_vorbis_c_abi(r, name, seg):
    from symphonia_amd import VorbisDsp
    flags, spectra, pcm, frames, b0, b1, n2 = load_vorbis(name)
    nch, nb = spectra.shape[0], len(flags)
    dsp = VorbisDsp(r.ctx, b0, b1)
    bf = np.tile(flags.astype(np.uint8), (nch, 1))
    so, po = dsp.layout(bf, np.full(nch, -1))
    assert so[0, -1] == spectra.shape[1]
    # packet 0 primes the overlap: the reference (gapless) returns no samples for it (lib.rs:333-336), the batched call computes the
    # block against a zero overlap in its place; from packet 1 on the outputs are the reference's
    assert np.array_equal(np.diff(po[0])[1:], frames[1:])
    zero = np.zeros((nch, (1 << b1) >> 1), np.float32)
    r.ctx.set_segment(seg)
    if isinstance(r, Emu):
        got = dsp.synth(np.ascontiguousarray(spectra), bf, np.full(nch, -1, np.int32), zero, int(po[0, -1]))[0]
    else:
        got = r.host(dsp.synth(r.dev(np.ascontiguousarray(spectra)), r.dev(bf), r.dev(np.full(nch, -1, np.int32)), r.dev(zero), int(po[0, -1])))
    assert np.array_equal(bits(got[:, po[0, 1]:]), bits(pcm)), (name, seg)


@pytest.mark.parametrize("seg", [0, 1, 2])
@pytest.mark.parametrize("name", VORBIS)
def test_emulated_c_abi_gives_the_reference_vorbis_decoders_pcm(emu_ctx, name, seg):  # noqa: F811
    check_vorbis_c_abi(Emu(emu_ctx), name, seg)


@pytest.mark.gpu
@pytest.mark.parametrize("seg", [0, 1, 2])
@pytest.mark.parametrize("name", VORBIS)
def test_gpu_c_abi_gives_the_reference_vorbis_decoders_pcm(gpu_ctx, name, seg):  # noqa: F811
    check_vorbis_c_abi(Gpu(gpu_ctx), name, seg)


@pytest.mark.parametrize("make_dll", LIBS)
@pytest.mark.parametrize("name,max_batch", [("vorbis_small", 4), ("vorbis_256_2048", 2)])
def test_the_shim_vorbis_decoder_gives_the_reference_decoders_pcm(make_dll, name, max_batch):
    from rs_harness import f32_vec, usize
    from rsinterp import interp as I
    flags, spectra, pcm, frames, b0, b1, n2 = load_vorbis(name)
    nch, nb = spectra.shape[0], len(flags)
    h = harness(make_dll, "vorbis.rs")
    h.it.load_file(Path(__file__).resolve().parent / "rust" / "mocks.rs")
    so = np.concatenate([[0], np.cumsum(n2)])
    script = I.Arr([I.Struct("ParsedVorbis", {"trim": (usize(0), usize(0)), "long_block": bool(flags[b]), "spectra": f32_vec(spectra[:, so[b]:so[b + 1]]), "fused": I.NONE})
                    for b in range(nb)], True)
    front = I.Struct("ScriptedVorbisFront", {"nch": usize(nch), "bs0_exp": I.Int(b0, "i32"), "bs1_exp": I.Int(b1, "i32"), "script": script, "parses": usize(0), "resets": usize(0)})
    r = h.it.call("HipVorbisDecoder::try_new", h.params("CODEC_ID_VORBIS", 44100, nch), h.opts(), front, usize(max_batch))
    assert r.variant == "Ok", r
    dec = r.f["0"]
    packets = I.Arr([h.packet(key(t), t, track=2, owned=True) for t in range(nb)], True)
    reader = h.it.call("LookaheadReader::new", h.it.call("MockReader::new", packets), usize(16))
    po = np.concatenate([[0], np.cumsum(frames)])
    for t in range(nb):
        p = h.it.call_method("LookaheadReader", "next_packet", reader).f["0"].f["0"]
        st, got = h.decode("HipVorbisDecoder", dec, h.it.call_method("Packet", "as_packet_ref", p))
        assert st == "ok" and got.shape == (nch, frames[t]) and np.array_equal(bits(got), bits(pcm[:, po[t]:po[t + 1]])), (name, t)


# ------------------------------------------------------------------------------------------------ FLAC, ALAC (through the shim decoders)

@pytest.mark.parametrize("make_dll", LIBS)
@pytest.mark.parametrize("name,max_batch", [("flac_24bit", 2), ("flac_16bit", 8)])
def test_the_shim_flac_decoder_gives_the_reference_decoders_pcm(make_dll, name, max_batch):
    from rs_harness import i32_vec, usize
    from rsinterp import interp as I
    f = np.load(PACKETS / (name + ".npz"))
    pcm, words, desc, coeffs = f["pcm"], f["words"], f["desc"], f["coeffs"]     # [frame][channel][blocksize], ..., [frame][channel][4], [frame][channel][32]
    nfr, nch, bs = words.shape
    bps = int(f["bps"][0])
    h = harness(make_dll, "flac.rs")
    h.it.load_file(Path(__file__).resolve().parent / "rust" / "mocks.rs")
    script = []
    for t_ in range(nfr):
        descs = [I.Struct("SymaccelFlacDesc", {"kind": I.Int(int(d[0]), "u8"), "order": I.Int(int(d[1]), "u8"), "shift": I.Int(int(d[2]), "u8"),
                                              "wasted_bits": I.Int(int(d[3]), "u8")}) for d in desc[t_]]
        script.append(I.Struct("ParsedFlac", {"blocksize": usize(bs), "words": i32_vec(words[t_]), "desc": I.Arr(descs, True), "coeffs": i32_vec(coeffs[t_]),
                                               "pair_mode": I.Int(int(f["pair_mode"][t_]), "u8"), "out_shift": I.Int(int(f["out_shift"][t_]), "u32")}))
    params = h.params("CODEC_ID_FLAC", 44100, nch, bps=bps)
    front = I.Struct("ScriptedFlacFront", {"params": params, "nch": usize(nch), "max_bs": usize(bs), "script": I.Arr(script, True), "parses": usize(0)})
    r = h.it.call("HipFlacDecoder::try_new", params, h.opts(), front, usize(max_batch))
    assert r.variant == "Ok", r
    dec = r.f["0"]
    packets = I.Arr([h.packet(key(t_), bs * t_, track=2, owned=True) for t_ in range(nfr)], True)
    reader = h.it.call("LookaheadReader::new", h.it.call("MockReader::new", packets), usize(16))
    for t_ in range(nfr):
        p = h.it.call_method("LookaheadReader", "next_packet", reader).f["0"].f["0"]
        st, got = h.decode("HipFlacDecoder", dec, h.it.call_method("Packet", "as_packet_ref", p))
        assert st == "ok" and np.array_equal(got, pcm[t_]), (name, t_)
    assert "symaccel_flac_restore" in h.bridge.calls


@pytest.mark.parametrize("make_dll", LIBS)
@pytest.mark.parametrize("name,max_batch", [("alac_24bit_stereo", 3), ("alac_20bit_6ch", 1)])
def test_the_shim_alac_decoder_gives_the_reference_decoders_pcm(make_dll, name, max_batch):
    from rs_harness import i32_vec, usize
    from rsinterp import interp as I
    f = np.load(PACKETS / (name + ".npz"))
    pcm, words, desc, coeffs, pairs, tails, tail_bits = f["pcm"], f["words"], f["desc"], f["coeffs"], f["pairs"], f["tails"], f["tail_bits"]
    npk, nch, n = words.shape
    depth = int(f["depth"][0])
    h = harness(make_dll, "alac.rs")
    h.it.load_file(Path(__file__).resolve().parent / "rust" / "mocks.rs")
    script = []
    for t_ in range(npk):
        descs = [I.Struct("SymaccelAlacDesc", {"mode": I.Int(int(d[0]), "u8"), "lpc_order": I.Int(int(d[1]), "u8"), "shift": I.Int(int(d[2]), "u8"),
                                              "bps": I.Int(int(d[3]), "u8")}) for d in desc[t_]]
        prs = [I.Struct("AlacPair", {"plane0": usize(q[1]), "plane1": usize(q[2]), "weight": I.Int(int(q[3]), "i32"), "shift": I.Int(int(q[4]), "u8")})
               for q in pairs if q[0] == t_]
        tls = [I.Struct("AlacTail", {"plane0": usize(q[1]), "plane1": I.some(usize(q[2])) if q[2] >= 0 else I.NONE, "shift": I.Int(int(q[3]), "u8"),
                                     "bits": I.Arr([I.Int(int(b), "u16") for b in tail_bits[q[4]:q[4] + q[5]]], True)}) for q in tails if q[0] == t_]
        script.append(I.Struct("ParsedAlac", {"frames": usize(n), "words": i32_vec(words[t_]), "desc": I.Arr(descs, True), "coeffs": i32_vec(coeffs[t_]),
                                               "pairs": I.Arr(prs, True), "tails": I.Arr(tls, True), "out_shift": I.Int(int(f["out_shift"][t_]), "u32")}))
    params = h.params("CODEC_ID_ALAC", 44100, nch, bps=depth)
    front = I.Struct("ScriptedAlacFront", {"params": params, "nch": usize(nch), "max_frames": usize(n), "script": I.Arr(script, True), "parses": usize(0)})
    r = h.it.call("HipAlacDecoder::try_new", params, h.opts(), front, usize(max_batch))
    assert r.variant == "Ok", r
    dec = r.f["0"]
    packets = I.Arr([h.packet(key(t_), n * t_, track=2, owned=True) for t_ in range(npk)], True)
    reader = h.it.call("LookaheadReader::new", h.it.call("MockReader::new", packets), usize(16))
    for t_ in range(npk):
        p = h.it.call_method("LookaheadReader", "next_packet", reader).f["0"].f["0"]
        st, got = h.decode("HipAlacDecoder", dec, h.it.call_method("Packet", "as_packet_ref", p))
        assert st == "ok" and np.array_equal(got, pcm[t_]), (name, t_)
    assert "symaccel_alac_predict" in h.bridge.calls


# ------------------------------------------------------------------------------------------------ the second-generation seams
# <name>_fused.npz: the same packets, the same reference `pcm`, but what the reference's decoders hand over ONE STAGE EARLIER -- the
# entropy decoder's integers + side records (MP3), the spectrum decoder's coefficients + joint-stereo descriptors + TNS filters (AAC),
# residue vectors + floor posts + coupling steps (Vorbis).  The host-pointer entry points the shim decoders call
# (symaccel_mp3_decode_pipelined, symaccel_aac_decode_pipelined, symaccel_vorbis_decode) must give `pcm` back, bit for bit.

def lib_ctx(r):
    return r.ctx


def check_aac_fused(ctx, name, chunk):
    from symphonia_amd import AAC_JS_DTYPE, AAC_TNS_DTYPE, AacSpectralTools
    f = np.load(PACKETS / (name + "_fused.npz"))
    coded, side, pcm = f["coded"], f["side"], f["pcm"]           # [packet][channel][1024], [packet][channel], [packet][channel][1024]
    npk, nch = coded.shape[:2]
    chains = np.ascontiguousarray(coded.transpose(1, 0, 2))
    sides = np.ascontiguousarray(side.T)
    lefts = sorted(set(int(x) for x in f["joint_where"][:, 1]))
    pairs = np.array([[l, l + 1] for l in lefts], np.int32).reshape(-1, 2)
    desc = np.zeros((len(lefts), npk), AAC_JS_DTYPE)
    desc["num_windows"] = 1
    for (pk, left), d in zip(f["joint_where"], f["joint_desc"].view(AAC_JS_DTYPE).reshape(-1)):
        desc[lefts.index(int(left)), pk] = d
    tns = f["tns"].view(AAC_TNS_DTYPE).reshape(-1).copy()
    tns["frame"] = f["tns_where"][:, 1].astype(np.uint32) * npk + f["tns_where"][:, 0].astype(np.uint32)  # [channel][packet]
    swb_long = f["swb_long"] if f["swb_long"].size else np.array([0, 1024], np.uint16)
    swb_short = f["swb_short"] if f["swb_short"].size else np.array([0, 128], np.uint16)
    got, delay = np.zeros_like(chains), np.zeros((nch, 1024), np.float32)
    AacSpectralTools(ctx, swb_long, swb_short).decode(chains, sides, delay, pairs if len(lefts) else None, desc if len(lefts) else None,
                                                      tns if len(tns) else None, got, chunk_frames=chunk)
    want = pcm.transpose(1, 0, 2)
    assert np.array_equal(bits(got), bits(want)), (name, chunk, float(np.abs(got - want).max()))
    return len(lefts), len(tns)


def check_mp3_fused(ctx, name, chunk):
    f = np.load(PACKETS / (name + "_fused.npz"))
    quant, pcm, sr_idx = f["quant"], f["pcm"], int(f["sample_rate_idx"][0])   # [packet][granule][channel][576]
    npk, ngr, nch = quant.shape[:3]
    to_chain = lambda a: np.ascontiguousarray(a.transpose(2, 0, 1, *range(3, a.ndim)).reshape((nch, npk * ngr) + a.shape[3:]))  # noqa: E731
    q, rq = to_chain(quant), to_chain(f["rq"])
    st = np.ascontiguousarray(f["st"].reshape(1, npk * ngr, 48))
    s = to_chain(f["side"])
    side = np.zeros((nch, npk * ngr), MP3_SIDE)
    side["block_type"], side["is_mixed"], side["rzero"] = s[..., 0], s[..., 1], s[..., 2]
    want = pcm.reshape(npk, nch, ngr, 576).transpose(1, 0, 2, 3).reshape(nch, npk * ngr, 576)
    ov, vv, vf = np.zeros((nch, 576), np.float32), np.zeros((nch, 1024), np.float32), np.zeros(nch, np.int32)
    got = np.zeros((nch, npk * ngr, 576), np.float32)
    pairs = np.array([[0, 1]], np.int32)
    ctx._call(ctx.lib.dll.symaccel_mp3_decode_pipelined, q.ctypes.data, rq.ctypes.data, pairs.ctypes.data if nch == 2 else None,
              st.ctypes.data if nch == 2 else None, 1 if nch == 2 else 0, side.ctypes.data, sr_idx, ov.ctypes.data, vv.ctypes.data, vf.ctypes.data,
              got.ctypes.data, nch, npk * ngr, chunk)
    assert np.array_equal(bits(got), bits(want)), (name, chunk, float(np.abs(got - want).max()))


def check_vorbis_fused(ctx, name):
    from symphonia_amd import VORBIS_FLOOR1_DTYPE, VorbisDsp
    f = np.load(PACKETS / (name + "_fused.npz"))
    flags, residue, pcm = f["long_block"], f["residue"], f["pcm"]
    b0, b1 = (int(x) for x in f["block_exps"])
    nch, nb = residue.shape[0], len(flags)
    dsp = VorbisDsp(ctx, b0, b1)
    bf = np.tile(flags.astype(np.uint8), (nch, 1))
    so, po = dsp.layout(bf, np.full(nch, -1))
    spec_stride, pcm_stride = (int(so[0, -1]) + 3) & ~3, (int(po[0, -1]) + 3) & ~3
    res = np.zeros((nch, spec_stride), np.float32)
    res[:, :residue.shape[1]] = residue
    floor = np.ascontiguousarray(f["floor"].T)                     # [channel][packet]
    posts = np.ascontiguousarray(f["posts"].transpose(1, 0, 2))    # [channel][packet][65]
    floors = np.ascontiguousarray(f["floors"]).view(VORBIS_FLOOR1_DTYPE).reshape(-1)
    prev, ov = np.full(nch, -1, np.int32), np.zeros((nch, (1 << b1) >> 1), np.float32)
    got = np.zeros((nch, pcm_stride), np.float32)
    dsp.decode(res, bf, floor, posts, floors, nch, np.ascontiguousarray(f["coupling"]), np.ascontiguousarray(f["coupling_first"]), prev, ov,
               pcm_stride, got)
    # (packet 0 primes the overlap: check_vorbis_c_abi)
    assert np.array_equal(bits(got[:, po[0, 1]:po[0, -1]]), bits(pcm)), name
    return int(f["coupling_first"][-1]), int((floor == 255).sum())


@pytest.mark.parametrize("name", AAC)
def test_emulated_fused_entry_point_gives_the_reference_aac_decoders_pcm(emu_ctx, name):  # noqa: F811
    for chunk in (0, 3):
        n_pairs, n_tns = check_aac_fused(emu_ctx, name, chunk)
    assert n_tns >= 1 and (n_pairs >= 1 or name == "aac_mono")  # (the fixtures exercise what the entry point is for)


@pytest.mark.parametrize("name", MP3)
def test_emulated_fused_entry_point_gives_the_reference_mp3_decoders_pcm(emu_ctx, name):  # noqa: F811
    for chunk in (0, 4):
        check_mp3_fused(emu_ctx, name, chunk)


@pytest.mark.parametrize("name", VORBIS)
def test_emulated_fused_entry_point_gives_the_reference_vorbis_decoders_pcm(emu_ctx, name):  # noqa: F811
    steps, unused = check_vorbis_fused(emu_ctx, name)
    assert steps >= 1 and unused >= 1


@pytest.mark.gpu
@pytest.mark.parametrize("name", AAC + MP3 + VORBIS)
def test_gpu_fused_entry_points_give_the_reference_decoders_pcm(name):
    from symphonia_amd import Context
    ctx = Context(0)
    try:
        if name in AAC:
            for chunk in (0, 3):
                check_aac_fused(ctx, name, chunk)
        elif name in MP3:
            for chunk in (0, 4):
                check_mp3_fused(ctx, name, chunk)
        else:
            check_vorbis_fused(ctx, name)
    finally:
        ctx.close()
