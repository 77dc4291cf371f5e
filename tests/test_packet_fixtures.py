"""Whole-decoder golden vectors (tests/golden/packets/*.npz, tools/make_packet_fixtures.py): `pcm` is what the REFERENCE's own
decoder returns for packets written by tests/aac_writer.py (its whole decoder, executed from /root/reference when the fixture was
made), `coeffs` / `side` what the reference's parse stage hands to its synthesis stage for the same packets.  Here the product does
the synthesis and must give `pcm` back, bit for bit -- no oracle in between, nothing read from /root/reference:

  * through the C ABI (symaccel_aac_synth, one chain per channel, the packets as its frames), in several segmentations;
  * through the Rust shim's HipAacDecoder (run by tools/rsinterp, `extern "C"` bound to the library) with a front end that replays
    the parse results, packet by packet and in look-ahead batches.

`-m gpu`: the hipcc-built library on the MI355X.  Without a GPU the same checks run on the CPU emulation build of the kernel sources.
tests/test_aac_packets.py (localref) is where the fixture's two halves are shown to belong together."""
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
