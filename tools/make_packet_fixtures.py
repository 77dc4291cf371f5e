#!/usr/bin/env python3
"""Generates tests/golden/packets/*.npz (needs /root/reference): whole-decoder golden vectors.

For each stream, packets from the writers under tests/ are decoded by the REFERENCE's own decoder, executed from /root/reference by
tools/rsinterp (tests/test_aac_packets.py describes the set-up): `pcm` is what it returns -- outputs of the reference itself.  The same
packets are parsed by the shim's front end (the patched reference decoder with the recording backend): `coeffs` / `side` are what the
reference hands to its synthesis stage, packet by packet.  tests/test_packet_fixtures.py replays them through libsymaccel -- on the
CPU-emulation build and, `-m gpu`, on the MI355X, where /root/reference does not exist -- and compares with `pcm` bit for bit.

    python tools/make_packet_fixtures.py [aac] [mp3] [vorbis]   # rewrites the files (deterministic: the same arrays every time)
"""
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))

import test_aac_packets as A  # noqa: E402
import test_mp3_packets as M  # noqa: E402
import test_vorbis_packets as V  # noqa: E402
from rs_harness import REF, Harness, patched_tree  # noqa: E402
from rsinterp import interp as I  # noqa: E402

OUT = ROOT / "tests" / "golden" / "packets"
AAC_STREAMS = {"aac_mono": (1, 7, 1), "aac_stereo": (2, 8, 2)}  # name -> (seed, packets, channels): two of test_aac_packets.STREAMS


def aac(name, seed, n_packets, nch, tree):
    packets = [p for p, _ in A.stream(seed, n_packets, nch)]
    ref = Harness(None, reference=True, aac_tree=REF / A.CRATE / "src")
    ref_dec = A.cpu_decoder(ref, nch)
    h, _ = A.hip_decoder(tree, nch)  # (loads the shim; the decoder itself is not used: the front end is driven directly)
    front = h.it.call("aac_front_end", h.params("CODEC_ID_AAC", 44100, nch), h.opts())
    assert front.variant == "Ok", front
    front = h.f32_buffers(front.f["0"])
    pcm, coeffs, side = [], [], []
    for i, pk in enumerate(packets):
        st, planes = ref.decode("AacDecoder", ref_dec, ref.packet(pk, i * 1024))
        assert st == "ok"
        pcm.append(planes)
        r = h.it.call_method("SeamFrontEnd", "parse", front, h.packet(pk, i * 1024))
        assert r.variant == "Ok", r
        parsed = r.f["0"]
        coeffs.append(np.array([np.float32(x) for x in parsed.f["coeffs"].a], np.float32).reshape(nch, 1024))
        side.append(np.array([x.v for x in parsed.f["side"].a], np.uint8))
    lens = np.array([len(p) for p in packets], np.int32)
    np.savez_compressed(OUT / (name + ".npz"), packet_bytes=np.frombuffer(b"".join(packets), np.uint8), packet_lens=lens,
                        coeffs=np.stack(coeffs), side=np.stack(side), pcm=np.stack(pcm).astype(np.float32))
    print(name, "packets", len(packets), "bytes", int(lens.sum()), "peak", float(np.abs(np.stack(pcm)).max()))


MP3_STREAMS = {"mp3_joint": (2, 6, "joint", True, 1), "mp3_lsf": (4, 6, "joint", False, 0)}  # two of test_mp3_packets.ALL_STREAMS


def mp3(name, seed, n, mode, mpeg1, sr_code, tree):
    s, packets = M.stream(seed, n, mode, mpeg1, sr_code)
    per_frame = 1152 if mpeg1 else 576
    ref = Harness(None, reference=True, mp3_tree=REF / M.CRATE / "src")
    ref_dec = M.cpu_decoder(ref, s)
    h = M.shim(tree)
    front = h.it.call("mpa_front_end", h.params("CODEC_ID_MP3", s.rate, s.nch), h.opts())
    assert front.variant == "Ok", front
    front = front.f["0"]
    pcm, xr, side = [], [], []
    for i, (pk, _) in enumerate(packets):
        st, planes = ref.decode("MpaDecoder", ref_dec, ref.packet(pk, i * per_frame))
        assert st == "ok"
        pcm.append(planes)
        r = h.it.call_method("SeamFrontEnd", "parse", front, h.packet(pk, i * per_frame))
        assert r.variant == "Ok", r
        parsed = r.f["0"]
        ngr = parsed.f["n_granules"].v
        xr.append(np.array([np.float32(x) for x in parsed.f["xr"].a], np.float32).reshape(ngr, s.nch, 576))      # [granule][channel][576]
        side.append(np.array([[q.f["block_type"].v, q.f["is_mixed"].v, q.f["rzero"].v] for q in parsed.f["side"].a], np.int32).reshape(ngr, s.nch, 3))
    data = [p for p, _ in packets]
    lens = np.array([len(p) for p in data], np.int32)
    np.savez_compressed(OUT / (name + ".npz"), packet_bytes=np.frombuffer(b"".join(data), np.uint8), packet_lens=lens, xr=np.stack(xr),
                        side=np.stack(side), pcm=np.stack(pcm).astype(np.float32), sample_rate_idx=np.array([s.sr_idx], np.int32),
                        sample_rate=np.array([s.rate], np.int32))
    print(name, "packets", len(data), "bytes", int(lens.sum()), "peak", float(np.abs(np.stack(pcm)).max()))


# the second stream has config 4's block sizes (256 / 2048)
VORBIS_STREAMS = {"vorbis_small": (1, 9, 2, 6, 9, (2, 1), True), "vorbis_256_2048": (4, 7, 2, 8, 11, (2, 2), True)}


def vorbis(name, args, tree):
    s, packets = V.stream(*args)
    ref = Harness(None, reference=True, vorbis_tree=REF / V.CRATE / "src")
    ref_dec = V.cpu_decoder(ref, s)
    h, _ = V.hip_decoder(tree, s, max_batch=1)
    front = h.it.call("vorbis_front_end", h.params("CODEC_ID_VORBIS", 44100, s.nch, extra=s.extra_data()), h.opts())
    assert front.variant == "Ok", front
    front = front.f["0"]
    flags, spectra, pcm, counts = [], [], [], []
    for pk, _ in packets:
        st, planes = ref.decode("VorbisDecoder", ref_dec, ref.packet(pk, 0))
        assert st == "ok"
        pcm.append(planes.astype(np.float32).reshape(s.nch, -1))
        counts.append(planes.shape[1] if planes.size else 0)
        r = h.it.call_method("SeamFrontEnd", "parse", front, h.packet(pk, 0))
        assert r.variant == "Ok", r
        parsed = r.f["0"]
        flags.append(int(bool(parsed.f["long_block"])))
        spectra.append(np.array([np.float32(x) for x in parsed.f["spectra"].a], np.float32).reshape(s.nch, -1))
    data = [p for p, _ in packets]
    lens = np.array([len(p) for p in data], np.int32)
    np.savez_compressed(OUT / (name + ".npz"), packet_bytes=np.frombuffer(b"".join(data), np.uint8), packet_lens=lens,
                        long_block=np.array(flags, np.uint8), spectra=np.concatenate(spectra, axis=1), pcm=np.concatenate(pcm, axis=1),
                        frames=np.array(counts, np.int32), block_exps=np.array([s.bs0_exp, s.bs1_exp], np.int32))
    print(name, "packets", len(data), "bytes", int(lens.sum()), "frames", counts, "peak", float(np.abs(np.concatenate(pcm, axis=1)).max()))


def main():
    OUT.mkdir(parents=True, exist_ok=True)
    which = set(sys.argv[1:]) or {"aac", "mp3", "vorbis"}
    if "aac" in which:
        tree = patched_tree((A.CRATE,)) / A.CRATE / "src"
        for name, (seed, n, nch) in AAC_STREAMS.items():
            aac(name, seed, n, nch, tree)
    if "mp3" in which:
        tree = patched_tree((M.CRATE,)) / M.CRATE / "src"
        for name, args in MP3_STREAMS.items():
            mp3(name, *args, tree)
    if "vorbis" in which:
        tree = patched_tree((V.CRATE,)) / V.CRATE / "src"
        for name, args in VORBIS_STREAMS.items():
            vorbis(name, args, tree)


if __name__ == "__main__":
    main()
