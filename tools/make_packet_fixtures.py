#!/usr/bin/env python3
"""Generates tests/golden/packets/*.npz (needs /root/reference): whole-decoder golden vectors.

For each stream, packets from the writers under tests/ are decoded by the REFERENCE's own decoder, executed from /root/reference by
tools/rsinterp (tests/test_aac_packets.py describes the set-up): `pcm` is what it returns -- outputs of the reference itself.  The same
packets are parsed by the shim's front end (the patched reference decoder with the recording backend): `coeffs` / `side` are what the
reference hands to its synthesis stage, packet by packet.  tests/test_packet_fixtures.py replays them through libsymaccel -- on the
CPU-emulation build and, `-m gpu`, on the MI355X, where /root/reference does not exist -- and compares with `pcm` bit for bit.

    python tools/make_packet_fixtures.py [aac] [mp3] [vorbis] [flac] [alac]   # rewrites the files (deterministic: the same arrays every time)
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
import test_flac_packets as F  # noqa: E402
import test_alac_packets as L  # noqa: E402
import flac_writer  # noqa: E402
from rs_harness import REF, Harness, patched_tree  # noqa: E402
from rsinterp import interp as I  # noqa: E402

OUT = ROOT / "tests" / "golden" / "packets"
AAC_STREAMS = {"aac_mono": (1, 7, 1), "aac_stereo": (2, 8, 2)}  # name -> (seed, packets, channels): two of test_aac_packets.STREAMS


def aac(name, seed, n_packets, nch, tree):
    packets = [p for p, _ in A.stream(seed, n_packets, nch)]
    ref = Harness(None, reference=True, aac_tree=REF / A.CRATE / "src")
    ref_dec = A.cpu_decoder(ref, nch)
    h, _ = A.hip_decoder(tree, nch)  # (loads the shim; the decoder itself is not used: the front end is driven directly)
    front = h.it.call("SeamFrontEnd::try_new_at", h.params("CODEC_ID_AAC", 44100, nch), h.opts(), False)  # the first-generation seam
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
    front = h.it.call("SeamFrontEnd::try_new_at", h.params("CODEC_ID_MP3", s.rate, s.nch), h.opts(), False)  # the first-generation seam
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
    front = h.it.call("SeamFrontEnd::try_new_at", h.params("CODEC_ID_VORBIS", 44100, s.nch, extra=s.extra_data()), h.opts(), False)  # first generation
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


def ints(arr):
    return np.array([x.v for x in arr.a], np.int64)


# ---------------------------------------------------------------------------------------------------------------------------------
# The SECOND-generation seams (the decoders hand over what the entropy / spectrum / packet decoder produced; requantize + stereo,
# joint stereo + TNS, coupling + floor + dot product run on the device): <name>_fused.npz beside <name>.npz, same packets, same `pcm`.

def records(values, dtype):
    """interpreter structs (#[repr(C)] records of bindings/rust/symaccel_sys.rs) -> a numpy structured array of the ABI's layout"""
    out = np.zeros(len(values), dtype)
    for i, v in enumerate(values):
        v = I.deref(v)
        for name in dtype.names:
            f = I.deref(v.f[name])
            if dtype[name].shape:
                out[i][name] = [x.v if isinstance(x, I.Int) else np.float32(x) for x in f.a]
            else:
                out[i][name] = f.v if isinstance(f, I.Int) else np.float32(f)
    return out


def aac_fused(name, seed, n_packets, nch, tree):
    from symphonia_amd import AAC_JS_DTYPE, AAC_TNS_DTYPE
    packets = [p for p, _ in A.stream(seed, n_packets, nch)]
    h, _ = A.hip_decoder(tree, nch)
    front = h.it.call("aac_front_end", h.params("CODEC_ID_AAC", 44100, nch), h.opts())
    front = h.f32_buffers(front.f["0"])
    coded, side, joint, tns, swb = [], [], [], [], {"long": np.zeros(0, np.uint16), "short": np.zeros(0, np.uint16)}
    for i, pk in enumerate(packets):
        parsed = h.it.call_method("SeamFrontEnd", "parse", front, h.packet(pk, i * 1024)).f["0"]
        fused = parsed.f["fused"].f["0"]
        coded.append(np.array([np.float32(x) for x in parsed.f["coeffs"].a], np.float32).reshape(nch, 1024))
        side.append(np.array([x.v for x in parsed.f["side"].a], np.uint8))
        for left, d in (I.deref(e) for e in fused.f["joint"].a):
            joint.append((i, I.deref(left).v, records([d], AAC_JS_DTYPE)[0]))
        t = records(fused.f["tns"].a, AAC_TNS_DTYPE)
        tns.append(np.stack([np.full(len(t), i, np.uint32), t["frame"]], axis=1) if len(t) else np.zeros((0, 2), np.uint32))
        tns.append(t)
        for key in ("long", "short"):
            tab = np.array([x.v for x in fused.f["swb_" + key].a], np.uint16)
            if tab.size:
                swb[key] = tab
    where = np.concatenate(tns[0::2])
    filters = np.concatenate(tns[1::2])
    ref = np.load(OUT / (name + ".npz"))
    np.savez_compressed(OUT / (name + "_fused.npz"), coded=np.stack(coded), side=np.stack(side), pcm=ref["pcm"],
                        joint_where=np.array([(i, left) for i, left, _ in joint], np.int32).reshape(-1, 2),
                        joint_desc=np.array([d for _, _, d in joint], AAC_JS_DTYPE).view(np.uint8).reshape(-1, 644),
                        tns_where=where.astype(np.int32), tns=filters.view(np.uint8).reshape(-1, 92), swb_long=swb["long"], swb_short=swb["short"])
    print(name + "_fused", "packets", len(packets), "jointly coded pairs", len(joint), "tns filters", len(filters))


def mp3_fused(name, seed, n, mode, mpeg1, sr_code, tree):
    from symphonia_amd.backend import MP3_REQUANT_DTYPE, MP3_STEREO_DTYPE
    s, packets = M.stream(seed, n, mode, mpeg1, sr_code)
    per_frame = 1152 if mpeg1 else 576
    h = M.shim(tree)
    front = h.it.call("mpa_front_end", h.params("CODEC_ID_MP3", s.rate, s.nch), h.opts()).f["0"]
    quant, rq, st, side = [], [], [], []
    for i, (pk, _) in enumerate(packets):
        parsed = h.it.call_method("SeamFrontEnd", "parse", front, h.packet(pk, i * per_frame)).f["0"]
        fused = parsed.f["fused"].f["0"]
        ngr = parsed.f["n_granules"].v
        quant.append(np.array([x.v for x in fused.f["quant"].a], np.int16).reshape(ngr, s.nch, 576))
        rq.append(records(fused.f["rq"].a, MP3_REQUANT_DTYPE).reshape(ngr, s.nch))
        st.append(records(fused.f["st"].a, MP3_STEREO_DTYPE).reshape(ngr))
        side.append(np.array([[q.f["block_type"].v, q.f["is_mixed"].v, q.f["rzero"].v] for q in parsed.f["side"].a], np.int32).reshape(ngr, s.nch, 3))
    ref = np.load(OUT / (name + ".npz"))
    np.savez_compressed(OUT / (name + "_fused.npz"), quant=np.stack(quant), rq=np.stack(rq).view(np.uint8).reshape(len(packets), -1, s.nch, 52),
                        st=np.stack(st).view(np.uint8).reshape(len(packets), -1, 48), side=np.stack(side), pcm=ref["pcm"],
                        sample_rate_idx=ref["sample_rate_idx"])
    print(name + "_fused", "packets", len(packets), "joint-stereo granules", int((np.stack(st)["flags"] & 3 != 0).sum()))


def vorbis_fused(name, args, tree):
    from symphonia_amd import VORBIS_FLOOR1_DTYPE
    s, packets = V.stream(*args)
    h, _ = V.hip_decoder(tree, s, max_batch=1)
    front = h.it.call("vorbis_front_end", h.params("CODEC_ID_VORBIS", 44100, s.nch, extra=s.extra_data()), h.opts()).f["0"]
    floors = records(h.it.call_method("SeamFrontEnd", "floors", front).a, VORBIS_FLOOR1_DTYPE)
    flags, residue, floor, posts, coupling, first = [], [], [], [], [], [0]
    for pk, _ in packets:
        parsed = h.it.call_method("SeamFrontEnd", "parse", front, h.packet(pk, 0)).f["0"]
        fused = parsed.f["fused"].f["0"]
        flags.append(int(bool(parsed.f["long_block"])))
        residue.append(np.array([np.float32(x) for x in parsed.f["spectra"].a], np.float32).reshape(s.nch, -1))
        floor.append(np.array([x.v for x in fused.f["floor"].a], np.uint8))
        posts.append(np.array([x.v for x in fused.f["posts"].a], np.uint32).reshape(s.nch, 65))
        coupling.extend(x.v for x in fused.f["coupling"].a)
        first.append(len(coupling) // 2)
    ref = np.load(OUT / (name + ".npz"))
    np.savez_compressed(OUT / (name + "_fused.npz"), long_block=np.array(flags, np.uint8), residue=np.concatenate(residue, axis=1),
                        floor=np.stack(floor), posts=np.stack(posts), coupling=np.array(coupling, np.uint8).reshape(-1, 2),
                        coupling_first=np.array(first, np.uint32), floors=floors.view(np.uint8).reshape(-1, 264), pcm=ref["pcm"],
                        frames=ref["frames"], block_exps=ref["block_exps"])
    print(name + "_fused", "packets", len(packets), "coupling steps", len(coupling) // 2, "unused floors", int((np.stack(floor) == 255).sum()))


def flac(name, seed, n_frames, nch, bps, blocksize):
    frames, _ = flac_writer.random_stream(seed, n_frames, nch, bps, blocksize)
    tree = patched_tree(("symphonia-bundle-flac",)) / "symphonia-bundle-flac" / "src"
    ref = Harness(None, reference=True, flac_tree=REF / "symphonia-bundle-flac" / "src")
    ref_dec = F.cpu_decoder(ref, nch, bps, blocksize)
    h, _ = F.hip_decoder(tree, nch, bps, blocksize, max_batch=1)
    front = h.it.call("flac_front_end", h.params("CODEC_ID_FLAC", extra=F.streaminfo(blocksize, 44100, nch, bps)), h.opts())
    assert front.variant == "Ok", front
    front = front.f["0"]
    out = {"pcm": [], "words": [], "desc": [], "coeffs": [], "pair_mode": [], "out_shift": []}
    for i, fr in enumerate(frames):
        st, planes = ref.decode("FlacDecoder", ref_dec, ref.packet(fr, i * blocksize))
        assert st == "ok"
        out["pcm"].append(planes)
        r = h.it.call_method("SeamFrontEnd", "parse", front, h.packet(fr, i * blocksize))
        assert r.variant == "Ok", r
        q = r.f["0"]
        assert q.f["blocksize"].v == blocksize
        out["words"].append(ints(q.f["words"]).reshape(nch, blocksize))
        out["desc"].append(np.array([[d.f[k].v for k in ("kind", "order", "shift", "wasted_bits")] for d in q.f["desc"].a], np.uint8))
        out["coeffs"].append(ints(q.f["coeffs"]).reshape(nch, 32))
        out["pair_mode"].append(q.f["pair_mode"].v)
        out["out_shift"].append(q.f["out_shift"].v)
    np.savez_compressed(OUT / (name + ".npz"), pcm=np.stack(out["pcm"]).astype(np.int32), words=np.stack(out["words"]).astype(np.int32),
                        desc=np.stack(out["desc"]), coeffs=np.stack(out["coeffs"]).astype(np.int32), pair_mode=np.array(out["pair_mode"], np.uint8),
                        out_shift=np.array(out["out_shift"], np.uint32), packet_bytes=np.frombuffer(b"".join(frames), np.uint8),
                        packet_lens=np.array([len(f) for f in frames], np.int32), bps=np.array([bps], np.int32))
    print(name, "frames", len(frames), "bytes", sum(len(f) for f in frames))


def alac(name, args):
    seed, n_packets, nch, depth, frame_length = args
    packets, _ = L.stream(seed, n_packets, nch, depth, frame_length)
    packets, n_frames = packets[:-1], frame_length   # (without the partial last packet: one shape for every array)
    tree = patched_tree((L.CRATE,)) / L.CRATE / "src"
    ref = Harness(None, reference=True, alac_tree=REF / L.CRATE / "src")
    ref_dec = L.cpu_decoder(ref, nch, depth, frame_length)
    h, _ = L.hip_decoder(tree, nch, depth, frame_length, max_batch=1)
    front = h.it.call("alac_front_end", h.params("CODEC_ID_ALAC", extra=L.W.cookie(frame_length, depth, nch)), h.opts())
    assert front.variant == "Ok", front
    front = front.f["0"]
    out = {"pcm": [], "words": [], "desc": [], "coeffs": [], "pairs": [], "tails": [], "tail_bits": [], "out_shift": []}
    for i, pk in enumerate(packets):
        st, planes = ref.decode("AlacDecoder", ref_dec, ref.packet(pk, i * frame_length))
        assert st == "ok" and planes.shape[1] == n_frames
        out["pcm"].append(planes)
        r = h.it.call_method("SeamFrontEnd", "parse", front, h.packet(pk, i * frame_length))
        assert r.variant == "Ok", r
        q = r.f["0"]
        out["words"].append(ints(q.f["words"]).reshape(nch, n_frames))
        out["desc"].append(np.array([[d.f[k].v for k in ("mode", "lpc_order", "shift", "bps")] for d in q.f["desc"].a], np.uint8))
        out["coeffs"].append(ints(q.f["coeffs"]).reshape(nch, 32))
        out["out_shift"].append(q.f["out_shift"].v)
        for pr in q.f["pairs"].a:
            out["pairs"].append([i, pr.f["plane0"].v, pr.f["plane1"].v, pr.f["weight"].v, pr.f["shift"].v])
        for tl in q.f["tails"].a:
            p1 = tl.f["plane1"]
            out["tails"].append([i, tl.f["plane0"].v, p1.f["0"].v if p1.variant == "Some" else -1, tl.f["shift"].v, len(out["tail_bits"]), len(tl.f["bits"].a)])
            out["tail_bits"].extend(x.v for x in tl.f["bits"].a)
    np.savez_compressed(OUT / (name + ".npz"), pcm=np.stack(out["pcm"]).astype(np.int32), words=np.stack(out["words"]).astype(np.int32),
                        desc=np.stack(out["desc"]), coeffs=np.stack(out["coeffs"]).astype(np.int32),
                        pairs=np.array(out["pairs"], np.int32).reshape(-1, 5), tails=np.array(out["tails"], np.int32).reshape(-1, 6),
                        tail_bits=np.array(out["tail_bits"], np.uint16), out_shift=np.array(out["out_shift"], np.uint32),
                        packet_bytes=np.frombuffer(b"".join(packets), np.uint8), packet_lens=np.array([len(p) for p in packets], np.int32),
                        depth=np.array([depth], np.int32))
    print(name, "packets", len(packets), "bytes", sum(len(p) for p in packets), "pairs", len(out["pairs"]), "tails", len(out["tails"]))


def main():
    OUT.mkdir(parents=True, exist_ok=True)
    which = set(sys.argv[1:]) or {"aac", "mp3", "vorbis", "flac", "alac", "aac_fused", "mp3_fused", "vorbis_fused"}
    if "aac" in which:
        tree = patched_tree((A.CRATE,)) / A.CRATE / "src"
        for name, (seed, n, nch) in AAC_STREAMS.items():
            aac(name, seed, n, nch, tree)
    if "aac_fused" in which:
        tree = patched_tree((A.CRATE,)) / A.CRATE / "src"
        for name, (seed, n, nch) in AAC_STREAMS.items():
            aac_fused(name, seed, n, nch, tree)
    if "mp3_fused" in which:
        tree = patched_tree((M.CRATE,)) / M.CRATE / "src"
        for name, args in MP3_STREAMS.items():
            mp3_fused(name, *args, tree)
    if "vorbis_fused" in which:
        tree = patched_tree((V.CRATE,)) / V.CRATE / "src"
        for name, args in VORBIS_STREAMS.items():
            vorbis_fused(name, args, tree)
    if "mp3" in which:
        tree = patched_tree((M.CRATE,)) / M.CRATE / "src"
        for name, args in MP3_STREAMS.items():
            mp3(name, *args, tree)
    if "flac" in which:
        flac("flac_24bit", 3, 5, 2, 24, 100)
        flac("flac_16bit", 1, 6, 2, 16, 192)
    if "alac" in which:
        alac("alac_24bit_stereo", (5, 7, 2, 24, 160))
        alac("alac_20bit_6ch", (4, 6, 6, 20, 64))
    if "vorbis" in which:
        tree = patched_tree((V.CRATE,)) / V.CRATE / "src"
        for name, args in VORBIS_STREAMS.items():
            vorbis(name, args, tree)


if __name__ == "__main__":
    main()
