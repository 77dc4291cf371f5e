"""MP3 Layer III joint stereo (layer3/stereo.rs:485-556).

The reference has no test for this stage: the oracle's restatement is "parity unpinned by the reference".  It is pinned
here by (a) the closed forms of the two ratio tables (ISO/IEC 11172-3 2.4.3.4.9.3, 13818-3 2.4.3.2), (b) an
independent numpy restatement of the band walk written from the standard's description (intensity-coded = the zero
part of channel 1 from the top, per window for short blocks; mid/side below the bound), using the band tables
recorded from the reference, and (c) invertibility of mid/side.  The kernel is compared bit for bit with the oracle:
CPU emulation here, MI355X under -m gpu."""
import json
from pathlib import Path

import numpy as np
import pytest

import oracle
from emu_lib import emu_ctx  # noqa: F401
from helpers import bit_equal

KATS = json.loads((Path(__file__).parent / "golden" / "ref_kats.json").read_text())
SHORT = 2
F = np.float32
C = F(0.70710678118654752440)


def ms(a, b):
    return ((a + b) * C).astype(F), ((a - b) * C).astype(F)


def numpy_stereo(ch0, ch1, d, sr):
    """Independent restatement: decide an action per band first, then apply."""
    ch0, ch1 = ch0.copy(), ch1.copy()
    mid_side, intensity = bool(d["flags"] & 1), bool(d["flags"] & 2)
    if not (mid_side or intensity):
        return ch0, ch1
    end = max(int(d["rzero0"]), int(d["rzero1"]))
    m1, m2 = oracle.mp3_intensity_ratios()
    table, inv = (m1, 7) if d["flags"] & 4 else (m2[1 if d["flags"] & 8 else 0], 31)
    sf = d["scalefacs1"]
    actions = {}  # (start, end) -> ("is", kl, kr) | "ms"
    bound = end

    def zero(a, b):
        return not np.any(ch1[a:b] != 0)

    def intensity_or_ms(pos):
        if pos < inv:
            return ("is", table[pos][0], table[pos][1])
        return "ms" if mid_side else None

    if intensity:
        if d["block_type"] == SHORT:
            mixed = bool(d["is_mixed"])
            edges = KATS["mp3_sfb_mixed"][sr] if mixed else KATS["mp3_sfb_short"][sr]
            sw = KATS["mp3_sfb_mixed_switch"][sr] if mixed else 0
            pos = list(sf[:36]) + list(sf[33:36])
            n_win = (len(edges) - 1 - sw) // 3 * 3  # the short part, whole bands of three windows
            alive = [True, True, True]
            found = False
            for band_top in range(sw + n_win, sw, -3):  # bands from the top, windows 2, 1, 0
                for w in (2, 1, 0):
                    k = band_top - 3 + w
                    alive[w] = alive[w] and zero(edges[k], edges[k + 1])
                    actions[(edges[k], edges[k + 1])] = intensity_or_ms(pos[k]) if alive[w] else ("ms" if mid_side else None)
                bound = edges[band_top - 3]
                if not any(alive):
                    found = True
                    break
            if not found and mixed:
                for k in range(sw - 1, -1, -1):
                    if not zero(edges[k], edges[k + 1]):
                        break
                    actions[(edges[k], edges[k + 1])] = intensity_or_ms(pos[k])
                    bound = edges[k]
        else:
            edges = KATS["mp3_sfb_long"][sr]
            pos = list(sf[:21]) + [sf[20]]
            for k in range(21, -1, -1):
                if not (edges[k] >= int(d["rzero1"]) or zero(edges[k], edges[k + 1])):
                    break
                actions[(edges[k], edges[k + 1])] = intensity_or_ms(pos[k])
                bound = edges[k]
    o0, o1 = ch0.copy(), ch1.copy()
    for (a, b), act in actions.items():
        if act == "ms":
            o0[a:b], o1[a:b] = ms(ch0[a:b], ch1[a:b])
        elif act is not None:
            o0[a:b], o1[a:b] = (F(act[1]) * ch0[a:b]).astype(F), (F(act[2]) * ch0[a:b]).astype(F)
    if mid_side and bound > 0:
        o0[:bound], o1[:bound] = ms(ch0[:bound], ch1[:bound])
    return o0, o1


def make_granule(rng, sr, kind, flags):
    d = np.zeros((), oracle.MP3_STEREO_DTYPE)
    d["flags"] = flags
    d["block_type"] = SHORT if kind in ("short", "mixed") else rng.choice([0, 1, 3])
    d["is_mixed"] = kind == "mixed"
    inv = 7 if flags & 4 else 31
    d["scalefacs1"] = rng.integers(0, inv + 1 if rng.random() < 0.7 else 8, 39)
    ch0 = rng.standard_normal(576).astype(F)
    ch1 = rng.standard_normal(576).astype(F)
    rz0 = int(rng.integers(0, 289)) * 2
    style = rng.integers(0, 4)
    if style == 0:      # channel 1 stops early: the classic intensity layout
        rz1 = int(rng.integers(0, rz0 // 2 + 1)) * 2
    elif style == 1:    # channel 1 longer than channel 0
        rz1 = int(rng.integers(rz0 // 2, 289)) * 2
    else:
        rz1 = rz0
    ch0[rz0:] = 0
    ch1[rz1:] = 0
    if style == 3 and kind != "long":  # holes in single windows of short bands
        edges = KATS["mp3_sfb_mixed" if kind == "mixed" else "mp3_sfb_short"][sr]
        for k in rng.integers(0, len(edges) - 1, 6):
            ch1[edges[k]:edges[k + 1]] = 0
    if style == 3 and kind == "long":
        edges = KATS["mp3_sfb_long"][sr]
        for k in rng.integers(10, 22, 4):
            ch1[edges[k]:edges[k + 1]] = 0
    d["rzero0"], d["rzero1"] = rz0, rz1
    return ch0, ch1, d


def cases(seed, n, sr):
    rng = np.random.default_rng(seed)
    out = []
    for i in range(n):
        kind = ("long", "short", "mixed")[i % 3]
        flags = int(rng.choice([0, 1, 2, 3, 3, 3])) | (4 if rng.random() < 0.5 else 0) | (8 if rng.random() < 0.5 else 0)
        out.append(make_granule(rng, sr, kind, flags))
    return out


# ---------------------------------------------------------------- oracle pins

def test_ratio_tables_closed_form():
    m1, m2 = oracle.mp3_intensity_ratios()
    for p in range(6):
        r = np.tan(p * np.pi / 12)
        np.testing.assert_allclose(m1[p], [r / (1 + r), 1 / (1 + r)], rtol=2e-7, atol=1e-8)
    assert m1[6, 0] == 1 and m1[6, 1] == 0 and m1[0, 0] == 0 and m1[0, 1] == 1
    for k, i0 in enumerate((2 ** -0.25, 2 ** -0.5)):
        for p in range(32):
            want = (i0 ** ((p + 1) // 2), 1.0) if p & 1 else (1.0, i0 ** (p // 2))
            np.testing.assert_allclose(m2[k, p], want, rtol=2e-7)


@pytest.mark.parametrize("sr", [0, 2, 3, 8])
def test_oracle_stereo_matches_the_band_walk(sr):
    for ch0, ch1, d in cases(sr, 90, sr):
        got = oracle.mp3_stereo(ch0, ch1, d, sr)
        want = numpy_stereo(ch0, ch1, d, sr)
        assert bit_equal(got[0], want[0]) and bit_equal(got[1], want[1]), d


def test_mid_side_round_trip():
    rng = np.random.default_rng(3)
    l, r = rng.standard_normal(576).astype(F), rng.standard_normal(576).astype(F)
    m, s = ms(l, r)  # the encoder's (l + r) / sqrt 2, (l - r) / sqrt 2
    d = np.zeros((), oracle.MP3_STEREO_DTYPE)
    d["flags"], d["rzero0"], d["rzero1"] = 1, 576, 576
    o0, o1 = oracle.mp3_stereo(m, s, d, 0)
    np.testing.assert_allclose(o0, l, atol=1e-6)
    np.testing.assert_allclose(o1, r, atol=1e-6)


# ---------------------------------------------------------------- kernel vs oracle

def batch(seed, sr, n_pairs, granules):
    cs = cases(seed, n_pairs * granules, sr)
    chains = 2 * n_pairs + 1  # one mono chain in the middle stays untouched
    xr = np.random.default_rng(seed).standard_normal((chains, granules, 576)).astype(F)
    order = np.random.default_rng(seed + 1).permutation(chains)
    pairs = np.array([[order[2 * p], order[2 * p + 1]] for p in range(n_pairs)], np.int32)
    desc = np.zeros((n_pairs, granules), oracle.MP3_STEREO_DTYPE)
    want = xr.copy()
    for p in range(n_pairs):
        for g in range(granules):
            ch0, ch1, d = cs[p * granules + g]
            xr[pairs[p, 0], g], xr[pairs[p, 1], g], desc[p, g] = ch0, ch1, d
            want[pairs[p, 0], g], want[pairs[p, 1], g] = oracle.mp3_stereo(ch0, ch1, d, sr)
    want[order[-1]] = xr[order[-1]]
    return xr, pairs, desc, want


@pytest.mark.parametrize("sr,n_pairs,granules", [(0, 3, 7), (3, 1, 1), (8, 2, 5), (1, 4, 6)])
def test_emu_mp3_stereo(emu_ctx, sr, n_pairs, granules):
    from symphonia_amd import Mp3Stereo
    xr, pairs, desc, want = batch(sr + n_pairs, sr, n_pairs, granules)
    got = xr.copy()
    Mp3Stereo(emu_ctx, sr).stereo(got, pairs, desc)
    assert bit_equal(got, want)


def test_emu_requantize_stereo_synth_chain(emu_ctx):
    """The tail of Layer3::decode for a stereo stream (layer3/mod.rs:393-476): requantize both channels, stereo, then the
    synthesis tail with both channels' rzero = max(rzero0, rzero1) (stereo.rs:549-553)."""
    from symphonia_amd import Mp3Requantize, Mp3Stereo, Mp3Synthesis
    import test_mp3_requantize as rq
    granules, sr = 5, 0
    q, d = rq.make_case(42, 2 * granules, kinds=("long",))
    d["block_type"][:] = np.tile([0, 2, 2, 1, 3], 2)         # same block type in both channels of a granule
    d["is_mixed"][:] = np.tile([0, 0, 1, 0, 0], 2)
    q = q.reshape(2, granules, 576)
    q[1, :, 300:] = 0                                         # channel 1 ends early: intensity-coded top
    d = d.reshape(2, granules)
    d["rzero"][1] = 300
    sd = np.zeros((1, granules), oracle.MP3_STEREO_DTYPE)
    sd["flags"], sd["block_type"], sd["is_mixed"] = 3 | 4, d["block_type"][1], d["is_mixed"][1]
    sd["rzero0"], sd["rzero1"], sd["scalefacs1"] = d["rzero"][0], d["rzero"][1], d["scalefacs"][1] % 8
    pairs = np.array([[0, 1]], np.int32)
    # oracle chain
    xr_ref = oracle.mp3_requantize(q, d, sr).reshape(2, granules, 576)
    for g in range(granules):
        xr_ref[0, g], xr_ref[1, g] = oracle.mp3_stereo(xr_ref[0, g], xr_ref[1, g], sd[0, g], sr)
    end = np.maximum(d["rzero"][0], d["rzero"][1])
    side = oracle.mp3_side(d["block_type"], d["is_mixed"], np.stack([end, end]))
    ov, vv, vf = np.zeros((2, 576), F), np.zeros((2, 1024), F), np.zeros(2, np.int32)
    want = oracle.mp3_synth(xr_ref, side, sr, ov, vv, vf)[0]
    # product chain
    xr = Mp3Requantize(emu_ctx, sr).requantize(q, d).reshape(2, granules, 576)
    Mp3Stereo(emu_ctx, sr).stereo(xr, pairs, sd)
    assert bit_equal(xr, xr_ref)
    got = Mp3Synthesis(emu_ctx, sr).synth(xr, side, ov, vv, vf)[0]
    assert bit_equal(got, want)


def fused_case(seed, sr, n_pairs, granules):
    """Quantised samples + side info for 2 * n_pairs + 1 chains; the oracle's requantize -> stereo as the expectation."""
    import test_mp3_requantize as rq
    rng = np.random.default_rng(seed)
    chains = 2 * n_pairs + 1
    q, rd = rq.make_case(seed, chains * granules, big=True)
    q, rd = q.reshape(chains, granules, 576), rd.reshape(chains, granules)
    order = rng.permutation(chains)
    pairs = np.array([[order[2 * p], order[2 * p + 1]] for p in range(n_pairs)], np.int32)
    sd = np.zeros((n_pairs, granules), oracle.MP3_STEREO_DTYPE)
    for p, (c0, c1) in enumerate(pairs):
        rd["block_type"][c1], rd["is_mixed"][c1] = rd["block_type"][c0], rd["is_mixed"][c0]  # stereo.rs:502-504
        cut = rng.integers(0, 577, granules)
        for g in range(granules):
            if rng.random() < 0.6:
                q[c1, g, cut[g]:] = 0  # channel 1 ends early: intensity-coded top
        sd["flags"][p] = rng.choice([0, 1, 2, 3, 3], granules) | rng.choice([0, 4], granules) | rng.choice([0, 8], granules)
        sd["block_type"][p], sd["is_mixed"][p] = rd["block_type"][c1], rd["is_mixed"][c1]
        sd["rzero0"][p], sd["rzero1"][p] = rd["rzero"][c0], rd["rzero"][c1]
        sd["scalefacs1"][p] = rng.integers(0, 9, (granules, 39))
    want = oracle.mp3_requantize(q, rd, sr).reshape(chains, granules, 576)
    for p, (c0, c1) in enumerate(pairs):
        for g in range(granules):
            want[c0, g], want[c1, g] = oracle.mp3_stereo(want[c0, g], want[c1, g], sd[p, g], sr)
    return q, rd, pairs, sd, want, int(order[-1])


@pytest.mark.parametrize("sr,n_pairs,granules", [(0, 2, 6), (3, 1, 3), (8, 3, 4)])
def test_emu_requantize_stereo_fused(emu_ctx, sr, n_pairs, granules):
    from symphonia_amd import Mp3Stereo
    q, rd, pairs, sd, want, mono = fused_case(50 + sr, sr, n_pairs, granules)
    xr = np.full(want.shape, np.float32(123.0))
    Mp3Stereo(emu_ctx, sr).requantize_stereo(q, rd, pairs, sd, xr)
    paired = sorted(pairs.reshape(-1))
    assert bit_equal(xr[paired], want[paired])
    assert (xr[mono] == 123.0).all()  # chains outside every pair are not this call's business


def side_of(rd, pairs, sd):
    """The synthesis side records after stereo: both channels of a joint-stereo granule get rzero = max (stereo.rs:549-553)."""
    rz = rd["rzero"].astype(np.int64).copy()
    for p, (c0, c1) in enumerate(pairs):
        joint = (sd["flags"][p] & 3) != 0
        end = np.maximum(rd["rzero"][c0], rd["rzero"][c1])
        rz[c0] = np.where(joint, end, rz[c0])
        rz[c1] = np.where(joint, end, rz[c1])
    return oracle.mp3_side(rd["block_type"], rd["is_mixed"], rz)


def tail_case(seed, sr, n_pairs, granules):
    """fused_case + incoming synthesis state; expectation = oracle requantize -> stereo -> synth (rzero = end when joint)."""
    q, rd, pairs, sd, xr_want, mono = fused_case(seed, sr, n_pairs, granules)
    rng = np.random.default_rng(seed + 7)
    chains = q.shape[0]
    ov = rng.standard_normal((chains, 576)).astype(F)
    vv = rng.standard_normal((chains, 1024)).astype(F)
    vf = rng.integers(0, 16, chains).astype(np.int32)
    rz = rd["rzero"].astype(np.int64).copy()
    for p, (c0, c1) in enumerate(pairs):
        joint = (sd["flags"][p] & 3) != 0
        end = np.maximum(rd["rzero"][c0], rd["rzero"][c1])
        rz[c0] = np.where(joint, end, rz[c0])
        rz[c1] = np.where(joint, end, rz[c1])
    side = oracle.mp3_side(rd["block_type"], rd["is_mixed"], rz)
    paired = sorted(pairs.reshape(-1))
    want = oracle.mp3_synth(xr_want[paired], side[paired], sr, ov[paired], vv[paired], vf[paired])
    return q, rd, pairs, sd, ov, vv, vf, paired, mono, want


@pytest.mark.parametrize("sr,n_pairs,granules,seg", [(0, 2, 7, 0), (3, 1, 1, 0), (8, 2, 9, 3), (1, 3, 6, 2)])
def test_emu_requantize_stereo_then_synth(emu_ctx, sr, n_pairs, granules, seg):
    from symphonia_amd import Mp3Stereo
    q, rd, pairs, sd, ov, vv, vf, paired, mono, want = tail_case(70 + sr, sr, n_pairs, granules)
    from symphonia_amd import Mp3Synthesis
    xr = np.zeros(q.shape, np.float32)
    emu_ctx.set_segment(seg)
    Mp3Stereo(emu_ctx, sr).requantize_stereo(q, rd, pairs, sd, xr)
    got = Mp3Synthesis(emu_ctx, sr).synth(xr[paired], side_of(rd, pairs, sd)[paired], ov[paired], vv[paired], vf[paired])
    emu_ctx.set_segment(0)
    assert bit_equal(got[0], want[0]), "pcm"
    assert bit_equal(got[1], want[1]) and bit_equal(got[2], want[2]) and np.array_equal(got[3], want[3]), "state"


@pytest.mark.gpu
@pytest.mark.parametrize("sr,n_pairs,granules", [(0, 9, 40), (8, 2, 17)])
def test_gpu_requantize_stereo_fused(sr, n_pairs, granules):
    import torch
    from symphonia_amd import Context, Mp3Stereo
    if not torch.cuda.is_available():
        pytest.fail("no GPU visible")
    q, rd, pairs, sd, want, mono = fused_case(150 + sr, sr, n_pairs, granules)
    with Context(0) as ctx:
        xr = torch.full(want.shape, 123.0, device="cuda")
        Mp3Stereo(ctx, sr).requantize_stereo(torch.from_numpy(q).cuda(), torch.from_numpy(rd.view(np.uint8).reshape(rd.shape + (52,))).cuda(),
                                             torch.from_numpy(pairs).cuda(), torch.from_numpy(sd.view(np.uint8).reshape(sd.shape + (48,))).cuda(), xr)
        ctx.sync()
        got = xr.cpu().numpy()
    paired = sorted(pairs.reshape(-1))
    assert bit_equal(got[paired], want[paired])
    assert (got[mono] == 123.0).all()


@pytest.mark.gpu
@pytest.mark.parametrize("sr,n_pairs,granules", [(0, 16, 64), (4, 5, 33), (8, 3, 20)])
def test_gpu_mp3_stereo(sr, n_pairs, granules):
    import torch
    from symphonia_amd import Context, Mp3Stereo
    if not torch.cuda.is_available():
        pytest.fail("no GPU visible")
    xr, pairs, desc, want = batch(100 + sr, sr, n_pairs, granules)
    with Context(0) as ctx:
        d = torch.from_numpy(xr).cuda()
        Mp3Stereo(ctx, sr).stereo(d, torch.from_numpy(pairs).cuda(),
                                  torch.from_numpy(desc.view(np.uint8).reshape(n_pairs, granules, 48)).cuda())
        ctx.sync()
        assert bit_equal(d.cpu().numpy(), want)


def decode_pipelined_case(seed, sr, n_pairs, granules, with_mono):
    """fused_case (2 n_pairs + 1 chains: one mono) or its paired chains alone; expectation = oracle requantize -> stereo -> synth."""
    q, rd, pairs, sd, xr_want, mono = fused_case(seed, sr, n_pairs, granules)
    side = side_of(rd, pairs, sd)
    if not with_mono:  # drop the mono chain and renumber
        keep = [c for c in range(q.shape[0]) if c != mono]
        remap = {c: i for i, c in enumerate(keep)}
        q, rd, xr_want, side = q[keep], rd[keep], xr_want[keep], side[keep]
        pairs = np.array([[remap[int(a)], remap[int(b)]] for a, b in pairs], np.int32)
    rng = np.random.default_rng(seed + 11)
    chains = q.shape[0]
    ov = rng.standard_normal((chains, 576)).astype(F)
    vv = rng.standard_normal((chains, 1024)).astype(F)
    vf = rng.integers(0, 16, chains).astype(np.int32)
    want = oracle.mp3_synth(xr_want, side, sr, ov, vv, vf)
    return q, rd, pairs, sd, side, ov, vv, vf, want


def run_decode_pipelined(ctx, sr, n_pairs, granules, with_mono, chunks):
    q, rd, pairs, sd, side, ov, vv, vf, want = decode_pipelined_case(90 + sr + n_pairs, sr, n_pairs, granules, with_mono)
    d = ctx.lib.dll
    chains = q.shape[0]
    q, rd, sd, side = (np.ascontiguousarray(a) for a in (q, rd, sd, side))
    for ch in chunks:
        o, v, f = ov.copy(), vv.copy(), vf.copy()
        pcm = np.zeros((chains, granules, 576), F)
        ctx._call(d.symaccel_mp3_decode_pipelined, q.ctypes.data, rd.ctypes.data, pairs.ctypes.data if len(pairs) else None,
                  sd.ctypes.data if len(pairs) else None, len(pairs), side.ctypes.data, sr, o.ctypes.data, v.ctypes.data, f.ctypes.data,
                  pcm.ctypes.data, chains, granules, ch)
        assert bit_equal(pcm, want[0]), ("pcm", ch)
        assert bit_equal(o, want[1]) and bit_equal(v, want[2]) and np.array_equal(f, want[3]), ("state", ch)


@pytest.mark.parametrize("sr,n_pairs,granules,with_mono", [(0, 2, 9, True), (3, 1, 5, False), (8, 3, 7, False), (1, 0, 4, True)])
def test_emu_mp3_decode_pipelined(emu_ctx, sr, n_pairs, granules, with_mono):
    """symaccel_mp3_decode_pipelined: int16 Huffman samples + records in host memory -> PCM in host memory (requantize, joint
    stereo, synthesis: layer3/mod.rs:421-477), in chunks of every size, with and without chains outside the pairs."""
    run_decode_pipelined(emu_ctx, sr, n_pairs, granules, with_mono, [2, 3, granules, 0])
    pairs_bad = np.array([[0, 0]], np.int32)
    z = np.zeros(8, np.uint8)
    with pytest.raises(Exception):
        emu_ctx._call(emu_ctx.lib.dll.symaccel_mp3_decode_pipelined, z.ctypes.data, z.ctypes.data, pairs_bad.ctypes.data, z.ctypes.data, 1,
                      z.ctypes.data, 0, z.ctypes.data, z.ctypes.data, z.ctypes.data, z.ctypes.data, 2, 1, 0)


@pytest.mark.gpu
@pytest.mark.parametrize("sr,n_pairs,granules,with_mono", [(0, 4, 33, True), (8, 5, 20, False)])
def test_gpu_mp3_decode_pipelined(sr, n_pairs, granules, with_mono):
    from symphonia_amd import Context
    ctx = Context(0)
    run_decode_pipelined(ctx, sr, n_pairs, granules, with_mono, [4, 0])
    ctx.close()


# ---------------------------------------------------------------- the fused kernel on device buffers (int16 -> PCM, one launch)

def units_of(pairs, chains):
    """unit_chains of symaccel_mp3_decode_*_device: the pairs, then {chain, -1} for every chain no pair names"""
    named = set(int(c) for c in np.asarray(pairs).ravel())
    rows = [[int(a), int(b)] for a, b in pairs] + [[c, -1] for c in range(chains) if c not in named]
    return np.array(rows, np.int32)


def run_decode_device(ctx, dev, host, sr, n_pairs, granules, with_mono, seg, seed=0):
    from symphonia_amd import Mp3Synthesis
    q, rd, pairs, sd, side, ov, vv, vf, want = decode_pipelined_case(70 + seed + sr + n_pairs, sr, n_pairs, granules, with_mono)
    chains = q.shape[0]
    units = units_of(pairs, chains)
    sdu = np.zeros((len(units), granules), oracle.MP3_STEREO_DTYPE)
    sdu[: len(pairs)] = sd
    as_bytes = lambda a: np.ascontiguousarray(a).view(np.uint8).reshape(a.shape + (-1,))  # noqa: E731
    d_ov, d_vv, d_vf = dev(ov), dev(vv), dev(vf)
    pcm = dev(np.zeros((chains, granules, 576), F))
    ctx.set_segment(seg)
    Mp3Synthesis(ctx, sr).decode(dev(q), dev(as_bytes(rd)), dev(units), dev(as_bytes(sdu)), dev(as_bytes(np.ascontiguousarray(side))), d_ov, d_vv,
                                 d_vf, pcm)
    ctx.set_segment(0)
    assert bit_equal(host(pcm), want[0]), ("pcm", seg)
    assert bit_equal(host(d_ov), want[1]) and bit_equal(host(d_vv), want[2]) and np.array_equal(host(d_vf), want[3]), ("state", seg)
    # the ping-pong form: state in untouched, state out written
    o2, v2, f2 = dev(np.zeros_like(ov)), dev(np.zeros_like(vv)), dev(np.zeros_like(vf))
    d_ov, d_vv, d_vf = dev(ov), dev(vv), dev(vf)
    pcm2 = dev(np.zeros((chains, granules, 576), F))
    Mp3Synthesis(ctx, sr).decode(dev(q), dev(as_bytes(rd)), dev(units), dev(as_bytes(sdu)), dev(as_bytes(np.ascontiguousarray(side))), d_ov, d_vv,
                                 d_vf, pcm2, state_out=(o2, v2, f2))
    assert bit_equal(host(pcm2), want[0]) and bit_equal(host(o2), want[1]) and bit_equal(host(v2), want[2]) and np.array_equal(host(f2), want[3])
    assert bit_equal(host(d_ov), ov) and bit_equal(host(d_vv), vv)


@pytest.mark.parametrize("sr,n_pairs,granules,with_mono,seg", [(0, 2, 9, True, 0), (3, 1, 5, False, 2), (8, 3, 11, False, 3), (1, 0, 4, True, 0),
                                                                (4, 5, 1, True, 0), (2, 2, 13, True, 4)])
def test_emu_mp3_decode_device(emu_ctx, sr, n_pairs, granules, with_mono, seg):
    """symaccel_mp3_decode_device / _pp_device: requantize + joint stereo inside the synthesis kernel's load path (csrc/mp3.hip
    mp3_front) == oracle requantize -> stereo -> synthesis, bit for bit; segments with their two-granule halo included."""
    run_decode_device(emu_ctx, lambda a: np.array(a, copy=True, order="C"), lambda a: a, sr, n_pairs, granules, with_mono, seg)


def run_decode_device_loose_records(ctx, dev, host, sr, n_pairs, granules, seed):
    """The stereo record's rzero0 / rzero1 need not equal the requantize records' rzero (they are separate arguments of the C ABI): with a
    bound BELOW a channel's rzero the lines from the bound on keep their requantised values (stereo.rs:522-543 on the record's own numbers) --
    the fused front's select path; a front end that fills both from one granule never takes it."""
    from symphonia_amd import Mp3Synthesis
    q, rd, pairs, sd, side, ov, vv, vf, _ = decode_pipelined_case(300 + seed + sr, sr, n_pairs, granules, True)
    rng = np.random.default_rng(seed)
    sd = sd.copy()
    q = q.copy()
    for p, (c0, c1) in enumerate(pairs):
        for g in range(granules):
            r = rng.random()
            if r < 0.5:
                sd["rzero0"][p, g] = rng.integers(0, int(rd["rzero"][c0, g]) + 1)
                sd["rzero1"][p, g] = rng.integers(0, int(rd["rzero"][c1, g]) + 1)
            elif r < 0.7:
                sd["rzero0"][p, g], sd["rzero1"][p, g] = 576, 0
    q[:, :, 5::41] = rng.choice(np.array([-8206, 8206, 8205, -1, 0], np.int16), q[:, :, 5::41].shape)  # (the domain's edge: linbits escapes end at 8206)
    chains = q.shape[0]
    xr = oracle.mp3_requantize(q, rd, sr).reshape(chains, granules, 576)
    for p, (c0, c1) in enumerate(pairs):
        for g in range(granules):
            xr[c0, g], xr[c1, g] = oracle.mp3_stereo(xr[c0, g], xr[c1, g], sd[p, g], sr)
    side = oracle.mp3_side(rd["block_type"], rd["is_mixed"], np.full(rd.shape, 576))  # (the synthesis sees every line)
    want = oracle.mp3_synth(xr, side, sr, ov, vv, vf)
    units = units_of(pairs, chains)
    sdu = np.zeros((len(units), granules), oracle.MP3_STEREO_DTYPE)
    sdu[: len(pairs)] = sd
    as_bytes = lambda a: np.ascontiguousarray(a).view(np.uint8).reshape(a.shape + (-1,))  # noqa: E731
    d_ov, d_vv, d_vf = dev(ov), dev(vv), dev(vf)
    pcm = dev(np.zeros((chains, granules, 576), F))
    Mp3Synthesis(ctx, sr).decode(dev(q), dev(as_bytes(rd)), dev(units), dev(as_bytes(sdu)), dev(as_bytes(np.ascontiguousarray(side))), d_ov, d_vv,
                                 d_vf, pcm)
    assert bit_equal(host(pcm), want[0])
    assert bit_equal(host(d_ov), want[1]) and bit_equal(host(d_vv), want[2]) and np.array_equal(host(d_vf), want[3])


@pytest.mark.parametrize("sr,n_pairs,granules", [(0, 3, 8), (5, 2, 5)])
def test_emu_mp3_decode_device_loose_records(emu_ctx, sr, n_pairs, granules):
    run_decode_device_loose_records(emu_ctx, lambda a: np.array(a, copy=True, order="C"), lambda a: a, sr, n_pairs, granules, 1)


@pytest.mark.gpu
@pytest.mark.parametrize("sr,n_pairs,granules", [(0, 17, 30), (5, 40, 12)])
def test_gpu_mp3_decode_device_loose_records(sr, n_pairs, granules):
    import torch
    from symphonia_amd import Context
    ctx = Context(0)
    ctx.use_torch_stream()

    def host(t):
        torch.cuda.synchronize()
        return t.cpu().numpy()
    run_decode_device_loose_records(ctx, lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda(), host, sr, n_pairs, granules, 2)
    ctx.close()


@pytest.mark.gpu
@pytest.mark.parametrize("sr,n_pairs,granules,with_mono,seg", [(0, 9, 40, True, 0), (8, 5, 33, False, 7), (3, 33, 17, True, 2), (0, 1, 1, False, 0),
                                                                (5, 70, 24, True, 0)])
def test_gpu_mp3_decode_device(sr, n_pairs, granules, with_mono, seg):
    import torch
    from symphonia_amd import Context
    ctx = Context(0)
    ctx.use_torch_stream()

    def host(t):
        torch.cuda.synchronize()
        return t.cpu().numpy()
    run_decode_device(ctx, lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda(), host, sr, n_pairs, granules, with_mono, seg, seed=3)
    ctx.close()
