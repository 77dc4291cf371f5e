"""symaccel_vorbis_decode: the Vorbis tail host to host from what the packet decoder produces -- residue vectors, floor-1 posts,
the coupling steps of every block (lib.rs:250-331).  Expectation = the oracle chain in the reference's order: inverse coupling
(lib.rs:252-278, steps in order), floor curve (floor.rs:568-653), dot product (lib.rs:282-292; an unused floor is all zeros), then
DspChannel::synth (dsp.rs:68-126).  Several streams per batch, several floor configurations, chained coupling steps, channels whose
floor is unused with and without a coupled partner.  CPU emulation here, the MI355X under `-m gpu`."""
import numpy as np
import pytest

import oracle
from emu_lib import emu_ctx  # noqa: F401
from helpers import bit_equal
from symphonia_amd import VORBIS_FLOOR1_DTYPE, SymaccelError, VorbisDsp


def make_floors(rng, bs0e, bs1e, n_floors):
    """floor-1 configurations: the x list must fit the SHORT block too (floor1_X values are below the block's half size in a real
    setup; a post at or beyond n/2 just ends the rendering early, floor.rs:644-652)"""
    floors = np.zeros(n_floors, VORBIS_FLOOR1_DTYPE)
    lists = []
    for f in range(n_floors):
        n_posts, mult = int(rng.integers(2, 30)), int(rng.integers(1, 5))
        top = 1 << (bs1e - 1)
        xs = [0, top] + rng.permutation(np.arange(1, top))[:n_posts - 2].tolist()
        floors[f]["multiplier"], floors[f]["n_posts"] = mult, n_posts
        floors[f]["x_list"][:n_posts] = xs
        lists.append((xs, mult))
    return floors, lists


def case(seed, bs0e, bs1e, n_streams, cps, nb, n_floors=2):
    rng = np.random.default_rng(seed)
    nch = n_streams * cps
    flags = np.repeat(rng.integers(0, 2, (n_streams, nb)).astype(np.uint8), cps, axis=0)
    prev = np.repeat(rng.integers(-1, 2, n_streams).astype(np.int32), cps)
    so, po = oracle.vorbis_layout(bs0e, bs1e, flags, prev)
    spec_stride = int(so[:, -1].max() + 3) & ~3
    pcm_stride = int(po[:, -1].max() + 3) & ~3
    residue = (rng.standard_normal((nch, spec_stride)) * np.exp2(rng.integers(-3, 6, (nch, 1)))).astype(np.float32)
    residue[rng.random(residue.shape) < 0.2] = 0.0
    residue[rng.random(residue.shape) < 0.02] *= -0.0
    overlap = rng.standard_normal((nch, 1 << (bs1e - 1))).astype(np.float32)
    floors, lists = make_floors(rng, bs0e, bs1e, n_floors)
    floor = rng.integers(0, n_floors, (nch, nb)).astype(np.uint8)
    floor[rng.random((nch, nb)) < 0.25] = 255
    posts = np.zeros((nch, nb, 65), np.uint32)
    for c in range(nch):
        for b in range(nb):
            if floor[c, b] != 255:
                mult = lists[floor[c, b]][1]
                y = rng.integers(0, [256, 128, 86, 64][mult - 1], 65)
                y[rng.random(65) < 0.2] = 0
                posts[c, b] = y
    steps, first = [], [0]
    for s in range(n_streams):
        for b in range(nb):
            for _ in range(int(rng.integers(0, 4)) if cps > 1 else 0):
                m, a = rng.choice(cps, 2, replace=False)
                steps.append((int(m), int(a)))
            first.append(len(steps))
    coupling = np.array(steps, np.uint8).reshape(-1, 2)
    first = np.array(first, np.uint32)
    # a do-not-decode channel (unused floor and, after the propagation of lib.rs:215-228, no coupled partner in use): zero residue
    for s in range(n_streams):
        for b in range(nb):
            lo, hi = first[s * nb + b], first[s * nb + b + 1]
            used = np.array([floor[s * cps + c, b] != 255 for c in range(cps)])
            dnd = ~used
            for m, a in coupling[lo:hi]:
                if dnd[m] != dnd[a]:
                    dnd[m] = dnd[a] = False
            for c in range(cps):
                if dnd[c]:
                    residue[s * cps + c, so[s * cps, b]:so[s * cps, b + 1]] = 0.0
    return flags, prev, residue, overlap, pcm_stride, floors, lists, floor, posts, coupling, first, so


def expectation(bs0e, bs1e, cps, flags, prev, residue, overlap, pcm_stride, lists, floor, posts, coupling, first, so):
    nch, nb = flags.shape
    res = residue.copy()
    spectrum = np.zeros_like(res)
    for s in range(nch // cps):
        for b in range(nb):
            lo, hi = so[s * cps, b], so[s * cps, b + 1]
            for m, a in coupling[first[s * nb + b]:first[s * nb + b + 1]]:
                res[s * cps + m, lo:hi], res[s * cps + a, lo:hi] = oracle.vorbis_inverse_coupling(res[s * cps + m, lo:hi], res[s * cps + a, lo:hi])
            for c in range(s * cps, (s + 1) * cps):
                n2 = hi - lo
                if floor[c, b] == 255:
                    curve = np.zeros(n2, np.float32)
                else:
                    xs, mult = lists[floor[c, b]]
                    curve = oracle.vorbis_floor1(xs, posts[c, b, :len(xs)], mult, n2)
                spectrum[c, lo:hi] = oracle.vorbis_dot_product(curve, res[c, lo:hi])
    return oracle.vorbis_synth(bs0e, bs1e, spectrum, flags, prev, overlap, pcm_stride)


def run(ctx, seed, bs0e, bs1e, n_streams, cps, nb):
    flags, prev, residue, overlap, pcm_stride, floors, lists, floor, posts, coupling, first, so = case(seed, bs0e, bs1e, n_streams, cps, nb)
    want = expectation(bs0e, bs1e, cps, flags, prev, residue, overlap, pcm_stride, lists, floor, posts, coupling, first, so)
    v = VorbisDsp(ctx, bs0e, bs1e)
    pf, ov = prev.copy(), overlap.copy()
    pcm = np.zeros((flags.shape[0], pcm_stride), np.float32)
    before = residue.copy()
    v.decode(residue, flags, floor, posts, floors, cps, coupling, first, pf, ov, pcm_stride, pcm)
    assert bit_equal(residue, before), "the caller's residue must not be touched"
    assert bit_equal(pcm, want[0]), (bs0e, bs1e, n_streams, cps, nb)
    assert bit_equal(ov, want[1]) and np.array_equal(pf, want[2])


CASES = [(8, 11, 2, 2, 7), (6, 9, 1, 3, 9), (7, 10, 3, 1, 5), (8, 8, 1, 2, 4), (9, 12, 1, 4, 5)]


@pytest.mark.parametrize("bs0e,bs1e,n_streams,cps,nb", CASES)
def test_emu_vorbis_decode(emu_ctx, bs0e, bs1e, n_streams, cps, nb):
    run(emu_ctx, 40 + bs0e + 16 * bs1e + nb, bs0e, bs1e, n_streams, cps, nb)


def test_emu_vorbis_decode_argument_checks(emu_ctx):
    flags, prev, residue, overlap, pcm_stride, floors, lists, floor, posts, coupling, first, so = case(3, 8, 11, 1, 2, 4)
    v = VorbisDsp(emu_ctx, 8, 11)
    pcm = np.zeros((2, pcm_stride), np.float32)
    args = lambda **kw: dict(dict(residue=residue, block_flag=flags, floor=floor, posts=posts, floors=floors, channels_per_stream=2,  # noqa: E731
                                  coupling=coupling, coupling_first=first, prev_flag=prev.copy(), overlap=overlap.copy(), pcm_stride=pcm_stride,
                                  pcm=pcm), **kw)
    bad_flags = flags.copy()
    bad_flags[1, 0] ^= 1  # the channels of a stream must share their block flags
    with pytest.raises(SymaccelError):
        v.decode(**args(block_flag=bad_flags))
    bad_floor = floor.copy()
    bad_floor[0, 0] = 7  # names no configuration
    with pytest.raises(SymaccelError):
        v.decode(**args(floor=bad_floor))
    bad_steps = np.array([[0, 0]], np.uint8)  # a channel coupled with itself (lib.rs:253)
    with pytest.raises(SymaccelError):
        v.decode(**args(coupling=bad_steps, coupling_first=np.array([0, 1, 1, 1, 1], np.uint32)))
    big = posts.copy()
    big[floor != 255] = 600  # outside the closed form's range: the caller renders such a block itself
    with pytest.raises(SymaccelError) as e:
        v.decode(**args(posts=big))
    assert e.value.status == -2


@pytest.mark.gpu
@pytest.mark.parametrize("bs0e,bs1e,n_streams,cps,nb", CASES + [(8, 11, 8, 8, 40), (10, 13, 2, 2, 9), (12, 13, 1, 2, 6)])
def test_gpu_vorbis_decode(bs0e, bs1e, n_streams, cps, nb):
    from symphonia_amd import Context
    ctx = Context(0)
    run(ctx, 140 + bs0e + 16 * bs1e + nb, bs0e, bs1e, n_streams, cps, nb)
    ctx.close()
