"""The batch kinds of ABI 7 (csrc/batcher.cpp): SYMACCEL_BATCH_VORBIS_DECODE (residue + floor-1 posts + coupling steps -> PCM:
symaccel_vorbis_decode across streams, lib.rs:250-331), SYMACCEL_BATCH_FLAC_RESTORE (decoder.rs:663-752, with and without the
fused decorrelation of :32-82, :239-242) and SYMACCEL_BATCH_ALAC_PREDICT (alac/lib.rs:165-264, with and without :664-671) -- every
stream gets back bit for bit what its own per-stream call (the oracle chain) gives, the streams share launches; the status of a
launch is kept PER TICKET (a submission whose descriptors do not add up fails alone); closing groups are dealt to lanes.
CPU: the emulation build; GPU: libsymaccel.so."""
import threading

import numpy as np
import pytest

import oracle
from emu_lib import emu_ctx  # noqa: F401
from helpers import bit_equal
from symphonia_amd import (BATCH_AAC_DECODE, BATCH_AAC_SYNTH, BATCH_ALAC_PREDICT, BATCH_FLAC_RESTORE, BATCH_MP3_DECODE, BATCH_VORBIS_DECODE, Batcher,
                           SymaccelError, alac_desc, flac_desc)
from symphonia_amd.backend import FLAC_FIXED, FLAC_LPC, FLAC_VERBATIM
from test_staging import aac_case

F = np.float32


# ------------------------------------------------------------------------------------------------ Vorbis from posts

def vorbis_stream(b, seed, bs0e, bs1e, cps, nb, n_floors=2):
    """one stream in the batcher's terms (planes at their largest, floor indices as the batcher registered them) + its expectation"""
    import test_vorbis_decode as V
    flags, prev, residue, overlap, pcm_stride, floors, lists, floor, posts, coupling, first, so = V.case(seed, bs0e, bs1e, 1, cps, nb, n_floors)
    want = V.expectation(bs0e, bs1e, cps, flags, prev, residue, overlap, pcm_stride, lists, floor, posts, coupling, first, so)
    index = [b.vorbis_floor(mult, xs) for xs, mult in lists]
    assert [b.vorbis_floor(mult, xs) for xs, mult in lists] == index  # the same configuration: the same index
    gfloor = np.full_like(floor, 255)
    for f, gi in enumerate(index):
        gfloor[floor == f] = gi
    cap = nb << (bs1e - 1)
    res = np.zeros((cps, cap), F)
    res[:, :residue.shape[1]] = residue
    lay = oracle.vorbis_layout(bs0e, bs1e, flags, prev)
    return dict(res=res, flags=np.ascontiguousarray(flags), floor=np.ascontiguousarray(gfloor), posts=np.ascontiguousarray(posts), coupling=coupling,
                first=np.ascontiguousarray(first), prev=prev.copy(), overlap=overlap.copy(), pcm=np.full((cps, cap), np.nan, F), want=want, lay=lay,
                prev0=prev.copy())


def check_vorbis_stream(s):
    for c in range(s["pcm"].shape[0]):
        # the samples the chain's blocks emit (a first block after a reset owns slots nothing writes: lib.rs:298-303)
        start = int(s["lay"][1][c, 1]) if s["prev0"][c] < 0 else 0
        end = int(s["lay"][1][c, -1])
        assert bit_equal(s["pcm"][c, start:end], np.asarray(s["want"][0])[c, start:end]), c
    assert bit_equal(s["overlap"], np.asarray(s["want"][1])) and np.array_equal(s["prev"], np.asarray(s["want"][2]))


def run_vorbis_decode(ctx, bs0e, bs1e, shapes=((2, 3), (1, 2), (3, 1)), nb=7):
    """shapes = (channels, streams of that channel count): every channel count is a group of its own, its streams share the launch"""
    b = Batcher(ctx, 0)
    subs, seed = [], 4000 + bs0e + 16 * bs1e
    for cps, n in shapes:
        for _ in range(n):
            seed += 1
            s = vorbis_stream(b, seed, bs0e, bs1e, cps, nb, n_floors=1 + seed % 3)
            s["t"] = b.submit_vorbis_decode(bs0e, bs1e, s["res"], s["flags"], s["floor"], s["posts"], s["coupling"], s["first"], s["prev"], s["overlap"],
                                            s["pcm"])
            subs.append(s)
    assert b.stats()["pending"] == len(subs)
    for s in reversed(subs):
        b.collect(s["t"])
        check_vorbis_stream(s)
    st = b.stats()
    assert st["launches"] == len(shapes) and st["failed_tickets"] == 0, st
    b.close()


@pytest.mark.parametrize("bs0e,bs1e", [(8, 11), (6, 9), (8, 8)])
def test_emu_vorbis_decode_streams_share_launches(emu_ctx, bs0e, bs1e):
    run_vorbis_decode(emu_ctx, bs0e, bs1e)


def run_vorbis_decode_bad_ticket(ctx):
    """floor1_Y values above 511 (symaccel_vorbis_decode: SYMACCEL_ERR_UNSUPPORTED), block flags that differ between the channels of
    a stream, a step that couples a channel with itself: each fails ALONE, the neighbours of the launch are bit-exact"""
    b = Batcher(ctx, 0)
    good = [vorbis_stream(b, 4100 + i, 8, 11, 2, 5) for i in range(3)]
    bad_posts = vorbis_stream(b, 4110, 8, 11, 2, 5)
    bad_posts["floor"][:] = bad_posts["floor"][bad_posts["floor"] != 255].flat[0] if (bad_posts["floor"] != 255).any() else 0
    bad_posts["posts"][:] = 600
    bad_flags = vorbis_stream(b, 4111, 8, 11, 2, 5)
    bad_flags["flags"][1, 0] ^= 1
    bad_step = vorbis_stream(b, 4112, 8, 11, 2, 5)
    bad_step["coupling"] = np.array([[1, 1]], np.uint8)
    bad_step["first"] = np.array([0, 1, 1, 1, 1, 1], np.uint32)
    order = [good[0], bad_posts, good[1], bad_flags, bad_step, good[2]]
    for s in order:
        s["t"] = b.submit_vorbis_decode(8, 11, s["res"], s["flags"], s["floor"], s["posts"], s["coupling"], s["first"], s["prev"], s["overlap"], s["pcm"])
    for s, status in ((bad_posts, -2), (bad_flags, -1), (bad_step, -1)):
        with pytest.raises(SymaccelError) as e:
            b.collect(s["t"])
        assert e.value.status == status
    for s in good:
        b.collect(s["t"])
        check_vorbis_stream(s)
    st = b.stats()
    assert st["launches"] == 1 and st["failed_tickets"] == 3 and st["pending"] == 0, st
    b.close()


def test_emu_vorbis_decode_bad_submission_fails_alone(emu_ctx):
    run_vorbis_decode_bad_ticket(emu_ctx)


# ------------------------------------------------------------------------------------------------ FLAC / ALAC

def flac_stream(rng, n_frames, nch, blocksize, valid=True):
    nb = n_frames * nch
    buf = rng.integers(-(1 << 22), 1 << 22, (nb, blocksize)).astype(np.int32)
    kind = rng.integers(0, 3, nb).astype(np.uint8)
    order = np.minimum(np.where(kind == FLAC_FIXED, rng.integers(0, 5, nb), rng.integers(1, 33, nb)), blocksize).astype(np.uint8)
    kind[(kind == FLAC_LPC) & (order == 0)] = FLAC_VERBATIM
    shift = rng.integers(0, 16, nb).astype(np.uint8)
    wasted = np.where(rng.random(nb) < 0.2, rng.integers(1, 4, nb), 0).astype(np.uint8)
    coeffs = rng.integers(-(1 << 14), 1 << 14, (nb, 32)).astype(np.int32)
    return buf, kind, order, shift, wasted, coeffs


def run_flac(ctx, blocksize=100):
    """streams of 1 / 2 / 3 channels and DIFFERENT numbers of frames per batch share ONE launch (a chain is a subframe); the fused
    stereo form (param 0x100 | shift) is a group of its own"""
    rng = np.random.default_rng(blocksize)
    b = Batcher(ctx, 0)
    subs = []
    for s in range(6):
        buf, kind, order, shift, wasted, coeffs = flac_stream(rng, 2 + s, 1 + s % 3, blocksize)
        want = oracle.flac_restore(buf, oracle.flac_desc(kind, order, shift, wasted), coeffs)
        got = buf.copy()
        subs.append((b.submit_flac_restore(got, np.ascontiguousarray(flac_desc(kind, order, shift, wasted)), coeffs), got, want))
    stereo = []
    for s in range(4):
        buf, kind, order, shift, wasted, coeffs = flac_stream(rng, 3 + s, 2, blocksize)
        mode = rng.integers(0, 4, buf.shape[0] // 2).astype(np.uint8)
        want = oracle.flac_restore(buf, oracle.flac_desc(kind, order, shift, wasted), coeffs)
        for p in range(buf.shape[0] // 2):
            x, y = oracle.flac_decorrelate(int(mode[p]), want[2 * p], want[2 * p + 1])
            want[2 * p], want[2 * p + 1] = oracle.flac_shl(x, 8), oracle.flac_shl(y, 8)
        got = buf.copy()
        stereo.append((b.submit_flac_restore(got, np.ascontiguousarray(flac_desc(kind, order, shift, wasted)), coeffs, mode, 8), got, want))
    assert b.stats()["pending"] == 10
    for t, got, want in subs + stereo:
        b.collect(t)
        assert np.array_equal(got, want)
    st = b.stats()
    assert st["launches"] == 2 and st["max_chains_per_launch"] == sum((2 + s) * (1 + s % 3) for s in range(6)) and st["failed_tickets"] == 0, st
    # a submission with a predictor order above the block size (decoder.rs:456-458) fails alone
    buf, kind, order, shift, wasted, coeffs = flac_stream(rng, 2, 2, 8)
    kind[1], order[1] = FLAC_LPC, 9
    bad = b.submit_flac_restore(buf.copy(), np.ascontiguousarray(flac_desc(kind, order, shift, wasted)), coeffs)
    buf2, kind2, order2, shift2, wasted2, coeffs2 = flac_stream(rng, 3, 1, 8)
    got2 = buf2.copy()
    ok = b.submit_flac_restore(got2, np.ascontiguousarray(flac_desc(kind2, order2, shift2, wasted2)), coeffs2)
    with pytest.raises(SymaccelError) as e:
        b.collect(bad)
    assert e.value.status == -1
    b.collect(ok)
    assert np.array_equal(got2, oracle.flac_restore(buf2, oracle.flac_desc(kind2, order2, shift2, wasted2), coeffs2))
    assert b.stats()["failed_tickets"] == 1
    with pytest.raises(Exception):
        b.reserve(BATCH_FLAC_RESTORE, 0x108, 3, 64)  # pairs: an even number of subframes
    with pytest.raises(Exception):
        b.reserve(BATCH_FLAC_RESTORE, 0, 2, 65536)   # frame.rs:58: a block size has 16 bits
    b.close()


def test_emu_flac_streams_share_a_launch(emu_ctx):
    run_flac(emu_ctx)


def test_emu_flac_streams_padded_device_rows(emu_ctx):
    """1024-sample blocks: the group's device plane has its rows at symaccel_row_stride(1024) = 1152 words (gather / scatter row by row),
    the slots stay compact; the same streams with SYMACCEL_BATCH_ROW_PAD=0 would take the compact plane -- both bit-equal to the oracle"""
    assert emu_ctx.lib.dll.symaccel_row_stride(1024) == 1152
    run_flac(emu_ctx, 1024)


def alac_stream(rng, nb, blocksize):
    from test_alac import alac_case
    buf, mode, order, shift, bps, coeffs = alac_case(int(rng.integers(1 << 30)), nb, blocksize)
    mode[mode == 7] = 15  # (an invalid mode fails the submission: checked on its own below)
    return buf, mode, order, shift, bps, coeffs


def run_alac(ctx, blocksize=100):
    rng = np.random.default_rng(3 * blocksize)
    b = Batcher(ctx, 0)
    subs = []
    for s in range(5):
        buf, mode, order, shift, bps, coeffs = alac_stream(rng, (2 + s) * (1 + s % 2), blocksize)
        want = oracle.alac_predict(buf, oracle.alac_desc(mode, order, shift, bps), coeffs)
        got = buf.copy()
        subs.append((b.submit_alac_predict(got, np.ascontiguousarray(alac_desc(mode, order, shift, bps)), coeffs), got, want))
    for s in range(3):
        nb = 2 * (3 + s)
        buf, mode, order, shift, bps, coeffs = alac_stream(rng, nb, blocksize)
        weight = rng.integers(-3, 4, nb // 2).astype(np.int32)
        msh = rng.integers(0, 32, nb // 2).astype(np.uint8)
        want = oracle.alac_predict(buf, oracle.alac_desc(mode, order, shift, bps), coeffs)
        for p in range(nb // 2):
            if weight[p]:
                want[2 * p], want[2 * p + 1] = oracle.alac_decorrelate_mid_side(want[2 * p], want[2 * p + 1], int(weight[p]), int(msh[p]))
        got = buf.copy()
        subs.append((b.submit_alac_predict(got, np.ascontiguousarray(alac_desc(mode, order, shift, bps)), coeffs, weight, msh), got, want))
    for t, got, want in subs:
        b.collect(t)
        assert np.array_equal(got, want)
    st = b.stats()
    assert st["launches"] == 2 and st["failed_tickets"] == 0, st
    # lib.rs:167-169: a mode between 1 and 14 is a decode error of THAT packet's submission
    buf, mode, order, shift, bps, coeffs = alac_stream(rng, 4, blocksize)
    mode[2] = 7
    bad = b.submit_alac_predict(buf.copy(), np.ascontiguousarray(alac_desc(mode, order, shift, bps)), coeffs)
    buf2, mode2, order2, shift2, bps2, coeffs2 = alac_stream(rng, 3, blocksize)
    got2 = buf2.copy()
    ok = b.submit_alac_predict(got2, np.ascontiguousarray(alac_desc(mode2, order2, shift2, bps2)), coeffs2)
    with pytest.raises(SymaccelError) as e:
        b.collect(bad)
    assert e.value.status == -5
    b.collect(ok)
    assert np.array_equal(got2, oracle.alac_predict(buf2, oracle.alac_desc(mode2, order2, shift2, bps2), coeffs2))
    b.close()


def test_emu_alac_streams_share_a_launch(emu_ctx):
    run_alac(emu_ctx)


def test_emu_alac_streams_padded_device_rows(emu_ctx):
    run_alac(emu_ctx, 1024)


# ------------------------------------------------------------------------------------------------ per-ticket status, lanes

def run_bad_aac_blob(ctx):
    """An AAC_DECODE reservation whose blob announces more pairs than it has chains (zero-copy form: nothing checked it on the way in)
    and a reservation released before commit(): both run as empty descriptions, the neighbours decode bit-exactly"""
    import test_aac_tools as T
    from test_aac_js_fused import decode_case
    b = Batcher(ctx, 0)
    bands = b.aac_bands(T.SWB_LONG, T.SWB_SHORT)
    frames = 5
    subs = []
    for s in range(3):
        coeffs, side, delay, pairs, desc, filt, want_pcm, want_delay = decode_case(2100 + s, 1, s % 2, frames, 0.5)
        pcm, d = np.zeros_like(coeffs), delay.copy()
        subs.append((b.submit_aac_decode(bands, coeffs, side, pairs, desc, filt if len(filt) else None, d, pcm), pcm, d, want_pcm, want_delay))
        if s == 0:
            bad, slot = b.reserve(BATCH_AAC_DECODE, bands, 2, frames)
            np.frombuffer((__import__("ctypes").c_char * 16).from_address(slot.input[2]), np.uint32)[:] = [9, 1 << 30, 0, 0]
            b.commit(bad)
        if s == 1:
            gone, _ = b.reserve(BATCH_AAC_DECODE, bands, 2, frames)
            b.release(gone)  # (never committed: zeroed, launched with the group, nobody looks)
    with pytest.raises(SymaccelError) as e:
        b.wait(bad)
    assert e.value.status == -1
    b.release(bad)
    for t, pcm, d, want_pcm, want_delay in subs:
        b.collect(t)
        assert bit_equal(pcm, want_pcm) and bit_equal(d, want_delay)
    st = b.stats()
    assert st["failed_tickets"] == 1 and st["pending"] == 0, st
    b.close()


def test_emu_bad_blob_fails_alone(emu_ctx):
    run_bad_aac_blob(emu_ctx)


def run_lanes(ctx, lanes, n_threads=4, rounds=3):
    """Groups of different shapes closing at the same time from several threads: they are dealt to `lanes` pipelines and enqueued
    outside the batcher's mutex; every result equals the oracle's"""
    b = Batcher(ctx, 0)
    b.configure(lanes=lanes)
    errors = []
    barrier = threading.Barrier(n_threads)

    def work(i):
        try:
            nfr = 2 + i  # a shape (hence a group) per thread
            coeffs, side, delay = aac_case(2, nfr * rounds, 300 + i)
            want = oracle.aac_synth(coeffs, side, delay)
            delay = delay.copy()
            for r in range(rounds):
                pcm = np.zeros((2, nfr, 1024), F)
                t = b.submit(BATCH_AAC_SYNTH, 0, [np.ascontiguousarray(coeffs[:, nfr * r:nfr * (r + 1)]), np.ascontiguousarray(side[:, nfr * r:nfr * (r + 1)])],
                             [delay], pcm)
                barrier.wait()
                b.collect(t)
                if not bit_equal(pcm, want[0][:, nfr * r:nfr * (r + 1)]):
                    errors.append((i, r))
            if not bit_equal(delay, want[1]):
                errors.append((i, "delay"))
        except Exception as e:  # noqa: BLE001
            errors.append((i, repr(e)))
            barrier.abort()
    ths = [threading.Thread(target=work, args=(i,)) for i in range(n_threads)]
    for t in ths:
        t.start()
    for t in ths:
        t.join()
    assert not errors, errors
    st = b.stats()
    assert st["launches"] == n_threads * rounds and st["lanes"] == min(lanes, 8) and st["launch_host_ns"] > 0, st
    assert b.last_error() == ""
    b.close()


@pytest.mark.parametrize("lanes", [1, 3])
def test_emu_groups_are_dealt_to_lanes(emu_ctx, lanes):
    run_lanes(emu_ctx, lanes)


def test_emu_slot_classes_reuse_memory(emu_ctx):
    """tail batches of odd lengths land in the size class of their neighbours: the page-locked pool does not grow with every new shape"""
    b = Batcher(emu_ctx, 0)
    seen = []
    for units in (64, 63, 62, 61, 60, 59, 58, 57):
        t, _ = b.reserve(BATCH_MP3_DECODE, 0, 2, units)
        b.commit(t)
        b.wait(t)
        b.release(t)
        seen.append(b.stats()["staging_bytes"])
    assert seen[-1] == seen[0], seen
    b.close()


@pytest.fixture(scope="module")
def gpu_ctx():
    import torch
    from symphonia_amd import Context
    if not torch.cuda.is_available():
        pytest.fail("no GPU visible")
    ctx = Context(0)
    yield ctx
    ctx.close()


@pytest.mark.gpu
@pytest.mark.parametrize("bs0e,bs1e", [(8, 11), (7, 10), (8, 8), (10, 13)])
def test_gpu_vorbis_decode_streams_share_launches(gpu_ctx, bs0e, bs1e):
    run_vorbis_decode(gpu_ctx, bs0e, bs1e, shapes=((2, 9), (1, 3), (6, 4)), nb=12)


@pytest.mark.gpu
def test_gpu_new_kinds_and_ticket_status(gpu_ctx):
    run_vorbis_decode_bad_ticket(gpu_ctx)
    run_flac(gpu_ctx, 100)
    run_flac(gpu_ctx, 4096)
    run_alac(gpu_ctx, 100)
    run_alac(gpu_ctx, 4096)
    run_flac(gpu_ctx, 1024)
    run_alac(gpu_ctx, 2048)
    run_bad_aac_blob(gpu_ctx)
    run_lanes(gpu_ctx, 1, 6, 4)
    run_lanes(gpu_ctx, 3, 6, 4)
