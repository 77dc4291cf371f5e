"""The cross-stream batcher (csrc/batcher.cpp, `symaccel_batcher_*`): many streams submit their look-ahead batches, ONE launch
per (kind, units per chain) group serves all of them, and every stream gets back bit for bit what its own
symaccel_aac_synth / symaccel_mp3_synth / symaccel_mp3_decode_pipelined call would have given (the oracle's result) -- across
consecutive batches (the carried state), ragged groups, a flush forced by `flush_bytes`, concurrent submitters, abandoned
tickets and the zero-copy reserve / commit / wait / release form.  CPU: the emulation build; GPU: libsymaccel.so."""
import threading

import numpy as np
import pytest

import oracle
from emu_lib import emu_ctx  # noqa: F401
from helpers import bit_equal
from symphonia_amd import BATCH_AAC_SYNTH, BATCH_MP3_DECODE, BATCH_MP3_SYNTH, BATCH_VORBIS_SYNTH, Batcher, mp3_side
from test_staging import aac_case

F = np.float32


def aac_streams(n_streams, frames, seed):
    """(coeffs, side, delay, want_pcm per batch, want_delay per batch) per stream; frames = list of batch lengths"""
    out = []
    for s in range(n_streams):
        nch = 1 + (s % 2)
        coeffs, side, delay = aac_case(nch, sum(frames), seed + s)
        want_pcm, want_delay = oracle.aac_synth(coeffs, side, delay)
        out.append((coeffs, side, delay, want_pcm, want_delay))
    return out


def run_aac_streams(ctx, flush_bytes, n_streams, frames):
    b = Batcher(ctx, flush_bytes)
    streams = aac_streams(n_streams, frames, 40)
    delays = [np.ascontiguousarray(s[2].copy()) for s in streams]
    t0 = 0
    for nf in frames:
        tickets, pcms = [], []
        for s, (coeffs, side, _, _, _) in enumerate(streams):
            pcm = np.zeros((coeffs.shape[0], nf, 1024), F)
            tickets.append(b.submit(BATCH_AAC_SYNTH, 0, [np.ascontiguousarray(coeffs[:, t0:t0 + nf]), np.ascontiguousarray(side[:, t0:t0 + nf])],
                                    [delays[s]], pcm))
            pcms.append(pcm)
        for s in reversed(range(n_streams)):  # (any order)
            b.collect(tickets[s])
            assert bit_equal(pcms[s], streams[s][3][:, t0:t0 + nf]), (s, t0)
        t0 += nf
    for s in range(n_streams):
        assert bit_equal(delays[s], streams[s][4]), s
    st = b.stats()
    b.close()
    return st


def test_emu_aac_streams_share_launches(emu_ctx):
    st = run_aac_streams(emu_ctx, 0, 5, [4, 4, 4])
    assert st["submissions"] == 15 and st["launches"] == 3 and st["max_chains_per_launch"] == 7 and st["pending"] == 0


def test_emu_aac_flush_bytes_cuts_groups(emu_ctx):
    # 4 frames x 4 KiB x (1 or 2 chains): a group is launched as soon as 40 KiB of input are in it
    st = run_aac_streams(emu_ctx, 40 << 10, 6, [4, 4])
    assert st["submissions"] == 12 and st["launches"] > 2 and st["max_chains_per_launch"] <= 4


def test_emu_ragged_batches_form_groups_per_length(emu_ctx):
    b = Batcher(emu_ctx, 0)
    lens = [3, 5, 3, 1, 5]
    cases = [aac_case(2, n, 70 + i) for i, n in enumerate(lens)]
    want = [oracle.aac_synth(*c) for c in cases]
    delays = [c[2].copy() for c in cases]
    pcms = [np.zeros((2, n, 1024), F) for n in lens]
    tickets = [b.submit(BATCH_AAC_SYNTH, 0, [c[0], c[1]], [delays[i]], pcms[i]) for i, c in enumerate(cases)]
    assert b.stats()["pending"] == 5
    b.collect(tickets[3])  # one waiter: everything pending goes, three groups
    assert b.stats()["launches"] == 3 and b.stats()["pending"] == 0
    for i in (0, 1, 2, 4):
        b.collect(tickets[i])
    for i in range(5):
        assert bit_equal(pcms[i], want[i][0]) and bit_equal(delays[i], want[i][1]), i
    b.close()


def mp3_streams(rng, n, ngr):
    from test_emu_codecs import mp3_case
    out = []
    for s in range(n):
        nch = 1 + (s % 2)
        xr, bt, mx, rz = mp3_case(rng, nch, ngr)
        ov = rng.standard_normal((nch, 576)).astype(F)
        vv = rng.standard_normal((nch, 1024)).astype(F)
        vf = rng.integers(0, 16, nch).astype(np.int32)
        want = oracle.mp3_synth(xr, oracle.mp3_side(bt, mx, rz), 2, ov, vv, vf)
        out.append((xr, np.ascontiguousarray(mp3_side(bt, mx, rz)), ov.copy(), vv.copy(), vf.copy(), want))
    return out


def run_mp3_synth(ctx):
    rng = np.random.default_rng(9)
    b = Batcher(ctx, 0)
    streams = mp3_streams(rng, 4, 6)
    pcms, tickets = [], []
    for xr, side, ov, vv, vf, _ in streams:
        pcm = np.zeros(xr.shape, F)
        tickets.append(b.submit(BATCH_MP3_SYNTH, 2, [xr, side], [ov, vv, vf], pcm))
        pcms.append(pcm)
    for t in tickets:
        b.collect(t)
    assert b.stats()["launches"] == 1
    for (xr, side, ov, vv, vf, want), pcm in zip(streams, pcms):
        assert bit_equal(pcm, np.asarray(want[0])) and bit_equal(ov, np.asarray(want[1])) and bit_equal(vv, np.asarray(want[2]))
        assert np.array_equal(vf, np.asarray(want[3]))
    b.close()


def test_emu_mp3_synth_streams(emu_ctx):
    run_mp3_synth(emu_ctx)


def run_mp3_decode(ctx, n_streams=5, granules=6):
    """One submission per stream (a joint-stereo pair or a mono stream): int16 samples + records in, PCM out; expectation = the
    oracle's requantize -> stereo -> synthesis of the same stream."""
    from test_mp3_stereo import fused_case, side_of
    b = Batcher(ctx, 0)
    subs = []
    for s in range(n_streams):
        q, rd, pairs, sd, xr_want, mono = fused_case(300 + s, 1, 1, granules)  # chains: one pair + one mono
        side = side_of(rd, pairs, sd)
        rng = np.random.default_rng(500 + s)
        chains = [int(pairs[0][0]), int(pairs[0][1])] if s % 3 != 2 else [int(mono)]
        ov = rng.standard_normal((len(chains), 576)).astype(F)
        vv = rng.standard_normal((len(chains), 1024)).astype(F)
        vf = rng.integers(0, 16, len(chains)).astype(np.int32)
        want = oracle.mp3_synth(np.ascontiguousarray(xr_want[chains]), np.ascontiguousarray(side[chains]), 1, ov, vv, vf)
        pcm = np.zeros((len(chains), granules, 576), F)
        ins = [np.ascontiguousarray(q[chains]), np.ascontiguousarray(rd[chains]), np.ascontiguousarray(side[chains]),
               np.ascontiguousarray(sd[0]) if len(chains) == 2 else None]
        t = b.submit(BATCH_MP3_DECODE, 1, ins, [ov, vv, vf], pcm)
        subs.append((t, pcm, ov, vv, vf, want))
    for t, pcm, ov, vv, vf, want in subs:
        b.collect(t)
        assert bit_equal(pcm, np.asarray(want[0])), "pcm"
        assert bit_equal(ov, np.asarray(want[1])) and bit_equal(vv, np.asarray(want[2])) and np.array_equal(vf, np.asarray(want[3]))
    assert b.stats()["launches"] == 1 and b.stats()["submissions"] == n_streams
    b.close()


def test_emu_mp3_decode_streams(emu_ctx):
    run_mp3_decode(emu_ctx)


def run_vorbis(ctx, bs0e, bs1e, n_streams=5, nb=11):
    """Streams with DIFFERENT block flags (different packed lengths) in one launch: every chain's planes at their largest,
    nb * bs1 / 2, the packed data at the front; expectation = the oracle on each stream's own strides."""
    from test_emu_codecs import vorbis_case
    b = Batcher(ctx, 0)
    half = (1 << bs1e) // 2
    cap = nb * half
    subs = []
    for s in range(n_streams):
        rng = np.random.default_rng(700 + s)
        nch = 1 + (s % 3)
        flags, prev, spectra, overlap, pcm_stride = vorbis_case(rng, bs0e, bs1e, nch, nb, p_long=(0.1, 0.5, 0.9)[s % 3])
        want = oracle.vorbis_synth(bs0e, bs1e, spectra, flags, prev, overlap, pcm_stride)
        lay = oracle.vorbis_layout(bs0e, bs1e, flags, prev)
        spec = np.zeros((nch, cap), F)
        spec[:, :spectra.shape[1]] = spectra
        pcm = np.full((nch, cap), np.nan, F)
        pv, ov = prev.copy(), overlap.copy()
        t = b.submit(BATCH_VORBIS_SYNTH, bs0e | (bs1e << 8), [spec, np.ascontiguousarray(flags)], [pv, ov], pcm.reshape(nch, nb, half))
        subs.append((t, pcm, pv, ov, want, lay, prev, flags, pcm_stride))
    assert b.stats()["pending"] == n_streams
    for t, pcm, pv, ov, want, lay, prev, flags, pcm_stride in subs:
        b.collect(t)
        for c in range(pcm.shape[0]):
            # the samples the chain's blocks emit (a first block after a reset owns slots nothing writes: lib.rs:298-303)
            start = int(lay[1][c, 1]) if prev[c] < 0 else 0
            end = int(lay[1][c, -1])
            assert bit_equal(pcm[c, start:end], np.asarray(want[0])[c, start:end]), c
        assert bit_equal(ov, np.asarray(want[1])) and np.array_equal(pv, np.asarray(want[2]))
    assert b.stats()["launches"] == 1
    with pytest.raises(Exception):
        b.reserve(BATCH_VORBIS_SYNTH, 5 | (11 << 8), 1, 4)   # 32-sample blocks do not exist
    with pytest.raises(Exception):
        b.reserve(BATCH_VORBIS_SYNTH, 11 | (8 << 8), 1, 4)   # bs0 > bs1
    b.close()


@pytest.mark.parametrize("bs0e,bs1e", [(8, 11), (6, 9)])
def test_emu_vorbis_streams_with_different_flags_share_a_launch(emu_ctx, bs0e, bs1e):
    run_vorbis(emu_ctx, bs0e, bs1e)


def run_aac_decode(ctx, n_streams=5, frames=7, seed0=900):
    """symaccel_aac_decode_pipelined's work for many streams in ONE launch: every stream its own channel count, pairs (in any chain
    order), joint-stereo descriptors and TNS filters; expectation = the reference's order per stream (joint stereo, TNS, Dsp::synth:
    tests/test_aac_js_fused.py decode_case).  The pair frames with TNS take the list pass, every other frame the fused walk."""
    import test_aac_tools as T
    from test_aac_js_fused import decode_case
    b = Batcher(ctx, 0)
    bands = b.aac_bands(T.SWB_LONG, T.SWB_SHORT)
    assert b.aac_bands(T.SWB_LONG, T.SWB_SHORT) == bands  # the same tables: the same index
    subs = []
    shapes = [(1, 0, 0.3), (2, 1, 0.5), (0, 2, 0.4), (3, 0, 0.0), (1, 1, 1.0), (0, 1, 0.0), (2, 0, 0.7)]
    for s in range(n_streams):
        n_pairs, extra, p_tns = shapes[s % len(shapes)]
        coeffs, side, delay, pairs, desc, filt, want_pcm, want_delay = decode_case(seed0 + s, n_pairs, extra, frames, p_tns)
        pcm, d = np.zeros_like(coeffs), delay.copy()
        before = coeffs.copy()
        t = b.submit_aac_decode(bands, coeffs, side, pairs if n_pairs else None, desc if n_pairs else None, filt if len(filt) else None, d, pcm)
        assert bit_equal(coeffs, before)
        subs.append((t, pcm, d, want_pcm, want_delay))
    assert b.stats()["pending"] == n_streams
    for t, pcm, d, want_pcm, want_delay in subs:
        b.collect(t)
        assert bit_equal(pcm, want_pcm) and bit_equal(d, want_delay)
    assert b.stats()["launches"] == 1
    # what the typed entry refuses: a chain in two pairs, a pair outside the stream, tables that were never registered
    coeffs, side, delay, pairs, desc, filt, _, _ = decode_case(1, 1, 0, frames, 0.5)
    pcm = np.zeros_like(coeffs)
    with pytest.raises(Exception):
        b.submit_aac_decode(bands, coeffs, side, np.array([[0, 1], [1, 0]], np.int32), np.zeros((2, frames), desc.dtype), None, delay.copy(), pcm)
    with pytest.raises(Exception):
        b.submit_aac_decode(bands, coeffs, side, np.array([[0, 2]], np.int32), desc, None, delay.copy(), pcm)
    with pytest.raises(Exception):
        b.submit_aac_decode(bands + 7, coeffs, side, None, None, None, delay.copy(), pcm)
    assert b.stats()["pending"] == 0
    b.close()


def test_emu_aac_decode_streams_share_a_launch(emu_ctx):
    run_aac_decode(emu_ctx)


def run_zero_copy(ctx):
    b = Batcher(ctx, 0)
    cases = [aac_case(2, 4, 90 + i) for i in range(3)]
    want = [oracle.aac_synth(*c) for c in cases]
    tickets = []
    for coeffs, side, delay in cases:
        t, slot = b.reserve(BATCH_AAC_SYNTH, 0, 2, 4)
        ins, sts, _ = Batcher.slot_arrays(slot, BATCH_AAC_SYNTH, 2, 4)
        ins[0][:], ins[1][:], sts[0][:] = coeffs, side, delay
        b.commit(t)
        tickets.append(t)
    abandoned, _ = b.reserve(BATCH_AAC_SYNTH, 0, 1, 4)  # reserved, never filled: released without a commit
    b.release(abandoned)
    for i, t in enumerate(tickets):
        slot = b.wait(t)
        _, sts, out = Batcher.slot_arrays(slot, BATCH_AAC_SYNTH, 2, 4)
        assert bit_equal(out, want[i][0]) and bit_equal(sts[0], want[i][1]), i
        b.release(t)
    with pytest.raises(Exception):
        b.wait(tickets[0])  # released: the ticket is gone
    with pytest.raises(Exception):
        b.reserve(77, 0, 1, 1)
    with pytest.raises(Exception):
        b.reserve(BATCH_MP3_DECODE, 0, 3, 2)  # one stream per submission
    with pytest.raises(Exception):
        b.reserve(BATCH_MP3_SYNTH, 9, 1, 2)
    # the slot pool: the next round of the same shape allocates nothing
    staging = b.stats()["staging_bytes"]
    t, slot = b.reserve(BATCH_AAC_SYNTH, 0, 2, 4)
    b.commit(t)
    b.wait(t)
    b.release(t)
    assert b.stats()["staging_bytes"] == staging
    b.close()


def test_emu_zero_copy_slots_and_errors(emu_ctx):
    run_zero_copy(emu_ctx)


def run_threads(ctx, n_threads=4, rounds=3):
    """submitters on several threads, every thread a stream of its own: whoever waits first launches for all"""
    b = Batcher(ctx, 0)
    cases = [aac_case(2, 3 * rounds, 120 + i) for i in range(n_threads)]
    want = [oracle.aac_synth(*c) for c in cases]
    errors = []
    barrier = threading.Barrier(n_threads)

    def work(i):
        try:
            coeffs, side, delay = cases[i]
            delay = delay.copy()
            for r in range(rounds):
                pcm = np.zeros((2, 3, 1024), F)
                t = b.submit(BATCH_AAC_SYNTH, 0, [np.ascontiguousarray(coeffs[:, 3 * r:3 * r + 3]), np.ascontiguousarray(side[:, 3 * r:3 * r + 3])],
                             [delay], pcm)
                barrier.wait()
                b.collect(t)
                if not bit_equal(pcm, want[i][0][:, 3 * r:3 * r + 3]):
                    errors.append((i, r))
            if not bit_equal(delay, want[i][1]):
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
    assert st["submissions"] == n_threads * rounds and st["launches"] == rounds
    b.close()


def test_emu_concurrent_submitters(emu_ctx):
    run_threads(emu_ctx)


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
def test_gpu_batcher_aac_streams(gpu_ctx):
    st = run_aac_streams(gpu_ctx, 0, 9, [8, 8, 3])
    assert st["launches"] == 3 and st["max_chains_per_launch"] == 13
    st = run_aac_streams(gpu_ctx, 256 << 10, 12, [8, 8])  # several chunks per group / several groups per round
    assert st["launches"] >= 4


@pytest.mark.gpu
def test_gpu_batcher_mp3_and_zero_copy_and_threads(gpu_ctx):
    run_mp3_synth(gpu_ctx)
    run_mp3_decode(gpu_ctx, 7, 12)
    for pair in ((8, 11), (7, 10), (12, 13)):
        run_vorbis(gpu_ctx, *pair, n_streams=9, nb=14)
    run_aac_decode(gpu_ctx, n_streams=14, frames=33, seed0=1200)
    run_aac_decode(gpu_ctx, n_streams=40, frames=9, seed0=1300)
    run_zero_copy(gpu_ctx)
    run_threads(gpu_ctx, 6, 4)


@pytest.mark.gpu
def test_gpu_batcher_large_group_is_chunked(gpu_ctx):
    """512 chains x 16 frames = 32 MiB of spectra in one group: two overlapped chunks (round 6: half a full group per chunk, three measured
    slower -- profiles/r06z5_copy_grid.jsonl), same bits as one symaccel_aac_synth call"""
    from symphonia_amd import AacDsp
    rng = np.random.default_rng(3)
    n_streams, nfr = 256, 16
    coeffs = (rng.standard_normal((n_streams, 2, nfr, 1024)) * 50).astype(F)
    side = np.full((n_streams, 2, nfr), 0 | (1 << 2) | (1 << 3), np.uint8)
    delay = rng.standard_normal((n_streams, 2, 1024)).astype(F)
    want_pcm, want_delay = AacDsp(gpu_ctx).synth(coeffs.reshape(-1, nfr, 1024), side.reshape(-1, nfr), delay.reshape(-1, 1024))
    b = Batcher(gpu_ctx, 0)
    pcm = np.zeros_like(coeffs)
    got_delay = delay.copy()
    tickets = [b.submit(BATCH_AAC_SYNTH, 0, [coeffs[s], side[s]], [got_delay[s]], pcm[s]) for s in range(n_streams)]
    for t in tickets:
        b.collect(t)
    st = b.stats()
    assert st["launches"] == 1 and st["chunks"] >= 2 and st["max_chains_per_launch"] == 512 and st["staging_bytes"] <= 3 * (64 << 20)
    assert bit_equal(pcm.reshape(-1, nfr, 1024), np.asarray(want_pcm)) and bit_equal(got_delay.reshape(-1, 1024), np.asarray(want_delay))
    b.close()


def run_fuzz(ctx, seed, rounds=4, n_threads=3):
    """Random traffic: threads submitting AAC / MP3 / Vorbis batches of random shapes in random order, collecting in random order,
    abandoning some -- every collected result equals the oracle's, nothing is lost, the pool is quiet at the end."""
    from test_emu_codecs import mp3_case, vorbis_case
    b = Batcher(ctx, int(np.random.default_rng(seed).choice([0, 96 << 10, 1 << 20])))
    errors, done = [], []
    lock = threading.Lock()

    def work(tid):
        rng = np.random.default_rng(seed * 100 + tid)
        try:
            for r in range(rounds):
                subs = []
                for _ in range(int(rng.integers(1, 5))):
                    kind = int(rng.choice([BATCH_AAC_SYNTH, BATCH_MP3_SYNTH, BATCH_VORBIS_SYNTH]))
                    nch, units = int(rng.integers(1, 4)), int(rng.choice([1, 2, 3, 5, 8]))
                    if kind == BATCH_AAC_SYNTH:
                        coeffs, side, delay = aac_case(nch, units, int(rng.integers(1 << 30)))
                        want = oracle.aac_synth(coeffs, side, delay)
                        pcm, st = np.zeros((nch, units, 1024), F), [delay.copy()]
                        t = b.submit(kind, 0, [coeffs, side], st, pcm)
                        check = lambda pcm=pcm, st=st, want=want: bit_equal(pcm, want[0]) and bit_equal(st[0], want[1])
                    elif kind == BATCH_MP3_SYNTH:
                        xr, bt, mx, rz = mp3_case(rng, nch, units)
                        ov, vv = rng.standard_normal((nch, 576)).astype(F), rng.standard_normal((nch, 1024)).astype(F)
                        vf = rng.integers(0, 16, nch).astype(np.int32)
                        sr = int(rng.integers(0, 9))
                        want = oracle.mp3_synth(xr, oracle.mp3_side(bt, mx, rz), sr, ov, vv, vf)
                        pcm, st = np.zeros(xr.shape, F), [ov.copy(), vv.copy(), vf.copy()]
                        t = b.submit(kind, sr, [xr, np.ascontiguousarray(mp3_side(bt, mx, rz))], st, pcm)
                        check = lambda pcm=pcm, st=st, want=want: (bit_equal(pcm, np.asarray(want[0])) and bit_equal(st[0], np.asarray(want[1]))
                                                                   and bit_equal(st[1], np.asarray(want[2])) and np.array_equal(st[2], np.asarray(want[3])))
                    else:
                        e0, e1 = [(8, 11), (6, 9), (7, 10)][int(rng.integers(0, 3))]
                        half = (1 << e1) // 2
                        flags, prev, spectra, overlap, pcm_stride = vorbis_case(rng, e0, e1, nch, units, p_long=float(rng.random()))
                        want = oracle.vorbis_synth(e0, e1, spectra, flags, prev, overlap, pcm_stride)
                        lay = oracle.vorbis_layout(e0, e1, flags, prev)
                        spec = np.zeros((nch, units * half), F)
                        spec[:, :spectra.shape[1]] = spectra
                        pcm, st = np.zeros((nch, units, half), F), [prev.copy(), overlap.copy()]
                        t = b.submit(kind, e0 | (e1 << 8), [spec, np.ascontiguousarray(flags)], st, pcm)

                        def check(pcm=pcm, st=st, want=want, lay=lay, prev=prev, nch=nch):
                            flat = pcm.reshape(nch, -1)
                            ok = bit_equal(st[1], np.asarray(want[1])) and np.array_equal(st[0], np.asarray(want[2]))
                            for c in range(nch):
                                a = int(lay[1][c, 1]) if prev[c] < 0 else 0
                                ok = ok and bit_equal(flat[c, a:int(lay[1][c, -1])], np.asarray(want[0])[c, a:int(lay[1][c, -1])])
                            return ok
                    subs.append((t, check, kind))
                order = rng.permutation(len(subs))
                for i in order:
                    t, check, kind = subs[i]
                    if rng.random() < 0.15:
                        ctx.lib.check(b.dll.symaccel_batcher_abandon(b.handle, t), ctx.handle)
                        continue
                    b.collect(t)
                    if not check():
                        errors.append((tid, r, kind))
                    with lock:
                        done.append(t)
        except Exception as e:  # noqa: BLE001
            errors.append((tid, repr(e)))
    ths = [threading.Thread(target=work, args=(i,)) for i in range(n_threads)]
    for t in ths:
        t.start()
    for t in ths:
        t.join()
    assert not errors, errors[:5]
    st = b.stats()
    assert st["pending"] == 0 and st["launches"] > 0 and len(done) > 0
    b.close()


@pytest.mark.parametrize("seed", [1, 2])
def test_emu_batcher_fuzz(emu_ctx, seed):
    run_fuzz(emu_ctx, seed, rounds=2, n_threads=3)


@pytest.mark.gpu
@pytest.mark.parametrize("seed", [3, 4, 5])
def test_gpu_batcher_fuzz(gpu_ctx, seed):
    run_fuzz(gpu_ctx, seed, rounds=6, n_threads=6)
