"""The C-level multi-GPU entry points (csrc/multi.cpp): symaccel_shard_range, symaccel_scatter_streams / symaccel_gather_streams
with a caller-supplied transport (an in-process mailbox: one thread per rank, emulation build, host memory as device memory),
and -- on the GPU box -- the RCCL path itself at world size 1 (library lookup, unique id, communicator, local copies)."""
import ctypes as C
import threading

import numpy as np
import pytest

from emu_lib import emu_library
from symphonia_amd import Context, SymaccelError, _ffi
from symphonia_amd.sharding import shard_streams


def test_shard_range_is_the_python_sharding(tmp_path):
    d = emu_library().dll
    first, count = C.c_size_t(), C.c_size_t()
    for n in (0, 1, 7, 64, 65, 1000):
        for world in (1, 2, 3, 8):
            covered = 0
            for rank in range(world):
                assert d.symaccel_shard_range(n, world, rank, C.byref(first), C.byref(count)) == 0
                b, e = shard_streams(n, world, rank)
                assert (first.value, first.value + count.value) == (b, e)
                assert first.value == covered
                covered += count.value
            assert covered == n
    assert d.symaccel_shard_range(4, 2, 2, C.byref(first), C.byref(count)) == _ffi.ERR_INVALID_ARG
    assert d.symaccel_shard_range(4, 0, 0, C.byref(first), C.byref(count)) == _ffi.ERR_INVALID_ARG


class Mailbox:
    """send / recv between rank threads: what ncclSend / ncclRecv do between GPUs.  `comm` carries the caller's rank."""

    def __init__(self):
        self.box, self.cv, self.log = {}, threading.Condition(), []
        SEND = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p, C.c_void_p)
        GRP = C.CFUNCTYPE(C.c_int)

        def send(buf, nbytes, peer, comm, stream):
            me = C.cast(comm, C.POINTER(C.c_int))[0]
            data = C.string_at(buf, nbytes)
            with self.cv:
                self.box.setdefault((me, peer), []).append(data)
                self.log.append(("send", me, peer, nbytes))
                self.cv.notify_all()
            return 0

        def recv(buf, nbytes, peer, comm, stream):
            me = C.cast(comm, C.POINTER(C.c_int))[0]
            with self.cv:
                ok = self.cv.wait_for(lambda: self.box.get((peer, me)), timeout=30)
                if not ok:
                    return 1
                data = self.box[(peer, me)].pop(0)
            if len(data) != nbytes:
                return 2
            C.memmove(buf, data, nbytes)
            return 0

        self._keep = (SEND(send), SEND(recv), GRP(lambda: 0), GRP(lambda: 0))

        class Transport(C.Structure):
            _fields_ = [("group_start", GRP), ("group_end", GRP), ("send", SEND), ("recv", SEND)]
        self.struct = Transport(self._keep[2], self._keep[3], self._keep[0], self._keep[1])


@pytest.mark.parametrize("world,n_streams,root", [(2, 5, 0), (3, 8, 1), (4, 3, 0), (3, 0, 2)])
def test_scatter_and_gather_over_a_mailbox_transport(world, n_streams, root):
    lib = emu_library()
    d = lib.dll
    mb = Mailbox()
    assert d.symaccel_multi_set_transport(C.byref(mb.struct)) == 0
    try:
        bps = 3 * 16 * 4  # bytes per stream: 3 channels x 16 floats
        rng = np.random.default_rng(world * 10 + n_streams)
        full = rng.integers(0, 255, (n_streams, bps), dtype=np.uint8)
        back = np.zeros_like(full)
        errors = []

        def rank_main(rank):
            try:
                ctx = Context(0, library=lib)
                me = C.c_int(rank)
                b, e = shard_streams(n_streams, world, rank)
                mine = np.zeros((max(e - b, 1), bps), np.uint8)
                all_p = full.ctypes.data if rank == root else None
                ctx._call(d.symaccel_scatter_streams, C.addressof(me), world, rank, root, all_p, mine.ctypes.data, n_streams, bps)
                ctx.sync()
                assert np.array_equal(mine[: e - b], full[b:e]), "rank %d received the wrong slice" % rank
                mine[: e - b] ^= 0x5A  # "decode"
                back_p = back.ctypes.data if rank == root else None
                ctx._call(d.symaccel_gather_streams, C.addressof(me), world, rank, root, mine.ctypes.data, back_p, n_streams, bps)
                ctx.sync()
                ctx.close()
            except Exception as exc:  # noqa: BLE001
                errors.append((rank, repr(exc)))

        threads = [threading.Thread(target=rank_main, args=(r,)) for r in range(world)]
        for t in threads:
            t.start()
        for t in threads:
            t.join(60)
        assert not errors, errors
        assert np.array_equal(back, full ^ 0x5A)
        # the root's own slice never went through the transport; every other non-empty slice did, once each way
        expect = sum(1 for r in range(world) if r != root and shard_streams(n_streams, world, r)[1] > shard_streams(n_streams, world, r)[0])
        assert sum(1 for x in mb.log if x[0] == "send") == 2 * expect
        assert all(root in (x[1], x[2]) for x in mb.log)
    finally:
        assert d.symaccel_multi_set_transport(None) == 0


@pytest.mark.parametrize("world,n_streams,root,n_chunks", [(2, 9, 0, 3), (3, 8, 1, 2), (4, 3, 0, 4), (1, 5, 0, 2), (3, 20, 2, 1)])
def test_pipelined_exchange_over_a_mailbox_transport(world, n_streams, root, n_chunks):
    """symaccel_exchange_pipelined: scatter, the caller's step and gather chunk by chunk (the next chunk's scatter and the previous
    chunk's gather posted around each step); every rank's step sees its slice exactly once, in chunk order, and the gathered
    result is what one scatter -> step -> gather gives."""
    lib = emu_library()
    d = lib.dll
    mb = Mailbox()
    assert d.symaccel_multi_set_transport(C.byref(mb.struct)) == 0
    STEP = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_size_t, C.c_size_t)
    try:
        in_bps, out_bps = 3 * 16, 2 * 16  # bytes per stream in / out
        rng = np.random.default_rng(world * 100 + n_streams)
        full = rng.integers(0, 255, (n_streams, in_bps), dtype=np.uint8)
        back = np.zeros((n_streams, out_bps), np.uint8)
        errors, seen = [], {}

        def rank_main(rank):
            try:
                ctx = Context(0, library=lib)
                me = C.c_int(rank)
                b, e = shard_streams(n_streams, world, rank)
                mine_in = np.zeros((max(e - b, 1), in_bps), np.uint8)
                mine_out = np.zeros((max(e - b, 1), out_bps), np.uint8)
                calls = seen.setdefault(rank, [])

                def step(user, first, count):
                    calls.append((first, count))
                    mine_out[first:first + count] = mine_in[first:first + count, :out_bps] ^ 0x3C  # "decode"
                    return 0
                cb = STEP(step)
                ctx._call(d.symaccel_exchange_pipelined, C.addressof(me), world, rank, root, full.ctypes.data if rank == root else None,
                          mine_in.ctypes.data, in_bps, back.ctypes.data if rank == root else None, mine_out.ctypes.data, out_bps, n_streams,
                          n_chunks, cb, None)
                ctx.sync()
                assert np.array_equal(mine_in[: e - b], full[b:e]), "rank %d received the wrong slice" % rank
                ctx.close()
            except Exception as exc:  # noqa: BLE001
                errors.append((rank, repr(exc)))

        threads = [threading.Thread(target=rank_main, args=(r,)) for r in range(world)]
        for t in threads:
            t.start()
        for t in threads:
            t.join(60)
        assert not errors, errors
        assert np.array_equal(back, full[:, :out_bps] ^ 0x3C)
        for rank in range(world):
            b, e = shard_streams(n_streams, world, rank)
            calls = seen[rank]
            assert sum(c for _, c in calls) == e - b and all(c > 0 for _, c in calls) and len(calls) <= n_chunks
            pos = 0
            for first, count in calls:  # contiguous, in order
                assert first == pos
                pos += count
        assert all(root in (x[1], x[2]) for x in mb.log)
    finally:
        assert d.symaccel_multi_set_transport(None) == 0


def test_pipelined_exchange_reports_a_failing_step():
    lib = emu_library()
    d = lib.dll
    STEP = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_size_t, C.c_size_t)
    ctx = Context(0, library=lib)
    a, b2 = np.zeros((4, 16), np.uint8), np.zeros((4, 16), np.uint8)
    o, o2 = np.zeros((4, 16), np.uint8), np.zeros((4, 16), np.uint8)
    with pytest.raises(SymaccelError) as e:
        ctx._call(d.symaccel_exchange_pipelined, None, 1, 0, 0, a.ctypes.data, b2.ctypes.data, 16, o.ctypes.data, o2.ctypes.data, 16, 4, 2,
                  STEP(lambda u, f, c: 1), None)
    assert e.value.status == _ffi.ERR_DEVICE
    with pytest.raises(SymaccelError) as e:
        ctx._call(d.symaccel_exchange_pipelined, None, 1, 0, 0, a.ctypes.data, b2.ctypes.data, 16, o.ctypes.data, o2.ctypes.data, 16, 4, 0,
                  STEP(lambda u, f, c: 0), None)
    assert e.value.status == _ffi.ERR_INVALID_ARG
    ctx.close()


def test_exchange_argument_checks():
    lib = emu_library()
    d = lib.dll
    ctx = Context(0, library=lib)
    buf = np.zeros(64, np.uint8)
    for args in ((None, 2, 0, 0, buf.ctypes.data, buf.ctypes.data, 4, 16),   # world 2 without a communicator
                 (None, 1, 1, 0, buf.ctypes.data, buf.ctypes.data, 4, 16),   # rank out of range
                 (None, 1, 0, 0, None, buf.ctypes.data, 4, 16),              # the root without its full buffer
                 (None, 1, 0, 0, buf.ctypes.data, None, 4, 16)):             # a rank with streams but no buffer
        with pytest.raises(SymaccelError) as e:
            ctx._call(d.symaccel_scatter_streams, *args)
        assert e.value.status == _ffi.ERR_INVALID_ARG
    # world 1: a local copy, no communicator needed
    src = np.arange(64, dtype=np.uint8)
    dst = np.zeros(64, np.uint8)
    ctx._call(d.symaccel_scatter_streams, None, 1, 0, 0, src.ctypes.data, dst.ctypes.data, 4, 16)
    ctx.sync()
    assert np.array_equal(src, dst)
    ctx.close()


@pytest.mark.gpu
def test_rccl_path_at_world_size_one():
    """On the GPU box: librccl.so is found, a unique id and a one-rank communicator are created on the context's device, scatter
    and gather run (local copies at world size 1), the communicator is destroyed."""
    import torch
    ctx = Context(0)
    ctx.use_torch_stream()
    d = ctx.lib.dll
    uid = (C.c_char * 128)()
    assert d.symaccel_comm_unique_id(C.addressof(uid)) == 0  # (no context argument)
    assert any(b != 0 for b in uid.raw)
    comm = C.c_void_p()
    ctx._call(d.symaccel_comm_init, C.addressof(uid), 1, 0, C.byref(comm))
    assert comm.value
    full = torch.randint(0, 1 << 30, (6, 1024), dtype=torch.int32, device="cuda")
    mine = torch.zeros_like(full)
    back = torch.zeros_like(full)
    ctx._call(d.symaccel_scatter_streams, comm, 1, 0, 0, full.data_ptr(), mine.data_ptr(), 6, 4096)
    ctx._call(d.symaccel_gather_streams, comm, 1, 0, 0, mine.data_ptr(), back.data_ptr(), 6, 4096)
    torch.cuda.synchronize()
    assert torch.equal(full, mine) and torch.equal(full, back)
    assert d.symaccel_comm_destroy(comm) == 0
    ctx.close()


@pytest.mark.gpu
@pytest.mark.parametrize("world", [2, 5, 8])
def test_multi_selftest_on_one_gpu(world):
    """symphonia_amd/selftest.py (also `python bench.py --selftest-multi WORLD`): the RCCL binding at world size 1 and the scatter ->
    synthesis -> gather leg with `world` in-process ranks (one context + one HIP stream each on cuda:0) over a caller-supplied
    transport on DEVICE buffers; the gathered PCM equals the PCM of the whole batch decoded in one call."""
    from symphonia_amd.selftest import multi_selftest
    r = multi_selftest(world)
    assert r["rccl_world1"]["ok"] and r["in_process_ranks"]["ok"] and len(r["in_process_ranks"]["per_rank"]) == world
    # ... and the same leg through symaccel_exchange_pipelined (three chunks, the product's AAC synthesis as the step)
    assert r["pipelined_exchange"]["ok"] and sum(r["pipelined_exchange"]["steps_per_rank"]) >= world
