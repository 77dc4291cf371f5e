"""One-GPU self-test of the N > 1 C path (csrc/multi.cpp), for boxes where only one GPU can be reached.

`python bench.py --selftest-multi [WORLD]` and tests/test_multi_c.py::test_multi_selftest_on_one_gpu run it:
  1. RCCL through the library's own binding (dlopen of librccl.so, no header): symaccel_comm_unique_id, symaccel_comm_init at world
     size 1 on cuda:0, scatter + gather (local copies), symaccel_comm_destroy -- the library lookup, the by-value 128-byte id
     and the communicator handle are what a first 8-GPU run would otherwise meet for the first time;
  2. the scatter -> synthesis -> gather leg with WORLD in-process "ranks" (one thread + one symaccel context + one HIP stream
     each, all on cuda:0) over a caller-supplied transport (symaccel_multi_set_transport) that moves DEVICE buffers through a
     host mailbox: every rank decodes its shard of an AAC batch with the product kernel and the root's gathered PCM must equal
     the PCM of the whole batch decoded in one call -- the sharding arithmetic, the root-centric exchange and the per-rank
     streams of the real path, minus xGMI.
Returns a dict for the bench line; raises on any mismatch."""
import ctypes as C
import threading
import time

import numpy as np


class DeviceMailbox:
    """symaccel_transport over a host mailbox: send = stream sync + D2H into the box, recv = wait + H2D.  `comm` carries the rank."""

    def __init__(self, hip):
        self.hip, self.box, self.cv, self.log = hip, {}, threading.Condition(), []
        SEND = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p, C.c_void_p)
        GRP = C.CFUNCTYPE(C.c_int)

        def send(buf, nbytes, peer, comm, stream):
            me = C.cast(comm, C.POINTER(C.c_int))[0]
            host = (C.c_char * nbytes)()
            if hip.hipStreamSynchronize(C.c_void_p(stream)) != 0 or hip.hipMemcpy(host, C.c_void_p(buf), C.c_size_t(nbytes), 2) != 0:  # D2H
                return 1
            with self.cv:
                self.box.setdefault((me, peer), []).append(host)
                self.log.append((me, peer, nbytes))
                self.cv.notify_all()
            return 0

        def recv(buf, nbytes, peer, comm, stream):
            me = C.cast(comm, C.POINTER(C.c_int))[0]
            with self.cv:
                if not self.cv.wait_for(lambda: self.box.get((peer, me)), timeout=60):
                    return 1
                host = self.box[(peer, me)].pop(0)
            if len(host) != nbytes:
                return 2
            return 0 if hip.hipMemcpy(C.c_void_p(buf), host, C.c_size_t(nbytes), 1) == 0 else 3  # H2D

        self._keep = (SEND(send), SEND(recv), GRP(lambda: 0), GRP(lambda: 0))

        class Transport(C.Structure):
            _fields_ = [("group_start", GRP), ("group_end", GRP), ("send", SEND), ("recv", SEND)]
        self.struct = Transport(self._keep[2], self._keep[3], self._keep[0], self._keep[1])


def multi_selftest(world=4, streams=None, frames=24):
    import torch
    import symphonia_amd as sa
    from symphonia_amd.sharding import shard_streams
    assert torch.cuda.is_available(), "the self-test needs cuda:0"
    out = {"world": int(world)}
    lib = sa._ffi.default_library()
    d = lib.dll
    # ---- 1. RCCL at world size 1 through the C binding
    ctx = sa.Context(0)
    ctx.use_torch_stream()
    t0 = time.perf_counter()
    uid = (C.c_char * 128)()
    st = d.symaccel_comm_unique_id(C.addressof(uid))
    if st != 0:
        raise RuntimeError("symaccel_comm_unique_id failed (%d): is librccl.so on the loader path?" % st)
    comm = C.c_void_p()
    ctx._call(d.symaccel_comm_init, C.addressof(uid), 1, 0, C.byref(comm))
    full = torch.randint(0, 1 << 30, (5, 2048), dtype=torch.int32, device="cuda")
    mine, back = torch.zeros_like(full), torch.zeros_like(full)
    ctx._call(d.symaccel_scatter_streams, comm, 1, 0, 0, full.data_ptr(), mine.data_ptr(), 5, 8192)
    ctx._call(d.symaccel_gather_streams, comm, 1, 0, 0, mine.data_ptr(), back.data_ptr(), 5, 8192)
    torch.cuda.synchronize()
    if not (torch.equal(full, mine) and torch.equal(full, back)):
        raise RuntimeError("world-1 scatter / gather did not copy")
    assert d.symaccel_comm_destroy(comm) == 0
    try:
        ver = torch.cuda.nccl.version()
    except Exception:  # noqa: BLE001
        ver = None
    out["rccl_world1"] = {"ok": True, "seconds": time.perf_counter() - t0, "rccl_version": list(ver) if ver else None}
    # ---- 2. WORLD in-process ranks on this device over the mailbox transport
    streams = streams if streams is not None else 2 * world + 1  # (an uneven split: some ranks get one stream more)
    nch = 2
    g = torch.Generator(device="cuda").manual_seed(11)
    coeffs = torch.randn((streams, nch, frames, 1024), generator=g, device="cuda")
    side = torch.full((streams, nch, frames), int(sa.aac_side(0, 1, 1)), dtype=torch.uint8, device="cuda")
    delay = torch.zeros((streams * nch, 1024), device="cuda")
    want = sa.AacDsp(ctx).synth(coeffs.view(streams * nch, frames, 1024), side.view(streams * nch, frames), delay.clone())
    torch.cuda.synchronize()
    gathered = torch.zeros_like(coeffs)
    hip = C.CDLL("libamdhip64.so")
    mb = DeviceMailbox(hip)
    assert d.symaccel_multi_set_transport(C.byref(mb.struct)) == 0
    errors, per_rank = [], [None] * world
    bytes_per_stream = nch * frames * 1024 * 4
    try:
        def rank_main(rank):
            try:
                torch.cuda.set_device(0)
                s = torch.cuda.Stream()
                with torch.cuda.stream(s):
                    c = sa.Context(0)
                    c.use_torch_stream()
                    me = C.c_int(rank)
                    b, e = shard_streams(streams, world, rank)
                    n = max(e - b, 1)
                    my_in = torch.zeros((n, nch, frames, 1024), device="cuda")
                    my_side = side[b:b + n].contiguous() if e > b else torch.zeros((n, nch, frames), dtype=torch.uint8, device="cuda")
                    t1 = time.perf_counter()
                    c._call(d.symaccel_scatter_streams, C.addressof(me), world, rank, 0, coeffs.data_ptr() if rank == 0 else None,
                            my_in.data_ptr(), streams, bytes_per_stream)
                    my_out = sa.AacDsp(c).synth(my_in.view(n * nch, frames, 1024), my_side.view(n * nch, frames),
                                                torch.zeros((n * nch, 1024), device="cuda"))
                    c._call(d.symaccel_gather_streams, C.addressof(me), world, rank, 0, my_out.data_ptr(),
                            gathered.data_ptr() if rank == 0 else None, streams, bytes_per_stream)
                    s.synchronize()
                    per_rank[rank] = {"streams": [int(b), int(e)], "ms": (time.perf_counter() - t1) * 1e3}
                    c.close()
            except Exception as exc:  # noqa: BLE001
                errors.append((rank, repr(exc)))
        threads = [threading.Thread(target=rank_main, args=(r,)) for r in range(world)]
        for t in threads:
            t.start()
        for t in threads:
            t.join(120)
        if errors or any(t.is_alive() for t in threads):
            raise RuntimeError("rank threads failed: %r" % (errors or "timeout",))
    finally:
        assert d.symaccel_multi_set_transport(None) == 0
    torch.cuda.synchronize()
    if not torch.equal(gathered.view(streams * nch, frames, 1024).view(torch.int32), want.view(torch.int32)):
        raise RuntimeError("the gathered PCM of %d in-process ranks differs from the one-call PCM" % world)
    expect = sum(1 for r in range(1, world) if shard_streams(streams, world, r)[1] > shard_streams(streams, world, r)[0])
    if len(mb.log) != 2 * expect or not all(0 in (a, b) for a, b, _ in mb.log):
        raise RuntimeError("unexpected transfers: %r" % (mb.log,))
    out["in_process_ranks"] = {"ok": True, "streams": streams, "transfers": len(mb.log), "per_rank": per_rank,
                               "transport": "symaccel_multi_set_transport: device buffers through a host mailbox (no xGMI on a one-GPU box)"}
    # ---- 3. the same leg chunked and overlapped: symaccel_exchange_pipelined, the step = the product's AAC synthesis on a chunk
    # of the rank's streams (chains are independent, so a chunk needs no state from the one before it)
    n_chunks = 3
    gathered2 = torch.zeros_like(coeffs)
    mb2 = DeviceMailbox(hip)
    assert d.symaccel_multi_set_transport(C.byref(mb2.struct)) == 0
    STEP = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_size_t, C.c_size_t)
    errors, steps_seen = [], [0] * world
    try:
        def rank_main2(rank):
            try:
                torch.cuda.set_device(0)
                s = torch.cuda.Stream()
                with torch.cuda.stream(s):
                    c = sa.Context(0)
                    c.use_torch_stream()
                    me = C.c_int(rank)
                    b, e = shard_streams(streams, world, rank)
                    n = max(e - b, 1)
                    my_in = torch.zeros((n, nch, frames, 1024), device="cuda")
                    my_out = torch.zeros((n, nch, frames, 1024), device="cuda")
                    my_side = side[b:b + n].contiguous() if e > b else torch.zeros((n, nch, frames), dtype=torch.uint8, device="cuda")
                    dsp = sa.AacDsp(c)

                    def step(user, first, count):
                        try:
                            with torch.cuda.stream(s):
                                dsp.synth(my_in[first:first + count].view(count * nch, frames, 1024), my_side[first:first + count].view(count * nch, frames),
                                          torch.zeros((count * nch, 1024), device="cuda"), pcm=my_out[first:first + count].view(count * nch, frames, 1024))
                            steps_seen[rank] += 1
                            return 0
                        except Exception as exc:  # noqa: BLE001
                            errors.append((rank, "step: " + repr(exc)))
                            return 1
                    cb = STEP(step)
                    c._call(d.symaccel_exchange_pipelined, C.addressof(me), world, rank, 0, coeffs.data_ptr() if rank == 0 else None,
                            my_in.data_ptr(), bytes_per_stream, gathered2.data_ptr() if rank == 0 else None, my_out.data_ptr(), bytes_per_stream,
                            streams, n_chunks, cb, None)
                    s.synchronize()
                    c.close()
            except Exception as exc:  # noqa: BLE001
                errors.append((rank, repr(exc)))
        threads = [threading.Thread(target=rank_main2, args=(r,)) for r in range(world)]
        for t in threads:
            t.start()
        for t in threads:
            t.join(120)
        if errors or any(t.is_alive() for t in threads):
            raise RuntimeError("pipelined exchange: rank threads failed: %r" % (errors or "timeout",))
    finally:
        assert d.symaccel_multi_set_transport(None) == 0
    torch.cuda.synchronize()
    if not torch.equal(gathered2.view(streams * nch, frames, 1024).view(torch.int32), want.view(torch.int32)):
        raise RuntimeError("the PCM gathered by symaccel_exchange_pipelined differs from the one-call PCM")
    out["pipelined_exchange"] = {"ok": True, "chunks": n_chunks, "steps_per_rank": steps_seen, "transfers": len(mb2.log)}
    ctx.close()
    return out
