#!/usr/bin/env python3
"""Throughput of symaccel_vorbis_synth_device for every block-size pair class: the wavefront kernel (256 / 2048) and the
generic LDS kernel (everything else).  Development tool, run on the GPU box: python tools/vorbis_pairs_probe.py"""
import sys
from pathlib import Path

import numpy as np
import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import symphonia_amd as sa  # noqa: E402

ctx = sa.Context(0)
ctx.use_torch_stream()


def run(e0, e1, p_ll=0.9, p_ss=0.7):
    nch = 64
    nb = max(64, (4096 * 2048) >> e1)  # about the same number of lines per chain for every pair
    rng = np.random.default_rng(1)
    flags = np.zeros((nch, nb), np.uint8)
    cur = np.ones(nch, bool)
    for b in range(nb):
        r = rng.random(nch)
        cur = np.where(cur, r < p_ll, r >= p_ss)
        flags[:, b] = cur
    v = sa.VorbisDsp(ctx, e0, e1)
    so, po = v.layout(flags, np.full(nch, -1))
    ss, ps = int(so[:, -1].max()), int(po[:, -1].max())
    spectra = torch.randn((nch, ss), device="cuda") * 0.1
    dfl = torch.from_numpy(flags).cuda()
    # ping-pong state like bench.py's Vorbis step (symaccel_vorbis_synth_pp_device: one launch per step, no state copy in front);
    # every step starts from the same state (prev = -1, empty overlap)
    prev = torch.full((nch,), -1, dtype=torch.int32, device="cuda")
    ov = torch.zeros((nch, (1 << e1) // 2), device="cuda")
    prev_out, ov_out = torch.empty_like(prev), torch.empty_like(ov)
    pcm = torch.zeros((nch, ps), device="cuda")

    def step():
        v.synth(spectra, dfl, prev, ov, ps, pcm, state_out=(prev_out, ov_out))
    import time
    t0 = time.perf_counter()  # sustained clocks: ~25 ms of load take the board out of its idle state (profiles/r03w_step_timeline.txt)
    while time.perf_counter() - t0 < 0.06:
        for _ in range(3):
            step()
        torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(10):
        step()
    b.record()
    torch.cuda.synchronize()
    t = a.elapsed_time(b) / 10
    byts = 4 * (so[:, -1].sum() + po[:, -1].sum())
    print("%4d / %-5d  %6d blocks/chain  long share %.2f  %.3f ms  %.2f TB/s (%.1f %% of 8 TB/s)" % (
        1 << e0, 1 << e1, nb, flags.mean(), t, byts / t / 1e9, byts / t / 1e9 / 8 * 100))


PAIRS = ((8, 11), (7, 10), (9, 12), (6, 9), (8, 10), (10, 13), (11, 11))
if len(sys.argv) > 1:  # python tools/vorbis_pairs_probe.py 7,10 9,12
    PAIRS = tuple(tuple(int(x) for x in a.split(",")) for a in sys.argv[1:])
import os
MIX = [tuple(float(x) for x in m.split(",")) for m in os.environ.get("PAIRS_MIX", "0.9,0.7").split(";")]  # PAIRS_MIX="0.9,0.7;1,0;0,1": (p_ll, p_ss) ...
for e0, e1 in PAIRS:
    for p_ll, p_ss in MIX:
        run(e0, e1, p_ll, p_ss)
