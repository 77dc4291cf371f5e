#!/usr/bin/env python3
"""NOTE (round 4): the MP3 cycle-counter build (SYMACCEL_TUNE_MP3_CLOCK) was removed from csrc/mp3.hip with the other measurement-only
branches; check out a round-3 tree to reproduce profiles/r03*_mp3_clock*.  The AAC form lives on in csrc/experiments/aac_wave_walk.h.

Development tool: the shader clock and the cycles per round inside mp3_synth_kernel / aac_synth_kernel, from a library built with
SYMACCEL_TUNE_MP3_CLOCK=1 (or SYMACCEL_TUNE_AAC_SINK=1 SYMACCEL_TUNE_AAC_CLOCK=1; argument: mp3 | aac) (each half-wave's walk leaves its cycle / 100 MHz tick counts in its first PCM granule).
  SYMACCEL_LIB=build_ab/mp3_clock.so python tools/kernel_clock_probe.py"""
import sys
from pathlib import Path

import numpy as np
import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import bench  # noqa: E402
import symphonia_amd as sa  # noqa: E402


def main():
    ctx = sa.Context(0)
    ctx.use_torch_stream()
    name = sys.argv[1] if len(sys.argv) > 1 else "mp3"
    unit = {"mp3": 576, "aac": 1024}[name]
    step, *_rest, pcm = bench.make_workload(name, torch, ctx, 0)
    for _ in range(600):  # sustained: the board needs ~25 ms of load to leave its idle state, the clock then follows the load
        step()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    step()
    e1.record()
    torch.cuda.synchronize()
    words = pcm.view(torch.int32).cpu().numpy().view(np.uint32).reshape(-1, unit)
    tagged = words[words[:, 0] == 0x51A7C10C]
    cyc, ticks, rounds, start = (tagged[:, i].astype(np.float64) for i in (1, 2, 3, 4))
    us = ticks / 100.0
    print("kernel %.1f us (events); %d walks stamped" % (e0.elapsed_time(e1) * 1e3, len(tagged)))
    print("per walk: %.0f rounds, %.1f us (min %.1f max %.1f), %.0f shader cycles -> clock %.3f GHz" % (
        rounds.mean(), us.mean(), us.min(), us.max(), cyc.mean(), (cyc / (ticks * 10.0)).mean()))
    print("cycles per round: %.0f mean (min %.0f, max %.0f); us per round %.2f" % (
        (cyc / rounds).mean(), (cyc / rounds).min(), (cyc / rounds).max(), (us / rounds).mean()))
    if name == "aac":
        wait = tagged[:, 7].astype(np.float64)
        print("cycles in the wait for the prefetched frame: %.1f %% of the walk (%.0f per round)" % (100 * (wait / cyc).mean(), (wait / rounds).mean()))
    span = (start.max() - start.min()) / 100.0
    print("walk starts spread over %.1f us" % span)
    # HW_ID (gfx9): wave_id [3:0], simd_id [5:4], pipe [7:6], cu_id [11:8], sh [12], se_id [15:13]; XCC_ID [3:0]
    hw, xcc = tagged[:, 5], tagged[:, 6] & 0xF
    simd_key = ((xcc.astype(np.int64) << 16) | (hw & 0xFF30)).astype(np.int64)  # xcc, se, sh, cu, simd
    order = np.argsort(simd_key, kind="stable")
    keys, counts = np.unique(simd_key, return_counts=True)
    print("%d SIMDs carried walks; wavefronts per SIMD: %s" % (len(keys), dict(zip(*np.unique(counts, return_counts=True)))))
    # each wavefront stamps twice (two half-waves): per SIMD, the distinct wavefront durations sorted
    per = {}
    for k, u in zip(simd_key, us):
        per.setdefault(int(k), []).append(u)
    ranks = {}
    for k, v in per.items():
        v = sorted(v)[::2] if name == "mp3" else sorted(v)  # (mp3: the two halves of a wavefront stamp the same walk)
        for i, u in enumerate(v):
            ranks.setdefault((len(v), i), []).append(u)
    for (n, i), v in sorted(ranks.items()):
        print("  SIMDs with %d wavefronts: the %d. to finish took %.1f us on average (min %.1f, max %.1f; %d SIMDs)" % (
            n, i + 1, np.mean(v), np.min(v), np.max(v), len(v)))
    for x in range(8):
        m = xcc == x
        if m.any():
            print("  XCC %d: %.1f us mean, clock %.3f GHz" % (x, us[m].mean(), (cyc[m] / (ticks[m] * 10.0)).mean()))


if __name__ == "__main__":
    main()
