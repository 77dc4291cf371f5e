#!/usr/bin/env python3
"""HBM access-pattern probe (GPU box): symaccel_probe_copy_device over 512 MiB in / 512 MiB out in copy, read-only and
write-only mode, plain and non-temporal, for the access shapes a per-chain walker can have:
  k consecutive 4 KiB frames per wavefront (k = 1 ... 64), and the same frames shared round-robin by the 4 wavefronts of a
  workgroup (16 KiB contiguous per workgroup and step).   python tools/probe_patterns.py > gpurun_out/probe_patterns.txt"""
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import symphonia_amd as sa  # noqa: E402

ctx = sa.Context(0)
ctx.use_torch_stream()
d = ctx.lib.dll
nbytes = 512 << 20
a = torch.empty(nbytes // 4, dtype=torch.float32, device="cuda").normal_()
b = torch.empty_like(a)


def run(k, flags, reps=10):
    for _ in range(reps):  # (untimed: sustained clocks)
        ctx._call(d.symaccel_probe_copy_device, a.data_ptr(), b.data_ptr(), nbytes, k, flags)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        ctx._call(d.symaccel_probe_copy_device, a.data_ptr(), b.data_ptr(), nbytes, k, flags)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


print("# 512 MiB read and / or 512 MiB written per launch; ms and TB/s of the bytes actually moved")
for rnd in range(2):
    print("# round", rnd)
    for mode, name, moved in ((0, "copy ", 2), (1, "read ", 1), (2, "write", 1)):
        for nt in (1, 0):
            for group in (0, 8, 16, 24):
                row = []
                for k in (1, 4, 16, 64):
                    ms = run(k, nt | (mode << 1) | group)
                    row.append("k=%-2d %.3f ms %.2f" % (k, ms, moved * nbytes / ms / 1e9))
                print("%s %s %s | %s" % (name, "nt   " if nt else "plain", {0: "per-wavefront   ", 8: "4 waves together ", 16: "8 waves together ", 24: "16 waves together"}[group], " | ".join(row)))
    ms = run(0, 0)
    print("copy  plain grid-stride float4: %.3f ms %.2f" % (ms, 2 * nbytes / ms / 1e9))
    ms = run(0, 1)
    print("copy  nt    grid-stride float4: %.3f ms %.2f" % (ms, 2 * nbytes / ms / 1e9))
ctx.close()
