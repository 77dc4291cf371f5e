#!/usr/bin/env python3
"""Development tool: ms per step over time from a cold start (events every `stride` steps): the board's clock / power management
shows as a curve.  python tools/step_timeline.py <workload> [steps] [stride]"""
import sys
import time
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import bench  # noqa: E402
import symphonia_amd as sa  # noqa: E402


def main():
    name = sys.argv[1] if len(sys.argv) > 1 else "aac"
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 4000
    stride = int(sys.argv[3]) if len(sys.argv) > 3 else 5
    ctx = sa.Context(0)
    ctx.use_torch_stream()
    step = bench.make_workload(name, torch, ctx, 0)[0]
    torch.cuda.synchronize()
    time.sleep(1.0)  # idle clocks
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(steps // stride + 1)]
    ev[0].record()
    for i in range(steps):
        step()
        if (i + 1) % stride == 0:
            ev[(i + 1) // stride].record()
    torch.cuda.synchronize()
    ms = [ev[i].elapsed_time(ev[i + 1]) / stride for i in range(len(ev) - 1)]
    t = 0.0
    marks = [0, 1, 2, 3, 4, 6, 8, 12, 16, 24, 32, 48, 64, 96, 128, 192, 256, 384, 512, 768]
    print("%s: ms per step in windows of %d steps from a cold start" % (name, stride))
    for a, b in zip(marks, marks[1:]):
        if a >= len(ms):
            break
        seg = ms[a:min(b, len(ms))]
        print("  steps %5d..%5d (from %7.1f ms): %.4f ms/step" % (a * stride, min(b, len(ms)) * stride, sum(ms[:a]) * stride, sum(seg) / len(seg)))
    print("  whole run: %.4f ms/step over %.0f ms" % (sum(ms) / len(ms), sum(ms) * stride))


if __name__ == "__main__":
    main()
