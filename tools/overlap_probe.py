"""Development probe (GPU box): does a multi-kernel pipeline gain from running SLICES of the batch on separate HIP streams?

    python tools/overlap_probe.py --workload vorbisf --parts 1 2 4 [--steps 200]

For each P: P contexts (each with a stream of its own), each with the workload at scale 1 / P (bench.make_workload), their steps enqueued
round-robin with nothing in between; the line gives the time per WHOLE batch (P slices).  P = 1 is the pipeline as the product runs it.
If the latency-bound stage of one slice (floor render, TNS filters) hides beside the bandwidth-bound stage of another, P > 1 is faster.
"""
import argparse
import json
import sys
import time
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="vorbisf")
    ap.add_argument("--parts", type=int, nargs="+", default=[1, 2, 4])
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--reps", type=int, default=3)
    args = ap.parse_args()
    import torch
    import bench
    import symphonia_amd as sa
    for parts in args.parts:
        ctxs = [sa.Context(0) for _ in range(parts)]
        built = [bench.make_workload(args.workload, torch, c, 1234 + i, 1.0 / parts) for i, c in enumerate(ctxs)]
        steps = [b[0] for b in built]
        units = sum(b[1] for b in built)
        torch.cuda.synchronize()
        t_spin = time.perf_counter()
        while time.perf_counter() - t_spin < 0.1:
            for _ in range(4):
                for s in steps:
                    s()
            torch.cuda.synchronize()
        best = []
        for _ in range(args.reps):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(args.steps):
                for s in steps:
                    s()
            torch.cuda.synchronize()
            best.append((time.perf_counter() - t0) / args.steps * 1e3)
        print(json.dumps({"workload": args.workload, "parts": parts, "units": units, "ms_per_batch": [round(b, 4) for b in best]}), flush=True)
        del steps, built
        for c in ctxs:
            c.close()
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
