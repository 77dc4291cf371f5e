#!/usr/bin/env python3
"""Round 6: the ALAC predictor on wavefronts of MIXED orders (what a batch of streams from an encoder that picks the order per frame looks like: ffmpeg writes 4 .. 6) against
the uniform order 8 of Apple's encoder; 262 144 blocks of 4096 samples, 16 bit, sampled blocks checked against the oracle."""
import json
import sys
from pathlib import Path

import numpy as np
import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import symphonia_amd as sa  # noqa: E402


def main():
    import oracle
    ctx = sa.Context()
    ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    nb, bs = 262144, 4096
    rng = np.random.default_rng(5)
    for name, orders in (("uniform 8", np.full(nb, 8)), ("4..6 mixed", rng.integers(4, 7, nb)), ("uniform 6", np.full(nb, 6)), ("uniform 4", np.full(nb, 4)),
                         ("1..8 mixed", rng.integers(1, 9, nb)), ("1..4 mixed", rng.integers(1, 5, nb))):
        g = torch.Generator(device="cuda").manual_seed(3)
        buf = torch.randint(-(1 << 9), 1 << 9, (nb, bs), generator=g, device="cuda", dtype=torch.int32)
        desc_np = sa.alac_desc(np.zeros(nb), orders, np.full(nb, 9), np.full(nb, 16))
        desc = torch.from_numpy(desc_np.view(np.uint8).reshape(nb, 4)).cuda()
        co = torch.randint(-200, 200, (nb, 32), generator=g, device="cuda", dtype=torch.int32)
        ap = sa.AlacPredictor(ctx)
        rows = [0, 1, 63, 64, nb // 2, nb - 1]
        bad = 0
        for _ in range(2):
            before = buf[rows].cpu().numpy()
            ap.predict(buf, desc, co)
            torch.cuda.synchronize()
            bad += int((buf[rows].cpu().numpy() != oracle.alac_predict(before, desc_np[rows], co[rows].cpu().numpy())).sum())
        for _ in range(8):
            ap.predict(buf, desc, co)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(16):
            ap.predict(buf, desc, co)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 16
        print(json.dumps({"orders": name, "ms_per_launch": round(ms, 4), "frac_of_8TBps": round(nb * bs * 8 / (ms * 1e-3) / 8e12, 4), "mismatches_vs_oracle": bad}), flush=True)
        del buf


if __name__ == "__main__":
    main()
