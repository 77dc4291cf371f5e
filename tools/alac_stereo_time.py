#!/usr/bin/env python3
"""Round 5: launch time of the ALAC predictor with decorrelate_mid_side fused into the write-back (symaccel_alac_predict_stereo_device),
the bench's ALAC batch as 131 072 stereo pairs, half of them mixed (weight 1, shift 2); checked against the oracle on sampled pairs."""
import json
import sys
from pathlib import Path

import numpy as np
import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import symphonia_amd as sa  # noqa: E402


def main():
    ctx = sa.Context()
    ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    g = torch.Generator(device="cuda").manual_seed(7)
    nb, bs = 262144, 4096
    buf = torch.randint(-(1 << 9), 1 << 9, (nb, bs), generator=g, device="cuda", dtype=torch.int32)
    desc_np = sa.alac_desc(np.zeros(nb), np.full(nb, 8), np.full(nb, 9), np.full(nb, 16))
    desc = torch.from_numpy(desc_np.view(np.uint8).reshape(nb, 4)).cuda()
    co = torch.randint(-200, 200, (nb, 32), generator=g, device="cuda", dtype=torch.int32)
    w_np = (np.arange(nb // 2) % 2).astype(np.int32)
    weight = torch.from_numpy(w_np).cuda()
    shift = torch.full((nb // 2,), 2, dtype=torch.uint8, device="cuda")
    ap = sa.AlacPredictor(ctx)
    rows = [0, 1, 2, 3, 126, 127, nb - 2, nb - 1]
    before = buf[rows].cpu().numpy()
    ap.predict_stereo(buf, desc, co, weight, shift)
    torch.cuda.synchronize()
    got = buf[rows].cpu().numpy()
    import oracle
    want = oracle.alac_predict(before, desc_np[rows], co[rows].cpu().numpy())
    for i in range(0, len(rows), 2):
        p = rows[i] // 2
        if w_np[p]:
            a, b = oracle.alac_decorrelate_mid_side(want[i].copy(), want[i + 1].copy(), int(w_np[p]), 2)
            want[i], want[i + 1] = a, b
    bad = int((got != want).sum())
    for _ in range(3):
        ap.predict_stereo(buf, desc, co, weight, shift)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(8):
        ap.predict_stereo(buf, desc, co, weight, shift)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 8
    print(json.dumps({"workload": "ALAC order 8, 131072 stereo pairs of 4096 samples, every second pair mixed", "ms_per_launch": ms,
                      "frac_of_8TBps": nb * bs * 8 / (ms * 1e-3) / 8e12, "mismatches_vs_oracle": bad, "samples_compared": int(got.size)}))


if __name__ == "__main__":
    main()
