#!/usr/bin/env python3
"""Round 6: launch time of the ALAC predictor by channel bit depth -- 16 / 20 bit (the narrow form: 24-bit multiplies), 24 bit (the wide form: orders <= 8 on
split 24-bit multiplies when the wavefront's blocks allow it, else full 32-bit multiplies) -- the bench's batch (262 144 blocks of 4096 samples, order 8),
sampled blocks checked against the oracle."""
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
    for bps, order in ((16, 8), (20, 8), (24, 8), (24, 4), (16, 16), (24, 16), (16, 31), (24, 31)):
        g = torch.Generator(device="cuda").manual_seed(bps)
        buf = torch.randint(-(1 << (bps - 7)), 1 << (bps - 7), (nb, bs), generator=g, device="cuda", dtype=torch.int32)
        desc_np = sa.alac_desc(np.zeros(nb), np.full(nb, order), np.full(nb, 9), np.full(nb, bps))
        desc = torch.from_numpy(desc_np.view(np.uint8).reshape(nb, 4)).cuda()
        co = torch.randint(-200, 200, (nb, 32), generator=g, device="cuda", dtype=torch.int32)
        ap = sa.AlacPredictor(ctx)
        rows = [0, 1, 63, 64, nb // 2, nb - 1]
        bad = 0
        for _ in range(3):  # (in place: later passes run on clipped full-range data)
            before = buf[rows].cpu().numpy()
            ap.predict(buf, desc, co)
            torch.cuda.synchronize()
            got = buf[rows].cpu().numpy()
            bad += int((got != oracle.alac_predict(before, desc_np[rows], co[rows].cpu().numpy())).sum())
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(8):
            ap.predict(buf, desc, co)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 8
        print(json.dumps({"bps": bps, "order": order, "ms_per_launch": round(ms, 4), "frac_of_8TBps": round(nb * bs * 8 / (ms * 1e-3) / 8e12, 4),
                          "mismatches_vs_oracle": bad}), flush=True)
        del buf


if __name__ == "__main__":
    main()
