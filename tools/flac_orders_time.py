#!/usr/bin/env python3
"""Round 6: the FLAC restore kernel by predictor order as real streams have them (flac -5: LPC order <= 8, -8: <= 12, 16-bit, 12-bit coefficients) against config 5's order 32;
524 288 blocks of 4096 samples, mixed orders per wavefront, sampled blocks checked against the oracle."""
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
    nb = 524288
    bs = int(sys.argv[1]) if len(sys.argv) > 1 else 4096  # (row stride in samples: 4096 = 16 KiB, a power of two)
    rng = np.random.default_rng(9)
    for name, orders, bits in (("uniform 32, 24 bit", np.full(nb, 32), 24), ("uniform 12, 16 bit", np.full(nb, 12), 16), ("6..12 mixed, 16 bit", rng.integers(6, 13, nb), 16),
                               ("uniform 8, 16 bit", np.full(nb, 8), 16), ("1..8 mixed, 16 bit", rng.integers(1, 9, nb), 16), ("fixed 0..4, 16 bit", None, 16)):
        g = torch.Generator(device="cuda").manual_seed(3)
        buf = torch.randint(-(1 << 6), 1 << 6, (nb, bs), generator=g, device="cuda", dtype=torch.int32)
        if orders is None:
            kind, od = np.full(nb, 1, np.uint8), rng.integers(0, 5, nb)
        else:
            kind, od = np.full(nb, 2, np.uint8), orders
        desc_np = sa.flac_desc(kind, od, np.full(nb, 11), np.zeros(nb))
        desc = torch.from_numpy(desc_np.view(np.uint8).reshape(nb, 4)).cuda()
        # small, decaying coefficients (|sum| well inside the FP64 path's bound) so that repeated in-place passes stay in range
        co_np = np.zeros((nb, 32), np.int32)
        base = (1500 * (0.6 ** np.arange(32)) * np.where(np.arange(32) % 2, -1, 1)).astype(np.int32)
        co_np[:] = base
        co = torch.from_numpy(co_np).cuda()
        fp = sa.FlacPredictor(ctx)
        rows = [0, 1, 63, 64, nb // 2, nb - 1]
        before = buf[rows].cpu().numpy()
        fp.restore(buf, desc, co)
        torch.cuda.synchronize()
        bad = int((buf[rows].cpu().numpy() != oracle.flac_restore(before, oracle.flac_desc(kind[rows], od[rows], np.full(len(rows), 11), np.zeros(len(rows))), co_np[rows])).sum())
        for _ in range(3):
            fp.restore(buf, desc, co)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(8):
            fp.restore(buf, desc, co)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 8
        print(json.dumps({"blocksize": bs, "case": name, "ms_per_launch": round(ms, 4), "frac_of_8TBps": round(nb * bs * 8 / (ms * 1e-3) / 8e12, 4), "mismatches_vs_oracle": bad}), flush=True)
        del buf


if __name__ == "__main__":
    main()
