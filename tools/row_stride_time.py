#!/usr/bin/env python3
"""Round 6: the lane-per-block kernels (FLAC restore, ALAC predict) by ROW PITCH through symaccel_*_strided_device: blocks of `blocksize` samples in rows `blocksize + pad` words apart.
What symaccel_row_stride() returns comes from this sweep.  Sampled rows are checked against the oracle, the padding for staying untouched.

    python tools/row_stride_time.py flac|alac [blocksize[,blocksize..]] [pad[,pad..]]"""
import json
import sys
from pathlib import Path

import numpy as np
import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import symphonia_amd as sa  # noqa: E402

SENTINEL = 0x5A5A5A5A


def main():
    import oracle
    codec = sys.argv[1] if len(sys.argv) > 1 else "flac"
    sizes = [int(v) for v in sys.argv[2].split(",")] if len(sys.argv) > 2 else [4096]
    pads = [int(v) for v in sys.argv[3].split(",")] if len(sys.argv) > 3 else [0, 16, 32, 64, 96, 128, 192, 256, 384, 512, 1024]
    ctx = sa.Context()
    ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    for bs in sizes:
        nb = (1 << 31) // (bs * 4) // 64 * 64  # 2 GiB of samples
        rng = np.random.default_rng(9)
        if codec == "flac":
            kind, od = np.full(nb, 2, np.uint8), np.full(nb, 12)
            desc_np = sa.flac_desc(kind, od, np.full(nb, 11), np.zeros(nb))
            co_np = np.zeros((nb, 32), np.int32)
            co_np[:] = (1500 * (0.6 ** np.arange(32)) * np.where(np.arange(32) % 2, -1, 1)).astype(np.int32)
            amp = 1 << 6
        else:
            od = np.full(nb, 8)
            desc_np = sa.alac_desc(np.zeros(nb), od, np.full(nb, 9), np.full(nb, 16))
            co_np = rng.integers(-200, 200, (nb, 32)).astype(np.int32)
            amp = 1 << 9
        desc = torch.from_numpy(desc_np.view(np.uint8).reshape(nb, 4)).cuda()
        co = torch.from_numpy(co_np).cuda()
        rows = [0, 1, 63, 64, nb // 2, nb - 1]
        for pad in pads:
            stride = bs + pad
            g = torch.Generator(device="cuda").manual_seed(3)
            buf = torch.full((nb, stride), SENTINEL, device="cuda", dtype=torch.int32)
            buf[:, :bs] = torch.randint(-amp, amp, (nb, bs), generator=g, device="cuda", dtype=torch.int32)
            before = buf[rows, :bs].cpu().numpy()
            if codec == "flac":
                run = lambda: sa.FlacPredictor(ctx).restore_strided(buf, desc, co, bs)
                want = lambda: oracle.flac_restore(before, oracle.flac_desc(kind[rows], od[rows], np.full(len(rows), 11), np.zeros(len(rows))), co_np[rows])
            else:
                run = lambda: sa.AlacPredictor(ctx).predict_strided(buf, desc, co, bs)
                want = lambda: oracle.alac_predict(before, desc_np[rows], co_np[rows])
            run()
            torch.cuda.synchronize()
            bad = int((buf[rows, :bs].cpu().numpy() != want()).sum())
            for _ in range(3):
                run()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(8):
                run()
            e1.record()
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / 8
            pad_ok = bool((buf[:, bs:] == SENTINEL).all()) if pad else True
            print(json.dumps({"codec": codec, "blocksize": bs, "pad_words": pad, "row_pitch_bytes": stride * 4, "blocks": nb, "ms_per_launch": round(ms, 4),
                              "frac_of_8TBps": round(nb * bs * 8 / (ms * 1e-3) / 8e12, 4), "mismatches_vs_oracle": bad, "padding_untouched": pad_ok}), flush=True)
            del buf


if __name__ == "__main__":
    main()
