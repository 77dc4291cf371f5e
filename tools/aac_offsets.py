#!/usr/bin/env python3
"""Round 5: does config 2's launch time depend on where its two 512 MiB buffers sit (relative to each other, to a 2 MiB page)?
One process, one pool, the spectra at pool + oi and the PCM at pool + 768 MiB + oo; K launches per point, the list walked three times."""
import json
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from symphonia_amd import backend  # noqa: E402


def main():
    ctx = backend.Context()
    ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    d = ctx.lib.dll
    nch, nfr = 128, 1024
    nbytes = nch * nfr * 4096
    pool = torch.empty((1536 << 20) + (64 << 20), dtype=torch.uint8, device="cuda")
    fl = pool.view(torch.float32)
    fl[: (nbytes + (32 << 20)) // 4].normal_()
    side = torch.zeros(nch, nfr, dtype=torch.uint8, device="cuda")
    from oracle import aac_side  # (only the side byte's encoding: window sequence / shapes)
    side.fill_(int(aac_side(0, 1, 1)))
    d0 = torch.zeros(nch, 1024, dtype=torch.float32, device="cuda")
    d1 = torch.zeros_like(d0)
    base = pool.data_ptr()

    def run(oi, oo, steps=20):
        pin, pout = base + oi, base + (768 << 20) + oo
        def step():
            ctx._call(d.symaccel_aac_synth_pp_device, pin, side.data_ptr(), d0.data_ptr(), d1.data_ptr(), pout, nch, nfr)
        for _ in range(5):
            step()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(steps):
            step()
        e1.record()
        torch.cuda.synchronize()
        return round(e0.elapsed_time(e1) / steps * 1e3, 1)  # us

    for _ in range(300):
        run(0, 0, 1)
    few = "--few" in sys.argv
    full = [(0, 0), (0, 4096), (0, 8192), (0, 16384), (0, 32768), (0, 65536), (0, 1 << 20), (0, (1 << 20) + 4096), (4096, 0), (4096, 4096),
            (0, 2048), (0, 12288), (0, 20480), (0, (2 << 20) + 4096), (8192, 0), (0, 256), (0, 1024)]
    pts = [(0, 0), (0, 4096), (4096, 4096), (8192, 8192)] if few else full
    import os
    out = {"lib": os.environ.get("SYMACCEL_LIB", "product"), "pool_mod_2MiB": base % (2 << 20), "unit": "us per launch", "points": {}}
    for _ in range(3):
        for oi, oo in pts:
            out["points"].setdefault("%d,%d" % (oi, oo), []).append(run(oi, oo))
    print(json.dumps(out))


if __name__ == "__main__":
    main()
