#!/usr/bin/env python3
"""Round 5: is the 10 % between the workgroup walk's copy rate and the plain copy's a power-of-two stride effect?

The walk's workgroups sit `4 * per_wave` frames apart (config 2: 1 MiB); this sweeps per_wave around 64 for the chain-major walk
(flags 9) and the window-major one (flags 41), several trials each in alternation, so that a bimodal pattern shows as such.
Prints one JSON object."""
import json
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from symphonia_amd import backend  # noqa: E402


def main():
    ctx = backend.Context()
    ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    nbytes = 512 << 20
    a = torch.empty(nbytes // 4 + (1 << 20), dtype=torch.float32, device="cuda").normal_()
    b = torch.empty_like(a)
    d = ctx.lib.dll

    def rate(fpw, flags, off=0, reps=10):
        pa, pb = a.data_ptr() + off, b.data_ptr() + off
        for _ in range(reps):
            ctx._call(d.symaccel_probe_copy_device, pa, pb, nbytes, fpw, flags)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            ctx._call(d.symaccel_probe_copy_device, pa, pb, nbytes, fpw, flags)
        e1.record()
        torch.cuda.synchronize()
        return round(2 * nbytes * reps / (e0.elapsed_time(e1) * 1e-3) / 1e9)

    for _ in range(30):
        rate(64, 9)
    out = {"base_a_mod_2MiB": a.data_ptr() % (2 << 20), "base_b_mod_2MiB": b.data_ptr() % (2 << 20), "unit": "GB/s", "trials": 4}
    per_waves = [32, 48, 56, 60, 63, 64, 65, 66, 68, 72, 80, 96, 128]
    res = {"walk": {}, "window": {}, "plain_nt": [], "one_frame_nt": []}
    for trial in range(4):
        res["plain_nt"].append(rate(0, 1))
        res["one_frame_nt"].append(rate(1, 1))
        for pw in per_waves:
            res["walk"].setdefault(str(pw), []).append(rate(pw, 9))
            res["window"].setdefault(str(pw), []).append(rate(pw, 41))
    out.update(res)
    # the walk at 64 with the two buffers shifted against each other / against the 2 MiB page
    out["walk_64_offsets"] = {str(off): [rate(64, 9, off) for _ in range(3)] for off in (0, 4096, 65536, 1 << 20, (1 << 20) + 65536)}
    out["window_64_offsets"] = {str(off): [rate(64, 41, off) for _ in range(3)] for off in (0, 4096, 65536, 1 << 20, (1 << 20) + 65536)}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
