"""symaccel_probe_copy_device (the same-run copy ceiling of bench.py): it copies, in every mode, and rejects
what it documents.  CPU: emulation build; GPU: the real library."""
import numpy as np
import pytest

from emu_lib import emu_ctx  # noqa: F401
from symphonia_amd import SymaccelError, _ffi


def _modes():
    return [(0, 0), (0, 1), (1, 1), (3, 0), (64, 1), (2, 9), (5, 8), (2, 17), (1, 25), (3, 24), (2, 41), (1, 40), (7, 41)]


def test_emu_probe_copy(emu_ctx):
    d = emu_ctx.lib.dll
    rng = np.random.default_rng(3)
    src = rng.integers(0, 1 << 32, 5 * 1024, dtype=np.uint32)  # five 4 KiB frames
    for fpw, flags in _modes():
        dst = np.zeros_like(src)
        emu_ctx._call(d.symaccel_probe_copy_device, src.ctypes.data, dst.ctypes.data, src.nbytes, fpw, flags)
        emu_ctx.sync()
        assert np.array_equal(src, dst), (fpw, flags)
    dst = np.zeros_like(src)
    for args in ((src.ctypes.data, dst.ctypes.data, 4095, 0, 0),            # not whole frames
                 (src.ctypes.data + 4, dst.ctypes.data, 4096, 0, 0),        # unaligned
                 (src.ctypes.data, src.ctypes.data + 4096, 8192, 0, 0),     # overlapping
                 (src.ctypes.data, dst.ctypes.data, 4096, 0, 16),           # unknown flag
                 (0, dst.ctypes.data, 4096, 0, 0)):
        with pytest.raises(SymaccelError) as e:
            emu_ctx._call(d.symaccel_probe_copy_device, *args)
        assert e.value.status == _ffi.ERR_INVALID_ARG


@pytest.mark.gpu
def test_gpu_probe_copy():
    import torch
    from symphonia_amd import Context
    ctx = Context(0)
    ctx.use_torch_stream()
    d = ctx.lib.dll
    src = torch.randint(0, 1 << 31, (37 * 1024,), dtype=torch.int32, device="cuda")
    for fpw, flags in _modes():
        dst = torch.zeros_like(src)
        ctx._call(d.symaccel_probe_copy_device, src.data_ptr(), dst.data_ptr(), src.numel() * 4, fpw, flags)
        torch.cuda.synchronize()
        assert torch.equal(src, dst), (fpw, flags)
    ctx.close()
