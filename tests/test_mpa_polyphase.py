"""Layer I / Layer II polyphase synthesis (synthesis::synthesis with n_frames 12 / 36, synthesis.rs:158-336):
the kernel against the oracle's so_mp3_polyphase, packet by packet with state carried -- in CPU emulation (logic) and,
gpu-marked, on the MI355X.  The oracle's polyphase is pinned by the dct32 KAT (synthesis.rs:868-881) and the ISO
11172-3 closed form in tests/test_oracle_*.py."""
import numpy as np
import pytest

import oracle
from emu_lib import emu_ctx  # noqa: F401
from helpers import bit_equal


def case(seed, nch, npk, n_frames):
    rng = np.random.default_rng(seed)
    x = (rng.standard_normal((nch, npk, 32 * n_frames)) * 0.2).astype(np.float32)
    vv = rng.standard_normal((nch, 1024)).astype(np.float32)
    vf = rng.integers(0, 16, nch).astype(np.int32)
    return x, vv, vf


def reference(x, vv, vf, n_frames):
    nch, npk = x.shape[:2]
    out = np.empty_like(x)
    vv, vf = vv.copy(), vf.copy()
    for c in range(nch):
        v, f = vv[c], int(vf[c])
        for p in range(npk):
            out[c, p], v, f = oracle.mp3_polyphase(v, f, n_frames, x[c, p])
        vv[c], vf[c] = v, f
    return out, vv, vf


@pytest.mark.parametrize("n_frames,npk,seg", [(12, 7, 0), (36, 5, 0), (12, 9, 2), (36, 6, 1), (12, 1, 0), (36, 1, 0)])
def test_emu_mpa_polyphase(emu_ctx, n_frames, npk, seg):
    from symphonia_amd import MpaPolyphase
    x, vv, vf = case(n_frames + npk, 3, npk, n_frames)
    emu_ctx.set_segment(seg)
    got = MpaPolyphase(emu_ctx, n_frames).synth(x, vv, vf)
    emu_ctx.set_segment(0)
    want = reference(x, vv, vf, n_frames)
    assert bit_equal(got[0], want[0]), "pcm"
    assert bit_equal(got[1], want[1]), "v_vec"
    assert np.array_equal(got[2], want[2]), "v_front"


def test_unsupported_n_frames(emu_ctx):
    from symphonia_amd import MpaPolyphase
    with pytest.raises(ValueError):
        MpaPolyphase(emu_ctx, 18)


@pytest.mark.gpu
@pytest.mark.parametrize("n_frames,npk,seg", [(12, 40, 0), (36, 33, 0), (12, 64, 5), (36, 50, 3)])
def test_gpu_mpa_polyphase(n_frames, npk, seg):
    import torch
    from symphonia_amd import Context, MpaPolyphase
    if not torch.cuda.is_available():
        pytest.fail("no GPU visible")
    x, vv, vf = case(100 + n_frames + npk, 9, npk, n_frames)
    with Context(0) as ctx:
        ctx.use_torch_stream()
        ctx.set_segment(seg)
        d_vv, d_vf = torch.from_numpy(vv.copy()).cuda(), torch.from_numpy(vf.copy()).cuda()
        pcm = MpaPolyphase(ctx, n_frames).synth(torch.from_numpy(x).cuda(), d_vv, d_vf)
        torch.cuda.synchronize()
        got = (pcm.cpu().numpy(), d_vv.cpu().numpy(), d_vf.cpu().numpy())
    want = reference(x, vv, vf, n_frames)
    assert bit_equal(got[0], want[0]) and bit_equal(got[1], want[1]) and np.array_equal(got[2], want[2])
