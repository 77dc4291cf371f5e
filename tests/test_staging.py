"""The staged host entry points (pinned / chunked / double-buffered: csrc/stage.cpp) give exactly what the one-shot host
entry points give, for every chunking -- chunk boundaries carry the state like consecutive decode() calls.  CPU: emulation
(the copies are synchronous there: this checks the chunk arithmetic and the state hand-over); GPU: the real three-stream
pipeline, on pageable and on pinned memory."""
import numpy as np
import pytest

import oracle
from emu_lib import emu_ctx  # noqa: F401
from helpers import aac_sequence_chain, aac_spectra, bit_equal
from symphonia_amd import AacDsp, FlacPredictor, Mp3Synthesis, aac_side, flac_desc, mp3_side


def aac_case(nch, nfr, seed):
    rng = np.random.default_rng(seed)
    coeffs = aac_spectra(rng, (nch, nfr))
    side = np.empty((nch, nfr), np.uint8)
    for c in range(nch):
        s, sh, pv = aac_sequence_chain(rng, nfr, p_switch=0.3)
        side[c] = aac_side(s, sh, pv)
    return coeffs, side, rng.standard_normal((nch, 1024)).astype(np.float32)


def run_aac(ctx, chunks):
    coeffs, side, delay = aac_case(3, 23, 1)
    want_pcm, want_delay = oracle.aac_synth(coeffs, side, delay)
    for ch in chunks:
        pcm, nd = AacDsp(ctx).synth(coeffs, side, delay, chunk_frames=ch)
        assert bit_equal(pcm, want_pcm) and bit_equal(nd, want_delay), ch


def run_mp3(ctx, chunks):
    from test_emu_codecs import mp3_case
    rng = np.random.default_rng(2)
    nch, ngr = 3, 19
    xr, bt, mx, rz = mp3_case(rng, nch, ngr)
    ov = rng.standard_normal((nch, 576)).astype(np.float32)
    vv, vf = np.zeros((nch, 1024), np.float32), np.array([0, 3, 9], np.int32)
    want = oracle.mp3_synth(xr, oracle.mp3_side(bt, mx, rz), 1, ov, vv, vf)
    for ch in chunks:
        got = Mp3Synthesis(ctx, 1).synth(xr, mp3_side(bt, mx, rz), ov, vv, vf, chunk_granules=ch)
        for a, b in zip(got, want):
            assert bit_equal(a, np.asarray(b)), ch


def run_flac(ctx, chunks):
    rng = np.random.default_rng(3)
    nb, bs = 37, 96
    buf = rng.integers(-2000, 2000, (nb, bs)).astype(np.int32)
    kind, order = rng.integers(0, 3, nb), rng.integers(1, 5, nb)
    order[kind == 2] = rng.integers(1, 33, int((kind == 2).sum()))
    shift = rng.integers(0, 15, nb)
    co = (rng.integers(-300, 300, (nb, 32)) * 0.6 ** np.arange(32)).astype(np.int32)
    desc = flac_desc(kind, order, shift, np.zeros(nb))
    want = oracle.flac_restore(buf, oracle.flac_desc(kind, order, shift, 0 * shift), co)
    for ch in chunks:
        assert np.array_equal(FlacPredictor(ctx).restore(buf, desc, co, chunk_blocks=ch), want), ch


def test_emu_staged_paths(emu_ctx):
    run_aac(emu_ctx, [1, 2, 5, 23, 64, 0])
    run_mp3(emu_ctx, [2, 3, 7, 19, 0])
    run_flac(emu_ctx, [1, 8, 36, 37, 100, 0])


@pytest.fixture(scope="module")
def gpu_ctx():
    torch = pytest.importorskip("torch")
    if not torch.cuda.is_available():
        pytest.fail("no GPU visible: the gpu-marked tests must run on an MI355X (there is no CPU path)")
    from symphonia_amd import Context
    c = Context(0)
    yield c
    c.close()


@pytest.mark.gpu
def test_gpu_staged_paths(gpu_ctx):
    run_aac(gpu_ctx, [1, 4, 23, 0])
    run_mp3(gpu_ctx, [2, 7, 0])
    run_flac(gpu_ctx, [1, 8, 0])


@pytest.mark.gpu
def test_gpu_staged_aac_on_pinned_memory_matches_the_one_shot_path(gpu_ctx):
    """A batch big enough for several chunks (96 MiB of spectra), in page-locked buffers: the overlapped pipeline must give
    the one-shot result bit for bit (sampled chains are checked against the oracle)."""
    from symphonia_amd import PinnedBuffer
    nch, nfr = 24, 1024
    rng = np.random.default_rng(4)
    pc, pp = PinnedBuffer((nch, nfr, 1024), np.float32), PinnedBuffer((nch, nfr, 1024), np.float32)
    pc.array[:] = rng.standard_normal((nch, nfr, 1024)).astype(np.float32)
    side = np.full((nch, nfr), aac_side(0, 1, 1), np.uint8)
    delay = np.zeros((nch, 1024), np.float32)
    d = gpu_ctx.lib.dll
    nd = delay.copy()
    gpu_ctx._call(d.symaccel_aac_synth_pipelined, pc.array.ctypes.data, side.ctypes.data, nd.ctypes.data, pp.array.ctypes.data, nch, nfr, 128)
    one, nd1 = AacDsp(gpu_ctx).synth(pc.array, side, delay, chunk_frames=nfr)
    assert bit_equal(pp.array, one) and bit_equal(nd, nd1)
    pick = [0, 11, 23]
    want, _ = oracle.aac_synth(pc.array[pick], side[pick], delay[pick])
    assert bit_equal(pp.array[pick], want)
    pc.free()
    pp.free()
