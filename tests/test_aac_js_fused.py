"""Dsp::synth with the joint-stereo decoding done as the lines are loaded (symaccel_aac_synth_js_*): bit for bit what joint stereo
(aac/cpe.rs:110-157) followed by Dsp::synth (aac/dsp.rs:57-158) gives -- oracle chain -- for channel pairs in any chain order,
chains outside every pair, long and EIGHT_SHORT frames (the stereo map follows the frame's window sequence), every `max_sfb`,
segment lengths that put pair partners into different workgroups.  CPU emulation here, the MI355X under `-m gpu`."""
import numpy as np
import pytest

import oracle
import test_aac_tools as T
from emu_lib import emu_ctx  # noqa: F401
from helpers import aac_sequence_chain, aac_spectra, bit_equal
from symphonia_amd import AacSpectralTools, aac_side


def case(rng, n_pairs, extra, frames, p_switch=0.3):
    chains = 2 * n_pairs + extra
    coeffs = aac_spectra(rng, (chains, frames), band_limit=int(rng.choice([672, 1024])))
    order = rng.permutation(chains)
    pairs = np.array([[order[2 * p], order[2 * p + 1]] for p in range(n_pairs)], np.int32).reshape(n_pairs, 2)
    side = np.zeros((chains, frames), np.uint8)
    seqs = {}
    for c in range(chains):
        seqs[c] = aac_sequence_chain(rng, frames, p_switch)
    for l, r in pairs:  # the channels of a pair share their window sequence (common_window: cpe.rs:58-66)
        seqs[int(r)] = seqs[int(l)]
    for c in range(chains):
        seq, shape, prev = seqs[c]
        side[c] = aac_side(seq, shape, prev)
    desc = np.zeros((n_pairs, frames), T.oracle_dtype_js())
    for p, (l, r) in enumerate(pairs):
        for f in range(frames):
            desc[p, f] = T.js_frame(rng, short=bool(seqs[int(l)][0][f] == 2))
    delay = rng.standard_normal((chains, 1024)).astype(np.float32)
    return coeffs, side, delay, pairs, desc


def want_of(coeffs, side, delay, pairs, desc):
    decoded = T.js_reference(coeffs, pairs, desc) if len(pairs) else coeffs
    return oracle.aac_synth(decoded, side, delay)


def run(ctx, to_dev, to_host, seed, n_pairs, extra, frames, seg, pp):
    rng = np.random.default_rng(seed)
    coeffs, side, delay, pairs, desc = case(rng, n_pairs, extra, frames)
    want_pcm, want_delay = want_of(coeffs, side, delay, pairs, desc)
    tools = AacSpectralTools(ctx, T.SWB_LONG, T.SWB_SHORT)
    d_coeffs, d_side = to_dev(coeffs), to_dev(side)
    d_pairs = to_dev(pairs) if n_pairs else None
    d_desc = to_dev(desc.view(np.uint8).reshape(n_pairs, frames, 644)) if n_pairs else None
    d_delay = to_dev(delay.copy())
    pcm = to_dev(np.zeros_like(coeffs))
    ctx.set_segment(seg)
    if pp:
        d_out = to_dev(np.zeros_like(delay))
        tools.synth_joint_stereo(d_coeffs, d_side, d_delay, d_pairs, d_desc, pcm, delay_out=d_out)
        got_delay = to_host(d_out)
    else:
        tools.synth_joint_stereo(d_coeffs, d_side, d_delay, d_pairs, d_desc, pcm)
        got_delay = to_host(d_delay)
    ctx.set_segment(0)
    assert bit_equal(to_host(d_coeffs), coeffs), "the coded spectra must not be touched"
    assert bit_equal(to_host(pcm), want_pcm), (n_pairs, extra, frames, seg)
    assert bit_equal(got_delay, want_delay)


CASES = [(1, 0, 9, 0, True), (2, 1, 13, 4, False), (3, 2, 6, 8, True), (0, 3, 5, 0, True), (4, 0, 21, 12, True)]


@pytest.mark.parametrize("n_pairs,extra,frames,seg,pp", CASES)
def test_emu_aac_synth_with_joint_stereo_on_load(emu_ctx, n_pairs, extra, frames, seg, pp):
    run(emu_ctx, lambda a: a, lambda a: a, 100 + 7 * n_pairs + frames, n_pairs, extra, frames, seg, pp)


def test_emu_js_fused_argument_checks(emu_ctx):
    from symphonia_amd import SymaccelError
    rng = np.random.default_rng(5)
    coeffs, side, delay, pairs, desc = case(rng, 1, 0, 4)
    tools = AacSpectralTools(emu_ctx, T.SWB_LONG, T.SWB_SHORT)
    pcm = np.zeros_like(coeffs)
    three = np.zeros((2, 2), np.int32)  # more pairs than the chains can hold
    with pytest.raises(SymaccelError):
        tools.synth_joint_stereo(coeffs, side, delay.copy(), three, np.zeros((2, 4, 644), np.uint8), pcm)
    bad = AacSpectralTools(emu_ctx, [0, 4, 7, 1024], T.SWB_SHORT)  # offsets that are not multiples of four
    with pytest.raises(SymaccelError):
        bad.synth_joint_stereo(coeffs, side, delay.copy(), pairs, desc.view(np.uint8).reshape(1, 4, 644), pcm)


@pytest.fixture(scope="module")
def gpu():
    torch = pytest.importorskip("torch")
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from symphonia_amd import Context
    ctx = Context(0)
    ctx.use_torch_stream()
    return ctx, (lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()), (lambda t: t.cpu().numpy())


@pytest.mark.gpu
@pytest.mark.parametrize("n_pairs,extra,frames,seg,pp", CASES + [(16, 3, 70, 0, True), (5, 1, 300, 0, False)])
def test_gpu_aac_synth_with_joint_stereo_on_load(gpu, n_pairs, extra, frames, seg, pp):
    run(*gpu, 900 + 7 * n_pairs + frames, n_pairs, extra, frames, seg, pp)
