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


# ---------------------------------------------------------------- the whole tail host to host: joint stereo + TNS + synthesis

def decode_case(seed, n_pairs, extra, frames, p_tns):
    """case() + TNS filters on a share of the channel frames (of paired and unpaired chains): expectation = the reference's order --
    joint stereo on the pair (cpe.rs:110-157), then per channel TNS (ics/tns.rs:149-199) and Dsp::synth (ics/mod.rs:449-468)"""
    from symphonia_amd import AAC_TNS_DTYPE
    rng = np.random.default_rng(seed)
    coeffs, side, delay, pairs, desc = case(rng, n_pairs, extra, frames)
    chains = coeffs.shape[0]
    filt = []
    for c in range(chains):
        for f in range(frames):
            if rng.random() >= p_tns:
                continue
            short = (side[c, f] & 3) == 2
            for w in (rng.choice(8, int(rng.integers(1, 4)), replace=False) if short else [0]):
                lo, hi = sorted(rng.choice(T.SWB_SHORT if short else T.SWB_LONG, 2, replace=False))
                order = int(rng.integers(1, 8 if short else 13))
                filt.append((c * frames + f, int(w) * 128 + lo, int(w) * 128 + hi, order, int(rng.integers(0, 2)), 0,
                             T.tns_lpc(rng, order, coef_res=bool(rng.integers(0, 2)))))
    filt = np.array(filt, AAC_TNS_DTYPE) if filt else np.zeros(0, AAC_TNS_DTYPE)
    filt = filt[rng.permutation(len(filt))]
    decoded = T.js_reference(coeffs, pairs, desc) if len(pairs) else coeffs.copy()
    decoded = T.tns_reference(decoded.reshape(-1, 1024), filt, chains * frames).reshape(coeffs.shape)
    want_pcm, want_delay = oracle.aac_synth(decoded, side, delay)
    return coeffs, side, delay, pairs, desc, filt, want_pcm, want_delay


def run_decode(ctx, seed, n_pairs, extra, frames, p_tns, chunks):
    coeffs, side, delay, pairs, desc, filt, want_pcm, want_delay = decode_case(seed, n_pairs, extra, frames, p_tns)
    tools = AacSpectralTools(ctx, T.SWB_LONG, T.SWB_SHORT)
    for chunk in chunks:
        d, pcm = delay.copy(), np.zeros_like(coeffs)
        before = coeffs.copy()
        tools.decode(coeffs, side, d, pairs if n_pairs else None, np.ascontiguousarray(desc) if n_pairs else None, filt if len(filt) else None,
                     pcm, chunk_frames=chunk)
        assert bit_equal(coeffs, before), "the caller's spectra must not be touched"
        assert bit_equal(pcm, want_pcm), (n_pairs, extra, frames, p_tns, chunk)
        assert bit_equal(d, want_delay), chunk


DECODE_CASES = [(1, 0, 9, 0.3), (2, 1, 7, 0.5), (0, 2, 5, 0.4), (3, 0, 6, 0.0), (2, 2, 11, 1.0)]


@pytest.mark.parametrize("n_pairs,extra,frames,p_tns", DECODE_CASES)
def test_emu_aac_decode_pipelined(emu_ctx, n_pairs, extra, frames, p_tns):
    """symaccel_aac_decode_pipelined: coded spectra + stereo descriptors + TNS filters in host memory -> PCM in host memory, in chunks
    of every size (the filters are sorted into the chunks; pair frames with TNS take the list pass, all others the fused walk)"""
    run_decode(emu_ctx, 300 + 11 * n_pairs + frames, n_pairs, extra, frames, p_tns, [0, 2, 3, frames])


def test_emu_aac_decode_argument_checks(emu_ctx):
    from symphonia_amd import SymaccelError
    coeffs, side, delay, pairs, desc, filt, _, _ = decode_case(1, 1, 0, 3, 0.5)
    tools = AacSpectralTools(emu_ctx, T.SWB_LONG, T.SWB_SHORT)
    pcm = np.zeros_like(coeffs)
    with pytest.raises(SymaccelError):  # a chain in two pairs
        tools.decode(coeffs, side, delay.copy(), np.array([[0, 1], [1, 0]], np.int32), np.zeros((2, 3), T.oracle_dtype_js()), None, pcm)
    with pytest.raises(SymaccelError):  # a pair outside the batch
        tools.decode(coeffs, side, delay.copy(), np.array([[0, 2]], np.int32), np.ascontiguousarray(desc), None, pcm)
    far = filt.copy()
    far["frame"] = 10 ** 6  # filters outside the batch are skipped, like symaccel_aac_tns_device does
    want_pcm, _ = want_of(coeffs, side, delay, pairs, desc)
    tools.decode(coeffs, side, delay.copy(), pairs, np.ascontiguousarray(desc), far, pcm)
    assert bit_equal(pcm, want_pcm)


@pytest.mark.gpu
@pytest.mark.parametrize("n_pairs,extra,frames,p_tns", DECODE_CASES + [(16, 3, 70, 0.3), (5, 1, 300, 0.1)])
def test_gpu_aac_decode_pipelined(n_pairs, extra, frames, p_tns):
    from symphonia_amd import Context
    ctx = Context(0)
    run_decode(ctx, 700 + 11 * n_pairs + frames, n_pairs, extra, frames, p_tns, [0, 4, 64])
    ctx.close()
