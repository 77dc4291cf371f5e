"""The floor-1 curve as dB-table indices (one byte per line) and its fused use in the synthesis kernels.

floor.rs:785-825 (render_line) stores FLOOR1_INVERSE_DB_TABLE[y] and nothing else, so a curve is fully described by one byte
per line; lib.rs:289-291 then multiplies it with the residue.  symaccel_vorbis_floor1_y_device writes the byte plane,
symaccel_vorbis_synth_fy_* looks the table up and multiplies as it loads the residue.  Checked bit for bit against the oracle
chain floor-1 curve -> dot product -> synthesis, on the CPU emulation of the kernels here and on the MI355X (`-m gpu`)."""
import numpy as np
import pytest

import oracle
from emu_lib import emu_ctx  # noqa: F401
from helpers import bit_equal
from symphonia_amd import VorbisDsp


def db_table():
    """FLOOR1_INVERSE_DB_TABLE through the oracle: two posts at the same height render a flat curve of table[y]"""
    return np.array([oracle.vorbis_floor1([0, 16], np.array([v, v], np.uint32), 1, 16)[0] for v in range(256)], np.float32)


FLOOR_CASES = ((1024, 30, 2, 7, 0.3), (128, 9, 1, 7, 0.3), (1024, 65, 4, 7, 0.3), (128, 2, 3, 7, 0.3), (2048, 65, 1, 70, 0.02),
               (32, 5, 2, 130, 0.5), (4096, 40, 2, 3, 0.9))


def floor_case(rng, n, n_posts, mult, count, p_zero):
    xs = [0, n] + rng.permutation(np.arange(1, n))[:n_posts - 2].tolist()
    rr = [256, 128, 86, 64][mult - 1]
    ys = rng.integers(0, rr, size=(count, n_posts)).astype(np.uint32)
    ys[rng.random((count, n_posts)) < p_zero] = 0
    return xs, ys


def check_floor1_y(ctx, to_dev, to_host):
    rng = np.random.default_rng(606)
    v = VorbisDsp(ctx, 8, 11)
    db = db_table()
    for n, n_posts, mult, count, p_zero in FLOOR_CASES:
        xs, ys = floor_case(rng, n, n_posts, mult, count, p_zero)
        want = np.stack([oracle.vorbis_floor1(xs, y, mult, n) for y in ys])
        plane = to_dev(np.full((count, n), 0xA5, np.uint8))
        v.floor1(xs, mult, to_dev(ys), n, None, count, y_plane=plane)
        assert bit_equal(db[to_host(plane)], want), (n, n_posts, mult)
        # scattered: block b at a line offset of its own (a packed plane with other blocks in between), untouched bytes stay
        offs = (rng.permutation(count).astype(np.uint32) * np.uint32(2 * n) + np.uint32(4 * 3))
        big = to_dev(np.full(2 * n * count + 64, 0x5A, np.uint8))
        v.floor1(xs, mult, to_dev(ys), n, None, count, y_plane=big, line_offsets=to_dev(offs))
        got = to_host(big)
        seen = np.zeros(got.size, bool)
        for b in range(count):
            assert bit_equal(db[got[offs[b]:offs[b] + n]], want[b]), (n, b)
            seen[offs[b]:offs[b] + n] = True
        assert np.all(got[~seen] == 0x5A)


def test_emu_floor1_y_plane(emu_ctx):
    check_floor1_y(emu_ctx, lambda a: a, lambda a: a)


def check_floor1_y_jobs(ctx, to_dev, to_host):
    """symaccel_vorbis_floor1_y_jobs_device: several (floor, block size) renders into one plane, two per launch, == one call each ==
    the oracle; an empty job is skipped; a malformed job fails the call before anything is written"""
    from symphonia_amd import SymaccelError
    rng = np.random.default_rng(707)
    v = VorbisDsp(ctx, 8, 11)
    db = db_table()
    for shapes in (((1024, 40, 2, 130), (128, 12, 2, 70)), ((1024, 30, 1, 5), (2048, 65, 1, 9), (64, 4, 3, 200)), ((128, 9, 4, 1),),
                   ((256, 20, 2, 0), (512, 33, 1, 66), (4096, 50, 2, 3), (128, 2, 1, 65))):
        jobs, want, base = [], [], 16
        for n, n_posts, mult, count in shapes:
            xs, ys = floor_case(rng, n, n_posts, mult, max(count, 1), 0.25)
            ys = ys[:count]
            offs = (rng.permutation(max(count, 1))[:count].astype(np.uint32) * np.uint32(n + 16) + np.uint32(base))
            base += (max(count, 1)) * (n + 16) + 32
            jobs.append((xs, mult, to_dev(ys) if count else None, n, to_dev(offs) if count else None, count))
            want.append((n, offs, [oracle.vorbis_floor1(xs, y, mult, n) for y in ys]))
        plane = to_dev(np.full(base + 64, 0x5A, np.uint8))
        v.floor1_y_jobs(jobs, plane)
        got = to_host(plane)
        seen = np.zeros(got.size, bool)
        for n, offs, curves in want:
            for o, c in zip(offs, curves):
                assert bit_equal(db[got[o:o + n]], c), (shapes, n, o)
                seen[o:o + n] = True
        assert np.all(got[~seen] == 0x5A), shapes
        # == one call per job
        single = to_dev(np.full(base + 64, 0x5A, np.uint8))
        for xs, mult, ys, n, offs, count in jobs:
            if count:
                v.floor1(xs, mult, ys, n, None, count, y_plane=single, line_offsets=offs)
        assert np.array_equal(to_host(single), got)
    # the second job's x list has a repeated value (render_line would divide by zero): nothing is launched
    xs, ys = floor_case(rng, 128, 6, 1, 3, 0.0)
    plane = to_dev(np.full(3 * 128, 0x5A, np.uint8))
    with pytest.raises(SymaccelError):
        v.floor1_y_jobs([(xs, 1, to_dev(ys), 128, None, 3), ([0, 128, 7, 7, 9, 11], 1, to_dev(ys), 128, None, 3)], plane)
    ctx.sync()
    assert np.all(to_host(plane) == 0x5A)


def test_emu_floor1_y_jobs(emu_ctx):
    check_floor1_y_jobs(emu_ctx, lambda a: a, lambda a: a)


def fy_case(rng, bs0e, bs1e, nch, nb):
    from test_emu_codecs import vorbis_case
    flags, prev, residue, overlap, pcm_stride = vorbis_case(rng, bs0e, bs1e, nch, nb)
    pad = (-residue.shape[1]) % 4
    residue = np.pad(residue, ((0, 0), (0, pad)))
    residue[rng.random(residue.shape) < 0.1] = 0.0
    residue[0, ::7] = -0.0
    pcm_stride += (-pcm_stride) % 4
    ypl = rng.integers(0, 256, residue.shape).astype(np.uint8)
    return flags, prev, residue, overlap, pcm_stride, ypl


PAIRS = [(8, 11, 3), (6, 9, 4), (7, 10, 2), (8, 8, 1), (7, 12, 2), (9, 12, 3), (10, 13, 2), (12, 13, 2)]


@pytest.mark.parametrize("bs0e,bs1e,seg", PAIRS)
def test_emu_synth_floor_y(emu_ctx, bs0e, bs1e, seg):
    rng = np.random.default_rng(700 + 16 * bs0e + bs1e)
    flags, prev, residue, overlap, pcm_stride, ypl = fy_case(rng, bs0e, bs1e, 3, 9 if bs1e < 12 else 5)
    v = VorbisDsp(emu_ctx, bs0e, bs1e)
    want = oracle.vorbis_synth(bs0e, bs1e, db_table()[ypl] * residue, flags, prev, overlap, pcm_stride)
    for pp in (False, True):
        emu_ctx.set_segment(seg)
        pf, ov = prev.copy(), overlap.copy()
        pcm = np.zeros((flags.shape[0], pcm_stride), np.float32)
        if pp:
            pf_out, ov_out = np.empty_like(pf), np.empty_like(ov)
            v.synth_floor_y(ypl, residue, flags, pf, ov, pcm_stride, pcm, state_out=(pf_out, ov_out))
            pf, ov = pf_out, ov_out
        else:
            v.synth_floor_y(ypl, residue, flags, pf, ov, pcm_stride, pcm)
        emu_ctx.set_segment(0)
        assert bit_equal(pcm, want[0]) and bit_equal(ov, want[1]) and np.array_equal(pf, want[2]), pp


def pipeline_case(rng, bs0e, bs1e, nch, nb):
    """a mixed stream: posts per channel-block, the packed line offsets of every block, residue"""
    flags, prev, residue, overlap, pcm_stride, _ = fy_case(rng, bs0e, bs1e, nch, nb)
    so = oracle.vorbis_layout(bs0e, bs1e, flags, prev)[0]  # spectrum offsets [chain][nb + 1]
    stride = residue.shape[1]
    classes = {}
    for flag, e in ((0, bs0e), (1, bs1e)):
        n = (1 << e) // 2
        n_posts, mult = (12, 2) if flag == 0 else (40, 1)
        xs = [0, n] + rng.permutation(np.arange(1, n))[:n_posts - 2].tolist()
        where = np.argwhere(flags == flag)
        ys = rng.integers(0, [256, 128][mult - 1], size=(len(where), n_posts)).astype(np.uint32)
        ys[rng.random(ys.shape) < 0.2] = 0
        offs = np.array([c * stride + so[c, b] for c, b in where], np.uint32)
        classes[flag] = (n, xs, mult, ys, offs)
    return flags, prev, residue, overlap, pcm_stride, classes


def run_pipeline(ctx, to_dev, to_host, bs0e, bs1e, seed):
    rng = np.random.default_rng(seed)
    flags, prev, residue, overlap, pcm_stride, classes = pipeline_case(rng, bs0e, bs1e, 4, 23)
    v = VorbisDsp(ctx, bs0e, bs1e)
    spectrum = np.zeros_like(residue)
    plane = to_dev(np.zeros(residue.shape, np.uint8))
    for flag, (n, xs, mult, ys, offs) in classes.items():
        if len(ys) == 0:
            continue
        v.floor1(xs, mult, to_dev(ys), n, None, len(ys), y_plane=plane, line_offsets=to_dev(offs))  # one call per block-size class
        flat = spectrum.reshape(-1)
        for y, o in zip(ys, offs):
            flat[o:o + n] = oracle.vorbis_floor1(xs, y, mult, n) * residue.reshape(-1)[o:o + n]  # lib.rs:289-291
    want = oracle.vorbis_synth(bs0e, bs1e, spectrum, flags, prev, overlap, pcm_stride)
    pf, ov = to_dev(prev.copy()), to_dev(overlap.copy())
    pcm = to_dev(np.zeros((flags.shape[0], pcm_stride), np.float32))
    v.synth_floor_y(plane, to_dev(residue), to_dev(flags), pf, ov, pcm_stride, pcm)
    assert bit_equal(to_host(pcm), want[0]) and bit_equal(to_host(ov), want[1]) and np.array_equal(to_host(pf), want[2])
    # the f32 form of the same pipeline: curve x residue written at the blocks' packed offsets, then plain synthesis
    spec_dev = to_dev(np.zeros_like(residue))
    d_res = to_dev(residue)
    for flag, (n, xs, mult, ys, offs) in classes.items():
        if len(ys):
            v.floor1(xs, mult, to_dev(ys), n, spec_dev, len(ys), residue=d_res, line_offsets=to_dev(offs))
    assert bit_equal(to_host(spec_dev), spectrum)


@pytest.mark.parametrize("bs0e,bs1e", [(8, 11), (7, 10)])
def test_emu_floor_posts_to_pcm(emu_ctx, bs0e, bs1e):
    """floor-1 posts + residue -> PCM in two kernels with a byte plane in between == the reference's curve, dot product, synthesis"""
    run_pipeline(emu_ctx, lambda a: a, lambda a: a, bs0e, bs1e, 41 + bs1e)


def test_emu_floor_y_rejects_what_it_cannot_align(emu_ctx):
    from symphonia_amd import SymaccelError
    rng = np.random.default_rng(3)
    flags, prev, residue, overlap, pcm_stride, ypl = fy_case(rng, 8, 11, 2, 5)
    v = VorbisDsp(emu_ctx, 8, 11)
    pcm = np.zeros((2, pcm_stride), np.float32)
    odd = np.ascontiguousarray(residue[:, :-1])  # a stride that is not a multiple of four lines
    with pytest.raises(SymaccelError):
        v.synth_floor_y(np.ascontiguousarray(ypl[:, :-1]), odd, flags, prev.copy(), overlap.copy(), pcm_stride, pcm)


# ---- the same on the MI355X
@pytest.fixture(scope="module")
def gpu():
    torch = pytest.importorskip("torch")
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from symphonia_amd import Context
    ctx = Context(0)
    ctx.use_torch_stream()
    dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()  # noqa: E731
    host = lambda t: t.cpu().numpy()  # noqa: E731
    return ctx, dev, host


@pytest.mark.gpu
def test_gpu_floor1_y_plane(gpu):
    check_floor1_y(*gpu)


@pytest.mark.gpu
def test_gpu_floor1_y_jobs(gpu):
    check_floor1_y_jobs(*gpu)


@pytest.mark.gpu
@pytest.mark.parametrize("bs0e,bs1e,seg", PAIRS + [(8, 11, 0), (6, 9, 0), (10, 13, 0)])
def test_gpu_synth_floor_y(gpu, bs0e, bs1e, seg):
    ctx, dev, host = gpu
    rng = np.random.default_rng(900 + 16 * bs0e + bs1e)
    flags, prev, residue, overlap, pcm_stride, ypl = fy_case(rng, bs0e, bs1e, 6, 40 if bs1e < 12 else 9)
    v = VorbisDsp(ctx, bs0e, bs1e)
    want = oracle.vorbis_synth(bs0e, bs1e, db_table()[ypl] * residue, flags, prev, overlap, pcm_stride)
    ctx.set_segment(seg)
    pf, ov = dev(prev), dev(overlap)
    pf_out, ov_out = dev(np.empty_like(prev)), dev(np.empty_like(overlap))
    pcm = dev(np.zeros((flags.shape[0], pcm_stride), np.float32))
    v.synth_floor_y(dev(ypl), dev(residue), dev(flags), pf, ov, pcm_stride, pcm, state_out=(pf_out, ov_out))
    ctx.set_segment(0)
    assert bit_equal(host(pcm), want[0]) and bit_equal(host(ov_out), want[1]) and np.array_equal(host(pf_out), want[2])


@pytest.mark.gpu
@pytest.mark.parametrize("bs0e,bs1e", [(8, 11), (7, 10), (9, 12)])
def test_gpu_floor_posts_to_pcm(gpu, bs0e, bs1e):
    run_pipeline(*gpu, bs0e, bs1e, 141 + bs1e)
