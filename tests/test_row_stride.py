"""Padded row pitch for the lane-per-block planes (ABI 8): symaccel_flac_restore_strided_device, symaccel_alac_predict_strided_device,
symaccel_row_stride.

The arithmetic is the reference's (symphonia-bundle-flac/src/decoder.rs:199-242, 663-752; symphonia-codec-alac/src/lib.rs:165-264,
541-560, 664-671); the only new thing is WHERE a row starts, so the checks are: the first `blocksize` words of every row equal the
oracle on the compact rows bit for bit, and the padding words come back as they were (the kernels neither read nor write them).
CPU emulation here, gpu-marked twins on the MI355X."""
import numpy as np
import pytest

import oracle
from emu_lib import emu_ctx, emu_library  # noqa: F401
from test_alac import alac_case

FLAC_VERBATIM, FLAC_FIXED, FLAC_LPC = 0, 1, 2
SENTINEL = np.int32(0x5A5A5A5A)


def flac_case(seed, nb, blocksize):
    rng = np.random.default_rng(seed)
    buf = rng.integers(-(1 << 20), 1 << 20, (nb, blocksize)).astype(np.int32)
    kind = rng.integers(0, 3, nb).astype(np.uint8)
    order = np.minimum(np.where(kind == FLAC_FIXED, rng.integers(0, 5, nb), rng.integers(1, 33, nb)), blocksize).astype(np.uint8)
    kind[(kind == FLAC_LPC) & (order == 0)] = FLAC_VERBATIM
    shift = rng.integers(0, 16, nb).astype(np.uint8)
    wasted = np.where(rng.random(nb) < 0.2, rng.integers(1, 4, nb), 0).astype(np.uint8)
    coeffs = rng.integers(-(1 << 14), 1 << 14, (nb, 32)).astype(np.int32)
    coeffs[nb // 2:] *= 16  # the second half of the blocks: magnitudes summing past 2^20, the v_mad_i64_i32 kernel's wavefronts
    mode = rng.integers(0, 4, nb // 2).astype(np.uint8)
    return buf, kind, order, shift, wasted, coeffs, mode


def padded(buf, stride):
    out = np.full((buf.shape[0], stride), SENTINEL, np.int32)
    out[:, :buf.shape[1]] = buf
    return out


def flac_want(buf, kind, order, shift, wasted, coeffs, mode=None, out_shift=0):
    want = oracle.flac_restore(buf, oracle.flac_desc(kind, order, shift, wasted), coeffs)
    if mode is not None:
        for p in range(len(mode)):
            a, b = oracle.flac_decorrelate(int(mode[p]), want[2 * p], want[2 * p + 1])
            want[2 * p], want[2 * p + 1] = oracle.flac_shl(a, out_shift), oracle.flac_shl(b, out_shift)
    return want


def alac_want(buf, mode, order, shift, bps, coeffs, weight=None, msh=None):
    want = oracle.alac_predict(buf, oracle.alac_desc(mode, order, shift, bps), coeffs)
    if weight is not None:
        for p in range(len(weight)):
            if weight[p]:
                want[2 * p], want[2 * p + 1] = oracle.alac_decorrelate_mid_side(want[2 * p], want[2 * p + 1], int(weight[p]), int(msh[p]))
    return want


def check(got, want, blocksize):
    assert np.array_equal(got[:, :blocksize], want), np.argwhere(got[:, :blocksize] != want)[:5]
    assert np.all(got[:, blocksize:] == SENTINEL), "padding words were written"


# (blocksize, stride, blocks): aligned pitches (the 16-byte tile path), a pitch that is not a multiple of four (the ragged path), stride ==
# blocksize (the compact form through the new entry), block sizes that are not a multiple of the 32-sample tile, a ragged last wavefront
SHAPES = [(64, 80, 64), (100, 104, 70), (192, 192, 6), (31, 45, 130), (128, 131, 66), (33, 36, 64), (1, 4, 66)]


def test_row_stride_arithmetic():
    """symaccel_row_stride: >= blocksize, a multiple of four, an eighth of padding for rows of 1024 samples and more that are a multiple of 2 KiB long."""
    dll = emu_library().dll
    for bs in [0, 1, 3, 4, 33, 511, 512, 513, 576, 1021, 1024, 1152, 2048, 4093, 4095, 4096, 4097, 4608, 8192, 16384, 65535]:
        s = dll.symaccel_row_stride(bs)
        r4 = (bs + 3) // 4 * 4
        assert s >= bs and s % 4 == 0, (bs, s)
        assert s == (r4 + r4 // 8 if r4 >= 1024 and r4 % 512 == 0 else r4), (bs, s)
    assert dll.symaccel_row_stride(4096) == 4608 and dll.symaccel_row_stride(4000) == 4000 and dll.symaccel_row_stride(512) == 512


@pytest.mark.parametrize("blocksize,stride,nb", SHAPES)
def test_emu_flac_restore_strided(emu_ctx, blocksize, stride, nb):
    from symphonia_amd import FlacPredictor, flac_desc
    buf, kind, order, shift, wasted, coeffs, mode = flac_case(blocksize * 7 + stride, nb, blocksize)
    desc = flac_desc(kind, order, shift, wasted)
    got = padded(buf, stride)
    FlacPredictor(emu_ctx).restore_strided(got, desc, coeffs, blocksize)
    check(got, flac_want(buf, kind, order, shift, wasted, coeffs), blocksize)
    got = padded(buf, stride)
    FlacPredictor(emu_ctx).restore_strided(got, desc, coeffs, blocksize, pair_mode=mode, out_shift=8)
    check(got, flac_want(buf, kind, order, shift, wasted, coeffs, mode, 8), blocksize)


@pytest.mark.parametrize("blocksize,stride,nb", SHAPES)
def test_emu_alac_predict_strided(emu_ctx, blocksize, stride, nb):
    from symphonia_amd import AlacPredictor, alac_desc
    buf, mode, order, shift, bps, coeffs = alac_case(blocksize * 3 + stride, nb, blocksize)
    desc = alac_desc(mode, order, shift, bps)
    got = padded(buf, stride)
    AlacPredictor(emu_ctx).predict_strided(got, desc, coeffs, blocksize)
    check(got, alac_want(buf, mode, order, shift, bps, coeffs), blocksize)
    rng = np.random.default_rng(stride)
    weight, msh = rng.integers(-3, 4, nb // 2).astype(np.int32), rng.integers(0, 32, nb // 2).astype(np.uint8)
    got = padded(buf, stride)
    AlacPredictor(emu_ctx).predict_strided(got, desc, coeffs, blocksize, weight, msh)
    check(got, alac_want(buf, mode, order, shift, bps, coeffs, weight, msh), blocksize)


def test_emu_alac_strided_uniform_order_8(emu_ctx):
    """the steady instantiation (every block order 8, narrow) over padded rows"""
    from symphonia_amd import AlacPredictor, alac_desc
    rng = np.random.default_rng(5)
    nb, blocksize, stride = 128, 160, 176
    buf = rng.integers(-200, 200, (nb, blocksize)).astype(np.int32)
    d = (np.zeros(nb, np.uint8), np.full(nb, 8, np.uint8), np.full(nb, 9, np.uint8), np.full(nb, 16, np.uint8))
    coeffs = rng.integers(-300, 300, (nb, 32)).astype(np.int32)
    got = padded(buf, stride)
    AlacPredictor(emu_ctx).predict_strided(got, alac_desc(*d), coeffs, blocksize)
    check(got, alac_want(buf, *d, coeffs), blocksize)


def test_strided_argument_errors(emu_ctx):
    """a pitch below the block size, an odd block count with pairs, one pair array without the other: INVALID_ARG, nothing launched"""
    import ctypes as C
    from symphonia_amd import _ffi
    dll = emu_ctx.lib.dll
    i32, u8 = np.zeros(64, np.int32), np.zeros(64, np.uint8)
    p = lambda a: a.ctypes.data_as(C.c_void_p)
    h = emu_ctx.handle
    assert dll.symaccel_flac_restore_strided_device(h, p(i32), p(u8), p(i32), None, 0, 2, 8, 4) == _ffi.ERR_INVALID_ARG
    assert dll.symaccel_flac_restore_strided_device(h, p(i32), p(u8), p(i32), p(u8), 0, 3, 4, 8) == _ffi.ERR_INVALID_ARG
    assert dll.symaccel_flac_restore_strided_device(h, p(i32), p(u8), p(i32), None, 32, 2, 4, 8) == _ffi.ERR_INVALID_ARG
    assert dll.symaccel_flac_restore_strided_device(h, None, p(u8), p(i32), None, 0, 2, 4, 8) == _ffi.ERR_INVALID_ARG
    assert dll.symaccel_flac_restore_strided_device(h, None, None, None, None, 0, 0, 4, 8) == _ffi.OK
    assert dll.symaccel_alac_predict_strided_device(h, p(i32), p(u8), p(i32), None, None, 2, 8, 4) == _ffi.ERR_INVALID_ARG
    assert dll.symaccel_alac_predict_strided_device(h, p(i32), p(u8), p(i32), p(i32), None, 2, 4, 8) == _ffi.ERR_INVALID_ARG
    assert dll.symaccel_alac_predict_strided_device(h, p(i32), p(u8), p(i32), p(i32), p(u8), 3, 4, 8) == _ffi.ERR_INVALID_ARG
    assert dll.symaccel_alac_predict_strided_device(h, None, None, None, None, None, 0, 4, 8) == _ffi.OK


# ---------------------------------------------------------------------------------------------------------------- MI355X

def _gpu_ctx():
    import torch
    from symphonia_amd import Context
    if not torch.cuda.is_available():
        pytest.fail("no GPU visible: the gpu-marked tests must run on an MI355X (there is no CPU path)")
    return Context(0)


GPU_SHAPES = SHAPES + [(4096, 4160, 200), (4096, 4096 + 4, 130), (1152, 1152 + 64, 300), (4095, 4099, 70)]


@pytest.mark.gpu
@pytest.mark.parametrize("blocksize,stride,nb", GPU_SHAPES)
def test_gpu_flac_restore_strided(blocksize, stride, nb):
    import torch
    from symphonia_amd import FlacPredictor, flac_desc
    buf, kind, order, shift, wasted, coeffs, mode = flac_case(blocksize * 7 + stride, nb, blocksize)
    with _gpu_ctx() as ctx:
        ctx.use_torch_stream()
        desc = torch.from_numpy(flac_desc(kind, order, shift, wasted).view(np.uint8).reshape(-1, 4)).cuda()
        co = torch.from_numpy(coeffs).cuda()
        d = torch.from_numpy(padded(buf, stride)).cuda()
        FlacPredictor(ctx).restore_strided(d, desc, co, blocksize)
        d2 = torch.from_numpy(padded(buf, stride)).cuda()
        FlacPredictor(ctx).restore_strided(d2, desc, co, blocksize, pair_mode=torch.from_numpy(mode).cuda(), out_shift=8)
        torch.cuda.synchronize()
        got, got2 = d.cpu().numpy(), d2.cpu().numpy()
    check(got, flac_want(buf, kind, order, shift, wasted, coeffs), blocksize)
    check(got2, flac_want(buf, kind, order, shift, wasted, coeffs, mode, 8), blocksize)


@pytest.mark.gpu
@pytest.mark.parametrize("blocksize,stride,nb", GPU_SHAPES)
def test_gpu_alac_predict_strided(blocksize, stride, nb):
    import torch
    from symphonia_amd import AlacPredictor, alac_desc
    buf, mode, order, shift, bps, coeffs = alac_case(blocksize * 3 + stride, nb, blocksize)
    rng = np.random.default_rng(stride)
    weight, msh = rng.integers(-3, 4, nb // 2).astype(np.int32), rng.integers(0, 32, nb // 2).astype(np.uint8)
    with _gpu_ctx() as ctx:
        ctx.use_torch_stream()
        desc = torch.from_numpy(alac_desc(mode, order, shift, bps).view(np.uint8).reshape(-1, 4)).cuda()
        co = torch.from_numpy(coeffs).cuda()
        d = torch.from_numpy(padded(buf, stride)).cuda()
        AlacPredictor(ctx).predict_strided(d, desc, co, blocksize)
        d2 = torch.from_numpy(padded(buf, stride)).cuda()
        AlacPredictor(ctx).predict_strided(d2, desc, co, blocksize, torch.from_numpy(weight).cuda(), torch.from_numpy(msh).cuda())
        torch.cuda.synchronize()
        got, got2 = d.cpu().numpy(), d2.cpu().numpy()
    check(got, alac_want(buf, mode, order, shift, bps, coeffs), blocksize)
    check(got2, alac_want(buf, mode, order, shift, bps, coeffs, weight, msh), blocksize)


@pytest.mark.gpu
def test_gpu_strided_equals_compact_at_config5_shape():
    """BASELINE config 5's shape (4096-sample subframes, order-32 LPC, 24 bit) at the recommended pitch: the padded plane's rows ==
    the compact plane's rows, on a sample of the full batch (size-independent: every wavefront runs the same code)."""
    import torch
    from symphonia_amd import FlacPredictor, flac_desc
    rng = np.random.default_rng(55)
    nb, bs = 8192, 4096
    with _gpu_ctx() as ctx:
        ctx.use_torch_stream()
        stride = int(ctx.lib.dll.symaccel_row_stride(bs))
        assert stride > bs
        buf = rng.integers(-(1 << 12), 1 << 12, (nb, bs)).astype(np.int32)
        kind = np.full(nb, FLAC_LPC, np.uint8)
        order, shift = np.full(nb, 32, np.uint8), np.full(nb, 12, np.uint8)
        coeffs = rng.integers(-500, 500, (nb, 32)).astype(np.int32)
        desc = torch.from_numpy(flac_desc(kind, order, shift, 0 * shift).view(np.uint8).reshape(-1, 4)).cuda()
        co = torch.from_numpy(coeffs).cuda()
        compact = torch.from_numpy(buf).cuda()
        FlacPredictor(ctx).restore(compact, desc, co)
        wide = torch.full((nb, stride), int(SENTINEL), dtype=torch.int32, device="cuda")
        wide[:, :bs] = torch.from_numpy(buf).cuda()
        FlacPredictor(ctx).restore_strided(wide, desc, co, bs)
        torch.cuda.synchronize()
        assert torch.equal(wide[:, :bs], compact) and bool((wide[:, bs:] == int(SENTINEL)).all())
        rows = rng.choice(nb, 6, replace=False)
        want = oracle.flac_restore(buf[rows], oracle.flac_desc(kind[rows], order[rows], shift[rows], 0 * shift[rows]), coeffs[rows])
        assert np.array_equal(compact[torch.from_numpy(rows).cuda()].cpu().numpy(), want)
