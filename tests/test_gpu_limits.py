"""GPU parity at the reference's size limits.  ZuneJpegOptions allows 16384 x 16384 by default
(/root/reference/src/options.rs:34-35), the frame header carries u16 dimensions (src/headers.rs:240-248,
src/decoder.rs:652-668), so make_plan accepts up to 65535 x 65535 (csrc/zj_plan.h).  Every call goes through the C ABI;
the checker is the oracle, plus -- where a frame's output passes 2^31 bytes -- the strip-independence property
(src/mcu.rs:225-226), which does not depend on the oracle's own 64-bit arithmetic."""
import ctypes as C
import importlib
import os

import numpy as np
import pytest

import oracle_c as oc

pytestmark = pytest.mark.gpu
MODES = {"none": (1, 1), "h": (2, 1), "v": (1, 2), "hv": (2, 2)}


@pytest.fixture(scope="module")
def zj():
    return importlib.import_module("zune-jpeg_amd")


@pytest.fixture(scope="module")
def ctx(zj):
    c = zj.Context(zj.BACKEND_HIP, 0)
    yield c
    c.close()


def assert_same(out, exp, what):
    if not np.array_equal(out, exp):
        bad = np.nonzero(np.asarray(out).reshape(-1) != np.asarray(exp).reshape(-1))[0]
        raise AssertionError(f"{what}: {bad.size} of {exp.size} values differ, first at {bad[:10]}")


def mem_available_gb():
    try:
        for ln in open("/proc/meminfo"):
            if ln.startswith("MemAvailable:"):
                return int(ln.split()[1]) / 1e6
    except OSError:
        pass
    return 0.0


@pytest.mark.parametrize("mode", list(MODES))
@pytest.mark.parametrize("out_cs", [oc.RGB, oc.GRAYSCALE, oc.YCBCR])
@pytest.mark.parametrize("wh", [(16384, 32), (32, 16384), (65535, 32), (65520, 32), (32, 65535), (65535, 1), (1, 65535), (16, 65504)])
def test_slivers_at_the_dimension_limits(ctx, zj, synth, mode, out_cs, wh):
    """The default limit (16384) and the u16 limit (65535; 65520 = the widest aligned row) in each direction, every
    sampling mode and output: the longest rows (tile columns, the row tail) and the most strips a frame can have."""
    hs, vs = MODES[mode]
    w, h = wh
    planes, qts = synth.make_frame(w, h, hs, vs, 3, seed=w + h)
    rc, exp = oc.decode_planes(oc.make_frame(w, h, hs, vs, 3, out_cs, qts), planes)
    d = zj.FrameDesc.make(w, h, hs, vs, 3, out_cs, qts)
    if rc != 0:
        with pytest.raises(zj.ZjError) as e:
            ctx.decode_planes(d, planes)
        assert e.value.status == -5
        return
    assert_same(ctx.decode_planes(d, planes), exp, (mode, out_cs, wh))


def test_beyond_the_u16_limit_is_an_argument_error(ctx, zj, synth):
    planes, qts = synth.make_frame(64, 64, 2, 2, 3, seed=1)
    for w, h in ((65536, 16), (16, 65536), (0, 16), (16, 0)):
        with pytest.raises(zj.ZjError) as e:
            ctx.decode_planes(zj.FrameDesc.make(w, h, 2, 2, 3, 0, qts), planes)
        assert e.value.status == -1


def _device_frame(zj, synth, ctx, w, h, hs, vs, out_cs, seed):
    """planes generated on the GPU (synth.make_frame_t), decoded resident through zj_decode_planes_device"""
    import torch
    dev = torch.device("cuda", 0)
    planes, qts = synth.make_frame_t(w, h, hs, vs, 3, seed=seed, frame_index=0, device=dev)
    d = zj.FrameDesc.make(w, h, hs, vs, 3, out_cs, qts)
    out = torch.full((zj.lib().zj_out_len(C.byref(d)),), 0xAA, dtype=torch.uint8, device=dev)
    torch.cuda.synchronize()
    ctx.decode_planes_device(d, 1, planes[0].data_ptr(), planes[1].data_ptr(), planes[2].data_ptr(), out.data_ptr())
    ctx.sync()
    return planes, qts, d, out


def _check_strips(zj, ctx, planes, qts, out, w, h, hs, vs, out_cs, ncomp, strips):
    """strip s of the frame decoded as a frame of its own (one strip tall) must give rows [s*SH, (s+1)*SH) of the frame"""
    import torch
    sh = 8 * vs * (2 if hs == 2 else 1)                 # luma rows per strip (two MCU rows when h-subsampled, mcu.rs:145-156)
    mcu_x = (w + 8 * hs - 1) // (8 * hs)
    ystrip, cstrip = mcu_x * hs * 64 * (sh // 8), mcu_x * 64 * (sh // (8 * vs))
    d1 = zj.FrameDesc.make(w, sh, hs, vs, 3, out_cs, qts)
    o1 = torch.empty(w * sh * ncomp, dtype=torch.uint8, device=out.device)
    for s in strips:
        ctx.decode_planes_device(d1, 1, planes[0].data_ptr() + 2 * s * ystrip, planes[1].data_ptr() + 2 * s * cstrip,
                                 planes[2].data_ptr() + 2 * s * cstrip, o1.data_ptr())
        ctx.sync()
        lo = s * sh * w * ncomp
        assert torch.equal(out[lo:lo + o1.numel()], o1), ("strip", s)


def test_default_limit_16384x16384_vs_oracle(ctx, zj, synth):
    """options.rs:34-35: the largest frame the reference decodes by default, 4:2:0 -> RGB (805 MB of pixels), every byte
    against the oracle; then the same planes through the host pipeline (strip-range units over three streams)."""
    w = h = 16384
    planes, qts, d, out = _device_frame(zj, synth, ctx, w, h, 2, 2, zj.ColorSpace.RGB, seed=16384)
    hp = [p.cpu().numpy() for p in planes]
    rc, exp = oc.decode_planes(oc.make_frame(w, h, 2, 2, 3, oc.RGB, qts), hp)
    assert rc == 0
    got = out.cpu().numpy()
    assert_same(got, exp, "16384x16384 device")
    del got, out
    assert_same(ctx.decode_planes(d, hp), exp, "16384x16384 host pipeline")


def test_one_frame_beyond_2_to_the_31_output_bytes(ctx, zj, synth):
    """32768 x 32768 4:2:0 -> RGB: 3.2 GB of pixels in ONE frame, so row offsets, strip offsets and the frame's length all
    pass 2^31 (and 2^32 for the last rows' byte offsets x 1): the kernel's 64-bit address arithmetic (zj_device.h:
    frame_pixels, phase_color's row_bytes, color_copyout's tile_out).  Checked against the oracle on the whole frame when
    the host has the memory for it (planes 3 GB + two outputs 6.4 GB), and always by strip independence, including the
    strips on both sides of the 2^31- and 2^32-byte lines.  Falls back to 24576 x 24576 -> RGBA (2.4 GB) on a small host."""
    import torch
    big = mem_available_gb() >= 24.0
    w = h = 32768 if big else 24576
    out_cs = zj.ColorSpace.RGB if big else zj.ColorSpace.RGBA
    ncomp = 3 if big else 4
    planes, qts, d, out = _device_frame(zj, synth, ctx, w, h, 2, 2, out_cs, seed=777)
    assert out.numel() > 2 ** 31
    sh, row = 32, w * ncomp
    n_strips = h // sh
    edge31, edge32 = 2 ** 31 // (sh * row), 2 ** 32 // (sh * row)
    strips = sorted({0, 1, edge31 - 1, edge31, edge31 + 1, n_strips // 2, n_strips - 2, n_strips - 1} |
                    ({edge32 - 1, edge32, edge32 + 1} if edge32 + 1 < n_strips else set()))
    _check_strips(zj, ctx, planes, qts, out, w, h, 2, 2, out_cs, ncomp, strips)
    # the Q5/Q6 byte pattern on every row: the last 16 bytes of an RGB row are never written by the reference (zeros here)
    rows = out.view(h, row)
    if big:
        assert not bool(rows[:, -16:].any()) and bool(rows[:, :-16].any(dim=1).all())
    if big:
        hp = [p.cpu().numpy() for p in planes]
        rc, exp = oc.decode_planes(oc.make_frame(w, h, 2, 2, 3, oc.RGB, qts), hp)
        assert rc == 0
        del hp
        got = out.cpu().numpy()
        # compare in slabs: a 3.2 GB boolean temporary is not needed to find a difference
        step = 1 << 28
        for lo in range(0, got.size, step):
            assert np.array_equal(got[lo:lo + step], exp[lo:lo + step]), f"first difference in bytes [{lo}, {lo + step})"
