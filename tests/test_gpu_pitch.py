"""GPU parity for zj_frame_desc.out_pitch (rows laid out wider than they are, for outputs that stay in HBM): every call
through the C ABI of libzjhip.so, every row against the oracle's row (the tight layout is the reference's,
/root/reference/src/mcu.rs:375-379), the bytes between a row's end and the next row's start against the fill the test put
there.  A pitch that is a multiple of 128 bytes is what the layout exists for (DESIGN.md 4.0 "row pitch"); any other is
legal too."""
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


def _variants():
    """the kernel variants the library under test carries: the product build has 0 (packed, staged stores) and 2 (packed,
    direct stores); `make VARIANTS=all` adds 1 (round 1's wide generation), the N-version cross-check of earlier rounds"""
    names = {0: "packed", 1: "wide", 2: "packed-direct"}
    have = importlib.import_module("zune-jpeg_amd").variants_available()
    return have, [names[v] for v in have]


@pytest.fixture(scope="module", params=_variants()[0], ids=_variants()[1])
def ctx(zj, request):
    c = zj.Context(zj.BACKEND_HIP, 0)
    c.set_variant(request.param)
    yield c
    c.close()


def expected(synth, w, h, hs, vs, out_cs, flags, layout, qts, planes):
    """the tight bytes, as rows: (planes * h, row_bytes)"""
    f = oc.make_frame(w, h, hs, vs, 3, out_cs, qts)
    ext = flags != 0 or layout == 1 or out_cs in (oc.RGBA,)
    rc, exp = oc.decode_planes(f, planes, plain=True) if ext else oc.decode_planes(f, planes)
    if rc != 0:
        return rc, None
    ncomp = {oc.RGB: 3, oc.GRAYSCALE: 1, oc.YCBCR: 3, oc.RGBA: 4}[out_cs]
    if layout == 1:
        return 0, np.ascontiguousarray(exp.reshape(h, w, 3).transpose(2, 0, 1)).reshape(3 * h, w)
    return 0, exp.reshape(h, w * ncomp)


def run_device(zj, ctx, d, planes, n=1, fill=0xAA):
    out_len = zj.lib().zj_out_len(C.byref(d))
    bufs = [ctx.device_alloc(max(p.nbytes, 16)) for p in planes] + [ctx.device_alloc(n * out_len)]
    try:
        for b, p in zip(bufs, planes):
            ctx.h2d(b, p)
        zj.lib().zj_device_memset(ctx.handle, bufs[3], fill, n * out_len)
        ctx.decode_planes_device(d, n, bufs[0], bufs[1], bufs[2], bufs[3])
        ctx.sync()
        got = np.empty(n * out_len, np.uint8)
        ctx.d2h(got, bufs[3])
        return got
    finally:
        for b in bufs:
            ctx.device_free(b)


@pytest.mark.parametrize("mode", list(MODES))
@pytest.mark.parametrize("kind", ["rgb", "gray", "ycbcr", "plain", "rgba", "chw"])
@pytest.mark.parametrize("wh", [(272, 100), (64, 64), (1040, 33), (2500, 70), (303, 40), (37, 50)])
def test_padded_rows_on_the_device_vs_oracle(ctx, zj, synth, mode, kind, wh):
    hs, vs = MODES[mode]
    w, h = wh
    out_cs, flags, layout, ncomp = {"rgb": (oc.RGB, 0, 0, 3), "gray": (oc.GRAYSCALE, 0, 0, 1), "ycbcr": (oc.YCBCR, 0, 0, 3),
                                    "plain": (oc.RGB, 1, 0, 3), "rgba": (oc.RGBA, 0, 0, 4), "chw": (oc.RGB, 0, 1, 3)}[kind]
    planes, qts = synth.make_frame(w, h, hs, vs, 3, seed=w + h)
    rc, exp = expected(synth, w, h, hs, vs, out_cs, flags, layout, qts, planes)
    row = w if layout == 1 else w * ncomp
    for pitch in ((row + 127) // 128 * 128, (row + 15) // 16 * 16 + 16):
        d = zj.FrameDesc.make(w, h, hs, vs, 3, out_cs, qts, flags=flags, out_layout=layout, out_pitch=pitch)
        if rc != 0:
            with pytest.raises(zj.ZjError) as e:
                run_device(zj, ctx, d, planes)
            assert e.value.status == -5
            return
        assert zj.lib().zj_out_len(C.byref(d)) == pitch * h * (3 if layout == 1 else 1)
        got = run_device(zj, ctx, d, planes).reshape(-1, pitch)
        bad = np.nonzero(got[:, :row] != exp)
        assert bad[0].size == 0, (mode, kind, wh, pitch, bad[0][:5], bad[1][:5])
        # the padding: untouched, except in rows the strips never reach (zeroed whole with the row, Q6)
        pad = got[:, row:]
        rows_touched = (pad != 0xAA).any(axis=1)
        assert not pad[~rows_touched].size or (pad[~rows_touched] == 0xAA).all()
        assert (pad[rows_touched] == 0).all() and not exp[rows_touched].any(), (mode, kind, wh, pitch)


def test_padded_rows_batches_scattered_strided_and_multi(zj, synth):
    """the other device entry points take the same descriptor: a contiguous batch (frame stride = out_pitch * height), a
    scattered batch, a strided one, two device slots -- all equal to the one-frame result"""
    ctx = zj.Context(zj.BACKEND_HIP, 0)
    w, h, n = 2500, 40, 5
    frames = [synth.make_frame(w, h, 2, 2, 3, seed=9, frame_index=i) for i in range(n)]
    qts = frames[0][1]
    pitch = (3 * w + 127) // 128 * 128
    d = zj.FrameDesc.make(w, h, 2, 2, 3, zj.ColorSpace.RGB, qts, out_pitch=pitch)
    out_len = zj.lib().zj_out_len(C.byref(d))
    assert out_len == pitch * h
    f = oc.make_frame(w, h, 2, 2, 3, oc.RGB, qts)
    want = []
    for fr in frames:
        e = np.full((h, pitch), 0x5C, np.uint8)
        e[:, :3 * w] = oc.decode_planes(f, fr[0])[1].reshape(h, 3 * w)
        e[32:] = 0   # rows below the last complete strip (Q6) are zeroed whole, padding included
        want.append(e.reshape(-1))
    cat = [np.concatenate([fr[0][c] for fr in frames]) for c in range(3)]
    bufs = [ctx.device_alloc(p.nbytes) for p in cat] + [ctx.device_alloc(n * (out_len + 4096))]
    try:
        for b, p in zip(bufs, cat):
            ctx.h2d(b, p)
        ylen, clen = frames[0][0][0].size, frames[0][0][1].size
        for which in ("batch", "scattered", "strided", "multi"):
            stride = out_len if which in ("batch", "multi") else out_len + 4096
            zj.lib().zj_device_memset(ctx.handle, bufs[3], 0x5C, n * (out_len + 4096))
            if which == "batch":
                ctx.decode_planes_device(d, n, bufs[0], bufs[1], bufs[2], bufs[3])
            elif which == "strided":
                ctx.decode_planes_device_strided(d, n, bufs[0], bufs[1], bufs[2], bufs[3], ylen, clen, stride)
            elif which == "scattered":
                order = [3, 0, 4, 1, 2]
                ctx.decode_frames_device(d, [bufs[0] + 2 * i * ylen for i in order], [bufs[1] + 2 * i * clen for i in order],
                                         [bufs[2] + 2 * i * clen for i in order], [bufs[3] + i * stride for i in order])
            else:
                m = zj.Multi([0, 0])
                try:
                    m.decode_frames_device(d, [bufs[0] + 2 * i * ylen for i in range(n)], [bufs[1] + 2 * i * clen for i in range(n)],
                                           [bufs[2] + 2 * i * clen for i in range(n)], [bufs[3] + i * stride for i in range(n)])
                finally:
                    m.close()
            ctx.sync()
            got = np.empty(n * (out_len + 4096), np.uint8)
            ctx.d2h(got, bufs[3])
            for i in range(n):
                assert np.array_equal(got[i * stride:i * stride + out_len], want[i]), (which, i)
        # host outputs are tight: a padded pitch is refused there, and so are pitches no kernel can serve
        with pytest.raises(zj.ZjError) as e:
            ctx.decode_planes(d, frames[0][0])
        assert e.value.status == -2
        for bad in (3 * w - 1, (1 << 20) + 128):
            with pytest.raises(zj.ZjError) as e:
                ctx.decode_planes_device(zj.FrameDesc.make(w, h, 2, 2, 3, zj.ColorSpace.RGB, qts, out_pitch=bad), 1, bufs[0], bufs[1], bufs[2], bufs[3])
            assert e.value.status == -1
        with pytest.raises(zj.ZjError) as e:   # aligned widths: rows of the aligned kernels start on 16-byte boundaries
            ctx.decode_planes_device(zj.FrameDesc.make(2496, h, 2, 2, 3, zj.ColorSpace.RGB, qts, out_pitch=3 * 2496 + 8), 1, bufs[0], bufs[1], bufs[2], bufs[3])
        assert e.value.status == -1
    finally:
        for b in bufs:
            ctx.device_free(b)
        ctx.close()


@pytest.mark.parametrize("kind", ["hwc", "chw", "gray", "rgba"])
@pytest.mark.parametrize("padded", [False, True])
def test_tensor_views_of_the_output(zj, synth, kind, padded):
    """zune-jpeg_amd/tensors.py: the decoded bytes as [N, H, W, C] / [N, 3, H, W] / [N, H, W] uint8 tensors, tight or strided
    over a padded pitch; .contiguous() of the padded view is the tight tensor"""
    import torch
    tz = importlib.import_module("zune-jpeg_amd.tensors")
    ctx = zj.Context(zj.BACKEND_HIP, 0)
    try:
        w, h, n = 2500, 70, 3
        dev = torch.device("cuda", 0)
        out_cs, layout, flags = {"hwc": (oc.RGB, 0, 0), "chw": (oc.RGB, 1, 0), "gray": (oc.GRAYSCALE, 0, 0), "rgba": (oc.RGBA, 0, 0)}[kind]
        frames = [synth.make_frame(w, h, 2, 2, 3, seed=21, frame_index=i) for i in range(n)]
        qts = frames[0][1]
        d = zj.FrameDesc.make(w, h, 2, 2, 3, out_cs, qts, out_layout=layout)
        if padded:
            d = tz.padded_desc(d)
            assert d.out_pitch % 128 == 0 and 0 <= d.out_pitch - tz.row_bytes(d) < 128
        planes = [torch.from_numpy(np.concatenate([fr[0][c] for fr in frames])).to(dev) for c in range(3)]
        view = tz.decode_to_tensor(ctx, d, planes, n)
        torch.cuda.synchronize()
        got = view.contiguous().cpu().numpy()
        for i, fr in enumerate(frames):
            rc, exp = expected(synth, w, h, 2, 2, out_cs, flags, layout, qts, fr[0])
            assert rc == 0
            shape = {"hwc": (h, w, 3), "chw": (3, h, w), "gray": (h, w), "rgba": (h, w, 4)}[kind]
            assert got[i].shape == shape and np.array_equal(got[i].reshape(exp.shape), exp), (kind, padded, i)
    finally:
        ctx.close()


def test_random_descriptors_the_gpu_and_the_emulation_agree(zj, synth):
    """Random descriptors, legal and not (sampling factors, colour spaces, flags, layouts, pitches), through
    zj_decode_planes_device and through the CPU emulation of the same kernels (tests/emu): the same verdict, and where it is
    ZJ_OK the same bytes -- padding included."""
    import emu_c
    rng = np.random.default_rng(777)
    ctx = zj.Context(zj.BACKEND_HIP, 0)
    L = emu_c.lib()
    verdicts = {}
    try:
        for it in range(250):
            w, h = int(rng.integers(1, 700)), int(rng.integers(1, 100))
            hs, vs = int(rng.choice([1, 2, 1, 2, 3])), int(rng.choice([1, 2, 1, 2, 4]))
            out_cs = int(rng.choice([oc.RGB, oc.GRAYSCALE, oc.YCBCR, oc.RGBA, oc.CMYK, 9]))
            flags = int(rng.choice([0, 0, 1, 2, 4, 7, 8]))
            layout = int(rng.choice([0, 0, 0, 1, 2]))
            ncomp = {oc.RGB: 3, oc.GRAYSCALE: 1, oc.YCBCR: 3}.get(out_cs, 4)
            row = w if (layout == 1 and out_cs == oc.RGB) else w * ncomp
            pitch = int(rng.choice([0, 0, row, row + 16, (row + 127) // 128 * 128, max(row - 1, 1), row + int(rng.integers(1, 300))]))
            geo = hs in (1, 2) and vs in (1, 2)
            planes, qts = synth.make_frame(w, h, hs if geo else 1, vs if geo else 1, 3, seed=1000 + it)
            planes = [np.ascontiguousarray(p, np.int16) for p in planes]
            d = zj.FrameDesc.make(w, h, hs, vs, 3, out_cs, qts, flags=flags, out_layout=layout, out_pitch=pitch)
            cap = max(pitch, 4 * w) * h * 3 + 64
            # emulation
            e = emu_c.FrameDesc()
            C.memmove(C.byref(e), C.byref(d), C.sizeof(e))
            want = np.full(cap, 0x5C, np.uint8)
            rc_e = L.zje_decode_planes(C.byref(e), C.c_size_t(1), C.c_void_p(planes[0].ctypes.data), C.c_void_p(planes[1].ctypes.data),
                                       C.c_void_p(planes[2].ctypes.data), C.c_void_p(want.ctypes.data), C.c_int(1))
            # GPU
            bufs = [ctx.device_alloc(max(p.nbytes, 16)) for p in planes] + [ctx.device_alloc(cap)]
            try:
                for b, p in zip(bufs, planes):
                    ctx.h2d(b, p)
                zj.lib().zj_device_memset(ctx.handle, bufs[3], 0x5C, cap)
                try:
                    ctx.decode_planes_device(d, 1, bufs[0], bufs[1], bufs[2], bufs[3])
                    ctx.sync()
                    rc_g = 0
                except zj.ZjError as err:
                    rc_g = err.status
                got = np.empty(cap, np.uint8)
                ctx.d2h(got, bufs[3])
            finally:
                for b in bufs:
                    ctx.device_free(b)
            assert rc_g == rc_e, (it, rc_g, rc_e, w, h, hs, vs, out_cs, flags, layout, pitch)
            verdicts[rc_g] = verdicts.get(rc_g, 0) + 1
            if rc_g == 0:
                assert np.array_equal(got, want), (it, w, h, hs, vs, out_cs, flags, layout, pitch, np.nonzero(got != want)[0][:5])
    finally:
        ctx.close()
    assert verdicts.get(0, 0) > 30 and verdicts.get(-1, 0) > 20, verdicts


def test_scan_entry_points_refuse_a_padded_pitch_for_host_outputs(zj, synth):
    """zj_decode_scan / zj_decode_scans copy pitch x height bytes out of a reused staging arena for a host output;
    the kernels never write the padding, so stale pixels of earlier decodes would travel.  A padded pitch is a device-only
    layout (include/zjhip.h): refused for host outputs, served for device outputs."""
    import ctypes as C
    data = open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "test-baseline.jpg"), "rb").read()
    ctx = zj.Context(zj.BACKEND_HIP, 0)
    o = zj.ZuneJpegOptions()
    o.entropy = zj.ENTROPY_GPU_ALWAYS
    dec = zj.Decoder(o, ctx)
    try:
        desc, info = dec.prepare(data)
        blob = dec.scan_blob()
        assert blob is not None
        tight, rc, st = ctx.decode_scan(desc, blob)
        assert rc == 0
        padded = zj.FrameDesc.make(info.width, info.height, info.h_max, info.v_max, 3, zj.ColorSpace.RGB,
                                   list(np.ctypeslib.as_array(desc.qt)), out_pitch=3 * info.width + 128)
        with pytest.raises(zj.ZjError) as e:
            ctx.decode_scan(padded, blob)
        assert e.value.status == -2                                   # ZJ_ERR_UNSUPPORTED
        out_len = zj.lib().zj_out_len(C.byref(padded))
        buf = ctx.device_alloc(out_len)
        try:
            zj.lib().zj_device_memset(ctx.handle, buf, 0x5A, out_len)
            _, rc, st = ctx.decode_scan(padded, blob, device_out=buf)
            assert rc == 0
            ctx.sync()
            got = np.empty(out_len, np.uint8)
            ctx.d2h(got, buf)
            rows = got.reshape(info.height, 3 * info.width + 128)
            assert np.array_equal(rows[:, :3 * info.width].reshape(-1), tight) and (rows[:, 3 * info.width:] == 0x5A).all()
        finally:
            ctx.device_free(buf)
    finally:
        dec.close()
        ctx.close()
