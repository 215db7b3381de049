"""GPU parity for the round-5 entry points, every call through the C ABI of libzjhip.so and bit for bit against the oracle:

  zj_decode_frames_device           frames as independent device allocations, deliberately mis-ordered (the reference's
                                    callers own a fresh Vec per strip / per decode: src/mcu.rs:238-250, src/decoder.rs:178)
  zj_decode_planes_device_strided   frames at a uniform distance
  zj_decode_frames                  the same for host buffers (three-stream pipeline)
  zj_pool_create_multi / zj_multi_* image-level sharding over device slots, here two slots on the box's one GPU
"""
import ctypes as C
import importlib
import os
import sys

import numpy as np
import pytest

import oracle_c as oc

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
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


@pytest.fixture(scope="module")
def ctx0(zj):
    c = zj.Context(zj.BACKEND_HIP, 0)
    yield c
    c.close()


def assert_same(out, exp, what):
    if not np.array_equal(out, exp):
        bad = np.nonzero(np.asarray(out).reshape(-1) != np.asarray(exp).reshape(-1))[0]
        raise AssertionError(f"{what}: {bad.size} of {exp.size} values differ, first at {bad[:10]}")


class DeviceFrames:
    """n frames, every plane and every output its own device allocation, allocated in a shuffled order with odd-sized
    spacers in between so that neither addresses nor distances are regular"""

    def __init__(self, ctx, frames, out_len, seed=0):
        self.ctx, self.n, self.out_len = ctx, len(frames), out_len
        rng = np.random.default_rng(seed)
        self.ptr = [[None] * 4 for _ in frames]
        self.all = []
        jobs = [(f, c) for f in range(self.n) for c in range(4)]
        rng.shuffle(jobs)
        for f, c in jobs:
            nbytes = out_len if c == 3 else frames[f][c].nbytes
            p = ctx.device_alloc(nbytes + 64)
            self.all.append(p)
            self.all.append(ctx.device_alloc(int(rng.integers(1, 5)) * 4096 + 256))  # spacer
            self.ptr[f][c] = p
            if c < 3:
                ctx.h2d(p, frames[f][c])
            else:
                zjm = importlib.import_module("zune-jpeg_amd")
                zjm.lib().zj_device_memset(ctx.handle, p, 0xAA, nbytes)

    def col(self, c, order):
        return [self.ptr[f][c] for f in order]

    def out(self, f):
        got = np.empty(self.out_len, np.uint8)
        self.ctx.d2h(got, self.ptr[f][3])
        return got

    def free(self):
        for p in self.all:
            self.ctx.device_free(p)


@pytest.mark.parametrize("mode", list(MODES))
@pytest.mark.parametrize("out_cs", [oc.RGB, oc.GRAYSCALE, oc.YCBCR])
@pytest.mark.parametrize("wh,n", [((272, 100), 5), ((64, 64), 40), ((1001, 33), 3)])
def test_scattered_frames_on_the_device_vs_oracle(ctx, zj, synth, mode, out_cs, wh, n):
    """All four sampling modes x three outputs; frames at scattered, mis-ordered addresses; 40 frames = more than one
    launch's table (ZJ_SCATTER_MAX = 32); a ragged width takes the generic store path."""
    hs, vs = MODES[mode]
    w, h = wh
    frames = [synth.make_frame(w, h, hs, vs, 3, seed=501, frame_index=i % 6) for i in range(n)]
    qts = frames[0][1]
    f = oc.make_frame(w, h, hs, vs, 3, out_cs, qts)
    d = zj.FrameDesc.make(w, h, hs, vs, 3, out_cs, qts)
    exp = {}
    for i in range(min(n, 6)):
        rc, e = oc.decode_planes(f, frames[i][0])
        if rc != 0:  # the reference panics on this geometry: the same verdict from the scattered entry point
            with pytest.raises(zj.ZjError) as err:
                ctx.decode_frames_device(d, [256], [256], [256], [256])
            assert err.value.status == -5
            return
        exp[i] = e
    df = DeviceFrames(ctx, [fr[0] for fr in frames], zj.lib().zj_out_len(C.byref(d)), seed=n)
    try:
        order = list(np.random.default_rng(7).permutation(n))
        gray = out_cs == oc.GRAYSCALE
        ctx.decode_frames_device(d, df.col(0, order), None if gray else df.col(1, order), None if gray else df.col(2, order), df.col(3, order))
        ctx.sync()
        for i in range(n):
            assert_same(df.out(i), exp[i % 6], (mode, out_cs, wh, "frame", i))
    finally:
        df.free()


def test_scattered_frames_extensions_and_full_size(ctx0, zj, synth):
    """CHW / RGBA / corrected-mode flags go through the same pointer table; and two 4096x4096 4:2:0 frames (configs[1]'s
    shape, the staggered short launch among them) in one scattered launch equal the contiguous batch."""
    ctx = ctx0
    w, h = 528, 72
    frames = [synth.make_frame(w, h, 2, 2, 3, seed=77, frame_index=i) for i in range(3)]
    qts = frames[0][1]
    for out_cs, flags, layout in ((zj.ColorSpace.RGB, 0, zj.LAYOUT_CHW), (zj.ColorSpace.RGBA, 0, 0), (zj.ColorSpace.RGB, zj.FLAG_CORRECTED, 0)):
        d = zj.FrameDesc.make(w, h, 2, 2, 3, out_cs, qts, flags=flags, out_layout=layout)
        want = [ctx.decode_planes(d, fr[0]) for fr in frames]
        df = DeviceFrames(ctx, [fr[0] for fr in frames], want[0].size, seed=3)
        try:
            order = [2, 0, 1]
            ctx.decode_frames_device(d, df.col(0, order), df.col(1, order), df.col(2, order), df.col(3, order))
            ctx.sync()
            for i in range(3):
                assert_same(df.out(i), want[i], (int(out_cs), flags, layout, i))
        finally:
            df.free()
    import torch
    W = H = 4096
    dev = torch.device("cuda", 0)
    planes, outs = [], []
    for i in range(2):
        pl, qts = synth.make_frame_t(W, H, 2, 2, 3, seed=1234, frame_index=900 + i, device=dev)
        planes.append(pl)
        outs.append(torch.empty(W * H * 3 + 4096, dtype=torch.uint8, device=dev))
    d = zj.FrameDesc.make(W, H, 2, 2, 3, zj.ColorSpace.RGB, qts)
    torch.cuda.synchronize()
    ctx.decode_frames_device(d, [planes[1][0].data_ptr(), planes[0][0].data_ptr()], [planes[1][1].data_ptr(), planes[0][1].data_ptr()],
                             [planes[1][2].data_ptr(), planes[0][2].data_ptr()], [outs[1].data_ptr(), outs[0].data_ptr()])
    ctx.sync()
    ref = torch.empty(W * H * 3, dtype=torch.uint8, device=dev)
    for i in range(2):
        ctx.decode_planes_device(d, 1, planes[i][0].data_ptr(), planes[i][1].data_ptr(), planes[i][2].data_ptr(), ref.data_ptr())
        ctx.sync()
        assert torch.equal(outs[i][:W * H * 3], ref), i
    golden = __import__("json").load(open(os.path.join(HERE, "golden", "checksums_seed1234.json")))
    assert synth.frame_checksum_t(outs[0][:W * H * 3]) == int(golden["rgb"][900], 16)


def test_strided_frames_and_the_uniform_shortcut(ctx, zj, synth):
    """zj_decode_planes_device_strided, and scattered pointers that happen to be equally spaced (one launch, strided form):
    the gaps between the frames keep their bytes."""
    w, h, n = 256, 64, 6
    frames = [synth.make_frame(w, h, 2, 2, 3, seed=70, frame_index=i) for i in range(n)]
    qts = frames[0][1]
    d = zj.FrameDesc.make(w, h, 2, 2, 3, zj.ColorSpace.RGB, qts)
    f = oc.make_frame(w, h, 2, 2, 3, oc.RGB, qts)
    exp = [oc.decode_planes(f, fr[0])[1] for fr in frames]
    out_len = exp[0].size
    ylen, clen = frames[0][0][0].size, frames[0][0][1].size
    ys, cs, os_ = ylen + 64, clen + 8, out_len + 4096          # elements, elements, bytes
    hy = np.full(n * ys, 0x1111, np.int16)
    hc = [np.full(n * cs, 0x1111, np.int16) for _ in range(2)]
    for i, fr in enumerate(frames):
        hy[i * ys:i * ys + ylen] = fr[0][0]
        for c in range(2):
            hc[c][i * cs:i * cs + clen] = fr[0][1 + c]
    bufs = [ctx.device_alloc(hy.nbytes), ctx.device_alloc(hc[0].nbytes), ctx.device_alloc(hc[1].nbytes), ctx.device_alloc(n * os_)]
    try:
        ctx.h2d(bufs[0], hy)
        ctx.h2d(bufs[1], hc[0])
        ctx.h2d(bufs[2], hc[1])
        for which in ("strided", "scattered-uniform"):
            zj.lib().zj_device_memset(ctx.handle, bufs[3], 0x5C, n * os_)
            if which == "strided":
                ctx.decode_planes_device_strided(d, n, bufs[0], bufs[1], bufs[2], bufs[3], ys, cs, os_)
            else:
                ctx.decode_frames_device(d, [bufs[0] + 2 * i * ys for i in range(n)], [bufs[1] + 2 * i * cs for i in range(n)],
                                         [bufs[2] + 2 * i * cs for i in range(n)], [bufs[3] + i * os_ for i in range(n)])
            ctx.sync()
            got = np.empty(n * os_, np.uint8)
            ctx.d2h(got, bufs[3])
            for i in range(n):
                assert_same(got[i * os_:i * os_ + out_len], exp[i], (which, i))
                assert (got[i * os_ + out_len:(i + 1) * os_] == 0x5C).all(), (which, "gap", i)
        for bad in ((ylen - 8, cs, os_), (ys + 4, cs, os_), (ys, cs, os_ + 8), (ys, cs, out_len - 16)):
            with pytest.raises(zj.ZjError) as e:
                ctx.decode_planes_device_strided(d, n, bufs[0], bufs[1], bufs[2], bufs[3], *bad)
            assert e.value.status == -1
        with pytest.raises(zj.ZjError) as e:  # a frame pointer that is not 16-byte aligned
            ctx.decode_frames_device(d, [bufs[0], bufs[0] + 2 * ys + 2], [bufs[1], bufs[1] + 2 * cs], [bufs[2], bufs[2] + 2 * cs], [bufs[3], bufs[3] + os_])
        assert e.value.status == -1
        with pytest.raises(zj.ZjError) as e:  # a missing chroma table for a colour output
            ctx.decode_frames_device(d, [bufs[0]], None, None, [bufs[3]])
        assert e.value.status == -1
    finally:
        for b in bufs:
            ctx.device_free(b)


@pytest.mark.parametrize("mode,out_cs,w,h,n", [
    ("hv", oc.RGB, 256, 72, 11),        # odd MCU row: the last 8 rows of every frame stay 0 (Q6); grouped frames
    ("hv", oc.RGB, 4096, 2200, 2),      # 25 MB per frame: each frame is cut into strip ranges
    ("none", oc.YCBCR, 1920, 1080, 3),  # whole frames
    ("h", oc.RGB, 1001, 57, 7),         # generic store path
    ("v", oc.GRAYSCALE, 640, 200, 4),   # luma only: no chroma tables at all
])
def test_host_frames_as_independent_allocations(ctx0, zj, synth, mode, out_cs, w, h, n):
    """zj_decode_frames: the three-stream host pipeline over frames that are separate numpy arrays; same bytes as the
    oracle frame by frame and as the packed batch."""
    ctx = ctx0
    hs, vs = MODES[mode]
    frames = [synth.make_frame(w, h, hs, vs, 3, seed=300, frame_index=i % 3) for i in range(n)]
    qts = frames[0][1]
    d = zj.FrameDesc.make(w, h, hs, vs, 3, out_cs, qts)
    gray = out_cs == oc.GRAYSCALE
    outs = ctx.decode_frames(d, [[np.array(p) for p in (fr[0][:1] if gray else fr[0])] for fr in frames])
    packed = ctx.decode_planes(d, [np.concatenate([fr[0][c] for fr in frames]) for c in range(3)], nframes=n)
    olen = packed.size // n
    exp = {}
    for i in range(n):
        if i % 3 not in exp:
            rc, e = oc.decode_planes(oc.make_frame(w, h, hs, vs, 3, out_cs, qts), frames[i][0])
            assert rc == 0
            exp[i % 3] = e
        assert_same(outs[i], exp[i % 3], f"frame {i}")
        assert_same(packed[i * olen:(i + 1) * olen], exp[i % 3], f"packed frame {i}")


def test_scans_with_mixed_destinations_share_the_pixel_launch(ctx0, zj):
    """zj_decode_scans / zj_decoder_finish_pixels_batch: files of one geometry whose device outputs are NOT equally spaced,
    with a file of another geometry in between -- before round 5 that meant one pixel launch per file; now the scattered
    form carries them in one.  Bytes equal each file's own decode."""
    ctx = ctx0
    sys.path.insert(0, HERE)
    from test_huff_emu import pil_jpeg
    files = [pil_jpeg(640, 480, quality=88, seed=160 + k) for k in range(6)]
    files.insert(2, pil_jpeg(656, 480, quality=88, seed=199))
    want = [zj.Decoder(None, ctx).decode_buffer(f) for f in files]
    decs = []
    for f in files:
        o = zj.ZuneJpegOptions()
        o.entropy = zj.ENTROPY_GPU_ALWAYS
        dd = zj.Decoder(o, ctx)
        dd.prepare(f)
        decs.append(dd)
    sizes = [w.size for w in want]
    ptrs = []
    spacers = []
    for k in (4, 1, 6, 0, 3, 5, 2):  # allocation order != file order
        ptrs.append((k, ctx.device_alloc(sizes[k] + 64)))
        spacers.append(ctx.device_alloc(4096 * (k + 1) + 128))
    byk = dict(ptrs)
    try:
        lens, rcs = zj.finish_pixels_batch(decs, ctx, device_ptrs=[(byk[k], sizes[k]) for k in range(len(files))])
        assert not any(rcs)
        for k, w in enumerate(want):
            got = np.zeros(w.size, np.uint8)
            ctx.d2h(got, byk[k])
            assert lens[k] == w.size
            assert_same(got, w, f"file {k}")
    finally:
        for _, p in ptrs:
            ctx.device_free(p)
        for p in spacers:
            ctx.device_free(p)
        for dd in decs:
            dd.close()


# ---- image-level sharding inside the library: two device slots on the box's one GPU ---------------------------------
def test_multi_planes_sharded_over_two_slots(zj, synth):
    w, h, n = 528, 136, 7
    frames = [synth.make_frame(w, h, 2, 2, 3, seed=610, frame_index=i) for i in range(n)]
    qts = frames[0][1]
    d = zj.FrameDesc.make(w, h, 2, 2, 3, zj.ColorSpace.RGB, qts)
    f = oc.make_frame(w, h, 2, 2, 3, oc.RGB, qts)
    exp = [oc.decode_planes(f, fr[0])[1] for fr in frames]
    m = zj.Multi([0, 0])
    try:
        out = m.decode_planes(d, [np.concatenate([fr[0][c] for fr in frames]) for c in range(3)], n)
        olen = out.size // n
        for i in range(n):
            assert_same(out[i * olen:(i + 1) * olen], exp[i], f"packed frame {i}")
        assert [s[1] for s in m.slot_stats()] == [4, 3] and [s[0] for s in m.slot_stats()] == [0, 0]  # zj_shard_range(7, ., 2)
        outs = m.decode_frames(d, [[np.array(p) for p in fr[0]] for fr in frames])
        for i in range(n):
            assert_same(outs[i], exp[i], f"scattered host frame {i}")
        assert [s[1] for s in m.slot_stats()] == [8, 6]
        # device-resident, scattered: every frame its own allocations
        c0 = zj.Context(zj.BACKEND_HIP, 0)
        df = DeviceFrames(c0, [fr[0] for fr in frames], olen, seed=5)
        try:
            order = list(range(n))
            m.decode_frames_device(d, df.col(0, order), df.col(1, order), df.col(2, order), df.col(3, order))
            for i in range(n):
                assert_same(df.out(i), exp[i], f"device frame {i}")
        finally:
            df.free()
            c0.close()
        one = m.decode_planes(d, frames[0][0], 1)   # fewer frames than slots: the second shard is empty
        assert_same(one, exp[0], "one frame")
        with pytest.raises(zj.ZjError):             # an error in a shard is the call's error
            m.decode_planes(zj.FrameDesc.make(w, h, 4, 1, 3, 0, qts), frames[0][0], 1)
    finally:
        m.close()


@pytest.mark.parametrize("entropy", ["cpu", "gpu"])
def test_multi_device_pool_decodes_a_mixed_batch(zj, synth, entropy):
    """zj_pool_create_multi with devices = {0, 0}: two independent sets of submitters and contexts on one GPU.  Host
    outputs are dealt by readiness; device outputs go to the slots of the device that owns the pointer (rotating over
    them).  Every file equals the oracle on the planes it was encoded from; a broken file reports its own status."""
    sys.path.insert(0, os.path.join(os.path.dirname(HERE), "tools"))
    import jpeg_enc
    qts = synth.quant_tables(88)
    cases = []
    for i, (w, h, hs, vs, kind) in enumerate([(208, 96, 2, 2, "rst"), (64, 64, 1, 1, "prog"), (130, 50, 2, 1, "base"),
                                              (96, 72, 1, 2, "rst"), (320, 200, 2, 2, "prog"), (48, 40, 2, 2, "base"),
                                              (200, 120, 2, 2, "rst"), (640, 480, 2, 2, "base"), (640, 480, 2, 2, "base")]):
        planes = jpeg_enc.small_planes(w, h, hs, vs, 3, seed=20 + i)
        if kind == "prog":
            blob = jpeg_enc.encode_progressive(planes, qts, w, h, hs, vs, 3)
        else:
            blob = jpeg_enc.encode_baseline(planes, qts, w, h, hs, vs, 3, restart=5 if kind == "rst" else 0)
        rc, exp = oc.decode_planes(oc.make_frame(w, h, hs, vs, 3, oc.RGB, qts), planes)
        assert rc == 0
        cases.append((blob, exp))
    for name in ("test-baseline.jpg", "test-progressive.jpg"):
        blob = open(os.path.join(HERE, "golden", name), "rb").read()
        cases.append((blob, zj.Decoder().decode_buffer(blob)))
    bad = bytes(cases[0][0][:40])
    o = zj.ZuneJpegOptions()
    o.num_threads = 2
    if entropy == "gpu":
        o.entropy = zj.ENTROPY_GPU_ALWAYS
    with zj.Pool(threads=2, options=o, devices=[0, 0]) as pool:
        assert pool.threads == 4 and len(pool.device_stats()) == 2
        for _ in range(2):
            outs, infos, sts = pool.decode_files([c[0] for c in cases] * 3 + [bad], raise_on_error=False)
            assert sts[:-1] == [0] * (3 * len(cases)) and sts[-1] != 0
            for i in range(3 * len(cases)):
                assert_same(outs[i], cases[i % len(cases)][1], f"file {i}")
        st = pool.device_stats()
        assert [s[0] for s in st] == [0, 0] and sum(s[2] for s in st) == 2 * 3 * len(cases)
        # pixels left in HBM: routed by the owner of each output pointer, rotating over that device's two slots
        c0 = zj.Context(zj.BACKEND_HIP, 0)
        ptrs = [c0.device_alloc(c[1].size + 64) for c in cases]
        try:
            before = [s[2] for s in pool.device_stats()]
            lens, _, sts = pool.decode_files_device([c[0] for c in cases], ptrs, [c[1].size for c in cases])
            assert all(s == 0 for s in sts) and lens == [c[1].size for c in cases]
            for i, (p, c) in enumerate(zip(ptrs, cases)):
                got = np.zeros(c[1].size, np.uint8)
                c0.d2h(got, p)
                assert_same(got, c[1], f"device file {i}")
            after = [s[2] for s in pool.device_stats()]
            assert [a - b for a, b in zip(after, before)] == [(len(cases) + 1) // 2, len(cases) // 2]
            assert zj.pointer_device(ptrs[0]) == 0
            host = np.zeros(64, np.uint8)
            assert zj.pointer_device(host.ctypes.data) < 0
            with pytest.raises(zj.DecodeError):  # an output that is not device memory of the pool
                pool.decode_files_device([cases[0][0]], [host.ctypes.data], [host.size])
        finally:
            for p in ptrs:
                c0.device_free(p)
            c0.close()
