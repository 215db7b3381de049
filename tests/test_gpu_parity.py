"""GPU parity tests: every call goes through the C ABI of libzjhip.so (include/zjhip.h) and is compared
bit for bit with the oracle on the same seeded inputs.  Mirrors the reference's own unit tests
(src/idct.rs:66-127 KATs, src/upsampler.rs:123-151 ramps) and integration sizes
(tests/large_images.rs, tests/medium_images.rs)."""
import glob
import importlib
import json
import os

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
    c = zj.Context(zj.BACKEND_HIP, 0)  # no GPU -> raises; nothing falls back to the CPU
    c.set_variant(request.param)       # both kernel variants must be bit-exact
    yield c
    c.close()


def assert_same(out, exp, what):
    if not np.array_equal(out, exp):
        bad = np.nonzero(np.asarray(out).reshape(-1) != np.asarray(exp).reshape(-1))[0]
        raise AssertionError(f"{what}: {bad.size} of {exp.size} values differ, first at {bad[:10]}")


# ---- strip level: IDCTPtr ---------------------------------------------------------------------
KAT = json.load(open(os.path.join(HERE, "golden", "idct_kat.json")))


@pytest.mark.parametrize("name", ["zeroes", "max", "min"])
def test_idct_reference_kat(ctx, name):
    """src/idct.rs:66-127: test_zeroes / test_max / test_min"""
    coeff = np.full(64, KAT[name]["coeff"], np.int16)
    out = ctx.idct_strip(coeff, np.ones(64, np.int32), 8, 1, 1)
    assert_same(out, np.array(KAT[name]["expected"], np.int16), name)


def test_idct_golden_blocks(ctx):
    z = np.load(os.path.join(HERE, "golden", "idct_blocks.npz"))
    n = z["blocks"].shape[0]
    out = ctx.idct_strip(z["blocks"].reshape(-1), z["qt"], 8 * n, 1, 1)
    assert_same(out.reshape(8, n, 8).transpose(1, 0, 2), z["expected"], "idct_blocks")


@pytest.mark.parametrize("kind", ["full", "sparse", "dc"])
def test_idct_random_blocks_vs_oracle(ctx, kind):
    rng = np.random.default_rng(17)
    n = 100_000
    if kind == "full":
        b = rng.integers(-32768, 32768, size=(n, 64))
    elif kind == "sparse":
        b = rng.integers(-300, 301, size=(n, 64))
        b[rng.random((n, 64)) < 0.8] = 0
    else:
        b = np.zeros((n, 64), np.int64)
        b[:, 0] = rng.integers(-32768, 32768, size=n)
    b = b.astype(np.int16).reshape(-1)
    qt = rng.integers(1, 256, size=64).astype(np.int32)
    rc, exp = oc.idct_strip(b, qt, 8 * n, 1, 1)
    assert rc == 0
    assert_same(ctx.idct_strip(b, qt, 8 * n, 1, 1), exp, kind)


def test_idct_strip_geometry_and_panics(ctx, zj):
    rng = np.random.default_rng(3)
    qt = rng.integers(1, 64, size=64).astype(np.int32)
    # 4:2:0 luma strip (stride 16*mcu_x, samp 4, v 1) and chroma strip (stride 8*mcu_x, samp 4, v 2)
    for n, stride, samp, vs in ((4 * 2 * 24 * 64, 16 * 24, 4, 1), (2 * 24 * 64, 8 * 24, 4, 2), (40 * 64, 8 * 40 + 5, 1, 1)):
        c = rng.integers(-200, 200, size=n).astype(np.int16)
        rc, exp = oc.idct_strip(c, qt, stride, samp, vs)
        if rc == 0:
            assert_same(ctx.idct_strip(c, qt, stride, samp, vs), exp, (n, stride))
        else:
            with pytest.raises(zj.ZjError) as e:
                ctx.idct_strip(c, qt, stride, samp, vs)
            assert e.value.status == -5
    with pytest.raises(zj.ZjError) as e:  # stride too small: reference panics on get_mut().unwrap()
        ctx.idct_strip(np.zeros(128, np.int16), qt, 4096, 1, 1)
    assert e.value.status == -5
    assert ctx.idct_strip(np.zeros(0, np.int16), qt, 8, 1, 1).size == 0  # empty input


# ---- strip level: UpSampler ---------------------------------------------------------------------
def test_upsample_ramps_like_reference_tests(ctx):
    """src/upsampler.rs:123-151 use ramps 0..128 and (0..1280).rev()"""
    for inp in (np.arange(128), np.arange(1280)[::-1]):
        inp = inp.astype(np.int16)
        rc, exp = oc.upsample_h(inp, 2 * inp.size)
        assert rc == 0
        assert_same(ctx.upsample_horizontal(inp, 2 * inp.size), exp, "ramp")


@pytest.mark.parametrize("n", [8, 24, 16 * 40, 32768, 65536 + 8])
def test_upsamplers_vs_oracle(ctx, n):
    rng = np.random.default_rng(n)
    inp = rng.integers(-3968, 4224, size=n).astype(np.int16)
    for name, fn, ofn, olen in (("h", ctx.upsample_horizontal, oc.upsample_h, 2 * n),
                                ("v", ctx.upsample_vertical, oc.upsample_v, 2 * n),
                                ("hv", ctx.upsample_hv, oc.upsample_hv, 4 * n)):
        rc, exp = ofn(inp, olen)
        assert rc == 0
        assert_same(fn(inp, olen), exp, (name, n))


def test_upsample_panics_and_odd_lengths(ctx, zj):
    with pytest.raises(zj.ZjError) as e:
        ctx.upsample_horizontal(np.zeros(2, np.int16), 16)  # "Too Short of a vector"
    assert e.value.status == -5
    with pytest.raises(zj.ZjError):
        ctx.upsample_vertical(np.zeros(4, np.int16), 8)     # stride 0
    rng = np.random.default_rng(9)
    inp = rng.integers(-100, 100, size=50).astype(np.int16)
    for olen in (100, 90, 60):  # output shorter than 2n: zip stops early, tail still written
        rc, exp = oc.upsample_h(inp, olen)
        assert rc == 0
        assert_same(ctx.upsample_horizontal(inp, olen), exp, olen)


# ---- strip level: ColorConvert16Ptr -------------------------------------------------------------
def test_rgb16_vs_oracle(ctx, zj):
    rng = np.random.default_rng(23)
    for _ in range(20):
        y, cb, cr = (rng.integers(-4100, 4400, size=16).astype(np.int16) for _ in range(3))
        out = np.zeros(112, np.uint8)
        exp = np.zeros(112, np.uint8)
        pos = ctx.ycbcr_to_rgb_16(y, cb, cr, out, 16)
        rc, epos = oc.ycbcr_to_rgb16(y, cb, cr, exp, 16)
        assert rc == 0 and pos == epos == 64
        assert_same(out, exp, "rgb16")
    with pytest.raises(zj.ZjError) as e:
        ctx.ycbcr_to_rgb_16(y, cb, cr, np.zeros(60, np.uint8), 13)  # "Slice to small cannot write"
    assert e.value.status == -5


# ---- post_process (one strip) -------------------------------------------------------------------
@pytest.mark.parametrize("mode", list(MODES))
@pytest.mark.parametrize("out_cs", [oc.RGB, oc.GRAYSCALE, oc.YCBCR])
def test_post_process_strip_vs_oracle(ctx, zj, synth, mode, out_cs):
    hs, vs = MODES[mode]
    rows = 32 if (hs, vs) == (2, 2) else (16 if hs == 2 or vs == 2 else 8)
    w = 320
    mcu_x = w // (8 * hs)
    for adversarial in (False, True):
        mk = synth.make_adversarial_frame if adversarial else synth.make_frame
        planes, qts = mk(w, rows, hs, vs, 3, seed=41)
        ncomp = {oc.RGB: 3, oc.GRAYSCALE: 1, oc.YCBCR: 3}[out_cs]
        exp = np.zeros(rows * w * ncomp, np.uint8)   # callers hand over zeroed chunks (mcu.rs:222)
        assert oc.post_process(planes, oc.make_components(hs, vs, mcu_x, qts), oc.YCBCR, out_cs, exp, w) == 0
        out = np.zeros(rows * w * ncomp, np.uint8)
        comps = (zj.Component * 3)()
        for c in range(3):
            comps[c].horizontal_sample = hs if c == 0 else 1
            comps[c].vertical_sample = vs if c == 0 else 1
            comps[c].width_stride = (hs if c == 0 else 1) * mcu_x * 8
            for k in range(64):
                comps[c].quantization_table[k] = int(qts[c][k])
        ctx.post_process(planes, comps, oc.YCBCR, out_cs, out, w)
        assert_same(out, exp, (mode, out_cs, adversarial))


def test_post_process_leaves_unwritten_bytes_untouched(ctx, zj, synth):
    w, rows, hs, vs = 64, 32, 2, 2
    planes, qts = synth.make_frame(w, rows, hs, vs, 3, seed=2)
    comps = (zj.Component * 3)()
    for c in range(3):
        comps[c].horizontal_sample = hs if c == 0 else 1
        comps[c].vertical_sample = vs if c == 0 else 1
        comps[c].width_stride = (hs if c == 0 else 1) * 4 * 8
        for k in range(64):
            comps[c].quantization_table[k] = int(qts[c][k])
    out = np.full(rows * w * 3, 0x77, np.uint8)
    ctx.post_process(planes, comps, oc.YCBCR, oc.RGB, out, w)
    assert (out.reshape(rows, 3 * w)[:, -16:] == 0x77).all()  # Q5/Q6: never written by the reference


# ---- frame level --------------------------------------------------------------------------------
GOLDEN = sorted(glob.glob(os.path.join(HERE, "golden", "strip_*.npz")) +
                glob.glob(os.path.join(HERE, "golden", "frame_*.npz")) +
                glob.glob(os.path.join(HERE, "golden", "adversarial_*.npz")))


@pytest.mark.parametrize("path", GOLDEN, ids=[os.path.basename(p) for p in GOLDEN])
def test_golden_fixtures(ctx, zj, path):
    z = np.load(path)
    d = zj.FrameDesc.make(int(z["width"]), int(z["height"]), int(z["h_max"]), int(z["v_max"]), 3,
                          int(z["out_cs"]), list(z["qt"]))
    assert_same(ctx.decode_planes(d, [z["y"], z["cb"], z["cr"]]), z["expected"], os.path.basename(path))


GOLDEN_EXT = sorted(glob.glob(os.path.join(HERE, "golden", "ext_*.npz")))


@pytest.mark.parametrize("path", GOLDEN_EXT, ids=[os.path.basename(p) for p in GOLDEN_EXT])
def test_golden_extension_fixtures(ctx, zj, path):
    z = np.load(path)
    d = zj.FrameDesc.make(int(z["width"]), int(z["height"]), int(z["h_max"]), int(z["v_max"]), 3,
                          int(z["out_cs"]), list(z["qt"]), flags=int(z["flags"]), out_layout=int(z["out_layout"]))
    assert_same(ctx.decode_planes(d, [z["y"], z["cb"], z["cr"]]), z["expected"], os.path.basename(path))


@pytest.mark.parametrize("mode", list(MODES))
@pytest.mark.parametrize("out_cs", [oc.RGB, oc.GRAYSCALE, oc.YCBCR])
@pytest.mark.parametrize("wh", [(64, 64), (528, 40), (32, 8), (272, 100), (304, 48), (1040, 33), (2080, 16), (1920, 1080)])
def test_decode_planes_vs_oracle(ctx, zj, synth, mode, out_cs, wh):
    hs, vs = MODES[mode]
    w, h = wh
    for adversarial in (False, True):
        if adversarial and w * h > 600_000:
            continue
        mk = synth.make_adversarial_frame if adversarial else synth.make_frame
        planes, qts = mk(w, h, hs, vs, 3, seed=61)
        rc, exp = oc.decode_planes(oc.make_frame(w, h, hs, vs, 3, out_cs, qts), planes)
        assert rc == 0
        d = zj.FrameDesc.make(w, h, hs, vs, 3, out_cs, qts)
        assert_same(ctx.decode_planes(d, planes), exp, (mode, out_cs, wh, adversarial))


RAGGED = [(100, 32), (37, 50), (17, 16), (1001, 33), (24, 24), (5, 3), (2500, 1786)]


@pytest.mark.parametrize("mode", list(MODES))
@pytest.mark.parametrize("out_cs", [oc.RGB, oc.GRAYSCALE, oc.YCBCR])
@pytest.mark.parametrize("wh", RAGGED)
def test_decode_planes_ragged_widths(ctx, zj, synth, mode, out_cs, wh):
    """Widths that are not multiples of 16 (tests/medium_images.rs uses 2500x1786), width < 16,
    P % 16 == 8; where the reference panics the ABI reports ZJ_ERR_PANIC."""
    hs, vs = MODES[mode]
    w, h = wh
    planes, qts = synth.make_frame(w, h, hs, vs, 3, seed=83)
    rc, exp = oc.decode_planes(oc.make_frame(w, h, hs, vs, 3, out_cs, qts), planes)
    d = zj.FrameDesc.make(w, h, hs, vs, 3, out_cs, qts)
    if rc != 0:
        with pytest.raises(zj.ZjError) as e:
            ctx.decode_planes(d, planes)
        assert e.value.status == -5
        return
    assert_same(ctx.decode_planes(d, planes), exp, (mode, out_cs, wh))


def test_large_image_7680x4320(ctx, zj, synth):
    """tests/large_images.rs / benches: 7680x4320, here 4:2:0 -> RGB"""
    w, h = 7680, 4320
    planes, qts = synth.make_frame(w, h, 2, 2, 3, seed=5)
    rc, exp = oc.decode_planes(oc.make_frame(w, h, 2, 2, 3, oc.RGB, qts), planes)
    assert rc == 0
    d = zj.FrameDesc.make(w, h, 2, 2, 3, zj.ColorSpace.RGB, qts)
    assert_same(ctx.decode_planes(d, planes), exp, "7680x4320")


def test_decode_grayscale_jpeg(ctx, zj, synth):
    """1-component input -> GRAYSCALE (benches/decode_grayscale.rs path)"""
    planes, qts = synth.make_frame(640, 200, 1, 1, 1, seed=4)
    rc, exp = oc.decode_planes(oc.make_frame(640, 200, 1, 1, 1, oc.GRAYSCALE, qts), planes)
    assert rc == 0
    d = zj.FrameDesc.make(640, 200, 1, 1, 1, zj.ColorSpace.GRAYSCALE, qts)
    assert_same(ctx.decode_planes(d, planes), exp, "gray-in")


def test_decode_batch_and_device_api(ctx, zj, synth):
    """Batch of frames through the device-pointer API used by bench.py (different tables per call)."""
    import ctypes as C
    w, h, n = 256, 64, 5
    frames = [synth.make_frame(w, h, 2, 2, 3, seed=70, frame_index=i) for i in range(n)]
    qts = frames[0][1]
    planes = [np.concatenate([f[0][c] for f in frames]) for c in range(3)]
    d = zj.FrameDesc.make(w, h, 2, 2, 3, zj.ColorSpace.RGB, qts)
    out_len = zj.lib().zj_out_len(C.byref(d))
    bufs = [ctx.device_alloc(p.nbytes) for p in planes] + [ctx.device_alloc(n * out_len)]
    try:
        for p, b in zip(planes, bufs):
            ctx.h2d(b, p)
        ctx.decode_planes_device(d, n, bufs[0], bufs[1], bufs[2], bufs[3])
        out = np.empty(n * out_len, np.uint8)
        ctx.d2h(out, bufs[3])
        for i, f in enumerate(frames):
            rc, exp = oc.decode_planes(oc.make_frame(w, h, 2, 2, 3, oc.RGB, qts), f[0])
            assert rc == 0
            assert_same(out[i * out_len:(i + 1) * out_len], exp, f"frame {i}")
        ms, each, name = ctx.time_decode_device(d, n, bufs[0], bufs[1], bufs[2], bufs[3], 3)
        assert ms > 0 and each > 0 and "zj_fused" in name and "<2, 2, 0" in name
    finally:
        for b in bufs:
            ctx.device_free(b)


@pytest.mark.parametrize("mode,out_cs,w,h,n", [
    ("hv", oc.RGB, 256, 72, 11),        # odd MCU row: the last 8 rows of every frame stay 0 (Q6); grouped frames
    ("hv", oc.RGB, 4096, 2200, 2),      # 25 MB per frame: each frame is cut into strip ranges, last one ragged
    ("none", oc.YCBCR, 1920, 1080, 3),  # 12 MB per frame, whole frames
    ("h", oc.RGB, 1001, 57, 7),         # generic store path, H mode drops the odd MCU row
    ("v", oc.GRAYSCALE, 2048, 3000, 1), # one large luma-only frame, split
])
def test_host_pipeline_units(ctx, zj, synth, mode, out_cs, w, h, n):
    """zj_decode_planes_batch cuts the batch into units (frame groups or strip ranges) that overlap on three
    streams: same bytes as the oracle frame by frame, and as the unpipelined single-unit path."""
    hs, vs = MODES[mode]
    frames = [synth.make_frame(w, h, hs, vs, 3, seed=300, frame_index=i % 3) for i in range(n)]
    qts = frames[0][1]
    planes = [np.concatenate([f[0][c] for f in frames]) for c in range(3)]
    d = zj.FrameDesc.make(w, h, hs, vs, 3, out_cs, qts)
    out = ctx.decode_planes(d, planes, nframes=n)
    ctx.set_pipeline(0)
    try:
        single = ctx.decode_planes(d, planes, nframes=n)
    finally:
        ctx.set_pipeline(1)
    assert_same(out, single, "pipelined vs single unit")
    olen = out.size // n
    exp = {}
    for i in range(n):
        if i % 3 not in exp:
            rc, e = oc.decode_planes(oc.make_frame(w, h, hs, vs, 3, out_cs, qts), frames[i][0])
            assert rc == 0
            exp[i % 3] = e
        assert_same(out[i * olen:(i + 1) * olen], exp[i % 3], f"frame {i}")


def test_unsupported_and_invalid_arguments(ctx, zj, synth):
    planes, qts = synth.make_frame(64, 64, 2, 2, 3, seed=1)
    with pytest.raises(zj.ZjError) as e:  # 16-bit quantisation tables are rejected (headers.rs:154-174)
        ctx.decode_planes(zj.FrameDesc.make(64, 64, 2, 2, 3, 0, [np.full(64, 300, np.int32)] * 3), planes)
    assert e.value.status == -2
    with pytest.raises(zj.ZjError) as e:  # sampling factor 4 is not a mode the reference knows
        ctx.decode_planes(zj.FrameDesc.make(64, 64, 4, 1, 3, 0, qts), planes)
    assert e.value.status == -1
    with pytest.raises(zj.ZjError):       # CMYK output is a no-op in the reference
        ctx.decode_planes(zj.FrameDesc.make(64, 64, 2, 2, 3, int(zj.ColorSpace.CMYK), qts), planes)
    with pytest.raises(zj.ZjError):       # unknown flag bits / layouts are argument errors
        ctx.decode_planes(zj.FrameDesc.make(64, 64, 2, 2, 3, 0, qts, flags=8), planes)
    with pytest.raises(zj.ZjError):
        ctx.decode_planes(zj.FrameDesc.make(64, 64, 2, 2, 3, 0, qts, out_layout=7), planes)


# ---- BASELINE.json full size: 4096x4096, properties that do not need the (slow) oracle everywhere ----
def _crc_rows(a, w3):
    return np.add.reduce(a.reshape(-1, w3).astype(np.uint64) * (np.arange(w3, dtype=np.uint64) % 251 + 1), axis=1)


@pytest.mark.parametrize("mode,out_cs", [("hv", oc.RGB), ("none", oc.RGB), ("none", oc.GRAYSCALE)])
def test_full_size_4096(ctx, zj, synth, mode, out_cs):
    """configs[1] / configs[2]: 4096x4096.  (a) strips are independent: decoding the frame equals
    decoding each strip as its own frame (checked on a sample of strips against the oracle);
    (b) the Q5/Q6 byte pattern holds on every row; (c) decoding twice is idempotent."""
    hs, vs = MODES[mode]
    w = h = 4096
    planes, qts = synth.make_frame(w, h, hs, vs, 3, seed=1234)
    d = zj.FrameDesc.make(w, h, hs, vs, 3, out_cs, qts)
    out = ctx.decode_planes(d, planes)
    ncomp = 3 if out_cs == oc.RGB else 1
    rows = out.reshape(h, w * ncomp)
    if out_cs == oc.RGB:
        assert not rows[:, -16:].any()
        assert rows[:, :-16].any(axis=1).all()
    strip_rows = 32 if mode == "hv" else 8
    ybr = strip_rows // 8
    ypl = planes[0].reshape(h // 8, -1)
    cpl = [p.reshape(h // (8 * vs), -1) for p in planes[1:]]
    cbr = 2 if mode == "hv" else 1
    for s in (0, 1, h // strip_rows // 2, h // strip_rows - 1):
        sp = [ypl[s * ybr:(s + 1) * ybr].reshape(-1)] + [c[s * cbr:(s + 1) * cbr].reshape(-1) for c in cpl]
        rc, exp = oc.decode_planes(oc.make_frame(w, strip_rows, hs, vs, 3, out_cs, qts), sp)
        assert rc == 0
        assert_same(rows[s * strip_rows:(s + 1) * strip_rows].reshape(-1), exp, (mode, "strip", s))
    out2 = ctx.decode_planes(d, planes)
    assert np.array_equal(_crc_rows(out, w * ncomp), _crc_rows(out2, w * ncomp))


@pytest.mark.parametrize("mode,out_cs", [("hv", oc.RGB), ("none", oc.RGB), ("none", oc.GRAYSCALE), ("h", oc.RGB), ("v", oc.RGB)])
def test_full_size_4096_whole_frame_vs_oracle(ctx, zj, synth, mode, out_cs):
    """Complete 4096x4096 frames against the oracle, every byte: configs[1] (4:2:0 -> RGB), configs[2] (4:4:4 -> RGB and
    -> GRAYSCALE), and the reference's other two sampling modes (the oracle needs a few seconds per frame)."""
    hs, vs = MODES[mode]
    w = h = 4096
    planes, qts = synth.make_frame(w, h, hs, vs, 3, seed=99)
    rc, exp = oc.decode_planes(oc.make_frame(w, h, hs, vs, 3, out_cs, qts), planes)
    assert rc == 0
    d = zj.FrameDesc.make(w, h, hs, vs, 3, out_cs, qts)
    assert_same(ctx.decode_planes(d, planes), exp, ("4096", mode, out_cs))


def test_integer_generator_on_the_gpu_equals_the_cpu(zj, synth):
    """bench.py generates its frames on the GPU (synth.make_frame_t): the same integers as on the CPU, and the luma plane
    of frame 7 carries the checksum recorded beside the oracle's output checksums."""
    import json
    import torch
    g = json.load(open(os.path.join(HERE, "golden", "checksums_seed1234.json")))
    pg, _ = synth.make_frame_t(4096, 4096, 2, 2, 3, seed=g["seed"], frame_index=7, device="cuda:0")
    assert f"{synth.frame_checksum_sum(pg[0].cpu().numpy().view('u1')):016x}" == g["y_plane"][7]
    pc, _ = synth.make_frame_t(1024, 512, 2, 2, 3, seed=5, frame_index=2, device="cpu")
    pg, _ = synth.make_frame_t(1024, 512, 2, 2, 3, seed=5, frame_index=2, device="cuda:0")
    for a, b in zip(pc, pg):
        assert torch.equal(a, b.cpu())


# ---- whole decoder: CPU entropy front-end -> GPU pixel path (BASELINE.json configs[0] and [3]) ----------
@pytest.mark.parametrize("name", ["test-baseline.jpg", "test-progressive.jpg"])
@pytest.mark.parametrize("out_cs", [oc.RGB, oc.GRAYSCALE, oc.YCBCR])
def test_decode_buffer_reference_images(ctx, zj, name, out_cs):
    """Decoder::decode_buffer on the reference's own 1920x1080 test images: Huffman on the CPU, pixels on
    the GPU; must equal the oracle pixel path run on the same coefficient planes, byte for byte."""
    data = open(os.path.join(HERE, "golden", name), "rb").read()
    o = zj.ZuneJpegOptions()
    o.out_colorspace = zj.ColorSpace(out_cs)
    dec = zj.Decoder(o, ctx)
    out = dec.decode_buffer(data)
    desc, planes, info = dec.decode_coefficients(data)
    assert (info.width, info.height) == (1920, 1080)
    rc, exp = oc.decode_planes(oc.make_frame(1920, 1080, 1, 1, 3, out_cs, list(np.ctypeslib.as_array(desc.qt))), planes)
    assert rc == 0
    assert_same(out, exp, (name, out_cs))


def test_decode_buffer_synthetic_progressive_420(ctx, zj, synth):
    """A progressive 4:2:0 stream (the reference ships none) built by tools/jpeg_enc.py"""
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(HERE), "tools"))
    import jpeg_enc
    w, h = 208, 96
    planes = jpeg_enc.small_planes(w, h, 2, 2, 3, seed=11)
    qts = synth.quant_tables(90)
    for enc in (jpeg_enc.encode_baseline, jpeg_enc.encode_progressive):
        out = zj.Decoder(None, ctx).decode_buffer(enc(planes, qts, w, h, 2, 2, 3))
        rc, exp = oc.decode_planes(oc.make_frame(w, h, 2, 2, 3, oc.RGB, qts), planes)
        assert rc == 0
        assert_same(out, exp, enc.__name__)


@pytest.mark.parametrize("threads", [1, 3])
def test_pool_decodes_a_mixed_batch(zj, synth, threads):
    """zj_pool_decode_files: several workers, each entropy decoder + GPU context; every file equals the oracle on the
    planes it was encoded from, a broken file reports its own status without disturbing the others."""
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(HERE), "tools"))
    import jpeg_enc
    qts = synth.quant_tables(88)
    cases = []
    for i, (w, h, hs, vs, kind) in enumerate([(208, 96, 2, 2, "rst"), (64, 64, 1, 1, "prog"), (130, 50, 2, 1, "base"),
                                              (96, 72, 1, 2, "rst"), (320, 200, 2, 2, "prog"), (48, 40, 2, 2, "base"),
                                              (200, 120, 2, 2, "rst")]):
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
    with zj.Pool(threads=threads, options=o) as pool:
        assert pool.threads == threads
        for _ in range(2):  # the pool is reused across batches
            outs, infos, sts = pool.decode_files([c[0] for c in cases] + [bad], raise_on_error=False)
            assert sts[:-1] == [0] * len(cases) and sts[-1] != 0
            for i, (blob, exp) in enumerate(cases):
                assert_same(outs[i], exp, f"file {i}")
        with pytest.raises(zj.DecodeError):
            pool.decode_files([bad])


def _plain_expected(w, h, hs, vs, qts, planes, kind):
    if kind == "rgba":
        return oc.decode_planes(oc.make_frame(w, h, hs, vs, 3, oc.RGBA, qts), planes, plain=True)
    rc, rgb = oc.decode_planes(oc.make_frame(w, h, hs, vs, 3, oc.RGB, qts), planes, plain=True)
    if kind == "chw":
        rgb = np.ascontiguousarray(rgb.reshape(h, w, 3).transpose(2, 0, 1)).reshape(-1)
    return rc, rgb


@pytest.mark.parametrize("mode", list(MODES))
@pytest.mark.parametrize("kind", ["plain", "rgba", "chw"])
@pytest.mark.parametrize("wh", [(64, 64), (528, 40), (1040, 33), (100, 32), (37, 50), (5, 3), (250, 72), (1920, 1080)])
def test_extensions_plain_rgba_chw(ctx, zj, synth, mode, kind, wh):
    """Beyond the reference: ZJ_FLAG_PLAIN_TAIL, ZJ_CS_RGBA, ZJ_LAYOUT_CHW against the oracle's
    plain-placement restatement; same strips / filters / arithmetic, every pixel at its own position."""
    hs, vs = MODES[mode]
    w, h = wh
    for adversarial in (False, True):
        if adversarial and w * h > 600_000:
            continue
        mk = synth.make_adversarial_frame if adversarial else synth.make_frame
        planes, qts = mk(w, h, hs, vs, 3, seed=77)
        rc, exp = _plain_expected(w, h, hs, vs, qts, planes, kind)
        assert rc == 0
        d = zj.FrameDesc.make(w, h, hs, vs, 3, zj.ColorSpace.RGBA if kind == "rgba" else zj.ColorSpace.RGB, qts,
                              flags=zj.FLAG_PLAIN_TAIL if kind == "plain" else 0,
                              out_layout=zj.LAYOUT_CHW if kind == "chw" else zj.LAYOUT_HWC)
        assert_same(ctx.decode_planes(d, planes), exp, (mode, kind, wh, adversarial))


@pytest.mark.parametrize("kind", ["rgba", "chw", "plain"])
def test_extensions_batches_through_the_host_pipeline(ctx, zj, synth, kind):
    """frame groups, an odd MCU row (rows that stay 0 in every plane), and a frame large enough to be split"""
    for (w, h, n) in [(256, 72, 7), (4096, 2200, 2)]:
        frames = [synth.make_frame(w, h, 2, 2, 3, seed=410, frame_index=i % 2) for i in range(n)]
        qts = frames[0][1]
        planes = [np.concatenate([f[0][c] for f in frames]) for c in range(3)]
        d = zj.FrameDesc.make(w, h, 2, 2, 3, zj.ColorSpace.RGBA if kind == "rgba" else zj.ColorSpace.RGB, qts,
                              flags=zj.FLAG_PLAIN_TAIL if kind == "plain" else 0,
                              out_layout=zj.LAYOUT_CHW if kind == "chw" else zj.LAYOUT_HWC)
        out = ctx.decode_planes(d, planes, nframes=n)
        olen = out.size // n
        for i in range(min(n, 3)):
            rc, exp = _plain_expected(w, h, 2, 2, qts, frames[i][0], kind)
            assert rc == 0
            assert_same(out[i * olen:(i + 1) * olen], exp, (kind, w, h, i))


def test_decoder_corrected_mode_and_planar_options(ctx, zj, synth):
    """zj_options.flags / out_layout reach the pixel path: Decoder with ZJ_FLAG_CORRECTED and with planar output"""
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(HERE), "tools"))
    import jpeg_enc
    w, h = 208, 96
    planes = jpeg_enc.small_planes(w, h, 2, 2, 3, seed=13)
    qts = synth.quant_tables(90)
    blob = jpeg_enc.encode_baseline(planes, qts, w, h, 2, 2, 3)
    f = oc.make_frame(w, h, 2, 2, 3, oc.RGB, qts)
    o = zj.ZuneJpegOptions()
    o.flags = zj.FLAG_CORRECTED
    rc, exp = oc.decode_planes(f, planes, ext=7)
    assert rc == 0
    assert_same(zj.Decoder(o, ctx).decode_buffer(blob), exp, "decoder corrected")
    o = zj.ZuneJpegOptions()
    o.out_layout = zj.LAYOUT_CHW
    rc, exp = oc.decode_planes(f, planes, plain=True)
    assert rc == 0
    assert_same(zj.Decoder(o, ctx).decode_buffer(blob), np.ascontiguousarray(exp.reshape(h, w, 3).transpose(2, 0, 1)).reshape(-1),
                "decoder chw")


def test_decoder_rgba_option(ctx, zj, synth):
    """Decoder with out_colorspace RGBA: the reference's own RGBA arm is malformed; here R G B 255 per pixel"""
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(HERE), "tools"))
    import jpeg_enc
    w, h = 208, 96
    planes = jpeg_enc.small_planes(w, h, 2, 2, 3, seed=12)
    qts = synth.quant_tables(90)
    o = zj.ZuneJpegOptions()
    o.out_colorspace = zj.ColorSpace.RGBA
    out = zj.Decoder(o, ctx).decode_buffer(jpeg_enc.encode_baseline(planes, qts, w, h, 2, 2, 3))
    rc, exp = oc.decode_planes(oc.make_frame(w, h, 2, 2, 3, oc.RGBA, qts), planes, plain=True)
    assert rc == 0
    assert_same(out, exp, "decoder rgba")


def test_random_geometry_sweep_all_output_kinds(ctx, zj, synth):
    """300 random (width, height, mode, output kind) cases through the C ABI, against the oracle; reference panics
    must come back as ZJ_ERR_PANIC."""
    rng = np.random.default_rng(20261002)
    kinds = ["rgb", "gray", "ycbcr", "plain", "rgba", "chw"]
    done = 0
    for case in range(300):
        mode = list(MODES)[int(rng.integers(4))]
        hs, vs = MODES[mode]
        w = int(rng.integers(1, 900)) if case % 3 else int(rng.integers(1, 60)) * 16
        h = int(rng.integers(1, 120))
        kind = kinds[int(rng.integers(len(kinds)))]
        planes, qts = synth.make_frame(w, h, hs, vs, 3, seed=3000 + case)
        cs = {"rgb": zj.ColorSpace.RGB, "gray": zj.ColorSpace.GRAYSCALE, "ycbcr": zj.ColorSpace.YCbCr,
              "plain": zj.ColorSpace.RGB, "rgba": zj.ColorSpace.RGBA, "chw": zj.ColorSpace.RGB}[kind]
        d = zj.FrameDesc.make(w, h, hs, vs, 3, cs, qts, flags=zj.FLAG_PLAIN_TAIL if kind == "plain" else 0,
                              out_layout=zj.LAYOUT_CHW if kind == "chw" else zj.LAYOUT_HWC)
        if kind in ("rgb", "gray", "ycbcr"):
            rc, exp = oc.decode_planes(oc.make_frame(w, h, hs, vs, 3, int(cs), qts), planes)
            if rc != 0:
                with pytest.raises(zj.ZjError) as e:
                    ctx.decode_planes(d, planes)
                assert e.value.status == -5, (case, w, h, mode, kind)
                continue
        else:
            rc, exp = _plain_expected(w, h, hs, vs, qts, planes, kind)
            assert rc == 0
        assert_same(ctx.decode_planes(d, planes), exp, (case, w, h, mode, kind))
        done += 1
    assert done > 220


def test_default_ctx_is_per_thread_and_safe_under_concurrency(zj, synth):
    """The fn-pointer shims use zj_default_ctx() and the reference calls them from four worker threads at once
    (src/mcu.rs:356, fresh threads per decode): every thread gets its own context, contexts of finished threads are
    reused, and concurrent strip calls return the oracle's values."""
    import ctypes as C
    import threading
    L = zj.lib()
    rng = np.random.default_rng(5)
    qt = rng.integers(1, 256, 64).astype(np.int32)
    jobs = []
    for i in range(8):
        coeff = synth.make_frame(512, 8, 1, 1, 3, seed=900 + i)[0][0].astype(np.int16)  # one block row, 64 blocks
        rc, exp = oc.idct_strip(coeff, qt, 512, 1, 1)
        assert rc == 0
        jobs.append((coeff, exp))
    L.zj_idct_strip.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_size_t, C.c_size_t, C.c_void_p]
    seen, errors = [], []

    def worker(k):
        ctx = L.zj_default_ctx()
        seen.append(ctx)
        for rep in range(20):
            coeff, exp = jobs[(k + rep) % len(jobs)]
            out = np.empty_like(coeff)
            rc = L.zj_idct_strip(ctx, coeff.ctypes.data, coeff.size, qt.ctypes.data, 512, 1, 1, out.ctypes.data)
            if rc != 0 or not np.array_equal(out, exp):
                errors.append((k, rep, rc))

    for round_ in range(2):  # second round: new threads pick up the recycled contexts
        ts = [threading.Thread(target=worker, args=(k,)) for k in range(4)]
        [t.start() for t in ts]
        [t.join() for t in ts]
    assert not errors, errors[:4]
    assert len(set(seen[:4])) == 4            # four concurrent threads, four contexts
    assert set(seen[4:]) <= set(seen[:4])     # recycled, not rebuilt


def test_decode_to_tensor_for_torch_consumers(zj, synth):
    """device tensors in, [N, 3, H, W] / [N, H, W, 3] uint8 tensors out on torch's current stream"""
    import torch
    w, h, n = 320, 64, 3
    frames = [synth.make_frame(w, h, 2, 2, 3, seed=555, frame_index=i) for i in range(n)]
    qts = frames[0][1]
    dev = torch.device("cuda:0")
    y, cb, cr = [torch.from_numpy(np.concatenate([f[0][c] for f in frames])).to(dev) for c in range(3)]
    ctx = zj.Context()
    try:
        for layout in (zj.LAYOUT_CHW, zj.LAYOUT_HWC):
            d = zj.FrameDesc.make(w, h, 2, 2, 3, zj.ColorSpace.RGB, qts, out_layout=layout,
                                  flags=zj.FLAG_PLAIN_TAIL if layout == zj.LAYOUT_HWC else 0)
            with torch.cuda.stream(torch.cuda.Stream()):
                t = ctx.decode_to_tensor(d, y, cb, cr)
                torch.cuda.current_stream().synchronize()
            assert tuple(t.shape) == ((n, 3, h, w) if layout == zj.LAYOUT_CHW else (n, h, w, 3)) and t.dtype == torch.uint8
            for i, f in enumerate(frames):
                rc, rgb = oc.decode_planes(oc.make_frame(w, h, 2, 2, 3, oc.RGB, qts), f[0], plain=True)
                assert rc == 0
                exp = rgb.reshape(h, w, 3)
                got = t[i].cpu().numpy()
                if layout == zj.LAYOUT_CHW:
                    got = got.transpose(1, 2, 0)
                assert np.array_equal(got, exp), (layout, i)
    finally:
        ctx.close()



def test_single_frame_decodes_replay_from_a_hip_graph(zj, synth):
    """zj_decode_planes_device is a pure kernel launch (tables by value, no staging, no synchronisation): a run of
    single-frame decodes over two branches can be captured into a HIP graph and replayed (tools/single_frame_graph.py)"""
    import torch
    w, h, nf = 512, 96, 6
    frames = [synth.make_frame(w, h, 2, 2, 3, seed=77, frame_index=i) for i in range(nf)]
    qts = frames[0][1]
    desc = zj.FrameDesc.make(w, h, 2, 2, 3, zj.ColorSpace.RGB, qts)
    dev = torch.device("cuda:0")
    d = [torch.from_numpy(np.concatenate([f[0][c] for f in frames])).to(dev) for c in range(3)]
    out = torch.zeros(nf * w * h * 3, dtype=torch.uint8, device=dev)
    yl, cl, ol = frames[0][0][0].size * 2, frames[0][0][1].size * 2, w * h * 3
    ctx = zj.Context()
    try:
        streams = [torch.cuda.Stream(), torch.cuda.Stream()]
        cap = torch.cuda.Stream()
        g = torch.cuda.CUDAGraph()
        torch.cuda.synchronize()
        with torch.cuda.stream(cap):
            g.capture_begin()
            fork = torch.cuda.Event()
            fork.record(cap)
            for s in streams:
                s.wait_event(fork)
            for f in range(nf):
                ctx.decode_planes_device(desc, 1, d[0].data_ptr() + f * yl, d[1].data_ptr() + f * cl, d[2].data_ptr() + f * cl,
                                         out.data_ptr() + f * ol, streams[f % 2].cuda_stream)
            for s in streams:
                e = torch.cuda.Event()
                e.record(s)
                cap.wait_event(e)
            g.capture_end()
        for _ in range(2):
            out.zero_()
            g.replay()
            torch.cuda.synchronize()
            got = out.cpu().numpy()
            for i, f in enumerate(frames):
                rc, exp = oc.decode_planes(oc.make_frame(w, h, 2, 2, 3, oc.RGB, qts), f[0])
                assert rc == 0
                assert np.array_equal(got[i * ol:(i + 1) * ol], exp), i
    finally:
        ctx.close()


@pytest.mark.parametrize("mode", list(MODES))
@pytest.mark.parametrize("flags,out_cs,layout", [(7, oc.RGB, 0), (2, oc.GRAYSCALE, 0), (4, oc.YCBCR, 0), (6, oc.RGBA, 0), (6, oc.RGB, 1)])
@pytest.mark.parametrize("wh", [(528, 40), (1040, 33), (100, 32), (250, 72), (1920, 1080)])
def test_corrected_mode_flags(ctx, zj, synth, mode, flags, out_cs, layout, wh):
    """ZJ_FLAG_CLAMP_DC / ZJ_FLAG_EDGE_REPLICATE / ZJ_FLAG_CORRECTED through the C ABI against zjo_decode_planes_ext"""
    hs, vs = MODES[mode]
    w, h = wh
    for adversarial in (False, True):
        if adversarial and w * h > 600_000:
            continue
        mk = synth.make_adversarial_frame if adversarial else synth.make_frame
        planes, qts = mk(w, h, hs, vs, 3, seed=91)
        ext = flags | (oc.EXT_PLAIN if (out_cs == oc.RGBA or layout == 1) else 0)
        rc, exp = oc.decode_planes(oc.make_frame(w, h, hs, vs, 3, out_cs, qts), planes, ext=ext)
        d = zj.FrameDesc.make(w, h, hs, vs, 3, out_cs, qts, flags=flags, out_layout=layout)
        if rc != 0:
            with pytest.raises(zj.ZjError) as e:
                ctx.decode_planes(d, planes)
            assert e.value.status == -5
            continue
        if layout == 1:
            exp = np.ascontiguousarray(exp.reshape(h, w, 3).transpose(2, 0, 1)).reshape(-1)
        assert_same(ctx.decode_planes(d, planes), exp, (mode, flags, out_cs, layout, wh, adversarial))


@pytest.mark.parametrize("out_cs", [oc.RGB, oc.GRAYSCALE, oc.YCBCR])
def test_reference_medium_image_2500x1786_h_sampled(ctx, zj, out_cs):
    """The reference's medium image (tests/medium_images.rs: 2500x1786, (2,1) sampling; a copy travels under tests/golden/ref):
    the CPU walker's coefficient planes through the pixel kernels of every variant, every byte against the oracle over the
    same planes -- a ragged width (2500 = 156 * 16 + 4) on real image content."""
    data = open(os.path.join(HERE, "golden", "ref", "medium_horiz_samp_2500x1786.jpg"), "rb").read()
    o = zj.ZuneJpegOptions()
    o.num_threads = 1
    desc, planes, info = zj.Decoder(o).decode_coefficients(data)
    assert (info.width, info.height, info.h_max, info.v_max) == (2500, 1786, 2, 1)
    qts = list(np.ctypeslib.as_array(desc.qt))
    rc, exp = oc.decode_planes(oc.make_frame(2500, 1786, 2, 1, 3, out_cs, qts), planes)
    assert rc == 0
    d = zj.FrameDesc.make(2500, 1786, 2, 1, 3, out_cs, qts)
    assert_same(ctx.decode_planes(d, planes), exp, ("medium_horiz_samp_2500x1786.jpg", out_cs))
