"""GPU: the entropy stage on the device (zune-jpeg_amd/csrc/zj_huff.hip) through the C ABI -- zj_decoder_prepare /
zj_decoder_finish_pixels / zj_decode_scan / zj_pool with zj_options.entropy -- against the product's CPU walker on the
same files (pixels must be identical; tests/test_jpeg_frontend.py pins that walker to the reference), plus the golden
reference file with its early exit at EOI, damaged scans (handed back, then equal to the CPU path) and pixels that stay
in HBM."""
import importlib
import os

import numpy as np
import pytest

from test_huff_emu import document_like, pil_jpeg

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def zj():
    return importlib.import_module("zune-jpeg_amd")


@pytest.fixture(scope="module")
def ctx(zj):
    c = zj.Context(zj.BACKEND_HIP, 0)
    yield c
    c.close()


def decoders(zj, ctx, cs=None, sub=None):
    old = os.environ.get("ZJ_HUFF_SUB")
    if sub:
        os.environ["ZJ_HUFF_SUB"] = str(sub)
    try:
        og, oc_ = zj.ZuneJpegOptions(), zj.ZuneJpegOptions()
        og.entropy = zj.ENTROPY_GPU_ALWAYS
        if cs is not None:
            og.out_colorspace = oc_.out_colorspace = cs
        return zj.Decoder(og, ctx), zj.Decoder(oc_, ctx)
    finally:
        if sub:
            if old is None:
                del os.environ["ZJ_HUFF_SUB"]
            else:
                os.environ["ZJ_HUFF_SUB"] = old


CASES = {
    "420": dict(quality=90), "444": dict(quality=90, subsampling=0), "422": dict(quality=90, subsampling=1),
    "gray": dict(quality=50, gray=True), "420-opt": dict(quality=75, optimize=True),
    "420-ri7": dict(quality=90, restart_marker_blocks=7), "420-ri1": dict(quality=90, restart_marker_blocks=1),
    "420-q20": dict(quality=20), "420-rows": dict(quality=90, restart_marker_rows=1),
    "444-flat": dict(quality=30, subsampling=0, flat=True), "420-q98": dict(quality=98),
}


@pytest.mark.parametrize("sub", [32, 64, 128])
@pytest.mark.parametrize("case", list(CASES))
@pytest.mark.parametrize("wh", [(333, 211), (1601, 1203)])
def test_files_decode_to_the_same_pixels(zj, ctx, case, wh, sub):
    data = pil_jpeg(wh[0], wh[1], seed=len(case) + sub, **CASES[case])
    g, c = decoders(zj, ctx, sub=sub)
    got = g.decode_buffer(data)
    assert g.gpu_status() == 0, [v for k, v in zj.HUFF_ST.items() if g.gpu_status() & k]
    assert np.array_equal(got, c.decode_buffer(data))


@pytest.mark.parametrize("cs", ["RGB", "GRAYSCALE", "YCbCr", "RGBA"])
def test_output_colorspaces(zj, ctx, cs):
    data = pil_jpeg(640, 427, quality=88, seed=3)
    g, c = decoders(zj, ctx, cs=getattr(zj.ColorSpace, cs))
    assert np.array_equal(g.decode_buffer(data), c.decode_buffer(data))
    assert g.gpu_status() == 0


@pytest.mark.parametrize("mode", ["none", "h", "v", "hv"])
@pytest.mark.parametrize("restart", [0, 5])
def test_planes_on_the_device_equal_the_encoder_s_coefficients(zj, ctx, synth, mode, restart):
    """Ground truth that owes nothing to the CPU walker: tools/jpeg_enc.py writes exactly the planes it is given."""
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import jpeg_enc
    hs, vs = {"none": (1, 1), "h": (2, 1), "v": (1, 2), "hv": (2, 2)}[mode]
    w, h = 120, 88
    planes = jpeg_enc.small_planes(w, h, hs, vs, 3, seed=w + hs + vs)
    data = jpeg_enc.encode_baseline(planes, synth.quant_tables(85), w, h, hs, vs, 3, restart=restart)
    g, _ = decoders(zj, ctx, sub=64)
    desc, _ = g.prepare(data)
    out, rc, st = ctx.decode_scan(desc, g.scan_blob())
    assert rc == 0 and st == 0
    for c, p in enumerate(ctx.scan_planes()):
        assert np.array_equal(p, planes[c]), (mode, restart, c)


@pytest.mark.parametrize("name", ["hv", "hv_rst", "none", "h_rst", "gray"])
def test_entropy_golden_fixtures(zj, ctx, name):
    """tests/golden/entropy_*.npz (tools/make_golden.py): the planes the device leaves in HBM are the encoder's."""
    z = np.load(os.path.join(ROOT, "tests", "golden", f"entropy_{name}.npz"))
    g, _ = decoders(zj, ctx, sub=32)
    desc, _ = g.prepare(z["jpeg"].tobytes())
    out, rc, st = ctx.decode_scan(desc, g.scan_blob())
    assert rc == 0 and st == 0
    for p, c in zip(ctx.scan_planes(), ("y", "cb", "cr")):
        assert np.array_equal(p, z[c]), (name, c)


def test_planes_on_the_device_equal_the_cpu_walker(zj, ctx):
    data = pil_jpeg(1024, 768, quality=93, seed=8)
    g, c = decoders(zj, ctx)
    desc, info = g.prepare(data)
    blob = g.scan_blob()
    assert blob is not None
    out, rc, st = ctx.decode_scan(desc, blob)
    assert rc == 0 and st == 0
    assert np.array_equal(out, c.decode_buffer(data))
    _, want, _ = zj.Decoder().decode_coefficients(data)
    for a, b in zip(ctx.scan_planes(), want):
        assert np.array_equal(a, b)
    rounds, _ = ctx.scan_stats()
    assert 1 <= rounds <= 32


def test_more_rounds_than_planned(zj, ctx, monkeypatch):
    """The rounds are launched ahead and looked at once, after the pixels: with too few planned (here 1) the stage adds
    rounds, clears what the premature write pass scattered and runs the rest again."""
    data = pil_jpeg(1201, 801, quality=95, seed=13)
    g, c = decoders(zj, ctx)
    want = c.decode_buffer(data)
    monkeypatch.setenv("ZJ_HUFF_ROUNDS", "1")
    got = g.decode_buffer(data)
    assert g.gpu_status() == 0
    assert np.array_equal(got, want)
    assert ctx.scan_stats()[0] > 1


@pytest.mark.parametrize("kind", ["page", "page-gray", "mixed"])
@pytest.mark.parametrize("mode", ["GPU", "GPU_ALWAYS"])
def test_flat_areas_whatever_path_they_take(zj, ctx, kind, mode):
    """Flat areas converge one sub-sequence per round (tests/test_huff_emu.py): the scan may stay on the CPU, crawl
    through extra rounds or come back from the device -- the pixels are the same."""
    data = document_like(1600, 1200, mixed=kind == "mixed", gray=kind == "page-gray")
    o = zj.ZuneJpegOptions()
    o.entropy = getattr(zj, "ENTROPY_" + mode)
    g = zj.Decoder(o, ctx)
    assert np.array_equal(g.decode_buffer(data), zj.Decoder(None, ctx).decode_buffer(data))
    assert g.gpu_status() in (0, 32)


def test_reference_file_with_the_eoi_cut(zj, ctx):
    data = open(os.path.join(ROOT, "tests", "golden", "test-baseline.jpg"), "rb").read()
    for sub in (32, 128):
        g, c = decoders(zj, ctx, sub=sub)
        got = g.decode_buffer(data)
        assert g.gpu_status() == 0
        assert np.array_equal(got, c.decode_buffer(data))


def test_damaged_scans_fall_back_to_the_cpu_walker(zj, ctx):
    base = pil_jpeg(320, 200, quality=85, seed=5)
    sos = base.index(b"\xff\xda") + 14
    rng = np.random.default_rng(12)
    handed = 0
    for trial in range(40):
        b = bytearray(base)
        if trial % 3 == 2:
            b = b[: sos + int(rng.integers(8, len(base) - sos - 2))] + b"\xff\xd9"
        else:
            k = int(rng.integers(sos, len(b) - 2))
            b[k] ^= 1 << int(rng.integers(0, 8))
        g, c = decoders(zj, ctx)
        try:
            want = c.decode_buffer(bytes(b))
        except zj.DecodeError as e:
            with pytest.raises(zj.DecodeError) as ei:
                g.decode_buffer(bytes(b))
            assert ei.value.status == e.status
            continue
        assert np.array_equal(g.decode_buffer(bytes(b)), want), trial
        handed += g.gpu_status() != 0
    assert handed


def test_pixels_stay_in_hbm(zj, ctx):
    data = pil_jpeg(800, 600, quality=90, seed=21)
    g, c = decoders(zj, ctx)
    want = c.decode_buffer(data)
    g.prepare(data)
    p = ctx.device_alloc(want.size + 64)
    try:
        n = g.finish_pixels_device(p, want.size)
        assert n == want.size
        got = np.zeros(want.size, np.uint8)
        ctx.d2h(got, p)
        assert np.array_equal(got, want)
        # the CPU-entropy path into HBM
        c.prepare(data)
        assert c.finish_pixels_device(p, want.size) == want.size
        ctx.d2h(got, p)
        assert np.array_equal(got, want)
    finally:
        ctx.device_free(p)


def test_pool_with_the_device_entropy_stage(zj):
    files = [pil_jpeg(640 + 16 * k, 480, quality=80 + k, seed=k, **({"restart_marker_rows": 1} if k % 2 else {})) for k in range(12)]
    o = zj.ZuneJpegOptions()
    o.entropy = zj.ENTROPY_GPU_ALWAYS
    with zj.Pool(4, o) as pool:
        outs, infos, sts = pool.decode_files(files)
    assert all(s == 0 for s in sts)
    with zj.Pool(4) as pool:
        want, _, _ = pool.decode_files(files)
    for a, b in zip(outs, want):
        assert np.array_equal(a, b)


def test_pool_leaves_pixels_in_hbm(zj, ctx):
    files = [pil_jpeg(512 + 16 * k, 384, quality=85, seed=30 + k) for k in range(6)]
    with zj.Pool(2) as pool:
        want, _, _ = pool.decode_files(files)
    o = zj.ZuneJpegOptions()
    o.entropy = zj.ENTROPY_GPU_ALWAYS
    ptrs = [ctx.device_alloc(w.size + 64) for w in want]
    try:
        with zj.Pool(2, o) as pool:
            lens, _, sts = pool.decode_files_device(files, ptrs, [w.size for w in want])
        assert all(s == 0 for s in sts) and lens == [w.size for w in want]
        for p, w in zip(ptrs, want):
            got = np.zeros(w.size, np.uint8)
            ctx.d2h(got, p)
            assert np.array_equal(got, w)
    finally:
        for p in ptrs:
            ctx.device_free(p)


def test_batch_of_scans_in_one_go(zj, ctx):
    """zj_decoder_finish_pixels_batch: device scans of different geometry together (one launch per phase), a progressive
    file and a damaged one (handed back) among them; every result equals the file's own decode."""
    from PIL import Image
    import io
    files = [pil_jpeg(400 + 40 * k, 300 + 8 * k, quality=70 + 3 * k, seed=40 + k, subsampling=k % 3) for k in range(9)]
    files.append(pil_jpeg(320, 240, quality=80, gray=True, seed=3))
    b = io.BytesIO()
    Image.fromarray(np.random.default_rng(5).integers(0, 256, (200, 300, 3), dtype=np.uint8)).save(b, "JPEG", quality=80, progressive=True)
    files.append(b.getvalue())
    bad = bytearray(files[2])
    bad[len(bad) // 2] ^= 0x55
    files.append(bytes(bad))
    want = []
    for f in files:
        try:
            want.append(zj.Decoder(None, ctx).decode_buffer(f))
        except zj.DecodeError as e:
            want.append(e.status)
    decs = []
    for f in files:
        o = zj.ZuneJpegOptions()
        o.entropy = zj.ENTROPY_GPU_ALWAYS
        d = zj.Decoder(o, ctx)
        d.prepare(f)
        decs.append(d)
    outs, rcs = zj.finish_pixels_batch(decs, ctx)
    for k, (o, rc, w) in enumerate(zip(outs, rcs, want)):
        if isinstance(w, int):
            assert rc == w, k
        else:
            assert rc == 0 and np.array_equal(o, w), k
    # and with the pixels left in HBM
    ptrs = [(ctx.device_alloc(o.size + 64), o.size) for o in outs]
    try:
        for d, f in zip(decs, files):
            d.prepare(f)
        lens, rcs = zj.finish_pixels_batch(decs, ctx, device_ptrs=ptrs)
        for k, (p, w) in enumerate(zip(ptrs, want)):
            if isinstance(w, int):
                continue
            got = np.zeros(w.size, np.uint8)
            ctx.d2h(got, p[0])
            assert rcs[k] == 0 and lens[k] == w.size and np.array_equal(got, w), k
    finally:
        for p, _ in ptrs:
            ctx.device_free(p)


def test_scans_of_one_geometry_share_a_pixel_launch(zj, ctx):
    """Same size, same tables: the batch runs them through the pixel kernel as frames of one launch (planes at the arena's
    stride; device outputs equally spaced, like the images of one tensor).  A different file in the middle splits the run."""
    files = [pil_jpeg(640, 480, quality=88, seed=60 + k) for k in range(7)]
    files.insert(3, pil_jpeg(656, 480, quality=88, seed=99))
    want = [zj.Decoder(None, ctx).decode_buffer(f) for f in files]
    decs = []
    for f in files:
        o = zj.ZuneJpegOptions()
        o.entropy = zj.ENTROPY_GPU_ALWAYS
        d = zj.Decoder(o, ctx)
        d.prepare(f)
        decs.append(d)
    outs, rcs = zj.finish_pixels_batch(decs, ctx)
    assert not any(rcs)
    for o, w in zip(outs, want):
        assert np.array_equal(o, w)
    step = 656 * 480 * 3
    base = ctx.device_alloc(step * len(files) + 64)
    try:
        for d, f in zip(decs, files):
            d.prepare(f)
        lens, rcs = zj.finish_pixels_batch(decs, ctx, device_ptrs=[(base + k * step, step) for k in range(len(files))])
        assert not any(rcs)
        for k, w in enumerate(want):
            got = np.zeros(w.size, np.uint8)
            ctx.d2h(got, base + k * step)
            assert lens[k] == w.size and np.array_equal(got, w), k
    finally:
        ctx.device_free(base)


def test_files_to_a_torch_tensor(zj, ctx):
    """FileBatchDecoder.to_tensor: 20 files (more than one batch) of one size -> [N, H, W, 3] uint8 on the GPU; a
    progressive file among them takes the CPU walker, same bytes."""
    import io
    import torch
    from PIL import Image
    files = [pil_jpeg(512, 384, quality=85, seed=70 + k) for k in range(19)]
    b = io.BytesIO()
    Image.fromarray(np.random.default_rng(9).integers(0, 256, (384, 512, 3), dtype=np.uint8)).save(b, "JPEG", quality=85, progressive=True)
    files.insert(7, b.getvalue())
    fb = zj.FileBatchDecoder(ctx)
    try:
        t = fb.to_tensor(files)
        assert tuple(t.shape) == (20, 384, 512, 3) and t.dtype == torch.uint8 and t.is_cuda
        got = t.cpu().numpy()
        for k, f in enumerate(files):
            assert np.array_equal(got[k].reshape(-1), zj.Decoder(None, ctx).decode_buffer(f)), k
    finally:
        fb.close()


def test_blown_out_sky_is_bridged_by_the_periodic_run_rule(zj, ctx):
    """A third of the image is one flat run (hundreds of identical sub-sequences): without the rule of zj_huff.h the true
    state would cross it one sub-sequence per round and the scan would come back to the CPU."""
    from PIL import Image
    import io
    a = np.full((2048, 2048, 3), 255, np.uint8)
    small = np.random.default_rng(1).integers(0, 256, (45, 64, 3), dtype=np.uint8)
    a[700:] = np.asarray(Image.fromarray(small).resize((2048, 1348), Image.BICUBIC))
    b = io.BytesIO()
    Image.fromarray(a).save(b, "JPEG", quality=90)
    data = b.getvalue()
    o = zj.ZuneJpegOptions()
    o.entropy = zj.ENTROPY_GPU
    g = zj.Decoder(o, ctx)
    got = g.decode_buffer(data)
    assert g.scan_blob() is not None and g.gpu_status() == 0
    assert ctx.scan_stats()[0] <= 24
    assert np.array_equal(got, zj.Decoder(None, ctx).decode_buffer(data))


def test_pool_to_a_torch_tensor(zj, ctx):
    import torch
    files = [pil_jpeg(512, 384, quality=82, seed=120 + k) for k in range(24)]
    o = zj.ZuneJpegOptions()
    o.entropy = zj.ENTROPY_GPU_ALWAYS
    with zj.Pool(3, o) as pool:
        t = pool.to_tensor(files)
    assert tuple(t.shape) == (24, 384, 512, 3) and t.dtype == torch.uint8
    got = t.cpu().numpy()
    for k, f in enumerate(files):
        assert np.array_equal(got[k].reshape(-1), zj.Decoder(None, ctx).decode_buffer(f)), k


def test_malformed_scan_blobs_are_argument_errors(zj, ctx):
    """zj_decode_scan takes any blob.  Every header field the kernels index with is bounded on the host, so a
    stale or corrupted prepared scan comes back as ZJ_ERR_ARG instead of an out-of-bounds access on the device."""
    import struct
    data = pil_jpeg(256, 128, quality=90, seed=3)
    g, _ = decoders(zj, ctx)
    desc, info = g.prepare(data)
    blob = g.scan_blob()
    assert blob is not None
    out, rc, st = ctx.decode_scan(desc, blob)
    assert rc == 0 and st == 0
    # u32 fields of HuffScan (csrc/zj_huff.h): index -> a value that must be refused
    names = ["magic", "blob_bytes", "nsub", "nseg", "ri_mcus", "bpm", "ncomp", "mcu_x", "mcu_y", "total_mcus", "is_eoi", "rowlen",
             "tab_entries", "off_tab", "off_sub", "off_seg", "off_stream", "stream_bytes", "sub_bytes", "round_budget", "off_per"]
    bad = {"nseg": 1 << 20, "bpm": 11, "mcu_x": 0, "total_mcus": 7, "rowlen": 0, "tab_entries": 60000, "off_tab": 8,
           "off_sub": len(blob) - 16, "off_seg": len(blob), "off_stream": len(blob) - 48, "stream_bytes": 1 << 30,
           "sub_bytes": 4096, "off_per": len(blob) + 1024, "ri_mcus": 0}
    for name, val in bad.items():
        b = blob.copy()
        struct.pack_into("<I", b, 4 * names.index(name), val)
        with pytest.raises(zj.ZjError) as e:
            ctx.decode_scan(desc, b)
        assert e.value.status == -1, name
    # a block of the MCU that claims a component the scan does not have
    b = blob.copy()
    off_blk = 4 * len(names) + 4 * 2 + 2 * 8   # ... off_per, nper, comp_of_blk, dc_off[4], ac_off[4] -> blk[]
    b[off_blk] = 7
    with pytest.raises(zj.ZjError):
        ctx.decode_scan(desc, b)
    # three more inputs the kernels index with.  (1) periodic-run words: a period outside 1..8, or less than two
    # whole periods in front of the sub-sequence; (2) a plane narrower than the MCU grid writes into it; (3) a first-level
    # table entry naming a second-level table beyond the tables
    hdr = struct.unpack_from("<%dI" % len(names), blob, 0)
    off_per, off_tab, nsub = hdr[names.index("off_per")], hdr[names.index("off_tab")], hdr[names.index("nsub")]
    assert nsub > 8
    for word in ((1 << 28) | 5, (9 << 28) | 0, (0 << 28) | 1, (2 << 28) | 3):
        b = blob.copy()
        struct.pack_into("<I", b, off_per + 4 * 6, word)          # word of sub-sequence 6
        with pytest.raises(zj.ZjError) as e:
            ctx.decode_scan(desc, b)
        assert e.value.status == -1, hex(word)
    off_comp = off_blk + 4 * 10                                    # blk[HUFF_MAX_BPM] -> comp[]: h, v, bw, bh
    for field in (2, 3):
        b = blob.copy()
        cur = struct.unpack_from("<I", b, off_comp + 4 * field)[0]
        struct.pack_into("<I", b, off_comp + 4 * field, cur - 1)
        with pytest.raises(zj.ZjError) as e:
            ctx.decode_scan(desc, b)
        assert e.value.status == -1, field
    b = blob.copy()
    struct.pack_into("<H", b, off_tab + 2 * 3, 0x80FF)             # first-level entry 3 of the first table
    with pytest.raises(zj.ZjError) as e:
        ctx.decode_scan(desc, b)
    assert e.value.status == -1
    # and the context still works
    out2, rc, st = ctx.decode_scan(desc, blob)
    assert rc == 0 and st == 0 and np.array_equal(out, out2)
