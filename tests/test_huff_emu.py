"""CPU: the GPU entropy stage (zune-jpeg_amd/csrc/zj_huff_device.h) run thread by thread by the emulation harness
(tests/emu) over scans the product's front-end prepared (zj_decoder_prepare), against the planes of the product's CPU
walker (zj_decoder_decode_coefficients) -- which tests/test_jpeg_frontend.py pins to the encoder's coefficients, to
Pillow/libjpeg and to the reference's walk (oracle/ref_walk.py).  The same comparisons run on the GPU in
tests/test_gpu_entropy.py."""
import importlib
import io
import os
import sys

import numpy as np
import pytest

import emu_c

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import jpeg_enc  # noqa: E402

MODES = {"none": (1, 1), "h": (2, 1), "v": (1, 2), "hv": (2, 2)}


@pytest.fixture(scope="module")
def zj():
    return importlib.import_module("zune-jpeg_amd")


def gpu_vs_cpu(zj, data, sub=None, expect_status=0, truth=None):
    """prepare + emulate; returns (stats, status).  Asserts equal planes when the device keeps the scan."""
    old = os.environ.get("ZJ_HUFF_SUB")
    if sub:
        os.environ["ZJ_HUFF_SUB"] = str(sub)
    try:
        o = zj.ZuneJpegOptions()
        o.entropy = zj.ENTROPY_GPU_ALWAYS
        d = zj.Decoder(o)
    finally:
        if sub:
            if old is None:
                del os.environ["ZJ_HUFF_SUB"]
            else:
                os.environ["ZJ_HUFF_SUB"] = old
    desc, info = d.prepare(data)
    blob = d.scan_blob()
    assert blob is not None, "the front-end did not prepare the scan for the device"
    _, want, _ = zj.Decoder().decode_coefficients(data)
    got, status, st = emu_c.huff_decode(blob, [p.size for p in want])
    if expect_status is not None:
        assert status == expect_status, (status, st)
    if status == 0:
        for c, (a, b) in enumerate(zip(got, want)):
            assert np.array_equal(a, b), (c, st, np.nonzero(a != b)[0][:8])
        if truth is not None:  # the encoder's coefficients: ground truth that owes nothing to the CPU walker
            for c, (a, b) in enumerate(zip(got, truth)):
                assert np.array_equal(a, b), (c, st)
    return st, status


def pil_jpeg(w, h, quality, subsampling=2, gray=False, seed=0, flat=False, **kw):
    from PIL import Image
    rng = np.random.default_rng(seed)
    yy, xx = np.mgrid[0:h, 0:w]
    img = (128 + 80 * np.sin(xx / 37.0 + seed) * np.cos(yy / 23.0))[..., None] + rng.normal(0, 0 if flat else 18, (h, w, 3))
    im = Image.fromarray(np.clip(img, 0, 255).astype(np.uint8))
    if gray:
        im = im.convert("L")
    else:
        kw["subsampling"] = subsampling
    b = io.BytesIO()
    im.save(b, "JPEG", quality=quality, **kw)
    return b.getvalue()


@pytest.mark.parametrize("mode", list(MODES))
@pytest.mark.parametrize("wh", [(64, 48), (50, 37), (17, 9), (200, 120)])
@pytest.mark.parametrize("restart", [0, 3])
def test_encoder_round_trip(zj, synth, mode, wh, restart):
    """jpeg_enc writes one table pair for all components: the place in the MCU only heals from the front, one
    sub-sequence per round -- the slowest way through the rounds."""
    hs, vs = MODES[mode]
    w, h = wh
    planes = jpeg_enc.small_planes(w, h, hs, vs, 3, seed=w + h)
    data = jpeg_enc.encode_baseline(planes, synth.quant_tables(85), w, h, hs, vs, 3, restart=restart)
    if restart == 0 and w > 100:
        # ~200 sub-sequences in one segment: more rounds than the device spends before it hands the scan back
        st, status = gpu_vs_cpu(zj, data, sub=128, expect_status=None)
        assert status in (0, 32)
        return
    st, _ = gpu_vs_cpu(zj, data, sub=64 if restart == 0 else 16, truth=planes)
    assert st["nsub"] > 1


def test_grayscale_and_extreme_coefficients(zj, synth):
    w, h = 120, 64
    planes = jpeg_enc.small_planes(w, h, 1, 1, 1, seed=3, amp=900, dc=1000)  # 10-bit magnitudes, long runs
    gpu_vs_cpu(zj, jpeg_enc.encode_baseline(planes, synth.quant_tables(90), w, h, 1, 1, 1), sub=32)


@pytest.mark.parametrize("sub", [16, 48, 128])
@pytest.mark.parametrize("case", ["420", "444", "422", "gray", "420-opt", "420-ri7", "420-ri1", "420-q20", "420-rows", "444-flat"])
def test_libjpeg_files(zj, case, sub):
    kw = dict(quality=90)
    if case == "444":
        kw.update(subsampling=0)
    elif case == "422":
        kw.update(subsampling=1)
    elif case == "gray":
        kw.update(gray=True, quality=50)
    elif case == "420-opt":
        kw.update(optimize=True, quality=75)     # optimised tables: many codes longer than 9 bits
    elif case == "420-ri7":
        kw.update(restart_marker_blocks=7)
    elif case == "420-ri1":
        kw.update(restart_marker_blocks=1)       # a restart segment per MCU: sub-sequences of ~40 bytes
    elif case == "420-q20":
        kw.update(quality=20)
    elif case == "420-rows":
        kw.update(restart_marker_rows=1)
    elif case == "444-flat":
        kw.update(subsampling=0, flat=True, quality=30)   # cheap last MCUs: the reference's early exit at EOI
    w, h = (333, 211) if case != "420-opt" else (520, 301)
    st, _ = gpu_vs_cpu(zj, pil_jpeg(w, h, seed=len(case), **kw), sub=sub)
    assert st["rounds"] * sub <= 4096, st  # real tables: a wrong guess falls into step within a few hundred bytes


def test_reference_file_with_the_eoi_cut(zj):
    """tests/golden/test-baseline.jpg: the reference never decodes the last 7 MCUs of the last row (oracle/ref_walk.py,
    tests/test_jpeg_frontend.py); the device stage finds the same MCU with the same rule and clears them."""
    data = open(os.path.join(ROOT, "tests", "golden", "test-baseline.jpg"), "rb").read()
    st, _ = gpu_vs_cpu(zj, data)
    assert st["first_seen"] == 240 * 135 - 8
    st, _ = gpu_vs_cpu(zj, data, sub=32)
    assert st["first_seen"] == 240 * 135 - 8


def test_damaged_scans_are_handed_back_or_equal(zj):
    """Bit flips and truncation: whatever the device keeps (status 0) must equal the CPU walker's planes; everything
    else must come back with a status.  Never a crash, never an endless loop."""
    base = pil_jpeg(160, 96, quality=85, seed=5)
    sos = base.index(b"\xff\xda") + 14
    rng = np.random.default_rng(11)
    kept = handed = 0
    for trial in range(60):
        b = bytearray(base)
        if trial % 3 == 2:
            b = b[: sos + int(rng.integers(8, len(base) - sos - 2))] + b"\xff\xd9"
        else:
            for _ in range(1 + trial % 2):
                k = int(rng.integers(sos, len(b) - 2))
                b[k] ^= 1 << int(rng.integers(0, 8))
        o = zj.ZuneJpegOptions()
        o.entropy = zj.ENTROPY_GPU_ALWAYS
        d = zj.Decoder(o)
        try:
            d.prepare(bytes(b))
        except zj.DecodeError:
            continue
        blob = d.scan_blob()
        if blob is None:
            continue  # a stray marker in the data: the front-end kept the scan
        try:
            _, want, _ = zj.Decoder().decode_coefficients(bytes(b))
        except zj.DecodeError:
            want = None
        got, status, st = emu_c.huff_decode(blob, [(160 // 16 + 0) * 16 // 8 * (96 // 8) * 64, 10 * 6 * 64, 10 * 6 * 64])
        if status == 0:
            assert want is not None, (trial, "the CPU walker rejects what the device kept")
            for a, w_ in zip(got, want):
                assert np.array_equal(a[: w_.size], w_), trial
            kept += 1
        else:
            handed += 1
    assert kept and handed


def test_truncated_restart_segments_are_handed_back_or_equal(zj):
    """bytes deleted right in front of an RSTn marker.  The segment's last symbol then begins inside the segment
    and runs past its end; the device used to read the next segment's bytes there while the CPU walker's reader feeds
    zeros -- different coefficients with status 0.  Now such a scan comes back with HUFF_ST_EXHAUSTED."""
    import io
    from PIL import Image
    rng = np.random.default_rng(21)
    kept = handed = 0
    for trial in range(260):   # without the status, trials 177 and 231 of this sequence keep a scan with other coefficients
        w, h = 160, 96
        a = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
        bio = io.BytesIO()
        sub = [0, 2, 1][trial % 3]
        Image.fromarray(a).save(bio, "JPEG", quality=60 + trial % 30, subsampling=sub, restart_marker_blocks=2 + trial % 5)
        base = bio.getvalue()
        sos = base.index(b"\xff\xda") + 14
        rst = [i for i in range(sos, len(base) - 1) if base[i] == 0xFF and 0xD0 <= base[i + 1] <= 0xD7]
        if not rst:
            continue
        k = rst[int(rng.integers(0, len(rst)))]
        cut = 1 + trial % 3
        if any(x == 0xFF or x == 0 for x in base[k - cut - 1:k]):  # keep the byte stuffing intact
            continue
        b = base[:k - cut] + base[k:]
        o = zj.ZuneJpegOptions()
        o.entropy = zj.ENTROPY_GPU_ALWAYS
        d = zj.Decoder(o)
        try:
            d.prepare(bytes(b))
        except zj.DecodeError:
            continue
        blob = d.scan_blob()
        if blob is None:
            continue
        try:
            desc, want, info = zj.Decoder().decode_coefficients(bytes(b))
        except zj.DecodeError:
            want = None
        lens = [p.size for p in want] if want is not None else None
        if lens is None:
            continue
        for ss in (None,):
            got, status, st = emu_c.huff_decode(blob, lens)
            if status == 0:
                for g_, w_ in zip(got, want):
                    assert np.array_equal(g_[: w_.size], w_), (trial, "device kept a truncated segment with other coefficients")
                kept += 1
            else:
                handed += 1
    assert handed > 150, (kept, handed)


def document_like(w, h, mixed=False, gray=False, **kw):
    """white page, a few bars; mixed: the lower half is noise"""
    from PIL import Image
    a = np.full((h, w, 3), 255, np.uint8)
    a[h // 10:h // 10 + 30, w // 20:w - w // 20] = 0
    a[h // 3:h // 3 + 50, w // 6:w - w // 4] = (200, 30, 30)
    if mixed:
        a[h // 2:] = np.random.default_rng(3).integers(0, 256, (h - h // 2, w, 3), dtype=np.uint8)
    im = Image.fromarray(a)
    if gray:
        im = im.convert("L")
    b = io.BytesIO()
    im.save(b, "JPEG", quality=85, **kw)
    return b.getvalue()


@pytest.mark.parametrize("kind", ["420", "444", "422", "gray", "mixed", "sky", "sky-opt"])
def test_flat_areas_and_the_periodic_run_rule(zj, kind):
    """A stream of identical tiny blocks is periodic: an out-of-step decoder settles into a cycle of its own, and the
    true state would advance one sub-sequence per round (a white page: 260 rounds for 580 sub-sequences).  Where
    sub-sequence i repeats the bytes of sub-sequence i - q the device copies the exit states of the run's second period
    forward (zj_huff.h) and verifies them by decoding: a few dozen rounds, still exact."""
    from PIL import Image
    if kind in ("sky", "sky-opt"):  # a blown-out sky over noise: one long run, then busy data
        a = np.full((1024, 1024, 3), 255, np.uint8)
        a[400:] = np.random.default_rng(1).integers(0, 256, (624, 1024, 3), dtype=np.uint8)
        b = io.BytesIO()
        Image.fromarray(a).save(b, "JPEG", quality=90, optimize=kind == "sky-opt")
        data = b.getvalue()
    else:
        kw = {"420": {}, "444": {"subsampling": 0}, "422": {"subsampling": 1}, "gray": {"gray": True}, "mixed": {"mixed": True}}[kind]
        data = document_like(2048, 1536, **kw)
    st, status = gpu_vs_cpu(zj, data)
    assert st["rounds"] <= 90, st


def test_flat_pages_are_left_to_the_cpu(zj):
    """Scans of few bits per block are the CPU walker's cheapest; the front-end keeps them there unless forced."""
    o = zj.ZuneJpegOptions()
    o.entropy = zj.ENTROPY_GPU
    d = zj.Decoder(o)
    d.prepare(document_like(2048, 1536))  # 60 KB of scan for 3 megapixels
    assert d.scan_blob() is None
    d.prepare(document_like(1024, 768, mixed=True))
    assert d.scan_blob() is not None


def test_shared_tables_are_left_to_the_cpu_unless_forced(zj, synth):
    planes = jpeg_enc.small_planes(64, 48, 2, 2, 3, seed=1)
    data = jpeg_enc.encode_baseline(planes, synth.quant_tables(85), 64, 48, 2, 2, 3)
    o = zj.ZuneJpegOptions()
    o.entropy = zj.ENTROPY_GPU
    d = zj.Decoder(o)
    d.prepare(data)
    assert d.scan_blob() is None  # (also: far below the 32 KB a trip to the device is worth)
    big = pil_jpeg(640, 480, quality=92, seed=2)
    d.prepare(big)
    assert d.scan_blob() is not None


def test_long_dc_symbols_the_reference_reads_short_are_handed_back(zj):
    """src/bitstream.rs:278: a DC symbol longer than the reference's reader holds is read short and the stream
    desynchronises (tests/test_jpeg_frontend.py, zj_jpeg.cpp ref_dc_misread).  Only the CPU walker follows that reader;
    the device stage must never keep such a scan (it would decode the file CORRECTLY, i.e. differently): whatever it keeps
    equals the walker's planes, and every file with a short read comes back with HUFF_ST_DC_LONG."""
    import ref_walk
    from test_jpeg_frontend import _noisy_jpeg
    kept = back = short_files = 0
    for seed in range(60):
        data = _noisy_jpeg(seed, [0, 2, 1][seed % 3], w=128, h=96)
        try:
            _, short, _ = ref_walk.decode_baseline_planes(data)
        except ValueError:   # the desynchronised stream ran into a code that does not exist: a short read happened
            short = 1
        o = zj.ZuneJpegOptions()
        o.entropy = zj.ENTROPY_GPU_ALWAYS
        d = zj.Decoder(o)
        d.prepare(data)
        blob = d.scan_blob()
        if blob is None:
            continue
        try:
            _, want, _ = zj.Decoder().decode_coefficients(data)
        except zj.DecodeError:
            want = None
        if want is None:     # the walker fails like the reference (bad code behind the short read): size the planes by hand
            info = zj.Decoder().read_headers(data)
            mx, my = -(-info.width // (8 * info.h_max)), -(-info.height // (8 * info.v_max))
            lens = [mx * my * info.h_max * info.v_max * 64, mx * my * 64, mx * my * 64]
        else:
            lens = [p.size for p in want]
        got, status, st = emu_c.huff_decode(blob, lens)
        if short:
            short_files += 1
            assert status & 64, (seed, status)
        if status == 0:
            kept += 1
            assert want is not None, seed
            for a, b in zip(got, want):
                assert np.array_equal(a[: b.size], b), seed
        else:
            back += 1
    assert short_files >= 2 and kept >= 10, (short_files, kept, back)


def test_intervals_with_bytes_left_over_are_handed_back(zj):
    """Bytes inserted into a restart interval: its blocks are complete with data left in front of the marker, and the reference
    decodes the NEXT interval out of what is left (handle_restart() sees no marker yet), predictors and all.  The device's
    intervals are independent: it must say HUFF_ST_LEFT_OVER (128) -- until round 6 it kept such scans, with other
    coefficients than the serial walk's (and than the reference's)."""
    data = pil_jpeg(320, 240, 90, subsampling=2, seed=5, restart_marker_rows=1)
    rst = [i for i in range(len(data) - 1) if data[i] == 0xFF and 0xD0 <= data[i + 1] <= 0xD7]
    assert len(rst) == 14
    rng = np.random.default_rng(8)
    flagged = 0
    for trial in range(40):
        k = int(rng.integers(0, len(rst) - 1))
        at = int(rng.integers(rst[k] + 30, rst[k + 1] - 30))
        d = bytearray(data)
        d[at:at] = bytes(rng.integers(1, 255, int(rng.integers(12, 60)), dtype=np.uint8))
        d = bytes(d)
        o = zj.ZuneJpegOptions()
        o.entropy = zj.ENTROPY_GPU_ALWAYS
        dec = zj.Decoder(o)
        try:
            dec.prepare(d)
        except zj.DecodeError:
            continue
        blob = dec.scan_blob()
        if blob is None:
            continue
        s1 = zj.ZuneJpegOptions()
        s1.num_threads = 1
        try:
            _, want, _ = zj.Decoder(s1).decode_coefficients(d)
        except zj.DecodeError:
            want = None
        lens = [20 * 15 * 4 * 64, 20 * 15 * 64, 20 * 15 * 64]
        got, status, st = emu_c.huff_decode(blob, lens)
        if status == 0:   # kept: then it must be what one thread makes of the file
            assert want is not None
            for g_, w_ in zip(got, want):
                assert np.array_equal(g_[: w_.size], w_), (trial, "the device kept an interval with bytes left over")
        flagged += bool(status & 128)
    assert flagged >= 10, flagged
