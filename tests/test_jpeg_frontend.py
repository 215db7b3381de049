"""CPU: the entropy front-end (zune-jpeg_amd/csrc/zj_jpeg.cpp: markers + Huffman, baseline and
progressive) -- bit-exact coefficient round trips through tools/jpeg_enc.py, the reference's own
error-path tests (tests/invalid_images.rs), and Pillow/libjpeg as an independent decoder."""
import importlib
import io
import os
import sys

import numpy as np
import pytest

import oracle_c as oc
import ref_walk

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import jpeg_enc  # noqa: E402

MODES = {"none": (1, 1), "h": (2, 1), "v": (1, 2), "hv": (2, 2)}


@pytest.fixture(scope="module")
def zj():
    return importlib.import_module("zune-jpeg_amd")


@pytest.mark.parametrize("mode", list(MODES))
@pytest.mark.parametrize("wh", [(64, 48), (50, 37), (17, 9)])
@pytest.mark.parametrize("kind", ["baseline", "baseline_rst", "progressive"])
def test_coefficient_round_trip(zj, synth, mode, wh, kind):
    hs, vs = MODES[mode]
    w, h = wh
    planes = jpeg_enc.small_planes(w, h, hs, vs, 3, seed=w + h)
    qts = synth.quant_tables(85)
    if kind == "progressive":
        data = jpeg_enc.encode_progressive(planes, qts, w, h, hs, vs, 3)
    else:
        data = jpeg_enc.encode_baseline(planes, qts, w, h, hs, vs, 3, restart=3 if kind == "baseline_rst" else 0)
    dec = zj.Decoder()
    desc, got, info = dec.decode_coefficients(data)
    assert (info.width, info.height, info.components) == (w, h, 3)
    assert (info.h_max, info.v_max) == (hs, vs)
    assert info.progressive == (kind == "progressive")
    assert info.scans == (10 if kind == "progressive" else 1)
    for c in range(3):
        assert np.array_equal(got[c], planes[c]), (mode, wh, kind, c)
        assert np.array_equal(np.ctypeslib.as_array(desc.qt[c]), qts[c])


@pytest.mark.parametrize("kind", ["baseline", "progressive"])
def test_grayscale_round_trip(zj, synth, kind):
    w, h = 70, 33
    planes = jpeg_enc.small_planes(w, h, 1, 1, 1, seed=9)
    qts = synth.quant_tables(90)
    enc = jpeg_enc.encode_progressive if kind == "progressive" else jpeg_enc.encode_baseline
    desc, got, info = zj.Decoder().decode_coefficients(enc(planes, qts, w, h, 1, 1, 1))
    assert info.components == 1 and desc.out_colorspace == 1  # forced GRAYSCALE, headers.rs:283-290
    assert np.array_equal(got[0], planes[0])


@pytest.mark.parametrize("kind", ["baseline", "progressive"])
def test_pillow_decodes_the_same_file(zj, synth, kind):
    """4:4:4 so that only IDCT rounding differs between libjpeg and the reference's arithmetic."""
    from PIL import Image
    w, h = 64, 48
    planes = jpeg_enc.small_planes(w, h, 1, 1, 3, seed=4, amp=3, dc=40)  # keeps every pixel inside 0..255
    qts = synth.quant_tables(90)
    enc = jpeg_enc.encode_progressive if kind == "progressive" else jpeg_enc.encode_baseline
    data = enc(planes, qts, w, h, 1, 1, 3)
    im = Image.open(io.BytesIO(data))
    im.draft("YCbCr", im.size)  # libjpeg hands out YCbCr directly: no RGB round trip, no gamut clipping
    assert im.mode == "YCbCr"
    pil = np.asarray(im, np.int32)
    desc, got, info = zj.Decoder().decode_coefficients(data)
    rc, ours = oc.decode_planes(oc.make_frame(w, h, 1, 1, 3, oc.YCBCR, list(np.ctypeslib.as_array(desc.qt))), got)
    assert rc == 0
    d = np.abs(ours.reshape(h, w, 3).astype(np.int32) - pil)
    assert d.max() <= 3 and d.mean() < 0.6


def _err(zj, data):
    with pytest.raises(zj.DecodeError) as e:
        zj.Decoder().decode_coefficients(bytes(data))
    return e.value


def test_reference_invalid_image_cases(zj):
    """tests/invalid_images.rs:3-81 -- same variants and (where asserted) the same strings"""
    assert _err(zj, [0xff, 0xd8, 0xa4]).status == -20                                   # eof -> Format(_)
    e = _err(zj, [0xff, 0xd8, 0xff, 0x00, 0x00, 0x00])                                  # bad_ff_marker_size
    assert e.status == -20 and e.text == "Found a marker with invalid length : 0"
    e = _err(zj, [255, 216, 255, 218, 232, 197, 255])                                   # bad_number_of_scans
    assert e.status == -25 and e.text == "Bad SOS length,corrupt jpeg"
    e = _err(zj, [255, 216, 255, 196, 0, 0])                                            # huffman_length_subtraction_overflow
    assert e.status == -20 and e.text == "Invalid Huffman length in image"
    e = _err(zj, [255, 216, 255, 192, 255, 1, 8, 9, 119, 48, 255, 192])                 # mul_with_overflow
    assert e.status == -26 and e.text == "Length of start of frame differs from expected 584,value is 65281"
    assert _err(zj, [0x12, 0x34]).status == -21                                         # IllegalMagicBytes


def test_limits_and_unsupported(zj, synth):
    planes = jpeg_enc.small_planes(40, 24, 1, 1, 3, seed=1)
    data = bytearray(jpeg_enc.encode_baseline(planes, synth.quant_tables(80), 40, 24))
    o = zj.ZuneJpegOptions()
    o.max_width = 32
    with pytest.raises(zj.DecodeError) as e:
        zj.Decoder(o).decode_coefficients(bytes(data))
    assert "greater than width limit 32" in e.value.text
    sof = data.index(b"\xff\xc0")
    bad = bytearray(data); bad[sof + 4] = 12                                            # 12-bit precision
    assert "8-bit images" in _err(zj, bad).text
    bad = bytearray(data); bad[sof + 1] = 0xC9                                          # arithmetic SOF9 is skipped as unknown
    assert _err(zj, bad).status in (-26, -25)                                           # ... then SOS finds no frame
    info = zj.Decoder().read_headers(bytes(data))
    assert (info.width, info.height, info.components, info.progressive) == (40, 24, 3, 0)


REF_IMAGES = [("test-baseline.jpg", 0, 1), ("test-progressive.jpg", 1, 10)]


@pytest.mark.parametrize("name,prog,scans", REF_IMAGES)
def test_reference_images_vs_pillow(zj, name, prog, scans):
    """BASELINE.json configs[0]/[3]: the reference's 1920x1080 4:4:4 test images (copies under
    tests/golden).  CPU entropy decode -> oracle pixel path, compared with libjpeg (Pillow)."""
    from PIL import Image
    path = os.path.join(ROOT, "tests", "golden", name)
    data = open(path, "rb").read()
    # libjpeg decodes the values the file codes: ZJ_FLAG_FULL_AC_VALUES.  Without it the front-end gives what the reference
    # gives -- 958 AC coefficients of the progressive file (in 649 blocks) cut to six bits by its fast-AC table
    # (src/huffman.rs:251, include/zjhip.h), none in the baseline file, whose standard tables have no short code for size 6
    o = zj.ZuneJpegOptions()
    o.flags = zj.FLAG_FULL_AC_VALUES
    desc, planes, info = zj.Decoder(o).decode_coefficients(data)
    _, planes_ref, _ = zj.Decoder().decode_coefficients(data)
    cut = np.nonzero(np.concatenate(planes) != np.concatenate(planes_ref))[0]
    assert (cut.size, np.unique(cut // 64).size) == ((958, 649) if prog else (0, 0))
    assert (info.width, info.height, info.components, info.progressive, info.scans) == (1920, 1080, 3, prog, scans)
    qts = list(np.ctypeslib.as_array(desc.qt))
    rc, ours = oc.decode_planes(oc.make_frame(1920, 1080, 1, 1, 3, oc.YCBCR, qts), planes)
    assert rc == 0
    im = Image.open(path)
    im.draft("YCbCr", im.size)
    assert im.mode == "YCbCr"
    pil = np.asarray(im, np.int32)
    d = np.abs(ours.reshape(1080, 1920, 3).astype(np.int32) - pil)
    if not prog:
        # the reference never decodes the last 7 MCUs of this image (it leaves the row loop once its reader has come
        # across EOI, src/mcu.rs:337-343; test_reference_eoi_cut below): they stay zero coefficients = mid grey
        rows = ref_walk.decoded_mcus_per_row(data)
        assert [(i, n) for i, n in enumerate(rows) if n != 240] == [(134, 233)]
        d[134 * 8:, 233 * 8:] = 0
    assert d.max() <= 4 and d.mean() < 0.5


_CUTS = {}


def _flat_tail_planes(w, h, hs, vs, busy_rows, seed):
    """random blocks in the first `busy_rows` MCU rows, then DC-only blocks with a constant non-zero DC: the last
    MCUs cost only their shortest codes, and a decoded block is never all-zero"""
    planes = jpeg_enc.small_planes(w, h, hs, vs, 3, seed=seed)
    mcu_x = (w + 8 * hs - 1) // (8 * hs)
    out = []
    for c, pl in enumerate(planes):
        ch, cv = (hs, vs) if c == 0 else (1, 1)
        bw = mcu_x * ch
        b = np.array(pl, np.int16).reshape(-1, bw, 64)
        b[busy_rows * cv:] = 0
        b[busy_rows * cv:, :, 0] = (50, 20, -30)[c]
        out.append(b.reshape(-1))
    return out


@pytest.mark.parametrize("mode", list(MODES))
@pytest.mark.parametrize("wh", [(64, 48), (200, 40), (17, 33)])
@pytest.mark.parametrize("restart", [0, 5])
@pytest.mark.parametrize("threads", [1, 4])
def test_reference_eoi_cut(zj, synth, mode, wh, restart, threads):
    """Which MCUs get entropy-decoded at all: the front-end against the bit-level model of the reference's reader
    and MCU loop (oracle/ref_walk.py).  Flat image tails make the last MCUs cheap enough for the reference to
    come across EOI before it has decoded them."""
    hs, vs = MODES[mode]
    w, h = wh
    mcu_x, mcu_y = (w + 8 * hs - 1) // (8 * hs), (h + 8 * vs - 1) // (8 * vs)
    cut_somewhere = False
    for busy_rows in (0, 1, mcu_y):
        planes = _flat_tail_planes(w, h, hs, vs, busy_rows, seed=w + h + busy_rows)
        data = jpeg_enc.encode_baseline(planes, synth.quant_tables(85), w, h, hs, vs, 3, restart=restart)
        rows = ref_walk.decoded_mcus_per_row(data)
        desc, got, info = zj.Decoder(_opts(zj, threads)).decode_coefficients(data)
        y = np.array(got[0], np.int16).reshape(mcu_y * vs, mcu_x * hs, 64)
        cb = np.array(got[1], np.int16).reshape(mcu_y, mcu_x, 64)
        for my in range(mcu_y):
            n = rows[my]
            if n is None:      # an MCU row the reference never walks (odd last row of (2,1)/(2,2)): its pixels are
                continue       # dropped by the pixel path whatever the front-end puts there
            cut_somewhere |= n < mcu_x
            exp_y = np.array(planes[0], np.int16).reshape(mcu_y * vs, mcu_x * hs, 64)
            exp_cb = np.array(planes[1], np.int16).reshape(mcu_y, mcu_x, 64)
            assert np.array_equal(y[my * vs:(my + 1) * vs, :n * hs], exp_y[my * vs:(my + 1) * vs, :n * hs]), (busy_rows, my)
            assert np.array_equal(cb[my, :n], exp_cb[my, :n])
            if busy_rows < mcu_y:  # (random tails are too expensive for a cut; values there are covered by the round trips)
                assert not y[my * vs:(my + 1) * vs, n * hs:].any() and not cb[my, n:].any(), (busy_rows, my, n)
    _CUTS[(mode, wh, restart)] = cut_somewhere


@pytest.mark.parametrize("subsampling", [0, 1, 2], ids=["444", "422", "420"])
@pytest.mark.parametrize("wh", [(64, 48), (200, 40), (17, 33), (640, 64), (1000, 24)])
@pytest.mark.parametrize("restart_rows", [0, 1])
@pytest.mark.parametrize("threads", [1, 4])
def test_reference_eoi_cut_libjpeg_files(zj, subsampling, wh, restart_rows, threads):
    """the same on files written by libjpeg (Pillow) with the standard Huffman tables, whose flat MCUs cost 6-18 bits:
    the reference drops up to a handful of MCUs at the end of such images"""
    from PIL import Image
    w, h = wh
    rng = np.random.default_rng(w * h)
    a = np.zeros((h, w, 3), np.uint8)
    a[:] = (200, 60, 30)                       # far from grey: every decoded chroma DC is non-zero
    a[:8] = rng.integers(0, 256, (8, w, 3))    # one busy MCU row on top
    b = io.BytesIO()
    kw = dict(quality=90, subsampling=subsampling)
    if restart_rows:
        kw["restart_marker_rows"] = restart_rows
    Image.fromarray(a).save(b, "JPEG", **kw)
    data = b.getvalue()
    rows = ref_walk.decoded_mcus_per_row(data)
    hs, vs = [(1, 1), (2, 1), (2, 2)][subsampling]
    mcu_x = (w + 8 * hs - 1) // (8 * hs)
    desc, planes, info = zj.Decoder(_opts(zj, threads)).decode_coefficients(data)
    assert bool(info.restart_interval) == bool(restart_rows)
    cb = np.array(planes[1], np.int16).reshape(-1, mcu_x, 64)
    first_flat = 16 // (8 * vs) if vs == 2 else 1      # MCU rows made only of flat pixels
    for r in range(first_flat, cb.shape[0]):
        if rows[r] is None:
            continue
        n = rows[r]
        assert (cb[r, :n, 0] != 0).all() and not cb[r, n:].any(), (r, n, rows)
    _CUTS[("pillow", subsampling, wh, restart_rows)] = any(n is not None and n < mcu_x for n in rows)


def test_reference_eoi_cut_was_exercised():
    """the cases above must contain images where the reference drops MCUs (else they test nothing)"""
    if os.environ.get("PYTEST_XDIST_WORKER"):
        pytest.skip("counts what the cases above saw in THIS process: meaningful in a serial run only (python -m pytest tests -x -q)")
    assert sum(_CUTS.values()) >= 6, _CUTS


def _opts(zj, threads):
    o = zj.ZuneJpegOptions()
    o.num_threads = threads
    return o


@pytest.mark.parametrize("mode,wh,ri", [("hv", (200, 120), 13), ("none", (96, 64), 12), ("h", (130, 40), 1), ("v", (64, 200), 8)])
def test_restart_segments_decode_concurrently(zj, synth, mode, wh, ri):
    """DRI/RSTn (decoder.rs:376-389, mcu.rs:386-419): with num_threads > 1 the restart segments of a baseline scan are
    decoded on several threads; same planes as the strictly serial walk."""
    hs, vs = MODES[mode]
    w, h = wh
    planes = jpeg_enc.small_planes(w, h, hs, vs, 3, seed=ri)
    data = jpeg_enc.encode_baseline(planes, synth.quant_tables(80), w, h, hs, vs, 3, restart=ri)
    serial, par = zj.Decoder(_opts(zj, 1)), zj.Decoder(_opts(zj, 4))
    _, a, info = serial.decode_coefficients(data)
    _, b, _ = par.decode_coefficients(data)
    mcus = ((w + 8 * hs - 1) // (8 * hs)) * ((h + 8 * vs - 1) // (8 * vs))
    assert info.restart_interval == ri
    assert serial.parallel_segments() == 0
    assert par.parallel_segments() == (mcus + ri - 1) // ri
    for c in range(3):
        assert np.array_equal(a[c], b[c]) and np.array_equal(a[c], planes[c])


def test_damaged_restart_structure_takes_the_serial_walk(zj, synth):
    """A missing or out-of-sequence RSTn, or a bad code inside a segment: the concurrent path steps aside and the
    serial walk produces the result (or the error text) it always did."""
    w, h, ri = 128, 64, 4
    planes = jpeg_enc.small_planes(w, h, 2, 2, 3, seed=9)
    data = bytearray(jpeg_enc.encode_baseline(planes, synth.quant_tables(80), w, h, 2, 2, 3, restart=ri))
    rst = [i for i in range(len(data) - 1) if data[i] == 0xFF and 0xD0 <= data[i + 1] <= 0xD7]
    assert len(rst) >= 3

    def both(blob):
        res = []
        for t in (1, 4):
            d = zj.Decoder(_opts(zj, t))
            try:
                _, pl, _ = d.decode_coefficients(bytes(blob))
                res.append(("ok", [p.copy() for p in pl], d.parallel_segments()))
            except zj.DecodeError as e:
                res.append(("err", (e.status, e.text), d.parallel_segments()))
        return res

    out_of_seq = bytearray(data)
    out_of_seq[rst[1] + 1] = 0xD5
    dropped = data[:rst[1]] + data[rst[1] + 2:]
    truncated = data[:rst[2] + 2] + b"\xff\xd9"
    for blob in (out_of_seq, dropped, truncated):
        s, p = both(blob)
        assert p[2] == 0 and s[0] == p[0]
        if s[0] == "ok":
            assert all(np.array_equal(x, y) for x, y in zip(s[1], p[1]))
        else:
            assert s[1] == p[1]


def test_pinned_plane_option_without_device(zj, synth):
    """pinned_planes falls back to heap planes when no HIP device can pin memory (CPU container)."""
    planes = jpeg_enc.small_planes(64, 48, 2, 2, 3, seed=3)
    data = jpeg_enc.encode_baseline(planes, synth.quant_tables(85), 64, 48, 2, 2, 3)
    o = zj.ZuneJpegOptions()
    o.pinned_planes = True
    _, got, _ = zj.Decoder(o).decode_coefficients(data)
    assert all(np.array_equal(g, p) for g, p in zip(got, planes))


# ---- the reference's short read of long DC symbols (src/bitstream.rs:278; zj_jpeg.cpp ref_dc_misread) ------------------
def _walked_blocks_equal(got, planes, rows, mcu_x, hs, vs):
    """coefficient planes equal on every MCU row the reference walks (rows[my] is not None)"""
    for c in range(3):
        h, v = (hs, vs) if c == 0 else (1, 1)
        g = np.array(got[c], np.int16).reshape(-1, mcu_x * h, 64)
        p = np.array(planes[c], np.int16).reshape(-1, mcu_x * h, 64)
        for my, n in enumerate(rows):
            if n is not None and not np.array_equal(g[my * v:(my + 1) * v], p[my * v:(my + 1) * v]):
                return False
    return True


def _noisy_jpeg(seed, subsampling, w=64, h=48):
    """quality 100, blocks of opposite brightness with full-amplitude pixel noise: DC differences of category 10 / 11
    (18 / 20-bit symbols with the standard tables) behind blocks whose last symbol is long"""
    import io
    from PIL import Image
    rng = np.random.default_rng(seed)
    base = (rng.integers(0, 2, (h // 8, w // 8, 1)) * 2 - 1) * rng.integers(60, 128, (h // 8, w // 8, 1))
    a = 128 + base.repeat(8, 0).repeat(8, 1).repeat(3, 2) + (rng.integers(0, 2, (h, w, 3)) * 2 - 1) * rng.integers(0, 127, (h, w, 3))
    b = io.BytesIO()
    Image.fromarray(np.clip(a, 0, 255).astype(np.uint8)).save(b, "JPEG", quality=100, subsampling=subsampling)
    return b.getvalue()


def test_reference_reads_long_dc_symbols_short_and_so_does_the_front_end(zj):
    """decode_dc refills only below 16 bits (src/bitstream.rs:278) although a DC symbol can have 17-27: with 16..26 bits left
    the reference decodes the code, gets zeros for the magnitude bits it does not hold, and then parses those bits AGAIN
    as the next symbol -- garbage from there on, but the reference's garbage.  The literal model of its reader
    (oracle/ref_walk.py decode_baseline_planes) and the front-end must agree coefficient for coefficient on every MCU
    row the reference walks -- on files where that happens and on files where it does not."""
    hit = clean = 0
    for seed in range(60):
        sub = [0, 2, 1][seed % 3]
        hs, vs = [(1, 1), (2, 2), (2, 1)][seed % 3]
        data = _noisy_jpeg(seed, sub)
        planes, short, rows = ref_walk.decode_baseline_planes(data)
        o = _opts(zj, 1)
        try:
            desc, got, info = zj.Decoder(o).decode_coefficients(data)
        except zj.DecodeError:
            assert short > 0, seed      # only a desynchronised stream may end in a bad code
            continue
        mcu_x = (64 + 8 * hs - 1) // (8 * hs)
        assert _walked_blocks_equal(got, planes, rows, mcu_x, hs, vs), (seed, short)
        hit += short > 0
        clean += short == 0
    assert hit >= 1 and clean >= 10, (hit, clean)


def test_short_dc_read_picks_up_stale_rotated_bits(zj, synth):
    """the reference's get_bits ROTATES aligned_buffer (src/bitstream.rs:394-402), so the magnitude bits it has
    handed out since the last real refill lie below the zeros that refill left, and a short DC read that reaches past those
    zeros is served stale magnitude bits, not zeros.  It takes a refill at bits_left near 32 (a gap of few zeros), a short
    code right behind it whose magnitude goes through get_bits (code + magnitude > 9 bits: not in the fast-AC table), no
    further refill, and a DC symbol far longer than what is left: hand-made Huffman tables (tools/jpeg_enc.py
    canonical_tables) with a 16-bit code for DC category 11 and 2..6-bit codes for (run, size 10) AC symbols build that
    on purpose.  The front-end must produce the literal reader model's coefficients -- and not the zeros-only ones -- in
    every block decoded before the end marker comes into the reader's view."""
    ac_len = {0xFA: 2, 0x0A: 2, 0xFF: 3, 0xEA: 4, 0x0F: 4, 0x00: 5, 0x3A: 5, 0x7A: 6, 0x01: 6, 0xF0: 7}
    dc_len = {0: 1, 11: 16, 2: 2, 6: 3}
    tabs = jpeg_enc.canonical_tables(dc_len, ac_len)
    zz = jpeg_enc.ZIGZAG
    stale_files = 0
    for seq in ([0xFA, 0xFA, 0xEA, 0x7A, 0x7A], [0xFA, 0xFA, 0x7A, 0xEA, 0x7A], [0xFA, 0x0A, 0xEA, 0xEA, 0xFF]):
        nb = 12                                   # one row of 12 blocks: the event in blocks 0/1, real data behind it
        p = np.zeros((nb, 64), np.int16)
        p[0, 0] = 40                              # category 6
        pos = 1
        for sym in seq:                           # block 0 ends at coefficient 63 without an EOB
            pos += sym >> 4
            sz = sym & 15
            p[0, zz[pos]] = (1 << sz) - 1 if sz < 15 else 32767   # all-ones magnitudes: the stale bits are ones
            pos += 1
        p[1, 0] = p[0, 0] + 1500                  # category 11 behind it: 16 + 11 bits, the reader holds 17..23
        for b in range(2, nb):                    # plain blocks: something real to go on parsing (garbled, but real bits)
            p[b, 0] = p[b - 1, 0] + (40 if b % 2 else -40)
            p[b, zz[1]], p[b, zz[2]] = 600, -700
        data = jpeg_enc.encode_baseline([p.reshape(-1)], [synth.quant_tables(85)[0]], 8 * nb, 8, 1, 1, 1, tables=tabs)
        try:
            want, short, rows = ref_walk.decode_baseline_planes(data)
        except ValueError:
            want = None
        st = dict(ref_walk.last_stats)
        assert st["short"] >= 1 and st["stale"] >= 1, (seq, st)
        stale_files += 1
        try:
            desc, got, info = zj.Decoder(_opts(zj, 1)).decode_coefficients(data)
        except zj.DecodeError:
            assert want is None, seq              # only if the literal reader ends in a bad code as well
            continue
        assert want is not None, seq
        g = np.array(got[0], np.int16).reshape(-1, 64)
        w_ = want[0].reshape(-1, 64)
        upto = st["first_marker_block"] if st["first_marker_block"] is not None else nb
        assert upto >= 2, (seq, st)
        assert np.array_equal(g[:upto], w_[:upto]), (seq, st, [int(x) for x in g[:upto, 0]], [int(x) for x in w_[:upto, 0]])
        assert int(g[1, 0]) != 40 + 1024          # 1064 = what zeros below the held bits would give (rounds 2-3)
    assert stale_files == 3
    # the WHOLE 64-bit register is history.  An AC refill at bits_left == 32 makes it 64; symbols worth 29 bits
    # follow without another refill, then a 4 + 15-bit symbol that lands on coefficient 63 (bits_left 35 -> 16), then the
    # 16-bit DC code of category 11 (no refill: 16 bits are held): the 11 magnitude bits are read with NOTHING left, all of
    # them from the rotated history (ref_dc_misread: nhist == 64).  Sequences found by a search over this table's symbols;
    # block 1's difference is chosen so that the literal reader's parse survives to the end of the scan.
    seq = [0xFA, 0xFA, 0xFA, 0x0A, 0x0A, 0x0A, 0x0A, 0x0A, 0x0A, 0x0A, 0x01, 0x3A, 0x01, 0x01, 0x0F]
    for diff, dc1 in ((1556, -1984), (-1529, -763)):
        nb = 12
        p = np.zeros((nb, 64), np.int16)
        pos = 1                                   # block 0: DC 0 (category 0), then the sequence up to coefficient 63
        for sym in seq:
            pos += sym >> 4
            sz = sym & 15
            p[0, zz[pos]] = (1 << sz) - 1 if sz < 15 else 32767
            pos += 1
        assert pos == 64
        p[1, 0] = diff
        for b in range(2, nb):
            p[b, 0] = p[b - 1, 0] + (40 if b % 2 else -40)
            p[b, zz[1]], p[b, zz[2]] = 600, -700
        data = jpeg_enc.encode_baseline([p.reshape(-1)], [synth.quant_tables(85)[0]], 8 * nb, 8, 1, 1, 1, tables=tabs)
        want, short, rows = ref_walk.decode_baseline_planes(data)
        st = dict(ref_walk.last_stats)
        assert st["short"] >= 1 and st["stale"] >= 1 and st["first_marker_block"] is None, st
        desc, got, info = zj.Decoder(_opts(zj, 1)).decode_coefficients(data)
        g = np.array(got[0], np.int16).reshape(-1, 64)
        w_ = want[0].reshape(-1, 64)
        assert np.array_equal(g, w_), (diff, [int(x) for x in g[:, 0]], [int(x) for x in w_[:, 0]])
        assert int(g[1, 0]) == dc1 and dc1 != -2047   # -2047 = zeros for all eleven bits: what the mask gave at nhist == 64


# ---- the reference's fast-AC table, followed to the letter (round 6; src/huffman.rs:204-251, src/bitstream.rs:339-347) ----
def _hand_made_scan(tabs, blocks, synth, width_blocks=None):
    """One-component baseline file from symbol lists: blocks = [(dc_diff, [(symbol, value or None), ...]), ...], emitted
    verbatim (no EOB unless listed) -- streams no encoder writes: runs past coefficient 63, size-0 symbols with a run."""
    nb = len(blocks) if width_blocks is None else width_blocks
    out = jpeg_enc._headers(8 * nb, 8 * (len(blocks) // nb), 1, 1, 1, [synth.quant_tables(85)[0]], False, 0, tabs)
    out += jpeg_enc._sos([0], 0, 63, 0, 0)
    bw = jpeg_enc.BitWriter()
    for diff, syms in blocks:
        s = jpeg_enc._nbits(diff)
        bw.put(*tabs["dc"]["codes"][s])
        if s:
            bw.put(diff if diff >= 0 else diff + (1 << s) - 1, s)
        for sym, v in syms:
            bw.put(*tabs["ac"]["codes"][sym])
            if sym & 15:
                bw.put(v if v >= 0 else v + (1 << (sym & 15)) - 1, sym & 15)
    bw.flush()
    return bytes(out) + bytes(bw.out) + b"\xff\xd9"


_DC_LEN = {0: 2, 1: 3, 2: 3, 3: 3, 4: 3, 5: 3, 6: 4, 7: 5, 8: 6, 9: 7, 10: 8, 11: 9}


def _model_and_front_end(zj, data, nblocks):
    want, short, rows = ref_walk.decode_baseline_planes(data)
    desc, got, info = zj.Decoder(_opts(zj, 1)).decode_coefficients(data)
    g = np.array(got[0], np.int16).reshape(-1, 64)[:nblocks]
    w = want[0].reshape(-1, 64)[:nblocks]
    return g, w


@pytest.mark.parametrize("reps", [1, 300])  # 300: tens of KB of scan, decode_mcus_v2 carries them (the last 4 KB go block by block)
def test_fast_ac_values_keep_six_bits(zj, synth, reps):
    """The reference packs a fast-AC value as `k << 10` into an i16: a size of 6..8 behind a code short enough for the fast
    table (code + size <= 9) comes back as its low six bits, sign-extended -- +32 reads as -32, 64 as 0.  Standard tables
    never get there (size 6 has a 7-bit code); hand-made ones do, and the front-end must give the reference's values."""
    ac_len = {0x08: 1, 0x07: 2, 0x06: 3, 0x00: 4, 0x01: 5, 0x16: 6, 0x05: 7, 0xF0: 8}
    tabs = jpeg_enc.canonical_tables(_DC_LEN, ac_len)
    zz = jpeg_enc.ZIGZAG
    vals = [32, -32, 33, 63, -63, 64, -64, 100, 127, -127, -128, 128, 255, -255, 31, -31, 17]
    blocks = []
    for i in range(0, len(vals), 3):
        syms = []
        for v in vals[i:i + 3]:
            syms.append((jpeg_enc._nbits(v), v))        # run 0: sizes 6, 7, 8 behind 3-, 2-, 1-bit codes; size 5 behind 7 bits
        syms.append((0x16, 45))                         # run 1, size 6 behind a 6-bit code: 12 bits, the general path
        syms.append((0x00, None))
        blocks.append((10 * (i + 1), syms))
    npat = len(blocks)
    data = _hand_made_scan(tabs, blocks * reps, synth, npat)
    g, w = _model_and_front_end(zj, data, npat * reps)
    assert np.array_equal(g, w), (g[:, :12], w[:, :12])
    blocks = blocks[:npat]
    flat = [int(g[b, zz[1 + j]]) for b in range(len(blocks)) for j in range(len(vals[3 * b:3 * b + 3]))]
    expect = [((v & 63) ^ 32) - 32 if -128 <= v <= 127 and jpeg_enc._nbits(v) >= 6 else v for v in vals]
    assert flat == expect, (flat, expect)
    assert flat[0] == -32 and flat[5] == 0 and flat[11] == 128 and flat[14] == 31  # cut / cut / too wide for the table / small
    assert all(int(g[b, zz[5]]) == 45 for b in range(len(blocks) - 1))            # the general path keeps its value


@pytest.mark.parametrize("reps", [1, 300])
def test_runs_past_coefficient_63(zj, synth, reps):
    """A damaged stream can push the zig-zag position past 63.  The reference's general path then writes at pos & 63
    (src/bitstream.rs:359) -- over a low coefficient -- and its fast path at min(pos, 63) (:343); both end the block."""
    ac_len = {0x01: 2, 0xF1: 3, 0x00: 3, 0xFA: 4, 0x31: 4, 0xF0: 5, 0x0A: 6}
    tabs = jpeg_enc.canonical_tables(_DC_LEN, ac_len)
    zz = jpeg_enc.ZIGZAG
    walk = [(0xF1, 1), (0xF1, -1), (0xF1, 1)]          # k = 1 -> 49 with coefficients at 16, 32, 48
    fill = [(0x01, 1)] * 8                              # ... -> 57, coefficients at 49..56
    blocks = [(5, walk + fill + [(0xF1, -1)]),          # fast path (3 + 1 bits): 57 + 15 = 72 -> written at 63
              (7, walk + fill + [(0xFA, 700)]),         # general path (4 + 10 bits): 72 & 63 = 8 -> over zig-zag 8
              (-3, walk + fill + [(0x31, 1)]),          # 57 + 3 = 60: nothing special, then an EOB
              (2, [(0x01, -1), (0x00, None)])]
    blocks[2][1].append((0x00, None))
    blocks += [(1, [(0x01, 1)] * 20 + [(0x00, None)])] * 8  # (the reference stops decoding once EOI comes into its reader's view)
    data = _hand_made_scan(tabs, blocks * reps, synth, len(blocks))
    g, w = _model_and_front_end(zj, data, len(blocks) * reps)
    assert np.array_equal(g, w), (g, w)
    assert int(g[0, 63]) == -1 and int(g[0, zz[8]]) == 0
    assert int(g[1, zz[8]]) == 700 and int(g[1, 63]) == 0
    assert int(g[2, zz[60]]) == 1 and int(g[3, zz[1]]) == -1 and int(g[3, 0]) == 11


@pytest.mark.parametrize("reps", [1, 300])
def test_size_zero_symbols_with_a_run(zj, synth, reps):
    """(run 1..14, size 0) is not a baseline symbol.  Behind a code of up to 9 bits the reference's fast table treats it like
    ZRL -- it skips run + 1 coefficients (src/huffman.rs:217-233) -- and behind a longer code the general path ends the
    block (src/bitstream.rs:365-367).  Both, in one file."""
    ac_len = {0x01: 2, 0x30: 3, 0x00: 3, 0x02: 4, 0xF0: 5, 0x50: 10, 0x11: 6}
    tabs = jpeg_enc.canonical_tables(_DC_LEN, ac_len)
    zz = jpeg_enc.ZIGZAG
    blocks = [(4, [(0x01, 1), (0x30, None), (0x02, -3), (0x00, None)]),      # 1 at k=1, skip 4 (k = 2..5), -3 at k = 6
              (1, [(0x01, -1), (0x50, None)]),                               # the 10-bit code ends the block ...
              (1, [(0x02, 2), (0xF0, None), (0x11, 1), (0x00, None)]),       # ... so this is the next block: 2 at 1, 1 at 19
              (-6, [(0x30, None), (0x30, None), (0x01, 1), (0x00, None)])]   # k = 1 -> 5 -> 9
    blocks += [(1, [(0x01, 1)] * 20 + [(0x00, None)])] * 8  # (the reference stops decoding once EOI comes into its reader's view)
    data = _hand_made_scan(tabs, blocks * reps, synth, len(blocks))
    g, w = _model_and_front_end(zj, data, len(blocks) * reps)
    assert np.array_equal(g, w), (g, w)
    assert int(g[0, zz[1]]) == 1 and int(g[0, zz[6]]) == -3 and np.count_nonzero(g[0]) == 3
    assert int(g[1, zz[1]]) == -1 and np.count_nonzero(g[1]) == 2
    assert int(g[2, zz[1]]) == 2 and int(g[2, zz[19]]) == 1 and int(g[2, 0]) == 6
    assert int(g[3, zz[9]]) == 1 and np.count_nonzero(g[3]) == 1  # (the DC is 6 - 6 = 0)


def _first_scan_only(data):
    """a progressive file cut behind its first scan (+ EOI): what the front-end holds after that scan alone"""
    sos = data.index(b"\xff\xda")
    p = sos + 2 + int.from_bytes(data[sos + 2:sos + 4], "big")
    while not (data[p] == 0xFF and data[p + 1] not in (0x00, 0xFF) and not 0xD0 <= data[p + 1] <= 0xD7):
        p += 1
    return data[:p] + b"\xff\xd9"


def test_reference_short_dc_reads_in_progressive_dc_scans(zj):
    """A progressive DC scan is a run of decode_dc calls and nothing else, so bits_left wanders through 16..47 and every DC
    symbol longer than 16 bits has a fair chance of being read short (src/bitstream.rs:278, :407-415).  libjpeg writes
    progressive files with optimised tables, so the long symbols are the RARE categories: mild images with a few
    black / white blocks.  The DC scan alone, through the front-end and through the literal reader model
    (oracle/ref_walk.py): the same coefficients, or -- when the desynchronised stream runs into a code that does not
    exist -- an error from both."""
    import io
    from PIL import Image
    hit = clean = 0
    for seed in range(60):
        rng = np.random.default_rng(seed)
        w, h = 160, 96
        nb = (h // 8) * (w // 8)
        vals = rng.integers(100, 156, nb)
        k = rng.choice(nb, size=3, replace=False)
        vals[k] = rng.choice([0, 255], size=k.size)
        a = vals.reshape(h // 8, w // 8, 1).repeat(8, 0).repeat(8, 1).repeat(3, 2).astype(np.uint8)
        b = io.BytesIO()
        Image.fromarray(a).save(b, "JPEG", quality=100, subsampling=[0, 2, 1][seed % 3], progressive=True)
        data = _first_scan_only(b.getvalue())
        try:
            want, short = ref_walk.decode_progressive_dc_first(data)
        except ValueError:          # "bad Huffman code": the reference fails with DecodeErrors::HuffmanDecode
            want, short = None, 1
        o = zj.ZuneJpegOptions()
        o.num_threads = 1
        try:
            desc, got, info = zj.Decoder(o).decode_coefficients(data)
        except zj.DecodeError as e:
            assert want is None and "Huffman" in str(e), (seed, str(e))
            hit += 1
            continue
        assert want is not None, seed
        assert info.progressive == 1 and info.scans == 1
        for c in range(3):
            g = np.array(got[c], np.int16).reshape(want[c].shape[0], want[c].shape[1], 64)
            assert np.array_equal(g[:, :, 0], want[c]), (seed, c, short)
            assert not g[:, :, 1:].any()
        hit += short > 0
        clean += short == 0
    assert hit >= 3 and clean >= 10, (hit, clean)


def test_fill_bytes_in_front_of_a_stuffed_zero_read_like_the_reference(zj):
    """FF FF 00 inside a scan: the reference's refill appends the first 0xFF, skips the fill bytes, finds a zero -- no marker --
    and goes on: the sequence reads like FF 00, one data byte 0xFF (src/bitstream.rs:183-211).  Until round 6 the front-end
    appended a ZERO byte there.  The literal model of the reference's reader (oracle/ref_walk.py) decides."""
    checked = 0
    for seed in range(40):
        sub = [0, 2, 1][seed % 3]
        hs, vs = [(1, 1), (2, 2), (2, 1)][seed % 3]
        data = bytearray(_noisy_jpeg(seed, sub))
        sos = data.index(b"\xff\xda")
        start = sos + 2 + int.from_bytes(data[sos + 2:sos + 4], "big")
        pairs = [i for i in range(start, len(data) - 3) if data[i] == 0xFF and data[i + 1] == 0x00]
        if not pairs:
            continue
        at = pairs[len(pairs) // 2]
        data[at:at] = b"\xff" * (1 + seed % 3)        # one to three fill bytes in front of the stuffed pair
        data = bytes(data)
        planes, short, rows = ref_walk.decode_baseline_planes(data)
        try:
            desc, got, info = zj.Decoder(_opts(zj, 1)).decode_coefficients(data)
        except zj.DecodeError:
            assert short > 0, seed
            continue
        mcu_x = (64 + 8 * hs - 1) // (8 * hs)
        assert _walked_blocks_equal(got, planes, rows, mcu_x, hs, vs), (seed, short)
        # and it is the file without the fill bytes, coefficient for coefficient
        clean = bytearray(data)
        del clean[at:at + 1 + seed % 3]
        _, want, _ = zj.Decoder(_opts(zj, 1)).decode_coefficients(bytes(clean))
        assert all(np.array_equal(a, b) for a, b in zip(got, want)), seed
        checked += 1
    assert checked >= 10, checked


def test_damaged_scans_block_for_block_like_the_literal_model(zj):
    """A miniature of tools/ref_walk_soak.py: small files with and without restart intervals, damaged five ways -- among them
    intervals that run out of data, where the reference decodes what its rotating aligned_buffer holds behind the marker
    (src/bitstream.rs:254-258,394-402) and the front-end hands the scan to its literal restatement (zj_jpeg.cpp
    scan_baseline_literal) -- against oracle/ref_walk.py: every block of every MCU row the reference walks, or an error on both
    sides."""
    import io
    from PIL import Image
    rng = np.random.default_rng(2024)
    compared = both_raise = ran_dry = 0
    for f in range(60):
        w, h = int(rng.integers(3, 12)) * 8, int(rng.integers(3, 9)) * 8
        sub = int(rng.integers(0, 3))
        hs, vs = [(1, 1), (2, 1), (2, 2)][sub]
        base = (rng.integers(0, 2, (h // 8 + 1, w // 8 + 1, 1)) * 2 - 1) * rng.integers(0, 128, (h // 8 + 1, w // 8 + 1, 1))
        img = 128 + base.repeat(8, 0).repeat(8, 1)[:h, :w].repeat(3, 2) + (rng.integers(0, 2, (h, w, 3)) * 2 - 1) * rng.integers(0, int(rng.integers(1, 127)), (h, w, 3))
        kw = {"restart_marker_blocks": int(rng.integers(1, 9))} if f % 2 else {}
        b = io.BytesIO()
        Image.fromarray(np.clip(img, 0, 255).astype(np.uint8)).save(b, "JPEG", quality=int(rng.choice([50, 90, 100])), subsampling=sub, **kw)
        data = b.getvalue()
        sos = data.index(b"\xff\xda")
        start = sos + 2 + int.from_bytes(data[sos + 2:sos + 4], "big")
        for trial in range(5):
            d = bytearray(data)
            at = int(rng.integers(start, len(d) - 2))
            if trial == 0:
                d[at] ^= 1 << int(rng.integers(0, 8))
            elif trial == 1:
                del d[at:at + int(rng.integers(1, 5))]
            elif trial == 2:
                d[at:at] = bytes(rng.integers(0, 255, int(rng.integers(1, 12)), dtype=np.uint8))
            elif trial == 3:
                d[at:at] = bytes([0xFF, 0xD0 + int(rng.integers(0, 8))])
            else:
                d[at:at] = b"\xff\xd9"
            d = bytes(d)
            if any(d[i] == 0xFF and d[i + 1] not in (0x00, 0xFF, 0xD9) and not 0xD0 <= d[i + 1] <= 0xD7 for i in range(start, len(d) - 1)):
                continue   # (a header's or an unknown marker in the scan: the model does not follow those)
            try:
                planes, short, rows = ref_walk.decode_baseline_planes(d)
                want_error = False
            except (ValueError, IndexError):
                want_error = True
            try:
                _, got, info = zj.Decoder(_opts(zj, 1)).decode_coefficients(d)
                got_error = False
            except zj.DecodeError:
                got_error = True
            assert want_error == got_error, (f, trial)
            if want_error:
                both_raise += 1
                continue
            mcu_x = (w + 8 * hs - 1) // (8 * hs)
            assert _walked_blocks_equal(got, planes, rows, mcu_x, hs, vs), (f, trial, w, h, sub, kw)
            compared += 1
            ran_dry += ref_walk.last_stats["first_marker_block"] is not None and trial in (1, 3, 4)
    assert compared > 200 and ran_dry > 40, (compared, both_raise, ran_dry)
