"""CPU: the baseline walker's stretch decoder (zj_jpeg.cpp decode_mcus_v2, round 6) against the block-at-a-time decoder it
replaced on the hot path (ZJ_WALKER_V1=1 keeps every block on decode_block_baseline): same planes, same status, same error
text -- on intact files of every sampling mode (Pillow's tables and optimised ones, with and without restart intervals, one
thread and restart segments on several) and on the same files with bits flipped inside the scan, where what comes out is
the reference's garbage (runs past coefficient 63, bad codes, short DC reads) and must be the same garbage.  The files are
large enough (tens of KB of scan) for the stretch decoder to carry them: it leaves the scan's last 4 KB to the old code."""
import importlib
import io
import os

import numpy as np
import pytest


@pytest.fixture(scope="module")
def zj():
    return importlib.import_module("zune-jpeg_amd")


def _jpeg(seed, w, h, subsampling, quality, optimize, restart_rows):
    from PIL import Image, ImageFile
    ImageFile.MAXBLOCK = max(ImageFile.MAXBLOCK, 4 * w * h)  # (optimize=True needs the whole file in one encoder buffer)
    rng = np.random.default_rng(seed)
    small = rng.integers(0, 256, (max(2, h // 16), max(2, w // 16), 3), dtype=np.uint8)
    img = Image.fromarray(small, "RGB").resize((w, h), Image.BICUBIC)
    noise = rng.integers(-24, 25, (h, w, 3), dtype=np.int16) * (1 + seed % 3)
    img = Image.fromarray(np.clip(np.asarray(img).astype(np.int16) + noise, 0, 255).astype(np.uint8), "RGB")
    b = io.BytesIO()
    kw = {"restart_marker_rows": restart_rows} if restart_rows else {}
    img.save(b, "JPEG", quality=quality, subsampling=subsampling, optimize=optimize, **kw)
    return b.getvalue()


def _decode(zj, data, threads, v1, par=False):
    """v1: every block through the block-at-a-time decoder.  par: scans without restart markers may be entered at one point per
    thread (scan_baseline_parallel), even small ones; otherwise that path is off, so that `threads` only means restart segments."""
    if v1:
        os.environ["ZJ_WALKER_V1"] = "1"
    else:
        os.environ.pop("ZJ_WALKER_V1", None)
    if par:
        os.environ["ZJ_PAR_MIN_CHUNK"] = "1024"
        os.environ.pop("ZJ_PAR_SCAN", None)
    else:
        os.environ["ZJ_PAR_SCAN"] = "off"
    try:
        o = zj.ZuneJpegOptions()
        o.num_threads = threads
        dec = zj.Decoder(o)
        try:
            _, planes, info = dec.decode_coefficients(data)
            if par:
                _PAR_MCUS.append(dec.parallel_mcus())
            return ("ok", [p.tobytes() for p in planes], (info.width, info.height, info.scans), dec.parallel_segments())
        except zj.DecodeError as e:
            return ("error", str(e))
        finally:
            dec.close()
    finally:
        os.environ.pop("ZJ_WALKER_V1", None)
        os.environ.pop("ZJ_PAR_MIN_CHUNK", None)
        os.environ.pop("ZJ_PAR_SCAN", None)


_PAR_MCUS = []


CASES = [(seed, w, h, sub, q, opt, rst)
         for seed, (w, h) in enumerate([(448, 336), (512, 320), (333, 477), (1024, 160)])
         for sub in (0, 1, 2)
         for q, opt, rst in ((35, False, 0), (90, True, 0), (100, False, 2), (97, True, 1))]


@pytest.mark.parametrize("case", CASES, ids=[f"{c[1]}x{c[2]}-s{c[3]}-q{c[4]}{'-opt' if c[5] else ''}{'-rst' if c[6] else ''}" for c in CASES])
def test_stretch_decoder_equals_block_decoder(zj, case):
    seed, w, h, sub, q, opt, rst = case
    data = _jpeg(seed, w, h, sub, q, opt, rst)
    assert len(data) > 12000  # the stretch decoder gets most of the scan
    for threads in (1, 4) if rst else (1,):
        a = _decode(zj, data, threads, v1=False)
        b = _decode(zj, data, threads, v1=True)
        # (quality-100 noise can END in an error even undamaged: a long DC symbol read short desynchronises the reference's
        # reader, tests/test_jpeg_frontend.py -- then both must report the same one)
        assert a == b and (a[0] == "ok" or q == 100), (case, threads, a[0], b[0])
    # the same file with damage inside the scan: flipped bits, a byte made 0xFF, a few bytes dropped
    rng = np.random.default_rng(1000 + seed)
    sos = data.index(b"\xff\xda")
    start = sos + 2 + int.from_bytes(data[sos + 2:sos + 4], "big")
    differing = 0
    for trial in range(12):
        d = bytearray(data)
        at = int(rng.integers(start + 16, len(d) - 5000))
        kind = trial % 3
        if kind == 0:
            d[at] ^= 1 << int(rng.integers(0, 8))
        elif kind == 1:
            d[at] = 0xFF
            d[at + 1] = [0x00, 0xD9, 0xD3, 0x17][trial % 4]
        else:
            del d[at:at + int(rng.integers(1, 4))]
        d = bytes(d)
        a = _decode(zj, d, 1, v1=False)
        b = _decode(zj, d, 1, v1=True)
        assert a == b, (case, trial, kind, a[0], b[0], a[1] if a[0] == "error" else "", b[1] if b[0] == "error" else "")
        differing += a[0] == "error" or a[1] != _decode(zj, data, 1, v1=False)[1]
    assert differing >= 6  # the damage did reach the decoder


def test_grayscale_files_through_the_stretch_decoder(zj):
    """one component: one block per MCU, no chroma tables"""
    from PIL import Image
    rng = np.random.default_rng(77)
    for w, h, q, rst in ((640, 480, 85, 0), (1001, 333, 95, 3)):
        small = rng.integers(0, 256, (h // 16, w // 16), dtype=np.uint8)
        img = Image.fromarray(small, "L").resize((w, h), Image.BICUBIC)
        img = Image.fromarray(np.clip(np.asarray(img).astype(np.int16) + rng.integers(-20, 21, (h, w), dtype=np.int16), 0, 255).astype(np.uint8), "L")
        b = io.BytesIO()
        img.save(b, "JPEG", quality=q, **({"restart_marker_rows": rst} if rst else {}))
        data = b.getvalue()
        assert len(data) > 12000
        for threads in (1, 4):
            a = _decode(zj, data, threads, v1=False)
            c = _decode(zj, data, threads, v1=True)
            assert a[0] == "ok" and a == c and len(a[1]) == 1, (w, h, threads)


@pytest.mark.parametrize("case", [c for c in CASES if not c[6]], ids=[f"{c[1]}x{c[2]}-s{c[3]}-q{c[4]}{'-opt' if c[5] else ''}" for c in CASES if not c[6]])
def test_scans_without_restart_markers_on_several_threads(zj, case):
    """scan_baseline_parallel: the scan entered at one point per thread, the threads falling into step with the true symbol
    sequence -- same planes, same status, same error text as the serial walk, on intact files and on damaged ones (where it
    must either drop the attempt or arrive at the same garbage)."""
    seed, w, h, sub, q, opt, rst = case
    data = _jpeg(seed, w, h, sub, q, opt, rst)
    del _PAR_MCUS[:]
    ref = _decode(zj, data, 1, v1=False)
    for threads in (2, 3, 4, 7):
        got = _decode(zj, data, threads, v1=False, par=True)
        assert got[:3] == ref[:3] if got[0] == "ok" else got == ref, (case, threads)
    if ref[0] == "ok":
        assert sum(1 for m in _PAR_MCUS if m > 0) >= 3, _PAR_MCUS   # the path was taken, not just fallen back from
    rng = np.random.default_rng(2000 + seed)
    sos = data.index(b"\xff\xda")
    start = sos + 2 + int.from_bytes(data[sos + 2:sos + 4], "big")
    for trial in range(10):
        d = bytearray(data)
        at = int(rng.integers(start + 16, len(d) - 5000))
        kind = trial % 3
        if kind == 0:
            d[at] ^= 1 << int(rng.integers(0, 8))
        elif kind == 1:
            d[at] = 0xFF
            d[at + 1] = [0x00, 0xD9, 0xD3, 0x17][trial % 4]
        else:
            del d[at:at + int(rng.integers(1, 4))]
        d = bytes(d)
        a = _decode(zj, d, 1, v1=False)
        b = _decode(zj, d, 4, v1=False, par=True)
        assert (a[:3] == b[:3]) if a[0] == "ok" else a == b, (case, trial, kind, a[0], b[0])


@pytest.mark.parametrize("sub", [0, 2])
def test_parallel_scan_bridges_flat_areas(zj, sub):
    """A band of one colour in the middle of the picture: runs of identical two-symbol MCUs, where a reader that enters out of
    step stays out of step.  The stitching walks `patience` MCUs into it (ZJ_PAR_PATIENCE; the product: 512 or 1/64 of the
    picture), then the calling thread decodes the rest of that chunk for real and the stitching goes on behind it: the chunks
    below the band are in step again -- same planes, and the whole region is covered whatever the patience."""
    from PIL import Image, ImageFile
    ImageFile.MAXBLOCK = max(ImageFile.MAXBLOCK, 1 << 24)
    rng = np.random.default_rng(77 + sub)
    w, h = 640, 960
    arr = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
    arr[200:760] = (90, 140, 200)
    b = io.BytesIO()
    Image.fromarray(arr, "RGB").save(b, "JPEG", quality=80, subsampling=sub)
    data = b.getvalue()
    ref = _decode(zj, data, 1, v1=False)
    assert ref[0] == "ok"
    total = ((w + 15) // 16 if sub else (w + 7) // 8) * ((h + 15) // 16 if sub == 2 else (h + 7) // 8)
    for patience in ("2", "40", None):
        for threads in (3, 4, 6):
            if patience:
                os.environ["ZJ_PAR_PATIENCE"] = patience
            try:
                del _PAR_MCUS[:]
                got = _decode(zj, data, threads, v1=False, par=True)
            finally:
                os.environ.pop("ZJ_PAR_PATIENCE", None)
            assert got[:3] == ref[:3], (patience, threads)
            # everything but the scan's last 8 KB (which stay with the serial walk) went through the parallel path
            assert 0.6 * total < _PAR_MCUS[0] <= total, (patience, threads, _PAR_MCUS, total)


def test_a_decoder_with_helper_threads_survives_fork(zj):
    """The decoder's helper threads (Crew) do not exist in the child of a fork(): the child's first parallel region starts its
    own, and closing the decoder there -- with or without having used it -- does not wait for threads that are not there."""
    data = _jpeg(3, 1024, 768, 2, 90, False, 0)
    os.environ["ZJ_PAR_MIN_CHUNK"] = "1024"
    try:
        o = zj.ZuneJpegOptions()
        o.num_threads = 4
        decs = [zj.Decoder(o), zj.Decoder(o)]
        ref = None
        for d in decs:
            _, planes, _ = d.decode_coefficients(data)
            assert d.parallel_mcus() > 0
            ref = [p.tobytes() for p in planes]
        r, w = os.pipe()
        pid = os.fork()
        if pid == 0:   # the child: one decoder used again, one only closed
            code = 1
            try:
                _, planes, _ = decs[0].decode_coefficients(data)
                same = [p.tobytes() for p in planes] == ref and decs[0].parallel_mcus() > 0
                decs[0].close()
                decs[1].close()
                os.write(w, b"ok" if same else b"different")
                code = 0
            finally:
                os._exit(code)
        os.close(w)
        import select
        ready, _, _ = select.select([r], [], [], 60)
        msg = os.read(r, 64) if ready else b"timeout"
        os.close(r)
        if not ready:
            os.kill(pid, 9)
        os.waitpid(pid, 0)
        assert msg == b"ok", msg
        for d in decs:   # the parent's decoders are unaffected
            _, planes, _ = d.decode_coefficients(data)
            assert [p.tobytes() for p in planes] == ref
            d.close()
    finally:
        os.environ.pop("ZJ_PAR_MIN_CHUNK", None)


def test_damaged_restart_intervals_go_to_the_serial_walk(zj):
    """Restart segments side by side (num_threads > 1) are only the reference's output where every interval ends at its marker.
    Bytes inserted into an interval: the reference decodes the NEXT interval out of what is left, predictors and all.  An
    header's marker in the last interval, the interval being full: handle_restart() says "Marker found in bitstream"; a marker
    the reference has no name for: its refill says "Unknown marker 0xFF17".  All used to be decoded segment by segment to other
    results (rounds 4-5; found by tools/stream_soak.py in round 6)."""
    # 28 x 22 MCUs, an interval per MCU row: every interval is full.  (An EVEN number of MCU rows: with sub-sampled chroma the
    # reference walks its rows in pairs and never reads an odd last one -- whatever is wrong in there is not an error.)
    data = _jpeg(11, 448, 352, 2, 90, False, 1)
    ref = _decode(zj, data, 1, v1=False)
    par = _decode(zj, data, 4, v1=False)
    assert ref[0] == "ok" and par[:3] == ref[:3] and par[3] == 22   # (intact: 22 segments decoded side by side)
    rst = [i for i in range(len(data) - 1) if data[i] == 0xFF and 0xD0 <= data[i + 1] <= 0xD7]
    assert len(rst) == 21
    rng = np.random.default_rng(5)
    seen = set()
    for trial in range(24):
        d = bytearray(data)
        if trial % 3 == 0:      # bytes inserted in mid-interval
            k = int(rng.integers(0, len(rst) - 1))
            at = int(rng.integers(rst[k] + 40, rst[k + 1] - 40))
            d[at:at] = bytes(rng.integers(1, 255, int(rng.integers(2, 40)), dtype=np.uint8))
        elif trial % 3 == 1:    # a marker in the last interval: one the reference has no name for, or a header's
            at = int(rng.integers(rst[-1] + 40, len(d) - 40))
            d[at:at + 2] = b"\xff\x17" if trial % 2 else b"\xff\xc4"
        else:                   # a restart marker that is not due, in mid-interval
            k = int(rng.integers(0, len(rst) - 1))
            at = int(rng.integers(rst[k] + 40, rst[k + 1] - 40))
            d[at:at + 2] = bytes([0xFF, 0xD0 + int(rng.integers(0, 8))])
        d = bytes(d)
        a = _decode(zj, d, 1, v1=False)
        b = _decode(zj, d, 4, v1=False)
        assert (a[:3] == b[:3]) if a[0] == "ok" else a == b, (trial, a[0], b[0], a[1] if a[0] == "error" else "", b[1] if b[0] == "error" else "")
        seen.add(a[0] if a[0] == "ok" else a[1])
    assert any("Marker found in bitstream" in s for s in seen), seen     # (FF C4 at a restart boundary: src/mcu.rs:409-414)
    assert any("Unknown marker 0xFF17" in s for s in seen), seen         # (src/bitstream.rs:199-206: the refill itself gives up)


def test_an_odd_last_mcu_row_the_reference_never_reads_cannot_fail(zj):
    """With horizontally sub-sampled chroma the reference walks its MCU rows in pairs (src/mcu.rs:145-152,225-231): the last row
    of an odd number is never decoded -- its pixels stay zero -- so damage in its data is not an error (it was, until
    tools/ref_walk_soak.py compared the serial walk with the literal model of the reference in round 6)."""
    for sub, h in ((2, 336), (1, 168)):              # 4:2:0: 21 MCU rows of 16; 4:2:2: 21 MCU rows of 8
        data = _jpeg(12, 448, h, sub, 90, False, 0)
        ref = _decode(zj, data, 1, v1=False)
        assert ref[0] == "ok"
        tail = len(data) - (len(data) - data.index(b"\xff\xda")) // 42   # well inside the last of 21 rows
        seen_ok = 0
        rng = np.random.default_rng(3)
        for trial in range(12):
            d = bytearray(data)
            at = int(rng.integers(tail, len(d) - 8))
            if trial % 2:
                d[at:at + 2] = b"\xff\x17"           # "Unknown marker 0xFF17" anywhere the reference reads
            else:
                d[at:at + 6] = bytes(rng.integers(1, 255, 6, dtype=np.uint8))
            for threads in (1, 4):
                got = _decode(zj, bytes(d), threads, v1=False)
                assert got[0] == "ok", (sub, trial, threads, got)
                seen_ok += 1
                # the rows the reference walks are what they were
                for a, b, rows in zip(got[1], ref[1], (20 * (2 if sub == 2 else 1), 20, 20)):
                    bw = len(b) // 128 // (21 * (2 if (sub == 2 and rows == 40) else 1))
                    assert a[: rows * bw * 128] == b[: rows * bw * 128], (sub, trial, threads)
        assert seen_ok == 24
