#!/usr/bin/env python3
"""CPU soak of the front-end's SERIAL walk against the literal model of the reference's reader and MCU loop (oracle/ref_walk.py):
small random baseline files (the model is pure Python), with and without restart intervals, intact and damaged -- bit flips,
deleted and inserted bytes, FF 00 / FF FF 00 / RSTn / EOI inserted -- coefficient for coefficient on every MCU row the reference
walks, or an error on both sides.  Variants that bring a marker into the scan which the model does not follow (anything but
RSTn and EOI: the reference parses headers in mid-scan or gives up in its refill) are skipped.
  python tools/ref_walk_soak.py [--seconds 120] [--seed 1]
"""
import argparse
import importlib
import io
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle")):
    sys.path.insert(0, p)
import ref_walk  # noqa: E402

zj = importlib.import_module("zune-jpeg_amd")


def walked_equal(got, planes, rows, mcu_x, hs, vs, ncomp, mcu_limit=None):
    """coefficient planes equal on every MCU the reference walks -- up to MCU `mcu_limit` (scan order) when given: behind a
    marker the reference's reader serves what its rotating aligned_buffer holds, not zeros"""
    for c in range(ncomp):
        h, v = (hs, vs) if c == 0 else (1, 1)
        g = np.array(got[c], np.int16).reshape(-1, mcu_x * h, 64)
        p = np.array(planes[c], np.int16).reshape(-1, mcu_x * h, 64)
        for my, n in enumerate(rows):
            if n is None:
                continue
            for mx in range(mcu_x):
                if mcu_limit is not None and my * mcu_x + mx >= mcu_limit:
                    break
                if not np.array_equal(g[my * v:(my + 1) * v, mx * h:(mx + 1) * h], p[my * v:(my + 1) * v, mx * h:(mx + 1) * h]):
                    return False
    return True


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=120)
    ap.add_argument("--seed", type=int, default=1)
    a = ap.parse_args()
    from PIL import Image
    rng = np.random.default_rng(a.seed)
    os.environ["ZJ_PAR_SCAN"] = "off"
    t0 = time.time()
    files = variants = skipped = both_raise = with_restarts = behind_marker = 0
    while time.time() - t0 < a.seconds:
        w, h = int(rng.integers(3, 14)) * 8, int(rng.integers(3, 10)) * 8
        sub = int(rng.integers(0, 3))
        hs, vs = [(1, 1), (2, 1), (2, 2)][sub]
        amp = int(rng.integers(0, 127))
        base = (rng.integers(0, 2, (h // 8 + 1, w // 8 + 1, 1)) * 2 - 1) * rng.integers(0, 128, (h // 8 + 1, w // 8 + 1, 1))
        img = 128 + base.repeat(8, 0).repeat(8, 1)[:h, :w].repeat(3, 2) + (rng.integers(0, 2, (h, w, 3)) * 2 - 1) * rng.integers(0, amp + 1, (h, w, 3))
        kw = {}
        if rng.integers(0, 2):
            kw["restart_marker_blocks"] = int(rng.integers(1, 9))
        b = io.BytesIO()
        Image.fromarray(np.clip(img, 0, 255).astype(np.uint8)).save(b, "JPEG", quality=int(rng.choice([50, 90, 100])), subsampling=sub, **kw)
        data = b.getvalue()
        files += 1
        with_restarts += bool(kw)
        sos = data.index(b"\xff\xda")
        start = sos + 2 + int.from_bytes(data[sos + 2:sos + 4], "big")
        for trial in range(8):
            d = bytearray(data)
            if trial:
                at = int(rng.integers(start, len(d) - 2))
                kind = int(rng.integers(0, 6))
                if kind == 0:
                    d[at] ^= 1 << int(rng.integers(0, 8))
                elif kind == 1:
                    del d[at:at + int(rng.integers(1, 5))]
                elif kind == 2:
                    d[at:at] = bytes(rng.integers(0, 255, int(rng.integers(1, 12)), dtype=np.uint8))
                elif kind == 3:
                    d[at:at] = [b"\xff\x00", b"\xff\xff\x00", b"\xff\xff\xff\x00"][int(rng.integers(0, 3))]
                elif kind == 4:
                    d[at:at] = bytes([0xFF, 0xD0 + int(rng.integers(0, 8))])
                else:
                    d[at:at] = b"\xff\xd9"
            d = bytes(d)
            # markers the model does not follow
            odd = False
            i = start
            while True:
                i = d.find(b"\xff", i)
                if i < 0 or i + 1 >= len(d):
                    break
                m = d[i + 1]
                if m not in (0x00, 0xFF, 0xD9) and not 0xD0 <= m <= 0xD7:
                    odd = True
                    break
                i += 1 if m == 0xFF else 2
            if odd:
                skipped += 1
                continue
            variants += 1
            try:
                planes, short, rows = ref_walk.decode_baseline_planes(d)
                want_error = None
            except (ValueError, IndexError) as e:
                want_error = str(e)
            o = zj.ZuneJpegOptions()
            o.num_threads = 1
            dec = zj.Decoder(o)
            try:
                _, got, info = dec.decode_coefficients(d)
                got_error = None
            except zj.DecodeError as e:
                got_error = str(e)
            finally:
                dec.close()
            if want_error is not None or got_error is not None:
                if (want_error is None) != (got_error is None):
                    open("/tmp/ref_walk_soak_failure.jpg", "wb").write(d)
                assert (want_error is None) == (got_error is None), ("one side raises", files, trial, want_error, got_error)
                both_raise += 1
                continue
            mcu_x = (w + 8 * hs - 1) // (8 * hs)
            # Blocks the reference decodes with a marker pending are compared only where the file is undamaged and read in step (trial 0, no short DC
            # read: its data lasts to the last block).  In a damaged file the decode may run past the data, and behind a marker the
            # reference serves what its rotating aligned_buffer holds -- zero gap, then its own history -- where the front-end
            # serves zeros.
            fm = ref_walk.last_stats.get("first_marker_block") if ((trial or short) and os.environ.get("ZJ_LITERAL") in ("off", "0")) else None
            limit = None if fm is None else fm // (hs * vs + 2)   # the first MCU the reference decoded with a marker pending
            behind_marker += fm is not None
            if not walked_equal(got, planes, rows, mcu_x, hs, vs, 3, limit):
                open("/tmp/ref_walk_soak_failure.jpg", "wb").write(d)
            assert walked_equal(got, planes, rows, mcu_x, hs, vs, 3, limit), ("coefficients differ", files, trial, w, h, sub, kw, limit)
    print(f"ref_walk_soak: {files} files ({with_restarts} with restart intervals), {variants} variants compared with the literal model "
          f"({both_raise} end in an error on both sides; {behind_marker} compared only up to the first MCU decoded behind a marker), {skipped} skipped (markers the model does not follow), 0 differences; "
          f"{time.time() - t0:.0f} s, seed {a.seed}")


if __name__ == "__main__":
    main()
