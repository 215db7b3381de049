#!/usr/bin/env python3
"""CPU soak of scan_baseline_parallel (a baseline scan without restart markers entered at one point per thread): random files --
sizes, sampling modes, qualities, standard and optimised tables, grayscale -- intact and damaged, decoded by the serial walk and
by 2..8 threads with a small chunk threshold; planes, status and error text must be equal.  No GPU needed.
  python tools/par_scan_soak.py [--seconds 120] [--seed 1]
"""
import argparse
import importlib
import io
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
zj = importlib.import_module("zune-jpeg_amd")


def decode(data, threads, par, patience=None):
    os.environ.pop("ZJ_PAR_PATIENCE", None)
    if par:
        os.environ["ZJ_PAR_MIN_CHUNK"] = "600"
        if patience:  # how many MCUs the stitching may walk before it leaves the rest of the scan to the serial walk
            os.environ["ZJ_PAR_PATIENCE"] = str(patience)
        os.environ.pop("ZJ_PAR_SCAN", None)
    else:
        os.environ["ZJ_PAR_SCAN"] = "off"
    o = zj.ZuneJpegOptions()
    o.num_threads = threads
    dec = zj.Decoder(o)
    try:
        _, planes, info = dec.decode_coefficients(data, copy=False)
        return ("ok", [p.tobytes() for p in planes], dec.parallel_mcus())
    except zj.DecodeError as e:
        return ("error", str(e), 0)
    finally:
        dec.close()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=120)
    ap.add_argument("--seed", type=int, default=1)
    a = ap.parse_args()
    from PIL import Image, ImageFile
    ImageFile.MAXBLOCK = 1 << 24
    rng = np.random.default_rng(a.seed)
    t0 = time.time()
    files = taken = damaged = errors = 0
    while time.time() - t0 < a.seconds:
        w, h = int(rng.integers(64, 1400)), int(rng.integers(64, 900))
        gray = rng.integers(0, 6) == 0
        small = rng.integers(0, 256, (max(2, h // int(rng.integers(4, 40))), max(2, w // int(rng.integers(4, 40)))) + (() if gray else (3,)), dtype=np.uint8)
        img = Image.fromarray(small, "L" if gray else "RGB").resize((w, h), Image.BICUBIC)
        amp = int(rng.integers(0, 60))
        arr = np.asarray(img).astype(np.int16) + rng.integers(-amp, amp + 1, np.asarray(img).shape, dtype=np.int16)
        if rng.integers(0, 2):  # a flat band: runs of identical two-symbol MCUs, where a reader out of step stays out of step
            y0 = int(rng.integers(0, h - 16))
            arr[y0:y0 + int(rng.integers(16, max(17, h // 2)))] = int(rng.integers(0, 256))
        img = Image.fromarray(np.clip(arr, 0, 255).astype(np.uint8), "L" if gray else "RGB")
        b = io.BytesIO()
        kw = {} if gray else {"subsampling": int(rng.integers(0, 3))}
        img.save(b, "JPEG", quality=int(rng.choice([30, 60, 85, 92, 97, 100])), optimize=bool(rng.integers(0, 2)), **kw)
        data = b.getvalue()
        if len(data) < 12000:
            continue
        files += 1
        ref = decode(data, 1, False)
        for threads in (2, 4, int(rng.integers(3, 9))):
            got = decode(data, threads, True, int(rng.choice([0, 1, 8, 64])))
            if got[:2] != ref[:2]:
                open("/tmp/par_scan_soak_failure.jpg", "wb").write(data)
            assert got[:2] == ref[:2], ("intact", w, h, threads, got[0], ref[0])
            taken += got[2] > 0
        sos = data.index(b"\xff\xda")
        start = sos + 2 + int.from_bytes(data[sos + 2:sos + 4], "big")
        for _ in range(6):
            d = bytearray(data)
            at = int(rng.integers(start + 8, len(d) - 16))
            kind = int(rng.integers(0, 4))
            if kind == 0:
                d[at] ^= 1 << int(rng.integers(0, 8))
            elif kind == 1:
                d[at:at + 2] = bytes([0xFF, int(rng.choice([0x00, 0xD9, 0xD0, 0x17, 0xFF]))])
            elif kind == 2:
                del d[at:at + int(rng.integers(1, 6))]
            else:
                d[at:at] = bytes(rng.integers(0, 256, int(rng.integers(1, 9)), dtype=np.uint8))
            d = bytes(d)
            r1 = decode(d, 1, False)
            r2 = decode(d, int(rng.integers(2, 9)), True, int(rng.choice([0, 1, 8, 64])))
            if r1[:2] != r2[:2]:
                open("/tmp/par_scan_soak_failure.jpg", "wb").write(d)  # (kept for the post-mortem)
            assert r1[:2] == r2[:2], ("damaged", w, h, kind, at, r1[0], r2[0], r1[1] if r1[0] == "error" else "", r2[1] if r2[0] == "error" else "")
            damaged += 1
            errors += r1[0] == "error"
    print(f"par_scan_soak: {files} files ({taken} parallel decodes taken), {damaged} damaged variants ({errors} ending in an error), 0 differences; {time.time() - t0:.0f} s, seed {a.seed}")


if __name__ == "__main__":
    main()
