#!/usr/bin/env python3
"""GPU soak of zj_decoder_decode_buffer with everything round 6 put between the file and the pixels switched on at once: pinned
planes, strips streamed to the GPU behind the walker (zj_frame_*), a scan without restart markers entered at one point per thread
(scan_baseline_parallel: chunk 0's rows leave after pass A, the rest from inside pass B), bridges over flat areas
(ZJ_PAR_PATIENCE), small chunk thresholds -- against the same file decoded on ONE thread with the stages apart.  Random files
(sizes, sampling modes, qualities, optimised tables, grayscale, flat bands, restart intervals), intact and damaged: pixels or
status + error text must be equal.
  python tools/stream_soak.py [--seconds 180] [--seed 1] [--cpu]
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


def decode(ctx, data, threads, stream, patience=None, gray=False):
    os.environ.pop("ZJ_PAR_PATIENCE", None)
    os.environ.pop("ZJ_STREAM", None)
    if not stream:
        os.environ["ZJ_STREAM"] = "off"
    if threads > 1:
        os.environ["ZJ_PAR_MIN_CHUNK"] = "2048"
        if patience:
            os.environ["ZJ_PAR_PATIENCE"] = str(patience)
    else:
        os.environ.pop("ZJ_PAR_MIN_CHUNK", None)
    o = zj.ZuneJpegOptions()
    o.num_threads, o.pinned_planes = threads, True
    if gray:
        o.out_colorspace = zj.ColorSpace.GRAYSCALE
    dec = zj.Decoder(o, ctx) if ctx is not None else zj.Decoder(o)
    try:
        if ctx is None:   # --cpu: the front-end alone (coefficient planes), for reproducing a finding without a GPU
            _, planes, _ = dec.decode_coefficients(data, copy=False)
            return ("ok", b"".join(p.tobytes() for p in planes), dec.parallel_mcus())
        px = dec.decode_buffer(data)
        return ("ok", px.tobytes(), dec.parallel_mcus())
    except zj.DecodeError as e:
        return ("error", str(e), 0)
    finally:
        dec.close()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=180)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--cpu", action="store_true", help="no GPU: compare the coefficient planes of the front-end instead of pixels")
    a = ap.parse_args()
    from PIL import Image, ImageFile
    ImageFile.MAXBLOCK = 1 << 26
    rng = np.random.default_rng(a.seed)
    ctx = None if a.cpu else zj.Context()
    t0 = time.time()
    files = taken = damaged = errors = unreadable = 0
    while time.time() - t0 < a.seconds:
        w, h = int(rng.integers(200, 2600)), int(rng.integers(200, 1800))
        gray = rng.integers(0, 8) == 0
        small = rng.integers(0, 256, (max(2, h // int(rng.integers(4, 40))), max(2, w // int(rng.integers(4, 40)))) + (() if gray else (3,)), dtype=np.uint8)
        img = Image.fromarray(small, "L" if gray else "RGB").resize((w, h), Image.BICUBIC)
        amp = int(rng.integers(0, 40))
        arr = np.asarray(img).astype(np.int16) + rng.integers(-amp, amp + 1, np.asarray(img).shape, dtype=np.int16)
        if rng.integers(0, 2):
            y0 = int(rng.integers(0, h - 16))
            arr[y0:y0 + int(rng.integers(16, max(17, h // 2)))] = int(rng.integers(0, 256))
        img = Image.fromarray(np.clip(arr, 0, 255).astype(np.uint8), "L" if gray else "RGB")
        b = io.BytesIO()
        kw = {} if gray else {"subsampling": int(rng.integers(0, 3))}
        if rng.integers(0, 6) == 0:
            kw["restart_marker_rows"] = int(rng.integers(1, 5))
        img.save(b, "JPEG", quality=int(rng.choice([40, 75, 90, 97])), optimize=bool(rng.integers(0, 2)), **kw)
        data = b.getvalue()
        if len(data) < 30000:
            continue
        files += 1
        out_gray = bool(rng.integers(0, 4) == 0)
        ref = decode(ctx, data, 1, False, gray=out_gray)
        # (an intact file may well end in an error: the reference reads DC symbols of 17 bits and more short -- src/bitstream.rs:278;
        # DESIGN.md section 7 -- and loses its place; Pillow's noisy q97 files have them.  The threaded decode must say the same.)
        unreadable += ref[0] != "ok"
        for threads in (2, 4, int(rng.integers(3, 17))):
            got = decode(ctx, data, threads, True, int(rng.choice([0, 1, 16, 200])), gray=out_gray)
            if got[:2] != ref[:2]:
                open("/tmp/stream_soak_failure.jpg", "wb").write(data)
            assert got[:2] == ref[:2], ("intact", w, h, threads, got[0])
            taken += got[2] > 0
        sos = data.index(b"\xff\xda")
        start = sos + 2 + int.from_bytes(data[sos + 2:sos + 4], "big")
        for _ in range(3):
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
            r1 = decode(ctx, d, 1, False, gray=out_gray)
            r2 = decode(ctx, d, int(rng.integers(2, 17)), True, int(rng.choice([0, 1, 16, 200])), gray=out_gray)
            if r1[:2] != r2[:2]:
                open("/tmp/stream_soak_failure.jpg", "wb").write(d)
            assert r1[:2] == r2[:2], ("damaged", w, h, kind, at, r1[0], r2[0], r1[1] if r1[0] == "error" else "", r2[1] if r2[0] == "error" else "")
            damaged += 1
            errors += r1[0] == "error"
    print(f"stream_soak: {files} files ({unreadable} of them end in an error on one thread as well: the reference's short DC reads; {taken} threaded decodes went through the parallel scan), {damaged} damaged variants ({errors} ending in an error), "
          f"0 differences; {time.time() - t0:.0f} s, seed {a.seed}")
    if ctx is not None:
        ctx.close()


if __name__ == "__main__":
    main()
