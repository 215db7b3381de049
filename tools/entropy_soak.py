#!/usr/bin/env python3
"""GPU box: soak test of the device entropy stage -- random sizes, qualities, sampling modes, restart intervals, table
optimisation, content (noise, smooth, flat areas, gradients) written by Pillow/libjpeg; every file decoded with the device
entropy stage (forced) and with the CPU walker: the pixels must be identical.  Prints a histogram of how the device
disposed of the scans (kept, handed back and why) and of the synchronisation rounds.

    python tools/entropy_soak.py [--seconds 120] [--seed 1]"""
import argparse
import collections
import importlib
import io
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
from PIL import Image  # noqa: E402

zj = importlib.import_module("zune-jpeg_amd")


def content(rng, w, h):
    kind = int(rng.integers(0, 6))
    if kind == 0:
        a = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
    elif kind == 1:
        small = rng.integers(0, 256, (max(2, h // 32), max(2, w // 32), 3), dtype=np.uint8)
        a = np.asarray(Image.fromarray(small).resize((w, h), Image.BICUBIC))
    elif kind == 2:
        a = np.full((h, w, 3), int(rng.integers(0, 256)), np.uint8)
        y0 = int(rng.integers(0, h))
        a[y0:] = rng.integers(0, 256, (h - y0, w, 3), dtype=np.uint8)
    elif kind == 3:
        yy, xx = np.mgrid[0:h, 0:w]
        a = np.stack([(xx * 255 // max(1, w - 1)), (yy * 255 // max(1, h - 1)), ((xx + yy) % 256)], -1).astype(np.uint8)
    elif kind == 4:
        a = np.full((h, w, 3), 255, np.uint8)
        for _ in range(int(rng.integers(1, 6))):
            y0, x0 = int(rng.integers(0, h)), int(rng.integers(0, w))
            a[y0:y0 + int(rng.integers(1, 60)), x0:x0 + int(rng.integers(1, w))] = rng.integers(0, 256, 3, dtype=np.uint8)
    else:
        small = rng.integers(0, 256, (max(2, h // 8), max(2, w // 8), 3), dtype=np.uint8)
        a = np.asarray(Image.fromarray(small).resize((w, h), Image.NEAREST))
    return a


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=120)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--mode", choices=["always", "gpu"], default="always", help="always: every eligible scan goes to the device; gpu: ZJ_ENTROPY_GPU's own choice (size, bits per block, round budget)")
    args = ap.parse_args()
    rng = np.random.default_rng(args.seed)
    ctx = zj.Context()
    disposal, rounds, n, bad = collections.Counter(), collections.Counter(), 0, 0
    t_end = time.time() + args.seconds
    while time.time() < t_end:
        w, h = int(rng.integers(8, 2600)), int(rng.integers(8, 2000))
        a = content(rng, w, h)
        kw = dict(quality=int(rng.integers(5, 101)), optimize=bool(rng.integers(0, 2)))
        gray = rng.integers(0, 6) == 0
        im = Image.fromarray(a)
        if gray:
            im = im.convert("L")
        else:
            kw["subsampling"] = int(rng.integers(0, 3))
        r = int(rng.integers(0, 4))
        if r == 1:
            kw["restart_marker_blocks"] = int(rng.integers(1, 40))
        elif r == 2:
            kw["restart_marker_rows"] = int(rng.integers(1, 4))
        b = io.BytesIO()
        try:
            im.save(b, "JPEG", **kw)
        except OSError:  # (libjpeg refuses some combinations, e.g. optimised tables with tiny buffers)
            continue
        data = b.getvalue()
        sub = int(rng.choice([32, 64, 128, 128, 128]))
        os.environ["ZJ_HUFF_SUB"] = str(sub)
        og, oc = zj.ZuneJpegOptions(), zj.ZuneJpegOptions()
        og.entropy = zj.ENTROPY_GPU_ALWAYS if args.mode == "always" else zj.ENTROPY_GPU
        cs = zj.ColorSpace.RGB if rng.integers(0, 3) else zj.ColorSpace.YCbCr
        og.out_colorspace = oc.out_colorspace = cs
        g, c = zj.Decoder(og, ctx), zj.Decoder(oc, ctx)
        try:
            want = c.decode_buffer(data)
        except zj.DecodeError as e:  # e.g. geometry on which the reference would panic: the same status either way
            try:
                g.decode_buffer(data)
                print(f"MISMATCH: the CPU path raises {e.status}, the device path does not ({w}x{h} {kw})", flush=True)
                bad += 1
            except zj.DecodeError as e2:
                if e2.status != e.status:
                    print(f"MISMATCH: statuses {e.status} / {e2.status} ({w}x{h} {kw})", flush=True)
                    bad += 1
            disposal["both raise"] += 1
            n += 1
            continue
        got = g.decode_buffer(data)
        n += 1
        st = g.gpu_status()
        key = "kept" if st == 0 and g.scan_blob() is not None else ("not prepared" if g.scan_blob() is None else "+".join(v for k, v in zj.HUFF_ST.items() if st & k))
        disposal[key] += 1
        if key == "kept":
            rounds[min(ctx.scan_stats()[0] // 8 * 8, 96)] += 1
        if not np.array_equal(got, want):
            bad += 1
            name = f"/tmp/soak_bad_{args.seed}_{n}.jpg"
            open(name, "wb").write(data)
            print(f"MISMATCH case {n}: {w}x{h} {kw} gray={gray} sub={sub} status={st} -> {name}", flush=True)
        g.close(); c.close()
    print(f"{n} files, {bad} mismatches; disposal {dict(disposal)}; rounds (bucketed by 8) {dict(sorted(rounds.items()))}")
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
