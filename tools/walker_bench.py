#!/usr/bin/env python3
"""The CPU walker by itself (container parsing + Huffman -> coefficient planes; zj_decoder_prepare with the CPU entropy
setting, which leaves the planes where they are) on ONE host thread, beside Pillow's / libjpeg-turbo's FULL decode of the same file on the same core.  No GPU needed: the planes are
pinned when a device is there (ZuneJpegOptions.pinned_planes) and malloc'd otherwise.

  python tools/walker_bench.py [files ...] [--reps 7] [--pinned] [--synthetic 4096]

Without file arguments: the reference's 7680x4320 speed_bench.jpg (4:4:4) and speed_bench_hv_subsampling.jpg (tests/golden/ref),
tests/golden/test-baseline.jpg, test-progressive.jpg, and a 4096x4096 4:2:0 q90 file written by Pillow (tools/files_bench.py).
"""
import argparse
import importlib
import io
import os
import statistics
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
zj = importlib.import_module("zune-jpeg_amd")


def blocks_of(info):
    mx = (info.width + 8 * info.h_max - 1) // (8 * info.h_max)
    my = (info.height + 8 * info.v_max - 1) // (8 * info.v_max)
    per = info.h_max * info.v_max + 2 if info.components == 3 else 1
    return mx * my * per


def time_walker(data, reps, pinned, threads=1):
    o = zj.ZuneJpegOptions()
    o.num_threads = threads
    o.pinned_planes = pinned
    dec = zj.Decoder(o)
    ts = []
    info = None
    for _ in range(reps):
        t0 = time.perf_counter()
        _, info = dec.prepare(data)  # (decode_coefficients would add a copy of the planes to every pass: 199 MB for speed_bench.jpg)
        ts.append(time.perf_counter() - t0)
    dec.close()
    return ts, info


def time_pillow(data, reps):
    try:
        from PIL import Image
    except Exception:  # noqa: BLE001
        return None
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter()
        im = Image.open(io.BytesIO(data))
        im.draft(None, None)
        im.load()
        ts.append(time.perf_counter() - t0)
    return ts


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("files", nargs="*")
    ap.add_argument("--reps", type=int, default=7)
    ap.add_argument("--pinned", action="store_true")
    ap.add_argument("--threads", type=int, default=1)
    ap.add_argument("--synthetic", type=int, default=4096)
    ap.add_argument("--no-pillow", action="store_true")
    ap.add_argument("--with-copy", action="store_true", help="also time Decoder.decode_coefficients, which copies the planes out (what a Python caller that wants the coefficients pays)")
    a = ap.parse_args()
    items = []
    if a.files:
        items = [(os.path.basename(f), open(f, "rb").read()) for f in a.files]
    else:
        for n in ("speed_bench.jpg", "speed_bench_hv_subsampling.jpg"):  # the reference's benchmark images (benches/decode.rs)
            items.append((n, open(os.path.join(ROOT, "tests", "golden", "ref", n), "rb").read()))
        for n in ("test-baseline.jpg", "test-progressive.jpg"):
            items.append((n, open(os.path.join(ROOT, "tests", "golden", n), "rb").read()))
        if a.synthetic:
            import files_bench
            items.append((f"pillow-q90-420-{a.synthetic}", files_bench.make_jpeg(a.synthetic, 0, 0)))
    print(f"walker on {a.threads} thread(s), planes {'pinned' if a.pinned else 'malloc'}, {a.reps} passes; "
          f"ms = min / median (the first pass also allocates and faults the planes in)")
    print(f"{'file':<28}{'MB':>7}{'blocks':>10}{'walker ms':>18}{'ns/block':>10}{'MP/s':>8}{'pillow full ms':>18}")
    for name, data in items:
        ts, info = time_walker(data, a.reps, a.pinned, a.threads)
        nb = blocks_of(info)
        best = min(ts)
        pl = None if a.no_pillow else time_pillow(data, max(3, a.reps // 2))
        extra = ""
        if a.with_copy:
            o = zj.ZuneJpegOptions()
            o.num_threads = a.threads
            dec = zj.Decoder(o)
            tc = []
            for _ in range(max(3, a.reps // 2)):
                t0 = time.perf_counter()
                dec.decode_coefficients(data)
                tc.append(time.perf_counter() - t0)
            dec.close()
            extra = f"   decode_coefficients (walker + a copy of the planes): {min(tc) * 1e3:.2f} ms"
        print(f"{name:<28}{len(data) / 1e6:>7.2f}{nb:>10}{best * 1e3:>9.2f} /{statistics.median(ts) * 1e3:>7.2f}"
              f"{best * 1e9 / nb:>10.1f}{info.width * info.height / 1e6 / best:>8.0f}"
              + (f"{min(pl) * 1e3:>9.2f} /{statistics.median(pl) * 1e3:>7.2f}" if pl else f"{'n/a':>18}") + extra)


if __name__ == "__main__":
    main()
