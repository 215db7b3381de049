#!/usr/bin/env python3
"""Single-thread time of the CPU entropy stage (zj_decoder_decode_coefficients) on a 4096x4096 4:2:0 q90 file made like
tools/files_bench.py's, and on a mostly-flat one (per-block overhead).  ZJ_LIB selects an A/B library."""
import importlib
import io
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np  # noqa: E402
import files_bench  # noqa: E402

zj = importlib.import_module("zune-jpeg_amd")


def flat_jpeg(size):
    from PIL import Image
    a = np.zeros((size, size, 3), np.uint8)
    a[:] = (90, 140, 200)
    a[::64, ::64] = 255
    b = io.BytesIO()
    Image.fromarray(a).save(b, "JPEG", quality=90, subsampling=2)
    return b.getvalue()


files = {"busy 4096 q90": files_bench.make_jpeg(4096, 0, restart_rows=0), "flat 4096 q90": flat_jpeg(4096)}
res = []
for name, data in files.items():
    o = zj.ZuneJpegOptions()
    o.num_threads = 1
    d = zj.Decoder(o)
    d.decode_coefficients(data)
    best = 1e9
    for _ in range(25):
        t = time.perf_counter()
        d.decode_coefficients(data)
        best = min(best, (time.perf_counter() - t) * 1e3)
    res.append(f"{name}: {best:6.2f} ms ({len(data) / 1e6:.2f} MB)")
print(os.environ.get("ZJ_LIB", "libzjhip.so"), " | ".join(res))
