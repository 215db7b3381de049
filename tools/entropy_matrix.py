#!/usr/bin/env python3
"""GPU box: the device entropy stage over qualities and sampling modes (4096x4096 photo-like content from
tools/files_bench.py's generator): file size, rounds, milliseconds to pixels in HBM, the CPU walker beside it."""
import importlib
import io
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
from PIL import Image  # noqa: E402

zj = importlib.import_module("zune-jpeg_amd")
S = 4096
rng = np.random.default_rng(0)
small = rng.integers(0, 256, (S // 32, S // 32, 3), dtype=np.uint8)
img = Image.fromarray(small, "RGB").resize((S, S), Image.BICUBIC)
arr = np.asarray(img).astype(np.int16) + rng.integers(-6, 7, (S, S, 3), dtype=np.int16)
img = Image.fromarray(np.clip(arr, 0, 255).astype(np.uint8), "RGB")
ctx = zj.Context()
p = ctx.device_alloc(S * S * 3)
for q in (50, 75, 90, 95, 98):
    for ss, name in ((2, "4:2:0"), (1, "4:2:2"), (0, "4:4:4")):
        b = io.BytesIO()
        img.save(b, "JPEG", quality=q, subsampling=ss)
        data = b.getvalue()
        o = zj.ZuneJpegOptions()
        o.entropy, o.pinned_planes = zj.ENTROPY_GPU, True
        d = zj.Decoder(o, ctx)
        best, rounds = 1e9, 0
        for _ in range(4):
            d.prepare(data)
            t = time.perf_counter(); d.finish_pixels_device(p, S * S * 3); dt = time.perf_counter() - t
            if dt < best:
                best, rounds = dt, ctx.scan_stats()[0]
        oc = zj.ZuneJpegOptions()
        oc.num_threads = 1
        c = zj.Decoder(oc, ctx)
        tc = 1e9
        for _ in range(2):
            t = time.perf_counter(); c.prepare(data); tc = min(tc, time.perf_counter() - t)
        print(f"q{q} {name}: {len(data) / 1e6:5.2f} MB  rounds {rounds:3d}  device {best * 1e3:6.2f} ms  status {d.gpu_status()}  on device {d.scan_blob() is not None}  | CPU walker {tc * 1e3:6.1f} ms", flush=True)
