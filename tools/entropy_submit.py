#!/usr/bin/env python3
"""GPU box: one 4096x4096 file through the device entropy stage with the pixels left in HBM: wall time of
zj_decoder_finish_pixels_device and the part of it the host spends submitting (API calls in front of the final
synchronisation) -- what bounds the rate when several files are in flight."""
import importlib
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import files_bench  # noqa: E402

zj = importlib.import_module("zune-jpeg_amd")
data = files_bench.make_jpeg(4096, 0, restart_rows=0)
ctx = zj.Context()
o = zj.ZuneJpegOptions()
o.entropy = zj.ENTROPY_GPU
o.pinned_planes = True
d = zj.Decoder(o, ctx)
p = ctx.device_alloc(4096 * 4096 * 3)
best = (1e9, None)
for _ in range(20):
    d.prepare(data)
    t = time.perf_counter()
    d.finish_pixels_device(p, 4096 * 4096 * 3)
    dt = time.perf_counter() - t
    if dt < best[0]:
        best = (dt, ctx.scan_stats())
print(f"finish_pixels_device {best[0] * 1e3:.3f} ms, host submission {best[1][1][3]:.3f} ms, rounds {best[1][0]}")
