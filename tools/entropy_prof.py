#!/usr/bin/env python3
"""GPU box, under rocprofv3 --kernel-trace --stats: N decodes of one 4096x4096 4:2:0 q90 file through the device
entropy stage (sub-sequence size from ZJ_HUFF_SUB), nothing else on the GPU."""
import importlib
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np  # noqa: E402
import files_bench  # noqa: E402

zj = importlib.import_module("zune-jpeg_amd")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
data = files_bench.make_jpeg(4096, 0, restart_rows=0)
o = zj.ZuneJpegOptions()
o.entropy = zj.ENTROPY_GPU_ALWAYS
o.pinned_planes = True
ctx = zj.Context()
d = zj.Decoder(o, ctx)
out = np.zeros(4096 * 4096 * 3, np.uint8)
for _ in range(n):
    d.prepare(data)
    d.finish_pixels(out)
print("rounds", ctx.scan_stats()[0], "status", d.gpu_status())
