#!/usr/bin/env python3
"""GPU box: zj_pool with the device entropy stage and device outputs (files -> pixels in HBM) for a few worker counts;
knobs come from the environment (ZJ_POOL_SUBMITTERS, GPU_MAX_HW_QUEUES, ZJ_HUFF_SUB)."""
import importlib
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import files_bench  # noqa: E402

zj = importlib.import_module("zune-jpeg_amd")
S, NB = 4096, 32
blobs = [files_bench.make_jpeg(S, s, 0) for s in range(4)]
files = [blobs[i % 4] for i in range(NB)]
ctx = zj.Context()
base = ctx.device_alloc(S * S * 3 * NB)  # one allocation, images equally spaced (as in a tensor)
dptr = [base + k * S * S * 3 for k in range(NB)]
o = zj.ZuneJpegOptions()
o.entropy = zj.ENTROPY_GPU
res = []
for workers in (2, 4, 8):
    with zj.Pool(threads=workers, options=o) as pool:
        for _ in range(2):
            pool.decode_files_device(files, dptr, [S * S * 3] * NB)
        t0 = time.perf_counter()
        for _ in range(4):
            pool.decode_files_device(files, dptr, [S * S * 3] * NB)
        dt = time.perf_counter() - t0
        res.append(f"{workers} workers {4 * NB / dt:7.0f} files/s")
print({k: os.environ.get(k) for k in ("ZJ_POOL_SUBMITTERS", "GPU_MAX_HW_QUEUES", "ZJ_HUFF_SUB")}, " | ".join(res), flush=True)
