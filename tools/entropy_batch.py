#!/usr/bin/env python3
"""GPU box: K prepared 4096x4096 scans through zj_decoder_finish_pixels_batch (one launch per phase for all of them),
pixels left in HBM: milliseconds per batch and files/s from ONE host thread."""
import importlib
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import files_bench  # noqa: E402

zj = importlib.import_module("zune-jpeg_amd")
S = 4096
blobs = [files_bench.make_jpeg(S, s, 0) for s in range(4)]
ctx = zj.Context()
o = zj.ZuneJpegOptions()
o.entropy = zj.ENTROPY_GPU
o.pinned_planes = True
KMAX = 16
decs = [zj.Decoder(o, ctx) for _ in range(KMAX)]
base = ctx.device_alloc(S * S * 3 * KMAX)  # one allocation, images equally spaced (as in a tensor): frames of one pixel launch
ptrs = [(base + k * S * S * 3, S * S * 3) for k in range(KMAX)]
for K in (1, 2, 4, 8, 16):
    best = 1e9
    for rep in range(6):
        for k in range(K):
            decs[k].prepare(blobs[k % 4])
        t = time.perf_counter()
        lens, rcs = zj.finish_pixels_batch(decs[:K], ctx, device_ptrs=ptrs[:K])
        dt = time.perf_counter() - t
        assert not any(rcs)
        best = min(best, dt)
    print(f"batch of {K:2d}: {best * 1e3:7.3f} ms  = {best * 1e3 / K:6.3f} ms/file  {K / best:8.0f} files/s  {K * S * S / 1e6 / best:9.0f} MP/s  (rounds {ctx.scan_stats()[0]})", flush=True)
