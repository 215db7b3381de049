#!/usr/bin/env python3
"""GPU box: the device entropy stage on 4096x4096 4:2:0 q90 files (tools/files_bench.py's generator) -- correctness
against the CPU walker, synchronisation rounds, and where the time goes: the CPU preparation (unstuffing, grid), the
device stage by phase (ZJ_HUFF_TIME events), one file end to end with either entropy setting."""
import importlib
import os
import sys
import time

os.environ["ZJ_HUFF_TIME"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np  # noqa: E402
import files_bench  # noqa: E402

zj = importlib.import_module("zune-jpeg_amd")
size = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
ctx = zj.Context()
for name, data in (("no restart markers", files_bench.make_jpeg(size, 0, restart_rows=0)),
                   ("restart marker per MCU row", files_bench.make_jpeg(size, 1, restart_rows=1))):
    for sub in (128, 64, 32):
        os.environ["ZJ_HUFF_SUB"] = str(sub)
        og, oc = zj.ZuneJpegOptions(), zj.ZuneJpegOptions()
        og.entropy = zj.ENTROPY_GPU_ALWAYS
        og.pinned_planes = oc.pinned_planes = True
        oc.num_threads = 1
        g, c = zj.Decoder(og, ctx), zj.Decoder(oc, ctx)
        want = c.decode_buffer(data)
        got = g.decode_buffer(data)
        same = np.array_equal(got, want)
        tp = 1e9
        for _ in range(5):
            t = time.perf_counter(); g.prepare(data); tp = min(tp, time.perf_counter() - t)
        import ctypes as C
        pin = zj.lib().zj_alloc_pinned(want.size)  # a pageable destination would make the download a blocking staged copy
        out = np.ctypeslib.as_array(C.cast(pin, C.POINTER(C.c_uint8)), shape=(want.size,))
        best, ms_best = 1e9, None
        for _ in range(8):
            g.prepare(data)
            t = time.perf_counter(); g.finish_pixels(out); dt = time.perf_counter() - t
            if dt < best:
                best, ms_best = dt, ctx.scan_stats()
        tc = 1e9
        for _ in range(3):
            t = time.perf_counter(); c.decode_buffer(data); tc = min(tc, time.perf_counter() - t)
        out = None
        zj.lib().zj_free_pinned(pin)
        rounds, ms = ms_best
        print(f"{size}x{size} {name} ({len(data) / 1e6:.2f} MB) sub {sub:3d}: same={same} status={g.gpu_status()} rounds={rounds} | "
              f"prepare {tp * 1e3:.2f} ms | device: upload+rounds {ms[0]:.3f} scan+write {ms[1]:.3f} pixels+download {ms[2]:.3f} ms | "
              f"finish_pixels {best * 1e3:.2f} ms (host submission {ms[3]:.3f}) | CPU-entropy decode_buffer {tc * 1e3:.1f} ms", flush=True)
