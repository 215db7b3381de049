#!/usr/bin/env python3
"""The whole zj_decoder_decode_buffer call (one host thread, planes and pixels pinned) with the strips streamed to the GPU
behind the walker, for several unit sizes (ZJ_STREAM_UNIT_MB), and with the stages apart (ZJ_STREAM=off).
  python tools/stream_sweep.py [--threads 1]      (--threads 4: the reference's default; scans without restart markers are
                                                  then entered at four points, zj_jpeg.cpp scan_baseline_parallel)
"""
import argparse
import ctypes as C
import importlib
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import numpy as np  # noqa: E402

zj = importlib.import_module("zune-jpeg_amd")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--threads", type=int, default=1)
    a = ap.parse_args()
    import files_bench
    L = zj.lib()
    L.zj_alloc_pinned.restype = C.c_void_p
    L.zj_alloc_pinned.argtypes = [C.c_size_t]
    L.zj_free_pinned.argtypes = [C.c_void_p]
    files = [("test-baseline.jpg 1920x1080 4:4:4", open(os.path.join(ROOT, "tests", "golden", "test-baseline.jpg"), "rb").read()),
             ("q90 2048x2048 4:2:0", files_bench.make_jpeg(2048, 1, 0)),
             ("q90 4096x4096 4:2:0", files_bench.make_jpeg(4096, 0, 0))]
    ctx = zj.Context()
    node = zj.bind_thread_near_device(0)
    print(f"thread bound to NUMA node {node}; num_threads {a.threads}; ms = best of 9 calls of Decoder.decode_buffer into pinned pixels")
    print(f"{'file':<36}{'stages apart':>14}" + "".join(f"{'unit ' + u + ' MB':>14}" for u in ("1", "2", "4", "8", "16")))
    for name, data in files:
        o = zj.ZuneJpegOptions()
        o.num_threads, o.pinned_planes = a.threads, True
        dec = zj.Decoder(o, ctx)
        info = dec.read_headers(data)
        n = int(info.width) * int(info.height) * 3
        pin = L.zj_alloc_pinned(n)
        out = np.ctypeslib.as_array(C.cast(pin, C.POINTER(C.c_uint8)), shape=(n,))
        cells = []
        ref = None
        for env in (("ZJ_STREAM", "off"),) + tuple(("ZJ_STREAM_UNIT_MB", u) for u in ("1", "2", "4", "8", "16")):
            os.environ[env[0]] = env[1]
            best = 1e9
            for _ in range(10):
                t0 = time.perf_counter()
                got = dec.decode_buffer(data, out=out)
                best = min(best, time.perf_counter() - t0)
            os.environ.pop(env[0])
            if ref is None:
                ref = got.copy()
            assert np.array_equal(got, ref)
            cells.append(best * 1e3)
        print(f"{name:<36}" + "".join(f"{c:>14.3f}" for c in cells))
        dec.close()
        L.zj_free_pinned(pin)
    ctx.close()


if __name__ == "__main__":
    main()
