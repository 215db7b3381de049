#!/usr/bin/env python3
"""zj_pool with FEWER files than workers (round 6): a batch of at most half as many files as workers gets the idle workers'
share of the CPUs inside its files -- a scan without restart markers is entered at several points (zj_jpeg.cpp
scan_baseline_parallel).  Batches of 1 ... 48 4096 x 4096 4:2:0 q90 files on a pool of 16 workers, pixels into pinned host
memory, with the lending and without (ZJ_POOL_LEND=off); every result compared with one decoder's.
  python tools/pool_short_batch.py [--workers 16] [--size 4096]
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
    ap.add_argument("--workers", type=int, default=16)
    ap.add_argument("--size", type=int, default=4096)
    a = ap.parse_args()
    import files_bench
    S = a.size
    blobs = [files_bench.make_jpeg(S, s, 0) for s in range(3)]
    ctx = zj.Context()
    o = zj.ZuneJpegOptions()
    o.num_threads = 1
    dec = zj.Decoder(o, ctx)
    refs = [dec.decode_buffer(b).copy() for b in blobs]
    dec.close()
    L = zj.lib()
    L.zj_alloc_pinned.restype = C.c_void_p
    L.zj_alloc_pinned.argtypes = [C.c_size_t]
    L.zj_free_pinned.argtypes = [C.c_void_p]
    counts = [n for n in (1, 2, 4, 8, 16, 48)]
    pins = [L.zj_alloc_pinned(S * S * 3) for _ in range(max(counts))]
    outs = [np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_uint8)), shape=(S * S * 3,)) for p in pins]
    print(f"pool of {a.workers} workers, {S}x{S} 4:2:0 q90 files without restart markers ({len(blobs[0]) / 1e6:.2f} MB), pixels into pinned memory; ms per batch, best of 5")
    print(f"{'files':>6}{'lending (default)':>20}{'ZJ_POOL_LEND=off':>20}{'threads per file':>18}")
    for n in counts:
        files = [blobs[i % len(blobs)] for i in range(n)]
        cells = []
        for lend in (None, "off"):
            if lend:
                os.environ["ZJ_POOL_LEND"] = lend
            try:
                with zj.Pool(threads=a.workers) as pool:
                    best = 1e9
                    for rep in range(7):
                        t0 = time.perf_counter()
                        res, _, sts = pool.decode_files(files, outs=outs[:n])
                        dt = time.perf_counter() - t0
                        if rep >= 2:   # (the first passes pin planes and start the decoders' helper threads)
                            best = min(best, dt)
                        assert not any(sts)
                        for i in range(n):
                            assert np.array_equal(res[i], refs[i % len(blobs)]), (n, lend, i)
                    cells.append(best * 1e3)
            finally:
                os.environ.pop("ZJ_POOL_LEND", None)
        per = a.workers // n if n * 2 <= a.workers else 1
        print(f"{n:>6}{cells[0]:>20.2f}{cells[1]:>20.2f}{min(per, 16):>18}")
    for p in pins:
        L.zj_free_pinned(p)
    ctx.close()


if __name__ == "__main__":
    main()
