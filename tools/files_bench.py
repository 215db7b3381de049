#!/usr/bin/env python3
"""End-to-end rate on whole JPEG FILES (never bench.py's `value`): CPU Huffman + PCIe + GPU pixel path.

Builds 4096x4096 4:2:0 baseline JPEGs with Pillow (quality 90, one restart interval per MCU row) from smooth
synthetic images, then times
  (a) one Decoder, strictly serial entropy decode            (num_threads = 1)
  (b) one Decoder, restart segments decoded on T threads     (num_threads = T)
  (c) zj_pool with N workers over a batch of files
Usage: python tools/files_bench.py [--files 32] [--size 4096]
"""
import argparse
import importlib
import io
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402

zj = importlib.import_module("zune-jpeg_amd")
EFF = importlib.import_module("zune-jpeg_amd.shard").effective_cpus()


def make_jpeg(size, seed, restart_rows=1):
    from PIL import Image
    rng = np.random.default_rng(seed)
    small = rng.integers(0, 256, (size // 32, size // 32, 3), dtype=np.uint8)
    img = Image.fromarray(small, "RGB").resize((size, size), Image.BICUBIC)
    arr = np.asarray(img).astype(np.int16) + rng.integers(-6, 7, (size, size, 3), dtype=np.int16)
    img = Image.fromarray(np.clip(arr, 0, 255).astype(np.uint8), "RGB")
    bio = io.BytesIO()
    img.save(bio, "JPEG", quality=90, subsampling=2, restart_marker_rows=restart_rows)
    return bio.getvalue()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--files", type=int, default=64)
    ap.add_argument("--distinct", type=int, default=4)
    ap.add_argument("--size", type=int, default=4096)
    ap.add_argument("--entropy", choices=["cpu", "gpu"], default="cpu", help="where baseline Huffman scans are decoded")
    ap.add_argument("--restart-rows", type=int, default=1, help="restart interval in MCU rows (0: none)")
    args = ap.parse_args()
    ENT = zj.ENTROPY_GPU if args.entropy == "gpu" else zj.ENTROPY_CPU
    S = args.size
    blobs = [make_jpeg(S, s, args.restart_rows) for s in range(args.distinct)]
    mp = S * S / 1e6
    print(f"entropy stage: {args.entropy}; {args.distinct} distinct {S}x{S} 4:2:0 q90 files (restart rows {args.restart_rows}), {sum(map(len, blobs)) / len(blobs) / 1e6:.2f} MB each; "
          f"host has {os.cpu_count()} logical CPUs, cgroup quota {EFF}")
    ctx = zj.Context()
    ref = None
    for t in (1, 4, 16, 64):
        if t > (os.cpu_count() or 1):
            continue
        o = zj.ZuneJpegOptions()
        o.num_threads = t
        o.pinned_planes = True
        o.entropy = ENT
        dec = zj.Decoder(o, ctx)
        out = dec.decode_buffer(blobs[0])
        if ref is None:
            ref = out.copy()
        assert np.array_equal(out, ref)
        n, t0 = 0, time.perf_counter()
        while time.perf_counter() - t0 < 2.0:
            dec.decode_buffer(blobs[n % len(blobs)])
            n += 1
        dt = (time.perf_counter() - t0) / n
        print(f"one decoder, num_threads {t:3d}: {dt*1e3:8.2f} ms/file  {mp/dt:9.1f} MP/s   (restart segments in parallel: {dec.parallel_segments()})")
        dec.close()
    files = [blobs[i % len(blobs)] for i in range(args.files)]
    L = zj.lib()
    import ctypes as C
    pins = [L.zj_alloc_pinned(S * S * 3) for _ in range(args.files)]
    outs = [np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_uint8)), shape=(S * S * 3,)) for p in pins]
    for workers in (1, 4, 8, 16, 32, 64):
        if workers > (os.cpu_count() or 1):
            continue
        po = zj.ZuneJpegOptions()
        po.entropy = ENT
        with zj.Pool(threads=workers, options=po) as pool:
            for _ in range(2):  # warm every worker: pinned planes, streams, device buffers
                pool.decode_files(files, outs=outs)
            e0, g0, n0 = pool.stats()
            t0 = time.perf_counter()
            res, _, sts = pool.decode_files(files, outs=outs)
            dt = time.perf_counter() - t0
            e1, g1, n1 = pool.stats()
            assert not any(sts) and np.array_equal(res[0], ref)
            print(f"zj_pool, {workers:3d} workers: {args.files} files in {dt*1e3:8.1f} ms  {args.files/dt:8.1f} files/s  {args.files*mp/dt:9.1f} MP/s"
                  f"   per file: entropy {(e1-e0)/(n1-n0)*1e3:6.1f} ms, GPU stage {(g1-g0)/(n1-n0)*1e3:6.2f} ms")
    for p in pins:
        L.zj_free_pinned(p)
    # files -> pixels that STAY in HBM (consumers on the GPU): T host threads, each with its own decoder, context and
    # device buffer (ctypes releases the GIL inside the library)
    import threading
    NB = 32
    dbase = ctx.device_alloc(S * S * 3 * NB)  # one allocation, images equally spaced (as in a tensor)
    dptr = [dbase + k * S * S * 3 for k in range(NB)]
    for workers in (1, 4, 16):
        po = zj.ZuneJpegOptions()
        po.entropy = ENT
        with zj.Pool(threads=workers, options=po) as pool:
            fl = [files[i % len(files)] for i in range(NB)]
            for _ in range(2):
                pool.decode_files_device(fl, dptr, [S * S * 3] * NB)
            t0 = time.perf_counter()
            reps = max(2, args.files // NB)
            for _ in range(reps):
                pool.decode_files_device(fl, dptr, [S * S * 3] * NB)
            dt = time.perf_counter() - t0
            print(f"zj_pool into HBM, {workers:3d} workers: {NB * reps} files in {dt*1e3:8.1f} ms  {NB*reps/dt:8.1f} files/s  {NB*reps*mp/dt:9.1f} MP/s")
    ctx.device_free(dbase)
    for T in (1, 2, 4, 8, 16):
        per = max(8, args.files // T)
        def work(k, res):
            c = zj.Context()
            o = zj.ZuneJpegOptions()
            o.num_threads = 1
            o.pinned_planes = True
            o.entropy = ENT
            d = zj.Decoder(o, c)
            p = c.device_alloc(S * S * 3)
            for i in range(2):
                d.prepare(blobs[i % len(blobs)]); d.finish_pixels_device(p, S * S * 3)
            res[k] = (c, d, p)
        res = [None] * T
        th = [threading.Thread(target=work, args=(k, res)) for k in range(T)]
        [t.start() for t in th]; [t.join() for t in th]
        def run(k):
            c, d, p = res[k]
            for i in range(per):
                d.prepare(blobs[(i + k) % len(blobs)]); d.finish_pixels_device(p, S * S * 3)
        th = [threading.Thread(target=run, args=(k,)) for k in range(T)]
        t0 = time.perf_counter()
        [t.start() for t in th]; [t.join() for t in th]
        dt = time.perf_counter() - t0
        print(f"pixels stay in HBM, {T} host threads: {T * per} files in {dt*1e3:8.1f} ms  {T*per/dt:8.1f} files/s  {T*per*mp/dt:9.1f} MP/s")
        for c, d, p in res:
            c.device_free(p); d.close(); c.close()


if __name__ == "__main__":
    main()
