#!/usr/bin/env python3
"""End-to-end rates that are NOT bench.py's `value` (planes resident in HBM): (a) host planes ->
H2D -> kernel -> D2H (PCIe-inclusive), pageable and pinned; (b) whole JPEG decode: CPU Huffman +
GPU pixels on the reference's 1920x1080 test images."""
import ctypes as C
import importlib
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np

zj = importlib.import_module("zune-jpeg_amd")
synth = importlib.import_module("zune-jpeg_amd.synth")
ctx = zj.Context()
W = H = 4096
planes, qts = synth.make_frame(W, H, 2, 2, 3, seed=1234)
desc = zj.FrameDesc.make(W, H, 2, 2, 3, zj.ColorSpace.RGB, qts)


def rate(fn, n=10):
    fn()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    return n * W * H / 1e6 / (time.perf_counter() - t0)


L = zj.lib()
out_page = np.zeros(W * H * 3, np.uint8)  # touched once: DMA into never-touched pages pays their first fault (5 ms / 50 MB)


def pageable():
    rc = L.zj_decode_planes(ctx.handle, C.byref(desc), planes[0].ctypes.data, planes[1].ctypes.data, planes[2].ctypes.data,
                            out_page.ctypes.data)
    assert rc == 0


print(f"4096x4096 4:2:0, host planes (pageable, resident) -> RGB on host: {rate(pageable):9.1f} MP/s")
print(f"   same through the Python wrapper (fresh output array per call): {rate(lambda: ctx.decode_planes(desc, planes)):9.1f} MP/s")
sizes = [p.nbytes for p in planes] + [W * H * 3]
pins = [L.zj_alloc_pinned(s) for s in sizes]
for p, pin in zip(planes, pins):
    C.memmove(pin, p.ctypes.data, p.nbytes)


def pinned():
    rc = L.zj_decode_planes(ctx.handle, C.byref(desc), pins[0], pins[1], pins[2], pins[3])
    assert rc == 0


print(f"4096x4096 4:2:0, host planes (pinned)   -> RGB on host: {rate(pinned):9.1f} MP/s   (100.7 MB over PCIe per frame; strip ranges overlapped on 3 streams)")
ctx.set_pipeline(0)
print(f"   same, one unit per call (no overlap)                  : {rate(pinned):9.1f} MP/s")
ctx.set_pipeline(1)
NB = 8
bp = [L.zj_alloc_pinned(s * NB) for s in sizes]
for p, pin in zip(planes, bp):
    for i in range(NB):
        C.memmove(pin + i * p.nbytes, p.ctypes.data, p.nbytes)


def pinned_batch():
    rc = L.zj_decode_planes_batch(ctx.handle, C.byref(desc), NB, bp[0], bp[1], bp[2], bp[3])
    assert rc == 0


print(f"   batch of {NB} frames (pinned), overlapped                 : {NB * rate(pinned_batch, 5):9.1f} MP/s")
ctx.set_pipeline(0)
print(f"   batch of {NB} frames (pinned), one unit                   : {NB * rate(pinned_batch, 5):9.1f} MP/s")
ctx.set_pipeline(1)
for p in bp:
    L.zj_free_pinned(p)
for p in pins:
    L.zj_free_pinned(p)
for name in ("test-baseline.jpg", "test-progressive.jpg"):
    data = open(os.path.join(ROOT, "tests", "golden", name), "rb").read()
    dec = zj.Decoder(None, ctx)
    dec.decode_buffer(data)
    n, t0 = 10, time.perf_counter()
    for _ in range(n):
        dec.decode_buffer(data)
    dt = (time.perf_counter() - t0) / n
    t0 = time.perf_counter()
    for _ in range(n):
        dec.decode_coefficients(data)
    dc = (time.perf_counter() - t0) / n
    print(f"{name:22s} decode_buffer {dt*1e3:7.2f} ms ({1920*1080/1e6/dt:7.1f} MP/s), of which CPU entropy decode {dc*1e3:7.2f} ms")
