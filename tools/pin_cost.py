#!/usr/bin/env python3
"""What pinning costs (round 6): zj_alloc_pinned / first touch / zj_free_pinned of 8 ... 224 MB, with the library's cache of freed
pinned blocks (the second 224 MB request is served from it) and, with ZJ_PINNED_CACHE_MB=0 in the environment, without.
  python tools/pin_cost.py;  ZJ_PINNED_CACHE_MB=0 python tools/pin_cost.py        (profiles/r06_reference_bench.txt)
"""
import sys, time, ctypes as C, importlib, os
sys.path.insert(0, os.getcwd())
zj = importlib.import_module("zune-jpeg_amd")
L = zj.lib()
L.zj_alloc_pinned.restype = C.c_void_p; L.zj_alloc_pinned.argtypes = [C.c_size_t]; L.zj_free_pinned.argtypes = [C.c_void_p]
ctx = zj.Context()
for mb in (8, 64, 224, 224, 100):
    t0 = time.perf_counter(); p = L.zj_alloc_pinned(mb << 20); t1 = time.perf_counter()
    C.memset(p, 1, mb << 20); t2 = time.perf_counter()
    L.zj_free_pinned(p); t3 = time.perf_counter()
    print(f"{mb:4d} MB: alloc {1e3*(t1-t0):8.2f} ms, first touch {1e3*(t2-t1):7.2f} ms, free {1e3*(t3-t2):7.2f} ms")
