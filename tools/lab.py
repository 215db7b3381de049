#!/usr/bin/env python3
"""Times IDCT / colour formulations in isolation (csrc/lab/zj_lab.hip; libzjlab.so).  cycles = per wave-iteration per
SIMD, i.e. the VALU time one wave needs for 64 blocks (IDCT) or 64 x 16 pixels (colour)."""
import importlib
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import labctx
    ctx = labctx.Lab()
    L = ctx.L
    mhz = ctx.clock_mhz()
    blocks, iters = 4096, 20
    print(f"clock {mhz:.0f} MHz; {'variant':52s} {'ms':>8s} {'cycles/wave-iter/SIMD':>22s}")
    for v in range(L.zjlab_lab_count()):
        ms = ctx.lab(v, blocks, iters, 3)
        cyc = ms * 1e-3 * mhz * 1e6 * 1024 / (blocks * 4 * iters)
        print(f"{'':16s}{L.zjlab_lab_name(v).decode():52s} {ms:8.3f} {cyc:22.0f}")
    nbytes = 384 * 256 * 8192  # 805 MB in + 805 MB out, the size of one 16-frame launch
    for v in range(L.zjlab_labmem_count()):
        ms = ctx.labmem(v, nbytes, 20)
        name = L.zjlab_labmem_name(v).decode()
        moved = nbytes if name.startswith("read only") else 2 * nbytes
        print(f"{'':16s}{name:68s} {ms:8.3f} ms {moved / ms / 1e6:10.1f} GB/s")
    L.zjlab_rd_check.argtypes = [__import__("ctypes").c_void_p, __import__("ctypes").c_int, __import__("ctypes").c_int]
    for mode in (1, 2):
        print(f"LDS-DMA read path, mode {mode}: blocks differing from the plain loads over 4096 tiles: {L.zjlab_rd_check(ctx.h, mode, 4096)}")
    ctx.close()


if __name__ == "__main__":
    main()
