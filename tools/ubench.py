#!/usr/bin/env python3
"""Issue cost of gfx950 integer VALU instructions (inline-asm kernels in csrc/lab/zj_ubench.hip; libzjlab.so), in
cycles per wave64 instruction per SIMD.  Needs libzjlab.so and a GPU.  The result table drives the
instruction selection of the IDCT / colour code (DESIGN.md "VALU cost model")."""
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
    blocks, iters = 4096, 200            # 16 blocks per CU -> 8 waves per SIMD resident
    n_simd = 256 * 4
    wave_instr = blocks * 4 * iters * 64  # wave-instructions per launch
    print(f"shader clock during a 1-wave spin: {mhz:.0f} MHz (s_memtime / event time)")
    print(f"{'instruction':40s} {'ms':>8s} {'cyc/wave-instr/SIMD @clk':>26s}")
    for op in range(L.zjlab_ubench_count()):
        ms = ctx.ubench(op, blocks, iters, 5)
        cyc = ms * 1e-3 * mhz * 1e6 / (wave_instr / n_simd)
        print(f"{L.zjlab_ubench_name(op).decode():40s} {ms:8.3f} {cyc:26.2f}")
    ctx.close()


if __name__ == "__main__":
    main()
