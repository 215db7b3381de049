#!/usr/bin/env python3
"""Integer-VALU issue-rate micro-benchmark on the GPU (needs libzjhip.so and a device).
Reports lane-ops/s per op kind relative to v_add_u32; decides e.g. whether v_mul_lo_u32 is
quarter-rate on gfx950 (it is why the IDCT uses v_mul_i32_i24 / v_mad_i32_i24)."""
import importlib
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
OPS = ["v_add_u32", "v_mul_lo_u32", "v_mul_i32_i24", "v_mad_i32_i24 (mul24+add)", "v_pk_mul_lo_u16",
       "v_pk_mad_u16 (pk mul+add)", "v_ashrrev+v_add (2 ops)", "v_perm_b32", "add+med3+add (3 ops)"]
NOPS = [1, 1, 1, 1, 1, 1, 2, 1, 3]


def main():
    zj = importlib.import_module("zune-jpeg_amd")
    ctx = zj.Context()
    blocks, iters = 4096, 400
    per_launch = blocks * 256 * iters * 64  # statements executed per launch
    base = None
    print(f"{'op':32s} {'ms':>9s} {'G stmts/s':>12s} {'rel. to add':>12s}")
    for op, name in enumerate(OPS):
        ms = ctx.ubench(op, blocks, iters, 5)
        rate = per_launch / (ms * 1e-3) / 1e9
        base = base or rate
        print(f"{name:32s} {ms:9.3f} {rate:12.1f} {rate / base:12.3f}   ({NOPS[op]} VALU op(s) per statement)")
    ctx.close()


if __name__ == "__main__":
    main()
