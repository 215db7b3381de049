#!/usr/bin/env python3
"""The persistent-workgroup / LDS-DMA-prefetch experiment (csrc/lab/zj_persist.hip, libzjlab.so) against the product kernel
on bench.py's workload: 16 frames of 4096x4096 4:2:0 -> RGB per launch, 16 distinct generated frames.  Every variant's
output is compared with the product's byte for byte.  usage: python tools/persist_lab.py [reps]"""
import ctypes as C
import importlib
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch  # noqa: E402

import labctx  # noqa: E402

zj = importlib.import_module("zune-jpeg_amd")
synth = importlib.import_module("zune-jpeg_amd.synth")
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 60
W = H = 4096
B = 16
dev = torch.device("cuda", 0)
pe = [synth.plane_blocks(W, H, 2, 2, c)[0] * synth.plane_blocks(W, H, 2, 2, c)[1] * 64 for c in range(3)]
d_planes = [torch.empty(B * k, dtype=torch.int16, device=dev) for k in pe]
for j in range(B):
    _, qts = synth.make_frame_t(W, H, 2, 2, 3, seed=1234, frame_index=j, device=dev, out=[d_planes[c][j * pe[c]:(j + 1) * pe[c]] for c in range(3)])
fo = W * H * 3
d_ref = torch.empty(B * fo, dtype=torch.uint8, device=dev)
d_out = torch.empty(B * fo, dtype=torch.uint8, device=dev)
desc = zj.FrameDesc.make(W, H, 2, 2, 3, zj.ColorSpace.RGB, qts)
ctx = zj.Context(zj.BACKEND_HIP, 0)
side = torch.cuda.Stream(device=dev)
ptr = [t.data_ptr() for t in d_planes]
ms, _, kname = ctx.time_decode_device(desc, B, ptr[0], ptr[1], ptr[2], d_ref.data_ptr(), 150, side.cuda_stream)
ms, _, kname = ctx.time_decode_device(desc, B, ptr[0], ptr[1], ptr[2], d_ref.data_ptr(), reps, side.cuda_stream)
bytes_ = B * W * H * 6
print(f"product kernel ({kname.split('<')[1].split('>')[0]}): {ms * 1e3:8.1f} us per 16-frame launch = {bytes_ / ms / 1e6:7.1f} GB/s = {bytes_ / ms / 1e6 / 8000:.3f} of 8 TB/s")
lab = labctx.Lab()
L = lab.L
L.zjlab_persist.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_float)]
names = {0: "persistent + LDS-DMA prefetch (3 WG/CU)", 1: "persistent only, VGPR loads  (5 WG/CU)"}
for rep in range(2):
    for mode, groups in ((0, 0), (0, 512), (0, 256), (1, 0), (1, 1536), (1, 1024), (1, 768)):
        d_out.zero_()
        torch.cuda.synchronize()
        t = C.c_float(0)
        g = L.zjlab_persist(lab.h, C.byref(desc), B, ptr[0], ptr[1], ptr[2], d_out.data_ptr(), groups, mode, reps, C.byref(t))
        if g < 0:
            print(f"mode {mode} groups {groups}: failed ({g})")
            continue
        torch.cuda.synchronize()
        ok = bool(torch.equal(d_out, d_ref))
        m = t.value / reps
        print(f"{names[mode]:44s} grid {g:5d}: {m * 1e3:8.1f} us = {bytes_ / m / 1e6:7.1f} GB/s = {bytes_ / m / 1e6 / 8000:.3f}   output == product: {ok}", flush=True)
ms, _, _ = ctx.time_decode_device(desc, B, ptr[0], ptr[1], ptr[2], d_ref.data_ptr(), reps, side.cuda_stream)
print(f"product kernel again: {ms * 1e3:8.1f} us = {bytes_ / ms / 1e6 / 8000:.3f} of 8 TB/s")
lab.close()
ctx.close()
