#!/usr/bin/env python3
"""A few launches of the fused 4:2:0 -> RGB kernel (16 frames of bench.py's generator) under one ablation / cut mask of
the diagnostic build -- the program tools/valu_ledger.sh puts behind `rocprofv3 --pmc`.  usage: ledger_run.py MASK [LAUNCHES]"""
import importlib
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

zj = importlib.import_module("zune-jpeg_amd")
synth = importlib.import_module("zune-jpeg_amd.synth")
mask = int(sys.argv[1]) if len(sys.argv) > 1 else 0
n = int(sys.argv[2]) if len(sys.argv) > 2 else 6
W = H = 4096
B = 16
dev = torch.device("cuda", 0)
pe = [synth.plane_blocks(W, H, 2, 2, c)[0] * synth.plane_blocks(W, H, 2, 2, c)[1] * 64 for c in range(3)]
d_planes = [torch.empty(B * k, dtype=torch.int16, device=dev) for k in pe]
for j in range(B):
    _, qts = synth.make_frame_t(W, H, 2, 2, 3, seed=1234, frame_index=j, device=dev, out=[d_planes[c][j * pe[c]:(j + 1) * pe[c]] for c in range(3)])
d_out = torch.empty(B * W * H * 3, dtype=torch.uint8, device=dev)
desc = zj.FrameDesc.make(W, H, 2, 2, 3, zj.ColorSpace.RGB, qts)
ctx = zj.Context(zj.BACKEND_HIP, 0)
ctx.set_ablation(mask)
side = torch.cuda.Stream(device=dev)
for _ in range(n):
    ctx.decode_planes_device(desc, B, d_planes[0].data_ptr(), d_planes[1].data_ptr(), d_planes[2].data_ptr(), d_out.data_ptr(), side.cuda_stream)
torch.cuda.synchronize()
ctx.close()
