#!/usr/bin/env python3
"""Needs the diagnostic build: tools/build_variant.sh ablate "-DZJ_ABLATION=1" and ZJ_LIB=libzjhip_ablate.so.  Ablation: how long does the fused kernel take when the IDCT and/or the colour math are skipped?
Tells VALU-bound from memory/latency-bound (results are wrong in the ablated runs; diagnostics only)."""
import importlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
zj = importlib.import_module("zune-jpeg_amd")
synth = importlib.import_module("zune-jpeg_amd.synth")
W = H = 4096
B = 16
dev = torch.device("cuda", 0)
frames = [synth.make_frame(W, H, 2, 2, 3, seed=1234, frame_index=i) for i in range(2)]
qts = frames[0][1]
desc = zj.FrameDesc.make(W, H, 2, 2, 3, zj.ColorSpace.RGB, qts)
d_planes = [torch.from_numpy(np.concatenate([frames[i % 2][0][c] for i in range(B)])).to(dev) for c in range(3)]
d_out = torch.empty(B * W * H * 3, dtype=torch.uint8, device=dev)
ctx = zj.Context(zj.BACKEND_HIP, 0)
ptrs = [t.data_ptr() for t in d_planes] + [d_out.data_ptr()]
side = torch.cuda.Stream().cuda_stream
ctx.time_decode_device(desc, B, *ptrs, 150, side)  # settle clocks
for name, mask in (("full", 0), ("no IDCT", 1), ("no colour math", 2), ("neither", 3), ("no loads", 4), ("no stores", 8),
                   ("no loads/stores", 12), ("no loads, no compute", 7), ("no stores, no compute", 11), ("no chroma LDS reads / filters", 16),
                   ("no chroma LDS, no colour math", 18), ("full", 0)):
    ctx.set_ablation(mask)
    ms, each, _ = ctx.time_decode_device(desc, B, *ptrs, 100, side)
    print(f"{name:24s} {ms*1e3:8.1f} us/launch   {B*W*H*6/ms/1e6:8.1f} GB/s")
ctx.set_ablation(0)
