#!/usr/bin/env python3
"""Where does bench.py's ms_per_step - kernel_ms gap come from?  Python-loop launches on the NULL
stream vs a dedicated stream vs the C loop inside zj_time_decode_device."""
import importlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
zj = importlib.import_module("zune-jpeg_amd")
synth = importlib.import_module("zune-jpeg_amd.synth")
W = H = 4096
B = 16
dev = torch.device("cuda", 0)
planes, qts = synth.make_frame(W, H, 2, 2, 3, seed=1234)
desc = zj.FrameDesc.make(W, H, 2, 2, 3, zj.ColorSpace.RGB, qts)
d_planes = [torch.from_numpy(np.tile(p, B)).to(dev) for p in planes]
d_out = torch.empty(B * W * H * 3, dtype=torch.uint8, device=dev)
ctx = zj.Context(zj.BACKEND_HIP, 0)
ptrs = [t.data_ptr() for t in d_planes] + [d_out.data_ptr()]
for name, stream in (("null stream", torch.cuda.current_stream().cuda_stream), ("torch side stream", torch.cuda.Stream().cuda_stream), ("ctx stream", None)):
    for _ in range(5):
        ctx.decode_planes_device(desc, B, *ptrs, stream)
    torch.cuda.synchronize()
    n = 100
    t0 = time.perf_counter()
    for _ in range(n):
        ctx.decode_planes_device(desc, B, *ptrs, stream)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    ms, each, _ = ctx.time_decode_device(desc, B, *ptrs, 50, stream)
    print(f"{name:18s}: python issue {1e3*(t1-t0)/n:.4f} ms/launch, issue+drain {1e3*(t2-t0)/n:.4f} ms/launch, C-loop events {ms:.4f} ms, single {each:.4f} ms")
