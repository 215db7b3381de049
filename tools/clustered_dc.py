#!/usr/bin/env python3
"""The synthetic benchmark frame scatters its 35 % DC-only blocks at random, the worst case for a lane-per-block
kernel (a wave runs the IDCT if ANY of its 64 blocks needs it).  Photographs cluster them (sky, walls): this times the
same frame with the same blocks made DC-only but gathered in contiguous rows, and an all-DC-only / no-DC-only pair."""
import importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
zj = importlib.import_module("zune-jpeg_amd")
synth = importlib.import_module("zune-jpeg_amd.synth")
W = H = 4096
B = 16
dev = torch.device("cuda", 0)
planes, qts = synth.make_frame(W, H, 2, 2, 3, seed=1234)
desc = zj.FrameDesc.make(W, H, 2, 2, 3, zj.ColorSpace.RGB, qts)
ctx = zj.Context()
side = torch.cuda.Stream().cuda_stream


def run(name, pl):
    d = [torch.from_numpy(np.tile(p, B)).to(dev) for p in pl]
    out = torch.empty(B * W * H * 3, dtype=torch.uint8, device=dev)
    ptrs = [t.data_ptr() for t in d] + [out.data_ptr()]
    ctx.time_decode_device(desc, B, *ptrs, 150, side)
    ms, _, _ = ctx.time_decode_device(desc, B, *ptrs, 200, side)
    frac = np.mean([float(np.mean(~np.any(p.reshape(-1, 64)[:, 1:] != 0, axis=1))) for p in pl])
    print(f"{name:44s} DC-only {100*frac:5.1f} %   {ms*1e3:7.1f} us / 16 frames   {B*W*H*6/ms/1e9:6.2f} TB/s")


run("benchmark frame (DC-only blocks scattered)", planes)
clustered = []
for p in planes:
    b = p.reshape(-1, 64).copy()
    dc = ~np.any(b[:, 1:] != 0, axis=1)
    order = np.argsort(~dc, kind="stable")   # DC-only blocks first, the others after, values unchanged
    clustered.append(b[order].reshape(-1))
run("same blocks, DC-only ones gathered", clustered)
alldc = [np.where(np.arange(p.size) % 64 == 0, p, 0).astype(np.int16) for p in planes]
run("every block DC-only", alldc)
nodc = []
for p in planes:
    b = p.reshape(-1, 64).copy()
    b[:, 1] = np.where(b[:, 1] == 0, 1, b[:, 1])
    nodc.append(b.reshape(-1))
run("no block DC-only", nodc)
