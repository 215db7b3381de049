#!/usr/bin/env python3
"""One 4096x4096 4:2:0 frame per launch (BASELINE.json configs[1] read literally): back-to-back launches on ONE
stream expose every launch's tail (1664 workgroups = 1.3 rounds of the chip); launches rotating over K streams let
the tail of one frame overlap the head of the next.  Prints us per frame for K = 1, 2, 3, 4."""
import importlib
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

zj = importlib.import_module("zune-jpeg_amd")
synth = importlib.import_module("zune-jpeg_amd.synth")
W = H = 4096
NF = 8
planes, qts = synth.make_frame(W, H, 2, 2, 3, seed=1234)
desc = zj.FrameDesc.make(W, H, 2, 2, 3, zj.ColorSpace.RGB, qts)
dev = torch.device("cuda:0")
d = [torch.from_numpy(np.tile(p, NF)).to(dev) for p in planes]
out = torch.empty(NF * W * H * 3, dtype=torch.uint8, device=dev)
ctx = zj.Context()
yl, cl, ol = planes[0].size * 2, planes[1].size * 2, W * H * 3
for K in (1, 2, 3, 4):
    streams = [torch.cuda.Stream() for _ in range(K)]

    def run(n):
        for i in range(n):
            f = i % NF
            ctx.decode_planes_device(desc, 1, d[0].data_ptr() + f * yl, d[1].data_ptr() + f * cl, d[2].data_ptr() + f * cl,
                                     out.data_ptr() + f * ol, streams[i % K].cuda_stream)
    run(200)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    run(2000)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 2000
    print(f"{K} stream(s): {dt*1e6:6.1f} us per frame  {W*H/1e6/dt/1e3:7.1f} k megapixels/s  {W*H*6/dt/1e12:5.2f} TB/s")
