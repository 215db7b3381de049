#!/usr/bin/env python3
"""One 4096x4096 4:2:0 frame per launch, replayed from a HIP graph: zj_decode_planes_device is a pure kernel launch
(the tables travel in the kernel arguments; nothing is staged, nothing synchronises), so a frame-at-a-time caller can
capture a run of decodes -- here 16 frames over K parallel branches -- and replay it with one host call.
Prints us per frame for plain launches on K streams and for the captured graph."""
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
NF = 16
planes, qts = synth.make_frame(W, H, 2, 2, 3, seed=1234)
desc = zj.FrameDesc.make(W, H, 2, 2, 3, zj.ColorSpace.RGB, qts)
dev = torch.device("cuda:0")
d = [torch.from_numpy(np.tile(p, NF)).to(dev) for p in planes]
out = torch.empty(NF * W * H * 3, dtype=torch.uint8, device=dev)
ref = torch.empty_like(out)
ctx = zj.Context()
yl, cl, ol = planes[0].size * 2, planes[1].size * 2, W * H * 3
ctx.decode_planes_device(desc, NF, d[0].data_ptr(), d[1].data_ptr(), d[2].data_ptr(), ref.data_ptr(), torch.cuda.current_stream().cuda_stream)
torch.cuda.synchronize()


def one(f, stream):
    ctx.decode_planes_device(desc, 1, d[0].data_ptr() + f * yl, d[1].data_ptr() + f * cl, d[2].data_ptr() + f * cl,
                             out.data_ptr() + f * ol, stream.cuda_stream)


for K in (1, 2, 4):
    streams = [torch.cuda.Stream() for _ in range(K)]

    def run(n):
        for i in range(n):
            one(i % NF, streams[i % K])
    run(200)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    run(2000)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 2000
    print(f"plain launches, {K} stream(s): {dt*1e6:6.1f} us per frame  {W*H*6/dt/1e12:5.2f} TB/s")

    # the same NF decodes as a graph: fork K branches from the capture stream, join them at the end
    g = torch.cuda.CUDAGraph()
    cap = torch.cuda.Stream()
    out.zero_()
    torch.cuda.synchronize()
    with torch.cuda.stream(cap):
        g.capture_begin()
        fork = torch.cuda.Event()
        fork.record(cap)
        for s in streams:
            s.wait_event(fork)
        for f in range(NF):
            one(f, streams[f % K])
        for s in streams:
            e = torch.cuda.Event()
            e.record(s)
            cap.wait_event(e)
        g.capture_end()
    g.replay()
    torch.cuda.synchronize()
    assert torch.equal(out, ref), "graph replay differs from the batched decode"
    for _ in range(20):
        g.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    reps = 200
    for _ in range(reps):
        g.replay()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / (reps * NF)
    print(f"HIP graph of {NF} frames, {K} branch(es): {dt*1e6:6.1f} us per frame  {W*H*6/dt/1e12:5.2f} TB/s")
