#!/usr/bin/env python3
"""One 4096x4096 4:2:0 frame per call, cut into N strip ranges launched on N internal streams that are forked from and
joined to the caller's stream by events: does the overlap of one range's loads with another's colour
phase pay for the fork / join?  Compared with the single launch (with and without the first-wave stagger).
usage: python tools/single_frame_split.py"""
import importlib
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

zj = importlib.import_module("zune-jpeg_amd")
synth = importlib.import_module("zune-jpeg_amd.synth")
W = H = 4096
S = 16
dev = torch.device("cuda", 0)
pe = [synth.plane_blocks(W, H, 2, 2, c)[0] * synth.plane_blocks(W, H, 2, 2, c)[1] * 64 for c in range(3)]
d_planes = [torch.empty(S * n, dtype=torch.int16, device=dev) for n in pe]
for j in range(S):
    _, qts = synth.make_frame_t(W, H, 2, 2, 3, seed=1234, frame_index=j, device=dev, out=[d_planes[c][j * pe[c]:(j + 1) * pe[c]] for c in range(3)])
fo = W * H * 3
d_out = torch.empty(S * fo, dtype=torch.uint8, device=dev)
side = torch.cuda.Stream(device=dev)
base = [t.data_ptr() for t in d_planes] + [d_out.data_ptr()]
fstr = [2 * n for n in pe] + [fo]
ref = None


def run(label, stagger, nsplit):
    global ref
    os.environ["ZJ_STAGGER"] = str(stagger)
    ctx = zj.Context(zj.BACKEND_HIP, 0)
    rows = H // nsplit
    assert rows % 32 == 0
    desc = zj.FrameDesc.make(W, rows, 2, 2, 3, zj.ColorSpace.RGB, qts)
    inner = [torch.cuda.Stream(device=dev) for _ in range(nsplit)] if nsplit > 1 else []
    fork = torch.cuda.Event()
    joins = [torch.cuda.Event() for _ in range(nsplit)]
    off = [[(r * rows // 8) * (W // 8) * 64 * 2, (r * rows // 16) * (W // 16) * 64 * 2, (r * rows // 16) * (W // 16) * 64 * 2, r * rows * W * 3]
           for r in range(nsplit)]

    def one(i):
        f = i % S
        p = [base[k] + f * fstr[k] for k in range(4)]
        if nsplit == 1:
            ctx.decode_planes_device(desc, 1, p[0], p[1], p[2], p[3], side.cuda_stream)
            return
        fork.record(side)
        for r in range(nsplit):
            inner[r].wait_event(fork)
            ctx.decode_planes_device(desc, 1, p[0] + off[r][0], p[1] + off[r][1], p[2] + off[r][2], p[3] + off[r][3], inner[r].cuda_stream)
            joins[r].record(inner[r])
            side.wait_event(joins[r])
    for i in range(100):
        one(i)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record(side)
    for i in range(640):
        one(i)
    e1.record(side)
    e1.synchronize()
    ms = e0.elapsed_time(e1) / 640
    # isolated: one frame, then idle
    tot = 0.0
    for i in range(40):
        torch.cuda.synchronize()
        e0.record(side)
        one(i)
        e1.record(side)
        e1.synchronize()
        tot += e0.elapsed_time(e1)
    for i in range(S):
        one(i)
    torch.cuda.synchronize()
    if ref is None:
        ref = d_out.clone()
    ok = bool(torch.equal(ref, d_out))
    print(f"{label:46s} back-to-back {ms * 1e3:6.2f} us ({W * H * 6 / ms / 1e6 / 8000:.3f} of 8 TB/s)   isolated {tot / 40 * 1e3:6.2f} us   identical: {ok}", flush=True)
    ctx.close()


for rep in range(2):
    run("one launch", 0, 1)
    run("one launch, stagger 16 (x128 cycles)", 16, 1)
    run("2 strip ranges on 2 streams, event fork/join", 0, 2)
    run("4 strip ranges on 4 streams, event fork/join", 0, 4)
    run("2 strip ranges, stagger 16", 16, 2)
