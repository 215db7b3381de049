#!/usr/bin/env python3
"""One 4096x4096 4:2:0 frame per launch (BASELINE configs[1] read literally) under different first-wave stagger settings
(ZJ_STAGGER, read at context creation): back-to-back launches on ONE stream walking 16 distinct frames, HIP events on that
stream; isolated launches (own event pair); output checked against the un-staggered decode.
usage: python tools/single_frame_ab.py [delays ...]      (default 0 1 2 3 4 6 8)"""
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
ref = None
side = torch.cuda.Stream(device=dev)
desc = zj.FrameDesc.make(W, H, 2, 2, 3, zj.ColorSpace.RGB, qts)
delays = [int(a) for a in sys.argv[1:]] or [0, 1, 2, 3, 4, 6, 8]
base = [t.data_ptr() for t in d_planes] + [d_out.data_ptr()]
fstr = [2 * n for n in pe] + [fo]
for rep in range(2):
    for dl in delays:
        os.environ["ZJ_STAGGER"] = str(dl)
        ctx = zj.Context(zj.BACKEND_HIP, 0)

        def one(i, n=1):
            f = i % S
            ctx.decode_planes_device(desc, n, base[0] + f * fstr[0], base[1] + f * fstr[1], base[2] + f * fstr[2], base[3] + f * fstr[3], side.cuda_stream)
        for i in range(100):
            one(i)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(side)
        for i in range(640):
            one(i)
        e1.record(side)
        e1.synchronize()
        ms = e0.elapsed_time(e1) / 640
        _, each, _ = ctx.time_decode_device(desc, 1, base[0], base[1], base[2], base[3], 100, side.cuda_stream)
        # the 16-frame launch must not care
        one(0, S)
        torch.cuda.synchronize()
        e0.record(side)
        for i in range(50):
            one(0, S)
        e1.record(side)
        e1.synchronize()
        ms16 = e0.elapsed_time(e1) / 50
        if ref is None:
            ref = d_out.clone()
        ok = bool(torch.equal(ref, d_out))
        print(f"ZJ_STAGGER={dl:2d}  one frame back-to-back {ms * 1e3:6.2f} us ({W * H * 6 / ms / 1e6 / 8000:.3f} of 8 TB/s)   isolated {each * 1e3:6.2f} us   "
              f"16 frames {ms16 * 1e3:7.2f} us   identical output: {ok}", flush=True)
        ctx.close()
