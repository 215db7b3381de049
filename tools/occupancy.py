#!/usr/bin/env python3
"""Occupancy probe for the fused 4:2:0 -> RGB kernel (GPU box only; diagnostic build: tools/build_variant.sh ablate
"-DZJ_ABLATION", run with ZJ_LIB=libzjhip_ablate.so).

Pads every launch with dynamic LDS to force fewer resident workgroups per CU and times the kernel, and prints
hipOccupancyMaxActiveBlocksPerMultiprocessor for each padding.  Answers: how many workgroups per CU does the
default build get, and how sensitive is the kernel to that number.
"""
import ctypes
import importlib
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402

zj = importlib.import_module("zune-jpeg_amd")
synth = importlib.import_module("zune-jpeg_amd.synth")


def main():
    import torch
    lib = zj.lib()
    lib.zj_set_pad_lds.argtypes = [ctypes.c_int]
    lib.zj_fused_occupancy.argtypes = [ctypes.c_int]
    ctx = zj.Context()
    W = H = 4096
    nframes = 16
    planes, qts = synth.make_frame(W, H, 2, 2, 3, seed=1234)
    d = zj.FrameDesc.make(W, H, 2, 2, 3, zj.ColorSpace.RGB, qts)
    dev = torch.device("cuda:0")
    y, cb, cr = [torch.from_numpy(np.tile(pl, nframes)).to(dev) for pl in planes]
    out = torch.zeros(nframes * W * H * 3, dtype=torch.uint8, device=dev)
    torch.cuda.synchronize()
    pads = [0, 64, 2048, 5600, 8192, 13600, 16384, 22000, 27000, 30000, 49000]
    side = torch.cuda.Stream(device=dev).cuda_stream
    for nf in (nframes, 1):   # 16 frames per launch (the bench step) and ONE frame per launch (configs[1] read literally)
        for rep in range(2):
            for pad in pads:
                lib.zj_set_pad_lds(pad)
                occ = lib.zj_fused_occupancy(pad)
                ctx.time_decode_device(d, nf, y.data_ptr(), cb.data_ptr(), cr.data_ptr(), out.data_ptr(), 100, side)
                ms, each, _ = ctx.time_decode_device(d, nf, y.data_ptr(), cb.data_ptr(), cr.data_ptr(), out.data_ptr(), 300, side)
                if rep:
                    print(f"frames/launch {nf:3d}  pad_lds {pad:6d}  workgroups/CU {occ}  kernel {ms*1000:7.1f} us  ({ms*1000/nf:6.2f} us/frame)", flush=True)
    lib.zj_set_pad_lds(0)


if __name__ == "__main__":
    main()
