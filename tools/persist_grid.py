#!/usr/bin/env python3
"""Variant 2 (prefetching walk) with different grid sizes: from persistent (1024 workgroups x 26 tiles) down to
two tiles per workgroup.  Is it persistence or the prefetch that costs?"""
import ctypes, importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
zj = importlib.import_module("zune-jpeg_amd")
synth = importlib.import_module("zune-jpeg_amd.synth")
W = H = 4096
B = 16
dev = torch.device("cuda", 0)
frames = [synth.make_frame(W, H, 2, 2, 3, seed=1234, frame_index=i) for i in range(2)]
desc = zj.FrameDesc.make(W, H, 2, 2, 3, zj.ColorSpace.RGB, frames[0][1])
d_planes = [torch.from_numpy(np.concatenate([frames[i % 2][0][c] for i in range(B)])).to(dev) for c in range(3)]
d_out = torch.empty(B * W * H * 3, dtype=torch.uint8, device=dev)
ctx = zj.Context()
ptrs = [t.data_ptr() for t in d_planes] + [d_out.data_ptr()]
side = torch.cuda.Stream().cuda_stream
L = zj.lib()
L.zj_set_persistent_grid.argtypes = [ctypes.c_int]
total = B * 1664
ctx.set_variant(0)
ctx.time_decode_device(desc, B, *ptrs, 150, side)
ms, _, _ = ctx.time_decode_device(desc, B, *ptrs, 200, side)
print(f"onepass                         {ms*1e3:7.1f} us")
ctx.set_variant(2)
for tiles_per_wg in (26, 13, 8, 4, 2, 1):
    L.zj_set_persistent_grid(total // tiles_per_wg)
    ms, _, _ = ctx.time_decode_device(desc, B, *ptrs, 200, side)
    print(f"prefetching walk, {tiles_per_wg:2d} tiles/WG ({total // tiles_per_wg:6d} WGs) {ms*1e3:7.1f} us")
