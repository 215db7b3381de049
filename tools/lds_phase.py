#!/usr/bin/env python3
"""LDS bank conflicts by phase (diagnostic build, ZJ_LIB=libzjhip_ablate.so): launches the 16-frame 4:2:0 -> RGB kernel with
one ablation mask (argv[1]) so that a rocprofv3 --pmc pass around this script sees that variant only.
masks: 0 full | 16 no chroma LDS reads at all | 32 no 4-byte neighbour reads | 2 no colour math | 1 no IDCT"""
import importlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
zj = importlib.import_module("zune-jpeg_amd")
synth = importlib.import_module("zune-jpeg_amd.synth")
mask = int(sys.argv[1]) if len(sys.argv) > 1 else 0
W = H = 4096; B = 16
dev = torch.device("cuda", 0)
pe = [synth.plane_blocks(W, H, 2, 2, c)[0] * synth.plane_blocks(W, H, 2, 2, c)[1] * 64 for c in range(3)]
d_planes = [torch.empty(B * n, dtype=torch.int16, device=dev) for n in pe]
for j in range(B):
    _, qts = synth.make_frame_t(W, H, 2, 2, 3, seed=1234, frame_index=j, device=dev, out=[d_planes[c][j * pe[c]:(j + 1) * pe[c]] for c in range(3)])
desc = zj.FrameDesc.make(W, H, 2, 2, 3, zj.ColorSpace.RGB, qts)
d_out = torch.empty(B * W * H * 3, dtype=torch.uint8, device=dev)
torch.cuda.synchronize()
ctx = zj.Context(zj.BACKEND_HIP, 0)
ctx.set_ablation(mask)
side = torch.cuda.Stream().cuda_stream
ptrs = [t.data_ptr() for t in d_planes] + [d_out.data_ptr()]
ctx.time_decode_device(desc, B, *ptrs, 30, side)
ms, _, _ = ctx.time_decode_device(desc, B, *ptrs, 30, side)
print(f"mask {mask}: {ms * 1e3:.1f} us per launch")
