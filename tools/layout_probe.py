#!/usr/bin/env python3
"""Does the kernel's rate depend on WHERE in HBM a step's planes and pixels lie?  (bench.py's scattered_batch: 16 frames
picked at random from the shard cost 3 % more than 16 adjacent ones, through either form of addressing.)  One 16-frame
launch, 4096 x 4096 4:2:0 -> RGB, under several placements of the same data:
  planar     bench.py's: all Y planes of the shard back to back, then all Cb, all Cr, all outputs (one tensor each)
  skewed     the same with the four tensors' bases skewed by odd multiples of 1 MiB + 4 KiB
  per-frame  frame f's y | cb | cr | out adjacent (one arena, scattered launch)
  spread     the 16 frames of a step 8 frames apart in the planar shard (scattered launch)
  random / random-sorted / adjacent-shuffled   16 frames picked at random (in random or ascending order), the frames of a step shuffled
"""
import importlib
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402
zj = importlib.import_module("zune-jpeg_amd")
synth = importlib.import_module("zune-jpeg_amd.synth")
W = H = 4096
B, S = 16, 128


def timed(fn, side, iters=200):
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    for k in range(40):
        fn(k)
    ev[0].record(side)
    for k in range(iters):
        fn(k)
    ev[1].record(side)
    ev[1].synchronize()
    return ev[0].elapsed_time(ev[1]) / iters


def main():
    dev = torch.device("cuda", 0)
    ctx = zj.Context(zj.BACKEND_HIP, 0)
    side = torch.cuda.Stream(device=dev)
    pe = [synth.plane_blocks(W, H, 2, 2, c)[0] * synth.plane_blocks(W, H, 2, 2, c)[1] * 64 for c in range(3)]
    fo = W * H * 3
    frames = []
    for j in range(S):
        pl, qts = synth.make_frame_t(W, H, 2, 2, 3, seed=1234, frame_index=j, device=dev)
        frames.append(pl)
    d = zj.FrameDesc.make(W, H, 2, 2, 3, zj.ColorSpace.RGB, qts)
    torch.cuda.synchronize()
    nsub = S // B
    res = {}
    for rep in range(2):
        # planar / skewed
        for name, skew in (("planar", 0), ("skewed", 1)):
            pads = [(2 * i + 1) * ((1 << 20) + 4096) * skew for i in range(4)]
            arenas = [torch.empty(S * n * 2 + pads[i] + 256, dtype=torch.uint8, device=dev) for i, n in enumerate(pe)]
            arenas.append(torch.empty(S * fo + pads[3] + 256, dtype=torch.uint8, device=dev))
            base = [a.data_ptr() + pads[i] for i, a in enumerate(arenas)]
            base = [(b + 255) & ~255 for b in base]
            for c in range(3):
                for j in range(S):
                    off = base[c] - arenas[c].data_ptr() + j * pe[c] * 2
                    arenas[c][off:off + pe[c] * 2].view(torch.int16).copy_(frames[j][c])
            torch.cuda.synchronize()

            def step(k, base=base):
                q = k % nsub
                ctx.decode_planes_device(d, B, base[0] + q * B * pe[0] * 2, base[1] + q * B * pe[1] * 2, base[2] + q * B * pe[2] * 2,
                                         base[3] + q * B * fo, side.cuda_stream)
            res.setdefault(name, []).append(timed(step, side))
            if name == "planar":
                def spread(k, base=base):
                    q = k % 8
                    idx = [q + 8 * i for i in range(B)]
                    ctx.decode_frames_device(d, [base[0] + f * pe[0] * 2 for f in idx], [base[1] + f * pe[1] * 2 for f in idx],
                                             [base[2] + f * pe[2] * 2 for f in idx], [base[3] + f * fo for f in idx], side.cuda_stream)
                res.setdefault("spread", []).append(timed(spread, side))
                rng = np.random.default_rng(5)
                picks = [[int(v) for v in rng.permutation(S)[:B]] for _ in range(8)]
                for label, sets in (("random", picks), ("random-sorted", [sorted(p_) for p_ in picks]),
                                    ("adjacent-shuffled", [[q * B + int(v) for v in rng.permutation(B)] for q in range(nsub)])):
                    def scat(k, base=base, sets=sets):
                        idx = sets[k % len(sets)]
                        ctx.decode_frames_device(d, [base[0] + f * pe[0] * 2 for f in idx], [base[1] + f * pe[1] * 2 for f in idx],
                                                 [base[2] + f * pe[2] * 2 for f in idx], [base[3] + f * fo for f in idx], side.cuda_stream)
                    res.setdefault(label, []).append(timed(scat, side))
            del arenas
            torch.cuda.empty_cache()
        # per-frame
        per = pe[0] * 2 + 2 * pe[1] * 2 + fo
        per = (per + 4095) & ~4095
        arena = torch.empty(S * per + 256, dtype=torch.uint8, device=dev)
        a0 = (arena.data_ptr() + 255) & ~255
        o0 = a0 - arena.data_ptr()
        for j in range(S):
            o = o0 + j * per
            arena[o:o + pe[0] * 2].view(torch.int16).copy_(frames[j][0])
            arena[o + pe[0] * 2:o + pe[0] * 2 + pe[1] * 2].view(torch.int16).copy_(frames[j][1])
            arena[o + pe[0] * 2 + pe[1] * 2:o + pe[0] * 2 + 2 * pe[1] * 2].view(torch.int16).copy_(frames[j][2])
        torch.cuda.synchronize()

        def perframe(k):
            q = k % nsub
            idx = range(q * B, (q + 1) * B)
            ys = [a0 + f * per for f in idx]
            ctx.decode_frames_device(d, ys, [y + pe[0] * 2 for y in ys], [y + pe[0] * 2 + pe[1] * 2 for y in ys],
                                     [y + pe[0] * 2 + 2 * pe[1] * 2 for y in ys], side.cuda_stream)
        res.setdefault("per-frame", []).append(timed(perframe, side))
        del arena
        torch.cuda.empty_cache()
    for k, v in res.items():
        print(f"{k:10s} " + "  ".join(f"{t:.4f} ms ({B * W * H * 6 / (t * 1e-3) / 1e9 / 8000:.4f})" for t in v))
    ctx.close()


if __name__ == "__main__":
    main()
