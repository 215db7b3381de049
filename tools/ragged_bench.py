#!/usr/bin/env python3
"""Kernel time of ragged-width frames (the reference's medium images are 2500 x 1786, tests/medium_images.rs): 32 resident
frames per launch, HIP events over 100 launches, per sampling mode -> RGB, for the library as built (fast interior + generic
edge launches, zj_plan.h: split_ragged) and, with ZJ_LIB set to an older build, whatever that build does."""
import importlib
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
zj = importlib.import_module("zune-jpeg_amd")
synth = importlib.import_module("zune-jpeg_amd.synth")


def main():
    dev = torch.device("cuda", 0)
    ctx = zj.Context(zj.BACKEND_HIP, 0)
    side = torch.cuda.Stream(device=dev)
    B = int(os.environ.get("ZJ_RAGGED_B", "32"))
    MODES = os.environ.get("ZJ_RAGGED_MODES", "420,444,422").split(",")
    PITCH = int(os.environ.get("ZJ_RAGGED_PITCH", "0"))  # e.g. 128: output rows at the next multiple of 128 bytes
    sizes = [tuple(int(v) for v in a.split("x")) for a in sys.argv[1:]] or [(2500, 1786), (2512, 1786), (4090, 4096), (4096, 4096)]
    for (w, h) in sizes:
        for name, (hs, vs), bpp in (("420", (2, 2), 6.0), ("444", (1, 1), 9.0), ("422", (2, 1), 7.0)):
            if name not in MODES:
                continue
            nb = B if w * h < 8e6 else (32 if w * h < 12e6 else 16)
            pe = [synth.plane_blocks(w, h, hs, vs, c)[0] * synth.plane_blocks(w, h, hs, vs, c)[1] * 64 for c in range(3)]
            pl = [torch.empty(nb * n, dtype=torch.int16, device=dev) for n in pe]
            for j in range(nb):
                _, qts = synth.make_frame_t(w, h, hs, vs, 3, seed=1234, frame_index=j, device=dev,
                                            out=[pl[c][j * pe[c]:(j + 1) * pe[c]] for c in range(3)])
            pitch = (3 * w + PITCH - 1) // PITCH * PITCH if PITCH else 0   # zj_frame_desc.out_pitch (0 = the tight rows)
            d = zj.FrameDesc.make(w, h, hs, vs, 3, zj.ColorSpace.RGB, qts, out_pitch=pitch)
            o = torch.empty(nb * (pitch or 3 * w) * h + 64, dtype=torch.uint8, device=dev)
            torch.cuda.synchronize()
            # frames packed back to back need out_len % 16 == 0 only for the aligned path; ragged frames take any byte
            ptr = [t.data_ptr() for t in pl] + [o.data_ptr()]
            ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
            for _ in range(100):
                ctx.decode_planes_device(d, nb, ptr[0], ptr[1], ptr[2], ptr[3], side.cuda_stream)
            ev[0].record(side)
            for _ in range(100):
                ctx.decode_planes_device(d, nb, ptr[0], ptr[1], ptr[2], ptr[3], side.cuda_stream)
            ev[1].record(side)
            ev[1].synchronize()
            ms = ev[0].elapsed_time(ev[1]) / 100
            gbs = nb * w * h * bpp / (ms * 1e-3) / 1e9
            _, _, kname = ctx.time_decode_device(d, nb, ptr[0], ptr[1], ptr[2], ptr[3], 1, side.cuda_stream)
            print(f"{w}x{h}{'/' + str(pitch) if pitch else ''} {name}->RGB  {nb} frames/launch  {ms:.4f} ms  {nb * w * h / 1e6 / (ms * 1e-3):.0f} MP/s  "
                  f"{gbs:.0f} GB/s  frac {gbs / 8000:.4f}  {kname[18:60]}", flush=True)
            del pl, o
            torch.cuda.empty_cache()
    ctx.close()


if __name__ == "__main__":
    main()
