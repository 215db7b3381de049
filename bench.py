#!/usr/bin/env python3
"""
bench.py -- BASELINE.json's metric: megapixels/s decoded (dequantize+IDCT -> h2v2 up-sample -> RGB)
on synthetic 4096x4096 4:2:0 baseline frames, coefficient planes resident in HBM, plus the achieved
HBM GB/s of the fused kernel against the MI355X roofline and the CPU baseline timed beside it.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--frames B]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

A "step" = one pass of the hot path (one launch of the fused kernel) over a batch of B frames per GPU.
B defaults to 16 so one step moves 1.6 GB, far more than the 256 MiB Infinity Cache.  Frames are
image-sharded across ranks with no data-path collective (weak scaling); RCCL is used only for the
barrier, the MAX over ranks and the trivial gather of per-rank checksums after the timed region.
Rank 0 prints ONE JSON line.
"""
import argparse
import importlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for _p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "oracle")):
    if _p not in sys.path:
        sys.path.insert(0, _p)

W = H = 4096
HBM_PEAK_GBS = 8000.0            # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
BYTES_PER_PX = 6.0               # 3 B coefficients read + 3 B RGB written per pixel


def cpu_baseline(planes, qts, budget_s=20.0):
    """The reference's x86 path restated (kind "port"): oracle/zj_avx2.c follows src/idct/avx2.rs,
    src/color_convert/avx.rs and the strip-per-job pool of src/mcu.rs:356 (AVX2, N threads), timed on
    whole 4096x4096 4:2:0 frames at 4 threads (the reference's default, src/options.rs:33), at the container's
    CPU quota and at twice that (about 20 s of CPU work in total); `value` is the fastest with its thread count in `cores`.  The scalar oracle
    (1 thread) is timed beside it.  Checker/baseline only -- never on the product path."""
    import numpy as np
    import avx2_c
    import oracle_c as oc
    f = oc.make_frame(W, H, 2, 2, 3, oc.RGB, qts)
    out = np.zeros(W * H * 3, np.uint8)
    ncpu = os.cpu_count() or 1

    def run(fn, budget):
        fn()  # warm
        n, t0 = 0, time.perf_counter()
        while True:
            fn()
            n += 1
            dt = time.perf_counter() - t0
            if dt >= budget:
                return n * W * H / 1e6 / dt, n, dt

    res = {}
    eff = importlib.import_module("zune-jpeg_amd.shard").effective_cpus()  # cgroup quota, not just the CPU count
    tset = sorted({min(t, ncpu) for t in (4, eff, 2 * eff)})
    for t in tset:
        def fn(t=t):
            rc, _ = avx2_c.decode_planes_mt(f, planes, 1, t, out)
            assert rc == 0
        res[t] = run(fn, budget_s * (0.35 if t == 4 else 0.5 / max(len(tset) - 1, 1)))
    sc = run(lambda: oc.decode_planes(f, planes), budget_s * 0.15)
    best = max(res, key=lambda t: res[t][0])
    detail = "; ".join(f"{t} threads {res[t][0]:.0f} MP/s ({res[t][1]} frames, {res[t][2]:.1f} s)" for t in sorted(res))
    return {"value": round(res[best][0], 1), "unit": "megapixels/s", "cores": best, "kind": "port",
            "sample": f"restated zune-jpeg AVX2 path (oracle/zj_avx2.c) on 4096x4096 4:2:0 frames: {detail}; "
                      f"scalar restatement 1 thread {sc[0]:.0f} MP/s; host has {ncpu} logical CPUs, cgroup quota {eff}"}


def from_files(zj, ctx, size=4096, batch=16, reps=4):
    """Never `value`: whole 4096x4096 4:2:0 q90 JPEG FILES -> RGB left in HBM, Huffman decoding on the device too
    (zj_decoder_prepare on one host thread, zj_decoder_finish_pixels_batch in batches of 16), and the same
    files with the CPU walker in front of the pixel kernel.  None if anything is missing (Pillow writes the files)."""
    try:
        sys.path.insert(0, os.path.join(ROOT, "tools"))
        import files_bench
        blobs = [files_bench.make_jpeg(size, s, 0) for s in range(2)]
        mp = size * size / 1e6
        out = {}
        for name, mode in (("gpu_entropy", zj.ENTROPY_GPU), ("cpu_entropy", zj.ENTROPY_CPU)):
            o = zj.ZuneJpegOptions()
            o.entropy, o.pinned_planes, o.num_threads = mode, True, 1
            decs = [zj.Decoder(o, ctx) for _ in range(batch)]
            base = ctx.device_alloc(size * size * 3 * batch)
            ptrs = [(base + k * size * size * 3, size * size * 3) for k in range(batch)]
            n = batch if mode == zj.ENTROPY_GPU else 2
            best_total, best_prep = 1e9, 1e9
            for _ in range(reps if mode == zj.ENTROPY_GPU else 2):  # (the first pass also allocates: best of the passes)
                t0 = time.perf_counter()
                for k in range(n):
                    decs[k].prepare(blobs[k % len(blobs)])
                t1 = time.perf_counter()
                _, rcs = zj.finish_pixels_batch(decs[:n], ctx, device_ptrs=ptrs[:n])
                t2 = time.perf_counter()
                assert not any(rcs)
                best_total, best_prep = min(best_total, (t2 - t0) / n), min(best_prep, (t1 - t0) / n)
            out[name] = {"megapixels_per_s": round(mp / best_total, 1), "ms_per_file": round(best_total * 1e3, 3),
                         "host_ms_per_file": round(best_prep * 1e3, 3)}
            ctx.device_free(base)
            for d in decs:
                d.close()
        out["what"] = f"{size}x{size} 4:2:0 q90 baseline JPEG files ({len(blobs[0]) / 1e6:.2f} MB) -> RGB in HBM, one host thread, batches of {batch}; host_ms_per_file = container parsing + Huffman (cpu_entropy) or + scan preparation (gpu_entropy)"
        return out
    except Exception as e:  # the headline must not depend on this
        return {"error": repr(e)[:200]}


def load_traffic():
    """HBM bytes per launch from the committed rocprofv3 --pmc summary (tools/pmc_summary.py), or None."""
    p = os.path.join(ROOT, "profiles", "pmc_latest.json")
    try:
        return json.load(open(p))
    except Exception:
        return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=300)
    ap.add_argument("--warmup", type=int, default=100,
                    help="untimed steps; the first ~70 launches after idle run 5-35%% slow while the clocks settle")
    ap.add_argument("--frames", type=int, default=16, help="4096x4096 frames per GPU per step")
    ap.add_argument("--distinct", type=int, default=2, help="distinct synthetic frames generated (tiled to --frames)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--single-frame", action="store_true",
                    help="also time ONE frame per launch (configs[1] read literally); off by default so that a profiler "
                         "run of this command sees a single launch shape")
    ap.add_argument("--workload", choices=["420-rgb", "444-rgb", "444-gray", "422-rgb", "440-rgb", "420-rgba", "420-chw"], default="420-rgb",
                    help="420-rgb = BASELINE.json configs[1] (the headline); 444-* are configs[2]; 422 / 440 the reference's other sampling modes; 420-rgba / 420-chw are "
                         "the output extensions (4 B/px interleaved, planar u8)")
    ap.add_argument("--variant", choices=["packed", "wide", "packed-direct"], default=None, help="kernel variant (default: library default)")
    args = ap.parse_args()

    import numpy as np
    import torch  # before libzjhip: both bind the same libamdhip64.so.7
    zj = importlib.import_module("zune-jpeg_amd")
    synth = importlib.import_module("zune-jpeg_amd.synth")
    shard = importlib.import_module("zune-jpeg_amd.shard")

    rank, local_rank, world = shard.env_world()
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            sys.exit("bench.py --gpus N>1 must be launched with torch.distributed.run (one rank per GPU)")
        args.gpus = world
    if not torch.cuda.is_available():
        sys.exit("bench.py needs a GPU: the product path has no CPU fallback")
    # ZJ_BENCH_SAME_GPU=1 is a plumbing test only (tests the multi-rank control flow on a 1-GPU box: every rank uses
    # cuda:0 and the collectives run on gloo/CPU tensors); the driver's runs use one GPU per rank over RCCL
    same_gpu = os.environ.get("ZJ_BENCH_SAME_GPU") == "1"
    gpu_index = 0 if same_gpu else local_rank
    torch.cuda.set_device(gpu_index)
    dev = torch.device("cuda", gpu_index)
    shard.init_process_group("gloo" if same_gpu else "nccl", rank, world)
    coll_dev = "cpu" if same_gpu else dev

    B = args.frames
    ctx = zj.Context(zj.BACKEND_HIP, gpu_index)
    if args.variant:
        ctx.set_variant({"packed": 0, "wide": 1, "packed-direct": 2}[args.variant])
    # synthetic data, SURVEY.md 8d generator; every rank decodes its own shard of the global batch
    lo, _ = shard.shard_range(B * world, rank, world)
    hs, vs, out_cs, bytes_per_px = {"420-rgb": (2, 2, zj.ColorSpace.RGB, 6.0), "444-rgb": (1, 1, zj.ColorSpace.RGB, 9.0),
                                    "444-gray": (1, 1, zj.ColorSpace.GRAYSCALE, 3.0),
                                    "422-rgb": (2, 1, zj.ColorSpace.RGB, 7.0), "440-rgb": (1, 2, zj.ColorSpace.RGB, 7.0),
                                    "420-rgba": (2, 2, zj.ColorSpace.RGBA, 7.0), "420-chw": (2, 2, zj.ColorSpace.RGB, 6.0)}[args.workload]
    frames = [synth.make_frame(W, H, hs, vs, 3, seed=1234, frame_index=(lo + i) % max(args.distinct, 1))
              for i in range(min(args.distinct, B))]
    qts = frames[0][1]
    desc = zj.FrameDesc.make(W, H, hs, vs, 3, out_cs, qts, out_layout=zj.LAYOUT_CHW if args.workload == "420-chw" else zj.LAYOUT_HWC)
    host = [np.concatenate([frames[i % len(frames)][0][c] for i in range(B)]) for c in range(3)]
    d_planes = [torch.from_numpy(h).to(dev) for h in host]
    d_out = torch.empty(B * W * H * out_cs.num_components(), dtype=torch.uint8, device=dev)
    # a dedicated stream: launches on the legacy NULL stream serialise against every blocking stream
    # and cost ~30 us each (tools/launch_overhead.py); torch.cuda.synchronize() still covers it
    torch.cuda.synchronize()
    side = torch.cuda.Stream(device=dev)
    stream = side.cuda_stream
    ptrs = [t.data_ptr() for t in d_planes] + [d_out.data_ptr()]

    def step():
        ctx.decode_planes_device(desc, B, ptrs[0], ptrs[1], ptrs[2], ptrs[3], stream)

    # the first ~70 launches after idle run 5-35 % slow while the clocks settle: whatever --warmup says, at least 100
    # untimed launches precede the timed region (the W warmup steps are part of them)
    for _ in range(max(args.warmup, 100)):
        step()
    torch.cuda.synchronize()
    shard.barrier(world)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    shard.barrier(world)
    elapsed = shard.max_over_ranks(time.perf_counter() - t0, world, coll_dev)

    # dominant-kernel duration, HIP events recorded on the launch stream (outside the timed region)
    kiters = max(10, min(args.steps, 50))
    kernel_ms, kernel_ms_each, kname = ctx.time_decode_device(desc, B, ptrs[0], ptrs[1], ptrs[2], ptrs[3], kiters, stream)
    # configs[1] read literally is ONE 4096x4096 frame per launch: time that shape too (1664 workgroups = 1.3 waves of
    # the chip's 1280 workgroup slots, so the tail of every launch is exposed); reported beside the batched figure
    one_ms = one_ms_each = one_ms_4s = None
    if args.single_frame:
        one_ms, one_ms_each, _ = ctx.time_decode_device(desc, 1, ptrs[0], ptrs[1], ptrs[2], ptrs[3], 200, stream)
        # the same shape fed the way a frame-at-a-time caller would: launches rotating over four streams, so the tail
        # of one frame overlaps the head of the next (tools/single_frame_streams.py)
        four = [torch.cuda.Stream(device=dev) for _ in range(4)]
        ysz, csz, osz = d_planes[0].numel() // B * 2, d_planes[1].numel() // B * 2, d_out.numel() // B

        def rot(n):
            for i in range(n):
                f = i % B
                ctx.decode_planes_device(desc, 1, ptrs[0] + f * ysz, ptrs[1] + f * csz, ptrs[2] + f * csz, ptrs[3] + f * osz,
                                         four[i % 4].cuda_stream)
        rot(200)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        rot(2000)
        torch.cuda.synchronize()
        one_ms_4s = (time.perf_counter() - t1) / 2000 * 1e3
    # trivial gather: per-rank checksum of frame 0 (all ranks decode the same synthetic seeds modulo shard)
    torch.cuda.synchronize()
    first = d_out[: W * H * out_cs.num_components()].cpu().numpy()
    sums = shard.gather_checksums([shard.frame_checksum(first)], world, coll_dev)

    if rank == 0:
        mp_total = world * B * args.steps * W * H / 1e6
        algo_bytes = B * W * H * bytes_per_px
        achieved = algo_bytes / (kernel_ms * 1e-3) / 1e9
        tr = load_traffic()
        if tr and not (tr.get("workload", "420-rgb") == args.workload and tr.get("frames_per_launch", 16) == B):
            tr = None  # the committed counters describe another launch shape
        # Second bound, reported beside the HBM one: integer VALU issue.  A wave64 integer instruction occupies
        # its SIMD for 4 cycles (16 lanes per SIMD per clock; profiles/r01_ubench_valu_issue_cost.txt), so the
        # chip retires at most 1024 SIMDs x 2.4 GHz / 4 wave-instructions per second.
        valu = None
        if tr and tr.get("sq_insts_valu_per_launch"):
            peak_wi = 1024 * 2.4e9 / 4.0
            ach_wi = tr["sq_insts_valu_per_launch"] / (kernel_ms * 1e-3)
            valu = {"wave_insts_per_launch": tr["sq_insts_valu_per_launch"], "achieved": round(ach_wi / 1e9, 1),
                    "peak": round(peak_wi / 1e9, 1), "unit": "G wave64-instructions/s", "frac": round(ach_wi / peak_wi, 4),
                    "source": tr.get("sq_source")}
        res = {
            "metric": "megapixels/sec decoded (IDCT->RGB), 4K 4:2:0 baseline" if args.workload == "420-rgb" else f"megapixels/sec decoded, 4K {args.workload}",
            "value": round(mp_total / elapsed, 1),
            "unit": "megapixels/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "i32",  # the IDCT's fixed-point type (i16 coefficients in, packed-i16 colour math, u8 pixels out)
            "data": "synthetic",
            "config": {"workload": {"420-rgb": "configs[1]: 4096x4096 baseline 4:2:0, dequant+IDCT+h2v2+YCbCr->RGB, planes resident in HBM",
                                    "444-rgb": "configs[2]: 4096x4096 baseline 4:4:4, dequant+IDCT+YCbCr->RGB, planes resident in HBM",
                                    "444-gray": "configs[2]: 4096x4096 4:4:4 -> GRAYSCALE (luma only), planes resident in HBM",
                                    "422-rgb": "4096x4096 baseline 4:2:2 (h2v1), dequant+IDCT+horizontal upsample+YCbCr->RGB (the reference's benches/decode.rs horizontal case)",
                                    "440-rgb": "4096x4096 baseline 4:4:0 (h1v2), dequant+IDCT+vertical upsample+YCbCr->RGB (the reference's benches/decode.rs vertical case)",
                                    "420-rgba": "extension: 4096x4096 4:2:0 -> RGBA (R G B 255), planes resident in HBM",
                                    "420-chw": "extension: 4096x4096 4:2:0 -> planar u8 RGB (C x H x W), planes resident in HBM"}[args.workload],
                       "frames_per_gpu_per_step": B, "sharding": f"image-level x{world}, no data-path collective",
                       "distinct_frames": len(frames), "untimed_launches_before_timing": max(args.warmup, 100)},
            "roofline": {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 4),
                         "traffic": (tr or {}).get("hbm_bytes_per_launch"),
                         "kernel": kname, "kernel_ms": round(kernel_ms, 4), "kernel_ms_single_launch": round(kernel_ms_each, 4),
                         "algorithmic_bytes_per_launch": int(algo_bytes),
                         "traffic_source": (tr or {}).get("source"), "valu_issue": valu,
                         "single_frame_launch": None if one_ms is None else {
                             "kernel_ms": round(one_ms, 4), "kernel_ms_single_launch": round(one_ms_each, 4),
                             "megapixels_per_s": round(W * H / 1e6 / (one_ms * 1e-3), 1),
                             "achieved": round(W * H * bytes_per_px / (one_ms * 1e-3) / 1e9, 1),
                             "frac": round(W * H * bytes_per_px / (one_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                             "four_streams_ms_per_frame": round(one_ms_4s, 4),
                             "four_streams_megapixels_per_s": round(W * H / 1e6 / (one_ms_4s * 1e-3), 1),
                             "four_streams_frac": round(W * H * bytes_per_px / (one_ms_4s * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)}},
            # every rank's first frame is synthetic frame (rank*B) % distinct: identical data when B % distinct == 0
            "checksums_equal_across_ranks": (len({tuple(s) for s in sums}) == 1) if (world > 1 and B % max(args.distinct, 1) == 0) else None,
        }
        if not args.no_cpu_baseline and args.workload == "420-rgb" and world == 1:  # host baseline: rank 0 at N=1 only
            res["cpu_baseline"] = cpu_baseline(frames[0][0], qts)
            res["from_files"] = from_files(zj, ctx)
        print(json.dumps(res), flush=True)
    shard.barrier(world)
    ctx.close()
    if world > 1:
        import torch.distributed as dist
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
