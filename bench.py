#!/usr/bin/env python3
"""
bench.py -- BASELINE.json's metric: megapixels/s decoded (dequantize+IDCT -> h2v2 up-sample -> RGB)
on synthetic 4096x4096 4:2:0 baseline frames, coefficient planes resident in HBM, plus the achieved
HBM GB/s of the fused kernel against the MI355X roofline and the CPU baseline timed beside it.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--frames B] [--shard-frames S]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

Workload = BASELINE.json configs[4] sharded by image (configs[1] is its N = 1 case): every rank holds a shard of
S = 128 distinct frames in HBM (global frame i <- synth.make_frame_t(seed 1234, frame_index i), generated on the GPU
by an integer-only generator; 8 ranks = the 1024 frames of configs[4]).  A "step" = one pass of the hot path = ONE launch
of the fused kernel over B = 16 consecutive frames of the shard (1.6 GB of HBM traffic, far more than the 256 MiB
Infinity Cache); successive steps walk the shard.  The same per-GPU work at every N (weak scaling), no data-path
collective; RCCL is used only for the barrier, the MAX over ranks and the trivial gather of per-frame checksums after
the timed region, which are compared with the ORACLE's (tests/golden/checksums_seed1234.json).  Rank 0 prints ONE JSON line.

`--as-rank R/N` (one process, one GPU): the shard, the frame indices and the golden offsets of rank R of an N-rank run --
global frames [R*S, (R+1)*S) -- with world-size-1 collectives; eight such runs (R = 0..7) decode all 1024 frames of
configs[4] on one GPU (tools/virtual_ranks.py, tests/test_gpu_bench.py).

`--gpus N` without a torchrun environment is self-launching: the parent starts the N ranks as child processes BEFORE
anything touches a GPU, waits, relays rank 0's line and fails if any rank fails.
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


def cpu_baseline(frames, qts, budget_s=20.0):
    """The reference's x86 path restated (kind "port"): oracle/zj_avx2.c follows src/idct/avx2.rs, src/upsampler/avx2.rs
    (upsample_hv_avx, literally since round 3), src/color_convert/avx.rs and the strip-per-job pool of src/mcu.rs:356
    (AVX2, N threads), timed on whole 4096x4096 4:2:0 frames at 4 threads (the reference's default, src/options.rs:33),
    at the container's CPU quota and at twice that (about 20 s of CPU work in total); `value` is the fastest with its
    thread count in `cores`.  `frames` = the coefficient planes of >= 8 DISTINCT frames of the GPU's shard, decoded in
    rotation into as many output buffers (0.8 GB working set: nothing stays in the host's L3 between visits).  The scalar
    oracle (1 thread) is timed beside it.  Checker/baseline only -- never on the product path."""
    import numpy as np
    import avx2_c
    import oracle_c as oc
    f = oc.make_frame(W, H, 2, 2, 3, oc.RGB, qts)
    outs = [np.zeros(W * H * 3, np.uint8) for _ in frames]
    ncpu = os.cpu_count() or 1
    turn = [0]

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
            k = turn[0] = (turn[0] + 1) % len(frames)
            rc, _ = avx2_c.decode_planes_mt(f, frames[k], 1, t, outs[k])
            assert rc == 0
        res[t] = run(fn, budget_s * (0.35 if t == 4 else 0.5 / max(len(tset) - 1, 1)))

    def scalar():
        k = turn[0] = (turn[0] + 1) % len(frames)
        oc.decode_planes(f, frames[k])
    sc = run(scalar, budget_s * 0.15)
    best = max(res, key=lambda t: res[t][0])
    detail = "; ".join(f"{t} threads {res[t][0]:.0f} MP/s ({res[t][1]} frames, {res[t][2]:.1f} s)" for t in sorted(res))
    # `cores` = the cores the fastest run could actually occupy: its threads, capped by the cgroup's CPU quota (32 threads on
    # a 16-CPU quota are 16 cores' worth of work, time-sliced); the thread count, the quota and the host's size beside it
    return {"value": round(res[best][0], 1), "unit": "megapixels/s", "cores": min(best, eff), "threads": best, "cpu_quota": eff,
            "logical_cpus": ncpu, "kind": "port",
            "by_threads": {str(t): round(res[t][0], 1) for t in sorted(res)},
            "distinct_frames": len(frames), "l3_resident": False,
            "sample": f"restated zune-jpeg AVX2 path (oracle/zj_avx2.c) on 4096x4096 4:2:0 frames, {len(frames)} distinct "
                      f"frames in rotation: {detail}; "
                      f"scalar restatement 1 thread {sc[0]:.0f} MP/s; host has {ncpu} logical CPUs, cgroup quota {eff}"}


def pcie_probe(dev, nbytes=48 << 20, copies=16):
    """This host's PCIe ceilings, measured now: pinned <-> device copies the size of one frame's pixels (48 MiB), `copies`
    of them queued back to back from ONE host thread -- one direction at a time, then both directions on a stream each (the
    way zj_decode_planes_batch drives its upload and download streams; tools/pcie_probe.py, profiles/r01_pcie_probe.txt).
    GB/s per direction.  The duplex figure, not the link's nominal 63 GB/s, is the yardstick for e2e_pinned.  (Two host
    threads that each wait for their own copy, or copies of hundreds of MB, serialise the two directions on this host.)"""
    import torch
    try:
        h_up = torch.empty(nbytes, dtype=torch.uint8).pin_memory()
        h_dn = torch.empty(nbytes, dtype=torch.uint8).pin_memory()
        d_up = torch.empty(nbytes, dtype=torch.uint8, device=dev)
        d_dn = torch.empty(nbytes, dtype=torch.uint8, device=dev)
        s1, s2 = torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)

        def run(up, down):
            best = 1e9
            for _ in range(3):
                torch.cuda.synchronize(dev)
                t0 = time.perf_counter()
                for _ in range(copies):
                    if up:
                        with torch.cuda.stream(s1):
                            d_up.copy_(h_up, non_blocking=True)
                    if down:
                        with torch.cuda.stream(s2):
                            h_dn.copy_(d_dn, non_blocking=True)
                torch.cuda.synchronize(dev)
                best = min(best, time.perf_counter() - t0)
            return round(copies * nbytes / best / 1e9, 1)
        return {"h2d_alone_gbs": run(True, False), "d2h_alone_gbs": run(False, True),
                "duplex_gbs_per_direction": run(True, True), "bytes_per_copy": nbytes, "copies": copies,
                "how": "pinned <-> device copies queued back to back from one thread, a stream per direction"}
    except Exception as e:  # noqa: BLE001
        return {"error": repr(e)[:200]}


def e2e_pinned(zj, ctx, desc, d_planes, plane_elems, frame_out, golden_sums=None, nb=8, reps=4, probe=None):
    """Never `value`: the boundary as north_star words it -- coefficient planes in PINNED host memory streamed to HBM with
    hipMemcpyAsync, the fused kernel, the pixels streamed back to pinned host memory (zj_decode_planes_batch: three
    streams, uploads / kernels / downloads overlapped over units of ~16 MB).  `nb` distinct frames of the shard.  PCIe
    Gen5 x16 carries ~63 GB/s per direction at best, so this rate is the link's, not the kernel's."""
    import ctypes as C
    try:
        L = zj.lib()
        sizes = [2 * n * nb for n in plane_elems] + [frame_out * nb]
        pins = [L.zj_alloc_pinned(sz) for sz in sizes]
        if not all(pins):
            return {"error": "zj_alloc_pinned failed"}
        try:
            for c in range(3):
                host = d_planes[c][:nb * plane_elems[c]].cpu().numpy()
                C.memmove(pins[c], host.ctypes.data, host.nbytes)

            def once():
                rc = L.zj_decode_planes_batch(ctx.handle, C.byref(desc), nb, pins[0], pins[1], pins[2], pins[3])
                if rc != 0:
                    raise RuntimeError(f"zj_decode_planes_batch: {rc}")
            once()
            best = 1e9
            for _ in range(reps):
                t0 = time.perf_counter()
                once()
                best = min(best, time.perf_counter() - t0)
            import numpy as np
            got = np.ctypeslib.as_array((C.c_uint8 * frame_out).from_address(pins[3] + (nb - 1) * frame_out))
            synth = importlib.import_module("zune-jpeg_amd.synth")
            last_sum = synth.frame_checksum_sum(got)
        finally:
            for p_ in pins:
                L.zj_free_pinned(p_)
        up = sum(sizes[:3]) / best / 1e9
        down = sizes[3] / best / 1e9
        duplex = (probe or {}).get("duplex_gbs_per_direction")
        return {"megapixels_per_s": round(nb * W * H / 1e6 / best, 1), "ms_per_frame": round(best / nb * 1e3, 3),
                "frames": nb, "h2d_gbs": round(up, 1), "d2h_gbs": round(down, 1), "pcie_peak_gbs_per_direction": 63.0,
                "h2d_frac": round(up / 63.0, 3), "d2h_frac": round(down / 63.0, 3),
                # the yardstick that applies: both directions busy at once on THIS host, measured in this run
                "pcie_probe": probe,
                "frac_of_duplex_ceiling": round(min(up, down) / duplex, 3) if duplex else None,
                "last_frame_matches_golden": (last_sum == golden_sums[nb - 1]) if golden_sums else None,
                "what": f"{nb} distinct 4096x4096 4:2:0 frames: pinned host planes -> hipMemcpyAsync -> fused kernel -> "
                        f"hipMemcpyAsync -> pinned host RGB (zj_decode_planes_batch, 3 streams), best of {reps} passes"}
    except Exception as e:  # the headline must not depend on this
        return {"error": repr(e)[:200]}


def from_files(zj, ctx, size=4096, batch=16, reps=4):
    """Never `value`: whole 4096x4096 4:2:0 q90 JPEG FILES -> RGB left in HBM, Huffman decoding on the device too
    (DESIGN_ENTROPY.md; zj_decoder_prepare on one host thread, zj_decoder_finish_pixels_batch in batches of 16), and the same
    files with the CPU walker in front of the pixel kernel.  None if anything is missing (Pillow writes the files)."""
    try:
        sys.path.insert(0, os.path.join(ROOT, "tools"))
        import files_bench
        blobs = [files_bench.make_jpeg(size, s, 0) for s in range(2)]
        mp = size * size / 1e6
        out = {}
        for name, mode, nthreads in (("gpu_entropy", zj.ENTROPY_GPU, 1), ("cpu_entropy", zj.ENTROPY_CPU, 1),
                                     ("cpu_entropy_4_threads", zj.ENTROPY_CPU, 4),    # 4: the reference's default (src/options.rs:33)
                                     ("cpu_entropy_16_threads", zj.ENTROPY_CPU, 16)):  # 16: the most one scan is cut into
            o = zj.ZuneJpegOptions()
            o.entropy, o.pinned_planes, o.num_threads = mode, True, nthreads
            decs = [zj.Decoder(o, ctx) for _ in range(batch)]
            base = ctx.device_alloc(size * size * 3 * batch)
            ptrs = [(base + k * size * size * 3, size * size * 3) for k in range(batch)]
            n = batch if mode == zj.ENTROPY_GPU else 2
            best_total, best_prep = 1e9, 1e9
            for _ in range(reps if mode == zj.ENTROPY_GPU else 3):  # (the first pass also allocates and starts the decoders' helper threads: best of the passes)
                t0 = time.perf_counter()
                for k in range(n):
                    decs[k].prepare(blobs[k % len(blobs)])
                t1 = time.perf_counter()
                _, rcs = zj.finish_pixels_batch(decs[:n], ctx, device_ptrs=ptrs[:n])
                t2 = time.perf_counter()
                assert not any(rcs)
                best_total, best_prep = min(best_total, (t2 - t0) / n), min(best_prep, (t1 - t0) / n)
            out[name] = {"megapixels_per_s": round(mp / best_total, 1), "ms_per_file": round(best_total * 1e3, 3),
                         "host_ms_per_file": round(best_prep * 1e3, 3), "planes": "pinned", "host_threads": nthreads,
                         "mcus_decoded_in_parallel": decs[0].parallel_mcus() if mode == zj.ENTROPY_CPU else None,
                         "blocks": 6 * (size // 16) ** 2, "ns_per_block": round(best_prep * 1e9 / (6 * (size // 16) ** 2), 1)}
            ctx.device_free(base)
            for d in decs:
                d.close()
        out["what"] = f"{size}x{size} 4:2:0 q90 baseline JPEG files ({len(blobs[0]) / 1e6:.2f} MB) -> RGB in HBM, one host thread, batches of {batch}; host_ms_per_file = container parsing + Huffman (cpu_entropy; cpu_entropy_4_threads / _16_threads: the scan, which has no restart markers, entered at one point per thread, zj_jpeg.cpp scan_baseline_parallel) or + scan preparation (gpu_entropy)"
        return out
    except Exception as e:  # the headline must not depend on this
        return {"error": repr(e)[:200]}


def load_traffic():
    """Counters of the committed rocprofv3 --pmc summary (tools/pmc_summary.py), or None."""
    p = os.path.join(ROOT, "profiles", "pmc_latest.json")
    try:
        return json.load(open(p))
    except Exception:
        return None


def live_traffic(workload, B, S, timeout_s=200):
    """HBM bytes and VALU instructions per launch of the fused kernel measured in THIS run: three child runs of this script
    under `rocprofv3 --pmc` (FETCH_SIZE, WRITE_SIZE and SQ_INSTS_VALU in separate passes, as
    /opt/skills/guides/MI355X_MICROARCH.md prescribes: the TCC counters do not fit its four slots together), a few launches
    of the same shape each, summarised by tools/pmc_summary.py's rules (KiB units; FETCH_SIZE doubled on gfx950 for wide
    streaming reads).  The children are started as child processes (never exec'd from this GPU-initialised process).
    Always returns a dict: `hbm_bytes_per_launch` / `sq_insts_valu_per_launch` when the passes worked, `dropped` = why not
    otherwise (the caller then falls back to the committed summary and says so)."""
    import csv
    import glob
    import shutil
    import subprocess
    import tempfile
    exe = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(exe):
        return {"dropped": "rocprofv3 not found"}
    tmp = tempfile.mkdtemp(prefix="zj_pmc_", dir="/tmp")
    means, why = {}, {}
    try:
        for counter in ("FETCH_SIZE", "WRITE_SIZE", "SQ_INSTS_VALU"):
            d = os.path.join(tmp, counter)
            env = dict(os.environ, TMPDIR="/tmp")
            for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE"):
                env.pop(k, None)
            cmd = [exe, "--pmc", counter, "--output-format", "csv", "-d", d, "-o", "pmc", "--",
                   sys.executable, os.path.abspath(__file__), "--steps", "6", "--warmup", "2", "--min-untimed", "2",
                   "--frames", str(B), "--shard-frames", str(S), "--workload", workload, "--child"]
            try:
                r = subprocess.run(cmd, cwd="/tmp", env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=timeout_s)
            except subprocess.TimeoutExpired:
                why[counter] = f"child pass exceeded {timeout_s} s"
                continue
            if r.returncode != 0:
                why[counter] = f"child pass exited with {r.returncode}"
                continue
            vals = []
            for f in glob.glob(os.path.join(d, "**", "*counter_collection*.csv"), recursive=True):
                with open(f, newline="") as fh:
                    for row in csv.DictReader(fh):
                        if "zj_fused_kernel" in row.get("Kernel_Name", "") and row.get("Counter_Name") == counter:
                            vals.append(float(row["Counter_Value"]))
            if not vals:
                why[counter] = "no rows for the fused kernel in the counter CSV"
                continue
            means[counter] = (sum(vals) / len(vals), len(vals))
        out = {}
        if "FETCH_SIZE" in means and "WRITE_SIZE" in means:
            fetch, write = means["FETCH_SIZE"][0] * 1024, means["WRITE_SIZE"][0] * 1024
            out.update({"hbm_bytes_per_launch": int(2 * fetch + write), "fetch_bytes_raw": fetch, "write_bytes": write,
                        "launches": means["FETCH_SIZE"][1],
                        "source": f"rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes, child runs of this bench.py invocation "
                                  f"({means['FETCH_SIZE'][1]} launches of {B} frames each, walking the same {S}-frame shard as the timed steps); "
                                  f"FETCH_SIZE x2 per MI355X_MICROARCH.md"})
        if "SQ_INSTS_VALU" in means:
            out["sq_insts_valu_per_launch"] = int(means["SQ_INSTS_VALU"][0])
            out["sq_source"] = (f"rocprofv3 --pmc SQ_INSTS_VALU, its own pass, child run of this bench.py invocation "
                                f"({means['SQ_INSTS_VALU'][1]} launches)")
        if why:
            out["dropped"] = "; ".join(f"{k}: {v}" for k, v in why.items())
        return out
    except Exception as e:  # noqa: BLE001 -- the headline must not depend on the profiler
        return {"dropped": repr(e)[:200]}
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def other_workloads(zj, synth, ctx, dev, side, names=("444-rgb", "444-gray", "422-rgb", "440-rgb", "420-rgb-2500x1786", "420-rgb-2500x1786-pitch128"), B=16, iters=100):
    """Never `value`: BASELINE configs[2] (4:4:4 -> RGB, -> GRAYSCALE; benches/decode.rs:44-131, decode_grayscale.rs:12-58)
    and the reference's two other sampling modes, each timed like the headline's kernel: 16 resident 4096x4096 frames per
    launch, HIP events on the launch stream around `iters` launches, against each workload's own algorithmic bytes per
    pixel; and the reference's medium image size, a ragged width (60 frames per launch: the same pixels per launch), in the
    reference's tight layout and with its rows laid out at a pitch that is a multiple of 128 bytes (zj_frame_desc.out_pitch:
    a layout for outputs that stay in HBM; its rows must equal the tight layout's).
    Another kernel variant decodes frame 0 once more and must give the same bytes."""
    import torch
    out = {}
    B16 = B
    for name in names:
        try:
            hs, vs, cs_name, bpp, what = WORKLOADS[name][:5]
            W, H = workload_dims(name)
            B = max(1, round(B16 * 4096 * 4096 / (W * H)))
            cs = getattr(zj.ColorSpace, cs_name)
            pe = [synth.plane_blocks(W, H, hs, vs, c)[0] * synth.plane_blocks(W, H, hs, vs, c)[1] * 64 for c in range(3)]
            pl = [torch.empty(B * n, dtype=torch.int16, device=dev) for n in pe]
            for j in range(B):
                _, qts = synth.make_frame_t(W, H, hs, vs, 3, seed=1234, frame_index=j, device=dev,
                                            out=[pl[c][j * pe[c]:(j + 1) * pe[c]] for c in range(3)])
            row = W * cs.num_components()
            align = WORKLOADS[name][6] if len(WORKLOADS[name]) > 6 else 0
            pitch = (row + align - 1) // align * align if align else 0
            desc = zj.FrameDesc.make(W, H, hs, vs, 3, cs, qts, out_pitch=pitch)
            fo = (pitch or row) * H
            o = torch.empty(B * fo, dtype=torch.uint8, device=dev)
            torch.cuda.synchronize()
            ptr = [t.data_ptr() for t in pl] + [o.data_ptr()]
            ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
            for _ in range(120):  # (the CPU baseline left the GPU idle for seconds: the first ~70 launches after idle run slow)
                ctx.decode_planes_device(desc, B, ptr[0], ptr[1], ptr[2], ptr[3], side.cuda_stream)
            ev[0].record(side)
            for _ in range(iters):
                ctx.decode_planes_device(desc, B, ptr[0], ptr[1], ptr[2], ptr[3], side.cuda_stream)
            ev[1].record(side)
            ev[1].synchronize()
            ms = ev[0].elapsed_time(ev[1]) / iters
            _, _, kname = ctx.time_decode_device(desc, B, ptr[0], ptr[1], ptr[2], ptr[3], 1, side.cuda_stream)
            first = o[:fo].clone()
            # the same frame through another kernel variant: the wide generation where the library carries it
            # (make VARIANTS=all), else the packed generation with direct stores -- different store path, same bytes
            other = 1 if 1 in zj.variants_available() else 2
            ctx.set_variant(other)
            try:
                ctx.decode_planes_device(desc, 1, ptr[0], ptr[1], ptr[2], ptr[3], side.cuda_stream)
                side.synchronize()
            finally:
                ctx.set_variant(0)
            tight_ok = None
            if pitch:  # the padded layout's rows are the tight layout's rows
                t = torch.empty(row * H, dtype=torch.uint8, device=dev)
                ctx.decode_planes_device(zj.FrameDesc.make(W, H, hs, vs, 3, cs, qts), 1, ptr[0], ptr[1], ptr[2], t.data_ptr(), side.cuda_stream)
                side.synchronize()
                tight_ok = bool(torch.equal(first.view(H, pitch)[:, :row], t.view(H, row)))
                del t
            gbs = B * W * H * bpp / (ms * 1e-3) / 1e9
            out[name] = {"kernel_ms": round(ms, 4), "megapixels_per_s": round(B * W * H / 1e6 / (ms * 1e-3), 1),
                         "frames_per_launch": B, "width": W, "height": H, "bytes_per_px": bpp, "achieved": round(gbs, 1), "frac": round(gbs / HBM_PEAK_GBS, 4), "kernel": kname,
                         "matches_other_variant": bool(torch.equal(first, o[:fo])), "other_variant": {1: "wide", 2: "packed-direct"}[other], "what": what}
            if pitch:
                out[name].update({"out_pitch": pitch, "rows_match_tight_layout": tight_ok})
            del pl, o, first
            torch.cuda.empty_cache()
        except Exception as e:  # noqa: BLE001 -- the headline must not depend on this
            out[name] = {"error": repr(e)[:200]}
    return out


def reference_files(zj, ctx, reps=5):
    """Never `value`: BASELINE configs[0] / configs[3] -- the reference's own 1920x1080 test images (copies under
    tests/golden/) through Decoder: entropy stage on ONE host thread (baseline: the MCU walk of src/mcu.rs:231-351;
    progressive: the 10-scan coefficient accumulation of src/mcu_prog.rs:49) into pinned planes, then the pixel path on the
    GPU (upload, fused kernel, download).  The output must hash to what tests/golden/ref_images.json records."""
    import ctypes as C
    import hashlib
    import numpy as np
    zj.lib().zj_alloc_pinned.restype = C.c_void_p
    zj.lib().zj_alloc_pinned.argtypes = [C.c_size_t]
    zj.lib().zj_free_pinned.argtypes = [C.c_void_p]
    out = {}
    try:
        rec = {r["file"]: r for r in json.load(open(os.path.join(ROOT, "tests", "golden", "ref_images.json")))["files"]}
    except Exception as e:  # noqa: BLE001
        return {"error": repr(e)[:200]}
    for name in ("test-baseline.jpg", "test-progressive.jpg"):
        try:
            data = open(os.path.join(ROOT, "tests", "golden", name), "rb").read()
            r = rec["test-images/" + name]
            o = zj.ZuneJpegOptions()
            o.num_threads, o.pinned_planes = 1, True
            dec = zj.Decoder(o, ctx)
            host = gpu = 1e9
            px = None
            for _ in range(reps):
                t0 = time.perf_counter()
                dec.prepare(data)
                t1 = time.perf_counter()
                px = dec.finish_pixels(px)
                t2 = time.perf_counter()
                host, gpu = min(host, t1 - t0), min(gpu, t2 - t1)
            # the whole call a drop-in caller makes -- Decoder::decode_buffer, src/decoder.rs:178 -- into pinned pixels: for a
            # baseline file the strips go to the GPU while the walker is still in later rows (zj_frame_*; ZJ_STREAM=off: first
            # all of the Huffman stage, then upload + kernel + download)
            whole = {}
            pin = zj.lib().zj_alloc_pinned(r["width"] * r["height"] * 3)
            try:
                pout = np.ctypeslib.as_array(C.cast(pin, C.POINTER(C.c_uint8)), shape=(r["width"] * r["height"] * 3,))
                modes = (("decode_buffer_ms", "decode_buffer_matches", None),
                         ("decode_buffer_ms_stages_apart", "decode_buffer_stages_apart_matches", "off"))
                best = {m[0]: 1e9 for m in modes}
                for _ in range(reps + 2):          # the two modes in turn, so that a slow phase of the host hits both
                    for key, mkey, env in modes:
                        if env:
                            os.environ["ZJ_STREAM"] = env
                        try:
                            t0 = time.perf_counter()
                            got = dec.decode_buffer(data, out=pout)
                            best[key] = min(best[key], time.perf_counter() - t0)
                            whole[mkey] = whole.get(mkey, True) and bool(np.array_equal(got, px))
                        finally:
                            os.environ.pop("ZJ_STREAM", None)
                for key, _, _ in modes:
                    whole[key] = round(best[key] * 1e3, 3)
            finally:
                zj.lib().zj_free_pinned(pin)
            dec.close()
            # the same stage with the reference's default of four threads (src/options.rs:33).  Expect no gain on these two: a
            # progressive file's scans stay serial, and the baseline one is 73 KB -- scan_baseline_parallel starts at 96 KB of
            # scan (from_files.cpu_entropy_4_threads is the case it is for)
            o4 = zj.ZuneJpegOptions()
            o4.num_threads, o4.pinned_planes = 4, True
            dec4 = zj.Decoder(o4, ctx)
            host4, px4 = 1e9, None
            for _ in range(reps + 2):
                t0 = time.perf_counter()
                dec4.prepare(data)
                host4 = min(host4, time.perf_counter() - t0)
                par4 = dec4.parallel_mcus()
                px4 = dec4.finish_pixels(px4)
            whole["host_entropy_ms_4_threads"] = round(host4 * 1e3, 3)
            whole["mcus_decoded_in_parallel"] = int(par4)
            whole["four_threads_match"] = bool(np.array_equal(px4, px))
            dec4.close()
            mp = r["width"] * r["height"] / 1e6
            blocks = 3 * ((r["width"] + 7) // 8) * ((r["height"] + 7) // 8)  # 4:4:4: three planes of (w/8) x (h/8) blocks
            out[name] = {"host_entropy_ms": round(host * 1e3, 3), "gpu_pixels_ms": round(gpu * 1e3, 3),
                         "blocks": blocks, "ns_per_block": round(host * 1e9 / blocks, 1), "planes": "pinned", "host_threads": 1,
                         "megapixels_per_s": round(mp / (host + gpu), 1), "scans": r["scans"], "progressive": bool(r["progressive"]),
                         "width": r["width"], "height": r["height"],
                         "sha256_matches_golden": hashlib.sha256(np.ascontiguousarray(px).tobytes()).hexdigest() == r["sha256_rgb"]}
            out[name].update(whole)
        except Exception as e:  # noqa: BLE001
            out[name] = {"error": repr(e)[:200]}
    out["what"] = ("the reference's test-images/*.jpg (1920x1080 4:4:4): Huffman / progressive accumulation on one host thread into "
                   "pinned planes (host_entropy_ms), then upload + fused kernel + download (gpu_pixels_ms); best of 5")
    return out


def reference_bench(zj, ctx, reps=4):
    """Never `value`: the reference's OWN benchmark (benches/decode.rs:9-14,44-131: `Decoder::new()` + `decode_buffer` of
    benches/images/speed_bench*.jpg, 7680 x 4320; Benches.md quotes the whole-decode times -- Huffman included, four worker
    threads, a Ryzen 5 4500U) on this host + GPU, the whole call, two ways:
      as_the_reference_does  a NEW decoder per call with default options (4 threads, pageable planes), pixels into a fresh buffer
      new_decoder_pinned     a new decoder per call with zj_options.pinned_planes (the library hands freed pinned blocks to the
                             next decoder), pixels into a fresh pageable buffer
      steady_state           one decoder kept, planes and pixels pinned (what a caller that decodes many files would do)
    Bytes checked against the hashes the oracle recorded (tests/golden/ref_images.json).  Context, like BASELINE.md section 2: a
    different CPU, and here a GPU does the pixel path -- `published_ms` is the reference's own figure for the same file."""
    import ctypes as C
    import hashlib
    import numpy as np
    out = {}
    try:
        rec = {r["file"]: r for r in json.load(open(os.path.join(ROOT, "tests", "golden", "ref_images.json")))["files"]}
    except Exception as e:  # noqa: BLE001
        return {"error": repr(e)[:200]}
    L = zj.lib()
    L.zj_alloc_pinned.restype = C.c_void_p
    L.zj_alloc_pinned.argtypes = [C.c_size_t]
    L.zj_free_pinned.argtypes = [C.c_void_p]
    cases = (("speed_bench.jpg", zj.ColorSpace.RGB, "sha256_rgb", 62.246, 98.343, "Benches.md:82-86"),
             ("speed_bench.jpg", zj.ColorSpace.GRAYSCALE, "sha256_gray", 45.598, 46.648, "Benches.md:96-99"),
             ("speed_bench_hv_subsampling.jpg", zj.ColorSpace.RGB, "sha256_rgb", 52.175, 78.343, "Benches.md:135-139"))
    for name, cs, key, pub, turbo, where in cases:
        label = name + (" -> GRAYSCALE" if cs == zj.ColorSpace.GRAYSCALE else " -> RGB")
        try:
            data = open(os.path.join(ROOT, "tests", "golden", "ref", name), "rb").read()
            r = rec["benches/images/" + name]
            n = r["width"] * r["height"] * cs.num_components()
            fresh = 1e9
            for _ in range(reps):
                t0 = time.perf_counter()
                o = zj.ZuneJpegOptions()
                o.out_colorspace = cs
                dec = zj.Decoder(o, ctx)
                px = dec.decode_buffer(data)
                fresh = min(fresh, time.perf_counter() - t0)
                dec.close()
            ok = hashlib.sha256(np.ascontiguousarray(px).tobytes()).hexdigest() == r[key]
            del px
            # the same with pinned planes: the library keeps freed pinned blocks for the next decoder (zj_alloc_pinned)
            fresh_pinned = 1e9
            for _ in range(reps + 1):
                t0 = time.perf_counter()
                o = zj.ZuneJpegOptions()
                o.out_colorspace, o.pinned_planes = cs, True
                dec = zj.Decoder(o, ctx)
                px = dec.decode_buffer(data)
                fresh_pinned = min(fresh_pinned, time.perf_counter() - t0)
                dec.close()
            ok = ok and hashlib.sha256(np.ascontiguousarray(px).tobytes()).hexdigest() == r[key]
            del px
            o = zj.ZuneJpegOptions()
            o.out_colorspace, o.pinned_planes = cs, True
            dec = zj.Decoder(o, ctx)
            pin = L.zj_alloc_pinned(n)
            try:
                pout = np.ctypeslib.as_array(C.cast(pin, C.POINTER(C.c_uint8)), shape=(n,))
                steady, host4 = 1e9, 1e9
                for _ in range(reps + 2):
                    t0 = time.perf_counter()
                    got = dec.decode_buffer(data, out=pout)
                    steady = min(steady, time.perf_counter() - t0)
                ok = ok and hashlib.sha256(np.ascontiguousarray(got).tobytes()).hexdigest() == r[key]
                for _ in range(3):
                    t0 = time.perf_counter()
                    dec.prepare(data)
                    host4 = min(host4, time.perf_counter() - t0)
                par = dec.parallel_mcus()
            finally:
                dec.close()
                L.zj_free_pinned(pin)
            mp = r["width"] * r["height"] / 1e6
            out[label] = {"as_the_reference_does_ms": round(fresh * 1e3, 2), "new_decoder_pinned_planes_ms": round(fresh_pinned * 1e3, 2),
                          "steady_state_ms": round(steady * 1e3, 2),
                          "steady_state_megapixels_per_s": round(mp / steady, 1), "host_entropy_ms": round(host4 * 1e3, 2),
                          "host_threads": 4, "mcus_decoded_in_parallel": int(par), "width": r["width"], "height": r["height"],
                          "sampling": f"{r['h_max']}x{r['v_max']}", "file_bytes": len(data), "sha256_matches_golden": bool(ok),
                          "published_ms": pub, "published_libjpeg_turbo_ms": turbo,
                          "published_on": "AMD Ryzen 5 4500U, 4 worker threads, " + where}
        except Exception as e:  # noqa: BLE001
            out[label] = {"error": repr(e)[:200]}
    out["what"] = ("the reference's benchmark images through the whole decode_buffer call (default options: four threads; a scan without "
                   "restart markers is entered at four points): as_the_reference_does_ms = new decoder + pageable planes + a fresh pixel "
                   "buffer per call, as benches/decode.rs does; new_decoder_pinned_planes_ms = the same with pinned planes (cached by the library between decoders); steady_state_ms = one decoder, pinned planes and pixels; best of "
                   f"{reps} / {reps + 2}")
    return out


def load_golden():
    try:
        g = json.load(open(os.path.join(ROOT, "tests", "golden", "checksums_seed1234.json")))
        return g
    except Exception:
        return None


def golden_match(all_sums, shard_frames, golden, first_rank=0):
    """all_sums[r][j] = checksum of frame j of rank (first_rank + r)'s shard = global frame (first_rank + r) * shard_frames + j
    (frames beyond the 1024 of configs[4] wrap around): True iff every one equals the oracle's
    (tests/golden/checksums_seed1234.json), None if the golden file does not reach that far."""
    gl = [int(x, 16) for x in golden["rgb"]]
    idx = [((first_rank + r) * shard_frames + j) % 1024 for r in range(len(all_sums)) for j in range(len(all_sums[r]))]
    if not idx or max(idx) >= len(gl):
        return None
    return all(c == gl[i] for c, i in zip((c for r in all_sums for c in r), idx))


def launch_ranks(n, argv, rank_timeout=600.0):
    """`python bench.py --gpus N` outside torchrun: start the N ranks as CHILD processes (this parent never imports torch
    or touches a GPU), relay rank 0's stdout, fail if any rank fails -- or if the ranks are not all done `rank_timeout`
    seconds after the launch (a rank hung in RCCL initialisation must not cost the caller its whole time budget): the
    children still alive are named on stderr, terminated (exact PIDs; fresh processes, nothing is ever re-exec'd) and the
    exit code is 124.
    Rendezvous: no port is probed here (a probe socket closed before the children bind is a race).  Rank 0 binds port 0
    itself and publishes the kernel's choice through a file the other ranks wait for (shard._store_from_port_file)."""
    import subprocess
    import tempfile
    import threading
    tmpdir = tempfile.mkdtemp(prefix="zj_bench_")
    port_file = os.path.join(tmpdir, "port")
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", ZJ_BENCH_PORT_FILE=port_file)
        env.pop("MASTER_PORT", None)
        # The pool's host driver supports only dmabuf IPC: with the legacy mode RCCL's (and torch's) cross-process device
        # memory sharing fails with `hipIpcGetMemHandle: invalid argument`.  The image exports this variable already; it is
        # repeated here only so that a caller with a scrubbed environment gets the same ranks.  A/B on one GPU
        # (profiles/r04_probe_toolchain_rccl.txt): RCCL initialises and runs its collectives with 0, 1 and unset alike --
        # the setting matters for peer-to-peer mappings only, i.e. from 2 GPUs up.
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=env,
                                      stdout=subprocess.PIPE if r == 0 else sys.stderr))
    lines = []
    t = threading.Thread(target=lambda: lines.extend(procs[0].stdout.read().decode().splitlines()), daemon=True)
    t.start()
    rc = 0
    live = set(range(n))
    t0 = time.monotonic()
    while live and rc == 0:
        time.sleep(0.2)
        for r in sorted(live):
            c = procs[r].poll()
            if c is not None:
                live.discard(r)
                if c != 0:
                    rc = c if c > 0 else 1
                    print(f"bench.py: rank {r} exited with {c}", file=sys.stderr)
        if live and rc == 0 and time.monotonic() - t0 > rank_timeout:
            rc = 124
            print(f"bench.py: rank(s) {sorted(live)} still running {rank_timeout:.0f} s after the launch (--rank-timeout): "
                  f"terminating them", file=sys.stderr)
    for r in live:  # a rank failed or hung: stop the others (exact PIDs)
        procs[r].terminate()
    for p in procs:
        try:
            p.wait(timeout=10)
        except Exception:  # noqa: BLE001
            p.kill()
    t.join(5)
    try:
        import shutil
        shutil.rmtree(tmpdir, ignore_errors=True)
    except Exception:  # noqa: BLE001
        pass
    for ln in lines:
        print(ln, flush=True)
    if rc == 0 and not any(ln.startswith("{") for ln in lines):
        print("bench.py: rank 0 printed no result line", file=sys.stderr)
        rc = 1
    sys.exit(rc)


WORKLOADS = {  # name: (h_samp, v_samp, output colourspace name, algorithmic bytes per pixel, description)
    "420-rgb": (2, 2, "RGB", 6.0, "4096x4096 baseline 4:2:0, dequant+IDCT+h2v2+YCbCr->RGB, planes resident in HBM (BASELINE configs[1]; sharded = configs[4])"),
    "444-rgb": (1, 1, "RGB", 9.0, "configs[2]: 4096x4096 baseline 4:4:4, dequant+IDCT+YCbCr->RGB, planes resident in HBM"),
    "444-gray": (1, 1, "GRAYSCALE", 3.0, "configs[2]: 4096x4096 4:4:4 -> GRAYSCALE (luma only), planes resident in HBM"),
    "422-rgb": (2, 1, "RGB", 7.0, "4096x4096 baseline 4:2:2 (h2v1), dequant+IDCT+horizontal upsample+YCbCr->RGB (the reference's benches/decode.rs horizontal case)"),
    "440-rgb": (1, 2, "RGB", 7.0, "4096x4096 baseline 4:4:0 (h1v2), dequant+IDCT+vertical upsample+YCbCr->RGB (the reference's benches/decode.rs vertical case)"),
    "420-rgba": (2, 2, "RGBA", 7.0, "extension: 4096x4096 4:2:0 -> RGBA (R G B 255), planes resident in HBM"),
    "420-chw": (2, 2, "RGB", 6.0, "extension: 4096x4096 4:2:0 -> planar u8 RGB (C x H x W), planes resident in HBM"),
    # a ragged width (not a multiple of 16): the size of the reference's own medium test images (tests/medium_images.rs);
    # rows start at any byte, the row ends follow the reference's any-width rules (worker.rs:143-251)
    "420-rgb-2500x1786": (2, 2, "RGB", 6.0, "2500x1786 baseline 4:2:0 -> RGB (the reference's medium image size, a ragged width), planes resident in HBM", (2500, 1786)),
    "444-rgb-2500x1786": (1, 1, "RGB", 9.0, "2500x1786 baseline 4:4:4 -> RGB (ragged width), planes resident in HBM", (2500, 1786)),
    # the same frames with their rows at a pitch of 7552 bytes (zj_frame_desc.out_pitch, a multiple of 128): every tile's row
    # segment on whole cache lines; an extension for outputs that stay in HBM, the reference's layout is the tight one
    "420-rgb-2500x1786-pitch128": (2, 2, "RGB", 6.0, "2500x1786 baseline 4:2:0 -> RGB, output rows at a pitch of 7552 bytes (multiple of 128; extension), planes resident in HBM", (2500, 1786), 128),
}


def workload_dims(name):
    wl = WORKLOADS[name]
    return wl[5] if len(wl) > 5 else (4096, 4096)


def main():
    global W, H  # the frame size follows --workload (4096 x 4096 unless the workload names another)
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=300)
    ap.add_argument("--warmup", type=int, default=100,
                    help="untimed steps; the first ~70 launches after idle run 5-35%% slow while the clocks settle")
    ap.add_argument("--min-untimed", type=int, default=100, help="at least this many untimed launches precede the timed region")
    ap.add_argument("--frames", type=int, default=16, help="4096x4096 frames per launch (= per step) per GPU")
    ap.add_argument("--shard-frames", type=int, default=128,
                    help="distinct frames resident per GPU (configs[4]: 1024 frames over 8 GPUs = 128); steps walk the shard")
    ap.add_argument("--legacy-data", action="store_true",
                    help="rounds 1-2 input: two numpy-generated frames (synth.make_frame) tiled to one 16-frame batch")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-single-frame", action="store_true",
                    help="skip the one-frame-per-launch timing (configs[1] read literally); for profiler runs that should "
                         "see a single launch shape")
    ap.add_argument("--no-live-traffic", action="store_true", help="do not run the rocprofv3 --pmc child passes")
    ap.add_argument("--child", action="store_true", help="(internal) a profiler child: kernel launches only, no extras")
    ap.add_argument("--rank-timeout", type=float, default=600.0,
                    help="self-launched ranks (--gpus N outside torchrun): seconds after which ranks still running are "
                         "terminated and the run fails with exit code 124")
    ap.add_argument("--gather-rgb", action="store_true",
                    help="after the timed region: every rank's decoded RGB frames gathered to rank 0 (RCCL over xGMI), "
                         "reported separately as `gather_rgb`, never part of `value` (SURVEY.md 8e, optional)")
    ap.add_argument("--no-e2e", action="store_true", help="skip the pinned-host end-to-end figure (e2e_pinned)")
    ap.add_argument("--no-prewarm", action="store_true", help="do not settle the clocks with another kernel before the first launch")
    ap.add_argument("--no-dense-control", action="store_true", help="skip roofline.dense_control (profiler runs that should see only the shard's launches)")
    ap.add_argument("--workload", choices=sorted(WORKLOADS), default="420-rgb",
                    help="420-rgb = BASELINE.json configs[1]/[4] (the headline); 444-* are configs[2]; 422 / 440 the reference's "
                         "other sampling modes; 420-rgba / 420-chw are the output extensions (4 B/px interleaved, planar u8)")
    ap.add_argument("--variant", choices=["packed", "wide", "packed-direct"], default=None, help="kernel variant (default: library default)")
    ap.add_argument("--as-rank", default=None, metavar="R/N",
                    help="one process on one GPU playing rank R of an N-rank run: its shard [R*S, (R+1)*S), its golden offsets; "
                         "collectives at world size 1 (virtual ranks: everything about configs[4] one GPU can prove)")
    ap.add_argument("--no-other-workloads", action="store_true", help="skip other_workloads / reference_files / from_files / scattered_batch")
    args = ap.parse_args()
    if args.child:
        args.no_cpu_baseline = args.no_single_frame = args.no_live_traffic = args.no_e2e = True
        # a profiler child serialises every dispatch: nothing but the shard's launches (no pre-warm passes, no control)
        args.no_prewarm = args.no_dense_control = args.no_other_workloads = True
    virt = None
    if args.as_rank:
        try:
            vr, vn = (int(x) for x in args.as_rank.split("/"))
            assert 0 <= vr < vn <= 1024
        except Exception:  # noqa: BLE001
            sys.exit("--as-rank wants R/N with 0 <= R < N")
        if args.gpus != 1 or "WORLD_SIZE" in os.environ and os.environ["WORLD_SIZE"] != "1":
            sys.exit("--as-rank is a single-process mode")
        virt = (vr, vn)

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        launch_ranks(args.gpus, sys.argv[1:], args.rank_timeout)  # does not return
    if os.environ.get("ZJ_BENCH_TEST_HANG_RANK") in (os.environ.get("RANK", "0"), "all"):
        time.sleep(1e6)  # test knob (tests/test_dist.py): this rank never reaches the rendezvous

    import numpy as np
    import torch  # before libzjhip: both bind the same libamdhip64.so.7
    zj = importlib.import_module("zune-jpeg_amd")
    synth = importlib.import_module("zune-jpeg_amd.synth")
    shard = importlib.import_module("zune-jpeg_amd.shard")

    rank, local_rank, world = shard.env_world()
    args.gpus = world
    if not torch.cuda.is_available():
        sys.exit("bench.py needs a GPU: the product path has no CPU fallback")
    # ZJ_BENCH_SAME_GPU=1 is a plumbing test only (tests the multi-rank control flow on a 1-GPU box: every rank uses
    # cuda:0 and the collectives run on gloo/CPU tensors); the driver's runs use one GPU per rank over RCCL
    same_gpu = os.environ.get("ZJ_BENCH_SAME_GPU") == "1"
    gpu_index = 0 if same_gpu else local_rank
    torch.cuda.set_device(gpu_index)
    # this rank's host side next to its GPU (zj_numa.cpp): the main thread now, and with it every thread started from here on
    # (RCCL's, the pinned-plane feeders of the side measurements).  ZJ_NUMA=off leaves the rank where the launcher put it.
    numa_bound = zj.bind_thread_near_device(gpu_index)
    numa_here = (zj.device_numa_node(gpu_index), zj.thread_numa_node(), numa_bound >= 0)
    dev = torch.device("cuda", gpu_index)
    backend = "gloo" if same_gpu else "nccl"
    port_file = os.environ.get("ZJ_BENCH_PORT_FILE")
    if world == 1 and args.gather_rgb and not port_file and "MASTER_PORT" not in os.environ:
        import tempfile
        port_file = os.path.join(tempfile.mkdtemp(prefix="zj_bench_"), "port")  # N = 1 plumbing run of the frame gather
    shard.init_process_group(backend, rank, world, force=args.gather_rgb and not args.child, port_file=port_file,
                             timeout_s=args.rank_timeout + 60,  # the launcher's deadline (rank_timeout) comes first and names the ranks
                             device_id=None if same_gpu else dev)
    coll_dev = "cpu" if same_gpu else dev

    B = args.frames
    S = B if args.legacy_data else max(B, args.shard_frames - args.shard_frames % B)
    nsub = S // B
    ctx = zj.Context(zj.BACKEND_HIP, gpu_index)
    if args.variant:
        ctx.set_variant({"packed": 0, "wide": 1, "packed-direct": 2}[args.variant])
    hs, vs, cs_name, bytes_per_px, what = WORKLOADS[args.workload][:5]
    W, H = workload_dims(args.workload)
    out_cs = getattr(zj.ColorSpace, cs_name)
    ncomp_out = out_cs.num_components()
    # every rank decodes its own contiguous shard [lo, lo + S) of the global batch (image-level sharding)
    lo, hi = shard.shard_range(S * virt[1], virt[0], virt[1]) if virt else shard.shard_range(S * world, rank, world)
    assert hi - lo == S
    golden = load_golden() if (args.workload == "420-rgb" and not args.legacy_data) else None
    plane_elems = [synth.plane_blocks(W, H, hs, vs, c)[0] * synth.plane_blocks(W, H, hs, vs, c)[1] * 64 for c in range(3)]
    t_gen = time.perf_counter()
    d_planes = [torch.empty(S * n, dtype=torch.int16, device=dev) for n in plane_elems]
    if args.legacy_data:
        frames = [synth.make_frame(W, H, hs, vs, 3, seed=1234, frame_index=i) for i in range(2)]
        qts = frames[0][1]
        for c in range(3):
            d_planes[c].copy_(torch.from_numpy(np.concatenate([frames[i % 2][0][c] for i in range(B)])))
        first_planes = frames[0][0]
    else:
        for j in range(S):
            _, qts = synth.make_frame_t(W, H, hs, vs, 3, seed=1234, frame_index=(lo + j) % 1024, device=dev,
                                        out=[d_planes[c][j * plane_elems[c]:(j + 1) * plane_elems[c]] for c in range(3)])
        first_planes = None
    torch.cuda.synchronize()
    t_gen = time.perf_counter() - t_gen
    wl_align = WORKLOADS[args.workload][6] if len(WORKLOADS[args.workload]) > 6 else 0
    out_pitch = (W * ncomp_out + wl_align - 1) // wl_align * wl_align if wl_align else 0   # zj_frame_desc.out_pitch (0 = tight rows)
    desc = zj.FrameDesc.make(W, H, hs, vs, 3, out_cs, qts, out_layout=zj.LAYOUT_CHW if args.workload == "420-chw" else zj.LAYOUT_HWC,
                             out_pitch=out_pitch)
    frame_out = out_pitch * H if out_pitch else W * H * ncomp_out
    if out_pitch:
        args.no_e2e = True   # host outputs are tight (the pipeline refuses a padded pitch)
    d_out = torch.empty(S * frame_out, dtype=torch.uint8, device=dev)
    # a dedicated stream: launches on the legacy NULL stream serialise against every blocking stream
    # and cost ~30 us each (tools/launch_overhead.py); torch.cuda.synchronize() still covers it
    torch.cuda.synchronize()
    side = torch.cuda.Stream(device=dev)
    stream = side.cuda_stream
    base = [t.data_ptr() for t in d_planes] + [d_out.data_ptr()]
    stride = [2 * n * B for n in plane_elems] + [frame_out * B]          # bytes per sub-batch of B frames

    def sub(k):
        return [base[i] + (k % nsub) * stride[i] for i in range(4)]

    subs = [sub(k) for k in range(nsub)]

    def step(k):
        p = subs[k % nsub]
        ctx.decode_planes_device(desc, B, p[0], p[1], p[2], p[3], stream)

    # The first ~70 launches after idle run 5-35 % slow while the clocks settle.  Two measures: (1) half a second of another
    # kernel (a torch elementwise pass over the output buffer, which the decode overwrites anyway) brings the clocks up BEFORE
    # the first fused launch, so that a profiler's average over ALL launches of the fused kernel (rocprofv3 --stats,
    # profiles/*kernel_stats.csv) describes the same steady state as the HIP events do; (2) whatever --warmup says, at least
    # --min-untimed (100) untimed launches precede the timed region (the W warmup steps are part of them).
    if not args.no_prewarm:
        t_pw = time.perf_counter()
        with torch.cuda.stream(side):
            while time.perf_counter() - t_pw < 0.5:
                for _ in range(8):
                    d_out[:B * frame_out].add_(1)
                side.synchronize()
    untimed = max(args.warmup, args.min_untimed, nsub)
    for k in range(untimed):
        step(k)
    torch.cuda.synchronize()
    shard.barrier(world)
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
    t0 = time.perf_counter()
    ev[0].record(side)  # HIP events on the launch stream, around exactly the timed region's K launches
    for k in range(args.steps):
        step(k)
    ev[1].record(side)
    torch.cuda.synchronize()
    el_local = time.perf_counter() - t0
    shard.barrier(world)
    elapsed = shard.max_over_ranks(time.perf_counter() - t0, world, coll_dev)
    per_rank_ms = shard.gather_values(el_local / args.steps * 1e3, world, coll_dev)
    kernel_ms_region = ev[0].elapsed_time(ev[1]) / args.steps

    # dominant-kernel duration for the roofline: HIP events on the launch stream over >= 100 launches WALKING THE SHARD
    # like the timed steps do (whatever --steps says: the driver's --steps 20 would average 20).  When the timed region
    # itself has >= 100 launches its own event pair is the figure; otherwise a second, untimed run of 104+ launches.
    kiters = args.steps if (args.steps >= 100 or args.child) else -(-100 // nsub) * nsub
    if kiters == args.steps:
        kernel_ms = kernel_ms_region
    else:
        ev[2].record(side)
        for k in range(kiters):
            step(k)
        ev[3].record(side)
        ev[3].synchronize()
        kernel_ms = ev[2].elapsed_time(ev[3]) / kiters
    # every launch bracketed by its own event pair (what a profiler reports per dispatch), a few per sub-batch
    p0 = subs[0]
    each, kname = [], None
    for k in range(1 if args.child else nsub):
        pk = subs[k]
        _, e1, kname = ctx.time_decode_device(desc, B, pk[0], pk[1], pk[2], pk[3], 2 if args.child else max(2, -(-48 // nsub)), stream)
        each.append(e1)
    kernel_ms_each = sum(each) / len(each)
    # every rank's own figure: the roofline fraction below is the SLOWEST rank's, not rank 0's
    per_rank_kernel_ms = shard.gather_values(kernel_ms, world, coll_dev)
    per_rank_numa = [{"device_node": int(a), "thread_node": int(b), "bound": bool(c)} for a, b, c in
                     zip(*(shard.gather_values(float(v), world, coll_dev) for v in numa_here))]
    kernel_ms_rank0, kernel_ms = kernel_ms, max(per_rank_kernel_ms)
    # Control for data dependence: the kernel's only data-dependent shortcut is the reference's own DC-only one, and it is
    # taken lane by lane (a wave transforms as long as any of its 64 blocks needs it).  The same sub-batch with ONE coefficient
    # (row 1, column 7) set to 1 in every block -- no DC-only block, nothing sparse about any column -- must therefore decode at
    # nearly the same rate (measured: within 3 %; the output differs, of course).  Reported beside the headline, never as it.
    dense_ms = None
    if not args.child and not args.no_dense_control:
        dpl = [d_planes[c][:B * plane_elems[c]].clone() for c in range(3)]
        for t_ in dpl:
            t_.view(-1, 64)[:, 15] = 1
        dp = [t_.data_ptr() for t_ in dpl]
        for _ in range(20):
            ctx.decode_planes_device(desc, B, dp[0], dp[1], dp[2], subs[0][3], stream)
        ev[2].record(side)
        for _ in range(100):
            ctx.decode_planes_device(desc, B, dp[0], dp[1], dp[2], subs[0][3], stream)
        ev[3].record(side)
        ev[3].synchronize()
        dense_ms = ev[2].elapsed_time(ev[3]) / 100
        del dpl
    # A yardstick from the same box, same minute: a plain device-to-device copy of one launch's bytes (as many read as the
    # kernel reads, as many written as it writes) by torch's Tensor.copy_, timed like the kernel.  8 TB/s is the pin rate; a
    # hand-written streaming copy reaches ~6.3 TB/s on this part (MI355X_MICROARCH.md, tools/store_probe), torch's 5.1-5.2
    # (the fused kernel decodes 1.10 x faster than torch copies the same bytes).  Never `value`, never `frac`'s denominator.
    copy_gbs = None
    if not args.child and not args.no_dense_control:
        try:
            half = int(B * W * H * bytes_per_px) // 2 // 16 * 16
            src_t = torch.empty(half, dtype=torch.uint8, device=dev)
            dst_t = torch.empty(half, dtype=torch.uint8, device=dev)
            with torch.cuda.stream(side):
                for _ in range(10):
                    dst_t.copy_(src_t)
                ev[2].record(side)
                for _ in range(30):
                    dst_t.copy_(src_t)
                ev[3].record(side)
            ev[3].synchronize()
            copy_gbs = 2 * half / (ev[2].elapsed_time(ev[3]) / 30 * 1e-3) / 1e9
            del src_t, dst_t
        except Exception:  # noqa: BLE001 -- a yardstick, not the measurement
            copy_gbs = None
    # configs[1] read literally is ONE 4096x4096 frame per launch: time that shape too, reported beside the batched figure
    one_ms = one_ms_each = one_ms_4s = None
    if not args.no_single_frame:
        fstr = [2 * n for n in plane_elems] + [frame_out]

        def one(i, st):
            f = i % S  # successive launches walk the shard's frames
            ctx.decode_planes_device(desc, 1, base[0] + f * fstr[0], base[1] + f * fstr[1], base[2] + f * fstr[2],
                                     base[3] + f * fstr[3], st)
        for i in range(50):
            one(i, stream)
        ev[2].record(side)
        for i in range(384):  # ONE caller stream, launches back to back
            one(i, stream)
        ev[3].record(side)
        ev[3].synchronize()
        one_ms = ev[2].elapsed_time(ev[3]) / 384
        eachs = []
        for f in range(0, S, max(1, S // 8)):  # isolated launches (own event pair each) on 8 frames of the shard
            _, e1, _ = ctx.time_decode_device(desc, 1, base[0] + f * fstr[0], base[1] + f * fstr[1], base[2] + f * fstr[2],
                                              base[3] + f * fstr[3], 24, stream)
            eachs.append(e1)
        one_ms_each = sum(eachs) / len(eachs)
        # the same shape fed the way a frame-at-a-time caller would: launches rotating over four streams, so the tail
        # of one frame overlaps the head of the next (tools/single_frame_streams.py)
        four = [torch.cuda.Stream(device=dev) for _ in range(4)]

        def rot(n):
            for i in range(n):
                one(i, four[i % 4].cuda_stream)
        rot(200)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        rot(2000)
        torch.cuda.synchronize()
        one_ms_4s = (time.perf_counter() - t1) / 2000 * 1e3
    # The scattered form (zj_decode_frames_device): 16 NON-adjacent frames of the shard, each named by its own four
    # pointers, in ONE launch -- what a caller who owns its frames as independent allocations gets (the reference's callers
    # do: src/mcu.rs:238-250, src/decoder.rs:178).  Eight different irregular frame sets in rotation; every frame is
    # decoded to its own place in d_out, so the checksums below also cover this path's bytes.
    scat_ms = scat_adj_ms = None
    if not args.child and not args.no_other_workloads and S >= 2 * B:
        fstr = [2 * n for n in plane_elems] + [frame_out]
        rng = np.random.default_rng(5)
        sets = []
        for _ in range(8):
            idx = [int(v) for v in rng.permutation(S)[:B]]
            sets.append([[base[i] + f * fstr[i] for f in idx] for i in range(4)])
        # (the one-frame launches above leave the clocks low: the same >= 100 untimed launches as in front of the timed region)
        for k in range(max(args.min_untimed, 24)):
            q = sets[k % 8]
            ctx.decode_frames_device(desc, q[0], q[1], q[2], q[3], stream)
        ev[2].record(side)
        for k in range(104):
            q = sets[k % 8]
            ctx.decode_frames_device(desc, q[0], q[1], q[2], q[3], stream)
        ev[3].record(side)
        ev[3].synchronize()
        scat_ms = ev[2].elapsed_time(ev[3]) / 104
        # control: the SAME frames a contiguous step decodes (sub-batch k), named one by one in a shuffled order -- what the
        # pointer table itself costs, apart from where in HBM the frames lie
        adj = []
        for k in range(nsub):
            idx = [k * B + int(v) for v in rng.permutation(B)]
            adj.append([[base[i] + f * fstr[i] for f in idx] for i in range(4)])
        for k in range(nsub):
            ctx.decode_frames_device(desc, adj[k][0], adj[k][1], adj[k][2], adj[k][3], stream)
        ev[2].record(side)
        for k in range(104):
            q = adj[k % nsub]
            ctx.decode_frames_device(desc, q[0], q[1], q[2], q[3], stream)
        ev[3].record(side)
        ev[3].synchronize()
        scat_adj_ms = ev[2].elapsed_time(ev[3]) / 104
    # every frame of the shard decoded once more (untimed), then the trivial gather: per-frame checksums, computed on the
    # GPU, gathered over RCCL and compared with the oracle's (tests/golden/checksums_seed1234.json)
    for k in range(nsub):
        step(k)
    torch.cuda.synchronize()
    sums = []
    scat_same = None
    if not args.child:
        wts = synth.checksum_weights_t(frame_out // 8, dev)
        sums = [synth.frame_checksum_t(d_out[j * frame_out:(j + 1) * frame_out], wts) for j in range(S)]
        if scat_ms is not None:
            # ... and once more through the scattered form, over a cleared output: frames [k*B, (k+1)*B) in reversed order
            # (two of them swapped: not equally spaced, so the strided shortcut cannot take it), one launch per sub-batch
            d_out.zero_()
            torch.cuda.synchronize()  # (zero_ ran on torch's stream, the launches go to `stream`)
            fstr = [2 * n for n in plane_elems] + [frame_out]
            for k in range(nsub):
                idx = list(range(k * B, (k + 1) * B))[::-1]
                idx[0], idx[3 % B] = idx[3 % B], idx[0]
                q = [[base[i] + f * fstr[i] for f in idx] for i in range(4)]
                ctx.decode_frames_device(desc, q[0], q[1], q[2], q[3], stream)
            torch.cuda.synchronize()
            scat_same = sums == [synth.frame_checksum_t(d_out[j * frame_out:(j + 1) * frame_out], wts) for j in range(S)]
        del wts
    all_sums = shard.gather_checksums(sums, world, coll_dev)
    match = golden_match(all_sums, S, golden, first_rank=virt[0] if virt else 0) if (golden and sums) else None
    # SURVEY.md 8e, optional: the decoded frames themselves to rank 0 -- the only step that puts real bytes on xGMI.
    # Measured on its own, after everything else; never part of `value`.
    gather_rgb = None
    if args.gather_rgb and not args.child:
        torch.cuda.empty_cache()
        try:
            g_in = d_out if not same_gpu else d_out.cpu()
            outs, g_s = shard.gather_frames(g_in, rank, world, always=True)
            g_s = shard.max_over_ranks(g_s, world, coll_dev)
            ok = None
            if rank == 0 and outs is not None and golden:
                # spot check: the LAST frame of the last rank's shard, as it arrived on rank 0
                last = outs[-1][(S - 1) * frame_out:].to(dev)
                ok = synth.frame_checksum_t(last) == int(golden["rgb"][((world - 1) * S + S - 1) % 1024], 16)
            remote = (world - 1) * S * frame_out
            gather_rgb = {"seconds": round(g_s, 4), "bytes_total": world * S * frame_out, "bytes_remote": remote,
                          "gbs_remote": round(remote / g_s / 1e9, 1) if (g_s > 0 and remote) else None,
                          "backend": backend, "last_frame_matches_golden": ok,
                          "what": f"dist.gather of every rank's {S} decoded RGB frames ({S * frame_out / 1e9:.2f} GB) to rank 0"}
            del outs
        except Exception as e:  # noqa: BLE001 -- optional measurement; the headline must not depend on it
            gather_rgb = {"error": repr(e)[:300]}

    if rank == 0:
        mp_total = world * B * args.steps * W * H / 1e6
        algo_bytes = B * W * H * bytes_per_px
        achieved = algo_bytes / (kernel_ms * 1e-3) / 1e9
        tr = load_traffic()
        if tr and not (tr.get("workload", "420-rgb") == args.workload and tr.get("frames_per_launch", 16) == B):
            tr = None  # the committed counters describe another launch shape
        live = {} if (args.no_live_traffic or world > 1) else live_traffic(args.workload, B, S)
        live_hbm = live if live.get("hbm_bytes_per_launch") else None
        # Second bound, reported beside the HBM one: integer VALU issue.  A wave64 integer instruction occupies
        # its SIMD for 4 cycles (16 lanes per SIMD per clock; profiles/r01_ubench_valu_issue_cost.txt), so the
        # chip retires at most 1024 SIMDs x 2.4 GHz / 4 wave-instructions per second.
        valu = None
        sq = live if live.get("sq_insts_valu_per_launch") else tr
        if sq and sq.get("sq_insts_valu_per_launch"):
            peak_wi = 1024 * 2.4e9 / 4.0
            ach_wi = sq["sq_insts_valu_per_launch"] / (kernel_ms * 1e-3)
            valu = {"wave_insts_per_launch": sq["sq_insts_valu_per_launch"], "achieved": round(ach_wi / 1e9, 1),
                    "peak": round(peak_wi / 1e9, 1), "unit": "G wave64-instructions/s", "frac": round(ach_wi / peak_wi, 4),
                    "source": sq.get("sq_source"), "replayed": sq is not live}
        traffic = live_hbm["hbm_bytes_per_launch"] if live_hbm else (tr or {}).get("hbm_bytes_per_launch")
        res = {
            "metric": "megapixels/sec decoded (IDCT->RGB), 4K 4:2:0 baseline" if args.workload == "420-rgb" else f"megapixels/sec decoded, 4K {args.workload}",
            "value": round(mp_total / elapsed, 1),
            "unit": "megapixels/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "i32",  # the IDCT's fixed-point type (i16 coefficients in, packed-i16 colour math, u8 pixels out)
            "data": "synthetic",
            "config": {"workload": what,
                       "frames_per_gpu_per_step": B, "frames_resident_per_gpu": S, "frames_total": S * world,
                       "sharding": f"image-level x{world}, rank r holds global frames [r*{S}, (r+1)*{S}), no data-path collective",
                       "distinct_frames": 2 if args.legacy_data else S * world,
                       "generator": "synth.make_frame (numpy, rounds 1-2)" if args.legacy_data else
                                    "synth.make_frame_t(seed 1234, frame_index = global frame), integer-only, generated on the GPU",
                       "generate_s": round(t_gen, 2), "untimed_launches_before_timing": untimed},
            "roofline": {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 4),
                         "traffic": traffic,
                         "traffic_replayed": (not live_hbm) if traffic is not None else None,
                         "traffic_source": live_hbm["source"] if live_hbm else (tr or {}).get("source"),
                         "live_counters_dropped": live.get("dropped") or ("N > 1: the counter passes run at N = 1 only" if world > 1 else
                                                                          ("--no-live-traffic" if args.no_live_traffic else None)),
                         "kernel": kname, "kernel_ms": round(kernel_ms, 4), "kernel_ms_rank0": round(kernel_ms_rank0, 4),
                         "per_rank_kernel_ms": [round(v, 4) for v in per_rank_kernel_ms],
                         "per_rank_numa": per_rank_numa,
                         "kernel_ms_single_launch": round(kernel_ms_each, 4),
                         "kernel_launches_timed": kiters, "kernel_ms_timed_region": round(kernel_ms_region, 4),
                         "kernel_timing": f"HIP events on the launch stream; {kiters} launches walking the shard's {nsub} "
                                          f"sub-batches of {B} frames (>= 100 whatever --steps); kernel_ms_timed_region = "
                                          f"events around the {args.steps} timed steps; kernel_ms_single_launch = own event "
                                          f"pair per launch; kernel_ms (and frac) = the slowest rank's, per_rank_kernel_ms = every rank's",
                         "algorithmic_bytes_per_launch": int(algo_bytes),
                         "dense_control": None if dense_ms is None else {
                             "kernel_ms": round(dense_ms, 4), "frac": round(algo_bytes / (dense_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                             "vs_kernel_ms": round(dense_ms / kernel_ms_rank0, 4),
                             "what": "the same 16 frames with coefficient (row 1, column 7) of every block set to 1: no DC-only "
                                     "block, no empty column anywhere; the rate stays within 3 % of the shard's (vs_kernel_ms)"},
                         "valu_issue": valu,
                         "same_box_copy": None if not copy_gbs else {
                             "gbs": round(copy_gbs, 1), "kernel_vs_copy": round(achieved / copy_gbs, 4),
                             "what": "torch's device-to-device copy (Tensor.copy_) of half a launch's algorithmic bytes -- as many read and "
                                     "written as the kernel reads and writes --, HIP events around 30 copies on the launch stream, same "
                                     "process: a yardstick from this box beside the 8 TB/s pin rate that `frac` is quoted against (a "
                                     "hand-written copy, tools/store_probe, reaches 6.2-6.3 TB/s warm)"},
                         "single_frame_launch": None if one_ms is None else {
                             "kernel_ms": round(one_ms, 4), "kernel_ms_single_launch": round(one_ms_each, 4),
                             "megapixels_per_s": round(W * H / 1e6 / (one_ms * 1e-3), 1),
                             "achieved": round(W * H * bytes_per_px / (one_ms * 1e-3) / 1e9, 1),
                             "frac": round(W * H * bytes_per_px / (one_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                             "four_streams_ms_per_frame": round(one_ms_4s, 4),
                             "four_streams_megapixels_per_s": round(W * H / 1e6 / (one_ms_4s * 1e-3), 1),
                             "four_streams_frac": round(W * H * bytes_per_px / (one_ms_4s * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)},
                         "scattered_batch": None if scat_ms is None else {
                             "kernel_ms": round(scat_ms, 4), "vs_kernel_ms": round(scat_ms / kernel_ms_rank0, 4),
                             "frac": round(algo_bytes / (scat_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                             "same_checksums_as_contiguous": scat_same,
                             "adjacent_frames_kernel_ms": round(scat_adj_ms, 4), "adjacent_frames_vs_kernel_ms": round(scat_adj_ms / kernel_ms_rank0, 4),
                             "what": f"zj_decode_frames_device: {B} non-adjacent frames of the shard per launch, each named by its own "
                                     f"four pointers (by value in the kernel arguments); 104 launches over 8 irregular frame sets; adjacent_frames_* = "
                                     f"the frames of a contiguous step, named one by one in shuffled order (the table's own cost)"}},
            "rccl_ranks": world if (world > 1 and backend == "nccl") else 0,
            "collective_backend": None if world == 1 else backend,
            "per_rank_ms": [round(v, 4) for v in per_rank_ms],
            "as_rank": None if not virt else {"rank": virt[0], "of": virt[1], "global_frames": [lo, hi]},
            "frames_checksummed": sum(len(r) for r in all_sums),
            "checksums_match_golden": match,
        }
        if not args.no_cpu_baseline and args.workload == "420-rgb":
            # host baseline, rank 0, "in the same run" at every N (north_star): ~20 s of CPU work at N = 1, ~8 s at N > 1,
            # where the other ranks sleep on the store meanwhile (shard.wait_for_rank0: no spinning on the host cores being timed)
            nf = 2 if args.legacy_data else min(8, S)
            cpu_frames = [[d_planes[c][j * plane_elems[c]:(j + 1) * plane_elems[c]].cpu().numpy() for c in range(3)]
                          for j in range(nf)]
            res["cpu_baseline"] = cpu_baseline(cpu_frames, qts, budget_s=20.0 if world == 1 else 8.0)
            del cpu_frames
        if not args.no_other_workloads and args.workload == "420-rgb" and world == 1 and not virt:
            res["other_workloads"] = other_workloads(zj, synth, ctx, dev, side)
            res["reference_files"] = reference_files(zj, ctx)
            res["from_files"] = from_files(zj, ctx)
            res["reference_bench"] = reference_bench(zj, ctx)
        if not args.no_e2e and args.workload == "420-rgb" and world == 1 and not virt:
            gsums = [int(x, 16) for x in golden["rgb"][lo:lo + 8]] if golden else None
            res["e2e_pinned"] = e2e_pinned(zj, ctx, desc, d_planes, plane_elems, frame_out, gsums, nb=min(8, S), probe=pcie_probe(dev))
        if gather_rgb is not None:
            res["gather_rgb"] = gather_rgb
        print(json.dumps(res), flush=True)
        shard.signal_from_rank0("zj_bench_rank0_done", world)
    else:
        shard.wait_for_rank0("zj_bench_rank0_done", world, timeout_s=args.rank_timeout)
    shard.barrier(world)
    ctx.close()
    import torch.distributed as dist
    if dist.is_initialized():
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
