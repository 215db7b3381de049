#!/usr/bin/env python3
"""The host feeder A/B (round 6): why is the baseline walker several times slower into pinned planes than
into heap memory on the GPU boxes?  One host thread, pinned to one CPU, for every combination of

  planes  : heap (aligned_alloc) | hipHostMalloc portable | default | portable+NumaUser | heap + hipHostRegister | portable+non-coherent
  stores  : ZJ_PLANE_STORE 0 nt16 | 1 plain | 2 in place | 3 nt32 | 4 nt64          (zj_jpeg.cpp: STORE_*)
  thread  : a CPU of the GPU's NUMA node | a CPU of another node (when the cgroup allows one)

it reports (a) the raw store / load rate of a 64 MB buffer of that kind and the node its pages live on, (b) zj_decoder_prepare
(container parsing + Huffman -> planes) on tests/golden/test-baseline.jpg, test-progressive.jpg and a 4096x4096 4:2:0 q90 file,
(c) the H2D copy rate out of that buffer.  Output: a table for profiles/r06_feeder_ab.txt.

  python tools/feeder_ab.py [--reps 7] [--quick]
"""
import argparse
import ctypes as C
import glob
import importlib
import os
import statistics
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
zj = importlib.import_module("zune-jpeg_amd")

KINDS = [("heap", None), ("hip-portable", 0), ("hip-default", 1), ("hip-numa-user", 2), ("registered", 3), ("hip-noncoherent", 4)]
STORES = [(0, "nt16"), (1, "plain"), (2, "in-place"), (3, "nt32"), (4, "nt64")]


def membench():
    so = os.path.join(ROOT, "tools", "membench", "libmembench.so")
    if not os.path.exists(so):
        subprocess.check_call(["gcc", "-O2", "-shared", "-fPIC", "-o", so, os.path.join(ROOT, "tools", "membench", "membench.c")])
    L = C.CDLL(so)
    for f in ("mb_fill_nt16", "mb_fill_nt32", "mb_fill_nt64", "mb_fill_plain16", "mb_blocks_nt16", "mb_memset"):
        getattr(L, f).restype = C.c_double
        getattr(L, f).argtypes = [C.c_void_p, C.c_size_t]
    L.mb_read.restype = C.c_double
    L.mb_read.argtypes = [C.c_void_p, C.c_size_t, C.POINTER(C.c_uint64)]
    L.mb_page_node.argtypes = [C.c_void_p]
    return L


def parse_cpulist(s):
    out = []
    for part in s.strip().split(","):
        if not part:
            continue
        a, _, b = part.partition("-")
        out += list(range(int(a), int(b or a) + 1))
    return out


def topology():
    nodes = {}
    for d in sorted(glob.glob("/sys/devices/system/node/node[0-9]*")):
        nodes[int(d.rsplit("node", 1)[1])] = parse_cpulist(open(d + "/cpulist").read())
    gpu_nodes = []
    for d in sorted(glob.glob("/sys/class/drm/card[0-9]*/device")):
        try:
            vendor = open(d + "/vendor").read().strip()
            node = int(open(d + "/numa_node").read())
            gpu_nodes.append((os.path.basename(os.path.dirname(d)), vendor, node, os.path.basename(os.path.realpath(d))))
        except Exception:  # noqa: BLE001
            pass
    return nodes, gpu_nodes


def flags_have(flag):
    try:
        return flag in open("/proc/cpuinfo").read().split("flags", 1)[1].split("\n", 1)[0].split()
    except Exception:  # noqa: BLE001
        return False


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=7)
    ap.add_argument("--quick", action="store_true", help="fewer combinations: kinds heap / portable / registered, stores 0 1 4")
    ap.add_argument("--no-gpu", action="store_true", help="dry run without a device: heap planes only")
    a = ap.parse_args()
    MB = membench()
    L = zj.lib()
    nodes, gpus = topology()
    allowed = sorted(os.sched_getaffinity(0))
    print(f"host: {os.cpu_count()} logical CPUs, {len(allowed)} allowed ({allowed[0]}..{allowed[-1]}), cgroup cpu.max = "
          f"{open('/sys/fs/cgroup/cpu.max').read().strip() if os.path.exists('/sys/fs/cgroup/cpu.max') else 'n/a'}")
    try:
        model = [l.split(":", 1)[1].strip() for l in open("/proc/cpuinfo") if l.startswith("model name")][0]
    except Exception:  # noqa: BLE001
        model = "?"
    print(f"cpu: {model}; avx512f {'yes' if flags_have('avx512f') else 'no'}")
    for n, cpus in nodes.items():
        ok = [c for c in cpus if c in allowed]
        print(f"  node {n}: {len(cpus)} CPUs ({cpus[0]}..{cpus[-1]}), {len(ok)} allowed")
    for g in gpus:
        print(f"  {g[0]}: vendor {g[1]} pci {g[3]} numa_node {g[2]}")
    ctx = None if a.no_gpu else zj.Context()
    bus = C.create_string_buffer(64)
    gpu_node = -1
    try:
        if a.no_gpu:
            raise RuntimeError("no device asked for")
        hip = C.CDLL("libamdhip64.so")
        if hip.hipDeviceGetPCIBusId(bus, 64, 0) == 0:
            pci = bus.value.decode().lower()
            gpu_node = int(open(f"/sys/bus/pci/devices/{pci}/numa_node").read())
            print(f"device 0: pci {pci}, numa_node {gpu_node}")
    except Exception as e:  # noqa: BLE001
        print("device 0: numa node unknown:", repr(e)[:100])
    if gpu_node < 0 and gpus:
        amd = [g for g in gpus if g[1] == "0x1002"]
        gpu_node = amd[0][2] if amd else -1
    near = [c for c in nodes.get(gpu_node, allowed) if c in allowed] or allowed
    far = [c for n, cpus in nodes.items() if n != gpu_node for c in cpus if c in allowed]
    places = [("near", near[len(near) // 2])]
    if far:
        places.append(("far", far[len(far) // 2]))
    print(f"thread places: {places} (gpu node {gpu_node})")

    files = []
    for n in ("test-baseline.jpg", "test-progressive.jpg"):
        files.append((n, open(os.path.join(ROOT, "tests", "golden", n), "rb").read()))
    import files_bench
    files.append(("q90-420-4096", files_bench.make_jpeg(4096, 0, 0)))
    kinds = [k for k in KINDS if not a.quick or k[0] in ("heap", "hip-portable", "registered")]
    if a.no_gpu:
        kinds = kinds[:1]
    stores = [s for s in STORES if not a.quick or s[0] in (0, 1, 4)]
    if not flags_have("avx512f"):
        stores = [s for s in stores if s[0] != 4]
    SZ = 64 << 20
    sink = C.c_uint64()
    d_buf = ctx.device_alloc(SZ) if ctx else None
    hipc = None
    try:
        if ctx:
            hipc = C.CDLL("libamdhip64.so")
            hipc.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
    except Exception:  # noqa: BLE001
        pass

    for place, cpu in places:
        os.sched_setaffinity(0, {cpu})
        time.sleep(0.05)
        print(f"\n=== thread on CPU {cpu} ({place} the GPU; now on cpu {MB.mb_cpu()}) ===")
        print("--- raw 64 MB buffer: GB/s, best of 5 (first pass faults the pages in and is dropped) ---")
        print(f"{'planes':<17}{'node':>5}{'nt16':>8}{'nt32':>8}{'nt64':>8}{'plain16':>9}{'memset':>8}{'blk-nt16':>10}{'read':>8}{'h2d':>8}")
        libc = C.CDLL(None)
        libc.aligned_alloc.restype = C.c_void_p
        libc.aligned_alloc.argtypes = [C.c_size_t, C.c_size_t]
        libc.free.argtypes = [C.c_void_p]
        L.zj_alloc_pinned.restype = C.c_void_p
        L.zj_alloc_pinned.argtypes = [C.c_size_t]
        L.zj_free_pinned.argtypes = [C.c_void_p]
        for kname, kind in kinds:
            if kind is None:
                p = libc.aligned_alloc(4096, SZ)
            else:
                os.environ["ZJ_PINNED_KIND"] = str(kind)
                p = L.zj_alloc_pinned(SZ)
            if not p:
                print(f"{kname:<17} allocation failed")
                continue
            MB.mb_memset(p, SZ)
            node = MB.mb_page_node(p + SZ // 2)
            row = []
            for fn in ("mb_fill_nt16", "mb_fill_nt32", "mb_fill_nt64", "mb_fill_plain16", "mb_memset", "mb_blocks_nt16"):
                if fn == "mb_fill_nt64" and not flags_have("avx512f"):
                    row.append(float("nan"))
                    continue
                row.append(SZ / min(getattr(MB, fn)(p, SZ) for _ in range(5)) / 1e9)
            row.append(SZ / min(MB.mb_read(p, SZ, C.byref(sink)) for _ in range(5)) / 1e9)
            h2d = float("nan")
            if hipc is not None:
                ts = []
                for _ in range(5):
                    t0 = time.perf_counter()
                    hipc.hipMemcpy(d_buf, p, SZ, 1)
                    ts.append(time.perf_counter() - t0)
                h2d = SZ / min(ts) / 1e9
            print(f"{kname:<17}{node:>5}" + "".join(f"{v:>{w}.2f}" for v, w in zip(row, (8, 8, 8, 9, 8, 10, 8))) + f"{h2d:>8.2f}")
            if kind is None:
                libc.free(p)
            else:
                L.zj_free_pinned(p)

        print(f"--- zj_decoder_prepare on one thread: ms min / median of {a.reps} (first pass, which allocates, dropped) ---")
        print(f"{'planes':<17}{'stores':<10}" + "".join(f"{n:>24}" for n, _ in files))
        for kname, kind in kinds:
            for sv, sname in stores:
                os.environ["ZJ_PLANE_STORE"] = str(sv)
                if kind is not None:
                    os.environ["ZJ_PINNED_KIND"] = str(kind)
                cells = []
                for fname, data in files:
                    o = zj.ZuneJpegOptions()
                    o.num_threads, o.pinned_planes = 1, kind is not None
                    dec = zj.Decoder(o, ctx)
                    ts = []
                    for _ in range(a.reps + 1):
                        t0 = time.perf_counter()
                        dec.prepare(data)
                        ts.append(time.perf_counter() - t0)
                    dec.close()
                    ts = ts[1:]
                    cells.append(f"{min(ts) * 1e3:>13.2f} /{statistics.median(ts) * 1e3:>8.2f}")
                print(f"{kname:<17}{sname:<10}" + "".join(cells), flush=True)
        os.sched_setaffinity(0, set(allowed))
    os.environ.pop("ZJ_PLANE_STORE", None)
    os.environ.pop("ZJ_PINNED_KIND", None)
    if ctx:
        ctx.device_free(d_buf)


if __name__ == "__main__":
    main()
