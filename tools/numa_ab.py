#!/usr/bin/env python3
"""NUMA placement A/B on the GPU box (round 6): the same measurements with the rank / the pool's threads
  near   ZJ_NUMA on (default): bound to the GPU's node by the library
  off    ZJ_NUMA=off, the process left where the scheduler puts it (what every earlier round measured)
  far    ZJ_NUMA=off under `taskset` on the OTHER node's CPUs: what "off" gives on an unlucky day
Measured: bench.py's e2e_pinned (pinned planes -> GPU -> pinned pixels, 3 streams) and pcie_probe, and tools/files_bench.py's
pool of 16 entropy workers over 4096x4096 q90 files.  Prints one table for profiles/.
  python tools/numa_ab.py
"""
import glob
import importlib
import json
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def cpulist(node):
    try:
        return open(f"/sys/devices/system/node/node{node}/cpulist").read().strip()
    except OSError:
        return None


def main():
    zj = importlib.import_module("zune-jpeg_amd")
    node = zj.device_numa_node(0)
    nodes = sorted(int(os.path.basename(p)[4:]) for p in glob.glob("/sys/devices/system/node/node[0-9]*"))
    far = [n for n in nodes if n != node]
    print(f"device 0 on NUMA node {node}; nodes {nodes}")
    modes = [("near", {}, [])]
    modes.append(("off", {"ZJ_NUMA": "off"}, []))
    if node >= 0 and far and cpulist(far[0]):
        modes.append(("far", {"ZJ_NUMA": "off"}, ["taskset", "-c", cpulist(far[0])]))
        modes.append(("near-by-taskset", {"ZJ_NUMA": "off"}, ["taskset", "-c", cpulist(node)]))
    rows = []
    for name, env, prefix in modes:
        e = dict(os.environ, **env)
        if not env:
            e.pop("ZJ_NUMA", None)
        r = subprocess.run(prefix + [sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "10", "--warmup", "2", "--shard-frames", "16",
                                     "--no-cpu-baseline", "--no-live-traffic", "--no-single-frame", "--no-dense-control", "--no-other-workloads"],
                           capture_output=True, text=True, env=e, timeout=900)
        line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
        res = json.loads(line[-1]) if line else {}
        e2e = res.get("e2e_pinned", {})
        probe = e2e.get("pcie_probe", {})
        f = subprocess.run(prefix + [sys.executable, os.path.join(ROOT, "tools", "files_bench.py"), "--files", "48", "--distinct", "3", "--restart-rows", "0"],
                           capture_output=True, text=True, env=e, timeout=900)
        pool = [ln for ln in f.stdout.splitlines() if ln.startswith("zj_pool") and (" 16 workers" in ln or "  4 workers" in ln)]
        rows.append((name, res.get("roofline", {}).get("per_rank_numa"), e2e.get("megapixels_per_s"), probe.get("h2d_gbs"), probe.get("d2h_gbs"),
                     probe.get("duplex_gbs_per_direction"), pool, f.stdout[-1500:] if not pool else ""))
    print(f"{'mode':<18}{'rank placement':<58}{'e2e MP/s':>10}{'h2d':>8}{'d2h':>8}{'duplex':>8}")
    for name, numa, mp, up, down, dup, pool, tail in rows:
        print(f"{name:<18}{json.dumps(numa):<58}{mp if mp is not None else 'n/a':>10}{up if up is not None else 'n/a':>8}{down if down is not None else 'n/a':>8}{dup if dup is not None else 'n/a':>8}")
        for ln in pool:
            print(f"{'':<18}{ln}")
        if tail:
            print(tail)


if __name__ == "__main__":
    main()
