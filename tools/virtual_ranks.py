#!/usr/bin/env python3
"""Everything about BASELINE configs[4] that ONE GPU can prove: bench.py --as-rank R/N for R = 0..N-1, one after the other,
each a fresh process playing rank R of an N-rank run -- its shard [R*128, (R+1)*128) of the 1024 frames, generated with the
global frame indices, decoded and checked against the golden checksums at the global offsets.  Prints one line per rank and
a summary (profiles/r05_virtual_ranks.txt).

    python tools/virtual_ranks.py [--ranks 8] [--steps 20] [--warmup 5]
"""
import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ranks", type=int, default=8)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    a = ap.parse_args()
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE")}
    total, ok, values = 0, True, []
    t_all = time.time()
    for r in range(a.ranks):
        t0 = time.time()
        p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--as-rank", f"{r}/{a.ranks}", "--steps", str(a.steps),
                            "--warmup", str(a.warmup), "--no-cpu-baseline", "--no-e2e", "--no-live-traffic", "--no-other-workloads"],
                           capture_output=True, env=env, timeout=900)
        lines = [ln for ln in p.stdout.decode().splitlines() if ln.startswith("{")]
        if p.returncode != 0 or len(lines) != 1:
            print(f"rank {r}/{a.ranks}: FAILED rc {p.returncode}: {p.stderr.decode()[-400:]}")
            ok = False
            continue
        d = json.loads(lines[0])
        rf = d["roofline"]
        total += d["frames_checksummed"]
        ok = ok and d["checksums_match_golden"] is True
        values.append(d["value"])
        print(f"rank {r}/{a.ranks}: global frames {d['as_rank']['global_frames']}  checksummed {d['frames_checksummed']}  "
              f"match_golden {d['checksums_match_golden']}  value {d['value']:.0f} MP/s  kernel_ms {rf['kernel_ms']}  frac {rf['frac']}  "
              f"one frame {rf['single_frame_launch']['kernel_ms']} ms  ({time.time() - t0:.1f} s)", flush=True)
    print(f"summary: {total} frames of configs[4] decoded on one GPU as {a.ranks} virtual ranks, all golden: {ok}; "
          f"per-rank value min {min(values):.0f} max {max(values):.0f} MP/s; {time.time() - t_all:.0f} s")
    sys.exit(0 if ok and total == 128 * a.ranks else 1)


if __name__ == "__main__":
    main()
