"""The RCCL arm of bench.py's collectives on real hardware: a world-size-1 "nccl" process group on cuda:0 initialises
RCCL and runs barrier / all_reduce(MAX, SUM) / all_gather / gather on DEVICE tensors through zune-jpeg_amd/shard.py --
exactly the calls the 8-GPU run makes (BASELINE configs[4]), with the world-size-1 short cuts switched off.
A fresh child process, because RCCL initialisation is per process."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_rccl_world_size_one_runs_every_collective_of_the_bench_on_device_tensors():
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "nccl_probe.py")], capture_output=True, timeout=600, env=env)
    assert r.returncode == 0, r.stderr.decode()[-3000:]
    line = [ln for ln in r.stdout.decode().splitlines() if ln.startswith("{")][-1]
    res = json.loads(line)
    assert res["backend"] == "nccl" and res["world"] == 1
    assert res["max"] == 3.5 and res["sum"] == 2.0 and res["gather_values"] == [0.2911]
    assert res["checksums_ok"] is True and res["gather_frames_ok"] is True


@pytest.mark.gpu
def test_bench_gather_rgb_runs_on_one_gpu():
    """--gather-rgb at N = 1: the frame gather's plumbing (dist.gather on the device tensor through RCCL needs a process
    group, so at world size 1 the bench initialises one) and its spot check against the golden checksum."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "4", "--warmup", "1", "--min-untimed", "1",
                        "--shard-frames", "16", "--no-cpu-baseline", "--no-live-traffic", "--no-single-frame", "--no-e2e",
                        "--gather-rgb"], capture_output=True, timeout=900, env=env)
    assert r.returncode == 0, r.stderr.decode()[-3000:]
    res = json.loads([ln for ln in r.stdout.decode().splitlines() if ln.startswith("{")][-1])
    g = res["gather_rgb"]
    assert "error" not in g, g
    assert g["bytes_total"] == 16 * 4096 * 4096 * 3 and g["bytes_remote"] == 0
    assert g["last_frame_matches_golden"] is True and res["checksums_match_golden"] is True
