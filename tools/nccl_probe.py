"""World-size-1 RCCL run of every collective bench.py uses, on device tensors of cuda:0, through zune-jpeg_amd/shard.py
(`always=True`: the world-size-1 short cuts are off).  Run as a FRESH process (tests/test_gpu_rccl.py, tools/gpu_r04_*.sh):
it initialises RCCL.  Prints one JSON line."""
import importlib
import json
import os
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

shard = importlib.import_module("zune-jpeg_amd.shard")
torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
port_file = os.path.join(tempfile.mkdtemp(prefix="zj_nccl_probe_"), "port")
t0 = time.perf_counter()
shard.init_process_group("nccl", 0, 1, force=True, port_file=port_file, timeout_s=120, device_id=dev)
t_init = time.perf_counter() - t0
t0 = time.perf_counter()
shard.barrier(1, always=True)
mx = shard.max_over_ranks(3.5, 1, dev, always=True)
sm = shard.sum_over_ranks(2.0, 1, dev, always=True)
vals = shard.gather_values(0.2911, 1, dev, always=True)
sums = [0, 1, (1 << 64) - 1, 0x9E3779B97F4A7C15] * 32
got = shard.gather_checksums(sums, 1, dev, always=True)
frames = torch.arange(1 << 22, dtype=torch.int32, device=dev).view(torch.uint8)  # 16 MB "decoded frames"
outs, g_s = shard.gather_frames(frames, 0, 1, always=True)
torch.cuda.synchronize()
res = {"backend": dist.get_backend(), "world": dist.get_world_size(), "init_s": round(t_init, 2),
       "collectives_s": round(time.perf_counter() - t0, 2), "max": mx, "sum": sm, "gather_values": vals,
       "checksums_ok": got == [sums], "gather_frames_ok": bool(torch.equal(outs[0], frames)), "gather_frames_s": round(g_s, 4),
       "HSA_ENABLE_IPC_MODE_LEGACY": os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY")}
print(json.dumps(res), flush=True)
dist.destroy_process_group()
