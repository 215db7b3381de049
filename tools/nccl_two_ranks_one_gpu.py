"""Can two RCCL ranks share ONE GPU?  (They cannot: this is why configs[4] over RCCL stays unmeasured on a one-GPU box.)
Two child processes, both on cuda:0, backend nccl, one all_reduce; prints what happens.  tools/gpu_r04_i.sh."""
import os
import subprocess
import sys
import tempfile

CHILD = r'''
import os, sys, datetime
import torch, torch.distributed as dist
rank = int(sys.argv[1]); port_file = sys.argv[2]
torch.cuda.set_device(0)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(sys.argv[3]))))
import importlib
shard = importlib.import_module("zune-jpeg_amd.shard")
try:
    shard.init_process_group("nccl", rank, 2, port_file=port_file, timeout_s=60, device_id=torch.device("cuda", 0))
    t = torch.ones(4, device="cuda:0")
    dist.all_reduce(t)
    torch.cuda.synchronize()
    print(f"rank {rank}: all_reduce ok -> {t.tolist()}", flush=True)
except Exception as e:
    print(f"rank {rank}: {type(e).__name__}: {str(e)[:300]}", flush=True)
    sys.exit(3)
'''
d = tempfile.mkdtemp(prefix="zj_two_ranks_")
pf = os.path.join(d, "port")
open(os.path.join(d, "child.py"), "w").write(CHILD)
ps = [subprocess.Popen([sys.executable, os.path.join(d, "child.py"), str(r), pf, os.path.abspath(__file__)], stdout=subprocess.PIPE, stderr=subprocess.STDOUT) for r in range(2)]
for r, p in enumerate(ps):
    try:
        out, _ = p.communicate(timeout=150)
    except subprocess.TimeoutExpired:
        p.kill()
        out, _ = p.communicate()
        out += b"\n(killed after 150 s)"
    lines = [ln for ln in out.decode(errors="replace").splitlines() if ln.startswith("rank") or "Duplicate" in ln or "invalid" in ln.lower()]
    print(f"--- rank {r} exit {p.returncode}")
    print("\n".join(lines[-6:]))
