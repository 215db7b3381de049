"""
Image-level sharding across the GPUs of one node: frames are independent, so rank r
of W decodes its own contiguous shard and there is NO collective on the data path.  torch.distributed
(backend "nccl" == RCCL over xGMI on the GPU box, "gloo" in the CPU tests) is used only for the
barrier around the timed region, the MAX over ranks of the elapsed time, and the trivial gather of
per-frame checksums after it.
"""
import os

import numpy as np


def env_world():
    """(rank, local_rank, world_size) from the torchrun environment (1 process: 0, 0, 1)."""
    return (int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")),
            int(os.environ.get("WORLD_SIZE", "1")))


def shard_range(nframes, rank, world):
    """Contiguous shard [lo, hi) of frame indices for `rank`; sizes differ by at most one."""
    base, rem = divmod(nframes, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def frame_checksum(out_bytes):
    """64-bit position-weighted checksum of one decoded frame (numpy uint8 array)."""
    a = np.asarray(out_bytes, dtype=np.uint8).reshape(-1)
    n = a.size - a.size % 8
    w = a[:n].view(np.uint64)
    k = (np.arange(w.size, dtype=np.uint64) * np.uint64(0x9E3779B97F4A7C15)) | np.uint64(1)
    with np.errstate(over="ignore"):
        s = np.bitwise_xor.reduce(w * k) if w.size else np.uint64(0)
        s ^= np.uint64(int(a[n:].astype(np.uint64).sum()) + a.size)
    return int(s)


def init_process_group(backend, rank, world):
    import torch.distributed as dist
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        dist.init_process_group(backend=backend, rank=rank, world_size=world)


def barrier(world):
    if world > 1:
        import torch.distributed as dist
        dist.barrier()


def max_over_ranks(value, world, device="cpu"):
    if world == 1:
        return float(value)
    import torch
    import torch.distributed as dist
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def sum_over_ranks(value, world, device="cpu"):
    if world == 1:
        return float(value)
    import torch
    import torch.distributed as dist
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return float(t.item())


def gather_checksums(local, world, device="cpu"):
    """The 'trivial gather': every rank contributes a fixed-length int64 vector of checksums."""
    if world == 1:
        return [list(local)]
    import torch
    import torch.distributed as dist
    # int64 two's complement carries the 64-bit pattern
    t = torch.tensor([c - (1 << 64) if c >= (1 << 63) else c for c in local], dtype=torch.int64, device=device)
    outs = [torch.empty_like(t) for _ in range(world)]
    dist.all_gather(outs, t)
    return [[int(v) & ((1 << 64) - 1) for v in o.cpu().tolist()] for o in outs]


def effective_cpus():
    """Host threads this process may actually run at once: min(affinity mask, cgroup v2 cpu.max quota).
    The GPU boxes expose 256 logical CPUs but cap the container at 16 (cpu.max = 1600000 100000)."""
    import os
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = max(1, min(n, int(int(quota) / int(period))))
    except Exception:
        pass
    return n
