"""
Image-level sharding across the GPUs of one node: frames are independent, so rank r
of W decodes its own contiguous shard and there is NO collective on the data path.  torch.distributed
(backend "nccl" == RCCL over xGMI on the GPU box, "gloo" in the CPU tests) is used only for the
barrier around the timed region, the MAX over ranks of the elapsed time, the trivial gather of
per-frame checksums after it and -- optional, measured separately -- the gather of the decoded RGB
frames to rank 0 (`gather_frames`).

Every collective short-circuits at world size 1 unless `always=True`: a `-m gpu` test runs them with
`always` on a world-size-1 "nccl" group, which initialises RCCL and executes the device-tensor arm on
the box's one GPU (tests/test_gpu_rccl.py).
"""
import datetime
import os
import time

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


def _store_from_port_file(path, rank, world, timeout_s):
    """Rendezvous without a port race: rank 0 lets the kernel pick a free port (TCPStore on port 0), publishes it through
    `path` (written to a temporary name, then renamed), the other ranks wait for the file.  bench.py's self-launcher uses
    this; under torchrun MASTER_PORT is given and env:// is used instead."""
    import torch.distributed as dist
    to = datetime.timedelta(seconds=timeout_s)
    if rank == 0:
        store = dist.TCPStore("127.0.0.1", 0, world, is_master=True, timeout=to, wait_for_workers=False)
        tmp = f"{path}.{os.getpid()}"
        with open(tmp, "w") as f:
            f.write(str(store.port))
        os.replace(tmp, path)
        return store
    t0 = time.monotonic()
    while True:
        try:
            port = int(open(path).read())
            break
        except (OSError, ValueError):
            if time.monotonic() - t0 > timeout_s:
                raise TimeoutError(f"rank {rank}: no port published in {path} after {timeout_s} s")
            time.sleep(0.05)
    return dist.TCPStore("127.0.0.1", port, world, is_master=False, timeout=to)


def init_process_group(backend, rank, world, force=False, port_file=None, timeout_s=600, device_id=None):
    """world > 1 (or `force`): initialise torch.distributed.  `port_file`: see _store_from_port_file.  `device_id`
    (a torch.device) binds the RCCL communicator to this rank's GPU at init time, so the first collective does not have to
    guess it from the global rank."""
    import torch.distributed as dist
    if (world > 1 or force) and not dist.is_initialized():
        kw = {"timeout": datetime.timedelta(seconds=timeout_s)}
        if device_id is not None and backend == "nccl":
            kw["device_id"] = device_id
        if port_file:
            kw["store"] = _store_from_port_file(port_file, rank, world, timeout_s)
        else:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29511")
        dist.init_process_group(backend=backend, rank=rank, world_size=world, **kw)


def _default_store():
    try:
        import torch.distributed as dist
        return dist.distributed_c10d._get_default_store()
    except Exception:  # noqa: BLE001
        return None


def signal_from_rank0(tag, world):
    """Rank 0 has finished a host-only leg (bench.py's cpu_baseline at N > 1): release the ranks parked in wait_for_rank0."""
    if world > 1:
        st = _default_store()
        if st is not None:
            st.set(tag, "1")


def wait_for_rank0(tag, world, timeout_s=600):
    """Park this rank WITHOUT spinning until rank 0 signals `tag`: a blocking wait on the rendezvous store's socket.  A
    collective barrier would do the waiting inside a stream synchronisation, which polls a host core per rank -- on the
    very cores rank 0 is timing the CPU baseline on.  Falls through (the barrier that follows still synchronises) when
    there is no store."""
    if world > 1:
        st = _default_store()
        if st is not None:
            try:
                st.wait([tag], datetime.timedelta(seconds=timeout_s))
            except Exception:  # noqa: BLE001 -- timeout: let the barrier's own timeout name the problem
                pass


def barrier(world, always=False):
    if world > 1 or always:
        import torch.distributed as dist
        dist.barrier()


def _reduce(value, world, device, op_name, always):
    if world == 1 and not always:
        return float(value)
    import torch
    import torch.distributed as dist
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=getattr(dist.ReduceOp, op_name))
    return float(t.item())


def max_over_ranks(value, world, device="cpu", always=False):
    return _reduce(value, world, device, "MAX", always)


def sum_over_ranks(value, world, device="cpu", always=False):
    return _reduce(value, world, device, "SUM", always)


def gather_values(value, world, device="cpu", always=False):
    """One float64 per rank, on every rank (per-rank timings: a straggler GPU shows up here, not in the MAX)."""
    if world == 1 and not always:
        return [float(value)]
    import torch
    import torch.distributed as dist
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    outs = [torch.empty_like(t) for _ in range(world)]
    dist.all_gather(outs, t)
    return [float(o.item()) for o in outs]


def gather_checksums(local, world, device="cpu", always=False):
    """The 'trivial gather': every rank contributes a fixed-length int64 vector of checksums."""
    if world == 1 and not always:
        return [list(local)]
    import torch
    import torch.distributed as dist
    # int64 two's complement carries the 64-bit pattern
    t = torch.tensor([c - (1 << 64) if c >= (1 << 63) else c for c in local], dtype=torch.int64, device=device)
    outs = [torch.empty_like(t) for _ in range(world)]
    dist.all_gather(outs, t)
    return [[int(v) & ((1 << 64) - 1) for v in o.cpu().tolist()] for o in outs]


def gather_frames(local, rank, world, always=False):
    """SURVEY.md 8e's optional data gather: every rank's decoded frames (one uint8 tensor, device or host) to rank 0
    -- over xGMI when the tensors are on GPUs and the backend is RCCL.  Returns (list of world tensors on rank 0 | None,
    seconds): the time is this rank's, bracketed by device synchronisation when `local` is a device tensor; callers take
    the MAX over ranks.  Never part of the decode throughput."""
    import torch
    import torch.distributed as dist
    if world == 1 and not always:
        return [local], 0.0
    on_gpu = local.is_cuda
    outs = [torch.empty_like(local) for _ in range(world)] if rank == 0 else None
    if on_gpu:
        torch.cuda.synchronize(local.device)
    t0 = time.perf_counter()
    dist.gather(local, outs, dst=0)
    if on_gpu:
        torch.cuda.synchronize(local.device)
    return outs, time.perf_counter() - t0


def effective_cpus():
    """Host threads this process may actually run at once: min(affinity mask, cgroup v2 cpu.max quota).
    The GPU boxes expose 256 logical CPUs but cap the container at 16 (cpu.max = 1600000 100000)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = max(1, min(n, int(int(quota) / int(period))))
    except Exception:
        pass
    return n
