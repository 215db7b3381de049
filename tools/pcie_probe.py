#!/usr/bin/env python3
"""PCIe copy behaviour on the GPU box (pinned host memory): one direction, both directions on two streams,
and chunked copies -- decides how zj_decode_planes_batch should cut and overlap its units."""
import time
import torch

dev = torch.device("cuda:0")
N = 50 * 1024 * 1024
h_in = torch.empty(N, dtype=torch.uint8).pin_memory()
h_out = torch.empty(N, dtype=torch.uint8).pin_memory()
d_a = torch.empty(N, dtype=torch.uint8, device=dev)
d_b = torch.empty(N, dtype=torch.uint8, device=dev)
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()


def timeit(fn, n=20):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n


def h2d():
    with torch.cuda.stream(s1):
        d_a.copy_(h_in, non_blocking=True)


def d2h():
    with torch.cuda.stream(s2):
        h_out.copy_(d_b, non_blocking=True)


def both():
    h2d(); d2h()


def chunked(k, two_streams=True):
    c = N // k
    for i in range(k):
        with torch.cuda.stream(s1):
            d_a[i * c:(i + 1) * c].copy_(h_in[i * c:(i + 1) * c], non_blocking=True)
        with torch.cuda.stream(s2 if two_streams else s1):
            h_out[i * c:(i + 1) * c].copy_(d_b[i * c:(i + 1) * c], non_blocking=True)


t = timeit(h2d); print(f"H2D 50 MiB            {t*1e3:7.3f} ms  {N/t/1e9:6.1f} GB/s")
t = timeit(d2h); print(f"D2H 50 MiB            {t*1e3:7.3f} ms  {N/t/1e9:6.1f} GB/s")
t = timeit(both); print(f"H2D + D2H, 2 streams  {t*1e3:7.3f} ms  {2*N/t/1e9:6.1f} GB/s total")
for k in (4, 13, 50):
    t = timeit(lambda: chunked(k)); print(f"  in {k:3d} chunks each, 2 streams {t*1e3:7.3f} ms  {2*N/t/1e9:6.1f} GB/s total")
    t = timeit(lambda: chunked(k, False)); print(f"  in {k:3d} chunks each, 1 stream  {t*1e3:7.3f} ms  {2*N/t/1e9:6.1f} GB/s total")
