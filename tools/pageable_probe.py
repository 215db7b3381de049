#!/usr/bin/env python3
"""How should host buffers the caller did NOT pin be fed?  Times, for 50 MiB each way:
  hipMemcpy from/to pageable memory | hipHostRegister + DMA + hipHostUnregister | DMA via a pinned bounce + memcpy."""
import ctypes as C
import time

import numpy as np
import torch

hip = C.CDLL("libamdhip64.so")
N = 50 * 1024 * 1024
dev = torch.empty(N, dtype=torch.uint8, device="cuda")
d = C.c_void_p(dev.data_ptr())
page = np.ones(N, np.uint8)
pin_t = torch.empty(N, dtype=torch.uint8).pin_memory()
pin = C.c_void_p(pin_t.data_ptr())
H2D, D2H = 1, 2


def t(fn, n=10):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


pp = C.c_void_p(page.ctypes.data)
print(f"H2D pageable hipMemcpy          {t(lambda: hip.hipMemcpy(d, pp, C.c_size_t(N), H2D)):7.2f} ms")
print(f"D2H pageable hipMemcpy          {t(lambda: hip.hipMemcpy(pp, d, C.c_size_t(N), D2H)):7.2f} ms")
print(f"H2D pinned   hipMemcpy          {t(lambda: hip.hipMemcpy(d, pin, C.c_size_t(N), H2D)):7.2f} ms")
print(f"D2H pinned   hipMemcpy          {t(lambda: hip.hipMemcpy(pin, d, C.c_size_t(N), D2H)):7.2f} ms")


def reg_h2d():
    assert hip.hipHostRegister(pp, C.c_size_t(N), 0) == 0
    hip.hipMemcpy(d, pp, C.c_size_t(N), H2D)
    hip.hipHostUnregister(pp)


def reg_d2h():
    assert hip.hipHostRegister(pp, C.c_size_t(N), 0) == 0
    hip.hipMemcpy(pp, d, C.c_size_t(N), D2H)
    hip.hipHostUnregister(pp)


print(f"H2D register+DMA+unregister     {t(reg_h2d):7.2f} ms")
print(f"D2H register+DMA+unregister     {t(reg_d2h):7.2f} ms")
print(f"register + unregister alone     {t(lambda: (hip.hipHostRegister(pp, C.c_size_t(N), 0), hip.hipHostUnregister(pp))):7.2f} ms")
pin_np = np.ctypeslib.as_array(C.cast(pin, C.POINTER(C.c_uint8)), shape=(N,))
print(f"memcpy pinned -> pageable (1 thr){t(lambda: np.copyto(page, pin_np)):7.2f} ms")
print(f"memcpy pageable -> pinned (1 thr){t(lambda: np.copyto(pin_np, page)):7.2f} ms")


def fresh_d2h():
    a = np.empty(N, np.uint8)  # untouched pages, like vec![0; n] / calloc in the caller
    hip.hipMemcpy(C.c_void_p(a.ctypes.data), d, C.c_size_t(N), D2H)


def fresh_via_bounce():
    a = np.empty(N, np.uint8)
    hip.hipMemcpy(pin, d, C.c_size_t(N), D2H)
    np.copyto(a, pin_np)


def fresh_zeroed_d2h():
    a = np.zeros(N, np.uint8)
    a[::4096] = 0  # touch every page first
    hip.hipMemcpy(C.c_void_p(a.ctypes.data), d, C.c_size_t(N), D2H)


print(f"D2H into a FRESH (untouched) buffer      {t(fresh_d2h, 5):7.2f} ms")
print(f"D2H to pinned bounce + memcpy into fresh {t(fresh_via_bounce, 5):7.2f} ms")
print(f"touch pages first, then D2H              {t(fresh_zeroed_d2h, 5):7.2f} ms")
print(f"np.empty + touch every page alone        {t(lambda: np.empty(N, np.uint8).__setitem__(slice(None, None, 4096), 0), 5):7.2f} ms")
