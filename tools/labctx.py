"""ctypes binding of zune-jpeg_amd/libzjlab.so (micro-benchmark + lab kernels; tools only, never the product)."""
import ctypes as C
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


class Lab:
    def __init__(self, device=0):
        p = os.path.join(ROOT, "zune-jpeg_amd", "libzjlab.so")
        if not os.path.exists(p):
            raise ImportError(f"{p} not found: make -C zune-jpeg_amd/csrc lab")
        L = self.L = C.CDLL(p)
        vp, f = C.c_void_p, C.POINTER(C.c_float)
        L.zjlab_create.restype = vp
        L.zjlab_create.argtypes = [C.c_int]
        L.zjlab_destroy.argtypes = [vp]
        for n in ("ubench", "labmem", "lab"):
            getattr(L, f"zjlab_{n}_name").restype = C.c_char_p
            getattr(L, f"zjlab_{n}_name").argtypes = [C.c_int]
        L.zjlab_ubench.argtypes = [vp, C.c_int, C.c_int, C.c_int, C.c_int, f]
        L.zjlab_lab.argtypes = [vp, C.c_int, C.c_int, C.c_int, C.c_int, f]
        L.zjlab_labmem.argtypes = [vp, C.c_int, C.c_longlong, C.c_int, f]
        L.zjlab_clock.argtypes = [vp, C.c_int, C.POINTER(C.c_double), f]
        self.h = L.zjlab_create(device)
        if not self.h:
            raise RuntimeError("zjlab_create failed (no HIP device?)")

    def close(self):
        if self.h:
            self.L.zjlab_destroy(self.h)
            self.h = None

    def _t(self, fn, reps, *a):
        ms = C.c_float(0)
        if fn(self.h, *a, reps, C.byref(ms)):
            raise RuntimeError("lab call failed")
        return ms.value / reps

    def ubench(self, op, blocks=2048, iters=200, reps=5):
        return self._t(self.L.zjlab_ubench, reps, op, blocks, iters)

    def lab(self, variant, blocks=4096, iters=20, reps=3):
        return self._t(self.L.zjlab_lab, reps, variant, blocks, iters)

    def labmem(self, variant, nbytes, reps=20):
        return self._t(self.L.zjlab_labmem, reps, variant, nbytes)

    def clock_mhz(self, iters=200000):
        cyc, ms = C.c_double(0), C.c_float(0)
        if self.L.zjlab_clock(self.h, iters, C.byref(cyc), C.byref(ms)):
            raise RuntimeError("zjlab_clock failed")
        return cyc.value / (ms.value * 1e3)
