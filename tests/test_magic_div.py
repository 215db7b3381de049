"""tile_from_id divides a workgroup id by two launch constants with host-computed multipliers (zj_device.h: magic_u31 /
magic_div) instead of the compiler's reciprocal sequence.  The claim: exact for every 0 <= n < 2^31 and 1 <= d < 2^30."""
import ctypes as C

import numpy as np

import emu_c


def _div(d, n):
    n = np.ascontiguousarray(n, np.uint32)
    q = np.empty_like(n)
    emu_c.lib().zje_magic_div(C.c_uint32(d), n.ctypes.data_as(C.c_void_p), C.c_size_t(n.size), q.ctypes.data_as(C.c_void_p))
    return q


def test_magic_division_is_exact_at_the_edges_of_every_quotient():
    rng = np.random.default_rng(5)
    ds = list(range(1, 300)) + [511, 512, 513, 1023, 4095, 4096, 65535, 65536, 65537, (1 << 20) + 7, (1 << 29) + 1, (1 << 30) - 1]
    ds += [int(x) for x in rng.integers(1, 1 << 30, 200)]
    top = (1 << 31) - 1
    for d in ds:
        ks = np.unique(np.concatenate([np.arange(0, 40), rng.integers(0, top // d + 1, 400), [top // d - 1, top // d]]))
        ks = ks[(ks >= 0)].astype(np.int64)
        n = np.concatenate([ks * d - 1, ks * d, ks * d + 1, ks * d + d - 1, [top, top - 1, 0, 1]])
        n = n[(n >= 0) & (n <= top)].astype(np.uint32)
        assert np.array_equal(_div(d, n), n // np.uint32(d)), d


def test_magic_division_exhaustive_for_the_grids_the_kernels_launch():
    # tiles per row 1..64 and strips per frame 1..2048 over every id a 16-frame 4096x4096 launch forms (32768) and well beyond
    L = emu_c.lib()
    L.zje_magic_sweep.restype = C.c_longlong
    for d in list(range(1, 65)) + [127, 128, 129, 255, 256, 512, 1024, 2047, 2048]:
        assert L.zje_magic_sweep(C.c_uint32(d), C.c_uint32(0), C.c_uint32(1 << 21)) == -1, d
        assert L.zje_magic_sweep(C.c_uint32(d), C.c_uint32((1 << 31) - (1 << 18)), C.c_uint32((1 << 31) - 1)) == -1, d


def test_tile_from_id_enumerates_the_grid():
    L = emu_c.lib()
    for nf, ns, tpr in [(1, 1, 1), (3, 5, 7), (16, 128, 16), (2, 135, 30), (1, 2048, 1), (5, 1, 64)]:
        out = (C.c_int * 3)()
        seen = []
        for i in range(nf * ns * tpr):
            L.zje_tile_from_id(nf, ns, tpr, i, out)
            seen.append((out[0], out[1], out[2]))
        assert seen == [(f, s, t) for f in range(nf) for s in range(ns) for t in range(tpr)]
