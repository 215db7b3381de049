"""ctypes binding of the CPU emulation of the HIP workgroup phases (tests/emu).  TEST ONLY."""
import ctypes as C
import fcntl
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
_LIB = None


def lib():
    global _LIB
    if _LIB is None:
        so = os.path.join(HERE, "emu", "libzjemu.so")
        srcs = [os.path.join(HERE, "emu", "zj_emu.cpp"),
                os.path.join(ROOT, "zune-jpeg_amd", "csrc", "zj_device.h"),
                os.path.join(ROOT, "zune-jpeg_amd", "csrc", "zj_plan.h"),
                os.path.join(ROOT, "zune-jpeg_amd", "csrc", "zj_huff.h"),
                os.path.join(ROOT, "zune-jpeg_amd", "csrc", "zj_huff_device.h")]
        def stale():
            return not os.path.exists(so) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in srcs)
        if stale():
            # pytest-xdist workers import this together: one of them builds (to a name of its own, renamed when whole),
            # the others wait at the lock and find the library fresh
            with open(so + ".lock", "w") as lk:
                fcntl.flock(lk, fcntl.LOCK_EX)
                if stale():
                    tmp = f"{so}.{os.getpid()}.tmp"
                    subprocess.check_call(["g++", "-O1", "-g", "-std=c++17", "-fPIC", "-shared", "-fno-strict-aliasing",
                                           "-Wall", "-Wno-unknown-pragmas", "-o", tmp, srcs[0]])
                    os.replace(tmp, so)
        _LIB = C.CDLL(so)
    return _LIB


def set_variant(variant):
    """0 = packed generation (default), 1 = wide generation (round 1), 2 = packed with direct stores"""
    lib().zje_set_variant(C.c_int(int(variant)))


def stats():
    """(DC-only blocks, packed-IDCT blocks, wide-IDCT blocks, tiles redone wide) of the last decode_planes call"""
    a = (C.c_longlong * 4)()
    lib().zje_stats(a)
    return tuple(int(x) for x in a)


class FrameDesc(C.Structure):  # zj_frame_desc (include/zjhip.h)
    _fields_ = [("width", C.c_uint32), ("height", C.c_uint32), ("h_max", C.c_uint32),
                ("v_max", C.c_uint32), ("in_components", C.c_uint32), ("out_colorspace", C.c_int32),
                ("qt", (C.c_int32 * 64) * 3), ("flags", C.c_uint32), ("out_layout", C.c_uint32),
                ("out_pitch", C.c_uint32)]


def decode_planes(frame, planes, nframes=1, zero_fill=1, poison=0xAA, flags=0, out_layout=0, out_pitch=0):
    """frame: the oracle's zjo_frame (or anything with the same leading fields); flags / out_layout / out_pitch are the
    extension fields of zj_frame_desc.  With out_pitch the result has out_pitch bytes per row (the padding keeps `poison`)."""
    arrs = [np.ascontiguousarray(p, np.int16) for p in planes]
    while len(arrs) < 3:
        arrs.append(np.zeros(8, np.int16))
    d = FrameDesc()
    for name in ("width", "height", "h_max", "v_max", "in_components", "out_colorspace"):
        setattr(d, name, getattr(frame, name))
    C.memmove(d.qt, frame.qt, 3 * 64 * 4)
    d.flags, d.out_layout, d.out_pitch = flags, out_layout, out_pitch
    frame = d
    w, h = frame.width, frame.height
    ncomp = {0: 3, 1: 1, 2: 3, 5: 4, 6: 4}[frame.out_colorspace]
    out = np.full(nframes * (out_pitch * h * (3 if out_layout == 1 and ncomp == 3 else 1) if out_pitch else w * h * ncomp), poison, np.uint8)
    rc = lib().zje_decode_planes(C.byref(frame), C.c_size_t(nframes), C.c_void_p(arrs[0].ctypes.data),
                                 C.c_void_p(arrs[1].ctypes.data), C.c_void_p(arrs[2].ctypes.data),
                                 C.c_void_p(out.ctypes.data), C.c_int(zero_fill))
    return rc, out


def _q(q):
    return np.ascontiguousarray(q, np.int32)


def classify(coeff, q):
    """classify_block per block: 0 DC-only, 1 packed transform exact, 2 wide transform.  coeff: (n, 64) int16"""
    c = np.ascontiguousarray(coeff, np.int16).reshape(-1, 64)
    out = np.zeros(c.shape[0], np.int32)
    lib().zje_classify(C.c_void_p(c.ctypes.data), C.c_size_t(c.shape[0]), C.c_void_p(_q(q).ctypes.data), C.c_void_p(out.ctypes.data))
    return out


def idct_packed(coeff, q):
    """idct_block_packed on every block, guard NOT consulted; (n, 64) uint8"""
    c = np.ascontiguousarray(coeff, np.int16).reshape(-1, 64)
    out = np.zeros((c.shape[0], 64), np.uint8)
    lib().zje_idct_packed(C.c_void_p(c.ctypes.data), C.c_size_t(c.shape[0]), C.c_void_p(_q(q).ctypes.data), C.c_void_p(out.ctypes.data))
    return out


def idct_wide(coeff, q):
    c = np.ascontiguousarray(coeff, np.int16).reshape(-1, 64)
    out = np.zeros((c.shape[0], 64), np.int16)
    lib().zje_idct_wide(C.c_void_p(c.ctypes.data), C.c_size_t(c.shape[0]), C.c_void_p(_q(q).ctypes.data), C.c_void_p(out.ctypes.data))
    return out


def guard_limit():
    return int(lib().zje_guard_limit())


def huff_decode(blob, plane_lens):
    """The GPU entropy stage (zj_huff_device.h) thread by thread over a prepared scan (host.Decoder.scan_blob()).
    Returns (planes, status bits, dict(rounds, nsub, decodes, first_seen))."""
    blob = np.ascontiguousarray(blob, np.uint8)
    planes = [np.zeros(max(int(n), 8), np.int16) for n in plane_lens]
    while len(planes) < 3:
        planes.append(np.zeros(8, np.int16))
    status = C.c_uint32(0)
    stats = (C.c_uint32 * 4)()
    rc = lib().zje_huff_decode(C.c_void_p(blob.ctypes.data), C.c_void_p(planes[0].ctypes.data),
                               C.c_void_p(planes[1].ctypes.data), C.c_void_p(planes[2].ctypes.data),
                               C.byref(status), stats)
    if rc:
        raise RuntimeError(f"zje_huff_decode: {rc}")
    return planes[: len(plane_lens)], int(status.value), dict(rounds=stats[0], nsub=stats[1], decodes=stats[2], first_seen=stats[3])


def decode_frames(frame, frames_planes, order=None, zero_fill=1, poison=0xAA, flags=0, out_layout=0):
    """The scattered form: frames_planes[f] = the planes of frame f, each an allocation of its own; the launch table is
    filled in `order` (a permutation; default: reversed) so that table index != allocation order.  Returns (rc, [out per
    frame])."""
    n = len(frames_planes)
    order = list(range(n))[::-1] if order is None else list(order)
    d = FrameDesc()
    for name in ("width", "height", "h_max", "v_max", "in_components", "out_colorspace"):
        setattr(d, name, getattr(frame, name))
    C.memmove(d.qt, frame.qt, 3 * 64 * 4)
    d.flags, d.out_layout = flags, out_layout
    ncomp = {0: 3, 1: 1, 2: 3, 5: 4, 6: 4}[d.out_colorspace]
    arrs = [[np.ascontiguousarray(p, np.int16) for p in pl] + [np.zeros(8, np.int16)] * (3 - len(pl)) for pl in frames_planes]
    outs = [np.full(d.width * d.height * ncomp, poison, np.uint8) for _ in range(n)]
    tab = [(C.c_void_p * n)(*[arrs[f][c].ctypes.data for f in order]) for c in range(3)]
    otab = (C.c_void_p * n)(*[outs[f].ctypes.data for f in order])
    rc = lib().zje_decode_frames(C.byref(d), C.c_size_t(n), tab[0], tab[1], tab[2], otab, C.c_int(zero_fill))
    return rc, outs
