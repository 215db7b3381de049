"""ctypes binding of the C oracle (oracle/libzjoracle.so).  TEST INFRASTRUCTURE ONLY."""
import ctypes as C
import fcntl
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
_LIB = None

OK, ERR_PANIC, ERR_ARG, ERR_NOMEM = 0, -1, -2, -3
RGB, GRAYSCALE, YCBCR, CMYK, YCCK, RGBA, RGBX = range(7)


class Component(C.Structure):
    _fields_ = [("horizontal_sample", C.c_size_t), ("vertical_sample", C.c_size_t),
                ("width_stride", C.c_size_t), ("quantization_table", C.c_int32 * 64)]


class Frame(C.Structure):
    _fields_ = [("width", C.c_uint32), ("height", C.c_uint32), ("h_max", C.c_uint32),
                ("v_max", C.c_uint32), ("in_components", C.c_uint32), ("out_colorspace", C.c_int32),
                ("qt", (C.c_int32 * 64) * 3)]


def build(force=False):
    so = os.path.join(ORACLE_DIR, "libzjoracle.so")
    src = os.path.join(ORACLE_DIR, "zj_oracle.c")
    def stale():
        return not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src)
    if force or stale():
        with open(so + ".lock", "w") as lk:          # pytest-xdist workers: one builds, the others wait and find it fresh
            fcntl.flock(lk, fcntl.LOCK_EX)
            if force or stale():
                subprocess.check_call(["make", "-s", "-C", ORACLE_DIR, so])
    return so


def lib():
    global _LIB
    if _LIB is None:
        _LIB = C.CDLL(build())
        _LIB.zjo_plane_len.restype = C.c_size_t
        _LIB.zjo_out_len.restype = C.c_size_t
        _LIB.zjo_num_components.restype = C.c_size_t
    return _LIB


def _p(a, t):
    return a.ctypes.data_as(C.POINTER(t))


def _i16(a):
    return np.ascontiguousarray(a, dtype=np.int16)


def idct_strip(coeff, qt, stride, samp_factors, v_samp):
    coeff = _i16(coeff)
    qt = np.ascontiguousarray(qt, dtype=np.int32)
    out = np.empty(coeff.size, np.int16)
    rc = lib().zjo_idct_strip(_p(coeff, C.c_int16), C.c_size_t(coeff.size), _p(qt, C.c_int32),
                              C.c_size_t(stride), C.c_size_t(samp_factors), C.c_size_t(v_samp),
                              _p(out, C.c_int16))
    return rc, out


def _ups(fn, inp, out_len):
    inp = _i16(inp)
    out = np.empty(out_len, np.int16)
    rc = fn(_p(inp, C.c_int16), C.c_size_t(inp.size), _p(out, C.c_int16), C.c_size_t(out_len))
    return rc, out


def upsample_h(inp, out_len):
    return _ups(lib().zjo_upsample_h, inp, out_len)


def upsample_v(inp, out_len):
    return _ups(lib().zjo_upsample_v, inp, out_len)


def upsample_hv(inp, out_len):
    return _ups(lib().zjo_upsample_hv, inp, out_len)


def ycbcr_to_rgb16(y, cb, cr, out, pos):
    y, cb, cr = _i16(y), _i16(cb), _i16(cr)
    p = C.c_size_t(pos)
    rc = lib().zjo_ycbcr_to_rgb16(_p(y, C.c_int16), _p(cb, C.c_int16), _p(cr, C.c_int16),
                                  _p(out, C.c_uint8), C.c_size_t(out.size), C.byref(p))
    return rc, p.value


def make_components(h, v, mcu_x, qts):
    comps = (Component * 3)()
    for c in range(3):
        comps[c].horizontal_sample = h if c == 0 else 1
        comps[c].vertical_sample = v if c == 0 else 1
        comps[c].width_stride = (h if c == 0 else 1) * mcu_x * 8
        q = np.asarray(qts[min(c, len(qts) - 1)], np.int32)
        for k in range(64):
            comps[c].quantization_table[k] = int(q[k])
    return comps


def post_process(coeff, comps, in_cs, out_cs, out, width):
    arrs = [_i16(c) for c in coeff]
    ptrs = (C.POINTER(C.c_int16) * 3)(*[_p(a, C.c_int16) for a in arrs])
    lens = (C.c_size_t * 3)(*[a.size for a in arrs])
    return lib().zjo_post_process(ptrs, lens, comps, C.c_int(in_cs), C.c_int(out_cs),
                                  _p(out, C.c_uint8), C.c_size_t(out.size), C.c_size_t(width))


def make_frame(width, height, h_max, v_max, in_components, out_cs, qts):
    f = Frame()
    f.width, f.height, f.h_max, f.v_max = width, height, h_max, v_max
    f.in_components, f.out_colorspace = in_components, out_cs
    for c in range(3):
        q = np.asarray(qts[min(c, len(qts) - 1)], np.int32)
        for k in range(64):
            f.qt[c][k] = int(q[k])
    return f


def plane_len(frame, comp):
    return lib().zjo_plane_len(C.byref(frame), C.c_int(comp))


EXT_PLAIN, EXT_CLAMP_DC, EXT_EDGE_REP = 1, 2, 4


def decode_planes(frame, planes, plain=False, ext=0):
    """plain=True: the oracle's EXTENSION (every pixel at its own position; RGB / RGBA / RGBX), the checker for
    ZJ_FLAG_PLAIN_TAIL, ZJ_CS_RGBA/RGBX and ZJ_LAYOUT_CHW."""
    arrs = [_i16(p) for p in planes]
    while len(arrs) < 3:
        arrs.append(np.zeros(1, np.int16))
    out = np.zeros(lib().zjo_out_len(C.byref(frame)), np.uint8)
    ext |= EXT_PLAIN if plain else 0
    if ext:  # any combination of the extension flags (zjo_decode_planes_ext)
        rc = lib().zjo_decode_planes_ext(C.byref(frame), C.c_int(ext), _p(arrs[0], C.c_int16), _p(arrs[1], C.c_int16),
                                         _p(arrs[2], C.c_int16), _p(out, C.c_uint8))
    else:
        rc = lib().zjo_decode_planes(C.byref(frame), _p(arrs[0], C.c_int16), _p(arrs[1], C.c_int16), _p(arrs[2], C.c_int16),
                                     _p(out, C.c_uint8))
    return rc, out
