"""ctypes binding of the restated AVX2 CPU baseline (oracle/libzjavx2.so).  TEST/BENCH ONLY."""
import ctypes as C
import os
import subprocess

import numpy as np

import oracle_c as oc

_LIB = None


def lib():
    global _LIB
    if _LIB is None:
        so = os.path.join(oc.ORACLE_DIR, "libzjavx2.so")
        srcs = [os.path.join(oc.ORACLE_DIR, f) for f in ("zj_avx2.c", "zj_oracle.c")]
        if not os.path.exists(so) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in srcs):
            subprocess.check_call(["make", "-s", "-C", oc.ORACLE_DIR, so])
        _LIB = C.CDLL(so)
    return _LIB


def _p(a):
    return C.c_void_p(a.ctypes.data)


def idct_strip(coeff, qt, stride, samp_factors, v_samp):
    coeff = np.ascontiguousarray(coeff, np.int16)
    qt = np.ascontiguousarray(qt, np.int32)
    out = np.empty(coeff.size, np.int16)
    rc = lib().zja_idct_strip_avx2(_p(coeff), C.c_size_t(coeff.size), _p(qt), C.c_size_t(stride),
                                   C.c_size_t(samp_factors), C.c_size_t(v_samp), _p(out))
    return rc, out


def upsample_hv(inp, out_len):
    inp = np.ascontiguousarray(inp, np.int16)
    mid = np.empty(inp.size * 2, np.int16)
    out = np.empty(out_len, np.int16)
    rc = lib().zja_upsample_v(_p(inp), C.c_size_t(inp.size), _p(mid), C.c_size_t(mid.size))
    if rc == 0:
        rc = lib().zja_upsample_h(_p(mid), C.c_size_t(mid.size), _p(out), C.c_size_t(out_len))
    return rc, out


def upsample_hv_avx(inp, out_len, simd_entry=False):
    """upsample_hv_avx (src/upsampler/avx2.rs:29) or, with simd_entry, upsample_hv_simd (:15: scalar below 500 samples)."""
    inp = np.ascontiguousarray(inp, np.int16)
    out = np.empty(out_len, np.int16)
    fn = lib().zja_upsample_hv_simd if simd_entry else lib().zja_upsample_hv_avx
    rc = fn(_p(inp), C.c_size_t(inp.size), _p(out), C.c_size_t(out_len))
    return rc, out


def upsample_h_sse(inp, out_len):
    """upsample_horizontal_sse_u (src/upsampler/sse.rs:24)."""
    inp = np.ascontiguousarray(inp, np.int16)
    out = np.empty(out_len, np.int16)
    rc = lib().zja_upsample_h_sse(_p(inp), C.c_size_t(inp.size), _p(out), C.c_size_t(out_len))
    return rc, out


def ycbcr_to_rgb16(y, cb, cr, out, pos):
    """ycbcr_to_rgb_avx2 (src/color_convert/avx.rs:81); returns (rc, new pos)."""
    y, cb, cr = (np.ascontiguousarray(a, np.int16) for a in (y, cb, cr))
    p = C.c_size_t(pos)
    rc = lib().zja_ycbcr_to_rgb16(_p(y), _p(cb), _p(cr), _p(out), C.c_size_t(out.size), C.byref(p))
    return rc, p.value


def decode_planes_mt(frame, planes, nframes=1, nthreads=4, out=None):
    arrs = [np.ascontiguousarray(p, np.int16) for p in planes]
    if out is None:
        out = np.zeros(nframes * frame.width * frame.height * 3, np.uint8)
    rc = lib().zja_decode_planes_mt(C.byref(frame), C.c_size_t(nframes), _p(arrs[0]), _p(arrs[1]), _p(arrs[2]),
                                    _p(out), C.c_int(nthreads))
    return rc, out
