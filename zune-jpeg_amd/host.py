"""
host.py -- ctypes binding of libzjhip.so mirroring the reference's interface for the pixel path.

Names follow the reference (paths relative to the zune-jpeg tree):
  ColorSpace                         src/misc.rs:88-121
  ZuneJpegOptions                    src/options.rs:6-160 (+ a `backend` knob: scalar / avx2 / hip)
  choose_idct_func                   src/idct.rs:40
  choose_upsample_func               Decoder::set_upsampling, src/decoder.rs:468-523
  choose_ycbcr_to_rgb_convert_func   src/color_convert.rs:61
  Context.post_process               src/worker.rs:32
  Context.decode_planes              strip loop of src/mcu_prog.rs:132-246 over whole-image planes

Only the HIP arm exists here; asking for scalar/avx2 raises ZjError(ZJ_ERR_BACKEND) because those
arms live in the host application (the Rust crate), not in this library.
"""
import ctypes as C
import enum
import os
import sys

import numpy as np

BACKEND_SCALAR, BACKEND_AVX2, BACKEND_HIP = 0, 1, 2

OK, ERR_ARG, ERR_UNSUPPORTED, ERR_HIP, ERR_NOMEM, ERR_PANIC, ERR_NO_DEVICE, ERR_BACKEND = 0, -1, -2, -3, -4, -5, -6, -7


class ColorSpace(enum.IntEnum):  # src/misc.rs:88-106
    RGB = 0
    GRAYSCALE = 1
    YCbCr = 2
    CMYK = 3
    YCCK = 4
    RGBA = 5
    RGBX = 6

    def num_components(self):  # src/misc.rs:113-121
        return {0: 3, 2: 3, 1: 1}.get(int(self), 4)


def num_components(cs):
    return ColorSpace(cs).num_components()


class ZjError(RuntimeError):
    def __init__(self, status, what=""):
        self.status = status
        msg = what
        try:
            msg = f"{what}: {lib().zj_strerror(status).decode()}"
        except Exception:
            pass
        super().__init__(f"[zj status {status}] {msg}")


class Component(C.Structure):  # zj_component  <->  Components, src/components.rs:18-43
    _fields_ = [("horizontal_sample", C.c_size_t), ("vertical_sample", C.c_size_t),
                ("width_stride", C.c_size_t), ("quantization_table", C.c_int32 * 64)]


FLAG_PLAIN_TAIL = 1   # zj_frame_desc.flags: extension, every pixel at its own position (no Q5/Q6)
FLAG_CLAMP_DC = 2     # extension: DC-only shortcut value clamped to 0..255 (Q1 corrected)
FLAG_EDGE_REPLICATE = 4  # extension: horizontal chroma filter per row with replicated edges (Q4 corrected)
FLAG_CORRECTED = 7
FLAG_FULL_AC_VALUES = 8  # zj_options.flags only: the front-end yields AC values as coded (the reference cuts some to six bits)
LAYOUT_HWC, LAYOUT_CHW = 0, 1


class FrameDesc(C.Structure):  # zj_frame_desc
    _fields_ = [("width", C.c_uint32), ("height", C.c_uint32), ("h_max", C.c_uint32),
                ("v_max", C.c_uint32), ("in_components", C.c_uint32), ("out_colorspace", C.c_int32),
                ("qt", (C.c_int32 * 64) * 3), ("flags", C.c_uint32), ("out_layout", C.c_uint32),
                ("out_pitch", C.c_uint32)]

    @classmethod
    def make(cls, width, height, h_max, v_max, in_components, out_colorspace, qts, flags=0, out_layout=0, out_pitch=0):
        d = cls()
        d.width, d.height, d.h_max, d.v_max = width, height, h_max, v_max
        d.in_components, d.out_colorspace = in_components, int(out_colorspace)
        d.flags, d.out_layout, d.out_pitch = int(flags), int(out_layout), int(out_pitch)
        for c in range(3):
            q = np.asarray(qts[min(c, len(qts) - 1)], np.int32).reshape(64)
            C.memmove(d.qt[c], q.ctypes.data, 256)
        return d


ENTROPY_CPU, ENTROPY_GPU, ENTROPY_GPU_ALWAYS = 0, 1, 2
ZJ_SCAN_BATCH_MAX = 16
RETRY_CPU = 1  # zj_decode_scan: the device hands the scan back
# csrc/zj_huff.h HUFF_ST_*
HUFF_ST = {1: "bad code", 2: "run past 63", 4: "bits exhausted", 8: "EOI cut before the last row loop", 16: "phase",
           32: "no synchronisation", 64: "a DC symbol the reference may read short"}


class Options(C.Structure):  # zj_options
    _fields_ = [("out_colorspace", C.c_int32), ("strict_mode", C.c_int32), ("max_width", C.c_int32),
                ("max_height", C.c_int32), ("max_scans", C.c_int32), ("num_threads", C.c_int32),
                ("pinned_planes", C.c_int32), ("flags", C.c_uint32), ("out_layout", C.c_uint32),
                ("entropy", C.c_int32)]


class ImageInfo(C.Structure):  # zj_image_info  <->  ImageInfo, src/decoder.rs:652-668
    _fields_ = [("width", C.c_uint16), ("height", C.c_uint16), ("components", C.c_uint8),
                ("progressive", C.c_uint8), ("h_max", C.c_uint8), ("v_max", C.c_uint8),
                ("scans", C.c_uint16), ("restart_interval", C.c_uint16)]


class ZuneJpegOptions:
    """Mirror of src/options.rs:6-40 (defaults :26-40) plus the dispatch knob."""

    def __init__(self):
        self.use_unsafe = True
        self.out_colorspace = ColorSpace.RGB
        self.num_threads = 4
        self.max_width = 16384
        self.max_height = 16384
        self.max_scans = 64
        self.strict_mode = False
        self.backend = BACKEND_HIP
        self.device = 0
        self.pinned_planes = False
        self.flags = 0          # FLAG_* extensions of the pixel path (0 = the reference's bytes)
        self.out_layout = LAYOUT_HWC
        self.entropy = ENTROPY_CPU  # ENTROPY_GPU / ENTROPY_GPU_ALWAYS: baseline Huffman scans on the device

    def to_c(self):
        o = Options()
        o.out_colorspace = int(self.out_colorspace)
        o.strict_mode = int(self.strict_mode)
        o.max_width, o.max_height, o.max_scans = self.max_width, self.max_height, self.max_scans
        o.num_threads = int(self.num_threads)
        o.pinned_planes = int(bool(self.pinned_planes))
        o.flags, o.out_layout = int(self.flags), int(self.out_layout)
        o.entropy = int(self.entropy)
        return o


_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

ABI = [  # every symbol include/zjhip.h declares
    "zj_abi_version", "zj_device_count", "zj_ctx_create", "zj_ctx_destroy", "zj_default_ctx",
    "zj_strerror", "zj_last_error", "zj_idct_strip", "zj_upsample_h", "zj_upsample_v",
    "zj_upsample_hv", "zj_ycbcr_to_rgb16", "zj_post_process_strip", "zj_choose_idct_func",
    "zj_choose_upsample_func", "zj_choose_ycbcr_to_rgb_convert_func", "zj_plane_len", "zj_out_len",
    "zj_num_components", "zj_decode_planes", "zj_decode_planes_batch", "zj_decode_planes_device",
    "zj_time_decode_device", "zj_alloc_pinned", "zj_free_pinned", "zj_set_thread_device", "zj_device_alloc",
    "zj_device_free", "zj_memcpy_h2d", "zj_memcpy_d2h", "zj_sync", "zj_device_memset",
    "zj_decode_planes_to_device", "zj_decode_scan", "zj_decode_scans", "zj_decoder_finish_pixels_batch", "zj_scan_stats", "zj_scan_planes", "zj_decoder_prepare",
    "zj_decoder_finish_pixels_device", "zj_decoder_scan_blob", "zj_decoder_gpu_status", "zj_pool_decode_files_device",
    "zj_decoder_new", "zj_decoder_free", "zj_decoder_error", "zj_decoder_read_headers",
    "zj_decoder_decode_coefficients", "zj_decoder_finish_pixels", "zj_decoder_decode_buffer",
    "zj_decoder_parallel_segments", "zj_decoder_parallel_mcus", "zj_decoder_set_num_threads",
    "zj_pool_create", "zj_pool_destroy", "zj_pool_threads", "zj_pool_error", "zj_pool_stats",
    "zj_pool_decode_files", "zj_set_variant", "zj_variant_available", "zj_set_pipeline",
    "zj_decode_frames", "zj_decode_planes_device_strided", "zj_decode_frames_device", "zj_pointer_device",
    "zj_pool_create_multi", "zj_pool_devices", "zj_pool_device_stats",
    "zj_shard_range", "zj_multi_create", "zj_multi_destroy", "zj_multi_devices", "zj_multi_ctx", "zj_multi_slot_stats",
    "zj_frame_begin", "zj_frame_rows_ready", "zj_frame_end", "zj_frame_abort",
    "zj_device_pci_bus_id", "zj_device_numa_node", "zj_bind_thread_to_numa_node", "zj_bind_thread_near_device",
    "zj_thread_numa_node", "zj_pool_slot_numa", "zj_multi_slot_numa",
    "zj_multi_decode_planes_batch", "zj_multi_decode_frames", "zj_multi_decode_frames_device",
]
SCATTER_MAX = 32  # ZJ_SCATTER_MAX: frames per launch of the scattered form


def abi_symbols():
    return list(ABI)


def lib_path():
    return os.path.join(_HERE, os.environ.get("ZJ_LIB", "libzjhip.so"))  # ZJ_LIB: A/B builds (tools/ab_lib.sh)


def _share_torch_hip_runtime():
    """PyTorch-ROCm wheels bundle their own libamdhip64.so (soname libamdhip64.so.7, asked for as `libamdhip64.so`);
    libzjhip.so asks for `libamdhip64.so.7` and otherwise brings in /opt/rocm's copy.  Two HIP runtimes in one process
    do not coexist: whichever initialises second reports "No HIP GPUs are available".  A process that uses both must
    therefore `import torch` BEFORE the first call into this module (bench.py and tests/conftest.py do), or set
    ZJ_TORCH_HIP=1, which opens the wheel's copy here (without importing torch) so that libzjhip.so binds to it by
    soname.  It is not the default: the runtime bundled with torch 2.10+rocm7.0 runs the three-stream host pipeline of
    zj_decode_planes at half the rate of ROCm 7.2's (5.9 k against 11.9 k megapixels/s for one 4096x4096 frame,
    tools/e2e.py), so torch-free processes keep the system runtime."""
    if not os.environ.get("ZJ_TORCH_HIP") or "torch" in sys.modules:
        return
    try:
        import importlib.util
        spec = importlib.util.find_spec("torch")
        if spec is None or not spec.origin:
            return
        cand = os.path.join(os.path.dirname(spec.origin), "lib", "libamdhip64.so")
        if os.path.exists(cand):
            C.CDLL(cand, mode=C.RTLD_GLOBAL)
    except Exception:  # noqa: BLE001 -- best effort; the plain load below still works without torch in the process
        pass


def lib():
    """Loads libzjhip.so; raises (never falls back) when it is missing."""
    global _LIB
    if _LIB is not None:
        return _LIB
    p = lib_path()
    if not os.path.exists(p):
        raise ImportError(f"{p} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                          "or `make -C zune-jpeg_amd/csrc` (there is no CPU fallback)")
    _share_torch_hip_runtime()
    L = C.CDLL(p)
    vp, sz, i16p, i32p, u8p = C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_void_p
    L.zj_strerror.restype = C.c_char_p
    L.zj_strerror.argtypes = [C.c_int]
    L.zj_last_error.restype = C.c_char_p
    L.zj_last_error.argtypes = [vp]
    L.zj_ctx_create.restype = vp
    L.zj_ctx_create.argtypes = [C.c_int, C.c_int, C.POINTER(C.c_int)]
    L.zj_ctx_destroy.argtypes = [vp]
    L.zj_default_ctx.restype = vp
    L.zj_plane_len.restype = sz
    L.zj_plane_len.argtypes = [C.POINTER(FrameDesc), C.c_int]
    L.zj_out_len.restype = sz
    L.zj_out_len.argtypes = [C.POINTER(FrameDesc)]
    L.zj_idct_strip.argtypes = [vp, i16p, sz, i32p, sz, sz, sz, i16p]
    for f in (L.zj_upsample_h, L.zj_upsample_v, L.zj_upsample_hv):
        f.argtypes = [vp, i16p, sz, i16p, sz]
    L.zj_ycbcr_to_rgb16.argtypes = [vp, i16p, i16p, i16p, u8p, sz, C.POINTER(sz)]
    L.zj_post_process_strip.argtypes = [vp, C.POINTER(C.c_void_p), C.POINTER(sz), C.POINTER(Component),
                                        C.c_int, C.c_int, u8p, sz, sz]
    L.zj_decode_planes.argtypes = [vp, C.POINTER(FrameDesc), i16p, i16p, i16p, u8p]
    L.zj_decode_planes_batch.argtypes = [vp, C.POINTER(FrameDesc), sz, i16p, i16p, i16p, u8p]
    L.zj_decode_planes_device.argtypes = [vp, C.POINTER(FrameDesc), sz, vp, vp, vp, vp, vp]
    L.zj_time_decode_device.argtypes = [vp, C.POINTER(FrameDesc), sz, vp, vp, vp, vp, vp, C.c_int,
                                        C.POINTER(C.c_float), C.POINTER(C.c_float), C.POINTER(C.c_char_p)]
    L.zj_alloc_pinned.restype = vp
    L.zj_alloc_pinned.argtypes = [sz]
    L.zj_free_pinned.argtypes = [vp]
    L.zj_set_thread_device.argtypes = [C.c_int]
    L.zj_device_alloc.restype = vp
    L.zj_device_alloc.argtypes = [vp, sz]
    L.zj_device_free.argtypes = [vp, vp]
    L.zj_memcpy_h2d.argtypes = [vp, vp, vp, sz]
    L.zj_memcpy_d2h.argtypes = [vp, vp, vp, sz]
    L.zj_sync.argtypes = [vp]
    L.zj_choose_idct_func.restype = vp
    L.zj_choose_idct_func.argtypes = [C.c_int]
    L.zj_choose_upsample_func.restype = vp
    L.zj_choose_upsample_func.argtypes = [C.c_int, C.c_int, C.c_int]
    L.zj_choose_ycbcr_to_rgb_convert_func.restype = vp
    L.zj_choose_ycbcr_to_rgb_convert_func.argtypes = [C.c_int, C.c_int]
    L.zj_decoder_new.restype = vp
    L.zj_decoder_new.argtypes = [C.POINTER(Options)]
    L.zj_decoder_free.argtypes = [vp]
    L.zj_decoder_error.restype = C.c_char_p
    L.zj_decoder_error.argtypes = [vp]
    L.zj_decoder_read_headers.argtypes = [vp, vp, sz, C.POINTER(ImageInfo)]
    L.zj_decoder_decode_coefficients.argtypes = [vp, vp, sz, C.POINTER(FrameDesc), C.POINTER(C.c_void_p), C.POINTER(sz), C.POINTER(ImageInfo)]
    L.zj_decoder_decode_buffer.argtypes = [vp, vp, vp, sz, vp, sz, C.POINTER(sz), C.POINTER(ImageInfo)]
    L.zj_decoder_parallel_segments.argtypes = [vp]
    L.zj_decoder_parallel_mcus.restype = C.c_longlong
    L.zj_decoder_parallel_mcus.argtypes = [vp]
    L.zj_decoder_set_num_threads.argtypes = [vp, C.c_int]
    L.zj_decoder_prepare.argtypes = [vp, vp, sz, C.POINTER(FrameDesc), C.POINTER(ImageInfo)]
    L.zj_decoder_finish_pixels.argtypes = [vp, vp, vp, sz, C.POINTER(sz)]
    L.zj_decoder_finish_pixels_device.argtypes = [vp, vp, vp, sz, C.POINTER(sz)]
    L.zj_decoder_scan_blob.argtypes = [vp, C.POINTER(C.c_void_p), C.POINTER(sz)]
    L.zj_decoder_gpu_status.restype = C.c_uint
    L.zj_decoder_gpu_status.argtypes = [vp]
    L.zj_decode_scan.argtypes = [vp, C.POINTER(FrameDesc), vp, sz, vp, C.c_int, C.POINTER(C.c_uint)]
    L.zj_decode_planes_to_device.argtypes = [vp, C.POINTER(FrameDesc), i16p, i16p, i16p, vp]
    L.zj_scan_stats.argtypes = [vp, C.POINTER(C.c_int), C.POINTER(C.c_float)]
    L.zj_scan_planes.argtypes = [vp, vp, vp, vp, C.POINTER(sz)]
    L.zj_decode_scans.argtypes = [vp, sz, vp, vp, vp, vp, C.c_int, vp, vp]
    L.zj_decoder_finish_pixels_batch.argtypes = [vp, sz, vp, vp, vp, vp, C.c_int, vp]
    L.zj_device_memset.argtypes = [vp, vp, C.c_int, sz]
    L.zj_pool_create.restype = vp
    L.zj_pool_create.argtypes = [C.c_int, C.c_int, C.POINTER(Options), C.POINTER(C.c_int)]
    L.zj_pool_destroy.argtypes = [vp]
    L.zj_pool_threads.argtypes = [vp]
    L.zj_pool_error.restype = C.c_char_p
    L.zj_pool_error.argtypes = [vp]
    L.zj_pool_stats.argtypes = [vp, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(sz)]
    L.zj_pool_decode_files.argtypes = [vp, sz, vp, vp, vp, vp, vp, vp, vp]
    L.zj_pool_decode_files_device.argtypes = [vp, sz, vp, vp, vp, vp, vp, vp, vp]
    L.zj_set_pipeline.argtypes = [vp, C.c_int]
    L.zj_set_variant.argtypes = [vp, C.c_int]
    L.zj_variant_available.argtypes = [C.c_int]
    L.zj_decode_frames.argtypes = [vp, C.POINTER(FrameDesc), sz, vp, vp, vp, vp]
    L.zj_decode_planes_device_strided.argtypes = [vp, C.POINTER(FrameDesc), sz, vp, vp, vp, vp, sz, sz, sz, vp]
    L.zj_decode_frames_device.argtypes = [vp, C.POINTER(FrameDesc), sz, vp, vp, vp, vp, vp]
    L.zj_pointer_device.argtypes = [vp]
    L.zj_pool_create_multi.restype = vp
    L.zj_pool_create_multi.argtypes = [C.POINTER(C.c_int), C.c_int, C.c_int, C.POINTER(Options), C.POINTER(C.c_int)]
    L.zj_pool_devices.argtypes = [vp]
    L.zj_pool_device_stats.argtypes = [vp, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_double), C.POINTER(sz)]
    L.zj_shard_range.restype = None
    L.zj_shard_range.argtypes = [sz, C.c_int, C.c_int, C.POINTER(sz), C.POINTER(sz)]
    L.zj_multi_create.restype = vp
    L.zj_multi_create.argtypes = [C.POINTER(C.c_int), C.c_int, C.POINTER(C.c_int)]
    L.zj_multi_destroy.argtypes = [vp]
    L.zj_multi_devices.argtypes = [vp]
    L.zj_multi_ctx.restype = vp
    L.zj_multi_ctx.argtypes = [vp, C.c_int]
    L.zj_multi_slot_stats.argtypes = [vp, C.c_int, C.POINTER(C.c_int), C.POINTER(sz)]
    L.zj_multi_decode_planes_batch.argtypes = [vp, C.POINTER(FrameDesc), sz, i16p, i16p, i16p, u8p, C.POINTER(C.c_int)]
    L.zj_multi_decode_frames.argtypes = [vp, C.POINTER(FrameDesc), sz, vp, vp, vp, vp, C.POINTER(C.c_int)]
    L.zj_multi_decode_frames_device.argtypes = [vp, C.POINTER(FrameDesc), sz, vp, vp, vp, vp, C.POINTER(C.c_int)]
    L.zj_frame_begin.argtypes = [vp, C.POINTER(FrameDesc), i16p, i16p, i16p, vp, C.c_int]
    L.zj_frame_rows_ready.argtypes = [vp, sz]
    L.zj_frame_end.argtypes = [vp]
    L.zj_frame_abort.argtypes = [vp]
    ip = C.POINTER(C.c_int)
    L.zj_device_pci_bus_id.argtypes = [C.c_int, C.c_char_p, sz]
    L.zj_device_numa_node.argtypes = [C.c_int]
    L.zj_bind_thread_to_numa_node.argtypes = [C.c_int]
    L.zj_bind_thread_near_device.argtypes = [C.c_int]
    L.zj_thread_numa_node.argtypes = []
    L.zj_pool_slot_numa.argtypes = [vp, C.c_int, ip, ip, ip]
    L.zj_multi_slot_numa.argtypes = [vp, C.c_int, ip, ip, ip]
    if hasattr(L, "zj_set_ablation"):  # diagnostic build only (tools/build_variant.sh ablate "-DZJ_ABLATION")
        L.zj_set_ablation.argtypes = [vp, C.c_int]
    _LIB = L
    return L


def device_count():
    return lib().zj_device_count()


def _check(rc, what, ctx=None):
    if rc != OK:
        detail = what
        if rc == ERR_HIP and ctx is not None:
            detail = f"{what} ({lib().zj_last_error(ctx).decode()})"
        raise ZjError(rc, detail)


def _ptr(a):
    return C.c_void_p(a.ctypes.data)


def _i16(a):
    return np.ascontiguousarray(a, dtype=np.int16)


IDCT_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_size_t,
                      C.c_size_t, C.c_void_p)
UPSAMPLE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t)
CC16_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t,
                      C.POINTER(C.c_size_t))


def _fn_or_raise(addr, proto, what):
    if not addr:
        raise ZjError(ERR_BACKEND, what)
    return proto(addr)


def choose_idct_func(backend=BACKEND_HIP):
    """src/idct.rs:40 -- returns the C function pointer of the chosen arm (HIP only)."""
    return _fn_or_raise(lib().zj_choose_idct_func(backend), IDCT_FN, "choose_idct_func")


def choose_upsample_func(backend, h_max, v_max):
    """Decoder::set_upsampling, src/decoder.rs:468-523."""
    return _fn_or_raise(lib().zj_choose_upsample_func(backend, h_max, v_max), UPSAMPLE_FN, "choose_upsample_func")


def choose_ycbcr_to_rgb_convert_func(backend, out_cs=ColorSpace.RGB):
    """src/color_convert.rs:61"""
    return _fn_or_raise(lib().zj_choose_ycbcr_to_rgb_convert_func(backend, int(out_cs)), CC16_FN,
                        "choose_ycbcr_to_rgb_convert_func")


class Context:
    """zj_ctx: one HIP stream + scratch buffers on one device."""

    def __init__(self, backend=BACKEND_HIP, device=0):
        st = C.c_int(0)
        self._h = lib().zj_ctx_create(backend, device, C.byref(st))
        if not self._h:
            raise ZjError(st.value, "zj_ctx_create")
        self.device = device

    def close(self):
        if getattr(self, "_h", None):
            lib().zj_ctx_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def handle(self):
        return self._h

    # ---- strip level (fn-pointer compatible) -------------------------------------------------
    def idct_strip(self, coeff, qt, stride, samp_factors, v_samp):
        """IDCTPtr: dequantize_and_idct_int(vector, qt_table, stride, samp_factors, v_samp)."""
        coeff = _i16(coeff)
        qt = np.ascontiguousarray(qt, dtype=np.int32)
        out = np.empty(coeff.size, np.int16)
        _check(lib().zj_idct_strip(self._h, _ptr(coeff), coeff.size, _ptr(qt), stride, samp_factors, v_samp,
                                   _ptr(out)), "zj_idct_strip", self._h)
        return out

    def _ups(self, fn, inp, out_len, what):
        inp = _i16(inp)
        out = np.empty(out_len, np.int16)
        _check(fn(self._h, _ptr(inp), inp.size, _ptr(out), out_len), what, self._h)
        return out

    def upsample_horizontal(self, inp, output_len):
        return self._ups(lib().zj_upsample_h, inp, output_len, "zj_upsample_h")

    def upsample_vertical(self, inp, output_len):
        return self._ups(lib().zj_upsample_v, inp, output_len, "zj_upsample_v")

    def upsample_hv(self, inp, output_len):
        return self._ups(lib().zj_upsample_hv, inp, output_len, "zj_upsample_hv")

    def ycbcr_to_rgb_16(self, y, cb, cr, output, pos):
        """ColorConvert16Ptr; returns the new pos."""
        y, cb, cr = _i16(y), _i16(cb), _i16(cr)
        p = C.c_size_t(pos)
        _check(lib().zj_ycbcr_to_rgb16(self._h, _ptr(y), _ptr(cb), _ptr(cr), _ptr(output), output.size,
                                       C.byref(p)), "zj_ycbcr_to_rgb16", self._h)
        return p.value

    def post_process(self, coeff, comps, in_cs, out_cs, output, width):
        """post_process(coeff, component_data, .., input_colorspace, output_colorspace, output, width)"""
        arrs = [_i16(c) for c in coeff]
        while len(arrs) < 3:
            arrs.append(np.zeros(0, np.int16))
        ptrs = (C.c_void_p * 3)(*[a.ctypes.data if a.size else None for a in arrs])
        lens = (C.c_size_t * 3)(*[a.size for a in arrs])
        _check(lib().zj_post_process_strip(self._h, ptrs, lens, comps, int(in_cs), int(out_cs), _ptr(output),
                                           output.size, width), "zj_post_process_strip", self._h)

    # ---- frame level ---------------------------------------------------------------------------
    def decode_planes(self, desc, planes, nframes=1):
        arrs = [_i16(p) for p in planes]
        while len(arrs) < 3:
            arrs.append(np.zeros(8, np.int16))
        out = np.empty(nframes * lib().zj_out_len(C.byref(desc)), np.uint8)
        _check(lib().zj_decode_planes_batch(self._h, C.byref(desc), nframes, _ptr(arrs[0]), _ptr(arrs[1]),
                                            _ptr(arrs[2]), _ptr(out)), "zj_decode_planes_batch", self._h)
        return out

    def decode_planes_device(self, desc, nframes, d_y, d_cb, d_cr, d_out, stream=None):
        """Device pointers (ints), asynchronous on `stream` (int handle or None = ctx stream)."""
        _check(lib().zj_decode_planes_device(self._h, C.byref(desc), nframes, d_y, d_cb, d_cr, d_out, stream),
               "zj_decode_planes_device", self._h)

    def decode_planes_device_strided(self, desc, nframes, d_y, d_cb, d_cr, d_out, y_stride=0, c_stride=0, out_stride=0, stream=None):
        """Frames at a uniform distance (int16 elements for the planes, bytes for the pixels; 0 = packed): ONE launch."""
        _check(lib().zj_decode_planes_device_strided(self._h, C.byref(desc), nframes, d_y, d_cb, d_cr, d_out, y_stride, c_stride,
                                                     out_stride, stream), "zj_decode_planes_device_strided", self._h)

    def decode_frames_device(self, desc, d_y, d_cb, d_cr, d_out, stream=None):
        """Scattered batch: lists of device pointers (ints), one entry per frame, any order, independent allocations.
        d_cb / d_cr may be None for GRAYSCALE output.  Asynchronous on `stream`."""
        n = len(d_y)
        arr = lambda v: (C.c_void_p * n)(*v) if v is not None else None
        _check(lib().zj_decode_frames_device(self._h, C.byref(desc), n, arr(d_y), arr(d_cb), arr(d_cr), arr(d_out), stream),
               "zj_decode_frames_device", self._h)

    def decode_frames(self, desc, frames_planes, outs=None):
        """Host frames that are independent allocations (zj_decode_frames): frames_planes[f] = [y, cb, cr] arrays of frame
        f.  Returns the list of per-frame uint8 arrays."""
        n = len(frames_planes)
        arrs = [[_i16(p) for p in pl] for pl in frames_planes]
        nc = max(len(a) for a in arrs)
        if outs is None:
            outs = [np.empty(lib().zj_out_len(C.byref(desc)), np.uint8) for _ in range(n)]
        tab = [(C.c_void_p * n)(*[a[c].ctypes.data for a in arrs]) if c < nc else None for c in range(3)]
        otab = (C.c_void_p * n)(*[o.ctypes.data for o in outs])
        _check(lib().zj_decode_frames(self._h, C.byref(desc), n, tab[0], tab[1], tab[2], otab), "zj_decode_frames", self._h)
        return outs

    def decode_to_tensor(self, desc, y, cb=None, cr=None, out=None):
        """Device-resident in, device-resident out, for PyTorch-ROCm consumers: y / cb / cr are int16
        CUDA tensors holding N frames of coefficient planes back to back; returns a uint8 CUDA tensor shaped
        [N, C, H, W] when desc.out_layout is LAYOUT_CHW, else [N, H, W, C].  Runs on torch's current stream."""
        import torch
        ylen = lib().zj_plane_len(C.byref(desc), 0)
        n = y.numel() // ylen
        assert n >= 1 and y.numel() == n * ylen and y.dtype == torch.int16 and y.is_cuda
        nc = ColorSpace(desc.out_colorspace).num_components()
        shape = (n, nc, desc.height, desc.width) if desc.out_layout == LAYOUT_CHW else (n, desc.height, desc.width, nc)
        if out is None:
            out = torch.empty(shape, dtype=torch.uint8, device=y.device)
        assert out.is_contiguous() and out.numel() == n * lib().zj_out_len(C.byref(desc))
        ptr = lambda t: t.data_ptr() if t is not None else None
        self.decode_planes_device(desc, n, ptr(y), ptr(cb), ptr(cr), out.data_ptr(), torch.cuda.current_stream(y.device).cuda_stream)
        return out

    # ---- one frame whose planes are still being written (zj_frame_*): pointers are raw addresses, the caller keeps them alive
    def frame_begin(self, desc, y, cb, cr, out, out_on_device=False):
        _check(lib().zj_frame_begin(self._h, C.byref(desc), y, cb, cr, out, 1 if out_on_device else 0), "zj_frame_begin", self._h)

    def frame_rows_ready(self, mcu_rows):
        _check(lib().zj_frame_rows_ready(self._h, int(mcu_rows)), "zj_frame_rows_ready", self._h)

    def frame_end(self):
        _check(lib().zj_frame_end(self._h), "zj_frame_end", self._h)

    def frame_abort(self):
        _check(lib().zj_frame_abort(self._h), "zj_frame_abort", self._h)

    def time_decode_device(self, desc, nframes, d_y, d_cb, d_cr, d_out, iters, stream=None):
        """HIP-event timing on the launch stream.  Returns (ms per launch from `iters` back-to-back
        launches, mean ms of individually bracketed launches, kernel name)."""
        ms, each = C.c_float(0), C.c_float(0)
        name = C.c_char_p()
        _check(lib().zj_time_decode_device(self._h, C.byref(desc), nframes, d_y, d_cb, d_cr, d_out, stream,
                                           iters, C.byref(ms), C.byref(each), C.byref(name)),
               "zj_time_decode_device", self._h)
        return ms.value / iters, each.value, (name.value or b"").decode()

    def set_variant(self, variant):
        """Kernel variant: 0 = packed generation (dot2 IDCT under an exact guard, byte luma staging, staged stores;
        the default), 1 = wide generation (round 1's kernel), 2 = packed with direct stores.  All are bit-exact."""
        _check(lib().zj_set_variant(self._h, int(variant)), "zj_set_variant", self._h)

    def set_pipeline(self, on):
        """zj_decode_planes_batch: 1 (default) = units of ~8 MB overlapped over three streams, 0 = one unit."""
        _check(lib().zj_set_pipeline(self._h, int(bool(on))), "zj_set_pipeline", self._h)

    def set_ablation(self, mask):
        """Diagnostic build only (ZJ_LIB=libzjhip_ablate.so): bit 0 skips the IDCT, bit 1 the colour math (output is
        wrong when set).  The product library has no such switch."""
        if not hasattr(lib(), "zj_set_ablation"):
            raise ZjError(ERR_ARG, "zj_set_ablation: not in this build (tools/build_variant.sh ablate \"-DZJ_ABLATION\")")
        _check(lib().zj_set_ablation(self._h, int(mask)), "zj_set_ablation", self._h)

    def device_alloc(self, nbytes):
        p = lib().zj_device_alloc(self._h, nbytes)
        if not p:
            raise ZjError(ERR_NOMEM, "zj_device_alloc")
        return p

    def device_free(self, p):
        lib().zj_device_free(self._h, p)

    def h2d(self, dst, arr):
        arr = np.ascontiguousarray(arr)
        _check(lib().zj_memcpy_h2d(self._h, dst, _ptr(arr), arr.nbytes), "zj_memcpy_h2d", self._h)

    def d2h(self, arr, src):
        _check(lib().zj_memcpy_d2h(self._h, _ptr(arr), src, arr.nbytes), "zj_memcpy_d2h", self._h)

    def sync(self):
        _check(lib().zj_sync(self._h), "zj_sync", self._h)

    def scan_stats(self):
        """(synchronisation rounds, [ms upload + rounds, ms prefix sums + write pass, ms pixel kernel + download, host ms
        spent submitting]) of the last scan the GPU entropy stage decoded on this context; the device times need
        ZJ_HUFF_TIME in the environment."""
        r = C.c_int(0)
        ms = (C.c_float * 4)()
        _check(lib().zj_scan_stats(self._h, C.byref(r), ms), "zj_scan_stats", self._h)
        return r.value, [float(x) for x in ms]

    def scan_planes(self):
        """The coefficient planes the last decode_scan / device-entropy decode on this context left in HBM (copies)."""
        lens = (C.c_size_t * 3)()
        _check(lib().zj_scan_planes(self._h, None, None, None, lens), "zj_scan_planes", self._h)
        planes = [np.zeros(max(int(n), 1), np.int16) for n in lens]
        _check(lib().zj_scan_planes(self._h, _ptr(planes[0]), _ptr(planes[1]) if lens[1] else None,
                                    _ptr(planes[2]) if lens[2] else None, lens), "zj_scan_planes", self._h)
        return [p[: int(n)] for p, n in zip(planes, lens) if n]

    def decode_scan(self, desc, blob, out=None, device_out=None):
        """zj_decode_scan: a prepared scan -> pixels (host array, or a device pointer with device_out).  Returns
        (pixels or None, return code, status bits)."""
        blob = np.ascontiguousarray(blob, np.uint8)
        st = C.c_uint(0)
        if device_out is not None:
            rc = lib().zj_decode_scan(self._h, C.byref(desc), _ptr(blob), blob.size, device_out, 1, C.byref(st))
            return None, rc, st.value
        if out is None:
            out = np.zeros(lib().zj_out_len(C.byref(desc)), np.uint8)
        rc = lib().zj_decode_scan(self._h, C.byref(desc), _ptr(blob), blob.size, _ptr(out), 0, C.byref(st))
        if rc < 0:
            _check(rc, "zj_decode_scan", self._h)
        return out, rc, st.value


class DecodeError(ZjError):
    """Mirrors DecodeErrors (src/errors.rs:16-43): .status is the variant, str() carries the text."""

    def __init__(self, status, text):
        RuntimeError.__init__(self, f"[zj status {status}] {text}")
        self.status = status
        self.text = text


class Decoder:
    """Mirror of zune_jpeg::Decoder (src/decoder.rs:60): CPU entropy decode + GPU pixel path."""

    def __init__(self, options=None, ctx=None):
        o = options.to_c() if options is not None else Options()
        self._d = lib().zj_decoder_new(C.byref(o))
        self._out_cs = int(o.out_colorspace)
        self._ctx = ctx
        self._info = None

    def close(self):
        if getattr(self, "_d", None):
            lib().zj_decoder_free(self._d)
            self._d = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _raise(self, rc):
        raise DecodeError(rc, lib().zj_decoder_error(self._d).decode(errors="replace"))

    def parallel_segments(self):
        """Restart segments the last baseline scan decoded concurrently (0 = serial walk)."""
        return lib().zj_decoder_parallel_segments(self._d)

    def parallel_mcus(self):
        """MCUs of the last baseline scan without restart markers that several threads decoded (0 = serial walk)."""
        return int(lib().zj_decoder_parallel_mcus(self._d))

    def set_num_threads(self, threads):  # decoder.rs:591-603 (deprecated there: options are the way)
        rc = lib().zj_decoder_set_num_threads(self._d, int(threads))
        if rc:
            self._raise(rc)

    def read_headers(self, buf):  # decoder.rs:452
        b = np.frombuffer(bytes(buf), np.uint8)
        info = ImageInfo()
        rc = lib().zj_decoder_read_headers(self._d, _ptr(b), b.size, C.byref(info))
        if rc:
            self._raise(rc)
        self._info = info
        return info

    def info(self):  # decoder.rs:210
        return self._info

    def decode_coefficients(self, buf, copy=True):
        """CPU half only: (FrameDesc, [planes], ImageInfo).  The planes are copies (copy=False: views of the decoder's own
        planes, valid until its next call or its end).  NOT for timing the walker: the copy of the planes -- 199 MB for a
        7680 x 4320 4:4:4 file -- costs more than the Huffman stage; prepare() leaves the planes where they are
        (tools/walker_bench.py)."""
        b = np.frombuffer(bytes(buf), np.uint8)
        desc, info = FrameDesc(), ImageInfo()
        ptrs = (C.c_void_p * 3)()
        lens = (C.c_size_t * 3)()
        rc = lib().zj_decoder_decode_coefficients(self._d, _ptr(b), b.size, C.byref(desc), ptrs, lens, C.byref(info))
        if rc:
            self._raise(rc)
        planes = [np.ctypeslib.as_array(C.cast(ptrs[c], C.POINTER(C.c_int16)), shape=(lens[c],)) for c in range(info.components)]
        if copy:
            planes = [p.copy() for p in planes]
        self._info = info
        return desc, planes, info

    def prepare(self, buf):
        """Stage 1 (CPU) as the options ask: (FrameDesc, ImageInfo).  With a GPU entropy setting a baseline scan is
        only prepared (scan_blob() then returns it); `buf` is kept alive by this object until the next call."""
        self._src = np.frombuffer(bytes(buf), np.uint8)
        desc, info = FrameDesc(), ImageInfo()
        rc = lib().zj_decoder_prepare(self._d, _ptr(self._src), self._src.size, C.byref(desc), C.byref(info))
        if rc:
            self._raise(rc)
        self._info = info
        return desc, info

    def scan_blob(self):
        """The prepared scan (csrc/zj_huff.h) of the last prepare() as a uint8 array, or None."""
        p, n = C.c_void_p(), C.c_size_t(0)
        if lib().zj_decoder_scan_blob(self._d, C.byref(p), C.byref(n)):
            return None
        return np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_uint8)), shape=(n.value,)).copy()

    def gpu_status(self):
        """HUFF_ST bits with which the device handed the last scan back to the CPU walker (0: it kept it)."""
        return int(lib().zj_decoder_gpu_status(self._d))

    def finish_pixels(self, out=None):
        """Stage 2 after prepare() / decode_coefficients(): the pixels as a uint8 array."""
        if self._ctx is None:
            self._ctx = Context()
        info = self._info
        ncomp = 1 if info.components == 1 else ColorSpace(self._out_cs).num_components()
        if out is None:
            out = np.zeros(int(info.width) * int(info.height) * ncomp, np.uint8)
        n = C.c_size_t(0)
        rc = lib().zj_decoder_finish_pixels(self._d, self._ctx.handle, _ptr(out), out.size, C.byref(n))
        if rc:
            self._raise(rc)
        return out[: n.value]

    def finish_pixels_device(self, d_out, cap):
        """Stage 2 with the pixels left in HBM at device pointer d_out; returns their length in bytes."""
        if self._ctx is None:
            self._ctx = Context()
        n = C.c_size_t(0)
        rc = lib().zj_decoder_finish_pixels_device(self._d, self._ctx.handle, d_out, cap, C.byref(n))
        if rc:
            self._raise(rc)
        return n.value

    def decode_buffer(self, buf, out=None):  # decoder.rs:178
        """out: a uint8 array to decode into (e.g. a view of pinned memory, so that the downloads of a streamed baseline
        decode overlap as well); allocated when None."""
        if self._ctx is None:
            self._ctx = Context()
        b = np.frombuffer(bytes(buf), np.uint8)
        info = self.read_headers(buf)
        ncomp = 1 if info.components == 1 else ColorSpace(self._out_cs).num_components()
        if out is None:
            out = np.zeros(int(info.width) * int(info.height) * ncomp, np.uint8)
        n = C.c_size_t(0)
        rc = lib().zj_decoder_decode_buffer(self._d, self._ctx.handle, _ptr(b), b.size, _ptr(out), out.size,
                                            C.byref(n), C.byref(info))
        if rc:
            self._raise(rc)
        self._info = info
        return out[: n.value]


def finish_pixels_batch(decoders, ctx, outs=None, device_ptrs=None):
    """zj_decoder_finish_pixels_batch over Decoder objects that went through prepare(): the scans left for the device are
    decoded together.  outs: list of uint8 arrays (allocated when None and no device_ptrs); device_ptrs: list of
    (pointer, capacity) to leave the pixels in HBM.  Returns (outs or lengths, list of return codes)."""
    n = len(decoders)
    dptr = (C.c_void_p * n)(*[d._d for d in decoders])
    if device_ptrs is None and outs is None:
        outs = []
        for d in decoders:
            i = d._info
            nc = 1 if i.components == 1 else ColorSpace(d._out_cs).num_components()
            outs.append(np.zeros(int(i.width) * int(i.height) * nc, np.uint8))
    if device_ptrs is not None:
        optr = (C.c_void_p * n)(*[p for p, _ in device_ptrs])
        caps = (C.c_size_t * n)(*[cap for _, cap in device_ptrs])
    else:
        optr = (C.c_void_p * n)(*[o.ctypes.data for o in outs])
        caps = (C.c_size_t * n)(*[o.size for o in outs])
    lens = (C.c_size_t * n)()
    rcs = (C.c_int * n)()
    _check(lib().zj_decoder_finish_pixels_batch(dptr, n, ctx.handle, optr, caps, lens, 1 if device_ptrs is not None else 0, rcs),
           "zj_decoder_finish_pixels_batch", ctx.handle)
    if device_ptrs is not None:
        return list(lens), list(rcs)
    return [o[: lens[k]] for k, o in enumerate(outs)], list(rcs)


class FileBatchDecoder:
    """JPEG files -> pixels left in HBM, e.g. a uint8 CUDA tensor [N, H, W, C] for PyTorch-ROCm consumers.
    Keeps `capacity` decoders (their prepared scans live in pinned memory) and one context; every call prepares the files
    on the CPU (headers + byte-level preparation) and finishes them in batches through zj_decoder_finish_pixels_batch:
    the entropy stage of a batch is one launch per phase, and images of one size that land equally spaced -- the rows of
    a tensor do -- share one pixel-kernel launch.  Progressive files and whatever the device hands back take the CPU
    walker; the bytes are the same."""

    def __init__(self, ctx=None, options=None, capacity=ZJ_SCAN_BATCH_MAX):
        if options is None:
            options = ZuneJpegOptions()
            options.entropy = ENTROPY_GPU
            options.pinned_planes = True
        self._ctx = ctx if ctx is not None else Context()
        self._out_cs = int(options.out_colorspace)
        self._layout = int(options.out_layout)
        self._decs = [Decoder(options, self._ctx) for _ in range(max(1, min(int(capacity), ZJ_SCAN_BATCH_MAX)))]

    def close(self):
        for d in self._decs:
            d.close()
        self._decs = []

    def decode(self, files, device_ptrs):
        """files[k] -> device_ptrs[k] = (pointer, capacity).  Returns (lengths, infos); raises DecodeError on the first failure."""
        lens, infos = [], []
        for base in range(0, len(files), len(self._decs)):
            chunk = files[base:base + len(self._decs)]
            for d, f in zip(self._decs, chunk):
                infos.append(d.prepare(f)[1])
            ln, rcs = finish_pixels_batch(self._decs[:len(chunk)], self._ctx, device_ptrs=device_ptrs[base:base + len(chunk)])
            for d, rc in zip(self._decs, rcs):
                if rc:
                    d._raise(rc)
            lens += ln
        return lens, infos

    def to_tensor(self, files, device="cuda:0"):
        """Files of ONE size -> uint8 tensor [N, H, W, C] ([N, C, H, W] with LAYOUT_CHW) on `device`."""
        import torch
        info = self._decs[0].read_headers(files[0])
        nc = 1 if info.components == 1 else ColorSpace(self._out_cs).num_components()
        h, w = int(info.height), int(info.width)
        shape = (len(files), nc, h, w) if self._layout == LAYOUT_CHW else (len(files), h, w, nc)
        t = torch.empty(shape, dtype=torch.uint8, device=device)
        # torch's caching allocator may hand back a block that work queued on torch's current stream still reads; the
        # library writes it on ITS OWN stream, so that work has to be over first (completion is the library's business)
        torch.cuda.current_stream(t.device).synchronize()
        step = h * w * nc
        if step % 16:
            raise ZjError(ERR_ARG, "to_tensor: width * height * components must be a multiple of 16 (rows of the tensor are the outputs)")
        lens, infos = self.decode(files, [(t.data_ptr() + k * step, step) for k in range(len(files))])
        for i in infos:
            if (int(i.width), int(i.height)) != (w, h):
                raise ZjError(ERR_ARG, "to_tensor: files of different sizes")
        return t


def device_numa_node(device=0):
    """NUMA node of a HIP device (PCI bus id -> sysfs), -1 unknown"""
    return lib().zj_device_numa_node(int(device))


def bind_thread_near_device(device=0):
    """Bind the calling thread (and the threads it starts from now on) to the CPUs of the device's NUMA node; returns the
    node, or -1 if nothing was bound (unknown node, ZJ_NUMA=off, an affinity that excludes the node)."""
    return lib().zj_bind_thread_near_device(int(device))


def thread_numa_node():
    return lib().zj_thread_numa_node()


def variants_available():
    """Kernel variants this build of libzjhip.so carries: [0, 2], plus 1 when it was built with `make VARIANTS=all`."""
    return [v for v in (0, 1, 2) if lib().zj_variant_available(v)]


def pointer_device(p):
    """HIP device that owns device pointer p, or a negative zj_status (host memory, unknown pointer)."""
    return lib().zj_pointer_device(p)


def shard_range(nframes, slot, nslots):
    """zj_shard_range: contiguous shard [lo, hi) of `slot` (the rule of shard.shard_range, in the library)."""
    lo, hi = C.c_size_t(0), C.c_size_t(0)
    lib().zj_shard_range(nframes, slot, nslots, C.byref(lo), C.byref(hi))
    return lo.value, hi.value


class Multi:
    """zj_multi: image-level sharding of plane batches over device slots (one context + one host thread per slot)."""

    def __init__(self, devices):
        devs = (C.c_int * len(devices))(*devices)
        st = C.c_int(0)
        self._m = lib().zj_multi_create(devs, len(devices), C.byref(st))
        if not self._m:
            raise ZjError(st.value, "zj_multi_create")
        self.nslots = len(devices)

    def close(self):
        if getattr(self, "_m", None):
            lib().zj_multi_destroy(self._m)
            self._m = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def slot_stats(self):
        """[(device, frames decoded)] per slot"""
        res = []
        for k in range(self.nslots):
            d, n = C.c_int(0), C.c_size_t(0)
            lib().zj_multi_slot_stats(self._m, k, C.byref(d), C.byref(n))
            res.append((d.value, n.value))
        return res

    def slot_numa(self):
        """[(NUMA node of the slot's device, node its host thread runs on, bound?)] per slot; -1 = unknown"""
        res = []
        for k in range(self.nslots):
            a, b, c = C.c_int(-1), C.c_int(-1), C.c_int(0)
            lib().zj_multi_slot_numa(self._m, k, C.byref(a), C.byref(b), C.byref(c))
            res.append((a.value, b.value, bool(c.value)))
        return res

    def decode_planes(self, desc, planes, nframes):
        """Packed host frames -> packed host pixels, sharded over the slots."""
        arrs = [_i16(p) for p in planes]
        while len(arrs) < 3:
            arrs.append(np.zeros(8, np.int16))
        out = np.empty(nframes * lib().zj_out_len(C.byref(desc)), np.uint8)
        sts = (C.c_int * self.nslots)()
        rc = lib().zj_multi_decode_planes_batch(self._m, C.byref(desc), nframes, _ptr(arrs[0]), _ptr(arrs[1]), _ptr(arrs[2]),
                                                _ptr(out), sts)
        if rc:
            raise ZjError(rc, f"zj_multi_decode_planes_batch (per slot: {list(sts)})")
        return out

    def decode_frames(self, desc, frames_planes):
        """Host frames that are independent allocations, sharded over the slots; returns the per-frame arrays."""
        n = len(frames_planes)
        arrs = [[_i16(p) for p in pl] for pl in frames_planes]
        nc = max(len(a) for a in arrs)
        outs = [np.empty(lib().zj_out_len(C.byref(desc)), np.uint8) for _ in range(n)]
        tab = [(C.c_void_p * n)(*[a[c].ctypes.data for a in arrs]) if c < nc else None for c in range(3)]
        otab = (C.c_void_p * n)(*[o.ctypes.data for o in outs])
        sts = (C.c_int * self.nslots)()
        rc = lib().zj_multi_decode_frames(self._m, C.byref(desc), n, tab[0], tab[1], tab[2], otab, sts)
        if rc:
            raise ZjError(rc, f"zj_multi_decode_frames (per slot: {list(sts)})")
        return outs

    def decode_frames_device(self, desc, d_y, d_cb, d_cr, d_out):
        """Device pointers per frame (frame f on the device of the slot whose shard holds f); returns when all are done."""
        n = len(d_y)
        arr = lambda v: (C.c_void_p * n)(*v) if v is not None else None
        sts = (C.c_int * self.nslots)()
        rc = lib().zj_multi_decode_frames_device(self._m, C.byref(desc), n, arr(d_y), arr(d_cb), arr(d_cr), arr(d_out), sts)
        if rc:
            raise ZjError(rc, f"zj_multi_decode_frames_device (per slot: {list(sts)})")


class Pool:
    """zj_pool: persistent host workers (entropy decoder + GPU context each) for batches of JPEG files.  `devices` (a
    list) makes it a multi-device pool (zj_pool_create_multi): `threads` entropy workers per device slot."""

    def __init__(self, threads=4, options=None, device=0, devices=None):
        o = options.to_c() if options is not None else Options()
        st = C.c_int(0)
        if devices is not None:
            devs = (C.c_int * len(devices))(*devices)
            self._p = lib().zj_pool_create_multi(devs, len(devices), int(threads), C.byref(o), C.byref(st))
        else:
            self._p = lib().zj_pool_create(int(device), int(threads), C.byref(o), C.byref(st))
        if not self._p:
            raise ZjError(st.value, "zj_pool_create")
        self._out_cs = int(o.out_colorspace)

    def device_stats(self):
        """[(device, GPU-stage seconds, files)] per device slot"""
        res = []
        for k in range(lib().zj_pool_devices(self._p)):
            d, s_, n = C.c_int(0), C.c_double(0), C.c_size_t(0)
            lib().zj_pool_device_stats(self._p, k, C.byref(d), C.byref(s_), C.byref(n))
            res.append((d.value, s_.value, n.value))
        return res

    def slot_numa(self):
        """[(NUMA node of the slot's device, threads bound there, threads)] per device slot; node -1 = unknown"""
        res = []
        for k in range(lib().zj_pool_devices(self._p)):
            a, b, c = C.c_int(-1), C.c_int(0), C.c_int(0)
            lib().zj_pool_slot_numa(self._p, k, C.byref(a), C.byref(b), C.byref(c))
            res.append((a.value, b.value, c.value))
        return res

    def close(self):
        if getattr(self, "_p", None):
            lib().zj_pool_destroy(self._p)
            self._p = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    @property
    def threads(self):
        return lib().zj_pool_threads(self._p)

    def stats(self):
        """(entropy-stage seconds, GPU-stage seconds, files) accumulated since creation."""
        e, g, n = C.c_double(0), C.c_double(0), C.c_size_t(0)
        lib().zj_pool_stats(self._p, C.byref(e), C.byref(g), C.byref(n))
        return e.value, g.value, n.value

    def decode_files_device(self, blobs, d_outs, caps, raise_on_error=True):
        """blobs -> pixels left in HBM at the device pointers d_outs (capacities caps).  Returns (lengths, infos, statuses)."""
        n = len(blobs)
        arrs = [np.frombuffer(bytes(b), np.uint8) for b in blobs]
        bufs = (C.c_void_p * n)(*[a.ctypes.data for a in arrs])
        lens = (C.c_size_t * n)(*[a.size for a in arrs])
        optr = (C.c_void_p * n)(*d_outs)
        capv = (C.c_size_t * n)(*caps)
        olen = (C.c_size_t * n)()
        infos = (ImageInfo * n)()
        sts = (C.c_int * n)()
        rc = lib().zj_pool_decode_files_device(self._p, n, bufs, lens, optr, capv, olen, infos, sts)
        if rc and raise_on_error:
            raise DecodeError(rc, lib().zj_pool_error(self._p).decode(errors="replace"))
        return list(olen), list(infos), list(sts)

    def to_tensor(self, blobs, device="cuda:0", layout=LAYOUT_HWC):
        """Files of ONE size -> uint8 tensor [N, H, W, C] on `device` (the pool's): the workers prepare, the submitters
        finish in batches into the rows of the tensor (with a GPU entropy setting nothing but the files crosses PCIe)."""
        import torch
        dec = Decoder()
        info = dec.read_headers(blobs[0])
        dec.close()
        nc = 1 if info.components == 1 else ColorSpace(self._out_cs).num_components()
        h, w = int(info.height), int(info.width)
        t = torch.empty((len(blobs), nc, h, w) if layout == LAYOUT_CHW else (len(blobs), h, w, nc), dtype=torch.uint8, device=device)
        torch.cuda.current_stream(t.device).synchronize()   # see FileBatchDecoder.to_tensor: the pool writes on its own streams
        step = h * w * nc
        if step % 16:
            raise ZjError(ERR_ARG, "to_tensor: width * height * components must be a multiple of 16")
        lens, infos, sts = self.decode_files_device(blobs, [t.data_ptr() + k * step for k in range(len(blobs))], [step] * len(blobs))
        if any(int(i.width) != w or int(i.height) != h for i in infos):
            raise ZjError(ERR_ARG, "to_tensor: files of different sizes")
        return t

    def decode_files(self, blobs, outs=None, raise_on_error=True):
        """blobs: list of bytes-like JPEG files.  Returns (list of uint8 arrays, list of ImageInfo, statuses).
        `outs` may supply preallocated uint8 arrays (e.g. views of pinned memory)."""
        n = len(blobs)
        arrs = [np.frombuffer(bytes(b), np.uint8) for b in blobs]
        if outs is None:
            dec = Decoder()
            outs = []
            for a in arrs:
                try:
                    info = dec.read_headers(a)
                    nc = 1 if info.components == 1 else ColorSpace(self._out_cs).num_components()
                    outs.append(np.zeros(int(info.width) * int(info.height) * nc, np.uint8))
                except DecodeError:
                    outs.append(np.zeros(16, np.uint8))
            dec.close()
        bufs = (C.c_void_p * n)(*[a.ctypes.data for a in arrs])
        lens = (C.c_size_t * n)(*[a.size for a in arrs])
        optr = (C.c_void_p * n)(*[o.ctypes.data for o in outs])
        caps = (C.c_size_t * n)(*[o.size for o in outs])
        olen = (C.c_size_t * n)()
        infos = (ImageInfo * n)()
        sts = (C.c_int * n)()
        rc = lib().zj_pool_decode_files(self._p, n, bufs, lens, optr, caps, olen, infos, sts)
        if rc and raise_on_error:
            raise DecodeError(rc, lib().zj_pool_error(self._p).decode(errors="replace"))
        return [o[: olen[i]] for i, o in enumerate(outs)], list(infos), list(sts)
