"""
zune-jpeg_amd -- MI355X-native implementation of zune-jpeg's post-entropy pixel path
(dequantize + 8x8 integer IDCT, chroma up-sampling, YCbCr->RGB), behind the C ABI of
include/zjhip.h.  This Python package is only the thin ctypes binding used by tests/ and bench.py;
the product is libzjhip.so (zune-jpeg_amd/csrc).

There is NO CPU fallback: if libzjhip.so is missing, or no HIP device is usable, the calls raise.
Import with importlib.import_module("zune-jpeg_amd") (the directory name carries a hyphen).
"""
from .host import (  # noqa: F401
    BACKEND_AVX2, BACKEND_HIP, BACKEND_SCALAR, FLAG_CLAMP_DC, FLAG_CORRECTED, FLAG_EDGE_REPLICATE, FLAG_FULL_AC_VALUES, FLAG_PLAIN_TAIL, LAYOUT_CHW, LAYOUT_HWC, ColorSpace, Component, Context, DecodeError, Decoder, FrameDesc,
    ENTROPY_CPU, ENTROPY_GPU, ENTROPY_GPU_ALWAYS, HUFF_ST, RETRY_CPU, FileBatchDecoder, ImageInfo, Pool, ZjError,
    ZuneJpegOptions, abi_symbols, choose_idct_func, choose_upsample_func,
    choose_ycbcr_to_rgb_convert_func, device_count, finish_pixels_batch, lib, lib_path, num_components,
    SCATTER_MAX, Multi, pointer_device, shard_range, device_numa_node, bind_thread_near_device, thread_numa_node, variants_available,
)
