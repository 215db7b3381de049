"""Decoded frames as PyTorch-ROCm tensors (the consumer-side layouts).  Plumbing over the C ABI, nothing
computes here: the tensors are views of the bytes zj_decode_planes_device writes.

  layout                      desc                                   view (uint8)                 strides (bytes)
  interleaved (the reference)  out_layout = LAYOUT_HWC                [N, H, W, C]                 out_len, pitch, C, 1
  planar                       out_layout = LAYOUT_CHW, RGB           [N, 3, H, W]                 out_len, pitch*H, pitch, 1
  grayscale                    out_colorspace = GRAYSCALE             [N, H, W]                    out_len, pitch, 1

pitch = desc.out_pitch, or the row's own length when that is 0 (the reference's tight rows, /root/reference/src/mcu.rs:375-379).
With a padded pitch (padded_desc: the next multiple of 128 bytes, which keeps every tile's row segment on whole cache
lines -- DESIGN.md 4.0) the view is simply strided over the padding; `.contiguous()` gives the tight tensor.
"""
import copy
import ctypes as C

from .host import LAYOUT_CHW, ColorSpace, lib


def row_bytes(desc):
    """bytes of one output row (CHW: of one plane's row)"""
    ncomp = ColorSpace(desc.out_colorspace).num_components()
    planar = desc.out_layout == LAYOUT_CHW and ncomp == 3
    return desc.width if planar else desc.width * ncomp


def padded_desc(desc, align=128):
    """a copy of `desc` whose rows lie at the next multiple of `align` bytes (zj_frame_desc.out_pitch; device outputs only)"""
    d = copy.copy(desc)
    d.out_pitch = (row_bytes(desc) + align - 1) // align * align
    return d


def output_tensor(desc, nframes, device, fill=None):
    """(storage, view): a flat uint8 tensor of nframes * zj_out_len(desc) bytes on `device` to hand to
    Context.decode_planes_device(..., storage.data_ptr()), and its view in the layout the descriptor names."""
    import torch
    out_len = lib().zj_out_len(C.byref(desc))
    if out_len == 0:
        raise ValueError("zj_out_len(desc) == 0: not a decodable frame descriptor")
    storage = torch.empty(nframes * out_len, dtype=torch.uint8, device=device) if fill is None else \
        torch.full((nframes * out_len,), fill, dtype=torch.uint8, device=device)
    return storage, view_of(desc, storage, nframes)


def view_of(desc, storage, nframes):
    """the [N, H, W, C] / [N, 3, H, W] / [N, H, W] view of a flat output buffer (see the module docstring)"""
    out_len = lib().zj_out_len(C.byref(desc))
    ncomp = ColorSpace(desc.out_colorspace).num_components()
    pitch = desc.out_pitch or row_bytes(desc)
    h, w = desc.height, desc.width
    if desc.out_layout == LAYOUT_CHW and ncomp == 3:
        return storage.as_strided((nframes, 3, h, w), (out_len, pitch * h, pitch, 1))
    if ncomp == 1:
        return storage.as_strided((nframes, h, w), (out_len, pitch, 1))
    return storage.as_strided((nframes, h, w, ncomp), (out_len, pitch, ncomp, 1))


def decode_to_tensor(ctx, desc, planes, nframes=1, stream=None):
    """planes: three int16 CUDA tensors (Y, Cb, Cr; `nframes` frames back to back, include/zjhip.h "whole-frame layout").
    Launches on `stream` (a torch.cuda.Stream, or None = torch's current stream) and returns the view; the caller
    synchronises as with any other kernel on that stream."""
    import torch
    dev = planes[0].device
    cur = torch.cuda.current_stream(dev)
    s = stream if stream is not None else cur
    # The output is allocated (and, with a fill, written) under `s`, so that the caching allocator ties the block to the
    # stream the kernel writes it on: allocated under another stream it could be handed out again while the launch is
    # still pending.  The planes were produced on the caller's current stream: `s` waits for it before the launch.
    with torch.cuda.stream(s):
        storage, view = output_tensor(desc, nframes, dev)
    if s != cur:
        s.wait_stream(cur)
        for p in planes:
            p.record_stream(s)
    ctx.decode_planes_device(desc, nframes, planes[0].data_ptr(), planes[1].data_ptr(), planes[2].data_ptr(), storage.data_ptr(), s.cuda_stream)
    return view
