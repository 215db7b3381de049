"""CPU: the stride arithmetic of zune-jpeg_amd/tensors.py on host tensors (no decode; the GPU half is tests/test_gpu_pitch.py)."""
import ctypes as C
import importlib

import numpy as np
import pytest

zj = importlib.import_module("zune-jpeg_amd")
tz = importlib.import_module("zune-jpeg_amd.tensors")


@pytest.mark.parametrize("cs,layout,shape_of", [(0, 0, lambda h, w: (h, w, 3)), (0, 1, lambda h, w: (3, h, w)), (1, 0, lambda h, w: (h, w)),
                                                (5, 0, lambda h, w: (h, w, 4)), (2, 0, lambda h, w: (h, w, 3))])
@pytest.mark.parametrize("align", [0, 128, 16])
def test_views_address_the_bytes_the_header_describes(cs, layout, shape_of, align):
    import torch
    w, h, n = 250, 7, 3
    qts = [np.ones(64, np.int32)] * 3
    d = zj.FrameDesc.make(w, h, 2, 2, 3, cs, qts, out_layout=layout)
    if align:
        d = tz.padded_desc(d, align)
        assert d.out_pitch % align == 0 and d.out_pitch >= tz.row_bytes(d)
    out_len = zj.lib().zj_out_len(C.byref(d))
    ncomp = zj.ColorSpace(cs).num_components()
    row = w if layout == 1 else w * ncomp
    pitch = d.out_pitch or row
    assert out_len == pitch * h * (3 if layout == 1 else 1)
    storage = torch.arange(n * out_len, dtype=torch.int64).to(torch.uint8)   # byte k holds k mod 256
    flat = np.arange(n * out_len, dtype=np.int64)
    view = tz.view_of(d, storage, n)
    assert tuple(view.shape) == (n,) + shape_of(h, w)
    # element (f, y, x, c) / (f, c, y, x) / (f, y, x) lives at the byte offset include/zjhip.h defines
    rng = np.random.default_rng(5)
    for _ in range(50):
        f, y, x, c = int(rng.integers(n)), int(rng.integers(h)), int(rng.integers(w)), int(rng.integers(ncomp))
        if layout == 1:
            off, v = f * out_len + c * pitch * h + y * pitch + x, view[f, c, y, x]
        elif ncomp == 1:
            off, v = f * out_len + y * pitch + x, view[f, y, x]
        else:
            off, v = f * out_len + y * pitch + x * ncomp + c, view[f, y, x, c]
        assert int(v) == int(flat[off]) % 256
