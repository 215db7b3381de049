"""GPU box: very large images (up to the 16384 limit of the options) through the device entropy stage and the CPU walker."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import importlib, io, sys, time
import numpy as np
from PIL import Image
Image.MAX_IMAGE_PIXELS = None
zj = importlib.import_module("zune-jpeg_amd")
ctx = zj.Context()
for (w, h, ss) in ((16384, 8192, 2), (16000, 4000, 0), (9001, 7001, 1)):
    rng = np.random.default_rng(w)
    small = rng.integers(0, 256, (h // 32 + 1, w // 32 + 1, 3), dtype=np.uint8)
    img = Image.fromarray(small).resize((w, h), Image.BICUBIC)
    b = io.BytesIO(); img.save(b, "JPEG", quality=90, subsampling=ss); data = b.getvalue()
    del img
    oc = zj.ZuneJpegOptions(); oc.num_threads = 16
    og = zj.ZuneJpegOptions(); og.entropy = zj.ENTROPY_GPU
    c, g = zj.Decoder(oc, ctx), zj.Decoder(og, ctx)
    t = time.perf_counter(); want = c.decode_buffer(data); tc = time.perf_counter() - t
    t = time.perf_counter(); got = g.decode_buffer(data); tg = time.perf_counter() - t
    t = time.perf_counter(); got = g.decode_buffer(data); tg = min(tg, time.perf_counter() - t)
    print(f"{w}x{h} ss{ss} {len(data)/1e6:.1f} MB: same={np.array_equal(got, want)} status={g.gpu_status()} on device={g.scan_blob() is not None} rounds={ctx.scan_stats()[0]} cpu {tc*1e3:.0f} ms gpu {tg*1e3:.0f} ms", flush=True)
    c.close(); g.close()
