"""GPU: one frame whose planes are still being written (zj_frame_begin / _rows_ready / _end, round 6): the strips go to the GPU
as the caller declares MCU rows final -- what the reference does when it hands strip N to a pool thread while its Huffman
decoder is in strip N + 1 (src/mcu.rs:356-368).  Same bytes as zj_decode_planes, whatever the increments, for device, pinned
and pageable outputs; and zj_decoder_decode_buffer, which streams baseline files out of pinned planes, against itself with
ZJ_STREAM=off."""
import ctypes as C
import importlib
import io
import os

import numpy as np
import pytest

import oracle_c as oc

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
MODES = {"none": (1, 1), "h": (2, 1), "v": (1, 2), "hv": (2, 2)}


@pytest.fixture(scope="module")
def zj():
    return importlib.import_module("zune-jpeg_amd")


def _pinned(zj, nbytes):
    L = zj.lib()
    L.zj_alloc_pinned.restype = C.c_void_p
    L.zj_alloc_pinned.argtypes = [C.c_size_t]
    L.zj_free_pinned.argtypes = [C.c_void_p]
    p = L.zj_alloc_pinned(nbytes)
    assert p
    return p, np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_uint8)), shape=(nbytes,))


@pytest.mark.parametrize("mode", list(MODES))
@pytest.mark.parametrize("wh", [(1024, 768), (2500, 333), (1840, 1040), (96, 40)])
@pytest.mark.parametrize("out_kind", ["device", "pinned", "pageable"])
def test_streamed_frame_equals_the_oracle(zj, synth, mode, wh, out_kind):
    hs, vs = MODES[mode]
    w, h = wh
    out_cs = [oc.RGB, oc.GRAYSCALE, oc.YCBCR][(w + hs + 2 * vs) % 3]
    planes, qts = synth.make_frame(w, h, hs, vs, 3, seed=w + h)
    rc, exp = oc.decode_planes(oc.make_frame(w, h, hs, vs, 3, out_cs, qts), planes)
    assert rc == 0
    d = zj.FrameDesc.make(w, h, hs, vs, 3, out_cs, qts)
    ctx = zj.Context(zj.BACKEND_HIP, 0)
    pins = []
    try:
        addr = []
        for pl in planes:                                   # pinned planes, filled "as the walker goes": zeros first
            p, view = _pinned(zj, pl.nbytes)
            pins.append(p)
            view[:] = 0
            addr.append((p, view, pl.view(np.uint8)))
        mcu_y = (h + 8 * vs - 1) // (8 * vs)
        out_len = exp.size
        rng = np.random.default_rng(w * 7 + h)
        for unit_mb in (None, "1"):                          # the default unit, and many small ones
            if unit_mb:
                os.environ["ZJ_STREAM_UNIT_MB"] = unit_mb
            try:
                if out_kind == "device":
                    dptr = ctx.device_alloc(out_len)
                    zj.lib().zj_device_memset(ctx.handle, dptr, 0x77, out_len)
                    optr = dptr
                elif out_kind == "pinned":
                    op, oview = _pinned(zj, out_len)
                    pins.append(op)
                    oview[:] = 0x77
                    optr = op
                else:
                    oarr = np.full(out_len, 0x77, np.uint8)
                    optr = oarr.ctypes.data
                for _, view, _ in addr:
                    view[:] = 0
                ctx.frame_begin(d, addr[0][0], addr[1][0], addr[2][0], optr, out_on_device=out_kind == "device")
                done = 0
                while done < mcu_y:
                    step = int(rng.integers(1, max(2, mcu_y // 3)))
                    nxt = min(mcu_y, done + step)
                    for c, (_, view, src) in enumerate(addr):   # rows [done, nxt) of every plane become final
                        rows = (vs if c == 0 else 1)
                        per_row = src.size // (mcu_y * rows) * rows
                        view[done * per_row:nxt * per_row] = src[done * per_row:nxt * per_row]
                    done = nxt
                    ctx.frame_rows_ready(done)
                    if rng.integers(0, 3) == 0:
                        ctx.frame_rows_ready(done)              # (saying it twice changes nothing)
                ctx.frame_end()
                if out_kind == "device":
                    got = np.empty(out_len, np.uint8)
                    ctx.d2h(got, dptr)
                    ctx.device_free(dptr)
                elif out_kind == "pinned":
                    got = oview.copy()
                else:
                    got = oarr
                bad = np.nonzero(got != exp)[0]
                assert bad.size == 0, (mode, wh, out_kind, unit_mb, bad[:8], bad.size)
            finally:
                os.environ.pop("ZJ_STREAM_UNIT_MB", None)
        # an aborted frame leaves the context usable
        scrap = np.zeros(out_len, np.uint8)                      # (kept alive: zj_frame_begin clears the rows no strip reaches)
        ctx.frame_begin(d, addr[0][0], addr[1][0], addr[2][0], scrap.ctypes.data)
        ctx.frame_rows_ready(mcu_y // 2)
        ctx.frame_abort()
        assert np.array_equal(ctx.decode_planes(d, planes), exp)
        with pytest.raises(zj.ZjError):
            ctx.frame_rows_ready(1)                              # no frame is open
    finally:
        for p in pins:
            zj.lib().zj_free_pinned(p)
        ctx.close()


def _jpeg(seed, w, h, subsampling, quality, restart_rows=0):
    from PIL import Image
    rng = np.random.default_rng(seed)
    small = rng.integers(0, 256, (max(2, h // 16), max(2, w // 16), 3), dtype=np.uint8)
    img = Image.fromarray(small, "RGB").resize((w, h), Image.BICUBIC)
    img = Image.fromarray(np.clip(np.asarray(img).astype(np.int16) + rng.integers(-20, 21, (h, w, 3), dtype=np.int16), 0, 255).astype(np.uint8), "RGB")
    b = io.BytesIO()
    kw = {"restart_marker_rows": restart_rows} if restart_rows else {}
    img.save(b, "JPEG", quality=quality, subsampling=subsampling, **kw)
    return b.getvalue()


def _decode_buffer(zj, ctx, data, stream, threads=1, out_cs=None, pinned_out=None):
    if stream:
        os.environ.pop("ZJ_STREAM", None)
    else:
        os.environ["ZJ_STREAM"] = "off"
    try:
        o = zj.ZuneJpegOptions()
        o.num_threads, o.pinned_planes = threads, True
        if out_cs is not None:
            o.out_colorspace = out_cs
        dec = zj.Decoder(o, ctx)
        try:
            return ("ok", dec.decode_buffer(data, out=pinned_out).copy())
        except zj.DecodeError as e:
            return ("error", str(e))
        finally:
            dec.close()
    finally:
        os.environ.pop("ZJ_STREAM", None)


def test_decode_buffer_streams_baseline_files_to_the_same_bytes(zj):
    ctx = zj.Context(zj.BACKEND_HIP, 0)
    try:
        files = [open(os.path.join(HERE, "golden", n), "rb").read() for n in ("test-baseline.jpg", "test-progressive.jpg")]
        files.append(open(os.path.join(HERE, "golden", "ref", "medium_horiz_samp_2500x1786.jpg"), "rb").read())
        files.append(open(os.path.join(HERE, "golden", "ref", "single_qt.jpeg"), "rb").read())       # DRI 1005
        files += [_jpeg(3, 2048, 1536, 2, 90), _jpeg(4, 1600, 1200, 1, 85, restart_rows=2), _jpeg(5, 333, 277, 0, 100),
                  _jpeg(6, 4096, 4096, 2, 90)]
        for i, data in enumerate(files):
            for cs in (zj.ColorSpace.RGB, zj.ColorSpace.GRAYSCALE):
                one = _decode_buffer(zj, ctx, data, False, 1, cs)
                assert one[0] == "ok"
                # (four threads: restart segments side by side, or -- the files without restart markers, from 96 KB of scan --
                # the scan entered at four points, its first chunk's rows on their way to the GPU while the rest is decoded)
                for threads in (1, 4, 7):
                    for stream in (True, False):
                        a = _decode_buffer(zj, ctx, data, stream, threads, cs)
                        assert a[0] == "ok" and np.array_equal(a[1], one[1]), (i, threads, stream, cs)
        # into pinned memory: the downloads overlap as well
        data = files[4]
        ref = _decode_buffer(zj, ctx, data, False)[1]
        p, view = _pinned(zj, ref.size)
        try:
            view[:] = 0x55
            got = _decode_buffer(zj, ctx, data, True, pinned_out=view)
            assert got[0] == "ok" and np.array_equal(got[1], ref) and np.array_equal(view, ref)
        finally:
            zj.lib().zj_free_pinned(p)
        # damage in mid-scan: the same error with and without streaming, and the decoder / context go on working
        good = files[4]
        sos = good.index(b"\xff\xda")
        rng = np.random.default_rng(9)
        seen_error = 0
        for trial in range(16):
            d = bytearray(good)
            at = int(rng.integers(sos + 200, len(d) - 6000))
            d[at:at + 2] = [b"\xff\xd9", b"\xff\x17", bytes([d[at] ^ 0x10, d[at + 1]])][trial % 3]
            a = _decode_buffer(zj, ctx, bytes(d), True)
            b = _decode_buffer(zj, ctx, bytes(d), False)
            assert a[0] == b[0] and (np.array_equal(a[1], b[1]) if a[0] == "ok" else a[1] == b[1]), (trial, a[0], b[0])
            c = _decode_buffer(zj, ctx, bytes(d), True, threads=4)
            assert c[0] == b[0] and (np.array_equal(c[1], b[1]) if c[0] == "ok" else c[1] == b[1]), (trial, "4 threads", c[0], b[0])
            seen_error += a[0] == "error"
            assert np.array_equal(_decode_buffer(zj, ctx, good, True)[1], ref)
    finally:
        ctx.close()


def test_pool_lends_idle_workers_to_a_short_batch(zj):
    """Fewer files than workers: the files get the idle workers' threads (zj_pool.cpp: zj_decoder_set_num_threads; a scan
    without restart markers is then entered at several points) -- same bytes as one decoder on one thread, with the lending
    and without (ZJ_POOL_LEND=off), for batches of 1, 2, 3 and more files than workers, host and device outputs."""
    ctx = zj.Context(zj.BACKEND_HIP, 0)
    try:
        blobs = [_jpeg(11, 2048, 2048, 2, 90), _jpeg(12, 1920, 1080, 0, 92), _jpeg(13, 1600, 1200, 1, 85, restart_rows=2),
                 open(os.path.join(HERE, "golden", "test-progressive.jpg"), "rb").read()]
        refs = [_decode_buffer(zj, ctx, b, False, 1)[1] for b in blobs]
        for lend in (None, "off"):
            if lend:
                os.environ["ZJ_POOL_LEND"] = lend
            try:
                with zj.Pool(threads=8) as pool:
                    for n in (1, 2, 3, 4, 11):
                        files = [blobs[i % len(blobs)] for i in range(n)]
                        for _ in range(2):
                            outs, _, sts = pool.decode_files(files)
                            assert not any(sts)
                            for i in range(n):
                                assert np.array_equal(np.asarray(outs[i]).reshape(-1), refs[i % len(blobs)]), (lend, n, i)
                    # pixels left in HBM
                    n = 2
                    sizes = [refs[i].size for i in range(n)]
                    base = ctx.device_alloc(sum((s + 255) // 256 * 256 for s in sizes))
                    try:
                        ptrs, off = [], 0
                        for s_ in sizes:
                            ptrs.append(base + off)
                            off += (s_ + 255) // 256 * 256
                        pool.decode_files_device(blobs[:n], ptrs, sizes)
                        for i in range(n):
                            got = np.empty(sizes[i], np.uint8)
                            ctx.d2h(got, ptrs[i])
                            assert np.array_equal(got, refs[i]), (lend, "device", i)
                    finally:
                        ctx.device_free(base)
            finally:
                os.environ.pop("ZJ_POOL_LEND", None)
    finally:
        ctx.close()
