#!/usr/bin/env python3
"""Robustness of the CPU entropy front-end on damaged files: zj_jpeg.cpp built alone with AddressSanitizer + UBSan
(tests/fuzz/jpeg_stubs.cpp stands in for the GPU calls) and fed thousands of mutated baseline / progressive / DRI
files: bit flips, byte splices, truncations, marker-length edits.  Any status is fine; a sanitizer report is not.
Every third file also goes through zj_decoder_prepare with the device entropy setting (the byte-level preparation of
the scan), and what it prepares through the DEVICE code of the entropy stage (zj_huff_device.h) run thread by thread by
the emulation harness, also built with the sanitizers: a damaged scan must neither read nor write out of bounds there.

    python tools/fuzz_frontend.py [--iters 4000]      (re-executes itself under LD_PRELOAD=libasan.so)
"""
import argparse
import ctypes as C
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SO = "/tmp/libzjjpeg_asan.so"
EMU = "/tmp/libzjemu_asan.so"


def build():
    subprocess.check_call(["g++", "-O1", "-g", "-std=c++17", "-fPIC", "-shared", "-pthread", "-fsanitize=address,undefined",
                           "-fno-omit-frame-pointer", "-o", SO,
                           os.path.join(ROOT, "zune-jpeg_amd", "csrc", "zj_jpeg.cpp"), os.path.join(ROOT, "tests", "fuzz", "jpeg_stubs.cpp")])
    subprocess.check_call(["g++", "-O1", "-g", "-std=c++17", "-fPIC", "-shared", "-fno-strict-aliasing", "-fsanitize=address,undefined",
                           "-fno-sanitize=alignment", "-fno-omit-frame-pointer", "-Wno-unknown-pragmas", "-o", EMU,
                           os.path.join(ROOT, "tests", "emu", "zj_emu.cpp")])


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=4000)
    ap.add_argument("--child", action="store_true")
    a = ap.parse_args()
    if not a.child:
        build()
        asan = subprocess.check_output(["gcc", "-print-file-name=libasan.so"]).decode().strip()
        env = dict(os.environ, LD_PRELOAD=asan, ASAN_OPTIONS="detect_leaks=0:halt_on_error=1:abort_on_error=1",
                   UBSAN_OPTIONS="print_stacktrace=1:halt_on_error=1")
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "--child", "--iters", str(a.iters)], env=env)
        print("sanitizer-clean" if r.returncode == 0 else f"FAILED (exit {r.returncode})")
        sys.exit(r.returncode)

    import numpy as np
    sys.path[:0] = [ROOT, os.path.join(ROOT, "tools")]
    import importlib
    import jpeg_enc
    synth = importlib.import_module("zune-jpeg_amd.synth")
    L = C.CDLL(SO)
    L.zj_decoder_new.restype = C.c_void_p
    L.zj_decoder_new.argtypes = [C.c_void_p]
    L.zj_decoder_free.argtypes = [C.c_void_p]
    L.zj_decoder_decode_coefficients.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    L.zj_decoder_prepare.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p]
    L.zj_decoder_scan_blob.argtypes = [C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_size_t)]
    E = C.CDLL(EMU)
    E.zje_huff_decode.argtypes = [C.c_void_p] * 6
    E.zje_huff_plane_len.restype = C.c_size_t
    E.zje_huff_plane_len.argtypes = [C.c_void_p, C.c_int]

    class Opt(C.Structure):
        _fields_ = [(n, C.c_int32) for n in ("out_colorspace", "strict_mode", "max_width", "max_height", "max_scans", "num_threads", "pinned_planes", "flags", "out_layout", "entropy")]

    qts = synth.quant_tables(85)
    seeds = []
    for (w, h, hs, vs) in [(64, 48, 2, 2), (50, 37, 1, 1), (40, 24, 2, 1), (33, 70, 1, 2)]:
        pl = jpeg_enc.small_planes(w, h, hs, vs, 3, seed=w)
        seeds.append(jpeg_enc.encode_baseline(pl, qts, w, h, hs, vs, 3))
        seeds.append(jpeg_enc.encode_baseline(pl, qts, w, h, hs, vs, 3, restart=3))
        seeds.append(jpeg_enc.encode_progressive(pl, qts, w, h, hs, vs, 3))
    for name in ("test-baseline.jpg", "test-progressive.jpg"):
        seeds.append(open(os.path.join(ROOT, "tests", "golden", name), "rb").read()[:6000])  # headers + start of data
    # flat pages: periodic runs, which the device stage bridges with a rule of its own (csrc/zj_huff.h)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from test_huff_emu import document_like
    seeds.append(document_like(640, 480))
    seeds.append(document_like(512, 512, gray=True))
    seeds.append(document_like(400, 300, subsampling=0))
    # scans of tens of KB: what decode_mcus_v2 carries (it leaves a scan's last 4 KB to the block-at-a-time decoder, so the
    # small seeds above never reach it) -- standard and optimised tables, with and without restart intervals
    import io
    from PIL import Image, ImageFile
    ImageFile.MAXBLOCK = max(ImageFile.MAXBLOCK, 1 << 22)
    rs = np.random.default_rng(11)
    for (w, h, sub, q, opt, rst) in [(512, 384, 2, 85, False, 0), (400, 300, 0, 95, True, 0), (640, 256, 1, 70, True, 2)]:
        small = rs.integers(0, 256, (h // 16, w // 16, 3), dtype=np.uint8)
        img = Image.fromarray(small, "RGB").resize((w, h), Image.BICUBIC)
        img = Image.fromarray(np.clip(np.asarray(img).astype(np.int16) + rs.integers(-30, 31, (h, w, 3), dtype=np.int16), 0, 255).astype(np.uint8), "RGB")
        bio = io.BytesIO()
        img.save(bio, "JPEG", quality=q, subsampling=sub, optimize=opt, **({"restart_marker_rows": rst} if rst else {}))
        seeds.append(bio.getvalue())
    os.environ["ZJ_PAR_MIN_CHUNK"] = "512"  # the three-thread decoder enters scans without restart markers at three points
                                            # (scan_baseline_parallel) whenever more than 8 KB + 1.5 KB of scan are there
    rng = np.random.default_rng(7)
    decs = []
    for threads in (1, 3):
        o = Opt(0, 0, 0, 0, 0, threads, 0, 0, 0, 0)
        decs.append(L.zj_decoder_new(C.byref(o)))
    o = Opt(0, 0, 0, 0, 0, 1, 0, 0, 0, 2)
    gdec = L.zj_decoder_new(C.byref(o))
    emu_runs = emu_kept = 0
    stats = {}
    desc = (C.c_char * 1024)()
    info = (C.c_char * 64)()
    for it in range(a.iters):
        b = bytearray(seeds[it % len(seeds)])
        kind = it % 5
        if kind == 0:
            for _ in range(int(rng.integers(1, 8))):
                b[int(rng.integers(len(b)))] ^= 1 << int(rng.integers(8))
        elif kind == 1:
            b = b[: int(rng.integers(2, len(b)))]
        elif kind == 2:
            i = int(rng.integers(len(b)))
            b[i:i] = bytes(rng.integers(0, 256, int(rng.integers(1, 40)), dtype=np.uint8))
        elif kind == 3:
            i, j = sorted(int(x) for x in rng.integers(0, len(b), 2))
            del b[i:min(j, i + 200)]
        else:
            i = int(rng.integers(len(b) - 4))
            b[i:i + 2] = bytes([0xFF, int(rng.integers(0xC0, 0xFF))])
        arr = np.frombuffer(bytes(b), np.uint8)
        rc = L.zj_decoder_decode_coefficients(decs[it & 1], arr.ctypes.data, arr.size, desc, None, None, info)
        stats[rc] = stats.get(rc, 0) + 1
        if it % 3 == 0 and L.zj_decoder_prepare(gdec, arr.ctypes.data, arr.size, desc, info) == 0:
            p, n = C.c_void_p(), C.c_size_t(0)
            if L.zj_decoder_scan_blob(gdec, C.byref(p), C.byref(n)) == 0:
                st = C.c_uint32(0)
                # planes of exactly the size the scan's header states (HuffScan.comp[i].bw * bh * 64, zj_huff.h), so
                # that a store past a plane is a heap overflow the sanitizer sees
                planes = [np.zeros(max(64, int(E.zje_huff_plane_len(p, c))), np.int16) for c in range(3)]
                E.zje_huff_decode(p, planes[0].ctypes.data, planes[1].ctypes.data, planes[2].ctypes.data, C.byref(st), None)
                emu_runs += 1
                emu_kept += st.value == 0
    print(f"device entropy stage (emulated): {emu_runs} prepared scans, {emu_kept} kept by the device")
    L.zj_decoder_free(gdec)
    for d in decs:
        L.zj_decoder_free(d)
    print("statuses:", dict(sorted(stats.items())))


if __name__ == "__main__":
    main()
