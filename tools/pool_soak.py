#!/usr/bin/env python3
"""GPU box: soak test of zj_pool and of the batch entry points with the device entropy stage -- random batches of random
files (sizes, qualities, sampling modes, restart intervals, a few progressive and a few damaged ones) through
zj_pool_decode_files, zj_pool_decode_files_device and zj_decoder_finish_pixels_batch; every result against the file's own
decode with the CPU walker (pixels, or the same status).

    python tools/pool_soak.py [--seconds 120] [--seed 1]"""
import argparse
import importlib
import io
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np  # noqa: E402
from PIL import Image  # noqa: E402
import entropy_soak  # noqa: E402

zj = importlib.import_module("zune-jpeg_amd")


def make_file(rng, same_size=None):
    w, h = same_size if same_size else (int(rng.integers(16, 1400)), int(rng.integers(16, 1000)))
    a = entropy_soak.content(rng, w, h)
    kw = dict(quality=int(rng.integers(20, 99)))
    if rng.integers(0, 8) == 0:
        kw["progressive"] = True
    else:
        kw["subsampling"] = int(rng.integers(0, 3))
        r = int(rng.integers(0, 3))
        if r == 1:
            kw["restart_marker_rows"] = int(rng.integers(1, 3))
    b = io.BytesIO()
    try:
        Image.fromarray(a).save(b, "JPEG", **kw)
    except OSError:  # (libjpeg refuses some combinations)
        b = io.BytesIO()
        Image.fromarray(a).save(b, "JPEG", quality=80)
    data = bytearray(b.getvalue())
    if rng.integers(0, 12) == 0:  # damage the entropy-coded part
        sos = data.index(b"\xff\xda") + 14
        if len(data) > sos + 16:
            data[int(rng.integers(sos, len(data) - 2))] ^= 1 << int(rng.integers(0, 8))
    return bytes(data)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=120)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--devices", default=None, help="comma-separated device slots for a multi-device pool (zj_pool_create_multi), e.g. 0,0")
    ap.add_argument("--entropy", choices=["gpu", "cpu"], default="gpu")
    args = ap.parse_args()
    rng = np.random.default_rng(args.seed)
    ctx = zj.Context()
    o = zj.ZuneJpegOptions()
    o.entropy = zj.ENTROPY_GPU_ALWAYS if args.entropy == "gpu" else zj.ENTROPY_CPU
    pool = zj.Pool(4, o) if not args.devices else zj.Pool(2, o, devices=[int(v) for v in args.devices.split(",")])
    ref = zj.Decoder(None, ctx)
    nfiles = nbatches = bad = 0
    t_end = time.time() + args.seconds
    while time.time() < t_end:
        n = int(rng.integers(1, 40))
        same = (int(rng.integers(64, 900)) // 16 * 16, int(rng.integers(64, 700))) if rng.integers(0, 3) == 0 else None
        files = [make_file(rng, same) for _ in range(n)]
        want = []
        for f in files:
            try:
                want.append(ref.decode_buffer(f))
            except zj.DecodeError as e:
                want.append(e.status)
        how = int(rng.integers(0, 3))
        if how == 0:  # pool, host outputs
            outs, _, sts = pool.decode_files(files, raise_on_error=False)
            got = [o_ if s == 0 else s for o_, s in zip(outs, sts)]
        elif how == 1:  # pool, device outputs (equally spaced when the sizes allow)
            # (a file the reference walk rejects still gets the room its header asks for: with less the pool answers "buffer
            # too small" before it meets the damage -- seen once in session r05ak, a verdict about the caller, not the file)
            def room(f):
                try:
                    wd, ht = Image.open(io.BytesIO(f)).size
                    return wd * ht * 3
                except Exception:  # noqa: BLE001
                    return 1400 * 1000 * 3
            sizes = [w.size if not isinstance(w, int) else room(files[k]) for k, w in enumerate(want)]
            step = (max(sizes) + 255) // 256 * 256
            base = ctx.device_alloc(step * n + 64)
            lens, _, sts = pool.decode_files_device(files, [base + k * step for k in range(n)], [step] * n, raise_on_error=False)
            got = []
            for k in range(n):
                if sts[k]:
                    got.append(sts[k])
                else:
                    a = np.zeros(lens[k], np.uint8)
                    ctx.d2h(a, base + k * step)
                    got.append(a)
            ctx.device_free(base)
        else:  # finish_pixels_batch from one thread
            got = []
            for base_k in range(0, n, 16):
                chunk = files[base_k:base_k + 16]
                decs, prep_err = [], {}
                for k, f in enumerate(chunk):
                    d = zj.Decoder(o, ctx)
                    try:
                        d.prepare(f)
                    except zj.DecodeError as e:
                        prep_err[k] = e.status
                    decs.append(d)
                live = [k for k in range(len(chunk)) if k not in prep_err]
                outs, rcs = zj.finish_pixels_batch([decs[k] for k in live], ctx) if live else ([], [])
                res = dict(zip(live, [(o_ if rc == 0 else rc) for o_, rc in zip(outs, rcs)]))
                for k in range(len(chunk)):
                    got.append(prep_err[k] if k in prep_err else res[k])
                for d in decs:
                    d.close()
        for k, (g, w) in enumerate(zip(got, want)):
            same_result = (g == w) if isinstance(w, int) or isinstance(g, int) else np.array_equal(g, w)
            if not (same_result if isinstance(same_result, bool) else bool(same_result)):
                bad += 1
                print(f"MISMATCH batch {nbatches} file {k} how={how}: want {'status ' + str(w) if isinstance(w, int) else w.size} got {'status ' + str(g) if isinstance(g, int) else g.size}", flush=True)
        nfiles += n
        nbatches += 1
    pool.close()
    print(f"{nbatches} batches, {nfiles} files, {bad} mismatches")
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
