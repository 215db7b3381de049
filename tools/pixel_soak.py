#!/usr/bin/env python3
"""
Soak of the PIXEL path (GPU box): random geometries, sampling modes, output colourspaces, extension flags, layouts, kernel
variants and coefficient statistics through the C ABI against the oracle (oracle/zj_oracle.c), byte for byte, for --seconds
seconds.  Every case goes through one entry point picked at random: zj_decode_planes_batch (packed host frames),
zj_decode_frames (host frames as independent allocations), zj_decode_frames_device (device frames at scattered addresses,
shuffled table order), zj_decode_planes_device_strided (padded device frames), zj_multi_decode_frames (two device slots); half of the device
outputs with their rows at a padded pitch (zj_frame_desc.out_pitch).  Where the reference panics, the ABI must report ZJ_ERR_PANIC.  Checker
infrastructure: the oracle is only the judge here.

    python tools/pixel_soak.py --seconds 600 > gpurun_out/pixel_soak.txt
"""
import argparse
import collections
import importlib
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "oracle")):
    sys.path.insert(0, p)


def decode(zj, ctx, multi, path, d, frames, planes, nframes, rng):
    """the frames through one entry point; returns the packed pixels of all frames"""
    import ctypes as C
    if path == "batch":
        return ctx.decode_planes(d, planes, nframes)
    per = [[np.array(p) for p in f[0]] for f in frames]
    if path == "frames":
        return np.concatenate(ctx.decode_frames(d, per))
    if path == "multi":
        return np.concatenate(multi.decode_frames(d, per))
    out_len = zj.lib().zj_out_len(C.byref(d))
    if path == "frames_device":
        # every plane and every output its own device allocation, made in a shuffled order; table order shuffled again
        jobs = [(f, c) for f in range(nframes) for c in range(4)]
        rng.shuffle(jobs)
        ptr = [[0] * 4 for _ in range(nframes)]
        allp = []
        for f, c in jobs:
            nb = out_len if c == 3 else per[f][c].nbytes
            p = ctx.device_alloc(nb + 64)
            allp.append(p)
            ptr[f][c] = p
            if c < 3:
                ctx.h2d(p, per[f][c])
            elif d.out_pitch:
                zj.lib().zj_device_memset(ctx.handle, p, 0xAA, nb)   # the padding of a padded pitch must keep this
        order = [int(v) for v in rng.permutation(nframes)]
        try:
            ctx.decode_frames_device(d, [ptr[f][0] for f in order], [ptr[f][1] for f in order], [ptr[f][2] for f in order],
                                     [ptr[f][3] for f in order])
            ctx.sync()
            got = np.empty(nframes * out_len, np.uint8)
            for f in range(nframes):
                ctx.d2h(got[f * out_len:(f + 1) * out_len], ptr[f][3])
            return got
        finally:
            for p in allp:
                ctx.device_free(p)
    # strided: frames at a padded, uniform distance inside one allocation per plane
    ylen, clen = per[0][0].size, per[0][1].size
    ys, cs = ylen + 8 * int(rng.integers(0, 9)), clen + 8 * int(rng.integers(0, 9))
    fast = d.width % 16 == 0 and d.width >= 32
    os_ = out_len + (16 * int(rng.integers(0, 9)) if fast else int(rng.integers(0, 40)))
    hy = np.zeros(nframes * ys, np.int16)
    hc = [np.zeros(nframes * cs, np.int16) for _ in range(2)]
    for f in range(nframes):
        hy[f * ys:f * ys + ylen] = per[f][0]
        for c in range(2):
            hc[c][f * cs:f * cs + clen] = per[f][1 + c]
    bufs = [ctx.device_alloc(hy.nbytes + 64), ctx.device_alloc(hc[0].nbytes + 64), ctx.device_alloc(hc[1].nbytes + 64), ctx.device_alloc(nframes * os_ + 64)]
    try:
        ctx.h2d(bufs[0], hy)
        ctx.h2d(bufs[1], hc[0])
        ctx.h2d(bufs[2], hc[1])
        if d.out_pitch:
            zj.lib().zj_device_memset(ctx.handle, bufs[3], 0xAA, nframes * os_)
        ctx.decode_planes_device_strided(d, nframes, bufs[0], bufs[1], bufs[2], bufs[3], ys, cs, os_)
        ctx.sync()
        raw = np.empty(nframes * os_, np.uint8)
        ctx.d2h(raw, bufs[3])
        return np.concatenate([raw[f * os_:f * os_ + out_len] for f in range(nframes)])
    finally:
        for b in bufs:
            ctx.device_free(b)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=300)
    ap.add_argument("--seed", type=int, default=2026)
    a = ap.parse_args()
    import oracle_c as oc
    zj = importlib.import_module("zune-jpeg_amd")
    synth = importlib.import_module("zune-jpeg_amd.synth")
    rng = np.random.default_rng(a.seed)
    ctx = zj.Context(zj.BACKEND_HIP, 0)
    multi = zj.Multi([0, 0])
    modes = [(1, 1), (2, 1), (1, 2), (2, 2)]
    outs = [oc.RGB, oc.GRAYSCALE, oc.YCBCR, oc.RGBA]
    stats = collections.Counter()
    bad = []
    t0 = time.time()
    n = 0
    while time.time() - t0 < a.seconds:
        n += 1
        hs, vs = modes[int(rng.integers(0, 4))]
        kind = int(rng.integers(0, 10))
        if kind < 5:        # widths around the tile boundaries of every mode (256 / 512 / 1024 pixels) and the 16-pixel rules
            w = int(rng.choice([256, 512, 1024, 768, 1280, 2048])) + int(rng.integers(-40, 41))
        elif kind < 8:
            w = int(rng.integers(1, 700))
        else:
            w = 16 * int(rng.integers(2, 160))
        w = max(w, 1)
        h = int(rng.integers(1, 130)) if kind != 9 else 8 * vs * int(rng.integers(1, 12))
        out_cs = outs[int(rng.integers(0, 4))]
        flags = int(rng.integers(0, 8)) if rng.random() < 0.4 else 0
        layout = 1 if (out_cs == oc.RGB and rng.random() < 0.15) else 0
        if out_cs == oc.RGBA or layout == 1:
            flags |= 0  # (RGBA / CHW place every pixel at its own position whatever the flags say)
        variant = int(rng.choice(zj.variants_available()))  # (the product build: 0 and 2; make VARIANTS=all adds 1)
        adversarial = rng.random() < 0.3
        path = ["batch", "frames", "frames_device", "strided", "multi"][int(rng.integers(0, 5))]
        nframes = int(rng.integers(1, 4)) if w * h < 200_000 else 1
        if path in ("frames_device", "multi") and w * h < 20_000 and rng.random() < 0.2:
            nframes = int(rng.integers(33, 40))  # more frames than one launch's pointer table holds
        mk = synth.make_adversarial_frame if adversarial else synth.make_frame
        frames = [mk(w, h, hs, vs, 3, seed=int(rng.integers(0, 1 << 30)), frame_index=i) for i in range(nframes)]
        qts = frames[0][1]
        ext = flags | (oc.EXT_PLAIN if (out_cs == oc.RGBA or layout == 1) else 0)
        exp, rc_all = [], 0
        for f in frames:
            rc, e = oc.decode_planes(oc.make_frame(w, h, hs, vs, 3, out_cs, qts), f[0], ext=ext)
            rc_all |= rc != 0
            if layout == 1 and rc == 0:
                e = np.ascontiguousarray(e.reshape(h, w, 3).transpose(2, 0, 1)).reshape(-1)
            exp.append(e)
        # device outputs: half of them with their rows at a pitch of the caller's choosing (zj_frame_desc.out_pitch)
        ncomp = {oc.RGB: 3, oc.GRAYSCALE: 1, oc.YCBCR: 3, oc.RGBA: 4}[out_cs]
        row = w if layout == 1 else w * ncomp
        pitch = 0
        if path in ("frames_device", "strided") and rng.random() < 0.5:
            fast = w % 16 == 0 and w >= 32
            pitch = [(row + 127) // 128 * 128, row + (16 * int(rng.integers(0, 12)) if fast else int(rng.integers(0, 200)))][int(rng.integers(0, 2))]
        d = zj.FrameDesc.make(w, h, hs, vs, 3, out_cs, qts, flags=flags, out_layout=layout, out_pitch=pitch)
        ctx.set_variant(variant)
        planes = [np.concatenate([f[0][c] for f in frames]) for c in range(3)]
        key = f"{path} {hs}x{vs} out={out_cs} flags={flags} layout={layout} variant={variant} adv={int(adversarial)} pitch={pitch}"
        try:
            got = decode(zj, ctx, multi, path, d, frames, planes, nframes, rng)
            if pitch and not rc_all:   # rows of the padded layout -> the tight bytes; the padding keeps the 0xAA it was given,
                rows = got.reshape(-1, pitch)   # except in rows the strips never reach (zeroed whole, Q6)
                pad = rows[:, row:]
                touched = (pad != 0xAA).any(axis=1)
                if pad.size and ((pad[touched] != 0).any() or rows[touched][:, :row].any()):
                    bad.append((key, w, h, "bytes between the rows were written"))
                    stats["MISMATCH"] += 1
                    continue
                got = np.ascontiguousarray(rows[:, :row]).reshape(-1)
                stats["equal with a padded pitch"] += 0
            if rc_all:
                bad.append((key, w, h, "the oracle panics, the ABI did not"))
                stats["MISMATCH"] += 1
                continue
            want = np.concatenate(exp)
            if not np.array_equal(got, want):
                k = int(np.nonzero(got != want)[0][0])
                bad.append((key, w, h, nframes, f"first differing byte {k} of {want.size}"))
                stats["MISMATCH"] += 1
            else:
                stats["equal"] += 1
                stats[f"equal {hs}x{vs}"] += 1
                stats[f"equal via {path}"] += 1
                stats["equal with a padded pitch"] += 1 if pitch else 0
        except zj.ZjError as e:
            if rc_all and e.status == -5:
                stats["both panic"] += 1
            else:
                bad.append((key, w, h, f"ZjError {e.status}"))
                stats["MISMATCH"] += 1
    multi.close()
    ctx.close()
    print(f"pixel soak: {n} random decodes in {time.time() - t0:.0f} s, seed {a.seed}")
    for k in sorted(stats):
        print(f"  {k:24s} {stats[k]}")
    print("  differences:", len(bad))
    for b in bad[:40]:
        print("   ", b)
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
