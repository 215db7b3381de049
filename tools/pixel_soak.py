#!/usr/bin/env python3
"""
Soak of the PIXEL path (GPU box): random geometries, sampling modes, output colourspaces, extension flags, layouts, kernel
variants and coefficient statistics through the C ABI (zj_decode_planes[_batch]) against the oracle (oracle/zj_oracle.c),
byte for byte, for --seconds seconds.  Where the reference panics, the ABI must report ZJ_ERR_PANIC.  Checker
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
        variant = int(rng.integers(0, 3))
        adversarial = rng.random() < 0.3
        nframes = int(rng.integers(1, 4)) if w * h < 200_000 else 1
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
        d = zj.FrameDesc.make(w, h, hs, vs, 3, out_cs, qts, flags=flags, out_layout=layout)
        ctx.set_variant(variant)
        planes = [np.concatenate([f[0][c] for f in frames]) for c in range(3)]
        key = f"{hs}x{vs} out={out_cs} flags={flags} layout={layout} variant={variant} adv={int(adversarial)}"
        try:
            got = ctx.decode_planes(d, planes, nframes)
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
        except zj.ZjError as e:
            if rc_all and e.status == -5:
                stats["both panic"] += 1
            else:
                bad.append((key, w, h, f"ZjError {e.status}"))
                stats["MISMATCH"] += 1
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
