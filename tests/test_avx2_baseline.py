"""CPU: the restated AVX2 arms (oracle/zj_avx2.c, the timed CPU baseline).  Not a parity target: its
IDCT runs the row pass first and clamps the DC-only value, so it is checked for the
properties the reference's own tests assert (KATs equal to scalar) and for |diff| <= 1 otherwise."""
import json
import os

import numpy as np
import pytest

import avx2_c
import oracle_c as oc

HERE = os.path.dirname(os.path.abspath(__file__))
KAT = json.load(open(os.path.join(HERE, "golden", "idct_kat.json")))


@pytest.mark.parametrize("name", ["zeroes", "max", "min"])
def test_avx2_idct_kats_equal_scalar(name):
    """src/idct.rs:66-127 assert scalar == AVX2 on exactly these inputs"""
    coeff = np.full(64, KAT[name]["coeff"], np.int16)
    rc, out = avx2_c.idct_strip(coeff, np.ones(64, np.int32), 8, 1, 1)
    assert rc == 0 and np.array_equal(out, np.array(KAT[name]["expected"], np.int16))


def test_avx2_idct_within_one_of_scalar():
    rng = np.random.default_rng(5)
    n = 20000
    b = rng.integers(-200, 201, size=(n, 64))
    b[rng.random((n, 64)) < 0.7] = 0
    b[:, 0] = rng.integers(-900, 900, size=n)
    b = b.astype(np.int16).reshape(-1)
    qt = rng.integers(1, 40, size=64).astype(np.int32)
    rc, a = avx2_c.idct_strip(b, qt, 8 * n, 1, 1)
    rc2, s = oc.idct_strip(b, qt, 8 * n, 1, 1)
    assert rc == 0 and rc2 == 0
    d = np.abs(a.astype(np.int32) - np.clip(s, 0, 255))
    assert d.max() <= 1 and (d != 0).any()  # row-first order really differs, by at most one level


def test_avx2_upsample_hv_equals_scalar():
    rng = np.random.default_rng(6)
    inp = rng.integers(-100, 356, size=16 * 2048).astype(np.int16)
    rc, out = avx2_c.upsample_hv(inp, 4 * inp.size)
    rc2, exp = oc.upsample_hv(inp, 4 * inp.size)
    assert rc == 0 and rc2 == 0 and np.array_equal(out, exp)


@pytest.mark.parametrize("mode", [(1, 1), (2, 2)])
def test_avx2_frame_close_to_scalar(mode, synth):
    hs, vs = mode
    w, h = 512, 128
    planes, qts = synth.make_frame(w, h, hs, vs, 3, seed=12)
    f = oc.make_frame(w, h, hs, vs, 3, oc.RGB, qts)
    rc, exp = oc.decode_planes(f, planes)
    outs = [avx2_c.decode_planes_mt(f, planes, 1, t) for t in (1, 4)]
    assert rc == 0 and all(r == 0 for r, _ in outs)
    assert np.array_equal(outs[0][1], outs[1][1])  # thread count does not change the result
    d = np.abs(outs[0][1].astype(np.int32) - exp.astype(np.int32))
    assert d.max() <= 4 and np.mean(d != 0) < 0.5
    assert not outs[0][1].reshape(h, 3 * w)[:, -16:].any()  # same Q5/Q6 tail as the scalar worker
