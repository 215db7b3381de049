"""CPU tests of the oracle itself: reference KATs (src/idct.rs:66-127), C restatement vs the
independent numpy restatement, and the hand-derived quirk register of SURVEY.md 8a."""
import json
import os

import numpy as np
import pytest

import oracle_c as oc
import oracle_np as onp

HERE = os.path.dirname(os.path.abspath(__file__))
KAT = json.load(open(os.path.join(HERE, "golden", "idct_kat.json")))
ONES = np.ones(64, np.int32)


@pytest.mark.parametrize("name", ["zeroes", "max", "min"])
def test_reference_idct_kat(name):
    coeff = np.full(64, KAT[name]["coeff"], np.int16)
    exp = np.array(KAT[name]["expected"], np.int16)
    rc, out = oc.idct_strip(coeff, ONES, 8, 1, 1)
    assert rc == 0
    assert np.array_equal(out, exp)
    assert np.array_equal(onp.idct_strip(coeff, ONES, 8, 1, 1), exp)


def _rand_blocks(rng, n, kind):
    if kind == "full":
        b = rng.integers(-32768, 32768, size=(n, 64))
    elif kind == "small":
        b = rng.integers(-64, 65, size=(n, 64))
        b[rng.random((n, 64)) < 0.7] = 0
    else:  # dc-only incl. wrap
        b = np.zeros((n, 64), np.int64)
        b[:, 0] = rng.integers(-32768, 32768, size=n)
    return b.astype(np.int16)


@pytest.mark.parametrize("kind", ["full", "small", "dc"])
def test_idct_c_vs_numpy_blocks(kind):
    rng = np.random.default_rng(7)
    n = 4096
    blocks = _rand_blocks(rng, n, kind)
    qt = rng.integers(1, 256, size=64).astype(np.int32)
    rc, out = oc.idct_strip(blocks.reshape(-1), qt, 8 * n, 1, 1)
    assert rc == 0
    exp = onp.idct_strip(blocks.reshape(-1), qt, 8 * n, 1, 1)
    assert np.array_equal(out, exp)
    if kind == "dc":  # Q1: unclamped, floor, i16 wrap
        v = ((blocks[:, 0].astype(np.int64) * qt[0] + 32768) % 65536 - 32768 >> 3) + 128
        assert np.array_equal(out.reshape(8, n, 8)[0, :, 0], v.astype(np.int16))
        assert out.min() < 0 and out.max() > 255


def test_idct_strip_layout_hv():
    """Y: (stride 4*64, samp 4, v_samp 1) -> 4 chunks; C: (stride .., samp 4, v_samp 2) -> 2 chunks."""
    rng = np.random.default_rng(3)
    mcu_x = 4
    qt = rng.integers(1, 32, size=64).astype(np.int32)
    y = _rand_blocks(rng, 4 * 2 * mcu_x, "small").reshape(-1)
    c = _rand_blocks(rng, 2 * mcu_x, "small").reshape(-1)
    for coeff, stride, vs in ((y, 16 * mcu_x, 1), (c, 8 * mcu_x, 2)):
        rc, out = oc.idct_strip(coeff, qt, stride, 4, vs)
        assert rc == 0
        assert np.array_equal(out, onp.idct_strip(coeff, qt, stride, 4, vs))
        # raster check: block (br, bc) occupies rows 8br.., cols 8bc..
        nbc = stride // 8
        px = onp.idct_blocks(coeff.reshape(-1, 64), qt).reshape(-1, nbc, 8, 8)
        ras = out.reshape(-1, stride)
        for br in range(px.shape[0]):
            for bc in range(nbc):
                assert np.array_equal(ras[8 * br:8 * br + 8, 8 * bc:8 * bc + 8], px[br, bc])


def test_upsample_horizontal_quirks():
    inp = np.array([10, 20, 30, 40, 50, 60, 70, 80], np.int16)
    rc, out = oc.upsample_h(inp, 16)
    assert rc == 0
    assert np.array_equal(out, onp.upsample_horizontal(inp, 16))
    assert out[0] == 10 and out[15] == 80
    assert out[13] == out[14] == (3 * 70 + 80 + 2) >> 2  # Q4: swapped weights on out[2n-2]
    assert out[2] == (3 * 20 + 10 + 2) >> 2 and out[3] == (3 * 20 + 30 + 2) >> 2
    assert oc.upsample_h(inp[:2], 16)[0] == oc.ERR_PANIC
    with pytest.raises(onp.Panic):
        onp.upsample_horizontal(inp[:2], 16)


def test_upsample_vertical_schedule():
    assert onp._vertical_schedule(8) == [(0, 0), (1, 2), (2, 3), (3, 4), (4, 5), (5, 6), (6, 7), (7, 7)]
    rng = np.random.default_rng(5)
    w = 24
    inp = rng.integers(-4000, 4300, size=8 * w).astype(np.int16)
    rc, out = oc.upsample_v(inp, 16 * w)
    assert rc == 0
    assert np.array_equal(out, onp.upsample_vertical(inp, 16 * w))
    rows, o = inp.reshape(8, w).astype(np.int32), out.reshape(16, w)
    assert np.array_equal(o[0], rows[0]) and np.array_equal(o[1], rows[0])
    assert np.array_equal(o[2], (3 * rows[1] + rows[2] + 2) >> 2)
    assert np.array_equal(o[3], (3 * rows[2] + rows[1] + 2) >> 2)
    assert np.array_equal(o[14], rows[7]) and np.array_equal(o[15], rows[7])


@pytest.mark.parametrize("n,w", [(16 * 8, 8), (16 * 40, 40), (16 * 2048, 2048)])
def test_upsample_hv_c_vs_numpy_and_q3(n, w):
    rng = np.random.default_rng(11)
    inp = rng.integers(-3968, 4224, size=n).astype(np.int16)
    rc, out = oc.upsample_hv(inp, 4 * n)
    assert rc == 0
    assert np.array_equal(out, onp.upsample_hv(inp, 4 * n))
    # Q3: vertical pass sees 8 "rows" of 2 real rows each
    C_ = inp.reshape(16, w).astype(np.int32)
    sched = [(0, 0), (1, 2), (2, 3), (3, 4), (4, 5), (5, 6), (6, 7), (7, 7)]
    V = np.zeros((32, w), np.int32)
    for m in range(32):
        k, half, farw = m // 4, m % 2, (m % 4) >= 2
        nr, fr = sched[k]
        if farw:
            nr, fr = fr, nr
        V[m] = (3 * C_[2 * nr + half] + C_[2 * fr + half] + 2) >> 2
    v = V.reshape(-1)
    exp = np.zeros(4 * n, np.int32)
    i = np.arange(1, 2 * n - 1)
    exp[2 * i] = (3 * v[i] + v[i - 1] + 2) >> 2       # Q4: flat, rows bleed into each other
    exp[2 * i + 1] = (3 * v[i] + v[i + 1] + 2) >> 2
    exp[0], exp[1] = v[0], (3 * v[0] + v[1] + 2) >> 2
    exp[4 * n - 2], exp[4 * n - 1] = (3 * v[-2] + v[-1] + 2) >> 2, v[-1]
    assert np.array_equal(out.astype(np.int32), exp)


def test_rgb16_wrapping_and_panic():
    rng = np.random.default_rng(13)
    for _ in range(50):
        y, cb, cr = (rng.integers(-4100, 4400, size=16).astype(np.int16) for _ in range(3))
        out = np.zeros(100, np.uint8)
        rc, pos = oc.ycbcr_to_rgb16(y, cb, cr, out, 4)
        assert rc == 0 and pos == 52
        exp = np.zeros(100, np.uint8)
        assert onp.ycbcr_to_rgb_16(y, cb, cr, exp, 4) == 52
        assert np.array_equal(out, exp)
    out = np.zeros(60, np.uint8)
    assert oc.ycbcr_to_rgb16(y, cb, cr, out, 13)[0] == oc.ERR_PANIC  # "Slice to small cannot write"
    # wrap really happens: 45 * (4223-128) overflows i16
    y = np.zeros(16, np.int16); cb = np.full(16, 128, np.int16); cr = np.full(16, 4223, np.int16)
    out = np.zeros(48, np.uint8)
    oc.ycbcr_to_rgb16(y, cb, cr, out, 0)
    r = (((45 * (4223 - 128) + 32768) % 65536) - 32768) >> 5
    assert out[0] == min(max(r, 0), 255)


MODES = {"none": (1, 1), "h": (2, 1), "v": (1, 2), "hv": (2, 2)}


@pytest.mark.parametrize("mode", list(MODES))
@pytest.mark.parametrize("out_cs", [oc.RGB, oc.GRAYSCALE, oc.YCBCR])
@pytest.mark.parametrize("wh", [(64, 64), (200, 72), (37, 50), (16, 16), (100, 33)])
def test_decode_planes_c_vs_numpy(mode, out_cs, wh, synth):
    h, v = MODES[mode]
    w, hh = wh
    for adversarial in (False, True):
        mk = synth.make_adversarial_frame if adversarial else synth.make_frame
        planes, qts = mk(w, hh, h, v, 3, seed=21)
        f = oc.make_frame(w, hh, h, v, 3, out_cs, qts)
        for c in range(3):
            assert oc.plane_len(f, c) == planes[c].size == onp.plane_len(w, hh, h, v, c)
        rc, out = oc.decode_planes(f, planes)
        try:
            exp = onp.decode_planes(w, hh, h, v, 3, out_cs, qts, planes)
        except onp.Panic:
            assert rc == oc.ERR_PANIC
            continue
        assert rc == 0, (mode, out_cs, wh)
        assert np.array_equal(out, exp)


def test_decode_planes_gray_input(synth):
    planes, qts = synth.make_frame(120, 40, 1, 1, 1, seed=4)
    f = oc.make_frame(120, 40, 1, 1, 1, oc.GRAYSCALE, qts)
    rc, out = oc.decode_planes(f, planes)
    assert rc == 0
    assert np.array_equal(out, onp.decode_planes(120, 40, 1, 1, 1, onp.GRAYSCALE, qts, planes))


def test_rgb_tail_quirk_q5(synth):
    """W multiple of 16: last 16 px land 16 bytes early; the final 16 bytes of each row stay 0."""
    w, hh = 64, 16
    planes, qts = synth.make_frame(w, hh, 1, 1, 3, seed=8)
    f = oc.make_frame(w, hh, 1, 1, 3, oc.RGB, qts)
    rc, out = oc.decode_planes(f, planes)
    assert rc == 0
    rows = out.reshape(hh, 3 * w)
    assert not rows[:, 3 * w - 16:].any()
    # build the "sane" image and compare the shifted tail
    y, cb, cr = (onp.idct_strip(planes[c][:w * 8], qts[c], w, 1, 1).reshape(8, w) for c in range(3))
    sane = onp.ycbcr_to_rgb_px(y, cb, cr).reshape(8, 3 * w)
    assert np.array_equal(rows[:8, :3 * w - 64], sane[:, :3 * w - 64])
    assert np.array_equal(rows[:8, 3 * w - 64:3 * w - 16], sane[:, 3 * w - 48:])
