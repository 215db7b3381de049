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
    d = np.abs(outs[0][1].astype(np.int32) - exp.astype(np.int32)).reshape(h, w, 3)
    if mode == (2, 2):
        # the reference's AVX2 h2v2 arm (upsample_hv_avx, restated literally) mis-weights the neighbour taps it carries
        # between its 16-sample iterations and filters its row tails differently (test_reference_hv_avx_arm below):
        # pixel columns 0, 32k - 1, 32k and the last 48 (32 + the Q5 shift) are its own
        keep = np.ones(w, bool)
        keep[0] = False
        keep[31::32] = False
        keep[32::32] = False
        keep[-48:] = False
        d = d[:, keep]
    assert d.max() <= 4 and np.mean(d != 0) < 0.5
    assert not outs[0][1].reshape(h, 3 * w)[:, -16:].any()  # same Q5/Q6 tail as the scalar worker


# ---- the reference's own relational tests, replayed (the only vectors it holds beyond the IDCT KATs) ----

@pytest.mark.parametrize("ramp", [np.arange(0, 128, dtype=np.int16), np.arange(1279, -1, -1, dtype=np.int16)],
                         ids=["upsample_sse_v1: 0..128", "upsample_sse_v2: (0..1280).rev()"])
def test_reference_upsample_sse_equals_scalar_on_its_ramps(ramp):
    """src/upsampler.rs:126-151: upsample_horizontal_sse(v, 2 len) == scalar::upsample_horizontal(v, 2 len) on exactly
    these two inputs.  Both sides are restatements (oracle/zj_avx2.c zja_upsample_h_sse <- src/upsampler/sse.rs:24-134,
    oracle/zj_oracle.c zjo_upsample_h <- src/upsampler/scalar.rs:5-60): the reference's assertion pins them to each other."""
    rc, sse = avx2_c.upsample_h_sse(ramp, 2 * ramp.size)
    rc2, sc = oc.upsample_h(ramp, 2 * ramp.size)
    assert rc == 0 and rc2 == 0
    assert np.array_equal(sse, sc)
    # ... and to the numpy restatement, and to the closed form of a unit-step ramp's triangle filter
    import oracle_np as onp
    assert np.array_equal(onp.upsample_horizontal(ramp, 2 * ramp.size), sc)
    step = int(ramp[1]) - int(ramp[0])
    i = np.arange(1, ramp.size - 1)
    assert np.array_equal(sc[2 * i].astype(np.int64), (4 * ramp[i].astype(np.int64) - step + 2) >> 2)
    assert np.array_equal(sc[2 * i + 1].astype(np.int64), (4 * ramp[i].astype(np.int64) + step + 2) >> 2)


def test_reference_upsample_sse_differs_from_scalar_only_in_three_of_its_last_eight_outputs():
    """What the ramps cannot show: sse.rs:113-131 writes the last eight outputs with other taps than the scalar arm at
    positions 2n-5, 2n-4, 2n-3 (oracle/zj_avx2.c header); everywhere else the two arms agree on arbitrary data."""
    rng = np.random.default_rng(8)
    seen = set()
    for n in (8, 12, 64, 1280, 4096):
        for _ in range(20):
            v = rng.integers(-300, 600, size=n).astype(np.int16)
            rc, sse = avx2_c.upsample_h_sse(v, 2 * n)
            rc2, sc = oc.upsample_h(v, 2 * n)
            assert rc == 0 and rc2 == 0
            bad = np.nonzero(sse != sc)[0]
            assert set(bad - 2 * n) <= {-5, -4, -3}
            seen |= set(bad - 2 * n)
            x = v.astype(np.int64)
            assert sse[2 * n - 5] == (4 * x[n - 3] + 2) >> 2 and sse[2 * n - 4] == (4 * x[n - 2] + 2) >> 2
            assert sse[2 * n - 3] == (3 * x[n - 2] + x[n - 3] + 2) >> 2
    assert seen == {-5, -4, -3}
    # the reference's asserts (sse.rs:33): out.len() > 8 and input.len() > 5
    assert avx2_c.upsample_h_sse(np.zeros(5, np.int16), 10)[0] == oc.ERR_PANIC
    assert avx2_c.upsample_h_sse(np.zeros(8, np.int16), 8)[0] == oc.ERR_PANIC


def test_restated_avx2_colour_equals_scalar_colour():
    """ycbcr_to_rgb_avx2 (src/color_convert/avx.rs:81-192) and ycbcr_to_rgb_16_scalar
    (src/color_convert/scalar.rs:52-89) are the same arithmetic -- sub 128, i16 wrapping products, arithmetic shifts,
    clamp -- so the two restatements must agree on EVERY i16 triple: 10^6 random ones over the full i16 range, 10^6 over
    the range the IDCT can emit incl. the unclamped DC-only values (Q1: -3968 ... 4223), and the corners."""
    rng = np.random.default_rng(9)
    edge = np.array([-32768, -32767, -4097, -4096, -3968, -129, -128, -1, 0, 1, 127, 128, 255, 256, 4223, 4224, 32767], np.int16)
    sets = [rng.integers(-32768, 32768, size=(3, 1_000_000 // 16 * 16)).astype(np.int16),
            rng.integers(-3968, 4224, size=(3, 1_000_000 // 16 * 16)).astype(np.int16),
            np.stack(np.meshgrid(edge, edge, edge, indexing="ij")).reshape(3, -1)[:, : edge.size ** 3 // 16 * 16].astype(np.int16)]
    for ycc in sets:
        n = ycc.shape[1]
        a, s = np.zeros(3 * n, np.uint8), np.zeros(3 * n, np.uint8)
        pa = ps = 0
        for g in range(0, n, 16):
            rc, pa = avx2_c.ycbcr_to_rgb16(ycc[0, g:g + 16], ycc[1, g:g + 16], ycc[2, g:g + 16], a, pa)
            assert rc == 0
        # the scalar arm over the same groups in one C call per 4096 groups would need a loop in C; the wrapper is cheap enough
        for g in range(0, n, 16):
            rc, ps = oc.ycbcr_to_rgb16(ycc[0, g:g + 16], ycc[1, g:g + 16], ycc[2, g:g + 16], s, ps)
            assert rc == 0
        assert pa == ps == 3 * n
        assert np.array_equal(a, s)
    # same panic rule: 48 bytes must fit behind *pos (avx.rs:91 / scalar.rs:60)
    small = np.zeros(47, np.uint8)
    z = np.zeros(16, np.int16)
    assert avx2_c.ycbcr_to_rgb16(z, z, z, small, 0)[0] == oc.ERR_PANIC and oc.ycbcr_to_rgb16(z, z, z, small, 0)[0] == oc.ERR_PANIC


def test_reference_hv_avx_arm_equals_scalar_in_the_interior_only():
    """upsample_hv_avx (src/upsampler/avx2.rs:29-342), restated statement by statement for the timed baseline
    (oracle/zj_avx2.c zja_upsample_hv_avx).  Not the scalar arm's function: the neighbours it carries from one 16-sample
    iteration to the next are 3 * (a + b + 2) >> 2 (method-call precedence, :261-267), its row tails filter the raw input
    rows, five outputs per row pair are copies.  So: equal to upsample_hv wherever neither applies -- every output column
    except 0, 32k - 1, 32k and the last 32 -- and different there on arbitrary data."""
    rng = np.random.default_rng(10)
    for W in (256, 512, 4096):
        inp = rng.integers(0, 256, size=8 * (W // 2)).astype(np.int16)
        rc, a = avx2_c.upsample_hv_avx(inp, 16 * W)
        rc2, s = oc.upsample_hv(inp, 16 * W)
        assert rc == 0 and rc2 == 0
        own = np.zeros(W, bool)
        own[0] = True
        own[31::32] = True
        own[32::32] = True
        own[-32:] = True
        diff = (a != s).reshape(16, W)
        assert not diff[:, ~own].any()
        assert diff[:, own].any(axis=0).sum() > own.sum() // 2      # ... and it really is another function there
    # upsample_hv_simd (:15-23): fewer than 500 input samples take the scalar arm
    small = rng.integers(0, 256, size=8 * 32).astype(np.int16)
    rc, a = avx2_c.upsample_hv_avx(small, 16 * 64, simd_entry=True)
    rc2, s = oc.upsample_hv(small, 16 * 64)
    assert rc == 0 and rc2 == 0 and np.array_equal(a, s)
    # slices the Rust would panic on
    assert avx2_c.upsample_hv_avx(np.zeros(16, np.int16), 64)[0] == oc.ERR_PANIC
    assert avx2_c.upsample_hv_avx(np.zeros(1024, np.int16), 1024)[0] == oc.ERR_PANIC    # output too short for the stores
