"""CPU: round 2's packed IDCT (v_dot2_i32_i16 on 16-bit pairs, zj_device.h idct_block_packed) and the guard
that decides when it may replace the wide transform (classify_block), emulated instruction by instruction
(dot2 / sad_u16 / SDWA shift-pack have literal CPU models in zj_device.h) against the numpy restatement of
src/idct/scalar.rs.  The claim under test: wherever classify_block says 1, idct_block_packed produces the
reference's integers; everything else is routed to idct_block, which is exact for every input."""
import numpy as np
import pytest

import emu_c
import oracle_np as onp

ANNEX_K_LUMA = np.array([
    16, 11, 10, 16, 24, 40, 51, 61, 12, 12, 14, 19, 26, 58, 60, 55, 14, 13, 16, 24, 40, 57, 69, 56,
    14, 17, 22, 29, 51, 87, 80, 62, 18, 22, 37, 56, 68, 109, 103, 77, 24, 35, 55, 64, 81, 104, 113, 92,
    49, 64, 78, 87, 103, 121, 120, 101, 72, 92, 95, 98, 112, 100, 103, 99], np.int32)


def scaled(quality):
    s = 5000 // quality if quality < 50 else 200 - 2 * quality
    return np.clip((ANNEX_K_LUMA * s + 50) // 100, 1, 255).astype(np.int32)


def ref_bytes(coeff, q):
    """scalar.rs full transform (no DC-only shortcut: callers filter those), as bytes"""
    px = onp.idct_blocks(coeff, q).reshape(-1, 64)
    return px


def full_blocks(coeff):
    return np.any(coeff.reshape(-1, 64)[:, 1:] != 0, axis=1)


def test_matrix_entries_fit_16_bits_and_bound():
    """the effective 8x8 matrix of the butterfly (scalar.rs:79-148): entries are what dot2 multiplies by; the guard's
    bound uses max |entry| = 5683"""
    m = np.zeros((8, 8), np.int64)
    for k in range(8):
        e = [np.array([1 if i == k else 0], np.int32) for i in range(8)]
        m[:, k] = [int(v[0]) for v in onp._pass(e, np.int32(0))]
    assert np.abs(m).max() == 5683
    lim = emu_c.guard_limit()
    assert 5683 * lim + 512 < 2 ** 25          # pass-1 results (x >> 10) fit i16
    assert lim <= 32767                         # dequantized coefficients fit i16


@pytest.mark.parametrize("quality", [10, 35, 50, 75, 90, 100])
@pytest.mark.parametrize("amp", [4, 40, 400, 1500, 6000])
def test_packed_equals_reference_wherever_the_guard_passes(quality, amp):
    q = scaled(quality)
    rng = np.random.default_rng(quality * 7 + amp)
    n = 4000
    # dequantized-domain Laplace coefficients of scale amp * exp(-zigzag-ish decay), quantized by q
    decay = np.exp(-np.add.outer(np.arange(8), np.arange(8)).reshape(64) / 3.0)
    deq = rng.laplace(0, 1, (n, 64)) * amp * decay[None, :]
    deq[:, 0] = rng.uniform(-1024, 1016, n)
    coeff = np.clip(np.round(deq / q[None, :]), -32768, 32767).astype(np.int16)
    cls = emu_c.classify(coeff, q)
    full = full_blocks(coeff)
    assert np.array_equal(cls == 0, ~full)
    ok = cls == 1
    exp = ref_bytes(coeff, q)
    got = emu_c.idct_packed(coeff, q)
    assert np.array_equal(got[ok].astype(np.int16), exp[ok]), (quality, amp)
    # the wide transform is exact for every full block, whatever the class
    wide = emu_c.idct_wide(coeff, q)
    assert np.array_equal(wide[full], exp[full])
    if amp <= 40:
        assert ok.sum() == full.sum(), "ordinary data must never leave the packed path"


def test_guard_boundary_single_coefficient():
    """one non-zero AC coefficient of growing size at every position: the class flips exactly where
    |c| * (weight of its row group and column pair) crosses GUARD_LIMIT, and the packed result is exact up to there"""
    lim = emu_c.guard_limit()
    q = scaled(50)
    for pos in range(1, 64):
        k, j = divmod(pos, 8)
        rows = range(0, 4) if k < 4 else range(4, 8)
        wgt = max(int(q[8 * r + 2 * (j // 2) + h]) for r in rows for h in (0, 1))
        cmax = lim // wgt
        vals = np.array([cmax - 1, cmax, cmax + 1, -cmax, -(cmax + 1)], np.int16)
        coeff = np.zeros((len(vals), 64), np.int16)
        coeff[:, pos] = vals
        cls = emu_c.classify(coeff, q)
        assert list(cls) == [1, 1, 2, 1, 2], (pos, cmax, list(cls))
        exp = ref_bytes(coeff, q)
        got = emu_c.idct_packed(coeff, q)
        assert np.array_equal(got[cls == 1].astype(np.int16), exp[cls == 1])


def test_guard_worst_case_columns():
    """all of a column's budget on the row whose matrix entry is the largest (5683, row 1), with both signs, for
    every column and flat tables: pass-1 results reach +-(2^25 - small) and must still be exact"""
    lim = emu_c.guard_limit()
    for qv in (1, 2, 3, 16, 255):
        q = np.full(64, qv, np.int32)
        c = lim // qv
        blocks = []
        for j in range(8):
            for sgn in (1, -1):
                b = np.zeros(64, np.int16)
                b[8 * 1 + j] = sgn * c
                blocks.append(b)
                b = np.zeros(64, np.int16)        # the same budget spread over the column
                for k in range(8):
                    b[8 * k + j] = sgn * (c // 8) * (1 if k % 2 else -1)
                blocks.append(b)
        coeff = np.array(blocks, np.int16)
        cls = emu_c.classify(coeff, q)
        assert np.all(cls == 1)
        assert np.array_equal(emu_c.idct_packed(coeff, q).astype(np.int16), ref_bytes(coeff, q))


def test_extreme_values_are_routed_wide():
    """i16 extremes (|-32768| is what v_sad_u16 against 0x8000 must get right), full-range noise, q = 255"""
    q = np.full(64, 255, np.int32)
    rng = np.random.default_rng(5)
    coeff = rng.integers(-32768, 32768, (2000, 64)).astype(np.int16)
    coeff[0, :] = -32768
    coeff[1, :] = 32767
    coeff[2, :] = 0
    coeff[2, 63] = -32768
    cls = emu_c.classify(coeff, q)
    assert np.all(cls == 2)
    exp = ref_bytes(coeff, q)
    assert np.array_equal(emu_c.idct_wide(coeff, q), exp)
    # and the guard is not decoration: the packed transform is wrong on (most of) these
    got = emu_c.idct_packed(coeff, q)
    assert np.any(got.astype(np.int16) != exp)


def test_zero_table_entries():
    """a table may hold zeros (8-bit DQT allows them): weights of 0 must not upset the guard"""
    q = scaled(50).copy()
    q[5:] = 0
    rng = np.random.default_rng(9)
    coeff = rng.integers(-40, 40, (500, 64)).astype(np.int16)
    cls = emu_c.classify(coeff, q)
    ok = cls == 1
    assert ok.any() and not ok.all()
    assert np.array_equal(emu_c.idct_packed(coeff, q)[ok].astype(np.int16), ref_bytes(coeff, q)[ok])
