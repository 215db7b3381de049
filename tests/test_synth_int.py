"""The integer-only frame generator of round 3 (zune-jpeg_amd/synth.py make_frame_t): pinned to a numpy restatement
written here with unsigned 64-bit arithmetic, to splitmix64's published first output, and (4096x4096) to the golden
luma-plane checksums that tools/make_batch_checksums.py recorded beside the oracle's output checksums."""
import importlib
import json
import os

import numpy as np
import pytest
import torch

synth = importlib.import_module("zune-jpeg_amd.synth")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G, M1, M2 = np.uint64(0x9E3779B97F4A7C15), np.uint64(0xBF58476D1CE4E5B9), np.uint64(0x94D049BB133111EB)


def _mix(x):
    with np.errstate(over="ignore"):
        z = x + G
        z = (z ^ (z >> np.uint64(30))) * M1
        z = (z ^ (z >> np.uint64(27))) * M2
        return z ^ (z >> np.uint64(31))


def _plane_np(key_ac, key_blk, nblocks, q):
    """Loop-free numpy statement of synth._plane_t (no chunking, searchsorted per position, float-free)."""
    bounds, cum_end = synth.ac_thresholds(q)
    with np.errstate(over="ignore"):
        idx = np.arange(nblocks * 64, dtype=np.uint64)
        h = _mix(idx * G + np.uint64(key_ac)).reshape(nblocks, 64)
        hb = _mix(np.arange(nblocks, dtype=np.uint64) * G + np.uint64(key_blk))
    u = (h >> np.uint64(33)).astype(np.int64)
    neg = ((h >> np.uint64(32)) & np.uint64(1)).astype(bool)
    val = np.zeros((nblocks, 64), np.int64)
    for k in range(1, 64):
        tk = np.array(bounds[cum_end[k - 1]:cum_end[k]], dtype=np.int64) - (k << 31)   # ascending thresholds of position k
        val[:, k] = tk.size - np.searchsorted(tk, u[:, k], side="right")               # thresholds above u
    val = np.where(neg, -val, val)
    byte = lambda s: ((hb >> np.uint64(s)) & np.uint64(255)).astype(np.int64)
    walk = np.cumsum((byte(32) + byte(40) + byte(48) + byte(56) - 510) * 83)
    span = 2040 << 10
    dc = np.abs(np.mod(walk + (1024 << 10), 2 * span) - span) - (1024 << 10)
    den = int(q[0]) << 10
    val[:, 0] = np.floor_divide(2 * dc + den, 2 * den)
    val[(hb & np.uint64(0xFFFF)).astype(np.int64) < 22938, 1:] = 0
    val[((hb >> np.uint64(16)) & np.uint64(0xFFFF)).astype(np.int64) < 32768, 21:] = 0
    nat = np.zeros((nblocks, 64), np.int16)
    nat[:, synth.UN_ZIGZAG] = val.astype(np.int16)
    return nat.reshape(-1)


def test_splitmix64_known_answer():
    assert synth.splitmix64_int(0) == 0xE220A8397B1DCDAF     # first output of splitmix64 seeded with 0 (Vigna's reference)
    x = torch.tensor([0, 1, -1, 1 << 62], dtype=torch.int64)
    got = [int(v) & synth._U64 for v in synth._splitmix64_t(x - synth._s64(synth._GOLD) + synth._s64(synth._GOLD)).tolist()]
    assert got == [synth.splitmix64_int(v & synth._U64) for v in (0, 1, -1, 1 << 62)]


def test_thresholds_are_the_laplace_tail():
    q = synth.quant_tables(90)[0]
    bounds, cum_end = synth.ac_thresholds(q)
    assert bounds == sorted(bounds) and len(cum_end) == 64 and cum_end[-1] == len(bounds)
    import math
    for k in (1, 5, 20):
        tk = [b - (k << 31) for b in bounds[cum_end[k - 1]:cum_end[k]]][::-1]          # n = 1, 2, ...
        s, qk = 24.0 * math.exp(-k / 6.0), float(q[synth.UN_ZIGZAG[k]])
        for n, t in enumerate(tk[:6], 1):
            assert abs(t / 2.0 ** 31 - math.exp(-(n - 0.5) * qk / s)) < 1e-8


@pytest.mark.parametrize("hs,vs", [(2, 2), (1, 1), (2, 1), (1, 2)])
def test_torch_generator_equals_numpy_restatement(hs, vs):
    w, h = 200, 72
    planes, qts = synth.make_frame_t(w, h, hs, vs, 3, seed=77, frame_index=5)
    for c in range(3):
        br, bc = synth.plane_blocks(w, h, hs, vs, c)
        exp = _plane_np(synth.frame_key(77, 5, c, 0), synth.frame_key(77, 5, c, 1), br * bc, qts[c])
        assert np.array_equal(planes[c].numpy(), exp)


def test_chunking_does_not_change_the_walk():
    # 70 000 blocks cross the 65 536-block chunk of _plane_t: the DC walk must carry over
    q = synth.quant_tables(90)[0]
    got = synth._plane_t(torch, 11, 12, 70000, q, "cpu").numpy()
    assert np.array_equal(got, _plane_np(11, 12, 70000, q))


def test_frame_statistics_match_the_survey_generator():
    planes, _ = synth.make_frame_t(1024, 1024, 2, 2, 3, seed=1234, frame_index=3)
    old, _ = synth.make_frame(1024, 1024, 2, 2, 3, seed=1234, frame_index=3)
    for c in range(3):
        a, b = planes[c].numpy().reshape(-1, 64), old[c].reshape(-1, 64)
        assert abs((a[:, 1:] == 0).all(1).mean() - 0.35) < 0.02                       # DC-only blocks
        assert abs((a != 0).mean() - (b != 0).mean()) < 0.01                          # non-zero density
        assert np.abs(a[:, 0]).max() <= 342 and abs(np.diff(a[:, 0]).std() - np.diff(b[:, 0]).std()) < 0.3


def test_checksum_torch_equals_numpy():
    rng = np.random.default_rng(3)
    a = rng.integers(0, 256, size=8 * 1000, dtype=np.uint8)
    assert synth.frame_checksum_t(torch.from_numpy(a)) == synth.frame_checksum_sum(a)
    assert synth.frame_checksum_sum(a) != synth.frame_checksum_sum(a[::-1].copy())


def test_golden_checksum_file_pins_the_generator():
    p = os.path.join(ROOT, "tests", "golden", "checksums_seed1234.json")
    g = json.load(open(p))
    assert g["frames"] == 1024 and len(g["rgb"]) == 1024 and len(set(g["rgb"])) == 1024
    planes, _ = synth.make_frame_t(4096, 4096, 2, 2, 3, seed=g["seed"], frame_index=1000)
    assert f"{synth.frame_checksum_sum(planes[0].numpy().view('u1')):016x}" == g["y_plane"][1000]
