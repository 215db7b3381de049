"""CPU: the HIP workgroup phases (zune-jpeg_amd/csrc/zj_device.h), emulated thread by thread, against
the oracle.  Checks tile / halo / wrap-around / tail indexing and the 24-bit-multiplier exactness
claim (the emulated v_mul_i32_i24 truncates operands to 24 bits like the hardware)."""
import os

import numpy as np
import pytest

import emu_c
import oracle_c as oc

MODES = {"none": (1, 1), "h": (2, 1), "v": (1, 2), "hv": (2, 2)}


@pytest.fixture(autouse=True)
def _default_variant():
    yield
    emu_c.set_variant(0)


@pytest.mark.parametrize("mode", list(MODES))
@pytest.mark.parametrize("out_cs", [oc.RGB, oc.GRAYSCALE, oc.YCBCR])
@pytest.mark.parametrize("wh", [(64, 64), (528, 40), (32, 8), (272, 100), (1040, 33), (2080, 16)])
@pytest.mark.parametrize("compact", [0, 1, 2])
def test_emulated_kernel_matches_oracle(mode, out_cs, wh, synth, compact):
    emu_c.set_variant(compact)
    hs, vs = MODES[mode]
    w, h = wh
    for adversarial in (False, True):
        mk = synth.make_adversarial_frame if adversarial else synth.make_frame
        planes, qts = mk(w, h, hs, vs, 3, seed=31)
        f = oc.make_frame(w, h, hs, vs, 3, out_cs, qts)
        rc, exp = oc.decode_planes(f, planes)
        assert rc == 0
        rc, out = emu_c.decode_planes(f, planes)
        assert rc == 0
        bad = np.nonzero(out != exp)[0]
        assert bad.size == 0, (mode, out_cs, wh, adversarial, bad[:8])


def test_emulated_batch_and_untouched_bytes(synth):
    """nframes > 1 and zero_fill = 0 (strip-level contract: never-written bytes stay untouched)."""
    w, h = 64, 32
    frames = [synth.make_frame(w, h, 2, 2, 3, seed=5, frame_index=i) for i in range(3)]
    qts = frames[0][1]
    planes = [np.concatenate([fr[0][c] for fr in frames]) for c in range(3)]
    f = oc.make_frame(w, h, 2, 2, 3, oc.RGB, qts)
    rc, out = emu_c.decode_planes(f, planes, nframes=3, zero_fill=0, poison=0x5C)
    assert rc == 0
    for i, fr in enumerate(frames):
        rc, exp = oc.decode_planes(f, fr[0])
        got = out[i * exp.size:(i + 1) * exp.size].reshape(h, 3 * w)
        e = exp.reshape(h, 3 * w)
        assert np.array_equal(got[:, :3 * w - 16], e[:, :3 * w - 16])
        assert (got[:, 3 * w - 16:] == 0x5C).all()  # Q6 bytes untouched


RAGGED = [(100, 32), (37, 50), (2500, 24), (17, 16), (200, 72), (1000, 40), (24, 24), (8, 8), (5, 3), (16, 16), (1001, 33)]


@pytest.mark.parametrize("mode", list(MODES))
@pytest.mark.parametrize("out_cs", [oc.RGB, oc.GRAYSCALE, oc.YCBCR])
@pytest.mark.parametrize("wh", RAGGED)
def test_emulated_kernel_ragged_widths(mode, out_cs, wh, synth):
    """Any width: padded rows, the RGB tail at odd offsets, P % 16 == 8, width < 16 (worker.rs:143-251)."""
    hs, vs = MODES[mode]
    w, h = wh
    for adversarial in (False, True):
        mk = synth.make_adversarial_frame if adversarial else synth.make_frame
        planes, qts = mk(w, h, hs, vs, 3, seed=77)
        f = oc.make_frame(w, h, hs, vs, 3, out_cs, qts)
        rc, exp = oc.decode_planes(f, planes)
        rce, out = emu_c.decode_planes(f, planes)
        if rc != 0:
            assert rce == -5  # the reference panics on this geometry -> ZJ_ERR_PANIC
            continue
        assert rce == 0
        bad = np.nonzero(out != exp)[0]
        assert bad.size == 0, (mode, out_cs, wh, adversarial, bad[:8])


def _plain_expected(w, h, hs, vs, qts, planes, kind):
    """Expected bytes of the extensions from the oracle's plain-placement restatement (zjo_decode_planes_plain)."""
    if kind == "rgba":
        rc, exp = oc.decode_planes(oc.make_frame(w, h, hs, vs, 3, oc.RGBA, qts), planes, plain=True)
        return rc, exp
    rc, rgb = oc.decode_planes(oc.make_frame(w, h, hs, vs, 3, oc.RGB, qts), planes, plain=True)
    if kind == "chw":
        rgb = np.ascontiguousarray(rgb.reshape(h, w, 3).transpose(2, 0, 1)).reshape(-1)
    return rc, rgb


@pytest.mark.parametrize("mode", list(MODES))
@pytest.mark.parametrize("kind", ["plain", "rgba", "chw"])
@pytest.mark.parametrize("wh", [(64, 64), (528, 40), (32, 8), (1040, 33), (100, 32), (37, 50), (17, 16), (5, 3), (250, 72)])
def test_emulated_extensions_match_plain_oracle(mode, kind, wh, synth):
    """ZJ_FLAG_PLAIN_TAIL, ZJ_CS_RGBA and ZJ_LAYOUT_CHW (include/zjhip.h): same strips and arithmetic as the reference,
    every pixel at its own position; odd MCU rows (72 = 4.5 MCU rows in H/HV) still stay zero."""
    emu_c.set_variant(0)
    hs, vs = MODES[mode]
    w, h = wh
    for adversarial in (False, True):
        mk = synth.make_adversarial_frame if adversarial else synth.make_frame
        planes, qts = mk(w, h, hs, vs, 3, seed=77)
        rc, exp = _plain_expected(w, h, hs, vs, qts, planes, kind)
        assert rc == 0
        f = oc.make_frame(w, h, hs, vs, 3, oc.RGBA if kind == "rgba" else oc.RGB, qts)
        rc, out = emu_c.decode_planes(f, planes, flags=1 if kind == "plain" else 0, out_layout=1 if kind == "chw" else 0)
        assert rc == 0
        bad = np.nonzero(out != exp)[0]
        assert bad.size == 0, (mode, kind, wh, adversarial, bad[:8])


def test_plain_oracle_cross_check_numpy(synth):
    """the two restatements agree on the extension too, and plain differs from the reference bytes only in the tail"""
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
    import oracle_np as onp
    for (w, h, hs, vs) in [(64, 48, 2, 2), (100, 37, 2, 1), (17, 9, 1, 1), (250, 40, 1, 2), (8, 8, 2, 2)]:
        planes, qts = synth.make_frame(w, h, hs, vs, 3, seed=5)
        for cs in (oc.RGB, oc.RGBA):
            rc, a = oc.decode_planes(oc.make_frame(w, h, hs, vs, 3, cs, qts), planes, plain=True)
            assert rc == 0
            assert np.array_equal(a, onp.decode_planes(w, h, hs, vs, 3, cs, qts, planes, plain=True))
    planes, qts = synth.make_frame(64, 48, 2, 2, 3, seed=5)
    f = oc.make_frame(64, 48, 2, 2, 3, oc.RGB, qts)
    q = oc.decode_planes(f, planes)[1].reshape(48, 192)
    p = oc.decode_planes(f, planes, plain=True)[1].reshape(48, 192)
    assert np.array_equal(q[:, :128], p[:, :128]) and np.array_equal(q[:, 128:176], p[:, 144:]) and not q[:, 176:].any()


def test_random_geometry_sweep_all_output_kinds(synth):
    """400 random (width, height, mode, output) cases, widths 1..700 and heights 1..90: emulated kernel == oracle
    (reference bytes for RGB / GRAYSCALE / YCbCr, plain-placement restatement for the extensions); where the oracle
    reports a reference panic the plan must report ZJ_ERR_PANIC (-5)."""
    emu_c.set_variant(0)
    rng = np.random.default_rng(20261001)
    kinds = ["rgb", "gray", "ycbcr", "plain", "rgba", "chw"]
    done = panics = 0
    for case in range(400):
        mode = list(MODES)[int(rng.integers(4))]
        hs, vs = MODES[mode]
        w = int(rng.integers(1, 700)) if case % 3 else int(rng.integers(1, 44)) * 16
        h = int(rng.integers(1, 90))
        kind = kinds[int(rng.integers(len(kinds)))]
        planes, qts = synth.make_frame(w, h, hs, vs, 3, seed=1000 + case)
        if kind in ("rgb", "gray", "ycbcr"):
            cs = {"rgb": oc.RGB, "gray": oc.GRAYSCALE, "ycbcr": oc.YCBCR}[kind]
            f = oc.make_frame(w, h, hs, vs, 3, cs, qts)
            rc, exp = oc.decode_planes(f, planes)
            rce, out = emu_c.decode_planes(f, planes)
            if rc != 0:
                assert rce == -5, (case, w, h, mode, kind, rc, rce)
                panics += 1
                continue
        else:
            rc, exp = _plain_expected(w, h, hs, vs, qts, planes, kind)
            assert rc == 0
            f = oc.make_frame(w, h, hs, vs, 3, oc.RGBA if kind == "rgba" else oc.RGB, qts)
            rce, out = emu_c.decode_planes(f, planes, flags=1 if kind == "plain" else 0, out_layout=1 if kind == "chw" else 0)
        assert rce == 0, (case, w, h, mode, kind, rce)
        bad = np.nonzero(out != exp)[0]
        assert bad.size == 0, (case, w, h, mode, kind, bad[:8])
        done += 1
    assert done > 300


@pytest.mark.parametrize("mode", list(MODES))
@pytest.mark.parametrize("flags,out_cs,layout", [(7, oc.RGB, 0), (2, oc.GRAYSCALE, 0), (4, oc.YCBCR, 0), (6, oc.RGB, 0),
                                                 (6, oc.RGBA, 0), (6, oc.RGB, 1), (3, oc.RGB, 0), (5, oc.RGB, 0)])
@pytest.mark.parametrize("wh", [(64, 64), (528, 40), (32, 8), (1040, 33), (100, 32), (37, 50), (17, 16), (250, 72)])
@pytest.mark.parametrize("compact", [0, 1])
def test_emulated_corrected_mode_flags(mode, flags, out_cs, layout, wh, compact, synth):
    """ZJ_FLAG_CLAMP_DC (Q1) and ZJ_FLAG_EDGE_REPLICATE (Q4), alone and combined with PLAIN_TAIL / RGBA / CHW, against
    the oracle's zjo_decode_planes_ext; the adversarial set is where unclamped DC-only values actually occur."""
    emu_c.set_variant(compact)
    hs, vs = MODES[mode]
    w, h = wh
    for adversarial in (False, True):
        mk = synth.make_adversarial_frame if adversarial else synth.make_frame
        planes, qts = mk(w, h, hs, vs, 3, seed=91)
        f = oc.make_frame(w, h, hs, vs, 3, out_cs, qts)
        ext = flags | (oc.EXT_PLAIN if (out_cs == oc.RGBA or layout == 1) else 0)
        rc, exp = oc.decode_planes(f, planes, ext=ext)
        rce, out = emu_c.decode_planes(f, planes, flags=flags, out_layout=layout)
        if rc != 0:
            assert rce == -5
            continue
        if layout == 1:
            exp = np.ascontiguousarray(exp.reshape(h, w, 3).transpose(2, 0, 1)).reshape(-1)
        assert rce == 0
        bad = np.nonzero(out != exp)[0]
        assert bad.size == 0, (mode, flags, out_cs, layout, wh, adversarial, bad[:8])


@pytest.mark.parametrize("mode", list(MODES))
@pytest.mark.parametrize("out_cs", [oc.RGB, oc.GRAYSCALE, oc.YCBCR])
@pytest.mark.parametrize("wh,n", [((64, 64), 5), ((272, 40), 3), ((37, 50), 4)])
def test_emulated_scattered_frames_match_oracle(mode, out_cs, wh, n, synth):
    """zj_decode_frames_device's form: every frame an allocation of its own, the launch's pointer table filled in an
    order that is not the allocation order (Params::fptr, set_scatter)."""
    hs, vs = MODES[mode]
    w, h = wh
    frames = [synth.make_frame(w, h, hs, vs, 3, seed=77, frame_index=i) for i in range(n)]
    f = oc.make_frame(w, h, hs, vs, 3, out_cs, frames[0][1])
    order = [(3 * i + 1) % n for i in range(n)] if n % 3 else list(range(n))[::-1]
    rc, outs = emu_c.decode_frames(f, [fr[0] for fr in frames], order=order)
    orc = oc.decode_planes(f, frames[0][0])[0]
    if orc != 0:  # a geometry on which the reference panics (padded width, GRAYSCALE): the same verdict, nothing decoded
        assert rc == -5
        return
    assert rc == 0
    for i, fr in enumerate(frames):
        rc, exp = oc.decode_planes(f, fr[0])
        assert rc == 0
        assert np.array_equal(outs[i], exp), (mode, out_cs, wh, i)


def test_emulated_scattered_more_frames_than_one_table(synth):
    """More frames than a launch's table holds (SCATTER_MAX = 32): cut into launches, every frame still lands where its
    own pointer says; zero_fill = 0 leaves the never-written bytes alone in every frame."""
    w, h, n = 32, 16, 70
    frames = [synth.make_frame(w, h, 2, 1, 3, seed=3, frame_index=i % 7) for i in range(n)]
    f = oc.make_frame(w, h, 2, 1, 3, oc.RGB, frames[0][1])
    rc, outs = emu_c.decode_frames(f, [fr[0] for fr in frames], zero_fill=0, poison=0x33)
    assert rc == 0
    exps = [oc.decode_planes(f, frames[i][0])[1].reshape(h, 3 * w) for i in range(7)]
    for i in range(n):
        got = outs[i].reshape(h, 3 * w)
        assert np.array_equal(got[:, :3 * w - 16], exps[i % 7][:, :3 * w - 16]), i
        assert (got[:, 3 * w - 16:] == 0x33).all()


@pytest.mark.parametrize("mode", list(MODES))
@pytest.mark.parametrize("w", [65, 255, 257, 289, 302, 303, 304, 305, 311, 312, 313, 319, 321, 511, 513, 535, 560, 561, 569, 570, 575, 577, 767, 769,
                               1023, 1025, 1079, 1081, 1082, 2500, 2501, 2502, 2503])
def test_emulated_ragged_width_split_into_fast_interior_and_generic_edge(mode, w, synth):
    """A ragged width runs the RAG kernels (zj_device.h: phase_color): every ordinary 16-pixel group of a row on the fast
    path -- rows starting at any byte --, the groups at the row's end on the generic store path, in one launch.  Widths on
    both sides of every boundary (tile widths 256 / 512 pixels, the 48-pixel margin in front of the early-written tail,
    P % 16 == 8), every output kind, every kernel variant, both the reference's bytes (zero_fill) and the strip-level
    contract (never-written bytes untouched)."""
    hs, vs = MODES[mode]
    h = 8 * vs * (2 if hs == 2 else 1) + 3   # one strip and a clipped second one
    planes, qts = synth.make_frame(w, h, hs, vs, 3, seed=w)
    for out_cs in (oc.RGB, oc.GRAYSCALE, oc.YCBCR):
        f = oc.make_frame(w, h, hs, vs, 3, out_cs, qts)
        rc, exp = oc.decode_planes(f, planes)
        for variant in (2, 1, 0):  # direct stores with the ragged split, the wide generation (generic kernels), staged stores
            emu_c.set_variant(variant)
            rce, out = emu_c.decode_planes(f, planes)
            if rc == 0:
                assert rce == 0 and np.array_equal(out, exp), (mode, w, out_cs, variant)
        if rc != 0:
            assert rce == -5
            continue
        assert rce == 0
        bad = np.nonzero(out != exp)[0]
        assert bad.size == 0, (mode, w, out_cs, bad[:8], bad.size)
        if out_cs == oc.RGB:  # zero_fill = 0: whatever the reference never writes keeps the caller's bytes
            rce, raw = emu_c.decode_planes(f, planes, zero_fill=0, poison=0x5C)
            rce2, raw2 = emu_c.decode_planes(f, planes, zero_fill=0, poison=0xA3)
            assert rce == 0 and rce2 == 0
            never = raw != raw2                      # bytes that kept the poison
            assert np.array_equal(raw[~never], exp[~never]) and not exp[never].any()
    # the extensions take the same split: plain placement, RGBA, planar
    for out_cs, flags, layout in ((oc.RGB, 1, 0), (oc.RGBA, 0, 0), (oc.RGB, 0, 1)):
        f = oc.make_frame(w, h, hs, vs, 3, out_cs, qts)
        rc, exp = oc.decode_planes(f, planes, plain=True)
        assert rc == 0
        if layout == 1:
            exp = np.ascontiguousarray(exp.reshape(h, w, 3).transpose(2, 0, 1)).reshape(-1)
        rce, out = emu_c.decode_planes(f, planes, flags=flags, out_layout=layout)
        assert rce == 0 and np.array_equal(out, exp), (mode, w, out_cs, flags, layout)


@pytest.mark.parametrize("mode", list(MODES))
@pytest.mark.parametrize("w", [32, 64, 250, 272, 303, 528, 1000, 1040, 2500])
def test_emulated_out_pitch_rows_are_the_tight_rows(mode, w, synth):
    """zj_frame_desc.out_pitch (rows laid out wider than they are, for outputs that stay in HBM): every row of the padded
    layout holds the bytes of the tight layout's row, the bytes between a row's end and the next row's start keep the
    caller's, in every output kind and kernel variant; pitches that are multiples of 128 / 16 / (ragged widths) of nothing;
    the arguments make_plan refuses."""
    hs, vs = MODES[mode]
    h = 8 * vs * (2 if hs == 2 else 1) + 3   # one strip and a clipped second one
    planes, qts = synth.make_frame(w, h, hs, vs, 3, seed=w + 7)
    for out_cs, flags, layout, ncomp in ((oc.RGB, 0, 0, 3), (oc.GRAYSCALE, 0, 0, 1), (oc.YCBCR, 0, 0, 3), (oc.RGB, 1, 0, 3),
                                         (oc.RGBA, 0, 0, 4), (oc.RGB, 0, 1, 3)):
        f = oc.make_frame(w, h, hs, vs, 3, out_cs, qts)
        row = w if layout == 1 else w * ncomp
        planes_out = 3 if layout == 1 else 1
        pitches = [(row + 127) // 128 * 128, (row + 15) // 16 * 16 + 16] + ([row + 5] if w % 16 else [])
        for variant in (0, 2, 1):
            emu_c.set_variant(variant)
            for zf in (0, 1):
                rc, tight = emu_c.decode_planes(f, planes, zero_fill=zf, poison=0x5C, flags=flags, out_layout=layout)
                if rc != 0:
                    assert rc == -5
                    continue
                for pitch in pitches:
                    rc, out = emu_c.decode_planes(f, planes, zero_fill=zf, poison=0x5C, flags=flags, out_layout=layout, out_pitch=pitch)
                    assert rc == 0 and out.size == pitch * h * planes_out, (mode, w, out_cs, layout, variant, pitch)
                    o = out.reshape(planes_out * h, pitch)
                    assert np.array_equal(o[:, :row], tight.reshape(planes_out * h, row)), (mode, w, out_cs, flags, layout, variant, zf, pitch)
                    if zf == 0:
                        assert (o[:, row:] == 0x5C).all(), (mode, w, out_cs, layout, variant, pitch)
    emu_c.set_variant(0)
    f = oc.make_frame(w, h, hs, vs, 3, oc.RGB, qts)
    assert emu_c.decode_planes(f, planes, out_pitch=3 * w - 1)[0] == -1          # shorter than a row
    if w % 16 == 0:
        assert emu_c.decode_planes(f, planes, out_pitch=3 * w + 8)[0] == -1      # aligned kernels: 16-byte rows
    assert emu_c.decode_planes(f, planes, out_pitch=(1 << 20) + 16)[0] == -1


def test_random_descriptors_are_decoded_or_refused_never_anything_else(synth):
    """make_plan (csrc/zj_plan.h, shared by the product and the emulation) over random descriptors, legal and not: sampling
    factors, colour spaces, flags, layouts, pitches.  Every call returns one of the documented verdicts, an illegal
    descriptor is never accepted, and whatever is accepted decodes to the rows of the tight layout (which the other tests
    pin to the oracle)."""
    import ctypes as C
    rng = np.random.default_rng(20260)
    L = emu_c.lib()

    def raw(w, h, hs, vs, out_cs, qts, planes, flags, layout, pitch):
        d = emu_c.FrameDesc()
        d.width, d.height, d.h_max, d.v_max, d.in_components, d.out_colorspace = w, h, hs, vs, 3, out_cs
        for c in range(3):
            q = np.ascontiguousarray(qts[c], np.int32)
            C.memmove(d.qt[c], q.ctypes.data, 256)
        d.flags, d.out_layout, d.out_pitch = flags, layout, pitch
        cap = (max(pitch if pitch <= (1 << 20) else 0, 4 * w) * h * 3 + 64) if pitch <= (1 << 20) else 64
        buf = np.full(cap, 0x5C, np.uint8)
        rc = L.zje_decode_planes(C.byref(d), C.c_size_t(1), C.c_void_p(planes[0].ctypes.data), C.c_void_p(planes[1].ctypes.data),
                                 C.c_void_p(planes[2].ctypes.data), C.c_void_p(buf.ctypes.data), C.c_int(0))
        return rc, buf

    seen = {0: 0, -1: 0, -2: 0, -5: 0}
    for it in range(500):
        w, h = int(rng.integers(1, 400)), int(rng.integers(1, 80))
        hs, vs = int(rng.choice([1, 2, 1, 2, 3, 0])), int(rng.choice([1, 2, 1, 2, 4]))
        out_cs = int(rng.choice([oc.RGB, oc.GRAYSCALE, oc.YCBCR, oc.RGBA, oc.CMYK, oc.YCCK, 9]))
        flags = int(rng.choice([0, 0, 1, 2, 4, 7, 8, 1 << 20]))
        layout = int(rng.choice([0, 0, 0, 1, 2]))
        ncomp = {oc.RGB: 3, oc.GRAYSCALE: 1, oc.YCBCR: 3}.get(out_cs, 4)
        planar = layout == 1 and out_cs == oc.RGB
        row = w if planar else w * ncomp
        pitch = int(rng.choice([0, 0, row, row + 16, (row + 127) // 128 * 128, max(row - 1, 1), row + int(rng.integers(1, 300)), (1 << 20) + 16]))
        geo = hs in (1, 2) and vs in (1, 2)
        planes, qts = synth.make_frame(w, h, hs if geo else 1, vs if geo else 1, 3, seed=it)
        planes = [np.ascontiguousarray(p, np.int16) for p in planes]
        rc, out = raw(w, h, hs, vs, out_cs, qts, planes, flags, layout, pitch)
        assert rc in seen, (it, rc, w, h, hs, vs, out_cs, flags, layout, pitch)
        seen[rc] += 1
        fast = w % 16 == 0 and w >= 32
        legal = (geo and out_cs in (oc.RGB, oc.GRAYSCALE, oc.YCBCR, oc.RGBA) and flags in (0, 1, 2, 4, 7) and layout in (0, 1)
                 and not (layout == 1 and out_cs in (oc.YCBCR, oc.RGBA)) and (pitch == 0 or (row <= pitch <= (1 << 20) and not (fast and pitch % 16))))
        if not legal:
            assert rc != 0, (it, w, h, hs, vs, out_cs, flags, layout, pitch)
            continue
        if rc != 0:
            assert rc == -5, (it, rc)      # the reference itself panics on this geometry
            continue
        rc0, tight = raw(w, h, hs, vs, out_cs, qts, planes, flags, layout, 0)
        assert rc0 == 0
        p_, nrows = pitch or row, h * (3 if planar else 1)
        assert np.array_equal(out[:nrows * p_].reshape(nrows, p_)[:, :row], tight[:nrows * row].reshape(nrows, row)), (it, w, h, hs, vs, out_cs, flags, layout, pitch)
        assert (out[:nrows * p_].reshape(nrows, p_)[:, row:] == 0x5C).all()
    assert seen[0] > 40 and seen[-1] > 40 and seen[-2] > 5, seen
