"""
Synthetic quantized-coefficient planes for tests and bench (BASELINE.md "Inputs").

Frames are produced directly as whole-image coefficient planes in the layout the reference's entropy
decoder leaves behind (src/mcu_prog.rs:73-79, src/bitstream.rs:343,359): per component
[block_row][block_col][64] int16, natural (un-zigzagged) order, NOT dequantized.  No JPEG file is
involved.  numpy only; deterministic per (seed, frame index).
"""
import numpy as np

# JPEG Annex K tables (zig-zag order as printed in the standard, row-major == natural order here)
_K_LUMA = np.array([
    16, 11, 10, 16, 24, 40, 51, 61, 12, 12, 14, 19, 26, 58, 60, 55, 14, 13, 16, 24, 40, 57, 69, 56,
    14, 17, 22, 29, 51, 87, 80, 62, 18, 22, 37, 56, 68, 109, 103, 77, 24, 35, 55, 64, 81, 104, 113, 92,
    49, 64, 78, 87, 103, 121, 120, 101, 72, 92, 95, 98, 112, 100, 103, 99], dtype=np.int32)
_K_CHROMA = np.array([
    17, 18, 24, 47, 99, 99, 99, 99, 18, 21, 26, 66, 99, 99, 99, 99, 24, 26, 56, 99, 99, 99, 99, 99,
    47, 66, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99,
    99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99], dtype=np.int32)

# natural index of the k-th zig-zag coefficient (same table as src/misc.rs:39-48, first 64 entries)
UN_ZIGZAG = np.array([
    0, 1, 8, 16, 9, 2, 3, 10, 17, 24, 32, 25, 18, 11, 4, 5, 12, 19, 26, 33, 40, 48, 41, 34,
    27, 20, 13, 6, 7, 14, 21, 28, 35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23, 30, 37, 44, 51,
    58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63], dtype=np.int64)


def quant_tables(quality=90):
    """Annex-K tables scaled libjpeg-style to `quality`, clamped to 1..255, natural order, int32.
    Returns [luma, chroma, chroma]."""
    scale = 5000 // quality if quality < 50 else 200 - 2 * quality
    out = []
    for base in (_K_LUMA, _K_CHROMA):
        q = np.clip((base * scale + 50) // 100, 1, 255).astype(np.int32)
        out.append(q)
    return [out[0], out[1], out[1].copy()]


def geometry(width, height, h_max, v_max):
    """mcu_x, mcu_y as src/headers.rs:317-319."""
    return ((width + 8 * h_max - 1) // (8 * h_max), (height + 8 * v_max - 1) // (8 * v_max))


def plane_blocks(width, height, h_max, v_max, comp):
    """(block_rows, block_cols) of component `comp` (src/mcu_prog.rs:76)."""
    mcu_x, mcu_y = geometry(width, height, h_max, v_max)
    hs, vs = (h_max, v_max) if comp == 0 else (1, 1)
    return mcu_y * vs, mcu_x * hs


def _plane(rng, nblocks, q, p_dc_only=0.35, p_lowpass=0.5):
    qz = q[UN_ZIGZAG].astype(np.float64)  # quantizer per zig-zag position
    k = np.arange(64, dtype=np.float64)
    scale = 24.0 * np.exp(-k / 6.0)
    ac = rng.laplace(0.0, 1.0, size=(nblocks, 64)) * scale[None, :]
    coef = np.rint(ac / qz[None, :])
    # DC: clipped random walk over the blocks
    dc = np.cumsum(rng.normal(0.0, 12.0, size=nblocks))
    # reflect the walk into [-1024, 1016] so it does not stick to a rail
    span = 1016.0 + 1024.0
    dc = np.abs(((dc + 1024.0) % (2 * span)) - span)  # triangle wave in [0, span]
    dc = dc - 1024.0
    coef[:, 0] = np.rint(np.clip(dc, -1024, 1016) / qz[0])
    dc_only = rng.random(nblocks) < p_dc_only
    coef[dc_only, 1:] = 0
    lowpass = rng.random(nblocks) < p_lowpass
    coef[lowpass, 21:] = 0
    nat = np.zeros((nblocks, 64), dtype=np.int16)
    nat[:, UN_ZIGZAG] = np.clip(coef, -32768, 32767).astype(np.int16)
    return nat.reshape(-1)


def make_frame(width, height, h_max=2, v_max=2, in_components=3, seed=1234, frame_index=0,
               quality=90):
    """Realistic frame: returns (planes, qts); planes = list of flat int16 arrays (1 or 3)."""
    rng = np.random.default_rng(seed + frame_index)
    qts = quant_tables(quality)
    planes = []
    for c in range(in_components):
        br, bc = plane_blocks(width, height, h_max, v_max, c)
        planes.append(_plane(rng, br * bc, qts[c]))
    return planes, qts


def make_adversarial_frame(width, height, h_max=2, v_max=2, in_components=3, seed=99, frame_index=0):
    """Full-range uniform int16 coefficients and uniform 1..255 tables: exercises every wrap-around
    path (i32 wrap in the IDCT, i16 wrap in the DC-only shortcut and in colour conversion)."""
    rng = np.random.default_rng(seed + frame_index)
    qts = [rng.integers(1, 256, size=64).astype(np.int32) for _ in range(3)]
    planes = []
    for c in range(in_components):
        br, bc = plane_blocks(width, height, h_max, v_max, c)
        p = rng.integers(-32768, 32768, size=(br * bc, 64)).astype(np.int16)
        # a third of the blocks DC-only (Q1 with wrapping products), a few all-zero
        sel = rng.random(br * bc)
        p[sel < 0.33, 1:] = 0
        p[sel < 0.03, :] = 0
        planes.append(p.reshape(-1))
    return planes, qts
