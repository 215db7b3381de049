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


# ---------------------------------------------------------------------------------------------------------------------
# Integer-only generator (round 3): the same frame statistics as make_frame (DC random walk reflected into
# [-1024, 1016], Laplace AC with scale 24*exp(-k/6) per zig-zag position, 35 % DC-only blocks, 50 % low-pass blocks,
# Annex-K tables at quality 90), but every sample is a pure function of (seed + frame index, component, element index)
# computed with 64-bit integer operations only -- splitmix64 hashing, threshold tables, an integer cumulative sum -- so
# torch produces the SAME planes on the CPU (the build container, where the oracle turns them into the golden checksums
# of tests/golden/checksums_seed1234.json) and on the GPU (bench.py generates a 128-frame shard in HBM in a second instead
# of 3 s per frame on the host).  BASELINE.json configs[4]: 1024 frames, frame i <- seed 1234 + i.
# ---------------------------------------------------------------------------------------------------------------------
_GOLD = 0x9E3779B97F4A7C15
_M1 = 0xBF58476D1CE4E5B9
_M2 = 0x94D049BB133111EB
_U64 = (1 << 64) - 1


def _s64(c):
    """Python int (mod 2^64) -> the signed value torch.int64 holds for that bit pattern."""
    c &= _U64
    return c - (1 << 64) if c >= (1 << 63) else c


def splitmix64_int(x):
    """splitmix64 finaliser on a Python int (the definition the tensor versions are tested against)."""
    z = (x + _GOLD) & _U64
    z = ((z ^ (z >> 30)) * _M1) & _U64
    z = ((z ^ (z >> 27)) * _M2) & _U64
    return z ^ (z >> 31)


def _lsr(t, s):
    """Logical right shift of an int64 tensor (torch's >> is arithmetic)."""
    return (t >> s) & ((1 << (64 - s)) - 1)


def _splitmix64_t(x):
    z = x + _s64(_GOLD)
    z = (z ^ _lsr(z, 30)) * _s64(_M1)
    z = (z ^ _lsr(z, 27)) * _s64(_M2)
    return z ^ _lsr(z, 31)


_THR_CACHE = {}


def ac_thresholds(q_nat):
    """Per zig-zag position k = 1..63: T_k[n] = floor(2^31 * P(|round(Laplace(s_k) / q_k)| >= n)), n = 1, 2, ... while
    positive, with s_k = 24 exp(-k/6) and P(|v| >= n) = exp(-(n - 1/2) q_k / s_k).  Computed with `decimal` (correctly
    rounded exp at 50 digits): the same integers on every machine.  Returns (boundaries, cum_end): all k*2^31 + T_k[n] in
    ascending order, and cum_end[k] = how many of them belong to positions <= k."""
    key = tuple(int(v) for v in q_nat)
    if key in _THR_CACHE:
        return _THR_CACHE[key]
    import decimal
    ctx = decimal.Context(prec=50)
    bounds, cum_end = [], [0]
    for k in range(1, 64):
        qk = decimal.Decimal(int(q_nat[UN_ZIGZAG[k]]))
        sk = ctx.multiply(decimal.Decimal(24), ctx.exp(ctx.divide(decimal.Decimal(-k), decimal.Decimal(6))))
        ts, n = [], 1
        while True:
            p = ctx.exp(ctx.divide(ctx.multiply(-(decimal.Decimal(n) - decimal.Decimal("0.5")), qk), sk))
            t = int(ctx.multiply(p, decimal.Decimal(1 << 31)).to_integral_value(rounding=decimal.ROUND_FLOOR))
            if t < 1:
                break
            ts.append(t)
            n += 1
        bounds.extend(sorted((k << 31) + t for t in ts))
        cum_end.append(len(bounds))
    _THR_CACHE[key] = (bounds, cum_end)
    return _THR_CACHE[key]


def frame_key(seed, frame_index, comp, stream):
    """64-bit key of one (frame, component, stream): stream 0 = per-coefficient, 1 = per-block."""
    return splitmix64_int(((seed + frame_index) * 8 + comp * 2 + stream) & _U64)


def _plane_t(torch, key_ac, key_blk, nblocks, q_nat, device, p_dc_only=22938, p_lowpass=32768):
    """One component's plane [nblocks * 64] int16 (natural order) on `device`."""
    i64 = torch.int64
    bounds, cum_end = ac_thresholds(q_nat)
    bnd = torch.tensor(bounds, dtype=i64, device=device)
    cend = torch.tensor(cum_end, dtype=i64, device=device)           # index k (cum_end[0] = 0 unused)
    k = torch.arange(64, dtype=i64, device=device)
    out = torch.empty((nblocks, 64), dtype=torch.int16, device=device)
    nat = torch.tensor(UN_ZIGZAG, dtype=i64, device=device)
    chunk = 1 << 16                                                    # blocks per pass: bounded temporaries
    carry = 0                                                          # DC walk state between chunks (Python int)
    q0 = int(q_nat[0])
    span = 2040 << 10
    for b0 in range(0, nblocks, chunk):
        nb = min(chunk, nblocks - b0)
        blk = torch.arange(b0, b0 + nb, dtype=i64, device=device)
        # per-coefficient hash -> sign and magnitude (zig-zag position k of block b is element b*64 + k)
        h = _splitmix64_t((blk[:, None] * 64 + k[None, :]) * _s64(_GOLD) + _s64(key_ac))
        u = _lsr(h, 33)                                                # 31 bits
        neg = (h >> 32) & 1
        i = torch.bucketize((k[None, :] << 31) + u, bnd, right=True)   # boundaries <= key
        mag = cend[k][None, :] - i                                     # thresholds of position k above u
        mag = torch.where(k[None, :] == 0, torch.zeros_like(mag), mag)
        val = torch.where(neg == 1, -mag, mag)
        # per-block hash -> DC step (sum of four bytes: Irwin-Hall, sigma 147.8 -> 83/1024 of it = 11.98), DC-only, low-pass
        hb = _splitmix64_t(blk * _s64(_GOLD) + _s64(key_blk))
        step = (((hb >> 32) & 255) + ((hb >> 40) & 255) + ((hb >> 48) & 255) + ((hb >> 56) & 255) - 510) * 83
        walk = torch.cumsum(step, 0) + carry
        carry = int(walk[-1].item())
        dc = torch.abs(torch.remainder(walk + (1024 << 10), 2 * span) - span) - (1024 << 10)   # triangle wave, 1/1024 units
        den = q0 << 10
        val[:, 0] = torch.div(2 * dc + den, 2 * den, rounding_mode="floor")
        dc_only = (hb & 0xFFFF) < p_dc_only
        lowpass = ((hb >> 16) & 0xFFFF) < p_lowpass
        val = torch.where(dc_only[:, None] & (k[None, :] >= 1), torch.zeros_like(val), val)
        val = torch.where(lowpass[:, None] & (k[None, :] >= 21), torch.zeros_like(val), val)
        out[b0:b0 + nb].index_copy_(1, nat, val.to(torch.int16))       # zig-zag position k -> natural index
    return out.reshape(-1)


def make_frame_t(width, height, h_max=2, v_max=2, in_components=3, seed=1234, frame_index=0, quality=90, device="cpu",
                 out=None):
    """Integer-only frame (see above): (planes, qts), planes = flat int16 torch tensors on `device` (written into the
    tensors of `out` when given)."""
    import torch
    qts = quant_tables(quality)
    planes = []
    for c in range(in_components):
        br, bc = plane_blocks(width, height, h_max, v_max, c)
        p = _plane_t(torch, frame_key(seed, frame_index, c, 0), frame_key(seed, frame_index, c, 1), br * bc, qts[c], device)
        if out is not None:
            out[c].copy_(p)
            p = out[c]
        planes.append(p)
    return planes, qts


def checksum_weights_t(nwords, device):
    import torch
    return (torch.arange(nwords, dtype=torch.int64, device=device) * _s64(_GOLD)) | 1


def frame_checksum_t(out_u8, weights=None):
    """64-bit checksum of one decoded frame held in a torch uint8 tensor (size % 8 == 0): the wrapping sum of its
    little-endian 64-bit words times odd position weights.  Same integer on CPU and GPU; frame_checksum_sum is the
    numpy statement of it."""
    import torch
    w = out_u8.reshape(-1).view(torch.int64)
    if weights is None:
        weights = checksum_weights_t(w.numel(), w.device)
    return int((w * weights).sum().item()) & _U64


def frame_checksum_sum(out_bytes):
    a = np.ascontiguousarray(np.asarray(out_bytes, dtype=np.uint8).reshape(-1))
    assert a.size % 8 == 0
    w = a.view(np.uint64)
    with np.errstate(over="ignore"):
        k = (np.arange(w.size, dtype=np.uint64) * np.uint64(_GOLD)) | np.uint64(1)
        return int((w * k).sum(dtype=np.uint64))
