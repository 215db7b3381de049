"""
oracle_np.py -- second, INDEPENDENT restatement (numpy) of zune-jpeg's scalar post-entropy pixel path.

TEST INFRASTRUCTURE ONLY.  Written from the Rust sources (not from oracle/zj_oracle.c) and
vectorised across blocks / rows so that it shares no control flow with the C restatement; the two are
cross-checked bit-for-bit in tests/ (N-version check).  It also generates the
committed golden fixtures (tools/make_golden.py).  Only tests/, tools/ and bench/smoke checkers may
import this module; the product path never does.

Reference paths are relative to /root/reference.  numpy int32/int16 array arithmetic wraps silently,
which is exactly the Rust release-mode behaviour the reference relies on (src/idct.rs:86-87).
"""
import numpy as np

RGB, GRAYSCALE, YCBCR, CMYK, YCCK, RGBA, RGBX = range(7)  # src/misc.rs:88-106


def num_components(cs):  # src/misc.rs:113-121
    return {RGB: 3, YCBCR: 3, GRAYSCALE: 1, CMYK: 4, YCCK: 4, RGBA: 4, RGBX: 4}[cs]


class Panic(Exception):
    """The reference would panic here (bounds check / unwrap / assert!)."""


# ---------------------------------------------------------------------------------------------
# IDCT  (src/idct/scalar.rs)
# ---------------------------------------------------------------------------------------------
SCALE_BITS = 512 + 65536 + (128 << 17)  # scalar.rs:6

_I32 = np.int32


def _pass(s, bias):
    """8-point pass on int32 arrays s[0..7] (each any shape).  scalar.rs:81-166 / :175-269."""
    c = lambda v: _I32(v)
    with np.errstate(over="ignore"):
        p2, p3 = s[2], s[6]
        p1 = (p2 + p3) * c(2217)
        t2 = p1 + p3 * c(-7567)
        t3 = p1 + p2 * c(3135)
        p2, p3 = s[0], s[4]
        t0 = (p2 + p3) << c(12)
        t1 = (p2 - p3) << c(12)
        x0 = t0 + t3 + c(bias)
        x3 = t0 - t3 + c(bias)
        x1 = t1 + t2 + c(bias)
        x2 = t1 - t2 + c(bias)
        t0, t1, t2, t3 = s[7], s[5], s[3], s[1]
        p3 = t0 + t2
        p4 = t1 + t3
        p1 = t0 + t3
        p2 = t1 + t2
        p5 = (p3 + p4) * c(4816)  # f2f(1.175875602), scalar.rs:224,287
        t0 = t0 * c(1223)
        t1 = t1 * c(8410)
        t2 = t2 * c(12586)
        t3 = t3 * c(6149)
        p1 = p5 + p1 * c(-3685)
        p2 = p5 + p2 * c(-10497)
        p3 = p3 * c(-8034)
        p4 = p4 * c(-1597)
        t3 = t3 + (p1 + p4)
        t2 = t2 + (p2 + p3)
        t1 = t1 + (p2 + p4)
        t0 = t0 + (p1 + p3)
        return [x0 + t3, x1 + t2, x2 + t1, x3 + t0, x3 - t0, x2 - t1, x1 - t2, x0 - t3]


def idct_blocks(blocks, qt):
    """blocks: (N,64) int16 natural order; qt: (64,) int32.  Returns (N,8,8) int16 pixel blocks
    [row][col] exactly as dequantize_and_idct_int produces them (scalar.rs:40-279)."""
    blocks = np.ascontiguousarray(blocks, dtype=np.int16).reshape(-1, 64)
    qt = np.asarray(qt, dtype=np.int32).reshape(64)
    n = blocks.shape[0]
    with np.errstate(over="ignore"):
        deq = (blocks.astype(np.int32) * qt[None, :]).reshape(n, 8, 8)  # [vfreq k][col]
        # column pass (scalar.rs:79-167): transform axis = k (rows of the coefficient matrix)
        o = _pass([deq[:, k, :] for k in range(8)], 512)
        tmp = np.stack([v >> _I32(10) for v in o], axis=1)  # tmp[n][k][col] == tmp[ptr + 8k]
        # row pass (scalar.rs:170-274): for row i, inputs tmp[i*8 + j]
        o2 = _pass([tmp[:, :, j] for j in range(8)], SCALE_BITS)
        full = np.stack([np.clip(v >> _I32(17), 0, 255) for v in o2], axis=2).astype(np.int16)
        # DC-only shortcut (scalar.rs:45-74): i16 wrapping mul, arithmetic >>3, +128, NOT clamped
        dc_only = ~np.any(blocks[:, 1:] != 0, axis=1)
        dcv = ((blocks[:, 0] * qt[0].astype(np.int16)) >> np.int16(3)) + np.int16(128)
    out = full
    out[dc_only] = dcv[dc_only][:, None, None]
    return out


def idct_strip(coeff, qt, stride, samp_factors, v_samp, clamp_dc=False):
    """src/idct/scalar.rs:19-282 incl. the strip layout: chunk c (= one block row) is written as a
    raster of 8 rows x `stride`, block k at columns 8k..8k+8.
    clamp_dc=True is an EXTENSION: the DC-only shortcut value is clamped to 0..255 (Q1 corrected)."""
    coeff = np.ascontiguousarray(coeff, dtype=np.int16)
    n = coeff.size
    out = np.zeros(n, dtype=np.int16)
    if n == 0:
        return out
    chunks = n * v_samp // samp_factors
    if chunks == 0:
        raise Panic("chunks_exact(0)")
    nchunks = n // chunks
    nblk = chunks // 64
    if nblk == 0:
        return out
    # bounds: last block's last row must fit inside the chunk (get_mut(..).unwrap())
    if (nblk - 1) * 8 + 7 * stride + 8 > chunks:
        raise Panic("idct out-of-chunk write")
    blocks = coeff[: nchunks * chunks].reshape(nchunks, chunks)[:, : nblk * 64].reshape(-1, 64)
    px = idct_blocks(blocks, qt)
    if clamp_dc:
        dc_only = ~np.any(blocks[:, 1:] != 0, axis=1)
        px = px.copy()
        px[dc_only] = np.clip(px[dc_only], 0, 255)
    px = px.reshape(nchunks, nblk, 8, 8)
    view = out[: nchunks * chunks].reshape(nchunks, chunks)
    for r in range(8):
        # row r of every block lands at r*stride + 8k .. +8
        dst = view[:, r * stride : r * stride + nblk * 8].reshape(nchunks, nblk, 8)
        dst[...] = px[:, :, r, :]
    return out


# ---------------------------------------------------------------------------------------------
# Up-sampling  (src/upsampler/scalar.rs)
# ---------------------------------------------------------------------------------------------
def upsample_horizontal(inp, output_len):
    """scalar.rs:5-60.  Flat 1-D triangle filter over the WHOLE input."""
    inp = np.ascontiguousarray(inp, dtype=np.int16)
    n = inp.size
    if not (output_len > 4 and n > 2):
        raise Panic("Too Short of a vector, cannot upsample")
    out = np.zeros(output_len, dtype=np.int16)
    three, two = np.int16(3), np.int16(2)
    with np.errstate(over="ignore"):
        out[0] = inp[0]
        out[1] = (inp[0] * three + inp[1] + two) >> two
        m = min((output_len - 2) // 2, n - 2)  # zip(chunks_exact_mut(2), windows(3))
        mid = inp[1 : 1 + m] * three + two
        out[2 : 2 + 2 * m : 2] = (mid + inp[0:m]) >> two
        out[3 : 3 + 2 * m : 2] = (mid + inp[2 : 2 + m]) >> two
        out[output_len - 2] = (three * inp[n - 2] + inp[n - 1] + two) >> two  # scalar.rs:55
        out[output_len - 1] = inp[n - 1]  # scalar.rs:57
    return out


# (near, far) row schedule produced by the two chunks_exact iterators + next_row flag
# (scalar.rs:84-144), derived by running the iterator logic below once for 8 rows.
def _vertical_schedule(nrows):
    near_next = far_next = 0

    def nxt(which):
        nonlocal near_next, far_next
        if which == "n":
            if near_next < nrows:
                near_next += 1
                return near_next - 1
            return None
        if far_next < nrows:
            far_next += 1
            return far_next - 1
        return None

    rw_n = nxt("n")
    rw_f = nxt("f")
    sched = []
    next_row = True
    for _ in range(8):
        sched.append((rw_n, rw_f))
        previous = rw_n
        v = nxt("n")
        rw_n = previous if v is None else v
        v = nxt("f")
        rw_f = rw_n if v is None else v
        if next_row:
            v = nxt("f")
            rw_f = rw_n if v is None else v
            next_row = False
    return sched


def upsample_vertical(inp, output_len):
    """scalar.rs:64-147.  Treats the input as exactly 8 rows of len>>3 samples."""
    inp = np.ascontiguousarray(inp, dtype=np.int16)
    n = inp.size
    stride = n >> 3
    if stride == 0:
        raise Panic("chunks_exact(0)")
    nrows = n // stride
    rows = inp[: nrows * stride].reshape(nrows, stride)
    out = np.zeros(output_len, dtype=np.int16)
    three, two = np.int16(3), np.int16(2)
    i = 0
    with np.errstate(over="ignore"):
        for near, far in _vertical_schedule(nrows):
            if i + stride > output_len:
                raise Panic("split_at_mut")
            cnt = min(stride, output_len - i - stride)
            a, b = rows[near][:cnt], rows[far][:cnt]
            out[i : i + cnt] = (a * three + b + two) >> two
            out[i + stride : i + stride + cnt] = (b * three + a + two) >> two
            i += 2 * stride
    return out


def upsample_hv(inp, output_len):
    """scalar.rs:148-166"""
    inp = np.ascontiguousarray(inp, dtype=np.int16)
    return upsample_horizontal(upsample_vertical(inp, inp.size * 2), output_len)


# ---------------------------------------------------------------------------------------------
# Colour conversion  (src/color_convert/scalar.rs)
# ---------------------------------------------------------------------------------------------
def ycbcr_to_rgb_px(y, cb, cr):
    """scalar.rs:66-85 on int16 arrays of equal shape -> (...,3) uint8.  i16 wrapping products."""
    y = np.asarray(y, dtype=np.int16)
    with np.errstate(over="ignore"):
        cr = np.asarray(cr, dtype=np.int16) - np.int16(128)
        cb = np.asarray(cb, dtype=np.int16) - np.int16(128)
        r = y + ((np.int16(45) * cr) >> np.int16(5))
        g = y - ((np.int16(11) * cb + np.int16(23) * cr) >> np.int16(5))
        b = y + ((np.int16(113) * cb) >> np.int16(6))
    return np.stack([np.clip(c, 0, 255).astype(np.uint8) for c in (r, g, b)], axis=-1)


def ycbcr_to_rgb_16(y, cb, cr, out, pos):
    """scalar.rs:52-89; returns new pos."""
    if pos > out.size or out.size - pos < 48:
        raise Panic("Slice to small cannot write")
    out[pos : pos + 48] = ycbcr_to_rgb_px(y, cb, cr).reshape(48)
    return pos + 48


def ycbcr_to_grayscale(y, width, out):
    """scalar.rs:91-114 (`as u8` truncation)."""
    y = np.asarray(y, dtype=np.int16)
    t = y.astype(np.uint16).astype(np.uint8)  # low byte
    width_mcu = y.size // width
    if width_mcu == 0:
        raise Panic("div by zero")
    width_chunk = y.size // width_mcu
    if width > width_chunk:
        raise Panic("chunk[0..width]")
    rows = y.size // width_chunk
    if rows * width > out.size:
        raise Panic("output[start..end]")
    out[: rows * width] = t[: rows * width_chunk].reshape(rows, width_chunk)[:, :width].reshape(-1)


def ycbcr_to_ycbcr(ch, width, h_samp, v_samp, out):
    """scalar.rs:119-169"""
    n = ch[0].size
    width_chunk = (n // (h_samp * v_samp)) >> 3
    if width_chunk == 0:
        raise Panic("chunks_exact(0)")
    rows = n // width_chunk
    if width * 3 > width_chunk * 3 or rows * width * 3 > out.size:
        raise Panic("slice")
    planes = [np.asarray(c, np.int16).astype(np.uint16).astype(np.uint8)[: rows * width_chunk]
              .reshape(rows, width_chunk)[:, :width] for c in ch]
    out[: rows * width * 3] = np.stack(planes, axis=-1).reshape(-1)


# ---------------------------------------------------------------------------------------------
# Worker glue  (src/worker.rs)
# ---------------------------------------------------------------------------------------------
def color_convert_ycbcr(blk, width, h_samp, v_samp, out_cs, output, plain=False):
    """worker.rs:143-251.  Row-vectorised: the main groups and the early-written tail are applied
    in the reference's order (main first, tail overwrites).
    plain=True is an EXTENSION (no reference output exists for it): every pixel x < width at its own
    position, 4-byte pixels get 255 as 4th byte."""
    n = blk[0].size
    ncomp = num_components(out_cs)
    width_chunk = (n // (h_samp * v_samp)) >> 3
    stride = width * ncomp
    if width_chunk == 0:
        raise Panic("chunks_exact(0)")
    rows = n // width_chunk
    if rows * stride > output.size:
        raise Panic("output[start..end]")
    Y, CB, CR = (np.asarray(b, np.int16)[: rows * width_chunk].reshape(rows, width_chunk) for b in blk)
    outv = output[: rows * stride].reshape(rows, stride)
    if plain:
        m = min(width, width_chunk)
        px = outv.reshape(rows, width, ncomp)
        px[:, :m, :3] = ycbcr_to_rgb_px(Y[:, :m], CB[:, :m], CR[:, :m]).reshape(rows, m, 3)
        if ncomp == 4:
            px[:, :m, 3] = 255
        return
    if width < 16:
        if width_chunk > 16:
            raise Panic("copy_from_slice")
        pad = lambda a: np.pad(a, ((0, 0), (0, 16 - width_chunk)))
        rgb = ycbcr_to_rgb_px(pad(Y), pad(CB), pad(CR)).reshape(rows, 48)
        # temp has 16*ncomp bytes; the RGB routine writes 48 of them
        outv[:, : width * ncomp] = np.pad(rgb, ((0, 0), (0, 16 * ncomp - 48)))[:, : width * ncomp]
        return
    elements = max(width_chunk // 16 - 1, 0)
    position = 48 * elements
    if position > stride:
        raise Panic("Slice to small cannot write")
    if elements:
        m = 16 * elements
        outv[:, :position] = ycbcr_to_rgb_px(Y[:, :m], CB[:, :m], CR[:, :m]).reshape(rows, position)
    diff = max(64 - max(stride - position, 0), 0)
    position = max(position - diff, 0)
    if stride - position < 48:
        raise Panic("Slice to small cannot write")
    t0 = width_chunk - 16
    outv[:, position : position + 48] = ycbcr_to_rgb_px(Y[:, t0:], CB[:, t0:], CR[:, t0:]).reshape(rows, 48)


def upsample_h_rows(inp, row_len):
    """EXTENSION (Q4 corrected): the horizontal triangle filter row by row with replicated edges."""
    c = np.asarray(inp, np.int16)
    rows = c[: (c.size // row_len) * row_len].reshape(-1, row_len).astype(np.int32)
    left = np.concatenate([rows[:, :1], rows[:, :-1]], axis=1)
    right = np.concatenate([rows[:, 1:], rows[:, -1:]], axis=1)
    out = np.empty((rows.shape[0], 2 * row_len), np.int32)
    out[:, 0::2] = (3 * rows + left + 2) >> 2
    out[:, 1::2] = (3 * rows + right + 2) >> 2
    return out.astype(np.int16).reshape(-1)


def post_process(coeff, comps, in_cs, out_cs, output, width, plain=False, clamp_dc=False, edge_rep=False):
    """worker.rs:32-141.  comps: list of dicts {h, v, width_stride, qt}.  plain / clamp_dc / edge_rep: extensions."""
    h_samp, v_samp = comps[0]["h"], comps[0]["v"]
    x = min(num_components(in_cs), num_components(out_cs), 3)
    unp = [None, None, None]
    for z in range(x):
        unp[z] = idct_strip(coeff[z], comps[z]["qt"], comps[z]["width_stride"], h_samp * v_samp,
                            1 if z == 0 else v_samp, clamp_dc)
    if h_samp != 1 or v_samp != 1:
        up = {(2, 1): upsample_horizontal, (1, 2): upsample_vertical, (2, 2): upsample_hv}.get((h_samp, v_samp))
        if up is None:
            raise ValueError("Unknown down-sampling method")
        for i in range(1, x):
            if edge_rep and h_samp == 2:
                mid = unp[i] if v_samp == 1 else upsample_vertical(unp[i], 2 * unp[i].size)
                unp[i] = upsample_h_rows(mid, comps[i]["width_stride"])
            else:
                unp[i] = up(unp[i], unp[0].size)
    if in_cs in (YCBCR, GRAYSCALE) and out_cs == GRAYSCALE:
        ycbcr_to_grayscale(unp[0], width, output)
    elif in_cs == YCBCR and out_cs == YCBCR:
        ycbcr_to_ycbcr(unp, width, h_samp, v_samp, output)
    elif in_cs == YCBCR and out_cs in (RGB, RGBA, RGBX):
        color_convert_ycbcr(unp, width, h_samp, v_samp, out_cs, output, plain)


# ---------------------------------------------------------------------------------------------
# Whole-frame driver  (src/mcu_prog.rs:62-79, :132-246; src/headers.rs:306-339)
# ---------------------------------------------------------------------------------------------
def geometry(width, height, h_max, v_max):
    mcu_x = (width + 8 * h_max - 1) // (8 * h_max)
    mcu_y = (height + 8 * v_max - 1) // (8 * v_max)
    return mcu_x, mcu_y


def plane_len(width, height, h_max, v_max, comp):
    mcu_x, mcu_y = geometry(width, height, h_max, v_max)
    hs, vs = (h_max, v_max) if comp == 0 else (1, 1)
    return mcu_x * 64 * vs * hs * mcu_y


def decode_planes(width, height, h_max, v_max, in_components, out_cs, qts, planes, plain=False, clamp_dc=False,
                  edge_rep=False):
    mcu_x, mcu_y = geometry(width, height, h_max, v_max)
    ncomp = num_components(out_cs)
    in_cs = YCBCR if in_components == 3 else GRAYSCALE
    interleaved = h_max != 1 or v_max != 1
    comps = [dict(h=h_max if c == 0 else 1, v=v_max if c == 0 else 1,
                  width_stride=(h_max if c == 0 else 1) * mcu_x * 8,
                  qt=np.asarray(qts[min(c, len(qts) - 1)], np.int32)) for c in range(3)]
    mcu_width = mcu_x * 64
    bias = 1
    if (h_max, v_max) == (2, 1):
        mcu_width *= 2
    if (h_max, v_max) == (2, 2):
        bias = 2
    total = (width + 8) * (height + 8) * ncomp + int(interleaved) * 128 * height * ncomp
    out_vector = np.zeros(total, dtype=np.uint8)
    chunks_size = width * ncomp * 8 * h_max * v_max
    y_chunk = mcu_width * v_max * h_max * bias
    c_chunk = mcu_width * bias
    n_strips = min(planes[0].size // y_chunk, total // chunks_size)
    if in_components == 3:
        n_strips = min(n_strips, planes[1].size // c_chunk, planes[2].size // c_chunk)
    for s in range(n_strips):
        coeff = [planes[0][s * y_chunk : (s + 1) * y_chunk]]
        if in_components == 3:
            coeff += [planes[1][s * c_chunk : (s + 1) * c_chunk], planes[2][s * c_chunk : (s + 1) * c_chunk]]
        else:
            coeff += [np.zeros(0, np.int16)] * 2
        post_process(coeff, comps, in_cs, out_cs, out_vector[s * chunks_size : (s + 1) * chunks_size], width, plain,
                     clamp_dc, edge_rep)
    return out_vector[: width * height * ncomp].copy()
