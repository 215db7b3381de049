"""TEST INFRASTRUCTURE (oracle): a bit-level model of how the reference WALKS a baseline scan, i.e. which MCUs it
entropy-decodes at all.  Only tests/ may import this.

The reference leaves the loop over a row's MCUs as soon as its bit reader has come across EOI (src/mcu.rs:337-343),
and its reader looks ahead in groups of four data bytes (src/bitstream.rs:150-262), so on images whose last MCUs are
cheap the last few MCUs are never decoded and keep the zeros of their fresh buffers.  The product's front-end
(zune-jpeg_amd/csrc/zj_jpeg.cpp, EoiCut) reproduces this with a closed-form rule; this file is the literal model the
rule is checked against.  It follows

  src/bitstream.rs:150-262   refill(): nothing while bits_left > 32; otherwise four data bytes (a stuffed 0x00 after
                             0xFF is skipped, 0xFF fill bytes before a marker are skipped, a marker ends the refill
                             and is recorded); with a marker pending, bits_left is set to 63 (zeros are served)
  src/bitstream.rs:276-296   decode_dc(): refill only if bits_left < 16
  src/bitstream.rs:313-373   decode_mcu_block(): refill before every AC symbol
  src/mcu.rs:139-165         row geometry: (2,1) walks 2*mcu_x MCUs per row and mcu_y/2 rows, (2,2) mcu_y/2 x 2 rows
  src/mcu.rs:230-350         the MCU loop, `todo` / handle_rst (:382-419), and the EOI break

decoded_mcus_per_row() follows code lengths and magnitude-bit counts only; decode_baseline_planes() (round 3) also forms
the coefficient VALUES the reference's reader yields -- including the DC symbols it reads short (src/bitstream.rs:278:
no refill with 16..26 bits left although a DC symbol can be longer), which the front-end reproduces (ref_dc_misread in
zj_jpeg.cpp) and tests/test_jpeg_frontend.py checks against this model.
"""
import struct


def _parse(buf):
    p, dht, comps, ri = 2, {}, [], 0
    while True:
        assert buf[p] == 0xFF, "marker expected"
        m = buf[p + 1]
        p += 2
        if m == 0xD8 or m == 0xFF:
            continue
        length = struct.unpack(">H", buf[p:p + 2])[0]
        seg = buf[p + 2:p + length]
        if m == 0xC4:
            q = 0
            while q < len(seg):
                tc, th = seg[q] >> 4, seg[q] & 15
                counts = list(seg[q + 1:q + 17])
                n = sum(counts)
                dht[(tc, th)] = (counts, list(seg[q + 17:q + 17 + n]))
                q += 17 + n
        elif m == 0xC0:
            h, w, nc = struct.unpack(">HHB", seg[1:6])
            comps = [[seg[6 + 3 * i], seg[7 + 3 * i] >> 4, seg[7 + 3 * i] & 15] for i in range(nc)]
        elif m == 0xC2:
            raise ValueError("progressive: the reference's progressive walk has no early exit (src/mcu_prog.rs)")
        elif m == 0xDD:
            ri = struct.unpack(">H", seg)[0]
        elif m == 0xDA:
            ns = seg[0]
            sel = {seg[1 + 2 * i]: (seg[2 + 2 * i] >> 4, seg[2 + 2 * i] & 15) for i in range(ns)}
            return dict(w=w, h=h, comps=comps, dht=dht, sel=sel, start=p + length, ri=ri)
        p += length


def _canonical(counts, vals):
    code, k, table = 0, 0, {}
    for length in range(1, 17):
        for _ in range(counts[length - 1]):
            table[(length, code)] = vals[k]
            k += 1
            code += 1
        code <<= 1
    return table


_M64 = (1 << 64) - 1

# statistics of the last walk: DC symbols read short, and how many of those picked up STALE bits (see _Reader.get)
# "first_marker_block": index (in decode order) of the first block whose decoding began with a marker already in the
# reader's view, i.e. from which on the reference may serve bits it does not hold (None: never)
last_stats = {"short": 0, "stale": 0, "first_marker_block": None}


class _Reader:
    """src/bitstream.rs BitStream, LITERALLY: `buffer` (u64, :105), `aligned_buffer` (u64, :109), `bits_left` (u8, :117)
    and the marker logic of refill().  aligned_buffer is recomputed from `buffer` only by a real refill (:231, :248, :196);
    between refills drop_bits shifts it (zeros enter at the bottom, :386-390) and get_bits ROTATES it (:394-402): the
    magnitude bits a get_bits hands out re-enter at the bottom of aligned_buffer and stay there, below the
    (64 - bits_left after the last refill) zero bits the refill left under the valid data, moving up with every later
    drop / get.  A read that wants more bits than bits_left holds (the short DC read, :278) is served from there: zeros
    first, then those stale magnitude bits (round 4; rounds 2-3 modelled zeros only -- right for 98.8 % of the short reads
    of random streams, wrong when an AC refill at bits_left near 32 left a gap of few zero bits)."""

    def __init__(self, buf, pos):
        self.buf, self.pos, self.buffer, self.aligned, self.bl, self.marker = buf, pos, 0, 0, 0, None
        self.clean = 64  # model bookkeeping (not reference state): low bits of `aligned` known to be refill zeros

    def _byte(self):  # read_u8 (:689-703): zeros past the end
        v = self.buf[self.pos] if self.pos < len(self.buf) else 0
        self.pos += 1
        return v

    def _realign(self):  # self.aligned_buffer = self.buffer << (64 - self.bits_left)
        self.aligned = (self.buffer << (64 - self.bl)) & _M64

    def refill(self):
        if self.bl <= 32 and self.marker is None:
            if self.pos + 4 < len(self.buf) and 0xFF not in self.buf[self.pos:self.pos + 4]:  # :220-236
                self.buffer = ((self.buffer << 32) | int.from_bytes(self.buf[self.pos:self.pos + 4], "big")) & _M64
                self.pos += 4
                self.bl += 32
                self._realign()
                return
            for _ in range(4):  # the refill! macro, :168-213
                b = self._byte()
                self.buffer = ((self.buffer << 8) | b) & _M64
                self.bl += 8
                if b == 0xFF:
                    n = self._byte()
                    if n != 0:
                        while n == 0xFF:
                            n = self._byte()
                        if n != 0:
                            self.buffer >>= 8
                            self.bl -= 8
                            if self.bl != 0:  # :193-197
                                self._realign()
                            self.marker = n
                            return
            self._realign()
        elif self.marker is not None:  # :254-258: bits_left = 63, aligned_buffer stays as it is
            self.bl = 63

    def peek(self, n):  # peek_bits::<n> (:378-382)
        return self.aligned >> (64 - n)

    def drop(self, n):  # drop_bits (:386-390)
        self.bl = max(0, self.bl - n)
        self.aligned = (self.aligned << n) & _M64

    def get(self, n):  # get_bits (:394-402)
        self.aligned = ((self.aligned << n) | (self.aligned >> (64 - n))) & _M64
        self.bl = max(0, self.bl - n)
        return self.aligned & ((1 << n) - 1)

    def symbol(self, table):
        # decode_huff! (:49-86) resolves a code from the top 16 bits of aligned_buffer; every symbol decode is preceded by
        # a refill that leaves at least 16 valid bits there (or by the marker's zeros), so for the canonical tables of a
        # well-formed stream this is the first (length, code) of the prefix-free set found in those bits
        for length in range(1, 17):
            c = self.peek(length)
            if (length, c) in table:
                self.drop(length)
                self.last_len = length
                return table[(length, c)]
        raise ValueError("bad Huffman code")

    def reset(self):  # :671-678
        self.buffer, self.aligned, self.bl, self.marker = 0, 0, 0, None


def _zero_model_bits(s, size):
    """what rounds 2-3 assumed a short read yields: the bits held, zeros below"""
    have = min(s.bl, size)
    return ((s.aligned >> (64 - have)) << (size - have)) if have else 0


_UNZ = [0, 1, 8, 16, 9, 2, 3, 10, 17, 24, 32, 25, 18, 11, 4, 5, 12, 19, 26, 33, 40, 48, 41, 34, 27, 20, 13, 6, 7, 14, 21, 28,
        35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23, 30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55,
        62, 63]


def _sext6(v):
    return ((v & 63) ^ 32) - 32


def _extend(x, s):  # huff_extend, src/bitstream.rs:683-687
    return x - (1 << s) + 1 if x < (1 << (s - 1)) else x


def decoded_mcus_per_row(jpeg_bytes):
    """For every MCU row of the image (mcu_y rows of mcu_x MCUs, the front-end's geometry): how many leading MCUs the
    reference entropy-decodes.  Rows it never reaches at all (an odd last MCU row under (2,1)/(2,2) sampling) report
    None."""
    return _walk(jpeg_bytes, False)[0]


def decode_baseline_planes(jpeg_bytes):
    """The coefficient VALUES of the reference's baseline walk as well: (planes, short_reads, rows).  rows is what
    decoded_mcus_per_row returns (None: an MCU row the walk never reaches, its blocks stay zero here).  planes[c] is component
    c's whole-image plane [block row][block column][64] int16, natural order, zeros where the walk never went;
    short_reads counts the DC symbols the reference reads with fewer bits than they have (decode_dc refills only below
    16 bits, src/bitstream.rs:278: get_bits() then serves zeros for what is missing and the missing bits are parsed
    again as the next symbol).  Damaged streams included (round 6): the fast-AC table's rules -- min(pos, 63) against pos & 63
    once a run passes coefficient 63, size-0 symbols behind short codes, values cut to six bits -- are followed literally."""
    rows, planes, short = _walk(jpeg_bytes, True)
    return planes, short, rows


def _walk(jpeg_bytes, values):
    import numpy as np
    j = _parse(jpeg_bytes)
    hmax = max(c[1] for c in j["comps"])
    vmax = max(c[2] for c in j["comps"])
    ncomp = len(j["comps"])
    if ncomp == 1:
        hmax = vmax = 1  # mcu.rs:170-196
        j["comps"][0][1] = j["comps"][0][2] = 1
    mcu_x = (j["w"] + 8 * hmax - 1) // (8 * hmax)
    mcu_y = (j["h"] + 8 * vmax - 1) // (8 * vmax)
    if ncomp == 3 and hmax == 2 and vmax == 1:
        width, height, bias = 2 * mcu_x, mcu_y // 2, 1
    elif ncomp == 3 and hmax == 2 and vmax == 2:
        width, height, bias = mcu_x, mcu_y // 2, 2
    else:
        width, height, bias = mcu_x, mcu_y, 1
    tabs = {k: _canonical(*v) for k, v in j["dht"].items()}
    s = _Reader(jpeg_bytes, j["start"])
    todo = j["ri"] if j["ri"] else 1 << 62
    loops = []  # MCUs decoded by every pass of the `for j in 0..mcu_width` loop
    planes, pred, short, stale, nblk, first_marker = None, [0] * ncomp, 0, 0, 0, None
    if values:
        planes = [np.zeros((mcu_y * c[2], mcu_x * c[1], 64), np.int16) for c in j["comps"]]
    mcu = 0  # MCUs are coded in raster order whatever the loop shape (mcu.rs:145-152 only reshapes the loops)
    for _ in range(height):
        for _ in range(bias):
            n = 0
            mcu_pass0 = mcu
            for _ in range(width):
                mx, my = mcu % mcu_x, mcu // mcu_x
                for ci, (cid, h, v) in enumerate(j["comps"]):
                    td, ta = j["sel"][cid]
                    for b in range(h * v):
                        if s.marker is not None and first_marker is None:
                            first_marker = nblk
                        nblk += 1
                        if s.bl < 16:
                            s.refill()
                        size = s.symbol(tabs[(0, td)])
                        if size:
                            is_short = s.bl < size and s.marker is None
                            zero_bits = _zero_model_bits(s, size) if is_short else None
                            bits = s.get(size)
                            if is_short:
                                short += 1
                                stale += bits != zero_bits
                            pred[ci] = (pred[ci] + _extend(bits, size)) & 0xFFFFFFFF  # wrapping_add on i32
                        if values:
                            blk = planes[ci][my * v + b // h, mx * h + b % h]
                            blk[:] = 0
                            blk[0] = np.int16(np.uint16(pred[ci] & 0xFFFF))
                        pos = 1
                        while pos < 64:
                            s.refill()
                            rs = s.symbol(tabs[(1, ta)])
                            r, size = rs >> 4, rs & 15
                            if size:
                                pos += r
                                fast = False
                                if s.last_len + size <= 9:
                                    # the fast-AC table (src/huffman.rs:236-251, src/bitstream.rs:339-347) holds every code +
                                    # magnitude that fits the 9 look-ahead bits AND whose value fits a byte
                                    fast = -128 <= _extend(s.peek(size), size) <= 127
                                if fast:
                                    # ONE drop_bits for both, zeros enter aligned_buffer; only the general path (:350-362)
                                    # rotates with get_bits
                                    bits = s.peek(size)
                                    s.drop(size)
                                else:
                                    bits = s.get(size)
                                if values:
                                    # a damaged stream can push pos past 63: the fast path writes at min(pos, 63) (:343),
                                    # the general path at pos & 63 (:359)
                                    # ... and the fast table keeps six bits of the value: `k << 10` into an i16
                                    # (src/huffman.rs:248), read back with `>> 10` (src/bitstream.rs:343)
                                    val = _extend(bits, size)
                                    blk[_UNZ[min(pos, 63) if fast else pos & 63]] = _sext6(val) if fast else val
                                pos += 1
                            elif s.last_len <= 9:
                                # size 0 with a code of up to 9 bits is in the fast table too (src/huffman.rs:217-233): run 0
                                # (EOB) counts as 63, and ANY other run -- 15 (ZRL) and the 1..14 that are not baseline
                                # symbols alike -- skips run + 1 coefficients after "writing" a zero
                                pos += 63 if r == 0 else r
                                if values:
                                    blk[_UNZ[min(pos, 63)]] = 0
                                pos += 1
                            elif r != 15:
                                break  # the general path (:365-367): anything but ZRL ends the block
                            else:
                                pos += 16
                n += 1
                mcu += 1
                todo -= 1
                if todo == 0:  # handle_rst
                    todo = j["ri"]
                    if s.marker is not None and 0xD0 <= s.marker <= 0xD7:
                        s.reset()
                        pred = [0] * ncomp  # mcu.rs:401-404
                if s.marker == 0xD9:
                    break
            loops.append(n)
            mcu = mcu_pass0 + width  # the MCUs a cut pass never decoded keep their zeros; the next pass starts behind them
    rows = [None] * mcu_y
    if width == 2 * mcu_x:  # one loop pass covers two MCU rows
        for i, n in enumerate(loops):
            rows[2 * i] = min(n, mcu_x)
            rows[2 * i + 1] = max(0, n - mcu_x)
    else:
        for i, n in enumerate(loops):
            rows[i] = n
    last_stats["short"], last_stats["stale"], last_stats["first_marker_block"] = short, stale, first_marker
    return rows, ([p.reshape(-1) for p in planes] if values else None), short


def decode_progressive_dc_first(jpeg_bytes):
    """The FIRST scan of a progressive file when it is a DC scan (Ss = 0, Ah = 0), the way the reference reads it: a run of
    decode_dc calls (src/bitstream.rs:407-415 -> :264-296), each refilling only below 16 bits -- so DC symbols longer than
    what is left are read short here too, far more often than in baseline scans (nothing but DC symbols moves bits_left).
    Returns ([per component: 2-D int16 array [block row][block column] of DC coefficients after the point transform],
    short_reads).  Follows src/mcu_prog.rs:262-430 for the block order (one component: the blocks that cover the image;
    several: MCU-interleaved)."""
    import numpy as np
    buf = jpeg_bytes
    p, dht, comps, ri = 2, {}, [], 0
    while True:
        assert buf[p] == 0xFF
        m = buf[p + 1]
        p += 2
        if m in (0xD8, 0xFF):
            continue
        length = struct.unpack(">H", buf[p:p + 2])[0]
        seg = buf[p + 2:p + length]
        if m == 0xC4:
            q = 0
            while q < len(seg):
                tc, th = seg[q] >> 4, seg[q] & 15
                counts = list(seg[q + 1:q + 17])
                n = sum(counts)
                dht[(tc, th)] = _canonical(counts, list(seg[q + 17:q + 17 + n]))
                q += 17 + n
        elif m == 0xC2:
            h, w, nc = struct.unpack(">HHB", seg[1:6])
            comps = [[seg[6 + 3 * i], seg[7 + 3 * i] >> 4, seg[7 + 3 * i] & 15] for i in range(nc)]
        elif m == 0xDD:
            ri = struct.unpack(">H", seg)[0]
        elif m == 0xDA:
            ns = seg[0]
            scan = [(seg[1 + 2 * i], seg[2 + 2 * i] >> 4) for i in range(ns)]
            ss, se, ah, al = seg[1 + 2 * ns], seg[2 + 2 * ns], seg[3 + 2 * ns] >> 4, seg[3 + 2 * ns] & 15
            assert ss == 0 and se == 0 and ah == 0, "not a DC-first scan"
            start = p + length
            break
        p += length
    hmax, vmax = max(c[1] for c in comps), max(c[2] for c in comps)
    if len(comps) == 1:
        hmax = vmax = 1
        comps[0][1] = comps[0][2] = 1
    mcu_x, mcu_y = (w + 8 * hmax - 1) // (8 * hmax), (h + 8 * vmax - 1) // (8 * vmax)
    by_id = {c[0]: i for i, c in enumerate(comps)}
    out = [np.zeros((mcu_y * c[2], mcu_x * c[1]), np.int16) for c in comps]
    s = _Reader(buf, start)
    pred, short = [0] * len(comps), 0
    last_stats["stale"] = 0
    todo = ri if ri else 1 << 62

    def dc(ci, td):
        nonlocal short
        if s.bl < 16:
            s.refill()
        size = s.symbol(dht[(0, td)])
        if size:
            is_short = s.bl < size and s.marker is None
            zero_bits = _zero_model_bits(s, size) if is_short else None
            bits = s.get(size)
            if is_short:
                short += 1
                last_stats["stale"] += bits != zero_bits
            pred[ci] = (pred[ci] + _extend(bits, size)) & 0xFFFFFFFF
        return np.int16(np.uint16((pred[ci] << al) & 0xFFFF))       # (dc_pred as i16).wrapping_mul(1 << al), :413

    def rst():
        nonlocal todo, pred
        todo -= 1
        if todo == 0:
            todo = ri
            if s.marker is not None and 0xD0 <= s.marker <= 0xD7:
                s.reset()
                pred = [0] * len(comps)

    if ns == 1:
        ci = by_id[scan[0][0]]
        if ci == 0 or (hmax == 1 and vmax == 1):
            bw, bh = (w + 7) // 8, (h + 7) // 8
        else:
            bw, bh = mcu_x, mcu_y
        for y in range(bh):
            for x in range(bw):
                out[ci][y, x] = dc(ci, scan[0][1])
                rst()
    else:
        for my in range(mcu_y):
            for mx in range(mcu_x):
                for cid, td in scan:
                    ci = by_id[cid]
                    for v in range(comps[ci][2]):
                        for hh in range(comps[ci][1]):
                            out[ci][my * comps[ci][2] + v, mx * comps[ci][1] + hh] = dc(ci, td)
                rst()
    last_stats["short"] = short
    return out, short
