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

Coefficient VALUES are not modelled (the front-end's values are covered by tests/test_jpeg_frontend.py); only code
lengths and magnitude-bit counts matter for the walk.
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


class _Reader:
    """src/bitstream.rs BitStream: `buffer`/`bits_left` bookkeeping and the marker logic of refill()"""

    def __init__(self, buf, pos):
        self.buf, self.pos, self.buffer, self.bl, self.marker = buf, pos, 0, 0, None

    def _byte(self):  # read_u8 (:689-703): zeros past the end
        v = self.buf[self.pos] if self.pos < len(self.buf) else 0
        self.pos += 1
        return v

    def refill(self):
        if self.bl <= 32 and self.marker is None:
            if self.pos + 4 < len(self.buf) and 0xFF not in self.buf[self.pos:self.pos + 4]:  # :220-236
                self.buffer = (self.buffer << 32) | int.from_bytes(self.buf[self.pos:self.pos + 4], "big")
                self.pos += 4
                self.bl += 32
                return
            for _ in range(4):  # the refill! macro, :168-213
                b = self._byte()
                self.buffer = (self.buffer << 8) | b
                self.bl += 8
                if b == 0xFF:
                    n = self._byte()
                    if n != 0:
                        while n == 0xFF:
                            n = self._byte()
                        if n != 0:
                            self.buffer >>= 8
                            self.bl -= 8
                            self.marker = n
                            return
        elif self.marker is not None:  # :254-258: zeros from here on
            if self.bl < 63:
                self.buffer <<= 63 - self.bl
            self.bl = 63

    def peek(self, n):
        if self.bl >= n:
            return (self.buffer >> (self.bl - n)) & ((1 << n) - 1)
        return (self.buffer << (n - self.bl)) & ((1 << n) - 1)

    def drop(self, n):  # drop_bits / get_bits: saturating_sub on bits_left, zeros shift in
        if n > self.bl:
            self.buffer <<= n - self.bl
            self.bl = n
        self.bl -= n
        self.buffer &= (1 << self.bl) - 1

    def symbol(self, table):
        for length in range(1, 17):
            c = self.peek(length)
            if (length, c) in table:
                self.drop(length)
                return table[(length, c)]
        raise ValueError("bad Huffman code")

    def reset(self):  # :671-678
        self.buffer, self.bl, self.marker = 0, 0, None


def decoded_mcus_per_row(jpeg_bytes):
    """For every MCU row of the image (mcu_y rows of mcu_x MCUs, the front-end's geometry): how many leading MCUs the
    reference entropy-decodes.  Rows it never reaches at all (an odd last MCU row under (2,1)/(2,2) sampling) report
    None."""
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
    for _ in range(height):
        for _ in range(bias):
            n = 0
            for _ in range(width):
                for cid, h, v in j["comps"]:
                    td, ta = j["sel"][cid]
                    for _ in range(h * v):
                        if s.bl < 16:
                            s.refill()
                        size = s.symbol(tabs[(0, td)])
                        if size:
                            s.drop(size)
                        pos = 1
                        while pos < 64:
                            s.refill()
                            rs = s.symbol(tabs[(1, ta)])
                            r, size = rs >> 4, rs & 15
                            if size:
                                pos += r + 1
                                s.drop(size)
                            elif r != 15:
                                break
                            else:
                                pos += 16
                n += 1
                todo -= 1
                if todo == 0:  # handle_rst
                    todo = j["ri"]
                    if s.marker is not None and 0xD0 <= s.marker <= 0xD7:
                        s.reset()
                if s.marker == 0xD9:
                    break
            loops.append(n)
    rows = [None] * mcu_y
    if width == 2 * mcu_x:  # one loop pass covers two MCU rows
        for i, n in enumerate(loops):
            rows[2 * i] = min(n, mcu_x)
            rows[2 * i + 1] = max(0, n - mcu_x)
    else:
        for i, n in enumerate(loops):
            rows[i] = n
    return rows
