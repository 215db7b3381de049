#!/usr/bin/env python3
"""
Minimal JPEG *encoder over given quantized coefficients* -- TEST INFRASTRUCTURE for the CPU entropy
front-end (zune-jpeg_amd/csrc/zj_jpeg.cpp).  It writes exactly the coefficient planes it is given
(no DCT, no quantisation), so that decode(encode(planes)) == planes is a bit-exact round trip, and
so that Pillow/libjpeg can decode the same file as an independent cross-check.

Supports: baseline (SOF0, interleaved scan, optional restart interval) and progressive (SOF2, the
classic 10-scan script incl. successive approximation), 1 or 3 components, luma sampling (1,1),
(2,1), (1,2), (2,2).  Huffman tables are fixed-length (every symbol present), which is legal and
keeps the encoder tiny.  Written from ITU-T T.81 (Annex F, G), pure Python: use small images.

Plane layout = the decoder's: per component [block_row][block_col][64] int16, natural order.
"""
import struct

import numpy as np

ZIGZAG = [0, 1, 8, 16, 9, 2, 3, 10, 17, 24, 32, 25, 18, 11, 4, 5, 12, 19, 26, 33, 40, 48, 41, 34, 27, 20, 13, 6, 7,
          14, 21, 28, 35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23, 30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46,
          53, 60, 61, 54, 47, 55, 62, 63]


class BitWriter:
    def __init__(self):
        self.out = bytearray()
        self.acc = 0
        self.n = 0

    def put(self, value, nbits):
        if nbits == 0:
            return
        self.acc = (self.acc << nbits) | (value & ((1 << nbits) - 1))
        self.n += nbits
        while self.n >= 8:
            b = (self.acc >> (self.n - 8)) & 0xFF
            self.out.append(b)
            if b == 0xFF:
                self.out.append(0)
            self.n -= 8
        self.acc &= (1 << self.n) - 1 if self.n else 0

    def flush(self):
        if self.n:
            self.put((1 << (8 - self.n)) - 1, 8 - self.n)  # pad with ones


def _nbits(v):
    v = abs(int(v))
    return v.bit_length()


def _dc_code(sym):   # 16 symbols, 5-bit codes 0..15 (Kraft sum 1/2, no all-ones code)
    return sym, 5


def _ac_code(sym):   # 256 symbols: 255 nine-bit codes + one ten-bit code (a count byte holds <= 255)
    return (sym, 9) if sym < 255 else (510, 10)


def canonical_tables(dc_lengths, ac_lengths):
    """Custom Huffman tables for encode_baseline(tables=...): {symbol: code length} per class -> canonical codes
    (T.81 Annex C) and the DHT payload.  The same tables serve every component.  For tests that need particular code
    lengths (e.g. a 16-bit DC code in front of 11 magnitude bits)."""
    out = {}
    for name, lens in (("dc", dc_lengths), ("ac", ac_lengths)):
        order = sorted(lens, key=lambda sym: (lens[sym], sym))
        counts = [0] * 16
        codes, code, prev = {}, 0, 0
        for sym in order:
            length = lens[sym]
            code <<= length - prev
            prev = length
            assert code < (1 << length) - 1 or (length == 16 and code < (1 << 16) - 1), "all-ones code / over-subscribed lengths"
            codes[sym] = (code, length)
            counts[length - 1] += 1
            code += 1
        out[name] = {"codes": codes, "counts": counts, "symbols": order}
    return out


def _dht_segment(tables=None):
    if tables is not None:
        seg = bytearray()
        for cls, name in ((0, "dc"), (1, "ac")):
            t = tables[name]
            for idx in (0, 1):
                seg += bytes([(cls << 4) | idx]) + bytes(t["counts"]) + bytes(t["symbols"])
        return b"\xff\xc4" + struct.pack(">H", len(seg) + 2) + bytes(seg)
    seg = bytearray()
    for cls, nsym, length in ((0, 16, 5), (1, 256, 9)):
        counts = [0] * 16
        counts[length - 1] = min(nsym, 255)
        if nsym == 256:
            counts[length] = 1
        for idx in (0, 1):
            seg += bytes([(cls << 4) | idx]) + bytes(counts) + bytes(range(nsym))
    return b"\xff\xc4" + struct.pack(">H", len(seg) + 2) + bytes(seg)


def _geometry(w, h, hs, vs):
    mcu_x = (w + 8 * hs - 1) // (8 * hs)
    mcu_y = (h + 8 * vs - 1) // (8 * vs)
    return mcu_x, mcu_y


def _headers(w, h, hs, vs, ncomp, qts, progressive, restart, tables=None):
    out = bytearray(b"\xff\xd8")
    out += b"\xff\xe0" + struct.pack(">H", 16) + b"JFIF\x00\x01\x01\x00\x00\x01\x00\x01\x00\x00"
    ntab = 1 if ncomp == 1 else 2
    for t in range(ntab):
        q = np.asarray(qts[min(t, len(qts) - 1)], np.int32).reshape(64)
        out += b"\xff\xdb" + struct.pack(">H", 67) + bytes([t]) + bytes(int(q[ZIGZAG[i]]) for i in range(64))
    out += (b"\xff\xc2" if progressive else b"\xff\xc0") + struct.pack(">HBHHB", 8 + 3 * ncomp, 8, h, w, ncomp)
    for c in range(ncomp):
        samp = (hs << 4) | vs if c == 0 else 0x11
        out += bytes([c + 1, samp, 0 if c == 0 else 1])
    out += _dht_segment(tables)
    if restart:
        out += b"\xff\xdd" + struct.pack(">HH", 4, restart)
    return out


def _sos(comps, ss, se, ah, al):
    seg = bytes([len(comps)])
    for c in comps:
        t = 0 if c == 0 else 1
        seg += bytes([c + 1, (t << 4) | t])
    seg += bytes([ss, se, (ah << 4) | al])
    return b"\xff\xda" + struct.pack(">H", len(seg) + 2) + seg


class _Planes:
    def __init__(self, planes, w, h, hs, vs, ncomp):
        self.mcu_x, self.mcu_y = _geometry(w, h, hs, vs)
        self.hv = [(hs, vs) if c == 0 else (1, 1) for c in range(ncomp)]
        self.bw = [self.mcu_x * self.hv[c][0] for c in range(ncomp)]
        self.blocks = []
        for c in range(ncomp):
            bh = self.mcu_y * self.hv[c][1]
            self.blocks.append(np.asarray(planes[c], np.int16).reshape(bh, self.bw[c], 64).astype(np.int64))
        self.w, self.h = w, h

    def mcu_blocks(self, comps):
        """yields lists of (comp, by, bx) per MCU for an interleaved scan"""
        for my in range(self.mcu_y):
            for mx in range(self.mcu_x):
                lst = []
                for c in comps:
                    hs, vs = self.hv[c]
                    for v in range(vs):
                        for hh in range(hs):
                            lst.append((c, my * vs + v, mx * hs + hh))
                yield lst

    def single_blocks(self, c):
        """non-interleaved scan of one component: only blocks covering the image (T.81 A.2.3)"""
        hs, vs = self.hv[c]
        hmax, vmax = self.hv[0]
        cw = -(-self.w * hs // hmax)
        ch = -(-self.h * vs // vmax)
        for by in range(-(-ch // 8)):
            for bx in range(-(-cw // 8)):
                yield (c, by, bx)


def _emit_restart(bw, count):
    bw.flush()
    bw.out += bytes([0xFF, 0xD0 + (count & 7)])


def encode_baseline(planes, qts, w, h, hs=1, vs=1, ncomp=3, restart=0, tables=None):
    P = _Planes(planes, w, h, hs, vs, ncomp)
    out = _headers(w, h, hs, vs, ncomp, qts, False, restart, tables)
    dc_code = (lambda sym: tables["dc"]["codes"][sym]) if tables else _dc_code
    ac_code = (lambda sym: tables["ac"]["codes"][sym]) if tables else _ac_code
    out += _sos(list(range(ncomp)), 0, 63, 0, 0)
    bw = BitWriter()
    pred = [0] * ncomp
    n_mcu, rst = 0, 0
    for mcu in P.mcu_blocks(list(range(ncomp))):
        if restart and n_mcu and n_mcu % restart == 0:
            _emit_restart(bw, rst)
            rst += 1
            pred = [0] * ncomp
        for c, by, bx in mcu:
            blk = P.blocks[c][by, bx]
            diff = int(blk[0]) - pred[c]
            pred[c] = int(blk[0])
            s = _nbits(diff)
            bw.put(*dc_code(s))
            if s:
                bw.put(diff if diff >= 0 else diff + (1 << s) - 1, s)
            run = 0
            for k in range(1, 64):
                v = int(blk[ZIGZAG[k]])
                if v == 0:
                    run += 1
                    continue
                while run > 15:
                    bw.put(*ac_code(0xF0))
                    run -= 16
                s = _nbits(v)
                bw.put(*ac_code((run << 4) | s))
                bw.put(v if v >= 0 else v + (1 << s) - 1, s)
                run = 0
            if run:
                bw.put(*ac_code(0))
        n_mcu += 1
    bw.flush()
    return bytes(out) + bytes(bw.out) + b"\xff\xd9"


class _ProgAC:
    """AC scans of one component (T.81 G.1.2.2 / G.1.2.3) with EOBRUN and correction-bit buffering."""

    def __init__(self, bw):
        self.bw = bw
        self.eobrun = 0
        self.be = []  # buffered correction bits of the pending EOB run

    def flush_eobrun(self):
        if self.eobrun:
            n = self.eobrun.bit_length() - 1
            self.bw.put(*_ac_code(n << 4))
            if n:
                self.bw.put(self.eobrun & ((1 << n) - 1), n)
            self.eobrun = 0
        for b in self.be:
            self.bw.put(b, 1)
        self.be = []

    def first(self, blk, ss, se, al):
        run = 0
        for k in range(ss, se + 1):
            v = int(blk[ZIGZAG[k]])
            a = abs(v) >> al
            if a == 0:
                run += 1
                continue
            self.flush_eobrun()
            while run > 15:
                self.bw.put(*_ac_code(0xF0))
                run -= 16
            s = a.bit_length()
            self.bw.put(*_ac_code((run << 4) | s))
            self.bw.put(a if v >= 0 else (~a) & ((1 << s) - 1), s)
            run = 0
        if run:
            self.eobrun += 1
            if self.eobrun == 0x7FFF:
                self.flush_eobrun()

    def refine(self, blk, ss, se, al):
        absv = [abs(int(blk[ZIGZAG[k]])) >> al for k in range(64)]
        eob = 0
        for k in range(ss, se + 1):
            if absv[k] == 1:
                eob = k  # last newly-nonzero coefficient
        run = 0
        br = []  # correction bits since the last emitted symbol
        for k in range(ss, se + 1):
            a = absv[k]
            if a == 0:
                run += 1
                continue
            while run > 15 and k <= eob:
                self.flush_eobrun()
                self.bw.put(*_ac_code(0xF0))
                run -= 16
                for b in br:
                    self.bw.put(b, 1)
                br = []
            if a > 1:
                br.append(a & 1)  # already non-zero: one correction bit
                continue
            self.flush_eobrun()
            self.bw.put(*_ac_code((run << 4) | 1))
            self.bw.put(1 if int(blk[ZIGZAG[k]]) >= 0 else 0, 1)
            for b in br:
                self.bw.put(b, 1)
            br = []
            run = 0
        if run > 0 or br:
            self.eobrun += 1
            self.be += br
            if self.eobrun == 0x7FFF or len(self.be) > 900:
                self.flush_eobrun()


def encode_progressive(planes, qts, w, h, hs=1, vs=1, ncomp=3, restart=0):
    P = _Planes(planes, w, h, hs, vs, ncomp)
    out = bytearray(_headers(w, h, hs, vs, ncomp, qts, True, 0))
    allc = list(range(ncomp))
    if ncomp == 3:
        script = [(allc, 0, 0, 0, 1), ([0], 1, 5, 0, 2), ([2], 1, 63, 0, 1), ([1], 1, 63, 0, 1), ([0], 6, 63, 0, 2),
                  ([0], 1, 63, 2, 1), (allc, 0, 0, 1, 0), ([2], 1, 63, 1, 0), ([1], 1, 63, 1, 0), ([0], 1, 63, 1, 0)]
    else:
        script = [([0], 0, 0, 0, 1), ([0], 1, 5, 0, 2), ([0], 6, 63, 0, 2), ([0], 1, 63, 2, 1), ([0], 0, 0, 1, 0),
                  ([0], 1, 63, 1, 0)]
    for comps, ss, se, ah, al in script:
        out += _sos(comps, ss, se, ah, al)
        bw = BitWriter()
        if ss == 0:  # DC scan
            pred = [0] * ncomp
            units = P.mcu_blocks(comps) if len(comps) > 1 else ([b] for b in P.single_blocks(comps[0]))
            for mcu in units:
                for c, by, bx in mcu:
                    v = int(P.blocks[c][by, bx][0])
                    if ah == 0:
                        t = v >> al
                        diff = t - pred[c]
                        pred[c] = t
                        s = _nbits(diff)
                        bw.put(*_dc_code(s))
                        if s:
                            bw.put(diff if diff >= 0 else diff + (1 << s) - 1, s)
                    else:
                        bw.put((v >> al) & 1, 1)
        else:
            enc = _ProgAC(bw)
            for c, by, bx in P.single_blocks(comps[0]):
                blk = P.blocks[c][by, bx]
                if ah == 0:
                    enc.first(blk, ss, se, al)
                else:
                    enc.refine(blk, ss, se, al)
            enc.flush_eobrun()
        bw.flush()
        out += bw.out
    return bytes(out) + b"\xff\xd9"


def small_planes(w, h, hs, vs, ncomp, seed=0, amp=40, dc=300):
    """Random sparse coefficient planes whose padding blocks (outside the image) are zero, so that
    progressive non-interleaved scans (which skip them) round-trip exactly."""
    rng = np.random.default_rng(seed)
    mcu_x, mcu_y = _geometry(w, h, hs, vs)
    planes = []
    for c in range(ncomp):
        ch, cv = (hs, vs) if c == 0 else (1, 1)
        bw, bh = mcu_x * ch, mcu_y * cv
        p = rng.integers(-amp, amp + 1, size=(bh, bw, 64))
        p[rng.random((bh, bw, 64)) < 0.75] = 0
        p[:, :, 0] = rng.integers(-dc, dc, size=(bh, bw))
        p[rng.random((bh, bw)) < 0.2, 1:] = 0
        cw, chh = -(-w * ch // hs), -(-h * cv // vs)
        p[-(-chh // 8):, :, :] = 0
        p[:, -(-cw // 8):, :] = 0
        planes.append(p.astype(np.int16).reshape(-1))
    return planes
