// zj_jpeg.cpp -- the CPU side the north star keeps on the host: container parsing and the branchy
// Huffman bitstream (baseline + progressive), producing whole-image planes of quantized coefficients
// in the layout the GPU pixel path consumes, then handing them to zj_decode_planes.
//
// It plays the role of these reference files (paths relative to the zune-jpeg tree) and mirrors their
// observable behaviour, written from the JPEG standard (ITU-T T.81), not from their code:
//   src/decoder.rs:239-416   marker loop, supported schemes (SOF0 / SOF2, 8-bit, 1 or 3 components)
//   src/headers.rs:18-339    DHT / DQT / SOF / SOS / DRI
//   src/huffman.rs, src/bitstream.rs   Huffman tables, MSB-first bit reader with 0xFF00 stuffing
//   src/mcu.rs:127-380       baseline scan  -> planes (the strips the reference allocates, concatenated)
//   src/mcu_prog.rs:49-430   progressive multi-scan accumulation into whole-image planes
// Plane layout: per component [block_row][block_col][64] int16, natural order (un-zigzagged on write,
// src/bitstream.rs:343,359), block_cols = mcu_x * h_samp (width_stride / 8, src/headers.rs:338).
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/mman.h>
#include <unistd.h>
#include <immintrin.h> // _mm_stream_si128 and wider: the coefficient planes are written once and read by DMA

#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <functional>
#include <mutex>
#include <new>
#include <string>
#include <thread>
#include <vector>

#include "../../include/zjhip.h"
#include "zj_crew.h"
#include "zj_huff.h"

namespace {

const uint8_t kUnZigzag[64] = {0,  1,  8,  16, 9,  2,  3,  10, 17, 24, 32, 25, 18, 11, 4,  5,  12, 19, 26, 33, 40, 48,
                               41, 34, 27, 20, 13, 6,  7,  14, 21, 28, 35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23,
                               30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63};

constexpr int AC_BITS = 10;
inline int sext6(int v) { return (int)((unsigned)v << 26) >> 26; }
struct Huff {
    bool present = false;
    // canonical decoding (T.81 F.2.2.3) with an 9-bit lookahead table
    uint16_t look[512];    // (len << 8) | symbol, 0 = longer than 9 bits
    int16_t fast[512];     // AC shortcut: (value << 8) | (run << 4) | (code length + magnitude bits), 0 = none;
                           // set when code and magnitude bits both fit the 9-bit window and the value fits a byte
    int32_t maxcode[18];   // per length, -1 if none
    int32_t valoff[17];
    uint8_t vals[256];
    uint8_t nlen[17];      // codes per length, as the DHT segment gave them (for the GPU tables, gpu_table())
    // The baseline walker's AC table (decode_mcus_v2): one entry per AC_BITS-bit window for every code of up to AC_BITS bits.
    // Byte fields, each read by its own load (the loop is short of shift ports, not of load ports):
    //   total   what the entry consumes: code length + magnitude bits (+ the EOB code behind it, see fold); 0 = a longer code,
    //           see ac_escape()
    //   idxoff  what to add to the zig-zag position k to index kZZ (which coefficient the value goes to)
    //   kadv    what the entry adds to k
    //   len, sz code length and magnitude bits of the (first) symbol
    //   sx      26 where the reference's fast-AC table keeps only six bits of the value (see fast[] above), else 0
    //   t1, k1  what the (first) symbol alone consumes and adds to k: equal to total and kadv except in a folded entry, for the
    //           decoders that take one symbol at a time (decode_block_baseline, par_structure_run)
    // The value itself is computed from the bits (the arithmetic is off the critical path).
    struct AcEnt { uint8_t total, idxoff, kadv, len, sz, k1, sx, t1; };
    AcEnt actab[1 << AC_BITS];
    // The structure decode's table (par_structure_run: lengths and zig-zag advances only, round 6): what ALL the symbols
    // whose CODES lie inside the window consume together -- their magnitude bits may lie outside, nothing reads them --
    //   total, kadv  bits and zig-zag advance of the whole group (an EOB at its end: kadv reaches 64 from anywhere)
    //   kpre         the advance of all but the last symbol: if that already ends the block the group does not apply
    //   lastlen      bits of the group's last symbol (bits_left behind a block follows from where its last symbol began)
    //   nsym         symbols in the group (1: the window's first symbol alone); 0 = a code longer than the window (ac_escape)
    // Built on demand (build_groups): 8 KB per table, wanted by scan_baseline_parallel only.
    struct Group { uint8_t total, kadv, kpre, lastlen, nsym, pad[3]; };
    Group groups[1 << AC_BITS];
    bool have_groups = false;
    void build_groups()
    {
        if (have_groups) return;
        for (int w = 0; w < (1 << AC_BITS); w++) {
            Group g{};
            int pos = 0, kadv = 0, kpre = 0, nsym = 0, lastlen = 0;
            for (;;) {
                const AcEnt& e = actab[(w << pos) & ((1 << AC_BITS) - 1)];
                if (!e.total || e.len > AC_BITS - pos) break;        // (the code must lie inside the bits the window knows)
                if (pos + e.t1 > 250 || kadv + e.k1 > 250) break;
                kpre = kadv;
                kadv += e.k1;
                lastlen = e.t1;
                pos += e.t1;
                nsym++;
                if (e.k1 >= 64 || pos >= AC_BITS || nsym == 4) break; // (an EOB ends the group; the next code would begin outside)
            }
            if (nsym >= 1) { g.total = (uint8_t)pos; g.kadv = (uint8_t)kadv; g.kpre = (uint8_t)kpre; g.lastlen = (uint8_t)lastlen; g.nsym = (uint8_t)nsym; }
            groups[w] = g;
        }
        have_groups = true;
    }
    bool full_values = false; // ZJ_FLAG_FULL_AC_VALUES: fast-AC values are not cut to six bits
    int build(const uint8_t counts[17], const uint8_t* symbols, int nsym, std::string& err)
    {
        memcpy(nlen, counts, 17);
        have_groups = false;
        uint16_t codes[257];
        uint8_t sizes[257];
        int k = 0;
        for (int l = 1; l <= 16; l++)
            for (int i = 0; i < counts[l]; i++) sizes[k++] = (uint8_t)l;
        if (k != nsym) { err = "Bogus Huffman table definition"; return -1; }
        uint32_t code = 0;
        int si = k ? sizes[0] : 0, p = 0;
        while (p < k) {
            while (p < k && sizes[p] == si) codes[p++] = (uint16_t)code++;
            if (code > (1u << si)) { err = "Bad Huffman Table"; return -1; } // over-subscribed
            code <<= 1;
            si++;
        }
        memcpy(vals, symbols, (size_t)nsym);
        p = 0;
        for (int l = 1; l <= 16; l++) {
            if (counts[l]) {
                valoff[l] = p - (int)codes[p];
                p += counts[l];
                maxcode[l] = codes[p - 1];
            } else {
                maxcode[l] = -1;
                valoff[l] = 0;
            }
        }
        maxcode[17] = 0x7fffffff;
        memset(look, 0, sizeof look);
        p = 0;
        for (int l = 1; l <= 9; l++)
            for (int i = 0; i < counts[l]; i++, p++) {
                int base = codes[p] << (9 - l);
                for (int j = 0; j < (1 << (9 - l)); j++) look[base + j] = (uint16_t)((l << 8) | vals[p]);
            }
        for (int i = 0; i < 512; i++) {
            fast[i] = 0;
            const uint16_t e = look[i];
            if (!e) continue;
            const int len = e >> 8, run = (e & 0xff) >> 4, sz = e & 15;
            if (!sz || len + sz > 9) continue;
            const int bits = ((i << len) & 511) >> (9 - sz);
            const int v = bits < (1 << (sz - 1)) ? bits - (1 << sz) + 1 : bits; // T.81 F.2.2.1 EXTEND
            // The reference packs the value as `k << 10` into an i16 (src/huffman.rs:248) and reads it back with `>> 10`
            // (src/bitstream.rs:343): six bits of it survive, sign-extended.  Sizes up to 5 are unharmed; a size of 6..8 behind
            // a code of 1..3 bits (hand-made or heavily optimised tables only) comes back as sext6(v): 32 -> -32, 64 -> 0.
            // (ZJ_FLAG_FULL_AC_VALUES: the coded value)
            if (v >= -128 && v <= 127) fast[i] = (int16_t)(((full_values ? v : sext6(v)) * 256) | (run << 4) | (len + sz));
        }
        memset(actab, 0, sizeof actab);
        p = 0;
        for (int l = 1; l <= AC_BITS; l++)
            for (int i = 0; i < counts[l]; i++, p++) {
                const int base = codes[p] << (AC_BITS - l);
                for (int j = 0; j < (1 << (AC_BITS - l)); j++) {
                    const int w = base + j;
                    actab[w] = ac_entry(vals[p], l, l <= 9 && fast[w >> (AC_BITS - 9)] != 0);
                    if (full_values) actab[w].sx = 0;
                }
            }
        // A coefficient followed by the end-of-block symbol, both inside the window: one entry for the pair (fold) -- it
        // consumes both codes and ends the loop (k += 64).  decode_mcus_v2 undoes the second half when the coefficient
        // itself was the block's last (k reached 64: no EOB follows in the stream).
        {
            static_assert(sizeof(AcEnt) == 8, "AcEnt");
            AcEnt folded[1 << AC_BITS];
            for (int w = 0; w < (1 << AC_BITS); w++) {
                const AcEnt e1 = actab[w];
                folded[w] = e1;
                if (!e1.total || !e1.sz || e1.total >= AC_BITS) continue;
                const AcEnt e2 = actab[(w << e1.total) & ((1 << AC_BITS) - 1)];
                if (!e2.total || e2.sz || e2.kadv != 64 || e2.len > AC_BITS - e1.total) continue;
                folded[w].total = (uint8_t)(e1.total + e2.len);
                folded[w].kadv = 64;
                // (t1 and k1 stay the coefficient's own)
            }
            memcpy(actab, folded, sizeof actab);
        }
        present = true;
        return 0;
    }
    // The entry of AC symbol `rs` with a code of `len` bits, as src/bitstream.rs:332-372 treats it; ref_fast: the reference
    // takes it through its fast-AC table (code + magnitude within 9 bits and a value that fits a byte, src/huffman.rs:236-251).
    //   * a coefficient (size > 0): k += run, the value goes to zig-zag position k, k += 1.  When a damaged stream pushes k
    //     past 63 the general path writes at k & 63 (:359), the fast path at min(k, 63) (:343): kZZ[k + run] and
    //     kZZ[128 + k + run] are those two rules.
    //   * size 0 with a code of up to 9 bits: the reference's fast table has an entry for these too (src/huffman.rs:217-233):
    //     k += run (63 for run 0: the end of the block), a zero is written at min(k, 63) -- a coefficient that is still zero --
    //     and k += 1.  So a run of 1..14 with size 0 (not a baseline symbol) SKIPS run + 1 coefficients there.
    //   * size 0 with a longer code (the general path, :365-371): run 15 skips 16, anything else ends the block.
    // Zeros "written" for size 0 go to natural position 63, which is zero for as long as the loop runs.
    static AcEnt ac_entry(int rs, int len, bool ref_fast)
    {
        const int run = rs >> 4, sz = rs & 15;
        int idxoff, kadv;
        if (sz) { idxoff = ref_fast ? 128 + run : run; kadv = run + 1; }
        else {
            idxoff = 192; // k + 192 >= 128 + 63: natural position 63
            if (len <= 9) kadv = (run == 0 ? 63 : run) + 1;
            else kadv = run == 15 ? 16 : 64;
        }
        AcEnt e;
        memset(&e, 0, sizeof e);
        e.total = (uint8_t)(len + sz); e.idxoff = (uint8_t)idxoff; e.kadv = (uint8_t)kadv; e.len = (uint8_t)len; e.sz = (uint8_t)sz;
        e.t1 = e.total; e.k1 = e.kadv;
        e.sx = ref_fast && sz >= 6 ? 26 : 0;
        return e;
    }
    // a code of more than AC_BITS bits at the top of acc: its entry, or 0 if there is no such code
    __attribute__((noinline)) AcEnt ac_escape(uint64_t acc) const
    {
        const uint32_t code = (uint32_t)(acc >> 48);
        for (int l = AC_BITS + 1; l <= 16; l++) {
            const int32_t c = (int32_t)(code >> (16 - l));
            if (c <= maxcode[l]) return ac_entry(vals[(c + valoff[l]) & 0xff], l, false);
        }
        AcEnt none;
        memset(&none, 0, sizeof none);
        return none;
    }
};

// kZZ[k + entry's offset] -> natural index of the coefficient (Huff::ac_entry)
struct ZigzagPad {
    uint8_t t[256];
    ZigzagPad()
    {
        for (int i = 0; i < 128; i++) t[i] = kUnZigzag[i & 63];                         // general path: k & 63
        for (int i = 128; i < 256; i++) t[i] = kUnZigzag[i - 128 < 63 ? i - 128 : 63];  // fast path: min(k, 63)
    }
};
const ZigzagPad kZZ;
const uint32_t kSizeMask[16] = {0, 1, 3, 7, 15, 31, 63, 127, 255, 511, 1023, 2047, 4095, 8191, 16383, 32767};

struct BitReader {
    const uint8_t* p;
    const uint8_t* end;
    uint64_t acc = 0; // bits left-aligned
    int nbits = 0;
    int marker = 0;   // pending marker byte (0xD0.., 0xD9 ...) found in the entropy-coded data
    // bookkeeping for reference_saw_eoi(): the restart interval's first byte, stuffed zeros skipped since then, the
    // first 0xFF of the pending marker, zero bits appended after it (or after the end of the data), and the length in
    // bits of the last symbol a block decoded
    const uint8_t* istart = nullptr;
    const uint8_t* mpos = nullptr;
    uint32_t stuffed = 0;
    int pad = 0;
    int last_sym = 0;
    // the REFERENCE reader's `bits_left` (src/bitstream.rs:117), followed symbol by symbol as long as no marker has come
    // into its view: refill() adds 32 when it is called with bits_left <= 32 (before every AC symbol) or < 16 (before a DC
    // symbol, :278).  It exists for one case only, ref_dc_misread() below.
    int rbl = 0;
    // ... and what lies BELOW the valid bits in the reference's aligned_buffer (round 4).  A real refill leaves 64 - rbl0
    // zero bits there (rbl0 = bits_left right after it); from then on drop_bits shifts zeros in at the bottom and
    // get_bits ROTATES (src/bitstream.rs:394-402): the magnitude bits it hands out re-enter at the bottom.  rhist is that
    // history as a shift register (newest at the bottom; a fast-AC symbol is one drop_bits, src/bitstream.rs:339-347): the
    // low (rbl0 - bits_left) bits of it are what the reference holds under its zero gap.  Only ref_dc_misread() reads it.
    int rbl0 = 0;
    uint64_t rhist = 0;
    long long misreads = 0; // DC symbols the reference reads short (ref_dc_misread)
    void reset() { acc = 0; nbits = 0; marker = 0; istart = p; mpos = nullptr; stuffed = 0; pad = 0; rbl = 0; rbl0 = 0; rhist = 0; }
    // bits consumed since istart (zero padding included once the real bits are used up)
    long long consumed() const
    {
        const long long fed = (long long)((marker ? mpos : p) - istart) - (long long)stuffed;
        return 8 * fed - ((long long)nbits - pad);
    }
    void fill()
    {
        // eight bytes at once while no 0xFF (stuffing or marker) is among them; the bits of a partially taken
        // byte are OR-ed in again, unchanged, by the next fill
        if (!marker && end - p >= 8 && nbits >= 0 && nbits <= 56) {
            uint64_t x;
            memcpy(&x, p, 8);
            x = __builtin_bswap64(x);
            const uint64_t y = ~x;
            if (!((y - 0x0101010101010101ull) & ~y & 0x8080808080808080ull)) {
                acc |= x >> nbits;
                p += (63 - nbits) >> 3;
                nbits |= 56;
                return;
            }
        }
        while (nbits <= 56) {
            uint32_t b = 0;
            if (!marker && p < end) {
                b = *p++;
                if (b == 0xFF) {
                    uint32_t n = p < end ? *p : 0xD9;
                    if (n == 0) { p++; stuffed++; }       // stuffed zero
                    else {
                        const uint8_t* const first = p - 1;
                        uint32_t fills = 0;
                        while (n == 0xFF && p + 1 < end) { p++; n = *p; fills++; }
                        if (n == 0 && fills) {
                            // 0xFF fill bytes in front of a ZERO: the reference's refill skips them, finds no marker and keeps
                            // the 0xFF it has already appended as a data byte (src/bitstream.rs:183-211) -- FF FF 00 reads like
                            // FF 00.  (Until round 6 this appended a zero byte and counted it as padding; found while reading
                            // the refill for the restart-segment findings of tools/stream_soak.py.)
                            p++;
                            stuffed += fills + 1;
                        } else {                          // marker: stop feeding, pad with zeros
                            mpos = first;
                            marker = (int)n;
                            if (p < end) p++;
                            b = 0;
                            pad += 8;
                        }
                    }
                }
            } else {
                pad += 8; // zeros after a marker or after the end of the data
            }
            acc |= (uint64_t)b << (56 - nbits);
            nbits += 8;
        }
    }
    inline uint32_t peek(int n) { return (uint32_t)(acc >> (64 - n)); }
    inline void drop(int n) { acc <<= n; nbits -= n; }
    inline int32_t get(int n)
    {
        if (n == 0) return 0;
        if (nbits < n) fill();
        uint32_t v = peek(n);
        drop(n);
        return (int32_t)v;
    }
    inline int decode(const Huff& h)
    {
        if (nbits < 16) fill();
        uint32_t v = peek(9);
        uint16_t e = h.look[v];
        if (e) { drop(e >> 8); return e & 0xff; }
        uint32_t code = peek(16);
        for (int l = 10; l <= 16; l++) {
            int32_t c = (int32_t)(code >> (16 - l));
            if (c <= h.maxcode[l]) { drop(l); return h.vals[(c + h.valoff[l]) & 0xff]; }
        }
        drop(16);
        return -1;
    }
};

inline int32_t extend(int32_t v, int s) { return v < (1 << (s - 1)) ? v - (1 << s) + 1 : v; } // T.81 F.2.2.1

// ---- the reference's short read of long DC symbols (src/bitstream.rs:264-296) ---------------------------------------
// decode_dc refills only when bits_left < 16, but a DC symbol is a code of up to 16 bits PLUS up to 11 (16) magnitude
// bits.  With 16 <= bits_left < code + magnitude the reference decodes the code, then get_bits() rotates into the
// magnitude's low end whatever aligned_buffer holds below its bits_left valid bits, bits_left saturates at 0 -- and the
// bits it did not have are never skipped: the next refill continues right behind the last LOADED bit, so the rest of
// the magnitude is parsed as the next symbol.  Everything after that is garbage, but it is the reference's garbage, and
// "same bytes as the reference" includes it.  Standard tables reach 18 / 20 bits for |DC difference| >= 512 / 1024; in a
// baseline scan the case needs the previous block to end in a symbol of 14 bits or more, in a progressive DC scan
// (nothing but DC symbols, so bits_left wanders through 16..47) it is common once such differences occur.
// What lies below the valid bits: the zeros the last real refill left (64 - rbl0 of them), then the history of every
// drop_bits (zeros) and get_bits (the magnitude bits themselves, rotated in) since that refill -- BitReader::rhist.
// Mostly the read ends inside the zeros; when the last refill came at bits_left near 32 (a gap of few zeros) and was
// followed by a short code whose magnitude went through get_bits, it reaches that magnitude's stale bits (ADVICE r3;
// tests/test_jpeg_frontend.py::test_short_dc_read_picks_up_stale_rotated_bits).
// Called after the DC code (`len` bits) has been dropped and entered into `hist`, with rbl already raised by the < 16
// refill: true if the reference reads the `s` magnitude bits short; then *bits is what it gets and the reader has
// consumed what it had.
// need_hist (non-null from a caller that does not track rbl0 / the history -- the default: tracking costs the Huffman loop
// 9-13 %): a short read is only REPORTED; the caller then decodes the image again with tracking on.
inline bool ref_dc_misread(BitReader& br, int& rbl, const int rbl0, const uint64_t hist, const int len, const int s, int32_t* bits,
                           bool* need_hist = nullptr)
{
    if (len + s <= rbl) return false;
    // rbl describes the reference only while it has not come across the marker that ends the interval (after that it
    // holds every remaining bit and serves zeros): it has read (consumed + rbl) / 8 data bytes
    const uint8_t* m = br.marker ? br.mpos : nullptr;
    long long st = br.stuffed;
    if (!m) {
        for (const uint8_t* q = br.p; q + 1 < br.end; q++)
            if (q[0] == 0xFF) { if (q[1] == 0x00) { st++; q++; } else { m = q; break; } }
        if (!m) m = br.end;
    }
    const long long data_bits = 8 * ((long long)(m - br.istart) - st);
    const long long c_code_start = br.consumed() - len;
    if (c_code_start + rbl > data_bits) return false;
    const int avail = rbl - len; // 0 <= avail < s (len <= 16 <= rbl)
    if (need_hist) { *need_hist = true; return true; } // (a caller without the history: it discards this decode and starts over)
    // the reference's aligned_buffer at this moment: `avail` valid bits, the refill's zeros, the history since the refill
    // bits consumed since the last real refill (the DC code included): up to 64 -- an AC refill at bits_left == 32 gives
    // rbl0 == 64, and a long last AC symbol plus a 16-bit DC code can use all of it: then the whole register
    // is history
    const int nhist = rbl0 - avail;
    uint64_t aligned = nhist >= 64 ? hist : (nhist > 0 ? hist & ((1ull << nhist) - 1) : 0);
    if (avail > 0) aligned |= (uint64_t)br.peek(avail) << (64 - avail);
    *bits = (int32_t)(aligned >> (64 - s));
    if (avail > 0) br.drop(avail);
    rbl = 0;
    br.misreads++;
    return true;
}

struct Comp {
    int id = 0, h = 1, v = 1, tq = 0, td = 0, ta = 0;
    int32_t dc_pred = 0;
    int bw = 0, bh = 0; // plane size in blocks
    int16_t* coef = nullptr; // bw * bh * 64, owned by zj_decoder::store
    size_t coef_len = 0;
    uint16_t q[64] = {0};    // the component's table as it stood when SOF was parsed (headers.rs:327 copies it there;
                             // a DQT after SOF does not reach a frame's components in the reference)
};

// backing store of one coefficient plane: heap, or pinned host memory (zj_alloc_pinned) so that
// zj_decode_planes can DMA it without a staging copy
struct PlaneStore {
    void* p = nullptr;
    size_t cap = 0;
    bool pinned = false;
    void release()
    {
        if (p) { if (pinned) zj_free_pinned(p); else free(p); }
        p = nullptr; cap = 0;
    }
    int16_t* ensure(size_t bytes, bool want_pinned)
    {
        if (p && cap >= bytes && pinned == want_pinned) return (int16_t*)p;
        release();
        size_t c = (bytes + bytes / 8 + 4096 + 63) & ~(size_t)63;
        p = want_pinned ? zj_alloc_pinned(c) : nullptr;
        if (!p) { // pageable (asked for, or no device / no pinned memory left)
            want_pinned = false;
            // whole cache lines for the walker's wide stores; large planes on 2 MB boundaries with huge pages asked for: a
            // decoder made for one file touches every page of its planes for the first time (200 MB of them for a 33 MP
            // 4:4:4 file = 48 600 page faults of 4 KB)
            const size_t align = c >= ((size_t)4 << 20) ? (size_t)2 << 20 : 64;
            c = (c + align - 1) & ~(align - 1);
            p = aligned_alloc(align, c);
#ifdef MADV_HUGEPAGE
            if (p && align > 64) (void)madvise(p, c, MADV_HUGEPAGE);
#endif
        }
        if (!p) return nullptr;
        cap = c; pinned = want_pinned;
        return (int16_t*)p;
    }
};

} // namespace

struct zj_decoder {
    // options (src/options.rs:6-40)
    int out_colorspace = ZJ_CS_RGB;
    int strict_mode = 0;
    int max_width = 16384, max_height = 16384, max_scans = 64;
    zj::Crew crew;         // helper threads, started at the first parallel region
    int threads = 4;       // options.rs:33 (default 4): here, restart segments / plane zeroing in parallel
    bool pinned = false;   // coefficient planes in pinned host memory
    int plane_store = 2;   // how the baseline walker's blocks reach their plane (STORE_*; ZJ_PLANE_STORE): in place
    int entropy = 0;       // zj_options.entropy: 0 CPU walker, 1 GPU for baseline scans worth it, 2 GPU for every eligible scan
    int sub_bytes = 128;   // sub-sequence size of the GPU entropy stage (ZJ_HUFF_SUB: 16..128, multiple of 16)
    PlaneStore blob_store; // the scan as the GPU entropy stage wants it (zj_huff.h), pinned when the planes are
    size_t blob_len = 0;
    unsigned gpu_status = 0;             // HUFF_ST_* bits of the last device scan (0: it was used as it stood)
    bool scan_ready = false;             // blob_store holds the prepared scan of the image whose headers are parsed
    const uint8_t* src = nullptr;        // the caller's file (must stay valid until the pixels are finished): the CPU
    size_t src_len = 0;                  // walker decodes it if the device hands the scan back
    uint32_t flags = 0, out_layout = 0; // extensions of the pixel path, passed through to zj_frame_desc
    PlaneStore store[3];
    ~zj_decoder() { for (auto& st : store) st.release(); blob_store.release(); }
    // state
    std::string err;
    int err_code = 0;
    int width = 0, height = 0, ncomp = 0, progressive = 0, h_max = 1, v_max = 1, mcu_x = 0, mcu_y = 0;
    int restart_interval = 0;
    int seen_sof = 0, scans = 0;
    bool coef_valid = false; // the planes hold the coefficients of a complete, successful decode_all
    long long hist_retries = 0; // images decoded a second time for it (tests)
    long long literal_retries = 0; // images whose baseline scan was decoded again by scan_baseline_literal
    bool literal = false;    // third way of decode_all: the baseline scan through scan_baseline_literal (ZJ_INT_NEED_LITERAL)
    bool track_hist = false; // second pass of decode_all: the Huffman loop keeps the reference reader's bit history (BitReader::rhist)
    uint16_t qt[4][64];
    bool qt_present[4] = {false, false, false, false};
    Huff dc[4], ac[4];
    Comp comps[3];
    // current scan
    int ns = 0, order[3] = {0, 0, 0}, ss = 0, se = 63, ah = 0, al = 0;
    uint32_t eobrun = 0;
    int dri_parallel_segments = 0; // restart segments the last baseline scan decoded concurrently (0 = serial walk)
    long long par_scan_mcus = 0;   // MCUs of the last baseline scan decoded by scan_baseline_parallel (no restart markers; 0: serial)
    // zj_decoder_decode_buffer with pinned planes: the strips of a baseline scan go to the GPU while the walker is still in
    // later rows (zj_frame_begin / _rows_ready / _end; the reference overlaps the same two things, src/mcu.rs:356-368)
    struct Stream {
        zj_ctx* ctx = nullptr;  // armed by zj_decoder_decode_buffer for the duration of its prepare step
        uint8_t* out = nullptr;
        size_t cap = 0;
        bool active = false;    // zj_frame_begin succeeded for the scan being walked
        int rows = 0;           // MCU rows reported so far
    } stream;
};

namespace {

int fail(zj_decoder* d, int code, const std::string& msg) { d->err = msg; d->err_code = code; return code; }

struct Cursor {
    const uint8_t* p;
    const uint8_t* end;
    bool u8(int& v) { if (p >= end) return false; v = *p++; return true; }
    bool u16(int& v) { if (end - p < 2) return false; v = (p[0] << 8) | p[1]; p += 2; return true; }
};

int parse_dqt(zj_decoder* d, Cursor& c)
{
    int len;
    if (!c.u16(len) || len < 2) return fail(d, ZJ_ERR_FORMAT, "Invalid DQT length");
    len -= 2;
    while (len > 0) {
        int pq_tq;
        if (!c.u8(pq_tq)) return fail(d, ZJ_ERR_FORMAT, "Could not read DQT");
        int pq = pq_tq >> 4, tq = pq_tq & 15;
        if (tq > 3) return fail(d, ZJ_ERR_DQT, "Too large table position for QT :" + std::to_string(tq) + ", expected between 0 and 3");
        if (pq == 0) {
            if (c.end - c.p < 64) return fail(d, ZJ_ERR_DQT, "Could not read DQT bytes");
            for (int i = 0; i < 64; i++) d->qt[tq][kUnZigzag[i]] = c.p[i]; // natural order, headers.rs:533-543
            c.p += 64;
            len -= 65;
        } else if (pq == 1) {
            // headers.rs:154-174: 16-bit tables are not supported by the reference
            return fail(d, ZJ_ERR_DQT, "Support for 16 bit quantization table is not complete");
        } else {
            return fail(d, ZJ_ERR_DQT, "Expected QT precision value of either 0 or 1, found " + std::to_string(pq));
        }
        d->qt_present[tq] = true;
    }
    if (len != 0) return fail(d, ZJ_ERR_DQT, "Bogus DQT length");
    return ZJ_OK;
}

int parse_dht(zj_decoder* d, Cursor& c)
{
    int len;
    if (!c.u16(len)) return fail(d, ZJ_ERR_FORMAT, "Could not read Huffman length from image");
    if (len < 2) return fail(d, ZJ_ERR_FORMAT, "Invalid Huffman length in image");
    len -= 2;
    while (len > 16) {
        int info;
        if (!c.u8(info)) return fail(d, ZJ_ERR_HUFFMAN, "Could not read bytes into the buffer");
        int cls = (info >> 4) & 15, idx = info & 15;
        if (idx >= 4) return fail(d, ZJ_ERR_HUFFMAN, "Invalid DHT index " + std::to_string(idx) + ", expected between 0 and 3");
        if (cls > 1) return fail(d, ZJ_ERR_HUFFMAN, "Invalid DHT position " + std::to_string(cls) + ", should be 0 or 1");
        if (c.end - c.p < 16) return fail(d, ZJ_ERR_HUFFMAN, "Could not read bytes into the buffer");
        uint8_t counts[17] = {0};
        int sum = 0;
        for (int i = 1; i <= 16; i++) { counts[i] = c.p[i - 1]; sum += counts[i]; }
        c.p += 16;
        len -= 17;
        if (sum > 256) return fail(d, ZJ_ERR_HUFFMAN, "Encountered Huffman table with excessive length in DHT");
        if (sum > len) return fail(d, ZJ_ERR_HUFFMAN, "Excessive Huffman table of length " + std::to_string(sum) + " found when header length is " + std::to_string(len));
        if (c.end - c.p < sum) return fail(d, ZJ_ERR_FORMAT, "Could not read symbols into the buffer");
        std::string e;
        Huff& h = cls == 0 ? d->dc[idx] : d->ac[idx];
        h.full_values = (d->flags & ZJ_FLAG_FULL_AC_VALUES) != 0;
        if (h.build(counts, c.p, sum, e)) return fail(d, ZJ_ERR_HUFFMAN, e);
        c.p += sum;
        len -= sum;
    }
    if (len > 0) return fail(d, ZJ_ERR_HUFFMAN, "Bogus Huffman table definition");
    return ZJ_OK;
}

// host coefficient planes of the frame (pinned on request): for every scan the CPU decodes
int ensure_planes(zj_decoder* d)
{
    for (int i = 0; i < d->ncomp; i++) {
        Comp& cm = d->comps[i];
        if (cm.coef) continue;
        cm.coef = d->store[i].ensure(cm.coef_len * 2, d->pinned);
        if (!cm.coef) return fail(d, ZJ_ERR_NOMEM, "out of memory for the coefficient planes");
    }
    return ZJ_OK;
}

int parse_sof(zj_decoder* d, Cursor& c, int progressive)
{
    int len, prec, h, w, nc;
    if (!c.u16(len) || !c.u8(prec) || !c.u16(h) || !c.u16(w) || !c.u8(nc)) return fail(d, ZJ_ERR_SOF, "Could not read SOF");
    if (prec != 8) return fail(d, ZJ_ERR_SOF, "The library can only parse 8-bit images, the image has " + std::to_string(prec) + " bits of precision");
    if (w > d->max_width) return fail(d, ZJ_ERR_FORMAT, "Image width " + std::to_string(w) + " greater than width limit " + std::to_string(d->max_width) + ". If use `set_limits` if you want to support huge images");
    if (h > d->max_height) return fail(d, ZJ_ERR_FORMAT, "Image height " + std::to_string(h) + " greater than height limit " + std::to_string(d->max_height) + ". If use `set_limits` if you want to support huge images");
    if (w == 0 || h == 0) return fail(d, ZJ_ERR_ZERO, "Image width or height is set to zero, cannot continue");
    if (nc == 0) return fail(d, ZJ_ERR_SOF, "Number of components cannot be zero.");
    if (len != 8 + 3 * nc) return fail(d, ZJ_ERR_SOF, "Length of start of frame differs from expected " + std::to_string(8 + 3 * nc) + ",value is " + std::to_string(len));
    if (nc != 1 && nc != 3) return fail(d, ZJ_ERR_SOF, "Invalid components. Found " + std::to_string(nc) + ", expected either 1 or 3");
    d->width = w; d->height = h; d->ncomp = nc; d->progressive = progressive;
    d->h_max = d->v_max = 1;
    for (int i = 0; i < nc; i++) {
        int id, hv, tq;
        if (!c.u8(id) || !c.u8(hv) || !c.u8(tq)) return fail(d, ZJ_ERR_FORMAT, "Could not read component data");
        if (id < 1 || id > 3) return fail(d, ZJ_ERR_FORMAT, "Unknown component id found," + std::to_string(id) + ", expected value between 1 and 3\nNote I and Q components are not supported yet");
        Comp& cm = d->comps[i];
        cm = Comp();
        cm.id = id; cm.h = hv >> 4; cm.v = hv & 15; cm.tq = tq;
        if (tq >= 4) return fail(d, ZJ_ERR_FORMAT, "Too large quantization number :" + std::to_string(tq) + ", expected value between 0 and 4");
        auto pow2 = [](int x) { return x > 0 && (x & (x - 1)) == 0; };
        if (!pow2(cm.h)) return fail(d, ZJ_ERR_FORMAT, "Horizontal sample is not a power of two(" + std::to_string(cm.h) + ") cannot decode");
        if (!pow2(cm.v)) return fail(d, ZJ_ERR_FORMAT, "Vertical sub-sample is not power of two(" + std::to_string(cm.v) + ") cannot decode");
        if (cm.h > d->h_max) d->h_max = cm.h;
        if (cm.v > d->v_max) d->v_max = cm.v;
    }
    if (nc == 1) { // grayscale with a "down-sampled" component: reset like mcu.rs:170-196
        if ((d->comps[0].h != 1 || d->comps[0].v != 1) && d->strict_mode)
            return fail(d, ZJ_ERR_FORMAT, "[strict-mode]: Grayscale image with down-sampled component.");
        d->comps[0].h = d->comps[0].v = 1;
        d->h_max = d->v_max = 1;
    }
    d->mcu_x = (w + 8 * d->h_max - 1) / (8 * d->h_max);
    d->mcu_y = (h + 8 * d->v_max - 1) / (8 * d->v_max);
    for (int i = 0; i < nc; i++) {
        Comp& cm = d->comps[i];
        if (!d->qt_present[cm.tq]) return fail(d, ZJ_ERR_DQT, "No quantization table for component " + std::to_string(cm.id));
        memcpy(cm.q, d->qt[cm.tq], sizeof cm.q);
        cm.bw = d->mcu_x * cm.h;
        cm.bh = d->mcu_y * cm.v;
        cm.coef_len = (size_t)cm.bw * cm.bh * 64;
        cm.coef = nullptr; // ensure_planes(): a scan that goes to the device entropy stage never needs host planes
    }
    if (progressive) { const int rc = ensure_planes(d); if (rc) return rc; }
    // the entropy decoder only writes non-zero coefficients.  Progressive scans accumulate into the planes, so
    // they are cleared up front (in parallel when large); a baseline scan clears each block right before it
    // decodes into it (one pass over memory instead of two) and scan_baseline clears whatever it did not reach
    for (int i = 0; i < nc && progressive; i++) {
        Comp& cm = d->comps[i];
        const size_t bytes = cm.coef_len * 2, piece = (size_t)4 << 20;
        const int n = (int)((bytes + piece - 1) / piece);
        char* base = (char*)cm.coef;
        d->crew.each(n, d->threads, [&](int k) {
            const size_t o = (size_t)k * piece;
            memset(base + o, 0, o + piece <= bytes ? piece : bytes - o);
        });
    }
    if (nc == 3) {
        // check_component_dimensions (decoder.rs:609-646): chroma must be (1,1), luma one of the four modes
        for (int i = 1; i < 3; i++)
            if (d->comps[i].h != 1 || d->comps[i].v != 1)
                return fail(d, ZJ_ERR_FORMAT, "Invalid component sample for component " + std::to_string(d->comps[i].id) + ", expected (1,1)");
        if (d->h_max > 2 || d->v_max > 2) return fail(d, ZJ_ERR_FORMAT, "Unknown down-sampling method, cannot continue");
    }
    d->seen_sof = 1;
    return ZJ_OK;
}

int parse_sos(zj_decoder* d, Cursor& c)
{
    int ls, ns;
    if (!c.u16(ls) || !c.u8(ns)) return fail(d, ZJ_ERR_SOS, "Could not read SOS");
    if (ls != 6 + 2 * ns) return fail(d, ZJ_ERR_SOS, "Bad SOS length,corrupt jpeg");
    if (ns < 1 || ns > 3) return fail(d, ZJ_ERR_SOS, "Number of components in start of scan should be less than 3 but more than 0. Found " + std::to_string(ns));
    if (!d->seen_sof || d->ncomp == 0) return fail(d, ZJ_ERR_SOF, "Number of components cannot be zero.");
    if (ns > d->ncomp) return fail(d, ZJ_ERR_FORMAT, "Number of scans " + std::to_string(ns) + " cannot be greater than number of components, " + std::to_string(d->ncomp));
    bool seen[4] = {false, false, false, false};
    for (int i = 0; i < ns; i++) {
        int id, t;
        if (!c.u8(id) || !c.u8(t)) return fail(d, ZJ_ERR_SOS, "Could not read SOS");
        int j = 0;
        while (j < d->ncomp && d->comps[j].id != id) j++;
        if (j == d->ncomp) return fail(d, ZJ_ERR_SOF, "Invalid component id " + std::to_string(id) + ", expected a value between 0 and " + std::to_string(d->ncomp));
        if (seen[j]) return fail(d, ZJ_ERR_SOF, "Duplicate ID " + std::to_string(id) + " seen twice in the same component");
        seen[j] = true;
        d->comps[j].td = (t >> 4) & 15;
        d->comps[j].ta = t & 15;
        d->order[i] = j;
    }
    int ss, se, a;
    if (!c.u8(ss) || !c.u8(se) || !c.u8(a)) return fail(d, ZJ_ERR_SOS, "Could not read SOS");
    d->ns = ns; d->ss = ss & 63; d->se = se & 63; d->ah = a >> 4; d->al = a & 15;
    if (d->ah > 13) return fail(d, ZJ_ERR_SOF, "Invalid Ah parameter " + std::to_string(d->ah) + ", range should be 0-13");
    if (d->al > 13) return fail(d, ZJ_ERR_SOF, "Invalid Al parameter " + std::to_string(d->al) + ", range should be 0-13");
    return ZJ_OK;
}

// marker loop up to and including SOS (decoder.rs:239-301); returns ZJ_OK positioned at the scan data
int parse_headers(zj_decoder* d, Cursor& c, bool first)
{
    if (first) {
        int magic;
        if (!c.u16(magic)) return fail(d, ZJ_ERR_FORMAT, "Could not read the first two magic bytes");
        if (magic != 0xffd8) return fail(d, ZJ_ERR_MAGIC, "Error parsing image. Illegal start bytes:" + std::to_string(magic));
    }
    int last = 0, extra = 0;
    for (;;) {
        int m;
        if (!c.u8(m)) return fail(d, ZJ_ERR_FORMAT, "Exhausted data while looking for a marker");
        if (last == 0xFF && m != 0xFF) { // (0xFF fill bytes before a marker are tolerated, T.81 B.1.1.2)
            extra = 0;
            int rc = ZJ_OK;
            if (m == 0xC0 || m == 0xC2) rc = parse_sof(d, c, m == 0xC2);
            else if (m == 0xC4) rc = parse_dht(d, c);
            else if (m == 0xDB) rc = parse_dqt(d, c);
            else if (m == 0xDA) return parse_sos(d, c);
            else if (m == 0xD9) return fail(d, ZJ_ERR_FORMAT, "Premature End of image");
            else if (m == 0xCC || m == 0xDC) return fail(d, ZJ_ERR_FORMAT, "Parsing of the following header `" + std::string(m == 0xCC ? "DAC" : "DNL") + "` is not supported,cannot continue");
            else if (m == 0xDD) {
                int l, ri;
                if (!c.u16(l) || l != 4 || !c.u16(ri)) return fail(d, ZJ_ERR_FORMAT, "Bad DRI length, Corrupt JPEG");
                d->restart_interval = ri;
            } else {
                // Markers the reference knows but does not handle (COM, APP0/1/14, SOI, RSTn: decoder.rs:390-409)
                // and markers it does not know at all (everything else, incl. the other SOFn: marker.rs:48-77,
                // decoder.rs:277-292) are both skipped by their length; only the message differs.
                const bool known = m == 0xFE || m == 0xE0 || m == 0xE1 || m == 0xEE || (m >= 0xD0 && m <= 0xD8);
                int l;
                if (!c.u16(l)) return fail(d, ZJ_ERR_FORMAT, "Exhausted data while reading a marker length");
                if (l < 2) return fail(d, ZJ_ERR_FORMAT, known ? "Found a marker with invalid length:" + std::to_string(l) + "\n"
                                                                 : "Found a marker with invalid length : " + std::to_string(l));
                if (c.end - c.p < l - 2) c.p = c.end; else c.p += l - 2;
            }
            if (rc) return rc;
            m = 0;
        }
        last = m;
        if (d->strict_mode && ++extra > 3) return fail(d, ZJ_ERR_FORMAT, "[strict-mode]: Extra bytes between headers");
    }
}

// ---- entropy-coded segments ------------------------------------------------------------------
inline int16_t* block_at(Comp& cm, int bx, int by) { return cm.coef + ((size_t)by * cm.bw + bx) * 64; }

// one block of a baseline scan; *err is set instead of the decoder state so that restart segments can
// run on several threads
// TRACK: also record the bit length of the block's last symbol (BitReader::last_sym) -- needed only near the end of
// the scan, where reference_saw_eoi() looks at it; the hot instantiation carries no bookkeeping
// HIST: also keep the history register of the reference's aligned_buffer (BitReader::rhist).  Off in the first decode of an
// image; a short DC read that would need it returns ZJ_INT_NEED_HIST and decode_all starts over with it on.
constexpr int ZJ_INT_NEED_HIST = 1000; // internal status, never leaves this file
// How a finished block leaves for its plane (zj_decoder::plane_store; ZJ_PLANE_STORE, profiles/r06_feeder_ab.txt):
enum { STORE_NT16 = 0,   // assembled in a cached 128-byte buffer, written with eight 16-byte non-temporal stores
       STORE_PLAIN = 1,  // the same buffer, ordinary stores (the lines are allocated in the cache, read for ownership first)
       STORE_DIRECT = 2, // cleared and filled in place
       STORE_NT32 = 3,   // four 32-byte non-temporal stores (AVX)
       STORE_NT64 = 4 }; // two whole-line non-temporal stores (AVX-512F)
__attribute__((target("avx"))) void flush_nt32(const int16_t* src, int16_t* dst)
{
    for (int i = 0; i < 4; i++) _mm256_stream_si256((__m256i*)dst + i, _mm256_load_si256((const __m256i*)src + i));
}
__attribute__((target("avx512f"))) void flush_nt64(const int16_t* src, int16_t* dst)
{
    for (int i = 0; i < 2; i++) _mm512_stream_si512((__m512i*)dst + i, _mm512_load_si512((const __m512i*)src + i));
}
template <int STORE> inline void flush_block(const int16_t* src, int16_t* dst)
{
    if (STORE == STORE_NT16) for (int i = 0; i < 8; i++) _mm_stream_si128((__m128i*)dst + i, _mm_load_si128((const __m128i*)src + i));
    if (STORE == STORE_PLAIN) for (int i = 0; i < 8; i++) _mm_store_si128((__m128i*)dst + i, _mm_load_si128((const __m128i*)src + i));
    if (STORE == STORE_NT32) flush_nt32(src, dst);
    if (STORE == STORE_NT64) flush_nt64(src, dst);
    // STORE_DIRECT: nothing to move
}
template <bool TRACK, bool HIST, int STORE>
int decode_block_baseline(const zj_decoder* d, BitReader& br, const Comp& cm, int32_t& dc_pred, int16_t* out, const char** err)
{
    const Huff& hd = d->dc[cm.td & 3];
    const Huff& ha = d->ac[cm.ta & 3];
    // Where the block is assembled (STORE): in place in its plane with ordinary stores -- the default since round 6 -- or in a
    // cached 128-byte buffer that leaves with ordinary or non-temporal stores.  Non-temporal stores were the default through
    // round 5 (the planes, 50 MB for a 4096x4096 4:2:0 frame, are written once and read by DMA; no read-for-ownership) and
    // are fine on the build container's Xeon, but on the GPU hosts' EPYC 9575F the pattern "fill a stack buffer, copy it out
    // with NT stores" runs at 1.4 GB/s (0.7 GB/s from the other socket) against 92 GB/s for a plain NT fill: 9.5 ms instead
    // of 1.3 ms for the reference's test-baseline.jpg, heap and pinned memory alike (profiles/r06_feeder_ab.txt).
    alignas(64) int16_t stack_blk[64];
    int16_t* const blk = STORE == STORE_DIRECT ? out : stack_blk;
    memset(blk, 0, 128);
    struct Flush {
        const int16_t* src; int16_t* dst;
        ~Flush() { flush_block<STORE>(src, dst); }
    } flush{blk, out};
    int rbl = br.rbl, rbl0 = HIST ? br.rbl0 : 0; // the reference's bits_left and (HIST) its history (BitReader::rbl, rbl0, rhist),
    uint64_t hist = HIST ? br.rhist : 0;         // in registers through the block
    struct KeepRbl { BitReader& b; int& r; int& r0; uint64_t& h; ~KeepRbl() { b.rbl = r < 0 ? 0 : r; if (HIST) { b.rbl0 = r0; b.rhist = h; } } } keep{br, rbl, rbl0, hist};
    if (br.nbits < 32) br.fill();
    const int dc_before = br.nbits; // >= 32: decode() does not refill, the difference is the code's length
    int s = br.decode(hd);
    if (s < 0 || s > 16) { *err = "Bad Huffman code in DC"; return ZJ_ERR_HUFFMAN; }
    const int dc_len = dc_before - br.nbits;
    if (rbl < 16) { rbl += 32; if (HIST) rbl0 = rbl; } // bitstream.rs:278
    if (HIST) hist <<= dc_len;                // drop_bits(code)
    int32_t diff = 0, short_bits = 0;
    if (s) {
        bool need_hist = false;
        if (__builtin_expect(dc_len + s > rbl, 0) && ref_dc_misread(br, rbl, rbl0, hist, dc_len, s, &short_bits, HIST ? nullptr : &need_hist)) {
            if (need_hist) { *err = "short DC read reaches the reader's history"; return ZJ_INT_NEED_HIST; }
            diff = extend(short_bits, s);
        } else { const int32_t mag = br.get(s); diff = extend(mag, s); rbl -= dc_len + s; if (HIST) hist = (hist << s) | (uint32_t)mag; } // get_bits rotates
    } else rbl -= dc_len;
    dc_pred = (int32_t)((uint32_t)dc_pred + (uint32_t)diff);
    blk[0] = (int16_t)dc_pred; // bitstream.rs:330
    for (int k = 1; k < 64;) {
        if (br.nbits < 32) br.fill(); // a code (<= 16 bits) and its magnitude bits (<= 15) without another refill
        if (rbl <= 32) { rbl += 32; if (HIST) rbl0 = rbl; } // the reference's refill before every AC symbol (bitstream.rs:334)
        // the same table as decode_mcus_v2 (Huff::ac_entry holds the reference's rules), ONE symbol at a time: an entry that
        // folds a coefficient and the EOB behind it is taken apart again
        Huff::AcEnt en = ha.actab[br.peek(AC_BITS)];
        if (!en.total) {
            en = ha.ac_escape(br.acc);
            if (!en.total) { br.drop(16); *err = "Bad Huffman code in AC"; return ZJ_ERR_HUFFMAN; }
        }
        const int len = en.len, sz = en.sz;
        br.drop(len);
        rbl -= len;
        if (HIST) hist <<= len;
        br.last_sym = len;
        if (sz) {
            const int32_t bits = (int32_t)br.peek(sz);
            br.drop(sz);
            rbl -= sz;
            // the reference's fast-AC path (src/bitstream.rs:339-347) is ONE drop_bits for code and magnitude: zeros enter its
            // aligned_buffer; the general path reads the magnitude with get_bits, which rotates it back in
            if (HIST) hist = en.idxoff >= 128 ? hist << sz : (hist << sz) | (uint32_t)bits;
            br.last_sym += sz;
            // EXTEND (T.81 F.2.2.1) without a branch: values below 2^(sz-1) are negative
            const int32_t v0 = bits + ((((bits - (1 << (sz - 1))) >> 31)) & (1 - (1 << sz)));
            const int32_t v = (int32_t)((uint32_t)v0 << en.sx) >> en.sx; // (sx: Huff::AcEnt)
            blk[kZZ.t[k + en.idxoff]] = (int16_t)v;
            k += en.k1;
        } else k += en.kadv; // ZRL, EOB, and the reference's reading of the other size-0 symbols (Huff::ac_entry)
    }
    return ZJ_OK;
}

// ---- the hot instantiation (no TRACK, no HIST) rewritten in round 6: decode_block_v2 -------------------------------------
// Same results as decode_block_baseline<false, false, STORE>, block for block and bit for bit (reader state, rbl, error
// texts); what changed is the shape of the AC loop:
//   * the reader's accumulator and bit count live in registers through the block;
//   * ONE table lookup per symbol (Huff::actab: code length, magnitude bits, where the value goes, how far k moves), the
//     value computed from the bits without a branch; ZRL and EOB are entries like any other, so the loop has one exit;
//   * the reference's bits_left (rbl) is not followed symbol by symbol: refill-before-every-AC-symbol keeps it in (32, 64]
//     and congruent to its start minus the bits consumed, so its value after the block follows from the bits the AC symbols
//     consumed and the length of the last one.
// 128 bytes of zeros at a 16-byte aligned address with vector stores (gcc turns memset(p, 0, 128) into `rep stos`)
inline __attribute__((always_inline)) void zero_block(int16_t* p)
{
#ifdef __AVX__
    const __m256i z = _mm256_setzero_si256();
    for (int i = 0; i < 4; i++) _mm256_storeu_si256((__m256i*)p + i, z);
#else
    const __m128i z = _mm_setzero_si128();
    for (int i = 0; i < 8; i++) _mm_store_si128((__m128i*)p + i, z);
#endif
}
// x << (n & 63) -- the hardware masks the count by itself; written so that the shift can start from the table entry as loaded,
// without waiting for an `and` (the shift is on the loop's critical path: accumulator -> index -> entry -> accumulator)
template <bool BMI2> inline __attribute__((always_inline)) uint64_t shl_lo6(uint64_t x, uint32_t n)
{
#if defined(__x86_64__)
    if (BMI2) __asm__("shlx %1, %0, %0" : "+r"(x) : "r"((uint64_t)n));
    else __asm__("shlq %%cl, %0" : "+r"(x) : "c"(n) : "cc");
    return x;
#else
    return x << (n & 63);
#endif
}
// A stretch of `n` consecutive MCUs starting at MCU `m0` (row-major), the reader's state in registers from the first block to
// the last.  Stops in front of the first MCU that starts at or behind `stop` (the scan's last 4 KB belong to the TRACK
// instantiation, see near_end) -- *done tells how many it decoded.  On an error *done is the index (relative to m0) of the MCU
// that failed.  pred[]: the DC predictors by component.
template <int STORE, bool BMI2>
inline __attribute__((always_inline)) int decode_mcus_v2_body(const zj_decoder* d, zj_decoder* dm, BitReader& br, int32_t* pred,
                                                              long long m0, long long n, const uint8_t* stop, long long* done,
                                                              const char** err)
{
    uint64_t acc = br.acc;
    int nbits = br.nbits;
    const uint8_t* p = br.p;
    int rbl = br.rbl;
    int rc = ZJ_OK;
    auto put = [&]() __attribute__((always_inline)) { br.acc = acc; br.nbits = nbits; br.p = p; };
    auto get = [&]() __attribute__((always_inline)) { acc = br.acc; nbits = br.nbits; p = br.p; };
    auto refill = [&]() __attribute__((always_inline)) {
        if (__builtin_expect(!br.marker && br.end - p >= 8, 1)) {
            uint64_t x;
            memcpy(&x, p, 8);
            x = __builtin_bswap64(x);
            const uint64_t y = ~x;
            if (__builtin_expect(!((y - 0x0101010101010101ull) & ~y & 0x8080808080808080ull), 1)) { // no 0xFF among them
                acc |= x >> nbits;
                p += (63 - nbits) >> 3;
                nbits |= 56;
                return;
            }
        }
        put(); br.fill(); get();
    };
    const int mcu_x = d->mcu_x;
    int my = (int)(m0 / mcu_x), mx = (int)(m0 % mcu_x);
    int last_sym = br.last_sym; // bits of the last symbol decoded (handle_restart: has the reference come across the RSTn?)
    long long i = 0;
    for (; i < n; i++) {
        if (stop && p >= stop) break;
        for (int ci = 0; ci < d->ns && !rc; ci++) {
            const int c = d->order[ci];
            Comp& cm = dm->comps[c];
            const Huff& hd = d->dc[cm.td & 3];
            const Huff::AcEnt* tab = d->ac[cm.ta & 3].actab;
            __asm__("" : "+r"(tab)); // one register for the table: the lookup's address is base + index * 4, nothing to add first
            for (int v = 0; v < cm.v && !rc; v++)
                for (int h = 0; h < cm.h; h++) {
                    int16_t* const out = cm.coef + ((size_t)(my * cm.v + v) * cm.bw + (size_t)(mx * cm.h + h)) * 64;
                    alignas(64) int16_t stack_blk[64];
                    int16_t* const blk = STORE == STORE_DIRECT ? out : stack_blk;
                    zero_block(blk);
                    if (nbits < 32) refill();
                    // DC: src/bitstream.rs:264-296
                    int s, dc_len;
                    {
                        const uint16_t e = hd.look[acc >> 55];
                        if (__builtin_expect(e != 0, 1)) { dc_len = e >> 8; s = e & 0xff; acc <<= dc_len; nbits -= dc_len; }
                        else {
                            put();
                            s = br.decode(hd); // nbits >= 32: no refill in there
                            dc_len = nbits - br.nbits;
                            get();
                        }
                    }
                    if (s < 0 || s > 16) { *err = "Bad Huffman code in DC"; rc = ZJ_ERR_HUFFMAN; }
                    else {
                        if (rbl < 16) rbl += 32; // bitstream.rs:278
                        if (__builtin_expect(dc_len + s > rbl, 0) && s) {
                            put();
                            bool need_hist = false;
                            int32_t short_bits = 0;
                            if (ref_dc_misread(br, rbl, 0, 0, dc_len, s, &short_bits, &need_hist)) {
                                *err = "short DC read reaches the reader's history";
                                rc = ZJ_INT_NEED_HIST;
                            }
                        }
                    }
                    if (!rc) {
                        // magnitude and EXTEND without a branch (s == 0 gives 0)
                        const int64_t sgn = (int64_t)acc >> 63;
                        const uint32_t raw = (uint32_t)((acc >> 1) >> (63 - s));
                        const int32_t diff = (int32_t)raw - (int32_t)(~(uint32_t)sgn & (uint32_t)((1ull << s) - 1));
                        acc <<= s; nbits -= s; // s <= 16 <= nbits
                        rbl -= dc_len + s;
                        pred[c] = (int32_t)((uint32_t)pred[c] + (uint32_t)diff);
                        blk[0] = (int16_t)pred[c]; // bitstream.rs:330
                        // AC: src/bitstream.rs:332-372
                        const int T = rbl <= 32 ? rbl + 32 : rbl; // bits_left after the refill in front of the first AC symbol
                        const int nb0 = nbits;
                        int filled = 0;
                        unsigned total = 0;
                        int k = 1;
                        uint64_t acc0;
                        const Huff::AcEnt* en;
                        Huff::AcEnt esc;
                        do {
                            if (nbits < 32) { const int b = nbits; refill(); filled += nbits - b; } // code (<= 16) + magnitude (<= 15) fit
                            acc0 = acc;
                            en = tab + (acc >> (64 - AC_BITS));
                            total = en->total;
                            if (__builtin_expect(total == 0, 0)) {
                                esc = d->ac[cm.ta & 3].ac_escape(acc);
                                en = &esc;
                                total = esc.total;
                                if (!total) { acc <<= 16; nbits -= 16; *err = "Bad Huffman code in AC"; rc = ZJ_ERR_HUFFMAN; break; }
                            }
                            acc = shl_lo6<BMI2>(acc, total);
                            nbits -= (int)total;
                            const unsigned sz = en->sz;
                            const uint64_t m = acc0 << en->len;              // the magnitude bits at the top
                            const int64_t sg = (int64_t)m >> 63;            // all ones: a positive value (T.81 F.2.2.1)
                            const uint32_t mask = kSizeMask[sz];
                            const uint32_t rw = (uint32_t)(m >> ((0u - sz) & 63)) & mask; // (size 0: the mask clears it)
                            const int32_t v0 = (int32_t)rw - (int32_t)(~(uint32_t)sg & mask);
                            const int32_t val = (int32_t)((uint32_t)v0 << en->sx) >> en->sx; // (sx: Huff::AcEnt)
                            blk[kZZ.t[k + en->idxoff]] = (int16_t)val;
                            k += en->kadv;
                        } while (k < 64);
                        if (!rc) {
                            int last = (int)total; // length of the block's last symbol
                            if (en->total != en->t1) { // a folded entry
                                const int t1 = en->len + en->sz, off = en->idxoff;
                                if (k - 64 + (off & 127) + 1 >= 64) { // the coefficient ended the block by itself
                                    acc = acc0 << t1;
                                    nbits += (int)total - t1;
                                    last = t1;
                                } else last = (int)total - t1;
                            }
                            const int c_last = nb0 + filled - nbits - last; // bits the AC symbols in front of the last one consumed
                            rbl = (c_last == 0 ? T : 33 + ((T - c_last - 33) & 31)) - last;
                            last_sym = last;
                        }
                    }
                    if (STORE != STORE_DIRECT) flush_block<STORE>(blk, out);
                    if (rc) break;
                }
        }
        if (rc) break;
        if (++mx == mcu_x) { mx = 0; my++; }
    }
    put();
    br.rbl = rbl < 0 ? 0 : rbl;
    br.last_sym = last_sym;
    *done = i;
    return rc;
}
template <int STORE>
int decode_mcus_v2(const zj_decoder* d, zj_decoder* dm, BitReader& br, int32_t* pred, long long m0, long long n,
                   const uint8_t* stop, long long* done, const char** err)
{
    return decode_mcus_v2_body<STORE, false>(d, dm, br, pred, m0, n, stop, done, err);
}
// the same body compiled for BMI2 (three-operand shifts without the flags dependency) and AVX2
template <int STORE>
__attribute__((target("bmi2,avx2"))) int decode_mcus_v2_x3(const zj_decoder* d, zj_decoder* dm, BitReader& br, int32_t* pred, long long m0,
                                                           long long n, const uint8_t* stop, long long* done, const char** err)
{
    return decode_mcus_v2_body<STORE, true>(d, dm, br, pred, m0, n, stop, done, err);
}
// the three instantiations a scan uses, for the decoder's store mode
struct BlockFns {
    using Fn = int (*)(const zj_decoder*, BitReader&, const Comp&, int32_t&, int16_t*, const char**);
    using McuFn = int (*)(const zj_decoder*, zj_decoder*, BitReader&, int32_t*, long long, long long, const uint8_t*, long long*, const char**);
    Fn hot, track, hist;
    McuFn mcus; // null: ZJ_WALKER_V1
};
template <int STORE> BlockFns block_fns_of()
{
    // (read per scan, not cached: tests switch them inside one process)
    //   ZJ_WALKER_V1       every block through decode_block_baseline, as before round 6 (the A/B and the differential test)
    //   ZJ_WALKER_GENERIC  decode_mcus_v2 without the BMI2 / AVX2 build of its body
    const bool x3 = __builtin_cpu_supports("bmi2") && __builtin_cpu_supports("avx2") && !getenv("ZJ_WALKER_GENERIC");
    const bool v1 = getenv("ZJ_WALKER_V1") != nullptr;
    return {decode_block_baseline<false, false, STORE>, decode_block_baseline<true, false, STORE>, decode_block_baseline<true, true, STORE>,
            v1 ? nullptr : x3 ? decode_mcus_v2_x3<STORE> : decode_mcus_v2<STORE>};
}
BlockFns block_fns(int store)
{
    switch (store) {
    case STORE_PLAIN: return block_fns_of<STORE_PLAIN>();
    case STORE_DIRECT: return block_fns_of<STORE_DIRECT>();
    case STORE_NT32: return block_fns_of<STORE_NT32>();
    case STORE_NT64: return block_fns_of<STORE_NT64>();
    default: return block_fns_of<STORE_NT16>();
    }
}

// restart marker handling shared by all scan kinds (mcu.rs:386-419)
long long interval_data_bytes(const BitReader& br, const uint8_t* eoi);
// handle_rst looks at stream.marker as it stands (src/mcu.rs:391): the reference's reader has come across the RSTn only if one
// of its four-byte refills reached it -- right after the refill in front of the interval's last symbol, C bits into the interval,
// it has read min(D, 4 * (C / 32 + 2)) of the D data bytes in front of the marker (the closed form of the EOI cut, below).  If
// it has not, nothing is reset: the reference decodes the next interval out of the bits it holds and then zeros (the marker
// turns up at its next refill), with the old predictors, and resets one interval late.  Rare in intact files (the last
// symbol would have to be 26 bits or longer); with bytes inserted into an interval, the usual case (tools/ref_walk_soak.py).
inline bool reference_has_seen_restart_marker(const BitReader& br)
{
    const long long c_last = br.consumed() - br.last_sym;
    return c_last >= 0 && 4 * (c_last / 32 + 2) > interval_data_bytes(br, br.mpos);
}
int handle_restart(zj_decoder* d, BitReader& br, int& todo)
{
    todo = d->restart_interval;
    if (br.nbits < 64) br.fill(); // (this reader holds more than the reference's: whether THAT has the marker is decided below)
    if (br.marker >= 0xD0 && br.marker <= 0xD7 && br.mpos && !reference_has_seen_restart_marker(br)) return ZJ_OK;
    if (br.marker >= 0xD0 && br.marker <= 0xD7) {
        br.reset();
        for (int i = 0; i < d->ncomp; i++) d->comps[i].dc_pred = 0;
        d->eobrun = 0;
    } else if (br.marker != 0 && br.marker != 0xD9) {
        return fail(d, ZJ_ERR_MCU, "Marker found in bitstream, possibly corrupt jpeg");
    }
    return ZJ_OK;
}

// ---- the reference's early exit at EOI (src/mcu.rs:337-343) -------------------------------------------------------
// After every MCU the reference tests `stream.marker`: once its bit reader has come ACROSS the EOI marker -- which
// happens while it refills, up to ~90 bits before the last coded bit is consumed -- it leaves the loop over the MCUs of
// the current row; every later row then decodes exactly one MCU (from the bits that are left, then zeros) and leaves
// again.  MCUs it never decodes keep the zeros of their fresh buffers.  On images whose last MCUs are cheap (a flat
// lower right corner costs 6-18 bits per MCU) the last few MCUs are therefore never decoded: the reference's own
// test-images/test-baseline.jpg loses 7 (tests/test_jpeg_frontend.py).  Reproducing the output needs the exact moment:
//   src/bitstream.rs:150-262  refill() appends 4 data bytes whenever it is called with bits_left <= 32 (stuffed
//   zeros are skipped, a marker ends the refill and is recorded); decode_mcu_block calls it before every AC symbol
//   (:334) and decode_dc when bits_left < 16 (:278).  Hence, right after the refill of an AC call site that follows C
//   consumed bits, the reader has read min(D, 4 * (C / 32 + 2)) of the D data bytes that precede the marker, and it
//   has seen the marker iff 4 * (C / 32 + 2) > D.  The last call site of an MCU is the one in front of its last
//   symbol.  (Checked against a state-machine model of the reader over 200 000 random symbol streams; it assumes DC
//   symbols of at most 16 bits -- with longer ones the reference consumes bits it never loaded, src/bitstream.rs:278.)
struct EoiCut {
    const uint8_t* eoi = nullptr; // first 0xFF of the EOI marker that ends the scan, or null: no cut logic
    long long rowlen = 0;         // MCUs the reference decodes per row loop (2 * mcu_x for (2,1) sampling, mcu.rs:145-152)
    bool seen = false;
    long long cut_row = -1;
    int unknown = 0;              // the marker at `eoi` is not EOI but a byte the reference has no name for (see below): its code
};
// Marker::from_u8 (src/marker.rs:48-78): every other byte behind 0xFF ends the reference's refill -- and with it the decode --
// in "Unknown marker 0xFF17" (src/bitstream.rs:199-206), at the moment its reader comes across it: the moment the EOI cut
// follows for EOI.
inline bool reference_knows_marker(int m)
{
    return m == 0xFE || m == 0xC0 || m == 0xC2 || m == 0xC4 || m == 0xCC || (m >= 0xD0 && m <= 0xDD) || m == 0xE0 || m == 0xE1 || m == 0xEE;
}
// data bytes of the current restart interval that precede the marker at `eoi`
long long interval_data_bytes(const BitReader& br, const uint8_t* eoi)
{
    const uint8_t* from = br.marker ? br.mpos : br.p; // stuffed zeros before this point are in br.stuffed already
    long long st = br.stuffed;
    for (const uint8_t* q = from; q + 1 < eoi; q++)
        if (q[0] == 0xFF) { // FF 00, or FF FF .. 00 (fill bytes in front of the zero): one data byte, the rest is not
            const uint8_t* z = q + 1;
            while (z < eoi && *z == 0xFF) z++;
            if (z < eoi && *z == 0x00) { st += (long long)(z - q); q = z; }
            else if (z < eoi) return 1ll << 40; // another marker (an RSTn) lies in front of `eoi`: that one is met first
        }
    return (long long)(eoi - br.istart) - st;
}
// call after an MCU has been decoded (and after the restart handling that may follow it)
inline void eoi_cut_after_mcu(EoiCut& cut, const BitReader& br, long long mcu_index)
{
    if (!cut.eoi) return;
    if (!cut.seen) {
        if (br.marker && br.mpos && br.mpos < cut.eoi) return; // (the reader stands at an earlier marker: an RSTn)
        const uint8_t* at = br.marker ? br.mpos : br.p;
        if (cut.eoi - at > 64) return; // the reader cannot have reached the marker's refill group yet
        const long long c_last = br.consumed() - br.last_sym;
        if (c_last < 0 || 4 * (c_last / 32 + 2) <= interval_data_bytes(br, cut.eoi)) return;
        cut.seen = true;
    }
    cut.cut_row = mcu_index / cut.rowlen;
}
// first marker in [p, end) that is neither a stuffed 0xFF00, a fill byte nor RSTn: *is_eoi tells whether it is EOI;
// returns the first 0xFF of the marker (fill bytes in front of it included), or null
const uint8_t* find_scan_end(const uint8_t* p, const uint8_t* end, bool* is_eoi)
{
    const uint8_t* const start = p;
    *is_eoi = false;
    while (p < end) {
        const uint8_t* f = (const uint8_t*)memchr(p, 0xFF, (size_t)(end - p));
        if (!f || f + 1 >= end) return nullptr;
        const uint8_t m = f[1];
        if (m == 0x00 || (m >= 0xD0 && m <= 0xD7)) { p = f + 2; continue; }
        if (m == 0xFF) { p = f + 1; continue; }
        *is_eoi = m == 0xD9;
        while (f > start && f[-1] == 0xFF) f--; // fill bytes belong to the marker
        return f;
    }
    return nullptr;
}

// Restart markers cut a baseline scan into independently decodable segments (T.81 E.1.4: predictors
// and the bit buffer restart at every RSTn).  Returns true when [p, end) holds exactly the nseg-1
// sequential RST markers a well-formed scan has before its first other marker; seg[] gets the start
// of every segment and seg[nseg] the end of the last one.
bool find_restart_segments(const uint8_t* p, const uint8_t* end, int nseg, std::vector<const uint8_t*>& seg)
{
    seg.clear();
    seg.push_back(p);
    int expect = 0;
    while (p < end) {
        const uint8_t* f = (const uint8_t*)memchr(p, 0xFF, (size_t)(end - p));
        if (!f || f + 1 >= end) { p = end; break; }
        const uint8_t m = f[1];
        if (m == 0x00) { p = f + 2; continue; }        // stuffed byte
        if (m == 0xFF) { p = f + 1; continue; }        // fill byte
        if (m >= 0xD0 && m <= 0xD7) {
            if (m != 0xD0 + (expect & 7)) return false; // out of sequence: let the serial path decide
            expect++;
            seg.push_back(f + 2);
            p = f + 2;
            continue;
        }
        p = f;                                          // EOI or any other marker ends the scan
        break;
    }
    seg.push_back(p);
    return (int)seg.size() == nseg + 1;
}

inline void clear_mcu(zj_decoder* dm, int mx, int my)
{
    for (int ci = 0; ci < dm->ncomp; ci++) {
        Comp& cm = dm->comps[ci];
        for (int v = 0; v < cm.v; v++)
            for (int h = 0; h < cm.h; h++) memset(block_at(cm, mx * cm.h + h, my * cm.v + v), 0, 128);
    }
}

long long eoi_rowlen(const zj_decoder* d) { return (d->ncomp == 3 && d->h_max == 2 && d->v_max == 1) ? 2ll * d->mcu_x : d->mcu_x; }

// eoi: first 0xFF of the EOI marker when this segment is the one that ends with it, else null.
// must_end (every segment but the last): the reader must have come to the segment's end with its last MCU, as the serial walk
// must see the RSTn marker in handle_restart() -- with bytes left over (a damaged file) the reference goes on decoding the NEXT
// interval out of them, predictors and all, which only the serial walk reproduces (found by tools/stream_soak.py, round 6:
// rounds 4-5 decoded such segments independently and returned other garbage than the reference's).
int scan_baseline_segment(const zj_decoder* d, zj_decoder* dm, const uint8_t* p, const uint8_t* end, long long mcu0,
                          long long nmcu, const char** err, const uint8_t* eoi = nullptr, bool must_end = false)
{
    BitReader br;
    br.p = p; br.end = end; br.istart = p;
    const BlockFns fn = block_fns(d->plane_store);
    EoiCut cut;
    cut.eoi = eoi; cut.rowlen = eoi_rowlen(d);
    int32_t pred[3] = {0, 0, 0};
    const uint8_t* const stop = cut.eoi ? cut.eoi - (cut.eoi - p > 4096 ? 4096 : cut.eoi - p) : nullptr;
    for (long long i = 0; i < nmcu; i++) {
        if (fn.mcus && !d->track_hist && !cut.seen) { // as many MCUs as possible in one go (decode_mcus_v2)
            long long done = 0;
            const int rc = fn.mcus(d, dm, br, pred, mcu0 + i, nmcu - i, stop, &done, err);
            if (rc) return rc;
            i += done;
            if (i == nmcu) break;
        }
        const int my = (int)((mcu0 + i) / d->mcu_x), mx = (int)((mcu0 + i) % d->mcu_x);
        if (cut.seen && (mcu0 + i) / cut.rowlen == cut.cut_row) { clear_mcu(dm, mx, my); continue; }
        const bool near_end = cut.eoi && cut.eoi - br.p <= 4096; // an MCU is at most 6 blocks x 64 x 27 bits = 1.3 KB
        for (int ci = 0; ci < d->ns; ci++) {
            Comp& cm = dm->comps[d->order[ci]];
            for (int v = 0; v < cm.v; v++)
                for (int h = 0; h < cm.h; h++) {
                    int16_t* blk = block_at(cm, mx * cm.h + h, my * cm.v + v);
                    int rc = (d->track_hist ? fn.hist : near_end ? fn.track : fn.hot)(d, br, cm, pred[d->order[ci]], blk, err);
                    if (rc) return rc;
                }
        }
        eoi_cut_after_mcu(cut, br, mcu0 + i);
    }
    _mm_sfence(); // the blocks left with streaming stores (decode_block_baseline)
    // What the reader has met by now must be what the serial walk would have met (handle_restart: fill, then look at the
    // marker).  A segment but the last ends right behind its RSTn: that marker, and nothing in front of it -- with bytes left
    // over the reference goes on decoding the NEXT interval out of them, a marker in mid-segment resets it early or is "Marker
    // found in bitstream".  The last segment ends in front of the scan's closing marker: no marker at all.
    if (br.nbits < 64) br.fill();
    if (must_end ? !(br.marker >= 0xD0 && br.marker <= 0xD7 && br.p == br.end && reference_has_seen_restart_marker(br))
                 : (br.marker != 0 || br.mpos != nullptr)) {
        *err = "a restart interval that does not end at its marker";
        return ZJ_ERR_MCU;
    }
    if (br.nbits < br.pad) { // (made-up bits consumed: the interval ran out of data -- the literal decode's case; the last segment too,
                             // whose reader stops in front of the scan's closing marker and pads from there)
        *err = "a restart interval that does not end at its marker";
        return ZJ_ERR_MCU;
    }
    return ZJ_OK;
}


// ---- a baseline scan WITHOUT restart markers on several threads (round 6) ------------------------------------------------
// Restart markers cut a scan into independently decodable pieces (above); most files have none, and then `num_threads`
// (reference default 4, src/options.rs:33) bought nothing.  A Huffman stream can still be entered in the middle: a decoder
// that starts at an arbitrary byte with the wrong idea of where it is reads garbage for a while and then, with high
// probability, falls into step with the true sequence of symbols -- from then on it is at MCU starts exactly where the true
// decoder is.  So:
//   look (parallel) the scan is cut into one chunk per thread, and every thread looks through one chunk: nothing but data and
//                   stuffed zeros may be there, and the stuffed zeros in front of a chunk give its first bit's number;
//   A  (parallel)   chunk 0 begins where the scan begins: it is decoded for real straight away.  Every other thread decodes the
//                   STRUCTURE of its chunk only -- code lengths and zig-zag advances (in groups of symbols, Huff::groups), no
//                   values, no stores -- from its chunk's first byte, assuming an MCU starts there, and notes the reader's
//                   state at every 16th MCU start it believes in (ParSnap);
//   stitch (serial) the true reader's state at the end of chunk t-1 is looked up among chunk t's notes; if it is not one of
//                   them the true structure decode goes on into chunk t, a few MCUs at a time, until it is (typically within
//                   twenty MCUs).  From that note on chunk t's notes are true, and how many MCUs precede each is known: they
//                   join one list of ANCHORS over the whole scan (MCU index, reader, DC predictors);
//   B  (parallel)   the MCUs behind chunk 0 are cut at anchors into four parts per thread, equal in MCUs, which the threads
//                   take in order and decode for real (decode_mcus_v2).  The DC predictors a part starts with come from pass A as
//                   well: the structure decode forms the DC differences (one per block) and keeps their running sums, and
//                   differences of sums inside a chunk are true once the chunk is in step.
//   check           every part must end at the bit the next one began at, with the predictors the next one was given; what
//                   lies in front of the first link that does not hold is kept (chunk 0 always is: its decode is the serial
//                   walk's own), the serial walk decodes the rest.
// Flat areas (two symbols per block, identical MCUs) are where it does not work: a reader that enters such a run out of step
// stays out of step until the picture changes.  The stitching has `patience` for max(512, 1/64 of the picture) MCUs per
// chunk; then the calling thread decodes the rest of that chunk for real (a BRIDGE: a part like any other when the links
// are checked, only decoded early and alone) and the stitching goes on with the next chunk from where the bridge ends.
// The scan's last 8 KB -- where the reference's early exit at EOI lives -- and everything the walker treats specially stay
// with the serial walk: any DC symbol the reference might read short (ref_dc_misread: the structure decode follows bits_left
// by the same rules), any code that does not exist, any marker or 0xFF fill byte, a chunk that never falls into step -> the
// attempt is dropped (or ends there) and the serial walk decodes the rest as if nothing had happened (it clears every block
// before it fills it).
// 4096 x 4096 4:2:0 q = 90, 3.5 MB, on the GPU host: 17.7 ms on one thread, 13.1 on 2, 7.3 on 4, 4.2 on 8, 2.4 on 16; the
// reference's own test-baseline.jpg (73 KB, 6 bits per block, mostly sky; tried before the 96 KB threshold existed) 0.90 ->
// 0.97 ms: its second chunk begins in the flat part and the attempt ends there (profiles/r06_walker.txt).
// ZJ_PAR_SCAN=off; tools/par_scan_soak.py.
constexpr size_t kParEvery = 16;
struct ParSnap {
    const uint8_t* p;   // the reader at an MCU start: next byte to load, accumulator, bits in it
    uint64_t acc;
    long long dbits;    // data bits (stuffing removed) consumed since the scan began: the same number for every reader that is
                        // at this point of the stream, whatever its refill history
    int nbits;
    int rbl;            // the reference's bits_left here, if `exact`
    bool exact;
    bool hazard;        // the MCU that starts here holds a DC symbol the reference may read short
    int32_t dc[3];      // the components' DC predictors here, counted from whatever the run was started with (wrapping)
    long long mcu;      // MCUs decoded by the run before this one
};

// Structure-only decode from the state of `br` (br.istart marks the byte whose first bit is data bit `base_bits`): a ParSnap
// at every `every`-th MCU start, the first included, until an MCU starts at or behind `until` (recorded, not decoded) or
// `max_new` MCUs have been decoded (recorded as well).  A snapshot's hazard flag covers the MCUs up to the next snapshot.
// Every 16th is what the chunks keep: a chunk's notes then stay in the core's L2 (48 bytes x 16 K MCUs were 0.8 MB per
// thread, fresh pages each time), and the stitching walks at most 15 MCUs further to meet one.
// false: something undecodable (no such code, a marker or the end of the data in view).
bool par_structure_run(const zj_decoder* d, BitReader& br, long long base_bits, int rbl, bool exact, const int32_t pred0[3], const uint8_t* until,
                       size_t max_new, size_t every, std::vector<ParSnap>& out)
{
    struct Blk { const Huff* hd; const Huff* ha; int comp; };
    Blk pat[8];
    int bpm = 0;
    for (int ci = 0; ci < d->ns; ci++) {
        const Comp& cm = d->comps[d->order[ci]];
        for (int q = 0; q < cm.h * cm.v; q++) {
            if (bpm == 8) return false;
            pat[bpm++] = Blk{&d->dc[cm.td & 3], &d->ac[cm.ta & 3], d->order[ci]};
        }
    }
    uint32_t pred[3] = {(uint32_t)pred0[0], (uint32_t)pred0[1], (uint32_t)pred0[2]};
    uint64_t acc = br.acc;
    int nbits = br.nbits;
    const uint8_t* p = br.p;
    auto put = [&]() __attribute__((always_inline)) { br.acc = acc; br.nbits = nbits; br.p = p; };
    auto get = [&]() __attribute__((always_inline)) { acc = br.acc; nbits = br.nbits; p = br.p; };
    auto refill = [&]() __attribute__((always_inline)) {
        if (__builtin_expect(!br.marker && br.end - p >= 8, 1)) {
            uint64_t x;
            memcpy(&x, p, 8);
            x = __builtin_bswap64(x);
            const uint64_t y = ~x;
            if (__builtin_expect(!((y - 0x0101010101010101ull) & ~y & 0x8080808080808080ull), 1)) {
                acc |= x >> nbits;
                p += (63 - nbits) >> 3;
                nbits |= 56;
                return;
            }
        }
        put(); br.fill(); get();
    };
    put();
    long long bits = base_bits + br.consumed(); // data bits consumed, followed symbol by symbol (checked against the reader at the end)
    for (size_t done = 0;; done++) {
        if (br.marker) return false;
        const bool last_one = p >= until || done == max_new;
        if (last_one || done % every == 0)
            out.push_back(ParSnap{p, acc, bits, nbits, rbl, exact, false, {(int32_t)pred[0], (int32_t)pred[1], (int32_t)pred[2]}, (long long)done});
        if (last_one) { put(); return !br.marker && bits == base_bits + br.consumed(); }
        bool hazard = false;
        for (int j = 0; j < bpm; j++) {
            if (nbits < 32) refill();
            // DC (src/bitstream.rs:264-296): the reference refills below 16 bits only, the symbol may be longer
            int s, len;
            {
                const uint16_t e = pat[j].hd->look[acc >> 55];
                if (__builtin_expect(e != 0, 1)) { len = e >> 8; s = e & 0xff; }
                else {
                    put();
                    const int before = br.nbits;
                    s = br.decode(*pat[j].hd);
                    len = before - br.nbits;
                    br.acc = acc; br.nbits = nbits; // (only the length was wanted)
                }
            }
            if (s < 0 || s > 16) return false;
            if (rbl < 16) rbl += 32;
            if (s && len + s > (exact ? rbl : 16)) hazard = true;
            rbl -= len + s;
            if (rbl < 0) { rbl = 0; hazard = true; }
            // the DC difference itself: the predictors are part of the state the ranges start from (values of the AC
            // coefficients are not needed for that, and not formed)
            {   // magnitude and EXTEND without a branch (s == 0 gives 0), as in decode_mcus_v2
                const uint64_t m = acc << len;
                const int64_t sgn = (int64_t)m >> 63;
                const uint32_t raw = (uint32_t)((m >> 1) >> (63 - s));
                pred[pat[j].comp] += raw - (~(uint32_t)sgn & (uint32_t)((1ull << s) - 1));
            }
            acc <<= len + s; nbits -= len + s; bits += len + s;
            const int T = rbl <= 32 ? rbl + 32 : rbl;
            // AC (src/bitstream.rs:332-372), one symbol per table entry (an entry that carries the EOB as well is read for its
            // first symbol only)
            const Huff::AcEnt* const tab = pat[j].ha->actab;
            const Huff::Group* const grp = pat[j].ha->groups;
            int k = 1, nac = 0, last = 0;
            long long last_at = bits;
            do {
                if (nbits < 32) refill(); // (a group is at most AC_BITS bits of codes and magnitudes + 15 of the last magnitude)
                const unsigned w = (unsigned)(acc >> (64 - AC_BITS));
                const Huff::Group g = grp[w];
                int take = g.total, adv = g.kadv;
                last = g.lastlen;
                nac += g.nsym;
                // rare: a code longer than the window; a block that ends inside the group (coefficient 63 is coded and another
                // symbol's code follows in the window: that one belongs to the next block) -> the first symbol alone
                if (__builtin_expect(g.nsym == 0 || k + g.kpre >= 64, 0)) {
                    Huff::AcEnt en = tab[w];
                    if (!en.total) {
                        en = pat[j].ha->ac_escape(acc);
                        if (!en.total) return false;
                    }
                    take = en.t1; adv = en.k1; last = en.t1;
                    nac += 1 - g.nsym;
                }
                last_at = bits + take - last;
                k += adv;
                acc <<= take; nbits -= take; bits += take;
            } while (k < 64);
            // bits_left behind the block: behind the refill in front of an AC symbol at data bit C it is 64 - (C mod 32)
            // whatever came before -- except in front of a block's FIRST AC symbol, where it follows from the DC symbol
            if (nac >= 2) { rbl = 64 - (int)(last_at & 31) - last; exact = true; }
            else rbl = T - last;
            if (nbits < 0) return false; // (padding ran out: the end of the data)
        }
        if (hazard) out.back().hazard = true;
    }
}

// Returns the number of MCUs decoded from the start of the scan, with `br` at the start of the next MCU and the components'
// predictors set: all of the region when every check held, chunk 0 plus the parts whose links held otherwise (0: not
// attempted, or chunk 0 itself met something unusual; nothing but plane contents has changed then).
inline void stream_rows(zj_decoder* d, long long mcus_done);
long long scan_baseline_parallel(zj_decoder* d, BitReader& br, const BlockFns& fn, const uint8_t* scan_end)
{
    const uint8_t* const p0 = br.p;
    const auto t_in = std::chrono::steady_clock::now();
    int T = d->threads < 16 ? d->threads : 16;
    const long long usable = (long long)(scan_end - p0) - 8192; // the tail stays with the serial walk (EoiCut, near_end)
    // Below 16 KB per thread, or 96 KB in all, the attempt costs more than it can bring: two wake-ups of the crew (20-40 us
    // each; for a decoder's first file, starting its threads: ~50 us apiece) and, where the picture is flat, the stitching's
    // patience -- 0.08 ms on the reference's 73 KB test-baseline.jpg, which one thread decodes in 0.9 ms.
    long long min_chunk = 16384, min_scan = 98304;
    if (const char* e = getenv("ZJ_PAR_MIN_CHUNK")) { const long v = atol(e); if (v >= 256) { min_chunk = v; min_scan = 2 * v; } } // (tests: small files)
    if (usable < min_scan) return 0;
    if ((long long)T * min_chunk > usable) T = (int)(usable / min_chunk);
    if (T < 2) return 0;
    // chunk starts: never on the zero that follows a 0xFF
    std::vector<const uint8_t*> start((size_t)T + 1);
    std::vector<long long> base((size_t)T);
    // Chunk 0 is decoded for real straight away (its reader's state is known), the others structure first and coefficients
    // later, with chunk 0's thread helping: chunk 0 is sized so that its decode takes what a structure pass over one of the
    // others takes (0.65 of a decode per byte on the GPU hosts since the structure decode takes its symbols in groups)
    const double c0 = 0.65 / ((double)(T - 1) + 0.65);
    for (int t = 0; t <= T; t++) {
        const uint8_t* q = p0 + (t == 0 ? 0 : (long long)((double)usable * (c0 + (1.0 - c0) * (double)(t - 1) / (double)(T - 1))));
        if (t == T) q = p0 + usable;
        if (t && q[-1] == 0xFF && q[0] == 0x00) q++;
        start[(size_t)t] = q;
    }
    // Nothing but data and stuffed zeros may lie in the region (a marker or 0xFF fill bytes: not for this path), and the
    // stuffed zeros in front of a chunk are needed for its data-bit numbers: a region of its own ahead of pass A, every
    // thread looking through one chunk (13 600 memchr calls over the 3.5 MB file: 0.4 ms when one thread did it)
    std::vector<long long> stuffed_in((size_t)T, 0);
    std::vector<char> clean((size_t)T, 0);
    const auto look_through = [&](int t) {
        const uint8_t* q = start[(size_t)t];
        const uint8_t* const e = start[(size_t)t + 1];
        long long n = 0;
        while (q < e) {
            const uint8_t* f = (const uint8_t*)memchr(q, 0xFF, (size_t)(e - q));
            if (!f) break;
            if (f + 1 < scan_end && f[1] == 0x00) { n++; q = f + 2; } // (a pair across the cut: the next chunk starts behind its zero)
            else return;
        }
        stuffed_in[(size_t)t] = n;
        clean[(size_t)t] = 1;
    };
    for (int t = 1; t <= T; t++) if (start[(size_t)t] <= start[(size_t)t - 1]) return 0;
    d->crew.each(T, T, look_through);
    {   // data bits in front of every chunk (the structure decode needs them from the scan's first bit: the reference's
        // bits_left behind an AC symbol depends on the symbol's bit number mod 32)
        long long stuffed = 0;
        for (int t = 0; t < T; t++) {
            if (!clean[(size_t)t]) return 0;
            base[(size_t)t] = 8 * ((long long)(start[(size_t)t] - p0) - stuffed);
            stuffed += stuffed_in[(size_t)t];
        }
    }
    const bool dbg = getenv("ZJ_PAR_DEBUG") != nullptr;
    const auto clk = [] { return std::chrono::steady_clock::now(); };
    const auto t_a = clk();
    for (int ci = 0; ci < d->ns; ci++) d->ac[d->comps[d->order[ci]].ta & 3].build_groups();
    // A: chunk 0 for real; the structure of every other chunk, speculatively
    if (br.nbits != 0 || br.marker) return 0; // (the scan's reader has not been used yet: chunk 0 starts a fresh one at p0)
    std::vector<std::vector<ParSnap>> seen((size_t)T);
    std::vector<char> ok((size_t)T, 0);
    struct Out { int32_t pred[3], pred0[3]; long long begin_bits, end_bits; int rc; BitReader br; double t0 = 0, t1 = 0; }; // pred0: at the start
    Out head{};          // chunk 0
    std::vector<double> a_ms((size_t)T, 0.0); // (ZJ_PAR_DEBUG) when each chunk's pass A ended
    long long head_mcus = 0;
    d->crew.each(T, T, [&](int t) {
        BitReader r;
        r.p = start[(size_t)t]; r.end = br.end; r.istart = r.p;
        if (t == 0) {
            r.rbl = br.rbl;
            for (int c = 0; c < 3; c++) head.pred[c] = 0; // (scan_baseline has just set the predictors to 0)
            const char* err = nullptr;
            head.begin_bits = 0;
            head.rc = fn.mcus(d, d, r, head.pred, 0, (long long)d->mcu_x * d->mcu_y, start[1], &head_mcus, &err);
            if (!head.rc && (r.marker || r.pad || r.mpos)) head.rc = ZJ_ERR_HUFFMAN; // (belt and braces: look_through lets nothing but data and stuffed zeros in)
            head.end_bits = r.consumed();
            head.br = r;
            head.t1 = std::chrono::duration<double, std::milli>(clk() - t_a).count();
            return;
        }
        seen[(size_t)t].reserve((size_t)((start[(size_t)t + 1] - start[(size_t)t]) / (24 * kParEvery) + 64));
        const int32_t zero[3] = {0, 0, 0}; // (the scan begins with predictors 0; a chunk's own count starts anywhere)
        ok[(size_t)t] = par_structure_run(d, r, base[(size_t)t], 0, false, zero, start[(size_t)t + 1], (size_t)-1, kParEvery, seen[(size_t)t]);
        a_ms[(size_t)t] = std::chrono::duration<double, std::milli>(clk() - t_a).count();
    });
    const auto t_s = clk();
    // anything unusual in chunk 0 (a DC symbol the reference may read short, a bad code): the serial walk owns it
    if (head.rc || head_mcus < 1) return 0;
    // From here on chunk 0 is decoded, and decoded truly -- it is the serial walk's own decode of those MCUs -- whatever
    // becomes of the rest: the scan goes on behind `mcus` MCUs from the reader `o` ended with.
    const auto keep = [&](const Out& o, long long mcus) {
        for (int c = 0; c < d->ncomp; c++) d->comps[c].dc_pred = o.pred[c];
        const BitReader& r = o.br;
        const long long stuffed_total = ((long long)(r.p - p0) * 8 - (o.end_bits + r.nbits)) / 8; // 0xFF00 pairs in front of r.p
        br.p = r.p; br.acc = r.acc; br.nbits = r.nbits; br.rbl = r.rbl; br.marker = 0; br.mpos = nullptr; br.pad = 0;
        br.istart = p0; br.stuffed = (uint32_t)stuffed_total;
        br.rbl0 = 0; br.rhist = 0;
        return mcus;
    };
    stream_rows(d, head_mcus); // (zj_decoder_decode_buffer: those rows can go to the GPU while the rest is decoded)
    // stitch: one list of TRUE MCU starts (anchors) over the whole scan, each with the MCU's index and the predictors that hold
    // there; `cur` = the true reader behind the last anchored MCU.  A chunk's notes join the list from the note the true
    // reader lands on; the MCUs the true structure decode has to walk through to get there are anchors too (a chunk that never
    // falls into step -- long runs of identical short MCUs keep a shifted reader shifted -- is bridged, see below).
    struct Anchor { ParSnap s; long long mcu; };
    std::vector<Anchor> anchors;
    {
        size_t n = 1024;
        for (const auto& v : seen) n += v.size();
        anchors.reserve(n);
    }
    // v[from] is where the true reader `at` stands (MCU at_mcu): v[from, size - 1) become anchors, v.back() the new `cur`
    ParSnap cur{};
    long long cur_mcu = 0;
    const auto join = [&](const std::vector<ParSnap>& v, size_t from, const ParSnap& at, long long at_mcu) {
        const auto true_dc = [&](const ParSnap& x, int32_t* out) {
            for (int c = 0; c < 3; c++) out[c] = (int32_t)((uint32_t)at.dc[c] + ((uint32_t)x.dc[c] - (uint32_t)v[from].dc[c]));
        };
        for (size_t i = from; i + 1 < v.size(); i++) {
            Anchor a{v[i], at_mcu + (v[i].mcu - v[from].mcu)};
            true_dc(v[i], a.s.dc);
            if (i == from) { a.s.rbl = at.rbl; a.s.exact = at.exact; } // (a speculative run may not know bits_left yet; the true reader does)
            anchors.push_back(a);
        }
        ParSnap e = v.back();
        true_dc(v.back(), e.dc);
        if (v.size() == from + 1) { e.rbl = at.rbl; e.exact = at.exact; }
        cur_mcu = at_mcu + (v.back().mcu - v[from].mcu);
        cur = e;
    };
    cur = ParSnap{head.br.p, head.br.acc, head.end_bits, head.br.nbits, head.br.rbl, true, false, {head.pred[0], head.pred[1], head.pred[2]}, head_mcus};
    cur_mcu = head_mcus;
    std::vector<ParSnap> more;
    size_t walked = 0;
    // A reader that enters a run of identical short MCUs out of step (flat areas: two symbols per block) stays out of step
    // until the picture changes, and the true structure decode would have to walk through all of them one after the other
    // at more than the real decode costs for such blocks (a note per MCU).  Past `patience` MCUs the stitching BRIDGES the
    // rest of the chunk instead: the calling thread decodes it for real, here and now, up to the next chunk's first byte, and
    // the stitching goes on from where that ends -- the next chunk may well be in step (the reference's speed_bench_hv: sky
    // over a third of the picture, detail below).  A bridge is a part like any other when the links are checked.
    size_t patience = std::max<size_t>(512, (size_t)((long long)d->mcu_x * d->mcu_y / 64));
    if (const char* e = getenv("ZJ_PAR_PATIENCE")) { const long v = atol(e); if (v >= 1) patience = (size_t)v; } // (tests)
    struct Bridge { size_t at; long long m0, m1; Out out; }; // in front of anchors[at]: MCUs [m0, m1) are decoded
    std::vector<Bridge> bridges;
    long long bridged = 0;
    bool gave_up = false;
    for (int t = 1; t < T && !gave_up; t++) {
        const size_t walked0 = walked;
        size_t stride = 8;
        const std::vector<ParSnap>& cand = seen[(size_t)t];
        size_t j = 0;
        const auto meets = [&](const ParSnap& x) {
            while (j < cand.size() && cand[j].dbits < x.dbits) j++;
            return ok[(size_t)t] && j + 1 < cand.size() && cand[j].dbits == x.dbits;
        };
        for (;;) {
            if (meets(cur)) { join(cand, j, ParSnap(cur), cur_mcu); break; }
            if (cur.p >= start[(size_t)t + 1]) break; // past this chunk without ever meeting it: the chunk is dropped
            if (walked - walked0 > patience) {
                if (!cur.exact) { gave_up = true; break; }
                Bridge g{anchors.size(), cur_mcu, cur_mcu, Out{}};
                Out& o = g.out;
                for (int c = 0; c < 3; c++) o.pred[c] = o.pred0[c] = cur.dc[c];
                BitReader& r = o.br;
                r.p = cur.p; r.acc = cur.acc; r.nbits = cur.nbits; r.end = br.end; r.istart = cur.p;
                r.rbl = cur.rbl;
                o.begin_bits = cur.dbits;
                long long done = 0;
                const char* err = nullptr;
                o.rc = fn.mcus(d, d, r, o.pred, cur_mcu, (long long)d->mcu_x * d->mcu_y - cur_mcu, start[(size_t)t + 1], &done, &err);
                if (o.rc || done < 1 || r.marker || r.pad || r.mpos) { gave_up = true; break; } // (the serial walk owns whatever that was)
                o.end_bits = cur.dbits + cur.nbits + r.consumed();
                g.m1 = cur_mcu + done;
                cur = ParSnap{r.p, r.acc, o.end_bits, r.nbits, r.rbl, true, false, {o.pred[0], o.pred[1], o.pred[2]}, 0};
                cur_mcu = g.m1;
                bridged += done;
                bridges.push_back(g);
                continue; // (cur.p >= the next chunk's start now: on to the next junction)
            }
            // a few more true MCUs (8, then 16 ... 64 at a time: most chunks are met within twenty), then look again
            BitReader r;
            r.p = cur.p; r.acc = cur.acc; r.nbits = cur.nbits; r.end = br.end; r.istart = cur.p;
            more.clear();
            if (!par_structure_run(d, r, cur.dbits + cur.nbits, cur.rbl, cur.exact, cur.dc, start[(size_t)T], stride, 1, more) || more.size() < 2) { gave_up = true; break; }
            if (stride < 64) stride *= 2;
            size_t i = 1;
            for (;; i++) {
                anchors.push_back(Anchor{more[i - 1], cur_mcu++});
                walked++;
                if (i + 1 == more.size() || meets(more[i]) || more[i].p >= start[(size_t)t + 1]) break;
            }
            cur = more[i]; // (what the run decoded behind it is decoded again: by the chunk's own notes or the next run)
        }
    }
    const long long total_mcus = cur_mcu;
    if (total_mcus <= 0 || total_mcus > (long long)d->mcu_x * d->mcu_y || !cur.exact) return keep(head, head_mcus);
    // The MCUs that are left to decode, in about P = 4 T pieces of equal size, each beginning at an anchor whose bits_left is
    // known.  The threads take them in order, one at a time: a helper that wakes up late costs a quarter of what it would with
    // one piece per thread, and the picture becomes final from the top -- the calling thread, between its own pieces, checks
    // the links of what is finished and hands those rows to the GPU (zj_decoder_decode_buffer's streamed frame).
    struct Piece { size_t a0, a1; long long m0, m1; Out* out; bool bridge; };
    std::vector<Piece> pieces;
    {
        const int P = 4 * T;
        const long long to_do = total_mcus - head_mcus - bridged;
        size_t lo = 0;
        bool usable = true;
        for (size_t k = 0; k <= bridges.size() && usable; k++) {
            const size_t hi = k < bridges.size() ? bridges[k].at : anchors.size();
            const long long run_end = k < bridges.size() ? bridges[k].m0 : total_mcus;
            if (lo < hi) {
                if (!anchors[lo].s.exact) { usable = false; break; } // (cannot happen: a run begins where the true reader stood)
                const long long run_begin = anchors[lo].mcu, run = run_end - run_begin;
                int pk = to_do > 0 ? (int)((run * P + to_do / 2) / to_do) : 1;
                if (pk < 1) pk = 1;
                size_t a = lo;
                for (int q = 0; q < pk && a < hi; q++) {
                    size_t b = hi;
                    if (q + 1 < pk) {
                        const long long want = run_begin + run * (q + 1) / pk;
                        b = a + 1;
                        while (b < hi && (anchors[b].mcu < want || !anchors[b].s.exact)) b++;
                    }
                    pieces.push_back(Piece{a, b, anchors[a].mcu, b < hi ? anchors[b].mcu : run_end, nullptr, false});
                    a = b;
                }
            }
            if (k < bridges.size()) pieces.push_back(Piece{hi, hi, bridges[k].m0, bridges[k].m1, &bridges[k].out, true});
            lo = hi;
        }
        if (!usable) return keep(head, head_mcus);
    }
    const size_t NP = pieces.size();
    const auto t_b = clk();
    // B: the coefficients, every piece from its own first MCU start and the predictors that hold there
    std::vector<Out> res(NP);
    std::vector<std::atomic<int>> finished(NP);
    std::vector<int> todo;
    for (size_t q = 0; q < NP; q++) {
        finished[q].store(pieces[q].bridge ? 1 : 0, std::memory_order_relaxed);
        if (!pieces[q].bridge) { pieces[q].out = &res[q]; todo.push_back((int)q); }
    }
    // A piece that begins on the bit the one in front ended on, with the predictors it ended with, begins at a true MCU start
    // (by induction from chunk 0, whose decode IS the serial decode) and at the MCU index it was given -- so its own end is
    // true as well.  The first link that does not hold ends what is kept; the serial walk goes on from there.
    const Out* last = &head;
    long long kept = head_mcus;
    size_t checked = 0; // pieces [0, checked) are behind `last`
    const auto follow_links = [&](size_t upto) { // (one thread at a time: the caller)
        for (; checked < upto; checked++) {
            const Piece& pc = pieces[checked];
            if (!finished[checked].load(std::memory_order_acquire)) return false;
            const Out& o = *pc.out;
            if (o.rc || last->end_bits != o.begin_bits || memcmp(last->pred, o.pred0, sizeof last->pred) != 0) return false;
            last = &o;
            kept = pc.m1;
        }
        return true;
    };
    const std::thread::id caller = std::this_thread::get_id();
    d->crew.each((int)todo.size(), T, [&](int ti) {
        const size_t q = (size_t)todo[(size_t)ti];
        const Piece& pc = pieces[q];
        Out& o = res[q];
        o.t0 = std::chrono::duration<double, std::milli>(clk() - t_b).count();
        o.rc = 0; o.begin_bits = o.end_bits = -1;
        bool hazard = false; // a DC symbol the reference may read short anywhere in the piece: the serial walk decides what it reads there
        for (size_t i = pc.a0; i < pc.a1; i++) hazard |= anchors[i].s.hazard;
        const ParSnap& s0 = anchors[pc.a0].s;
        for (int c = 0; c < 3; c++) o.pred[c] = o.pred0[c] = s0.dc[c];
        o.begin_bits = s0.dbits;
        if (hazard) o.rc = ZJ_INT_NEED_HIST;
        else {
            const long long count = pc.m1 - pc.m0;
            BitReader& r = o.br;
            r.p = s0.p; r.acc = s0.acc; r.nbits = s0.nbits; r.end = br.end; r.istart = s0.p;
            r.rbl = s0.rbl;
            long long done = 0;
            const char* err = nullptr;
            o.rc = fn.mcus(d, d, r, o.pred, pc.m0, count, nullptr, &done, &err);
            if (!o.rc && done != count) o.rc = ZJ_ERR_HUFFMAN;
            if (!o.rc && (r.marker || r.pad || r.mpos)) o.rc = ZJ_ERR_HUFFMAN;
            if (!o.rc) o.end_bits = s0.dbits + s0.nbits + r.consumed();
        }
        o.t1 = std::chrono::duration<double, std::milli>(clk() - t_b).count();
        finished[q].store(1, std::memory_order_release);
        // (never the last piece from in here: its end is compared with the stitching's before anything of it leaves)
        if (d->stream.active && std::this_thread::get_id() == caller && NP) { (void)follow_links(NP - 1); stream_rows(d, kept); }
    });
    if (dbg) {
        const auto ms = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) { return std::chrono::duration<double, std::milli>(b - a).count(); };
        fprintf(stderr, "scan_baseline_parallel: %d threads, %lld MCUs (%zu decoded while stitching, %lld bridged in %zu bridges%s); structure %.2f ms, stitch %.2f ms, decode %.2f ms\n",
                T, total_mcus, walked, bridged, bridges.size(), gave_up ? ", then left to the serial walk" : "", ms(t_a, t_s), ms(t_s, t_b), ms(t_b, clk()));
        fprintf(stderr, "  looking for markers and counting stuffed zeros in front of pass A: %.3f ms\n", ms(t_in, t_a));
        fprintf(stderr, "  chunk 0: %lld MCUs for real, .. %.3f ms after the start\n", head_mcus, head.t1);
        for (int t = 1; t < T; t++) fprintf(stderr, "  chunk %d: %lld bytes of structure, .. %.3f ms\n", t, (long long)(start[(size_t)t + 1] - start[(size_t)t]), a_ms[(size_t)t]);
        for (size_t q = 0; q < NP; q++)
            fprintf(stderr, "  piece %zu%s: MCUs [%lld, %lld), rc %d, %.3f .. %.3f ms\n", q, pieces[q].bridge ? " (bridge)" : "", pieces[q].m0, pieces[q].m1,
                    pieces[q].out->rc, pieces[q].out->t0, pieces[q].out->t1);
    }
    if (NP && follow_links(NP - 1)) {
        // all links so far held: the last piece must also end where the stitching ended, or it is not kept
        const Out* const l0 = last;
        const long long k0 = kept;
        if (follow_links(NP) && (last->end_bits != cur.dbits || memcmp(last->pred, cur.dc, sizeof cur.dc) != 0)) { last = l0; kept = k0; }
    }
    return keep(*last, kept);
}

void fill_frame_desc(const zj_decoder* d, zj_frame_desc* fd);

// the walker has finished MCU rows [0, rows): hand the strips they complete to the GPU (no-op unless a stream is active)
inline void stream_rows(zj_decoder* d, long long mcus_done)
{
    if (!d->stream.active) return;
    const int rows = (int)(mcus_done / d->mcu_x);
    if (rows <= d->stream.rows) return;
    d->stream.rows = rows;
    if (zj_frame_rows_ready(d->stream.ctx, (size_t)rows) != ZJ_OK) { (void)zj_frame_abort(d->stream.ctx); d->stream.active = false; }
}
// before a baseline scan is walked on the CPU: start streaming if zj_decoder_decode_buffer armed it and everything fits
void stream_begin(zj_decoder* d)
{
    d->stream.active = false;
    d->stream.rows = 0;
    if (!d->stream.ctx || !d->stream.out) return;
    for (int i = 0; i < d->ncomp; i++) if (!d->store[i].pinned) return; // (copies out of pageable planes block the walker)
    zj_frame_desc fd;
    fill_frame_desc(d, &fd);
    if (fd.in_components == 1 && fd.out_colorspace != ZJ_CS_GRAYSCALE) return; // all-zero output: nothing is decoded
    if (zj_out_len(&fd) == 0 || d->stream.cap < zj_out_len(&fd)) return;       // (finish_impl reports it)
    d->stream.active = zj_frame_begin(d->stream.ctx, &fd, d->comps[0].coef, d->ncomp == 3 ? d->comps[1].coef : nullptr,
                                      d->ncomp == 3 ? d->comps[2].coef : nullptr, d->stream.out, 0) == ZJ_OK;
}

// ---- a baseline scan decoded the reference's way, LITERALLY (round 6) -------------------------------------------------------
// The walk above models the reference's reader where intact files and the usual damage need it: the early exit at EOI, the
// short DC reads with the bit history behind them, the restart boundaries.  One thing it does not follow bit by bit: a decode
// that runs PAST its data behind a marker (an RSTn or EOI in mid-interval, a stream that has lost its place).  The reference's
// refill then sets bits_left = 63 and serves what its rotating aligned_buffer happens to hold (src/bitstream.rs:254-258,
// 394-402): the zero gap of the last real refill, then the magnitude bits it has handed out since -- again and again.  The
// walk serves zeros; so the moment it consumes a bit it made up (BitReader::pad behind a marker) it gives up with
// ZJ_INT_NEED_LITERAL and the scan is decoded once more by this: the reference's BitStream and MCU loop restated field by
// field (buffer, aligned_buffer, bits_left, marker; src/bitstream.rs:158-258,264-402; src/mcu.rs:231-351,386-419), slow and
// exact, as oracle/ref_walk.py restates them in Python for the tests (tools/ref_walk_soak.py compares the two).
constexpr int ZJ_INT_NEED_LITERAL = 1001; // internal status, never leaves this file
struct RefStream {
    const uint8_t* buf; const uint8_t* end; const uint8_t* pos;
    uint64_t buffer = 0, aligned = 0;
    int bl = 0, marker = -1, last_len = 0, unknown = 0; // marker -1: none; unknown: a marker byte Marker::from_u8 has no name for
    uint32_t byte() { const uint32_t v = pos < end ? *pos : 0u; pos++; return v; } // read_u8: zeros past the end (:689-703)
    void realign() { aligned = bl ? buffer << (64 - bl) : buffer; }                // (x << 64 is x in release builds)
    bool refill() // false: "Unknown marker" (:199-206)
    {
        if (bl <= 32 && marker < 0) {
            if (pos + 4 < end && pos[0] != 0xFF && pos[1] != 0xFF && pos[2] != 0xFF && pos[3] != 0xFF) { // :220-236
                buffer = (buffer << 32) | ((uint64_t)pos[0] << 24 | (uint64_t)pos[1] << 16 | (uint64_t)pos[2] << 8 | pos[3]);
                pos += 4;
                bl += 32;
                realign();
                return true;
            }
            for (int i = 0; i < 4; i++) { // the refill! macro, :168-213
                const uint32_t b = byte();
                buffer = (buffer << 8) | b;
                bl += 8;
                if (b == 0xFF) {
                    uint32_t n = byte();
                    if (n != 0) {
                        while (n == 0xFF) n = byte();
                        if (n != 0) {
                            buffer >>= 8;
                            bl -= 8;
                            if (bl != 0) realign(); // :193-197
                            if (!reference_knows_marker((int)n)) { unknown = (int)n; return false; }
                            marker = (int)n;
                            return true;
                        }
                    }
                }
            }
            realign();
        } else if (marker >= 0) bl = 63; // :254-258: aligned_buffer stays as it is
        return true;
    }
    uint32_t peek(int n) const { return n ? (uint32_t)(aligned >> (64 - n)) : 0u; }                 // :378-382
    void drop(int n) { bl = bl > n ? bl - n : 0; aligned = n < 64 ? aligned << n : 0; }              // :386-390
    uint32_t get(int n)                                                                             // :394-402 (rotates)
    {
        if (!n) return 0;
        aligned = (aligned << n) | (aligned >> (64 - n));
        bl = bl > n ? bl - n : 0;
        return (uint32_t)(aligned & ((1ull << n) - 1));
    }
    int symbol(const Huff& h) // decode_huff!, :49-86: the first (length, code) of the table found in the top bits
    {
        for (int l = 1; l <= 16; l++) {
            const int32_t c = (int32_t)peek(l);
            if (h.maxcode[l] >= 0 && c <= h.maxcode[l] && c + h.valoff[l] >= 0) { drop(l); last_len = l; return h.vals[(c + h.valoff[l]) & 0xff]; }
        }
        return -1;
    }
    void reset() { buffer = aligned = 0; bl = 0; marker = -1; } // :671-678
};

int scan_baseline_literal(zj_decoder* d, const uint8_t* p0, const uint8_t* end)
{
    for (int i = 0; i < d->ncomp; i++) {
        Comp& cm = d->comps[i];
        memset(cm.coef, 0, (size_t)cm.bw * cm.bh * 128); // blocks the reference never reaches keep the zeros of its fresh vectors
    }
    RefStream s{p0, end, p0};
    const bool hsub = d->ncomp == 3 && d->h_max == 2;
    const long long width = hsub && d->v_max == 1 ? 2ll * d->mcu_x : d->mcu_x;     // mcu.rs:145-152
    const long long height = hsub ? d->mcu_y / 2 : d->mcu_y;
    const int bias = hsub && d->v_max == 2 ? 2 : 1;
    const bool full = (d->flags & ZJ_FLAG_FULL_AC_VALUES) != 0;
    long long todo = d->restart_interval ? d->restart_interval : (1ll << 62);
    uint32_t pred[4] = {0, 0, 0, 0};
    char text[48];
    long long mcu = 0;
    for (long long row = 0; row < height; row++)
        for (int pass = 0; pass < bias; pass++) {
            const long long mcu_pass0 = mcu;
            for (long long i = 0; i < width; i++) {
                const int mx = (int)(mcu % d->mcu_x), my = (int)(mcu / d->mcu_x);
                for (int ci = 0; ci < d->ns; ci++) {
                    const int c = d->order[ci];
                    Comp& cm = d->comps[c];
                    const Huff& hd = d->dc[cm.td & 3];
                    const Huff& ha = d->ac[cm.ta & 3];
                    for (int b = 0; b < cm.h * cm.v; b++) {
                        if (s.bl < 16 && !s.refill()) goto unknown_marker;                 // decode_dc, :264-296
                        int size = s.symbol(hd);
                        if (size < 0 || size > 16) return fail(d, ZJ_ERR_HUFFMAN, "Bad Huffman code in DC");
                        if (size) {
                            const uint32_t bits = s.get(size);
                            pred[c] += (uint32_t)extend((int32_t)bits, size);
                        }
                        int16_t* blk = block_at(cm, mx * cm.h + b % cm.h, my * cm.v + b / cm.h);
                        blk[0] = (int16_t)(uint16_t)pred[c];
                        int pos = 1;
                        while (pos < 64) {                                                  // :332-372
                            if (!s.refill()) goto unknown_marker;
                            const int rs = s.symbol(ha);
                            if (rs < 0) return fail(d, ZJ_ERR_HUFFMAN, "Bad Huffman code in AC");
                            const int r = rs >> 4;
                            size = rs & 15;
                            if (size) {
                                pos += r;
                                bool fast = false;
                                if (s.last_len + size <= 9) { // the fast-AC table: code + magnitude inside 9 bits, value inside a byte
                                    const int32_t v = extend((int32_t)s.peek(size), size);
                                    fast = v >= -128 && v <= 127;
                                }
                                uint32_t bits;
                                if (fast) { bits = s.peek(size); s.drop(size); } // ONE drop_bits for both (:339-347)
                                else bits = s.get(size);
                                const int32_t v = extend((int32_t)bits, size);
                                blk[kUnZigzag[fast ? (pos < 63 ? pos : 63) : (pos & 63)]] = (int16_t)(fast && !full ? sext6(v) : v);
                                pos += 1;
                            } else if (s.last_len <= 9) { // size 0 in the fast table too (src/huffman.rs:217-233)
                                pos += r == 0 ? 63 : r;
                                blk[kUnZigzag[pos < 63 ? pos : 63]] = 0;
                                pos += 1;
                            } else if (r != 15) break;    // the general path (:365-367)
                            else pos += 16;
                        }
                    }
                }
                mcu++;
                if (--todo == 0) { // handle_rst, mcu.rs:386-419
                    todo = d->restart_interval;
                    if (s.marker >= 0xD0 && s.marker <= 0xD7) { s.reset(); pred[0] = pred[1] = pred[2] = pred[3] = 0; }
                    else if (s.marker >= 0 && s.marker != 0xD9) return fail(d, ZJ_ERR_MCU, "Marker found in bitstream, possibly corrupt jpeg");
                }
                if (s.marker == 0xD9) break; // mcu.rs:337-343
            }
            mcu = mcu_pass0 + width; // the MCUs a cut pass never decoded keep their zeros; the next pass starts behind them
        }
    for (int i = 0; i < d->ncomp; i++) d->comps[i].dc_pred = (int32_t)pred[i];
    return ZJ_OK;
unknown_marker:
    snprintf(text, sizeof text, "Unknown marker 0xFF%X", s.unknown);
    return fail(d, ZJ_ERR_FORMAT, text);
}

int scan_baseline(zj_decoder* d, BitReader& br)
{
    for (int i = 0; i < d->ncomp; i++) {
        d->comps[i].dc_pred = 0;
        if (!d->dc[d->comps[i].td & 3].present) return fail(d, ZJ_ERR_HUFFMAN, "No DC table for component " + std::to_string(d->comps[i].id));
        if (!d->ac[d->comps[i].ta & 3].present) return fail(d, ZJ_ERR_HUFFMAN, "No AC table for component " + std::to_string(d->comps[i].id));
    }
    if (d->ns != d->ncomp) return fail(d, ZJ_ERR_UNSUPPORTED, "baseline scans must carry every component (src/mcu.rs:253-321)");
    const long long total = (long long)d->mcu_x * d->mcu_y;
    if (d->threads > 1 && d->restart_interval > 0 && total > d->restart_interval) {
        // well-formed restart structure: decode the segments concurrently; anything unusual (missing or
        // out-of-sequence markers, a bad code) is left to the serial walk below, which owns the error text
        const int ri = d->restart_interval;
        const int nseg = (int)((total + ri - 1) / ri);
        std::vector<const uint8_t*> seg;
        // (the scan must close with EOI: behind a full last interval handle_restart() looks at whatever marker is there, and
        // anything but RSTn / EOI is "Marker found in bitstream" -- the serial walk's to say)
        if (find_restart_segments(br.p, br.end, nseg, seg) && seg[(size_t)nseg] + 1 < br.end && seg[(size_t)nseg][0] == 0xFF && seg[(size_t)nseg][1] == 0xD9) {
            std::atomic<int> bad{0};
            d->crew.each(nseg, d->threads, [&](int k) {
                const char* err = nullptr;
                const long long m0 = (long long)k * ri, n = m0 + ri <= total ? ri : total - m0;
                // a segment ends where the next RSTn (or the closing marker) begins: the reader stops there
                const uint8_t* eoi = nullptr; // the last segment ends at the scan's closing marker
                if (k == nseg - 1 && seg[(size_t)nseg] + 1 < br.end && seg[(size_t)nseg][1] == 0xD9) {
                    eoi = seg[(size_t)nseg];
                    while (eoi > seg[(size_t)k] && eoi[-1] == 0xFF) eoi--;
                }
                if (scan_baseline_segment(d, d, seg[(size_t)k], seg[(size_t)k + 1], m0, n, &err, eoi, k != nseg - 1)) bad.store(1);
            });
            if (!bad.load()) { br.p = seg[(size_t)nseg]; br.reset(); d->dri_parallel_segments = nseg; return ZJ_OK; }
            // (the serial walk below clears every block again before it writes into it)
        }
    }
    d->dri_parallel_segments = 0;
    const BlockFns fn = block_fns(d->plane_store);
    int todo = d->restart_interval ? d->restart_interval : 0x7fffffff;
    EoiCut cut;
    {
        bool is_eoi = false;
        const uint8_t* e = find_scan_end(br.p, br.end, &is_eoi);
        cut.eoi = (e && is_eoi) ? e : nullptr;
        if (e && !is_eoi) {
            const uint8_t* t = e;
            while (t + 1 < br.end && t[1] == 0xFF) t++;
            const int m = t + 1 < br.end ? t[1] : 0xD9;
            if (!reference_knows_marker(m)) { cut.eoi = e; cut.unknown = m; }
        }
        cut.rowlen = eoi_rowlen(d);
    }
    // blocks of MCUs from (mx, my) on that the walk never reached stay zero, like the reference's fresh vectors
    auto clear_from = [&](int my0, int mx0) {
        _mm_sfence(); // order the streaming stores of the blocks decoded so far before the ordinary stores below
        for (int my = my0; my < d->mcu_y; my++)
            for (int mx = (my == my0 ? mx0 : 0); mx < d->mcu_x; mx++)
                for (int ci = 0; ci < d->ncomp; ci++) {
                    Comp& cm = d->comps[ci];
                    for (int v = 0; v < cm.v; v++)
                        for (int h = 0; h < cm.h; h++) memset(block_at(cm, mx * cm.h + h, my * cm.v + v), 0, 128);
                }
    };
    // (2,1): the reference walks 2*mcu_x MCUs per strip (mcu.rs:145-152); MCU order is unchanged
    const uint8_t* const stop = cut.eoi ? cut.eoi - (cut.eoi - br.p > 4096 ? 4096 : cut.eoi - br.p) : nullptr;
    long long m_first = 0;
    d->par_scan_mcus = 0;
    if (d->threads > 1 && !d->restart_interval && fn.mcus && !d->track_hist && d->plane_store == STORE_DIRECT && cut.eoi) {
        const char* e = getenv("ZJ_PAR_SCAN");
        if (!(e && (!strcmp(e, "off") || !strcmp(e, "0")))) {
            m_first = scan_baseline_parallel(d, br, fn, cut.eoi);
            d->par_scan_mcus = m_first;
            if (!m_first) for (int i = 0; i < d->ncomp; i++) d->comps[i].dc_pred = 0;
            stream_rows(d, m_first);
        }
    }
    // With horizontally sub-sampled chroma the reference walks its MCU rows in pairs -- mcu_height / 2 passes over two rows
    // (src/mcu.rs:145-152,225-231) -- so an odd last MCU row is never decoded (the pixel path leaves its rows zero: zj_plan.h).
    // The walk below still fills that row's coefficients, but whatever is wrong with its data is nobody's business: the
    // reference never reads it (found by tools/ref_walk_soak.py in round 6: a damaged last row made this an error).
    const long long walked_total = (d->ncomp == 3 && d->h_max == 2) ? (long long)d->mcu_x * (d->mcu_y & ~1) : total;
    static const bool literal_allowed = [] { const char* e = getenv("ZJ_LITERAL"); return !(e && (!strcmp(e, "off") || !strcmp(e, "0"))); }();
    for (long long m = m_first; m < total;) {
        if (fn.mcus && !d->track_hist && !cut.seen) {
            // as many MCUs as possible in one go (decode_mcus_v2): up to the next restart boundary, short of the scan's last 4 KB.
            // eoi_cut_after_mcu has nothing to do for them (it starts looking 64 bytes in front of the EOI marker).
            int32_t pred[3] = {d->comps[0].dc_pred, d->comps[1].dc_pred, d->comps[2].dc_pred};
            long long done = 0;
            const char* err = nullptr;
            long long n = total - m < todo ? total - m : (long long)todo;
            if (d->stream.active && n > 4ll * d->mcu_x) n = 4ll * d->mcu_x; // come back every few rows: the GPU takes what is final
            if (m < walked_total && n > walked_total - m) n = walked_total - m;  // (the row the reference never reads: on its own)
            const int rc = fn.mcus(d, d, br, pred, m, n, stop, &done, &err);
            for (int i = 0; i < 3; i++) d->comps[i].dc_pred = pred[i];
            m += done;
            if (rc) { clear_from((int)(m / d->mcu_x), (int)(m % d->mcu_x)); if (m >= walked_total) break; return fail(d, rc, err); }
            if (done) {
                // bits this reader made up behind a marker have been consumed: the reference serves its rotating buffer there
                if (br.marker && br.nbits < br.pad && m <= walked_total && literal_allowed) return ZJ_INT_NEED_LITERAL;
                todo -= (int)done;
                if (todo == 0) {
                    const int rc2 = handle_restart(d, br, todo);
                    if (rc2) { clear_from((int)(m / d->mcu_x), (int)(m % d->mcu_x)); if (m > walked_total) { d->err.clear(); d->err_code = 0; break; } return rc2; } // (m: one past the interval's last MCU)
                }
                stream_rows(d, m);
                continue;
            }
        }
        const int my = (int)(m / d->mcu_x), mx = (int)(m % d->mcu_x);
        // the reference left this row's loop at an earlier MCU (see EoiCut): the block keeps its zeros
        if (cut.seen && m / cut.rowlen == cut.cut_row) { clear_mcu(d, mx, my); m++; continue; }
        const bool near_end = cut.eoi && cut.eoi - br.p <= 4096; // an MCU is at most 6 blocks x 64 x 27 bits = 1.3 KB
        for (int ci = 0; ci < d->ns; ci++) {
            Comp& cm = d->comps[d->order[ci]];
            for (int v = 0; v < cm.v; v++)
                for (int h = 0; h < cm.h; h++) {
                    const char* err = nullptr;
                    int16_t* blk = block_at(cm, mx * cm.h + h, my * cm.v + v);
                    int rc = (d->track_hist ? fn.hist : near_end ? fn.track : fn.hot)(d, br, cm, cm.dc_pred, blk, &err);
                    if (rc) { clear_from(my, mx); if (m >= walked_total) goto walk_done; return fail(d, rc, err); }
                }
        }
        if (br.marker && br.nbits < br.pad && m < walked_total && literal_allowed) return ZJ_INT_NEED_LITERAL;
        bool restarted = false;
        if (--todo == 0) {
            restarted = br.marker >= 0xD0 && br.marker <= 0xD7;
            int rc = handle_restart(d, br, todo);
            if (rc) { clear_from(my, mx + 1); if (m >= walked_total) { d->err.clear(); d->err_code = 0; goto walk_done; } return rc; }
        }
        if (!restarted) eoi_cut_after_mcu(cut, br, m); // (a restart clears the reference's pending marker, mcu.rs:400-408)
        if (cut.seen && cut.unknown) {
            char text[40];
            snprintf(text, sizeof text, "Unknown marker 0xFF%X", cut.unknown);
            clear_from(my, mx + 1);
            if (m >= walked_total) goto walk_done;
            return fail(d, ZJ_ERR_FORMAT, text);
        }
        m++;
        if (m % d->mcu_x == 0) stream_rows(d, m);
    }
walk_done:
    _mm_sfence(); // the blocks left with streaming stores (decode_block_baseline)
    return ZJ_OK;
}

int dc_first(zj_decoder* d, BitReader& br, Comp& cm, int16_t* blk)
{
    if (br.nbits < 32) br.fill();
    const int before = br.nbits;
    int s = br.decode(d->dc[cm.td & 3]);
    if (s < 0 || s > 16) return fail(d, ZJ_ERR_HUFFMAN, "Bad Huffman code in DC");
    const int len = before - br.nbits;
    // a DC scan is nothing but decode_dc calls (bitstream.rs:407-415 -> :264): bits_left is refilled below 16 only
    if (br.rbl < 16) { br.rbl += 32; br.rbl0 = br.rbl; }
    br.rhist <<= len;
    int32_t diff = 0, short_bits = 0;
    if (s) {
        if (len + s > br.rbl && ref_dc_misread(br, br.rbl, br.rbl0, br.rhist, len, s, &short_bits)) diff = extend(short_bits, s);
        else { const int32_t mag = br.get(s); diff = extend(mag, s); br.rbl -= len + s; br.rhist = (br.rhist << s) | (uint32_t)mag; }
    } else br.rbl -= len;
    cm.dc_pred = (int32_t)((uint32_t)cm.dc_pred + (uint32_t)diff);
    blk[0] = (int16_t)((uint16_t)(int16_t)cm.dc_pred * (uint16_t)(1u << d->al)); // bitstream.rs:413
    return ZJ_OK;
}
inline void dc_refine(zj_decoder* d, BitReader& br, int16_t* blk)
{
    if (br.get(1)) blk[0] = (int16_t)(blk[0] | (1 << d->al));
}
int ac_first(zj_decoder* d, BitReader& br, const Huff& ha, int16_t* blk)
{
    if (d->eobrun > 0) { d->eobrun--; return ZJ_OK; }
    for (int k = d->ss; k <= d->se;) {
        if (br.nbits < 32) br.fill();
        const int16_t fa = ha.fast[br.peek(9)];
        if (fa) { // short code + small value from one table entry (same shortcut as the baseline scan)
            k += (fa >> 4) & 15;
            br.drop(fa & 15);
            blk[kUnZigzag[k < 63 ? k : 63]] = (int16_t)((uint16_t)(int16_t)(fa >> 8) * (uint16_t)(1u << d->al)); // min(k, 63): bitstream.rs:466
            k++;
            continue;
        }
        int rs = br.decode(ha);
        if (rs < 0) return fail(d, ZJ_ERR_HUFFMAN, "Bad Huffman code in AC");
        int r = rs >> 4, s = rs & 15;
        if (s) {
            k += r;
            int32_t v = extend(br.get(s), s);
            blk[kUnZigzag[k & 63]] = (int16_t)((uint16_t)(int16_t)v * (uint16_t)(1u << d->al));
            k++;
        } else if (r == 15) {
            k += 16;
        } else {
            d->eobrun = (1u << r) - 1;
            if (r) d->eobrun += (uint32_t)br.get(r);
            break;
        }
    }
    return ZJ_OK;
}
int ac_refine(zj_decoder* d, BitReader& br, const Huff& ha, int16_t* blk)
{
    const int16_t p1 = (int16_t)(1 << d->al), m1 = (int16_t)(-1 * (1 << d->al));
    int k = d->ss;
    if (d->eobrun == 0) {
        for (; k <= d->se; k++) {
            int rs = br.decode(ha);
            if (rs < 0) return fail(d, ZJ_ERR_HUFFMAN, "Bad Huffman code in AC");
            int r = rs >> 4, s = rs & 15;
            int16_t val = 0;
            if (s) {
                val = br.get(1) ? p1 : m1; // s must be 1
            } else if (r != 15) {
                d->eobrun = 1u << r;
                if (r) d->eobrun += (uint32_t)br.get(r);
                break;
            }
            // skip r zero-history coefficients, refining the non-zero ones passed (T.81 G.1.2.3)
            while (k <= d->se) {
                int16_t* c = blk + kUnZigzag[k];
                if (*c != 0) {
                    if (br.get(1) && (*c & p1) == 0) *c = (int16_t)(*c >= 0 ? *c + p1 : *c + m1);
                } else {
                    if (--r < 0) break;
                }
                k++;
            }
            if (s && k <= d->se) blk[kUnZigzag[k]] = val;
        }
    }
    if (d->eobrun > 0) {
        // Inside an end-of-band run only the coefficients that are already non-zero take a correction bit.  Most blocks of a
        // run have none (the run exists because the block is flat): one look at the block's 128 bytes instead of 63 loads
        // and tests (test-progressive.jpg: its four AC refinement scans were three quarters of the file's decode time).
        if (k == d->ss && k >= 1) {
            const __m128i* q = (const __m128i*)blk; // (blocks are 128-byte aligned in their plane)
            __m128i any = _mm_and_si128(_mm_load_si128(q), _mm_set_epi16(-1, -1, -1, -1, -1, -1, -1, 0)); // not the DC
            for (int i = 1; i < 8; i++) any = _mm_or_si128(any, _mm_load_si128(q + i));
            if (_mm_movemask_epi8(_mm_cmpeq_epi8(any, _mm_setzero_si128())) == 0xFFFF) { d->eobrun--; return ZJ_OK; }
        }
        for (; k <= d->se; k++) {
            int16_t* c = blk + kUnZigzag[k];
            if (*c != 0 && br.get(1) && (*c & p1) == 0) *c = (int16_t)(*c >= 0 ? *c + p1 : *c + m1);
        }
        d->eobrun--;
    }
    return ZJ_OK;
}

int scan_progressive(zj_decoder* d, BitReader& br)
{
    for (int i = 0; i < d->ncomp; i++) d->comps[i].dc_pred = 0;
    d->eobrun = 0;
    int todo = d->restart_interval ? d->restart_interval : 0x7fffffff;
    if (d->ns == 1) {
        if (d->se != 0 && d->ss == 0) return fail(d, ZJ_ERR_HUFFMAN, "Can't merge DC and AC corrupt jpeg");
        Comp& cm = d->comps[d->order[0]];
        // non-interleaved scan: only the blocks that cover the image (mcu_prog.rs:277-289)
        int bw, bh;
        if (d->order[0] == 0 || (d->h_max == 1 && d->v_max == 1)) { bw = (d->width + 7) / 8; bh = (d->height + 7) / 8; }
        else { bw = d->mcu_x; bh = d->mcu_y; }
        if (d->ss == 0) { if (d->ah == 0 && !d->dc[cm.td & 3].present) return fail(d, ZJ_ERR_FORMAT, "Huffman table at index  " + std::to_string(cm.td) + " not initialized"); }
        else if (!d->ac[cm.ta & 3].present) return fail(d, ZJ_ERR_FORMAT, "Huffman table at index  " + std::to_string(cm.ta) + " not initialized");
        const Huff& ha = d->ac[cm.ta & 3];
        for (int by = 0; by < bh; by++)
            for (int bx = 0; bx < bw; bx++) {
                int16_t* blk = block_at(cm, bx, by);
                int rc = ZJ_OK;
                if (d->ss == 0) { if (d->ah == 0) rc = dc_first(d, br, cm, blk); else dc_refine(d, br, blk); }
                else if (d->ah == 0) rc = ac_first(d, br, ha, blk);
                else rc = ac_refine(d, br, ha, blk);
                if (rc) return rc;
                if (--todo == 0) { rc = handle_restart(d, br, todo); if (rc) return rc; }
            }
    } else {
        if (d->se != 0) return fail(d, ZJ_ERR_HUFFMAN, "Can't merge dc and AC corrupt jpeg");
        for (int i = 0; i < d->ns; i++)
            if (d->ah == 0 && !d->dc[d->comps[d->order[i]].td & 3].present)
                return fail(d, ZJ_ERR_FORMAT, "Huffman table at index  " + std::to_string(d->comps[d->order[i]].td) + " not initialized");
        for (int my = 0; my < d->mcu_y; my++)
            for (int mx = 0; mx < d->mcu_x; mx++) {
                for (int ci = 0; ci < d->ns; ci++) {
                    Comp& cm = d->comps[d->order[ci]];
                    for (int v = 0; v < cm.v; v++)
                        for (int h = 0; h < cm.h; h++) {
                            int16_t* blk = block_at(cm, mx * cm.h + h, my * cm.v + v);
                            if (d->ah == 0) { int rc = dc_first(d, br, cm, blk); if (rc) return rc; }
                            else dc_refine(d, br, blk);
                        }
                }
                if (--todo == 0) { int rc = handle_restart(d, br, todo); if (rc) return rc; }
            }
    }
    return ZJ_OK;
}

// after a scan: find the next marker (mcu_prog.rs:436-472)
// ---- the scan as the GPU entropy stage wants it (zj_huff.h) --------------------------------------------------------
// Two-level decoding table: 2^B first-level entries indexed by the next B bits (B = 9 for DC, 11 for AC tables); codes
// longer than B bits hang off their B-bit prefix in a second level of 2^(16-B) entries.  Returns the entries used, or
// -1 if they exceed `room`.
int gpu_table(const Huff& h, bool ac, uint16_t* out, int room)
{
    const int B = ac ? zj::HUFF_L1_AC : zj::HUFF_L1_DC, L1 = 1 << B, L2 = 1 << (16 - B);
    constexpr uint16_t NONE = 16; // "no such code": 16 bits consumed, zig-zag advance 0 (zj_huff.h)
    if (room < L1) return -1;
    for (int q = 0; q < L1; q++) out[q] = NONE;
    uint32_t code = 0;
    int k = 0, links = 0;
    for (int l = 1; l <= 16; l++) {
        for (int i = 0; i < h.nlen[l]; i++, k++, code++) {
            // entry: bits consumed | zig-zag advance << 5 | magnitude bits << 11 (zj_huff.h)
            const int sym = h.vals[k], sz = ac ? (sym & 15) : sym, run = ac ? sym >> 4 : 0;
            const int zadv = !ac ? 1 : sz ? run + 1 : run == 15 ? 16 : 63;
            // AC symbols the device stage does not read the reference's way count as "no such code": if one OCCURS the write
            // pass raises HUFF_ST_BAD_CODE and the CPU walker decodes the file.  The device computes every value from the bits
            // and ends the block at any size-0 symbol but ZRL; the reference's fast-AC table skips run + 1 coefficients at a
            // size-0 run of 1..14 behind a code of up to 9 bits, and keeps six bits of a value of size 6..8 that fits its
            // 9-bit window (Huff::ac_entry, Huff::fast).  Tables often list such symbols (tools/jpeg_enc.py lists all 256);
            // scans written by an encoder do not use the former and need a 1..3-bit code for the latter.
            const bool walker_only = ac && ((!sz && run != 0 && run != 15 && l <= 9) || (sz >= 6 && l + sz <= 9 && !h.full_values));
            const uint16_t e = (sz > 15 || l + sz > 31 || walker_only) ? NONE : (uint16_t)((l + sz) | (zadv << 5) | (sz << 11));
            if (l <= B) {
                const uint32_t base = code << (B - l);
                for (uint32_t q = 0; q < (1u << (B - l)); q++) out[base + q] = e;
            } else {
                const uint32_t prefix = code >> (l - B);
                if (out[prefix] == NONE) {
                    if (links == 255 || L1 + (links + 1) * L2 > room) return -1;
                    for (int q = 0; q < L2; q++) out[L1 + links * L2 + q] = NONE;
                    out[prefix] = (uint16_t)(0x8000 | links++);
                }
                const uint32_t sub = (uint32_t)L1 + ((out[prefix] & 0xffu) << (16 - B));
                const uint32_t low = (code << (16 - l)) & (uint32_t)(L2 - 1);
                for (uint32_t q = 0; q < (1u << (16 - l)); q++) out[sub + low + q] = e;
            }
        }
        code <<= 1;
    }
    return L1 + links * L2;
}

// Copies entropy-coded bytes from s without the zero that follows every 0xFF, up to the first marker (a 0xFF followed by
// anything but 0x00; fill 0xFF bytes in front of it belong to it) or `end`.  *marker: the marker's first 0xFF, or null.
// 16 bytes at a time while none of them is 0xFF (seven of eight such groups in coded data); may write up to 15 bytes of
// garbage behind the length it returns (the caller's buffer has the slack and clears it).
size_t unstuff_segment(const uint8_t* s, const uint8_t* end, uint8_t* dst, const uint8_t** marker)
{
    uint8_t* const d0 = dst;
    const __m128i ff = _mm_set1_epi8((char)0xFF);
    *marker = nullptr;
    while (s < end) {
        if (end - s >= 16) {
            const __m128i v = _mm_loadu_si128((const __m128i*)s);
            const int m = _mm_movemask_epi8(_mm_cmpeq_epi8(v, ff));
            _mm_storeu_si128((__m128i*)dst, v);
            if (!m) { s += 16; dst += 16; continue; }
            const int k = __builtin_ctz((unsigned)m); // bytes in front of the first 0xFF: already stored
            s += k; dst += k;
        } else if (*s != 0xFF) { *dst++ = *s++; continue; }
        // *s == 0xFF
        if (s + 1 < end && s[1] == 0x00) { *dst++ = 0xFF; s += 2; continue; }
        *marker = s;
        break;
    }
    return (size_t)(dst - d0);
}

// ZJ_OK: d->blob_store holds the scan.  ZJ_ERR_UNSUPPORTED: not a scan for the device (not an error: the caller runs
// the CPU walker, which also owns every message about damaged files).
int prepare_scan(zj_decoder* d, const uint8_t* p, const uint8_t* end)
{
    using namespace zj;
    d->scan_ready = false;
    auto why = [](int n) { if (getenv("ZJ_HUFF_DEBUG")) fprintf(stderr, "prepare_scan: left to the CPU walker (reason %d)\n", n); return (int)ZJ_ERR_UNSUPPORTED; };
    if (d->progressive || d->ns != d->ncomp) return why(1);
    for (int i = 0; i < d->ncomp; i++)
        if (!d->dc[d->comps[i].td & 3].present || !d->ac[d->comps[i].ta & 3].present) return why(2);
    const long long total = (long long)d->mcu_x * d->mcu_y;
    int nseg = 1;
    long long ri = total;
    bool is_eoi = false;
    if (d->restart_interval > 0 && total > d->restart_interval) {
        ri = d->restart_interval;
        nseg = (int)((total + ri - 1) / ri);
    }
    const size_t file_left = (size_t)(end - p); // the scan's bytes and whatever follows them: an upper bound
    if (d->entropy < 2 && file_left < (size_t)32 << 10) return why(5); // not worth a trip
    if (file_left > (size_t)200 << 20 || (size_t)nseg > HUFF_SEG_MASK / 2) return why(6); // 32-bit bit positions
    // the MCU's blocks in scan order, distinct tables
    HuffScan h;
    memset(&h, 0, sizeof h);
    int ntab = 0, tab_key[HUFF_MAX_TABS], tab_off[HUFF_MAX_TABS], tab_used = 0;
    std::vector<uint16_t> tabs((size_t)HUFF_TAB_BUDGET);
    auto table_index = [&](int cls, int id) { // entry offset of the table, -1: the tables do not fit the device's LDS budget
        const int key = cls * 4 + id;
        for (int t = 0; t < ntab; t++) if (tab_key[t] == key) return tab_off[t];
        const int n = gpu_table(cls ? d->ac[id] : d->dc[id], cls != 0, tabs.data() + tab_used, HUFF_TAB_BUDGET - tab_used);
        if (n < 0) return -1;
        tab_key[ntab] = key;
        tab_off[ntab++] = tab_used;
        tab_used += n;
        return tab_used - n;
    };
    int bpm = 0;
    for (int ci = 0; ci < d->ns; ci++) {
        const Comp& cm = d->comps[d->order[ci]];
        const int tdc = table_index(0, cm.td & 3), tac = table_index(1, cm.ta & 3);
        if (tdc < 0 || tac < 0) return why(12);
        h.dc_off[d->order[ci]] = (uint16_t)tdc;
        h.ac_off[d->order[ci]] = (uint16_t)tac;
        for (int v = 0; v < cm.v; v++)
            for (int hh = 0; hh < cm.h; hh++) {
                if (bpm == HUFF_MAX_BPM) return why(7);
                h.comp_of_blk |= (uint32_t)d->order[ci] << (2 * bpm);
                HuffBlk& b = h.blk[bpm++];
                b.comp = (uint8_t)d->order[ci]; b.hx = (uint8_t)hh; b.vy = (uint8_t)v;
            }
    }
    if (d->entropy < 2 && bpm > 1) {
        // When every block of the MCU decodes with the same pair of tables the parse does not depend on the block's
        // place in the MCU, so a wrong guess of that place never corrects itself: it only heals one sub-sequence per
        // round from the front.  No encoder of YCbCr files shares luma and chroma tables; leave the odd one to the CPU.
        bool same = true;
        for (int b = 1; b < bpm; b++)
            same = same && h.dc_off[h.blk[b].comp] == h.dc_off[h.blk[0].comp] && h.ac_off[h.blk[b].comp] == h.ac_off[h.blk[0].comp];
        if (same) return why(13);
    }
    for (int i = 0; i < d->ncomp; i++) {
        h.comp[i].h = (uint32_t)d->comps[i].h; h.comp[i].v = (uint32_t)d->comps[i].v;
        h.comp[i].bw = (uint32_t)d->comps[i].bw; h.comp[i].bh = (uint32_t)d->comps[i].bh;
    }
    const double blocks = (double)total * bpm;
    const size_t sub = (size_t)d->sub_bytes;
    const size_t max_sub = file_left / sub + (size_t)nseg + 1;
    auto up16 = [](size_t x) { return (x + 15) & ~(size_t)15; };
    const size_t off_tab = up16(sizeof(HuffScan));
    const size_t off_sub = up16(off_tab + (size_t)tab_used * 2);
    const size_t off_per = up16(off_sub + (max_sub + 1) * sizeof(HuffSub));
    const size_t off_seg = up16(off_per + (max_sub + 1) * sizeof(uint32_t));
    const size_t off_stream = up16(off_seg + (size_t)nseg * sizeof(HuffSeg));
    const size_t cap = off_stream + file_left + 16 * (size_t)nseg + 64;
    uint8_t* blob = (uint8_t*)d->blob_store.ensure(cap, d->pinned);
    if (!blob) return why(8);
    memcpy(blob + off_tab, tabs.data(), (size_t)tab_used * 2);
    HuffSub* subs = (HuffSub*)(blob + off_sub);
    HuffSeg* segs = (HuffSeg*)(blob + off_seg);
    uint8_t* stream = blob + off_stream;
    const bool dbg = getenv("ZJ_HUFF_DEBUG") != nullptr;
    const auto tt0 = std::chrono::steady_clock::now();
    // One pass over the scan: unstuff, stop at every marker.  Exactly the nseg - 1 RSTn of a well-formed scan, in
    // sequence, then any other marker (or the end of the file); anything else is left to the CPU walker.
    size_t o = 0, nsub = 0;
    const uint8_t* at = p;
    for (int k = 0; k < nseg; k++) {
        const uint8_t* mk = nullptr;
        const size_t len = unstuff_segment(at, end, stream + o, &mk);
        if (mk) {
            const uint8_t* t = mk;
            while (t + 1 < end && t[1] == 0xFF) t++; // fill bytes
            const int m = t + 1 < end ? t[1] : 0xD9; // (a lone 0xFF at the very end: like the walker, as if EOI followed)
            if (m == 0x00) return why(11);           // 0xFF 0xFF 0x00: fill bytes in front of a stuffed byte
            if (m >= 0xD0 && m <= 0xD7) {
                if (k + 1 == nseg || m != 0xD0 + (k & 7)) return why(3); // one too many, or out of sequence
                at = t + 2;
            } else {
                if (k + 1 != nseg) return why(3); // the scan ends early
                is_eoi = m == 0xD9;
                // (behind a full last restart interval handle_restart() looks at whatever marker is there: anything but RSTn /
                // EOI is "Marker found in bitstream", src/mcu.rs:386-419 -- the CPU walker's to say)
                if (!is_eoi && nseg > 1) return why(15);
                // (a marker the reference has no name for ends its decode in "Unknown marker" once its reader comes across
                // it -- scan_baseline's EoiCut decides when)
                if (!is_eoi && !reference_knows_marker(m)) return why(16);
                at = mk;
            }
        } else {
            if (k + 1 != nseg) return why(4);
            at = end;
        }
        segs[k].start = (uint32_t)o;
        segs[k].end = (uint32_t)(o + len);
        const size_t first = nsub;
        for (size_t b = 0; b == 0 || b < len; b += sub) {
            subs[nsub].start = (uint32_t)(o + b);
            subs[nsub].seg = (uint32_t)k;
            nsub++;
        }
        subs[first].seg |= HUFF_FIRST;
        subs[nsub - 1].seg |= HUFF_LAST;
        const size_t next = up16(o + len);
        memset(stream + o + len, 0, next - (o + len));
        o = next;
    }
    memset(stream + o, 0, 32);
    const auto tt1 = std::chrono::steady_clock::now();
    const size_t scan_bytes = (size_t)(at - p);
    if (d->entropy < 2 && scan_bytes < (size_t)32 << 10) return why(5); // not worth a trip
    // Low-entropy scans that are not exactly periodic still crawl (the reference's test-baseline.jpg: 81 rounds for 72 KB);
    // they are also the ones the CPU walker is quickest with (few symbols per block), so they stay there: under 16 bits
    // per block on average.  The round budget is what the CPU walker would need instead, in rounds.
    if (d->entropy < 2 && (double)scan_bytes * 8.0 < 16.0 * blocks) return why(14);
    const double cpu_us = (double)scan_bytes * 0.008 + blocks * 0.05, round_us = 10.0 + 0.27 * (double)sub;
    long long budget = d->entropy >= 2 ? HUFF_MAX_ROUNDS : (long long)(cpu_us / round_us);
    if (budget < 16) budget = 16;
    if (budget > HUFF_MAX_ROUNDS) budget = HUFF_MAX_ROUNDS;
    // Periodic runs (zj_huff.h): sub-sequence i holds the bytes of sub-sequence i - q.  Only whole sub-sequences that
    // are not the last of their segment (the parse of one looks up to 31 bits into the next).
    uint32_t* per = (uint32_t*)(blob + off_per);
    memset(per, 0, (nsub + 1) * sizeof(uint32_t));
    size_t nper = 0;
    {
        std::vector<uint8_t> qof(nsub, 0);
        std::vector<uint64_t> head(nsub + 1); // first 8 bytes of every sub-sequence: almost always enough to say no
        for (size_t i = 0; i < nsub; i++) memcpy(&head[i], stream + subs[i].start, 8);
        for (size_t i = 1; i + 1 < nsub; i++) {
            if ((subs[i].seg & (HUFF_FIRST | HUFF_LAST)) || (subs[i + 1].seg & HUFF_FIRST)) continue;
            const uint64_t mine = head[i];
            const size_t qmax = i < HUFF_PER_MAXQ ? i : HUFF_PER_MAXQ;
            for (size_t q = 1; q <= qmax; q++) {
                if (head[i - q] != mine) continue;
                if ((subs[i - q].seg & HUFF_SEG_MASK) != (subs[i].seg & HUFF_SEG_MASK)) break;
                if (memcmp(stream + subs[i].start, stream + subs[i - q].start, sub) == 0) { qof[i] = (uint8_t)q; break; }
            }
        }
        for (size_t i = 1; i < nsub;) {
            if (!qof[i]) { i++; continue; }
            const size_t q = qof[i], s0 = i; // a stretch [s0, e] of consecutive sub-sequences with the same q
            size_t e = i;
            while (e + 1 < nsub && qof[e + 1] == q) e++;
            const size_t r0 = s0 - q; // the run: [r0, e]; predicted: [r0 + 2q, e - 1] (e looks into e + 1, which is not part of it)
            for (size_t k = r0 + 2 * q; k + 1 <= e; k++) { per[k] = (uint32_t)((q << HUFF_PER_QSHIFT) | r0); nper++; }
            i = e + 1;
        }
    }
    if (dbg) fprintf(stderr, "prepare_scan: unstuff + grid %.3f ms, periodic runs %.3f ms (%zu sub-sequences, %zu predicted)\n",
                     std::chrono::duration<double, std::milli>(tt1 - tt0).count(),
                     std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - tt1).count(), nsub, nper);
    subs[nsub].start = (uint32_t)o; // sentinel
    subs[nsub].seg = 0;
    h.magic = HUFF_MAGIC;
    h.blob_bytes = (uint32_t)(off_stream + o + 32);
    h.nsub = (uint32_t)nsub; h.nseg = (uint32_t)nseg; h.ri_mcus = (uint32_t)ri;
    h.bpm = (uint32_t)bpm; h.ncomp = (uint32_t)d->ncomp;
    h.mcu_x = (uint32_t)d->mcu_x; h.mcu_y = (uint32_t)d->mcu_y; h.total_mcus = (uint32_t)total;
    h.is_eoi = is_eoi ? 1 : 0;
    h.sub_bytes = (uint32_t)sub;
    h.round_budget = (uint32_t)budget;
    h.off_per = (uint32_t)off_per;
    h.nper = (uint32_t)nper;
    h.rowlen = (uint32_t)eoi_rowlen(d);
    h.tab_entries = (uint32_t)tab_used;
    h.off_tab = (uint32_t)off_tab; h.off_sub = (uint32_t)off_sub; h.off_seg = (uint32_t)off_seg;
    h.off_stream = (uint32_t)off_stream; h.stream_bytes = (uint32_t)(o + 32);
    memcpy(blob, &h, sizeof h);
    d->blob_len = h.blob_bytes;
    d->scan_ready = true;
    return ZJ_OK;
}

int next_marker(BitReader& br)
{
    if (br.marker) { int m = br.marker; br.marker = 0; return m; }
    while (br.p < br.end) {
        if (*br.p++ == 0xFF) {
            while (br.p < br.end && *br.p == 0xFF) br.p++;
            if (br.p < br.end && *br.p != 0) return *br.p++;
        }
    }
    return -1;
}

// for_device: a baseline scan the GPU entropy stage can take is only PREPARED (prepare_scan: scan_ready instead of
// coef_valid); everything else is decoded here as always
int decode_all_once(zj_decoder* d, const uint8_t* buf, size_t len, bool headers_only, bool for_device);

// The image is decoded WITHOUT the reference reader's bit history first (round 4: the history costs the Huffman loop 9-13 %,
// and only a short DC read ever looks at it -- 78 in 1 500 quality-100 noise files, none in ordinary images); the first
// block with a short read returns ZJ_INT_NEED_HIST and the whole image is decoded again with the history on.
int decode_all(zj_decoder* d, const uint8_t* buf, size_t len, bool headers_only, bool for_device = false)
{
    d->track_hist = false;
    int rc = decode_all_once(d, buf, len, headers_only, for_device);
    if (rc == ZJ_INT_NEED_HIST) {
        if (d->stream.active) { (void)zj_frame_abort(d->stream.ctx); d->stream.active = false; } // (the planes are written again)
        d->track_hist = true;
        d->hist_retries++;
        rc = decode_all_once(d, buf, len, headers_only, false);
        d->track_hist = false;
    }
    if (rc == ZJ_INT_NEED_LITERAL) { // the walk consumed bits it made up behind a marker: the scan again, the reference's way
        if (d->stream.active) { (void)zj_frame_abort(d->stream.ctx); d->stream.active = false; }
        d->literal = true;
        d->literal_retries++;
        rc = decode_all_once(d, buf, len, headers_only, false);
        d->literal = false;
    }
    return rc;
}

int decode_all_once(zj_decoder* d, const uint8_t* buf, size_t len, bool headers_only, bool for_device)
{
    d->err.clear(); d->err_code = 0; d->seen_sof = 0; d->scans = 0; d->restart_interval = 0;
    d->coef_valid = false; d->scan_ready = false; d->src = nullptr; d->src_len = 0;
    for (int i = 0; i < 4; i++) { d->qt_present[i] = false; d->dc[i].present = false; d->ac[i].present = false; }
    Cursor c{buf, buf + len};
    int rc = parse_headers(d, c, true);
    if (rc) return rc;
    if (headers_only) return ZJ_OK;
    if (!d->seen_sof) return fail(d, ZJ_ERR_SOF, "Number of components cannot be zero.");
    BitReader br;
    br.p = c.p; br.end = c.end; br.istart = c.p;
    if (!d->progressive) {
        if (for_device && d->entropy && prepare_scan(d, c.p, c.end) == ZJ_OK) {
            d->scans = 1; d->src = buf; d->src_len = len;
            return ZJ_OK;
        }
        if ((rc = ensure_planes(d))) return rc;
        if (d->literal) {
            if (d->ns != d->ncomp) return fail(d, ZJ_ERR_UNSUPPORTED, "baseline scans must carry every component (src/mcu.rs:253-321)");
            rc = scan_baseline_literal(d, br.p, br.end);
            d->scans = 1;
            d->coef_valid = rc == ZJ_OK;
            d->par_scan_mcus = 0; d->dri_parallel_segments = 0;
            return rc;
        }
        stream_begin(d);
        rc = scan_baseline(d, br);
        d->scans = 1;
        d->coef_valid = rc == ZJ_OK;
        return rc;
    }
    for (;;) {
        if (++d->scans > d->max_scans) return fail(d, ZJ_ERR_FORMAT, "Too many scans, exceeded limit of " + std::to_string(d->max_scans));
        br.reset();
        rc = scan_progressive(d, br);
        if (rc) return rc;
        // markers between scans: DHT / SOS / EOI (mcu_prog.rs:90-126)
        for (;;) {
            int m = next_marker(br);
            if (m < 0) return fail(d, ZJ_ERR_FORMAT, "Marker missing where expected");
            if (m == 0xD9) { d->coef_valid = true; return ZJ_OK; }
            Cursor cc{br.p, br.end};
            if (m == 0xC4) { rc = parse_dht(d, cc); br.p = cc.p; if (rc) return rc; continue; }
            if (m == 0xDB) { rc = parse_dqt(d, cc); br.p = cc.p; if (rc) return rc; continue; }
            if (m == 0xDD) { int l, ri; if (!cc.u16(l) || l != 4 || !cc.u16(ri)) return fail(d, ZJ_ERR_FORMAT, "Bad DRI length, Corrupt JPEG"); d->restart_interval = ri; br.p = cc.p; continue; }
            if (m == 0xDA) { rc = parse_sos(d, cc); br.p = cc.p; if (rc) return rc; break; }
            if (m >= 0xD0 && m <= 0xD7) continue;
            d->coef_valid = true;
            return ZJ_OK; // anything else ends the image like the reference's `_ => break 'eoi`
        }
    }
}

} // namespace

extern "C" {

zj_decoder* zj_decoder_new(const zj_options* opt)
{
    zj_decoder* d = new (std::nothrow) zj_decoder();
    if (d && opt) {
        d->out_colorspace = opt->out_colorspace;
        d->strict_mode = opt->strict_mode;
        if (opt->max_width) d->max_width = opt->max_width;
        if (opt->max_height) d->max_height = opt->max_height;
        if (opt->max_scans) d->max_scans = opt->max_scans;
        if (opt->num_threads > 0) d->threads = opt->num_threads;
        d->pinned = opt->pinned_planes != 0;
        d->flags = opt->flags;
        d->out_layout = opt->out_layout;
        d->entropy = opt->entropy;
    }
    if (d) {
        if (const char* e = getenv("ZJ_ENTROPY")) d->entropy = atoi(e); // A/B switch for whole applications
        if (const char* e = getenv("ZJ_PLANE_STORE")) {
            const int v = atoi(e);
            const bool ok = v >= 0 && v <= 4 && (v != 3 || __builtin_cpu_supports("avx")) && (v != 4 || __builtin_cpu_supports("avx512f"));
            if (ok) d->plane_store = v;
        }
        if (const char* e = getenv("ZJ_HUFF_SUB")) { const int v = atoi(e); if (v >= 16 && v <= zj::HUFF_SUB_MAX && v % 16 == 0) d->sub_bytes = v; }
    }
    return d;
}
void zj_decoder_free(zj_decoder* d) { delete d; }
const char* zj_decoder_error(const zj_decoder* d) { return d ? d->err.c_str() : ""; }
int zj_decoder_parallel_segments(const zj_decoder* d) { return d ? d->dri_parallel_segments : 0; }
int64_t zj_decoder_parallel_mcus(const zj_decoder* d) { return d ? (int64_t)d->par_scan_mcus : 0; }
int zj_decoder_set_num_threads(zj_decoder* d, int threads)
{   // Decoder::set_num_threads (src/decoder.rs:591-603): "Cannot set zero threads to decode image"
    if (!d) return ZJ_ERR_ARG;
    if (threads <= 0) return fail(d, ZJ_ERR_FORMAT, "Cannot set zero threads to decode image");
    d->threads = threads;
    return ZJ_OK;
}

static void fill_info(const zj_decoder* d, zj_image_info* info, zj_frame_desc* fd)
{
    if (info) {
        info->width = (uint16_t)d->width; info->height = (uint16_t)d->height;
        info->components = (uint8_t)d->ncomp; info->progressive = (uint8_t)d->progressive;
        info->h_max = (uint8_t)d->h_max; info->v_max = (uint8_t)d->v_max;
        info->scans = (uint16_t)d->scans; info->restart_interval = (uint16_t)d->restart_interval;
    }
    if (fd) {
        memset(fd, 0, sizeof *fd);
        fd->width = (uint32_t)d->width; fd->height = (uint32_t)d->height;
        fd->h_max = (uint32_t)d->h_max; fd->v_max = (uint32_t)d->v_max;
        fd->in_components = (uint32_t)d->ncomp;
        // single-component images are always decoded to GRAYSCALE (headers.rs:283-290)
        fd->out_colorspace = d->ncomp == 1 ? (int)ZJ_CS_GRAYSCALE : d->out_colorspace;
        fd->flags = d->flags & ZJ_FLAG_CORRECTED; // (ZJ_FLAG_FULL_AC_VALUES is the front-end's own)
        fd->out_layout = d->out_layout;
        for (int c = 0; c < 3; c++) {
            const Comp& cm = d->comps[c < d->ncomp ? c : 0];
            for (int k = 0; k < 64; k++) fd->qt[c][k] = cm.q[k]; // the SOF-time snapshot, not d->qt as it stands now
        }
    }
}

} // extern "C"
namespace { void fill_frame_desc(const zj_decoder* d, zj_frame_desc* fd) { fill_info(d, nullptr, fd); } }
extern "C" {

int zj_decoder_read_headers(zj_decoder* d, const uint8_t* buf, size_t len, zj_image_info* info)
{
    if (!d || !buf) return ZJ_ERR_ARG;
    int rc = decode_all(d, buf, len, true);
    if (rc) return rc;
    fill_info(d, info, nullptr);
    return ZJ_OK;
}

int zj_decoder_decode_coefficients(zj_decoder* d, const uint8_t* buf, size_t len, zj_frame_desc* desc,
                                   const int16_t** planes, size_t* plane_len, zj_image_info* info)
{
    if (!d || !buf) return ZJ_ERR_ARG;
    int rc = decode_all(d, buf, len, false);
    if (rc) return rc;
    fill_info(d, info, desc);
    for (int c = 0; c < 3; c++) {
        if (planes) planes[c] = c < d->ncomp ? d->comps[c].coef : nullptr;
        if (plane_len) plane_len[c] = c < d->ncomp ? d->comps[c].coef_len : 0;
    }
    return ZJ_OK;
}

int zj_decoder_prepare(zj_decoder* d, const uint8_t* buf, size_t len, zj_frame_desc* desc, zj_image_info* info)
{
    if (!d || !buf) return ZJ_ERR_ARG;
    int rc = decode_all(d, buf, len, false, true);
    if (rc) return rc;
    fill_info(d, info, desc);
    return ZJ_OK;
}

int zj_decoder_scan_blob(const zj_decoder* d, const void** blob, size_t* len)
{
    if (!d || !d->scan_ready) return ZJ_ERR_ARG;
    if (blob) *blob = d->blob_store.p;
    if (len) *len = d->blob_len;
    return ZJ_OK;
}

static int finish_impl(zj_decoder* d, zj_ctx* ctx, uint8_t* out, size_t out_cap, size_t* out_len, int on_device)
{
    if (!d || !ctx || !out) return ZJ_ERR_ARG;
    // read_headers alone allocates the planes but decodes nothing into them: only a complete decode_all (or a prepared
    // scan) counts
    if (!d->seen_sof || d->err_code || (!d->coef_valid && !d->scan_ready)) return fail(d, ZJ_ERR_ARG, "no successfully decoded coefficients to finish");
    zj_frame_desc fd;
    fill_info(d, nullptr, &fd);
    // grayscale JPEG decoded to RGB: the reference converts nothing and returns zeros (worker.rs:131)
    const size_t need = zj_out_len(&fd);
    if (out_len) *out_len = need;
    if (out_cap < need) return fail(d, ZJ_ERR_ARG, "output buffer too small");
    const bool zeros = fd.in_components == 1 && fd.out_colorspace != ZJ_CS_GRAYSCALE;
    if (d->scan_ready && !zeros) {
        unsigned status = 0;
        const int rc = zj_decode_scan(ctx, &fd, d->blob_store.p, d->blob_len, out, on_device, &status);
        d->gpu_status = status;
        if (rc == ZJ_OK) return ZJ_OK;
        if (rc != ZJ_RETRY_CPU) return fail(d, rc, std::string("GPU entropy stage: ") + zj_strerror(rc) + " " + zj_last_error(ctx));
        // the device met something only the CPU walker treats the way the reference does: decode the file there
        const uint8_t* src = d->src;
        const size_t src_len = d->src_len;
        const int rc2 = decode_all(d, src, src_len, false, false);
        if (rc2) return rc2;
    } else if (d->scan_ready) {
        // (nothing is decoded for the all-zero output; a damaged scan would have been an error on the CPU path)
        const uint8_t* src = d->src;
        const size_t src_len = d->src_len;
        const int rc2 = decode_all(d, src, src_len, false, false);
        if (rc2) return rc2;
    }
    if (zeros) {
        if (on_device) { const int rc = zj_device_memset(ctx, out, 0, need); if (rc) return fail(d, rc, "memset"); }
        else memset(out, 0, need);
        return ZJ_OK;
    }
    int rc;
    if (on_device) rc = zj_decode_planes_to_device(ctx, &fd, d->comps[0].coef, d->ncomp == 3 ? d->comps[1].coef : nullptr,
                                                   d->ncomp == 3 ? d->comps[2].coef : nullptr, out);
    else rc = zj_decode_planes(ctx, &fd, d->comps[0].coef, d->ncomp == 3 ? d->comps[1].coef : nullptr,
                               d->ncomp == 3 ? d->comps[2].coef : nullptr, out);
    if (rc) return fail(d, rc, std::string("pixel path: ") + zj_strerror(rc) + " " + zj_last_error(ctx));
    return ZJ_OK;
}

int zj_decoder_finish_pixels(zj_decoder* d, zj_ctx* ctx, uint8_t* out, size_t out_cap, size_t* out_len)
{
    return finish_impl(d, ctx, out, out_cap, out_len, 0);
}

// Stage 2 of several decoders on one context.  Those whose zj_decoder_prepare left a scan for the device go through
// zj_decode_scans together (one launch per phase for all of them); what the device hands back, and every decoder that
// holds CPU-decoded planes, is finished one by one.
int zj_decoder_finish_pixels_batch(zj_decoder* const* ds, size_t n, zj_ctx* ctx, uint8_t* const* outs, const size_t* out_caps,
                                   size_t* out_lens, int outs_on_device, int* rcs)
{
    if (!ds || !n || !ctx || !outs || !out_caps || !rcs) return ZJ_ERR_ARG;
    std::vector<char> done(n, 0);
    for (size_t base = 0; base < n;) {
        // the next chunk of device scans
        size_t idx[ZJ_SCAN_BATCH_MAX], m = 0, k = base;
        zj_frame_desc fds[ZJ_SCAN_BATCH_MAX];
        const void* blobs[ZJ_SCAN_BATCH_MAX];
        size_t lens[ZJ_SCAN_BATCH_MAX];
        uint8_t* o[ZJ_SCAN_BATCH_MAX];
        for (; k < n && m < (size_t)ZJ_SCAN_BATCH_MAX; k++) {
            zj_decoder* d = ds[k];
            if (!d || !outs[k] || !d->seen_sof || d->err_code || !d->scan_ready) continue;
            fill_info(d, nullptr, &fds[m]);
            if (fds[m].in_components == 1 && fds[m].out_colorspace != ZJ_CS_GRAYSCALE) continue; // all-zero output: finish_impl
            const size_t need = zj_out_len(&fds[m]);
            if (out_caps[k] < need) continue;                                                      // finish_impl reports it
            if (out_lens) out_lens[k] = need;
            blobs[m] = d->blob_store.p; lens[m] = d->blob_len; o[m] = outs[k]; idx[m++] = k;
        }
        base = k;
        if (m < 2) continue; // a single scan: finish_impl below does the same
        int sub_rc[ZJ_SCAN_BATCH_MAX];
        unsigned st[ZJ_SCAN_BATCH_MAX];
        const int rc = zj_decode_scans(ctx, m, fds, blobs, lens, o, outs_on_device, sub_rc, st);
        if (rc) continue; // the call itself failed: one by one below, where the error gets its text
        for (size_t q = 0; q < m; q++) {
            zj_decoder* d = ds[idx[q]];
            d->gpu_status = st[q];
            if (sub_rc[q] == ZJ_OK) { rcs[idx[q]] = ZJ_OK; done[idx[q]] = 1; }
            else if (sub_rc[q] == ZJ_RETRY_CPU) {
                // the CPU walker decodes the file; finish_impl then sees planes, not a scan
                const uint8_t* src = d->src;
                const size_t src_len = d->src_len;
                const int rc2 = decode_all(d, src, src_len, false, false);
                if (rc2) { rcs[idx[q]] = rc2; done[idx[q]] = 1; }
            }
        }
    }
    for (size_t k = 0; k < n; k++)
        if (!done[k]) rcs[k] = ds[k] ? finish_impl(ds[k], ctx, outs[k], out_caps[k], out_lens ? &out_lens[k] : nullptr, outs_on_device) : (int)ZJ_ERR_ARG;
    return ZJ_OK;
}

int zj_decoder_finish_pixels_device(zj_decoder* d, zj_ctx* ctx, uint8_t* d_out, size_t out_cap, size_t* out_len)
{
    return finish_impl(d, ctx, d_out, out_cap, out_len, 1);
}

unsigned zj_decoder_gpu_status(const zj_decoder* d) { return d ? d->gpu_status : 0; }

int zj_decoder_decode_buffer(zj_decoder* d, zj_ctx* ctx, const uint8_t* buf, size_t len, uint8_t* out,
                             size_t out_cap, size_t* out_len, zj_image_info* info)
{
    if (!d || !ctx || !buf || !out) return ZJ_ERR_ARG;
    // a baseline scan walked on the CPU into pinned planes: its strips go to the GPU while the walker is in later rows
    // (stream_begin decides; ZJ_STREAM=off keeps the two stages apart)
    const char* e = getenv("ZJ_STREAM");
    if (!(e && (!strcmp(e, "off") || !strcmp(e, "0")))) { d->stream.ctx = ctx; d->stream.out = out; d->stream.cap = out_cap; }
    int rc = zj_decoder_prepare(d, buf, len, nullptr, info);
    d->stream.ctx = nullptr; d->stream.out = nullptr; // (armed for this call only)
    const bool streamed = d->stream.active;
    d->stream.active = false;
    if (rc) { if (streamed) (void)zj_frame_abort(ctx); return rc; }
    if (streamed) {
        rc = zj_frame_end(ctx);
        if (rc) return fail(d, rc, std::string("pixel path: ") + zj_strerror(rc) + " " + zj_last_error(ctx));
        if (out_len) { zj_frame_desc fd; fill_info(d, nullptr, &fd); *out_len = zj_out_len(&fd); }
        return ZJ_OK;
    }
    return zj_decoder_finish_pixels(d, ctx, out, out_cap, out_len);
}

} // extern "C"
