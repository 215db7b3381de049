// zj_huff_device.h -- device code of the GPU entropy stage (see zj_huff.h for the algorithm and what it replaces).
// Per-thread functions only; the kernels around them are in zj_huff.hip.  Like zj_device.h this header is also compiled
// by g++ into the CPU emulation harness (tests/emu, ZJ_EMU), which runs the threads of a workgroup one after another:
// test infrastructure, never part of libzjhip.so.
//
// The parse restates src/bitstream.rs:314-373 (decode_mcu_block: DC difference, then run/size symbols until EOB or 64
// coefficients; EXTEND per T.81 F.2.2.1) and src/huffman.rs:73-276 (canonical codes) in the form of the product's CPU
// walker (zj_jpeg.cpp decode_block_baseline), whose planes it must reproduce bit for bit.
#pragma once

#include "zj_huff.h"

#if defined(ZJ_EMU)
#ifndef ZJ_DEV
#define ZJ_DEV inline
#endif
#else
#include <hip/hip_runtime.h>
#ifndef ZJ_DEV
#define ZJ_DEV __device__ __forceinline__
#endif
#endif

namespace zj {

// ---- LDS image of a workgroup --------------------------------------------------------------------------------------
constexpr int HUFF_LANE_DATA = HUFF_SUB_MAX / 4 + 4;   // stream words a thread stages: its sub-sequence + 16 bytes (a symbol
                                                       // that begins before the limit may end up to 31 bits after it)
constexpr int HUFF_LANE_WORDS = HUFF_LANE_DATA + 1;    // odd stride: lanes reading "their word k" hit different banks
struct HuffLds {
    uint32_t W[HUFF_WG * HUFF_LANE_WORDS];    // big-endian stream words, a private stretch per thread (a sparse round's
                                              // threads decode sub-sequences from all over the scan)
    alignas(16) uint16_t T[HUFF_TAB_BUDGET];  // decoding tables
    HuffScan hdr;
    uint8_t unz[64];                          // zig-zag index -> natural position
};
struct alignas(16) HuffU4 { uint32_t x, y, z, w; };

#if defined(ZJ_EMU)
ZJ_DEV void huff_or(uint32_t* p, uint32_t v) { *p |= v; }
ZJ_DEV void huff_max(uint32_t* p, uint32_t v) { if (v > *p) *p = v; }
ZJ_DEV void huff_add(uint32_t* p, uint32_t v) { *p += v; }
ZJ_DEV uint32_t huff_add_return(uint32_t* p, uint32_t v) { const uint32_t o = *p; *p += v; return o; }
ZJ_DEV uint32_t huff_bswap(uint32_t v) { return __builtin_bswap32(v); }
#else
ZJ_DEV void huff_or(uint32_t* p, uint32_t v) { atomicOr(p, v); }
ZJ_DEV void huff_max(uint32_t* p, uint32_t v) { atomicMax(p, v); }
ZJ_DEV void huff_add(uint32_t* p, uint32_t v) { atomicAdd(p, v); }
ZJ_DEV uint32_t huff_add_return(uint32_t* p, uint32_t v) { return atomicAdd(p, v); }
ZJ_DEV uint32_t huff_bswap(uint32_t v) { return __builtin_bswap32(v); }
#endif

ZJ_DEV const HuffScan* huff_hdr(const uint8_t* blob) { return (const HuffScan*)blob; }
ZJ_DEV const HuffSub* huff_subs(const uint8_t* blob) { return (const HuffSub*)(blob + huff_hdr(blob)->off_sub); }
ZJ_DEV const HuffSeg* huff_segs(const uint8_t* blob) { return (const HuffSeg*)(blob + huff_hdr(blob)->off_seg); }

// T.81 figure A.6, zig-zag index -> natural order (src/misc.rs:24-33 UN_ZIGZAG)
ZJ_DEV uint32_t huff_unzigzag(int k)
{
    // packed 4 x 6 bits... kept as a byte table: the compiler places it in constant memory
    const uint8_t t[64] = {0,  1,  8,  16, 9,  2,  3,  10, 17, 24, 32, 25, 18, 11, 4,  5,  12, 19, 26, 33, 40, 48,
                           41, 34, 27, 20, 13, 6,  7,  14, 21, 28, 35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23,
                           30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63};
    return t[k];
}

// Staging.  Cooperative (NT threads): header, decoding tables, zig-zag table.  Per thread: the stream words of the
// sub-sequence `sub_index` it is about to decode, if any (16-byte loads, all issued before the first LDS write).
// A barrier must follow.
template <int NT>
ZJ_DEV void huff_stage(const uint8_t* blob, int tid, bool mine, uint32_t sub_index, HuffLds& L)
{
    const HuffScan* g = huff_hdr(blob);
    if (mine) {
        const uint32_t b0 = huff_subs(blob)[sub_index].start;          // 16-byte aligned
        const uint32_t pieces_left = (g->stream_bytes - b0) / 16;       // the stream ends with >= 32 zero bytes
        const HuffU4* src = (const HuffU4*)(blob + g->off_stream + b0);
        constexpr int PIECES = HUFF_LANE_DATA / 4;
        HuffU4 v[PIECES];
#pragma unroll
        for (int q = 0; q < PIECES; q++) {
            v[q].x = v[q].y = v[q].z = v[q].w = 0;
            if ((uint32_t)q < pieces_left) v[q] = src[q];
        }
        uint32_t* dst = L.W + (uint32_t)tid * HUFF_LANE_WORDS;
#pragma unroll
        for (int q = 0; q < PIECES; q++) {
            dst[4 * q] = huff_bswap(v[q].x); dst[4 * q + 1] = huff_bswap(v[q].y);
            dst[4 * q + 2] = huff_bswap(v[q].z); dst[4 * q + 3] = huff_bswap(v[q].w);
        }
    }
    const HuffU4* tsrc = (const HuffU4*)(blob + g->off_tab);
    HuffU4* tdst = (HuffU4*)L.T;
    const uint32_t tq = (g->tab_entries + 7) / 8;
    for (uint32_t k = (uint32_t)tid; k < tq; k += NT) tdst[k] = tsrc[k];
    const uint32_t* hsrc = (const uint32_t*)blob;
    uint32_t* hdst = (uint32_t*)&L.hdr;
    for (uint32_t k = (uint32_t)tid; k < sizeof(HuffScan) / 4; k += NT) hdst[k] = hsrc[k];
    if (tid < 64) L.unz[tid] = (uint8_t)huff_unzigzag(tid);
}

// ---- the parse -------------------------------------------------------------------------------------------------------
struct HuffState { uint32_t pos, j, z; }; // bit position in the stream; block of the MCU; zig-zag index of the next coefficient (0: DC)
ZJ_DEV unsigned long long huff_pack(HuffState s) { return ((unsigned long long)s.pos << 16) | (unsigned long long)(s.j << 8) | s.z; }
ZJ_DEV HuffState huff_unpack(unsigned long long v) { HuffState s; s.pos = (uint32_t)(v >> 16); s.j = (uint32_t)(v >> 8) & 255; s.z = (uint32_t)v & 255; return s; }

struct HuffWrite { // what only the write pass carries
    int16_t* plane[3];
    uint32_t blk, blk_end;  // block in progress (scan order, absolute), end of the restart segment
    int32_t pred[3];        // DC predictors
    uint32_t mcu, mx, my;   // MCU in progress
    uint32_t seg_start_bits, eoi_d; // for the reference's early exit at EOI (zj_jpeg.cpp EoiCut): start of the segment,
    int eoi_seg;                    // data bytes in front of the marker; whether this is the segment that ends with EOI
    uint32_t seg_end_bits;          // exact end of the segment's data; 0 for the scan's last segment (nothing follows it)
    uint32_t* ctl;
};

ZJ_DEV int16_t* huff_block_ptr(const HuffScan& h, const HuffWrite& w, const HuffBlk b)
{
    const HuffComp c = h.comp[b.comp];
    const size_t bx = (size_t)w.mx * c.h + b.hx, by = (size_t)w.my * c.v + b.vy;
    int16_t* base = b.comp == 0 ? w.plane[0] : b.comp == 1 ? w.plane[1] : w.plane[2]; // (selects, not an indexed array)
    return base + (by * c.bw + bx) * 64;
}

// the 32 bits that follow `off` consumed bits of hi (1 <= off <= 32; off == 32: lo itself): one v_alignbit_b32
ZJ_DEV uint32_t huff_window(uint32_t hi, uint32_t lo, uint32_t off)
{
#if defined(ZJ_EMU)
    return (uint32_t)(((((unsigned long long)hi << 32) | lo) >> (32u - off)) & 0xffffffffu);
#else
    return __builtin_amdgcn_alignbit(hi, lo, 32u - off);
#endif
}

// Decodes symbols from state `s` while they BEGIN before bit `limit`.  Sync rounds (WRITE false) only track the state,
// the blocks completed and the DC differences; the write pass stores coefficients, stops at the end of its segment's
// blocks and raises status bits for anything a well-formed scan cannot contain.
// Garbage in (a wrong guess) must be harmless: every read is bounded, every symbol advances by at least one bit.
//
// One table entry (zj_huff.h) carries all a symbol does to the state -- bits consumed (code + magnitude) and the
// advance of the zig-zag index (DC: 1, run + 1, ZRL: 16, EOB: 63) -- so a sync round needs the magnitude bits of DC
// symbols only.  The bit window is 32 bits rebuilt per symbol from two cached stream words (a code is at most 16
// bits, its magnitude at most 15); a third word is always in flight from LDS.
// `start_bits`: first bit of the sub-sequence whose words thread `tid` has staged.
// Would the reference read a DC symbol of `total` (> 16) bits short (src/bitstream.rs:278, zj_jpeg.cpp ref_dc_misread)?
// c_prev / c_sym: bits consumed since the segment began in front of the previous symbol (the last AC symbol of the block
// before) and in front of this one.  Behind the refill of the AC call site at c_prev the reference's reader holds
// 64 - (c_prev mod 32) bits -- 32 or 64 when c_prev mod 32 == 0, depending on its history: assume 32 -- then the previous
// symbol goes, and decode_dc adds 32 if fewer than 16 are left.  Errs on the side of "yes"; the CPU walker, which follows
// the reader bit by bit, decides.
ZJ_DEV bool huff_ref_reads_short(const uint32_t c_prev, const uint32_t c_sym, const uint32_t total)
{
    const uint32_t r = c_prev & 31u;
    int32_t rbl = (int32_t)(r ? 64u - r : 32u) - (int32_t)(c_sym - c_prev);
    if (rbl < 16) rbl += 32;
    return (int32_t)total > rbl;
}

template <bool WRITE>
ZJ_DEV HuffState huff_run(const HuffLds& L, uint32_t tid, uint32_t start_bits, HuffState s, uint32_t limit, bool last_sub, HuffI4& aux, HuffWrite* w)
{
    const HuffScan& h = L.hdr;
    const uint32_t* Wl = L.W + tid * HUFF_LANE_WORDS;
    constexpr uint32_t nwords = HUFF_LANE_DATA;
    const uint32_t rel = s.pos - start_bits;
    uint32_t k = rel >> 5, off = rel & 31;
    if (off == 0) { off = 32; k -= 1; } // (k may wrap for the very first bit: the word is never looked at)
    uint32_t hi = k < nwords ? Wl[k] : 0u;
    uint32_t lo = k + 1 < nwords ? Wl[k + 1] : 0u;
    uint32_t nx = k + 2 < nwords ? Wl[k + 2] : 0u;
    k += 3;
    uint32_t pos = s.pos, j = s.j, z = s.z;
    int32_t n = 0, d0 = 0, d1 = 0, d2 = 0; // blocks completed, DC differences per component
    int32_t p0 = 0, p1 = 0, p2 = 0;        // DC predictors (write pass)
    if (WRITE) { p0 = w->pred[0]; p1 = w->pred[1]; p2 = w->pred[2]; }
    uint32_t status = 0;
    if (j >= h.bpm) j = 0; // (only a corrupted exit word could say so)
    // per component: table offsets; per block of the MCU: its component (2 bits each) -- registers, not LDS, because
    // every iteration of a wave has some lane at a block boundary.  m0..m2: all-ones for the current block's component
    // (the DC difference is added to that component's sum, masked adds instead of selects)
    const uint32_t cmask = h.comp_of_blk;
    const uint32_t dc0 = h.dc_off[0], dc1 = h.dc_off[1], dc2 = h.dc_off[2], ac0 = h.ac_off[0], ac1 = h.ac_off[1], ac2 = h.ac_off[2];
    uint32_t comp = (cmask >> (2 * j)) & 3u;
    uint32_t dcb = comp == 0 ? dc0 : comp == 1 ? dc1 : dc2, acb = comp == 0 ? ac0 : comp == 1 ? ac1 : ac2;
    int32_t m0 = comp == 0 ? -1 : 0, m1 = comp == 1 ? -1 : 0, m2 = comp == 2 ? -1 : 0;
    int16_t* dst = nullptr;
    if (WRITE) dst = huff_block_ptr(h, *w, h.blk[j]);
    uint32_t prev_start = 0xffffffffu; // (write pass) first bit of the previous symbol; none yet in this sub-sequence
    for (;;) {
        if (WRITE && w->blk >= w->blk_end) {
            // the interval's blocks are done: what is left in front of its marker must be padding, and the reference's reader
            // must have come across the marker -- 4 * (C / 32 + 2) > D, C = bits consumed in front of the interval's last
            // symbol (zj_jpeg.cpp handle_restart) -- or it resets nothing there (HUFF_ST_LEFT_OVER)
            if (w->seg_end_bits && (w->seg_end_bits >= pos + 8u ||
                                    (prev_start != 0xffffffffu && 4u * ((prev_start - w->seg_start_bits) / 32u + 2u) <= w->eoi_d))) status |= HUFF_ST_LEFT_OVER;
            break;
        }
        if (pos >= limit) {
            if (WRITE && last_sub) status |= HUFF_ST_EXHAUSTED; // blocks are missing and the segment has no more bits
            break;
        }
        const uint32_t win = huff_window(hi, lo, off);
        const bool is_dc = z == 0;
        const uint32_t tb = is_dc ? dcb : acb;
        const uint32_t l1 = is_dc ? (uint32_t)HUFF_L1_DC : (uint32_t)HUFF_L1_AC;
        uint32_t e = L.T[tb + (win >> (32u - l1))];
        if (e & 0x8000u) e = L.T[tb + (1u << l1) + ((e & 0xffu) << (16u - l1)) + ((win >> 16) & ((1u << (16u - l1)) - 1u))];
        // (no such code: the entry says 16 bits, zig-zag advance 0 -- a fixed step for a guess gone wrong, a status for
        // the write pass)
        const uint32_t total = e & 31u, zadv = (e >> 5) & 63u;
        if (WRITE && zadv == 0) status |= HUFF_ST_BAD_CODE;
        const uint32_t sym_start = pos;
        if (is_dc || WRITE) {
            // magnitude bits and EXTEND (T.81 F.2.2.1) without a branch on the size: t holds the bits left-aligned
            const uint32_t sz = (e >> 11) & 15u;
            const uint32_t t = win << (total - sz);
            const uint32_t bits = (t >> 1) >> (31u - sz);                                   // sz == 0: 0
            const int32_t val = (int32_t)bits + ((int32_t)~((int32_t)t >> 31) & (1 - (int32_t)(1u << sz))); // top bit 0: negative
            if (is_dc) {
                // (the first symbol of a sub-sequence knows no predecessor: the lane in front of it looks ahead, below)
                if (WRITE && total > 16u && prev_start != 0xffffffffu &&
                    huff_ref_reads_short(prev_start - w->seg_start_bits, sym_start - w->seg_start_bits, total)) status |= HUFF_ST_DC_LONG;
                const int32_t dv = sz ? val : 0; // (sz == 0: t's top bit belongs to the next symbol)
                d0 += dv & m0; d1 += dv & m1; d2 += dv & m2;
                if (WRITE) {
                    p0 = (int32_t)((uint32_t)p0 + (uint32_t)(dv & m0));
                    p1 = (int32_t)((uint32_t)p1 + (uint32_t)(dv & m1));
                    p2 = (int32_t)((uint32_t)p2 + (uint32_t)(dv & m2));
                    dst[0] = (int16_t)((p0 & m0) | (p1 & m1) | (p2 & m2)); // bitstream.rs:330
                }
            } else if (WRITE && sz) {
                const uint32_t zz = z + zadv - 1;
                if (zz <= 63) dst[L.unz[zz]] = (int16_t)val;
                else status |= HUFF_ST_RUN_OVER;
            }
        }
        if (WRITE) prev_start = sym_start;
        z += zadv;
        pos += total;
        off += total;
        if (off > 32) {
            off -= 32;
            hi = lo;
            lo = nx;
            nx = k < nwords ? Wl[k] : 0u;
            k++;
        }
        if (z >= 64) { // the block is complete
            z = 0;
            n++;
            j++;
            if (WRITE) w->blk++;
            if (j == h.bpm) {
                j = 0;
                if (WRITE) {
                    if (w->eoi_seg) {
                        // the reference has come across EOI iff 4 * (C / 32 + 2) > D, C = bits consumed in front of the
                        // MCU's last symbol (zj_jpeg.cpp eoi_cut_after_mcu); C grows with the MCU index, so the first MCU
                        // that satisfies it is the minimum
                        const uint32_t c_last = sym_start - w->seg_start_bits;
                        if (4u * (c_last / 32u + 2u) > w->eoi_d) huff_max(&w->ctl[HUFF_CTL_SEEN], ~w->mcu); // (the complement: all control words start at 0)
                    }
                    w->mcu++;
                    if (++w->mx == h.mcu_x) { w->mx = 0; w->my++; }
                }
            }
            comp = (cmask >> (2 * j)) & 3u;
            dcb = comp == 0 ? dc0 : comp == 1 ? dc1 : dc2;
            acb = comp == 0 ? ac0 : comp == 1 ? ac1 : ac2;
            m0 = comp == 0 ? -1 : 0; m1 = comp == 1 ? -1 : 0; m2 = comp == 2 ? -1 : 0;
            if (WRITE && w->blk < w->blk_end) dst = huff_block_ptr(h, *w, h.blk[j]);
        }
    }
    // A segment that was cut short: its last symbol began inside the segment but ran past its exact end, into the padding
    // or the next segment's bytes.  The CPU walker's reader feeds zero bits behind a marker (bitstream.rs:150-262), so the
    // two would disagree about that symbol: hand the scan back (a well-formed segment ends at or before its limit).
    if (WRITE && last_sub && pos > limit) status |= HUFF_ST_EXHAUSTED;
    // the next sub-sequence begins with a DC symbol whose predecessor is this lane's last symbol: the short-read test for
    // it happens here (a look at its table entry; the stream words behind the limit are staged, HUFF_LANE_DATA)
    if (WRITE && !last_sub && z == 0 && pos >= limit && prev_start != 0xffffffffu && w->blk < w->blk_end) {
        const uint32_t win = huff_window(hi, lo, off);
        uint32_t e = L.T[dcb + (win >> (32u - (uint32_t)HUFF_L1_DC))];
        if (e & 0x8000u) e = L.T[dcb + (1u << HUFF_L1_DC) + ((e & 0xffu) << (16u - HUFF_L1_DC)) + ((win >> 16) & ((1u << (16u - HUFF_L1_DC)) - 1u))];
        const uint32_t total = e & 31u;
        if (total > 16u && huff_ref_reads_short(prev_start - w->seg_start_bits, pos - w->seg_start_bits, total)) status |= HUFF_ST_DC_LONG;
    }
    aux.x = n; aux.y = d0; aux.z = d1; aux.w = d2;
    if (WRITE && status) huff_or(&w->ctl[HUFF_CTL_STATUS], status);
    HuffState o;
    o.pos = pos; o.j = j; o.z = z;
    return o;
}

// bit limit of sub-sequence i: where the next one of the same segment begins, or the exact end of the segment
ZJ_DEV uint32_t huff_limit(const uint8_t* blob, const HuffSub* subs, uint32_t i, const HuffSub sub)
{
    if (sub.seg & HUFF_LAST) return huff_segs(blob)[sub.seg & HUFF_SEG_MASK].end * 8u;
    return subs[i + 1].start * 8u;
}

// lanes per work-list entry (see HuffArgs::spread); entries > 0
ZJ_DEV uint32_t huff_spread(const HuffArgs& a, uint32_t nsub, uint32_t entries)
{
    if (entries > HUFF_LIST_FACTOR * nsub) entries = HUFF_LIST_FACTOR * nsub; // (the counter ran past the list: status raised)
    uint32_t S = nsub / entries;
    if (S > (uint32_t)a.spread) S = (uint32_t)a.spread;
    return S ? S : 1u;
}

// which sub-sequence does thread t of a synchronisation round decode?  Rounds 0 and 1: its own (round 1: unless it begins
// a restart segment -- its entry state is known, round 0 was final).  From round 2 on: entry t of the round's work list.
// nsub: no work.
ZJ_DEV uint32_t huff_sync_pick(const HuffArgs& a, uint32_t t, uint32_t nsub, const HuffSub* subs)
{
    if (a.round <= 1) return (t < nsub && (a.round == 0 || !(subs[t].seg & HUFF_FIRST))) ? t : nsub;
    uint32_t entries = a.ctl[HUFF_CTL_ROUND0 + a.round - 1];
    if (entries > HUFF_LIST_FACTOR * nsub) entries = HUFF_LIST_FACTOR * nsub;
    const uint32_t S = huff_spread(a, nsub, entries), e = t / S;
    return (e * S == t && e < entries) ? a.list[(size_t)(a.round & 1) * HUFF_LIST_FACTOR * nsub + e] : nsub;
}

// round 0, thread t of nthreads: its share of the planes' clearing, coalesced (piece p by thread p mod nthreads)
ZJ_DEV void huff_clear_planes(const HuffArgs& a, uint32_t t, uint32_t nthreads)
{
    HuffU4 zero;
    zero.x = zero.y = zero.z = zero.w = 0;
    for (uint32_t p = t; p < a.zero_pieces; p += nthreads) ((HuffU4*)a.zero_base)[p] = zero;
}

// one thread of a synchronisation round, decoding sub-sequence i (after staging)
ZJ_DEV void huff_sync_thread(const HuffArgs& a, const HuffLds& L, uint32_t tid, uint32_t i)
{
    const uint32_t nsub = L.hdr.nsub;
    if (i >= nsub) return;
    const HuffSub* subs = huff_subs(a.blob);
    const HuffSub sub = subs[i];
    HuffState s;
    if (a.round == 0) { s.pos = sub.start * 8u; s.j = 0; s.z = 0; }
    else s = huff_unpack(a.exit_rd[i - 1]);
    HuffI4 aux;
    const HuffState o = huff_run<false>(L, tid, sub.start * 8u, s, huff_limit(a.blob, subs, i, sub), false, aux, nullptr);
    const unsigned long long packed = huff_pack(o);
    const bool differs = a.round == 0 || packed != a.exit[i];
    a.exit[i] = packed;
    a.aux[i] = aux;
    // a changed exit state puts the successor on the next round's list (round 0 changes everything: round 1 needs no list)
    if (differs && a.round && i + 1 < nsub && !(subs[i + 1].seg & HUFF_FIRST)) {
        // (the periodic pass may put a sub-sequence on a list a second time: room for that, and a status if it runs out)
        const uint32_t at = huff_add_return(&a.ctl[HUFF_CTL_ROUND0 + a.round], 1u);
        if (at < HUFF_LIST_FACTOR * nsub) a.list[(size_t)((a.round + 1) & 1) * HUFF_LIST_FACTOR * nsub + at] = i + 1;
        else huff_or(&a.ctl[HUFF_CTL_STATUS], HUFF_ST_NO_SYNC);
    }
}

// one thread of the write pass (after staging)
ZJ_DEV void huff_write_thread(const HuffArgs& a, const HuffLds& L, uint32_t tid, uint32_t i)
{
    const HuffScan& h = L.hdr;
    if (i >= h.nsub) return;
    const HuffSub* subs = huff_subs(a.blob);
    const HuffSub sub = subs[i];
    const uint32_t k = sub.seg & HUFF_SEG_MASK;
    const HuffSeg seg = huff_segs(a.blob)[k];
    HuffState s;
    if (sub.seg & HUFF_FIRST) { s.pos = sub.start * 8u; s.j = 0; s.z = 0; }
    else s = huff_unpack(a.exit[i - 1]);
    HuffI4 b = a.base[i];
    if (a.rel[i]) { // relative to its prefix-sum workgroup
        const HuffAgg p = a.wgpre[i / HUFF_SCAN_WG];
        b.x = (int32_t)((uint32_t)b.x + (uint32_t)p.v[0]); b.y = (int32_t)((uint32_t)b.y + (uint32_t)p.v[1]);
        b.z = (int32_t)((uint32_t)b.z + (uint32_t)p.v[2]); b.w = (int32_t)((uint32_t)b.w + (uint32_t)p.v[3]);
    }
    HuffWrite w;
    w.plane[0] = a.plane[0]; w.plane[1] = a.plane[1]; w.plane[2] = a.plane[2];
    const uint32_t total_blocks = h.total_mcus * h.bpm;
    const unsigned long long end64 = (unsigned long long)(k + 1) * h.ri_mcus * h.bpm;
    w.blk = (uint32_t)b.x;
    w.blk_end = end64 < total_blocks ? (uint32_t)end64 : total_blocks;
    if (w.blk >= w.blk_end) return;
    w.pred[0] = b.y; w.pred[1] = b.z; w.pred[2] = b.w;
    w.mcu = w.blk / h.bpm;
    if (w.blk - w.mcu * h.bpm != s.j) { huff_or(&a.ctl[HUFF_CTL_STATUS], HUFF_ST_PHASE); return; }
    w.my = w.mcu / h.mcu_x;
    w.mx = w.mcu - w.my * h.mcu_x;
    w.seg_start_bits = seg.start * 8u;
    w.eoi_d = seg.end - seg.start;
    w.eoi_seg = h.is_eoi && k + 1 == h.nseg;
    w.seg_end_bits = k + 1 == h.nseg ? 0u : seg.end * 8u;
    w.ctl = a.ctl;
    HuffI4 aux;
    (void)huff_run<true>(L, tid, sub.start * 8u, s, huff_limit(a.blob, subs, i, sub), (sub.seg & HUFF_LAST) != 0, aux, &w);
}

// ---- periodic runs (zj_huff.h) -----------------------------------------------------------------------------------------
// One thread per sub-sequence, between two rounds; `next_round` is the round that will verify what this pass changes.
ZJ_DEV void huff_periodic_thread(const HuffArgs& a, uint32_t i, int next_round)
{
    const HuffScan* g = huff_hdr(a.blob);
    const uint32_t nsub = g->nsub;
    if (i >= nsub) return;
    const uint32_t* per = (const uint32_t*)(a.blob + g->off_per);
    const uint32_t w = per[i];
    if (!w) return;
    const uint32_t q = w >> HUFF_PER_QSHIFT, r0 = w & HUFF_PER_MASK, a0 = r0 + q;
    const uint32_t shift = q * g->sub_bytes * 8u; // bits from one period to the next
    // does the second period close on itself?
    const HuffState e1 = huff_unpack(a.exit[a0 - 1]), e2 = huff_unpack(a.exit[a0 + q - 1]);
    if (e2.pos - e1.pos != shift || e2.j != e1.j || e2.z != e1.z) return;
    const uint32_t t = a0 + (i - a0) % q; // the counterpart in the second period
    HuffState p = huff_unpack(a.exit[t]);
    p.pos += (i - t) * g->sub_bytes * 8u;
    const unsigned long long packed = huff_pack(p);
    // (the counts travel with the exit state even when the state is already the predicted one: they may stem from a decode
    // that entered sub-sequence i in another state, and nothing re-decodes i unless its predecessor's exit changes -- under
    // the closure condition above the counterpart's counts are i's)
    if (packed == a.exit[i]) { if (t != i) a.aux[i] = a.aux[t]; return; }
    a.exit[i] = packed;
    a.aux[i] = a.aux[t];
    // verified by decoding in the next round; so is the sub-sequence behind the run's last predicted one
    uint32_t* cnt = &a.ctl[HUFF_CTL_ROUND0 + next_round - 1];
    uint32_t* list = a.list + (size_t)(next_round & 1) * HUFF_LIST_FACTOR * nsub;
    const uint32_t cap = HUFF_LIST_FACTOR * nsub;
    uint32_t at = huff_add_return(cnt, 1u);
    if (at < cap) list[at] = i; else huff_or(&a.ctl[HUFF_CTL_STATUS], HUFF_ST_NO_SYNC);
    if (i + 1 < nsub && per[i + 1] != w && !(huff_subs(a.blob)[i + 1].seg & HUFF_FIRST)) {
        at = huff_add_return(cnt, 1u);
        if (at < cap) list[at] = i + 1; else huff_or(&a.ctl[HUFF_CTL_STATUS], HUFF_ST_NO_SYNC);
    }
}

// ---- prefix sums: first block and DC predictors of every sub-sequence ----------------------------------------------
// A segmented exclusive scan of aux[]: a sub-sequence that begins a restart segment restarts the sums at (first block
// of the segment, 0, 0, 0).  Two levels: workgroups of HUFF_SCAN_WG sub-sequences scan themselves (one element per
// thread) and publish their totals; the workgroup that finishes last scans the totals; the write pass adds the
// workgroup's prefix to the values that are still relative (rel[i]).
ZJ_DEV HuffAgg huff_scan_identity() { HuffAgg r; r.v[0] = r.v[1] = r.v[2] = r.v[3] = 0; r.reset = 0; return r; }
// left then right
ZJ_DEV HuffAgg huff_scan_op(const HuffAgg l, const HuffAgg r)
{
    if (r.reset) return r;
    HuffAgg o;
    for (int q = 0; q < 4; q++) o.v[q] = (int32_t)((uint32_t)l.v[q] + (uint32_t)r.v[q]);
    o.reset = l.reset;
    return o;
}
ZJ_DEV int32_t huff_seg_first_block(const HuffScan* g, const HuffSub sub)
{
    const unsigned long long b = (unsigned long long)(sub.seg & HUFF_SEG_MASK) * g->ri_mcus * g->bpm;
    const unsigned long long cap = (unsigned long long)g->total_mcus * g->bpm;
    return (int32_t)(b < cap ? b : cap);
}
// what sub-sequence i contributes: its counts, or -- at the head of a segment -- the absolute value after it
ZJ_DEV HuffAgg huff_scan_element(const HuffArgs& a, uint32_t i)
{
    const HuffScan* g = huff_hdr(a.blob);
    if (i >= g->nsub) return huff_scan_identity();
    const HuffSub sub = huff_subs(a.blob)[i];
    const HuffI4 x = a.aux[i];
    HuffAgg r;
    r.v[0] = x.x; r.v[1] = x.y; r.v[2] = x.z; r.v[3] = x.w;
    r.reset = 0;
    if (sub.seg & HUFF_FIRST) { r.v[0] = (int32_t)((uint32_t)r.v[0] + (uint32_t)huff_seg_first_block(g, sub)); r.reset = 1; }
    return r;
}
// excl: the scan of the elements in front of i within its workgroup
ZJ_DEV void huff_scan_store(const HuffArgs& a, uint32_t i, const HuffAgg excl)
{
    const HuffScan* g = huff_hdr(a.blob);
    if (i >= g->nsub) return;
    const HuffSub sub = huff_subs(a.blob)[i];
    HuffI4 b;
    if (sub.seg & HUFF_FIRST) { b.x = huff_seg_first_block(g, sub); b.y = b.z = b.w = 0; a.rel[i] = 0; }
    else { b.x = excl.v[0]; b.y = excl.v[1]; b.z = excl.v[2]; b.w = excl.v[3]; a.rel[i] = excl.reset ? 0 : 1; }
    a.base[i] = b;
}
// (one thread) exclusive scan of the workgroup totals
ZJ_DEV void huff_scan_totals(const HuffArgs& a, uint32_t nwg)
{
    HuffAgg run = huff_scan_identity();
    for (uint32_t w = 0; w < nwg; w++) { a.wgpre[w] = run; run = huff_scan_op(run, a.wgagg[w]); }
}

// ---- the reference's early exit at EOI ---------------------------------------------------------------------------------
// ~ctl[HUFF_CTL_SEEN] = first MCU after which the reference leaves its row loop (or >= total_mcus: none).  The MCUs that follow it
// in the same row loop keep the zeros of the reference's fresh buffers.  The cut has to fall into the LAST row loop
// (anything earlier shifts later MCUs, zj_jpeg.cpp scan_baseline: left to the CPU walker).  Returns the number of
// 16-byte pieces to clear and fills first/count; `piece` p of them is cleared by huff_cut_clear.
ZJ_DEV uint32_t huff_cut_plan(const HuffArgs& a, uint32_t* first_mcu)
{
    const HuffScan* g = huff_hdr(a.blob);
    const uint32_t fs = ~a.ctl[HUFF_CTL_SEEN];
    if (fs >= g->total_mcus || fs + 1 >= g->total_mcus) return 0;
    if (fs / g->rowlen != (g->total_mcus - 1) / g->rowlen) { huff_or(&a.ctl[HUFF_CTL_STATUS], HUFF_ST_CUT_EARLY); return 0; }
    *first_mcu = fs + 1;
    return (g->total_mcus - fs - 1) * g->bpm * 8u;
}
ZJ_DEV void huff_cut_clear(const HuffArgs& a, uint32_t first_mcu, uint32_t piece)
{
    const HuffScan* g = huff_hdr(a.blob);
    const uint32_t per_mcu = g->bpm * 8u;
    const uint32_t m = first_mcu + piece / per_mcu, rest = piece % per_mcu;
    const HuffBlk b = g->blk[rest / 8u];
    const HuffComp c = g->comp[b.comp];
    const uint32_t my = m / g->mcu_x, mx = m - my * g->mcu_x;
    const size_t bx = (size_t)mx * c.h + b.hx, by = (size_t)my * c.v + b.vy;
    int16_t* base = b.comp == 0 ? a.plane[0] : b.comp == 1 ? a.plane[1] : a.plane[2];
    HuffI4* p = (HuffI4*)(base + (by * c.bw + bx) * 64) + (rest & 7u);
    HuffI4 zero;
    zero.x = zero.y = zero.z = zero.w = 0;
    *p = zero;
}

} // namespace zj
