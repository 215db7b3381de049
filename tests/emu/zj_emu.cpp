// zj_emu.cpp -- CPU EMULATION of the HIP workgroup phases in zune-jpeg_amd/csrc/zj_device.h.
//
// TEST INFRASTRUCTURE ONLY: it lets the CPU test-suite (-m "not gpu") check the kernel's tile /
// halo / tail indexing against the oracle without a GPU, by running phase_idct and phase_color for
// every (workgroup, thread) sequentially with the barrier between them.  It is never linked into
// libzjhip.so and nothing in the product path can reach it.
#define ZJ_EMU 1
#include <stdlib.h>
#include <string.h>

#include <vector>

#include "../../zune-jpeg_amd/csrc/zj_plan.h"

using namespace zj;

static int g_variant = 0; // 0 packed generation (staged stores where they apply), 1 wide generation, 2 packed with direct stores
extern "C" void zje_set_variant(int v) { g_variant = v; }
// statistics of the last zje_decode_planes call: blocks by class (classify_block), tiles redone wide
static long long g_cls[3] = {0, 0, 0}, g_redo = 0;
extern "C" void zje_stats(long long out[4]) { out[0] = g_cls[0]; out[1] = g_cls[1]; out[2] = g_cls[2]; out[3] = g_redo; }

template <class C, int HS, int VS, int OUT, bool FAST, bool RAG = false>
static void tile_wide(const Params& p, const TileId t, char* lds)
{
    for (int tid = 0; tid < C::NT; tid++) phase_setup<C, HS, VS, GEN_WIDE>(p, tid, lds);
    /* __syncthreads() */
    for (int tid = 0; tid < C::NT; tid++) {
        const BlockLoc L = locate<C, GEN_WIDE>(p, t, tid, lds);
        U4 raw[8];
        load_block(L, raw);
        finish_block<C, GEN_WIDE, false>(L, raw, lds, 0, p.clamp_dc);
    }
    /* __syncthreads() */
    for (int tid = 0; tid < C::NT; tid++) phase_color<C, HS, VS, OUT, GEN_WIDE, FAST, false, RAG>(p, t, tid, lds);
}

template <int HS, int VS, int OUT, bool FAST, bool RAG = false>
static void run(const Params& p)
{
    using C = Cfg<HS, VS, OUT>;
    constexpr bool NEED_Y16 = OUT == OUT_RGB || OUT == OUT_RGBA || OUT == OUT_RGB_CHW;
    constexpr bool CAN_TS = FAST && C::TSCAP;
    std::vector<char> lds_store(C::LDS_PACKED + 32);
    // 16-byte aligned like a real LDS allocation
    char* lds = (char*)(((uintptr_t)lds_store.data() + 15) & ~(uintptr_t)15);
    const bool ts = CAN_TS && g_variant == 0 && ts_eligible<C>(p, OUT, FAST, RAG);
    for (int bid = 0; bid < p.total_tiles; bid++) {
        memset(lds, 0x7B, C::LDS_PACKED); // poison: unwritten LDS must not matter
        const TileId t = decode_tile(p, bid);
        if (g_variant == 1) { tile_wide<C, HS, VS, OUT, FAST, RAG>(p, t, lds); continue; }
        for (int tid = 0; tid < C::NT; tid++) phase_setup<C, HS, VS, GEN_PACKED>(p, tid, lds);
        /* __syncthreads() */
        const int nblock_lanes = C::HALO_PURE ? C::HALO_T0 : C::NT;
        for (int tid = 0; tid < nblock_lanes; tid++) {
            const BlockLoc L = locate<C, GEN_PACKED>(p, t, tid, lds);
            U4 raw[8];
            load_block(L, raw);
            if (L.valid) g_cls[classify_block((const uint32_t*)raw, lds_tab<C, GEN_PACKED>(lds) + TAB_DW * L.comp + 32)]++;
            finish_block<C, GEN_PACKED, NEED_Y16>(L, raw, lds, 0, p.clamp_dc);
        }
        if (C::HALO_PURE) { // the halo wave: one lane per block column; all lanes do pass 1, then all do pass 2
            HaloLane H[64];
            for (int hl = 0; hl < 64; hl++) {
                H[hl] = halo_locate<C>(p, t, hl, lds);
                int32_t s8[8];
                halo_load(H[hl], s8);
                halo_pass1<C>(H[hl], s8, lds);
            }
            for (int hl = 0; hl < 64; hl++) halo_pass2<C>(H[hl], lds, p.clamp_dc);
            for (int hl = 0; hl < 64; hl++) halo_filter<C, HS, VS>(p, t, hl, lds);
        }
        /* __syncthreads() */
        if (NEED_Y16 && *lds_flag<C>(lds) != 0) { // Q1 value outside a byte: the whole tile again, wide
            g_redo++;
            memset(lds, 0x7B, C::LDS_PACKED);
            tile_wide<C, HS, VS, OUT, FAST, RAG>(p, t, lds);
            continue;
        }
        if (ts) {
            // a round is, per wave: all lanes compute (they read the luma bytes the staging reuses), all lanes stage,
            // all lanes copy out -- the GPU runs these back to back inside each wave, lanes in lockstep
            for (int round = 0; round * C::NT < C::NITEMS; round++)
                for (int w = 0; w < C::NW; w++) {
                    ItemOut io[64];  // hardware wave w plays the logical threads round_tid gives it; its staging bytes are its own
                    for (int l = 0; l < 64; l++) phase_color<C, HS, VS, OUT, GEN_PACKED, FAST, CAN_TS, RAG>(p, t, round_tid<C>(64 * w + l, round), lds, round, &io[l]);
                    for (int l = 0; l < 64; l++) stage_item<C>(io[l], round_tid<C>(64 * w + l, round), lds, round, w);
                    for (int l = 0; l < 64; l++) color_copyout<C, OUT, RAG>(p, t, round_tid<C>(64 * w + l, round), lds, round, w);
                }
        } else {
            for (int tid = 0; tid < C::NT; tid++) phase_color<C, HS, VS, OUT, GEN_PACKED, FAST, false, RAG>(p, t, tid, lds);
        }
    }
}

extern "C" int zje_threads_per_group(const zj_frame_desc* d)
{
    Plan pl;
    int rc = make_plan(d, pl);
    return rc ? rc : pl.nt;
}

// as launch_params of zj_api.cpp: zj_plan.h's launch_mode picks generic / aligned / ragged (the wide generation,
// variant 1, has no ragged form)
static int dispatch(const Plan& pl, const Params& p)
{
    const int mode = launch_mode(pl, g_variant);
#define ZJ_CASE(H, V, O) if (pl.hs == H && pl.vs == V && pl.out == O) { if (mode == 2) run<H, V, O, true, true>(p); else if (mode == 1) run<H, V, O, true>(p); else run<H, V, O, false>(p); return ZJ_OK; }
    ZJ_CASE(1, 1, OUT_RGB) ZJ_CASE(1, 1, OUT_GRAY) ZJ_CASE(1, 1, OUT_YCBCR)
    ZJ_CASE(2, 1, OUT_RGB) ZJ_CASE(2, 1, OUT_GRAY) ZJ_CASE(2, 1, OUT_YCBCR)
    ZJ_CASE(1, 2, OUT_RGB) ZJ_CASE(1, 2, OUT_GRAY) ZJ_CASE(1, 2, OUT_YCBCR)
    ZJ_CASE(2, 2, OUT_RGB) ZJ_CASE(2, 2, OUT_GRAY) ZJ_CASE(2, 2, OUT_YCBCR)
    ZJ_CASE(1, 1, OUT_RGBA) ZJ_CASE(2, 1, OUT_RGBA) ZJ_CASE(1, 2, OUT_RGBA) ZJ_CASE(2, 2, OUT_RGBA)
    ZJ_CASE(1, 1, OUT_RGB_CHW) ZJ_CASE(2, 1, OUT_RGB_CHW) ZJ_CASE(1, 2, OUT_RGB_CHW) ZJ_CASE(2, 2, OUT_RGB_CHW)
#undef ZJ_CASE
    return ZJ_ERR_UNSUPPORTED;
}

extern "C" int zje_decode_planes(const zj_frame_desc* d, size_t nframes, const int16_t* y,
                                 const int16_t* cb, const int16_t* cr, uint8_t* out, int zero_fill)
{
    Plan pl;
    int rc = make_plan(d, pl);
    if (rc) return rc;
    Params p;
    fill_params(d, pl, nframes, y, cb, cr, out, zero_fill, p);
    g_cls[0] = g_cls[1] = g_cls[2] = g_redo = 0;
    if (zero_fill) { // same remainder memset as zj_api.cpp
        size_t off[3], len[3];
        const int nr = uncovered_ranges(d, pl, off, len);
        for (size_t f = 0; f < nframes; f++)
            for (int r = 0; r < nr; r++) memset(out + f * pl.out_len + off[r], 0, len[r]);
    }
    return dispatch(pl, p);
}

// the scattered form (zj_decode_frames_device): per-frame pointers through the launch's table, in launches of at most
// SCATTER_MAX frames, exactly as decode_frames_device_impl of zj_api.cpp cuts them
extern "C" int zje_decode_frames(const zj_frame_desc* d, size_t nframes, const int16_t* const* y, const int16_t* const* cb,
                                 const int16_t* const* cr, uint8_t* const* out, int zero_fill)
{
    Plan pl;
    int rc = make_plan(d, pl);
    if (rc) return rc;
    g_cls[0] = g_cls[1] = g_cls[2] = g_redo = 0;
    const bool chroma = pl.out != OUT_GRAY;
    size_t off[3], len[3];
    const int nr = zero_fill ? uncovered_ranges(d, pl, off, len) : 0;
    for (size_t f0 = 0; f0 < nframes; f0 += SCATTER_MAX) {
        const int n = (int)(nframes - f0 < (size_t)SCATTER_MAX ? nframes - f0 : (size_t)SCATTER_MAX);
        Params p;
        fill_params(d, pl, (size_t)n, nullptr, nullptr, nullptr, nullptr, zero_fill, p);
        set_scatter(p, y, chroma ? cb : nullptr, chroma ? cr : nullptr, out, f0, n);
        for (int f = 0; f < n; f++)
            for (int r = 0; r < nr; r++) memset(out[f0 + f] + off[r], 0, len[r]);
        if ((rc = dispatch(pl, p))) return rc;
    }
    return ZJ_OK;
}

// ---- block level: the two transforms and the guard on their own (tests/test_packed_idct.py) -------------------
// coeff: nblocks x 64 (natural order), q: 64 entries 0..255
extern "C" void zje_classify(const int16_t* coeff, size_t nblocks, const int32_t q[64], int* cls)
{
    uint32_t tab[TAB_DW];
    build_table(q, tab);
    for (size_t b = 0; b < nblocks; b++) cls[b] = classify_block((const uint32_t*)(coeff + 64 * b), tab + 32);
}
// idct_block_packed WITHOUT consulting the guard (so a test can show both that it is exact under the guard and that
// the guard is needed); out: nblocks x 64 bytes, row-major 8x8
extern "C" void zje_idct_packed(const int16_t* coeff, size_t nblocks, const int32_t q[64], uint8_t* out)
{
    uint32_t tab[TAB_DW];
    build_table(q, tab);
    for (size_t b = 0; b < nblocks; b++) {
        U4 raw[8];
        memcpy(raw, coeff + 64 * b, 128);
        uint32_t px[16];
        idct_block_packed(raw, tab, px);
        memcpy(out + 64 * b, px, 64);
    }
}
// idct_block (wide); out: nblocks x 64 int16
extern "C" void zje_idct_wide(const int16_t* coeff, size_t nblocks, const int32_t q[64], int16_t* out)
{
    uint32_t tab[TAB_DW];
    build_table(q, tab);
    for (size_t b = 0; b < nblocks; b++) {
        U4 raw[8], px[8];
        memcpy(raw, coeff + 64 * b, 128);
        idct_block(raw, (const uint16_t*)tab, px);
        memcpy(out + 64 * b, px, 128);
    }
}
extern "C" int zje_guard_limit(void) { return GUARD_LIMIT; }

// tile_from_id's division by a launch constant (magic_u31 / magic_div): quotients of n[0..count) by d, and the first n
// of an exhaustive sweep [lo, hi) where it disagrees with `/` (-1: none)
extern "C" void zje_magic_div(uint32_t d, const uint32_t* n, size_t count, uint32_t* q)
{
    const Magic g = magic_u31(d);
    for (size_t i = 0; i < count; i++) q[i] = magic_div(n[i], g);
}
extern "C" long long zje_magic_sweep(uint32_t d, uint32_t lo, uint32_t hi)
{
    const Magic g = magic_u31(d);
    for (uint32_t n = lo; n < hi; n++)
        if (magic_div(n, g) != n / d) return (long long)n;
    return -1;
}
extern "C" void zje_tile_from_id(int nframes, int n_strips, int tiles_per_row, int id, int out[3])
{
    Params p;
    set_grid(p, nframes, n_strips, tiles_per_row);
    const TileId t = tile_from_id(p, id);
    out[0] = t.frame; out[1] = t.strip; out[2] = t.tile;
}

// ---- the GPU entropy stage (zj_huff_device.h), thread by thread -----------------------------------------------------
#include "../../zune-jpeg_amd/csrc/zj_huff_device.h"

// planes: three separate buffers here, so the caller zero-fills them (on the device round 0 clears the one allocation that
// holds all three, huff_clear_planes).  stats: [0] rounds run,
// [1] sub-sequences, [2] sub-sequence decodes over all sync rounds, [3] first-seen MCU of the EOI rule
extern "C" int zje_huff_decode(const uint8_t* blob, int16_t* y, int16_t* cb, int16_t* cr, uint32_t* status, uint32_t* stats)
{
    const HuffScan* g = huff_hdr(blob);
    if (g->magic != HUFF_MAGIC) return ZJ_ERR_ARG;
    const uint32_t nsub = g->nsub;
    std::vector<unsigned long long> exitv(nsub), exit_before(nsub);
    std::vector<HuffI4> aux(nsub), base(nsub);
    std::vector<uint32_t> list(2 * (size_t)HUFF_LIST_FACTOR * nsub);
    std::vector<uint8_t> rel(nsub);
    std::vector<uint32_t> ctl(HUFF_CTL_WORDS, 0);
    const uint32_t nscan = (nsub + HUFF_SCAN_WG - 1) / HUFF_SCAN_WG;
    std::vector<HuffAgg> wgagg(nscan), wgpre(nscan);
    HuffArgs a;
    a.blob = blob; a.exit = exitv.data(); a.exit_rd = exit_before.data(); a.aux = aux.data(); a.base = base.data();
    a.list = list.data(); a.rel = rel.data();
    a.wgagg = wgagg.data(); a.wgpre = wgpre.data();
    a.ctl = ctl.data(); a.plane[0] = y; a.plane[1] = cb; a.plane[2] = cr; a.round = 0;
    a.zero_base = nullptr; a.zero_pieces = 0;
    a.spread = getenv("ZJE_SPREAD") ? atoi(getenv("ZJE_SPREAD")) : 3; // lanes per work-list entry in the sparse rounds
    std::vector<HuffLds> lds(1);
    HuffLds& L = lds[0];
    uint32_t work = 0;
    int round = 0;
    for (;; round++) {
        if (round > HUFF_MAX_ROUNDS) { ctl[0] |= HUFF_ST_NO_SYNC; break; }
        a.round = round;
        if (g->nper && huff_periodic_before(round)) // the periodic-run rule in front of this round
            for (uint32_t i = 0; i < nsub; i++) huff_periodic_thread(a, i, round);
        if (round >= 2 && ctl[HUFF_CTL_ROUND0 + round - 1] == 0) break; // (the device's no-op rounds)
        // the device runs all threads of a round at once: nobody sees an exit state of the SAME round -- the round reads
        // its predecessors' states from a copy taken before it
        exit_before = exitv;
        a.exit_rd = exit_before.data();
        uint32_t todo = nsub;
        if (round >= 2) {
            uint32_t entries = ctl[HUFF_CTL_ROUND0 + round - 1];
            if (entries > HUFF_LIST_FACTOR * nsub) entries = HUFF_LIST_FACTOR * nsub;
            todo = entries * huff_spread(a, nsub, entries);
        }
        for (uint32_t wg = 0; wg * HUFF_WG < todo; wg++) {
            uint32_t pick[HUFF_WG];
            memset((void*)&L, 0x7B, sizeof L);
            for (int tid = 0; tid < HUFF_WG; tid++) {
                pick[tid] = huff_sync_pick(a, wg * HUFF_WG + tid, nsub, huff_subs(blob));
                huff_stage<HUFF_WG>(blob, tid, pick[tid] < nsub, pick[tid], L);
            }
            for (int tid = 0; tid < HUFF_WG; tid++) {
                if (pick[tid] < nsub) work++;
                huff_sync_thread(a, L, (uint32_t)tid, pick[tid]);
            }
        }
        if (round >= 1 && ctl[HUFF_CTL_ROUND0 + round] == 0) break;
    }
    const uint32_t nwg = (nsub + HUFF_WG - 1) / HUFF_WG;
    a.exit_rd = exitv.data();
    if (!(ctl[0] & HUFF_ST_NO_SYNC)) {
        for (uint32_t w = 0; w < nscan; w++) { // a prefix-sum workgroup: the scan the device does in log steps
            HuffAgg run = huff_scan_identity();
            for (uint32_t t = 0; t < (uint32_t)HUFF_SCAN_WG; t++) {
                const uint32_t i = w * HUFF_SCAN_WG + t;
                const HuffAgg el = huff_scan_element(a, i);
                huff_scan_store(a, i, run);
                run = huff_scan_op(run, el);
            }
            wgagg[w] = run;
        }
        huff_scan_totals(a, nscan);
        for (uint32_t wg = 0; wg < nwg; wg++) {
            memset((void*)&L, 0x7B, sizeof L);
            for (int tid = 0; tid < HUFF_WG; tid++) huff_stage<HUFF_WG>(blob, tid, wg * HUFF_WG + tid < nsub, wg * HUFF_WG + tid, L);
            for (int tid = 0; tid < HUFF_WG; tid++) huff_write_thread(a, L, (uint32_t)tid, wg * HUFF_WG + tid);
        }
        uint32_t first = 0;
        const uint32_t pieces = huff_cut_plan(a, &first);
        for (uint32_t p = 0; p < pieces; p++) huff_cut_clear(a, first, p);
    }
    if (status) *status = ctl[0];
    if (stats) { stats[0] = (uint32_t)round; stats[1] = nsub; stats[2] = work; stats[3] = ~ctl[HUFF_CTL_SEEN]; }
    return ZJ_OK;
}



// int16 elements of component c's plane as the scan's header states them
extern "C" size_t zje_huff_plane_len(const uint8_t* blob, int c)
{
    const HuffScan* g = huff_hdr(blob);
    return c >= 0 && c < (int)g->ncomp ? (size_t)g->comp[c].bw * g->comp[c].bh * 64 : 0;
}
