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

static int g_compact = 0, g_persistent_wgs = 24;
extern "C" void zje_set_variant(int compact) { g_compact = compact; }
extern "C" void zje_set_persistent_wgs(int n) { g_persistent_wgs = n; }

// phase 2 for every lane of the workgroup; with transposed stores (variants 4 and 7) a round is two half-steps per
// wave: all lanes stage, then all lanes copy out (the GPU runs them back to back inside each wave)
template <class C, int HS, int VS, int OUT, bool FAST>
static void color_all(const Params& p, const TileId t, int16_t* lds)
{
    constexpr bool CAN_TS = FAST && (OUT == OUT_RGB || OUT == OUT_YCBCR);
    if (CAN_TS && (g_compact & 4) && ts_eligible<C>(p, OUT, FAST)) {
        for (int round = 0; round * C::NT < C::NITEMS; round++)
            for (int w = 0; w < C::NT / 64; w++) {
                for (int l = 0; l < 64; l++) phase_color<C, HS, VS, OUT, FAST, CAN_TS>(p, t, 64 * w + l, lds, round);
                for (int l = 0; l < 64; l++) color_copyout<C, OUT>(p, t, 64 * w + l, lds, round);
            }
        return;
    }
    for (int tid = 0; tid < C::NT; tid++) phase_color<C, HS, VS, OUT, FAST>(p, t, tid, lds);
}

template <int HS, int VS, int OUT, bool FAST>
static void run(const Params& p)
{
    using C = Cfg<HS, VS, OUT>;
    std::vector<char> lds_store(C::LDS_BYTES_TS + 32);
    // 16-byte aligned like a real LDS allocation
    int16_t* lds = (int16_t*)(((uintptr_t)lds_store.data() + 15) & ~(uintptr_t)15);
    if (g_compact == 2 && FAST) { // persistent walk: every tile exactly once, in each workgroup's order
        const int nwg = g_persistent_wgs < p.total_tiles ? g_persistent_wgs : p.total_tiles;
        for (int wg = 0; wg < nwg; wg++) {
            memset(lds, 0x7B, C::LDS_BYTES_COMPACT);
            for (int tid = 0; tid < C::NT; tid++) phase_setup<C, HS, VS>(p, tid, lds);
            const TileWalk w = persistent_walk(p, wg, nwg);
            for (int id = w.first; id < w.last; id += w.step) {
                const TileId t = tile_from_id(p, id);
                for (int tid = 0; tid < C::NT; tid++) {
                    const BlockLoc L = locate<C>(p, t, tid, lds);
                    U4 raw[8];
                    load_block(L, raw);
                    finish_block<C>(L, raw, lds, 0, p.clamp_dc);
                }
                for (int tid = 0; tid < C::NT; tid++) phase_color<C, HS, VS, OUT, FAST>(p, t, tid, lds);
            }
        }
        return;
    }
    for (int bid = 0; bid < p.total_tiles; bid++) {
        memset(lds, 0x7B, C::LDS_BYTES_TS); // poison: unwritten LDS must not matter
        const TileId t = decode_tile(p, bid);
        for (int tid = 0; tid < C::NT; tid++) phase_setup<C, HS, VS>(p, tid, lds);
        /* __syncthreads() */
        if ((g_compact & 3) == 3 && FAST) { // work stealing: stage (all lanes), barrier, take + IDCT (all lanes)
            std::vector<StealState> st(C::NT);
            const int donor = (t.tile + t.strip + t.frame) % (C::NT / 64); // rotates like the kernel's
            for (int tid = 0; tid < C::NT; tid++) {
                const BlockLoc L = locate<C>(p, t, tid, lds);
                U4 raw[8];
                load_block(L, raw);
                st[tid] = steal_stage<C>(L, raw, p.qt[64 * L.comp], tid, lds, p.clamp_dc, donor);
            }
            /* __syncthreads() */
            for (int tid = 0; tid < C::NT; tid++) {
                const BlockLoc L = locate<C>(p, t, tid, lds);
                U4 raw[8];
                load_block(L, raw); // registers survive the barrier on the GPU; the emulator reloads
                steal_idct<C>(L, raw, st[tid], tid, lds, donor);
            }
            color_all<C, HS, VS, OUT, FAST>(p, t, lds);
            continue;
        }
        for (int tid = 0; tid < C::NT; tid++) {
            const BlockLoc L = locate<C>(p, t, tid, lds);
            U4 raw[8];
            load_block(L, raw);
            if (g_compact == 1 && FAST) classify_stage<C>(L, raw, p.qt[64 * L.comp], tid, lds, p.clamp_dc);
            else finish_block<C>(L, raw, lds, 0, p.clamp_dc);
        }
        /* __syncthreads() */
        if (g_compact == 1 && FAST) {
            for (int tid = 0; tid < C::NT; tid++) idct_queue<C>(tid, lds);
            /* __syncthreads() */
        }
        color_all<C, HS, VS, OUT, FAST>(p, t, lds);
    }
}

extern "C" int zje_threads_per_group(const zj_frame_desc* d)
{
    Plan pl;
    int rc = make_plan(d, pl);
    return rc ? rc : pl.nt;
}

extern "C" int zje_decode_planes(const zj_frame_desc* d, size_t nframes, const int16_t* y,
                                 const int16_t* cb, const int16_t* cr, uint8_t* out, int zero_fill)
{
    Plan pl;
    int rc = make_plan(d, pl);
    if (rc) return rc;
    Params p;
    fill_params(d, pl, nframes, y, cb, cr, out, &d->qt[0][0], zero_fill, p);
    if (zero_fill) { // same remainder memset as zj_api.cpp
        size_t off[3], len[3];
        const int nr = uncovered_ranges(d, pl, off, len);
        for (size_t f = 0; f < nframes; f++)
            for (int r = 0; r < nr; r++) memset(out + f * pl.out_len + off[r], 0, len[r]);
    }
#define ZJ_CASE(H, V, O) if (pl.hs == H && pl.vs == V && pl.out == O) { if (pl.fast) run<H, V, O, true>(p); else run<H, V, O, false>(p); return ZJ_OK; }
    ZJ_CASE(1, 1, OUT_RGB) ZJ_CASE(1, 1, OUT_GRAY) ZJ_CASE(1, 1, OUT_YCBCR)
    ZJ_CASE(2, 1, OUT_RGB) ZJ_CASE(2, 1, OUT_GRAY) ZJ_CASE(2, 1, OUT_YCBCR)
    ZJ_CASE(1, 2, OUT_RGB) ZJ_CASE(1, 2, OUT_GRAY) ZJ_CASE(1, 2, OUT_YCBCR)
    ZJ_CASE(2, 2, OUT_RGB) ZJ_CASE(2, 2, OUT_GRAY) ZJ_CASE(2, 2, OUT_YCBCR)
    ZJ_CASE(1, 1, OUT_RGBA) ZJ_CASE(2, 1, OUT_RGBA) ZJ_CASE(1, 2, OUT_RGBA) ZJ_CASE(2, 2, OUT_RGBA)
    ZJ_CASE(1, 1, OUT_RGB_CHW) ZJ_CASE(2, 1, OUT_RGB_CHW) ZJ_CASE(1, 2, OUT_RGB_CHW) ZJ_CASE(2, 2, OUT_RGB_CHW)
#undef ZJ_CASE
    return ZJ_ERR_UNSUPPORTED;
}
