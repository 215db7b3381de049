// zj_plan.h -- host-side geometry/launch planning shared by the product library (zj_api.cpp) and the
// CPU emulation harness used by the CPU test-suite (tests/emu).  No device code here.
//
// Geometry follows src/headers.rs:306-339 (mcu_x, mcu_y, width_stride) and the strip loop of
// src/mcu_prog.rs:132-246 / src/mcu.rs:139-230 (paths relative to the reference tree).
#pragma once

#include <stddef.h>
#include <stdint.h>

#include "../../include/zjhip.h"
#include "zj_device.h"

namespace zj {

struct Plan {
    int hs, vs;          // luma sampling factors
    int out;             // OUT_RGB / OUT_GRAY / OUT_YCBCR / OUT_RGBA / OUT_RGB_CHW
    int plain;           // OUT_RGB with ZJ_FLAG_PLAIN_TAIL
    int clamp_dc;        // ZJ_FLAG_CLAMP_DC
    int edge_rep;        // ZJ_FLAG_EDGE_REPLICATE (meaningful when hs == 2)
    int mcu_x, mcu_y;
    int n_strips;        // strips the reference's zip() would process
    int strip_rows;      // luma rows per strip
    int tiles_per_row;
    int nt;              // threads per workgroup
    size_t y_len, c_len; // i16 elements per plane
    size_t out_len;      // bytes per frame (out_pitch x height, CHW: x 3)
    size_t row_bytes;    // bytes of one output row (CHW: of a plane's row): width x components
    size_t out_pitch;    // bytes between rows: zj_frame_desc.out_pitch, or row_bytes when that is 0
    int ncomp_out;
    bool fast;           // aligned fast path (W % 16 == 0, W >= 32)
    int regular_px;      // !fast: pixels of a row made of ordinary 16-pixel groups; 0: the generic kernels (see make_plan)
    int rows_covered;    // n_strips * strip_rows; rows below stay 0 in the reference (Q6)
};

inline int ncomp_of(int cs)
{
    switch (cs) {
    case ZJ_CS_RGB: case ZJ_CS_YCBCR: return 3;
    case ZJ_CS_GRAYSCALE: return 1;
    case ZJ_CS_CMYK: case ZJ_CS_YCCK: case ZJ_CS_RGBA: case ZJ_CS_RGBX: return 4;
    default: return 0;
    }
}

template <int HS, int VS>
inline void plan_geo(Plan& pl, bool chroma)
{
    if (chroma) {
        using C = Cfg<HS, VS, OUT_RGB>;
        pl.strip_rows = C::SH; pl.tiles_per_row = (pl.mcu_x + C::TWC - 1) / C::TWC; pl.nt = C::NT;
    } else {
        using C = Cfg<HS, VS, OUT_GRAY>;
        pl.strip_rows = C::SH; pl.tiles_per_row = (pl.mcu_x + C::TWC - 1) / C::TWC; pl.nt = C::NT;
    }
}

// Returns ZJ_OK or an error status.
inline int make_plan(const zj_frame_desc* d, Plan& pl)
{
    if (!d || d->width == 0 || d->height == 0) return ZJ_ERR_ARG;
    if (!((d->h_max == 1 || d->h_max == 2) && (d->v_max == 1 || d->v_max == 2))) return ZJ_ERR_ARG;
    if (d->in_components != 1 && d->in_components != 3) return ZJ_ERR_ARG;
    if (d->width > 65535 || d->height > 65535) return ZJ_ERR_ARG; // u16 in the reference (decoder.rs:652)
    const int nout = ncomp_of(d->out_colorspace);
    if (nout == 0) return ZJ_ERR_ARG;
    // grayscale JPEG with a down-sampled component (mcu.rs:170-196) is reset to (1,1) by the
    // reference with a warning; callers must pass (1,1) here.
    if (d->in_components == 1 && (d->h_max != 1 || d->v_max != 1)) return ZJ_ERR_UNSUPPORTED;
    for (int c = 0; c < 3; c++)
        for (int k = 0; k < 64; k++)
            if (d->qt[c][k] < 0 || d->qt[c][k] > 255) return ZJ_ERR_UNSUPPORTED; // 8-bit DQT only
    pl.hs = (int)d->h_max;
    pl.vs = (int)d->v_max;
    if (d->out_colorspace == ZJ_CS_GRAYSCALE) pl.out = OUT_GRAY;
    else if (d->out_colorspace == ZJ_CS_RGB && d->in_components == 3) pl.out = OUT_RGB;
    else if (d->out_colorspace == ZJ_CS_YCBCR && d->in_components == 3) pl.out = OUT_YCBCR;
    // RGBA/RGBX are malformed in the reference itself; here they are an extension: R G B 255
    else if ((d->out_colorspace == ZJ_CS_RGBA || d->out_colorspace == ZJ_CS_RGBX) && d->in_components == 3) pl.out = OUT_RGBA;
    else return ZJ_ERR_UNSUPPORTED; // CMYK/YCCK are no-ops in the reference
    if (d->flags & ~(uint32_t)ZJ_FLAG_CORRECTED) return ZJ_ERR_ARG;
    pl.clamp_dc = (d->flags & ZJ_FLAG_CLAMP_DC) ? 1 : 0;
    pl.edge_rep = ((d->flags & ZJ_FLAG_EDGE_REPLICATE) && pl.hs == 2) ? 1 : 0;
    pl.plain = (pl.out == OUT_RGB && (d->flags & ZJ_FLAG_PLAIN_TAIL)) ? 1 : 0;
    if (d->out_layout == ZJ_LAYOUT_CHW) {
        if (pl.out == OUT_RGB) pl.out = OUT_RGB_CHW;            // planar u8 tensor layout, every pixel at its own place
        else if (pl.out != OUT_GRAY) return ZJ_ERR_UNSUPPORTED; // (one plane: CHW == HWC)
        pl.plain = 0;
    } else if (d->out_layout != ZJ_LAYOUT_HWC) return ZJ_ERR_ARG;
    pl.ncomp_out = nout;
    pl.mcu_x = (int)((d->width + 8 * d->h_max - 1) / (8 * d->h_max));  // headers.rs:317
    pl.mcu_y = (int)((d->height + 8 * d->v_max - 1) / (8 * d->v_max)); // headers.rs:319
    pl.y_len = (size_t)pl.mcu_x * 64 * d->v_max * d->h_max * pl.mcu_y; // mcu_prog.rs:76
    pl.c_len = d->in_components == 3 ? (size_t)pl.mcu_x * 64 * pl.mcu_y : 0;
    // rows may be laid out wider than they are (a pitch that is a multiple of 128 bytes keeps every tile's row segment on
    // whole cache lines: DESIGN.md 4.0 "row pitch"); the reference's own layout is the tight one
    pl.row_bytes = pl.out == OUT_RGB_CHW ? (size_t)d->width : (size_t)d->width * nout;
    pl.out_pitch = d->out_pitch ? (size_t)d->out_pitch : pl.row_bytes;
    if (pl.out_pitch < pl.row_bytes || pl.out_pitch > (1u << 20)) return ZJ_ERR_ARG;
    pl.out_len = pl.out_pitch * d->height * (pl.out == OUT_RGB_CHW ? 3 : 1);
    const bool chroma = pl.out != OUT_GRAY;
    if (pl.hs == 1 && pl.vs == 1) plan_geo<1, 1>(pl, chroma);
    else if (pl.hs == 2 && pl.vs == 1) plan_geo<2, 1>(pl, chroma);
    else if (pl.hs == 1 && pl.vs == 2) plan_geo<1, 2>(pl, chroma);
    else plan_geo<2, 2>(pl, chroma);
    // strips: (2,1) and (2,2) take two MCU rows per strip, an odd last MCU row is dropped by
    // chunks_exact (mcu_prog.rs:191-205; mcu.rs:145-156 `mcu_y / 2`)
    const int mcu_rows_per_strip = (pl.hs == 2) ? 2 : 1;
    pl.n_strips = pl.mcu_y / mcu_rows_per_strip;
    // the zip with out_chunks (mcu_prog.rs:188-189) can only cut it shorter on absurd aspect ratios
    const size_t interleaved = (pl.hs != 1 || pl.vs != 1) ? 1 : 0;
    const size_t total = ((size_t)d->width + 8) * ((size_t)d->height + 8) * nout + interleaved * 128 * d->height * nout;
    const size_t chunk = (size_t)d->width * nout * 8 * d->h_max * d->v_max;
    if (total / chunk < (size_t)pl.n_strips) pl.n_strips = (int)(total / chunk);
    pl.fast = (d->width % 16 == 0) && d->width >= 32;
    if (pl.fast && (pl.out_pitch & 15)) return ZJ_ERR_ARG; // the aligned kernels store 16 bytes per lane at 16-byte-aligned addresses
    // A ragged width (the reference's own medium images are 2500 pixels wide, tests/medium_images.rs) is irregular only
    // at the END of its rows: the last two 8-pixel units are written early (Q5), what lies between them and the padded
    // width is never converted, the row is clipped at 3W (worker.rs:143-251).  The tail region starts at most 61 bytes
    // below 48 * elements (elements = P/16 - 1, worker.rs:171), so every 16-pixel group G <= elements - 3 is an ordinary
    // one: 48 bytes at 48 * G, all inside the row.  Those groups take the aligned fast path (the staged, lane-contiguous
    // stores; a row may start at any byte, global stores take any alignment); the few groups beyond them take the generic
    // stores, by the lanes that hold them, in the same launch (the RAG instantiations, zj_device.h: phase_color).
    pl.regular_px = 0;
    if (!pl.fast && d->width >= 64) {
        const int P = pl.mcu_x * 8 * pl.hs, elements = P / 16 - 1;
        if (elements > 2) pl.regular_px = 16 * (elements - 2); // groups 0 .. elements - 3
    }
    if (pl.out == OUT_GRAY && pl.n_strips > 0) {
        // ycbcr_to_grayscale re-derives the row count as len/width (color_convert/scalar.rs:97-99);
        // when padding makes that exceed the real row count it walks off its output chunk and panics.
        const size_t P = (size_t)pl.mcu_x * 8 * d->h_max, rows = (size_t)pl.strip_rows;
        if (rows * P / d->width != rows) return ZJ_ERR_PANIC;
    }
    pl.rows_covered = pl.n_strips * pl.strip_rows;
    return ZJ_OK;
}

// Byte ranges of one frame's output that the strips never reach (rows below the last complete strip,
// Q6: the reference leaves them 0).  HWC: one range; CHW: one per plane.  Returns the number of ranges.
inline int uncovered_ranges(const zj_frame_desc* d, const Plan& pl, size_t off[3], size_t len[3])
{
    // (whole rows of out_pitch bytes: with a padded pitch the padding of THESE rows is zeroed along with them)
    const size_t H = d->height, Pb = pl.out_pitch;
    const size_t covered = (size_t)pl.rows_covered < H ? (size_t)pl.rows_covered : H;
    if (covered >= H) return 0;
    if (pl.out == OUT_RGB_CHW) {
        for (int c = 0; c < 3; c++) { off[c] = (size_t)c * Pb * H + covered * Pb; len[c] = (H - covered) * Pb; }
        return 3;
    }
    off[0] = covered * Pb;
    len[0] = pl.out_len - off[0];
    return 1;
}

// the launch grid: nframes x n_strips x tiles_per_row workgroups, and the multipliers tile_from_id divides with
inline void set_grid(Params& p, int nframes, int n_strips, int tiles_per_row)
{
    p.nframes = nframes; p.n_strips = n_strips; p.tiles_per_row = tiles_per_row;
    p.total_tiles = nframes * n_strips * tiles_per_row;
    const Magic gt = magic_u31((uint32_t)(tiles_per_row > 0 ? tiles_per_row : 1)), gs = magic_u31((uint32_t)(n_strips > 0 ? n_strips : 1));
    p.tpr_magic = gt.m; p.tpr_shift = gt.s; p.ns_magic = gs.m; p.ns_shift = gs.s;
    p.stagger_wgs = p.stagger_delay = 0; p.stagger_magic = p.stagger_shift = 0;
}

inline void fill_params(const zj_frame_desc* d, const Plan& pl, size_t nframes, const int16_t* y,
                        const int16_t* cb, const int16_t* cr, uint8_t* out, int zero_fill, Params& p)
{
    p.y = y; p.cb = cb; p.cr = cr; p.out = out;
    // the tables travel in the kernel arguments (no device copy to order against other streams); the reference
    // snapshots them per component at SOF time (headers.rs:327), d->qt is that snapshot
    for (int c = 0; c < 3; c++) build_table(d->qt[c], p.tab + TAB_DW * c);
    p.y_frame_stride = (long long)pl.y_len;
    p.c_frame_stride = (long long)pl.c_len;
    p.out_frame_stride = (long long)pl.out_len;
    p.width = (int)d->width; p.height = (int)d->height;
    p.mcu_x = pl.mcu_x; p.zero_fill = zero_fill;
    set_grid(p, (int)nframes, pl.n_strips, pl.tiles_per_row);
#if defined(ZJ_ABLATION)
    p.debug = 0;
#endif
    p.plain = pl.plain;
    p.clamp_dc = pl.clamp_dc;
    p.edge_rep = pl.edge_rep;
    p.out_pitch = (int)pl.out_pitch;
    p.plane_stride = (long long)pl.out_pitch * d->height;
    p.regular_px = pl.regular_px;
}

// which family of fused kernels a plan launches: 0 generic stores (any width), 1 aligned fast path, 2 ragged fast path
// (RAG).  The wide generation (variant 1) has no RAG form: its ragged frames take the generic kernels.
inline int launch_mode(const Plan& pl, int variant) { return pl.fast ? 1 : ((pl.regular_px > 0 && variant != 1) ? 2 : 0); }

// frames [f0, f0 + n) of a scattered batch: their addresses into the launch's table (n <= SCATTER_MAX)
inline void set_scatter(Params& p, const int16_t* const* y, const int16_t* const* cb, const int16_t* const* cr,
                        uint8_t* const* out, size_t f0, int n)
{
    p.y = nullptr; p.cb = nullptr; p.cr = nullptr; p.out = nullptr; // y == nullptr marks the launch as scattered (decode_tile)
    for (int f = 0; f < SCATTER_MAX; f++) {
        const bool in = f < n;
        p.fptr[f][0] = in ? (uint64_t)(uintptr_t)y[f0 + f] : 0;
        p.fptr[f][1] = in && cb ? (uint64_t)(uintptr_t)cb[f0 + f] : 0;
        p.fptr[f][2] = in && cr ? (uint64_t)(uintptr_t)cr[f0 + f] : 0;
        p.fptr[f][3] = in ? (uint64_t)(uintptr_t)out[f0 + f] : 0;
    }
}

} // namespace zj
