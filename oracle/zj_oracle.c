/*
 * zj_oracle.c -- CPU restatement (ORACLE) of zune-jpeg's scalar post-entropy pixel path.
 *
 * TEST INFRASTRUCTURE ONLY (see zj_oracle.h).  Plain C, scalar, one thread; written for fidelity
 * to the reference's scalar arms, not for speed.  Rust release-mode integer semantics are
 * restated explicitly: i32/i16 `+ - *` wrap (two's complement), `>>` on signed is arithmetic,
 * `as u8`/`as i16` truncate.  All wrap-around goes through the w32_/w16_ helpers below so the file
 * does not depend on -fwrapv.
 *
 * Reference paths are relative to /root/reference.
 */
#include "zj_oracle.h"

#include <stdlib.h>
#include <string.h>

/* ---- wrapping integer helpers (Rust release semantics) ------------------------------------ */
static inline int32_t w32_add(int32_t a, int32_t b) { return (int32_t)((uint32_t)a + (uint32_t)b); }
static inline int32_t w32_sub(int32_t a, int32_t b) { return (int32_t)((uint32_t)a - (uint32_t)b); }
static inline int32_t w32_mul(int32_t a, int32_t b) { return (int32_t)((uint32_t)a * (uint32_t)b); }
static inline int32_t w32_shl(int32_t a, int s) { return (int32_t)((uint32_t)a << s); }
static inline int32_t sar32(int32_t a, int s) { return a >> s; } /* gcc/clang: arithmetic */
static inline int16_t w16(int32_t a) { return (int16_t)(uint16_t)(uint32_t)a; }
static inline int16_t w16_add(int16_t a, int16_t b) { return w16((int32_t)a + (int32_t)b); }
static inline int16_t w16_sub(int16_t a, int16_t b) { return w16((int32_t)a - (int32_t)b); }
static inline int16_t w16_mul(int16_t a, int16_t b) { return w16((int32_t)a * (int32_t)b); }
static inline int16_t sar16(int16_t a, int s) { return (int16_t)(a >> s); }

/* misc.rs:113-121 ColorSpace::num_components */
size_t zjo_num_components(int cs)
{
    switch (cs) {
    case ZJO_CS_RGB:
    case ZJO_CS_YCBCR: return 3;
    case ZJO_CS_CMYK:
    case ZJO_CS_RGBA:
    case ZJO_CS_RGBX:
    case ZJO_CS_YCCK: return 4;
    case ZJO_CS_GRAYSCALE: return 1;
    default: return 0;
    }
}

/* ===========================================================================================
 * IDCT  --  src/idct/scalar.rs
 * =========================================================================================== */

/* scalar.rs:6 */
#define SCALE_BITS (512 + 65536 + (128 << 17))

/* scalar.rs:302-305 clamp */
static inline int16_t idct_clamp(int32_t a)
{
    if (a < 0) a = 0;
    if (a > 255) a = 255;
    return (int16_t)a;
}

/* One 8-point pass shared by the column loop (scalar.rs:79-167) and the row loop (:170-274).
 * s[0..8] are the inputs along the transform axis, bias = 512 or SCALE_BITS, o[] un-shifted sums. */
static inline void idct_pass(const int32_t s[8], int32_t bias, int32_t o[8])
{
    /* even part, scalar.rs:81-107 / :175-201 */
    int32_t p2 = s[2], p3 = s[6];
    int32_t p1 = w32_mul(w32_add(p2, p3), 2217);
    int32_t t2 = w32_add(p1, w32_mul(p3, -7567));
    int32_t t3 = w32_add(p1, w32_mul(p2, 3135));
    p2 = s[0];
    p3 = s[4];
    int32_t t0 = w32_shl(w32_add(p2, p3), 12); /* fsh, scalar.rs:294 */
    int32_t t1 = w32_shl(w32_sub(p2, p3), 12);
    int32_t x0 = w32_add(w32_add(t0, t3), bias);
    int32_t x3 = w32_add(w32_sub(t0, t3), bias);
    int32_t x1 = w32_add(w32_add(t1, t2), bias);
    int32_t x2 = w32_add(w32_sub(t1, t2), bias);
    /* odd part, scalar.rs:109-148 / :203-245 */
    t0 = s[7];
    t1 = s[5];
    t2 = s[3];
    t3 = s[1];
    p3 = w32_add(t0, t2);
    int32_t p4 = w32_add(t1, t3);
    p1 = w32_add(t0, t3);
    p2 = w32_add(t1, t2);
    int32_t p5 = w32_mul(w32_add(p3, p4), 4816); /* f2f(1.175875602) == 4816, scalar.rs:224,287 */
    t0 = w32_mul(t0, 1223);
    t1 = w32_mul(t1, 8410);
    t2 = w32_mul(t2, 12586);
    t3 = w32_mul(t3, 6149);
    p1 = w32_add(p5, w32_mul(p1, -3685));
    p2 = w32_add(p5, w32_mul(p2, -10497));
    p3 = w32_mul(p3, -8034);
    p4 = w32_mul(p4, -1597);
    t3 = w32_add(t3, w32_add(p1, p4));
    t2 = w32_add(t2, w32_add(p2, p3));
    t1 = w32_add(t1, w32_add(p2, p4));
    t0 = w32_add(t0, w32_add(p1, p3));
    /* scalar.rs:152-166 / :255-269 */
    o[0] = w32_add(x0, t3);
    o[1] = w32_add(x1, t2);
    o[2] = w32_add(x2, t1);
    o[3] = w32_add(x3, t0);
    o[4] = w32_sub(x3, t0);
    o[5] = w32_sub(x2, t1);
    o[6] = w32_sub(x1, t2);
    o[7] = w32_sub(x0, t3);
}

/* src/idct/scalar.rs:19-282 */
int zjo_idct_strip(const int16_t *coeff, size_t n, const int32_t qt[64], size_t stride,
                   size_t samp_factors, size_t v_samp, int16_t *out)
{
    if (samp_factors == 0) return ZJO_ERR_PANIC; /* division by zero, scalar.rs:30 */
    memset(out, 0, n * sizeof(int16_t));          /* vec![0; len], scalar.rs:26 */
    size_t chunks = n * v_samp / samp_factors;    /* scalar.rs:30 */
    if (chunks == 0) return n == 0 ? ZJO_OK : ZJO_ERR_PANIC; /* chunks_exact(0) panics */

    for (size_t c0 = 0; c0 + chunks <= n; c0 += chunks) { /* chunks_exact, scalar.rs:32-34 */
        const int16_t *in_chunk = coeff + c0;
        int16_t *out_chunk = out + c0;
        size_t pos = 0, x = 0;
        for (size_t b0 = 0; b0 + 64 <= chunks; b0 += 64) { /* chunks_exact(64), scalar.rs:40 */
            const int16_t *v = in_chunk + b0;
            int dc_only = 1;
            for (int k = 1; k < 64; k++)
                if (v[k] != 0) { dc_only = 0; break; } /* scalar.rs:45 */
            if (dc_only) {
                /* scalar.rs:48: ((vector[0].wrapping_mul(qt[0] as i16)) >> 3) + 128, all i16, NO clamp */
                int16_t val = w16_add(sar16(w16_mul(v[0], w16(qt[0])), 3), 128);
                for (int r = 0; r < 8; r++) { /* store! x8, scalar.rs:50-74 */
                    if (pos + 8 > chunks) return ZJO_ERR_PANIC;
                    for (int k = 0; k < 8; k++) out_chunk[pos + k] = val;
                    pos += stride;
                }
            } else {
                int32_t tmp[64];
                for (int ptr = 0; ptr < 8; ptr++) { /* columns, scalar.rs:79-167 */
                    int32_t s[8], o[8];
                    for (int k = 0; k < 8; k++) /* dequantize, scalar.rs:308-311 */
                        s[k] = w32_mul((int32_t)v[ptr + 8 * k], qt[ptr + 8 * k]);
                    idct_pass(s, 512, o);
                    for (int k = 0; k < 8; k++) tmp[ptr + 8 * k] = sar32(o[k], 10);
                }
                for (int i = 0; i < 64; i += 8) { /* rows, scalar.rs:170-274 */
                    int32_t o[8];
                    idct_pass(tmp + i, SCALE_BITS, o);
                    if (pos + 8 > chunks) return ZJO_ERR_PANIC; /* get_mut(pos..pos+8).unwrap() */
                    for (int k = 0; k < 8; k++) out_chunk[pos + k] = idct_clamp(sar32(o[k], 17));
                    pos += stride;
                }
            }
            x += 8;   /* scalar.rs:277-278 */
            pos = x;
        }
    }
    return ZJO_OK;
}

/* ===========================================================================================
 * Up-sampling  --  src/upsampler/scalar.rs
 * =========================================================================================== */

/* scalar.rs:5-60 upsample_horizontal: the WHOLE input is one flat 1-D signal. */
int zjo_upsample_h(const int16_t *in, size_t n, int16_t *out, size_t out_len)
{
    if (!(out_len > 4 && n > 2)) return ZJO_ERR_PANIC; /* assert!, scalar.rs:9-12 */
    memset(out, 0, out_len * sizeof(int16_t));
    out[0] = in[0];                                                    /* :13 */
    out[1] = sar16(w16_add(w16_add(w16_mul(in[0], 3), in[1]), 2), 2);  /* :15 */
    /* out[2..].chunks_exact_mut(2).zip(input.windows(3)), scalar.rs:30-42 */
    size_t n_out_pairs = (out_len - 2) / 2;
    size_t n_windows = n - 2;
    size_t m = n_out_pairs < n_windows ? n_out_pairs : n_windows;
    for (size_t w = 0; w < m; w++) {
        int16_t sample = w16_add(w16_mul(3, in[w + 1]), 2);
        out[2 + 2 * w] = sar16(w16_add(sample, in[w]), 2);
        out[2 + 2 * w + 1] = sar16(w16_add(sample, in[w + 2]), 2);
    }
    /* last two, scalar.rs:46-57 */
    int16_t a = in[n - 2], b = in[n - 1];
    out[out_len - 2] = sar16(w16_add(w16_add(w16_mul(3, a), b), 2), 2);
    out[out_len - 1] = b;
    return ZJO_OK;
}

/* scalar.rs:64-147 upsample_vertical: assumes EXACTLY 8 input "rows" of input.len()>>3 samples. */
int zjo_upsample_v(const int16_t *in, size_t n, int16_t *out, size_t out_len)
{
    size_t stride = n >> 3; /* :73 */
    if (stride == 0) return ZJO_ERR_PANIC; /* chunks_exact(0) */
    memset(out, 0, out_len * sizeof(int16_t));
    size_t nrows = n / stride; /* rows the two chunks_exact iterators can yield */
    /* iterator state: index of the next row each iterator will yield */
    size_t near_next = 0, far_next = 0;
    if (nrows == 0) return ZJO_ERR_PANIC;
    const int16_t *rw_n = in + stride * near_next++; /* :84 */
    const int16_t *rw_f = in + stride * far_next++;  /* :86 */
    const int16_t *previous;
    size_t i = 0;
    int next_row = 1;
    for (int it = 0; it < 8; it++) { /* :95 */
        if (i + stride > out_len) return ZJO_ERR_PANIC; /* split_at_mut(stride) on out[i..] */
        int16_t *out_near = out + i;
        int16_t *remainder = out + i + stride;
        size_t rem_len = out_len - i - stride;
        size_t cnt = stride < rem_len ? stride : rem_len; /* zip stops at the shortest */
        for (size_t k = 0; k < cnt; k++) {
            int16_t nr = rw_n[k], fr = rw_f[k];
            out_near[k] = sar16(w16_add(w16_add(w16_mul(nr, 3), fr), 2), 2);  /* :124 */
            remainder[k] = sar16(w16_add(w16_add(w16_mul(fr, 3), nr), 2), 2); /* :126 */
        }
        i += stride * 2; /* :130 */
        previous = rw_n;
        rw_n = near_next < nrows ? in + stride * near_next++ : previous; /* :134 unwrap_or(previous) */
        rw_f = far_next < nrows ? in + stride * far_next++ : rw_n;       /* :136 unwrap_or(rw_n) */
        if (next_row) {                                                   /* :140-144 */
            rw_f = far_next < nrows ? in + stride * far_next++ : rw_n;
            next_row = 0;
        }
    }
    return ZJO_OK;
}

/* scalar.rs:148-166 upsample_hv = horizontal(vertical(input, 2*len), output_len) */
int zjo_upsample_hv(const int16_t *in, size_t n, int16_t *out, size_t out_len)
{
    int16_t *first = (int16_t *)malloc((n ? n * 2 : 1) * sizeof(int16_t));
    if (!first) return ZJO_ERR_NOMEM;
    int rc = zjo_upsample_v(in, n, first, n * 2);
    if (rc == ZJO_OK) rc = zjo_upsample_h(first, n * 2, out, out_len);
    free(first);
    return rc;
}

/* ===========================================================================================
 * Colour conversion  --  src/color_convert/scalar.rs
 * =========================================================================================== */

/* scalar.rs:7-10 clamp (i16 -> u8) */
static inline uint8_t cc_clamp(int16_t a)
{
    if (a < 0) a = 0;
    if (a > 255) a = 255;
    return (uint8_t)a;
}

static inline void ycc_px(int16_t y, int16_t cb, int16_t cr, uint8_t *r, uint8_t *g, uint8_t *b)
{
    /* scalar.rs:68-76; every op is i16 and wraps */
    cr = w16_sub(cr, 128);
    cb = w16_sub(cb, 128);
    int16_t rr = w16_add(y, sar16(w16_mul(45, cr), 5));
    int16_t gg = w16_sub(y, sar16(w16_add(w16_mul(11, cb), w16_mul(23, cr)), 5));
    int16_t bb = w16_add(y, sar16(w16_mul(113, cb), 6));
    *r = cc_clamp(rr);
    *g = cc_clamp(gg);
    *b = cc_clamp(bb);
}

/* scalar.rs:52-89 ycbcr_to_rgb_16_scalar */
int zjo_ycbcr_to_rgb16(const int16_t y[16], const int16_t cb[16], const int16_t cr[16],
                       uint8_t *out, size_t out_len, size_t *pos)
{
    if (*pos > out_len) return ZJO_ERR_PANIC;      /* split_at_mut(*pos), :57 */
    if (out_len - *pos < 48) return ZJO_ERR_PANIC; /* .expect("Slice to small cannot write"), :60-64 */
    uint8_t *o = out + *pos;
    for (int i = 0; i < 16; i++) ycc_px(y[i], cb[i], cr[i], o + 3 * i, o + 3 * i + 1, o + 3 * i + 2);
    *pos += 48; /* :88 */
    return ZJO_OK;
}

/* scalar.rs:14-50 ycbcr_to_rgba_16_scalar (unreachable through the public API) */
int zjo_ycbcr_to_rgba16(const int16_t y[16], const int16_t cb[16], const int16_t cr[16],
                        uint8_t *out, size_t out_len, size_t *pos)
{
    if (*pos > out_len) return ZJO_ERR_PANIC;
    if (out_len - *pos < 64) return ZJO_ERR_PANIC;
    uint8_t *o = out + *pos;
    for (int i = 0; i < 16; i++) {
        ycc_px(y[i], cb[i], cr[i], o + 4 * i, o + 4 * i + 1, o + 4 * i + 2);
        o[4 * i + 3] = 255;
    }
    *pos += 64;
    return ZJO_OK;
}

/* scalar.rs:91-114 ycbcr_to_grayscale: `as u8` truncates (no clamp) */
int zjo_ycbcr_to_grayscale(const int16_t *y, size_t n, size_t width, uint8_t *out, size_t out_len)
{
    if (width == 0) return ZJO_ERR_PANIC;
    size_t width_mcu = n / width; /* :97 */
    if (width_mcu == 0) return ZJO_ERR_PANIC; /* division by zero at :99 */
    size_t width_chunk = n / width_mcu; /* :99 */
    if (width_chunk == 0) return ZJO_ERR_PANIC;
    size_t start = 0, end = width;
    for (size_t c0 = 0; c0 + width_chunk <= n; c0 += width_chunk) { /* chunks_exact, :105 */
        if (end > out_len) return ZJO_ERR_PANIC;  /* output[start..end] */
        if (width > width_chunk) return ZJO_ERR_PANIC; /* chunk[0..width] */
        for (size_t k = 0; k < width; k++) out[start + k] = (uint8_t)(uint16_t)y[c0 + k];
        start += width;
        end += width;
    }
    return ZJO_OK;
}

/* scalar.rs:119-169 ycbcr_to_ycbcr */
int zjo_ycbcr_to_ycbcr(const int16_t *const ch[3], size_t n, size_t width, size_t h_samp,
                       size_t v_samp, uint8_t *out, size_t out_len)
{
    if (h_samp * v_samp == 0) return ZJO_ERR_PANIC;
    size_t mcu_chunks = n / (h_samp * v_samp); /* :125 */
    size_t stride = width * 3;
    size_t start = 0, end = width * 3;
    size_t width_chunk = mcu_chunks >> 3; /* :138 */
    if (width_chunk == 0) return ZJO_ERR_PANIC;
    /* all three channels have n samples here (post up-sampling); zip = shortest */
    for (size_t c0 = 0; c0 + width_chunk <= n; c0 += width_chunk) {
        if (stride > width_chunk * 3) return ZJO_ERR_PANIC; /* temp_output[0..stride] */
        if (end > out_len) return ZJO_ERR_PANIC;
        for (size_t k = 0; k < width; k++) {
            out[start + 3 * k] = (uint8_t)(uint16_t)ch[0][c0 + k];
            out[start + 3 * k + 1] = (uint8_t)(uint16_t)ch[1][c0 + k];
            out[start + 3 * k + 2] = (uint8_t)(uint16_t)ch[2][c0 + k];
        }
        start += stride;
        end += stride;
    }
    return ZJO_OK;
}

/* ===========================================================================================
 * Worker glue  --  src/worker.rs
 * =========================================================================================== */

/* worker.rs:143-251 color_convert_ycbcr.  color_convert_16 is always the RGB routine, even when
 * the output colour space is RGBA/RGBX (decoder.rs:127-128). */
static int color_convert_ycbcr(const int16_t *const blk[3], size_t n, size_t width, size_t h_samp,
                               size_t v_samp, int out_cs, uint8_t *output, size_t out_len, int plain)
{
    size_t ncomp = zjo_num_components(out_cs);
    size_t mcu_chunks = n / (h_samp * v_samp); /* :148 */
    size_t width_chunk = mcu_chunks >> 3;      /* :150 */
    size_t stride = width * ncomp;             /* :151 */
    size_t start = 0, end = stride;
    if (width_chunk == 0) return ZJO_ERR_PANIC;
    if (plain) {
        /* EXTENSION, not reference behaviour (SURVEY 8f-3 "RGBA/RGBX done properly"): the same rows and the
         * same per-pixel arithmetic (scalar.rs:68-76), but every pixel x < width of a row is written at its
         * own position -- no early tail (Q5), no untouched bytes (Q6), 4-byte pixels get 255 as 4th byte. */
        for (size_t c0 = 0; c0 + width_chunk <= n; c0 += width_chunk, start += stride) {
            if (start + stride > out_len) return ZJO_ERR_PANIC;
            for (size_t xx = 0; xx < width && xx < width_chunk; xx++) {
                uint8_t *px = output + start + xx * ncomp;
                ycc_px(blk[0][c0 + xx], blk[1][c0 + xx], blk[2][c0 + xx], px, px + 1, px + 2);
                if (ncomp == 4) px[3] = 255;
            }
        }
        return ZJO_OK;
    }
    uint8_t temp[16 * 4];
    memset(temp, 0, sizeof temp);

    for (size_t c0 = 0; c0 + width_chunk <= n; c0 += width_chunk) { /* :166-169 */
        const int16_t *yw = blk[0] + c0, *cbw = blk[1] + c0, *crw = blk[2] + c0;
        size_t elements = width_chunk / 16;
        elements = elements ? elements - 1 : 0; /* saturating_sub(1), :171 */
        size_t position = 0;
        if (end > out_len) return ZJO_ERR_PANIC; /* &mut output[start..end], :174 */
        uint8_t *out = output + start;
        size_t olen = stride;

        if (width < 16) { /* :176-198 */
            int16_t yo[16] = {0}, cbo[16] = {0}, cro[16] = {0};
            if (width_chunk > 16) return ZJO_ERR_PANIC; /* copy_from_slice length mismatch */
            memcpy(yo, yw, width_chunk * 2);
            memcpy(cbo, cbw, width_chunk * 2);
            memcpy(cro, crw, width_chunk * 2);
            size_t p0 = 0;
            int rc = zjo_ycbcr_to_rgb16(yo, cbo, cro, temp, 16 * ncomp, &p0);
            if (rc) return rc;
            memcpy(out, temp, width * ncomp);
            start += stride;
            end += stride;
            continue;
        }
        size_t ngroups = width_chunk / 16; /* chunks_exact(16) ... .take(elements), :201-214 */
        size_t take = elements < ngroups ? elements : ngroups;
        for (size_t g = 0; g < take; g++) {
            int rc = zjo_ycbcr_to_rgb16(yw + 16 * g, cbw + 16 * g, crw + 16 * g, out, olen, &position);
            if (rc) return rc;
        }
        /* :221-246 tail: last 16 samples, written `diff` bytes early */
        size_t rem = stride > position ? stride - position : 0;
        size_t diff = 64 > rem ? 64 - rem : 0;
        position = position > diff ? position - diff : 0;
        if (width_chunk < 16) return ZJO_ERR_PANIC; /* rchunks_exact(16).next().unwrap() */
        size_t t0 = width_chunk - 16;
        int rc = zjo_ycbcr_to_rgb16(yw + t0, cbw + t0, crw + t0, out, olen, &position);
        if (rc) return rc;
        start += stride;
        end += stride;
    }
    return ZJO_OK;
}

/* worker.rs:32-86 post_process + :88-141 post_process_inner */
static int post_process_impl(const int16_t *const coeff[3], const size_t len[3],
                             const zjo_component comps[3], int in_cs, int out_cs, uint8_t *out,
                             size_t out_len, size_t width, int ext);

int zjo_post_process(const int16_t *const coeff[3], const size_t len[3],
                     const zjo_component comps[3], int in_cs, int out_cs, uint8_t *out,
                     size_t out_len, size_t width)
{
    return post_process_impl(coeff, len, comps, in_cs, out_cs, out, out_len, width, 0);
}

/* EXTENSION helpers (ZJO_EXT_*, no reference behaviour): clamp the unclamped DC-only values of a plane (Q1 corrected,
 * what src/idct/avx2.rs:163-167 does), and a horizontal up-sampler that treats every row on its own with replicated
 * edges instead of one flat array (Q4 corrected). */
static void ext_clamp_dc_only(const int16_t *coeff, size_t n, size_t stride, size_t samp_factors, size_t v_samp, int16_t *plane)
{
    size_t chunks = n * v_samp / samp_factors;
    if (chunks == 0) return;
    for (size_t c0 = 0; c0 + chunks <= n; c0 += chunks) {
        size_t x = 0;
        for (size_t b0 = 0; b0 + 64 <= chunks; b0 += 64, x += 8) {
            const int16_t *v = coeff + c0 + b0;
            int dc_only = 1;
            for (int k = 1; k < 64; k++)
                if (v[k] != 0) { dc_only = 0; break; }
            if (!dc_only) continue;
            for (int r = 0; r < 8; r++)
                for (int k = 0; k < 8; k++) {
                    int16_t *q = plane + c0 + x + (size_t)r * stride + k;
                    if (*q < 0) *q = 0;
                    if (*q > 255) *q = 255;
                }
        }
    }
}
static void ext_upsample_h_rows(const int16_t *in, size_t n, size_t row_len, int16_t *out)
{
    for (size_t r0 = 0; r0 + row_len <= n; r0 += row_len) {
        const int16_t *c = in + r0;
        int16_t *o = out + 2 * r0;
        for (size_t i = 0; i < row_len; i++) {
            int16_t l = c[i ? i - 1 : 0], rr = c[i + 1 < row_len ? i + 1 : row_len - 1];
            o[2 * i] = sar16(w16_add(w16_add(w16_mul(3, c[i]), l), 2), 2);
            o[2 * i + 1] = sar16(w16_add(w16_add(w16_mul(3, c[i]), rr), 2), 2);
        }
    }
}

static int post_process_impl(const int16_t *const coeff[3], const size_t len[3],
                             const zjo_component comps[3], int in_cs, int out_cs, uint8_t *out,
                             size_t out_len, size_t width, int ext)
{
    const int plain = ext & ZJO_EXT_PLAIN;
    size_t h_samp = comps[0].horizontal_sample, v_samp = comps[0].vertical_sample; /* :43-45 */
    size_t nin = zjo_num_components(in_cs), nout = zjo_num_components(out_cs);
    size_t x = nin < nout ? nin : nout; /* :56-59 */
    if (x > 3) x = 3;                   /* unprocessed has 3 slots; (0..x) beyond would panic */
    int16_t *unp[3] = {0, 0, 0};
    size_t ulen[3] = {0, 0, 0};
    int rc = ZJO_OK;

    for (size_t z = 0; z < x && rc == ZJO_OK; z++) { /* :63-81 */
        size_t v_samp_idct = z == 0 ? 1 : v_samp;
        unp[z] = (int16_t *)malloc((len[z] ? len[z] : 1) * sizeof(int16_t));
        if (!unp[z]) { rc = ZJO_ERR_NOMEM; break; }
        ulen[z] = len[z];
        rc = zjo_idct_strip(coeff[z], len[z], comps[z].quantization_table, comps[z].width_stride,
                            h_samp * v_samp, v_samp_idct, unp[z]);
        if (rc == ZJO_OK && (ext & ZJO_EXT_CLAMP_DC))
            ext_clamp_dc_only(coeff[z], len[z], comps[z].width_stride, h_samp * v_samp, v_samp_idct, unp[z]);
    }
    /* post_process_inner, :103-110 */
    if (rc == ZJO_OK && (h_samp != 1 || v_samp != 1)) {
        for (size_t i = 1; i < x && rc == ZJO_OK; i++) {
            size_t olen = ulen[0];
            int16_t *up = (int16_t *)malloc((olen ? olen : 1) * sizeof(int16_t));
            if (!up) { rc = ZJO_ERR_NOMEM; break; }
            /* Decoder::set_upsampling, decoder.rs:478-519 (scalar arms) */
            if ((ext & ZJO_EXT_EDGE_REP) && h_samp == 2 && 2 * ulen[i] * v_samp == olen) {
                /* rows of the chroma plane are width_stride samples long; the vertical pass keeps that row length */
                const size_t row_len = comps[i].width_stride;
                if (v_samp == 1) ext_upsample_h_rows(unp[i], ulen[i], row_len, up);
                else {
                    int16_t *mid = (int16_t *)malloc((olen / 2 ? olen / 2 : 1) * sizeof(int16_t));
                    if (!mid) { free(up); rc = ZJO_ERR_NOMEM; break; }
                    rc = zjo_upsample_v(unp[i], ulen[i], mid, olen / 2); /* the reference's vertical pass, Q3 unchanged */
                    if (rc == ZJO_OK) ext_upsample_h_rows(mid, olen / 2, row_len, up);
                    free(mid);
                }
            }
            else if (h_samp == 2 && v_samp == 1) rc = zjo_upsample_h(unp[i], ulen[i], up, olen);
            else if (h_samp == 1 && v_samp == 2) rc = zjo_upsample_v(unp[i], ulen[i], up, olen);
            else if (h_samp == 2 && v_samp == 2) rc = zjo_upsample_hv(unp[i], ulen[i], up, olen);
            else rc = ZJO_ERR_ARG; /* "Unknown down-sampling method" decoder.rs:513-518 */
            free(unp[i]);
            unp[i] = up;
            ulen[i] = olen;
        }
    }
    if (rc == ZJO_OK) { /* :113-133 */
        int in_y = (in_cs == ZJO_CS_YCBCR || in_cs == ZJO_CS_GRAYSCALE);
        if (in_y && out_cs == ZJO_CS_GRAYSCALE) {
            rc = zjo_ycbcr_to_grayscale(unp[0], ulen[0], width, out, out_len);
        } else if (in_cs == ZJO_CS_YCBCR && out_cs == ZJO_CS_YCBCR) {
            const int16_t *ch[3] = {unp[0], unp[1], unp[2]};
            rc = zjo_ycbcr_to_ycbcr(ch, ulen[0], width, h_samp, v_samp, out, out_len);
        } else if (in_cs == ZJO_CS_YCBCR &&
                   (out_cs == ZJO_CS_RGB || out_cs == ZJO_CS_RGBA || out_cs == ZJO_CS_RGBX)) {
            const int16_t *ch[3] = {unp[0], unp[1], unp[2]};
            rc = color_convert_ycbcr(ch, ulen[0], width, h_samp, v_samp, out_cs, out, out_len, plain);
        } /* else: nothing, :131-132 */
    }
    for (int i = 0; i < 3; i++) free(unp[i]);
    return rc;
}

/* ===========================================================================================
 * Whole-frame driver  --  src/mcu_prog.rs:62-79 (plane sizes), :132-246 (strip loop);
 * geometry src/headers.rs:306-339.
 * =========================================================================================== */

static int frame_geom(const zjo_frame *f, size_t *mcu_x, size_t *mcu_y)
{
    if (!f || f->width == 0 || f->height == 0) return ZJO_ERR_ARG;
    if (!((f->h_max == 1 || f->h_max == 2) && (f->v_max == 1 || f->v_max == 2))) return ZJO_ERR_ARG;
    if (f->in_components != 1 && f->in_components != 3) return ZJO_ERR_ARG;
    if (f->in_components == 1 && (f->h_max != 1 || f->v_max != 1)) return ZJO_ERR_ARG;
    /* headers.rs:317-319; the non-interleaved branch (mcu_prog.rs:66-69) gives the same numbers */
    *mcu_x = (f->width + 8 * f->h_max - 1) / (8 * f->h_max);
    *mcu_y = (f->height + 8 * f->v_max - 1) / (8 * f->v_max);
    return ZJO_OK;
}

size_t zjo_plane_len(const zjo_frame *f, int comp)
{
    size_t mx, my;
    if (frame_geom(f, &mx, &my)) return 0;
    if (comp < 0 || comp >= (int)f->in_components) return 0;
    size_t hs = comp == 0 ? f->h_max : 1, vs = comp == 0 ? f->v_max : 1;
    return mx * 64 * vs * hs * my; /* mcu_prog.rs:76 */
}

size_t zjo_out_len(const zjo_frame *f)
{
    return (size_t)f->width * f->height * zjo_num_components(f->out_colorspace);
}

static int decode_planes_impl(const zjo_frame *f, const int16_t *y, const int16_t *cb, const int16_t *cr,
                              uint8_t *out, int ext);

int zjo_decode_planes(const zjo_frame *f, const int16_t *y, const int16_t *cb, const int16_t *cr,
                      uint8_t *out)
{
    return decode_planes_impl(f, y, cb, cr, out, 0);
}

/* EXTENSION (see color_convert_ycbcr): RGB / RGBA / RGBX with every pixel at its own position */
int zjo_decode_planes_plain(const zjo_frame *f, const int16_t *y, const int16_t *cb, const int16_t *cr,
                            uint8_t *out)
{
    return decode_planes_impl(f, y, cb, cr, out, ZJO_EXT_PLAIN);
}

/* EXTENSION: any combination of ZJO_EXT_PLAIN / ZJO_EXT_CLAMP_DC / ZJO_EXT_EDGE_REP */
int zjo_decode_planes_ext(const zjo_frame *f, int ext, const int16_t *y, const int16_t *cb, const int16_t *cr,
                          uint8_t *out)
{
    return decode_planes_impl(f, y, cb, cr, out, ext);
}

static int decode_planes_impl(const zjo_frame *f, const int16_t *y, const int16_t *cb, const int16_t *cr,
                              uint8_t *out, int ext)
{
    size_t mcu_x, mcu_y;
    int rc = frame_geom(f, &mcu_x, &mcu_y);
    if (rc) return rc;
    int out_cs = f->out_colorspace;
    int in_cs = f->in_components == 3 ? ZJO_CS_YCBCR : ZJO_CS_GRAYSCALE;
    size_t ncomp = zjo_num_components(out_cs);
    if (ncomp == 0) return ZJO_ERR_ARG;
    size_t width = f->width, height = f->height;
    int interleaved = (f->h_max != 1 || f->v_max != 1);

    zjo_component comps[3];
    memset(comps, 0, sizeof comps);
    for (int c = 0; c < 3; c++) {
        comps[c].horizontal_sample = c == 0 ? f->h_max : 1;
        comps[c].vertical_sample = c == 0 ? f->v_max : 1;
        comps[c].width_stride = comps[c].horizontal_sample * mcu_x * 8; /* headers.rs:338 */
        memcpy(comps[c].quantization_table, f->qt[c], sizeof(int32_t) * 64);
    }

    size_t mcu_width = mcu_x * 64; /* mcu_prog.rs:71 */
    size_t bias = 1;
    if (f->h_max == 2 && f->v_max == 1) mcu_width *= 2; /* :138-141 */
    if (f->h_max == 2 && f->v_max == 2) bias = 2;       /* :142-144 */

    size_t extra_space = (size_t)interleaved * 128 * height * ncomp;     /* :173 */
    size_t capacity = (width + 8) * (height + 8);                        /* :174 */
    size_t total = capacity * ncomp + extra_space;
    uint8_t *out_vector = (uint8_t *)calloc(total ? total : 1, 1);       /* :176 vec![0_u8; ..] */
    if (!out_vector) return ZJO_ERR_NOMEM;

    size_t chunks_size = width * ncomp * 8 * f->h_max * f->v_max;        /* :188 */
    size_t y_chunk = mcu_width * f->v_max * f->h_max * bias;             /* :191-192 */
    size_t c_chunk = mcu_width * 1 * 1 * bias;                           /* :199-200 */
    size_t y_len = zjo_plane_len(f, 0);
    size_t c_len = f->in_components == 3 ? zjo_plane_len(f, 1) : 0;

    size_t n_strips = y_len / y_chunk;
    if (f->in_components == 3 && c_len / c_chunk < n_strips) n_strips = c_len / c_chunk;
    if (total / chunks_size < n_strips) n_strips = total / chunks_size;  /* zip = shortest */

    for (size_t s = 0; s < n_strips && rc == ZJO_OK; s++) {
        const int16_t *coeff[3] = {y + s * y_chunk, 0, 0};
        size_t len[3] = {y_chunk, 0, 0};
        if (f->in_components == 3) { /* :197-221 */
            coeff[1] = cb + s * c_chunk;
            coeff[2] = cr + s * c_chunk;
            len[1] = len[2] = c_chunk;
        }                             /* else one component, :222-235: post_process(&[y, &[], &[]]) */
        rc = post_process_impl(coeff, len, comps, in_cs, out_cs, out_vector + s * chunks_size,
                               chunks_size, width, ext);
    }
    if (rc == ZJO_OK) memcpy(out, out_vector, width * height * ncomp); /* truncate, :238-242 */
    free(out_vector);
    return rc;
}
