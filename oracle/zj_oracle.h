/*
 * zj_oracle.h -- CPU restatement (the ORACLE) of zune-jpeg's scalar post-entropy pixel path.
 *
 * THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only tests/, __graft_entry__.smoke() and the
 * cpu_baseline leg of bench.py may load it.  The product library (libzjhip.so) never links it.
 *
 * Parity status: the reference is Rust (no rustc/cargo in this image, deps not vendored), so it
 * cannot be compiled or imported here.  The IDCT restatement is PINNED by the reference's own three
 * known-answer tests (src/idct.rs:66-127).  Upsampling, colour conversion and the worker glue are
 * "PARITY UNPINNED" by the reference (it ships no value-level vectors for them): for
 * those the scalar source text is the specification, restated twice independently (this file and
 * oracle/oracle_np.py) and cross-checked bit-for-bit in tests/.
 *
 * Every function cites the reference file:line it follows (paths relative to /root/reference).
 * Rust panics (slice bounds, unwrap, assert!) are reported as ZJO_ERR_PANIC instead of aborting.
 */
#ifndef ZJ_ORACLE_H
#define ZJ_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ColorSpace, same order as src/misc.rs:88-106 */
enum {
    ZJO_CS_RGB = 0,
    ZJO_CS_GRAYSCALE = 1,
    ZJO_CS_YCBCR = 2,
    ZJO_CS_CMYK = 3,
    ZJO_CS_YCCK = 4,
    ZJO_CS_RGBA = 5,
    ZJO_CS_RGBX = 6
};

enum {
    ZJO_OK = 0,
    ZJO_ERR_PANIC = -1, /* the reference would panic (bounds / unwrap / assert) */
    ZJO_ERR_ARG = -2,
    ZJO_ERR_NOMEM = -3
};

/* The fields of `Components` (src/components.rs:18-43) that the pixel path reads. */
typedef struct zjo_component {
    size_t horizontal_sample;
    size_t vertical_sample;
    size_t width_stride; /* headers.rs:338: horizontal_sample * mcu_x * 8 */
    int32_t quantization_table[64]; /* natural order */
} zjo_component;

/* Frame geometry, derived the way src/headers.rs:306-339 does. */
typedef struct zjo_frame {
    uint32_t width, height;
    uint32_t h_max, v_max;      /* luma sampling factors: (1,1) (2,1) (1,2) (2,2) */
    uint32_t in_components;     /* 1 (grayscale JPEG) or 3 (YCbCr) */
    int32_t out_colorspace;     /* ZJO_CS_RGB / GRAYSCALE / YCBCR */
    int32_t qt[3][64];          /* natural order */
} zjo_frame;

size_t zjo_num_components(int colorspace); /* misc.rs:113-121 */

/* src/idct/scalar.rs:19-282  dequantize_and_idct_int */
int zjo_idct_strip(const int16_t *coeff, size_t n, const int32_t qt[64], size_t stride,
                   size_t samp_factors, size_t v_samp, int16_t *out /* n */);

/* src/upsampler/scalar.rs:5-60, 64-147, 148-166 */
int zjo_upsample_h(const int16_t *in, size_t n, int16_t *out, size_t out_len);
int zjo_upsample_v(const int16_t *in, size_t n, int16_t *out, size_t out_len);
int zjo_upsample_hv(const int16_t *in, size_t n, int16_t *out, size_t out_len);

/* src/color_convert/scalar.rs:52-89, 14-50, 91-114, 119-169 */
int zjo_ycbcr_to_rgb16(const int16_t y[16], const int16_t cb[16], const int16_t cr[16],
                       uint8_t *out, size_t out_len, size_t *pos);
int zjo_ycbcr_to_rgba16(const int16_t y[16], const int16_t cb[16], const int16_t cr[16],
                        uint8_t *out, size_t out_len, size_t *pos);
int zjo_ycbcr_to_grayscale(const int16_t *y, size_t n, size_t width, uint8_t *out, size_t out_len);
int zjo_ycbcr_to_ycbcr(const int16_t *const ch[3], size_t n, size_t width, size_t h_samp,
                       size_t v_samp, uint8_t *out, size_t out_len);

/* src/worker.rs:32-86 post_process (+ :88 post_process_inner, :143 color_convert_ycbcr).
 * The up-sampler is chosen like Decoder::set_upsampling (src/decoder.rs:468-523) from
 * comps[0].{horizontal,vertical}_sample, scalar arms only. */
int zjo_post_process(const int16_t *const coeff[3], const size_t len[3],
                     const zjo_component comps[3], int in_cs, int out_cs, uint8_t *out,
                     size_t out_len, size_t width);

/* Whole-frame driver: cuts whole-image coefficient planes into strips and calls post_process per
 * strip exactly like finish_progressive_decoding (src/mcu_prog.rs:132-246); baseline strips
 * (src/mcu.rs:222-368) concatenate to the same planes.  `out` receives width*height*ncomp bytes
 * (the truncate at mcu_prog.rs:238 / mcu.rs:375). Plane lengths: zjo_plane_len(). */
size_t zjo_plane_len(const zjo_frame *f, int comp);
size_t zjo_out_len(const zjo_frame *f);
int zjo_decode_planes(const zjo_frame *f, const int16_t *y, const int16_t *cb, const int16_t *cr,
                      uint8_t *out);
/* EXTENSION beyond the reference: same strips, same per-pixel arithmetic, but every pixel
 * x < width of every converted row is written at its own position (no early RGB tail Q5, no untouched
 * bytes Q6); out_colorspace RGB (3 B/px) or RGBA / RGBX (4 B/px, 4th byte 255).  Checker for
 * ZJ_FLAG_PLAIN_TAIL / ZJ_CS_RGBA / ZJ_LAYOUT_CHW of include/zjhip.h; no reference output exists for it. */
int zjo_decode_planes_plain(const zjo_frame *f, const int16_t *y, const int16_t *cb, const int16_t *cr,
                            uint8_t *out);
/* The full set of "corrected mode" extensions (checker for zj_frame_desc.flags): PLAIN as above; CLAMP_DC clamps the
 * DC-only shortcut value to 0..255 (Q1; what the reference's AVX2 arm does, src/idct/avx2.rs:163-167); EDGE_REP runs
 * the horizontal chroma filter row by row with replicated edges instead of over one flat array (Q4).  The vertical
 * schedule (Q3) and dropped odd MCU rows are strip-geometry properties and stay. */
#define ZJO_EXT_PLAIN 1
#define ZJO_EXT_CLAMP_DC 2
#define ZJO_EXT_EDGE_REP 4
int zjo_decode_planes_ext(const zjo_frame *f, int ext, const int16_t *y, const int16_t *cb, const int16_t *cr,
                          uint8_t *out);

#ifdef __cplusplus
}
#endif
#endif
