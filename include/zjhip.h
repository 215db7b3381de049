/*
 * zjhip.h -- C ABI of libzjhip.so: zune-jpeg's post-entropy pixel path on MI355X (gfx950).
 *
 * The library replaces, for the `hip` arm of the reference's function-pointer dispatch, exactly
 * the functions that sit behind these reference interfaces (paths relative to the zune-jpeg tree):
 *
 *   IDCTPtr            = fn(&[i16], &Aligned32<[i32;64]>, usize, usize, usize) -> Vec<i16>
 *                        src/decoder.rs:56, chosen by choose_idct_func          src/idct.rs:40
 *   UpSampler          = fn(&[i16], usize) -> Vec<i16>
 *                        src/components.rs:14, chosen in Decoder::set_upsampling src/decoder.rs:468
 *   ColorConvert16Ptr  = fn(&[i16;16], &[i16;16], &[i16;16], &mut [u8], &mut usize)
 *                        src/decoder.rs:47, chosen by choose_ycbcr_to_rgb_convert_func
 *                                                                         src/color_convert.rs:61
 *   post_process(coeff, component_data, idct_func, color_convert_16, in_cs, out_cs, output, width)
 *                        src/worker.rs:32, called per strip from src/mcu.rs:364 and
 *                        src/mcu_prog.rs:214,228
 *
 * plus the frame/batch-level entry points the GPU actually wants (whole-image coefficient planes,
 * the layout src/mcu_prog.rs:73-79 allocates; baseline strips concatenate to the same planes).
 *
 * Conventions: plain C types, caller-owned buffers, int status (0 = ZJ_OK, negative = error), no
 * exceptions or aborts across the boundary.  Where the reference would panic (slice bounds,
 * unwrap, assert!) the call returns ZJ_ERR_PANIC and leaves the output unspecified.  Results are
 * bit-exact with the reference's SCALAR arms (ZuneJpegOptions::set_use_unsafe(false)).
 * Everything runs on the GPU: there is no CPU fallback; without a usable HIP device every
 * entry point that computes returns ZJ_ERR_NO_DEVICE.
 *
 * Thread safety: a zj_ctx owns one HIP stream and scratch buffers; use one ctx per (thread, device).
 */
#ifndef ZJHIP_H
#define ZJHIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ZJ_ABI_VERSION 8

/* libzjhip.so is built with -fvisibility=hidden: the functions declared here are its whole dynamic symbol table */
#if defined(__GNUC__) || defined(__clang__)
#define ZJ_API __attribute__((visibility("default")))
#else
#define ZJ_API
#endif

/* ColorSpace, same order as src/misc.rs:88-106 */
typedef enum zj_colorspace {
    ZJ_CS_RGB = 0,
    ZJ_CS_GRAYSCALE = 1,
    ZJ_CS_YCBCR = 2,
    ZJ_CS_CMYK = 3,
    ZJ_CS_YCCK = 4,
    ZJ_CS_RGBA = 5,
    ZJ_CS_RGBX = 6
} zj_colorspace;

/* Arms of the reference's dispatch (src/idct.rs:40-61, src/upsampler.rs:82-112,
 * src/color_convert.rs:61-107).  SCALAR and AVX2 stay in the host application (the Rust crate);
 * this library implements only the HIP arm and reports ZJ_ERR_BACKEND for the others. */
typedef enum zj_backend { ZJ_BACKEND_SCALAR = 0, ZJ_BACKEND_AVX2 = 1, ZJ_BACKEND_HIP = 2 } zj_backend;

typedef enum zj_status {
    ZJ_OK = 0,
    ZJ_ERR_ARG = -1,         /* null pointer / inconsistent sizes */
    ZJ_ERR_UNSUPPORTED = -2, /* valid for the reference, not implemented by this library */
    ZJ_ERR_HIP = -3,         /* a HIP runtime call failed; see zj_last_error() */
    ZJ_ERR_NOMEM = -4,
    ZJ_ERR_PANIC = -5,       /* the reference would panic on these arguments */
    ZJ_ERR_NO_DEVICE = -6,   /* no usable HIP device / kernels could not be loaded */
    ZJ_ERR_BACKEND = -7,     /* backend other than ZJ_BACKEND_HIP requested */
    /* zj_decoder_* only: the variants of DecodeErrors (src/errors.rs:16-43); text via zj_decoder_error() */
    ZJ_ERR_FORMAT = -20,             /* Format / FormatStatic */
    ZJ_ERR_MAGIC = -21,              /* IllegalMagicBytes */
    ZJ_ERR_HUFFMAN = -22,            /* HuffmanDecode */
    ZJ_ERR_ZERO = -23,               /* ZeroError */
    ZJ_ERR_DQT = -24,                /* DqtError */
    ZJ_ERR_SOS = -25,                /* SosError */
    ZJ_ERR_SOF = -26,                /* SofError */
    ZJ_ERR_UNSUPPORTED_SCHEME = -27, /* Unsupported(UnsupportedSchemes) */
    ZJ_ERR_MCU = -28,                /* MCUError */
    ZJ_ERR_EXHAUSTED = -29,          /* ExhaustedData */
    ZJ_ERR_LARGE_DIM = -30,          /* LargeDimensions */
    /* zj_decode_scan only (not an error): the device met something in the scan that only the CPU walker treats the way
     * the reference does (a damaged stream, the reference's early exit before the last row loop): decode on the CPU */
    ZJ_RETRY_CPU = 1
} zj_status;

/* The fields of `Components` (src/components.rs:18-43) the pixel path reads. */
typedef struct zj_component {
    size_t horizontal_sample;       /* components.rs:25 */
    size_t vertical_sample;         /* components.rs:23 */
    size_t width_stride;            /* components.rs:40; headers.rs:338 = h_samp * mcu_x * 8 */
    int32_t quantization_table[64]; /* components.rs:33, natural order, values 0..255 */
} zj_component;

/* One frame of whole-image coefficient planes (src/mcu_prog.rs:62-79). */
typedef struct zj_frame_desc {
    uint32_t width, height;  /* pixels */
    uint32_t h_max, v_max;   /* luma sampling factors: (1,1) (2,1) (1,2) (2,2); chroma is (1,1) */
    uint32_t in_components;  /* 1 (grayscale JPEG) or 3 (YCbCr) */
    int32_t out_colorspace;  /* ZJ_CS_RGB, ZJ_CS_GRAYSCALE or ZJ_CS_YCBCR; extension: ZJ_CS_RGBA / ZJ_CS_RGBX */
    int32_t qt[3][64];       /* per component, natural order, values 0..255 (8-bit DQT only,
                                src/headers.rs:154-174) */
    uint32_t flags;          /* 0 = the reference's bytes exactly; ZJ_FLAG_* below */
    uint32_t out_layout;     /* ZJ_LAYOUT_HWC (0, the reference's interleaved bytes) or ZJ_LAYOUT_CHW */
    uint32_t out_pitch;      /* bytes between the starts of consecutive output rows (CHW: of a plane's rows); 0 = tight,
                                width x components: the reference's layout (src/mcu.rs:375-379).  Device outputs only
                                (zj_decode_planes_device*, zj_decode_frames_device, zj_multi_decode_frames_device). */
} zj_frame_desc;

/* Extensions beyond the reference, all off by default.  They keep the reference's strips and its
 * per-pixel arithmetic (Q3 and Q7 always hold) and have no reference output; the checker is the oracle's
 * zjo_decode_planes_ext:
 *   ZJ_FLAG_PLAIN_TAIL      (RGB) every pixel x < width of a converted row is written at 3x: the last 16 samples
 *                           are not written 16 bytes early (Q5) and no byte of the row stays 0 (Q6);
 *   ZJ_FLAG_CLAMP_DC        the DC-only shortcut value is clamped to 0..255 (Q1 corrected; the reference's AVX2 arm
 *                           does this, src/idct/avx2.rs:163-167, its scalar arm does not);
 *   ZJ_FLAG_EDGE_REPLICATE  (h2v1 / h2v2) the horizontal chroma filter runs row by row with replicated edges
 *                           instead of over the strip as one flat array (Q4 corrected);
 *   ZJ_FLAG_CORRECTED       all three: the "non-quirk" mode.  What stays: the vertical schedule (Q3) and the
 *                           dropped odd MCU row are properties of the reference's strip geometry (a corrected
 *                           vertical filter needs taps across strips);
 *   ZJ_CS_RGBA / RGBX       4 bytes per pixel, R G B 255 (the reference's own RGBA arm is malformed),
 *                           plain placement;
 *   ZJ_LAYOUT_CHW           (RGB) three u8 planes of width*height bytes per frame, the tensor layout ML consumers
 *                           want, plain placement.  GRAYSCALE is accepted (one plane: identical to HWC);
 *   out_pitch               rows laid out wider than they are, for outputs that stay in HBM: >= the row's bytes, a
 *                           multiple of 16 when width % 16 == 0, at most 1 MiB; zj_out_len() = out_pitch x height
 *                           (x 3 for CHW).  The bytes between a row's end and the next row's start are never
 *                           written.  A tile's row segment then starts on a cache-line boundary whenever the pitch is
 *                           a multiple of 128: frames whose tight pitch is not (any RGB width that is not a multiple
 *                           of 128 pixels) decode 10-20 % faster into such a pitch.  Host-output
 *                           entry points return ZJ_ERR_UNSUPPORTED for a padded pitch. */
#define ZJ_FLAG_PLAIN_TAIL 1u
#define ZJ_FLAG_CLAMP_DC 2u
#define ZJ_FLAG_EDGE_REPLICATE 4u
#define ZJ_FLAG_CORRECTED 7u
/* zj_options.flags only (the CPU front-end; never passed on to a zj_frame_desc): AC values as the file codes them.  The
 * reference packs a fast-AC value as `k << 10` into an i16 (src/huffman.rs:251) and reads it back with `>> 10`
 * (src/bitstream.rs:343,466): a value of size 6..8 whose code and magnitude fit its 9-bit look-ahead (a code of 1..3 bits --
 * optimised tables, progressive files) keeps six bits, sign-extended: +104 decodes as -24.  The default reproduces that
 * (the reference's own test-images/test-progressive.jpg: 958 coefficients in 649 of 97 200 blocks); with this flag the
 * front-end yields the coded values, which is what libjpeg decodes. */
#define ZJ_FLAG_FULL_AC_VALUES 8u
#define ZJ_LAYOUT_HWC 0u
#define ZJ_LAYOUT_CHW 1u

typedef struct zj_ctx zj_ctx;

/* ---- lifecycle ------------------------------------------------------------------------------ */
ZJ_API int zj_abi_version(void);
ZJ_API int zj_device_count(void);                              /* <0: zj_status */
ZJ_API zj_ctx *zj_ctx_create(int backend, int device, int *status);
ZJ_API void zj_ctx_destroy(zj_ctx *ctx);
ZJ_API zj_ctx *zj_default_ctx(void);        /* the calling thread's own lazily created ctx on device 0, for the fn-pointer
                                        shims (the reference calls them from several worker threads at once);
                                        recycled to the next new thread when its thread ends; never destroy it */
ZJ_API const char *zj_strerror(int status);
ZJ_API const char *zj_last_error(const zj_ctx *ctx);           /* detail of the last ZJ_ERR_HIP */

/* ---- strip level: pointer-compatible with the reference's fn types (host buffers) ----------- */
/* IDCTPtr, src/idct/scalar.rs:19 dequantize_and_idct_int(vector, qt_table, stride, samp_factors, v_samp) */
ZJ_API int zj_idct_strip(zj_ctx *ctx, const int16_t *coeff, size_t n, const int32_t qt[64], size_t stride,
                  size_t samp_factors, size_t v_samp, int16_t *out /* n */);
/* UpSampler, src/upsampler/scalar.rs:5 / :64 / :148 (input, output_len) */
ZJ_API int zj_upsample_h(zj_ctx *ctx, const int16_t *in, size_t n, int16_t *out, size_t out_len);
ZJ_API int zj_upsample_v(zj_ctx *ctx, const int16_t *in, size_t n, int16_t *out, size_t out_len);
ZJ_API int zj_upsample_hv(zj_ctx *ctx, const int16_t *in, size_t n, int16_t *out, size_t out_len);
/* ColorConvert16Ptr, src/color_convert/scalar.rs:52 ycbcr_to_rgb_16_scalar(y, cb, cr, output, pos) */
ZJ_API int zj_ycbcr_to_rgb16(zj_ctx *ctx, const int16_t y[16], const int16_t cb[16], const int16_t cr[16],
                      uint8_t *out, size_t out_len, size_t *pos);
/* post_process, src/worker.rs:32.  `out` must be zero-filled by the caller exactly as the
 * reference's callers do (src/mcu.rs:222, src/mcu_prog.rs:176): bytes the reference never writes
 * are left untouched. */
ZJ_API int zj_post_process_strip(zj_ctx *ctx, const int16_t *const coeff[3], const size_t len[3],
                          const zj_component comps[3], int in_cs, int out_cs, uint8_t *out,
                          size_t out_len, size_t width);

/* Function-pointer dispatch mirror of choose_idct_func & friends: returns the HIP arm. */
typedef int (*zj_idct_fn)(zj_ctx *, const int16_t *, size_t, const int32_t *, size_t, size_t, size_t,
                          int16_t *);
typedef int (*zj_upsample_fn)(zj_ctx *, const int16_t *, size_t, int16_t *, size_t);
typedef int (*zj_color_convert16_fn)(zj_ctx *, const int16_t *, const int16_t *, const int16_t *,
                                     uint8_t *, size_t, size_t *);
ZJ_API zj_idct_fn zj_choose_idct_func(int backend);                        /* src/idct.rs:40 */
ZJ_API zj_upsample_fn zj_choose_upsample_func(int backend, int h_max, int v_max); /* decoder.rs:468 */
ZJ_API zj_color_convert16_fn zj_choose_ycbcr_to_rgb_convert_func(int backend, int out_cs); /* color_convert.rs:61 */

/* ---- frame / batch level -------------------------------------------------------------------- */
ZJ_API size_t zj_plane_len(const zj_frame_desc *d, int comp); /* int16 elements; mcu_prog.rs:76 */
ZJ_API size_t zj_out_len(const zj_frame_desc *d);             /* bytes = width*height*ncomp */
ZJ_API int zj_num_components(int colorspace);                 /* misc.rs:113 */

/* Host buffers; H2D copy, fused kernel(s), D2H copy, synchronous.  Output is fully written
 * (including the bytes the reference leaves at their initial 0). */
ZJ_API int zj_decode_planes(zj_ctx *ctx, const zj_frame_desc *d, const int16_t *y, const int16_t *cb,
                     const int16_t *cr, uint8_t *out);
/* nframes frames of identical geometry, planes and outputs contiguous frame after frame. */
ZJ_API int zj_decode_planes_batch(zj_ctx *ctx, const zj_frame_desc *d, size_t nframes, const int16_t *y,
                           const int16_t *cb, const int16_t *cr, uint8_t *out);
/* The same for frames that are INDEPENDENT allocations, the way the reference's callers own them (a fresh Vec per strip,
 * src/mcu.rs:238-250; one Vec<u8> per decode, src/decoder.rs:178): y[f] / cb[f] / cr[f] / out[f] are frame f's host
 * buffers (cb / cr may be NULL for ZJ_CS_GRAYSCALE output).  Same pipeline, one copy per frame and plane. */
ZJ_API int zj_decode_frames(zj_ctx *ctx, const zj_frame_desc *d, size_t nframes, const int16_t *const *y,
                     const int16_t *const *cb, const int16_t *const *cr, uint8_t *const *out);
/* Device-resident variant (kernel-only path used by bench.py): all pointers are device pointers
 * on ctx's device, 16-byte aligned (the pixels of a frame whose width is not a multiple of 16 may start at any byte:
 * its rows do anyway); frames contiguous.  Asynchronous on `stream` (a hipStream_t;
 * NULL = the ctx stream).  The frame's quantisation tables travel by value in the kernel arguments:
 * nothing is uploaded, cached or ordered against other streams. */
ZJ_API int zj_decode_planes_device(zj_ctx *ctx, const zj_frame_desc *d, size_t nframes,
                            const int16_t *d_y, const int16_t *d_cb, const int16_t *d_cr,
                            uint8_t *d_out, void *stream);
/* Frames at a uniform distance instead of back to back (the images of one tensor with padding, the slots of an arena):
 * y_stride / c_stride in int16 elements (multiples of 8), out_stride in bytes (a multiple of 16); 0 = packed.  ONE launch. */
ZJ_API int zj_decode_planes_device_strided(zj_ctx *ctx, const zj_frame_desc *d, size_t nframes, const int16_t *d_y,
                                    const int16_t *d_cb, const int16_t *d_cr, uint8_t *d_out, size_t y_stride,
                                    size_t c_stride, size_t out_stride, void *stream);
/* Scattered batch: nframes frames of ONE geometry whose planes and pixels are independent device allocations, in any
 * order.  d_y / d_cb / d_cr / d_out are HOST arrays of nframes device pointers (each 16-byte aligned; d_cb / d_cr may be
 * NULL for ZJ_CS_GRAYSCALE output); they are read during the call -- the addresses travel in the kernel arguments, up to
 * ZJ_SCATTER_MAX frames per launch, nothing is staged and nothing has to outlive the call.  Frames that turn out to be
 * equally spaced share one launch whatever their number.  A caller with two or more frames in hand should use this (or a
 * contiguous batch) instead of one launch per frame: a launch of one 4096x4096 frame is 1.3 waves of workgroups and runs
 * at 0.49-0.54 of the HBM peak, a launch of 16 at 0.70.  Asynchronous on `stream`. */
#define ZJ_SCATTER_MAX 32
ZJ_API int zj_decode_frames_device(zj_ctx *ctx, const zj_frame_desc *d, size_t nframes, const int16_t *const *d_y,
                            const int16_t *const *d_cb, const int16_t *const *d_cr, uint8_t *const *d_out, void *stream);
/* Times zj_decode_planes_device with HIP events recorded on the launch stream: *ms_total = `iters`
 * back-to-back launches between one event pair; *ms_each (optional) = mean over `iters` launches
 * each bracketed by its own event pair; *kernel_name = the dominant kernel. */
ZJ_API int zj_time_decode_device(zj_ctx *ctx, const zj_frame_desc *d, size_t nframes, const int16_t *d_y,
                          const int16_t *d_cb, const int16_t *d_cr, uint8_t *d_out, void *stream,
                          int iters, float *ms_total, float *ms_each, const char **kernel_name);

/* ---- ONE frame whose planes are still being written (round 6): strips go to the GPU as they become final ----------------
 * The reference runs post_process on strip N while its Huffman decoder is in strip N + 1 (src/mcu.rs:356-368: the strip is
 * handed to a pool thread).  zj_frame_begin names the frame's planes and output; zj_frame_rows_ready(ctx, n) says that the
 * coefficients of MCU rows [0, n) are final (n only grows; call it as often as convenient); the strips that became complete
 * go through the three-stream pipeline in units of an eighth of the frame (at least 4 MB of coefficients, ZJ_STREAM_UNIT_MB)
 * while the caller fills later rows; zj_frame_end submits what is left and waits for the pixels.  All calls but
 * zj_frame_end return at once when the planes are pinned (zj_alloc_pinned).  Output: a device pointer (out_on_device), pinned
 * host memory (downloads overlap as well), or pageable host memory (decoded into a device buffer, copied once at the end).
 * One such frame at a time per context, and no other call on the context in between.  Same bytes as zj_decode_planes.
 * ZJ_ERR_UNSUPPORTED: ZJ_LAYOUT_CHW, a padded out_pitch with a host output.  zj_decoder_decode_buffer uses this for baseline
 * files when the decoder's planes are pinned (zj_options.pinned_planes); ZJ_STREAM=off keeps the two stages apart. */
ZJ_API int zj_frame_begin(zj_ctx *ctx, const zj_frame_desc *d, const int16_t *y, const int16_t *cb, const int16_t *cr,
                          uint8_t *out, int out_on_device);
ZJ_API int zj_frame_rows_ready(zj_ctx *ctx, size_t mcu_rows);
ZJ_API int zj_frame_end(zj_ctx *ctx);
ZJ_API int zj_frame_abort(zj_ctx *ctx);   /* drains what is in flight; the output's content is undefined */
/* Host planes -> pixels that stay in HBM (d_out: device pointer, 16-byte aligned). Synchronous. */
ZJ_API int zj_decode_planes_to_device(zj_ctx *ctx, const zj_frame_desc *d, const int16_t *y, const int16_t *cb,
                               const int16_t *cr, uint8_t *d_out);
/* A prepared baseline scan (zj_decoder_prepare / zj_decoder_scan_blob; host memory, pinned for an asynchronous upload)
 * -> pixels: upload, Huffman decoding on the device into whole-frame planes in HBM, pixel kernel, and -- unless
 * out_on_device -- the download.  Replaces the MCU walk of src/mcu.rs:231-351.  Synchronous.  ZJ_RETRY_CPU with
 * *status_bits (optional) when the device hands the scan back. */
ZJ_API int zj_decode_scan(zj_ctx *ctx, const zj_frame_desc *d, const void *blob, size_t blob_bytes, uint8_t *out,
                   int out_on_device, unsigned *status_bits);
/* The same for up to ZJ_SCAN_BATCH_MAX prepared scans of any geometry at once: every phase of the entropy stage is ONE
 * launch over all of them (a single file leaves most of the GPU idle); the scans of one geometry and one set of tables
 * also share ONE pixel-kernel launch, wherever their outputs lie (zj_decode_frames_device's scattered form).  rcs[k] = ZJ_OK, ZJ_RETRY_CPU
 * or a zj_status of scan k; status_bits[k] optional.  The return value reports failures of the call itself. */
#define ZJ_SCAN_BATCH_MAX 16
ZJ_API int zj_decode_scans(zj_ctx *ctx, size_t n, const zj_frame_desc *descs, const void *const *blobs, const size_t *blob_bytes,
                    uint8_t *const *outs, int outs_on_device, int *rcs, unsigned *status_bits);
/* diagnostics: the coefficient planes the last zj_decode_scan on ctx left in HBM, copied to host buffers of zj_plane_len
 * elements each (NULL: skipped); len[3] (optional) receives the lengths */
ZJ_API int zj_scan_planes(zj_ctx *ctx, int16_t *y, int16_t *cb, int16_t *cr, size_t len[3]);
/* of the last zj_decode_scan on ctx: synchronisation rounds; ms[3] = host milliseconds spent submitting (everything in
 * front of the final synchronisation); with ZJ_HUFF_TIME set in the environment, ms[0..2] = device milliseconds of
 * upload + rounds | prefix sums + write pass | pixel kernel (+ download) */
ZJ_API int zj_scan_stats(const zj_ctx *ctx, int *rounds, float ms[4]);

/* ---- whole decoder: the CPU front-end the path is fed by (container + Huffman on the host) ------
 * Mirrors Decoder / ZuneJpegOptions / ImageInfo (src/decoder.rs:60,178,452,652; src/options.rs:6-40):
 * SOF0 baseline and SOF2 progressive Huffman, 8-bit, 1 or 3 components, DRI/RST.  The entropy decode
 * runs on the CPU into whole-image coefficient planes; zj_decoder_decode_buffer then runs the pixel
 * path on the GPU through zj_decode_planes. */
typedef struct zj_options {      /* zero = reference default */
    int32_t out_colorspace;      /* ZJ_CS_RGB (default 0), ZJ_CS_GRAYSCALE, ZJ_CS_YCBCR */
    int32_t strict_mode;         /* options.rs:38 */
    int32_t max_width, max_height; /* options.rs:34-35, default 16384 */
    int32_t max_scans;           /* options.rs:36, default 64 */
    int32_t num_threads;         /* options.rs:33, default 4.  The reference spends them on post_process strips; here
                                    the pixel path is one GPU launch, so they decode restart segments (DRI/RSTn,
                                    baseline) concurrently, enter a baseline scan WITHOUT restart markers at one point
                                    per thread (at most 16; zj_decoder_parallel_mcus) and clear the planes; the
                                    decoder keeps its helper threads until zj_decoder_free; 1 = strictly serial */
    int32_t pinned_planes;       /* non-zero: coefficient planes live in pinned host memory (DMA without staging) */
    uint32_t flags;              /* ZJ_FLAG_* for the pixel path (0 = the reference's bytes), see zj_frame_desc;
                                    + ZJ_FLAG_FULL_AC_VALUES for the front-end itself */
    uint32_t out_layout;         /* ZJ_LAYOUT_HWC (0) or ZJ_LAYOUT_CHW */
    int32_t entropy;             /* where baseline Huffman scans are decoded: ZJ_ENTROPY_CPU (0), ZJ_ENTROPY_GPU (scans of
                                    32 KB and more and 16 bits per block and more on the device, the rest and everything
                                    the device hands back on the CPU), ZJ_ENTROPY_GPU_ALWAYS (every eligible scan: for
                                    tests).  Progressive files: always CPU.  The output bytes do not depend on it */
} zj_options;
#define ZJ_ENTROPY_CPU 0
#define ZJ_ENTROPY_GPU 1
#define ZJ_ENTROPY_GPU_ALWAYS 2
typedef struct zj_image_info {   /* ImageInfo, src/decoder.rs:652-668 (+ what the GPU path needs) */
    uint16_t width, height;
    uint8_t components, progressive, h_max, v_max;
    uint16_t scans, restart_interval;
} zj_image_info;
typedef struct zj_decoder zj_decoder;
ZJ_API zj_decoder *zj_decoder_new(const zj_options *opt);            /* Decoder::new_with_options */
ZJ_API void zj_decoder_free(zj_decoder *d);
ZJ_API const char *zj_decoder_error(const zj_decoder *d);            /* Display text of the last DecodeErrors */
ZJ_API int zj_decoder_read_headers(zj_decoder *d, const uint8_t *buf, size_t len, zj_image_info *info); /* decoder.rs:452 */
/* CPU half only: planes stay owned by the decoder until the next call (mcu_prog.rs:73-79 layout) */
ZJ_API int zj_decoder_decode_coefficients(zj_decoder *d, const uint8_t *buf, size_t len, zj_frame_desc *desc,
                                   const int16_t **planes /*[3]*/, size_t *plane_len /*[3]*/, zj_image_info *info);
/* Stage 1 on the CPU, whatever the options say it is: with entropy == ZJ_ENTROPY_CPU the same as
 * zj_decoder_decode_coefficients; with a GPU setting, headers + the byte-level preparation of a baseline scan
 * (stuffing removed, restart segments located, sub-sequence grid: csrc/zj_huff.h), leaving the Huffman decoding itself
 * (src/mcu.rs:231-351, src/bitstream.rs:314-373) to zj_decoder_finish_pixels.  `buf` must then stay valid until the
 * pixels are finished (a scan the device hands back is decoded from it on the CPU). */
ZJ_API int zj_decoder_prepare(zj_decoder *d, const uint8_t *buf, size_t len, zj_frame_desc *desc, zj_image_info *info);
/* GPU half: the pixel path over the planes the last zj_decoder_decode_coefficients / zj_decoder_prepare left in the
 * decoder; after a zj_decoder_prepare that left a scan for the device: entropy stage + pixel path on the GPU */
ZJ_API int zj_decoder_finish_pixels(zj_decoder *d, zj_ctx *ctx, uint8_t *out, size_t out_cap, size_t *out_len);
/* the same with the pixels left in HBM (d_out: device pointer on ctx's device, 16-byte aligned) for consumers on the GPU.
 * Stream ordering (also zj_decode_planes_device with stream = NULL, zj_decoder_finish_pixels_batch with device outputs and
 * zj_pool_decode_files_device): the library writes d_out on ITS OWN non-blocking stream and returns once that is complete;
 * it does NOT order itself after work the caller still has in flight on other streams.  A caller whose allocator recycles
 * device memory stream-ordered (a caching tensor allocator) must synchronise the stream that last used d_out before the call. */
ZJ_API int zj_decoder_finish_pixels_device(zj_decoder *d, zj_ctx *ctx, uint8_t *d_out, size_t out_cap, size_t *out_len);
/* stage 2 of n decoders on one context: the scans left for the device are decoded together (zj_decode_scans), the rest
 * one by one; rcs[k] is what zj_decoder_finish_pixels[_device] would have returned for decoder k */
ZJ_API int zj_decoder_finish_pixels_batch(zj_decoder *const *ds, size_t n, zj_ctx *ctx, uint8_t *const *outs,
                                   const size_t *out_caps, size_t *out_lens /*[n] or NULL*/, int outs_on_device, int *rcs);
/* diagnostics: the prepared scan of the last zj_decoder_prepare (ZJ_ERR_ARG: none); the status bits (csrc/zj_huff.h
 * HUFF_ST_*) with which the device handed the last scan back to the CPU (0: it did not) */
ZJ_API int zj_decoder_scan_blob(const zj_decoder *d, const void **blob, size_t *len);
ZJ_API unsigned zj_decoder_gpu_status(const zj_decoder *d);
/* Decoder::decode_buffer (decoder.rs:178): width*height*ncomp bytes into `out` */
ZJ_API int zj_decoder_decode_buffer(zj_decoder *d, zj_ctx *ctx, const uint8_t *buf, size_t len, uint8_t *out,
                             size_t out_cap, size_t *out_len, zj_image_info *info);
/* restart segments the last baseline scan decoded concurrently (0 = the serial walk was used) */
ZJ_API int zj_decoder_parallel_segments(const zj_decoder *d);
/* MCUs of the last baseline scan WITHOUT restart markers that several threads decoded (zj_options.num_threads > 1: the scan is
 * entered at one point per thread, the threads fall into step with the true symbol sequence; zj_jpeg.cpp); 0 = the serial walk */
ZJ_API int64_t zj_decoder_parallel_mcus(const zj_decoder *d);
/* Decoder::set_num_threads (src/decoder.rs:591-603, deprecated there in favour of the options): the thread count from the next
 * file on; 0 -> ZJ_ERR_FORMAT "Cannot set zero threads to decode image".  zj_pool uses it to lend the workers a short batch
 * leaves idle to the files it has (below). */
ZJ_API int zj_decoder_set_num_threads(zj_decoder *d, int threads);

/* ---- batches of files: `threads` persistent host workers, each with its own entropy
 * decoder, pinned coefficient planes and GPU context on `device`; file i is decoded by whichever worker is
 * free, so the CPU Huffman stage of one file overlaps the PCIe copies and kernels of the others.  Replaces a
 * caller-side loop over Decoder::decode_buffer (src/decoder.rs:178; the reference's own pool is per decode,
 * src/mcu.rs:135).  outs[i] must hold out_caps[i] >= width*height*ncomp bytes; statuses[i] (optional) gets
 * the zj_status of file i; the return value is the first error seen (ZJ_OK if none), text via zj_pool_error.
 * A batch of at most half as many files as workers gets the idle workers' share of the CPUs inside its files
 * (threads / files threads per file, at most 16: zj_decoder_set_num_threads), so two large files on a pool of sixteen do
 * not take what one file takes on one thread; opt->num_threads is the floor (default here: 1). */
typedef struct zj_pool zj_pool;
ZJ_API zj_pool *zj_pool_create(int device, int threads, const zj_options *opt, int *status);
/* Image-level sharding across the GPUs of one node, inside the library (north_star): one pool over `ndev`
 * device slots (devices[k] = HIP device of slot k; a device may fill several slots), threads_per_device entropy workers
 * per slot.  Every slot has its own submitters and contexts; workers and pinned plane sets are shared.  Files whose
 * pixels go to host memory are taken by whichever slot is free (dealt by readiness, so a slow GPU does not hold up a
 * fixed share); with zj_pool_decode_files_device each file goes to the slot(s) of the device that owns its output pointer
 * (zj_pointer_device), so a caller shards by allocating output i on device i * ndev / nfiles.  Results are in the caller's
 * order either way.  The 8-GPU node: devices = {0,...,7}. */
ZJ_API zj_pool *zj_pool_create_multi(const int *devices, int ndev, int threads_per_device, const zj_options *opt, int *status);
ZJ_API int zj_pool_devices(const zj_pool *pool);              /* device slots */
/* of one slot, accumulated since creation: its HIP device, seconds inside its GPU stage, files it finished */
ZJ_API int zj_pool_device_stats(zj_pool *pool, int slot, int *device, double *gpu_seconds, size_t *files);
/* of one slot: NUMA node of its device (-1 unknown), how many of its threads (entropy workers + submitters) were bound to
 * that node, how many it has.  Each slot has its own workers and plane sets; a file's planes are filled, pinned and DMA'd
 * on the socket of the GPU that decodes it (a slot with nothing to do may take over a file of another slot). */
ZJ_API int zj_pool_slot_numa(zj_pool *pool, int slot, int *device_node, int *threads_bound, int *threads);
ZJ_API void zj_pool_destroy(zj_pool *pool);
ZJ_API int zj_pool_threads(const zj_pool *pool);
ZJ_API const char *zj_pool_error(const zj_pool *pool);
/* accumulated since creation: seconds spent inside the entropy stage and inside the GPU stage (summed over the
 * threads of each stage) and files that reached the GPU stage */
ZJ_API int zj_pool_stats(zj_pool *pool, double *entropy_seconds, double *gpu_seconds, size_t *files);
ZJ_API int zj_pool_decode_files(zj_pool *pool, size_t nfiles, const uint8_t *const *bufs, const size_t *lens,
                         uint8_t *const *outs, const size_t *out_caps, size_t *out_lens /*[nfiles] or NULL*/,
                         zj_image_info *infos /*[nfiles] or NULL*/, int *statuses /*[nfiles] or NULL*/);

/* the same with device pointers (on the pool's device, 16-byte aligned) as outputs: the pixels stay in HBM for a
 * consumer on the GPU, nothing crosses PCIe but the compressed files (with a GPU entropy setting) or the planes */
ZJ_API int zj_pool_decode_files_device(zj_pool *pool, size_t nfiles, const uint8_t *const *bufs, const size_t *lens,
                                uint8_t *const *d_outs, const size_t *out_caps, size_t *out_lens /*[nfiles] or NULL*/,
                                zj_image_info *infos /*[nfiles] or NULL*/, int *statuses /*[nfiles] or NULL*/);

/* ---- image-level sharding of plane batches across the GPUs of one node ------------------------------
 * Frames are independent, so N frames over D device slots are D contiguous shards (zj_shard_range: sizes differ by at
 * most one; the same rule as bench.py's ranks) and no collective.  A zj_multi owns one context and one persistent host
 * thread per slot ("one host thread per GPU"); a call runs every slot's shard at once through that slot's three-stream
 * pipeline and returns when all are done.  statuses[slot] (optional) = that shard's zj_status; the return value is the
 * first error.  devices[k] may repeat (two slots on one GPU).  The 8-GPU node: devices = {0,...,7}. */
typedef struct zj_multi zj_multi;
ZJ_API void zj_shard_range(size_t nframes, int slot, int nslots, size_t *lo, size_t *hi); /* [lo, hi) of slot */
ZJ_API zj_multi *zj_multi_create(const int *devices, int ndev, int *status);
ZJ_API void zj_multi_destroy(zj_multi *m);
ZJ_API int zj_multi_devices(const zj_multi *m);
ZJ_API zj_ctx *zj_multi_ctx(zj_multi *m, int slot);            /* slot's context (owned by m), e.g. for zj_device_alloc */
ZJ_API int zj_multi_slot_stats(zj_multi *m, int slot, int *device, size_t *frames); /* frames decoded since creation */
/* where slot's device and its host thread live: NUMA node of the device (-1 unknown), node the thread runs on (-1 unknown),
 * whether the thread was bound (0 with ZJ_NUMA=off, an unknown node, or an affinity that excludes the node) */
ZJ_API int zj_multi_slot_numa(zj_multi *m, int slot, int *device_node, int *thread_node, int *bound);
/* zj_decode_planes_batch / zj_decode_frames, sharded: host planes (pinned: zj_alloc_pinned) -> host pixels */
ZJ_API int zj_multi_decode_planes_batch(zj_multi *m, const zj_frame_desc *d, size_t nframes, const int16_t *y,
                                 const int16_t *cb, const int16_t *cr, uint8_t *out, int *statuses /*[slots] or NULL*/);
ZJ_API int zj_multi_decode_frames(zj_multi *m, const zj_frame_desc *d, size_t nframes, const int16_t *const *y,
                           const int16_t *const *cb, const int16_t *const *cr, uint8_t *const *out, int *statuses);
/* zj_decode_frames_device, sharded: frame f's pointers must live on the device of the slot whose shard holds f;
 * returns once every shard is complete */
ZJ_API int zj_multi_decode_frames_device(zj_multi *m, const zj_frame_desc *d, size_t nframes, const int16_t *const *d_y,
                                  const int16_t *const *d_cb, const int16_t *const *d_cr, uint8_t *const *d_out,
                                  int *statuses);

/* ---- memory helpers ------------------------------------------------------------------------- */
/* hipHostMalloc, portable (usable as a DMA source/target from every device); NULL on failure.  zj_free_pinned keeps up to
 * ZJ_PINNED_CACHE_MB (default 512, 0 = none) of freed blocks for later requests of at least half their size from a thread on the
 * same device: pinning costs milliseconds per 10 MB, and a decoder made per file (zj_options.pinned_planes) would pay it per file */
ZJ_API void *zj_alloc_pinned(size_t bytes);
/* binds the CALLING thread to `device` (hipSetDevice): host threads that only allocate pinned memory or fill planes for a
 * context on device N call this first, so they neither initialise nor pin against device 0 */
ZJ_API int zj_set_thread_device(int device);
/* ---- NUMA placement of a device's host side (round 6).  On a two-socket node a GPU's feeder threads and the
 * pinned planes they fill belong on the GPU's socket.  hipHostMalloc already puts pinned memory on the node of the thread's
 * CURRENT device (so call zj_set_thread_device before zj_alloc_pinned); these bind the threads.  zj_pool and zj_multi bind
 * their own slot threads; a process-per-GPU caller (bench.py's ranks) calls zj_bind_thread_near_device first thing, so every
 * thread it starts later inherits the placement.  ZJ_NUMA=off turns all binding off.  A CPU affinity the process was
 * started with (taskset, cpuset) is respected: threads are bound to its intersection with the node, or left alone. */
ZJ_API int zj_device_pci_bus_id(int device, char *buf, size_t cap);  /* "0000:5a:00.0" */
ZJ_API int zj_device_numa_node(int device);          /* >= 0, or -1: unknown / single node */
ZJ_API int zj_bind_thread_to_numa_node(int node);    /* CPUs the calling thread may now run on, or -1: not bound */
ZJ_API int zj_bind_thread_near_device(int device);   /* the node the calling thread is now bound to, or -1: not bound */
ZJ_API int zj_thread_numa_node(void);                /* node of the CPU the calling thread is running on, or -1 */
/* the HIP device that owns device pointer p (>= 0), ZJ_ERR_ARG for host memory or an unknown pointer */
ZJ_API int zj_pointer_device(const void *p);
ZJ_API void zj_free_pinned(void *p);
ZJ_API void *zj_device_alloc(zj_ctx *ctx, size_t bytes);
ZJ_API void zj_device_free(zj_ctx *ctx, void *p);
ZJ_API int zj_memcpy_h2d(zj_ctx *ctx, void *dst, const void *src, size_t bytes);
ZJ_API int zj_memcpy_d2h(zj_ctx *ctx, void *dst, const void *src, size_t bytes);
ZJ_API int zj_device_memset(zj_ctx *ctx, void *d_ptr, int value, size_t bytes);
ZJ_API int zj_sync(zj_ctx *ctx);

/* ---- tuning knobs (every setting produces the same bytes) ------------------------------------ */
/* generation of the fused kernel: 0 = packed IDCT + staged stores (default), 1 = wide (24-bit multiplies, per-lane
 * stores; only in a `make VARIANTS=all` build), 2 = packed with direct stores.  Also settable per process with ZJ_VARIANT. */
ZJ_API int zj_set_variant(zj_ctx *ctx, int variant);
/* 1 if this build of the library carries the variant.  The default build is the product: 0 and 2.  Variant 1 (round 1's
 * generation, the A/B baseline and the parity suite's N-version cross-check) comes with `make VARIANTS=all`; without it
 * zj_set_variant(ctx, 1) returns ZJ_ERR_UNSUPPORTED. */
ZJ_API int zj_variant_available(int variant);
/* 0 = zj_decode_planes_batch runs upload / kernel / download back to back instead of overlapped on three streams */
ZJ_API int zj_set_pipeline(zj_ctx *ctx, int on);

#ifdef __cplusplus
}
#endif
#endif
