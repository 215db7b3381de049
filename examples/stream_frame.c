/* stream_frame.c -- one frame whose coefficient planes are still being written: its strips go to the GPU as rows become final.
 *
 *   gcc -std=c99 -I include examples/stream_frame.c -L zune-jpeg_amd -lzjhip -Wl,-rpath,$PWD/zune-jpeg_amd -o stream_frame
 *   ./stream_frame [width height]          (default 4096 x 4096, 4:2:0 -> RGB)
 *
 * The reference hands strip N to a pool thread while its Huffman decoder is in strip N + 1 (src/mcu.rs:356-368).  The same
 * overlap across the host / device boundary: the caller -- here a loop that "decodes" one MCU row at a time into pinned
 * planes -- says how many MCU rows are final (zj_frame_rows_ready), and uploads, kernels and downloads of the strips they
 * complete run behind its back.  The result is compared with zj_decode_planes of the finished planes. */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "zjhip.h"

int main(int argc, char **argv)
{
    const uint32_t w = argc > 2 ? (uint32_t)atoi(argv[1]) : 4096, h = argc > 2 ? (uint32_t)atoi(argv[2]) : 4096;
    int st = ZJ_OK;
    if (zj_device_count() <= 0) { fprintf(stderr, "no HIP device: this library has no CPU fallback\n"); return 1; }
    zj_ctx *ctx = zj_ctx_create(ZJ_BACKEND_HIP, 0, &st);
    if (!ctx) { fprintf(stderr, "zj_ctx_create: %s\n", zj_strerror(st)); return 1; }

    zj_frame_desc d;
    memset(&d, 0, sizeof d);
    d.width = w; d.height = h; d.h_max = 2; d.v_max = 2; d.in_components = 3; d.out_colorspace = ZJ_CS_RGB;
    for (int c = 0; c < 3; c++)
        for (int k = 0; k < 64; k++) d.qt[c][k] = 2 + (k >> 3) + (k & 7);
    const size_t ylen = zj_plane_len(&d, 0), clen = zj_plane_len(&d, 1), out_len = zj_out_len(&d);
    if (!ylen || !out_len) { fprintf(stderr, "bad geometry\n"); return 1; }
    const size_t mcu_y = (h + 15) / 16, y_row = ylen / mcu_y, c_row = clen / mcu_y; /* int16 per MCU row of a plane */

    /* pinned planes and pixels: every copy is asynchronous, in both directions */
    int16_t *y = zj_alloc_pinned(ylen * 2), *cb = zj_alloc_pinned(clen * 2), *cr = zj_alloc_pinned(clen * 2);
    uint8_t *out = zj_alloc_pinned(out_len), *want = malloc(out_len);
    if (!y || !cb || !cr || !out || !want) { fprintf(stderr, "out of memory\n"); return 1; }
    memset(y, 0, ylen * 2); memset(cb, 0, clen * 2); memset(cr, 0, clen * 2); memset(out, 0x55, out_len);

    if ((st = zj_frame_begin(ctx, &d, y, cb, cr, out, 0)) != ZJ_OK) { fprintf(stderr, "zj_frame_begin: %s\n", zj_strerror(st)); return 1; }
    uint32_t s = 2026u;
    for (size_t r = 0; r < mcu_y; r++) {            /* the "entropy decoder": one MCU row of coefficients at a time */
        for (size_t b = 0; b < y_row / 64; b++) { s = s * 1664525u + 1013904223u; int16_t *q = y + r * y_row + 64 * b; q[0] = (int16_t)((s >> 20) % 200) - 100; q[1 + (s >> 8) % 20] = (int16_t)((s >> 12) % 9) - 4; }
        for (size_t b = 0; b < c_row / 64; b++) { s = s * 1664525u + 1013904223u; cb[r * c_row + 64 * b] = (int16_t)((s >> 20) % 60) - 30; cr[r * c_row + 64 * b] = (int16_t)((s >> 10) % 60) - 30; }
        if ((st = zj_frame_rows_ready(ctx, r + 1)) != ZJ_OK) { fprintf(stderr, "zj_frame_rows_ready: %s (%s)\n", zj_strerror(st), zj_last_error(ctx)); return 1; }
    }
    if ((st = zj_frame_end(ctx)) != ZJ_OK) { fprintf(stderr, "zj_frame_end: %s (%s)\n", zj_strerror(st), zj_last_error(ctx)); return 1; }

    if ((st = zj_decode_planes(ctx, &d, y, cb, cr, want)) != ZJ_OK) { fprintf(stderr, "zj_decode_planes: %s\n", zj_strerror(st)); return 1; }
    size_t bad = 0;
    for (size_t i = 0; i < out_len; i++) bad += out[i] != want[i];
    printf("%u x %u 4:2:0 -> RGB, %zu MCU rows streamed: %zu bytes differ from zj_decode_planes\n", w, h, mcu_y, bad);
    zj_free_pinned(y); zj_free_pinned(cb); zj_free_pinned(cr); zj_free_pinned(out); free(want);
    zj_ctx_destroy(ctx);
    return bad != 0;
}
