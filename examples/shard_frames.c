/* shard_frames.c -- image-level sharding over the GPUs of one node from plain C: N frames of coefficient planes the caller
 * owns one by one (the shape the reference's callers have: a Vec per strip / per decode, src/mcu.rs:238-250,
 * src/decoder.rs:178) -> N RGB frames, every GPU decoding its contiguous shard.  No collective: frames are independent
 * (the reference's own unit of independence is the strip, src/mcu.rs:225-226, 356-368).
 *
 *   cc -I include examples/shard_frames.c -L zune-jpeg_amd -lzjhip -Wl,-rpath,$PWD/zune-jpeg_amd -o shard_frames
 *   ./shard_frames [frames] [width] [height]        (all devices of the node; one device works too)
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "zjhip.h"

int main(int argc, char **argv)
{
    const size_t n = argc > 1 ? (size_t)atol(argv[1]) : 16;
    const uint32_t w = argc > 2 ? (uint32_t)atoi(argv[2]) : 4096, h = argc > 3 ? (uint32_t)atoi(argv[3]) : 4096;
    int ndev = zj_device_count();
    if (ndev <= 0) { fprintf(stderr, "no HIP device: %s\n", zj_strerror(ndev)); return 1; } /* no CPU fallback */
    if (ndev > 64) ndev = 64;
    int devices[64], st = ZJ_OK;
    for (int k = 0; k < ndev; k++) devices[k] = k;
    zj_multi *m = zj_multi_create(devices, ndev, &st);
    if (!m) { fprintf(stderr, "zj_multi_create: %s\n", zj_strerror(st)); return 1; }

    zj_frame_desc d;
    memset(&d, 0, sizeof d);
    d.width = w; d.height = h; d.h_max = 2; d.v_max = 2; d.in_components = 3; d.out_colorspace = ZJ_CS_RGB;
    for (int c = 0; c < 3; c++)
        for (int k = 0; k < 64; k++) d.qt[c][k] = 1 + (k >> 3) + (k & 7); /* any 8-bit table */
    const size_t ylen = zj_plane_len(&d, 0), clen = zj_plane_len(&d, 1), olen = zj_out_len(&d);
    if (!ylen || !olen) { fprintf(stderr, "bad geometry\n"); return 1; }

    /* every frame: its own pinned planes and its own pinned output (DMA without staging) */
    const int16_t **y = calloc(n, sizeof *y), **cb = calloc(n, sizeof *cb), **cr = calloc(n, sizeof *cr);
    uint8_t **out = calloc(n, sizeof *out);
    for (size_t f = 0; f < n; f++) {
        int16_t *py = zj_alloc_pinned(2 * ylen), *pb = zj_alloc_pinned(2 * clen), *pr = zj_alloc_pinned(2 * clen);
        out[f] = zj_alloc_pinned(olen);
        if (!py || !pb || !pr || !out[f]) { fprintf(stderr, "zj_alloc_pinned failed\n"); return 1; }
        memset(py, 0, 2 * ylen); memset(pb, 0, 2 * clen); memset(pr, 0, 2 * clen);
        for (size_t b = 0; b < ylen / 64; b++) py[64 * b] = (int16_t)((b * 7 + f * 13) % 200 - 100); /* a DC pattern */
        y[f] = py; cb[f] = pb; cr[f] = pr;
    }
    int statuses[64];
    st = zj_multi_decode_frames(m, &d, n, y, cb, cr, out, statuses);
    for (int k = 0; k < ndev; k++) {
        size_t lo, hi, done;
        int dev;
        zj_shard_range(n, k, ndev, &lo, &hi);
        zj_multi_slot_stats(m, k, &dev, &done);
        printf("slot %d (device %d): frames [%zu, %zu)  status %d (%s)  decoded so far %zu\n", k, dev, lo, hi, statuses[k],
               zj_strerror(statuses[k]), done);
    }
    if (st != ZJ_OK) { fprintf(stderr, "decode: %s\n", zj_strerror(st)); return 1; }
    unsigned long long sum = 0;
    for (size_t f = 0; f < n; f++)
        for (size_t i = 0; i < olen; i += 4097) sum += out[f][i];
    printf("%zu frames of %ux%u decoded over %d device(s); sample sum %llu\n", n, w, h, ndev, sum);
    for (size_t f = 0; f < n; f++) {
        zj_free_pinned((void *)y[f]); zj_free_pinned((void *)cb[f]); zj_free_pinned((void *)cr[f]); zj_free_pinned(out[f]);
    }
    free(y); free(cb); free(cr); free(out);
    zj_multi_destroy(m);
    return 0;
}
