/* decode_file.c -- the C ABI end to end: JPEG file -> RGB (or RGBA / planar) pixels on an MI355X.
 *
 *   cc -I include examples/decode_file.c -L zune-jpeg_amd -lzjhip -Wl,-rpath,$PWD/zune-jpeg_amd -o decode_file
 *   ./decode_file in.jpg out.ppm [gpu]      "gpu": the Huffman stage runs on the device as well (zj_options.entropy)
 *
 * Mirrors `Decoder::new_with_options(...).decode_buffer(&bytes)` of the reference (src/decoder.rs:178). */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "zjhip.h"

int main(int argc, char **argv)
{
    if (argc < 3) { fprintf(stderr, "usage: %s in.jpg out.ppm [gpu]\n", argv[0]); return 2; }
    FILE *f = fopen(argv[1], "rb");
    if (!f) { perror(argv[1]); return 1; }
    fseek(f, 0, SEEK_END);
    long n = ftell(f);
    fseek(f, 0, SEEK_SET);
    uint8_t *buf = (uint8_t *)malloc((size_t)n);
    if (fread(buf, 1, (size_t)n, f) != (size_t)n) { fclose(f); return 1; }
    fclose(f);

    int st = ZJ_OK;
    zj_ctx *ctx = zj_ctx_create(ZJ_BACKEND_HIP, 0, &st);     /* fails loudly without a GPU: no CPU fallback */
    if (!ctx) { fprintf(stderr, "zj_ctx_create: %s\n", zj_strerror(st)); return 1; }
    zj_options opt;
    memset(&opt, 0, sizeof opt);                              /* zero = the reference's defaults */
    opt.out_colorspace = ZJ_CS_RGB;
    opt.num_threads = 4;                                      /* restart segments decode concurrently */
    opt.pinned_planes = 1;                                    /* coefficient planes are DMA sources */
    if (argc > 3 && !strcmp(argv[3], "gpu")) opt.entropy = ZJ_ENTROPY_GPU; /* baseline scans of 32 KB and more: Huffman on the
                                                                 device, same bytes; everything else as before */
    zj_decoder *dec = zj_decoder_new(&opt);

    zj_image_info info;
    if ((st = zj_decoder_read_headers(dec, buf, (size_t)n, &info)) != ZJ_OK) {
        fprintf(stderr, "headers: [%d] %s\n", st, zj_decoder_error(dec));
        return 1;
    }
    const size_t ncomp = info.components == 1 ? 1 : 3;
    const size_t cap = (size_t)info.width * info.height * ncomp;
    uint8_t *px = (uint8_t *)zj_alloc_pinned(cap);            /* pinned: the download is a plain DMA */
    size_t len = 0;
    if ((st = zj_decoder_decode_buffer(dec, ctx, buf, (size_t)n, px, cap, &len, &info)) != ZJ_OK) {
        fprintf(stderr, "decode: [%d] %s\n", st, zj_decoder_error(dec));
        return 1;
    }
    FILE *o = fopen(argv[2], "wb");
    if (!o) { perror(argv[2]); return 1; }
    fprintf(o, "P%d\n%u %u\n255\n", ncomp == 1 ? 5 : 6, info.width, info.height);
    fwrite(px, 1, len, o);
    fclose(o);
    printf("%ux%u, %u component(s), %s, %u scan(s): %zu bytes\n", info.width, info.height, info.components,
           info.progressive ? "progressive" : "baseline", info.scans, len);
    zj_free_pinned(px);
    zj_decoder_free(dec);
    zj_ctx_destroy(ctx);
    free(buf);
    return 0;
}
