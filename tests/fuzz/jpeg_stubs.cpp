// Test-only stand-ins for the few C-ABI functions zj_jpeg.cpp calls outside itself, so that the entropy front-end can be
// built alone (g++, AddressSanitizer + UBSan) and fed damaged files on the CPU.  Never part of the product library.
#include <stdlib.h>
#include <string.h>

#include "../../include/zjhip.h"

extern "C" {
void* zj_alloc_pinned(size_t bytes) { return malloc(bytes ? bytes : 1); }
void zj_free_pinned(void* p) { free(p); }
size_t zj_out_len(const zj_frame_desc* d)
{
    const int cs = d->out_colorspace;
    const size_t nc = (cs == ZJ_CS_GRAYSCALE) ? 1 : ((cs == ZJ_CS_RGB || cs == ZJ_CS_YCBCR) ? 3 : 4);
    return (size_t)d->width * d->height * nc;
}
int zj_decode_planes(zj_ctx*, const zj_frame_desc*, const int16_t*, const int16_t*, const int16_t*, uint8_t*) { return ZJ_ERR_NO_DEVICE; }
int zj_decode_planes_to_device(zj_ctx*, const zj_frame_desc*, const int16_t*, const int16_t*, const int16_t*, uint8_t*) { return ZJ_ERR_NO_DEVICE; }
int zj_decode_scan(zj_ctx*, const zj_frame_desc*, const void*, size_t, uint8_t*, int, unsigned*) { return ZJ_ERR_NO_DEVICE; }
int zj_decode_scans(zj_ctx*, size_t, const zj_frame_desc*, const void* const*, const size_t*, uint8_t* const*, int, int*, unsigned*) { return ZJ_ERR_NO_DEVICE; }
int zj_device_memset(zj_ctx*, void*, int, size_t) { return ZJ_ERR_NO_DEVICE; }
int zj_frame_begin(zj_ctx*, const zj_frame_desc*, const int16_t*, const int16_t*, const int16_t*, uint8_t*, int) { return ZJ_ERR_NO_DEVICE; }
int zj_frame_rows_ready(zj_ctx*, size_t) { return ZJ_ERR_NO_DEVICE; }
int zj_frame_end(zj_ctx*) { return ZJ_ERR_NO_DEVICE; }
int zj_frame_abort(zj_ctx*) { return ZJ_OK; }
const char* zj_strerror(int) { return "stub"; }
const char* zj_last_error(const zj_ctx*) { return ""; }
}
