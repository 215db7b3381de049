// zj_lab_api.cpp -- C entry points of libzjlab.so: timing harness around the micro-benchmark kernels (zj_ubench.hip:
// issue cost of gfx950 integer VALU instructions) and the lab kernels (zj_lab.hip: IDCT / colour formulations and
// memory access patterns in isolation).  Used by tools/ubench.py and tools/lab.py only; nothing of the decode path.
#include <hip/hip_runtime.h>

#include <cstddef>
#include <cstdint>

#include "../zj_plan.h"
#include "zj_lab_launch.h"

using namespace zj;

namespace {
struct LabCtx {
    int device = 0;
    hipStream_t stream = nullptr;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    void* buf[2] = {nullptr, nullptr};
    size_t cap[2] = {0, 0};
};
#define LAB_HIP(call) do { if ((call) != hipSuccess) return -1; } while (0)

int ensure(LabCtx* c, int i, size_t bytes)
{
    if (c->cap[i] >= bytes) return 0;
    if (c->buf[i]) { LAB_HIP(hipStreamSynchronize(c->stream)); LAB_HIP(hipFree(c->buf[i])); c->buf[i] = nullptr; c->cap[i] = 0; }
    LAB_HIP(hipMalloc(&c->buf[i], bytes));
    c->cap[i] = bytes;
    return 0;
}
template <class F>
int timed(LabCtx* c, int reps, float* ms, F&& launch)
{
    LAB_HIP(launch(-1)); // warm-up
    LAB_HIP(hipEventRecord(c->ev0, c->stream));
    for (int r = 0; r < reps; r++) LAB_HIP(launch(r));
    LAB_HIP(hipEventRecord(c->ev1, c->stream));
    LAB_HIP(hipEventSynchronize(c->ev1));
    LAB_HIP(hipEventElapsedTime(ms, c->ev0, c->ev1));
    return 0;
}
} // namespace

#define LAB_API extern "C" __attribute__((visibility("default")))

LAB_API void* zjlab_create(int device)
{
    if (hipSetDevice(device) != hipSuccess) return nullptr;
    LabCtx* c = new LabCtx;
    c->device = device;
    if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess || hipEventCreate(&c->ev0) != hipSuccess ||
        hipEventCreate(&c->ev1) != hipSuccess) { delete c; return nullptr; }
    return c;
}
LAB_API void zjlab_destroy(void* h)
{
    LabCtx* c = (LabCtx*)h;
    if (!c) return;
    (void)hipStreamSynchronize(c->stream);
    for (int i = 0; i < 2; i++) if (c->buf[i]) (void)hipFree(c->buf[i]);
    (void)hipEventDestroy(c->ev0); (void)hipEventDestroy(c->ev1); (void)hipStreamDestroy(c->stream);
    delete c;
}
LAB_API int zjlab_ubench_count(void) { return ubench2_count(); }
LAB_API const char* zjlab_ubench_name(int op) { return ubench2_name(op); }
LAB_API int zjlab_ubench(void* h, int op, int blocks, int iters, int reps, float* ms)
{
    LabCtx* c = (LabCtx*)h;
    if (!c || !ms || ensure(c, 0, (size_t)blocks * 256 * 4)) return -1;
    return timed(c, reps, ms, [&](int r) { return launch_ubench2(op, (int*)c->buf[0], blocks, iters, 12345 + (r < 0 ? 0 : r), c->stream); });
}
LAB_API int zjlab_labmem_count(void) { return labmem_count(); }
LAB_API const char* zjlab_labmem_name(int i) { return labmem_name(i); }
LAB_API int zjlab_labmem(void* h, int variant, long long bytes, int reps, float* ms)
{
    LabCtx* c = (LabCtx*)h;
    if (!c || !ms || bytes % (384 * 256) != 0 || ensure(c, 0, (size_t)bytes) || ensure(c, 1, (size_t)bytes)) return -1;
    LAB_HIP(hipMemsetAsync(c->buf[0], 1, (size_t)bytes, c->stream));
    for (int r = 0; r < 2; r++) LAB_HIP(launch_labmem(variant, c->buf[0], c->buf[1], bytes, c->stream));
    return timed(c, reps, ms, [&](int) { return launch_labmem(variant, c->buf[0], c->buf[1], bytes, c->stream); });
}
LAB_API int zjlab_lab_count(void) { return lab_count(); }
LAB_API const char* zjlab_lab_name(int i) { return lab_name(i); }
LAB_API int zjlab_lab(void* h, int variant, int blocks, int iters, int reps, float* ms)
{
    LabCtx* c = (LabCtx*)h;
    if (!c || !ms || ensure(c, 0, (size_t)blocks * 256 * 4)) return -1;
    int32_t qt3[3][64];
    for (int k = 0; k < 3; k++) for (int i = 0; i < 64; i++) qt3[k][i] = 1 + ((i * 7 + k * 3) % 29);
    return timed(c, reps, ms, [&](int) { return launch_lab(variant, qt3, (int*)c->buf[0], blocks, iters, c->stream); });
}
/* LDS-DMA read path of zj_lab.hip against the plain one: returns the number of blocks whose sums differ (0 = equal) */
LAB_API int zjlab_rd_check(void* h, int mode, int ntiles)
{
    LabCtx* c = (LabCtx*)h;
    const size_t bytes = (size_t)ntiles * 24576, nsum = (size_t)ntiles * 192;
    if (!c || ntiles <= 0 || ensure(c, 0, bytes) || ensure(c, 1, nsum * 8)) return -1;
    uint32_t* host = new uint32_t[bytes / 4];
    uint32_t x = 12345;
    for (size_t i = 0; i < bytes / 4; i++) { x = x * 1664525u + 1013904223u; host[i] = x; }
    LAB_HIP(hipMemcpy(c->buf[0], host, bytes, hipMemcpyHostToDevice));
    delete[] host;
    uint32_t* d = (uint32_t*)c->buf[1];
    LAB_HIP(hipMemsetAsync(d, 0, nsum * 8, c->stream));
    LAB_HIP(launch_lab_rd_check(0, c->buf[0], d, ntiles, c->stream));
    LAB_HIP(launch_lab_rd_check(mode, c->buf[0], d + nsum, ntiles, c->stream));
    LAB_HIP(hipStreamSynchronize(c->stream));
    uint32_t* got = new uint32_t[2 * nsum];
    LAB_HIP(hipMemcpy(got, d, nsum * 8, hipMemcpyDeviceToHost));
    int bad = 0;
    for (size_t i = 0; i < nsum; i++) bad += got[i] != got[nsum + i] || got[i] == 0;
    delete[] got;
    return bad;
}
/* shader clock: cycles counted by s_memtime in one wave over a fixed spin, and the wall ms of it */
LAB_API int zjlab_clock(void* h, int iters, double* cycles, float* ms)
{
    LabCtx* c = (LabCtx*)h;
    if (!c || !cycles || !ms || ensure(c, 0, 4096)) return -1;
    if (timed(c, 1, ms, [&](int) { return launch_ub_clock((unsigned long long*)c->buf[0], 1, iters, c->stream); })) return -1;
    unsigned long long v = 0;
    LAB_HIP(hipMemcpy(&v, c->buf[0], 8, hipMemcpyDeviceToHost));
    *cycles = (double)v;
    return 0;
}

/* persistent-workgroup experiment (lab/zj_persist.hip): `reps` launches of the whole batch, device pointers in the product's
 * plane layout; mode 0 = persistent + LDS-DMA prefetch, 1 = persistent only; groups = workgroups in the grid (0: CUs x
 * the kernel's occupancy).  *ms = total of the timed launches; returns the grid size used, or < 0. */
LAB_API int zjlab_persist(void* h, const zj_frame_desc* d, int nframes, const int16_t* y, const int16_t* cb, const int16_t* cr,
                          uint8_t* out, int groups, int mode, int reps, float* ms)
{
    LabCtx* c = (LabCtx*)h;
    Plan pl;
    if (!c || !ms || make_plan(d, pl) != ZJ_OK || pl.hs != 2 || pl.vs != 2 || pl.out != OUT_RGB || !pl.fast || nframes < 1) return -1;
    Params p;
    fill_params(d, pl, (size_t)nframes, y, cb, cr, out, 1, p);
    if (!ts_eligible<Cfg<2, 2, OUT_RGB>>(p, OUT_RGB, true)) return -2;
    if (groups <= 0) {
        hipDeviceProp_t prop;
        LAB_HIP(hipGetDeviceProperties(&prop, c->device));
        const int occ = persist_occupancy(mode);
        if (occ <= 0) return -3;
        groups = prop.multiProcessorCount * occ;
    }
    if (timed(c, reps, ms, [&](int) { return launch_persist(mode, p, groups, c->stream); })) return -4;
    return groups;
}
