// zj_api.cpp -- C ABI of libzjhip.so (see include/zjhip.h).  Host side of the HIP arm: argument
// validation that mirrors the reference's panics, geometry (zj_plan.h), buffer management and
// kernel launches.  There is no CPU compute path in this library.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <chrono>
#include <mutex>
#include <new>
#include <string>
#include <vector>

#include "../../include/zjhip.h"
#include "zj_launch.h"
#include "zj_plan.h"

using namespace zj;
static_assert(SCATTER_MAX == ZJ_SCATTER_MAX, "include/zjhip.h and zj_device.h disagree");

namespace {
constexpr int N_SCRATCH = 4;
constexpr int N_SLOTS = 3;                  // H2D of unit i+1 | kernel of unit i | D2H of unit i-1
constexpr size_t UNIT_TARGET_DEFAULT = 16u << 20; // coefficient bytes per pipeline unit: every copy costs ~15 us of
                                                   // launch overhead (tools/pcie_probe.py), so units stay large

// one buffer set of the host-buffer pipeline (zj_decode_planes_batch).  Three streams, one per engine:
// `up` carries every H2D copy, `run` every kernel, `down` every D2H copy (a stream per direction is what
// lets the two DMA directions overlap, tools/pcie_probe.py); events order the stages of a unit and the
// reuse of its buffers.
struct PipeSlot {
    void* buf[N_SCRATCH] = {nullptr, nullptr, nullptr, nullptr}; // y, cb, cr, out
    size_t cap[N_SCRATCH] = {0, 0, 0, 0};
    hipEvent_t up_done = nullptr, run_done = nullptr, down_done = nullptr;
    bool used = false; // events recorded at least once since the last full synchronisation
};
}

struct zj_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    void* scratch[N_SCRATCH] = {nullptr, nullptr, nullptr, nullptr};
    size_t scratch_cap[N_SCRATCH] = {0, 0, 0, 0};
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    PipeSlot slots[N_SLOTS];
    hipStream_t s_up = nullptr, s_run = nullptr, s_down = nullptr;
    int pipeline = 1;             // 0: one unit per call (no overlap), for A/B timing only
    // a frame whose planes are still being written (zj_frame_begin ... zj_frame_end)
    struct FrameStream {
        bool active = false;
        zj_frame_desc d;
        const int16_t* y = nullptr; const int16_t* cb = nullptr; const int16_t* cr = nullptr;
        uint8_t* out = nullptr;
        int on_device = 0;
        bool staged = false;      // host output in pageable memory: decoded into a device buffer, ONE copy at the end (an
                                  // asynchronous copy into pageable memory blocks its caller until the data has arrived)
        size_t submitted = 0, unit = 1, units = 0; // strips handed to the GPU so far | strips per unit | units so far
    } fs;
    std::string last_error;
#if defined(ZJ_ABLATION)
    int debug = 0;                // ablation switches, diagnostic build only (results are WRONG when set)
#endif
    int variant = 0;              // kernel variant: 0 packed generation (default), 1 wide generation (round 1), 2 packed with direct stores
    // first-wave stagger of short launches (zj_kernels.hip: stagger_start, launch_params below): CUs of the device, delay
    // per step
    int cus = 256, stagger_delay = 0;
    // GPU entropy stage (zj_decode_scan): blob + working set, the three planes (contiguous), control words read back
    // one slot per scan of a batch (zj_decode_scans): blob + working set | planes | pixels on their way to host memory
    struct HuffSlot { void* buf = nullptr; size_t cap = 0; void* planes = nullptr; void* out = nullptr; };
    HuffSlot hslot[ZJ_SCAN_BATCH_MAX];
    // planes and staged pixels of the slots live in two arenas at a uniform stride, so that scans of one geometry can go
    // through the pixel kernel as frames of ONE launch (zj_decode_scans)
    void* harena = nullptr; size_t harena_stride = 0; int harena_slots = 0;
    void* hout = nullptr; size_t hout_stride = 0; int hout_slots = 0;
    uint32_t* d_ctl = nullptr;    // the slots' control words, contiguous: one clear, one copy back per call
    uint32_t* h_ctl = nullptr;    // pinned: HUFF_CTL_WORDS per slot

    int huff_rounds = 0;          // synchronisation rounds of the last scan
    int huff_plane_slot = 0;
    size_t huff_plane_off[3] = {0, 0, 0}, huff_plane_len[3] = {0, 0, 0}; // the last call's first scan: its planes inside the slot (bytes / int16 elements)
    int huff_recent = 0;          // the most rounds a scan of the last few needed (decays): how many to launch ahead
    float huff_submit_ms = 0;     // host time of the last scan's submission (everything up to the final synchronisation)
    float huff_ms[3] = {0, 0, 0}; // with ZJ_HUFF_TIME: upload + sync rounds | scan + write + cut | pixel kernel (+ download) of the last scan
};

#define ZJ_HIP(ctx, call)                                                                          \
    do {                                                                                           \
        hipError_t e_ = (call);                                                                    \
        if (e_ != hipSuccess) {                                                                    \
            (ctx)->last_error = std::string(#call) + ": " + hipGetErrorString(e_);                 \
            return ZJ_ERR_HIP;                                                                     \
        }                                                                                          \
    } while (0)

static int ensure_scratch(zj_ctx* c, int i, size_t bytes)
{
    if (bytes <= c->scratch_cap[i]) return ZJ_OK;
    if (c->scratch[i]) { ZJ_HIP(c, hipStreamSynchronize(c->stream)); ZJ_HIP(c, hipFree(c->scratch[i])); c->scratch[i] = nullptr; c->scratch_cap[i] = 0; }
    size_t cap = bytes + bytes / 4 + 4096;
    ZJ_HIP(c, hipMalloc(&c->scratch[i], cap));
    c->scratch_cap[i] = cap;
    return ZJ_OK;
}

static int pipe_sync(zj_ctx* c)
{
    ZJ_HIP(c, hipStreamSynchronize(c->s_up));
    ZJ_HIP(c, hipStreamSynchronize(c->s_run));
    ZJ_HIP(c, hipStreamSynchronize(c->s_down));
    for (PipeSlot& sl : c->slots) sl.used = false;
    return ZJ_OK;
}

static int pipe_init(zj_ctx* c)
{
    if (c->s_up) return ZJ_OK;
    ZJ_HIP(c, hipStreamCreateWithFlags(&c->s_up, hipStreamNonBlocking));
    ZJ_HIP(c, hipStreamCreateWithFlags(&c->s_run, hipStreamNonBlocking));
    ZJ_HIP(c, hipStreamCreateWithFlags(&c->s_down, hipStreamNonBlocking));
    for (PipeSlot& sl : c->slots) {
        ZJ_HIP(c, hipEventCreateWithFlags(&sl.up_done, hipEventDisableTiming));
        ZJ_HIP(c, hipEventCreateWithFlags(&sl.run_done, hipEventDisableTiming));
        ZJ_HIP(c, hipEventCreateWithFlags(&sl.down_done, hipEventDisableTiming));
    }
    return ZJ_OK;
}

static int ensure_slot(zj_ctx* c, PipeSlot& sl, int i, size_t bytes)
{
    if (bytes <= sl.cap[i]) return ZJ_OK;
    if (sl.buf[i]) { int rc = pipe_sync(c); if (rc) return rc; ZJ_HIP(c, hipFree(sl.buf[i])); sl.buf[i] = nullptr; sl.cap[i] = 0; }
    const size_t cap = bytes + bytes / 8 + 4096;
    ZJ_HIP(c, hipMalloc(&sl.buf[i], cap));
    sl.cap[i] = cap;
    return ZJ_OK;
}

extern "C" {

int zj_abi_version(void) { return ZJ_ABI_VERSION; }

int zj_device_count(void)
{
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) return ZJ_ERR_NO_DEVICE;
    return n;
}

const char* zj_strerror(int status)
{
    switch (status) {
    case ZJ_OK: return "ok";
    case ZJ_ERR_ARG: return "invalid argument";
    case ZJ_ERR_UNSUPPORTED: return "valid for the reference but not supported by the HIP arm";
    case ZJ_ERR_HIP: return "HIP runtime error (see zj_last_error)";
    case ZJ_ERR_NOMEM: return "out of memory";
    case ZJ_ERR_PANIC: return "the reference would panic on these arguments";
    case ZJ_ERR_NO_DEVICE: return "no usable HIP device";
    case ZJ_ERR_BACKEND: return "only ZJ_BACKEND_HIP is implemented by this library";
    default: return "unknown status";
    }
}

const char* zj_last_error(const zj_ctx* ctx) { return ctx ? ctx->last_error.c_str() : ""; }

zj_ctx* zj_ctx_create(int backend, int device, int* status)
{
    int dummy;
    if (!status) status = &dummy;
    if (backend != ZJ_BACKEND_HIP) { *status = ZJ_ERR_BACKEND; return nullptr; }
    int n = zj_device_count();
    if (n <= 0 || device < 0 || device >= n) { *status = ZJ_ERR_NO_DEVICE; return nullptr; }
    zj_ctx* c = new (std::nothrow) zj_ctx();
    if (!c) { *status = ZJ_ERR_NOMEM; return nullptr; }
    c->device = device;
    if (const char* e = getenv("ZJ_VARIANT")) { int v = atoi(e); if (v >= 0 && v <= 2 && (v != 1 || fused_has_wide())) c->variant = v; }
    {
        hipDeviceProp_t prop;
        c->cus = hipGetDeviceProperties(&prop, device) == hipSuccess && prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
        c->stagger_delay = 16;                        // x 128 cycles per slot group; 0 = off (ZJ_STAGGER: the escape hatch)
        // ZJ_STAGGER = 0..255 (round 4's "delay + 256 * mode" encoding is gone: anything else is refused aloud, not ignored)
        if (const char* e = getenv("ZJ_STAGGER")) {
            char* end = nullptr;
            const long v = strtol(e, &end, 10);
            if (end != e && *end == 0 && v >= 0 && v < 256) c->stagger_delay = (int)v;
            else fprintf(stderr, "libzjhip: ZJ_STAGGER=%s is out of range (0..255, 0 = off); keeping %d\n", e, c->stagger_delay);
        }
    }
    // ZJ_BLOCKING_SYNC=1 (experiment, profiles/r06_pool_sync.txt): threads that wait for the device sleep instead of polling
    if (const char* e = getenv("ZJ_BLOCKING_SYNC")) { if (atoi(e) == 1 && hipSetDevice(device) == hipSuccess) { (void)hipSetDeviceFlags(hipDeviceScheduleBlockingSync); (void)hipGetLastError(); } }
    bool ok = hipSetDevice(device) == hipSuccess && hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) == hipSuccess &&
              hipEventCreate(&c->ev0) == hipSuccess && hipEventCreate(&c->ev1) == hipSuccess;
    if (!ok) { zj_ctx_destroy(c); *status = ZJ_ERR_NO_DEVICE; return nullptr; }
    *status = ZJ_OK;
    return c;
}

void zj_ctx_destroy(zj_ctx* c)
{
    if (!c) return;
    (void)hipSetDevice(c->device);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    for (int i = 0; i < N_SCRATCH; i++)
        if (c->scratch[i]) (void)hipFree(c->scratch[i]);
    for (auto& sl : c->hslot)
        if (sl.buf) (void)hipFree(sl.buf);
    if (c->harena) (void)hipFree(c->harena);
    if (c->hout) (void)hipFree(c->hout);
    if (c->d_ctl) (void)hipFree(c->d_ctl);
    if (c->h_ctl) (void)hipHostFree(c->h_ctl);

    for (hipStream_t st : {c->s_up, c->s_run, c->s_down})
        if (st) (void)hipStreamSynchronize(st);
    for (PipeSlot& sl : c->slots) {
        for (int i = 0; i < N_SCRATCH; i++)
            if (sl.buf[i]) (void)hipFree(sl.buf[i]);
        for (hipEvent_t ev : {sl.up_done, sl.run_done, sl.down_done})
            if (ev) (void)hipEventDestroy(ev);
    }
    for (hipStream_t st : {c->s_up, c->s_run, c->s_down})
        if (st) (void)hipStreamDestroy(st);
    if (c->ev0) (void)hipEventDestroy(c->ev0);
    if (c->ev1) (void)hipEventDestroy(c->ev1);
    if (c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
}

// The reference's fn-pointer types carry no context argument and are called concurrently from its
// worker threads (src/mcu.rs:356), so the context the shims use is per calling thread.  The reference
// makes a fresh pool for every decode (src/mcu.rs:135): contexts of finished threads go to a free list
// and are handed to the next new thread instead of being rebuilt (streams, scratch, tables).
namespace {
std::mutex g_def_mu;
std::vector<zj_ctx*> g_def_free;
struct TlsDefaultCtx {
    zj_ctx* c = nullptr;
    ~TlsDefaultCtx() // no HIP calls here: thread-exit order against the runtime's teardown is not ours
    {
        if (!c) return;
        std::lock_guard<std::mutex> lk(g_def_mu);
        g_def_free.push_back(c);
    }
};
}

zj_ctx* zj_default_ctx(void)
{
    thread_local TlsDefaultCtx t;
    if (!t.c) {
        std::lock_guard<std::mutex> lk(g_def_mu);
        if (!g_def_free.empty()) { t.c = g_def_free.back(); g_def_free.pop_back(); }
    }
    if (!t.c) { int st; t.c = zj_ctx_create(ZJ_BACKEND_HIP, 0, &st); }
    return t.c;
}

int zj_num_components(int cs) { return ncomp_of(cs); }

size_t zj_plane_len(const zj_frame_desc* d, int comp)
{
    if (!d || d->width == 0 || d->height == 0) return 0;
    if (!((d->h_max == 1 || d->h_max == 2) && (d->v_max == 1 || d->v_max == 2))) return 0;
    if (comp < 0 || comp >= (int)d->in_components || d->in_components > 3) return 0;
    const size_t mcu_x = (d->width + 8 * d->h_max - 1) / (8 * d->h_max);  // headers.rs:317
    const size_t mcu_y = (d->height + 8 * d->v_max - 1) / (8 * d->v_max); // headers.rs:319
    const size_t hs = comp == 0 ? d->h_max : 1, vs = comp == 0 ? d->v_max : 1;
    return mcu_x * 64 * vs * hs * mcu_y; // mcu_prog.rs:76
}

size_t zj_out_len(const zj_frame_desc* d)
{
    if (!d) return 0;
    // (make_plan's arithmetic without its checks: a length for any descriptor, as before)
    const bool chw = d->out_layout == ZJ_LAYOUT_CHW && d->out_colorspace == ZJ_CS_RGB;
    const size_t row = chw ? (size_t)d->width : (size_t)d->width * (size_t)ncomp_of(d->out_colorspace);
    const size_t pitch = d->out_pitch > row ? (size_t)d->out_pitch : row;
    return pitch * d->height * (chw ? 3 : 1);
}

/* ---- memory helpers ------------------------------------------------------------------------- */
// ZJ_PINNED_KIND chooses how zj_alloc_pinned gets its memory (an A/B switch: profiles/r06_feeder_ab.txt):
//   0 hipHostMalloc, portable (the default: the planes of a zj_pool are filled by entropy threads and read by submitter
//     threads' contexts, possibly on another device)
//   1 hipHostMalloc, default flags      2 hipHostMalloc, portable | NumaUser (the pages follow the calling thread's policy)
//   3 page-aligned heap memory, touched by the calling thread, then hipHostRegister'ed (portable)
//   4 hipHostMalloc, portable | non-coherent
// Registered blocks are remembered so that zj_free_pinned can undo them.
// Freed hipHostMalloc blocks are kept for the next caller (round 6): pinning 200 MB of planes costs tens of milliseconds, and
// a caller that makes a new decoder per file, as the reference's own benchmark does (benches/decode.rs:9-14), would pay it
// per file.  A block is handed out again to a request of at least half its size with the same flags on the same device (its
// pages live on that device's NUMA node); at most ZJ_PINNED_CACHE_MB (default 512, 0 = off) stay cached.
namespace {
std::mutex g_reg_mu;
std::vector<void*> g_registered;
struct PinnedBlock { void* p; size_t bytes; unsigned flags; int device; };
std::vector<PinnedBlock> g_pin_live, g_pin_free;
size_t g_pin_free_bytes = 0;
size_t pinned_cache_limit()
{
    static const size_t lim = [] {
        long mb = 512;
        if (const char* e = getenv("ZJ_PINNED_CACHE_MB")) mb = atol(e);
        return mb > 0 ? (size_t)mb << 20 : (size_t)0;
    }();
    return lim;
}
}
void* zj_alloc_pinned(size_t bytes)
{
    void* p = nullptr;
    int kind = 0;
    if (const char* e = getenv("ZJ_PINNED_KIND")) kind = atoi(e);
    if (!bytes) bytes = 1;
    if (kind == 3) {
        const size_t len = (bytes + 4095) & ~(size_t)4095;
        p = aligned_alloc(4096, len);
        if (!p) return nullptr;
        for (size_t o = 0; o < len; o += 4096) ((volatile char*)p)[o] = 0; // first touch: the pages land on this thread's node
        if (hipHostRegister(p, len, hipHostRegisterPortable) != hipSuccess) { (void)hipGetLastError(); free(p); return nullptr; }
        std::lock_guard<std::mutex> lk(g_reg_mu);
        g_registered.push_back(p);
        return p;
    }
    const unsigned flags = kind == 1 ? hipHostMallocDefault
                         : kind == 2 ? (hipHostMallocPortable | hipHostMallocNumaUser)
                         : kind == 4 ? (hipHostMallocPortable | hipHostMallocNonCoherent)
                                     : hipHostMallocPortable;
    int device = 0;
    if (hipGetDevice(&device) != hipSuccess) { (void)hipGetLastError(); device = 0; }
    {
        std::lock_guard<std::mutex> lk(g_reg_mu);
        size_t best = g_pin_free.size();
        for (size_t i = 0; i < g_pin_free.size(); i++) {
            const PinnedBlock& f = g_pin_free[i];
            if (f.flags == flags && f.device == device && f.bytes >= bytes && f.bytes / 2 <= bytes &&
                (best == g_pin_free.size() || f.bytes < g_pin_free[best].bytes)) best = i;
        }
        if (best != g_pin_free.size()) {
            const PinnedBlock f = g_pin_free[best];
            g_pin_free.erase(g_pin_free.begin() + (long)best);
            g_pin_free_bytes -= f.bytes;
            g_pin_live.push_back(f);
            return f.p;
        }
    }
    if (hipHostMalloc(&p, bytes, flags) != hipSuccess) {
        (void)hipGetLastError();
        // out of pinned memory with blocks lying idle: give them back and try once more
        std::vector<PinnedBlock> idle;
        {
            std::lock_guard<std::mutex> lk(g_reg_mu);
            idle.swap(g_pin_free);
            g_pin_free_bytes = 0;
        }
        if (idle.empty()) return nullptr;
        for (const PinnedBlock& f : idle) (void)hipHostFree(f.p);
        if (hipHostMalloc(&p, bytes, flags) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
    }
    std::lock_guard<std::mutex> lk(g_reg_mu);
    g_pin_live.push_back(PinnedBlock{p, bytes, flags, device});
    return p;
}
int zj_device_pci_bus_id(int device, char* buf, size_t cap)
{
    if (!buf || cap < 16) return ZJ_ERR_ARG;
    const int n = zj_device_count();
    if (n <= 0) return ZJ_ERR_NO_DEVICE;
    if (device < 0 || device >= n) return ZJ_ERR_ARG;
    if (hipDeviceGetPCIBusId(buf, (int)cap, device) != hipSuccess) { (void)hipGetLastError(); return ZJ_ERR_HIP; }
    return ZJ_OK;
}
int zj_set_thread_device(int device)
{
    const int n = zj_device_count();
    if (n <= 0) return ZJ_ERR_NO_DEVICE;
    if (device < 0 || device >= n) return ZJ_ERR_ARG;
    return hipSetDevice(device) == hipSuccess ? ZJ_OK : ZJ_ERR_HIP;
}
void zj_free_pinned(void* p)
{
    if (!p) return;
    {
        std::lock_guard<std::mutex> lk(g_reg_mu);
        for (size_t i = 0; i < g_registered.size(); i++)
            if (g_registered[i] == p) {
                g_registered.erase(g_registered.begin() + (long)i);
                (void)hipHostUnregister(p);
                free(p);
                return;
            }
        for (size_t i = 0; i < g_pin_live.size(); i++)
            if (g_pin_live[i].p == p) {
                const PinnedBlock f = g_pin_live[i];
                g_pin_live.erase(g_pin_live.begin() + (long)i);
                if (g_pin_free_bytes + f.bytes <= pinned_cache_limit()) { g_pin_free.push_back(f); g_pin_free_bytes += f.bytes; return; }
                break;
            }
    }
    (void)hipHostFree(p);
}
int zj_pointer_device(const void* p)
{
    if (!p) return ZJ_ERR_ARG;
    hipPointerAttribute_t a;
    memset(&a, 0, sizeof a);
    if (hipPointerGetAttributes(&a, p) != hipSuccess) { (void)hipGetLastError(); return ZJ_ERR_ARG; }
    if (a.type != hipMemoryTypeDevice) return ZJ_ERR_ARG;
    return a.device;
}
void* zj_device_alloc(zj_ctx* c, size_t bytes)
{
    if (!c || hipSetDevice(c->device) != hipSuccess) return nullptr;
    void* p = nullptr;
    if (hipMalloc(&p, bytes ? bytes : 1) != hipSuccess) return nullptr;
    return p;
}
void zj_device_free(zj_ctx* c, void* p) { if (c && p) { (void)hipSetDevice(c->device); (void)hipFree(p); } }
int zj_memcpy_h2d(zj_ctx* c, void* dst, const void* src, size_t bytes)
{
    if (!c) return ZJ_ERR_ARG;
    ZJ_HIP(c, hipSetDevice(c->device));
    ZJ_HIP(c, hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, c->stream));
    ZJ_HIP(c, hipStreamSynchronize(c->stream));
    return ZJ_OK;
}
int zj_memcpy_d2h(zj_ctx* c, void* dst, const void* src, size_t bytes)
{
    if (!c) return ZJ_ERR_ARG;
    ZJ_HIP(c, hipSetDevice(c->device));
    ZJ_HIP(c, hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, c->stream));
    ZJ_HIP(c, hipStreamSynchronize(c->stream));
    return ZJ_OK;
}
int zj_sync(zj_ctx* c)
{
    if (!c) return ZJ_ERR_ARG;
    ZJ_HIP(c, hipSetDevice(c->device));
    ZJ_HIP(c, hipStreamSynchronize(c->stream));
    return ZJ_OK;
}

// The launch itself, shared by the contiguous / strided form and the scattered form.
static int launch_params(zj_ctx* c, const Plan& pl, Params& p, hipStream_t s, bool may_stagger = true)
{
#if defined(ZJ_ABLATION)
    p.debug = c->debug;
#endif
    // Short launches of the 4:2:0 kernels (between one and two waves of workgroups: one 4096x4096 frame at a time, BASELINE
    // configs[1] read literally) start their first wave staggered, see stagger_start in zj_kernels.hip: 27.9 -> 24.3 us per
    // frame (profiles/r04_single_frame.txt).  That is the only regime it was measured in, so it is the only one that gets
    // it: a grid smaller than one wave of workgroups, or longer than two, starts as the hardware dispatches it.  The slots
    // per CU are the occupancy of the instantiation that is launched, not a constant.
    // (may_stagger false: the units of the pipelined host path -- strip ranges of a frame that overlap with copies on two
    // other streams; a unit can land in the 1..2-wave window, but nobody has measured the stagger there)
    if (may_stagger && c->stagger_delay > 0 && pl.hs == 2 && pl.vs == 2 && pl.out != OUT_GRAY && c->variant != 1) {
        const int slots = pl.fast ? fused_slots_per_cu(pl.hs, pl.vs, pl.out, c->variant, 1, p) : 0;
        const int wgs = c->cus * slots;
        if (slots >= 2 && p.total_tiles > wgs && p.total_tiles <= 2 * wgs) {
            p.stagger_wgs = wgs; p.stagger_delay = c->stagger_delay;
            const Magic g = magic_u31((uint32_t)c->cus);
            p.stagger_magic = g.m; p.stagger_shift = g.s;
        }
    }
    ZJ_HIP(c, launch_fused(pl.hs, pl.vs, pl.out, c->variant, launch_mode(pl, c->variant), p, s));
    return ZJ_OK;
}

/* ---- frame / batch level -------------------------------------------------------------------- */
// y_stride / c_stride (int16 elements) / out_stride (bytes): distance between the frames' planes / pixels when they are
// not packed back to back (0: they are)
static int decode_device_impl(zj_ctx* c, const zj_frame_desc* d, const Plan& pl, size_t nframes,
                              const int16_t* d_y, const int16_t* d_cb, const int16_t* d_cr,
                              uint8_t* d_out, hipStream_t s, int zero_fill, long long y_stride = 0, long long c_stride = 0,
                              long long out_stride = 0)
{
    Params p; // carries the quantisation tables by value: nothing to stage, nothing to order across streams
    fill_params(d, pl, nframes, d_y, d_cb, d_cr, d_out, zero_fill, p);
    if (y_stride) p.y_frame_stride = y_stride;
    if (c_stride) p.c_frame_stride = c_stride;
    if (out_stride) p.out_frame_stride = out_stride;
    const size_t ostride = out_stride ? (size_t)out_stride : pl.out_len;
    if (zero_fill) {
        // rows below the last complete strip are never written by the reference (Q6): zeros
        size_t off[3], len[3];
        ZeroRows z{};
        z.nr = uncovered_ranges(d, pl, off, len);
        for (int r = 0; r < z.nr; r++) { z.off[r] = off[r]; z.len[r] = len[r]; }
        z.frame_stride = (long long)ostride;
        for (size_t f0 = 0; z.nr && f0 < nframes; f0 += 16384) { // (grid.y <= 65535)
            z.out = d_out + f0 * ostride;
            z.nframes = (int)(nframes - f0 < 16384 ? nframes - f0 : 16384);
            ZJ_HIP(c, launch_zero_rows(z, s));
        }
    }
#if defined(ZJ_ABLATION)
    // Experiment of round 4, diagnostic build only: ONE frame cut into ZJ_SPLIT strip ranges (strips
    // are independent, src/mcu.rs:225-226) launched on internal streams that are forked from and joined to the caller's
    // stream by events.  Measured in profiles/r04_single_frame.txt: the fork / join costs more than the overlap gains.
    static const int split = [] { const char* e = getenv("ZJ_SPLIT"); const int v = e ? atoi(e) : 0; return v >= 2 && v <= 4 ? v : 0; }();
    if (split && nframes == 1 && pl.n_strips >= 2 * split) {
        static hipStream_t in[4] = {nullptr, nullptr, nullptr, nullptr};
        static hipEvent_t fork = nullptr, join[4];
        if (!fork) {
            ZJ_HIP(c, hipEventCreateWithFlags(&fork, hipEventDisableTiming));
            for (int i = 0; i < 4; i++) { ZJ_HIP(c, hipStreamCreateWithFlags(&in[i], hipStreamNonBlocking)); ZJ_HIP(c, hipEventCreateWithFlags(&join[i], hipEventDisableTiming)); }
        }
        ZJ_HIP(c, hipEventRecord(fork, s));
        const int per = (pl.n_strips + split - 1) / split;
        for (int r = 0; r < split; r++) {
            const int s0 = r * per, s1 = (s0 + per < pl.n_strips) ? s0 + per : pl.n_strips;
            Params q = p;
            const long long yrow = (long long)pl.mcu_x * pl.hs * 64 * (pl.strip_rows / 8), crow = (long long)pl.mcu_x * 64 * (pl.strip_rows / (8 * pl.vs));
            q.y = p.y + s0 * yrow; q.cb = p.cb ? p.cb + s0 * crow : nullptr; q.cr = p.cr ? p.cr + s0 * crow : nullptr;
            q.out = p.out + (size_t)s0 * pl.strip_rows * pl.out_pitch;
            q.height = (int)d->height - s0 * pl.strip_rows;
            set_grid(q, 1, s1 - s0, pl.tiles_per_row);
            ZJ_HIP(c, hipStreamWaitEvent(in[r], fork, 0));
            { const int rc = launch_params(c, pl, q, in[r]); if (rc) return rc; }
            ZJ_HIP(c, hipEventRecord(join[r], in[r]));
            ZJ_HIP(c, hipStreamWaitEvent(s, join[r], 0));
        }
        return ZJ_OK;
    }
#endif
    return launch_params(c, pl, p, s);
}

// Frames that are independent allocations (device pointers in HOST arrays, read during the call): launches of up to
// SCATTER_MAX frames whose addresses travel in the kernel arguments.  Frames that turn out to be equally spaced -- the
// images of one tensor, the slots of an arena -- take the strided form instead: one launch whatever their number.
static int decode_frames_device_impl(zj_ctx* c, const zj_frame_desc* d, const Plan& pl, size_t nframes, const int16_t* const* y,
                                     const int16_t* const* cb, const int16_t* const* cr, uint8_t* const* out, hipStream_t s,
                                     int zero_fill)
{
    const bool chroma = pl.out != OUT_GRAY;
    if (nframes == 1) return decode_device_impl(c, d, pl, 1, y[0], chroma ? cb[0] : nullptr, chroma ? cr[0] : nullptr, out[0], s, zero_fill);
    {
        const long long ys = y[1] - y[0], cs = chroma ? cb[1] - cb[0] : 0, os = out[1] - out[0];
        bool uniform = ys >= (long long)pl.y_len && os >= (long long)pl.out_len && (!chroma || (cs >= (long long)pl.c_len && cr[1] - cr[0] == cs));
        for (size_t f = 2; f < nframes && uniform; f++)
            uniform = y[f] - y[f - 1] == ys && out[f] - out[f - 1] == os && (!chroma || (cb[f] - cb[f - 1] == cs && cr[f] - cr[f - 1] == cs));
        if (uniform) return decode_device_impl(c, d, pl, nframes, y[0], chroma ? cb[0] : nullptr, chroma ? cr[0] : nullptr, out[0], s, zero_fill, ys, cs, os);
    }
    size_t off[3], len[3];
    const int nr = zero_fill ? uncovered_ranges(d, pl, off, len) : 0;
    for (size_t f0 = 0; f0 < nframes; f0 += SCATTER_MAX) {
        const int n = (int)(nframes - f0 < (size_t)SCATTER_MAX ? nframes - f0 : (size_t)SCATTER_MAX);
        Params p;
        fill_params(d, pl, (size_t)n, nullptr, nullptr, nullptr, nullptr, zero_fill, p);
        set_scatter(p, y, chroma ? cb : nullptr, chroma ? cr : nullptr, out, f0, n);
        if (nr) {
            ZeroRows z{};
            z.out = nullptr; z.frame_stride = 0; z.nr = nr; z.nframes = n;
            for (int r = 0; r < nr; r++) { z.off[r] = off[r]; z.len[r] = len[r]; }
            for (int f = 0; f < n; f++) z.fptr[f] = (uint64_t)(uintptr_t)out[f0 + f];
            ZJ_HIP(c, launch_zero_rows(z, s));
        }
        const int rc = launch_params(c, pl, p, s);
        if (rc) return rc;
    }
    return ZJ_OK;
}

static int check_frame_args(zj_ctx* c, const zj_frame_desc* d, size_t nframes, const void* y,
                            const void* cb, const void* cr, const void* out, Plan& pl)
{
    if (!c) return ZJ_ERR_ARG;
    int rc = make_plan(d, pl);
    if (rc) return rc;
    if (nframes == 0 || nframes > (size_t)1 << 20) return ZJ_ERR_ARG;
    if (!y || !out) return ZJ_ERR_ARG;
    if (pl.out != OUT_GRAY && (!cb || !cr)) return ZJ_ERR_ARG;
    if ((long long)nframes * pl.n_strips * pl.tiles_per_row > 0x7fffffffLL) return ZJ_ERR_ARG;
    return ZJ_OK;
}

int zj_decode_planes_device(zj_ctx* c, const zj_frame_desc* d, size_t nframes, const int16_t* d_y,
                            const int16_t* d_cb, const int16_t* d_cr, uint8_t* d_out, void* stream)
{
    Plan pl;
    int rc = check_frame_args(c, d, nframes, d_y, d_cb, d_cr, d_out, pl);
    if (rc) return rc;
    // planes: 16-byte aligned; pixels: 16-byte aligned when the rows are (width % 16 == 0) -- the rows of a ragged width
    // start at any byte anyway, and so may its frames (packed frames of 2500 x 1786 x 3 bytes are 8 bytes apart from it)
    if (((uintptr_t)d_y | (uintptr_t)d_cb | (uintptr_t)d_cr | (pl.fast ? (uintptr_t)d_out : 0)) & 15) return ZJ_ERR_ARG;
    ZJ_HIP(c, hipSetDevice(c->device));
    hipStream_t s = stream ? (hipStream_t)stream : c->stream;
    return decode_device_impl(c, d, pl, nframes, d_y, d_cb, d_cr, d_out, s, 1);
}

int zj_decode_planes_device_strided(zj_ctx* c, const zj_frame_desc* d, size_t nframes, const int16_t* d_y, const int16_t* d_cb,
                                    const int16_t* d_cr, uint8_t* d_out, size_t y_stride, size_t c_stride, size_t out_stride,
                                    void* stream)
{
    Plan pl;
    int rc = check_frame_args(c, d, nframes, d_y, d_cb, d_cr, d_out, pl);
    if (rc) return rc;
    if (((uintptr_t)d_y | (uintptr_t)d_cb | (uintptr_t)d_cr | (pl.fast ? (uintptr_t)d_out : 0)) & 15) return ZJ_ERR_ARG;
    // 0 = packed; otherwise at least a frame, and every frame as aligned as the first
    if ((y_stride && (y_stride < pl.y_len || (y_stride & 7))) || (c_stride && (c_stride < pl.c_len || (c_stride & 7))) ||
        (out_stride && (out_stride < pl.out_len || (pl.fast && (out_stride & 15))))) return ZJ_ERR_ARG;
    if ((y_stride | c_stride | out_stride) >> 62) return ZJ_ERR_ARG;
    ZJ_HIP(c, hipSetDevice(c->device));
    hipStream_t s = stream ? (hipStream_t)stream : c->stream;
    return decode_device_impl(c, d, pl, nframes, d_y, d_cb, d_cr, d_out, s, 1, (long long)y_stride, (long long)c_stride, (long long)out_stride);
}

int zj_decode_frames_device(zj_ctx* c, const zj_frame_desc* d, size_t nframes, const int16_t* const* d_y,
                            const int16_t* const* d_cb, const int16_t* const* d_cr, uint8_t* const* d_out, void* stream)
{
    Plan pl;
    int rc = check_frame_args(c, d, nframes, d_y, d_cb, d_cr, d_out, pl);
    if (rc) return rc;
    const bool chroma = pl.out != OUT_GRAY;
    for (size_t f = 0; f < nframes; f++) {
        if (!d_y[f] || !d_out[f] || (chroma && (!d_cb[f] || !d_cr[f]))) return ZJ_ERR_ARG;
        if (((uintptr_t)d_y[f] | (pl.fast ? (uintptr_t)d_out[f] : 0) | (chroma ? (uintptr_t)d_cb[f] | (uintptr_t)d_cr[f] : 0)) & 15) return ZJ_ERR_ARG;
    }
    ZJ_HIP(c, hipSetDevice(c->device));
    hipStream_t s = stream ? (hipStream_t)stream : c->stream;
    return decode_frames_device_impl(c, d, pl, nframes, d_y, d_cb, d_cr, d_out, s, 1);
}

int zj_time_decode_device(zj_ctx* c, const zj_frame_desc* d, size_t nframes, const int16_t* d_y,
                          const int16_t* d_cb, const int16_t* d_cr, uint8_t* d_out, void* stream,
                          int iters, float* ms_total, float* ms_each, const char** kernel_name)
{
    Plan pl;
    int rc = check_frame_args(c, d, nframes, d_y, d_cb, d_cr, d_out, pl);
    if (rc) return rc;
    if (iters <= 0 || !ms_total) return ZJ_ERR_ARG;
    ZJ_HIP(c, hipSetDevice(c->device));
    hipStream_t s = stream ? (hipStream_t)stream : c->stream;
    if (kernel_name) {
        Params p;
        fill_params(d, pl, nframes, d_y, d_cb, d_cr, d_out, 1, p);
        *kernel_name = fused_kernel_name(pl.hs, pl.vs, pl.out, c->variant, launch_mode(pl, c->variant), p);
    }
    // (1) `iters` back-to-back launches between one event pair
    ZJ_HIP(c, hipEventRecord(c->ev0, s));
    for (int i = 0; i < iters; i++) {
        rc = decode_device_impl(c, d, pl, nframes, d_y, d_cb, d_cr, d_out, s, 1);
        if (rc) return rc;
    }
    ZJ_HIP(c, hipEventRecord(c->ev1, s));
    ZJ_HIP(c, hipEventSynchronize(c->ev1));
    ZJ_HIP(c, hipEventElapsedTime(ms_total, c->ev0, c->ev1));
    // (2) every launch bracketed by its own event pair: the per-dispatch duration a profiler reports
    if (ms_each) {
        double sum = 0;
        for (int i = 0; i < iters; i++) {
            ZJ_HIP(c, hipEventRecord(c->ev0, s));
            rc = decode_device_impl(c, d, pl, nframes, d_y, d_cb, d_cr, d_out, s, 1);
            if (rc) return rc;
            ZJ_HIP(c, hipEventRecord(c->ev1, s));
            ZJ_HIP(c, hipEventSynchronize(c->ev1));
            float t = 0;
            ZJ_HIP(c, hipEventElapsedTime(&t, c->ev0, c->ev1));
            sum += t;
        }
        *ms_each = (float)(sum / iters);
    }
    return ZJ_OK;
}

namespace {
// the host frames of a batch: packed back to back behind one base pointer per plane (zj_decode_planes_batch), or
// independent allocations named by pointer arrays (zj_decode_frames)
struct HostFrames {
    const int16_t* y = nullptr; const int16_t* cb = nullptr; const int16_t* cr = nullptr; uint8_t* out = nullptr;
    const int16_t* const* ys = nullptr; const int16_t* const* cbs = nullptr; const int16_t* const* crs = nullptr; uint8_t* const* outs = nullptr;
    size_t y_len = 0, c_len = 0, out_len = 0;
    bool packed() const { return ys == nullptr; }
    const int16_t* Y(size_t f) const { return ys ? ys[f] : y + f * y_len; }
    const int16_t* Cb(size_t f) const { return cbs ? cbs[f] : (cb ? cb + f * c_len : nullptr); }
    const int16_t* Cr(size_t f) const { return crs ? crs[f] : (cr ? cr + f * c_len : nullptr); }
    uint8_t* Out(size_t f) const { return outs ? outs[f] : out + f * out_len; }
};
}

static int decode_planes_batch_impl(zj_ctx* c, const zj_frame_desc* d, const Plan& pl, size_t nframes, const HostFrames& hf);

// drains what an error half way left in flight on the three streams before the caller may free or reuse its buffers
// (the first error stays in last_error)
static int batch_guard(zj_ctx* c, int rc)
{
    if (rc && c && c->s_up) {
        const std::string keep = c->last_error;
        (void)pipe_sync(c);
        c->last_error = keep;
    }
    return rc;
}

int zj_decode_planes_batch(zj_ctx* c, const zj_frame_desc* d, size_t nframes, const int16_t* y,
                           const int16_t* cb, const int16_t* cr, uint8_t* out)
{
    Plan pl;
    const int rc = check_frame_args(c, d, nframes, y, cb, cr, out, pl);
    if (rc) return rc;
    HostFrames hf;
    hf.y = y; hf.cb = cb; hf.cr = cr; hf.out = out;
    hf.y_len = pl.y_len; hf.c_len = pl.c_len; hf.out_len = pl.out_len;
    return batch_guard(c, decode_planes_batch_impl(c, d, pl, nframes, hf));
}

int zj_decode_frames(zj_ctx* c, const zj_frame_desc* d, size_t nframes, const int16_t* const* y, const int16_t* const* cb,
                     const int16_t* const* cr, uint8_t* const* out)
{
    Plan pl;
    int rc = check_frame_args(c, d, nframes, y, cb, cr, out, pl);
    if (rc) return rc;
    const bool chroma = pl.out != OUT_GRAY;
    for (size_t f = 0; f < nframes; f++)
        if (!y[f] || !out[f] || (chroma && (!cb[f] || !cr[f]))) return ZJ_ERR_ARG;
    HostFrames hf;
    hf.ys = y; hf.cbs = chroma ? cb : nullptr; hf.crs = chroma ? cr : nullptr; hf.outs = out;
    hf.y_len = pl.y_len; hf.c_len = pl.c_len; hf.out_len = pl.out_len;
    return batch_guard(c, decode_planes_batch_impl(c, d, pl, nframes, hf));
}

// One unit of the three-stream host pipeline: strips [s0, s1) of frames [f0, f0 + nfr) -- whole frames (s0 == 0, s1 == n_strips)
// or a strip range of ONE frame -- uploaded on s_up, decoded on s_run, and (d_out null) downloaded on s_down into the host
// frames' pixels, or (d_out: the frame's device output) decoded straight into place with no copy back.  `u` numbers the units
// of the call: unit u uses buffer set u % N_SLOTS.
static int submit_unit(zj_ctx* c, const zj_frame_desc* d, const Plan& pl, const HostFrames& hf, size_t f0, size_t nfr, size_t s0, size_t s1,
                       size_t u, size_t covered, uint8_t* d_out)
{
    int rc;
    const bool chroma = pl.out != OUT_GRAY;
    const int mrps = pl.hs == 2 ? 2 : 1;                                     // MCU rows per strip
    const size_t ystrip = (size_t)pl.mcu_x * 64 * pl.hs * pl.vs * mrps;      // i16 per strip
    const size_t cstrip = (size_t)pl.mcu_x * 64 * mrps;
    const size_t ostrip = (size_t)d->width * pl.ncomp_out * pl.strip_rows;   // bytes per strip
    const size_t covered_bytes = covered * d->width * pl.ncomp_out;
    const bool planar = pl.out == OUT_RGB_CHW;
    const bool whole = s0 == 0 && s1 == (size_t)pl.n_strips;
    PipeSlot& sl = c->slots[u % N_SLOTS];
    // unit extents: whole frames keep the frame strides; a strip range is one short "frame"
    const size_t yel = whole ? nfr * pl.y_len : (s1 - s0) * ystrip;
    const size_t cel = whole ? nfr * pl.c_len : (s1 - s0) * cstrip;
    size_t obytes = whole ? nfr * pl.out_len : (s1 - s0) * ostrip;
    if (!whole && s0 * ostrip + obytes > covered_bytes) obytes = covered_bytes - s0 * ostrip;
    if ((rc = ensure_slot(c, sl, 0, yel * 2))) return rc;
    if (chroma && ((rc = ensure_slot(c, sl, 1, cel * 2)) || (rc = ensure_slot(c, sl, 2, cel * 2)))) return rc;
    if (!d_out && (rc = ensure_slot(c, sl, 3, whole ? nfr * pl.out_len : (s1 - s0) * ostrip))) return rc;
    // up: the slot's planes are free once the kernel of its previous unit has run
    if (sl.used) ZJ_HIP(c, hipStreamWaitEvent(c->s_up, sl.run_done, 0));
    if (!whole || hf.packed()) {
        const size_t yo = whole ? 0 : s0 * ystrip, co = whole ? 0 : s0 * cstrip;
        ZJ_HIP(c, hipMemcpyAsync(sl.buf[0], hf.Y(f0) + yo, yel * 2, hipMemcpyHostToDevice, c->s_up));
        if (chroma) {
            ZJ_HIP(c, hipMemcpyAsync(sl.buf[1], hf.Cb(f0) + co, cel * 2, hipMemcpyHostToDevice, c->s_up));
            ZJ_HIP(c, hipMemcpyAsync(sl.buf[2], hf.Cr(f0) + co, cel * 2, hipMemcpyHostToDevice, c->s_up));
        }
    } else {
        for (size_t f = 0; f < nfr; f++) {
            ZJ_HIP(c, hipMemcpyAsync((int16_t*)sl.buf[0] + f * pl.y_len, hf.Y(f0 + f), pl.y_len * 2, hipMemcpyHostToDevice, c->s_up));
            if (chroma) {
                ZJ_HIP(c, hipMemcpyAsync((int16_t*)sl.buf[1] + f * pl.c_len, hf.Cb(f0 + f), pl.c_len * 2, hipMemcpyHostToDevice, c->s_up));
                ZJ_HIP(c, hipMemcpyAsync((int16_t*)sl.buf[2] + f * pl.c_len, hf.Cr(f0 + f), pl.c_len * 2, hipMemcpyHostToDevice, c->s_up));
            }
        }
    }
    ZJ_HIP(c, hipEventRecord(sl.up_done, c->s_up));
    // run: after this unit's upload and after the slot's previous download has drained its pixels
    ZJ_HIP(c, hipStreamWaitEvent(c->s_run, sl.up_done, 0));
    if (sl.used && !d_out) ZJ_HIP(c, hipStreamWaitEvent(c->s_run, sl.down_done, 0));
    Params p;
    fill_params(d, pl, whole ? nfr : 1, (const int16_t*)sl.buf[0], (const int16_t*)sl.buf[1],
                (const int16_t*)sl.buf[2], d_out ? d_out + (whole ? 0 : s0 * (size_t)pl.strip_rows * pl.out_pitch) : (uint8_t*)sl.buf[3], 1, p);
    if (!whole) { // strips [s0, s1) of frame f0 as a frame of its own
        p.height = (int)d->height - (int)s0 * pl.strip_rows;
        set_grid(p, 1, (int)(s1 - s0), pl.tiles_per_row);
    }
    if ((rc = launch_params(c, pl, p, c->s_run, whole))) return rc;
    ZJ_HIP(c, hipEventRecord(sl.run_done, c->s_run));
    if (d_out) { sl.used = true; return ZJ_OK; } // the pixels are where they belong
    // down
    ZJ_HIP(c, hipStreamWaitEvent(c->s_down, sl.run_done, 0));
    if (planar) { // the covered rows of every plane of every frame
        for (size_t f = 0; f < nfr; f++)
            for (size_t pc = 0; pc < 3; pc++) {
                const size_t po = pc * (size_t)d->width * d->height;
                ZJ_HIP(c, hipMemcpyAsync(hf.Out(f0 + f) + po, (uint8_t*)sl.buf[3] + f * pl.out_len + po, covered * d->width,
                                         hipMemcpyDeviceToHost, c->s_down));
            }
    } else if (whole && (covered_bytes < pl.out_len || !hf.packed())) { // frame by frame, without the never-written rows
        for (size_t f = 0; f < nfr; f++)
            ZJ_HIP(c, hipMemcpyAsync(hf.Out(f0 + f), (uint8_t*)sl.buf[3] + f * pl.out_len, covered_bytes,
                                     hipMemcpyDeviceToHost, c->s_down));
    } else {
        ZJ_HIP(c, hipMemcpyAsync(hf.Out(f0) + (whole ? 0 : s0 * ostrip), sl.buf[3], obytes,
                                 hipMemcpyDeviceToHost, c->s_down));
    }
    ZJ_HIP(c, hipEventRecord(sl.down_done, c->s_down));
    sl.used = true;
    return ZJ_OK;
}

static int decode_planes_batch_impl(zj_ctx* c, const zj_frame_desc* d, const Plan& pl, size_t nframes, const HostFrames& hf)
{
    // Host planes -> host pixels.  The batch is cut into units of about 16 MB of coefficients (ZJ_UNIT_MB) --
    // several whole frames, or a strip range of one large frame (strips are independent: no filter tap
    // crosses a strip, Q3/Q4).  Uploads, kernels and downloads run on a stream each, chained by events over
    // N_SLOTS buffer sets, so the upload of one unit, the kernel of the previous and the download of the one
    // before overlap (PCIe is full duplex).  Pinned host buffers (zj_alloc_pinned) make the copies asynchronous.
    //   Frames that are independent allocations (zj_decode_frames) differ only in the copies: one per frame and plane
    // where packed frames take one per unit and plane; on the device a unit is packed either way.
    int rc;
    // a padded row pitch is a layout for outputs that stay on the device (the kernels never write the padding; the copies
    // of this pipeline move whole strips): host outputs are tight, as the reference's are
    if (pl.out_pitch != pl.row_bytes) return ZJ_ERR_UNSUPPORTED;
    ZJ_HIP(c, hipSetDevice(c->device));
    const bool chroma = pl.out != OUT_GRAY;
    const size_t covered = (size_t)pl.rows_covered < d->height ? (size_t)pl.rows_covered : d->height;
    // rows below the last complete strip are never written by the reference (Q6)
    {
        size_t off[3], len[3];
        const int nr = uncovered_ranges(d, pl, off, len);
        for (size_t f = 0; f < nframes; f++)
            for (int r = 0; r < nr; r++) memset(hf.Out(f) + off[r], 0, len[r]);
    }
    if (pl.n_strips == 0) return ZJ_OK;

    size_t UNIT_TARGET = UNIT_TARGET_DEFAULT;
    if (const char* e = getenv("ZJ_UNIT_MB")) { const long v = atol(e); if (v > 0 && v < 4096) UNIT_TARGET = (size_t)v << 20; }
    const size_t in_frame = (pl.y_len + (chroma ? 2 * pl.c_len : 0)) * 2;
    size_t split = 1, group = 1; // strip ranges per frame | frames per unit
    const bool planar = pl.out == OUT_RGB_CHW; // a strip range is not contiguous in a planar frame: whole frames only
    if (c->pipeline) {
        if (in_frame >= 2 * UNIT_TARGET && !planar) {
            split = (in_frame + UNIT_TARGET - 1) / UNIT_TARGET;
            if (split > (size_t)pl.n_strips) split = (size_t)pl.n_strips;
        } else {
            group = UNIT_TARGET / (in_frame ? in_frame : 1);
            if (group < 1) group = 1;
            if (group * N_SLOTS > nframes) group = (nframes + N_SLOTS - 1) / N_SLOTS; // keep every slot busy
        }
    } else group = nframes;
    const size_t strips_per_unit = ((size_t)pl.n_strips + split - 1) / split;

    if ((rc = pipe_init(c))) return rc;

    size_t u = 0;
    for (size_t f0 = 0; f0 < nframes; f0 += group) {
        const size_t nfr = f0 + group <= nframes ? group : nframes - f0;
        for (size_t s0 = 0; s0 < (size_t)pl.n_strips; s0 += strips_per_unit, u++) {
            const size_t s1 = s0 + strips_per_unit < (size_t)pl.n_strips ? s0 + strips_per_unit : (size_t)pl.n_strips;
            if ((rc = submit_unit(c, d, pl, hf, f0, nfr, s0, s1, u, covered, nullptr))) return rc;
        }
    }
    return pipe_sync(c);
}

/* ---- one frame whose planes are still being written: strips go to the GPU as they become final ----------------------
 * The reference runs post_process on strip N while its Huffman decoder is in strip N + 1 (src/mcu.rs:356-368).  Here the
 * caller names the frame (zj_frame_begin), says every now and then how many MCU rows are final (zj_frame_rows_ready),
 * and the library pushes the strips that became complete through the three-stream pipeline in units of a few MB while
 * the caller goes on filling the planes; zj_frame_end submits the rest and waits.  Every call returns at once (the copies
 * are asynchronous when the planes are pinned). */
static int frame_submit(zj_ctx* c, size_t upto, bool last)
{
    zj_ctx::FrameStream& f = c->fs;
    Plan pl;
    int rc = make_plan(&f.d, pl);
    if (rc) return rc;
    HostFrames hf;
    uint8_t* const d_out = f.on_device ? f.out : (f.staged ? (uint8_t*)c->scratch[3] : nullptr);
    hf.y = f.y; hf.cb = f.cb; hf.cr = f.cr; hf.out = d_out ? nullptr : f.out;
    hf.y_len = pl.y_len; hf.c_len = pl.c_len; hf.out_len = pl.out_len;
    const size_t covered = (size_t)pl.rows_covered < f.d.height ? (size_t)pl.rows_covered : f.d.height;
    if (upto > (size_t)pl.n_strips) upto = (size_t)pl.n_strips;
    while (f.submitted < upto && (last || upto - f.submitted >= f.unit)) {
        size_t s1 = f.submitted + f.unit;
        if (s1 > upto || (last && (size_t)pl.n_strips - s1 < f.unit / 2)) s1 = last ? (size_t)pl.n_strips : upto; // (no tiny last unit)
        if (s1 > (size_t)pl.n_strips) s1 = (size_t)pl.n_strips;
        if ((rc = submit_unit(c, &f.d, pl, hf, 0, 1, f.submitted, s1, f.units, covered, d_out))) return rc;
        f.units++;
        f.submitted = s1;
    }
    return ZJ_OK;
}

int zj_frame_begin(zj_ctx* c, const zj_frame_desc* d, const int16_t* y, const int16_t* cb, const int16_t* cr, uint8_t* out, int out_on_device)
{
    if (!c || c->fs.active) return ZJ_ERR_ARG;
    Plan pl;
    int rc = check_frame_args(c, d, 1, y, cb, cr, out, pl);
    if (rc) return rc;
    if (out_on_device && ((uintptr_t)out & 15)) return ZJ_ERR_ARG;
    if (!out_on_device && pl.out_pitch != pl.row_bytes) return ZJ_ERR_UNSUPPORTED; // (host outputs are tight)
    if (pl.out == OUT_RGB_CHW) return ZJ_ERR_UNSUPPORTED; // a strip range is not contiguous in a planar frame
    ZJ_HIP(c, hipSetDevice(c->device));
    if ((rc = pipe_init(c))) return rc;
    zj_ctx::FrameStream& f = c->fs;
    f.d = *d; f.y = y; f.cb = cb; f.cr = cr; f.out = out; f.on_device = out_on_device ? 1 : 0;
    f.submitted = 0; f.units = 0;
    f.staged = false;
    if (!out_on_device) {
        hipPointerAttribute_t a;
        memset(&a, 0, sizeof a);
        const bool pinned = hipPointerGetAttributes(&a, out) == hipSuccess && a.type == hipMemoryTypeHost;
        if (!pinned) { (void)hipGetLastError(); f.staged = true; if ((rc = ensure_scratch(c, 3, pl.out_len))) return rc; }
    }
    const bool to_device = out_on_device || f.staged;
    // units: an eighth of the frame, at least ZJ_STREAM_UNIT_MB (4) of coefficients -- every copy costs ~15 us of the calling
    // thread and of the link, and the part that cannot overlap with the caller's work is the last unit
    const bool chroma = pl.out != OUT_GRAY;
    const size_t in_frame = (pl.y_len + (chroma ? 2 * pl.c_len : 0)) * 2;
    size_t min_unit = (size_t)4 << 20;
    if (const char* e = getenv("ZJ_STREAM_UNIT_MB")) { const long v = atol(e); if (v > 0 && v < 4096) min_unit = (size_t)v << 20; }
    const size_t per_strip = pl.n_strips ? in_frame / (size_t)pl.n_strips : in_frame;
    size_t unit = ((size_t)pl.n_strips + 7) / 8;
    if (per_strip && unit * per_strip < min_unit) unit = (min_unit + per_strip - 1) / per_strip;
    if (unit < 1) unit = 1;
    f.unit = unit;
    // the buffer sets at their final size now (growing one later would synchronise the pipeline in mid-frame)
    const int mrps = pl.hs == 2 ? 2 : 1;
    const size_t span = (unit + unit / 2 + 1 < (size_t)pl.n_strips ? unit + unit / 2 + 1 : (size_t)pl.n_strips);
    for (PipeSlot& sl : c->slots) {
        if ((rc = ensure_slot(c, sl, 0, span * (size_t)pl.mcu_x * 64 * pl.hs * pl.vs * mrps * 2))) return rc;
        if (chroma && ((rc = ensure_slot(c, sl, 1, span * (size_t)pl.mcu_x * 64 * mrps * 2)) || (rc = ensure_slot(c, sl, 2, span * (size_t)pl.mcu_x * 64 * mrps * 2)))) return rc;
        if (!to_device && (rc = ensure_slot(c, sl, 3, span * (size_t)d->width * pl.ncomp_out * pl.strip_rows))) return rc;
    }
    // rows below the last complete strip are never written by the reference (Q6)
    {
        size_t off[3], len[3];
        const int nr = uncovered_ranges(d, pl, off, len);
        if (!out_on_device) { for (int r = 0; r < nr; r++) memset(out + off[r], 0, len[r]); }
        else if (nr) {
            ZeroRows z{};
            z.out = out; z.frame_stride = 0; z.nr = nr; z.nframes = 1;
            for (int r = 0; r < nr; r++) { z.off[r] = off[r]; z.len[r] = len[r]; }
            ZJ_HIP(c, launch_zero_rows(z, c->s_run));
        }
    }
    f.active = true;
    return ZJ_OK;
}

int zj_frame_rows_ready(zj_ctx* c, size_t mcu_rows)
{
    if (!c || !c->fs.active) return ZJ_ERR_ARG;
    const size_t mrps = c->fs.d.h_max == 2 ? 2 : 1; // MCU rows per strip (zj_plan.h)
    return batch_guard(c, frame_submit(c, mcu_rows / mrps, false));
}

int zj_frame_end(zj_ctx* c)
{
    if (!c || !c->fs.active) return ZJ_ERR_ARG;
    int rc = frame_submit(c, (size_t)-1, true);
    c->fs.active = false;
    if (!rc) rc = pipe_sync(c);
    if (!rc && c->fs.staged) { // the rows the strips cover (the rest was cleared in zj_frame_begin)
        Plan pl;
        if (!(rc = make_plan(&c->fs.d, pl))) {
            const size_t covered = (size_t)pl.rows_covered < c->fs.d.height ? (size_t)pl.rows_covered : c->fs.d.height;
            const hipError_t e = hipMemcpy(c->fs.out, c->scratch[3], covered * pl.out_pitch, hipMemcpyDeviceToHost);
            if (e != hipSuccess) { c->last_error = std::string("hipMemcpy (zj_frame_end): ") + hipGetErrorString(e); rc = ZJ_ERR_HIP; }
        }
    }
    return batch_guard(c, rc);
}

int zj_frame_abort(zj_ctx* c)
{
    if (!c) return ZJ_ERR_ARG;
    if (!c->fs.active) return ZJ_OK;
    c->fs.active = false;
    return c->s_up ? pipe_sync(c) : ZJ_OK;
}

int zj_decode_planes(zj_ctx* c, const zj_frame_desc* d, const int16_t* y, const int16_t* cb,
                     const int16_t* cr, uint8_t* out)
{
    return zj_decode_planes_batch(c, d, 1, y, cb, cr, out);
}

/* Host planes -> pixels that STAY in HBM (d_out: device pointer, 16-byte aligned): for consumers on the same GPU. */
int zj_decode_planes_to_device(zj_ctx* c, const zj_frame_desc* d, const int16_t* y, const int16_t* cb, const int16_t* cr,
                               uint8_t* d_out)
{
    Plan pl;
    int rc = check_frame_args(c, d, 1, y, cb, cr, d_out, pl);
    if (rc) return rc;
    if ((uintptr_t)d_out & 15) return ZJ_ERR_ARG;
    ZJ_HIP(c, hipSetDevice(c->device));
    const bool chroma = pl.out != OUT_GRAY;
    if ((rc = ensure_scratch(c, 0, pl.y_len * 2))) return rc;
    if (chroma && ((rc = ensure_scratch(c, 1, pl.c_len * 2)) || (rc = ensure_scratch(c, 2, pl.c_len * 2)))) return rc;
    ZJ_HIP(c, hipMemcpyAsync(c->scratch[0], y, pl.y_len * 2, hipMemcpyHostToDevice, c->stream));
    if (chroma) {
        ZJ_HIP(c, hipMemcpyAsync(c->scratch[1], cb, pl.c_len * 2, hipMemcpyHostToDevice, c->stream));
        ZJ_HIP(c, hipMemcpyAsync(c->scratch[2], cr, pl.c_len * 2, hipMemcpyHostToDevice, c->stream));
    }
    rc = decode_device_impl(c, d, pl, 1, (const int16_t*)c->scratch[0], (const int16_t*)c->scratch[1],
                            (const int16_t*)c->scratch[2], d_out, c->stream, 1);
    if (rc) return rc;
    ZJ_HIP(c, hipStreamSynchronize(c->stream));
    return ZJ_OK;
}

int zj_device_memset(zj_ctx* c, void* d_ptr, int value, size_t bytes)
{
    if (!c || !d_ptr) return ZJ_ERR_ARG;
    ZJ_HIP(c, hipSetDevice(c->device));
    ZJ_HIP(c, hipMemsetAsync(d_ptr, value, bytes, c->stream));
    ZJ_HIP(c, hipStreamSynchronize(c->stream));
    return ZJ_OK;
}

/* ---- GPU entropy stage + pixel path: a prepared baseline scan (zj_huff.h) -> pixels ------------------------------- */
static int ensure_buf(zj_ctx* c, void** p, size_t* cap, size_t bytes)
{
    if (bytes <= *cap) return ZJ_OK;
    if (*p) { ZJ_HIP(c, hipStreamSynchronize(c->stream)); ZJ_HIP(c, hipFree(*p)); *p = nullptr; *cap = 0; }
    const size_t want = bytes + bytes / 4 + 4096;
    ZJ_HIP(c, hipMalloc(p, want));
    *cap = want;
    return ZJ_OK;
}

namespace {
struct ScanJob {
    const zj_frame_desc* d = nullptr;
    const HuffScan* h = nullptr;
    const void* blob = nullptr;
    size_t blob_bytes = 0, nsub = 0, yb = 0, cbytes = 0, ylen = 0, clen = 0;
    bool chroma = false;
    Plan pl;
    HuffArgs a;            // device pointers of the slot
    uint8_t* out = nullptr;    // the caller's
    uint8_t* d_out = nullptr;  // where the pixel kernel writes
    uint32_t* h_ctl = nullptr; // this slot's control words on the host
    int slot = 0, max_rounds = 0, planned = 0, rounds = 0;
    int rc = ZJ_OK;        // ZJ_OK while the job is alive
    bool synced = false;
};
size_t up256(size_t x) { return (x + 255) & ~(size_t)255; }

// the arenas hold `slots` slots of `plane_bytes` / `out_bytes` each
int scan_arenas(zj_ctx* c, int slots, size_t plane_bytes, size_t out_bytes)
{
    auto grow = [&](void** p, size_t* stride, int* have, size_t need) -> int {
        if (*p && need <= *stride && slots <= *have) return ZJ_OK;
        const size_t st = need > *stride ? up256(need + need / 8) : *stride;
        const int n = slots > *have ? slots : *have;
        // nothing is remembered of an arena that is gone: if the hipMalloc below fails, the next call must not pass the
        // early-out on the strength of the freed arena's size
        if (*p) { ZJ_HIP(c, hipStreamSynchronize(c->stream)); ZJ_HIP(c, hipFree(*p)); *p = nullptr; }
        *stride = 0; *have = 0;
        ZJ_HIP(c, hipMalloc(p, st * (size_t)n));
        *stride = st; *have = n;
        return ZJ_OK;
    };
    int rc = grow(&c->harena, &c->harena_stride, &c->harena_slots, plane_bytes);
    if (rc) return rc;
    return out_bytes ? grow(&c->hout, &c->hout_stride, &c->hout_slots, out_bytes) : (int)ZJ_OK;
}

// A prepared scan is public input (zj_decode_scan / zj_decode_scans take any blob): every field the kernels index with is
// bounded here, so that a stale or malformed blob is an argument error and not an out-of-bounds access on the device.
bool scan_header_ok(const HuffScan* h, size_t blob_bytes)
{
    auto inside = [&](uint64_t off, uint64_t bytes) { return (off & 15) == 0 && off >= sizeof(HuffScan) && off + bytes <= blob_bytes; };
    if (h->ncomp != 1 && h->ncomp != 3) return false;
    if (h->bpm < 1 || h->bpm > (uint32_t)HUFF_MAX_BPM || h->nseg < 1 || h->nseg > h->nsub) return false;
    if (h->tab_entries == 0 || h->tab_entries > (uint32_t)HUFF_TAB_BUDGET || (h->tab_entries & 1)) return false;
    if (h->sub_bytes < 16 || h->sub_bytes > (uint32_t)HUFF_SUB_MAX || (h->sub_bytes & 15)) return false;
    if ((h->stream_bytes & 15) || h->stream_bytes < 32) return false;
    if (!inside(h->off_tab, (uint64_t)h->tab_entries * 2) || !inside(h->off_sub, ((uint64_t)h->nsub + 1) * sizeof(HuffSub)) ||
        !inside(h->off_seg, (uint64_t)h->nseg * sizeof(HuffSeg)) || !inside(h->off_per, (uint64_t)h->nsub * 4) ||
        !inside(h->off_stream, h->stream_bytes)) return false;
    if (h->mcu_x == 0 || h->mcu_y == 0 || (uint64_t)h->mcu_x * h->mcu_y != h->total_mcus || h->ri_mcus == 0 || h->rowlen == 0) return false;
    const uint16_t* tab = (const uint16_t*)((const uint8_t*)h + h->off_tab);
    // a decoding table: 2^l1 first-level entries at `base`, then the second-level tables its entries name (bit 15 set:
    // low byte = table number, 2^(16 - l1) entries each; huff_decode_* index base + 2^l1 + (number << (16 - l1)) + x)
    auto table_ok = [&](uint32_t base, uint32_t l1) {
        const uint64_t n1 = 1ull << l1, n2 = 1ull << (16 - l1);
        if ((uint64_t)base + n1 > h->tab_entries) return false;
        for (uint64_t q = 0; q < n1; q++) {
            const uint16_t e = tab[base + q];
            if ((e & 0x8000u) && (uint64_t)base + n1 + ((uint64_t)(e & 0xffu) + 1) * n2 > h->tab_entries) return false;
        }
        return true;
    };
    for (uint32_t k = 0; k < h->ncomp; k++) {
        if (h->comp[k].h < 1 || h->comp[k].h > 2 || h->comp[k].v < 1 || h->comp[k].v > 2) return false;
        // huff_block_ptr writes block (my * v + vy, mx * h + hx) of a plane of bw x bh blocks: the plane must hold the grid
        if ((uint64_t)h->comp[k].bw < (uint64_t)h->mcu_x * h->comp[k].h || (uint64_t)h->comp[k].bh < (uint64_t)h->mcu_y * h->comp[k].v) return false;
        if (!table_ok(h->dc_off[k], (uint32_t)HUFF_L1_DC) || !table_ok(h->ac_off[k], (uint32_t)HUFF_L1_AC)) return false;
    }
    uint32_t per_comp[3] = {0, 0, 0};
    for (uint32_t b = 0; b < h->bpm; b++) {
        const HuffBlk& k = h->blk[b];
        if (k.comp >= h->ncomp || k.hx >= h->comp[k.comp].h || k.vy >= h->comp[k.comp].v) return false;
        if (((h->comp_of_blk >> (2 * b)) & 3u) != k.comp) return false;
        per_comp[k.comp]++;
    }
    for (uint32_t k = 0; k < h->ncomp; k++)
        if (per_comp[k] != h->comp[k].h * h->comp[k].v) return false;
    // the grid and the segments: ascending, inside the stream
    const uint8_t* base = (const uint8_t*)h;
    const HuffSub* subs = (const HuffSub*)(base + h->off_sub);
    const HuffSeg* segs = (const HuffSeg*)(base + h->off_seg);
    for (uint32_t i = 0; i < h->nseg; i++)
        if (segs[i].start > segs[i].end || segs[i].end > h->stream_bytes - 32 || (segs[i].start & 15)) return false;
    for (uint32_t i = 0; i < h->nsub; i++) {
        const uint32_t sg = subs[i].seg & HUFF_SEG_MASK;
        if (sg >= h->nseg || subs[i].start < segs[sg].start || subs[i].start > segs[sg].end) return false;
        if (i && !(subs[i].seg & HUFF_FIRST) && subs[i].start <= subs[i - 1].start) return false;
    }
    // periodic-run words (zj_huff.h): huff_periodic_thread reads exit[r0 + q - 1], exit[r0 + 2q - 1] and
    // exit[r0 + q + (i - r0 - q) % q] for word i = (q << 28) | r0: a period of 1..8 sub-sequences, two whole periods in
    // front of i, all of it inside i's restart segment
    const uint32_t* per = (const uint32_t*)(base + h->off_per);
    for (uint32_t i = 0; i < h->nsub; i++) {
        const uint32_t w = per[i];
        if (!w) continue;
        const uint64_t q = w >> HUFF_PER_QSHIFT, r0 = w & HUFF_PER_MASK;
        if (q < 1 || q > HUFF_PER_MAXQ || r0 + 2 * q > i) return false;
        if ((subs[r0].seg & HUFF_SEG_MASK) != (subs[i].seg & HUFF_SEG_MASK)) return false;
    }
    return true;
}

// validates one scan, sizes its slot, fills the working-set pointers
int scan_setup(zj_ctx* c, ScanJob& j, int slot, const zj_frame_desc* d, const void* blob, size_t blob_bytes, uint8_t* out, int out_on_device)
{
    if (!d || !blob || !out || blob_bytes < sizeof(HuffScan)) return ZJ_ERR_ARG;
    const HuffScan* h = (const HuffScan*)blob;
    if (h->magic != HUFF_MAGIC || h->blob_bytes != blob_bytes || h->nsub == 0 || h->ncomp != d->in_components) return ZJ_ERR_ARG;
    if (!scan_header_ok(h, blob_bytes)) return ZJ_ERR_ARG;
    int rc = make_plan(d, j.pl);
    if (rc) return rc;
    // a padded pitch is a layout for outputs that stay in HBM (include/zjhip.h): the host copy moves pitch x height bytes out
    // of a reused staging arena, and the kernels never write the padding -- stale pixels of earlier decodes would go along
    if (!out_on_device && j.pl.out_pitch != j.pl.row_bytes) return ZJ_ERR_UNSUPPORTED;
    j.d = d; j.h = h; j.blob = blob; j.blob_bytes = blob_bytes; j.out = out; j.slot = slot;
    j.chroma = h->ncomp == 3;
    j.ylen = zj_plane_len(d, 0);
    j.clen = j.chroma ? zj_plane_len(d, 1) : 0;
    // the blob's grid is the descriptor's: same MCU counts, sampling factors and plane sizes as the plan the pixel kernel runs
    if ((int)h->mcu_x != j.pl.mcu_x || (int)h->mcu_y != j.pl.mcu_y || (int)h->comp[0].h != j.pl.hs || (int)h->comp[0].v != j.pl.vs) return ZJ_ERR_ARG;
    for (uint32_t k = 0; k < h->ncomp; k++) {
        if (k && (h->comp[k].h != 1 || h->comp[k].v != 1)) return ZJ_ERR_ARG;
        if (h->comp[k].bw != h->mcu_x * h->comp[k].h || h->comp[k].bh != h->mcu_y * h->comp[k].v) return ZJ_ERR_ARG;
    }
    if (j.ylen != (size_t)h->comp[0].bw * h->comp[0].bh * 64) return ZJ_ERR_ARG;
    if (j.chroma && (j.clen != (size_t)h->comp[1].bw * h->comp[1].bh * 64 || j.clen != (size_t)h->comp[2].bw * h->comp[2].bh * 64)) return ZJ_ERR_ARG;
    if (out_on_device && ((uintptr_t)out & 15)) return ZJ_ERR_ARG;
    j.nsub = h->nsub;
    const size_t nsub = j.nsub, nscan = (nsub + HUFF_SCAN_WG - 1) / HUFF_SCAN_WG;
    const size_t o_exit = up256(blob_bytes), o_aux = o_exit + up256(nsub * 8), o_base = o_aux + up256(nsub * 16),
                 o_lst = o_base + up256(nsub * 16), o_rel = o_lst + up256(nsub * 8 * HUFF_LIST_FACTOR),
                 o_agg = o_rel + up256(nsub), o_pre = o_agg + up256(nscan * sizeof(HuffAgg)),
                 total = o_pre + up256(nscan * sizeof(HuffAgg));
    zj_ctx::HuffSlot& sl = c->hslot[slot];
    if ((rc = ensure_buf(c, &sl.buf, &sl.cap, total))) return rc;
    j.yb = up256(j.ylen * 2);
    j.cbytes = up256(j.clen * 2);
    sl.planes = (uint8_t*)c->harena + (size_t)slot * c->harena_stride; // (the caller has sized the arenas for this call)
    sl.out = out_on_device ? nullptr : (uint8_t*)c->hout + (size_t)slot * c->hout_stride;
    uint8_t* base = (uint8_t*)sl.buf;
    HuffArgs& a = j.a;
    a.blob = base;
    a.exit = (unsigned long long*)(base + o_exit);
    a.aux = (HuffI4*)(base + o_aux);
    a.base = (HuffI4*)(base + o_base);
    a.exit_rd = a.exit;
    a.list = (uint32_t*)(base + o_lst);
    a.rel = base + o_rel;
    a.ctl = c->d_ctl + (size_t)slot * HUFF_CTL_WORDS;
    a.wgagg = (HuffAgg*)(base + o_agg);
    a.wgpre = (HuffAgg*)(base + o_pre);
    a.plane[0] = (int16_t*)sl.planes;
    a.plane[1] = (int16_t*)((uint8_t*)sl.planes + j.yb);
    a.plane[2] = (int16_t*)((uint8_t*)sl.planes + j.yb + j.cbytes);
    a.round = 0;
    a.spread = 1;
    a.zero_base = (uint8_t*)sl.planes;
    a.zero_pieces = (uint32_t)((j.yb + 2 * j.cbytes) / 16);
    j.d_out = out_on_device ? out : (uint8_t*)sl.out;
    j.h_ctl = c->h_ctl + (size_t)slot * HUFF_CTL_WORDS;
    const uint32_t sub_bytes = h->sub_bytes >= 16 && h->sub_bytes <= (uint32_t)HUFF_SUB_MAX ? h->sub_bytes : (uint32_t)HUFF_SUB_MAX;
    j.max_rounds = h->round_budget >= 1 && h->round_budget <= (uint32_t)HUFF_MAX_ROUNDS ? (int)h->round_budget : HUFF_MAX_ROUNDS;
    // Rounds launched ahead: a wrong guess falls into step within ~1 KB of 4:2:0 data, i.e. after 1024 / sub_bytes
    // rounds; files of one source need about the same number (quality 98 needs 24, a page with flat runs 40): two more
    // than the context's recent maximum, which decays by one per file
    j.planned = (int)(1536 / sub_bytes) + 2;
    if (c->huff_recent > 0) j.planned = c->huff_recent + 2;
    if (const char* e = getenv("ZJ_HUFF_ROUNDS")) { const int v = atoi(e); if (v >= 1) j.planned = v; }
    if (j.planned > j.max_rounds) j.planned = j.max_rounds;
    return ZJ_OK;
}

// before the write pass runs a second time: status, first-seen MCU and the prefix sums' ticket start over, and the planes
// hold coefficients of a wrong parse
int scan_clear_again(zj_ctx* c, ScanJob& j, hipStream_t s)
{
    ZJ_HIP(c, hipMemsetAsync(j.a.ctl, 0, (size_t)HUFF_CTL_ROUND0 * 4, s));
    ZJ_HIP(c, hipMemsetAsync(c->hslot[j.slot].planes, 0, j.yb + 2 * j.cbytes, s));
    return ZJ_OK;
}

// pixel kernel over the slot's planes, the pixels and the control words on their way back
int scan_pixels(zj_ctx* c, ScanJob& j, hipStream_t s, int out_on_device)
{
    int rc = decode_device_impl(c, j.d, j.pl, 1, j.a.plane[0], j.chroma ? j.a.plane[1] : nullptr, j.chroma ? j.a.plane[2] : nullptr, j.d_out, s, 1);
    if (rc) return rc;
    if (!out_on_device) ZJ_HIP(c, hipMemcpyAsync(j.out, j.d_out, j.pl.out_len, hipMemcpyDeviceToHost, s));
    ZJ_HIP(c, hipMemcpyAsync(j.h_ctl, j.a.ctl, (size_t)HUFF_CTL_WORDS * 4, hipMemcpyDeviceToHost, s));
    return ZJ_OK;
}

bool scan_check(ScanJob& j, int launched)
{
    for (int r = 1; r <= launched; r++)
        if (j.h_ctl[HUFF_CTL_ROUND0 + r] == 0) { j.synced = true; j.rounds = r; return true; }
    j.rounds = launched;
    return false;
}
} // namespace

// The scans of up to ZJ_SCAN_BATCH_MAX files as ONE launch per phase (blockIdx.y = file): a file's entropy kernels are
// latency-bound at under half a wave per SIMD (DESIGN_ENTROPY.md), several at once cost hardly more than one.
//   Synchronisation rounds are launched ahead and turn into no-ops once one of them changed nothing; a look at the
// counters costs a stream synchronisation, so it happens once, after the pixels.  A scan whose last planned round still
// changed something gets more rounds on its own (looking after each group), has what its premature write pass
// scattered cleared, and runs the rest again.
static int decode_scans_impl(zj_ctx* c, size_t n, const zj_frame_desc* descs, const void* const* blobs, const size_t* blob_bytes,
                             uint8_t* const* outs, int outs_on_device, int* rcs, unsigned* status_bits);

int zj_decode_scans(zj_ctx* c, size_t n, const zj_frame_desc* descs, const void* const* blobs, const size_t* blob_bytes,
                    uint8_t* const* outs, int outs_on_device, int* rcs, unsigned* status_bits)
{
    if (!c || !n || n > (size_t)ZJ_SCAN_BATCH_MAX || !descs || !blobs || !blob_bytes || !outs || !rcs) return ZJ_ERR_ARG;
    const int rc = decode_scans_impl(c, n, descs, blobs, blob_bytes, outs, outs_on_device, rcs, status_bits);
    if (rc) {
        // an error half way leaves copies and kernels in flight that read the caller's blobs and write its outputs:
        // drain them before the caller may free or reuse its buffers (the first error stays in last_error)
        const std::string keep = c->last_error;
        (void)hipStreamSynchronize(c->stream);
        c->last_error = keep;
        for (size_t k = 0; k < n; k++) if (rcs[k] == ZJ_OK) rcs[k] = rc;
    }
    return rc;
}

static int decode_scans_impl(zj_ctx* c, size_t n, const zj_frame_desc* descs, const void* const* blobs, const size_t* blob_bytes,
                             uint8_t* const* outs, int outs_on_device, int* rcs, unsigned* status_bits)
{
    ZJ_HIP(c, hipSetDevice(c->device));
    hipStream_t s = c->stream;
    if (!c->h_ctl) ZJ_HIP(c, hipHostMalloc((void**)&c->h_ctl, (size_t)ZJ_SCAN_BATCH_MAX * HUFF_CTL_WORDS * 4, hipHostMallocPortable));
    if (!c->d_ctl) ZJ_HIP(c, hipMalloc((void**)&c->d_ctl, (size_t)ZJ_SCAN_BATCH_MAX * HUFF_CTL_WORDS * 4));
    static_assert(ZJ_SCAN_BATCH_MAX == HUFF_BATCH_MAX, "include/zjhip.h and zj_huff.h disagree");
    HuffBatch batch;
    const bool timing = getenv("ZJ_HUFF_TIME") != nullptr;
    const auto t_submit0 = std::chrono::steady_clock::now();
    hipEvent_t ev[4] = {nullptr, nullptr, nullptr, nullptr};
    if (timing) { for (auto& e : ev) ZJ_HIP(c, hipEventCreate(&e)); ZJ_HIP(c, hipEventRecord(ev[0], s)); }
    {   // size the arenas for the largest scan of the call (a scan that fails validation below asks for nothing)
        size_t pmax = 0, omax = 0;
        for (size_t k = 0; k < n; k++) {
            if (!blobs[k] || blob_bytes[k] < sizeof(HuffScan)) continue;
            const HuffScan* h = (const HuffScan*)blobs[k];
            Plan pl;
            if (h->magic != HUFF_MAGIC || make_plan(&descs[k], pl)) continue;
            const size_t pb = up256(zj_plane_len(&descs[k], 0) * 2) + 2 * up256((h->ncomp == 3 ? zj_plane_len(&descs[k], 1) : 0) * 2);
            if (pb > pmax) pmax = pb;
            if (!outs_on_device && pl.out_len > omax) omax = pl.out_len;
        }
        if (pmax) { const int rc = scan_arenas(c, (int)n, pmax, omax); if (rc) return rc; }
    }
    std::vector<ScanJob> jobs(n);
    int live[ZJ_SCAN_BATCH_MAX], nlive = 0; // jobs that passed validation, packed: the kernels' argument array
    uint32_t max_nsub = 0;
    int planned = 1;
    for (size_t k = 0; k < n; k++) {
        if (status_bits) status_bits[k] = 0;
        rcs[k] = jobs[k].rc = scan_setup(c, jobs[k], (int)k, &descs[k], blobs[k], blob_bytes[k], outs[k], outs_on_device);
        if (jobs[k].rc) continue;
        live[nlive] = (int)k;
        batch.a[nlive++] = jobs[k].a;
        if (jobs[k].nsub > max_nsub) max_nsub = (uint32_t)jobs[k].nsub;
        if (jobs[k].planned > planned) planned = jobs[k].planned;
    }
    if (!nlive) return ZJ_OK;
    {   // alone on the GPU a scan's sparse rounds spread out (one entry per wave), in a batch they pack (HuffArgs::spread)
        int spread = nlive == 1 ? 64 : nlive == 2 ? 16 : nlive <= 4 ? 4 : 1;
        if (const char* e = getenv("ZJ_HUFF_SPREAD")) { const int v = atoi(e); if (v >= 1 && v <= 64) spread = v; }
        for (int q = 0; q < nlive; q++) batch.a[q].spread = spread;
    }
    ZJ_HIP(c, hipMemsetAsync(c->d_ctl, 0, n * (size_t)HUFF_CTL_WORDS * 4, s)); // (the planes: round 0 clears them on the side)
    for (int q = 0; q < nlive; q++) {
        ScanJob& j = jobs[(size_t)live[q]];
        ZJ_HIP(c, hipMemcpyAsync((void*)j.a.blob, j.blob, j.blob_bytes, hipMemcpyHostToDevice, s));
    }
    bool periodic = false; // some scan has periodic runs (flat areas): the rule of zj_huff.h runs between the rounds
    for (int q = 0; q < nlive; q++) periodic = periodic || jobs[(size_t)live[q]].h->nper != 0;
    for (int round = 0; round <= planned; round++) ZJ_HIP(c, launch_huff_sync(batch, nlive, max_nsub, round, periodic, s));
    if (timing) ZJ_HIP(c, hipEventRecord(ev[1], s));
    ZJ_HIP(c, launch_huff_finish(batch, nlive, max_nsub, s));
    if (timing) ZJ_HIP(c, hipEventRecord(ev[2], s));
    // Pixel kernel: the scans of one geometry and one set of tables go through it as the frames of ONE launch, wherever
    // their planes (arena slots) and their pixels (the staging arena, or whatever the caller's pointers are) happen to lie:
    // equally spaced frames take the strided form, anything else the scattered form (decode_frames_device_impl)
    {
        bool launched[ZJ_SCAN_BATCH_MAX] = {};
        for (int q = 0; q < nlive; q++) {
            if (launched[q]) continue;
            const ScanJob& j0 = jobs[(size_t)live[q]];
            const int16_t* fy[ZJ_SCAN_BATCH_MAX]; const int16_t* fcb[ZJ_SCAN_BATCH_MAX]; const int16_t* fcr[ZJ_SCAN_BATCH_MAX];
            uint8_t* fo[ZJ_SCAN_BATCH_MAX];
            size_t run = 0;
            for (int r = q; r < nlive; r++) {
                const ScanJob& jn = jobs[(size_t)live[r]];
                if (launched[r] || memcmp(jn.d, j0.d, sizeof(zj_frame_desc)) != 0) continue;
                fy[run] = jn.a.plane[0]; fcb[run] = jn.a.plane[1]; fcr[run] = jn.a.plane[2]; fo[run] = jn.d_out;
                launched[r] = true;
                run++;
            }
            const int rc = decode_frames_device_impl(c, j0.d, j0.pl, run, fy, fcb, fcr, fo, s, 1);
            if (rc) return rc;
        }
    }
    for (int q = 0; q < nlive && !outs_on_device; q++) {
        ScanJob& j = jobs[(size_t)live[q]];
        ZJ_HIP(c, hipMemcpyAsync(j.out, j.d_out, j.pl.out_len, hipMemcpyDeviceToHost, s));
    }
    ZJ_HIP(c, hipMemcpyAsync(c->h_ctl, c->d_ctl, n * (size_t)HUFF_CTL_WORDS * 4, hipMemcpyDeviceToHost, s)); // every slot's control words
    if (timing) ZJ_HIP(c, hipEventRecord(ev[3], s));
    c->huff_submit_ms = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t_submit0).count();
    ZJ_HIP(c, hipStreamSynchronize(s));
    if (timing) {
        for (int k = 0; k < 3; k++) ZJ_HIP(c, hipEventElapsedTime(&c->huff_ms[k], ev[k], ev[k + 1]));
        for (auto& e : ev) (void)hipEventDestroy(e);
    }
    int most = 0;
    c->huff_rounds = 0;
    for (int q = 0; q < nlive; q++) {
        ScanJob& j = jobs[(size_t)live[q]];
        if (!scan_check(j, planned)) {
            // the slow way, this scan alone
            HuffBatch one;
            one.a[0] = j.a;
            one.a[0].spread = 64;
            int round = planned, group = 16;
            while (!j.synced && round < j.max_rounds) {
                const int first = round + 1;
                for (int k = 0; k < group && round < j.max_rounds; k++) ZJ_HIP(c, launch_huff_sync(one, 1, (uint32_t)j.nsub, ++round, j.h->nper != 0, s));
                ZJ_HIP(c, hipMemcpyAsync(j.h_ctl, j.a.ctl, (size_t)HUFF_CTL_WORDS * 4, hipMemcpyDeviceToHost, s));
                ZJ_HIP(c, hipStreamSynchronize(s));
                for (int r = first; r <= round; r++)
                    if (j.h_ctl[HUFF_CTL_ROUND0 + r] == 0) { j.synced = true; j.rounds = r; break; }
                group *= 2;
            }
            if (!j.synced) { // the CPU walker is the faster way out (nothing to learn from for the next files)
                j.rounds = round;
                rcs[live[q]] = ZJ_RETRY_CPU;
                if (status_bits) status_bits[live[q]] = HUFF_ST_NO_SYNC;
                if (j.rounds > c->huff_rounds) c->huff_rounds = j.rounds;
                continue;
            }
            int rc = scan_clear_again(c, j, s);
            if (rc) return rc;
            ZJ_HIP(c, launch_huff_finish(one, 1, (uint32_t)j.nsub, s));
            if ((rc = scan_pixels(c, j, s, outs_on_device))) return rc;
            ZJ_HIP(c, hipStreamSynchronize(s));
        }
        if (j.rounds > most) most = j.rounds;
        if (status_bits) status_bits[live[q]] = j.h_ctl[HUFF_CTL_STATUS];
        rcs[live[q]] = j.h_ctl[HUFF_CTL_STATUS] ? ZJ_RETRY_CPU : ZJ_OK;
    }
    if (most > c->huff_rounds) c->huff_rounds = most;
    // (rounds in units of the scans' sub-sequence size; a context is normally fed one kind of file; an outlier fades
    // by a quarter per call)
    if (most >= c->huff_recent) c->huff_recent = most;
    else { const int faded = c->huff_recent - (c->huff_recent / 4 > 1 ? c->huff_recent / 4 : 1); c->huff_recent = faded > most ? faded : most; }
    const ScanJob& j0 = jobs[(size_t)live[0]]; // zj_scan_planes looks at the first scan of the last call
    c->huff_plane_slot = j0.slot;
    c->huff_plane_off[0] = 0; c->huff_plane_off[1] = j0.yb; c->huff_plane_off[2] = j0.yb + j0.cbytes;
    c->huff_plane_len[0] = j0.ylen; c->huff_plane_len[1] = c->huff_plane_len[2] = j0.clen;
    return ZJ_OK;
}

int zj_decode_scan(zj_ctx* c, const zj_frame_desc* d, const void* blob, size_t blob_bytes, uint8_t* out, int out_on_device,
                   unsigned* status_bits)
{
    if (status_bits) *status_bits = 0;
    if (!c || !d) return ZJ_ERR_ARG;
    int rc1 = ZJ_OK;
    const int rc = zj_decode_scans(c, 1, d, &blob, &blob_bytes, &out, out_on_device, &rc1, status_bits);
    return rc ? rc : rc1;
}

int zj_scan_planes(zj_ctx* c, int16_t* y, int16_t* cb, int16_t* cr, size_t len[3])
{
    if (!c || !c->hslot[c->huff_plane_slot].planes || !c->huff_plane_len[0]) return ZJ_ERR_ARG;
    ZJ_HIP(c, hipSetDevice(c->device));
    int16_t* dst[3] = {y, cb, cr};
    for (int k = 0; k < 3; k++) {
        if (len) len[k] = c->huff_plane_len[k];
        if (dst[k] && c->huff_plane_len[k])
            ZJ_HIP(c, hipMemcpyAsync(dst[k], (const uint8_t*)c->hslot[c->huff_plane_slot].planes + c->huff_plane_off[k], c->huff_plane_len[k] * 2, hipMemcpyDeviceToHost, c->stream));
    }
    ZJ_HIP(c, hipStreamSynchronize(c->stream));
    return ZJ_OK;
}

int zj_scan_stats(const zj_ctx* c, int* rounds, float ms[4])
{
    if (!c) return ZJ_ERR_ARG;
    if (rounds) *rounds = c->huff_rounds;
    if (ms) { for (int k = 0; k < 3; k++) ms[k] = c->huff_ms[k]; ms[3] = c->huff_submit_ms; }
    return ZJ_OK;
}

/* ---- strip level ---------------------------------------------------------------------------- */
int zj_idct_strip(zj_ctx* c, const int16_t* coeff, size_t n, const int32_t qt[64], size_t stride,
                  size_t samp_factors, size_t v_samp, int16_t* out)
{
    if (!c || !qt || (n && (!coeff || !out))) return ZJ_ERR_ARG;
    if (samp_factors == 0) return ZJ_ERR_PANIC;          // division by zero, scalar.rs:30
    if (n == 0) return ZJ_OK;
    for (int k = 0; k < 64; k++)
        if (qt[k] < 0 || qt[k] > 255) return ZJ_ERR_UNSUPPORTED;
    const size_t chunks = n * v_samp / samp_factors;     // scalar.rs:30
    if (chunks == 0) return ZJ_ERR_PANIC;                // chunks_exact(0)
    const size_t nchunks = n / chunks, bpc = chunks / 64;
    // get_mut(pos..pos+8).unwrap() (scalar.rs:52,249): the last block's last row must fit the chunk
    if (bpc > 0 && (bpc - 1) * 8 + 7 * stride + 8 > chunks) return ZJ_ERR_PANIC;
    ZJ_HIP(c, hipSetDevice(c->device));
    int rc;
    if ((rc = ensure_scratch(c, 0, n * 2)) || (rc = ensure_scratch(c, 1, n * 2))) return rc;
    hipStream_t s = c->stream;
    ZJ_HIP(c, hipMemcpyAsync(c->scratch[0], coeff, n * 2, hipMemcpyHostToDevice, s));
    ZJ_HIP(c, hipMemsetAsync(c->scratch[1], 0, n * 2, s)); // vec![0; len], scalar.rs:26
    ZJ_HIP(c, launch_idct_strip((const int16_t*)c->scratch[0], qt, (int16_t*)c->scratch[1],
                                (long long)(nchunks * bpc), (long long)chunks, (long long)bpc, (long long)stride, s));
    ZJ_HIP(c, hipMemcpyAsync(out, c->scratch[1], n * 2, hipMemcpyDeviceToHost, s));
    ZJ_HIP(c, hipStreamSynchronize(s));
    return ZJ_OK;
}

// runs one flat filter on device buffers (scratch[si] -> scratch[so]); no sync
static int upsample_h_dev(zj_ctx* c, int si, size_t n, int so, size_t out_len)
{
    if (!(out_len > 4 && n > 2)) return ZJ_ERR_PANIC; // assert!, upsampler/scalar.rs:9-12
    const size_t m = ((out_len - 2) / 2 < n - 2) ? (out_len - 2) / 2 : n - 2; // zip(chunks_exact_mut(2), windows(3))
    ZJ_HIP(c, hipMemsetAsync(c->scratch[so], 0, out_len * 2, c->stream));
    ZJ_HIP(c, launch_upsample_h((const int16_t*)c->scratch[si], (long long)n, (int16_t*)c->scratch[so],
                                (long long)out_len, (long long)m, c->stream));
    return ZJ_OK;
}
static int upsample_v_dev(zj_ctx* c, int si, size_t n, int so, size_t out_len)
{
    const size_t stride = n >> 3; // upsampler/scalar.rs:73
    if (stride == 0) return ZJ_ERR_PANIC;
    // out[i..].split_at_mut(stride) for i = 0, 2*stride, ... 14*stride (scalar.rs:95-110)
    if (14 * stride + stride > out_len) return ZJ_ERR_PANIC;
    ZJ_HIP(c, hipMemsetAsync(c->scratch[so], 0, out_len * 2, c->stream));
    ZJ_HIP(c, launch_upsample_v((const int16_t*)c->scratch[si], (long long)stride, (int16_t*)c->scratch[so],
                                (long long)out_len, c->stream));
    return ZJ_OK;
}

static int upsample_common(zj_ctx* c, int kind, const int16_t* in, size_t n, int16_t* out, size_t out_len)
{
    if (!c || !in || !out) return ZJ_ERR_ARG;
    ZJ_HIP(c, hipSetDevice(c->device));
    int rc;
    const size_t mid = n * 2;
    if ((rc = ensure_scratch(c, 0, (n ? n : 1) * 2)) || (rc = ensure_scratch(c, 1, (out_len ? out_len : 1) * 2)) ||
        (rc = ensure_scratch(c, 2, (mid ? mid : 1) * 2))) return rc;
    ZJ_HIP(c, hipMemcpyAsync(c->scratch[0], in, n * 2, hipMemcpyHostToDevice, c->stream));
    if (kind == 0) rc = upsample_h_dev(c, 0, n, 1, out_len);
    else if (kind == 1) rc = upsample_v_dev(c, 0, n, 1, out_len);
    else { // upsample_hv = horizontal(vertical(input, 2*len), output_len), scalar.rs:148-166
        rc = upsample_v_dev(c, 0, n, 2, mid);
        if (!rc) rc = upsample_h_dev(c, 2, mid, 1, out_len);
    }
    if (rc) { (void)hipStreamSynchronize(c->stream); return rc; }
    ZJ_HIP(c, hipMemcpyAsync(out, c->scratch[1], out_len * 2, hipMemcpyDeviceToHost, c->stream));
    ZJ_HIP(c, hipStreamSynchronize(c->stream));
    return ZJ_OK;
}

int zj_upsample_h(zj_ctx* c, const int16_t* in, size_t n, int16_t* out, size_t out_len) { return upsample_common(c, 0, in, n, out, out_len); }
int zj_upsample_v(zj_ctx* c, const int16_t* in, size_t n, int16_t* out, size_t out_len) { return upsample_common(c, 1, in, n, out, out_len); }
int zj_upsample_hv(zj_ctx* c, const int16_t* in, size_t n, int16_t* out, size_t out_len) { return upsample_common(c, 2, in, n, out, out_len); }

int zj_ycbcr_to_rgb16(zj_ctx* c, const int16_t y[16], const int16_t cb[16], const int16_t cr[16],
                      uint8_t* out, size_t out_len, size_t* pos)
{
    if (!c || !y || !cb || !cr || !out || !pos) return ZJ_ERR_ARG;
    if (*pos > out_len || out_len - *pos < 48) return ZJ_ERR_PANIC; // "Slice to small cannot write", scalar.rs:57-64
    ZJ_HIP(c, hipSetDevice(c->device));
    int rc;
    if ((rc = ensure_scratch(c, 0, 96)) || (rc = ensure_scratch(c, 1, 48))) return rc;
    int16_t h[48];
    memcpy(h, y, 32); memcpy(h + 16, cb, 32); memcpy(h + 32, cr, 32);
    ZJ_HIP(c, hipMemcpyAsync(c->scratch[0], h, 96, hipMemcpyHostToDevice, c->stream));
    ZJ_HIP(c, hipStreamSynchronize(c->stream)); // `h` is pageable stack memory
    ZJ_HIP(c, launch_rgb16((const int16_t*)c->scratch[0], (uint8_t*)c->scratch[1], c->stream));
    ZJ_HIP(c, hipMemcpyAsync(out + *pos, c->scratch[1], 48, hipMemcpyDeviceToHost, c->stream));
    ZJ_HIP(c, hipStreamSynchronize(c->stream));
    *pos += 48;
    return ZJ_OK;
}

int zj_post_process_strip(zj_ctx* c, const int16_t* const coeff[3], const size_t len[3],
                          const zj_component comps[3], int in_cs, int out_cs, uint8_t* out,
                          size_t out_len, size_t width)
{
    if (!c || !coeff || !len || !comps || !out) return ZJ_ERR_ARG;
    const size_t hs = comps[0].horizontal_sample, vs = comps[0].vertical_sample; // worker.rs:43-45
    if (!((hs == 1 || hs == 2) && (vs == 1 || vs == 2))) return ZJ_ERR_UNSUPPORTED;
    if (in_cs != ZJ_CS_YCBCR && in_cs != ZJ_CS_GRAYSCALE) return ZJ_ERR_UNSUPPORTED;
    if (width == 0 || width > 65535) return ZJ_ERR_ARG;
    // One strip == a frame that is exactly one strip tall (mcu.rs:225-226).
    zj_frame_desc d;
    memset(&d, 0, sizeof d);
    d.width = (uint32_t)width;
    d.h_max = (uint32_t)hs; d.v_max = (uint32_t)vs;
    d.in_components = in_cs == ZJ_CS_YCBCR ? 3 : 1;
    d.out_colorspace = out_cs;
    const uint32_t strip_rows = (hs == 2 && vs == 2) ? 32 : ((hs == 2 || vs == 2) ? 16 : 8);
    d.height = strip_rows;
    for (int k = 0; k < 3; k++) memcpy(d.qt[k], comps[k < (int)d.in_components ? k : 0].quantization_table, 256);
    Plan pl;
    int rc = make_plan(&d, pl);
    if (rc) return rc;
    // the strip must have the geometry the reference's callers produce (headers.rs:338, mcu.rs:238-250)
    if (comps[0].width_stride != (size_t)pl.mcu_x * 8 * hs || len[0] != pl.y_len) return ZJ_ERR_UNSUPPORTED;
    const bool chroma = pl.out != OUT_GRAY;
    if (chroma && (len[1] != pl.c_len || len[2] != pl.c_len || comps[1].width_stride != (size_t)pl.mcu_x * 8 ||
                   comps[2].width_stride != (size_t)pl.mcu_x * 8)) return ZJ_ERR_UNSUPPORTED;
    if (!coeff[0] || (chroma && (!coeff[1] || !coeff[2]))) return ZJ_ERR_ARG;
    if (out_len < pl.out_len) return ZJ_ERR_PANIC; // &mut output[start..end], worker.rs:174
    ZJ_HIP(c, hipSetDevice(c->device));
    const size_t yb = pl.y_len * 2, cbytes = chroma ? pl.c_len * 2 : 0;
    if ((rc = ensure_scratch(c, 0, yb))) return rc;
    if (chroma && ((rc = ensure_scratch(c, 1, cbytes)) || (rc = ensure_scratch(c, 2, cbytes)))) return rc;
    if ((rc = ensure_scratch(c, 3, pl.out_len))) return rc;
    hipStream_t s = c->stream;
    ZJ_HIP(c, hipMemcpyAsync(c->scratch[0], coeff[0], yb, hipMemcpyHostToDevice, s));
    if (chroma) {
        ZJ_HIP(c, hipMemcpyAsync(c->scratch[1], coeff[1], cbytes, hipMemcpyHostToDevice, s));
        ZJ_HIP(c, hipMemcpyAsync(c->scratch[2], coeff[2], cbytes, hipMemcpyHostToDevice, s));
    }
    // bytes the reference never writes keep the caller's content: round-trip the caller's buffer
    ZJ_HIP(c, hipMemcpyAsync(c->scratch[3], out, pl.out_len, hipMemcpyHostToDevice, s));
    rc = decode_device_impl(c, &d, pl, 1, (const int16_t*)c->scratch[0], (const int16_t*)c->scratch[1],
                            (const int16_t*)c->scratch[2], (uint8_t*)c->scratch[3], s, 0);
    if (rc) return rc;
    ZJ_HIP(c, hipMemcpyAsync(out, c->scratch[3], pl.out_len, hipMemcpyDeviceToHost, s));
    ZJ_HIP(c, hipStreamSynchronize(s));
    return ZJ_OK;
}

/* ---- dispatch mirror ------------------------------------------------------------------------ */
zj_idct_fn zj_choose_idct_func(int backend) { return backend == ZJ_BACKEND_HIP ? zj_idct_strip : nullptr; }
zj_upsample_fn zj_choose_upsample_func(int backend, int h_max, int v_max)
{
    if (backend != ZJ_BACKEND_HIP) return nullptr;
    if (h_max == 2 && v_max == 1) return zj_upsample_h;  // decoder.rs:480-491
    if (h_max == 1 && v_max == 2) return zj_upsample_v;  // decoder.rs:492-501
    if (h_max == 2 && v_max == 2) return zj_upsample_hv; // decoder.rs:502-511
    return nullptr;                                      // (1,1): upsample_no_op / unknown ratio
}
zj_color_convert16_fn zj_choose_ycbcr_to_rgb_convert_func(int backend, int out_cs)
{
    // the reference only ever asks for ColorSpace::RGB (decoder.rs:127-128)
    if (backend != ZJ_BACKEND_HIP || out_cs != ZJ_CS_RGB) return nullptr;
    return zj_ycbcr_to_rgb16;
}

/* kernel-variant switch for A/B measurements (both variants are bit-exact) */
int zj_variant_available(int variant) { return variant == 0 || variant == 2 || (variant == 1 && fused_has_wide()) ? 1 : 0; }
int zj_set_variant(zj_ctx* c, int variant)
{
    if (!c || variant < 0 || variant > 2) return ZJ_ERR_ARG;
    if (variant == 1 && !fused_has_wide()) return ZJ_ERR_UNSUPPORTED; // round 1's generation: `make VARIANTS=all` builds (tests, A/B)
    c->variant = variant;
    return ZJ_OK;
}
/* 0 = one unit per zj_decode_planes_batch call (no copy/compute overlap); A/B timing only */
int zj_set_pipeline(zj_ctx* c, int on) { if (!c) return ZJ_ERR_ARG; c->pipeline = on ? 1 : 0; return ZJ_OK; }
} // extern "C"

#if defined(ZJ_ABLATION)
/* diagnostic build only (tools/build_variant.sh ablate "-DZJ_ABLATION"; never in the product library) */
/* occupancy probe (tools/occupancy.py): pad every fused launch with dynamic LDS; query workgroups per CU */
extern "C" __attribute__((visibility("default"))) int zj_set_pad_lds(int bytes) { set_pad_lds(bytes); return ZJ_OK; }
extern "C" __attribute__((visibility("default"))) int zj_fused_occupancy(int pad_lds) { return fused_occupancy_420_rgb(pad_lds); }
/* ablation for tools/ablate.py: bit 0 skips the IDCT, bit 1 the colour math; output is WRONG when non-zero */
extern "C" __attribute__((visibility("default"))) int zj_set_ablation(zj_ctx* c, int mask) { if (!c) return ZJ_ERR_ARG; c->debug = mask; return ZJ_OK; }
#endif

