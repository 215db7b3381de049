// zj_pool.cpp -- batch decoding of JPEG files with a persistent two-stage host pipeline:
//
//   `threads` entropy workers   container parsing + Huffman (zj_jpeg.cpp) into PINNED coefficient planes
//            | queue of decoded files (bounded by the number of plane sets)
//   3 GPU submitters            each with its own zj_ctx: H2D -> fused kernel -> D2H (zj_api.cpp)
//
// The serial, branchy entropy stage (the true end-to-end bottleneck) is spread over the host
// cores; three submitters are enough to keep both PCIe directions and the kernel queue busy (while one waits
// for its download, another uploads) without dozens of threads contending inside the HIP runtime.  A plane
// set (= one zj_decoder) goes back to the free list as soon as its pixels are out.
// The reference creates a scoped_threadpool per decode for its post_process strips (src/mcu.rs:135); this
// pool lives across calls because streams, pinned planes and device buffers are worth keeping.
//
// Several devices (zj_pool_create_multi): image-level sharding inside the library (north_star: "independent frames across
// the 8 GPUs of one node"; SURVEY.md 8e: one host thread per GPU).  Every device slot is a small pool of its own -- entropy
// workers, plane sets, submitters, contexts -- whose threads are bound to the NUMA node of its GPU (zj_numa.cpp) and whose
// plane sets are pinned from there, so a file's coefficients are written, pinned and DMA'd on one socket (round 6; before,
// workers and plane sets were shared and landed wherever the creating thread ran).  A file whose pixels go to host memory
// is decoded by whichever worker gets to it and submitted by that worker's slot; a slot whose queue is empty takes a
// prepared file from a slot with a backlog (so a slow GPU does not hold up its share).  A file whose pixels stay in HBM is
// decoded by a worker of the slot that owns its output pointer -- or, when that slot's workers have run dry, by another
// slot's -- and always submitted on the owning device.  Results land in the caller's order because every file carries
// its own output pointer.
//
// Only the C ABI of the library is used (include/zjhip.h, plus the zj_set_pipeline knob).
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include <atomic>
#include <chrono>
#include <condition_variable>
#include <deque>
#include <mutex>
#include <new>
#include <string>
#include <thread>
#include <vector>

#include "../../include/zjhip.h"

extern "C" int zj_set_pipeline(zj_ctx* c, int on);

namespace {
// Three submitters keep both PCIe directions and the kernel queue busy when planes go up and pixels come down.  With
// the entropy stage on the device and the pixels left there, a submitter takes up to eight ready files at once (their
// scans run as one launch per phase, zj_decode_scans): three submitters x 8 files measured best (4 900 files/s; eight
// submitters x 4: 3 600).
constexpr int GPU_SUBMITTERS = 3;
}

struct zj_pool {
    struct Batch {
        size_t n = 0;
        const uint8_t* const* bufs = nullptr;
        const size_t* lens = nullptr;
        uint8_t* const* outs = nullptr;
        const size_t* caps = nullptr;
        size_t* out_lens = nullptr;
        zj_image_info* infos = nullptr;
        int* statuses = nullptr;
        bool on_device = false; // outs[] are device pointers on the pool's device: the pixels stay in HBM
        size_t next = 0;     // next file to entropy-decode   (under mu)
        size_t done = 0;     // files finished or failed       (under mu)
        int first_error = 0;
    };
    struct Job { size_t index; zj_decoder* dec; int home; }; // home: the slot whose plane set dec is
    struct Slot {                        // one device of the pool (the same device may fill several slots)
        int device = 0;
        int node = -1, n_threads = 0, n_bound = 0; // NUMA node of the device; this slot's threads, and how many were bound there
        std::deque<Job> ready;           // prepared files this slot's submitters should take
        std::vector<zj_decoder*> free_dec; // this slot's plane sets (pinned by its workers: on its device's node)
        std::deque<size_t> todo;         // device-output batch: files whose output pointer lives on this slot's device
        std::condition_variable cv;      // this slot's submitters
        int idle = 0;                    // ... of which so many are waiting for a file
        std::condition_variable cv_work; // this slot's entropy workers: files are left and one of its plane sets is free
        double gpu_s = 0;                // under mu
        size_t files = 0;
    };

    std::vector<std::thread> threads;
    std::vector<zj_decoder*> decoders;   // all plane sets
    std::vector<zj_ctx*> ctxs;           // one per submitter
    std::deque<Slot> slots;              // (a deque: condition variables do not move)
    std::mutex mu;
    // one condition per kind of waiter and slot, so that a finished file wakes one submitter or one worker and not all 20+
    // threads of the pool (with the device entropy stage a file is ~1 ms of work: the wake-ups showed)
    std::condition_variable cv_done;     // the caller: the batch is complete
    int settled = 0, expected_threads = 0; // threads that have placed themselves (bind_here), threads there will be
    Batch* batch = nullptr;
    bool stop = false;
    std::mutex call_mu;                  // serialises zj_pool_decode_files callers
    std::string last_error;
    int n_workers = 0;
    int file_threads = 1; // zj_options.num_threads the pool was created with (threads inside one file)
    bool lend_idle = true; // ZJ_POOL_LEND=off: a short batch's files stay on file_threads threads (the A/B)
    std::vector<int> out_slot;           // per file of a device-output batch: the slot its output pointer belongs to
    int device_batch = 8;                // files a submitter takes at once when the pixels stay on the device (ZJ_POOL_BATCH)
    // accumulated over the pool's life (under mu): seconds inside the entropy stage / the GPU stage, files
    double entropy_s = 0, gpu_s = 0;
    size_t files_done = 0;
    static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

    void finish(Batch& b, size_t i, int rc, const char* text, size_t olen, const zj_image_info* info)
    {   // under mu
        if (b.out_lens) b.out_lens[i] = olen;
        if (b.infos && info) b.infos[i] = *info;
        if (b.statuses) b.statuses[i] = rc;
        if (rc && !b.first_error) { b.first_error = rc; last_error = "file " + std::to_string(i) + ": " + (text ? text : ""); }
        b.done++;
    }

    bool files_left(const Batch& b, int k) const
    {   // under mu: is there a file a worker of slot k may decode?
        if (!b.on_device) return b.next < b.n;
        if (!slots[(size_t)k].todo.empty()) return true;
        for (const Slot& sl : slots) if (!sl.todo.empty()) return true; // (its own have run dry: help another slot)
        return false;
    }
    size_t take_file(Batch& b, int k)
    {   // under mu, files_left(b, k)
        if (!b.on_device) return b.next++;
        Slot* from = &slots[(size_t)k];
        if (from->todo.empty())
            for (Slot& sl : slots) if (sl.todo.size() > from->todo.size()) from = &sl;
        const size_t i = from->todo.front();
        from->todo.pop_front();
        b.next++;
        return i;
    }
    void bind_here(Slot& me)
    {   // the calling thread next to the slot's GPU (zj_numa.cpp); counts are read by zj_pool_slot_numa
        const int node = zj_device_numa_node(me.device);
        const bool bound = zj_bind_thread_near_device(me.device) >= 0;
        std::lock_guard<std::mutex> lk(mu);
        me.node = node;
        me.n_threads++;
        if (bound) me.n_bound++;
        if (++settled == expected_threads) cv_done.notify_all();
    }

    void entropy_loop(int slot_index)
    {
        Slot& me = slots[(size_t)slot_index];
        // the planes this thread fills are pinned (zj_alloc_pinned) and DMA'd by the submitters' contexts:
        // bind the thread to the slot's device, so a pool on device N never touches device 0 -- and hipHostMalloc places the
        // planes on that device's NUMA node -- and to that node's CPUs
        (void)zj_set_thread_device(me.device);
        bind_here(me);
        std::unique_lock<std::mutex> lk(mu);
        for (;;) {
            me.cv_work.wait(lk, [&] { return stop || (batch && !me.free_dec.empty() && files_left(*batch, slot_index)); });
            if (stop) return;
            Batch& b = *batch;
            const size_t i = take_file(b, slot_index);
            zj_decoder* dec = me.free_dec.back();
            me.free_dec.pop_back();
            // A batch of fewer files than workers leaves workers idle: their share of the CPUs goes to the files there are
            // (restart segments, or a scan without them entered at several points: zj_jpeg.cpp scan_baseline_parallel).
            // Two 4096^2 q90 files on a pool of 16: 18 -> 5 ms of Huffman each.
            const int lend = lend_idle && b.n && b.n * 2 <= (size_t)n_workers ? (int)((size_t)n_workers / b.n) : 1;
            lk.unlock();
            (void)zj_decoder_set_num_threads(dec, lend > file_threads ? (lend < 16 ? lend : 16) : file_threads);
            zj_image_info info;
            memset(&info, 0, sizeof info);
            zj_frame_desc fd;
            const double t0 = now();
            const int rc = zj_decoder_prepare(dec, b.bufs[i], b.lens[i], &fd, &info); // Huffman here, or only the byte-level preparation for the device (zj_options.entropy)
            const double dt = now() - t0;
            lk.lock();
            entropy_s += dt;
            if (rc) {
                finish(b, i, rc, zj_decoder_error(dec), 0, &info);
                me.free_dec.push_back(dec);
                if (b.done == b.n) cv_done.notify_all();
                me.cv_work.notify_one();
            } else {
                if (b.infos) b.infos[i] = info;
                // device output: the slot that owns the pointer; host output: this slot (its planes are on this socket) --
                // and once this slot has a backlog, one submitter of every other slot gets a look (gpu_loop steals)
                Slot& to = b.on_device ? slots[(size_t)out_slot[i]] : me;
                to.ready.push_back(Job{i, dec, slot_index});
                to.cv.notify_one();
                if (!b.on_device && to.ready.size() > (size_t)to.idle)
                    for (Slot& sl : slots) if (&sl != &to) sl.cv.notify_one();
            }
        }
    }

    void gpu_loop(zj_ctx* ctx, int slot_index)
    {
        Slot& me = slots[(size_t)slot_index];
        (void)zj_set_thread_device(me.device);
        bind_here(me);
        std::unique_lock<std::mutex> lk(mu);
        auto backlog_elsewhere = [&]() -> Slot* { // host outputs only: the slot with the longest queue of prepared files
            if (!batch || batch->on_device) return nullptr;
            Slot* best = nullptr;
            for (Slot& sl : slots) // (a file an idle submitter of its own slot is about to take is not a backlog)
                if (&sl != &me && sl.ready.size() > (size_t)sl.idle && (!best || sl.ready.size() > best->ready.size())) best = &sl;
            return best;
        };
        for (;;) {
            me.idle++;
            me.cv.wait(lk, [&] { return stop || !me.ready.empty() || backlog_elsewhere(); });
            me.idle--;
            if (stop) return;
            Batch& b = *batch;
            // Pixels that stay on the device: take what is ready, up to a batch -- the device entropy stage runs the scans of
            // several files as one launch per phase (zj_decode_scans).  Host outputs: one file at a time, the 48 MB
            // downloads of three submitters are what fills PCIe.
            Job jobs[ZJ_SCAN_BATCH_MAX];
            size_t nj = 0;
            const size_t want = b.on_device ? (size_t)device_batch : 1;
            while (nj < want && !me.ready.empty()) { jobs[nj++] = me.ready.front(); me.ready.pop_front(); }
            if (!nj) { Slot* from = backlog_elsewhere(); if (from) { jobs[nj++] = from->ready.front(); from->ready.pop_front(); } }
            if (!nj) continue;
            lk.unlock();
            zj_decoder* decs[ZJ_SCAN_BATCH_MAX];
            uint8_t* outs[ZJ_SCAN_BATCH_MAX];
            size_t caps[ZJ_SCAN_BATCH_MAX], olens[ZJ_SCAN_BATCH_MAX];
            int rcs[ZJ_SCAN_BATCH_MAX];
            for (size_t q = 0; q < nj; q++) { decs[q] = jobs[q].dec; outs[q] = b.outs[jobs[q].index]; caps[q] = b.caps[jobs[q].index]; olens[q] = 0; rcs[q] = 0; }
            const double t0 = now();
            if (nj == 1) rcs[0] = b.on_device ? zj_decoder_finish_pixels_device(decs[0], ctx, outs[0], caps[0], &olens[0])
                                              : zj_decoder_finish_pixels(decs[0], ctx, outs[0], caps[0], &olens[0]);
            else {
                const int rc = zj_decoder_finish_pixels_batch(decs, nj, ctx, outs, caps, olens, b.on_device ? 1 : 0, rcs);
                if (rc) for (size_t q = 0; q < nj; q++) rcs[q] = rc;
            }
            const double dt = now() - t0;
            lk.lock();
            gpu_s += dt;
            files_done += nj;
            me.gpu_s += dt;
            me.files += nj;
            for (size_t q = 0; q < nj; q++) {
                finish(b, jobs[q].index, rcs[q], rcs[q] ? zj_decoder_error(jobs[q].dec) : nullptr, olens[q], nullptr);
                Slot& home = slots[(size_t)jobs[q].home];
                home.free_dec.push_back(jobs[q].dec); // the plane set goes back to the slot (the socket) it belongs to
                home.cv_work.notify_one();
            }
            if (b.done == b.n) cv_done.notify_all();
        }
    }
};

extern "C" {

void zj_pool_destroy(zj_pool* p)
{
    if (!p) return;
    {
        std::lock_guard<std::mutex> lk(p->mu);
        p->stop = true;
    }
    for (auto& sl : p->slots) { sl.cv_work.notify_all(); sl.cv.notify_all(); }
    for (auto& th : p->threads) if (th.joinable()) th.join();
    for (zj_decoder* d : p->decoders) zj_decoder_free(d);
    for (zj_ctx* c : p->ctxs) zj_ctx_destroy(c);
    delete p;
}

zj_pool* zj_pool_create_multi(const int* devices, int ndev, int threads_per_device, const zj_options* opt, int* status)
{
    int dummy;
    if (!status) status = &dummy;
    if (!devices || ndev <= 0 || ndev > 64 || threads_per_device <= 0 || (long long)threads_per_device * ndev > 1024) { *status = ZJ_ERR_ARG; return nullptr; }
    const int ngpu = zj_device_count();
    if (ngpu <= 0) { *status = ZJ_ERR_NO_DEVICE; return nullptr; }
    for (int k = 0; k < ndev; k++)
        if (devices[k] < 0 || devices[k] >= ngpu) { *status = ZJ_ERR_NO_DEVICE; return nullptr; }
    zj_pool* p = new (std::nothrow) zj_pool();
    if (!p) { *status = ZJ_ERR_NOMEM; return nullptr; }
    zj_options o;
    memset(&o, 0, sizeof o);
    if (opt) o = *opt;
    o.pinned_planes = getenv("ZJ_POOL_HEAP_PLANES") ? 0 : 1; // planes are DMA sources (the env knob is for A/B timing)
    if (o.num_threads <= 0) o.num_threads = 1; // the pool is the parallelism; > 1 adds restart-segment threads per file
    const int threads = threads_per_device * ndev;
    p->n_workers = threads;
    p->file_threads = o.num_threads;
    if (const char* e = getenv("ZJ_POOL_LEND")) p->lend_idle = !(!strcmp(e, "off") || !strcmp(e, "0"));
    for (int k = 0; k < ndev; k++) { p->slots.emplace_back(); p->slots.back().device = devices[k]; }
    *status = ZJ_OK;
    if (const char* e = getenv("ZJ_POOL_BATCH")) { const int v = atoi(e); if (v >= 1 && v <= ZJ_SCAN_BATCH_MAX) p->device_batch = v; }
    int submitters = GPU_SUBMITTERS;
    if (const char* e = getenv("ZJ_POOL_SUBMITTERS")) { const int v = atoi(e); if (v >= 1 && v <= 64) submitters = v; }
    std::vector<int> ctx_slot;
    for (int k = 0; k < ndev && *status == ZJ_OK; k++)
        for (int g = 0; g < submitters && *status == ZJ_OK; g++) {
            int st = ZJ_OK;
            zj_ctx* c = zj_ctx_create(ZJ_BACKEND_HIP, devices[k], &st);
            if (!c) { *status = st ? st : ZJ_ERR_NOMEM; break; }
            zj_set_pipeline(c, 0); // one unit per file: the overlap comes from the other submitters
            p->ctxs.push_back(c);
            ctx_slot.push_back(k);
        }
    // plane sets, per slot: one per entropy worker plus what the submitters hold plus one in the queue each
    // (with the device entropy stage a plane set is a few MB of prepared scan, and a submitter may hold a batch of them).
    // They are empty shells here: the planes are pinned by the worker that first fills them, bound to the slot's device.
    for (int k = 0; k < ndev && *status == ZJ_OK; k++)
        for (int q = 0; q < threads_per_device + (o.entropy ? submitters * p->device_batch : 2 * submitters) && *status == ZJ_OK; q++) {
            zj_decoder* d = zj_decoder_new(&o);
            if (!d) { *status = ZJ_ERR_NOMEM; break; }
            p->decoders.push_back(d);
            p->slots[(size_t)k].free_dec.push_back(d);
        }
    if (*status != ZJ_OK) { zj_pool_destroy(p); return nullptr; }
    p->expected_threads = threads + (int)p->ctxs.size();
    for (int t = 0; t < threads; t++) { const int k = t % ndev; p->threads.emplace_back([p, k] { p->entropy_loop(k); }); }
    for (size_t g = 0; g < p->ctxs.size(); g++) { zj_ctx* c = p->ctxs[g]; const int k = ctx_slot[g]; p->threads.emplace_back([p, c, k] { p->gpu_loop(c, k); }); }
    {   // every thread has placed itself before the first batch (and before zj_pool_slot_numa is asked)
        std::unique_lock<std::mutex> lk(p->mu);
        p->cv_done.wait(lk, [&] { return p->settled == p->expected_threads; });
    }
    return p;
}

zj_pool* zj_pool_create(int device, int threads, const zj_options* opt, int* status)
{
    return zj_pool_create_multi(&device, 1, threads, opt, status);
}

int zj_pool_devices(const zj_pool* p) { return p ? (int)p->slots.size() : 0; }

int zj_pool_device_stats(zj_pool* p, int slot, int* device, double* gpu_seconds, size_t* files)
{
    if (!p || slot < 0 || slot >= (int)p->slots.size()) return ZJ_ERR_ARG;
    std::lock_guard<std::mutex> lk(p->mu);
    const zj_pool::Slot& sl = p->slots[(size_t)slot];
    if (device) *device = sl.device;
    if (gpu_seconds) *gpu_seconds = sl.gpu_s;
    if (files) *files = sl.files;
    return ZJ_OK;
}

int zj_pool_slot_numa(zj_pool* p, int slot, int* device_node, int* threads_bound, int* threads)
{
    if (!p || slot < 0 || slot >= (int)p->slots.size()) return ZJ_ERR_ARG;
    std::lock_guard<std::mutex> lk(p->mu);
    const zj_pool::Slot& sl = p->slots[(size_t)slot];
    if (device_node) *device_node = sl.node;
    if (threads_bound) *threads_bound = sl.n_bound;
    if (threads) *threads = sl.n_threads;
    return ZJ_OK;
}

int zj_pool_threads(const zj_pool* p) { return p ? p->n_workers : 0; }

const char* zj_pool_error(const zj_pool* p) { return p ? p->last_error.c_str() : ""; }

int zj_pool_stats(zj_pool* p, double* entropy_seconds, double* gpu_seconds, size_t* files)
{
    if (!p) return ZJ_ERR_ARG;
    std::lock_guard<std::mutex> lk(p->mu);
    if (entropy_seconds) *entropy_seconds = p->entropy_s;
    if (gpu_seconds) *gpu_seconds = p->gpu_s;
    if (files) *files = p->files_done;
    return ZJ_OK;
}

static int pool_decode(zj_pool* p, size_t nfiles, const uint8_t* const* bufs, const size_t* lens, uint8_t* const* outs,
                       const size_t* out_caps, size_t* out_lens, zj_image_info* infos, int* statuses, bool on_device);

int zj_pool_decode_files(zj_pool* p, size_t nfiles, const uint8_t* const* bufs, const size_t* lens,
                         uint8_t* const* outs, const size_t* out_caps, size_t* out_lens, zj_image_info* infos,
                         int* statuses)
{
    return pool_decode(p, nfiles, bufs, lens, outs, out_caps, out_lens, infos, statuses, false);
}

int zj_pool_decode_files_device(zj_pool* p, size_t nfiles, const uint8_t* const* bufs, const size_t* lens,
                                uint8_t* const* d_outs, const size_t* out_caps, size_t* out_lens, zj_image_info* infos,
                                int* statuses)
{
    return pool_decode(p, nfiles, bufs, lens, d_outs, out_caps, out_lens, infos, statuses, true);
}

static int pool_decode(zj_pool* p, size_t nfiles, const uint8_t* const* bufs, const size_t* lens, uint8_t* const* outs,
                       const size_t* out_caps, size_t* out_lens, zj_image_info* infos, int* statuses, bool on_device)
{
    if (!p || (nfiles && (!bufs || !lens || !outs || !out_caps))) return ZJ_ERR_ARG;
    if (nfiles == 0) return ZJ_OK;
    for (size_t i = 0; i < nfiles; i++)
        if (!bufs[i] || !outs[i]) return ZJ_ERR_ARG;
    std::lock_guard<std::mutex> call(p->call_mu);
    if (on_device) {
        // every output goes to the slot(s) of the device that owns it; the files of one device rotate over its slots
        const size_t ns = p->slots.size();
        std::vector<size_t> turn(ns, 0); // indexed by the first slot of a device
        p->out_slot.assign(nfiles, 0);
        for (size_t i = 0; i < nfiles && ns > 1; i++) {
            const int dev = zj_pointer_device(outs[i]);
            size_t first = ns, cand = 0;
            for (size_t k = 0; k < ns; k++)
                if (p->slots[k].device == dev) { if (first == ns) first = k; cand++; }
            if (!cand) { p->last_error = "file " + std::to_string(i) + ": the output pointer is not on a device of this pool"; return ZJ_ERR_ARG; }
            size_t nth = turn[first]++ % cand;
            for (size_t k = first; k < ns; k++)
                if (p->slots[k].device == dev && nth-- == 0) { p->out_slot[i] = (int)k; break; }
        }
    }
    zj_pool::Batch b;
    b.n = nfiles; b.bufs = bufs; b.lens = lens; b.outs = outs; b.caps = out_caps;
    b.out_lens = out_lens; b.infos = infos; b.statuses = statuses; b.on_device = on_device;
    std::unique_lock<std::mutex> lk(p->mu);
    for (auto& sl : p->slots) sl.todo.clear();
    if (on_device)
        for (size_t i = 0; i < nfiles; i++) p->slots[(size_t)p->out_slot[i]].todo.push_back(i);
    p->batch = &b;
    for (auto& sl : p->slots) sl.cv_work.notify_all();
    p->cv_done.wait(lk, [&] { return b.done == b.n; });
    p->batch = nullptr;
    return b.first_error;
}

} // extern "C"
