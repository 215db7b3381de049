// zj_multi.cpp -- image-level sharding of coefficient-plane batches across the GPUs of one node, inside the library.
//
// Frames are independent (the reference's own unit of independence is smaller still: the strip, src/mcu.rs:225-226,
// 356-368), so a batch of N frames over D devices is D contiguous shards [lo, hi) -- sizes differ by at most one -- and
// no collective: slot k's persistent host thread runs its shard through its own zj_ctx (three streams: uploads, kernels,
// downloads overlapped, zj_decode_planes_batch / zj_decode_frames), all slots at once.  SURVEY.md 8e: "one host thread
// (or process) per GPU"; this is the thread form for a C / Rust caller, bench.py's ranks are the process form.
//
// Only the C ABI of the library is used (include/zjhip.h).
#include <stdint.h>
#include <string.h>

#include <condition_variable>
#include <deque>
#include <mutex>
#include <new>
#include <thread>
#include <vector>

#include "../../include/zjhip.h"

struct zj_multi {
    struct Work {
        const zj_frame_desc* d = nullptr;
        size_t nframes = 0;
        // packed frames (base pointers) or independent allocations (pointer arrays)
        const int16_t* y = nullptr; const int16_t* cb = nullptr; const int16_t* cr = nullptr; uint8_t* out = nullptr;
        const int16_t* const* ys = nullptr; const int16_t* const* cbs = nullptr; const int16_t* const* crs = nullptr;
        uint8_t* const* outs = nullptr;
        int on_device = 0;
    };
    struct Slot {
        int device = 0;
        zj_ctx* ctx = nullptr;
        std::thread th;
        int rc = 0;          // of the slot's last shard
        size_t frames = 0;   // decoded since creation
        int node = -1, thread_node = -1, bound = 0; // NUMA: of the device, of the thread once it runs, whether it was bound (zj_numa.cpp)
        bool settled = false; // the thread has placed itself
    };
    std::deque<Slot> slots;
    std::mutex mu, call_mu;
    std::condition_variable cv_go, cv_done;
    Work work;
    unsigned long long epoch = 0; // bumped per call: every slot runs each epoch once
    int pending = 0;
    bool stop = false;

    void loop(int k)
    {
        Slot& me = slots[(size_t)k];
        (void)zj_set_thread_device(me.device);
        // the slot's host thread next to its GPU: it stages, submits and waits for that device's transfers
        const int node = zj_device_numa_node(me.device), bound = zj_bind_thread_near_device(me.device) >= 0 ? 1 : 0;
        const int here = zj_thread_numa_node();
        unsigned long long seen = 0;
        std::unique_lock<std::mutex> lk(mu);
        me.node = node; me.bound = bound; me.thread_node = here; me.settled = true;
        cv_done.notify_all(); // (zj_multi_create waits for every slot to have settled)
        for (;;) {
            cv_go.wait(lk, [&] { return stop || epoch != seen; });
            if (stop) return;
            seen = epoch;
            const Work w = work;
            lk.unlock();
            size_t lo = 0, hi = 0;
            zj_shard_range(w.nframes, k, (int)slots.size(), &lo, &hi);
            int rc = ZJ_OK;
            if (hi > lo) {
                const size_t n = hi - lo;
                if (w.ys) {
                    if (w.on_device) rc = zj_decode_frames_device(me.ctx, w.d, n, w.ys + lo, w.cbs ? w.cbs + lo : nullptr, w.crs ? w.crs + lo : nullptr, w.outs + lo, nullptr);
                    else rc = zj_decode_frames(me.ctx, w.d, n, w.ys + lo, w.cbs ? w.cbs + lo : nullptr, w.crs ? w.crs + lo : nullptr, w.outs + lo);
                    if (!rc && w.on_device) rc = zj_sync(me.ctx);
                } else {
                    const size_t yl = zj_plane_len(w.d, 0), cl = w.cb ? zj_plane_len(w.d, 1) : 0, ol = zj_out_len(w.d);
                    rc = zj_decode_planes_batch(me.ctx, w.d, n, w.y + lo * yl, w.cb ? w.cb + lo * cl : nullptr,
                                                w.cr ? w.cr + lo * cl : nullptr, w.out + lo * ol);
                }
            }
            lk.lock();
            me.rc = rc;
            if (!rc) me.frames += hi - lo;
            if (--pending == 0) cv_done.notify_all();
        }
    }

    int run(const Work& w, int* statuses)
    {
        std::lock_guard<std::mutex> call(call_mu);
        std::unique_lock<std::mutex> lk(mu);
        work = w;
        pending = (int)slots.size();
        epoch++;
        cv_go.notify_all();
        cv_done.wait(lk, [&] { return pending == 0; });
        int first = ZJ_OK;
        for (size_t k = 0; k < slots.size(); k++) {
            if (statuses) statuses[k] = slots[k].rc;
            if (slots[k].rc && !first) first = slots[k].rc;
        }
        return first;
    }
};

extern "C" {

void zj_shard_range(size_t nframes, int slot, int nslots, size_t* lo, size_t* hi)
{
    size_t a = 0, b = 0;
    if (nslots > 0 && slot >= 0 && slot < nslots) {
        const size_t base = nframes / (size_t)nslots, rem = nframes % (size_t)nslots, s = (size_t)slot;
        a = s * base + (s < rem ? s : rem);
        b = a + base + (s < rem ? 1 : 0);
    }
    if (lo) *lo = a;
    if (hi) *hi = b;
}

void zj_multi_destroy(zj_multi* m)
{
    if (!m) return;
    {
        std::lock_guard<std::mutex> lk(m->mu);
        m->stop = true;
    }
    m->cv_go.notify_all();
    for (auto& sl : m->slots) if (sl.th.joinable()) sl.th.join();
    for (auto& sl : m->slots) if (sl.ctx) zj_ctx_destroy(sl.ctx);
    delete m;
}

zj_multi* zj_multi_create(const int* devices, int ndev, int* status)
{
    int dummy;
    if (!status) status = &dummy;
    if (!devices || ndev <= 0 || ndev > 64) { *status = ZJ_ERR_ARG; return nullptr; }
    zj_multi* m = new (std::nothrow) zj_multi();
    if (!m) { *status = ZJ_ERR_NOMEM; return nullptr; }
    *status = ZJ_OK;
    for (int k = 0; k < ndev; k++) {
        int st = ZJ_OK;
        zj_ctx* c = zj_ctx_create(ZJ_BACKEND_HIP, devices[k], &st);
        if (!c) { *status = st ? st : ZJ_ERR_NOMEM; break; }
        m->slots.emplace_back();
        m->slots.back().device = devices[k];
        m->slots.back().ctx = c;
    }
    if (*status != ZJ_OK) { zj_multi_destroy(m); return nullptr; }
    for (int k = 0; k < ndev; k++) m->slots[(size_t)k].th = std::thread([m, k] { m->loop(k); });
    {
        std::unique_lock<std::mutex> lk(m->mu);
        m->cv_done.wait(lk, [&] { for (auto& sl : m->slots) if (!sl.settled) return false; return true; });
    }
    return m;
}

int zj_multi_devices(const zj_multi* m) { return m ? (int)m->slots.size() : 0; }

zj_ctx* zj_multi_ctx(zj_multi* m, int slot) { return (m && slot >= 0 && slot < (int)m->slots.size()) ? m->slots[(size_t)slot].ctx : nullptr; }

int zj_multi_slot_stats(zj_multi* m, int slot, int* device, size_t* frames)
{
    if (!m || slot < 0 || slot >= (int)m->slots.size()) return ZJ_ERR_ARG;
    std::lock_guard<std::mutex> lk(m->mu);
    if (device) *device = m->slots[(size_t)slot].device;
    if (frames) *frames = m->slots[(size_t)slot].frames;
    return ZJ_OK;
}

int zj_multi_slot_numa(zj_multi* m, int slot, int* device_node, int* thread_node, int* bound)
{
    if (!m || slot < 0 || slot >= (int)m->slots.size()) return ZJ_ERR_ARG;
    std::lock_guard<std::mutex> lk(m->mu);
    const zj_multi::Slot& sl = m->slots[(size_t)slot];
    if (device_node) *device_node = sl.node;
    if (thread_node) *thread_node = sl.thread_node;
    if (bound) *bound = sl.bound;
    return ZJ_OK;
}

int zj_multi_decode_planes_batch(zj_multi* m, const zj_frame_desc* d, size_t nframes, const int16_t* y, const int16_t* cb,
                                 const int16_t* cr, uint8_t* out, int* statuses)
{
    if (!m || !d || !nframes || !y || !out) return ZJ_ERR_ARG;
    if (zj_plane_len(d, 0) == 0 || zj_out_len(d) == 0) return ZJ_ERR_ARG;
    zj_multi::Work w;
    w.d = d; w.nframes = nframes; w.y = y; w.cb = cb; w.cr = cr; w.out = out;
    return m->run(w, statuses);
}

static int multi_frames(zj_multi* m, const zj_frame_desc* d, size_t nframes, const int16_t* const* y, const int16_t* const* cb,
                        const int16_t* const* cr, uint8_t* const* out, int* statuses, int on_device)
{
    if (!m || !d || !nframes || !y || !out) return ZJ_ERR_ARG;
    zj_multi::Work w;
    w.d = d; w.nframes = nframes; w.ys = y; w.cbs = cb; w.crs = cr; w.outs = out; w.on_device = on_device;
    return m->run(w, statuses);
}

int zj_multi_decode_frames(zj_multi* m, const zj_frame_desc* d, size_t nframes, const int16_t* const* y, const int16_t* const* cb,
                           const int16_t* const* cr, uint8_t* const* out, int* statuses)
{
    return multi_frames(m, d, nframes, y, cb, cr, out, statuses, 0);
}

int zj_multi_decode_frames_device(zj_multi* m, const zj_frame_desc* d, size_t nframes, const int16_t* const* d_y,
                                  const int16_t* const* d_cb, const int16_t* const* d_cr, uint8_t* const* d_out, int* statuses)
{
    return multi_frames(m, d, nframes, d_y, d_cb, d_cr, d_out, statuses, 1);
}

} // extern "C"
