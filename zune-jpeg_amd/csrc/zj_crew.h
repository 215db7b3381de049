// zj_crew.h -- the entropy front-end's helper threads (round 6; used by zj_jpeg.cpp only, no GPU code here).
#ifndef ZJ_CREW_H
#define ZJ_CREW_H

#include <unistd.h>

#include <immintrin.h>

#include <atomic>
#include <chrono>
#include <condition_variable>
#include <functional>
#include <mutex>
#include <new>
#include <thread>
#include <vector>

namespace zj {

// The decoder's helper threads (round 6).  Rounds 4-5 started std::threads for every parallel region -- plane zeroing, restart
// segments, and now the two passes of scan_baseline_parallel: three regions per file, and on the GPU hosts a clone() costs
// 30-60 us, paid one after the other by the thread that should be working (16 threads: the last helper started 0.6 ms late,
// twice, in a scan that takes 4 ms).  The crew is started once per decoder, at the first region that wants it, sleeps on a
// condition variable between regions and is joined by zj_decoder_destroy.  Helpers inherit the creating thread's affinity
// (zj_numa.cpp binds that one).
class Crew {
public:
    Crew() = default;
    Crew(const Crew&) = delete;
    Crew& operator=(const Crew&) = delete;
    ~Crew()
    {
        orphaned();
        { std::lock_guard<std::mutex> g(m_); stop_ = true; }
        wake_.notify_all();
        for (auto& t : th_) t.join();
    }
    // fn(i) for every i in [0, n) on at most `threads` threads, the caller among them; items are claimed one at a time
    template <class F> void each(int n, int threads, F fn)
    {
        if (threads > n) threads = n;
        if (threads <= 1) { for (int i = 0; i < n; i++) fn(i); return; }
        std::atomic<int> next{0};
        auto work = [&]() { for (int i = next.fetch_add(1); i < n; i = next.fetch_add(1)) fn(i); };
        const std::function<void()> job = std::cref(work);
        orphaned();
        {
            std::lock_guard<std::mutex> g(m_);
            if (th_.empty()) pid_ = getpid();
            while ((int)th_.size() < threads - 1) th_.emplace_back([this] { helper(); });
            job_ = &job; seats_ = threads - 1; gen_++;
            hint_.store(gen_, std::memory_order_relaxed);
        }
        wake_.notify_all();
        work();
        std::unique_lock<std::mutex> g(m_);
        job_ = nullptr; // (a helper that has not woken up yet finds nothing to do: every item is claimed)
        idle_.wait(g, [this] { return busy_ == 0; });
    }
private:
    // in the child of a fork() the helpers do not exist (only the forking thread does): forget them, start new ones on demand
    void orphaned()
    {
        if (th_.empty() || getpid() == pid_) return;
        // (no pthread call on the stale handles -- the child's own new threads may already live in those stacks: the handles
        // are moved to a vector that is never destroyed)
        (void)new std::vector<std::thread>(std::move(th_));
        th_.clear();
        job_ = nullptr; seats_ = 0; busy_ = 0;
        // the condition variables still count the parent's sleeping helpers as waiters (destroying one would wait for them):
        // fresh ones in their place, without running the old ones' destructors
        new (&wake_) std::condition_variable();
        new (&idle_) std::condition_variable();
        new (&m_) std::mutex();
    }
    void helper()
    {
        unsigned long seen = 0;
        std::unique_lock<std::mutex> g(m_);
        for (;;) {
            wake_.wait(g, [&] { return stop_ || (job_ && gen_ != seen && seats_ > 0); });
            if (stop_) return;
            seen = gen_; seats_--; busy_++;
            const std::function<void()>* job = job_;
            g.unlock();
            (*job)();
            g.lock();
            if (--busy_ == 0) idle_.notify_all();
            // regions come in quick succession (pass A, a few microseconds of stitching, pass B): stay awake for a moment
            g.unlock();
            const auto t0 = std::chrono::steady_clock::now();
            while (hint_.load(std::memory_order_relaxed) == seen && std::chrono::steady_clock::now() - t0 < std::chrono::microseconds(80)) _mm_pause();
            g.lock();
        }
    }
    std::mutex m_;
    std::condition_variable wake_, idle_;
    std::vector<std::thread> th_;
    const std::function<void()>* job_ = nullptr;
    unsigned long gen_ = 0;
    std::atomic<unsigned long> hint_{0}; // gen_ again, for helpers that poll without the lock
    int seats_ = 0, busy_ = 0;
    bool stop_ = false;
    pid_t pid_ = 0; // the process the helpers belong to
};

} // namespace zj

#endif
