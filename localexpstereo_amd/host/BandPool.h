// BandPool.h -- a small persistent team of helper threads owned by the calling thread.
//
// The cuts of the coarsest layer (4-6 cells of ~400 x 400 nodes per lock-step) split three of their phases over row bands:
// the node load, the parallel first phase of the max-flow and the segment read-out (ExpansionMove.h, GridMaxFlow.h).  Creating
// and joining 7 std::threads costs ~0.3 ms per phase -- as much as a fifth of such a cut -- so every calling thread (an OpenMP
// worker of the lock-step) keeps its helpers alive between cuts and only wakes them.  run(n, f) executes f(0) ... f(n - 1), f(0)
// on the caller, and returns when all are done.  Helpers spin briefly for the next phase of the same cut before they sleep.
#pragma once

#include <algorithm>
#include <atomic>
#include <condition_variable>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <memory>
#include <mutex>
#include <thread>
#include <vector>

namespace les_host {

// CPUs this process may keep busy: the hardware threads, capped by a cgroup CPU-time quota (cpu.max "quota period"; the MI355X boxes show
// 256 hardware threads and grant 16 CPUs of time).  Read once.
inline int cpuBudget()
{
    static const int budget = [] {
        int hw = (int)std::thread::hardware_concurrency();
        if (hw <= 0) hw = 1;
        long long quota = -1, period = -1;
        if (FILE* f = fopen("/sys/fs/cgroup/cpu.max", "r")) {
            char q[32] = {0};
            if (fscanf(f, "%31s %lld", q, &period) == 2 && q[0] != 'm') quota = atoll(q);
            fclose(f);
        } else {
            if (FILE* g = fopen("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "r")) { if (fscanf(g, "%lld", &quota) != 1) quota = -1; fclose(g); }
            if (FILE* g = fopen("/sys/fs/cgroup/cpu/cpu.cfs_period_us", "r")) { if (fscanf(g, "%lld", &period) != 1) period = -1; fclose(g); }
        }
        if (quota > 0 && period > 0) hw = (int)std::min<long long>(hw, std::max<long long>(1, (quota + period - 1) / period));
        return hw;
    }();
    return budget;
}

class BandPool {
public:
    // How long idle helpers spin for the next phase before they sleep.  A lock-step whose cells x bands exceed the CPU budget sets it
    // low (spinning helpers would burn the quota the working threads need); the band COUNT never depends on the machine.
    static std::atomic<int>& spinLimit()
    {
        static std::atomic<int> limit{20000};
        return limit;
    }
    static BandPool& mine()
    {
        static thread_local BandPool pool;
        return pool;
    }
    // A second team of the calling thread for work whose items use mine() themselves: run() is not re-entrant, and item 0 of a team runs on the
    // calling thread -- a cell-per-thread team whose cells split their phases over row bands (ResidualCut.h from the tiled solver's hand-over)
    // must therefore not be the pool the bands run on.
    static BandPool& outer()
    {
        static thread_local BandPool pool;
        return pool;
    }
    template <class F>
    void run(int n, F&& f)
    {
        if (n <= 1) { f(0); return; }
        ensure(n - 1);
        ctx_ = (void*)&f;
        fn_ = [](void* c, int b) { (*(typename std::remove_reference<F>::type*)c)(b); };
        {
            std::lock_guard<std::mutex> lk(m_);
            n_ = n;
            remaining_.store(n - 1, std::memory_order_relaxed);
            gen_.fetch_add(1, std::memory_order_release);
        }
        cv_start_.notify_all();
        f(0);
        // the helpers of a cut finish within microseconds of each other: spin first, sleep only if one of them is late
        for (int spin = 0; spin < 4096 && remaining_.load(std::memory_order_acquire) != 0; spin++) cpu_relax();
        if (remaining_.load(std::memory_order_acquire) != 0) {
            std::unique_lock<std::mutex> lk(m_);
            cv_done_.wait(lk, [&] { return remaining_.load(std::memory_order_acquire) == 0; });
        }
    }
    ~BandPool()
    {
        {
            std::lock_guard<std::mutex> lk(m_);
            stop_ = true;
            gen_.fetch_add(1, std::memory_order_release);
        }
        cv_start_.notify_all();
        for (auto& t : threads_) t.join();
    }

private:
    BandPool() = default;
    static void cpu_relax()
    {
#if defined(__x86_64__) || defined(__i386__)
        __builtin_ia32_pause();
#endif
    }
    void ensure(int helpers)
    {
        while ((int)threads_.size() < helpers) {
            const int index = (int)threads_.size() + 1;
            const uint64_t seen = gen_.load(std::memory_order_acquire);
            threads_.emplace_back([this, index, seen] { loop(index, seen); });
        }
    }
    void loop(int index, uint64_t seen)
    {
        for (;;) {
            // wait for the next generation: spin for the back-to-back phases of one cut, then sleep
            bool fresh = false;
            const int limit = spinLimit().load(std::memory_order_relaxed);
            for (int spin = 0; spin < limit; spin++) {
                if (gen_.load(std::memory_order_acquire) != seen) { fresh = true; break; }
                cpu_relax();
            }
            if (!fresh) {
                std::unique_lock<std::mutex> lk(m_);
                cv_start_.wait(lk, [&] { return gen_.load(std::memory_order_acquire) != seen; });
            }
            int n;
            {
                std::lock_guard<std::mutex> lk(m_);            // (n_, fn_, ctx_ are published under the mutex)
                seen = gen_.load(std::memory_order_acquire);
                if (stop_) return;
                n = n_;
            }
            if (index < n) {
                fn_(ctx_, index);
                if (remaining_.fetch_sub(1, std::memory_order_acq_rel) == 1) {
                    std::lock_guard<std::mutex> lk(m_);
                    cv_done_.notify_one();
                }
            }
        }
    }

    std::vector<std::thread> threads_;
    std::mutex m_;
    std::condition_variable cv_start_, cv_done_;
    std::atomic<uint64_t> gen_{0};
    std::atomic<int> remaining_{0};
    int n_ = 0;
    bool stop_ = false;
    void (*fn_)(void*, int) = nullptr;
    void* ctx_ = nullptr;
};

}  // namespace les_host
