// workers.hpp — internal to the libraries built from csrc/: the host side's parallel regions on threads that are kept.
//
// Every threaded host stage has the same shape — fn(t) for t in [0, T) on T threads, then join — and there are a dozen of them in one
// core step (planning, the exception list's sort / unpack / expansion / verdicts, page prefaults, the FASTA scan).  A std::thread costs
// 20-30 us to start and they start one after the other: the 32 planning threads of a 131072-row alignment were 0.7 ms of a 2.5 ms
// stage before the last one ran.  Here the threads are started once per process and sleep on a condition variable between regions.
//
// One region at a time: a second caller (another Python thread inside the library, or a region opened from inside a region) finds the
// pool busy and starts its own threads as before.  A forked child (--batch-procs) has none of the parent's threads: the pool notices
// the new process id and starts over.  The threads are detached and the pool is never destroyed (no join at process exit).
#pragma once

#include <unistd.h>

#include <atomic>
#include <condition_variable>
#include <cstdlib>
#include <functional>
#include <mutex>
#include <new>
#include <thread>
#include <vector>

namespace mp {

class Workers {
    struct State {
        std::mutex m;
        std::condition_variable go, done;
        const std::function<void(int)> *fn = nullptr;
        int n_threads = 0;                     // helper threads alive (indices 1 .. n_threads of a region)
        int n_active = 0, remaining = 0;
        unsigned long gen = 0;
    };
    State *s = new State;                      // replaced (and the old one leaked, its mutex may be held by a thread that is gone) after a fork
    std::atomic<bool> busy{false};
    pid_t pid = getpid();

    static void helper(State *s, int index, unsigned long seen) {
        for (;;) {
            std::unique_lock<std::mutex> lk(s->m);
            s->go.wait(lk, [&] { return s->gen != seen; });
            seen = s->gen;
            if (index > s->n_active) continue;
            const std::function<void(int)> *f = s->fn;
            lk.unlock();
            (*f)(index);
            lk.lock();
            if (--s->remaining == 0) s->done.notify_one();
        }
    }

  public:
    static constexpr int kMaxThreads = 128;
    static Workers &get() {
        static Workers *w = new Workers;       // never destroyed
        return *w;
    }
    // fn(0) on the caller, fn(1 .. T - 1) on kept threads; false when the pool is taken (the caller runs the region its own way)
    bool run(int T, const std::function<void(int)> &fn) {
        if (T > kMaxThreads + 1) return false;
        bool expected = false;
        if (!busy.compare_exchange_strong(expected, true)) return false;
        if (pid != getpid()) {                 // a forked child
            s = new State;
            pid = getpid();
        }
        bool started = true;
        {
            std::unique_lock<std::mutex> lk(s->m);
            while (s->n_threads < T - 1) {
                try {
                    std::thread(helper, s, s->n_threads + 1, s->gen).detach();
                } catch (...) {                // no more threads to be had: the caller's own way decides what to do about that
                    started = false;
                    break;
                }
                s->n_threads++;
            }
            if (started) {
                s->fn = &fn;
                s->n_active = s->remaining = T - 1;
                s->gen++;
            }
        }
        if (!started) { busy.store(false); return false; }
        s->go.notify_all();
        fn(0);
        {
            std::unique_lock<std::mutex> lk(s->m);
            s->done.wait(lk, [&] { return s->remaining == 0; });
            s->fn = nullptr;
            s->n_active = 0;
        }
        busy.store(false);
        return true;
    }
};

// fn(t) for t in [0, T) on T threads (the caller is one of them), back when all are done.  fn must not throw.
template <typename Fn>
inline void run_on_threads(int T, Fn &&fn) {
    if (T <= 1) { fn(0); return; }
    const std::function<void(int)> f = [&](int t) { fn(t); };
    static const bool off = [] { const char *e = getenv("MP_HOST_POOL"); return e && e[0] == '0'; }();
    if (!off && Workers::get().run(T, f)) return;
    std::vector<std::thread> th;
    for (int t = 1; t < T; t++) th.emplace_back(f, t);
    f(0);
    for (auto &x : th) x.join();
}

}  // namespace mp
