// rsn_helpers.h -- what keeps a C++ exception from crossing the C boundary, and the helper threads of the pipelined host calls.
// Plain C++ (no HIP): tests/thread_fail_test.cpp compiles it on a machine without a device.
//
// The reference's only recover() is engine.go:315-328: a panic under AsyncBenchmarkFile becomes a "failed" row of the table.  The cgo
// shim turns a negative return code into that panic; a C++ exception that leaves an extern "C" function is std::terminate and a dead Go
// process instead.  Two sources exist on this side of the boundary: an allocation that fails (std::bad_alloc / std::length_error from a
// vector or a string) and a thread that cannot be created (std::system_error from std::thread: a Go runtime under load can sit at the
// process's thread limit).  So
//   (1) every entry point of rsn.h runs inside guarded(): whatever is thrown becomes RSN_ERR_NOMEM / RSN_ERR_DEVICE + a message;
//   (2) helper threads are asked of ONE pool (HelperPool::run) that answers "none" instead of throwing, and the callers have a serial
//       form for that answer; a helper that has finished stays for the next call -- with its thread context (stream, scratch,
//       rsn_common.h) -- and leaves only after half a minute without work: a pipelined host call costs no thread or stream creation
//       once the first has run;
//   (3) what runs ON a helper is wrapped as well (HelperPool's loop): a job that throws reports through the handle it was started with.
#pragma once

#include <chrono>
#include <condition_variable>
#include <cstdio>
#include <exception>
#include <functional>
#include <memory>
#include <mutex>
#include <new>
#include <stdexcept>
#include <string>
#include <system_error>
#include <thread>
#include <vector>

namespace rsn {

// what an exception becomes at the boundary: (code, text) through `report`, which must not throw
template <class Ret, class F, class Report> Ret guarded_call(F &&f, Report &&report) noexcept {
    char msg[256];
    int code;
    try {
        return f();
    } catch (const std::bad_alloc &) {
        code = -5; snprintf(msg, sizeof msg, "out of host memory (std::bad_alloc)");
    } catch (const std::length_error &e) {
        code = -5; snprintf(msg, sizeof msg, "a size beyond what a host container takes (%s)", e.what());
    } catch (const std::system_error &e) {
        code = -5; snprintf(msg, sizeof msg, "a thread or lock could not be had from the system (%s)", e.what());
    } catch (const std::exception &e) {
        code = -4; snprintf(msg, sizeof msg, "internal error: %s", e.what());
    } catch (...) {
        code = -4; snprintf(msg, sizeof msg, "internal error: an exception of unknown type");
    }
    return (Ret)report(code, msg);
}

class HelperPool {
public:
    // one started job: wait() returns when it has run; threw() says it ended in an exception (the text: what())
    struct Task {
        std::mutex mu; std::condition_variable cv; bool done = false, thrown = false; std::string text;
        void wait() { std::unique_lock<std::mutex> lk(mu); cv.wait(lk, [&] { return done; }); }
        bool threw() { std::lock_guard<std::mutex> lk(mu); return thrown; }
        std::string what() { std::lock_guard<std::mutex> lk(mu); return text; }
    };
    using Handle = std::shared_ptr<Task>;

    // Starts `job` on a helper thread of its own (never queued behind another job: the stages of a pipeline wait for each other).
    // `device` is a hint: an idle helper whose context already lives on that device is taken first.
    // Returns nullptr -- and the job has NOT run -- when no thread is to be had; never throws.
    template <class F> static Handle run(int device, F &&callable) noexcept {
        try {
            std::function<void()> job(std::forward<F>(callable));            // (may allocate: inside the try)
            Pool &P = pool();
            Handle t = std::make_shared<Task>();
            std::unique_lock<std::mutex> lk(P.mu);
            Helper *h = nullptr;
            for (size_t i = P.idle.size(); i-- > 0;)
                if (P.idle[i]->device == device) { h = P.idle[i]; P.idle.erase(P.idle.begin() + (long)i); break; }
            if (!h && !P.idle.empty()) { h = P.idle.back(); P.idle.pop_back(); }
            if (h) {
                h->job = std::move(job); h->task = t; h->device = device; h->has = true;
                h->cv.notify_one();                                          // (under the lock: a helper may be gone -- job done, linger over -- once it is released)
                return t;
            }
            lk.unlock();
            std::unique_ptr<Helper> nh(new Helper());
            nh->job = std::move(job); nh->task = t; nh->device = device; nh->has = true;
            std::thread th(loop, nh.get());                                  // std::system_error when the process is at its thread limit
            th.detach();
            nh.release();                                                    // the thread owns it now
            { std::lock_guard<std::mutex> lk2(P.mu); P.created++; }
            return t;
        } catch (...) {
            return nullptr;
        }
    }
    static void wait(const Handle &h) { if (h) h->wait(); }
    // `callable` once on every helper that is idle right now, all at once; returns when they have run (rsn_trim: the helpers' contexts
    // hold scratch of their own).  Helpers at work are not waited for.
    template <class F> static void on_idle(F &&callable) noexcept {
        try {
            const std::function<void()> job(std::forward<F>(callable));
            Pool &P = pool();
            std::vector<Helper *> hs; std::vector<Handle> ts;
            {
                std::lock_guard<std::mutex> lk(P.mu);
                hs.reserve(P.idle.size()); ts.reserve(P.idle.size());
                while (ts.size() < P.idle.size()) ts.push_back(std::make_shared<Task>());      // (everything that may throw, before a helper leaves the list)
                std::vector<std::function<void()>> copies(P.idle.size(), job);
                hs.swap(P.idle);
                for (size_t i = 0; i < hs.size(); i++) { hs[i]->job = std::move(copies[i]); hs[i]->task = ts[i]; hs[i]->has = true; hs[i]->cv.notify_one(); }
            }
            for (auto &t : ts) t->wait();
        } catch (...) {}
    }
    // helper threads started so far in this process (tests: a second pipelined call starts none)
    static unsigned long long created() { Pool &P = pool(); std::lock_guard<std::mutex> lk(P.mu); return P.created; }
    static size_t idle() { Pool &P = pool(); std::lock_guard<std::mutex> lk(P.mu); return P.idle.size(); }
    // how long a helper without work stays (milliseconds; tests shorten it)
    static void set_linger_ms(long ms) { Pool &P = pool(); std::lock_guard<std::mutex> lk(P.mu); P.linger_ms = ms; }

private:
    struct Helper { int device = -1; std::function<void()> job; Handle task; bool has = false; std::condition_variable cv; };
    struct Pool { std::mutex mu; std::vector<Helper *> idle; unsigned long long created = 0; long linger_ms = 30000; };
    static Pool &pool() { static Pool *p = new Pool(); return *p; }           // never destroyed: helpers may still wait on it while the process exits

    static void loop(Helper *h) noexcept {
        Pool &P = pool();
        for (;;) {
            std::function<void()> job; Handle t;
            {
                std::unique_lock<std::mutex> lk(P.mu);
                while (!h->has) {
                    const long ms = P.linger_ms;
                    // (the system clock: pthread_cond_timedwait, which ThreadSanitizer follows; it does not see through wait_for's clockwait)
                    if (h->cv.wait_until(lk, std::chrono::system_clock::now() + std::chrono::milliseconds(ms)) == std::cv_status::timeout && !h->has) {
                        for (size_t i = 0; i < P.idle.size(); i++)
                            if (P.idle[i] == h) { P.idle.erase(P.idle.begin() + (long)i); lk.unlock(); delete h; return; }   // (the thread's context parks itself, Ctx::~Ctx)
                    }
                }
                job = std::move(h->job); t = std::move(h->task); h->job = nullptr; h->task = nullptr; h->has = false;
            }
            bool thrown = false; std::string text;
            try { job(); }
            catch (const std::exception &e) { thrown = true; try { text = e.what(); } catch (...) {} }
            catch (...) { thrown = true; }
            job = nullptr;                                                   // (captures go before the waiter is told)
            { std::lock_guard<std::mutex> lk(t->mu); t->done = true; t->thrown = thrown; t->text.swap(text); }
            t->cv.notify_all();
            t.reset();
            bool parked = false;
            try { std::lock_guard<std::mutex> lk(P.mu); P.idle.push_back(h); parked = true; } catch (...) {}
            if (!parked) { delete h; return; }
        }
    }
};

// A piece of host work beside the caller's own (the header's text beside the tree, the lookup tables beside the code lengths): on a
// helper when one is to be had, else on the caller's thread when it asks for the result.  The destructor waits: the job works on the
// caller's frame, which an exception on the caller's side must not unwind under it.
class SideJob {
    std::function<void()> fn; HelperPool::Handle h; bool finished = false;
public:
    template <class F> explicit SideJob(F &&f) : fn(std::forward<F>(f)) { h = HelperPool::run(-1, [this] { fn(); }); }
    SideJob(const SideJob &) = delete;
    SideJob &operator=(const SideJob &) = delete;
    void finish() {
        if (finished) return;
        finished = true;
        if (!h) { fn(); return; }
        h->wait();
        if (h->threw()) throw std::runtime_error(h->what());                 // (on the caller's thread, where the entry point's guard sees it)
    }
    ~SideJob() { if (!finished && h) h->wait(); }
};

}  // namespace rsn
