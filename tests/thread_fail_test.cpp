// thread_fail_test.cpp -- rsn_helpers.h without a device: the helper pool answers "none" when the system refuses a thread (run under
// tests/pthread_fail_shim.c), side jobs then run on the caller, helpers are reused, linger and leave; exceptions become codes.
#include "rsn_helpers.h"

#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <dlfcn.h>

using namespace rsn;
#define CHECK(c) do { if (!(c)) { fprintf(stderr, "FAILED %s:%d: %s\n", __FILE__, __LINE__, #c); exit(1); } } while (0)

int main() {
    auto fail_threads = (void (*)(int))dlsym(RTLD_DEFAULT, "rsn_test_fail_threads");
    auto refused = (long (*)(void))dlsym(RTLD_DEFAULT, "rsn_test_threads_refused");
    CHECK(fail_threads && refused);                                           // (the shim is preloaded)
    std::atomic<int> ran{0};

    fail_threads(1);
    CHECK(HelperPool::run(0, [&] { ran++; }) == nullptr && ran == 0);         // no thread: an answer, not std::terminate; the job has not run
    CHECK(refused() == 1);
    { SideJob j([&] { ran += 10; }); CHECK(ran == 0); j.finish(); CHECK(ran == 10); j.finish(); CHECK(ran == 10); }   // on the caller's thread, once
    { SideJob j([&] { ran += 100; }); }                                       // never asked for and never started: dropped
    CHECK(ran == 10);

    fail_threads(0);
    HelperPool::Handle a = HelperPool::run(0, [&] { ran++; });
    CHECK(a); a->wait(); CHECK(ran == 11 && !a->threw());
    CHECK(HelperPool::created() == 1);
    for (int spin = 0; HelperPool::idle() != 1 && spin < 2000; spin++) std::this_thread::sleep_for(std::chrono::milliseconds(1));
    fail_threads(1);                                                          // a helper that exists needs no new thread
    HelperPool::Handle b = HelperPool::run(0, [&] { ran++; });
    CHECK(b); b->wait(); CHECK(ran == 12 && HelperPool::created() == 1 && refused() == 3);
    // a second job at the same time needs a second thread: refused while the first helper is busy
    std::mutex mu; std::condition_variable cv; bool go = false;
    for (int spin = 0; HelperPool::idle() != 1 && spin < 2000; spin++) std::this_thread::sleep_for(std::chrono::milliseconds(1));
    HelperPool::Handle c1 = HelperPool::run(0, [&] { std::unique_lock<std::mutex> lk(mu); cv.wait(lk, [&] { return go; }); });
    CHECK(c1 && HelperPool::run(0, [&] { ran += 1000; }) == nullptr);
    { std::lock_guard<std::mutex> lk(mu); go = true; }
    cv.notify_all();
    c1->wait();
    fail_threads(0);

    // a job that throws: reported through its handle, the helper lives on; a side job's exception surfaces on the caller's thread
    HelperPool::Handle t = HelperPool::run(0, [] { throw std::runtime_error("boom"); });
    CHECK(t); t->wait(); CHECK(t->threw() && t->what() == "boom");
    bool caught = false;
    try { SideJob j([] { throw std::bad_alloc(); }); j.finish(); } catch (const std::exception &) { caught = true; }
    CHECK(caught);

    // guarded_call: the codes of rsn.h
    auto rep = [](int code, const char *) { return code; };
    CHECK(guarded_call<int>([]() -> int { throw std::bad_alloc(); }, rep) == -5);
    CHECK(guarded_call<int>([]() -> int { throw std::system_error(std::make_error_code(std::errc::resource_unavailable_try_again)); }, rep) == -5);
    CHECK(guarded_call<int>([]() -> int { std::vector<int> v; v.reserve((size_t)-1 / 2); return 0; }, rep) == -5);
    CHECK(guarded_call<int>([]() -> int { throw std::logic_error("x"); }, rep) == -4);
    CHECK(guarded_call<long long>([]() -> long long { throw 7; }, rep) == -4);
    CHECK(guarded_call<int>([]() -> int { return 3; }, rep) == 3);

    // on_idle reaches every idle helper; helpers without work leave after the linger time
    for (int spin = 0; HelperPool::idle() < 1 && spin < 2000; spin++) std::this_thread::sleep_for(std::chrono::milliseconds(1));
    const size_t n_idle = HelperPool::idle();
    std::atomic<int> seen{0};
    HelperPool::on_idle([&] { seen++; });
    CHECK((size_t)seen.load() == n_idle && n_idle >= 1);
    HelperPool::set_linger_ms(20);
    HelperPool::on_idle([] {});                                               // (wakes them so that the new linger time applies)
    for (int spin = 0; HelperPool::idle() != 0 && spin < 5000; spin++) std::this_thread::sleep_for(std::chrono::milliseconds(1));
    CHECK(HelperPool::idle() == 0);
    HelperPool::Handle again = HelperPool::run(0, [&] { ran++; });
    CHECK(again); again->wait();
    printf("thread fail test ok\n");
    return 0;
}
