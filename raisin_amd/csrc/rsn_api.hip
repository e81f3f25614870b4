// rsn_api.hip -- extern "C" surface of librsn (include/rsn.h) and the per-thread
// context.  Host-buffer entry points stage through device memory and call the
// device-resident codecs; there is no CPU compute path.
#include "codecs.h"
#include "rsn_common.h"
#include "rsn_helpers.h"

#include <sys/mman.h>

#include <algorithm>
#include <atomic>
#include <cmath>
#include <chrono>
#include <condition_variable>
#include <functional>
#include <mutex>
#include <thread>

namespace rsn {

namespace {
int env_int(const char *name, int dflt) { const char *e = getenv(name); return e && *e ? atoi(e) : dflt; }
std::atomic<size_t> g_arena_bytes[64];                                   // device scratch held by all contexts of the process, per device
}  // namespace

Ctx &ctx() {
    static thread_local Ctx c;
    return c;
}

// A thread that exits leaves its stream, scratch buffers and pinned staging behind for the next
// thread that needs a context on the same device: nothing leaks when callers come and go (a Go
// runtime retires OS threads), and no HIP call runs inside a thread-exit or process-exit destructor.
namespace {
struct Parked {
    int device; hipStream_t stream; Ctx::Buf bufs[Ctx::N_BUFS]; void *pinned; size_t pinned_cap; std::vector<hipEvent_t> events;
};
std::mutex g_park_mu;
std::vector<Parked> *g_parked = nullptr;      // heap-allocated and never destroyed: still valid while other threads unwind at exit

void release_parked(Parked &pk) {              // (HIP calls: only from a live thread, never from a thread-exit destructor)
    if (hipSetDevice(pk.device) != hipSuccess) return;
    for (auto &b : pk.bufs) if (b.p) { (void)hipFree(b.p); g_arena_bytes[pk.device & 63] -= b.cap; }
    if (pk.pinned) (void)hipHostFree(pk.pinned);
    for (auto e : pk.events) (void)hipEventDestroy(e);
    if (pk.stream) (void)hipStreamDestroy(pk.stream);
}

// A storm of short-lived caller threads (a Go runtime under load retires and creates OS threads) must not leave an unbounded
// number of parked scratch arenas in HBM: beyond RSN_MAX_PARKED the OLDEST ones are released by the next thread that initialises
// a context.
// (default: 8 plus three per visible device -- a batch call over G devices retires an uploader, an encoder and a downloader thread per device,
//  and the next batch wants their rings back)
size_t max_parked() {
    static const size_t v = [] {
        if (getenv("RSN_MAX_PARKED")) return (size_t)std::max(0, atoi(getenv("RSN_MAX_PARKED")));
        int cnt = 0;
        if (hipGetDeviceCount(&cnt) != hipSuccess || cnt < 0) cnt = 0;
        return (size_t)(8 + 3 * cnt);
    }();
    return v;
}

void trim_parked_excess(int restore_device) {
    std::vector<Parked> victims;
    {
        std::lock_guard<std::mutex> lk(g_park_mu);
        if (!g_parked) return;
        while (g_parked->size() > max_parked()) { victims.push_back(std::move(g_parked->front())); g_parked->erase(g_parked->begin()); }
    }
    if (victims.empty()) return;
    for (auto &pk : victims) release_parked(pk);
    (void)hipSetDevice(restore_device);
}

// The device of a thread that never called rsn_device_set(): 0, or RSN_DEVICE=<n>; RSN_DEVICE=rr hands the visible devices
// out round-robin, one per new thread context -- how a Go host whose goroutines land on different OS threads
// (runtime.LockOSThread in the shim) spreads its concurrent calls over the GPUs of a node without any API of its own.
std::atomic<unsigned> g_rr{0};
int default_device(int cnt) {
    const char *e = getenv("RSN_DEVICE");
    if (!e || !*e) return 0;
    if (!strcmp(e, "rr") || !strcmp(e, "all")) return (int)(g_rr.fetch_add(1) % (unsigned)cnt);
    return atoi(e);
}
}  // namespace

Ctx::~Ctx() {
    if (!inited) return;
    std::lock_guard<std::mutex> lk(g_park_mu);
    if (!g_parked) g_parked = new std::vector<Parked>();
    Parked pk;
    pk.device = device; pk.stream = own_stream; pk.pinned = pinned; pk.pinned_cap = pinned_cap;
    for (int i = 0; i < N_BUFS; i++) pk.bufs[i] = bufs[i];
    pk.events = free_events;
    for (auto &sl : slots) for (auto &pr : sl.pending) { pk.events.push_back(pr.first); pk.events.push_back(pr.second); }
    g_parked->push_back(std::move(pk));
}

int ctx_init(Ctx &c) {
    if (c.inited) {
        hipError_t e = hipSetDevice(c.device);
        if (e != hipSuccess) return c.fail(RSN_ERR_DEVICE, "hipSetDevice(%d): %s", c.device, hipGetErrorString(e));
        return RSN_OK;
    }
    int cnt = 0;
    hipError_t e = hipGetDeviceCount(&cnt);
    if (e != hipSuccess || cnt <= 0)
        return c.fail(RSN_ERR_DEVICE, "no HIP device available (%s); librsn has no CPU fallback", e != hipSuccess ? hipGetErrorString(e) : "device count 0");
    if (!c.device_chosen) { c.device = default_device(cnt); c.device_chosen = true; }
    if (c.device < 0 || c.device >= cnt) return c.fail(RSN_ERR_DEVICE, "device %d out of range (%d visible)", c.device, cnt);
    RSN_HIP(hipSetDevice(c.device));
    trim_parked_excess(c.device);
    {
        std::lock_guard<std::mutex> lk(g_park_mu);
        if (g_parked)
            for (size_t i = g_parked->size(); i-- > 0;)
                if ((*g_parked)[i].device == c.device) {                  // adopt what an earlier thread left behind (most recent first: its buffers are the warm ones)
                    Parked &pk = (*g_parked)[i];
                    if (c.pinned) (void)hipHostFree(c.pinned);              // a context that switched devices kept its staging: do not leak it
                    c.own_stream = pk.stream; c.pinned = pk.pinned; c.pinned_cap = pk.pinned_cap;
                    for (int k = 0; k < Ctx::N_BUFS; k++) c.bufs[k] = pk.bufs[k];
                    c.free_events = std::move(pk.events);
                    g_parked->erase(g_parked->begin() + (long)i);
                    c.inited = true;
                    return RSN_OK;
                }
    }
    RSN_HIP(hipStreamCreateWithFlags(&c.own_stream, hipStreamNonBlocking));
    c.inited = true;
    return RSN_OK;
}

// ---- admission of calls that need gigabytes of scratch (LZSS encode: ~12 bytes per position).  Every calling thread has an arena of
// its own, so 64 goroutines compressing 1 GiB files at once would ask for 64 x 12 GiB: instead of hipErrorOutOfMemory for most of
// them, a call states its need up front and WAITS while the needs of the calls in flight on its device add up to more than the
// device holds (RSN_SCRATCH_GIB, default 85 % of the device's memory); a call that finishes while others wait, or while the arenas
// on the device have grown past half of that, gives its large buffers back instead of keeping them for its thread's next call.
namespace {
struct ScratchGate { std::mutex mu; std::condition_variable cv; size_t cap = 0, in_flight = 0; int calls = 0, waiting = 0; unsigned long long queued = 0; };
ScratchGate g_gate[64];
}  // namespace

size_t scratch_admit(Ctx &c, size_t need) {
    if (c.gate_held || need == 0) return 0;                               // inside a host-buffer call that stated staging + codec scratch up front (host_call)
    ScratchGate &g = g_gate[c.device & 63];
    std::unique_lock<std::mutex> lk(g.mu);
    if (g.cap == 0) {
        const char *e = getenv("RSN_SCRATCH_GIB");
        size_t fr = 0, tot = 0;
        if (e && atof(e) > 0) g.cap = (size_t)(atof(e) * (double)((size_t)1 << 30));
        else if (hipMemGetInfo(&fr, &tot) == hipSuccess && tot) g.cap = tot / 100 * 85;
        else g.cap = (size_t)64 << 30;
    }
    if (g.calls > 0 && g.in_flight + need > g.cap) {                     // (a lone call always runs: its own allocation failing is the answer then)
        static const bool dbg = getenv("RSN_SCRATCH_DEBUG") != nullptr;
        g.waiting++; g.queued++;
        if (dbg) fprintf(stderr, "librsn: a call that needs %.2f GiB of scratch waits: %d in flight need %.2f of %.2f GiB\n", need / 1073741824.0, g.calls, g.in_flight / 1073741824.0, g.cap / 1073741824.0);
        g.cv.wait(lk, [&] { return g.calls == 0 || g.in_flight + need <= g.cap; });
        g.waiting--;
    }
    g.calls++; g.in_flight += need;
    c.gate_held = need;
    return need;
}

void scratch_release(Ctx &c, size_t need, unsigned long long slots) {
    if (need == 0) return;
    c.gate_held = 0;
    ScratchGate &g = g_gate[c.device & 63];
    bool give_back;
    {
        std::lock_guard<std::mutex> lk(g.mu);
        g.calls--; g.in_flight -= need;
        give_back = g.waiting > 0 || g_arena_bytes[c.device & 63].load() > g.cap / 2;
    }
    if (give_back && hipSetDevice(c.device) == hipSuccess) {
        for (int k = 0; k < Ctx::N_BUFS; k++) {
            if (!((slots >> k) & 1ull)) continue;
            Ctx::Buf &b = c.bufs[k];
            if (b.p && b.cap >= ((size_t)64 << 20)) { (void)hipFree(b.p); g_arena_bytes[c.device & 63] -= b.cap; b.p = nullptr; b.cap = 0; }
        }
    }
    g.cv.notify_all();
}

void scratch_forget(Ctx &c, size_t bytes) { g_arena_bytes[c.device & 63] -= bytes; }
unsigned long long scratch_queued(int device) { ScratchGate &g = g_gate[device & 63]; std::lock_guard<std::mutex> lk(g.mu); return g.queued; }

// the calling thread's scratch and staging back to the device / the host (rsn_trim; an idle helper when memory runs out)
static void trim_own_context() {
    Ctx &c = ctx();
    if (c.inited && hipSetDevice(c.device) == hipSuccess) {
        (void)hipStreamSynchronize(c.own_stream);
        for (auto &b : c.bufs) { if (b.p) { (void)hipFree(b.p); g_arena_bytes[c.device & 63] -= b.cap; } b.p = nullptr; b.cap = 0; }
        if (c.pinned) (void)hipHostFree(c.pinned);
        c.pinned = nullptr; c.pinned_cap = 0;
    }
}

int dev_buf(Ctx &c, int slot, size_t bytes, void **out) {
    Ctx::Buf &b = c.bufs[slot];
    if (bytes > b.cap) {
        if (b.p) { RSN_HIP(hipFree(b.p)); g_arena_bytes[c.device & 63] -= b.cap; b.p = nullptr; b.cap = 0; }
        const size_t want = round_up(bytes + bytes / 8, 4096);
        hipError_t e = hipMalloc(&b.p, want);
        if (e != hipSuccess) {                                            // what exited threads left parked on the device goes first, then once more
            (void)hipGetLastError();
            std::vector<Parked> victims;
            { std::lock_guard<std::mutex> lk(g_park_mu); if (g_parked) victims.swap(*g_parked); }
            for (auto &pk : victims) release_parked(pk);
            HelperPool::on_idle(trim_own_context);                        // (r06: idle helpers of the pipelined calls keep contexts of their own for half a minute)
            (void)hipSetDevice(c.device);
            e = hipMalloc(&b.p, want);
        }
        if (e != hipSuccess) { b.p = nullptr; return c.fail(RSN_ERR_NOMEM, "hipMalloc(%zu): %s", want, hipGetErrorString(e)); }
        b.cap = want;
        static std::atomic<unsigned long long> next_gen{0};
        b.gen = ++next_gen;
        g_arena_bytes[c.device & 63] += want;
    }
    *out = b.p;
    return RSN_OK;
}

int pinned_buf(Ctx &c, size_t bytes, void **out) {
    if (bytes > c.pinned_cap) {
        if (c.pinned) { RSN_HIP(hipHostFree(c.pinned)); c.pinned = nullptr; c.pinned_cap = 0; }
        hipError_t e = hipHostMalloc(&c.pinned, bytes, hipHostMallocDefault);
        if (e != hipSuccess) { c.pinned = nullptr; return c.fail(RSN_ERR_NOMEM, "hipHostMalloc(%zu): %s", bytes, hipGetErrorString(e)); }
        c.pinned_cap = bytes;
    }
    *out = c.pinned;
    return RSN_OK;
}

int func_dyn_lds(Ctx &c, const void *fn, size_t bytes) {
    static std::mutex mu;
    static std::vector<std::pair<std::pair<const void *, int>, size_t>> *seen = new std::vector<std::pair<std::pair<const void *, int>, size_t>>();
    std::lock_guard<std::mutex> lk(mu);
    for (auto &e : *seen)
        if (e.first.first == fn && e.first.second == c.device) {
            if (bytes <= e.second) return RSN_OK;
            RSN_HIP(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
            e.second = bytes;
            return RSN_OK;
        }
    RSN_HIP(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
    seen->push_back({{fn, c.device}, bytes});
    return RSN_OK;
}

void prof_collect(Ctx &c) {
    for (auto &sl : c.slots) {
        for (auto &pr : sl.pending) {
            float ms = 0;
            if (hipEventSynchronize(pr.second) == hipSuccess && hipEventElapsedTime(&ms, pr.first, pr.second) == hipSuccess) {
                sl.total_ms += ms;
                sl.launches++;
            }
            c.free_events.push_back(pr.first);
            c.free_events.push_back(pr.second);
        }
        sl.pending.clear();
    }
}

namespace {

// Result buffers.  A fresh 1 GiB malloc costs ~150 ms of page faults when the D2H copy first
// touches it -- several times the PCIe transfer itself -- so blocks of 1 MiB and more go back
// to a small process-wide pool on rsn_free() and the next call of similar size reuses pages
// that are already mapped (the cgo shim's pattern: call, C.GoBytes, rsn_free).  A 64-byte header
// in front of the returned pointer carries the capacity.
constexpr size_t RES_HDR = 64, POOL_MIN = 1u << 20;
constexpr size_t POOL_BLOCKS = 12;      // a batch of 8 chunks holds 8 results at once
struct ResHdr { unsigned long long magic, cap; };
constexpr unsigned long long RES_MAGIC = 0x52534E5F52455330ull;
std::mutex g_pool_mu;
std::vector<std::pair<size_t, void *>> g_pool;                    // (capacity, base)

bool pool_enabled() { return true; }

void *result_alloc(size_t n) {
    const size_t need = n ? n : 1;
    if (need >= POOL_MIN && pool_enabled()) {
        std::lock_guard<std::mutex> lk(g_pool_mu);
        size_t best = g_pool.size();
        for (size_t i = 0; i < g_pool.size(); i++)
            if (g_pool[i].first >= need && g_pool[i].first <= 2 * need + (16u << 20) && (best == g_pool.size() || g_pool[i].first < g_pool[best].first)) best = i;
        if (best != g_pool.size()) {
            void *base = g_pool[best].second;
            g_pool.erase(g_pool.begin() + (long)best);
            return (uint8_t *)base + RES_HDR;
        }
    }
    const bool big = need >= POOL_MIN;
    const size_t cap = big ? round_up(need + need / 16, (size_t)2 << 20) : need;   // headroom: the next result is rarely the same size
    void *base = nullptr;
    if (posix_memalign(&base, big ? ((size_t)2 << 20) : 64, cap + RES_HDR)) return nullptr;
    if (big) (void)madvise(base, cap + RES_HDR, MADV_HUGEPAGE);
    ResHdr *h = (ResHdr *)base;
    h->magic = RES_MAGIC; h->cap = cap;
    return (uint8_t *)base + RES_HDR;
}

void result_free(void *p) {
    if (!p) return;
    void *base = (uint8_t *)p - RES_HDR;
    const ResHdr *h = (const ResHdr *)base;
    if (h->magic != RES_MAGIC) { fprintf(stderr, "librsn: rsn_free() of a pointer librsn did not return\n"); return; }
    if (h->cap >= POOL_MIN && pool_enabled()) {
        std::lock_guard<std::mutex> lk(g_pool_mu);
        if (g_pool.size() < POOL_BLOCKS) { g_pool.emplace_back((size_t)h->cap, base); return; }
        size_t small = 0;                                              // keep the largest blocks
        for (size_t i = 1; i < g_pool.size(); i++) if (g_pool[i].first < g_pool[small].first) small = i;
        if (g_pool[small].first < h->cap) { void *victim = g_pool[small].second; g_pool[small] = {(size_t)h->cap, base}; base = victim; }
    }
    ((ResHdr *)base)->magic = 0;
    free(base);
}

// ---- the caller's INPUT memory, registered in place for the pipelined calls, under a process-wide table (r06).  The input is borrowed and
// read-only (rsn.h), so two goroutines may hand the SAME bytes to two calls at once -- decompressing one blob twice is legal -- and r05's
// per-call hipHostRegister / hipHostUnregister then failed in a way found by tests/test_gpu_host_pipeline.py::test_four_pipelined_calls_at_once:
// the second call's registration is refused (already registered), its copies run from memory the FIRST call has pinned, and the first call
// unregisters it under them ("pointer does not correspond to a registered memory region"; once an abort inside the runtime).  Here a range
// is registered once and counted: an identical range is shared, a call whose bytes merely OVERLAP registered ranges (a serial call's
// upload, another slicing of the same buffer) holds them for as long as it copies, and the last one out unregisters.
namespace {
struct PinRange { uintptr_t lo, hi; int refs; };
std::mutex g_pin_mu;
std::vector<PinRange> *g_pins = nullptr;                                 // (never destroyed: helpers may release at exit)
struct PinHold { std::vector<std::pair<uintptr_t, uintptr_t>> held; bool registered = false; };   // what one acquire took: release() gives it back

// [p, p + len): registers it (true in h.registered) unless something overlapping is registered already -- an identical range is shared
// (registered stays true: it IS pinned), any other overlap is held as it is (registered false: the copy takes what it finds)
void pin_acquire(const void *p, size_t len, bool want_register, PinHold &h) {
    if (!len) return;
    const uintptr_t lo = (uintptr_t)p, hi = lo + len;
    std::lock_guard<std::mutex> lk(g_pin_mu);
    if (!g_pins) g_pins = new std::vector<PinRange>();
    bool overlap = false;
    for (PinRange &r : *g_pins)
        if (r.lo < hi && lo < r.hi) {
            overlap = true;
            r.refs++;
            h.held.emplace_back(r.lo, r.hi);
            if (r.lo == lo && r.hi == hi) h.registered = true;
        }
    if (overlap || !want_register) return;
    if (hipHostRegister((void *)p, len, hipHostRegisterDefault) != hipSuccess) { (void)hipGetLastError(); return; }   // (memory some other library has registered, a mapping that cannot be pinned: copied as it is)
    g_pins->push_back(PinRange{lo, hi, 1});
    h.held.emplace_back(lo, hi);
    h.registered = true;
}
void pin_release(PinHold &h) {
    std::vector<uintptr_t> gone;
    {
        std::lock_guard<std::mutex> lk(g_pin_mu);
        if (g_pins)
            for (auto &pr : h.held)
                for (size_t i = 0; i < g_pins->size(); i++)
                    if ((*g_pins)[i].lo == pr.first && (*g_pins)[i].hi == pr.second) {
                        if (--(*g_pins)[i].refs == 0) { gone.push_back(pr.first); g_pins->erase(g_pins->begin() + (long)i); }
                        break;
                    }
    }
    h.held.clear(); h.registered = false;
    for (uintptr_t a : gone) if (hipHostUnregister((void *)a) != hipSuccess) (void)hipGetLastError();
}
struct PinGuard { PinHold h; ~PinGuard() { pin_release(h); } };         // a serial upload: holds what overlaps for as long as it copies
// host -> device of [src, src + len) whose pages may be PARTLY inside ranges other calls have registered (h.held, h.registered false): one
// copy command per stretch that lies wholly inside one registered range or wholly outside all of them -- a single command over memory of
// two kinds is decided by how its first byte is mapped
hipError_t copy_up(void *dst, const void *src, size_t len, hipStream_t st, const PinHold &h) {
    if (h.registered || h.held.empty()) return len ? hipMemcpyAsync(dst, src, len, hipMemcpyHostToDevice, st) : hipSuccess;
    const uintptr_t a = (uintptr_t)src, b = a + len;
    std::vector<uintptr_t> cuts{a, b};
    for (const auto &pr : h.held) { if (pr.first > a && pr.first < b) cuts.push_back(pr.first); if (pr.second > a && pr.second < b) cuts.push_back(pr.second); }
    std::sort(cuts.begin(), cuts.end());
    for (size_t i = 0; i + 1 < cuts.size(); i++) {
        if (cuts[i + 1] == cuts[i]) continue;
        const hipError_t e = hipMemcpyAsync((uint8_t *)dst + (cuts[i] - a), (const void *)cuts[i], cuts[i + 1] - cuts[i], hipMemcpyHostToDevice, st);
        if (e != hipSuccess) return e;
    }
    return hipSuccess;
}
}  // namespace

// Host-buffer wrapper: H2D, run `fn` on device buffers, D2H into a library-owned result.
// codec_need: what the codec itself will ask the gate for (its scratch besides the two staging buffers): stated HERE, before the
// staging is allocated -- a call that waited behind the gate inside the codec already held n + 1.125 x bound bytes of staging, uncounted
// (64 goroutines with 1 GiB each: 218 GiB next to the admitted 85 %), and the codec's own admission is covered by this one.
template <class Fn>
int host_call(const uint8_t *in, size_t n, uint8_t **out, size_t *out_n, size_t bound, size_t codec_need, Fn fn) {
    Ctx &c = ctx();
    if (!out || !out_n || (!in && n)) return c.fail(RSN_ERR_ARG, "null argument");
    *out = nullptr; *out_n = 0;
    int rc = ctx_init(c); if (rc) return rc;
    hipStream_t s = c.own_stream;
    struct Admitted {                                                  // released, with the large buffers when others wait, on every way out
        Ctx &c; size_t held;
        ~Admitted() { scratch_release(c, held, (3ull << 20) | (0xFFFull << 8) | (0x3Full << 22) | (3ull << 35)); }   // staging + the codecs' slots (8..19, 22..27, 35, 36)
    };
    const size_t total_need = round_up(n, 16) + 64 + bound + bound / 8 + codec_need;
    Admitted gate{c, total_need >= ((size_t)64 << 20) ? scratch_admit(c, total_need) : 0};
    static const bool timing = getenv("RSN_HOST_TIMING") != nullptr;
    auto stamp = [] { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    const double t_in = timing ? stamp() : 0;
    void *d_in, *d_out;
    rc = dev_buf(c, 20, round_up(n, 16) + 64, &d_in); if (rc) return rc;
    RSN_HIP(hipMemsetAsync((uint8_t *)d_in + (n & ~(size_t)15), 0, 64, s));
    {
        PinGuard g;                                                       // (another call may have these bytes registered: not released under this copy)
        pin_acquire(in, n, false, g.h);
        RSN_HIP(copy_up(d_in, in, n, s, g.h));
        if (n && !g.h.held.empty()) RSN_HIP(hipStreamSynchronize(s));
    }
    if (timing) RSN_HIP(hipStreamSynchronize(s));
    const double t_up = timing ? stamp() : 0;
    size_t cap = bound, got = 0;
    for (int attempt = 0;; attempt++) {
        rc = dev_buf(c, 21, cap, &d_out); if (rc) return rc;
        rc = fn(c, s, (const uint8_t *)d_in, (uint8_t *)d_out, cap, &got);
        if (rc == RSN_ERR_CAPACITY && attempt == 0 && got > cap) { cap = got; continue; }
        break;
    }
    if (rc) return rc;
    if (timing) RSN_HIP(hipStreamSynchronize(s));
    const double t_codec = timing ? stamp() : 0;
    uint8_t *res = (uint8_t *)result_alloc(got);
    if (!res) return c.fail(RSN_ERR_NOMEM, "allocating %zu result bytes failed", got);
    const double t_alloc = timing ? stamp() : 0;
    if (got) {
        RSN_HIP(hipMemcpyAsync(res, d_out, got, hipMemcpyDeviceToHost, s));
        RSN_HIP(hipStreamSynchronize(s));
    }
    if (timing)
        fprintf(stderr, "host call %p: in at %.2f ms, %zu B up by +%.2f, codec +%.2f, result block +%.2f, %zu B down +%.2f\n", (void *)&c,
                fmod(t_in, 100000.0), n, t_up - t_in, t_codec - t_up, t_alloc - t_codec, got, stamp() - t_alloc);
    *out = res; *out_n = got;
    return RSN_OK;
}

// ---- a host-buffer call as a PIPELINE: an uploader thread sends the input up in pieces, the calling thread runs the codec in slices as
// they land (`run` gets a SliceStream), a downloader thread brings every finished range of the output down while the next is produced.
// out_cap: what the output can take at most (the result block and the device buffer are that large); codec_need: the codec's scratch, for
// the admission.  `run` returns 1 when the stream turns out not to be for slicing: the caller then takes the serial call.
template <class Run>
static int piped_call(const uint8_t *in, size_t n, size_t out_cap, size_t codec_need, uint8_t **out, size_t *out_n, Run run) {
    // Pieces of 64 MiB, each REGISTERED (pinned in place, hipHostRegister: 0.65 ms for 64 MiB) before its copy is queued, the next
    // piece's registration under the copy of this one: copies of pageable memory in pieces ran at half the link's rate once the other
    // direction was busy too (measured: 27 ms up + the download behind it = the serial call's 36 ms; two 256 MiB copies at once: 94 GB/s).
    // The registrations are RELEASED when the codec has returned, not piece by piece: hipHostUnregister waits for every stream of the
    // process, the codec's kernels included (scripts/probes/copy_under_kernel.hip: 256 MiB up beside a 20 ms kernel in 4.8 ms when the
    // pieces stay registered, in 24-35 ms when each is released behind its copy -- the upload then only moves between kernels).
    constexpr size_t PIECE = (size_t)64 << 20;
    // [lo, hi) of a host range whose pages no neighbouring piece shares: piece k of a buffer at `base` is [cut(k), cut(k + 1))
    auto cut = [](const uint8_t *base, size_t total, size_t k) -> size_t {
        if (k == 0) return 0;
        const uintptr_t a = ((uintptr_t)base + k * PIECE + 4095) & ~(uintptr_t)4095;
        return std::min(total, (size_t)(a - (uintptr_t)base));
    };
    Ctx &c = ctx();
    int rc = ctx_init(c); if (rc) return rc;
    hipStream_t s = c.own_stream;
    struct Admitted { Ctx &c; size_t held; ~Admitted() { scratch_release(c, held, (3ull << 20) | (0xFFFull << 8) | (0x3Full << 22) | (3ull << 35)); } };
    Admitted gate{c, scratch_admit(c, round_up(n, 16) + 64 + out_cap + codec_need)};
    void *d_in, *d_out;
    rc = dev_buf(c, 20, round_up(n, 16) + 64, &d_in); if (rc) return rc;
    rc = dev_buf(c, 21, round_up(out_cap, 16) + 64, &d_out); if (rc) return rc;
    uint8_t *res = (uint8_t *)result_alloc(out_cap);
    if (!res) return c.fail(RSN_ERR_NOMEM, "allocating %zu result bytes failed", out_cap);
    RSN_HIP(hipMemsetAsync((uint8_t *)d_in + (n & ~(size_t)15), 0, 64, s));
    RSN_HIP(hipStreamSynchronize(s));                                     // (before the uploader's first piece lands on the same bytes)

    struct Pipe {
        std::mutex mu; std::condition_variable cv;
        size_t uploaded = 0;                                              // bytes of the stream on the device
        std::vector<std::pair<size_t, size_t>> ready;                     // decoded ranges not yet asked for by the downloader
        size_t taken = 0; bool decoded_all = false, failed = false; std::string msg;
        void fail(const char *m) noexcept { try { std::lock_guard<std::mutex> lk(mu); if (!failed) { failed = true; try { msg = m; } catch (...) {} } } catch (...) {} cv.notify_all(); }
    } P;
    // what runs on a helper: an exception there (a vector that cannot grow) fails the pipe at once -- the other stages wait on P
    auto stage = [&P](auto body) { return [&P, body] { guarded_call<int>([&] { body(); return 0; }, [&](int, const char *m) { P.fail(m); return 0; }); }; };
    const int device = c.device;
    static const bool timing = getenv("RSN_HOST_TIMING") != nullptr;
    const auto t_start = std::chrono::steady_clock::now();
    auto since = [&] { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_start).count(); };
    std::vector<PinHold> in_pinned((n + PIECE - 1) / PIECE + 2);         // the input's pieces: registered or shared through the table above (released below)
    // Both helpers come from the pool (rsn_helpers.h): no thread and no stream is created once a first call has run, and "no thread to be
    // had" is an answer, not an exception -- the caller then takes the serial call.
    HelperPool::Handle uploader = HelperPool::run(device, stage([&] {
        if (rsn_device_set(device) != RSN_OK) { P.fail(rsn_last_error()); return; }
        hipStream_t su = ctx().own_stream;
        std::vector<PinHold> &pinned = in_pinned;
        auto pin = [&](size_t k) { const size_t lo = cut(in, n, k), hi = cut(in, n, k + 1); if (hi > lo) pin_acquire(in + lo, hi - lo, true, pinned[k]); };
        pin(0);
        for (size_t k = 0; cut(in, n, k) < n; k++) {
            const size_t lo = cut(in, n, k), hi = cut(in, n, k + 1);
            hipError_t e = copy_up((uint8_t *)d_in + lo, in + lo, hi - lo, su, pinned[k]);
            if (e == hipSuccess && hi < n) pin(k + 1);                        // (under this piece's copy)
            if (e == hipSuccess) e = hipStreamSynchronize(su);
            if (e != hipSuccess) { P.fail(hipGetErrorString(e)); return; }
            { std::lock_guard<std::mutex> lk(P.mu); if (P.failed) return; P.uploaded = hi; }
            P.cv.notify_all();
        }
        if (timing) { size_t np_ = 0; for (const PinHold &x : pinned) np_ += x.registered; fprintf(stderr, "piped call: %zu B up by +%.2f ms (%zu pieces registered)\n", n, since(), np_); }
    }));
    if (!uploader) { result_free(res); return 1; }
    HelperPool::Handle downloader = HelperPool::run(device, stage([&] {
        if (rsn_device_set(device) != RSN_OK) { P.fail(rsn_last_error()); return; }
        hipStream_t sd = ctx().own_stream;
        const size_t total_out = out_cap;
        std::vector<char> pinned((total_out + PIECE - 1) / PIECE + 1, 0);
        size_t next_pin = 0;                                              // pieces of the result block below this index have been tried
        for (;;) {
            std::pair<size_t, size_t> r;
            {
                std::unique_lock<std::mutex> lk(P.mu);
                P.cv.wait(lk, [&] { return P.failed || P.taken < P.ready.size() || P.decoded_all; });
                if (P.failed || P.taken >= P.ready.size()) break;
                r = P.ready[P.taken++];
            }
            if (!r.second) continue;
            auto pin_below = [&](size_t limit) {                              // the pieces of the result block that begin below `limit`
                while (cut(res, total_out, next_pin) < std::min(total_out, limit)) {
                    const size_t lo = cut(res, total_out, next_pin), hi = cut(res, total_out, next_pin + 1);
                    if (hi > lo) { pinned[next_pin] = hipHostRegister(res + lo, hi - lo, hipHostRegisterDefault) == hipSuccess; if (!pinned[next_pin]) (void)hipGetLastError(); }
                    next_pin++;
                }
            };
            pin_below(r.first + r.second);                                    // the pieces this range lands in (all but the first call's were pinned a range ahead)
            hipError_t e = hipSuccess;                                        // (a copy may not straddle two registrations: cut at the pieces' edges)
            for (size_t at = r.first, end = r.first + r.second; at < end && e == hipSuccess;) {
                size_t k = at / PIECE;                                        // the piece that holds `at`: cut(k) <= at < cut(k + 1)
                while (cut(res, total_out, k + 1) <= at) k++;
                while (k > 0 && cut(res, total_out, k) > at) k--;
                const size_t stop = std::min(end, cut(res, total_out, k + 1));
                e = hipMemcpyAsync(res + at, (const uint8_t *)d_out + at, stop - at, hipMemcpyDeviceToHost, sd);
                at = stop;
            }
            if (e == hipSuccess) pin_below(r.first + 2 * r.second);            // under these copies: where the next range will land
            if (e == hipSuccess) e = hipStreamSynchronize(sd);
            if (e != hipSuccess) { P.fail(hipGetErrorString(e)); break; }
            if (timing) fprintf(stderr, "piped call: [%zu, +%zu) down by +%.2f ms\n", r.first, r.second, since());
        }
        if (timing) { size_t np_ = 0; for (size_t k = 0; k < next_pin; k++) np_ += pinned[k] != 0; fprintf(stderr, "piped call: %zu of %zu pieces of the result registered\n", np_, next_pin); }
        for (size_t k = 0; k < next_pin; k++) if (pinned[k]) (void)hipHostUnregister(res + cut(res, total_out, k));
    }));
    if (!downloader) {                                                    // (the uploader is at work on P and d_in: it ends before they do)
        P.fail("no helper thread for the downloads");
        HelperPool::wait(uploader);
        (void)hipSetDevice(c.device);
        for (PinHold &x : in_pinned) pin_release(x);
        result_free(res);
        return 1;
    }
    SliceStream st;
    st.slice_bytes = (size_t)64 << 20;
    st.need_in = [&](size_t bytes) { std::unique_lock<std::mutex> lk(P.mu); P.cv.wait(lk, [&] { return P.failed || P.uploaded >= std::min(bytes, n); }); return !P.failed; };
    st.in_so_far = [&] { std::lock_guard<std::mutex> lk(P.mu); return P.uploaded; };
    st.have_out = [&](size_t off, size_t len) {
        if (timing) fprintf(stderr, "piped call: [%zu, +%zu) decoded by +%.2f ms\n", off, len, since());
        { std::lock_guard<std::mutex> lk(P.mu); if (P.failed) return false; P.ready.emplace_back(off, len); }
        P.cv.notify_all();
        return true;
    };
    size_t got = 0;
    // (an exception out of the codec -- a vector that cannot grow -- must not unwind this frame while the helpers work on it)
    rc = guarded_call<int>([&] { return run(c, s, (const uint8_t *)d_in, (uint8_t *)d_out, out_cap, &got, &st); },
                           [&](int code, const char *m) { try { c.err = m; } catch (...) {} return code; });
    if (rc != RSN_OK) P.fail(c.err.c_str());                              // (stops both helpers)
    { std::lock_guard<std::mutex> lk(P.mu); P.decoded_all = true; }
    P.cv.notify_all();
    HelperPool::wait(uploader);
    HelperPool::wait(downloader);
    (void)hipSetDevice(c.device);
    for (PinHold &x : in_pinned) pin_release(x);
    if (timing) fprintf(stderr, "piped call: the codec returned %d (%s) at +%.2f ms\n", rc, rc > 0 || rc == RSN_OK ? "" : c.err.c_str(), since());
    if (rc == RSN_ERR_CAPACITY || rc == 1) { result_free(res); return 1; }   // more output than the caller allowed for, or not a stream for slices: the serial call
    if (rc != RSN_OK && P.failed && P.msg != c.err) { result_free(res); return c.fail(rc, "%s (%s)", std::string(c.err).c_str(), P.msg.c_str()); }
    if (rc != RSN_OK) { result_free(res); return rc; }
    if (P.failed) { result_free(res); return c.fail(RSN_ERR_DEVICE, "a transfer of the pipelined call failed: %s", P.msg.c_str()); }
    *out = res; *out_n = got;
    return RSN_OK;
}

// the codecs read d_in while they write d_out: the two ranges must not overlap
bool ranges_overlap(const void *a, size_t na, const void *b, size_t nb) {
    if (!a || !b || !na || !nb) return false;
    const uintptr_t x = (uintptr_t)a, y = (uintptr_t)b;
    return x < y + nb && y < x + na;
}

int dev_prologue(Ctx &c, void *stream, hipStream_t *s) {
    int rc = ctx_init(c); if (rc) return rc;
    *s = stream ? (hipStream_t)stream : c.own_stream;
    return RSN_OK;
}

}  // namespace
}  // namespace rsn

using namespace rsn;

// (implementations; the extern "C" wrappers that guard them are at the end of the file)

static int rsn_device_set_impl(int device) {
    Ctx &c = ctx();
    if (device < 0) return c.fail(RSN_ERR_ARG, "negative device");
    if (c.inited && c.device != device) {
        // drop per-device resources; they are re-created lazily on the new device
        (void)hipSetDevice(c.device);
        for (auto &b : c.bufs) { if (b.p) { (void)hipFree(b.p); g_arena_bytes[c.device & 63] -= b.cap; } b.p = nullptr; b.cap = 0; }
        if (c.own_stream) (void)hipStreamDestroy(c.own_stream);
        c.own_stream = nullptr;
        prof_collect(c);
        for (auto e : c.free_events) (void)hipEventDestroy(e);
        c.free_events.clear();
        c.inited = false;
    }
    c.device = device;
    c.device_chosen = true;
    return ctx_init(c);
}

static int rsn_device_count_impl(void) {
    int cnt = 0;
    hipError_t e = hipGetDeviceCount(&cnt);
    if (e != hipSuccess) return ctx().fail(RSN_ERR_DEVICE, "hipGetDeviceCount: %s", hipGetErrorString(e));
    return cnt;
}

const char *rsn_last_error(void) { return ctx().err.c_str(); }
const char *rsn_version(void) { return "librsn 0.1 (gfx950)"; }

static void rsn_trim_impl(void) {
    Ctx &c = ctx();
    trim_own_context();
    HelperPool::on_idle(trim_own_context);                                // the idle helpers of the pipelined calls keep contexts of their own
    std::vector<Parked> parked;
    {
        std::lock_guard<std::mutex> lk(g_park_mu);
        if (g_parked) parked.swap(*g_parked);
    }
    for (auto &pk : parked) release_parked(pk);
    if (c.inited) (void)hipSetDevice(c.device);
    std::vector<std::pair<size_t, void *>> pool;
    {
        std::lock_guard<std::mutex> lk(g_pool_mu);
        pool.swap(g_pool);
    }
    for (auto &pr : pool) { ((ResHdr *)pr.second)->magic = 0; free(pr.second); }
}
static void rsn_free_impl(void *p) { result_free(p); }

size_t rsn_huffman_compress_bound(size_t n) { return huff_compress_bound(n); }
size_t rsn_lzss_compress_bound(size_t n) { return lzss_compress_bound(n); }

static int rsn_huffman_compress_dev_impl(const void *d_in, size_t n, void *d_out, size_t out_cap, size_t *out_n, void *stream) {
    Ctx &c = ctx(); hipStream_t s;
    if (!d_in || !d_out || !out_n) return c.fail(RSN_ERR_ARG, "null argument");
    if (ranges_overlap(d_in, n, d_out, out_cap)) return c.fail(RSN_ERR_ARG, "input and output ranges overlap");
    int rc = dev_prologue(c, stream, &s); if (rc) return rc;
    return huff_encode_dev(c, s, (const uint8_t *)d_in, n, (uint8_t *)d_out, out_cap, out_n, nullptr, nullptr);
}

static int rsn_huffman_decompress_dev_impl(const void *d_in, size_t n, void *d_out, size_t out_cap, size_t *out_n, void *stream) {
    Ctx &c = ctx(); hipStream_t s;
    if (!d_in || !out_n) return c.fail(RSN_ERR_ARG, "null argument");
    if (ranges_overlap(d_in, n, d_out, out_cap)) return c.fail(RSN_ERR_ARG, "input and output ranges overlap");
    int rc = dev_prologue(c, stream, &s); if (rc) return rc;
    return huff_decode_dev(c, s, (const uint8_t *)d_in, n, (uint8_t *)d_out, out_cap, out_n);
}

static int rsn_lzss_compress_dev_impl(const void *d_in, size_t n, int64_t window, void *d_out, size_t out_cap, size_t *out_n, void *stream) {
    Ctx &c = ctx(); hipStream_t s;
    if ((!d_in && n) || !d_out || !out_n) return c.fail(RSN_ERR_ARG, "null argument");
    if (ranges_overlap(d_in, n, d_out, out_cap)) return c.fail(RSN_ERR_ARG, "input and output ranges overlap");
    int rc = dev_prologue(c, stream, &s); if (rc) return rc;
    return lzss_encode_dev(c, s, (const uint8_t *)d_in, n, window, (uint8_t *)d_out, out_cap, out_n);
}

static int rsn_lzss_decompress_dev_impl(const void *d_in, size_t n, void *d_out, size_t out_cap, size_t *out_n, void *stream) {
    Ctx &c = ctx(); hipStream_t s;
    if ((!d_in && n) || !out_n) return c.fail(RSN_ERR_ARG, "null argument");
    if (ranges_overlap(d_in, n, d_out, out_cap)) return c.fail(RSN_ERR_ARG, "input and output ranges overlap");
    int rc = dev_prologue(c, stream, &s); if (rc) return rc;
    return lzss_decode_dev(c, s, (const uint8_t *)d_in, n, (uint8_t *)d_out, out_cap, out_n);
}

static int huffman_compress_single(const uint8_t *in, size_t n, uint8_t **out, size_t *out_n) {
    // typical outputs are < n; the exact need is reported back on RSN_ERR_CAPACITY
    return host_call(in, n, out, out_n, n + n / 8 + (1 << 16), n / 32,
                     [n](Ctx &c, hipStream_t s, const uint8_t *di, uint8_t *dout, size_t cap, size_t *got) {
                         return huff_encode_dev(c, s, di, n, dout, cap, got, nullptr, nullptr);
                     });
}

// the small-input codec's result (in pinned staging) into a block of the caller's; 1 stays 1
static int small_result(Ctx &c, int rc, const uint8_t *p, size_t got, uint8_t **out, size_t *out_n) {
    if (rc != RSN_OK) return rc;
    uint8_t *res = (uint8_t *)result_alloc(got);
    if (!res) return c.fail(RSN_ERR_NOMEM, "allocating %zu result bytes failed", got);
    memcpy(res, p, got);
    *out = res; *out_n = got;
    return RSN_OK;
}

static int rsn_huffman_compress_impl(const uint8_t *in, size_t n, uint8_t **out, size_t *out_n) {
    if (n == 0) return ctx().fail(RSN_ERR_EMPTY, "huffman: empty input (reference panics in heap.Pop, huffman.go:102)");
    if (in && out && out_n && n <= 65536) {                              // (the general path for the same bytes: the device-pointer entry points, tests/test_gpu_huffman_small.py)
        Ctx &c = ctx(); const uint8_t *p = nullptr; size_t got = 0;
        const int rc = small_result(c, huff_small_compress(c, in, n, &p, &got), p, got, out, out_n);
        if (rc != 1) return rc;
    }
    // RSN_HUFF_SHARDS=<G> > 1: one stream out of G slices, a worker (and, with RSN_BATCH_DEVICES, a device) each -- same bytes
    static const int env_shards = env_int("RSN_HUFF_SHARDS", 0);
    if (env_shards > 1 && n >= ((size_t)1 << 16)) return rsn_huffman_compress_sharded(in, n, env_shards, out, out_n);
    return huffman_compress_single(in, n, out, out_n);
}

static int huffman_decompress_serial(const uint8_t *in, size_t n, uint8_t **out, size_t *out_n) {
    return host_call(in, n, out, out_n, 4 * n + (1 << 16), n / 8,
                     [n](Ctx &c, hipStream_t s, const uint8_t *di, uint8_t *dout, size_t cap, size_t *got) {
                         return huff_decode_dev(c, s, di, n, dout, cap, got);
                     });
}

// A large host-buffer decode as a PIPELINE (what the cgo shim binds was upload, then codec, then download -- 16 + 0.5 + 19 ms
// for a GiB of 2a, the codec the smallest part).  The header is on the host already: its counts give the output's size (what a stream this
// library or the reference wrote decodes to), so the result block and both device buffers exist before a byte has moved; an uploader thread
// sends the stream up in pieces, this thread decodes slice after slice as the pieces land (huff_decode_dev with a SliceStream: a slice
// starts where its predecessor's last codeword ended), a downloader thread brings every finished slice down while the next decodes.  PCIe
// is full duplex: the call takes the longer of the two transfers, not their sum.  A stream whose payload decodes to more than its header
// says (a foreign header: the counts only shape the tree) is decoded again by the serial call.  ENCODE has nothing to overlap: the first
// output byte -- header counts, pad (huffman.go:245-255) -- depends on the last input byte.
// Returns 1 when the stream is not for this path (small, malformed, foreign): the caller takes the serial call, which also words the errors.
static int huffman_decompress_piped(const uint8_t *in, size_t n, uint8_t **out, size_t *out_n) {
    constexpr size_t PIPE_MIN = (size_t)32 << 20;
    if (n < PIPE_MIN) return 1;
    // ---- the header: strings.SplitN(content, "\\\n", 2) (huffman.go:261), the counts (huffman.go:196-227)
    size_t sep = (size_t)-1;
    for (size_t i = 0; i + 1 < std::min(n, (size_t)64 << 20); i++) if (in[i] == 0x5C && in[i + 1] == 0x0A) { sep = i; break; }
    if (sep == (size_t)-1 || sep + 3 > n) return 1;
    std::vector<HuffSym> syms; std::string msg;
    if (!parse_header(in, sep, syms, msg) || syms.size() < 2) return 1;
    const size_t sn = n - sep - 2;
    const unsigned long long nbits = (unsigned long long)(sn - 1) * 8, diff = in[sep + 2];
    if (diff > nbits || nbits == diff) return 1;
    const unsigned long long max_bits = nbits - diff;
    unsigned long long total = 0, expect = 0;
    for (const HuffSym &sy : syms) {
        if (sy.freq > max_bits - total) return 1;                         // more symbols than the payload has bits: a foreign header
        total += sy.freq;
        expect += sy.freq * (unsigned long long)utf8_len(sy.rune);
    }
    if (expect == 0) return 1;
    return piped_call(in, n, (size_t)expect, n / 8, out, out_n,
                      [n](Ctx &c, hipStream_t s, const uint8_t *di, uint8_t *dout, size_t cap, size_t *got, const SliceStream *st) {
                          return huff_decode_dev(c, s, di, n, dout, cap, got, st);
                      });
}

static int rsn_huffman_decompress_impl(const uint8_t *in, size_t n, uint8_t **out, size_t *out_n) {
    Ctx &c = ctx();
    if (!out || !out_n || (!in && n)) return c.fail(RSN_ERR_ARG, "null argument");
    *out = nullptr; *out_n = 0;
    static const bool serial = getenv("RSN_HOST_SERIAL") != nullptr;     // A/B switch (tests): upload, decode, download, one after the other
    if (n <= 65536 + 2048) {
        const uint8_t *p = nullptr; size_t got = 0;
        const int rc = small_result(c, huff_small_decompress(c, in, n, &p, &got), p, got, out, out_n);
        if (rc != 1) return rc;
    }
    if (!serial) {
        const int rc = huffman_decompress_piped(in, n, out, out_n);
        if (rc != 1) return rc;
    }
    return huffman_decompress_serial(in, n, out, out_n);
}

static int rsn_lzss_compress_impl(const uint8_t *in, size_t n, int64_t window, uint8_t **out, size_t *out_n) {
    // From 128 MiB up, with the engine's kind of window: the pipeline (piped_call) -- sections of a quarter of the input (at most 256 MiB),
    // each encoded as soon as its bytes are up, its tokens on their way down while the next is encoded.  The upload (19 ms per GiB) and the
    // download (12) disappear under the encoder's 31; what does not: the first section's upload and the last one's download.  For inputs in
    // which nothing needs an escape (so the output is at most the input): the encoder checks the input as it lands and lets the first output
    // byte go only when it has seen the last input byte (the upload is done well before the encoder is); an input with a 5C or an FF in it,
    // and every call under RSN_HOST_SERIAL=1, takes the serial call.
    if (in && out && out_n && n <= 1024) {                                // (the reference's own table is files of 13-25 bytes, README.md:153-167: one launch, lzss_small.hip)
        Ctx &c = ctx(); const uint8_t *p = nullptr; size_t got = 0;
        *out = nullptr; *out_n = 0;
        const int rc = small_result(c, lzss_small_compress(c, in, n, window, &p, &got), p, got, out, out_n);
        if (rc != 1) return rc;
    }
    static const bool serial = getenv("RSN_HOST_SERIAL") != nullptr;
    if (!serial && in && out && out_n && n >= ((size_t)128 << 20) && window > 0 && window <= 4096) {
        const size_t sec = std::min((size_t)256 << 20, n / 4);            // (eight sections instead of four: the same 46 ms per GiB -- half a millisecond of fixed cost per section)
        const int rc = piped_call(in, n, n + 4096, 15 * (sec + ((size_t)1 << 20)), out, out_n,
                                  [n, window, sec](Ctx &c, hipStream_t s, const uint8_t *di, uint8_t *dout, size_t cap, size_t *got, const SliceStream *st) {
                                      SliceStream mine = *st;
                                      mine.slice_bytes = sec;
                                      return lzss_encode_sliced(c, s, di, n, window, dout, cap, got, mine);
                                  });
        if (rc != 1) return rc;
    }
    // (the encoder's own statement of need, lzss_encode_dev: ~13 bytes of scratch per position of a pass + the escaped stream)
    return host_call(in, n, out, out_n, lzss_compress_bound(n), n < ((size_t)32 << 20) ? 0 : 13 * std::min(n, (size_t)3 << 29) + 2 * n,
                     [n, window](Ctx &c, hipStream_t s, const uint8_t *di, uint8_t *dout, size_t cap, size_t *got) {
                         return lzss_encode_dev(c, s, di, n, window, dout, cap, got);
                     });
}

static int rsn_lzss_compress_legacy_impl(const uint8_t *in, size_t n, int64_t window, uint8_t **out, size_t *out_n) {
    Ctx &c = ctx();
    if (!out || !out_n || (!in && n)) return c.fail(RSN_ERR_ARG, "null argument");
    *out = nullptr; *out_n = 0;
    {   // an O(n * W) serial loop that cannot be interrupted from Go: refuse what would run for hours (rsn.h)
        static const bool no_limit = getenv("RSN_LEGACY_NO_LIMIT") != nullptr;
        const size_t bound = (window <= 0 || window > 65536) ? ((size_t)1 << 20) : ((size_t)64 << 20);
        if (!no_limit && n > bound)
            return c.fail(RSN_ERR_LIMIT, "lz.Compress (legacy, host code): %zu bytes with window %lld is above the %zu-byte bound of this serial O(n*W) path; "
                          "use rsn_lzss_compress (CompressAsync), or set RSN_LEGACY_NO_LIMIT=1", n, (long long)window, bound);
    }
    std::string res;
    lzss_compress_legacy_host(in, n, window, res);
    uint8_t *buf = (uint8_t *)result_alloc(res.size());
    if (!buf) return c.fail(RSN_ERR_NOMEM, "allocating %zu result bytes failed", res.size());
    memcpy(buf, res.data(), res.size());
    *out = buf; *out_n = res.size();
    return RSN_OK;
}

static int rsn_lzss_decompress_impl(const uint8_t *in, size_t n, uint8_t **out, size_t *out_n) {
    if (in && out && out_n && n <= 2048) {                                // (lzss_small.hip: one launch; what it does not take -- malformed, large output -- goes on below)
        Ctx &c = ctx(); const uint8_t *p = nullptr; size_t got = 0;
        *out = nullptr; *out_n = 0;
        const int rc = small_result(c, lzss_small_decompress(c, in, n, &p, &got), p, got, out, out_n);
        if (rc != 1) return rc;
    }
    // From 64 MiB up: the pipeline (piped_call + lzss_decode_sliced, r06) -- the stream is decoded slice by slice as its pieces land, every
    // slice's bytes on their way down while the next ones come up.  The result block must exist before the first byte is decoded and the
    // stream does not say how long it is (lzss.go:323-364 appends as it goes): the first 256 KiB are parsed HERE, on the host (4 MiB took 4 ms of a 27 ms call), and the
    // expansion they show + 15 % + 32 MiB is what is allocated; a stream that outgrows that, or holds a 5C, or whose sample has a token the
    // parser does not take, is decoded by the serial call below (which also words the errors).
    static const bool serial = getenv("RSN_HOST_SERIAL") != nullptr;
    if (!serial && in && out && out_n && n >= ((size_t)64 << 20)) {
        const size_t sample = std::min(n, (size_t)256 << 10);
        unsigned long long esc = 0; size_t i = 0; bool ok = true;
        while (i < sample && ok) {
            const uint8_t b = in[i];
            if (b == 0x5C) ok = false;
            else if (b != '<') { esc++; i++; }
            else {
                size_t q = i + 1; unsigned long long ptr = 0, len = 0; int nd = 0;
                while (q < n && nd < 10 && in[q] >= '0' && in[q] <= '9') { ptr = ptr * 10 + (in[q] - '0'); q++; nd++; }
                ok = nd && q < n && in[q] == ','; q++; nd = 0;
                while (ok && q < n && nd < 10 && in[q] >= '0' && in[q] <= '9') { len = len * 10 + (in[q] - '0'); q++; nd++; }
                ok = ok && nd && q < n && in[q] == '>' && len <= ptr;
                esc += len; i = q + 1;
            }
        }
        if (ok && i > 0) {
            const double ratio = (double)esc / (double)i;
            const size_t cap = (size_t)((double)n * ratio * 1.15) + ((size_t)32 << 20);
            if (cap < ((size_t)1 << 32) - ((size_t)1 << 20)) {
                const size_t slice = (size_t)64 << 20;
                const int rc = piped_call(in, n, cap, cap + 10 * (size_t)((double)slice * ratio) + ((size_t)64 << 20), out, out_n,
                                          [n](Ctx &c, hipStream_t s, const uint8_t *di, uint8_t *dout, size_t ocap, size_t *got, const SliceStream *st) {
                                              SliceStream mine = *st;
                                              mine.slice_bytes = slice;
                                              return lzss_decode_sliced(c, s, di, n, dout, ocap, got, mine);
                                          });
                if (rc != 1) return rc;
            }
        }
    }
    // (the decoder asks for 4 bytes per escaped byte from 64 MiB of them up, lzss_decode.hip: stated for an expansion of two -- text is 1.5;
    //  a stream that expands further decodes inside this admission all the same)
    return host_call(in, n, out, out_n, 8 * n + (1 << 16), n < ((size_t)32 << 20) ? 0 : 8 * n,
                     [n](Ctx &c, hipStream_t s, const uint8_t *di, uint8_t *dout, size_t cap, size_t *got) {
                         return lzss_decode_dev(c, s, di, n, dout, cap, got);
                     });
}

// Independent chunks, one complete .rsn segment each (engine.CompressFiles: one file per input,
// engine.go:150-154).  A single host call is PCIe-bound -- ~4.7 ms per 256 MiB each way against 0.3 ms
// of kernels -- and PCIe is full duplex, so on ONE device the batch is a three-stage pipeline over a ring of
// device buffers: one thread uploads chunk k+1 while a second encodes chunk k and a third copies chunk k-1's
// segment down.  (Lanes that each run whole calls fall into step -- all uploading, then all downloading -- and
// overlap little: 58.6 ms against the serial loop's 73.5 on 8 x 256 MiB, r02k.)
// On a node with several GPUs the chunks are dealt out first: chunk k -> worker k mod G, worker w on device
// (caller's device + w) mod visible devices, every worker its own pipeline and its own PCIe link; results land in the
// caller's host arrays, so there is nothing to exchange between devices (no data-path collective).
//   RSN_BATCH_DEVICES=<d>|all  use up to d devices, starting at the caller's (default 1: the caller's device only)
//   RSN_BATCH_WORKERS=<w>  number of pipelines (default: one per device used); more workers than devices share
//                          devices round-robin -- how the split is exercised on a one-GPU box
//   RSN_BATCH_KEEP_MIB=<m> ring buffers above m MiB per worker (default 1024) are released when the batch ends instead of
//                          staying with the worker's (parked) context for the next batch
namespace {
struct BatchPipe {
    enum { RING = 3 };
    std::mutex mu; std::condition_variable cv;
    size_t uploaded = 0, encoded = 0, downloaded = 0;      // chunks that have left each stage
    int rc = RSN_OK; std::string msg;
    void *d_in[RING] = {}, *d_out[RING] = {}, *d_tmp[RING] = {};
    size_t got[RING] = {};
    void fail(int code, const char *m) noexcept { try { std::lock_guard<std::mutex> lk(mu); if (rc == RSN_OK) { rc = code; try { msg = m; } catch (...) {} } } catch (...) {} cv.notify_all(); }
    // waits until `counter` has passed `want` chunks; false when another stage failed
    bool wait(const size_t &counter, size_t want) { std::unique_lock<std::mutex> lk(mu); cv.wait(lk, [&] { return rc != RSN_OK || counter >= want; }); return rc == RSN_OK; }
    void done(size_t &counter) { { std::lock_guard<std::mutex> lk(mu); counter++; } cv.notify_all(); }
};

// The chunks `idx` through one device's pipeline, on the calling thread's context (already initialised on its device).
// On failure the chunks this worker has produced stay in outs[] for the caller to undo; the message is in c.err.
static int batch_on_device(Ctx &c, const std::vector<size_t> &idx, const uint8_t *const *ins, const size_t *lens, uint8_t **outs, size_t *out_lens) {
    const size_t m = idx.size();
    if (m == 0) return RSN_OK;
    auto one_by_one = [&] {                                              // no pipeline: a single chunk, or no helper threads to be had
        for (size_t j = 0; j < m; j++) {
            const size_t i = idx[j];
            if (outs[i]) { rsn_free(outs[i]); outs[i] = nullptr; out_lens[i] = 0; }
            const int rc = rsn_huffman_compress(ins[i], lens[i], &outs[i], &out_lens[i]);
            if (rc != RSN_OK) return rc;
        }
        return (int)RSN_OK;
    };
    if (m < 2) return one_by_one();
    size_t max_len = 0;
    for (size_t i : idx) max_len = std::max(max_len, lens[i]);
    BatchPipe P;
    const size_t in_cap = round_up(max_len, 16) + 64, out_cap = max_len + max_len / 8 + (1 << 16);   // typical outputs are < n; a chunk that needs more gets its own block
    for (int r = 0; r < BatchPipe::RING; r++) {
        int rc = dev_buf(c, 28 + r, in_cap, &P.d_in[r]); if (rc) return rc;
        rc = dev_buf(c, 28 + BatchPipe::RING + r, out_cap, &P.d_out[r]); if (rc) return rc;
    }
    const int device = c.device;
    static const bool timing = getenv("RSN_HOST_TIMING") != nullptr;
    auto stamp = [] { return fmod(std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(), 100000.0); };

    auto stage = [&P](auto body) { return [&P, body] { guarded_call<int>([&] { body(); return 0; }, [&](int code, const char *msg) { P.fail(code, msg); return 0; }); }; };
    HelperPool::Handle encoder = HelperPool::run(device, stage([&] {
        if (rsn_device_set(device) != RSN_OK) { P.fail(RSN_ERR_DEVICE, rsn_last_error()); return; }
        Ctx &ce = ctx(); hipStream_t s = ce.own_stream;
        for (size_t j = 0; j < m; j++) {
            if (!P.wait(P.uploaded, j + 1)) return;
            const size_t i = idx[j];
            const int r = (int)(j % BatchPipe::RING);
            size_t got = 0;
            int rc = huff_encode_dev(ce, s, (const uint8_t *)P.d_in[r], lens[i], (uint8_t *)P.d_out[r], out_cap, &got, nullptr, nullptr);
            if (rc == RSN_ERR_CAPACITY && got > out_cap) {
                if (hipMalloc(&P.d_tmp[r], got) != hipSuccess) { P.d_tmp[r] = nullptr; P.fail(RSN_ERR_NOMEM, "hipMalloc of an oversized segment failed"); return; }
                rc = huff_encode_dev(ce, s, (const uint8_t *)P.d_in[r], lens[i], (uint8_t *)P.d_tmp[r], got, &got, nullptr, nullptr);
            }
            if (rc == RSN_OK && hipStreamSynchronize(s) != hipSuccess) { rc = RSN_ERR_DEVICE; ce.fail(rc, "hipStreamSynchronize after encode failed"); }
            if (rc != RSN_OK) { P.fail(rc, rsn_last_error()); return; }
            P.got[r] = got;
            if (timing) fprintf(stderr, "batch dev %d chunk %zu encoded at %.2f ms\n", device, i, stamp());
            P.done(P.encoded);
        }
    }));
    if (!encoder) return one_by_one();
    HelperPool::Handle downloader = HelperPool::run(device, stage([&] {
        if (rsn_device_set(device) != RSN_OK) { P.fail(RSN_ERR_DEVICE, rsn_last_error()); return; }
        hipStream_t s = ctx().own_stream;
        for (size_t j = 0; j < m; j++) {
            if (!P.wait(P.encoded, j + 1)) return;
            const size_t i = idx[j];
            const int r = (int)(j % BatchPipe::RING);
            const size_t got = P.got[r];
            uint8_t *res = (uint8_t *)result_alloc(got);
            if (!res) { P.fail(RSN_ERR_NOMEM, "allocating a result block failed"); return; }
            outs[i] = res; out_lens[i] = got;
            hipError_t e = hipMemcpyAsync(res, P.d_tmp[r] ? P.d_tmp[r] : P.d_out[r], got, hipMemcpyDeviceToHost, s);
            if (e == hipSuccess) e = hipStreamSynchronize(s);
            if (P.d_tmp[r]) { (void)hipFree(P.d_tmp[r]); P.d_tmp[r] = nullptr; }
            if (e != hipSuccess) { P.fail(RSN_ERR_DEVICE, hipGetErrorString(e)); return; }
            if (timing) fprintf(stderr, "batch dev %d chunk %zu down at %.2f ms\n", device, i, stamp());
            P.done(P.downloaded);
        }
    }));
    if (!downloader) {                                                    // (the encoder waits for the first upload: told to stop, it ends)
        P.fail(RSN_ERR_NOMEM, "no helper thread");
        HelperPool::wait(encoder);
        return one_by_one();
    }
    // this thread uploads
    for (size_t j = 0; j < m; j++) {
        if (j >= BatchPipe::RING && !P.wait(P.downloaded, j + 1 - BatchPipe::RING)) break;   // the ring slot is free once its segment is down
        const size_t i = idx[j];
        const int r = (int)(j % BatchPipe::RING);
        hipStream_t s = c.own_stream;
        hipError_t e = hipMemsetAsync((uint8_t *)P.d_in[r] + (lens[i] & ~(size_t)15), 0, 64, s);
        PinGuard g;                                                       // (bytes another call has registered stay registered under this copy)
        pin_acquire(ins[i], lens[i], false, g.h);
        if (e == hipSuccess) e = copy_up(P.d_in[r], ins[i], lens[i], s, g.h);
        if (e == hipSuccess) e = hipStreamSynchronize(s);
        if (e != hipSuccess) { P.fail(RSN_ERR_DEVICE, hipGetErrorString(e)); break; }
        if (timing) fprintf(stderr, "batch dev %d chunk %zu up at %.2f ms\n", device, i, stamp());
        P.done(P.uploaded);
    }
    HelperPool::wait(encoder);
    HelperPool::wait(downloader);
    for (int r = 0; r < BatchPipe::RING; r++) if (P.d_tmp[r]) (void)hipFree(P.d_tmp[r]);
    if ((in_cap + out_cap) * BatchPipe::RING > ((size_t)std::max(0, env_int("RSN_BATCH_KEEP_MIB", 1024)) << 20))
        for (int k = 28; k < 28 + 2 * BatchPipe::RING; k++) { if (c.bufs[k].p) { (void)hipFree(c.bufs[k].p); scratch_forget(c, c.bufs[k].cap); } c.bufs[k].p = nullptr; c.bufs[k].cap = 0; }
    if (P.rc != RSN_OK) return c.fail(P.rc, "%s", P.msg.c_str());
    return RSN_OK;
}
}  // namespace

static int rsn_huffman_compress_batch_impl(size_t n_chunks, const uint8_t *const *ins, const size_t *lens, uint8_t **outs, size_t *out_lens) {
    Ctx &c = ctx();
    if (!ins || !lens || !outs || !out_lens) return c.fail(RSN_ERR_ARG, "null argument");
    for (size_t i = 0; i < n_chunks; i++) { outs[i] = nullptr; out_lens[i] = 0; }
    for (size_t i = 0; i < n_chunks; i++) {
        if (!ins[i] && lens[i]) return c.fail(RSN_ERR_ARG, "null argument");
        if (lens[i] == 0) return c.fail(RSN_ERR_EMPTY, "huffman: empty input (reference panics in heap.Pop, huffman.go:102)");
    }
    int rc0 = ctx_init(c); if (rc0) return rc0;
    auto undo = [&](int rc, const char *msg) {
        for (size_t k = 0; k < n_chunks; k++) { if (outs[k]) rsn_free(outs[k]); outs[k] = nullptr; out_lens[k] = 0; }
        return c.fail(rc, "%s", msg);
    };
    // Other devices are an opt-in: in the one-process-per-GPU model every rank sees every GPU, and a batch that spread
    // by default would put contexts and ~GiB rings on the other ranks' devices.  RSN_BATCH_DEVICES=<d>|all asks for it.
    int visible = 1;
    if (hipGetDeviceCount(&visible) != hipSuccess || visible < 1) visible = 1;
    const char *bd = getenv("RSN_BATCH_DEVICES");
    const int want_dev = !bd || !*bd ? 1 : (!strcmp(bd, "all") ? visible : atoi(bd));
    const int n_dev = std::max(1, std::min(visible, want_dev));
    const size_t n_workers = std::min<size_t>((size_t)std::max(1, env_int("RSN_BATCH_WORKERS", n_dev)), std::max<size_t>(n_chunks, 1));
    if (n_workers <= 1) {                                        // one device, one pipeline, on the caller's own context
        std::vector<size_t> idx(n_chunks);
        for (size_t i = 0; i < n_chunks; i++) idx[i] = i;
        const int rc = batch_on_device(c, idx, ins, lens, outs, out_lens);
        if (rc != RSN_OK) { const std::string m = c.err; return undo(rc, m.c_str()); }
        return RSN_OK;
    }
    // chunk k -> worker k mod G; worker w -> device (caller's + w mod n_dev) mod visible
    std::vector<int> rcs(n_workers, RSN_OK);
    std::vector<std::string> msgs(n_workers);
    std::vector<HelperPool::Handle> workers(n_workers);
    const int base = c.device;
    auto share = [&](size_t w, bool own_thread) {                         // worker w's chunks; on the caller's thread they stay on the caller's device
        guarded_call<int>([&] {
            int rc = own_thread ? rsn_device_set((base + (int)(w % (size_t)n_dev)) % visible) : RSN_OK;
            if (rc == RSN_OK) {
                std::vector<size_t> idx;
                for (size_t k = w; k < n_chunks; k += n_workers) idx.push_back(k);
                rc = batch_on_device(ctx(), idx, ins, lens, outs, out_lens);
            }
            rcs[w] = rc;
            if (rc != RSN_OK) msgs[w] = rsn_last_error();
            return 0;
        }, [&](int code, const char *m) { rcs[w] = code; try { msgs[w] = m; } catch (...) {} return 0; });
    };
    for (size_t w = 0; w < n_workers; w++) workers[w] = HelperPool::run((base + (int)(w % (size_t)n_dev)) % visible, [&share, w] { share(w, true); });
    for (size_t w = 0; w < n_workers; w++) if (!workers[w]) share(w, false);   // no thread to be had for that worker: its chunks on this one
    for (auto &t : workers) HelperPool::wait(t);
    (void)hipSetDevice(c.device);
    for (size_t w = 0; w < n_workers; w++) if (rcs[w] != RSN_OK) return undo(rcs[w], msgs[w].c_str());
    return RSN_OK;
}

// ---- one Huffman stream from G slices of ONE input ("intra-file sharding"): the slices' symbol counts are summed on the
// host, ONE tree and ONE header are built (so the result is byte for byte what rsn_huffman_compress returns), every slice's bit total
// is its counts times the code lengths, and slice w writes its code bits at bit offset base + sum of the totals before it: the single
// front pad of the format (huffman.go:245-255) makes the slices meet at BIT positions, so each worker encodes into a buffer of its own
// at the right bit phase and the bytes two neighbours share are ORed together on the way down.  Worker w runs on device
// (caller's + w mod D) mod visible, D from RSN_BATCH_DEVICES (default 1: the workers share the caller's device, which is how the
// split is exercised on a one-GPU box); on a node with several GPUs each slice has its own PCIe link, histogram and emit.
// Cuts fall on rune starts (a UTF-8 sequence is never split: huffman.go:309 decodes the whole string).
namespace {
struct ShardSync {
    std::mutex mu; std::condition_variable cv;
    size_t arrived = 0, generation = 0, parties;
    int rc = RSN_OK; std::string msg;
    explicit ShardSync(size_t p) : parties(p) {}
    void fail(int code, const char *m) noexcept { try { std::lock_guard<std::mutex> lk(mu); if (rc == RSN_OK) { rc = code; try { msg = m; } catch (...) {} } } catch (...) {} cv.notify_all(); }
    // all parties meet; `last` runs on the last one to arrive, before the others go on.  false: somebody failed
    bool meet(const std::function<void()> &last) {
        std::unique_lock<std::mutex> lk(mu);
        if (rc != RSN_OK) return false;
        const size_t gen = generation;
        if (++arrived == parties) { lk.unlock(); last(); lk.lock(); arrived = 0; generation++; cv.notify_all(); }
        else cv.wait(lk, [&] { return generation != gen || rc != RSN_OK; });
        return rc == RSN_OK;
    }
};
}  // namespace

static int rsn_huffman_compress_sharded_impl(const uint8_t *in, size_t n, int shards, uint8_t **out, size_t *out_n) {
    Ctx &c = ctx();
    if (!out || !out_n || (!in && n)) return c.fail(RSN_ERR_ARG, "null argument");
    *out = nullptr; *out_n = 0;
    if (n == 0) return c.fail(RSN_ERR_EMPTY, "huffman: empty input (reference panics in heap.Pop, huffman.go:102)");
    int rc0 = ctx_init(c); if (rc0) return rc0;
    int G = shards > 0 ? shards : env_int("RSN_HUFF_SHARDS", 0);
    int visible = 1;
    if (hipGetDeviceCount(&visible) != hipSuccess || visible < 1) visible = 1;
    const char *bd = getenv("RSN_BATCH_DEVICES");
    const int n_dev = std::max(1, std::min(visible, !bd || !*bd ? 1 : (!strcmp(bd, "all") ? visible : atoi(bd))));
    if (G <= 0) G = n_dev;
    // (every slice is a thread with a stream and an arena of its own that meets the others at a barrier -- a few per device
    //  is all that can overlap anything; 64 or 255 of them only left that many parked contexts behind)
    G = std::min(G, std::max(16, 4 * n_dev));
    std::vector<size_t> cut;
    huff_slice_cuts(in, n, G, cut);                                       // on rune starts (huff_host.cpp: host logic, tested without a device)
    const size_t S = cut.size() - 1;                                      // slices
    if (S == 1) return huffman_compress_single(in, n, out, out_n);

    struct Slice { HuffSlice hs; unsigned long long bits = 0, start = 0; uint8_t edge[2] = {0, 0}; size_t first = 0, last = 0; bool has = false; };
    std::vector<Slice> sl(S);
    HuffTree tree; HuffCodes codes; std::string hdr;
    bool flat = false;
    uint8_t *res = nullptr; size_t total = 0;
    ShardSync sync(S);
    const int base_dev = c.device;

    auto plan = [&] {                                                     // runs once, on the last worker to finish its histogram
        std::vector<HuffSym> all;
        bool ascii = true;
        for (auto &x : sl) { all.insert(all.end(), x.hs.syms.begin(), x.hs.syms.end()); ascii = ascii && x.hs.ascii; }
        std::sort(all.begin(), all.end(), [](const HuffSym &a, const HuffSym &b) { return a.rune < b.rune; });
        std::vector<HuffSym> syms;
        for (const HuffSym &y : all) { if (!syms.empty() && syms.back().rune == y.rune) syms.back().freq += y.freq; else syms.push_back(y); }
        emit_header(syms, hdr);
        std::vector<HuffSym> by_rune = syms;
        std::string msg;
        if (!build_tree(syms, tree, msg)) { sync.fail(RSN_ERR_EMPTY, msg.c_str()); return; }
        if (!assign_codes(tree, codes, msg, false)) { sync.fail(RSN_ERR_LIMIT, msg.c_str()); return; }
        std::vector<std::pair<uint32_t, uint8_t>> len_of(tree.n_leaves);  // rune -> code length, ascending rune
        for (uint32_t i = 0; i < tree.n_leaves; i++) len_of[i] = {tree.rune[i], codes.len[i]};
        std::sort(len_of.begin(), len_of.end());
        unsigned long long sum = 0;
        for (auto &x : sl) {
            x.bits = 0;
            for (const HuffSym &y : x.hs.syms) {
                const auto it = std::lower_bound(len_of.begin(), len_of.end(), std::make_pair(y.rune, (uint8_t)0));
                x.bits += y.freq * it->second;
            }
            sum += x.bits;
        }
        if (sum != codes.total_bits) { sync.fail(RSN_ERR_DEVICE, "huffman: the slices' bit totals do not add up (internal error)"); return; }
        const unsigned pad = (unsigned)((8 - codes.total_bits % 8) % 8);      // huffman.go:245-249
        hdr.append("\\\n");
        hdr.push_back((char)pad);
        unsigned long long at = 8ull * hdr.size() + pad;
        for (auto &x : sl) { x.start = at; at += x.bits; }
        total = hdr.size() + (size_t)((codes.total_bits + pad) / 8);
        flat = ascii && huff_flat_code(tree, codes);
        res = (uint8_t *)result_alloc(total);
        if (!res) sync.fail(RSN_ERR_NOMEM, "allocating the result block failed");
    };

    auto worker = [&](size_t w) {
        const int dev = (base_dev + (int)(w % (size_t)n_dev)) % visible;
        if (rsn_device_set(dev) != RSN_OK) { sync.fail(RSN_ERR_DEVICE, rsn_last_error()); return; }
        Ctx &cw = ctx(); hipStream_t s = cw.own_stream;
        auto bail = [&](int rc) { sync.fail(rc, cw.err.c_str()); };
        const size_t n_w = cut[w + 1] - cut[w];
        void *d_in = nullptr, *d_out = nullptr;
        int rc = dev_buf(cw, 20, round_up(n_w, 16) + 64, &d_in); if (rc) return bail(rc);
        hipError_t e = hipMemsetAsync((uint8_t *)d_in + (n_w & ~(size_t)15), 0, 64, s);
        {
            PinGuard g;
            pin_acquire(in + cut[w], n_w, false, g.h);
            if (e == hipSuccess) e = copy_up(d_in, in + cut[w], n_w, s, g.h);
            if (e == hipSuccess && !g.h.held.empty()) e = hipStreamSynchronize(s);
        }
        if (e != hipSuccess) { sync.fail(RSN_ERR_DEVICE, hipGetErrorString(e)); return; }
        rc = huff_slice_hist(cw, s, (const uint8_t *)d_in, n_w, sl[w].hs); if (rc) return bail(rc);
        if (!sync.meet(plan)) return;
        Slice &x = sl[w];
        // the slice's buffer begins at a 16-byte boundary of the stream at or before its first bit (the kernels want that alignment)
        const size_t origin = w == 0 ? 0 : (size_t)(x.start / 8) & ~(size_t)15;
        const unsigned long long base = x.start - 8ull * origin;
        const size_t local = (size_t)((base + x.bits + 7) / 8);
        rc = dev_buf(cw, 21, round_up(local, 16) + 64, &d_out); if (rc) return bail(rc);
        rc = huff_slice_emit(cw, s, (const uint8_t *)d_in, n_w, x.hs, tree, codes, flat, w == 0 ? hdr : std::string(), base, x.bits, (uint8_t *)d_out);
        if (rc) return bail(rc);
        // down: the bytes that are this slice's alone straight into the result, its first and last payload byte (which a neighbour
        // may share) into `edge`, to be ORed in when everybody is done
        size_t lo = w == 0 ? 0 : (size_t)(x.start / 8), hi = x.bits ? (size_t)((x.start + x.bits - 1) / 8) : (w == 0 ? hdr.size() - 1 : lo);   // inclusive
        if (x.bits == 0 && w != 0) return;                                  // (writes nothing)
        x.has = x.bits != 0;
        x.first = (size_t)(x.start / 8); x.last = hi;
        const uint8_t *dl = (const uint8_t *)d_out - origin;                // dl[k] = byte k of the stream
        e = hipSuccess;
        if (x.has) {
            e = hipMemcpyAsync(&x.edge[0], dl + x.first, 1, hipMemcpyDeviceToHost, s);
            if (e == hipSuccess) e = hipMemcpyAsync(&x.edge[1], dl + x.last, 1, hipMemcpyDeviceToHost, s);
            if (w == 0 && x.first > 0 && e == hipSuccess) e = hipMemcpyAsync(res, dl, x.first, hipMemcpyDeviceToHost, s);   // the header
            if (x.last > x.first + 1 && e == hipSuccess) e = hipMemcpyAsync(res + x.first + 1, dl + x.first + 1, x.last - x.first - 1, hipMemcpyDeviceToHost, s);
        } else e = hipMemcpyAsync(res, dl, hi + 1 - lo, hipMemcpyDeviceToHost, s);   // one distinct symbol: the header is the stream
        if (e == hipSuccess) e = hipStreamSynchronize(s);
        if (e != hipSuccess) sync.fail(RSN_ERR_DEVICE, hipGetErrorString(e));
    };

    // a slice whose thread cannot be had, or whose worker throws, fails the call: the workers that did start stop at the barrier
    auto safely = [&](size_t w) { guarded_call<int>([&] { worker(w); return 0; }, [&](int code, const char *m) { sync.fail(code, m); return 0; }); };
    std::vector<HelperPool::Handle> threads;
    threads.reserve(S);
    for (size_t w = 1; w < S; w++) {
        threads.push_back(HelperPool::run((base_dev + (int)(w % (size_t)n_dev)) % visible, [&safely, w] { safely(w); }));
        if (!threads.back()) { sync.fail(RSN_ERR_NOMEM, "huffman: no helper thread for a slice of the sharded stream"); break; }
    }
    safely(0);                                                            // the caller is worker 0, on its own context
    for (auto &t : threads) HelperPool::wait(t);
    (void)hipSetDevice(c.device);
    if (sync.rc != RSN_OK) { if (res) result_free(res); return c.fail(sync.rc, "%s", sync.msg.c_str()); }
    for (auto &x : sl) if (x.has) { res[x.first] = 0; res[x.last] = 0; }
    for (auto &x : sl) if (x.has) { res[x.first] |= x.edge[0]; if (x.last != x.first) res[x.last] |= x.edge[1]; }
    *out = res; *out_n = total;
    return RSN_OK;
}

static void rsn_prof_enable_impl(int on) { ctx().prof = on != 0; }
static void rsn_prof_reset_impl(void) {
    Ctx &c = ctx();
    prof_collect(c);
    for (auto &s : c.slots) { s.launches = 0; s.total_ms = 0; }
}
static int rsn_prof_get_impl(rsn_prof_entry *entries, int cap) {
    Ctx &c = ctx();
    prof_collect(c);
    int k = 0;
    for (auto &s : c.slots) {
        if (!s.launches) continue;
        if (k < cap && entries) {
            snprintf(entries[k].name, sizeof entries[k].name, "%s", s.name.c_str());
            entries[k].launches = s.launches;
            entries[k].total_ms = s.total_ms;
        }
        k++;
    }
    return k;
}

static int64_t rsn_huffman_table_impl(const uint8_t *in, size_t n, uint32_t *runes, uint64_t *freqs, uint64_t *codes, uint8_t *lens, size_t cap) {
    Ctx &c = ctx();
    if (n == 0) return c.fail(RSN_ERR_EMPTY, "huffman: empty input");
    int rc = ctx_init(c); if (rc) return rc;
    hipStream_t s = c.own_stream;
    void *d_in;
    rc = dev_buf(c, 20, round_up(n, 16) + 64, &d_in); if (rc) return rc;
    {
        PinGuard g;
        pin_acquire(in, n, false, g.h);
        RSN_HIP(copy_up(d_in, in, n, s, g.h));
        if (!g.h.held.empty()) RSN_HIP(hipStreamSynchronize(s));
    }
    HuffTree t; HuffCodes hc; size_t dummy = 0;
    rc = huff_encode_dev(c, s, (const uint8_t *)d_in, n, nullptr, 0, &dummy, &t, &hc);
    if (rc) return rc;
    for (size_t k = 0; k < hc.dfs.size() && k < cap; k++) {
        const uint32_t id = hc.dfs[k];
        runes[k] = t.rune[id]; freqs[k] = t.freq[id]; codes[k] = hc.code[id]; lens[k] = hc.len[id];
    }
    return (int64_t)hc.dfs.size();
}

static int64_t rsn_huffman_plan_impl(const uint32_t *runes, const uint64_t *counts, size_t n_syms, uint32_t *out_runes, uint64_t *out_codes,
                         uint8_t *out_lens, uint8_t *header, size_t header_cap, size_t *header_len) {
    Ctx &c = ctx();
    std::vector<HuffSym> syms(n_syms);
    for (size_t i = 0; i < n_syms; i++) syms[i] = {runes[i], counts[i]};
    std::sort(syms.begin(), syms.end(), [](const HuffSym &a, const HuffSym &b) { return a.rune < b.rune; });
    std::string hdr;
    emit_header(syms, hdr);
    if (header_len) *header_len = hdr.size();
    if (header) { if (hdr.size() > header_cap) return c.fail(RSN_ERR_CAPACITY, "header needs %zu bytes", hdr.size()); memcpy(header, hdr.data(), hdr.size()); }
    HuffTree t; HuffCodes hc; std::string msg;
    if (!build_tree(syms, t, msg)) return c.fail(RSN_ERR_EMPTY, "%s", msg.c_str());
    if (!assign_codes(t, hc, msg)) return c.fail(RSN_ERR_LIMIT, "%s", msg.c_str());
    for (size_t k = 0; k < hc.dfs.size(); k++) {
        const uint32_t id = hc.dfs[k];
        out_runes[k] = t.rune[id]; out_codes[k] = hc.code[id]; out_lens[k] = hc.len[id];
    }
    return (int64_t)hc.dfs.size();
}

static int64_t rsn_huffman_slice_cuts_impl(const uint8_t *in, size_t n, int shards, size_t *cuts, size_t cap) {
    Ctx &c = ctx();
    if ((!in && n) || !cuts) return c.fail(RSN_ERR_ARG, "null argument");
    if (n == 0) return c.fail(RSN_ERR_EMPTY, "huffman: empty input");
    std::vector<size_t> cut;
    huff_slice_cuts(in, n, shards, cut);
    if (cut.size() > cap) return c.fail(RSN_ERR_CAPACITY, "%zu cuts, room for %zu", cut.size(), cap);
    for (size_t i = 0; i < cut.size(); i++) cuts[i] = cut[i];
    return (int64_t)cut.size() - 1;
}

static int64_t rsn_huffman_parse_header_impl(const uint8_t *header, size_t n, uint32_t *runes, uint64_t *counts, size_t cap) {
    Ctx &c = ctx();
    std::vector<HuffSym> syms; std::string msg;
    if (!parse_header(header, n, syms, msg)) return c.fail(RSN_ERR_FORMAT, "%s", msg.c_str());
    for (size_t i = 0; i < syms.size() && i < cap; i++) { runes[i] = syms[i].rune; counts[i] = syms[i].freq; }
    return (int64_t)syms.size();
}


// ---- the entry points of rsn.h: each runs its implementation inside guarded_call (rsn_helpers.h) -- a std::bad_alloc, a
// std::system_error or anything else thrown on this side becomes a negative code and a message, never std::terminate in the host
// (engine.go:315-328 can recover a panic the shim raises from a code; nothing recovers a dead process).  A host-buffer call that
// ends that way hands out nothing: *out stays NULL.
namespace {
int boundary_error(int code, const char *m) noexcept { try { ctx().err = m; } catch (...) {} return code; }
}  // namespace

extern "C" {
int rsn_device_set(int device) { return guarded_call<int>([&] { return rsn_device_set_impl(device); }, boundary_error); }
int rsn_device_count(void) { return guarded_call<int>([&] { return rsn_device_count_impl(); }, boundary_error); }
void rsn_trim(void) { guarded_call<int>([&] { rsn_trim_impl(); return 0; }, boundary_error); }
void rsn_free(void *p) { guarded_call<int>([&] { rsn_free_impl(p); return 0; }, boundary_error); }
int rsn_huffman_compress_dev(const void *d_in, size_t n, void *d_out, size_t out_cap, size_t *out_n, void *stream) { return guarded_call<int>([&] { return rsn_huffman_compress_dev_impl(d_in, n, d_out, out_cap, out_n, stream); }, boundary_error); }
int rsn_huffman_decompress_dev(const void *d_in, size_t n, void *d_out, size_t out_cap, size_t *out_n, void *stream) { return guarded_call<int>([&] { return rsn_huffman_decompress_dev_impl(d_in, n, d_out, out_cap, out_n, stream); }, boundary_error); }
int rsn_lzss_compress_dev(const void *d_in, size_t n, int64_t window, void *d_out, size_t out_cap, size_t *out_n, void *stream) { return guarded_call<int>([&] { return rsn_lzss_compress_dev_impl(d_in, n, window, d_out, out_cap, out_n, stream); }, boundary_error); }
int rsn_lzss_decompress_dev(const void *d_in, size_t n, void *d_out, size_t out_cap, size_t *out_n, void *stream) { return guarded_call<int>([&] { return rsn_lzss_decompress_dev_impl(d_in, n, d_out, out_cap, out_n, stream); }, boundary_error); }
int rsn_huffman_compress(const uint8_t *in, size_t n, uint8_t **out, size_t *out_n) {
    return guarded_call<int>([&] { return rsn_huffman_compress_impl(in, n, out, out_n); }, [&](int code, const char *m) { if (out && *out) { result_free(*out); *out = nullptr; } if (out_n) *out_n = 0; return boundary_error(code, m); });
}
int rsn_huffman_decompress(const uint8_t *in, size_t n, uint8_t **out, size_t *out_n) {
    return guarded_call<int>([&] { return rsn_huffman_decompress_impl(in, n, out, out_n); }, [&](int code, const char *m) { if (out && *out) { result_free(*out); *out = nullptr; } if (out_n) *out_n = 0; return boundary_error(code, m); });
}
int rsn_lzss_compress(const uint8_t *in, size_t n, int64_t window, uint8_t **out, size_t *out_n) {
    return guarded_call<int>([&] { return rsn_lzss_compress_impl(in, n, window, out, out_n); }, [&](int code, const char *m) { if (out && *out) { result_free(*out); *out = nullptr; } if (out_n) *out_n = 0; return boundary_error(code, m); });
}
int rsn_lzss_compress_legacy(const uint8_t *in, size_t n, int64_t window, uint8_t **out, size_t *out_n) {
    return guarded_call<int>([&] { return rsn_lzss_compress_legacy_impl(in, n, window, out, out_n); }, [&](int code, const char *m) { if (out && *out) { result_free(*out); *out = nullptr; } if (out_n) *out_n = 0; return boundary_error(code, m); });
}
int rsn_lzss_decompress(const uint8_t *in, size_t n, uint8_t **out, size_t *out_n) {
    return guarded_call<int>([&] { return rsn_lzss_decompress_impl(in, n, out, out_n); }, [&](int code, const char *m) { if (out && *out) { result_free(*out); *out = nullptr; } if (out_n) *out_n = 0; return boundary_error(code, m); });
}
int rsn_huffman_compress_batch(size_t n_chunks, const uint8_t *const *ins, const size_t *lens, uint8_t **outs, size_t *out_lens) {
    return guarded_call<int>([&] { return rsn_huffman_compress_batch_impl(n_chunks, ins, lens, outs, out_lens); }, [&](int code, const char *m) {
        if (outs && out_lens) for (size_t k = 0; k < n_chunks; k++) { if (outs[k]) result_free(outs[k]); outs[k] = nullptr; out_lens[k] = 0; }
        return boundary_error(code, m);
    });
}
int rsn_huffman_compress_sharded(const uint8_t *in, size_t n, int shards, uint8_t **out, size_t *out_n) {
    return guarded_call<int>([&] { return rsn_huffman_compress_sharded_impl(in, n, shards, out, out_n); }, [&](int code, const char *m) { if (out && *out) { result_free(*out); *out = nullptr; } if (out_n) *out_n = 0; return boundary_error(code, m); });
}
void rsn_prof_enable(int on) { guarded_call<int>([&] { rsn_prof_enable_impl(on); return 0; }, boundary_error); }
void rsn_prof_reset(void) { guarded_call<int>([&] { rsn_prof_reset_impl(); return 0; }, boundary_error); }
int rsn_prof_get(rsn_prof_entry *entries, int cap) { return guarded_call<int>([&] { return rsn_prof_get_impl(entries, cap); }, boundary_error); }
int64_t rsn_huffman_table(const uint8_t *in, size_t n, uint32_t *runes, uint64_t *freqs, uint64_t *codes, uint8_t *lens, size_t cap) { return guarded_call<int64_t>([&] { return rsn_huffman_table_impl(in, n, runes, freqs, codes, lens, cap); }, boundary_error); }
int64_t rsn_huffman_plan(const uint32_t *runes, const uint64_t *counts, size_t n_syms, uint32_t *out_runes, uint64_t *out_codes, uint8_t *out_lens, uint8_t *header, size_t header_cap, size_t *header_len) { return guarded_call<int64_t>([&] { return rsn_huffman_plan_impl(runes, counts, n_syms, out_runes, out_codes, out_lens, header, header_cap, header_len); }, boundary_error); }
int64_t rsn_huffman_slice_cuts(const uint8_t *in, size_t n, int shards, size_t *cuts, size_t cap) { return guarded_call<int64_t>([&] { return rsn_huffman_slice_cuts_impl(in, n, shards, cuts, cap); }, boundary_error); }
int64_t rsn_huffman_parse_header(const uint8_t *header, size_t n, uint32_t *runes, uint64_t *counts, size_t cap) { return guarded_call<int64_t>([&] { return rsn_huffman_parse_header_impl(header, n, runes, counts, cap); }, boundary_error); }

}  // extern "C"
