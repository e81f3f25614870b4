// rsn_common.h -- per-thread context, scratch arena, error plumbing and
// kernel-launch profiling shared by the codec translation units.
#pragma once

#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/rsn.h"

namespace rsn {

struct ProfSlot {
    std::string name;
    uint64_t launches = 0;
    double total_ms = 0;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> pending;
};

// One per host thread (thread_local): device, stream, reusable device scratch
// buffers and pinned staging.  Nothing here is shared between threads.
struct Ctx {
    int device = 0;
    bool device_chosen = false;     // by rsn_device_set(); otherwise the first call picks RSN_DEVICE's (rsn_api.hip)
    bool inited = false;
    hipStream_t own_stream = nullptr;
    std::string err;
    bool prof = false;
    std::vector<ProfSlot> slots;
    std::vector<hipEvent_t> free_events;

    struct Buf { void *p = nullptr; size_t cap = 0; };
    enum { N_BUFS = 34 };   // 28..33: the batch pipeline's ring
    Buf bufs[N_BUFS];
    void *pinned = nullptr; size_t pinned_cap = 0;

    ~Ctx();     // parks the device resources for the next thread (rsn_api.hip); makes no HIP call

    int fail(int code, const char *fmt, ...) {
        char tmp[512];
        va_list ap; va_start(ap, fmt); vsnprintf(tmp, sizeof tmp, fmt, ap); va_end(ap);
        err = tmp;
        return code;
    }
};

Ctx &ctx();
int ctx_init(Ctx &c);                       // lazy: picks device, creates stream
int dev_buf(Ctx &c, int slot, size_t bytes, void **out);   // grow-only scratch
int pinned_buf(Ctx &c, size_t bytes, void **out);
void prof_collect(Ctx &c);

#define RSN_HIP(call)                                                                              \
    do {                                                                                           \
        hipError_t e__ = (call);                                                                   \
        if (e__ != hipSuccess)                                                                     \
            return c.fail(RSN_ERR_DEVICE, "%s failed: %s (%s:%d)", #call, hipGetErrorString(e__),  \
                          __FILE__, __LINE__);                                                     \
    } while (0)

// Brackets a kernel launch with events when profiling is on.
struct ProfScope {
    Ctx &c; hipStream_t s; int slot = -1; hipEvent_t a = nullptr, b = nullptr;
    ProfScope(Ctx &c_, hipStream_t s_, const char *name) : c(c_), s(s_) {
        if (!c.prof) return;
        for (size_t i = 0; i < c.slots.size(); i++) if (c.slots[i].name == name) slot = (int)i;
        if (slot < 0) { c.slots.emplace_back(); c.slots.back().name = name; slot = (int)c.slots.size() - 1; }
        auto get = [&]() { hipEvent_t e = nullptr; if (!c.free_events.empty()) { e = c.free_events.back(); c.free_events.pop_back(); } else (void)hipEventCreate(&e); return e; };
        a = get(); b = get();
        (void)hipEventRecord(a, s);
    }
    ~ProfScope() {
        if (slot < 0) return;
        (void)hipEventRecord(b, s);
        c.slots[slot].pending.emplace_back(a, b);
    }
};

#define RSN_LAUNCH(name, kernel, grid, block, shmem, stream, ...)                                  \
    do {                                                                                           \
        {                                                                                          \
            rsn::ProfScope ps__(c, stream, name);                                                  \
            hipLaunchKernelGGL(kernel, grid, block, shmem, stream, __VA_ARGS__);                   \
        }                                                                                          \
        RSN_HIP(hipGetLastError());                                                                \
    } while (0)

static inline size_t ceil_div(size_t a, size_t b) { return (a + b - 1) / b; }
static inline size_t round_up(size_t a, size_t b) { return ceil_div(a, b) * b; }

}  // namespace rsn
