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

    struct Buf { void *p = nullptr; size_t cap = 0; unsigned long long gen = 0; };   // gen: process-unique number of this allocation (dev_buf) -- an address can come back with other contents
    enum { N_BUFS = 39 };   // 28..33: the batch pipeline's ring; 34: the one-pass Huffman decoder's tile words; 35: k_esc_try's block flags; 36: an LZSS section's stream (encoder: the aligned copy; decoder: the escaped bytes in front + the section's tokens); 37, 38: the small-input Huffman path's device copy / its decoder's block maps
    Buf bufs[N_BUFS];
    void *pinned = nullptr; size_t pinned_cap = 0;
    bool lz_runs = false;           // the LZSS input in hand holds runs of a byte (k_esc_try's flag): lzss_encode_stream walks it with k_match_chain<RUNS>
    size_t gate_held = 0;           // scratch this thread's call in progress has been admitted with (rsn_api.hip: a nested admission is covered by it)

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
// admission of calls with gigabytes of scratch (rsn_api.hip): waits while the device's calls in flight need more than it holds
size_t scratch_admit(Ctx &c, size_t need);                                   // returns what to hand to scratch_release: `need`, or 0 inside a call that has been admitted already
void scratch_release(Ctx &c, size_t held, unsigned long long slots);         // ... and gives those slots' large buffers back when others wait (bit k = slot k); held == 0: nothing to do
void scratch_forget(Ctx &c, size_t bytes);
unsigned long long scratch_queued(int device);                               // calls that have had to wait so far (tests)
void prof_collect(Ctx &c);
// hipFuncAttributeMaxDynamicSharedMemorySize belongs to the FUNCTION on a device, not to the calling thread: raised only, under a lock (r06:
// every thread kept its own "largest so far" and set the attribute whenever its own grew -- a thread with a small need lowered what another's
// next launch relied on)
int func_dyn_lds(Ctx &c, const void *fn, size_t bytes);

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

// Streaming (touched-once) global accesses as `nt` loads / stores.  Which of the headline kernels' streams use them is a
// build-time A/B mask (Makefile: EXTRA=-DRSN_NT_MASK=<bits>, scripts/ab_nt.sh):
//   1 k_byte_hist loads   2 k_emit_flat loads   4 k_emit_flat stores   8 k_dec_flat loads   16 k_dec_flat stores
//   32 k_lzd_resolve's descriptor stores   64 k_lzd_emit's descriptor loads
// Measured (r03, 1 GiB, step = encode + decode in a loop; gpurun_out/ab_nt*.txt): only bit 1 pays -- k_byte_hist
// 0.231 -> 0.185 ms (5.8 TB/s) and the step 1.175 -> 1.137 ms.  The three kernels trade the write-back of the previous kernel's
// dirty lines among themselves (nt stores in k_dec_flat: that kernel +0.05 ms, the histogram after it -0.035; nt loads in the
// histogram: k_emit_flat +0.015 because the decoder's dirty lines are then still in the cache when it starts), so every other
// bit moves time between kernels and loses a little in total; walking k_emit_flat's chunks from the end (to meet what the
// histogram read last in the Infinity Cache) changed nothing.
#ifndef RSN_NT_MASK
#define RSN_NT_MASK 1
#endif
#ifdef __HIPCC__
typedef uint32_t rsn_u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t rsn_u32x4_a4 __attribute__((ext_vector_type(4), aligned(4)));
template <bool NT> __device__ __forceinline__ uint4 ld16(const uint4 *p) {
    if constexpr (NT) { const rsn_u32x4 v = __builtin_nontemporal_load(reinterpret_cast<const rsn_u32x4 *>(p)); return make_uint4(v.x, v.y, v.z, v.w); }
    else return *p;
}
template <bool NT> __device__ __forceinline__ uint32_t ld4(const uint32_t *p) {
    if constexpr (NT) return __builtin_nontemporal_load(p);
    else return *p;
}
template <bool NT> __device__ __forceinline__ void st16(uint4 *p, const uint4 &v) {
    if constexpr (NT) { rsn_u32x4 x = {v.x, v.y, v.z, v.w}; __builtin_nontemporal_store(x, reinterpret_cast<rsn_u32x4 *>(p)); }
    else *p = v;
}
// four dwords to an address that is only dword-aligned (global stores accept that)
template <bool NT> __device__ __forceinline__ void st16_a4(uint32_t *p, uint32_t a, uint32_t b, uint32_t c_, uint32_t d) {
    rsn_u32x4_a4 x = {a, b, c_, d};
    if constexpr (NT) __builtin_nontemporal_store(x, reinterpret_cast<rsn_u32x4_a4 *>(p));
    else *reinterpret_cast<rsn_u32x4_a4 *>(p) = x;
}
#endif

static inline size_t ceil_div(size_t a, size_t b) { return (a + b - 1) / b; }
static inline size_t round_up(size_t a, size_t b) { return ceil_div(a, b) * b; }

}  // namespace rsn
