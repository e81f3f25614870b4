// lzss_big.hip -- LZSS encode for search buffers above 8192 bytes, including the unbounded one
// (NewWriterLevel(w, level) with any level >= 0, lzss.go:42-51; maxSearchBufferLength <= 0 means the
// whole prefix, lzss.go:123-127).  The engine never asks for this (it passes 4096, lzss.go:35-38); it is
// API surface, built for exactness at any size of L and distance rather than for speed:
//   B1 k_big_match   one wavefront per position: 64 candidate starts per round straight from the
//                    escaped stream (L2 serves the window a block's positions share), 8-byte
//                    compares, the maximum of  length << 32 | distance  = longest, then farthest back
//                    (bytes.Index, lzss.go:419); no 16-bit field anywhere
//   B2 k_big_jump    the greedy chain of lzss.go:136-151 by pointer doubling over the whole stream:
//                    after round r every position reachable from 0 in < 2^(r+1) steps is flagged
//   B3 k_big_count / k_scan_u64 / k_big_emit   token bytes per flagged position, offsets, output
// Cost: W/64 rounds per position (~0.4 wavefront instructions per (position, candidate) pair): a
// 300 KB stream with an unbounded window takes tens of milliseconds, 1 GiB at W = 16384 seconds.
// Streams whose candidates share prefixes of kilobytes (long runs) would make B1 quadratic in the
// window; a work budget per position turns that into RSN_ERR_LIMIT instead of a hung device.
#include "codecs.h"

namespace rsn {


namespace {
constexpr int BB = 256;                       // threads per block (4 wavefronts)
constexpr int BPOS = 64;                      // positions per block in B1
constexpr uint32_t BIG_BUDGET = 1u << 16;     // extension steps per position before giving up
constexpr int BT = 1024;                      // positions per block in B3

__device__ __forceinline__ unsigned long long load8g(const uint8_t *p) {      // 8 bytes at any address
    unsigned long long v;
    __builtin_memcpy(&v, p, 8);
    return v;
}

__global__ __launch_bounds__(BB) void k_big_match(const uint8_t *__restrict__ fc, uint32_t E, uint32_t W, unsigned long long *__restrict__ keys,
                                                  uint32_t *__restrict__ give_up) {
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    for (int k = wv; k < BPOS; k += BB / 64) {
        const unsigned long long i64 = (unsigned long long)blockIdx.x * BPOS + (unsigned)k;
        if (i64 >= E) return;
        if (__hip_atomic_load(give_up, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) return;   // somebody ran out of budget: the call fails anyway
        const uint32_t i = (uint32_t)i64, ws = i > W ? i - W : 0, capE = E - i;
        // fc is padded with 16 readable bytes, so an 8-byte load at any position < E stays inside the buffer
        const unsigned long long pat = load8g(fc + i);
        unsigned long long best = 0;
        uint32_t budget = 0;
        for (uint32_t j0 = ws; j0 < i; j0 += 64) {
            const uint32_t j = j0 + (uint32_t)lane;
            if (j < i) {
                const uint32_t d = i - j, cap = min(d, capE);                 // entirely inside the window, and inside the stream
                unsigned long long x = load8g(fc + j) ^ pat;
                uint32_t n = x ? (uint32_t)__builtin_ctzll(x) >> 3 : 8u, off = 0;
                while (n == 8 && off + 8 < cap) {                              // rare on text: the lanes concerned go on, eight bytes at a time
                    off += 8;
                    x = load8g(fc + j + off) ^ load8g(fc + i + off);
                    n = x ? (uint32_t)__builtin_ctzll(x) >> 3 : 8u;
                    if (++budget > BIG_BUDGET) { *give_up = 1; break; }
                }
                const uint32_t len = min(off + n, cap);
                if (len) best = max(best, ((unsigned long long)len << 32) | d);
            }
        }
        for (int dd = 32; dd; dd >>= 1) best = max(best, (unsigned long long)__shfl_xor((long long)best, dd));
        if (lane == 0) keys[i] = best;
    }
}

// nxt[i] = i + max(1, L), saturated at E (lzss.go:139-142); position 0 is on the chain
__global__ void k_big_init(const unsigned long long *__restrict__ keys, uint32_t E, uint32_t *__restrict__ nxt, uint32_t *__restrict__ on) {
    const unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i > E) return;
    if (i == E) { nxt[E] = E; return; }
    const unsigned long long L = keys[i] >> 32;
    nxt[i] = (uint32_t)min((unsigned long long)E, i + (L ? L : 1));
    if (i == 0) atomicOr(&on[0], 1u);
}

// one doubling round: flagged positions flag their 2^r-th successor, every position doubles its jump
__global__ void k_big_jump(const uint32_t *__restrict__ nxt, uint32_t *__restrict__ nxt_out, uint32_t E, uint32_t *__restrict__ on) {
    const unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i > E) return;
    const uint32_t j = nxt[i];
    if (i < E && j < E && ((on[i >> 5] >> (i & 31)) & 1)) atomicOr(&on[j >> 5], 1u << (j & 31));
    nxt_out[i] = nxt[j];
}

__device__ __forceinline__ uint32_t dec_digits(uint32_t v) {
    uint32_t k = 1;
    while (v >= 10) { v /= 10; k++; }
    return k;
}
// bytes position i contributes: 1 for a literal, else min(len("<d,L>"), L) with the token only if strictly shorter (lzss.go:143)
__device__ __forceinline__ uint32_t out_bytes(unsigned long long key) {
    const uint32_t L = (uint32_t)(key >> 32), d = (uint32_t)key;
    if (L == 0) return 1;
    const uint32_t e = 3 + dec_digits(d) + dec_digits(L);
    return e < L ? e : L;
}

__global__ __launch_bounds__(BT) void k_big_count(const unsigned long long *__restrict__ keys, const uint32_t *__restrict__ on, uint32_t E,
                                                  unsigned long long *__restrict__ blk_bytes) {
    __shared__ unsigned long long part[BT / 64];
    const unsigned long long i = (unsigned long long)blockIdx.x * BT + threadIdx.x;
    unsigned long long b = 0;
    if (i < E && ((on[i >> 5] >> (i & 31)) & 1)) b = out_bytes(keys[i]);
    for (int dd = 32; dd; dd >>= 1) b += __shfl_down(b, dd);
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = b;
    __syncthreads();
    if (threadIdx.x == 0) { unsigned long long t = 0; for (int k = 0; k < BT / 64; k++) t += part[k]; blk_bytes[blockIdx.x] = t; }
}

__global__ __launch_bounds__(BT) void k_big_emit(const uint8_t *__restrict__ fc, const unsigned long long *__restrict__ keys,
                                                 const uint32_t *__restrict__ on, uint32_t E, const unsigned long long *__restrict__ blk_off,
                                                 uint8_t *__restrict__ out) {
    __shared__ unsigned long long part[BT / 64];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const unsigned long long i = (unsigned long long)blockIdx.x * BT + threadIdx.x;
    const bool mine = i < E && ((on[i >> 5] >> (i & 31)) & 1);
    const unsigned long long key = mine ? keys[i] : 0;
    const unsigned long long b = mine ? out_bytes(key) : 0;
    unsigned long long incl = b;
    for (int dd = 1; dd < 64; dd <<= 1) { const unsigned long long y = __shfl_up(incl, dd); if (lane >= dd) incl += y; }
    if (lane == 63) part[wv] = incl;
    __syncthreads();
    unsigned long long pre = blk_off[blockIdx.x];
    for (int k = 0; k < wv; k++) pre += part[k];
    if (!mine) return;
    uint8_t *o = out + pre + incl - b;
    const uint32_t L = (uint32_t)(key >> 32), d = (uint32_t)key;
    if (L == 0) { *o = fc[i]; return; }
    if (b < L) {                                                       // "<d,L>" (getEncoding, lzss.go:318-320)
        auto put = [&](uint32_t v) { char t[10]; int k = 0; do { t[k++] = (char)('0' + v % 10); v /= 10; } while (v); while (k) *o++ = (uint8_t)t[--k]; };
        *o++ = '<'; put(d); *o++ = ','; put(L); *o++ = '>';
    } else for (uint32_t q = 0; q < L; q++) o[q] = fc[i + q];          // a reference no shorter than its bytes is written out (lzss.go:146)
}
}  // namespace

int lzss_encode_big(Ctx &c, hipStream_t s, const uint8_t *d_fc, uint32_t E, uint32_t W, uint8_t *d_out, size_t out_cap, size_t *out_n) {
    void *p; int rc;
    rc = dev_buf(c, 10, (size_t)E * 8 + 64, &p); if (rc) return rc;
    unsigned long long *d_keys = (unsigned long long *)p;
    rc = dev_buf(c, 11, ((size_t)E + 1) * 8 + 64, &p); if (rc) return rc;
    uint32_t *d_nxt0 = (uint32_t *)p, *d_nxt1 = d_nxt0 + ((size_t)E + 1);
    const size_t on_words = (size_t)E / 32 + 2;
    const uint32_t n_blk = (uint32_t)ceil_div(E, BT);
    rc = dev_buf(c, 12, on_words * 4 + ((size_t)n_blk * 2 + 2) * 8 + 64, &p); if (rc) return rc;
    unsigned long long *d_bbytes = (unsigned long long *)p, *d_boff = d_bbytes + n_blk, *d_btot = d_boff + n_blk;
    uint32_t *d_on = (uint32_t *)(d_btot + 2);
    uint32_t *d_give_up = (uint32_t *)(d_btot + 1);
    RSN_HIP(hipMemsetAsync(d_btot, 0, 16 + on_words * 4, s));
    RSN_LAUNCH("lzss_big_match", k_big_match, dim3((uint32_t)ceil_div(E, BPOS)), dim3(BB), 0, s, d_fc, E, W, d_keys, d_give_up);
    const uint32_t g1 = (uint32_t)ceil_div((size_t)E + 1, 256);
    RSN_LAUNCH("lzss_big_init", k_big_init, dim3(g1), dim3(256), 0, s, d_keys, E, d_nxt0, d_on);
    uint32_t *a = d_nxt0, *b = d_nxt1;
    for (unsigned long long reach = 1; reach < (unsigned long long)E; reach <<= 1) {   // after the round: everything within < 2*reach steps of position 0
        RSN_LAUNCH("lzss_big_jump", k_big_jump, dim3(g1), dim3(256), 0, s, a, b, E, d_on);
        std::swap(a, b);
    }
    RSN_LAUNCH("lzss_big_count", k_big_count, dim3(n_blk), dim3(BT), 0, s, d_keys, d_on, E, d_bbytes);
    rc = scan_u64(c, s, "lzss_scan", d_bbytes, d_boff, n_blk, d_btot); if (rc) return rc;
    void *hp; rc = pinned_buf(c, 64, &hp); if (rc) return rc;
    unsigned long long *h64 = (unsigned long long *)hp;
    RSN_HIP(hipMemcpyAsync(h64, d_btot, 16, hipMemcpyDeviceToHost, s));
    RSN_HIP(hipStreamSynchronize(s));
    if ((uint32_t)h64[1]) return c.fail(RSN_ERR_LIMIT, "lzss: window %u on a stream whose candidates share kilobyte-long prefixes (long runs) is outside "
                                        "the work budget of the large-window search; windows up to 8192 have no such limit", W);
    const size_t total = (size_t)h64[0];
    *out_n = total;
    if (total > out_cap) { *out_n = round_up(total, 16) + 16; return c.fail(RSN_ERR_CAPACITY, "lzss: output needs %zu bytes, buffer holds %zu", total, out_cap); }
    RSN_LAUNCH("lzss_big_emit", k_big_emit, dim3(n_blk), dim3(BT), 0, s, d_fc, d_keys, d_on, E, d_boff, d_out);
    RSN_HIP(hipStreamSynchronize(s));
    return RSN_OK;
}

}  // namespace rsn
