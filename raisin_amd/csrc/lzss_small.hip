// lzss_small.hip -- the LZSS codec for a host buffer of at most 1 KiB to compress, 2 KiB to decompress (r06).
//
// The reference's own benchmark table is files of 13 to 25 bytes (README.md:153-167), and what BenchmarkFile times on them is one
// Compress and one Decompress (engine.go:379-406).  At that size the general path is all fixed cost: the check and its round trip, a
// chain walk that takes 0.17 ms however few the positions, a dozen launches, three host round trips -- 0.23 ms for 3.4 KB (r05).
// Here a call is ONE launch of ONE block that reads the caller's bytes from pinned host memory, does the whole codec in LDS and stores its
// result into pinned host memory; the host polls a word instead of waiting for the stream (as huff_small.hip).
//
//   compress    EncodeOpeningSymbols (lzss.go:369-389) by a block-wide scan; for EVERY escaped position the longest match that lies
//               entirely inside the window and its leftmost occurrence (lzss.go:166-184,418-421) by looking at every distance -- at
//               most 2048 x 2048 / 2 byte compares, a thread per position; the greedy chain from 0 (lzss.go:134-151) by pointer doubling
//               over next(i) = i + max(1, L); item sizes, a scan, the bytes (lzss.go:143,318-320).
//   decompress  every '<' parses its token (lzss.go:323-364); literal / token bytes get their output offsets from a scan; every output
//               byte its source (itself, or the byte `pointer` before it), resolved by pointer doubling; DecodeOpeningSymbols
//               (lzss.go:391-406) from the parity of the 5C run in front of every byte, a scan, the bytes.
// What the path does not take returns 1 and goes through the general path, which also words the errors: inputs above the sizes below,
// streams whose tokens are malformed or point outside the data, outputs above 8 KiB.
#include <atomic>
#include <chrono>

#include "codecs.h"
#include "lzss_match.h"

namespace rsn {
namespace {

constexpr uint32_t SL_IN_MAX = 1024;            // bytes of input the encoder takes (measured: 0.04 / 0.06 / 0.10 ms for 256 / 512 / 1024 bytes of text, 0.25 for 2048 -- where the general path is as fast)
constexpr uint32_t SL_E_MAX = 2048;             // escaped positions (every byte of the input a 5C or an FF: twice its length)
constexpr uint32_t SL_DEC_IN_MAX = 2048;        // bytes of a compressed stream the decoder takes
constexpr uint32_t SL_DEC_E_MAX = 8192;         // bytes of the escaped stream it expands to, at most
constexpr int SLT = 1024;                       // threads of the one block
constexpr uint32_t SL_PENDING = 0xFFFFFFFFu, SL_NOT_MINE = 0xFFFFFFFEu;

// pinned staging of one call: offsets into Ctx::pinned
constexpr size_t SP_IN = 0, SP_OUT = 4096, SP_FLAG = SP_OUT + SL_DEC_E_MAX + 64, SP_BYTES = SP_FLAG + 64;

__device__ __forceinline__ void sl_done(uint32_t *flag, uint32_t value) {
    __threadfence_system();
    __syncthreads();
    if (threadIdx.x == 0) __hip_atomic_store(flag, value, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

// exclusive scan of one value per thread over the block; *total: the sum
__device__ __forceinline__ uint32_t sl_scan(uint32_t v, uint32_t *s_wave /*[SLT / 64 + 1]*/, uint32_t *total) {
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint32_t inc = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) { const uint32_t o = __shfl_up(inc, d, 64); if (lane >= (uint32_t)d) inc += o; }
    __syncthreads();                                                      // (the previous scan's readers are done with s_wave)
    if (lane == 63) s_wave[wave] = inc;
    __syncthreads();
    if (threadIdx.x < 64) {
        uint32_t w = threadIdx.x < SLT / 64 ? s_wave[threadIdx.x] : 0u, wi = w;
#pragma unroll
        for (int d = 1; d < SLT / 64; d <<= 1) { const uint32_t o = __shfl_up(wi, d, 64); if (lane >= (uint32_t)d) wi += o; }
        if (threadIdx.x < SLT / 64) s_wave[threadIdx.x] = wi - w;
        if (threadIdx.x == SLT / 64 - 1) s_wave[SLT / 64] = wi;
    }
    __syncthreads();
    *total = s_wave[SLT / 64];
    return s_wave[wave] + inc - v;
}

// 0xFF in every byte of w that equals the byte c (exact per byte)
__device__ __forceinline__ uint32_t sl_bytes_equal(uint32_t w, uint32_t c) {
    const uint32_t t = w ^ (c * 0x01010101u);
    const uint32_t z = ~(((t & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | t | 0x7F7F7F7Fu);
    return (z >> 7) * 0xFFu;
}
constexpr uint32_t SL_STEP_CAP = 4096;          // eight-byte extension steps a thread spends before the call is handed to the general path (runs, short periods: every candidate matches as far as it may)

__device__ __forceinline__ uint32_t sl_digits(uint32_t v) { return v < 10 ? 1u : v < 100 ? 2u : v < 1000 ? 3u : v < 10000 ? 4u : 5u; }

// ---------------------------------------------------------------- compress
// (r06, measured: the same search spread over sixteen blocks of two wavefronts with the last block to finish walking the chain was no
//  faster -- 0.05 / 0.12 / 0.18 ms for 256 / 1024 / 2048 bytes against 0.04 / 0.10 / 0.25 here: what a position's search costs is the
//  candidates of its wavefront's 64 different positions taken one after the other, not the CU's issue rate; hence the size limit.)
__global__ __launch_bounds__(SLT) void k_lzss_small_enc(const uint8_t *__restrict__ hin, uint32_t n, uint32_t W, uint8_t *__restrict__ hout, uint32_t *__restrict__ flag) {
    __shared__ __attribute__((aligned(16))) uint8_t s_in[SL_IN_MAX + 16];
    __shared__ __attribute__((aligned(16))) uint8_t s_fc[SL_E_MAX + 48];        // the escaped stream (lzss.go:369-389), zeros behind it
    __shared__ uint32_t s_key[SL_E_MAX];                                        // per position: L << 16 | distance (0: a literal)
    __shared__ uint16_t s_jmp[2][SL_E_MAX + 1];                                 // where the chain is 2^r steps on (E: beyond the end)
    __shared__ uint8_t s_on[SL_E_MAX + 1];                                      // the greedy chain from 0 lands here
    __shared__ __attribute__((aligned(16))) uint8_t s_out[SL_E_MAX + 32];
    __shared__ uint32_t s_wave[SLT / 64 + 1];
    __shared__ uint32_t s_bail;
    const uint32_t tid = threadIdx.x;
    if (tid * 16 < n) reinterpret_cast<uint4 *>(s_in)[tid] = reinterpret_cast<const uint4 *>(hin)[tid];   // (pinned host memory, zero behind n: one PCIe round trip)
    for (uint32_t i = tid; i < (SL_E_MAX + 48) / 4; i += SLT) reinterpret_cast<uint32_t *>(s_fc)[i] = 0;
    if (tid == 0) s_bail = 0;
    __syncthreads();
    // ---- EncodeOpeningSymbols: 3C -> FF; FF -> 5C FF; 5C -> 5C 5C.  Two input bytes per thread.
    uint32_t E;
    {
        const uint32_t i0 = 2 * tid;
        const uint32_t b0 = i0 < n ? s_in[i0] : 0u, b1 = i0 + 1 < n ? s_in[i0 + 1] : 0u;
        const uint32_t c0 = i0 < n ? ((b0 == 0x5C || b0 == 0xFF) ? 2u : 1u) : 0u, c1 = i0 + 1 < n ? ((b1 == 0x5C || b1 == 0xFF) ? 2u : 1u) : 0u;
        uint32_t at = sl_scan(c0 + c1, s_wave, &E);
        if (E > SL_E_MAX) { sl_done(flag, SL_NOT_MINE); return; }                 // (block-uniform)
        if (c0 == 2) { s_fc[at++] = 0x5C; s_fc[at++] = (uint8_t)b0; } else if (c0 == 1) s_fc[at++] = b0 == 0x3C ? (uint8_t)0xFF : (uint8_t)b0;
        if (c1 == 2) { s_fc[at++] = 0x5C; s_fc[at++] = (uint8_t)b1; } else if (c1 == 1) s_fc[at++] = b1 == 0x3C ? (uint8_t)0xFF : (uint8_t)b1;
    }
    __syncthreads();
    // ---- every position: the longest L >= 1 with fc[i, i + L) inside the window (L <= distance, L <= E - i), at its largest distance.
    // Sixteen candidates a load: the bytes equal to fc[i] among them are the candidates, each compared eight bytes at a time.
    const uint32_t Wc = W == 0 ? E : W;                                           // (0: the unbounded search buffer, lzss.go:125)
    const uint32_t *fw = reinterpret_cast<const uint32_t *>(s_fc);
    uint32_t steps = 0;
    for (uint32_t i = tid; i < E; i += SLT) {
        const uint32_t bi = s_fc[i], cap = E - i;
        const unsigned long long pat = lds_load8(fw, i);
        uint32_t best = 0;
        const uint32_t jlo = i - min(i, Wc);
        auto cand = [&](uint32_t j) {                                              // one candidate: fc[j] == fc[i], j in [jlo, i)
            const uint32_t d = i - j, lim = min(d, cap), bl = best >> 16;
            // (candidates come farthest first: a later one only counts if it is LONGER than the best so far -- the byte at that
            //  length decides for most of them before anything else is compared)
            if (lim <= bl || s_fc[j + bl] != s_fc[i + bl]) return;
            unsigned long long x = lds_load8(fw, j) ^ pat;
            uint32_t L = x ? (uint32_t)__builtin_ctzll(x) >> 3 : 8u;
            if (!x) {
                uint32_t off = 8;
                while (off < lim) {
                    x = lds_load8(fw, j + off) ^ lds_load8(fw, i + off);
                    steps++;
                    if (x) { off += (uint32_t)__builtin_ctzll(x) >> 3; break; }
                    off += 8;
                }
                L = off;
            }
            best = max(best, (min(L, lim) << 16) | d);                            // longest, then farthest back (bytes.Index finds the leftmost, lzss.go:419)
        };
        auto word = [&](uint32_t j0, uint32_t m) {                                 // the candidates among four bytes (m: 0xFF where the byte equals fc[i])
            while (m) { const uint32_t k = (uint32_t)__builtin_ctz(m) >> 3; m &= ~(0xFFu << (8 * k)); cand(j0 + k); }
        };
        // whole 16-byte units inside [jlo, i): one test for sixteen candidates; the ragged ends byte by byte
        const uint32_t u0 = (jlo + 15) & ~15u, u1 = i & ~15u;
        for (uint32_t j = jlo; j < min(u0, i); j++) if (s_fc[j] == bi) cand(j);
        for (uint32_t base = u0; base < u1 && steps <= SL_STEP_CAP; base += 16) {
            const uint4 q = *reinterpret_cast<const uint4 *>(s_fc + base);
            const uint32_t m0 = sl_bytes_equal(q.x, bi), m1 = sl_bytes_equal(q.y, bi), m2 = sl_bytes_equal(q.z, bi), m3 = sl_bytes_equal(q.w, bi);
            if (m0 | m1 | m2 | m3) { word(base, m0); word(base + 4, m1); word(base + 8, m2); word(base + 12, m3); }
        }
        for (uint32_t j = max(u1, u0); j < i; j++) if (s_fc[j] == bi) cand(j);
        s_key[i] = best;
    }
    if (steps > SL_STEP_CAP) s_bail = 1;
    __syncthreads();
    if (s_bail) { sl_done(flag, SL_NOT_MINE); return; }
    // ---- the greedy chain from 0: i -> i + max(1, L) (lzss.go:139-142), marked by pointer doubling
    for (uint32_t i = tid; i <= E; i += SLT) {
        s_jmp[0][i] = (uint16_t)(i < E ? min(E, i + max(1u, s_key[i] >> 16)) : E);
        s_on[i] = i == 0;
    }
    __syncthreads();
    int cur = 0;
    for (uint32_t reach = 1; reach < E + 1; reach <<= 1) {                        // after the round: everything within 2 * reach - 1 steps of 0
        for (uint32_t i = tid; i < E; i += SLT) if (s_on[i]) s_on[s_jmp[cur][i]] = 1;
        for (uint32_t i = tid; i <= E; i += SLT) s_jmp[cur ^ 1][i] = s_jmp[cur][s_jmp[cur][i]];
        cur ^= 1;
        __syncthreads();
    }
    // ---- what every chain position puts out: a token iff it is shorter than what it stands for (lzss.go:143), else the bytes
    uint32_t total = 0;
    {
        const uint32_t i0 = 2 * tid;
        uint32_t sz[2] = {0, 0}, el[2] = {0, 0};
        for (int k = 0; k < 2; k++) {
            const uint32_t i = i0 + k;
            if (i < E && s_on[i]) {
                const uint32_t key = s_key[i], L = key >> 16, d = key & 0xFFFFu;
                el[k] = L ? 3 + sl_digits(d) + sl_digits(L) : 0u;
                sz[k] = L == 0 ? 1u : (el[k] < L ? el[k] : L);
            }
        }
        uint32_t at = sl_scan(sz[0] + sz[1], s_wave, &total);
        for (int k = 0; k < 2; k++) {
            const uint32_t i = i0 + k;
            if (!sz[k]) continue;
            const uint32_t key = s_key[i], L = key >> 16, d = key & 0xFFFFu;
            if (L && el[k] < L) {                                                  // "<" + itoa(d) + "," + itoa(L) + ">" (lzss.go:318-320)
                uint32_t p = at + el[k];
                s_out[--p] = '>';
                for (uint32_t v = L; ; v /= 10) { s_out[--p] = (uint8_t)('0' + v % 10); if (v < 10) break; }
                s_out[--p] = ',';
                for (uint32_t v = d; ; v /= 10) { s_out[--p] = (uint8_t)('0' + v % 10); if (v < 10) break; }
                s_out[--p] = '<';
            } else for (uint32_t q = 0; q < sz[k]; q++) s_out[at + q] = s_fc[i + q];
            at += sz[k];
        }
    }
    __syncthreads();
    if (tid * 16 < total) reinterpret_cast<uint4 *>(hout)[tid] = reinterpret_cast<const uint4 *>(s_out)[tid];
    sl_done(flag, total);
}

// ---------------------------------------------------------------- decompress
__global__ __launch_bounds__(SLT) void k_lzss_small_dec(const uint8_t *__restrict__ hin, uint32_t n, uint8_t *__restrict__ hout, uint32_t *__restrict__ flag) {
    __shared__ __attribute__((aligned(16))) uint8_t s_in[SL_DEC_IN_MAX + 32];
    __shared__ uint8_t s_cov[SL_DEC_IN_MAX + 32];                                // the byte belongs to a token's text
    __shared__ uint16_t s_src[2][SL_DEC_E_MAX];                                  // per escaped byte: where it comes from (itself: a literal)
    __shared__ __attribute__((aligned(16))) uint8_t s_val[SL_DEC_E_MAX + 16];
    __shared__ uint32_t s_wave[SLT / 64 + 1];
    __shared__ uint32_t s_bad;
    const uint32_t tid = threadIdx.x;
    if (tid * 16 < n + 16) reinterpret_cast<uint4 *>(s_in)[tid] = tid * 16 < n ? reinterpret_cast<const uint4 *>(hin)[tid] : make_uint4(0, 0, 0, 0);
    for (uint32_t i = tid; i < SL_DEC_IN_MAX + 32; i += SLT) s_cov[i] = 0;
    if (tid == 0) s_bad = 0;
    __syncthreads();
    // ---- two input bytes per thread: a '<' parses its token (lzss.go:331-352) and covers its text
    const uint32_t k0 = 2 * tid;
    uint32_t tptr[2] = {0, 0}, tlen[2] = {0, 0}, ttl[2] = {0, 0};
    for (int k = 0; k < 2; k++) {
        const uint32_t p = k0 + k;
        if (p >= n || s_in[p] != '<') continue;
        uint32_t q = p + 1; unsigned long long v = 0; int nd = 0;
        while (q < n && nd < 10 && s_in[q] >= '0' && s_in[q] <= '9') { v = v * 10 + (s_in[q] - '0'); q++; nd++; }
        bool ok = nd && q < n && s_in[q] == ',' && v <= 0xFFFFull;
        tptr[k] = (uint32_t)v; q++; v = 0; nd = 0;
        while (ok && q < n && nd < 10 && s_in[q] >= '0' && s_in[q] <= '9') { v = v * 10 + (s_in[q] - '0'); q++; nd++; }
        ok = ok && nd && q < n && s_in[q] == '>' && v <= 0xFFFFull && (uint32_t)v <= tptr[k];   // (len <= ptr: lzss.go:350's slice stays inside the data)
        if (!ok) { s_bad = 1; continue; }
        tlen[k] = (uint32_t)v; ttl[k] = q + 1 - p;
        for (uint32_t t = 0; t < ttl[k]; t++) s_cov[p + t] = 1;
    }
    __syncthreads();
    if (s_bad) { sl_done(flag, SL_NOT_MINE); return; }                             // (a '<' inside a token's text is a malformed token too: it set s_bad itself or parses as one -- checked below)
    // a '<' inside another token's text: the outer token's digits would have stopped at it, so it cannot happen once every token parsed
    uint32_t outl[2] = {0, 0};
    for (int k = 0; k < 2; k++) { const uint32_t p = k0 + k; if (p < n) outl[k] = ttl[k] ? tlen[k] : (s_cov[p] ? 0u : 1u); }
    uint32_t E;
    uint32_t at = sl_scan(outl[0] + outl[1], s_wave, &E);
    if (E > SL_DEC_E_MAX || E == 0) { sl_done(flag, E == 0 ? 0u : SL_NOT_MINE); return; }
    for (int k = 0; k < 2; k++) {
        const uint32_t p = k0 + k;
        if (!outl[k]) continue;
        if (ttl[k]) {
            if (tptr[k] > at) s_bad = 1;                                           // the slice starts before the data (lzss.go:349)
            else for (uint32_t t = 0; t < tlen[k]; t++) s_src[0][at + t] = (uint16_t)(at + t - tptr[k]);
        } else { s_src[0][at] = (uint16_t)at; s_val[at] = s_in[p]; }
        at += outl[k];
    }
    __syncthreads();
    if (s_bad) { sl_done(flag, SL_NOT_MINE); return; }
    // ---- every byte's literal: src <- src[src] until nothing moves (a copied byte always lies before the byte that copies it)
    int cur = 0;
    for (int round = 0; round < 14; round++) {
        bool moved = false;
        for (uint32_t q = tid; q < E; q += SLT) { const uint32_t a = s_src[cur][q], b = s_src[cur][a]; s_src[cur ^ 1][q] = (uint16_t)b; moved = moved || a != b; }
        cur ^= 1;
        if (!__syncthreads_or(moved)) break;
    }
    uint32_t v8[8];
    for (int k = 0; k < 8; k++) { const uint32_t q = 8 * tid + k; v8[k] = q < E ? s_val[s_src[cur][q]] : 0u; }
    __syncthreads();                                                               // (a literal's slot is read by its copies: all reads before the writes)
    for (int k = 0; k < 8; k++) if (8 * tid + k < E) s_val[8 * tid + k] = (uint8_t)v8[k];
    __syncthreads();
    // ---- DecodeOpeningSymbols (lzss.go:391-406): a byte is escaped iff the run of 5C right in front of it has odd length AND that run's
    //      first 5C is itself unescaped -- i.e. counting from the last byte that is not 5C.  last[q] = the last position <= q whose byte is
    //      not 5C (a max-scan); the run in front of q + 1 is q - last[q] long.
    uint32_t mine = 0;                                                            // 1 + the last of this thread's eight positions whose byte is not 5C; 0: none
    for (int k = 0; k < 8; k++) { const uint32_t q = 8 * tid + k; if (q < E && s_val[q] != 0x5C) mine = q + 1; }
    uint32_t inc = mine;
    {
        const uint32_t lane = tid & 63, wave = tid >> 6;
        for (int d = 1; d < 64; d <<= 1) { const uint32_t o = __shfl_up(inc, d, 64); if (lane >= (uint32_t)d) inc = max(inc, o); }
        __syncthreads();
        if (lane == 63) s_wave[wave] = inc;
        __syncthreads();
    }
    uint32_t keep[8], ov[8], cnt = 0;
    // run of 5C in front of q: q - (last non-5C position before q, + 1)
    {
        uint32_t before_q = 0;                                                     // last non-5C position + 1 among positions < q
        {
            const uint32_t lane = tid & 63, wave = tid >> 6;
            uint32_t b = 0;
            for (uint32_t w = 0; w < wave; w++) b = max(b, s_wave[w]);
            uint32_t prev = __shfl_up(inc, 1, 64);
            if (lane == 0) prev = 0;
            before_q = max(b, prev);
        }
        for (int k = 0; k < 8; k++) {
            const uint32_t q = 8 * tid + k;
            keep[k] = 0; ov[k] = 0;
            if (q < E) {
                const uint32_t run = q - before_q, b = s_val[q];
                const bool esc = run & 1u;
                keep[k] = esc || b != 0x5C;
                ov[k] = esc ? b : (b == 0xFF ? 0x3Cu : b);
                cnt += keep[k];
                if (b != 0x5C) before_q = q + 1;
            }
        }
    }
    uint32_t total;
    uint32_t o = sl_scan(cnt, s_wave, &total);
    __shared__ __attribute__((aligned(16))) uint8_t s_res[SL_DEC_E_MAX + 16];
    for (int k = 0; k < 8; k++) if (keep[k]) s_res[o++] = (uint8_t)ov[k];
    __syncthreads();
    for (uint32_t u = tid; u * 16 < total; u += SLT) reinterpret_cast<uint4 *>(hout)[u] = reinterpret_cast<const uint4 *>(s_res)[u];
    sl_done(flag, total);
}

int sl_wait(Ctx &c, hipStream_t s, const uint32_t *f) {
    const volatile uint32_t *vf = f;
    const auto t0 = std::chrono::steady_clock::now();
    for (uint32_t spins = 1;; spins++) {
        if (*vf != SL_PENDING) { std::atomic_thread_fence(std::memory_order_acquire); return RSN_OK; }
        if ((spins & 1023) == 0 && std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(5)) {
            RSN_HIP(hipStreamSynchronize(s));
            if (*vf == SL_PENDING) return c.fail(RSN_ERR_DEVICE, "lzss: the small-input kernel finished without its answer");
            return RSN_OK;
        }
        __builtin_ia32_pause();
    }
}

}  // namespace

// 1: not an input for this path (the caller takes the general one).  *out: the result in the context's pinned staging.
int lzss_small_compress(Ctx &c, const uint8_t *in, size_t n, int64_t window, const uint8_t **out, size_t *out_n) {
    if (n == 0 || n > SL_IN_MAX || window > 0xFFFF) return 1;
    int rc = ctx_init(c); if (rc) return rc;
    hipStream_t s = c.own_stream;
    void *pp; rc = pinned_buf(c, SP_BYTES, &pp); if (rc) return rc;
    uint8_t *pin = (uint8_t *)pp;
    memcpy(pin + SP_IN, in, n);
    memset(pin + SP_IN + n, 0, 16);
    uint32_t *flag = (uint32_t *)(pin + SP_FLAG);
    *flag = SL_PENDING;
    RSN_LAUNCH("lzss_small_enc", k_lzss_small_enc, dim3(1), dim3(SLT), 0, s, (const uint8_t *)(pin + SP_IN), (uint32_t)n, (uint32_t)(window <= 0 ? 0 : window), pin + SP_OUT, flag);
    rc = sl_wait(c, s, flag); if (rc) return rc;
    if (*flag == SL_NOT_MINE) return 1;
    *out = pin + SP_OUT; *out_n = *flag;
    return RSN_OK;
}

int lzss_small_decompress(Ctx &c, const uint8_t *in, size_t n, const uint8_t **out, size_t *out_n) {
    if (n == 0 || n > SL_DEC_IN_MAX) return 1;
    int rc = ctx_init(c); if (rc) return rc;
    hipStream_t s = c.own_stream;
    void *pp; rc = pinned_buf(c, SP_BYTES, &pp); if (rc) return rc;
    uint8_t *pin = (uint8_t *)pp;
    memcpy(pin + SP_IN, in, n);
    memset(pin + SP_IN + n, 0, 32);
    uint32_t *flag = (uint32_t *)(pin + SP_FLAG);
    *flag = SL_PENDING;
    RSN_LAUNCH("lzss_small_dec", k_lzss_small_dec, dim3(1), dim3(SLT), 0, s, (const uint8_t *)(pin + SP_IN), (uint32_t)n, pin + SP_OUT, flag);
    rc = sl_wait(c, s, flag); if (rc) return rc;
    if (*flag == SL_NOT_MINE) return 1;
    *out = pin + SP_OUT; *out_n = *flag;
    return RSN_OK;
}

}  // namespace rsn
