// huff_encode.hip -- Huffman encode for gfx950 (MI355X).
//
// Replaces the inner loops of huffman.Compress (compressor/huffman/huffman.go:
// histogram :306-311, encode :229-256, bitString.AsByteSlice :174-191).
//
// Pipeline (all on one stream):
//   K1  k_byte_hist      N bytes -> per-tile (64 KiB) 256-bin histograms + global histogram
//                        (LDS sub-histograms, 16 B/lane coalesced loads)          HBM-bound: N read
//   --  D2H 2 KiB, host builds the Go-exact tree and code table (<= 256 leaves: microseconds)
//   K2  k_tile_bits      per-tile bit totals from the tile histograms (no second input read)
//   K3  k_scan_u64       exclusive scan -> bit offset of every tile
//   K4  k_emit_init      zero the 16-byte units shared by neighbouring blocks
//   K5  k_emit<MODE>     N bytes -> code words, bit-packed through LDS, 16 B/lane stores
//                                                                  HBM-bound: N read + C written
// Inputs with bytes >= 0x80 take the rune path (Go UTF-8 semantics, huffman.go:309):
//   K1r k_rune_hist, K2r k_tile_bits_rune, K5 k_emit<MODE_RUNE> (one extra read of the input).
//
// Bit layout (huffman.go:245-255): out = header || "\\\n" || byte(pad) || bytes(0^pad || S),
// S MSB-first.  The kernels treat `out` as one big-endian bit string whose first
// code bit sits at bit 8*(hdr+3)+pad; tile t starts at that plus tile_off[t].
#include <chrono>
#include <cstddef>

#include <optional>
#include "codecs.h"
#include "rsn_helpers.h"

namespace rsn {

constexpr int TILE = 65536;        // input bytes per tile
constexpr int SMALL_TILE = 4096;   // ... of an input up to SMALL_INPUT bytes.  A tile is one block's work and a block takes a few microseconds per 4 KiB
constexpr size_t SMALL_INPUT = (size_t)2 << 20;   // round whatever the input's size: 64 KiB as one tile is 16 rounds in a row on one CU beside 255 idle ones
// (64 KiB of UTF-8 text 318 -> 144 us, 1 MiB 331 -> 153, ASCII 83 -> 58 and 74 -> 64; from 8 MiB the per-tile histograms cost the ASCII path more than the
//  blocks gain -- 90 -> 121 us -- and at 64 MiB both paths lose: r02k A/B, scripts/huff_tile_ab.py)
constexpr size_t RUNE_SMALL_INPUT = (size_t)16 << 20;   // the same switch for inputs with bytes >= 0x80 (their kernels read no tile histogram; at 64 MiB: 636 -> 680 us)
constexpr int HB = 256;            // threads per block (4 wavefronts)
constexpr int ROUND = HB * 16;     // input bytes per block round (16 B per lane)

enum { MODE_ASCII = 0, MODE_ASCII_WIDE = 1, MODE_RUNE = 2 };

// ---------------------------------------------------------------- K1: byte histogram
// 128 bins x 32 copies, bin-major: copy = lane % 32 IS the LDS bank, so the 32 lanes of a
// half-wavefront never collide whatever the data.  Only 7-bit symbols are counted: an input
// with any byte >= 0x80 is merely flagged (ghist[128]) and re-histogrammed by the rune path,
// whose symbols are runes, not bytes (huffman.go:309).
template <int INFLIGHT, int TB>   // TB: bytes per tile
__global__ __launch_bounds__(HB) void k_byte_hist(const uint8_t *__restrict__ in, size_t n, uint32_t n_tiles,
                                                  uint32_t *__restrict__ tile_hist,
                                                  unsigned long long *__restrict__ ghist) {
    __shared__ uint32_t h[128 * 32];
    const int tid = threadIdx.x;
    const uint32_t copy = tid & 31;
    unsigned long long mine = 0;
    uint32_t hi_any = 0;
    static_assert(TB % ROUND == 0 && (TB / ROUND) % INFLIGHT == 0, "whole rounds, whole batches of loads");
    auto add16 = [&](const uint4 &v) {
        const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int j = 0; j < 4; j++) {
            hi_any |= w[j];
            atomicAdd(&h[((w[j] & 0x7F) << 5) | copy], 1u);
            atomicAdd(&h[(((w[j] >> 8) & 0x7F) << 5) | copy], 1u);
            atomicAdd(&h[(((w[j] >> 16) & 0x7F) << 5) | copy], 1u);
            atomicAdd(&h[(((w[j] >> 24) & 0x7F) << 5) | copy], 1u);
        }
    };
    for (uint32_t t = blockIdx.x; t < n_tiles; t += gridDim.x) {
        for (int i = tid; i < 128 * 32; i += HB) h[i] = 0;
        __syncthreads();
        const size_t base = (size_t)t * TB;
        if (base + TB <= n) {
            // full tile: branch-free, INFLIGHT loads in flight per lane before the first LDS atomic
            const uint4 *src = reinterpret_cast<const uint4 *>(in + base) + tid;
#pragma unroll
            for (int k0 = 0; k0 < TB / ROUND; k0 += INFLIGHT) {
                uint4 v[INFLIGHT];
#pragma unroll
                for (int k = 0; k < INFLIGHT; k++) v[k] = ld16<(RSN_NT_MASK & 1) != 0>(src + (k0 + k) * HB);
#pragma unroll
                for (int k = 0; k < INFLIGHT; k++) add16(v[k]);
            }
        } else {
            for (int k = 0; k < TB / ROUND; k++) {
                const size_t off = base + (size_t)(k * HB + tid) * 16;
                if (off + 16 <= n) add16(*reinterpret_cast<const uint4 *>(in + off));
                else if (off < n) for (size_t p = off; p < n; p++) { hi_any |= in[p]; atomicAdd(&h[((in[p] & 0x7F) << 5) | copy], 1u); }
            }
        }
        __syncthreads();
        if (tid < 128) {
            uint32_t s = 0;
#pragma unroll
            for (int r = 0; r < 32; r++) s += h[(tid << 5) | ((r + tid) & 31)];   // rotated: conflict-free
            tile_hist[(size_t)t * 128 + tid] = s;
            mine += s;
        }
        __syncthreads();
    }
    if (tid < 128 && mine) atomicAdd(&ghist[tid], mine);
    if (hi_any & 0x80808080u) atomicAdd(&ghist[128], 1ull);
}

// ---------------------------------------------------------------- Go UTF-8 classification (rune path)
// Length of the valid sequence starting with b0 (1 ASCII, 2..4), 0 if invalid
// (=> U+FFFD consuming one byte).  Accept ranges: go1.15 unicode/utf8.
__device__ __forceinline__ int seq_len(uint32_t b0, uint32_t b1, uint32_t b2, uint32_t b3) {
    if (b0 < 0x80) return 1;
    if (b0 < 0xC2 || b0 > 0xF4) return 0;
    uint32_t lo = 0x80, hi = 0xBF;
    if (b0 == 0xE0) lo = 0xA0;
    else if (b0 == 0xED) hi = 0x9F;
    else if (b0 == 0xF0) lo = 0x90;
    else if (b0 == 0xF4) hi = 0x8F;
    if (b1 < lo || b1 > hi) return 0;
    if (b0 < 0xE0) return 2;
    if ((b2 & 0xC0) != 0x80) return 0;
    if (b0 < 0xF0) return 3;
    if ((b3 & 0xC0) != 0x80) return 0;
    return 4;
}

__device__ __forceinline__ uint32_t load_word_clamped(const uint8_t *in, size_t n, long long off) {
    // 4 bytes at `off` (multiple of 4); bytes outside [0,n) read as 0
    if (off < 0 || (size_t)off >= n) return 0;
    if ((size_t)off + 4 <= n) return *reinterpret_cast<const uint32_t *>(in + off);
    uint32_t w = 0;
    for (int k = 0; k < 4 && (size_t)off + k < n; k++) w |= (uint32_t)in[off + k] << (8 * k);
    return w;
}

// Classifies the 16 positions [P, P+16).  A position is a rune start unless a
// VALID multi-byte sequence begins 1..3 bytes before it (lead bytes are never
// continuation bytes, so every valid sequence start is itself a rune start:
// the decision is local, DESIGN.md "rune classification").  Returns the start
// mask; rune[k] is meaningful where bit k is set.
__device__ __forceinline__ uint32_t classify16(const uint8_t *__restrict__ in, size_t n, size_t P, uint32_t rune[16]) {
    uint32_t w[6];
#pragma unroll
    for (int j = 0; j < 6; j++) w[j] = load_word_clamped(in, n, (long long)P - 4 + 4 * j);
    const uint32_t valid = (P + 16 <= n) ? 0xFFFFu : (P < n ? ((1u << (n - P)) - 1u) : 0u);
    if (((w[0] | w[1] | w[2] | w[3] | w[4] | w[5]) & 0x80808080u) == 0) {
#pragma unroll
        for (int k = 0; k < 16; k++) rune[k] = (w[1 + (k >> 2)] >> (8 * (k & 3))) & 0xFF;
        return valid;
    }
    auto B = [&](int i) -> uint32_t { return (w[(i + 4) >> 2] >> (8 * ((i + 4) & 3))) & 0xFF; };   // i in [-4, 19]
    int v[19];
#pragma unroll
    for (int q = -3; q < 16; q++) v[q + 3] = seq_len(B(q), B(q + 1), B(q + 2), B(q + 3));
    uint32_t mask = 0;
#pragma unroll
    for (int k = 0; k < 16; k++) {
        const bool consumed = v[k + 2] > 1 || v[k + 1] > 2 || v[k] > 3;
        if (!consumed) mask |= 1u << k;
        const int L = v[k + 3];
        const uint32_t b0 = B(k), b1 = B(k + 1), b2 = B(k + 2), b3 = B(k + 3);
        uint32_t r = b0;
        if (L == 0) r = kRuneError;
        else if (L == 2) r = ((b0 & 0x1F) << 6) | (b1 & 0x3F);
        else if (L == 3) r = ((b0 & 0x0F) << 12) | ((b1 & 0x3F) << 6) | (b2 & 0x3F);
        else if (L == 4) r = ((b0 & 0x07) << 18) | ((b1 & 0x3F) << 12) | ((b2 & 0x3F) << 6) | (b3 & 0x3F);
        rune[k] = r;
    }
    return mask & valid;
}

// The same runes from a START MAP (r03): k_rune_hist classifies every position once -- Go's accept ranges, 19 validity tests per lane --
// and leaves the 16-bit start mask of every 16 positions; the two later passes over the input take the starts from the map.  A start
// whose next start is L > 1 bytes on is a valid L-byte sequence by construction (only valid sequences consume bytes), one whose next
// start is the next byte is an ASCII byte or an invalid one (U+FFFD): no tests, a shift and two masks per rune.
__device__ __forceinline__ uint32_t runes16_from_map(const uint8_t *__restrict__ in, size_t n, size_t P, const uint16_t *__restrict__ smask, uint32_t rune[16]) {
    uint32_t w[5];
#pragma unroll
    for (int j = 0; j < 5; j++) w[j] = load_word_clamped(in, n, (long long)P + 4 * j);
    const uint32_t m = smask[P >> 4];
    if (((w[0] | w[1] | w[2] | w[3]) & 0x80808080u) == 0) {
#pragma unroll
        for (int k = 0; k < 16; k++) rune[k] = (w[k >> 2] >> (8 * (k & 3))) & 0xFF;
        return m;
    }
    const uint32_t m32 = m | (P + 16 < n ? (uint32_t)smask[(P >> 4) + 1] << 16 : 0u);
    auto B = [&](int i) -> uint32_t { return (w[i >> 2] >> (8 * (i & 3))) & 0xFF; };   // i in [0, 19]
    const uint32_t left = P < n ? (uint32_t)min((size_t)32, n - P) : 0u;   // bytes of the stream from P on (as far as it matters)
#pragma unroll
    for (int k = 0; k < 16; k++) {
        const uint32_t rest = m32 >> (k + 1);
        // bytes of the rune that starts at k: up to the next start, or -- no start within reach: the stream's last rune -- to the end of the stream
        const uint32_t L = rest ? (uint32_t)__builtin_ctz(rest) + 1u : (left > (uint32_t)k ? min(left - (uint32_t)k, 4u) : 1u);
        const uint32_t b0 = B(k), b1 = B(k + 1), b2 = B(k + 2), b3 = B(k + 3);
        uint32_t r = b0 < 0x80 ? b0 : kRuneError;                            // one byte: itself, or an invalid byte
        if (L == 2) r = ((b0 & 0x1F) << 6) | (b1 & 0x3F);
        else if (L == 3) r = ((b0 & 0x0F) << 12) | ((b1 & 0x3F) << 6) | (b2 & 0x3F);
        else if (L >= 4) r = ((b0 & 0x07) << 18) | ((b1 & 0x3F) << 12) | ((b2 & 0x3F) << 6) | (b3 & 0x3F);
        rune[k] = r;
    }
    return m;
}

// ---------------------------------------------------------------- K1r: rune histogram
__global__ __launch_bounds__(HB) void k_rune_hist(const uint8_t *__restrict__ in, size_t n,
                                                  unsigned long long *__restrict__ ghist, uint16_t *__restrict__ smask) {
    // runes < 0x800 (ASCII, Latin, Greek, Cyrillic, Hebrew, Arabic ...): dense LDS bins.  The rest: a small
    // open-addressing table per block -- real text has a handful of them (quotes, dashes, currency signs),
    // each hot, and a global atomic per occurrence would serialise on a few L2 lines.  A full table (CJK
    // text: thousands of distinct runes, none of them hot) falls through to global atomics.
    constexpr uint32_t NS = 512, EMPTY = 0xFFFFFFFFu;
    __shared__ uint32_t h[2048];
    __shared__ uint32_t s_key[NS], s_cnt[NS];
    __shared__ uint32_t s_full;                                        // a probe sequence has failed: stop searching long for newcomers
    const int tid = threadIdx.x;
    if (tid == 0) s_full = 0;
    for (int i = tid; i < 2048; i += HB) h[i] = 0;
    for (int i = tid; i < (int)NS; i += HB) { s_key[i] = EMPTY; s_cnt[i] = 0; }
    __syncthreads();
    uint32_t n_err = 0;            // U+FFFD is the hot bin on binary data: count it in a register
    const size_t rounds = (n + ROUND - 1) / ROUND;
    for (size_t rd = blockIdx.x; rd < rounds; rd += gridDim.x) {
        const size_t P = rd * ROUND + (size_t)tid * 16;
        if (P >= n) continue;
        uint32_t rune[16];
        const uint32_t m = classify16(in, n, P, rune);
        if (smask) smask[P >> 4] = (uint16_t)m;
#pragma unroll
        for (int k = 0; k < 16; k++) {
            if (!((m >> k) & 1)) continue;
            const uint32_t r = rune[k];
            if (r < 0x800) atomicAdd(&h[r], 1u);
            else if (r == kRuneError) n_err++;
            else {
                uint32_t slot = (r * 0x9E3779B1u) >> 23;                  // 9 bits
                bool done = false;
                const int max_probe = s_full ? 2 : 8;
                for (int probe = 0; probe < max_probe && !done; probe++, slot = (slot + 1) & (NS - 1)) {
                    uint32_t key = s_key[slot];
                    if (key == EMPTY) key = atomicCAS(&s_key[slot], EMPTY, r);
                    if (key == EMPTY || key == r) { atomicAdd(&s_cnt[slot], 1u); done = true; }
                }
                if (!done) { s_full = 1; atomicAdd(&ghist[r], 1ull); }
            }
        }
    }
    __syncthreads();
    for (int i = tid; i < 2048; i += HB) if (h[i]) atomicAdd(&ghist[i], (unsigned long long)h[i]);
    for (int i = tid; i < (int)NS; i += HB) if (s_cnt[i]) atomicAdd(&ghist[s_key[i]], (unsigned long long)s_cnt[i]);
    for (int d = 32; d; d >>= 1) n_err += __shfl_down(n_err, d);
    if ((tid & 63) == 0 && n_err) atomicAdd(&ghist[kRuneError], (unsigned long long)n_err);
}

// ---------------------------------------------------------------- K2: tile bit totals
// one wavefront per tile: sum_s hist[t][s] * len[s] over the 128 byte bins
__global__ __launch_bounds__(HB) void k_tile_bits(const uint32_t *__restrict__ tile_hist, const uint8_t *__restrict__ lens,
                                                  uint32_t n_tiles, unsigned long long *__restrict__ tile_bits) {
    const uint32_t t = blockIdx.x * (HB / 64) + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (t >= n_tiles) return;
    const uint32_t *h = tile_hist + (size_t)t * 128;
    unsigned long long s = (unsigned long long)h[lane] * lens[lane] + (unsigned long long)h[lane + 64] * lens[lane + 64];
    for (int d = 32; d; d >>= 1) s += __shfl_down(s, d);
    if (lane == 0) tile_bits[t] = s;
}

// K2r: rune path -- re-reads the tile, sums len[rune] over rune starts
__global__ __launch_bounds__(HB) void k_tile_bits_rune(const uint8_t *__restrict__ in, size_t n, const uint8_t *__restrict__ rlen,
                                                       uint32_t n_tiles, uint32_t tile, unsigned long long *__restrict__ tile_bits,
                                                       const uint16_t *__restrict__ smask) {   // the start map of k_rune_hist, or null: classify again
    __shared__ unsigned long long part[HB / 64];
    __shared__ uint8_t s_len[2048];                                // code lengths of the dense runes: an LDS read instead of a global gather
    const int tid = threadIdx.x;
    for (int i = tid; i < 2048 / 4; i += HB) reinterpret_cast<uint32_t *>(s_len)[i] = reinterpret_cast<const uint32_t *>(rlen)[i];
    __syncthreads();
    for (uint32_t t = blockIdx.x; t < n_tiles; t += gridDim.x) {
        unsigned long long s = 0;
        for (uint32_t k = 0; k < tile / ROUND; k++) {
            const size_t P = (size_t)t * tile + (size_t)(k * HB + tid) * 16;
            if (P >= n) break;
            uint32_t rune[16];
            const uint32_t m = smask ? runes16_from_map(in, n, P, smask, rune) : classify16(in, n, P, rune);
#pragma unroll
            for (int j = 0; j < 16; j++) if ((m >> j) & 1) s += rune[j] < 0x800 ? s_len[rune[j]] : rlen[rune[j]];
        }
        for (int d = 32; d; d >>= 1) s += __shfl_down(s, d);
        if ((tid & 63) == 0) part[tid >> 6] = s;
        __syncthreads();
        if (tid == 0) tile_bits[t] = part[0] + part[1] + part[2] + part[3];
        __syncthreads();
    }
}

// ---------------------------------------------------------------- K3: exclusive scan
// 8 consecutive items per lane, 8192 per block.  Up to SCAN_ONE items: one block sweeps them all.  Beyond (a 1 GiB call scans
// 131072 .. 262144 per-tile counts): one block per 8192 items scans locally and leaves its total, a second launch adds the
// totals of the blocks before -- a single block spends 0.14 ms on that many items (one CU's 64 cache lines per load).
constexpr uint32_t SCAN_BLK = 1024 * 8, SCAN_ONE = 2 * SCAN_BLK;

__device__ __forceinline__ unsigned long long scan_sweep(const unsigned long long *__restrict__ in, unsigned long long *__restrict__ out,
                                                        uint32_t base, uint32_t n, unsigned long long carry, unsigned long long *wsum) {
    constexpr int IT = 8;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const uint32_t i0 = base + tid * IT;
    unsigned long long x[IT], loc = 0;
#pragma unroll
    for (int k = 0; k < IT; k++) { x[k] = i0 + k < n ? in[i0 + k] : 0; loc += x[k]; }
    unsigned long long s = loc;
    for (int d = 1; d < 64; d <<= 1) { const unsigned long long y = __shfl_up(s, d); if (lane >= d) s += y; }
    if (lane == 63) wsum[wv] = s;
    __syncthreads();
    unsigned long long pre = carry, all = 0;
    for (int k = 0; k < 16; k++) { if (k < wv) pre += wsum[k]; all += wsum[k]; }
    unsigned long long run = pre + s - loc;
#pragma unroll
    for (int k = 0; k < IT; k++) { if (i0 + k < n) out[i0 + k] = run; run += x[k]; }
    __syncthreads();
    return carry + all;
}

__global__ __launch_bounds__(1024) void k_scan_u64(const unsigned long long *__restrict__ in, unsigned long long *__restrict__ out,
                                                   uint32_t n, unsigned long long *__restrict__ total) {
    __shared__ unsigned long long wsum[16];
    unsigned long long carry = 0;
    for (uint32_t base = 0; base < n; base += SCAN_BLK) carry = scan_sweep(in, out, base, n, carry, wsum);
    if (threadIdx.x == 0 && total) *total = carry;
}

__global__ __launch_bounds__(1024) void k_scan_local(const unsigned long long *__restrict__ in, unsigned long long *__restrict__ out,
                                                     uint32_t n, unsigned long long *__restrict__ part) {
    __shared__ unsigned long long wsum[16];
    const unsigned long long all = scan_sweep(in, out, blockIdx.x * SCAN_BLK, n, 0, wsum);
    if (threadIdx.x == 0) part[blockIdx.x] = all;
}

__global__ __launch_bounds__(1024) void k_scan_add(unsigned long long *__restrict__ out, uint32_t n, const unsigned long long *__restrict__ part,
                                                   unsigned long long *__restrict__ total) {
    __shared__ unsigned long long red[16];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const uint32_t nb = gridDim.x, b = blockIdx.x;
    const bool last = b + 1 == nb;
    unsigned long long s = 0;                                            // sum of the totals before this block (all of them in the last block: the grand total)
    for (uint32_t k = tid; k < (last ? nb : b); k += 1024) s += part[k];
    for (int d = 32; d; d >>= 1) s += __shfl_down(s, d);
    if (lane == 0) red[wv] = s;
    __syncthreads();
    unsigned long long pre = 0;
    for (int k = 0; k < 16; k++) pre += red[k];
    if (last) { if (tid == 0 && total) *total = pre; pre -= part[b]; }
    if (pre) for (uint32_t i = b * SCAN_BLK + tid; i < min(n, (b + 1) * SCAN_BLK); i += 1024) out[i] += pre;
}

// Exclusive scan of n counts on the stream; *total (may be null) receives their sum.  in and out must not overlap.
int scan_u64(Ctx &c, hipStream_t s, const char *name, const unsigned long long *in, unsigned long long *out, uint32_t n, unsigned long long *total) {
    if (n <= SCAN_ONE) {
        RSN_LAUNCH(name, k_scan_u64, dim3(1), dim3(1024), 0, s, in, out, n, total);
        return RSN_OK;
    }
    const uint32_t nb = (uint32_t)ceil_div(n, SCAN_BLK);
    void *p; int rc = dev_buf(c, 24, (size_t)nb * 8, &p); if (rc) return rc;
    unsigned long long *part = (unsigned long long *)p;
    RSN_LAUNCH(name, k_scan_local, dim3(nb), dim3(1024), 0, s, in, out, n, part);
    RSN_LAUNCH(name, k_scan_add, dim3(nb), dim3(1024), 0, s, out, n, (const unsigned long long *)part, total);
    return RSN_OK;
}

// ---------------------------------------------------------------- K4: boundary units
// Zero the 16-byte unit that holds the first bit of every emit block's range and
// the unit that holds the end of the stream: those are the only units written
// with atomicOr (they are shared by two neighbouring blocks / the header).
__global__ void k_emit_init(uint32_t *__restrict__ out_words, const unsigned long long *__restrict__ tile_off,
                            unsigned long long base_bits, uint32_t tiles_per_block, uint32_t n_tiles, uint32_t n_blocks,
                            unsigned long long end_bit) {
    const uint32_t b = blockIdx.x * blockDim.x + threadIdx.x;
    unsigned long long unit;
    if (b < n_blocks) {
        const unsigned long long t0 = (unsigned long long)b * tiles_per_block;
        if (t0 >= n_tiles) return;
        unit = (base_bits + tile_off[t0]) >> 7;
    } else if (b == n_blocks) {
        unit = end_bit >> 7;
    } else return;
    uint4 z = {0, 0, 0, 0};
    *reinterpret_cast<uint4 *>(out_words + unit * 4) = z;
}

// ---------------------------------------------------------------- K5: emit
template <int MODE> struct EmitCfg;
constexpr int TAB_LEN_SHIFT = 24;                      // table entry = len<<24 | code: the length is the top BYTE (SDWA operand)
constexpr uint32_t TAB_CODE_MASK = (1u << TAB_LEN_SHIFT) - 1;
template <> struct EmitCfg<MODE_ASCII> { static constexpr int MAXLEN = TAB_LEN_SHIFT; };
template <> struct EmitCfg<MODE_ASCII_WIDE> { static constexpr int MAXLEN = 64; };
template <> struct EmitCfg<MODE_RUNE> { static constexpr int MAXLEN = 64; };

struct EmitArgs {
    const uint8_t *in; size_t n;
    const uint32_t *tab32;               // MODE_ASCII: 256 x (len<<24 | code)
    const unsigned long long *code64;    // WIDE: [256]; RUNE: [kMaxRune]
    const uint8_t *len8;                 // WIDE: [256]; RUNE: [kMaxRune]
    const unsigned long long *tile_off;
    unsigned long long base_bits;
    uint32_t tiles_per_block, n_tiles;
    uint32_t *out_words;
    uint32_t tile;                       // input bytes per tile (TILE or SMALL_TILE)
    const uint16_t *smask;               // RUNE: the start map of k_rune_hist (16 positions per entry), or null
};

// Per-lane bit packer into the block's LDS window (big-endian 32-bit words:
// bit 31 of word w is stream bit 32*w).  The first word a lane touches and its
// trailing partial word may be shared with neighbours -> ds_or; words in between
// are wholly owned -> plain ds_write.
struct Packer {
    uint32_t *win; unsigned long long acc; uint32_t nb, w;
    __device__ __forceinline__ void start(uint32_t *window, uint32_t bit_off) { win = window; acc = 0; nb = bit_off & 31; w = bit_off >> 5; }
    __device__ __forceinline__ void put(uint32_t code, uint32_t len) {   // len <= 26
        acc = (acc << len) | code;
        nb += len;
        if (nb >= 32) {
            nb -= 32;
            atomicOr(&win[w], (uint32_t)(acc >> nb));   // the window is zeroed every round: OR == store, and is safe on shared words
            w++;
        }
    }
    __device__ __forceinline__ void put64(unsigned long long code, uint32_t len) {   // len <= 64
        if (len > 52) { put((uint32_t)(code >> 52), len - 52); len = 52; }
        if (len > 26) { put((uint32_t)(code >> 26) & 0x3FFFFFFu, len - 26); len = 26; }
        put((uint32_t)code & ((1u << len) - 1u), len);
    }
    __device__ __forceinline__ void finish() {
        if (nb) atomicOr(&win[w], (uint32_t)(acc << (32 - nb)));   // garbage above bit 31 falls off
    }
};

// Wavefront inclusive scan on DPP (row_shr 1/2/4/8, row_bcast 15/31): no LDS round trips.
__device__ __forceinline__ uint32_t wave_incl_scan(uint32_t x) {
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x111, 0xF, 0xF, false);   // row_shr:1
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x112, 0xF, 0xF, false);   // row_shr:2
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x114, 0xF, 0xF, false);   // row_shr:4
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x118, 0xF, 0xF, false);   // row_shr:8
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x142, 0xA, 0xF, false);   // row_bcast:15 -> rows 1,3
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x143, 0xC, 0xF, false);   // row_bcast:31 -> rows 2,3
    return x;
}

template <int MODE>
__global__ __launch_bounds__(HB) void k_emit(EmitArgs a) {
    constexpr int MAXLEN = EmitCfg<MODE>::MAXLEN;
    constexpr int WIN_WORDS = ROUND * MAXLEN / 32 + 8;
    __shared__ __attribute__((aligned(16))) uint32_t s_win[WIN_WORDS];
    __shared__ uint32_t s_tab[MODE == MODE_ASCII ? 128 : 1];
    __shared__ unsigned long long s_code[MODE == MODE_ASCII_WIDE ? 256 : 1];
    __shared__ uint8_t s_len[MODE == MODE_ASCII_WIDE ? 256 : 1];
    __shared__ uint32_t s_wsum[HB / 64];

    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const uint32_t t0 = blockIdx.x * a.tiles_per_block;
    if (t0 >= a.n_tiles) return;
    const uint32_t t1 = min(t0 + a.tiles_per_block, a.n_tiles);
    const size_t in0 = (size_t)t0 * a.tile;
    const size_t in1 = min((size_t)t1 * a.tile, a.n);

    if (MODE == MODE_ASCII && tid < 128) s_tab[tid] = a.tab32[tid];
    if (MODE == MODE_ASCII_WIDE) { s_code[tid] = a.code64[tid]; s_len[tid] = a.len8[tid]; }
    for (int i = tid; i < WIN_WORDS; i += HB) s_win[i] = 0;

    const unsigned long long bit0 = a.base_bits + a.tile_off[t0];
    const unsigned long long unit0 = bit0 >> 7;   // first 16-byte unit this block touches (shared with block-1)
    unsigned long long win_unit = unit0;          // global unit index of s_win[0]
    uint32_t fill = (uint32_t)(bit0 & 127);       // bits of the window that precede this block's data
    __syncthreads();

    uint4 pre = {0, 0, 0, 0};                                   // software prefetch of the next round's 16 bytes per lane
    if (MODE != MODE_RUNE && in0 + (size_t)tid * 16 + 16 <= in1) pre = *reinterpret_cast<const uint4 *>(a.in + in0 + (size_t)tid * 16);
    for (size_t pos = in0; pos < in1; pos += ROUND) {
        const size_t P = pos + (size_t)tid * 16;
        // ---- pass 1: look up, sum code lengths
        uint32_t e[MODE == MODE_ASCII ? 16 : 1];
        uint32_t rune[MODE == MODE_RUNE ? 16 : 1];
        uint32_t smask = 0, w4[4] = {0, 0, 0, 0};
        uint32_t mylen = 0;
        if (MODE == MODE_RUNE) {
            smask = P < in1 ? (a.smask ? runes16_from_map(a.in, a.n, P, a.smask, rune) : classify16(a.in, a.n, P, rune)) : 0;
#pragma unroll
            for (int k = 0; k < 16; k++) if ((smask >> k) & 1) mylen += a.len8[rune[k]];
        } else {
            if (P + 16 <= in1) {
                w4[0] = pre.x; w4[1] = pre.y; w4[2] = pre.z; w4[3] = pre.w;
                smask = 0xFFFF;
                if (P + ROUND + 16 <= in1) pre = *reinterpret_cast<const uint4 *>(a.in + P + ROUND);
            } else if (P < in1) {
                const int cnt = (int)(in1 - P);
                for (int k = 0; k < cnt; k++) w4[k >> 2] |= (uint32_t)a.in[P + k] << (8 * (k & 3));
                smask = (1u << cnt) - 1;
            }
            if (MODE == MODE_ASCII && pos + ROUND <= in1) {
                // full round (block-uniform): no per-symbol masking, 16 independent LDS lookups
#pragma unroll
                for (int k = 0; k < 16; k++) {
                    e[k] = s_tab[(w4[k >> 2] >> (8 * (k & 3))) & 0x7F];
                    mylen += e[k] >> TAB_LEN_SHIFT;
                }
            } else {
#pragma unroll
                for (int k = 0; k < 16; k++) {
                    const uint32_t b = (w4[k >> 2] >> (8 * (k & 3))) & 0xFF;
                    if (MODE == MODE_ASCII) {
                        const uint32_t ent = ((smask >> k) & 1) ? s_tab[b & 0x7F] : 0;
                        e[k] = ent;
                        mylen += ent >> TAB_LEN_SHIFT;
                    } else {
                        if ((smask >> k) & 1) mylen += s_len[b];
                    }
                }
            }
        }
        // ---- block exclusive scan of mylen
        const uint32_t incl = wave_incl_scan(mylen);
        if (lane == 63) s_wsum[wv] = incl;
        __syncthreads();
        uint32_t wpre = 0, total = 0;
#pragma unroll
        for (int k = 0; k < HB / 64; k++) { const uint32_t x = s_wsum[k]; if (k < wv) wpre += x; total += x; }
        // ---- pass 2: pack into the LDS window
        Packer pk;
        pk.start(s_win, fill + wpre + incl - mylen);
        if (MODE == MODE_ASCII) {
#pragma unroll
            for (int k = 0; k < 16; k++) pk.put(e[k] & TAB_CODE_MASK, e[k] >> TAB_LEN_SHIFT);
        } else if (MODE == MODE_ASCII_WIDE) {
#pragma unroll
            for (int k = 0; k < 16; k++) {
                const uint32_t b = (w4[k >> 2] >> (8 * (k & 3))) & 0xFF;
                if ((smask >> k) & 1) pk.put64(s_code[b], s_len[b]);
            }
        } else {
#pragma unroll
            for (int k = 0; k < 16; k++) if ((smask >> k) & 1) pk.put64(a.code64[rune[k]], a.len8[rune[k]]);
        }
        if (mylen) pk.finish();
        __syncthreads();
        // ---- flush complete 16-byte units
        const uint32_t tot_bits = fill + total;
        const uint32_t n_units = tot_bits >> 7;
        for (uint32_t u = tid; u < n_units; u += HB) {
            uint4 v = *reinterpret_cast<const uint4 *>(&s_win[u * 4]);
            v.x = __builtin_bswap32(v.x); v.y = __builtin_bswap32(v.y); v.z = __builtin_bswap32(v.z); v.w = __builtin_bswap32(v.w);
            const unsigned long long g = win_unit + u;
            uint32_t *dst = a.out_words + g * 4;
            if (g == unit0) {
                if (v.x) atomicOr(dst + 0, v.x);
                if (v.y) atomicOr(dst + 1, v.y);
                if (v.z) atomicOr(dst + 2, v.z);
                if (v.w) atomicOr(dst + 3, v.w);
            } else {
                *reinterpret_cast<uint4 *>(dst) = v;
            }
        }
        uint32_t carry = 0;
        if (tid < 4) carry = s_win[n_units * 4 + tid];
        __syncthreads();
        const uint32_t used = n_units * 4 + 4;
        for (uint32_t i = tid; i < used; i += HB) s_win[i] = i < 4 ? carry : 0;   // lanes 0..3 of wave 0 hold the carry words
        win_unit += n_units;
        fill = tot_bits & 127;
        // the scan barrier of the next round orders these writes before the next pack
    }
    __syncthreads();
    if (fill && tid < 4) {
        const uint32_t v = __builtin_bswap32(s_win[tid]);
        if (v) atomicOr(a.out_words + win_unit * 4 + tid, v);
    }
}

// ---------------------------------------------------------------- K5a: emit, ASCII alphabet, codes <= 24 bits
// The hot kernel.  Same bit layout and hand-over rules as k_emit, tuned for issue slots
// (the kernel is VALU-issue-bound, profiles/): 32 symbols per lane and round halve the
// per-round work (scan, barriers, flush), full rounds are branch-free, and the flush pass
// clears the LDS window as it drains it.
template <int A32_SPL>                      // symbols per lane per round (multiple of 16)
__global__ __launch_bounds__(HB) void k_emit_ascii32(EmitArgs a) {
    constexpr int A32_ROUND = HB * A32_SPL;
    constexpr int A32_WIN = A32_ROUND * TAB_LEN_SHIFT / 32 + 8;
    constexpr int NV = A32_SPL / 16;
    __shared__ __attribute__((aligned(16))) uint32_t s_win[A32_WIN];
    __shared__ uint32_t s_tab[128];
    __shared__ uint32_t s_wsum[HB / 64];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const uint32_t t0 = blockIdx.x * a.tiles_per_block;
    if (t0 >= a.n_tiles) return;
    const uint32_t t1 = min(t0 + a.tiles_per_block, a.n_tiles);
    const size_t in0 = (size_t)t0 * a.tile;
    const size_t in1 = min((size_t)t1 * a.tile, a.n);
    if (tid < 128) s_tab[tid] = a.tab32[tid];
    for (int i = tid; i < A32_WIN; i += HB) s_win[i] = 0;
    const unsigned long long bit0 = a.base_bits + a.tile_off[t0];
    const unsigned long long unit0 = bit0 >> 7;
    unsigned long long win_unit = unit0;
    uint32_t fill = (uint32_t)(bit0 & 127);
    __syncthreads();

    uint4 pre[NV];
#pragma unroll
    for (int k = 0; k < NV; k++) pre[k] = make_uint4(0, 0, 0, 0);
    if (in0 + A32_ROUND <= in1) {
        const uint4 *src = reinterpret_cast<const uint4 *>(a.in + in0 + (size_t)tid * A32_SPL);
#pragma unroll
        for (int k = 0; k < NV; k++) pre[k] = src[k];
    }
    for (size_t pos = in0; pos < in1; pos += A32_ROUND) {
        const size_t P = pos + (size_t)tid * A32_SPL;
        uint32_t e[A32_SPL];
        uint32_t mylen = 0;
        if (pos + A32_ROUND <= in1) {
            uint32_t w[NV * 4];
#pragma unroll
            for (int k = 0; k < NV; k++) { w[4 * k] = pre[k].x; w[4 * k + 1] = pre[k].y; w[4 * k + 2] = pre[k].z; w[4 * k + 3] = pre[k].w; }
            if (pos + 2 * A32_ROUND <= in1) {
                const uint4 *src = reinterpret_cast<const uint4 *>(a.in + P + A32_ROUND);
#pragma unroll
                for (int k = 0; k < NV; k++) pre[k] = src[k];
            }
#pragma unroll
            for (int k = 0; k < A32_SPL; k++) {
                e[k] = s_tab[(w[k >> 2] >> (8 * (k & 3))) & 0x7F];
                mylen += e[k] >> TAB_LEN_SHIFT;
            }
        } else {
            // last, partial round of the input
#pragma unroll
            for (int k = 0; k < A32_SPL; k++) {
                e[k] = (P + k < in1) ? s_tab[a.in[P + k] & 0x7F] : 0;
                mylen += e[k] >> TAB_LEN_SHIFT;
            }
        }
        const uint32_t incl = wave_incl_scan(mylen);
        if (lane == 63) s_wsum[wv] = incl;
        __syncthreads();
        uint32_t wpre = 0, total = 0;
#pragma unroll
        for (int k = 0; k < HB / 64; k++) { const uint32_t x = s_wsum[k]; if (k < wv) wpre += x; total += x; }
        Packer pk;
        pk.start(s_win, fill + wpre + incl - mylen);
#pragma unroll
        for (int k = 0; k < A32_SPL; k++) pk.put(e[k] & TAB_CODE_MASK, e[k] >> TAB_LEN_SHIFT);
        if (mylen) pk.finish();
        __syncthreads();
        // drain complete 16-byte units and clear them behind us
        const uint32_t tot_bits = fill + total;
        const uint32_t n_units = tot_bits >> 7;
        for (uint32_t u = tid; u < n_units; u += HB) {
            uint4 v = *reinterpret_cast<const uint4 *>(&s_win[u * 4]);
            *reinterpret_cast<uint4 *>(&s_win[u * 4]) = make_uint4(0, 0, 0, 0);
            v.x = __builtin_bswap32(v.x); v.y = __builtin_bswap32(v.y); v.z = __builtin_bswap32(v.z); v.w = __builtin_bswap32(v.w);
            const unsigned long long g = win_unit + u;
            uint32_t *dst = a.out_words + g * 4;
            if (g == unit0) {
                if (v.x) atomicOr(dst + 0, v.x);
                if (v.y) atomicOr(dst + 1, v.y);
                if (v.z) atomicOr(dst + 2, v.z);
                if (v.w) atomicOr(dst + 3, v.w);
            } else {
                *reinterpret_cast<uint4 *>(dst) = v;
            }
        }
        uint32_t carry = 0;
        if (tid < 4) carry = s_win[n_units * 4 + tid];
        __syncthreads();
        if (tid < 4) {                      // the partial unit becomes the front of the next window
            if (n_units) s_win[n_units * 4 + tid] = 0;
            s_win[tid] = carry;
        }
        win_unit += n_units;
        fill = tot_bits & 127;
        // the scan barrier of the next round orders these writes before the next pack
    }
    __syncthreads();
    if (fill && tid < 4) {
        const uint32_t v = __builtin_bswap32(s_win[tid]);
        if (v) atomicOr(a.out_words + win_unit * 4 + tid, v);
    }
}

// ---------------------------------------------------------------- K5f: emit, flat code
// Every code has the same length L (balanced tree: alphabets of 2^L near-equal symbols,
// config 2a).  Bit offsets are arithmetic: no per-tile bit counts, no scan, no LDS bit window.
// A lane packs 32 symbols into exactly L words with compile-time field positions; the
// stream's start phase (base_bits % 32) is one block-uniform funnel shift against the last
// word of the lane before.  Every output word has ONE owner (plain stores); only the word
// that also holds header bytes is merged with atomicOr.
constexpr int FE_INLINE = 128 + 2304;    // code table + header carried in the kernel-argument segment (no upload, no extra stream op)
struct FlatEmitArgs {
    const uint8_t *in; size_t n;
    const uint8_t *codes;                // 128 bytes: symbol -> L-bit code, followed by the header bytes; null: they are in `inl`
    unsigned long long base_bits;        // bit position of the first code bit in out
    uint32_t *out_words;
    uint32_t hdr_len;                    // bytes of header || "\\\n" || pad byte, staged behind the code table
    uint8_t inl[FE_INLINE];
};
constexpr int FE_SPL = 32;
constexpr int FE_SYMS = HB * FE_SPL;     // symbols per block

template <int L>
__global__ __launch_bounds__(HB) void k_emit_flat(FlatEmitArgs a) {
    __shared__ uint32_t s_code[128 * 32];                         // 32 dword copies per symbol: the copy index IS the bank
    __shared__ uint32_t s_o[HB * L + HB * L / 32 + 2];            // block's output words (swizzled), drained coalesced
    __shared__ uint32_t s_last[HB / 64];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    // (through the kernel-argument segment pointer: indexing the by-value copy would spill it to scratch)
    const uint8_t *tab = a.codes ? a.codes : (const uint8_t *)__builtin_amdgcn_kernarg_segment_ptr() + offsetof(FlatEmitArgs, inl);
    for (int i = tid; i < 128 * 32; i += HB) s_code[i] = tab[i >> 5];
    __syncthreads();
    const uint32_t rep = tid & 31;
    const uint32_t n_chunks = (uint32_t)((a.n + FE_SYMS - 1) / FE_SYMS);
    for (uint32_t chunk = blockIdx.x; chunk < n_chunks; chunk += gridDim.x) {   // persistent: the code table is staged once
    const size_t s0 = (size_t)chunk * FE_SYMS + (size_t)tid * FE_SPL;
    auto pack32 = [&](size_t first, uint32_t *v) {                // 32 symbols at `first` -> L big-endian words
        uint32_t w[8];
        if (first + 32 <= a.n) {
            const uint4 *src = reinterpret_cast<const uint4 *>(a.in + first);
            const uint4 x = ld16<(RSN_NT_MASK & 2) != 0>(src), y = ld16<(RSN_NT_MASK & 2) != 0>(src + 1);
            w[0] = x.x; w[1] = x.y; w[2] = x.z; w[3] = x.w; w[4] = y.x; w[5] = y.y; w[6] = y.z; w[7] = y.w;
        } else {
#pragma unroll
            for (int k = 0; k < 8; k++) w[k] = 0;
            for (int k = 0; k < 32 && first + k < a.n; k++) w[k >> 2] |= (uint32_t)a.in[first + k] << (8 * (k & 3));
        }
#pragma unroll
        for (int j = 0; j < L; j++) v[j] = 0;
#pragma unroll
        for (int k = 0; k < 32; k++) {
            uint32_t code = s_code[(((w[k >> 2] >> (8 * (k & 3))) & 0x7F) << 5) | rep];
            if (first + 32 > a.n && first + k >= a.n) code = 0;   // past the end of the input: zero bits
            const int bp = k * L, wi = bp >> 5, sh = bp & 31;
            if (sh + L <= 32) v[wi] |= code << (32 - sh - L);
            else { v[wi] |= code >> (sh + L - 32); v[wi + 1 < L ? wi + 1 : wi] |= code << (64 - sh - L); }
        }
    };
    uint32_t v[L];
    if (s0 < a.n) pack32(s0, v);
    else {
#pragma unroll
        for (int j = 0; j < L; j++) v[j] = 0;
    }
    // last word of the previous lane's group: DPP inside the wavefront, LDS across wavefronts,
    // recomputed from the input across blocks
    uint32_t prev = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v[L - 1], 0x138, 0xF, 0xF, false);   // wave_shr:1
    if (lane == 63) s_last[wv] = v[L - 1];
    __syncthreads();
    if (lane == 0) {
        if (wv > 0) prev = s_last[wv - 1];
        else if (chunk > 0) { uint32_t pv[L]; pack32(s0 - 32, pv); prev = pv[L - 1]; }
        else prev = 0;                                            // in front of the stream: header bytes, merged below
    }
    const uint32_t o0 = (uint32_t)(a.base_bits & 31);
#pragma unroll
    for (int j = 0; j < L; j++) {
        const uint32_t hi = j ? v[j - 1] : prev;
        const uint32_t ow = (uint32_t)((((unsigned long long)hi << 32) | v[j]) >> o0);   // (hi << (32-o0)) | (v[j] >> o0)
        const uint32_t i = tid * L + j;
        s_o[i + (i >> 5)] = __builtin_bswap32(ow);
    }
    __syncthreads();
    // drain: the block's HB*L words are contiguous in the output
    const unsigned long long total_syms = a.n;
    const unsigned long long wbase = (a.base_bits >> 5) + (unsigned long long)chunk * (HB * L);
    const unsigned long long end_word = (a.base_bits + total_syms * L + 31) >> 5;                 // one past the last word holding stream bits
    if (chunk > 0 && wbase + (unsigned long long)(HB * L) <= end_word) {
        // interior block: four words per store (the payload is only dword-aligned, which global stores accept)
        for (int q = tid; q < HB * L / 4; q += HB) {
            const int i = 4 * q;                                   // 4 | 32: the four words never straddle a swizzle step
            const uint32_t *sp = s_o + i + (i >> 5);
            st16_a4<(RSN_NT_MASK & 4) != 0>(a.out_words + wbase + i, sp[0], sp[1], sp[2], sp[3]);
        }
    } else
    for (int i = tid; i < HB * L; i += HB) {
        const unsigned long long g = wbase + i;
        if (g >= end_word) break;
        const uint32_t val = s_o[i + (i >> 5)];
        if (g == (a.base_bits >> 5)) {
            // this dword also holds the last header bytes (and the pad bits): merge them in registers
            uint32_t hb = 0;
            for (uint32_t k = (uint32_t)(g * 4); k < a.hdr_len; k++) hb |= (uint32_t)tab[128 + k] << (8 * (k & 3));
            a.out_words[g] = val | hb;
        } else a.out_words[g] = val;
    }
    // the word after the block holds the low o0 bits of this block's last logical word; the NEXT block writes it
    // (as its first word) unless this is the last block
    if (o0 && tid == HB - 1) {
        const unsigned long long g = wbase + (unsigned long long)(HB * L);
        const bool last_block = chunk + 1 == n_chunks;
        if (last_block && g < end_word) a.out_words[g] = __builtin_bswap32(v[L - 1] << (32 - o0));
    }
    __syncthreads();                                               // s_o / s_last are reused by the next chunk
    }
    if (blockIdx.x == 0) {                                         // header bytes in front of the first payload dword
        const uint32_t whole = (uint32_t)(a.base_bits >> 5) * 4;
        uint8_t *o8 = reinterpret_cast<uint8_t *>(a.out_words);
        for (uint32_t k = tid; k < whole && k < a.hdr_len; k += HB) o8[k] = tab[128 + k];
    }
}

// ======================================================================= host side
namespace {

// The rune histogram is 0x110000 counters of which a call uses a few dozen to a few 10^5: the present ones are compacted on the
// device (flags, scan, scatter -- in rune order, which is the order the header wants) and only they cross PCIe; the code table goes
// the other way the same way (a list of present runes scattered into the rune-indexed arrays, which are never cleared: absent runes
// are never looked up).  The full-table round trips cost every non-ASCII call 1.1 ms, a 4 KB one included.
__global__ void k_rune_flags(const unsigned long long *__restrict__ hist, unsigned long long *__restrict__ flag) {
    const uint32_t r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r < kMaxRune) flag[r] = hist[r] != 0;
}
__global__ void k_rune_pairs(const unsigned long long *__restrict__ hist, const unsigned long long *__restrict__ off, HuffSym *__restrict__ out) {
    const uint32_t r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r < kMaxRune && hist[r]) { out[off[r]].rune = r; out[off[r]].freq = hist[r]; }
}
struct RuneCode { uint32_t rune, len; unsigned long long code; };
__global__ void k_rune_table(const RuneCode *__restrict__ list, uint32_t k, unsigned long long *__restrict__ code64, uint8_t *__restrict__ len8) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < k) { const RuneCode e = list[i]; code64[e.rune] = e.code; len8[e.rune] = (uint8_t)e.len; }
}

int hist_ascii_or_rune(Ctx &c, hipStream_t s, const uint8_t *d_in, size_t n, uint32_t n_tiles, uint32_t tile, uint32_t *d_tile_hist,
                       std::vector<HuffSym> &syms, bool &ascii, uint16_t **smask_out) {
    void *p;
    int rc = dev_buf(c, 1, 256 * 8, &p); if (rc) return rc;
    unsigned long long *d_gh = (unsigned long long *)p;
    RSN_HIP(hipMemsetAsync(d_gh, 0, 256 * 8, s));
    const uint32_t grid = (uint32_t)std::min<size_t>(n_tiles, 2048);     // persistent blocks (8 loads in flight / other grid sizes: within noise, r01d A/B)
    if (tile == TILE) RSN_LAUNCH("huff_byte_hist", (k_byte_hist<4, TILE>), dim3(grid), dim3(HB), 0, s, d_in, n, n_tiles, d_tile_hist, d_gh);
    else RSN_LAUNCH("huff_byte_hist", (k_byte_hist<1, SMALL_TILE>), dim3(grid), dim3(HB), 0, s, d_in, n, n_tiles, d_tile_hist, d_gh);
    void *hp; rc = pinned_buf(c, 256 * 8, &hp); if (rc) return rc;
    unsigned long long *h = (unsigned long long *)hp;
    RSN_HIP(hipMemcpyAsync(h, d_gh, 256 * 8, hipMemcpyDeviceToHost, s));
    RSN_HIP(hipStreamSynchronize(s));
    ascii = h[128] == 0;   // no byte >= 0x80 seen
    syms.clear();
    if (ascii) {
        for (uint32_t b = 0; b < 128; b++) if (h[b]) syms.push_back({b, h[b]});
        return RSN_OK;
    }
    // rune path: Go UTF-8 semantics (huffman.go:309)
    rc = dev_buf(c, 2, (size_t)kMaxRune * 8, &p); if (rc) return rc;
    unsigned long long *d_rh = (unsigned long long *)p;
    RSN_HIP(hipMemsetAsync(d_rh, 0, (size_t)kMaxRune * 8, s));
    const size_t rounds = ceil_div(n, ROUND);
    uint16_t *d_smask = nullptr;
    { rc = dev_buf(c, 26, (ceil_div(n, 16) + 2) * 2 + 64, &p); if (rc) return rc; d_smask = (uint16_t *)p; }
    *smask_out = d_smask;
    RSN_LAUNCH("huff_rune_hist", k_rune_hist, dim3((uint32_t)std::min<size_t>(rounds, 4096)), dim3(HB), 0, s, d_in, n, d_rh, d_smask);
    // the present runes, compacted in rune order (see k_rune_flags)
    const size_t bound = std::min<size_t>(n, kMaxRune);                // distinct runes never exceed the input's bytes
    rc = dev_buf(c, 6, (size_t)kMaxRune * 16 + 16 + bound * sizeof(HuffSym) + 64, &p); if (rc) return rc;
    unsigned long long *d_flag = (unsigned long long *)p, *d_off = d_flag + kMaxRune, *d_cnt = d_off + kMaxRune;
    HuffSym *d_pairs = (HuffSym *)(d_cnt + 2);
    const dim3 rg((kMaxRune + 255) / 256);
    RSN_LAUNCH("huff_rune_hist", k_rune_flags, rg, dim3(256), 0, s, d_rh, d_flag);
    rc = scan_u64(c, s, "huff_scan", d_flag, d_off, kMaxRune, d_cnt); if (rc) return rc;
    RSN_LAUNCH("huff_rune_hist", k_rune_pairs, rg, dim3(256), 0, s, d_rh, d_off, d_pairs);
    const bool one_trip = bound * sizeof(HuffSym) <= (1u << 20);      // small inputs: count and pairs in one copy
    rc = pinned_buf(c, 16 + (one_trip ? bound * sizeof(HuffSym) : 0), &hp); if (rc) return rc;
    h = (unsigned long long *)hp;
    RSN_HIP(hipMemcpyAsync(h, d_cnt, 16 + (one_trip ? bound * sizeof(HuffSym) : 0), hipMemcpyDeviceToHost, s));   // (d_pairs follows d_cnt[2])
    RSN_HIP(hipStreamSynchronize(s));
    const size_t present = (size_t)h[0];
    if (!one_trip) {
        rc = pinned_buf(c, 16 + present * sizeof(HuffSym), &hp); if (rc) return rc;
        h = (unsigned long long *)hp;
        RSN_HIP(hipMemcpyAsync(h + 2, d_pairs, present * sizeof(HuffSym), hipMemcpyDeviceToHost, s));
        RSN_HIP(hipStreamSynchronize(s));
    }
    const HuffSym *hs = reinterpret_cast<const HuffSym *>(h + 2);
    syms.assign(hs, hs + present);
    return RSN_OK;
}

}  // namespace

size_t huff_compress_bound(size_t n) {
    // an optimal prefix code never costs more than the fixed-length code: <= 21 bits per rune
    const size_t syms = std::min<size_t>(n, kMaxRune);
    return n * 21 / 8 + syms * 26 + 96;
}

// The encoder in two halves, so that one stream can be produced from several SLICES of the input (rsn_huffman_compress_sharded: one
// slice per worker / device).  huff_slice_hist: the slice's symbol counts (and what the emit pass wants kept: tile histograms, the
// rune-start map).  huff_slice_emit: the slice's code bits at a given bit position of d_out, from the tree and codes of the WHOLE
// input.  huff_encode_dev is the two in a row for a single slice.
int huff_slice_hist(Ctx &c, hipStream_t s, const uint8_t *d_in, size_t n, HuffSlice &sl) {
    sl.tile = n <= SMALL_INPUT ? SMALL_TILE : TILE;
    sl.n_tiles = (uint32_t)ceil_div(n, sl.tile);
    void *p;
    int rc = dev_buf(c, 0, (size_t)sl.n_tiles * 128 * 4, &p); if (rc) return rc;
    sl.d_tile_hist = (uint32_t *)p;
    sl.d_smask = nullptr;                                                 // rune path: which positions start a rune (k_rune_hist's classification, kept)
    rc = hist_ascii_or_rune(c, s, d_in, n, sl.n_tiles, sl.tile, sl.d_tile_hist, sl.syms, sl.ascii, &sl.d_smask); if (rc) return rc;
    if (!sl.ascii && n <= RUNE_SMALL_INPUT) {   // the rune path has no per-tile histograms to pay for: small tiles longer (8 MiB: 413 -> 353 us)
        sl.tile = SMALL_TILE;
        sl.n_tiles = (uint32_t)ceil_div(n, sl.tile);
    }
    return RSN_OK;
}

// hdr: the bytes in front of the payload that THIS slice writes (header || "\\\n" || pad byte for the first slice, nothing for the
// others); base_bits: where the slice's first code bit goes, counted from d_out; slice_bits: how many it writes; flat: every code of the
// (all-ASCII) alphabet has the same length.  Synchronises the stream.
int huff_slice_emit(Ctx &c, hipStream_t s, const uint8_t *d_in, size_t n, const HuffSlice &sl, const HuffTree &tree, const HuffCodes &codes, bool flat,
                    const std::string &hdr, unsigned long long base_bits, unsigned long long slice_bits, uint8_t *d_out) {
    void *p; int rc;
    const size_t H = hdr.size();
    const uint32_t tile = sl.tile, n_tiles = sl.n_tiles;
    if (slice_bits == 0) {   // one distinct symbol: code "" (huffman.go:110-116), no payload bytes
        if (H) RSN_HIP(hipMemcpyAsync(d_out, hdr.data(), H, hipMemcpyHostToDevice, s));
        RSN_HIP(hipStreamSynchronize(s));
        return RSN_OK;
    }

    // ---- flat code: offsets are arithmetic (decided before any table is uploaded: this path needs none)
    if (flat) {
        const unsigned L = codes.max_len;
        FlatEmitArgs fa{};
        fa.in = d_in; fa.n = n; fa.base_bits = base_bits; fa.out_words = (uint32_t *)d_out; fa.hdr_len = (uint32_t)H;
        uint8_t *hcodes = fa.inl;
        if (128 + H > (size_t)FE_INLINE) {                         // a header too long for the argument segment (10-digit counts): upload
            void *hp2; rc = pinned_buf(c, 128 + H + 16, &hp2); if (rc) return rc;
            hcodes = (uint8_t *)hp2;
            memset(hcodes, 0, 128);
        }
        for (uint32_t i = 0; i < tree.n_leaves; i++) hcodes[tree.rune[i]] = (uint8_t)codes.code[i];
        memcpy(hcodes + 128, hdr.data(), H);
        if (hcodes != fa.inl) {
            rc = dev_buf(c, 3, 128 + H + 16, &p); if (rc) return rc;
            RSN_HIP(hipMemcpyAsync(p, hcodes, 128 + H, hipMemcpyHostToDevice, s));
            fa.codes = (const uint8_t *)p;
        }
        const dim3 grid((uint32_t)std::min<size_t>(ceil_div(n, FE_SYMS), 256 * 8));
        switch (L) {
#define RSN_FE(LL) case LL: RSN_LAUNCH("huff_emit", k_emit_flat<LL>, grid, dim3(HB), 0, s, fa); break;
            RSN_FE(1) RSN_FE(2) RSN_FE(3) RSN_FE(4) RSN_FE(5) RSN_FE(6) RSN_FE(7)
#undef RSN_FE
            default: break;
        }
        RSN_HIP(hipStreamSynchronize(s));
        return RSN_OK;
    }

    // ---- code tables.  (An all-ASCII slice of an input that has runes elsewhere still takes the byte-indexed tables: its own bytes' codes.)
    const int mode = !sl.ascii ? MODE_RUNE : (codes.max_len <= (unsigned)TAB_LEN_SHIFT ? MODE_ASCII : MODE_ASCII_WIDE);
    EmitArgs a{};
    uint8_t *d_len8 = nullptr;
    if (mode == MODE_RUNE) {
        // the present runes' codes go up as a list and are scattered into the rune-indexed arrays on the device (see k_rune_flags)
        const uint32_t k = tree.n_leaves;
        void *hp; rc = pinned_buf(c, (size_t)k * sizeof(RuneCode), &hp); if (rc) return rc;
        RuneCode *hl = (RuneCode *)hp;
        for (uint32_t i = 0; i < k; i++) hl[i] = RuneCode{tree.rune[i], codes.len[i], codes.code[i]};
        rc = dev_buf(c, 3, (size_t)kMaxRune * 9 + 64 + (size_t)k * sizeof(RuneCode), &p); if (rc) return rc;
        RuneCode *d_list = (RuneCode *)((uint8_t *)p + round_up((size_t)kMaxRune * 9, 64));
        RSN_HIP(hipMemcpyAsync(d_list, hl, (size_t)k * sizeof(RuneCode), hipMemcpyHostToDevice, s));
        a.code64 = (const unsigned long long *)p;
        d_len8 = (uint8_t *)p + (size_t)kMaxRune * 8;
        RSN_LAUNCH("huff_rune_table", k_rune_table, dim3((k + 255) / 256), dim3(256), 0, s, (const RuneCode *)d_list, k, (unsigned long long *)p, d_len8);
    } else {
        struct Tab { uint32_t t32[256]; unsigned long long c64[256]; uint8_t l8[256]; };
        void *hp; rc = pinned_buf(c, sizeof(Tab), &hp); if (rc) return rc;
        Tab *ht = (Tab *)hp;
        memset(ht, 0, sizeof *ht);
        for (uint32_t i = 0; i < tree.n_leaves; i++) {
            const uint32_t r = tree.rune[i];
            if (r >= 256) continue;                                // (a rune of another slice)
            ht->c64[r] = codes.code[i]; ht->l8[r] = codes.len[i];
            if (codes.max_len <= (unsigned)TAB_LEN_SHIFT) ht->t32[r] = ((uint32_t)codes.len[i] << TAB_LEN_SHIFT) | (uint32_t)codes.code[i];
        }
        rc = dev_buf(c, 3, sizeof(Tab), &p); if (rc) return rc;
        RSN_HIP(hipMemcpyAsync(p, ht, sizeof(Tab), hipMemcpyHostToDevice, s));
        Tab *dt = (Tab *)p;
        a.tab32 = dt->t32; a.code64 = dt->c64;
        d_len8 = dt->l8;
    }
    a.len8 = d_len8;

    // ---- tile bit offsets
    rc = dev_buf(c, 4, ((size_t)n_tiles * 2 + 2) * 8, &p); if (rc) return rc;
    unsigned long long *d_tile_bits = (unsigned long long *)p;
    unsigned long long *d_tile_off = d_tile_bits + n_tiles;
    if (mode == MODE_RUNE) {
        RSN_LAUNCH("huff_tile_bits_rune", k_tile_bits_rune, dim3(std::min<uint32_t>(n_tiles, 4096)), dim3(HB), 0, s, d_in, n, d_len8, n_tiles, tile, d_tile_bits, (const uint16_t *)sl.d_smask);
    } else {
        RSN_LAUNCH("huff_tile_bits", k_tile_bits, dim3((uint32_t)ceil_div(n_tiles, HB / 64)), dim3(HB), 0, s, sl.d_tile_hist, d_len8, n_tiles, d_tile_bits);
    }
    rc = scan_u64(c, s, "huff_scan", d_tile_bits, d_tile_off, n_tiles, d_tile_off + n_tiles); if (rc) return rc;

    // ---- emit
    const uint32_t tiles_per_block = (uint32_t)std::max<size_t>(1, ceil_div(n_tiles, 2048));
    const uint32_t n_blocks = (uint32_t)ceil_div(n_tiles, tiles_per_block);
    a.in = d_in; a.n = n; a.tile_off = d_tile_off; a.base_bits = base_bits;
    a.tiles_per_block = tiles_per_block; a.n_tiles = n_tiles; a.out_words = (uint32_t *)d_out; a.tile = tile; a.smask = sl.d_smask;
    RSN_LAUNCH("huff_emit_init", k_emit_init, dim3((uint32_t)ceil_div(n_blocks + 1, 256)), dim3(256), 0, s,
               (uint32_t *)d_out, d_tile_off, base_bits, tiles_per_block, n_tiles, n_blocks, base_bits + slice_bits);
    if (H) RSN_HIP(hipMemcpyAsync(d_out, hdr.data(), H, hipMemcpyHostToDevice, s));
    if (mode == MODE_ASCII) RSN_LAUNCH("huff_emit", k_emit_ascii32<32>, dim3(n_blocks), dim3(HB), 0, s, a);   // (32 symbols a lane: 48 / 64 measured no better, r02)
    else if (mode == MODE_ASCII_WIDE) RSN_LAUNCH("huff_emit_wide", k_emit<MODE_ASCII_WIDE>, dim3(n_blocks), dim3(HB), 0, s, a);
    else RSN_LAUNCH("huff_emit_rune", k_emit<MODE_RUNE>, dim3(n_blocks), dim3(HB), 0, s, a);
    RSN_HIP(hipStreamSynchronize(s));   // hdr (host memory) must outlive the copy
    return RSN_OK;
}

bool huff_flat_code(const HuffTree &tree, const HuffCodes &codes) {
    static const bool no_flat_emit = getenv("RSN_NO_FLAT") != nullptr;
    if (no_flat_emit || codes.min_len != codes.max_len || codes.max_len > 7 || codes.max_len == 0) return false;
    for (uint32_t i = 0; i < tree.n_leaves; i++) if (tree.rune[i] >= 0x80) return false;
    return true;
}

// Encodes d_in[0..n) into d_out.  On RSN_ERR_CAPACITY *out_n holds the needed capacity.
int huff_encode_dev(Ctx &c, hipStream_t s, const uint8_t *d_in, size_t n, uint8_t *d_out, size_t out_cap, size_t *out_n,
                    HuffTree *tree_out, HuffCodes *codes_out) {
    if (n == 0) return c.fail(RSN_ERR_EMPTY, "huffman: empty input (reference panics in heap.Pop, huffman.go:102)");
    if (((uintptr_t)d_in & 15) || ((uintptr_t)d_out & 15)) return c.fail(RSN_ERR_ARG, "huffman: device buffers must be 16-byte aligned");
    static const bool host_timing = getenv("RSN_HOST_TIMING") != nullptr;   // prints where the host side of a call spends its time
    auto now = [] { return std::chrono::steady_clock::now(); };
    auto ms = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) { return std::chrono::duration<double, std::milli>(b - a).count(); };
    const auto t0 = now();
    HuffSlice sl;
    int rc = huff_slice_hist(c, s, d_in, n, sl); if (rc) return rc;
    const auto t1 = now();

    std::string hdr;
    HuffTree tree; HuffCodes codes; std::string msg;
    // (rune alphabets of 10^5 symbols -- config 2b: 3 * 10^5 -- spend 2 ms on the header's decimal counts and 20 ms in the Go-exact
    //  heap: the header is written by a second thread meanwhile, from a copy of the table -- build_tree reorders its own)
    std::vector<HuffSym> by_rune;
    std::optional<SideJob> hdr_writer;                                  // (a pooled helper, rsn_helpers.h; on this thread when none is to be had)
    if (sl.syms.size() > 4096) { by_rune = sl.syms; hdr_writer.emplace([&] { emit_header(by_rune, hdr); }); }
    else emit_header(sl.syms, hdr);   // syms is ascending by rune here
    const auto t2 = now();
    const bool tree_ok = build_tree(sl.syms, tree, msg);
    if (hdr_writer) hdr_writer->finish();
    if (!tree_ok) return c.fail(RSN_ERR_EMPTY, "%s", msg.c_str());
    const auto t3 = now();
    if (!assign_codes(tree, codes, msg, codes_out != nullptr)) return c.fail(RSN_ERR_LIMIT, "%s", msg.c_str());
    const auto t4 = now();
    if (host_timing) fprintf(stderr, "huffman encode host: histogram (kernels + D2H + compaction) %.2f ms, header %.2f ms, tree %.2f ms, codes %.2f ms, %u symbols\n",
                             ms(t0, t1), ms(t1, t2), ms(t2, t3), ms(t3, t4), tree.n_leaves);
    if (tree_out) *tree_out = tree;
    if (codes_out) *codes_out = codes;
    if (!d_out) return RSN_OK;   // table introspection only

    const unsigned pad = (unsigned)((8 - codes.total_bits % 8) % 8);            // huffman.go:245-249
    hdr.append("\\\n");
    hdr.push_back((char)pad);
    const size_t H = hdr.size();
    const size_t total = H + (size_t)((codes.total_bits + pad) / 8);
    *out_n = total;
    const size_t need = round_up(total, 16) + 32;
    if (need > out_cap) { *out_n = need; return c.fail(RSN_ERR_CAPACITY, "huffman: output needs %zu bytes, buffer holds %zu", need, out_cap); }
    return huff_slice_emit(c, s, d_in, n, sl, tree, codes, sl.ascii && huff_flat_code(tree, codes), hdr, 8ull * H + pad, codes.total_bits, d_out);
}

}  // namespace rsn
