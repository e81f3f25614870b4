// huff_small.hip -- the Huffman codec for a host buffer of at most 64 KiB (BASELINE configs[0]: the reference's own README case).
//
// At this size a call is its fixed costs: the general path is ~10 launches, five copy commands and three host round trips for work the
// device does in a few microseconds (r04: 89 us to compress 64 KiB, 129 us to decompress it).  Here a call is
//   compress    host memcpy into pinned memory -> k_small_hist (a block per 2 KiB tile reads it over PCIe, keeps a device copy, writes counts
//               straight back into pinned memory) -> the Go-exact tree, codes and header on the host (huffman.go:58-127,312-318)
//               -> k_small_emit (code table, header and tile bit positions in the KERNEL ARGUMENTS; a block per 2 KiB tile stores its words
//               into pinned memory) -> host memcpy into the result block:                    2 launches, no copy command
//   decompress  header and tree on the host (the stream is in host memory: huffman.go:196-227,261) -> host memcpy of the stream into pinned
//               memory -> k_small_dec (up to 32 blocks; the tree in the kernel arguments, every block builds its lookup table; lanes
//               take subsequences and settle where their codewords start -- see the kernel; decoded bytes into pinned memory)
//               -> host memcpy:                                                              1 launch, no copy command
// The host does not wait for the stream either: every block ends by setting a word in pinned memory that the host polls (block_done).
// 64 KiB of the README's text, host buffer to host buffer: compress 89 -> 32 us, decompress 129 -> 52 us (r05).
// For byte alphabets (every symbol < 0x80); everything else -- runes, a single symbol,
// foreign headers, malformed streams and their error texts -- returns 1 and takes the general path, which words the errors.
#include <atomic>
#include <chrono>

#include "codecs.h"
#include "huff_host.h"
#include "huff_pathmap.h"

namespace rsn {
namespace {

constexpr uint32_t SMALL_MAX = 65536;           // bytes of input (compress) / of output (decompress)
constexpr uint32_t HDR_MAX = 1100;              // 128 entries of at most 5 digits + '|' + 2 bytes, + "\\\n" + pad
constexpr uint32_t DEC_STREAM_MAX = 65536 + 2048;   // bytes of a stream the decoder takes
constexpr int DEC_K = 9;                        // index bits of the decoder's table, at most (every block builds the table: 11 bits cost 2 us more than they save on 64 KiB)
constexpr int DEC_ROUNDS = 64;                  // rounds of the synchronisation before the general decoder is asked instead

// pinned staging of one call (Ctx::pinned): offsets
constexpr size_t PIN_IN = 0;                                        // the caller's bytes, zero-padded to 16
constexpr size_t PIN_OUT = PIN_IN + DEC_STREAM_MAX + 128;          // what the kernels produced
constexpr size_t PIN_CNT = PIN_OUT + SMALL_MAX + 64;               // encoder: every tile's 128 counts (u16), then a word per tile: a byte >= 0x80 was seen; decoder: status words
constexpr size_t PIN_BYTES = PIN_CNT + 32 * 128 * 2 + 2 * 32 * 4 + 256;
static_assert(PIN_OUT % 16 == 0 && PIN_CNT % 16 == 0, "16-byte stores");

// A block's last act: its stores to host memory made visible, then ONE word the host is polling (the host does not wait for the stream: a
// hipStreamSynchronize is 5-10 us of wake-up, the kernel is as long).
constexpr uint32_t FLAG_PENDING = 0xFFFFFFFFu;
__device__ __forceinline__ void block_done(uint32_t *flag, uint32_t value) {
    __threadfence_system();
    __syncthreads();
    if (threadIdx.x == 0) __hip_atomic_store(flag, value, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

// ---------------------------------------------------------------- compress, kernel 1: counts (huffman.go:306-311) + a device copy
// One block per tile of 2 KiB, a 16-byte load per lane straight from the caller's bytes in pinned host memory (one PCIe round trip for the
// whole kernel: r05, one block looping over 64 KiB: 15 us, four dependent round trips).  The tile's counts go back to pinned memory as
// they are: the host adds them up (the alphabet's counts) and, once it has the code lengths, turns them into every tile's bit position.
constexpr uint32_t ST = 2048;                   // bytes per tile
constexpr uint32_t ST_MAX = SMALL_MAX / ST;     // tiles
__global__ __launch_bounds__(128) void k_small_hist(const uint4 *__restrict__ hin, uint32_t n, uint4 *__restrict__ d_copy, uint16_t *__restrict__ tile_hist, uint32_t *__restrict__ tile_high) {
    __shared__ uint32_t s_h[2][128];
    __shared__ uint32_t s_high;
    const uint32_t tid = threadIdx.x, wave = tid >> 6, at = blockIdx.x * ST + tid * 16;
    uint4 v = make_uint4(0, 0, 0, 0);
    if (at < n) v = hin[at / 16];                                      // (host memory; the pad behind n is zero)
    s_h[0][tid] = 0; s_h[1][tid] = 0;
    if (tid == 0) s_high = 0;
    __syncthreads();
    if (at < n) {
        d_copy[at / 16] = v;
        const uint32_t w[4] = {v.x, v.y, v.z, v.w};
        const uint32_t valid = min(16u, n - at);
        uint32_t high = 0;
#pragma unroll
        for (int j = 0; j < 16; j++) {
            const uint32_t b = (w[j >> 2] >> (8 * (j & 3))) & 0xFF;
            if ((uint32_t)j < valid) { high |= b & 0x80; atomicAdd(&s_h[wave][b & 0x7F], 1u); }
        }
        if (high) s_high = 1;
    }
    __syncthreads();
    tile_hist[blockIdx.x * 128 + tid] = (uint16_t)(s_h[0][tid] + s_h[1][tid]);
    block_done(&tile_high[blockIdx.x], s_high);
}

// ---------------------------------------------------------------- compress, kernel 2: the stream (huffman.go:229-256,174-191)
// One block per tile again.  Code table, header and every tile's first bit position come in the KERNEL ARGUMENTS (no upload); a block packs
// its tile's codes into an LDS image of the words they touch and stores the words it owns into pinned host memory.  Two neighbours share
// the word a tile boundary falls into: it belongs to the LATER tile, which works out the earlier one's last few bits itself (from the up to
// 31 symbols before its first) -- no atomics on memory, no zeroed output, nothing between the blocks.
struct SmallEmitArgs {
    const uint8_t *d_copy; uint32_t *hout; uint32_t *done;   // done[tile]: FLAG_PENDING until the tile's words are out
    uint32_t n, H, total;                // input bytes; header bytes incl. "\\\n" and the pad byte; bytes of the stream
    uint32_t P[ST_MAX + 1];              // first code bit of every tile, counted from the stream's first byte (P[0] = 8 H + pad)
    uint32_t tab[128];                   // len << 24 | code (a 64 KiB input cannot produce a code beyond 22 bits)
    uint8_t hdr[HDR_MAX + 4];
};

template <int WAVES>
__device__ __forceinline__ uint32_t block_excl_scan(uint32_t v, uint32_t *s_wave /*[WAVES + 1]*/) {
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint32_t inc = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) { const uint32_t o = __shfl_up(inc, d, 64); if (lane >= (uint32_t)d) inc += o; }
    if (lane == 63) s_wave[wave] = inc;
    __syncthreads();
    if (threadIdx.x == 0) { uint32_t acc = 0; for (int w = 0; w < WAVES; w++) { const uint32_t t = s_wave[w]; s_wave[w] = acc; acc += t; } s_wave[WAVES] = acc; }
    __syncthreads();
    return s_wave[wave] + inc - v;
}

constexpr uint32_t EMIT_IMG_WORDS = ST * 24 / 32 + (HDR_MAX + 3) / 4 + 8;
__global__ __launch_bounds__(256) void k_small_emit(SmallEmitArgs a) {
    __shared__ uint32_t s_img[EMIT_IMG_WORDS];       // the words this tile's bits fall into, in memory order (block 0: from the stream's first byte)
    __shared__ uint32_t s_tab[128];
    __shared__ uint32_t s_wave[5];
    __shared__ uint32_t s_prev[32];
    const uint32_t tid = threadIdx.x, b = blockIdx.x, n_tiles = gridDim.x;
    const uint32_t Pb = a.P[b], Pn = a.P[b + 1];
    const uint32_t W0 = b ? Pb >> 5 : 0u;                                        // the image's first word
    const uint32_t img_words = (Pn - (W0 << 5) + 31) >> 5;
    const uint32_t hw = b ? 0u : (a.H + 3) / 4;
    for (uint32_t w = tid; w < img_words + 1; w += 256) {
        uint32_t v = 0;
        if (w < hw) {
#pragma unroll
            for (int k = 0; k < 4; k++) { const uint32_t at = 4 * w + k; if (at < a.H) v |= (uint32_t)a.hdr[at] << (8 * k); }
        }
        s_img[w] = v;
    }
    if (tid < 128) s_tab[tid] = a.tab[tid];
    const uint32_t need = b ? Pb & 31 : 0u;                                      // bits of the tile before in this tile's first word
    if (need && tid < 32) s_prev[tid] = a.tab[a.d_copy[b * ST - 1 - tid] & 0x7F];   // (the tile before is a whole one: 2048 symbols, a bit each at least)
    // eight symbols a lane
    const uint32_t lo = b * ST + tid * 8;
    uint2 v = make_uint2(0, 0);
    if (lo < a.n) v = *reinterpret_cast<const uint2 *>(a.d_copy + lo);
    const uint32_t valid = lo < a.n ? min(8u, a.n - lo) : 0u;
    __syncthreads();
    uint32_t e[8], bits = 0;
#pragma unroll
    for (int j = 0; j < 8; j++) {
        const uint32_t byte = ((j < 4 ? v.x : v.y) >> (8 * (j & 3))) & 0x7F;
        e[j] = (uint32_t)j < valid ? s_tab[byte] : 0u;
        bits += e[j] >> 24;
    }
    const uint32_t pos0 = Pb - (W0 << 5) + block_excl_scan<4>(bits, s_wave);
    if (bits) {
        uint32_t w = pos0 >> 5, nacc = pos0 & 31;                     // the word being filled; bits of it that are in `acc` (the first nacc: a neighbour's, zero here)
        unsigned long long acc = 0;                                    // left-aligned
        bool first = true;
#pragma unroll
        for (int j = 0; j < 8; j++) {
            const uint32_t l = e[j] >> 24;
            acc |= (unsigned long long)(e[j] & 0xFFFFFFu) << (64 - nacc - l);      // (l == 0: nothing)
            nacc += l;
            if (nacc >= 32) {
                const uint32_t be = __builtin_bswap32((uint32_t)(acc >> 32));
                if (first) atomicOr(&s_img[w], be); else s_img[w] = be;     // (the first word may be shared with the lane before)
                first = false;
                w++; acc <<= 32; nacc -= 32;
            }
        }
        if (nacc) atomicOr(&s_img[w], __builtin_bswap32((uint32_t)(acc >> 32)));   // (... the last with the lane after)
    }
    if (need && tid == 0) {                                            // the tile before ends in this tile's first word
        unsigned long long tail = 0;
        uint32_t have = 0;
        for (uint32_t k = 0; have < need; k++) { const uint32_t ent = s_prev[k]; tail |= (unsigned long long)(ent & 0xFFFFFFu) << have; have += ent >> 24; }
        const uint32_t t32 = (uint32_t)tail & ((1u << need) - 1u);
        atomicOr(&s_img[0], __builtin_bswap32(t32 << (32 - need)));
    }
    __syncthreads();
    // the words this tile owns: up to, not including, the one the next tile starts in; the last tile: to the stream's end
    const uint32_t own = b + 1 < n_tiles ? (Pn >> 5) - W0 : (a.total + 3) / 4 - W0;
    for (uint32_t i = tid; i < own; i += 256) a.hout[W0 + i] = s_img[i];
    block_done(&a.done[b], 0);
}

// ---------------------------------------------------------------- decompress (huffman.go:131-153,258-297)
// The payload is one bit string without an index: a lane that takes the subsequence [lo, lo + S) does not know where its first codeword
// starts.  It starts at lo and trusts the code to synchronise; the lane before tells it where it really ended, the lane redoes its
// subsequence from there, and so on until nothing changes.  On most data two rounds settle it.  On PERIODIC data (the README's samiam.txt
// repeated to 64 KiB is that) a wrong start can stay a wrong parse for kilobytes -- a second phase the code is just as happy in -- and
// "redo from the predecessor's exit" then walks the stretch one lane a round (r05, first version: one block, 52 rounds, 48 of its 94 us).
// So a lane keeps a small MAP instead of one answer: for every start a predecessor might hand it (at most four), where it ends and how many
// symbols it takes.  A round only adds the starts not seen before -- with two phases every lane has both after two rounds, whoever is
// right -- and when no lane learns a new start the true path is a composition of the maps along the lanes (a wavefront scan).
// One CU decodes 64 KiB in 12 us a pass however it is arranged, so the lanes are spread over up to 16 BLOCKS, and the blocks must not wait
// for each other round by round: a block's FIRST lane answers all 32 starts its subsequence could be entered at (32 lanes, one each), the
// block composes its lanes into a map "entered at c -> left at c', n symbols", publishes it, and only then looks at the blocks before it
// (their maps, composed, give its true entry and its first output byte).  One wait per call, behind work that every block does alone.
constexpr int DL = 256;                          // lanes per block that take a subsequence
constexpr int DT = DL + 64;                      // ... and a fifth wavefront: the 32 entries of the block's first lane, beside the others' first guesses
constexpr uint32_t DEC_BLOCKS = 32;
constexpr uint32_t DEC_S_MAX = 96;               // bits per lane at most: (64 KiB + 2 KiB) * 8 / 8192 lanes, in whole words (at least 64: an exit lies within 32 bits of the next lane's first)
constexpr uint32_t DEC_PAY_WORDS = DL * DEC_S_MAX / 32 + 8;
constexpr uint32_t DEC_OUT_CAP = 32768;          // bytes one block may produce (text: 256 lanes * 96 bits / 3 bits = 8 KiB)
constexpr uint32_t OFF_BAD = 0xFF;               // an exit that is none: the path ran off the payload (or the entry cannot occur)

struct SmallDecArgs {
    const uint32_t *pay;      // the stream from a 4-byte boundary at or before its first payload byte (pinned host memory; zero behind its end)
    uint32_t pay_words;
    uint32_t p0, end;         // first code bit / the bit behind the last, counted from `pay`
    uint32_t K, root, n_child;  // index bits of the lookup table; the tree's root (an internal node)
    uint32_t S, T;            // bits per lane; lanes that have a subsequence
    uint32_t flat;            // every code has this length (0: lengths differ): the boundaries are where the arithmetic says -- as many phases as the code has bits, and none to find
    uint32_t seq;             // this call's number: what a block's flag holds once its map is out
    uint8_t *hout; uint32_t *status;      // status[b]: FLAG_PENDING, then 0 = block b done, 1 = not for this kernel; status[DEC_BLOCKS]: decoded bytes
    uint32_t *g_maps, *g_flags;           // device memory: [block][32] (exit << 24 | symbols), [block]
    uint16_t child[256];      // [2 * node + bit]: 0x8000 | byte for a leaf, else the internal node
};

// a reader of the big-endian words in LDS: the next bits left-aligned in a register, a word fetched for every 32 consumed
struct BitReader {
    const uint32_t *pay; unsigned long long buf; uint32_t pos, nextw; int avail;
    __device__ __forceinline__ void seek(uint32_t p) {
        const uint32_t w = p >> 5, o = p & 31;
        buf = (((unsigned long long)pay[w] << 32) | pay[w + 1]) << o;
        avail = 64 - (int)o; nextw = w + 2; pos = p;
    }
    __device__ __forceinline__ void skip(uint32_t l) {
        buf <<= l; avail -= (int)l; pos += l;
        if (avail < 32) { buf |= (unsigned long long)pay[nextw++] << (32 - avail); avail += 32; }
    }
};
struct DecTab { const uint32_t *lut; const uint16_t *child; uint32_t K, end; };
// the codeword at the reader's position: its byte; the reader moves behind it (past `end`: the caller's to notice)
// (an entry of the table: one codeword or two, see the kernel's table walk)
__device__ __forceinline__ uint32_t ent_len1(uint32_t e) { return (e >> 16) & 31u; }
__device__ __forceinline__ uint32_t ent_len2(uint32_t e) { return (e >> 21) & 31u; }
__device__ __forceinline__ bool ent_two(uint32_t e) { return (e >> 26) & 1u; }
// a codeword longer than the table's K bits, from the internal node its first K bits lead to
__device__ __forceinline__ uint32_t dec_long(BitReader &r, const DecTab &t, uint32_t e) {
    uint32_t node = e & 0xFFFF;
    r.skip(t.K);
    for (;;) {
        const uint32_t bit = (uint32_t)(r.buf >> 63);
        r.skip(1);
        node = t.child[2 * node + bit];
        if ((node & 0x8000) || r.pos > t.end + 64) break;             // (garbage behind the end: stop)
    }
    return node & 0xFF;
}

// The kernel is launched for a few microseconds of work on CUs whose instruction caches are cold: its time is its CODE SIZE (r05: the
// first multi-block version, every loop inlined wherever it was used -- 36 KB of instructions, 56 us; the arithmetic is two).  So the two
// loops that walk codewords exist ONCE, as functions.
// the codewords that start in [from, hi): exit offset (from hi; OFF_BAD: the path ran off the payload) << 24 | their number
__device__ __noinline__ uint32_t run_path(const uint32_t *pay, const uint32_t *lut, const uint16_t *child, uint32_t K, uint32_t end, uint32_t from, uint32_t hi) {
    const DecTab tab{lut, child, K, end};
    BitReader r; r.pay = pay; r.seek(from);
    uint32_t c = 0;
    while (r.pos < hi) {
        const uint32_t e = lut[(uint32_t)(r.buf >> (64 - K))];
        if (e >> 31) { (void)dec_long(r, tab, e); c++; }
        else if (ent_two(e) && r.pos + ent_len1(e) < hi) { r.skip(ent_len2(e)); c += 2; }     // (the second one starts in this subsequence too)
        else { r.skip(ent_len1(e)); c++; }
        if (r.pos > end) break;
    }
    return (r.pos > end ? OFF_BAD : r.pos - hi) << 24 | c;
}
// `count` codewords from `from`, their bytes to out[0 ..)
__device__ __noinline__ void emit_path(const uint32_t *pay, const uint32_t *lut, const uint16_t *child, uint32_t K, uint32_t end, uint32_t from, uint32_t count, uint8_t *out) {
    const DecTab tab{lut, child, K, end};
    BitReader r; r.pay = pay; r.seek(from);
    for (uint32_t i = 0; i < count;) {
        const uint32_t e = lut[(uint32_t)(r.buf >> (64 - K))];
        if (e >> 31) out[i++] = (uint8_t)dec_long(r, tab, e);
        else if (ent_two(e) && i + 1 < count) { out[i] = (uint8_t)e; out[i + 1] = (uint8_t)(e >> 8); i += 2; r.skip(ent_len2(e)); }
        else { out[i++] = (uint8_t)e; r.skip(ent_len1(e)); }
    }
}
__device__ __forceinline__ uint32_t byte_of(uint32_t packed, uint32_t j) { return (packed >> (8 * j)) & 0xFF; }
__global__ __launch_bounds__(DT) void k_small_dec(SmallDecArgs a) {
    __shared__ uint32_t s_pay[DEC_PAY_WORDS];                 // big-endian words of this block's part of the stream
    __shared__ uint32_t s_lut[1u << DEC_K];                   // byte | second byte << 8 | length << 16 | length of both << 21 | two << 26, or 0x80000000 | internal node reached after K bits
    __shared__ uint16_t s_child[256];
    __shared__ uint32_t s_st[DL], s_ex[DL], s_n[DL];          // per lane: its starts (offsets from its subsequence's first bit, a byte each), the exits that belong to them (offsets from the next subsequence's first bit), their number
    __shared__ uint32_t s_exmask;                             // ... the exits that occur among them, a bit each
    __shared__ uint32_t s_ex32[32], s_cn32[32];               // the block's first lane: exit and symbols for every start in its first 32 bits
    __shared__ uint32_t s_wto[5], s_wc[5][4];                 // the wavefronts' maps
    __shared__ uint32_t s_blk[4][2];                          // entry j of lane 1 -> (exit of the block's last lane, symbols of lanes 1..)
    __shared__ uint32_t s_all[DEC_BLOCKS * 32];               // the maps of the blocks before this one
    __shared__ uint32_t s_true[3];                            // this block's true entry (a start of its first lane), its first output byte, a failure
    __shared__ __attribute__((aligned(16))) uint8_t s_out[DEC_OUT_CAP + 32];
    const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, b = blockIdx.x, n_blk = gridDim.x;
    const uint32_t S = a.S;
    const uint32_t blk_lo = a.p0 + b * DL * S;                                   // this block's first bit
    const uint32_t wlo = blk_lo >> 5;                                           // ... the word it is in: bit positions below are counted from it
    const uint32_t n_words = DL * S / 32 + 6;
    {   // every load of the staging in flight at once (the bytes are in host memory, a round trip is microseconds); the table meanwhile
        constexpr int PER = (DEC_PAY_WORDS + DT - 1) / DT;
        uint32_t v[PER];
        const uint32_t ch = tid < a.n_child ? a.child[tid] : 0u;               // (asked for first: the table is built while the stream's words are on their way)
#pragma unroll
        for (int k = 0; k < PER; k++) { const uint32_t i = tid + k * DT; v[k] = (i < n_words && wlo + i < a.pay_words) ? a.pay[wlo + i] : 0u; }
        if (tid < a.n_child) s_child[tid] = (uint16_t)ch;
        __syncthreads();
        // window v of K bits: down the tree from the root, two windows a lane at a time (the steps depend on each other, the windows do
        // not); a leaf met with bits to spare sends the walk back to the root for a SECOND codeword: an entry holds up to two
        constexpr int G = 2;
        for (uint32_t v0 = tid * G; v0 < (1u << a.K); v0 += DT * G) {
            uint32_t node[G], ent[G], got[G];
#pragma unroll
            for (int q = 0; q < G; q++) { node[q] = a.root; ent[q] = 0; got[q] = 0; }
            for (uint32_t d = 0; d < a.K; d++) {
#pragma unroll
                for (int q = 0; q < G; q++) {
                    if (got[q] == 2) continue;
                    const uint32_t bit = ((v0 + q) >> (a.K - 1 - d)) & 1;
                    node[q] = s_child[2 * node[q] + bit];
                    if (node[q] & 0x8000) {
                        if (got[q] == 0) ent[q] = (node[q] & 0xFF) | (d + 1) << 16;                     // byte, length
                        else ent[q] |= (node[q] & 0xFF) << 8 | (d + 1) << 21 | 1u << 26;             // second byte, length of both, "two"
                        got[q]++;
                        node[q] = a.root;
                    }
                }
            }
#pragma unroll
            for (int q = 0; q < G; q++) if (v0 + q < (1u << a.K)) s_lut[v0 + q] = got[q] ? ent[q] : (0x80000000u | node[q]);
        }
#pragma unroll
        for (int k = 0; k < PER; k++) { const uint32_t i = tid + k * DT; if (i < n_words + 2) s_pay[i] = i < n_words ? __builtin_bswap32(v[k]) : 0u; }
    }
    if (tid < 32) { s_ex32[tid] = OFF_BAD; s_cn32[tid] = 0; }
    if (tid == 0) s_exmask = 0;
    __syncthreads();
    const uint32_t base_bit = wlo << 5;
    const uint32_t end = a.end - base_bit;                                      // (all positions from here on: bits from s_pay[0])
    const uint32_t g = b * DL + tid;                                            // the lane's subsequence
    const uint32_t L = min((uint32_t)DL, a.T - b * DL);                         // lanes of this block that have one
    const bool real = tid < L;                                                  // (the fifth wavefront's lanes: none)
    const uint32_t my_lo = blk_lo - base_bit + tid * S;
    const uint32_t my_hi = min(end, my_lo + S);                                 // (the next lane's first bit; the stream's last lane: the end)
    auto run_in = [&](uint32_t lo, uint32_t hi, uint32_t off, uint32_t &exit_off, uint32_t &count) {   // the codewords that start in [lo + off, hi)
        const uint32_t r = run_path(s_pay, s_lut, s_child, a.K, end, lo + off, hi);
        exit_off = r >> 24; count = r & 0xFFFFFFu;
    };
    // ---- the block's first lane, entered at every bit a codeword could start at (block 0: at the stream's first code bit, nowhere else)
    const uint32_t flat0 = a.flat ? (a.flat - (b * DL * S) % a.flat) % a.flat : 0u;      // flat code: the one bit this block can be entered at
    if (tid >= DL && tid < DL + 32 && (b == 0 ? tid == DL : (a.flat == 0 || (uint32_t)(tid - DL) == flat0))) {
        const uint32_t hi0 = min(end, blk_lo - base_bit + S);
        uint32_t e, c;
        run_in(blk_lo - base_bit, hi0, tid - DL, e, c);
        s_ex32[tid - DL] = e; s_cn32[tid - DL] = c;
        if (e < 32) atomicOr(&s_exmask, 1u << e);
    }
    uint32_t st[4] = {0, 0, 0, 0}, ex[4] = {OFF_BAD, OFF_BAD, OFF_BAD, OFF_BAD}, cn[4] = {0, 0, 0, 0}, n_ent = 0;
    auto add = [&](uint32_t off) {                                              // (n_ent < 4)
        uint32_t e, c;
        run_in(my_lo, my_hi, off, e, c);
#pragma unroll
        for (int j = 0; j < 4; j++) if ((uint32_t)j == n_ent) { st[j] = off; ex[j] = e; cn[j] = c; }
        n_ent++;
    };
    auto publish = [&] {
        s_st[tid] = st[0] | st[1] << 8 | st[2] << 16 | st[3] << 24;
        s_ex[tid] = ex[0] | ex[1] << 8 | ex[2] << 16 | ex[3] << 24;
        s_n[tid] = n_ent;
    };
    const bool chained = real && tid >= 1;                                      // lanes 1.. learn their starts from the lane before
    if (chained) add(a.flat ? (a.flat - (g * S) % a.flat) % a.flat : 0u);        // a first guess: the subsequence's first bit (flat code: the first boundary in it)
    if (tid < DL) publish();
    bool lost = false;
    for (int round = 0;; round++) {
        __syncthreads();
        uint32_t fresh[4], n_fresh = 0;
        bool over = false;
        auto offer = [&](uint32_t off) {
            if (off == OFF_BAD) return;
            bool seen = false;
#pragma unroll
            for (int k = 0; k < 4; k++) seen |= ((uint32_t)k < n_ent && st[k] == off) || ((uint32_t)k < n_fresh && fresh[k] == off);
            if (seen) return;
            if (n_ent + n_fresh >= 4) { over = true; return; }
#pragma unroll
            for (int k = 0; k < 4; k++) if ((uint32_t)k == n_fresh) fresh[k] = off;
            n_fresh++;
        };
        if (chained && tid == 1) { for (uint32_t mk = s_exmask; mk; mk &= mk - 1) offer((uint32_t)__builtin_ctz(mk)); }
        else if (chained) {
            const uint32_t pe = s_ex[tid - 1], pn = s_n[tid - 1];
#pragma unroll
            for (int j = 0; j < 4; j++) if ((uint32_t)j < pn) offer(byte_of(pe, j));
        }
        __syncthreads();
#pragma unroll 1
        for (uint32_t k = 0; k < n_fresh; k++) add(k == 0 ? fresh[0] : k == 1 ? fresh[1] : k == 2 ? fresh[2] : fresh[3]);
        if (n_fresh) publish();                                                  // (only lanes 1 .. L - 1 ever have any)
        const int flags = __syncthreads_or((n_fresh ? 1 : 0) | (over ? 2 : 0));
        if (flags & 2) { lost = true; break; }                                  // more than four phases: the general decoder
        if (!flags) break;
        if (round >= DEC_ROUNDS) { lost = true; break; }
    }
    // ---- the lanes' maps composed: lane t's entry j leads to the entry of lane t + 1 whose start is j's exit
    PathMap m; m.to = PM_ID; m.c0 = m.c1 = m.c2 = m.c3 = 0;                     // (lane 0 and lanes without a subsequence: nothing)
    if (chained) {
        m.c0 = cn[0]; m.c1 = cn[1]; m.c2 = cn[2]; m.c3 = cn[3];
        if (tid + 1 < L) {
            const uint32_t ns = s_st[tid + 1], nn = s_n[tid + 1];
            m.to = 0;
#pragma unroll
            for (int j = 0; j < 4; j++) {
                uint32_t to = 7;
#pragma unroll
                for (int k = 0; k < 4; k++) if ((uint32_t)j < n_ent && (uint32_t)k < nn && ex[j] != OFF_BAD && byte_of(ns, k) == ex[j]) to = k;
                m.to |= to << (3 * j);
            }
        } else {                                                                // the block's last lane: its entries stay as they are (their exits are read below)
            m.to = 0;
#pragma unroll
            for (int j = 0; j < 4; j++) m.to |= ((uint32_t)j < n_ent ? (uint32_t)j : 7u) << (3 * j);
        }
    }
    PathMap inc = m;                                                            // this wavefront's lanes up to and including this one
#pragma unroll 1
    for (int d = 1; d < 64; d <<= 1) { const PathMap o = pm_shfl_up(inc, d); if (lane >= (uint32_t)d) inc = pm_then(o, inc); }
    if (lane == 63) { s_wto[wave] = inc.to; s_wc[wave][0] = inc.c0; s_wc[wave][1] = inc.c1; s_wc[wave][2] = inc.c2; s_wc[wave][3] = inc.c3; }
    __syncthreads();
    if (tid == 0) {
        PathMap acc; acc.to = PM_ID; acc.c0 = acc.c1 = acc.c2 = acc.c3 = 0;
#pragma unroll 1
        for (int w = 0; w < DL / 64; w++) {
            PathMap t; t.to = s_wto[w]; t.c0 = s_wc[w][0]; t.c1 = s_wc[w][1]; t.c2 = s_wc[w][2]; t.c3 = s_wc[w][3];
            s_wto[w] = acc.to; s_wc[w][0] = acc.c0; s_wc[w][1] = acc.c1; s_wc[w][2] = acc.c2; s_wc[w][3] = acc.c3;
            acc = pm_then(acc, t);
        }
    }
    __syncthreads();
    PathMap before; before.to = s_wto[wave]; before.c0 = s_wc[wave][0]; before.c1 = s_wc[wave][1]; before.c2 = s_wc[wave][2]; before.c3 = s_wc[wave][3];
    {
        const PathMap prev = pm_shfl_up(inc, 1);                                // (the lanes before this one, within the wavefront)
        if (lane) before = pm_then(before, prev);                               // entry j of lane 1 -> (entry of THIS lane, symbols of lanes 1 .. this - 1)
    }
    if (real && tid + 1 == L && L > 1) {                                        // the block's last lane: where every entry of lane 1 leaves the block
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const uint32_t idx = pm_to(before, j);
            uint32_t e = OFF_BAD, c = 0;
#pragma unroll
            for (int k = 0; k < 4; k++) if ((uint32_t)k == idx && idx < n_ent) { e = ex[k]; c = pm_c(before, j) + cn[k]; }
            s_blk[j][0] = e; s_blk[j][1] = c;
        }
    }
    __syncthreads();
    // ---- the block's map out, the earlier blocks' maps in
    auto lane1_index = [&](uint32_t off) -> uint32_t {                          // which entry of lane 1 starts at `off`
        const uint32_t s1 = s_st[1], n1 = s_n[1];
        uint32_t j = 7;
#pragma unroll
        for (int k = 0; k < 4; k++) if ((uint32_t)k < n1 && byte_of(s1, k) == off) j = k;
        return j;
    };
    if (tid < 32) {
        uint32_t e = s_ex32[tid], c = s_cn32[tid];
        if (L > 1 && e != OFF_BAD) { const uint32_t j = lane1_index(e); if (j < 4) { c += s_blk[j][1]; e = s_blk[j][0]; } else e = OFF_BAD; }
        if (lost) e = OFF_BAD;
        a.g_maps[b * 32 + tid] = e << 24 | (c & 0xFFFFFFu);
    }
    __threadfence();
    __syncthreads();
    if (tid == 0) {
        __hip_atomic_store(&a.g_flags[b], a.seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        for (uint32_t i = 0; i < b; i++) while (__hip_atomic_load(&a.g_flags[i], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) != a.seq) __builtin_amdgcn_s_sleep(1);
    }
    __syncthreads();
    for (uint32_t i = tid; i < b * 32; i += DT) s_all[i] = __hip_atomic_load(&a.g_maps[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    if (tid == 0) {
        uint32_t c = 0, base = 0, fail = lost ? 1u : 0u;                        // (block 0 is entered at its first bit)
        for (uint32_t i = 0; i < b && !fail; i++) {
            const uint32_t e = s_all[i * 32 + c];
            if ((e >> 24) == OFF_BAD || (e >> 24) >= 32) { fail = 1; break; }
            base += e & 0xFFFFFFu; c = e >> 24;
        }
        s_true[0] = c; s_true[1] = base; s_true[2] = fail;
    }
    __syncthreads();
    const uint32_t c_in = s_true[0], out_base = s_true[1];
    bool fail = s_true[2] != 0;
    // ---- the true path through this block
    uint32_t my_start = 0, my_cnt = 0, my_at = 0, my_exit = OFF_BAD;            // (my_at: symbols of this block before this lane's)
    if (!fail) {
        const uint32_t e0 = s_ex32[c_in];
        if (tid == 0) { my_start = c_in; my_cnt = s_cn32[c_in]; my_exit = e0; }
        else if (chained) {
            const uint32_t j1 = e0 == OFF_BAD ? 7u : lane1_index(e0);
            const uint32_t idx = pm_to(before, j1);
            my_at = s_cn32[c_in] + (j1 < 4 ? pm_c(before, j1) : 0u);
#pragma unroll
            for (int k = 0; k < 4; k++) if ((uint32_t)k == idx && idx < n_ent) { my_start = st[k]; my_cnt = cn[k]; my_exit = ex[k]; }
        }
    }
    const bool is_last_lane = real && g + 1 == a.T;
    const bool broken = !fail && real && (my_exit == OFF_BAD || (is_last_lane && my_exit != 0));   // off the payload, or the stream ends inside a codeword
    const uint32_t blk_cnt_hint = (real && tid + 1 == L) ? my_at + my_cnt : 0u;
    const int bad = __syncthreads_or((broken || fail) ? 1 : 0);
    if (real && tid + 1 == L) s_true[0] = blk_cnt_hint;
    __syncthreads();
    const uint32_t blk_cnt = s_true[0];
    if (bad || blk_cnt + 16 > DEC_OUT_CAP || out_base + blk_cnt > SMALL_MAX) { block_done(&a.status[b], 1); return; }
    const uint32_t shift = out_base & 15;                                       // LDS byte i + shift <-> output byte out_base + i: 16-byte units line up
    if (real) emit_path(s_pay, s_lut, s_child, a.K, end, my_lo + my_start, my_cnt, s_out + my_at + shift);
    __syncthreads();
    {   // whole 16-byte units as they are; the first and the last are shared with the neighbours: their bytes one by one
        const uint32_t lo = shift, hi = shift + blk_cnt;                        // LDS byte range
        uint8_t *dst = a.hout + (out_base - shift);
        for (uint32_t u = tid; u * 16 < hi; u += DT) {
            const uint32_t u0 = u * 16, u1 = u0 + 16;
            if (u0 >= lo && u1 <= hi) *reinterpret_cast<uint4 *>(dst + u0) = *reinterpret_cast<const uint4 *>(s_out + u0);
            else for (uint32_t x = max(u0, lo); x < min(u1, hi); x++) dst[x] = s_out[x];
        }
    }
    if (tid == 0 && b + 1 == n_blk) a.status[DEC_BLOCKS] = out_base + blk_cnt;
    block_done(&a.status[b], 0);
}

}  // namespace

namespace {
// until none of f[0 .. n) is FLAG_PENDING; a kernel that has not answered within 5 ms is waited for the ordinary way
int wait_flags(Ctx &c, hipStream_t s, const uint32_t *f, uint32_t n) {
    const volatile uint32_t *vf = f;
    const auto t0 = std::chrono::steady_clock::now();
    for (uint32_t spins = 1;; spins++) {
        uint32_t pending = 0;
        for (uint32_t i = 0; i < n; i++) pending |= vf[i] == FLAG_PENDING;
        if (!pending) { std::atomic_thread_fence(std::memory_order_acquire); return RSN_OK; }
        if ((spins & 1023) == 0 && std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(5)) {
            RSN_HIP(hipStreamSynchronize(s));
            for (uint32_t i = 0; i < n; i++) if (vf[i] == FLAG_PENDING) return c.fail(RSN_ERR_DEVICE, "huffman: a small-input kernel finished without its answer");
            return RSN_OK;
        }
        __builtin_ia32_pause();
    }
}
}  // namespace

// 1: not an input for this path (the caller takes the general one)
int huff_small_compress(Ctx &c, const uint8_t *in, size_t n, const uint8_t **out, size_t *out_n) {
    if (n < 2 || n > SMALL_MAX) return 1;                                          // (r06: from 2 bytes up -- the README's own files are 13 and 25 bytes; one distinct symbol: the general path)
    int rc = ctx_init(c); if (rc) return rc;
    hipStream_t s = c.own_stream;
    void *pp; rc = pinned_buf(c, PIN_BYTES, &pp); if (rc) return rc;
    uint8_t *pin = (uint8_t *)pp;
    void *dp; rc = dev_buf(c, 37, SMALL_MAX + 64, &dp); if (rc) return rc;
    memcpy(pin + PIN_IN, in, n);
    memset(pin + PIN_IN + n, 0, 16);
    const uint32_t n_tiles = (uint32_t)ceil_div(n, ST);
    uint16_t *th = (uint16_t *)(pin + PIN_CNT);
    uint32_t *hi = (uint32_t *)(th + ST_MAX * 128);
    uint32_t *done = hi + ST_MAX;
    for (uint32_t t = 0; t < n_tiles; t++) { hi[t] = FLAG_PENDING; done[t] = FLAG_PENDING; }
    RSN_LAUNCH("huff_small_hist", k_small_hist, dim3(n_tiles), dim3(128), 0, s, (const uint4 *)(pin + PIN_IN), (uint32_t)n, (uint4 *)dp, th, hi);
    rc = wait_flags(c, s, hi, n_tiles); if (rc) return rc;
    uint32_t cnt[128] = {0};
    for (uint32_t t = 0; t < n_tiles; t++) {
        if (hi[t]) return 1;                                                    // a byte >= 0x80: runes (huffman.go:309)
        for (uint32_t b = 0; b < 128; b++) cnt[b] += th[t * 128 + b];
    }
    std::vector<HuffSym> syms;
    for (uint32_t b = 0; b < 128; b++) if (cnt[b]) syms.push_back({b, cnt[b]});
    if (syms.size() < 2) return 1;
    std::string hdr, msg;
    emit_header(syms, hdr);
    HuffTree tree; HuffCodes codes;
    if (!build_tree(syms, tree, msg) || !assign_codes(tree, codes, msg, false)) return 1;
    if (codes.max_len > 24) return 1;
    const unsigned pad = (unsigned)((8 - codes.total_bits % 8) % 8);            // huffman.go:245-249
    hdr.append("\\\n");
    hdr.push_back((char)pad);
    const size_t H = hdr.size();
    const size_t total = H + (size_t)((codes.total_bits + pad) / 8);
    if (H > HDR_MAX || total > SMALL_MAX) return 1;
    SmallEmitArgs a{};
    a.d_copy = (const uint8_t *)dp; a.hout = (uint32_t *)(pin + PIN_OUT); a.done = done;
    a.n = (uint32_t)n; a.H = (uint32_t)H; a.total = (uint32_t)total;
    uint32_t len_of[128] = {0};
    for (uint32_t i = 0; i < tree.n_leaves; i++) { a.tab[tree.rune[i]] = ((uint32_t)codes.len[i] << 24) | (uint32_t)codes.code[i]; len_of[tree.rune[i]] = codes.len[i]; }
    a.P[0] = (uint32_t)(8 * H + pad);
    for (uint32_t t = 0; t < n_tiles; t++) {                                    // where every tile's first code bit goes
        uint32_t bits = 0;
        for (uint32_t b = 0; b < 128; b++) bits += th[t * 128 + b] * len_of[b];
        a.P[t + 1] = a.P[t] + bits;
    }
    memcpy(a.hdr, hdr.data(), H);
    RSN_LAUNCH("huff_small_emit", k_small_emit, dim3(n_tiles), dim3(256), 0, s, a);
    rc = wait_flags(c, s, done, n_tiles); if (rc) return rc;
    *out = pin + PIN_OUT; *out_n = total;
    return RSN_OK;
}

int huff_small_decompress(Ctx &c, const uint8_t *in, size_t n, const uint8_t **out, size_t *out_n) {
    if (n < 8 || n > DEC_STREAM_MAX) return 1;
    // ---- strings.SplitN(content, "\\\n", 2) (huffman.go:261); the counts (huffman.go:196-227)
    size_t sep = (size_t)-1;
    for (size_t i = 0; i + 1 < std::min<size_t>(n, HDR_MAX + 8); i++) if (in[i] == 0x5C && in[i + 1] == 0x0A) { sep = i; break; }
    if (sep == (size_t)-1 || sep + 4 > n) return 1;
    std::vector<HuffSym> syms; std::string msg;
    if (!parse_header(in, sep, syms, msg) || syms.size() < 2 || syms.size() > 128) return 1;
    unsigned long long expect = 0;
    for (const HuffSym &sy : syms) { if (sy.rune >= 0x80 || sy.freq > SMALL_MAX) return 1; expect += sy.freq; }
    if (expect == 0 || expect > SMALL_MAX) return 1;
    const size_t pay = sep + 3, sn = n - sep - 2;
    const unsigned diff = in[sep + 2];
    const unsigned long long nbits = (unsigned long long)(sn - 1) * 8;
    if (diff >= nbits) return 1;
    HuffTree tree; HuffCodes codes;
    if (!build_tree(syms, tree, msg) || !assign_codes(tree, codes, msg, false)) return 1;
    if (codes.max_len > 32 || codes.max_len == 0) return 1;
    int rc = ctx_init(c); if (rc) return rc;
    hipStream_t s = c.own_stream;
    void *pp; rc = pinned_buf(c, PIN_BYTES, &pp); if (rc) return rc;
    uint8_t *pin = (uint8_t *)pp;
    void *dp; rc = dev_buf(c, 38, (DEC_BLOCKS * 32 + DEC_BLOCKS) * 4, &dp); if (rc) return rc;      // the blocks' maps and flags
    // A flag holds the number of the call that set it: nothing to clear between calls on the same ALLOCATION.  Keyed on the allocation's
    // process-unique number, not its address (rsn_device_set / rsn_trim / the admission gate free the slot, and a later
    // hipMalloc can hand the same address back with other contents -- a stale flag equal to the current number would pass for this call's).
    static thread_local uint32_t seq = 0;
    static thread_local unsigned long long seq_of = 0;
    if (seq_of != c.bufs[38].gen || ++seq == 0) { seq = 1; seq_of = c.bufs[38].gen; RSN_HIP(hipMemsetAsync(dp, 0, (DEC_BLOCKS * 32 + DEC_BLOCKS) * 4, s)); }
    // ---- the tree as the kernel wants it (every block fills its lookup table from it)
    const int K = (int)std::min<unsigned>(codes.max_len, DEC_K);
    SmallDecArgs a{};
    const uint32_t A = tree.n_leaves;
    const size_t n_int = tree.freq.size() - A;
    for (size_t i = 0; i < n_int; i++) {
        const int32_t kids[2] = {tree.left[A + i], tree.right[A + i]};
        for (int b = 0; b < 2; b++) a.child[2 * i + b] = tree.is_leaf(kids[b]) ? (uint16_t)(0x8000u | tree.rune[kids[b]]) : (uint16_t)(kids[b] - (int32_t)A);
    }
    a.root = (uint32_t)(tree.root - (int32_t)A);
    const size_t A0 = pay & ~(size_t)3;
    memcpy(pin + PIN_IN, in + A0, n - A0);
    memset(pin + PIN_IN + (n - A0), 0, 64);
    uint32_t *status = (uint32_t *)(pin + PIN_CNT);
    a.pay = (const uint32_t *)(pin + PIN_IN);
    a.pay_words = (uint32_t)((n - A0 + 3) / 4) + 8;
    a.p0 = (uint32_t)(8 * (pay - A0) + diff);
    a.end = (uint32_t)(8 * (pay - A0) + nbits);
    a.K = (uint32_t)K; a.n_child = (uint32_t)(2 * n_int);
    const uint32_t span = a.end - a.p0;
    a.S = std::max<uint32_t>(64, (uint32_t)round_up(ceil_div(span, DEC_BLOCKS * DL), 32));
    if (a.S > DEC_S_MAX) return 1;
    a.T = (uint32_t)ceil_div(span, a.S);
    const uint32_t n_blk = (uint32_t)ceil_div(a.T, DL);
    a.seq = seq;
    a.flat = codes.min_len == codes.max_len ? codes.max_len : 0u;
    a.hout = pin + PIN_OUT; a.status = status;
    a.g_maps = (uint32_t *)dp; a.g_flags = a.g_maps + DEC_BLOCKS * 32;
    for (uint32_t b = 0; b <= DEC_BLOCKS; b++) status[b] = FLAG_PENDING;
    RSN_LAUNCH("huff_small_dec", k_small_dec, dim3(n_blk), dim3(DT), 0, s, a);
    rc = wait_flags(c, s, status, n_blk); if (rc) return rc;
    for (uint32_t b = 0; b < n_blk; b++) if (status[b] != 0) return 1;
    *out = pin + PIN_OUT; *out_n = status[DEC_BLOCKS];
    return RSN_OK;
}

}  // namespace rsn
