// lzss_encode.hip -- LZSS encode for gfx950 (MI355X).
//
// Replaces lz.CompressAsync (compressor/lz/lzss.go:109): EncodeOpeningSymbols :369,
// the per-position compressorWorker fan-out :119-130,:166-184 and the greedy
// compaction :134-151.
//
//   E1 k_esc_count / k_esc_write   '<'->FF, FF->5C FF, 5C->5C 5C (lzss.go:369-389)
//   E2 match search: for EVERY escaped position i the longest L such that fc[i:i+L] occurs
//                  entirely inside the window fc[max(0,i-W):i], and the distance to its
//                  LEFTMOST occurrence (bytes.Index, lzss.go:419).  key[i] = L<<16 | distance
//                  (0 => literal).
//        k_match_hash  common case (W <= 4096): bigram buckets in LDS, work proportional to how
//                      often a bigram recurs in the window; exact, hands pathological strips back
//        k_match2      general case / hand-backs: diagonal sweep, O(1) per (position, distance)
//                      pair whatever the data (two diagonals per lane, lengths only: the distance is recovered where it matters)
//        k_match_chain the fast path (r03-): the keys only where greedy chains land, eight chains a wavefront; its second instance
//                      (ChainCfg::RUNS, r06) resolves a position inside a stretch of a short period -- a run of a byte, a line
//                      repeated -- from the window's stretches (chain_period_visit); period_sample_block picks the instance
//   E3 k_parse_exit / k_parse_super / k_parse_chain / k_parse_fill / k_parse_mark
//                  the greedy chain of lzss.go:136-151 (position i is visited iff no earlier
//                  visited reference covers it), without a serial walk over the stream
//   E4 k_tok_emit  "<off,len>" if shorter than the match, else the raw bytes (lzss.go:143-149)
//
// E2 is the cost.  The chain steps over every match, encodable or not, so the exact L is needed
// wherever the chain can land: no cut-offs.  Diagonal sweep: lane = position,
// step = diagonal.  On diagonal d (candidate start i-d) the match length obeys
//     run_d(i) = fc[i]==fc[i-d] ? run_d(i+1)+1 : 0,
// so a wavefront sweeps 64 consecutive positions against one candidate byte per step (uniform
// LDS read), the run counters move one lane down per step (DPP wave_shl:1) and every (i,d)
// pair costs O(1) regardless of the data.
#include "codecs.h"
#include "lzss_match.h"
#include <chrono>

namespace rsn {

int lzss_encode_big(Ctx &c, hipStream_t s, const uint8_t *d_fc, uint32_t E, uint32_t W, uint8_t *d_out, size_t out_cap, size_t *out_n);

constexpr int LB = 256;                 // threads per block
constexpr int ESC_TILE = LB * 16;       // input bytes per escape block
constexpr int PT = 8192;                // positions per parse tile
constexpr uint32_t MAX_WINDOW = 8192;
constexpr uint32_t NO_ENTRY = 0xFFFFFFFFu;
constexpr uint32_t PERIODIC_TAIL_MIN_TILES = 16;   // a W-periodic rest of at least this many tiles is placed by arithmetic (lzss_encode_admitted)

// ------------------------------------------------------------------ E1: escape
__device__ __forceinline__ uint32_t count_special16(const uint32_t w[4]) {
    uint32_t c = 0;
#pragma unroll
    for (int j = 0; j < 4; j++)
#pragma unroll
        for (int b = 0; b < 4; b++) { const uint32_t v = (w[j] >> (8 * b)) & 0xFF; c += (v == 0x5C || v == 0xFF); }
    return c;
}

// 0xFF in every byte of w that equals the byte `c` (exact per byte: no borrow crosses a byte boundary)
__device__ __forceinline__ uint32_t bytes_equal(uint32_t w, uint32_t c) {
    const uint32_t t = w ^ (c * 0x01010101u);
    const uint32_t z = ~(((t & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | t | 0x7F7F7F7Fu);   // 0x80 where the byte of t is zero
    return (z >> 7) * 0xFFu;
}

__device__ __forceinline__ void load16(const uint8_t *in, size_t n, size_t P, uint32_t w[4], int *cnt) {
    w[0] = w[1] = w[2] = w[3] = 0;
    if (P + 16 <= n) {
        const uint4 v = *reinterpret_cast<const uint4 *>(in + P);
        w[0] = v.x; w[1] = v.y; w[2] = v.z; w[3] = v.w; *cnt = 16;
    } else if (P < n) {
        *cnt = (int)(n - P);
        for (int k = 0; k < *cnt; k++) w[k >> 2] |= (uint32_t)in[P + k] << (8 * (k & 3));
    } else *cnt = 0;
}

__global__ __launch_bounds__(LB) void k_esc_count(const uint8_t *__restrict__ in, size_t n, unsigned long long *__restrict__ blk_extra) {
    __shared__ uint32_t part[LB / 64];
    const size_t P = (size_t)blockIdx.x * ESC_TILE + threadIdx.x * 16;
    uint32_t w[4]; int cnt;
    load16(in, n, P, w, &cnt);
    uint32_t c = 0;
    if (cnt == 16) c = count_special16(w);
    else for (int k = 0; k < cnt; k++) { const uint32_t v = (w[k >> 2] >> (8 * (k & 3))) & 0xFF; c += (v == 0x5C || v == 0xFF); }
    for (int d = 32; d; d >>= 1) c += __shfl_down(c, d);
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = c;
    __syncthreads();
    if (threadIdx.x == 0) blk_extra[blockIdx.x] = part[0] + part[1] + part[2] + part[3];
}

// A block's output is contiguous: bytes go to an LDS image shifted so that its 16-byte units line
// up with global memory, then out with 16-byte stores (per-lane byte stores cost 3x the time).
// the same, leaving the image zeroed behind it (for writers that OR their bytes in)
__device__ __forceinline__ void drain_block_zero(uint8_t *s_img, uint32_t al, uint32_t total, uint8_t *dst) {
    const uint32_t span = al + total;
    uint8_t *gbase = dst - al;
    for (uint32_t u = threadIdx.x; u * 16 < span + 16; u += blockDim.x) {   // (one unit past the span: an entry's four words may reach there, as zeros)
        const uint32_t b0 = u * 16;
        if (b0 >= al && b0 + 16 <= span) *reinterpret_cast<uint4 *>(gbase + b0) = *reinterpret_cast<const uint4 *>(s_img + b0);
        else for (uint32_t k = max(b0, al); k < min(b0 + 16, span); k++) gbase[k] = s_img[k];
        *reinterpret_cast<uint4 *>(s_img + b0) = make_uint4(0u, 0u, 0u, 0u);
    }
}
__device__ __forceinline__ void drain_block(const uint8_t *s_img, uint32_t al, uint32_t total, uint8_t *dst) {
    // s_img[al .. al+total) is the output, dst - al is 16-byte aligned
    const uint32_t span = al + total;
    uint8_t *gbase = dst - al;
    for (uint32_t u = threadIdx.x; u * 16 < span; u += blockDim.x) {
        const uint32_t b0 = u * 16;
        if (b0 >= al && b0 + 16 <= span) *reinterpret_cast<uint4 *>(gbase + b0) = *reinterpret_cast<const uint4 *>(s_img + b0);
        else for (uint32_t k = max(b0, al); k < min(b0 + 16, span); k++) gbase[k] = s_img[k];
    }
}

__global__ __launch_bounds__(LB) void k_esc_write(const uint8_t *__restrict__ in, size_t n, const unsigned long long *__restrict__ blk_off,
                                                  uint8_t *__restrict__ fc) {
    __shared__ uint32_t wsum[LB / 64];
    __shared__ __attribute__((aligned(16))) uint8_t s_img[2 * ESC_TILE + 32];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const size_t P = (size_t)blockIdx.x * ESC_TILE + tid * 16;
    uint32_t w[4]; int cnt;
    load16(in, n, P, w, &cnt);
    uint32_t c = 0;
    if (cnt == 16) c = count_special16(w);
    else for (int k = 0; k < cnt; k++) { const uint32_t v = (w[k >> 2] >> (8 * (k & 3))) & 0xFF; c += (v == 0x5C || v == 0xFF); }
    uint32_t incl = c;
    for (int d = 1; d < 64; d <<= 1) { const uint32_t y = __shfl_up(incl, d); if (lane >= d) incl += y; }
    if (lane == 63) wsum[wv] = incl;
    __syncthreads();
    uint32_t pre = 0, extra = 0;
    for (int k = 0; k < LB / 64; k++) { if (k < wv) pre += wsum[k]; extra += wsum[k]; }
    uint8_t *dst = fc + (size_t)blockIdx.x * ESC_TILE + blk_off[blockIdx.x];
    const uint32_t al = (uint32_t)((uintptr_t)dst & 15);
    const uint32_t total = (uint32_t)min((size_t)ESC_TILE, n - (size_t)blockIdx.x * ESC_TILE) + extra;
    if (extra == 0 && (al & 3) == 0) {
        // no byte of the block needs an escape (most blocks of most files): every byte keeps its place, '<' becomes FF
        // (lzss.go:373-377); a 16-byte store only needs dword alignment
#pragma unroll
        for (int j = 0; j < 4; j++) w[j] |= bytes_equal(w[j], 0x3Cu);
        uint8_t *d = dst + tid * 16;
        if (cnt == 16) *reinterpret_cast<uint4 *>(d) = make_uint4(w[0], w[1], w[2], w[3]);
        else for (int k = 0; k < cnt; k++) d[k] = (uint8_t)(w[k >> 2] >> (8 * (k & 3)));
        return;
    }
    uint8_t *o = s_img + al + tid * 16 + pre + incl - c;
#pragma unroll
    for (int k = 0; k < 16; k++) {
        if (k < cnt) {
            uint32_t v = (w[k >> 2] >> (8 * (k & 3))) & 0xFF;
            if (v == 0x3C) v = 0xFF;                       // '<' -> EncodedOpening (lzss.go:373-377)
            else if (v == 0xFF || v == 0x5C) *o++ = 0x5C;  // escape byte first (lzss.go:378-379)
            *o++ = (uint8_t)v;
        }
    }
    __syncthreads();
    drain_block(s_img, al, total, dst);
}

// The optimistic form: most inputs hold no byte that needs an escape (5C, FF) at all, and then the escaped stream is the input with
// '<' mapped to FF (lzss.go:373-377), every byte in its place.  One pass writes exactly that and raises a flag if it met a 5C or FF;
// only then do the counting pass, the scan and k_esc_write run (the flagged attempt cost one copy).
// (r04) The same pass also answers k_tile_periodic's question for the common window -- does every byte equal the byte Wp before it? -- from
// the bytes it has in registers and a second load Wp earlier that the neighbouring block has just brought into the L2: one flag per block;
// k_tiles_from_blocks turns them into the tiles' records.  (Config 3: k_tile_periodic read tile + window again from memory, 0.60 ms per
// GiB; comparing the INPUT is the same as comparing the escaped stream while nothing needs an escape: '<' -> FF is injective on such input.)
constexpr int ESC_RUN = 8;                  // consecutive 4 KiB chunks per block of k_esc_try: with the engine's window (4096 = one chunk) the bytes "Wp before" are the
                                            // previous chunk's, still in the lane's registers -- 1.125 N fetched for the comparison instead of 2 N
__global__ __launch_bounds__(LB) void k_esc_try(const uint8_t *__restrict__ in, size_t n, uint8_t *__restrict__ fc, unsigned long long *__restrict__ flag,
                                                uint32_t Wp, uint8_t *__restrict__ same_blk, uint32_t n_chunks) {
    if (!fc && same_blk && Wp == (uint32_t)ESC_TILE && blockIdx.x > 0 && (size_t)(blockIdx.x + 1) * ESC_RUN * ESC_TILE <= n) {
        // The check alone, on whole chunks inside the stream (r05): the run's nine loads -- its eight chunks and the one before -- in flight
        // together, the chunks' verdicts reduced once (the loop below waits for a load and a barrier per chunk: 3.6 TB/s; this: the read rate).
        const uint4 *p = reinterpret_cast<const uint4 *>(in + (size_t)blockIdx.x * ESC_RUN * ESC_TILE + threadIdx.x * 16);
        uint4 v[ESC_RUN + 1];
#pragma unroll
        for (int k = 0; k <= ESC_RUN; k++) v[k] = p[(k - 1) * (ESC_TILE / 16)];
        uint32_t same_bits = 0, special = 0, lt = 0;
        // (r06: "is there a byte equal to X" as an ACCUMULATED zero-byte test -- (t - 0x01010101) & ~t has bit 7 of a byte set iff that byte
        //  of t = w ^ XXXX is zero or lies above one that is: exact for existence, four instructions a value and dword instead of seven)
        auto zacc = [](uint32_t w, uint32_t x4) { const uint32_t t = w ^ x4; return (t - 0x01010101u) & ~t; };
#pragma unroll
        for (int k = 1; k <= ESC_RUN; k++) {
            const uint32_t d = (v[k].x ^ v[k - 1].x) | (v[k].y ^ v[k - 1].y) | (v[k].z ^ v[k - 1].z) | (v[k].w ^ v[k - 1].w);
            same_bits |= (d == 0 ? 1u : 0u) << (k - 1);
            const uint32_t w[4] = {v[k].x, v[k].y, v[k].z, v[k].w};
#pragma unroll
            for (int j = 0; j < 4; j++) { special |= zacc(w[j], 0x5C5C5C5Cu) | zacc(w[j], 0xFFFFFFFFu); lt |= zacc(w[j], 0x3C3C3C3Cu); }
        }
        special &= 0x80808080u; lt &= 0x80808080u;
        __shared__ uint32_t s_and[LB / 64], s_any[LB / 64];
        uint32_t wave_and = 0;
#pragma unroll
        for (int k = 0; k < ESC_RUN; k++) wave_and |= (__ballot((same_bits >> k) & 1u) == ~0ull ? 1u : 0u) << k;
        const uint32_t wave_any = (__ballot(special != 0) ? 1u : 0u) | (__ballot(lt != 0) ? 2u : 0u);   // (bit 2, "holds runs of a byte or stretches of a short period", is the sample's: period_sample_block -- measured here as five more instructions per 16 bytes it cost config 3's check 0.24 -> 0.26 ms)
        if ((threadIdx.x & 63) == 0) { s_and[threadIdx.x >> 6] = wave_and; s_any[threadIdx.x >> 6] = wave_any; }
        __syncthreads();
        uint32_t all = ~0u, any = 0;
#pragma unroll
        for (int wv = 0; wv < LB / 64; wv++) { all &= s_and[wv]; any |= s_any[wv]; }
        if (threadIdx.x < (uint32_t)ESC_RUN) same_blk[blockIdx.x * ESC_RUN + threadIdx.x] = (uint8_t)((all >> threadIdx.x) & 1u);
        if (threadIdx.x == 0 && (any & ~__atomic_load_n(flag, __ATOMIC_RELAXED))) atomicOr(flag, (unsigned long long)any);
        return;
    }
    uint32_t prev[4] = {0, 0, 0, 0};
    int prev_cnt = -1;                                                     // -1: nothing in `prev` (the run's first chunk, or Wp is not a chunk)
    uint32_t seen = 0;                                                     // (bit 2 of the flag word, r06: the stream holds runs of a byte or stretches of a short period -- k_match_chain<RUNS> is its walk; set by period_sample_block)
                                                                           // bit 0: a 5C / FF, bit 1: a '<' -- one look at the flag per BLOCK, at the end (r05: a stream with a '<' in
                                                                           // every wavefront's 1 KiB -- config 3 -- had sixteen million wavefronts read the one flag word: 0.44 ms against 0.25)
    for (uint32_t k = 0; k < (uint32_t)ESC_RUN; k++) {
        const uint32_t chunk = blockIdx.x * ESC_RUN + k;
        if (chunk >= n_chunks) break;                                      // (block-uniform)
        const size_t P = (size_t)chunk * ESC_TILE + threadIdx.x * 16;
        uint32_t w[4]; int cnt;
        load16(in, n, P, w, &cnt);                                         // (bytes beyond n read as zero: neither special nor '<')
        bool same = true;
        if (Wp) {                                                          // (a multiple of 16: the earlier bytes are one aligned load too)
            if (P < Wp) same = cnt == 0;
            else if (cnt) {
                uint32_t v[4]; int c2 = 16;
                if (Wp == (uint32_t)ESC_TILE && prev_cnt >= 0) { v[0] = prev[0]; v[1] = prev[1]; v[2] = prev[2]; v[3] = prev[3]; }
                else load16(in, n, P - Wp, v, &c2);
                if (cnt == 16) same = ((w[0] ^ v[0]) | (w[1] ^ v[1]) | (w[2] ^ v[2]) | (w[3] ^ v[3])) == 0;
                else for (int q = 0; q < cnt; q++) same = same && ((w[q >> 2] ^ v[q >> 2]) >> (8 * (q & 3)) & 0xFF) == 0;
            }
            prev[0] = w[0]; prev[1] = w[1]; prev[2] = w[2]; prev[3] = w[3]; prev_cnt = cnt;   // (the INPUT bytes: before '<' becomes FF below)
        }
        uint32_t special = 0, lt = 0;
#pragma unroll
        for (int j = 0; j < 4; j++) { special |= bytes_equal(w[j], 0x5Cu) | bytes_equal(w[j], 0xFFu); const uint32_t m = bytes_equal(w[j], 0x3Cu); lt |= m; w[j] |= m; }
        seen |= (special != 0 ? 1u : 0u) | (lt != 0 ? 2u : 0u);
        if (fc) {                                                          // (null: the check alone -- r05: the stream is then the input itself, '<' mapped where it is loaded)
            uint8_t *d = fc + P;
            if (cnt == 16) *reinterpret_cast<uint4 *>(d) = make_uint4(w[0], w[1], w[2], w[3]);
            else for (int q = 0; q < cnt; q++) d[q] = (uint8_t)(w[q >> 2] >> (8 * (q & 3)));
        }
        if (Wp) {
            const int all = __syncthreads_and(same);
            if (threadIdx.x == 0) same_blk[chunk] = (uint8_t)all;
        }
    }
    const int any1 = __syncthreads_or(seen & 1u), any2 = __syncthreads_or(seen & 2u);
    if (threadIdx.x == 0) {
        const unsigned long long want = (any1 ? 1ull : 0ull) | (any2 ? 2ull : 0ull);
        if (want & ~__atomic_load_n(flag, __ATOMIC_RELAXED)) atomicOr(flag, want);
    }
}

// (r06) Does the input hold runs of a byte or stretches of a short period -- a zero-filled buffer, a line or a record repeated?  Then the
// walk is k_match_chain<RUNS>'s.  A walk by the lean instance that ends in "heavy" tiles is done again by the other, but what it walked
// before giving up is lost (sparse CSV, 16 MiB: 3 of 6 ms).  A sample instead, into bit 2 of the check's flag word: up to 64 chunks of
// 4 KiB spread over the input, every eighth position asks whether its 24 bytes come again within 64; a chunk where one in fifty does
// raises the flag.  Text: none.  (Eight bytes alike at a 16-byte boundary, tested by k_esc_try on all of the input, was the first form:
// five instructions per 16 bytes that cost config 3's check 0.24 -> 0.26 ms.)
constexpr uint32_t PSAMPLE_CHUNK = 4096, PSAMPLE_MAX_P = 64;
// (the block's threads all call it; which: the sample's index, < n_samples.  As a function: a large input's sample rides on k_last_unlike's
//  launch -- config 3's 0.54 ms call has no 0.02 ms for a launch of its own)
__device__ __forceinline__ void period_sample_block(const uint8_t *__restrict__ in, size_t n, uint32_t n_samples, uint32_t which, unsigned long long *__restrict__ flag) {
    __shared__ __attribute__((aligned(16))) uint32_t s_w[(PSAMPLE_CHUNK + PSAMPLE_MAX_P + 32 + 16) / 4];
    __shared__ uint32_t s_hits;
    const size_t span = PSAMPLE_CHUNK + PSAMPLE_MAX_P + 32;
    if (n < span || which >= n_samples) return;
    const size_t at = n_samples > 1 ? (size_t)((unsigned __int128)(n - span) * which / (n_samples - 1)) & ~(size_t)15 : 0;
    for (uint32_t v = threadIdx.x; v < span / 16; v += 256) reinterpret_cast<uint4 *>(s_w)[v] = *reinterpret_cast<const uint4 *>(in + at + 16 * (size_t)v);
    if (threadIdx.x == 0) s_hits = 0;
    __syncthreads();
    uint32_t hits = 0;
    for (uint32_t i = 8u * threadIdx.x; i < PSAMPLE_CHUNK; i += 8u * 256u) {
        const unsigned long long a0 = lds_load8(s_w, i), a1 = lds_load8(s_w, i + 8), a2 = lds_load8(s_w, i + 16);
        bool hit = false;
        for (uint32_t P0 = 1; P0 <= PSAMPLE_MAX_P && !hit; P0 += 8) {     // (eight distances' first eight bytes at once: one at a time the loop is a chain of LDS round trips, 12 us of config 3's 540)
            uint32_t m = 0;
#pragma unroll
            for (uint32_t j = 0; j < 8; j++) m |= (lds_load8(s_w, i + P0 + j) == a0 ? 1u : 0u) << j;
            for (; m && !hit; m &= m - 1) { const uint32_t P = P0 + (uint32_t)__builtin_ctz(m); hit = lds_load8(s_w, i + P + 8) == a1 && lds_load8(s_w, i + P + 16) == a2; }
        }
        hits += hit ? 1u : 0u;
    }
    if (hits) atomicAdd(&s_hits, hits);
    __syncthreads();
    if (threadIdx.x == 0 && s_hits * 50u >= PSAMPLE_CHUNK / 8u && !(__atomic_load_n(flag, __ATOMIC_RELAXED) & 4ull)) atomicOr(flag, 4ull);
}
__global__ __launch_bounds__(256) void k_period_sample(const uint8_t *__restrict__ in, size_t n, uint32_t n_samples, unsigned long long *__restrict__ flag) {
    period_sample_block(in, n, n_samples, blockIdx.x, flag);
}

// k_tile_periodic's records from k_esc_try's block flags: a tile is W-periodic iff every block that overlaps [t0, t0 + tile + W - 1) repeats
// the bytes W before it (block granularity: a stretch that ends inside the last block is walked like any other tile)
constexpr uint32_t PREV_BLK = 8192;                                  // tiles per block of k_prev_walked
struct TileChain { uint32_t entry, exit, walked, pad; };   // where the block's chain enters its tile / first lands beyond it (stream positions)
// k_match_chain's arguments.  What only the prologue and the epilogue need sits in `tail` and is read from the
// kernel-argument segment THERE instead of staying live across the chain walk.  The kernel must stay at or below
// 80 SGPRs (it has 58): above that a CU holds one of its 16-wavefront blocks instead of two, and the walk halves
// in speed -- which is also why the in-tile parse is a kernel of its own (k_chain_tail) and not this one's epilogue.
struct ChainTail { uint32_t *heavy, *dense; TileChain *tchain; uint8_t *dump; const uint32_t *redo_list; uint32_t *n_dense; uint32_t *step; const uint32_t *redo_start; uint32_t *ckeys, *ckn; };   // ckeys / ckn: the keys of a tile's claimed positions, in position order, and how many (k_chain_serial's input; may be null)   // redo_list: the tiles of a partial launch (ChainArgs::redo & 2); n_dense: counts the tiles that gave up as dense (may be null)
struct ChainArgs { const uint8_t *fc; uint32_t E; uint32_t W; uint32_t *keys; uint32_t redo; uint32_t pfrom; unsigned long long *stats; ChainTail tail; };   // pfrom: every position from here on repeats the W bytes before it as far as a match may reach (the head of a W-periodic input, lzss_encode_admitted): its key is arithmetic; 0xFFFFFFFF: none   // redo: bit 0 = no density test (second look), bit 1 = tiles from tail.redo_list
// k_match_chain's record of a claimed position: .x its key, .y = next record (14 bits) | position from t0 - CH (14 bits) << 14
// (what the position puts into the output -- a token's text or the bytes themselves -- is worked out by the reader: here it would cost
//  the walk's kernel what it saves the reader)
// SECTIONS (r04).  A stream too long for one pass (positions are 32 bits) is encoded section by section.  A section's stream begins
// HALO_TILES tiles before the position the true chain enters it -- the exit of the section before -- so that every position of the
// section sees its whole window; those tiles are never walked (ChainArgs::redo bit 2), their records say "the chain steps from tile
// to tile" (k_halo_init), the first tile behind them starts its chain AT its first position instead of a warm-up zone before it,
// and the general parse starts there too (k_parse_chain).  HALO_TILES is a group of the general parse: it enters a group through
// its first tile.
constexpr uint32_t HALO_TILES = 64;
constexpr uint32_t CK_EXIT = 0x3FFEu, CK_BROKEN = 0x3FFFu;           // "next record" when the chain leaves the tile / lands on a position nobody evaluated
constexpr uint32_t NO_LIST = 0xFFFFFFFFu;                            // ccnt[tile]: the tile has no compact key list (k_tok_emit reads its flags and keys); ckn[tile]: nor the claimed positions' keys
__device__ __forceinline__ ChainTail chain_tail() {
    return *(const ChainTail *)((const char *)__builtin_amdgcn_kernarg_segment_ptr() + offsetof(ChainArgs, tail));
}

__device__ __forceinline__ uint32_t enc_len(uint32_t off, uint32_t len);


// ------------------------------------------------------------------ E2''': the bucket search, evaluated only where a greedy chain lands
// The output depends on key[i] only at the positions the greedy chain from 0 visits (lzss.go:136-151:
// a reference skips size-1 positions): about one position in five on text, and the cheap ones (after
// a match the chain tends to land where few candidates share the bigram).  Which positions those are
// is only known serially, but greedy chains merge: two chains that ever visit the same position
// coincide from there on, and from an arbitrary start a chain meets the true one after a match or
// two (measured on the Zipf text: merge distance p50 7, p99 39 positions).  So a block walks chains:
//   * one WAVEFRONT per chain.  The 64 lanes examine 64 candidates of the chain's current position at
//     once (entry, window and tag tests, 8-byte compares against the position's bytes, which every
//     lane reads from the same LDS address), a DPP/readlane maximum gives the exact key
//     L<<16 | distance, lane 0 stores it, claims the landing position i + max(1, L) in an LDS bitmap
//     and the wavefront goes on from there -- all control flow is wave-uniform;
//   * chains start every CS positions of the tile (handed out from a shared counter) and stop when
//     they land on a position somebody has claimed: that owner walks the rest;
//   * one more chain starts CH positions BEFORE the tile: where the true chain enters the tile
//     depends on every earlier tile, so the block walks a warm-up chain instead, which has merged
//     with the true one long before it reaches the tile.
// Every key written is exact; what is speculative is only WHICH positions get one.  All others keep
// KEY_UNKNOWN; if the true chain ever lands on one (k_parse_mark notices), that strip is redone for
// all positions by k_match_hash and the parse is repeated -- correctness never rests on the merge,
// only the speed does.
template <int CT_, int CTH_, int CS_, bool RUNS_ = false>
struct ChainCfg {
    static constexpr int CT = CT_, CTH = CTH_, CS = CS_;
    static constexpr bool RUNS = RUNS_;         // visits inside a stretch of a short period -- a run of a byte, a line repeated -- are resolved from the window's stretches (chain_period_visit): the kernel for streams that hold such
    static constexpr int CSH = 9;               // a bucket's entries are ordered by staged offset >> CSH
    static constexpr int CH = 128;                          // warm-up positions before the tile
    static constexpr int NS = HWMAX + CH + CT;              // staged positions that can be candidates
    static constexpr int OFFB = NS <= 8192 ? 13 : NS <= 16384 ? 14 : 15;   // bits of a staged offset in a list entry
    static constexpr int TAGB = 16 - OFFB;                  // the rest carries bits 5.. of the second byte
    static constexpr int OFF0 = TAGB >= 3 ? 2 : 1;          // bytes known equal when bucket and tag agree
    static constexpr int STAGE = NS + HLMAX + 32;
    static constexpr int NBLK = (NS + (1 << CSH) - 1) >> CSH;
    static constexpr uint32_t ROUND_CAP = 1u << 16;         // candidate rounds per wavefront before the strip is handed back
    static constexpr int DUMP_BYTES = ((CH + CT) / 8 + 15) / 16 * 16;   // k_match_chain's record of a tile for k_chain_tail: the claim bitmap
    static_assert(CTH % (1 << CSH) == 0 && (HNB / 2) % CTH == 0 && NS <= 32768 && NBLK <= 64, "round structure");
    static_assert(MATCH_STRIP % CT == 0 && CT % 32 == 0 && CH % 32 == 0 && STAGE % 16 == 0, "tile structure");
};

#define RSN_DPP_QUAD_XOR1 0xB1
#define RSN_DPP_QUAD_XOR2 0x4E
#define RSN_DPP_ROW_HALF_MIRROR 0x141
#define RSN_DPP_ROW_MIRROR 0x140
#define RSN_DPP_ROW_BCAST15 0x142
#define RSN_DPP_ROW_BCAST31 0x143
__device__ __forceinline__ uint32_t wave_max_u32(uint32_t v) {       // the maximum over the 64 lanes, wave-uniform
    v = max(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, RSN_DPP_QUAD_XOR1, 0xF, 0xF, true));
    v = max(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, RSN_DPP_QUAD_XOR2, 0xF, 0xF, true));
    v = max(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, RSN_DPP_ROW_HALF_MIRROR, 0xF, 0xF, true));
    v = max(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, RSN_DPP_ROW_MIRROR, 0xF, 0xF, true));   // every lane: maximum of its row of 16
    v = max(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, RSN_DPP_ROW_BCAST15, 0xA, 0xF, false)); // rows 1 and 3 take in the row before
    v = max(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, RSN_DPP_ROW_BCAST31, 0xC, 0xF, false)); // rows 2 and 3 take in lane 31: lane 63 has it all
    return (uint32_t)__builtin_amdgcn_readlane((int)v, 63);
}
// a barrier that orders the block's LDS traffic only: the loads from memory a thread has in flight stay in flight across it
__device__ __forceinline__ void lds_barrier() {
    __builtin_amdgcn_s_waitcnt(0xC07F);                                   // lgkmcnt(0); vmcnt and expcnt untouched
    __builtin_amdgcn_s_barrier();
}
__device__ __forceinline__ uint32_t uni(uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); }
// the same value, but in a vector register and opaque to the compiler's uniformity analysis: what is computed from it is
// computed by the vector units (this kernel is bound by the CU's one scalar unit)
__device__ __forceinline__ uint32_t vec(uint32_t x) { asm volatile("" : "+v"(x)); return x; }
// The same reductions over a ROW of LW = 16 or 32 lanes (k_match_chain with several chains per wavefront): every lane of the row gets the result.
template <int LW>
__device__ __forceinline__ uint32_t row_max_u32(uint32_t v) {
    static_assert(LW == 4 || LW == 8 || LW == 16 || LW == 32, "a quad, half a DPP row, a row, or two");
    v = max(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, RSN_DPP_QUAD_XOR1, 0xF, 0xF, true));
    v = max(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, RSN_DPP_QUAD_XOR2, 0xF, 0xF, true));
    if constexpr (LW >= 8) v = max(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, RSN_DPP_ROW_HALF_MIRROR, 0xF, 0xF, true));
    if constexpr (LW >= 16) v = max(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, RSN_DPP_ROW_MIRROR, 0xF, 0xF, true));
    if constexpr (LW == 32) v = max(v, (uint32_t)__builtin_amdgcn_ds_swizzle((int)v, 0x401F));   // lane ^ 16
    return v;
}
template <int LW>
__device__ __forceinline__ uint32_t row_ballot(bool p, int lane) {       // bit k: lane k of the caller's row
    const unsigned long long b = __ballot(p) >> (lane & (64 - LW));
    return LW == 32 ? (uint32_t)b : (uint32_t)b & ((1u << (LW & 31)) - 1u);
}
template <int LW>
__device__ __forceinline__ uint32_t row_read(uint32_t v, uint32_t j, int lane) {   // v of lane j of the caller's row (j the same within the row)
    return (uint32_t)__builtin_amdgcn_ds_bpermute((int)((((uint32_t)lane & (uint32_t)(64 - LW)) + j) << 2), (int)v);
}

__global__ void k_tiles_from_blocks(const uint8_t *__restrict__ same_blk, uint32_t E, uint32_t W, uint32_t tile, uint32_t n_tiles, TileChain *__restrict__ tchain, uint32_t *__restrict__ step) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n_tiles) return;
    const unsigned long long t0 = (unsigned long long)t * tile;
    bool ok = t0 >= W;
    if (ok) {
        const unsigned long long q_end = min(t0 + tile + W - 1, (unsigned long long)E);
        for (unsigned long long b = t0 / ESC_TILE; b * ESC_TILE < q_end; b++) ok = ok && same_blk[b];
    }
    tchain[t] = TileChain{0, 0, ok ? 2u : 0u, 0};
    step[t] = 0;
}

// W-periodic tiles (config 3 is nothing else): fc[q] == fc[q - W] for every q the tile's matches can reach, so every position p has the
// key (min(W, E-p), W) (see k_match2).  Found by a pass of its own, straight from memory and without LDS -- inside k_match_chain the same
// check ran at that kernel's two blocks per CU and took 1.5 ms per GiB of periodic data; other tiles fail the first wavefront's sample
// (64 x 16 bytes) and cost next to nothing.  Nobody writes a periodic tile's keys (4 bytes per position for nothing): k_chain_periodic
// places the chain by arithmetic, k_tok_emit computes the key of a flagged position, k_chain_unknown stores them only if the general
// parse has to take over.  Every tile's record is (re)initialised here: walked = 2 for a periodic tile, 0 otherwise.
__global__ __launch_bounds__(256) void k_tile_periodic(const uint8_t *__restrict__ fc, uint32_t E, uint32_t W, uint32_t tile, TileChain *__restrict__ tchain, uint32_t *__restrict__ step) {
    const int tid = threadIdx.x;
    const long long t0 = (long long)blockIdx.x * tile;
    bool ok = t0 >= (long long)W;                                         // (the first window has nothing to repeat)
    if (ok) {
        const long long q_end = min(t0 + (long long)tile + (long long)W - 1, (long long)E);
        auto same16 = [&](long long q) {                                  // fc[q .. q+16) against the same bytes W earlier, clipped at q_end
            if (q >= q_end) return true;
            if ((W & 15u) == 0 && q + 16 <= q_end) {
                const uint4 x = *reinterpret_cast<const uint4 *>(fc + q), y = *reinterpret_cast<const uint4 *>(fc + q - W);
                return x.x == y.x && x.y == y.y && x.z == y.z && x.w == y.w;
            }
            bool eq = true;
            for (long long k = q; k < min(q + 16, q_end); k++) eq = eq && fc[k] == fc[k - W];
            return eq;
        };
        ok = tid >= 64 || same16(t0 + 16ll * tid);                        // the sample: the tile's first kilobyte
        if (__syncthreads_and(ok)) {
            for (long long q = t0 + 16ll * tid; q < q_end; q += 16ll * 256) ok = ok && same16(q);
            ok = __syncthreads_and(ok);
        } else ok = false;
    }
    if (tid == 0) { tchain[blockIdx.x] = TileChain{0, 0, ok ? 2u : 0u, 0}; step[blockIdx.x] = 0; }
}

// (r06) k_match_chain's visit of a position p that stands in a stretch of period P <= PMAX: b[i + P] == b[i] for the 64 bytes from p on
// and more -- a run of one byte (P = 1: zero-filled and sparse buffers), a line or a record repeated (sparse CSV, a log line over and
// over).  Such a visit has hundreds or thousands of candidates that agree further than the stage's 256 bytes; the rounds compare them
// eight bytes at a time, and the rule for more long candidates than a visit lists -- the farthest decides unless another has the
// position's byte where it stops -- gave nearly every tile up as "heavy": the sweep did the stream at 4096 compares a position, 0.4-0.6 GB/s.
//
// With r = the bytes from p on that keep the period (capped at min(W, E - p)), a candidate q at distance d that begins with p's unit U =
// b[p .. p + P) and keeps the period for R bytes matches min(r, R, d) bytes unless R == r -- then both stretches end together and the match
// goes on behind them.  The window's stretches are what lies between its MARKS, the positions i with b[i] != b[i - P]: marks absent on
// [a, b) make [a - P, b) one stretch.  Inside a stretch the candidates are the positions that begin with U, P apart; R and d both fall
// with the offset, so the first of them is the stretch's best and farthest candidate and the one with R == r the only other that can
// count.  In p's own stretch R > r: the farthest position a multiple of P back.  So: find P (lane j tries j + 1), r through memory, then
// the wavefront reads the window once, 512 bytes a trip, and every lane that finds a mark closes the stretch in front of it -- finds the
// first position that begins with U, evaluates its key, follows the R == r candidate if the bytes behind the two stretches agree.
//
// What it does not see: gaps between two marks inside one lane's eight bytes (six bytes at most).  A candidate that matches l >= P + 7 bytes has
// no mark on [q + P, q + l), so it lies in a stretch that is seen, at or behind that stretch's first position that begins with U: the
// maximum over the keys above is exact as soon as it is P + 7 or more -- else 0 is returned and the rounds decide.
// Returns L << 16 | distance.  sw: the stage; c_irel: p's staged offset; c_ipos: its stream position; zrel: the staged offset of stream
// position 0.  All arguments are the same in the 64 lanes.
template <int PMAX>
__device__ __forceinline__ uint32_t chain_period_visit(const uint32_t *sw, const uint8_t *fc, uint32_t c_irel, uint32_t c_capE, uint32_t c_ipos, uint32_t W, uint32_t zrel, int lane) {
    const uint8_t *sb = reinterpret_cast<const uint8_t *>(sw);
    const uint32_t wlo = max(c_irel - W, zrel);
    uint32_t P = 0;
    static_assert(PMAX + 24 + 8 <= HLMAX + 32, "the period is found in the bytes the stage holds behind a position");
    for (uint32_t p0 = 0; p0 < (uint32_t)PMAX && !P; p0 += 64) {          // the least period of the bytes from p on (what lies BEFORE p may be anything: the first of a row of equal lines)
        const uint32_t Pj = p0 + (uint32_t)lane + 1u;
        // (24 bytes that repeat at distance Pj -- eight for a run of one byte: a buffer with a byte in a hundred set has runs of a dozen
        //  zeros everywhere, and each visit of one that goes to the rounds instead compares four thousand candidates)
        bool ok = Pj <= (uint32_t)PMAX && lds_load8(sw, c_irel) == lds_load8(sw, c_irel + Pj);
        if (Pj > 1u) ok = ok && lds_load8(sw, c_irel + 8u) == lds_load8(sw, c_irel + 8u + Pj) && lds_load8(sw, c_irel + 16u) == lds_load8(sw, c_irel + 16u + Pj);
        const unsigned long long m = __ballot(ok);
        if (m) P = p0 + (uint32_t)__builtin_ctzll(m) + 1u;
    }
    if (!P) return 0u;
    const uint32_t lim0 = min(W, c_capE), T = P + 7u;
    auto wave_first_diff = [&](uint32_t d, uint32_t from, uint32_t lim) -> uint32_t {   // the first q in [from, lim) with fc[p + q] != fc[p - d + q], or lim
        const uint8_t *pa = fc + (size_t)c_ipos, *pb = pa - d;
        uint32_t res = lim;
        for (uint32_t base = from; base < lim && res == lim; base += 512u) {
            const uint32_t q = base + 8u * (uint32_t)lane;
            uint32_t mm = lim;
            if (q + 8 <= lim) {
                unsigned long long u, v;
                __builtin_memcpy(&u, pa + q, 8); __builtin_memcpy(&v, pb + q, 8);
                if (u != v) mm = q + ((uint32_t)__builtin_ctzll(u ^ v) >> 3);
            } else for (uint32_t k = q; k < lim && mm == lim; k++) if (pa[k] != pb[k]) mm = k;
            res = ~wave_max_u32(~mm);
        }
        return res;
    };
    // r: p's unit, then the bytes that repeat the byte P before them -- out of the stage as far as it shows them (a load from memory is a
    // microsecond or two, and a buffer with a byte in a hundred set has 160 such visits a tile), through memory behind that; capped at
    // lim0 (the stage is zero behind the stream's end)
    uint32_t r = 0xFFFFFFFFu;
    {
        const uint32_t k = P + 8u * (uint32_t)lane;
        if (k + 8u <= (uint32_t)HLMAX + 24u) { const unsigned long long x = lds_load8(sw, c_irel + k) ^ lds_load8(sw, c_irel + k - P); if (x) r = k + ((uint32_t)__builtin_ctzll(x) >> 3); }
        r = ~wave_max_u32(~r);
        if (r == 0xFFFFFFFFu) r = wave_first_diff(P, min(P + 512u, (uint32_t)HLMAX + 24u) / 8u * 8u - 8u, lim0);   // (from a multiple of eight the stage has shown alike)
        r = min(r, lim0);
    }
    if (r < T || (P > 1u && r < 2u * P + 16u)) return 0u;                  // (a unit that comes again once and a bit -- table rows that share their mark-up -- is the rounds' business: the window's scan costs a thousand instructions)
    auto begins_with_unit = [&](uint32_t q) {                              // b[q .. q + P) == U  (eight bytes at a time: up to seven bytes past the unit, which both sides repeat -- q + T and p + T lie inside their stretches)
        bool eq = true;
        for (uint32_t k = 0; k < P && eq; k += 8) eq = lds_load8(sw, q + k) == lds_load8(sw, c_irel + k);
        return eq;
    };
    const uint32_t after = r < (uint32_t)HLMAX ? sb[c_irel + r] : 0x100u;  // the byte behind p's stretch, if the stage shows it
    uint32_t carry = wlo + P, kb = 0;                                      // the offset behind the last mark so far ([wlo, wlo + P) counts as marked: no unit in front of it inside the window); the lane's best key
    const uint32_t scan_end = c_irel + P;                                  // (the marks of p's own unit too: is it the first of its stretch?)
    for (uint32_t cb = carry; cb < scan_end; cb += 512u) {
        const uint32_t o = cb + 8u * (uint32_t)lane;
        unsigned long long x = o < scan_end ? lds_load8(sw, o) ^ lds_load8(sw, o - P) : 0ull;
        if (o < scan_end && o + 8u > scan_end) x &= ~0ull >> (8u * (o + 8u - scan_end));   // from p + P on there is no mark for r bytes
        const uint32_t first = o + ((uint32_t)__builtin_ctzll(x | (1ull << 63)) >> 3);
        uint32_t pe = x ? o + 8u - ((uint32_t)__builtin_clzll(x) >> 3) : 0u;       // the offset behind the lane's last mark
        pe = max(pe, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)pe, 0x111, 0xF, 0xF, true));   // a running maximum over the lanes (as the sum in k_match_chain's 2b.)
        pe = max(pe, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)pe, 0x112, 0xF, 0xF, true));
        pe = max(pe, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)pe, 0x114, 0xF, 0xF, true));
        pe = max(pe, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)pe, 0x118, 0xF, 0xF, true));
        pe = max(pe, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)pe, RSN_DPP_ROW_BCAST15, 0xA, 0xF, false));
        pe = max(pe, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)pe, RSN_DPP_ROW_BCAST31, 0xC, 0xF, false));
        uint32_t a0 = __shfl_up(pe, 1);
        a0 = max(lane ? a0 : 0u, carry);                                  // the marks are absent on [a0, first): the stretch [a0 - P, first)
        carry = max(carry, (uint32_t)__builtin_amdgcn_readlane((int)pe, 63));
        uint32_t ds = 0;                                                  // the candidate whose stretch ends with p's, if it is to be followed
        if (x && first >= a0 - P + T) {
            uint32_t qf = 0xFFFFFFFFu;                                    // the stretch's first position that begins with U and has T bytes of the stretch in front of it
            for (uint32_t q = a0 - P; q < a0 && q + T <= first && q < c_irel && qf == 0xFFFFFFFFu; q++) if (begins_with_unit(q)) qf = q;
            if (qf != 0xFFFFFFFFu) {
                const uint32_t R = first - qf, d = c_irel - qf;
                kb = max(kb, (min(min(r, R), d) << 16) | d);              // (R == r: at least that; followed below if the bytes behind the stretches agree)
                if (R >= r && r < lim0 && first - r < c_irel && (after == 0x100u || after == sb[first]) && begins_with_unit(first - r)) ds = c_irel - (first - r);
            }
        }
        if (ds) {
            // ... out of the stage as far as it shows p's bytes, every lane for itself (sparse CSV: half a dozen of a visit's forty stretches
            // end like p's does, and each would be a load from memory the wavefront waits a microsecond or two for)
            const uint32_t lim1 = min(ds, c_capE);
            uint32_t off = r;
            bool open = true;
            while (open && off < lim1 && off + 8u <= (uint32_t)HLMAX + 24u) {
                const unsigned long long y = lds_load8(sw, c_irel + off) ^ lds_load8(sw, c_irel - ds + off);
                if (y) { off += (uint32_t)__builtin_ctzll(y) >> 3; open = false; } else off += 8u;
            }
            if (!open || off >= lim1) { kb = max(kb, (min(off, lim1) << 16) | ds); ds = 0; }
        }
        for (unsigned long long am = __ballot(ds != 0); am; am &= am - 1) {   // what goes on behind the stage: through memory, by the wavefront
            const int l = __builtin_ctzll(am);
            const uint32_t d1 = (uint32_t)__builtin_amdgcn_readlane((int)ds, l), L1 = wave_first_diff(d1, r, min(d1, c_capE));
            if (lane == l) kb = max(kb, (L1 << 16) | d1);
        }
    }
    uint32_t wb = wave_max_u32(kb);
    if (carry <= c_irel) {   // no mark in p's own unit: its stretch is [carry - P, p + r), the farthest candidate a multiple of P back
        const uint32_t d = (c_irel - (carry - P)) / P * P;
        if (d) wb = max(wb, (min(r, d) << 16) | d);
    }
    return (wb >> 16) < T ? 0u : wb;
}

template <class C>
__global__ __launch_bounds__(C::CTH) __attribute__((amdgpu_num_sgpr(80), amdgpu_waves_per_eu(8, 8))) void k_match_chain(ChainArgs a) {   // (80: see ChainArgs)
    constexpr int LW = 8;                                                 // lanes of a chain's home row: eight chains per wavefront (r03: a wavefront per chain 37.5 ms, four chains 31, sixteen 30.3, eight 28.8)
    constexpr int CT = C::CT, CTH = C::CTH, CSH = C::CSH, CS = C::CS, CH = C::CH, NS = C::NS;
    constexpr uint32_t TAGM = (1u << C::TAGB) - 1;
    // A list entry is staged offset << TAGB | tag.  The test of a candidate -- its start in [i - W, i) and the same tag -- is then one
    // subtraction, one rotation and one compare: with Q = (offset of i - 1) << TAGB | tag of i, Q - entry is (distance - 1) << TAGB when
    // the tags agree and has low bits set when they do not; rotated right by TAGB that is distance - 1, or something above any window.
    auto cand_q = [](uint32_t irel, uint32_t tag) { return ((irel - 1u) << C::TAGB) | tag; };
    auto cand_rot = [](uint32_t q, uint32_t e) { const uint32_t d = q - e; return (uint32_t)__builtin_amdgcn_alignbit(d, d, (uint32_t)C::TAGB); };   // < W: a candidate at distance rot + 1
    __shared__ __attribute__((aligned(1024))) uint32_t sw[C::STAGE / 4];   // fc[r0, r0 + STAGE), zero outside the stream  (the alignment puts it first in the LDS: its address is an instruction offset, not an add)
    __shared__ __attribute__((aligned(16))) uint8_t s_pool[(HNB / 2) * 4 + NS * 2];   // the bucket index; later the two jump arrays of the in-tile parse
    uint32_t *s_cur = reinterpret_cast<uint32_t *>(s_pool);              // two 16-bit counters per word: counts, then starts, then ends
    uint16_t *s_list = reinterpret_cast<uint16_t *>(s_pool + (HNB / 2) * 4);   // staged offset | tag << OFFB, grouped by bucket
    __shared__ __attribute__((aligned(16))) uint32_t s_claim[C::DUMP_BYTES / 4];   // positions somebody has taken, relative to t0 - CH
    __shared__ unsigned long long s_present[256];                         // per byte value: the 2^CSH-position blocks of the stage it occurs in
    __shared__ uint32_t s_part[CTH / 64];
    __shared__ uint32_t s_heavy, s_next, s_dense, s_votes;
    __shared__ uint32_t s_nstep, s_stepmin, s_stepmax;                    // commits whose match runs over its whole distance (L = distance >= HLMAX), and the range of those distances
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const uint32_t E = a.E, W = a.W;
    // the tile.  A partial launch takes it from a list (k_chain_verify's: the tiles that gave up as "dense", and -- top bit set -- the tiles
    // whose warm-up chain had not merged with the true chain by the time it entered them: those start from the tile before's exit instead)
    const uint32_t list_entry = (a.redo & 2u) ? chain_tail().redo_list[blockIdx.x] : blockIdx.x;
    const uint32_t bx = list_entry & 0x3FFFFFFFu;                         // (bit 31: walk from redo_start; bit 30: that chain alone, see nitems)
    const long long t0 = (long long)bx * CT;
    const long long r0 = t0 - CH - HWMAX;
    const uint8_t *sb = reinterpret_cast<const uint8_t *>(sw);
    if ((a.redo & 4u) && bx < HALO_TILES) return;                         // a section's halo: candidates for the tiles behind it, no chain of its own
    if (chain_tail().tchain[bx].walked == 2) return;                      // W-periodic (k_tile_periodic found so): no chain of its own, see there
    const bool raw = (a.redo & 8u) != 0;                                  // fc is the INPUT (nothing in it needs an escape): '<' -> FF on the way into the stage (lzss.go:373-377)
    for (uint32_t v = tid; v < C::STAGE / 16; v += CTH) {
        const long long P = r0 + 16ll * v;
        uint4 x = {0, 0, 0, 0};
        if (P >= 0 && P + 16 <= (long long)E) x = *reinterpret_cast<const uint4 *>(a.fc + P);
        else if (P + 16 > 0 && P < (long long)E) {
            uint32_t w[4] = {0, 0, 0, 0};
            for (int k = 0; k < 16; k++) { const long long q = P + k; if (q >= 0 && q < (long long)E) w[k >> 2] |= (uint32_t)a.fc[q] << (8 * (k & 3)); }
            x = {w[0], w[1], w[2], w[3]};
        }
        if (raw) { x.x |= bytes_equal(x.x, 0x3Cu); x.y |= bytes_equal(x.y, 0x3Cu); x.z |= bytes_equal(x.z, 0x3Cu); x.w |= bytes_equal(x.w, 0x3Cu); }
        reinterpret_cast<uint4 *>(sw)[v] = x;
    }
    for (int i = tid; i < HNB / 2; i += CTH) s_cur[i] = 0;
    for (int i = tid; i < C::DUMP_BYTES / 4; i += CTH) s_claim[i] = 0;
    for (int i = tid; i < 256; i += CTH) s_present[i] = 0;
    if (tid == 0) { s_heavy = 0; s_next = 0; s_dense = 0; s_votes = 0; s_nstep = 0; s_stepmin = 0xFFFFFFFFu; s_stepmax = 0; }
    __syncthreads();

    // ---- group the staged positions by bigram (as k_match_hash): candidates are [rlo, rhi)
    const long long q0 = max(0ll, t0 - CH);                               // the warm-up chain starts here
    const uint32_t zrel = (uint32_t)max(0ll, -r0);                        // staged offset of stream position 0 (or of r0)
    const uint32_t rlo = (uint32_t)(max(0ll, q0 - (long long)W) - r0), rhi = (uint32_t)(min((long long)E - 1, t0 + (long long)CT) - r0);
    for (uint32_t rel = tid; rel < (uint32_t)NS; rel += CTH) {
        if (rel < rlo || rel >= rhi) continue;
        const uint32_t b0 = sb[rel], h = (b0 << 5) | (sb[rel + 1] & 31u);
        atomicAdd(&s_cur[h >> 1], 1u << (16 * (h & 1)));
        const unsigned long long bit = 1ull << (rel >> CSH);
        if (!(s_present[b0] & bit)) atomicOr(&s_present[b0], bit);
    }
    __syncthreads();
    {
        constexpr int PER = HNB / 2 / CTH;                                // counter words per thread
        uint32_t sum = 0;
        for (int k = 0; k < PER; k++) { const uint32_t x = s_cur[tid * PER + k]; sum += (x & 0xFFFF) + (x >> 16); }
        uint32_t incl = sum;
        for (int dd = 1; dd < 64; dd <<= 1) { const uint32_t y = __shfl_up(incl, dd); if (lane >= dd) incl += y; }
        if (lane == 63) s_part[wv] = incl;
        __syncthreads();
        uint32_t run = incl - sum;
        for (int k = 0; k < wv; k++) run += s_part[k];
        for (int k = 0; k < PER; k++) {
            const uint32_t x = s_cur[tid * PER + k], c0 = x & 0xFFFF, c1 = x >> 16;
            s_cur[tid * PER + k] = run | ((run + c0) << 16);
            run += c0 + c1;
        }
    }
    __syncthreads();
    // scatter in rounds of 2^CSH consecutive offsets with a barrier between them: inside a bucket the
    // entries end up ordered by offset >> CSH, which is all the window bisection below needs
    // (bucket and entry are worked out by all threads at once, before the rounds: between two barriers only the atomic and the store)
    for (uint32_t base = 0; base < (uint32_t)NS; base += CTH) {
        const uint32_t rel = base + tid;
        const bool in = rel >= rlo && rel < rhi;
        const uint32_t b1 = sb[in ? rel + 1 : 0u];
        const uint32_t h = ((uint32_t)sb[in ? rel : 0u] << 5) | (b1 & 31u), ent = (rel << C::TAGB) | ((b1 >> 5) & TAGM);
        const uint32_t sh = 16 * (h & 1);
#pragma unroll
        for (int g = 0; g < (CTH >> CSH); g++) {
            if ((tid >> CSH) == g && in) {
                const uint32_t slot = (atomicAdd(&s_cur[h >> 1], 1u << sh) >> sh) & 0xFFFF;
                s_list[slot] = (uint16_t)ent;
            }
            __syncthreads();
        }
    }
    const uint16_t *ends = reinterpret_cast<const uint16_t *>(s_cur);    // ends[h]; the bucket starts at ends[h-1]

    // ---- chains.  Positions are counted from t0 - CH (kp); staged offset = HWMAX + kp.  Everything
    //      named u_* is the same in all 64 lanes.
    const uint32_t npos = (uint32_t)min((long long)CT, (long long)E - t0);
    const uint32_t kp_end = CH + npos;                                    // a chain stops when it leaves the tile
    uint32_t kp_first = (uint32_t)(q0 - (t0 - CH));                       // the warm-up start (CH in tile 0: the true start)
    if ((a.redo & 4u) && bx == HALO_TILES) kp_first = CH;                  // the first tile of a section: the true chain enters at its first position
    if (list_entry >> 31) kp_first = chain_tail().redo_start[blockIdx.x] - (uint32_t)(t0 - CH);   // the true entry as k_chain_verify worked it out: the tile before's exit, or a whole-distance stretch's arithmetic
    // (a tile of a whole-distance stretch walked from its predicted entry: the chain is two or three visits long, and the other 64
    //  starts would each verify a 4 KiB match for nothing -- should the prediction be wrong, the one chain walks the tile alone)
    // (the head of a W-periodic input -- pfrom set -- is two or three tiles that a whole call waits for, the first of them incompressible:
    //  a start every 32 positions there, so that every row of the block has a stretch of it to walk -- config 3's head 0.22 -> ? ms)
    const uint32_t cs = a.pfrom != 0xFFFFFFFFu ? 32u : (uint32_t)CS;
    const uint32_t nitems = (list_entry & 0x40000000u) ? 1u : 1 + (npos + cs - 1) / cs;
    // a wavefront that gives up (s_heavy) also pushes the start counter past every item: the others
    // finish the chain they are on (at most CS-odd positions) and find nothing more to start
    constexpr uint32_t GIVE_UP = 0x40000000u;
    // Nearly incompressible data (chain steps of one or two positions) is the one case where walking
    // chains loses: almost every position is visited, and a whole wavefront per visit costs more than
    // k_match_hash's one lane per position with its near-empty buckets.  A wavefront whose chain
    // advanced less than two positions per visit hands the strip to k_match_hash.
    // (64 visits: at 32, a few text tiles in every ten thousand -- two or three rare words in a row are a few dozen one-byte steps --
    //  gave up, and each such tile costs the stream a second look; noise gives up at 64 as surely as at 32)
        // (r06: HALF the walkers, not a quarter -- 100-byte records with 20 random bytes each have a fifth of the starts inside the random bytes:
    //  54 % of the tiles gave up and the bucket search did them at 2.7 GB/s; with the half 23 GB/s.  Noise, where every walker says so
    //  after its eighth visit, gives up at the same moment either way: 16 MiB 1.97 ms, 1 GiB 41.7 ms before and after.)
#ifndef RSN_DENSE_DIV
#define RSN_DENSE_DIV 2
#endif
    constexpr uint32_t DENSE_EVALS = 64, DENSE_VOTE = 8;
    constexpr uint32_t LONG_CAP = 8;                                      // candidates with a common prefix of HLMAX bytes and more that a visit follows through memory
    // (Structured control flow on purpose -- no break / continue out of the walk: with them the compiler turns the loop into a
    //  state machine and spends ~40 scalar instructions per visit on its masks, and scalar issue is this kernel's bound.)
    {
        // ---- the same walk with K = 64 / LW chains per wavefront.  This kernel is bound by vector issue (scripts/valu_probe.cpp: a wave64
        //      integer VALU instruction holds its SIMD for ~4 cycles however many wavefronts share it; the walk's counters come to 3.4), and
        //      a visit on text examines a dozen candidates (half the visits: eight or fewer): with a wavefront per chain most of those
        //      instructions work on lanes that hold nothing, and the visit's fixed part -- claim, bucket bounds, the position's bytes, the
        //      maximum, the store -- is paid per chain.  Here
        //        1. each chain has a HOME ROW of LW lanes that does its bookkeeping (the rows diverge like threads do; what the wave version
        //           calls u_* is the same within a row): claim a position, find the bucket, trim it to the window at both ends;
        //        2. the candidates of all K visits are DEALT over the wavefront in rows of LW entries: row-slot s of round r takes the
        //           (K r + s)-th row of the concatenated lists, whichever chain it belongs to (its parameters come from LDS), so a visit with
        //           3 candidates and one with 300 cost 1 + 19 rows = 5 rounds, not 19; a row's maximum goes to its chain by an LDS atomic;
        //        3. the home rows commit (and take the rare paths: long matches, no bigram match).
        static_assert(LW == 16 || LW == 8 || LW == 4, "a DPP row, half of one, or a quad per chain");
        constexpr int K = 64 / LW, NWV = CTH / 64;
        constexpr uint32_t NARROW = 64;                        // a bucket is trimmed while it holds more entries than this
        constexpr uint32_t HEAVY_ROWS = 16;   // a visit with this many rows of candidates takes the whole wavefront (a row index is four bits of the row map)
        constexpr uint32_t LCAP = K > 8 ? 4 : LONG_CAP;                       // long candidates listed per chain (sixteen chains: the lists' LDS)
        // a wavefront's scratch, one record (one base address in a register; the members are offsets in the LDS instructions)
        struct WaveScratch {
            uint32_t par[K][8];                                               // a visit's parameters for the lanes its candidates are dealt to
            uint32_t best[K], lcnt[K], lfar[K];                               // its maximum; its long candidates: how many, the farthest
            uint32_t llist[K][LCAP][2];                                       // the first LCAP long candidates (distance | limit << 16, bytes known equal)
            uint8_t rowmap[K * HEAVY_ROWS];                                   // dealt row -> chain << 4 | row of that chain (a chain with HEAVY_ROWS rows is not dealt)
        };
        __shared__ __attribute__((aligned(16))) WaveScratch s_ws[NWV];
        WaveScratch &ws = s_ws[wv];
        static_assert(HEAVY_ROWS <= 16 && K <= 16, "chain and row share a byte of the row map");
        const int rl = lane & (LW - 1), slot = lane / LW;                     // lane within the row; the row (= the home chain)
        const bool leader = rl == 0;
        if (leader) { ws.best[slot] = 0; ws.lcnt[slot] = 0; ws.lfar[slot] = 0; }
        for (int i = lane; i < K * 8; i += 64) (&ws.par[0][0])[i] = 0;
        bool alive = true, voted = false;
        uint32_t next = 0xFFFFFFFFu, visits = 0, from_kp = 0;
        while (__ballot(alive)) {
            // ---- 1. home rows: a position to evaluate -- where the chain in hand landed, unless that is somebody else's already (that row
            //      walks the rest) or beyond the tile: then a new start, and the next one if that is taken too
            uint32_t kp = next;
            next = 0xFFFFFFFFu;
            bool mine = false;
            while (alive && !mine) {
                if (kp >= kp_end) {
                    uint32_t kq = 0;
                    if (leader) kq = atomicAdd(&s_next, 1u);
                    kq = row_read<LW>(kq, 0, lane);
                    alive = kq < nitems;
                    kp = kq == 0 ? kp_first : CH + (kq - 1) * cs;
                    visits = 0; from_kp = kp;
                }
                if (alive) {
                    uint32_t old = 0;
                    if (leader) old = atomicOr(&s_claim[kp >> 5], 1u << (kp & 31));
                    mine = row_ballot<LW>(leader && ((old >> (kp & 31)) & 1), lane) == 0;
                }
                if (!mine) kp = 0xFFFFFFFFu;
            }
            // a position in the W-periodic rest of the head's stream (r06): key (min(W, E - p), W) whatever the bytes are -- k_tile_periodic's
            // argument, position by position (searched, each start in config 3's second half-tile verifies a 4 KiB match through memory)
            if (mine && (uint32_t)(t0 - CH) + kp >= a.pfrom) {
                const uint32_t ipos = (uint32_t)(t0 - CH) + kp, L = min(W, E - ipos);
                if (leader) a.keys[ipos] = (L << 16) | W;
                next = kp + L; visits++;
                mine = false;
            }
            // (what 3. needs of the position it works out again from kp: fewer registers live across the rounds -- the kernel has 64)
            uint32_t nrows = 0, irel = 0, capE = 0, tag = 0, lo = 0, hi = 0;
            unsigned long long pat0 = 0;
            if (mine) {
                irel = HWMAX + kp; capE = E - ((uint32_t)(t0 - CH) + kp);
                const uint32_t b0 = sb[irel], b1 = sb[irel + 1];
                const uint32_t h = (b0 << 5) | (b1 & 31u);
                tag = (b1 >> 5) & TAGM;
                lo = h ? (uint32_t)ends[h - 1] : 0u; hi = ends[h];
                const uint32_t blk_lo = (irel - W) >> CSH, blk_i = irel >> CSH;
                pat0 = lds_load8(sw, irel + C::OFF0);
                // (r04, measured and dropped: the window's slice of the bucket by INTERPOLATION -- entries are ordered by block and text spreads
                //  a bigram evenly over the stage, so four guesses at each end, checked exactly by the block of the entry in front of / at the
                //  guess, trim any bucket of more than two rows for one LDS read.  It works -- 85.5 entries per visit become 36.0 instead of
                //  40.2 -- and changes nothing: 28.17 against 28.16 ms.  RSN_CHAIN_STATS says why: of the 36 entries dealt per visit 27.7 ARE
                //  candidates (in the window, same tag), the trips below already do four fifths of the trimming, and half of all candidate
                //  rounds belong to the 12 % of visits with a bucketful -- the commonest bigrams of a Zipf text -- which no trimming shortens.)
                // the bucket's entries are ordered by block of 2^CSH positions: LW samples per trip trim the entries before the
                // window's first block and those after the position's own
                // (while it pays: a trip costs the wavefront about as much as two dealt rows)
                bool narrowing = hi - lo > NARROW;
                while (narrowing) {
                    const uint32_t n = hi - lo, stride = (n + LW - 1) / LW;
                    const uint32_t idx = min(lo + __umul24((uint32_t)rl, stride), hi - 1);
                    const uint32_t blk = (uint32_t)s_list[idx] >> (C::TAGB + CSH);
                    const uint32_t in = row_ballot<LW>(blk >= blk_lo, lane), after = row_ballot<LW>(blk > blk_i, lane);
                    const uint32_t first = in ? (uint32_t)__builtin_ctz(in) : (uint32_t)LW;   // samples below `first` lie before the window
                    const uint32_t skip = max(first, 1u) - 1u;                                 // whole strides known to
                    const uint32_t nlo = min(lo + __umul24(skip, stride), hi - 1);
                    const uint32_t nhi = after ? min(lo + __umul24((uint32_t)__builtin_ctz(after), stride), hi - 1) : hi;   // that sample and everything behind it: after i
                    narrowing = (nlo != lo || nhi != hi) && nhi - nlo > NARROW;
                    lo = nlo; hi = max(nhi, nlo);
                }
                nrows = (hi - lo + LW - 1) / LW;
            }
            // ---- 2. the candidates
            // one candidate per lane: its key L << 16 | distance (0: none).  c_*: the visit it belongs to, `ch` its chain.  A candidate that
            // agrees with the position for HLMAX bytes and more -- further than the stage reaches -- is put on the chain's list: the home row
            // follows up to LONG_CAP of them through memory (see the wave version); its key stays as the lower bound it is.
            auto eval = [&](bool valid, uint32_t e, uint32_t c_irel, uint32_t c_q, unsigned long long c_pat0, uint32_t c_capE, uint32_t ch, uint32_t &long_dn) -> uint32_t {
                const uint32_t rel = e >> C::TAGB, rot = cand_rot(c_q, e), dn = rot + 1u;
                // candidate start in [i-W, i) and the same bigram up to the tag: one compare (a tag difference lands above any W)
                const bool ok = valid && rot < W;
                uint32_t lim = 0, off = C::OFF0;
                unsigned long long x = 1;
                if (ok) {
                    lim = min(dn, c_capE);                                    // entirely inside the window, and inside the stream
                    x = lds_load8(sw, rel + off) ^ c_pat0;
                    uint32_t room = x == 0 ? lim : 0u;
                    while (off + 8 < room) {                                  // longer than eight bytes: this lane goes on, eight at a time
                        off += 8;
                        x = lds_load8(sw, rel + off) ^ lds_load8(sw, c_irel + off);
                        if (x != 0 || off + 8 >= HLMAX) room = 0;             // (the stage reaches HLMAX bytes past a position)
                    }
                }
                const bool fl = x == 0 && off + 8 >= HLMAX;
                long_dn = fl ? dn : 0u;
                if (__ballot(fl)) {                                           // (runs and short periods: every candidate of a visit -- one atomic per row, not per lane)
                    const uint32_t rm = row_ballot<LW>(fl, lane);
                    if (rm) {
                        const uint32_t far = row_max_u32<LW>(fl ? dn : 0u), first = (uint32_t)__builtin_ctz(rm);
                        uint32_t k0 = 0;
                        if ((uint32_t)rl == first) { k0 = atomicAdd(&ws.lcnt[ch], (uint32_t)__builtin_popcount(rm)); atomicMax(&ws.lfar[ch], far); }
                        k0 = row_read<LW>(k0, first, lane);
                        const uint32_t k = k0 + (uint32_t)__builtin_popcount(rm & ((1u << rl) - 1u));
                        if (fl && k < LCAP) { ws.llist[ch][k][0] = dn | (lim << 16); ws.llist[ch][k][1] = off + 8; }
                    }
                }
                const uint32_t nb = x ? (uint32_t)__builtin_ctzll(x) >> 3 : 8u;
                uint32_t len = min(off + nb, lim);
                if (C::OFF0 < 2) len = len < 2 ? 0u : len;                    // the untagged bit of the second byte differed
                return len ? (len << 16) | dn : 0u;                           // longest, then farthest back (bytes.Index, lzss.go:419)
            };
            // 2a. a visit with a bucketful of candidates (few-letter alphabets, runs, the common bigrams of text) takes the whole wavefront,
            //     round after round, its parameters in scalar registers: dealt row by row each round would fetch them again
            unsigned long long hm = __ballot(leader && nrows >= HEAVY_ROWS);
            if (nrows >= HEAVY_ROWS) nrows = 0;
            else if (mine && leader) {
                *reinterpret_cast<uint4 *>(&ws.par[slot][0]) = uint4{irel, cand_q(irel, tag), (uint32_t)pat0, (uint32_t)(pat0 >> 32)};
                *reinterpret_cast<uint4 *>(&ws.par[slot][4]) = uint4{lo, hi, capE, 0u};
            }
            if constexpr (C::RUNS) {
                // (r06) a heavy visit of a position in a stretch of period <= 64 -- a run of a byte, a short line repeated: resolved from the
                //  window's stretches (chain_period_visit) before any candidate round.  In the kernel instance for such streams only:
                //  never entered, the code still costs text 0.7 of 27.7 ms (registers of a kernel at its limits, LEDGER.md).
                //  (Rows of a table that share 24 bytes with the row before -- "</td></tr>\n<tr><td>" -- have a period and no stretch to
                //  speak of: resolving them this way took such data from 27 to 11 GB/s, hence the function's second gate -- two units and
                //  more behind the first -- which a row like that fails after a hundred instructions: 26.)
                for (unsigned long long hr = hm; hr; hr &= hr - 1) {
                    const int hl = __builtin_ctzll(hr);
                    const uint32_t c_irel = (uint32_t)__builtin_amdgcn_readlane((int)irel, hl);
                    const uint32_t wb = chain_period_visit<64>(sw, a.fc, c_irel, (uint32_t)__builtin_amdgcn_readlane((int)capE, hl), (uint32_t)(t0 - CH) + (c_irel - HWMAX), W, zrel, lane);
                    if (wb) {                                             // (exact; as long as a candidate followed through memory: 3. counts it as one)
                        if (lane == 0) { ws.best[(uint32_t)hl / LW] = wb; if ((wb >> 16) >= (uint32_t)HLMAX) ws.lcnt[(uint32_t)hl / LW] = 0x80000000u; }
                        hm &= ~(1ull << hl);
                    }
                }
            }
            while (hm) {
                const int hl = __builtin_ctzll(hm);
                hm &= hm - 1;
                const uint32_t c_irel = (uint32_t)__builtin_amdgcn_readlane((int)irel, hl), c_q = (uint32_t)__builtin_amdgcn_readlane((int)cand_q(irel, tag), hl),
                               c_lo = (uint32_t)__builtin_amdgcn_readlane((int)lo, hl), c_hi = (uint32_t)__builtin_amdgcn_readlane((int)hi, hl),
                               c_capE = (uint32_t)__builtin_amdgcn_readlane((int)capE, hl);
                const unsigned long long c_pat0 = (unsigned long long)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)pat0, hl) |
                                                  ((unsigned long long)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(pat0 >> 32), hl) << 32);
                uint32_t wb = 0, n_long = 0, far = 0;
                for (uint32_t base = c_lo; base < c_hi; base += 64) {
                    const uint32_t idx = base + (uint32_t)lane;
                    const bool valid = idx < c_hi;
                    const uint32_t e = s_list[valid ? idx : c_lo];
                    uint32_t long_dn;
                    wb = max(wb, eval(valid, e, c_irel, c_q, c_pat0, c_capE, (uint32_t)hl / LW, long_dn));
                    // Runs and short periods: thousands of candidates that all agree for HLMAX bytes and more.  Once more of them are known
                    // than are followed up, only the farthest counts (see 3.), and the entries come farthest first, block by block: when
                    // this round's last entry lies in a later block than the farthest long candidate, nothing behind it can matter.
                    const unsigned long long lm = __ballot(long_dn != 0);
                    if (lm) {
                        n_long += (uint32_t)__builtin_popcountll(lm);
                        far = max(far, wave_max_u32(long_dn));
                        const uint32_t last_blk = (uint32_t)__builtin_amdgcn_readlane((int)e, 63) >> (C::TAGB + CSH);   // (lane 63 holds the round's last entry, or re-reads the first: then the loop ends anyway)
                        if (n_long > LCAP && base + 64 < c_hi && last_blk > ((c_irel - far) >> CSH)) break;
                    }
                }
                wb = wave_max_u32(wb);
                if (lane == 0) ws.best[(uint32_t)hl / LW] = wb;
            }
            // 2b. the others, dealt in rows.  Which chain a row belongs to comes from a table the home rows fill in: chain c's rows are
            //     the entries [rows of chains 0 .. c-1 together, + its own).
            // (rows of the chains up to and including this lane's: a DPP prefix sum over the wavefront of the leaders' counts)
            uint32_t pre = leader ? nrows : 0u;
            pre += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)pre, 0x111, 0xF, 0xF, true);   // row_shr:1 (zero shifted in)
            pre += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)pre, 0x112, 0xF, 0xF, true);   // row_shr:2
            pre += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)pre, 0x114, 0xF, 0xF, true);   // row_shr:4
            pre += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)pre, 0x118, 0xF, 0xF, true);   // row_shr:8: every lane has its DPP row's sum up to itself
            pre += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)pre, RSN_DPP_ROW_BCAST15, 0xA, 0xF, false);   // DPP rows 1 and 3 take in the row before
            pre += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)pre, RSN_DPP_ROW_BCAST31, 0xC, 0xF, false);   // rows 2 and 3 take in lane 31
            const uint32_t n_all = (uint32_t)__builtin_amdgcn_readlane((int)pre, 63);
            {
                const uint32_t mine0 = pre - nrows;                               // rows of the chains before this one
                // (at most HEAVY_ROWS - 1 rows: a fixed number of predicated byte stores -- as a loop the compiler vectorised it into
                //  ~47 VALU instructions per wavefront-iteration, a tenth of the walk: r04)
#pragma unroll
                for (uint32_t t = 0; t < (HEAVY_ROWS + LW - 1) / LW; t++) {
                    const uint32_t j = (uint32_t)rl + t * LW;
                    if (j < nrows) ws.rowmap[mine0 + j] = (uint8_t)(((uint32_t)slot << 4) | j);
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
            for (uint32_t g0 = 0; g0 < n_all; g0 += K) {
                const uint32_t g = g0 + (uint32_t)slot;
                const uint32_t rm = ws.rowmap[g < n_all ? g : 0u], ch = rm >> 4;
                const uint4 p0 = *reinterpret_cast<const uint4 *>(&ws.par[ch][0]), p1 = *reinterpret_cast<const uint4 *>(&ws.par[ch][4]);
                const uint32_t idx = p1.x + (rm & 0xFu) * LW + (uint32_t)rl;
                const bool valid = g < n_all && idx < p1.y;
                uint32_t long_dn;
                const uint32_t key = eval(valid, s_list[valid ? idx : 0u], p0.x, p0.y, (unsigned long long)p0.z | ((unsigned long long)p0.w << 32), p1.z, ch, long_dn);
                const uint32_t rbest = row_max_u32<LW>(key);
                if (leader && rbest) atomicMax(&ws.best[ch], rbest);
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
            if constexpr (C::RUNS) {
                // (r06) ... and a visit the rounds leave with more long candidates than it lists -- a longer line repeated, a bucket of fewer
                //  than 128 -- the same way, periods up to 192, before the home row would let the farthest decide or give the tile up
                for (unsigned long long lm = __ballot(mine && leader && ws.lcnt[slot] > LCAP && !(ws.lcnt[slot] >> 31)); lm; lm &= lm - 1) {
                    const int hl = __builtin_ctzll(lm);
                    const uint32_t c_kp = (uint32_t)__builtin_amdgcn_readlane((int)kp, hl), c_ipos = (uint32_t)(t0 - CH) + c_kp;
                    const uint32_t wb = chain_period_visit<192>(sw, a.fc, HWMAX + c_kp, E - c_ipos, c_ipos, W, zrel, lane);
                    if (wb && lane == 0) { ws.best[(uint32_t)hl / LW] = max(ws.best[(uint32_t)hl / LW], wb); ws.lcnt[(uint32_t)hl / LW] = 0x80000000u; }
                }
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                __builtin_amdgcn_wave_barrier();
            }
            // ---- 3. home rows: commit
            if (mine) {
                const uint32_t ipos = (uint32_t)(t0 - CH) + kp, irel = HWMAX + kp, capE = E - ipos;
                uint32_t best = ws.best[slot];
                const uint32_t lcnt = ws.lcnt[slot], long_far = ws.lfar[slot];
                const bool longm = lcnt != 0;
                if (leader) { ws.best[slot] = 0; if (longm) { ws.lcnt[slot] = 0; ws.lfar[slot] = 0; } }
                // the first q in [from, lim) with fc[ipos + q] != fc[ipos - d + q], or lim: the row's lanes take eight bytes each per trip
                auto first_diff = [&](uint32_t d, uint32_t from, uint32_t lim) -> uint32_t {
                    const uint8_t *pa = a.fc + (size_t)ipos, *pb = pa - d;
                    uint32_t mm = lim;
                    for (uint32_t q = from + 8u * (uint32_t)rl; q < lim && mm == lim; q += 8u * LW) {
                        if (q + 8 <= lim) {
                            unsigned long long u, v;
                            __builtin_memcpy(&u, pa + q, 8); __builtin_memcpy(&v, pb + q, 8);
                            if (u != v) mm = q + ((uint32_t)__builtin_ctzll(u ^ v) >> 3);
                        } else for (uint32_t k = q; k < lim && mm == lim; k++) if (pa[k] != pb[k]) mm = k;
                    }
                    return ~row_max_u32<LW>(~mm);
                };
                auto in_byte_run = [&]() {                                   // the HLMAX bytes from the position on: one byte repeated?
                    const unsigned long long zz = 0x0101010101010101ull * sb[irel];
                    bool same = true;
                    for (uint32_t q = 8u * (uint32_t)rl; q < (uint32_t)HLMAX; q += 8u * LW) same = same && lds_load8(sw, irel + q) == zz;
                    return row_ballot<LW>(!same, lane) == 0;
                };
                auto commit = [&](uint32_t key) {
                    a.keys[ipos] = key;
                    next = kp + max(1u, key >> 16);                           // lzss.go:139-142: a reference skips size-1 positions
                    visits++;
                };
                if (best != 0 && !longm) commit(best);                        // the common case first
                else {
                    bool giveup_heavy = false, giveup_dense = false;
                    if (longm) {
                        if (lcnt >> 31) {}                                    // resolved by the wavefront from the window's stretches (chain_period_visit)
                        else if (lcnt <= LCAP) {                                   // every long candidate is followed to its end: the maximum is exact
                            // (farthest first: a candidate at distance d matches d bytes at most, so once the best reaches further than the
                            //  farthest one left, the rest cannot win -- a 1000-periodic stream follows one candidate over 4000 bytes, not four)
                            uint32_t done = 0;
                            for (uint32_t t = 0; t < lcnt; t++) {
                                uint32_t pick = 0, dj = 0;
                                for (uint32_t k = 0; k < lcnt; k++) {
                                    const uint32_t d = ws.llist[slot][k][0] & 0xFFFFu;
                                    if (!((done >> k) & 1u) && d > dj) { dj = d; pick = k; }
                                }
                                done |= 1u << pick;
                                if ((best >> 16) > dj) break;
                                best = max(best, (first_diff(dj, ws.llist[slot][pick][1], ws.llist[slot][pick][0] >> 16) << 16) | dj);
                            }
                        } else if (in_byte_run()) {
                            // (r06) The position stands in a run of one byte z, HLMAX of it and more ahead: zero-filled and sparse buffers, runs.
                            // Every candidate that agrees for HLMAX bytes lies in a run of z too; chain_period_visit's argument for P = 1, by the row: the
                            // window's runs of z out of the stage, 64 bytes a trip -- a dozen candidates with exact lengths instead of
                            // thousands that all agree further than the stage reaches (the farthest of which decided, or the strip went to
                            // the sweep: a buffer with a byte in a hundred set took 0.5 GB/s).  Runs of fewer than HLMAX - 16 bytes are left
                            // out: their candidates have their exact keys from the rounds.
                            const uint32_t lim0 = min(W, capE), r = first_diff(1u, 1u, lim0);
                            const unsigned long long zz = 0x0101010101010101ull * sb[irel];
                            uint32_t run_a = max(irel - W, zrel);
                            for (uint32_t cb = run_a; cb < irel; cb += 8u * LW) {
                                const uint32_t o = cb + 8u * (uint32_t)rl;
                                unsigned long long x = o < irel ? lds_load8(sw, o) ^ zz : 0ull;
                                if (o < irel && o + 8u > irel) x &= ~0ull >> (8u * (o + 8u - irel));   // the position's own bytes are z
                                uint32_t first = 0xFFFFFFFFu, last = 0;                      // the trip's first byte that is not z, and the offset behind its last
                                if (x) { first = o + ((uint32_t)__builtin_ctzll(x) >> 3); last = o + 8u - ((uint32_t)__builtin_clzll(x) >> 3); }
                                first = ~row_max_u32<LW>(~first); last = row_max_u32<LW>(last);
                                if (first != 0xFFFFFFFFu) {
                                    const uint32_t rq = first - run_a, dmax = irel - run_a;   // the run [run_a, first) ends here
                                    if (rq + 16u >= HLMAX) {
                                        best = max(best, (min(min(r, rq), dmax) << 16) | dmax);   // (rq == r: at least that)
                                        if (rq >= r && r < lim0) { const uint32_t ds = irel - first + r; best = max(best, (first_diff(ds, r, min(ds, capE)) << 16) | ds); }
                                    }
                                    run_a = last;
                                }
                            }
                            if (run_a < irel) best = max(best, (min(r, irel - run_a) << 16) | (irel - run_a));
                        } else {                                              // (the wave version explains why the farthest long candidate decides)
                            const uint32_t Lp = min(long_far, capE);
                            const uint32_t mm = first_diff(long_far, 0, Lp);
                            if (mm == Lp) best = max(best, (Lp << 16) | long_far);
                            else {
                                const uint32_t want = a.fc[(size_t)ipos + mm];
                                const uint32_t b1 = sb[irel + 1], h = ((uint32_t)sb[irel] << 5) | (b1 & 31u), tag = (b1 >> 5) & TAGM;
                                const uint32_t blo = h ? (uint32_t)ends[h - 1] : 0u, bhi = ends[h];
                                // Who can outlast the farthest: a candidate further back than mm bytes that has the position's byte at offset mm --
                                // and, r06, its first eight bytes and the stage's last eight, which every candidate that agrees for HLMAX bytes
                                // has (sorted lines: "aba\n" repeated after "ab\n" repeated -- a third of the earlier stretch's positions
                                // have the right byte at offset mm and agree with the position for three bytes; the byte alone sent 125 of
                                // 8192 tiles to the sweep, 12 of the call's 24 ms).  Up to eight of them are followed to their ends: exact.
                                // (counted first: a period broken now and then has every multiple of the period beyond mm among them, and
                                //  following eight of sixteen before giving up cost more than giving up at once)
                                best = max(best, (mm << 16) | long_far);
                                auto contender = [&](uint32_t idx, uint32_t &dn) {
                                    dn = cand_rot(cand_q(irel, tag), s_list[idx]) + 1u;
                                    return dn <= W && dn > mm && dn <= ipos && a.fc[(size_t)ipos - dn + mm] == want &&
                                           lds_load8(sw, irel - dn) == lds_load8(sw, irel) && lds_load8(sw, irel - dn + HLMAX - 8) == lds_load8(sw, irel + HLMAX - 8);
                                };
                                uint32_t n_cont = 0;
                                for (uint32_t idx0 = blo; idx0 < bhi && n_cont <= 8u; idx0 += LW) {
                                    uint32_t dn;
                                    n_cont += (uint32_t)__builtin_popcount(row_ballot<LW>(idx0 + (uint32_t)rl < bhi && contender(idx0 + (uint32_t)rl, dn), lane));
                                }
                                if (n_cont > 8u) giveup_heavy = true;
                                else if (n_cont) for (uint32_t idx0 = blo; idx0 < bhi; idx0 += LW) {
                                    uint32_t dn = 0;
                                    const bool cand = idx0 + (uint32_t)rl < bhi && contender(idx0 + (uint32_t)rl, dn);
                                    for (uint32_t cm = row_ballot<LW>(cand, lane); cm; cm &= cm - 1) {
                                        const uint32_t d1 = row_read<LW>(dn, (uint32_t)__builtin_ctz(cm), lane);
                                        best = max(best, (first_diff(d1, 0, min(d1, capE)) << 16) | d1);
                                    }
                                }
                            }
                        }
                    }
                    if (best == 0 && !giveup_heavy) {                         // no bigram of the window matches: L = 1 iff the byte occurs in the window at all
                        // (with 64 / LW chains per wavefront the block has that many times the walkers: if each walked DENSE_EVALS visits
                        //  before one gave up, the whole tile would have been evaluated by then.  A walker that looks dense after
                        //  DENSE_VOTE visits says so once; a quarter of the walkers saying so is the tile's verdict.)
                        if (!(a.redo & 1u) && bx != 0 && kp - from_kp < 2 * visits) {
                            if (visits >= DENSE_EVALS) giveup_dense = true;
                            else if (!voted && visits >= DENSE_VOTE) {
                                voted = true;
                                uint32_t v = 0;
                                if (leader) v = atomicAdd(&s_votes, 1u) + 1u;
                                giveup_dense = row_read<LW>(v, 0, lane) >= (uint32_t)(CTH / LW / RSN_DENSE_DIV);
                            }
                        }
                        if (!giveup_dense) {
                            const uint32_t ws = max(irel - min(W, irel), zrel);
                            const uint32_t fb_lo = (ws + (1u << CSH) - 1) >> CSH, fb_hi = irel >> CSH;
                            bool hit = false;
                            const uint32_t b0 = sb[irel];
                            if (fb_lo < fb_hi) {
                                const unsigned long long m = (fb_hi >= 64 ? ~0ull : (1ull << fb_hi) - 1) & ~((1ull << fb_lo) - 1);
                                hit = (s_present[b0] & m) != 0;
                            }
                            if (!hit) {                                       // the two ragged ends, four bytes per lane and trip
                                // (r04: byte by byte, a row's 8 lanes took up to 32 trips per end -- every visit of incompressible data comes through
                                //  here, config 3's first tile 4096 times in a row: 0.36 ms of that 1.4 ms call)
                                const uint32_t e1r = min(fb_lo << CSH, irel), s2 = max(min(fb_hi << CSH, irel), fb_lo < fb_hi ? ws : e1r);
                                auto any_eq = [&](uint32_t q0, uint32_t q1) {  // does byte b0 occur at a staged offset in [q0, q1)?  one aligned dword per lane and trip
                                    bool f = false;
                                    for (uint32_t wq = (q0 >> 2) + (uint32_t)rl; 4u * wq < q1; wq += LW) {
                                        uint32_t m = bytes_equal(sw[wq], b0);          // 0xFF in every byte that equals b0 (exact per byte)
                                        const uint32_t p0 = 4u * wq;
                                        if (p0 < q0) m &= 0xFFFFFFFFu << (8u * (q0 - p0));
                                        if (p0 + 4u > q1) m &= 0xFFFFFFFFu >> (8u * (p0 + 4u - q1));
                                        f = f || m != 0;
                                    }
                                    return f;
                                };
                                const bool f = any_eq(ws, e1r) || any_eq(s2, irel);
                                hit = row_ballot<LW>(f, lane) != 0;
                            }
                            best = hit ? (1u << 16) : 0u;
                        }
                    }
                    if (longm && !giveup_heavy && (best >> 16) == (best & 0xFFFFu) && leader) {
                        atomicAdd(&s_nstep, 1u); atomicMin(&s_stepmin, best & 0xFFFFu); atomicMax(&s_stepmax, best & 0xFFFFu);
                    }
                    if (giveup_heavy || giveup_dense) {
                        if (leader) { atomicOr(&s_next, GIVE_UP); if (giveup_heavy) s_heavy = 1; else s_dense = 1; }
                        alive = false;
                    } else commit(best);
                }
            }
        }
    }
    __syncthreads();
    // (the thread's index once more, opaque to the compiler: or what the prologue derived from it is kept for the epilogue's loops and,
    //  the walk needing every register, spilled across it)
    const int etid = (int)vec((uint32_t)threadIdx.x), elane = etid & 63, ewv = etid >> 6;
    const ChainTail T = chain_tail();
    if (etid == 0 && s_heavy) T.heavy[bx / (MATCH_STRIP / CT)] = 1;
    if (etid == 0 && s_dense && !s_heavy) { T.dense[bx / (MATCH_STRIP / CT)] = 1; if (T.n_dense) atomicAdd(T.n_dense, 1u); }
    if (s_heavy || s_dense) { if (etid == 0) { T.tchain[bx] = TileChain{0, 0, 0, 0}; T.step[bx] = 0; } return; }
    {   // Every visit of this tile a match over its whole distance, and the same distance d: the stretch repeats with a period that d is
        // the largest multiple of inside the window, every chain in it steps by d and keeps its phase -- the chains of neighbouring tiles
        // never join.  The tile says so (step = d): k_stretch_pred then places the true chain by arithmetic and the next look walks it.
        if (s_nstep == 0 || s_stepmin != s_stepmax) { if (etid == 0) T.step[bx] = 0; }   // (text: no such visit at all -- nothing to count)
        else {
            uint32_t cl = 0;
            for (int i = etid; i < C::DUMP_BYTES / 4; i += CTH) cl += (uint32_t)__builtin_popcount(s_claim[i]);
            for (int dd = 32; dd; dd >>= 1) cl += __shfl_down(cl, dd);
            if (elane == 0) s_part[ewv] = cl;
            __syncthreads();
            if (etid == 0) {
                uint32_t all = 0;
                for (int k = 0; k < CTH / 64; k++) all += s_part[k];
                T.step[bx] = s_nstep == all ? s_stepmin : 0u;
            }
        }
    }

    // ---- hand the claim bitmap to k_chain_tail (1 KB per tile): together with the keys it is all the in-tile parse needs
    {
        uint4 *dst = reinterpret_cast<uint4 *>(T.dump + (size_t)bx * C::DUMP_BYTES);
        for (int i = etid; i < C::DUMP_BYTES / 16; i += CTH) dst[i] = reinterpret_cast<const uint4 *>(s_claim)[i];
        if (etid == 0) T.tchain[bx] = TileChain{0, 0, 1u, kp_first};   // walked; entry and exit are k_chain_tail's to fill in
    }
    // ---- and for k_chain_serial, which follows the chain with one elane per tile: one record per claimed position, side by side in
    //      position order -- its key, where it is, and WHICH RECORD the chain goes on with (the rank of the landing position among
    //      the claims: a prefix popcount of the bitmap, here in LDS).  Through the position-indexed key array the elane would fetch
    //      every line of it, 4 GB per GiB, for a fifth of their contents, and each step would wait for its load; the records it can
    //      read a window ahead.  (The keys this block has just written come back from the L2.)
    if (T.ckeys) {
        constexpr int NWORDS = C::DUMP_BYTES / 4;
        uint32_t *s_pre = s_cur;                                          // (the bucket index has served)
        __syncthreads();
        // (the keys first -- a thread takes every CTH-th position, its loads are all asked for before anything else: one round trip to the
        //  L2 for the block, under the scan; a tile that turns out to have too many claims has read them for nothing)
        constexpr int PER = (NWORDS * 32 + CTH - 1) / CTH;
        const uint32_t *kb = a.keys + (t0 - CH);                              // (wave-uniform: a scalar base, 32-bit offsets)
        uint32_t kv[PER];
#pragma unroll
        for (int j = 0; j < PER; j++) {
            const uint32_t kp = (uint32_t)etid + (uint32_t)j * CTH;
            const bool on = kp < (uint32_t)NWORDS * 32u && ((s_claim[kp >> 5] >> (kp & 31)) & 1u);
            kv[j] = on ? kb[kp] : 0xFFFFFFFFu;                                // (no key has a length of 65535)
        }
        {   // claims before every word of the bitmap: the first NWORDS / 64 wavefronts scan 64 words each, then add what lies before them
            constexpr int NSW = (NWORDS + 63) / 64;
            const int w = etid;
            const uint32_t c = w < NWORDS ? (uint32_t)__builtin_popcount(s_claim[w]) : 0u;
            uint32_t incl = c;
            if (ewv < NSW) {
                for (int dd = 1; dd < 64; dd <<= 1) {                         // (bpermute addresses from the opaque lane index: see above)
                    const uint32_t y = (uint32_t)__builtin_amdgcn_ds_bpermute((elane - dd) << 2, (int)incl);
                    if (elane >= dd) incl += y;
                }
                if (elane == 63) s_part[ewv] = incl;
            }
            lds_barrier();
            if (ewv < NSW) {
                uint32_t before = 0;
                for (int k = 0; k < ewv; k++) before += s_part[k];
                if (w < NWORDS) s_pre[w] = before + incl - c;
                if (w == NWORDS - 1) s_pre[NWORDS] = before + incl;
            }
        }
        lds_barrier();
        const uint32_t total = s_pre[NWORDS];
        const bool fits = total <= (uint32_t)CT / 2 && (s_claim[kp_first >> 5] >> (kp_first & 31)) & 1u;   // (8 bytes a record in a region of 4 CT)
        if (fits) {
            uint2 *ck = reinterpret_cast<uint2 *>(T.ckeys + (size_t)bx * CT);
#pragma unroll
            for (int j = 0; j < PER; j++) {
                const uint32_t kp = (uint32_t)etid + (uint32_t)j * CTH, key = kv[j];
                if (key != 0xFFFFFFFFu) {
                    const uint32_t at = s_pre[kp >> 5] + (uint32_t)__builtin_popcount(s_claim[kp >> 5] & ((1u << (kp & 31)) - 1u));
                    const uint32_t kp2 = kp + max(1u, key >> 16);
                    uint32_t nr = CK_EXIT;                                // the chain leaves the tile
                    if (kp2 < kp_end) {
                        const uint32_t cw = s_claim[kp2 >> 5];
                        nr = (cw >> (kp2 & 31)) & 1u ? s_pre[kp2 >> 5] + (uint32_t)__builtin_popcount(cw & ((1u << (kp2 & 31)) - 1u)) : CK_BROKEN;
                    }
                    ck[at] = uint2{key, nr | (kp << 14)};
                }
            }
        }
        if (etid == 0)   // the number of records, and the record of the warm-up start
            T.ckn[bx] = fits ? total | ((s_pre[kp_first >> 5] + (uint32_t)__builtin_popcount(s_claim[kp_first >> 5] & ((1u << (kp_first & 31)) - 1u))) << 16) : NO_LIST;
    }
}

// The chain inside a tile, resolved from k_match_chain's claim bitmap and the keys instead of by the general parse
// (k_parse_exit .. k_parse_mark).  Every landing position of a walked chain was claimed and evaluated, so following
// i -> i + max(1, L) from the warm-up start never leaves the claimed set: the claimed positions (about one in five)
// are ranked by a prefix popcount, each gets the rank of its successor, and pointer doubling over that compact list
// flags what the warm-up start reaches (after round r: everything within < 2^(r+1) steps).  The block reports where
// the chain enters the tile and where it first lands beyond it; k_chain_verify accepts the whole stream iff each
// tile's exit IS the next tile's entry (tile 0 entering at position 0) -- then the flags written here are the true
// chain and the general parse is skipped; otherwise the general parse runs as before.
// CAP = claimed positions the block has room for.  Two instantiations run back to back: CAP = half of all positions
// (34 KB of LDS, four blocks per CU) resolves every ordinary tile; a tile with more claims than that -- a stretch of
// one- and two-byte steps -- is left to the second, full-size one (64 KB), which returns at once everywhere else.
template <class C, int CAP>
__global__ __launch_bounds__(512) void k_chain_tail(const uint8_t *__restrict__ dump, const uint32_t *__restrict__ keys, uint32_t E,
                                                    TileChain *__restrict__ tchain, uint32_t *__restrict__ flags, unsigned long long *__restrict__ tile_bytes,
                                                    uint32_t *__restrict__ ccnt) {
    constexpr int CT = C::CT, CH = C::CH, NKP = CH + CT, NW = NKP / 32, TT = 512;
    constexpr uint32_t OUT = 0xFFFFu;
    __shared__ __attribute__((aligned(16))) uint32_t s_claim[C::DUMP_BYTES / 4];
    __shared__ uint32_t s_pre[NW + 1];                                    // claimed positions before every bitmap word
    __shared__ uint16_t s_pos[CAP], s_j0[CAP], s_j1[CAP];                 // per rank: its position; the rank 2^r steps on (two buffers)
    __shared__ uint8_t s_bytes[CAP];                                      // per rank: bytes the position emits if it is on the chain
    __shared__ uint32_t s_on[NW], s_flag[CT / 32];                        // ranks on the chain; the same as position bits of the tile
    __shared__ uint32_t s_part[TT / 64];
    __shared__ uint32_t s_entry, s_exit, s_nout;
    __shared__ uint16_t s_out_rank[1024], s_out_nxt[1024];               // the few positions whose match reaches beyond the tile
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const TileChain tc = tchain[blockIdx.x];
    if (tc.walked != 1 || tc.exit != 0) return;                           // periodic / dense / heavy tile: the general parse will have to do it; or resolved by an earlier launch
    const long long t0 = (long long)blockIdx.x * CT;
    const uint32_t base = (uint32_t)(t0 - CH);
    const uint32_t npos = (uint32_t)min((long long)CT, (long long)E - t0), kp_end = CH + npos, kp_first = tc.pad;
    for (int i = tid; i < C::DUMP_BYTES / 16; i += TT) reinterpret_cast<uint4 *>(s_claim)[i] = reinterpret_cast<const uint4 *>(dump + (size_t)blockIdx.x * C::DUMP_BYTES)[i];
    for (int i = tid; i < NW; i += TT) s_on[i] = 0;
    for (int i = tid; i < CT / 32; i += TT) s_flag[i] = 0;
    if (tid == 0) { s_entry = 0xFFFFFFFFu; s_exit = 0xFFFFFFFFu; s_nout = 0; }
    __syncthreads();
    if (wv == 0) {                                                        // exclusive prefix of the word popcounts (NW = 264 words: one wavefront, 5 per lane)
        constexpr int PER = (NW + 63) / 64;
        uint32_t c[PER], sum = 0;
#pragma unroll
        for (int k = 0; k < PER; k++) { const int w = lane * PER + k; c[k] = w < NW ? (uint32_t)__builtin_popcount(s_claim[w]) : 0u; sum += c[k]; }
        uint32_t incl = sum;
        for (int dd = 1; dd < 64; dd <<= 1) { const uint32_t y = __shfl_up(incl, dd); if (lane >= dd) incl += y; }
        uint32_t run = incl - sum;
#pragma unroll
        for (int k = 0; k < PER; k++) { const int w = lane * PER + k; if (w < NW) s_pre[w] = run; run += c[k]; }
        if (lane == 63) s_pre[NW] = incl;
    }
    __syncthreads();
    const uint32_t n_cl = s_pre[NW];
    if (n_cl > (uint32_t)CAP) return;                                     // more claims than this instantiation holds: the full-size one takes the tile
    auto rank_of = [&](uint32_t p) { return s_pre[p >> 5] + (uint32_t)__builtin_popcount(s_claim[p >> 5] & ((1u << (p & 31)) - 1u)); };
    // the keys of the tile's NKP positions, 16 bytes per load (one position in five is claimed: scattered 4-byte loads
    // would touch every line anyway, one request each)
    for (uint32_t q = tid; q < (uint32_t)NKP / 4; q += TT) {
        const uint32_t p4 = 4 * q, nib = (s_claim[p4 >> 5] >> (p4 & 31)) & 15u;
        if (!nib) continue;
        const long long g = t0 - CH + (long long)p4;                      // stream position of p4 (negative only in tile 0's unused warm-up zone)
        uint32_t kk[4] = {0, 0, 0, 0};
        if (g >= 0 && g + 4 <= (long long)E) { const uint4 v = *reinterpret_cast<const uint4 *>(keys + g); kk[0] = v.x; kk[1] = v.y; kk[2] = v.z; kk[3] = v.w; }
        else for (int u = 0; u < 4; u++) if (g + u >= 0 && g + u < (long long)E) kk[u] = keys[g + u];
#pragma unroll
        for (int u = 0; u < 4; u++) {
            if (!((nib >> u) & 1)) continue;
            const uint32_t p = p4 + u, key = kk[u], L = key >> 16, nxt = p + max(1u, L), r = rank_of(p);
            const uint32_t el = enc_len(key & 0xFFFFu, L);
            s_pos[r] = (uint16_t)p;
            s_bytes[r] = (uint8_t)(L == 0 ? 1u : (el < L ? el : L));      // token only if strictly shorter than the bytes it stands for (lzss.go:143)
            uint32_t j = OUT;
            if (nxt < kp_end) j = ((s_claim[nxt >> 5] >> (nxt & 31)) & 1) ? rank_of(nxt) : OUT;   // (an unclaimed landing cannot happen in a finished walk)
            else { const uint32_t k = atomicAdd(&s_nout, 1u); if (k < 1024) { s_out_rank[k] = (uint16_t)r; s_out_nxt[k] = (uint16_t)nxt; } }
            s_j0[r] = (uint16_t)j;
        }
    }
    if (tid == 0) { const uint32_t r0 = rank_of(kp_first); s_on[r0 >> 5] = 1u << (r0 & 31); }
    __syncthreads();
    for (int round = 0; round < 16; round++) {
        const uint16_t *jc = (round & 1) ? s_j1 : s_j0;
        uint16_t *jn = (round & 1) ? s_j0 : s_j1;
        bool any = false;
        for (uint32_t r = tid; r < n_cl; r += TT) {
            const uint32_t j = jc[r];
            if (j != OUT) {
                if (((s_on[r >> 5] >> (r & 31)) & 1) && !((s_on[j >> 5] >> (j & 31)) & 1)) { atomicOr(&s_on[j >> 5], 1u << (j & 31)); any = true; }
                jn[r] = jc[j];
            } else jn[r] = (uint16_t)OUT;
        }
        if (!__syncthreads_or(any)) break;                                // nothing new at distance [2^r, 2^(r+1)): the chain is shorter than that
    }
    uint32_t bytes = 0;
    for (uint32_t r = tid; r < n_cl; r += TT) {
        if (!((s_on[r >> 5] >> (r & 31)) & 1)) continue;
        const uint32_t p = s_pos[r];
        if (p >= (uint32_t)CH) { bytes += s_bytes[r]; atomicMin(&s_entry, p); atomicOr(&s_flag[(p - CH) >> 5], 1u << ((p - CH) & 31)); }
    }
    for (uint32_t k = tid; k < min(s_nout, 1024u); k += TT) {             // exactly one chain position steps out of the tile
        const uint32_t r = s_out_rank[k];
        if ((s_on[r >> 5] >> (r & 31)) & 1) s_exit = s_out_nxt[k];
    }
    if (s_nout > 1024u && tid == 0) s_exit = 0xFFFFFFFFu;                 // (more candidates than the list holds: leave it to the general parse)
    for (int dd = 32; dd; dd >>= 1) bytes += __shfl_down(bytes, dd);
    if (lane == 0) s_part[wv] = bytes;
    __syncthreads();
    for (int w = tid; w < CT / 32; w += TT) flags[(size_t)blockIdx.x * (CT / 32) + w] = s_flag[w];
    if (tid == 0) {
        unsigned long long tb = 0;
        for (int k = 0; k < TT / 64; k++) tb += s_part[k];
        tile_bytes[blockIdx.x] = tb;
        if (ccnt) ccnt[blockIdx.x] = NO_LIST;                             // (a list k_chain_serial may have left for an earlier walk of this tile is stale)
        // a tile the chain jumps over entirely cannot happen (L <= W <= 4096 < CT); a chain that ends inside the warm-up zone can (last tile)
        tchain[blockIdx.x] = TileChain{s_entry == 0xFFFFFFFFu ? 0xFFFFFFFFu : base + s_entry, s_exit == 0xFFFFFFFFu ? 0xFFFFFFFFu : base + s_exit, 1u, 0u};
    }
}

// The same in-tile parse, serially: ONE LANE per walked tile follows i -> i + max(1, L) from the warm-up start through the keys
// in memory (every position it lands on has one: the walk claimed and evaluated it), flags what lies inside the tile, adds up the
// output bytes and reports entry and exit.  About 1 560 dependent loads per tile, most of them L2 hits (three chain positions share
// a 64-byte line), and every tile of the stream has its own lane -- no LDS, no ranks, no doubling rounds, and no block left waiting
// on the scalar unit, which is what bounded k_chain_tail (0.76 of its roofline: loop and mask bookkeeping of sixteen rounds).
// (Four lanes per tile, one per quarter with its own warm-up, measured slower -- 3.3 vs 2.95 ms: the kernel is bound by cache-line
//  requests, not by the length of a lane's chain.)
// The lane also leaves the keys it met inside the tile, in chain order, as a compact list (clist[tile * CT ..], ccnt[tile] entries): what
// k_tok_emit needs of the 4 bytes per position of the key array is these -- one position in six on text -- and their positions follow
// from the tile's entry by adding up max(1, L).
#define RSN_SERIAL_WIN 32                                                 // records a lane stages at a time
#define RSN_SERIAL_TILES 64                                               // tiles per wavefront (fewer: more wavefronts in flight, emptier)
template <class C>
__global__ __launch_bounds__(64) void k_chain_serial(const uint32_t *__restrict__ keys, uint32_t E, uint32_t n_tiles, TileChain *__restrict__ tchain,
                                                     uint32_t *__restrict__ flags, unsigned long long *__restrict__ tile_bytes,
                                                     uint32_t *clist, uint32_t *__restrict__ ccnt, const uint32_t *__restrict__ ckn) {
    constexpr uint32_t CT = C::CT, CH = C::CH, WORDS = CT / 32;
    const uint32_t k = blockIdx.x * RSN_SERIAL_TILES + threadIdx.x;
    const int lane = threadIdx.x;
    TileChain tc = {0, 0, 0, 0};
    const bool in_range = k < n_tiles && threadIdx.x < RSN_SERIAL_TILES;
    if (in_range) tc = tchain[k];
    bool todo = in_range && tc.walked == 1 && tc.exit == 0;            // (not: periodic / dense / heavy tile, or resolved by an earlier launch)
    const unsigned long long t0 = (unsigned long long)k * CT, t1 = min(t0 + CT, (unsigned long long)E);
    // ---- from k_match_chain's records (see its epilogue): record r holds the chain's key there, the position and the next record.
    // The 64 lanes are on 64 tiles, and a load that one of them waits for stalls them all: so all of them stage a WINDOW of records
    // in LDS at the same time (one wait per window), then step through their windows until each has left its own.  The chain's own
    // keys go back into the same region from its start, 4 bytes for every 8 read: the write cursor stays behind the read cursor.
    const uint32_t cn = (todo && ckn && clist) ? ckn[k] : NO_LIST;
    bool active = cn != NO_LIST;
    if (__any(active)) {
        constexpr uint32_t WIN = RSN_SERIAL_WIN;
        __shared__ uint2 s_win[WIN][64];
        __shared__ uint32_t s_out[8][64];
        const uint4 *cg = reinterpret_cast<const uint4 *>(clist + (size_t)(k < n_tiles ? k : 0) * CT);   // two records a load; the region holds CT / 2
        uint32_t *cl = clist + (size_t)(k < n_tiles ? k : 0) * CT;
        uint32_t *fw = flags + (size_t)(k < n_tiles ? k : 0) * WORDS;
        uint32_t r = active ? cn >> 16 : 0u, rbase = 0, entry = 0xFFFFFFFFu, exit_kp = 0;
        uint32_t wi = 0, word = 0, n_on = 0;
        unsigned long long bytes = 0;
        bool ok = true;
        while (__any(active)) {
            if (active) {
                rbase = r & ~1u;
                uint4 ld[WIN / 2];
#pragma unroll
                for (uint32_t j = 0; j < WIN / 2; j++) ld[j] = cg[min(rbase / 2 + j, CT / 4 - 1)];
#pragma unroll
                for (uint32_t j = 0; j < WIN / 2; j++) { s_win[2 * j][lane] = uint2{ld[j].x, ld[j].y}; s_win[2 * j + 1][lane] = uint2{ld[j].z, ld[j].w}; }
            }
            while (__any(active && r - rbase < WIN)) {
                if (active && r - rbase < WIN) {
                    const uint2 e = s_win[r - rbase][lane];
                    const uint32_t key = e.x, kp = (e.y >> 14) & 0x3FFFu, nr = e.y & 0x3FFFu;
                    if (kp >= CH) {
                        if (entry == 0xFFFFFFFFu) entry = kp;
                        const uint32_t q = kp - CH;
                        while (wi < (q >> 5)) { fw[wi++] = word; word = 0; }  // (every word of the tile is written exactly once, in order)
                        word |= 1u << (q & 31);
                        // (eight keys at a time, two 16-byte stores -- see below; they wait in LDS: picking one of eight registers by a
                        //  lane's own count is a tree of branches that every lane walks)
                        s_out[n_on & 7u][lane] = key;
                        n_on++;
                        if ((n_on & 7u) == 0) {
                            *reinterpret_cast<uint4 *>(cl + n_on - 8) = uint4{s_out[0][lane], s_out[1][lane], s_out[2][lane], s_out[3][lane]};
                            *reinterpret_cast<uint4 *>(cl + n_on - 4) = uint4{s_out[4][lane], s_out[5][lane], s_out[6][lane], s_out[7][lane]};
                        }
                        const uint32_t L = key >> 16, el = enc_len(key & 0xFFFFu, L);
                        bytes += L == 0 ? 1u : (el < L ? el : L);             // lzss.go:143
                    }
                    if (nr >= CK_EXIT) { active = false; ok = nr == CK_EXIT; exit_kp = kp + max(1u, key >> 16); }
                    else r = nr;
                }
            }
        }
        if (cn != NO_LIST) {
            if (ok) {
                while (wi < WORDS) { fw[wi++] = word; word = 0; }
                tile_bytes[k] = bytes;
                if (n_on & 7u) {
                    *reinterpret_cast<uint4 *>(cl + (n_on & ~7u)) = uint4{s_out[0][lane], s_out[1][lane], s_out[2][lane], s_out[3][lane]};
                    if ((n_on & 7u) > 4) *reinterpret_cast<uint4 *>(cl + (n_on & ~7u) + 4) = uint4{s_out[4][lane], s_out[5][lane], s_out[6][lane], s_out[7][lane]};
                }
                if (ccnt) ccnt[k] = n_on;
                if (entry == 0xFFFFFFFFu) entry = exit_kp;                    // (the chain jumps over what is left of the stream: as below, the landing position)
                tchain[k] = TileChain{(uint32_t)(t0 - CH) + entry, (uint32_t)(t0 - CH) + exit_kp, 1u, 0u};
                todo = false;
            } else {                                                      // (cannot happen for a tile that was walked to the end; the keys decide)
                if (ccnt) ccnt[k] = NO_LIST;
                clist = nullptr; ccnt = nullptr;                          // the region no longer holds what the list form would read: flags only
            }
        }
    }
    if (!todo) return;
    unsigned long long p = t0 - CH + tc.pad;                              // the warm-up start (position 0 in tile 0)
    // keys come in aligned groups of four (one 16-byte load); a short step often stays inside the group it has
    uint4 grp = {0, 0, 0, 0};
    unsigned long long have = ~0ull;
    auto key_at = [&](unsigned long long q) {
        if ((q >> 2) != have) {
            have = q >> 2;
            if (4 * have + 4 <= (unsigned long long)E) grp = *reinterpret_cast<const uint4 *>(keys + 4 * have);
            else { grp.x = keys[4 * have]; grp.y = 4 * have + 1 < E ? keys[4 * have + 1] : 0u; grp.z = 4 * have + 2 < E ? keys[4 * have + 2] : 0u; grp.w = 0u; }
        }
        const uint32_t sel = (uint32_t)q & 3u;
        return sel == 0 ? grp.x : sel == 1 ? grp.y : sel == 2 ? grp.z : grp.w;
    };
    while (p < t0) p += max(1u, key_at(p) >> 16);                         // the warm-up chain: merged with the true one long before the tile
    const unsigned long long entry = p;
    uint32_t *fw = flags + (size_t)k * WORDS;
    uint32_t *cl = clist ? clist + (size_t)k * CT : nullptr;
    uint32_t wi = 0, word = 0, n_on = 0;
    uint4 pend = {0, 0, 0, 0}, pend2 = {0, 0, 0, 0};
    unsigned long long bytes = 0;
    while (p < t1) {
        const uint32_t key = key_at(p), L = key >> 16, r = (uint32_t)(p - t0);
        while (wi < (r >> 5)) { fw[wi++] = word; word = 0; }              // (every word of the tile is written exactly once, in order)
        word |= 1u << (r & 31);
        // (eight keys at a time, two 16-byte stores: a 4-byte store per step to 64 lanes' 64 different lines made this kernel three times
        //  slower -- 5.7 against 1.9 ms per GiB of text; four at a time 2.6)
        const uint32_t sl = n_on & 7u;
        if (sl == 0) pend.x = key; else if (sl == 1) pend.y = key; else if (sl == 2) pend.z = key; else if (sl == 3) pend.w = key;
        else if (sl == 4) pend2.x = key; else if (sl == 5) pend2.y = key; else if (sl == 6) pend2.z = key; else pend2.w = key;
        n_on++;
        if (cl && sl == 7) { *reinterpret_cast<uint4 *>(cl + n_on - 8) = pend; *reinterpret_cast<uint4 *>(cl + n_on - 4) = pend2; }
        const uint32_t el = enc_len(key & 0xFFFFu, L);
        bytes += L == 0 ? 1u : (el < L ? el : L);                         // token only if strictly shorter than the bytes it stands for (lzss.go:143)
        p += max(1u, L);
    }
    while (wi < WORDS) { fw[wi++] = word; word = 0; }
    tile_bytes[k] = bytes;
    if (cl && (n_on & 7u)) {                                            // (the list has room for CT entries and n_on <= CT: a whole group of eight always fits)
        *reinterpret_cast<uint4 *>(cl + (n_on & ~7u)) = pend;
        if ((n_on & 7u) > 4) *reinterpret_cast<uint4 *>(cl + (n_on & ~7u) + 4) = pend2;
    }
    if (ccnt) ccnt[k] = n_on;
    tchain[k] = TileChain{(uint32_t)entry, (uint32_t)p, 1u, 0u};
}


// W-periodic tiles (config 3 is nothing else) have a key at every position -- L = min(W, E-p) at distance W -- and
// no chain of their own: chains of different phase never merge in periodic data, the phase is handed on from tile
// to tile.  k_prev_walked finds, for every tile, the nearest tile before it whose chain was really walked (an
// inclusive max-scan of "index if walked"); k_chain_periodic then places the chain of a periodic tile by
// arithmetic: from that tile's exit x the chain steps by W through the periodic stretch, so it enters this tile
// at x + W * ceil((t0 - x) / W) -- valid because every tile in between is periodic too (checked) -- and the flags,
// the output bytes and the tile's entry / exit follow.  k_chain_verify then checks the joints as for any tile.
__global__ __launch_bounds__(1024) void k_prev_walked(const TileChain *__restrict__ tc, uint32_t n_tiles, uint32_t *__restrict__ prev, uint32_t *__restrict__ part,
                                                      const uint32_t *__restrict__ vals = nullptr) {   // vals: scan these (index + 1 or 0 per tile) instead of "walked"
    // one block per PREV_BLK tiles: the inclusive maximum inside the block; k_prev_fix brings in the blocks before
    constexpr int IT = PREV_BLK / 1024;
    __shared__ uint32_t wmax[16];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const uint32_t i0 = blockIdx.x * PREV_BLK + tid * IT;
    uint32_t x[IT], loc = 0;
#pragma unroll
    for (int k = 0; k < IT; k++) { x[k] = i0 + k >= n_tiles ? 0u : vals ? vals[i0 + k] : (tc[i0 + k].walked == 1 ? i0 + k + 1 : 0u); loc = max(loc, x[k]); }   // 0 = none yet; else tile index + 1
    uint32_t sfx = loc;
    for (int d = 1; d < 64; d <<= 1) { const uint32_t y = __shfl_up(sfx, d); if (lane >= d) sfx = max(sfx, y); }
    if (lane == 63) wmax[wv] = sfx;
    __syncthreads();
    uint32_t pre = 0, all = 0;
    for (int k = 0; k < 16; k++) { if (k < wv) pre = max(pre, wmax[k]); all = max(all, wmax[k]); }
    uint32_t run = max(pre, (uint32_t)__shfl_up(sfx, 1));
    if (lane == 0) run = pre;
#pragma unroll
    for (int k = 0; k < IT; k++) { run = max(run, x[k]); if (i0 + k < n_tiles) prev[i0 + k] = run; }   // inclusive: a walked tile names itself
    if (tid == 0) part[blockIdx.x] = all;
}

__global__ __launch_bounds__(1024) void k_prev_fix(uint32_t *__restrict__ prev, uint32_t n_tiles, const uint32_t *__restrict__ part) {
    __shared__ uint32_t red[16];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const uint32_t b = blockIdx.x;
    uint32_t m = 0;
    for (uint32_t k = tid; k < b; k += 1024) m = max(m, part[k]);
    for (int d = 32; d; d >>= 1) m = max(m, (uint32_t)__shfl_down(m, d));
    if (lane == 0) red[wv] = m;
    __syncthreads();
    uint32_t pre = 0;
    for (int k = 0; k < 16; k++) pre = max(pre, red[k]);
    if (pre) for (uint32_t i = b * PREV_BLK + tid; i < min(n_tiles, (b + 1) * PREV_BLK); i += 1024) prev[i] = max(prev[i], pre);
}

// Only when the general parse has to take over: the chain walk wrote keys where a chain landed and nothing else
// (the key array is not cleared beforehand -- that alone was a 4-byte store per position), so every position
// without an exact key gets KEY_UNKNOWN here.  A position has one iff its tile was W-periodic, or its tile's
// finished walk claimed it, or it lies in the next tile's warm-up zone and that tile's finished walk claimed it.
// (A tile that gave up keeps none: its strip is redone for all positions anyway.)
template <class C>
__global__ __launch_bounds__(256) void k_chain_unknown(const uint8_t *__restrict__ dump, const TileChain *__restrict__ tchain, uint32_t n_tiles, uint32_t E,
                                                       uint32_t W, uint32_t *__restrict__ keys) {
    constexpr int CT = C::CT, CH = C::CH;
    static_assert(CH % 32 == 0 && CT % 32 == 0, "bitmap words line up with the tile");
    __shared__ uint32_t s_known[CT / 32];
    const int tid = threadIdx.x;
    const uint32_t k = blockIdx.x;
    const uint32_t w_own = tchain[k].walked, w_nxt = k + 1 < n_tiles ? tchain[k + 1].walked : 0u;
    const uint32_t *own = reinterpret_cast<const uint32_t *>(dump + (size_t)k * C::DUMP_BYTES);
    const uint32_t *nxt = reinterpret_cast<const uint32_t *>(dump + (size_t)(k + 1) * C::DUMP_BYTES);
    const size_t t0 = (size_t)k * CT;
    const uint32_t npos = (uint32_t)min((size_t)CT, (size_t)E - t0);
    if (w_own == 2) {                                                     // W-periodic tile: its keys are known but were never stored
        for (uint32_t i = tid; i < npos; i += 256) keys[t0 + i] = (min(W, E - (uint32_t)(t0 + i)) << 16) | W;
        return;
    }
    for (int w = tid; w < CT / 32; w += 256) {
        uint32_t m = w_own == 1 ? own[CH / 32 + w] : 0u;
        if (w >= (CT - CH) / 32 && w_nxt == 1) m |= nxt[w - (CT - CH) / 32];
        s_known[w] = m;
    }
    __syncthreads();
    for (uint32_t i = tid; i < npos; i += 256)
        if (!((s_known[i >> 5] >> (i & 31)) & 1)) keys[t0 + i] = KEY_UNKNOWN;
}

__global__ __launch_bounds__(256) void k_chain_periodic(TileChain *__restrict__ tc, const uint32_t *__restrict__ prev, uint32_t n_tiles, uint32_t E, uint32_t W,
                                                        uint32_t tile, uint32_t *__restrict__ flags, unsigned long long *__restrict__ tile_bytes) {
    const uint32_t k = blockIdx.x;
    if (tc[k].walked != 2) return;
    const uint32_t words = tile / 32;
    for (uint32_t w = threadIdx.x; w < words; w += blockDim.x) flags[(size_t)k * words + w] = 0;
    __syncthreads();
    if (threadIdx.x) return;
    const uint32_t pw = prev[k];
    if (pw == 0) return;                                                // nothing walked before it: stays unresolved, the general parse decides
    const TileChain src = tc[pw - 1];
    if (src.exit == 0 || src.exit == 0xFFFFFFFFu) return;
    const unsigned long long t0 = (unsigned long long)k * tile, t1 = min(t0 + tile, (unsigned long long)E);
    unsigned long long x = src.exit;                                    // first chain position beyond the walked tile: inside the periodic stretch
    if (x < (unsigned long long)pw * tile || x >= t1 + W) return;
    if (x < t0) x += (t0 - x + W - 1) / W * W;                          // whole steps of W through the periodic tiles in between
    if (x >= t1) {                                                      // (cannot happen: W <= 4096 < tile) the chain jumps over this tile
        tc[k] = TileChain{0xFFFFFFFFu, 0xFFFFFFFFu, 2u, 0};
        return;
    }
    const unsigned long long entry = x;
    unsigned long long bytes = 0;
    while (x < t1) {
        const uint32_t L = (uint32_t)min((unsigned long long)W, (unsigned long long)E - x), el = enc_len(W, L);
        bytes += el < L ? el : L;                                       // lzss.go:143
        const uint32_t r = (uint32_t)(x - t0);
        flags[(size_t)k * words + (r >> 5)] |= 1u << (r & 31);
        x += L;                                                         // L >= 1 here (x < E)
    }
    tile_bytes[k] = bytes;
    tc[k] = TileChain{(uint32_t)entry, (uint32_t)x, 2u, 1u};            // pad = 1: resolved
}

// Whole-distance stretches.  Where a stream repeats with a period p that does not divide the window, every position's match is the
// farthest multiple of p inside the window, d, over d bytes; every chain steps by d and keeps its phase, so the chains that
// neighbouring tiles walked from their own warm-up starts never join (k_tile_periodic's W-periodic tiles are the case d = W and have
// their own arithmetic, k_chain_periodic).  k_match_chain reports such a tile (step[k] = d).  Here the true chain is placed through a
// run of them: from the exit X of the tile just before the run, the chain enters the run's tile k at X + d * ceil((t0 - X) / d).
// That is a prediction, not a proof: the next look walks the tile from there and k_chain_verify judges the joints as always; if X
// itself changes in that look (the tile before the run was mended), the following look places the run again.
__global__ void k_stretch_flags(const uint32_t *__restrict__ step, uint32_t n_tiles, uint32_t *__restrict__ brk) {
    const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n_tiles) return;
    const uint32_t st = step[k];
    brk[k] = (k == 0 || st == 0 || st != step[k - 1]) ? k + 1 : 0u;       // a run of equal steps begins here (index + 1, for k_prev_walked's inclusive max-scan)
}
__global__ void k_stretch_pred(const TileChain *__restrict__ tc, const uint32_t *__restrict__ step, const uint32_t *__restrict__ head_brk,
                               uint32_t n_tiles, uint32_t tile, uint32_t E, uint32_t *__restrict__ pred) {
    const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n_tiles) return;
    uint32_t out = 0xFFFFFFFFu;
    const uint32_t st = step[k], hb = head_brk[k];
    if (st != 0 && tc[k].walked == 1 && hb >= 2) {                        // the run of tiles that step by st begins at tile hb - 1; the tile before it hands the chain over
        const uint32_t j = hb - 2;
        const TileChain h = tc[j];
        const uint32_t X = h.exit;
        if ((h.walked == 1 || (h.walked == 2 && h.pad == 1)) && X != 0xFFFFFFFFu && X < E && (unsigned long long)X >= (unsigned long long)(j + 1) * tile &&
            (unsigned long long)X < (unsigned long long)(j + 2) * tile) {
            const unsigned long long t0 = (unsigned long long)k * tile;
            unsigned long long x = X;
            if (x < t0) x += (unsigned long long)st * ((t0 - x + st - 1) / st);
            if (x < t0 + tile && x < E) out = (uint32_t)x;
        }
    }
    pred[k] = out;
}

// Accepts the per-tile chains of k_match_chain as THE chain iff they join up: tile 0 enters at position 0 and every
// tile's exit is the next tile's entry (the last tile's exit is at or beyond the end of the stream).
// the records of a section's halo tiles: resolved, nothing flagged, nothing emitted, the chain handed from tile to tile
template <class C>
__global__ __launch_bounds__(256) void k_halo_init(TileChain *__restrict__ tchain, uint32_t *__restrict__ flags, unsigned long long *__restrict__ tile_bytes,
                                                   uint8_t *__restrict__ dump, uint32_t *__restrict__ step, uint32_t tile) {
    const uint32_t k = blockIdx.x;
    for (uint32_t i = threadIdx.x; i < tile / 32; i += 256) flags[(size_t)k * (tile / 32) + i] = 0;
    for (uint32_t i = threadIdx.x; i < (uint32_t)C::DUMP_BYTES / 4; i += 256) reinterpret_cast<uint32_t *>(dump + (size_t)k * C::DUMP_BYTES)[i] = 0;
    if (threadIdx.x == 0) { tchain[k] = TileChain{k * tile, (k + 1) * tile, 1u, 0u}; tile_bytes[k] = 0; step[k] = 0; }
}

__global__ void k_sample_tiles(uint32_t *__restrict__ list, uint32_t n, uint32_t n_tiles) {   // n tiles spread evenly over the stream (not tile 0: it has no window)
    if (threadIdx.x < n) list[threadIdx.x] = (uint32_t)(((unsigned long long)threadIdx.x * n_tiles + n_tiles / 2) / n);
}

__global__ void k_chain_verify(const TileChain *__restrict__ tc, uint32_t n_tiles, uint32_t E, uint32_t tile, uint32_t *__restrict__ bad,
                               uint32_t *__restrict__ redo_list, uint32_t redo_cap, uint32_t *__restrict__ redo_start, const uint32_t *__restrict__ pred, uint32_t list_gave) {
    const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    const bool live = k < n_tiles;
    const TileChain c = live ? tc[k] : TileChain{0, 0, 1u, 0};
    bool ok = c.walked == 1 || (c.walked == 2 && c.pad == 1);           // walked, or periodic and placed by k_chain_periodic
    if (ok && live) {
        // (A last match that runs to the very end of the stream can jump over the final, partial tile: that tile then holds no chain
        //  position at all -- its parse reports no entry, or one beyond the stream, and flagged nothing -- and is in order as it is.)
        const bool over = c.exit != 0xFFFFFFFFu && c.exit >= E;           // this tile's chain leaves the stream
        const bool empty = c.walked == 1 && (c.entry == 0xFFFFFFFFu || c.entry >= E);
        if (k == 0) ok = c.entry == 0;
        if (k + 1 < n_tiles) {
            const TileChain nx = tc[k + 1];
            if (over) ok = ok && k + 2 == n_tiles && nx.walked == 1 && (nx.entry == 0xFFFFFFFFu || nx.entry >= E) && c.entry != 0xFFFFFFFFu;
            else ok = ok && (nx.walked == 1 || (nx.walked == 2 && nx.pad == 1)) && c.exit == nx.entry && (unsigned long long)c.exit < (unsigned long long)(k + 2) * tile &&
                      (unsigned long long)c.exit >= (unsigned long long)(k + 1) * tile && c.entry != 0xFFFFFFFFu;
        } else if (empty && k > 0) {
            const TileChain q = tc[k - 1];
            ok = (q.walked == 1 || (q.walked == 2 && q.pad == 1)) && q.exit != 0xFFFFFFFFu && q.exit >= E;
        } else ok = ok && over && c.entry != 0xFFFFFFFFu;
    }
    // bad[0]: tiles that gave up (dense / heavy), bad[1]: chains that do not join, bad[2]: periodic tiles that could not be placed,
    // bad[3]: length of the list for a second look (one atomic per wavefront and class: config 3's first look fails 131071 times)
    const uint32_t cls = c.walked == 1 ? 1u : (c.walked == 2 ? 2u : 0u);
#pragma unroll
    for (uint32_t q = 0; q < 3; q++) {
        const unsigned long long m = __ballot(live && !ok && cls == q);
        if (m && (threadIdx.x & 63) == 0) atomicAdd(&bad[q], (uint32_t)__builtin_popcountll(m));
    }
    // The list (every tile decides for itself): a tile that gave up is walked again without the density test; a walked tile whose entry
    // is not the exit of the (resolved) tile before it -- its warm-up chain had not merged with the true chain yet -- is walked again
    // from that exit; a tile of a whole-distance stretch (pred, see k_stretch_pred) from the arithmetic entry -- or not at all if it
    // is on that phase already, whatever the tile before it says.  bad[4] counts the entries of the last kind.
    const bool gave = live && cls == 0 && list_gave;                       // (not listed when too many of them gave up for a look: those are the bucket search's)
    uint32_t start = 0xFFFFFFFFu;
    bool predicted = false;
    if (live && c.walked == 1 && k > 0) {
        const TileChain q = tc[k - 1];
        if ((q.walked == 1 || q.pad == 1) && q.entry != 0xFFFFFFFFu && q.exit != c.entry && (unsigned long long)q.exit >= (unsigned long long)k * tile &&
            (unsigned long long)q.exit < (unsigned long long)(k + 1) * tile && q.exit < E) start = q.exit;
        const uint32_t pr = pred ? pred[k] : 0xFFFFFFFFu;
        if (pr != 0xFFFFFFFFu) { start = pr != c.entry ? pr : 0xFFFFFFFFu; predicted = start != 0xFFFFFFFFu; }
    }
    const bool fix = start != 0xFFFFFFFFu;
    const unsigned long long mp = __ballot(predicted);
    if (mp && (threadIdx.x & 63) == 0) atomicAdd(&bad[4], (uint32_t)__builtin_popcountll(mp));
    const unsigned long long ml = __ballot(gave || fix);
    if (ml) {
        uint32_t at = 0;
        if ((threadIdx.x & 63) == 0) at = atomicAdd(&bad[3], (uint32_t)__builtin_popcountll(ml));
        at = (uint32_t)__builtin_amdgcn_readfirstlane((int)at) + (uint32_t)__builtin_popcountll(ml & ((1ull << (threadIdx.x & 63)) - 1ull));
        if ((gave || fix) && at < redo_cap) { redo_list[at] = gave ? k : (k | 0x80000000u | (predicted ? 0x40000000u : 0u)); redo_start[at] = start; }
    }
}

// ------------------------------------------------------------------ E3: greedy chain
__device__ __forceinline__ uint32_t enc_len(uint32_t off, uint32_t len) {   // len("<off,len>"), lzss.go:318-320
    auto digits = [](uint32_t v) { return v < 10 ? 1u : v < 100 ? 2u : v < 1000 ? 3u : v < 10000 ? 4u : 5u; };
    return 3 + digits(off) + digits(len);
}

// next(i) = i + max(1, L_i) (lzss.go:139-142); exit_rel[i] = overshoot past the tile end of the chain started at i
__global__ __launch_bounds__(LB) void k_parse_exit(const uint32_t *__restrict__ keys, uint32_t E, uint16_t *__restrict__ exit_rel) {
    __shared__ uint16_t nxt[PT];
    const uint32_t base = blockIdx.x * PT;
    for (int i = threadIdx.x; i < PT; i += LB) {
        const uint32_t p = base + i;
        uint32_t L = p < E ? keys[p] >> 16 : 1;
        if (L == 0 || L == 0xFFFFu) L = 1;                            // literal; or not evaluated (k_match_chain): k_parse_mark reports it if the chain gets there
        nxt[i] = (uint16_t)(i + L);                                   // < PT + MAX_WINDOW <= 65535
    }
    __syncthreads();
    for (int round = 0; round < 13; round++) {                        // 2^13 = PT: every chain has left the tile by then
        if ((round & 3) == 1) {                                       // every fourth round also asks whether anything is still inside:
            bool more = false;                                        // long matches leave the tile in a round or two
            for (int i = threadIdx.x; i < PT; i += LB) {
                const uint32_t j = nxt[i];
                if (j < PT) { const uint32_t k = nxt[j]; nxt[i] = (uint16_t)k; more = more || k < PT; }
            }
            if (!__syncthreads_or(more)) break;
        } else {
            for (int i = threadIdx.x; i < PT; i += LB) {
                const uint32_t j = nxt[i];
                if (j < PT) nxt[i] = nxt[j];
            }
            __syncthreads();
        }
    }
    for (int i = threadIdx.x; i < PT; i += LB) if (base + i < E) exit_rel[base + i] = (uint16_t)(nxt[i] - PT);
}

// Where does the chain that starts at position 0 enter each tile?  One dependent load per
// tile if done naively; instead tiles are grouped SUPER at a time:
//   k_parse_super   for EVERY possible entry into a group's first tile, where the chain leaves
//                   the group (SUPER dependent loads per entry, all entries in parallel)
//   k_parse_chain   serial over groups only (n_tiles / SUPER dependent loads)
//   k_parse_fill    per group, walk its SUPER tiles from the now-known entry
constexpr int SUPER = 64;
static_assert(MAX_WINDOW <= PT, "the chain must enter every group through its first tile");
static_assert(HALO_TILES == SUPER, "a section's chain starts at a group's first position");

__device__ __forceinline__ unsigned long long chain_step(const uint16_t *__restrict__ exit_rel, unsigned long long pos, uint32_t t) {
    // pos is a global position inside tile t (or beyond it): returns the first chain position >= end of tile t
    const unsigned long long hi = (unsigned long long)(t + 1) * PT;
    return pos >= hi ? pos : hi + exit_rel[pos];
}

__global__ __launch_bounds__(LB) void k_parse_super(const uint16_t *__restrict__ exit_rel, uint32_t n_tiles, uint32_t E,
                                                    uint32_t *__restrict__ super_exit) {
    const uint32_t g = blockIdx.x, t0 = g * SUPER, t1 = min(t0 + SUPER, n_tiles);
    for (uint32_t e = threadIdx.x; e < PT; e += LB) {
        unsigned long long pos = (unsigned long long)t0 * PT + e;
        if (pos < E) for (uint32_t t = t0; t < t1; t++) pos = chain_step(exit_rel, pos, t);
        super_exit[(size_t)g * PT + e] = (uint32_t)(pos - (unsigned long long)t1 * PT);   // overshoot past the group
    }
}

__global__ void k_parse_chain(const uint32_t *__restrict__ super_exit, uint32_t n_groups, uint32_t n_tiles, unsigned long long *__restrict__ group_entry,
                              unsigned long long start) {           // start: 0, or the first position behind a section's halo (a group's first)
    if (threadIdx.x || blockIdx.x) return;
    unsigned long long pos = start;
    for (uint32_t g = 0; g < n_groups; g++) {
        group_entry[g] = pos;                                       // global position where the chain enters (or jumps over) group g
        const unsigned long long lo = (unsigned long long)g * SUPER * PT;
        const unsigned long long hi = (unsigned long long)min((g + 1) * SUPER, n_tiles) * PT;
        if (pos >= hi) continue;                                    // a halo group: the chain begins behind it
        // a match is at most W <= PT long, so the chain always lands inside the group's FIRST tile
        pos = hi + super_exit[(size_t)g * PT + (pos - lo)];
    }
}

__global__ void k_parse_fill(const uint16_t *__restrict__ exit_rel, const unsigned long long *__restrict__ group_entry, uint32_t n_groups,
                             uint32_t n_tiles, uint32_t *__restrict__ entry) {
    const uint32_t g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= n_groups) return;
    unsigned long long pos = group_entry[g];
    const uint32_t t0 = g * SUPER, t1 = min(t0 + SUPER, n_tiles);
    for (uint32_t t = t0; t < t1; t++) {
        const unsigned long long lo = (unsigned long long)t * PT;
        entry[t] = pos >= lo + PT ? NO_ENTRY : (uint32_t)(pos - lo);
        pos = chain_step(exit_rel, pos, t);
    }
}

// The distance of a match whose length is already known: the LARGEST d <= min(i,W) with
// fc[i-d : i-d+L] == fc[i : i+L] and d >= L (leftmost occurrence, bytes.Index lzss.go:419).
// One wavefront scans 64 candidate distances per step from the far end of the window; the
// bytes come from the tile image in LDS (sb[k] = fc[org + k]).
__device__ __forceinline__ uint32_t leftmost_distance(const uint8_t *sb, uint32_t org, uint32_t i, uint32_t L, uint32_t W, int lane) {
    const uint32_t dmax = min(i, W);
    const uint32_t mi = i - org;                                  // index of fc[i] in the tile image
    const uint32_t *sw = reinterpret_cast<const uint32_t *>(sb);
    auto load4 = [&](uint32_t k) {                                // 4 bytes at any byte index: two aligned dwords + v_alignbyte
        const uint32_t w0 = sw[k >> 2], w1 = sw[(k >> 2) + 1];
        return __builtin_amdgcn_alignbyte(w1, w0, k & 3);
    };
    const uint32_t pat4 = load4(mi);                              // L >= 6: the first four bytes filter the candidates
    auto verify = [&](uint32_t ci) {                              // bytes 4..L-1, four at a time
        uint32_t q = 4;
        while (q + 4 <= L && load4(ci + q) == load4(mi + q)) q += 4;
        if (q + 4 <= L) return false;
        if (q < L) return ((load4(ci + q) ^ load4(mi + q)) & (0xFFFFFFFFu >> (8 * (4 - (L - q))))) == 0;
        return true;
    };
    // 256 candidates per round (4 per lane, loads issued together), farthest first
    for (uint32_t c0 = 0; c0 + L <= dmax; c0 += 256) {
        uint32_t f[4];
#pragma unroll
        for (int u = 0; u < 4; u++) f[u] = load4(mi - dmax + c0 + u * 64 + (uint32_t)lane);   // may read a few bytes past the window: inside the image
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const uint32_t off = c0 + u * 64 + (uint32_t)lane;
            bool ok = off + L <= dmax && f[u] == pat4;
            if (ok) ok = verify(mi - dmax + off);
            const unsigned long long hit = __ballot(ok);
            if (hit) return dmax - (c0 + u * 64 + (uint32_t)__builtin_ctzll(hit));
        }
    }
    return 0;                                                      // unreachable: L was produced by some diagonal
}

// walk the tile's part of the chain, flag the visited positions, complete the keys of the
// visited positions whose match is long enough to become a token, count the output bytes
__global__ __launch_bounds__(LB) void k_parse_mark(const uint8_t *__restrict__ fc, uint32_t *__restrict__ keys, uint32_t E, uint32_t W,
                                                   const uint32_t *__restrict__ entry, uint32_t *__restrict__ flags,
                                                   unsigned long long *__restrict__ tile_bytes, uint32_t *__restrict__ redo,
                                                   unsigned long long *__restrict__ n_redo) {
    extern __shared__ __attribute__((aligned(16))) uint8_t s_bytes[];   // fc[base-W, base+PT+W): window before, longest match after
    __shared__ uint16_t nxt[PT];
    __shared__ uint32_t fl[PT / 32];
    __shared__ uint32_t part[LB / 64];
    __shared__ uint16_t need[PT / 6 + 8];             // chain positions whose distance is still unknown (matches of >= 6: at most PT/6)
    __shared__ uint32_t n_need;
    __shared__ uint16_t seg_entry[PT / 128];
    __shared__ unsigned long long unk[PT / 64];       // positions without a key (k_match_chain evaluates only where chains land)
    __shared__ uint32_t s_bad;
    const uint32_t base = blockIdx.x * PT;
    const uint32_t org = (base >= W ? base - W : 0) & ~15u;        // 16-byte aligned start of the tile image
    const uint32_t span = min(base + PT + W, E) - org;
    for (int i = threadIdx.x; i < PT; i += LB) {
        const uint32_t p = base + i;
        uint32_t L = p < E ? keys[p] >> 16 : 1;
        const unsigned long long um = __ballot(L == 0xFFFFu);      // 64 consecutive positions per wavefront
        if ((threadIdx.x & 63) == 0) unk[i >> 6] = um;
        if (L == 0 || L == 0xFFFFu) L = 1;
        nxt[i] = (uint16_t)(i + L);
    }
    for (int i = threadIdx.x; i < PT / 32; i += LB) fl[i] = 0;
    if (threadIdx.x == 0) { n_need = 0; s_bad = 0; }
    __syncthreads();
    // The chain inside the tile, without a serial walk over up to PT positions: (1) pointer jumping
    // gives, for every position, where its chain leaves its 128-position segment; (2) one lane
    // hops over the 64 segments to find each segment's entry; (3) 64 lanes walk their own segment.
    constexpr int SEG = 128;
    uint16_t *sx = reinterpret_cast<uint16_t *>(s_bytes);          // where the chain from each position leaves its segment (the byte image is staged afterwards)
    for (int i = threadIdx.x; i < PT; i += LB) sx[i] = nxt[i];
    __syncthreads();
    for (int round = 0; round < 7; round++) {                       // 2^7 = SEG hops
        bool more = false;
        for (int i = threadIdx.x; i < PT; i += LB) {
            const uint32_t j = sx[i], lim = (uint32_t)((i / SEG + 1) * SEG);
            if (j < lim) { const uint32_t k = sx[j]; sx[i] = (uint16_t)k; more = more || k < lim; }
        }
        if (!__syncthreads_or(more)) break;
    }
    const uint32_t lim = min((uint32_t)PT, E - base);
    if (threadIdx.x == 0) {
        uint32_t e = entry[blockIdx.x];
        for (int sgm = 0; sgm < PT / SEG; sgm++) {
            const uint32_t hi = (sgm + 1) * SEG;
            if (e >= hi) { seg_entry[sgm] = 0xFFFF; continue; }
            seg_entry[sgm] = (uint16_t)e;
            e = sx[e];
        }
    }
    __syncthreads();
    if (threadIdx.x < PT / SEG) {
        uint32_t i = seg_entry[threadIdx.x];
        const uint32_t hi = min((threadIdx.x + 1) * SEG, lim);
        while (i < hi) {                                             // this lane owns the flag words of its segment
            fl[i >> 5] |= 1u << (i & 31);
            if ((unk[i >> 6] >> (i & 63)) & 1) s_bad = 1;            // the chain is on a position nobody evaluated: this strip is redone
            const uint32_t j = nxt[i];
            if (j - i >= 6 && (keys[base + i] & 0xFFFFu) == 0) need[atomicAdd(&n_need, 1u)] = (uint16_t)i;   // shortest encodable match is 6 bytes (lzss.go:318-320,143); the bucket search already knows the distance
            i = j;
        }
    }
    __syncthreads();                                                 // sx is dead: the same LDS now receives the byte image
    if (n_need) {   // full 16-byte units first (loads issued together), then the ragged tail
        const uint32_t nv = span / 16;
        const uint4 *src = reinterpret_cast<const uint4 *>(fc + org);
        uint4 *dst = reinterpret_cast<uint4 *>(s_bytes);
        for (uint32_t i0 = 0; i0 < nv; i0 += 4 * LB) {
            uint4 v[4];
#pragma unroll
            for (int k = 0; k < 4; k++) { const uint32_t i = i0 + k * LB + threadIdx.x; if (i < nv) v[k] = src[i]; }
#pragma unroll
            for (int k = 0; k < 4; k++) { const uint32_t i = i0 + k * LB + threadIdx.x; if (i < nv) dst[i] = v[k]; }
        }
        for (uint32_t i = nv * 16 + threadIdx.x; i < span; i += LB) s_bytes[i] = fc[org + i];
    }
    __syncthreads();
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    for (uint32_t k = wv; k < n_need; k += LB / 64) {
        const uint32_t p = base + need[k];
        const uint32_t key = keys[p];
        if ((key & 0xFFFF) == 0) {
            const uint32_t d = leftmost_distance(s_bytes, org, p, key >> 16, W, lane);
            if (lane == 0) keys[p] = key | d;
        }
    }
    __syncthreads();
    uint32_t bytes = 0;
    for (int i = threadIdx.x; i < PT; i += LB) {
        if (!((fl[i >> 5] >> (i & 31)) & 1)) continue;
        const uint32_t k = keys[base + i], L = k >> 16;
        if (L == 0) bytes += 1;
        else { const uint32_t e = enc_len(k & 0xFFFF, L); bytes += e < L ? e : L; }   // lzss.go:143
    }
    for (int i = threadIdx.x; i < PT / 32; i += LB) flags[(size_t)blockIdx.x * (PT / 32) + i] = fl[i];
    for (int d = 32; d; d >>= 1) bytes += __shfl_down(bytes, d);
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = bytes;
    __syncthreads();
    if (threadIdx.x == 0) {
        tile_bytes[blockIdx.x] = (unsigned long long)part[0] + part[1] + part[2] + part[3];
        if (s_bad) { redo[blockIdx.x / (MATCH_STRIP / PT)] = 1; atomicAdd(n_redo, 1ull); }
    }
}

// ------------------------------------------------------------------ E4: token emit
__device__ __forceinline__ uint8_t *put_dec(uint8_t *o, uint32_t v) {
    char tmp[6]; int k = 0;
    do { tmp[k++] = (char)('0' + v % 10); v /= 10; } while (v);
    while (k) *o++ = (uint8_t)tmp[--k];
    return o;
}

// Two forms per tile.  LIST: the tile's chain keys arrive as k_chain_serial's compact list (ccnt[tile] entries from clist[tile * PT]); positions
// and output offsets are two running sums over it, literals and short raw copies come from the tile's staged bytes.  Per GiB of text that is
// 0.6 GB of list + 1 GB of stream instead of 4.3 GB of keys + flags + 1 GB (profiles/r03a_pmc_config4.json: k_tok_emit fetched 5.6 GB).
// FLAGS: the r02 form -- the tile's flag bitmap says which of its PT keys count -- for tiles without a list (W-periodic tiles, tiles resolved
// by k_chain_tail, everything after the general parse).
__global__ __launch_bounds__(LB) void k_tok_emit(const uint8_t *__restrict__ fc, const uint32_t *__restrict__ keys, uint32_t E,
                                                 const uint32_t *__restrict__ flags, const unsigned long long *__restrict__ tile_off,
                                                 uint8_t *__restrict__ out, const TileChain *__restrict__ tchain, uint32_t W,
                                                 const uint32_t *__restrict__ clist, const uint32_t *__restrict__ ccnt, uint32_t raw) {
    // raw: fc is the INPUT, not a copy of it with '<' turned into FF (nothing in it needs an escape, lzss.go:373-379): the map is applied here,
    // where its bytes are loaded (r05: the copy was N bytes written and read again for one byte value in 254)
    auto map16 = [&](uint4 x) { if (raw) { x.x |= bytes_equal(x.x, 0x3Cu); x.y |= bytes_equal(x.y, 0x3Cu); x.z |= bytes_equal(x.z, 0x3Cu); x.w |= bytes_equal(x.w, 0x3Cu); } return x; };
    auto map1 = [&](uint8_t v) -> uint8_t { return raw && v == 0x3C ? (uint8_t)0xFF : v; };
    constexpr int RP = LB * 16;                                    // positions per round
    constexpr int LE = 4, LR = LB * LE;                            // list form: entries per lane and per round (a round emits at most 11 * LR bytes)
    __shared__ uint32_t wsum[LB / 64];
    __shared__ __attribute__((aligned(16))) uint8_t s_raw[RP * 4 + (RP + 16) + (RP + 64)];
    static_assert(sizeof(s_raw) >= (size_t)(PT + 32) + 11 * LR + 64, "the list form's stage and image fit the same memory");
    uint32_t *s_keys = reinterpret_cast<uint32_t *>(s_raw);
    uint8_t *s_fc = s_raw + RP * 4;                                  // a raw copy of a short match runs at most 10 bytes past the round
    uint8_t *s_img = s_raw + RP * 4 + (RP + 16);                     // a round emits at most RP + 10 bytes
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const uint32_t base = blockIdx.x * PT;
    unsigned long long run = tile_off[blockIdx.x];
    const uint32_t n_list = ccnt ? ccnt[blockIdx.x] : NO_LIST;
    if (n_list != NO_LIST) {
        if (n_list == 0) return;                                    // (a last match can jump over the final partial tile)
        uint8_t *l_fc = s_raw, *l_img = s_raw + PT + 32;            // the tile's bytes (+ 10 of the next); the round's output
        for (int v = tid; v < (PT + 32) / 16; v += LB) {
            const uint32_t p = base + 16 * v;
            uint4 x = {0, 0, 0, 0};
            if (p + 16 <= E) x = *reinterpret_cast<const uint4 *>(fc + p);
            else if (p < E) { uint32_t w[4] = {0, 0, 0, 0}; for (uint32_t k = 0; p + k < E; k++) w[k >> 2] |= (uint32_t)fc[p + k] << (8 * (k & 3)); x = {w[0], w[1], w[2], w[3]}; }
            reinterpret_cast<uint4 *>(l_fc)[v] = map16(x);
        }
        for (int v = tid; v < (11 * LR + 64) / 16; v += LB) reinterpret_cast<uint4 *>(l_img)[v] = make_uint4(0u, 0u, 0u, 0u);   // the entries are ORed into it
        const uint32_t *cl = clist + (size_t)blockIdx.x * PT;
        uint32_t pos0 = tchain[blockIdx.x].entry - base;            // tile-relative position of the round's first entry
        for (uint32_t r0 = 0; r0 < n_list; r0 += LR) {
            const uint32_t i0 = r0 + LE * tid;
            uint32_t kk[LE] = {0, 0, 0, 0};
            if (i0 + LE <= n_list) { const uint4 v = *reinterpret_cast<const uint4 *>(cl + i0); kk[0] = v.x; kk[1] = v.y; kk[2] = v.z; kk[3] = v.w; }
            else for (int u = 0; u < LE; u++) if (i0 + u < n_list) kk[u] = cl[i0 + u];
            uint32_t mine = 0;                                       // steps << 16 | bytes, summed over the lane's entries (a tile's steps add up to < PT + 4096)
#pragma unroll
            for (int u = 0; u < LE; u++) {
                if (i0 + u >= n_list) break;
                const uint32_t L = kk[u] >> 16, e = enc_len(kk[u] & 0xFFFF, L);
                mine += (max(1u, L) << 16) | (L == 0 ? 1u : (e < L ? e : L));   // lzss.go:143
            }
            uint32_t incl = mine;
            for (int d = 1; d < 64; d <<= 1) { const uint32_t y = __shfl_up(incl, d); if (lane >= d) incl += y; }
            if (lane == 63) wsum[wv] = incl;
            __syncthreads();                                         // (the first round: the staged bytes are in place too)
            uint32_t pre = 0, tot = 0;
            for (int k = 0; k < LB / 64; k++) { if (k < wv) pre += wsum[k]; tot += wsum[k]; }
            const uint32_t ex = pre + incl - mine;
            uint8_t *dst = out + run;
            const uint32_t al = (uint32_t)((uintptr_t)dst & 15);
            uint32_t ob = al + (ex & 0xFFFF);                        // where in the image the lane's next entry goes
            uint32_t q = pos0 + (ex >> 16);
            // One path for every entry, no loops: its (at most 11) bytes -- a token's text put together from arithmetic digits, or
            // the bytes themselves out of the stage -- are twelve bytes in registers, cut to length, shifted to the byte they start
            // at and ORed into the zeroed image as four aligned words.  (Digit loops, byte stores and a copy loop per entry had every
            // lane of a wavefront walk all three paths: this kernel is bound by vector issue.)
            const uint32_t *fc32 = reinterpret_cast<const uint32_t *>(l_fc);
            uint32_t *img32 = reinterpret_cast<uint32_t *>(l_img);
#pragma unroll
            for (int u = 0; u < LE; u++) {
                const bool live = i0 + u < n_list;
                const uint32_t L = kk[u] >> 16, off = kk[u] & 0xFFFF, el = enc_len(off, L);
                const bool tok = L != 0 && el < L;
                const uint32_t nb = !live ? 0u : (L == 0 ? 1u : (tok ? el : L));   // 0 .. 11
                const uint32_t qa = q >> 2;
                const uint32_t r0 = fc32[qa], r1 = fc32[qa + 1], r2 = fc32[qa + 2], r3 = fc32[qa + 3];
                uint32_t x0 = __builtin_amdgcn_alignbyte(r1, r0, q), x1 = __builtin_amdgcn_alignbyte(r2, r1, q), x2 = __builtin_amdgcn_alignbyte(r3, r2, q);   // (v_alignbyte uses q[1:0])
                {
                    auto dec4 = [](uint32_t v, uint32_t &nd) {          // v < 10000: its digits, the first in byte 0, and how many
                        const uint32_t hi = (v * 5243u) >> 19, lo = v - hi * 100u;               // v / 100 (exact below 43699), v % 100
                        const uint32_t d3 = (hi * 103u) >> 10, d1 = __umul24(lo, 103u) >> 10;    // x / 10 for x < 100  (__umul24: the compiler cannot see that lo < 100 and would take the quarter-rate 32-bit multiply)
                        const uint32_t w = (d3 | ((hi - d3 * 10u) << 8) | (d1 << 16) | ((lo - d1 * 10u) << 24)) + 0x30303030u;
                        nd = 1u + (v > 9u ? 1u : 0u) + (v > 99u ? 1u : 0u) + (v > 999u ? 1u : 0u);
                        return w >> (8u * (4u - nd));
                    };
                    uint32_t nd1, nd2;
                    const uint32_t ow = dec4(off, nd1), lw = dec4(L, nd2);
                    const unsigned long long A = 0x3Cull | ((unsigned long long)ow << 8) | (0x2Cull << (8u * (1u + nd1)));   // '<' digits ','
                    const unsigned long long B = (unsigned long long)lw | (0x3Eull << (8u * nd2));                            // digits '>'
                    const uint32_t sh = 8u * (2u + nd1);                                                                      // 24 .. 48 bits
                    const unsigned long long T = A | (B << sh);
                    if (tok) { x0 = (uint32_t)T; x1 = (uint32_t)(T >> 32); x2 = (uint32_t)(B >> (64u - sh)); }
                }
                {   // cut to nb bytes
                    const unsigned long long M = nb >= 8u ? ~0ull : ((1ull << (8u * nb)) - 1ull);
                    x0 &= (uint32_t)M; x1 &= (uint32_t)(M >> 32);
                    x2 &= nb > 8u ? ((1u << (8u * (nb - 8u))) - 1u) : 0u;
                }
                const uint32_t b = ob & 3u, k = (4u - b) & 3u;
                const uint32_t y0 = x0 << (8u * b);
                const uint32_t y1 = b ? __builtin_amdgcn_alignbyte(x1, x0, k) : x1;
                const uint32_t y2 = b ? __builtin_amdgcn_alignbyte(x2, x1, k) : x2;
                const uint32_t y3 = b ? x2 >> (8u * k) : 0u;
                uint32_t *w = img32 + (ob >> 2);
                if (nb) { atomicOr(w, y0); atomicOr(w + 1, y1); atomicOr(w + 2, y2); atomicOr(w + 3, y3); }
                ob += nb;
                q += live ? max(1u, L) : 0u;
            }
            __syncthreads();
            drain_block_zero(l_img, al, tot & 0xFFFF, dst);
            run += tot & 0xFFFF;
            pos0 += tot >> 16;
            __syncthreads();
        }
        return;
    }
    const bool periodic = tchain && tchain[blockIdx.x].walked == 2;     // a W-periodic tile of the chain walk: its keys are arithmetic, not stored
    if (periodic && ccnt && tchain[blockIdx.x].pad == 1 && enc_len(W, W) < W && PT / W < (uint32_t)LB) {
        // ... and so are its tokens (k_chain_periodic placed the chain: from `entry` in steps of W): a lane each, no flags, no staging.
        // (Only when the chain walk's own parse was accepted -- ccnt is passed then: after the general parse the flags decide.)
        const uint32_t entry = tchain[blockIdx.x].entry, t1 = min(base + (uint32_t)PT, E);
        if (entry >= base && entry < t1) {
            const uint32_t x = entry + (uint32_t)tid * W;
            if (x >= entry && x < t1) {                                    // (x >= entry: no wrap-around)
                const uint32_t L = min(W, E - x);                          // < W only for the stream's last token
                uint8_t *o = out + run + (size_t)tid * enc_len(W, W);
                if (enc_len(W, L) < L) { *o++ = '<'; o = put_dec(o, W); *o++ = ','; o = put_dec(o, L); *o++ = '>'; }
                else for (uint32_t j = 0; j < L; j++) *o++ = map1(fc[x + j]);
            }
        }
        return;
    }
    auto key_at = [&](uint32_t p) { return periodic ? (min(W, E - p) << 16) | W : keys[p]; };
    for (int r = 0; r < PT / RP; r++) {
        const uint32_t rb = base + r * RP;
        // 16 consecutive positions per lane = one half-word of the flag mask
        const uint32_t i0 = tid * 16;
        const uint32_t fw = (flags[(size_t)blockIdx.x * (PT / 32) + ((r * RP + i0) >> 5)] >> (i0 & 31)) & 0xFFFF;
        const int busy = __syncthreads_count(fw != 0);             // lanes with a chain position in their 16
        if (busy == 0) continue;                                   // inside a long match
        if (busy < 8) {                                            // a handful of tokens: staging the round would cost more than it saves
            uint32_t mine = 0;
            for (uint32_t m = fw; m; m &= m - 1) {
                const uint32_t k = key_at(rb + i0 + __builtin_ctz(m)), L = k >> 16;
                if (L == 0) mine += 1;
                else { const uint32_t e = enc_len(k & 0xFFFF, L); mine += e < L ? e : L; }
            }
            uint32_t incl = mine;
            for (int d = 1; d < 64; d <<= 1) { const uint32_t y = __shfl_up(incl, d); if (lane >= d) incl += y; }
            if (lane == 63) wsum[wv] = incl;
            __syncthreads();
            uint32_t pre = 0, tot = 0;
            for (int k = 0; k < LB / 64; k++) { if (k < wv) pre += wsum[k]; tot += wsum[k]; }
            uint8_t *o = out + run + pre + incl - mine;
            for (uint32_t m = fw; m; m &= m - 1) {
                const uint32_t p = rb + i0 + __builtin_ctz(m);
                const uint32_t k = key_at(p), L = k >> 16, off = k & 0xFFFF;
                if (L == 0) { *o++ = map1(fc[p]); continue; }
                if (enc_len(off, L) < L) { *o++ = '<'; o = put_dec(o, off); *o++ = ','; o = put_dec(o, L); *o++ = '>'; }
                else for (uint32_t j = 0; j < L; j++) *o++ = map1(fc[p + j]);
            }
            run += tot;
            __syncthreads();
            continue;
        }
        for (int v = tid; v < RP / 4; v += LB) {                   // keys and bytes of the round, 16 bytes per load
            const uint32_t p = rb + 4 * v;
            uint4 x = {0, 0, 0, 0};
            if (periodic) { if (p < E) x.x = key_at(p); if (p + 1 < E) x.y = key_at(p + 1); if (p + 2 < E) x.z = key_at(p + 2); if (p + 3 < E) x.w = key_at(p + 3); }
            else if (p + 4 <= E) x = *reinterpret_cast<const uint4 *>(keys + p);
            else { if (p < E) x.x = keys[p]; if (p + 1 < E) x.y = keys[p + 1]; if (p + 2 < E) x.z = keys[p + 2]; }
            reinterpret_cast<uint4 *>(s_keys)[v] = x;
        }
        for (int v = tid; v < (RP + 16) / 16; v += LB) {
            const uint32_t p = rb + 16 * v;
            uint4 x = {0, 0, 0, 0};
            if (p + 16 <= E) x = *reinterpret_cast<const uint4 *>(fc + p);
            else if (p < E) { uint32_t w[4] = {0, 0, 0, 0}; for (uint32_t k = 0; p + k < E; k++) w[k >> 2] |= (uint32_t)fc[p + k] << (8 * (k & 3)); x = {w[0], w[1], w[2], w[3]}; }
            reinterpret_cast<uint4 *>(s_fc)[v] = map16(x);
        }
        __syncthreads();
        uint32_t mine = 0;
        for (uint32_t m = fw; m; m &= m - 1) {
            const uint32_t k = s_keys[i0 + __builtin_ctz(m)], L = k >> 16;
            if (L == 0) mine += 1;
            else { const uint32_t e = enc_len(k & 0xFFFF, L); mine += e < L ? e : L; }
        }
        uint32_t incl = mine;
        for (int d = 1; d < 64; d <<= 1) { const uint32_t y = __shfl_up(incl, d); if (lane >= d) incl += y; }
        if (lane == 63) wsum[wv] = incl;
        __syncthreads();
        uint32_t pre = 0, tot = 0;
        for (int k = 0; k < LB / 64; k++) { if (k < wv) pre += wsum[k]; tot += wsum[k]; }
        uint8_t *dst = out + run;
        const uint32_t al = (uint32_t)((uintptr_t)dst & 15);
        uint8_t *o = s_img + al + pre + incl - mine;
        for (uint32_t m = fw; m; m &= m - 1) {
            const uint32_t q = i0 + __builtin_ctz(m);
            const uint32_t k = s_keys[q], L = k >> 16, off = k & 0xFFFF;
            if (L == 0) { *o++ = s_fc[q]; continue; }
            if (enc_len(off, L) < L) { *o++ = '<'; o = put_dec(o, off); *o++ = ','; o = put_dec(o, L); *o++ = '>'; }
            else for (uint32_t j = 0; j < L; j++) *o++ = s_fc[q + j];
        }
        __syncthreads();
        drain_block(s_img, al, tot, dst);
        run += tot;
        __syncthreads();
    }
}

// ======================================================================= host side
size_t lzss_compress_bound(size_t n) { return 2 * n + 64; }

__global__ __launch_bounds__(256) void k_flag_listed(const uint32_t *__restrict__ list, uint32_t n, uint32_t *__restrict__ flags, uint32_t tiles_per_strip) {   // flags[strip of tile] = 1 for the listed tiles (bits 31, 30 of an entry: k_chain_verify's marks)
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i < n) flags[(list[i] & 0x3FFFFFFFu) / tiles_per_strip] = 1u;
}
__global__ __launch_bounds__(256) void k_count_flags(const uint32_t *__restrict__ flags, uint32_t n, unsigned long long *__restrict__ out) {   // how many of the flags are set
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    const unsigned long long m = __ballot(i < n && flags[i] != 0);
    if ((threadIdx.x & 63u) == 0 && m) atomicAdd(out, (unsigned long long)__builtin_popcountll(m));
}

// One pass over one escaped stream of E < 2^31 positions: match search, greedy chain, token emit.  halo: the stream is a SECTION of a
// longer one -- its first HALO_TILES tiles are window only and the chain enters at the first position behind them; stop_tile (0: none):
// the section ends in front of that tile -- *out_n is then the bytes of the items that begin before it, *exit_pos where the chain first
// lands in or behind it (the next section's entry).  d_same / Wp: k_esc_try's periodicity flags for this very stream, or null.
// materialize (r05; may be empty): d_fc is the caller's INPUT, which needs no escape -- the chain walk and the token emitter map '<' to FF
// where they load it (ChainArgs::redo bit 3, k_tok_emit's `raw`); every other kernel wants the escaped stream in memory, and the first
// time one of them is about to run the callback writes that copy and returns it.
static int lzss_encode_stream(Ctx &c, hipStream_t s, const uint8_t *d_fc, uint32_t E, uint32_t W, const uint8_t *d_same, uint32_t Wp, bool copied,
                              bool halo, uint32_t stop_tile, uint8_t *d_out, size_t out_cap, size_t *out_n, uint32_t *exit_pos,
                              const std::function<int(const uint8_t **)> &materialize = {}, uint32_t pfrom = 0xFFFFFFFFu) {
    void *p; int rc;
    bool raw = (bool)materialize;
    auto need_copy = [&]() -> int { if (!raw) return RSN_OK; raw = false; return materialize(&d_fc); };
    void *hp; rc = pinned_buf(c, 64, &hp); if (rc) return rc;
    unsigned long long *h64 = (unsigned long long *)hp;
    const uint32_t halo_bit = halo ? 4u : 0u;
    if (W > MAX_WINDOW) {
        if (halo || stop_tile) return c.fail(RSN_ERR_LIMIT, "lzss: a stream of 2 GiB and more is encoded in sections only for windows up to %u", MAX_WINDOW);
        rc = need_copy(); if (rc) return rc;
        return lzss_encode_big(c, s, d_fc, E, W, d_out, out_cap, out_n);   // lzss_big.hip: exact at any window, not fast
    }
    // ---- E2 + E3.  Chain mode (default, W <= 4096): keys only where greedy chains land, everything else
    //      KEY_UNKNOWN; if the parse finds the true chain on an unknown position, those strips are
    //      searched at every position and the parse runs again (never more rounds than strips).
    rc = dev_buf(c, 10, (size_t)E * 4 + 64, &p); if (rc) return rc;
    uint32_t *d_keys = (uint32_t *)p;
    const uint32_t n_strips = (uint32_t)ceil_div(E, MATCH_STRIP);
    rc = dev_buf(c, 18, (size_t)n_strips * 12 + 64, &p); if (rc) return rc;
    uint32_t *d_heavy = (uint32_t *)p, *d_redo = d_heavy + n_strips, *d_dense = d_redo + n_strips;
    static const bool allpos = getenv("RSN_LZSS_ALLPOS") != nullptr;                                            // A/B switch: bucket search at every position
    const bool hashed = W <= HWMAX;
    if ((halo || stop_tile) && !hashed) return c.fail(RSN_ERR_LIMIT, "lzss: a stream of 2 GiB and more is encoded in sections only for windows up to %d", HWMAX);
    bool chain_mode = hashed && !allpos;                             // (a sample of tiles may still send the whole stream to the bucket search, below)
    if (!chain_mode) { rc = need_copy(); if (rc) return rc; }
    // E3 buffers (the chain walk fills flags and tile bytes itself when its per-tile chains join up)
    const uint32_t n_pt = (uint32_t)ceil_div(E, PT);
    rc = dev_buf(c, 12, (size_t)n_pt * 8 + (size_t)n_pt * (PT / 32) * 4 + ((size_t)n_pt * 2 + 4) * 8 + (size_t)n_pt * sizeof(TileChain) + 64, &p); if (rc) return rc;
    unsigned long long *d_tbytes = (unsigned long long *)p, *d_toff = d_tbytes + n_pt, *d_ttot = d_toff + n_pt;   // d_ttot[1..2]: strips to redo / the chain walk's three failure counts
    uint32_t *d_entry = (uint32_t *)(d_ttot + 4);                       // (also k_prev_walked's output: the two are never live together)
    uint32_t *d_flags = d_entry + n_pt;
    TileChain *d_tchain = (TileChain *)(d_flags + (size_t)n_pt * (PT / 32));
    bool parsed = false;                                              // flags + tile offsets + total are final
    auto sweep = [&](const uint32_t *only) -> int {                   // k_match2 on every strip, or on the flagged ones
        // (r05: a block's time is its (strip + W) / 64 position blocks in a row -- 8 ms for 16384 positions under the engine's window -- so a
        //  short stream is swept in shorter strips: 1 MiB of a period broken every 100 KB 15.7 -> 4 ms, scripts/probes/periodic_lzss.py)
        uint32_t strip = n_strips <= 64 ? 2048u : n_strips <= 1024 ? 4096u : (uint32_t)MATCH_STRIP;
        bool wide = false;                                            // sixteen wavefronts a block instead of four (see k_match2)
        if (only) {
            // (r06) ... and what counts is how many strips are flagged, not how long the stream is: 16 MiB of 256-byte records with a
            // counter had three heavy strips of 2081 tiles and waited 7.9 ms of its 10.2 for three blocks.  Shorter strips while the
            // flagged ones make no more than 128 blocks (a block of 512 positions: 72 position blocks in a row instead of 320 -- 7.9 ->
            // 2.4 ms; at 272 blocks of 1024 instead of 136 of 2048 a period broken every 100 KB lost: 4.8 -> 5.6 ms a MiB).
            RSN_HIP(hipMemsetAsync(d_ttot + 3, 0, 8, s));
            RSN_LAUNCH("lzss_scan", k_count_flags, dim3((uint32_t)ceil_div(n_strips, 256)), dim3(256), 0, s, only, n_strips, (unsigned long long *)(d_ttot + 3));
            RSN_HIP(hipMemcpyAsync(h64 + 3, d_ttot + 3, 8, hipMemcpyDeviceToHost, s));
            RSN_HIP(hipStreamSynchronize(s));
            const uint64_t flagged = h64[3];
            if (flagged == 0) return RSN_OK;
            uint32_t fine = 512u;
            while (fine < (uint32_t)MATCH_STRIP && flagged * ((uint64_t)MATCH_STRIP / fine) > 128) fine *= 2;
            strip = std::min(strip, fine);
            static const int wide_env = [] { const char *e = getenv("RSN_LZSS_SWEEP_WIDE"); return e ? atoi(e) : -1; }();   // A/B switch: 0 never, 1 whenever the window allows
            wide = wide_env != 0 && W >= 1024;                        // (a sweep of flagged strips: 4.8 -> 4.05 ms on a period of 256 broken every 100 KB, 1 MiB; 9.1 -> 8.7 at 8 MiB; the mixed 96 MiB stream's 15 strips 4.3 -> 4.1)
            static const bool dbg_sweep = getenv("RSN_DEBUG") != nullptr;
            if (dbg_sweep) fprintf(stderr, "lzss sweep: %llu of %u strips flagged, blocks of %u positions\n", (unsigned long long)flagged, n_strips, strip);
        }
        const uint32_t mw = wide ? (uint32_t)MW2_WIDE : (uint32_t)MW2;
        MatchArgs m2{d_fc, E, W, (W + mw - 1) / mw, d_keys, only, strip};
        const uint32_t WUB = (W + 63) / 64 * 64, W4b = m2.DW * mw + 16;
        const size_t shmem2 = (size_t)((strip + WUB + W4b + 15) & ~15u) + (size_t)mw * ((m2.DW + 1) / 2) * 4 + 2 * mw * 64 * 4 + 16;
        return lzss_launch_match2(c, s, m2, (uint32_t)ceil_div(E, strip), shmem2, wide);
    };
    using CC = ChainCfg<8192, 1024, 64>;                           // 8192-position tiles (= parse tiles), 16 wavefronts, a start every 64 positions
    using CCR = ChainCfg<8192, 1024, 64, true>;                     // the same walk for a stream that holds runs of a byte (k_esc_try's flag, Ctx::lz_runs; RSN_LZSS_RUNS=0 / 1: never / always -- the tests)
    static const int runs_env = [] { const char *e = getenv("RSN_LZSS_RUNS"); return e ? atoi(e) : -1; }();
    bool runs = runs_env < 0 ? c.lz_runs : runs_env != 0;
    auto launch_chain = [&](const char *name, uint32_t blocks, const ChainArgs &ca) -> int {
        if (runs) RSN_LAUNCH(name, (k_match_chain<CCR>), dim3(blocks), dim3(CC::CTH), 0, s, ca);
        else RSN_LAUNCH(name, (k_match_chain<CC>), dim3(blocks), dim3(CC::CTH), 0, s, ca);
        return RSN_OK;
    };
    uint8_t *d_dump = nullptr;
    uint32_t *d_clist = nullptr, *d_ccnt = nullptr;
    bool parsed_by_walk = false;                                      // the chain walk's own per-tile parse was accepted: its lists are valid
    uint32_t *d_redo_list = nullptr, *d_redo_start = nullptr, *d_step = nullptr, *d_sbrk = nullptr, *d_hbrk = nullptr, *d_pred = nullptr;
    const uint32_t redo_cap = n_pt + 64;                              // the list of a second look: every tile can be on it (and room for the sample below)
    const uint32_t gave_cap = std::max(64u, n_pt / 64);               // ... which is worth it while this many tiles at most gave up as dense / heavy: beyond, the bucket search is the tool
    if (chain_mode) {
        void *dp; rc = dev_buf(c, 19, (size_t)n_pt * CC::DUMP_BYTES + 64, &dp); if (rc) return rc;   // (slot 19 is the decoder's too: never live at the same time)
        d_dump = (uint8_t *)dp;
        void *rl; rc = dev_buf(c, 26, (size_t)redo_cap * 8 + (size_t)n_pt * 16 + 64, &rl); if (rc) return rc;
        d_redo_list = (uint32_t *)rl;
        d_redo_start = d_redo_list + redo_cap;                           // the entry each listed tile is walked from
        d_step = d_redo_start + redo_cap;                                 // per tile: its whole-distance step (k_match_chain), then k_stretch_*'s scratch
        d_sbrk = d_step + n_pt; d_hbrk = d_sbrk + n_pt; d_pred = d_hbrk + n_pt;
    }
    // Nearly incompressible input is the one case where the chain walk loses (every tile walks a few dozen visits per wavefront,
    // gives up as dense, and the bucket search does it all again: 57 against 41 ms per GiB of random bytes).  Large inputs walk a
    // sample of 64 tiles first; if three quarters of them give up, the whole stream goes to the bucket search at every position.
    // (From 8192 tiles = 64 MiB: the sample is a launch that waits for one tile plus a host sync, 0.12 ms -- 15 % of an 8 MiB call,
    //  to save a stream of noise that size 0.13 ms.  RSN_LZSS_SAMPLE_MIN_TILES moves the threshold: the tests use it.)
    if (chain_mode && copied && d_same && W == Wp && !halo)
        RSN_LAUNCH("lzss_tile_periodic", k_tiles_from_blocks, dim3((uint32_t)ceil_div(n_pt, 256)), dim3(256), 0, s, (const uint8_t *)d_same, E, W, (uint32_t)PT, n_pt, d_tchain, d_step);
    else if (chain_mode) RSN_LAUNCH("lzss_tile_periodic", k_tile_periodic, dim3(n_pt), dim3(256), 0, s, d_fc, E, W, (uint32_t)PT, d_tchain, d_step);
    if (chain_mode && halo) RSN_LAUNCH("lzss_tile_periodic", k_halo_init<CC>, dim3(std::min(HALO_TILES, n_pt)), dim3(256), 0, s, d_tchain, d_flags, d_tbytes, d_dump, d_step, (uint32_t)PT);
    constexpr uint32_t SAMPLE_TILES = 64;
    const char *smin_env = getenv("RSN_LZSS_SAMPLE_MIN_TILES");
    const uint32_t sample_min = smin_env ? std::max<uint32_t>((uint32_t)atoi(smin_env), 16 * SAMPLE_TILES) : 128 * SAMPLE_TILES;
    if (chain_mode && n_pt >= sample_min) {
        RSN_HIP(hipMemsetAsync(d_heavy, 0, (size_t)n_strips * 12, s));
        RSN_HIP(hipMemsetAsync(d_ttot, 0, 32, s));
        RSN_LAUNCH("lzss_sample", k_sample_tiles, dim3(1), dim3(SAMPLE_TILES), 0, s, d_redo_list, SAMPLE_TILES, n_pt);
        ChainArgs hs{d_fc, E, W, d_keys, 2u | halo_bit | (raw ? 8u : 0u), pfrom, nullptr, ChainTail{d_heavy, d_dense, d_tchain, d_dump, d_redo_list, (uint32_t *)(d_ttot + 3), d_step, d_redo_start, nullptr, nullptr}};
        rc = launch_chain("lzss_sample", SAMPLE_TILES, hs); if (rc) return rc;
        RSN_HIP(hipMemcpyAsync(h64 + 3, d_ttot + 3, 8, hipMemcpyDeviceToHost, s));
        RSN_HIP(hipStreamSynchronize(s));
        if ((uint32_t)h64[3] * 4 >= SAMPLE_TILES * 3) { chain_mode = false; rc = need_copy(); if (rc) return rc; }
    }
    if (chain_mode) {
        RSN_HIP(hipMemsetAsync(d_heavy, 0, (size_t)n_strips * 12, s));   // (the keys are not cleared: k_chain_unknown marks the gaps if the general parse is needed)
        // k_chain_serial's compact key lists (k_tok_emit's input where a tile has one): in the memory the general parse would use for its
        // exits -- the lists are read only if that parse never runs.  Where k_chain_serial will run, k_match_chain leaves it the
        // claimed positions' keys in the same memory (RSN_LZSS_NO_CKEYS: it reads the key array, as it does for a tile whose records do
        // not fit -- the tests' way to that path; RSN_LZSS_TAIL_SERIAL: k_chain_serial whatever the size, the tests' way to what 256 MiB take).
        static const bool no_ckeys = getenv("RSN_LZSS_NO_CKEYS") != nullptr, tail_serial = getenv("RSN_LZSS_TAIL_SERIAL") != nullptr;
        uint32_t *d_ckn = nullptr;
        {
            void *lp; rc = dev_buf(c, 11, std::max((size_t)E * 2, (size_t)n_pt * PT * 4) + 64, &lp); if (rc) return rc;
            d_clist = (uint32_t *)lp;
            rc = dev_buf(c, 27, (size_t)n_pt * 8 + 64, &lp); if (rc) return rc;
            d_ccnt = (uint32_t *)lp;
            RSN_HIP(hipMemsetAsync(d_ccnt, 0xFF, (size_t)n_pt * 8, s));
            if (!no_ckeys && (n_pt >= 32768 || tail_serial)) d_ckn = d_ccnt + n_pt;
        }
        ChainArgs ha{d_fc, E, W, d_keys, halo_bit | (raw ? 8u : 0u), pfrom, nullptr, ChainTail{d_heavy, d_dense, d_tchain, d_dump, nullptr, nullptr, d_step, d_redo_start, d_ckn ? d_clist : nullptr, d_ckn}};
        rc = launch_chain("lzss_match_chain", (uint32_t)ceil_div(E, CC::CT), ha); if (rc) return rc;
        static_assert(CC::CT == PT, "the chain walk's tiles are the parse tiles");
        static const bool no_fused = getenv("RSN_LZSS_NO_FUSED_PARSE") != nullptr;   // A/B switch: always the general parse
        const uint32_t n_prev = (uint32_t)ceil_div(n_pt, PREV_BLK);
        void *pp; rc = dev_buf(c, 25, (size_t)n_prev * 4 + 64, &pp); if (rc) return rc;
        uint32_t *d_prev_part = (uint32_t *)pp;
        bool list_gave = true;                                                    // tiles that gave up are on the look's list (until there are too many of them)
        auto resolve = [&](bool second, bool with_pred) -> int {                  // in-tile chains, periodic stretches, the joints, the offsets; one host sync
            RSN_HIP(hipMemsetAsync(d_ttot, 0, 32, s));
            // (the second look concerns a handful of tiles, dense ones among them: a lone lane's 8 000 dependent loads would be all
            //  the call waits for -- there the block-per-tile kernel, which returns at once everywhere else, is the faster one.
            //  The same holds for a whole stream below 256 MiB: a lone lane's walk is 235 us on short steps and 0.7 ms on text
            //  however few the tiles, the blocks take 17-23 ns per tile -- text: 0.29 against 0.96 ms at 64 MiB, 0.56 against 0.95
            //  at 127 MiB, level at 256 MiB, 3 against 1.9 ms at 1 GiB)
            if (second || (n_pt < 32768 && !tail_serial)) {
                RSN_LAUNCH("lzss_chain_tail", (k_chain_tail<CC, (CC::CH + CC::CT) / 2>), dim3(n_pt), dim3(512), 0, s, d_dump, d_keys, E, d_tchain, d_flags, d_tbytes, d_ccnt);
                RSN_LAUNCH("lzss_chain_tail_big", (k_chain_tail<CC, CC::CH + CC::CT>), dim3(n_pt), dim3(512), 0, s, d_dump, d_keys, E, d_tchain, d_flags, d_tbytes, d_ccnt);
            } else {
                // (tile 0 by a block of its own first: it is exempt from the density test, and the literal-heavy start of a stream --
                //  config 3's first 4096 bytes match nothing -- is 8192 one-byte steps for a lone lane, 0.75 ms that every other
                //  lane would wait for; the serial kernel then finds the tile resolved)
                RSN_LAUNCH("lzss_chain_tail", (k_chain_tail<CC, (CC::CH + CC::CT) / 2>), dim3(1), dim3(512), 0, s, d_dump, d_keys, E, d_tchain, d_flags, d_tbytes, d_ccnt);
                RSN_LAUNCH("lzss_chain_tail_big", (k_chain_tail<CC, CC::CH + CC::CT>), dim3(1), dim3(512), 0, s, d_dump, d_keys, E, d_tchain, d_flags, d_tbytes, d_ccnt);
                RSN_LAUNCH("lzss_chain_tail", k_chain_serial<CC>, dim3((uint32_t)ceil_div(n_pt, RSN_SERIAL_TILES)), dim3(64), 0, s, d_keys, E, n_pt, d_tchain, d_flags, d_tbytes, d_clist, d_ccnt,
                           (const uint32_t *)d_ckn);
            }
            RSN_LAUNCH("lzss_chain_prev", k_prev_walked, dim3(n_prev), dim3(1024), 0, s, d_tchain, n_pt, d_entry, d_prev_part);
            if (n_prev > 1) RSN_LAUNCH("lzss_chain_prev", k_prev_fix, dim3(n_prev), dim3(1024), 0, s, d_entry, n_pt, (const uint32_t *)d_prev_part);
            RSN_LAUNCH("lzss_chain_periodic", k_chain_periodic, dim3(n_pt), dim3(256), 0, s, d_tchain, d_entry, n_pt, E, W, (uint32_t)PT, d_flags, d_tbytes);
            if (with_pred) {                                              // whole-distance stretches: where the true chain enters their tiles (k_stretch_pred)
                const dim3 tg((uint32_t)ceil_div(n_pt, 256));
                RSN_LAUNCH("lzss_chain_stretch", k_stretch_flags, tg, dim3(256), 0, s, (const uint32_t *)d_step, n_pt, d_sbrk);
                RSN_LAUNCH("lzss_chain_stretch", k_prev_walked, dim3(n_prev), dim3(1024), 0, s, d_tchain, n_pt, d_hbrk, d_prev_part, (const uint32_t *)d_sbrk);
                if (n_prev > 1) RSN_LAUNCH("lzss_chain_stretch", k_prev_fix, dim3(n_prev), dim3(1024), 0, s, d_hbrk, n_pt, (const uint32_t *)d_prev_part);
                RSN_LAUNCH("lzss_chain_stretch", k_stretch_pred, tg, dim3(256), 0, s, d_tchain, d_step, d_hbrk, n_pt, (uint32_t)PT, E, d_pred);
            }
            RSN_LAUNCH("lzss_chain_verify", k_chain_verify, dim3((uint32_t)ceil_div(n_pt, 256)), dim3(256), 0, s, d_tchain, n_pt, E, (uint32_t)PT, (uint32_t *)(d_ttot + 1), d_redo_list, redo_cap,
                       d_redo_start, (const uint32_t *)(with_pred ? d_pred : nullptr), list_gave ? 1u : 0u);
            rc = scan_u64(c, s, "lzss_scan", d_tbytes, d_toff, n_pt, d_ttot); if (rc) return rc;
            RSN_HIP(hipMemcpyAsync(h64, d_ttot, 32, hipMemcpyDeviceToHost, s));
            RSN_HIP(hipStreamSynchronize(s));
            return RSN_OK;
        };
        rc = resolve(false, false); if (rc) return rc;
        static const bool dbg = getenv("RSN_DEBUG") != nullptr;
        if (!runs && runs_env < 0 && (uint32_t)h64[1]) {
            // (r06) Some tile gave up.  If as "heavy" -- a visit's long candidates did not decide: what a line or a record repeated does to
            // the lean instance, whose input showed the sample nothing -- the walk is done again by the instance that resolves
            // such visits from the window's stretches.  (A heavy tile gives up at its first such visit: the first walk was short.)
            RSN_HIP(hipMemsetAsync(d_ttot + 3, 0, 8, s));
            RSN_LAUNCH("lzss_scan", k_count_flags, dim3((uint32_t)ceil_div(n_strips, 256)), dim3(256), 0, s, (const uint32_t *)d_heavy, n_strips, (unsigned long long *)(d_ttot + 3));
            RSN_HIP(hipMemcpyAsync(h64 + 3, d_ttot + 3, 8, hipMemcpyDeviceToHost, s));
            RSN_HIP(hipStreamSynchronize(s));
            if (h64[3]) {
                if (dbg) fprintf(stderr, "lzss chain walk: %u tiles gave up, heavy ones in %llu strips: once more with the instance for repeated stretches\n", (uint32_t)h64[1], (unsigned long long)h64[3]);
                runs = true;
                RSN_HIP(hipMemsetAsync(d_heavy, 0, (size_t)n_strips * 12, s));
                RSN_HIP(hipMemsetAsync(d_ccnt, 0xFF, (size_t)n_pt * 8, s));
                rc = launch_chain("lzss_match_chain", (uint32_t)ceil_div(E, CC::CT), ha); if (rc) return rc;
                rc = resolve(false, false); if (rc) return rc;
            }
        }
        parsed = h64[1] == 0 && (uint32_t)h64[2] == 0 && !no_fused;
        if (dbg) fprintf(stderr, "lzss chain walk: %u tiles, %u gave up, %u chains that do not join, %u periodic tiles not placed\n", n_pt, (uint32_t)h64[1], (uint32_t)(h64[1] >> 32), (uint32_t)h64[2]);
        if (dbg && (uint32_t)h64[1]) {                                    // which strips, and why (heavy: the long candidates of a visit did not decide; dense: steps of one and two bytes)
            std::vector<uint32_t> fl((size_t)n_strips * 3);
            RSN_HIP(hipMemcpyAsync(fl.data(), d_heavy, fl.size() * 4, hipMemcpyDeviceToHost, s));
            RSN_HIP(hipStreamSynchronize(s));
            std::string line;
            for (uint32_t k = 0, shown = 0; k < n_strips && shown < 12; k++)
                if (fl[k] || fl[(size_t)2 * n_strips + k]) { line += " " + std::to_string(k) + (fl[k] ? "h" : "d"); shown++; }
            fprintf(stderr, "lzss chain walk: strips of %d positions with a tile that gave up (h: heavy, d: dense):%s\n", MATCH_STRIP, line.c_str());
        }
        // A few tiles gave up (a stretch of one- and two-byte steps looked "dense"), or entered on a chain that had not merged with the
        // true one yet: walk just those -- without that test / from the true entry, the tile before's exit -- and check again.  A joint
        // next to a tile that gave up can only be judged once that tile is resolved, hence up to three looks (each costs a few tiles).
        // Joints by the thousand are data whose steps are long against the 128-position warm-up (runs of a few dozen bytes: a chain
        // has three steps to merge in): the tile before's exit is still right -- its chain merged further in -- so one look from
        // there mends them all, at the price of walking those tiles again (64 MiB of 37-byte runs: 65 ms through the general parse,
        // 2258 of 8192 joints; 12 ms with the look).  Chains that keep their phase (a period that does not divide the window) do not
        // converge: a look that leaves more than half of its list behind is the last, and a list of more than half the tiles is not tried.
        // Joints that fail by the hundred may also be whole-distance stretches (k_stretch_pred): the list is then drawn up again with the
        // true chain placed through them by arithmetic, and the look walks every tile of a stretch from its predicted entry.
        bool use_pred = false;
        // (With more tiles given up than a look takes -- sections of noise -- the stream ends in the general parse whatever happens; the
        //  looks still mend the joints and place the stretches of the rest, so that the parse finds the chain on evaluated positions
        //  there and does not send those strips to the bucket search and the sweep as well.)
        list_gave = (uint32_t)h64[1] <= gave_cap;
        // (r05: "by the hundred" OR most of a short stream's tiles -- 64 KiB of a 3461-byte file repeated is eight tiles, seven of them out
        //  of phase: four looks mended four, the general parse swept the rest at 4096 compares a position, 9.1 ms; placed by arithmetic 0.5)
        if (!parsed && !no_fused && ((uint32_t)(h64[1] >> 32) > 64 || ((uint32_t)(h64[1] >> 32) >= 2 && 2 * (uint32_t)(h64[1] >> 32) > n_pt))) {
            use_pred = true;
            rc = resolve(true, true); if (rc) return rc;
            if (dbg) fprintf(stderr, "lzss chain walk, stretches placed: list of %u tiles, %u of them by arithmetic\n", (uint32_t)(h64[2] >> 32), (uint32_t)h64[3]);
        }
        uint32_t prev_plain = 0xFFFFFFFFu;
        // (r06: ... and while a list of a few dozen joints keeps getting shorter the looks go on, up to twelve: a look over sixteen tiles is
        //  0.2 ms, the sweep those tiles' strips would go to 6.5 of the 96 MiB mixed stream's 18)
        for (int look = 2; look <= 12 && !parsed && !no_fused; look++) {   // (a stretch is placed in the look after the one that mends the tile before it)
            const uint32_t n_list = (uint32_t)(h64[2] >> 32), n_gave = (uint32_t)h64[1], n_arith = use_pred ? (uint32_t)h64[3] : 0u;
            const uint32_t n_plain = n_list - std::min(n_list, n_arith);         // entries that are not placed by arithmetic: tiles that gave up, joints to mend
            if (dbg) fprintf(stderr, "lzss chain walk, before look %d: list of %u, %u of them by arithmetic, %u before\n", look, n_list, n_arith, prev_plain);
            if (look > (use_pred ? 7 : 4) && !(n_plain <= 64u && n_plain < prev_plain)) break;
            if (n_list == 0 || n_list > redo_cap || (n_gave > gave_cap && !use_pred) || (n_plain > 64 && n_plain > prev_plain / 2) || (n_plain > n_pt / 2 + 64 && !(runs && look == 2))) break;   // (r06: lines repeated for longer than a tile -- the chains of a tile inside one stretch keep their phase, but every chain lands on the stretch's end: one look from the true entries mends them all, 16 MiB of log lines 38 -> ? ms)
            prev_plain = n_plain;
            ha.redo = 3u | halo_bit | (raw ? 8u : 0u); ha.tail.redo_list = d_redo_list;
            ha.tail.ckeys = nullptr; ha.tail.ckn = nullptr;                   // (a look's tiles are resolved by k_chain_tail, from the key array)
            rc = launch_chain("lzss_match_chain", n_list, ha); if (rc) return rc;
            rc = resolve(true, use_pred); if (rc) return rc;
            parsed = h64[1] == 0 && (uint32_t)h64[2] == 0;
            if (dbg) fprintf(stderr, "lzss chain walk, look %d: %u gave up, %u chains that do not join, %u periodic tiles not placed\n", look, (uint32_t)h64[1], (uint32_t)(h64[1] >> 32), (uint32_t)h64[2]);
        }
        parsed_by_walk = parsed;
        if (!parsed) {                                                // some tile was periodic / dense / heavy, or two chains did not join: the general parse decides
            rc = need_copy(); if (rc) return rc;                      // (the bucket search, the sweep and the general parse read the escaped stream itself)
            RSN_LAUNCH("lzss_chain_unknown", k_chain_unknown<CC>, dim3(n_pt), dim3(256), 0, s, d_dump, d_tchain, n_pt, E, W, d_keys);
            // (r06) The tiles still on the looks' list -- entered on a chain that is not the true one, or given up -- are where the parse
            // will land on positions nobody evaluated: their strips are searched now, with the dense ones, instead of in a second and a
            // third round of bucket search + sweep + parse each (a period broken every 100 KB, 8 MiB: three sweeps of 5.4 ms).
            {
                static const bool no_preflag = getenv("RSN_LZSS_NO_PREFLAG") != nullptr;   // A/B switch (the tests' way to the parse's redo rounds)
                const uint32_t n_list = (uint32_t)(h64[2] >> 32);
                if (n_list && n_list <= redo_cap && !no_preflag)
                    RSN_LAUNCH("lzss_chain_unknown", k_flag_listed, dim3((uint32_t)ceil_div(n_list, 256)), dim3(256), 0, s, (const uint32_t *)d_redo_list, n_list, d_dense, (uint32_t)(MATCH_STRIP / CC::CT));
            }
            HashArgs hd{d_fc, E, W, d_keys, d_heavy, d_dense};
            rc = lzss_launch_match_hash(c, s, hd); if (rc) return rc;   // the strips the chain walk found dense
            rc = sweep(d_heavy); if (rc) return rc;
        }
    } else if (hashed) {
        RSN_HIP(hipMemsetAsync(d_heavy, 0, (size_t)n_strips * 12, s));
        HashArgs ha{d_fc, E, W, d_keys, d_heavy, nullptr};
        rc = lzss_launch_match_hash(c, s, ha); if (rc) return rc;
        rc = sweep(d_heavy); if (rc) return rc;
    } else {
        RSN_HIP(hipMemsetAsync(d_redo, 0, (size_t)n_strips * 4, s));
        rc = sweep(nullptr); if (rc) return rc;
    }
    // ---- E3: the general parse (skipped when the chain walk's own per-tile parse was accepted)
    rc = dev_buf(c, 11, (size_t)E * 2 + 64, &p); if (rc) return rc;      // (grow-only: the lists' larger block stays where it is)
    uint16_t *d_exit = (uint16_t *)p;
    const uint32_t n_groups = (uint32_t)ceil_div(n_pt, SUPER);
    void *q; rc = dev_buf(c, 17, (size_t)n_groups * PT * 4 + (size_t)n_groups * 8 + 64, &q); if (rc) return rc;
    uint32_t *d_super = (uint32_t *)q;
    unsigned long long *d_gentry = (unsigned long long *)(d_super + (size_t)n_groups * PT);
    const size_t mark_sh = std::max<size_t>((size_t)PT + 2 * (size_t)W + 48, (size_t)PT * 2);
    rc = func_dyn_lds(c, reinterpret_cast<const void *>(k_parse_mark), mark_sh); if (rc) return rc;
    for (uint32_t round = 0; !parsed; round++) {
        RSN_HIP(hipMemsetAsync(d_ttot + 1, 0, 8, s));
        RSN_LAUNCH("lzss_parse_exit", k_parse_exit, dim3(n_pt), dim3(LB), 0, s, d_keys, E, d_exit);
        RSN_LAUNCH("lzss_parse_super", k_parse_super, dim3(n_groups), dim3(LB), 0, s, d_exit, n_pt, E, d_super);
        RSN_LAUNCH("lzss_parse_chain", k_parse_chain, dim3(1), dim3(64), 0, s, d_super, n_groups, n_pt, d_gentry, (unsigned long long)(halo ? (unsigned long long)HALO_TILES * PT : 0ull));
        RSN_LAUNCH("lzss_parse_fill", k_parse_fill, dim3((uint32_t)ceil_div(n_groups, 64)), dim3(64), 0, s, d_exit, d_gentry, n_groups, n_pt, d_entry);
        RSN_LAUNCH("lzss_parse_mark", k_parse_mark, dim3(n_pt), dim3(LB), mark_sh, s, d_fc, d_keys, E, W, d_entry, d_flags, d_tbytes, d_redo, d_ttot + 1);
        rc = scan_u64(c, s, "lzss_scan", d_tbytes, d_toff, n_pt, d_ttot); if (rc) return rc;
        RSN_HIP(hipMemcpyAsync(h64, d_ttot, 16, hipMemcpyDeviceToHost, s));
        RSN_HIP(hipStreamSynchronize(s));
        if (h64[1] == 0) break;
        if (!chain_mode || round > n_strips) return c.fail(RSN_ERR_DEVICE, "lzss: internal error: the parse met an unevaluated position outside chain mode");
        // the true chain landed where no speculative chain had been: search those strips at every position
        RSN_HIP(hipMemsetAsync(d_heavy, 0, (size_t)n_strips * 4, s));
        HashArgs ha{d_fc, E, W, d_keys, d_heavy, d_redo};
        rc = lzss_launch_match_hash(c, s, ha); if (rc) return rc;
        rc = sweep(d_heavy); if (rc) return rc;
        RSN_HIP(hipMemsetAsync(d_redo, 0, (size_t)n_strips * 4, s));
    }
    size_t total = (size_t)h64[0];
    if (stop_tile && stop_tile < n_pt) {
        // the section ends in front of stop_tile: what the items before it emit, and where the chain goes on
        RSN_HIP(hipMemcpyAsync(h64, d_toff + stop_tile, 8, hipMemcpyDeviceToHost, s));
        if (parsed_by_walk) RSN_HIP(hipMemcpyAsync(h64 + 1, &d_tchain[stop_tile].entry, 4, hipMemcpyDeviceToHost, s));
        else RSN_HIP(hipMemcpyAsync(h64 + 1, d_entry + stop_tile, 4, hipMemcpyDeviceToHost, s));
        RSN_HIP(hipStreamSynchronize(s));
        total = (size_t)h64[0];
        const uint32_t ent = (uint32_t)h64[1];
        if (ent == NO_ENTRY) return c.fail(RSN_ERR_DEVICE, "lzss: internal error: the chain jumped over a section's last tile");
        *exit_pos = parsed_by_walk ? ent : stop_tile * (uint32_t)PT + ent;
    } else if (exit_pos) *exit_pos = E;
    *out_n = total;
    if (total > out_cap) { *out_n = round_up(total, 16) + 16; return c.fail(RSN_ERR_CAPACITY, "lzss: output needs %zu bytes, buffer holds %zu", total, out_cap); }
    // ---- E4
    RSN_LAUNCH("lzss_tok_emit", k_tok_emit, dim3(stop_tile && stop_tile < n_pt ? stop_tile : n_pt), dim3(LB), 0, s, d_fc, d_keys, E, d_flags, d_toff, d_out, (const TileChain *)(chain_mode ? d_tchain : nullptr), W,
               (const uint32_t *)(parsed_by_walk ? d_clist : nullptr), (const uint32_t *)(parsed_by_walk ? d_ccnt : nullptr), raw ? 1u : 0u);
    RSN_HIP(hipStreamSynchronize(s));
    return RSN_OK;
}

static int lzss_encode_admitted(Ctx &c, hipStream_t s, const uint8_t *d_in, size_t n, int64_t window, uint8_t *d_out, size_t out_cap, size_t *out_n);

// ---- a stream that is W-periodic from some chunk on (BASELINE configs[2] is nothing else: one 4096-byte block repeated; r06).
// k_esc_try's flags say which 4 KiB chunks repeat the W bytes before them.  When every chunk behind chunk k_last does, the key of EVERY
// position p >= R0 = (k_last + 1) * 4096 is (min(W, E - p), W) whatever the bytes are (k_tile_periodic's argument: the farthest distance
// there is matches as far as a match may reach, lzss.go:166-184,418-421), so from the first position e >= R0 the greedy chain lands on
// (lzss.go:134-151) the output is arithmetic: floor((E - e) / W) times the text "<W,W>", then one item for the remainder.  r05 ran the
// whole pipeline over such a stream -- 131 072 blocks of k_match_chain that return at once, two tail kernels, a placement kernel and an
// emit kernel per tile: 0.94 ms of which the one pass that has to read the input (the check) is 0.25.  Now: the check, the ordinary
// encoder on the head [0, S + W) stopped in front of tile S / 8192 (the sections' mechanism: what the items before that tile emit, and
// where the chain goes on), and one kernel that writes the rest.
__global__ __launch_bounds__(256) void k_last_unlike(const uint8_t *__restrict__ same_blk, uint32_t n_chunks, unsigned long long *__restrict__ out,
                                                     const uint8_t *__restrict__ in, size_t n, uint32_t n_samples, unsigned long long *__restrict__ flag) {
    period_sample_block(in, n, n_samples, blockIdx.x, flag);             // (r06: the sample for short periods rides along, see k_period_sample)
    // 1 + the index of the last chunk that does not repeat (chunk 0 never does): sixteen flags per thread, one atomic per wavefront that
    // has anything to say (a periodic stream: block 0's first; text: every one of them -- 256 atomics per MiB of flags = per 4 GiB of input)
    const uint32_t i0 = (blockIdx.x * 256u + threadIdx.x) * 16u;
    uint32_t m = 0;
    if (i0 + 16 <= n_chunks) {
        const uint4 v = *reinterpret_cast<const uint4 *>(same_blk + i0);
        const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const uint32_t z = ~bytes_equal(w[j], 1u);                    // 0xFF where the flag is not 1
            if (z) m = i0 + 4u * j + (31u - (uint32_t)__builtin_clz(z)) / 8u + 1u;
        }
    } else for (uint32_t i = i0; i < n_chunks; i++) if (!same_blk[i]) m = i + 1;
    for (int d = 32; d; d >>= 1) m = max(m, (uint32_t)__shfl_down((int)m, d));
    if ((threadIdx.x & 63) == 0 && m) atomicMax(out, (unsigned long long)m);
}
struct PeriodicTail {
    uint8_t *out;                       // the output's base (16-byte aligned)
    unsigned long long at, n_rep;       // the tail begins at out[at]: n_rep times the token `rep`, then `last` (a token's text, or raw bytes from `raw`)
    const uint8_t *raw;                 // the remainder's input bytes when they go out as they are ('<' -> FF, lzss.go:373-377); null: `last` is text
    uint32_t rep_len, last_len;
    uint8_t rep[12], last[12];
};
__global__ __launch_bounds__(256) void k_periodic_tail(PeriodicTail a) {
    const uint8_t *arg = (const uint8_t *)__builtin_amdgcn_kernarg_segment_ptr();   // (the texts by address: indexing the by-value copy would spill it)
    const uint8_t *rep = arg + offsetof(PeriodicTail, rep), *last = arg + offsetof(PeriodicTail, last);
    const unsigned long long body = a.n_rep * a.rep_len, total = body + a.last_len;
    const unsigned long long u0 = (a.at >> 4) + (unsigned long long)blockIdx.x * 256 + threadIdx.x;     // this thread's 16-byte unit of the output
    const unsigned long long b0 = u0 << 4;
    if (b0 >= a.at + total) return;
    uint32_t w[4] = {0, 0, 0, 0};
    uint32_t ph = b0 >= a.at ? (uint32_t)((b0 - a.at) % a.rep_len) : 0u;
    bool whole = b0 >= a.at && b0 + 16 <= a.at + total;
    for (int k = 0; k < 16; k++) {
        const unsigned long long j = b0 + k;                              // byte j of the output
        uint32_t v = 0;
        if (j >= a.at && j < a.at + total) {
            const unsigned long long t = j - a.at;
            if (t < body) { v = rep[ph]; ph = ph + 1 == a.rep_len ? 0u : ph + 1; }
            else { const uint32_t q = (uint32_t)(t - body); v = a.raw ? a.raw[q] : last[q]; if (a.raw && v == 0x3Cu) v = 0xFFu; }
        }
        w[k >> 2] |= v << (8 * (k & 3));
    }
    if (whole) *reinterpret_cast<uint4 *>(a.out + b0) = make_uint4(w[0], w[1], w[2], w[3]);
    else for (int k = 0; k < 16; k++) { const unsigned long long j = b0 + k; if (j >= a.at && j < a.at + total) a.out[j] = (uint8_t)(w[k >> 2] >> (8 * (k & 3))); }
}

int lzss_encode_dev(Ctx &c, hipStream_t s, const uint8_t *d_in, size_t n, int64_t window, uint8_t *d_out, size_t out_cap, size_t *out_n) {
    *out_n = 0;
    if (n == 0) return RSN_OK;                                        // CompressAsync(empty) == empty
    if (((uintptr_t)d_in & 15) || ((uintptr_t)d_out & 15)) return c.fail(RSN_ERR_ARG, "lzss: device buffers must be 16-byte aligned");
    // ~12 bytes of scratch per position of a pass (keys 4, compact lists 4, chain records, the escaped stream, a section's copy) + the
    // escaped stream itself: small calls pass straight through, large ones are admitted one device-full at a time (rsn_api.hip)
    const size_t need = n < ((size_t)32 << 20) ? 0 : 13 * std::min(n, (size_t)3 << 29) + 2 * n;
    const size_t held = scratch_admit(c, need);                        // (0 inside a host-buffer call: that one's admission covers this need)
    const int rc = lzss_encode_admitted(c, s, d_in, n, window, d_out, out_cap, out_n);
    // (the encoder's scratch: slots 8..19, 22..27, 35, 36 -- not 20 / 21, the host-buffer entry points' staging, still in use by the caller)
    scratch_release(c, held, (0xFFFull << 8) | (0x3Full << 22) | (3ull << 35));
    return rc;
}

static int lzss_encode_admitted(Ctx &c, hipStream_t s, const uint8_t *d_in, size_t n, int64_t window, uint8_t *d_out, size_t out_cap, size_t *out_n) {
    void *p; int rc;
    // ---- E1
    const uint32_t n_eb = (uint32_t)ceil_div(n, ESC_TILE);
    rc = dev_buf(c, 8, ((size_t)n_eb * 2 + 3) * 8, &p); if (rc) return rc;
    unsigned long long *d_extra = (unsigned long long *)p, *d_eoff = d_extra + n_eb, *d_etot = d_eoff + n_eb;   // d_etot[1]: the check's flags; [2]: 1 + the last chunk that does not repeat
    void *hp; rc = pinned_buf(c, 64, &hp); if (rc) return rc;
    unsigned long long *h64 = (unsigned long long *)hp;
    bool copied = false;                                              // nothing in the input needs an escape: E = n, every byte keeps its place
    // (the window, should nothing need an escape -- then E = n; a multiple of 16 for k_esc_try's periodicity flags, see there)
    const uint32_t Wp = window > 0 && (uint64_t)window <= HWMAX && (uint64_t)window < n && window % 16 == 0 ? (uint32_t)window : 0u;
    uint8_t *d_same = nullptr;
    // The check first (r05: a READ of the input -- does anything need an escape, is there a '<', which 4 KiB chunks repeat the bytes Wp
    // before them): most inputs hold no 5C and no FF, and then the escaped stream is the input with '<' mapped to FF (lzss.go:373-377),
    // which the chain walk and the token emitter can do where they load it.  The copy (r04 wrote it here, N bytes out and N back in for
    // one byte value in 254: half of config 3's traffic) is written only if a kernel that wants the stream in memory has to run.
    if (Wp) { void *sp; rc = dev_buf(c, 35, (size_t)n_eb + 64, &sp); if (rc) return rc; d_same = (uint8_t *)sp; }
    RSN_HIP(hipMemsetAsync(d_etot + 1, 0, 16, s));
    RSN_LAUNCH("lzss_esc_check", k_esc_try, dim3((uint32_t)ceil_div(n_eb, ESC_RUN)), dim3(LB), 0, s, d_in, n, (uint8_t *)nullptr, d_etot + 1, d_same ? Wp : 0u, d_same, n_eb);
    static const bool no_tail = getenv("RSN_LZSS_NO_PERIODIC_TAIL") != nullptr;   // A/B switch (tests): a W-periodic stream through the whole pipeline
    const bool tail_cand = d_same && !no_tail && (uint64_t)window == Wp && n >= ((size_t)PERIODIC_TAIL_MIN_TILES + 2) * PT;
    const uint32_t n_psamp = (uint32_t)std::min<size_t>(64, std::max<size_t>(1, n / (4 * PSAMPLE_CHUNK)));
    if (tail_cand) RSN_LAUNCH("lzss_last_unlike", k_last_unlike, dim3(std::max((uint32_t)ceil_div(n_eb, 4096), n_psamp)), dim3(256), 0, s, (const uint8_t *)d_same, n_eb, d_etot + 2, d_in, n, n_psamp, d_etot + 1);
    else RSN_LAUNCH("lzss_esc_check", k_period_sample, dim3(n_psamp), dim3(256), 0, s, d_in, n, n_psamp, d_etot + 1);
    RSN_HIP(hipMemcpyAsync(h64, d_etot + 1, 16, hipMemcpyDeviceToHost, s));
    RSN_HIP(hipStreamSynchronize(s));
    copied = (h64[0] & 1ull) == 0;
    c.lz_runs = (h64[0] & 4ull) != 0;                                 // (lzss_encode_stream's choice of walk)
    const size_t unlike_end = tail_cand ? (size_t)h64[1] * ESC_TILE : n;   // every 4 KiB chunk from this byte on repeats the W bytes before it
    h64[0] = 0;
    if (!copied) {
        RSN_LAUNCH("lzss_esc_count", k_esc_count, dim3(n_eb), dim3(LB), 0, s, d_in, n, d_extra);
        rc = scan_u64(c, s, "lzss_scan", d_extra, d_eoff, n_eb, d_etot); if (rc) return rc;
        RSN_HIP(hipMemcpyAsync(h64, d_etot, 8, hipMemcpyDeviceToHost, s));
        RSN_HIP(hipStreamSynchronize(s));
    }
    const size_t E64 = n + (size_t)h64[0];
    uint8_t *d_fc = nullptr;
    auto write_stream = [&](size_t upto = 0) -> int {                 // the escaped stream in memory (slot 9), zeroed padding behind it (upto: a prefix of a stream that needs no escape)
        const size_t len = upto ? upto : E64;
        void *q; int r2 = dev_buf(c, 9, len + 64, &q); if (r2) return r2;
        d_fc = (uint8_t *)q;
        RSN_HIP(hipMemsetAsync(d_fc + len, 0, 64, s));
        const uint32_t n_b = (uint32_t)ceil_div(len, ESC_TILE);
        if (copied) RSN_LAUNCH("lzss_esc_write", k_esc_try, dim3((uint32_t)ceil_div(n_b, ESC_RUN)), dim3(LB), 0, s, d_in, len, d_fc, d_etot + 1, 0u, (uint8_t *)nullptr, n_b);
        else RSN_LAUNCH("lzss_esc_write", k_esc_write, dim3(n_eb), dim3(LB), 0, s, d_in, n, d_eoff, d_fc);
        return RSN_OK;
    };
    // ---- W-periodic from some chunk on (see k_periodic_tail): the head through the encoder, the rest by arithmetic
    if (tail_cand && copied) {
        const size_t S = round_up(std::max(unlike_end, (size_t)1), PT);               // the head's tiles: everything that does not repeat, rounded up to tiles
        if (S + (size_t)PERIODIC_TAIL_MIN_TILES * PT <= n && S + Wp < (((size_t)1 << 31) - ((size_t)1 << 24))) {
            const uint32_t Es = (uint32_t)(S + Wp), stop_tile = (uint32_t)(S / PT);  // (W bytes of look-ahead behind the last head position: a match that begins in the head may end there)
            size_t got = 0; uint32_t e = 0;
            rc = lzss_encode_stream(c, s, d_in, Es, Wp, d_same, Wp, true, false, stop_tile, d_out, out_cap, &got, &e,
                                    [&](const uint8_t **fcp) -> int { const int r2 = write_stream(Es); *fcp = d_fc; return r2; },
                                    (uint32_t)std::max(unlike_end, (size_t)ESC_TILE));
            if (rc != RSN_OK && rc != RSN_ERR_CAPACITY) return rc;
            if (rc == RSN_OK) {
                if ((size_t)e < S || e > Es) return c.fail(RSN_ERR_DEVICE, "lzss: internal error: the head's chain left it at %u", e);
                auto text = [](uint8_t *dst, uint32_t d, uint32_t l) { char t[24]; const int k = snprintf(t, sizeof t, "<%u,%u>", d, l); memcpy(dst, t, (size_t)k); return (uint32_t)k; };   // lzss.go:318-320
                PeriodicTail pt{};
                pt.out = d_out; pt.at = got;
                const unsigned long long rem = (unsigned long long)n - e, r = rem % Wp;
                pt.n_rep = rem / Wp;
                pt.rep_len = text(pt.rep, Wp, Wp);
                if (r) {                                                  // the last item: L = E - p < W (lzss.go:143: a token only if it is shorter than what it stands for)
                    pt.last_len = text(pt.last, Wp, (uint32_t)r);
                    if (pt.last_len >= r) { pt.last_len = (uint32_t)r; pt.raw = d_in + (n - (size_t)r); }
                }
                const unsigned long long tail = pt.n_rep * pt.rep_len + pt.last_len, total = got + tail;
                *out_n = (size_t)total;
                if (total > out_cap) { *out_n = round_up((size_t)total, 16) + 16; return c.fail(RSN_ERR_CAPACITY, "lzss: output needs %llu bytes, buffer holds %zu", total, out_cap); }
                const unsigned long long units = ((pt.at + tail + 15) >> 4) - (pt.at >> 4);
                if (tail) RSN_LAUNCH("lzss_periodic_tail", k_periodic_tail, dim3((uint32_t)ceil_div((size_t)units, 256)), dim3(256), 0, s, pt);
                RSN_HIP(hipStreamSynchronize(s));
                return RSN_OK;
            }
            d_fc = nullptr;                                               // (the head alone outgrew the buffer: the whole call reports what it needs)
        }
    }
    // One pass takes a stream of up to SEC_MAX positions (32-bit positions, and ~10 bytes of scratch per position); a longer one goes
    // section by section (RSN_LZSS_SECTION_MIB: a smaller section, for the tests).
    static const size_t sec_env = [] { const char *e = getenv("RSN_LZSS_SECTION_MIB"); return e && atoi(e) > 0 ? (size_t)atoi(e) << 20 : (size_t)0; }();
    constexpr size_t SEC_MAX = ((size_t)1 << 31) - ((size_t)1 << 24);
    const size_t halo_len = (size_t)HALO_TILES * PT;
    if (E64 <= SEC_MAX && (sec_env == 0 || E64 <= sec_env + halo_len)) {
        const uint32_t E = (uint32_t)E64;
        uint32_t W;
        if (window <= 0) W = E;                                       // unbounded search buffer (lzss.go:125)
        else W = (uint32_t)std::min<uint64_t>((uint64_t)window, E);   // a window longer than the stream never binds
        if (W == 0) W = 1;
        if (copied)                                                   // the input IS the stream up to the '<' map: no copy unless a kernel asks for one
            return lzss_encode_stream(c, s, d_in, E, W, d_same, Wp, copied, false, 0, d_out, out_cap, out_n, nullptr,
                                      [&](const uint8_t **fcp) -> int { const int r2 = write_stream(); *fcp = d_fc; return r2; });
        rc = write_stream(); if (rc) return rc;
        return lzss_encode_stream(c, s, d_fc, E, W, d_same, Wp, copied, false, 0, d_out, out_cap, out_n, nullptr);
    }
    rc = write_stream(); if (rc) return rc;                           // (sections are cut out of the stream in memory)
    // ---- sections: [entry, entry + sec) each, entry = where the chain left the section before (lzss.go:134-151 is one serial walk: the
    //      only thing a section needs from its predecessors is that position and the W bytes in front of it)
    if (window <= 0 || (uint64_t)window > HWMAX) return c.fail(RSN_ERR_LIMIT, "lzss: %zu escaped bytes need sections, which take windows up to %d (asked: %lld)", E64, HWMAX, (long long)window);
    const uint32_t W = (uint32_t)window;
    const size_t sec = (sec_env ? std::min(sec_env, (size_t)1 << 30) : (size_t)1 << 30) / PT * PT;   // positions per section: whole tiles
    size_t entry = 0, written = 0;
    void *sp = nullptr;
    while (entry < E64) {
        const bool halo = entry >= halo_len;                          // (the first section begins at the stream's beginning: no halo)
        const size_t a0 = halo ? entry - halo_len : 0;
        const bool last = E64 - entry <= sec + sec / 4;               // (a short remainder rides with the section before it)
        const size_t b0 = last ? E64 : entry + sec + W;               // W bytes of look-ahead: a match that begins in the section may end there
        const uint32_t Es = (uint32_t)(b0 - a0);
        const uint32_t stop_tile = last ? 0u : (uint32_t)((entry - a0 + sec) / PT);
        // an aligned copy of its own with zeroed padding behind it, like a whole stream's (the kernels load 16 bytes at a time from the
        // base, and what lies behind the last position must not look like data): 1 GiB device to device, 0.4 ms of a 30 ms section
        rc = dev_buf(c, 36, (size_t)Es + 64, &sp); if (rc) return rc;
        RSN_HIP(hipMemcpyAsync(sp, d_fc + a0, Es, hipMemcpyDeviceToDevice, s));
        RSN_HIP(hipMemsetAsync((uint8_t *)sp + Es, 0, 64, s));
        const uint8_t *fc_s = (const uint8_t *)sp;
        size_t got = 0; uint32_t exit_local = 0;
        rc = lzss_encode_stream(c, s, fc_s, Es, std::min(W, Es), nullptr, 0, copied, halo, stop_tile, d_out + written, out_cap > written ? out_cap - written : 0, &got, &exit_local);
        if (rc == RSN_ERR_CAPACITY) { *out_n = lzss_compress_bound(n); return rc; }
        if (rc) return rc;
        written += got;
        if (last) break;
        if ((size_t)exit_local < entry - a0 + sec || exit_local > Es) return c.fail(RSN_ERR_DEVICE, "lzss: internal error: a section's chain left it at %u", exit_local);
        entry = a0 + exit_local;
    }
    *out_n = written;
    return RSN_OK;
}

// '<' -> FF in place: a section's copy of the INPUT becomes the escaped stream it stands for (nothing in it needs an escape)
__global__ __launch_bounds__(LB) void k_map3c(uint8_t *p, size_t n) {
    const size_t P = ((size_t)blockIdx.x * LB + threadIdx.x) * 16;
    if (P + 16 <= n) {
        uint4 x = *reinterpret_cast<uint4 *>(p + P);
        x.x |= bytes_equal(x.x, 0x3Cu); x.y |= bytes_equal(x.y, 0x3Cu); x.z |= bytes_equal(x.z, 0x3Cu); x.w |= bytes_equal(x.w, 0x3Cu);
        *reinterpret_cast<uint4 *>(p + P) = x;
    } else for (size_t k = P; k < n; k++) if (p[k] == 0x3C) p[k] = 0xFF;
}

// The encoder in SECTIONS as the input lands (a host-buffer call whose upload is still running, rsn_api.hip piped_call): the same
// sections a stream of 2 GiB and more takes (above) -- entry = where the chain left the section before, 64 halo tiles in front of it,
// W bytes of look-ahead behind it -- only smaller, each encoded as soon as its bytes are on the device and announced as soon as its
// tokens are final, so that the upload of the next and the download of the last run under it.  For inputs in which nothing needs an
// escape (the escaped stream is then the input, position for position; checked range by range as it arrives): anything else returns 1
// before a byte of output has been announced, and the caller encodes it whole.  Same bytes as the single pass (tested).
int lzss_encode_sliced(Ctx &c, hipStream_t s, const uint8_t *d_in, size_t n, int64_t window, uint8_t *d_out, size_t out_cap, size_t *out_n, const SliceStream &st) {
    *out_n = 0;
    static const bool dbg = getenv("RSN_DEBUG") != nullptr;
    if (window <= 0 || (uint64_t)window > HWMAX) return 1;
    if (((uintptr_t)d_in & 15) || ((uintptr_t)d_out & 15)) return c.fail(RSN_ERR_ARG, "lzss: device buffers must be 16-byte aligned");
    const uint32_t W = (uint32_t)window;
    const size_t halo_len = (size_t)HALO_TILES * PT;
    const size_t sec = std::max<size_t>(st.slice_bytes, 4 * halo_len) / PT * PT;
    if (n < 2 * sec || n >= ((size_t)1 << 40)) { if (dbg) fprintf(stderr, "lzss sliced: %zu bytes are not worth sections of %zu\n", n, sec); return 1; }
    // what the sections have produced is ANNOUNCED (and goes down) only once the whole input has been checked: an escape in the last range
    // would otherwise turn up behind output that is already on its way
    const auto t_in = std::chrono::steady_clock::now();
    auto since = [&] { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_in).count(); };
    std::vector<std::pair<size_t, size_t>> held;
    auto announce = [&](bool all_checked) -> bool { if (!all_checked) return true; for (auto &r : held) if (!st.have_out(r.first, r.second)) return false; held.clear(); return true; };
    void *p; int rc;
    rc = dev_buf(c, 8, 64, &p); if (rc) return rc;
    unsigned long long *d_flag = (unsigned long long *)p;
    void *hp; rc = pinned_buf(c, 64, &hp); if (rc) return rc;
    unsigned long long *h64 = (unsigned long long *)hp;
    RSN_HIP(hipMemsetAsync(d_flag, 0, 8, s));
    size_t entry = 0, written = 0, checked = 0;
    c.lz_runs = false;                                                // (set by the slices' checks as they come)
    void *sp = nullptr;
    while (entry < n) {
        const bool halo = entry >= halo_len;
        const size_t a0 = halo ? entry - halo_len : 0;
        const bool last = n - entry <= sec + sec / 4;
        const size_t b0 = last ? n : entry + sec + W;
        const size_t upto = last ? n : std::min(n, round_up(b0, (size_t)ESC_TILE));      // (whole 4 KiB chunks: the check's unit)
        if (!st.need_in(upto)) return c.fail(RSN_ERR_DEVICE, "lzss: the upload of a sliced call failed");
        if (dbg) fprintf(stderr, "lzss sliced: +%.2f ms: input below %zu is up\n", since(), upto);
        size_t upto_chk = upto;                                        // (as far as the upload has come: the sooner the end has been seen, the sooner the output moves)
        if (st.in_so_far) { const size_t up = st.in_so_far(); upto_chk = std::max(upto, up >= n ? n : up / ESC_TILE * ESC_TILE); }
        if (upto_chk > checked) {                                     // does anything in the new range need an escape?  (lzss.go:378-379)
            const size_t upto = upto_chk, m = upto - checked;
            const uint32_t n_eb = (uint32_t)ceil_div(m, ESC_TILE);
            RSN_LAUNCH("lzss_esc_check", k_esc_try, dim3((uint32_t)ceil_div(n_eb, ESC_RUN)), dim3(LB), 0, s, d_in + checked, m, (uint8_t *)nullptr, d_flag, 0u, (uint8_t *)nullptr, n_eb);
            { const uint32_t ns = (uint32_t)std::min<size_t>(64, std::max<size_t>(1, m / (4 * PSAMPLE_CHUNK))); RSN_LAUNCH("lzss_esc_check", k_period_sample, dim3(ns), dim3(256), 0, s, d_in + checked, m, ns, d_flag); }
            RSN_HIP(hipMemcpyAsync(h64, d_flag, 8, hipMemcpyDeviceToHost, s));
            RSN_HIP(hipStreamSynchronize(s));
            if (h64[0] & 1ull) { if (dbg) fprintf(stderr, "lzss sliced: a byte that needs an escape below position %zu: encoded whole instead\n", upto); return 1; }
            if (h64[0] & 4ull) c.lz_runs = true;
            checked = upto;
            if (!announce(checked == n)) return c.fail(RSN_ERR_DEVICE, "lzss: the download of a sliced call failed");
        }
        const uint32_t Es = (uint32_t)(b0 - a0);
        const uint32_t stop_tile = last ? 0u : (uint32_t)((entry - a0 + sec) / PT);
        rc = dev_buf(c, 36, (size_t)Es + 64, &sp); if (rc) return rc;
        RSN_HIP(hipMemcpyAsync(sp, d_in + a0, Es, hipMemcpyDeviceToDevice, s));
        RSN_HIP(hipMemsetAsync((uint8_t *)sp + Es, 0, 64, s));
        size_t got = 0; uint32_t exit_local = 0;
        rc = lzss_encode_stream(c, s, (const uint8_t *)sp, Es, std::min(W, Es), nullptr, 0, true, halo, stop_tile, d_out + written, out_cap > written ? out_cap - written : 0, &got, &exit_local,
                                [&](const uint8_t **fcp) -> int {     // (the section's copy is the input: mapped in place should a kernel want the stream itself)
                                    RSN_LAUNCH("lzss_esc_write", k_map3c, dim3((uint32_t)ceil_div((size_t)Es, (size_t)LB * 16)), dim3(LB), 0, s, (uint8_t *)sp, (size_t)Es);
                                    *fcp = (const uint8_t *)sp;
                                    return RSN_OK;
                                });
        if (rc) return rc;                                            // (RSN_ERR_CAPACITY: the caller's buffer was sized for an input without escapes -- cannot happen; reported as it is)
        held.emplace_back(written, got);
        if (!announce(checked == n)) return c.fail(RSN_ERR_DEVICE, "lzss: the download of a sliced call failed");
        if (dbg) fprintf(stderr, "lzss sliced: +%.2f ms: section [%zu, %zu) -> %zu bytes at %zu%s\n", since(), entry, b0, got, written, checked == n ? "" : " (held: input not checked to its end yet)");
        written += got;
        if (last) break;
        if ((size_t)exit_local < entry - a0 + sec || exit_local > Es) return c.fail(RSN_ERR_DEVICE, "lzss: internal error: a section's chain left it at %u", exit_local);
        entry = a0 + exit_local;
    }
    *out_n = written;
    return RSN_OK;
}

}  // namespace rsn
