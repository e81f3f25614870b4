// huff_small.hip -- the Huffman codec for a host buffer of at most 64 KiB (BASELINE configs[0]: the reference's own README case).
//
// At this size a call is its fixed costs: the general path is ~10 launches, five copy commands and three host round trips for work the
// device does in a few microseconds (r04: 89 us to compress 64 KiB, 129 us to decompress it).  Here a call is
//   compress    host memcpy into pinned memory -> k_small_hist (a block per 2 KiB tile reads it over PCIe, keeps a device copy, writes counts
//               straight back into pinned memory) -> sync -> the Go-exact tree, codes and header on the host (huffman.go:58-127,312-318)
//               -> k_small_emit (code table, header and tile bit positions in the KERNEL ARGUMENTS; a block per 2 KiB tile stores its words
//               into pinned memory) -> sync -> host memcpy into the result block:          2 launches, 2 syncs, no copy command
//   decompress  header, tree and lookup table on the host (the stream is in host memory: huffman.go:196-227,261) -> host memcpy of
//               stream + table into pinned memory -> k_small_dec (one block: self-synchronising subsequences, a lane each, in LDS;
//               decoded bytes into pinned memory) -> sync -> host memcpy:                   1 launch, 1 sync, no copy command
// For byte alphabets (every symbol < 0x80) whose stream and output both fit a block's LDS; everything else -- runes, a single symbol,
// foreign headers, malformed streams and their error texts -- returns 1 and takes the general path, which words the errors.
#include "codecs.h"
#include "huff_host.h"

namespace rsn {
namespace {

constexpr int SB = 1024;                        // lanes of the one block
constexpr uint32_t SMALL_MAX = 65536;           // bytes of input (compress) / of output (decompress)
constexpr uint32_t HDR_MAX = 1100;              // 128 entries of at most 5 digits + '|' + 2 bytes, + "\\\n" + pad
constexpr uint32_t DEC_STREAM_MAX = 65536 + 2048;   // bytes of a stream the decoder takes
constexpr int DEC_K = 11;                       // index bits of the decoder's table, at most
constexpr int DEC_ROUNDS = 64;                  // rounds of the synchronisation before the general decoder is asked instead

// pinned staging of one call (Ctx::pinned): offsets
constexpr size_t PIN_IN = 0;                                        // the caller's bytes, zero-padded to 16
constexpr size_t PIN_TAB = PIN_IN + DEC_STREAM_MAX + 64;           // decoder: lut (2^K words), then the child table
constexpr size_t PIN_OUT = PIN_TAB + ((size_t)4 << DEC_K) + 1024;   // what the kernel produced
constexpr size_t PIN_CNT = PIN_OUT + SMALL_MAX + 64;               // encoder: every tile's 128 counts (u16), then a word per tile: a byte >= 0x80 was seen; decoder: status words
constexpr size_t PIN_BYTES = PIN_CNT + 32 * 128 * 2 + 32 * 4 + 256;
static_assert(PIN_TAB % 16 == 0 && PIN_OUT % 16 == 0 && PIN_CNT % 16 == 0, "16-byte stores");

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
    if (tid == 0) tile_high[blockIdx.x] = s_high;
}

// ---------------------------------------------------------------- compress, kernel 2: the stream (huffman.go:229-256,174-191)
// One block per tile again.  Code table, header and every tile's first bit position come in the KERNEL ARGUMENTS (no upload); a block packs
// its tile's codes into an LDS image of the words they touch and stores the words it owns into pinned host memory.  Two neighbours share
// the word a tile boundary falls into: it belongs to the LATER tile, which works out the earlier one's last few bits itself (from the up to
// 31 symbols before its first) -- no atomics on memory, no zeroed output, nothing between the blocks.
struct SmallEmitArgs {
    const uint8_t *d_copy; uint32_t *hout;
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
}

#define TSTAMP(arr, k) do { if (threadIdx.x == 0) (arr)[k] = (uint32_t)__builtin_readcyclecounter(); } while (0)
// ---------------------------------------------------------------- decompress: one block (huffman.go:131-153,258-297)
struct SmallDecArgs {
    const uint32_t *lut;      // 2^K entries: len << 8 | byte, or 0x80000000 | internal node reached after K bits
    const uint16_t *child;    // [2 * node + bit]: 0x8000 | byte for a leaf, else the internal node
    const uint32_t *pay;      // the stream from a 4-byte boundary at or before its first payload byte (pinned; zero behind its end)
    uint32_t pay_words;       // words to stage
    uint32_t p0, end;         // first code bit / the bit behind the last, counted from `pay`
    uint32_t K, n_child;
    uint4 *hout; uint32_t *status;    // status[0]: 0 = done, 1 = not for this kernel; status[1]: decoded bytes
};

struct DecLds {
    const uint32_t *pay, *lut; const uint16_t *child; uint32_t K, end;
    __device__ __forceinline__ uint32_t window(uint32_t pos) const {
        const uint32_t w = pos >> 5, o = pos & 31;
        const unsigned long long two = ((unsigned long long)pay[w] << 32) | pay[w + 1];
        return (uint32_t)((two << o) >> 32);
    }
    // the codeword at `pos`: its byte; pos moves behind it (past `end`: the caller's to notice)
    __device__ __forceinline__ uint32_t one(uint32_t &pos) const {
        const uint32_t win = window(pos);
        const uint32_t e = lut[win >> (32 - K)];
        if (!(e >> 31)) { pos += e >> 8; return e & 0xFF; }
        uint32_t node = e & 0xFFFF, q = pos + K;
        for (;;) {
            const uint32_t bit = (pay[q >> 5] >> (31 - (q & 31))) & 1;
            q++;
            node = child[2 * node + bit];
            if (node & 0x8000) break;
            if (q > end + 64) break;                                  // (garbage behind the end: stop)
        }
        pos = q;
        return node & 0xFF;
    }
};

__global__ __launch_bounds__(SB) void k_small_dec(SmallDecArgs a) {
    extern __shared__ uint32_t s_mem[];
    uint32_t *s_pay = s_mem;                                  // big-endian words of the stream (+ 4 of zeros)
    uint32_t *s_lut = s_pay + (DEC_STREAM_MAX / 4 + 4);
    uint32_t *s_exit = s_lut + (1u << DEC_K);
    uint32_t *s_wave = s_exit + SB;
    uint16_t *s_child = reinterpret_cast<uint16_t *>(s_wave + 20);
    uint8_t *s_out = reinterpret_cast<uint8_t *>(s_child + 512);
    const uint32_t tid = threadIdx.x;
    TSTAMP(a.status + 4, 0);
    for (uint32_t i = tid; i < a.pay_words + 4; i += SB) s_pay[i] = i < a.pay_words ? __builtin_bswap32(a.pay[i]) : 0u;
    for (uint32_t i = tid; i < (1u << a.K); i += SB) s_lut[i] = a.lut[i];
    for (uint32_t i = tid; i < a.n_child; i += SB) s_child[i] = a.child[i];
    __syncthreads();
    TSTAMP(a.status + 4, 1);
    const DecLds d{s_pay, s_lut, s_child, a.K, a.end};
    // subsequences of S bits, a lane each
    const uint32_t span = a.end - a.p0;
    const uint32_t S = max(64u, ((span + SB - 1) / SB + 31) / 32 * 32);
    const uint32_t my_lo = a.p0 + tid * S;                                     // (may lie behind the end: such a lane takes nothing)
    const uint32_t my_hi = min(a.end, my_lo + S);
    constexpr uint32_t BAD = 0xFFFFFFFFu;
    uint32_t start = my_lo, cnt = 0, exit_ = 0;
    auto run = [&] {                                                           // codewords that START in [start, my_hi)
        cnt = 0;
        uint32_t pos = start;
        if (pos == BAD) { exit_ = BAD; return; }
        while (pos < my_hi) { (void)d.one(pos); cnt++; if (pos > a.end) { pos = BAD; break; } }
        exit_ = pos;
    };
    run();                                                                     // (a lane behind the end takes nothing and hands its start on)
    s_exit[tid] = exit_;
    TSTAMP(a.status + 4, 2);
    bool lost = false;
    int rounds = 0;
    for (int round = 0;; round++) {
        __syncthreads();
        const uint32_t want = tid == 0 ? a.p0 : s_exit[tid - 1];       // where the lane before stopped is where this one starts
        const bool redo = want != start;
        __syncthreads();
        if (redo) { start = want; run(); s_exit[tid] = exit_; }
        rounds++;
        if (!__syncthreads_or(redo ? 1 : 0)) break;
        if (round >= DEC_ROUNDS) { lost = true; break; }
    }
    // (every lane agrees on `lost`: the loop's exits are block-uniform)
    TSTAMP(a.status + 4, 3);
    if (tid == 0) a.status[12] = rounds;
    const uint32_t last = s_exit[SB - 1];
    const uint32_t at = block_excl_scan<16>(cnt, s_wave);
    const uint32_t total = s_wave[16];
    if (lost || last != a.end || total > SMALL_MAX || total == 0) {           // ends inside a codeword, never synchronised, too large: the general decoder
        if (tid == 0) { a.status[0] = 1; a.status[1] = 0; }
        return;
    }
    TSTAMP(a.status + 4, 4);
    {
        uint32_t pos = start, o = at;
        if (pos != BAD) while (pos < my_hi) s_out[o++] = (uint8_t)d.one(pos);
    }
    TSTAMP(a.status + 4, 5);
    __syncthreads();
    TSTAMP(a.status + 4, 6);
    const uint4 *o4 = reinterpret_cast<const uint4 *>(s_out);
    for (uint32_t i = tid; i < (total + 15) / 16; i += SB) a.hout[i] = o4[i];
    TSTAMP(a.status + 4, 7);
    if (tid == 0) { a.status[0] = 0; a.status[1] = total; }
}
constexpr size_t DEC_LDS = (size_t)(DEC_STREAM_MAX / 4 + 4) * 4 + ((size_t)4 << DEC_K) + SB * 4 + 20 * 4 + 512 * 2 + SMALL_MAX + 16;

}  // namespace

// 1: not an input for this path (the caller takes the general one)
int huff_small_compress(Ctx &c, const uint8_t *in, size_t n, const uint8_t **out, size_t *out_n) {
    if (n < 64 || n > SMALL_MAX) return 1;
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
    RSN_LAUNCH("huff_small_hist", k_small_hist, dim3(n_tiles), dim3(128), 0, s, (const uint4 *)(pin + PIN_IN), (uint32_t)n, (uint4 *)dp, th, hi);
    RSN_HIP(hipStreamSynchronize(s));
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
    a.d_copy = (const uint8_t *)dp; a.hout = (uint32_t *)(pin + PIN_OUT);
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
    RSN_HIP(hipStreamSynchronize(s));
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
    static thread_local bool lds_set = false;
    if (!lds_set) { RSN_HIP(hipFuncSetAttribute((const void *)k_small_dec, hipFuncAttributeMaxDynamicSharedMemorySize, (int)DEC_LDS)); lds_set = true; }
    void *pp; rc = pinned_buf(c, PIN_BYTES, &pp); if (rc) return rc;
    uint8_t *pin = (uint8_t *)pp;
    // ---- tables, written where the kernel reads them
    const int K = (int)std::min<unsigned>(codes.max_len, DEC_K);
    uint32_t *lut = (uint32_t *)(pin + PIN_TAB);
    uint16_t *child = (uint16_t *)(lut + ((size_t)1 << K));
    const uint32_t A = tree.n_leaves;
    const size_t n_int = tree.freq.size() - A;
    for (size_t i = 0; i < n_int; i++) {
        const int32_t kids[2] = {tree.left[A + i], tree.right[A + i]};
        for (int b = 0; b < 2; b++) child[2 * i + b] = tree.is_leaf(kids[b]) ? (uint16_t)(0x8000u | tree.rune[kids[b]]) : (uint16_t)(kids[b] - (int32_t)A);
    }
    {
        struct It { int32_t node; uint32_t prefix; int depth; };
        It st[300]; int top = 0;
        st[top++] = {tree.root, 0, 0};
        while (top) {
            const It it = st[--top];
            if (tree.is_leaf(it.node)) {
                const uint32_t ent = ((uint32_t)it.depth << 8) | tree.rune[it.node];
                const uint32_t lo = it.prefix << (K - it.depth);
                for (uint32_t x = 0; x < (1u << (K - it.depth)); x++) lut[lo + x] = ent;
            } else if (it.depth == K) {
                lut[it.prefix] = 0x80000000u | (uint32_t)(it.node - (int32_t)A);
            } else {
                st[top++] = {tree.right[it.node], (it.prefix << 1) | 1, it.depth + 1};
                st[top++] = {tree.left[it.node], it.prefix << 1, it.depth + 1};
            }
        }
    }
    const size_t A0 = pay & ~(size_t)3;
    memcpy(pin + PIN_IN, in + A0, n - A0);
    memset(pin + PIN_IN + (n - A0), 0, 16);
    uint32_t *status = (uint32_t *)(pin + PIN_CNT) + 132;
    status[0] = 2; status[1] = 0;
    SmallDecArgs a{};
    a.lut = lut; a.child = child; a.pay = (const uint32_t *)(pin + PIN_IN);
    a.pay_words = (uint32_t)((n - A0 + 3) / 4);
    a.p0 = (uint32_t)(8 * (pay - A0) + diff);
    a.end = (uint32_t)(8 * (pay - A0) + nbits);
    a.K = (uint32_t)K; a.n_child = (uint32_t)(2 * n_int);
    a.hout = (uint4 *)(pin + PIN_OUT); a.status = status;
    RSN_LAUNCH("huff_small_dec", k_small_dec, dim3(1), dim3(SB), DEC_LDS, s, a);
    RSN_HIP(hipStreamSynchronize(s));
    if (getenv("RSN_DEBUG")) { fprintf(stderr, "dec stamps:"); for (int k = 1; k < 8; k++) fprintf(stderr, " %u", status[4 + k] - status[4]); fprintf(stderr, "  rounds %u\n", status[12]); }
    if (status[0] != 0) return 1;
    *out = pin + PIN_OUT; *out_n = status[1];
    return RSN_OK;
}

}  // namespace rsn
