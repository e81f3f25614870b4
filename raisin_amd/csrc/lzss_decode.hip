// lzss_decode.hip -- LZSS decode for gfx950 (MI355X).
//
// Replaces lz.Decompress (compressor/lz/lzss.go:323-364) and
// DecodeOpeningSymbols (:391-406).
//
//   L1 k_lzd_count2 / k_lzd_expand  parse "<ptr,len>" tokens in parallel ('<' never
//        occurs in literals: EncodeOpeningSymbols maps it to FF, lzss.go:373-377), scan
//        the output lengths, then give every escaped-stream byte a SOURCE: itself for a
//        literal, position-ptr for a byte produced by a token.
//   L2 a token may copy bytes that were themselves produced by a token (the reference resolves
//        that by running serially, lzss.go:349-353).  Tile path (every back-pointer <= DT):
//          k_lzd_check / k_lzd_tilemap (k_lzd_tiles for streams with huge tokens)   validate the tokens, find the item that starts every DT-byte output tile
//          k_lzd_resolve  one block per tile: parse its items into LDS descriptors (literal /
//                         position inside the tile / position in the previous tile's tail),
//                         pointer-jump the in-tile references inside LDS
//          k_lzd_compose / k_lzd_mapscan   tails only depend on the previous tail: compose the
//                         tail maps of a group of tiles, scan the groups' maps under composition (a few MB in total)
//          k_lzd_emit     per group, tile after tile: literal or byte of the previous tail
//        Fallback for larger pointers: k_lzd_expand / k_lzd_jump / k_lzd_gather, pointer
//        jumping src[p] = src[src[p]] over the whole stream in HBM.
//   L4 k_une_*        DecodeOpeningSymbols: an escape byte 5C consumes the next byte, so a
//        byte is escaped iff the run of 5C immediately before it has odd length; run
//        parities are combined per lane, per block and across blocks, then count/scan/write.
// Streams the reference would panic on (ptr past the start, len > ptr, malformed token)
// return RSN_ERR_FORMAT.
#include "codecs.h"

namespace rsn {


constexpr int ZB = 256;
constexpr int ZTILE = ZB * 16;          // compressed bytes per block, 16 per lane
constexpr int MAXTOK = 24;              // longest token accepted: "<4294967295,4294967295>" is 23 bytes

struct Tok { uint32_t ptr, len, tl; bool ok; };

// parse the token that starts at in[j] == '<'
__device__ __forceinline__ Tok parse_tok(const uint8_t *__restrict__ in, size_t n, size_t j) {
    Tok t{0, 0, 0, false};
    size_t k = j + 1;
    unsigned long long v = 0; int nd = 0;
    while (k < n && nd < 10 && in[k] >= '0' && in[k] <= '9') { v = v * 10 + (in[k] - '0'); k++; nd++; }
    if (nd == 0 || k >= n || in[k] != ',' || v > 0xFFFFFFFFull) return t;
    t.ptr = (uint32_t)v; k++;
    v = 0; nd = 0;
    while (k < n && nd < 10 && in[k] >= '0' && in[k] <= '9') { v = v * 10 + (in[k] - '0'); k++; nd++; }
    if (nd == 0 || k >= n || in[k] != '>' || v > 0xFFFFFFFFull) return t;
    t.len = (uint32_t)v; t.tl = (uint32_t)(k + 1 - j); t.ok = true;
    return t;
}

// first position >= s that is not inside a token which started before s
__device__ __forceinline__ size_t skip_open_token(const uint8_t *__restrict__ in, size_t n, size_t s, int *err) {
    for (int back = 1; back < MAXTOK && (size_t)back <= s; back++) {
        const uint8_t b = in[s - back];
        if (b == '>') return s;
        if (b == '<') {
            const Tok t = parse_tok(in, n, s - back);
            if (!t.ok) { *err = 1; return s; }
            const size_t end = s - back + t.tl;
            return end > s ? end : s;
        }
    }
    return s;
}

// ------------------------------------------------------------------ span parser
// A lane owns 16 staged bytes.  Instead of walking them byte by byte (serial LDS reads, and a
// wavefront pays for every lane's control path) it classifies them with bit masks.  Every '<'
// opens a token (a literal '<' is escaped, lzss.go:373-377) that runs to the first '>' after
// it, so with C = the bytes that are not '>' and S = the '<' bytes (plus bit 0 when the span
// starts inside a token) the carry chain of C + S marks exactly the token bytes.  A token is at
// most MAXTOK - 1 = 23 bytes, so whether a span starts inside one is decided by the last '<' or
// '>' among the 22 bytes before it.
__device__ __forceinline__ uint32_t pack_bit7(uint32_t z) { return ((z >> 7) * 0x00204081u >> 21) & 0xFu; }   // bit 7 of each byte -> 4 bits

// Sixteen flag bytes (0x80 or 0x00 each, four dwords) -> a 16-bit mask, bit j = byte j.  A flag byte times its weight 2^j summed over a
// dword is one v_dot4_u32_u8; two dwords share an accumulator (weights 1..8, 16..128), the mask comes out times 0x80.  (r04: the
// multiply-and-shift of pack_bit7 is a quarter-rate v_mul_lo_u32 per dword and mask -- sixteen per span in the parsers, a tenth of
// k_lzd_count2's issue slots.)
__device__ __forceinline__ uint32_t pack16_bit7(uint32_t z0, uint32_t z1, uint32_t z2, uint32_t z3) {
    uint32_t lo = __builtin_amdgcn_udot4(z0, 0x08040201u, 0u, false);
    lo = __builtin_amdgcn_udot4(z1, 0x80402010u, lo, false);
    uint32_t hi = __builtin_amdgcn_udot4(z2, 0x08040201u, 0u, false);
    hi = __builtin_amdgcn_udot4(z3, 0x80402010u, hi, false);
    return ((hi << 8) | lo) >> 7;
}

__device__ __forceinline__ uint32_t mask_5c(const uint32_t w[4]) {                             // 16-bit mask of the 5C bytes
    uint32_t z[4];
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const uint32_t y = w[k] ^ 0x5C5C5C5Cu;
        const uint32_t t = ((y & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | y;
        z[k] = ~t & 0x80808080u;
    }
    return pack16_bit7(z[0], z[1], z[2], z[3]);
}

__device__ __forceinline__ void load_span(const uint32_t *sw, int sbyte, uint32_t w[4]) {      // 16 staged bytes from any offset
    const int q = sbyte >> 2; const uint32_t sh = (uint32_t)(sbyte & 3) * 8;
    uint32_t d[5];
#pragma unroll
    for (int k = 0; k < 5; k++) d[k] = sw[q + k];
#pragma unroll
    for (int k = 0; k < 4; k++) w[k] = __builtin_amdgcn_alignbit(d[k + 1], d[k], sh);
}

// '<' mask in the low half, '>' mask in the high half
__device__ __forceinline__ uint32_t ltgt_masks(const uint32_t w[4]) {
    uint32_t zl[4], zg[4];
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const uint32_t y = w[k] ^ 0x3C3C3C3Cu;                            // '<' -> 00, '>' -> 02
        const uint32_t y2 = y & 0xFDFDFDFDu;
        const uint32_t t = ((y2 & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | y2;       // bit 7 clear iff the byte is '<' or '>'
        const uint32_t z = ~t & 0x80808080u;
        zg[k] = z & (y << 6);                                             // bit 1 tells them apart
        zl[k] = z & ~zg[k];
    }
    return pack16_bit7(zl[0], zl[1], zl[2], zl[3]) | (pack16_bit7(zg[0], zg[1], zg[2], zg[3]) << 16);
}

// ',' mask in the low half, digit mask in the high half (the span's share of what parse_tok_w works out per token: a token's
// twelve bytes after its '<' are then a shift of two spans' masks)
__device__ __forceinline__ uint32_t cd_masks(const uint32_t w[4]) {
    uint32_t zc[4], zd[4];
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const uint32_t y = w[k] ^ 0x30303030u;                            // digits -> 00..09, ',' -> 1C
        const uint32_t td = ((y & 0x7F7F7F7Fu) + 0x76767676u) | y;        // bit 7 clear iff 00..09
        const uint32_t yc = y ^ 0x1C1C1C1Cu, tc = ((yc & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | yc;
        zd[k] = ~td & 0x80808080u;
        zc[k] = ~tc & 0x80808080u;
    }
    return pack16_bit7(zc[0], zc[1], zc[2], zc[3]) | (pack16_bit7(zd[0], zd[1], zd[2], zd[3]) << 16);
}

// the ',' mask alone (k_lzd_resolve: its tokens were validated by the counting pass, the digits need no second look)
__device__ __forceinline__ uint32_t c_mask(const uint32_t w[4]) {
    uint32_t zc[4];
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const uint32_t yc = w[k] ^ 0x2C2C2C2Cu, tc = ((yc & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | yc;
        zc[k] = ~tc & 0x80808080u;
    }
    return pack16_bit7(zc[0], zc[1], zc[2], zc[3]);
}

// the token whose '<' is staged byte j; sw = dword view, zero past the data, readable 28 bytes past j
__device__ __forceinline__ Tok parse_tok_w(const uint32_t *sw, int j) {
    const int q = (j + 1) >> 2; const uint32_t sh = (uint32_t)((j + 1) & 3) * 8;
    uint32_t a[6];
    uint32_t cm = 0, gm = 0, dm = 0;                                      // ',' / '>' / digit masks over the bytes after '<'
    auto classify = [&](int k0, int k1) {                                 // bytes 4*k0 .. 4*k1-1
        uint32_t d[4];
        for (int k = k0; k <= k1; k++) d[k - k0] = sw[q + k];
        for (int k = k0; k < k1; k++) {
            a[k] = __builtin_amdgcn_alignbit(d[k + 1 - k0], d[k - k0], sh);
            const uint32_t y = a[k] ^ 0x30303030u;                        // digits -> 00..09, ',' -> 1C, '>' -> 0E
            const uint32_t td = ((y & 0x7F7F7F7Fu) + 0x76767676u) | y;    // bit 7 clear iff 00..09
            const uint32_t yc = y ^ 0x1C1C1C1Cu, tc = ((yc & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | yc;
            const uint32_t yg = y ^ 0x0E0E0E0Eu, tg = ((yg & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | yg;
            dm |= pack_bit7(~td & 0x80808080u) << (4 * k);
            cm |= pack_bit7(~tc & 0x80808080u) << (4 * k);
            gm |= pack_bit7(~tg & 0x80808080u) << (4 * k);
        }
    };
#pragma unroll
    for (int k = 0; k < 6; k++) a[k] = 0;
    classify(0, 3);                                                       // 12 bytes: every token with numbers up to 9999 ends here
    if (gm == 0) classify(3, 6);                                          // the rare long spelling: 12 more
    Tok t{0, 0, 0, false};
    const uint32_t pc = (uint32_t)__builtin_ctz(cm | (1u << 24)), pg = (uint32_t)__builtin_ctz(gm | (1u << 24));
    if (pc < 1 || pc > 10 || pg < pc + 2 || pg > pc + 11) return t;     // 1..10 digits, ',', 1..10 digits, '>' (the first one)
    const uint32_t need = ((1u << pg) - 1u) & ~(1u << pc);
    if ((dm & need) != need) return t;
    if (pc <= 4 && pg - pc - 1 <= 4) {
        // Both numbers have at most four digits (every token of a window up to 9999: all the engine ever writes): no digit loops.
        // The field's bytes in memory order (first digit lowest) are byte-swapped and shifted so that the LAST digit is the lowest
        // byte and nothing lies above the first; then two multiply-adds fold the four digits.
        auto dec4 = [](uint32_t x, uint32_t cnt) {
            const uint32_t y = __builtin_bswap32(x & 0x0F0F0F0Fu) >> (8 * (4 - cnt));
            const uint32_t u = (y & 0x00FF00FFu) + 10u * ((y >> 8) & 0x00FF00FFu);
            return (u & 0xFFFFu) + 100u * (u >> 16);
        };
        t.ptr = dec4(a[0], pc);
        const uint32_t s = pc + 1;                                        // the second number starts here: byte 2 .. 5 after '<'
        const uint32_t f_lo = __builtin_amdgcn_alignbyte(a[1], a[0], s), f_hi = __builtin_amdgcn_alignbyte(a[2], a[1], s);   // (v_alignbyte uses s[1:0])
        t.len = dec4(s < 4 ? f_lo : f_hi, pg - pc - 1);
        t.tl = pg + 2; t.ok = true;
        return t;
    }
    const unsigned long long A0 = a[0] | ((unsigned long long)a[1] << 32), A1 = a[2] | ((unsigned long long)a[3] << 32),
                             A2 = a[4] | ((unsigned long long)a[5] << 32);
    auto dig = [&](uint32_t k) { const unsigned long long W = k < 8 ? A0 : k < 16 ? A1 : A2; return (uint32_t)(W >> (8 * (k & 7))) & 0xFu; };
    unsigned long long v = 0;
    for (uint32_t k = 0; k < pc; k++) v = v * 10 + dig(k);
    if (v > 0xFFFFFFFFull) return t;
    t.ptr = (uint32_t)v;
    v = 0;
    for (uint32_t k = pc + 1; k < pg; k++) v = v * 10 + dig(k);
    if (v > 0xFFFFFFFFull) return t;
    t.len = (uint32_t)v; t.tl = pg + 2; t.ok = true;
    return t;
}

struct Span {
    uint32_t w[4];                       // the 16 bytes
    uint32_t lit;                        // bit j: byte j is a literal
    uint32_t ntok;                       // tokens that START in the span (a valid stream has at most 4: "<1,1>" is 5 bytes)
    uint32_t tj[4], tptr[4], tlen[4];
    uint32_t out;                        // bytes the span's items produce
    bool err;
};

// masks[] holds ltgt_masks of every span, two spans of lead-in and one span beyond included: masks[sp + 2] is span sp;
// cdm[] the same for cd_masks (spans sp and sp + 1 are read)
// TRUSTED: every token of the stream is known to be well-formed (the counting pass has said so): cdm holds the ',' masks alone
template <bool TRUSTED = false>
__device__ __forceinline__ void span_parse(const uint32_t *sw, const uint32_t *masks, const uint32_t *cdm, int sp, int sbyte, int valid, Span &r) {
    const uint32_t m0 = masks[sp + 2], m1 = masks[sp + 1], m2 = masks[sp];
    const uint32_t mn = masks[sp + 3], cd0 = cdm[sp + 2], cd1 = cdm[sp + 3];
    const uint32_t gt32 = (m0 >> 16) | (mn & 0xFFFF0000u), cm32 = (cd0 & 0xFFFFu) | (cd1 << 16), dm32 = (cd0 >> 16) | (cd1 & 0xFFFF0000u);   // this span's bytes and the next one's
    const uint32_t ltw = (m1 << 16) | (m2 & 0xFFFFu), gtw = (m1 & 0xFFFF0000u) | (m2 >> 16);   // the 32 bytes before the span
    const uint32_t cin = (ltw & 0xFFFFFC00u) > (gtw & 0xFFFFFC00u) ? 1u : 0u;                  // last of '<','>' in the 22 bytes before is '<'
    const uint32_t vm = valid >= 16 ? 0xFFFFu : ((1u << valid) - 1u);
    const uint32_t lt = m0 & vm, C = ~(m0 >> 16) & 0xFFFFu;
    const uint32_t T = C + (lt | cin);
    r.lit = ~(T ^ C) & vm;
    r.ntok = 0; r.err = false;
    r.out = (uint32_t)__builtin_popcount(r.lit);
#pragma unroll
    for (int k = 0; k < 4; k++) { r.tj[k] = 16; r.tptr[k] = 0; r.tlen[k] = 0; }
    uint32_t starts = lt;
#pragma unroll
    for (int k = 0; k < 4; k++) {
        if (starts) {
            const int j = __builtin_ctz(starts);
            starts &= starts - 1;
            // What every token the engine writes looks like -- "<d..,d..>" with at most four digits each -- is recognised from the
            // spans' masks (the twelve bytes after the '<': a shift) and decoded from two 4-byte fields; anything else goes to
            // parse_tok_w, which classifies the bytes itself and decides.
            Tok t;
            {
                const uint32_t sh = (uint32_t)j + 1u;
                const uint32_t g12 = (gt32 >> sh) & 0xFFFu, c12 = (cm32 >> sh) & 0xFFFu, d12 = (dm32 >> sh) & 0xFFFu;
                const uint32_t pc = (uint32_t)__builtin_ctz(c12 | 0x1000u), pg = (uint32_t)__builtin_ctz(g12 | 0x1000u);
                const uint32_t need = ((1u << pg) - 1u) & ~(1u << pc);
                if (pc >= 1 && pc <= 4 && pg >= pc + 2 && pg <= pc + 5 && (TRUSTED || (d12 & need) == need)) {
                    const int b = sbyte + j + 1, q = b >> 2;
                    const uint32_t d0 = sw[q], d1 = sw[q + 1], d2 = sw[q + 2], d3 = sw[q + 3];
                    const uint32_t a0 = __builtin_amdgcn_alignbyte(d1, d0, (uint32_t)b), a1 = __builtin_amdgcn_alignbyte(d2, d1, (uint32_t)b),
                                   a2 = __builtin_amdgcn_alignbyte(d3, d2, (uint32_t)b);   // (v_alignbyte uses b[1:0])
                    auto dec4 = [](uint32_t x, uint32_t cnt) {            // (as in parse_tok_w)
                        const uint32_t y = __builtin_bswap32(x & 0x0F0F0F0Fu) >> (8 * (4 - cnt));
                        const uint32_t u = (y & 0x00FF00FFu) + 10u * ((y >> 8) & 0x00FF00FFu);
                        return (u & 0xFFFFu) + 100u * (u >> 16);
                    };
                    const uint32_t s2 = pc + 1;                           // the second number starts here: byte 2 .. 5 after '<'
                    const uint32_t f = s2 < 4 ? __builtin_amdgcn_alignbyte(a1, a0, s2) : __builtin_amdgcn_alignbyte(a2, a1, s2);
                    t.ptr = dec4(a0, pc); t.len = dec4(f, pg - pc - 1); t.tl = pg + 2; t.ok = true;
                } else t = parse_tok_w(sw, sbyte + j);
            }
            if (!t.ok) r.err = true;
            else { r.tj[k] = (uint32_t)j; r.tptr[k] = t.ptr; r.tlen[k] = t.len; r.out += t.len; r.ntok = k + 1; }
        }
    }
    if (starts) r.err = true;                                             // a fifth '<' cannot start a well-formed token
}

constexpr int ZPAD = 32;                // staged bytes either side of a block (two spans; a token is at most MAXTOK - 1 long)

// stage in[blk0 - ZPAD, blk0 + ZTILE + ZPAD) with 16-byte loads (zero outside the stream) and classify every span
__device__ __forceinline__ int stage_block(const uint8_t *__restrict__ in, size_t n, size_t blk0, uint32_t *sw, uint32_t *masks, uint32_t *cdm, Span &r) {
    for (int v = threadIdx.x; v < (ZTILE + 2 * ZPAD) / 16; v += ZB) {
        const long long P = (long long)blk0 - ZPAD + 16ll * v;
        uint4 x = {0, 0, 0, 0};
        if (P >= 0 && P + 16 <= (long long)n) x = *reinterpret_cast<const uint4 *>(in + P);
        else if (P + 16 > 0 && P < (long long)n) {
            uint32_t w[4] = {0, 0, 0, 0};
            for (int k = 0; k < 16; k++) { const long long q = P + k; if (q >= 0 && q < (long long)n) w[k >> 2] |= (uint32_t)in[q] << (8 * (k & 3)); }
            x = {w[0], w[1], w[2], w[3]};
        }
        reinterpret_cast<uint4 *>(sw)[v] = x;
    }
    __syncthreads();
    const int tid = threadIdx.x;
    load_span(sw, ZPAD + 16 * tid, r.w);
    masks[tid + 2] = ltgt_masks(r.w);
    cdm[tid + 2] = cd_masks(r.w);
    if (tid < 3) {                                                        // the two spans before the block, and the one after it
        uint32_t w[4];
        const int sp = tid < 2 ? tid : ZB + 2;
        load_span(sw, 16 * sp, w);
        masks[sp] = ltgt_masks(w);
        cdm[sp] = cd_masks(w);
    }
    __syncthreads();
    const long long left = (long long)n - (long long)blk0 - 16ll * tid;   // valid bytes of this lane's span
    return left >= 16 ? 16 : left > 0 ? (int)left : 0;
}

__global__ __launch_bounds__(ZB) void k_lzd_expand(const uint8_t *__restrict__ in, size_t n, const unsigned long long *__restrict__ blk_off,
                                                   uint32_t *__restrict__ src, uint8_t *__restrict__ esc, int *__restrict__ err) {
    __shared__ unsigned long long wsum[ZB / 64];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const size_t s = (size_t)blockIdx.x * ZTILE + tid * 16;
    const size_t lim = min(s + 16, n);
    unsigned long long mine = 0;
    size_t pos0 = s;
    int e = 0;
    if (s < n) {
        pos0 = skip_open_token(in, n, s, &e);
        size_t pos = pos0;
        while (pos < lim) {
            if (in[pos] == '<') { const Tok t = parse_tok(in, n, pos); if (!t.ok) { pos++; continue; } mine += t.len; pos += t.tl; }
            else { mine++; pos++; }
        }
    }
    unsigned long long incl = mine;
    for (int d = 1; d < 64; d <<= 1) { const unsigned long long y = __shfl_up(incl, d); if (lane >= d) incl += y; }
    if (lane == 63) wsum[wv] = incl;
    __syncthreads();
    unsigned long long o = blk_off[blockIdx.x] + incl - mine;
    for (int k = 0; k < wv; k++) o += wsum[k];
    if (s >= n) return;
    size_t pos = pos0;
    while (pos < lim) {
        if (in[pos] == '<') {
            const Tok t = parse_tok(in, n, pos);
            if (!t.ok) { pos++; continue; }
            // absolutePointer = len(out) - pointer must be >= 0 and the slice must end inside out (lzss.go:349-350)
            if (t.ptr > o || t.len > t.ptr) { e = 1; }
            else for (uint32_t k = 0; k < t.len; k++) src[o + k] = (uint32_t)(o + k - t.ptr);
            o += t.len; pos += t.tl;
        } else { src[o] = (uint32_t)o; esc[o] = in[pos]; o++; pos++; }
    }
    if (e) atomicOr(err, 1);
}

__global__ __launch_bounds__(ZB) void k_lzd_jump(uint32_t *__restrict__ src, uint32_t E, int *__restrict__ changed) {
    const uint32_t stride = gridDim.x * ZB;
    bool any = false;
    for (uint32_t p = blockIdx.x * ZB + threadIdx.x; p < E; p += stride) {
        const uint32_t a = src[p];
        if (a == p) continue;
        const uint32_t b = src[a];
        if (b != a) { src[p] = b; any = true; }
    }
    if (__syncthreads_or(any) && threadIdx.x == 0) *changed = 1;
}

__global__ __launch_bounds__(ZB) void k_lzd_gather(const uint32_t *__restrict__ src, uint8_t *__restrict__ esc, uint32_t E) {
    const uint32_t stride = gridDim.x * ZB;
    for (uint32_t p = blockIdx.x * ZB + threadIdx.x; p < E; p += stride) {
        const uint32_t a = src[p];
        if (a != p) esc[p] = esc[a];      // roots (literals) are never written here
    }
}

// ------------------------------------------------------------------ L2: tile path
constexpr int DT = 16384;               // escaped-stream bytes per resolve tile (8192 with twice the tiles per group: resolve 7 % faster, compose + chain 2x slower)
constexpr int DTH = 1024;               // threads of the tile kernels: 16 bytes per lane
constexpr int DGRP = 128;               // most tiles per chain group: a block of k_lzd_compose / k_lzd_emit walks its group's tiles in order,
                                        // 2.6 us each, so a stream gets about 512 groups if it has the tiles (see group_tiles)
__host__ inline uint32_t group_tiles(uint32_t n_tiles) {
    // 1 GiB = 65536 tiles in 512 groups of 128 (32 per group measured slower there: four times the group maps to scan); a 32 MiB
    // stream in groups of 128 is 16 blocks on 256 CUs, each 330 us in a row -- groups of 4 are 10 us and a few more scan rounds
    if (getenv("RSN_LZSS_DEC_GROUP128")) return DGRP;                      // A/B switch (read per call: the tests flip it)
    return std::max<uint32_t>(4, std::min<uint32_t>(DGRP, n_tiles / 512));
}
constexpr uint32_t D_LOC = 0x4000u;     // descriptor: position inside the tile
constexpr uint32_t D_EXT = 0x8000u;     // descriptor: position inside the previous tile's tail; otherwise the literal byte
constexpr uint32_t D_PAY = 0x3FFFu;
static_assert(DT / 16 == DTH && DT <= 16384, "one 16-byte span per lane; 14-bit payload");

// validate every token (lzss.go:349-350), record the largest back-pointer and, for every output
// tile, the item that produces its first byte: {input position, output position of the item}
__global__ __launch_bounds__(ZB) void k_lzd_tiles(const uint8_t *__restrict__ in, size_t n, const unsigned long long *__restrict__ blk_off,
                                                  uint2 *__restrict__ tile_info, uint32_t *__restrict__ maxptr, int *__restrict__ err) {
    __shared__ __attribute__((aligned(16))) uint32_t sw[(ZTILE + 2 * ZPAD) / 4 + 8];
    __shared__ uint32_t masks[ZB + 3], cdm[ZB + 3];
    __shared__ unsigned long long wsum[ZB / 64];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const size_t blk0 = (size_t)blockIdx.x * ZTILE;
    Span r;
    const int valid = stage_block(in, n, blk0, sw, masks, cdm, r);
    unsigned long long mine = 0;
    int e = 0;
    if (valid) { span_parse(sw, masks, cdm, tid, ZPAD + 16 * tid, valid, r); mine = r.out; e = r.err; }
    unsigned long long incl = mine;
    for (int d = 1; d < 64; d <<= 1) { const unsigned long long y = __shfl_up(incl, d); if (lane >= d) incl += y; }
    if (lane == 63) wsum[wv] = incl;
    __syncthreads();
    unsigned long long o = blk_off[blockIdx.x] + incl - mine;
    for (int k = 0; k < wv; k++) o += wsum[k];
    uint32_t mp = 0;
    if (valid && !r.err) {
        const uint32_t ipos0 = (uint32_t)(blk0 + 16 * (size_t)tid);
        unsigned long long before = 0;
#pragma unroll
        for (int k = 0; k < 4; k++) {
            if ((uint32_t)k < r.ntok) {
                const unsigned long long oo = o + (uint32_t)__builtin_popcount(r.lit & ((1u << r.tj[k]) - 1u)) + before;
                const uint32_t ptr = r.tptr[k], len = r.tlen[k];
                if (ptr > oo || len > ptr) e = 1;           // absolutePointer >= 0 and the slice ends inside out (lzss.go:349-350)
                else if (len) {
                    mp = max(mp, ptr);
                    for (unsigned long long t = (oo + DT - 1) / DT; t * DT < oo + len; t++) tile_info[t] = make_uint2(ipos0 + r.tj[k], (uint32_t)oo);
                }
                before += len;
            }
        }
        const unsigned long long first = (o + DT - 1) / DT * DT;              // first tile boundary at or after the span's output
        if (first < o + mine) {                                              // rare: some literal of the span may be a tile's first byte
#pragma unroll
            for (int j = 0; j < 16; j++) {
                if ((r.lit >> j) & 1) {
                    unsigned long long oo = o + (uint32_t)__builtin_popcount(r.lit & ((1u << j) - 1u));
#pragma unroll
                    for (int k = 0; k < 4; k++) oo += r.tj[k] < (uint32_t)j ? r.tlen[k] : 0u;
                    if (oo % DT == 0) tile_info[oo / DT] = make_uint2(ipos0 + j, (uint32_t)oo);
                }
            }
        }
    }
    for (int d = 32; d; d >>= 1) mp = max(mp, (uint32_t)__shfl_down(mp, d));
    if (lane == 0 && mp > __atomic_load_n(maxptr, __ATOMIC_RELAXED)) atomicMax(maxptr, mp);   // same-address atomics serialise: only raise it
    if (e) atomicOr(err, 1);
}

// ------------------------------------------------------------------ L1': the front end in one pass over the tokens (r03)
// k_lzd_count + scan + k_lzd_tiles parse every token twice (0.6 + 0.75 ms per GiB of text) and cost two host round trips.  k_lzd_tiles
// needs the absolute output offsets for two things only, and both can do without a second parse:
//   * "the slice starts inside the decoded data" (ptr <= output so far, lzss.go:349): the counting pass knows every token's offset
//     RELATIVE to its 4 KB block, so it leaves need[b] = max(ptr - relative offset); the stream is valid iff need[b] <= blk_off[b] for every
//     block -- one compare per block after the scan.  (len <= ptr is local and checked at once.  Until that compare has been read back the
//     tile kernels run on a stream that may point before its own start: they index LDS tails of TL >= the largest pointer and never memory,
//     so the worst case is garbage in a buffer whose call returns RSN_ERR_FORMAT.)
//   * the item that produces the first byte of every output tile: the pass also leaves what each 16-byte span produces (16 bits), so a
//     tile's first item is found by a binary search over the block offsets, a scan of that block's 256 span outputs and a walk through
//     ONE span -- k_lzd_tilemap, one wavefront per tile.
// A span that produces 65535 bytes or more (a foreign stream with huge tokens) raises flags[5]: the host then runs k_lzd_tiles as before.
__global__ __launch_bounds__(ZB) void k_lzd_count2(const uint8_t *__restrict__ in, size_t n, unsigned long long *__restrict__ blk_len, uint16_t *__restrict__ span_out,
                                                   unsigned long long *__restrict__ need, uint32_t *__restrict__ maxptr, int *__restrict__ flags, uint32_t b0 = 0) {
    const uint32_t bx = blockIdx.x + b0;                                  // (b0: the first block of a slice of the stream, lzss_decode_sliced)
    __shared__ __attribute__((aligned(16))) uint32_t sw[(ZTILE + 2 * ZPAD) / 4 + 8];
    __shared__ uint32_t masks[ZB + 3], cdm[ZB + 3];
    __shared__ unsigned long long wsum[ZB / 64], wneed[ZB / 64];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    Span r;
    const int valid = stage_block(in, n, (size_t)bx * ZTILE, sw, masks, cdm, r);
    unsigned long long mine = 0;
    int e = 0;
    if (valid) { span_parse(sw, masks, cdm, tid, ZPAD + 16 * tid, valid, r); mine = r.out; if (r.err) e = 1; }
    span_out[(size_t)bx * ZB + tid] = (uint16_t)min(mine, 0xFFFFull);
    if (mine >= 0xFFFFull) e |= 4;
    if (__ballot(valid && mask_5c(r.w) != 0) && lane == 0 && __atomic_load_n(&flags[4], __ATOMIC_RELAXED) == 0) atomicOr(&flags[4], 1);
    unsigned long long incl = mine;
    for (int d = 1; d < 64; d <<= 1) { const unsigned long long y = __shfl_up(incl, d); if (lane >= d) incl += y; }
    if (lane == 63) wsum[wv] = incl;
    __syncthreads();
    unsigned long long o = incl - mine;                                    // output offset of the span's first item, relative to the block
    for (int k = 0; k < wv; k++) o += wsum[k];
    unsigned long long nd = 0;
    uint32_t mp = 0;
    if (valid && !r.err) {
        unsigned long long before = 0;
#pragma unroll
        for (int k = 0; k < 4; k++) {
            if ((uint32_t)k < r.ntok) {
                const unsigned long long oo = o + (uint32_t)__builtin_popcount(r.lit & ((1u << r.tj[k]) - 1u)) + before;
                const uint32_t ptr = r.tptr[k], len = r.tlen[k];
                if (len > ptr) e |= 2;                                   // the slice ends inside out (lzss.go:350)
                if (ptr > oo) nd = max(nd, (unsigned long long)ptr - oo);   // absolutePointer >= 0 iff the block starts at least this far into the output
                if (len) mp = max(mp, ptr);
                before += len;
            }
        }
    }
    for (int d = 32; d; d >>= 1) { nd = max(nd, (unsigned long long)__shfl_down(nd, d)); mp = max(mp, (uint32_t)__shfl_down(mp, d)); }
    if (lane == 0) wneed[wv] = nd;
    if (lane == 0 && mp > __atomic_load_n(maxptr, __ATOMIC_RELAXED)) atomicMax(maxptr, mp);   // same-address atomics serialise: only raise it
    if (e & 3) atomicOr(&flags[0], e & 3);                                  // 1: malformed token, 2: reference outside the decoded data
    if (e & 4) atomicOr(&flags[5], 1);
    __syncthreads();
    if (tid == 0) {
        blk_len[bx] = wsum[0] + wsum[1] + wsum[2] + wsum[3];
        need[bx] = max(max(wneed[0], wneed[1]), max(wneed[2], wneed[3]));
    }
}

__global__ __launch_bounds__(256) void k_lzd_check(const unsigned long long *__restrict__ need, const unsigned long long *__restrict__ blk_off, uint32_t n_cb, int *__restrict__ flags) {
    const uint32_t b = blockIdx.x * 256 + threadIdx.x;
    const bool bad = b < n_cb && need[b] > blk_off[b];
    if (__ballot(bad) && (threadIdx.x & 63) == 0) atomicOr(&flags[0], 2);
}

// tile_info[t] = {input position, output position} of the item that produces output byte t * DT: one wavefront per tile.
// Latency is all there is to it (65 536 tiles per GiB, a few dependent steps each), so the steps are few and wide: a 64-ary search of
// the block offsets, one load of the block's span outputs, one load of the 64 bytes around the span -- the walk through the span's
// items then reads its bytes out of that register with v_readlane, all lanes in step.
__global__ __launch_bounds__(256) void k_lzd_tilemap(const uint8_t *__restrict__ in, size_t n, const unsigned long long *__restrict__ blk_off, uint32_t n_cb,
                                                     const uint16_t *__restrict__ span_out, uint32_t n_tiles, uint2 *__restrict__ tile_info) {
    const uint32_t t = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (t >= n_tiles) return;
    const unsigned long long T = (unsigned long long)t * DT;
    // the last block whose first item starts at or before T (blocks that produce nothing share an offset: the last one wins, and it is
    // the one with output): [lo, hi) always holds it, 64 probes a step
    uint32_t lo = 0, hi = n_cb;
    while (hi - lo > 1) {
        const uint32_t span = hi - lo, stride = (span + 63) / 64;
        const uint32_t idx = lo + (uint32_t)lane * stride;                 // lane 0 probes lo itself: always <= T
        const bool le = idx < hi && blk_off[idx] <= T;
        const unsigned long long m = __ballot(le);                        // a prefix of the lanes (the offsets ascend)
        const uint32_t last = 63u - (uint32_t)__builtin_clzll(m);
        const uint32_t nlo = lo + last * stride;
        hi = min(hi, nlo + stride);
        lo = nlo;
    }
    const uint32_t b = lo;
    const unsigned long long boff = blk_off[b];
    const uint32_t rel = (uint32_t)(T - boff);                            // < the block's output (T < the stream's length)
    // the block's 256 span outputs, four per lane; the span whose items cover `rel`
    const uint2 pk = *reinterpret_cast<const uint2 *>(span_out + (size_t)b * ZB + 4 * lane);
    const uint32_t s0 = pk.x & 0xFFFF, s1 = pk.x >> 16, s2 = pk.y & 0xFFFF, s3 = pk.y >> 16, sum = s0 + s1 + s2 + s3;
    uint32_t incl = sum;
    for (int d = 1; d < 64; d <<= 1) { const uint32_t y = __shfl_up(incl, d); if (lane >= d) incl += y; }
    const uint32_t ex = incl - sum;
    const unsigned long long hm = __ballot(rel >= ex && rel < incl);       // exactly one lane (spans that produce nothing never qualify)
    if (!hm) { if (lane == 0) tile_info[t] = make_uint2(0u, 0u); return; } // (cannot happen for a consistent count; the resolve kernel then fails its own checks)
    uint32_t sp = 4 * lane, so = ex;                                       // every lane as if it were the one: span index inside the block, output offset of its first item
    if (rel >= so + s0) { so += s0; sp++; if (rel >= so + s1) { so += s1; sp++; if (rel >= so + s2) { so += s2; sp++; } } }
    const int src = __builtin_ctzll(hm);
    sp = (uint32_t)__builtin_amdgcn_readlane((int)sp, src);
    so = (uint32_t)__builtin_amdgcn_readlane((int)so, src);
    // the 64 bytes from 24 before the span (a token that began before it is at most 23 long) to 40 after its start (a token that begins in it ends there)
    const long long s_beg = (long long)b * ZTILE + 16ll * sp, w0 = s_beg - 24;
    const long long qa = w0 + lane;
    const uint32_t wbyte = qa >= 0 && qa < (long long)n ? in[qa] : 0u;
    auto B = [&](long long q) -> uint32_t {                                // byte q of the stream, q wave-uniform and inside the window (0 beyond the stream)
        return (uint32_t)__builtin_amdgcn_readlane((int)wbyte, (int)(q - w0));
    };
    auto tok = [&](long long j, uint32_t &len, uint32_t &tl) -> bool {    // the token whose '<' is byte j (as parse_tok)
        long long k = j + 1;
        unsigned long long v = 0; int nd = 0;
        while (k < (long long)n && nd < 10 && B(k) >= '0' && B(k) <= '9') { v = v * 10 + (B(k) - '0'); k++; nd++; }
        if (nd == 0 || k >= (long long)n || B(k) != ',' || v > 0xFFFFFFFFull) return false;
        k++; v = 0; nd = 0;
        while (k < (long long)n && nd < 10 && B(k) >= '0' && B(k) <= '9') { v = v * 10 + (B(k) - '0'); k++; nd++; }
        if (nd == 0 || k >= (long long)n || B(k) != '>' || v > 0xFFFFFFFFull) return false;
        len = (uint32_t)v; tl = (uint32_t)(k + 1 - j);
        return true;
    };
    long long pos = s_beg;                                                 // first position of the span that is not inside a token begun before it (skip_open_token)
    for (int back = 1; back < MAXTOK && back <= s_beg; back++) {
        const uint32_t c = B(s_beg - back);
        if (c == '>') break;
        if (c == '<') { uint32_t len, tl; if (tok(s_beg - back, len, tl)) pos = max(pos, s_beg - back + (long long)tl); break; }
    }
    const long long lim = min(s_beg + 16, (long long)n);
    unsigned long long o = boff + so;
    uint2 res = make_uint2((uint32_t)pos, (uint32_t)o);
    while (pos < lim) {
        uint32_t len = 1, adv = 1;
        if (B(pos) == '<') { if (!tok(pos, len, adv)) break; }
        if (o + len > T) { res = make_uint2((uint32_t)pos, (uint32_t)o); break; }   // (o <= T here: the items before it end at or before T)
        o += len; pos += adv;
    }
    if (lane == 0) tile_info[t] = res;
}

// RUN TILES (r03).  A tile that holds nothing but a few tokens -- W-periodic data is four tokens per tile, long runs and repeated blocks
// look the same -- needs no per-position work at all: its descriptors are a handful of runs "positions x0.. come from positions r0.. of
// the previous tile's tail", found by resolving the tokens against each other as INTERVALS (one lane, a few dozen steps).  Such a tile
// is recognised by its input alone -- at most RT_BYTES bytes of tokens for 16 KiB of output -- and resolved by k_lzd_runs, one wavefront
// per tile: rt_cnt[tile] = number of runs, rt_runs[tile][..] = x0 | r0 << 16 (a run ends where the next begins), no 32 KB of descriptors;
// k_lzd_resolve returns at once for it, k_lzd_compose and k_lzd_emit expand the runs on the fly.  Config 3: 2 bytes per output byte
// written and read back -> none.  (The same path INSIDE k_lzd_resolve, taken after its parse, cost text 10 % of that kernel through
// register allocation alone, whether taken or not.)
constexpr int RT_ITEMS = 32, RT_RUNS = 64;                                // most tokens of a run tile, most runs after resolving them
constexpr int RT_BYTES = 448;                                             // most input bytes of a run tile (32 tokens of up to 13 bytes, and room to see a 33rd)
struct ResolveArgs { const uint8_t *in; size_t n; const uint2 *tile_info; uint32_t n_tiles, E, TL; uint16_t *desc; int *fallback;
                     uint32_t *rt_cnt, *rt_runs; };  // per tile: 0 = descriptors, else the number of runs; RT_RUNS words per tile

__global__ __launch_bounds__(64) void k_lzd_runs(const uint8_t *__restrict__ in, size_t n, const uint2 *__restrict__ tile_info, uint32_t n_tiles, uint32_t E, uint32_t TL,
                                                 uint32_t *__restrict__ rt_cnt, uint32_t *__restrict__ rt_runs) {
    __shared__ __attribute__((aligned(16))) uint32_t s_in32[(RT_BYTES + MAXTOK + 8 + 32) / 4];   // (a dword view for parse_tok_w: zero past the data, readable 28 bytes past a token's '<')
    uint8_t *s_in = reinterpret_cast<uint8_t *>(s_in32);
    __shared__ uint32_t s_tok[RT_ITEMS + 1];                              // x | (ptr - 1) << 16 in stream order; then the end of the last token
    __shared__ uint32_t s_runs[RT_RUNS];
    __shared__ uint32_t s_nr;
    const int lane = threadIdx.x;
    const uint32_t k = blockIdx.x, ts = k * DT;
    const int tlen = (int)min((uint32_t)DT, E - ts);
    const uint2 info = tile_info[k];
    const size_t in0 = info.x, in1 = k + 1 < n_tiles ? (size_t)tile_info[k + 1].x : n;   // the tile's items begin in [in0, in1)
    if (in1 <= in0 || in1 - in0 > (size_t)RT_BYTES) { if (lane == 0) rt_cnt[k] = 0; return; }
    const uint32_t nb = (uint32_t)(in1 - in0), stage = min((uint32_t)(n - in0), nb + (uint32_t)MAXTOK);
    for (uint32_t i = lane; i < RT_BYTES + MAXTOK + 8 + 32; i += 64) s_in[i] = i < stage ? in[in0 + i] : 0;
    if (lane == 0) s_nr = 0;
    __syncthreads();
    // every '<' among the first nb bytes starts a token (a literal '<' is escaped); the tile is a run tile iff those tokens follow each
    // other without a byte between them from byte 0 on -- then nothing but tokens produces the tile
    uint32_t ntok = 0;
    bool ok = true;
    int x = (int)((long long)info.y - (long long)ts);                     // output position of the first item, relative to the tile: <= 0
    if (lane == 0) {
        uint32_t pos = 0;
        while (x < tlen && ok) {                                            // (the token that reaches the tile's end may begin at byte nb: it is staged too)
            if (pos >= stage || s_in[pos] != '<' || ntok >= (uint32_t)RT_ITEMS) { ok = false; break; }
            const Tok t = parse_tok_w(s_in32, (int)pos);                  // (the dword parser: the byte loop's dependent LDS reads were most of this kernel on periodic data, 0.15 ms per GiB)
            if (!t.ok || pos + t.tl > stage || t.ptr == 0 || t.ptr > TL || t.len > t.ptr) { ok = false; break; }   // (validated already; a run tile needs ptr <= TL like every tile of the tile path)
            if (t.len) {                                                    // (a zero-length token produces nothing)
                const int xs = max(x, 0);
                if (x + (int)t.len > 0 && xs < tlen) s_tok[ntok++] = (uint32_t)xs | ((t.ptr - 1u) << 16);
                x += (int)t.len;
            }
            pos += t.tl;
        }
        ok = ok && ntok >= 1 && (s_tok[0] & 0xFFFFu) == 0 && x >= tlen;   // the tokens cover the tile from its first byte to its last
        if (ok) {
            s_tok[ntok] = (uint32_t)tlen;
            // Every token copies [x0 - ptr, x0 - ptr + len): what lies before the tile is the previous tail (r = TL + position), what lies
            // inside it is made of runs already resolved -- the tokens before it cover every earlier position -- and is copied piecewise.
            uint32_t nr = 0;
            auto push = [&](uint32_t x0, uint32_t r0) {
                if (nr && r0 == (s_runs[nr - 1] >> 16) + (x0 - (s_runs[nr - 1] & 0xFFFFu))) return;   // continues the run before it
                if (nr >= (uint32_t)RT_RUNS) { ok = false; return; }
                s_runs[nr++] = x0 | (r0 << 16);
            };
            for (uint32_t it = 0; it < ntok && ok; it++) {
                const uint32_t x0 = s_tok[it] & 0xFFFFu, ptr = (s_tok[it] >> 16) + 1u, xe = min(s_tok[it + 1] & 0xFFFFu, (uint32_t)tlen);
                uint32_t out = x0, left = xe - x0;
                int pos = (int)x0 - (int)ptr;
                if (pos < 0) { const uint32_t n1 = min(left, (uint32_t)(-pos)); push(out, (uint32_t)((int)TL + pos)); out += n1; pos += (int)n1; left -= n1; }
                uint32_t q = 0;
                while (left && ok) {                                       // pos >= 0: inside the tile, covered by the runs so far
                    while (q + 1 < nr && (s_runs[q + 1] & 0xFFFFu) <= (uint32_t)pos) q++;
                    const uint32_t rx = s_runs[q] & 0xFFFFu, rend = q + 1 < nr ? (s_runs[q + 1] & 0xFFFFu) : out;   // (the last run ends where this token is writing)
                    if ((uint32_t)pos < rx || (uint32_t)pos >= rend) { ok = false; break; }
                    const uint32_t n1 = min(left, rend - (uint32_t)pos);
                    push(out, (s_runs[q] >> 16) + ((uint32_t)pos - rx));
                    out += n1; pos += (int)n1; left -= n1;
                }
            }
            if (ok) s_nr = nr;
        }
    }
    __syncthreads();
    const uint32_t nr = s_nr;
    if ((uint32_t)lane < nr) rt_runs[(size_t)k * RT_RUNS + lane] = s_runs[lane];
    if (lane == 0) rt_cnt[k] = nr;
}


__global__ __launch_bounds__(DTH, 8) void k_lzd_resolve(ResolveArgs a) {   // 8 waves per SIMD: two blocks per CU
    __shared__ __attribute__((aligned(16))) uint32_t sw[(DT + 128) / 4 + 8];
    __shared__ __attribute__((aligned(16))) uint16_t sd[DT];
    __shared__ uint32_t masks[DTH + 3], cdm[DTH + 3];
    __shared__ uint32_t s_part[DTH / 64];
    __shared__ uint32_t s_wlast[DTH / 64];
    // Item marks (phases A and B only): 0 = no item starts here; bit 15 = a RUN of literals starts here, low 15 bits = (staged byte index -
    // output position) mod 2^15 -- the byte of position y of the run is stage[(y + mark) mod 2^15]; bit 14 = a token, low 14 bits = ptr - 1.
    // (r02 marked every literal with its own byte: 16 address computations and 16 LDS stores per lane where a span holds one or two runs.)
    constexpr uint32_t NONE = 0u, M_LIT = 0x8000u, M_TOK = 0x4000u;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const uint32_t k = blockIdx.x, ts = k * DT;
    const uint32_t is_run_tile = a.rt_cnt ? a.rt_cnt[k] : 0u;             // (arrives together with the tile's record below)
    const int tlen = (int)min((uint32_t)DT, a.E - ts);
    const uint2 info = a.tile_info[k];
    const size_t in0 = info.x;
    const size_t in1 = k + 1 < a.n_tiles ? min(a.n, (size_t)a.tile_info[k + 1].x + MAXTOK) : a.n;
    if (__builtin_amdgcn_readfirstlane((int)is_run_tile)) return;         // k_lzd_runs has resolved it (block-uniform: a scalar branch; before the staging loads: config 3 is nothing but such tiles)
    if (in1 - in0 > (size_t)(DT + 64)) { if (tid == 0) *a.fallback = 1; return; }   // zero-length tokens can stretch a tile's input without bound
    const size_t inA = in0 & ~(size_t)15;
    const int lo = (int)(in0 - inA), hi = lo + (int)(in1 - in0);
    for (int v = tid; v * 16 < hi + 32; v += DTH) {                       // 32 bytes past the range: a token near its end is read whole
        const size_t P = inA + 16 * (size_t)v;
        uint4 x = {0, 0, 0, 0};
        if (P + 16 <= a.n) x = *reinterpret_cast<const uint4 *>(a.in + P);
        else if (P < a.n) { uint32_t w[4] = {0, 0, 0, 0}; for (int q = 0; q < 16 && P + q < a.n; q++) w[q >> 2] |= (uint32_t)a.in[P + q] << (8 * (q & 3)); x = {w[0], w[1], w[2], w[3]}; }
        reinterpret_cast<uint4 *>(sw)[v] = x;
    }
    for (int v = tid; v < DT / 8; v += DTH) reinterpret_cast<uint4 *>(sd)[v] = make_uint4(0u, 0u, 0u, 0u);
    __syncthreads();
    const int TL = (int)a.TL;
    // ---- A: every item marks the output position it starts at -- a literal with its byte, a token with D_LOC | (ptr - 1).
    // every token here has 1 <= len <= ptr <= DT (k_lzd_tiles checked it), so 32-bit offsets cannot overflow
    int run = (int)((long long)info.y - (long long)ts);                   // output offset (relative to the tile) of the first staged item: <= 0
    for (int base = 0; lo + 16 * base < hi; base += DTH) {
        const int sp = base + tid, sbyte = lo + 16 * sp;
        const int valid = hi - sbyte >= 16 ? 16 : hi - sbyte > 0 ? hi - sbyte : 0;
        Span r;
        uint32_t m = 0, cd = 0, lead = 0, lead_cd = 0;
        if (valid) { load_span(sw, sbyte, r.w); m = ltgt_masks(r.w); cd = c_mask(r.w); }
        if (tid < 2 && base) { lead = masks[DTH + tid]; lead_cd = cdm[DTH + tid]; }   // the last two spans of the previous round
        __syncthreads();
        masks[tid + 2] = m; cdm[tid + 2] = cd;
        if (tid < 2) { masks[tid] = lead; cdm[tid] = lead_cd; }
        if (tid == 2) {                                                   // the span after the round's last (staged: 32 bytes past the range)
            uint32_t w[4] = {0, 0, 0, 0};
            const int nb = lo + 16 * (base + DTH);
            if (nb < hi + 16) load_span(sw, nb, w);
            masks[DTH + 2] = ltgt_masks(w); cdm[DTH + 2] = c_mask(w);
        }
        __syncthreads();
        uint32_t mine = 0;
        if (valid) { span_parse<true>(sw, masks, cdm, tid, sbyte, valid, r); mine = r.out; }
        uint32_t incl = mine;
        for (int d = 1; d < 64; d <<= 1) { const uint32_t y = __shfl_up(incl, d); if (lane >= d) incl += y; }
        if (lane == 63) s_part[wv] = incl;
        __syncthreads();
        int o = run + (int)(incl - mine);
        uint32_t tot;
        {   // the sixteen wavefront sums: one scan by shuffles instead of sixteen LDS reads and adds per lane
            uint32_t ws = lane < DTH / 64 ? s_part[lane] : 0u;
            for (int d = 1; d < DTH / 64; d <<= 1) { const uint32_t y = __shfl_up(ws, d); if (lane >= d) ws += y; }
            tot = (uint32_t)__builtin_amdgcn_readlane((int)ws, DTH / 64 - 1);
            if (wv) o += (int)(uint32_t)__builtin_amdgcn_readlane((int)ws, wv - 1);
        }
        if (valid && o < tlen) {
            for (uint32_t st = r.lit & ~(r.lit << 1); st; st &= st - 1) {   // the first literal of every run of literals in the span
                const int j = __builtin_ctz(st);
                int x = o + __builtin_popcount(r.lit & ((1u << j) - 1u));
#pragma unroll
                for (int q = 0; q < 4; q++) x += r.tj[q] < (uint32_t)j ? (int)r.tlen[q] : 0;
                if (x >= 0 && x < tlen) sd[x] = (uint16_t)(M_LIT | ((uint32_t)(sbyte + j - x) & 0x7FFFu));
            }
            int before = 0;
#pragma unroll
            for (int q = 0; q < 4; q++) {
                if ((uint32_t)q < r.ntok) {
                    const int x0 = o + __builtin_popcount(r.lit & ((1u << r.tj[q]) - 1u)) + before;
                    const uint32_t ptr = r.tptr[q], len = r.tlen[q];
                    const int xs = max(x0, 0);                            // a token that began in the previous tile: its first byte HERE
                    if (len && x0 + (int)len > 0 && xs < tlen) sd[xs] = (uint16_t)(M_TOK | (ptr - 1u));
                    before += (int)len;
                }
            }
        }
        run += (int)tot;
    }
    __syncthreads();
    // ---- B: a lane owns 16 consecutive output positions.  The item a position belongs to is the last mark at or before it
    // (fill forward: a scan with "rightmost mark" inside the lane, across the wavefront, across the block), and a token's
    // bytes get their descriptor from the position and the token's back-pointer alone.
    // (r04) Marks, positions and descriptors are 16-bit: the arithmetic runs on PAIRS of positions in packed halves (v_pk_add_u16 /
    // v_pk_sub_u16 / v_pk_ashrrev_i16 and bit selects), branch-free -- it had been sixteen copies of a two-way branch per lane,
    // ~530 vector and ~340 scalar instructions; position j's literal flag lives at bit (j >> 1) + 16 (j & 1) of litm.
    const int xb = 16 * tid;
    uint32_t pk[8];
    {
        const uint4 v0 = reinterpret_cast<const uint4 *>(sd + xb)[0], v1 = reinterpret_cast<const uint4 *>(sd + xb)[1];
        pk[0] = v0.x; pk[1] = v0.y; pk[2] = v0.z; pk[3] = v0.w; pk[4] = v1.x; pk[5] = v1.y; pk[6] = v1.z; pk[7] = v1.w;
    }
    uint32_t mylast = NONE;                                               // the lane's last mark: the last nonzero dword's high half, else its low half
    {
        uint32_t lastw = 0;
#pragma unroll
        for (int k = 0; k < 8; k++) lastw = pk[k] ? pk[k] : lastw;
        mylast = (lastw >> 16) ? (lastw >> 16) : lastw;
    }
    uint32_t incl = mylast;
    for (int d = 1; d < 64; d <<= 1) { const uint32_t y = __shfl_up(incl, d); if (lane >= d && incl == NONE) incl = y; }
    if (lane == 63) s_wlast[wv] = incl;
    uint32_t cur = __shfl_up(incl, 1);
    if (lane == 0) cur = NONE;
    __syncthreads();
    {   // what the wavefronts before this one leave open: the same scan over their (at most 16) last marks
        uint32_t wl = lane < DTH / 64 ? s_wlast[lane] : NONE;
        for (int d = 1; d < DTH / 64; d <<= 1) { const uint32_t y = __shfl_up(wl, d); if (lane >= d && wl == NONE) wl = y; }
        const uint32_t carry = wv ? (uint32_t)__builtin_amdgcn_readlane((int)wl, wv - 1) : NONE;
        if (cur == NONE) cur = carry;
    }
    typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));
    typedef short i16x2 __attribute__((ext_vector_type(2)));
    auto pk_add = [](uint32_t a, uint32_t b) { return __builtin_bit_cast(uint32_t, __builtin_bit_cast(u16x2, a) + __builtin_bit_cast(u16x2, b)); };
    auto pk_sub = [](uint32_t a, uint32_t b) { return __builtin_bit_cast(uint32_t, __builtin_bit_cast(u16x2, a) - __builtin_bit_cast(u16x2, b)); };
    auto pk_sign = [](uint32_t a) { return __builtin_bit_cast(uint32_t, __builtin_bit_cast(i16x2, a) >> (i16x2)15); };   // 0xFFFF in a half whose bit 15 is set
    auto bsel = [](uint32_t m, uint32_t a, uint32_t b) { return (m & a) | (~m & b); };                                   // v_bfi_b32
    uint32_t dp[8], lm[8];                                                // descriptors and "is a literal" masks, two positions a word
    uint32_t litm = 0;
    const uint8_t *sb = reinterpret_cast<const uint8_t *>(sw);
    {
        const uint32_t c_ext = ((uint32_t)TL + D_EXT) & 0xFFFFu, sel_ext = c_ext | (c_ext << 16), sel_loc = D_LOC | (D_LOC << 16);
        const uint32_t xp0 = (uint32_t)xb | ((uint32_t)(xb + 1) << 16);
#pragma unroll
        for (int k = 0; k < 8; k++) {
            const uint32_t lo = pk[k] & 0xFFFFu, hi = pk[k] >> 16;
            const uint32_t c0 = lo != NONE ? lo : cur, c1 = hi != NONE ? hi : c0;   // the marks that cover positions 2k and 2k + 1
            cur = c1;
            const uint32_t C = c0 | (c1 << 16);
            const uint32_t xp = xp0 + (uint32_t)k * 0x00020002u;          // the two positions (below 16384: no carry between the halves)
            const uint32_t L = pk_sign(C);                                // M_LIT is bit 15
            const uint32_t d_lit = pk_add(xp, C) & 0x7FFF7FFFu;           // for now: the staged index of the literal's byte
            const uint32_t ptr = (C & (D_PAY | (D_PAY << 16))) + 0x00010001u;
            const uint32_t q = pk_sub(xp, ptr);                           // position of the source inside the tile, or negative: before it
            const uint32_t S = pk_sign(q);
            const uint32_t d_tok = pk_add(q, bsel(S, sel_ext, sel_loc));  // D_LOC | q, or D_EXT | (TL + q)
            dp[k] = bsel(L, d_lit, d_tok);                                // (a position no item covers cannot occur in a validated stream)
            lm[k] = L;
            litm |= (L & 0x00010001u) << k;
        }
        // the positions behind the tile's end take no part (the last tile of a stream)
        const uint32_t nv = (uint32_t)min(max(tlen - xb, 0), 16);
        const uint32_t vm = ((1u << ((nv + 1) >> 1)) - 1u) | (((1u << (nv >> 1)) - 1u) << 16);
        litm &= vm;
    }
    if (__ballot(litm != 0)) {   // the literals' bytes out of the stage, all reads issued together (a slot that holds no literal reads some byte of the block's LDS with everybody else)
        uint32_t by[16];
#pragma unroll
        for (int j = 0; j < 16; j++) { const uint32_t a2 = dp[j >> 1] & lm[j >> 1] & 0x7FFF7FFFu; by[j] = sb[(j & 1) ? a2 >> 16 : a2 & 0xFFFFu]; }   // (not a literal: byte 0)
#pragma unroll
        for (int k = 0; k < 8; k++) dp[k] = bsel(lm[k], by[2 * k] | (by[2 * k + 1] << 16), dp[k]);
    }
    if (tlen != DT) {   // the stream's last tile: what lies behind its end is nobody's -- a resolved zero, or it would chase pointers for nothing
        const uint32_t nv = (uint32_t)min(max(tlen - xb, 0), 16);
#pragma unroll
        for (int k = 0; k < 8; k++) dp[k] &= (nv > 2u * k ? 0xFFFFu : 0u) | (nv > 2u * k + 1u ? 0xFFFF0000u : 0u);
    }
    auto write_back = [&]() {
        reinterpret_cast<uint4 *>(sd + xb)[0] = make_uint4(dp[0], dp[1], dp[2], dp[3]);
        reinterpret_cast<uint4 *>(sd + xb)[1] = make_uint4(dp[4], dp[5], dp[6], dp[7]);
    };
    __syncthreads();                                                      // every lane has read its marks
    write_back();
    __syncthreads();
    // ---- C: in-tile pointer jumping, the lane's 16 descriptors in registers (packed in pairs, as B left them); only the unresolved
    // ones -- bits 15:14 == 01: D_LOC -- read LDS, two hops a round.
    // (A reader may see another lane's descriptor before or after that lane's update of the same round: both name the same byte.
    //  Giving lane t the positions t, t + DTH, ... instead -- consecutive lanes on consecutive addresses -- measured slower.)
    auto loc_halves = [&](uint32_t d) { return pk_sign((d << 1) & ~d); }; // 0xFFFF in a half that holds a D_LOC descriptor
    const uint8_t *sdb = reinterpret_cast<const uint8_t *>(sd);
    for (;;) {
        uint32_t um[8], any = 0;
#pragma unroll
        for (int k = 0; k < 8; k++) { um[k] = loc_halves(dp[k]); any |= um[k]; }
        if (any) {
#pragma unroll
            for (int hop = 0; hop < 2; hop++) {                           // all 16 reads of a hop are issued together
                if (hop && !__ballot(any != 0)) break;
                uint32_t w[16];                                           // (a resolved slot reads sd[0] with everybody else: a broadcast)
#pragma unroll
                for (int k = 0; k < 8; k++) {
                    const uint32_t a2 = (dp[k] & um[k] & (D_PAY | (D_PAY << 16))) << 1;   // byte offsets of the two sources
                    w[2 * k] = *reinterpret_cast<const uint16_t *>(sdb + (a2 & 0xFFFFu));
                    w[2 * k + 1] = *reinterpret_cast<const uint16_t *>(sdb + (a2 >> 16));
                }
                any = 0;
#pragma unroll
                for (int k = 0; k < 8; k++) {
                    dp[k] = bsel(um[k], w[2 * k] | (w[2 * k + 1] << 16), dp[k]);
                    um[k] = loc_halves(dp[k]); any |= um[k];
                }
            }
            write_back();
        }
        if (!__syncthreads_or(any != 0)) break;
    }
    for (int v = tid; v * 8 < tlen; v += DTH) st16<(RSN_NT_MASK & 32) != 0>(reinterpret_cast<uint4 *>(a.desc + ts) + v, reinterpret_cast<const uint4 *>(sd)[v]);
}

// C_g = the tail map of the group's last tile expressed in the tail that precedes the group
// descriptor of position x of a run tile: the last run that begins at or before x (runs[0] begins at 0)
__device__ __forceinline__ uint32_t run_desc(const uint32_t *runs, uint32_t nr, uint32_t x) {
    uint32_t lo = 0, hi = nr;                                              // runs[lo].x0 <= x < runs[hi].x0
    while (hi - lo > 1) { const uint32_t mid = (lo + hi) >> 1; if ((runs[mid] & 0xFFFFu) <= x) lo = mid; else hi = mid; }
    return D_EXT | (((runs[lo] >> 16) + (x - (runs[lo] & 0xFFFFu))) & D_PAY);
}

__global__ __launch_bounds__(DTH) void k_lzd_compose(const uint16_t *__restrict__ desc, uint32_t TL, uint32_t dgrp, uint16_t *__restrict__ comp,
                                                     const uint32_t *__restrict__ rt_cnt, const uint32_t *__restrict__ rt_runs) {
    extern __shared__ __attribute__((aligned(16))) uint8_t dsm[];
    __shared__ uint32_t s_rt[RT_RUNS];
    uint16_t *cur = reinterpret_cast<uint16_t *>(dsm), *nxt = cur + TL;
    const size_t k0 = (size_t)blockIdx.x * dgrp;
    auto tail_entry = [&](size_t k, uint32_t nr, uint32_t j) -> uint32_t {   // the tile's descriptor of position DT - TL + j
        return nr ? run_desc(s_rt, nr, DT - TL + j) : desc[k * DT + DT - TL + j];
    };
    uint32_t nr = rt_cnt ? rt_cnt[k0] : 0u;
    if (threadIdx.x < nr) s_rt[threadIdx.x] = rt_runs[k0 * RT_RUNS + threadIdx.x];
    __syncthreads();
    for (uint32_t j = threadIdx.x; j < TL; j += DTH) cur[j] = (uint16_t)tail_entry(k0, nr, j);
    __syncthreads();
    // (r04) A step used to wait twice for memory -- the tile's kind, then its tail descriptors -- 2.7 us of which the composition is a
    // fraction: the kinds of the whole group are fetched up front, and the NEXT tile's tail (windows up to 4096: four entries a lane) is
    // asked for before this tile's is composed -- unless that tile is a run tile, which has no descriptors (config 3 is nothing else:
    // fetching them regardless cost it 0.3 ms, r04 section 8).
    __shared__ uint8_t s_kind[DGRP];                                     // the number of runs of every tile of the group (0: descriptors)
    static_assert(RT_RUNS <= 255, "a tile's run count fits a byte");
    // A lane composes FOUR CONSECUTIVE entries (windows up to 4096; larger ones go round again): a descriptor tile's come as one
    // 8-byte load, a run tile's usually lie in one run -- one bisection, and the four entries of the map in hand are consecutive
    // too: three dwords and a funnel shift instead of four bisections and four 2-byte gathers (the step is bound by the LDS).
    const bool pf_on = TL <= 4 * DTH;
    for (uint32_t t = threadIdx.x; t < dgrp; t += DTH) s_kind[t] = (uint8_t)min(rt_cnt ? rt_cnt[k0 + t] : 0u, 255u);
    __syncthreads();
    const uint32_t j4 = 4 * threadIdx.x;
    uint2 pf = {0, 0};
    auto fetch = [&](size_t k) { if (j4 < TL) pf = *reinterpret_cast<const uint2 *>(desc + k * DT + DT - TL + j4); };   // (TL, DT multiples of 256: 8-byte aligned)
    uint32_t rt_pf = 0;                                                   // ... and a run tile's runs come a step ahead in the same way
    if (pf_on && dgrp > 1 && s_kind[1] == 0) fetch(k0 + 1);
    if (dgrp > 1 && threadIdx.x < s_kind[1]) rt_pf = rt_runs[(k0 + 1) * RT_RUNS + threadIdx.x];
    auto through = [&](uint32_t v) -> uint32_t { return (v & D_EXT) ? cur[v & D_PAY] : (v & 0xFFFFu); };   // one entry through the map in hand
    for (uint32_t t = 1; t < dgrp; t++) {
        nr = s_kind[t];                                                   // (a run tile has at most RT_RUNS = 64 runs: the byte IS the count)
        if (threadIdx.x < nr) s_rt[threadIdx.x] = rt_pf;
        const uint2 mine_v = pf;
        if (pf_on && t + 1 < dgrp && s_kind[t + 1] == 0) fetch(k0 + t + 1);
        if (t + 1 < dgrp && threadIdx.x < s_kind[t + 1]) rt_pf = rt_runs[(k0 + t + 1) * RT_RUNS + threadIdx.x];
        __syncthreads();
        for (uint32_t j0 = j4; j0 < TL; j0 += 4 * DTH) {
            uint32_t o0, o1;                                              // entries j0, j0 + 1 | j0 + 2, j0 + 3
            if (nr) {
                const uint32_t x = DT - TL + j0;
                uint32_t lo = 0, hi = nr;
                while (hi - lo > 1) { const uint32_t mid = (lo + hi) >> 1; if ((s_rt[mid] & 0xFFFFu) <= x) lo = mid; else hi = mid; }
                const uint32_t r = ((s_rt[lo] >> 16) + (x - (s_rt[lo] & 0xFFFFu))) & D_PAY;
                const uint32_t nxt_x = lo + 1 < nr ? (s_rt[lo + 1] & 0xFFFFu) : 0xFFFFFFFFu;
                if (nxt_x >= x + 4 && r + 4 <= TL) {                      // one run: the map's entries r .. r + 3
                    const uint32_t *c32 = reinterpret_cast<const uint32_t *>(cur) + (r >> 1);
                    const uint32_t d0 = c32[0], d1 = c32[1], d2 = (r & 1) ? c32[2] : 0u, sh = 16u * (r & 1u);
                    o0 = __builtin_amdgcn_alignbit(d1, d0, sh); o1 = __builtin_amdgcn_alignbit(d2, d1, sh);
                } else {
                    const uint32_t e0 = through(run_desc(s_rt, nr, x)), e1 = through(run_desc(s_rt, nr, x + 1));
                    const uint32_t e2 = through(run_desc(s_rt, nr, x + 2)), e3 = through(run_desc(s_rt, nr, x + 3));
                    o0 = e0 | (e1 << 16); o1 = e2 | (e3 << 16);
                }
            } else {
                uint2 v = mine_v;
                if (!pf_on || j0 != j4) v = *reinterpret_cast<const uint2 *>(desc + (k0 + t) * DT + DT - TL + j0);
                o0 = through(v.x & 0xFFFFu) | (through(v.x >> 16) << 16);
                o1 = through(v.y & 0xFFFFu) | (through(v.y >> 16) << 16);
            }
            *reinterpret_cast<uint2 *>(nxt + j0) = make_uint2(o0, o1);
        }
        __syncthreads();
        uint16_t *sw = cur; cur = nxt; nxt = sw;
    }
    for (uint32_t j = threadIdx.x; j < TL; j += DTH) comp[(size_t)blockIdx.x * TL + j] = cur[j];
}

// gtail[g] = the bytes of the tail that ends group g: one block walks the groups in order

// The same without the serial walk: an inclusive scan of the group maps under composition.  After the round with stride d, map g
// is expressed in the tail that precedes group g - 2d + 1 (or in nothing at all: only literals left); ceil(log2(links)) rounds of
// one gather per entry, every group in parallel, then the literals are the tails' bytes (512 links per GiB: 9 rounds of 4 MB
// instead of 512 dependent steps of one block).
__global__ __launch_bounds__(256) void k_lzd_mapscan(const uint16_t *__restrict__ src, uint16_t *__restrict__ dst, uint32_t TL, uint32_t d) {
    const uint32_t g = blockIdx.x;
    const uint16_t *m = src + (size_t)g * TL, *before = src + (size_t)(g >= d ? g - d : 0) * TL;
    for (uint32_t j = threadIdx.x; j < TL; j += 256) {
        uint32_t v = m[j];
        if (g >= d && (v & D_EXT)) v = before[v & D_PAY];
        dst[(size_t)g * TL + j] = (uint16_t)v;
    }
}
__global__ __launch_bounds__(256) void k_lzd_map_bytes(const uint16_t *__restrict__ maps, uint32_t TL, uint8_t *__restrict__ gtail) {
    const size_t base = (size_t)blockIdx.x * TL;
    for (uint32_t j = threadIdx.x; j < TL; j += 256) { const uint32_t v = maps[base + j]; gtail[base + j] = (v & D_EXT) ? 0 : (uint8_t)v; }   // (a reference before the stream: validated away)
}

// the bytes of every tile of a group, tile after tile: a literal, or a byte of the previous tile's tail
// Also leaves the unescape stage its per-block summaries (k_une_summary's output) while the bytes
// are still in registers: one pass over the escaped stream less.
// 0xFF in every byte of w that equals the byte `c` (exact per byte)
__device__ __forceinline__ uint32_t emit_bytes_equal(uint32_t w, uint32_t c) {
    const uint32_t t = w ^ (c * 0x01010101u);
    const uint32_t z = ~(((t & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | t | 0x7F7F7F7Fu);
    return (z >> 7) * 0xFFu;
}

// map_ff: the stream holds no 5C at all, so unescaping is the byte map FF -> '<' (lzss.go:391-406, '<' never occurs in the escaped
// stream): applied to the stored bytes here, `esc` is the caller's output buffer and no unescape pass follows (summ unused).
__global__ __launch_bounds__(DTH, 8) void k_lzd_emit(const uint16_t *__restrict__ desc, uint32_t TL, uint32_t n_tiles, uint32_t dgrp, uint32_t E,   // (64 registers: two blocks per CU)
                                                  const uint8_t *__restrict__ gtail, uint8_t *__restrict__ esc, uint8_t *__restrict__ summ, int map_ff,
                                                  const uint32_t *__restrict__ rt_cnt, const uint32_t *__restrict__ rt_runs) {
    extern __shared__ __attribute__((aligned(16))) uint8_t dsm[];
    __shared__ uint32_t s_last[2][DT / ZTILE];                           // per ZTILE block of the tile: index past its last non-5C byte (by tile parity)
    __shared__ uint32_t s_rt[2][RT_RUNS];                                 // a run tile's runs (by tile parity)
    uint8_t *prev = dsm, *nxt = dsm + TL;
    const uint32_t g = blockIdx.x;
    if (threadIdx.x < 2 * DT / ZTILE) (&s_last[0][0])[threadIdx.x] = 0;
    for (uint32_t j = threadIdx.x; j < TL; j += DTH) prev[j] = g ? gtail[(size_t)(g - 1) * TL + j] : 0;
    const uint32_t x0 = threadIdx.x * 16;
    // (r04) the kinds of the group's tiles up front -- the number of runs, 0 for a tile with descriptors -- and a run tile's runs a step
    // ahead, like the next tile's descriptors: a step waited for them in turn
    __shared__ uint8_t s_kind[DGRP + 1];
    for (uint32_t t = threadIdx.x; t <= dgrp && t <= (uint32_t)DGRP; t += DTH) { const uint32_t k = g * dgrp + t; s_kind[t] = (uint8_t)(rt_cnt && t < dgrp && k < n_tiles ? rt_cnt[k] : 0u); }
    __syncthreads();
    auto load = [&](uint32_t t, uint4 &a0, uint4 &a1, uint32_t &rp) {     // (a run tile has no descriptors in memory: its runs are loaded instead)
        const uint32_t k = g * dgrp + t;
        if (t >= dgrp || k >= n_tiles) return;
        const uint32_t kind = s_kind[t];
        if (kind) { if (threadIdx.x < kind) rp = rt_runs[(size_t)k * RT_RUNS + threadIdx.x]; }
        else if (k * DT + x0 < E) { const uint4 *p = reinterpret_cast<const uint4 *>(desc + (size_t)k * DT + x0); a0 = ld16<(RSN_NT_MASK & 64) != 0>(p); a1 = ld16<(RSN_NT_MASK & 64) != 0>(p + 1); }
    };
    uint4 d0 = {0, 0, 0, 0}, d1 = {0, 0, 0, 0}, e0 = d0, e1 = d1;
    uint32_t rp0 = 0, rp1 = 0;
    load(0, d0, d1, rp0);
    __syncthreads();
    for (uint32_t t = 0; t < dgrp; t++) {
        const uint32_t k = g * dgrp + t;
        if (k >= n_tiles) break;
        load(t + 1, e0, e1, rp1);                                          // next tile's descriptors (or runs) in flight during this one
        const uint32_t ts = k * DT, len = min((uint32_t)DT, E - ts);
        const uint32_t nr = s_kind[t];                                    // a run tile: its descriptors are nr runs (block-uniform)
        if (nr) {
            if (threadIdx.x < nr) s_rt[t & 1][threadIdx.x] = rp0;
            __syncthreads();
        }
        if (x0 < len) {
            const uint32_t w[8] = {d0.x, d0.y, d0.z, d0.w, d1.x, d1.y, d1.z, d1.w};
            uint32_t o[4] = {0, 0, 0, 0};
            if (nr) {
                // the run of the lane's first position by bisection, then forward: r follows the position, a new run takes over at its x0
                const uint32_t *runs = s_rt[t & 1];
                uint32_t lo = 0, hi = nr;
                while (hi - lo > 1) { const uint32_t mid = (lo + hi) >> 1; if ((runs[mid] & 0xFFFFu) <= x0) lo = mid; else hi = mid; }
                uint32_t r = (runs[lo] >> 16) + (x0 - (runs[lo] & 0xFFFFu)), nxt_x = lo + 1 < nr ? (runs[lo + 1] & 0xFFFFu) : 0xFFFFFFFFu;
                if (nxt_x >= x0 + 16) {
                    // all 16 positions in one run (runs are thousands of bytes long): 16 consecutive bytes of the tail, five aligned dwords and a byte shift
                    const uint32_t *p32 = reinterpret_cast<const uint32_t *>(prev) + (r >> 2);
                    const uint32_t d0_ = p32[0], d1_ = p32[1], d2_ = p32[2], d3_ = p32[3], d4_ = p32[4];
                    o[0] = __builtin_amdgcn_alignbyte(d1_, d0_, r); o[1] = __builtin_amdgcn_alignbyte(d2_, d1_, r);   // (v_alignbyte uses r[1:0])
                    o[2] = __builtin_amdgcn_alignbyte(d3_, d2_, r); o[3] = __builtin_amdgcn_alignbyte(d4_, d3_, r);
                } else {
#pragma unroll
                    for (int q = 0; q < 16; q++) {
                        if (x0 + (uint32_t)q == nxt_x) { lo++; r = runs[lo] >> 16; nxt_x = lo + 1 < nr ? (runs[lo + 1] & 0xFFFFu) : 0xFFFFFFFFu; }
                        o[q >> 2] |= (uint32_t)prev[r & D_PAY] << (8 * (q & 3));
                        r++;
                    }
                }
            } else {
#pragma unroll
            for (int q = 0; q < 16; q++) {
                const uint32_t v = (w[q >> 1] >> (16 * (q & 1))) & 0xFFFF;
                const uint32_t b = (v & D_EXT) ? prev[v & D_PAY] : (v & 0xFF);
                o[q >> 2] |= b << (8 * (q & 3));
            }
            }
            uint32_t so[4] = {o[0], o[1], o[2], o[3]};                     // what is stored (the tail below keeps the escaped bytes)
            if (map_ff) {
#pragma unroll
                for (int j = 0; j < 4; j++) { const uint32_t m = emit_bytes_equal(so[j], 0xFFu); so[j] = (so[j] & ~m) | (0x3C3C3C3Cu & m); }
            }
            if (x0 + 16 <= len) *reinterpret_cast<uint4 *>(esc + ts + x0) = make_uint4(so[0], so[1], so[2], so[3]);
            else for (uint32_t q = 0; x0 + q < len; q++) esc[ts + x0 + q] = (uint8_t)(so[q >> 2] >> (8 * (q & 3)));
            if (x0 >= DT - TL) *reinterpret_cast<uint4 *>(nxt + (x0 - (DT - TL))) = make_uint4(o[0], o[1], o[2], o[3]);   // the span lies in the tile's tail (TL and x0 are multiples of 16: whole spans)
            if (!map_ff) {                                                // (the unescape pass's block summaries: nobody reads them when the stream holds no 5C)
                const uint32_t nv = min(16u, len - x0);
                const uint32_t non = ~mask_5c(o) & (nv >= 16 ? 0xFFFFu : ((1u << nv) - 1u));
                if (non) atomicMax(&s_last[t & 1][x0 / ZTILE], (x0 % ZTILE) + 32u - (uint32_t)__builtin_clz(non));
            }
        }
        __syncthreads();
        if (!map_ff && threadIdx.x < DT / ZTILE && threadIdx.x * ZTILE < len) {
            const uint32_t blen = min((uint32_t)ZTILE, len - threadIdx.x * ZTILE), last = s_last[t & 1][threadIdx.x];
            summ[ts / ZTILE + threadIdx.x] = (uint8_t)((last == 0 ? 2 : 0) | ((blen - last) & 1));   // bit1: all 5C; bit0: trailing-run parity
            s_last[t & 1][threadIdx.x] = 0;                               // two tiles (and two barriers) later it is used again
        }
        uint8_t *sw = prev; prev = nxt; nxt = sw;
        d0 = e0; d1 = e1; rp0 = rp1;
    }
}

// ------------------------------------------------------------------ L4: unescape
// Every lane takes its 16 bytes with one load; the 5C bytes become a 16-bit mask.
__device__ __forceinline__ uint32_t load16_esc(const uint8_t *__restrict__ esc, size_t E, size_t s, uint32_t w[4]) {
    w[0] = w[1] = w[2] = w[3] = 0;
    if (s + 16 <= E) { const uint4 v = *reinterpret_cast<const uint4 *>(esc + s); w[0] = v.x; w[1] = v.y; w[2] = v.z; w[3] = v.w; return 16; }
    if (s >= E) return 0;
    const uint32_t len = (uint32_t)(E - s);
    for (uint32_t k = 0; k < len; k++) w[k >> 2] |= (uint32_t)esc[s + k] << (8 * (k & 3));
    return len;
}

// (all bytes are 5C) << 1 | parity of the trailing 5C run, for `len` bytes with 5C mask m
__device__ __forceinline__ uint32_t run_summary(uint32_t m, uint32_t len) {
    const uint32_t vm = len >= 16 ? 0xFFFFu : ((1u << len) - 1u);
    const uint32_t non = ~m & vm;                                         // bytes that are not 5C
    const uint32_t last = non ? 32u - (uint32_t)__builtin_clz(non) : 0u;  // index past the last of them
    return (non ? 0u : 2u) | ((len - last) & 1u);
}

// per-block summary: is the whole block 5C, and the parity of its trailing 5C run
__global__ __launch_bounds__(ZB) void k_une_summary(const uint8_t *__restrict__ esc, size_t E, uint8_t *__restrict__ summ) {
    __shared__ uint32_t s_last;          // highest index in the block that is not 5C, +1 (0 = none)
    if (threadIdx.x == 0) s_last = 0;
    __syncthreads();
    const size_t base = (size_t)blockIdx.x * ZTILE;                        // (escaped streams of 4 GiB and more: r04)
    uint32_t w[4];
    const uint32_t len = load16_esc(esc, E, base + threadIdx.x * 16, w);
    const uint32_t non = ~mask_5c(w) & (len >= 16 ? 0xFFFFu : ((1u << len) - 1u));
    if (non) atomicMax(&s_last, threadIdx.x * 16 + 32u - (uint32_t)__builtin_clz(non));
    __syncthreads();
    if (threadIdx.x == 0) {
        const uint32_t blen = (uint32_t)min((size_t)ZTILE, E - base);
        summ[blockIdx.x] = (uint8_t)((s_last == 0 ? 2 : 0) | ((blen - s_last) & 1));   // bit1: all 5C; bit0: trailing-run parity
    }
}

// in_par[b] = parity of the 5C run that ends right before block b.  The (all-5C, parity)
// summaries combine associatively, so one block of 1024 lanes scans them: every lane folds a
// contiguous chunk, the 1024 chunk summaries are scanned by shuffles, every lane replays its chunk.
__global__ __launch_bounds__(1024) void k_une_carry(const uint8_t *__restrict__ summ, uint32_t n_blk, uint8_t *__restrict__ in_par) {
    __shared__ uint8_t s_w[16];
    const uint32_t per = ((n_blk + 1023) / 1024 + 15) & ~15u;           // a multiple of 16: chunks are read and written in 16-byte units
    const uint32_t b0 = threadIdx.x * per, b1 = min(b0 + per, n_blk);
    auto fold = [](uint32_t &all, uint32_t &par, uint32_t v) { if (v & 2) par ^= v & 1; else { all = 0; par = v & 1; } };
    uint32_t all = 2, par = 0;                       // summary of this lane's chunk, same encoding as summ[]
    for (uint32_t b = b0; b < b1; b += 16) {
        if (b + 16 <= n_blk) {
            const uint4 v = *reinterpret_cast<const uint4 *>(summ + b);
            const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
            for (int k = 0; k < 16; k++) fold(all, par, (w[k >> 2] >> (8 * (k & 3))) & 0xFF);
        } else for (uint32_t q = b; q < b1; q++) fold(all, par, summ[q]);
    }
    // the chunk summaries' exclusive scan under the same fold (identity: all-5C, parity 0), by shuffles: one lane chaining 1024 of
    // them was 20 us of every escaped stream's decode, a 4 KiB one included
    auto comb = [](uint32_t a, uint32_t b) { return (b & 2) ? ((a & 2) | ((a ^ b) & 1)) : b; };   // a, then b
    const uint32_t lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    uint32_t inc = all | par;
    for (uint32_t d = 1; d < 64; d <<= 1) { const uint32_t y = __shfl_up(inc, d); if (lane >= d) inc = comb(y, inc); }
    if (lane == 63) s_w[wv] = (uint8_t)inc;
    __syncthreads();
    uint32_t before = 2;
    for (uint32_t k = 0; k < wv; k++) before = comb(before, s_w[k]);
    uint32_t exc = __shfl_up(inc, 1);
    if (lane == 0) exc = 2;
    uint32_t p = comb(before, exc) & 1;
    for (uint32_t b = b0; b < b1; b += 16) {
        if (b + 16 <= n_blk) {
            const uint4 v = *reinterpret_cast<const uint4 *>(summ + b);
            const uint32_t w[4] = {v.x, v.y, v.z, v.w};
            uint32_t o[4] = {0, 0, 0, 0};
#pragma unroll
            for (int k = 0; k < 16; k++) {
                o[k >> 2] |= p << (8 * (k & 3));
                const uint32_t x = (w[k >> 2] >> (8 * (k & 3))) & 0xFF;
                p = (x & 2) ? p ^ (x & 1) : (x & 1);
            }
            *reinterpret_cast<uint4 *>(in_par + b) = make_uint4(o[0], o[1], o[2], o[3]);
        } else for (uint32_t q = b; q < b1; q++) { in_par[q] = (uint8_t)p; const uint32_t x = summ[q]; p = (x & 2) ? p ^ (x & 1) : (x & 1); }
    }
}

// shared front end of count/write: loads the lane's bytes and returns the escape state at their start
__device__ __forceinline__ uint32_t lane_in_parity(const uint8_t *__restrict__ esc, size_t E, size_t base, uint32_t blk_par, uint8_t *s_sum,
                                                   uint32_t w[4], uint32_t *len) {
    const int tid = threadIdx.x;
    *len = load16_esc(esc, E, base + tid * 16, w);
    s_sum[tid] = (uint8_t)run_summary(mask_5c(w), *len);
    __syncthreads();
    uint32_t par = 0;
    int q = tid - 1;
    for (; q >= 0; q--) { const uint32_t v = s_sum[q]; par ^= v & 1; if (!(v & 2)) break; }
    if (q < 0) par ^= blk_par;
    return par;
}

__global__ __launch_bounds__(ZB) void k_une_count(const uint8_t *__restrict__ esc, size_t E, const uint8_t *__restrict__ in_par,
                                                  unsigned long long *__restrict__ blk_len) {
    __shared__ uint8_t s_sum[ZB];
    __shared__ uint32_t part[ZB / 64];
    uint32_t w[4], len;
    uint32_t st = lane_in_parity(esc, E, (size_t)blockIdx.x * ZTILE, in_par[blockIdx.x], s_sum, w, &len);
    uint32_t cnt = 0;
#pragma unroll
    for (uint32_t k = 0; k < 16; k++) {
        const uint32_t v = (w[k >> 2] >> (8 * (k & 3))) & 0xFF;
        if (k < len) { if (v == 0x5C && !st) st = 1; else { st = 0; cnt++; } }     // lzss.go:395-403
    }
    for (int d = 32; d; d >>= 1) cnt += __shfl_down(cnt, d);
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = cnt;
    __syncthreads();
    if (threadIdx.x == 0) blk_len[blockIdx.x] = (unsigned long long)part[0] + part[1] + part[2] + part[3];
}

// 0xFF in every byte of w that equals the byte `c` (exact per byte)
__device__ __forceinline__ uint32_t une_bytes_equal(uint32_t w, uint32_t c) {
    const uint32_t t = w ^ (c * 0x01010101u);
    const uint32_t z = ~(((t & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | t | 0x7F7F7F7Fu);
    return (z >> 7) * 0xFFu;
}

// the block's output is contiguous: bytes go to LDS first, then out in 16-byte units
__global__ __launch_bounds__(ZB) void k_une_write(const uint8_t *__restrict__ esc, size_t E, const uint8_t *__restrict__ in_par,
                                                  const unsigned long long *__restrict__ blk_off, uint8_t *__restrict__ out) {
    __shared__ uint8_t s_sum[ZB];
    __shared__ uint32_t wsum[ZB / 64];
    __shared__ __attribute__((aligned(16))) uint32_t s_out[ZTILE / 4 + 8];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    uint32_t w[4], len;
    const uint32_t st0 = lane_in_parity(esc, E, (size_t)blockIdx.x * ZTILE, in_par[blockIdx.x], s_sum, w, &len);
    uint32_t st = st0, cnt = 0;
#pragma unroll
    for (uint32_t k = 0; k < 16; k++) {
        const uint32_t v = (w[k >> 2] >> (8 * (k & 3))) & 0xFF;
        if (k < len) { if (v == 0x5C && !st) st = 1; else { st = 0; cnt++; } }
    }
    uint32_t incl = cnt;
    for (int d = 1; d < 64; d <<= 1) { const uint32_t y = __shfl_up(incl, d); if (lane >= d) incl += y; }
    if (lane == 63) wsum[wv] = incl;
    __syncthreads();
    uint32_t pre = 0, tot = 0;
    for (int k = 0; k < ZB / 64; k++) { if (k < wv) pre += wsum[k]; tot += wsum[k]; }
    uint8_t *const g0 = out + blk_off[blockIdx.x];
    {   // no escape marker anywhere in the block (most blocks of most files): every byte keeps its place, FF becomes '<'
        const bool plain = st0 == 0 && !(une_bytes_equal(w[0], 0x5Cu) | une_bytes_equal(w[1], 0x5Cu) | une_bytes_equal(w[2], 0x5Cu) | une_bytes_equal(w[3], 0x5Cu));
        if (__syncthreads_and(plain) && ((uintptr_t)g0 & 3) == 0) {
#pragma unroll
            for (int j = 0; j < 4; j++) { const uint32_t m = une_bytes_equal(w[j], 0xFFu); w[j] = (w[j] & ~m) | (0x3C3C3C3Cu & m); }
            uint8_t *d = g0 + tid * 16;
            if (len == 16) *reinterpret_cast<uint4 *>(d) = make_uint4(w[0], w[1], w[2], w[3]);   // (a 16-byte store only needs dword alignment)
            else for (uint32_t k = 0; k < len; k++) d[k] = (uint8_t)(w[k >> 2] >> (8 * (k & 3)));
            return;
        }
    }
    uint8_t *so = reinterpret_cast<uint8_t *>(s_out);
    uint32_t o = pre + incl - cnt;
    st = st0;
#pragma unroll
    for (uint32_t k = 0; k < 16; k++) {
        const uint32_t v = (w[k >> 2] >> (8 * (k & 3))) & 0xFF;
        if (k < len) {
            if (v == 0xFF && !st) so[o++] = 0x3C;                         // EncodedOpening -> '<'
            else if (v == 0x5C && !st) st = 1;                            // escape marker: emits nothing
            else { st = 0; so[o++] = (uint8_t)v; }
        }
    }
    __syncthreads();
    uint8_t *g = out + blk_off[blockIdx.x];
    const uint32_t head = min(tot, (uint32_t)((16 - ((uintptr_t)g & 15)) & 15));   // bytes before the first 16-byte boundary
    if ((uint32_t)tid < head) g[tid] = so[tid];
    const uint32_t units = (tot - head) / 16;
    for (uint32_t u = tid; u < units; u += ZB) {
        uint32_t x[4];
        load_span(s_out, (int)(head + 16 * u), x);
        *reinterpret_cast<uint4 *>(g + head + 16 * u) = make_uint4(x[0], x[1], x[2], x[3]);
    }
    const uint32_t tail0 = head + 16 * units;
    if (tail0 + tid < tot) g[tail0 + tid] = so[tail0 + tid];
}

// ======================================================================= host side
namespace {

// The decoder's large scratch -- 2 bytes of descriptors per escaped byte, the escaped stream itself, the tail maps -- goes through the
// same admission gate as the encoder's (rsn_api.hip: a goroutine storm of gigabyte calls queues instead of running the device out of
// memory); released, with the buffers when others wait, on every way out.  Slots 13 .. 16, 19, 22, 23, 25, 27, 36.
struct DecGate {
    Ctx &c; size_t held = 0; bool asked = false;
    explicit DecGate(Ctx &cc) : c(cc) {}
    void admit(size_t escaped) { if (!asked && escaped >= ((size_t)64 << 20)) { asked = true; held = scratch_admit(c, 4 * escaped); } }   // (0 inside a host-buffer call: covered there)
    ~DecGate() { scratch_release(c, held, (0xFull << 13) | (1ull << 19) | (3ull << 22) | (1ull << 25) | (1ull << 27) | (1ull << 36)); }
};

// L4 over an escaped stream of any length: d_esc[0, E) -> d_out.  have_summ: the per-block summaries are in place already (the tile
// path's emit kernel writes them on its way out).  Scratch slot 16.
int lzss_unescape(Ctx &c, hipStream_t s, const uint8_t *d_esc, size_t E, bool have_summ, uint8_t *d_out, size_t out_cap, size_t *out_n) {
    void *p; int rc;
    if (ceil_div(E, (size_t)ZTILE) > 0xFFFFFF00ull) return c.fail(RSN_ERR_LIMIT, "lzss: decoded stream too large for one call");
    const uint32_t n_ub = (uint32_t)ceil_div(E, (size_t)ZTILE);          // unescape blocks
    rc = dev_buf(c, 16, ((size_t)n_ub * 2 + 2) * 8 + (size_t)round_up(n_ub, 16) * 2 + 64, &p); if (rc) return rc;
    unsigned long long *d_ulen = (unsigned long long *)p, *d_uoff = d_ulen + n_ub, *d_utot = d_uoff + n_ub;
    uint8_t *d_summ = (uint8_t *)(d_utot + 2), *d_inpar = d_summ + round_up(n_ub, 16);   // both 16-byte aligned
    void *hp; rc = pinned_buf(c, 64, &hp); if (rc) return rc;
    unsigned long long *h64 = (unsigned long long *)hp;
    if (!have_summ) RSN_LAUNCH("lzss_une_summary", k_une_summary, dim3(n_ub), dim3(ZB), 0, s, d_esc, E, d_summ);
    RSN_LAUNCH("lzss_une_carry", k_une_carry, dim3(1), dim3(1024), 0, s, (const uint8_t *)d_summ, n_ub, d_inpar);
    RSN_LAUNCH("lzss_une_count", k_une_count, dim3(n_ub), dim3(ZB), 0, s, d_esc, E, (const uint8_t *)d_inpar, d_ulen);
    rc = scan_u64(c, s, "lzss_dec_scan", d_ulen, d_uoff, n_ub, d_utot); if (rc) return rc;
    RSN_HIP(hipMemcpyAsync(h64, d_utot, 8, hipMemcpyDeviceToHost, s));
    RSN_HIP(hipStreamSynchronize(s));
    const size_t total = (size_t)h64[0];
    *out_n = total;
    if (!d_out || total > out_cap) { *out_n = round_up(total, 16) + 16; return c.fail(RSN_ERR_CAPACITY, "lzss: output needs %zu bytes, buffer holds %zu", total, out_cap); }
    RSN_LAUNCH("lzss_une_write", k_une_write, dim3(n_ub), dim3(ZB), 0, s, d_esc, E, (const uint8_t *)d_inpar, (const unsigned long long *)d_uoff, d_out);
    RSN_HIP(hipStreamSynchronize(s));
    return RSN_OK;
}

int lzss_decode_sections(Ctx &c, hipStream_t s, const uint8_t *d_in, size_t n, uint8_t *d_out, size_t out_cap, size_t *out_n);
int lzss_decode_impl(Ctx &c, hipStream_t s, const uint8_t *d_in, size_t n, uint8_t *d_out, size_t out_cap, size_t *out_n, uint8_t *esc_dst);

// ---------------------------------------------------------------- a stream that ends in ONE token repeated (r06)
// What CompressAsync writes for data that repeats with the window's length (BASELINE configs[2]: one 4096-byte block over and over) is a
// short head, then "<4096,4096>" a quarter of a million times, then one last item (lzss.go:134-151).  A token <P,P> copies the P bytes
// before it: behind the head the output repeats with period P, whatever the bytes are (lzss.go:349-353).  r05 decoded such a stream like
// any other -- run tiles, 0.95 ms per GiB of output, of which writing the gigabyte is 0.3.  Now the host reads the stream's last 64 bytes
// (with the counting pass's results: no extra round trip), and when they end in "<P,P><P,P>" + at most one more item, a kernel finds where
// that token's run begins in the COMPRESSED stream (the last byte that differs from the byte |token| before it: 3 MB to look at, not a
// gigabyte); the counting pass has accepted every token as well-formed, so a '<' is never inside one and the run's first '<' is an item
// boundary of the true parse.  The head in front of it is decoded by the ordinary decoder, the run is written by arithmetic --
// out[q] = out[q - P] -- and the last item by a block of its own.  Only for streams without a 5C (nothing to unescape but FF -> '<', which
// the head's emit pass has applied to the bytes the run copies); everything else about the stream is checked as always.
__global__ __launch_bounds__(256) void k_lzd_run_start(const uint8_t *__restrict__ in, unsigned long long t_end, uint32_t tl, unsigned long long *__restrict__ last_break) {
    // *last_break = 1 + the largest i in [tl, t_end) with in[i] != in[i - tl]   (0: none)
    const unsigned long long i0 = tl + ((unsigned long long)blockIdx.x * 256 + threadIdx.x) * 16;
    unsigned long long m = 0;
    for (int k = 0; k < 16; k++) { const unsigned long long i = i0 + k; if (i < t_end && in[i] != in[i - tl]) m = i + 1; }
    for (int d = 32; d; d >>= 1) m = max(m, (unsigned long long)__shfl_down(m, d));
    if ((threadIdx.x & 63) == 0 && m) atomicMax(last_break, m);
}
constexpr uint32_t RUN_P_MAX = 8192;                                     // periods the fill kernel keeps in LDS
struct RunFill { uint8_t *out; uint32_t lo, hi, P; };                    // out[q] = out[q - P] for q in [lo, hi); out[lo - P, lo) is in place
__global__ __launch_bounds__(256) void k_lzd_run_fill(RunFill a) {
    __shared__ __attribute__((aligned(16))) uint8_t s_pat[RUN_P_MAX + 32];   // the period, and its first 16 bytes once more behind it
    const uint8_t *src = a.out + (a.lo - a.P);
    for (uint32_t i = threadIdx.x; i < a.P + 16; i += 256) s_pat[i] = src[i % a.P];
    __syncthreads();
    constexpr uint32_t CHUNK = 256 * 16 * 16;                             // output bytes per block
    const unsigned long long u_first = a.lo >> 4;
    for (int it = 0; it < 16; it++) {
        const unsigned long long u = u_first + (unsigned long long)blockIdx.x * (CHUNK / 16) + (unsigned)it * 256 + threadIdx.x;
        const unsigned long long b0 = u << 4;
        if (b0 >= a.hi) break;
        if (b0 >= a.lo && b0 + 16 <= a.hi) {
            const uint32_t ph = (uint32_t)(b0 - a.lo) % a.P;
            uint4 v;
            __builtin_memcpy(&v, s_pat + ph, 16);
            *reinterpret_cast<uint4 *>(a.out + b0) = v;
        } else for (int k = 0; k < 16; k++) { const unsigned long long q = b0 + k; if (q >= a.lo && q < a.hi) a.out[q] = s_pat[(uint32_t)(q - a.lo) % a.P]; }
    }
}
// the stream's last item behind the run: a token (out[x + k] = out[x - d + k], k < r <= d) or r literal bytes (FF -> '<': DecodeOpeningSymbols, lzss.go:391-406)
__global__ __launch_bounds__(256) void k_lzd_run_last(uint8_t *__restrict__ out, uint32_t x, uint32_t d, uint32_t r, const uint8_t *__restrict__ lit) {
    for (uint32_t k = threadIdx.x; k < r; k += 256) { const uint8_t b = lit ? lit[k] : out[x - d + k]; out[x + k] = lit && b == 0xFF ? (uint8_t)0x3C : b; }
}

// 1 = not a stream for this path (the caller decodes it as always); otherwise the call's result.
// tail: the stream's last min(n, 64) bytes (host copy); E: its decoded length (the counting pass's); d_brk: a device word for k_lzd_run_start
static int lzss_decode_run_tail(Ctx &c, hipStream_t s, const uint8_t *d_in, size_t n, const uint8_t *tail, uint32_t E, unsigned long long *d_brk,
                                unsigned long long *h_brk, uint8_t *d_out, size_t out_cap, size_t *out_n) {
    const size_t k = std::min<size_t>(n, 64), base = n - k;
    auto token_at = [&](size_t i, uint32_t *d, uint32_t *l) -> size_t {   // the length of the well-formed token at tail[i], 0 if there is none
        if (i >= k || tail[i] != '<') return 0;
        size_t j = i + 1; unsigned long long v = 0; int nd = 0;
        while (j < k && nd < 10 && tail[j] >= '0' && tail[j] <= '9') { v = v * 10 + (tail[j] - '0'); j++; nd++; }
        if (!nd || j >= k || tail[j] != ',' || v > 0xFFFFFFFFull) return 0;
        *d = (uint32_t)v; j++; v = 0; nd = 0;
        while (j < k && nd < 10 && tail[j] >= '0' && tail[j] <= '9') { v = v * 10 + (tail[j] - '0'); j++; nd++; }
        if (!nd || j >= k || tail[j] != '>' || v > 0xFFFFFFFFull) return 0;
        *l = (uint32_t)v;
        return j + 1 - i;
    };
    // the last token <P,P> of the tail, one more of it right in front, and behind it nothing, one token, or literals
    size_t ti = k; uint32_t P = 0, tl = 0;
    for (size_t i = k; i-- > 0;) {
        uint32_t d = 0, l = 0;
        const size_t len = token_at(i, &d, &l);
        if (len && d == l && d >= 1) { ti = i; P = d; tl = (uint32_t)len; break; }
    }
    if (ti == k || P > RUN_P_MAX || ti < tl || memcmp(tail + ti - tl, tail + ti, tl) != 0) return 1;
    const size_t ri = ti + tl;                                            // the rest: [ri, k)
    uint32_t rd = 0, rr = 0; bool rest_lit = false;
    if (ri < k) {
        if (tail[ri] == '<') { if (token_at(ri, &rd, &rr) != k - ri || rr > rd) return 1; }
        else { for (size_t i = ri; i < k; i++) if (tail[i] == '<') return 1; rest_lit = true; rr = (uint32_t)(k - ri); }
    }
    const unsigned long long t_end = base + ri;
    // ---- where the run begins
    RSN_HIP(hipMemsetAsync(d_brk, 0, 8, s));
    if (t_end > tl) RSN_LAUNCH("lzss_dec_run_start", k_lzd_run_start, dim3((uint32_t)ceil_div((size_t)(t_end - tl), (size_t)4096)), dim3(256), 0, s, d_in, t_end, tl, d_brk);
    RSN_HIP(hipMemcpyAsync(h_brk, d_brk, 8, hipMemcpyDeviceToHost, s));
    RSN_HIP(hipStreamSynchronize(s));
    const unsigned long long brk = *h_brk;
    unsigned long long a0 = std::max<unsigned long long>(brk, tl) - tl;   // [a0, t_end) repeats with the token's length and ends in the token
    a0 += (t_end - a0) % tl;                                              // ... and from here it is whole tokens
    const unsigned long long m = (t_end - a0) / tl;
    if (a0 == 0 || m * P < E / 2 || m * P + rr > E) return 1;            // (nothing in front of the run: its first token points before the data -- the ordinary path words that)
    // ---- the head
    static thread_local int depth = 0;
    if (depth) return 1;
    size_t e_head = 0;
    depth++;
    int rc = lzss_decode_impl(c, s, d_in, (size_t)a0, d_out, out_cap, &e_head, nullptr);
    depth--;
    if (rc == RSN_ERR_CAPACITY) { *out_n = round_up((size_t)E, 16) + 16; return c.fail(RSN_ERR_CAPACITY, "lzss: output needs %u bytes, buffer holds %zu", E, out_cap); }
    if (rc) return rc;
    if (e_head + m * P + rr != E) return c.fail(RSN_ERR_DEVICE, "lzss: internal error: a token run of %llu x %u behind %zu bytes in a stream of %u", m, P, e_head, E);
    if (P > e_head || (!rest_lit && rr && rd > e_head + m * P))
        return c.fail(RSN_ERR_FORMAT, "lzss: back-reference outside the decoded data (reference: slice bounds out of range, lzss.go:350)");
    *out_n = E;
    if (E > out_cap) { *out_n = round_up((size_t)E, 16) + 16; return c.fail(RSN_ERR_CAPACITY, "lzss: output needs %u bytes, buffer holds %zu", E, out_cap); }
    RunFill rf{d_out, (uint32_t)e_head, (uint32_t)(e_head + m * P), P};
    const unsigned long long units = (((unsigned long long)rf.hi + 15) >> 4) - (rf.lo >> 4);
    RSN_LAUNCH("lzss_dec_run_fill", k_lzd_run_fill, dim3((uint32_t)ceil_div((size_t)units, (size_t)4096)), dim3(256), 0, s, rf);
    if (rr) RSN_LAUNCH("lzss_dec_run_fill", k_lzd_run_last, dim3(1), dim3(256), 0, s, d_out, rf.hi, rd, rr, rest_lit ? d_in + (size_t)t_end : (const uint8_t *)nullptr);
    RSN_HIP(hipStreamSynchronize(s));
    return RSN_OK;
}

// One stream of less than 4 GiB, compressed and decoded.  esc_dst == nullptr: the decoder proper (-> d_out).  Otherwise only the token
// expansion (L1 .. L3): the ESCAPED stream goes to esc_dst (16-byte aligned, room for the whole of it plus 64 bytes: the caller knows
// its length from its own counting pass) and *out_n is that length -- what lzss_decode_sections runs per section.
int lzss_decode_impl(Ctx &c, hipStream_t s, const uint8_t *d_in, size_t n, uint8_t *d_out, size_t out_cap, size_t *out_n, uint8_t *esc_dst) {
    *out_n = 0;
    if (n == 0) return RSN_OK;
    if (((uintptr_t)d_in & 15) || ((uintptr_t)d_out & 15)) return c.fail(RSN_ERR_ARG, "lzss: device buffers must be 16-byte aligned");
    static const size_t sec_env = [] { const char *e = getenv("RSN_LZSS_DEC_SECTION_MIB"); return e ? (size_t)std::max(atoi(e), 1) << 20 : (size_t)0; }();   // testing switch: sections on small streams
    if (!esc_dst && (n >= (1ull << 32) - 65536 || (sec_env && n > sec_env / 2))) return lzss_decode_sections(c, s, d_in, n, d_out, out_cap, out_n);
    if (n >= (1ull << 32) - 65536) return c.fail(RSN_ERR_LIMIT, "lzss: compressed input too large for one call");
    void *p; int rc;
    const uint32_t n_cb = (uint32_t)ceil_div(n, ZTILE);
    rc = dev_buf(c, 13, ((size_t)n_cb * 2 + 5) * 8, &p); if (rc) return rc;   // ... + flags: [0] error, [1] changed, [2] largest back-pointer, [3] tile path gave up
    unsigned long long *d_blen = (unsigned long long *)p, *d_boff = d_blen + n_cb, *d_btot = d_boff + n_cb;
    int *d_flag = (int *)(d_btot + 1);
    unsigned long long *d_brk = d_btot + 4;                             // (k_lzd_run_start's word)
    void *hp; rc = pinned_buf(c, 192, &hp); if (rc) return rc;
    unsigned long long *h64 = (unsigned long long *)hp;
    uint8_t *h_tail = (uint8_t *)(h64 + 8);                             // the stream's last 64 bytes, read with the counting pass's results; h64[16]: k_lzd_run_start's answer
    volatile int *hflag = (volatile int *)(h64 + 1);
    RSN_HIP(hipMemsetAsync(d_flag, 0, 24, s));                          // ... [4] the stream holds a 5C byte, [5] a span produces 65535 bytes or more
    uint16_t *d_span = nullptr; unsigned long long *d_need = nullptr;
    rc = dev_buf(c, 27, (size_t)n_cb * ZB * 2 + (size_t)n_cb * 8 + 64, &p); if (rc) return rc;
    d_need = (unsigned long long *)p; d_span = (uint16_t *)(d_need + n_cb);
    RSN_LAUNCH("lzss_dec_count", k_lzd_count2, dim3(n_cb), dim3(ZB), 0, s, d_in, n, d_blen, d_span, d_need, (uint32_t *)(d_flag + 2), d_flag);
    rc = scan_u64(c, s, "lzss_dec_scan", d_blen, d_boff, n_cb, d_btot); if (rc) return rc;
    RSN_HIP(hipMemcpyAsync(h64, d_btot, 8, hipMemcpyDeviceToHost, s));
    RSN_HIP(hipMemcpyAsync((void *)hflag, d_flag, 24, hipMemcpyDeviceToHost, s));
    static const bool no_run_tail = getenv("RSN_LZSS_DEC_NO_RUN_TAIL") != nullptr;   // A/B switch (tests): a stream that ends in one token repeated through the ordinary decoder
    const bool run_cand = !esc_dst && d_out && !no_run_tail && n >= ((size_t)1 << 16);
    if (run_cand) RSN_HIP(hipMemcpyAsync(h_tail, d_in + (n - 64), 64, hipMemcpyDeviceToHost, s));
    RSN_HIP(hipStreamSynchronize(s));
    if (hflag[0] & 1) return c.fail(RSN_ERR_FORMAT, "lzss: malformed \"<ptr,len>\" token");
    if (hflag[0] & 2) return c.fail(RSN_ERR_FORMAT, "lzss: back-reference outside the decoded data (reference: slice bounds out of range, lzss.go:350)");
    if (h64[0] >= (1ull << 32) - 65536 || (sec_env && !esc_dst && h64[0] > sec_env)) {
        if (esc_dst) return c.fail(RSN_ERR_LIMIT, "lzss: decoded stream too large for one call");
        return lzss_decode_sections(c, s, d_in, n, d_out, out_cap, out_n);   // (counts again: a stream of 4 GiB does not notice)
    }
    const bool one_pass = hflag[5] == 0;                 // the counting pass left everything k_lzd_tiles would work out again
    const uint32_t E = (uint32_t)h64[0];
    if (E == 0) return RSN_OK;
    if (run_cand && one_pass && hflag[4] == 0 && E >= (1u << 20)) {      // (no 5C: what is left of DecodeOpeningSymbols is bytewise)
        rc = lzss_decode_run_tail(c, s, d_in, n, h_tail, E, d_brk, h64 + 16, d_out, out_cap, out_n);
        if (rc != 1) return rc;
        *out_n = 0;
    }
    DecGate gate(c);
    if (!esc_dst && d_out) gate.admit(E);                               // (a section is admitted by lzss_decode_sections; the size query allocates nothing)
    if (!d_out && !esc_dst) {   // the size query: the escaped length is known here, and unescaping only ever shortens it -- a capacity that suffices, one pass over the tokens
        *out_n = round_up((size_t)E, 16) + 16;
        return c.fail(RSN_ERR_CAPACITY, "lzss: output needs at most %u bytes", E);
    }
    const bool plain = hflag[4] == 0 && !esc_dst;          // no 5C anywhere: unescaping is FF -> '<', done by k_lzd_emit on its way out
    uint8_t *d_esc = esc_dst;
    if (!d_esc) { rc = dev_buf(c, 15, (size_t)E + 64, &p); if (rc) return rc; d_esc = (uint8_t *)p; }
    const uint32_t n_ub = (uint32_t)ceil_div(E, ZTILE);                 // unescape blocks
    rc = dev_buf(c, 16, ((size_t)n_ub * 2 + 2) * 8 + (size_t)round_up(n_ub, 16) * 2 + 64, &p); if (rc) return rc;
    uint8_t *d_summ = (uint8_t *)((unsigned long long *)p + (size_t)n_ub * 2 + 2);   // (lzss_unescape's layout of slot 16: k_lzd_emit leaves the block summaries where it looks for them)
    // ---- L2
    const uint32_t n_tiles = (uint32_t)ceil_div(E, DT), dgrp = group_tiles(n_tiles), n_groups = (uint32_t)ceil_div(n_tiles, dgrp);
    rc = dev_buf(c, 19, (size_t)n_tiles * 8 + 64, &p); if (rc) return rc;
    uint2 *d_tinfo = (uint2 *)p;
    uint32_t *d_maxptr = (uint32_t *)(d_flag + 2);
    int *d_fallback = d_flag + 3;
    volatile uint32_t *hmax = (volatile uint32_t *)(hflag + 2);
    if (one_pass) {
        // (no round trip here: the tile kernels below cannot leave their LDS arrays whatever the pointers are -- len <= ptr <= DT is known --
        //  and the verdict of k_lzd_check is read with the flags after them)
        RSN_LAUNCH("lzss_dec_tiles", k_lzd_check, dim3((uint32_t)ceil_div(n_cb, 256)), dim3(256), 0, s, (const unsigned long long *)d_need, (const unsigned long long *)d_boff, n_cb, d_flag);
        RSN_LAUNCH("lzss_dec_tiles", k_lzd_tilemap, dim3((uint32_t)ceil_div(n_tiles, 4)), dim3(256), 0, s, d_in, n, (const unsigned long long *)d_boff, n_cb, (const uint16_t *)d_span, n_tiles, d_tinfo);
    } else {
        RSN_HIP(hipMemsetAsync(d_maxptr, 0, 4, s));      // (k_lzd_tiles raises it again)
        RSN_LAUNCH("lzss_dec_tiles", k_lzd_tiles, dim3(n_cb), dim3(ZB), 0, s, d_in, n, d_boff, d_tinfo, d_maxptr, d_flag);
        RSN_HIP(hipMemcpyAsync((void *)hflag, d_flag, 16, hipMemcpyDeviceToHost, s));
        RSN_HIP(hipStreamSynchronize(s));   // nothing may chase a pointer before every token has been validated
        if (hflag[0]) return c.fail(RSN_ERR_FORMAT, "lzss: back-reference outside the decoded data (reference: slice bounds out of range, lzss.go:350)");
    }
    static const bool force_jump = getenv("RSN_LZSS_DEC_JUMP") != nullptr;   // A/B switch: whole-stream pointer jumping
    bool tile_path = !force_jump && hmax[0] <= (uint32_t)DT;
    if (tile_path) {
        uint32_t TL = 256;
        while (TL < hmax[0]) TL <<= 1;
        rc = dev_buf(c, 14, ((size_t)n_tiles * DT + 64) * 2, &p); if (rc) return rc;
        uint16_t *d_desc = (uint16_t *)p;
        rc = dev_buf(c, 22, (size_t)n_groups * TL * 5 + 64, &p); if (rc) return rc;
        uint16_t *d_comp = (uint16_t *)p, *d_comp2 = d_comp + (size_t)n_groups * TL;   // (two map arrays: the scan below ping-pongs)
        uint8_t *d_gtail = (uint8_t *)(d_comp2 + (size_t)n_groups * TL);
        static const bool no_runs = getenv("RSN_LZSS_DEC_NO_RUNS") != nullptr;   // A/B switch: descriptors for every tile (no run tiles)
        uint32_t *d_rt_cnt = nullptr, *d_rt_runs = nullptr;
        if (!no_runs) {
            rc = dev_buf(c, 23, (size_t)n_tiles * (RT_RUNS + 1) * 4 + 128, &p); if (rc) return rc;
            d_rt_runs = (uint32_t *)p; d_rt_cnt = d_rt_runs + (size_t)n_tiles * RT_RUNS;
        }
        ResolveArgs ra{d_in, n, d_tinfo, n_tiles, E, TL, d_desc, d_fallback, d_rt_cnt, d_rt_runs};
        if (d_rt_cnt) RSN_LAUNCH("lzss_dec_runs", k_lzd_runs, dim3(n_tiles), dim3(64), 0, s, d_in, n, (const uint2 *)d_tinfo, n_tiles, E, TL, d_rt_cnt, d_rt_runs);
        RSN_LAUNCH("lzss_dec_resolve", k_lzd_resolve, dim3(n_tiles), dim3(DTH), 0, s, ra);
        rc = func_dyn_lds(c, reinterpret_cast<const void *>(k_lzd_compose), (size_t)TL * 4); if (rc) return rc;
        if (n_groups > 1) {
            RSN_LAUNCH("lzss_dec_compose", k_lzd_compose, dim3(n_groups - 1), dim3(DTH), (size_t)TL * 4, s, d_desc, TL, dgrp, d_comp, (const uint32_t *)d_rt_cnt, (const uint32_t *)d_rt_runs);
            const uint32_t n_links = n_groups - 1;
            {
                const uint16_t *cur = d_comp; uint16_t *oth = d_comp2;
                for (uint32_t d = 1; d < n_links; d <<= 1) {
                    RSN_LAUNCH("lzss_dec_chain", k_lzd_mapscan, dim3(n_links), dim3(256), 0, s, cur, oth, TL, d);
                    const uint16_t *t = cur; cur = oth; oth = const_cast<uint16_t *>(t);
                }
                RSN_LAUNCH("lzss_dec_chain", k_lzd_map_bytes, dim3(n_links), dim3(256), 0, s, cur, TL, d_gtail);
            }
        }
        if (plain) {
            *out_n = E;
            if (!d_out || E > out_cap) {
                // (k_lzd_check's verdict before the capacity answer -- a malformed stream is RSN_ERR_FORMAT on the first call,
                //  whatever the buffer; a round trip on this path only)
                if (one_pass) {
                    RSN_HIP(hipMemcpyAsync((void *)hflag, d_flag, 16, hipMemcpyDeviceToHost, s));
                    RSN_HIP(hipStreamSynchronize(s));
                    if (hflag[0] & 2) return c.fail(RSN_ERR_FORMAT, "lzss: back-reference outside the decoded data (reference: slice bounds out of range, lzss.go:350)");
                }
                *out_n = round_up((size_t)E, 16) + 16;
                return c.fail(RSN_ERR_CAPACITY, "lzss: output needs %u bytes, buffer holds %zu", E, out_cap);
            }
        }
        RSN_LAUNCH("lzss_dec_emit", k_lzd_emit, dim3(n_groups), dim3(DTH), (size_t)TL * 2 + 16, s, d_desc, TL, n_tiles, dgrp, E, d_gtail, plain ? d_out : d_esc, d_summ, plain ? 1 : 0,
                   (const uint32_t *)d_rt_cnt, (const uint32_t *)d_rt_runs);   // (+16: a run tile reads whole dwords of the tail, up to three bytes past it)
        RSN_HIP(hipMemcpyAsync((void *)hflag, d_flag, 16, hipMemcpyDeviceToHost, s));
        RSN_HIP(hipStreamSynchronize(s));
        if (hflag[0] & 2) return c.fail(RSN_ERR_FORMAT, "lzss: back-reference outside the decoded data (reference: slice bounds out of range, lzss.go:350)");   // k_lzd_check's verdict
        if (hflag[3]) tile_path = false;                                  // a tile's input did not fit (zero-length tokens): redo with the general path
        else if (plain) return RSN_OK;                                    // bytes are in place
    } else if (one_pass) {                                                // the general path chases pointers through memory: k_lzd_check's verdict first
        RSN_HIP(hipMemcpyAsync((void *)hflag, d_flag, 16, hipMemcpyDeviceToHost, s));
        RSN_HIP(hipStreamSynchronize(s));
        if (hflag[0] & 2) return c.fail(RSN_ERR_FORMAT, "lzss: back-reference outside the decoded data (reference: slice bounds out of range, lzss.go:350)");
    }
    if (!tile_path) {
        rc = dev_buf(c, 14, (size_t)E * 4 + 64, &p); if (rc) return rc;
        uint32_t *d_src = (uint32_t *)p;
        RSN_LAUNCH("lzss_dec_expand", k_lzd_expand, dim3(n_cb), dim3(ZB), 0, s, d_in, n, d_boff, d_src, d_esc, d_flag);
        const uint32_t grid = (uint32_t)std::min<size_t>(ceil_div(E, ZB), 8192);
        for (int round = 0;; round++) {
            if (round > 40) return c.fail(RSN_ERR_DEVICE, "lzss: pointer jumping did not converge");
            RSN_HIP(hipMemsetAsync(d_flag + 1, 0, 4, s));
            RSN_LAUNCH("lzss_dec_jump", k_lzd_jump, dim3(grid), dim3(ZB), 0, s, d_src, E, d_flag + 1);
            RSN_HIP(hipMemcpyAsync((void *)hflag, d_flag, 8, hipMemcpyDeviceToHost, s));
            RSN_HIP(hipStreamSynchronize(s));
            if (!hflag[1]) break;
        }
        RSN_LAUNCH("lzss_dec_gather", k_lzd_gather, dim3(grid), dim3(ZB), 0, s, d_src, d_esc, E);
    }
    if (esc_dst) { RSN_HIP(hipStreamSynchronize(s)); *out_n = E; return RSN_OK; }
    // ---- unescape (the tile path's emit kernel has left the block summaries in slot 16, where lzss_unescape expects them)
    return lzss_unescape(c, s, d_esc, (size_t)E, tile_path, d_out, out_cap, out_n);
}

// ---------------------------------------------------------------- streams of 4 GiB and more (r04), compressed or decoded
// The tile kernels count positions in 32 bits, so a long stream is expanded SECTION BY SECTION: the counting pass over the whole
// stream gives every 4 KiB block of the compressed stream its offset in the escaped one; cuts are taken at block starts (moved past
// a token that straddles one) so that a section holds at most 1 GiB on either side; and a section is decoded as a stream of its own
// that BEGINS WITH THE ESCAPED BYTES BEFORE IT, as many as the largest back-pointer of the stream reaches (the escaped stream holds
// no '<' -- lzss.go:373-377 -- so those bytes parse as literals and every pointer of the section finds what it points at).  The
// expansion of that virtual stream is written over the place it came from: the prefix lands on itself, the section behind it.
// Unescaping then runs once over the whole escaped stream (its kernels count blocks, not bytes).
int lzss_decode_sections(Ctx &c, hipStream_t s, const uint8_t *d_in, size_t n, uint8_t *d_out, size_t out_cap, size_t *out_n) {
    static const size_t sec_env = [] { const char *e = getenv("RSN_LZSS_DEC_SECTION_MIB"); return e ? (size_t)std::max(atoi(e), 1) << 20 : (size_t)0; }();
    const size_t SEC = sec_env ? sec_env : (size_t)1 << 30;
    static const bool dbg = getenv("RSN_DEBUG") != nullptr;
    void *p; int rc;
    if (ceil_div(n, (size_t)ZTILE) > 0xFFFFFF00ull) return c.fail(RSN_ERR_LIMIT, "lzss: compressed input too large for one call");
    const uint32_t n_cb = (uint32_t)ceil_div(n, (size_t)ZTILE);
    std::vector<unsigned long long> boff((size_t)n_cb + 1);
    int hflag[6];
    {   // the counting pass over the whole stream (its per-span outputs are not kept: every section counts its own)
        rc = dev_buf(c, 13, ((size_t)n_cb * 2 + 4) * 8, &p); if (rc) return rc;
        unsigned long long *d_blen = (unsigned long long *)p, *d_boff = d_blen + n_cb, *d_btot = d_boff + n_cb;
        int *d_flag = (int *)(d_btot + 1);
        RSN_HIP(hipMemsetAsync(d_flag, 0, 24, s));
        rc = dev_buf(c, 27, (size_t)n_cb * ZB * 2 + (size_t)n_cb * 8 + 64, &p); if (rc) return rc;
        unsigned long long *d_need = (unsigned long long *)p; uint16_t *d_span = (uint16_t *)(d_need + n_cb);
        RSN_LAUNCH("lzss_dec_count", k_lzd_count2, dim3(n_cb), dim3(ZB), 0, s, d_in, n, d_blen, d_span, d_need, (uint32_t *)(d_flag + 2), d_flag);
        rc = scan_u64(c, s, "lzss_dec_scan", d_blen, d_boff, n_cb, d_btot); if (rc) return rc;
        RSN_LAUNCH("lzss_dec_tiles", k_lzd_check, dim3((uint32_t)ceil_div(n_cb, 256)), dim3(256), 0, s, (const unsigned long long *)d_need, (const unsigned long long *)d_boff, n_cb, d_flag);
        RSN_HIP(hipMemcpyAsync(boff.data(), d_boff, (size_t)n_cb * 8, hipMemcpyDeviceToHost, s));
        RSN_HIP(hipMemcpyAsync(&boff[n_cb], d_btot, 8, hipMemcpyDeviceToHost, s));
        RSN_HIP(hipMemcpyAsync(hflag, d_flag, 24, hipMemcpyDeviceToHost, s));
        RSN_HIP(hipStreamSynchronize(s));
    }
    if (hflag[0] & 1) return c.fail(RSN_ERR_FORMAT, "lzss: malformed \"<ptr,len>\" token");
    if (hflag[0] & 2) return c.fail(RSN_ERR_FORMAT, "lzss: back-reference outside the decoded data (reference: slice bounds out of range, lzss.go:350)");
    const size_t E = (size_t)boff[n_cb];
    const size_t hmax = (size_t)(uint32_t)hflag[2];                       // the largest back-pointer of a token that produces anything
    if (E == 0) return RSN_OK;
    if (!d_out) { *out_n = round_up(E, 16) + 16; return c.fail(RSN_ERR_CAPACITY, "lzss: output needs at most %zu bytes", E); }
    if (hmax + 16 > SEC && hmax + 16 > ((size_t)1 << 30)) return c.fail(RSN_ERR_LIMIT, "lzss: a stream of 4 GiB and more with back-pointers beyond 1 GiB");
    DecGate gate(c);
    gate.admit(E / 4 + std::min(E, SEC + ((size_t)1 << 28)));          // (admit() charges four times its argument: the whole escaped stream + four bytes per escaped byte of a section)
    rc = dev_buf(c, 15, E + 64, &p); if (rc) return rc;
    uint8_t *d_esc = (uint8_t *)p;
    // the first item that STARTS at or after compressed position q (a block start): the tokens are known to be well-formed
    auto item_start = [&](size_t q, size_t *out) -> int {
        if (q == 0 || q >= n) { *out = std::min(q, n); return RSN_OK; }
        uint8_t buf[2 * MAXTOK];
        const size_t lo = q >= (size_t)MAXTOK ? q - MAXTOK : 0, hi = std::min(n, q + MAXTOK);
        RSN_HIP(hipMemcpyAsync(buf, d_in + lo, hi - lo, hipMemcpyDeviceToHost, s));
        RSN_HIP(hipStreamSynchronize(s));
        *out = q;
        for (size_t k = q; k-- > lo;) {                                   // the last '<' or '>' before q
            if (buf[k - lo] == '>') break;
            if (buf[k - lo] == '<') {                                     // q lies inside this token: the item after it
                size_t e = k + 1;
                while (e < hi && buf[e - lo] != '>') e++;
                if (e >= hi) return c.fail(RSN_ERR_DEVICE, "lzss: a token without its end at a section cut");
                *out = std::max(q, e + 1);
                break;
            }
        }
        return RSN_OK;
    };
    size_t sec_buf = 0;
    uint8_t *d_v = nullptr;
    uint32_t b0 = 0;
    size_t c0 = 0;
    int n_sec = 0;
    while (b0 < n_cb) {
        // the section's last block: at most SEC escaped bytes and SEC compressed bytes (one block at least)
        uint32_t lo_b = b0 + 1, hi_b = (uint32_t)std::min<size_t>(n_cb, (size_t)b0 + std::max<size_t>(SEC / ZTILE, 1));
        while (lo_b < hi_b) { const uint32_t mid = lo_b + (hi_b - lo_b + 1) / 2; if (boff[mid] - boff[b0] <= SEC) lo_b = mid; else hi_b = mid - 1; }
        const uint32_t b1 = lo_b;
        size_t c1; rc = item_start((size_t)b1 * ZTILE, &c1); if (rc) return rc;
        if (b1 == n_cb) c1 = n;
        const size_t e0 = (size_t)boff[b0], e1 = (size_t)boff[b1];
        // the escaped bytes in front: the largest back-pointer's worth, a little more so that the destination is 16-byte aligned
        const size_t W = e0 <= hmax ? e0 : hmax + ((e0 - hmax) & 15);
        const size_t vn = W + (c1 - c0);
        if (vn + 64 > sec_buf) { sec_buf = vn + 64 + (vn >> 3); rc = dev_buf(c, 36, sec_buf, &p); if (rc) return rc; d_v = (uint8_t *)p; }
        if (W) RSN_HIP(hipMemcpyAsync(d_v, d_esc + (e0 - W), W, hipMemcpyDeviceToDevice, s));
        if (c1 > c0) RSN_HIP(hipMemcpyAsync(d_v + W, d_in + c0, c1 - c0, hipMemcpyDeviceToDevice, s));
        size_t got = 0;
        if (vn) {
            rc = lzss_decode_impl(c, s, d_v, vn, nullptr, 0, &got, d_esc + (e0 - W)); if (rc) return rc;
            if (got != W + (e1 - e0)) return c.fail(RSN_ERR_DEVICE, "lzss: section %d expands to %zu bytes, the counting pass says %zu", n_sec, got, W + (e1 - e0));
        }
        if (dbg) fprintf(stderr, "lzss decode section %d: blocks [%u, %u), compressed [%zu, %zu), escaped [%zu, %zu), %zu bytes in front\n", n_sec, b0, b1, c0, c1, e0, e1, W);
        b0 = b1; c0 = c1; n_sec++;
    }
    return lzss_unescape(c, s, d_esc, E, false, d_out, out_cap, out_n);
}
}  // namespace

int lzss_decode_dev(Ctx &c, hipStream_t s, const uint8_t *d_in, size_t n, uint8_t *d_out, size_t out_cap, size_t *out_n) {
    return lzss_decode_impl(c, s, d_in, n, d_out, out_cap, out_n, nullptr);
}

// ---------------------------------------------------------------- a host-buffer decode in SLICES as the stream lands (r06)
// rsn_lzss_decompress was upload (12 ms per GiB of text's 0.7 GB stream), decoder (5), download (19), one after the other.  PCIe is full
// duplex: here the stream is decoded slice by slice as its pieces land, each slice's bytes on their way down while the next ones come up
// and decode.  A slice is what lzss_decode_sections calls a section -- the escaped bytes in front of it, as many as the largest back-pointer
// so far reaches, then its tokens: the escaped stream holds no '<' (lzss.go:373-377), so the prefix parses as literals and every pointer
// finds what it points at -- expanded by the ordinary decoder into the escaped stream's buffer, then mapped FF -> '<' into the output
// (all DecodeOpeningSymbols does in a stream without a 5C, lzss.go:391-406).  What the slices cannot know is the TOTAL: the caller sizes
// the output from a sample of the stream (rsn_api.hip: the first 256 KiB, parsed on the host) and takes the serial call when a slice would overflow it (RSN_ERR_CAPACITY) -- or when a 5C turns
// up (returns 1: bytes already announced were announced unescaped).  Cuts fall on item boundaries behind 4 KiB block starts.
__global__ __launch_bounds__(256) void k_lzd_map_ff(const uint8_t *__restrict__ esc, uint8_t *__restrict__ out, unsigned long long lo, unsigned long long hi) {
    const unsigned long long b0 = ((lo >> 4) + (unsigned long long)blockIdx.x * 256 + threadIdx.x) << 4;   // this thread's 16-byte unit (both buffers are 16-byte aligned)
    if (b0 >= hi) return;
    if (b0 >= lo && b0 + 16 <= hi) {
        uint4 v = *reinterpret_cast<const uint4 *>(esc + b0);
        uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int k = 0; k < 4; k++) {                                     // FF -> 3C: clear bits 0, 1, 6, 7 of every byte that is FF
            const uint32_t t = ~w[k], z = ~(((t & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | t | 0x7F7F7F7Fu);   // 0x80 where the byte of w is FF
            w[k] &= ~((z >> 7) * 0xC3u);
        }
        *reinterpret_cast<uint4 *>(out + b0) = make_uint4(w[0], w[1], w[2], w[3]);
    } else for (int k = 0; k < 16; k++) { const unsigned long long q = b0 + k; if (q >= lo && q < hi) { const uint8_t b = esc[q]; out[q] = b == 0xFF ? (uint8_t)0x3C : b; } }
}

int lzss_decode_sliced(Ctx &c, hipStream_t s, const uint8_t *d_in, size_t n, uint8_t *d_out, size_t out_cap, size_t *out_n, const SliceStream &st) {
    *out_n = 0;
    if (n == 0 || n >= (1ull << 32) - 65536 || out_cap >= (1ull << 32) - 65536) return 1;
    void *p; int rc;
    const uint32_t n_cb = (uint32_t)ceil_div(n, (size_t)ZTILE);
    const size_t SL = std::max<size_t>(st.slice_bytes / ZTILE, 16) * ZTILE;
    rc = dev_buf(c, 15, out_cap + 64, &p); if (rc) return rc;
    uint8_t *d_esc = (uint8_t *)p;
    void *hp; rc = pinned_buf(c, 256, &hp); if (rc) return rc;
    size_t c0 = 0, e0 = 0, hmax = 0, sec_buf = 0;
    uint32_t b0 = 0;
    uint8_t *d_v = nullptr;
    while (b0 < n_cb) {
        const uint32_t b1 = (uint32_t)std::min<size_t>(n_cb, (size_t)b0 + SL / ZTILE);
        const size_t q1 = b1 == n_cb ? n : (size_t)b1 * ZTILE, avail = std::min(n, q1 + 64);   // (a token that begins in front of q1 ends before q1 + 24)
        if (!st.need_in(avail)) return c.fail(RSN_ERR_DEVICE, "lzss: the upload of a sliced call failed");
        // ---- the slice's counts: its escaped length, its largest back-pointer, is there a 5C (the decoder below validates the tokens itself)
        rc = dev_buf(c, 13, ((size_t)n_cb * 2 + 5) * 8, &p); if (rc) return rc;
        unsigned long long *d_blen = (unsigned long long *)p, *d_boff = d_blen + n_cb, *d_btot = d_boff + n_cb;
        int *d_flag = (int *)(d_btot + 1);
        rc = dev_buf(c, 27, (size_t)n_cb * ZB * 2 + (size_t)n_cb * 8 + 64, &p); if (rc) return rc;
        unsigned long long *d_need = (unsigned long long *)p; uint16_t *d_span = (uint16_t *)(d_need + n_cb);
        RSN_HIP(hipMemsetAsync(d_flag, 0, 24, s));
        RSN_LAUNCH("lzss_dec_count", k_lzd_count2, dim3(b1 - b0), dim3(ZB), 0, s, d_in, avail, d_blen, d_span, d_need, (uint32_t *)(d_flag + 2), d_flag, b0);
        rc = scan_u64(c, s, "lzss_dec_scan", d_blen + b0, d_boff + b0, b1 - b0, d_btot); if (rc) return rc;
        unsigned long long *h64 = (unsigned long long *)hp;
        volatile int *hflag = (volatile int *)(h64 + 1);
        uint8_t *hcut = (uint8_t *)(h64 + 8);
        const size_t cut_lo = q1 >= (size_t)MAXTOK ? q1 - MAXTOK : 0, cut_hi = std::min(n, q1 + MAXTOK);
        RSN_HIP(hipMemcpyAsync(h64, d_btot, 8, hipMemcpyDeviceToHost, s));
        RSN_HIP(hipMemcpyAsync((void *)hflag, d_flag, 24, hipMemcpyDeviceToHost, s));
        if (b1 < n_cb) RSN_HIP(hipMemcpyAsync(hcut, d_in + cut_lo, cut_hi - cut_lo, hipMemcpyDeviceToHost, s));
        RSN_HIP(hipStreamSynchronize(s));
        if (hflag[0] & 1) return c.fail(RSN_ERR_FORMAT, "lzss: malformed \"<ptr,len>\" token");
        if (hflag[0] & 2) return c.fail(RSN_ERR_FORMAT, "lzss: back-reference outside the decoded data (reference: slice bounds out of range, lzss.go:350)");
        if (hflag[4] || hflag[5]) return 1;                               // a 5C: the stream needs DecodeOpeningSymbols' pass; a huge token: the serial decoder's other front end
        const size_t e1 = e0 + (size_t)h64[0];
        hmax = std::max(hmax, (size_t)(uint32_t)hflag[2]);
        if (e1 > out_cap) { *out_n = round_up(e1, 16) + 16; return c.fail(RSN_ERR_CAPACITY, "lzss: the stream expands beyond the %zu bytes its sample promised", out_cap); }
        if (hmax + 16 > ((size_t)1 << 30)) return 1;
        // the first item that starts at or after q1 (lzss_decode_sections' item_start, from the bytes around the cut)
        size_t c1 = q1;
        if (b1 < n_cb) {
            for (size_t k = q1; k-- > cut_lo;) {
                if (hcut[k - cut_lo] == '>') break;
                if (hcut[k - cut_lo] == '<') {
                    size_t e = k + 1;
                    while (e < cut_hi && hcut[e - cut_lo] != '>') e++;
                    if (e >= cut_hi) return c.fail(RSN_ERR_DEVICE, "lzss: a token without its end at a slice cut");
                    c1 = std::max(q1, e + 1);
                    break;
                }
            }
        }
        // ---- the slice as a stream of its own: the escaped bytes in front (16-byte aligned destination), then its items
        const size_t W = e0 <= hmax ? e0 : hmax + ((e0 - hmax) & 15);
        const size_t vn = W + (c1 - c0);
        if (vn + 64 > sec_buf) { sec_buf = vn + 64 + (vn >> 3); rc = dev_buf(c, 36, sec_buf, &p); if (rc) return rc; d_v = (uint8_t *)p; }
        if (W) RSN_HIP(hipMemcpyAsync(d_v, d_esc + (e0 - W), W, hipMemcpyDeviceToDevice, s));
        if (c1 > c0) RSN_HIP(hipMemcpyAsync(d_v + W, d_in + c0, c1 - c0, hipMemcpyDeviceToDevice, s));
        RSN_HIP(hipMemsetAsync(d_v + vn, 0, 64, s));
        size_t got = 0;
        if (vn) {
            rc = lzss_decode_impl(c, s, d_v, vn, nullptr, 0, &got, d_esc + (e0 - W)); if (rc) return rc;
            if (got != W + (e1 - e0)) return c.fail(RSN_ERR_DEVICE, "lzss: a slice expands to %zu bytes, its counting pass says %zu", got, W + (e1 - e0));
        }
        if (e1 > e0) {
            const unsigned long long units = (((unsigned long long)e1 + 15) >> 4) - (e0 >> 4);
            RSN_LAUNCH("lzss_dec_map", k_lzd_map_ff, dim3((uint32_t)ceil_div((size_t)units, (size_t)256)), dim3(256), 0, s, (const uint8_t *)d_esc, d_out, (unsigned long long)e0, (unsigned long long)e1);
            RSN_HIP(hipStreamSynchronize(s));
            if (!st.have_out(e0, e1 - e0)) return c.fail(RSN_ERR_DEVICE, "lzss: the download of a sliced call failed");
        }
        b0 = b1; c0 = c1; e0 = e1;
    }
    *out_n = e0;
    return RSN_OK;
}

}  // namespace rsn
