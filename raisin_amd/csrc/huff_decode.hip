// huff_decode.hip -- Huffman decode for gfx950 (MI355X).
//
// Replaces huffman.Decompress (compressor/huffman/huffman.go:327): decode :258,
// decodeTree :196 (host, huff_host.cpp), findCodes :131 (the per-bit tree walk).
//
// The .rsn payload is ONE contiguous bit string with no block index, and the
// format must stay bit-exact, so the decoder is a self-synchronising parallel
// decode:
//   D1  k_dec_sync   the bit string is cut into subsequences of S bits, one per
//                    lane.  Every lane decodes from a guessed entry point and
//                    publishes where it leaves its subsequence; lanes whose
//                    predecessor's exit differs from the entry they used decode
//                    again.  Iterated inside the block (LDS) and across blocks
//                    (relaunch) until nothing changes: entry[g] == exit[g-1]
//                    for every g with entry[0] exact, i.e. the true parse.
//   D2  k_scan_u64   output offsets from the per-block byte counts
//   D3  k_dec_emit   decode again from the now-exact entries; bytes are staged in
//                    LDS and leave as 16-byte coalesced stores.
//   F   k_dec_flat   when every code has the same length L (flat alphabets: the
//                    tree is perfectly balanced) the payload is an array of L-bit
//                    fields: no synchronisation is needed, each lane unpacks 16
//                    fields and stores 16 bytes.
// Code lookup: ONE 2^K-entry table in LDS (K = min(maxlen, 11)), replicated across banks when it is small.
// For byte alphabets (every rune < 0x80) an entry describes up to three whole codewords inside the K-bit
// window -- the counting walk takes them all in one lookup, a single-symbol step uses the first -- or, when
// the first codeword is longer than K bits, the tree node the window leads to; longer codes finish with a
// bit-by-bit walk of the tree's child array (staged in LDS for byte alphabets).  One table instead of r1's
// two (single + multi) is what lets seven blocks share a CU in the counting pass instead of four: the walks
// are bound by the latency of their own dependent LDS reads, i.e. by how many wavefronts are there to hide it.
#include <chrono>
#include <optional>
#include "codecs.h"
#include "rsn_helpers.h"

namespace rsn {


constexpr int DB = 256;             // lanes per block (r04: 384 / 640 lanes load the SIMDs unevenly and lose a third, 512 ties)
constexpr int FDB = 256;            // lanes per block of k_dec_flat
constexpr int SW = 8;               // 32-bit words per subsequence (S = 256 bits; 128: emit 5 % faster, sync 40 % slower -- its warm-up is per subsequence)
constexpr int SBITS = SW * 32;
constexpr int ORG_WORDS = 8;              // words staged in front of the block: warm-up room for the entry guess
constexpr int ORG = ORG_WORDS * 32;       // block-relative bit positions are offset by this
constexpr int DATA_WORDS = ORG_WORDS + DB * SW + 8;   // + overrun for a code that starts inside and ends outside
constexpr int LUT_BITS_MAX = 11;    // index bits of the first-level table (r04: 10 bits lose 9 % on `skewed`)
constexpr int LUT_WORDS = 1 << LUT_BITS_MAX;
constexpr int OUT_STAGE = DB * 64;  // bytes of block output staged in LDS (larger blocks store directly)
constexpr int OUT_STAGE_RUNE = 40960; // the same for rune alphabets: a rune is up to four bytes (two blocks per CU instead of four, but coalesced stores)
constexpr int CHILD_LDS = 256;      // child[] of a byte alphabet (<= 2 * 127 entries) is staged in LDS
constexpr int LUT2_LDS = 768;       // second-level entries that fit in LDS (larger second levels are read through L2)
constexpr uint32_t LUT2_GLOBAL_MAX = 1u << 20;   // entries of a second level kept in device memory (4 MB)
constexpr int SYNC_ROUNDS = 24;     // rounds of a block's fixed point before it is called stuck (text: two or three; a stretch in a second phase: one per lane)
constexpr uint32_t BAD_REL = 0xFFFF;
constexpr uint32_t BAD_POS = 0xFFFFFFFFu;

struct DecArgs {
    const uint8_t *base;        // 16-byte aligned pointer at or before the first payload byte
    size_t nbytes;              // readable bytes from base
    unsigned long long p0;      // bit position (from base) of the first code bit
    unsigned long long end;     // bit position one past the last payload bit
    uint32_t n_sub;             // number of subsequences
    const uint32_t *lut; int K; int rep_log2;   // (1<<K) entries, each replicated 1<<rep_log2 times in LDS
    const int32_t *child;       // 2 per internal node: >=0 internal index, <0 -(rune+1)
    uint32_t min_len; int flat_guess; int warm;   // warm: bits decoded ahead of a subsequence so that the guess self-synchronises
    uint16_t *exit_rel, *entry_rel, *nbyte;
    unsigned long long *blk_bytes;
    int *changed; int pass;
    int *stuck;                 // set by a block whose lanes are still handing corrections on after SYNC_ROUNDS rounds: the stream is settled by k_dec_phase instead
    const uint32_t *fix_list, *fix_count;   // pass > 0: the blocks whose first entry is not their predecessor's exit (k_dec_fix_list); null: every block looks for itself
    const unsigned long long *blk_off; uint8_t *out;   // D3
    uint32_t child_n;           // entries of child[]
    const uint32_t *lut2; uint32_t lut2_n;   // second level: sub-tables for the K-bit prefixes that lead inside the tree
    unsigned long long out_cap;
};

__device__ __forceinline__ uint32_t swz(uint32_t j) { return j + (j >> 5); }
// Subsequence image: every 8 payload words are followed by a COPY of the next word, so the two
// dwords that hold any 32-bit window are always adjacent (one ds_read2_b32), and a lane stride of
// 9 words puts the 32 lanes of a half-wavefront on 32 different banks.
__device__ __forceinline__ uint32_t grp(uint32_t w) { return w + (w >> 3); }
constexpr int DATA_PHYS = (ORG_WORDS + DB * SW + 8) / 8 * 9 + 2;

__device__ __forceinline__ void stage_lut(const DecArgs &a, uint32_t *s_lut) {
    const int total = (1 << a.K) << a.rep_log2;
    for (int i = threadIdx.x; i < total; i += DB) s_lut[i] = a.lut[i >> a.rep_log2];
}

// Stages payload words [w0, w0+DATA_WORDS) into the grouped LDS image, LSB-FIRST: stream bit k of a word sits at bit k
// (bytes in memory order, the bits of every byte reversed), so that the 32 stream bits from any position are one
// v_alignbit_b32 of two adjacent dwords and a table index is a plain AND -- the tables are stored bit-reversed to match.
__device__ __forceinline__ uint32_t lsb_first(uint32_t le_word) { return __builtin_bitreverse32(__builtin_bswap32(le_word)); }
// Interior blocks take a branch-free path: all 16-byte loads are issued before the first use.
__device__ __forceinline__ void put_word(uint32_t *s_data, uint32_t j, uint32_t v) {
    const uint32_t p = grp(j);
    s_data[p] = v;
    if ((j & 7) == 0 && j) s_data[p - 1] = v;      // the copy that closes the previous group
}

// staged word i is payload word w0+i; w0*4 is a multiple of 16 (w0 < 0: the words before the base read as zero)
__device__ __forceinline__ void stage_words(const DecArgs &a, const long long w0, uint32_t *s_data) {
    static_assert(DATA_WORDS % 4 == 0, "staged in 16-byte units");
    constexpr int NV = DATA_WORDS / 4, PER = (NV + DB - 1) / DB;
    if (w0 >= 0 && (size_t)(w0 + DATA_WORDS) * 4 <= a.nbytes) {
        const uint4 *src = reinterpret_cast<const uint4 *>(a.base + (size_t)w0 * 4);
        uint4 v[PER];
#pragma unroll
        for (int k = 0; k < PER; k++) { const int idx = threadIdx.x + k * DB; if (idx < NV) v[k] = src[idx]; }
#pragma unroll
        for (int k = 0; k < PER; k++) {
            const int idx = threadIdx.x + k * DB;
            if (idx < NV) {
                const uint32_t j = 4 * idx, pj = grp(j);   // 4 consecutive words never straddle a group
                const uint32_t x = lsb_first(v[k].x);
                s_data[pj] = x; s_data[pj + 1] = lsb_first(v[k].y);
                s_data[pj + 2] = lsb_first(v[k].z); s_data[pj + 3] = lsb_first(v[k].w);
                if ((j & 7) == 0 && j) s_data[pj - 1] = x;
            }
        }
        return;
    }
    for (int i = threadIdx.x; i < DATA_WORDS; i += DB) {
        const long long w = w0 + i;
        uint32_t v = 0;
        if (w >= 0) {
            const size_t off = (size_t)w * 4;
            if (off + 4 <= a.nbytes) v = *reinterpret_cast<const uint32_t *>(a.base + off);
            else for (int k = 0; k < 4; k++) if (off + k < a.nbytes) v |= (uint32_t)a.base[off + k] << (8 * k);
        }
        put_word(s_data, (uint32_t)i, lsb_first(v));
    }
}
__device__ __forceinline__ void stage_data(const DecArgs &a, uint32_t blk, uint32_t *s_data) {
    stage_words(a, (long long)blk * DB * SW - ORG_WORDS, s_data);
}

__device__ __forceinline__ int dev_utf8_len(uint32_t r) { return r < 0x80 ? 1 : r < 0x800 ? 2 : r < 0x10000 ? 3 : 4; }

// The 32 stream bits that start at block-relative bit `pos` (the first one in bit 0): two adjacent dwords, one funnel shift.
__device__ __forceinline__ uint32_t window32(const uint32_t *s_data, uint32_t pos) {
    const uint32_t p = grp(pos >> 5);
    return __builtin_amdgcn_alignbit(s_data[p + 1], s_data[p], pos & 31);
}
// The same for a walk that stays inside ONE group of eight words (a lane inside its own subsequence, or inside the one before it
// during the warm-up): there grp(w) = w + g with g the group's number, so a position that carries 32 g on top -- sp = pos + 32 g,
// the same low five bits -- gives the dword address with one shift and one mask instead of grp()'s five instructions.
__device__ __forceinline__ uint32_t window32_sp(const uint32_t *s_data, uint32_t sp) {
    const uint32_t *w = reinterpret_cast<const uint32_t *>(reinterpret_cast<const uint8_t *>(s_data) + ((sp >> 3) & ~3u));
    return __builtin_amdgcn_alignbit(w[1], w[0], sp);                     // (v_alignbit_b32 shifts by the low five bits)
}
__device__ __forceinline__ uint32_t bit_at(const uint32_t *s_data, uint32_t q) { return (s_data[grp(q >> 5)] >> (q & 31)) & 1; }
// Index of the first-level table entry for a window (the lane's own copy of it when the table is replicated).
__device__ __forceinline__ uint32_t lut_index(const DecArgs &a, uint32_t win, uint32_t lane_r) { return ((win & ((1u << a.K) - 1u)) << a.rep_log2) | lane_r; }
// The entry itself, addressed in BYTES (index and copy number shifted once: an AND and a shift-OR per lookup, not three instructions).
__device__ __forceinline__ uint32_t lut_at(const DecArgs &a, const uint32_t *s_lut, uint32_t win, uint32_t lane_r) {
    const uint32_t off = ((win & ((1u << a.K) - 1u)) << (a.rep_log2 + 2)) | (lane_r << 2);
    return *reinterpret_cast<const uint32_t *>(reinterpret_cast<const uint8_t *>(s_lut) + off);
}

// One codeword at `pos`: returns the rune and advances pos.  No state is carried between
// symbols, so the loop has no refill branch: a wavefront never diverges inside a step.
// Table entries.  Rune alphabets:  len << 24 | rune, or a LONG entry when the code is longer than K bits.
// Byte alphabets (unified):        sym1 | sym2 << 8 | sym3 << 16 | bits of all n << 24 | n << 28, n = 1..3 whole codewords inside
//                                  the window, or a LONG entry.  The symbols are 7 bits wide and sit where the output bytes
//                                  do (r04: e & 0x7F7F7F IS the three bytes -- six instructions of unpacking per lookup less
//                                  in the writing walk); the bits of the FIRST codeword, which only the single steps at a
//                                  subsequence's end read, are spread over what is left: bits 7, 15, 23 and 30.
// LONG entry:                      0x80000000 | sb << 26 | x.  sb > 0: the next sb bits of the window index the
//                                  2^sb-entry sub-table at lut2[x] (one more dependent lookup instead of one per bit);
//                                  sb == 0: x is the tree node the K bits lead to, walked bit by bit.
// Second-level entries:            len << 24 | rune (len counts all bits of the code), or 0x80000000 | node when the
//                                  code is longer than K + sb bits still (the bit-by-bit walk goes on from there).
__host__ __device__ __forceinline__ uint32_t u_used(uint32_t e) { return (e >> 24) & 15u; }
__host__ __device__ __forceinline__ uint32_t u_n(uint32_t e) { return (e >> 28) & 3u; }
__host__ __device__ __forceinline__ uint32_t u_len1(uint32_t e) { return ((e >> 7) & 1u) | ((e >> 14) & 2u) | ((e >> 21) & 4u) | ((e >> 27) & 8u); }
__host__ __device__ __forceinline__ uint32_t u_bytes(uint32_t e) { return e & 0x007F7F7Fu; }
__host__ __device__ __forceinline__ uint32_t u_entry(uint32_t syms, uint32_t used, uint32_t n, uint32_t len1) {   // syms: sym1 | sym2 << 8 | sym3 << 16
    return syms | (used << 24) | (n << 28) | ((len1 & 1u) << 7) | ((len1 & 2u) << 14) | ((len1 & 4u) << 21) | ((len1 & 8u) << 27);
}

struct Lut2 { const uint32_t *lds; bool in_lds; };   // where the second level is (uniform per launch)

// A codeword longer than K bits: `ent` is the LONG entry its first K bits led to, `win` the 32 stream bits from pos.
__device__ __forceinline__ uint32_t decode_long(const DecArgs &a, const uint32_t *s_data, uint32_t win, uint32_t ent, uint32_t &pos, const Lut2 &l2) {
    const uint32_t sb = (ent >> 26) & 31u;
    int32_t node = (int32_t)(ent & 0x3FFFFFFu);
    uint32_t l = a.K, rune = 0;
    if (sb) {
        const uint32_t i2 = (ent & 0x3FFFFFFu) + __builtin_amdgcn_ubfe(win, (uint32_t)a.K, sb);
        const uint32_t e2 = l2.in_lds ? l2.lds[i2] : a.lut2[i2];
        if (!(e2 & 0x80000000u)) { pos += e2 >> 24; return e2 & 0x1FFFFFu; }
        node = (int32_t)(e2 & 0x7FFFFFFFu);
        l += sb;
    }
    for (;;) {
        const int32_t nxt = a.child[2 * node + bit_at(s_data, pos + l)];
        l++;
        if (nxt < 0) { rune = (uint32_t)(-(nxt + 1)); break; }
        if (l >= 64) { l = 65; break; }   // cannot happen for a tree accepted by the host (codes <= 64 bits)
        node = nxt;
    }
    pos += l;
    return rune;
}

template <bool ASCII, bool SHORT>
__device__ __forceinline__ uint32_t decode_one(const DecArgs &a, const uint32_t *s_data, const uint32_t *s_lut, uint32_t lane_r, uint32_t &pos, const Lut2 &l2) {
    const uint32_t win = window32(s_data, pos);
    const uint32_t ent = lut_at(a, s_lut, win, lane_r);
    if (SHORT || !(ent & 0x80000000u)) {
        if (ASCII) { pos += u_len1(ent); return ent & 0x7Fu; }
        pos += ent >> 24;
        return ent & 0x1FFFFFu;
    }
    return decode_long(a, s_data, win, ent, pos, l2);
}

template <bool ASCII, bool SHORT>
__device__ __forceinline__ uint32_t decode_one_sp(const DecArgs &a, const uint32_t *s_data, const uint32_t *s_lut, uint32_t lane_r, uint32_t &sp, uint32_t spo, const Lut2 &l2) {
    const uint32_t win = window32_sp(s_data, sp);
    const uint32_t ent = lut_at(a, s_lut, win, lane_r);
    if (SHORT || !(ent & 0x80000000u)) {
        if (ASCII) { sp += u_len1(ent); return ent & 0x7Fu; }
        sp += ent >> 24;
        return ent & 0x1FFFFFu;
    }
    uint32_t pos = sp - spo;
    const uint32_t r = decode_long(a, s_data, win, ent, pos, l2);
    sp = pos + spo;
    return r;
}

// Walk from block-relative bit `pos` to the first code boundary >= lim.  A code that runs past
// the end of the payload can only be the last one of the walk: checked once, after the loop.
template <bool ASCII, bool SHORT, bool MULTI>
__device__ __forceinline__ uint32_t advance(const DecArgs &a, const uint32_t *s_data, const uint32_t *s_lut, uint32_t lane_r, const Lut2 &l2,
                                            uint32_t &pos, uint32_t lim, uint32_t grp_no) {
    // grp_no: the group of eight staged words every codeword of this walk STARTS in (positions [256 grp_no, 256 grp_no + 256) of the
    // block image): the walk runs on sp = pos + 32 grp_no, see window32_sp
    const uint32_t spo = 32u * grp_no, slim = lim + spo;
    uint32_t sp = pos + spo;
    uint32_t nb = 0;
    if (ASCII && MULTI) {
        // up to 3 codewords per table lookup while a whole K-bit step stays inside the subsequence
        const uint32_t K = (uint32_t)a.K, safe = lim >= K ? slim - K : 0;
        // ... and two lookups per 32-bit window (r04): the first takes <= K <= 11 bits, so >= 21 bits of the window are left for the
        // second -- one window fetch (address arithmetic, an LDS round trip, the funnel shift) and one loop turn per two lookups
        while (sp <= safe && lim >= K) {
            const uint32_t win = window32_sp(s_data, sp);
            const uint32_t e = lut_at(a, s_lut, win, lane_r);
            if (!SHORT && __ballot((e & 0x80000000u) != 0)) {             // first code longer than K bits, somewhere in the wavefront (rare: a scalar branch around it)
                if (e & 0x80000000u) { uint32_t q = sp - spo; (void)decode_long(a, s_data, win, e, q, l2); sp = q + spo; nb++; continue; }
            }
            const uint32_t u1 = u_used(e);
            sp += u1; nb += u_n(e);
            if (sp <= safe) {
                const uint32_t e2 = lut_at(a, s_lut, win >> u1, lane_r);
                if (SHORT || !(e2 & 0x80000000u)) { sp += u_used(e2); nb += u_n(e2); }      // (a long code waits for the next turn's fresh window)
            }
        }
    }
    while (sp < slim) {
        const uint32_t r = decode_one_sp<ASCII, SHORT>(a, s_data, s_lut, lane_r, sp, spo, l2);
        nb += ASCII ? 1 : dev_utf8_len(r);
    }
    pos = sp - spo;
    return nb;
}

template <bool ASCII, bool SHORT, bool MULTI>
__device__ __forceinline__ void walk(const DecArgs &a, const uint32_t *s_data, const uint32_t *s_lut, uint32_t lane_r, const Lut2 &l2,
                                     uint32_t pos, uint32_t lim, uint32_t grp_no, uint32_t end_rel, uint32_t *exit_pos, uint32_t *nbytes) {
    *nbytes = advance<ASCII, SHORT, MULTI>(a, s_data, s_lut, lane_r, l2, pos, lim, grp_no);
    *exit_pos = pos > end_rel ? BAD_POS : pos;
}

template <bool ASCII, bool SHORT, bool MULTI>
__global__ __launch_bounds__(DB) void k_dec_sync(DecArgs a, uint32_t n_blk) {
    __shared__ __attribute__((aligned(1024))) uint32_t s_data[DATA_PHYS];   // (the alignment puts it first in the LDS: its address is an instruction offset, not an add)
    __shared__ uint32_t s_lut[LUT_WORDS];
    __shared__ uint32_t s_exit[DB];
    __shared__ unsigned long long s_part[DB / 64];
    __shared__ int s_skip;
    __shared__ int32_t s_child[SHORT ? 1 : CHILD_LDS];
    __shared__ uint32_t s_lut2[SHORT ? 1 : LUT2_LDS];
    const int tid = threadIdx.x;
    const uint32_t lane_r = tid & ((1u << a.rep_log2) - 1);
    const uint32_t my0 = ORG + tid * SBITS;
    // a fixing pass works through a list (r04: 74 000 blocks of a 1 GiB `skewed` stream each staging the tables to find out that they have
    // nothing to do was 0.07 ms of a 1.7 ms decode; the list holds a few dozen)
    const uint32_t n_work = a.fix_list ? *a.fix_count : n_blk;
    if (blockIdx.x >= n_work) return;
    stage_lut(a, s_lut);   // once per (persistent) block
    if (!SHORT && a.child_n <= (uint32_t)CHILD_LDS) { for (uint32_t i = tid; i < a.child_n; i += DB) s_child[i] = a.child[i]; a.child = s_child; }
    const Lut2 l2{s_lut2, a.lut2_n <= (uint32_t)LUT2_LDS};
    if (!SHORT && l2.in_lds) for (uint32_t i = tid; i < a.lut2_n; i += DB) s_lut2[i] = a.lut2[i];
    for (uint32_t wi = blockIdx.x; wi < n_work; wi += gridDim.x) {
        const uint32_t blk = a.fix_list ? a.fix_list[wi] : wi;
        const unsigned long long blk_bit0 = (unsigned long long)blk * DB * SBITS;
        const uint32_t g = blk * DB + tid;
        const bool live = g < a.n_sub;
        // block-relative positions (the block spans DB*SBITS bits; exits overshoot by < 64)
        const uint32_t end_rel = (uint32_t)min(a.end - blk_bit0 + ORG, (unsigned long long)(ORG + DB * SBITS + 4096));
        const uint32_t lim = min(my0 + SBITS, end_rel);
        __syncthreads();   // previous iteration is done with s_skip / s_data / s_part
        // lane 0's entry: the true start for block 0, else the predecessor block's published exit
        if (tid == 0) {
            int skip = 0;
            if (a.pass > 0) {
                unsigned long long e = a.p0;
                if (blk > 0) {
                    const uint32_t xr = a.exit_rel[g - 1];
                    e = xr == BAD_REL ? ~0ull : blk_bit0 + xr;
                }
                const uint32_t er = a.entry_rel[g];
                skip = er == BAD_REL ? e == ~0ull : e == blk_bit0 + er;
            }
            // (r06) some block has called the stream stuck: k_dec_phase settles it and rewrites the three arrays, so what this block would
            // work out is not used -- it leaves "no entry, no bytes" (what the emit pass queued behind skips) and goes on: runs of a byte
            // are a bit string with the code length for a period, every block of which ran its SYNC_ROUNDS rounds first (0.84 + 0.68 ms
            // per 64 MiB in the two passes; text: 0.06 + 0.02)
            if (__atomic_load_n(a.stuck, __ATOMIC_RELAXED)) skip = 2;
            s_skip = skip;
        }
        __syncthreads();
        if (s_skip == 2) {
            if (live) { a.entry_rel[g] = (uint16_t)BAD_REL; a.exit_rel[g] = (uint16_t)BAD_REL; a.nbyte[g] = 0; }
            if (tid == 0) a.blk_bytes[blk] = 0;
            continue;
        }
        if (s_skip) continue;
        stage_data(a, blk, s_data);
        __syncthreads();   // the warm-up guess below reads the staged words

        uint32_t e;   // block-relative entry; BAD_POS = the predecessor ran off the end
        if (!live) e = BAD_POS;
        else if (g == 0) e = (uint32_t)a.p0 + ORG;
        else if (a.pass == 0 || tid > 0) {
            const long long p0_rel = (long long)a.p0 - (long long)blk_bit0 + ORG;
            if (p0_rel >= (long long)my0) e = (uint32_t)p0_rel;          // the stream starts inside/after this subsequence
            else if (a.flat_guess) {
                // codes of one length: boundaries are phase-locked to p0 (exact)
                const unsigned long long d = (unsigned long long)((long long)my0 - p0_rel) + a.min_len - 1;   // > 2^32 on large inputs
                e = (uint32_t)(p0_rel + (long long)(d - d % a.min_len));
            } else {
                // decode from WARM bits ahead of the subsequence: prefix codes self-synchronise within a few
                // codewords, so the first boundary at or after my0 is very likely the true entry
                const uint32_t start = (uint32_t)max(p0_rel, (long long)my0 - a.warm);
                uint32_t q = start;
                (void)advance<ASCII, SHORT, MULTI>(a, s_data, s_lut, lane_r, l2, q, my0, (uint32_t)tid);   // (several codewords per lookup here too; the group before the lane's own)
                e = q;
            }
        } else {
            const uint32_t xr = a.exit_rel[g - 1];
            e = xr == BAD_REL ? BAD_POS : ORG + xr;
        }
        bool have = false;
        uint32_t x = 0, nb = 0;
        if (live && a.pass > 0 && tid > 0) {
            // results of the previous pass stay valid while the entry they were computed from stands
            const uint32_t er = a.entry_rel[g];
            const uint32_t xr = a.exit_rel[g];
            e = er == BAD_REL ? BAD_POS : my0 + er;
            x = xr == BAD_REL ? BAD_POS : lim + xr;
            nb = a.nbyte[g];
            have = true;
        }
        __syncthreads();
        for (int round = 0; round <= DB; round++) {
            if (live && !have) {
                if (e == BAD_POS) { x = BAD_POS; nb = 0; }
                else walk<ASCII, SHORT, MULTI>(a, s_data, s_lut, lane_r, l2, e, lim, (uint32_t)tid + 1u, end_rel, &x, &nb);
                have = true;
            }
            s_exit[tid] = x;
            __syncthreads();
            bool changed = false;
            if (live && tid > 0) {
                const uint32_t en = s_exit[tid - 1];
                if (en != e) { e = en; have = false; changed = true; }
            }
            if (!__syncthreads_or(changed)) break;
            if (round == SYNC_ROUNDS) { if (tid == 0) *a.stuck = 1; break; }   // (what the lanes hold is then not one parse: the host does not use it)
        }
        // A lane that has just taken a new entry at the stuck break holds the OLD entry's exit and byte count: stored as they are, the emit
        // pass that is queued before the host has read `stuck` would walk from the new entry and could write more bytes than the scan
        // reserved for the lane -- past out_cap in the last block of an exact-fit buffer.  Such a lane stores "no entry, no
        // bytes", which the emit pass skips; k_dec_phase rewrites all three arrays before anything is final.
        if (live && !have) { e = BAD_POS; x = BAD_POS; nb = 0; }
        if (live) {
            const uint16_t new_exit = x == BAD_POS ? (uint16_t)BAD_REL : (uint16_t)(x - lim);
            // Another pass is needed only if some block hands its successor a different exit than before: a block that decoded
            // again from a corrected entry and re-synchronised inside itself changes nobody else's parse (only its own byte
            // count, which the scan picks up).  So the usual decode is pass 0 + one fixing pass, without a third that only verifies.
            if (a.pass > 0 && tid == DB - 1 && a.exit_rel[g] != new_exit) *a.changed = 1;
            a.entry_rel[g] = e == BAD_POS ? BAD_REL : (uint16_t)(e - my0);
            a.exit_rel[g] = new_exit;
            a.nbyte[g] = (uint16_t)nb;
        }
        unsigned long long sum = live ? nb : 0;
        for (int d = 32; d; d >>= 1) sum += __shfl_down(sum, d);
        if ((tid & 63) == 0) s_part[tid >> 6] = sum;
        __syncthreads();
        if (tid == 0) {
            unsigned long long all = 0;
            for (int k = 0; k < DB / 64; k++) all += s_part[k];
            a.blk_bytes[blk] = all;
        }
    }
}

// The blocks a fixing pass has to decode again: those whose first lane's entry is not the exit the block before has published (the
// test k_dec_sync makes for itself when it has no list).  Block 0 starts from the exact entry.
__global__ __launch_bounds__(256) void k_dec_fix_list(const uint16_t *__restrict__ exit_rel, const uint16_t *__restrict__ entry_rel, uint32_t n_blk,
                                                      uint32_t *__restrict__ list, uint32_t *__restrict__ count) {
    const uint32_t b = blockIdx.x * 256 + threadIdx.x;
    const bool need = b >= 1 && b < n_blk && exit_rel[(size_t)b * DB - 1] != entry_rel[(size_t)b * DB];
    const unsigned long long m = __ballot(need);
    if (!m) return;
    const int lane = threadIdx.x & 63;
    uint32_t base = 0;
    if (lane == 0) base = atomicAdd(count, (uint32_t)__builtin_popcountll(m));
    base = (uint32_t)__builtin_amdgcn_readfirstlane((int)base);
    if (need) list[base + (uint32_t)__builtin_popcountll(m & ((1ull << lane) - 1ull))] = b;
}

// ---------------------------------------------------------------- the synchronisation when it does not converge: every entry of every lane
// k_dec_sync's fixed point hands a correction on lane by lane inside a block and block by block from launch to launch.  That is two or
// three rounds on data whose parses merge -- and as many rounds as a stretch has lanes, as many launches as it has blocks, where they do
// not: PERIODIC data can be parsed in a second phase for as long as the period lasts, and no guess made inside the stretch ever lands in
// the true one (r05: 4 MiB of a 20-byte UTF-8 unit repeated took 474 ms, 160 launches of up to 256 rounds; found by
// scripts/probes/periodic_decode.py).  When the fixing passes have not settled a stream this kernel does, in two launches whatever the
// length and with nothing left to chance: a codeword that ends in a subsequence begins less than the longest code's length C before it, so
// a lane can only be entered at one of its first C bits -- the kernel decodes EVERY lane from EVERY one of them (C times a pass's work:
// the price of a stream that does not synchronise), a table "lane t entered at bit c -> left at x, n bytes" in LDS, 64 lanes at a time;
//   mode 0  C lanes follow the C possible entries of the block through the table: "block entered at c -> left at x, n bytes" (to the host,
//           which chains the blocks: block 0 is entered at the stream's first code bit)
//   mode 2  one lane follows the true entry and writes what k_dec_sync would have left: entry_rel, exit_rel, nbyte, blk_bytes -- the
//           offsets and the emit pass then run as always
constexpr uint32_t PH_CAND = 64;      // a code has at most 64 bits (huff_host.cpp refuses longer ones)
struct PhaseArgs {
    uint2 *blk_map;                   // [PH_CAND * n_blk]: the block entered at c: (exit, bytes); bytes = 0xFFFFFFFF: no such path
    const uint8_t *true_c;            // mode 2: the block's true entry; 0xFF: the stream ran off its payload before this block
    uint32_t cand;                    // C: the longest code's length, at most PH_CAND
};

template <bool ASCII, bool SHORT, bool MULTI>
__global__ __launch_bounds__(DB) void k_dec_phase(DecArgs a, PhaseArgs p, uint32_t n_blk, int mode) {
    __shared__ __attribute__((aligned(1024))) uint32_t s_data[DATA_PHYS];
    __shared__ uint32_t s_lut[LUT_WORDS];
    __shared__ int32_t s_child[SHORT ? 1 : CHILD_LDS];
    __shared__ uint32_t s_lut2[SHORT ? 1 : LUT2_LDS];
    __shared__ uint32_t s_T[64 * PH_CAND];           // [lane of the group][entry bit]: exit | bytes << 16
    const int tid = threadIdx.x;
    const uint32_t lane = tid & 63, wave = tid >> 6;
    const uint32_t lane_r = tid & ((1u << a.rep_log2) - 1);
    stage_lut(a, s_lut);
    if (!SHORT && a.child_n <= (uint32_t)CHILD_LDS) { for (uint32_t i = tid; i < a.child_n; i += DB) s_child[i] = a.child[i]; a.child = s_child; }
    const Lut2 l2{s_lut2, a.lut2_n <= (uint32_t)LUT2_LDS};
    if (!SHORT && l2.in_lds) for (uint32_t i = tid; i < a.lut2_n; i += DB) s_lut2[i] = a.lut2[i];
    for (uint32_t blk = blockIdx.x; blk < n_blk; blk += gridDim.x) {
        const unsigned long long blk_bit0 = (unsigned long long)blk * DB * SBITS;
        const uint32_t L = min((uint32_t)DB, a.n_sub - blk * DB);          // lanes of this block that have a subsequence
        const uint32_t end_rel = (uint32_t)min(a.end - blk_bit0 + ORG, (unsigned long long)(ORG + DB * SBITS + 4096));
        const uint32_t C = p.cand;
        __syncthreads();                                                    // (the previous block of this loop is done with the LDS)
        stage_data(a, blk, s_data);
        // the followers: mode 0 lane c < C is the block entered at bit c; mode 2 lane 0 is the block entered at its true bit
        uint32_t e = mode == 0 ? (uint32_t)tid : (uint32_t)p.true_c[blk];
        if (mode == 2 && e == 0xFF) e = BAD_REL;
        if (blk == 0 && mode == 0 && tid > 0) e = 0x10000u;                 // (block 0 has one entry, the stream's first code bit: the table's bit 0 of its lane 0)
        uint32_t bytes = 0;
        const bool follower = mode == 0 ? (uint32_t)tid < C : tid == 0;
        for (uint32_t grp0 = 0; grp0 < L; grp0 += 64) {
            __syncthreads();                                                // (the followers are done with the table of the group before; the first time: the staging)
            // ---- the table of lanes grp0 .. grp0 + 63: a wavefront takes every fourth entry bit of all of them
            const uint32_t t = grp0 + lane;                                 // the lane of the block this thread decodes for
            if (t < L) {
                const uint32_t t0 = ORG + t * SBITS, lim = min(t0 + (uint32_t)SBITS, end_rel);
                const uint32_t first = blk == 0 && t == 0 ? (uint32_t)a.p0 : 0u;   // (the stream begins inside block 0's lane 0)
                for (uint32_t c0 = wave; c0 < C; c0 += DB / 64) {
                    uint32_t x = BAD_POS, nbs = 0;
                    walk<ASCII, SHORT, MULTI>(a, s_data, s_lut, lane_r, l2, t0 + first + c0, lim, t + 1u, end_rel, &x, &nbs);
                    s_T[lane * PH_CAND + c0] = (x == BAD_POS ? (uint32_t)BAD_REL : x - lim) | nbs << 16;
                }
            }
            __syncthreads();
            if (follower) {
                const uint32_t n_here = min(64u, L - grp0);
                for (uint32_t k = 0; k < n_here; k++) {
                    const uint32_t g = blk * DB + grp0 + k;
                    uint32_t x = BAD_REL, nbs = 0;
                    if (e < C) { const uint32_t v = s_T[k * PH_CAND + e]; x = v & 0xFFFFu; nbs = v >> 16; }
                    else if (e != BAD_REL) e = 0x10000u;                    // an entry that cannot be: no path (mode 0), an internal error (mode 2)
                    if (mode == 2) {
                        a.entry_rel[g] = (uint16_t)(e == BAD_REL || e >= C ? (uint32_t)BAD_REL : (blk == 0 && grp0 + k == 0 ? (uint32_t)a.p0 : e));
                        a.exit_rel[g] = (uint16_t)x; a.nbyte[g] = (uint16_t)nbs;
                    }
                    bytes += nbs;
                    if (e < C || e == BAD_REL) e = x;                       // (BAD stays BAD: x is BAD_REL then)
                }
            }
        }
        if (follower) {
            if (mode == 0) p.blk_map[(size_t)blk * PH_CAND + tid] = make_uint2(e & 0xFFFFu, e > 0xFFFFu ? 0xFFFFFFFFu : bytes);
            else a.blk_bytes[blk] = bytes;
        }
    }
}

// Per-lane byte sink for the direct (unstaged) path: aligned 8-byte stores, byte stores
// only for the partial first/last word.
struct Sink {
    uint8_t *wordp; unsigned long long acc; uint32_t cnt, lo;
    __device__ __forceinline__ void start(uint8_t *p) { lo = (uint32_t)((uintptr_t)p & 7); wordp = p - lo; cnt = lo; acc = 0; }
    __device__ __forceinline__ void put(uint32_t b) {
        acc |= (unsigned long long)b << (8 * cnt);
        if (++cnt == 8) {
            if (lo == 0) *reinterpret_cast<unsigned long long *>(wordp) = acc;
            else for (uint32_t k = lo; k < 8; k++) wordp[k] = (uint8_t)(acc >> (8 * k));
            wordp += 8; acc = 0; cnt = 0; lo = 0;
        }
    }
    __device__ __forceinline__ void finish() { for (uint32_t k = lo; k < cnt; k++) wordp[k] = (uint8_t)(acc >> (8 * k)); }
};

template <class Put>
__device__ __forceinline__ void put_rune(uint32_t rune, Put put) {   // string(rune) (huffman.go:138)
    if (rune < 0x80) put(rune);
    else if (rune < 0x800) { put(0xC0 | (rune >> 6)); put(0x80 | (rune & 0x3F)); }
    else if (rune < 0x10000) { put(0xE0 | (rune >> 12)); put(0x80 | ((rune >> 6) & 0x3F)); put(0x80 | (rune & 0x3F)); }
    else { put(0xF0 | (rune >> 18)); put(0x80 | ((rune >> 12) & 0x3F)); put(0x80 | ((rune >> 6) & 0x3F)); put(0x80 | (rune & 0x3F)); }
}

// Decode loop of D3: entries are exact, so the walk never runs off the payload.
template <bool ASCII, bool SHORT, class Put>
__device__ __forceinline__ void emit_walk(const DecArgs &a, const uint32_t *s_data, const uint32_t *s_lut,
                                          uint32_t lane_r, const Lut2 &l2, uint32_t pos, uint32_t lim, Put put) {
    while (pos < lim) {
        const uint32_t r = decode_one<ASCII, SHORT>(a, s_data, s_lut, lane_r, pos, l2);
        if (ASCII) put(r); else put_rune(r, put);
    }
}

#define RSN_EMIT_WORD(p, v) atomicOr((p), (v))
template <bool ASCII, bool SHORT, bool MULTI>
__global__ __launch_bounds__(DB) void k_dec_emit(DecArgs a, uint32_t n_blk) {
    __shared__ __attribute__((aligned(1024))) uint32_t s_data[DATA_PHYS];   // (the alignment puts it first in the LDS: its address is an instruction offset, not an add)
    __shared__ uint32_t s_lut[LUT_WORDS];
    constexpr int STAGE = ASCII ? OUT_STAGE : OUT_STAGE_RUNE;
    __shared__ __attribute__((aligned(16))) uint8_t s_out[STAGE + 32];
    __shared__ uint32_t s_wsum[DB / 64];
    __shared__ int32_t s_child[SHORT ? 1 : CHILD_LDS];
    __shared__ uint32_t s_lut2[SHORT ? 1 : LUT2_LDS];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const uint32_t lane_r = tid & ((1u << a.rep_log2) - 1);
    const uint32_t my0 = ORG + tid * SBITS;
    stage_lut(a, s_lut);
    if (!SHORT && a.child_n <= (uint32_t)CHILD_LDS) { for (uint32_t i = tid; i < a.child_n; i += DB) s_child[i] = a.child[i]; a.child = s_child; }
    const Lut2 l2{s_lut2, a.lut2_n <= (uint32_t)LUT2_LDS};
    if (!SHORT && l2.in_lds) for (uint32_t i = tid; i < a.lut2_n; i += DB) s_lut2[i] = a.lut2[i];
    if (ASCII) for (int i = tid; i < (STAGE + 32) / 16; i += DB) reinterpret_cast<uint4 *>(s_out)[i] = make_uint4(0, 0, 0, 0);
    for (uint32_t blk = blockIdx.x; blk < n_blk; blk += gridDim.x) {
        const unsigned long long blk_bit0 = (unsigned long long)blk * DB * SBITS;
        const uint32_t g = blk * DB + tid;
        const bool live = g < a.n_sub;
        const uint32_t end_rel = (uint32_t)min(a.end - blk_bit0 + ORG, (unsigned long long)(ORG + DB * SBITS + 4096));
        const uint32_t lim = min(my0 + SBITS, end_rel);
        __syncthreads();   // previous iteration has drained s_out / s_data / s_wsum
        stage_data(a, blk, s_data);
        const uint32_t nb = live ? a.nbyte[g] : 0;
        const uint32_t er = live ? a.entry_rel[g] : BAD_REL;
        uint32_t incl = nb;
        for (int d = 1; d < 64; d <<= 1) { const uint32_t y = __shfl_up(incl, d); if (lane >= d) incl += y; }
        if (lane == 63) s_wsum[wv] = incl;
        __syncthreads();
        uint32_t wpre = 0, total = 0;
        for (int k = 0; k < DB / 64; k++) { if (k < wv) wpre += s_wsum[k]; total += s_wsum[k]; }
        const uint32_t my_off = wpre + incl - nb;                 // byte offset of this lane inside the block's output
        if (a.blk_off[blk] + total > a.out_cap) continue;         // uniform per block: the host reports RSN_ERR_CAPACITY from the total
        uint8_t *dst = a.out + a.blk_off[blk];
        const uint32_t al = (uint32_t)((uintptr_t)dst & 15);      // LDS image is shifted so that 16-byte units line up with global memory
        const bool staged = total + al <= (uint32_t)STAGE;        // uniform per block
        if (ASCII && staged) {
            // Byte alphabets: the lane knows how many symbols it owes (nb), so the walk is count-driven and takes up to
            // three codewords per table lookup to the very end; the bytes collect in a 64-bit register and leave as
            // aligned 4-byte LDS ORs into the zeroed image -- the words two lanes share need no special case.
            if (live && er != BAD_REL && nb != 0) {
                const uint32_t o0 = al + my_off;
                uint32_t *o = reinterpret_cast<uint32_t *>(s_out) + (o0 >> 2);
                const uint32_t spo = 32u * ((uint32_t)tid + 1u);           // every codeword the lane owes starts inside its own group: see window32_sp
                uint32_t cnt = o0 & 3, remaining = nb, pos = my0 + er + spo;
                unsigned long long acc = 0;
                while (remaining) {
                    const uint32_t win = window32_sp(s_data, pos);
                    const uint32_t e = lut_at(a, s_lut, win, lane_r);
                    uint32_t bytes = u_bytes(e);
                    uint32_t take = min(u_n(e), remaining);               // (the last lookup may list more codewords than the lane owes:
                    uint32_t used = u_used(e);                            //  whatever lies above `cnt` bytes is masked off at the end, pos is dead by then)
                    if (!SHORT && __ballot((e & 0x80000000u) != 0)) {     // a code longer than K bits somewhere in the wavefront (rare: a scalar branch around it, no exec masks on the common path)
                        if (e & 0x80000000u) { uint32_t q = pos - spo; bytes = decode_long(a, s_data, win, e, q, l2); take = 1; used = q + spo - pos; }
                    }
                    acc |= (unsigned long long)bytes << (8 * cnt);
                    cnt += take; remaining -= take; pos += used;
                    if (cnt >= 4) { RSN_EMIT_WORD(o, (uint32_t)acc); o++; acc >>= 32; cnt -= 4; }
                    // the second lookup out of the same window (r04): `used` <= 11 bits are gone, >= 21 are left
                    if (used <= (uint32_t)LUT_BITS_MAX && remaining) {
                        uint32_t e2 = lut_at(a, s_lut, win >> used, lane_r);
                        if (!SHORT) e2 = (int32_t)e2 < 0 ? 0u : e2;       // (a long code waits for the next turn's fresh window: an entry of no symbols and no bits)
                        {
                            const uint32_t b2 = u_bytes(e2);
                            const uint32_t t2 = min(u_n(e2), remaining);
                            acc |= (unsigned long long)b2 << (8 * cnt);   // (remaining != 0: the first lookup was taken whole, acc holds nothing above cnt)
                            cnt += t2; remaining -= t2; pos += u_used(e2);
                            if (cnt >= 4) { RSN_EMIT_WORD(o, (uint32_t)acc); o++; acc >>= 32; cnt -= 4; }
                        }
                    }
                }
                if (cnt) atomicOr(o, (uint32_t)acc & (0xFFFFFFFFu >> (32 - 8 * cnt)));
            }
        } else if (live && er != BAD_REL && nb != 0) {
            if (staged) {
                uint8_t *o = s_out + al + my_off;
                uint32_t pos = my0 + er;
                emit_walk<ASCII, SHORT>(a, s_data, s_lut, lane_r, l2, pos, lim, [&](uint32_t b) { *o++ = (uint8_t)b; });
            } else {
                Sink sink;
                sink.start(dst + my_off);
                emit_walk<ASCII, SHORT>(a, s_data, s_lut, lane_r, l2, my0 + er, lim, [&](uint32_t b) { sink.put(b); });
                sink.finish();
            }
        }
        if (!staged) continue;
        __syncthreads();
        const uint32_t span = al + total;                          // bytes [al, span) of s_out are this block's output
        uint8_t *gbase = dst - al;                                 // 16-byte aligned
        for (uint32_t u = tid; u * 16 < span; u += DB) {
            const uint32_t b0 = u * 16;
            if (b0 >= al && b0 + 16 <= span) *reinterpret_cast<uint4 *>(gbase + b0) = *reinterpret_cast<const uint4 *>(s_out + b0);
            else for (uint32_t k = max(b0, al); k < min(b0 + 16, span); k++) gbase[k] = s_out[k];
            if (ASCII) *reinterpret_cast<uint4 *>(s_out + b0) = make_uint4(0, 0, 0, 0);   // the ORs of the next block need a zeroed image
        }
    }
}

// ---------------------------------------------------------------- F: flat codes
// Every code is L bits: symbol i sits at bits [p0 + i*L, +L).  A lane takes 32
// symbols = exactly L words (after a block-uniform funnel shift by p0 % 32), so
// every field position inside those words is a compile-time constant.
struct FlatArgs {
    const uint8_t *base; size_t nbytes;
    unsigned long long p0;       // bit position of the first field
    unsigned long long n_sym;
    uint8_t *out;
    uint8_t lut[128];            // 2^L bytes: field value -> symbol.  In the kernel-argument segment: no upload, no extra stream op
};
constexpr int FLAT_LANE_SYMS = 32;
constexpr int FLAT_SYMS = FDB * FLAT_LANE_SYMS;                  // symbols per block

template <int L>
__global__ __launch_bounds__(FDB) void k_dec_flat(FlatArgs a) {
    constexpr int WORDS = FDB * L + 2;
    constexpr int RL = 5;                                       // 32 dword copies per entry: the copy index IS the bank
    __shared__ uint32_t s_data[WORDS + WORDS / 32 + 2];
    __shared__ uint32_t s_lut[(1 << L) << RL];
    const int tid = threadIdx.x;
    {   // the table is read through the kernel-argument segment pointer: indexing the by-value copy would spill it to scratch
        const uint8_t *lut = (const uint8_t *)__builtin_amdgcn_kernarg_segment_ptr() + offsetof(FlatArgs, lut);
        for (int i = tid; i < ((1 << L) << RL); i += FDB) s_lut[i] = lut[i >> RL];
    }
    const uint32_t n_chunks = (uint32_t)((a.n_sym + FLAT_SYMS - 1) / FLAT_SYMS);
    for (uint32_t chunk = blockIdx.x; chunk < n_chunks; chunk += gridDim.x) {   // persistent: the table is staged once
    const unsigned long long sym0 = (unsigned long long)chunk * FLAT_SYMS;
    const unsigned long long bit0 = a.p0 + sym0 * L;            // FLAT_SYMS*L is a multiple of 32: bit0 % 32 == p0 % 32
    const size_t w0 = (size_t)(bit0 >> 5);
    const uint32_t o0 = (uint32_t)(bit0 & 31);
    if ((w0 + WORDS) * 4 <= a.nbytes) {                         // interior block: branch-free, loads issued together
        const uint32_t *src = reinterpret_cast<const uint32_t *>(a.base) + w0;
        constexpr int PER = (WORDS + FDB - 1) / FDB;
        uint32_t v[PER];
#pragma unroll
        for (int k = 0; k < PER; k++) { const int i = tid + k * FDB; if (i < WORDS) v[k] = ld4<(RSN_NT_MASK & 8) != 0>(src + i); }
#pragma unroll
        for (int k = 0; k < PER; k++) { const int i = tid + k * FDB; if (i < WORDS) s_data[swz(i)] = __builtin_bswap32(v[k]); }
    } else {
        for (int i = tid; i < WORDS; i += FDB) {
            const size_t off = (w0 + i) * 4;
            uint32_t v = 0;
            if (off + 4 <= a.nbytes) v = __builtin_bswap32(*reinterpret_cast<const uint32_t *>(a.base + off));
            else for (int k = 0; k < 4; k++) if (off + k < a.nbytes) v |= (uint32_t)a.base[off + k] << (24 - 8 * k);
            s_data[swz(i)] = v;
        }
    }
    __syncthreads();
    const unsigned long long s_first = sym0 + (unsigned long long)tid * FLAT_LANE_SYMS;
    if (s_first < a.n_sym) {
    const uint32_t lane_r = tid & ((1u << RL) - 1);
    uint32_t w[L + 1], v[L];
#pragma unroll
    for (int j = 0; j <= L; j++) w[j] = s_data[swz(tid * L + j)];
#pragma unroll
    for (int j = 0; j < L; j++) v[j] = (uint32_t)((((unsigned long long)w[j] << 32) | w[j + 1]) >> (32 - o0));
    uint32_t o[8];
#pragma unroll
    for (int k = 0; k < FLAT_LANE_SYMS; k++) {
        constexpr uint32_t mask = (1u << L) - 1;
        const int bp = k * L, wi = bp >> 5, sh = bp & 31;
        uint32_t f;
        if (sh + L <= 32) f = (v[wi] >> (32 - sh - L)) & mask;
        else f = (uint32_t)((((unsigned long long)v[wi] << 32) | v[wi + 1 < L ? wi + 1 : wi]) >> (64 - sh - L)) & mask;
        const uint32_t sym = s_lut[(f << RL) | lane_r];
        if ((k & 3) == 0) o[k >> 2] = sym; else o[k >> 2] |= sym << (8 * (k & 3));
    }
    uint8_t *dst = a.out + s_first;
    if (s_first + FLAT_LANE_SYMS <= a.n_sym) {
        st16<(RSN_NT_MASK & 16) != 0>(reinterpret_cast<uint4 *>(dst), make_uint4(o[0], o[1], o[2], o[3]));
        st16<(RSN_NT_MASK & 16) != 0>(reinterpret_cast<uint4 *>(dst + 16), make_uint4(o[4], o[5], o[6], o[7]));
    } else {
        const uint32_t cnt = (uint32_t)(a.n_sym - s_first);
        for (uint32_t k = 0; k < cnt; k++) dst[k] = (uint8_t)(o[k >> 2] >> (8 * (k & 3)));
    }
    }
    __syncthreads();                                               // s_data is restaged by the next chunk
    }
}

namespace {

// Fills the 2^K lookup table and the child array (host).
void build_tables(const HuffTree &t, int K, std::vector<uint32_t> &lut, std::vector<int32_t> &child) {
    const uint32_t A = t.n_leaves;
    const size_t n_int = t.freq.size() - A;
    child.assign(2 * std::max<size_t>(n_int, 1), 0);
    for (size_t i = 0; i < n_int; i++) {
        const int32_t id = (int32_t)(A + i);
        const int32_t kids[2] = {t.left[id], t.right[id]};
        for (int b = 0; b < 2; b++)
            child[2 * i + b] = t.is_leaf(kids[b]) ? -(int32_t)(t.rune[kids[b]] + 1) : kids[b] - (int32_t)A;
    }
    lut.assign((size_t)1 << K, 0);
    struct It { int32_t node; uint32_t prefix; int depth; };
    std::vector<It> st;
    st.push_back({t.root, 0, 0});
    while (!st.empty()) {
        const It it = st.back();
        st.pop_back();
        if (t.is_leaf(it.node)) {
            const uint32_t ent = ((uint32_t)it.depth << 24) | t.rune[it.node];
            const uint32_t lo = it.prefix << (K - it.depth);
            for (uint32_t x = 0; x < (1u << (K - it.depth)); x++) lut[lo + x] = ent;
        } else if (it.depth == K) {
            lut[it.prefix] = 0x80000000u | (uint32_t)(it.node - (int32_t)A);
        } else {
            st.push_back({t.right[it.node], (it.prefix << 1) | 1, it.depth + 1});
            st.push_back({t.left[it.node], it.prefix << 1, it.depth + 1});
        }
    }
}

// Second level: every K-bit prefix that ends INSIDE the tree gets a sub-table indexed by the next sb bits, sb = the
// depth of the sub-tree below it, capped so that (a) K + sb <= 32 (the index comes out of the same 32-bit window) and
// (b) all sub-tables together fit the budget: LDS if the whole sub-trees fit there (byte alphabets: always LDS, as
// deep as it allows), else LUT2_GLOBAL_MAX entries read through L2 (rune alphabets with 10^5 symbols: half the symbols
// are longer than K bits and child[] is far too large for LDS).  Codes deeper than K + sb continue bit by bit.
void build_second_level(const std::vector<int32_t> &child, int K, bool byte_alphabet, std::vector<uint32_t> &lut, std::vector<uint32_t> &lut2) {
    lut2.clear();
    const size_t n_int = child.size() / 2;
    std::vector<uint8_t> depth(n_int, 0);                          // bits from an internal node to its deepest leaf (children precede parents)
    for (size_t i = 0; i < n_int; i++) {
        unsigned d = 0;
        for (int b = 0; b < 2; b++) { const int32_t k = child[2 * i + b]; d = std::max(d, k < 0 ? 1u : 1u + depth[(size_t)k]); }
        depth[i] = (uint8_t)std::min(d, 255u);
    }
    std::vector<uint32_t> longs;
    for (uint32_t v = 0; v < (1u << K); v++) if (lut[v] & 0x80000000u) longs.push_back(v);
    if (longs.empty()) return;
    auto total = [&](unsigned cap) { uint64_t t = 0; for (uint32_t v : longs) t += 1ull << std::min<unsigned>(depth[lut[v] & 0x3FFFFFFu], cap); return t; };
    unsigned cap = 32u - (unsigned)K;
    if (total(cap) > (uint64_t)LUT2_LDS) {
        const uint64_t budget = byte_alphabet ? (uint64_t)LUT2_LDS : (uint64_t)LUT2_GLOBAL_MAX;
        while (cap > 0 && total(cap) > budget) cap--;
    }
    if (cap == 0) return;                                          // (the entries stay "node, walked bit by bit")
    // Sub-tables are written in the device's index order straight away: the device reads the stream LSB-first, so the
    // path bits of a node form the LOW bits of its index (first bit lowest) and a leaf at depth q fills every index whose low
    // q bits are its path -- a strided fill instead of a contiguous one, and no permutation pass over 10^6 entries afterwards.
    struct It { int32_t node; uint32_t path; unsigned q; };               // path: bit i = the i-th bit taken below the K-bit prefix
    std::vector<It> st;
    for (uint32_t v : longs) {
        const int32_t top = (int32_t)(lut[v] & 0x3FFFFFFu);
        const unsigned sb = std::min<unsigned>(depth[(size_t)top], cap);
        const uint32_t off = (uint32_t)lut2.size();
        lut2.resize(lut2.size() + ((size_t)1 << sb));
        st.push_back({top, 0, 0});
        while (!st.empty()) {
            const It it = st.back();
            st.pop_back();
            if (it.q == sb) { lut2[off + it.path] = 0x80000000u | (uint32_t)it.node; continue; }   // still inside the tree after K + sb bits
            for (int b = 0; b < 2; b++) {
                const int32_t k = child[2 * (size_t)it.node + b];
                const uint32_t path = it.path | ((uint32_t)b << it.q);
                const unsigned q1 = it.q + 1;
                if (k >= 0) { st.push_back({k, path, q1}); continue; }
                const uint32_t ent = ((uint32_t)(K + q1) << 24) | (uint32_t)(-(k + 1));
                for (uint32_t y = 0; y < (1u << (sb - q1)); y++) lut2[off + path + (y << q1)] = ent;
            }
        }
        lut[v] = 0x80000000u | (sb << 26) | off;
    }
}

int rep_for(int K) {   // replicate small tables up to 32x (one copy per LDS bank) within LUT_WORDS
    int r = 0;
    while (r < 5 && ((1 << K) << (r + 1)) <= LUT_WORDS) r++;
    return r;
}

}  // namespace

int huff_decode_dev(Ctx &c, hipStream_t s, const uint8_t *d_in, size_t n, uint8_t *d_out, size_t out_cap, size_t *out_n, const SliceStream *st) {
    if (((uintptr_t)d_in & 15) || ((uintptr_t)d_out & 15)) return c.fail(RSN_ERR_ARG, "huffman: device buffers must be 16-byte aligned");
    *out_n = 0;
    // ---- header to the host: strings.SplitN(content, "\\\n", 2) (huffman.go:261)
    // (fetched into pinned memory: 4 KiB covers every ASCII-alphabet header, larger ones grow 8x per try)
    std::vector<uint8_t> head;
    size_t sep = (size_t)-1;
    for (size_t want = 4096;; want *= 8) {
        const size_t k = std::min(want, n);
        const size_t old = head.size();
        if (k > old) {
            if (st && !st->need_in(k)) return c.fail(RSN_ERR_DEVICE, "huffman: the upload of a sliced call failed");
            void *hpin; int prc = pinned_buf(c, k - old + 64, &hpin); if (prc) return prc;
            RSN_HIP(hipMemcpyAsync(hpin, d_in + old, k - old, hipMemcpyDeviceToHost, s));
            RSN_HIP(hipStreamSynchronize(s));
            head.resize(k);
            memcpy(head.data() + old, hpin, k - old);
        }
        for (size_t i = old ? old - 1 : 0; i + 1 < k; i++) if (head[i] == 0x5C && head[i + 1] == 0x0A) { sep = i; break; }
        if (sep != (size_t)-1 || k == n) break;
    }
    if (sep == (size_t)-1) return c.fail(RSN_ERR_FORMAT, "huffman: no '\\\\\\n' separator (reference: index out of range, huffman.go:264)");
    if (sep + 3 > head.size() && sep + 3 <= n) {   // make sure the pad byte is on the host
        const size_t old = head.size();
        if (st && !st->need_in(sep + 3)) return c.fail(RSN_ERR_DEVICE, "huffman: the upload of a sliced call failed");
        void *hpin; int prc = pinned_buf(c, 64, &hpin); if (prc) return prc;
        RSN_HIP(hipMemcpyAsync(hpin, d_in + old, sep + 3 - old, hipMemcpyDeviceToHost, s));
        RSN_HIP(hipStreamSynchronize(s));
        head.resize(sep + 3);
        memcpy(head.data() + old, hpin, sep + 3 - old);
    }
    std::vector<HuffSym> syms; std::string msg;
    static const bool host_timing = getenv("RSN_HOST_TIMING") != nullptr;   // prints where the host side of a call spends its time
    auto now = [] { return std::chrono::steady_clock::now(); };
    auto ms = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) { return std::chrono::duration<double, std::milli>(b - a).count(); };
    const auto t0 = now();
    if (!parse_header(head.data(), sep, syms, msg)) return c.fail(RSN_ERR_FORMAT, "%s", msg.c_str());
    const auto t1 = now();
    const size_t sn = n - sep - 2;                       // bytes after the separator
    const unsigned diff = sn ? head[sep + 2] : 0;         // byteArr[0] (huffman.go:275)
    const unsigned long long nbits = sn ? (unsigned long long)(sn - 1) * 8 : 0;
    if (diff > nbits) return c.fail(RSN_ERR_FORMAT, "huffman: pad exceeds the payload (reference: slice bounds out of range, huffman.go:294)");
    const unsigned long long max = nbits - diff;
    if (!d_out) {   // the size query, answered from the header alone (before any tree is built): every symbol as often as the header says,
                    // which is what a stream this library wrote decodes to.  The cheap format checks come first, so that a malformed
                    // stream is refused here and not only on the call that follows, and the sum cannot wrap: with two symbols and more
                    // every one takes at least a bit of the payload, so counts that add up to more than the payload has bits (a foreign
                    // header; parse_header saturates a count at 2^63-1) are answered by the payload's own bound, not by exabytes.
        if (syms.empty()) return c.fail(RSN_ERR_FORMAT, "huffman: empty header (reference panics in heap.Pop, huffman.go:102)");
        if (syms.size() == 1) {
            if (max > 0) return c.fail(RSN_ERR_FORMAT, "huffman: single-symbol tree with a non-empty payload (reference recurses without end, huffman.go:139-140)");
            *out_n = 32;
            return c.fail(RSN_ERR_CAPACITY, "huffman: output needs %d bytes by the header's counts", utf8_len(syms[0].rune));
        }
        if (max == 0) return c.fail(RSN_ERR_FORMAT, "huffman: payload ends inside a codeword (reference: index out of range, huffman.go:145)");
        unsigned long long total = 0, expect = 0;
        bool beyond = false;                                    // the header counts more symbols than the payload has bits: a foreign stream
        for (const HuffSym &sy : syms) {                        // (the counts only shape the tree, huffman.go:196-227 -- still decodable)
            if (sy.freq > max - total) { beyond = true; break; }
            total += sy.freq;
            expect += sy.freq * (unsigned long long)utf8_len(sy.rune);     // <= 4 * max: cannot wrap
        }
        if (beyond) expect = 4 * max;                           // what the payload can decode to at most: a bit per symbol, four bytes per rune
        *out_n = round_up((size_t)expect, 16) + 16;
        return c.fail(RSN_ERR_CAPACITY, beyond ? "huffman: output needs at most %llu bytes (the header counts more symbols than the payload holds)"
                                               : "huffman: output needs %llu bytes by the header's counts", expect);
    }
    HuffTree tree; HuffCodes codes;
    if (!build_tree(syms, tree, msg)) return c.fail(RSN_ERR_FORMAT, "%s", msg.c_str());
    const auto t2 = now();
    // A large alphabet (config 2b: 3*10^5 runes) spends milliseconds both on the code lengths and on the lookup tables; neither needs
    // the other -- more than 2^11 leaves means codes longer than any first-level table, so K is known -- so a second host thread
    // builds the tables meanwhile.
    constexpr int k_env = LUT_BITS_MAX;                                 // index bits of the first-level table, at most
    std::vector<uint32_t> lut, lut2; std::vector<int32_t> child;
    const bool prebuilt = tree.n_leaves > 4096;
    std::optional<SideJob> table_builder;                               // (a pooled helper, rsn_helpers.h; on this thread when none is to be had)
    if (prebuilt) table_builder.emplace([&] { build_tables(tree, k_env, lut, child); build_second_level(child, k_env, false, lut, lut2); });
    const bool codes_ok = assign_codes(tree, codes, msg, false);
    if (prebuilt) table_builder->finish();
    if (!codes_ok) return c.fail(RSN_ERR_LIMIT, "%s", msg.c_str());
    const auto t3 = now();
    if (host_timing) fprintf(stderr, "huffman decode host: header of %zu bytes parsed in %.2f ms, tree %.2f ms, codes %.2f ms, %zu symbols\n", sep, ms(t0, t1), ms(t1, t2), ms(t2, t3), syms.size());

    if (tree.n_leaves == 1) {                             // bare-leaf tree (huffman.go:136-143)
        if (max > 0) return c.fail(RSN_ERR_FORMAT, "huffman: single-symbol tree with a non-empty payload (reference recurses without end, huffman.go:139-140)");
        uint8_t u[4];
        const size_t k = (size_t)go_encode_rune(tree.rune[0], u);
        *out_n = k;
        if (!d_out || out_cap < k) { return c.fail(RSN_ERR_CAPACITY, "huffman: output needs %zu bytes", k); }
        RSN_HIP(hipMemcpyAsync(d_out, u, k, hipMemcpyHostToDevice, s));
        RSN_HIP(hipStreamSynchronize(s));
        return RSN_OK;
    }
    if (max == 0) return c.fail(RSN_ERR_FORMAT, "huffman: payload ends inside a codeword (reference: index out of range, huffman.go:145)");

    bool ascii = true;
    for (uint32_t i = 0; i < tree.n_leaves; i++) if (tree.rune[i] >= 0x80) ascii = false;
    const size_t pay = sep + 3;                           // first payload byte
    const size_t A0 = pay & ~(size_t)15;
    void *p; int rc;

    // ---- F: every code has the same length -> fixed-width unpack
    static const bool no_flat = getenv("RSN_NO_FLAT") != nullptr;
    if (ascii && codes.min_len == codes.max_len && codes.max_len <= 7 && !no_flat) {
        const uint32_t L = codes.max_len;
        if (max % L) return c.fail(RSN_ERR_FORMAT, "huffman: payload ends inside a codeword (reference: index out of range, huffman.go:145)");
        const unsigned long long n_sym = max / L;
        *out_n = (size_t)n_sym;
        if (!d_out || n_sym > out_cap) { *out_n = round_up((size_t)n_sym, 16) + 16; return c.fail(RSN_ERR_CAPACITY, "huffman: output needs %llu bytes, buffer holds %zu", n_sym, out_cap); }
        FlatArgs fa{};
        fa.base = d_in + A0; fa.nbytes = n - A0; fa.p0 = 8ull * (pay - A0) + diff; fa.n_sym = n_sym;
        fa.out = d_out;
        for (uint32_t i = 0; i < tree.n_leaves; i++) fa.lut[codes.code[i]] = (uint8_t)tree.rune[i];
        // (a sliced call: symbols [q0, q1) a launch -- a multiple of FLAT_SYMS, so that a slice's first field keeps the stream's bit phase --
        //  once the bytes that hold them, and the words a block loads beyond its own, are up)
        const unsigned long long per = st ? std::max<unsigned long long>(FLAT_SYMS, (unsigned long long)st->slice_bytes * 8 / L / FLAT_SYMS * FLAT_SYMS) : n_sym;
        const unsigned long long p0_all = fa.p0;
        for (unsigned long long q0 = 0; q0 < n_sym || q0 == 0; q0 += per) {
            const unsigned long long q1 = std::min(n_sym, q0 + per);
            if (st) {
                const size_t upto = std::min(n, A0 + (size_t)((p0_all + q1 * L + 7) / 8) + 4096);
                if (!st->need_in(upto)) return c.fail(RSN_ERR_DEVICE, "huffman: the upload of a sliced call failed");
                fa.nbytes = upto - A0;
            }
            fa.p0 = p0_all + q0 * L; fa.n_sym = q1 - q0; fa.out = d_out + q0;
            const dim3 grid((uint32_t)std::min<size_t>(ceil_div((size_t)(q1 - q0), FLAT_SYMS), 256 * 8));
            switch (L) {
#define RSN_FLAT_CASE(LL) case LL: RSN_LAUNCH("huff_dec_flat", k_dec_flat<LL>, grid, dim3(FDB), 0, s, fa); break;
                RSN_FLAT_CASE(1) RSN_FLAT_CASE(2) RSN_FLAT_CASE(3) RSN_FLAT_CASE(4) RSN_FLAT_CASE(5) RSN_FLAT_CASE(6)
                RSN_FLAT_CASE(7)   // ASCII alphabets hold at most 2^7 symbols
#undef RSN_FLAT_CASE
                default: return c.fail(RSN_ERR_LIMIT, "huffman: flat code length %u", L);
            }
            RSN_HIP(hipStreamSynchronize(s));
            if (st && !st->have_out((size_t)q0, (size_t)(q1 - q0))) return c.fail(RSN_ERR_DEVICE, "huffman: the download of a sliced call failed");
            if (q1 >= n_sym) break;
        }
        return RSN_OK;
    }

    // ---- tables
    // First-level index bits.  Rune alphabets: the longest code, capped.  Byte alphabets (r04): room for THREE codewords -- an entry
    // lists up to three, and with K = the longest code a text alphabet (longest code 7, mean 4.4 bits) got 1.6 symbols out of a
    // lookup where 11 bits hold 2.4; smaller tables are replicated across the LDS banks as before.
    const int K = (int)std::min<unsigned>(ascii ? std::max(codes.max_len, 3u * codes.min_len + 2u) : codes.max_len, (unsigned)k_env);
    const bool short_codes = codes.max_len <= (unsigned)K;
    if (!prebuilt) build_tables(tree, K, lut, child);
    const auto t4 = now();
    // byte alphabets: the unified table (up to three whole codewords per K-bit window)
    static const bool no_multi = getenv("RSN_NO_MULTI") != nullptr;
    unsigned long long n_syms_total = 0;
    for (uint32_t i = 0; i < tree.n_leaves; i++) n_syms_total += tree.freq[i];
    // taking several codewords per lookup is worth it when a K-bit window usually holds two or more
    const bool multi = ascii && !no_multi && n_syms_total && codes.total_bits * 10 <= n_syms_total * 65;   // mean code <= 6.5 bits
    if (ascii) {
        const std::vector<uint32_t> plain = lut;                              // len << 24 | rune, or 0x80000000 | node
        for (uint32_t v = 0; v < (1u << K); v++) {
            if (plain[v] & 0x80000000u) continue;                                 // first code longer than K bits: the node stays
            uint32_t used = 0, nsym = 0, syms = 0, len1 = 0;
            while (nsym < 3 && used < (uint32_t)K) {
                int32_t node = tree.root;
                uint32_t q = used;
                while (!tree.is_leaf(node) && q < (uint32_t)K) { node = ((v >> (K - 1 - q)) & 1) ? tree.right[node] : tree.left[node]; q++; }
                if (!tree.is_leaf(node)) break;                                   // ran out of bits inside a codeword
                syms |= (tree.rune[node] & 0x7Fu) << (8 * nsym);
                if (nsym == 0) len1 = q;
                used = q; nsym++;
            }
            lut[v] = u_entry(syms, used, nsym, len1);                             // nsym >= 1 here: the first codeword fits
        }
    }
    if (!prebuilt && !short_codes) build_second_level(child, K, ascii, lut, lut2);
    const auto t5 = now();
    {   // the device reads the stream LSB-first: first-level entry v moves to the index with v's K bits reversed
        //     (the second level is built in that order, see build_second_level)
        std::vector<uint32_t> t(lut.size());
        for (uint32_t v = 0; v < (1u << K); v++) t[K ? __builtin_bitreverse32(v) >> (32 - K) : 0u] = lut[v];
        lut.swap(t);
    }
    if (host_timing) fprintf(stderr, "huffman decode host: first level + child array %.2f ms, second level %.2f ms (%zu entries), bit-reversal %.2f ms\n", ms(t3, t4), ms(t4, t5), lut2.size(), ms(t5, now()));
    static const bool dbg = getenv("RSN_DEBUG") != nullptr;
    if (dbg) fprintf(stderr, "huffman decode tables: K %d, longest code %u, %zu tree nodes, second level %zu entries (%s)\n", K, codes.max_len, child.size() / 2, lut2.size(), lut2.size() <= (size_t)LUT2_LDS ? "LDS" : "L2");
    rc = dev_buf(c, 5, (lut.size() + child.size() + lut2.size()) * 4 + 64, &p); if (rc) return rc;
    uint32_t *d_lut = (uint32_t *)p;
    int32_t *d_child = (int32_t *)(d_lut + lut.size());
    uint32_t *d_lut2 = (uint32_t *)(d_child + child.size());
    std::vector<uint32_t> packed;                                       // (small tables go up in one copy: a call on a few KB is all fixed costs)
    if (lut.size() + child.size() + lut2.size() <= (64u << 10)) {
        packed.reserve(lut.size() + child.size() + lut2.size());
        packed.insert(packed.end(), lut.begin(), lut.end());
        packed.insert(packed.end(), reinterpret_cast<const uint32_t *>(child.data()), reinterpret_cast<const uint32_t *>(child.data()) + child.size());
        packed.insert(packed.end(), lut2.begin(), lut2.end());
        RSN_HIP(hipMemcpyAsync(d_lut, packed.data(), packed.size() * 4, hipMemcpyHostToDevice, s));
    } else {
        RSN_HIP(hipMemcpyAsync(d_lut, lut.data(), lut.size() * 4, hipMemcpyHostToDevice, s));
        RSN_HIP(hipMemcpyAsync(d_child, child.data(), child.size() * 4, hipMemcpyHostToDevice, s));
        if (!lut2.empty()) RSN_HIP(hipMemcpyAsync(d_lut2, lut2.data(), lut2.size() * 4, hipMemcpyHostToDevice, s));
    }

    // ---- the payload, whole or slice by slice.  A range is the subsequences of [base, base + 32 n_sub): its first codeword starts at bit p0
    //      of it (the pad for the stream's first range, the predecessor's last exit for a later one), no codeword is taken that starts at or
    //      after bit `end`; *last_exit: where the codeword after the range's last one starts, in bits past the range (BAD_REL: the range ran
    //      off the payload).
    void *hp; rc = pinned_buf(c, 64, &hp); if (rc) return rc;
    static const int warm_env = [] { const char *e = getenv("RSN_DEC_WARM"); return e ? atoi(e) : 128; }();
    auto decode_range = [&](const uint8_t *rbase, size_t rnbytes, unsigned long long rp0, unsigned long long rend, unsigned long long rsubs, uint8_t *rout, size_t rcap,
                            size_t *rtotal, uint32_t *rlast) -> int {
    DecArgs a{};
    a.base = rbase; a.nbytes = rnbytes;
    a.p0 = rp0;
    a.end = rend;
    const unsigned long long n_sub64 = rsubs;                           // (a slice's `end` lies behind its last subsequence: what its last codeword may reach into)
    if (n_sub64 > 0xFFFFFF00ull) return c.fail(RSN_ERR_LIMIT, "huffman: payload too large for one call");
    a.n_sub = (uint32_t)n_sub64;
    a.lut = d_lut; a.K = K; a.rep_log2 = rep_for(K); a.child = d_child; a.child_n = (uint32_t)child.size(); a.min_len = codes.min_len;
    a.lut2 = d_lut2; a.lut2_n = (uint32_t)lut2.size();
    a.flat_guess = codes.min_len == codes.max_len;
    a.warm = std::min(std::max(warm_env, 0), ORG - 32);
    const uint32_t n_blk = (uint32_t)ceil_div(a.n_sub, DB);
    rc = dev_buf(c, 6, (size_t)a.n_sub * 6 + 64, &p); if (rc) return rc;
    a.exit_rel = (uint16_t *)p; a.entry_rel = a.exit_rel + a.n_sub; a.nbyte = a.entry_rel + a.n_sub;
    rc = dev_buf(c, 7, ((size_t)n_blk * 2 + 4) * 8 + (size_t)n_blk * 4 + 64, &p); if (rc) return rc;
    a.blk_bytes = (unsigned long long *)p;
    unsigned long long *d_blk_off = a.blk_bytes + n_blk;
    unsigned long long *d_total = d_blk_off + n_blk;
    int *d_changed = (int *)(d_total + 1);
    int *d_stuck = (int *)(d_total + 2);
    a.stuck = d_stuck;
    uint32_t *d_fix_count = (uint32_t *)(d_changed + 1);                 // (zeroed together with `changed`)
    uint32_t *d_fix_list = (uint32_t *)(d_total + 4);
    a.changed = d_changed;

    constexpr uint32_t grid_env = 256u * 8u * 2u;                        // persistent blocks: two waves of them per CU slot
    const uint32_t grid_p = std::min<uint32_t>(n_blk, grid_env);          // persistent blocks: the LUT is staged once per block
    auto launch_sync = [&]() -> int {
        const char *nm = a.pass == 0 ? "huff_dec_sync" : a.pass == 1 ? "huff_dec_sync_fix" : "huff_dec_sync_verify";
        if (multi && short_codes) RSN_LAUNCH(nm, (k_dec_sync<true, true, true>), dim3(grid_p), dim3(DB), 0, s, a, n_blk);
        else if (multi) RSN_LAUNCH(nm, (k_dec_sync<true, false, true>), dim3(grid_p), dim3(DB), 0, s, a, n_blk);
        else if (ascii && short_codes) RSN_LAUNCH(nm, (k_dec_sync<true, true, false>), dim3(grid_p), dim3(DB), 0, s, a, n_blk);
        else if (ascii) RSN_LAUNCH(nm, (k_dec_sync<true, false, false>), dim3(grid_p), dim3(DB), 0, s, a, n_blk);
        else if (short_codes) RSN_LAUNCH(nm, (k_dec_sync<false, true, false>), dim3(grid_p), dim3(DB), 0, s, a, n_blk);
        else RSN_LAUNCH(nm, (k_dec_sync<false, false, false>), dim3(grid_p), dim3(DB), 0, s, a, n_blk);
        return RSN_OK;
    };
    // ---- D1 (fixed point), D2 (offsets), D3 (bytes) are queued together and read back ONCE (r04: the call used to stop three times --
    // for `changed`, for the total, at the end -- ~0.15 ms of a 1.8 ms decode): D3 skips any block that would write past out_cap, so it
    // can run before the host has seen the total; if the fixing pass did hand some block a different exit (rare), everything
    // after it is queued again.
    a.blk_off = d_blk_off; a.out = rout; a.out_cap = rcap;
    struct Tail { unsigned long long total; uint16_t last_exit; int changed; int stuck; };
    Tail *ht = (Tail *)hp;
    auto launch_emit = [&]() -> int {
        // (8-byte table entries with the symbols already spread to bytes measured slower than unpacking the 4-byte ones: 1.19 vs 0.98 ms)
        // (r04: a lookup's bytes as ONE four-byte LDS store at the lane's byte offset instead of the 64-bit collector and the ORs -- gfx950
        //  takes dword stores at any address, scripts/probes/lds_unaligned.cpp, and the walk drops from 46 to 25 vector instructions a turn --
        //  measured 1.40 against 0.80 ms: the LDS serialises a wavefront's unaligned stores.)
        // (This kernel scales almost linearly with blocks per CU up to the four its LDS allows.  Without the LDS output stage --
        //  a lane's dwords straight to memory -- seven blocks fit, but the scattered partial-line stores cost more than that
        //  buys: 1.66 vs 0.96 ms.)
        if (ascii && short_codes) RSN_LAUNCH("huff_dec_emit", (k_dec_emit<true, true, false>), dim3(grid_p), dim3(DB), 0, s, a, n_blk);
        else if (ascii) RSN_LAUNCH("huff_dec_emit", (k_dec_emit<true, false, false>), dim3(grid_p), dim3(DB), 0, s, a, n_blk);
        else if (short_codes) RSN_LAUNCH("huff_dec_emit", (k_dec_emit<false, true, false>), dim3(grid_p), dim3(DB), 0, s, a, n_blk);
        else RSN_LAUNCH("huff_dec_emit", (k_dec_emit<false, false, false>), dim3(grid_p), dim3(DB), 0, s, a, n_blk);
        return RSN_OK;
    };
    auto offsets_and_bytes = [&]() -> int {                             // D2 + D3 + the read-back of total and final exit, queued
        int rc2 = scan_u64(c, s, "huff_dec_scan", a.blk_bytes, d_blk_off, n_blk, d_total); if (rc2) return rc2;
        rc2 = launch_emit(); if (rc2) return rc2;
        RSN_HIP(hipMemcpyAsync(&ht->total, d_total, 8, hipMemcpyDeviceToHost, s));
        RSN_HIP(hipMemcpyAsync(&ht->last_exit, a.exit_rel + (a.n_sub - 1), 2, hipMemcpyDeviceToHost, s));
        RSN_HIP(hipMemcpyAsync(&ht->stuck, d_stuck, 4, hipMemcpyDeviceToHost, s));
        return RSN_OK;
    };
    // ---- the phase solver (k_dec_phase): RSN_OK = the arrays hold the true path; 1 = it gave up (the passes go on)
    auto phase_solve = [&]() -> int {
        static const bool dbg2 = getenv("RSN_DEBUG") != nullptr;
        const auto ph_t0 = std::chrono::steady_clock::now();
        auto ph_ms = [&]() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - ph_t0).count(); };
        const size_t nbk = n_blk;
        void *pp2; int r2 = dev_buf(c, 34, nbk * (PH_CAND * 8 + 1) + 256, &pp2); if (r2) return r2;
        PhaseArgs ph{};
        ph.blk_map = (uint2 *)pp2;
        uint8_t *d_true = (uint8_t *)(ph.blk_map + PH_CAND * nbk);
        ph.true_c = d_true;
        ph.cand = std::min<uint32_t>(PH_CAND, std::max<uint32_t>(codes.max_len, 1));
        auto launch_phase = [&](int mode) -> int {
            const char *nm = mode == 0 ? "huff_dec_phase" : "huff_dec_phase_write";
            if (multi && short_codes) RSN_LAUNCH(nm, (k_dec_phase<true, true, true>), dim3(grid_p), dim3(DB), 0, s, a, ph, n_blk, mode);
            else if (multi) RSN_LAUNCH(nm, (k_dec_phase<true, false, true>), dim3(grid_p), dim3(DB), 0, s, a, ph, n_blk, mode);
            else if (ascii && short_codes) RSN_LAUNCH(nm, (k_dec_phase<true, true, false>), dim3(grid_p), dim3(DB), 0, s, a, ph, n_blk, mode);
            else if (ascii) RSN_LAUNCH(nm, (k_dec_phase<true, false, false>), dim3(grid_p), dim3(DB), 0, s, a, ph, n_blk, mode);
            else if (short_codes) RSN_LAUNCH(nm, (k_dec_phase<false, true, false>), dim3(grid_p), dim3(DB), 0, s, a, ph, n_blk, mode);
            else RSN_LAUNCH(nm, (k_dec_phase<false, false, false>), dim3(grid_p), dim3(DB), 0, s, a, ph, n_blk, mode);
            return RSN_OK;
        };
        void *hq; r2 = pinned_buf(c, 64 + nbk * PH_CAND * 8, &hq); if (r2) return r2;      // (the pinned block may have moved: `ht` lives at its start)
        hp = hq;
        ht = (Tail *)hq;
        ht->changed = 1;
        uint2 *hmap = (uint2 *)((uint8_t *)hq + 64);
        r2 = launch_phase(0); if (r2) return r2;
        RSN_HIP(hipMemcpyAsync(hmap, ph.blk_map, nbk * PH_CAND * 8, hipMemcpyDeviceToHost, s));
        RSN_HIP(hipStreamSynchronize(s));
        if (dbg2) fprintf(stderr, "huffman decode, phases: +%.3f ms: every entry of every lane walked, the blocks' maps on the host\n", ph_ms());
        // the blocks chained: block 0 is entered at the stream's first code bit (bit 0 of its table), block b + 1 at the bit b left at
        std::vector<uint8_t> tc(nbk, 0xFF);
        uint32_t e = 0;
        for (size_t b = 0; b < nbk; b++) {
            // (the maps have just come in by DMA: every block's two lines are misses, and which entry is read depends on the block before --
            //  6802 blocks took 0.8-1.0 ms of a 2.5 ms decode; fetched sixteen blocks ahead: 0.03)
            if (b + 16 < nbk) for (uint32_t k = 0; k < ph.cand; k += 8) __builtin_prefetch(&hmap[(b + 16) * PH_CAND + k]);
            if (e == BAD_REL) break;                                          // (the stream ran off its payload: the rest stays 0xFF)
            if (e >= ph.cand) { if (dbg2) fprintf(stderr, "huffman decode, phases: block %zu is entered at bit %u\n", b, e); return 1; }
            const uint2 m = hmap[b * PH_CAND + e];
            if (m.y == 0xFFFFFFFFu) { if (dbg2) fprintf(stderr, "huffman decode, phases: block %zu has no path from bit %u\n", b, e); return 1; }
            tc[b] = (uint8_t)e;
            e = m.x;
        }
        if (dbg2) fprintf(stderr, "huffman decode, phases: +%.3f ms: the blocks chained\n", ph_ms());
        RSN_HIP(hipMemcpyAsync(d_true, tc.data(), nbk, hipMemcpyHostToDevice, s));
        r2 = launch_phase(2); if (r2) return r2;
        RSN_HIP(hipStreamSynchronize(s));                                     // (tc is host memory: the copy has left it)
        if (dbg2) fprintf(stderr, "huffman decode, phases: +%.3f ms: the true path written\n", ph_ms());
        if (dbg2) fprintf(stderr, "huffman decode: settled by every entry of every lane (%u blocks, %u entries a lane)\n", n_blk, ph.cand);
        return RSN_OK;
    };
    a.pass = 0;
    RSN_HIP(hipMemsetAsync(d_stuck, 0, 4, s));
    rc = launch_sync(); if (rc) return rc;
    ht->changed = 0; ht->stuck = 0;
    if (n_blk > 1) {                                                  // (a single block starts from the exact entry and iterates to its fixed point in LDS)
        a.pass = 1;
        RSN_HIP(hipMemsetAsync(d_changed, 0, 8, s));
        RSN_LAUNCH("huff_dec_fix_list", k_dec_fix_list, dim3((uint32_t)ceil_div(n_blk, 256)), dim3(256), 0, s, (const uint16_t *)a.exit_rel, (const uint16_t *)a.entry_rel, n_blk, d_fix_list, d_fix_count);
        a.fix_list = d_fix_list; a.fix_count = d_fix_count;
        rc = launch_sync(); if (rc) return rc;
        RSN_HIP(hipMemcpyAsync(&ht->changed, d_changed, 4, hipMemcpyDeviceToHost, s));
    }
    rc = offsets_and_bytes(); if (rc) return rc;                      // ... on the assumption that the one fixing pass settled it (it nearly always does)
    RSN_HIP(hipStreamSynchronize(s));
    if (ht->stuck) {                                                    // some block's lanes did not settle among themselves: every entry of every lane
        const int prc = phase_solve();
        if (prc < 0) return prc;
        if (prc != RSN_OK) return c.fail(RSN_ERR_DEVICE, "huffman: synchronisation did not converge");
        rc = offsets_and_bytes(); if (rc) return rc;
        RSN_HIP(hipStreamSynchronize(s));
    } else if (ht->changed) {
        // the fixing pass handed some block a different exit: more passes, the synchronisation alone, until nothing changes, then D2 + D3
        // again.  Two more passes settle what one did not; a stream that is still moving then is in more than one PHASE (periodic data,
        // a code whose lengths are all even): k_dec_phase settles it in a few launches, where the passes take one per block of the stretch.
        static const bool dbg_p = getenv("RSN_DEBUG") != nullptr;
        for (uint32_t pass = 2; ht->changed; pass++) {
            if (dbg_p) fprintf(stderr, "huffman decode: pass %u, still moving\n", pass);
            if (pass == 3) {
                const int prc = phase_solve();
                if (prc < 0) return prc;
                if (prc == RSN_OK) break;
            }
            if (pass > n_blk + 2) return c.fail(RSN_ERR_DEVICE, "huffman: synchronisation did not converge");
            a.pass = (int)pass;
            RSN_HIP(hipMemsetAsync(d_changed, 0, 8, s));
            RSN_LAUNCH("huff_dec_fix_list", k_dec_fix_list, dim3((uint32_t)ceil_div(n_blk, 256)), dim3(256), 0, s, (const uint16_t *)a.exit_rel, (const uint16_t *)a.entry_rel, n_blk, d_fix_list, d_fix_count);
            rc = launch_sync(); if (rc) return rc;
            RSN_HIP(hipMemcpyAsync(&ht->changed, d_changed, 4, hipMemcpyDeviceToHost, s));
            RSN_HIP(hipMemcpyAsync(&ht->stuck, d_stuck, 4, hipMemcpyDeviceToHost, s));
            RSN_HIP(hipStreamSynchronize(s));
            if (ht->stuck) {
                const int prc = phase_solve();
                if (prc < 0) return prc;
                if (prc != RSN_OK) return c.fail(RSN_ERR_DEVICE, "huffman: synchronisation did not converge");
                break;
            }
        }
        rc = offsets_and_bytes(); if (rc) return rc;
        RSN_HIP(hipStreamSynchronize(s));
    }
    *rlast = ht->last_exit;
    *rtotal = (size_t)ht->total;
    return RSN_OK;
    };

    const uint8_t *base0 = d_in + A0;
    const unsigned long long end0 = 8ull * (n - A0), p00 = 8ull * (pay - A0) + diff;
    auto finish = [&](size_t total, uint32_t last_exit, size_t cap) -> int {
        if (last_exit != 0)
            return c.fail(RSN_ERR_FORMAT, "huffman: payload ends inside a codeword (reference: index out of range, huffman.go:145)");
        *out_n = total;
        if (total > cap) { *out_n = round_up(total, 16) + 16; return c.fail(RSN_ERR_CAPACITY, "huffman: output needs %zu bytes, buffer holds %zu", total, out_cap); }
        return RSN_OK;
    };
    if (!st) {
        size_t total = 0; uint32_t last = 0;
        rc = decode_range(base0, n - A0, p00, end0, (end0 + SBITS - 1) / SBITS, d_out, out_cap, &total, &last); if (rc) return rc;
        return finish(total, last, out_cap);
    }
    // slices of whole blocks of subsequences; a slice may look 4096 bits past its own end (a codeword that begins in it ends there, and the
    // kernels stage a few words beyond a block), the next one begins where its last codeword ended
    const unsigned long long slice_bits = std::max<unsigned long long>((unsigned long long)DB * SBITS, (unsigned long long)st->slice_bytes * 8 / (DB * SBITS) * (DB * SBITS));
    size_t produced = 0;
    unsigned long long entry = p00;                                      // (bits from the slice's base)
    for (unsigned long long b0 = 0; b0 < end0; b0 += slice_bits) {
        const bool fin = b0 + slice_bits >= end0;
        const unsigned long long rend = fin ? end0 - b0 : std::min(end0 - b0, slice_bits + 4096);
        const unsigned long long rsubs = fin ? (rend + SBITS - 1) / SBITS : slice_bits / SBITS;
        const size_t upto = fin ? n : std::min(n, A0 + (size_t)((b0 + rend) / 8) + 4096);
        if (!st->need_in(upto)) return c.fail(RSN_ERR_DEVICE, "huffman: the upload of a sliced call failed");
        size_t total = 0; uint32_t last = 0;
        rc = decode_range(base0 + b0 / 8, upto - A0 - (size_t)(b0 / 8), entry, rend, rsubs, d_out + produced, out_cap > produced ? out_cap - produced : 0, &total, &last);
        if (rc) return rc;
        if (total > out_cap - std::min(out_cap, produced)) return finish(produced + total, fin ? last : 0, out_cap);   // more than the header announced: the caller decodes it whole
        if (!fin && last == BAD_REL) return c.fail(RSN_ERR_FORMAT, "huffman: payload ends inside a codeword (reference: index out of range, huffman.go:145)");
        if (!st->have_out(produced, total)) return c.fail(RSN_ERR_DEVICE, "huffman: the download of a sliced call failed");
        produced += total;
        if (fin) return finish(produced, last, out_cap);
        entry = last;
    }
    return finish(produced, 0, out_cap);
}

}  // namespace rsn
