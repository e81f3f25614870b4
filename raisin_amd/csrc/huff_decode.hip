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
//                    The first guess assumes fixed-length codes of the minimum
//                    length, which is exact for flat alphabets (config 2a).
//   D2  k_scan_u64   output offsets from the per-block byte counts
//   D3  k_dec_emit   decode again from the now-exact entries, write the bytes.
// Code lookup: a 2^K-entry table in LDS (K = min(maxlen,12)); longer codes finish
// with a bit-by-bit walk of the tree's child array (global, L2 resident).
#include "codecs.h"

namespace rsn {

__global__ void k_scan_u64(const unsigned long long *in, unsigned long long *out, uint32_t n, unsigned long long *total);

constexpr int DB = 256;             // lanes per block
constexpr int SW = 8;               // 32-bit words per subsequence (S = 256 bits)
constexpr int SBITS = SW * 32;
constexpr int DATA_WORDS = DB * SW + 8;   // + overrun for a code that starts inside and ends outside
constexpr int LUT_BITS_MAX = 12;
constexpr uint32_t BAD_REL = 0xFFFF;

struct DecArgs {
    const uint8_t *base;        // 16-byte aligned pointer at or before the first payload byte
    size_t nbytes;              // readable bytes from base
    unsigned long long p0;      // bit position (from base) of the first code bit
    unsigned long long end;     // bit position one past the last payload bit
    uint32_t n_sub;             // number of subsequences
    const uint32_t *lut; int K;
    const int32_t *child;       // 2 per internal node: >=0 internal index, <0 -(rune+1)
    uint32_t min_len; int ascii;
    uint16_t *exit_rel, *entry_rel, *nbyte;
    unsigned long long *blk_bytes;
    int *changed; int pass;
    const unsigned long long *blk_off; uint8_t *out;   // D3
};

__device__ __forceinline__ uint32_t swz(uint32_t j) { return j + (j >> 5); }

__device__ __forceinline__ void stage(const DecArgs &a, uint32_t blk, uint32_t *s_data, const uint32_t *__restrict__ lut_g, uint32_t *s_lut) {
    const size_t w0 = (size_t)blk * DB * SW;
    for (int i = threadIdx.x; i < DATA_WORDS; i += DB) {
        const size_t off = (w0 + i) * 4;
        uint32_t v = 0;
        if (off + 4 <= a.nbytes) v = __builtin_bswap32(*reinterpret_cast<const uint32_t *>(a.base + off));
        else for (int k = 0; k < 4; k++) if (off + k < a.nbytes) v |= (uint32_t)a.base[off + k] << (24 - 8 * k);
        s_data[swz(i)] = v;
    }
    for (int i = threadIdx.x; i < (1 << a.K); i += DB) s_lut[i] = lut_g[i];
}

__device__ __forceinline__ int dev_utf8_len(uint32_t r) { return r < 0x80 ? 1 : r < 0x800 ? 2 : r < 0x10000 ? 3 : 4; }

// Decodes one codeword at block-relative bit `pos`; returns its length, rune in *rune.
__device__ __forceinline__ uint32_t decode_one(const DecArgs &a, const uint32_t *s_data, const uint32_t *s_lut, uint32_t pos, uint32_t *rune) {
    const uint32_t wi = pos >> 5;
    const unsigned long long win = (((unsigned long long)s_data[swz(wi)] << 32) | s_data[swz(wi + 1)]) << (pos & 31);
    const uint32_t ent = s_lut[(uint32_t)(win >> (64 - a.K))];
    if (!(ent & 0x80000000u)) { *rune = ent & 0x1FFFFFu; return ent >> 24; }
    int32_t node = (int32_t)(ent & 0x7FFFFFFFu);
    uint32_t l = a.K;
    for (;;) {
        const uint32_t q = pos + l;
        const uint32_t bit = (s_data[swz(q >> 5)] >> (31 - (q & 31))) & 1;
        const int32_t nxt = a.child[2 * node + bit];
        l++;
        if (nxt < 0) { *rune = (uint32_t)(-(nxt + 1)); return l; }
        if (l >= 64) { *rune = 0; return 65; }   // cannot happen for a tree accepted by the host (codes <= 64 bits)
        node = nxt;
    }
}

// Walk from block-relative bit `pos` to the first code boundary >= lim.
__device__ __forceinline__ void walk(const DecArgs &a, const uint32_t *s_data, const uint32_t *s_lut, uint32_t pos, uint32_t lim,
                                     uint32_t end_rel, uint32_t *exit_pos, uint32_t *nbytes) {
    uint32_t nb = 0;
    bool bad = false;
    while (pos < lim) {
        uint32_t rune;
        const uint32_t l = decode_one(a, s_data, s_lut, pos, &rune);
        if (pos + l > end_rel) { bad = true; break; }   // code would run past the end of the payload
        pos += l;
        nb += a.ascii ? 1 : dev_utf8_len(rune);
    }
    *exit_pos = bad ? 0xFFFFFFFFu : pos;
    *nbytes = nb;
}

__global__ __launch_bounds__(DB) void k_dec_sync(DecArgs a) {
    __shared__ uint32_t s_data[DATA_WORDS + DATA_WORDS / 32 + 2];
    __shared__ uint32_t s_lut[1 << LUT_BITS_MAX];
    __shared__ uint32_t s_exit[DB];
    __shared__ unsigned long long s_part[DB / 64];
    __shared__ int s_skip;
    const int tid = threadIdx.x;
    const uint32_t blk = blockIdx.x;
    const unsigned long long blk_bit0 = (unsigned long long)blk * DB * SBITS;
    const uint32_t g = blk * DB + tid;
    const bool live = g < a.n_sub;
    // block-relative positions (the block spans DB*SBITS bits; exits overshoot by < 64)
    const uint32_t end_rel = (uint32_t)min(a.end - blk_bit0, (unsigned long long)(DB * SBITS + 4096));
    const uint32_t my0 = tid * SBITS;
    const uint32_t lim = min(my0 + SBITS, end_rel);

    // entry of lane 0: the true start for block 0, else the predecessor block's published exit
    if (tid == 0) {
        int skip = 0;
        if (a.pass > 0) {
            unsigned long long e = a.p0;
            if (blk > 0) {
                const uint32_t xr = a.exit_rel[g - 1];
                e = xr == BAD_REL ? ~0ull : min(blk_bit0, a.end) + xr;
            }
            const unsigned long long used = blk_bit0 + a.entry_rel[g];
            skip = (e == used) || (e == ~0ull && a.entry_rel[g] == BAD_REL);
        }
        s_skip = skip;
    }
    __syncthreads();
    if (s_skip) return;
    stage(a, blk, s_data, a.lut, s_lut);

    uint32_t e;   // block-relative entry; 0xFFFFFFFF = predecessor ran off the end
    if (!live) e = 0xFFFFFFFFu;
    else if (g == 0) e = (uint32_t)a.p0;
    else if (a.pass == 0 || tid > 0) {
        // first guess: codes of the minimum length, phase-locked to p0
        const unsigned long long s0 = blk_bit0 + my0;
        unsigned long long q = a.p0;
        if (s0 > a.p0) q = a.p0 + (s0 - a.p0 + a.min_len - 1) / a.min_len * a.min_len;
        e = (uint32_t)(q - blk_bit0);
    } else {
        const uint32_t xr = a.exit_rel[g - 1];
        e = xr == BAD_REL ? 0xFFFFFFFFu : (uint32_t)(min(blk_bit0, a.end) - blk_bit0) + xr;
    }
    bool have = false;
    uint32_t x = 0, nb = 0;
    if (live && a.pass > 0 && tid > 0) {
        // results of the previous pass stay valid while the entry they were computed from stands
        const uint32_t er = a.entry_rel[g];
        const uint32_t xr = a.exit_rel[g];
        e = er == BAD_REL ? 0xFFFFFFFFu : my0 + er;
        x = xr == BAD_REL ? 0xFFFFFFFFu : lim + xr;
        nb = a.nbyte[g];
        have = true;
    }
    __syncthreads();
    for (int round = 0; round <= DB; round++) {
        if (live && !have) {
            if (e == 0xFFFFFFFFu) { x = 0xFFFFFFFFu; nb = 0; }
            else walk(a, s_data, s_lut, e, lim, end_rel, &x, &nb);
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
    }
    if (live) {
        a.entry_rel[g] = e == 0xFFFFFFFFu ? BAD_REL : (uint16_t)(e - my0);
        a.exit_rel[g] = x == 0xFFFFFFFFu ? BAD_REL : (uint16_t)(x - lim);
        a.nbyte[g] = (uint16_t)nb;
    }
    unsigned long long s = live ? nb : 0;
    for (int d = 32; d; d >>= 1) s += __shfl_down(s, d);
    if ((tid & 63) == 0) s_part[tid >> 6] = s;
    __syncthreads();
    if (tid == 0) {
        a.blk_bytes[blk] = s_part[0] + s_part[1] + s_part[2] + s_part[3];
        if (a.pass > 0) *a.changed = 1;
    }
}

// Per-lane byte sink: aligned 8-byte stores, byte stores only for the partial first/last word.
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

__global__ __launch_bounds__(DB) void k_dec_emit(DecArgs a) {
    __shared__ uint32_t s_data[DATA_WORDS + DATA_WORDS / 32 + 2];
    __shared__ uint32_t s_lut[1 << LUT_BITS_MAX];
    __shared__ uint32_t s_wsum[DB / 64];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const uint32_t blk = blockIdx.x;
    const unsigned long long blk_bit0 = (unsigned long long)blk * DB * SBITS;
    const uint32_t g = blk * DB + tid;
    const bool live = g < a.n_sub;
    const uint32_t end_rel = (uint32_t)min(a.end - blk_bit0, (unsigned long long)(DB * SBITS + 4096));
    const uint32_t my0 = tid * SBITS;
    const uint32_t lim = min(my0 + SBITS, end_rel);
    stage(a, blk, s_data, a.lut, s_lut);
    const uint32_t nb = live ? a.nbyte[g] : 0;
    const uint32_t er = live ? a.entry_rel[g] : BAD_REL;
    uint32_t incl = nb;
    for (int d = 1; d < 64; d <<= 1) { const uint32_t y = __shfl_up(incl, d); if (lane >= d) incl += y; }
    if (lane == 63) s_wsum[wv] = incl;
    __syncthreads();
    uint32_t wpre = 0;
    for (int k = 0; k < wv; k++) wpre += s_wsum[k];
    if (!live || er == BAD_REL || nb == 0) return;
    Sink sink;
    sink.start(a.out + a.blk_off[blk] + wpre + incl - nb);
    uint32_t pos = my0 + er;
    while (pos < lim) {
        uint32_t rune;
        const uint32_t l = decode_one(a, s_data, s_lut, pos, &rune);
        if (pos + l > end_rel) break;
        pos += l;
        if (a.ascii || rune < 0x80) sink.put(rune);
        else if (rune < 0x800) { sink.put(0xC0 | (rune >> 6)); sink.put(0x80 | (rune & 0x3F)); }
        else if (rune < 0x10000) { sink.put(0xE0 | (rune >> 12)); sink.put(0x80 | ((rune >> 6) & 0x3F)); sink.put(0x80 | (rune & 0x3F)); }
        else { sink.put(0xF0 | (rune >> 18)); sink.put(0x80 | ((rune >> 12) & 0x3F)); sink.put(0x80 | ((rune >> 6) & 0x3F)); sink.put(0x80 | (rune & 0x3F)); }
    }
    sink.finish();
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

}  // namespace

int huff_decode_dev(Ctx &c, hipStream_t s, const uint8_t *d_in, size_t n, uint8_t *d_out, size_t out_cap, size_t *out_n) {
    if (((uintptr_t)d_in & 15) || ((uintptr_t)d_out & 15)) return c.fail(RSN_ERR_ARG, "huffman: device buffers must be 16-byte aligned");
    *out_n = 0;
    // ---- header to the host: strings.SplitN(content, "\\\n", 2) (huffman.go:261)
    std::vector<uint8_t> head;
    size_t sep = (size_t)-1;
    for (size_t want = 1 << 16;; want *= 8) {
        const size_t k = std::min(want, n);
        const size_t old = head.size();
        head.resize(k);
        if (k > old) {
            RSN_HIP(hipMemcpyAsync(head.data() + old, d_in + old, k - old, hipMemcpyDeviceToHost, s));
            RSN_HIP(hipStreamSynchronize(s));
        }
        for (size_t i = old ? old - 1 : 0; i + 1 < k; i++) if (head[i] == 0x5C && head[i + 1] == 0x0A) { sep = i; break; }
        if (sep != (size_t)-1 || k == n) break;
    }
    if (sep == (size_t)-1) return c.fail(RSN_ERR_FORMAT, "huffman: no '\\\\\\n' separator (reference: index out of range, huffman.go:264)");
    if (sep + 3 > head.size() && sep + 3 <= n) {   // make sure the pad byte is on the host
        const size_t old = head.size();
        head.resize(sep + 3);
        RSN_HIP(hipMemcpyAsync(head.data() + old, d_in + old, sep + 3 - old, hipMemcpyDeviceToHost, s));
        RSN_HIP(hipStreamSynchronize(s));
    }
    std::vector<HuffSym> syms; std::string msg;
    if (!parse_header(head.data(), sep, syms, msg)) return c.fail(RSN_ERR_FORMAT, "%s", msg.c_str());
    HuffTree tree; HuffCodes codes;
    if (!build_tree(syms, tree, msg)) return c.fail(RSN_ERR_FORMAT, "%s", msg.c_str());
    if (!assign_codes(tree, codes, msg)) return c.fail(RSN_ERR_LIMIT, "%s", msg.c_str());

    const size_t sn = n - sep - 2;                       // bytes after the separator
    const unsigned diff = sn ? head[sep + 2] : 0;         // byteArr[0] (huffman.go:275)
    const unsigned long long nbits = sn ? (unsigned long long)(sn - 1) * 8 : 0;
    if (diff > nbits) return c.fail(RSN_ERR_FORMAT, "huffman: pad exceeds the payload (reference: slice bounds out of range, huffman.go:294)");
    const unsigned long long max = nbits - diff;

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

    // ---- tables
    const int K = (int)std::min<unsigned>(codes.max_len, LUT_BITS_MAX);
    std::vector<uint32_t> lut; std::vector<int32_t> child;
    build_tables(tree, K, lut, child);
    void *p;
    int rc = dev_buf(c, 5, lut.size() * 4 + child.size() * 4, &p); if (rc) return rc;
    uint32_t *d_lut = (uint32_t *)p;
    int32_t *d_child = (int32_t *)(d_lut + lut.size());
    RSN_HIP(hipMemcpyAsync(d_lut, lut.data(), lut.size() * 4, hipMemcpyHostToDevice, s));
    RSN_HIP(hipMemcpyAsync(d_child, child.data(), child.size() * 4, hipMemcpyHostToDevice, s));

    DecArgs a{};
    const size_t pay = sep + 3;                           // first payload byte
    const size_t A0 = pay & ~(size_t)15;
    a.base = d_in + A0; a.nbytes = n - A0;
    a.p0 = 8ull * (pay - A0) + diff;
    a.end = 8ull * (n - A0);
    const unsigned long long n_sub64 = (a.end + SBITS - 1) / SBITS;
    if (n_sub64 > 0xFFFFFF00ull) return c.fail(RSN_ERR_LIMIT, "huffman: payload too large for one call");
    a.n_sub = (uint32_t)n_sub64;
    a.lut = d_lut; a.K = K; a.child = d_child; a.min_len = codes.min_len;
    bool ascii = true;
    for (uint32_t i = 0; i < tree.n_leaves; i++) if (tree.rune[i] >= 0x80) ascii = false;
    a.ascii = ascii;
    const uint32_t n_blk = (uint32_t)ceil_div(a.n_sub, DB);
    rc = dev_buf(c, 6, (size_t)a.n_sub * 6 + 64, &p); if (rc) return rc;
    a.exit_rel = (uint16_t *)p; a.entry_rel = a.exit_rel + a.n_sub; a.nbyte = a.entry_rel + a.n_sub;
    rc = dev_buf(c, 7, ((size_t)n_blk * 2 + 4) * 8, &p); if (rc) return rc;
    a.blk_bytes = (unsigned long long *)p;
    unsigned long long *d_blk_off = a.blk_bytes + n_blk;
    unsigned long long *d_total = d_blk_off + n_blk;
    int *d_changed = (int *)(d_total + 1);
    a.changed = d_changed;
    void *hp; rc = pinned_buf(c, 64, &hp); if (rc) return rc;
    volatile int *h_changed = (volatile int *)hp;

    // ---- D1: iterate to the fixed point
    a.pass = 0;
    RSN_LAUNCH("huff_dec_sync", k_dec_sync, dim3(n_blk), dim3(DB), 0, s, a);
    for (uint32_t pass = 1;; pass++) {
        if (pass > n_blk + 2) return c.fail(RSN_ERR_DEVICE, "huffman: synchronisation did not converge");
        a.pass = (int)pass;
        RSN_HIP(hipMemsetAsync(d_changed, 0, 4, s));
        RSN_LAUNCH("huff_dec_sync", k_dec_sync, dim3(n_blk), dim3(DB), 0, s, a);
        RSN_HIP(hipMemcpyAsync((void *)h_changed, d_changed, 4, hipMemcpyDeviceToHost, s));
        RSN_HIP(hipStreamSynchronize(s));
        if (!*h_changed) break;
    }
    // ---- D2: offsets, total, validity of the final exit
    RSN_LAUNCH("huff_dec_scan", k_scan_u64, dim3(1), dim3(1024), 0, s, a.blk_bytes, d_blk_off, n_blk, d_total);
    struct Tail { unsigned long long total; uint16_t last_exit; };
    Tail *ht = (Tail *)hp;
    RSN_HIP(hipMemcpyAsync(&ht->total, d_total, 8, hipMemcpyDeviceToHost, s));
    RSN_HIP(hipMemcpyAsync(&ht->last_exit, a.exit_rel + (a.n_sub - 1), 2, hipMemcpyDeviceToHost, s));
    RSN_HIP(hipStreamSynchronize(s));
    if (ht->last_exit != 0)
        return c.fail(RSN_ERR_FORMAT, "huffman: payload ends inside a codeword (reference: index out of range, huffman.go:145)");
    const size_t total = (size_t)ht->total;
    *out_n = total;
    if (!d_out || total > out_cap) { *out_n = round_up(total, 16) + 16; return c.fail(RSN_ERR_CAPACITY, "huffman: output needs %zu bytes, buffer holds %zu", total, out_cap); }
    // ---- D3
    a.blk_off = d_blk_off; a.out = d_out;
    RSN_LAUNCH("huff_dec_emit", k_dec_emit, dim3(n_blk), dim3(DB), 0, s, a);
    RSN_HIP(hipStreamSynchronize(s));
    return RSN_OK;
}

}  // namespace rsn
