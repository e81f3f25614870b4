// huff_host.cpp -- see huff_host.h.
#include "huff_host.h"

#include <algorithm>
#include <cstdio>
#include <vector>

namespace rsn {

// unicode/utf8 accept ranges (go1.15 src/unicode/utf8/utf8.go), as applied by
// the `range` loops at huffman.go:235,309: an invalid or short sequence is
// U+FFFD and advances ONE byte.
uint32_t go_decode_rune(const uint8_t *p, size_t avail, int *size) {
    const uint8_t b0 = p[0];
    *size = 1;
    if (b0 < 0x80) return b0;
    if (b0 < 0xC2 || b0 > 0xF4) return kRuneError;
    const int need = b0 < 0xE0 ? 2 : b0 < 0xF0 ? 3 : 4;
    if (avail < (size_t)need) return kRuneError;
    uint8_t lo = 0x80, hi = 0xBF;
    switch (b0) {
        case 0xE0: lo = 0xA0; break;
        case 0xED: hi = 0x9F; break;
        case 0xF0: lo = 0x90; break;
        case 0xF4: hi = 0x8F; break;
        default: break;
    }
    if (p[1] < lo || p[1] > hi) return kRuneError;
    uint32_t r = need == 2 ? (b0 & 0x1Fu) : need == 3 ? (b0 & 0x0Fu) : (b0 & 0x07u);
    r = (r << 6) | (p[1] & 0x3Fu);
    for (int k = 2; k < need; k++) {
        if ((p[k] & 0xC0) != 0x80) return kRuneError;
        r = (r << 6) | (p[k] & 0x3Fu);
    }
    *size = need;
    return r;
}

int go_encode_rune(uint32_t r, uint8_t o[4]) {
    if (r >= kMaxRune || (r >= 0xD800 && r < 0xE000)) r = kRuneError;
    switch (utf8_len(r)) {
        case 1: o[0] = (uint8_t)r; return 1;
        case 2: o[0] = 0xC0 | (r >> 6); o[1] = 0x80 | (r & 0x3F); return 2;
        case 3: o[0] = 0xE0 | (r >> 12); o[1] = 0x80 | ((r >> 6) & 0x3F); o[2] = 0x80 | (r & 0x3F); return 3;
        default: o[0] = 0xF0 | (r >> 18); o[1] = 0x80 | ((r >> 12) & 0x3F); o[2] = 0x80 | ((r >> 6) & 0x3F); o[3] = 0x80 | (r & 0x3F); return 4;
    }
}

// does a rune of Go's decoding of in[0, n) begin at q?  A byte that is not a continuation byte always begins one (a valid sequence
// or U+FFFD); a continuation byte does unless a valid sequence that begins at one of the three bytes before it reaches it -- and the
// nearest non-continuation byte before q is the only one that can begin such a sequence.
static bool is_rune_start(const uint8_t *in, size_t n, size_t q) {
    if ((in[q] & 0xC0) != 0x80) return true;
    for (size_t k = 1; k <= 3 && k <= q; k++) {
        if ((in[q - k] & 0xC0) == 0x80) continue;
        int sz = 1;
        (void)go_decode_rune(in + (q - k), n - (q - k), &sz);
        return (size_t)sz <= k;
    }
    return true;                                                          // three continuation bytes before it: no sequence is that long
}

void huff_slice_cuts(const uint8_t *in, size_t n, int shards, std::vector<size_t> &cut) {
    cut.assign(1, 0);
    size_t G = (size_t)std::max(1, std::min(shards, 256));
    G = std::min(G, std::max<size_t>(1, n / 64));                        // (a slice of a few bytes is all overhead)
    for (size_t w = 1; w < G; w++) {
        size_t p = (size_t)((unsigned __int128)n * w / G);
        while (p > cut.back() && !is_rune_start(in, n, p)) p--;
        if (p > cut.back()) cut.push_back(p);
    }
    cut.push_back(n);
}

namespace {
// Go container/heap (go1.15 src/container/heap/heap.go) over tree nodes with
// Less(i,j) = freq[i] < freq[j] (huffman.go:43-45).  Ties are broken only by
// the sift order below, so it is reproduced exactly.  Items carry their
// frequency inline (Wide: the general form; when every frequency and their sum fit
// 32 bits GoHeapSplit below does the same on two arrays).
struct Wide {
    uint64_t f; int32_t i;
    static Wide make(uint64_t f, int32_t id) { return {f, id}; }
    uint64_t freq() const { return f; }
    int32_t id() const { return i; }
};

template <typename Item>
class GoHeap {
  public:
    explicit GoHeap(std::vector<Item> &items) : h_(items) {}
    void init() { const int n = (int)h_.size(); for (int i = n / 2 - 1; i >= 0; i--) down(i, n); }
    void push(Item x) { h_.push_back(x); up((int)h_.size() - 1); }
    Item pop() {
        const int n = (int)h_.size() - 1;
        std::swap(h_[0], h_[n]);
        down(0, n);
        const Item x = h_.back();
        h_.pop_back();
        return x;
    }
    size_t size() const { return h_.size(); }

  private:
    // up/down carry the moving item in a register and shift the others past it: the same final
    // layout as heap.go's swap at every level, with half the stores.
    void up(int j) {
        Item *h = h_.data();
        const Item x = h[j];
        for (;;) {
            const int i = (j - 1) / 2;   // parent; truncating division, so j=0 gives 0
            if (i == j || !(x.freq() < h[i].freq())) break;
            h[j] = h[i];
            j = i;
        }
        h[j] = x;
    }
    void down(int i, int n) {
        Item *h = h_.data();
        const Item x = h[i];
        const uint64_t fx = x.freq();
        // (r06) three levels a step while the node's whole three-level subtree exists: the path's next three nodes from fourteen loads that
        // depend on i alone -- heap.go's choice at every level, the right child only if it is strictly less -- instead of three rounds of
        // load, compare, select (the sift is a chain of dependent loads: 2b's 3 * 10^5 leaves, 6 * 10^5 sifts of 18 levels)
        while (8 * (long long)i + 14 < (long long)n) {
            const Item *a = h + 2 * i + 1, *b = h + 4 * i + 3, *c = h + 8 * i + 7;
            const unsigned r1 = a[1].freq() < a[0].freq();
            const unsigned m2 = (unsigned)(b[1].freq() < b[0].freq()) | ((unsigned)(b[3].freq() < b[2].freq()) << 1);
            const unsigned m3 = (unsigned)(c[1].freq() < c[0].freq()) | ((unsigned)(c[3].freq() < c[2].freq()) << 1) |
                                ((unsigned)(c[5].freq() < c[4].freq()) << 2) | ((unsigned)(c[7].freq() < c[6].freq()) << 3);
            const unsigned i2 = 2 * r1 + ((m2 >> r1) & 1u), i3 = 2 * i2 + ((m3 >> i2) & 1u);
            const int j1 = 2 * i + 1 + (int)r1, j2 = 4 * i + 3 + (int)i2, j3 = 8 * i + 7 + (int)i3;
            __builtin_prefetch(h + std::min(8LL * j3 + 7, (long long)n - 1));
            __builtin_prefetch(h + std::min(8LL * j3 + 14, (long long)n - 1));
            if (!(h[j1].freq() < fx)) { h[i] = x; return; }
            h[i] = h[j1];
            if (!(h[j2].freq() < fx)) { h[j1] = x; return; }
            h[j1] = h[j2];
            if (!(h[j3].freq() < fx)) { h[j2] = x; return; }
            h[j2] = h[j3];
            i = j3;
        }
        for (;;) {
            const int l = 2 * i + 1;
            if (l >= n || l < 0) break;
            __builtin_prefetch(h + std::min(4 * i + 3, n - 1));   // grandchildren: the loads below are a dependent chain
            int j = l;
            if (l + 1 < n) j += (int)(h[l + 1].freq() < h[l].freq());
            if (!(h[j].freq() < x.freq())) break;
            h[i] = h[j];
            i = j;
        }
        h[i] = x;
    }
    std::vector<Item> &h_;
};

template <typename Item>
void run_heap(HuffTree &t, size_t a) {
    std::vector<Item> items;
    items.reserve(a + 1);
    for (size_t i = 0; i < a; i++) items.push_back(Item::make(t.freq[i], (int32_t)i));
    GoHeap<Item> heap(items);
    heap.init();                                    // huffman.go:93
    while (heap.size() > 1) {                       // huffman.go:96-101
        const Item x = heap.pop();
        const Item y = heap.pop();
        const int32_t id = (int32_t)t.freq.size();
        const uint64_t f = x.freq() + y.freq();
        t.freq.push_back(f);
        t.left.push_back(x.id()); t.right.push_back(y.id()); t.rune.push_back(0);
        heap.push(Item::make(f, id));
    }
    t.root = heap.pop().id();                       // huffman.go:102
}

// (r06) Every frequency and their sum below 2^32 (the usual case): frequencies and ids in arrays of their own -- the path of a sift is
// found in the frequencies alone, half the bytes to bring in -- and three levels a step as above.  The same heap.go, the same order.
// (2b-like table of 314 299 leaves, this container: build_tree 51-53 ms -> 41-43; four levels a step: 44; the second of the two pops
//  sifting three levels behind the first, with an undo for the case that the first takes the second's item: no faster than one after
//  the other -- out-of-order execution already overlaps what can be.)
class GoHeapSplit {
  public:
    explicit GoHeapSplit(size_t a) : F(a + 1), ID(a + 1), n(0) {}
    std::vector<uint32_t> F, ID;
    int n;
    void push(uint32_t f, uint32_t id) {
        int j = n++;
        uint32_t *f_ = F.data(), *d_ = ID.data();
        for (;;) {
            const int i = (j - 1) / 2;
            if (i == j || !(f < f_[i])) break;
            f_[j] = f_[i]; d_[j] = d_[i];
            j = i;
        }
        f_[j] = f; d_[j] = id;
    }
    void pop(uint32_t &of, uint32_t &oid) {
        uint32_t *f_ = F.data(), *d_ = ID.data();
        const int m = n - 1;
        of = f_[0]; oid = d_[0];
        const uint32_t fx = f_[m], dx = d_[m];
        f_[m] = of; d_[m] = oid;                                       // (heap.go swaps; the slot is dropped right after)
        n = m;
        int i = 0;
        while (8 * (long long)i + 14 < (long long)m) {
            const uint32_t *a = f_ + 2 * i + 1, *b = f_ + 4 * i + 3, *c = f_ + 8 * i + 7;
            const unsigned r1 = a[1] < a[0];
            const unsigned m2 = (unsigned)(b[1] < b[0]) | ((unsigned)(b[3] < b[2]) << 1);
            const unsigned m3 = (unsigned)(c[1] < c[0]) | ((unsigned)(c[3] < c[2]) << 1) | ((unsigned)(c[5] < c[4]) << 2) | ((unsigned)(c[7] < c[6]) << 3);
            const unsigned i2 = 2 * r1 + ((m2 >> r1) & 1u), i3 = 2 * i2 + ((m3 >> i2) & 1u);
            const int j1 = 2 * i + 1 + (int)r1, j2 = 4 * i + 3 + (int)i2, j3 = 8 * i + 7 + (int)i3;
            __builtin_prefetch(f_ + std::min(8LL * j3 + 7, (long long)m - 1));
            __builtin_prefetch(d_ + j1); __builtin_prefetch(d_ + j2); __builtin_prefetch(d_ + j3);
            if (!(f_[j1] < fx)) { f_[i] = fx; d_[i] = dx; return; }
            f_[i] = f_[j1]; d_[i] = d_[j1];
            if (!(f_[j2] < fx)) { f_[j1] = fx; d_[j1] = dx; return; }
            f_[j1] = f_[j2]; d_[j1] = d_[j2];
            if (!(f_[j3] < fx)) { f_[j2] = fx; d_[j2] = dx; return; }
            f_[j2] = f_[j3]; d_[j2] = d_[j3];
            i = j3;
        }
        for (;;) {
            const int l = 2 * i + 1;
            if (l >= m || l < 0) break;
            int j = l;
            if (l + 1 < m) j += (int)(f_[l + 1] < f_[l]);
            if (!(f_[j] < fx)) break;
            f_[i] = f_[j]; d_[i] = d_[j];
            i = j;
        }
        f_[i] = fx; d_[i] = dx;
    }
};
void run_heap_split(HuffTree &t, size_t a) {
    GoHeapSplit hp(a);
    for (size_t i = 0; i < a; i++) { hp.F[i] = (uint32_t)t.freq[i]; hp.ID[i] = (uint32_t)i; }   // ascending: a heap already (heap.Init, huffman.go:93, moves nothing)
    hp.n = (int)a;
    while (hp.n > 1) {                              // huffman.go:96-101
        uint32_t fx, ix, fy, iy;
        hp.pop(fx, ix); hp.pop(fy, iy);
        const int32_t id = (int32_t)t.freq.size();
        const uint32_t f = fx + fy;
        t.freq.push_back(f);
        t.left.push_back((int32_t)ix); t.right.push_back((int32_t)iy); t.rune.push_back(0);
        hp.push(f, (uint32_t)id);
    }
    uint32_t fr, ir;
    hp.pop(fr, ir);                                 // huffman.go:102
    t.root = (int32_t)ir;
}

// (freq asc, rune asc).  Callers nearly always hand the table over ascending by rune, where a
// stable LSD radix sort on the frequency alone gives the same order in a few linear passes.
void sort_leaves(std::vector<HuffSym> &syms) {
    const size_t a = syms.size();
    bool by_rune = true;
    uint64_t fmax = 0;
    for (size_t i = 0; i < a; i++) {
        fmax = std::max(fmax, syms[i].freq);
        if (i && syms[i - 1].rune >= syms[i].rune) by_rune = false;
    }
    if (!by_rune || a < 4096) {
        std::sort(syms.begin(), syms.end(), [](const HuffSym &x, const HuffSym &y) {
            return x.freq != y.freq ? x.freq < y.freq : x.rune < y.rune;
        });
        return;
    }
    std::vector<HuffSym> tmp(a);
    if (fmax >= 2048) {
        // (r06) A rune alphabet's table is nearly all small counts with a few large ones (2b: 3 * 10^5 runes, 99 % below 100, byte values
        // in the millions): ONE counting pass places the counts below 2048, the others follow in the table's order and are sorted among
        // themselves -- stable, so equal counts stay ascending by rune.  (Three 11-bit radix passes over 5 MB before: 8 of build_tree's
        // 42 ms on a 2b-like table in this container; now 3.)
        size_t count[2049] = {0}, n_big = 0;
        for (size_t i = 0; i < a; i++) { if (syms[i].freq < 2048) count[syms[i].freq + 1]++; else n_big++; }
        for (int k = 0; k < 2048; k++) count[k + 1] += count[k];
        size_t big_at = a - n_big;
        for (size_t i = 0; i < a; i++) { if (syms[i].freq < 2048) tmp[count[syms[i].freq]++] = syms[i]; else tmp[big_at++] = syms[i]; }
        std::stable_sort(tmp.begin() + (long)(a - n_big), tmp.end(), [](const HuffSym &x, const HuffSym &y) { return x.freq < y.freq; });
        syms.swap(tmp);
        return;
    }
    HuffSym *src = syms.data(), *dst = tmp.data();
    for (unsigned shift = 0; shift < 64 && (fmax >> shift) != 0; shift += 11) {
        size_t count[2049] = {0};
        for (size_t i = 0; i < a; i++) count[((src[i].freq >> shift) & 2047) + 1]++;
        for (int k = 0; k < 2048; k++) count[k + 1] += count[k];
        for (size_t i = 0; i < a; i++) dst[count[(src[i].freq >> shift) & 2047]++] = src[i];
        std::swap(src, dst);
    }
    if (src != syms.data()) syms.swap(tmp);
}
}  // namespace

bool build_tree(std::vector<HuffSym> &syms, HuffTree &t, std::string &msg) {
    const size_t a = syms.size();
    if (a == 0) { msg = "huffman: no symbols (reference panics in heap.Pop on an empty heap, huffman.go:102)"; return false; }
    sort_leaves(syms);
    t = HuffTree();
    t.n_leaves = (uint32_t)a;
    t.freq.reserve(2 * a); t.left.reserve(2 * a); t.right.reserve(2 * a); t.rune.reserve(2 * a);
    uint64_t total = 0;
    bool wide = a >= (1u << 30);
    for (size_t i = 0; i < a; i++) {
        t.freq.push_back(syms[i].freq); t.rune.push_back(syms[i].rune);
        t.left.push_back(-1); t.right.push_back(-1);
        if (syms[i].freq >> 32) wide = true;
        total += syms[i].freq & 0xFFFFFFFFull;
    }
    if (wide || (total >> 32)) run_heap<Wide>(t, a);
    else run_heap_split(t, a);
    return true;
}

bool assign_codes(const HuffTree &t, HuffCodes &c, std::string &msg, bool want_dfs) {
    const size_t a = t.n_leaves;
    c.code.assign(a, 0); c.len.assign(a, 0); c.dfs.clear();
    c.min_len = ~0u; c.max_len = 0; c.total_bits = 0;
    if (!want_dfs) {
        // A node is created after both its children (run_heap), so ids fall from the root down: one pass in descending id order sees
        // every parent before its children.  code = the path from the root, '0' to the left (huffman.go:118-119).
        const size_t nn = t.freq.size();
        if ((size_t)t.root + 1 != nn && !(a == 1 && t.root == 0)) { msg = "huffman: internal error (tree ids out of order)"; return false; }
        std::vector<uint64_t> code(nn, 0);
        std::vector<uint8_t> len(nn, 0);
        for (size_t id = nn; id-- > a;) {
            const uint32_t l = (uint32_t)len[id] + 1;
            if (l > 64) { msg = "huffman: code longer than 64 bits"; return false; }
            const uint64_t base = l <= 64 && len[id] < 64 ? code[id] << 1 : 0;
            const int32_t le = t.left[id], ri = t.right[id];
            code[le] = base; len[le] = (uint8_t)l;
            code[ri] = base | 1; len[ri] = (uint8_t)l;
        }
        for (size_t i = 0; i < a; i++) {
            const uint32_t l = len[i];
            c.code[i] = code[i]; c.len[i] = (uint8_t)l;
            c.min_len = std::min(c.min_len, l); c.max_len = std::max(c.max_len, l);
            c.total_bits += t.freq[i] * l;
        }
        return true;
    }
    c.dfs.reserve(a);
    struct Item { int32_t node; uint64_t code; uint32_t len; };
    std::vector<Item> stack;
    stack.push_back({t.root, 0, 0});
    while (!stack.empty()) {
        const Item it = stack.back();
        stack.pop_back();
        if (t.is_leaf(it.node)) {
            if (it.len > 64) { msg = "huffman: code longer than 64 bits"; return false; }
            c.code[it.node] = it.code; c.len[it.node] = (uint8_t)it.len;
            c.dfs.push_back((uint32_t)it.node);
            c.min_len = std::min(c.min_len, it.len); c.max_len = std::max(c.max_len, it.len);
            c.total_bits += t.freq[it.node] * it.len;
            continue;
        }
        const uint64_t base = it.len < 64 ? it.code << 1 : 0;
        stack.push_back({t.right[it.node], base | 1, it.len + 1});   // visited second
        stack.push_back({t.left[it.node], base, it.len + 1});        // '0' first (huffman.go:118-119)
    }
    return true;
}

void emit_header(const std::vector<HuffSym> &by_rune, std::string &out) {
    const size_t a = by_rune.size();
    const size_t base = out.size();
    out.resize(base + 26 * a);                                                          // at most 20 digits + '|' + 4 bytes per entry; trimmed below
    char *w = &out[base];                                                               // (written through a raw pointer: 3*10^5 entries on config 2b)
    auto entry = [&](const HuffSym &s) {
        char num[24];                                                                   // strconv.Itoa(val)
        int k = 0;
        uint64_t v = s.freq;
        do { num[k++] = (char)('0' + v % 10); v /= 10; } while (v);
        while (k) *w++ = num[--k];
        *w++ = '|';
        if (s.rune == 10) { *w++ = '\\'; *w++ = 'n'; return; }                          // huffman.go:316
        w += go_encode_rune(s.rune, (uint8_t *)w);
    };
    // '\\' as the LAST entry makes the reference decoder index past the header
    // (huffman.go:210); any order is a legal reference output, so move it first.
    const bool move_bs = a > 1 && by_rune.back().rune == 0x5C;
    if (move_bs) entry(by_rune.back());
    for (size_t i = 0; i + (move_bs ? 1 : 0) < a; i++) entry(by_rune[i]);
    out.resize((size_t)(w - out.data()));
}

bool parse_header(const uint8_t *h, size_t n, std::vector<HuffSym> &syms, std::string &msg) {
    // symFreqs (huffman.go:197): a map, so a later entry for the same rune overwrites an earlier one.
    // Kept as an append-only list, resolved by one stable sort at the end.
    struct Ent { uint32_t rune; uint64_t freq; };
    std::vector<Ent> table;
    table.reserve(std::max<size_t>(256, n / 4));      // an entry is at least three bytes; rune alphabets bring 10^5 of them
    uint64_t acc = 0;
    int digits = 0;
    for (size_t i = 0; i < n; i++) {
        const uint8_t ch = h[i];
        if (ch != '|') {
            if (ch >= '0' && ch <= '9') {             // only single digits pass strconv.Atoi (huffman.go:203)
                // strconv.Atoi of the digit string: any number of leading zeros, values up to 2^63-1 (19 digits); beyond
                // that Atoi returns MaxInt64 with an ErrRange the reference drops (huffman.go:207)
                const uint64_t kMax = 0x7FFFFFFFFFFFFFFFull, d = (uint64_t)(ch - '0');
                acc = (acc > (kMax - d) / 10) ? kMax : acc * 10 + d;
                (void)digits;
            }
            continue;
        }
        const uint64_t f = acc;                       // Atoi("") == 0 with the error dropped (huffman.go:207)
        acc = 0; digits = 0;
        if (i + 1 >= n) { msg = "huffman: header ends after '|' (reference: index out of range, huffman.go:210)"; return false; }
        if (h[i + 1] == '\\') {
            if (i + 2 >= n) { msg = "huffman: header ends after '\\' (reference: index out of range, huffman.go:210)"; return false; }
            if (h[i + 2] == 'n') { table.push_back({10, f}); i += 2; continue; }   // huffman.go:211-212,222
        }
        int sz;
        const uint32_t r = go_decode_rune(h + i + 1, n - (i + 1), &sz);  // the rune that starts at i+1 (huffman.go:214-220)
        table.push_back({r, f});
        i += 1;                                       // huffman.go:222: one byte is skipped, whatever the rune's width
    }
    bool ascending = true;                            // this library's own headers already are
    for (size_t i = 1; i < table.size() && ascending; i++) ascending = table[i - 1].rune <= table[i].rune;
    if (!ascending) std::stable_sort(table.begin(), table.end(), [](const Ent &a, const Ent &b) { return a.rune < b.rune; });
    syms.clear();
    syms.reserve(table.size());
    for (size_t i = 0; i < table.size(); i++) {
        if (i + 1 < table.size() && table[i + 1].rune == table[i].rune) continue;   // overwritten by a later entry
        syms.push_back({table[i].rune, table[i].freq});
    }
    return true;
}

}  // namespace rsn
