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

namespace {
// Go container/heap (go1.15 src/container/heap/heap.go) over node ids with
// Less(i,j) = freq[i] < freq[j] (huffman.go:43-45).  Ties are broken only by
// the sift order below, so it is reproduced exactly.
class GoHeap {
  public:
    GoHeap(std::vector<int32_t> &items, const std::vector<uint64_t> &freq) : h_(items), f_(freq) {}
    void init() { const int n = (int)h_.size(); for (int i = n / 2 - 1; i >= 0; i--) down(i, n); }
    void push(int32_t x) { h_.push_back(x); up((int)h_.size() - 1); }
    int32_t pop() {
        const int n = (int)h_.size() - 1;
        std::swap(h_[0], h_[n]);
        down(0, n);
        const int32_t x = h_.back();
        h_.pop_back();
        return x;
    }
    size_t size() const { return h_.size(); }

  private:
    bool less(int a, int b) const { return f_[h_[a]] < f_[h_[b]]; }
    void up(int j) {
        for (;;) {
            const int i = (j - 1) / 2;   // parent; truncating division, so j=0 gives 0
            if (i == j || !less(j, i)) break;
            std::swap(h_[i], h_[j]);
            j = i;
        }
    }
    void down(int i, int n) {
        for (;;) {
            const int l = 2 * i + 1;
            if (l >= n || l < 0) break;
            int j = l;
            if (l + 1 < n && less(l + 1, l)) j = l + 1;
            if (!less(j, i)) break;
            std::swap(h_[i], h_[j]);
            i = j;
        }
    }
    std::vector<int32_t> &h_;
    const std::vector<uint64_t> &f_;
};
}  // namespace

bool build_tree(std::vector<HuffSym> &syms, HuffTree &t, std::string &msg) {
    const size_t a = syms.size();
    if (a == 0) { msg = "huffman: no symbols (reference panics in heap.Pop on an empty heap, huffman.go:102)"; return false; }
    std::sort(syms.begin(), syms.end(), [](const HuffSym &x, const HuffSym &y) {
        return x.freq != y.freq ? x.freq < y.freq : x.rune < y.rune;
    });
    t = HuffTree();
    t.n_leaves = (uint32_t)a;
    t.freq.reserve(2 * a); t.left.reserve(2 * a); t.right.reserve(2 * a); t.rune.reserve(2 * a);
    std::vector<int32_t> items(a);
    for (size_t i = 0; i < a; i++) {
        t.freq.push_back(syms[i].freq); t.rune.push_back(syms[i].rune);
        t.left.push_back(-1); t.right.push_back(-1);
        items[i] = (int32_t)i;
    }
    GoHeap heap(items, t.freq);
    heap.init();                                    // huffman.go:93
    while (heap.size() > 1) {                       // huffman.go:96-101
        const int32_t x = heap.pop();
        const int32_t y = heap.pop();
        const int32_t id = (int32_t)t.freq.size();
        t.freq.push_back(t.freq[x] + t.freq[y]);
        t.left.push_back(x); t.right.push_back(y); t.rune.push_back(0);
        heap.push(id);
    }
    t.root = heap.pop();                            // huffman.go:102
    return true;
}

bool assign_codes(const HuffTree &t, HuffCodes &c, std::string &msg) {
    const size_t a = t.n_leaves;
    c.code.assign(a, 0); c.len.assign(a, 0); c.dfs.clear(); c.dfs.reserve(a);
    c.min_len = ~0u; c.max_len = 0; c.total_bits = 0;
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
    auto entry = [&](const HuffSym &s) {
        char num[24];                                                                   // strconv.Itoa(val)
        int k = 0;
        uint64_t v = s.freq;
        do { num[k++] = (char)('0' + v % 10); v /= 10; } while (v);
        while (k) out.push_back(num[--k]);
        out.push_back('|');
        if (s.rune == 10) { out.append("\\n"); return; }                                // huffman.go:316
        uint8_t u[4];
        out.append((const char *)u, (size_t)go_encode_rune(s.rune, u));
    };
    // '\\' as the LAST entry makes the reference decoder index past the header
    // (huffman.go:210); any order is a legal reference output, so move it first.
    const bool move_bs = a > 1 && by_rune.back().rune == 0x5C;
    if (move_bs) entry(by_rune.back());
    for (size_t i = 0; i + (move_bs ? 1 : 0) < a; i++) entry(by_rune[i]);
}

bool parse_header(const uint8_t *h, size_t n, std::vector<HuffSym> &syms, std::string &msg) {
    // symFreqs (huffman.go:197): a map, so a later entry for the same rune overwrites an earlier one.
    // Kept as an append-only list with a sequence number, resolved by one stable sort at the end.
    struct Ent { uint32_t rune; uint64_t freq; };
    std::vector<Ent> table;
    table.reserve(256);
    uint64_t acc = 0;
    int digits = 0;
    for (size_t i = 0; i < n; i++) {
        const uint8_t ch = h[i];
        if (ch != '|') {
            if (ch >= '0' && ch <= '9') {             // only single digits pass strconv.Atoi (huffman.go:203)
                if (++digits > 18) { msg = "huffman: frequency has more than 18 digits"; return false; }
                acc = acc * 10 + (ch - '0');
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
    std::stable_sort(table.begin(), table.end(), [](const Ent &a, const Ent &b) { return a.rune < b.rune; });
    syms.clear();
    syms.reserve(table.size());
    for (size_t i = 0; i < table.size(); i++) {
        if (i + 1 < table.size() && table[i + 1].rune == table[i].rune) continue;   // overwritten by a later entry
        syms.push_back({table[i].rune, table[i].freq});
    }
    return true;
}

}  // namespace rsn
