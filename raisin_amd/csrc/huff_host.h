// huff_host.h -- host half of the Huffman codec: what stays on the CPU by design
// (a few hundred symbols of tree building; the bytes never pass through here).
// Mirrors the tree/code/header semantics of
// /root/reference/compressor/huffman/huffman.go (line cites inline).
#pragma once

#include <cstdint>
#include <string>
#include <vector>

namespace rsn {

constexpr uint32_t kRuneError = 0xFFFD;
constexpr uint32_t kMaxRune = 0x110000;   // table size for rune-indexed arrays

struct HuffSym {
    uint32_t rune;
    uint64_t freq;
};

// Leaves are nodes [0, n_leaves) in (freq asc, rune asc) order (huffman.go:64-87);
// internal nodes follow in creation order (huffman.go:96-101).
struct HuffTree {
    uint32_t n_leaves = 0;
    int32_t root = -1;
    std::vector<uint64_t> freq;
    std::vector<int32_t> left, right;   // -1 on leaves
    std::vector<uint32_t> rune;         // leaves only
    bool is_leaf(int32_t n) const { return left[n] < 0; }
};

struct HuffCodes {
    std::vector<uint64_t> code;   // per leaf id, MSB-first, right aligned
    std::vector<uint8_t> len;     // per leaf id
    std::vector<uint32_t> dfs;    // leaf ids in printCodes order (huffman.go:110-127)
    unsigned min_len = 0, max_len = 0;
    uint64_t total_bits = 0;      // sum freq*len
};

// Go utf8 decoding as done by `range string` (huffman.go:235,309).
uint32_t go_decode_rune(const uint8_t *p, size_t avail, int *size);
int go_encode_rune(uint32_t r, uint8_t out[4]);   // string(rune) (huffman.go:138,314)
static inline int utf8_len(uint32_t r) { return r < 0x80 ? 1 : r < 0x800 ? 2 : r < 0x10000 ? 3 : 4; }

// Where one input may be cut into `shards` slices that are encoded apart (rsn_huffman_compress_sharded): 0 = cut[0] < cut[1] < ... <
// cut[S] = n, every cut on a rune START of Go's decoding of the whole string (huffman.go:309), so that no UTF-8 sequence is split and
// the slices' runes are the input's; fewer slices than asked where the input is short (64 bytes a slice at least).
void huff_slice_cuts(const uint8_t *in, size_t n, int shards, std::vector<size_t> &cut);

// buildTree (huffman.go:58-103) incl. Go container/heap order.  syms: any order,
// destroyed.  Returns false (with msg) for an empty table.
bool build_tree(std::vector<HuffSym> &syms, HuffTree &t, std::string &msg);
// printCodes (huffman.go:110-127).  Fails if a code exceeds 64 bits.  want_dfs: also the leaves in the order printCodes visits them
// (c.dfs: what rsn_huffman_table / rsn_huffman_plan hand out); without it the codes come from one pass over the internal nodes, parents
// before children -- no stack, a third of the time on the 3 * 10^5 symbols of a rune alphabet.
bool assign_codes(const HuffTree &t, HuffCodes &c, std::string &msg, bool want_dfs = true);
// Header in this library's canonical order (ascending rune, '\\' never last), huffman.go:312-318.
void emit_header(const std::vector<HuffSym> &by_rune_asc, std::string &out);
// decodeTree's header scan (huffman.go:196-227).  Returns symbols ascending by rune.
bool parse_header(const uint8_t *h, size_t n, std::vector<HuffSym> &syms, std::string &msg);

}  // namespace rsn
