#include "huff_host.h"
#include "lzss_legacy.h"
#include <cstdio>
#include <algorithm>
#include <random>
#include <string>
using namespace rsn;
int main() {
    std::mt19937_64 rng(7);
    for (int it = 0; it < 300; it++) {
        std::vector<HuffSym> syms;
        const int k = 1 + (int)(rng() % (it % 7 == 0 ? 50000 : 300));
        uint32_t r = 0;
        for (int i = 0; i < k; i++) {
            r += 1 + (uint32_t)(rng() % (it % 3 ? 5 : 4000));
            if (r >= 0x110000) break;
            if (r >= 0xD800 && r < 0xE000) r = 0xE000;
            uint64_t f = (it % 5 == 0) ? (rng() >> (rng() % 64)) | 1 : 1 + rng() % 1000;
            syms.push_back({r, f});
        }
        bool has = false; for (auto &x : syms) has = has || x.rune == 0x5C;
        if (it % 11 == 0 && !has) syms.push_back({0x5C, 3}), std::sort(syms.begin(), syms.end(), [](const HuffSym &a, const HuffSym &b) { return a.rune < b.rune; });
        std::string hdr;
        emit_header(syms, hdr);
        std::vector<HuffSym> back; std::string msg;
        if (!parse_header((const uint8_t *)hdr.data(), hdr.size(), back, msg)) { printf("parse failed: %s\n", msg.c_str()); return 1; }
        if (back.size() != syms.size()) { printf("size mismatch %zu %zu (it %d)\n", back.size(), syms.size(), it); return 1; }
        HuffTree t; HuffCodes c;
        std::vector<HuffSym> s2 = syms;
        if (!build_tree(s2, t, msg)) { printf("tree failed\n"); return 1; }
        assign_codes(t, c, msg);
    }
    for (int it = 0; it < 200; it++) {
        std::string in;
        const int n = (int)(rng() % 5000);
        for (int i = 0; i < n; i++) in.push_back((char)("ab<\\\xff c"[rng() % 7]));
        std::string out;
        lzss_compress_legacy_host((const uint8_t *)in.data(), in.size(), (long long)(rng() % 3 ? 4096 : 17), out);
    }
    for (int it = 0; it < 400; it++) {                                    // slice cuts of the sharded Huffman stream: rune starts, strictly increasing
        std::string in;
        const int n = 1 + (int)(rng() % 3000);
        static const char *pieces[] = {"a", "\xC3\xA9", "\xE2\x82\xAC", "\xF0\x9D\x84\x9E", "\x80", "\xE2\x82", "\xF0\x9D", "\xFF", "\xC3", "\xBF\xBF\xBF\xBF"};
        while ((int)in.size() < n) in += pieces[rng() % 10];
        std::vector<size_t> cut;
        const int G = 1 + (int)(rng() % 40);
        huff_slice_cuts((const uint8_t *)in.data(), in.size(), G, cut);
        if (cut.front() != 0 || cut.back() != in.size() || cut.size() > (size_t)G + 1) { printf("cuts: bad ends\n"); return 1; }
        std::vector<char> start(in.size() + 1, 0);                          // Go's decoding of the whole string, rune by rune
        for (size_t i = 0; i < in.size();) { start[i] = 1; int sz = 1; (void)go_decode_rune((const uint8_t *)in.data() + i, in.size() - i, &sz); i += (size_t)sz; }
        start[in.size()] = 1;
        for (size_t k = 0; k + 1 < cut.size(); k++) if (cut[k] >= cut[k + 1] || !start[cut[k]]) { printf("cuts: %zu is not a rune start (it %d)\n", cut[k], it); return 1; }
    }
    printf("host sanitizer run ok\n");
    return 0;
}
