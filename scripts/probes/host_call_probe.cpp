// the host-buffer calls from a plain C++ process (no Python, no torch): <MiB> <kind>
//   kind 0: 2a-like bytes (128 equiprobable symbols), Huffman; 1: skewed symbols, Huffman; 2: Zipf text over 4096 words, LZSS (window 4096);
//   @<file>: the first <MiB> of that file, LZSS; h@<file>: the same, Huffman (bench.py hands config 4's text and config 2a's bytes over this way)
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <algorithm>
#include <cmath>
#include <vector>
#include "../../include/rsn.h"
static double now() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main(int argc, char **argv) {
    const size_t n = (size_t)(argc > 1 ? atoi(argv[1]) : 1024) << 20;
    const char *file = argc > 2 && argv[2][0] == '@' ? argv[2] + 1 : argc > 2 && argv[2][0] == 'h' && argv[2][1] == '@' ? argv[2] + 2 : nullptr;
    const int skew = file ? (argv[2][0] == 'h' ? 0 : 2) : argc > 2 ? atoi(argv[2]) : 0;
    uint8_t *src = (uint8_t *)malloc(n);
    unsigned long long z = 88172645463325252ull;
    auto next = [&] { z ^= z << 13; z ^= z >> 7; z ^= z << 17; return z; };
    if (file) {
        FILE *f = fopen(file, "rb");
        if (!f || fread(src, 1, n, f) != n) { fprintf(stderr, "%s: fewer than %zu bytes\n", file, n); return 1; }
        fclose(f);
    } else if (skew == 2) {
        std::vector<std::string> words(4096);
        for (auto &w : words) { const int len = 2 + (int)(next() % 8); for (int k = 0; k < len; k++) w.push_back((char)('a' + next() % 26)); w.push_back(' '); }
        std::vector<double> cdf(4096); double acc = 0;
        for (int k = 0; k < 4096; k++) { acc += std::pow(k + 1.0, -1.3); cdf[k] = acc; }
        for (size_t i = 0; i < n;) {
            const double u = (double)(next() >> 11) / 9007199254740992.0 * acc;
            const std::string &w = words[std::lower_bound(cdf.begin(), cdf.end(), u) - cdf.begin()];
            for (size_t k = 0; k < w.size() && i < n; k++) src[i++] = (uint8_t)w[k];
        }
    } else
    for (size_t i = 0; i < n; i++) { next(); src[i] = skew ? (uint8_t)(32 + __builtin_ctzll(z | (1ull << 40)) * 2 + (z >> 60 & 1)) : (uint8_t)(z >> 33 & 127); }
    uint8_t *c = nullptr, *d = nullptr; size_t cn = 0, dn = 0;
    auto compress = [&](uint8_t **o, size_t *on) { return skew == 2 ? rsn_lzss_compress(src, n, 4096, o, on) : rsn_huffman_compress(src, n, o, on); };
    auto decompress = [&](uint8_t **o, size_t *on) { return skew == 2 ? rsn_lzss_decompress(c, cn, o, on) : rsn_huffman_decompress(c, cn, o, on); };
    double best_c = 1e30, best_d = 1e30; int same = 1;
    for (int rep = 0; rep < 3; rep++) {
        if (c) rsn_free(c);
        const double t = now();
        if (compress(&c, &cn)) { fprintf(stderr, "compress: %s\n", rsn_last_error()); return 1; }
        const double ms = now() - t;
        printf("compress %zu MiB -> %zu B: %.1f ms\n", n >> 20, cn, ms);
        if (ms < best_c) best_c = ms;
    }
    for (int rep = 0; rep < 4; rep++) {
        const double t = now();
        if (decompress(&d, &dn)) { fprintf(stderr, "decompress: %s\n", rsn_last_error()); return 1; }
        const double ms = now() - t;
        const int ok = dn == n && memcmp(d, src, n) == 0;
        printf("decompress -> %zu B: %.1f ms, same=%d\n", dn, ms, ok);
        same = same && ok;
        if (ms < best_d) best_d = ms;
        rsn_free(d);
    }
    printf("RESULT {\"bytes\": %zu, \"compressed\": %zu, \"compress_ms\": %.2f, \"decompress_ms\": %.2f, \"lossless\": %s}\n", n, cn, best_c, best_d, same ? "true" : "false");
    return 0;
}
