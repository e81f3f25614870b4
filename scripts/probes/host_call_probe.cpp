// the host-buffer Huffman calls from a plain C++ process (no Python, no torch): 1 GiB of 2a-like bytes, compress once, decompress four times
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include "../../include/rsn.h"
static double now() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main(int argc, char **argv) {
    const size_t n = (size_t)(argc > 1 ? atoi(argv[1]) : 1024) << 20;
    const int skew = argc > 2 ? atoi(argv[2]) : 0;
    uint8_t *src = (uint8_t *)malloc(n);
    unsigned long long z = 88172645463325252ull;
    for (size_t i = 0; i < n; i++) { z ^= z << 13; z ^= z >> 7; z ^= z << 17; src[i] = skew ? (uint8_t)(32 + __builtin_ctzll(z | (1ull << 40)) * 2 + (z >> 60 & 1)) : (uint8_t)(z >> 33 & 127); }
    uint8_t *c = nullptr, *d = nullptr; size_t cn = 0, dn = 0;
    double best_c = 1e30, best_d = 1e30; int same = 1;
    for (int rep = 0; rep < 3; rep++) {
        if (c) rsn_free(c);
        const double t = now();
        if (rsn_huffman_compress(src, n, &c, &cn)) { fprintf(stderr, "compress: %s\n", rsn_last_error()); return 1; }
        const double ms = now() - t;
        printf("compress %zu MiB -> %zu B: %.1f ms\n", n >> 20, cn, ms);
        if (ms < best_c) best_c = ms;
    }
    for (int rep = 0; rep < 4; rep++) {
        const double t = now();
        if (rsn_huffman_decompress(c, cn, &d, &dn)) { fprintf(stderr, "decompress: %s\n", rsn_last_error()); return 1; }
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
