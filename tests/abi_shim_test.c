/*
 * abi_shim_test.c -- the cgo overlay's call sequence (go/overlay/compressor/{huffman,lz}/*_rsn.go), replayed in C.
 * The overlay cannot be compiled here (no Go toolchain), so the contract it relies on is compiled and run in this
 * file instead, against the same header and library a cgo build would bind:
 *   rsnCall:  input BORROWED for the call (never modified, may be freed right after), output owned by the library,
 *             COPIED by the caller, then released with rsn_free(); a non-zero return code and rsn_last_error()
 *             (thread-local) carry what the Go side turns into panic();
 *   engine:   eight goroutines = eight OS threads calling concurrently (engine.go:235-244), each through its own
 *             per-thread context, with results that must not depend on the interleaving.
 * Exit code 0 = every check passed.  `abi_shim_test nodev` runs only the part that needs no device (symbols link,
 * the host-only entry points work, device entry points fail cleanly with RSN_ERR_DEVICE, an allocation that cannot be had is a
 * code and not an abort).  `abi_shim_test threadfail` (under LD_PRELOAD=pthread_fail_shim.so, on a GPU): the calls that use helper
 * threads -- the pipelined host calls, the batch, the sharded stream, the side jobs of large alphabets -- while the system refuses
 * every new thread: each returns its bytes by its serial form or a negative code, none ends the process (engine.go:315-328 recovers
 * a panic; nothing recovers std::terminate).
 */
#define _GNU_SOURCE
#include <dlfcn.h>
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "rsn.h"

#define CHECK(c) do { if (!(c)) { fprintf(stderr, "FAILED %s:%d: %s (last error: %s)\n", __FILE__, __LINE__, #c, rsn_last_error()); exit(1); } } while (0)

typedef int (*codec_fn)(const uint8_t *, size_t, uint8_t **, size_t *);
static int lz_c(const uint8_t *p, size_t n, uint8_t **o, size_t *on) { return rsn_lzss_compress(p, n, RSN_LZSS_DEFAULT_WINDOW, o, on); }
static int lz_legacy(const uint8_t *p, size_t n, uint8_t **o, size_t *on) { return rsn_lzss_compress_legacy(p, n, RSN_LZSS_DEFAULT_WINDOW, o, on); }

/* rsnCall of the overlay: returns a malloc'ed COPY (the Go slice), or NULL with *rc set (the panic) */
static uint8_t *rsn_call(codec_fn f, const uint8_t *in, size_t n, size_t *out_n, int *rc) {
    uint8_t *borrowed = malloc(n ? n : 1);                  /* the pinned Go slice */
    memcpy(borrowed, in, n);
    uint8_t *out = NULL; size_t on = 0;
    *rc = f(n ? borrowed : NULL, n, &out, &on);
    if (*rc == 0) CHECK(memcmp(borrowed, in, n) == 0);      /* input never modified */
    memset(borrowed, 0xA5, n); free(borrowed);              /* the borrow ends with the call */
    if (*rc != 0) { CHECK(out == NULL && on == 0); return NULL; }
    uint8_t *copy = malloc(on ? on : 1);
    memcpy(copy, out, on);                                  /* copy(res, unsafe.Slice(out, n)) */
    rsn_free(out);                                          /* defer C.rsn_free(out) */
    *out_n = on;
    return copy;
}

static uint8_t *make_text(size_t n, unsigned seed) {
    static const char *w[] = {"the", "quick", "brown", "fox", "jumps", "over", "lazy", "dog", "raisin", "huffman", "lzss", "window"};
    uint8_t *b = malloc(n + 16); size_t k = 0;
    while (k < n) { seed = seed * 1103515245u + 12345u; const char *s = w[(seed >> 16) % 12]; size_t l = strlen(s); memcpy(b + k, s, l); k += l; b[k++] = ' '; }
    return b;
}

typedef struct { int id; int rounds; int ok; } job_t;
static void *worker(void *p) {                               /* one goroutine of engine.BenchmarkSuite: compress, decompress, compare */
    job_t *j = p;
    const size_t n = 200000 + 1000 * (size_t)j->id;
    uint8_t *src = make_text(n, 17u * (unsigned)j->id + 1);
    for (int r = 0; r < j->rounds; r++) {
        int rc; size_t cn, dn, c2n, d2n;
        uint8_t *c = rsn_call(j->id & 1 ? lz_c : rsn_huffman_compress, src, n, &cn, &rc); CHECK(rc == 0);
        uint8_t *d = rsn_call(j->id & 1 ? rsn_lzss_decompress : rsn_huffman_decompress, c, cn, &dn, &rc); CHECK(rc == 0);
        CHECK(dn == n && memcmp(d, src, n) == 0);
        /* layered, as `-algorithm=lzss,huffman` (engine.go:443-479) */
        uint8_t *l1 = rsn_call(lz_c, src, n, &cn, &rc); CHECK(rc == 0);
        uint8_t *l2 = rsn_call(rsn_huffman_compress, l1, cn, &c2n, &rc); CHECK(rc == 0);
        uint8_t *b1 = rsn_call(rsn_huffman_decompress, l2, c2n, &d2n, &rc); CHECK(rc == 0 && d2n == cn && memcmp(b1, l1, cn) == 0);
        uint8_t *b0 = rsn_call(rsn_lzss_decompress, b1, d2n, &dn, &rc); CHECK(rc == 0 && dn == n && memcmp(b0, src, n) == 0);
        /* a failing call on THIS thread: its message is this thread's own */
        size_t xn; uint8_t *x = rsn_call(rsn_huffman_decompress, (const uint8_t *)"no separator here", 17, &xn, &rc);
        CHECK(x == NULL && rc == RSN_ERR_FORMAT && strlen(rsn_last_error()) > 0);
        free(c); free(d); free(l1); free(l2); free(b1); free(b0);
    }
    free(src);
    j->ok = 1;
    return NULL;
}

/* every call that asks the library for helper threads, once; `refuse` = the system hands out no thread meanwhile */
static void helper_thread_calls(void (*fail_threads)(int), int refuse, uint8_t **keep, size_t *keep_n) {
    int rc; size_t n;
    const size_t big = (size_t)160 << 20;
    uint8_t *text = make_text(big, 99u);
    /* a stream large enough for the pipelined decode (>= 32 MiB), produced with threads available: 48 MiB of 7-bit noise -> 42 MiB */
    const size_t nr = (size_t)48 << 20;
    uint8_t *noise = malloc(nr);
    { unsigned long long x = 0x9E3779B97F4A7C15ull; for (size_t i = 0; i < nr; i++) { x = x * 6364136223846793005ull + 1442695040888963407ull; noise[i] = (uint8_t)(x >> 57); } }
    size_t cn; uint8_t *c = rsn_call(rsn_huffman_compress, noise, nr, &cn, &rc); CHECK(rc == 0 && cn > ((size_t)32 << 20));
    fail_threads(refuse);
    uint8_t *d = rsn_call(rsn_huffman_decompress, c, cn, &n, &rc);               /* piped_call -> no helper -> the serial call */
    CHECK(rc == 0 && n == nr && memcmp(d, noise, n) == 0); free(d); free(noise);
    uint8_t *l = rsn_call(lz_c, text, big, &n, &rc);                               /* the sectioned encode (>= 128 MiB) -> the serial call */
    CHECK(rc == 0 && n > 0 && n < big);
    if (keep[0]) CHECK(n == keep_n[0] && memcmp(l, keep[0], n) == 0); else { keep[0] = l; keep_n[0] = n; l = NULL; }
    free(l);
    enum { NC = 4 };
    const uint8_t *ins[NC]; size_t lens[NC]; uint8_t *outs[NC]; size_t out_lens[NC];
    for (int i = 0; i < NC; i++) { ins[i] = text + ((size_t)i << 20); lens[i] = 700000 + 1111 * (size_t)i; }
    CHECK(rsn_huffman_compress_batch(NC, ins, lens, outs, out_lens) == 0);       /* no pipeline: chunk after chunk */
    for (int i = 0; i < NC; i++) {
        size_t sn; uint8_t *single = rsn_call(rsn_huffman_compress, ins[i], lens[i], &sn, &rc);
        CHECK(rc == 0 && sn == out_lens[i] && memcmp(single, outs[i], sn) == 0);
        free(single); rsn_free(outs[i]);
    }
    /* one stream from four slices: every slice but the caller's needs a thread -- refused: a code and a message, *out NULL (helpers that
     * are idle from an earlier round need no new thread: then the stream comes out, and is the single call's) */
    uint8_t *so = (uint8_t *)1; size_t son = 1;
    rc = rsn_huffman_compress_sharded(text, (size_t)8 << 20, 4, &so, &son);
    if (rc != 0) CHECK(refuse && rc == RSN_ERR_NOMEM && so == NULL && son == 0 && strstr(rsn_last_error(), "helper"));
    else {
        size_t sn; uint8_t *single = rsn_call(rsn_huffman_compress, text, (size_t)8 << 20, &sn, &rc);
        CHECK(rc == 0 && so != NULL && sn == son && memcmp(single, so, sn) == 0);
        free(single); rsn_free(so);
    }
    /* 20 000 distinct runes: the header's text and the decoder's tables are side jobs -- on the caller's thread when refused */
    {
        enum { NR = 20000 };
        uint8_t *wide = malloc(6 * NR);
        for (int i = 0; i < 2 * NR; i++) { const unsigned r = 0x800u + (unsigned)(i % NR); wide[3 * i] = 0xE0 | (r >> 12); wide[3 * i + 1] = 0x80 | ((r >> 6) & 63); wide[3 * i + 2] = 0x80 | (r & 63); }
        size_t wn; uint8_t *w = rsn_call(rsn_huffman_compress, wide, 6 * NR, &wn, &rc); CHECK(rc == 0);
        if (keep[1]) CHECK(wn == keep_n[1] && memcmp(w, keep[1], wn) == 0);
        size_t bn; uint8_t *b = rsn_call(rsn_huffman_decompress, w, wn, &bn, &rc); CHECK(rc == 0 && bn == 6 * NR && memcmp(b, wide, bn) == 0);
        if (!keep[1]) { keep[1] = w; keep_n[1] = wn; w = NULL; }
        free(w); free(b); free(wide);
    }
    fail_threads(0);
    free(c); free(text);
}

int main(int argc, char **argv) {
    const int nodev = argc > 1 && strcmp(argv[1], "nodev") == 0;
    const int threadfail = argc > 1 && strcmp(argv[1], "threadfail") == 0;
    int rc; size_t n;
    CHECK(strlen(rsn_version()) > 0);
    /* host-only entry point: README.md:165's 21-byte answer of the legacy encoder, no device needed */
    uint8_t *g = rsn_call(lz_legacy, (const uint8_t *)"abcabcabcabcabcabcabcabc\n", 25, &n, &rc);
    CHECK(rc == 0 && n == 21 && memcmp(g, "abcabca<6,6>b<12,10>\n", 21) == 0);
    free(g);
    /* ... and its limit: the shim's lz.Compress panics ("librsn: ... above the ...-byte bound") where the reference would have run its
     * O(n * W) loop for hours -- 1 MiB + 1 with an unbounded window (window 0) is refused before any work, with this thread's message
     * (RSN_LEGACY_NO_LIMIT=1 lifts the bound: tests/test_abi_and_host.py) */
    {
        const size_t big = ((size_t)1 << 20) + 1;
        uint8_t *zeros = calloc(big, 1), *lo = NULL; size_t lon = 0;
        CHECK(rsn_lzss_compress_legacy(zeros, big, 0, &lo, &lon) == RSN_ERR_LIMIT && lo == NULL && lon == 0 && strstr(rsn_last_error(), "legacy"));
        if (!getenv("RSN_LEGACY_NO_LIMIT")) {
            uint8_t *x = rsn_call(lz_legacy, zeros, 100, &n, &rc); CHECK(x != NULL && rc == 0); free(x);   /* the next call on the thread is unaffected */
        }
        free(zeros);
    }
    /* an allocation the host cannot make is RSN_ERR_NOMEM with this thread's message, not an exception through the C boundary
     * (2^60 symbols: the table's vector refuses before it reads a single one) */
    CHECK(rsn_huffman_plan(NULL, NULL, (size_t)1 << 60, NULL, NULL, NULL, NULL, 0, NULL) == RSN_ERR_NOMEM && strlen(rsn_last_error()) > 0);
    if (threadfail) {
        void (*fail_threads)(int) = (void (*)(int))dlsym(RTLD_DEFAULT, "rsn_test_fail_threads");
        long (*refused)(void) = (long (*)(void))dlsym(RTLD_DEFAULT, "rsn_test_threads_refused");
        CHECK(fail_threads && refused);                      /* LD_PRELOAD=pthread_fail_shim.so */
        uint8_t *keep[2] = {NULL, NULL}; size_t keep_n[2] = {0, 0};
        uint8_t *w = rsn_call(rsn_huffman_compress, (const uint8_t *)"warm", 4, &n, &rc); CHECK(rc == 0); free(w);   /* the device is up */
        helper_thread_calls(fail_threads, 1, keep, keep_n);  /* no thread to be had */
        const long r1 = refused();
        CHECK(r1 >= 5);                                      /* (the library did ask) */
        helper_thread_calls(fail_threads, 0, keep, keep_n);  /* helpers available: the same bytes */
        helper_thread_calls(fail_threads, 1, keep, keep_n);  /* refused again: the helpers that exist by now serve where they are idle */
        free(keep[0]); free(keep[1]);
        rsn_trim();
        printf("abi shim (threads refused %ld times): ok\n", refused());
        return 0;
    }
    if (nodev) {
        if (rsn_device_count() <= 0) {                       /* no GPU: every codec call fails loudly, none computes on the CPU */
            uint8_t *x = rsn_call(rsn_huffman_compress, (const uint8_t *)"abc", 3, &n, &rc);
            CHECK(x == NULL && rc == RSN_ERR_DEVICE && strstr(rsn_last_error(), "no CPU fallback"));
            x = rsn_call(lz_c, (const uint8_t *)"abc", 3, &n, &rc);
            CHECK(x == NULL && rc == RSN_ERR_DEVICE);
        }
        printf("abi shim (no device): ok\n");
        return 0;
    }
    /* known answers through the shim sequence */
    uint8_t *c = rsn_call(rsn_huffman_compress, (const uint8_t *)"ab", 2, &n, &rc);
    CHECK(rc == 0 && n == 10 && memcmp(c, "1|a1|b\\\n\x06\x01", 10) == 0); free(c);
    c = rsn_call(lz_c, (const uint8_t *)"abcabcabcabcabcabcabcabc\n", 25, &n, &rc);
    CHECK(rc == 0 && n == 19 && memcmp(c, "abcabc<6,6><12,12>\n", 19) == 0); free(c);
    c = rsn_call(lz_c, NULL, 0, &n, &rc); CHECK(rc == 0 && n == 0); free(c);          /* CompressAsync(empty) == empty */
    c = rsn_call(rsn_huffman_compress, NULL, 0, &n, &rc);                              /* reference panics in heap.Pop (huffman.go:102) */
    CHECK(c == NULL && rc == RSN_ERR_EMPTY && strstr(rsn_last_error(), "empty"));
    c = rsn_call(rsn_lzss_decompress, (const uint8_t *)"ab<9,2>", 7, &n, &rc);        /* pointer past the decoded data */
    CHECK(c == NULL && rc == RSN_ERR_FORMAT);
    /* rsn_last_error() is per thread and survives until the next call on this thread */
    CHECK(strlen(rsn_last_error()) > 0);
    /* eight concurrent callers */
    enum { T = 8 };
    pthread_t th[T]; job_t jobs[T];
    for (int i = 0; i < T; i++) { jobs[i] = (job_t){i, 3, 0}; CHECK(pthread_create(&th[i], NULL, worker, &jobs[i]) == 0); }
    for (int i = 0; i < T; i++) { pthread_join(th[i], NULL); CHECK(jobs[i].ok); }
    /* batch form: one .rsn segment per chunk, each identical to the single call */
    enum { NC = 5 };
    const uint8_t *ins[NC]; size_t lens[NC]; uint8_t *outs[NC]; size_t out_lens[NC]; uint8_t *bufs[NC];
    for (int i = 0; i < NC; i++) { lens[i] = 300000 + 77777 * (size_t)i; bufs[i] = make_text(lens[i], 1000u + (unsigned)i); ins[i] = bufs[i]; }
    CHECK(rsn_huffman_compress_batch(NC, ins, lens, outs, out_lens) == 0);
    for (int i = 0; i < NC; i++) {
        size_t sn; uint8_t *single = rsn_call(rsn_huffman_compress, bufs[i], lens[i], &sn, &rc);
        CHECK(rc == 0 && sn == out_lens[i] && memcmp(single, outs[i], sn) == 0);
        free(single); rsn_free(outs[i]); free(bufs[i]);
    }
    /* a chunk whose segment outgrows the pipeline's output slot (50 000 distinct runes: the header alone is > 1 MB for
     * 150 KB of input) between two ordinary ones, and a batch with an empty chunk (refused as the single call refuses it) */
    {
        enum { NR = 50000 };
        uint8_t *wide = malloc(3 * NR);
        for (int i = 0; i < NR; i++) { const unsigned r = 0x800u + (unsigned)i; wide[3 * i] = 0xE0 | (r >> 12); wide[3 * i + 1] = 0x80 | ((r >> 6) & 63); wide[3 * i + 2] = 0x80 | (r & 63); }
        uint8_t *t0 = make_text(400000, 7u), *t1 = make_text(500000, 8u);
        const uint8_t *in3[4] = {t0, wide, t1, t0}; size_t len3[4] = {400000, 3 * NR, 500000, 123457}; uint8_t *out3[4]; size_t on3[4];
        CHECK(rsn_huffman_compress_batch(4, in3, len3, out3, on3) == 0);
        CHECK(on3[1] > 3 * NR + 3 * NR / 8 + 65536);
        for (int i = 0; i < 4; i++) {
            size_t sn; uint8_t *single = rsn_call(rsn_huffman_compress, in3[i], len3[i], &sn, &rc);
            CHECK(rc == 0 && sn == on3[i] && memcmp(single, out3[i], sn) == 0);
            free(single); rsn_free(out3[i]);
        }
        len3[2] = 0;
        CHECK(rsn_huffman_compress_batch(4, in3, len3, out3, on3) == RSN_ERR_EMPTY);
        for (int i = 0; i < 4; i++) CHECK(out3[i] == NULL && on3[i] == 0);
        free(wide); free(t0); free(t1);
    }
    rsn_trim();
    printf("abi shim: ok\n");
    return 0;
}
