/*
 * cpu_baseline.c -- the CPU restatement run on ALL host cores: the
 * number bench.py prints next to every GPU number (`cpu_baseline`, kind "port": the reference is Go
 * and cannot run here).
 *
 * TEST / MEASUREMENT INFRASTRUCTURE, like the rest of oracle/: never linked, loaded or called by the
 * product (raisin_amd/).  It shares the oracle's functions (same semantics, same bytes out -- pinned by
 * tests/test_oracle.py) and adds only the threading a Go programmer would add with goroutines:
 *   Huffman encode   per-thread histograms of byte ranges -> one tree -> every thread packs its range
 *                    at its exact bit offset (ranges meet inside a byte: atomic OR on the two edge bytes)
 *   Huffman decode   the stream has no block index, so threads decode their slice from a guessed entry,
 *                    a serial pass corrects every slice's entry from its predecessor's exit (a prefix
 *                    code re-synchronises within a few codewords), then the slices decode again for real
 *   LZSS encode      the reference's own shape (lzss.go:117-130): a Reference for EVERY position,
 *                    position ranges handed to the threads (grain 1 = one task per position, the
 *                    goroutine-per-byte form), then the serial compaction of lzss.go:134-151
 *   LZSS decode      serial, as the reference (lzss.go:323-364)
 *   LZSS check       "is this candidate stream the oracle's output?" at sizes where producing that output takes minutes:
 *                    the candidate is cut right after tokens (a position right after a token lies on the greedy chain if
 *                    the token's start does), every segment is re-encoded by the oracle's own loop from its start, and must
 *                    reproduce the candidate's bytes AND land exactly on the next segment's start -- by induction from
 *                    position 0 the candidate then equals rsn_oracle_lzss_compress()'s output byte for byte.  Only chain
 *                    positions are evaluated (a fifth of all on text) and segments are independent: 1 GiB in seconds.
 * Inputs with bytes >= 0x80 (rune != byte) fall back to the single-threaded oracle.
 */
#include <pthread.h>
#include <stdatomic.h>
#include <stdlib.h>
#include <string.h>

#include "rsn_oracle.h"

int rsn_oracle_huffman_plan_bytes(const uint64_t hist[256], uint64_t code[256], uint8_t len[256], uint8_t **hdr, size_t *hdr_n);
int rsn_oracle_huffman_parse(const uint8_t *in, size_t n, size_t *payload_off, unsigned *pad, int32_t **left, int32_t **right,
                             uint32_t **rune, int32_t *root, uint32_t *n_nodes);
int rsn_oracle_lzss_matches_range(const uint8_t *esc, size_t e, int64_t window, size_t lo, size_t hi, uint32_t *off, uint32_t *size);
int rsn_oracle_lzss_compact(const uint8_t *esc, size_t e, const uint32_t *off, const uint32_t *size, uint8_t **out, size_t *out_n);
int rsn_oracle_lzss_compress_range(const uint8_t *esc, size_t e, int64_t window, size_t start, size_t stop, uint8_t **out, size_t *out_n, size_t *landed);

typedef void (*job_fn)(void *ctx, int t, int nt);
typedef struct { job_fn fn; void *ctx; int t, nt; } job_t;
static void *job_main(void *p) { job_t *j = p; j->fn(j->ctx, j->t, j->nt); return NULL; }
static void run_threads(job_fn fn, void *ctx, int nt) {
    pthread_t *th = malloc((size_t)nt * sizeof *th);
    job_t *jb = malloc((size_t)nt * sizeof *jb);
    for (int t = 0; t < nt; t++) { jb[t] = (job_t){fn, ctx, t, nt}; if (t) pthread_create(&th[t], NULL, job_main, &jb[t]); }
    job_main(&jb[0]);
    for (int t = 1; t < nt; t++) pthread_join(th[t], NULL);
    free(th); free(jb);
}

/* ------------------------------------------------------------------ Huffman encode */
typedef struct {
    const uint8_t *in; size_t n;
    uint64_t (*hist)[256];
    const uint64_t *code; const uint8_t *len;
    uint64_t *bit_off; uint8_t *payload;
    _Atomic int high;
} henc_t;
static void slice(size_t n, int t, int nt, size_t *lo, size_t *hi) { *lo = n / (size_t)nt * (size_t)t; *hi = t == nt - 1 ? n : n / (size_t)nt * (size_t)(t + 1); }

static void henc_hist(void *c, int t, int nt) {
    henc_t *h = c; size_t lo, hi; slice(h->n, t, nt, &lo, &hi);
    uint64_t *H = h->hist[t]; int high = 0;
    for (size_t i = lo; i < hi; i++) { H[h->in[i]]++; high |= h->in[i] >> 7; }
    if (high) atomic_store(&h->high, 1);
}
static void henc_emit(void *c, int t, int nt) {
    henc_t *h = c; size_t lo, hi; slice(h->n, t, nt, &lo, &hi);
    uint64_t bp = h->bit_off[t], first = bp >> 3, last = (h->bit_off[t + 1] - (h->bit_off[t + 1] > bp)) >> 3;
    uint8_t *P = h->payload;
    uint64_t acc = 0; unsigned fill = (unsigned)(bp & 7);          /* bits already in the current byte (owned by the previous slice) */
    uint64_t byte = first;
    for (size_t i = lo; i < hi; i++) {
        const uint64_t cw = h->code[h->in[i]]; unsigned l = h->len[h->in[i]];
        while (l) {                                                 /* MSB-first, huffman.go:239 */
            unsigned take = 8 - fill; if (take > l) take = l;
            acc = (acc << take) | ((cw >> (l - take)) & ((1u << take) - 1));
            fill += take; l -= take;
            if (fill == 8) {
                if (byte == first || byte == last) atomic_fetch_or((_Atomic uint8_t *)&P[byte], (uint8_t)acc); else P[byte] = (uint8_t)acc;
                byte++; acc = 0; fill = 0;
            }
        }
    }
    if (fill) { uint8_t v = (uint8_t)(acc << (8 - fill)); atomic_fetch_or((_Atomic uint8_t *)&P[byte], v); }
}

int rsn_baseline_huffman_compress_mt(const uint8_t *in, size_t n, int threads, uint8_t **out, size_t *out_n) {
    if (threads < 1) threads = 1;
    if (n < (size_t)threads * 64) threads = 1;
    henc_t h; memset(&h, 0, sizeof h);
    h.in = in; h.n = n; h.hist = calloc((size_t)threads, sizeof *h.hist);
    run_threads(henc_hist, &h, threads);
    if (atomic_load(&h.high) || n == 0) { free(h.hist); return rsn_oracle_huffman_compress(in, n, out, out_n); }   /* runes != bytes */
    uint64_t tot[256] = {0};
    for (int t = 0; t < threads; t++) for (int s = 0; s < 256; s++) tot[s] += h.hist[t][s];
    uint64_t code[256]; uint8_t len[256]; uint8_t *hdr; size_t hn;
    if (rsn_oracle_huffman_plan_bytes(tot, code, len, &hdr, &hn)) { free(h.hist); return RSN_ORACLE_ERR; }
    uint64_t total_bits = 0;
    for (int s = 0; s < 256; s++) total_bits += tot[s] * len[s];
    const unsigned pad = (unsigned)((8 - total_bits % 8) % 8);      /* huffman.go:245-249 */
    const size_t pay = (size_t)((total_bits + pad) / 8);
    uint8_t *o = calloc(hn + 3 + pay + 1, 1);
    memcpy(o, hdr, hn); o[hn] = 0x5C; o[hn + 1] = 0x0A; o[hn + 2] = (uint8_t)pad;
    free(hdr);
    h.bit_off = malloc(((size_t)threads + 1) * sizeof *h.bit_off);
    uint64_t run = pad;
    for (int t = 0; t < threads; t++) { h.bit_off[t] = run; for (int s = 0; s < 256; s++) run += h.hist[t][s] * len[s]; }
    h.bit_off[threads] = run;
    h.code = code; h.len = len; h.payload = o + hn + 3;
    run_threads(henc_emit, &h, threads);
    free(h.hist); free(h.bit_off);
    *out = o; *out_n = hn + 3 + pay;
    return RSN_ORACLE_OK;
}

/* ------------------------------------------------------------------ Huffman decode */
#define SYNC_MAX 256
typedef struct {
    const int32_t *left, *right; const uint32_t *rune; int32_t root;
    const uint8_t *P; uint64_t pad, max;                            /* data bit i = payload bit pad+i, i < max */
    uint64_t *lo, *entry, *exit_, *nsym, *out_off;
    uint64_t (*marks)[SYNC_MAX]; uint32_t *n_marks;                 /* first codeword starts of the speculative parse */
    uint8_t *out; int write; _Atomic int bad;
} hdec_t;
static inline int bit_at(const hdec_t *d, uint64_t i) { const uint64_t bp = d->pad + i; return (d->P[bp >> 3] >> (7 - (bp & 7))) & 1; }
/* decode codewords that START in [from, hi); returns the first start >= hi (or max) */
static uint64_t walk(const hdec_t *d, uint64_t from, uint64_t hi, uint64_t *count, uint64_t *marks, uint32_t *n_marks, uint8_t *out) {
    uint64_t i = from, c = 0;
    while (i < hi && i < d->max) {
        if (marks && *n_marks < SYNC_MAX) marks[(*n_marks)++] = i;
        int32_t node = d->root;
        while (d->left[node] >= 0) { if (i >= d->max) { *count = c; return (uint64_t)-1; } node = bit_at(d, i) ? d->right[node] : d->left[node]; i++; }
        if (out) out[c] = (uint8_t)d->rune[node];
        c++;
    }
    *count = c;
    return i;
}
static void hdec_pass(void *c, int t, int nt) {
    hdec_t *d = c; (void)nt;
    const uint64_t hi = d->lo[t + 1];
    if (!d->write) { d->n_marks[t] = 0; d->exit_[t] = walk(d, d->entry[t], hi, &d->nsym[t], d->marks[t], &d->n_marks[t], NULL); }
    else { uint64_t k; if (walk(d, d->entry[t], hi, &k, NULL, NULL, d->out + d->out_off[t]) == (uint64_t)-1 || k != d->nsym[t]) atomic_store(&d->bad, 1); }
}

int rsn_baseline_huffman_decompress_mt(const uint8_t *in, size_t n, int threads, uint8_t **out, size_t *out_n) {
    size_t poff; unsigned pad; int32_t *left, *right, root; uint32_t *rune, nn;
    if (rsn_oracle_huffman_parse(in, n, &poff, &pad, &left, &right, &rune, &root, &nn)) return RSN_ORACLE_ERR;
    int simple = left[root] >= 0 && poff <= n;
    for (uint32_t k = 0; k < nn && simple; k++) if (left[k] < 0 && rune[k] >= 0x80) simple = 0;
    const uint64_t nbits = poff <= n ? (uint64_t)(n - poff) * 8 : 0;
    if (!simple || pad > nbits || threads < 2 || nbits < (uint64_t)threads * 4096) {
        free(left); free(right); free(rune);
        return rsn_oracle_huffman_decompress(in, n, 0, out, out_n);
    }
    hdec_t d; memset(&d, 0, sizeof d);
    d.left = left; d.right = right; d.rune = rune; d.root = root; d.P = in + poff; d.pad = pad; d.max = nbits - pad;
    const int T = threads;
    d.lo = malloc((size_t)(T + 1) * 8); d.entry = malloc((size_t)T * 8); d.exit_ = malloc((size_t)T * 8); d.nsym = malloc((size_t)T * 8);
    d.out_off = malloc((size_t)(T + 1) * 8); d.marks = malloc((size_t)T * sizeof *d.marks); d.n_marks = malloc((size_t)T * 4);
    for (int t = 0; t <= T; t++) d.lo[t] = t == T ? d.max : d.max / (uint64_t)T * (uint64_t)t;
    for (int t = 0; t < T; t++) d.entry[t] = d.lo[t];
    run_threads(hdec_pass, &d, T);                                   /* speculative: every slice from its own first bit */
    int rc = RSN_ORACLE_OK;
    for (int t = 1; t < T && !rc; t++) {                             /* serial fix-up: true entry = predecessor's exit */
        const uint64_t e = d.exit_[t - 1];
        if (e == (uint64_t)-1) { rc = RSN_ORACLE_ERR; break; }
        if (e == d.entry[t]) continue;
        /* walk the true parse from e until it lands on a codeword start of the speculative parse */
        uint64_t i = e, mine = 0; uint32_t k = 0; int synced = 0;
        while (i < d.lo[t + 1] && i < d.max) {
            while (k < d.n_marks[t] && d.marks[t][k] < i) k++;
            if (k < d.n_marks[t] && d.marks[t][k] == i) { synced = 1; break; }
            if (k >= d.n_marks[t] && d.n_marks[t] == SYNC_MAX) break;   /* beyond what was recorded */
            int32_t node = root;
            while (left[node] >= 0) { if (i >= d.max) { rc = RSN_ORACLE_ERR; break; } node = bit_at(&d, i) ? right[node] : left[node]; i++; }
            if (rc) break;
            mine++;
        }
        if (rc) break;
        d.entry[t] = e;
        if (synced) d.nsym[t] = d.nsym[t] - k + mine;                /* k speculative codewords before the meeting point, `mine` true ones */
        else if (i >= d.lo[t + 1] || i >= d.max) { d.nsym[t] = mine; d.exit_[t] = i; }
        else d.exit_[t] = walk(&d, e, d.lo[t + 1], &d.nsym[t], NULL, NULL, NULL);   /* never met within the record: count this slice again */
    }
    if (!rc && d.exit_[T - 1] != d.max) rc = RSN_ORACLE_ERR;          /* payload ends inside a codeword */
    if (!rc) {
        uint64_t run = 0;
        for (int t = 0; t < T; t++) { d.out_off[t] = run; run += d.nsym[t]; }
        d.out = malloc(run ? run : 1); d.write = 1;
        run_threads(hdec_pass, &d, T);
        if (atomic_load(&d.bad)) { free(d.out); rc = RSN_ORACLE_ERR; } else { *out = d.out; *out_n = run; }
    }
    free(d.lo); free(d.entry); free(d.exit_); free(d.nsym); free(d.out_off); free(d.marks); free(d.n_marks);
    free(left); free(right); free(rune);
    if (rc) return rsn_oracle_huffman_decompress(in, n, 0, out, out_n);   /* anything unusual: the plain oracle decides */
    return RSN_ORACLE_OK;
}

/* ------------------------------------------------------------------ LZSS encode */
typedef struct { const uint8_t *esc; size_t e; int64_t window; size_t grain; uint32_t *off, *size; _Atomic size_t next; } lz_t;
static void lz_worker(void *c, int t, int nt) {
    lz_t *z = c; (void)t; (void)nt;
    for (;;) {
        const size_t lo = atomic_fetch_add(&z->next, z->grain);      /* grain 1: one task per position (lzss.go:117-130) */
        if (lo >= z->e) break;
        rsn_oracle_lzss_matches_range(z->esc, z->e, z->window, lo, lo + z->grain, z->off, z->size);
    }
}
int rsn_baseline_lzss_compress_mt(const uint8_t *in, size_t n, int64_t window, int threads, size_t grain, uint8_t **out, size_t *out_n) {
    uint8_t *esc; size_t e;
    if (rsn_oracle_lzss_escape(in, n, &esc, &e)) return RSN_ORACLE_ERR;
    lz_t z; memset(&z, 0, sizeof z);
    z.esc = esc; z.e = e; z.window = window; z.grain = grain ? grain : 4096;
    z.off = malloc((e ? e : 1) * 4); z.size = malloc((e ? e : 1) * 4);
    run_threads(lz_worker, &z, threads < 1 ? 1 : threads);
    const int rc = rsn_oracle_lzss_compact(esc, e, z.off, z.size, out, out_n);
    free(z.off); free(z.size); rsn_oracle_free(esc);
    return rc;
}

/* ------------------------------------------------------------------ LZSS check (see the header comment) */
typedef struct { size_t c0, f0; } cut_t;
typedef struct { const uint8_t *esc; size_t e; int64_t window; const uint8_t *cand; size_t cand_n; const cut_t *cuts; size_t n_cuts; _Atomic size_t next; _Atomic size_t bad; } lzc_t;
static void lzc_worker(void *c, int t, int nt) {
    lzc_t *z = c; (void)t; (void)nt;
    for (;;) {
        const size_t k = atomic_fetch_add(&z->next, 1);
        if (k + 1 >= z->n_cuts) break;
        const cut_t a = z->cuts[k], b = z->cuts[k + 1];
        uint8_t *o; size_t on, landed;
        rsn_oracle_lzss_compress_range(z->esc, z->e, z->window, a.f0, b.f0, &o, &on, &landed);
        const int ok = landed == b.f0 && on == b.c0 - a.c0 && memcmp(o, z->cand + a.c0, on) == 0;
        rsn_oracle_free(o);
        if (!ok) { size_t cur = atomic_load(&z->bad); while (a.c0 < cur && !atomic_compare_exchange_weak(&z->bad, &cur, a.c0)) {} }
    }
}
/* 0 = `cand` is exactly the oracle's CompressAsync output for `in`; 1 = it is not (*bad_at = candidate offset of the first
 * segment that differs, or cand_n if the candidate does not even parse to the escaped length).  seg = escaped bytes per segment. */
int rsn_baseline_lzss_check(const uint8_t *in, size_t n, int64_t window, int threads, size_t seg, const uint8_t *cand, size_t cand_n, size_t *bad_at) {
    uint8_t *esc; size_t e;
    if (rsn_oracle_lzss_escape(in, n, &esc, &e)) return RSN_ORACLE_ERR;
    if (!seg) seg = 1u << 18;
    size_t cap = e / seg + 4, nc = 0;
    cut_t *cuts = malloc(cap * sizeof *cuts);
    cuts[nc++] = (cut_t){0, 0};
    size_t f = 0, i = 0; int okparse = 1;
    while (i < cand_n) {                                             /* lzss.go:323-364's state machine, positions only */
        if (cand[i] != '<') { f++; i++; continue; }
        size_t j = i + 1; unsigned long long off = 0, len = 0; int d1 = 0, d2 = 0;
        while (j < cand_n && cand[j] >= '0' && cand[j] <= '9' && d1 < 19) { off = off * 10 + (cand[j] - '0'); j++; d1++; }
        if (j >= cand_n || cand[j] != ',' || !d1) { okparse = 0; break; }
        j++;
        while (j < cand_n && cand[j] >= '0' && cand[j] <= '9' && d2 < 19) { len = len * 10 + (cand[j] - '0'); j++; d2++; }
        if (j >= cand_n || cand[j] != '>' || !d2) { okparse = 0; break; }
        (void)off;
        f += (size_t)len; i = j + 1;
        if (f - cuts[nc - 1].f0 >= seg && i < cand_n) { if (nc + 2 > cap) { cap *= 2; cuts = realloc(cuts, cap * sizeof *cuts); } cuts[nc++] = (cut_t){i, f}; }
    }
    int rc = 0;
    if (!okparse || f != e) { *bad_at = cand_n; rc = 1; }
    else {
        cuts[nc++] = (cut_t){cand_n, e};
        lzc_t z; memset(&z, 0, sizeof z);
        z.esc = esc; z.e = e; z.window = window; z.cand = cand; z.cand_n = cand_n; z.cuts = cuts; z.n_cuts = nc;
        atomic_store(&z.bad, (size_t)-1);
        run_threads(lzc_worker, &z, threads < 1 ? 1 : threads);
        if (atomic_load(&z.bad) != (size_t)-1) { *bad_at = atomic_load(&z.bad); rc = 1; }
    }
    free(cuts); rsn_oracle_free(esc);
    return rc;
}
