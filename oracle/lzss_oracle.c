/*
 * lzss_oracle.c -- CPU restatement of raisin's compressor/lz/lzss.go.
 * TEST INFRASTRUCTURE ONLY (see rsn_oracle.h).  Parity: partially pinned.
 * Line citations are into /root/reference/compressor/lz/lzss.go.
 */
#define _GNU_SOURCE
#include "rsn_oracle.h"

#include <stdio.h>
#include <stdlib.h>
#include <string.h>

void rsn_oracle_set_error(const char *m);

typedef struct { uint8_t *p; size_t n, cap; } buf_t;
static void buf_put(buf_t *b, const void *s, size_t k) {
    if (b->n + k > b->cap) {
        size_t c = b->cap ? b->cap * 2 : 256;
        while (c < b->n + k) c *= 2;
        b->p = realloc(b->p, c); b->cap = c;
    }
    memcpy(b->p + b->n, s, k); b->n += k;
}
static void buf_putc(buf_t *b, uint8_t c) { buf_put(b, &c, 1); }

/* EncodeOpeningSymbols lzss.go:369-389.  foundEscape is never set to true
 * (the branch at :380 is unreachable), so: '<'->FF, FF->5C FF, 5C->5C 5C. */
static void escape(const uint8_t *in, size_t n, buf_t *o) {
    for (size_t i = 0; i < n; i++) {
        uint8_t v = in[i];
        if (v == 0x3C) v = 0xFF;
        else if (v == 0xFF || v == 0x5C) buf_putc(o, 0x5C);
        buf_putc(o, v);
    }
}

/* DecodeOpeningSymbols lzss.go:391-406 */
static void unescape(const uint8_t *in, size_t n, buf_t *o) {
    int esc = 0;
    for (size_t i = 0; i < n; i++) {
        uint8_t v = in[i];
        if (v == 0xFF && !esc) buf_putc(o, 0x3C);
        else if (v == 0x5C && !esc) esc = 1;
        else { esc = 0; buf_putc(o, v); }
    }
}

int rsn_oracle_lzss_escape(const uint8_t *in, size_t n, uint8_t **out, size_t *out_n) {
    buf_t o = {0}; escape(in, n, &o);
    if (!o.p) o.p = malloc(1);
    *out = o.p; *out_n = o.n; return RSN_ORACLE_OK;
}
int rsn_oracle_lzss_unescape(const uint8_t *in, size_t n, uint8_t **out, size_t *out_n) {
    buf_t o = {0}; unescape(in, n, &o);
    if (!o.p) o.p = malloc(1);
    *out = o.p; *out_n = o.n; return RSN_ORACLE_OK;
}

/* getEncoding lzss.go:318-320 */
static int encoding(size_t off, size_t size, char *o) { return sprintf(o, "<%zu,%zu>", off, size); }

/* compressorWorker lzss.go:166-184 + FindReverseSlice :418-421, closed form:
 * the longest L such that fc[i:i+L] occurs entirely inside the window
 * fc[ws:i]; offset = i - (leftmost start of that occurrence).  size 0 = literal. */
static void match_at(const uint8_t *fc, size_t e, size_t i, int64_t window, size_t *off, size_t *size) {
    size_t ws = 0;
    if (window > 0 && i > (size_t)window) ws = i - (size_t)window; /* :123-127 */
    size_t best = 0, bestj = 0;
    size_t rem = e - i;
    const uint8_t *p = fc + ws, *end = fc + i;
    while (p < end) {
        p = memchr(p, fc[i], (size_t)(end - p));
        if (!p) break;
        size_t j = (size_t)(p - fc);
        size_t cap = i - j; if (cap > rem) cap = rem;
        if (cap > best) { /* a later (righter) candidate only wins if strictly longer => leftmost kept */
            size_t l = 1;
            while (l < cap && fc[j + l] == fc[i + l]) l++;
            if (l > best) { best = l; bestj = j; }
        }
        p++;
    }
    *size = best; *off = best ? i - bestj : 0;
}

/* compaction lzss.go:134-151 */
static void compact_emit(buf_t *o, const uint8_t *fc, size_t i, size_t off, size_t size) {
    if (size) {
        char enc[48]; int k = encoding(off, size, enc);
        if ((size_t)k < size) buf_put(o, enc, (size_t)k);   /* :143 */
        else buf_put(o, fc + i, size);                        /* :146 ref.value */
    } else buf_putc(o, fc[i]);                                /* :149 */
}

int rsn_oracle_lzss_compress(const uint8_t *in, size_t n, int64_t window, uint8_t **out, size_t *out_n) {
    buf_t fc = {0}; escape(in, n, &fc);
    buf_t o = {0};
    size_t i = 0;
    while (i < fc.n) {
        size_t off, size; match_at(fc.p, fc.n, i, window, &off, &size);
        compact_emit(&o, fc.p, i, off, size);
        i += size ? size : 1;  /* ignoreNextChars = size-1 (:142) */
    }
    free(fc.p);
    if (!o.p) o.p = malloc(1);
    *out = o.p; *out_n = o.n; return RSN_ORACLE_OK;
}

/* The same greedy loop (lzss.go:134-151) over the ESCAPED stream from position `start` until it first lands at or beyond
 * `stop`: the output of that stretch and where it landed.  If `start` lies on the chain of the whole stream, these are the
 * whole stream's bytes for the stretch -- the unit of work of rsn_baseline_lzss_check (oracle/cpu_baseline.c). */
int rsn_oracle_lzss_compress_range(const uint8_t *esc, size_t e, int64_t window, size_t start, size_t stop, uint8_t **out, size_t *out_n, size_t *landed) {
    buf_t o = {0};
    size_t i = start;
    while (i < e && i < stop) {
        size_t off, size; match_at(esc, e, i, window, &off, &size);
        compact_emit(&o, esc, i, off, size);
        i += size ? size : 1;
    }
    if (!o.p) o.p = malloc(1);
    *out = o.p; *out_n = o.n; *landed = i; return RSN_ORACLE_OK;
}

/* positions [lo, hi) only: the unit of work of oracle/cpu_baseline.c's threads */
int rsn_oracle_lzss_matches_range(const uint8_t *esc, size_t e, int64_t window, size_t lo, size_t hi, uint32_t *off, uint32_t *size) {
    for (size_t i = lo; i < hi && i < e; i++) { size_t o, s; match_at(esc, e, i, window, &o, &s); off[i] = (uint32_t)o; size[i] = (uint32_t)s; }
    return RSN_ORACLE_OK;
}

/* compaction lzss.go:134-151 over a finished match table */
int rsn_oracle_lzss_compact(const uint8_t *esc, size_t e, const uint32_t *off, const uint32_t *size, uint8_t **out, size_t *out_n) {
    buf_t o = {0};
    for (size_t i = 0; i < e; ) { compact_emit(&o, esc, i, off[i], size[i]); i += size[i] ? size[i] : 1; }
    if (!o.p) o.p = malloc(1);
    *out = o.p; *out_n = o.n; return RSN_ORACLE_OK;
}

int rsn_oracle_lzss_matches(const uint8_t *esc, size_t e, int64_t window, uint32_t *off, uint32_t *size) {
    for (size_t i = 0; i < e; i++) { size_t o, s; match_at(esc, e, i, window, &o, &s); off[i] = (uint32_t)o; size[i] = (uint32_t)s; }
    return RSN_ORACLE_OK;
}

/* Literal form: compressorWorker's recursion unrolled -- extend scanBytes one
 * byte at a time, each time a fresh leftmost search (bytes.Index) over the
 * whole window, exactly as lzss.go:166-184 does. */
int rsn_oracle_lzss_compress_allpos(const uint8_t *in, size_t n, int64_t window, uint8_t **out, size_t *out_n) {
    buf_t fc = {0}; escape(in, n, &fc);
    size_t e = fc.n;
    size_t *offs = malloc((e ? e : 1) * sizeof *offs), *sizes = malloc((e ? e : 1) * sizeof *sizes);
    for (size_t i = 0; i < e; i++) {
        size_t ws = 0;
        if (window > 0 && i > (size_t)window) ws = i - (size_t)window;
        const uint8_t *sb = fc.p + ws; size_t sbn = i - ws;
        size_t L = 0, off = 0;
        for (size_t len = 1; i + len <= e; len++) {           /* scanBytes = fc[i:i+len] */
            const uint8_t *hit = sbn >= len ? memmem(sb, sbn, fc.p + i, len) : NULL;
            if (!hit) break;                                   /* !found => previous level stands (:175-176) */
            L = len; off = sbn - (size_t)(hit - sb);           /* negativeOffset = len(searchBuffer) - index (:170) */
        }
        offs[i] = off; sizes[i] = L;
    }
    buf_t o = {0};
    size_t ignore = 0;
    for (size_t i = 0; i < e; i++) {                           /* :136-151 */
        if (ignore > 0) { ignore--; continue; }
        if (sizes[i]) ignore = sizes[i] - 1;
        compact_emit(&o, fc.p, i, offs[i], sizes[i]);
    }
    free(offs); free(sizes); free(fc.p);
    if (!o.p) o.p = malloc(1);
    *out = o.p; *out_n = o.n; return RSN_ORACLE_OK;
}

/* FindReverse lzss.go:423-433: the loop decrements twice per iteration, so only
 * indices len-1, len-3, ... are tested. */
static long find_reverse(const uint8_t *s, size_t n, uint8_t v) {
    for (long i = (long)n - 1; i >= 0; i--) { if (s[i] == v) return i; i--; }
    return -1;
}

/* legacy Compress lzss.go:224-316, transliterated including its quirks */
int rsn_oracle_lzss_compress_legacy(const uint8_t *in, size_t n, int64_t window, uint8_t **out, size_t *out_n) {
    buf_t fc = {0}; escape(in, n, &fc);
    buf_t sb = {0}, o = {0}, add = {0};
    int checkNext = 0; size_t startPtr = 0, checkOff = 0;
    for (size_t k = 0; k < fc.n; k++) {
        uint8_t b = fc.p[k];
        long index = 0; int found = 0;
        if (!checkNext) { index = find_reverse(sb.p, sb.n, b); found = index >= 0; }
        else {
            size_t dim = 0;
            if (window > 0 && sb.n > (size_t)window) dim = sb.n - (size_t)window;  /* :249-251 */
            buf_putc(&add, b);                                                      /* append(checkBytesToAdd, fileByte) */
            const uint8_t *hit = (sb.n - dim) >= add.n ? memmem(sb.p + dim, sb.n - dim, add.p, add.n) : NULL;
            add.n--;
            if (hit) { index = (long)(hit - (sb.p + dim)); found = 1; }            /* index relative to the SLICED buffer (:252) */
        }
        if (found && checkNext) {
            startPtr = sb.n - (size_t)index; checkOff++; buf_putc(&add, b);        /* :256-259 pointer from UNSLICED length */
        } else if (found && !checkNext) {
            startPtr = sb.n - (size_t)index; checkOff = 1; checkNext = 1; buf_putc(&add, b);
        } else {
            if (checkNext) {
                char enc[48]; int el = encoding(startPtr, checkOff, enc);
                int shouldAdd = !((size_t)el > add.n);                              /* :272 */
                if (shouldAdd) buf_put(&o, enc, (size_t)el); else buf_put(&o, add.p, add.n);
                startPtr = 0; checkOff = 0; checkNext = 0;
                buf_put(&sb, add.p, add.n); add.n = 0;
            }
            buf_putc(&o, b);
        }
        if (!checkNext) buf_putc(&sb, b);
    }
    if (checkNext) {
        char enc[48]; int el = encoding(startPtr, checkOff, enc);
        if (!((size_t)el > add.n)) buf_put(&o, enc, (size_t)el); else buf_put(&o, add.p, add.n);
    }
    free(fc.p); free(sb.p); free(add.p);
    if (!o.p) o.p = malloc(1);
    *out = o.p; *out_n = o.n; return RSN_ORACLE_OK;
}

/* strconv.Atoi as used at lzss.go:338,346 (error ignored => 0 on any syntax error) */
static long long go_atoi(const char *s, int n) {
    int i = 0, neg = 0;
    if (n == 0) return 0;
    if (s[0] == '+' || s[0] == '-') { neg = s[0] == '-'; i = 1; if (n == 1) return 0; }
    long long v = 0;
    for (; i < n; i++) { if (s[i] < '0' || s[i] > '9' || v > (1LL << 56)) return 0; v = v * 10 + (s[i] - '0'); }
    return neg ? -v : v;
}

/* Decompress lzss.go:323-364.  Note: Go's searchBuffer[a:a+len] may legally reach
 * past len() up to cap(); that only happens on corrupt input and is reported as
 * an error here. */
int rsn_oracle_lzss_decompress(const uint8_t *in, size_t n, uint8_t **out, size_t *out_n) {
    *out = NULL; *out_n = 0;
    buf_t sbuf = {0};
    int state = 0; /* 0: looking for '<', 1: reading pointer until ',', 2: reading length until '>' */
    char num[32]; int nn = 0; long long pointer = 0;
    for (size_t i = 0; i < n; i++) {
        uint8_t c = in[i];
        if (state == 0 && c == '<') { state = 1; nn = 0; }
        else if (state == 1) {
            if (c == ',') { pointer = go_atoi(num, nn); nn = 0; state = 2; }
            else if (nn < 31) num[nn++] = (char)c; else num[0] = 'x';
        } else if (state == 2) {
            if (c == '>') {
                long long len = go_atoi(num, nn); nn = 0; state = 0;
                long long absp = (long long)sbuf.n - pointer;
                if (absp < 0 || len < 0 || absp + len > (long long)sbuf.n) {
                    free(sbuf.p); rsn_oracle_set_error("lzss: back-reference outside the buffer (reference: slice bounds out of range, lzss.go:350)"); return RSN_ORACLE_ERR;
                }
                for (long long k = 0; k < len; k++) buf_putc(&sbuf, sbuf.p[absp + k]);
            } else if (nn < 31) num[nn++] = (char)c; else num[0] = 'x';
        } else { buf_putc(&sbuf, c); }
    }
    buf_t o = {0}; unescape(sbuf.p, sbuf.n, &o);
    free(sbuf.p);
    if (!o.p) o.p = malloc(1);
    *out = o.p; *out_n = o.n; return RSN_ORACLE_OK;
}
