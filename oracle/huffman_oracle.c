/*
 * huffman_oracle.c -- CPU restatement of raisin's compressor/huffman/huffman.go.
 * TEST INFRASTRUCTURE ONLY (see rsn_oracle.h).  Parity: partially pinned.
 *
 * Every function cites the reference lines it follows (paths relative to
 * /root/reference/compressor/huffman/huffman.go unless noted).
 */
#include "rsn_oracle.h"

#include <stdio.h>
#include <stdlib.h>
#include <string.h>

static __thread char g_err[256];
const char *rsn_oracle_last_error(void) { return g_err; }
void rsn_oracle_set_error(const char *m) { snprintf(g_err, sizeof g_err, "%s", m); }
void rsn_oracle_free(void *p) { free(p); }

#define RUNE_ERROR 0xFFFDu
#define MAX_RUNE 0x110000u

/* Go's UTF-8 decoding as performed by `for _, c := range string(b)`
 * (huffman.go:235,309; semantics of unicode/utf8.DecodeRune, go1.15):
 * invalid or truncated sequences yield U+FFFD and consume exactly ONE byte. */
static uint32_t go_decode_rune(const uint8_t *p, size_t n, int *size) {
    uint8_t b0 = p[0];
    *size = 1;
    if (b0 < 0x80) return b0;
    int need;
    uint8_t lo = 0x80, hi = 0xBF;
    if (b0 >= 0xC2 && b0 <= 0xDF) need = 2;
    else if (b0 == 0xE0) { need = 3; lo = 0xA0; }
    else if ((b0 >= 0xE1 && b0 <= 0xEC) || b0 == 0xEE || b0 == 0xEF) need = 3;
    else if (b0 == 0xED) { need = 3; hi = 0x9F; }
    else if (b0 == 0xF0) { need = 4; lo = 0x90; }
    else if (b0 >= 0xF1 && b0 <= 0xF3) need = 4;
    else if (b0 == 0xF4) { need = 4; hi = 0x8F; }
    else return RUNE_ERROR;
    if (n < (size_t)need) return RUNE_ERROR;
    uint8_t b1 = p[1];
    if (b1 < lo || b1 > hi) return RUNE_ERROR;
    if (need == 2) { *size = 2; return ((uint32_t)(b0 & 0x1F) << 6) | (b1 & 0x3F); }
    uint8_t b2 = p[2];
    if (b2 < 0x80 || b2 > 0xBF) return RUNE_ERROR;
    if (need == 3) { *size = 3; return ((uint32_t)(b0 & 0x0F) << 12) | ((uint32_t)(b1 & 0x3F) << 6) | (b2 & 0x3F); }
    uint8_t b3 = p[3];
    if (b3 < 0x80 || b3 > 0xBF) return RUNE_ERROR;
    *size = 4;
    return ((uint32_t)(b0 & 0x07) << 18) | ((uint32_t)(b1 & 0x3F) << 12) | ((uint32_t)(b2 & 0x3F) << 6) | (b3 & 0x3F);
}

/* Go string(rune) (huffman.go:138,314): UTF-8 encoding; decoded runes are
 * always valid scalar values or U+FFFD. */
static int go_encode_rune(uint32_t r, uint8_t *o) {
    if (r < 0x80) { o[0] = (uint8_t)r; return 1; }
    if (r < 0x800) { o[0] = 0xC0 | (r >> 6); o[1] = 0x80 | (r & 0x3F); return 2; }
    if (r >= MAX_RUNE || (r >= 0xD800 && r <= 0xDFFF)) r = RUNE_ERROR;
    if (r < 0x10000) { o[0] = 0xE0 | (r >> 12); o[1] = 0x80 | ((r >> 6) & 0x3F); o[2] = 0x80 | (r & 0x3F); return 3; }
    o[0] = 0xF0 | (r >> 18); o[1] = 0x80 | ((r >> 12) & 0x3F); o[2] = 0x80 | ((r >> 6) & 0x3F); o[3] = 0x80 | (r & 0x3F);
    return 4;
}

size_t rsn_oracle_utf8_runes(const uint8_t *in, size_t n, uint32_t *runes) {
    size_t i = 0, k = 0;
    while (i < n) { int sz; runes[k++] = go_decode_rune(in + i, n - i, &sz); i += sz; }
    return k;
}

/* ---- tree: leaves ordered (freq asc, rune asc) (huffman.go:64-87), then Go
 * container/heap Pop/Pop/Push with Less = freq only (huffman.go:43-45,93-102). */
typedef struct {
    uint32_t n_leaves;
    uint32_t n_nodes;        /* leaves first, then internal nodes in creation order */
    uint64_t *freq;
    int32_t *left, *right;   /* -1 for leaves */
    uint32_t *rune;          /* for leaves */
    int32_t root;
} tree_t;

static void tree_free(tree_t *t) { free(t->freq); free(t->left); free(t->right); free(t->rune); memset(t, 0, sizeof *t); }

typedef struct { uint32_t rune; uint64_t freq; } leaf_t;
static int leaf_cmp(const void *a, const void *b) {
    const leaf_t *x = a, *y = b;
    if (x->freq != y->freq) return x->freq < y->freq ? -1 : 1;
    return x->rune < y->rune ? -1 : (x->rune > y->rune);
}

/* Go container/heap (go1.15 src/container/heap/heap.go), on an int32 array of node ids */
static void heap_up(int32_t *h, const uint64_t *f, int j) {
    for (;;) {
        int i = (j - 1) / 2; /* Go integer division truncates toward zero: j=0 -> i=0 */
        if (i == j || !(f[h[j]] < f[h[i]])) break;
        int32_t t = h[i]; h[i] = h[j]; h[j] = t;
        j = i;
    }
}
static void heap_down(int32_t *h, const uint64_t *f, int i0, int n) {
    int i = i0;
    for (;;) {
        int j1 = 2 * i + 1;
        if (j1 >= n || j1 < 0) break;
        int j = j1;
        int j2 = j1 + 1;
        if (j2 < n && f[h[j2]] < f[h[j1]]) j = j2;
        if (!(f[h[j]] < f[h[i]])) break;
        int32_t t = h[i]; h[i] = h[j]; h[j] = t;
        i = j;
    }
}

/* buildTree huffman.go:58-103.  leaves[] need not be sorted on entry. */
static int build_tree(leaf_t *leaves, uint32_t a, tree_t *t) {
    memset(t, 0, sizeof *t);
    if (a == 0) { rsn_oracle_set_error("huffman: empty symbol table (reference panics in heap.Pop, huffman.go:102)"); return -1; }
    qsort(leaves, a, sizeof *leaves, leaf_cmp);
    uint32_t cap = 2 * a;
    t->freq = malloc(cap * sizeof *t->freq);
    t->left = malloc(cap * sizeof *t->left);
    t->right = malloc(cap * sizeof *t->right);
    t->rune = malloc(cap * sizeof *t->rune);
    int32_t *h = malloc(a * sizeof *h);
    for (uint32_t i = 0; i < a; i++) {
        t->freq[i] = leaves[i].freq; t->rune[i] = leaves[i].rune; t->left[i] = t->right[i] = -1; h[i] = (int32_t)i;
    }
    t->n_leaves = a; t->n_nodes = a;
    int n = (int)a;
    /* heap.Init (huffman.go:93) */
    for (int i = n / 2 - 1; i >= 0; i--) heap_down(h, t->freq, i, n);
    while (n > 1) {
        /* a := heap.Pop */
        int m = n - 1; int32_t tmp = h[0]; h[0] = h[m]; h[m] = tmp; heap_down(h, t->freq, 0, m); int32_t x = h[m]; n = m;
        /* b := heap.Pop */
        m = n - 1; tmp = h[0]; h[0] = h[m]; h[m] = tmp; heap_down(h, t->freq, 0, m); int32_t y = h[m]; n = m;
        uint32_t id = t->n_nodes++;
        t->freq[id] = t->freq[x] + t->freq[y]; t->left[id] = x; t->right[id] = y; t->rune[id] = 0;
        /* heap.Push */
        h[n] = (int32_t)id; n++; heap_up(h, t->freq, n - 1);
    }
    t->root = h[0];
    free(h);
    return 0;
}

/* printCodes huffman.go:110-127: DFS, left='0' first.  Fills per-leaf code/len
 * (indexed by leaf id) and, optionally, DFS order. */
static int assign_codes(const tree_t *t, uint64_t *code, uint8_t *len, uint32_t *dfs_order) {
    uint32_t k = 0;
    typedef struct { int32_t node; uint64_t code; uint32_t len; } fr;
    fr *st = malloc((t->n_nodes + 1) * sizeof *st);
    int sp = 0;
    st[sp++] = (fr){t->root, 0, 0};
    while (sp) {
        fr f = st[--sp];
        if (t->left[f.node] < 0) {
            if (f.len > 64) { free(st); rsn_oracle_set_error("huffman: code longer than 64 bits (oracle limit)"); return -1; }
            code[f.node] = f.code; len[f.node] = (uint8_t)f.len;
            if (dfs_order) dfs_order[k] = (uint32_t)f.node;
            k++;
        } else {
            /* push right first so that left is visited first */
            st[sp++] = (fr){t->right[f.node], (f.len < 64 ? (f.code << 1) : 0) | 1, f.len + 1};
            st[sp++] = (fr){t->left[f.node], (f.len < 64 ? (f.code << 1) : 0), f.len + 1};
        }
    }
    free(st);
    return 0;
}

typedef struct { uint8_t *p; size_t n, cap; } buf_t;
static int buf_put(buf_t *b, const void *s, size_t k) {
    if (b->n + k > b->cap) {
        size_t c = b->cap ? b->cap * 2 : 256;
        while (c < b->n + k) c *= 2;
        uint8_t *q = realloc(b->p, c);
        if (!q) return -1;
        b->p = q; b->cap = c;
    }
    memcpy(b->p + b->n, s, k); b->n += k;
    return 0;
}

/* collect the rune histogram (huffman.go:306-311) into leaves[] in ascending rune order */
static leaf_t *histogram(const uint8_t *in, size_t n, uint32_t *a_out) {
    uint64_t *cnt = calloc(MAX_RUNE, sizeof *cnt);
    size_t i = 0;
    while (i < n) { int sz; uint32_t r = go_decode_rune(in + i, n - i, &sz); cnt[r]++; i += sz; }
    uint32_t a = 0;
    for (uint32_t r = 0; r < MAX_RUNE; r++) if (cnt[r]) a++;
    leaf_t *lv = malloc((a ? a : 1) * sizeof *lv);
    uint32_t k = 0;
    for (uint32_t r = 0; r < MAX_RUNE; r++) if (cnt[r]) { lv[k].rune = r; lv[k].freq = cnt[r]; k++; }
    free(cnt);
    *a_out = a;
    return lv;
}

/* header emit huffman.go:312-318 in the canonical order documented in rsn_oracle.h */
static void emit_header(buf_t *b, const leaf_t *by_rune, uint32_t a) {
    /* by_rune is ascending rune.  If the last entry is '\\', emit it first. */
    int bs_first = (a > 1 && by_rune[a - 1].rune == 0x5C);
    for (uint32_t pass = 0; pass < 2; pass++) {
        for (uint32_t i = 0; i < a; i++) {
            int is_bs_last = (bs_first && i == a - 1);
            if ((pass == 0) != (is_bs_last != 0)) continue; /* pass0: only the moved '\\'; pass1: the rest */
            char num[32];
            int k = snprintf(num, sizeof num, "%llu|", (unsigned long long)by_rune[i].freq);
            buf_put(b, num, (size_t)k);
            if (by_rune[i].rune == 10) buf_put(b, "\\n", 2);
            else { uint8_t u[4]; int m = go_encode_rune(by_rune[i].rune, u); buf_put(b, u, (size_t)m); }
        }
    }
}

int rsn_oracle_huffman_compress(const uint8_t *in, size_t n, uint8_t **out, size_t *out_n) {
    *out = NULL; *out_n = 0;
    uint32_t a;
    leaf_t *by_rune = histogram(in, n, &a);
    if (a == 0) { free(by_rune); rsn_oracle_set_error("huffman: empty input (reference panics in heap.Pop, huffman.go:102)"); return RSN_ORACLE_ERR; }
    buf_t b = {0};
    emit_header(&b, by_rune, a);
    leaf_t *lv = malloc(a * sizeof *lv);
    memcpy(lv, by_rune, a * sizeof *lv);
    tree_t t;
    if (build_tree(lv, a, &t)) { free(lv); free(by_rune); free(b.p); return RSN_ORACLE_ERR; }
    uint64_t *code = calloc(t.n_nodes, sizeof *code);
    uint8_t *len = calloc(t.n_nodes, 1);
    if (assign_codes(&t, code, len, NULL)) { free(code); free(len); tree_free(&t); free(lv); free(by_rune); free(b.p); return RSN_ORACLE_ERR; }
    /* rune -> leaf id */
    int32_t *leaf_of = malloc(MAX_RUNE * sizeof *leaf_of);
    memset(leaf_of, 0xFF, MAX_RUNE * sizeof *leaf_of);
    uint64_t total_bits = 0;
    for (uint32_t i = 0; i < a; i++) { leaf_of[t.rune[i]] = (int32_t)i; total_bits += t.freq[i] * len[i]; }
    /* encode huffman.go:229-256: pad = (8 - len%8), 8 -> 0; payload = 0^pad || S, big-endian */
    unsigned pad = (unsigned)((8 - total_bits % 8) % 8);
    buf_put(&b, "\\\n", 2);
    uint8_t pb = (uint8_t)pad;
    buf_put(&b, &pb, 1);
    size_t pay = (size_t)((total_bits + pad) / 8);
    size_t base = b.n;
    {
        uint8_t *z = calloc(pay ? pay : 1, 1);
        buf_put(&b, z, pay);
        free(z);
    }
    uint8_t *P = b.p + base;
    uint64_t bitpos = pad;
    size_t i = 0;
    while (i < n) {
        int sz; uint32_t r = go_decode_rune(in + i, n - i, &sz); i += sz;
        int32_t id = leaf_of[r];
        uint64_t c = code[id]; unsigned l = len[id];
        for (int k = (int)l - 1; k >= 0; k--) {
            if ((c >> k) & 1) P[bitpos >> 3] |= (uint8_t)(0x80u >> (bitpos & 7));
            bitpos++;
        }
    }
    free(leaf_of); free(code); free(len); tree_free(&t); free(lv); free(by_rune);
    *out = b.p; *out_n = b.n;
    return RSN_ORACLE_OK;
}

int64_t rsn_oracle_huffman_table(const uint8_t *in, size_t n, uint32_t *runes, uint64_t *freqs,
                                 uint64_t *codes, uint8_t *lens, size_t cap) {
    uint32_t a;
    leaf_t *lv = histogram(in, n, &a);
    if (a == 0) { free(lv); rsn_oracle_set_error("huffman: empty input"); return -1; }
    tree_t t;
    if (build_tree(lv, a, &t)) { free(lv); return -1; }
    uint64_t *code = calloc(t.n_nodes, sizeof *code);
    uint8_t *len = calloc(t.n_nodes, 1);
    uint32_t *order = malloc(a * sizeof *order);
    int rc = assign_codes(&t, code, len, order);
    if (!rc) for (uint32_t k = 0; k < a && k < cap; k++) {
        uint32_t id = order[k];
        runes[k] = t.rune[id]; freqs[k] = t.freq[id]; codes[k] = code[id]; lens[k] = len[id];
    }
    free(order); free(code); free(len); tree_free(&t); free(lv);
    return rc ? -1 : (int64_t)a;
}

/* decodeTree huffman.go:196-227: byte-indexed scan of the header.  Returns leaves. */
static leaf_t *parse_header(const uint8_t *h, size_t hn, uint32_t *a_out) {
    uint64_t *freq = calloc(MAX_RUNE, sizeof *freq);
    uint8_t *present = calloc(MAX_RUNE, 1);
    uint64_t acc = 0; int digits = 0;
    for (size_t i = 0; i < hn; i++) {
        uint8_t c = h[i];
        if (c != '|') {
            if (c >= '0' && c <= '9') { /* strconv.Atoi(string(tree[i])) succeeds only for a digit (:203) */
                /* Atoi of the collected digits: leading zeros are fine, a value past 2^63-1 comes back as MaxInt64 (ErrRange dropped, :207) */
                const uint64_t kmax = 0x7FFFFFFFFFFFFFFFull, dg = (uint64_t)(c - '0');
                acc = (acc > (kmax - dg) / 10) ? kmax : acc * 10 + dg; digits++;
            }
        } else {
            uint64_t f = acc; /* Atoi("") -> 0, error ignored (:207) */
            acc = 0; digits = 0;
            if (i + 1 >= hn) { free(freq); free(present); rsn_oracle_set_error("huffman: header ends after '|' (reference: index out of range, huffman.go:210)"); return NULL; }
            if (h[i + 1] == '\\') {
                if (i + 2 >= hn) { free(freq); free(present); rsn_oracle_set_error("huffman: header ends after '\\' (reference: index out of range, huffman.go:210)"); return NULL; }
                if (h[i + 2] == 'n') { freq[10] = f; present[10] = 1; i += 2; continue; } /* i++ (:212) then i++ (:222) */
            }
            int sz; uint32_t r = go_decode_rune(h + i + 1, hn - (i + 1), &sz); /* rune starting at byte i+1 (:214-220) */
            freq[r] = f; present[r] = 1;
            i++; /* :222 -- skips exactly one byte; continuation bytes are ignored by the scan */
        }
    }
    uint32_t a = 0;
    for (uint32_t r = 0; r < MAX_RUNE; r++) if (present[r]) a++;
    leaf_t *lv = malloc((a ? a : 1) * sizeof *lv);
    uint32_t k = 0;
    for (uint32_t r = 0; r < MAX_RUNE; r++) if (present[r]) { lv[k].rune = r; lv[k].freq = freq[r]; k++; }
    free(freq); free(present);
    *a_out = a;
    return lv;
}

int rsn_oracle_huffman_decompress(const uint8_t *in, size_t n, int strict_ref_limit,
                                  uint8_t **out, size_t *out_n) {
    *out = NULL; *out_n = 0;
    /* strings.SplitN(content, "\\\n", 2) huffman.go:261 */
    size_t sep = (size_t)-1;
    for (size_t i = 0; i + 1 < n; i++) if (in[i] == 0x5C && in[i + 1] == 0x0A) { sep = i; break; }
    if (sep == (size_t)-1) { rsn_oracle_set_error("huffman: no '\\\\\\n' separator (reference: index out of range, huffman.go:264)"); return RSN_ORACLE_ERR; }
    uint32_t a;
    leaf_t *lv = parse_header(in, sep, &a);
    if (!lv) return RSN_ORACLE_ERR;
    tree_t t;
    if (build_tree(lv, a, &t)) { free(lv); return RSN_ORACLE_ERR; }
    free(lv);
    const uint8_t *sec = in + sep + 2; size_t sn = n - sep - 2;
    unsigned diff = sn ? sec[0] : 0;         /* byteArr[0] huffman.go:275-277 */
    uint64_t nbits = sn ? (uint64_t)(sn - 1) * 8 : 0;
    if (diff > nbits) { tree_free(&t); rsn_oracle_set_error("huffman: pad exceeds payload bits (reference: slice bounds out of range, huffman.go:294)"); return RSN_ORACLE_ERR; }
    const uint8_t *P = sec + 1;
    uint64_t max = nbits - diff;              /* bit i of data = payload bit diff+i */
    buf_t b = {0};
    /* findCodes huffman.go:131-153 */
    int32_t node = t.root; uint64_t i = 0;
    int rc = RSN_ORACLE_OK;
    for (;;) {
        if (strict_ref_limit && i > 900000) { rsn_oracle_set_error("huffman: Max recursion depth (huffman.go:132-134)"); rc = RSN_ORACLE_ERR; break; }
        if (t.left[node] < 0) {
            uint8_t u[4]; int m = go_encode_rune(t.rune[node], u);
            buf_put(&b, u, (size_t)m);
            if (i < max) {
                if (t.root == node && t.left[t.root] < 0) { rsn_oracle_set_error("huffman: single-leaf tree with non-empty payload (reference recurses forever, huffman.go:139-140)"); rc = RSN_ORACLE_ERR; break; }
                node = t.root; continue;
            }
            break;
        }
        if (i >= max) { rsn_oracle_set_error("huffman: payload ends inside a codeword (reference: index out of range, huffman.go:145)"); rc = RSN_ORACLE_ERR; break; }
        uint64_t bp = diff + i;
        int bit = (P[bp >> 3] >> (7 - (bp & 7))) & 1;
        node = bit ? t.right[node] : t.left[node];
        i++;
    }
    tree_free(&t);
    if (rc) { free(b.p); return rc; }
    if (!b.p) b.p = malloc(1);
    *out = b.p; *out_n = b.n;
    return RSN_ORACLE_OK;
}

/* ---- helpers for oracle/cpu_baseline.c (the threaded CPU baseline; same semantics, same code) ---- */

/* Header, codes and lengths for a BYTE histogram (every symbol < 0x80, so rune == byte): what
 * Compress derives from symFreqs (huffman.go:312-318, buildTree :58, printCodes :110). */
int rsn_oracle_huffman_plan_bytes(const uint64_t hist[256], uint64_t code[256], uint8_t len[256], uint8_t **hdr, size_t *hdr_n) {
    leaf_t by_rune[256]; uint32_t a = 0;
    for (uint32_t r = 0; r < 256; r++) if (hist[r]) { by_rune[a].rune = r; by_rune[a].freq = hist[r]; a++; }
    if (a == 0) { rsn_oracle_set_error("huffman: empty input"); return RSN_ORACLE_ERR; }
    buf_t b = {0};
    emit_header(&b, by_rune, a);
    leaf_t lv[256]; memcpy(lv, by_rune, a * sizeof *lv);
    tree_t t;
    if (build_tree(lv, a, &t)) { free(b.p); return RSN_ORACLE_ERR; }
    uint64_t *c = calloc(t.n_nodes, sizeof *c); uint8_t *l = calloc(t.n_nodes, 1);
    int rc = assign_codes(&t, c, l, NULL);
    memset(code, 0, 256 * sizeof *code); memset(len, 0, 256);
    if (!rc) for (uint32_t i = 0; i < a; i++) { code[t.rune[i]] = c[i]; len[t.rune[i]] = l[i]; }
    free(c); free(l); tree_free(&t);
    if (rc) { free(b.p); return RSN_ORACLE_ERR; }
    *hdr = b.p; *hdr_n = b.n;
    return RSN_ORACLE_OK;
}

/* decode's framing and decodeTree (huffman.go:258-297,196-227): the tree as child arrays (left < 0: leaf),
 * where the payload starts, the pad.  The arrays are released with rsn_oracle_free. */
int rsn_oracle_huffman_parse(const uint8_t *in, size_t n, size_t *payload_off, unsigned *pad, int32_t **left, int32_t **right,
                             uint32_t **rune, int32_t *root, uint32_t *n_nodes) {
    size_t sep = (size_t)-1;
    for (size_t i = 0; i + 1 < n; i++) if (in[i] == 0x5C && in[i + 1] == 0x0A) { sep = i; break; }
    if (sep == (size_t)-1) { rsn_oracle_set_error("huffman: no separator"); return RSN_ORACLE_ERR; }
    uint32_t a;
    leaf_t *lv = parse_header(in, sep, &a);
    if (!lv) return RSN_ORACLE_ERR;
    tree_t t;
    if (build_tree(lv, a, &t)) { free(lv); return RSN_ORACLE_ERR; }
    free(lv);
    *payload_off = sep + 3; *pad = n > sep + 2 ? in[sep + 2] : 0;
    *left = t.left; *right = t.right; *rune = t.rune; *root = t.root; *n_nodes = t.n_nodes;
    free(t.freq);
    return RSN_ORACLE_OK;
}

