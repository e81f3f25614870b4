/*
 * rsn_oracle.h -- CPU restatement of go-compression/raisin's Huffman and LZSS
 * codecs (compressor/huffman/huffman.go, compressor/lz/lzss.go).
 *
 * THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only tests/, the smoke test
 * in __graft_entry__.py and the `cpu_baseline` leg of bench.py may load it.
 * The shipped library (raisin_amd/librsn.so) never links, loads or calls it.
 *
 * PARITY PINNING STATUS: "partially pinned".  The reference is pure Go and no
 * Go toolchain exists in the build image, so the reference itself cannot be
 * executed.  The reference's own tests hold no golden bytes for this path
 * (round-trips only: compressor/lz/lzss_test.go:25-47, cmd/cli_test.go:33-40).
 * What pins this oracle:
 *   - the reference-published compressed SIZES in README.md:153-167
 *     (huffman 13B->40B, 25B->23B; lzss 13B->13B; legacy lzss 25B->21B);
 *   - round-trip losslessness on the reference's only fixture (samIAm);
 *   - byte-for-byte agreement of two independently written restatements
 *     (this C code vs. the survey's throw-away model, hashes in SURVEY.md 8c),
 *     and, for LZSS, agreement between the literal all-positions form and the
 *     lazy form below.
 * Go's container/heap (stdlib, not under /root/reference) is restated from its
 * published algorithm (go1.15 src/container/heap/heap.go: Init/Push/Pop/up/down).
 */
#ifndef RSN_ORACLE_H
#define RSN_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RSN_ORACLE_OK 0
#define RSN_ORACLE_ERR (-1)

/* huffman.go:299 Compress.  Header entries are written in ascending rune order,
 * except that '\\' is moved to the front when it would be the last entry (the
 * reference's own decoder panics on that order, huffman.go:210).  Any order is a
 * possible reference output (Go map iteration, huffman.go:312). */
int rsn_oracle_huffman_compress(const uint8_t *in, size_t n, uint8_t **out, size_t *out_n);

/* huffman.go:327 Decompress (decode :258, decodeTree :196, findCodes :131).
 * strict_ref_limit != 0 reproduces the reference's 900000-bit recursion limit
 * (huffman.go:132) as an error. */
int rsn_oracle_huffman_decompress(const uint8_t *in, size_t n, int strict_ref_limit,
                                  uint8_t **out, size_t *out_n);

/* Introspection for tests: the (rune, freq, code, len) table in printCodes DFS
 * order (huffman.go:110).  Returns number of symbols or <0. codes are MSB-first
 * values right-aligned in 64 bits. */
int64_t rsn_oracle_huffman_table(const uint8_t *in, size_t n, uint32_t *runes, uint64_t *freqs,
                                 uint64_t *codes, uint8_t *lens, size_t cap);

/* Go `for _, c := range string(b)` (huffman.go:309): writes one rune per decoded
 * position, returns the count.  runes must hold n entries. */
size_t rsn_oracle_utf8_runes(const uint8_t *in, size_t n, uint32_t *runes);

/* lzss.go:109 CompressAsync(data, _, window); window<=0 means unbounded
 * (lzss.go:125).  Lazy form: matches are only evaluated at parse positions
 * (the output depends on nothing else). */
int rsn_oracle_lzss_compress(const uint8_t *in, size_t n, int64_t window, uint8_t **out, size_t *out_n);

/* Same bytes, literal form: a Reference is computed for EVERY position by
 * repeated leftmost substring search exactly as compressorWorker does
 * (lzss.go:166-184), then compacted (lzss.go:134-151).  O(N*W*L): small inputs. */
int rsn_oracle_lzss_compress_allpos(const uint8_t *in, size_t n, int64_t window, uint8_t **out, size_t *out_n);

/* lzss.go:224 legacy synchronous Compress (not on the .rsn path; README's 21-byte answer). */
int rsn_oracle_lzss_compress_legacy(const uint8_t *in, size_t n, int64_t window, uint8_t **out, size_t *out_n);

/* lzss.go:323 Decompress */
int rsn_oracle_lzss_decompress(const uint8_t *in, size_t n, uint8_t **out, size_t *out_n);

/* lzss.go:369 / :391 */
int rsn_oracle_lzss_escape(const uint8_t *in, size_t n, uint8_t **out, size_t *out_n);
int rsn_oracle_lzss_unescape(const uint8_t *in, size_t n, uint8_t **out, size_t *out_n);

/* Per-position match table on the ESCAPED stream (lzss.go:166-184): size[i]==0
 * means literal; otherwise (off[i], size[i]).  Arrays hold e entries. */
int rsn_oracle_lzss_matches(const uint8_t *esc, size_t e, int64_t window, uint32_t *off, uint32_t *size);

/* ---- oracle/cpu_baseline.c: the same functions on `threads` host cores (bench.py's cpu_baseline).
 * Same bytes out as the single-threaded forms above (tests/test_oracle.py). */
int rsn_baseline_huffman_compress_mt(const uint8_t *in, size_t n, int threads, uint8_t **out, size_t *out_n);
int rsn_baseline_huffman_decompress_mt(const uint8_t *in, size_t n, int threads, uint8_t **out, size_t *out_n);
/* grain = positions per task (1 = one task per position, lzss.go:117-130) */
int rsn_baseline_lzss_compress_mt(const uint8_t *in, size_t n, int64_t window, int threads, size_t grain, uint8_t **out, size_t *out_n);

/* Is `cand` byte for byte what rsn_oracle_lzss_compress(in, n, window) returns?  Decided without producing that output
 * serially: by induction over segments cut right after the candidate's tokens, each re-encoded by the oracle's own greedy
 * loop (cpu_baseline.c).  Returns 0 = identical, 1 = not (*bad_at = candidate offset of the first differing segment). */
int rsn_baseline_lzss_check(const uint8_t *in, size_t n, int64_t window, int threads, size_t seg, const uint8_t *cand, size_t cand_n, size_t *bad_at);

void rsn_oracle_free(void *p);
const char *rsn_oracle_last_error(void);

#ifdef __cplusplus
}
#endif
#endif
