/*
 * rsn.h -- C ABI of librsn: the MI355X (gfx950) implementation of raisin's
 * Huffman and LZSS codecs (go-compression/raisin compressor/huffman,
 * compressor/lz).  This is the drop-in boundary: a Go host binds these entry
 * points with cgo in place of the per-package Compress/Decompress functions
 * (binding shown in INTEGRATION.md).  Plain pointers and sizes only.
 *
 * All compute runs in hand-written HIP kernels; there is NO CPU fallback.  If
 * no HIP device can be initialised every codec call fails with RSN_ERR_DEVICE.
 *
 * Threading: every entry point is re-entrant and may be called concurrently
 * from any number of host threads (the reference engine runs codecs from
 * concurrent goroutines, engine/engine.go:235-244).  State is per calling
 * thread; nothing is process-global (unlike huffman.go:56,129,193-194).
 *
 * Errors: the reference panics (check(e), index out of range); this library
 * returns a negative code and a thread-local message instead and never aborts.
 * The cgo shim turns a non-zero code back into panic() to keep engine behaviour
 * (engine.go:315-328 recovers it into a "failed" row).  "Never aborts" includes
 * the C++ side's own failures: every entry point below runs inside a guard that
 * turns std::bad_alloc / std::system_error / anything thrown into RSN_ERR_NOMEM
 * or RSN_ERR_DEVICE, and the helper threads of the pipelined calls come from a
 * pool that answers "none to be had" (the call then takes its serial form)
 * instead of throwing -- csrc/rsn_helpers.h, tests/thread_fail_test.cpp.
 */
#ifndef RSN_H
#define RSN_H

#include <stddef.h>
#include <stdint.h>

/* The library is built with -fvisibility=hidden: the entry points below are its whole dynamic symbol table. */
#if defined(__GNUC__) || defined(__clang__)
#define RSN_API __attribute__((visibility("default")))
#else
#define RSN_API
#endif

#ifdef __cplusplus
extern "C" {
#endif

#define RSN_OK 0
#define RSN_ERR_ARG (-1)      /* bad argument (e.g. negative level, lzss.go:43-45) */
#define RSN_ERR_EMPTY (-2)    /* huffman of empty input: reference panics in heap.Pop (huffman.go:102) */
#define RSN_ERR_FORMAT (-3)   /* malformed compressed stream: reference panics (index/slice out of range) */
#define RSN_ERR_DEVICE (-4)   /* HIP runtime / device failure */
#define RSN_ERR_NOMEM (-5)
#define RSN_ERR_LIMIT (-6)    /* outside implementation limits (documented in DESIGN.md) */
#define RSN_ERR_CAPACITY (-7) /* caller-provided device output buffer too small */

#define RSN_LZSS_DEFAULT_WINDOW 4096 /* lzss.go:35 DefaultWindowSize */

/* ---- library / device ------------------------------------------------- */
/* Select the HIP device used by the calling thread.  A thread that never calls this uses device 0,
 * or what RSN_DEVICE says: a number, or "rr" = new thread contexts take the visible devices in turn. */
RSN_API int rsn_device_set(int device);
/* Number of visible HIP devices, or a negative error. */
RSN_API int rsn_device_count(void);
RSN_API const char *rsn_last_error(void); /* thread-local, valid until the next call on this thread */
RSN_API const char *rsn_version(void);
/* Releases what the library keeps between calls: the calling thread's device scratch and pinned
 * staging, the contexts parked by threads that have exited, and the recycled result buffers.
 * Safe at any time between calls; the next call re-allocates what it needs. */
RSN_API void rsn_trim(void);
RSN_API void rsn_free(void *p);           /* releases buffers returned through `out` below (only rsn_free may: they carry a
                                     library header; large ones are recycled for the next result) */

/* ---- host-buffer entry points (what the cgo shim binds) ----------------
 * Input is borrowed for the duration of the call and never modified -- the same
 * bytes may be handed to several calls at once (the pipelined calls pin them in
 * place under a shared, reference-counted table).  Output is allocated by the
 * library and released with rsn_free().                                     */

/* replaces huffman.Compress([]byte) []byte            huffman.go:299 */
RSN_API int rsn_huffman_compress(const uint8_t *in, size_t n, uint8_t **out, size_t *out_n);
/* replaces huffman.Decompress([]byte) []byte          huffman.go:327 */
RSN_API int rsn_huffman_decompress(const uint8_t *in, size_t n, uint8_t **out, size_t *out_n);
/* replaces lz.CompressAsync([]byte, bool, int) []byte lzss.go:109
 * (the engine path: Writer.Write lzss.go:53-57).  window <= 0 = unbounded
 * search buffer (lzss.go:125).  The progress-bar argument has no equivalent. */
RSN_API int rsn_lzss_compress(const uint8_t *in, size_t n, int64_t window, uint8_t **out, size_t *out_n);
/* replaces lz.Compress([]byte, bool, int) []byte      lzss.go:224 -- the older synchronous encoder.
 * FOR SMALL INPUTS ONLY: it is the reference's O(n * window) loop (O(n^2) for window <= 0) on the calling
 * thread; inputs above 64 MiB (above 1 MiB for window <= 0 or > 65536) return RSN_ERR_LIMIT instead of
 * blocking a cgo call for hours (RSN_LEGACY_NO_LIMIT=1 lifts the bound).
 * Not on the .rsn path (the engine calls CompressAsync) and not accelerated: a host-side
 * restatement for API completeness, quirks included (every-second-byte FindReverse :425-431,
 * offsets computed from the unsliced buffer :249-257, `<=` token threshold :272).  Needs no device. */
RSN_API int rsn_lzss_compress_legacy(const uint8_t *in, size_t n, int64_t window, uint8_t **out, size_t *out_n);
/* replaces lz.Decompress([]byte, bool) []byte         lzss.go:323 */
RSN_API int rsn_lzss_decompress(const uint8_t *in, size_t n, uint8_t **out, size_t *out_n);

/* Batch form for independent chunks (one .rsn segment per chunk, as
 * engine.CompressFiles produces one file per input, engine.go:150-154).  By default the batch
 * stays on the calling thread's device; with RSN_BATCH_DEVICES=<G>|all the chunks are dealt out
 * over G visible devices -- chunk k -> worker k mod G, worker w on device (calling thread's
 * device + w) mod visible.  On each device chunk k+1's upload, chunk k's encode and chunk k-1's
 * download run at once.  Nothing is exchanged between devices.  Each outs[i] equals what
 * rsn_huffman_compress() returns for ins[i]; on any error every outs[i] is NULL.
 * (RSN_BATCH_WORKERS, RSN_BATCH_KEEP_MIB: see rsn_api.hip / INTEGRATION.md.) */
RSN_API int rsn_huffman_compress_batch(size_t n_chunks, const uint8_t *const *ins, const size_t *lens,
                               uint8_t **outs, size_t *out_lens);

/* ONE stream from `shards` slices of ONE input (intra-file sharding): per-slice histograms are summed, one tree and one
 * header are built, every slice is encoded at its exact bit offset (the format has a single front pad, huffman.go:245-255) by a
 * worker of its own, and the pieces are stitched on the way down.  The result is byte for byte rsn_huffman_compress(in, n).
 * shards <= 0: RSN_HUFF_SHARDS, else one per device used.  Worker w runs on device (calling thread's + w mod D) mod visible,
 * D = RSN_BATCH_DEVICES (default 1: the workers share the caller's device).  rsn_huffman_compress itself takes this path
 * when RSN_HUFF_SHARDS > 1 is set in the environment.                                                                    */
RSN_API int rsn_huffman_compress_sharded(const uint8_t *in, size_t n, int shards, uint8_t **out, size_t *out_n);

/* ---- device-resident entry points --------------------------------------
 * d_in / d_out are HIP device pointers on the calling thread's device; stream
 * is a hipStream_t (NULL = the thread's own stream).  The call returns after
 * the result size is known on the host; d_out is complete once `stream` has
 * been synchronised (the calls below synchronise it before returning).
 * d_out must be 16-byte aligned and hold rsn_*_bound() bytes; [d_in, d_in+n)
 * and [d_out, d_out+out_cap) must not overlap (RSN_ERR_ARG).                */
RSN_API size_t rsn_huffman_compress_bound(size_t n);
RSN_API size_t rsn_lzss_compress_bound(size_t n);
RSN_API int rsn_huffman_compress_dev(const void *d_in, size_t n, void *d_out, size_t out_cap, size_t *out_n, void *stream);
/* The decoded size is only known after the header is parsed.  When the buffer is too small -- or
 * d_out is NULL / out_cap 0, the size query -- the call returns RSN_ERR_CAPACITY, sets the error
 * string, and stores in *out_n a capacity that WOULD suffice (the exact size rounded up to 16, plus
 * 16: not the exact size); call again with a buffer of at least that many bytes, the second call
 * returns RSN_OK and the exact size.  rsn_lzss_decompress_dev and the two compress_dev calls follow
 * the same contract.  On RSN_ERR_FORMAT and on RSN_ERR_CAPACITY the contents of d_out are unspecified (the decoders
 * write while they validate, and what fits a too-small buffer may have been written before the total is known; nothing
 * is ever written outside [d_out, d_out + out_cap)).  (The Huffman query with d_out NULL is answered from the header's counts alone,
 * without touching the payload: a foreign stream whose payload decodes to more than its header
 * announces reports the larger need on the call that follows.)                                  */
RSN_API int rsn_huffman_decompress_dev(const void *d_in, size_t n, void *d_out, size_t out_cap, size_t *out_n, void *stream);
RSN_API int rsn_lzss_compress_dev(const void *d_in, size_t n, int64_t window, void *d_out, size_t out_cap, size_t *out_n, void *stream);
RSN_API int rsn_lzss_decompress_dev(const void *d_in, size_t n, void *d_out, size_t out_cap, size_t *out_n, void *stream);

/* ---- measurement --------------------------------------------------------
 * When enabled, every kernel launch of the calling thread is bracketed by HIP
 * events on the launch stream; rsn_prof_get() reports per-kernel totals since
 * the last rsn_prof_reset().                                               */
typedef struct {
    char name[48];
    uint64_t launches;
    double total_ms;
} rsn_prof_entry;
RSN_API void rsn_prof_enable(int on);
RSN_API void rsn_prof_reset(void);
RSN_API int rsn_prof_get(rsn_prof_entry *entries, int cap); /* returns the number of entries */

/* Introspection used by the parity tests: the code table the encoder builds for
 * `in` (device buffer not needed; runs the histogram on the device, the tree on
 * the host).  Arrays hold `cap` entries in printCodes DFS order (huffman.go:110);
 * returns the symbol count or a negative error. */
RSN_API int64_t rsn_huffman_table(const uint8_t *in, size_t n, uint32_t *runes, uint64_t *freqs,
                          uint64_t *codes, uint8_t *lens, size_t cap);

/* ---- host-side helpers (no device needed) --------------------------------
 * The part of the Huffman codec that stays on the host by design: the Go-exact
 * tree (buildTree huffman.go:58-103 with container/heap order), the codes
 * (printCodes :110-127) and the textual header (:312-318 / decodeTree :196-227).
 * Exposed so that the host logic can be checked on a machine without a GPU.   */

/* (rune,count) pairs in any order -> codes in printCodes DFS order plus the
 * header this library writes (ascending rune, '\\' never last).  Returns the
 * symbol count, or a negative error.  header may be NULL.                      */
RSN_API int64_t rsn_huffman_plan(const uint32_t *runes, const uint64_t *counts, size_t n_syms,
                         uint32_t *out_runes, uint64_t *out_codes, uint8_t *out_lens,
                         uint8_t *header, size_t header_cap, size_t *header_len);
/* decodeTree's scan of a header (bytes before "\\\n").  Returns the number of
 * distinct symbols (ascending rune), or a negative error where the reference
 * would index out of range.                                                   */
RSN_API int64_t rsn_huffman_parse_header(const uint8_t *header, size_t n, uint32_t *runes, uint64_t *counts, size_t cap);

/* Where rsn_huffman_compress_sharded cuts `in` into slices: cuts[0] = 0 < ... < cuts[S] = n, every cut on a rune start of Go's
 * decoding of the whole input (huffman.go:309: a UTF-8 sequence is never split).  Returns S (<= shards; short inputs get fewer
 * slices), or a negative error; cuts must hold shards + 1 entries.                                                           */
RSN_API int64_t rsn_huffman_slice_cuts(const uint8_t *in, size_t n, int shards, size_t *cuts, size_t cap);

#ifdef __cplusplus
}
#endif
#endif
