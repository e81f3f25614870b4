// codecs.h -- device-resident codec entry points shared between translation units.
#pragma once

#include <functional>

#include "huff_host.h"
#include "lzss_legacy.h"
#include "rsn_common.h"

namespace rsn {

size_t huff_compress_bound(size_t n);
size_t lzss_compress_bound(size_t n);

// All four: buffers are device pointers (16-byte aligned), the call synchronises
// `s` before returning, *out_n is the exact result size; on RSN_ERR_CAPACITY it
// is the capacity that would have sufficed.
int huff_encode_dev(Ctx &c, hipStream_t s, const uint8_t *d_in, size_t n, uint8_t *d_out, size_t out_cap, size_t *out_n,
                    HuffTree *tree_out, HuffCodes *codes_out);
// the same in two halves, for one stream out of several slices of the input (rsn_huffman_compress_sharded, rsn_api.hip)
struct HuffSlice { uint32_t tile = 0, n_tiles = 0; uint32_t *d_tile_hist = nullptr; uint16_t *d_smask = nullptr; bool ascii = true; std::vector<HuffSym> syms; };
int huff_slice_hist(Ctx &c, hipStream_t s, const uint8_t *d_in, size_t n, HuffSlice &sl);
int huff_slice_emit(Ctx &c, hipStream_t s, const uint8_t *d_in, size_t n, const HuffSlice &sl, const HuffTree &tree, const HuffCodes &codes, bool flat,
                    const std::string &hdr, unsigned long long base_bits, unsigned long long slice_bits, uint8_t *d_out);
bool huff_flat_code(const HuffTree &tree, const HuffCodes &codes);
// A host-buffer decode in SLICES (rsn_api.hip: the upload is still running while the first slices decode, and their bytes go down while
// the later ones decode): the decoder asks for its input as it needs it and announces its output as it becomes final.  The stream is one
// bit string without a block index (huffman.go:258-297), so a slice starts exactly where its predecessor's last codeword ended -- the
// hand-over is that bit position and the output offset, nothing else.
struct SliceStream {
    std::function<bool(size_t)> need_in;            // returns once d_in[0, bytes) is on the device; false: give up (the call fails)
    std::function<bool(size_t, size_t)> have_out;   // d_out[off, off + len) is final; false: give up
    std::function<size_t()> in_so_far;              // how much of d_in is on the device now (does not wait; may be empty)
    size_t slice_bytes = (size_t)32 << 20;          // payload bytes per slice (a multiple of 8 KiB: whole blocks of subsequences)
};
int huff_decode_dev(Ctx &c, hipStream_t s, const uint8_t *d_in, size_t n, uint8_t *d_out, size_t out_cap, size_t *out_n, const SliceStream *st = nullptr);
int lzss_encode_dev(Ctx &c, hipStream_t s, const uint8_t *d_in, size_t n, int64_t window, uint8_t *d_out, size_t out_cap, size_t *out_n);
// the same in SECTIONS as the input lands (a host-buffer call, rsn_api.hip): returns 1 when the input is not for it -- something in it needs
// an escape, or the window is not one the sections take -- and the caller encodes it whole.  slice_bytes = positions per section.
int lzss_encode_sliced(Ctx &c, hipStream_t s, const uint8_t *d_in, size_t n, int64_t window, uint8_t *d_out, size_t out_cap, size_t *out_n, const SliceStream &st);
// host buffers of at most 64 KiB, byte alphabets (huff_small.hip): two launches / one launch, no copy command; 1 = not for this path.
// *out: the result in the context's pinned staging (valid until the thread's next call), for the caller to copy
int huff_small_compress(Ctx &c, const uint8_t *in, size_t n, const uint8_t **out, size_t *out_n);
int huff_small_decompress(Ctx &c, const uint8_t *in, size_t n, const uint8_t **out, size_t *out_n);
// host buffers of at most 2 KiB (lzss_small.hip): one launch of one block each way, no copy command; 1 = not for this path
int lzss_small_compress(Ctx &c, const uint8_t *in, size_t n, int64_t window, const uint8_t **out, size_t *out_n);
int lzss_small_decompress(Ctx &c, const uint8_t *in, size_t n, const uint8_t **out, size_t *out_n);
int lzss_decode_dev(Ctx &c, hipStream_t s, const uint8_t *d_in, size_t n, uint8_t *d_out, size_t out_cap, size_t *out_n);

// the decoder in SLICES as the stream lands (a host-buffer call, rsn_api.hip): 1 = not a stream for it (a 5C, huge tokens): the caller decodes it whole;
// RSN_ERR_CAPACITY: it expands beyond out_cap (what the caller's sample promised)
int lzss_decode_sliced(Ctx &c, hipStream_t s, const uint8_t *d_in, size_t n, uint8_t *d_out, size_t out_cap, size_t *out_n, const SliceStream &st);

// exclusive scan of n counts on the stream (huff_encode.hip); *total (may be null) receives the sum; in and out must not overlap
int scan_u64(Ctx &c, hipStream_t s, const char *name, const unsigned long long *in, unsigned long long *out, uint32_t n, unsigned long long *total);

}  // namespace rsn
