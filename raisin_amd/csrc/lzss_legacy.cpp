// lzss_legacy.cpp -- lz.Compress (compressor/lz/lzss.go:224-316), the reference's older synchronous
// encoder.  It is API surface only: the engine and the .rsn path use
// CompressAsync (lzss.go:53-57), nothing here runs on the GPU and nothing on the hot path calls
// it.  The function is one long byte-at-a-time state machine over a growing search buffer and is
// not worth a kernel (serial by construction); it is offered so that every exported function of
// package lz has a drop-in counterpart, quirks included:
//   * a match is opened by FindReverse, which looks at every SECOND byte from the end of the whole
//     search buffer (the loop decrements twice, lzss.go:425-431) and ignores the window;
//   * it is extended by a leftmost search of the grown string inside the LAST `window` bytes of the
//     search buffer, but the pointer is then computed from the UNSLICED length (lzss.go:249-257):
//     once the buffer is longer than the window the offsets are wrong and the stream decodes to
//     something else -- the reference's own data (ai/data.json) shows lossless=false for such files;
//   * a reference is written when len("<p,n>") <= n  (lzss.go:272: `>` where CompressAsync has `<`).
// The bytes of the open match join the search buffer only when the match closes.
#include <algorithm>
#include <cstdint>
#include <string>

#include "lzss_legacy.h"

namespace rsn {

namespace {
// EncodeOpeningSymbols, lzss.go:369-389 (foundEscape is never set: the third branch is dead)
std::string escape_opening(const uint8_t *in, size_t n) {
    std::string fc;
    fc.reserve(n + n / 16 + 16);
    for (size_t i = 0; i < n; i++) {
        uint8_t v = in[i];
        if (v == '<') v = 0xFF;
        else if (v == 0xFF || v == 0x5C) fc.push_back((char)0x5C);
        fc.push_back((char)v);
    }
    return fc;
}

void put_token_or_bytes(std::string &out, size_t pointer, size_t offset, const std::string &pending) {
    const std::string enc = "<" + std::to_string(pointer) + "," + std::to_string(offset) + ">";
    if (enc.size() > pending.size()) out += pending;          // shouldAdd = false (lzss.go:271-275)
    else out += enc;                                           // len(pending) > -1 always holds (:277)
}
}  // namespace

void lzss_compress_legacy_host(const uint8_t *in, size_t n, int64_t window, std::string &out) {
    const std::string fc = escape_opening(in, n);
    std::string search;                                        // searchBuffer
    std::string pending;                                       // checkBytesToAdd
    search.reserve(fc.size());
    out.clear();
    bool open = false;                                         // checkNextByte
    size_t pointer = 0, offset = 0;                            // checkStartPointer, checkOffset
    for (const char ch : fc) {
        bool found = false;
        size_t index = 0;
        if (!open) {
            for (size_t k = search.size(); k >= 1; k = k >= 2 ? k - 2 : 0)         // indices len-1, len-3, ... (lzss.go:425-431)
                if (search[k - 1] == ch) { found = true; index = k - 1; break; }
        } else {
            size_t cut = 0;
            if (window > 0 && search.size() > (size_t)window) cut = search.size() - (size_t)window;
            pending.push_back(ch);
            const auto first = search.begin() + (std::ptrdiff_t)cut;
            const auto hit = std::search(first, search.end(), pending.begin(), pending.end());
            pending.pop_back();
            if (hit != search.end()) { found = true; index = (size_t)(hit - first); }       // relative to the sliced buffer
        }
        if (found) {
            pointer = search.size() - index;                                                // from the unsliced length
            offset = open ? offset + 1 : 1;
            open = true;
            pending.push_back(ch);
            continue;                                                                       // the byte joins the buffer when the match closes
        }
        if (open) {
            put_token_or_bytes(out, pointer, offset, pending);
            search += pending;
            pending.clear();
            open = false; pointer = 0; offset = 0;
        }
        out.push_back(ch);
        search.push_back(ch);
    }
    if (open) put_token_or_bytes(out, pointer, offset, pending);
}

}  // namespace rsn
