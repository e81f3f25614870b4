// huff_pathmap.h -- maps between the alternative parses of neighbouring subsequences, composed along the lanes (device code).
// A Huffman payload is one bit string without an index; a lane that decodes the subsequence [lo, lo + S) learns where its first codeword
// starts from the lane before.  On periodic data a wrong start can stay a wrong parse for as long as the period lasts -- a second PHASE --
// and passing corrections on lane by lane (or block by block, launch by launch) then takes as many rounds as the stretch has lanes.  So a
// lane keeps up to four (start -> exit, count) entries, one per start a predecessor may hand it, and the true path is the composition
// of the lanes' maps: huff_small.hip (one launch, up to 32 blocks) and huff_decode.hip's k_dec_phase (any size, a few launches).
#pragma once

#include <cstdint>

namespace rsn {

// "entry j of the first lane -> entry to[j] of the lane behind the last, c[j] symbols on the way"; to[j] == 7: no such path
struct PathMap { uint32_t to, c0, c1, c2, c3; };
__device__ __forceinline__ uint32_t pm_to(const PathMap &m, uint32_t j) { return j < 4 ? (m.to >> (3 * j)) & 7 : 7u; }
__device__ __forceinline__ uint32_t pm_c(const PathMap &m, uint32_t j) { return j == 0 ? m.c0 : j == 1 ? m.c1 : j == 2 ? m.c2 : m.c3; }
__device__ __forceinline__ PathMap pm_then(const PathMap &a, const PathMap &b) {         // a's lanes first, then b's
    PathMap r;
    uint32_t t[4], c[4];
#pragma unroll
    for (int j = 0; j < 4; j++) { const uint32_t mid = pm_to(a, j); t[j] = pm_to(b, mid); c[j] = pm_c(a, j) + (mid < 4 ? pm_c(b, mid) : 0u); }
    r.to = t[0] | t[1] << 3 | t[2] << 6 | t[3] << 9; r.c0 = c[0]; r.c1 = c[1]; r.c2 = c[2]; r.c3 = c[3];
    return r;
}
constexpr uint32_t PM_ID = 0 | 1 << 3 | 2 << 6 | 3 << 9;
__device__ __forceinline__ PathMap pm_shfl_up(const PathMap &m, int d) {
    PathMap r; r.to = __shfl_up(m.to, d, 64); r.c0 = __shfl_up(m.c0, d, 64); r.c1 = __shfl_up(m.c1, d, 64); r.c2 = __shfl_up(m.c2, d, 64); r.c3 = __shfl_up(m.c3, d, 64);
    return r;
}

}  // namespace rsn
