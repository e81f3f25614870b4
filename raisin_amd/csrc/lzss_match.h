// lzss_match.h -- the two match searches that look at EVERY position of a strip (lzss_match.hip), as the encoder's driver
// (lzss_encode.hip) launches them: the packed sweep k_match2 (any window) and the bigram-bucket search k_match_hash (windows up to
// HWMAX).  The chain walk (lzss_encode.hip) evaluates only the positions a greedy chain lands on and hands strips back to these.
#pragma once

#include "rsn_common.h"

namespace rsn {

constexpr int MATCH_STRIP = 16384;      // positions per match block
constexpr uint32_t KEY_UNKNOWN = 0xFFFFFFFFu;                     // chain mode: no key evaluated for this position (the parse redoes the strip if the true chain lands here)

struct MatchArgs {
    const uint8_t *fc; uint32_t E; uint32_t W; uint32_t DW;   // DW = diagonals per wave
    uint32_t *keys;
    const uint32_t *only;                                      // non-null: sweep only the strips flagged here (k_match_hash's hand-backs), a flag per MATCH_STRIP positions
    uint32_t strip;                                            // positions per block: MATCH_STRIP, or a smaller multiple of 64 that divides it
};
constexpr int MW2_WIDE = 16;            // ... and where only a handful of strips are swept: a strip's latency is what the call waits for
constexpr int MW2 = 4;                  // wavefronts per block of the packed sweep: fewer, longer diagonal ranges (less pipeline fill)

constexpr int HT = 4096;                 // positions per block
constexpr int HTH = 512;                 // threads per block
constexpr int HSH = 9;                   // log2(HTH): entries of a bucket are ordered by staged offset >> HSH
constexpr int HWMAX = 4096;              // largest window this path takes
constexpr int HLMAX = 256;               // longest common prefix examined before the strip is handed back
constexpr int HNB = 8192;                // buckets
constexpr int H_STAGE = HWMAX + HT + HLMAX + 32;
constexpr uint32_t H_ITER_CAP = (HT / HTH) * 384;   // trips per lane before the strip is handed back (the sweep costs about as much as 250 trips per position)
static_assert(HWMAX + HT <= 8192, "an entry keeps the staged offset in 13 bits");
static_assert(MATCH_STRIP % HT == 0, "a strip is a whole number of hash tiles");
static_assert((1 << HSH) == HTH && (HWMAX + HT) % HTH == 0 && (HNB / 2) % HTH == 0, "round structure");

struct HashArgs { const uint8_t *fc; uint32_t E; uint32_t W; uint32_t *keys; uint32_t *heavy; const uint32_t *only; };

#ifdef __HIPCC__
// eight staged bytes from any byte offset: three aligned dwords and two v_alignbyte.  (gfx950's LDS takes unaligned accesses, and
// the compiler emits ONE ds_read_b64 for an align-1 load -- but an unaligned wave-instruction is replayed: the chain walk went from
// 37 to 70 ms with it, k_match_hash from 20 to 27.)
__device__ __forceinline__ unsigned long long lds_load8(const uint32_t *sw, uint32_t rel) {
    const uint32_t q = rel >> 2;
    const uint32_t w0 = sw[q], w1 = sw[q + 1], w2 = sw[q + 2];
    return (unsigned long long)__builtin_amdgcn_alignbyte(w1, w0, rel) | ((unsigned long long)__builtin_amdgcn_alignbyte(w2, w1, rel) << 32);   // v_alignbyte uses rel[1:0]
}
#endif

// the launches (lzss_match.hip); shmem2: k_match2's dynamic LDS for this window
int lzss_launch_match2(Ctx &c, hipStream_t s, const MatchArgs &m2, uint32_t n_blocks, size_t shmem2, bool wide = false);   // wide: MW2_WIDE wavefronts a block (m2.DW and shmem2 worked out for it)
int lzss_launch_match_hash(Ctx &c, hipStream_t s, const HashArgs &h);

}  // namespace rsn
