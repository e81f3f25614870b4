// lzss_match.hip -- LZSS match search at every position of a strip, for gfx950 (MI355X): what lz.Compress's inner loop
// (compressor/lz/lzss.go:109-184, the window scan at :166-184) computes for a position, for ALL positions of a 16 K strip at once.
// The encoder's fast path (lzss_encode.hip: k_match_chain) only evaluates the positions a greedy chain visits; these two kernels are
// what it falls back to for strips it gives up on (dense or periodic data), for windows above 4096, and under RSN_LZSS_ALLPOS (tests:
// an independent second formulation of the same keys).
#include "lzss_match.h"

namespace rsn {

// ------------------------------------------------------------------ E2: match search
#define RSN_DPP_WAVE_SHL1 0x130   // lane i <- lane i+1
#define RSN_DPP_WAVE_SHR1 0x138   // lane i <- lane i-1

// ------------------------------------------------------------------ E2': packed match search (lengths only)
// The sweep with two diagonals per lane packed in the 16-bit halves of one VGPR
// (v_pk_* arithmetic): lane l meets diagonals Dk+1+j (low half) and Dk+H+1+j (high half),
// j = t-63+l, so a wave's diagonal range takes half the steps.  Only the capped run length
// survives packing (the 16-bit maximum cannot carry the distance); the distance of the few
// positions that end up on the parse chain with a token-sized match is recovered afterwards
// (k_parse_mark), which costs far less than carrying it through every (position, diagonal) pair.
typedef unsigned short us2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ us2 as_us2(uint32_t v) { return __builtin_bit_cast(us2, v); }
__device__ __forceinline__ uint32_t as_u32(us2 v) { return __builtin_bit_cast(uint32_t, v); }

template <bool MASKED, bool CHECKED>
__device__ __forceinline__ void match2_chunk(const uint8_t *ybase, long long y0, uint32_t H, uint32_t E, uint32_t t0, uint32_t nsteps,
                                             uint32_t Hk_lo, uint32_t Hk_hi, int lane, uint32_t X2, uint32_t vCin, uint32_t &vCout,
                                             uint32_t &C2, uint32_t &D2, uint32_t &best2) {
    const us2 one = {1, 1};
#pragma unroll
    for (int k = 0; k < 64; k++) {
        const uint32_t t = t0 + k;
        if (!MASKED || t < nsteps) {
            uint32_t Ylo = ybase[-(int)t], Yhi = ybase[-(int)t - (int)H];
            if (CHECKED) {
                const long long y = y0 - (long long)t;
                if (y < 0 || y >= (long long)E) Ylo = 0x200;
                if (y - (long long)H < 0 || y - (long long)H >= (long long)E) Yhi = 0x200;
            }
            const uint32_t Y2 = Ylo | (Yhi << 16);
            const uint32_t cin = (uint32_t)__builtin_amdgcn_readlane((int)vCin, k);
            const uint32_t sh = (uint32_t)__builtin_amdgcn_update_dpp((int)cin, (int)C2, RSN_DPP_WAVE_SHL1, 0xF, 0xF, false);
            us2 m = __builtin_elementwise_sub_sat(one, as_us2(X2 ^ Y2));            // 1 where the bytes are equal
            if (MASKED) {
                const uint32_t j = t + (uint32_t)lane - 63u;                          // local diagonal index of this lane
                const uint32_t vm = (j < Hk_lo ? 0x0000FFFFu : 0u) | (j < Hk_hi ? 0xFFFF0000u : 0u);
                m = as_us2(as_u32(m) & vm);
            }
            D2 = as_u32(as_us2(D2) + one);                                            // per-half add: a wrapping low half must not carry
            uint32_t grown;                                                          // (run + 1) * eq in one v_pk_mad_u16
            asm("v_pk_mad_u16 %0, %1, %2, %2" : "=v"(grown) : "v"(sh), "v"(as_u32(m)));
            const us2 c = __builtin_elementwise_min(as_us2(grown), as_us2(D2));      // eq ? min(run+1, d) : 0
            C2 = as_u32(c);
            best2 = as_u32(__builtin_elementwise_max(as_us2(best2), c));
        }
        vCout = (uint32_t)__builtin_amdgcn_update_dpp((int)C2, (int)vCout, RSN_DPP_WAVE_SHR1, 0xF, 0xF, false);
    }
}

// (MW: wavefronts per block.  MW = 4 where the flagged strips fill the chip -- fewer, longer diagonal ranges, less pipeline fill; 16 for a
//  handful of strips, r06: a block's time is its position blocks in a row times the steps of a wavefront's diagonal range, 575 or 191)
template <int MW>
__global__ __launch_bounds__(MW * 64) void k_match2(MatchArgs a) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    const uint32_t W4 = a.DW * MW + 16;                      // lead-in of the staged region (+16: the high half of an odd DW reads one byte further)
    const uint32_t WUB = (a.W + 63) / 64 * 64;
    const uint32_t STRIP = a.strip;                                   // positions of this block (r05: MATCH_STRIP, or less where the strips are few -- a block's
                                                                      // time is (STRIP + W) / 64 position blocks in a row, and 64 blocks leave three CUs in four idle)
    const uint32_t RLEN = (STRIP + WUB + W4 + 15) & ~15u;
    const uint32_t H = (a.DW + 1) / 2;                                // diagonals per half
    uint8_t *s_b = smem;
    uint32_t *s_carry = reinterpret_cast<uint32_t *>(smem + RLEN);    // [MW][H] packed runs entering from the block above
    uint32_t *s_comb = s_carry + MW * H;                     // [2][MW][64]

    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    if (a.only && !a.only[(size_t)blockIdx.x * STRIP / MATCH_STRIP]) return;   // (the flags are per MATCH_STRIP positions)
    const long long b0 = (long long)blockIdx.x * STRIP;
    const long long r0 = b0 - (long long)W4;
    for (uint32_t i = tid; i < RLEN; i += MW * 64) {
        const long long p = r0 + i;
        s_b[i] = (p >= 0 && p < (long long)a.E) ? a.fc[p] : 0;
    }
    for (uint32_t i = tid; i < MW * H; i += MW * 64) s_carry[i] = 0;
    __syncthreads();
    {   // Shortcut for W-periodic stretches: a match on diagonal W that fills the whole window (or reaches the end of the stream)
        // cannot be beaten -- L <= W, and W is the largest distance, i.e. the leftmost occurrence.  If that holds for every position
        // of the strip (no mismatch fc[q] != fc[q-W] anywhere in [b0, b0+STRIP+W)), the search is skipped.
        const long long q_end = min(b0 + (long long)STRIP + (long long)a.W - 1, (long long)a.E);
        bool ok = b0 >= (long long)a.W;
        if (ok) for (long long q = b0 + tid; q < q_end; q += MW * 64) ok = ok && (s_b[q - r0] == s_b[q - (long long)a.W - r0]);
        if (__syncthreads_and(ok)) {
            for (long long p = b0 + tid; p < min(b0 + (long long)STRIP, (long long)a.E); p += MW * 64) {
                const uint32_t L = (uint32_t)min((long long)a.W, (long long)a.E - p);
                a.keys[p] = (L << 16) | a.W;
            }
            return;
        }
    }
    const uint32_t Dk = wv * a.DW;                                    // this wave: diagonals Dk+1 .. Dk+DWk
    const uint32_t DWk = Dk >= a.W ? 0 : min(a.DW, a.W - Dk);
    const uint32_t Hk_lo = min(H, DWk), Hk_hi = DWk > H ? DWk - H : 0; // valid local indices in each half
    const uint32_t nsteps = Hk_lo ? Hk_lo + 63 : 0;
    uint32_t *carry = s_carry + wv * H;
    const int nPB = (int)((STRIP + WUB) / 64);

    for (int pb = nPB - 1; pb >= 0; pb--) {
        const long long P0 = b0 + 64ll * pb;
        if (P0 >= (long long)a.E) continue;
        const long long p = P0 + lane;
        const uint32_t X = p < (long long)a.E ? (uint32_t)s_b[p - r0] : 0x300u;
        const uint32_t X2 = X | (X << 16);
        uint32_t best2 = 0, C2 = 0;
        // lane l meets local index j = t-63+l: diagonals Dk+1+j (low) and Dk+H+1+j (high); D2 holds both, one step behind
        const uint32_t d0 = Dk + (uint32_t)lane - 63u;                // diagonal of the low half before the first increment
        uint32_t D2 = (d0 & 0xFFFFu) | (((d0 + H) & 0xFFFFu) << 16);
        const long long y0 = P0 + 62 - (long long)Dk;                 // candidate index of the low half at t = 0
        const uint8_t *ybase = s_b + (y0 - r0);
        const bool checked = P0 < (long long)W4 || P0 + 63 >= (long long)a.E;
        for (uint32_t t0 = 0; t0 < nsteps; t0 += 64) {
            const uint32_t ci = t0 + lane;
            const uint32_t vCin = ci < Hk_lo ? carry[ci] : 0u;
            uint32_t vCout = 0;
            if (checked) match2_chunk<true, true>(ybase, y0, H, a.E, t0, nsteps, Hk_lo, Hk_hi, lane, X2, vCin, vCout, C2, D2, best2);
            else if (t0 >= 63 && t0 + 64 <= Hk_hi) match2_chunk<false, false>(ybase, y0, H, a.E, t0, nsteps, Hk_lo, Hk_hi, lane, X2, vCin, vCout, C2, D2, best2);
            else match2_chunk<true, false>(ybase, y0, H, a.E, t0, nsteps, Hk_lo, Hk_hi, lane, X2, vCin, vCout, C2, D2, best2);
            const uint32_t co = t0 - (uint32_t)lane;                  // value of step k sits in lane 63-k
            if (co < Hk_lo) carry[co] = vCout;
        }
        uint32_t *comb = s_comb + (pb & 1) * (MW * 64);
        comb[wv * 64 + lane] = max(best2 & 0xFFFFu, best2 >> 16);
        __syncthreads();
        if (wv == 0 && pb < (int)(STRIP / 64) && p < (long long)a.E) {
            uint32_t L = comb[lane];
#pragma unroll
            for (int w = 1; w < MW; w++) L = max(L, comb[w * 64 + lane]);
            a.keys[p] = L << 16;                                      // distance filled in later for chain positions that need it
        }
    }
}

// ------------------------------------------------------------------ E2'': bigram-bucket match search
// The sweeps above cost W compares per position whatever the data.  Here a block takes HT
// positions plus their window, groups every staged position by its first two bytes (counting
// sort into 8192 LDS buckets: first byte | low five bits of the second, the other three bits
// kept as a tag in the entry), and a position only examines the entries of its own bucket:
//   * every candidate start j with fc[j:j+2] == fc[i:i+2] is in that bucket, so
//     best = max over them of min(lcp(i,j), i-j) is exact whenever the answer is >= 2, and
//     the packed maximum (L<<16 | distance) also yields the leftmost occurrence;
//   * an answer of 0 or 1 is decided by whether ANY in-window entry of the 32 buckets that
//     share the first byte exists.
// Work is proportional to how often the bigram at i occurs in the window instead of W, which
// on text is ~1% of W.  The result is the same function as k_match2; input where the
// assumption fails (an lcp of HLMAX or more, or a lane running past H_ITER_CAP entries) flags
// its strip and k_match2 redoes exactly those strips.
__global__ __launch_bounds__(HTH) void k_match_hash(HashArgs a) {
    __shared__ __attribute__((aligned(16))) uint32_t sw[H_STAGE / 4];   // fc[r0, r0 + H_STAGE), zero outside the stream
    __shared__ uint32_t s_cur[HNB / 2];                                  // two 16-bit counters per word: counts, then starts, then ends
    __shared__ uint16_t s_list[HWMAX + HT];                              // staged offset | tag << 13, grouped by bucket
    __shared__ uint32_t s_part[HTH / 64];
    __shared__ uint16_t s_order[HT];                                     // the tile's positions, longest buckets first
    __shared__ uint32_t s_cls[4];
    __shared__ uint32_t s_heavy, s_next;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const uint32_t E = a.E, W = a.W;
    if (a.only && !a.only[blockIdx.x / (MATCH_STRIP / HT)]) return;     // fix-up round: only the strips where the chain met an unevaluated position
    const long long t0 = (long long)blockIdx.x * HT;
    const long long r0 = t0 - HWMAX;
    const uint8_t *sb = reinterpret_cast<const uint8_t *>(sw);
    for (uint32_t v = tid; v < H_STAGE / 16; v += HTH) {
        const long long P = r0 + 16ll * v;
        uint4 x = {0, 0, 0, 0};
        if (P >= 0 && P + 16 <= (long long)E) x = *reinterpret_cast<const uint4 *>(a.fc + P);
        else if (P + 16 > 0 && P < (long long)E) {
            uint32_t w[4] = {0, 0, 0, 0};
            for (int k = 0; k < 16; k++) { const long long q = P + k; if (q >= 0 && q < (long long)E) w[k >> 2] |= (uint32_t)a.fc[q] << (8 * (k & 3)); }
            x = {w[0], w[1], w[2], w[3]};
        }
        reinterpret_cast<uint4 *>(sw)[v] = x;
    }
    for (int i = tid; i < HNB / 2; i += HTH) s_cur[i] = 0;
    if (tid == 0) { s_heavy = 0; s_next = 0; }
    if (tid < 4) s_cls[tid] = 0;
    __syncthreads();

    if (t0 >= (long long)W) {   // W-periodic tile: L = min(W, E-p) at distance W for every position (see k_match2)
        bool ok = true;
        const uint32_t qn = (uint32_t)min((long long)(HT + HLMAX), (long long)E - t0);
        for (uint32_t q = tid; q < qn; q += HTH) ok = ok && sb[HWMAX + q] == sb[HWMAX + q - W];
        if (__syncthreads_and(ok)) {
            const long long q_end = min(t0 + (long long)HT + (long long)W - 1, (long long)E);
            for (long long q = t0 + HT + HLMAX + tid; q < q_end; q += HTH) ok = ok && a.fc[q] == a.fc[q - W];
            if (__syncthreads_and(ok)) {
                for (long long p = t0 + tid; p < min(t0 + (long long)HT, (long long)E); p += HTH)
                    a.keys[p] = ((uint32_t)min((long long)W, (long long)E - p) << 16) | W;
                return;
            }
        }
    }

    // ---- group the staged positions by bigram: candidates are the positions [lo, hi)
    const uint32_t rlo = (uint32_t)(max(0ll, t0 - (long long)W) - r0), rhi = (uint32_t)(min((long long)E - 1, t0 + (long long)HT) - r0);
    for (uint32_t rel = tid; rel < HWMAX + HT; rel += HTH) {
        if (rel < rlo || rel >= rhi) continue;
        const uint32_t h = ((uint32_t)sb[rel] << 5) | (sb[rel + 1] & 31u);
        atomicAdd(&s_cur[h >> 1], 1u << (16 * (h & 1)));
    }
    __syncthreads();
    {
        constexpr int PER = HNB / 2 / HTH;                               // counter words per thread
        uint32_t sum = 0;
        for (int k = 0; k < PER; k++) { const uint32_t x = s_cur[tid * PER + k]; sum += (x & 0xFFFF) + (x >> 16); }
        uint32_t incl = sum;
        for (int dd = 1; dd < 64; dd <<= 1) { const uint32_t y = __shfl_up(incl, dd); if (lane >= dd) incl += y; }
        if (lane == 63) s_part[wv] = incl;
        __syncthreads();
        uint32_t run = incl - sum;
        for (int k = 0; k < wv; k++) run += s_part[k];
        for (int k = 0; k < PER; k++) {
            const uint32_t x = s_cur[tid * PER + k], c0 = x & 0xFFFF, c1 = x >> 16;
            s_cur[tid * PER + k] = run | ((run + c0) << 16);
            run += c0 + c1;
        }
    }
    __syncthreads();
    // rounds of HTH consecutive offsets with a barrier between them: inside a bucket the entries end
    // up ordered by offset >> HSH, which is all the lower-bound search below needs
    for (uint32_t rel = tid; rel < HWMAX + HT; rel += HTH) {
        if (rel >= rlo && rel < rhi) {
            const uint32_t b1 = sb[rel + 1];
            const uint32_t h = ((uint32_t)sb[rel] << 5) | (b1 & 31u), sh = 16 * (h & 1);
            const uint32_t slot = (atomicAdd(&s_cur[h >> 1], 1u << sh) >> sh) & 0xFFFF;
            s_list[slot] = (uint16_t)(rel | ((b1 >> 5) << 13));
        }
        __syncthreads();
    }
    const uint16_t *ends = reinterpret_cast<const uint16_t *>(s_cur);    // ends[h]; the bucket starts at ends[h-1]

    // ---- every lane takes positions off a shared counter and walks their buckets.  One flat trip
    //      body (no continue): each lane reads at most one entry and makes at most one 8-byte
    //      compare per trip, so lanes in different states share every trip.
    const uint32_t npos = (uint32_t)min((long long)HT, (long long)E - t0);
    // Longest buckets first: a position whose bigram fills a bucket walks hundreds of trips, and
    // handed out last it would leave most of the block's lanes idle behind it.  Four size classes,
    // counted and scattered with one LDS atomic per wavefront and class.
    {
        constexpr int PP = HT / HTH;                                      // positions per lane
        uint32_t cls[PP];
#pragma unroll
        for (int k = 0; k < PP; k++) {
            const uint32_t kp = tid + k * HTH;
            uint32_t size = 0;
            if (kp < npos && kp + 1 < E - (uint32_t)t0) {
                const uint32_t b0 = sb[HWMAX + kp], b1 = sb[HWMAX + kp + 1], h = (b0 << 5) | (b1 & 31u);
                size = ends[h] - (h ? ends[h - 1] : 0);
            }
            cls[k] = kp >= npos ? 4u : size > 192 ? 0u : size > 64 ? 1u : size > 16 ? 2u : 3u;
#pragma unroll
            for (uint32_t c = 0; c < 4; c++) {
                const unsigned long long m = __ballot(cls[k] == c);
                if (lane == 0 && m) atomicAdd(&s_cls[c], (uint32_t)__builtin_popcountll(m));
            }
        }
        __syncthreads();
        const uint32_t c0 = s_cls[0], c1 = s_cls[1], c2 = s_cls[2];
        __syncthreads();
        if (tid == 0) { s_cls[0] = 0; s_cls[1] = c0; s_cls[2] = c0 + c1; s_cls[3] = c0 + c1 + c2; }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < PP; k++) {
#pragma unroll
            for (uint32_t c = 0; c < 4; c++) {
                const unsigned long long m = __ballot(cls[k] == c);
                uint32_t base = 0;
                if (lane == 0 && m) base = atomicAdd(&s_cls[c], (uint32_t)__builtin_popcountll(m));
                base = (uint32_t)__builtin_amdgcn_readfirstlane((int)base);
                if (cls[k] == c) s_order[base + (uint32_t)__builtin_popcountll(m & ((1ull << lane) - 1ull))] = (uint16_t)(tid + k * HTH);
            }
        }
        __syncthreads();
    }
    bool have = false, ext = false, bytemode = false, search = false;
    uint32_t ipos = 0, irel = 0, cur = 0, end = 0, bcur = 0, bend = 0, tag = 0, best = 0, prel = 0, off = 0, capE = 0, d = 0, iters = 0, slo = 0, shi = 0;
    uint32_t blk_lo = 0, blk_i = 0;                                      // offset >> HSH of the window start and of the position
    for (;;) {
        if (!have) {                                                      // the only divergent region of a trip
            if (s_heavy) break;
            const uint32_t kq = atomicAdd(&s_next, 1u);
            if (kq >= npos) break;
            const uint32_t kp = s_order[kq];
            ipos = (uint32_t)t0 + kp;
            irel = HWMAX + kp;
            const uint32_t b0 = sb[irel], b1 = sb[irel + 1];
            capE = E - ipos;
            best = 0; have = true; off = 0; prel = irel; d = 0;
            bcur = b0 ? ends[(b0 << 5) - 1] : 0; bend = ends[(b0 << 5) | 31];       // every staged position that starts with b0
            const uint32_t h = (b0 << 5) | (b1 & 31u);
            bytemode = capE < 2;
            cur = bytemode ? bcur : (h ? ends[h - 1] : 0); end = bytemode ? bend : ends[h]; tag = b1 >> 5;
            search = !bytemode && end - cur > 8;                          // long bucket: skip the entries before the window
            slo = cur; shi = end;
            blk_lo = (irel - W) >> HSH; blk_i = irel >> HSH;
        }
        if (++iters > H_ITER_CAP) { s_heavy = 1; break; }
        // ---- entry step (skipped while a compare is being extended): one entry of the bucket, or one bisection step
        const bool fetch = !ext;
        const bool rd = fetch && (search || cur < end);
        const uint32_t idx = search ? (slo + shi) >> 1 : cur;
        const uint32_t e = s_list[rd ? idx : 0];
        const uint32_t rel = e & 8191u;
        const bool s_step = rd && search, w_step = rd && !search;
        const uint32_t blk = rel >> HSH;
        const bool below = blk < blk_lo;
        slo = (s_step && below) ? idx + 1 : slo;
        shi = (s_step && !below) ? idx : shi;
        const bool s_done = s_step && slo >= shi;
        search = search && !s_done;
        cur = s_done ? slo : cur + (w_step ? 1u : 0u);
        const uint32_t dn = irel - rel;
        const bool inwin = dn - 1u < W;                                   // candidate start in [i-W, i)
        const bool byte_hit = w_step && bytemode && inwin;                // the byte occurs in the window: L = 1
        const bool past = w_step && !bytemode && blk > blk_i;             // the rest of the bucket starts after i
        const bool start = w_step && !bytemode && inwin && (e >> 13) == tag && dn > (best >> 16);   // same bigram, far enough back to beat the best
        best = byte_hit ? (1u << 16) : best;
        cur = (byte_hit || past) ? end : cur;
        prel = start ? rel : prel;
        d = start ? dn : d;
        off = start ? 2u : off;                                           // the bigram itself is known to match: compare from the third byte
        // ---- list exhausted -- noticed in the trip that took the last entry unless that entry starts a
        //      compare: L = 0 so far falls back to the first-byte range, anything else is final
        const bool exh = fetch && !start && !search && cur >= end;
        const bool to_byte = exh && !bytemode && best == 0;
        if (exh && !to_byte) { a.keys[ipos] = best; have = false; }
        cur = to_byte ? bcur : cur;
        end = to_byte ? bend : end;
        bytemode = bytemode || to_byte;
        // ---- compare step: eight more bytes of the current candidate
        const bool cmp = ext || start;
        const unsigned long long x = lds_load8(sw, prel + off) ^ lds_load8(sw, irel + off);
        const uint32_t n = x ? (uint32_t)__builtin_ctzll(x) >> 3 : 8u;
        const uint32_t cap = min(d, capE);                                // entirely inside the window, and inside the stream
        ext = cmp && n == 8 && off + 8 < cap;
        best = (cmp && !ext) ? max(best, (min(off + n, cap) << 16) | d) : best;     // longest, then farthest back (bytes.Index, lzss.go:419)
        off += ext ? 8u : 0u;
        if (off >= HLMAX) { s_heavy = 1; break; }
    }
    __syncthreads();
    if (tid == 0 && s_heavy) a.heavy[blockIdx.x / (MATCH_STRIP / HT)] = 1;
}

int lzss_launch_match2(Ctx &c, hipStream_t s, const MatchArgs &m2, uint32_t n_blocks, size_t shmem2, bool wide) {
    if (wide) {
        { const int rc = func_dyn_lds(c, reinterpret_cast<const void *>(k_match2<MW2_WIDE>), shmem2); if (rc) return rc; }
        RSN_LAUNCH("lzss_match", k_match2<MW2_WIDE>, dim3(n_blocks), dim3(MW2_WIDE * 64), shmem2, s, m2);
        return RSN_OK;
    }
    { const int rc = func_dyn_lds(c, reinterpret_cast<const void *>(k_match2<MW2>), shmem2); if (rc) return rc; }
    RSN_LAUNCH("lzss_match", k_match2<MW2>, dim3(n_blocks), dim3(MW2 * 64), shmem2, s, m2);
    return RSN_OK;
}

int lzss_launch_match_hash(Ctx &c, hipStream_t s, const HashArgs &h) {
    RSN_LAUNCH("lzss_match_hash", k_match_hash, dim3((uint32_t)ceil_div(h.E, HT)), dim3(HTH), 0, s, h);
    return RSN_OK;
}

}  // namespace rsn
