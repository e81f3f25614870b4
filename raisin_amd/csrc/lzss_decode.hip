// lzss_decode.hip -- LZSS decode for gfx950 (MI355X).
//
// Replaces lz.Decompress (compressor/lz/lzss.go:323-364) and
// DecodeOpeningSymbols (:391-406).
//
//   L1 k_lzd_count / k_lzd_expand   parse "<ptr,len>" tokens in parallel ('<' never
//        occurs in literals: EncodeOpeningSymbols maps it to FF, lzss.go:373-377), scan
//        the output lengths, then give every escaped-stream byte a SOURCE: itself for a
//        literal, position-ptr for a byte produced by a token.
//   L2 k_lzd_jump     pointer jumping src[p] = src[src[p]] until every byte points at a
//        literal (a token may copy bytes that were themselves produced by a token: the
//        reference resolves that by running serially, lzss.go:349-353).
//   L3 k_lzd_gather   fetch the literal each byte resolved to.
//   L4 k_une_*        DecodeOpeningSymbols: an escape byte 5C consumes the next byte, so a
//        byte is escaped iff the run of 5C immediately before it has odd length; run
//        parities are combined per lane, per block and across blocks, then count/scan/write.
// Streams the reference would panic on (ptr past the start, len > ptr, malformed token)
// return RSN_ERR_FORMAT.
#include "codecs.h"

namespace rsn {

__global__ void k_scan_u64(const unsigned long long *in, unsigned long long *out, uint32_t n, unsigned long long *total);

constexpr int ZB = 256;
constexpr int ZTILE = ZB * 16;          // compressed bytes per block, 16 per lane
constexpr int MAXTOK = 24;              // longest token accepted: "<4294967295,4294967295>" is 23 bytes

struct Tok { uint32_t ptr, len, tl; bool ok; };

// parse the token that starts at in[j] == '<'
__device__ __forceinline__ Tok parse_tok(const uint8_t *__restrict__ in, size_t n, size_t j) {
    Tok t{0, 0, 0, false};
    size_t k = j + 1;
    unsigned long long v = 0; int nd = 0;
    while (k < n && nd < 10 && in[k] >= '0' && in[k] <= '9') { v = v * 10 + (in[k] - '0'); k++; nd++; }
    if (nd == 0 || k >= n || in[k] != ',' || v > 0xFFFFFFFFull) return t;
    t.ptr = (uint32_t)v; k++;
    v = 0; nd = 0;
    while (k < n && nd < 10 && in[k] >= '0' && in[k] <= '9') { v = v * 10 + (in[k] - '0'); k++; nd++; }
    if (nd == 0 || k >= n || in[k] != '>' || v > 0xFFFFFFFFull) return t;
    t.len = (uint32_t)v; t.tl = (uint32_t)(k + 1 - j); t.ok = true;
    return t;
}

// first position >= s that is not inside a token which started before s
__device__ __forceinline__ size_t skip_open_token(const uint8_t *__restrict__ in, size_t n, size_t s, int *err) {
    for (int back = 1; back < MAXTOK && (size_t)back <= s; back++) {
        const uint8_t b = in[s - back];
        if (b == '>') return s;
        if (b == '<') {
            const Tok t = parse_tok(in, n, s - back);
            if (!t.ok) { *err = 1; return s; }
            const size_t end = s - back + t.tl;
            return end > s ? end : s;
        }
    }
    return s;
}

__global__ __launch_bounds__(ZB) void k_lzd_count(const uint8_t *__restrict__ in, size_t n, unsigned long long *__restrict__ blk_len, int *__restrict__ err) {
    __shared__ unsigned long long part[ZB / 64];
    const size_t s = (size_t)blockIdx.x * ZTILE + threadIdx.x * 16;
    unsigned long long mine = 0;
    if (s < n) {
        int e = 0;
        size_t pos = skip_open_token(in, n, s, &e);
        const size_t lim = min(s + 16, n);
        while (pos < lim) {
            if (in[pos] == '<') {
                const Tok t = parse_tok(in, n, pos);
                if (!t.ok) { e = 1; pos++; continue; }
                mine += t.len; pos += t.tl;
            } else { mine++; pos++; }
        }
        if (e) atomicOr(err, 1);
    }
    for (int d = 32; d; d >>= 1) mine += __shfl_down(mine, d);
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = mine;
    __syncthreads();
    if (threadIdx.x == 0) blk_len[blockIdx.x] = part[0] + part[1] + part[2] + part[3];
}

__global__ __launch_bounds__(ZB) void k_lzd_expand(const uint8_t *__restrict__ in, size_t n, const unsigned long long *__restrict__ blk_off,
                                                   uint32_t *__restrict__ src, uint8_t *__restrict__ esc, int *__restrict__ err) {
    __shared__ unsigned long long wsum[ZB / 64];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const size_t s = (size_t)blockIdx.x * ZTILE + tid * 16;
    const size_t lim = min(s + 16, n);
    unsigned long long mine = 0;
    size_t pos0 = s;
    int e = 0;
    if (s < n) {
        pos0 = skip_open_token(in, n, s, &e);
        size_t pos = pos0;
        while (pos < lim) {
            if (in[pos] == '<') { const Tok t = parse_tok(in, n, pos); if (!t.ok) { pos++; continue; } mine += t.len; pos += t.tl; }
            else { mine++; pos++; }
        }
    }
    unsigned long long incl = mine;
    for (int d = 1; d < 64; d <<= 1) { const unsigned long long y = __shfl_up(incl, d); if (lane >= d) incl += y; }
    if (lane == 63) wsum[wv] = incl;
    __syncthreads();
    unsigned long long o = blk_off[blockIdx.x] + incl - mine;
    for (int k = 0; k < wv; k++) o += wsum[k];
    if (s >= n) return;
    size_t pos = pos0;
    while (pos < lim) {
        if (in[pos] == '<') {
            const Tok t = parse_tok(in, n, pos);
            if (!t.ok) { pos++; continue; }
            // absolutePointer = len(out) - pointer must be >= 0 and the slice must end inside out (lzss.go:349-350)
            if (t.ptr > o || t.len > t.ptr) { e = 1; }
            else for (uint32_t k = 0; k < t.len; k++) src[o + k] = (uint32_t)(o + k - t.ptr);
            o += t.len; pos += t.tl;
        } else { src[o] = (uint32_t)o; esc[o] = in[pos]; o++; pos++; }
    }
    if (e) atomicOr(err, 1);
}

__global__ __launch_bounds__(ZB) void k_lzd_jump(uint32_t *__restrict__ src, uint32_t E, int *__restrict__ changed) {
    const uint32_t stride = gridDim.x * ZB;
    bool any = false;
    for (uint32_t p = blockIdx.x * ZB + threadIdx.x; p < E; p += stride) {
        const uint32_t a = src[p];
        if (a == p) continue;
        const uint32_t b = src[a];
        if (b != a) { src[p] = b; any = true; }
    }
    if (__syncthreads_or(any) && threadIdx.x == 0) *changed = 1;
}

__global__ __launch_bounds__(ZB) void k_lzd_gather(const uint32_t *__restrict__ src, uint8_t *__restrict__ esc, uint32_t E) {
    const uint32_t stride = gridDim.x * ZB;
    for (uint32_t p = blockIdx.x * ZB + threadIdx.x; p < E; p += stride) {
        const uint32_t a = src[p];
        if (a != p) esc[p] = esc[a];      // roots (literals) are never written here
    }
}

// ------------------------------------------------------------------ L4: unescape
// per-block summary: is the whole block 5C, and the parity of its trailing 5C run
__global__ __launch_bounds__(ZB) void k_une_summary(const uint8_t *__restrict__ esc, uint32_t E, uint8_t *__restrict__ summ) {
    __shared__ uint32_t s_last;          // highest index in the block that is not 5C, +1 (0 = none)
    if (threadIdx.x == 0) s_last = 0;
    __syncthreads();
    const uint32_t base = blockIdx.x * ZTILE;
    const uint32_t s = base + threadIdx.x * 16;
    uint32_t last = 0;
    for (uint32_t k = 0; k < 16 && s + k < E; k++) if (esc[s + k] != 0x5C) last = threadIdx.x * 16 + k + 1;
    if (last) atomicMax(&s_last, last);
    __syncthreads();
    if (threadIdx.x == 0) {
        const uint32_t len = min((uint32_t)ZTILE, E - base);
        summ[blockIdx.x] = (uint8_t)((s_last == 0 ? 2 : 0) | ((len - s_last) & 1));   // bit1: all 5C; bit0: trailing-run parity
    }
}

// in_par[b] = parity of the 5C run that ends right before block b.  The (all-5C, parity)
// summaries combine associatively, so one block of 1024 lanes scans them: every lane folds a
// contiguous chunk, lane 0 chains the 1024 chunk summaries, every lane replays its chunk.
__global__ __launch_bounds__(1024) void k_une_carry(const uint8_t *__restrict__ summ, uint32_t n_blk, uint8_t *__restrict__ in_par) {
    __shared__ uint8_t s_chunk[1024];
    __shared__ uint8_t s_in[1024];
    const uint32_t per = (n_blk + 1023) / 1024;
    const uint32_t b0 = threadIdx.x * per, b1 = min(b0 + per, n_blk);
    uint32_t all = 2, par = 0;                       // summary of this lane's chunk, same encoding as summ[]
    for (uint32_t b = b0; b < b1; b++) {
        const uint32_t v = summ[b];
        if (v & 2) par ^= v & 1; else { all = 0; par = v & 1; }
    }
    s_chunk[threadIdx.x] = (uint8_t)(all | par);
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t p = 0;
        for (int k = 0; k < 1024; k++) { s_in[k] = (uint8_t)p; const uint32_t v = s_chunk[k]; p = (v & 2) ? p ^ (v & 1) : (v & 1); }
    }
    __syncthreads();
    uint32_t p = s_in[threadIdx.x];
    for (uint32_t b = b0; b < b1; b++) {
        in_par[b] = (uint8_t)p;
        const uint32_t v = summ[b];
        p = (v & 2) ? p ^ (v & 1) : (v & 1);
    }
}

// shared front end of count/write: the escape state at the start of this lane's 16 bytes
__device__ __forceinline__ uint32_t lane_in_parity(const uint8_t *__restrict__ esc, uint32_t E, uint32_t base, uint32_t blk_par, uint8_t *s_sum) {
    const int tid = threadIdx.x;
    const uint32_t s = base + tid * 16;
    uint32_t last = 0, len = 0;
    for (uint32_t k = 0; k < 16 && s + k < E; k++) { len++; if (esc[s + k] != 0x5C) last = k + 1; }
    s_sum[tid] = (uint8_t)((last == 0 ? 2 : 0) | ((len - last) & 1));
    __syncthreads();
    uint32_t par = 0;
    int q = tid - 1;
    for (; q >= 0; q--) { const uint32_t v = s_sum[q]; par ^= v & 1; if (!(v & 2)) break; }
    if (q < 0) par ^= blk_par;
    return par;
}

__global__ __launch_bounds__(ZB) void k_une_count(const uint8_t *__restrict__ esc, uint32_t E, const uint8_t *__restrict__ in_par,
                                                  unsigned long long *__restrict__ blk_len) {
    __shared__ uint8_t s_sum[ZB];
    __shared__ uint32_t part[ZB / 64];
    const uint32_t base = blockIdx.x * ZTILE;
    uint32_t st = lane_in_parity(esc, E, base, in_par[blockIdx.x], s_sum);
    const uint32_t s = base + threadIdx.x * 16;
    uint32_t cnt = 0;
    for (uint32_t k = 0; k < 16 && s + k < E; k++) {
        const uint8_t v = esc[s + k];
        if (v == 0x5C && !st) st = 1; else { st = 0; cnt++; }     // lzss.go:395-403
    }
    for (int d = 32; d; d >>= 1) cnt += __shfl_down(cnt, d);
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = cnt;
    __syncthreads();
    if (threadIdx.x == 0) blk_len[blockIdx.x] = (unsigned long long)part[0] + part[1] + part[2] + part[3];
}

__global__ __launch_bounds__(ZB) void k_une_write(const uint8_t *__restrict__ esc, uint32_t E, const uint8_t *__restrict__ in_par,
                                                  const unsigned long long *__restrict__ blk_off, uint8_t *__restrict__ out) {
    __shared__ uint8_t s_sum[ZB];
    __shared__ uint32_t wsum[ZB / 64];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const uint32_t base = blockIdx.x * ZTILE;
    const uint32_t st0 = lane_in_parity(esc, E, base, in_par[blockIdx.x], s_sum);
    const uint32_t s = base + tid * 16;
    uint32_t st = st0, cnt = 0;
    for (uint32_t k = 0; k < 16 && s + k < E; k++) { const uint8_t v = esc[s + k]; if (v == 0x5C && !st) st = 1; else { st = 0; cnt++; } }
    uint32_t incl = cnt;
    for (int d = 1; d < 64; d <<= 1) { const uint32_t y = __shfl_up(incl, d); if (lane >= d) incl += y; }
    if (lane == 63) wsum[wv] = incl;
    __syncthreads();
    uint32_t pre = 0;
    for (int k = 0; k < wv; k++) pre += wsum[k];
    uint8_t *o = out + blk_off[blockIdx.x] + pre + incl - cnt;
    st = st0;
    for (uint32_t k = 0; k < 16 && s + k < E; k++) {
        const uint8_t v = esc[s + k];
        if (v == 0xFF && !st) *o++ = 0x3C;                 // EncodedOpening -> '<'
        else if (v == 0x5C && !st) st = 1;                 // escape marker: emits nothing
        else { st = 0; *o++ = v; }
    }
}

// ======================================================================= host side
int lzss_decode_dev(Ctx &c, hipStream_t s, const uint8_t *d_in, size_t n, uint8_t *d_out, size_t out_cap, size_t *out_n) {
    *out_n = 0;
    if (n == 0) return RSN_OK;
    if (((uintptr_t)d_in & 15) || ((uintptr_t)d_out & 15)) return c.fail(RSN_ERR_ARG, "lzss: device buffers must be 16-byte aligned");
    if (n >= (1ull << 32) - 65536) return c.fail(RSN_ERR_LIMIT, "lzss: compressed input too large for one call");
    void *p; int rc;
    const uint32_t n_cb = (uint32_t)ceil_div(n, ZTILE);
    rc = dev_buf(c, 13, ((size_t)n_cb * 2 + 4) * 8, &p); if (rc) return rc;
    unsigned long long *d_blen = (unsigned long long *)p, *d_boff = d_blen + n_cb, *d_btot = d_boff + n_cb;
    int *d_flag = (int *)(d_btot + 1);                                    // [0] error, [1] changed
    void *hp; rc = pinned_buf(c, 64, &hp); if (rc) return rc;
    unsigned long long *h64 = (unsigned long long *)hp;
    volatile int *hflag = (volatile int *)(h64 + 1);
    RSN_HIP(hipMemsetAsync(d_flag, 0, 8, s));
    RSN_LAUNCH("lzss_dec_count", k_lzd_count, dim3(n_cb), dim3(ZB), 0, s, d_in, n, d_blen, d_flag);
    RSN_LAUNCH("lzss_dec_scan", k_scan_u64, dim3(1), dim3(1024), 0, s, d_blen, d_boff, n_cb, d_btot);
    RSN_HIP(hipMemcpyAsync(h64, d_btot, 8, hipMemcpyDeviceToHost, s));
    RSN_HIP(hipMemcpyAsync((void *)hflag, d_flag, 8, hipMemcpyDeviceToHost, s));
    RSN_HIP(hipStreamSynchronize(s));
    if (hflag[0]) return c.fail(RSN_ERR_FORMAT, "lzss: malformed \"<ptr,len>\" token");
    if (h64[0] >= (1ull << 32) - 65536) return c.fail(RSN_ERR_LIMIT, "lzss: decoded stream too large for one call");
    const uint32_t E = (uint32_t)h64[0];
    if (E == 0) return RSN_OK;
    rc = dev_buf(c, 14, (size_t)E * 4 + 64, &p); if (rc) return rc;
    uint32_t *d_src = (uint32_t *)p;
    rc = dev_buf(c, 15, (size_t)E + 64, &p); if (rc) return rc;
    uint8_t *d_esc = (uint8_t *)p;
    RSN_LAUNCH("lzss_dec_expand", k_lzd_expand, dim3(n_cb), dim3(ZB), 0, s, d_in, n, d_boff, d_src, d_esc, d_flag);
    RSN_HIP(hipMemcpyAsync((void *)hflag, d_flag, 8, hipMemcpyDeviceToHost, s));
    RSN_HIP(hipStreamSynchronize(s));   // src[] is only safe to chase once every token has been validated
    if (hflag[0]) return c.fail(RSN_ERR_FORMAT, "lzss: back-reference outside the decoded data (reference: slice bounds out of range, lzss.go:350)");
    const uint32_t grid = (uint32_t)std::min<size_t>(ceil_div(E, ZB), 8192);
    for (int round = 0;; round++) {
        if (round > 40) return c.fail(RSN_ERR_DEVICE, "lzss: pointer jumping did not converge");
        RSN_HIP(hipMemsetAsync(d_flag + 1, 0, 4, s));
        RSN_LAUNCH("lzss_dec_jump", k_lzd_jump, dim3(grid), dim3(ZB), 0, s, d_src, E, d_flag + 1);
        RSN_HIP(hipMemcpyAsync((void *)hflag, d_flag, 8, hipMemcpyDeviceToHost, s));
        RSN_HIP(hipStreamSynchronize(s));
        if (hflag[0]) return c.fail(RSN_ERR_FORMAT, "lzss: back-reference outside the decoded data (reference: slice bounds out of range, lzss.go:350)");
        if (!hflag[1]) break;
    }
    RSN_LAUNCH("lzss_dec_gather", k_lzd_gather, dim3(grid), dim3(ZB), 0, s, d_src, d_esc, E);
    // ---- unescape
    const uint32_t n_ub = (uint32_t)ceil_div(E, ZTILE);
    rc = dev_buf(c, 16, ((size_t)n_ub * 2 + 2) * 8 + (size_t)n_ub * 2 + 64, &p); if (rc) return rc;
    unsigned long long *d_ulen = (unsigned long long *)p, *d_uoff = d_ulen + n_ub, *d_utot = d_uoff + n_ub;
    uint8_t *d_summ = (uint8_t *)(d_utot + 2), *d_inpar = d_summ + n_ub;
    RSN_LAUNCH("lzss_une_summary", k_une_summary, dim3(n_ub), dim3(ZB), 0, s, d_esc, E, d_summ);
    RSN_LAUNCH("lzss_une_carry", k_une_carry, dim3(1), dim3(1024), 0, s, d_summ, n_ub, d_inpar);
    RSN_LAUNCH("lzss_une_count", k_une_count, dim3(n_ub), dim3(ZB), 0, s, d_esc, E, d_inpar, d_ulen);
    RSN_LAUNCH("lzss_dec_scan", k_scan_u64, dim3(1), dim3(1024), 0, s, d_ulen, d_uoff, n_ub, d_utot);
    RSN_HIP(hipMemcpyAsync(h64, d_utot, 8, hipMemcpyDeviceToHost, s));
    RSN_HIP(hipStreamSynchronize(s));
    const size_t total = (size_t)h64[0];
    *out_n = total;
    if (!d_out || total > out_cap) { *out_n = round_up(total, 16) + 16; return c.fail(RSN_ERR_CAPACITY, "lzss: output needs %zu bytes, buffer holds %zu", total, out_cap); }
    RSN_LAUNCH("lzss_une_write", k_une_write, dim3(n_ub), dim3(ZB), 0, s, d_esc, E, d_inpar, d_uoff, d_out);
    RSN_HIP(hipStreamSynchronize(s));
    return RSN_OK;
}

}  // namespace rsn
