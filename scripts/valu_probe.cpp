// What does a wave64 integer VALU instruction cost a gfx950 SIMD when 1, 2, 4 or 8 wavefronts share it?
// (k_match_chain's counters say 3.4 SIMD-cycles per VALU instruction at 8 wavefronts per SIMD: is that the pipe's limit or latency?)
// hipcc --offload-arch=gfx950 -O3 -o valu_probe valu_probe.cpp
#include <hip/hip_runtime.h>
#include <cstdio>
#pragma clang diagnostic ignored "-Wunused-value"
#include <cstdint>
#define DPP_XOR1 0xB1
template <int KIND> __global__ __launch_bounds__(1024) void k(uint32_t *out, int iters, uint32_t seed) {
    uint32_t a = threadIdx.x * 2654435761u + seed, b = a ^ 0x9E3779B9u, c = a + 77u, d = b * 3u, e = a >> 3, f = b + 5u, g = c ^ d, h = d + 11u;
    unsigned long long q = ((unsigned long long)a << 32) | b, r = ((unsigned long long)c << 32) | d;
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int u = 0; u < 16; u++) {
            if (KIND == 0) { a += b; c ^= d; e += f; g ^= h; b += c; d ^= e; f += g; h ^= a; }                                  // 8 independent-ish 32-bit adds / xors
            if (KIND == 1) { a = a < b ? c : a + 1; c = c < d ? e : c + 1; e = e < f ? g : e + 1; g = g < h ? a : g + 1; }       // cmp + cndmask + add (3 VALU each, 4 chains)
            if (KIND == 2) { a = max(a, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)a, DPP_XOR1, 0xF, 0xF, true)) + 1; c = max(c, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)c, DPP_XOR1, 0xF, 0xF, true)) + 1;
                             e = max(e, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)e, DPP_XOR1, 0xF, 0xF, true)) + 1; g = max(g, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)g, DPP_XOR1, 0xF, 0xF, true)) + 1; }   // max_dpp + add
            if (KIND == 3) { a = __builtin_amdgcn_alignbyte(a, b, c); c = __builtin_amdgcn_alignbyte(c, d, e); e = __builtin_amdgcn_alignbyte(e, f, g); g = __builtin_amdgcn_alignbyte(g, h, a); }
            if (KIND == 4) { q += r; r ^= q >> 7; }                                                                              // 64-bit add (2 VALU), shift (1 b64), xor (2)
            if (KIND == 5) { a += (q == r) ? 1u : 2u; q += a; c += (q < r) ? 3u : 1u; r += c; }                                // 64-bit compares
            if (KIND == 6) { a = a * 3u + b; c = c * 5u + d; e = e * 7u + f; g = g * 9u + h; }                                  // v_mad_u32_u24 / mul_lo
        }
    }
    if ((a ^ b ^ c ^ d ^ e ^ f ^ g ^ h ^ (uint32_t)q ^ (uint32_t)r) == 0x12345678u) out[0] = a;
}
template <int KIND> void run(const char *name, int valu_per_iter, uint32_t *d) {   // valu_per_iter: counted in the ISA of the loop (16 unrolled bodies)
    for (int threads : {256, 512, 1024}) for (int blocks_per_cu : {1, 2}) {
        if (threads != 1024 && blocks_per_cu == 2) continue;
        const int iters = 4000;
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        k<KIND><<<256 * blocks_per_cu, threads>>>(d, 10, 1);
        hipEventRecord(e0);
        k<KIND><<<256 * blocks_per_cu, threads>>>(d, iters, 1);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        const int waves_per_simd = threads / 64 * blocks_per_cu / 4;
        const double insts = (double)iters * valu_per_iter * waves_per_simd;          // per SIMD
        printf("%-28s %d wavefront(s) per SIMD: %.3f ms, %.2f ns per wave-instruction per SIMD (x clock = cycles; 2.4 GHz: %.2f)\n", name, waves_per_simd, ms, ms * 1e6 / insts, ms * 1e6 / insts * 2.4);
    }
}
int main() {
    uint32_t *d; hipMalloc(&d, 64);
    run<0>("add / xor u32", 112, d);
    run<1>("cmp + cndmask + add", 192, d);
    run<2>("max_dpp + add", 128, d);
    run<3>("alignbyte", 64, d);
    run<4>("u64: lshl_add, lshr, xor", 94, d);
    run<5>("u64 compares + cndmask", 144, d);
    run<6>("v_mad_u64_u32", 64, d);
    return 0;
}
