// Probe: does an LDS dword store / load at an address that is not a multiple of four do what the address says on this GPU?
// (gfx950; the answer decides whether k_dec_emit may write a lookup's bytes with one ds_write_b32 at a byte offset.)
// build: hipcc --offload-arch=gfx950 -O2 -o lds_unaligned lds_unaligned.cpp ; run: ./lds_unaligned
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <vector>

__global__ void k(uint8_t *out, uint32_t *rd) {
    __shared__ __attribute__((aligned(16))) uint8_t s[1024];
    const int t = threadIdx.x;
    for (int i = t; i < 1024; i += 64) s[i] = 0xEE;
    __syncthreads();
    const uint32_t addr = (uint32_t)(uintptr_t)s + 5u * t + 1u;           // 1, 6, 11, ... : every alignment
    const uint32_t v = 0x04030201u + 0x10101010u * (t & 7);
    asm volatile("ds_write_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" ::"v"(addr), "v"(v) : "memory");
    __syncthreads();
    for (int i = t; i < 1024; i += 64) out[i] = s[i];
    uint32_t r;
    asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(r) : "v"(addr) : "memory");
    rd[t] = r;
}

int main() {
    uint8_t *d; uint32_t *dr;
    hipMalloc(&d, 1024); hipMalloc(&dr, 256);
    k<<<1, 64>>>(d, dr);
    std::vector<uint8_t> h(1024); std::vector<uint32_t> hr(64);
    hipMemcpy(h.data(), d, 1024, hipMemcpyDeviceToHost); hipMemcpy(hr.data(), dr, 256, hipMemcpyDeviceToHost);
    int bad = 0, badr = 0;
    // lanes execute in one instruction: lane t's 4 bytes at 5t+1 .. 5t+4 (no overlap between lanes: stride 5)
    for (int t = 0; t < 64; t++) {
        const uint32_t v = 0x04030201u + 0x10101010u * (t & 7);
        for (int b = 0; b < 4; b++) if (h[5 * t + 1 + b] != (uint8_t)(v >> (8 * b))) bad++;
        if (h[5 * t] != 0xEE) bad++;
        if (hr[t] != v) badr++;
    }
    printf("unaligned ds_write_b32: %s (%d wrong bytes); unaligned ds_read_b32: %s (%d wrong)\n", bad ? "NOT byte-addressed" : "ok", bad, badr ? "NOT byte-addressed" : "ok", badr);
    return bad || badr;
}
