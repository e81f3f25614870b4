// Does a host-to-device copy on one stream proceed while a kernel runs on another?  (The host pipeline's question: rsn_api.hip piped_call.)
//   hipcc --offload-arch=gfx950 -O2 -o /tmp/cuk scripts/probes/copy_under_kernel.hip -lpthread && /tmp/cuk
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
__global__ void spin(unsigned long long ticks, unsigned long long *sink) {
    const unsigned long long t0 = __builtin_readcyclecounter();
    unsigned long long x = 0;
    while (__builtin_readcyclecounter() - t0 < ticks) x++;
    if (x == 12345 && sink) *sink = x;
}
static double now() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main() {
    const size_t N = (size_t)256 << 20, PIECE = (size_t)64 << 20;
    uint8_t *h_pinned, *h_paged, *d;
    CK(hipHostMalloc((void **)&h_pinned, N, hipHostMallocDefault));
    h_paged = (uint8_t *)aligned_alloc(4096, N); memset(h_paged, 1, N); memset(h_pinned, 2, N);
    CK(hipMalloc((void **)&d, N));
    hipStream_t sk, sc;
    CK(hipStreamCreateWithFlags(&sk, hipStreamNonBlocking));
    CK(hipStreamCreateWithFlags(&sc, hipStreamNonBlocking));
    // how many counter ticks is a millisecond?
    spin<<<1, 64, 0, sk>>>(1000, nullptr); CK(hipStreamSynchronize(sk));
    double t = now(); spin<<<1, 64, 0, sk>>>(100000000ull, nullptr); CK(hipStreamSynchronize(sk)); const double per_ms = 100000000.0 / (now() - t);
    printf("%.0f ticks per ms\n", per_ms);
    for (int blocks : {0, 1, 512, 4096}) for (int kind = 0; kind < 4; kind++) for (int threaded = 0; threaded < 2; threaded++) {
        const unsigned long long ticks = (unsigned long long)(per_ms * (blocks > 512 ? 2.5 : 20));     // 4096 blocks of 1024: 8 rounds on 256 CUs x 2
        if (blocks) spin<<<blocks, blocks == 1 ? 64 : 1024, 0, sk>>>(ticks, nullptr);
        const double t0 = now();
        auto copy = [&] {
            CK(hipSetDevice(0));
            uint8_t *src = kind == 0 ? h_pinned : h_paged;
            for (size_t at = 0; at < N; at += PIECE) {
                if (kind >= 2) CK(hipHostRegister(src + at, PIECE, hipHostRegisterDefault));
                CK(hipMemcpyAsync(d + at, src + at, PIECE, hipMemcpyHostToDevice, sc));
                CK(hipStreamSynchronize(sc));
                if (kind == 2) CK(hipHostUnregister(src + at));
            }
        };
        if (threaded) { std::thread th(copy); th.join(); } else copy();
        const double t_copy = now() - t0;
        CK(hipStreamSynchronize(sk));
        const double t1 = now();
        if (kind == 3) { for (size_t at = 0; at < N; at += PIECE) CK(hipHostUnregister(h_paged + at)); printf("  (4 pieces unregistered after the kernel in %.2f ms)\n", now() - t1); }
        printf("kernel of %4d blocks, %s%s: 256 MiB up in %6.2f ms, kernel done at %6.2f ms\n", blocks, kind == 0 ? "hipHostMalloc" : kind == 1 ? "pageable" : kind == 2 ? "registered and unregistered by piece" : "registered by piece, unregistered at the end",
               threaded ? ", copies from a second thread" : "", t_copy, now() - t0);
    }
    return 0;
}
