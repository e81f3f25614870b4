// PCIe probe behind DESIGN.md section 5's host-buffer numbers: what a pageable / registered / pinned host buffer costs each way,
// and how much an upload on one thread overlaps a download on another.   hipcc -O2 -o /tmp/pcie_probe scripts/pcie_probe.cpp -pthread
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <sys/mman.h>

static double now() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
#define CK(x) do { hipError_t err_ = (x); if (err_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(err_)); exit(1); } } while (0)

int main(int argc, char **argv) {
    const size_t n = (size_t)(argc > 1 ? atoi(argv[1]) : 256) << 20;
    void *d_a, *d_b;
    CK(hipMalloc(&d_a, n)); CK(hipMalloc(&d_b, n));
    hipStream_t s1, s2; CK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
    void *pg_in = nullptr, *pg_out = nullptr;
    posix_memalign(&pg_in, 2 << 20, n); posix_memalign(&pg_out, 2 << 20, n);
    madvise(pg_in, n, MADV_HUGEPAGE); madvise(pg_out, n, MADV_HUGEPAGE);
    memset(pg_in, 1, n); memset(pg_out, 2, n);
    void *pin_out; double t = now(); CK(hipHostMalloc(&pin_out, n, hipHostMallocDefault)); printf("hipHostMalloc %zu MiB: %.2f ms\n", n >> 20, now() - t);
    t = now(); memset(pin_out, 3, n); printf("  first touch: %.2f ms\n", now() - t);
    for (int rep = 0; rep < 3; rep++) {
        t = now(); CK(hipMemcpyAsync(d_a, pg_in, n, hipMemcpyHostToDevice, s1)); CK(hipStreamSynchronize(s1));
        double a = now() - t;
        t = now(); CK(hipMemcpyAsync(pg_out, d_b, n, hipMemcpyDeviceToHost, s2)); CK(hipStreamSynchronize(s2));
        double b = now() - t;
        t = now(); CK(hipMemcpyAsync(pin_out, d_b, n, hipMemcpyDeviceToHost, s2)); CK(hipStreamSynchronize(s2));
        double c = now() - t;
        t = now(); CK(hipMemcpyAsync(d_a, pin_out, n, hipMemcpyHostToDevice, s1)); CK(hipStreamSynchronize(s1));
        double d = now() - t;
        printf("H2D pageable %.2f ms (%.1f GB/s) | D2H pageable %.2f ms (%.1f GB/s) | D2H pinned %.2f ms (%.1f GB/s) | H2D pinned %.2f ms (%.1f GB/s)\n",
               a, n / a / 1e6, b, n / b / 1e6, c, n / c / 1e6, d, n / d / 1e6);
    }
    // register the pageable input in place
    for (int rep = 0; rep < 2; rep++) {
        t = now(); CK(hipHostRegister(pg_in, n, hipHostRegisterDefault)); double r = now() - t;
        t = now(); CK(hipMemcpyAsync(d_a, pg_in, n, hipMemcpyHostToDevice, s1)); CK(hipStreamSynchronize(s1)); double c = now() - t;
        t = now(); CK(hipHostUnregister(pg_in)); double u = now() - t;
        printf("hipHostRegister %.2f ms, H2D from it %.2f ms (%.1f GB/s), unregister %.2f ms\n", r, c, n / c / 1e6, u);
    }
    // overlap: upload on one thread, download on another
    auto both = [&](void *dst_host, const char *what) {
        for (int rep = 0; rep < 3; rep++) {
            t = now();
            std::thread th([&] { CK(hipMemcpyAsync(dst_host, d_b, n, hipMemcpyDeviceToHost, s2)); CK(hipStreamSynchronize(s2)); });
            CK(hipMemcpyAsync(d_a, pg_in, n, hipMemcpyHostToDevice, s1)); CK(hipStreamSynchronize(s1));
            th.join();
            double a = now() - t;
            printf("H2D pageable || D2H %s: %.2f ms for 2 x %zu MiB (%.1f GB/s summed)\n", what, a, n >> 20, 2 * n / a / 1e6);
        }
    };
    both(pg_out, "pageable");
    both(pin_out, "pinned");
    // staged upload: CPU memcpy pageable -> pinned ring, DMA from the ring, 2 threads copying
    {
        const size_t CH = 8u << 20; void *ring; CK(hipHostMalloc(&ring, 4 * CH, hipHostMallocDefault)); memset(ring, 0, 4 * CH);
        for (int rep = 0; rep < 3; rep++) {
            hipEvent_t ev[4]; for (auto &e : ev) CK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
            t = now();
            for (size_t o = 0, k = 0; o < n; o += CH, k++) {
                const size_t m = n - o < CH ? n - o : CH; const int slot = (int)(k & 3);
                if (k >= 4) CK(hipEventSynchronize(ev[slot]));
                memcpy((char *)ring + slot * CH, (char *)pg_in + o, m);
                CK(hipMemcpyAsync((char *)d_a + o, (char *)ring + slot * CH, m, hipMemcpyHostToDevice, s1));
                CK(hipEventRecord(ev[slot], s1));
            }
            CK(hipStreamSynchronize(s1));
            double a = now() - t;
            printf("H2D staged through a pinned ring (one copying thread): %.2f ms (%.1f GB/s)\n", a, n / a / 1e6);
            for (auto &e : ev) CK(hipEventDestroy(e));
        }
    }
    return 0;
}
