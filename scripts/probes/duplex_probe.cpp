// duplex in pieces: does H2D || D2H keep its summed rate when both sides go in 64 MiB pieces with a sync after each, from malloc'ed memory?
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#define CK(x) do { hipError_t err_ = (x); if (err_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(err_)); exit(1); } } while (0)
static double now() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
__global__ void k_touch(const uint4 *a, uint4 *b, size_t n16) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n16; i += (size_t)gridDim.x * blockDim.x) { uint4 v = a[i]; v.x ^= 1; b[i] = v; }
}
int main(int argc, char **argv) {
    const size_t n = (size_t)1 << 30, piece = (size_t)(argc > 1 ? atoi(argv[1]) : 64) << 20;
    void *d_a, *d_b; CK(hipMalloc(&d_a, n)); CK(hipMalloc(&d_b, n));
    hipStream_t s1, s2; CK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
    char *in = (char *)malloc(n), *out = (char *)malloc(n);
    memset(in, 1, n); memset(out, 2, n);
    for (int rep = 0; rep < 3; rep++) {
        double t = now();
        CK(hipMemcpyAsync(d_a, in, n, hipMemcpyHostToDevice, s1)); CK(hipStreamSynchronize(s1));
        double a = now() - t; t = now();
        CK(hipMemcpyAsync(out, d_b, n, hipMemcpyDeviceToHost, s2)); CK(hipStreamSynchronize(s2));
        double b = now() - t;
        printf("alone, one copy: H2D %.2f ms (%.1f GB/s), D2H %.2f ms (%.1f GB/s)\n", a, n / a / 1e6, b, n / b / 1e6);
    }
    for (int rep = 0; rep < 3; rep++) {
        double t = now();
        std::thread th([&] { for (size_t o = 0; o < n; o += piece) { CK(hipMemcpyAsync(out + o, (char *)d_b + o, piece, hipMemcpyDeviceToHost, s2)); CK(hipStreamSynchronize(s2)); } });
        for (size_t o = 0; o < n; o += piece) { CK(hipMemcpyAsync((char *)d_a + o, in + o, piece, hipMemcpyHostToDevice, s1)); CK(hipStreamSynchronize(s1)); }
        double up = now() - t;
        th.join();
        double a = now() - t;
        printf("both at once in %zu MiB pieces: up done %.2f ms, all done %.2f ms (%.1f GB/s summed)\n", piece >> 20, up, a, 2 * n / a / 1e6);
    }
    for (int rep = 0; rep < 3; rep++) {
        double t = now();
        std::thread th([&] { CK(hipMemcpyAsync(out, d_b, n, hipMemcpyDeviceToHost, s2)); CK(hipStreamSynchronize(s2)); });
        CK(hipMemcpyAsync(d_a, in, n, hipMemcpyHostToDevice, s1)); CK(hipStreamSynchronize(s1));
        th.join();
        double a = now() - t;
        printf("both at once, one copy each: %.2f ms (%.1f GB/s summed)\n", a, 2 * n / a / 1e6);
    }
    // ... and with a third thread that runs kernels over device memory and synchronises its own stream all the while (what a codec does)
    hipStream_t s3; CK(hipStreamCreateWithFlags(&s3, hipStreamNonBlocking));
    void *d_c, *d_d; CK(hipMalloc(&d_c, n)); CK(hipMalloc(&d_d, n));
    for (int mode = 0; mode < 2; mode++)
    for (int rep = 0; rep < 3; rep++) {
        volatile bool stop = false; long launches = 0;
        double t = now();
        std::thread th3([&] { while (!stop) { if (mode == 0) k_touch<<<2048, 256, 0, s3>>>((const uint4 *)d_c, (uint4 *)d_d, (size_t)(64 << 20) / 16); else CK(hipMemsetAsync(d_d, 0, 64, s3)); CK(hipStreamSynchronize(s3)); launches++; } });
        std::thread th([&] { for (size_t o = 0; o < n; o += piece) { CK(hipMemcpyAsync(out + o, (char *)d_b + o, piece, hipMemcpyDeviceToHost, s2)); CK(hipStreamSynchronize(s2)); } });
        for (size_t o = 0; o < n; o += piece) { CK(hipMemcpyAsync((char *)d_a + o, in + o, piece, hipMemcpyHostToDevice, s1)); CK(hipStreamSynchronize(s1)); }
        double up = now() - t;
        th.join();
        double a = now() - t;
        stop = true; th3.join();
        printf("both at once in %zu MiB pieces + a third thread (%s, %ld rounds): up done %.2f ms, all done %.2f ms (%.1f GB/s summed)\n", piece >> 20, mode == 0 ? "64 MiB kernel + sync" : "memset + sync", launches, up, a, 2 * n / a / 1e6);
    }
    return 0;
}
