// what a process pays before its first kernel: hipInit + the first hipMalloc + a stream + one empty kernel (the code object's load)
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
static double now() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
__global__ void k_nop(int *p) { if (p && threadIdx.x == 12345) *p = 1; }
int main() {
    const double t0 = now();
    hipInit(0); const double t1 = now();
    int n = 0; hipGetDeviceCount(&n); hipSetDevice(0); const double t2 = now();
    void *p; hipMalloc(&p, 1 << 20); const double t3 = now();
    hipStream_t s; hipStreamCreateWithFlags(&s, hipStreamNonBlocking); const double t4 = now();
    k_nop<<<1, 64, 0, s>>>((int *)p); hipStreamSynchronize(s); const double t5 = now();
    printf("hipInit %.1f ms, device %.1f, first hipMalloc %.1f, stream %.1f, first kernel (+ code object) %.1f: total %.1f ms\n", t1 - t0, t2 - t1, t3 - t2, t4 - t3, t5 - t4, t5 - t0);
    return 0;
}
