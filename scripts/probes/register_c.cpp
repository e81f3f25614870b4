#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
static double now() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main() {
    const size_t n = (size_t)1 << 30, piece = (size_t)64 << 20;
    char *a = (char *)malloc(n); memset(a, 1, n);
    hipSetDevice(0);
    for (int rep = 0; rep < 2; rep++) {
        printf("register 64 MiB:");
        for (int k = 0; k < 4; k++) {
            char *p = (char *)(((uintptr_t)a + k * piece + 4095) & ~(uintptr_t)4095);
            double t0 = now(); hipError_t e = hipHostRegister(p, piece - 4096, 0); double t1 = now();
            hipHostUnregister(p); double t2 = now();
            printf(" %.2f ms (rc %d) / unregister %.2f ms;", t1 - t0, (int)e, t2 - t1);
        }
        printf("\n");
    }
    return 0;
}
