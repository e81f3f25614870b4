// How many blocks of a given LDS size and width share a CU on this part?  (hipOccupancyMaxActiveBlocksPerMultiprocessor, and a timing check.)
#include <hip/hip_runtime.h>
#include <cstdio>
template <int S, int T> __global__ __launch_bounds__(T) void k(int *out, int iters) {
    __shared__ int s[S / 4];
    for (int i = threadIdx.x; i < S / 4; i += T) s[i] = i;
    __syncthreads();
    int a = 0;
    for (int it = 0; it < iters; it++) a += s[(threadIdx.x * 33 + it * 7) % (S / 4)];
    if (a == 123456789) out[0] = a;
}
template <int S, int T> void probe() {
    int nb = 0;
    hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, k<S, T>, T, 0);
    printf("LDS %6d B, %4d threads: %d blocks per CU (%d wavefronts)\n", S, T, nb, nb * T / 64);
}
int main() {
    hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
    printf("%s: %d CUs, sharedMemPerBlock %zu, maxSharedMemoryPerMultiProcessor %zu, regsPerBlock %d, maxThreadsPerMultiProcessor %d\n", p.name, p.multiProcessorCount,
           p.sharedMemPerBlock, p.maxSharedMemoryPerMultiProcessor, p.regsPerBlock, p.maxThreadsPerMultiProcessor);
    probe<38016, 256>(); probe<50856, 384>(); probe<62400, 512>(); probe<69000, 576>(); probe<76488, 640>(); probe<53808, 1024>(); probe<57096, 1024>();
    probe<72976, 1024>(); probe<74000, 1024>(); probe<75816, 1024>(); probe<78000, 1024>(); probe<80000, 1024>(); probe<81920, 1024>();
    probe<22000, 256>(); probe<32768, 256>(); probe<40960, 256>(); probe<65536, 256>(); probe<81920, 256>(); probe<16384, 64>();
    return 0;
}
