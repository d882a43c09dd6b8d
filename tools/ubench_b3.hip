// Microbenchmark: BLAKE3 compression passes per SIMD on gfx950 with nothing else in the kernel (message words in
// registers, no memory traffic).  Gives the VALU floor of the hash for a given number of waves per SIMD.
// Build: hipcc -O3 --offload-arch=gfx950 -I znippy_amd/csrc tools/ubench_b3.hip -o tools/ubench_b3.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include "blake3_dev.h"
template <int WPS>
__global__ __launch_bounds__(256, WPS) void k(uint32_t *out, uint32_t seed, int passes) {
    uint32_t cv[8], m[16];
    for (int i = 0; i < 8; i++) cv[i] = threadIdx.x * 7 + i + seed;
    for (int i = 0; i < 16; i++) m[i] = threadIdx.x * 13 + i * seed;
#pragma unroll 1
    for (int p = 0; p < passes; p++) {
        b3::compress(cv, m, p, 0, 64, 0);
        m[p & 15] ^= cv[0];
    }
    uint32_t x = 0;
    for (int i = 0; i < 8; i++) x ^= cv[i];
    out[blockIdx.x * 256 + threadIdx.x] = x;
}
template <int WPS>
void run(int passes_per_simd) {
    uint32_t *d;
    int grid = 256 * WPS;  // WPS blocks of 4 waves per CU -> WPS waves per SIMD
    (void)hipMalloc(&d, grid * 256 * 4);
    int passes = passes_per_simd / WPS;
    hipEvent_t t0, t1;
    (void)hipEventCreate(&t0); (void)hipEventCreate(&t1);
    k<WPS><<<grid, 256>>>(d, 1, passes);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(t0);
    k<WPS><<<grid, 256>>>(d, 2, passes);
    (void)hipEventRecord(t1);
    (void)hipDeviceSynchronize();
    float ms;
    (void)hipEventElapsedTime(&ms, t0, t1);
    printf("waves/SIMD=%d  %d passes/SIMD: %.3f ms  -> %.1f ns per pass per SIMD; C2 hash (325.6 passes/SIMD) = %.3f ms\n", WPS,
           passes * WPS, ms, ms * 1e6 / (passes * WPS), ms / (passes * WPS) * 325.6);
    (void)hipFree(d);
}
int main() {
    run<1>(3200); run<2>(3200); run<3>(3200 / 3 * 3); run<4>(3200); run<5>(3200); run<8>(3200);
    return 0;
}
