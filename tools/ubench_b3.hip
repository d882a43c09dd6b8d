// Microbenchmark: BLAKE3 compression passes per SIMD on gfx950.
//   variant 0: nothing else in the loop (message words in registers, no memory traffic) — the VALU floor;
//   variant 1: the message comes from LDS as in the fused kernel (four unaligned ds_read_b128 per block at a
//              per-lane position inside a 45-byte period, next block in flight during the compression);
//   variant 2: variant 1 + the four 16-byte global stores per block of the register-sourced row write.
// Build: hipcc -O3 --offload-arch=gfx950 -I znippy_amd/csrc tools/ubench_b3.hip -o tools/ubench_b3.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include "blake3_dev.h"
typedef uint32_t u4v __attribute__((ext_vector_type(4)));
typedef u4v __attribute__((aligned(1))) u4v_unaligned;
typedef __attribute__((address_space(3))) u4v_unaligned lds_u4;
template <int WPS, int VAR>
__global__ __launch_bounds__(256, WPS) void k(uint32_t *out, uint8_t *sink, uint32_t seed, int passes) {
    __shared__ __attribute__((aligned(16))) uint8_t win[4][6 * 608];
    uint32_t cv[8], m[16];
    for (int i = 0; i < 8; i++) cv[i] = threadIdx.x * 7 + i + seed;
    for (int i = 0; i < 16; i++) m[i] = threadIdx.x * 13 + i * seed;
    const uint32_t lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    for (uint32_t i = lane; i < 6 * 608; i += 64) win[w][i] = (uint8_t)(i * seed);
    __syncthreads();
    const uint8_t *Y = win[w] + (lane / 10 % 6) * 608 + 20;
    uint32_t r = (lane * 1024) % 45;
    u4v n0 = 0, n1 = 0, n2 = 0, n3 = 0;
    uint8_t *dst = sink + ((size_t)(blockIdx.x * 4 + w) * 64 + lane) * 1024;
    auto fetch = [&]() {
        const lds_u4 *q = (const lds_u4 *)(Y + r);
        n0 = q[0]; n1 = q[1]; n2 = q[2]; n3 = q[3];
        r += 19; if (r >= 45) r -= 45;
    };
    if (VAR) fetch();
#pragma unroll 1
    for (int p = 0; p < passes; p++) {
        if (VAR) {
            m[0] = n0.x; m[1] = n0.y; m[2] = n0.z; m[3] = n0.w; m[4] = n1.x; m[5] = n1.y; m[6] = n1.z; m[7] = n1.w;
            m[8] = n2.x; m[9] = n2.y; m[10] = n2.z; m[11] = n2.w; m[12] = n3.x; m[13] = n3.y; m[14] = n3.z; m[15] = n3.w;
            fetch();
            if (VAR == 2) {
                u4v *d = (u4v *)(dst + (p & 15) * 64);
                d[0] = n0; d[1] = n1; d[2] = n2; d[3] = n3;
            }
        }
        b3::compress(cv, m, p, 0, 64, 0);
        if (!VAR) m[p & 15] ^= cv[0];
    }
    uint32_t x = 0;
    for (int i = 0; i < 8; i++) x ^= cv[i];
    out[blockIdx.x * 256 + threadIdx.x] = x;
}
template <int WPS, int VAR>
void run(int passes_per_simd) {
    uint32_t *d;
    uint8_t *sink;
    int grid = 256 * WPS;  // WPS blocks of 4 waves per CU -> WPS waves per SIMD
    (void)hipMalloc(&d, grid * 256 * 4);
    (void)hipMalloc(&sink, (size_t)grid * 256 * 1024);
    int passes = passes_per_simd / WPS;
    hipEvent_t t0, t1;
    (void)hipEventCreate(&t0); (void)hipEventCreate(&t1);
    k<WPS, VAR><<<grid, 256>>>(d, sink, 1, passes);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(t0);
    k<WPS, VAR><<<grid, 256>>>(d, sink, 2, passes);
    (void)hipEventRecord(t1);
    (void)hipDeviceSynchronize();
    float ms;
    (void)hipEventElapsedTime(&ms, t0, t1);
    printf("variant %d waves/SIMD=%d  %d passes/SIMD: %.3f ms  -> %.1f ns per pass per SIMD; C2 hash (325.6 passes/SIMD) = %.3f ms\n", VAR, WPS,
           passes * WPS, ms, ms * 1e6 / (passes * WPS), ms / (passes * WPS) * 325.6);
    (void)hipFree(d); (void)hipFree(sink);
}
int main() {
    run<1, 0>(3200); run<2, 0>(3200); run<4, 0>(3200); run<8, 0>(3200);
    run<4, 1>(3200); run<4, 2>(3200); run<3, 1>(3198); run<3, 2>(3198);
    return 0;
}
