// Microbenchmark: what a dependent LDS read -> write step costs a lone wave, aligned vs unaligned 16-byte accesses, ds_* vs FLAT.
// hipcc --offload-arch=gfx950 -O3 -o /tmp/ubench_lds tools/ubench_lds_unaligned.hip && /tmp/ubench_lds
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef __attribute__((address_space(3))) uint8_t lds8;
__device__ __forceinline__ uint4 ld_ds(const lds8 *p) { uint4 v; __builtin_memcpy(&v, (const __attribute__((address_space(3))) void *)p, 16); return v; }
__device__ __forceinline__ void st_ds(lds8 *p, uint4 v) { __builtin_memcpy((__attribute__((address_space(3))) void *)p, &v, 16); }
__device__ __forceinline__ uint4 ld_flat(const uint8_t *p) { uint4 v; __builtin_memcpy(&v, p, 16); return v; }
__device__ __forceinline__ void st_flat(uint8_t *p, uint4 v) { __builtin_memcpy(p, &v, 16); }
template <int MODE>
__global__ __launch_bounds__(64) void k(unsigned long long *out, int iters) {
    __shared__ __attribute__((aligned(16))) uint8_t W[16384 + 64];
    const uint32_t lane = threadIdx.x;
    for (uint32_t i = lane; i < 16384 / 4; i += 64) ((uint32_t *)W)[i] = i * 2654435761u;
    __syncthreads();
    lds8 *w = (lds8 *)W;
    uint32_t pos = lane * 200 + (MODE == 0 ? 0 : 5);  // MODE 0: 16-byte aligned addresses; others: odd addresses
    if (MODE == 0) pos &= ~15u;
    uint4 acc = make_uint4(0, 0, 0, 0);
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; it++) {
        uint4 v;
        if (MODE == 2) v = ld_flat(W + pos);  // generic pointer: FLAT
        else v = ld_ds(w + pos);
        acc.x ^= v.x; acc.y += v.y; acc.z ^= v.z; acc.w += v.w;
        const uint32_t d = (pos + 64 + (v.x & (MODE == 0 ? 0x30 : 0x3F))) & 8191;  // the next address depends on the data read
        if (MODE == 2) st_flat(W + 8192 + d, acc);
        else st_ds(w + 8192 + (MODE == 0 ? (d & ~15u) : d), acc);
        pos = MODE == 0 ? (d & ~15u) : d;
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (lane == 0) { out[0] = t1 - t0; out[1] = acc.x ^ acc.y ^ acc.z ^ acc.w; }
}
int main() {
    unsigned long long *d, h[2];
    hipMalloc(&d, 16);
    const int iters = 2000;
    const char *names[3] = {"ds aligned", "ds unaligned", "flat unaligned"};
    for (int rep = 0; rep < 2; rep++)
        for (int m = 0; m < 3; m++) {
            if (m == 0) hipLaunchKernelGGL(k<0>, dim3(1), dim3(64), 0, 0, d, iters);
            if (m == 1) hipLaunchKernelGGL(k<1>, dim3(1), dim3(64), 0, 0, d, iters);
            if (m == 2) hipLaunchKernelGGL(k<2>, dim3(1), dim3(64), 0, 0, d, iters);
            hipMemcpy(h, d, 16, hipMemcpyDeviceToHost);
            printf("%-16s %.1f cycles per dependent read->write step (%llx)\n", names[m], (double)h[0] / iters, h[1]);
        }
    return 0;
}
