// Microbenchmark: does the ORDER in which a wave visits the 64 leaves of a 64 KiB tile bound a copy?  The store path
// (k_hash_tiles<COPY>) moves a leaf's bytes 128 at a time (8 lanes per leaf, 8 leaves per instruction, every leaf
// visited 8 times a compression pair apart): every visit opens the leaf's DRAM row for 128 bytes.  Here the same copy
// with 128 / 256 / 512 / 1024 contiguous bytes per leaf per visit, the same 8 x 16-byte loads in flight per lane, with
// and without VALU filler between the groups (standing for the two compressions).
// hipcc --offload-arch=gfx950 -O3 -o /tmp/ubench_copy tools/ubench_copy_pattern.hip && /tmp/ubench_copy
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>

__device__ __forceinline__ uint4 ld16(const uint8_t *p) { return *reinterpret_cast<const uint4 *>(p); }
__device__ __forceinline__ void st16(uint8_t *p, uint4 v) { *reinterpret_cast<uint4 *>(p) = v; }

// LOGL = log2(lanes per leaf): 3 -> 128 B per leaf per visit ... 6 -> 1 KiB (a leaf per instruction)
template <int LOGL>
__global__ __launch_bounds__(256) void k_copy(const uint8_t *src, uint8_t *dst, uint32_t n_tiles, int filler, uint32_t *sink) {
    __shared__ uint32_t pad[36864 / 4];  // the production kernel's LDS stage: 4 workgroups (16 waves) per CU
    if (filler < 0) pad[threadIdx.x] = 1;
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t tile = blockIdx.x * 4 + wave;
    if (tile >= n_tiles) return;
    constexpr uint32_t LPL = 1u << LOGL, LEAVES_PER_INST = 64 / LPL, CHUNK = 16 * LPL;
    const uint8_t *s = src + ((uint64_t)tile << 16);
    uint8_t *d = dst + ((uint64_t)tile << 16);
    auto off = [&](uint32_t i) -> uint32_t {  // instruction i of the tile's 64
        const uint32_t v = i / LPL, j = i % LPL;
        const uint32_t leaf = j * LEAVES_PER_INST + (lane >> LOGL);
        return leaf * 1024 + v * CHUNK + 16 * (lane & (LPL - 1));
    };
    uint4 v[8], vn[8];
#pragma unroll
    for (int j = 0; j < 8; j++) v[j] = ld16(s + off(j));
    uint32_t acc = lane;
#pragma unroll 1
    for (uint32_t g = 0; g < 8; g++) {
        if (g < 7) {
#pragma unroll
            for (int j = 0; j < 8; j++) vn[j] = ld16(s + off((g + 1) * 8 + j));
        }
#pragma unroll
        for (int j = 0; j < 8; j++) st16(d + off(g * 8 + j), v[j]);
        for (int f = 0; f < filler; f++) {  // dependent VALU work: 4 chains
            acc = __builtin_amdgcn_alignbit(acc, acc, 7) + v[0].x;
            acc ^= __builtin_amdgcn_alignbit(acc, acc, 12) + v[1].y;
            acc = __builtin_amdgcn_alignbit(acc, acc, 16) + v[2].z;
            acc ^= __builtin_amdgcn_alignbit(acc, acc, 8) + v[3].w;
        }
#pragma unroll
        for (int j = 0; j < 8; j++) v[j] = vn[j];
    }
    if (acc == 0x12345678u) sink[0] = acc + pad[lane];
}

template <int LOGL>
static void run(const uint8_t *src, uint8_t *dst, size_t bytes, int filler, uint32_t *sink) {
    const uint32_t n_tiles = (uint32_t)(bytes >> 16);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int w = 0; w < 2; w++) k_copy<LOGL><<<(n_tiles + 3) / 4, 256>>>(src, dst, n_tiles, filler, sink);
    hipEventRecord(e0);
    const int reps = 10;
    for (int r = 0; r < reps; r++) k_copy<LOGL><<<(n_tiles + 3) / 4, 256>>>(src, dst, n_tiles, filler, sink);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    ms /= reps;
    printf("bytes per leaf per visit %4u  filler %4d : %.4f ms  %.2f TB/s moved (read + write)\n", 16u << LOGL, filler, ms,
           2.0 * bytes / (ms * 1e-3) / 1e12);
}

int main() {
    const size_t bytes = 512ull << 20;
    uint8_t *src, *dst;
    uint32_t *sink;
    hipMalloc(&src, bytes); hipMalloc(&dst, bytes); hipMalloc(&sink, 64);
    hipMemset(src, 0x5A, bytes); hipMemset(dst, 0, bytes);
    for (int filler : {0, 100, 200, 300, 400}) {
        run<3>(src, dst, bytes, filler, sink);
        run<4>(src, dst, bytes, filler, sink);
        run<5>(src, dst, bytes, filler, sink);
        run<6>(src, dst, bytes, filler, sink);
    }
    // check one pattern copied everything
    uint8_t *h = (uint8_t *)malloc(bytes);
    hipMemcpy(h, dst, bytes, hipMemcpyDeviceToHost);
    size_t bad = 0;
    for (size_t i = 0; i < bytes; i++) bad += h[i] != 0x5A;
    printf("bytes not copied: %zu\n", bad);
    return bad != 0;
}
