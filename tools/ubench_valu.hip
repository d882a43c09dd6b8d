// Microbenchmark: integer VALU issue rate on gfx950 for the ops BLAKE3 is made of (inline asm so
// the compiler cannot fold anything).  8 independent registers, one op each, repeated.
// Build: hipcc -O3 --offload-arch=gfx950 tools/ubench_valu.hip -o tools/ubench_valu.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define N_ITERS 2048
#define REP8(X) X(a) X(b) X(c) X(d) X(e) X(f) X(g) X(h)
#define OP_ADD(r) asm volatile("v_add_u32 %0, %0, %1" : "+v"(r) : "v"(s));
#define OP_XOR(r) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(r) : "v"(s));
#define OP_ROT(r) asm volatile("v_alignbit_b32 %0, %0, %0, 7" : "+v"(r));
#define OP_ADD3(r) asm volatile("v_add3_u32 %0, %0, %1, %2" : "+v"(r) : "v"(s), "v"(t));
#define OP_PERM(r) asm volatile("v_perm_b32 %0, %0, %0, %1" : "+v"(r) : "v"(s));
#define OP_XAD(r) asm volatile("v_xad_u32 %0, %0, %1, %2" : "+v"(r) : "v"(s), "v"(t));
#define OP_XOR3(r) asm volatile("v_or3_b32 %0, %0, %1, %2" : "+v"(r) : "v"(s), "v"(t));
#define OP_LSHLOR(r) asm volatile("v_lshl_or_b32 %0, %0, 3, %1" : "+v"(r) : "v"(s));
#define OP_MOVDPP(r) asm volatile("v_mov_b32_dpp %0, %0 quad_perm:[1,2,3,0] row_mask:0xf bank_mask:0xf" : "+v"(r));
#define OP_ADDDPP(r) asm volatile("v_add_u32_dpp %0, %1, %0 quad_perm:[1,2,3,0] row_mask:0xf bank_mask:0xf" : "+v"(r) : "v"(s));
template <int OP>
__global__ __launch_bounds__(256) void k(uint32_t *out, uint32_t seed) {
    uint32_t a = threadIdx.x + seed, b = a * 3 + 1, c = a ^ 0x55, d = a + 7, e = b ^ a, f = c + 9, g = d * 5, h = e + 11;
    uint32_t s = a * 7 + 3, t = a * 11 + 5;
    for (int i = 0; i < N_ITERS; i++) {
#pragma unroll
        for (int j = 0; j < 8; j++) {
            if (OP == 0) { REP8(OP_ADD) }
            if (OP == 1) { REP8(OP_XOR) }
            if (OP == 2) { REP8(OP_ROT) }
            if (OP == 3) { REP8(OP_ADD3) }
            if (OP == 4) { REP8(OP_PERM) }
            if (OP == 5) { REP8(OP_XAD) }
            if (OP == 6) { REP8(OP_XOR3) }
            if (OP == 7) { REP8(OP_LSHLOR) }
            if (OP == 8) { REP8(OP_MOVDPP) }
            if (OP == 9) { REP8(OP_ADDDPP) }
        }
    }
    out[blockIdx.x * 256 + threadIdx.x] = a ^ b ^ c ^ d ^ e ^ f ^ g ^ h;
}
template <int OP>
void run(const char *name, int blocks_per_cu) {
    uint32_t *d;
    int grid = 256 * blocks_per_cu;
    (void)hipMalloc(&d, grid * 256 * 4);
    hipEvent_t t0, t1;
    (void)hipEventCreate(&t0); (void)hipEventCreate(&t1);
    k<OP><<<grid, 256>>>(d, 1);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(t0);
    k<OP><<<grid, 256>>>(d, 2);
    (void)hipEventRecord(t1);
    (void)hipDeviceSynchronize();
    float ms;
    (void)hipEventElapsedTime(&ms, t0, t1);
    double wave_instr = (double)grid * 4 * N_ITERS * 64;
    double per_simd = wave_instr / (256.0 * 4);
    printf("%-10s waves/SIMD=%d  %.3f ms  %.2f cycles/wave-instr/SIMD @2.4GHz  %.1f T lane-ops/s\n", name, blocks_per_cu, ms,
           ms * 1e6 / per_simd * 2.4, wave_instr * 64 / (ms * 1e-3) / 1e12);
    (void)hipFree(d);
}
int main() {
    for (int bpc : {1, 2, 4, 8}) {
        run<0>("add", bpc); run<1>("xor", bpc); run<2>("alignbit", bpc); run<3>("add3", bpc); run<4>("perm", bpc);
        run<5>("xad", bpc); run<6>("or3", bpc); run<7>("lshl_or", bpc); run<8>("mov_dpp", bpc); run<9>("add_dpp", bpc);
    }
    return 0;
}
