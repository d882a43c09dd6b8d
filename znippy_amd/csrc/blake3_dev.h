// BLAKE3 compression function for CDNA4 lanes (one lane = one compression; the 7 rounds
// are fully unrolled with a compile-time message schedule so every message word stays in
// a VGPR).  Replaces the arithmetic behind `blake3::hash` as called at
// znippy-compress/src/stream_packer.rs:L219, slot_packer.rs:L553 and
// znippy-common/src/decompress.rs:L172 (crate blake3 1.8.5, not vendored in the reference).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace b3 {

constexpr uint32_t CHUNK_START = 1u, CHUNK_END = 2u, PARENT = 4u, ROOT = 8u;
constexpr uint32_t IV0 = 0x6A09E667u, IV1 = 0xBB67AE85u, IV2 = 0x3C6EF372u, IV3 = 0xA54FF53Au,
                   IV4 = 0x510E527Fu, IV5 = 0x9B05688Cu, IV6 = 0x1F83D9ABu, IV7 = 0x5BE0CD19u;

__device__ __forceinline__ uint32_t rotr(uint32_t x, int n) {
    return __builtin_amdgcn_alignbit(x, x, n);  // v_alignbit_b32 = 1-op rotate
}

#define B3_G(a, b, c, d, mx, my) \
    a = a + b + (mx);            \
    d = rotr(d ^ a, 16);         \
    c = c + d;                   \
    b = rotr(b ^ c, 12);         \
    a = a + b + (my);            \
    d = rotr(d ^ a, 8);          \
    c = c + d;                   \
    b = rotr(b ^ c, 7);

#define B3_ROUND(m0, m1, m2, m3, m4, m5, m6, m7, m8, m9, m10, m11, m12, m13, m14, m15) \
    B3_G(v0, v4, v8, v12, m0, m1)                                                      \
    B3_G(v1, v5, v9, v13, m2, m3)                                                      \
    B3_G(v2, v6, v10, v14, m4, m5)                                                     \
    B3_G(v3, v7, v11, v15, m6, m7)                                                     \
    B3_G(v0, v5, v10, v15, m8, m9)                                                     \
    B3_G(v1, v6, v11, v12, m10, m11)                                                   \
    B3_G(v2, v7, v8, v13, m12, m13)                                                    \
    B3_G(v3, v4, v9, v14, m14, m15)

// cv (in/out) <- first 8 words of compress(cv, m, counter, block_len, flags)
__device__ __forceinline__ void compress(uint32_t cv[8], const uint32_t m[16], uint32_t t_lo,
                                         uint32_t t_hi, uint32_t block_len, uint32_t flags) {
    uint32_t v0 = cv[0], v1 = cv[1], v2 = cv[2], v3 = cv[3], v4 = cv[4], v5 = cv[5], v6 = cv[6],
             v7 = cv[7];
    uint32_t v8 = IV0, v9 = IV1, v10 = IV2, v11 = IV3, v12 = t_lo, v13 = t_hi, v14 = block_len,
             v15 = flags;
    // message schedule: round r uses m[PERM^r(i)]
    B3_ROUND(m[0], m[1], m[2], m[3], m[4], m[5], m[6], m[7], m[8], m[9], m[10], m[11], m[12], m[13], m[14], m[15])
    B3_ROUND(m[2], m[6], m[3], m[10], m[7], m[0], m[4], m[13], m[1], m[11], m[12], m[5], m[9], m[14], m[15], m[8])
    B3_ROUND(m[3], m[4], m[10], m[12], m[13], m[2], m[7], m[14], m[6], m[5], m[9], m[0], m[11], m[15], m[8], m[1])
    B3_ROUND(m[10], m[7], m[12], m[9], m[14], m[3], m[13], m[15], m[4], m[0], m[11], m[2], m[5], m[8], m[1], m[6])
    B3_ROUND(m[12], m[13], m[9], m[11], m[15], m[10], m[14], m[8], m[7], m[2], m[5], m[3], m[0], m[1], m[6], m[4])
    B3_ROUND(m[9], m[14], m[11], m[5], m[8], m[12], m[15], m[1], m[13], m[3], m[0], m[10], m[2], m[6], m[4], m[7])
    B3_ROUND(m[11], m[15], m[5], m[0], m[1], m[9], m[8], m[6], m[14], m[10], m[2], m[12], m[3], m[4], m[7], m[13])
    cv[0] = v0 ^ v8;  cv[1] = v1 ^ v9;  cv[2] = v2 ^ v10; cv[3] = v3 ^ v11;
    cv[4] = v4 ^ v12; cv[5] = v5 ^ v13; cv[6] = v6 ^ v14; cv[7] = v7 ^ v15;
}

__device__ __forceinline__ void set_iv(uint32_t cv[8]) {
    cv[0] = IV0; cv[1] = IV1; cv[2] = IV2; cv[3] = IV3; cv[4] = IV4; cv[5] = IV5; cv[6] = IV6; cv[7] = IV7;
}

// parent node: out <- first 8 words of compress(IV, l||r, 0, 64, PARENT|root)
__device__ __forceinline__ void parent(uint32_t out[8], const uint32_t l[8], const uint32_t r[8], bool root) {
    uint32_t m[16];
#pragma unroll
    for (int i = 0; i < 8; i++) { m[i] = l[i]; m[8 + i] = r[i]; }
    set_iv(out);
    compress(out, m, 0, 0, 64, PARENT | (root ? ROOT : 0u));
}

}  // namespace b3
