/*
 * oracle/blake3_ref.c — TEST INFRASTRUCTURE ONLY (checker, never the product path).
 *
 * CPU restatement of the hash the reference calls as `blake3::hash(&[u8]) -> Hash`
 * at znippy-compress/src/stream_packer.rs:L219, znippy-compress/src/slot_packer.rs:L553
 * and znippy-common/src/decompress.rs:L172.  The arithmetic lives in the third-party
 * crate `blake3 1.8.5` (Cargo.lock:L629-632), which is NOT under /root/reference, so
 * this file restates the published BLAKE3 specification (default hash mode: no key,
 * 1 KiB chunks, 64-B blocks, 7 rounds, binary tree, ROOT flag on the last compression).
 *
 * Pinned by: the public known-answer vectors for "" / "abc" / the official
 * `i % 251` input pattern (tests/golden/blake3_kat.json, checked in tests/test_oracle.py).
 */
#include <stdint.h>
#include <stddef.h>
#include <string.h>

#define B3_CHUNK_START 1u
#define B3_CHUNK_END 2u
#define B3_PARENT 4u
#define B3_ROOT 8u

static const uint32_t B3_IV[8] = {0x6A09E667u, 0xBB67AE85u, 0x3C6EF372u, 0xA54FF53Au,
                                  0x510E527Fu, 0x9B05688Cu, 0x1F83D9ABu, 0x5BE0CD19u};

static const uint8_t B3_PERM[16] = {2, 6, 3, 10, 7, 0, 4, 13, 1, 11, 12, 5, 9, 14, 15, 8};

static inline uint32_t rotr32(uint32_t x, int n) { return (x >> n) | (x << (32 - n)); }

static inline void b3_g(uint32_t *s, int a, int b, int c, int d, uint32_t mx, uint32_t my) {
    s[a] = s[a] + s[b] + mx;
    s[d] = rotr32(s[d] ^ s[a], 16);
    s[c] = s[c] + s[d];
    s[b] = rotr32(s[b] ^ s[c], 12);
    s[a] = s[a] + s[b] + my;
    s[d] = rotr32(s[d] ^ s[a], 8);
    s[c] = s[c] + s[d];
    s[b] = rotr32(s[b] ^ s[c], 7);
}

/* One compression; out = full 16-word state after the feed-forward xors. */
static void b3_compress(const uint32_t cv[8], const uint32_t block[16], uint64_t counter,
                        uint32_t block_len, uint32_t flags, uint32_t out[16]) {
    uint32_t s[16], m[16], t[16];
    int r, i;
    for (i = 0; i < 8; i++) s[i] = cv[i];
    s[8] = B3_IV[0]; s[9] = B3_IV[1]; s[10] = B3_IV[2]; s[11] = B3_IV[3];
    s[12] = (uint32_t)counter; s[13] = (uint32_t)(counter >> 32);
    s[14] = block_len; s[15] = flags;
    memcpy(m, block, 64);
    for (r = 0; r < 7; r++) {
        b3_g(s, 0, 4, 8, 12, m[0], m[1]);
        b3_g(s, 1, 5, 9, 13, m[2], m[3]);
        b3_g(s, 2, 6, 10, 14, m[4], m[5]);
        b3_g(s, 3, 7, 11, 15, m[6], m[7]);
        b3_g(s, 0, 5, 10, 15, m[8], m[9]);
        b3_g(s, 1, 6, 11, 12, m[10], m[11]);
        b3_g(s, 2, 7, 8, 13, m[12], m[13]);
        b3_g(s, 3, 4, 9, 14, m[14], m[15]);
        for (i = 0; i < 16; i++) t[i] = m[B3_PERM[i]];
        memcpy(m, t, 64);
    }
    for (i = 0; i < 8; i++) {
        out[i] = s[i] ^ s[i + 8];
        out[i + 8] = s[i + 8] ^ cv[i];
    }
}

static void b3_load_block(const uint8_t *p, size_t n, uint32_t w[16]) {
    uint8_t buf[64];
    int i;
    memset(buf, 0, 64);
    if (n) memcpy(buf, p, n);
    for (i = 0; i < 16; i++)
        w[i] = (uint32_t)buf[4 * i] | ((uint32_t)buf[4 * i + 1] << 8) |
               ((uint32_t)buf[4 * i + 2] << 16) | ((uint32_t)buf[4 * i + 3] << 24);
}

/* Chaining value of one <=1 KiB chunk; if is_root the ROOT flag goes on its last block. */
static void b3_chunk_cv(const uint8_t *p, size_t n, uint64_t chunk_counter, int is_root,
                        uint32_t cv_out[8]) {
    uint32_t cv[8], w[16], st[16];
    size_t nblocks = n == 0 ? 1 : (n + 63) / 64, b;
    memcpy(cv, B3_IV, 32);
    for (b = 0; b < nblocks; b++) {
        size_t off = b * 64, len = n - off < 64 ? n - off : 64;
        uint32_t flags = 0;
        if (n == 0) len = 0;
        if (b == 0) flags |= B3_CHUNK_START;
        if (b == nblocks - 1) flags |= B3_CHUNK_END | (is_root ? B3_ROOT : 0);
        b3_load_block(p + off, len, w);
        b3_compress(cv, w, chunk_counter, (uint32_t)len, flags, st);
        memcpy(cv, st, 32);
    }
    memcpy(cv_out, cv, 32);
}

static void b3_parent_cv(const uint32_t l[8], const uint32_t r[8], int is_root, uint32_t out[8]) {
    uint32_t w[16], st[16];
    memcpy(w, l, 32);
    memcpy(w + 8, r, 32);
    b3_compress(B3_IV, w, 0, 64, B3_PARENT | (is_root ? B3_ROOT : 0), st);
    memcpy(out, st, 32);
}

/* Recursive tree hash exactly as the spec defines it: the left subtree takes the largest
 * power-of-two number of chunks that still leaves at least one byte for the right. */
static void b3_subtree(const uint8_t *p, size_t n, uint64_t chunk_counter, int is_root,
                       uint32_t cv_out[8]) {
    if (n <= 1024) {
        b3_chunk_cv(p, n, chunk_counter, is_root, cv_out);
        return;
    }
    {
        size_t chunks = (n + 1023) / 1024, left_chunks = 1;
        uint32_t l[8], r[8];
        while (left_chunks * 2 < chunks) left_chunks *= 2;
        b3_subtree(p, left_chunks * 1024, chunk_counter, 0, l);
        b3_subtree(p + left_chunks * 1024, n - left_chunks * 1024, chunk_counter + left_chunks, 0, r);
        b3_parent_cv(l, r, is_root, cv_out);
    }
}

/* blake3::hash(input) -> 32 bytes. */
void oracle_blake3(const uint8_t *input, size_t len, uint8_t out[32]) {
    uint32_t cv[8];
    int i;
    b3_subtree(input, len, 0, 1, cv);
    for (i = 0; i < 8; i++) {
        out[4 * i] = (uint8_t)cv[i];
        out[4 * i + 1] = (uint8_t)(cv[i] >> 8);
        out[4 * i + 2] = (uint8_t)(cv[i] >> 16);
        out[4 * i + 3] = (uint8_t)(cv[i] >> 24);
    }
}
