/*
 * oracle/zstd_ref.c — TEST INFRASTRUCTURE ONLY (checker, never the product path).
 *
 * CPU restatement of what the reference reaches through
 *   codec::decompress_into            znippy-common/src/codec.rs:L67-78
 *   zl_get_decompressed_size          znippy-common/src/codec.rs:L69
 *   zl_decompress                     znippy-common/src/codec.rs:L74
 * The arithmetic lives in `openzl-sys-rs 0.2.0` (Cargo.lock:L2576-2583; statically linked
 * facebook/openzl whose source is downloaded at build time) and is NOT under /root/reference.
 * For untyped serial input OpenZL's default graph bottoms out in a zstd frame; this build's
 * codec wire format is the plain Zstandard frame of RFC 8878, so this file restates the
 * published RFC 8878 decoding algorithm (frame header, raw/RLE/compressed blocks, Huffman
 * literals incl. FSE-compressed weights and treeless mode, FSE sequences with
 * predefined/RLE/compressed/repeat tables, repeat offsets, XXH64 content checksum).
 *
 * PARITY UNPINNED against OpenZL bytes: the reference ships no golden blobs
 * (SURVEY.md §8c).  Pinned against the container's libzstd 1.4.8 (an independent
 * implementation of the same RFC) in tests/test_oracle.py and by committed frames in
 * tests/golden/.
 */
#include <stdint.h>
#include <stddef.h>
#include <string.h>
#include <stdlib.h>

#define ZR_OK 0
#define ZR_ERR_SRC_TRUNC (-1)
#define ZR_ERR_MAGIC (-2)
#define ZR_ERR_HEADER (-3)
#define ZR_ERR_DST_SMALL (-4)
#define ZR_ERR_CORRUPT (-5)
#define ZR_ERR_UNSUPPORTED (-6)
#define ZR_ERR_CHECKSUM (-7)
#define ZR_ERR_SIZE_UNKNOWN (-8)

#define ZR_BLOCK_MAX (128 * 1024)
#define ZR_HUF_MAX_LOG 11
#define ZR_LL_MAX_LOG 9
#define ZR_OF_MAX_LOG 8
#define ZR_ML_MAX_LOG 9

typedef struct {
    uint8_t symbol[512];
    uint8_t nbits[512];
    uint16_t base[512];
    int log;
} zr_fse_t;

typedef struct {
    uint8_t symbol[1 << ZR_HUF_MAX_LOG];
    uint8_t nbits[1 << ZR_HUF_MAX_LOG];
    int log;
    int valid;
} zr_huf_t;

typedef struct {
    zr_huf_t huf;
    zr_fse_t ll, of, ml;
    int ll_valid, of_valid, ml_valid;
    uint64_t rep[3];
    uint8_t *lit; /* ZR_BLOCK_MAX bytes of literal scratch */
} zr_frame_t;

static const uint32_t ZR_LL_BASE[36] = {0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16, 18,
                                        20, 22, 24, 28, 32, 40, 48, 64, 128, 256, 512, 1024, 2048,
                                        4096, 8192, 16384, 32768, 65536};
static const uint8_t ZR_LL_BITS[36] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 1, 1,
                                       1, 1, 2, 2, 3, 3, 4, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16};
static const uint32_t ZR_ML_BASE[53] = {3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16, 17, 18, 19, 20,
                                        21, 22, 23, 24, 25, 26, 27, 28, 29, 30, 31, 32, 33, 34, 35, 37,
                                        39, 41, 43, 47, 51, 59, 67, 83, 99, 131, 259, 515, 1027, 2051,
                                        4099, 8195, 16387, 32771, 65539};
static const uint8_t ZR_ML_BITS[53] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0,
                                       0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 1, 1,
                                       1, 1, 2, 2, 3, 3, 4, 4, 5, 7, 8, 9, 10, 11,
                                       12, 13, 14, 15, 16};
static const int16_t ZR_LL_DEFAULT[36] = {4, 3, 2, 2, 2, 2, 2, 2, 2, 2, 2, 2, 2, 1, 1, 1, 2, 2,
                                          2, 2, 2, 2, 2, 2, 2, 3, 2, 1, 1, 1, 1, 1, -1, -1, -1, -1};
static const int16_t ZR_ML_DEFAULT[53] = {1, 4, 3, 2, 2, 2, 2, 2, 2, 1, 1, 1, 1, 1, 1, 1, 1, 1,
                                          1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1,
                                          1, 1, 1, 1, 1, 1, 1, 1, 1, 1, -1, -1, -1, -1, -1, -1, -1};
static const int16_t ZR_OF_DEFAULT[29] = {1, 1, 1, 1, 1, 1, 2, 2, 2, 1, 1, 1, 1, 1, 1,
                                          1, 1, 1, 1, 1, 1, 1, 1, 1, -1, -1, -1, -1, -1};

static inline int zr_highbit(uint32_t v) { return 31 - __builtin_clz(v); }

/* ---------------- XXH64 (content checksum, RFC 8878 §3.1.1) ---------------- */
#define XP1 0x9E3779B185EBCA87ULL
#define XP2 0xC2B2AE3D27D4EB4FULL
#define XP3 0x165667B19E3779F9ULL
#define XP4 0x85EBCA77C2B2AE63ULL
#define XP5 0x27D4EB2F165667C5ULL
static inline uint64_t xrotl(uint64_t x, int r) { return (x << r) | (x >> (64 - r)); }
static inline uint64_t xrd64(const uint8_t *p) { uint64_t v; memcpy(&v, p, 8); return v; }
static inline uint32_t xrd32(const uint8_t *p) { uint32_t v; memcpy(&v, p, 4); return v; }
static inline uint64_t xround(uint64_t acc, uint64_t in) { acc += in * XP2; acc = xrotl(acc, 31); return acc * XP1; }
static inline uint64_t xmerge(uint64_t acc, uint64_t v) { v = xround(0, v); acc ^= v; return acc * XP1 + XP4; }
uint64_t oracle_xxh64(const uint8_t *p, size_t len, uint64_t seed) {
    const uint8_t *end = p + len;
    uint64_t h;
    if (len >= 32) {
        uint64_t v1 = seed + XP1 + XP2, v2 = seed + XP2, v3 = seed, v4 = seed - XP1;
        const uint8_t *lim = end - 32;
        do {
            v1 = xround(v1, xrd64(p)); v2 = xround(v2, xrd64(p + 8));
            v3 = xround(v3, xrd64(p + 16)); v4 = xround(v4, xrd64(p + 24));
            p += 32;
        } while (p <= lim);
        h = xrotl(v1, 1) + xrotl(v2, 7) + xrotl(v3, 12) + xrotl(v4, 18);
        h = xmerge(h, v1); h = xmerge(h, v2); h = xmerge(h, v3); h = xmerge(h, v4);
    } else {
        h = seed + XP5;
    }
    h += (uint64_t)len;
    while (p + 8 <= end) { h ^= xround(0, xrd64(p)); h = xrotl(h, 27) * XP1 + XP4; p += 8; }
    if (p + 4 <= end) { h ^= (uint64_t)xrd32(p) * XP1; h = xrotl(h, 23) * XP2 + XP3; p += 4; }
    while (p < end) { h ^= (*p) * XP5; h = xrotl(h, 11) * XP1; p++; }
    h ^= h >> 33; h *= XP2; h ^= h >> 29; h *= XP3; h ^= h >> 32;
    return h;
}

/* ---------------- forward bit reader (FSE table descriptions) ---------------- */
typedef struct { const uint8_t *p; size_t n; size_t bitpos; } zr_fwd_t;
static uint32_t zr_fwd_read(zr_fwd_t *b, int nbits) {
    uint32_t v = 0;
    int i;
    for (i = 0; i < nbits; i++) {
        size_t bp = b->bitpos + i;
        uint32_t bit = (bp >> 3) < b->n ? (b->p[bp >> 3] >> (bp & 7)) & 1u : 0u;
        v |= bit << i;
    }
    b->bitpos += nbits;
    return v;
}

/* ---------------- backward bit reader (FSE / Huffman streams) ----------------
 * `pos` = number of not-yet-consumed bits; reading n bits returns bits [pos-n,pos) with the
 * most significant at pos-1; bits below 0 read as zero (RFC 8878 §4.1). */
typedef struct { const uint8_t *p; int64_t pos; } zr_bwd_t;
static int zr_bwd_init(zr_bwd_t *b, const uint8_t *p, size_t n) {
    if (n == 0 || p[n - 1] == 0) return ZR_ERR_CORRUPT;
    b->p = p;
    b->pos = (int64_t)n * 8 - (8 - zr_highbit(p[n - 1]));
    return ZR_OK;
}
static inline uint64_t zr_bwd_peek(const zr_bwd_t *b, int nbits) {
    /* value of the nbits bits ending at pos (zero-filled below bit 0) */
    uint64_t v = 0;
    int i;
    for (i = 0; i < nbits; i++) {
        int64_t bp = b->pos - 1 - i;
        uint64_t bit = bp >= 0 ? (b->p[bp >> 3] >> (bp & 7)) & 1u : 0u;
        v |= bit << (nbits - 1 - i);
    }
    return v;
}
static inline uint64_t zr_bwd_read(zr_bwd_t *b, int nbits) {
    uint64_t v = zr_bwd_peek(b, nbits);
    b->pos -= nbits;
    return v;
}

/* ---------------- FSE ---------------- */
static int zr_fse_read_ncount(const uint8_t *src, size_t n, int max_log, int max_sym,
                              int16_t *norm, int *nsym, int *log, size_t *consumed) {
    zr_fwd_t b = {src, n, 0};
    int alog, remaining, s = 0;
    if (n == 0) return ZR_ERR_SRC_TRUNC;
    alog = 5 + (int)zr_fwd_read(&b, 4);
    if (alog > max_log) return ZR_ERR_CORRUPT;
    remaining = 1 << alog;
    while (remaining > 0 && s <= max_sym) {
        int bits = zr_highbit((uint32_t)remaining + 1) + 1;
        uint32_t val = zr_fwd_read(&b, bits);
        uint32_t lower_mask = (1u << (bits - 1)) - 1;
        uint32_t threshold = (1u << bits) - 1 - ((uint32_t)remaining + 1);
        int proba;
        if ((val & lower_mask) < threshold) {
            b.bitpos -= 1;
            val &= lower_mask;
        } else if (val > lower_mask) {
            val -= threshold;
        }
        proba = (int)val - 1;
        remaining -= proba < 0 ? -proba : proba;
        norm[s++] = (int16_t)proba;
        if (proba == 0) {
            uint32_t rep = zr_fwd_read(&b, 2);
            for (;;) {
                uint32_t i;
                for (i = 0; i < rep && s <= max_sym; i++) norm[s++] = 0;
                if (rep == 3) rep = zr_fwd_read(&b, 2); else break;
            }
        }
    }
    if (remaining != 0) return ZR_ERR_CORRUPT;
    if ((b.bitpos + 7) / 8 > n) return ZR_ERR_SRC_TRUNC;
    *nsym = s;
    *log = alog;
    *consumed = (b.bitpos + 7) / 8;
    return ZR_OK;
}

static int zr_fse_build(zr_fse_t *t, const int16_t *norm, int nsym, int log) {
    int size = 1 << log, high = size, s, i, pos = 0;
    int step = (size >> 1) + (size >> 3) + 3, mask = size - 1;
    uint16_t next[256];
    t->log = log;
    for (s = 0; s < nsym; s++)
        if (norm[s] == -1) { t->symbol[--high] = (uint8_t)s; next[s] = 1; }
    for (s = 0; s < nsym; s++) {
        if (norm[s] <= 0) continue;
        next[s] = (uint16_t)norm[s];
        for (i = 0; i < norm[s]; i++) {
            t->symbol[pos] = (uint8_t)s;
            do { pos = (pos + step) & mask; } while (pos >= high);
        }
    }
    if (pos != 0) return ZR_ERR_CORRUPT;
    for (i = 0; i < size; i++) {
        uint16_t ns = next[t->symbol[i]]++;
        int nb = log - zr_highbit(ns);
        t->nbits[i] = (uint8_t)nb;
        t->base[i] = (uint16_t)(((uint32_t)ns << nb) - size);
    }
    return ZR_OK;
}

static void zr_fse_rle(zr_fse_t *t, uint8_t sym) {
    t->log = 0; t->symbol[0] = sym; t->nbits[0] = 0; t->base[0] = 0;
}

/* ---------------- Huffman ---------------- */
static int zr_huf_build(zr_huf_t *h, const uint8_t *weights, int nsym_given) {
    /* weights[0..nsym_given) decoded; the last symbol's weight is implied */
    uint32_t total = 0, left;
    int i, maxbits, nsym = nsym_given + 1, lastw;
    uint8_t bits[256];
    uint32_t rank_count[ZR_HUF_MAX_LOG + 2], rank_idx[ZR_HUF_MAX_LOG + 2];
    uint8_t w[256];
    if (nsym_given < 1 || nsym > 256) return ZR_ERR_CORRUPT;
    for (i = 0; i < nsym_given; i++) {
        if (weights[i] > ZR_HUF_MAX_LOG + 1) return ZR_ERR_CORRUPT;
        w[i] = weights[i];
        total += weights[i] ? 1u << (weights[i] - 1) : 0;
    }
    if (total == 0) return ZR_ERR_CORRUPT;
    maxbits = zr_highbit(total) + 1;
    if (maxbits > ZR_HUF_MAX_LOG) return ZR_ERR_CORRUPT;
    left = (1u << maxbits) - total;
    if (left & (left - 1)) return ZR_ERR_CORRUPT; /* must be a power of two */
    lastw = zr_highbit(left) + 1;
    w[nsym_given] = (uint8_t)lastw;
    memset(rank_count, 0, sizeof rank_count);
    for (i = 0; i < nsym; i++) {
        bits[i] = w[i] ? (uint8_t)(maxbits + 1 - w[i]) : 0;
        rank_count[bits[i]]++;
    }
    rank_idx[maxbits] = 0;
    for (i = maxbits; i >= 1; i--) {
        rank_idx[i - 1] = rank_idx[i] + rank_count[i] * (1u << (maxbits - i));
        memset(&h->nbits[rank_idx[i]], i, rank_idx[i - 1] - rank_idx[i]);
    }
    if (rank_idx[0] != (1u << maxbits)) return ZR_ERR_CORRUPT;
    for (i = 0; i < nsym; i++) {
        if (bits[i]) {
            uint32_t len = 1u << (maxbits - bits[i]);
            memset(&h->symbol[rank_idx[bits[i]]], i, len);
            rank_idx[bits[i]] += len;
        }
    }
    h->log = maxbits;
    h->valid = 1;
    return ZR_OK;
}

static int zr_huf_read_tree(zr_huf_t *h, const uint8_t *src, size_t n, size_t *consumed) {
    uint8_t weights[256];
    int nw = 0;
    uint8_t hb;
    if (n < 1) return ZR_ERR_SRC_TRUNC;
    hb = src[0];
    if (hb >= 128) {
        int num = hb - 127, i;
        size_t bytes = (size_t)(num + 1) / 2;
        if (1 + bytes > n) return ZR_ERR_SRC_TRUNC;
        for (i = 0; i < num; i++) {
            uint8_t b = src[1 + i / 2];
            weights[i] = (i & 1) ? (b & 15) : (b >> 4);
        }
        nw = num;
        *consumed = 1 + bytes;
    } else {
        /* FSE-compressed weights: table description then two interleaved states */
        int16_t norm[256];
        int nsym, log, rc;
        size_t hdr;
        zr_fse_t t;
        zr_bwd_t b;
        uint32_t s1, s2;
        if (hb == 0 || (size_t)1 + hb > n) return ZR_ERR_SRC_TRUNC;
        rc = zr_fse_read_ncount(src + 1, hb, 6, 255, norm, &nsym, &log, &hdr);
        if (rc) return rc;
        rc = zr_fse_build(&t, norm, nsym, log);
        if (rc) return rc;
        if (hdr >= hb) return ZR_ERR_CORRUPT;
        rc = zr_bwd_init(&b, src + 1 + hdr, hb - hdr);
        if (rc) return rc;
        s1 = (uint32_t)zr_bwd_read(&b, log);
        s2 = (uint32_t)zr_bwd_read(&b, log);
        for (;;) {
            if (nw >= 255) return ZR_ERR_CORRUPT;
            weights[nw++] = t.symbol[s1];
            s1 = t.base[s1] + (uint32_t)zr_bwd_read(&b, t.nbits[s1]);
            if (b.pos < 0) {
                if (nw >= 255) return ZR_ERR_CORRUPT;
                weights[nw++] = t.symbol[s2];
                break;
            }
            if (nw >= 255) return ZR_ERR_CORRUPT;
            weights[nw++] = t.symbol[s2];
            s2 = t.base[s2] + (uint32_t)zr_bwd_read(&b, t.nbits[s2]);
            if (b.pos < 0) {
                if (nw >= 255) return ZR_ERR_CORRUPT;
                weights[nw++] = t.symbol[s1];
                break;
            }
        }
        *consumed = (size_t)1 + hb;
    }
    return zr_huf_build(h, weights, nw);
}

static int zr_huf_decode_stream(const zr_huf_t *h, const uint8_t *src, size_t n, uint8_t *dst,
                                size_t nout) {
    zr_bwd_t b;
    size_t i;
    int rc = zr_bwd_init(&b, src, n);
    if (rc) return rc;
    for (i = 0; i < nout; i++) {
        uint32_t idx = (uint32_t)zr_bwd_peek(&b, h->log);
        dst[i] = h->symbol[idx];
        b.pos -= h->nbits[idx];
    }
    if (b.pos != 0) return ZR_ERR_CORRUPT;
    return ZR_OK;
}

/* ---------------- literals section ---------------- */
typedef struct { const uint8_t *ptr; size_t len; int rle; uint8_t rle_byte; } zr_lits_t;

static int zr_decode_literals(zr_frame_t *f, const uint8_t *src, size_t n, zr_lits_t *out,
                              size_t *consumed) {
    int type, sf;
    size_t regen, comp = 0, hdr;
    if (n < 1) return ZR_ERR_SRC_TRUNC;
    type = src[0] & 3;
    sf = (src[0] >> 2) & 3;
    if (type <= 1) {
        if ((sf & 1) == 0) { regen = src[0] >> 3; hdr = 1; }
        else if (sf == 1) { if (n < 2) return ZR_ERR_SRC_TRUNC; regen = (src[0] >> 4) + ((size_t)src[1] << 4); hdr = 2; }
        else { if (n < 3) return ZR_ERR_SRC_TRUNC; regen = (src[0] >> 4) + ((size_t)src[1] << 4) + ((size_t)src[2] << 12); hdr = 3; }
        if (regen > ZR_BLOCK_MAX) return ZR_ERR_CORRUPT;
        if (type == 0) {
            if (hdr + regen > n) return ZR_ERR_SRC_TRUNC;
            out->ptr = src + hdr; out->len = regen; out->rle = 0;
            *consumed = hdr + regen;
        } else {
            if (hdr + 1 > n) return ZR_ERR_SRC_TRUNC;
            out->ptr = NULL; out->len = regen; out->rle = 1; out->rle_byte = src[hdr];
            *consumed = hdr + 1;
        }
        return ZR_OK;
    }
    {
        int streams;
        uint64_t h = 0;
        const uint8_t *p;
        size_t remain, i;
        int rc;
        if (n < 5) { for (i = 0; i < n; i++) h |= (uint64_t)src[i] << (8 * i); }
        else { for (i = 0; i < 5; i++) h |= (uint64_t)src[i] << (8 * i); }
        if (sf == 0) { streams = 1; regen = (h >> 4) & 0x3FF; comp = (h >> 14) & 0x3FF; hdr = 3; }
        else if (sf == 1) { streams = 4; regen = (h >> 4) & 0x3FF; comp = (h >> 14) & 0x3FF; hdr = 3; }
        else if (sf == 2) { streams = 4; regen = (h >> 4) & 0x3FFF; comp = (h >> 18) & 0x3FFF; hdr = 4; }
        else { streams = 4; regen = (h >> 4) & 0x3FFFF; comp = (h >> 22) & 0x3FFFF; hdr = 5; }
        if (hdr + comp > n) return ZR_ERR_SRC_TRUNC;
        if (regen > ZR_BLOCK_MAX) return ZR_ERR_CORRUPT;
        p = src + hdr;
        remain = comp;
        if (type == 2) {
            size_t used;
            rc = zr_huf_read_tree(&f->huf, p, remain, &used);
            if (rc) return rc;
            p += used; remain -= used;
        } else if (!f->huf.valid) {
            return ZR_ERR_CORRUPT; /* treeless without a previous table */
        }
        if (streams == 1) {
            rc = zr_huf_decode_stream(&f->huf, p, remain, f->lit, regen);
            if (rc) return rc;
        } else {
            size_t s1, s2, s3, s4, seg = (regen + 3) / 4;
            if (remain < 6) return ZR_ERR_CORRUPT;
            s1 = p[0] | ((size_t)p[1] << 8); s2 = p[2] | ((size_t)p[3] << 8); s3 = p[4] | ((size_t)p[5] << 8);
            if (6 + s1 + s2 + s3 > remain) return ZR_ERR_CORRUPT;
            s4 = remain - 6 - s1 - s2 - s3;
            if (3 * seg > regen) return ZR_ERR_CORRUPT;
            p += 6;
            rc = zr_huf_decode_stream(&f->huf, p, s1, f->lit, seg); if (rc) return rc;
            rc = zr_huf_decode_stream(&f->huf, p + s1, s2, f->lit + seg, seg); if (rc) return rc;
            rc = zr_huf_decode_stream(&f->huf, p + s1 + s2, s3, f->lit + 2 * seg, seg); if (rc) return rc;
            rc = zr_huf_decode_stream(&f->huf, p + s1 + s2 + s3, s4, f->lit + 3 * seg, regen - 3 * seg); if (rc) return rc;
        }
        out->ptr = f->lit; out->len = regen; out->rle = 0;
        *consumed = hdr + comp;
        return ZR_OK;
    }
}

/* ---------------- sequences section ---------------- */
static int zr_seq_table(zr_fse_t *t, int *valid, int mode, const uint8_t **pp, const uint8_t *end,
                        const int16_t *def, int def_n, int def_log, int max_log, int max_sym) {
    const uint8_t *p = *pp;
    int rc;
    switch (mode) {
    case 0:
        rc = zr_fse_build(t, def, def_n, def_log);
        if (rc) return rc;
        *valid = 1;
        return ZR_OK;
    case 1:
        if (p >= end) return ZR_ERR_SRC_TRUNC;
        if (*p > max_sym) return ZR_ERR_CORRUPT;
        zr_fse_rle(t, *p);
        *pp = p + 1;
        *valid = 1;
        return ZR_OK;
    case 2: {
        int16_t norm[64];
        int nsym, log;
        size_t used;
        rc = zr_fse_read_ncount(p, (size_t)(end - p), max_log, max_sym, norm, &nsym, &log, &used);
        if (rc) return rc;
        rc = zr_fse_build(t, norm, nsym, log);
        if (rc) return rc;
        *pp = p + used;
        *valid = 1;
        return ZR_OK;
    }
    default:
        return *valid ? ZR_OK : ZR_ERR_CORRUPT;
    }
}

static inline void zr_copy_lits(uint8_t *dst, const zr_lits_t *l, size_t at, size_t n) {
    if (l->rle) memset(dst, l->rle_byte, n);
    else memcpy(dst, l->ptr + at, n);
}

static int zr_decode_block(zr_frame_t *f, const uint8_t *src, size_t n, uint8_t *dst_base,
                           size_t dst_pos, size_t dst_cap, size_t *produced) {
    zr_lits_t lits;
    size_t used, lit_at = 0, out = dst_pos;
    memset(&lits, 0, sizeof lits);
    const uint8_t *p, *end = src + n;
    uint32_t nseq;
    int rc = zr_decode_literals(f, src, n, &lits, &used);
    if (rc) return rc;
    p = src + used;
    if (p >= end) return ZR_ERR_SRC_TRUNC;
    if (p[0] == 0) { nseq = 0; p += 1; }
    else if (p[0] < 128) { nseq = p[0]; p += 1; }
    else if (p[0] < 255) { if (p + 2 > end) return ZR_ERR_SRC_TRUNC; nseq = ((uint32_t)(p[0] - 128) << 8) + p[1]; p += 2; }
    else { if (p + 3 > end) return ZR_ERR_SRC_TRUNC; nseq = (uint32_t)p[1] + ((uint32_t)p[2] << 8) + 0x7F00; p += 3; }
    if (nseq) {
        uint8_t modes;
        zr_bwd_t b;
        uint32_t sl, so, sm, i;
        if (p >= end) return ZR_ERR_SRC_TRUNC;
        modes = *p++;
        if (modes & 3) return ZR_ERR_CORRUPT;
        rc = zr_seq_table(&f->ll, &f->ll_valid, (modes >> 6) & 3, &p, end, ZR_LL_DEFAULT, 36, 6, ZR_LL_MAX_LOG, 35); if (rc) return rc;
        rc = zr_seq_table(&f->of, &f->of_valid, (modes >> 4) & 3, &p, end, ZR_OF_DEFAULT, 29, 5, ZR_OF_MAX_LOG, 31); if (rc) return rc;
        rc = zr_seq_table(&f->ml, &f->ml_valid, (modes >> 2) & 3, &p, end, ZR_ML_DEFAULT, 53, 6, ZR_ML_MAX_LOG, 52); if (rc) return rc;
        if (p >= end) return ZR_ERR_SRC_TRUNC;
        rc = zr_bwd_init(&b, p, (size_t)(end - p));
        if (rc) return rc;
        sl = (uint32_t)zr_bwd_read(&b, f->ll.log);
        so = (uint32_t)zr_bwd_read(&b, f->of.log);
        sm = (uint32_t)zr_bwd_read(&b, f->ml.log);
        for (i = 0; i < nseq; i++) {
            uint32_t oc = f->of.symbol[so], lc = f->ll.symbol[sl], mc = f->ml.symbol[sm];
            uint64_t ov, offset, ml, ll;
            if (oc > 31 || lc > 35 || mc > 52) return ZR_ERR_CORRUPT;
            ov = ((uint64_t)1 << oc) + zr_bwd_read(&b, (int)oc);
            ml = ZR_ML_BASE[mc] + zr_bwd_read(&b, ZR_ML_BITS[mc]);
            ll = ZR_LL_BASE[lc] + zr_bwd_read(&b, ZR_LL_BITS[lc]);
            if (i + 1 < nseq) {
                sl = f->ll.base[sl] + (uint32_t)zr_bwd_read(&b, f->ll.nbits[sl]);
                sm = f->ml.base[sm] + (uint32_t)zr_bwd_read(&b, f->ml.nbits[sm]);
                so = f->of.base[so] + (uint32_t)zr_bwd_read(&b, f->of.nbits[so]);
            }
            if (b.pos < 0) return ZR_ERR_CORRUPT;
            if (ov > 3) {
                offset = ov - 3;
                f->rep[2] = f->rep[1]; f->rep[1] = f->rep[0]; f->rep[0] = offset;
            } else {
                uint32_t idx = (uint32_t)ov - 1 + (ll == 0 ? 1 : 0);
                if (idx == 0) {
                    offset = f->rep[0];
                } else {
                    offset = idx < 3 ? f->rep[idx] : f->rep[0] - 1;
                    if (offset == 0) return ZR_ERR_CORRUPT;
                    if (idx > 1) f->rep[2] = f->rep[1];
                    f->rep[1] = f->rep[0];
                    f->rep[0] = offset;
                }
            }
            if (lit_at + ll > lits.len) return ZR_ERR_CORRUPT;
            if (out + ll + ml > dst_cap) return ZR_ERR_DST_SMALL;
            zr_copy_lits(dst_base + out, &lits, lit_at, ll);
            lit_at += ll; out += ll;
            if (offset > out) return ZR_ERR_CORRUPT;
            {
                uint8_t *d = dst_base + out;
                const uint8_t *s = d - offset;
                uint64_t k;
                for (k = 0; k < ml; k++) d[k] = s[k]; /* byte order matters when offset < ml */
            }
            out += ml;
        }
        if (b.pos != 0) return ZR_ERR_CORRUPT;
    } else if (p != end) {
        return ZR_ERR_CORRUPT;
    }
    {
        size_t rest = lits.len - lit_at;
        if (out + rest > dst_cap) return ZR_ERR_DST_SMALL;
        zr_copy_lits(dst_base + out, &lits, lit_at, rest);
        out += rest;
    }
    if (out - dst_pos > ZR_BLOCK_MAX) return ZR_ERR_CORRUPT;
    *produced = out - dst_pos;
    return ZR_OK;
}

/* ---------------- frame ---------------- */
typedef struct {
    uint64_t content_size; int has_size; uint64_t window; int checksum; uint32_t dict_id; size_t hdr_len;
} zr_hdr_t;

static int zr_parse_header(const uint8_t *src, size_t n, zr_hdr_t *h) {
    uint8_t fhd;
    int fcs_flag, single, did_flag, fcs_bytes, did_bytes;
    size_t pos = 5;
    static const int did_sizes[4] = {0, 1, 2, 4};
    if (n < 5) return ZR_ERR_SRC_TRUNC;
    if (xrd32(src) != 0xFD2FB528u) return ZR_ERR_MAGIC;
    fhd = src[4];
    fcs_flag = fhd >> 6; single = (fhd >> 5) & 1; did_flag = fhd & 3;
    if (fhd & 0x08) return ZR_ERR_HEADER; /* reserved bit */
    h->checksum = (fhd >> 2) & 1;
    fcs_bytes = fcs_flag == 0 ? single : (1 << fcs_flag);
    did_bytes = did_sizes[did_flag];
    if (n < pos + (single ? 0 : 1) + did_bytes + fcs_bytes) return ZR_ERR_SRC_TRUNC;
    h->window = 0;
    if (!single) {
        uint8_t wd = src[pos++];
        int wlog = 10 + (wd >> 3);
        uint64_t base = (uint64_t)1 << wlog;
        if (wlog > 41) return ZR_ERR_HEADER;
        h->window = base + (base / 8) * (wd & 7);
    }
    h->dict_id = 0;
    { int i; for (i = 0; i < did_bytes; i++) h->dict_id |= (uint32_t)src[pos + i] << (8 * i); pos += did_bytes; }
    h->has_size = fcs_bytes != 0;
    h->content_size = 0;
    { int i; for (i = 0; i < fcs_bytes; i++) h->content_size |= (uint64_t)src[pos + i] << (8 * i); pos += fcs_bytes; }
    if (fcs_bytes == 2) h->content_size += 256;
    if (single) h->window = h->content_size;
    h->hdr_len = pos;
    return ZR_OK;
}

/* zl_get_decompressed_size: content size of the (first non-skippable) frame. */
int oracle_zstd_get_decompressed_size(const uint8_t *src, size_t n, uint64_t *out) {
    zr_hdr_t h;
    int rc;
    while (n >= 8 && (xrd32(src) & 0xFFFFFFF0u) == 0x184D2A50u) {
        uint32_t sz = xrd32(src + 4);
        if ((size_t)8 + sz > n) return ZR_ERR_SRC_TRUNC;
        src += 8 + sz; n -= 8 + sz;
    }
    rc = zr_parse_header(src, n, &h);
    if (rc) return rc;
    if (!h.has_size) return ZR_ERR_SIZE_UNKNOWN;
    *out = h.content_size;
    return ZR_OK;
}

/* zl_decompress: returns bytes written (>=0) or a negative ZR_ERR_*. */
int64_t oracle_zstd_decompress(uint8_t *dst, size_t cap, const uint8_t *src, size_t n) {
    size_t out = 0;
    uint8_t *lit = (uint8_t *)malloc(ZR_BLOCK_MAX + 64);
    zr_frame_t *f = (zr_frame_t *)calloc(1, sizeof(zr_frame_t));
    int64_t result;
    if (!lit || !f) { free(lit); free(f); return ZR_ERR_UNSUPPORTED; }
    while (n > 0) {
        zr_hdr_t h;
        size_t frame_start = out;
        int rc, last = 0;
        if (n >= 8 && (xrd32(src) & 0xFFFFFFF0u) == 0x184D2A50u) {
            uint32_t sz = xrd32(src + 4);
            if ((size_t)8 + sz > n) { result = ZR_ERR_SRC_TRUNC; goto done; }
            src += 8 + sz; n -= 8 + sz;
            continue;
        }
        rc = zr_parse_header(src, n, &h);
        if (rc) { result = rc; goto done; }
        if (h.dict_id) { result = ZR_ERR_UNSUPPORTED; goto done; }
        src += h.hdr_len; n -= h.hdr_len;
        memset(f, 0, sizeof *f);
        f->lit = lit;
        f->rep[0] = 1; f->rep[1] = 4; f->rep[2] = 8;
        while (!last) {
            uint32_t bh, btype, bsize;
            if (n < 3) { result = ZR_ERR_SRC_TRUNC; goto done; }
            bh = src[0] | ((uint32_t)src[1] << 8) | ((uint32_t)src[2] << 16);
            last = bh & 1; btype = (bh >> 1) & 3; bsize = bh >> 3;
            src += 3; n -= 3;
            if (btype == 0) {
                if (bsize > n) { result = ZR_ERR_SRC_TRUNC; goto done; }
                if (out + bsize > cap) { result = ZR_ERR_DST_SMALL; goto done; }
                memcpy(dst + out, src, bsize);
                out += bsize; src += bsize; n -= bsize;
            } else if (btype == 1) {
                if (n < 1) { result = ZR_ERR_SRC_TRUNC; goto done; }
                if (out + bsize > cap) { result = ZR_ERR_DST_SMALL; goto done; }
                memset(dst + out, src[0], bsize);
                out += bsize; src += 1; n -= 1;
            } else if (btype == 2) {
                size_t produced;
                if (bsize > n) { result = ZR_ERR_SRC_TRUNC; goto done; }
                if (bsize > ZR_BLOCK_MAX) { result = ZR_ERR_CORRUPT; goto done; }
                /* matches may reach back to the start of THIS frame only */
                rc = zr_decode_block(f, src, bsize, dst + frame_start, out - frame_start,
                                     cap - frame_start, &produced);
                if (rc) { result = rc; goto done; }
                out += produced; src += bsize; n -= bsize;
            } else {
                result = ZR_ERR_CORRUPT; goto done;
            }
        }
        if (h.has_size && out - frame_start != h.content_size) { result = ZR_ERR_CORRUPT; goto done; }
        if (h.checksum) {
            uint32_t want;
            if (n < 4) { result = ZR_ERR_SRC_TRUNC; goto done; }
            want = xrd32(src);
            if ((uint32_t)oracle_xxh64(dst + frame_start, out - frame_start, 0) != want) { result = ZR_ERR_CHECKSUM; goto done; }
            src += 4; n -= 4;
        }
    }
    result = (int64_t)out;
done:
    free(lit); free(f);
    return result;
}
