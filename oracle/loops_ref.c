/*
 * oracle/loops_ref.c — TEST INFRASTRUCTURE ONLY (checker + timed CPU baseline, never the
 * product path).
 *
 * CPU restatement of the reference's three Gatling worker loops over in-memory buffers:
 *   read  : znippy-common/src/decompress.rs:L113-192  (atomic row cursor L104/L136, pread
 *           L148-153, decode-or-passthrough L156-166, blake3 verify L172-184, pwrite L186-189,
 *           stats merge L195-221)
 *   write : znippy-compress/src/stream_packer.rs:L215-248 (barrel: blake3 L219, skip L222-227,
 *           compress_into L229-231) + the single writer L255-284 (out_cursor.fetch_add L258)
 *           == znippy-compress/src/slot_packer.rs:L551-580 + L586-609
 * File I/O is replaced by memcpy from/to flat buffers (pread/pwrite analogues) so the loop
 * can be timed against the GPU path on the same resident bytes.
 *
 * Codec on the CPU: the container's libzstd (dlopen'd, level as given; the reference uses
 * level 19, common_config.rs:L37) when `use_libzstd` != 0, else oracle/zstd_ref.c (decode only).
 */
#define _GNU_SOURCE
#include <stdint.h>
#include <stddef.h>
#include <string.h>
#include <stdlib.h>
#include <pthread.h>
#include <dlfcn.h>
#include <stdatomic.h>

void oracle_blake3(const uint8_t *input, size_t len, uint8_t out[32]);
int64_t oracle_zstd_decompress(uint8_t *dst, size_t cap, const uint8_t *src, size_t n);
int oracle_zstd_get_decompressed_size(const uint8_t *src, size_t n, uint64_t *out);

/* ---- libzstd via dlopen (no headers needed) ---- */
typedef size_t (*fn_compressBound)(size_t);
typedef void *(*fn_createCCtx)(void);
typedef size_t (*fn_freeCCtx)(void *);
typedef size_t (*fn_compressCCtx)(void *, void *, size_t, const void *, size_t, int);
typedef size_t (*fn_decompress)(void *, size_t, const void *, size_t);
typedef unsigned (*fn_isError)(size_t);
typedef unsigned long long (*fn_getFrameContentSize)(const void *, size_t);
static struct {
    void *h;
    fn_compressBound compressBound; fn_createCCtx createCCtx; fn_freeCCtx freeCCtx;
    fn_compressCCtx compressCCtx; fn_decompress decompress; fn_isError isError;
    fn_getFrameContentSize getFrameContentSize;
} Z;
static pthread_once_t z_once = PTHREAD_ONCE_INIT;
static void z_load(void) {
    const char *names[] = {"libzstd.so.1", "libzstd.so", "/lib/x86_64-linux-gnu/libzstd.so.1", NULL};
    int i;
    for (i = 0; names[i] && !Z.h; i++) Z.h = dlopen(names[i], RTLD_NOW | RTLD_LOCAL);
    if (!Z.h) return;
    Z.compressBound = (fn_compressBound)dlsym(Z.h, "ZSTD_compressBound");
    Z.createCCtx = (fn_createCCtx)dlsym(Z.h, "ZSTD_createCCtx");
    Z.freeCCtx = (fn_freeCCtx)dlsym(Z.h, "ZSTD_freeCCtx");
    Z.compressCCtx = (fn_compressCCtx)dlsym(Z.h, "ZSTD_compressCCtx");
    Z.decompress = (fn_decompress)dlsym(Z.h, "ZSTD_decompress");
    Z.isError = (fn_isError)dlsym(Z.h, "ZSTD_isError");
    Z.getFrameContentSize = (fn_getFrameContentSize)dlsym(Z.h, "ZSTD_getFrameContentSize");
    if (!Z.compressBound || !Z.createCCtx || !Z.freeCCtx || !Z.compressCCtx || !Z.decompress ||
        !Z.isError || !Z.getFrameContentSize) { dlclose(Z.h); Z.h = NULL; }
}
int oracle_have_libzstd(void) { pthread_once(&z_once, z_load); return Z.h != NULL; }

size_t oracle_libzstd_compress_bound(size_t n) { pthread_once(&z_once, z_load); return Z.h ? Z.compressBound(n) : 0; }
int64_t oracle_libzstd_compress(uint8_t *dst, size_t cap, const uint8_t *src, size_t n, int level) {
    void *c; size_t r;
    pthread_once(&z_once, z_load);
    if (!Z.h) return -100;
    c = Z.createCCtx();
    r = Z.compressCCtx(c, dst, cap, src, n, level);
    Z.freeCCtx(c);
    return Z.isError(r) ? -101 : (int64_t)r;
}
int64_t oracle_libzstd_decompress(uint8_t *dst, size_t cap, const uint8_t *src, size_t n) {
    size_t r;
    pthread_once(&z_once, z_load);
    if (!Z.h) return -100;
    r = Z.decompress(dst, cap, src, n);
    return Z.isError(r) ? -101 : (int64_t)r;
}

/* ---- read loop ---- */
typedef struct {
    uint64_t total_chunks, total_written_bytes, verified_bytes, corrupt_bytes, corrupt_rows,
        decode_errors;
} oracle_verify_stats;

typedef struct {
    const uint8_t *blobs; const uint64_t *blob_offset, *blob_size, *out_offset, *uncompressed_size;
    const uint8_t *compressed_bitmap, *checksum;
    uint64_t row_end; uint8_t *out; int use_libzstd;
    _Atomic uint64_t *cursor;
    oracle_verify_stats st;
    uint64_t *corrupt_list; size_t corrupt_cap; _Atomic uint64_t *corrupt_n;
    size_t max_out;
} rd_worker_t;

static void *rd_worker(void *arg) {
    rd_worker_t *w = (rd_worker_t *)arg;
    uint8_t *out_buf = (uint8_t *)malloc(w->max_out ? w->max_out : 1); /* reused (codec.rs:L64-66) */
    for (;;) {
        uint64_t row = atomic_fetch_add_explicit(w->cursor, 1, memory_order_relaxed);
        const uint8_t *blob, *res;
        uint64_t bsz, len;
        uint8_t dig[32];
        int compressed;
        if (row >= w->row_end) break;
        w->st.total_chunks++;
        blob = w->blobs + w->blob_offset[row];
        bsz = w->blob_size[row];
        compressed = (w->compressed_bitmap[row >> 3] >> (row & 7)) & 1;
        if (compressed) {
            int64_t r;
            uint64_t dsz;
            if (w->use_libzstd) {
                unsigned long long fcs = Z.getFrameContentSize(blob, bsz);
                if (fcs > w->max_out) { w->st.decode_errors++; continue; }
                r = (int64_t)Z.decompress(out_buf, (size_t)fcs, blob, bsz);
                if (Z.isError((size_t)r)) r = -1;
            } else {
                if (oracle_zstd_get_decompressed_size(blob, bsz, &dsz) || dsz > w->max_out) { w->st.decode_errors++; continue; }
                r = oracle_zstd_decompress(out_buf, (size_t)dsz, blob, bsz);
            }
            if (r < 0) { w->st.decode_errors++; continue; } /* decompress.rs:L159-162: log + continue */
            res = out_buf; len = (uint64_t)r;
        } else {
            res = blob; len = bsz;
        }
        w->st.total_written_bytes += len;
        oracle_blake3(res, len, dig);
        if (memcmp(dig, w->checksum + 32 * row, 32) == 0) {
            w->st.verified_bytes += len;
        } else {
            uint64_t k = atomic_fetch_add(w->corrupt_n, 1);
            w->st.corrupt_bytes += len;
            w->st.corrupt_rows++;
            if (w->corrupt_list && k < w->corrupt_cap) w->corrupt_list[k] = row;
        }
        if (w->out) memcpy(w->out + w->out_offset[row], res, len); /* pwrite analogue; still written on mismatch (L186-189) */
    }
    free(out_buf);
    return NULL;
}

int oracle_decompress_rows(const uint8_t *blobs, const uint64_t *blob_offset, const uint64_t *blob_size,
                           const uint64_t *uncompressed_size, const uint64_t *out_offset,
                           const uint8_t *compressed_bitmap, const uint8_t *checksum,
                           uint64_t row_begin, uint64_t row_end, uint8_t *out, int n_threads,
                           int use_libzstd, oracle_verify_stats *stats, uint64_t *corrupt_list,
                           size_t corrupt_cap) {
    _Atomic uint64_t cursor = row_begin, corrupt_n = 0;
    pthread_t *th;
    rd_worker_t *ws;
    size_t max_out = 0;
    uint64_t r;
    int i;
    if (use_libzstd && !oracle_have_libzstd()) return -100;
    if (n_threads < 1) n_threads = 1;
    for (r = row_begin; r < row_end; r++) if (uncompressed_size[r] > max_out) max_out = uncompressed_size[r];
    th = (pthread_t *)calloc(n_threads, sizeof *th);
    ws = (rd_worker_t *)calloc(n_threads, sizeof *ws);
    for (i = 0; i < n_threads; i++) {
        rd_worker_t *w = &ws[i];
        w->blobs = blobs; w->blob_offset = blob_offset; w->blob_size = blob_size; w->out_offset = out_offset;
        w->uncompressed_size = uncompressed_size; w->compressed_bitmap = compressed_bitmap; w->checksum = checksum;
        w->row_end = row_end; w->out = out; w->use_libzstd = use_libzstd; w->cursor = &cursor;
        w->corrupt_list = corrupt_list; w->corrupt_cap = corrupt_cap; w->corrupt_n = &corrupt_n; w->max_out = max_out;
        pthread_create(&th[i], NULL, rd_worker, w);
    }
    memset(stats, 0, sizeof *stats);
    for (i = 0; i < n_threads; i++) {
        pthread_join(th[i], NULL);
        stats->total_chunks += ws[i].st.total_chunks;
        stats->total_written_bytes += ws[i].st.total_written_bytes;
        stats->verified_bytes += ws[i].st.verified_bytes;
        stats->corrupt_bytes += ws[i].st.corrupt_bytes;
        stats->corrupt_rows += ws[i].st.corrupt_rows;
        stats->decode_errors += ws[i].st.decode_errors;
    }
    free(th); free(ws);
    return 0;
}

/* ---- write loop ---- */
typedef struct {
    const uint8_t *src; const uint64_t *off, *len; const uint8_t *skip;
    uint64_t n_rounds; int level;
    uint8_t *blob_out; size_t blob_cap;
    _Atomic uint64_t *cursor, *out_cursor;
    uint64_t *blob_offset, *blob_size; uint8_t *checksum, *compressed;
    int err;
} wr_worker_t;

static void *wr_worker(void *arg) {
    wr_worker_t *w = (wr_worker_t *)arg;
    void *cctx = Z.createCCtx(); /* one ctx per worker (codec.rs:L16-28) */
    uint8_t *buf = NULL; size_t buf_cap = 0;
    for (;;) {
        uint64_t i = atomic_fetch_add_explicit(w->cursor, 1, memory_order_relaxed), n, o;
        const uint8_t *s, *payload;
        if (i >= w->n_rounds) break;
        s = w->src + w->off[i];
        oracle_blake3(s, w->len[i], w->checksum + 32 * i); /* ORIGINAL bytes, pre-compression */
        if (w->skip[i]) {
            payload = s; n = w->len[i]; w->compressed[i] = 0;
        } else {
            size_t bound = Z.compressBound(w->len[i]), r;
            if (buf_cap < bound) { free(buf); buf = (uint8_t *)malloc(bound); buf_cap = bound; }
            r = Z.compressCCtx(cctx, buf, buf_cap, s, w->len[i], w->level);
            if (Z.isError(r)) { w->err = 1; break; } /* compress error aborts the worker (`?`) */
            payload = buf; n = r; w->compressed[i] = 1;
        }
        o = atomic_fetch_add_explicit(w->out_cursor, n, memory_order_relaxed); /* writer L258 */
        if (o + n > w->blob_cap) { w->err = 2; break; }
        memcpy(w->blob_out + o, payload, n);
        w->blob_offset[i] = o; w->blob_size[i] = n;
    }
    free(buf);
    Z.freeCCtx(cctx);
    return NULL;
}

/* rounds (offset,len,skip) over one staging buffer -> blobs + per-round metadata.
 * Returns total blob bytes (>=0) or <0 on error. */
int64_t oracle_compress_rounds(const uint8_t *src, const uint64_t *off, const uint64_t *len,
                               const uint8_t *skip, uint64_t n_rounds, int level, int n_threads,
                               uint8_t *blob_out, size_t blob_cap, uint64_t *blob_offset,
                               uint64_t *blob_size, uint8_t *checksum, uint8_t *compressed) {
    _Atomic uint64_t cursor = 0, out_cursor = 0;
    pthread_t *th;
    wr_worker_t *ws;
    int i, err = 0;
    if (!oracle_have_libzstd()) return -100;
    if (n_threads < 1) n_threads = 1;
    th = (pthread_t *)calloc(n_threads, sizeof *th);
    ws = (wr_worker_t *)calloc(n_threads, sizeof *ws);
    for (i = 0; i < n_threads; i++) {
        wr_worker_t *w = &ws[i];
        w->src = src; w->off = off; w->len = len; w->skip = skip; w->n_rounds = n_rounds; w->level = level;
        w->blob_out = blob_out; w->blob_cap = blob_cap; w->cursor = &cursor; w->out_cursor = &out_cursor;
        w->blob_offset = blob_offset; w->blob_size = blob_size; w->checksum = checksum; w->compressed = compressed;
        pthread_create(&th[i], NULL, wr_worker, w);
    }
    for (i = 0; i < n_threads; i++) { pthread_join(th[i], NULL); if (ws[i].err) err = ws[i].err; }
    free(th); free(ws);
    return err ? -err : (int64_t)out_cursor;
}
