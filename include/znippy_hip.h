/*
 * znippy_hip.h — C ABI of the MI355X-native per-chunk codec + hash path for the Znippy
 * archive format.  Plain pointers and sizes only; no C++/torch/HIP types in signatures
 * (a HIP stream is passed as `void*`).  Integer status codes, no exceptions, caller-
 * allocated outputs.  Thread-safe for concurrent calls on DISTINCT contexts.
 *
 * What each entry point replaces in the reference (paths relative to the reference root):
 *   znippy_compress_bound        zl_compress_bound             znippy-common/src/codec.rs:L32,L45
 *   znippy_compress              CompressCtx::compress_into    znippy-common/src/codec.rs:L43-55
 *   znippy_get_decompressed_size zl_get_decompressed_size      znippy-common/src/codec.rs:L69
 *   znippy_decompress            codec::decompress_into        znippy-common/src/codec.rs:L67-78
 *   znippy_blake3                blake3::hash                  stream_packer.rs:L219, slot_packer.rs:L553,
 *                                                              decompress.rs:L172
 *   znippy_rows_* + znippy_decode_verify_rows
 *                                body of the read worker loop  znippy-common/src/decompress.rs:L135-190
 *                                (+ stats merge L195-221)      over index columns (index.rs:L43-54)
 *   znippy_rounds_* + znippy_encode_hash_rounds
 *                                barrel + writer bodies        znippy-compress/src/stream_packer.rs:L217-284,
 *                                                              znippy-compress/src/slot_packer.rs:L551-609
 *   znippy_hash_rounds           blake3-only / store path      slot_packer.rs:L553-560 (skip branch)
 *
 * Codec wire format: one standard Zstandard frame (RFC 8878) per chunk.  (The reference's
 * OpenZL framing cannot be reproduced or checked offline — see DESIGN.md "Oracle".)
 */
#ifndef ZNIPPY_HIP_H
#define ZNIPPY_HIP_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ZNIPPY_OK 0
#define ZNIPPY_E_INVAL (-1)       /* bad argument */
#define ZNIPPY_E_HIP (-2)         /* HIP runtime error (znippy_last_error has the text) */
#define ZNIPPY_E_NOMEM (-3)
#define ZNIPPY_E_DST_SMALL (-4)   /* caller buffer too small */
#define ZNIPPY_E_CORRUPT (-5)     /* malformed frame */
#define ZNIPPY_E_UNSUPPORTED (-6) /* dictionary frames, unknown content size */
#define ZNIPPY_E_CHECKSUM (-7)    /* frame content checksum (XXH64) mismatch */

typedef struct znippy_ctx znippy_ctx;
typedef struct znippy_rows znippy_rows;     /* read side: a range of index rows, device-resident */
typedef struct znippy_rounds znippy_rounds; /* write side: a batch of Rounds, device-resident */

/* One context per worker/GPU (the analogue of one CompressCtx per thread, codec.rs:L8-28).
 * `hip_stream` may be NULL (the context then owns a non-blocking stream). */
int znippy_ctx_create(int device, void *hip_stream, znippy_ctx **out);
/* Lifetime rule: a table (znippy_rows, znippy_rounds) holds a reference to the context it was created on.
 * znippy_ctx_destroy on a context that still has tables CLOSES it — every later call that takes this context, with or
 * without a table, returns ZNIPPY_E_INVAL and touches nothing — and the context's memory and device resources are
 * released when its last table is destroyed (znippy_rows_destroy / znippy_rounds_destroy are always safe to call, in
 * any order relative to znippy_ctx_destroy).  Without tables it is released at once.  Calling it twice is harmless
 * while tables keep it alive; after the release the pointer is dangling, as with any destroy call. */
void znippy_ctx_destroy(znippy_ctx *ctx);
/* CompressCtx::new(compression_level) (znippy-common/src/codec.rs:L16-28): the effort of every later encode call of
 * this context.  1..22; a new context starts at 19 (CONFIG.compression_level, common_config.rs:L37).  Two tiers:
 * levels 1-3 use the fast block matcher (one probe per position, raw literals beyond 128 symbols, predefined
 * sequence tables), levels 4-22 the higher effort one (4-way buckets, lazy choice, in-block repeat offsets,
 * per-block entropy tables).  Frames of both tiers are plain RFC 8878.  znippy_ctx_level returns the level. */
int znippy_ctx_set_level(znippy_ctx *ctx, int level);
int znippy_ctx_level(const znippy_ctx *ctx);
const char *znippy_last_error(const znippy_ctx *ctx);
/* Block until everything queued on the context's stream has finished. */
int znippy_ctx_sync(znippy_ctx *ctx);

/* ---- (1) bounds ---------------------------------------------------------------------- */
size_t znippy_compress_bound(size_t n);

/* ---- (5) single-chunk synchronous shims, HOST buffers, codec.rs semantics -------------- */
int znippy_get_decompressed_size(const void *frame, size_t n, uint64_t *out_size);
int znippy_decompress(znippy_ctx *ctx, const void *frame, size_t n, void *dst, size_t cap,
                      size_t *written);
int znippy_compress(znippy_ctx *ctx, const void *src, size_t n, void *dst, size_t cap,
                    size_t *written);
int znippy_blake3(znippy_ctx *ctx, const void *src, size_t n, uint8_t out[32]);

/* ---- (3) batch decode + verify over a row range ---------------------------------------- */
/* Counters of the read loop (WorkerStats, decompress.rs:L22-28; VerifyReport is derived from
 * them on the host, decompress.rs:L195-221).  decode_errors = rows whose frame failed to
 * decode: counted in total_chunks and in no byte counter (decompress.rs:L140,L159-162). */
typedef struct {
    uint64_t total_chunks;
    uint64_t total_written_bytes;
    uint64_t verified_bytes;
    uint64_t corrupt_bytes;
    uint64_t corrupt_rows;
    uint64_t decode_errors;
} znippy_verify_counters;

/* Upload rows [row_begin,row_end) of the index columns (HOST pointers, indexed by absolute
 * row number, exactly the Arrow buffers the reference downcasts at decompress.rs:L115-129):
 *   blob_offset, blob_size, uncompressed_size : u64 per row
 *   compressed_bitmap                         : Arrow boolean bitmap, LSB-first (NULL = all compressed)
 *   checksum                                  : 32 bytes per row (NULL = no verification)
 *   out_offset                                : u64 per row, byte position of the row's decoded
 *                                               bytes in the caller's flat output region
 *                                               (stands for (file, fdata_offset), L186-189)
 * Also builds the work plan (tiles of <=64 BLAKE3 leaves) that drives the kernels' cursor. */
int znippy_rows_create(znippy_ctx *ctx, const uint64_t *blob_offset, const uint64_t *blob_size,
                       const uint8_t *compressed_bitmap, const uint64_t *uncompressed_size,
                       const uint64_t *out_offset, const uint8_t *checksum, uint64_t row_begin,
                       uint64_t row_end, znippy_rows **out);
void znippy_rows_destroy(znippy_rows *rows);
/* Declare the size in bytes of the blob region the table will be run against (d_blobs of the calls below).  Every
 * run validates each row on the host, once per distinct (blob_base, blob_cap, out_cap): a row whose blob does not
 * lie inside [blob_base, blob_base + blob_cap), or whose bytes would not fit inside out_cap, is NOT touched by any
 * kernel and reports ZNIPPY_E_CORRUPT / ZNIPPY_E_DST_SMALL in row_status (counted as a decode error) — a crafted
 * or damaged index is an error code, never a device fault.  Without this call only the output side is checked.
 * A stored row (compressed = 0) is its blob: its length is blob_size, as in the reference (decompress.rs:L143-166). */
int znippy_rows_set_blob_cap(znippy_rows *rows, uint64_t blob_cap);

/* Decode-or-passthrough + BLAKE3 + compare for every row of the table.
 *   d_blobs   : DEVICE pointer to the blob region; row r's blob is d_blobs[blob_offset[r]-blob_base ..]
 *   d_out     : DEVICE pointer to the flat output region (row r lands at d_out + out_offset[r]);
 *               out_cap = its size in bytes
 *   counters  : HOST, filled after the call (the call synchronises the stream)
 *   corrupt_rows / corrupt_cap : HOST list receiving absolute row numbers whose checksum
 *               mismatched (ascending); may be NULL
 *   row_status: HOST, optional (NULL ok): one int32 per row of the table, 0 = decoded, <0 = ZNIPPY_E_*
 * Asynchronous variant: znippy_decode_verify_rows_async queues the work only; results are read
 * back with znippy_rows_results after znippy_ctx_sync.
 * Device memory a context takes for this call beyond the table's own columns (kept until the context goes): pools of the
 * batch path sized from the table's content (literals 1 byte, sequence records 1.5, decoding tables 1 per content byte of
 * its compressed rows, each capped at 16 GiB) and — only for tables with compressed rows above 64 KiB — 4 bytes per byte of
 * those rows for the resolve path, capped at 8 GiB (ZNIPPY_NO_RX=1 in the environment of znippy_ctx_create: none, such
 * frames are then executed by one wave each).  A pool that cannot be allocated is not an error: its path is not used.
 * The bytes in d_out belong to the caller once a results call for that run has returned (znippy_rows_results,
 * znippy_rows_results_lagged, znippy_rows_digests — the synchronous call ends in one): a table whose previous run needed
 * nothing behind its main kernel is run without the kernels that stand behind it, and a run that turns out to have needed
 * them after all (the blobs changed) is repeated in full, with the arguments it was given, inside the first results call
 * that looks at it.  d_blobs / d_out of a queued run must therefore stay valid until its results have been read
 * (ZNIPPY_NO_LEAN=1 in the environment of znippy_ctx_create: every run is a full one). */
int znippy_decode_verify_rows(znippy_ctx *ctx, znippy_rows *rows, const void *d_blobs,
                              uint64_t blob_base, void *d_out, uint64_t out_cap,
                              znippy_verify_counters *counters, uint64_t *corrupt_rows,
                              uint64_t corrupt_cap, int32_t *row_status);
int znippy_decode_verify_rows_async(znippy_ctx *ctx, znippy_rows *rows, const void *d_blobs,
                                    uint64_t blob_base, void *d_out, uint64_t out_cap);
int znippy_rows_results(znippy_ctx *ctx, znippy_rows *rows, znippy_verify_counters *counters,
                        uint64_t *corrupt_rows, uint64_t corrupt_cap, int32_t *row_status);
/* Counters of the run `lag` (0 or 1) runs before the latest one queued on this table: waits for THAT run only,
 * so a caller that keeps two runs in flight reads run k's counters while run k+1 executes — the read loop
 * reports after the loop, not per row (decompress.rs:L195-221).  ZNIPPY_E_INVAL if no such run exists. */
int znippy_rows_results_lagged(znippy_ctx *ctx, znippy_rows *rows, unsigned lag, znippy_verify_counters *counters);
/* Computed digests of the last run (HOST, 32 bytes per row of the table). */
int znippy_rows_digests(znippy_ctx *ctx, znippy_rows *rows, uint8_t *digests);

/* ---- (2)+(4) batch encode + hash over Rounds -------------------------------------------- */
/* A Round is (offset,len,skip) into one staging buffer (slotpool.rs:L39-47,
 * stream_packer.rs:L98-106).  HOST arrays, n entries. */
int znippy_rounds_create(znippy_ctx *ctx, const uint64_t *src_offset, const uint64_t *len,
                         const uint8_t *skip, uint64_t n, znippy_rounds **out);
void znippy_rounds_destroy(znippy_rounds *rounds);
/* Opt-in content-based store path (the reference's wish list, TODO_NOW.md:L37-38):
 * a round whose frame would not be smaller than its input is emitted as-is with compressed = 0.
 * OFF by default — the reference decides by file extension only (index.rs:L470-488), so turning this
 * on changes the `compressed` column for incompressible rounds. */
int znippy_rounds_set_store_incompressible(znippy_rounds *rounds, int on);
/* Upper bound of the blob bytes znippy_encode_hash_rounds can produce for this batch. */
uint64_t znippy_rounds_blob_bound(const znippy_rounds *rounds);

/* For every round: checksum = BLAKE3(src slice) (pre-compression bytes); skip -> the raw bytes
 * are the payload (compressed=0), else one zstd frame (compressed=1).  Payloads are packed
 * back-to-back from d_blob_out[0] in round order (blob offsets are a running sum — the
 * writer's out_cursor.fetch_add, stream_packer.rs:L258).
 *   d_src      : DEVICE staging buffer the rounds point into
 *   d_blob_out : DEVICE, blob_cap bytes (>= znippy_rounds_blob_bound)
 *   HOST outputs, n entries each: blob_offset, blob_size (= on_disk_len), checksum (32 B each),
 *   compressed (0/1).  *blob_bytes = total payload bytes. */
int znippy_encode_hash_rounds(znippy_ctx *ctx, znippy_rounds *rounds, const void *d_src,
                              void *d_blob_out, uint64_t blob_cap, uint64_t *blob_offset,
                              uint64_t *blob_size, uint8_t *checksum, uint8_t *compressed,
                              uint64_t *blob_bytes);
int znippy_encode_hash_rounds_async(znippy_ctx *ctx, znippy_rounds *rounds, const void *d_src,
                                    void *d_blob_out, uint64_t blob_cap);
int znippy_rounds_results(znippy_ctx *ctx, znippy_rounds *rounds, uint64_t *blob_offset,
                          uint64_t *blob_size, uint8_t *checksum, uint8_t *compressed,
                          uint64_t *blob_bytes);

/* Zero-copy results: pointers into the table's pinned host mirror (one D2H), valid until the next
 * encode call on the same table. */
int znippy_rounds_results_view(znippy_ctx *ctx, znippy_rounds *rounds, const uint64_t **blob_offset,
                               const uint64_t **blob_size, const uint8_t **checksum, uint64_t *blob_bytes);

/* The same for the run `lag` (0 or 1) runs before the latest one; waits for that run's result copy only (every
 * run's results leave on a copy stream into their own pinned mirror).  Pointers stay valid until two more encode
 * calls have been queued on the table. */
int znippy_rounds_results_lagged(znippy_ctx *ctx, znippy_rounds *rounds, unsigned lag, const uint64_t **blob_offset,
                                 const uint64_t **blob_size, const uint8_t **checksum, uint64_t *blob_bytes);

/* Hash only (store path / verify-only): digests[i] = BLAKE3(d_src[off_i .. off_i+len_i]).
 * digests: HOST, 32 bytes per round. */
int znippy_hash_rounds(znippy_ctx *ctx, znippy_rounds *rounds, const void *d_src, uint8_t *digests);

/* ---- measurement hooks (bench.py): device time of the last async call's kernels, by HIP
 * events on the context's stream.  names/ms: up to cap entries; returns the count. */
int znippy_last_kernel_times(znippy_ctx *ctx, const char **names, float *ms, int cap);
/* How many of those event pairs a call records: 2 = around every kernel (default), 1 = around the dominant read
 * kernels only (decode_verify_*), 0 = none (a caller that never asks for kernel times: each pair costs the stream
 * two markers).  ZNIPPY_KTIME in the environment sets the initial level. */
int znippy_ctx_set_kernel_timing(znippy_ctx *ctx, int level);
/* Statistics of the last run's two-phase path for foreign multi-block frames (frames another zstd writer produced;
 * codec.rs:L67-78 decodes whatever the archive holds): stats[0] literal-pool bytes and stats[1] sequence-pool records
 * handed out, stats[2] frames decoded by that path, stats[3] blocks it left to the serial decoder, of which stats[4]
 * for an error the serial decoder will report, stats[5] a Treeless / Repeat_Mode table more than 64 blocks back,
 * stats[6] a pool that ran out, stats[7] a value outside the record format.  Synchronises. */
int znippy_rows_foreign_stats(znippy_ctx *ctx, znippy_rows *rows, uint64_t stats[8]);
/* The hash's VALU floor, measured: nanoseconds one 64-lane BLAKE3 compress pass costs a SIMD when nothing else runs
 * (a kernel of compressions only, 4 waves per SIMD on every CU), and the shader clock that kernel held (may be NULL).
 * bench.py prices the read step's passes with it. */
int znippy_measure_blake3_pass_ns(znippy_ctx *ctx, float *ns_per_pass_per_simd, float *shader_ghz);
/* Shader clock (GHz) one wave of the read side's small-row kernel saw during the last run of a context created with
 * ZNIPPY_DBG bit 32768 set: its life in shader cycles / in 100 MHz ticks.  0 if nothing was recorded. */
int znippy_last_shader_ghz(znippy_ctx *ctx, float *ghz);

#ifdef __cplusplus
}
#endif
#endif /* ZNIPPY_HIP_H */
