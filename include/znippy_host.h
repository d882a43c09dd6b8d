/*
 * znippy_host.h — compiled host side of the Znippy hot path (C ABI), layered on znippy_hip.h.
 * Mirrors the reference's pipeline entry points for this path; every function cites what it stands for
 * (paths relative to the reference root).  The reference is Rust; with no Rust toolchain in the image the
 * host layer is C++17 (znippy_amd/csrc/host/), callable from Rust through the same `extern "C"` surface.
 *
 *   znippy_compress_stream / znippy_stream_send / znippy_stream_finish
 *        compress_stream(&PathBuf, no_skip) -> StreamCompressor{sender(), finish()}
 *        znippy-compress/src/stream_packer.rs:L58-87, pipeline L127-372 (reader chunking L146-206,
 *        writer L255-284, finalizer L293-346)
 *   znippy_compress_dir            compress_dir(&input_dir, &output, no_skip, plugin = None, repo) -> CompressionReport
 *        znippy-compress/src/slot_packer.rs:L55-209 (walk L63-78, partition L92-101, big pass rounds L265-280,
 *        small pass rounds L499, one sub-index of both passes L141-189)
 *   znippy_decompress_archive      decompress_archive(index_path, save_data, out_dir) -> VerifyReport
 *        znippy-common/src/decompress.rs:L39-222
 *   znippy_archive_*               ZnippyArchive::{open,file_count,contains/file_size,extract_file}
 *        znippy-common/src/archive.rs:L53-168
 *   znippy_index_*                 read_znippy_index / read_znippy_manifest / interpret_footer
 *        znippy-common/src/index.rs:L269-277,L374-468
 */
#ifndef ZNIPPY_HOST_H
#define ZNIPPY_HOST_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct {  /* CompressionReport, znippy-common/src/lib.rs:L39-51 */
    uint64_t total_files, compressed_files, uncompressed_files, total_dirs;
    uint64_t total_bytes_in, total_bytes_out, compressed_bytes, uncompressed_bytes, chunks;
    float compression_ratio;
} znippy_compression_report;

typedef struct {  /* VerifyReport, znippy-common/src/index.rs:L490-499 */
    uint64_t total_files, verified_files, corrupt_files;
    uint64_t total_bytes, verified_bytes, corrupt_bytes, chunks;
} znippy_verify_report;

typedef struct {  /* ManifestEntry, index.rs:L248-256 (strings owned by the handle) */
    int8_t pkg_type;
    const char *repo, *module_name;
    uint64_t index_offset, index_len, row_count;
} znippy_manifest_entry;

typedef struct znippy_stream znippy_stream;
typedef struct znippy_archive znippy_archive;
typedef struct znippy_index znippy_index;

/* Text of the last error on the calling thread ("" if none). */
const char *znippy_host_last_error(void);

/* ---- write side ---- */
int znippy_compress_stream(const char *output, int no_skip, int device, znippy_stream **out);
/* pkg_type < 0 = None, repo NULL = None (ArchiveEntry, stream_packer.rs:L34-44). Data is copied (once, into
 * page-locked staging); full staging slots are encoded while the caller keeps sending. */
int znippy_stream_send(znippy_stream *s, const char *relative_path, const void *data, size_t len,
                       int pkg_type, const char *repo);
/* The same for n entries in one call (a caller that already holds its entries packed — the reference's reader stage
 * hands StreamCompressor whole batches, stream_packer.rs:L146-206 — or a binding whose per-call cost matters):
 * entry i has the path paths[path_off[i] .. path_off[i+1]) (no terminator) and the bytes data[data_off[i] .. data_off[i+1]);
 * pkg_type may be NULL (= None for all), repo NULL = None for all.  Stops at the first error. */
int znippy_stream_send_packed(znippy_stream *s, uint64_t n, const char *paths, const uint64_t *path_off,
                              const void *data, const uint64_t *data_off, const int32_t *pkg_type, const char *repo);
/* Drains the pipeline, writes the metadata layer of `<output>.znippy`, fills the report and frees the handle. */
int znippy_stream_finish(znippy_stream *s, znippy_compression_report *report);

/* Directory ingest.  repo NULL = None.  Files are read by a few host threads straight into page-locked staging;
 * metadata plugins are outside this path. */
int znippy_compress_dir(const char *input_dir, const char *output, int no_skip, const char *repo, int device,
                        znippy_compression_report *report);

/* ---- read side ---- */
/* rank/world split the row cursor into contiguous ranges balanced by uncompressed bytes (world = 1:
 * everything).  Counters are this rank's share; total_files is global.  corrupt_rows receives the
 * absolute row numbers with a checksum mismatch (ascending). */
int znippy_decompress_archive(const char *index_path, int save_data, const char *out_dir, int device,
                              uint32_t rank, uint32_t world, znippy_verify_report *report,
                              uint64_t *corrupt_rows, uint64_t corrupt_cap, uint64_t *n_corrupt);

int znippy_archive_open(const char *path, int device, znippy_archive **out);
uint64_t znippy_archive_file_count(const znippy_archive *a);
int64_t znippy_archive_file_size(const znippy_archive *a, const char *relative_path); /* -1: not found */
int znippy_archive_extract_file(znippy_archive *a, const char *relative_path, void *dst, size_t cap,
                                size_t *written);
/* The same with every chunk's BLAKE3 checked against the index's checksum column (the reference's extract_file has
 * no verification, archive.rs:L144-168; SURVEY §8f rank 1 asks for it): ZNIPPY_E_CHECKSUM on a mismatch. */
int znippy_archive_extract_file_verified(znippy_archive *a, const char *relative_path, void *dst, size_t cap,
                                         size_t *written);
void znippy_archive_close(znippy_archive *a);

/* ---- index / container ---- */
int znippy_index_open(const char *path, znippy_index **out);  /* footer -> manifest -> all sub-indexes */
uint64_t znippy_index_rows(const znippy_index *ix);
uint64_t znippy_index_manifest_len(const znippy_index *ix);
int znippy_index_manifest_entry(const znippy_index *ix, uint64_t i, znippy_manifest_entry *out);
/* Column access for row i (pointers stay valid until close). */
int znippy_index_row(const znippy_index *ix, uint64_t i, const char **relative_path, uint32_t *chunk_seq,
                     uint64_t *fdata_offset, int *compressed, uint64_t *uncompressed_size,
                     uint64_t *blob_offset, uint64_t *blob_size, const uint8_t **checksum32);
/* Value of a schema-metadata key of the first sub-index (NULL if absent). */
const char *znippy_index_metadata(const znippy_index *ix, const char *key);
void znippy_index_close(znippy_index *ix);

/* interpret_footer (index.rs:L269-277): returns 1 = multi (v0.7), 0 = single (v0.6); *offset = payload. */
int znippy_interpret_footer(const uint8_t *tail, size_t n, uint64_t *offset);

/* Serialise / parse a manifest stream (write_manifest_bytes / read_manifest_bytes, index.rs:L291-367).
 * write: returns the byte count, copies up to cap bytes into dst. */
size_t znippy_write_manifest_bytes(const znippy_manifest_entry *entries, size_t n, uint8_t *dst, size_t cap);

#ifdef __cplusplus
}
#endif
#endif
