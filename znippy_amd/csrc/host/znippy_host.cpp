// Compiled host side of the Znippy hot path (include/znippy_host.h): the reference's pipeline entry
// points restated in C++17 over the device C ABI (include/znippy_hip.h).  What runs here is host
// plumbing only — chunking rules, staging, file I/O, index/manifest/footer serialisation, report
// arithmetic; every byte of codec/hash work goes through libznippy_hip's kernels.
#include "../../../include/znippy_host.h"
#include "../../../include/znippy_hip.h"
#include "arrow_ipc.h"

#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <fcntl.h>
#include <map>
#include <memory>
#include <sched.h>
#include <string>
#include <sys/stat.h>
#include <thread>
#include <unistd.h>
#include <unordered_map>
#include <unordered_set>
#include <vector>

namespace {

thread_local std::string g_err;
int fail(int code, const std::string &msg) {
    g_err = msg;
    return code;
}

constexpr uint64_t SLICE_SIZE = 8ull * 1024 * 1024;  // stream_packer.rs:L31
constexpr uint64_t BATCH_BYTES = 1ull << 30;          // staging handed to the GPU at once
constexpr uint64_t RANGE_BYTES = 4ull << 30;          // decoded bytes per GPU hand-off on the read side
const char MAGIC[8] = {'Z', 'N', 'P', 'Y', 'M', 'I', 'D', 'X'};  // index.rs:L245

// is_probably_compressed, index.rs:L470-484 — last extension, case-insensitive
bool should_skip_compression(const std::string &path) {
    static const char *exts[] = {"zip", "gz", "bz2", "xz", "lz", "lzma", "7z", "rar", "cab", "jar", "war", "ear",
                                 "zst", "sz", "lz4", "tgz", "txz", "tbz", "apk", "dmg", "deb", "rpm", "arrow",
                                 "mpeg", "mpg", "jpeg", "jpg", "gif", "bmp", "png", "crate", "znippy", "zdata",
                                 "parquet", "webp", "webm"};
    size_t slash = path.find_last_of('/');
    std::string name = slash == std::string::npos ? path : path.substr(slash + 1);
    size_t dot = name.find_last_of('.');
    if (dot == std::string::npos || dot == 0 || dot + 1 == name.size()) return false;  // ".gz" has no extension
    std::string ext = name.substr(dot + 1);
    for (auto &ch : ext) ch = (char)tolower((unsigned char)ch);
    for (const char *e : exts)
        if (ext == e) return true;
    return false;
}

std::string with_extension(const std::string &path, const char *ext) {  // Path::with_extension
    size_t slash = path.find_last_of('/');
    size_t start = slash == std::string::npos ? 0 : slash + 1;
    std::string name = path.substr(start);
    size_t dot = name.find_last_of('.');
    if (dot != std::string::npos && dot != 0) name = name.substr(0, dot);
    return path.substr(0, start) + name + "." + ext;
}

std::map<std::string, std::string> config_metadata() {  // index.rs:L73-85 over CONFIG (common_config.rs:L25-78)
    unsigned cores = std::thread::hardware_concurrency();
    cpu_set_t set;
    if (sched_getaffinity(0, sizeof set, &set) == 0) cores = (unsigned)CPU_COUNT(&set);
    if (!cores) cores = 1;
    unsigned in_flight = (unsigned)std::ceil(cores * 0.90);
    long pages = sysconf(_SC_PHYS_PAGES), psz = sysconf(_SC_PAGE_SIZE);
    uint64_t mem = pages > 0 && psz > 0 ? (uint64_t)pages * (uint64_t)psz : 0;
    uint64_t max_chunks = std::min<uint64_t>(mem / (10ull * 1024 * 1024), 128);
    return {{"znippy_format_version", "3"},
            {"max_core_in_flight", std::to_string(in_flight)},
            {"max_core_in_compress", std::to_string(cores > in_flight ? cores - in_flight : 0)},
            {"max_mem_allowed", std::to_string(mem)},
            {"min_free_memory_ratio", "0"},
            {"file_split_block_size", std::to_string(10 * 1024 * 1024)},
            {"max_chunks", std::to_string(max_chunks)},
            {"compression_level", "19"},
            {"zstd_output_buffer_size", std::to_string(1024 * 1024)}};
}

aipc::Batch index_schema() {  // index.rs:L43-54
    aipc::Batch b;
    const char *names[8] = {"relative_path", "chunk_seq", "fdata_offset", "compressed", "uncompressed_size",
                            "blob_offset", "blob_size", "checksum"};
    const aipc::Kind kinds[8] = {aipc::Kind::Utf8, aipc::Kind::UInt32, aipc::Kind::UInt64, aipc::Kind::Bool,
                                 aipc::Kind::UInt64, aipc::Kind::UInt64, aipc::Kind::UInt64, aipc::Kind::FixedBin32};
    for (int i = 0; i < 8; i++) {
        aipc::Column c;
        c.name = names[i];
        c.kind = kinds[i];
        b.cols.push_back(c);
    }
    return b;
}

aipc::Batch manifest_schema() {  // index.rs:L279-288
    aipc::Batch b;
    const char *names[6] = {"pkg_type", "repo", "module_name", "index_offset", "index_len", "row_count"};
    const aipc::Kind kinds[6] = {aipc::Kind::Int8, aipc::Kind::Utf8, aipc::Kind::Utf8, aipc::Kind::UInt64,
                                 aipc::Kind::UInt64, aipc::Kind::UInt64};
    for (int i = 0; i < 6; i++) {
        aipc::Column c;
        c.name = names[i];
        c.kind = kinds[i];
        b.cols.push_back(c);
    }
    return b;
}

struct Manifest {
    int8_t pkg_type;
    std::string repo, module_name;
    uint64_t index_offset, index_len, row_count;
};

std::vector<uint8_t> manifest_bytes(const std::vector<Manifest> &m) {
    aipc::Batch b = manifest_schema();
    for (const auto &e : m) {
        b.cols[0].u8.push_back((uint8_t)e.pkg_type);
        b.cols[1].str.push_back(e.repo);
        b.cols[2].str.push_back(e.module_name);
        b.cols[3].u64.push_back(e.index_offset);
        b.cols[4].u64.push_back(e.index_len);
        b.cols[5].u64.push_back(e.row_count);
    }
    return aipc::write_stream({b}, manifest_schema(), {});
}

// ---- device buffer (grow-only) ----
struct DevBuf {
    uint8_t *p = nullptr;
    size_t cap = 0;
    bool reserve(size_t n) {
        if (n <= cap && p) return true;
        if (p) (void)hipFree(p);
        p = nullptr;
        cap = 0;
        size_t want = std::max<size_t>(n + 64, 1 << 20);
        if (hipMalloc(&p, want) != hipSuccess) return false;
        cap = want;
        return true;
    }
    ~DevBuf() { if (p) (void)hipFree(p); }
};

bool pread_all(int fd, void *dst, size_t n, uint64_t off) {
    uint8_t *d = (uint8_t *)dst;
    while (n) {
        ssize_t r = pread(fd, d, n, (off_t)off);
        if (r <= 0) return false;
        d += r; off += (uint64_t)r; n -= (size_t)r;
    }
    return true;
}
bool pwrite_all(int fd, const void *src, size_t n, uint64_t off) {
    const uint8_t *s = (const uint8_t *)src;
    while (n) {
        ssize_t r = pwrite(fd, s, n, (off_t)off);
        if (r <= 0) return false;
        s += r; off += (uint64_t)r; n -= (size_t)r;
    }
    return true;
}

void mkdirs(const std::string &dir) {
    if (dir.empty()) return;
    std::string cur;
    for (size_t i = 0; i < dir.size(); i++) {
        cur += dir[i];
        if (dir[i] == '/' || i + 1 == dir.size()) mkdir(cur.c_str(), 0755);
    }
}

// Contiguous [begin,end) per rank, balanced by sum(weights)
std::pair<uint64_t, uint64_t> split_rows(const std::vector<uint64_t> &w, uint32_t rank, uint32_t world) {
    const uint64_t n = w.size();
    if (world <= 1) return {0, n};
    std::vector<long double> c(n);
    long double acc = 0;
    for (uint64_t i = 0; i < n; i++) { acc += (long double)std::max<uint64_t>(w[i], 1); c[i] = acc; }
    std::vector<uint64_t> cuts(world + 1, n);
    cuts[0] = 0;
    for (uint32_t r = 1; r < world; r++) {
        long double target = acc * r / world;
        cuts[r] = (uint64_t)(std::lower_bound(c.begin(), c.end(), target) - c.begin());
    }
    for (uint32_t r = 1; r <= world; r++) cuts[r] = std::max(cuts[r], cuts[r - 1]);
    return {cuts[rank], cuts[rank + 1]};
}

}  // namespace

struct znippy_index {
    aipc::Batch rows = index_schema();
    std::vector<Manifest> manifest;
    std::map<std::string, std::string> metadata;
    uint64_t blob_end = 0, file_size = 0;
    size_t n() const { return rows.rows(); }
};

static int load_index(const char *path, znippy_index *ix) {
    int fd = open(path, O_RDONLY);
    if (fd < 0) return fail(ZNIPPY_E_INVAL, std::string("cannot open ") + path);
    struct stat st;
    fstat(fd, &st);
    ix->file_size = (uint64_t)st.st_size;
    if (ix->file_size < 16) { close(fd); return fail(ZNIPPY_E_CORRUPT, "file too small to be a v0.7 znippy archive"); }
    uint8_t tail[16];
    pread_all(fd, tail, 16, ix->file_size - 16);
    uint64_t moff;
    if (!znippy_interpret_footer(tail, 16, &moff)) {
        close(fd);
        return fail(ZNIPPY_E_UNSUPPORTED, "v0.6 archives are not supported; re-compress with v0.7");  // index.rs:L387-389
    }
    const uint64_t mend = ix->file_size - 16;
    if (moff > mend) { close(fd); return fail(ZNIPPY_E_CORRUPT, "corrupt v0.7 manifest_offset"); }
    std::vector<uint8_t> mb(mend - moff);
    pread_all(fd, mb.data(), mb.size(), moff);
    aipc::Batch m;
    std::string err;
    if (!aipc::read_stream(mb.data(), mb.size(), &m, nullptr, &err) || m.cols.size() != 6) {
        close(fd);
        return fail(ZNIPPY_E_CORRUPT, "manifest: " + err);
    }
    ix->blob_end = moff;
    for (size_t i = 0; i < m.rows(); i++) {
        Manifest e{(int8_t)m.cols[0].u8[i], m.cols[1].str[i], m.cols[2].str[i], m.cols[3].u64[i], m.cols[4].u64[i],
                   m.cols[5].u64[i]};
        ix->blob_end = std::min(ix->blob_end, e.index_offset);
        ix->manifest.push_back(e);
    }
    ix->rows.cols.clear();
    bool first = true;
    for (const auto &e : ix->manifest) {
        if (e.index_offset + e.index_len > ix->file_size) { close(fd); return fail(ZNIPPY_E_CORRUPT, "sub-index out of range"); }
        std::vector<uint8_t> sb(e.index_len);
        pread_all(fd, sb.data(), sb.size(), e.index_offset);
        if (!aipc::read_stream(sb.data(), sb.size(), &ix->rows, first ? &ix->metadata : nullptr, &err)) {
            close(fd);
            return fail(ZNIPPY_E_CORRUPT, "sub-index: " + err);
        }
        first = false;
    }
    close(fd);
    if (ix->rows.cols.empty()) ix->rows = index_schema();
    // locate the 8 base columns by name (module columns may follow them)
    const aipc::Batch want = index_schema();
    aipc::Batch ordered;
    for (const auto &wc : want.cols) {
        bool found = false;
        for (auto &c : ix->rows.cols)
            if (c.name == wc.name && c.kind == wc.kind) { ordered.cols.push_back(std::move(c)); found = true; break; }
        if (!found) return fail(ZNIPPY_E_CORRUPT, "index is missing column " + wc.name);
    }
    ix->rows = std::move(ordered);
    return ZNIPPY_OK;
}

// ---------------------------------------------------------------------------------------------------
struct Entry {
    std::string path;
    std::vector<uint8_t> data;
    int pkg_type;       // < 0 = None
    bool has_repo;
    std::string repo;
};

struct znippy_stream {
    std::string output;
    bool no_skip;
    int device;
    std::vector<Entry> entries;
};

struct Round {
    uint32_t file_index;
    uint64_t start, len;
    bool skip;
    uint64_t fdata_offset;
    uint32_t chunk_seq;
};

struct RowMeta {  // BlobMeta + ChunkMeta, meta.rs:L4-21
    uint32_t file_index, chunk_seq;
    uint64_t fdata_offset, usize, blob_offset, blob_size;
    bool compressed;
    uint8_t checksum[32];
};

extern "C" {

const char *znippy_host_last_error(void) { return g_err.c_str(); }

int znippy_interpret_footer(const uint8_t *tail, size_t n, uint64_t *offset) {
    if (n < 8) return 0;
    std::memcpy(offset, tail + n - 8, 8);
    return n >= 16 && std::memcmp(tail + n - 16, MAGIC, 8) == 0;
}

size_t znippy_write_manifest_bytes(const znippy_manifest_entry *entries, size_t n, uint8_t *dst, size_t cap) {
    std::vector<Manifest> m;
    for (size_t i = 0; i < n; i++)
        m.push_back({entries[i].pkg_type, entries[i].repo ? entries[i].repo : "", entries[i].module_name ? entries[i].module_name : "",
                     entries[i].index_offset, entries[i].index_len, entries[i].row_count});
    std::vector<uint8_t> b = manifest_bytes(m);
    if (dst && cap) std::memcpy(dst, b.data(), std::min(cap, b.size()));
    return b.size();
}

// ---- write side -------------------------------------------------------------------------------
int znippy_compress_stream(const char *output, int no_skip, int device, znippy_stream **out) {
    if (!output || !out) return fail(ZNIPPY_E_INVAL, "null argument");
    znippy_stream *s = new znippy_stream();
    s->output = output;
    s->no_skip = no_skip != 0;
    s->device = device;
    *out = s;
    return ZNIPPY_OK;
}

int znippy_stream_send(znippy_stream *s, const char *relative_path, const void *data, size_t len, int pkg_type,
                       const char *repo) {
    if (!s || !relative_path || (len && !data)) return fail(ZNIPPY_E_INVAL, "null argument");
    Entry e;
    e.path = relative_path;
    e.data.assign((const uint8_t *)data, (const uint8_t *)data + len);
    e.pkg_type = pkg_type;
    e.has_repo = repo != nullptr;
    if (repo) e.repo = repo;
    s->entries.push_back(std::move(e));
    return ZNIPPY_OK;
}

int znippy_stream_finish(znippy_stream *sp, znippy_compression_report *report) {
    if (!sp) return fail(ZNIPPY_E_INVAL, "null stream");
    std::unique_ptr<znippy_stream> s(sp);
    const std::string out_path = with_extension(s->output, "znippy");  // stream_packer.rs:L132
    // the reader: entries -> Rounds (stream_packer.rs:L146-206)
    std::vector<Round> rounds;
    uint64_t uf = 0, ub = 0, cf = 0, cb = 0;
    for (uint32_t fi = 0; fi < s->entries.size(); fi++) {
        const Entry &e = s->entries[fi];
        const bool skip = !s->no_skip && should_skip_compression(e.path);
        const uint64_t total = e.data.size();
        if (skip) { uf++; ub += total; } else { cf++; cb += total; }
        if (total == 0) { rounds.push_back({fi, 0, 0, skip, 0, 0}); continue; }  // L169-183
        const bool small = total <= SLICE_SIZE;
        uint64_t off = 0;
        uint32_t seq = 0;
        while (off < total) {
            const uint64_t len = small ? total : std::min(SLICE_SIZE, total - off);
            rounds.push_back({fi, off, len, skip, off, seq});
            off += len;
            seq++;
        }
    }
    int fd = open(out_path.c_str(), O_CREAT | O_TRUNC | O_RDWR, 0644);
    if (fd < 0) return fail(ZNIPPY_E_INVAL, "cannot create " + out_path);
    {
        hipError_t e = hipSetDevice(s->device);
        if (e != hipSuccess) { close(fd); return fail(ZNIPPY_E_HIP, std::string("hipSetDevice failed: ") + hipGetErrorString(e)); }
    }
    znippy_ctx *ctx = nullptr;
    int rc = znippy_ctx_create(s->device, nullptr, &ctx);
    if (rc) { close(fd); return fail(rc, "znippy_ctx_create failed"); }
    std::vector<RowMeta> rows;
    rows.reserve(rounds.size());
    uint64_t out_cursor = 0;  // blob region starts at 0 (L134)
    DevBuf d_src, d_blob;
    std::vector<uint8_t> staging, blob;
    size_t i = 0;
    while (i < rounds.size() && rc == ZNIPPY_OK) {
        size_t j = i;
        uint64_t nbytes = 0;
        while (j < rounds.size() && (j == i || nbytes + rounds[j].len <= BATCH_BYTES)) nbytes += rounds[j++].len;
        const size_t nb = j - i;
        staging.resize(nbytes);
        std::vector<uint64_t> off(nb), len(nb), boff(nb), bsz(nb);
        std::vector<uint8_t> skip(nb), comp(nb), ck(32 * nb);
        uint64_t pos = 0;
        for (size_t k = 0; k < nb; k++) {
            const Round &r = rounds[i + k];
            if (r.len) std::memcpy(&staging[pos], s->entries[r.file_index].data.data() + r.start, r.len);
            off[k] = pos; len[k] = r.len; skip[k] = r.skip;
            pos += r.len;
        }
        znippy_rounds *rt = nullptr;
        uint64_t blob_bytes = 0;
        if (!d_src.reserve(nbytes + 64)) rc = ZNIPPY_E_NOMEM;
        if (!rc && nbytes && hipMemcpy(d_src.p, staging.data(), nbytes, hipMemcpyHostToDevice) != hipSuccess) rc = ZNIPPY_E_HIP;
        if (!rc) rc = znippy_rounds_create(ctx, off.data(), len.data(), skip.data(), nb, &rt);
        if (!rc && !d_blob.reserve(znippy_rounds_blob_bound(rt) + 64)) rc = ZNIPPY_E_NOMEM;
        if (!rc) rc = znippy_encode_hash_rounds(ctx, rt, d_src.p, d_blob.p, d_blob.cap, boff.data(), bsz.data(), ck.data(),
                                                comp.data(), &blob_bytes);
        if (rt) znippy_rounds_destroy(rt);
        if (rc) break;
        blob.resize(blob_bytes);
        if (blob_bytes && hipMemcpy(blob.data(), d_blob.p, blob_bytes, hipMemcpyDeviceToHost) != hipSuccess) { rc = ZNIPPY_E_HIP; break; }
        if (!pwrite_all(fd, blob.data(), blob_bytes, out_cursor)) { rc = ZNIPPY_E_INVAL; break; }  // the writer, L255-284
        for (size_t k = 0; k < nb; k++) {
            const Round &r = rounds[i + k];
            RowMeta m{r.file_index, r.chunk_seq, r.fdata_offset, r.len, out_cursor + boff[k], bsz[k], comp[k] != 0, {0}};
            std::memcpy(m.checksum, &ck[32 * k], 32);
            rows.push_back(m);
        }
        out_cursor += blob_bytes;
        i = j;
    }
    if (rc) {
        std::string e = znippy_last_error(ctx);
        znippy_ctx_destroy(ctx);
        close(fd);
        return fail(rc, "compress pipeline failed: " + e);
    }
    znippy_ctx_destroy(ctx);
    // finalizer (L293-346): rows sorted by (file_index, chunk_seq), grouped by (pkg_type, repo) in BTreeMap order
    std::stable_sort(rows.begin(), rows.end(), [](const RowMeta &a, const RowMeta &b) {
        return a.file_index != b.file_index ? a.file_index < b.file_index : a.chunk_seq < b.chunk_seq;
    });
    std::map<std::pair<int8_t, std::string>, std::vector<size_t>> groups;
    for (size_t k = 0; k < rows.size(); k++) {
        const Entry &e = s->entries[rows[k].file_index];
        groups[{(int8_t)(e.pkg_type < 0 ? 0 : e.pkg_type), e.has_repo ? e.repo : std::string()}].push_back(k);
    }
    const auto meta = config_metadata();
    uint64_t cursor = out_cursor;
    std::vector<Manifest> manifest;
    for (const auto &g : groups) {
        aipc::Batch b = index_schema();
        for (size_t k : g.second) {
            const RowMeta &m = rows[k];
            b.cols[0].str.push_back(s->entries[m.file_index].path);
            b.cols[1].u32.push_back(m.chunk_seq);
            b.cols[2].u64.push_back(m.fdata_offset);
            b.cols[3].u8.push_back(m.compressed);
            b.cols[4].u64.push_back(m.usize);
            b.cols[5].u64.push_back(m.blob_offset);
            b.cols[6].u64.push_back(m.blob_size);
            b.cols[7].u8.insert(b.cols[7].u8.end(), m.checksum, m.checksum + 32);
        }
        std::vector<uint8_t> sub = aipc::write_stream({b}, index_schema(), meta);
        pwrite_all(fd, sub.data(), sub.size(), cursor);  // ArrowIpcSink::push_subindex, meta_sink.rs:L71-101
        manifest.push_back({g.first.first, g.first.second, "", cursor, sub.size(), g.second.size()});
        cursor += sub.size();
    }
    const uint64_t manifest_offset = cursor;  // ArrowIpcSink::finish, meta_sink.rs:L103-118
    std::vector<uint8_t> mb = manifest_bytes(manifest);
    pwrite_all(fd, mb.data(), mb.size(), cursor);
    cursor += mb.size();
    pwrite_all(fd, MAGIC, 8, cursor);
    pwrite_all(fd, &manifest_offset, 8, cursor + 8);
    fsync(fd);
    close(fd);
    const uint64_t total_bytes_out = cursor + 16;
    if (report) {
        *report = znippy_compression_report{uf + cf, cf, uf, 0, cb + ub, total_bytes_out, cb, ub, (uint64_t)rows.size(),
                                            (cb > 0 && total_bytes_out > ub) ? (float)cb / (float)(total_bytes_out - ub) * 100.0f : 0.0f};
    }
    return ZNIPPY_OK;
}

// ---- read side --------------------------------------------------------------------------------
static int decode_rows(znippy_ctx *ctx, int arc_fd, const znippy_index &ix, const std::vector<uint64_t> &row_ids, bool verify,
                       DevBuf &d_blobs, DevBuf &d_out, std::vector<uint8_t> &out, std::vector<uint64_t> &out_off,
                       znippy_verify_counters *cnt, std::vector<uint64_t> *corrupt, std::vector<int32_t> *status) {
    const size_t n = row_ids.size();
    std::vector<uint64_t> bo(n), bs(n), us(n);
    std::vector<uint8_t> bitmap((n + 7) / 8, 0), ck(verify ? 32 * n : 0);
    out_off.assign(n, 0);
    uint64_t lo = UINT64_MAX, hi = 0, total = 0;
    for (size_t k = 0; k < n; k++) {
        const uint64_t r = row_ids[k];
        bo[k] = ix.rows.cols[5].u64[r]; bs[k] = ix.rows.cols[6].u64[r]; us[k] = ix.rows.cols[4].u64[r];
        if (ix.rows.cols[3].u8[r]) bitmap[k >> 3] |= (uint8_t)(1u << (k & 7));
        if (verify) std::memcpy(&ck[32 * k], &ix.rows.cols[7].u8[32 * r], 32);
        out_off[k] = total;
        total += us[k];
        lo = std::min(lo, bo[k]);
        hi = std::max(hi, bo[k] + bs[k]);
    }
    if (!n) { out.clear(); return ZNIPPY_OK; }
    if (hi > ix.file_size) return fail(ZNIPPY_E_CORRUPT, "blob range outside the archive");
    std::vector<uint8_t> blobs(hi - lo);
    if (!pread_all(arc_fd, blobs.data(), blobs.size(), lo)) return fail(ZNIPPY_E_INVAL, "failed to read blob from archive");
    if (!d_blobs.reserve(blobs.size() + 64) || !d_out.reserve(total + 64)) return fail(ZNIPPY_E_NOMEM, "device allocation failed");
    if (!blobs.empty() && hipMemcpy(d_blobs.p, blobs.data(), blobs.size(), hipMemcpyHostToDevice) != hipSuccess)
        return fail(ZNIPPY_E_HIP, "H2D failed");
    znippy_rows *rt = nullptr;
    int rc = znippy_rows_create(ctx, bo.data(), bs.data(), bitmap.data(), us.data(), out_off.data(), verify ? ck.data() : nullptr, 0, n, &rt);
    if (rc) return fail(rc, "znippy_rows_create failed");
    std::vector<uint64_t> cr(n);
    status->assign(n, 0);
    rc = znippy_decode_verify_rows(ctx, rt, d_blobs.p, lo, d_out.p, total, cnt, cr.data(), n, status->data());
    znippy_rows_destroy(rt);
    if (rc) return fail(rc, std::string("decode failed: ") + znippy_last_error(ctx));
    out.resize(total);
    if (total && hipMemcpy(out.data(), d_out.p, total, hipMemcpyDeviceToHost) != hipSuccess) return fail(ZNIPPY_E_HIP, "D2H failed");
    if (corrupt)
        for (uint64_t k = 0; k < cnt->corrupt_rows && k < n; k++) corrupt->push_back(row_ids[cr[k]]);
    return ZNIPPY_OK;
}

int znippy_decompress_archive(const char *index_path, int save_data, const char *out_dir, int device, uint32_t rank,
                              uint32_t world, znippy_verify_report *report, uint64_t *corrupt_rows, uint64_t corrupt_cap,
                              uint64_t *n_corrupt) {
    if (!index_path || !report || (save_data && !out_dir) || world == 0 || rank >= world) return fail(ZNIPPY_E_INVAL, "bad argument");
    znippy_index ix;
    int rc = load_index(index_path, &ix);
    if (rc) return rc;
    const size_t total_rows = ix.n();
    std::unordered_set<std::string> uniq(ix.rows.cols[0].str.begin(), ix.rows.cols[0].str.end());  // L64-69
    const auto range = split_rows(ix.rows.cols[4].u64, rank, world);
    (void)total_rows;
    // pre-create output files (L74-101)
    std::unordered_map<std::string, int> fds;
    if (save_data) {
        for (uint64_t r = range.first; r < range.second; r++) {
            const std::string &p = ix.rows.cols[0].str[r];
            if (fds.count(p)) continue;
            std::string full = std::string(out_dir) + "/" + p;
            size_t slash = full.find_last_of('/');
            mkdirs(full.substr(0, slash));
            int fd = open(full.c_str(), O_CREAT | O_WRONLY | (world > 1 ? 0 : O_TRUNC), 0644);
            if (fd < 0) return fail(ZNIPPY_E_INVAL, "failed to open output file " + full);
            fds[p] = fd;
        }
    }
    int arc = open(index_path, O_RDONLY);
    if (arc < 0) return fail(ZNIPPY_E_INVAL, "cannot open archive");
    znippy_ctx *ctx = nullptr;
    if (range.second > range.first) {
        if (hipSetDevice(device) != hipSuccess || (rc = znippy_ctx_create(device, nullptr, &ctx))) {
            close(arc);
            return fail(rc ? rc : ZNIPPY_E_HIP, "no usable GPU: the codec/hash path has no CPU fallback");
        }
    }
    znippy_verify_counters tot{};
    std::vector<uint64_t> corrupt;
    DevBuf d_blobs, d_out;
    std::vector<uint8_t> out;
    uint64_t i = range.first;
    while (i < range.second && rc == ZNIPPY_OK) {
        uint64_t j = i, nbytes = 0;
        while (j < range.second && (j == i || nbytes + ix.rows.cols[4].u64[j] <= RANGE_BYTES)) nbytes += ix.rows.cols[4].u64[j++];
        std::vector<uint64_t> ids(j - i), out_off;
        for (uint64_t k = i; k < j; k++) ids[k - i] = k;
        znippy_verify_counters c{};
        std::vector<int32_t> status;
        rc = decode_rows(ctx, arc, ix, ids, true, d_blobs, d_out, out, out_off, &c, &corrupt, &status);
        if (rc) break;
        tot.total_chunks += c.total_chunks; tot.total_written_bytes += c.total_written_bytes;
        tot.verified_bytes += c.verified_bytes; tot.corrupt_bytes += c.corrupt_bytes;
        tot.corrupt_rows += c.corrupt_rows; tot.decode_errors += c.decode_errors;
        if (save_data) {
            for (uint64_t k = i; k < j; k++) {
                if (status[k - i] < 0) {  // decode error: logged + skipped (L159-162)
                    fprintf(stderr, "[decomp] row %llu error=%d\n", (unsigned long long)k, status[k - i]);
                    continue;
                }
                pwrite_all(fds[ix.rows.cols[0].str[k]], out.data() + out_off[k - i], ix.rows.cols[4].u64[k], ix.rows.cols[2].u64[k]);
            }
        }
        i = j;
    }
    for (auto &kv : fds) close(kv.second);
    close(arc);
    if (ctx) znippy_ctx_destroy(ctx);
    if (rc) return rc;
    std::sort(corrupt.begin(), corrupt.end());
    for (uint64_t r : corrupt) fprintf(stderr, "[verify] MISMATCH row=%llu\n", (unsigned long long)r);
    const uint64_t corrupt_files = corrupt.size();  // number of corrupt ROWS (L210)
    report->total_files = uniq.size();
    report->corrupt_files = corrupt_files;
    report->verified_files = uniq.size() > corrupt_files ? uniq.size() - corrupt_files : 0;
    report->total_bytes = tot.total_written_bytes;
    report->verified_bytes = tot.verified_bytes;
    report->corrupt_bytes = tot.corrupt_bytes;
    report->chunks = tot.total_chunks;
    if (n_corrupt) *n_corrupt = corrupt.size();
    if (corrupt_rows)
        for (uint64_t k = 0; k < corrupt.size() && k < corrupt_cap; k++) corrupt_rows[k] = corrupt[k];
    return ZNIPPY_OK;
}

// ---- ZnippyArchive ------------------------------------------------------------------------------
}  // extern "C"

struct znippy_archive {
    std::string path;
    int device = 0;
    znippy_index ix;
    std::unordered_map<std::string, std::vector<uint64_t>> files;  // rows sorted by fdata_offset (archive.rs:L131-133)
    znippy_ctx *ctx = nullptr;
    int fd = -1;
    DevBuf d_blobs, d_out;
};

extern "C" {

int znippy_archive_open(const char *path, int device, znippy_archive **out) {
    if (!path || !out) return fail(ZNIPPY_E_INVAL, "null argument");
    std::unique_ptr<znippy_archive> a(new znippy_archive());
    a->path = path;
    a->device = device;
    int rc = load_index(path, &a->ix);
    if (rc) return rc;
    for (uint64_t r = 0; r < a->ix.n(); r++) a->files[a->ix.rows.cols[0].str[r]].push_back(r);
    for (auto &kv : a->files)
        std::stable_sort(kv.second.begin(), kv.second.end(), [&](uint64_t x, uint64_t y) {
            return a->ix.rows.cols[2].u64[x] < a->ix.rows.cols[2].u64[y];
        });
    a->fd = open(path, O_RDONLY);
    if (a->fd < 0) return fail(ZNIPPY_E_INVAL, "cannot open archive");
    *out = a.release();
    return ZNIPPY_OK;
}

uint64_t znippy_archive_file_count(const znippy_archive *a) { return a ? a->files.size() : 0; }

int64_t znippy_archive_file_size(const znippy_archive *a, const char *rel) {
    if (!a || !rel) return -1;
    auto it = a->files.find(rel);
    if (it == a->files.end()) return -1;
    uint64_t s = 0;
    for (uint64_t r : it->second) s += a->ix.rows.cols[4].u64[r];
    return (int64_t)s;
}

int znippy_archive_extract_file(znippy_archive *a, const char *rel, void *dst, size_t cap, size_t *written) {
    if (!a || !rel || !written) return fail(ZNIPPY_E_INVAL, "null argument");
    auto it = a->files.find(rel);
    if (it == a->files.end()) return fail(ZNIPPY_E_INVAL, std::string("file not found in archive: ") + rel);
    if (!a->ctx) {
        if (hipSetDevice(a->device) != hipSuccess) return fail(ZNIPPY_E_HIP, "hipSetDevice failed");
        int rc = znippy_ctx_create(a->device, nullptr, &a->ctx);
        if (rc) return fail(rc, "no usable GPU: the codec/hash path has no CPU fallback");
    }
    std::vector<uint8_t> out;
    std::vector<uint64_t> out_off;
    std::vector<int32_t> status;
    znippy_verify_counters c{};
    int rc = decode_rows(a->ctx, a->fd, a->ix, it->second, false, a->d_blobs, a->d_out, out, out_off, &c, nullptr, &status);
    if (rc) return rc;
    for (int32_t st : status)
        if (st < 0) return fail(st, "OpenZL-equivalent decompress failed");  // propagates (archive.rs:L160)
    if (out.size() > cap) return fail(ZNIPPY_E_DST_SMALL, "destination too small");
    if (!out.empty()) std::memcpy(dst, out.data(), out.size());
    *written = out.size();
    return ZNIPPY_OK;
}

void znippy_archive_close(znippy_archive *a) {
    if (!a) return;
    if (a->ctx) znippy_ctx_destroy(a->ctx);
    if (a->fd >= 0) close(a->fd);
    delete a;
}

// ---- index ------------------------------------------------------------------------------------
int znippy_index_open(const char *path, znippy_index **out) {
    if (!path || !out) return fail(ZNIPPY_E_INVAL, "null argument");
    std::unique_ptr<znippy_index> ix(new znippy_index());
    int rc = load_index(path, ix.get());
    if (rc) return rc;
    *out = ix.release();
    return ZNIPPY_OK;
}
uint64_t znippy_index_rows(const znippy_index *ix) { return ix ? ix->n() : 0; }
uint64_t znippy_index_manifest_len(const znippy_index *ix) { return ix ? ix->manifest.size() : 0; }
int znippy_index_manifest_entry(const znippy_index *ix, uint64_t i, znippy_manifest_entry *out) {
    if (!ix || !out || i >= ix->manifest.size()) return ZNIPPY_E_INVAL;
    const Manifest &m = ix->manifest[i];
    *out = znippy_manifest_entry{m.pkg_type, m.repo.c_str(), m.module_name.c_str(), m.index_offset, m.index_len, m.row_count};
    return ZNIPPY_OK;
}
int znippy_index_row(const znippy_index *ix, uint64_t i, const char **relative_path, uint32_t *chunk_seq, uint64_t *fdata_offset,
                     int *compressed, uint64_t *uncompressed_size, uint64_t *blob_offset, uint64_t *blob_size,
                     const uint8_t **checksum32) {
    if (!ix || i >= ix->n()) return ZNIPPY_E_INVAL;
    const auto &c = ix->rows.cols;
    if (relative_path) *relative_path = c[0].str[i].c_str();
    if (chunk_seq) *chunk_seq = c[1].u32[i];
    if (fdata_offset) *fdata_offset = c[2].u64[i];
    if (compressed) *compressed = c[3].u8[i];
    if (uncompressed_size) *uncompressed_size = c[4].u64[i];
    if (blob_offset) *blob_offset = c[5].u64[i];
    if (blob_size) *blob_size = c[6].u64[i];
    if (checksum32) *checksum32 = &c[7].u8[32 * i];
    return ZNIPPY_OK;
}
const char *znippy_index_metadata(const znippy_index *ix, const char *key) {
    if (!ix || !key) return nullptr;
    auto it = ix->metadata.find(key);
    return it == ix->metadata.end() ? nullptr : it->second.c_str();
}
void znippy_index_close(znippy_index *ix) { delete ix; }

}  // extern "C"
