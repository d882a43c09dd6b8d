// Compiled host side of the Znippy hot path (include/znippy_host.h): the reference's pipeline entry
// points restated in C++17 over the device C ABI (include/znippy_hip.h).  What runs here is host
// plumbing only — chunking rules, staging, file I/O, index/manifest/footer serialisation, report
// arithmetic; every byte of codec/hash work goes through libznippy_hip's kernels.
#include "../../../include/znippy_host.h"
#include "../../../include/znippy_hip.h"
#include "arrow_ipc.h"

#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <condition_variable>
#include <deque>
#include <dirent.h>
#include <mutex>
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
constexpr uint64_t SLOT_SIZE = 200ull * 1024 * 1024;  // slot_packer.rs:L30
constexpr int N_STAGE = 3;                            // pinned staging slots in flight on the write side
constexpr size_t MAX_SLOT_ROUNDS = 1u << 20;

uint64_t env_mb(const char *name, uint64_t dflt_mb) {
    const char *v = getenv(name);
    const uint64_t mb = v && *v ? strtoull(v, nullptr, 10) : 0;
    return (mb ? mb : dflt_mb) << 20;
}
// Bytes per GPU hand-off: one pinned staging slot on the write side, one decoded range on the read side.
uint64_t stage_bytes() { return env_mb("ZNIPPY_HOST_SLOT_MB", 128); }
uint64_t range_bytes(bool save) { return env_mb("ZNIPPY_HOST_RANGE_MB", save ? 256 : 1024); }
// File I/O helper threads.  Readers (compress_dir: open/pread/close, lstat) scale to ~16 threads; writers that
// CREATE files in one directory contend on its lock and are fastest at ~4 (tmpfs, 100k files: 1 writer 0.50 s,
// 2: 0.41 s, 4: 0.38 s, 8: 0.52 s, 16: 0.78 s, 32: 1.2 s).  ZNIPPY_HOST_READERS / ZNIPPY_HOST_WRITERS override.
unsigned env_threads(const char *name, unsigned dflt) {
    unsigned hw = std::thread::hardware_concurrency();
    cpu_set_t set;
    if (sched_getaffinity(0, sizeof set, &set) == 0) hw = (unsigned)CPU_COUNT(&set);
    const char *v = getenv(name);
    unsigned want = v && *v ? (unsigned)strtoul(v, nullptr, 10) : dflt;
    return std::max(1u, std::min(want, hw ? hw : 1u));
}
unsigned reader_threads() { return env_threads("ZNIPPY_HOST_READERS", 16); }
unsigned writer_threads() { return env_threads("ZNIPPY_HOST_WRITERS", 4); }

double now_s() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
bool trace_on() { const char *v = getenv("ZNIPPY_HOST_TRACE"); return v && *v && *v != '0'; }
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

// ---- page-locked host buffer (grow-only): DMA source/target, faulted in once and reused ----
struct PinBuf {
    uint8_t *p = nullptr;
    size_t cap = 0;
    bool reserve(size_t n) {
        if (n <= cap && p) return true;
        if (p) (void)hipHostFree(p);
        p = nullptr;
        cap = 0;
        size_t want = std::max<size_t>(n, 1 << 20);
        if (hipHostMalloc((void **)&p, want, hipHostMallocDefault) != hipSuccess) { p = nullptr; return false; }
        cap = want;
        return true;
    }
    ~PinBuf() { if (p) (void)hipHostFree(p); }
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

// A file is untrusted input: every length is checked against the file before anything is allocated or indexed,
// every read is checked, the manifest's columns are checked against the schema before they are indexed.
static int load_index_impl(const char *path, znippy_index *ix) {
    int fd = open(path, O_RDONLY);
    if (fd < 0) return fail(ZNIPPY_E_INVAL, std::string("cannot open ") + path);
    struct stat st;
    if (fstat(fd, &st) != 0) { close(fd); return fail(ZNIPPY_E_INVAL, std::string("cannot stat ") + path); }
    ix->file_size = (uint64_t)st.st_size;
    if (ix->file_size < 16) { close(fd); return fail(ZNIPPY_E_CORRUPT, "file too small to be a v0.7 znippy archive"); }
    uint8_t tail[16];
    if (!pread_all(fd, tail, 16, ix->file_size - 16)) { close(fd); return fail(ZNIPPY_E_CORRUPT, "cannot read the footer"); }
    uint64_t moff;
    if (!znippy_interpret_footer(tail, 16, &moff)) {
        close(fd);
        return fail(ZNIPPY_E_UNSUPPORTED, "v0.6 archives are not supported; re-compress with v0.7");  // index.rs:L387-389
    }
    const uint64_t mend = ix->file_size - 16;
    if (moff > mend) { close(fd); return fail(ZNIPPY_E_CORRUPT, "corrupt v0.7 manifest_offset"); }
    std::vector<uint8_t> mb(mend - moff);
    if (!mb.empty() && !pread_all(fd, mb.data(), mb.size(), moff)) { close(fd); return fail(ZNIPPY_E_CORRUPT, "cannot read the manifest"); }
    aipc::Batch m;
    std::string err;
    if (!aipc::read_stream(mb.data(), mb.size(), &m, nullptr, &err) || m.cols.size() != 6) {
        close(fd);
        return fail(ZNIPPY_E_CORRUPT, "manifest: " + err);
    }
    {  // the six manifest columns, by kind and with equal row counts (index.rs:L279-288)
        const aipc::Batch want = manifest_schema();
        const size_t nr = m.cols[0].rows();
        for (size_t c = 0; c < 6; c++)
            if (m.cols[c].kind != want.cols[c].kind || m.cols[c].rows() != nr) {
                close(fd);
                return fail(ZNIPPY_E_CORRUPT, "manifest: column " + want.cols[c].name + " has the wrong type or length");
            }
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
        if (e.index_offset > ix->file_size || e.index_len > ix->file_size - e.index_offset) { close(fd); return fail(ZNIPPY_E_CORRUPT, "sub-index out of range"); }
        std::vector<uint8_t> sb(e.index_len);
        if (!sb.empty() && !pread_all(fd, sb.data(), sb.size(), e.index_offset)) { close(fd); return fail(ZNIPPY_E_CORRUPT, "cannot read a sub-index"); }
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
    const size_t nr = ix->rows.cols[0].rows();
    for (const auto &c : ix->rows.cols)
        if (c.rows() != nr) return fail(ZNIPPY_E_CORRUPT, "index column " + c.name + " has the wrong length");
    return ZNIPPY_OK;
}

// No C++ exception crosses the C ABI: allocation failures and anything a parser throws become status codes.
template <class F>
static int guarded(F &&f) {
    try {
        return f();
    } catch (const std::bad_alloc &) {
        return fail(ZNIPPY_E_NOMEM, "out of host memory");
    } catch (const std::exception &e) {
        return fail(ZNIPPY_E_CORRUPT, std::string("malformed input: ") + e.what());
    } catch (...) {
        return fail(ZNIPPY_E_CORRUPT, "malformed input");
    }
}

static int load_index(const char *path, znippy_index *ix) {
    return guarded([&] { return load_index_impl(path, ix); });
}

namespace {

// ---------------------------------------------------------------------------------------------------
// Write side.  Rounds are laid into page-locked staging slots in reservation order; a device thread takes
// full slots (H2D -> encode+hash kernels -> D2H of the packed blobs -> pwrite) while the producer fills the
// next one.  Stands for the reader -> barrel -> writer channel chain of stream_packer.rs:L146-284 and the
// Magazine of slot_packer.rs:L222-330: the channels carry slots instead of Rounds, the barrels are kernels.
struct RoundRec {
    uint32_t file_index, chunk_seq;
    uint64_t fdata_offset, len, pos;  // pos = byte position in the slot
    bool skip;
    uint8_t pass;                     // compress_dir: 0 = big pass, 1 = small pass
};

struct RowMeta {  // BlobMeta + ChunkMeta, meta.rs:L4-21
    uint32_t file_index, chunk_seq;
    uint64_t fdata_offset, usize, blob_offset, blob_size;
    bool compressed;
    uint8_t pass;
    uint8_t checksum[32];
};

struct StageSlot {
    PinBuf buf;
    size_t used = 0;
    std::vector<RoundRec> rounds;
    std::atomic<int> pending{0};  // asynchronous reads still landing in this slot
};

class Packer {
public:
    Packer(int fd, int device, uint64_t max_round) : fd_(fd), device_(device), slot_cap_(std::max(stage_bytes(), max_round)) {}
    ~Packer() {
        if (thread_.joinable()) { push_full(nullptr); thread_.join(); }
        if (ctx_) znippy_ctx_destroy(ctx_);
    }
    int start() {
        hipError_t e = hipSetDevice(device_);
        if (e != hipSuccess) return fail(ZNIPPY_E_HIP, std::string("hipSetDevice failed: ") + hipGetErrorString(e));
        int rc = znippy_ctx_create(device_, nullptr, &ctx_);
        if (rc) return fail(rc, "znippy_ctx_create failed: no usable GPU (the codec/hash path has no CPU fallback)");
        for (auto &s : slots_) free_.push_back(&s);
        thread_ = std::thread([this] { run(); });
        return ZNIPPY_OK;
    }
    // Reserve r.len bytes for the next round.  async = the bytes arrive later (reader threads call landed()).
    uint8_t *reserve(RoundRec r, bool async, StageSlot **slot_out = nullptr) {
        if (!cur_ || cur_->used + r.len > cur_->buf.cap || cur_->rounds.size() >= MAX_SLOT_ROUNDS) {
            if (cur_) push_full(cur_);
            cur_ = pop_free();
            if (!cur_->buf.reserve(slot_cap_)) { set_error(ZNIPPY_E_NOMEM, "page-locked staging allocation failed"); return nullptr; }
        }
        r.pos = cur_->used;
        cur_->rounds.push_back(r);
        cur_->used += r.len;
        if (async && r.len) cur_->pending.fetch_add(1, std::memory_order_relaxed);
        if (slot_out) *slot_out = cur_;
        return cur_->buf.p + r.pos;
    }
    // True when the next reserve(len) has to hand the current slot over (and may wait for a free one).
    bool would_switch(uint64_t len) const { return !cur_ || cur_->used + len > cur_->buf.cap || cur_->rounds.size() >= MAX_SLOT_ROUNDS; }
    void landed(StageSlot *s) {
        if (s->pending.fetch_sub(1, std::memory_order_acq_rel) == 1) {  // lock so that the wake cannot slip past the waiter's check
            std::lock_guard<std::mutex> g(mu_);
            cv_.notify_all();
        }
    }
    void set_error(int rc, const std::string &msg) {
        std::lock_guard<std::mutex> g(mu_);
        if (!rc_) { rc_ = rc; err_ = msg; }
    }
    int error() { std::lock_guard<std::mutex> g(mu_); return rc_; }
    // Flush, wait for the device thread, return the pipeline status (message via fail()).
    int finish() {
        if (cur_) { push_full(cur_); cur_ = nullptr; }
        push_full(nullptr);
        thread_.join();
        if (trace_on())
            fprintf(stderr, "[host] write: slots %d  wait %.1f ms  h2d %.1f ms  kernels %.1f ms  d2h %.1f ms  pwrite %.1f ms\n", n_slots_,
                    t_wait_ * 1e3, t_h2d_ * 1e3, t_kern_ * 1e3, t_d2h_ * 1e3, t_write_ * 1e3);
        return rc_ ? fail(rc_, "compress pipeline failed: " + err_) : ZNIPPY_OK;
    }
    std::vector<RowMeta> rows;
    uint64_t out_cursor = 0;  // blob region starts at 0 (stream_packer.rs:L134)

private:
    void push_full(StageSlot *s) {
        std::lock_guard<std::mutex> g(mu_);
        full_.push_back(s);
        cv_.notify_all();
    }
    StageSlot *pop_free() {
        std::unique_lock<std::mutex> g(mu_);
        cv_.wait(g, [this] { return !free_.empty(); });
        StageSlot *s = free_.front();
        free_.pop_front();
        return s;
    }
    void run() {
        (void)hipSetDevice(device_);
        for (;;) {
            StageSlot *s;
            const double t0 = now_s();
            {
                std::unique_lock<std::mutex> g(mu_);
                cv_.wait(g, [this] { return !full_.empty(); });
                s = full_.front();
                full_.pop_front();
                if (s) cv_.wait(g, [s] { return s->pending.load(std::memory_order_acquire) == 0; });
            }
            t_wait_ += now_s() - t0;
            if (!s) return;
            if (!error()) {
                std::string msg;
                int rc = process(*s, &msg);
                if (rc) set_error(rc, msg);
            }
            s->used = 0;
            s->rounds.clear();
            std::lock_guard<std::mutex> g(mu_);
            free_.push_back(s);
            cv_.notify_all();
        }
    }
    int process(StageSlot &s, std::string *msg) {
        const size_t nb = s.rounds.size();
        if (!nb) return ZNIPPY_OK;
        n_slots_++;
        std::vector<uint64_t> off(nb), len(nb), boff(nb), bsz(nb);
        std::vector<uint8_t> skip(nb), comp(nb), ck(32 * nb);
        for (size_t k = 0; k < nb; k++) { off[k] = s.rounds[k].pos; len[k] = s.rounds[k].len; skip[k] = s.rounds[k].skip; }
        double t = now_s();
        if (!d_src_.reserve(s.buf.cap + 64)) { *msg = "device staging allocation failed"; return ZNIPPY_E_NOMEM; }
        if (s.used && hipMemcpy(d_src_.p, s.buf.p, s.used, hipMemcpyHostToDevice) != hipSuccess) { *msg = "H2D failed"; return ZNIPPY_E_HIP; }
        t_h2d_ += now_s() - t; t = now_s();
        znippy_rounds *rt = nullptr;
        uint64_t blob_bytes = 0;
        int rc = znippy_rounds_create(ctx_, off.data(), len.data(), skip.data(), nb, &rt);
        if (!rc && !d_blob_.reserve(znippy_rounds_blob_bound(rt) + 64)) rc = ZNIPPY_E_NOMEM;
        if (!rc) rc = znippy_encode_hash_rounds(ctx_, rt, d_src_.p, d_blob_.p, d_blob_.cap, boff.data(), bsz.data(), ck.data(), comp.data(), &blob_bytes);
        if (rt) znippy_rounds_destroy(rt);
        if (rc) { *msg = znippy_last_error(ctx_); return rc; }
        t_kern_ += now_s() - t; t = now_s();
        if (!blob_pin_.reserve(blob_bytes)) { *msg = "page-locked blob allocation failed"; return ZNIPPY_E_NOMEM; }
        if (blob_bytes && hipMemcpy(blob_pin_.p, d_blob_.p, blob_bytes, hipMemcpyDeviceToHost) != hipSuccess) { *msg = "D2H failed"; return ZNIPPY_E_HIP; }
        t_d2h_ += now_s() - t; t = now_s();
        if (!pwrite_all(fd_, blob_pin_.p, blob_bytes, out_cursor)) { *msg = "archive write failed"; return ZNIPPY_E_INVAL; }  // the writer, L255-284
        t_write_ += now_s() - t;
        for (size_t k = 0; k < nb; k++) {
            const RoundRec &r = s.rounds[k];
            RowMeta m{r.file_index, r.chunk_seq, r.fdata_offset, r.len, out_cursor + boff[k], bsz[k], comp[k] != 0, r.pass, {0}};
            std::memcpy(m.checksum, &ck[32 * k], 32);
            rows.push_back(m);
        }
        out_cursor += blob_bytes;
        return ZNIPPY_OK;
    }

    int fd_, device_;
    uint64_t slot_cap_;
    znippy_ctx *ctx_ = nullptr;
    StageSlot slots_[N_STAGE];
    StageSlot *cur_ = nullptr;
    std::deque<StageSlot *> free_, full_;
    std::mutex mu_;
    std::condition_variable cv_;
    std::thread thread_;
    int rc_ = 0;
    std::string err_;
    DevBuf d_src_, d_blob_;
    PinBuf blob_pin_;
    int n_slots_ = 0;
    double t_wait_ = 0, t_h2d_ = 0, t_kern_ = 0, t_d2h_ = 0, t_write_ = 0;
};

// The metadata layer: one Arrow-IPC sub-index stream per group, then manifest + footer (ArrowIpcSink,
// meta_sink.rs:L71-118).  Returns total_bytes_out.
struct SubIndex {
    int8_t pkg_type;
    std::string repo;
    std::vector<std::vector<size_t>> batches;  // row numbers per record batch
};

template <class PathOf>
uint64_t write_metadata(int fd, uint64_t blob_end, const std::vector<RowMeta> &rows, PathOf path_of, const std::vector<SubIndex> &subs) {
    const auto meta = config_metadata();
    uint64_t cursor = blob_end;
    std::vector<Manifest> manifest;
    for (const auto &g : subs) {
        std::vector<aipc::Batch> bs;
        uint64_t n_rows = 0;
        for (const auto &ids : g.batches) {
            aipc::Batch b = index_schema();
            for (size_t k : ids) {
                const RowMeta &m = rows[k];
                b.cols[0].str.push_back(path_of(m.file_index));
                b.cols[1].u32.push_back(m.chunk_seq);
                b.cols[2].u64.push_back(m.fdata_offset);
                b.cols[3].u8.push_back(m.compressed);
                b.cols[4].u64.push_back(m.usize);
                b.cols[5].u64.push_back(m.blob_offset);
                b.cols[6].u64.push_back(m.blob_size);
                b.cols[7].u8.insert(b.cols[7].u8.end(), m.checksum, m.checksum + 32);
            }
            n_rows += ids.size();
            bs.push_back(std::move(b));
        }
        std::vector<uint8_t> sub = aipc::write_stream(bs, index_schema(), meta);
        pwrite_all(fd, sub.data(), sub.size(), cursor);  // push_subindex, meta_sink.rs:L71-101
        manifest.push_back({g.pkg_type, g.repo, "", cursor, sub.size(), n_rows});
        cursor += sub.size();
    }
    const uint64_t manifest_offset = cursor;  // finish, meta_sink.rs:L103-118
    std::vector<uint8_t> mb = manifest_bytes(manifest);
    pwrite_all(fd, mb.data(), mb.size(), cursor);
    cursor += mb.size();
    pwrite_all(fd, MAGIC, 8, cursor);
    pwrite_all(fd, &manifest_offset, 8, cursor + 8);
    fsync(fd);
    return cursor + 16;
}

struct FileMeta {
    std::string path;
    int pkg_type;  // < 0 = None
    bool has_repo;
    std::string repo;
};

}  // namespace

struct znippy_stream {
    std::string out_path;
    bool no_skip = false;
    int fd = -1;
    std::unique_ptr<Packer> packer;
    std::vector<FileMeta> files;
    uint64_t uf = 0, ub = 0, cf = 0, cb = 0;
    double t_open = 0, t_send = 0;
    ~znippy_stream() {
        packer.reset();
        if (fd >= 0) close(fd);
    }
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

// ---- write side: compress_stream --------------------------------------------------------------
static int znippy_compress_stream_impl(const char *output, int no_skip, int device, znippy_stream **out) {
    if (!output || !out) return fail(ZNIPPY_E_INVAL, "null argument");
    std::unique_ptr<znippy_stream> s(new znippy_stream());
    s->t_open = now_s();
    s->out_path = with_extension(output, "znippy");  // stream_packer.rs:L132
    s->no_skip = no_skip != 0;
    s->fd = open(s->out_path.c_str(), O_CREAT | O_TRUNC | O_RDWR, 0644);
    if (s->fd < 0) return fail(ZNIPPY_E_INVAL, "cannot create " + s->out_path);
    s->packer.reset(new Packer(s->fd, device, SLICE_SIZE));
    int rc = s->packer->start();
    if (rc) return rc;
    *out = s.release();
    return ZNIPPY_OK;
}

// The reader's chunking (stream_packer.rs:L146-206) runs on the caller's thread: each entry is cut into
// Rounds and copied once, straight into page-locked staging.
static int znippy_stream_send_impl(znippy_stream *s, const char *relative_path, const void *data, size_t len, int pkg_type,
                       const char *repo) {
    if (!s || !relative_path || (len && !data)) return fail(ZNIPPY_E_INVAL, "null argument");
    const double t0 = now_s();
    const uint32_t fi = (uint32_t)s->files.size();
    s->files.push_back({relative_path, pkg_type, repo != nullptr, repo ? repo : ""});
    const bool skip = !s->no_skip && should_skip_compression(s->files.back().path);
    if (skip) { s->uf++; s->ub += len; } else { s->cf++; s->cb += len; }
    if (len == 0) {  // L169-183: one zero-length row
        if (!s->packer->reserve({fi, 0, 0, 0, 0, skip, 0}, false)) return fail(ZNIPPY_E_NOMEM, "staging allocation failed");
    } else {
        const bool small = len <= SLICE_SIZE;
        uint64_t off = 0;
        uint32_t seq = 0;
        while (off < len) {
            const uint64_t n = small ? len : std::min<uint64_t>(SLICE_SIZE, len - off);
            uint8_t *dst = s->packer->reserve({fi, seq, off, n, 0, skip, 0}, false);
            if (!dst) return fail(ZNIPPY_E_NOMEM, "staging allocation failed");
            std::memcpy(dst, (const uint8_t *)data + off, n);
            off += n;
            seq++;
        }
    }
    s->t_send += now_s() - t0;
    return ZNIPPY_OK;
}

static int znippy_stream_send_packed_impl(znippy_stream *s, uint64_t n, const char *paths, const uint64_t *path_off, const void *data,
                                          const uint64_t *data_off, const int32_t *pkg_type, const char *repo) {
    if (!s || (n && (!paths || !path_off || !data_off))) return fail(ZNIPPY_E_INVAL, "null argument");
    std::string path;
    for (uint64_t i = 0; i < n; i++) {
        if (path_off[i + 1] < path_off[i] || data_off[i + 1] < data_off[i]) return fail(ZNIPPY_E_INVAL, "offsets must not decrease");
        path.assign(paths + path_off[i], paths + path_off[i + 1]);
        const uint64_t len = data_off[i + 1] - data_off[i];
        if (len && !data) return fail(ZNIPPY_E_INVAL, "null data");
        const int rc = znippy_stream_send_impl(s, path.c_str(), len ? (const uint8_t *)data + data_off[i] : nullptr, (size_t)len,
                                               pkg_type ? (int)pkg_type[i] : -1, repo);
        if (rc) return rc;
    }
    return ZNIPPY_OK;
}

static int znippy_stream_finish_impl(znippy_stream *sp, znippy_compression_report *report) {
    if (!sp) return fail(ZNIPPY_E_INVAL, "null stream");
    std::unique_ptr<znippy_stream> s(sp);
    const double t0 = now_s();
    int rc = s->packer->finish();
    if (rc) return rc;
    const double t1 = now_s();
    std::vector<RowMeta> &rows = s->packer->rows;
    // finalizer (L293-346): rows sorted by (file_index, chunk_seq), grouped by (pkg_type, repo) in BTreeMap order
    std::stable_sort(rows.begin(), rows.end(), [](const RowMeta &a, const RowMeta &b) {
        return a.file_index != b.file_index ? a.file_index < b.file_index : a.chunk_seq < b.chunk_seq;
    });
    std::map<std::pair<int8_t, std::string>, std::vector<size_t>> groups;
    for (size_t k = 0; k < rows.size(); k++) {
        const FileMeta &e = s->files[rows[k].file_index];
        groups[{(int8_t)(e.pkg_type < 0 ? 0 : e.pkg_type), e.has_repo ? e.repo : std::string()}].push_back(k);
    }
    std::vector<SubIndex> subs;
    for (auto &g : groups) subs.push_back({g.first.first, g.first.second, {std::move(g.second)}});
    const uint64_t total_bytes_out =
        write_metadata(s->fd, s->packer->out_cursor, rows, [&](uint32_t fi) -> const std::string & { return s->files[fi].path; }, subs);
    if (trace_on())
        fprintf(stderr, "[host] compress_stream: open->finish %.1f ms (send calls %.1f ms)  drain %.1f ms  metadata %.1f ms\n",
                (t0 - s->t_open) * 1e3, s->t_send * 1e3, (t1 - t0) * 1e3, (now_s() - t1) * 1e3);
    if (report) {
        *report = znippy_compression_report{s->uf + s->cf, s->cf, s->uf, 0, s->cb + s->ub, total_bytes_out, s->cb, s->ub, (uint64_t)rows.size(),
                                            (s->cb > 0 && total_bytes_out > s->ub) ? (float)s->cb / (float)(total_bytes_out - s->ub) * 100.0f : 0.0f};
    }
    return ZNIPPY_OK;
}

// ---- write side: compress_dir (slot_packer.rs:L55-209) ------------------------------------------
}  // extern "C"

namespace {

// WalkDir order (L63-78): a directory's files, then its sub-directories, both in name order; only regular
// files are ingested, every directory (the root too) is counted.
void walk_dir(const std::string &dir, std::vector<std::string> *files, uint64_t *n_dirs) {
    DIR *d = opendir(dir.c_str());
    if (!d) return;
    (*n_dirs)++;
    std::vector<std::string> fs, ds;
    while (struct dirent *e = readdir(d)) {
        const std::string name = e->d_name;
        if (name == "." || name == "..") continue;
        if (e->d_type == DT_DIR) { ds.push_back(name); continue; }
        if (e->d_type == DT_REG) { fs.push_back(name); continue; }
        if (e->d_type != DT_UNKNOWN) continue;
        struct stat st;  // a file system that does not report entry types
        if (lstat((dir + "/" + name).c_str(), &st) != 0) continue;
        if (S_ISDIR(st.st_mode)) ds.push_back(name);
        else if (S_ISREG(st.st_mode)) fs.push_back(name);
    }
    closedir(d);
    std::sort(fs.begin(), fs.end());
    std::sort(ds.begin(), ds.end());
    for (const auto &f : fs) files->push_back(dir + "/" + f);
    for (const auto &sub : ds) walk_dir(dir + "/" + sub, files, n_dirs);
}

// file sizes, a few threads at a time (100k lstat calls are the larger half of a serial walk)
void stat_sizes(const std::vector<std::string> &files, std::vector<uint64_t> *sizes) {
    sizes->assign(files.size(), 0);
    const unsigned nt = std::max(1u, std::min<unsigned>(reader_threads(), (unsigned)(files.size() / 1024 + 1)));
    std::vector<std::thread> th;
    for (unsigned t = 0; t < nt; t++)
        th.emplace_back([&, t] {
            for (size_t i = t; i < files.size(); i += nt) {
                struct stat st;
                if (lstat(files[i].c_str(), &st) == 0) (*sizes)[i] = (uint64_t)st.st_size;
            }
        });
    for (auto &x : th) x.join();
}

struct ReadTask {
    uint32_t file_index;
    uint64_t off, len;
    uint8_t *dst;
    StageSlot *slot;
};

// File readers feeding the staging slots (the io_uring reader's job, slot_packer.rs:L236-330, as plain
// positional reads on a few threads — the device path downstream is what this library is about).
class ReaderPool {
public:
    ReaderPool(Packer *pk, const std::vector<std::string> *files, unsigned n) : pk_(pk), files_(files) {
        for (unsigned i = 0; i < n; i++) th_.emplace_back([this] { run(); });
    }
    // Tasks are handed over in batches (one lock + wake per batch, not per file); flush() before anything that
    // may wait for a slot to drain.
    void push(const ReadTask &t) {
        batch_.push_back(t);
        if (batch_.size() >= 256) flush();
    }
    void flush() {
        if (batch_.empty()) return;
        std::lock_guard<std::mutex> g(mu_);
        q_.insert(q_.end(), batch_.begin(), batch_.end());
        batch_.clear();
        cv_.notify_all();
    }
    void join() {
        flush();
        {
            std::lock_guard<std::mutex> g(mu_);
            done_ = true;
            cv_.notify_all();
        }
        for (auto &t : th_) t.join();
        th_.clear();
    }
    ~ReaderPool() { if (!th_.empty()) join(); }

private:
    void run() {
        int fd = -1;
        uint32_t cur = UINT32_MAX;
        for (;;) {
            ReadTask ts[16];
            int nt = 0;
            {
                std::unique_lock<std::mutex> g(mu_);
                cv_.wait(g, [this] { return done_ || !q_.empty(); });
                if (q_.empty()) break;
                while (nt < 16 && !q_.empty()) { ts[nt++] = q_.front(); q_.pop_front(); }
            }
            for (int i = 0; i < nt; i++) {
                const ReadTask &t = ts[i];
                if (t.file_index != cur) {
                    if (fd >= 0) close(fd);
                    fd = open((*files_)[t.file_index].c_str(), O_RDONLY);
                    cur = t.file_index;
                }
                if (fd < 0 || !pread_all(fd, t.dst, t.len, t.off)) pk_->set_error(ZNIPPY_E_INVAL, "failed to read " + (*files_)[t.file_index]);
                pk_->landed(t.slot);
            }
        }
        if (fd >= 0) close(fd);
    }
    Packer *pk_;
    const std::vector<std::string> *files_;
    std::vector<std::thread> th_;
    std::deque<ReadTask> q_;
    std::vector<ReadTask> batch_;
    std::mutex mu_;
    std::condition_variable cv_;
    bool done_ = false;
};

}  // namespace

extern "C" {

static int znippy_compress_dir_impl(const char *input_dir, const char *output, int no_skip, const char *repo, int device,
                        znippy_compression_report *report) {
    if (!input_dir || !output) return fail(ZNIPPY_E_INVAL, "null argument");
    const double t_begin = now_s();
    std::string root = input_dir;
    while (root.size() > 1 && root.back() == '/') root.pop_back();
    std::vector<std::string> files;
    std::vector<uint64_t> sizes;
    uint64_t total_dirs = 0;
    walk_dir(root, &files, &total_dirs);
    stat_sizes(files, &sizes);
    const double t_walk = now_s();
    unsigned cores = std::thread::hardware_concurrency();
    cpu_set_t set;
    if (sched_getaffinity(0, sizeof set, &set) == 0) cores = (unsigned)CPU_COUNT(&set);
    const uint64_t num_workers = std::max<uint64_t>((uint64_t)std::ceil(std::max(cores, 1u) * 0.90), 1);  // CONFIG.max_core_in_flight
    const uint64_t slice_size = SLOT_SIZE / num_workers;  // L89
    const std::string out_path = with_extension(output, "znippy");
    int fd = open(out_path.c_str(), O_CREAT | O_TRUNC | O_RDWR, 0644);
    if (fd < 0) return fail(ZNIPPY_E_INVAL, "cannot create " + out_path);
    uint64_t uf = 0, ub = 0, cf = 0, cb = 0;
    std::vector<RowMeta> rows;
    uint64_t blob_bytes = 0;
    {
        Packer pk(fd, device, std::max<uint64_t>(slice_size, 1));
        int rc = pk.start();
        if (rc) { close(fd); return rc; }
        ReaderPool readers(&pk, &files, reader_threads());
        bool ok = true;
        for (int pass = 0; pass < 2 && ok; pass++) {  // partition L92-101: pass 0 = big (or empty) files, pass 1 = small
            for (uint32_t i = 0; i < files.size() && ok; i++) {
                const uint64_t size = sizes[i];
                const bool big = size > slice_size || size == 0;
                if (big != (pass == 0)) continue;
                const bool skip = !no_skip && should_skip_compression(files[i]);
                if (skip) { uf++; ub += size; } else { cf++; cb += size; }
                uint64_t off = 0;
                uint32_t seq = 0;
                do {  // big: slice_size rounds with fdata_offset/chunk_seq (L265-280); small: one round (L499)
                    const uint64_t n = std::min(slice_size, size - off);
                    StageSlot *slot = nullptr;
                    if (pk.would_switch(n)) readers.flush();
                    uint8_t *dst = pk.reserve({i, seq, off, n, 0, skip, (uint8_t)pass}, true, &slot);
                    if (!dst) { ok = false; break; }
                    if (n) readers.push({i, off, n, dst, slot});
                    off += n;
                    seq++;
                } while (off < size);
            }
        }
        readers.join();
        rc = pk.finish();
        if (rc) { close(fd); return rc; }
        rows = std::move(pk.rows);
        blob_bytes = pk.out_cursor;
    }
    const double t_pack = now_s();
    // one sub-index, one record batch per non-empty pass, rows in round order (L141-189)
    SubIndex sub{0, repo ? repo : "", {}};
    for (uint8_t pass = 0; pass < 2; pass++) {
        std::vector<size_t> ids;
        for (size_t k = 0; k < rows.size(); k++)
            if (rows[k].pass == pass) ids.push_back(k);
        if (!ids.empty()) sub.batches.push_back(std::move(ids));
    }
    std::vector<std::string> rel(files.size());
    for (size_t i = 0; i < files.size(); i++) rel[i] = files[i].compare(0, root.size() + 1, root + "/") == 0 ? files[i].substr(root.size() + 1) : files[i];
    const uint64_t total_bytes_out = write_metadata(fd, blob_bytes, rows, [&](uint32_t fi) -> const std::string & { return rel[fi]; }, {sub});
    close(fd);
    if (trace_on())
        fprintf(stderr, "[host] compress_dir: walk %.1f ms  pack %.1f ms  metadata %.1f ms\n", (t_walk - t_begin) * 1e3, (t_pack - t_walk) * 1e3,
                (now_s() - t_pack) * 1e3);
    if (report) {
        *report = znippy_compression_report{(uint64_t)files.size(), cf, uf, total_dirs, cb + ub, total_bytes_out, cb, ub, (uint64_t)rows.size(),
                                            ub > 0 ? (float)cb / (float)std::max<uint64_t>(blob_bytes, 1) * 100.0f : 0.0f};
    }
    return ZNIPPY_OK;
}

// ---- read side --------------------------------------------------------------------------------
}  // extern "C"

namespace {

struct DecodedRange {
    std::vector<uint64_t> out_off;  // position of each row's bytes in the decoded region
    std::vector<uint64_t> len;      // bytes of each row: uncompressed_size, or blob_size for a stored row (decompress.rs:L160-166)
    uint64_t total = 0;
    std::vector<int32_t> status;
    znippy_verify_counters cnt{};
    std::vector<uint64_t> corrupt;  // absolute row numbers
};

struct ReadBufs {
    DevBuf d_blobs, d_out;
    PinBuf blob_pin;
    double t_read = 0, t_h2d = 0, t_kern = 0;
};

// Blob bytes -> HBM -> decode+verify kernels; the decoded rows stay in bufs.d_out (the caller decides whether
// they cross PCIe at all).  Stands for the worker loop body, decompress.rs:L135-190.
int decode_range(znippy_ctx *ctx, int arc_fd, const znippy_index &ix, const uint64_t *row_ids, size_t n, bool verify, ReadBufs &bufs,
                 DecodedRange *dr) {
    std::vector<uint64_t> bo(n), bs(n), us(n);
    std::vector<uint8_t> bitmap((n + 7) / 8, 0), ck(verify ? 32 * n : 0);
    dr->out_off.assign(n, 0);
    dr->len.assign(n, 0);
    dr->status.assign(n, 0);
    dr->total = 0;
    dr->cnt = znippy_verify_counters{};
    uint64_t lo = UINT64_MAX, hi = 0, sum = 0;
    for (size_t k = 0; k < n; k++) {
        const uint64_t r = row_ids[k];
        bo[k] = ix.rows.cols[5].u64[r]; bs[k] = ix.rows.cols[6].u64[r];
        us[k] = ix.rows.cols[3].u8[r] ? ix.rows.cols[4].u64[r] : bs[k];  // a stored row is its blob
        dr->len[k] = us[k];
        if (dr->total + us[k] < dr->total) return fail(ZNIPPY_E_CORRUPT, "row sizes overflow");
        if (ix.rows.cols[3].u8[r]) bitmap[k >> 3] |= (uint8_t)(1u << (k & 7));
        if (verify) std::memcpy(&ck[32 * k], &ix.rows.cols[7].u8[32 * r], 32);
        dr->out_off[k] = dr->total;
        dr->total += us[k];
        if (bo[k] > ix.file_size || bs[k] > ix.file_size - bo[k]) return fail(ZNIPPY_E_CORRUPT, "blob range outside the archive");
        lo = std::min(lo, bo[k]);
        hi = std::max(hi, bo[k] + bs[k]);
        sum += bs[k];
    }
    if (!n) return ZNIPPY_OK;
    double t = now_s();
    uint64_t base = lo, nblob = hi - lo;
    if (nblob > 2 * sum + (1u << 20)) {  // scattered rows: pack them instead of reading the span between them
        if (!bufs.blob_pin.reserve(sum + 64)) return fail(ZNIPPY_E_NOMEM, "page-locked allocation failed");
        uint64_t pos = 0;
        for (size_t k = 0; k < n; k++) {
            if (bs[k] && !pread_all(arc_fd, bufs.blob_pin.p + pos, bs[k], bo[k])) return fail(ZNIPPY_E_INVAL, "failed to read blob from archive");
            bo[k] = pos;
            pos += bs[k];
        }
        base = 0;
        nblob = sum;
    } else {
        if (!bufs.blob_pin.reserve(nblob + 64)) return fail(ZNIPPY_E_NOMEM, "page-locked allocation failed");
        if (nblob && !pread_all(arc_fd, bufs.blob_pin.p, nblob, lo)) return fail(ZNIPPY_E_INVAL, "failed to read blob from archive");
    }
    bufs.t_read += now_s() - t; t = now_s();
    if (!bufs.d_blobs.reserve(nblob + 64) || !bufs.d_out.reserve(dr->total + 64)) return fail(ZNIPPY_E_NOMEM, "device allocation failed");
    if (nblob && hipMemcpy(bufs.d_blobs.p, bufs.blob_pin.p, nblob, hipMemcpyHostToDevice) != hipSuccess) return fail(ZNIPPY_E_HIP, "H2D failed");
    bufs.t_h2d += now_s() - t; t = now_s();
    znippy_rows *rt = nullptr;
    int rc = znippy_rows_create(ctx, bo.data(), bs.data(), bitmap.data(), us.data(), dr->out_off.data(), verify ? ck.data() : nullptr, 0, n, &rt);
    if (rc) return fail(rc, "znippy_rows_create failed");
    znippy_rows_set_blob_cap(rt, nblob);  // a row pointing outside what was read is an error code, not a device fault
    std::vector<uint64_t> cr(n);
    rc = znippy_decode_verify_rows(ctx, rt, bufs.d_blobs.p, base, bufs.d_out.p, dr->total, &dr->cnt, cr.data(), n, dr->status.data());
    znippy_rows_destroy(rt);
    if (rc) return fail(rc, std::string("decode failed: ") + znippy_last_error(ctx));
    bufs.t_kern += now_s() - t;
    for (uint64_t k = 0; k < dr->cnt.corrupt_rows && k < n; k++) dr->corrupt.push_back(row_ids[cr[k]]);
    return ZNIPPY_OK;
}

// Positioned writes of decoded rows [k0,k1) of one range (decompress.rs:L186-189).  A file is created (and,
// when this rank owns the whole archive, truncated) at its first row; rows of one file are adjacent, so one
// cached descriptor per writer replaces the reference's table of open files.
struct RowWriter {
    const znippy_index *ix;
    const char *out_dir;
    const std::vector<uint8_t> *first_touch;
    bool truncate;  // this process owns the whole archive: O_TRUNC on first touch, like the reference (L74-101)
    const std::unordered_map<std::string, uint64_t> *final_size;  // several ranks share the files: first touch sets the final size
    std::atomic<int> *err;
    void operator()(const uint8_t *bytes, const DecodedRange *dr, uint64_t row0, uint64_t k0, uint64_t k1) const {
        std::string cur_dir;
        const std::string *cur_path = nullptr;
        int fd = -1;
        for (uint64_t k = k0; k < k1; k++) {
            const uint64_t r = row0 + k;
            if (dr->status[k] < 0) {  // decode error: logged + skipped (L159-162)
                fprintf(stderr, "[decomp] row %llu error=%d\n", (unsigned long long)r, dr->status[k]);
                continue;
            }
            const std::string &p = ix->rows.cols[0].str[r];
            if (!cur_path || *cur_path != p) {
                if (fd >= 0) close(fd);
                const std::string full = std::string(out_dir) + "/" + p;
                const bool first = (*first_touch)[r] != 0;
                if (first) {
                    const std::string dir = full.substr(0, full.find_last_of('/'));
                    if (dir != cur_dir) { mkdirs(dir); cur_dir = dir; }
                }
                fd = open(full.c_str(), O_CREAT | O_WRONLY | ((first && truncate) ? O_TRUNC : 0), 0644);
                cur_path = &p;
                if (fd < 0) { err->store(1); g_open_failed(p); return; }
                // Ranks write disjoint parts of a file in any order, so none of them may O_TRUNC; instead every rank
                // sets the file to its final length when it first touches it (idempotent: every row lies inside
                // that length) — a longer file left from an earlier run loses its stale tail.
                if (first && !truncate && final_size) {
                    auto it = final_size->find(p);
                    if (it != final_size->end() && ftruncate(fd, (off_t)it->second) != 0) err->store(2);
                }
            }
            if (!pwrite_all(fd, bytes + dr->out_off[k], dr->len[k], ix->rows.cols[2].u64[r])) err->store(2);
        }
        if (fd >= 0) close(fd);
    }
    static void g_open_failed(const std::string &p) { fprintf(stderr, "[decomp] failed to open output file %s\n", p.c_str()); }
};

}  // namespace

extern "C" {

static int znippy_decompress_archive_impl(const char *index_path, int save_data, const char *out_dir, int device, uint32_t rank,
                              uint32_t world, znippy_verify_report *report, uint64_t *corrupt_rows, uint64_t corrupt_cap,
                              uint64_t *n_corrupt) {
    if (!index_path || !report || (save_data && !out_dir) || world == 0 || rank >= world) return fail(ZNIPPY_E_INVAL, "bad argument");
    const double t_begin = now_s();
    znippy_index ix;
    int rc = load_index(index_path, &ix);
    if (rc) return rc;
    const double t_index = now_s();
    const uint64_t n_rows = ix.n();
    const auto &paths = ix.rows.cols[0].str;
    // unique paths (L64-69) + the first row of every path; rows of a path that are not adjacent (duplicate entries)
    // force a single writer so that the first-touch truncate cannot race with a later row's write
    std::vector<uint8_t> first_touch(n_rows, 0);
    bool adjacent = true;
    size_t n_unique = 0;
    {
        std::unordered_set<std::string> uniq;
        uniq.reserve(n_rows * 2);
        for (uint64_t r = 0; r < n_rows; r++) {
            if (r && paths[r] == paths[r - 1]) continue;
            if (uniq.insert(paths[r]).second) first_touch[r] = 1;
            else adjacent = false;
        }
        n_unique = uniq.size();
    }
    const auto range = split_rows(ix.rows.cols[4].u64, rank, world);
    std::unordered_map<std::string, uint64_t> final_size;
    if (world > 1 && save_data) {
        // first touch is per rank (the first row of a path inside MY range), and comes with the file's final length
        std::fill(first_touch.begin(), first_touch.end(), 0);
        std::unordered_set<std::string> seen;
        for (uint64_t r = range.first; r < range.second; r++) {
            if (r > range.first && paths[r] == paths[r - 1]) continue;
            if (seen.insert(paths[r]).second) first_touch[r] = 1;
        }
        final_size.reserve(n_unique * 2);
        for (uint64_t r = 0; r < n_rows; r++) {
            const uint64_t len = ix.rows.cols[3].u8[r] ? ix.rows.cols[4].u64[r] : ix.rows.cols[6].u64[r];
            uint64_t &fs = final_size[paths[r]];
            fs = std::max(fs, ix.rows.cols[2].u64[r] + len);
        }
    }
    if (save_data) mkdirs(out_dir);
    int arc = open(index_path, O_RDONLY);
    if (arc < 0) return fail(ZNIPPY_E_INVAL, "cannot open archive");
    znippy_ctx *ctx = nullptr;
    if (range.second > range.first) {
        if (hipSetDevice(device) != hipSuccess || (rc = znippy_ctx_create(device, nullptr, &ctx))) {
            close(arc);
            return fail(rc ? rc : ZNIPPY_E_HIP, "no usable GPU: the codec/hash path has no CPU fallback");
        }
    }
    const double t_ctx = now_s();
    znippy_verify_counters tot{};
    std::vector<uint64_t> corrupt;
    ReadBufs bufs;
    // save_data: the decoded range crosses PCIe into one of two page-locked slabs and a few writer threads
    // scatter it into the output files while the GPU works on the next range.  verify-only: nothing is copied back.
    struct OutSlab {
        PinBuf pin;
        DecodedRange dr;
        std::vector<std::thread> writers;
        uint64_t last_row = 0;
        void join() { for (auto &t : writers) t.join(); writers.clear(); }
    } slabs[2];
    std::atomic<int> werr{0};
    const RowWriter writer{&ix, out_dir, &first_touch, world == 1, world > 1 ? &final_size : nullptr, &werr};
    const unsigned n_writers = adjacent ? writer_threads() : 1;
    const uint64_t batch_bytes = range_bytes(save_data != 0);
    double t_d2h = 0, t_wjoin = 0;
    int which = 0, n_ranges = 0;
    uint64_t i = range.first;
    std::vector<uint64_t> ids;
    while (i < range.second && rc == ZNIPPY_OK && !werr.load()) {
        uint64_t j = i, nbytes = 0;
        while (j < range.second && (j == i || nbytes + ix.rows.cols[4].u64[j] <= batch_bytes)) nbytes += ix.rows.cols[4].u64[j++];
        ids.resize(j - i);
        for (uint64_t k = i; k < j; k++) ids[k - i] = k;
        OutSlab &sl = slabs[which];
        double t = now_s();
        sl.join();  // the slab's previous range has been written out
        // a file continuing from the range still being written must see its first-touch create/truncate first
        OutSlab &other = slabs[which ^ 1];
        if (!other.writers.empty() && i && paths[i] == paths[i - 1]) other.join();
        t_wjoin += now_s() - t;
        sl.dr.corrupt.clear();
        rc = decode_range(ctx, arc, ix, ids.data(), ids.size(), true, bufs, &sl.dr);
        if (rc) break;
        n_ranges++;
        const znippy_verify_counters &c = sl.dr.cnt;
        tot.total_chunks += c.total_chunks; tot.total_written_bytes += c.total_written_bytes;
        tot.verified_bytes += c.verified_bytes; tot.corrupt_bytes += c.corrupt_bytes;
        tot.corrupt_rows += c.corrupt_rows; tot.decode_errors += c.decode_errors;
        corrupt.insert(corrupt.end(), sl.dr.corrupt.begin(), sl.dr.corrupt.end());
        if (save_data) {
            t = now_s();
            if (!sl.pin.reserve(sl.dr.total + 64)) { rc = fail(ZNIPPY_E_NOMEM, "page-locked allocation failed"); break; }
            if (sl.dr.total && hipMemcpy(sl.pin.p, bufs.d_out.p, sl.dr.total, hipMemcpyDeviceToHost) != hipSuccess) { rc = fail(ZNIPPY_E_HIP, "D2H failed"); break; }
            t_d2h += now_s() - t;
            // split [i,j) into contiguous parts of about equal bytes, cut only where the path changes
            const uint64_t n = j - i;
            uint64_t k0 = 0;
            for (unsigned w = 0; w < n_writers && k0 < n; w++) {
                uint64_t k1 = n;
                if (w + 1 < n_writers) {
                    const uint64_t target = sl.dr.total / n_writers * (w + 1);
                    k1 = (uint64_t)(std::lower_bound(sl.dr.out_off.begin() + k0, sl.dr.out_off.end(), target) - sl.dr.out_off.begin());
                    k1 = std::max(k1, k0 + 1);
                    while (k1 < n && paths[i + k1] == paths[i + k1 - 1]) k1++;
                }
                sl.writers.emplace_back(writer, sl.pin.p, &sl.dr, i, k0, k1);
                k0 = k1;
            }
            which ^= 1;
        }
        i = j;
    }
    double t = now_s();
    slabs[0].join();
    slabs[1].join();
    t_wjoin += now_s() - t;
    close(arc);
    if (ctx) znippy_ctx_destroy(ctx);
    if (rc) return rc;
    if (werr.load()) return fail(ZNIPPY_E_INVAL, werr.load() == 1 ? "failed to open output file" : "failed to write output file");
    if (trace_on())
        fprintf(stderr, "[host] decompress_archive: index %.1f ms  ctx %.1f ms  ranges %d  pread %.1f ms  h2d %.1f ms  kernels %.1f ms  d2h %.1f ms  "
                        "writer-wait %.1f ms  total %.1f ms\n",
                (t_index - t_begin) * 1e3, (t_ctx - t_index) * 1e3, n_ranges, bufs.t_read * 1e3, bufs.t_h2d * 1e3, bufs.t_kern * 1e3, t_d2h * 1e3,
                t_wjoin * 1e3, (now_s() - t_begin) * 1e3);
    std::sort(corrupt.begin(), corrupt.end());
    for (uint64_t r : corrupt) fprintf(stderr, "[verify] MISMATCH row=%llu\n", (unsigned long long)r);
    const uint64_t corrupt_files = corrupt.size();  // number of corrupt ROWS (L210)
    report->total_files = n_unique;
    report->corrupt_files = corrupt_files;
    report->verified_files = n_unique > corrupt_files ? n_unique - corrupt_files : 0;
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
    ReadBufs bufs;
};

extern "C" {

static int znippy_archive_open_impl(const char *path, int device, znippy_archive **out) {
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

static int znippy_archive_extract_file_impl(znippy_archive *a, const char *rel, void *dst, size_t cap, size_t *written, int verify = 0) {
    if (!a || !rel || !written) return fail(ZNIPPY_E_INVAL, "null argument");
    auto it = a->files.find(rel);
    if (it == a->files.end()) return fail(ZNIPPY_E_INVAL, std::string("file not found in archive: ") + rel);
    if (!a->ctx) {
        if (hipSetDevice(a->device) != hipSuccess) return fail(ZNIPPY_E_HIP, "hipSetDevice failed");
        int rc = znippy_ctx_create(a->device, nullptr, &a->ctx);
        if (rc) return fail(rc, "no usable GPU: the codec/hash path has no CPU fallback");
    }
    DecodedRange dr;
    int rc = decode_range(a->ctx, a->fd, a->ix, it->second.data(), it->second.size(), verify != 0, a->bufs, &dr);
    if (rc) return rc;
    for (int32_t st : dr.status)
        if (st < 0) return fail(st, "OpenZL-equivalent decompress failed");  // propagates (archive.rs:L160)
    // optional verify (the reference's extract_file has none, archive.rs:L144-168; SURVEY §8f rank 1 adds it): every
    // chunk's BLAKE3 against the index's checksum column, computed by the same kernels as the bulk path
    if (verify && dr.cnt.corrupt_rows) return fail(ZNIPPY_E_CHECKSUM, std::string("checksum mismatch in ") + rel);
    if (dr.total > cap) return fail(ZNIPPY_E_DST_SMALL, "destination too small");
    if (dr.total && hipMemcpy(dst, a->bufs.d_out.p, dr.total, hipMemcpyDeviceToHost) != hipSuccess) return fail(ZNIPPY_E_HIP, "D2H failed");
    *written = dr.total;
    return ZNIPPY_OK;
}

void znippy_archive_close(znippy_archive *a) {
    if (!a) return;
    if (a->ctx) znippy_ctx_destroy(a->ctx);
    if (a->fd >= 0) close(a->fd);
    delete a;
}

// ---- index ------------------------------------------------------------------------------------
static int znippy_index_open_impl(const char *path, znippy_index **out) {
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

// ---- the entry points above, behind the exception guard ----
extern "C" {

int znippy_archive_extract_file_verified(znippy_archive *a, const char *relative_path, void *dst, size_t cap, size_t *written) {
    return guarded([&] { return znippy_archive_extract_file_impl(a, relative_path, dst, cap, written, 1); });
}

int znippy_compress_stream(const char *output, int no_skip, int device, znippy_stream **out) {
    return guarded([&] { return znippy_compress_stream_impl(output, no_skip, device, out); });
}

int znippy_stream_send(znippy_stream *s, const char *relative_path, const void *data, size_t len, int pkg_type, const char *repo) {
    return guarded([&] { return znippy_stream_send_impl(s, relative_path, data, len, pkg_type, repo); });
}

int znippy_stream_send_packed(znippy_stream *s, uint64_t n, const char *paths, const uint64_t *path_off, const void *data,
                              const uint64_t *data_off, const int32_t *pkg_type, const char *repo) {
    return guarded([&] { return znippy_stream_send_packed_impl(s, n, paths, path_off, data, data_off, pkg_type, repo); });
}

int znippy_stream_finish(znippy_stream *sp, znippy_compression_report *report) {
    return guarded([&] { return znippy_stream_finish_impl(sp, report); });
}

int znippy_compress_dir(const char *input_dir, const char *output, int no_skip, const char *repo, int device, znippy_compression_report *report) {
    return guarded([&] { return znippy_compress_dir_impl(input_dir, output, no_skip, repo, device, report); });
}

int znippy_decompress_archive(const char *index_path, int save_data, const char *out_dir, int device, uint32_t rank, uint32_t world, znippy_verify_report *report, uint64_t *corrupt_rows, uint64_t corrupt_cap, uint64_t *n_corrupt) {
    return guarded([&] { return znippy_decompress_archive_impl(index_path, save_data, out_dir, device, rank, world, report, corrupt_rows, corrupt_cap, n_corrupt); });
}

int znippy_archive_open(const char *path, int device, znippy_archive **out) {
    return guarded([&] { return znippy_archive_open_impl(path, device, out); });
}

int znippy_archive_extract_file(znippy_archive *a, const char *rel, void *dst, size_t cap, size_t *written) {
    return guarded([&] { return znippy_archive_extract_file_impl(a, rel, dst, cap, written, 0); });
}

int znippy_index_open(const char *path, znippy_index **out) {
    return guarded([&] { return znippy_index_open_impl(path, out); });
}

}  // extern "C"
