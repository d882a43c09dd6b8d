// C ABI of libznippy_hip.so (include/znippy_hip.h).  Host-side plumbing only: uploads of the
// index columns / Round tables, the tile plan that drives the kernels' work cursor, kernel
// launches on the context's HIP stream, and result read-back.
#include <unistd.h>
#include "common.h"
#include "encode.h"
#include "../../include/znippy_hip.h"

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <chrono>
#include <string>
#include <vector>
#include <unordered_map>

namespace zn {
int measure_b3_pass_ns(int cus, hipStream_t s, float *ns_per_pass_per_simd, float *ghz);
size_t decode_lit_scratch_bytes(int grid);
void set_fused_dbg(unsigned long long *p);
void set_fused_abl(int v);
}  // namespace zn

using namespace zn;
extern "C" void zn_rounds_totals(const uint64_t *len, const uint8_t *skip, size_t n, uint64_t out[8]);  // host/extents.cpp
extern "C" void zn_rows_extents(const uint64_t *bo, const uint64_t *bs, const uint64_t *oo, const uint64_t *us, size_t n, uint64_t out[8]);  // host/extents.cpp
extern "C" int zn_rows_pack32(const uint64_t *bo, const uint64_t *bs, const uint64_t *oo, const uint64_t *us, size_t n, uint32_t *dst);

#define HIPCHK(ctx, call)                                                                         \
    do {                                                                                          \
        hipError_t e_ = (call);                                                                   \
        if (e_ != hipSuccess) {                                                                   \
            (ctx)->err = std::string(#call) + ": " + hipGetErrorString(e_);                       \
            return ZNIPPY_E_HIP;                                                                  \
        }                                                                                         \
    } while (0)

// diagnostic (ZNIPPY_TDBG): wall time of the sections of a table constructor
struct TDbg {
    bool on;
    const char *what;
    std::chrono::steady_clock::time_point t;
    std::string line;
    TDbg(bool on_, const char *what_) : on(on_), what(what_), t(std::chrono::steady_clock::now()) {}
    void mark(const char *name) {
        if (!on) return;
        const auto now = std::chrono::steady_clock::now();
        char b[96];
        snprintf(b, sizeof b, " %s=%.3f", name, std::chrono::duration<double, std::milli>(now - t).count());
        line += b;
        t = now;
    }
    ~TDbg() { if (on) fprintf(stderr, "[znippy tdbg] %s:%s ms\n", what, line.c_str()); }
};

struct KTime {
    const char *name;
    hipEvent_t t0, t1;
};

struct znippy_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    bool own_stream = false;
    std::string err;
    // decode scratch
    int decode_grid = 0;
    uint8_t *lit_scratch = nullptr;
    uint32_t *cursor = nullptr;
    // shim scratch (grow-only)
    uint8_t *shim_in = nullptr, *shim_out = nullptr;
    size_t shim_in_cap = 0, shim_out_cap = 0;
    // encoder scratch (grow-only) + tables
    int encode_grid = 0, encode_grid_small = 0;
    int gen_share = 3;  // workgroups per CU the general decoder takes while block items / foreign frames run beside it (4 = the whole register file)
    int level = 19;  // CompressCtx::new(compression_level), codec.rs:L16-28; CONFIG.compression_level is 19 (common_config.rs:L37)
    uint8_t *enc_prov = nullptr;
    size_t enc_prov_cap = 0;
    uint32_t *enc_seq = nullptr;
    EncTables *enc_tabs = nullptr;
    // auxiliary stream: the write side hashes on it while the main stream encodes
    hipStream_t aux = nullptr;
    hipStream_t copy = nullptr;  // result read-back of the write side (D2H beside the next run's kernels)
    uint8_t *lit_scratch_b = nullptr;  // literal scratch of the block-item launch (runs next to the general decoder)
    // pools of the two-phase path for foreign frames: decoded literals and 8-byte sequence records of every block of
    // every candidate frame of a run (grow-only, sized by the largest table seen; see ensure_fz_pools)
    uint8_t *fz_lit_pool = nullptr;
    unsigned long long *fz_seq_pool = nullptr;
    uint64_t fz_lit_cap = 0, fz_seq_cap = 0;
    // the batch path's table pools (zstd_batch.hip, k_bx_*): FSE decoding tables as 4-byte cells (the predefined ones at
    // the head), Huffman decoding tables as 2-byte cells
    uint16_t *bx_fse_pool = nullptr;
    uint16_t *bx_huf_pool = nullptr;
    uint64_t bx_fse_cap = 0, bx_huf_cap = 0;
    // the resolve path's word pool (k_rx_*: one 32-bit word per output byte of the big frames) and its chunk -> frame table
    uint32_t *rx_pool = nullptr, *rx_chunk = nullptr;
    uint8_t *rx_cdone = nullptr;
    uint64_t rx_cap = 0;  // words
    hipEvent_t ev_fork = nullptr, ev_join = nullptr, ev_join2 = nullptr;
    // kernel timing
    std::vector<KTime> ktimes;
    int n_ktimes = 0;
    std::vector<hipEvent_t> event_pool;  // disable-timing events handed back by destroyed tables
    bool ktime_open = false;
    // page-locked host buffers handed back by destroyed tables: locking pages costs ~1 ms per 4 MB, more than a
    // whole C2 encode pass, so a table takes its result mirror from here when one is big enough
    std::vector<std::pair<size_t, void *>> pinned_pool;
    size_t pinned_pool_bytes = 0;
    std::vector<std::pair<size_t, void *>> dev_pool;  // device buffers handed back by destroyed tables
    std::unordered_map<void *, size_t> dev_sizes;     // capacity of every pooled-kind buffer that is in use
    size_t dev_pool_bytes = 0;
    // diagnostic switches (ZNIPPY_* environment), read ONCE when the context is created: nothing on the hot path
    // calls getenv
    struct {
        int dbg = 0;             // ZNIPPY_DBG bit set (FusedArgs::dbg)
        unsigned lds_pad = 0;    // ZNIPPY_LDS_PAD
        int store_g = 0;         // ZNIPPY_STORE_G: tiles per wave of the store path kernel (0 = from the tile count)
        bool no_block_items = false, no_fused_blocks = false, ddbg = false, edbg = false, no_fused_store = false,
             nohash = false, no_roles = false, no_fz = false, fz_only = false, no_bx = false, tdbg = false, no_fuse_hash = false, trace = false, no_rx = false, no_pack = false, no_lean = false, no_stored_only = false;
        // ZNIPPY_ROLES_MIN: small tiles from which the role-split persistent kernel takes the table (0 = never; below
        // a few CU-fillings a persistent grid only adds start-up latency)
        unsigned roles_min = 2048;
        unsigned bx_big = BX_BIG_SEQ;  // ZNIPPY_BX_BIG
        bool bx_big_set = false;
        int ktime = 2;  // per-kernel HIP events: 2 = every kernel, 1 = the dominant read kernels only, 0 = none
    } sw;
    int cus = 256;
    // Lifetime (znippy_hip.h): tables hold a reference to their context.  znippy_ctx_destroy with tables still alive only
    // closes the context (every call on it fails with ZNIPPY_E_INVAL from then on); its memory and device resources go
    // when the last table is destroyed.
    int live_tables = 0;
    bool closing = false;
    unsigned long long *clk_buf = nullptr;  // diagnostic (ZNIPPY_DBG & 32768): shader cycles / 100 MHz ticks of one wave
};

static void read_switches(znippy_ctx *ctx) {
    auto on = [](const char *n) { const char *v = getenv(n); return v && *v && *v != '0'; };
    if (const char *e = getenv("ZNIPPY_DBG")) ctx->sw.dbg = atoi(e);
    if (const char *e = getenv("ZNIPPY_LDS_PAD")) ctx->sw.lds_pad = (unsigned)atoi(e);
    ctx->sw.no_block_items = on("ZNIPPY_NO_BLOCK_ITEMS");
    ctx->sw.no_fused_blocks = on("ZNIPPY_NO_FUSED_BLOCKS");
    ctx->sw.ddbg = on("ZNIPPY_DDBG");
    ctx->sw.edbg = on("ZNIPPY_EDBG");
    ctx->sw.no_fused_store = on("ZNIPPY_NO_FUSED_STORE");
    ctx->sw.nohash = on("ZNIPPY_NOHASH");
    ctx->sw.no_roles = on("ZNIPPY_NO_ROLES");
    ctx->sw.no_fz = on("ZNIPPY_NO_FZ");
    ctx->sw.tdbg = on("ZNIPPY_TDBG");
    ctx->sw.trace = on("ZNIPPY_TRACE");
    ctx->sw.no_lean = on("ZNIPPY_NO_LEAN");  // A/B: every run launches the kernels behind the role-split one
    ctx->sw.no_pack = on("ZNIPPY_NO_PACK");  // A/B: the index columns always as four 64-bit copies
    if (const char *e = getenv("ZNIPPY_STORE_G")) ctx->sw.store_g = atoi(e) == 1 ? 1 : (atoi(e) == 2 ? 2 : 0);  // A/B, tests: tiles per wave of the store path kernel
    ctx->sw.no_stored_only = on("ZNIPPY_NO_STORED_ONLY");  // A/B: tables without a compressed row through the fused small-row kernels (until round 3's last day)
    ctx->sw.no_rx = on("ZNIPPY_NO_RX");  // A/B: big foreign frames executed by a wave each (round 3's first form)
    ctx->sw.no_fuse_hash = on("ZNIPPY_NO_FUSE_HASH");  // A/B: the write side's hash as a kernel of its own beside the encoder (round 2)
    ctx->sw.no_bx = on("ZNIPPY_NO_BX");
    if (const char *e = getenv("ZNIPPY_BX_BIG")) { ctx->sw.bx_big = (unsigned)atoi(e); ctx->sw.bx_big_set = true; }  // A/B: foreign frames through the round-2 paths (serial decoder + wave-per-block two-phase path)
    if (const char *lv = getenv("ZNIPPY_LEVEL")) { const int v = atoi(lv); if (v >= 1 && v <= 22) ctx->level = v; }  // initial level of every context (tests, A/B runs)
    if (const char *gs = getenv("ZNIPPY_GEN_SHARE")) { const int v = atoi(gs); if (v >= 1 && v <= 4) ctx->gen_share = v; }  // A/B
    ctx->sw.fz_only = on("ZNIPPY_FZ_ONLY");  // test hook: no serial fallback behind the two-phase path (what it leaves shows up as corrupt rows)
    if (const char *e = getenv("ZNIPPY_ROLES_MIN")) ctx->sw.roles_min = (unsigned)atoi(e);
    if (const char *e = getenv("ZNIPPY_KTIME")) ctx->sw.ktime = atoi(e);
}

static void *pinned_take(znippy_ctx *ctx, size_t bytes, size_t *cap) {
    size_t best = ctx->pinned_pool.size();
    for (size_t i = 0; i < ctx->pinned_pool.size(); i++)
        if (ctx->pinned_pool[i].first >= bytes && ctx->pinned_pool[i].first <= 4 * bytes + 4096 &&
            (best == ctx->pinned_pool.size() || ctx->pinned_pool[i].first < ctx->pinned_pool[best].first))
            best = i;
    if (best != ctx->pinned_pool.size()) {
        void *p = ctx->pinned_pool[best].second;
        *cap = ctx->pinned_pool[best].first;
        ctx->pinned_pool_bytes -= *cap;
        ctx->pinned_pool.erase(ctx->pinned_pool.begin() + best);
        return p;
    }
    void *p = nullptr;
    if (hipHostMalloc(&p, bytes) != hipSuccess) return nullptr;
    *cap = bytes;
    return p;
}
static void pinned_give(znippy_ctx *ctx, void *p, size_t cap) {
    if (!p) return;
    if (ctx->pinned_pool.size() < 16 && ctx->pinned_pool_bytes + cap <= (256ull << 20)) {
        ctx->pinned_pool.emplace_back(cap, p);
        ctx->pinned_pool_bytes += cap;
    } else (void)hipHostFree(p);
}
// Events of tables (two or four each) come from a per-context free list: creating one costs ~0.1 ms.
static hipEvent_t event_take(znippy_ctx *ctx) {
    if (!ctx->event_pool.empty()) {
        hipEvent_t e = ctx->event_pool.back();
        ctx->event_pool.pop_back();
        return e;
    }
    hipEvent_t e = nullptr;
    if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) return nullptr;
    return e;
}
static void event_give(znippy_ctx *ctx, hipEvent_t e) {
    if (!e) return;
    if (ctx->event_pool.size() < 64) ctx->event_pool.push_back(e);
    else (void)hipEventDestroy(e);
}

// Device memory of the tables (index columns, plans, per-run scratch) comes from a per-context pool: a table is
// built and torn down per hand-off in the host pipelines, and ~15 hipMalloc + hipFree pairs cost more than the
// kernels of a small hand-off.  Nothing in a table relies on fresh memory: every array is either uploaded, reset by
// the run (memset) or written before it is read.
template <class T>
static hipError_t tmalloc(znippy_ctx *ctx, T **out, size_t bytes) {
    bytes = std::max<size_t>(bytes, 16);
    size_t best = ctx->dev_pool.size();
    for (size_t i = 0; i < ctx->dev_pool.size(); i++)
        if (ctx->dev_pool[i].first >= bytes && ctx->dev_pool[i].first <= 2 * bytes + 4096 &&
            (best == ctx->dev_pool.size() || ctx->dev_pool[i].first < ctx->dev_pool[best].first))
            best = i;
    if (best != ctx->dev_pool.size()) {
        *out = (T *)ctx->dev_pool[best].second;
        ctx->dev_sizes[*out] = ctx->dev_pool[best].first;
        ctx->dev_pool_bytes -= ctx->dev_pool[best].first;
        ctx->dev_pool.erase(ctx->dev_pool.begin() + best);
        return hipSuccess;
    }
    void *p = nullptr;
    const hipError_t e = hipMalloc(&p, bytes);
    if (e != hipSuccess) return e;
    *out = (T *)p;
    ctx->dev_sizes[p] = bytes;
    return hipSuccess;
}
static void tfree(znippy_ctx *ctx, void *p) {
    if (!p) return;
    auto it = ctx->dev_sizes.find(p);
    if (it == ctx->dev_sizes.end()) { (void)hipFree(p); return; }
    const size_t cap = it->second;
    ctx->dev_sizes.erase(it);
    if (ctx->dev_pool.size() < 96 && ctx->dev_pool_bytes + cap <= (1ull << 30)) {
        ctx->dev_pool.emplace_back(cap, p);
        ctx->dev_pool_bytes += cap;
    } else (void)hipFree(p);
}

struct PlanBuf {
    std::vector<Tile> tiles;
    std::vector<BigUnit> big;
    uint32_t n_tile_cv = 0;
    // units with more than 64 tile CVs: their CVs are first folded in groups of 64 by independent waves
    // (group g of big unit b: grp_big[g] = b, first CV = big[b].cv_base + 64 * grp_k[g]); results live behind the
    // tile CVs, from BigUnit::pad on
    std::vector<uint32_t> grp_big, grp_k;
};

// Greedy tile plan over unit lengths: whole small units are packed until a wave's 64 lanes are
// full; a unit with more than 64 leaves becomes ceil(leaves/64) slice tiles + one BigUnit.
template <class LenOf>
static void build_plan(LenOf len_of, uint32_t n, PlanBuf &p) {
    p.tiles.reserve((size_t)n / 4 + 16);
    uint32_t cur_first = 0, cur_units = 0, cur_leaves = 0;
    auto flush = [&]() {
        if (cur_units) p.tiles.push_back(Tile{cur_first, cur_units, 0, cur_leaves, 0, 0});
        cur_units = 0;
        cur_leaves = 0;
    };
    for (uint32_t u = 0; u < n; u++) {
        const uint64_t len_u = len_of(u);
        uint64_t leaves64 = len_u ? (len_u + 1023) >> 10 : 1;
        if (leaves64 > 64) {
            flush();
            uint32_t n_cvs = (uint32_t)((leaves64 + 63) / 64);
            p.big.push_back(BigUnit{u, p.n_tile_cv, n_cvs, 0});
            if (n_cvs > 64)
                for (uint32_t g = 0; g < (n_cvs + 63) / 64; g++) { p.grp_big.push_back((uint32_t)p.big.size() - 1); p.grp_k.push_back(g); }
            for (uint32_t t = 0; t < n_cvs; t++) {
                uint32_t first = t * 64;
                uint32_t nl = (uint32_t)std::min<uint64_t>(64, leaves64 - first);
                p.tiles.push_back(Tile{u, 0, first, nl, p.n_tile_cv + t, 0});
            }
            p.n_tile_cv += n_cvs;
        } else {
            uint32_t leaves = (uint32_t)leaves64;
            if (cur_units == 64 || cur_leaves + leaves > 64) flush();
            if (!cur_units) cur_first = u;
            cur_units++;
            cur_leaves += leaves;
            // a run of units of this same size (archives of fixed-size chunks: every BASELINE config): whole tiles at once —
            // exactly the tiles the unit-by-unit rule above would cut
            if (cur_units == 1 && u + 1 < n && len_of(u + 1) == len_u) {
                uint32_t v = u + 1;
                while (v < n && len_of(v) == len_u) v++;
                const uint32_t per = std::min<uint32_t>(64 / leaves, 64), run = v - u, full = run / per;
                if (full >= 1) {
                    for (uint32_t t = 0; t < full; t++) p.tiles.push_back(Tile{u + t * per, per, 0, per * leaves, 0, 0});
                    cur_units = 0; cur_leaves = 0;
                    const uint32_t rest = run - full * per;
                    if (rest) { cur_first = u + full * per; cur_units = rest; cur_leaves = rest * leaves; }
                    u = v - 1;
                }
            }
        }
    }
    flush();
    uint32_t gb = 0;  // where each grouped unit's group CVs start (behind all tile CVs)
    for (BigUnit &b : p.big)
        if (b.n_cvs > 64) { b.pad = p.n_tile_cv + gb; gb += (b.n_cvs + 63) / 64; }
}

struct DevPlan {
    Tile *tiles = nullptr;
    BigUnit *big = nullptr;
    uint32_t *tile_cv = nullptr;
    uint32_t *grp_big = nullptr, *grp_k = nullptr;
    uint32_t n_grp = 0;
    uint32_t n_tiles = 0, n_big = 0;
    uint32_t max_cvs = 0;  // tile CVs of the biggest unit (picks the merge launch)
};

static int upload_plan(znippy_ctx *ctx, const PlanBuf &p, DevPlan &d) {
    d.n_tiles = (uint32_t)p.tiles.size();
    d.n_big = (uint32_t)p.big.size();
    d.max_cvs = 0;
    for (const BigUnit &b : p.big) d.max_cvs = std::max(d.max_cvs, b.n_cvs);
    if (d.n_tiles) {
        HIPCHK(ctx, tmalloc(ctx, &d.tiles, sizeof(Tile) * d.n_tiles));
        HIPCHK(ctx, hipMemcpy(d.tiles, p.tiles.data(), sizeof(Tile) * d.n_tiles, hipMemcpyHostToDevice));
    }
    if (d.n_big) {
        HIPCHK(ctx, tmalloc(ctx, &d.big, sizeof(BigUnit) * d.n_big));
        HIPCHK(ctx, hipMemcpy(d.big, p.big.data(), sizeof(BigUnit) * d.n_big, hipMemcpyHostToDevice));
        HIPCHK(ctx, tmalloc(ctx, &d.tile_cv, 32 * ((size_t)p.n_tile_cv + p.grp_big.size())));
        d.n_grp = (uint32_t)p.grp_big.size();
        if (d.n_grp) {
            HIPCHK(ctx, tmalloc(ctx, &d.grp_big, 4 * (size_t)d.n_grp));
            HIPCHK(ctx, tmalloc(ctx, &d.grp_k, 4 * (size_t)d.n_grp));
            HIPCHK(ctx, hipMemcpy(d.grp_big, p.grp_big.data(), 4 * (size_t)d.n_grp, hipMemcpyHostToDevice));
            HIPCHK(ctx, hipMemcpy(d.grp_k, p.grp_k.data(), 4 * (size_t)d.n_grp, hipMemcpyHostToDevice));
        }
    }
    return ZNIPPY_OK;
}

static void free_plan(znippy_ctx *ctx, DevPlan &d) {
    tfree(ctx, d.tiles);
    tfree(ctx, d.big);
    tfree(ctx, d.tile_cv);
    tfree(ctx, d.grp_big);
    tfree(ctx, d.grp_k);
    d = DevPlan();
}

struct znippy_rows {
    znippy_ctx *ctx = nullptr;
    uint64_t row_begin = 0;
    uint32_t n = 0;
    uint32_t n_compressed = 0;
    uint64_t *blob_off = nullptr, *blob_size = nullptr, *usize = nullptr, *out_off = nullptr;
    uint8_t *compressed = nullptr, *checksum = nullptr, *d_bitmap = nullptr;
    uint32_t *h_pack = nullptr, *d_pack = nullptr;  // front-to-back tables: the two size columns as 32-bit values (page-locked / device)
    size_t h_pack_cap = 0;
    unsigned long long *d_pack_sums = nullptr;
    // One allocation, cleared (or preset) by ONE stream operation per run: [counters 8 x u64][hand-over counts 16 x u32]
    // [work cursors 16 x u32][pad 64 B][status n x i32]
    uint8_t *ctl = nullptr;
    static constexpr size_t CTL_HEAD = 512;  // [256, 384): pool counters of the batch path (16 x u64), [384, 448): its work counters (16 x u32)
    size_t ctl_bytes = 0;
    int32_t *status = nullptr;
    uint32_t *digests = nullptr;
    uint64_t *counters = nullptr;
    uint32_t *cursor = nullptr;
    // pinned mirror of the counters, filled by the run's own D2H copy.  Two slots + one event each: run k uses slot
    // k & 1, so the counters of run k can be read while run k + 1 is already executing (znippy_rows_results_lagged)
    uint64_t *h_counters = nullptr;  // per slot 16 x u64: the counters, then the 16 hand-over counts (u32)
    size_t h_counters_cap = 0;
    // Does the batch path have anything to do?  Its six launches cost ~0.1 ms even when every list is empty (the BASELINE
    // archives: every row is taken by the fused kernels), so a table remembers what its last finished run found: -1 not
    // known yet (launch it), 0 nothing handed over (the serial decoder alone stands behind the fused kernels, as a
    // safety net), 1 something was.
    int bx_hint = -1;
    uint64_t hint_seq = 0;  // runs whose mirror has been looked at
    // Lean runs.  A table of small rows whose last finished run left nothing behind the role-split kernel — no tile on its
    // list, no row handed over — is run as memset + that kernel + verify: the three launches behind it (left-over tiles,
    // serial decoder, second hash pass) cost ~25 us of a 0.5 ms step for looking at empty lists.  The verify kernel checks
    // the lists; if this run did leave something (the blobs changed), its counters come back flagged and whoever reads the
    // run's results first runs it again in full (rows_settle) — same inputs, the results the caller would have had.
    bool lean_ok = false;      // the table's shape allows it (set at creation)
    int lean_hint = -1;        // last finished run: 1 nothing left behind the roles kernel, 0 something was, -1 not known
    bool lean_blocks_ok = false;  // the same for tables of big multi-block rows only (the fused block kernel in front)
    bool lean_mixed_ok = false;   // small rows beside big stored / hashed units: the small rows' kernel beside the second hash pass
    bool roles_off = false;       // a run of the role-split kernel left every tile on its list (rows of no shape it takes: 0.15 ms of looking)
    bool small_ok = false, small_off = false;  // the fused kernels handed over every row: later runs skip them (all_rows = the batch path's list)
    uint32_t *all_rows = nullptr;
    int lean_hint2 = -1;
    bool last_lean = false;
    struct RunArgs { const void *blobs = nullptr; void *out = nullptr; uint64_t base = 0, cap = 0; } run_args[2];  // per mirror slot: what the run was given
    hipEvent_t ev_done[2] = {nullptr, nullptr};
    uint64_t run_seq = 0;  // async runs queued so far
    // Host copies of the columns a run is validated against (one pass per distinct (blob_base, blob_cap, out_cap)):
    // a row whose blob lies outside the blob region, or whose bytes would land outside the output region, gets its
    // status from the host (status_init) and no kernel touches it — a crafted index is an error code, not a fault.
    // extents of the table's rows (one pass at creation): a run whose regions contain them has no bad row; otherwise the
    // columns come back from the device for the per-row pass (h_*: filled then)
    uint64_t ext_min_bo = 0, ext_max_bend = 0, ext_max_oend = 0;
    bool ext_wrap = false;
    std::vector<uint64_t> h_blob_off, h_blob_size, h_len, h_out_off;
    uint64_t blob_cap = ~0ull;  // size of the caller's blob region (znippy_rows_set_blob_cap); ~0 = not declared
    uint64_t val_base = 0, val_bcap = 0, val_ocap = 0;
    bool val_done = false;
    uint32_t n_bad = 0;
    bool force_full = false;  // a lean run came back flagged: this table runs in full from now on
    uint8_t *status_init = nullptr;  // image of ctl with the host-decided statuses (rows_validate)
    bool odd_out = false;  // some stored row's output offset is not a multiple of 16 (store-path kernel variant)
    uint64_t *corrupt = nullptr;
    uint32_t corrupt_cap = 0;
    uint32_t *list_a = nullptr;   // compressed rows with > 64 leaves: general decoder
    uint32_t n_list_a = 0;
    bool wide_rows = false;       // big rows average >= 1 MiB: 1024-thread workgroups
    uint32_t *pending = nullptr;  // rows the fused kernel hands over (+ its counter)
    uint32_t *pending_count = nullptr;
    // block items: compressed rows of >= 2 blocks are tried block by block first
    uint32_t n_cand = 0, n_items = 0;
    uint32_t *cand_row = nullptr, *cand_base = nullptr, *cand_nblocks = nullptr, *pending2 = nullptr;
    // two-phase path for the candidates the block-item path gives up on (foreign frames): item slots per candidate
    // (2 x the expected blocks + 8: a writer may split blocks), the work list of the run, pool demand
    uint32_t *fz_base = nullptr, *fz_cap = nullptr, *fz_it_cand = nullptr, *fz_nb = nullptr, *fz_work = nullptr;
    zn::FzItem *fz_items = nullptr;
    uint32_t fz_total = 0;
    uint64_t fz_bytes = 0;  // content bytes of all candidates
    // batch path (k_bx_*): candidate slots (one per compressed row at most), item slots, the two entropy work lists
    uint32_t bx_slots = 0, bx_item_cap = 0;
    uint64_t bx_bytes = 0;  // content bytes of all compressed rows
    uint64_t bx_nblk = 0;   // their 128 KiB blocks, as the index columns have them
    uint32_t *bx_cand_row = nullptr, *bx_cand_base = nullptr, *bx_cand_nb = nullptr, *bx_huf_list = nullptr, *bx_seq_list = nullptr, *bx_sort_tmp = nullptr;
    zn::FzItem *bx_items = nullptr;
    zn::BxPrep *bx_prep = nullptr;
    uint64_t rx_words = 0;  // resolve path: words its frames (compressed rows of >= RX_MIN bytes) can ask for
    uint64_t rx_words_small = 0;  // ... counting every row above 64 KiB
    uint32_t rx_min = zn::RX_MIN;  // a table whose rows above 64 KiB are few (<= 256 M words) resolves all of them: one wave per frame is the slower way
                                   // when the frames do not fill the chip (the image's source text: its 64-256 KiB frames were 1.6 ms of one-wave execution)
    uint32_t *rx_base = nullptr, *rx_fail = nullptr, *rx_blk = nullptr, *rx_list = nullptr;
    uint32_t *item_row = nullptr, *item_k = nullptr, *item_src = nullptr, *row_flag = nullptr;
    // fused block kernel: big-slice tiles of the candidate rows, the item each belongs to, and what it got done
    uint32_t n_bt = 0;
    uint32_t *bt_tile = nullptr, *bt_item = nullptr;
    uint8_t *tile_done = nullptr, *item_done = nullptr;  // one allocation (item_done lies behind tile_done)
    size_t done_bytes = 0;
    uint32_t *todo = nullptr;  // items left to the block decoder (count: third word of the control block)
    uint32_t n_small_tiles = 0;     // tiles of whole small rows (the fused kernels' work)
    uint32_t *slow_list = nullptr;  // tiles the role-split kernel leaves to k_fused_small (count: fourth word of the control block)
    DevPlan plan;
};

struct znippy_rounds {
    znippy_ctx *ctx = nullptr;
    uint32_t n = 0;
    uint64_t *src_off = nullptr, *len = nullptr;
    uint8_t *skip = nullptr;
    uint32_t *digests = nullptr;
    std::vector<uint8_t> h_skip;  // empty: no round is a store-path round
    void *plan_scratch[3] = {nullptr, nullptr, nullptr};  // per-round prefix sums of the plan kernels (freed with the table)
    uint64_t in_bytes = 0, enc_bytes = 0;  // all rounds / rounds that go through the encoder
    bool all_stored_aligned = false;       // every round is a skip round and every blob offset will be a multiple of 16
    uint64_t blob_bound = 0;
    DevPlan plan;
    // encoder plan: one item per output piece
    EncItem *items = nullptr;
    uint32_t n_items = 0;
    // Encoder variant per block, by the block's own length (so a round's frame does not depend on what else is
    // in the batch): blocks <= 16 KiB go to the small-table variant (more waves), the rest to the wide one.
    // order_*: item indices of each share; NULL when the whole plan is of one kind.
    uint32_t *order_small = nullptr, *order_wide = nullptr;
    uint32_t n_small = 0, n_wide = 0;
    uint32_t *retry_list = nullptr, *retry_count = nullptr;  // small blocks the small variant hands to the wide one
    uint64_t prov_bytes = 0;
    uint32_t *piece_len = nullptr, *piece_len_init = nullptr;
    uint64_t *piece_start = nullptr, *local_excl = nullptr, *block_tot = nullptr;
    // results live in ONE device slab (one D2H per call): [total u64][overflow u64][blob_offset n][blob_size n][digests 32n]
    // Two slabs + two mirrors + one event each: run k uses slot k & 1 and its D2H copy rides the context's copy
    // stream, so run k + 1 encodes while run k's results travel (and are read: znippy_rounds_results_lagged).
    uint8_t *res = nullptr, *h_res = nullptr;  // device slab + pinned host mirror of the CURRENT run's slot
    uint8_t *res_m[2] = {nullptr, nullptr}, *h_res_m[2] = {nullptr, nullptr};
    size_t h_res_cap_m[2] = {0, 0}, h_stored_cap = 0;
    hipEvent_t ev_enc[2] = {nullptr, nullptr}, ev_res[2] = {nullptr, nullptr};
    uint64_t run_seq = 0;
    size_t res_bytes = 0;
    bool store_incompressible = false;  // opt-in (znippy_rounds_set_store_incompressible)
    int fuse_tiles = 0;  // > 0: every round is a small encoded round (one block, no store path): the encoder hashes its tiles itself, this many per dequeue
    uint32_t *first_item = nullptr;     // first piece of every round
    uint8_t *stored = nullptr, *h_stored = nullptr;  // per round: turned into a raw payload by the opt-in pass
    uint64_t *blob_offset = nullptr, *blob_size = nullptr, *total = nullptr;
    uint32_t *overflow = nullptr;
};

// ------------------------------------------------------------------------------------------------
static bool ktime_on(const znippy_ctx *ctx, const char *name) {
    return ctx->sw.ktime >= 2 || (ctx->sw.ktime == 1 && !strncmp(name, "decode_verify_", 14));
}
static void ktime_begin(znippy_ctx *ctx, const char *name, hipStream_t on = nullptr) {
    if (ctx->sw.trace) { fprintf(stderr, "[znippy trace] %s ...", name); fflush(stderr); }
    ctx->ktime_open = ktime_on(ctx, name);
    if (!ctx->ktime_open) return;
    if ((int)ctx->ktimes.size() <= ctx->n_ktimes) {
        KTime k{name, nullptr, nullptr};
        (void)hipEventCreate(&k.t0);
        (void)hipEventCreate(&k.t1);
        ctx->ktimes.push_back(k);
    }
    ctx->ktimes[ctx->n_ktimes].name = name;
    (void)hipEventRecord(ctx->ktimes[ctx->n_ktimes].t0, on ? on : ctx->stream);
}
static void ktime_end(znippy_ctx *ctx, hipStream_t on = nullptr) {
    if (ctx->sw.trace) {  // diagnosis of a faulting launch: every bracketed launch runs to its end before the next starts
        const hipError_t e = hipStreamSynchronize(on ? on : ctx->stream);
        fprintf(stderr, " %s\n", e == hipSuccess ? "done" : hipGetErrorString(e));
        fflush(stderr);
    }
    if (!ctx->ktime_open) return;
    (void)hipEventRecord(ctx->ktimes[ctx->n_ktimes].t1, on ? on : ctx->stream);
    ctx->n_ktimes++;
}

template <class T>
static int dev_upload(znippy_ctx *ctx, T **d, const T *h, size_t n) {
    HIPCHK(ctx, tmalloc(ctx, d, sizeof(T) * n));
    if (n) HIPCHK(ctx, hipMemcpy(*d, h, sizeof(T) * n, hipMemcpyHostToDevice));
    return ZNIPPY_OK;
}

// Scratch that only one side of the path needs is allocated on that side's first call (a verify-only context
// never pays for the encoder's 400 MB of sequence scratch, nor an encode-only one for the literal scratch).
static int ensure_decoder(znippy_ctx *ctx) {
    if (ctx->lit_scratch) return ZNIPPY_OK;
    if (hipMalloc(&ctx->lit_scratch, decode_lit_scratch_bytes(ctx->decode_grid)) != hipSuccess) return ZNIPPY_E_NOMEM;
    return ZNIPPY_OK;
}
// Pools of the two-phase foreign-frame path.  Literals: a block's literals never exceed what it regenerates, so the
// candidates' content bytes (+ 80 bytes of slack per item) always suffice.  Sequence records (8 bytes each): real data
// runs at one sequence per 8-20 bytes; 1.5 bytes of pool per content byte covers one per 5.3 bytes, and a block that
// finds the pool empty simply stays with the serial decoder.  Both are capped (a 100 GB archive does not get 250 GB
// of scratch): what does not fit is decoded serially.
static int ensure_fz_pools(znippy_ctx *ctx, uint64_t content_bytes, uint32_t items) {
    constexpr uint64_t CAP = 16ull << 30;
    const uint64_t lit = std::min<uint64_t>(content_bytes + 80ull * items + 4096, CAP);
    const uint64_t seq = std::min<uint64_t>(content_bytes * 3 / 2 + 4096, CAP) / 8;
    if (lit > ctx->fz_lit_cap) {
        (void)hipStreamSynchronize(ctx->stream);
        if (ctx->fz_lit_pool) (void)hipFree(ctx->fz_lit_pool);
        ctx->fz_lit_pool = nullptr; ctx->fz_lit_cap = 0;
        if (hipMalloc(&ctx->fz_lit_pool, lit) != hipSuccess) { (void)hipGetLastError(); ctx->fz_lit_pool = nullptr; return ZNIPPY_OK; }  // no pool: the serial decoder keeps the frames
        ctx->fz_lit_cap = lit;
    }
    if (seq > ctx->fz_seq_cap) {
        (void)hipStreamSynchronize(ctx->stream);
        if (ctx->fz_seq_pool) (void)hipFree(ctx->fz_seq_pool);
        ctx->fz_seq_pool = nullptr; ctx->fz_seq_cap = 0;
        if (hipMalloc(&ctx->fz_seq_pool, seq * 8) != hipSuccess) { (void)hipGetLastError(); ctx->fz_seq_pool = nullptr; return ZNIPPY_OK; }
        ctx->fz_seq_cap = seq;
    }
    return ZNIPPY_OK;
}
// Table pools of the batch path: a 10 KiB text frame needs ~1.5 KB of FSE cells and ~2 KB of Huffman cells; half a byte of
// each per content byte (+ 64 bytes per item) covers frames down to ~1 KiB, and a block that finds a pool empty stays with
// the serial decoder.
static int ensure_bx_pools(znippy_ctx *ctx, uint64_t content_bytes, uint32_t items) {
    constexpr uint64_t CAP = 16ull << 30;
    const uint64_t bytes = std::min<uint64_t>(content_bytes / 2 + 64ull * items + 4096, CAP);
    const uint64_t fse = bytes / 2 + BX_POOL_FIRST, huf = bytes / 2;
    if (fse > ctx->bx_fse_cap) {
        (void)hipStreamSynchronize(ctx->stream);
        if (ctx->bx_fse_pool) (void)hipFree(ctx->bx_fse_pool);
        ctx->bx_fse_pool = nullptr; ctx->bx_fse_cap = 0;
        if (hipMalloc(&ctx->bx_fse_pool, fse * 2) != hipSuccess) { (void)hipGetLastError(); ctx->bx_fse_pool = nullptr; return ZNIPPY_OK; }
        uint16_t predef[BX_POOL_FIRST];
        bx_predefined_tables(predef);
        if (hipMemcpy(ctx->bx_fse_pool, predef, sizeof predef, hipMemcpyHostToDevice) != hipSuccess) return ZNIPPY_E_HIP;
        ctx->bx_fse_cap = fse;
    }
    if (huf > ctx->bx_huf_cap) {
        (void)hipStreamSynchronize(ctx->stream);
        if (ctx->bx_huf_pool) (void)hipFree(ctx->bx_huf_pool);
        ctx->bx_huf_pool = nullptr; ctx->bx_huf_cap = 0;
        if (hipMalloc(&ctx->bx_huf_pool, huf * 2) != hipSuccess) { (void)hipGetLastError(); ctx->bx_huf_pool = nullptr; return ZNIPPY_OK; }
        ctx->bx_huf_cap = huf;
    }
    return ZNIPPY_OK;
}
// Word pool of the resolve path: 4 bytes per output byte of the frames it takes, at most 2^31 - 2^20 words (a word with bit 31
// clear is an index).  hipMalloc of 8 GiB takes 0.3 ms on this system; frames that find no room are executed by a wave each.
static int ensure_rx_pool(znippy_ctx *ctx, uint64_t words) {
    constexpr uint64_t CAP = (1ull << 31) - (1ull << 20);
    words = std::min<uint64_t>((words + 1023) & ~1023ull, CAP);
    if (words <= ctx->rx_cap) return ZNIPPY_OK;
    (void)hipStreamSynchronize(ctx->stream);
    if (ctx->rx_pool) (void)hipFree(ctx->rx_pool);
    if (ctx->rx_chunk) (void)hipFree(ctx->rx_chunk);
    if (ctx->rx_cdone) (void)hipFree(ctx->rx_cdone);
    ctx->rx_pool = nullptr; ctx->rx_chunk = nullptr; ctx->rx_cdone = nullptr; ctx->rx_cap = 0;
    if (hipMalloc(&ctx->rx_pool, words * 4) != hipSuccess || hipMalloc(&ctx->rx_chunk, words / 1024 * 4) != hipSuccess || hipMalloc(&ctx->rx_cdone, words / 1024) != hipSuccess) {
        (void)hipGetLastError();
        if (ctx->rx_pool) (void)hipFree(ctx->rx_pool);
        if (ctx->rx_chunk) (void)hipFree(ctx->rx_chunk);
        ctx->rx_pool = nullptr; ctx->rx_chunk = nullptr; ctx->rx_cdone = nullptr;
        return ZNIPPY_OK;  // no pool: the frames are executed by a wave each
    }
    ctx->rx_cap = words;
    return ZNIPPY_OK;
}
static int ensure_encoder(znippy_ctx *ctx) {
    if (ctx->enc_tabs) return ZNIPPY_OK;
    EncTables t;
    build_encode_tables(&t);
    if (hipMalloc(&ctx->enc_seq, (size_t)ctx->encode_grid * MAX_SEQ * 3 * 4) != hipSuccess ||
        hipMalloc(&ctx->enc_tabs, sizeof(EncTables)) != hipSuccess ||
        hipMemcpy(ctx->enc_tabs, &t, sizeof t, hipMemcpyHostToDevice) != hipSuccess) {
        if (ctx->enc_tabs) { (void)hipFree(ctx->enc_tabs); ctx->enc_tabs = nullptr; }
        return ZNIPPY_E_NOMEM;
    }
    return ZNIPPY_OK;
}

// rows of a front-to-back table arrive as two 32-bit size columns (zn_rows_pack32): the four 64-bit columns the kernels read
// are made here — sizes widened, offsets = first offset + running sum (per 1,024 rows: sums, their scan, fill)
__device__ __forceinline__ void block_excl_scan2(unsigned long long &a, unsigned long long &b, unsigned long long *sa, unsigned long long *sb,
                                                 unsigned long long *ta, unsigned long long *tb) {  // 256 threads: exclusive sums + totals
    const uint32_t t = threadIdx.x;
    const unsigned long long a0 = a, b0 = b;
    sa[t] = a; sb[t] = b;
    __syncthreads();
    for (uint32_t d = 1; d < 256; d <<= 1) {
        unsigned long long xa = 0, xb = 0;
        if (t >= d) { xa = sa[t - d]; xb = sb[t - d]; }
        __syncthreads();
        sa[t] += xa; sb[t] += xb;
        __syncthreads();
    }
    *ta = sa[255]; *tb = sb[255];
    a = sa[t] - a0; b = sb[t] - b0;
    __syncthreads();
}
__global__ __launch_bounds__(256) void k_rows_unpack_sums(const uint32_t *bs32, const uint32_t *us32, uint32_t n, unsigned long long *sums) {
    __shared__ unsigned long long sa[256], sb[256];
    const uint32_t lo = blockIdx.x * 1024 + threadIdx.x * 4;
    unsigned long long a = 0, b = 0, ta, tb;
    for (uint32_t k = 0; k < 4; k++) if (lo + k < n) { a += bs32[lo + k]; b += us32[lo + k]; }
    block_excl_scan2(a, b, sa, sb, &ta, &tb);
    if (threadIdx.x == 0) { sums[2 * blockIdx.x] = ta; sums[2 * blockIdx.x + 1] = tb; }
}
__global__ __launch_bounds__(256) void k_rows_unpack_scan(unsigned long long *sums, uint32_t nblk) {  // exclusive, in place, one workgroup
    __shared__ unsigned long long sa[256], sb[256];
    unsigned long long ca = 0, cb = 0;
    for (uint32_t b0 = 0; b0 < nblk; b0 += 256) {
        const uint32_t i = b0 + threadIdx.x;
        unsigned long long a = i < nblk ? sums[2 * i] : 0, b = i < nblk ? sums[2 * i + 1] : 0, ta, tb;
        block_excl_scan2(a, b, sa, sb, &ta, &tb);
        if (i < nblk) { sums[2 * i] = ca + a; sums[2 * i + 1] = cb + b; }
        ca += ta; cb += tb;
    }
}
__global__ __launch_bounds__(256) void k_rows_unpack_fill(const uint32_t *bs32, const uint32_t *us32, uint32_t n, const unsigned long long *sums,
                                                          uint64_t bo0, uint64_t oo0, uint64_t *bo, uint64_t *bs, uint64_t *oo, uint64_t *us) {
    __shared__ unsigned long long sa[256], sb[256];
    const uint32_t lo = blockIdx.x * 1024 + threadIdx.x * 4;
    uint32_t vb[4] = {0, 0, 0, 0}, vu[4] = {0, 0, 0, 0};
    unsigned long long a = 0, b = 0, ta, tb;
    for (uint32_t k = 0; k < 4; k++) if (lo + k < n) { vb[k] = bs32[lo + k]; vu[k] = us32[lo + k]; a += vb[k]; b += vu[k]; }
    block_excl_scan2(a, b, sa, sb, &ta, &tb);
    unsigned long long pb = bo0 + sums[2 * blockIdx.x] + a, po = oo0 + sums[2 * blockIdx.x + 1] + b;
    for (uint32_t k = 0; k < 4; k++) if (lo + k < n) {
        bo[lo + k] = pb; bs[lo + k] = vb[k]; oo[lo + k] = po; us[lo + k] = vu[k];
        pb += vb[k]; po += vu[k];
    }
}

__global__ void k_iota32(uint32_t *p, uint32_t n) {
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) p[i] = i;
}

// rows: the bit column -> one byte per row, and a stored row's length = its blob (the reference hashes and writes the
// blob bytes of a stored row and never looks at the index's uncompressed_size for it, decompress.rs:L143-166)
__global__ void k_rows_fixup(const uint8_t *bitmap, uint64_t row_begin, uint32_t n, uint8_t *comp, uint64_t *usize, const uint64_t *blob_size) {
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const uint64_t row = row_begin + i;
    const uint8_t c = bitmap ? (bitmap[row >> 3] >> (row & 7)) & 1 : 1;
    comp[i] = c;
    if (!c) usize[i] = blob_size[i];
}

// ---- the encoder plan of a Round table, built on the device (znippy_rounds_create) ----
// One item per output piece, in round order: an encoded round is ceil(len / 128 KiB) blocks (item.prov = the block's
// provisional slot), a store-path round ceil(len / 64 KiB) pieces (item.prov = the piece's offset inside the round).
struct RoundCount { uint32_t items, small, wide; uint64_t prov; };
__device__ __forceinline__ RoundCount round_count(uint64_t L, uint32_t sk) {
    RoundCount c;
    if (sk) { c.items = (uint32_t)(L ? (L + SKIP_PIECE - 1) / SKIP_PIECE : 1); c.small = 0; c.wide = 0; c.prov = 0; return c; }
    const uint32_t nb = (uint32_t)(L ? (L + BLOCK_BYTES - 1) / BLOCK_BYTES : 1);
    const uint32_t tail = (uint32_t)(L - (uint64_t)(nb - 1) * BLOCK_BYTES), tw = tail > 16 * 1024 ? 1u : 0u;
    c.items = nb; c.wide = nb - 1 + tw; c.small = 1 - tw;
    c.prov = (uint64_t)(nb - 1) * enc_slot_bytes(BLOCK_BYTES) + enc_slot_bytes(tail);
    return c;
}
// Three small kernels: per-workgroup sums of the four per-round counts (1,024 rounds per workgroup), their exclusive scan
// (one workgroup), and the fill — every workgroup scans its own 1,024 rounds again on chip, adds its base and writes its
// rounds' items.  (A first version scanned all rounds in ONE workgroup, every thread a contiguous run: 0.42 ms per 100k
// rounds of strided reads, in front of the table's first encode.)
struct RoundSums { uint32_t items, small, wide, pad; uint64_t prov; };
__device__ __forceinline__ void round_block_scan(uint32_t &ai, uint32_t &as, uint32_t &aw, uint64_t &ap, uint32_t *s_i, uint32_t *s_s, uint32_t *s_w, uint64_t *s_p,
                                                 RoundSums *total) {
    // inclusive scan over the 256 threads of the workgroup (values in / exclusive prefixes out); *total = the workgroup's sums
    const uint32_t t = threadIdx.x;
    const uint32_t vi = ai, vs = as, vw = aw;
    const uint64_t vp = ap;
    s_i[t] = ai; s_s[t] = as; s_w[t] = aw; s_p[t] = ap;
    __syncthreads();
    for (uint32_t d = 1; d < 256; d <<= 1) {
        uint32_t bi = 0, bs = 0, bw = 0; uint64_t bp = 0;
        if (t >= d) { bi = s_i[t - d]; bs = s_s[t - d]; bw = s_w[t - d]; bp = s_p[t - d]; }
        __syncthreads();
        s_i[t] += bi; s_s[t] += bs; s_w[t] += bw; s_p[t] += bp;
        __syncthreads();
    }
    ai = s_i[t] - vi; as = s_s[t] - vs; aw = s_w[t] - vw; ap = s_p[t] - vp;
    if (total) { total->items = s_i[255]; total->small = s_s[255]; total->wide = s_w[255]; total->pad = 0; total->prov = s_p[255]; }
    __syncthreads();
}
__global__ __launch_bounds__(256) void k_rounds_sums(const uint64_t *len, const uint8_t *skip, uint32_t n, RoundSums *sums) {
    __shared__ uint32_t s_i[256], s_s[256], s_w[256];
    __shared__ uint64_t s_p[256];
    const uint32_t lo = blockIdx.x * 1024 + threadIdx.x * 4;
    uint32_t ai = 0, as = 0, aw = 0;
    uint64_t ap = 0;
    for (uint32_t i = lo; i < lo + 4 && i < n; i++) { const RoundCount c = round_count(len[i], skip ? skip[i] : 0); ai += c.items; as += c.small; aw += c.wide; ap += c.prov; }
    RoundSums tot;
    round_block_scan(ai, as, aw, ap, s_i, s_s, s_w, s_p, &tot);
    if (threadIdx.x == 0) sums[blockIdx.x] = tot;
}
__global__ __launch_bounds__(256) void k_rounds_scan(RoundSums *sums, uint32_t nblk) {  // exclusive scan of the workgroup sums, in place
    __shared__ uint32_t s_i[256], s_s[256], s_w[256];
    __shared__ uint64_t s_p[256];
    RoundSums carry{0, 0, 0, 0, 0};
    for (uint32_t b0 = 0; b0 < nblk; b0 += 256) {
        const uint32_t b = b0 + threadIdx.x;
        RoundSums v{0, 0, 0, 0, 0};
        if (b < nblk) v = sums[b];
        uint32_t ai = v.items, as = v.small, aw = v.wide;
        uint64_t ap = v.prov;
        RoundSums tot;
        round_block_scan(ai, as, aw, ap, s_i, s_s, s_w, s_p, &tot);
        if (b < nblk) sums[b] = RoundSums{carry.items + ai, carry.small + as, carry.wide + aw, 0, carry.prov + ap};
        carry.items += tot.items; carry.small += tot.small; carry.wide += tot.wide; carry.prov += tot.prov;
    }
}
__global__ __launch_bounds__(256) void k_rounds_fill(const uint64_t *len, const uint8_t *skip, uint32_t n, const RoundSums *sums, uint32_t *first_item,
                                                      EncItem *items, uint32_t *plen, uint32_t *ord_small, uint32_t *ord_wide) {
    __shared__ uint32_t s_i[256], s_s[256], s_w[256];
    __shared__ uint64_t s_p[256];
    const uint32_t lo = blockIdx.x * 1024 + threadIdx.x * 4;
    RoundCount c[4];
    uint32_t ai = 0, as = 0, aw = 0;
    uint64_t ap = 0;
    for (uint32_t q = 0; q < 4; q++) {
        const uint32_t i = lo + q;
        c[q] = i < n ? round_count(len[i], skip ? skip[i] : 0) : RoundCount{0, 0, 0, 0};
        ai += c[q].items; as += c[q].small; aw += c[q].wide; ap += c[q].prov;
    }
    round_block_scan(ai, as, aw, ap, s_i, s_s, s_w, s_p, nullptr);
    const RoundSums base = sums[blockIdx.x];
    uint32_t at = base.items + ai, so = base.small + as, wo = base.wide + aw;
    uint64_t prov = base.prov + ap;
    for (uint32_t q = 0; q < 4; q++) {
        const uint32_t i = lo + q;
        if (i >= n) break;
        const uint64_t L = len[i];
        first_item[i] = at;
        if (skip && skip[i]) {
            const uint32_t np = c[q].items;
            for (uint32_t k = 0; k < np; k++) {
                const uint64_t o = (uint64_t)k * SKIP_PIECE;
                items[at + k] = EncItem{i, k, np, ITEM_SKIP | (k == 0 ? ITEM_FIRST : 0u), o};
                plen[at + k] = (uint32_t)(L - o < SKIP_PIECE ? L - o : SKIP_PIECE);
            }
            at += np;
            continue;
        }
        const uint32_t nb = c[q].items;
        for (uint32_t k = 0; k < nb; k++) {
            const uint32_t bl = (uint32_t)(L - (uint64_t)k * BLOCK_BYTES < BLOCK_BYTES ? L - (uint64_t)k * BLOCK_BYTES : BLOCK_BYTES);
            if (bl > 16 * 1024) ord_wide[wo++] = at + k; else ord_small[so++] = at + k;
            items[at + k] = EncItem{i, k, nb, k == 0 ? ITEM_FIRST : 0u, prov};
            plen[at + k] = 0;
            prov += enc_slot_bytes(bl);
        }
        at += nb;
    }
}

extern "C" {

int znippy_ctx_create(int device, void *hip_stream, znippy_ctx **out) {
    if (!out) return ZNIPPY_E_INVAL;
    znippy_ctx *ctx = new znippy_ctx();
    ctx->device = device;
    read_switches(ctx);
    if (hipSetDevice(device) != hipSuccess) { delete ctx; return ZNIPPY_E_HIP; }
    if (hip_stream) ctx->stream = (hipStream_t)hip_stream;
    else {
        if (hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking) != hipSuccess) { delete ctx; return ZNIPPY_E_HIP; }
        ctx->own_stream = true;
    }
    init_fused_tables();
    if (hipStreamCreateWithFlags(&ctx->aux, hipStreamNonBlocking) != hipSuccess ||
        hipStreamCreateWithFlags(&ctx->copy, hipStreamNonBlocking) != hipSuccess ||
        hipEventCreateWithFlags(&ctx->ev_fork, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&ctx->ev_join, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&ctx->ev_join2, hipEventDisableTiming) != hipSuccess) {
        znippy_ctx_destroy(ctx);
        return ZNIPPY_E_HIP;
    }
    ctx->decode_grid = decode_grid_size(device);
    if (hipMalloc(&ctx->cursor, 64) != hipSuccess) {
        znippy_ctx_destroy(ctx);
        return ZNIPPY_E_NOMEM;
    }
    {
        hipDeviceProp_t p;
        int cus = hipGetDeviceProperties(&p, device) == hipSuccess ? p.multiProcessorCount : 256;
        ctx->cus = cus;
        ctx->encode_grid = cus * 8;         // 16 KiB hash table per wave
        ctx->encode_grid_small = cus * 16;  // 4 KiB hash table per wave
    }
    *out = ctx;
    return ZNIPPY_OK;
}

static void ctx_teardown(znippy_ctx *ctx);
void znippy_ctx_destroy(znippy_ctx *ctx) {
    if (!ctx || ctx->closing) return;
    if (ctx->live_tables) {  // tables outlive the call: they keep the context's memory until the last of them is destroyed
        (void)hipSetDevice(ctx->device);
        (void)hipStreamSynchronize(ctx->stream);
        ctx->closing = true;
        return;
    }
    ctx_teardown(ctx);
}
static void table_released(znippy_ctx *ctx) {
    if (--ctx->live_tables == 0 && ctx->closing) ctx_teardown(ctx);
}
static void ctx_teardown(znippy_ctx *ctx) {
    (void)hipSetDevice(ctx->device);
    (void)hipStreamSynchronize(ctx->stream);
    for (auto &k : ctx->ktimes) { (void)hipEventDestroy(k.t0); (void)hipEventDestroy(k.t1); }
    if (ctx->lit_scratch) (void)hipFree(ctx->lit_scratch);
    if (ctx->lit_scratch_b) (void)hipFree(ctx->lit_scratch_b);
    if (ctx->fz_lit_pool) (void)hipFree(ctx->fz_lit_pool);
    if (ctx->fz_seq_pool) (void)hipFree(ctx->fz_seq_pool);
    if (ctx->bx_fse_pool) (void)hipFree(ctx->bx_fse_pool);
    if (ctx->bx_huf_pool) (void)hipFree(ctx->bx_huf_pool);
    if (ctx->rx_pool) (void)hipFree(ctx->rx_pool);
    if (ctx->rx_chunk) (void)hipFree(ctx->rx_chunk);
    if (ctx->rx_cdone) (void)hipFree(ctx->rx_cdone);
    if (ctx->cursor) (void)hipFree(ctx->cursor);
    if (ctx->shim_in) (void)hipFree(ctx->shim_in);
    if (ctx->shim_out) (void)hipFree(ctx->shim_out);
    if (ctx->enc_prov) (void)hipFree(ctx->enc_prov);
    if (ctx->enc_seq) (void)hipFree(ctx->enc_seq);
    if (ctx->enc_tabs) (void)hipFree(ctx->enc_tabs);
    if (ctx->aux) { (void)hipStreamSynchronize(ctx->aux); (void)hipStreamDestroy(ctx->aux); }
    if (ctx->copy) { (void)hipStreamSynchronize(ctx->copy); (void)hipStreamDestroy(ctx->copy); }
    for (auto &e : ctx->pinned_pool) (void)hipHostFree(e.second);
    for (auto &e : ctx->dev_pool) (void)hipFree(e.second);
    for (hipEvent_t e : ctx->event_pool) (void)hipEventDestroy(e);
    if (ctx->ev_fork) (void)hipEventDestroy(ctx->ev_fork);
    if (ctx->ev_join) (void)hipEventDestroy(ctx->ev_join);
    if (ctx->ev_join2) (void)hipEventDestroy(ctx->ev_join2);
    if (ctx->own_stream) (void)hipStreamDestroy(ctx->stream);
    delete ctx;
}

const char *znippy_last_error(const znippy_ctx *ctx) { return ctx ? ctx->err.c_str() : "null context"; }

int znippy_ctx_sync(znippy_ctx *ctx) {
    if (ctx && ctx->closing) return ZNIPPY_E_INVAL;  // destroyed context kept alive by its tables
    if (!ctx) return ZNIPPY_E_INVAL;
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    if (ctx->copy) HIPCHK(ctx, hipStreamSynchronize(ctx->copy));
    return ZNIPPY_OK;
}

int znippy_last_kernel_times(znippy_ctx *ctx, const char **names, float *ms, int cap) {
    if (ctx && ctx->closing) return ZNIPPY_E_INVAL;  // destroyed context kept alive by its tables
    if (!ctx) return 0;
    int n = std::min(cap, ctx->n_ktimes);
    for (int i = 0; i < n; i++) {
        names[i] = ctx->ktimes[i].name;
        ms[i] = 0.f;
        (void)hipEventElapsedTime(&ms[i], ctx->ktimes[i].t0, ctx->ktimes[i].t1);
    }
    return n;
}

int znippy_rows_foreign_stats(znippy_ctx *ctx, znippy_rows *r, uint64_t stats[8]) {
    if (ctx && ctx->closing) return ZNIPPY_E_INVAL;  // destroyed context kept alive by its tables
    if (!ctx || !r || r->ctx != ctx || !stats) return ZNIPPY_E_INVAL;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    // the path that ran for this table: the batch path's counters sit at +256, the round-2 two-phase path's at +192
    HIPCHK(ctx, hipMemcpy(stats, r->ctl + (r->bx_slots ? 256 : 192), 64, hipMemcpyDeviceToHost));
    return ZNIPPY_OK;
}

int znippy_ctx_set_level(znippy_ctx *ctx, int level) {
    if (ctx && ctx->closing) return ZNIPPY_E_INVAL;  // destroyed context kept alive by its tables
    if (!ctx || level < 1 || level > 22) return ZNIPPY_E_INVAL;
    ctx->level = level;
    return ZNIPPY_OK;
}

int znippy_ctx_level(const znippy_ctx *ctx) { return ctx ? ctx->level : ZNIPPY_E_INVAL; }

int znippy_ctx_set_kernel_timing(znippy_ctx *ctx, int level) {
    if (ctx && ctx->closing) return ZNIPPY_E_INVAL;  // destroyed context kept alive by its tables
    if (!ctx || level < 0 || level > 2) return ZNIPPY_E_INVAL;
    ctx->sw.ktime = level;
    return ZNIPPY_OK;
}

int znippy_measure_blake3_pass_ns(znippy_ctx *ctx, float *ns_per_pass_per_simd, float *shader_ghz) {
    if (ctx && ctx->closing) return ZNIPPY_E_INVAL;  // destroyed context kept alive by its tables
    if (!ctx || !ns_per_pass_per_simd) return ZNIPPY_E_INVAL;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    return measure_b3_pass_ns(ctx->cus, ctx->stream, ns_per_pass_per_simd, shader_ghz);
}

// Shader clock a read-side kernel held during the last run with ZNIPPY_DBG bit 32768 set: one wave's life in shader
// cycles / in 100 MHz ticks (MI355X_MICROARCH.md, DVFS give-back (6)).  0 if nothing was recorded.
int znippy_last_shader_ghz(znippy_ctx *ctx, float *ghz) {
    if (ctx && ctx->closing) return ZNIPPY_E_INVAL;  // destroyed context kept alive by its tables
    if (!ctx || !ghz) return ZNIPPY_E_INVAL;
    *ghz = 0.f;
    if (!ctx->clk_buf) return ZNIPPY_OK;
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    unsigned long long h[2] = {0, 0};
    HIPCHK(ctx, hipMemcpy(h, ctx->clk_buf, 16, hipMemcpyDeviceToHost));
    if (h[1]) *ghz = (float)((double)h[0] / (double)h[1] * 0.1);
    return ZNIPPY_OK;
}

// ---- frame header (host) ------------------------------------------------------------------------
// Leading skippable frames (RFC 8878 3.1.2) carry no content: the single-chunk shims step over them — the size
// query and the decoder both start at the first Zstandard frame.  -1: a skippable frame runs past the input.
static ptrdiff_t skippable_prefix(const uint8_t *p, size_t n) {
    size_t at = 0;
    while (n - at >= 8) {
        uint32_t magic, sz;
        memcpy(&magic, p + at, 4);
        if ((magic & 0xFFFFFFF0u) != 0x184D2A50u) break;
        memcpy(&sz, p + at + 4, 4);
        if ((uint64_t)8 + sz > n - at) return -1;
        at += 8 + (size_t)sz;
    }
    return (ptrdiff_t)at;
}

int znippy_get_decompressed_size(const void *frame, size_t n, uint64_t *out_size) {
    const uint8_t *p = (const uint8_t *)frame;
    if (!p || !out_size) return ZNIPPY_E_INVAL;
    {
        const ptrdiff_t skip = skippable_prefix(p, n);
        if (skip < 0) return ZNIPPY_E_CORRUPT;
        p += skip;
        n -= (size_t)skip;
    }
    if (n < 5) return ZNIPPY_E_CORRUPT;
    uint32_t magic;
    memcpy(&magic, p, 4);
    if (magic != 0xFD2FB528u) return ZNIPPY_E_CORRUPT;
    uint32_t fhd = p[4], fcs_flag = fhd >> 6, single = (fhd >> 5) & 1, did_flag = fhd & 3;
    if (fhd & 8) return ZNIPPY_E_CORRUPT;
    uint32_t fcs_bytes = fcs_flag == 0 ? single : (1u << fcs_flag);
    uint32_t did_bytes = did_flag == 3 ? 4 : did_flag;
    size_t pos = 5 + (single ? 0 : 1) + did_bytes;
    if (n < pos + fcs_bytes) return ZNIPPY_E_CORRUPT;
    if (!fcs_bytes) return ZNIPPY_E_UNSUPPORTED;
    uint64_t fcs = 0;
    for (uint32_t i = 0; i < fcs_bytes; i++) fcs |= (uint64_t)p[pos + i] << (8 * i);
    if (fcs_bytes == 2) fcs += 256;
    *out_size = fcs;
    return ZNIPPY_OK;
}

// ---- rows ---------------------------------------------------------------------------------------

void znippy_rows_destroy(znippy_rows *r) {
    if (!r) return;
    (void)hipSetDevice(r->ctx->device);
    void *ptrs[] = {r->blob_off, r->blob_size, r->usize, r->out_off, r->compressed, r->checksum,
                    r->ctl, r->digests, r->corrupt, r->list_a, r->pending,
                    r->cand_row, r->cand_base, r->cand_nblocks, r->fz_base, r->fz_cap, r->fz_it_cand, r->fz_nb, r->fz_work, r->fz_items, r->item_row, r->item_k, r->item_src, r->row_flag, r->pending2,
                    r->bt_tile, r->bt_item, r->tile_done, r->todo, r->status_init, r->slow_list,
                    r->bx_cand_row, r->bx_cand_base, r->bx_cand_nb, r->bx_huf_list, r->bx_seq_list, r->bx_items, r->bx_prep, r->d_bitmap, r->bx_sort_tmp,
                    r->rx_base, r->rx_fail, r->rx_blk, r->rx_list, r->d_pack, r->d_pack_sums, r->all_rows};
    for (void *p : ptrs)
        tfree(r->ctx, p);
    if (r->h_counters) {
        (void)hipStreamSynchronize(r->ctx->stream);  // a queued run may still copy into the slot
        pinned_give(r->ctx, r->h_counters, r->h_counters_cap);
    }
    if (r->h_pack) {
        (void)hipStreamSynchronize(r->ctx->stream);  // (the copy out of it is stream-ordered)
        pinned_give(r->ctx, r->h_pack, r->h_pack_cap);
    }
    for (hipEvent_t e : r->ev_done) event_give(r->ctx, e);
    free_plan(r->ctx, r->plan);
    znippy_ctx *const c = r->ctx;
    delete r;
    table_released(c);
}

int znippy_rows_create(znippy_ctx *ctx, const uint64_t *blob_offset, const uint64_t *blob_size,
                       const uint8_t *compressed_bitmap, const uint64_t *uncompressed_size,
                       const uint64_t *out_offset, const uint8_t *checksum, uint64_t row_begin,
                       uint64_t row_end, znippy_rows **out) {
    if (ctx && ctx->closing) return ZNIPPY_E_INVAL;  // destroyed context kept alive by its tables
    if (!ctx || !out || row_end < row_begin || !blob_offset || !blob_size || !uncompressed_size || !out_offset)
        return ZNIPPY_E_INVAL;
    if (row_end - row_begin >= 0xFFFFFFF0ull) return ZNIPPY_E_INVAL;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    TDbg td(ctx->sw.tdbg, "rows_create");
    znippy_rows *r = new znippy_rows();
    r->ctx = ctx;
    ctx->live_tables++;
    r->row_begin = row_begin;
    r->n = (uint32_t)(row_end - row_begin);
    const uint32_t n = r->n;
    const uint64_t *const bo_in = blob_offset + row_begin, *const bs_in = blob_size + row_begin, *const us_in = uncompressed_size + row_begin,
                   *const oo_in = out_offset + row_begin;
    bool allc = true;  // every row compressed (the usual archive): the per-row passes skip the bit column
    if (compressed_bitmap) {
        uint64_t row = row_begin;
        for (; row < row_end && (row & 7); row++) allc &= (compressed_bitmap[row >> 3] >> (row & 7)) & 1;
        for (; row + 8 <= row_end && allc; row += 8) allc &= compressed_bitmap[row >> 3] == 0xFF;
        for (; row < row_end && allc; row++) allc &= (compressed_bitmap[row >> 3] >> (row & 7)) & 1;
    }
    auto comp_of = [&](uint32_t i) -> uint32_t { if (allc) return 1u; const uint64_t row = row_begin + i; return (compressed_bitmap[row >> 3] >> (row & 7)) & 1u; };
    // a stored row IS its blob (see k_rows_fixup): its effective length is blob_size
    auto len_of = [&](uint32_t i) -> uint64_t { return comp_of(i) ? us_in[i] : bs_in[i]; };
    int rc = ZNIPPY_OK;
    // the columns as they are (the device derives the byte-per-row flags and the stored rows' lengths)
    uint8_t *&d_bitmap = r->d_bitmap;  // (kept until the table goes: k_rows_fixup reads it on the stream, uploads are not stream-ordered)
    const uint64_t bm0 = row_begin >> 3, bm1 = (row_end + 7) >> 3;
    // A table written front to back is its two size columns (zn_rows_pack32): 8 bytes per row go to the device, from
    // page-locked memory, and three small kernels make the four 64-bit columns there.  (The four pageable copies of 0.8 MB
    // each were 0.18 of the 0.36 ms a table of 100k rows took to build.)
    bool packed = false;
    if (n >= 64 && !ctx->sw.no_pack) {
        r->h_pack = (uint32_t *)pinned_take(ctx, 8 * (size_t)n, &r->h_pack_cap);
        if (r->h_pack && zn_rows_pack32(bo_in, bs_in, oo_in, us_in, n, r->h_pack)) {
            const uint32_t nblk = (n + 1023) / 1024;
            if (tmalloc(ctx, &r->d_pack, 8 * (size_t)n) == hipSuccess && tmalloc(ctx, &r->d_pack_sums, 16 * (size_t)nblk) == hipSuccess &&
                tmalloc(ctx, &r->blob_off, 8 * (size_t)n) == hipSuccess && tmalloc(ctx, &r->blob_size, 8 * (size_t)n) == hipSuccess &&
                tmalloc(ctx, &r->usize, 8 * (size_t)n) == hipSuccess && tmalloc(ctx, &r->out_off, 8 * (size_t)n) == hipSuccess &&
                hipMemcpyAsync(r->d_pack, r->h_pack, 8 * (size_t)n, hipMemcpyHostToDevice, ctx->stream) == hipSuccess) {
                hipLaunchKernelGGL(k_rows_unpack_sums, dim3(nblk), dim3(256), 0, ctx->stream, r->d_pack, r->d_pack + n, n, r->d_pack_sums);
                hipLaunchKernelGGL(k_rows_unpack_scan, dim3(1), dim3(256), 0, ctx->stream, r->d_pack_sums, nblk);
                hipLaunchKernelGGL(k_rows_unpack_fill, dim3(nblk), dim3(256), 0, ctx->stream, r->d_pack, r->d_pack + n, n, r->d_pack_sums, bo_in[0], oo_in[0],
                                   r->blob_off, r->blob_size, r->out_off, r->usize);
                packed = true;
            } else {
                znippy_rows_destroy(r);
                return ZNIPPY_E_NOMEM;
            }
        }
    }
    if ((!packed && ((rc = dev_upload(ctx, &r->blob_off, bo_in, n)) || (rc = dev_upload(ctx, &r->blob_size, bs_in, n)) ||
                     (rc = dev_upload(ctx, &r->usize, us_in, n)) || (rc = dev_upload(ctx, &r->out_off, oo_in, n)))) ||
        (compressed_bitmap && (rc = dev_upload(ctx, &d_bitmap, compressed_bitmap + bm0, (size_t)(bm1 - bm0)))) ||
        tmalloc(ctx, &r->compressed, std::max<size_t>(n, 16)) != hipSuccess) {
        znippy_rows_destroy(r);
        return rc ? rc : ZNIPPY_E_NOMEM;
    }
    if (n) hipLaunchKernelGGL(k_rows_fixup, dim3((n + 255) / 256), dim3(256), 0, ctx->stream, d_bitmap, row_begin - 8 * bm0, n, r->compressed, r->usize, r->blob_size);
    td.mark("columns_h2d");
    if (checksum && (rc = dev_upload(ctx, &r->checksum, checksum + 32 * row_begin, (size_t)32 * n))) {
        znippy_rows_destroy(r);
        return rc;
    }
    td.mark("checksum_h2d");
    r->corrupt_cap = std::max<uint32_t>(n, 1);
    r->ctl_bytes = znippy_rows::CTL_HEAD + std::max<size_t>(4 * (size_t)n, 16);
    if (tmalloc(ctx, &r->ctl, r->ctl_bytes) != hipSuccess ||
        tmalloc(ctx, &r->digests, std::max<size_t>(32 * (size_t)n, 32)) != hipSuccess ||
        !(r->h_counters = (uint64_t *)pinned_take(ctx, 256, &r->h_counters_cap)) ||
        !(r->ev_done[0] = event_take(ctx)) || !(r->ev_done[1] = event_take(ctx)) ||
        tmalloc(ctx, &r->corrupt, 8 * (size_t)r->corrupt_cap) != hipSuccess) {
        znippy_rows_destroy(r);
        return ZNIPPY_E_NOMEM;
    }
    r->counters = reinterpret_cast<uint64_t *>(r->ctl);
    r->pending_count = reinterpret_cast<uint32_t *>(r->ctl + 64);
    r->cursor = reinterpret_cast<uint32_t *>(r->ctl + 128);
    r->status = reinterpret_cast<int32_t *>(r->ctl + znippy_rows::CTL_HEAD);
    td.mark("allocs");
    PlanBuf p;
    build_plan(len_of, n, p);
    td.mark("plan");
    if ((rc = upload_plan(ctx, p, r->plan))) {
        znippy_rows_destroy(r);
        return rc;
    }
    td.mark("plan_h2d");
    for (const Tile &t : p.tiles) r->n_small_tiles += t.n_units != 0;
    if (r->n_small_tiles && tmalloc(ctx, &r->slow_list, 4 * (size_t)p.tiles.size()) != hipSuccess) {
        znippy_rows_destroy(r);
        return ZNIPPY_E_NOMEM;
    }
    // compressed rows above 64 KiB: frames of >= 2 blocks (and < 4 GiB) are tried block by block (each block a work
    // item), the others go straight to the general decoder
    constexpr uint64_t BLK = 128 * 1024;
    std::vector<uint32_t> la, cand_row, cand_base, cand_nb, item_row, item_k, fz_base, fz_cap, fz_it_cand;
    uint64_t big_bytes = 0, big_blob = 0, n_big = 0, nblk = 0;
    // ONE pass over the caller's columns: the flags the runs need, the extents a run is validated against (rows_validate:
    // a table whose extents fit the run's regions has no bad row — the per-row pass is for the others), and the big rows
    bool scan_rows = true;
    if (allc) {  // the vectorised pass; the row-by-row one below only if the table has big rows
        uint64_t e[8];
        zn_rows_extents(bo_in, bs_in, oo_in, us_in, n, e);
        if (e[6] == 0) {
            scan_rows = false;
            r->ext_min_bo = e[0]; r->ext_max_bend = e[1]; r->ext_max_oend = e[2]; r->ext_wrap = e[3] != 0;
            r->n_compressed = n; r->bx_bytes = e[4]; nblk = e[5];
        }
    }
    if (scan_rows) {
        uint64_t min_bo = ~0ull, max_bend = 0, max_oend = 0;
        bool wrap = false;
        for (uint32_t i = 0; i < n; i++) {
            const uint32_t c = comp_of(i);
            const uint64_t bo = bo_in[i], bs = bs_in[i], oo = oo_in[i], us = c ? us_in[i] : bs;
            r->n_compressed += c;
            if (!c && (oo & 15)) r->odd_out = true;
            min_bo = std::min(min_bo, bo);
            wrap |= bo + bs < bo || oo + us < oo;
            max_bend = std::max(max_bend, bo + bs);
            max_oend = std::max(max_oend, oo + us);
            if (!c) continue;
            nblk += us ? (us + BLK - 1) / BLK : 1;
            r->bx_bytes += us;
            if (us > 65536 && us < (1ull << 30)) r->rx_words_small += (us + 1023) & ~1023ull;
            if (us >= zn::RX_MIN && us < (1ull << 30)) r->rx_words += (us + 1023) & ~1023ull;
            if (us <= 64 * 1024) continue;
            big_bytes += us;
            big_blob += bs;
            n_big++;
            const uint64_t nb = (us + BLK - 1) / BLK;
            if (nb >= 2 && us < 0xFFFFFFFFull && item_row.size() + nb < 0x7FFFFFFFull && !ctx->sw.no_block_items) {
                cand_row.push_back(i);
                cand_base.push_back((uint32_t)item_row.size());
                cand_nb.push_back((uint32_t)nb);
                for (uint32_t k = 0; k < nb; k++) { item_row.push_back(i); item_k.push_back(k); }
                if (!ctx->sw.no_fz && ctx->sw.no_bx && fz_it_cand.size() + 2 * nb + 8 < 0x7FFFFFFFull) {
                    const uint32_t cap = (uint32_t)(2 * nb + 8);
                    fz_base.push_back((uint32_t)fz_it_cand.size());
                    fz_cap.push_back(cap);
                    fz_it_cand.insert(fz_it_cand.end(), cap, (uint32_t)cand_row.size() - 1);
                    r->fz_bytes += us;
                } else { fz_base.push_back(0); fz_cap.push_back(0); }
            } else la.push_back(i);
        }
        r->ext_min_bo = min_bo; r->ext_max_bend = max_bend; r->ext_max_oend = max_oend; r->ext_wrap = wrap;
    }
    r->n_list_a = (uint32_t)la.size();
    td.mark("big_rows");
    if (r->n_compressed) {
        // what the block-item path, the batch path and the serial decoder tell each other about a row, and the serial
        // decoder's list
        if (tmalloc(ctx, &r->row_flag, std::max<size_t>(4 * (size_t)n, 16)) != hipSuccess ||
            tmalloc(ctx, &r->pending2, std::max<size_t>(4 * (size_t)r->n_compressed, 16)) != hipSuccess) {
            znippy_rows_destroy(r);
            return ZNIPPY_E_NOMEM;
        }
        if (!ctx->sw.no_bx) {
            // a writer may split blocks (libzstd's high levels cut a 128 KiB block into 2-5; runs of equal bytes come as
            // strings of small RLE blocks): half as many again + up to 64k more, shared by all frames.  Frames that find no
            // slot stay with the serial decoder.
            r->bx_nblk = nblk;
            if (r->rx_words_small && r->rx_words_small <= (256ull << 20)) { r->rx_min = 65537; r->rx_words = r->rx_words_small; }
            const uint64_t cap = nblk + nblk / 2 + std::min<uint64_t>(3 * nblk, 65536) + 1024;
            if (cap < 0x7FFFFFFFull) {
                r->bx_slots = r->n_compressed;
                r->bx_item_cap = (uint32_t)cap;
                if (tmalloc(ctx, &r->bx_cand_row, 4 * (size_t)r->bx_slots) != hipSuccess || tmalloc(ctx, &r->bx_cand_base, 4 * (size_t)r->bx_slots) != hipSuccess ||
                    tmalloc(ctx, &r->bx_cand_nb, 4 * (size_t)r->bx_slots) != hipSuccess || tmalloc(ctx, &r->bx_huf_list, 4 * (size_t)cap) != hipSuccess ||
                    tmalloc(ctx, &r->bx_seq_list, 4 * 4 * (size_t)cap) != hipSuccess || tmalloc(ctx, &r->bx_sort_tmp, 5 * 4 * (size_t)cap) != hipSuccess || tmalloc(ctx, &r->bx_items, sizeof(zn::FzItem) * (size_t)cap) != hipSuccess ||
                    tmalloc(ctx, &r->bx_prep, sizeof(zn::BxPrep) * (size_t)cap) != hipSuccess ||
                    ((r->rx_words || r->rx_words_small) && !ctx->sw.no_rx &&
                     (tmalloc(ctx, &r->rx_base, 4 * (size_t)r->bx_slots) != hipSuccess || tmalloc(ctx, &r->rx_fail, 4 * (size_t)r->bx_slots) != hipSuccess ||
                      tmalloc(ctx, &r->rx_blk, 16 * (size_t)cap) != hipSuccess || tmalloc(ctx, &r->rx_list, 4 * (size_t)cap) != hipSuccess))) {
                    znippy_rows_destroy(r);
                    return ZNIPPY_E_NOMEM;
                }
            }
        }
    }
    // 1024-thread workgroups pay off where a few very long copies dominate (multi-MiB frames of periodic or stored
    // data: a frame that is < 2 % of its content); entropy-coded frames are a serial bitstream and want the narrow
    // variant's window execution instead, whatever their size
    r->wide_rows = n_big && big_bytes / n_big >= (1u << 20) && big_blob * 50 < big_bytes;
    r->n_cand = (uint32_t)cand_row.size();
    r->n_items = (uint32_t)item_row.size();
    r->small_ok = allc && !ctx->sw.no_lean && !ctx->sw.no_bx && r->n_list_a == 0 && r->n_cand == 0 && r->n_small_tiles == (uint32_t)p.tiles.size() &&
                  r->n_small_tiles > 0 && r->bx_slots;
    r->lean_mixed_ok = !ctx->sw.no_lean && r->n_list_a == 0 && r->n_cand == 0 && r->n_small_tiles > 0 && r->n_small_tiles < (uint32_t)p.tiles.size();
    r->lean_blocks_ok = allc && !ctx->sw.no_lean && r->n_list_a == 0 && r->n_cand > 0 && r->n_small_tiles == 0;
    r->lean_ok = allc && !ctx->sw.no_lean && r->n_list_a == 0 && r->n_cand == 0 && p.big.empty() && r->n_small_tiles == (uint32_t)p.tiles.size() &&
                 r->n_small_tiles > 0;
    if (r->n_cand) {
        if ((rc = dev_upload(ctx, &r->cand_row, cand_row.data(), cand_row.size())) ||
            (rc = dev_upload(ctx, &r->cand_base, cand_base.data(), cand_base.size())) ||
            (rc = dev_upload(ctx, &r->cand_nblocks, cand_nb.data(), cand_nb.size())) ||
            (rc = dev_upload(ctx, &r->item_row, item_row.data(), item_row.size())) ||
            (rc = dev_upload(ctx, &r->item_k, item_k.data(), item_k.size()))) {
            znippy_rows_destroy(r);
            return rc;
        }
        if (tmalloc(ctx, &r->item_src, 4 * (size_t)r->n_items) != hipSuccess) {
            znippy_rows_destroy(r);
            return ZNIPPY_E_NOMEM;
        }
        r->fz_total = (uint32_t)fz_it_cand.size();
        if (r->fz_total) {
            if ((rc = dev_upload(ctx, &r->fz_base, fz_base.data(), fz_base.size())) ||
                (rc = dev_upload(ctx, &r->fz_cap, fz_cap.data(), fz_cap.size())) ||
                (rc = dev_upload(ctx, &r->fz_it_cand, fz_it_cand.data(), fz_it_cand.size()))) {
                znippy_rows_destroy(r);
                return rc;
            }
            if (tmalloc(ctx, &r->fz_nb, 4 * (size_t)r->n_cand) != hipSuccess || tmalloc(ctx, &r->fz_work, 4 * (size_t)r->fz_total) != hipSuccess ||
                tmalloc(ctx, &r->fz_items, sizeof(zn::FzItem) * (size_t)r->fz_total) != hipSuccess) {
                znippy_rows_destroy(r);
                return ZNIPPY_E_NOMEM;
            }
        }
    }
    if (r->n_cand && !ctx->sw.no_fused_blocks) {
        std::vector<uint32_t> row_base(n, 0xFFFFFFFFu), bt_tile, bt_item;
        for (size_t c = 0; c < cand_row.size(); c++) row_base[cand_row[c]] = cand_base[c];
        for (uint32_t ti = 0; ti < (uint32_t)p.tiles.size(); ti++) {
            const Tile &t = p.tiles[ti];
            if (t.n_units == 0 && row_base[t.first_unit] != 0xFFFFFFFFu) {
                bt_tile.push_back(ti);
                bt_item.push_back(row_base[t.first_unit] + (t.first_leaf >> 7));
            }
        }
        r->n_bt = (uint32_t)bt_tile.size();
        if ((rc = dev_upload(ctx, &r->bt_tile, bt_tile.data(), bt_tile.size())) ||
            (rc = dev_upload(ctx, &r->bt_item, bt_item.data(), bt_item.size()))) {
            znippy_rows_destroy(r);
            return rc;
        }
        // (the two arrays of done flags in one allocation: one clear per run instead of two)
        const size_t td_bytes = (std::max<size_t>(p.tiles.size(), 16) + 15) & ~(size_t)15;
        r->done_bytes = td_bytes + std::max<size_t>(r->n_items, 16);
        if (tmalloc(ctx, &r->tile_done, r->done_bytes) != hipSuccess ||
            tmalloc(ctx, &r->todo, std::max<size_t>(4 * (size_t)r->n_items, 16)) != hipSuccess) {
            znippy_rows_destroy(r);
            return ZNIPPY_E_NOMEM;
        }
        r->item_done = r->tile_done + td_bytes;
    }
    if ((rc = dev_upload(ctx, &r->list_a, la.data(), la.size()))) {
        znippy_rows_destroy(r);
        return rc;
    }
    if (tmalloc(ctx, &r->pending, std::max<size_t>(4 * (size_t)n, 16)) != hipSuccess ||
        false) {
        znippy_rows_destroy(r);
        return ZNIPPY_E_NOMEM;
    }
    td.mark("rest");
    *out = r;
    return ZNIPPY_OK;
}

int znippy_rows_set_blob_cap(znippy_rows *r, uint64_t blob_cap) {
    if (!r) return ZNIPPY_E_INVAL;
    r->blob_cap = blob_cap;
    return ZNIPPY_OK;
}

// Every row's source range against the declared blob region and its output range against out_cap, overflow-safe,
// once per distinct triple.  Bad rows (normally none) get their status from the host.
static int rows_validate(znippy_ctx *ctx, znippy_rows *r, uint64_t blob_base, uint64_t out_cap) {
    if (r->val_done && r->val_base == blob_base && r->val_bcap == r->blob_cap && r->val_ocap == out_cap) return ZNIPPY_OK;
    std::vector<int32_t> init;
    uint32_t bad = 0;
    const uint64_t bcap = r->blob_cap;
    const bool fits = !r->ext_wrap && (r->n == 0 || (r->ext_min_bo >= blob_base && (bcap == ~0ull || r->ext_max_bend - blob_base <= bcap) && r->ext_max_oend <= out_cap));
    if (!fits && r->h_blob_off.empty() && r->n) {  // the columns, for the per-row verdicts (the device copies hold the effective lengths)
        r->h_blob_off.resize(r->n); r->h_blob_size.resize(r->n); r->h_len.resize(r->n); r->h_out_off.resize(r->n);
        HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
        HIPCHK(ctx, hipMemcpy(r->h_blob_off.data(), r->blob_off, 8 * (size_t)r->n, hipMemcpyDeviceToHost));
        HIPCHK(ctx, hipMemcpy(r->h_blob_size.data(), r->blob_size, 8 * (size_t)r->n, hipMemcpyDeviceToHost));
        HIPCHK(ctx, hipMemcpy(r->h_len.data(), r->usize, 8 * (size_t)r->n, hipMemcpyDeviceToHost));
        HIPCHK(ctx, hipMemcpy(r->h_out_off.data(), r->out_off, 8 * (size_t)r->n, hipMemcpyDeviceToHost));
    }
    for (uint32_t i = 0; i < (fits ? 0u : r->n); i++) {
        const uint64_t bo = r->h_blob_off[i], bs = r->h_blob_size[i], len = r->h_len[i], oo = r->h_out_off[i];
        int code = 0;
        if (bo < blob_base) code = ZNIPPY_E_CORRUPT;
        else if (bcap != ~0ull && (bs > bcap || bo - blob_base > bcap - bs)) code = ZNIPPY_E_CORRUPT;
        else if (len > out_cap || oo > out_cap - len) code = ZNIPPY_E_DST_SMALL;
        if (code) {
            if (init.empty()) init.assign(r->n, 0);
            init[i] = code;
            bad++;
        }
    }
    if (bad) {
        // image of the whole control block: zero head + the preset status column
        if (!r->status_init) {
            if (tmalloc(ctx, &r->status_init, r->ctl_bytes) != hipSuccess) return ZNIPPY_E_NOMEM;
            HIPCHK(ctx, hipMemset(r->status_init, 0, r->ctl_bytes));
        }
        HIPCHK(ctx, hipMemcpy(r->status_init + znippy_rows::CTL_HEAD, init.data(), 4 * (size_t)r->n, hipMemcpyHostToDevice));
    }
    r->n_bad = bad;
    r->val_base = blob_base; r->val_bcap = r->blob_cap; r->val_ocap = out_cap;
    r->val_done = true;
    return ZNIPPY_OK;
}

// what a finished run's mirror says about the batch path's work: rows the fused kernel handed over ([0]) + block
// candidates left flagged ([5]) + the host's own list of big single-block rows
static void rows_note_hint(znippy_rows *r, unsigned slot) {
    const uint32_t *pc = reinterpret_cast<const uint32_t *>(r->h_counters + 16 * slot + 8);
    r->bx_hint = (pc[0] || pc[1] || pc[5] || r->n_list_a) ? 1 : 0;  // ([1]: what went to the serial decoder)
    r->lean_hint = (pc[0] || pc[1] || pc[3] || pc[5]) ? 0 : 1;       // ([3]: tiles the role-split kernel left on its list)
    r->lean_hint2 = (pc[0] || pc[1] || pc[2] || pc[5]) ? 0 : 1;      // ([2]: block items the fused block kernel left)
    if (r->n_small_tiles && pc[3] >= r->n_small_tiles) r->roles_off = true;  // the role-split kernel took not one tile: not this table's kernel
    // ... and when the fused kernels handed over every row of a table of small compressed rows, the next runs give the rows to the
    // batch path themselves (100k rows of real text: 0.26 ms of parsing each frame only to pass it on)
    if (r->small_ok && pc[0] >= r->n) r->small_off = true;
    if (r->small_off) r->bx_hint = 1;
}

int znippy_decode_verify_rows_async(znippy_ctx *ctx, znippy_rows *r, const void *d_blobs,
                                    uint64_t blob_base, void *d_out, uint64_t out_cap) {
    if (ctx && ctx->closing) return ZNIPPY_E_INVAL;  // destroyed context kept alive by its tables
    if (!ctx || !r || r->ctx != ctx) return ZNIPPY_E_INVAL;
    if (r->n && (!d_blobs || !d_out)) return ZNIPPY_E_INVAL;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    hipStream_t s = ctx->stream;
    ctx->n_ktimes = 0;
    { const int rc0 = ensure_decoder(ctx); if (rc0) return rc0; }
    { const int rc0 = rows_validate(ctx, r, blob_base, out_cap); if (rc0) return rc0; }
    if (r->fz_total) { const int rc0 = ensure_fz_pools(ctx, r->fz_bytes, r->fz_total); if (rc0) return rc0; }
    if (r->bx_slots) {
        int rc0 = ensure_fz_pools(ctx, r->bx_bytes, r->bx_item_cap);
        if (!rc0) rc0 = ensure_bx_pools(ctx, r->bx_bytes, r->bx_item_cap);
        if (!rc0 && r->rx_base && r->bx_hint != 0) rc0 = ensure_rx_pool(ctx, r->rx_words);
        if (rc0) return rc0;
    }
    if (r->run_seq && r->bx_hint < 0 && r->n) {  // a run of this table has finished meanwhile?
        const unsigned slot = (unsigned)((r->run_seq - 1) & 1);
        if (hipEventQuery(r->ev_done[slot]) == hipSuccess) rows_note_hint(r, slot);
        else (void)hipGetLastError();
    }
    const bool bx = r->bx_slots && ctx->fz_lit_pool && ctx->fz_seq_pool && ctx->bx_fse_pool && ctx->bx_huf_pool && r->bx_hint != 0;
    const bool small_off = bx && r->small_off && !r->n_bad && !r->force_full && !ctx->sw.dbg;
    if (small_off && !r->all_rows) {
        if (tmalloc(ctx, &r->all_rows, 4 * (size_t)r->n) != hipSuccess) return ZNIPPY_E_NOMEM;
        hipLaunchKernelGGL(k_iota32, dim3((r->n + 255) / 256), dim3(256), 0, s, r->all_rows, r->n);
    }
    const int preset = r->n_bad ? 1 : 0;
    { auto &ra = r->run_args[r->run_seq & 1]; ra.blobs = d_blobs; ra.base = blob_base; ra.out = d_out; ra.cap = out_cap; }
    bool lean = false, lean_blocks = false, lean_mixed = false;
    // counters, hand-over counts, work cursors and the status column: one stream operation
    if (preset) HIPCHK(ctx, hipMemcpyAsync(r->ctl, r->status_init, r->ctl_bytes, hipMemcpyDeviceToDevice, s));
    else HIPCHK(ctx, hipMemsetAsync(r->ctl, 0, r->ctl_bytes, s));
    if (!r->n) { r->run_seq++; return ZNIPPY_OK; }
    // (the second hash pass looks at small tiles only when rows were handed over: word 0 of the hand-over counts)
    if (small_off) HIPCHK(ctx, hipMemsetD32Async((hipDeviceptr_t)(r->ctl + 64), (int)r->n, 1, s));
    // 1) fused small-row kernel: decode simple frames + hash (+ copy stored rows), one wave per tile
    HashArgs h{};
    h.tiles = r->plan.tiles; h.n_tiles = r->plan.n_tiles;
    h.len = r->usize;
    h.srcA = (const uint8_t *)d_blobs; h.offA = r->blob_off; h.baseA = blob_base;
    h.srcB = (uint8_t *)d_out; h.offB = r->out_off;
    h.sel = r->compressed; h.status = r->status; h.pending_count = r->pending_count;
    h.copy_to_B = 1;
    h.store_tiles = ctx->sw.store_g;
    h.misaligned_dst = r->odd_out || ((uintptr_t)d_out & 15) != 0;
    h.digests = r->digests; h.tile_cv = r->plan.tile_cv;
    // A table without a single compressed row (a repository of small files the reference stores as they are — png, jpg, gz —,
    // or one big jar): nothing to recognise, nothing to decode — one pass of the store path kernel over ALL tiles (hash + copy,
    // small tiles 64 bytes per leaf per step through the stage), the merge, the verify.  (The fused small-row kernel copies a
    // stored row with each lane's own 64-byte stores: 100k x 10 KiB stored rows 0.85 ms there, 0.63 here.)
    const bool stored_only = r->n_compressed == 0 && !preset && !ctx->sw.dbg && !ctx->sw.no_stored_only && !r->force_full;
    if (!small_off && !stored_only) {
        FusedArgs f{};
        f.h = h;
        f.h.pass = 1;  // PASS_FUSED
        f.blob_size = r->blob_size; f.out_cap = out_cap; f.status = r->status;
        f.preset = preset;
        f.pending = r->pending; f.pending_count = r->pending_count;
        f.dbg = ctx->sw.dbg;
        if (f.dbg & 8) {  // diagnostic: print the previous launch's phase stamps, then reset them
            static unsigned long long *dbg = nullptr;
            if (!dbg) { (void)hipMalloc(&dbg, 64); (void)hipMemset(dbg, 0, 64); }
            unsigned long long h4[8];
            (void)hipStreamSynchronize(s);
            (void)hipMemcpy(h4, dbg, 64, hipMemcpyDeviceToHost);
            if (h4[3]) fprintf(stderr, "[znippy dbg] waves=%llu prologue=%.0f decode=%.0f hash=%.0f | parse+lits=%.0f expand-build=%.0f stream-out=%.0f after-match=%.0f cycles/wave\n", h4[3],
                               (double)h4[0] / h4[3], (double)h4[1] / h4[3], (double)h4[2] / h4[3], (double)h4[6] / h4[3], (double)h4[4] / h4[3], (double)h4[5] / h4[3], (double)h4[7] / h4[3]);
            set_fused_dbg(dbg);
            (void)hipMemset(dbg, 0, 64);
            f.dbg_buf = dbg;
        }
        if (f.dbg & 32768) {
            if (!ctx->clk_buf) { (void)hipMalloc(&ctx->clk_buf, 64); (void)hipMemset(ctx->clk_buf, 0, 64); }
            f.dbg_buf = ctx->clk_buf;
        }
        if (f.dbg & (16 | 32 | 64)) set_fused_abl(f.dbg);
        f.lds_pad = ctx->sw.lds_pad;
        // Tables with enough small tiles go to the role-split persistent kernel first (loader + hasher waves: tiles whose
        // rows are all whole-leaf rows of the recognised periodic shape); what it leaves on its list — and small
        // tables, where a persistent grid only adds start-up latency — is k_fused_small's.
        const bool roles = !ctx->sw.no_roles && ctx->sw.roles_min != 0 && r->n_small_tiles >= ctx->sw.roles_min && r->n_small_tiles > 0 &&
                           !(f.dbg & (1 | 2 | 4 | 8 | 128)) && !r->roles_off;
        // A table of small rows AND big stored / hashed units (BASELINE configs[4]: 3,500 small files beside 6 GB of jars) whose
        // last run handed nothing over: the small rows' kernel runs on the auxiliary stream beside the second hash pass, which
        // then has only the big units' slices to do — nothing of the one depends on the other (C5: 0.28 ms of a 3.45 ms
        // step ran in front of the pass).  k_verify checks the hand-over lists as in a lean run.
        lean_mixed = r->lean_mixed_ok && r->lean_hint == 1 && r->bx_hint == 0 && !preset && !r->force_full && !f.dbg && !ctx->sw.ddbg;
        const hipStream_t fs = lean_mixed ? ctx->aux : s;
        if (lean_mixed) {
            HIPCHK(ctx, hipEventRecord(ctx->ev_fork, s));
            HIPCHK(ctx, hipStreamWaitEvent(ctx->aux, ctx->ev_fork, 0));
        }
        if (roles) {
            f.cursor = r->cursor + 2;
            f.tile_list = r->slow_list;
            f.tile_count = r->pending_count + 3;
            ktime_begin(ctx, "decode_verify_roles", fs);
            launch_fused_roles(f, ctx->cus, fs);
            ktime_end(ctx, fs);
            lean = r->lean_ok && r->lean_hint == 1 && r->bx_hint == 0 && !preset && !r->force_full && !f.dbg;
        }
        if (r->n_small_tiles && !lean) {
            ktime_begin(ctx, "decode_verify_fused", fs);
            launch_fused_small(f, fs, roles ? ctx->cus * 5 : 0);
            ktime_end(ctx, fs);
        }
        if (lean_mixed) HIPCHK(ctx, hipEventRecord(ctx->ev_join, ctx->aux));
    }
    // 2) the two decode paths run side by side: block items (frames of >= 2 blocks, every block a work item) on the
    //    auxiliary stream, the general decoder (single-block big rows + whatever the fused kernel handed over) on the
    //    main one — each is latency-bound on its own and leaves most of the chip idle.  Frames the block path gives up
    //    on are decoded by a second general launch afterwards.
    if (!lean && !lean_mixed && !stored_only) {  // (a lean run: nothing is expected behind the fused kernels; k_verify checks that — rows_settle)
    BlockScanArgs b{};
    // The block items run on the auxiliary stream beside whatever the main stream has — unless it has nothing: a table of big
    // rows only whose last run handed nothing over (`behind` below: the serial decoder is launched after the join anyway).
    // Then everything goes down one stream: the fork and the join between two streams were ~0.1 ms of C3's 1.09 ms step.
    const bool one_stream = r->n_cand && r->bx_hint == 0 && r->n_small_tiles == 0;
    const hipStream_t ba = one_stream ? s : ctx->aux;
    // ... and when its last run needed neither the serial block decoder nor the serial decoder behind it (every block item
    // was written and hashed by the fused block kernel), those three launches are left out, the way a lean run of a table
    // of small rows leaves out what stands behind the roles kernel: k_verify looks at the lists, a flagged run is repeated.
    lean_blocks = one_stream && r->lean_blocks_ok && r->lean_hint2 == 1 && !preset && !r->force_full && r->n_bt && !ctx->sw.ddbg && !r->fz_total;
    if (r->n_cand) {
        if (!ctx->lit_scratch_b && hipMalloc(&ctx->lit_scratch_b, decode_lit_scratch_bytes(ctx->decode_grid)) != hipSuccess) return ZNIPPY_E_NOMEM;
        b.cand_row = r->cand_row; b.cand_base = r->cand_base; b.cand_nblocks = r->cand_nblocks; b.n_cand = r->n_cand;
        b.blobs = (const uint8_t *)d_blobs; b.blob_base = blob_base;
        b.blob_off = r->blob_off; b.blob_size = r->blob_size; b.usize = r->usize; b.out_off = r->out_off; b.out_cap = out_cap;
        b.item_src = r->item_src; b.row_flag = r->row_flag; b.status = r->status;
        b.preset = preset;
        b.pending = r->pending2; b.pending_count = r->pending_count + 1;  // its own hand-over list (count: second word of the control block)
        if (!one_stream) {
            HIPCHK(ctx, hipEventRecord(ctx->ev_fork, s));
            HIPCHK(ctx, hipStreamWaitEvent(ba, ctx->ev_fork, 0));
        }
        ktime_begin(ctx, "zstd_block_scan", ba);
        launch_scan_blocks(b, ba);
        ktime_end(ctx, ba);
        if (r->n_bt) {  // blocks of the common shape: written and hashed in one go, skipped by the two passes below
            HIPCHK(ctx, hipMemsetAsync(r->tile_done, 0, r->done_bytes, ba));
            FusedBlocksArgs fb{};
            fb.h = h;
            fb.h.pass = 0;  // PASS_ALL
            fb.blob_size = r->blob_size;
            fb.bt_tile = r->bt_tile; fb.bt_item = r->bt_item; fb.n_bt = r->n_bt;
            fb.item_src = r->item_src; fb.row_flag = r->row_flag;
            fb.tile_done = r->tile_done; fb.item_done = r->item_done;
            fb.dbg = ctx->sw.dbg;
            ktime_begin(ctx, "decode_verify_fused_blocks", ba);
            launch_fused_blocks(fb, ba);
            ktime_end(ctx, ba);
            launch_compact_items(r->item_done, r->n_items, r->todo, r->pending_count + 2, ba);
        }
        DecodeArgs a{};
        a.preset = preset;
        a.block_mode = 1;
        a.item_row = r->item_row; a.item_k = r->item_k; a.item_src = r->item_src; a.n_items = r->n_items; a.row_flag = r->row_flag;
        a.item_done = r->n_bt ? r->item_done : nullptr;
        a.todo = r->n_bt ? r->todo : nullptr; a.n_todo = r->pending_count + 2;
        a.pending_count = r->pending_count;
        a.blobs = (const uint8_t *)d_blobs;
        a.blob_base = blob_base;
        a.blob_off = r->blob_off; a.blob_size = r->blob_size; a.usize = r->usize; a.out_off = r->out_off;
        a.compressed = r->compressed;
        a.out = (uint8_t *)d_out; a.out_cap = out_cap;
        a.status = r->status; a.n_rows = r->n; a.cursor = r->cursor + 4;
        a.lit_scratch = ctx->lit_scratch_b;
        if (ctx->sw.ddbg) {  // diagnostic: phase shares of the previous block-item launch
            static unsigned long long *dbg = nullptr;
            if (!dbg) { (void)hipMalloc(&dbg, 64); (void)hipMemset(dbg, 0, 64); }
            unsigned long long h[8];
            (void)hipStreamSynchronize(s);
            (void)hipStreamSynchronize(ba);
            (void)hipMemcpy(h, dbg, 64, hipMemcpyDeviceToHost);
            if (h[0]) fprintf(stderr, "[znippy ddbg] block items=%llu  cycles per item: literals=%.0f seq-tables=%.0f seq-decode=%.0f execute=%.0f tail=%.0f\n", h[0],
                              (double)h[1] / h[0], (double)h[2] / h[0], (double)h[3] / h[0], (double)h[4] / h[0], (double)h[5] / h[0]);
            (void)hipMemset(dbg, 0, 64);
            a.dbg = dbg;
        }
        if (!lean_blocks) {
            ktime_begin(ctx, "zstd_decode_blocks", ba);
            launch_decode(a, std::min<int>(ctx->decode_grid, (int)r->n_items), false, ba);
            ktime_end(ctx, ba);
        }
        if (r->fz_total) {
            // Foreign frames (what the block items gave up on): entropy-decode every block at once, then execute frame by
            // frame.  On the auxiliary stream, behind the block items and BESIDE the general decoder on the main stream:
            // both are a few long-lived waves per frame, neither fills the chip (real text, libzstd -19 frames: the two
            // used to run back to back, 7.0 + 10.6 ms).
            FzArgs z{};
            z.cand_row = r->cand_row; z.cand_fzbase = r->fz_base; z.cand_fzcap = r->fz_cap; z.n_cand = r->n_cand;
            z.it_cand = r->fz_it_cand; z.total_items = r->fz_total; z.cand_nb = r->fz_nb; z.items = r->fz_items;
            z.blobs = (const uint8_t *)d_blobs; z.blob_base = blob_base;
            z.blob_off = r->blob_off; z.blob_size = r->blob_size; z.usize = r->usize; z.out_off = r->out_off; z.out_cap = out_cap;
            z.out = (uint8_t *)d_out;
            z.row_flag = r->row_flag; z.status = r->status; z.preset = preset;
            z.lit_pool = ctx->fz_lit_pool; z.lit_cap = ctx->fz_lit_cap; z.seq_pool = ctx->fz_seq_pool; z.seq_cap = ctx->fz_seq_cap;
            z.pool_used = reinterpret_cast<unsigned long long *>(r->ctl + 192);  // [lit bytes, seq records], zeroed with the control block
            z.cursor = r->cursor + 12;
            if (ctx->sw.ddbg) {  // diagnostic: where the previous run's execute kernel spent its cycles
                static unsigned long long *dbg = nullptr;
                if (!dbg) { (void)hipMalloc(&dbg, 256); (void)hipMemset(dbg, 0, 256); }
                unsigned long long h[32];
                (void)hipStreamSynchronize(s);
                (void)hipStreamSynchronize(ba);
                (void)hipMemcpy(h, dbg, 256, hipMemcpyDeviceToHost);
                if (h[0]) fprintf(stderr, "[znippy ddbg] fz exec: frames=%llu groups=%llu seqs=%llu big=%llu rounds=%llu flushes=%llu histreads=%llu rep_groups=%llu | kcycles/frame: total=%.0f records=%.0f rep+scan=%.0f big=%.0f flush=%.0f histread=%.0f lits=%.0f matches=%.0f tail=%.0f\n",
                                  h[0], h[1], h[2], h[3], h[4], h[5], h[6], h[7], h[8] / 1e3 / h[0], h[9] / 1e3 / h[0], h[10] / 1e3 / h[0], h[11] / 1e3 / h[0],
                                  h[12] / 1e3 / h[0], h[13] / 1e3 / h[0], h[14] / 1e3 / h[0], h[15] / 1e3 / h[0], h[16] / 1e3 / h[0]);
                if (h[20]) fprintf(stderr, "[znippy ddbg] fz entropy: blocks=%llu seqs=%llu | kcycles/block: literal tree=%.0f literal table+streams=%.0f sequence tables=%.0f sequence decode=%.0f\n",
                                   h[20], h[25], h[21] / 1e3 / h[20], h[22] / 1e3 / h[20], h[23] / 1e3 / h[20], h[24] / 1e3 / h[20]);
                (void)hipMemset(dbg, 0, 256);
                z.dbg = dbg;
            }
            launch_fz_scan(z, r->fz_work, r->cursor + 13, ba);
            ktime_begin(ctx, "zstd_foreign_entropy", ba);
            launch_fz_entropy(z, ctx->cus, r->fz_work, r->cursor + 13, ba);
            ktime_end(ctx, ba);
            ktime_begin(ctx, "zstd_foreign_execute", ba);
            launch_fz_exec(z, ba);
            ktime_end(ctx, ba);
        }
        if (!one_stream) HIPCHK(ctx, hipEventRecord(ctx->ev_join, ctx->aux));
    }
    if (r->n_compressed) {
        DecodeArgs a{};
        a.preset = preset;
        a.list_a = r->list_a; a.n_list_a = r->n_list_a;
        a.pending = r->pending; a.pending_count = r->pending_count;
        a.blobs = (const uint8_t *)d_blobs;
        a.blob_base = blob_base;
        a.blob_off = r->blob_off; a.blob_size = r->blob_size; a.usize = r->usize; a.out_off = r->out_off;
        a.compressed = r->compressed;
        a.out = (uint8_t *)d_out; a.out_cap = out_cap;
        a.status = r->status; a.n_rows = r->n; a.cursor = r->cursor;
        a.lit_scratch = ctx->lit_scratch;
        if (ctx->sw.ddbg) {  // diagnostic: phase shares of the previous general-decoder launch
            static unsigned long long *dbg = nullptr;
            if (!dbg) { (void)hipMalloc(&dbg, 64); (void)hipMemset(dbg, 0, 64); }
            unsigned long long h[8];
            (void)hipStreamSynchronize(s);
            (void)hipMemcpy(h, dbg, 64, hipMemcpyDeviceToHost);
            if (h[0]) fprintf(stderr, "[znippy ddbg] general decoder frames=%llu  kcycles per frame: headers+tree=%.1f huffman-table=%.1f literal-streams=%.1f seq-tables=%.1f first-batch=%.1f decode+execute=%.1f tail=%.1f\n", h[0],
                              h[6] / 1e3 / h[0], h[7] / 1e3 / h[0], h[1] / 1e3 / h[0], h[2] / 1e3 / h[0], h[3] / 1e3 / h[0], h[4] / 1e3 / h[0], h[5] / 1e3 / h[0]);
            (void)hipMemset(dbg, 0, 64);
            a.dbg = dbg;
        }
        if (bx) {
            // Batch path: every frame that is still undecoded — big single-block rows, what the fused kernel handed over,
            // block candidates the block-item path flagged — goes through the lane-per-block kernels in ONE pass (the
            // more blocks, the fuller their waves), behind the block items; the serial decoder takes what they leave.
            if (r->n_cand) {
                HIPCHK(ctx, hipStreamWaitEvent(s, ctx->ev_join, 0));
                launch_finish_blocks(b, s, true);  // unflagged candidates are done; flagged ones wait for the batch path's verdict
            }
            BxArgs x{};
            x.list_a = r->list_a; x.n_list_a = r->n_list_a;
            x.pending = r->pending; x.pending_count = r->pending_count;
            if (small_off) {  // every row is the batch path's: its list is all rows, nothing was handed over (a word that stays zero)
                x.list_a = r->all_rows; x.n_list_a = r->n;
                x.pending_count = reinterpret_cast<uint32_t *>(r->ctl + 448) + 15;
            }
            x.bc_row = r->cand_row; x.n_bc = r->n_cand;
            x.blobs = (const uint8_t *)d_blobs; x.blob_base = blob_base;
            x.blob_off = r->blob_off; x.blob_size = r->blob_size; x.usize = r->usize; x.out_off = r->out_off; x.out_cap = out_cap;
            x.out = (uint8_t *)d_out;
            x.status = r->status; x.preset = preset; x.row_flag = r->row_flag;
            x.cand_row = r->bx_cand_row; x.cand_base = r->bx_cand_base; x.cand_nb = r->bx_cand_nb; x.slot_cap = r->bx_slots;
            x.items = r->bx_items; x.prep = r->bx_prep; x.item_cap = r->bx_item_cap;
            x.ctr = reinterpret_cast<uint32_t *>(r->ctl + 384);
            x.huf_list = r->bx_huf_list; x.seq_list = r->bx_seq_list; x.sort_tmp = r->bx_sort_tmp;
            x.lit_pool = ctx->fz_lit_pool; x.lit_cap = ctx->fz_lit_cap; x.seq_pool = ctx->fz_seq_pool; x.seq_cap = ctx->fz_seq_cap;
            x.fse_pool = ctx->bx_fse_pool; x.fse_cap = ctx->bx_fse_cap; x.huf_pool = ctx->bx_huf_pool; x.huf_cap = ctx->bx_huf_cap;
            x.pool_used = reinterpret_cast<unsigned long long *>(r->ctl + 256);
            x.pending2 = r->pending2; x.pending2_count = r->pending_count + 1;
            // 0: the blocks that get a wave of their own are picked from the table's histogram (k_bx_split: one workgroup per
            // list, worth its ~0.1-0.4 ms where the chip is not full of blocks anyway); big tables keep the fixed threshold
            x.big_seq = ctx->sw.bx_big_set || r->bx_nblk > 32768 ? ctx->sw.bx_big : 0u;
            const bool rx = r->rx_base && ctx->rx_pool && r->rx_words;
            if (rx) {
                x.rx_ptr = ctx->rx_pool; x.rx_cap = ctx->rx_cap; x.rx_chunk = ctx->rx_chunk; x.rx_cdone = ctx->rx_cdone;
                x.rx_base = r->rx_base; x.rx_fail = r->rx_fail; x.rx_blk = r->rx_blk; x.rx_list = r->rx_list;
                x.rx_pending = reinterpret_cast<uint32_t *>(r->ctl + 448);
                x.rx_bound = std::min<uint64_t>(r->rx_words, ctx->rx_cap);
                x.rx_min = r->rx_min;
            }
            x.small_frames = r->n_compressed && r->bx_bytes / r->n_compressed <= 65536;
            if (ctx->sw.ddbg) {  // diagnostic: where the previous run's table kernel spent its waves' time
                static unsigned long long *dbg = nullptr;
                if (!dbg) { (void)hipMalloc(&dbg, 1024); (void)hipMemset(dbg, 0, 1024); }
                unsigned long long h[128];
                (void)hipStreamSynchronize(s);
                (void)hipMemcpy(h, dbg, 1024, hipMemcpyDeviceToHost);
                if (h[64]) {
                    const unsigned long long *e = h + 64;
                    fprintf(stderr, "[znippy ddbg] batch execute: frames=%llu groups=%llu seqs=%llu big=%llu rounds=%llu flushes=%llu histreads=%llu | kcycles/frame: total=%.1f records=%.1f rep+scan=%.1f big=%.1f flush=%.1f histread=%.1f lits=%.1f matches=%.1f tail=%.1f\n",
                            e[0], e[1], e[2], e[3], e[4], e[5], e[6], e[8] / 1e3 / e[0], e[9] / 1e3 / e[0], e[10] / 1e3 / e[0], e[11] / 1e3 / e[0], e[12] / 1e3 / e[0], e[13] / 1e3 / e[0],
                            e[14] / 1e3 / e[0], e[15] / 1e3 / e[0], e[16] / 1e3 / e[0]);
                }
                if (h[32]) fprintf(stderr, "[znippy ddbg] batch tables: wave passes=%llu  kcycles per pass: literals header + weights=%.1f sequences header=%.1f huffman table=%.1f sequence tables=%.1f\n", h[32],
                                   h[33] / 1e3 / h[32], h[34] / 1e3 / h[32], h[35] / 1e3 / h[32], h[36] / 1e3 / h[32]);
                (void)hipMemset(dbg, 0, 1024);
                x.dbg = dbg;
            }
            static const char *const bx_names[10] = {"zstd_batch_scan", "zstd_batch_tables", "zstd_batch_huffman", "zstd_batch_sequences", "zstd_batch_execute", "zstd_batch_finish",
                                                     "zstd_batch_sequences_long", "zstd_batch_sort", "zstd_resolve_plan", "zstd_resolve_expand"};
            auto stage = [&](int st, hipStream_t on) {
                ktime_begin(ctx, bx_names[st], on);
                launch_bx_stage(x, ctx->cus, st, on);
                ktime_end(ctx, on);
            };
            stage(0, s);
            stage(1, s);
            stage(7, s);
            // the long chains (blocks of >= BX_BIG_SEQ sequences, a wave each) run on the auxiliary stream beside the Huffman
            // streams and the lane-per-block sequence kernel: each is a few hundred long-lived waves at most
            HIPCHK(ctx, hipEventRecord(ctx->ev_fork, s));
            HIPCHK(ctx, hipStreamWaitEvent(ctx->aux, ctx->ev_fork, 0));
            stage(6, ctx->aux);
            HIPCHK(ctx, hipEventRecord(ctx->ev_join, ctx->aux));
            // ... and the lane-per-block sequence kernel beside the Huffman streams on a third stream (the write side's copy
            // stream, idle here): the two touch different pools and different fields of a block's record.  A small table is
            // its longest chains: the image's source text had Huffman 2.5 + sequences 1.8 ms in a row beside 3.5 ms of long chains.
            // (Only where the chip is not full of blocks anyway: 100k blocks, both kernels chip-wide: 2.98 ms together against
            // 1.85 + 0.96 one after the other.)
            const bool third = r->bx_nblk <= 32768;
            if (third) {
                HIPCHK(ctx, hipStreamWaitEvent(ctx->copy, ctx->ev_fork, 0));
                stage(3, ctx->copy);
                HIPCHK(ctx, hipEventRecord(ctx->ev_join2, ctx->copy));
            }
            stage(2, s);
            if (!third) stage(3, s);
            HIPCHK(ctx, hipStreamWaitEvent(s, ctx->ev_join, 0));
            if (third) HIPCHK(ctx, hipStreamWaitEvent(s, ctx->ev_join2, 0));
            if (rx) {  // big frames: resolved in parallel (every byte a word, pointer jumping) instead of executed by a wave each
                stage(8, s);
                if (ctx->sw.trace) {
                    uint32_t g[16];
                    unsigned long long pu[16];
                    (void)hipMemcpy(g, r->ctl + 384, 64, hipMemcpyDeviceToHost);
                    (void)hipMemcpy(pu, r->ctl + 256, 128, hipMemcpyDeviceToHost);
                    fprintf(stderr, "[znippy trace] resolve plan: slots %u items %u list %u frames %u words %llu extent %llu cap %llu item_cap %u\n", g[0], g[1], g[9], g[11], pu[10], pu[11],
                            (unsigned long long)ctx->rx_cap, r->bx_item_cap);
                }
                // the frames the plan did not take are executed by a wave each, beside the resolve stages (they share nothing)
                HIPCHK(ctx, hipEventRecord(ctx->ev_fork, s));
                HIPCHK(ctx, hipStreamWaitEvent(ctx->aux, ctx->ev_fork, 0));
                stage(4, ctx->aux);
                HIPCHK(ctx, hipEventRecord(ctx->ev_join, ctx->aux));
                stage(9, s);
                static const char *const jump_names[12] = {"zstd_resolve_jump_0", "zstd_resolve_jump_1", "zstd_resolve_jump_2", "zstd_resolve_jump_3", "zstd_resolve_jump_4", "zstd_resolve_jump_5",
                                                           "zstd_resolve_jump_6", "zstd_resolve_jump_7", "zstd_resolve_jump_8", "zstd_resolve_jump_9", "zstd_resolve_jump_10", "zstd_resolve_jump_11"};
                static_assert(zn::RX_ROUNDS <= 12, "names");
                if (!ctx->sw.ddbg) ktime_begin(ctx, "zstd_resolve_jump", s);
                for (int rd = 0; rd < (int)zn::RX_ROUNDS; rd++) {
                    if (ctx->sw.ddbg) ktime_begin(ctx, jump_names[rd], s);  // diagnostic: every round by itself
                    launch_bx_stage(x, ctx->cus, 10 + rd, s);
                    if (ctx->sw.ddbg) ktime_end(ctx, s);
                }
                if (!ctx->sw.ddbg) ktime_end(ctx, s);
                ktime_begin(ctx, "zstd_resolve_store", s);
                launch_bx_stage(x, ctx->cus, 30, s);
                ktime_end(ctx, s);
                HIPCHK(ctx, hipStreamWaitEvent(s, ctx->ev_join, 0));
            } else stage(4, s);
            stage(5, s);
            a.list_a = nullptr; a.n_list_a = 0;
            a.pending = r->pending2; a.pending_count = r->pending_count + 1;
            a.cursor = r->cursor + 8;
            ktime_begin(ctx, "zstd_decode_fallback");
            if (!ctx->sw.fz_only)
                launch_decode(a, std::min<int>(ctx->decode_grid, (int)r->n_compressed), r->wide_rows, s);
            ktime_end(ctx);
        } else {
            // Round-2 flow (no batch path: ZNIPPY_NO_BX, no pools, or a table whose last run handed nothing over): the
            // serial decoder takes the host's list and what the fused kernel hands over, beside the block items on the
            // auxiliary stream — or behind them when nothing is expected (its launch then returns at once instead of
            // waiting for CUs next to the block kernels: C3's 0.25 ms "general decoder" that only waited).
            const bool behind = r->n_cand && r->bx_hint == 0;
            if (behind) {
                if (!one_stream) HIPCHK(ctx, hipStreamWaitEvent(s, ctx->ev_join, 0));
                launch_finish_blocks(b, s, false);
            }
            if (!lean_blocks) {
            ktime_begin(ctx, "zstd_decode_general");
            // A full grid of this kernel (4 workgroups per CU at 128 VGPRs) is the whole register file: whatever the auxiliary
            // stream launches then waits until workgroups run out of rows.  With candidates for the block / foreign-frame
            // paths it leaves them a quarter.
            const int gen_grid = r->n_cand && !behind ? ctx->decode_grid / 4 * ctx->gen_share : ctx->decode_grid;
            launch_decode(a, std::min<int>(gen_grid, (int)r->n_compressed), r->wide_rows, s);
            ktime_end(ctx);
            }
            if (r->n_cand && !lean_blocks) {  // join (block items and the foreign-frame path on the auxiliary stream), then what both gave up on
                if (!behind) {
                    HIPCHK(ctx, hipStreamWaitEvent(s, ctx->ev_join, 0));
                    launch_finish_blocks(b, s, false);
                }
                a.list_a = nullptr; a.n_list_a = 0;
                a.pending = r->pending2; a.pending_count = r->pending_count + 1;
                a.cursor = r->cursor + 8;
                ktime_begin(ctx, "zstd_decode_fallback");
                if (!ctx->sw.fz_only)
                    launch_decode(a, std::min<int>(ctx->decode_grid, (int)r->n_cand), r->wide_rows, s);
                ktime_end(ctx);
            }
        }
    }
    }
    if (!lean) {
    // 3) second hash pass: slices of big rows + rows the general decoder finished
    h.pass = stored_only ? 0 : 2;  // PASS_ALL : PASS_SECOND
    h.tile_done = r->n_bt && !stored_only ? r->tile_done : nullptr;
    ktime_begin(ctx, "blake3_second_pass");
    launch_hash_tiles(h, s);
    ktime_end(ctx);
    if (r->plan.n_big) {
        ktime_begin(ctx, "blake3_merge_big");
        launch_merge_big(r->plan.big, r->plan.n_big, r->plan.tile_cv, r->digests, r->plan.grp_big, r->plan.grp_k, r->plan.n_grp, r->plan.max_cvs, s);
        ktime_end(ctx);
    }
    }  // !lean
    if (lean_mixed) HIPCHK(ctx, hipStreamWaitEvent(s, ctx->ev_join, 0));
    r->last_lean = lean || lean_blocks || lean_mixed;
    ktime_begin(ctx, "verify");
    // the lists a lean run must have left empty: [0] rows handed over by the fused kernels, [1] rows for the serial decoder, [5]
    // flagged candidates; [3] tiles the roles kernel left (small rows), [2] block items the fused block kernel left (big rows)
    launch_verify(r->digests, r->checksum, r->usize, r->status, r->n, r->row_begin, r->counters, r->corrupt,
                  r->corrupt_cap, s, (lean || lean_blocks || lean_mixed) ? r->pending_count : nullptr, (lean || lean_mixed) ? 0x2Bu : 0x27u);
    ktime_end(ctx);
    {
        const unsigned slot = (unsigned)(r->run_seq & 1);
        HIPCHK(ctx, hipMemcpyAsync(r->h_counters + 16 * slot, r->counters, 128, hipMemcpyDeviceToHost, s));
        HIPCHK(ctx, hipEventRecord(r->ev_done[slot], s));
        r->run_seq++;
    }
    HIPCHK(ctx, hipGetLastError());
    return ZNIPPY_OK;
}

// A lean run whose lists were not empty after all (counters[7] set by k_verify): the run of mirror slot `slot` is done again
// in full with the arguments IT was given (a caller that alternates buffers gets the right buffer completed), and the slot
// gets the full run's counters.  The repeat is run k + 2 when another run is queued behind the flagged one — the same slot
// — or run k + 1 otherwise; a flagged run k + 1 is settled when its own results are read.  Called by everything that hands
// a run's results to the caller.
static int rows_settle(znippy_ctx *ctx, znippy_rows *r, unsigned slot) {
    if (!r->n || !r->run_seq || !(r->h_counters[16 * slot + 7])) return ZNIPPY_OK;
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    r->force_full = true;
    r->lean_hint = 0; r->lean_hint2 = 0;
    const znippy_rows::RunArgs ra = r->run_args[slot];
    const int rc = znippy_decode_verify_rows_async(ctx, r, ra.blobs, ra.base, ra.out, ra.cap);
    if (rc) return rc;
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    const unsigned now = (unsigned)((r->run_seq - 1) & 1);
    if (now != slot) {
        memcpy(r->h_counters + 16 * slot, r->h_counters + 16 * now, 128);
        r->run_args[slot] = ra;
    }
    return ZNIPPY_OK;
}

// Counters of the run `lag` runs before the latest one (lag 0 or 1): waits for THAT run only, so a caller that
// keeps two runs in flight reads run k's counters while run k + 1 executes (the read loop reports after the loop,
// not per row: decompress.rs:L195-221).
int znippy_rows_results_lagged(znippy_ctx *ctx, znippy_rows *r, unsigned lag, znippy_verify_counters *counters) {
    if (ctx && ctx->closing) return ZNIPPY_E_INVAL;  // destroyed context kept alive by its tables
    if (!ctx || !r || r->ctx != ctx || !counters || lag > 1 || r->run_seq <= lag) return ZNIPPY_E_INVAL;
    uint64_t c[8] = {0};
    if (r->n) {
        const unsigned slot = (unsigned)((r->run_seq - 1 - lag) & 1);
        HIPCHK(ctx, hipEventSynchronize(r->ev_done[slot]));
        { const int rc = rows_settle(ctx, r, slot); if (rc) return rc; }
        memcpy(c, r->h_counters + 16 * slot, 64);
        rows_note_hint(r, slot);
    }
    counters->total_chunks = c[0]; counters->total_written_bytes = c[1]; counters->verified_bytes = c[2];
    counters->corrupt_bytes = c[3]; counters->corrupt_rows = c[4]; counters->decode_errors = c[5];
    return ZNIPPY_OK;
}

int znippy_rows_results(znippy_ctx *ctx, znippy_rows *r, znippy_verify_counters *counters,
                        uint64_t *corrupt_rows, uint64_t corrupt_cap, int32_t *row_status) {
    if (ctx && ctx->closing) return ZNIPPY_E_INVAL;  // destroyed context kept alive by its tables
    if (!ctx || !r) return ZNIPPY_E_INVAL;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    uint64_t c[8] = {0};
    if (r->n && r->run_seq) {
        { const int rc = rows_settle(ctx, r, (unsigned)((r->run_seq - 1) & 1)); if (rc) return rc; }
        memcpy(c, r->h_counters + 16 * ((r->run_seq - 1) & 1), 64);  // copied by the run itself (pinned)
        rows_note_hint(r, (unsigned)((r->run_seq - 1) & 1));
    }
    if (counters) {
        counters->total_chunks = c[0]; counters->total_written_bytes = c[1]; counters->verified_bytes = c[2];
        counters->corrupt_bytes = c[3]; counters->corrupt_rows = c[4]; counters->decode_errors = c[5];
    }
    if (corrupt_rows && corrupt_cap && c[4]) {
        uint64_t k = std::min<uint64_t>(c[4], r->corrupt_cap);
        std::vector<uint64_t> tmp(k);
        HIPCHK(ctx, hipMemcpy(tmp.data(), r->corrupt, 8 * k, hipMemcpyDeviceToHost));
        std::sort(tmp.begin(), tmp.end());
        memcpy(corrupt_rows, tmp.data(), 8 * std::min<uint64_t>(k, corrupt_cap));
    }
    if (row_status && r->n) {
        HIPCHK(ctx, hipMemcpy(row_status, r->status, 4 * (size_t)r->n, hipMemcpyDeviceToHost));
        for (uint32_t i = 0; i < r->n; i++)
            if (row_status[i] > 0) row_status[i] = 0;  // internal routing states (1, 2) are successes
    }
    return ZNIPPY_OK;
}

int znippy_decode_verify_rows(znippy_ctx *ctx, znippy_rows *rows, const void *d_blobs,
                              uint64_t blob_base, void *d_out, uint64_t out_cap,
                              znippy_verify_counters *counters, uint64_t *corrupt_rows,
                              uint64_t corrupt_cap, int32_t *row_status) {
    if (ctx && ctx->closing) return ZNIPPY_E_INVAL;  // destroyed context kept alive by its tables
    int rc = znippy_decode_verify_rows_async(ctx, rows, d_blobs, blob_base, d_out, out_cap);
    if (rc) return rc;
    return znippy_rows_results(ctx, rows, counters, corrupt_rows, corrupt_cap, row_status);
}

int znippy_rows_digests(znippy_ctx *ctx, znippy_rows *r, uint8_t *digests) {
    if (ctx && ctx->closing) return ZNIPPY_E_INVAL;  // destroyed context kept alive by its tables
    if (!ctx || !r || !digests) return ZNIPPY_E_INVAL;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    if (r->n && r->run_seq) { const int rc = rows_settle(ctx, r, (unsigned)((r->run_seq - 1) & 1)); if (rc) return rc; }
    if (r->n) HIPCHK(ctx, hipMemcpy(digests, r->digests, 32 * (size_t)r->n, hipMemcpyDeviceToHost));
    return ZNIPPY_OK;
}

// ---- rounds -------------------------------------------------------------------------------------
static void rounds_select(znippy_rounds *r, unsigned slot) {  // the result slab the next run writes
    const size_t n = r->n;
    r->res = r->res_m[slot];
    r->h_res = r->h_res_m[slot];
    r->total = reinterpret_cast<uint64_t *>(r->res);
    r->overflow = reinterpret_cast<uint32_t *>(r->res + 8);
    r->blob_offset = reinterpret_cast<uint64_t *>(r->res + 16);
    r->blob_size = r->blob_offset + n;
    r->digests = reinterpret_cast<uint32_t *>(r->res + 16 + n * 16);
}

void znippy_rounds_destroy(znippy_rounds *r) {
    if (!r) return;
    (void)hipSetDevice(r->ctx->device);
    if (r->ctx->copy) (void)hipStreamSynchronize(r->ctx->copy);  // a result copy may still be reading a slab
    for (int k = 0; k < 2; k++) {
        pinned_give(r->ctx, r->h_res_m[k], r->h_res_cap_m[k]);
        event_give(r->ctx, r->ev_enc[k]);
        event_give(r->ctx, r->ev_res[k]);
    }
    void *ptrs[] = {r->src_off, r->len, r->skip, r->res_m[0], r->res_m[1], r->items, r->piece_len, r->piece_len_init,
                    r->piece_start, r->local_excl, r->block_tot, r->first_item, r->stored, r->order_small, r->order_wide, r->retry_list, r->retry_count,
                    r->plan_scratch[0], r->plan_scratch[1], r->plan_scratch[2]};
    pinned_give(r->ctx, r->h_stored, r->h_stored_cap);
    for (void *p : ptrs)
        tfree(r->ctx, p);
    free_plan(r->ctx, r->plan);
    znippy_ctx *const c = r->ctx;
    delete r;
    table_released(c);
}

int znippy_rounds_create(znippy_ctx *ctx, const uint64_t *src_offset, const uint64_t *len,
                         const uint8_t *skip, uint64_t n, znippy_rounds **out) {
    if (ctx && ctx->closing) return ZNIPPY_E_INVAL;  // destroyed context kept alive by its tables
    if (!ctx || !out || (n && (!src_offset || !len)) || n >= 0xFFFFFFF0ull) return ZNIPPY_E_INVAL;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    znippy_rounds *r = new znippy_rounds();
    r->ctx = ctx;
    ctx->live_tables++;
    r->n = (uint32_t)n;
    TDbg td(ctx->sw.tdbg, "rounds_create");
    if (skip) r->h_skip.assign(skip, skip + n);  // (results: which rounds are stored; empty = none)
    uint64_t tot[8];
    zn_rounds_totals(len, skip, n, tot);  // sizes of the device arrays; the arrays themselves are filled on the device
    r->blob_bound = tot[4]; r->in_bytes = tot[5]; r->enc_bytes = tot[6];
    r->all_stored_aligned = n > 0 && r->enc_bytes == 0 && !tot[7];
    if (tot[0] >= 0xFFFFFFF0ull) { znippy_rounds_destroy(r); return ZNIPPY_E_INVAL; }
    r->n_items = (uint32_t)tot[0]; r->prov_bytes = tot[1]; r->n_small = (uint32_t)tot[2]; r->n_wide = (uint32_t)tot[3];
    td.mark("totals");
    int rc;
    if ((rc = dev_upload(ctx, &r->src_off, src_offset, n)) || (rc = dev_upload(ctx, &r->len, len, n)) ||
        (skip ? (rc = dev_upload(ctx, &r->skip, skip, n)) : (tmalloc(ctx, &r->skip, std::max<size_t>(n, 16)) != hipSuccess ? (rc = ZNIPPY_E_NOMEM) : 0))) {
        znippy_rounds_destroy(r);
        return rc;
    }
    if (!skip && n) HIPCHK(ctx, hipMemsetAsync(r->skip, 0, n, ctx->stream));
    td.mark("columns_h2d");
    r->res_bytes = 16 + (size_t)n * (8 + 8 + 32);
    for (int k = 0; k < 2; k++)
        if (tmalloc(ctx, &r->res_m[k], r->res_bytes) != hipSuccess ||
            !(r->h_res_m[k] = (uint8_t *)pinned_take(ctx, r->res_bytes, &r->h_res_cap_m[k])) ||
            !(r->ev_enc[k] = event_take(ctx)) || !(r->ev_res[k] = event_take(ctx))) {
            znippy_rounds_destroy(r);
            return ZNIPPY_E_NOMEM;
        }
    rounds_select(r, 0);
    td.mark("slabs");
    PlanBuf p;
    build_plan([&](uint32_t u) { return len[u]; }, (uint32_t)n, p);
    if ((rc = upload_plan(ctx, p, r->plan))) {
        znippy_rounds_destroy(r);
        return rc;
    }
    if (r->n_items == n && r->n_wide == 0 && r->enc_bytes == r->in_bytes && n) {
        uint32_t max_units = 0;
        for (const Tile &t : p.tiles) max_units = std::max(max_units, t.n_units);
        r->fuse_tiles = 1;  // one tile per dequeue (C2: 6 rounds): 0.77 ms against 0.86 for two and 1.12 for ten on one box — with ~4 tiles per wave the finest grain balances best
        if (const char *e = getenv("ZNIPPY_FUSE_TILES")) { const int v = atoi(e); if (v >= 1 && v * (int)max_units <= 64) r->fuse_tiles = v; }  // A/B
    }
    td.mark("hash_plan");
    // encoder plan: one item per output piece, in round order — counted, scanned and filled on the device
    const size_t ni = std::max<size_t>(r->n_items, 1), nsb = (ni + 255) / 256;
    RoundSums *sums = nullptr;
    const uint32_t nblk = ((uint32_t)n + 1023) / 1024;
    if (tmalloc(ctx, &r->first_item, std::max<size_t>(4 * (size_t)n, 16)) != hipSuccess || tmalloc(ctx, &sums, std::max<size_t>(sizeof(RoundSums) * (size_t)nblk, 64)) != hipSuccess ||
        tmalloc(ctx, &r->order_small, std::max<size_t>(4 * (size_t)r->n_small, 16)) != hipSuccess ||
        tmalloc(ctx, &r->order_wide, std::max<size_t>(4 * (size_t)r->n_wide, 16)) != hipSuccess ||
        tmalloc(ctx, &r->retry_list, std::max<size_t>(4 * (size_t)r->n_small, 16)) != hipSuccess || tmalloc(ctx, &r->retry_count, 64) != hipSuccess ||
        tmalloc(ctx, &r->stored, std::max<size_t>(n, 16)) != hipSuccess ||
        !(r->h_stored = (uint8_t *)pinned_take(ctx, std::max<size_t>(n, 16), &r->h_stored_cap)) ||
        tmalloc(ctx, &r->items, sizeof(EncItem) * ni) != hipSuccess || tmalloc(ctx, &r->piece_len_init, 4 * ni) != hipSuccess ||
        tmalloc(ctx, &r->piece_len, 4 * ni) != hipSuccess || tmalloc(ctx, &r->piece_start, 8 * ni) != hipSuccess ||
        tmalloc(ctx, &r->local_excl, 8 * ni) != hipSuccess || tmalloc(ctx, &r->block_tot, 8 * nsb) != hipSuccess) {
        tfree(ctx, sums);
        znippy_rounds_destroy(r);
        return ZNIPPY_E_NOMEM;
    }
    if (n) {
        const uint8_t *const d_skip = skip ? r->skip : (const uint8_t *)nullptr;
        hipLaunchKernelGGL(k_rounds_sums, dim3(nblk), dim3(256), 0, ctx->stream, r->len, d_skip, (uint32_t)n, sums);
        hipLaunchKernelGGL(k_rounds_scan, dim3(1), dim3(256), 0, ctx->stream, sums, nblk);
        hipLaunchKernelGGL(k_rounds_fill, dim3(nblk), dim3(256), 0, ctx->stream, r->len, d_skip, (uint32_t)n, sums, r->first_item, r->items, r->piece_len_init,
                           r->order_small, r->order_wide);
    }
    r->plan_scratch[0] = sums;  // (parked until the table goes: the kernels above are still queued)
    td.mark("encoder_plan");
    // store-path pieces keep their fixed lengths for the table's lifetime; encoded pieces are rewritten by
    // the encoder on every run, so one copy at creation is enough
    if (r->n_items) HIPCHK(ctx, hipMemcpyAsync(r->piece_len, r->piece_len_init, 4 * (size_t)r->n_items, hipMemcpyDeviceToDevice, ctx->stream));
    td.mark("rest");
    *out = r;
    return ZNIPPY_OK;
}

uint64_t znippy_rounds_blob_bound(const znippy_rounds *r) { return r ? r->blob_bound : 0; }

// d_copy_out != nullptr: the stored (skip) rounds are copied to d_copy_out + blob_offset[round] while they are hashed
// (store-heavy tables; blob_offset must have been computed on the stream before)
static int hash_rounds_async(znippy_ctx *ctx, znippy_rounds *r, const void *d_src, hipStream_t on = nullptr,
                             void *d_copy_out = nullptr, uint64_t copy_cap = 0) {
    hipStream_t s = on ? on : ctx->stream;
    HashArgs h{};
    h.tiles = r->plan.tiles; h.n_tiles = r->plan.n_tiles;
    h.len = r->len;
    h.srcA = (const uint8_t *)d_src; h.offA = r->src_off; h.baseA = 0;
    h.digests = r->digests; h.tile_cv = r->plan.tile_cv;
    if (d_copy_out) {
        h.srcB = (uint8_t *)d_copy_out; h.offB = r->blob_offset; h.copy_to_B = 1;
        h.store_tiles = ctx->sw.store_g;
        h.copy_mask = r->skip; h.copy_cap = copy_cap;
        h.misaligned_dst = !(r->all_stored_aligned && ((uintptr_t)d_copy_out & 15) == 0);
    }
    // next to a busy encoder (auxiliary stream) the hash keeps out of LDS: the encoder's residency depends on it
    if (on && r->enc_bytes * 4 >= r->in_bytes) h.fold_tiles_max = 1;
    ktime_begin(ctx, "blake3_tiles", s);
    launch_hash_tiles(h, s);
    ktime_end(ctx, s);
    if (r->plan.n_big) {
        ktime_begin(ctx, "blake3_merge_big", s);
        // beside the persistent encoder (auxiliary stream) the merge stays with its one-wave workgroups: a four-wave one waits for
        // four free wave slots on one CU, which the encoder's waves leave only when they retire (seen: 53 ms behind a 65 ms encode)
        launch_merge_big(r->plan.big, r->plan.n_big, r->plan.tile_cv, r->digests, r->plan.grp_big, r->plan.grp_k, r->plan.n_grp, on ? 0xFFFFFFFFu : r->plan.max_cvs, s);
        ktime_end(ctx, s);
    }
    HIPCHK(ctx, hipGetLastError());
    return ZNIPPY_OK;
}

int znippy_hash_rounds(znippy_ctx *ctx, znippy_rounds *r, const void *d_src, uint8_t *digests) {
    if (ctx && ctx->closing) return ZNIPPY_E_INVAL;  // destroyed context kept alive by its tables
    if (!ctx || !r || r->ctx != ctx || !digests || (r->n && !d_src)) return ZNIPPY_E_INVAL;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    ctx->n_ktimes = 0;
    int rc = hash_rounds_async(ctx, r, d_src);
    if (rc) return rc;
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    if (r->n) HIPCHK(ctx, hipMemcpy(digests, r->digests, 32 * (size_t)r->n, hipMemcpyDeviceToHost));
    return ZNIPPY_OK;
}

// ---- single-chunk shims -------------------------------------------------------------------------
static int shim_reserve(znippy_ctx *ctx, size_t in_need, size_t out_need) {
    if (in_need > ctx->shim_in_cap || !ctx->shim_in) {
        if (ctx->shim_in) (void)hipFree(ctx->shim_in);
        ctx->shim_in = nullptr; ctx->shim_in_cap = 0;
        size_t cap = std::max<size_t>(in_need + 64, 1 << 16);
        HIPCHK(ctx, hipMalloc(&ctx->shim_in, cap));
        ctx->shim_in_cap = cap;
    }
    if (out_need > ctx->shim_out_cap || !ctx->shim_out) {
        if (ctx->shim_out) (void)hipFree(ctx->shim_out);
        ctx->shim_out = nullptr; ctx->shim_out_cap = 0;
        size_t cap = std::max<size_t>(out_need + 64, 1 << 16);
        HIPCHK(ctx, hipMalloc(&ctx->shim_out, cap));
        ctx->shim_out_cap = cap;
    }
    return ZNIPPY_OK;
}

int znippy_blake3(znippy_ctx *ctx, const void *src, size_t n, uint8_t out[32]) {
    if (ctx && ctx->closing) return ZNIPPY_E_INVAL;  // destroyed context kept alive by its tables
    if (!ctx || !out || (n && !src)) return ZNIPPY_E_INVAL;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    int rc = shim_reserve(ctx, n, 0);
    if (rc) return rc;
    if (n) HIPCHK(ctx, hipMemcpyAsync(ctx->shim_in, src, n, hipMemcpyHostToDevice, ctx->stream));
    uint64_t off = 0, len = n;
    znippy_rounds *r = nullptr;
    rc = znippy_rounds_create(ctx, &off, &len, nullptr, 1, &r);
    if (rc) return rc;
    rc = znippy_hash_rounds(ctx, r, ctx->shim_in, out);
    znippy_rounds_destroy(r);
    return rc;
}

int znippy_decompress(znippy_ctx *ctx, const void *frame, size_t n, void *dst, size_t cap, size_t *written) {
    if (ctx && ctx->closing) return ZNIPPY_E_INVAL;  // destroyed context kept alive by its tables
    if (!ctx || !frame || !written || (cap && !dst)) return ZNIPPY_E_INVAL;
    uint64_t usize = 0;
    int rc = znippy_get_decompressed_size(frame, n, &usize);
    if (rc) return rc;
    {  // the kernels want the Zstandard magic at byte 0: leading skippable frames stay on the host
        const ptrdiff_t skip = skippable_prefix((const uint8_t *)frame, n);
        frame = (const uint8_t *)frame + skip;
        n -= (size_t)skip;
    }
    if (usize > cap) return ZNIPPY_E_DST_SMALL;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    rc = shim_reserve(ctx, n, usize);
    if (rc) return rc;
    HIPCHK(ctx, hipMemcpyAsync(ctx->shim_in, frame, n, hipMemcpyHostToDevice, ctx->stream));
    uint64_t boff = 0, bsz = n, ooff = 0;
    znippy_rows *r = nullptr;
    rc = znippy_rows_create(ctx, &boff, &bsz, nullptr, &usize, &ooff, nullptr, 0, 1, &r);
    if (rc) return rc;
    int32_t status = 0;
    rc = znippy_decode_verify_rows(ctx, r, ctx->shim_in, 0, ctx->shim_out, ctx->shim_out_cap, nullptr, nullptr, 0, &status);
    znippy_rows_destroy(r);
    if (rc) return rc;
    if (status) return status;
    if (usize) HIPCHK(ctx, hipMemcpy(dst, ctx->shim_out, usize, hipMemcpyDeviceToHost));
    *written = usize;
    return ZNIPPY_OK;
}

}  // extern "C"


// A run's results (offsets, sizes, digests: 48 bytes per round) leave through this kernel, written straight into the
// slot's pinned host mirror (device-visible: hipHostMalloc).  hipMemcpyAsync did the same with a blit kernel of its own,
// but the CALL blocked 6.7-7.4 ms once per table — in the third run, whatever had been copied before (time marks around
// every HIP call of that run: everything else 0.06 ms together; tools/write_steps.py) — which put one 7-12 ms step among
// bench.py's 0.8 ms write steps, behind its three warm-up steps: a quarter of the write leg's rate.  A launch has no such
// mode.
__global__ __launch_bounds__(256) void k_results_out(uint4 *dst, const uint4 *src, uint32_t n16) {
    for (uint32_t i = blockIdx.x * 256 + threadIdx.x; i < n16; i += gridDim.x * 256) dst[i] = src[i];
}

// ---- write side ------------------------------------------------------------------------------------
extern "C" int znippy_encode_hash_rounds_async(znippy_ctx *ctx, znippy_rounds *r, const void *d_src, void *d_blob_out,
                                               uint64_t blob_cap) {
    if (ctx && ctx->closing) return ZNIPPY_E_INVAL;  // destroyed context kept alive by its tables
    if (!ctx || !r || r->ctx != ctx || (r->n && (!d_src || !d_blob_out))) return ZNIPPY_E_INVAL;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    hipStream_t s = ctx->stream;
    ctx->n_ktimes = 0;
    if (!r->n) return ZNIPPY_OK;
    { const int rc0 = ensure_encoder(ctx); if (rc0) return rc0; }
    if (r->prov_bytes + 64 > ctx->enc_prov_cap) {
        HIPCHK(ctx, hipStreamSynchronize(s));
        if (ctx->enc_prov) (void)hipFree(ctx->enc_prov);
        ctx->enc_prov = nullptr; ctx->enc_prov_cap = 0;
        HIPCHK(ctx, hipMalloc(&ctx->enc_prov, r->prov_bytes + 64));
        ctx->enc_prov_cap = r->prov_bytes + 64;
    }
    const unsigned slot = (unsigned)(r->run_seq & 1);
    rounds_select(r, slot);
    if (r->run_seq >= 2) HIPCHK(ctx, hipStreamWaitEvent(s, r->ev_res[slot], 0));  // the slab's previous results have left
    HIPCHK(ctx, hipMemsetAsync(r->res, 0, 16 + 16 * (size_t)r->n, s));  // total, overflow, blob_offset, blob_size
    HIPCHK(ctx, hipMemsetAsync(ctx->cursor, 0, 64, s));  // work cursors + (last word) the count of blocks handed to the wide variant
    // The fork comes AFTER the clears.  The hash kernel on the auxiliary stream starts at the fork and fills the chip; a
    // clear issued behind it is one tiny fill kernel that then waits ~100 us for its turn (kernel trace of the C2 write
    // step: two of them, 88 + 125 us, between the previous run's gather and this run's encoder).
    HIPCHK(ctx, hipEventRecord(ctx->ev_fork, s));
    EncodeArgs a{};
    a.items = r->items;
    a.src = (const uint8_t *)d_src; a.src_off = r->src_off; a.len = r->len;
    a.prov = ctx->enc_prov; a.seq_scratch = ctx->enc_seq;
    a.piece_len = r->piece_len; a.piece_start = r->piece_start; a.tabs = ctx->enc_tabs;
    a.tail_mark = a.high = ctx->level >= HIGH_TIER_LEVEL;
    if (ctx->sw.edbg) {  // diagnostic: phase shares of the previous run's wide-variant blocks
        static unsigned long long *dbg = nullptr;
        if (!dbg) { (void)hipMalloc(&dbg, 64); (void)hipMemset(dbg, 0, 64); }
        unsigned long long h[8];
        (void)hipStreamSynchronize(s);
        (void)hipMemcpy(h, dbg, 64, hipMemcpyDeviceToHost);
        if (h[0]) fprintf(stderr, "[znippy edbg] wide blocks=%llu  cycles per block: setup=%.0f matching=%.0f literals=%.0f sequences=%.0f\n", h[0],
                          (double)h[1] / h[0], (double)h[2] / h[0], (double)h[3] / h[0], (double)h[4] / h[0]);
        (void)hipMemset(dbg, 0, 64);
        a.dbg = dbg;
    }
    // Tables of small encoded rounds only: the encoder's waves hash the rounds they are about to encode (EncodeArgs::fuse_tiles)
    const bool fuse_hash = r->fuse_tiles && !r->store_incompressible && !ctx->sw.nohash && !ctx->sw.no_fuse_hash;
    ktime_begin(ctx, fuse_hash ? "zstd_encode_hash" : "zstd_encode");
    for (int wide = 1; wide >= 0; wide--) {  // the wide share first: its blocks are the long ones
        a.n_items = wide ? r->n_wide : r->n_small;
        if (!a.n_items) continue;
        a.order = a.n_items == r->n_items ? nullptr : (wide ? r->order_wide : r->order_small);  // all of one kind: no indirection
        a.cursor = ctx->cursor + (wide ? 8 : 0);
        const int g = wide ? ctx->encode_grid : ctx->encode_grid_small;
        a.batch = std::max<uint32_t>(1, std::min<uint32_t>(16, a.n_items / (uint32_t)(g * 2)));
        if (!wide) { a.retry_list = r->retry_list; a.retry_count = ctx->cursor + 15; }
        int grid = std::min<int>(g, (int)a.n_items);
        if (!wide && fuse_hash) {
            a.fuse_tiles = r->fuse_tiles;
            a.h = HashArgs{};
            a.h.tiles = r->plan.tiles; a.h.n_tiles = r->plan.n_tiles;
            a.h.len = r->len;
            a.h.srcA = (const uint8_t *)d_src; a.h.offA = r->src_off; a.h.baseA = 0;
            a.h.digests = r->digests; a.h.tile_cv = r->plan.tile_cv;
            grid = std::min<int>(g, (int)((r->plan.n_tiles + r->fuse_tiles - 1) / r->fuse_tiles));
        }
        launch_encode(a, grid, !wide, ctx->level >= HIGH_TIER_LEVEL, s);
        a.fuse_tiles = 0;
    }
    if (r->n_small) {  // second wide launch: whatever the small variant handed over (count on the device)
        a.order = r->retry_list; a.n_items = r->n_small; a.n_items_dev = ctx->cursor + 15;
        a.retry_list = nullptr; a.retry_count = nullptr;
        a.cursor = ctx->cursor + 12;
        a.batch = 1;
        launch_encode(a, std::min<int>(ctx->encode_grid, (int)r->n_small), false, ctx->level >= HIGH_TIER_LEVEL, s);
    }
    ktime_end(ctx);
    // checksum over the ORIGINAL bytes (stream_packer.rs:L219): VALU-bound, submitted to the
    // auxiliary stream right after the persistent (latency-bound) encoder so both share the CUs
    // Store-heavy table (most bytes are skip rounds): hashing and copying the stored bytes are one pass over them once
    // their blob offsets are known — scan and gather run first (the gather leaves the stored pieces alone), then the
    // hash kernel copies what it hashes.  Otherwise the hash runs beside the encoder on the auxiliary stream.
    // (Blobs are packed without gaps, so behind a compressed round the offsets are odd: unless every round is stored
    // and every length a multiple of 16, the launch uses the kernel variant that re-cuts the bytes to 16-byte
    // boundaries on their way out — plain 16-byte stores at odd addresses cost more than the pass they save.)
    const bool heavy = r->in_bytes && (r->in_bytes - r->enc_bytes) * 2 >= r->in_bytes;
    const bool fuse_store = heavy && !r->store_incompressible &&
                            !ctx->sw.no_fused_store && !ctx->sw.nohash;
    int rc = ZNIPPY_OK;
    if (!fuse_store && !fuse_hash) {
        HIPCHK(ctx, hipStreamWaitEvent(ctx->aux, ctx->ev_fork, 0));
        rc = ctx->sw.nohash ? ZNIPPY_OK : hash_rounds_async(ctx, r, d_src, ctx->aux);  // diagnostic switch
        if (rc) return rc;
        HIPCHK(ctx, hipEventRecord(ctx->ev_join, ctx->aux));
    }
    if (r->store_incompressible) {
        ktime_begin(ctx, "store_decide");
        launch_store_decide(r->first_item, r->items, r->len, r->skip, r->n, r->piece_len, r->stored, s);
        ktime_end(ctx);
    }
    ktime_begin(ctx, "piece_scan");
    launch_piece_scan(r->piece_len, r->n_items, r->local_excl, r->block_tot, s);
    ktime_end(ctx);
    GatherArgs g{};
    g.items = r->items; g.n_pieces = r->n_items; g.piece_len = r->piece_len; g.piece_start = r->piece_start;
    g.local_excl = r->local_excl; g.block_tot = r->block_tot; g.prov = ctx->enc_prov;
    g.src = (const uint8_t *)d_src; g.src_off = r->src_off;
    g.blob_out = (uint8_t *)d_blob_out; g.blob_cap = blob_cap;
    g.blob_offset = r->blob_offset; g.blob_size = r->blob_size; g.total = r->total; g.overflow = r->overflow;
    g.stored = r->store_incompressible ? r->stored : nullptr;
    g.skip_stored_copy = fuse_store ? 1 : 0;
    // lane-per-piece gather (64 consecutive pieces per wave): only for tables without blocks above 16 KiB — the consecutive
    // blocks of one big round land in one wave, which then copies them one after the other (a table of 4,900 text files, a
    // few of them above 1 MiB: 0.29 ms against 0.03 for a wave per piece).  Store-heavy table: the stored rounds' 64 KiB
    // slices are not copied here (the hash kernel does it) — only their bookkeeping is left, a lane's work.
    g.small_pieces = r->n_items && r->n_wide == 0 && (fuse_store || r->in_bytes / r->n_items <= 16384) ? 1 : 0;
    ktime_begin(ctx, "gather");
    launch_gather(g, s);
    ktime_end(ctx);
    if (fuse_store) {
        rc = hash_rounds_async(ctx, r, d_src, nullptr, d_blob_out, blob_cap);
        if (rc) return rc;
    } else if (!fuse_hash) {
        HIPCHK(ctx, hipStreamWaitEvent(s, ctx->ev_join, 0));  // digests are complete once the main stream drains
    }
    // the results leave on the copy stream (one DMA into the slot's pinned mirror) while the main stream is free for
    // the next run
    HIPCHK(ctx, hipEventRecord(r->ev_enc[slot], s));
    HIPCHK(ctx, hipStreamWaitEvent(ctx->copy, r->ev_enc[slot], 0));
    {
        const uint32_t n16 = (uint32_t)((r->res_bytes + 15) / 16);  // both buffers are allocated in whole 16-byte units and more
        // few workgroups: the copy only has to be done before the next run but one needs the slab, and it shares the chip with
        // the next run's kernels (4-16 workgroups: 0.84 ms per C2 write step, 64: 0.88, 1,024: 0.90)
        hipLaunchKernelGGL(k_results_out, dim3(std::min<uint32_t>((n16 + 255) / 256, 8)), dim3(256), 0, ctx->copy,
                           reinterpret_cast<uint4 *>(r->h_res), reinterpret_cast<const uint4 *>(r->res), n16);
    }
    HIPCHK(ctx, hipEventRecord(r->ev_res[slot], ctx->copy));
    r->run_seq++;
    HIPCHK(ctx, hipGetLastError());
    return ZNIPPY_OK;
}

// Results of the run `lag` runs before the latest one (0 or 1) as pointers into that run's pinned mirror; waits for
// that run's copy only.  Valid until two more encode calls have been queued on the table.
extern "C" int znippy_rounds_results_lagged(znippy_ctx *ctx, znippy_rounds *r, unsigned lag, const uint64_t **blob_offset,
                                            const uint64_t **blob_size, const uint8_t **checksum, uint64_t *blob_bytes) {
    if (ctx && ctx->closing) return ZNIPPY_E_INVAL;  // destroyed context kept alive by its tables
    if (!ctx || !r || r->ctx != ctx || lag > 1 || r->run_seq <= lag) return ZNIPPY_E_INVAL;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    const unsigned slot = (unsigned)((r->run_seq - 1 - lag) & 1);
    HIPCHK(ctx, hipEventSynchronize(r->ev_res[slot]));
    const uint8_t *h = r->h_res_m[slot];
    uint64_t total;
    uint32_t ovf;
    memcpy(&total, h, 8);
    memcpy(&ovf, h + 8, 4);
    if (ovf) return ZNIPPY_E_DST_SMALL;
    const size_t n = r->n;
    if (blob_offset) *blob_offset = reinterpret_cast<const uint64_t *>(h + 16);
    if (blob_size) *blob_size = reinterpret_cast<const uint64_t *>(h + 16 + 8 * n);
    if (checksum) *checksum = h + 16 + 16 * n;
    if (blob_bytes) *blob_bytes = total;
    return ZNIPPY_OK;
}

extern "C" int znippy_rounds_results(znippy_ctx *ctx, znippy_rounds *r, uint64_t *blob_offset, uint64_t *blob_size,
                                     uint8_t *checksum, uint8_t *compressed, uint64_t *blob_bytes) {
    if (ctx && ctx->closing) return ZNIPPY_E_INVAL;  // destroyed context kept alive by its tables
    if (!ctx || !r || r->ctx != ctx) return ZNIPPY_E_INVAL;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    if (blob_bytes) *blob_bytes = 0;
    if (!r->n) return ZNIPPY_OK;
    if (!r->run_seq) return ZNIPPY_E_INVAL;  // nothing has been encoded on this table yet
    HIPCHK(ctx, hipEventSynchronize(r->ev_res[(r->run_seq - 1) & 1]));  // the latest run's results have arrived
    if (r->store_incompressible) {
        HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
        HIPCHK(ctx, hipMemcpy(r->h_stored, r->stored, r->n, hipMemcpyDeviceToHost));
    }
    uint64_t total;
    uint32_t ovf;
    memcpy(&total, r->h_res, 8);
    memcpy(&ovf, r->h_res + 8, 4);
    if (ovf) return ZNIPPY_E_DST_SMALL;
    const size_t n = r->n;
    if (blob_offset) memcpy(blob_offset, r->h_res + 16, 8 * n);
    if (blob_size) memcpy(blob_size, r->h_res + 16 + 8 * n, 8 * n);
    if (checksum) memcpy(checksum, r->h_res + 16 + 16 * n, 32 * n);
    if (compressed)
        for (uint32_t i = 0; i < r->n; i++)
            compressed[i] = ((!r->h_skip.empty() && r->h_skip[i]) || (r->store_incompressible && r->h_stored[i])) ? 0 : 1;
    if (blob_bytes) *blob_bytes = total;
    return ZNIPPY_OK;
}

// Zero-copy variant: pointers into the table's pinned result mirror, valid until the next encode call
// on this table (compressed[] = !skip is known to the caller already).
extern "C" int znippy_rounds_results_view(znippy_ctx *ctx, znippy_rounds *r, const uint64_t **blob_offset,
                                          const uint64_t **blob_size, const uint8_t **checksum, uint64_t *blob_bytes) {
    if (ctx && ctx->closing) return ZNIPPY_E_INVAL;  // destroyed context kept alive by its tables
    int rc = znippy_rounds_results(ctx, r, nullptr, nullptr, nullptr, nullptr, blob_bytes);
    if (rc) return rc;
    const size_t n = r->n;
    if (blob_offset) *blob_offset = reinterpret_cast<const uint64_t *>(r->h_res + 16);
    if (blob_size) *blob_size = reinterpret_cast<const uint64_t *>(r->h_res + 16 + 8 * n);
    if (checksum) *checksum = r->h_res + 16 + 16 * n;
    return ZNIPPY_OK;
}

extern "C" int znippy_encode_hash_rounds(znippy_ctx *ctx, znippy_rounds *rounds, const void *d_src, void *d_blob_out,
                                         uint64_t blob_cap, uint64_t *blob_offset, uint64_t *blob_size,
                                         uint8_t *checksum, uint8_t *compressed, uint64_t *blob_bytes) {
    if (ctx && ctx->closing) return ZNIPPY_E_INVAL;  // destroyed context kept alive by its tables
    int rc = znippy_encode_hash_rounds_async(ctx, rounds, d_src, d_blob_out, blob_cap);
    if (rc) return rc;
    return znippy_rounds_results(ctx, rounds, blob_offset, blob_size, checksum, compressed, blob_bytes);
}

extern "C" int znippy_compress(znippy_ctx *ctx, const void *src, size_t n, void *dst, size_t cap, size_t *written) {
    if (ctx && ctx->closing) return ZNIPPY_E_INVAL;  // destroyed context kept alive by its tables
    if (!ctx || !written || !dst || (n && !src)) return ZNIPPY_E_INVAL;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    const size_t bound = znippy_compress_bound(n);
    int rc = shim_reserve(ctx, n, bound);
    if (rc) return rc;
    if (n) HIPCHK(ctx, hipMemcpyAsync(ctx->shim_in, src, n, hipMemcpyHostToDevice, ctx->stream));
    uint64_t off = 0, len = n, bsz = 0, total = 0;
    znippy_rounds *r = nullptr;
    rc = znippy_rounds_create(ctx, &off, &len, nullptr, 1, &r);
    if (rc) return rc;
    rc = znippy_encode_hash_rounds(ctx, r, ctx->shim_in, ctx->shim_out, ctx->shim_out_cap, nullptr, &bsz, nullptr, nullptr,
                                   &total);
    znippy_rounds_destroy(r);
    if (rc) return rc;
    if (total > cap) return ZNIPPY_E_DST_SMALL;
    HIPCHK(ctx, hipMemcpy(dst, ctx->shim_out, total, hipMemcpyDeviceToHost));
    *written = total;
    return ZNIPPY_OK;
}

extern "C" size_t znippy_compress_bound(size_t n) {
    // every 128 KiB block can fall back to a raw block (3-byte header) + frame header (<= 13) + the empty closing block
    // the higher effort tier puts behind frames of several blocks
    return n + 3 * (n / BLOCK_BYTES + 1) + 19;
}

extern "C" int znippy_rounds_set_store_incompressible(znippy_rounds *r, int on) {
    if (!r) return ZNIPPY_E_INVAL;
    r->store_incompressible = on != 0;
    return ZNIPPY_OK;
}
