// Internal declarations shared by zstd_encode.hip and api.hip.
#pragma once
#include "common.h"
#include <cstring>
#include <vector>

namespace zn {

constexpr uint32_t BLOCK_BYTES = 128 * 1024;  // zstd Block_Maximum_Size
constexpr uint32_t HDR_ROOM = 16;             // space in front of block 0 for the frame header
constexpr uint32_t MAX_SEQ = 16384;           // sequences per block (block_len/8 at most, wide variant)
constexpr uint32_t SKIP_PIECE = 64u << 10;    // store-path rounds are gathered in 64 KiB pieces (one wave each)

struct FseSymTT {
    int32_t delta_find_state;
    uint32_t delta_nb_bits;
};

struct EncTables {  // FSE encoding tables of the predefined LL / ML / OF distributions
    uint16_t ll_state[64], ml_state[64], of_state[32];
    FseSymTT ll_tt[36], ml_tt[53], of_tt[29];
};

constexpr uint32_t ITEM_SKIP = 1, ITEM_FIRST = 2;

// One output piece.  Encoded rounds: one item per <=128 KiB block (prov = offset of the item's
// provisional slot).  Store-path rounds: one item per <=1 MiB slice (prov = byte offset of the
// slice inside the round; the gather pass copies it straight from the staging buffer).
struct EncItem {
    uint32_t round;
    uint32_t block;
    uint32_t n_blocks;
    uint32_t flags;
    uint64_t prov;
};

// provisional slot of one block: literals + sequences section, plus room behind the raw literals where the wide
// variant stages its Huffman streams before moving them in front (any block can end up in the wide variant)
__host__ __device__ inline uint64_t enc_slot_bytes(uint32_t n) { return ((HDR_ROOM + 16 + 2ull * n + (n >> 2) + 64) + 15) & ~15ull; }

struct EncodeArgs {
    const EncItem *items;
    const uint32_t *order;  // optional: indices into items[] this launch works through (NULL = 0..n_items-1)
    uint32_t n_items;       // entries in order[] (or in items[])
    const uint32_t *n_items_dev;  // optional: the count is read from the device instead (retry launch)
    unsigned long long *dbg;  // diagnostic only (ZNIPPY_EDBG): [blocks, setup, matching, literals, sequences] cycle counters of the wide variant
    uint32_t *retry_list, *retry_count;  // small variant: blocks that ran out of sequence budget go here, for the wide variant
    uint32_t *cursor;
    uint32_t batch;  // items per cursor dequeue
    const uint8_t *src;
    const uint64_t *src_off, *len;
    uint8_t *prov;
    uint32_t *seq_scratch;  // per resident wave: 3 * MAX_SEQ words
    uint32_t *piece_len;
    uint64_t *piece_start;  // offset in prov where the finished piece begins
    const EncTables *tabs;
    int tail_mark;  // higher effort tier: frames of several blocks end with an empty raw block (zstd_encode.hip)
    int high;       // higher effort tier: the small variant keeps only blocks of the periodic shape
    // Hash tiles inside the encoder (small variant, tables of small encoded rounds only: one item per round, item index =
    // round index): the work unit is a hash TILE — the wave hashes the tile's rounds (blake3::hash(src), stream_packer.rs:L219),
    // then encodes them; the cursor counts tiles.  fuse_tiles == 0: items as usual, the hash is a kernel of its own.
    int fuse_tiles;
    HashArgs h;
};

struct GatherArgs {
    const EncItem *items;
    uint32_t n_pieces;
    const uint32_t *piece_len;
    const uint64_t *piece_start;
    const uint64_t *local_excl, *block_tot;
    const uint8_t *prov, *src;
    const uint64_t *src_off;
    uint8_t *blob_out;
    uint64_t blob_cap;
    uint64_t *blob_offset, *blob_size, *total;
    uint32_t *overflow;
    const uint8_t *stored;  // optional: rounds the store-if-incompressible pass turned into raw payloads
    int skip_stored_copy;   // store-heavy tables: the hash kernel copies the stored rounds while it hashes them
    int small_pieces;       // the table's rounds average <= 16 KiB: lane = piece (k_gather); else wave = piece (k_gather_wide)
};

void launch_encode(const EncodeArgs &a, int grid, bool small_blocks, bool high, hipStream_t s);
constexpr int HIGH_TIER_LEVEL = 4;  // compression levels from here up use the higher effort tier of the wide encoder
void launch_piece_scan(const uint32_t *piece_len, uint32_t n, uint64_t *local_excl, uint64_t *block_tot, hipStream_t s);
void launch_gather(const GatherArgs &g, hipStream_t s);
void launch_store_decide(const uint32_t *first_item, const EncItem *items, const uint64_t *len, const uint8_t *skip,
                         uint32_t n_rounds, uint32_t *piece_len, uint8_t *stored, hipStream_t s);
void build_encode_tables(EncTables *t);

}  // namespace zn
