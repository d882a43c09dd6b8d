// Shared host/device declarations for libznippy_hip.so (internal; the public surface is
// include/znippy_hip.h).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>

namespace zn {

// A hash work item = one wavefront's worth of BLAKE3 leaves (<= 64 x 1 KiB).
//   n_units >= 1 : `n_units` whole small units (each <= 64 leaves) packed so their leaves fill
//                  the wave's lanes; digests are finished inside the wave.
//   n_units == 0 : leaves [first_leaf, first_leaf+n_leaves) of big unit `first_unit`
//                  (first_leaf is a multiple of 64 -> a complete subtree); the subtree CV goes
//                  to tile_cv[cv_index] and the big-unit merge kernel finishes the tree.
struct Tile {
    uint32_t first_unit;
    uint32_t n_units;
    uint32_t first_leaf;
    uint32_t n_leaves;
    uint32_t cv_index;
    uint32_t pad;
};

struct BigUnit {  // a unit with more than 64 leaves
    uint32_t unit;
    uint32_t cv_base;  // first tile CV of this unit in tile_cv
    uint32_t n_cvs;    // ceil(leaves/64) >= 2
    uint32_t pad;      // n_cvs > 64: index (in CVs) of this unit's group CVs in tile_cv, ceil(n_cvs/64) of them
};

// Where each unit's bytes live.  Units with sel==0 (or all units when sel==nullptr) are hashed
// from srcA+offA[u]; when copy_to_B is set those bytes are also copied to srcB+offB[u] (store
// path: blob -> output, fused with the hash).  Units with sel!=0 are hashed from srcB+offB[u]
// (decoded output of the zstd kernel).
struct HashArgs {
    const Tile *tiles;
    uint32_t n_tiles;
    const uint64_t *len;
    const uint8_t *srcA;
    const uint64_t *offA;
    uint64_t baseA;  // subtracted from offA (blob_base)
    uint8_t *srcB;
    const uint64_t *offB;
    const uint8_t *sel;
    const int32_t *status;          // per-unit decode status (read side), see hash_dev.h PASS_*
    const uint32_t *pending_count;  // rows handed to the general decoder (PASS_SECOND early-out)
    int pass;
    int copy_to_B;
    uint32_t *digests;  // 8 words per unit
    uint32_t *tile_cv;  // 8 words per big-unit tile
    int store_tiles;     // store path (copy_to_B): 0 = tiles per wave picked from the tile count; 1 / 2 = forced (ZNIPPY_STORE_G, A/B and tests)
    int fold_tiles_max;  // 0 = pick from the tile count; 1 = no deferred folding (no LDS: the launch shares the CUs with the encoder)
    const uint8_t *tile_done;  // optional, PASS_SECOND: tiles already hashed by the fused block kernel
    // optional (write side, store-heavy tables): copy only units with copy_mask != 0, and only if they end inside copy_cap
    const uint8_t *copy_mask;
    uint64_t copy_cap;
    int misaligned_dst;  // store path: some copy destination is not 16-byte aligned (host knowledge; picks the kernel variant)
};

// Row status values on the read side: 0 = done (stored row, or decoded+hashed by the fused
// small-row kernel), 1 = handed to the general decoder, 2 = decoded by the general decoder
// (still to be hashed by the second pass), < 0 = ZNIPPY_E_* (frame failed to decode).
struct DecodeArgs {
    const uint32_t *list_a;         // rows the host routes to the general decoder (big compressed rows)
    uint32_t n_list_a;
    const uint32_t *pending;        // rows the fused kernel could not handle
    const uint32_t *pending_count;
    const uint8_t *blobs;
    uint64_t blob_base;
    const uint64_t *blob_off, *blob_size, *usize, *out_off;
    const uint8_t *compressed;  // per row 0/1
    uint8_t *out;
    uint64_t out_cap;
    int32_t *status;
    uint32_t n_rows;
    uint32_t *cursor;    // atomic row cursor (the reference's AtomicUsize, decompress.rs:L104)
    uint8_t *lit_scratch;  // per resident workgroup: LIT_SCRATCH bytes
    unsigned long long *dbg;  // diagnostic only (ZNIPPY_DDBG): phase cycle counters [items, literals, seq tables, seq decode, execute, tail], never an output
    // Block items: frames of >= 2 blocks are first tried block by block, every block a work item of its own (a
    // frame written by this library has self-contained 128 KiB blocks; any frame that turns out not to — repeat
    // offsets, reused tables, a match reaching into an earlier block, another block size — is flagged and decoded
    // serially afterwards).  block_mode != 0: the work list is the item table instead of rows.
    int block_mode;
    const uint32_t *item_row, *item_k;  // item -> row, block index inside the row's frame
    const uint32_t *item_src;           // item -> offset of the block header inside the frame (0xFFFFFFFF: skip)
    uint32_t n_items;
    uint32_t *row_flag;                 // per row, != 0: decode this frame serially
    const uint8_t *item_done;           // optional: items already written (and hashed) by the fused block kernel
    const uint32_t *todo, *n_todo;      // optional: the items that are left (k_compact_items); n_work = *n_todo
    uint32_t *seq_scratch;
    int preset;  // != 0: status[] was initialised with the host's verdicts — a row with status < 0 is left alone
};

struct FusedArgs {
    HashArgs h;  // tiles, len (= usize), srcA = blobs, srcB = out, sel = compressed, status, digests
    const uint64_t *blob_size;
    uint64_t out_cap;
    int32_t *status;
    uint32_t *pending;
    uint32_t *pending_count;
    // diagnostic only (ZNIPPY_DBG, bit set): 1 skip the hash, 2 skip the decode, 4 hash recognised rows from the output
    // instead of their windows, 8 phase stamps -> dbg_buf, 16 no row stores, 32 no pattern expansion (scalar decoder),
    // 64 no s_setprio on the scalar decoder, 128 no lane-parallel recognition, 256 parent trees folded per wave
    int dbg;
    unsigned long long *dbg_buf;  // diagnostic stamps (ZNIPPY_DBG & 8)
    uint32_t lds_pad;  // extra dynamic LDS per block: caps blocks/CU (in-flight footprint vs Infinity Cache)
    int preset;        // != 0: status[] was initialised with the host's verdicts — a row with status < 0 is left alone
    // role-split kernel (k_fused_roles): global work cursor over the plan's tiles, and the tiles it leaves to
    // k_fused_small (any tile that is not all "whole-leaf rows of the recognised periodic shape")
    uint32_t *cursor;
    uint32_t *tile_list;         // k_fused_roles: slow list (out); k_fused_small: tiles to process (nullptr = all)
    uint32_t *tile_count;
};

// Block items of the common shape (fused_small.hip, k_fused_blocks): the big-slice tiles of block-candidate rows.
struct FusedBlocksArgs {
    HashArgs h;  // as for the fused small-row kernel, pass = PASS_ALL
    const uint64_t *blob_size;
    const uint32_t *bt_tile, *bt_item;  // tile index in the plan, block item the tile belongs to
    uint32_t n_bt;
    const uint32_t *item_src;  // written by k_scan_blocks
    const uint32_t *row_flag;
    uint8_t *tile_done, *item_done;  // zeroed before the launch
    int dbg;
};
void launch_fused_blocks(const FusedBlocksArgs &a, hipStream_t s);

void launch_hash_tiles(const HashArgs &a, hipStream_t s);
void launch_fused_small(const FusedArgs &a, hipStream_t s, int grid_cap = 0);
void launch_fused_roles(const FusedArgs &a, int cus, hipStream_t s);
void init_fused_tables();
void launch_merge_big(const BigUnit *big, uint32_t n_big, uint32_t *tile_cv, uint32_t *digests, const uint32_t *grp_big,
                      const uint32_t *grp_k, uint32_t n_grp, uint32_t max_cvs, hipStream_t s);
void launch_verify(const uint32_t *digests, const uint8_t *checksum, const uint64_t *usize,
                   const int32_t *status, uint32_t n_rows, uint64_t row_begin, uint64_t *counters,
                   uint64_t *corrupt_rows, uint32_t corrupt_cap, hipStream_t s, const uint32_t *lean_lists = nullptr, uint32_t lean_mask = 0);
int decode_grid_size(int device);
void launch_decode(const DecodeArgs &a, int grid, bool wide, hipStream_t s);
// block-item path: header scan of the candidate frames, then (after the block-mode decode) the per-row verdict
struct BlockScanArgs {
    const uint32_t *cand_row, *cand_base, *cand_nblocks;  // candidate -> row, first item, number of items
    uint32_t n_cand;
    const uint8_t *blobs;
    uint64_t blob_base;
    const uint64_t *blob_off, *blob_size, *usize, *out_off;
    uint64_t out_cap;
    uint32_t *item_src;
    uint32_t *row_flag;
    int32_t *status;
    uint32_t *pending, *pending_count;
    int preset;  // != 0: rows with status < 0 (host verdict) are not candidates
};
void launch_scan_blocks(const BlockScanArgs &a, hipStream_t s);
void launch_compact_items(const uint8_t *item_done, uint32_t n_items, uint32_t *todo, uint32_t *n_todo, hipStream_t s);
void launch_finish_blocks(const BlockScanArgs &a, hipStream_t s, bool defer_flagged);  // defer_flagged: a flagged candidate is left to the batch path

// Foreign frames in two phases (zstd_decode.hip, k_fz_*): frames of >= 2 blocks that the block-item path gave up on
// (another writer's frames: repeat offsets, reused entropy tables, matches reaching into earlier blocks, any block
// size) are decoded block-PARALLEL where the format allows it — every block's literals and sequences are entropy-
// decoded by its own workgroup into scratch pools — and only the byte-moving sequence execution walks a frame in order.
struct FzItem {
    uint32_t src;       // offset of the block header inside the frame (written by the scan)
    uint32_t out;       // bytes the block regenerates
    uint32_t nseq;
    uint32_t lit_len;
    uint64_t seq_off;   // first record in the sequence pool
    uint64_t lit_off;   // lit_kind 0: offset inside the frame's blob; 1: the byte; 2: offset in the literal pool
    uint32_t lit_kind;
    int32_t err;        // != 0: this block needs the serial decoder (which also produces the error code)
    // Repeat offsets are resolved while the sequences are decoded (one scalar walk per block, all blocks at once),
    // against an UNKNOWN incoming history: a value is an offset, or FZ_SYM | k << 26 | d = "incoming entry k, minus d".
    // rep[] = the history the block leaves, in the same notation; the execute kernel substitutes frame by frame.
    uint32_t rep[3];
    uint32_t pad;
};
constexpr uint32_t FZ_SYM = 1u << 28;
struct FzArgs {
    const uint32_t *cand_row, *cand_fzbase, *cand_fzcap;  // candidate -> row, first item slot, item slots
    uint32_t n_cand;
    const uint32_t *it_cand;  // item slot -> candidate
    uint32_t total_items;
    uint32_t *cand_nb;        // blocks found per candidate (0: not taken by this path)
    FzItem *items;
    const uint8_t *blobs;
    uint64_t blob_base;
    const uint64_t *blob_off, *blob_size, *usize, *out_off;
    uint64_t out_cap;
    uint8_t *out;
    uint32_t *row_flag;       // in: != 0 = the block-item path gave up; out: 0 once the frame is decoded here
    const int32_t *status;
    int preset;
    uint8_t *lit_pool;
    uint64_t lit_cap;         // bytes
    unsigned long long *seq_pool;
    uint64_t seq_cap;         // records
    unsigned long long *pool_used;  // [0] literal bytes, [1] sequence records handed out, [2] frames decoded, [3] blocks given up (zeroed per run)
    uint32_t *cursor;
    unsigned long long *dbg;  // diagnostic only (ZNIPPY_DDBG): cycle / event counters of the execute kernel
};
void launch_fz_scan(const FzArgs &a, uint32_t *work, uint32_t *work_count, hipStream_t s);
void launch_fz_entropy(const FzArgs &a, int cus, const uint32_t *work, const uint32_t *work_count, hipStream_t s);
void launch_fz_exec(const FzArgs &a, hipStream_t s);

// Many foreign frames at once (zstd_batch.hip, k_bx_*): the serial chains of a block — table descriptions, Huffman
// streams, the FSE sequence bitstream — are decoded with LANE = block (or stream), 64 of them per wave, tables in
// scratch pools in device memory; execution stays wave = frame (k_fz_exec's body).  Everything is sized and listed on
// the device: the candidates are the host's list of big single-block rows, the rows the fused kernel handed over, and
// the block candidates the block-item path flagged.
struct BxPrep {            // per block item, written by k_bx_prep
    uint32_t frame;        // candidate slot
    uint32_t k;            // index of the block inside its frame
    uint32_t n_streams;    // Huffman streams (0: raw / RLE literals)
    uint32_t huf_off;      // decoding table in the Huffman pool (u16 cells), 1 << huf_log of them
    uint32_t huf_log;
    uint32_t st_off[4];    // streams: offset from the frame's first byte, bytes
    uint32_t st_len[4];
    uint32_t tab[3];       // LL / OF / ML decoding table in the FSE pool (2-byte cells, 16-byte aligned)
    uint32_t logs;         // log LL | log OF << 8 | log ML << 16
    uint32_t bs_off;       // sequences bitstream: offset from the frame's first byte
    uint32_t bs_len;       // bytes
};
constexpr uint32_t BX_BIG_SEQ = 2048;  // blocks of this many sequences get a wave of their own in the sequence stage
constexpr uint32_t BX_PREDEF_LL = 0, BX_PREDEF_OF = 64, BX_PREDEF_ML = 96, BX_POOL_FIRST = 160;  // predefined tables at the head of the FSE pool
// FSE pool cell (2 bytes): symbol:6 | ns:10 << 6 (zstd_batch.hip)
struct BxArgs {
    const uint32_t *list_a; uint32_t n_list_a;
    const uint32_t *pending; const uint32_t *pending_count;
    const uint32_t *bc_row; uint32_t n_bc;  // block candidates: taken when row_flag != 0
    const uint8_t *blobs; uint64_t blob_base;
    const uint64_t *blob_off, *blob_size, *usize, *out_off;
    uint64_t out_cap;
    uint8_t *out;
    int32_t *status;
    int preset;
    uint32_t *row_flag;  // per row: 1 = taken here and not (yet) decoded
    uint32_t *cand_row, *cand_base, *cand_nb;  // per candidate slot (written by the scan)
    uint32_t slot_cap;
    FzItem *items; BxPrep *prep; uint32_t item_cap;
    uint32_t *ctr;  // [0] slots, [1] items, [2] Huffman list, [3] [5] [6] [7] sequence lists (64 / 32 / 16 / 1 blocks per wave), [4] execute cursor, [8] cursor of the wave-per-block list (zeroed per run)
    uint32_t *sort_tmp;  // 5 x item_cap entries: scratch of k_bx_sort
    uint32_t *huf_list, *seq_list;  // seq_list: four lists of item_cap entries (64 / 32 / 16 blocks per wave, wave per block)
    uint8_t *lit_pool; uint64_t lit_cap;
    unsigned long long *seq_pool; uint64_t seq_cap;
    uint16_t *fse_pool; uint64_t fse_cap;   // cells
    uint16_t *huf_pool; uint64_t huf_cap;   // cells
    unsigned long long *pool_used;  // [0] literal bytes [1] records [2] frames decoded [3] blocks given up [4..7] why [8] FSE cells [9] Huffman cells (zeroed per run)
    uint32_t *pending2, *pending2_count;  // what is left for the serial decoder
    unsigned long long *dbg;
    int small_frames;  // the table's frames average <= 64 KiB: the execute stage runs its small-window variant (more frames per CU)
    uint32_t big_seq;  // blocks of this many sequences get a wave of their own (BX_BIG_SEQ; ZNIPPY_BX_BIG for A/B runs)
    // The resolve path (k_rx_*): frames of >= RX_MIN bytes are not executed by one wave each but resolved in parallel — every
    // output byte gets a 32-bit word, RX_DONE | value for a literal byte, the word index it copies from for a match byte;
    // rounds of pointer jumping turn every word into a value; a last pass stores the bytes.  ctr[9] list length, [10] expand
    // cursor, [11] frames taken; pool_used[10] words handed out, [11] words in use (the extent the rounds run over).
    uint32_t *rx_ptr; uint64_t rx_cap;  // the word pool (rx_cap < 2^31 words: a word with bit 31 clear is an index)
    uint32_t *rx_chunk;                 // 1,024-word chunk of the pool -> candidate slot
    uint8_t *rx_cdone;                  // ... -> every word of the chunk is a value (set by a jump round, cleared by the plan)
    uint32_t *rx_base, *rx_fail;        // per candidate slot: first word (RX_NONE: not taken), a block did not check out
    uint32_t *rx_blk;                   // per block item: first output byte inside the frame, incoming repeat offsets [3]
    uint32_t *rx_list;                  // block items to expand
    uint32_t *rx_pending;               // [round]: words still unresolved after that round (zeroed per run)
    uint64_t rx_bound;                  // host's bound on the words a run can use (grid sizing)
    uint32_t rx_min;                    // frames of at least this many bytes are resolved (RX_MIN; less for tables with few rows above 64 KiB)
};
constexpr uint32_t RX_NONE = 0xFFFFFFFFu, RX_DONE = 0x80000000u, RX_MIN = 256u << 10, RX_ROUNDS = 12, RX_JUMPS = 6;
void launch_bx_stage(const BxArgs &a, int cus, int stage, hipStream_t s);  // 0 scan, 1 prep, 7 sort the work lists, 2 huf, 3 fse (lane = block), 4 exec, 5 finish, 6 fse (wave = block), 8 resolve: plan, 9 expand, 10 + r jump round r, 30 store
void bx_predefined_tables(uint16_t cells[160]);  // host: the three predefined tables as pool cells

}  // namespace zn
