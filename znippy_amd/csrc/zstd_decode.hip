// Zstandard frame decoder for CDNA4 (gfx950) — the GPU side of codec::decompress_into
// (znippy-common/src/codec.rs:L67-78) as called from the read worker loop
// (znippy-common/src/decompress.rs:L135-166).
//
// Work distribution restates the reference's Gatling read model: persistent workgroups share
// ONE atomic row cursor (decompress.rs:L104,L136); each pulls the next index row, decodes that
// row's frame from the resident blob region straight to the row's output position.
//
// Inside a frame: header/entropy-table parsing and the FSE sequence bitstream are serial
// (lane 0), Huffman literal streams decode 4 lanes wide (one lane per stream), tables live in
// LDS, and every byte-moving step (raw/RLE blocks, literal runs, LZ matches incl. overlapping
// ones via period doubling) is a wave- or workgroup-cooperative coalesced copy.
// Integer/bitstream work: no MFMA.  Bounded by HBM bandwidth on the copy side.
#include "common.h"
#include "zstd_dev.h"

namespace zn {

// ---------------------------------------------------------------------------------------------
// the kernel
// ---------------------------------------------------------------------------------------------
// diagnostic phase stamps (ZNIPPY_DDBG): thread 0 adds the cycles since the previous stamp to counter i
#define DSTAMP(i) do { if (a.dbg && tid == 0) { const unsigned long long now_ = __builtin_amdgcn_s_memtime(); atomicAdd(&a.dbg[i], now_ - t_last); t_last = now_; } } while (0)

template <int NWAVES>
__global__ __launch_bounds__(NWAVES * 64, NWAVES == 4 ? 4 : 1) void k_zstd_decode(DecodeArgs a) {
    using Shared = SharedT<NWAVES>;
    __shared__ Shared S;
    const uint32_t tid = threadIdx.x, NT = NWAVES * 64;
    const bool wave0 = tid < 64;
    uint8_t *const lit_buf = a.lit_scratch + (size_t)blockIdx.x * LIT_SCRATCH_BYTES;

    // nothing routed here (every row was handled by the fused small-row kernel): leave at once
    if (!a.block_mode && a.n_list_a == 0 && *a.pending_count == 0) return;
    // block items: with a to-do list (k_compact_items) only the items the fused block kernel left are dequeued
    const uint32_t n_work = a.block_mode ? (a.todo ? *a.n_todo : a.n_items) : a.n_list_a + *a.pending_count;
    if (a.block_mode && n_work == 0) return;

    // predefined tables, once per workgroup (lane 0; tiny)
    if (tid == 0) {
        for (int i = 0; i < 36; i++) S.norm[i] = c_ll_default[i];
        fse_build(S, S.dll, 36, 6, K_LL);
        for (int i = 0; i < 53; i++) S.norm[i] = c_ml_default[i];
        fse_build(S, S.dml, 53, 6, K_ML);
        for (int i = 0; i < 29; i++) S.norm[i] = c_of_default[i];
        fse_build(S, S.dof, 29, 5, K_OF);
    }
    __syncthreads();

    for (;;) {
        if (tid == 0) {
            const uint32_t w = atomicAdd(a.cursor, 1u);
            S.row = (a.block_mode && a.todo && w < n_work) ? a.todo[w] : w;
            S.claim = w;
            // block items: one thread decides for the workgroup whether the item is still worth decoding (the flag is
            // raised concurrently by other workgroups, so it must be sampled once)
            const uint32_t it = S.row;
            S.skip = a.block_mode && w < n_work && (a.item_src[it] == 0xFFFFFFFFu || a.row_flag[a.item_row[it]] != 0 ||
                                                    (a.item_done && a.item_done[it]));
            if (!a.block_mode && a.preset && w < n_work)  // the host already ruled on this row (bad source / output range)
                S.skip = a.status[w < a.n_list_a ? a.list_a[w] : a.pending[w - a.n_list_a]] < 0;
        }
        __syncthreads();
        const uint32_t widx = S.row;
        if (S.claim >= n_work) break;
        uint32_t row, item_k = 0, item_src = 0;
        if (S.skip) { __syncthreads(); continue; }  // block item not eligible / frame already given up / host verdict
        if (a.block_mode) {
            row = a.item_row[widx]; item_k = a.item_k[widx]; item_src = a.item_src[widx];
        } else row = widx < a.n_list_a ? a.list_a[widx] : a.pending[widx - a.n_list_a];

        unsigned long long t_last = a.dbg ? __builtin_amdgcn_s_memtime() : 0;
        if (a.dbg && tid == 0) atomicAdd(&a.dbg[0], 1ull);
        const uint8_t *const src = a.blobs + (a.blob_off[row] - a.blob_base);
        const uint64_t src_n = a.blob_size[row];
        const uint8_t *const blob_end = src + src_n;
        uint8_t *const out = a.out + a.out_off[row];

        // ---- frame header (lane 0); a block item starts at its block header with a clean entropy state ----
        if (tid == 0 && a.block_mode) {
            const uint64_t fcs = a.usize[row], o0 = (uint64_t)item_k * BLOCK_MAX;
            S.out_pos = o0; S.win_n = 0; S.hist_n = 0;
            S.blk_base = o0;
            S.huf_valid = 0; S.valid[0] = S.valid[1] = S.valid[2] = 0;
            S.rep[0] = 0; S.rep[1] = 0; S.rep[2] = 0;  // the history the block before leaves is not known here: 0 = undefined
            S.has_cksum = 0;
            S.content_size = fcs;
            S.src_pos = item_src;
            S.src_end = src_n;
            S.out_end = o0 + BLOCK_MAX < fcs ? o0 + BLOCK_MAX : fcs;  // this block's share of the output, exactly
            S.err = 0;
            S.blk_last = 0;
        }
        if (tid == 0 && !a.block_mode) {
            int err = 0;
            uint64_t pos = 0;
            S.out_pos = 0; S.win_n = 0; S.hist_n = 0;
            S.blk_base = 0;
            S.huf_valid = 0; S.valid[0] = S.valid[1] = S.valid[2] = 0;
            S.rep[0] = 1; S.rep[1] = 4; S.rep[2] = 8;
            if (src_n < 5) err = E_TRUNC;
            else {
                uint32_t magic = src[0] | (src[1] << 8) | (src[2] << 16) | ((uint32_t)src[3] << 24);
                if (magic != 0xFD2FB528u) err = E_CORRUPT;
                else {
                    uint32_t fhd = src[4];
                    uint32_t fcs_flag = fhd >> 6, single = (fhd >> 5) & 1, did_flag = fhd & 3;
                    if (fhd & 8) err = E_CORRUPT;
                    S.has_cksum = (fhd >> 2) & 1;
                    uint32_t fcs_bytes = fcs_flag == 0 ? single : (1u << fcs_flag);
                    uint32_t did_bytes = did_flag == 3 ? 4 : did_flag;
                    pos = 5;
                    if (src_n < pos + (single ? 0 : 1) + did_bytes + fcs_bytes) err = E_TRUNC;
                    if (!err) {
                        if (!single) pos++;  // window descriptor: matches are bounded by the frame start below
                        uint32_t did = 0;
                        for (uint32_t i = 0; i < did_bytes; i++) did |= (uint32_t)src[pos + i] << (8 * i);
                        pos += did_bytes;
                        if (did) err = E_UNSUP;
                        uint64_t fcs = 0;
                        for (uint32_t i = 0; i < fcs_bytes; i++) fcs |= (uint64_t)src[pos + i] << (8 * i);
                        pos += fcs_bytes;
                        if (fcs_bytes == 2) fcs += 256;
                        if (fcs_bytes == 0) err = err ? err : E_UNSUP;  // zl_get_decompressed_size would fail
                        S.content_size = fcs;
                        if (!err && fcs != a.usize[row]) err = E_CORRUPT;  // index row and frame disagree
                        if (!err && a.out_off[row] + fcs > a.out_cap) err = E_DST;
                        if (!err && fcs >= 0xFFFFFFFFull) err = E_UNSUP;
                    }
                }
            }
            S.src_pos = pos;
            S.src_end = src_n;
            S.out_end = S.content_size;
            S.err = err;
            S.blk_last = 0;
        }
        __syncthreads();

        // ---- blocks ----
        for (;;) {
            const bool stop = S.err != 0 || S.blk_last != 0;
            __syncthreads();  // every lane has read the loop state before lane 0 rewrites it
            if (stop) break;
            if (tid == 0) {
                uint64_t p = S.src_pos;
                if (p + 3 > S.src_end) S.err = E_TRUNC;
                else {
                    uint32_t bh = src[p] | (src[p + 1] << 8) | (src[p + 2] << 16);
                    S.blk_last = a.block_mode ? 1u : (bh & 1); S.blk_type = (bh >> 1) & 3; S.blk_size = bh >> 3;  // a block item ends after its block
                    S.src_pos = p + 3;
                    uint32_t need = S.blk_type == 1 ? 1 : S.blk_size;
                    if (S.blk_type == 3) S.err = E_CORRUPT;
                    else if (S.src_pos + need > S.src_end) S.err = E_TRUNC;
                    else if (S.blk_type != 2 && S.out_pos + S.blk_size > S.out_end) S.err = E_CORRUPT;
                    else if (S.blk_type == 2 && S.blk_size > BLOCK_MAX) S.err = E_CORRUPT;
                }
            }
            __syncthreads();
            if (S.err) break;
            const uint32_t btype = S.blk_type, bsize = S.blk_size;
            const uint64_t bpos = S.src_pos;
            if (btype == 0) {
                coop_copy(out + S.out_pos, src + bpos, bsize, tid, NT);
                __syncthreads();
                if (tid == 0) { S.out_pos += bsize; S.src_pos += bsize; S.hist_n = 0; }
                __syncthreads();
                continue;
            }
            if (btype == 1) {
                coop_fill(out + S.out_pos, src[bpos], bsize, tid, NT);
                __syncthreads();
                if (tid == 0) { S.out_pos += bsize; S.src_pos += 1; S.hist_n = 0; }
                __syncthreads();
                continue;
            }

            // ======== compressed block ========
            const uint8_t *const bsrc = src + bpos;
            // -- literals header + Huffman tree (lane 0) --
            __syncthreads();
            if (tid == 0) {
                int err = 0;
                uint32_t n = bsize, used = 0;
                S.tree_n = 0;
                if (n < 1) err = E_TRUNC;
                else {
                    uint32_t b0 = bsrc[0], type = b0 & 3, sf = (b0 >> 2) & 3, regen = 0, comp = 0, hdr = 0;
                    if (type <= 1) {
                        if ((sf & 1) == 0) { regen = b0 >> 3; hdr = 1; }
                        else if (sf == 1) { if (n < 2) err = E_TRUNC; else { regen = (b0 >> 4) + ((uint32_t)bsrc[1] << 4); hdr = 2; } }
                        else { if (n < 3) err = E_TRUNC; else { regen = (b0 >> 4) + ((uint32_t)bsrc[1] << 4) + ((uint32_t)bsrc[2] << 12); hdr = 3; } }
                        if (!err && regen > BLOCK_MAX) err = E_CORRUPT;
                        if (!err) {
                            if (type == 0) {
                                if (hdr + regen > n) err = E_TRUNC;
                                S.lit_kind = 0; S.lit_src = bpos + hdr; used = hdr + regen;
                            } else {
                                if (hdr + 1 > n) err = E_TRUNC;
                                else { S.lit_kind = 1; S.lit_rle = bsrc[hdr]; used = hdr + 1; }
                            }
                            S.lit_len = regen; S.n_streams = 0;
                        }
                    } else {
                        uint64_t h = 0;
                        for (uint32_t i = 0; i < 5 && i < n; i++) h |= (uint64_t)bsrc[i] << (8 * i);
                        uint32_t streams;
                        if (sf == 0) { streams = 1; regen = (h >> 4) & 0x3FF; comp = (h >> 14) & 0x3FF; hdr = 3; }
                        else if (sf == 1) { streams = 4; regen = (h >> 4) & 0x3FF; comp = (h >> 14) & 0x3FF; hdr = 3; }
                        else if (sf == 2) { streams = 4; regen = (h >> 4) & 0x3FFF; comp = (h >> 18) & 0x3FFF; hdr = 4; }
                        else { streams = 4; regen = (h >> 4) & 0x3FFFF; comp = (h >> 22) & 0x3FFFF; hdr = 5; }
                        if (hdr + comp > n) err = E_TRUNC;
                        else if (regen > BLOCK_MAX) err = E_CORRUPT;
                        uint32_t p = hdr, remain = comp;
                        if (!err && type == 2) {
                            // the tree description: its size is in its first byte; wave 0 reads it after this pass
                            uint32_t tu = 0;
                            if (remain < 1) err = E_TRUNC;
                            else {
                                const uint32_t hb = bsrc[p];
                                tu = hb >= 128 ? 1 + (hb - 127 + 1) / 2 : 1 + hb;
                                if (hb == 0 || tu > remain) err = E_TRUNC;
                            }
                            if (!err) { S.tree_off = p; S.tree_n = remain; p += tu; remain -= tu; S.huf_valid = 2; }  // 2 = table must be (re)filled
                        } else if (!err && !S.huf_valid) err = E_CORRUPT;
                        if (!err) {
                            if (streams == 1) {
                                S.stream_off[0] = p; S.stream_len[0] = remain; S.stream_out[0] = 0; S.stream_n[0] = regen;
                            } else {
                                uint32_t seg = (regen + 3) / 4;
                                if (remain < 6 || 3 * seg > regen) err = E_CORRUPT;
                                else {
                                    uint32_t s1 = bsrc[p] | (bsrc[p + 1] << 8), s2 = bsrc[p + 2] | (bsrc[p + 3] << 8),
                                             s3 = bsrc[p + 4] | (bsrc[p + 5] << 8);
                                    if (6 + s1 + s2 + s3 > remain) err = E_CORRUPT;
                                    else {
                                        uint32_t s4 = remain - 6 - s1 - s2 - s3;
                                        p += 6;
                                        S.stream_off[0] = p; S.stream_len[0] = s1; S.stream_out[0] = 0; S.stream_n[0] = seg;
                                        S.stream_off[1] = p + s1; S.stream_len[1] = s2; S.stream_out[1] = seg; S.stream_n[1] = seg;
                                        S.stream_off[2] = p + s1 + s2; S.stream_len[2] = s3; S.stream_out[2] = 2 * seg; S.stream_n[2] = seg;
                                        S.stream_off[3] = p + s1 + s2 + s3; S.stream_len[3] = s4; S.stream_out[3] = 3 * seg; S.stream_n[3] = regen - 3 * seg;
                                    }
                                }
                            }
                            S.n_streams = streams; S.lit_kind = 2; S.lit_len = regen; used = hdr + comp;
                        }
                    }
                }
                S.err = err;
                S.lit_pos = 0;
                S.src_pos = bpos + used;  // now at the sequences section
            }
            __syncthreads();
            if (S.err) break;
            if (S.tree_n) {
                if (wave0) {
                    const int rc = huf_read_tree_wave(S, bsrc + S.tree_off, S.tree_n, blob_end, tid);
                    if (rc && tid == 0) S.err = rc;
                }
                __syncthreads();
                if (S.err) break;
            }
            DSTAMP(6);
            if (S.n_streams) {
                if (S.huf_valid == 2) {
                    // fill the decoding table: lane = symbol, each writes its canonical range
                    for (uint32_t sym = tid; sym < 256; sym += NT) {
                        uint32_t len = S.sym_len[sym];
                        if (len) {
                            uint32_t st = S.sym_start[sym];
                            uint16_t e = (uint16_t)(sym | ((S.huf_log + 1 - S.weights[sym]) << 8));
                            for (uint32_t i = 0; i < len; i++) S.huf[st + i] = e;
                        }
                    }
                    __syncthreads();
                    if (tid == 0) S.huf_valid = 1;
                }
                DSTAMP(7);
                if (tid < S.n_streams) {
                    // (four symbols per 8-byte load while 64 or more bits are unread: the foreign-frame path's stream decoder)
                    int rc = fz_huf_stream(S.huf, S.huf_log, bsrc + S.stream_off[tid], S.stream_len[tid], blob_end,
                                           lit_buf + S.stream_out[tid], S.stream_n[tid]);
                    if (rc) atomicMin(&S.err, rc);
                }
                __syncthreads();  // literal bytes visible to the whole workgroup (same CU)
                if (S.err) break;
            }

            // -- sequences header + tables (lane 0) --
            __syncthreads();
            DSTAMP(1);
            if (tid == 0) {
                int err = 0;
                const uint64_t end = bpos + bsize;
                uint64_t p = S.src_pos;
                uint32_t nseq = 0;
                S.bld[0] = S.bld[1] = S.bld[2] = 0;
                if (p >= end) err = E_TRUNC;
                else {
                    uint32_t b0 = src[p];
                    if (b0 == 0) { nseq = 0; p += 1; }
                    else if (b0 < 128) { nseq = b0; p += 1; }
                    else if (b0 < 255) { if (p + 2 > end) err = E_TRUNC; else { nseq = ((b0 - 128) << 8) + src[p + 1]; p += 2; } }
                    else { if (p + 3 > end) err = E_TRUNC; else { nseq = src[p + 1] + ((uint32_t)src[p + 2] << 8) + 0x7F00; p += 3; } }
                }
                if (!err && nseq) {
                    if (p >= end) err = E_TRUNC;
                    else {
                        uint32_t modes = src[p++];
                        if (modes & 3) err = E_CORRUPT;
                        const int shifts[3] = {6, 4, 2};
                        const int kinds[3] = {K_LL, K_OF, K_ML};
                        const int maxlog[3] = {9, 8, 9};
                        const int maxsym[3] = {35, 31, 52};
                        const int deflog[3] = {6, 5, 6};
                        for (int k = 0; k < 3 && !err; k++) {
                            uint32_t mode = (modes >> shifts[k]) & 3;
                            if (mode == 0) { S.sel[k] = 0; S.log_[k] = deflog[k]; S.valid[k] = 1; }
                            else if (mode == 1) {
                                if (p >= end) err = E_TRUNC;
                                else { err = fse_set_rle(&S.rle[k], src[p++], kinds[k]); S.sel[k] = 1; S.log_[k] = 0; S.valid[k] = 1; }
                            } else if (mode == 2) {
                                int nsym, log;
                                uint32_t used;
                                err = fse_read_ncount(S, src + p, (uint32_t)(end - p), maxlog[k], maxsym[k], &nsym, &log, &used, 64 * k);
                                if (!err) { p += used; S.sel[k] = 2; S.log_[k] = log; S.valid[k] = 1; S.bld[k] = (uint32_t)nsym; }
                            } else if (!S.valid[k]) err = E_CORRUPT;
                        }
                        if (!err) {
                            if (p >= end) err = E_TRUNC;
                            else {
                                BitR b;
                                if (!b.init(src + p, (uint32_t)(end - p), blob_end)) err = E_CORRUPT;
                                else {
                                    S.st_ll = b.read(S.log_[0]);
                                    S.st_of = b.read(S.log_[1]);
                                    S.st_ml = b.read(S.log_[2]);
                                    S.bs_pos = b.pos;
                                    S.bs_off = (uint32_t)p;
                                }
                            }
                        }
                    }
                } else if (!err && p != end) err = E_CORRUPT;
                S.nseq = nseq;
                S.err = err;
                S.src_pos = end;
            }
            __syncthreads();
            if (S.err) break;
            {   // the tables this block describes: wave k builds table k (LL, OF, ML) from the counts lane 0 left in S.norm
                const uint32_t w = tid >> 6;
                if (w < 3 && S.bld[w])
                    fse_build_wave(S.norm + 64 * w, S.bld[w], S.log_[w], w == 0 ? K_LL : (w == 1 ? K_OF : K_ML),
                                   w == 0 ? S.ll : (w == 1 ? S.of : S.ml), reinterpret_cast<uint16_t *>(S.fse_next) + 128 * w, tid & 63);
                __syncthreads();
            }

            DSTAMP(2);
            // -- sequences: lane 0 decodes a batch into LDS, the workgroup executes it --
            uint32_t seq_done = 0;
            const uint32_t nseq = S.nseq;
            const uint8_t *lit_ptr = S.lit_kind == 0 ? src + S.lit_src : lit_buf;
            // Decode one batch of sequences [seq_done, seq_done + bn) into buffer `buf` (wave 0).  abs0_in: output bytes in front of
            // the batch that a match may reach (inside the block for a block item), lit_room_in: literals still unused.
            auto decode_batch = [&](const uint32_t seq_done, const uint32_t bn, const uint32_t buf, const uint64_t abs0_in, const uint32_t lit_room_in,
                                    uint64_t &prod_out, uint32_t &lits_out) {
                    const uint32_t sel0 = uni(S.sel[0]), sel1 = uni(S.sel[1]), sel2 = uni(S.sel[2]);
                    const FseEntry *tl = sel0 == 0 ? S.dll : (sel0 == 1 ? &S.rle[0] : S.ll);
                    const FseEntry *to = sel1 == 0 ? S.dof : (sel1 == 1 ? &S.rle[1] : S.of);
                    const FseEntry *tm = sel2 == 0 ? S.dml : (sel2 == 1 ? &S.rle[2] : S.ml);
                    // Two stages per group of 64 sequences (one wave gets one issue slot every 4 cycles, so the serial chain carries
                    // nothing it does not have to; the same scheme as k_fz_entropy):
                    //   A  the three states walk through their tables and the bit position moves; the wave notes (states, position)
                    //      in lane (i mod 64).  The only bits it extracts are the next states', out of 512 bytes of the stream
                    //      kept in its registers (lane k holds bytes [wbase + 8k, +8): two readlane pairs and a funnel shift);
                    //   B  lane = sequence: table entries again, the 16 stream bytes that end at the lane's position, the three
                    //      values; then repeat offsets (in order only for groups that use one) and the literal / match bounds.
                    // Bytes in front of the stream read as zero; a sequence that needs more bits than are left is the corruption test.
                    const uint8_t *const bbase = src + uni(S.bs_off);
                    int32_t left = (int32_t)uni((uint32_t)S.bs_pos);  // a block's bitstream is < 2^20 bits
                    uint64_t wq = 0;
                    int32_t wbits = INT32_MAX;  // bit position of the end of the window's first 8 bytes (none loaded yet)
                    uint32_t sl = uni(S.st_ll), so = uni(S.st_of), sm = uni(S.st_ml);
                    uint32_t r0 = uni(S.rep[0]), r1 = uni(S.rep[1]), r2 = uni(S.rep[2]);
                    int err = 0;
                    uint64_t produced = 0;
                    uint32_t lits = 0;
                    const uint32_t lit_room = lit_room_in;
                    const uint64_t abs0 = abs0_in;  // bytes a match may reach back over
                    const uint64_t out_end = uni64(S.out_end);
                    const uint2 *const tl2 = reinterpret_cast<const uint2 *>(tl), *const to2 = reinterpret_cast<const uint2 *>(to),
                                *const tm2 = reinterpret_cast<const uint2 *>(tm);
                    const bool lane0 = (tid & 63) == 0;
                    const bool predef = sel0 == 0 && sel1 == 0 && sel2 == 0;
                    const uint32_t rlx = tl2[predef ? (tid & 63) : 0].x, rmx = tm2[predef ? (tid & 63) : 0].x, rox = to2[predef ? (tid & 31) : 0].x;
                    auto rdl = [](uint32_t v, uint32_t l) -> uint32_t { return (uint32_t)__builtin_amdgcn_readlane((int)v, (int)l); };
                    const uint32_t lane = tid & 63;
                    for (uint32_t g0 = 0; g0 < bn && !err; g0 += 64) {
                        const uint32_t cnt = bn - g0 < 64 ? bn - g0 : 64;
                        uint32_t my_so = 0, my_sm = 0, my_sl = 0, my_leftu = 0;
                        int32_t margin = 0;
                        // ---- A ----
                        // (the block's last sequence takes no state bits: it is handled behind the loop, which so has no such case)
                        const bool has_last = seq_done + g0 + cnt == nseq;
                        const uint32_t cnt_a = has_last ? cnt - 1 : cnt;
                        for (uint32_t g = 0; g < cnt_a; g++) {
                            // entry.x = next:16 | nbits:8 | addbits:8 (predefined tables sit in registers, one entry per lane)
                            uint32_t eox, emx, elx;
                            if (predef) { eox = rdl(rox, so); emx = rdl(rmx, sm); elx = rdl(rlx, sl); }
                            else { const uint32_t vo_ = to2[so].x, vm_ = tm2[sm].x, vl_ = tl2[sl].x; eox = uni(vo_); emx = uni(vm_); elx = uni(vl_); }  // three reads in flight, then the waits
                            const uint32_t need_v = (eox >> 24) + (emx >> 24) + (elx >> 24);
                            const uint32_t nbl = (elx >> 16) & 0xFF, nbm = (emx >> 16) & 0xFF, nbo = (eox >> 16) & 0xFF, need_s = nbl + nbm + nbo;
                            margin = min(margin, left - (int32_t)(need_v + need_s));  // below zero: the stream ends before a sequence does
                            wrlane4_u(my_so, my_sm, my_sl, my_leftu, so, sm, sl, (uint32_t)left, g);
                            // the 64 stream bits that end where the state bits end, straight out of the window: `off` is their first
                            // bit counted from the window's
                            const int32_t pos0 = left - (int32_t)need_v, pos = pos0 < 0 ? 0 : pos0;
                            int32_t off = pos - wbits;
                            if (off < 0) {
                                const int32_t nb0 = ((((pos + 7) >> 3) - 504) & ~7), wbase = nb0 < -8 ? -8 : nb0;  // bytes in front of the stream read as zero
                                const int32_t o8 = wbase + 8 * (int32_t)lane;
                                wq = o8 < 0 ? 0ull : load8_guard(bbase + o8, blob_end);
                                wbits = 8 * wbase + 64;
                                off = pos - wbits;
                            }
                            const uint32_t j = uni((uint32_t)off >> 6), r = uni((uint32_t)off & 63);
                            const uint64_t q0 = rdlane64_u(wq, j), q1 = rdlane64_u(wq, j + 1);
                            const uint64_t v64 = (q0 >> r) | ((q1 << 1) << (63 - r));
                            const uint32_t xs = (uint32_t)((v64 >> 1) >> (63 - need_s));  // its top need_s (<= 27) bits
                            sl = (elx & 0xFFFF) + (xs >> (nbm + nbo));
                            sm = (emx & 0xFFFF) + ((xs >> nbo) & ((1u << nbm) - 1u));
                            so = (eox & 0xFFFF) + (xs & ((1u << nbo) - 1u));
                            left -= (int32_t)(need_v + need_s);
                        }
                        if (has_last) {
                            uint32_t eox, emx, elx;
                            if (predef) { eox = rdl(rox, so); emx = rdl(rmx, sm); elx = rdl(rlx, sl); }
                            else { const uint32_t vo_ = to2[so].x, vm_ = tm2[sm].x, vl_ = tl2[sl].x; eox = uni(vo_); emx = uni(vm_); elx = uni(vl_); }
                            const uint32_t need_v = (eox >> 24) + (emx >> 24) + (elx >> 24);
                            margin = min(margin, left - (int32_t)need_v);
                            wrlane4_u(my_so, my_sm, my_sl, my_leftu, so, sm, sl, (uint32_t)left, cnt - 1);
                            left -= (int32_t)need_v;
                        }
                        if (margin < 0) { err = E_CORRUPT; break; }
                        // ---- B ----
                        const bool on = lane < cnt;
                        uint32_t ov = 4, ml = 0, ll = 0;
                        if (on) {
                            const uint2 eo = to2[my_so], em = tm2[my_sm], el = tl2[my_sl];
                            const uint32_t ofb = eo.x >> 24, mlb = em.x >> 24, llb = el.x >> 24, need_v = ofb + mlb + llb;
                            const int32_t my_left = (int32_t)my_leftu, bend = (my_left + 7) >> 3;
                            uint64_t lo8 = 0, hi8 = 0;  // stream bytes [bend - 16, bend - 8) and [bend - 8, bend)
                            if (bend >= 16) {
                                __builtin_memcpy(&lo8, bbase + bend - 16, 8);
                                __builtin_memcpy(&hi8, bbase + bend - 8, 8);
                            } else {
                                for (int32_t k = 0; k < 16; k++) {
                                    const int32_t o = bend - 16 + k;
                                    const uint64_t byte = o >= 0 ? bbase[o] : 0;
                                    if (k < 8) lo8 |= byte << (8 * k); else hi8 |= byte << (8 * (k - 8));
                                }
                            }
                            const uint32_t al = (uint32_t)(8 * bend - my_left);
                            const uint64_t H = (hi8 << al) | ((lo8 >> 1) >> (63 - al));
                            const uint64_t xv = (H >> 1) >> (63 - need_v);  // need_v <= 63
                            ov = eo.y + (uint32_t)(xv >> (mlb + llb));
                            ml = em.y + ((uint32_t)(xv >> llb) & ((1u << mlb) - 1u));
                            ll = el.y + ((uint32_t)xv & ((1u << llb) - 1u));
                            S.seq_ll[buf][g0 + lane] = ll;
                            S.seq_ml[buf][g0 + lane] = ml;
                        }
                        uint32_t offset = ov - 3;
                        if (__ballot(on && ov <= 3) != 0ull || cnt < 3) {
                            // A block item starts with an undefined history (all 0): a repeat code is good once the block's
                            // own sequences have defined the entry it names (this build's higher effort tier writes such
                            // blocks); one that reaches an undefined entry depends on the block before — the serial pass decides.
                            for (uint32_t j = 0; j < cnt; j++) {  // in order, on the scalar unit
                                const uint32_t ovj = rdlane_u(ov, j), llj = rdlane_u(ll, j);
                                uint32_t o;
                                if (ovj > 3) { o = ovj - 3; r2 = r1; r1 = r0; r0 = o; }
                                else {
                                    const uint32_t idx = ovj - 1 + (llj == 0 ? 1 : 0);
                                    if (idx == 0) { o = r0; if (o == 0) { err = E_CORRUPT; break; } }
                                    else {
                                        o = idx == 1 ? r1 : (idx == 2 ? r2 : (r0 ? r0 - 1 : 0));
                                        if (o == 0) { err = E_CORRUPT; break; }
                                        if (idx > 1) r2 = r1;
                                        r1 = r0; r0 = o;
                                    }
                                }
                                if (lane == j) offset = o;
                            }
                            if (err) break;
                        } else {  // explicit offsets only: the history is the group's last three
                            r0 = rdlane_u(offset, cnt - 1); r1 = rdlane_u(offset, cnt - 2); r2 = rdlane_u(offset, cnt - 3);
                        }
                        uint32_t linc = ll, pinc = ll + ml;  // inclusive scans: literals used, bytes produced
linc = wave_incl_scan(linc); pinc = wave_incl_scan(pinc);
                        // a match may reach back over what exists when it starts; literals may not run out
                        const bool bad = on && ((uint64_t)offset > abs0 + produced + pinc - ml || lits + linc > lit_room);
                        if (__ballot(bad) != 0ull) { err = E_CORRUPT; break; }
                        if (on) S.seq_off[buf][g0 + lane] = offset;
                        lits += rdlane_u(linc, 63);
                        produced += rdlane_u(pinc, 63);
                    }
                    if (!err && abs0 + uni64(S.blk_base) + produced > out_end) err = E_CORRUPT;
                    if (!err && seq_done + bn == nseq && left != 0) err = E_CORRUPT;
                    if (lane0) {
                        S.bs_pos = (int64_t)left;
                        S.st_ll = sl; S.st_of = so; S.st_ml = sm;
                        S.rep[0] = r0; S.rep[1] = r1; S.rep[2] = r2;
                        S.err = err;
                        S.batch_n = bn;
                    }
                    prod_out = produced;
                    lits_out = lits;
                            };
            if constexpr (NWAVES == 4) {
                // Execute one decoded batch.  solo: wave 1 alone, no barrier inside (wave 0 is decoding the next batch meanwhile);
                // otherwise the whole workgroup follows the same control flow (decisions depend only on the batch in LDS and on
                // counters every wave mirrors), the worker wave does the window work and all waves share the long direct copies.
                auto exec_batch = [&](const uint32_t buf, const uint32_t bn, const bool solo) {
                        uint64_t opos = S.out_pos;
                        uint32_t win_n = S.win_n, hist_n = S.hist_n, lpos = S.lit_pos;
                        const bool rle_lits = S.lit_kind == 1;
                        const uint8_t rle_byte = (uint8_t)S.lit_rle;
                        const uint32_t lane = tid & 63;
                        const bool wk = solo ? (tid >> 6) == 1 : wave0;  // who does the window work
                        const uint32_t ct = solo ? lane : tid;           // thread index inside cooperative copies
                        uint8_t *const W = S.ebuf;
                        bool dirty = false;
                        uint32_t si = 0;
                        while (si < bn) {
                            const uint32_t idx = si + lane;
                            const bool v = idx < bn;
                            const uint32_t ll = v ? S.seq_ll[buf][idx] : 0, ml = v ? S.seq_ml[buf][idx] : 0, off = v ? S.seq_off[buf][idx] : 1;
                            const uint32_t tot = ll + ml;
                            const uint64_t bigm = __ballot(v && (tot > WIN_SEQ_MAX || ll > WIN_SEQ_MAX));
                            const uint32_t nv = bn - si < 64 ? bn - si : 64;
                            const uint32_t ncand = bigm ? (uint32_t)__ffsll((long long)bigm) - 1 : nv;
                            if (ncand == 0) {
                                // ---- one long sequence, straight to HBM (the window is emptied first) ----
                                if (win_n) {
                                    if (wk) (void)win_flush(W, out, opos, win_n, hist_n, lane, false);
                                    opos += win_n;
                                    win_n = 0;
                                    if (!solo) __syncthreads();
                                }
                                hist_n = 0;
                                const uint32_t ll0 = S.seq_ll[buf][si], ml0 = S.seq_ml[buf][si], off0 = S.seq_off[buf][si];
                                if (ll0) {
                                    const bool big = !solo && ll0 >= BIG_COPY;
                                    if (big) __syncthreads();  // big implies !solo
                                    if (big || wk) {
                                        const uint32_t nt = big ? NT : 64;
                                        if (rle_lits) coop_fill(out + opos, rle_byte, ll0, ct, nt);
                                        else coop_copy(out + opos, lit_ptr + lpos, ll0, ct, nt);
                                    }
                                    if (big) __syncthreads();  // big implies !solo
                                    opos += ll0; lpos += ll0;
                                }
                                {
                                    const bool big = !solo && ml0 >= BIG_COPY;
                                    if (big) __syncthreads();  // big implies !solo
                                    else if (wk) wave_mem_sync();
                                    if (big || wk) coop_match<NWAVES>(out + opos, off0, ml0, ct, big, S.ebuf);
                                    if (big) __syncthreads();  // big implies !solo
                                    opos += ml0;
                                }
                                dirty = true;  // direct stores may still be in flight: whoever reads HBM back drains first
                                si++;
                                continue;
                            }
                            // ---- up to 64 short sequences inside the window ----
                            uint32_t end = lane < ncand ? tot : 0, lend = lane < ncand ? ll : 0;
                end = wave_incl_scan(end); lend = wave_incl_scan(lend);
                            uint32_t fit = (uint32_t)__popcll(__ballot(lane < ncand && end <= WIN_CAP - win_n));
                            if (fit == 0) {  // chunk full: stream it out, keep the newest bytes as history
                                uint32_t h = 0;
                                if (wk) h = win_flush(W, out, opos, win_n, hist_n, lane, true);
                                (void)h;
                                hist_n = hist_n + win_n < WIN_HIST ? hist_n + win_n : WIN_HIST;
                                opos += win_n;
                                win_n = 0;
                                fit = (uint32_t)__popcll(__ballot(lane < ncand && end <= WIN_CAP));
                            }
                            if (dirty) {  // far matches / the history read-back below must see every direct store
                                wave_mem_sync();
                                if (!solo) __syncthreads();
                                dirty = false;
                            }
                            const uint32_t want_h = opos < WIN_HIST ? (uint32_t)opos : WIN_HIST;
                            if (win_n == 0 && hist_n < want_h) {
                                // history lost to a direct copy / raw block: read the newest output back (it has landed:
                                // every direct write above ends with a drain + barrier)
                                if (wk) {
                                    coop_copy(W + WIN_HIST - want_h, out + opos - want_h, want_h, lane, 64);
                                }
                                hist_n = want_h;
                            }
                            if (wk) {
                                const bool on = lane < fit;
                                win_exec_group(W, out, opos, hist_n, lane, on, WIN_HIST + win_n + (end - tot), ll, ml, off,
                                               lit_ptr + lpos + (lend - ll), rle_lits, rle_byte);
                            }
                            win_n += rdlane_u(end, fit - 1);
                            lpos += rdlane_u(lend, fit - 1);
                            si += fit;
                        }
                        if (dirty) wave_mem_sync();  // with the barrier below: direct stores of this batch have landed
                        if (!solo) __syncthreads();
                        if (wk && lane == 0) { S.out_pos = opos; S.win_n = win_n; S.hist_n = hist_n; S.lit_pos = lpos; }
                                    };
                // wave 0 decodes batch k+1 while wave 1 executes batch k; the last batch of a block is executed by everyone
                uint64_t d_abs0 = S.out_pos + S.win_n - S.blk_base;  // the decoder's own running totals
                uint32_t d_room = S.lit_len - S.lit_pos;
                uint32_t bn = nseq < SEQ_BATCH ? nseq : SEQ_BATCH, buf = 0;
                __syncthreads();
                if (wave0 && bn) {
                    uint64_t p_ = 0; uint32_t l_ = 0;
                    decode_batch(0, bn, 0, d_abs0, d_room, p_, l_);
                    d_abs0 += p_; d_room -= l_ <= d_room ? l_ : d_room;
                }
                __syncthreads();
                DSTAMP(3);
                while (bn && !S.err) {
                    const uint32_t next0 = seq_done + bn;
                    const uint32_t nbn = nseq - next0 < SEQ_BATCH ? nseq - next0 : SEQ_BATCH;
                    if (nbn) {
                        if (wave0) {
                            uint64_t p_ = 0; uint32_t l_ = 0;
                            decode_batch(next0, nbn, buf ^ 1, d_abs0, d_room, p_, l_);
                            d_abs0 += p_; d_room -= l_ <= d_room ? l_ : d_room;
                        } else if ((tid >> 6) == 1) exec_batch(buf, bn, true);
                    } else exec_batch(buf, bn, false);
                    __syncthreads();
                    DSTAMP(4);
                    seq_done = next0; bn = nbn; buf ^= 1;
                }
            } else {
                while (seq_done < nseq) {
                    const uint32_t bn = nseq - seq_done < SEQ_BATCH ? nseq - seq_done : SEQ_BATCH;
                    __syncthreads();
                    if (wave0) {
                        uint64_t p_ = 0; uint32_t l_ = 0;
                        decode_batch(seq_done, bn, 0, uni64(S.out_pos) + uni(S.win_n) - uni64(S.blk_base), uni(S.lit_len) - uni(S.lit_pos), p_, l_);
                    }
                    __syncthreads();
                    DSTAMP(3);
                    if (S.err) break;
                    // execute: wave 0 walks the batch in order; long copies are shared by all waves
                    {
                        uint64_t opos = S.out_pos;
                        uint32_t lpos = S.lit_pos;
                        const bool rle_lits = S.lit_kind == 1;
                        const uint8_t rle_byte = (uint8_t)S.lit_rle;
                        for (uint32_t i = 0; i < bn; i++) {
                            const uint32_t ll = S.seq_ll[0][i], ml = S.seq_ml[0][i], off = S.seq_off[0][i];
                            if (ll) {
                                const bool big = NWAVES > 1 && ll >= BIG_COPY;
                                if (big) __syncthreads();
                                if (big || wave0) {
                                    const uint32_t nt = big ? NT : 64;
                                    if (rle_lits) coop_fill(out + opos, rle_byte, ll, tid, nt);
                                    else coop_copy(out + opos, lit_ptr + lpos, ll, tid, nt);
                                }
                                if (big) __syncthreads();
                                opos += ll; lpos += ll;
                            }
                            {
                                const bool big = NWAVES > 1 && ml >= BIG_COPY;
                                if (big) __syncthreads();
                                else if (wave0) wave_mem_sync();
                                if (big || wave0) coop_match<NWAVES>(out + opos, off, ml, tid, big, S.ebuf);
                                if (big) __syncthreads();
                                opos += ml;
                            }
                        }
                        __syncthreads();
                        if (tid == 0) { S.out_pos = opos; S.lit_pos = lpos; }
                    }
                                    seq_done += bn;
                    __syncthreads();
                    DSTAMP(4);
                }
            }
            if (S.err) break;
            if constexpr (NWAVES == 4) {
                // -- literals left after the last sequence, then the window goes out (history stays for the next block) --
                if (S.lit_len != S.lit_pos || S.win_n != 0) {
                    __syncthreads();
                    const uint32_t rest = S.lit_len - S.lit_pos;
                    const uint32_t win_n = S.win_n, hist_n = S.hist_n;
                    const uint64_t opos = S.out_pos;
                    const uint8_t *lit_ptr2 = S.lit_kind == 0 ? src + S.lit_src : lit_buf;
                    if (opos + win_n + rest > S.out_end) { __syncthreads(); if (tid == 0) S.err = E_CORRUPT; }
                    else {
                        const bool in_win = rest <= WIN_CAP - win_n;
                        uint32_t h = hist_n;
                        if (wave0) {
                            const uint32_t lane = tid & 63;
                            if (in_win && rest) {
                                uint8_t *d = S.ebuf + WIN_HIST + win_n;
                                if (S.lit_kind == 1) for (uint32_t i = lane; i < rest; i += 64) d[i] = (uint8_t)S.lit_rle;
                                else coop_copy(d, lit_ptr2 + S.lit_pos, rest, lane, 64);
                            }
                            h = win_flush(S.ebuf, out, opos, in_win ? win_n + rest : win_n, hist_n, lane, in_win);
                        }
                        __syncthreads();
                        if (!in_win) {  // long tail: straight to HBM by everyone, history is read back on demand
                            if (S.lit_kind == 1) coop_fill(out + opos + win_n, (uint8_t)S.lit_rle, rest, tid, NT);
                            else coop_copy(out + opos + win_n, lit_ptr2 + S.lit_pos, rest, tid, NT);
                            wave_mem_sync();
                            h = 0;
                        }
                        __syncthreads();
                        if (tid == 0) { S.out_pos = opos + win_n + rest; S.win_n = 0; S.hist_n = in_win ? h : 0; S.lit_pos += rest; }
                    }
                    __syncthreads();
                }
        
            } else {
                // -- literals left after the last sequence --
                {
                    const uint32_t rest = S.lit_len - S.lit_pos;
                    if (S.out_pos + rest > S.out_end) { if (tid == 0) S.err = E_CORRUPT; }
                    else if (rest) {
                        if (S.lit_kind == 1) coop_fill(out + S.out_pos, (uint8_t)S.lit_rle, rest, tid, NT);
                        else coop_copy(out + S.out_pos, lit_ptr + S.lit_pos, rest, tid, NT);
                    }
                    __syncthreads();
                    if (tid == 0) S.out_pos += rest;
                    __syncthreads();
                }
        
            }
        }
        DSTAMP(5);
        __syncthreads();  // every wave's output stores have landed (same CU)
        {
            int err = S.err;
            if (!err && S.out_pos != (a.block_mode ? S.out_end : S.content_size)) err = E_CORRUPT;
            if (!err && S.has_cksum) {
                if (S.src_pos + 4 > S.src_end) err = E_TRUNC;
                else if (wave0) {  // frame content checksum: low 32 bits of XXH64(content)
                    const uint64_t h = wave_xxh64(out, S.content_size, tid);
                    const uint8_t *c = src + S.src_pos;
                    const uint32_t want = c[0] | (c[1] << 8) | (c[2] << 16) | ((uint32_t)c[3] << 24);
                    if ((uint32_t)h != want) err = -7;  // ZNIPPY_E_CHECKSUM
                }
            }
            if (tid == 0) {
                if (a.block_mode) { if (err) atomicOr(&a.row_flag[row], 1u); }  // any trouble: the serial pass decides
                else a.status[row] = err ? err : 2;  // 2 = decoded here, to be hashed by the second pass
            }
        }
        __syncthreads();
    }
}

// ---------------------------------------------------------------------------------------------
// block items: header scan (one thread per candidate frame) and the per-row verdict afterwards
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void k_scan_blocks(BlockScanArgs a) {
    // one wave per candidate frame; every lane runs the frame-header part on the same addresses (broadcast loads)
    const uint32_t c = blockIdx.x, lane = threadIdx.x;
    if (c >= a.n_cand) return;
    const uint32_t row = a.cand_row[c], base = a.cand_base[c], nb = a.cand_nblocks[c];
    const uint8_t *src = a.blobs + (a.blob_off[row] - a.blob_base);
    const uint64_t n = (a.preset && a.status[row] < 0) ? 0 : a.blob_size[row], fcs_want = a.usize[row];  // host verdict: not a candidate
    bool ok = n >= 6 && (src[0] | (src[1] << 8) | (src[2] << 16) | ((uint32_t)src[3] << 24)) == 0xFD2FB528u;
    uint64_t pos = 5;
    if (ok) {
        const uint32_t fhd = src[4];
        const uint32_t fcs_flag = fhd >> 6, single = (fhd >> 5) & 1, did_flag = fhd & 3;
        const uint32_t fcs_bytes = fcs_flag == 0 ? single : (1u << fcs_flag);
        ok = !(fhd & 8) && !((fhd >> 2) & 1) && did_flag == 0 && fcs_bytes != 0;  // no checksum trailer, no dictionary
        if (ok) {
            if (!single) pos++;
            ok = pos + fcs_bytes <= n;
            uint64_t fcs = 0;
            for (uint32_t i = 0; ok && i < fcs_bytes; i++) fcs |= (uint64_t)src[pos + i] << (8 * i);
            if (fcs_bytes == 2) fcs += 256;
            pos += fcs_bytes;
            ok = ok && fcs == fcs_want && a.out_off[row] + fcs <= a.out_cap;  // anything odd: the serial pass reports it
        }
    }
    // Block headers.  The position of a header is only known once the previous one has been read — a chain of
    // dependent loads, 1,600 of them in a 200 MiB frame.  Blocks of periodic or stored data all have the same size,
    // so lane i guesses "header i blocks ahead = pos + i * (size of the previous block)": when the guess holds the
    // wave walks 64 headers per round trip, when it does not (real data) it advances by one, as before.
    uint32_t k = 0;
    bool last = false, style_first = true, tail_mark = false;
    uint64_t stride = 0;  // 3 + payload bytes of the previous block (0: no guess yet)
    while (ok && !last) {
        if (k == nb) {
            // all expected blocks seen and none was the last: the encoder's higher effort tier closes such a frame with an
            // empty raw last block — its statement that every block stands on its own (zstd_encode.hip, tail_mark)
            tail_mark = pos + 3 == n && src[pos] == 1 && src[pos + 1] == 0 && src[pos + 2] == 0;
            if (tail_mark) pos += 3;
            else ok = false;
            break;
        }
        const uint64_t p_i = pos + (uint64_t)lane * stride;
        const bool valid = (lane == 0 || stride != 0) && k + lane < nb && p_i + 3 <= n;
        uint32_t bh = 0;
        if (valid) bh = src[p_i] | (src[p_i + 1] << 8) | (src[p_i + 2] << 16);
        const uint32_t type = (bh >> 1) & 3, size = bh >> 3;
        const bool lst = bh & 1;
        const uint64_t step = 3 + (type == 1 ? 1 : size);
        // lanes 0 .. run sit on real headers: every lane before them continued with exactly the guessed stride
        const uint64_t contm = __ballot(valid && !lst && type != 3 && step == stride);
        const uint32_t run = contm == ~0ull ? 63u : (uint32_t)__ffsll((long long)~contm) - 1;
        const bool mine = lane <= run && valid;
        if (__ballot(lane <= run && !valid) & 1ull) { ok = false; break; }  // lane 0 itself has no header to read
        const uint32_t taken = (uint32_t)__popcll(__ballot(mine));
        const uint64_t o0 = (uint64_t)(k + lane) * BLOCK_MAX;
        const uint64_t want = o0 + BLOCK_MAX < fcs_want ? BLOCK_MAX : fcs_want - o0;
        const bool bad = mine && (type == 3 || (type != 2 && size != want));  // raw / RLE blocks carry their size
        if (__ballot(bad)) { ok = false; break; }
        if (k == 0) {  // lane 0 of the first step: the frame's first block (all lanes run it, same addresses)
            const uint32_t type0 = __shfl(type, 0), size0 = __shfl(size, 0);
            if (type0 == 2) {
                const uint32_t type = type0, size = size0;
                (void)type;
                    const uint64_t b = pos + 3, bend = b + size;
                    bool style = bend <= n && size >= 2;
                    uint64_t q = b;
                    if (style) {
                        const uint32_t b0 = src[q], lt = b0 & 3, sf = (b0 >> 2) & 3;
                        uint64_t h = 0;
                        for (uint32_t i = 0; i < 5 && q + i < bend; i++) h |= (uint64_t)src[q + i] << (8 * i);
                        if (lt == 3) style = false;  // treeless: needs the previous block's tree
                        else if (lt <= 1) {
                            const uint32_t lh = (sf & 1) == 0 ? 1 : (sf == 1 ? 2 : 3);
                            const uint32_t regen = (uint32_t)((sf & 1) == 0 ? (h & 0xFF) >> 3 : (sf == 1 ? (h & 0xFFFF) >> 4 : (h & 0xFFFFFF) >> 4));
                            q += lh + (lt == 0 ? regen : 1);
                        } else {
                            const uint32_t lh = sf <= 1 ? 3 : (sf == 2 ? 4 : 5), nbits = sf <= 1 ? 10 : (sf == 2 ? 14 : 18);
                            q += lh + (uint32_t)((h >> (4 + nbits)) & ((1u << nbits) - 1));
                        }
                    }
                    if (style && q < bend) {
                        const uint32_t s0 = src[q];
                        const uint32_t hl = s0 == 0 ? 0 : (s0 < 128 ? 1 : (s0 < 255 ? 2 : 3));
                        if (hl && (q + hl >= bend || src[q + hl] != 0)) style = false;  // modes byte: all Predefined
                    } else style = false;
                    style_first = style;  // not this encoder's plain style: only the closing mark can vouch for the frame
            }
        }
        if (mine) a.item_src[base + k + lane] = (uint32_t)p_i;
        const uint32_t lastl = taken - 1;  // the last lane taken decides where the walk continues
        pos = __shfl(p_i, lastl) + __shfl(step, lastl);
        stride = __shfl(step, lastl);
        last = __shfl(lst ? 1u : 0u, lastl) != 0;
        k += taken;
    }
    ok = ok && k == nb && pos == n && (style_first || tail_mark);  // the expected number of blocks, nothing behind the last one
    if (!ok)
        for (uint32_t i = lane; i < nb; i += 64) a.item_src[base + i] = 0xFFFFFFFFu;
    if (lane == 0) a.row_flag[row] = ok ? 0u : 1u;
}

__global__ __launch_bounds__(64) void k_finish_blocks(BlockScanArgs a, int defer_flagged) {
    const uint32_t c = blockIdx.x * 64 + threadIdx.x;
    if (c >= a.n_cand) return;
    const uint32_t row = a.cand_row[c];
    if (a.status[row] < 0) return;  // host verdict stands
    if (a.row_flag[row]) {
        if (defer_flagged) { atomicAdd(a.pending_count + 4, 1u); return; }  // the batch path rules on it (k_bx_finish); counted for the table's hint
        a.status[row] = 1;  // handed to the general decoder, like a row the fused kernel gave up on
        a.pending[atomicAdd(a.pending_count, 1u)] = row;
    } else a.status[row] = 2;
}

// to-do list of the block decoder: the items the fused block kernel did not take
__global__ __launch_bounds__(256) void k_compact_items(const uint8_t *item_done, uint32_t n_items, uint32_t *todo, uint32_t *n_todo) {
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    const bool keep = i < n_items && !item_done[i];
    const uint64_t m = __ballot(keep);
    if (!m) return;
    const uint32_t lane = threadIdx.x & 63;
    uint32_t base = 0;
    if (lane == 0) base = atomicAdd(n_todo, (uint32_t)__popcll(m));
    base = __shfl(base, 0);
    if (keep) todo[base + __popcll(m & ((1ull << lane) - 1))] = i;
}
void launch_compact_items(const uint8_t *item_done, uint32_t n_items, uint32_t *todo, uint32_t *n_todo, hipStream_t s) {
    if (n_items) hipLaunchKernelGGL(k_compact_items, dim3((n_items + 255) / 256), dim3(256), 0, s, item_done, n_items, todo, n_todo);
}

void launch_scan_blocks(const BlockScanArgs &a, hipStream_t s) {
    if (a.n_cand) hipLaunchKernelGGL(k_scan_blocks, dim3(a.n_cand), dim3(64), 0, s, a);
}
void launch_finish_blocks(const BlockScanArgs &a, hipStream_t s, bool defer_flagged) {
    if (a.n_cand) hipLaunchKernelGGL(k_finish_blocks, dim3((a.n_cand + 63) / 64), dim3(64), 0, s, a, defer_flagged ? 1 : 0);
}

int decode_grid_size(int device) {
    hipDeviceProp_t p;
    if (hipGetDeviceProperties(&p, device) != hipSuccess) return 1024;
    return p.multiProcessorCount * 4;
}

size_t decode_lit_scratch_bytes(int grid) { return (size_t)grid * LIT_SCRATCH_BYTES; }

void launch_decode(const DecodeArgs &a, int grid, bool wide, hipStream_t s) {
    if (!a.n_rows) return;
    // wide: few, multi-MiB rows -> 16 waves per row share the long copies (raw blocks, long matches)
    if (wide) hipLaunchKernelGGL(k_zstd_decode<16>, dim3(grid), dim3(1024), 0, s, a);
    else hipLaunchKernelGGL(k_zstd_decode<4>, dim3(grid), dim3(256), 0, s, a);
}

}  // namespace zn
