// Foreign zstd frames, block-parallel (gfx950): the two-phase path of codec::decompress_into
// (znippy-common/src/codec.rs:L67-78) for frames another writer produced.  See the section comments.
#include "zstd_dev.h"

namespace zn {

// =============================================================================================
// Foreign frames in two phases.  A frame of >= 2 blocks that the block-item path gave up on (another writer's
// frame: repeat offsets, entropy tables reused from earlier blocks, matches that reach into earlier blocks, blocks
// of any size) used to be decoded by ONE workgroup from front to back.  Most of that work does not depend on the
// bytes in front of the block: Huffman-decoding the literals and FSE-decoding the sequences need only the block's own
// section (plus, for Treeless / Repeat_Mode, the table description of an earlier block, which is re-read).  So:
//   k_fz_scan     one wave per frame: frame header, then the chain of block headers -> one item per block
//   k_fz_entropy  one workgroup (2 waves) per BLOCK, all blocks of all frames at once: wave 0 reads the Huffman tree
//                 and decodes the literal streams into the literal pool, wave 1 builds the FSE tables and decodes the
//                 sequences into 8-byte records {literal length, match length, offset value} in the sequence pool
//   k_fz_exec     one wave per frame, blocks in order: repeat offsets, bounds, and the byte-moving execution — 64
//                 sequences at a time, one per lane, inside an LDS window that is streamed out in whole chunks
// Anything unexpected (a checksum trailer, an offset past 2^29, pools exhausted, an error of any kind) leaves the
// frame flagged: the serial decoder then takes it and produces the error code.
// =============================================================================================
constexpr uint32_t FZ_BACK = 64;  // how far back a Treeless / Repeat_Mode block looks for its table

struct FzTmp {  // scratch of fse_read_ncount / fse_build / huf_read_tree: one per wave
    int16_t norm[256];
    uint16_t fse_next[256];
    uint8_t fse_sym[512];
    uint8_t weights[256];
    uint16_t sym_start[256];
    uint16_t sym_len[256];
    uint32_t seq_ll[128];  // 64 FseEntry: the FSE table of the Huffman weights
    uint32_t huf_log;
    uint32_t bld[3];  // sequence tables whose counts wait in norm[64 k ..] (alphabet size; 0 = nothing to build): built by the wave after lane 0's parse
};

struct FzShared {
    FseEntry ll[512], ml[512], of[256];
    FseEntry dll[64], dml[64], dof[32];
    FseEntry rle[3];
    uint16_t huf[2048];
    FzTmp ta, tb;
    uint32_t claim;
    int32_t err;
    uint32_t lit_type, lit_kind, lit_len, lit_rle, lit_hdr, lit_comp;
    uint64_t lit_off;
    uint32_t n_streams, stream_off[4], stream_len[4], stream_out[4], stream_n[4];
    uint32_t seq_pos;  // where the sequences section starts inside the block
    uint32_t sel[3], log_[3];
    uint32_t nseq, bs_off, st_ll, st_of, st_ml;
    int32_t bs_pos;
    uint64_t seq_off;
    uint32_t sum_ll, sum_ml;
    uint32_t rep_out[3];
    uint32_t why;  // statistics: what sent the block to the serial decoder (1 table too far back, 2 pool full, 3 value range)
};

__global__ __launch_bounds__(64) void k_fz_scan(FzArgs a, uint32_t *work, uint32_t *work_count) {
    // one wave per candidate frame; every lane walks the same addresses (broadcast loads)
    const uint32_t c = blockIdx.x, lane = threadIdx.x;
    if (c >= a.n_cand) return;
    const uint32_t row = a.cand_row[c], base = a.cand_fzbase[c], cap = a.cand_fzcap[c];
    uint32_t nb = 0;
    if (a.row_flag[row] != 0 && !(a.preset && a.status[row] < 0)) {
        const uint8_t *const src = a.blobs + (a.blob_off[row] - a.blob_base);
        const uint64_t n = a.blob_size[row], fcs_want = a.usize[row];
        bool ok = n >= 9 && n < 0xFFFF0000ull && (src[0] | (src[1] << 8) | (src[2] << 16) | ((uint32_t)src[3] << 24)) == 0xFD2FB528u;
        uint64_t pos = 5;
        if (ok) {
            const uint32_t fhd = src[4];
            const uint32_t fcs_flag = fhd >> 6, single = (fhd >> 5) & 1, did_flag = fhd & 3;
            const uint32_t fcs_bytes = fcs_flag == 0 ? single : (1u << fcs_flag);
            ok = !(fhd & 8) && !((fhd >> 2) & 1) && did_flag == 0 && fcs_bytes != 0;  // a checksum trailer is the serial decoder's
            if (ok) {
                if (!single) pos++;
                ok = pos + fcs_bytes <= n;
                uint64_t fcs = 0;
                for (uint32_t i = 0; ok && i < fcs_bytes; i++) fcs |= (uint64_t)src[pos + i] << (8 * i);
                if (fcs_bytes == 2) fcs += 256;
                pos += fcs_bytes;
                ok = ok && fcs == fcs_want && fcs < 0xFFFFFFFFull && a.out_off[row] + fcs <= a.out_cap;
            }
        }
        uint32_t k = 0;
        bool last = false;
        while (ok && !last) {
            if (k >= cap || pos + 3 > n) { ok = false; break; }
            const uint32_t bh = src[pos] | (src[pos + 1] << 8) | (src[pos + 2] << 16);
            const uint32_t type = (bh >> 1) & 3, size = bh >> 3;
            const uint64_t step = 3 + (type == 1 ? 1 : size);
            if (type == 3 || size > BLOCK_MAX || pos + step > n) { ok = false; break; }
            if (lane == 0) a.items[base + k].src = (uint32_t)pos;
            pos += step;
            last = bh & 1;
            k++;
        }
        ok = ok && pos == n;
        nb = ok ? k : 0;
    }
    uint32_t start = 0;
    if (nb) start = uni(atomicAdd(work_count, lane == 0 ? nb : 0u));  // every lane executes the atomic (lane 0 adds)
    for (uint32_t i = lane; i < nb; i += 64) work[start + i] = base + i;
    if (lane == 0) a.cand_nb[c] = nb;
}

// Treeless literals: the tree of the nearest earlier block of the frame that carries a description.
__device__ int fz_find_tree(FzTmp &T, const FzArgs &a, const uint8_t *src, const uint8_t *blob_end, uint32_t base, uint32_t k) {
    for (uint32_t back = 1; back <= FZ_BACK && back <= k; back++) {
        const uint32_t pos = a.items[base + k - back].src;
        const uint32_t bh = src[pos] | (src[pos + 1] << 8) | (src[pos + 2] << 16);
        if (((bh >> 1) & 3) != 2) continue;  // raw / RLE blocks pass the tree on
        const uint8_t *b = src + pos + 3;
        LitHdr h;
        if (fz_lit_header(b, bh >> 3, h)) return E_CORRUPT;
        if (h.type == 2) {
            uint32_t tu = 0;
            return huf_read_tree(T, b + h.hdr, h.comp, blob_end, &tu);
        }
        // raw / RLE literals and treeless blocks pass it on as well
    }
    return E_UNSUP;
}

// Sequences_Section_Header of the section [q, q + n): number of sequences, then the three table descriptions.  Kinds in
// `want` (bit 0 LL, 1 OF, 2 ML) that are described here are set up; those in Repeat_Mode come back in *missing.
__device__ int fz_seq_tables(FzShared &S, FzTmp &T, const uint8_t *q, uint32_t n, uint32_t want, uint32_t *missing,
                             uint32_t *nseq_out, uint32_t *bits_at) {
    if (n < 1) return E_TRUNC;
    uint32_t p = 0, nseq = 0;
    const uint32_t b0 = q[0];
    if (b0 == 0) { nseq = 0; p = 1; }
    else if (b0 < 128) { nseq = b0; p = 1; }
    else if (b0 < 255) { if (n < 2) return E_TRUNC; nseq = ((b0 - 128) << 8) + q[1]; p = 2; }
    else { if (n < 3) return E_TRUNC; nseq = q[1] + ((uint32_t)q[2] << 8) + 0x7F00; p = 3; }
    *nseq_out = nseq;
    *missing = want;
    *bits_at = p;
    if (!nseq) return 0;
    if (p >= n) return E_TRUNC;
    const uint32_t modes = q[p++];
    if (modes & 3) return E_CORRUPT;
    const int shifts[3] = {6, 4, 2};
    const int kinds[3] = {K_LL, K_OF, K_ML};
    const int maxlog[3] = {9, 8, 9};
    const int maxsym[3] = {35, 31, 52};
    const int deflog[3] = {6, 5, 6};
    uint32_t miss = 0;
    for (int k = 0; k < 3; k++) {
        const uint32_t mode = (modes >> shifts[k]) & 3;
        const bool wanted = (want >> k) & 1;
        if (mode == 0) { if (wanted) { S.sel[k] = 0; S.log_[k] = deflog[k]; } }
        else if (mode == 1) {
            if (p >= n) return E_TRUNC;
            if (wanted) {
                const int rc = fse_set_rle(&S.rle[k], q[p], kinds[k]);
                if (rc) return rc;
                S.sel[k] = 1; S.log_[k] = 0;
            }
            p++;
        } else if (mode == 2) {
            int nsym, log;
            uint32_t used;
            // counts of a wanted table stay in norm[64 k ..] for the wave to build from (k_fz_entropy); a table that is only
            // stepped over is read into the spare quarter
            int rc = fse_read_ncount(T, q + p, n - p, maxlog[k], maxsym[k], &nsym, &log, &used, wanted ? 64 * k : 192);
            if (rc) return rc;
            if (wanted) { T.bld[k] = (uint32_t)nsym; S.sel[k] = 2; S.log_[k] = log; }
            p += used;
        } else if (wanted) miss |= 1u << k;
    }
    *missing = miss;
    *bits_at = p;
    return 0;
}


// (held to 128 VGPRs — 165 unbounded, same speed alone: the kernel has to fit into the quarter of the register file the
// general decoder leaves it, api.hip gen_share)
__global__ __launch_bounds__(128, 4) void k_fz_entropy(FzArgs a, const uint32_t *work, const uint32_t *work_count) {
    __shared__ FzShared S;
    const uint32_t tid = threadIdx.x, lane = tid & 63;
    const bool wave0 = tid < 64;
    const uint32_t n_work = *work_count;
    if (n_work == 0) return;
    if (tid == 0) {
        for (int i = 0; i < 36; i++) S.ta.norm[i] = c_ll_default[i];
        fse_build(S.ta, S.dll, 36, 6, K_LL);
        for (int i = 0; i < 53; i++) S.ta.norm[i] = c_ml_default[i];
        fse_build(S.ta, S.dml, 53, 6, K_ML);
        for (int i = 0; i < 29; i++) S.ta.norm[i] = c_of_default[i];
        fse_build(S.ta, S.dof, 29, 5, K_OF);
    }
    __syncthreads();
    for (;;) {
        if (tid == 0) S.claim = atomicAdd(a.cursor, 1u);
        __syncthreads();
        const uint32_t wi = S.claim;
        if (wi >= n_work) break;
        const uint32_t slot = work[wi];
        const uint32_t c = a.it_cand[slot];
        const uint32_t base = a.cand_fzbase[c], k = slot - base, row = a.cand_row[c];
        const uint8_t *const src = a.blobs + (a.blob_off[row] - a.blob_base);
        const uint8_t *const blob_end = src + a.blob_size[row];
        const uint32_t pos = a.items[slot].src;
        const uint32_t bh = src[pos] | (src[pos + 1] << 8) | (src[pos + 2] << 16);
        const uint32_t btype = (bh >> 1) & 3, bsize = bh >> 3;
        const uint8_t *const bsrc = src + pos + 3;
        if (btype != 2) {  // raw / RLE block: literals only
            if (tid == 0) {
                FzItem it;
                it.src = pos; it.out = bsize; it.nseq = 0; it.lit_len = bsize; it.seq_off = 0;
                it.lit_kind = btype == 0 ? 0u : 1u;
                it.lit_off = btype == 0 ? (uint64_t)pos + 3 : (uint64_t)bsrc[0];
                it.err = 0;
                it.rep[0] = FZ_SYM; it.rep[1] = FZ_SYM | (1u << 26); it.rep[2] = FZ_SYM | (2u << 26); it.pad = 0;
                a.items[slot] = it;
            }
            __syncthreads();
            continue;
        }
        if (tid == 0) {
            LitHdr h;
            int err = fz_lit_header(bsrc, bsize, h);
            S.lit_type = h.type; S.lit_len = h.regen; S.lit_hdr = h.hdr; S.lit_comp = h.comp; S.n_streams = 0;
            if (!err) {
                if (h.type == 0) { S.lit_kind = 0; S.lit_off = (uint64_t)pos + 3 + h.hdr; }
                else if (h.type == 1) { S.lit_kind = 1; S.lit_off = bsrc[h.hdr]; }
                else { S.lit_kind = 2; S.n_streams = h.streams; }
                S.seq_pos = fz_lit_section_bytes(h);
            }
            S.err = err;
            S.nseq = 0; S.sum_ll = 0; S.sum_ml = 0; S.seq_off = 0; S.why = 0;
            S.rep_out[0] = FZ_SYM; S.rep_out[1] = FZ_SYM | (1u << 26); S.rep_out[2] = FZ_SYM | (2u << 26);
        }
        __syncthreads();
        unsigned long long t_e = a.dbg ? __builtin_amdgcn_s_memtime() : 0;  // diagnostic (ZNIPPY_DDBG): where a block's entropy stage spends its cycles
#define ESTAMPZ(i) do { if (a.dbg && lane == 0) { const unsigned long long now_ = __builtin_amdgcn_s_memtime(); atomicAdd(&a.dbg[i], now_ - t_e); t_e = now_; } } while (0)
        if (a.dbg && tid == 0) atomicAdd(&a.dbg[20], 1ull);
        if (S.err == 0) {
            if (wave0) {
                // ---- literals: tree, table, streams ----
                if (S.lit_kind == 2) {
                    if (lane == 0) {
                        int err = 0;
                        uint32_t p = S.lit_hdr, remain = S.lit_comp;
                        const uint32_t regen = S.lit_len;
                        if (S.lit_type == 2) {
                            uint32_t tu = 0;
                            err = huf_read_tree(S.ta, bsrc + p, remain, blob_end, &tu);
                            if (!err) { p += tu; remain -= tu; }
                        } else { err = fz_find_tree(S.ta, a, src, blob_end, base, k); if (err == E_UNSUP) S.why = 1; }
                        if (!err) {
                            if (S.n_streams == 1) {
                                S.stream_off[0] = p; S.stream_len[0] = remain; S.stream_out[0] = 0; S.stream_n[0] = regen;
                            } else {
                                const uint32_t seg = (regen + 3) / 4;
                                if (remain < 6 || 3 * seg > regen) err = E_CORRUPT;
                                else {
                                    const uint32_t s1 = bsrc[p] | (bsrc[p + 1] << 8), s2 = bsrc[p + 2] | (bsrc[p + 3] << 8),
                                                   s3 = bsrc[p + 4] | (bsrc[p + 5] << 8);
                                    if (6 + s1 + s2 + s3 > remain) err = E_CORRUPT;
                                    else {
                                        const uint32_t s4 = remain - 6 - s1 - s2 - s3;
                                        p += 6;
                                        S.stream_off[0] = p; S.stream_len[0] = s1; S.stream_out[0] = 0; S.stream_n[0] = seg;
                                        S.stream_off[1] = p + s1; S.stream_len[1] = s2; S.stream_out[1] = seg; S.stream_n[1] = seg;
                                        S.stream_off[2] = p + s1 + s2; S.stream_len[2] = s3; S.stream_out[2] = 2 * seg; S.stream_n[2] = seg;
                                        S.stream_off[3] = p + s1 + s2 + s3; S.stream_len[3] = s4; S.stream_out[3] = 3 * seg; S.stream_n[3] = regen - 3 * seg;
                                    }
                                }
                            }
                        }
                        if (!err) {
                            const uint64_t room = ((uint64_t)regen + 79) & ~15ull;
                            const uint64_t off = atomicAdd(&a.pool_used[0], (unsigned long long)room);
                            if (off + room > a.lit_cap) { err = E_UNSUP; S.why = 2; }
                            S.lit_off = off;
                        }
                        if (err) atomicMin(&S.err, err);
                    }
                    __builtin_amdgcn_wave_barrier();
                    ESTAMPZ(21);
                    if (S.err == 0) {
                        const uint32_t hlog = S.ta.huf_log;
                        for (uint32_t sym = lane; sym < 256; sym += 64) {
                            const uint32_t len = S.ta.sym_len[sym];
                            if (len) {
                                const uint32_t st = S.ta.sym_start[sym];
                                const uint16_t e = (uint16_t)(sym | ((hlog + 1 - S.ta.weights[sym]) << 8));
                                for (uint32_t i = 0; i < len; i++) S.huf[st + i] = e;
                            }
                        }
                        __builtin_amdgcn_wave_barrier();
                        if (lane < S.n_streams) {
                            const int rc = fz_huf_stream(S.huf, hlog, bsrc + S.stream_off[lane], S.stream_len[lane], blob_end,
                                                         a.lit_pool + S.lit_off + S.stream_out[lane], S.stream_n[lane]);
                            if (rc) atomicMin(&S.err, rc);
                        }
                        ESTAMPZ(22);
                    }
                }
            } else {
                // ---- sequences: header, tables, bitstream ----
                if (lane == 0) {
                    int err = 0;
                    const uint8_t *q = bsrc + S.seq_pos;
                    const uint32_t qn = bsize - S.seq_pos;
                    uint32_t miss = 0, nseq = 0, bits_at = 0;
                    S.tb.bld[0] = S.tb.bld[1] = S.tb.bld[2] = 0;
                    if (S.seq_pos >= bsize) err = E_TRUNC;
                    if (!err) err = fz_seq_tables(S, S.tb, q, qn, 7u, &miss, &nseq, &bits_at);
                    if (!err && nseq == 0 && bits_at != qn) err = E_CORRUPT;
                    if (!err && nseq && miss) {
                        for (uint32_t back = 1; back <= FZ_BACK && back <= k && miss && !err; back++) {
                            const uint32_t pj = a.items[base + k - back].src;
                            const uint32_t bj = src[pj] | (src[pj + 1] << 8) | (src[pj + 2] << 16);
                            if (((bj >> 1) & 3) != 2) continue;
                            const uint8_t *b = src + pj + 3;
                            const uint32_t sz = bj >> 3;
                            LitHdr h;
                            if (fz_lit_header(b, sz, h)) { err = E_CORRUPT; break; }
                            const uint32_t ls = fz_lit_section_bytes(h);
                            if (ls >= sz) { err = E_CORRUPT; break; }
                            uint32_t m2 = 0, n2 = 0, at2 = 0;
                            err = fz_seq_tables(S, S.tb, b + ls, sz - ls, miss, &m2, &n2, &at2);
                            if (!err && n2) miss = m2;
                        }
                        if (!err && miss) { err = E_UNSUP; S.why = 1; }
                    }
                    if (!err && nseq) {
                        if (bits_at >= qn) err = E_TRUNC;
                        else {
                            BitR b;
                            if (!b.init(q + bits_at, qn - bits_at, blob_end)) err = E_CORRUPT;
                            else {
                                S.st_ll = b.read(S.log_[0]);
                                S.st_of = b.read(S.log_[1]);
                                S.st_ml = b.read(S.log_[2]);
                                S.bs_pos = (int32_t)b.pos;
                                S.bs_off = S.seq_pos + bits_at;
                            }
                        }
                    }
                    if (!err && nseq) {
                        const uint64_t off = atomicAdd(&a.pool_used[1], (unsigned long long)nseq);
                        if (off + nseq > a.seq_cap) { err = E_UNSUP; S.why = 2; }
                        S.seq_off = off;
                    }
                    S.nseq = nseq;
                    if (err) atomicMin(&S.err, err);
                }
                __builtin_amdgcn_wave_barrier();
                if (S.err == 0 && uni(S.nseq))
                    for (uint32_t k = 0; k < 3; k++)
                        if (uni(S.tb.bld[k]))
                            fse_build_wave(S.tb.norm + 64 * k, uni(S.tb.bld[k]), uni(S.log_[k]), k == 0 ? K_LL : (k == 1 ? K_OF : K_ML),
                                           k == 0 ? S.ll : (k == 1 ? S.of : S.ml), S.tb.fse_next, lane);
                ESTAMPZ(23);
                const uint32_t nseq = uni(S.nseq);
                if (a.dbg && lane == 0) atomicAdd(&a.dbg[25], (unsigned long long)nseq);
                if (S.err == 0 && nseq) {
                    const uint32_t sel0 = uni(S.sel[0]), sel1 = uni(S.sel[1]), sel2 = uni(S.sel[2]);
                    const FseEntry *tl = sel0 == 0 ? S.dll : (sel0 == 1 ? &S.rle[0] : S.ll);
                    const FseEntry *to = sel1 == 0 ? S.dof : (sel1 == 1 ? &S.rle[1] : S.of);
                    const FseEntry *tm = sel2 == 0 ? S.dml : (sel2 == 1 ? &S.rle[2] : S.ml);
                    const uint8_t *const bbase = bsrc + uni(S.bs_off);
                    int32_t left = (int32_t)uni((uint32_t)S.bs_pos);
                    // Two stages per group of 64 sequences, because one wave gets one issue slot every 4 cycles and the slowest
                    // block sets this kernel's time:
                    //   A  the serial chain, and nothing else: the three states walk through their tables, the bit position
                    //      moves; per sequence the wave notes (states, position) in lane (i mod 64).  The only bits it extracts
                    //      are the next states' — out of 512 bytes of the stream kept in the wave's registers (lane k holds
                    //      bytes [wbase + 8k, +8); two readlane pairs and a funnel shift, no memory access in the chain);
                    //   B  the rest, 64 sequences at a time with lane = sequence: table entries again, 16 bytes of the stream
                    //      ending at the lane's position, the three values, the sums; then the repeat-offset rules in order
                    //      (scalar, only for groups that use a repeat code) and one coalesced store of the records.
                    uint64_t wq = 0;
                    int32_t wbits = INT32_MAX;  // bit position of the end of the window's first 8 bytes (none loaded yet)
                    uint32_t sl = uni(S.st_ll), so = uni(S.st_of), sm = uni(S.st_ml);
                    const uint2 *const tl2 = reinterpret_cast<const uint2 *>(tl), *const to2 = reinterpret_cast<const uint2 *>(to),
                                *const tm2 = reinterpret_cast<const uint2 *>(tm);
                    unsigned long long *const recs = a.seq_pool + uni64(S.seq_off);
                    uint32_t sum_ll = 0, sum_ml = 0;
                    uint32_t r0 = FZ_SYM, r1 = FZ_SYM | (1u << 26), r2 = FZ_SYM | (2u << 26);
                    int err = 0;
                    for (uint32_t g0 = 0; g0 < nseq && !err; g0 += 64) {
                        const uint32_t cnt = nseq - g0 < 64 ? nseq - g0 : 64;
                        uint32_t my_so = 0, my_sm = 0, my_sl = 0, my_leftu = 0, max_ofb = 0;
                        int32_t margin = 0;
                        // ---- A ----
                        // (the block's last sequence takes no state bits: it is handled behind the loop, which so has no such case;
                        // verdicts are a running min / max, not branches: one way out of the loop keeps its state in place — with
                        // early exits the compiler copied every loop-carried register twice per trip — and a stream that has gone
                        // wrong only moves `left` below zero and reads zero bytes in front of the stream)
                        const bool has_last = g0 + cnt == nseq;
                        const uint32_t cnt_a = has_last ? cnt - 1 : cnt;
                        for (uint32_t g = 0; g < cnt_a; g++) {
                            const uint32_t vo_ = to2[so].x, vm_ = tm2[sm].x, vl_ = tl2[sl].x;  // three LDS reads in flight together ...
                            const uint32_t eox = uni(vo_), emx = uni(vm_), elx = uni(vl_);    // ... before the first is waited for (next:16 | nbits:8 | addbits:8)
                            const uint32_t ofb = eox >> 24, need_v = ofb + (emx >> 24) + (elx >> 24);
                            const uint32_t nbl = (elx >> 16) & 0xFF, nbm = (emx >> 16) & 0xFF, nbo = (eox >> 16) & 0xFF, need_s = nbl + nbm + nbo;
                            max_ofb = max(max_ofb, ofb);
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
                            const uint32_t vo_ = to2[so].x, vm_ = tm2[sm].x, vl_ = tl2[sl].x;
                            const uint32_t eox = uni(vo_), emx = uni(vm_), elx = uni(vl_);
                            const uint32_t ofb = eox >> 24, need_v = ofb + (emx >> 24) + (elx >> 24);
                            max_ofb = max(max_ofb, ofb);
                            margin = min(margin, left - (int32_t)need_v);
                            wrlane4_u(my_so, my_sm, my_sl, my_leftu, so, sm, sl, (uint32_t)left, cnt - 1);
                            left -= (int32_t)need_v;
                        }
                        if (margin < 0) { err = E_CORRUPT; break; }
                        if (max_ofb > 27) { err = E_UNSUP; if (lane == 0) S.why = 3; break; }
                        // ---- B ----
                        const bool on = lane < cnt;
                        uint32_t ov = 4, ml = 0, ll = 0;
                        if (on) {
                            const uint2 eo = to2[my_so], em = tm2[my_sm], el = tl2[my_sl];
                            const uint32_t ofb = eo.x >> 24, mlb = em.x >> 24, llb = el.x >> 24, need_v = ofb + mlb + llb;
                            const int32_t my_left = (int32_t)my_leftu, bend = (my_left + 7) >> 3;
                            uint64_t lo8 = 0, hi8 = 0;  // stream bytes [bend - 16, bend - 8) and [bend - 8, bend); zero in front of the stream
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
                            const uint64_t H = (hi8 << al) | ((lo8 >> 1) >> (63 - al));  // the lane's position is bit 64 now
                            const uint64_t xv = (H >> 1) >> (63 - need_v);                 // the top need_v (<= 59) bits
                            ov = eo.y + (uint32_t)(xv >> (mlb + llb));
                            ml = em.y + ((uint32_t)(xv >> llb) & ((1u << mlb) - 1u));
                            ll = el.y + ((uint32_t)xv & ((1u << llb) - 1u));
                        }
                        {
                            uint32_t a_ll = ll, a_ml = ml;
#pragma unroll
                            for (int dd = 32; dd >= 1; dd >>= 1) { a_ll += __shfl_xor(a_ll, dd); a_ml += __shfl_xor(a_ml, dd); }
                            sum_ll += a_ll; sum_ml += a_ml;
                        }
                        if (sum_ll + sum_ml > BLOCK_MAX) { err = E_UNSUP; if (lane == 0) S.why = 3; break; }
                        // repeat offsets (RFC 8878 3.1.1.5), against the symbolic incoming history
                        uint32_t o_mine = ov - 3;
                        if (__ballot(on && ov <= 3) == 0ull && cnt >= 3) {
                            r0 = rdlane_u(o_mine, cnt - 1); r1 = rdlane_u(o_mine, cnt - 2); r2 = rdlane_u(o_mine, cnt - 3);
                        } else {
                            for (uint32_t j = 0; j < cnt; j++) {
                                const uint32_t ovj = rdlane_u(ov, j), llj = rdlane_u(ll, j);
                                uint32_t o;
                                if (ovj > 3) { o = ovj - 3; r2 = r1; r1 = r0; r0 = o; }
                                else {
                                    const uint32_t idx = ovj - 1 + (llj == 0 ? 1u : 0u);
                                    if (idx == 0) o = r0;
                                    else {
                                        if (idx < 3) o = idx == 1 ? r1 : r2;
                                        else if (r0 & FZ_SYM) {  // incoming entry minus one more
                                            o = r0 + 1;
                                            if ((o & 0x3FFFFFFu) == 0x3FFFFFFu) { err = E_UNSUP; if (lane == 0) S.why = 3; break; }
                                        } else {
                                            o = r0 - 1;
                                            if (o == 0) { err = E_CORRUPT; break; }
                                        }
                                        if (idx > 1) r2 = r1;
                                        r1 = r0; r0 = o;
                                    }
                                }
                                if (lane == j) o_mine = o;
                            }
                            if (err) break;
                        }
                        if (on) recs[g0 + lane] = (unsigned long long)ll | ((unsigned long long)ml << 17) | ((unsigned long long)o_mine << 35);
                    }
                    if (!err && left != 0) err = E_CORRUPT;
                    ESTAMPZ(24);
                    if (lane == 0) {
                        S.sum_ll = sum_ll; S.sum_ml = sum_ml;
                        S.rep_out[0] = r0; S.rep_out[1] = r1; S.rep_out[2] = r2;
                        if (err) atomicMin(&S.err, err);
                    }
                }
            }
        }
        __syncthreads();
        if (tid == 0) {
            int err = S.err;
            if (!err && S.sum_ll > S.lit_len) err = E_CORRUPT;
            if (!err && S.lit_len + S.sum_ml > BLOCK_MAX) err = E_CORRUPT;
            FzItem it;
            it.src = pos; it.out = S.lit_len + S.sum_ml; it.nseq = S.nseq; it.lit_len = S.lit_len;
            it.seq_off = S.seq_off; it.lit_off = S.lit_off; it.lit_kind = S.lit_kind; it.err = err;
            it.rep[0] = S.rep_out[0]; it.rep[1] = S.rep_out[1]; it.rep[2] = S.rep_out[2]; it.pad = 0;
            a.items[slot] = it;
            if (err) { atomicAdd(&a.pool_used[3], 1ull); atomicAdd(&a.pool_used[4 + (S.why & 3)], 1ull); }  // statistics: blocks left to the serial decoder, and why
        }
        __syncthreads();
    }
}

template <bool PROF>
__global__ __launch_bounds__(64) void k_fz_exec(FzArgs a) {
    __shared__ __attribute__((aligned(16))) uint8_t W[WIN_HIST + WIN_CAP + 64];
    const uint32_t c = blockIdx.x, lane = threadIdx.x;
    if (c >= a.n_cand) return;
    const uint32_t nb = a.cand_nb[c];
    if (!nb) return;
    const uint32_t row = a.cand_row[c], base = a.cand_fzbase[c];
    const uint8_t *const src = a.blobs + (a.blob_off[row] - a.blob_base);
    uint8_t *const out = a.out + a.out_off[row];
    const uint64_t fcs = a.usize[row];
    {   // every block came through the entropy phase and the sizes add up to the frame's content size
        unsigned long long tot = 0, seqs = 0;
        uint32_t bad = 0;
        for (uint32_t i = lane; i < nb; i += 64) {
            bad |= a.items[base + i].err != 0;
            tot += a.items[base + i].out;
            seqs += a.items[base + i].nseq;
        }
        for (int d = 32; d >= 1; d >>= 1) { tot += __shfl_xor(tot, d); seqs += __shfl_xor(seqs, d); bad |= __shfl_xor(bad, d); }
        if (bad || tot != fcs) return;
        // A frame of a few very long sequences (periodic or constant data: one 128 KiB match per block) is copy work, not
        // sequence work: the serial decoder's 1,024-thread variant moves it three times faster than one wave can
        // (16 x 8 MiB of periodic text: 0.72 ms against 2.2 ms here) — left to it.
        if (seqs * 2048 < fcs) return;
    }
    uint64_t opos = 0;  // output bytes already streamed to HBM
    uint32_t win_n = 0, hist_n = 0, r0 = 1, r1 = 4, r2 = 8;
    bool dirty = false;
    int err = 0;
    ExecProf prof;
    unsigned long long p_groups = 0, p_seqs = 0, p_big = 0, p_flush = 0, p_hist = 0, p_rep = 0;
    unsigned long long c_rec = 0, c_rep = 0, c_big = 0, c_flush = 0, c_hist = 0, c_tail = 0, t_mark = 0;
    const unsigned long long t_begin = PROF ? __builtin_amdgcn_s_memtime() : 0;
#define FZ_T0() do { if (PROF) t_mark = __builtin_amdgcn_s_memtime(); } while (0)
#define FZ_T1(acc) do { if (PROF) { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); const unsigned long long n_ = __builtin_amdgcn_s_memtime(); acc += n_ - t_mark; t_mark = n_; } } while (0)
    for (uint32_t k = 0; k < nb && !err; k++) {
        const FzItem it = a.items[base + k];
        const uint32_t nseq = uni(it.nseq), lit_len = uni(it.lit_len), kind = uni(it.lit_kind);
        const uint64_t lit_off = uni64(it.lit_off);
        const bool rle_lits = kind == 1;
        const uint8_t rle_byte = (uint8_t)lit_off;
        const uint8_t *const lit_ptr = kind == 0 ? src + lit_off : a.lit_pool + lit_off;
        const unsigned long long *const recs = a.seq_pool + uni64(it.seq_off);
        uint32_t lpos = 0;
        for (uint32_t g0 = 0; g0 < nseq && !err; g0 += 64) {
            const uint32_t cnt = nseq - g0 < 64 ? nseq - g0 : 64;
            const bool on = lane < cnt;
            FZ_T0();
            const unsigned long long rec = on ? recs[g0 + lane] : 0ull;
            FZ_T1(c_rec);
            if (PROF) { p_groups++; p_seqs += cnt; }
            const uint32_t ll0 = (uint32_t)rec & 0x1FFFFu, ml0 = (uint32_t)(rec >> 17) & 0x3FFFFu;
            const uint32_t ov = on ? (uint32_t)(rec >> 35) : 4u;
            // offsets come resolved from the entropy phase, up to the history this block started with
            uint32_t offset = ov;
            if (ov & FZ_SYM) {
                const uint32_t kk = (ov >> 26) & 3, dd = ov & 0x3FFFFFFu;
                const uint32_t in = kk == 0 ? r0 : (kk == 1 ? r1 : r2);
                offset = in > dd ? in - dd : 0u;
            }
            if (__ballot(on && offset == 0) != 0ull) { err = E_CORRUPT; break; }
            {
                uint32_t linc = ll0, pinc = ll0 + ml0;
linc = wave_incl_scan(linc); pinc = wave_incl_scan(pinc);
                const bool bad = on && ((uint64_t)offset > opos + win_n + pinc - ml0 || lpos + linc > lit_len);
                if (__ballot(bad) != 0ull) { err = E_CORRUPT; break; }
            }
            FZ_T1(c_rep);
            // ---- execute the group (the narrow serial decoder's window scheme, one wave) ----
            uint32_t si = 0;
            while (si < cnt) {
                const uint32_t idx = si + lane;
                const bool v = idx < cnt;
                uint32_t ll = __shfl(ll0, idx & 63), ml = __shfl(ml0, idx & 63), off = __shfl(offset, idx & 63);
                if (!v) { ll = 0; ml = 0; off = 1; }
                const uint32_t tot = ll + ml;
                const uint64_t bigm = __ballot(v && tot > WIN_SEQ_MAX);
                const uint32_t nv = cnt - si;
                const uint32_t ncand = bigm ? (uint32_t)__ffsll((long long)bigm) - 1 : nv;
                if (ncand == 0) {
                    FZ_T0();
                    if (PROF) p_big++;
                    // one long sequence, straight to HBM (the window is emptied first)
                    if (win_n) {
                        (void)win_flush(W, out, opos, win_n, hist_n, lane, false);
                        opos += win_n;
                        win_n = 0;
                    }
                    hist_n = 0;
                    const uint32_t llx = rdlane_u(ll, 0), mlx = rdlane_u(ml, 0), offx = rdlane_u(off, 0);
                    if (llx) {
                        if (rle_lits) coop_fill(out + opos, rle_byte, llx, lane, 64);
                        else coop_copy(out + opos, lit_ptr + lpos, llx, lane, 64);
                        opos += llx; lpos += llx;
                    }
                    wave_mem_sync();
                    coop_match<1>(out + opos, offx, mlx, lane, false, nullptr);
                    opos += mlx;
                    dirty = true;
                    si++;
                    FZ_T1(c_big);
                    continue;
                }
                uint32_t end = lane < ncand ? tot : 0, lend = lane < ncand ? ll : 0;
                end = wave_incl_scan(end); lend = wave_incl_scan(lend);
                uint32_t fit = (uint32_t)__popcll(__ballot(lane < ncand && end <= WIN_CAP - win_n));
                if (fit == 0) {  // chunk full: stream it out, keep the newest bytes as history
                    FZ_T0();
                    if (PROF) p_flush++;
                    (void)win_flush(W, out, opos, win_n, hist_n, lane, true);
                    hist_n = hist_n + win_n < WIN_HIST ? hist_n + win_n : WIN_HIST;
                    opos += win_n;
                    win_n = 0;
                    fit = (uint32_t)__popcll(__ballot(lane < ncand && end <= WIN_CAP));
                    FZ_T1(c_flush);
                }
                if (dirty) { wave_mem_sync(); dirty = false; }
                const uint32_t want_h = opos < WIN_HIST ? (uint32_t)opos : WIN_HIST;
                if (win_n == 0 && hist_n < want_h) {  // history lost to a direct copy: read the newest output back
                    FZ_T0();
                    if (PROF) p_hist++;
                    coop_copy(W + WIN_HIST - want_h, out + opos - want_h, want_h, lane, 64);
                    hist_n = want_h;
                    FZ_T1(c_hist);
                }
                win_exec_group(W, out, opos, hist_n, lane, lane < fit, WIN_HIST + win_n + (end - tot), ll, ml, off,
                               lit_ptr + lpos + (lend - ll), rle_lits, rle_byte, PROF ? &prof : nullptr);
                win_n += rdlane_u(end, fit - 1);
                lpos += rdlane_u(lend, fit - 1);
                si += fit;
            }
        }
        if (err) break;
        {   // the history this block leaves
            uint32_t nr[3];
#pragma unroll
            for (int i = 0; i < 3; i++) {
                const uint32_t x = uni(it.rep[i]);
                if (x & FZ_SYM) {
                    const uint32_t kk = (x >> 26) & 3, dd = x & 0x3FFFFFFu;
                    const uint32_t in = kk == 0 ? r0 : (kk == 1 ? r1 : r2);
                    nr[i] = in > dd ? in - dd : 0u;
                } else nr[i] = x;
            }
            r0 = nr[0]; r1 = nr[1]; r2 = nr[2];
        }
        // literals left after the last sequence, then the window goes out (history stays for the next block)
        FZ_T0();
        const uint32_t rest = lit_len - lpos;
        if (opos + win_n + rest > fcs) { err = E_CORRUPT; break; }
        const bool in_win = rest <= WIN_CAP - win_n;
        if (in_win && rest) {
            uint8_t *d = W + WIN_HIST + win_n;
            if (rle_lits) for (uint32_t i = lane; i < rest; i += 64) d[i] = rle_byte;
            else coop_copy(d, lit_ptr + lpos, rest, lane, 64);
        }
        if (dirty) { wave_mem_sync(); dirty = false; }
        uint32_t h = win_flush(W, out, opos, in_win ? win_n + rest : win_n, hist_n, lane, in_win);
        if (!in_win) {
            if (rle_lits) coop_fill(out + opos + win_n, rle_byte, rest, lane, 64);
            else coop_copy(out + opos + win_n, lit_ptr + lpos, rest, lane, 64);
            wave_mem_sync();
            h = 0;
        }
        opos += win_n + rest;
        win_n = 0;
        hist_n = in_win ? h : 0;
        FZ_T1(c_tail);
    }
    if (PROF && a.dbg && lane == 0) {
        const unsigned long long vals[17] = {1, p_groups, p_seqs, p_big, prof.rounds, p_flush, p_hist, p_rep,
                                             __builtin_amdgcn_s_memtime() - t_begin, c_rec, c_rep, c_big, c_flush, c_hist, prof.t_lits, prof.t_match, c_tail};
        for (int i = 0; i < 17; i++) atomicAdd(&a.dbg[i], vals[i]);
    }
#undef FZ_T0
#undef FZ_T1
    if (!err && opos == fcs && lane == 0) {
        a.row_flag[row] = 0;  // k_finish_blocks turns this into status 2 (hash me)
        atomicAdd(&a.pool_used[2], 1ull);  // statistics: frames decoded by this path
    }
}

void launch_fz_scan(const FzArgs &a, uint32_t *work, uint32_t *work_count, hipStream_t s) {
    hipLaunchKernelGGL(k_fz_scan, dim3(a.n_cand), dim3(64), 0, s, a, work, work_count);
}
void launch_fz_entropy(const FzArgs &a, int cus, const uint32_t *work, const uint32_t *work_count, hipStream_t s) {
    const uint32_t grid = std::min<uint32_t>(a.total_items, (uint32_t)cus * 6);
    hipLaunchKernelGGL(k_fz_entropy, dim3(grid), dim3(128), 0, s, a, work, work_count);
}
void launch_fz_exec(const FzArgs &a, hipStream_t s) {
    if (a.dbg) hipLaunchKernelGGL(k_fz_exec<true>, dim3(a.n_cand), dim3(64), 0, s, a);
    else hipLaunchKernelGGL(k_fz_exec<false>, dim3(a.n_cand), dim3(64), 0, s, a);
}


}  // namespace zn
