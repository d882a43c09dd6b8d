// Foreign zstd frames, block-parallel (gfx950): the two-phase path of codec::decompress_into
// (znippy-common/src/codec.rs:L67-78) for frames another writer produced.  See the section comments.
#include <algorithm>
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


// The sequences bitstream of ONE block by ONE wave -> records (k_fz_entropy, and the batch path's blocks of many sequences).
// tl2 / to2 / tm2: the block's decoding tables in LDS as {next:16 | nbits:8 << 16 | extra bits:8 << 24, base value}; sl / so /
// sm: the initial states, left: unread bits.  Repeat offsets are resolved against a symbolic incoming history (FZ_SYM).
__device__ __forceinline__ int fz_wave_sequences(const uint2 *const tl2, const uint2 *const to2, const uint2 *const tm2, const uint8_t *const bbase,
                                                 const uint8_t *const blob_end, int32_t left, uint32_t sl, uint32_t so, uint32_t sm, const uint32_t nseq,
                                                 unsigned long long *const recs, const uint32_t lane, uint32_t *sum_ll_out, uint32_t *sum_ml_out,
                                                 uint32_t rep_out[3], uint32_t *why) {
    uint64_t wq = 0;
    int32_t wbits = INT32_MAX;  // bit position of the end of the window's first 8 bytes (none loaded yet)
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
        if (max_ofb > 27) { err = E_UNSUP; *why = 3; break; }
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
        if (sum_ll + sum_ml > BLOCK_MAX) { err = E_UNSUP; *why = 3; break; }
        // repeat offsets (RFC 8878 3.1.1.5), against the symbolic incoming history
        uint32_t o_mine = ov - 3;
        const uint64_t repm = __ballot(on && ov <= 3);
        if (repm == 0ull && cnt >= 3) {
            r0 = rdlane_u(o_mine, cnt - 1); r1 = rdlane_u(o_mine, cnt - 2); r2 = rdlane_u(o_mine, cnt - 3);
        } else {
            // Only the sequences that USE the history are walked (a third of real text's, fewer in binaries): the ones in
            // between only push their offsets, and what three or more of them leave is just the last three.  (First form:
            // all 64 sequences one by one, ~30 instructions each — a quarter of this function's time.)
            uint32_t s0 = uni(r0), s1 = uni(r1), s2 = uni(r2), prev = 0;
            auto push_run = [&](uint32_t upto) {  // sequences prev .. upto - 1 carry their own offsets
                const uint32_t k = upto - prev;
                if (k >= 3) { s0 = rdlane_u(o_mine, upto - 1); s1 = rdlane_u(o_mine, upto - 2); s2 = rdlane_u(o_mine, upto - 3); }
                else if (k == 2) { s2 = s0; s1 = rdlane_u(o_mine, upto - 2); s0 = rdlane_u(o_mine, upto - 1); }
                else if (k == 1) { s2 = s1; s1 = s0; s0 = rdlane_u(o_mine, upto - 1); }
            };
            uint64_t m = repm;
            while (m) {
                const uint32_t j = (uint32_t)__ffsll((long long)m) - 1;
                m &= m - 1;
                push_run(j);
                const uint32_t ovj = rdlane_u(ov, j), llj = rdlane_u(ll, j);
                const uint32_t idx = ovj - 1 + (llj == 0 ? 1u : 0u);
                uint32_t o;
                if (idx == 0) o = s0;
                else {
                    if (idx < 3) o = idx == 1 ? s1 : s2;
                    else if (s0 & FZ_SYM) {  // incoming entry minus one more
                        o = s0 + 1;
                        if ((o & 0x3FFFFFFu) == 0x3FFFFFFu) { err = E_UNSUP; *why = 3; break; }
                    } else {
                        o = s0 - 1;
                        if (o == 0) { err = E_CORRUPT; break; }
                    }
                    if (idx > 1) s2 = s1;
                    s1 = s0; s0 = o;
                }
                o_mine = (uint32_t)zn_writelane((int)o, (int)j, (int)o_mine);
                prev = j + 1;
            }
            if (err) break;
            push_run(cnt);
            r0 = s0; r1 = s1; r2 = s2;
        }
        if (on) recs[g0 + lane] = (unsigned long long)ll | ((unsigned long long)ml << 17) | ((unsigned long long)o_mine << 35);
    }
    if (!err && left != 0) err = E_CORRUPT;
    *sum_ll_out = sum_ll; *sum_ml_out = sum_ml;
    rep_out[0] = r0; rep_out[1] = r1; rep_out[2] = r2;
    return err;
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
                    uint32_t sum_ll = 0, sum_ml = 0, rr[3] = {0, 0, 0}, why3 = 0;
                    const int err = fz_wave_sequences(reinterpret_cast<const uint2 *>(tl), reinterpret_cast<const uint2 *>(to), reinterpret_cast<const uint2 *>(tm),
                                                      bbase, blob_end, left, uni(S.st_ll), uni(S.st_of), uni(S.st_ml), nseq, a.seq_pool + uni64(S.seq_off), lane,
                                                      &sum_ll, &sum_ml, rr, &why3);
                    if (why3 && lane == 0) S.why = why3;
                    const uint32_t r0 = rr[0], r1 = rr[1], r2 = rr[2];
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

// One frame, executed by one wave: the blocks' records and literals come from the pools, W is the wave's LDS window.
constexpr uint32_t BX_SMALL_HIST = 2048, BX_SMALL_CAP = 4096;  // the small window (k_bx_exec)
#ifndef ZN_BXW
#define ZN_BXW 5
#endif
constexpr int BX_SMALL_WAVES = ZN_BXW;                                // ... and the waves per SIMD its kernel is compiled for
template <bool PROF, uint32_t WH = WIN_HIST, uint32_t WC = WIN_CAP, uint32_t WSM = WIN_SEQ_MAX>
__device__ __forceinline__ bool fz_exec_frame(const FzArgs &a, const uint32_t c, uint8_t *const W, const uint32_t lane) {  // true: the frame is decoded
    // (one frame per wave: everything about the frame is wave-uniform, and said to be — values loaded through vector
    // loads otherwise live in vector registers, and the small-window form, held at 102 of them, spilled these pointers
    // and reloaded them inside its copy loops)
    const uint32_t nb = uni(a.cand_nb[c]);
    if (!nb) return false;
    const uint32_t row = uni(a.cand_row[c]), base = uni(a.cand_fzbase[c]);
    const uint8_t *const src = a.blobs + (uni64(a.blob_off[row]) - a.blob_base);
    uint8_t *const out = a.out + uni64(a.out_off[row]);
    const uint64_t fcs = uni64(a.usize[row]);
    {   // every block came through the entropy phase and the sizes add up to the frame's content size
        unsigned long long tot = 0, seqs = 0;
        uint32_t bad = 0;
        for (uint32_t i = lane; i < nb; i += 64) {
            bad |= a.items[base + i].err != 0;
            tot += a.items[base + i].out;
            seqs += a.items[base + i].nseq;
        }
        for (int d = 32; d >= 1; d >>= 1) { tot += __shfl_xor(tot, d); seqs += __shfl_xor(seqs, d); bad |= __shfl_xor(bad, d); }
        if (bad || tot != fcs) return false;
        // A frame of a few very long sequences (periodic or constant data: one 128 KiB match per block) is copy work, not
        // sequence work: the serial decoder's 1,024-thread variant moves it three times faster than one wave can
        // (16 x 8 MiB of periodic text: 0.72 ms against 2.2 ms here) — left to it.
        if (fcs >= (1u << 20) && seqs * 2048 < fcs) return false;  // (a smaller frame is not worth a second launch's latency)
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
        unsigned long long rec_next = lane < nseq ? recs[lane] : 0ull;  // the records of a group are fetched one group ahead
        for (uint32_t g0 = 0; g0 < nseq && !err; g0 += 64) {
            const uint32_t cnt = nseq - g0 < 64 ? nseq - g0 : 64;
            const bool on = lane < cnt;
            FZ_T0();
            const unsigned long long rec = on ? rec_next : 0ull;
            rec_next = g0 + 64 + lane < nseq ? recs[g0 + 64 + lane] : 0ull;
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
                const uint64_t bigm = __ballot(v && tot > WSM);
                const uint32_t nv = cnt - si;
                const uint32_t ncand = bigm ? (uint32_t)__ffsll((long long)bigm) - 1 : nv;
                if (ncand == 0) {
                    FZ_T0();
                    if (PROF) p_big++;
                    // one long sequence, straight to HBM (the window is emptied first)
                    if (win_n) {
                        (void)win_flush<WH>(W, out, opos, win_n, hist_n, lane, false);
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
                uint32_t fit = (uint32_t)__popcll(__ballot(lane < ncand && end <= WC - win_n));
                if (fit == 0) {  // chunk full: stream it out, keep the newest bytes as history
                    FZ_T0();
                    if (PROF) p_flush++;
                    (void)win_flush<WH>(W, out, opos, win_n, hist_n, lane, true);
                    hist_n = hist_n + win_n < WH ? hist_n + win_n : WH;
                    opos += win_n;
                    win_n = 0;
                    fit = (uint32_t)__popcll(__ballot(lane < ncand && end <= WC));
                    FZ_T1(c_flush);
                }
                if (dirty) { wave_mem_sync(); dirty = false; }
                const uint32_t want_h = opos < WH ? (uint32_t)opos : WH;
                if (win_n == 0 && hist_n < want_h) {  // history lost to a direct copy: read the newest output back
                    FZ_T0();
                    if (PROF) p_hist++;
                    coop_copy(W + WH - want_h, out + opos - want_h, want_h, lane, 64);
                    hist_n = want_h;
                    FZ_T1(c_hist);
                }
                win_exec_group<WH, WC>(W, out, opos, hist_n, lane, lane < fit, WH + win_n + (end - tot), ll, ml, off,
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
        const bool in_win = rest <= WC - win_n;
        if (in_win && rest) {
            uint8_t *d = W + WH + win_n;
            if (rle_lits) for (uint32_t i = lane; i < rest; i += 64) d[i] = rle_byte;
            else coop_copy(d, lit_ptr + lpos, rest, lane, 64);
        }
        if (dirty) { wave_mem_sync(); dirty = false; }
        uint32_t h = win_flush<WH>(W, out, opos, in_win ? win_n + rest : win_n, hist_n, lane, in_win);
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
    const bool done = !err && opos == fcs;
    if (done && lane == 0) a.row_flag[row] = 0;  // the finish kernels turn this into status 2 (hash me)
    return done;
}

template <bool PROF>
__global__ __launch_bounds__(64) void k_fz_exec(FzArgs a) {
    __shared__ __attribute__((aligned(16))) uint8_t W[WIN_HIST + WIN_CAP + 64 + WIN_SCRATCH];
    if (blockIdx.x < a.n_cand && fz_exec_frame<PROF>(a, blockIdx.x, W, threadIdx.x) && threadIdx.x == 0)
        atomicAdd(&a.pool_used[2], 1ull);  // statistics: frames decoded by this path
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



// =============================================================================================
// Many foreign frames at once: LANE = block.
// The two-phase kernels above give every block a wave (or two) and run its serial chains — table descriptions, Huffman
// streams, the FSE sequence bitstream — on one lane or on the scalar unit: 63 of 64 lanes idle, and an archive of
// 100,000 small text frames decodes at the rate of ~1,000 lone waves.  With that many frames the chains
// are vectorised ACROSS blocks instead: 64 blocks per wave, one per lane, each lane running plain per-thread code on
// its own block with its tables in scratch pools in device memory (a lane's table lookups are gathers served by
// L2 / Infinity Cache; 64 chains per wave and thousands of waves hide their latency).
//   k_bx_scan   lane = candidate frame (the host's list of big single-block rows, the rows the fused kernel handed
//               over, the block candidates the block-item path flagged): frame header, walk of the block headers,
//               item slots handed out by one atomic per wave
//   k_bx_prep   lane = block: literals header, Huffman tree description -> weights -> decoding table (Huffman pool);
//               sequences header, table descriptions -> decoding tables (FSE pool, 4-byte cells); pool space for the
//               block's literals and records; the block joins the Huffman and / or the sequence list
//   k_bx_huf    lane = Huffman stream, 16 blocks per wave with their tables copied into LDS
//   k_bx_fse    lane = block: the sequence bitstream -> 8-byte records, repeat offsets resolved on the way (against a
//               symbolic incoming history for blocks that are not the first of their frame, as k_fz_entropy does)
//   k_bx_exec   wave = frame: fz_exec_frame
//   k_bx_finish lane = candidate: decoded -> status 2, anything else -> the serial decoder's list
// Any error or anything unsupported leaves the frame to the serial decoder, which also produces the error code.
// =============================================================================================
struct BxScratch {  // per wave; [i * 64 + lane]: a lane's i-th entry (consecutive lanes, consecutive addresses)
    int16_t norm[64 * 64];
    uint16_t nxt[64 * 64];
    // Two lives of one region: the Huffman side (phase 1 reads the weights, phase 2 fills the decoding table from them) is
    // over for a lane before it builds its first sequence table — 48 KB per wave instead of 66: three waves per CU, not two
    // (the kernel is lone waves waiting on their own chains: 100k blocks 1.58 -> 1.1 ms)
    union {
        struct {
            uint16_t wtab[64 * 64];  // FSE table of the Huffman weights: weight:4 | nbits:3 << 4 | next:6 << 7
            uint8_t weights[256 * 64];
            uint16_t rank[16 * 64];
        };
        uint8_t sym[512 * 64];       // a sequence table under construction: the symbol of every cell
    };
};

// every lane asks for `mine` units: one atomic per wave (all 64 lanes call this, converged)
__device__ __forceinline__ uint32_t wave_alloc32(uint32_t *counter, uint32_t mine, uint32_t lane) {
    const uint32_t incl = wave_incl_scan(mine);
    const uint32_t total = rdlane_u(incl, 63);
    uint32_t base = 0;
    if (lane == 0 && total) base = atomicAdd(counter, total);
    return rdlane_u(base, 0) + incl - mine;
}
__device__ __forceinline__ uint64_t wave_alloc64(unsigned long long *counter, uint32_t mine, uint32_t lane) {
    const uint32_t incl = wave_incl_scan(mine);
    const uint32_t total = rdlane_u(incl, 63);
    unsigned long long base = 0;
    if (lane == 0 && total) base = atomicAdd(counter, (unsigned long long)total);
    return rdlane64_u(base, 0) + incl - mine;
}

// forward bit reader with a 64-bit register window (table descriptions); bytes past n read as zero
struct FwdW {
    const uint8_t *p;
    uint32_t n, bitpos, wbit;
    uint64_t win;
    __device__ __forceinline__ void fill(uint32_t byte) {
        uint64_t v = 0;
        if (byte + 8 <= n) __builtin_memcpy(&v, p + byte, 8);
        else for (uint32_t i = 0; i < 8 && byte + i < n; i++) v |= (uint64_t)p[byte + i] << (8 * i);
        win = v;
        wbit = byte * 8;
    }
    __device__ __forceinline__ void init(const uint8_t *src, uint32_t len) { p = src; n = len; bitpos = 0; fill(0); }
    __device__ __forceinline__ uint32_t peek(uint32_t nb) {  // nb <= 25
        if (bitpos + nb > wbit + 64) fill(bitpos >> 3);
        return (uint32_t)(win >> (bitpos - wbit)) & ((1u << nb) - 1u);
    }
    __device__ __forceinline__ uint32_t read(uint32_t nb) { const uint32_t v = peek(nb); bitpos += nb; return v; }
};

#define BX_NORM(i) S.norm[(i) * 64 + lane]
#define BX_NXT(i) S.nxt[(i) * 64 + lane]
#define BX_W(i) S.weights[(i) * 64 + lane]

// fse_read_ncount for one lane (RFC 8878 4.1.1); at most 64 symbols (what a lane's scratch holds)
__device__ int bx_read_ncount(BxScratch &S, const uint32_t lane, const uint8_t *src, uint32_t n, int max_log, int max_sym, int *nsym,
                              int *log, uint32_t *consumed) {
    if (n == 0) return E_TRUNC;
    FwdW b;
    b.init(src, n);
    const int alog = 5 + (int)b.read(4);
    if (alog > max_log) return E_CORRUPT;
    int remaining = 1 << alog, s = 0;
    while (remaining > 0 && s <= max_sym) {
        const int bits = hibit((uint32_t)remaining + 1) + 1;
        uint32_t val = b.peek((uint32_t)bits);
        const uint32_t lower_mask = (1u << (bits - 1)) - 1, threshold = (1u << bits) - 1 - ((uint32_t)remaining + 1);
        if ((val & lower_mask) < threshold) { b.bitpos += (uint32_t)bits - 1; val &= lower_mask; }
        else { b.bitpos += (uint32_t)bits; if (val > lower_mask) val -= threshold; }
        const int proba = (int)val - 1;
        remaining -= proba < 0 ? -proba : proba;
        if (s >= 64) return E_UNSUP;
        BX_NORM(s++) = (int16_t)proba;
        if (proba == 0) {
            uint32_t rep = b.read(2);
            for (;;) {
                for (uint32_t i = 0; i < rep && s <= max_sym; i++) { if (s >= 64) return E_UNSUP; BX_NORM(s++) = 0; }
                if (rep == 3) rep = b.read(2); else break;
            }
        }
    }
    if (remaining != 0) return E_CORRUPT;
    if ((b.bitpos + 7) / 8 > n) return E_TRUNC;
    *nsym = s;
    *log = alog;
    *consumed = (b.bitpos + 7) / 8;
    return 0;
}

// Spread of the symbols over the table cells (RFC 8878 4.1.1): put(cell, symbol); the lane's next-state counters are set.
template <class Put>
__device__ __forceinline__ int bx_fse_spread(BxScratch &S, const uint32_t lane, int nsym, int log, Put put) {
    const int size = 1 << log;
    int high = size;
    for (int s = 0; s < nsym; s++)
        if (BX_NORM(s) == -1) { put((uint32_t)--high, (uint32_t)s); BX_NXT(s) = 1; }
    const int step = (size >> 1) + (size >> 3) + 3, mask = size - 1;
    int pos = 0;
    for (int s = 0; s < nsym; s++) {
        const int c = BX_NORM(s);
        if (c <= 0) continue;
        BX_NXT(s) = (uint16_t)c;
        for (int i = 0; i < c; i++) {
            put((uint32_t)pos, (uint32_t)s);
            do { pos = (pos + step) & mask; } while (pos >= high);
        }
    }
    return pos == 0 ? 0 : E_CORRUPT;
}

// FSE pool cell (2 bytes): symbol:6 | ns:10 << 6, ns = the state counter the table construction hands the cell
// (RFC 8878 4.1.1): the bits to read are log - hibit(ns), the next state's base is (ns << bits) - (1 << log); base value and
// extra bits follow from the symbol.  Half the size of a cell that spells these out: 64 blocks' tables fit one wave's LDS.
__device__ __forceinline__ int bx_check_sym(int kind, uint32_t sym) {
    return sym > (kind == K_LL ? 35u : (kind == K_ML ? 52u : 31u)) ? E_CORRUPT : 0;
}

// One sequence table of one lane's block, from its counts (BX_NORM) into cells[0 .. 1 << log) of the FSE pool (cells is
// 16-byte aligned and has room for a multiple of 8 cells).
__device__ int bx_build_seq_table(BxScratch &S, const uint32_t lane, int nsym, int log, int kind, uint16_t *cells) {
    int rc = bx_fse_spread(S, lane, nsym, log, [&](uint32_t u, uint32_t sym) { S.sym[u * 64 + lane] = (uint8_t)sym; });
    if (rc) return rc;
    for (int s_ = 0; s_ < nsym; s_++)
        if (BX_NORM(s_) != 0 && bx_check_sym(kind, (uint32_t)s_)) return E_CORRUPT;
    const uint32_t size = 1u << log;
    for (uint32_t u0 = 0; u0 < size; u0 += 8) {  // (a table has at least 32 cells)
        uint32_t w[4];
#pragma unroll
        for (uint32_t q = 0; q < 8; q++) {
            const uint32_t sym = S.sym[(u0 + q) * 64 + lane];
            const uint32_t ns = BX_NXT(sym);
            BX_NXT(sym) = (uint16_t)(ns + 1);
            const uint32_t c = sym | (ns << 6);
            if (q & 1) w[q >> 1] |= c << 16; else w[q >> 1] = c;
        }
        *reinterpret_cast<uint4 *>(cells + u0) = make_uint4(w[0], w[1], w[2], w[3]);
    }
    return 0;
}

// Huffman tree description -> this lane's weights (BX_W), checked; *consumed = bytes of the description, *hlog = the
// table's log, *nsym_out = symbols (the implied last one included).
__device__ int bx_read_weights(BxScratch &S, const uint32_t lane, const uint8_t *src, uint32_t n, const uint8_t *blob_end,
                               uint32_t *consumed, uint32_t *hlog, uint32_t *nsym_out) {
    if (n < 1) return E_TRUNC;
    const uint32_t hb = src[0];
    uint32_t nw = 0;
    if (hb >= 128) {
        nw = hb - 127;
        const uint32_t bytes = (nw + 1) / 2;
        if (1 + bytes > n) return E_TRUNC;
        for (uint32_t i = 0; i < nw; i++) {
            const uint8_t b = src[1 + i / 2];
            BX_W(i) = (i & 1) ? (b & 15) : (b >> 4);
        }
        *consumed = 1 + bytes;
    } else {
        if (hb == 0 || 1 + hb > n) return E_TRUNC;
        int nsym = 0, log = 0;
        uint32_t hdr = 0;
        int rc = bx_read_ncount(S, lane, src + 1, hb, 6, 255, &nsym, &log, &hdr);
        if (rc) return rc;
        rc = bx_fse_spread(S, lane, nsym, log, [&](uint32_t u, uint32_t sym) { S.wtab[u * 64 + lane] = (uint16_t)sym; });
        if (rc) return rc;
        const uint32_t size = 1u << log;
        for (uint32_t u = 0; u < size; u++) {
            const uint32_t sym = S.wtab[u * 64 + lane];
            const uint32_t ns = BX_NXT(sym);
            BX_NXT(sym) = (uint16_t)(ns + 1);
            const uint32_t nb = (uint32_t)log - (uint32_t)hibit(ns);
            S.wtab[u * 64 + lane] = (uint16_t)((sym > 15 ? 15u : sym) | (nb << 4) | (((ns << nb) - size) << 7));  // a weight above 12 is rejected below
        }
        if (hdr >= hb) return E_CORRUPT;
        BitR b;
        if (!b.init(src + 1 + hdr, hb - hdr, blob_end)) return E_CORRUPT;
        uint32_t s1 = b.read((uint32_t)log), s2 = b.read((uint32_t)log);
        for (;;) {
            const uint32_t e1 = S.wtab[s1 * 64 + lane];
            if (nw >= 255) return E_CORRUPT;
            BX_W(nw++) = (uint8_t)(e1 & 15);
            s1 = (e1 >> 7) + b.read((e1 >> 4) & 7);
            const uint32_t e2 = S.wtab[s2 * 64 + lane];
            if (b.pos < 0) {
                if (nw >= 255) return E_CORRUPT;
                BX_W(nw++) = (uint8_t)(e2 & 15);
                break;
            }
            if (nw >= 255) return E_CORRUPT;
            BX_W(nw++) = (uint8_t)(e2 & 15);
            s2 = (e2 >> 7) + b.read((e2 >> 4) & 7);
            if (b.pos < 0) {
                if (nw >= 255) return E_CORRUPT;
                BX_W(nw++) = (uint8_t)(S.wtab[s1 * 64 + lane] & 15);
                break;
            }
        }
        *consumed = 1 + hb;
    }
    // implied last weight, code lengths, first cell of every rank
    uint32_t total = 0;
    for (uint32_t i = 0; i < nw; i++) {
        const uint32_t w = BX_W(i);
        if (w > 12) return E_CORRUPT;
        total += w ? 1u << (w - 1) : 0;
    }
    if (total == 0) return E_CORRUPT;
    const uint32_t maxbits = (uint32_t)hibit(total) + 1;
    if (maxbits > 11) return E_CORRUPT;
    const uint32_t left = (1u << maxbits) - total;
    if (left & (left - 1)) return E_CORRUPT;
    BX_W(nw) = (uint8_t)(hibit(left) + 1);
    const uint32_t nsym = nw + 1;
    for (uint32_t i = 0; i < 16; i++) S.rank[i * 64 + lane] = 0;
    for (uint32_t i = 0; i < nsym; i++) {
        const uint32_t w = BX_W(i);
        if (w) S.rank[(maxbits + 1 - w) * 64 + lane] += 1;  // count per code length
    }
    {   // counts -> first cells, longest codes first (as the table is laid out)
        uint32_t start = 0;
        for (uint32_t bits = maxbits; bits >= 1; bits--) {
            const uint32_t cnt = S.rank[bits * 64 + lane];
            S.rank[bits * 64 + lane] = (uint16_t)start;
            start += cnt << (maxbits - bits);
        }
        if (start != (1u << maxbits)) return E_CORRUPT;
    }
    *hlog = maxbits;
    *nsym_out = nsym;
    return 0;
}

// the lane's weights and rank starts (bx_read_weights) -> the decoding table t[0 .. 1 << maxbits), t 16-byte aligned
__device__ void bx_fill_huf(BxScratch &S, const uint32_t lane, uint32_t nsym, uint32_t maxbits, uint16_t *t) {
    for (uint32_t sym = 0; sym < nsym; sym++) {
        const uint32_t w = BX_W(sym);
        if (!w) continue;
        const uint32_t bits = maxbits + 1 - w, len = 1u << (w - 1);
        const uint32_t st = S.rank[bits * 64 + lane];
        S.rank[bits * 64 + lane] = (uint16_t)(st + len);
        const uint32_t e = sym | (bits << 8), e2 = e | (e << 16);
        if (len >= 8) {
            const uint4 v = make_uint4(e2, e2, e2, e2);
            for (uint32_t i = 0; i < len; i += 8) *reinterpret_cast<uint4 *>(t + st + i) = v;  // st is a multiple of len
        } else if (len == 4) *reinterpret_cast<uint2 *>(t + st) = make_uint2(e2, e2);
        else if (len == 2) *reinterpret_cast<uint32_t *>(t + st) = e2;
        else t[st] = (uint16_t)e;
    }
}

// What the Sequences_Section_Header of the section [q, q + n) says about the kinds in `want` (bit 0 LL, 1 OF, 2 ML):
// mode[k] 0 predefined, 1 RLE (sym[k]), 2 described at desc[k] (log[k]); kinds in Repeat_Mode come back in *missing.
struct BxSeqHdr {
    uint32_t mode[3], log[3], sym[3];
    const uint8_t *desc[3];
    uint32_t desc_n[3];
};
__device__ int bx_seq_locate(BxScratch &S, const uint32_t lane, const uint8_t *q, uint32_t n, uint32_t want, uint32_t *missing,
                             uint32_t *nseq_out, uint32_t *bits_at, BxSeqHdr &H) {
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
    uint32_t miss = 0;
    for (int k = 0; k < 3; k++) {
        const uint32_t mode = (modes >> (6 - 2 * k)) & 3;
        const bool wanted = (want >> k) & 1;
        const int kind = k == 0 ? K_LL : (k == 1 ? K_OF : K_ML);
        if (mode == 0) {
            if (wanted) { H.mode[k] = 0; H.log[k] = k == 1 ? 5 : 6; }
        } else if (mode == 1) {
            if (p >= n) return E_TRUNC;
            if (wanted) {
                const uint32_t sym = q[p];
                const int rc = bx_check_sym(kind, sym);
                if (rc) return rc;
                H.mode[k] = 1; H.log[k] = 0; H.sym[k] = sym;
            }
            p++;
        } else if (mode == 2) {
            int nsym = 0, log = 0;
            uint32_t used = 0;
            const int rc = bx_read_ncount(S, lane, q + p, n - p, k == 1 ? 8 : 9, k == 0 ? 35 : (k == 1 ? 31 : 52), &nsym, &log, &used);
            if (rc) return rc;
            if (wanted) { H.mode[k] = 2; H.log[k] = (uint32_t)log; H.desc[k] = q + p; H.desc_n[k] = n - p; }
            p += used;
        } else if (wanted) miss |= 1u << k;
    }
    *missing = miss;
    *bits_at = p;
    return 0;
}

__global__ __launch_bounds__(64) void k_bx_scan(BxArgs a) {
    const uint32_t lane = threadIdx.x;
    const uint32_t n_pend = *a.pending_count;
    const uint32_t n_slots = a.n_list_a + n_pend + a.n_bc;
    if (blockIdx.x == 0 && lane == 0) a.ctr[0] = n_slots;
    for (uint32_t t0 = blockIdx.x * 64; t0 < n_slots; t0 += gridDim.x * 64) {
        const uint32_t t = t0 + lane;
        bool take = t < n_slots;
        uint32_t row = 0xFFFFFFFFu;
        if (take) {
            if (t < a.n_list_a) row = a.list_a[t];
            else if (t < a.n_list_a + n_pend) row = a.pending[t - a.n_list_a];
            else { row = a.bc_row[t - a.n_list_a - n_pend]; take = a.row_flag[row] != 0; }
            if (take && a.preset && a.status[row] < 0) take = false;  // the host has ruled on this row
        }
        uint32_t nb = 0;
        uint64_t first = 0, n = 0;
        const uint8_t *src = a.blobs;
        if (take) {
            a.row_flag[row] = 1;
            src = a.blobs + (a.blob_off[row] - a.blob_base);
            n = a.blob_size[row];
            const uint64_t fcs_want = a.usize[row];
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
            first = pos;
            uint32_t k = 0;
            bool last = false;
            while (ok && !last) {
                if (pos + 3 > n) { ok = false; break; }
                const uint32_t bh = src[pos] | (src[pos + 1] << 8) | (src[pos + 2] << 16);
                const uint32_t type = (bh >> 1) & 3, size = bh >> 3;
                const uint64_t step = 3 + (type == 1 ? 1 : size);
                if (type == 3 || size > BLOCK_MAX || pos + step > n) { ok = false; break; }
                pos += step;
                last = bh & 1;
                k++;
            }
            ok = ok && pos == n;
            nb = ok ? k : 0;
        }
        const uint32_t base = wave_alloc32(&a.ctr[1], nb, lane);
        if (nb && base + nb > a.item_cap) nb = 0;  // out of item slots: the serial decoder keeps the frame
        if (nb) {
            uint64_t pos = first;
            for (uint32_t k = 0; k < nb; k++) {
                const uint32_t bh = src[pos] | (src[pos + 1] << 8) | (src[pos + 2] << 16);
                a.items[base + k].src = (uint32_t)pos;
                a.prep[base + k].frame = t;
                a.prep[base + k].k = k;
                pos += 3 + (((bh >> 1) & 3) == 1 ? 1 : (bh >> 3));
            }
        }
        if (t < n_slots) { a.cand_row[t] = take ? row : 0xFFFFFFFFu; a.cand_base[t] = base; a.cand_nb[t] = nb; }
    }
}

__global__ __launch_bounds__(64) void k_bx_prep(BxArgs a) {
    __shared__ BxScratch S;
    const uint32_t lane = threadIdx.x;
    const uint32_t n_items = a.ctr[1] < a.item_cap ? a.ctr[1] : a.item_cap;
    for (uint32_t i0 = blockIdx.x * 64; i0 < n_items; i0 += gridDim.x * 64) {
        const uint32_t slot = i0 + lane;
        bool on = slot < n_items;
        if (on) {
            // A slot is a block only if the scan filled it.  When the item slots run out, the frames that no longer fit keep
            // nb = 0, but the counter has moved past them: the slots between the last frame that fitted and the capacity hold
            // whatever the memory held before (found by tools/soak_foreign.py: a stale frame index -> a read far outside the
            // table's columns).
            const uint32_t c = a.prep[slot].frame, k = a.prep[slot].k;
            on = c < a.ctr[0] && k < a.cand_nb[c] && a.cand_base[c] + k == slot;
        }
        unsigned long long t_p = a.dbg ? __builtin_amdgcn_s_memtime() : 0;  // diagnostic (ZNIPPY_DDBG): where the wave's time goes
#define PSTAMP(i) do { if (a.dbg) { const unsigned long long now_ = __builtin_amdgcn_s_memtime(); if (lane == 0) atomicAdd(&a.dbg[i], now_ - t_p); t_p = now_; } } while (0)
        if (a.dbg && lane == 0) atomicAdd(&a.dbg[32], 1ull);
        // ---- phase 1, lane = block: everything the headers say, and how much of each pool the block needs ----
        FzItem it;
        it.src = 0; it.out = 0; it.nseq = 0; it.lit_len = 0; it.seq_off = 0; it.lit_off = 0; it.lit_kind = 0; it.err = 0;
        it.rep[0] = FZ_SYM; it.rep[1] = FZ_SYM | (1u << 26); it.rep[2] = FZ_SYM | (2u << 26); it.pad = 0;
        BxPrep pr;
        pr.frame = 0; pr.k = 0; pr.n_streams = 0; pr.huf_off = 0; pr.huf_log = 0; pr.logs = 0; pr.bs_off = 0; pr.bs_len = 0;
        pr.tab[0] = pr.tab[1] = pr.tab[2] = 0;
        for (int j = 0; j < 4; j++) { pr.st_off[j] = 0; pr.st_len[j] = 0; }
        BxSeqHdr H;
        for (int j = 0; j < 3; j++) { H.mode[j] = 0; H.log[j] = 0; H.sym[j] = 0; H.desc[j] = nullptr; H.desc_n[j] = 0; }
        int err = 0;
        uint32_t why = 0, hsyms = 0;
        uint32_t need_lit = 0, need_seq = 0, need_fse = 0, need_huf = 0;
        if (on) {
            const BxPrep pr0 = a.prep[slot];
            pr.frame = pr0.frame; pr.k = pr0.k;
            const uint32_t c = pr0.frame, k = pr0.k, base = slot - k, row = a.cand_row[c];
            const uint8_t *const src = a.blobs + (a.blob_off[row] - a.blob_base);
            const uint8_t *const blob_end = src + a.blob_size[row];
            const uint32_t pos = a.items[slot].src;
            const uint32_t bh = src[pos] | (src[pos + 1] << 8) | (src[pos + 2] << 16);
            const uint32_t btype = (bh >> 1) & 3, bsize = bh >> 3;
            const uint8_t *const bsrc = src + pos + 3;
            it.src = pos;
            if (btype != 2) {  // raw / RLE block: literals only
                it.out = bsize; it.lit_len = bsize;
                it.lit_kind = btype == 0 ? 0u : 1u;
                it.lit_off = btype == 0 ? (uint64_t)pos + 3 : (uint64_t)bsrc[0];
            } else {
                LitHdr h;
                err = fz_lit_header(bsrc, bsize, h);
                uint32_t seq_pos = 0;
                if (!err) {
                    it.lit_len = h.regen;
                    seq_pos = fz_lit_section_bytes(h);
                    if (h.type == 0) { it.lit_kind = 0; it.lit_off = (uint64_t)pos + 3 + h.hdr; }
                    else if (h.type == 1) { it.lit_kind = 1; it.lit_off = bsrc[h.hdr]; }
                    else {
                        it.lit_kind = 2;
                        uint32_t p = h.hdr, remain = h.comp;
                        if (h.type == 2) {
                            uint32_t tu = 0;
                            err = bx_read_weights(S, lane, bsrc + p, remain, blob_end, &tu, &pr.huf_log, &hsyms);
                            if (!err) { p += tu; remain -= tu; }
                        } else {  // treeless: the tree of the nearest earlier block of the frame that carries a description (re-read)
                            err = E_UNSUP; why = 1;
                            for (uint32_t back = 1; back <= FZ_BACK && back <= k; back++) {
                                const uint32_t pj = a.items[base + k - back].src;
                                const uint32_t bj = src[pj] | (src[pj + 1] << 8) | (src[pj + 2] << 16);
                                if (((bj >> 1) & 3) != 2) continue;
                                LitHdr hj;
                                if (fz_lit_header(src + pj + 3, bj >> 3, hj)) { err = E_CORRUPT; break; }
                                if (hj.type == 2) {
                                    uint32_t tu = 0;
                                    err = bx_read_weights(S, lane, src + pj + 3 + hj.hdr, hj.comp, blob_end, &tu, &pr.huf_log, &hsyms);
                                    why = 0;
                                    break;
                                }
                            }
                        }
                        if (!err) {
                            const uint32_t regen = h.regen;
                            if (h.streams == 1) { pr.st_off[0] = pos + 3 + p; pr.st_len[0] = remain; }
                            else {
                                const uint32_t seg = (regen + 3) / 4;
                                if (remain < 6 || 3 * seg > regen) err = E_CORRUPT;
                                else {
                                    const uint32_t s1 = bsrc[p] | (bsrc[p + 1] << 8), s2 = bsrc[p + 2] | (bsrc[p + 3] << 8), s3 = bsrc[p + 4] | (bsrc[p + 5] << 8);
                                    if (6 + s1 + s2 + s3 > remain) err = E_CORRUPT;
                                    else {
                                        const uint32_t o = pos + 3 + p + 6;
                                        pr.st_off[0] = o; pr.st_len[0] = s1;
                                        pr.st_off[1] = o + s1; pr.st_len[1] = s2;
                                        pr.st_off[2] = o + s1 + s2; pr.st_len[2] = s3;
                                        pr.st_off[3] = o + s1 + s2 + s3; pr.st_len[3] = remain - 6 - s1 - s2 - s3;
                                    }
                                }
                            }
                            pr.n_streams = err ? 0 : h.streams;
                        }
                        if (!err) {
                            need_lit = (h.regen + 79) & ~15u;
                            need_huf = (1u << pr.huf_log) < 8 ? 8u : 1u << pr.huf_log;  // 16-byte granules
                        }
                    }
                }
                PSTAMP(33);
                // ---- sequences: header, where the table descriptions and the bitstream are ----
                if (!err) {
                    if (seq_pos >= bsize) err = E_TRUNC;
                    uint32_t miss = 0, nseq = 0, bits_at = 0;
                    const uint8_t *q = bsrc + seq_pos;
                    const uint32_t qn = bsize - seq_pos;
                    if (!err) err = bx_seq_locate(S, lane, q, qn, 7u, &miss, &nseq, &bits_at, H);
                    if (!err && nseq == 0 && bits_at != qn) err = E_CORRUPT;
                    if (!err && nseq && miss) {
                        for (uint32_t back = 1; back <= FZ_BACK && back <= k && miss && !err; back++) {
                            const uint32_t pj = a.items[base + k - back].src;
                            const uint32_t bj = src[pj] | (src[pj + 1] << 8) | (src[pj + 2] << 16);
                            if (((bj >> 1) & 3) != 2) continue;
                            const uint8_t *b = src + pj + 3;
                            const uint32_t sz = bj >> 3;
                            LitHdr hj;
                            if (fz_lit_header(b, sz, hj)) { err = E_CORRUPT; break; }
                            const uint32_t ls = fz_lit_section_bytes(hj);
                            if (ls >= sz) { err = E_CORRUPT; break; }
                            uint32_t m2 = 0, n2 = 0, at2 = 0;
                            err = bx_seq_locate(S, lane, b + ls, sz - ls, miss, &m2, &n2, &at2, H);
                            if (!err && n2) miss = m2;
                        }
                        if (!err && miss) { err = E_UNSUP; why = 1; }
                    }
                    if (!err && nseq) {
                        if (bits_at >= qn) err = E_TRUNC;
                        else if (q[qn - 1] == 0) err = E_CORRUPT;
                        else { pr.bs_off = pos + 3 + seq_pos + bits_at; pr.bs_len = qn - bits_at; }
                    }
                    if (!err && nseq) {
                        need_seq = (nseq + 7) & ~7u;  // records leave the sequence kernel eight at a time
                        for (int j = 0; j < 3; j++) need_fse += H.mode[j] == 0 ? 0u : ((1u << H.log[j]) < 8 ? 8u : 1u << H.log[j]);
                    }
                    pr.logs = H.log[0] | (H.log[1] << 8) | (H.log[2] << 16);
                    it.nseq = err ? 0 : nseq;
                    if (!err && nseq == 0) it.out = it.lit_len;
                }
            }
            if (err) { need_lit = need_seq = need_fse = need_huf = 0; }
        }
        PSTAMP(34);
        // ---- pool space: one atomic per wave and pool (all lanes, converged) ----
        const uint64_t o_lit = wave_alloc64(&a.pool_used[0], need_lit, lane), o_seq = wave_alloc64(&a.pool_used[1], need_seq, lane);
        const uint64_t o_fse = wave_alloc64(&a.pool_used[8], need_fse, lane) + BX_POOL_FIRST, o_huf = wave_alloc64(&a.pool_used[9], need_huf, lane);
        // ---- phase 2, lane = block: the decoding tables ----
        bool huf = false, seq = false;
        uint32_t cls = 0;
        if (on) {
            if (!err && (o_lit + need_lit > a.lit_cap || o_seq + need_seq > a.seq_cap || o_fse + need_fse > a.fse_cap || o_huf + need_huf > a.huf_cap)) { err = E_UNSUP; why = 2; }
            if (!err && need_huf) {
                pr.huf_off = (uint32_t)o_huf;
                bx_fill_huf(S, lane, hsyms, pr.huf_log, a.huf_pool + o_huf);
            }
            if (!err && need_lit) it.lit_off = o_lit;
            PSTAMP(35);
            if (!err && it.nseq) {
                it.seq_off = o_seq;
                uint32_t at = (uint32_t)o_fse;
                for (int j = 0; j < 3 && !err; j++) {
                    const int kind = j == 0 ? K_LL : (j == 1 ? K_OF : K_ML);
                    if (H.mode[j] == 0) pr.tab[j] = j == 0 ? BX_PREDEF_LL : (j == 1 ? BX_PREDEF_OF : BX_PREDEF_ML);
                    else if (H.mode[j] == 1) { a.fse_pool[at] = (uint16_t)(H.sym[j] | (1u << 6)); pr.tab[j] = at; at += 8; }  // one cell: log 0, ns = 1 -> no bits, next state 0
                    else {
                        int nsym = 0, log = 0;
                        uint32_t used = 0;
                        err = bx_read_ncount(S, lane, H.desc[j], H.desc_n[j], j == 1 ? 8 : 9, j == 0 ? 35 : (j == 1 ? 31 : 52), &nsym, &log, &used);
                        if (!err) err = bx_build_seq_table(S, lane, nsym, log, kind, a.fse_pool + at);
                        pr.tab[j] = at; at += 1u << log;
                    }
                }
                if (err) it.nseq = 0;
            }
            PSTAMP(36);
            it.err = err;
            if (err) { atomicAdd(&a.pool_used[3], 1ull); atomicAdd(&a.pool_used[4 + (why & 3)], 1ull); }  // statistics
            a.items[slot] = it;
            a.prep[slot] = pr;
            huf = !err && pr.n_streams != 0;
            seq = !err && it.nseq != 0;
            {   // the wave shape of the sequence kernel by the size of the three tables: 64 / 32 / 16 blocks per wave
                uint32_t cells = 0;
                for (int q = 0; q < 3; q++) { const uint32_t c = 1u << ((pr.logs >> (8 * q)) & 255); cells += c < 8 ? 8u : c; }
                cls = cells <= 384 ? 0u : (cells <= 768 ? 1u : 2u);
                // a long chain gets a wave of its own (k_bx_fse_wave); so does a stream that starts within 64 bytes of the blob
                // region's first byte (the lane kernel's buffer loads reach 64 bytes in front of the stream, the wave decoder's do not)
                // (a.big_seq == 0: the long chains are picked from the table's histogram afterwards, k_bx_split)
                if ((a.big_seq && it.nseq >= a.big_seq) || (a.blob_off[a.cand_row[pr.frame]] - a.blob_base) + pr.bs_off < 64) cls = 3;
            }
        }
        {   // the block joins the lists of the two entropy kernels
            const uint64_t hm = __ballot(huf);
            const uint64_t below = lane ? (~0ull >> (64 - lane)) : 0ull;
            uint32_t hb = 0;
            if (lane == 0 && hm) hb = atomicAdd(&a.ctr[2], (uint32_t)__popcll(hm));
            hb = rdlane_u(hb, 0);
            if (huf) a.huf_list[hb + (uint32_t)__popcll(hm & below)] = slot;
            for (uint32_t q = 0; q < 4; q++) {
                const bool mine = seq && cls == q;
                const uint64_t sm = __ballot(mine);
                uint32_t sb = 0;
                if (lane == 0 && sm) sb = atomicAdd(&a.ctr[q == 0 ? 3 : 4 + q], (uint32_t)__popcll(sm));
                sb = rdlane_u(sb, 0);
                if (mine) a.seq_list[(size_t)q * a.item_cap + sb + (uint32_t)__popcll(sm & below)] = slot;
            }
        }
    }
}

// Which blocks get a wave of their own in the sequence stage?  A chain runs ~4x faster there (0.23 against ~1.1 us per
// sequence when the wave is alone on its SIMD) but costs the chip ~1,100 SIMD cycles per sequence (measured: 19.9 M sequences
// of 8,192 blocks, 16 waves per CU: 10.6 ms = 0.55 us per sequence per SIMD), where the lane kernel does 64 sequences in
// ~2,000.  So: the longest blocks, as many as keep the wave kernel's work below the lane kernel's pole —
// sum(nseq >= T) * 0.55 us / 1,024 SIMDs <= T * 1.1 us, i.e. sum <= 2,048 T — with T taken from a histogram of the table's
// blocks (8 classes per octave).  A table of 8,192 blocks of ~2,400 sequences: T ~ 3,000 (10.6 -> ~3.5 ms); the image's
// source text (5,500 blocks, a few of 15,000 sequences): T ~ 1,000.  One workgroup per lane list: the histogram over all
// three, then its own list compacted in place, the long ones appended to the wave list.
__global__ __launch_bounds__(1024) void k_bx_split(BxArgs a) {
    __shared__ uint32_t hist[256], wcnt[16], sT, skept;
    const uint32_t which = blockIdx.x, t = threadIdx.x, lane = t & 63, wv = t >> 6;
    const uint32_t cidx[3] = {3, 5, 6};
    if (t < 256) hist[t] = 0;
    __syncthreads();
    for (uint32_t l = 0; l < 3; l++) {
        const uint32_t n = a.ctr[cidx[l]] < a.item_cap ? a.ctr[cidx[l]] : a.item_cap;
        const uint32_t *const list = a.seq_list + (size_t)l * a.item_cap;
        for (uint32_t i0 = 0; i0 < n; i0 += 1024) {  // (whole waves: the classes of a wave's 64 blocks are counted together —
            const uint32_t i = i0 + t;                 //  a table of like blocks is 100,000 additions to one LDS word otherwise)
            uint32_t k = 0xFFFFFFFFu;
            if (i < n) {
                const uint32_t v = a.items[list[i]].nseq;
                const uint32_t hb = v ? 31u - (uint32_t)__clz(v) : 0u;
                k = v < 16 ? v : 8 * hb + ((v >> (hb - 3)) & 7u);
                k = k > 255u ? 255u : k;
            }
            uint64_t left = __ballot(k != 0xFFFFFFFFu);
            while (left) {
                const uint32_t k0 = (uint32_t)__builtin_amdgcn_readlane((int)k, (int)((uint32_t)__ffsll((long long)left) - 1));
                const uint64_t same = __ballot(k == k0);
                if (lane == (uint32_t)__ffsll((long long)left) - 1) atomicAdd(&hist[k0], (uint32_t)__popcll(same));
                left &= ~same;
            }
        }
    }
    __syncthreads();
    if (t == 0) {
        unsigned long long S = 0;
        uint32_t T = 0xFFFFFFFFu;
        for (int k = 255; k >= 16; k--) {
            const uint32_t cn = hist[k];
            if (!cn) continue;
            const uint32_t hb = (uint32_t)k >> 3, lb = (8u + ((uint32_t)k & 7u)) << (hb - 3);  // smallest nseq of the class
            if (lb < 256) break;
            S += (unsigned long long)cn * (lb + (lb >> 4));
            if (S > 2048ull * lb) break;
            T = lb;
        }
        sT = T; skept = 0;
    }
    __syncthreads();
    const uint32_t T = sT;
    uint32_t *const list = a.seq_list + (size_t)which * a.item_cap;
    uint32_t *const wave_list = a.seq_list + 3 * (size_t)a.item_cap;
    const uint32_t n = a.ctr[cidx[which]] < a.item_cap ? a.ctr[cidx[which]] : a.item_cap;
    if (T == 0xFFFFFFFFu) return;  // (uniform) nothing is long enough
    for (uint32_t base = 0; base < n; base += 1024) {
        const uint32_t i = base + t;
        const uint32_t slot = i < n ? list[i] : 0;
        const bool mv = i < n && a.items[slot].nseq >= T, keep = i < n && !mv;
        const uint64_t km = __ballot(keep), mm = __ballot(mv);
        const uint64_t below = lane ? (~0ull >> (64 - lane)) : 0ull;
        if (lane == 0) wcnt[wv] = (uint32_t)__popcll(km);
        uint32_t mb = 0;
        if (lane == 0 && mm) mb = atomicAdd(&a.ctr[7], (uint32_t)__popcll(mm));
        mb = (uint32_t)__builtin_amdgcn_readfirstlane((int)mb);
        __syncthreads();  // every thread has read its entry; the waves' counts are in place
        uint32_t before = skept;
        for (uint32_t w = 0; w < wv; w++) before += wcnt[w];
        uint32_t tile = 0;
        for (uint32_t w = 0; w < 16; w++) tile += wcnt[w];
        if (keep) list[before + (uint32_t)__popcll(km & below)] = slot;
        if (mv) { const uint32_t at = mb + (uint32_t)__popcll(mm & below); if (at < a.item_cap) wave_list[at] = slot; }
        __syncthreads();
        if (t == 0) skept += tile;
        __syncthreads();
    }
    if (t == 0) a.ctr[cidx[which]] = skept;
}

// The entropy kernels run a wave for as long as its longest lane: the work lists are ordered by size (descending: the long
// ones start first), 8 size classes per octave.  One workgroup per list: histogram, scan, scatter through `tmp`, copy back.
__global__ __launch_bounds__(1024) void k_bx_sort(BxArgs a, uint32_t *tmp) {
    __shared__ uint32_t hist[256], base[256];
    const uint32_t which = blockIdx.x, t = threadIdx.x;  // 0 Huffman list, 1..4 sequence lists
    uint32_t *const list = which == 0 ? a.huf_list : a.seq_list + (size_t)(which - 1) * a.item_cap;
    uint32_t *const out = tmp + (size_t)which * a.item_cap;
    const uint32_t n = which == 0 ? a.ctr[2] : a.ctr[which == 1 ? 3 : 3 + which];  // ctr[2], [3], [5], [6], [7]
    if (n < 128) return;  // (nothing to balance)
    auto key_of = [&](uint32_t slot) -> uint32_t {
        const uint32_t v = which == 0 ? a.items[slot].lit_len : a.items[slot].nseq;
        const uint32_t hb = v ? 31u - (uint32_t)__clz(v) : 0u;
        const uint32_t k = v < 16 ? v : 8 * hb + ((v >> (hb - 3)) & 7u) - 16 + 16;  // 16.. : (octave, top three bits below the leading one)
        return 255u - (k > 255u ? 255u : k);  // descending
    };
    if (t < 256) hist[t] = 0;
    __syncthreads();
    for (uint32_t i = t; i < n; i += 1024) atomicAdd(&hist[key_of(list[i])], 1u);
    __syncthreads();
    if (t < 64) {  // exclusive scan of the 256 counters: four per lane
        uint32_t c[4], sum = 0;
        for (int q = 0; q < 4; q++) { c[q] = hist[4 * t + q]; sum += c[q]; }
        uint32_t incl = sum;
        for (int d = 1; d < 64; d <<= 1) { const uint32_t y = __shfl_up(incl, d); if (t >= (uint32_t)d) incl += y; }
        uint32_t at = incl - sum;
        for (int q = 0; q < 4; q++) { base[4 * t + q] = at; at += c[q]; }
    }
    __syncthreads();
    for (uint32_t i = t; i < n; i += 1024) {
        const uint32_t slot = list[i];
        out[atomicAdd(&base[key_of(slot)], 1u)] = slot;
    }
    __syncthreads();
    for (uint32_t i = t; i < n; i += 1024) list[i] = out[i];
}

// lane = Huffman stream; a wave takes 16 blocks of the list and keeps their decoding tables in LDS
// (Tried: a second form with 1,024- or 512-cell tables in 32 / 16 KB per wave for groups whose tables all fit, four / eight
// waves per CU — slower on the text archive, 1.10 / 1.20 against 0.96 ms: its tables are mostly the full 2,048 cells.)
constexpr uint32_t BX_HUF_BLOCKS = 16, BX_HUF_LDS = BX_HUF_BLOCKS * 2048;  // u16 cells
__global__ __launch_bounds__(64) void k_bx_huf(BxArgs a) {
    constexpr uint32_t STRIDE = 2048;
    __shared__ __attribute__((aligned(16))) uint16_t T[BX_HUF_LDS];
    typedef __attribute__((address_space(3))) uint16_t lds16;
    const uint32_t lane = threadIdx.x, j = lane >> 2, sidx = lane & 3;
    const uint32_t n_list = a.ctr[2];
    for (uint32_t g0 = blockIdx.x * BX_HUF_BLOCKS; g0 < n_list; g0 += gridDim.x * BX_HUF_BLOCKS) {
        const bool on = g0 + j < n_list;
        const uint32_t slot = on ? a.huf_list[g0 + j] : 0;
        BxPrep pr;
        pr.huf_log = 0; pr.huf_off = 0; pr.n_streams = 0; pr.frame = 0;
        if (on) pr = a.prep[slot];
        const uint32_t hlog = on ? pr.huf_log : 0;
        // tables -> LDS, 16 bytes per lane per step (a table is a whole number of 16-byte granules in the pool)
        __builtin_amdgcn_wave_barrier();
        for (uint32_t b = 0; b < BX_HUF_BLOCKS; b++) {
            const uint32_t cells = rdlane_u(on ? (1u << hlog) : 0u, 4 * b), off = rdlane_u(pr.huf_off, 4 * b);
            for (uint32_t i = lane * 8; i < cells; i += 512) {
                const uint4 v = *reinterpret_cast<const uint4 *>(a.huf_pool + off + i);
                *reinterpret_cast<uint4 *>(&T[b * STRIDE + i]) = v;
            }
        }
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_wave_barrier();
        if (on && sidx < pr.n_streams) {
            const uint32_t row = a.cand_row[pr.frame];
            const uint8_t *const src = a.blobs + (a.blob_off[row] - a.blob_base);
            const uint8_t *const blob_end = src + a.blob_size[row];
            const FzItem it = a.items[slot];
            const uint32_t regen = it.lit_len, seg = (regen + 3) / 4;
            const uint32_t n_out = pr.n_streams == 1 ? regen : (sidx < 3 ? seg : regen - 3 * seg);
            uint8_t *const dst = a.lit_pool + it.lit_off + (pr.n_streams == 1 ? 0 : sidx * seg);
            const uint8_t *const p = src + pr.st_off[sidx];
            const uint32_t n = pr.st_len[sidx];
            const lds16 *const huf = (const lds16 *)&T[j * STRIDE];
            int rc = 0;
            BitR b;
            if (!b.init(p, n, blob_end)) rc = E_CORRUPT;
            else {
                const uint32_t mask = (1u << hlog) - 1;
                uint32_t i = 0;
                int64_t pos = b.pos;
                // Far from the stream's first bytes: the 16 bytes the NEXT step needs are loaded while this step decodes.  A step
                // takes 4 .. 44 bits, so the next step's 44 bits lie inside the 16 bytes that end at byte ceil((pos - 4) / 8),
                // whatever this step consumes — the load's round trip runs beside the four dependent table lookups instead of
                // in front of them (the plain loop below: load, then lookups, ~1,300 cycles per four symbols).
                if (i + 4 <= n_out && pos >= 192) {
                    int64_t E = (pos + 7) >> 3;  // the window = stream bytes [E - 16, E)
                    uint4 W;
                    __builtin_memcpy(&W, p + E - 16, 16);
                    while (i + 4 <= n_out && pos >= 192) {
                        const int64_t En = (pos + 3) >> 3;
                        uint4 Wn;
                        __builtin_memcpy(&Wn, p + En - 16, 16);
                        const uint32_t s = (uint32_t)(8 * E - pos);  // bits of the window above the position: 0 .. 84
                        const uint64_t lo = (uint64_t)W.x | ((uint64_t)W.y << 32), hi = (uint64_t)W.z | ((uint64_t)W.w << 32);
                        uint64_t val = s < 64 ? (hi << s) | ((lo >> 1) >> (63 - s)) : lo << (s - 64);  // the next 64 bits, top-aligned
                        uint32_t w = 0, used = 0;
#pragma unroll
                        for (int q = 0; q < 4; q++) {
                            const uint32_t e = huf[(uint32_t)(val >> (64 - hlog))];
                            w |= (e & 0xFFu) << (8 * q);
                            val <<= e >> 8;
                            used += e >> 8;
                        }
                        pos -= used;
                        __builtin_memcpy(dst + i, &w, 4);
                        i += 4;
                        W = Wn;
                        E = En;
                    }
                }
                while (i + 4 <= n_out && pos >= 64) {  // four symbols (<= 44 bits) out of one 8-byte load, one 4-byte store
                    const int64_t b0 = ((pos + 7) >> 3) - 8;
                    uint64_t c;
                    __builtin_memcpy(&c, p + b0, 8);
                    int32_t avail = (int32_t)(pos - b0 * 8);
                    uint32_t w = 0;
#pragma unroll
                    for (int q = 0; q < 4; q++) {
                        const uint32_t e = huf[(uint32_t)(c >> (avail - (int32_t)hlog)) & mask];
                        w |= (e & 0xFFu) << (8 * q);
                        avail -= (int32_t)(e >> 8);
                    }
                    pos = b0 * 8 + avail;
                    __builtin_memcpy(dst + i, &w, 4);
                    i += 4;
                }
                b.pos = pos;
                b.refill();
                for (; i < n_out; i++) {
                    const uint32_t e = huf[b.peek(hlog)];
                    dst[i] = (uint8_t)e;
                    b.pos -= e >> 8;
                }
                if (b.pos != 0) rc = E_CORRUPT;
            }
            if (rc) a.items[slot].err = rc;
        }
        __builtin_amdgcn_wave_barrier();
    }
}

// the n (<= 32) bits of the 128-bit little-endian value hi:lo that start at bit s (s + n <= 128)
__device__ __forceinline__ uint32_t bx_ext(uint64_t lo, uint64_t hi, uint32_t s, uint32_t n) {
    const uint64_t v = s >= 64 ? hi >> (s & 63) : (lo >> s) | ((hi << 1) << (63 - s));
    return (uint32_t)v & (n >= 32 ? 0xFFFFFFFFu : (1u << n) - 1u);
}
__device__ __forceinline__ uint64_t bx_ext64(uint64_t lo, uint64_t hi, uint32_t s, uint32_t n) {  // n <= 63
    const uint64_t v = s >= 64 ? hi >> (s & 63) : (lo >> s) | ((hi << 1) << (63 - s));
    return v & ((1ull << n) - 1ull);
}

// wave = block, for the blocks of BX_BIG_SEQ sequences or more: a lane of k_bx_fse takes ~2,000 cycles per sequence (its
// ~350 instructions on a SIMD of its own), the two-stage wave decoder (fz_wave_sequences) ~700 — and with few blocks in the
// table, the longest chain is the kernel's time.
__global__ __launch_bounds__(64) void k_bx_fse_wave(BxArgs a, const uint32_t *list, const uint32_t *n_list_p) {
    __shared__ uint2 TL[512], TM[512], TO[256];
    const uint32_t lane = threadIdx.x;
    const uint32_t n_list = *n_list_p;
    for (;;) {
        uint32_t idx = 0;
        if (lane == 0) idx = atomicAdd(&a.ctr[8], 1u);
        idx = rdlane_u(idx, 0);
        if (idx >= n_list) break;
        const uint32_t slot = list[idx];
        const BxPrep pr = a.prep[slot];
        const FzItem it = a.items[slot];
        const uint32_t row = a.cand_row[pr.frame];
        const uint8_t *const src = a.blobs + (a.blob_off[row] - a.blob_base);
        const uint8_t *const blob_end = src + a.blob_size[row];
        __builtin_amdgcn_wave_barrier();
        for (int k = 0; k < 3; k++) {  // pool cells -> the wave decoder's entries
            const uint32_t log = (pr.logs >> (8 * k)) & 255, size = 1u << log;
            const uint16_t *const cells = a.fse_pool + pr.tab[k];
            uint2 *const t = k == 0 ? TL : (k == 1 ? TO : TM);
            for (uint32_t u = lane; u < size; u += 64) {
                const uint32_t c = cells[u], sym = c & 63u, ns = c >> 6;
                const uint32_t nb = log - (uint32_t)hibit(ns | 1u);
                uint32_t ab, base;
                if (k == 0) { ab = c_ll_bits[sym > 35 ? 35 : sym]; base = c_ll_base[sym > 35 ? 35 : sym]; }
                else if (k == 2) { ab = c_ml_bits[sym > 52 ? 52 : sym]; base = c_ml_base[sym > 52 ? 52 : sym]; }
                else { ab = sym; base = 1u << (sym & 31); }
                t[u] = make_uint2((((ns << nb) - size) & 0xFFFFu) | (nb << 16) | (ab << 24), base);
            }
        }
        __builtin_amdgcn_wave_barrier();
        const uint8_t *const bb = src + pr.bs_off;
        BitR b;
        int err = 0;
        uint32_t sl = 0, so = 0, sm = 0;
        if (!b.init(bb, pr.bs_len, blob_end)) err = E_CORRUPT;
        else {
            sl = b.read(pr.logs & 255); so = b.read((pr.logs >> 8) & 255); sm = b.read((pr.logs >> 16) & 255);
            if (b.pos < 0) err = E_CORRUPT;
        }
        uint32_t sum_ll = 0, sum_ml = 0, rr[3] = {FZ_SYM, FZ_SYM | (1u << 26), FZ_SYM | (2u << 26)}, why = 0;
        if (!err) err = fz_wave_sequences(TL, TO, TM, bb, blob_end, (int32_t)uni((uint32_t)b.pos), uni(sl), uni(so), uni(sm), uni(it.nseq),
                                          a.seq_pool + uni64(it.seq_off), lane, &sum_ll, &sum_ml, rr, &why);
        if (!err && sum_ll > it.lit_len) err = E_CORRUPT;
        if (!err && it.lit_len + sum_ml > BLOCK_MAX) err = E_CORRUPT;
        if (lane == 0) {
            a.items[slot].out = it.lit_len + sum_ml;
            a.items[slot].rep[0] = rr[0]; a.items[slot].rep[1] = rr[1]; a.items[slot].rep[2] = rr[2];
            if (err) { a.items[slot].err = err; atomicAdd(&a.pool_used[3], 1ull); atomicAdd(&a.pool_used[4 + 3], 1ull); }
        }
    }
}

// lane = block: the FSE sequence bitstream -> records.  LANES blocks per wave, by the size of their three tables (the
// tables live in the wave's LDS, every lane's in a region of its own: a lookup is a ds_read, not a gather from the pool).
// The stream reaches a lane through a 64-byte LDS buffer of its own, refilled every four sequences by loads that were
// issued four sequences earlier (all lanes at the same point of the loop: a lane that reloaded whenever it ran dry would
// make its whole wave wait out a memory round trip at nearly every step — vmcnt is in order, behind the record stores
// too); a lane whose sequences are so long that the buffer runs out before the next refill reloads on the spot.  Records
// leave eight at a time, 64 contiguous bytes per lane.
__global__ __launch_bounds__(64) void k_bx_fse(BxArgs a, const uint32_t *list, const uint32_t *n_list_p, const uint32_t LANES) {
    const uint32_t CELLS = LANES == 64 ? 384 : (LANES == 32 ? 768 : 1280), STRIDE = 2 * CELLS;
    __shared__ __attribute__((aligned(16))) uint8_t T[64 * 2 * 384];  // = the largest of the three shapes
    __shared__ __attribute__((aligned(16))) uint8_t BUF[64 * 64];
    __shared__ uint32_t s_ll[64], s_ml[64];  // base value | extra bits << 24, by symbol
    typedef __attribute__((address_space(3))) uint16_t lds16;
    const uint32_t lane = threadIdx.x;
    s_ll[lane] = lane < 36 ? c_ll_base[lane] | ((uint32_t)c_ll_bits[lane] << 24) : 0u;
    s_ml[lane] = lane < 53 ? c_ml_base[lane] | ((uint32_t)c_ml_bits[lane] << 24) : 0u;
    __builtin_amdgcn_wave_barrier();
    const uint32_t n_list = *n_list_p;
    lds8 *const my = (lds8 *)T + (lane < LANES ? lane : 0) * STRIDE;
    lds8 *const buf = (lds8 *)BUF + lane * 64;
    for (uint32_t g0 = blockIdx.x * LANES; g0 < n_list; g0 += gridDim.x * LANES) {
        const bool on0 = lane < LANES && g0 + lane < n_list;
        const uint32_t slot = on0 ? list[g0 + lane] : 0;
        BxPrep pr;
        pr.frame = 0; pr.k = 0; pr.logs = 0; pr.bs_off = 0; pr.bs_len = 0; pr.tab[0] = pr.tab[1] = pr.tab[2] = 0;
        FzItem it;
        it.nseq = 0; it.seq_off = 0; it.lit_len = 0;
        if (on0) { pr = a.prep[slot]; it = a.items[slot]; }
        const uint32_t nseq = on0 ? it.nseq : 0;
        const uint32_t row = on0 ? a.cand_row[pr.frame] : 0;
        const uint64_t foff = on0 ? a.blob_off[row] - a.blob_base : 0;  // the frame inside the blob region
        const uint8_t *const bb = a.blobs + foff + pr.bs_off;           // first byte of the bitstream
        const uint32_t log_l = pr.logs & 255, log_o = (pr.logs >> 8) & 255, log_m = (pr.logs >> 16) & 255;
        const uint32_t c_l = (1u << log_l) < 8 ? 8u : 1u << log_l, c_o = (1u << log_o) < 8 ? 8u : 1u << log_o, c_m = (1u << log_m) < 8 ? 8u : 1u << log_m;
        // this lane's tables -> its LDS region [LL | OF | ML], 16 bytes per step
        __builtin_amdgcn_wave_barrier();
        if (on0) {
            const uint16_t *const gl = a.fse_pool + pr.tab[0], *const go = a.fse_pool + pr.tab[1], *const gm = a.fse_pool + pr.tab[2];
            for (uint32_t i = 0; i < c_l; i += 8) { const uint4 v = *reinterpret_cast<const uint4 *>(gl + i); LDS_CP(my + 2 * i, &v, 16); }
            for (uint32_t i = 0; i < c_o; i += 8) { const uint4 v = *reinterpret_cast<const uint4 *>(go + i); LDS_CP(my + 2 * (c_l + i), &v, 16); }
            for (uint32_t i = 0; i < c_m; i += 8) { const uint4 v = *reinterpret_cast<const uint4 *>(gm + i); LDS_CP(my + 2 * (c_l + c_o + i), &v, 16); }
        }
        const lds8 *const tl = my, *const to = my + 2 * c_l, *const tm = my + 2 * (c_l + c_o);
        unsigned long long *const recs = a.seq_pool + it.seq_off;
        // the lane's buffer holds stream bytes [bufhi - 64, bufhi); pv = the bytes [pendhi - 64, pendhi) on their way
        int32_t bufhi = 0, pendhi = 0;
        uint4 pv0 = make_uint4(0, 0, 0, 0), pv1 = pv0, pv2 = pv0, pv3 = pv0;
        auto fetch = [&](int32_t hi_byte) {  // issue the loads of the 64 bytes that end at stream byte hi_byte
            pendhi = hi_byte;
            const uint8_t *g = bb + hi_byte - 64;
            __builtin_memcpy(&pv0, g, 16); __builtin_memcpy(&pv1, g + 16, 16); __builtin_memcpy(&pv2, g + 32, 16); __builtin_memcpy(&pv3, g + 48, 16);
        };
        auto commit = [&]() {  // the fetched bytes become the buffer
            LDS_CP(buf, &pv0, 16); LDS_CP(buf + 16, &pv1, 16); LDS_CP(buf + 32, &pv2, 16); LDS_CP(buf + 48, &pv3, 16);
            bufhi = pendhi;
        };
        // the 128 bits of the 16 stream bytes that end at byte `bend` (v = those bytes out of the buffer; bytes in front of the
        // stream read as zero)
        auto bits128 = [&](const uint4 v, const int32_t bend, uint64_t &lo, uint64_t &hi) {
            lo = (uint64_t)v.x | ((uint64_t)v.y << 32); hi = (uint64_t)v.z | ((uint64_t)v.w << 32);
            if (bend < 16) {  // (the last sequences of a block)
                const uint32_t z = 8u * (uint32_t)(16 - bend);  // bits, 8 .. 128
                if (z >= 64) { lo = 0; hi = z >= 128 ? 0 : (hi >> (z - 64)) << (z - 64); }
                else lo = (lo >> z) << z;
            }
        };
        int err = 0;
        int32_t left = 0;
        uint32_t sl = 0, so = 0, sm = 0;
        if (nseq) {
            const uint32_t last = bb[pr.bs_len - 1];  // != 0 (k_bx_prep)
            left = (int32_t)(pr.bs_len * 8) - (8 - hibit(last));
            const uint32_t need = log_l + log_o + log_m;
            if (left < (int32_t)need) err = E_CORRUPT;
            else if (foff + pr.bs_off < 64) err = E_UNSUP;  // (k_bx_prep sends such a block to k_bx_fse_wave: never taken)
            else {
                const int32_t bend = (left + 7) >> 3;
                fetch(bend); commit();
                uint4 v;
                LDS_LD(&v, buf + 48, 16);
                uint64_t lo, hi;
                bits128(v, bend, lo, hi);
                const uint32_t top = 128u - (uint32_t)(8 * bend - left);  // bit index just above the first unread bit
                sl = bx_ext(lo, hi, top - log_l, log_l);
                so = bx_ext(lo, hi, top - log_l - log_o, log_o);
                sm = bx_ext(lo, hi, top - need, log_m);
                left -= (int32_t)need;
            }
        }
        const bool first_block = pr.k == 0;
        uint32_t r0 = first_block ? 1u : FZ_SYM, r1 = first_block ? 4u : (FZ_SYM | (1u << 26)), r2 = first_block ? 8u : (FZ_SYM | (2u << 26));
        uint32_t sum_ll = 0, sum_ml = 0;
        uint32_t nmax = nseq;
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) nmax = max(nmax, (uint32_t)__shfl_xor((int)nmax, d));
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_wave_barrier();
        bool have_pending = false;
        const uint32_t size_l = 1u << log_l, size_o = 1u << log_o, size_m = 1u << log_m;
        // one sequence of this lane's block -> its record (0 for a lane that has none left)
        auto step = [&](const uint32_t i) -> unsigned long long {
            unsigned long long rec = 0;
            if (i < nseq && err == 0) {
                const int32_t bend = (left + 7) >> 3;
                if (bufhi - bend > 48) { fetch(bend); commit(); }  // the buffer ran out (long sequences): reload on the spot
                uint4 v;
                LDS_LD(&v, buf + (48 - (bufhi - bend)), 16);
                const uint32_t cl = *(const lds16 *)(tl + 2 * sl), co = *(const lds16 *)(to + 2 * so), cm = *(const lds16 *)(tm + 2 * sm);
                const uint32_t sy_l = cl & 63u, sy_o = co & 63u, sy_m = cm & 63u;
                const uint32_t vl = s_ll[sy_l], vm = s_ml[sy_m];
                const uint32_t ns_l = cl >> 6, ns_o = co >> 6, ns_m = cm >> 6;
                uint64_t lo, hi;
                bits128(v, bend, lo, hi);
                const uint32_t llb = vl >> 24, mlb = vm >> 24, ofb = sy_o, need_v = llb + ofb + mlb;
                const bool lastseq = i + 1 == nseq;
                const uint32_t nbl = log_l - (uint32_t)hibit(ns_l), nbo = log_o - (uint32_t)hibit(ns_o), nbm = log_m - (uint32_t)hibit(ns_m);
                const uint32_t need_s = lastseq ? 0u : nbl + nbo + nbm;
                if (ofb > 27) err = E_UNSUP;
                else if (left < (int32_t)(need_v + need_s)) err = E_CORRUPT;
                else {
                    const uint32_t top = 128u - (uint32_t)(8 * bend - left);  // >= 121 >= need_v + need_s
                    const uint64_t xv = bx_ext64(lo, hi, top - need_v, need_v);  // offset bits, match-length bits, literal-length bits
                    const uint32_t ov = (1u << ofb) + (uint32_t)(xv >> (mlb + llb));
                    const uint32_t ml = (vm & 0xFFFFFFu) + ((uint32_t)(xv >> llb) & ((1u << mlb) - 1u));
                    const uint32_t ll = (vl & 0xFFFFFFu) + ((uint32_t)xv & ((1u << llb) - 1u));
                    if (!lastseq) {
                        const uint32_t xs = bx_ext(lo, hi, top - need_v - need_s, need_s);  // LL, ML, OF state bits
                        sl = (ns_l << nbl) - size_l + (xs >> (nbm + nbo));
                        sm = (ns_m << nbm) - size_m + ((xs >> nbo) & ((1u << nbm) - 1u));
                        so = (ns_o << nbo) - size_o + (xs & ((1u << nbo) - 1u));
                    }
                    left -= (int32_t)(need_v + need_s);
                    // repeat offsets (RFC 8878 3.1.1.5); a value with FZ_SYM set = "incoming entry k, minus d"
                    uint32_t o;
                    if (ov > 3) { o = ov - 3; r2 = r1; r1 = r0; r0 = o; }
                    else {
                        const uint32_t idx = ov - 1 + (ll == 0 ? 1u : 0u);
                        if (idx == 0) o = r0;
                        else {
                            if (idx < 3) o = idx == 1 ? r1 : r2;
                            else if (r0 & FZ_SYM) {
                                o = r0 + 1;
                                if ((o & 0x3FFFFFFu) == 0x3FFFFFFu) err = E_UNSUP;
                            } else {
                                o = r0 - 1;
                                if (o == 0) err = E_CORRUPT;
                            }
                            if (idx > 1) r2 = r1;
                            r1 = r0; r0 = o;
                        }
                    }
                    sum_ll += ll; sum_ml += ml;
                    if (sum_ll + sum_ml > BLOCK_MAX) err = E_UNSUP;
                    rec = (unsigned long long)ll | ((unsigned long long)ml << 17) | ((unsigned long long)o << 35);
                }
            }
            return rec;
        };
        auto refill = [&](const uint32_t i) {  // every four sequences, all lanes: last time's bytes in, the next ones on their way
            if (have_pending) commit();
            if (i < nseq && err == 0) fetch((left + 7) >> 3);
            have_pending = true;
        };
        for (uint32_t i0 = 0; i0 < nmax; i0 += 8) {
            refill(i0);
            const unsigned long long q0 = step(i0), q1 = step(i0 + 1), q2 = step(i0 + 2), q3 = step(i0 + 3);
            refill(i0 + 4);
            const unsigned long long q4 = step(i0 + 4), q5 = step(i0 + 5), q6 = step(i0 + 6), q7 = step(i0 + 7);
            if (i0 < nseq) {  // the block's record space is a multiple of eight
                uint4 *const d = reinterpret_cast<uint4 *>(recs + i0);
                d[0] = make_uint4((uint32_t)q0, (uint32_t)(q0 >> 32), (uint32_t)q1, (uint32_t)(q1 >> 32));
                d[1] = make_uint4((uint32_t)q2, (uint32_t)(q2 >> 32), (uint32_t)q3, (uint32_t)(q3 >> 32));
                d[2] = make_uint4((uint32_t)q4, (uint32_t)(q4 >> 32), (uint32_t)q5, (uint32_t)(q5 >> 32));
                d[3] = make_uint4((uint32_t)q6, (uint32_t)(q6 >> 32), (uint32_t)q7, (uint32_t)(q7 >> 32));
            }
        }
        if (on0) {
            if (!err && nseq && left != 0) err = E_CORRUPT;
            if (!err && sum_ll > it.lit_len) err = E_CORRUPT;
            if (!err && it.lit_len + sum_ml > BLOCK_MAX) err = E_CORRUPT;
            a.items[slot].out = it.lit_len + sum_ml;
            a.items[slot].rep[0] = r0; a.items[slot].rep[1] = r1; a.items[slot].rep[2] = r2;
            if (err) { a.items[slot].err = err; atomicAdd(&a.pool_used[3], 1ull); atomicAdd(&a.pool_used[4 + 3], 1ull); }
        }
    }
}

__device__ __forceinline__ FzArgs bx_as_fz(const BxArgs &a) {
    FzArgs z{};
    z.cand_row = a.cand_row; z.cand_fzbase = a.cand_base; z.cand_fzcap = nullptr; z.n_cand = 0;
    z.it_cand = nullptr; z.total_items = 0; z.cand_nb = a.cand_nb; z.items = a.items;
    z.blobs = a.blobs; z.blob_base = a.blob_base;
    z.blob_off = a.blob_off; z.blob_size = a.blob_size; z.usize = a.usize; z.out_off = a.out_off; z.out_cap = a.out_cap;
    z.out = a.out; z.row_flag = a.row_flag; z.status = a.status; z.preset = a.preset;
    z.lit_pool = a.lit_pool; z.lit_cap = a.lit_cap; z.seq_pool = a.seq_pool; z.seq_cap = a.seq_cap;
    z.pool_used = a.pool_used; z.cursor = nullptr; z.dbg = nullptr;
    return z;
}

// wave = frame, frames dealt out by an atomic cursor (the slots are an upper bound, most runs use few of them)
// WH / WC: history and chunk bytes of the wave's LDS output window.  Tables of small frames run the small window: the
// executor is a lone wave's instruction stream per frame (~250 cycles per sequence, most of them waiting), so frames
// resident per CU are its throughput, and the window is what bounds them.
template <bool PROF, uint32_t WH, uint32_t WC, int WAVES>
__global__ __launch_bounds__(64, WAVES) void k_bx_exec(BxArgs a) {
    __shared__ __attribute__((aligned(16))) uint8_t W[WH + WC + 64 + WIN_SCRATCH];
    const uint32_t lane = threadIdx.x;
    const uint32_t n_slots = a.ctr[0];
    FzArgs z = bx_as_fz(a);
    z.dbg = PROF ? a.dbg + 64 : nullptr;  // diagnostic (ZNIPPY_DDBG): the execute stage's counters sit behind the table stage's
    uint32_t n_done = 0;
    const uint32_t chunk = n_slots >= 32768 ? 4u : 1u;  // big tables: four slots per atomic (one cursor word takes ~88 additions per microsecond)
    for (;;) {
        uint32_t c0 = 0;
        if (lane == 0) c0 = atomicAdd(&a.ctr[4], chunk);
        c0 = rdlane_u(c0, 0);
        if (c0 >= n_slots) break;
        for (uint32_t c = c0; c < c0 + chunk && c < n_slots; c++) {
            if (a.rx_base && a.rx_base[c] != RX_NONE) continue;  // the resolve path's frame
            if (fz_exec_frame<PROF, WH, WC, (WC / 4 < WIN_SEQ_MAX ? WC / 4 : WIN_SEQ_MAX)>(z, c, W, lane)) n_done++;
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_wave_barrier();
        }
    }
    if (lane == 0 && n_done) atomicAdd(&a.pool_used[2], (unsigned long long)n_done);  // statistics: frames decoded by this path
}


// =============================================================================================
// The resolve path: big frames without a serial executor.
//
// One frame is one LZ77 dependency chain for a wave that executes its sequences in order (fz_exec_frame: ~3 ns per byte,
// 20 ms for 8 MiB whatever else the chip does).  Seen per BYTE the dependencies are shallow: a byte is a literal, or a
// copy of an earlier byte — which is a literal or a copy ...  So: every output byte gets a 32-bit word, RX_DONE | value
// for a literal, the index of the word it copies from for a match byte (k_rx_expand: wave = block, positions by prefix
// sums, 64 bytes per step); then rounds of pointer jumping over all words at once (k_rx_jump: word = *word, until the
// word is a value; RX_JUMPS hops per launch, so RX_ROUNDS launches cover any chain a 2 GiB frame can hold — a launch
// returns at once when the round before it left nothing); then the values are stored as the rows' bytes (k_rx_store).
// Updates are in place and unordered: whatever a thread reads through a word is a position further up the same chain
// or the final value, and 32-bit stores are whole.  Malformed input (mutants) only ever produces words that point
// backwards inside their own frame: offsets are checked against the position when the words are written.
// =============================================================================================
// OR of a 64-bit value over the wave's 64 lanes (wave-uniform result), on the DPP path like wave_incl_scan
__device__ __forceinline__ uint32_t wave_or32(uint32_t v) {
    v |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, true);   // row_shr:1
    v |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xf, 0xf, true);   // row_shr:2
    v |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xf, 0xf, true);   // row_shr:4
    v |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xf, 0xf, true);   // row_shr:8
    v |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xa, 0xf, false);  // row_bcast:15 -> rows 1, 3
    v |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xc, 0xf, false);  // row_bcast:31 -> rows 2, 3
    return rdlane_u(v, 63);
}
__device__ __forceinline__ uint64_t wave_or64(uint64_t v) { return ((uint64_t)wave_or32((uint32_t)(v >> 32)) << 32) | wave_or32((uint32_t)v); }

__global__ __launch_bounds__(64) void k_rx_plan(BxArgs a) {
    const uint32_t lane = threadIdx.x;
    const uint32_t n_slots = a.ctr[0];
    for (uint32_t c = blockIdx.x; c < n_slots; c += gridDim.x) {
        if (lane == 0) { a.rx_base[c] = RX_NONE; a.rx_fail[c] = 0; }
        const uint32_t nb = uni(a.cand_nb[c]), row = uni(a.cand_row[c]);  // (every wave-uniform value made visibly so: see k_rx_expand)
        if (!nb || row == 0xFFFFFFFFu) continue;
        const uint64_t fcs = uni64(a.usize[row]);
        if (fcs < a.rx_min || fcs >= (1ull << 30)) continue;
        const uint32_t base = uni(a.cand_base[c]);
        {   // every block came through the entropy stages and the sizes add up (as fz_exec_frame checks)
            unsigned long long tot = 0, seqs = 0;
            uint32_t bad = 0;
            for (uint32_t i = lane; i < nb; i += 64) {
                bad |= a.items[base + i].err != 0;
                tot += a.items[base + i].out;
                seqs += a.items[base + i].nseq;
            }
            for (int d = 32; d >= 1; d >>= 1) { tot += __shfl_xor(tot, d); seqs += __shfl_xor(seqs, d); bad |= __shfl_xor(bad, d); }
            tot = uni64(tot); seqs = uni64(seqs); bad = uni(bad);
            if (bad || tot != fcs) continue;
            if (fcs >= (1u << 20) && seqs * 2048 < fcs) continue;  // a few very long copies: the serial decoder's wide variant (fz_exec_frame)
        }
        const uint32_t need = ((uint32_t)fcs + 1023u) & ~1023u;
        unsigned long long at = 0;
        if (lane == 0) at = atomicAdd(&a.pool_used[10], (unsigned long long)need);
        at = rdlane64_u(at, 0);
        if (at + need > a.rx_cap) continue;  // pool full (every later request fails too: the words in use stay one prefix)
        if (lane == 0) atomicMax(&a.pool_used[11], at + need);
        for (uint32_t i = lane; i < need / 1024; i += 64) { a.rx_chunk[at / 1024 + i] = c; a.rx_cdone[at / 1024 + i] = 0; }
        for (uint32_t i = (uint32_t)fcs + lane; i < need; i += 64) a.rx_ptr[at + i] = RX_DONE;  // padding behind the frame
        // per block: where its output starts, and the repeat offsets it starts with (the blocks' own records were decoded
        // against a symbolic history; composed here block by block, as the serial executor does between blocks)
        uint32_t r0 = 1, r1 = 4, r2 = 8, ob = 0;
        for (uint32_t g = 0; g < nb; g += 64) {
            const uint32_t i = g + lane;
            const bool on = i < nb;
            uint32_t o = 0, x0 = FZ_SYM, x1 = FZ_SYM | (1u << 26), x2 = FZ_SYM | (2u << 26);
            if (on) { const FzItem it = a.items[base + i]; o = it.out; x0 = it.rep[0]; x1 = it.rep[1]; x2 = it.rep[2]; }
            const uint32_t inc = wave_incl_scan(o);
            uint32_t h0 = 0, h1 = 0, h2 = 0;
            const uint32_t cnt = nb - g < 64 ? nb - g : 64;
            for (uint32_t j = 0; j < cnt; j++) {
                if (lane == j) { h0 = r0; h1 = r1; h2 = r2; }
                const uint32_t y[3] = {rdlane_u(x0, j), rdlane_u(x1, j), rdlane_u(x2, j)};
                uint32_t nr[3];
#pragma unroll
                for (int q = 0; q < 3; q++) {
                    const uint32_t x = y[q];
                    if (x & FZ_SYM) {
                        const uint32_t kk = (x >> 26) & 3, dd = x & 0x3FFFFFFu;
                        const uint32_t in = kk == 0 ? r0 : (kk == 1 ? r1 : r2);
                        nr[q] = in > dd ? in - dd : 0u;
                    } else nr[q] = x;
                }
                r0 = nr[0]; r1 = nr[1]; r2 = nr[2];
            }
            if (on) {
                uint32_t *const bk = a.rx_blk + 4 * (size_t)(base + i);
                bk[0] = ob + inc - o; bk[1] = h0; bk[2] = h1; bk[3] = h2;
            }
            ob += rdlane_u(inc, 63);
        }
        uint32_t lp = 0;
        if (lane == 0) lp = atomicAdd(&a.ctr[9], nb);
        lp = rdlane_u(lp, 0);
        for (uint32_t i = lane; i < nb; i += 64) a.rx_list[lp + i] = base + i;
        if (lane == 0) { a.rx_base[c] = (uint32_t)at; atomicAdd(&a.ctr[11], 1u); }
    }
}

// (Control flow kept visibly uniform for the compiler — work items by a static stride, every wave-uniform value through
// readfirstlane, no break out of the group loop.  The first form of this kernel took its items from an atomic cursor and
// left the group loop by `break`; the structurizer turned that into exec-masked loops whose exit mask on the failure path
// was the `lane == 0` mask of the statement behind it, and a wave that hit a damaged block went round the same block for
// ever: tests/test_gpu_fuzz.py::test_every_bit_of_block_heads_in_big_frames hung, any added store made it pass.)
__global__ __launch_bounds__(64) void k_rx_expand(BxArgs a) {
    const uint32_t lane = threadIdx.x;
    const uint32_t n = uni(a.ctr[9] < a.item_cap ? a.ctr[9] : a.item_cap);
    for (uint32_t wi = blockIdx.x; wi < n; wi += gridDim.x) {
        const uint32_t slot = uni(a.rx_list[wi]);
        const FzItem it = a.items[slot];
        const uint32_t c = uni(a.prep[slot].frame), row = uni(a.cand_row[c]);
        const uint32_t fb = uni(a.rx_base[c]);
        const uint32_t *const bk = a.rx_blk + 4 * (size_t)slot;
        const uint32_t h0 = uni(bk[1]), h1 = uni(bk[2]), h2 = uni(bk[3]);
        const uint8_t *const src = a.blobs + (uni64(a.blob_off[row]) - a.blob_base);
        uint32_t *const P = a.rx_ptr + fb;  // P[position inside the frame]
        const uint32_t nseq = uni(it.nseq), lit_len = uni(it.lit_len), kind = uni(it.lit_kind);
        const uint64_t lit_off = uni64(it.lit_off);
        const bool rle = kind == 1;
        const uint32_t rle_word = RX_DONE | (uint32_t)(uint8_t)lit_off;
        const uint8_t *const lit_ptr = kind == 0 ? src + lit_off : a.lit_pool + lit_off;
        const unsigned long long *const recs = a.seq_pool + uni64(it.seq_off);
        uint32_t pos = uni(bk[0]), lpos = 0;
        const uint32_t end = pos + uni(it.out);
        uint32_t fail = 0;  // wave-uniform
        unsigned long long rec_next = lane < nseq ? recs[lane] : 0ull;
        for (uint32_t g0 = 0; g0 < nseq && !fail; g0 += 64) {
            const uint32_t cnt = nseq - g0 < 64 ? nseq - g0 : 64;
            const bool on = lane < cnt;
            const unsigned long long rec = on ? rec_next : 0ull;
            rec_next = g0 + 64 + lane < nseq ? recs[g0 + 64 + lane] : 0ull;
            const uint32_t ll0 = (uint32_t)rec & 0x1FFFFu, ml0 = (uint32_t)(rec >> 17) & 0x3FFFFu;
            const uint32_t ov = on ? (uint32_t)(rec >> 35) : 4u;
            uint32_t offset = ov;
            if (ov & FZ_SYM) {
                const uint32_t kk = (ov >> 26) & 3, dd = ov & 0x3FFFFFFu;
                const uint32_t in = kk == 0 ? h0 : (kk == 1 ? h1 : h2);
                offset = in > dd ? in - dd : 0u;
            }
            const uint32_t linc = wave_incl_scan(ll0), pinc = wave_incl_scan(ll0 + ml0);
            const uint32_t total = rdlane_u(pinc, 63), lits = rdlane_u(linc, 63);
            // a match starts at pos + pinc - ml0 and may reach back to the frame's first byte, no further; literals and
            // output stay inside what the block declared
            const bool bad = on && (offset == 0 || offset > pos + pinc - ml0);
            fail = uni((__ballot(bad) != 0ull || lpos + lits > lit_len || total > end - pos) ? 1u : 0u);
            const uint32_t todo = fail ? 0u : total;
            // Which sequence holds byte b?  Sequences are at least 3 bytes long and start in increasing order: every
            // sequence lane whose first byte lies in the 64-byte window sets that byte's bit, the bits are OR-ed across
            // the wave (DPP: no LDS round trip), and byte lane b counts the bits at or below its own — the number of
            // sequences that have started — on top of those of the windows before.  (First form: a binary search over
            // the inclusive sums through __shfl, six dependent LDS crossbar trips per 64 bytes: 3.5 ms per GB.)
            const uint32_t start = pinc - (ll0 + ml0);
            uint32_t started = 0;  // sequences whose first byte lies before the window (wave-uniform)
            for (uint32_t b0 = 0; b0 < todo; b0 += 64) {
                const uint32_t b = b0 + lane;
                const bool v = b < total;
                const bool mine = on && (ll0 + ml0) != 0 && start >= b0 && start < b0 + 64;
                const uint64_t M = wave_or64(mine ? 1ull << (start - b0) : 0ull);
                const uint64_t le_mask = lane == 63 ? ~0ull : ((2ull << lane) - 1ull);
                uint32_t lo = started + (uint32_t)__popcll(M & le_mask);
                lo = lo ? lo - 1 : 0;  // (a group's first sequence starts at byte 0: lo >= 1 for every byte)
                started += (uint32_t)__popcll(M);
                const uint32_t pe = __shfl(pinc, lo), le = __shfl(linc, lo), li = __shfl(ll0, lo), mi = __shfl(ml0, lo), of = __shfl(offset, lo);
                const uint32_t bb = v ? b : pe - 1;
                const uint32_t w = bb - (pe - li - mi);  // position inside the sequence: literals first
                if (v) {
                    uint32_t word;
                    if (w < li) word = rle ? rle_word : (RX_DONE | lit_ptr[lpos + (le - li) + w]);
                    else word = fb + pos + b - of;
                    P[pos + b] = word;
                }
            }
            pos += todo; lpos += fail ? 0u : lits;
        }
        const uint32_t rest = lit_len - lpos;
        if (!fail && rest != end - pos) fail = 1;
        // literals behind the last sequence — or, when the block did not check out, values for every word it has not
        // written (the frame goes to the serial decoder; no word may stay stale)
        const uint32_t fill_n = end - pos;
        for (uint32_t t = lane; t < fill_n; t += 64) P[pos + t] = fail ? RX_DONE : (rle ? rle_word : (RX_DONE | lit_ptr[lpos + t]));
        if (fail && lane == 0) a.rx_fail[c] = 1;
    }
}

__global__ __launch_bounds__(256) void k_rx_jump(BxArgs a, uint32_t round) {
    if (round && a.rx_pending[round - 1] == 0) return;
    const unsigned long long used = a.pool_used[11];
    uint32_t *const P = a.rx_ptr;
    uint32_t pend = 0;
    // workgroup = one chunk of 1,024 words per step; a chunk whose words are all values is marked and skipped from then on
    // (after the first round most are: the later rounds then read a byte per chunk instead of 4 KiB)
    for (unsigned long long ch = blockIdx.x; ch * 1024 < used; ch += gridDim.x) {
        if (uni((uint32_t)a.rx_cdone[ch])) continue;  // (read by every thread, made visibly uniform: a barrier follows)
        const unsigned long long e0 = ch * 1024 + threadIdx.x * 4;
        uint4 w = *reinterpret_cast<const uint4 *>(P + e0);
        uint32_t v[4] = {w.x, w.y, w.z, w.w};
        bool changed = false;
        uint32_t mine = 0;
        // the four bytes of a thread mostly belong to one match: their words point at four consecutive words, which are
        // one 16-byte load (any 4-byte boundary) — and the words found there are often consecutive again
        uint32_t hops = 0;
        while (hops < RX_JUMPS && !((v[0] | v[1] | v[2] | v[3]) & RX_DONE) && v[1] == v[0] + 1 && v[2] == v[0] + 2 && v[3] == v[0] + 3 &&
               (unsigned long long)v[0] + 4 <= used) {
            const uint4 q = *reinterpret_cast<const uint4 *>(P + v[0]);
            v[0] = q.x; v[1] = q.y; v[2] = q.z; v[3] = q.w;
            changed = true;
            hops++;
        }
#pragma unroll
        for (int i = 0; i < 4; i++) {
            if (v[i] & RX_DONE) continue;
            uint32_t x = v[i];
            for (uint32_t j = hops; j < RX_JUMPS; j++) {
                if (x >= used) { x = RX_DONE; break; }  // (cannot happen for words this run wrote: kept as the bound of the read)
                x = P[x];
                if (x & RX_DONE) break;
            }
            v[i] = x; changed = true;
            mine |= !(x & RX_DONE);
        }
        if (changed) *reinterpret_cast<uint4 *>(P + e0) = make_uint4(v[0], v[1], v[2], v[3]);
        const int any = __syncthreads_or((int)mine);
        if (!any && threadIdx.x == 0) a.rx_cdone[ch] = 1;
        pend |= (uint32_t)any;
    }
    if (pend && threadIdx.x == 0) atomicAdd(&a.rx_pending[round], 1u);
}

// The values leave as the rows' bytes.  A row starts at any address: thread t of a chunk writes the ALIGNED dword that holds
// bytes [4t - m, 4t - m + 4) of the chunk (m = the row address's low two bits), from four consecutive words (a 16-byte
// load on a 4-byte boundary); the bytes at a row's two ends that do not fill a dword are stored one by one.
__global__ __launch_bounds__(256) void k_rx_store(BxArgs a) {
    const unsigned long long used = a.pool_used[11];
    for (unsigned long long ch = blockIdx.x; ch * 1024 < used; ch += gridDim.x) {
        const uint32_t c = uni(a.rx_chunk[ch]);
        const uint32_t fb = uni(a.rx_base[c]);
        if (fb == RX_NONE || uni(a.rx_fail[c])) continue;
        const uint32_t row = uni(a.cand_row[c]);
        const uint32_t fcs = uni((uint32_t)a.usize[row]);
        uint8_t *const D = a.out + uni64(a.out_off[row]);
        const uint32_t m = (uint32_t)((uintptr_t)D & 3);
        const uint32_t r0 = (uint32_t)(ch * 1024 - fb);
        const int64_t rel = (int64_t)r0 + 4 * (int64_t)threadIdx.x - m;  // first byte of this thread's dword, inside the row
        const uint32_t *const W = a.rx_ptr + fb;
        bool ok = true;
        if (rel >= 0 && rel + 4 <= (int64_t)fcs) {
            const uint4 w = *reinterpret_cast<const uint4 *>(W + rel);
            ok = ((w.x & w.y & w.z & w.w) & RX_DONE) != 0;
            *reinterpret_cast<uint32_t *>(D + rel) = (w.x & 255) | (w.y & 255) << 8 | (w.z & 255) << 16 | (w.w & 255) << 24;
        } else {
            for (int i = 0; i < 4; i++) {
                const int64_t p = rel + i;
                if (p >= 0 && p < (int64_t)fcs) { const uint32_t x = W[p]; ok = ok && (x & RX_DONE); D[p] = (uint8_t)x; }
            }
        }
        // the chunks cover [-m, need - m): a row that fills its last chunk to the end has m more bytes
        if (threadIdx.x == 255 && r0 + 1024 >= fcs)
            for (uint32_t p = r0 + 1024 - m; p < fcs; p++) { const uint32_t x = W[p]; ok = ok && (x & RX_DONE); D[p] = (uint8_t)x; }
        if (!ok) a.rx_fail[c] = 1;  // a chain longer than the rounds cover: cannot happen below 2^31 bytes
    }
}

__global__ __launch_bounds__(64) void k_bx_finish(BxArgs a) {
    const uint32_t lane = threadIdx.x;
    const uint32_t n_slots = a.ctr[0];
    for (uint32_t t0 = blockIdx.x * 64; t0 < n_slots; t0 += gridDim.x * 64) {
        const uint32_t t = t0 + lane;
        const uint32_t row = t < n_slots ? a.cand_row[t] : 0xFFFFFFFFu;
        bool left = false;
        if (row != 0xFFFFFFFFu) {
            if (a.rx_base && a.rx_base[t] != RX_NONE && !a.rx_fail[t]) {  // resolved and stored (k_rx_store found every word a value)
                a.row_flag[row] = 0;
                atomicAdd(&a.pool_used[2], 1ull);
            }
            if (a.row_flag[row]) { a.status[row] = 1; left = true; }
            else a.status[row] = 2;  // decoded: the second hash pass takes it
        }
        const uint64_t m = __ballot(left);
        const uint64_t below = lane ? (~0ull >> (64 - lane)) : 0ull;
        uint32_t b = 0;
        if (lane == 0 && m) b = atomicAdd(a.pending2_count, (uint32_t)__popcll(m));
        b = rdlane_u(b, 0);
        if (left) a.pending2[b + (uint32_t)__popcll(m & below)] = row;
    }
}

// stage 0..5 = scan, prep, huf, fse (lane = block), exec, finish; 6 = fse (wave = block, the long chains: beside stages 2 and 3 on
// another stream) — launched one by one so that each can be timed
void launch_bx_stage(const BxArgs &a, int cus, int stage, hipStream_t s) {
    const uint32_t slots = a.slot_cap, lane_grid = std::max(std::min<uint32_t>((slots + 63) / 64, (uint32_t)cus * 8), 1u);
    if (!slots) return;
    auto cap = [&](uint32_t want, uint32_t per_cu) { return dim3(std::max(std::min<uint32_t>(want, (uint32_t)cus * per_cu), 1u)); };
    switch (stage) {
    case 0: hipLaunchKernelGGL(k_bx_scan, dim3(lane_grid), dim3(64), 0, s, a); break;
    case 1: hipLaunchKernelGGL(k_bx_prep, cap((a.item_cap + 63) / 64, 3), dim3(64), 0, s, a); break;
    case 7:
        if (!a.big_seq) hipLaunchKernelGGL(k_bx_split, dim3(3), dim3(1024), 0, s, a);
        hipLaunchKernelGGL(k_bx_sort, dim3(5), dim3(1024), 0, s, a, a.sort_tmp);
        break;
    case 2: hipLaunchKernelGGL(k_bx_huf, cap((a.item_cap + BX_HUF_BLOCKS - 1) / BX_HUF_BLOCKS, 2), dim3(64), 0, s, a); break;
    case 3:
        hipLaunchKernelGGL(k_bx_fse, cap((a.item_cap + 63) / 64, 3), dim3(64), 0, s, a, a.seq_list, a.ctr + 3, 64u);
        hipLaunchKernelGGL(k_bx_fse, cap((a.item_cap + 31) / 32, 3), dim3(64), 0, s, a, a.seq_list + a.item_cap, a.ctr + 5, 32u);
        hipLaunchKernelGGL(k_bx_fse, cap((a.item_cap + 15) / 16, 3), dim3(64), 0, s, a, a.seq_list + 2 * (size_t)a.item_cap, a.ctr + 6, 16u);
        break;
    case 6: hipLaunchKernelGGL(k_bx_fse_wave, cap(a.item_cap, 16), dim3(64), 0, s, a, a.seq_list + 3 * (size_t)a.item_cap, a.ctr + 7); break;  // 10 KB of LDS per wave: 16 per CU
    case 4:
        if (a.dbg) hipLaunchKernelGGL((k_bx_exec<true, WIN_HIST, WIN_CAP, 1>), cap(slots, 12), dim3(64), 0, s, a);
        else if (a.small_frames) hipLaunchKernelGGL((k_bx_exec<false, BX_SMALL_HIST, BX_SMALL_CAP, BX_SMALL_WAVES>), cap(slots, 4 * BX_SMALL_WAVES), dim3(64), 0, s, a);
        else hipLaunchKernelGGL((k_bx_exec<false, WIN_HIST, WIN_CAP, 1>), cap(slots, 12), dim3(64), 0, s, a);
        break;
    case 8: hipLaunchKernelGGL(k_rx_plan, cap(slots, 8), dim3(64), 0, s, a); break;
    case 9: hipLaunchKernelGGL(k_rx_expand, cap(a.item_cap, 32), dim3(64), 0, s, a); break;  // (static stride over the list; lone chains: the more the better, 54 VGPRs, no LDS)
    case 30: hipLaunchKernelGGL(k_rx_store, cap((uint32_t)std::min<uint64_t>((a.rx_bound + 1023) / 1024, 1u << 30), 64), dim3(256), 0, s, a); break;
    case 5: hipLaunchKernelGGL(k_bx_finish, dim3(lane_grid), dim3(64), 0, s, a); break;
    default:
        if (stage >= 10 && stage < 10 + (int)RX_ROUNDS)
            hipLaunchKernelGGL(k_rx_jump, cap((uint32_t)std::min<uint64_t>((a.rx_bound + 1023) / 1024, 1u << 30), 32), dim3(256), 0, s, a, (uint32_t)(stage - 10));
        break;
    }
}

// host: the three predefined tables (RFC 8878 3.1.1.3.2.2.1) as FSE pool cells, by the same spread as the device's
void bx_predefined_tables(uint16_t cells[160]) {
    static const int8_t ll_def[36] = {4, 3, 2, 2, 2, 2, 2, 2, 2, 2, 2, 2, 2, 1, 1, 1, 2, 2, 2, 2, 2, 2, 2, 2, 2, 3, 2, 1, 1, 1, 1, 1, -1, -1, -1, -1};
    static const int8_t ml_def[53] = {1, 4, 3, 2, 2, 2, 2, 2, 2, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1,
                                      1, 1, 1, 1, 1, 1, 1, 1, 1, 1, -1, -1, -1, -1, -1, -1, -1};
    static const int8_t of_def[29] = {1, 1, 1, 1, 1, 1, 2, 2, 2, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, -1, -1, -1, -1, -1};
    auto build = [&](const int8_t *norm, int nsym, int log, int kind, uint16_t *t) {
        const int size = 1 << log;
        int high = size;
        uint8_t sym_of[64];
        uint16_t next[64];
        for (int s = 0; s < nsym; s++) if (norm[s] == -1) { sym_of[--high] = (uint8_t)s; next[s] = 1; }
        const int step = (size >> 1) + (size >> 3) + 3, mask = size - 1;
        int pos = 0;
        for (int s = 0; s < nsym; s++) {
            if (norm[s] <= 0) continue;
            next[s] = (uint16_t)norm[s];
            for (int i = 0; i < norm[s]; i++) { sym_of[pos] = (uint8_t)s; do { pos = (pos + step) & mask; } while (pos >= high); }
        }
        for (int u = 0; u < size; u++) {
            const uint32_t sym = sym_of[u], ns = next[sym]++;
            uint32_t hb = 31; while (!(ns >> hb)) hb--;  // (kept: documents that the bits to read are log - hb)
            (void)hb; (void)kind;
            t[u] = (uint16_t)(sym | (ns << 6));
        }
    };
    build(ll_def, 36, 6, K_LL, cells + BX_PREDEF_LL);
    build(of_def, 29, 5, K_OF, cells + BX_PREDEF_OF);
    build(ml_def, 53, 6, K_ML, cells + BX_PREDEF_ML);
}

}  // namespace zn
