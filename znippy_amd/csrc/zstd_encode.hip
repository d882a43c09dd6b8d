// Zstandard frame encoder for CDNA4 (gfx950) — the GPU side of CompressCtx::compress_into
// (znippy-common/src/codec.rs:L43-55) as called from the write worker loops
// (znippy-compress/src/stream_packer.rs:L217-246, slot_packer.rs:L551-580).
//
// Not a port of libzstd: the unit of work is one <=128 KiB zstd block of one Round, handled by
// one wavefront that pulls blocks from an atomic cursor.  Match finding is a greedy LZ over a
// per-wave LDS hash table probed 64 positions at a time (lane = position); a found match is
// extended 1 KiB per step by the whole wave.  Literals go out raw, sequences are FSE-coded
// with the RFC 8878 predefined distributions (no table description, no repeat offsets), so every
// block is self-contained and blocks of one frame are encoded in parallel.  A block that does
// not shrink is emitted as a raw block.  Output pieces land in a provisional area; a scan +
// gather pass packs them back-to-back in Round order (the writer's running out_cursor,
// stream_packer.rs:L258, made deterministic).
#include "common.h"
#include "encode.h"
#include "hash_dev.h"

namespace zn {


__device__ __forceinline__ uint32_t ld32(const uint8_t *p) {
    uint32_t v;
    __builtin_memcpy(&v, p, 4);
    return v;
}
__device__ __forceinline__ uint4 ld128(const uint8_t *p) {
    uint4 v;
    __builtin_memcpy(&v, p, 16);
    return v;
}
template <uint32_t HASH_LOG>
__device__ __forceinline__ uint32_t hash4(uint32_t v) { return (v * 2654435761u) >> (32 - HASH_LOG); }
__device__ __forceinline__ int hib(uint32_t v) { return 31 - __clz(v); }

__constant__ uint8_t c_ll_code[64] = {0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15,
                                      16, 16, 17, 17, 18, 18, 19, 19, 20, 20, 20, 20, 21, 21, 21, 21,
                                      22, 22, 22, 22, 22, 22, 22, 22, 23, 23, 23, 23, 23, 23, 23, 23,
                                      24, 24, 24, 24, 24, 24, 24, 24, 24, 24, 24, 24, 24, 24, 24, 24};
__constant__ uint8_t c_ml_code[128] = {
    0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16, 17, 18, 19, 20, 21, 22, 23, 24, 25, 26, 27, 28, 29, 30, 31,
    32, 32, 33, 33, 34, 34, 35, 35, 36, 36, 36, 36, 37, 37, 37, 37, 38, 38, 38, 38, 38, 38, 38, 38, 39, 39, 39, 39, 39, 39, 39, 39,
    40, 40, 40, 40, 40, 40, 40, 40, 40, 40, 40, 40, 40, 40, 40, 40, 41, 41, 41, 41, 41, 41, 41, 41, 41, 41, 41, 41, 41, 41, 41, 41,
    42, 42, 42, 42, 42, 42, 42, 42, 42, 42, 42, 42, 42, 42, 42, 42, 42, 42, 42, 42, 42, 42, 42, 42, 42, 42, 42, 42, 42, 42, 42, 42};
__constant__ uint32_t c_ll_base_e[36] = {0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16, 18,
                                         20, 22, 24, 28, 32, 40, 48, 64, 128, 256, 512, 1024, 2048,
                                         4096, 8192, 16384, 32768, 65536};
__constant__ uint8_t c_ll_bits_e[36] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 1, 1,
                                        1, 1, 2, 2, 3, 3, 4, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16};
__constant__ uint32_t c_ml_base_e[53] = {3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16, 17, 18, 19, 20,
                                         21, 22, 23, 24, 25, 26, 27, 28, 29, 30, 31, 32, 33, 34, 35, 37,
                                         39, 41, 43, 47, 51, 59, 67, 83, 99, 131, 259, 515, 1027, 2051,
                                         4099, 8195, 16387, 32771, 65539};
__constant__ uint8_t c_ml_bits_e[53] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0,
                                        0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 1, 1,
                                        1, 1, 2, 2, 3, 3, 4, 4, 5, 7, 8, 9, 10, 11,
                                        12, 13, 14, 15, 16};

// forward bit writer (lane 0): LSB-first accumulate, flushed 4 bytes at a time
struct BitW {
    uint8_t *p;
    uint64_t acc;
    uint32_t n;
    __device__ __forceinline__ void add(uint32_t v, uint32_t nb) {
        acc |= (uint64_t)(v & ((nb >= 32) ? 0xFFFFFFFFu : ((1u << nb) - 1))) << n;
        n += nb;
        if (n >= 32) {
            uint32_t w = (uint32_t)acc;
            __builtin_memcpy(p, &w, 4);
            p += 4;
            acc >>= 32;
            n -= 32;
        }
    }
    __device__ __forceinline__ uint8_t *close() {
        add(1, 1);
        while (n > 0) {
            *p++ = (uint8_t)acc;
            acc >>= 8;
            n = n > 8 ? n - 8 : 0;
        }
        return p;
    }
};

struct CState {
    uint32_t v;
    const uint16_t *st;
    const FseSymTT *tt;
    uint32_t log;
    __device__ __forceinline__ void init(const uint16_t *st_, const FseSymTT *tt_, uint32_t log_, uint32_t sym) {
        st = st_; tt = tt_; log = log_;
        const FseSymTT t = tt[sym];
        uint32_t nb = (t.delta_nb_bits + (1u << 15)) >> 16;
        uint32_t val = (nb << 16) - t.delta_nb_bits;
        v = st[(int32_t)(val >> nb) + t.delta_find_state];
    }
    __device__ __forceinline__ void encode(BitW &b, uint32_t sym) {
        const FseSymTT t = tt[sym];
        uint32_t nb = (v + t.delta_nb_bits) >> 16;
        b.add(v, nb);
        v = st[(int32_t)(v >> nb) + t.delta_find_state];
    }
    __device__ __forceinline__ void flush(BitW &b) { b.add(v, log); }
};

constexpr uint32_t SMALL_BLOCK = 16 * 1024;           // blocks of the small-table variant
constexpr uint32_t SMALL_MAX_SEQ = SMALL_BLOCK / 40;  // their sequence budget fits in LDS

template <uint32_t HASH_LOG>
struct EncShared {
    uint16_t table[1u << HASH_LOG];
    EncTables tabs;  // FSE encoding tables, copied once per persistent wave (lane 0 reads them per sequence)
    uint32_t seqs[HASH_LOG == 11 ? 3 * (SMALL_MAX_SEQ + 1) : 4];  // small variant: sequences stay on chip
    uint32_t sink[64];  // landing area of the prefetch touches (never read)
};

// n bytes src -> dst by one wave (ranges do not overlap)
__device__ __forceinline__ void wave_copy(uint8_t *dst, const uint8_t *src, uint32_t n, uint32_t lane) {
    if (n < 128) {
        for (uint32_t i = lane; i < n; i += 64) dst[i] = src[i];
        return;
    }
    const uint32_t head = (uint32_t)((16 - ((uintptr_t)dst & 15)) & 15);
    if (lane < head) dst[lane] = src[lane];
    const uint32_t body = (n - head) >> 4;
    for (uint32_t i = lane; i < body; i += 64) {
        uint4 v = ld128(src + head + (size_t)i * 16);
        *reinterpret_cast<uint4 *>(dst + head + (size_t)i * 16) = v;
    }
    const uint32_t done = head + body * 16;
    if (done + lane < n) dst[done + lane] = src[done + lane];
}

__device__ __forceinline__ uint32_t suni(uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); }
__device__ __forceinline__ uint32_t rdlane(uint32_t v, uint32_t l) { return (uint32_t)__builtin_amdgcn_readlane((int)v, (int)l); }

// Sequences section bitstream (predefined FSE tables), built by the whole wave.
//   * everything that does not depend on the FSE state is per-lane work (lane j = the j-th sequence in
//     encoding order of a batch of 64: codes, extra bits, table rows);
//   * the three state chains (the only serial dependency) run on the scalar unit: the state tables live in
//     VGPRs (lane k = entry k) and are indexed with v_readlane, the per-step outputs go back to lane j with
//     a compare+select — no memory access inside the chain;
//   * each lane then assembles its sequence's <= 75 bits, an exclusive scan places them, and the lanes OR
//     them into a small LDS window that is streamed out one dword per lane.
// seqs[3*i..] = {ll, ml-3, offset}; returns the end of the bitstream (after the closing 1 bit).
// ov_bias: what turns seqs[3*i+2] into the coded offset value (3 when the entry is a plain offset, 0 when the
// matcher already stored the offset value with its repeat codes).
template <class SH>
__device__ __noinline__ uint8_t *encode_sequences(SH &S, const uint32_t *seqs, uint32_t nseq, uint8_t *p, uint32_t lane, uint32_t ov_bias) {
    // bit window: 64 sequences x <= 75 bits + carry; lives in the hash table, which is dead once the block's matches are found
    uint32_t *const ebits = reinterpret_cast<uint32_t *>(S.table);
    const uint32_t st_ll = S.tabs.ll_state[lane & 63], st_ml = S.tabs.ml_state[lane & 63], st_of = S.tabs.of_state[lane & 31];
    for (uint32_t i = lane; i < 200; i += 64) ebits[i] = 0;
    uint32_t vl = 0, vm = 0, vo = 0;  // FSE states (scalar)
    uint32_t bitpos = 0;              // bits pending in the LDS window (scalar)
    for (uint32_t e0 = 0; e0 < nseq; e0 += 64) {
        const uint32_t cnt = nseq - e0 < 64 ? nseq - e0 : 64;
        const bool on = lane < cnt;
        const uint32_t i = nseq - 1 - (e0 + (on ? lane : 0));  // encoding order: last sequence first
        const uint32_t ll = seqs[3 * i], mlb = seqs[3 * i + 1], ov = seqs[3 * i + 2] + ov_bias;
        const uint32_t lc = ll < 64 ? c_ll_code[ll] : (uint32_t)hib(ll) + 19;
        const uint32_t mc = mlb < 128 ? c_ml_code[mlb] : (uint32_t)hib(mlb) + 36;
        const uint32_t oc = (uint32_t)hib(ov);
        const FseSymTT tl = S.tabs.ll_tt[lc], tm = S.tabs.ml_tt[mc], to = S.tabs.of_tt[oc];
        uint32_t f_l = 0, f_m = 0, f_o = 0;  // state bits of my sequence: value | nbits << 16 (written by the chain)
        for (uint32_t j = 0; j < cnt; j++) {
            const uint32_t dl = rdlane(tl.delta_nb_bits, j), fl = rdlane((uint32_t)tl.delta_find_state, j);
            const uint32_t dm = rdlane(tm.delta_nb_bits, j), fm = rdlane((uint32_t)tm.delta_find_state, j);
            const uint32_t d_o = rdlane(to.delta_nb_bits, j), fo = rdlane((uint32_t)to.delta_find_state, j);
            if (e0 + j == 0) {  // first symbol: initial states, no bits
                uint32_t nb = (dm + (1u << 15)) >> 16;
                vm = rdlane(st_ml, (uint32_t)((int32_t)(((nb << 16) - dm) >> nb) + (int32_t)fm));
                nb = (d_o + (1u << 15)) >> 16;
                vo = rdlane(st_of, (uint32_t)((int32_t)(((nb << 16) - d_o) >> nb) + (int32_t)fo));
                nb = (dl + (1u << 15)) >> 16;
                vl = rdlane(st_ll, (uint32_t)((int32_t)(((nb << 16) - dl) >> nb) + (int32_t)fl));
            } else {
                uint32_t nb = (vo + d_o) >> 16;
                if (lane == j) f_o = (vo & ((1u << nb) - 1)) | (nb << 16);
                vo = rdlane(st_of, (uint32_t)((int32_t)(vo >> nb) + (int32_t)fo));
                nb = (vm + dm) >> 16;
                if (lane == j) f_m = (vm & ((1u << nb) - 1)) | (nb << 16);
                vm = rdlane(st_ml, (uint32_t)((int32_t)(vm >> nb) + (int32_t)fm));
                nb = (vl + dl) >> 16;
                if (lane == j) f_l = (vl & ((1u << nb) - 1)) | (nb << 16);
                vl = rdlane(st_ll, (uint32_t)((int32_t)(vl >> nb) + (int32_t)fl));
            }
        }
        // my sequence's bits in stream order: OF, ML, LL state bits, then LL, ML, OF extra bits
        uint64_t lo = 0, hi = 0;
        uint32_t T = 0;
        auto put = [&](uint32_t v, uint32_t nb) {  // nb <= 17, v < 2^nb
            if (T < 64) {
                lo |= (uint64_t)v << T;
                if (T + nb > 64) hi |= (uint64_t)v >> (64 - T);
            } else hi |= (uint64_t)v << (T - 64);
            T += nb;
        };
        if (on) {
            put(f_o & 0xFFFF, f_o >> 16);
            put(f_m & 0xFFFF, f_m >> 16);
            put(f_l & 0xFFFF, f_l >> 16);
            put(ll - c_ll_base_e[lc], c_ll_bits_e[lc]);
            put(mlb + 3 - c_ml_base_e[mc], c_ml_bits_e[mc]);
            put(ov - (1u << oc), oc);
        }
        uint32_t inc = T;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const uint32_t y = __shfl_up(inc, d);
            if (lane >= (uint32_t)d) inc += y;
        }
        const uint32_t pos = bitpos + inc - T;
        if (T) {
            const uint32_t sft = pos & 31, w = pos >> 5;  // <= 75 + 31 bits after the shift
            const uint64_t x0 = lo << sft, x1 = (hi << sft) | (sft ? lo >> (64 - sft) : 0ull);
            const uint32_t w0 = (uint32_t)x0, w1 = (uint32_t)(x0 >> 32), w2 = (uint32_t)x1, w3 = (uint32_t)(x1 >> 32);
            if (w0) atomicOr(&ebits[w], w0);
            if (w1) atomicOr(&ebits[w + 1], w1);
            if (w2) atomicOr(&ebits[w + 2], w2);
            if (w3) atomicOr(&ebits[w + 3], w3);
        }
        bitpos = suni(bitpos + __shfl(inc, 63));
        __builtin_amdgcn_wave_barrier();
        // stream out the complete dwords, keep the partial one in front
        const uint32_t ndw = bitpos >> 5;
        uint32_t keep = 0;
        for (uint32_t k = lane; k < ndw; k += 64) {
            const uint32_t w = ebits[k];
            __builtin_memcpy(p + 4 * k, &w, 4);
        }
        keep = ebits[ndw];
        __builtin_amdgcn_wave_barrier();
        for (uint32_t k = lane; k <= ndw; k += 64) ebits[k] = 0;
        __builtin_amdgcn_wave_barrier();
        if (lane == 0) ebits[0] = keep;
        __builtin_amdgcn_wave_barrier();
        p += 4 * ndw;
        bitpos &= 31;
    }
    // final states (ML, OF, LL), closing bit
    if (lane == 0) {
        uint64_t acc = ebits[0];
        uint32_t nb = bitpos;
        acc |= (uint64_t)(vm & 63) << nb; nb += 6;
        acc |= (uint64_t)(vo & 31) << nb; nb += 5;
        acc |= (uint64_t)(vl & 63) << nb; nb += 6;
        acc |= 1ull << nb; nb += 1;
        for (uint32_t k = 0; k * 8 < nb; k++) p[k] = (uint8_t)(acc >> (8 * k));
    }
    return p + ((bitpos + 18 + 7) >> 3);
}

// One FSE table from a histogram, by the wave (lane = symbol, alphabets of at most 64 symbols).  c = the symbol's count
// (0 = absent; at least two symbols present), total = the sum of the counts, tl = Accuracy_Log (5..8).  Normalises the
// counts to 2^tl (every present symbol >= 1, the remainder to the most frequent one), writes the table description
// (RFC 8878 4.1.1) to q and returns its size, and leaves the encoder's tables: symbol rows in tt[0..64), the 2^tl states
// in stt.  cum, cur (64 words each) and hbits (32 words) are LDS scratch.
__device__ __noinline__ uint32_t fse_build(uint32_t c, uint32_t total, uint32_t tl, FseSymTT *tt, uint16_t *stt, uint32_t *cum, uint32_t *cur,
                                           uint32_t *hbits, uint8_t *q, uint32_t lane) {
    auto wsum = [&](uint32_t v) -> uint32_t {
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d);
        return v;
    };
    auto wmax = [&](uint32_t v) -> uint32_t {
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) { const uint32_t o = __shfl_xor(v, d); v = o > v ? o : v; }
        return v;
    };
    const uint32_t ts = 1u << tl, step = (ts >> 1) + (ts >> 3) + 3;
    uint32_t inv = step;  // inverse of the (odd) spread step modulo 2^tl: Newton, 3 -> 6 -> 12 correct bits
    inv *= 2u - step * inv; inv *= 2u - step * inv;
    const uint64_t below = lane ? (~0ull >> (64 - lane)) : 0ull;
    const uint64_t present = __ballot(c != 0);
    const uint32_t last = 63u - (uint32_t)__clzll(present);
    const uint32_t nseq = total;
    uint32_t nbytes_out = 0;
    {
        uint32_t norm = c ? (c * ts + (nseq >> 1)) / nseq : 0;
        if (c && !norm) norm = 1;
        const uint32_t sum = wsum(norm);
        if (sum <= ts) {
            const uint32_t big = wmax((c << 6) | lane) & 63;
            if (lane == big) norm += ts - sum;
        } else {
            uint32_t excess = sum - ts;
            while (excess) {  // take it from the largest counts, a quarter of one at a time
                const uint32_t kb = wmax(norm > 1 ? (norm << 6) | lane : 0u);
                const uint32_t nb = kb >> 6;
                uint32_t take = nb >> 2 ? nb >> 2 : 1u;
                take = take < excess ? take : excess;
                take = take < nb - 1 ? take : nb - 1;
                if (lane == (kb & 63)) norm -= take;
                excess -= take;
            }
        }
        uint32_t incl = norm;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const uint32_t y = __shfl_up(incl, d);
            if (lane >= (uint32_t)d) incl += y;
        }
        const uint32_t excl = incl - norm;
        // symbol rows
        if (norm) {
            FseSymTT r;
            if (norm == 1) { r.delta_nb_bits = (tl << 16) - ts; r.delta_find_state = (int32_t)excl - 1; }
            else {
                const uint32_t mbo = tl - (uint32_t)hib(norm - 1);
                r.delta_nb_bits = (mbo << 16) - (norm << mbo);
                r.delta_find_state = (int32_t)excl - (int32_t)norm;
            }
            tt[lane] = r;
        }
        cum[lane] = incl;
        cur[lane] = excl;
        for (uint32_t i = lane; i < 32; i += 64) hbits[i] = 0;
        __builtin_amdgcn_wave_barrier();
        // state table
        for (uint32_t u0 = 0; u0 < ts; u0 += 64) {
            const uint32_t u = u0 + lane, kf = (u * inv) & (ts - 1);
            uint32_t sidx = 0;
#pragma unroll
            for (uint32_t st2 = 32; st2; st2 >>= 1)
                if (cum[sidx + st2 - 1] <= kf) sidx += st2;
            uint64_t todo = ~0ull;
            uint32_t slot = 0;
            while (todo) {
                const uint32_t l = (uint32_t)__ffsll((long long)todo) - 1, sl = rdlane(sidx, l);
                const uint64_t same = __ballot(sidx == sl);
                const uint32_t b0 = cur[sl];
                if (sidx == sl) slot = b0 + (uint32_t)__popcll(same & below);
                if (lane == l) cur[sl] = b0 + (uint32_t)__popcll(same);
                todo &= ~same;
            }
            stt[slot] = (uint16_t)(ts + u);
        }
        // table description
        {
            const uint32_t prevn = __shfl_up(norm, 1);
            const uint64_t nz = __ballot(norm != 0);
            uint64_t val = 0;
            uint32_t nb = 0;
            const bool zero = norm == 0;
            if (lane <= last && !(zero && lane && prevn == 0)) {
                const uint32_t remaining = ts + 1 - excl;
                const uint32_t thr = 1u << hib(remaining), nbb = (uint32_t)hib(thr) + 1, mx = 2 * thr - 1 - remaining;
                uint32_t cnt = norm + 1;
                const bool small = cnt < mx;
                if (cnt >= thr) cnt += mx;
                val = cnt;
                nb = nbb - (small ? 1u : 0u);
                if (zero) {  // repeat flags: how many more absent symbols follow, 3 per flag, the last flag < 3
                    const uint32_t next = lane + (uint32_t)__ffsll((long long)(nz >> (lane + 1)));  // lane < last: one exists
                    const uint32_t more = next - lane - 1, q3 = more / 3, r3 = more % 3;
                    const uint64_t flags = ((1ull << (2 * q3)) - 1) | ((uint64_t)r3 << (2 * q3));
                    val |= flags << nb;
                    nb += 2 * q3 + 2;
                }
            }
            uint32_t inc = nb;
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) {
                const uint32_t y = __shfl_up(inc, d);
                if (lane >= (uint32_t)d) inc += y;
            }
            const uint32_t bp = 4 + inc - nb;  // the first 4 bits are Accuracy_Log - 5
            if (lane == 0) atomicOr(&hbits[0], tl - 5);
            if (nb) {
                const uint32_t sft = bp & 31, w = bp >> 5;  // nb <= 9 + 36
                const uint64_t x0 = val << sft;
                const uint32_t x1 = sft ? (uint32_t)(val >> (64 - sft)) : 0u;
                if ((uint32_t)x0) atomicOr(&hbits[w], (uint32_t)x0);
                if ((uint32_t)(x0 >> 32)) atomicOr(&hbits[w + 1], (uint32_t)(x0 >> 32));
                if (x1) atomicOr(&hbits[w + 2], x1);
            }
            const uint32_t tbits = 4 + rdlane(inc, 63), nbytes = (tbits + 7) >> 3;
            __builtin_amdgcn_wave_barrier();
            for (uint32_t i = lane; i < nbytes; i += 64) q[i] = (uint8_t)(hbits[i >> 2] >> (8 * (i & 3)));
            nbytes_out = nbytes;
        }
    }
    __builtin_amdgcn_wave_barrier();
    return nbytes_out;
}

// Sequences section with per-block FSE tables (RFC 8878 3.1.1.3.2.1: Compression_Modes FSE_Compressed / RLE), the
// higher effort tier's counterpart of encode_sequences.  tools/enc_model.c puts the predefined distributions at 12 % of
// the whole output on real text: they price this matcher's offsets (codes 8..16) at 5 bits each.  The wave
//   * counts the three code alphabets (LDS atomics) and normalises each to 2^tl (tl 6..8 by sequence count; every
//     present symbol >= 1, the remainder to the most frequent one) — lane = symbol;
//   * writes the table description with one field per lane: what a symbol's field looks like depends only on the
//     counts before it (an exclusive scan), and a run of absent symbols is its first member's field plus repeat flags;
//   * builds the encoding tables: symbol rows from the scan; the state table by walking the spread table 64 cells at a
//     time (cell -> flat occurrence index through the inverse of the spread step -> symbol by binary search in the
//     cumulative counts -> rank inside the symbol by ballots);
//   * runs the same batch loop as encode_sequences with the state tables in LDS.
// The hash table is dead by now and holds all of it.  p = the Compression_Modes byte; returns the section's end.
template <class SH>
__device__ __noinline__ uint8_t *encode_sequences_custom(SH &S, const uint32_t *seqs, uint32_t nseq, uint8_t *p, uint32_t lane) {
    uint32_t *const W = reinterpret_cast<uint32_t *>(S.table);
    uint32_t *const ebits = W, *const hist = W + 256;                 // bit window (256 words), 3 x 64 counts (LL, ML, OF)
    FseSymTT *const tt = reinterpret_cast<FseSymTT *>(W + 448);       // 3 x 64 symbol rows
    uint16_t *const stt = reinterpret_cast<uint16_t *>(W + 832);      // 3 x 256 states
    uint32_t *const cum = W + 1216, *const cur = W + 1280, *const hbits = W + 1344;  // 64, 64, 32 words
    for (uint32_t i = lane; i < 448; i += 64) W[i] = 0;
    __builtin_amdgcn_wave_barrier();
    for (uint32_t i = lane; i < nseq; i += 64) {
        const uint32_t ll = seqs[3 * i], mlb = seqs[3 * i + 1], ov = seqs[3 * i + 2];
        const uint32_t lc = ll < 64 ? c_ll_code[ll] : (uint32_t)hib(ll) + 19;
        const uint32_t mc = mlb < 128 ? c_ml_code[mlb] : (uint32_t)hib(mlb) + 36;
        atomicAdd(&hist[lc], 1u); atomicAdd(&hist[64 + mc], 1u); atomicAdd(&hist[128 + (uint32_t)hib(ov)], 1u);
    }
    __builtin_amdgcn_wave_barrier();
    uint32_t tl = (uint32_t)hib(nseq) - 2;
    tl = tl < 6 ? 6 : (tl > 8 ? 8 : tl);
    uint32_t lg[3] = {0, 0, 0}, modes = 0;  // table log per kind (0 = RLE), kinds: 0 LL, 1 ML, 2 OF
    uint8_t *q = p + 1;
    for (uint32_t t = 0; t < 3; t++) {  // section order: LL, OF, ML
        const uint32_t k = t == 0 ? 0u : (t == 1 ? 2u : 1u);
        const uint32_t c = hist[k * 64 + lane];
        const uint64_t present = __ballot(c != 0);
        const uint32_t last = 63u - (uint32_t)__clzll(present);
        const uint32_t mode_shift = k == 0 ? 6u : (k == 2 ? 4u : 2u);
        if (__popcll(present) == 1) {  // RLE_Mode: one byte, a table of one state that costs no bits
            modes |= 1u << mode_shift;
            if (lane == 0) { *q = (uint8_t)last; tt[k * 64 + last].delta_nb_bits = 0; tt[k * 64 + last].delta_find_state = 0; stt[k * 256] = 0; }
            q += 1;
            continue;
        }
        modes |= 2u << mode_shift;
        lg[k] = tl;
        q += fse_build(c, nseq, tl, tt + k * 64, stt + k * 256, cum, cur, hbits, q, lane);
        __builtin_amdgcn_wave_barrier();
    }
    if (lane == 0) *p = (uint8_t)modes;
    __builtin_amdgcn_wave_barrier();
    // ---- bitstream: encode_sequences' batch loop, state tables in LDS ----
    p = q;
    const uint16_t *const s_ll = stt, *const s_ml = stt + 256, *const s_of = stt + 512;
    uint32_t vl = 0, vm = 0, vo = 0;
    uint32_t bitpos = 0;
    for (uint32_t e0 = 0; e0 < nseq; e0 += 64) {
        const uint32_t cnt = nseq - e0 < 64 ? nseq - e0 : 64;
        const bool on = lane < cnt;
        const uint32_t i = nseq - 1 - (e0 + (on ? lane : 0));
        const uint32_t ll = seqs[3 * i], mlb = seqs[3 * i + 1], ov = seqs[3 * i + 2];
        const uint32_t lc = ll < 64 ? c_ll_code[ll] : (uint32_t)hib(ll) + 19;
        const uint32_t mc = mlb < 128 ? c_ml_code[mlb] : (uint32_t)hib(mlb) + 36;
        const uint32_t oc = (uint32_t)hib(ov);
        const FseSymTT tl_ = tt[lc], tm_ = tt[64 + mc], to_ = tt[128 + oc];
        uint32_t f_l = 0, f_m = 0, f_o = 0;
        for (uint32_t j = 0; j < cnt; j++) {
            const uint32_t dl = rdlane(tl_.delta_nb_bits, j), fl = rdlane((uint32_t)tl_.delta_find_state, j);
            const uint32_t dm = rdlane(tm_.delta_nb_bits, j), fm = rdlane((uint32_t)tm_.delta_find_state, j);
            const uint32_t d_o = rdlane(to_.delta_nb_bits, j), fo = rdlane((uint32_t)to_.delta_find_state, j);
            if (e0 + j == 0) {
                const uint32_t nbm = (dm + (1u << 15)) >> 16, nbo = (d_o + (1u << 15)) >> 16, nbl = (dl + (1u << 15)) >> 16;
                const uint32_t xm = s_ml[(uint32_t)((int32_t)(((nbm << 16) - dm) >> nbm) + (int32_t)fm)];
                const uint32_t xo = s_of[(uint32_t)((int32_t)(((nbo << 16) - d_o) >> nbo) + (int32_t)fo)];
                const uint32_t xl = s_ll[(uint32_t)((int32_t)(((nbl << 16) - dl) >> nbl) + (int32_t)fl)];
                vm = suni(xm); vo = suni(xo); vl = suni(xl);
            } else {
                const uint32_t nbo = (vo + d_o) >> 16, nbm = (vm + dm) >> 16, nbl = (vl + dl) >> 16;
                if (lane == j) {
                    f_o = (vo & ((1u << nbo) - 1)) | (nbo << 16);
                    f_m = (vm & ((1u << nbm) - 1)) | (nbm << 16);
                    f_l = (vl & ((1u << nbl) - 1)) | (nbl << 16);
                }
                const uint32_t xo = s_of[(uint32_t)((int32_t)(vo >> nbo) + (int32_t)fo)];
                const uint32_t xm = s_ml[(uint32_t)((int32_t)(vm >> nbm) + (int32_t)fm)];
                const uint32_t xl = s_ll[(uint32_t)((int32_t)(vl >> nbl) + (int32_t)fl)];
                vo = suni(xo); vm = suni(xm); vl = suni(xl);
            }
        }
        uint64_t lo = 0, hi = 0;
        uint32_t T = 0;
        auto put = [&](uint32_t v, uint32_t nb) {  // nb <= 17, v < 2^nb
            if (T < 64) {
                lo |= (uint64_t)v << T;
                if (T + nb > 64) hi |= (uint64_t)v >> (64 - T);
            } else hi |= (uint64_t)v << (T - 64);
            T += nb;
        };
        if (on) {
            put(f_o & 0xFFFF, f_o >> 16);
            put(f_m & 0xFFFF, f_m >> 16);
            put(f_l & 0xFFFF, f_l >> 16);
            put(ll - c_ll_base_e[lc], c_ll_bits_e[lc]);
            put(mlb + 3 - c_ml_base_e[mc], c_ml_bits_e[mc]);
            put(ov - (1u << oc), oc);
        }
        uint32_t inc = T;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const uint32_t y = __shfl_up(inc, d);
            if (lane >= (uint32_t)d) inc += y;
        }
        const uint32_t pos = bitpos + inc - T;
        if (T) {
            const uint32_t sft = pos & 31, w = pos >> 5;
            const uint64_t x0 = lo << sft, x1 = (hi << sft) | (sft ? lo >> (64 - sft) : 0ull);
            const uint32_t w0 = (uint32_t)x0, w1 = (uint32_t)(x0 >> 32), w2 = (uint32_t)x1, w3 = (uint32_t)(x1 >> 32);
            if (w0) atomicOr(&ebits[w], w0);
            if (w1) atomicOr(&ebits[w + 1], w1);
            if (w2) atomicOr(&ebits[w + 2], w2);
            if (w3) atomicOr(&ebits[w + 3], w3);
        }
        bitpos = suni(bitpos + __shfl(inc, 63));
        __builtin_amdgcn_wave_barrier();
        const uint32_t ndw = bitpos >> 5;
        uint32_t keep = 0;
        for (uint32_t k2 = lane; k2 < ndw; k2 += 64) {
            const uint32_t w = ebits[k2];
            __builtin_memcpy(p + 4 * k2, &w, 4);
        }
        keep = ebits[ndw];
        __builtin_amdgcn_wave_barrier();
        for (uint32_t k2 = lane; k2 <= ndw; k2 += 64) ebits[k2] = 0;
        __builtin_amdgcn_wave_barrier();
        if (lane == 0) ebits[0] = keep;
        __builtin_amdgcn_wave_barrier();
        p += 4 * ndw;
        bitpos &= 31;
    }
    const uint32_t fin = lg[1] + lg[2] + lg[0] + 1;  // final states (ML, OF, LL) and the closing bit
    if (lane == 0) {
        uint64_t acc = ebits[0];
        uint32_t nb = bitpos;
        acc |= (uint64_t)(vm & ((1u << lg[1]) - 1)) << nb; nb += lg[1];
        acc |= (uint64_t)(vo & ((1u << lg[2]) - 1)) << nb; nb += lg[2];
        acc |= (uint64_t)(vl & ((1u << lg[0]) - 1)) << nb; nb += lg[0];
        acc |= 1ull << nb; nb += 1;
        for (uint32_t k2 = 0; k2 * 8 < nb; k2++) p[k2] = (uint8_t)(acc >> (8 * k2));
    }
    return p + ((bitpos + fin + 7) >> 3);
}

// Huffman-coded literals section (RFC 8878 §4.2.1) for the wide variant, built by the whole wave:
//   histogram (LDS atomics) -> code lengths <= 11 (Shannon lengths, then greedy repair until the Kraft sum is
//   exactly 1: lengthen the rarest symbols while over-subscribed, shorten the most frequent ones that fit the gap)
//   -> canonical codes in the decoder's order (weight ascending, symbol ascending) -> direct 4-bit weight
//   description -> 1 or 4 backward bitstreams, 64 symbols per step (code lengths scanned into bit positions,
//   codes OR-ed into an LDS window, full dwords streamed out).
// The streams are first written to `tmp` (behind the raw literals) and then moved in front of them.  Returns the
// size of the section written at `dst` (header included), or 0 to keep raw literals: alphabet beyond 128 symbols
// (would need FSE-compressed weights), a single symbol, or no gain.
constexpr uint32_t HUF_MIN_LITS = 256, HUF_MAX_BITS = 11;
constexpr uint32_t CUSTOM_FSE_MIN_SEQ = 128;  // below this the three table descriptions cost more than they save
template <class SH>
__device__ __forceinline__ uint32_t huf_literals(SH &S, const uint8_t *lits, uint32_t n, uint8_t *tmp, uint8_t *dst, uint32_t lane) {
    uint32_t *const T = reinterpret_cast<uint32_t *>(S.table);  // the hash table is dead: hist | codes | bit window
    uint32_t *const hist = T, *const codes = T + 256, *const win = T + 512;
    for (uint32_t i = lane; i < 512 + 64; i += 64) T[i] = 0;
    __builtin_amdgcn_wave_barrier();
    // histogram in two stages: the first 4 KiB decide whether the alphabet can fit the direct weight form at all
    // (binary / incompressible literals leave here after 16 steps instead of 512)
    for (uint32_t stage = 0; stage < 2; stage++) {
        const uint32_t lo = stage ? 4096u : 0u, hi = stage ? n : (n < 4096u ? n : 4096u);
        for (uint32_t i = lo + lane * 4; i < hi; i += 256) {
            if (i + 4 <= hi) {
                const uint32_t v = ld32(lits + i);
                atomicAdd(&hist[v & 0xFF], 1u); atomicAdd(&hist[(v >> 8) & 0xFF], 1u);
                atomicAdd(&hist[(v >> 16) & 0xFF], 1u); atomicAdd(&hist[v >> 24], 1u);
            } else
                for (uint32_t k = i; k < hi; k++) atomicAdd(&hist[lits[k]], 1u);
        }
        __builtin_amdgcn_wave_barrier();
        // anything at or above 128 rules the direct weight form out
        if (__ballot((hist[lane + 128] | hist[lane + 192]) != 0)) return 0;
    }
    // lane owns symbols lane and lane + 64
    const uint32_t c0 = hist[lane], c1 = hist[lane + 64];
    const uint64_t p0 = __ballot(c0 != 0), p1 = __ballot(c1 != 0);
    const uint32_t nsym = (uint32_t)(__popcll(p0) + __popcll(p1));
    if (nsym < 2) return 0;
    const uint32_t last = p1 ? 64 + (63 - (uint32_t)__clzll(p1)) : 63 - (uint32_t)__clzll(p0);  // highest present symbol
    auto shannon = [&](uint32_t c) -> uint32_t {
        if (!c) return 0;
        uint32_t l = 1;
        while (l < HUF_MAX_BITS && ((uint64_t)c << l) < n) l++;
        return l;
    };
    uint32_t l0 = shannon(c0), l1 = shannon(c1);
    auto wsum = [&](uint32_t v) -> uint32_t {
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d);
        return v;
    };
    auto wmin = [&](uint32_t v) -> uint32_t {
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) { const uint32_t o = __shfl_xor(v, d); v = o < v ? o : v; }
        return v;
    };
    auto wmax = [&](uint32_t v) -> uint32_t {
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) { const uint32_t o = __shfl_xor(v, d); v = o > v ? o : v; }
        return v;
    };
    const uint32_t FULL = 1u << HUF_MAX_BITS;
    uint32_t K = wsum((l0 ? FULL >> l0 : 0) + (l1 ? FULL >> l1 : 0));
    // over-subscribed (only through the 11-bit clamp): lengthen the rarest symbol that still can be
    for (uint32_t guard = 0; K > FULL && guard < 4096; guard++) {
        const uint32_t k0 = (l0 && l0 < HUF_MAX_BITS) ? (c0 << 8) | lane : 0xFFFFFFFFu;
        const uint32_t k1 = (l1 && l1 < HUF_MAX_BITS) ? (c1 << 8) | (lane + 64) : 0xFFFFFFFFu;
        const uint32_t best = wmin(k0 < k1 ? k0 : k1);
        if (best == 0xFFFFFFFFu) return 0;
        const uint32_t sym = best & 0xFF;
        if (sym == lane) { K -= 0; l0++; }
        if (sym == lane + 64) l1++;
        K = wsum((l0 ? FULL >> l0 : 0) + (l1 ? FULL >> l1 : 0));
    }
    if (K > FULL) return 0;
    // under-subscribed: shorten the most frequent symbol whose gain still fits the gap
    for (uint32_t guard = 0; K < FULL && guard < 4096; guard++) {
        const uint32_t gap = FULL - K;
        const uint32_t k0 = (l0 > 1 && (FULL >> l0) <= gap) ? (c0 << 8) | lane : 0;
        const uint32_t k1 = (l1 > 1 && (FULL >> l1) <= gap) ? (c1 << 8) | (lane + 64) : 0;
        const uint32_t best = wmax(k0 > k1 ? k0 : k1);
        if (!best) return 0;
        const uint32_t sym = best & 0xFF;
        if (sym == lane && k0 == best) l0--;
        else if (sym == lane + 64 && k1 == best) l1--;
        K = wsum((l0 ? FULL >> l0 : 0) + (l1 ? FULL >> l1 : 0));
    }
    if (K != FULL) return 0;
    const uint32_t L = wmax(l0 > l1 ? l0 : l1);  // table log; weight = L + 1 - length
    const uint32_t w0 = l0 ? L + 1 - l0 : 0, w1 = l1 ? L + 1 - l1 : 0;
    // payload bits, known before a single bit is written
    const uint32_t pay_bits = wsum(c0 * l0 + c1 * l1);
    const uint32_t nstreams = n < 1024 ? 1u : 4u;
    const uint32_t tree_bytes = 1 + (last + 1) / 2;
    const uint32_t est = tree_bytes + (nstreams == 4 ? 6u : 0u) + (pay_bits + 7) / 8 + nstreams;  // + end marks
    if (est + 5 + (n >> 6) >= n) return 0;  // not worth a Huffman stage on the decode side
    // canonical codes: positions in the 2^L table by (weight, symbol) order; code = position >> (weight - 1)
    uint32_t base = 0, code0 = 0, code1 = 0;
    const uint64_t below = lane ? (~0ull >> (64 - lane)) : 0ull;
    for (uint32_t w = 1; w <= L; w++) {
        const uint64_t m0 = __ballot(w0 == w), m1 = __ballot(w1 == w);
        const uint32_t n0 = (uint32_t)__popcll(m0), n1 = (uint32_t)__popcll(m1);
        if (w0 == w) code0 = (base >> (w - 1)) + (uint32_t)__popcll(m0 & below);
        if (w1 == w) code1 = (base >> (w - 1)) + n0 + (uint32_t)__popcll(m1 & below);
        base += (n0 + n1) << (w - 1);
    }
    codes[lane] = code0 | (l0 << 16);
    codes[lane + 64] = code1 | (l1 << 16);
    __builtin_amdgcn_wave_barrier();
    // streams into tmp
    const uint32_t seg = (n + 3) / 4;
    uint32_t ssz[4] = {0, 0, 0, 0};
    uint8_t *q = tmp;
    for (uint32_t st = 0; st < nstreams; st++) {
        const uint32_t a = nstreams == 1 ? 0 : st * seg;
        const uint32_t b = nstreams == 1 ? n : (st == 3 ? n : a + seg);
        uint8_t *const q0 = q;
        uint32_t bitpos = 0;
        for (uint32_t e0 = 0; e0 < b - a; e0 += 64) {
            const uint32_t idx = e0 + lane;
            uint32_t cd = 0, ln = 0;
            if (idx < b - a) {
                const uint32_t cw = codes[lits[b - 1 - idx]];  // last symbol first
                cd = cw & 0xFFFF; ln = cw >> 16;
            }
            uint32_t inc = ln;
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) {
                const uint32_t y = __shfl_up(inc, d);
                if (lane >= (uint32_t)d) inc += y;
            }
            const uint32_t pos = bitpos + inc - ln;
            if (ln) {
                const uint64_t x = (uint64_t)cd << (pos & 31);
                atomicOr(&win[pos >> 5], (uint32_t)x);
                if (x >> 32) atomicOr(&win[(pos >> 5) + 1], (uint32_t)(x >> 32));
            }
            bitpos = suni(bitpos + __shfl(inc, 63));
            __builtin_amdgcn_wave_barrier();
            const uint32_t ndw = bitpos >> 5;
            if (lane < ndw) {
                const uint32_t w = win[lane];
                __builtin_memcpy(q + 4 * lane, &w, 4);
            }
            const uint32_t keep = win[ndw];
            __builtin_amdgcn_wave_barrier();
            if (lane <= ndw) win[lane] = 0;
            __builtin_amdgcn_wave_barrier();
            if (lane == 0) win[0] = keep;
            __builtin_amdgcn_wave_barrier();
            q += 4 * ndw;
            bitpos &= 31;
        }
        // closing bit, remaining bytes
        const uint32_t tail_bits = bitpos + 1, tail_bytes = (tail_bits + 7) >> 3;
        if (lane == 0) {
            const uint64_t acc = (uint64_t)win[0] | (1ull << bitpos);
            for (uint32_t k = 0; k < tail_bytes; k++) q[k] = (uint8_t)(acc >> (8 * k));
            win[0] = 0;
        }
        __builtin_amdgcn_wave_barrier();
        q += tail_bytes;
        ssz[st] = (uint32_t)(q - q0);
    }
    const uint32_t streams_bytes = (uint32_t)(q - tmp);
    const uint32_t comp = tree_bytes + (nstreams == 4 ? 6u : 0u) + streams_bytes;
    // size_format: 0 = 1 stream, 10-bit sizes; 2 = 4 streams, 14-bit; 3 = 4 streams, 18-bit
    uint32_t sf, hdr;
    if (nstreams == 1) { sf = 0; hdr = 3; if (comp > 1023) return 0; }
    else if (n <= 16383 && comp <= 16383) { sf = 2; hdr = 4; }
    else { sf = 3; hdr = 5; }
    if (hdr + comp + (n >> 6) >= 3 + n || (nstreams == 4 && (ssz[0] > 65535 || ssz[1] > 65535 || ssz[2] > 65535))) return 0;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the streams in tmp have landed before they are moved
    if (lane == 0) {
        const uint32_t nb = sf == 0 ? 10u : (sf == 2 ? 14u : 18u);
        const uint64_t h = 2ull | ((uint64_t)sf << 2) | ((uint64_t)n << 4) | ((uint64_t)comp << (4 + nb));
        for (uint32_t k = 0; k < hdr; k++) dst[k] = (uint8_t)(h >> (8 * k));
        dst[hdr] = (uint8_t)(127 + last);  // direct weights for symbols 0 .. last-1, the last one is implied
    }
    {
        // weight nibbles: byte i = weight[2i] << 4 | weight[2i+1]
        const uint32_t nbytes = (last + 1) / 2;
        for (uint32_t i = lane; i < nbytes; i += 64) {
            const uint32_t la = codes[2 * i] >> 16, lb = (2 * i + 1 < last) ? codes[2 * i + 1] >> 16 : 0;
            const uint32_t wa = la ? L + 1 - la : 0, wb = lb ? L + 1 - lb : 0;
            dst[hdr + 1 + i] = (uint8_t)((wa << 4) | wb);
        }
    }
    uint8_t *d = dst + hdr + tree_bytes;
    if (nstreams == 4) {
        if (lane == 0) {
            d[0] = (uint8_t)ssz[0]; d[1] = (uint8_t)(ssz[0] >> 8);
            d[2] = (uint8_t)ssz[1]; d[3] = (uint8_t)(ssz[1] >> 8);
            d[4] = (uint8_t)ssz[2]; d[5] = (uint8_t)(ssz[2] >> 8);
        }
        d += 6;
    }
    wave_copy(d, tmp, streams_bytes, lane);  // moves down over the raw literals: ranges do not overlap (comp < n)
    return hdr + comp;
}

// The higher effort tier's Huffman literals: huf_literals for any alphabet.  A lane owns four symbols (lane, +64,
// +128, +192); a tree whose highest symbol is beyond 127 cannot be described by direct 4-bit weights and goes out as
// FSE-compressed weights (RFC 8878 4.2.1.2): the weights of symbols 0 .. last-1 coded with one table of Accuracy_Log 6
// (fse_build) by two interleaved states, even positions on the first — lane 0 walks the at most 255 weights backwards.
// Binaries' literals are such alphabets; the fast tier leaves them raw.
template <class SH>
__device__ __forceinline__ uint32_t huf_literals_any(SH &S, const uint8_t *lits, uint32_t n, uint8_t *tmp, uint8_t *dst, uint32_t lane) {
    uint32_t *const T = reinterpret_cast<uint32_t *>(S.table);  // the hash table is dead: hist | codes | bit window | weight coder
    uint32_t *const hist = T, *const codes = T + 256, *const win = T + 512;
    FseSymTT *const wtt = reinterpret_cast<FseSymTT *>(T + 640);    // 64 rows
    uint16_t *const wstt = reinterpret_cast<uint16_t *>(T + 768);   // 64 states (room for 256)
    uint32_t *const cum = T + 896, *const cur = T + 960, *const hbits = T + 1024, *const whist = T + 1120;
    uint8_t *const wbuf = reinterpret_cast<uint8_t *>(T + 1056);    // 256 bytes: the compressed weights
    for (uint32_t i = lane; i < 512 + 128; i += 64) T[i] = 0;
    if (lane < 16) whist[lane] = 0;
    __builtin_amdgcn_wave_barrier();
    for (uint32_t i = lane * 4; i < n; i += 256) {
        if (i + 4 <= n) {
            const uint32_t v = ld32(lits + i);
            atomicAdd(&hist[v & 0xFF], 1u); atomicAdd(&hist[(v >> 8) & 0xFF], 1u);
            atomicAdd(&hist[(v >> 16) & 0xFF], 1u); atomicAdd(&hist[v >> 24], 1u);
        } else
            for (uint32_t k = i; k < n; k++) atomicAdd(&hist[lits[k]], 1u);
    }
    __builtin_amdgcn_wave_barrier();
    uint32_t c[4], l[4];
    uint64_t pm[4];
    uint32_t nsym = 0, last = 0;
#pragma unroll
    for (int j = 0; j < 4; j++) {
        c[j] = hist[lane + 64 * j];
        pm[j] = __ballot(c[j] != 0);
        nsym += (uint32_t)__popcll(pm[j]);
        if (pm[j]) last = 64 * j + 63 - (uint32_t)__clzll(pm[j]);
    }
    if (nsym < 2) return 0;
    auto wsum = [&](uint32_t v) -> uint32_t {
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d);
        return v;
    };
    auto wmax = [&](uint32_t v) -> uint32_t {
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) { const uint32_t o = __shfl_xor(v, d); v = o > v ? o : v; }
        return v;
    };
    const uint32_t FULL = 1u << HUF_MAX_BITS;
    auto kraft = [&]() -> uint32_t {
        uint32_t k = 0;
#pragma unroll
        for (int j = 0; j < 4; j++) k += l[j] ? FULL >> l[j] : 0;
        return wsum(k);
    };
    auto wmin64 = [&](uint64_t v) -> uint64_t {
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) { const uint64_t o = __shfl_xor(v, d); v = o < v ? o : v; }
        return v;
    };
    auto wmax64 = [&](uint64_t v) -> uint64_t {
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) { const uint64_t o = __shfl_xor(v, d); v = o > v ? o : v; }
        return v;
    };
    // Code lengths: Shannon lengths ROUNDED to the nearest integer (c * 2^l * sqrt 2 >= n), then a repair that prices a
    // step by what it costs per unit of Kraft sum — count << length: lengthen where that is smallest while
    // over-subscribed, shorten where it is largest (and fits the gap) while under-subscribed.  On the literals of real
    // text and binaries this lands within 0.1 % of optimal length-limited codes (tools/enc_model.c, hufopt 2 against 1).
#pragma unroll
    for (int j = 0; j < 4; j++) {
        uint32_t len = 0;
        if (c[j]) { len = 1; while (len < HUF_MAX_BITS && ((uint64_t)c[j] << len) * 181u < (uint64_t)n * 128u) len++; }
        l[j] = len;
    }
    uint32_t K = kraft();
    for (uint32_t guard = 0; K > FULL && guard < 4096; guard++) {
        uint64_t key = ~0ull;
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const uint64_t kj = (l[j] && l[j] < HUF_MAX_BITS) ? (((uint64_t)c[j] << l[j]) << 8) | (lane + 64 * j) : ~0ull;
            key = kj < key ? kj : key;
        }
        const uint64_t best = wmin64(key);
        if (best == ~0ull) return 0;
        const uint32_t sym = (uint32_t)best & 0xFF;
#pragma unroll
        for (int j = 0; j < 4; j++)
            if (sym == lane + 64 * j) l[j]++;
        K = kraft();
    }
    if (K > FULL) return 0;
    for (uint32_t guard = 0; K < FULL && guard < 4096; guard++) {
        const uint32_t gap = FULL - K;
        uint64_t key = 0;
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const uint64_t kj = (l[j] > 1 && (FULL >> l[j]) <= gap) ? (((uint64_t)c[j] << l[j]) << 8) | (lane + 64 * j) : 0ull;
            key = kj > key ? kj : key;
        }
        const uint64_t best = wmax64(key);
        if (!best) return 0;
        const uint32_t sym = (uint32_t)best & 0xFF;
#pragma unroll
        for (int j = 0; j < 4; j++)
            if (sym == lane + 64 * j && l[j] > 1) l[j]--;
        K = kraft();
    }
    if (K != FULL) return 0;
    uint32_t lm = 0, pay = 0;
#pragma unroll
    for (int j = 0; j < 4; j++) { lm = l[j] > lm ? l[j] : lm; pay += c[j] * l[j]; }
    const uint32_t L = wmax(lm);  // table log; weight = L + 1 - length
    uint32_t w[4];
#pragma unroll
    for (int j = 0; j < 4; j++) w[j] = l[j] ? L + 1 - l[j] : 0;
    const uint32_t pay_bits = wsum(pay);
    const uint32_t nstreams = n < 1024 ? 1u : 4u;
    if (16 + (nstreams == 4 ? 6u : 0u) + (pay_bits + 7) / 8 + nstreams + 5 + (n >> 6) >= n) return 0;  // not worth a Huffman stage on the decode side
    // canonical codes: positions in the 2^L table by (weight, symbol) order; code = position >> (weight - 1)
    uint32_t base = 0, code[4] = {0, 0, 0, 0};
    const uint64_t below = lane ? (~0ull >> (64 - lane)) : 0ull;
    for (uint32_t ww = 1; ww <= L; ww++) {
        uint32_t before = 0;
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const uint64_t m = __ballot(w[j] == ww);
            if (w[j] == ww) code[j] = (base >> (ww - 1)) + before + (uint32_t)__popcll(m & below);
            before += (uint32_t)__popcll(m);
        }
        base += before << (ww - 1);
    }
#pragma unroll
    for (int j = 0; j < 4; j++) codes[lane + 64 * j] = code[j] | (l[j] << 16);
    // ---- tree description ----
    uint32_t tree_bytes;
    const bool direct = last < 128;
    if (direct) tree_bytes = 1 + (last + 1) / 2;
    else {
        // weights of symbols 0 .. last-1 (the last one is implied), FSE-compressed
#pragma unroll
        for (int j = 0; j < 4; j++)
            if (lane + 64 * j < last) atomicAdd(&whist[w[j]], 1u);
        __builtin_amdgcn_wave_barrier();
        const uint32_t wc = lane < 16 ? whist[lane] : 0;
        if (__popcll(__ballot(wc != 0)) < 2) return 0;  // one weight only: no table can say that (and no gain to expect)
        const uint32_t nc = fse_build(wc, last, 6, wtt, wstt, cum, cur, hbits, wbuf + 1, lane);
        uint32_t wlen = 0;
        if (lane == 0) {
            BitW b{wbuf + 1 + nc, 0, 0};
            CState s1, s2;
            bool have1 = false, have2 = false;
            for (int32_t i = (int32_t)last - 1; i >= 0; i--) {
                const uint32_t len = codes[i] >> 16, wt = len ? L + 1 - len : 0;
                if (i & 1) {
                    if (!have2) { s2.init(wstt, wtt, 6, wt); have2 = true; } else s2.encode(b, wt);
                } else {
                    if (!have1) { s1.init(wstt, wtt, 6, wt); have1 = true; } else s1.encode(b, wt);
                }
            }
            s2.flush(b);
            s1.flush(b);
            wlen = (uint32_t)(b.close() - (wbuf + 1));
            wbuf[0] = (uint8_t)wlen;
        }
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");  // lane 0's stores into wbuf (possibly FLAT) before the ds reads below
        wlen = suni(wlen);
        if (wlen >= 128 || wlen + 1 > 250) return 0;  // the header byte of compressed weights is their size, below 128
        tree_bytes = 1 + wlen;
    }
    __builtin_amdgcn_wave_barrier();
    // ---- streams into tmp ----
    const uint32_t seg = (n + 3) / 4;
    uint32_t ssz[4] = {0, 0, 0, 0};
    uint8_t *q = tmp;
    for (uint32_t st = 0; st < nstreams; st++) {
        const uint32_t a = nstreams == 1 ? 0 : st * seg;
        const uint32_t b = nstreams == 1 ? n : (st == 3 ? n : a + seg);
        uint8_t *const q0 = q;
        uint32_t bitpos = 0;
        for (uint32_t e0 = 0; e0 < b - a; e0 += 64) {
            const uint32_t idx = e0 + lane;
            uint32_t cd = 0, ln = 0;
            if (idx < b - a) {
                const uint32_t cw = codes[lits[b - 1 - idx]];  // last symbol first
                cd = cw & 0xFFFF; ln = cw >> 16;
            }
            uint32_t inc = ln;
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) {
                const uint32_t y = __shfl_up(inc, d);
                if (lane >= (uint32_t)d) inc += y;
            }
            const uint32_t pos = bitpos + inc - ln;
            if (ln) {
                const uint64_t x = (uint64_t)cd << (pos & 31);
                atomicOr(&win[pos >> 5], (uint32_t)x);
                if (x >> 32) atomicOr(&win[(pos >> 5) + 1], (uint32_t)(x >> 32));
            }
            bitpos = suni(bitpos + __shfl(inc, 63));
            __builtin_amdgcn_wave_barrier();
            const uint32_t ndw = bitpos >> 5;
            if (lane < ndw) {
                const uint32_t wv = win[lane];
                __builtin_memcpy(q + 4 * lane, &wv, 4);
            }
            const uint32_t keep = win[ndw];
            __builtin_amdgcn_wave_barrier();
            if (lane <= ndw) win[lane] = 0;
            __builtin_amdgcn_wave_barrier();
            if (lane == 0) win[0] = keep;
            __builtin_amdgcn_wave_barrier();
            q += 4 * ndw;
            bitpos &= 31;
        }
        const uint32_t tail_bits = bitpos + 1, tail_bytes = (tail_bits + 7) >> 3;
        if (lane == 0) {
            const uint64_t acc = (uint64_t)win[0] | (1ull << bitpos);
            for (uint32_t k = 0; k < tail_bytes; k++) q[k] = (uint8_t)(acc >> (8 * k));
            win[0] = 0;
        }
        __builtin_amdgcn_wave_barrier();
        q += tail_bytes;
        ssz[st] = (uint32_t)(q - q0);
    }
    const uint32_t streams_bytes = (uint32_t)(q - tmp);
    const uint32_t comp = tree_bytes + (nstreams == 4 ? 6u : 0u) + streams_bytes;
    uint32_t sf, hdr;
    if (nstreams == 1) { sf = 0; hdr = 3; if (comp > 1023) return 0; }
    else if (n <= 16383 && comp <= 16383) { sf = 2; hdr = 4; }
    else { sf = 3; hdr = 5; }
    if (hdr + comp + (n >> 6) >= 3 + n || (nstreams == 4 && (ssz[0] > 65535 || ssz[1] > 65535 || ssz[2] > 65535))) return 0;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the streams in tmp have landed before they are moved
    if (lane == 0) {
        const uint32_t nb = sf == 0 ? 10u : (sf == 2 ? 14u : 18u);
        const uint64_t h = 2ull | ((uint64_t)sf << 2) | ((uint64_t)n << 4) | ((uint64_t)comp << (4 + nb));
        for (uint32_t k = 0; k < hdr; k++) dst[k] = (uint8_t)(h >> (8 * k));
        if (direct) dst[hdr] = (uint8_t)(127 + last);  // direct weights for symbols 0 .. last-1, the last one is implied
    }
    if (direct) {
        const uint32_t nbytes = (last + 1) / 2;  // weight nibbles: byte i = weight[2i] << 4 | weight[2i+1]
        for (uint32_t i = lane; i < nbytes; i += 64) {
            const uint32_t la = codes[2 * i] >> 16, lb = (2 * i + 1 < last) ? codes[2 * i + 1] >> 16 : 0;
            const uint32_t wa = la ? L + 1 - la : 0, wb = lb ? L + 1 - lb : 0;
            dst[hdr + 1 + i] = (uint8_t)((wa << 4) | wb);
        }
    } else {
        for (uint32_t i = lane; i < tree_bytes; i += 64) dst[hdr + i] = wbuf[i];
    }
    uint8_t *d = dst + hdr + tree_bytes;
    if (nstreams == 4) {
        if (lane == 0) {
            d[0] = (uint8_t)ssz[0]; d[1] = (uint8_t)(ssz[0] >> 8);
            d[2] = (uint8_t)ssz[1]; d[3] = (uint8_t)(ssz[1] >> 8);
            d[4] = (uint8_t)ssz[2]; d[5] = (uint8_t)(ssz[2] >> 8);
        }
        d += 6;
    }
    wave_copy(d, tmp, streams_bytes, lane);  // moves down over the raw literals: ranges do not overlap (comp < n)
    return hdr + comp;
}

// One wave encodes one block item pulled from the atomic cursor.  HASH_LOG 11 (4 KiB table, more
// resident waves) serves batches of small rounds, 13 serves 128 KiB blocks.
// HIGH (with HASH_LOG 13 only) is the higher effort tier (levels >= HIGH_TIER_LEVEL): 8-way buckets, one-step lazy
// choice, a cost gate on short far matches, in-block repeat offsets.
template <uint32_t HASH_LOG, bool HIGH = false>
__global__ __launch_bounds__(64) void k_zstd_encode(EncodeArgs a) {
    constexpr uint32_t HASH_SIZE = 1u << HASH_LOG;
    __shared__ __attribute__((aligned(16))) EncShared<HASH_LOG> S;
    __shared__ uint32_t s_item, s_raw;
    const uint32_t lane = threadIdx.x;
    if (a.n_items_dev && *a.n_items_dev == 0) return;  // retry launch with nothing handed over (periodic data)
    {
        const uint32_t *g = reinterpret_cast<const uint32_t *>(a.tabs);
        uint32_t *l = reinterpret_cast<uint32_t *>(&S.tabs);
        for (uint32_t i = lane; i < sizeof(EncTables) / 4; i += 64) l[i] = g[i];
    }
    // {ll, ml-3, offset} per sequence: LDS for the small-block variant, per-wave global scratch otherwise
    uint32_t *const seqs = HASH_LOG == 11 ? S.seqs : a.seq_scratch + (size_t)blockIdx.x * MAX_SEQ * 3;

    // Work cursor: one atomic per `a.batch` consecutive items (a single word saturates near 88
    // dequeues/us on this chip, so 100k one-item dequeues alone would cost > 1 ms).
    uint32_t next = 0, lim = 0, first = 0;
    // descriptors of the dequeued batch, one item per lane (loaded together: one memory round trip
    // per batch instead of three dependent ones per item)
    const uint32_t n_items = a.n_items_dev ? *a.n_items_dev : a.n_items;  // retry launch: the count lives on the device
    uint32_t d_round = 0, d_block = 0, d_nblocks = 0, d_flags = ITEM_SKIP, d_id = 0;
    uint64_t d_prov = 0, d_rlen = 0, d_soff = 0;
    for (;;) {
        if (next == lim) {
            __syncthreads();
            uint32_t batch = a.batch;
            if (HASH_LOG == 11 && !HIGH && a.fuse_tiles) {
                // Fused mode: the unit of work is a hash tile.  The wave hashes the tiles' rounds first — VALU work that
                // overlaps, on the SIMD, with the memory waits of the waves that are encoding — and then encodes the same
                // rounds, whose bytes are now in L2: the input is fetched from HBM once.
                const uint32_t FUSE_TILES = (uint32_t)a.fuse_tiles;  // tiles per dequeue (1; more only through ZNIPPY_FUSE_TILES, and never more than 64 rounds)
                if (lane == 0) s_item = atomicAdd(a.cursor, FUSE_TILES);
                __syncthreads();
                const uint32_t t0 = s_item;
                if (t0 >= a.h.n_tiles) break;
                uint32_t r_first = 0, r_end = 0;
                for (uint32_t k = 0; k < FUSE_TILES && t0 + k < a.h.n_tiles; k++) {
                    const Tile t = a.h.tiles[t0 + k];
                    hash_tile<false>(a.h, t);
                    if (k == 0) r_first = t.first_unit;
                    r_end = t.first_unit + t.n_units;
                }
                next = first = r_first;
                batch = r_end - r_first;  // <= 64 rounds per tile
                lim = r_end;
            } else {
                if (lane == 0) s_item = atomicAdd(a.cursor, a.batch);
                __syncthreads();
                next = first = s_item;
                lim = next + a.batch;
            }
            const uint32_t mine = first + lane;
            d_flags = ITEM_SKIP;
            if (lane < batch && mine < n_items) {
                d_id = a.order ? a.order[mine] : mine;  // this variant's share of the plan (or all of it)
                const EncItem e = a.items[d_id];
                d_round = e.round; d_block = e.block; d_nblocks = e.n_blocks; d_flags = e.flags; d_prov = e.prov;
                d_rlen = a.len[e.round];
                d_soff = a.src_off[e.round];
            }
        }
        const uint32_t item = next++;
        if (item >= n_items) break;
        __syncthreads();
        const uint32_t bl = item - first;
        EncItem it;
        it.round = __shfl(d_round, bl); it.block = __shfl(d_block, bl); it.n_blocks = __shfl(d_nblocks, bl);
        it.flags = __shfl(d_flags, bl); it.prov = __shfl(d_prov, bl);
        const uint32_t item_id = __shfl(d_id, bl);
        if (it.flags & ITEM_SKIP) continue;  // store path: the gather pass copies it from the staging buffer
        const uint64_t rlen = __shfl(d_rlen, bl);
        const uint8_t *const rsrc = a.src + __shfl(d_soff, bl);
        const uint64_t boff = (uint64_t)it.block * BLOCK_BYTES;
        const uint32_t n = (uint32_t)((rlen - boff) < BLOCK_BYTES ? (rlen - boff) : BLOCK_BYTES);
        const uint8_t *const in = rsrc + boff;
        const bool last = it.block + 1 == it.n_blocks;
        uint8_t *const blk = a.prov + it.prov + HDR_ROOM;  // block header goes here
        uint8_t *const lits = blk + 3 + 3;                 // literals after a 3-byte literals header
        // sequence budget: bounds the provisional slot (worst case ~5.4 bytes per sequence) and the sequence store
        const uint32_t seq_div = HASH_LOG == 13 ? 8u : 40u;
        const uint32_t max_seq = (n / seq_div) < MAX_SEQ ? (n / seq_div) : MAX_SEQ;
        // touch every 128-byte line of the block now: all of its HBM fetches are in flight at once and
        // the match finder's dependent loads below hit in L2/L1.  LDS-DMA loads (no VGPR destination,
        // nothing ever reads the sink) so no register is exposed to a late-arriving result.
        for (uint32_t o = lane * 128; o + 4 <= n; o += 64 * 128)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(in + o),
                                             (__attribute__((address_space(3))) void *)S.sink, 4, 0, 0);
        unsigned long long t_last = (HASH_LOG == 13 && a.dbg) ? __builtin_amdgcn_s_memtime() : 0;
#define ESTAMP(i) do { if (HASH_LOG == 13 && a.dbg && lane == 0) { const unsigned long long now_ = __builtin_amdgcn_s_memtime(); atomicAdd(&a.dbg[i], now_ - t_last); t_last = now_; } } while (0)
        if (HASH_LOG == 13 && a.dbg && lane == 0) atomicAdd(&a.dbg[0], 1ull);
        // ---- match finding ----
        {
            uint4 *t4 = reinterpret_cast<uint4 *>(S.table);
            const uint4 ones = make_uint4(~0u, ~0u, ~0u, ~0u);
            for (uint32_t i = lane; i < HASH_SIZE / 8; i += 64) t4[i] = ones;
        }
        __syncthreads();
        ESTAMP(1);
        // Every block is self-contained: no match reaches in front of the block and no repeat-offset code is
        // used, so the blocks of a frame can be decoded independently (zstd_decode.hip, block items).  PRE > 0
        // would seed the block with the bytes in front of it (saves one literal run per block on periodic data).
        constexpr uint32_t PRE = 0;
        const uint8_t *const inb = in - PRE;
        const uint32_t nq = n + PRE;
        uint32_t nseq = 0, lit_total = 0;
        uint32_t anchor = PRE;  // first byte not yet covered by a sequence
        uint32_t emitted = PRE; // wide variant: literal bytes [anchor, emitted) are already written
        uint32_t base = 0, misses = 0;
        bool gave_up = false;  // wide variant: incompressible block recognised early
        const uint32_t scan_end = nq >= 8 ? nq - 7 : 0;  // positions with >= 8 bytes ahead
        if constexpr (HIGH) {
            // Higher effort tier of the wide variant.  Same shape (every lane probes its position and extends its own
            // candidates, the window's matches are then picked left to right), with what tools/enc_model.c showed to pay:
            //   * 2^10 buckets x 8 ways in the same 16 KiB (one 16-byte LDS word per bucket, newest first): associativity
            //     beats table size on text and binaries alike (2^13 x 1 -> 2^11 x 4 -> 2^10 x 8: 0.297, 0.282, 0.279 in the model);
            //   * candidates at the three repeat offsets of the window's start, preferred when within a byte of the best;
            //   * a gate on short far matches (4 bytes beyond 2 KiB, 5 beyond 32 KiB cost more than their literals);
            //   * one-step lazy choice (the next position's match wins when it is two bytes longer);
            //   * offsets equal to one of the block's last three go out as repeat codes.  The history starts unknown
            //     (0): blocks are encoded independently, so a code is only used once this block has defined its entry.
            constexpr uint32_t LMAX = 64;
            uint4 *const B = reinterpret_cast<uint4 *>(S.table);
            uint32_t r0 = 0, r1 = 0, r2 = 0;
            while (base < scan_end && nseq < max_seq) {
                const uint32_t pos = base + lane;
                uint32_t cand = 0, mlen = 0, v = 0, back = 0;
                auto store_tail = [&]() {  // this window's bytes that no match of the window covers
                    const uint32_t wend = base + 64 < nq ? base + 64 : nq;
                    if (emitted < base) {  // a stretch skipped while accelerating through incompressible data
                        wave_copy(lits + lit_total + (emitted - anchor), inb + emitted, base - emitted, lane);
                        emitted = base;
                    }
                    if (emitted < wend) {
                        if (pos >= emitted && pos < wend) lits[lit_total + (pos - anchor)] = pos < scan_end ? (uint8_t)v : inb[pos];
                        emitted = wend;
                    }
                };
                const uint32_t r0w = r0, r1w = r1, r2w = r2;  // the history at the window's start
                if (pos < scan_end) {
                    v = ld32(inb + pos);
                    const uint32_t h = hash4<10>(v);
                    const uint4 b = B[h];
                    B[h] = make_uint4((b.x << 16) | (pos & 0xFFFFu), (b.y << 16) | (b.x >> 16), (b.z << 16) | (b.y >> 16), (b.w << 16) | (b.z >> 16));
                    uint32_t c[8] = {b.x & 0xFFFFu, b.x >> 16, b.y & 0xFFFFu, b.y >> 16, b.z & 0xFFFFu, b.z >> 16, b.w & 0xFFFFu, b.w >> 16};
                    const int nways = misses >= 8 ? 2 : 8;  // eight windows without one repeat: incompressible so far, probe lightly
                    // All candidates of the position — the bucket's eight ways and the three offsets of the history — are
                    // checked and extended TOGETHER: one 4-byte load each, then one loop in which a step loads the
                    // position's next 8 bytes once and every candidate's that is still alive, instead of one extension loop
                    // per candidate one after the other.  Lengths, and with them the choice below, are what the separate loops
                    // gave (the common prefix, capped at LMAX and at the block's end): the blobs of the real-data corpora are
                    // bit-identical (tools/ratio.py prints their SHA-1).  Together with the 24-byte first load below the matcher
                    // is 13-18 % faster (binaries 75 -> 65 ms, text 25 -> 21 ms); a window is still ~15,000 cycles on binaries
                    // (ZNIPPY_EDBG): what is left is the address processing of the wave's scattered loads, three dozen
                    // instructions of 64 different lines each.
                    uint32_t cc[11], len[11];
                    uint32_t alive = 0;
                    const uint32_t rw[3] = {r0w, r1w, r2w};
#pragma unroll
                    for (int w = 0; w < 8; w++) {
                        const bool have = c[w] != 0xFFFFu && w < nways;
                        c[w] |= pos & ~0xFFFFu;
                        if (c[w] >= pos) c[w] -= 0x10000u;  // wraps to a huge value when there is no earlier half
                        cc[w] = (!have || c[w] >= pos) ? 0xFFFFFFFFu : c[w];
                    }
#pragma unroll
                    for (int i = 0; i < 3; i++) cc[8 + i] = (rw[i] && pos >= rw[i]) ? pos - rw[i] : 0xFFFFFFFFu;
                    uint32_t bk[11];
                    {
                        const uint32_t lim = nq - pos < LMAX ? nq - pos : LMAX;
                        uint32_t k = 4;
                        const bool wide_first = misses < 8 && pos >= 8 && pos + 16 <= nq;  // (not through incompressible stretches: there nearly every candidate fails its first 4 bytes)
                        if (wide_first) {
                            // the first round trip brings 24 bytes around every candidate (8 in front, 16 from it on) and around the
                            // position: most matches end inside them, and what a chosen match can take over in front of it is
                            // known without a round trip of its own
                            uint64_t x0, x1, x2;
                            __builtin_memcpy(&x0, inb + pos - 8, 8);
                            __builtin_memcpy(&x1, inb + pos, 8);
                            __builtin_memcpy(&x2, inb + pos + 8, 8);
#pragma unroll
                            for (int w = 0; w < 11; w++) {
                                len[w] = 0; bk[w] = 0;
                                if (cc[w] != 0xFFFFFFFFu) {
                                    uint64_t y0 = ~x0, y1, y2;
                                    __builtin_memcpy(&y1, inb + cc[w], 8);
                                    __builtin_memcpy(&y2, inb + cc[w] + 8, 8);
                                    if (cc[w] >= 8) __builtin_memcpy(&y0, inb + cc[w] - 8, 8);
                                    const uint64_t d1 = x1 ^ y1, d2 = x2 ^ y2, d0 = x0 ^ y0;
                                    if ((uint32_t)d1 == 0) {
                                        if (d1) len[w] = (uint32_t)(__ffsll((long long)d1) - 1) >> 3;
                                        else if (d2) len[w] = 8 + ((uint32_t)(__ffsll((long long)d2) - 1) >> 3);
                                        else { len[w] = 16; if (lim > 16) alive |= 1u << w; }
                                        bk[w] = cc[w] >= 8 ? (d0 ? (uint32_t)__clzll((long long)d0) >> 3 : 8u) : 0u;
                                    }
                                }
                            }
                            k = 16;
                        } else {
#pragma unroll
                            for (int w = 0; w < 11; w++) {
                                const uint32_t cvw = cc[w] != 0xFFFFFFFFu ? ld32(inb + cc[w]) : ~v;
                                len[w] = 0; bk[w] = 0;
                                if (cvw == v && cc[w] != 0xFFFFFFFFu) { alive |= 1u << w; len[w] = 4; }
                            }
                        }
                        while (alive && k + 8 <= lim) {
                            uint64_t x;
                            __builtin_memcpy(&x, inb + pos + k, 8);
#pragma unroll
                            for (int w = 0; w < 11; w++)
                                if (alive & (1u << w)) {
                                    uint64_t y;
                                    __builtin_memcpy(&y, inb + cc[w] + k, 8);
                                    const uint64_t d = x ^ y;
                                    if (d) { len[w] = k + ((uint32_t)(__ffsll((long long)d) - 1) >> 3); alive &= ~(1u << w); }
                                    else len[w] = k + 8;
                                }
                            k += 8;
                        }
#pragma unroll
                        for (int w = 0; w < 11; w++)
                            if (alive & (1u << w)) {  // the last few bytes in front of the cap / the block's end
                                uint32_t kk = len[w];
                                while (kk < lim && inb[pos + kk] == inb[cc[w] + kk]) kk++;
                                len[w] = kk;
                            }
                    }
                    uint32_t pick = 11;
#pragma unroll
                    for (int w = 0; w < 8; w++)
                        if (len[w] > mlen) { mlen = len[w]; cand = cc[w]; pick = w; }  // ties stay with the newer (closer) one
                    if (mlen) {
                        const uint32_t off = pos - cand;
                        if ((mlen == 4 && off > 2048u) || (mlen == 5 && off > 32768u)) mlen = 0;
                    }
                    {   // the three offsets of the history: a repeat code is worth a byte of match length
                        uint32_t score = mlen;
#pragma unroll
                        for (int i = 0; i < 3; i++)
                            if (len[8 + i]) {
                                const uint32_t k = len[8 + i];
                                if (k + 1 > score || (i == 0 && k + 1 >= score)) { score = k + 1; mlen = k; cand = cc[8 + i]; pick = 8 + i; }
                            }
                    }
                    if (mlen && cand >= 8) {  // bytes in front of the match that agree too (up to 8): literals it can take over
                        if (misses < 8 && pos >= 8 && pos + 16 <= nq) {
#pragma unroll
                            for (int w = 0; w < 11; w++)
                                if (pick == (uint32_t)w) back = bk[w];
                        } else {
                            uint64_t x, y;
                            __builtin_memcpy(&x, inb + pos - 8, 8);
                            __builtin_memcpy(&y, inb + cand - 8, 8);
                            const uint64_t d = x ^ y;
                            back = d ? (uint32_t)__clzll((long long)d) >> 3 : 8u;
                        }
                    }
                }
                const uint64_t hitm = __ballot(mlen != 0);
                if (!hitm) {
                    if (misses < 2) store_tail();
                    misses++;
                    if (nseq == 0 && misses >= 96) { gave_up = true; break; }
                    base += 64 * (1 + (misses >> 4 > 7 ? 7 : misses >> 4));
                    continue;
                }
                misses = 0;
                while (nseq < max_seq) {
                    const uint32_t skip = anchor > base ? anchor - base : 0;
                    const uint64_t m = skip >= 64 ? 0ull : (hitm >> skip) << skip;
                    if (!m) break;
                    uint32_t win = (uint32_t)__ffsll((long long)m) - 1;
                    uint32_t ml = rdlane(mlen, win);
                    if (win < 63 && ((m >> (win + 1)) & 1) && base + win - rdlane(cand, win) != r0) {
                        const uint32_t ml1 = rdlane(mlen, win + 1);
                        if (ml1 > ml + 1) { win++; ml = ml1; }
                    }
                    uint32_t mpos = base + win, mcand = rdlane(cand, win);
                    {   // take over the literals in front that agree with what is in front of the candidate
                        const uint32_t bk0 = rdlane(back, win), room = mpos - anchor;
                        const uint32_t bk = bk0 < room ? bk0 : room;
                        mpos -= bk; mcand -= bk; ml += bk;
                    }
                    if (ml >= LMAX) {
                        for (;;) {  // cooperative extension: 16 bytes per lane per piece, 4 pieces (4 KiB) in flight per step
                            uint32_t good[4];
#pragma unroll
                            for (int q = 0; q < 4; q++) {
                                const uint32_t o = ml + q * 1024 + lane * 16;
                                uint32_t g = 0;
                                if (mpos + o + 16 <= nq) {
                                    uint4 x = ld128(inb + mpos + o), y = ld128(inb + mcand + o);
                                    const uint32_t d0 = x.x ^ y.x, d1 = x.y ^ y.y, d2 = x.z ^ y.z, d3 = x.w ^ y.w;
                                    g = 16;
                                    if (d0 | d1 | d2 | d3) {
                                        if (d0) g = (__ffs(d0) - 1) >> 3;
                                        else if (d1) g = 4 + ((__ffs(d1) - 1) >> 3);
                                        else if (d2) g = 8 + ((__ffs(d2) - 1) >> 3);
                                        else g = 12 + ((__ffs(d3) - 1) >> 3);
                                    }
                                } else {
                                    while (g < 16 && mpos + o + g < nq && inb[mpos + o + g] == inb[mcand + o + g]) g++;
                                }
                                good[q] = g;
                            }
                            bool done = false;
#pragma unroll
                            for (int q = 0; q < 4; q++) {
                                if (done) break;
                                const uint64_t partial = __ballot(good[q] != 16);
                                if (partial) {
                                    const uint32_t fl = __ffsll((unsigned long long)partial) - 1;
                                    ml += fl * 16 + __shfl(good[q], fl);
                                    done = true;
                                } else {
                                    ml += 1024;
                                }
                            }
                            if (done) break;
                        }
                    }
                    const uint32_t ll = mpos - anchor, off = mpos - mcand;
                    if (emitted < mpos) {
                        if (emitted < base) {
                            wave_copy(lits + lit_total + (emitted - anchor), inb + emitted, base - emitted, lane);
                            emitted = base;
                        }
                        if (pos >= emitted && pos < mpos) lits[lit_total + (pos - anchor)] = (uint8_t)v;
                    }
                    // offset value (RFC 8878 3.1.1.3.2.1.1 / 3.1.1.5): 1..3 name the history, shifted by one when ll == 0
                    uint32_t ov = off + 3;
                    if (ll) {
                        if (off == r0) ov = 1;
                        else if (off == r1) { ov = 2; r1 = r0; r0 = off; }
                        else if (off == r2) { ov = 3; r2 = r1; r1 = r0; r0 = off; }
                        else { r2 = r1; r1 = r0; r0 = off; }
                    } else {
                        if (off == r1) { ov = 1; r1 = r0; r0 = off; }
                        else if (off == r2) { ov = 2; r2 = r1; r1 = r0; r0 = off; }
                        else if (r0 > 1 && off == r0 - 1) { ov = 3; r2 = r1; r1 = r0; r0 = off; }
                        else { r2 = r1; r1 = r0; r0 = off; }
                    }
                    if (lane == 0) {
                        seqs[3 * nseq] = ll;
                        seqs[3 * nseq + 1] = ml - 3;
                        seqs[3 * nseq + 2] = ov;
                    }
                    lit_total += ll;
                    nseq++;
                    anchor = mpos + ml;
                    emitted = anchor;
                }
                store_tail();
                base = anchor > base + 64 ? anchor : base + 64;
            }
        } else if constexpr (HASH_LOG == 13) {
            // Wide variant: every lane probes its position AND extends its own candidate (up to LMAX bytes), then the
            // window's matches are picked left to right — several sequences per 64 positions on real data instead
            // of one.  Only a match that runs past LMAX is extended cooperatively (periodic data: 4 KiB per step).
            constexpr uint32_t LMAX = 64;
            // Literals are written as the windows go by, straight from the bytes each lane has just loaded: input
            // bytes [anchor, emitted) already sit behind lits + lit_total.  A sequence then costs stores only — no
            // load/store round trip per match (that round trip was most of the matcher's time on real data).
            while (base < scan_end && nseq < max_seq) {
                const uint32_t pos = base + lane;
                uint32_t cand = 0, mlen = 0, v = 0;
                auto store_tail = [&]() {  // this window's bytes that no match of the window covers
                    const uint32_t wend = base + 64 < nq ? base + 64 : nq;
                    if (emitted < base) {  // a stretch skipped while accelerating through incompressible data
                        wave_copy(lits + lit_total + (emitted - anchor), inb + emitted, base - emitted, lane);
                        emitted = base;
                    }
                    if (emitted < wend) {
                        if (pos >= emitted && pos < wend) lits[lit_total + (pos - anchor)] = pos < scan_end ? (uint8_t)v : inb[pos];
                        emitted = wend;
                    }
                };
                if (pos < scan_end) {
                    v = ld32(inb + pos);
                    const uint32_t h = hash4<HASH_LOG>(v);
                    const uint32_t e = S.table[h];
                    S.table[h] = (uint16_t)pos;  // low 16 bits; candidates are within 64 KiB
                    if (e != 0xFFFF) {
                        uint32_t c = (pos & ~0xFFFFu) | e;
                        if (c >= pos) c -= 0x10000u;  // wraps to a huge value when there is no earlier half
                        if (c < pos && pos >= PRE && ld32(inb + c) == v) { cand = c; mlen = 4; }
                    }
                }
                if (mlen) {
                    const uint32_t lim = nq - pos < LMAX ? nq - pos : LMAX;
                    uint32_t k = 4;
                    bool open = true;
                    while (open && k + 8 <= lim) {
                        uint64_t x, y;
                        __builtin_memcpy(&x, inb + pos + k, 8);
                        __builtin_memcpy(&y, inb + cand + k, 8);
                        const uint64_t d = x ^ y;
                        if (d) { k += (uint32_t)(__ffsll((long long)d) - 1) >> 3; open = false; }
                        else k += 8;
                    }
                    while (open && k < lim && inb[pos + k] == inb[cand + k]) k++;
                    mlen = k;
                }
                const uint64_t hitm = __ballot(mlen != 0);
                if (!hitm) {
                    if (misses < 2) store_tail();  // a run of misses is incompressible data: leave it to one bulk copy later
                    misses++;
                    // 96 windows in a row (21 KiB of the block sampled, 64 positions each) without one 4-byte repeat, and no
                    // sequence so far: the block is incompressible data (BASELINE's random.bin).  Stop searching — every
                    // further window is another dependent memory round trip — and let the block go out raw.
                    if (nseq == 0 && misses >= 96) { gave_up = true; break; }
                    base += 64 * (1 + (misses >> 4 > 7 ? 7 : misses >> 4));  // accelerate through incompressible data
                    continue;
                }
                misses = 0;
                while (nseq < max_seq) {
                    // leftmost candidate at or after the first byte not yet emitted
                    const uint32_t skip = anchor > base ? anchor - base : 0;
                    const uint64_t m = skip >= 64 ? 0ull : (hitm >> skip) << skip;
                    if (!m) break;
                    const uint32_t win = (uint32_t)__ffsll((long long)m) - 1;
                    const uint32_t mpos = base + win, mcand = rdlane(cand, win);
                    uint32_t ml = rdlane(mlen, win);
                    if (ml >= LMAX) {
                // cooperative extension: 16 bytes per lane per piece, 4 pieces (4 KiB) in flight per step
                for (;;) {
                    uint32_t good[4];
    #pragma unroll
                    for (int q = 0; q < 4; q++) {
                        const uint32_t o = ml + q * 1024 + lane * 16;
                        uint32_t g = 0;  // matching bytes in my 16-byte piece
                        if (mpos + o + 16 <= nq) {
                            uint4 x = ld128(inb + mpos + o), y = ld128(inb + mcand + o);
                            const uint32_t d0 = x.x ^ y.x, d1 = x.y ^ y.y, d2 = x.z ^ y.z, d3 = x.w ^ y.w;
                            g = 16;
                            if (d0 | d1 | d2 | d3) {  // rare: only the piece where the match ends
                                if (d0) g = (__ffs(d0) - 1) >> 3;
                                else if (d1) g = 4 + ((__ffs(d1) - 1) >> 3);
                                else if (d2) g = 8 + ((__ffs(d2) - 1) >> 3);
                                else g = 12 + ((__ffs(d3) - 1) >> 3);
                            }
                        } else {
                            while (g < 16 && mpos + o + g < nq && inb[mpos + o + g] == inb[mcand + o + g]) g++;
                        }
                        good[q] = g;
                    }
                    bool done = false;
    #pragma unroll
                    for (int q = 0; q < 4; q++) {
                        if (done) break;
                        const uint64_t partial = __ballot(good[q] != 16);
                        if (partial) {
                            const uint32_t fl = __ffsll((unsigned long long)partial) - 1;
                            ml += fl * 16 + __shfl(good[q], fl);
                            done = true;
                        } else {
                            ml += 1024;
                        }
                    }
                    if (done) break;
                }
                    }
                    const uint32_t ll = mpos - anchor;
                    if (emitted < mpos) {  // literals of this sequence not written yet: the part inside the window
                        if (emitted < base) {
                            wave_copy(lits + lit_total + (emitted - anchor), inb + emitted, base - emitted, lane);
                            emitted = base;
                        }
                        if (pos >= emitted && pos < mpos) lits[lit_total + (pos - anchor)] = (uint8_t)v;
                    }
                    if (lane == 0) {
                        seqs[3 * nseq] = ll;
                        seqs[3 * nseq + 1] = ml - 3;
                        seqs[3 * nseq + 2] = mpos - mcand;
                    }
                    lit_total += ll;
                    nseq++;
                    anchor = mpos + ml;
                    emitted = anchor;
                }
                store_tail();
                base = anchor > base + 64 ? anchor : base + 64;
            }
        } else {
            while (base < scan_end && nseq < max_seq) {
                const uint32_t pos = base + lane;
                uint32_t cand = 0, hitf = 0;
                if (pos < scan_end) {
                    const uint32_t v = ld32(inb + pos);
                    const uint32_t h = hash4<HASH_LOG>(v);
                    const uint32_t e = S.table[h];
                    S.table[h] = (uint16_t)pos;  // low 16 bits; candidates are within 64 KiB
                    if (e != 0xFFFF) {
                        uint32_t c = (pos & ~0xFFFFu) | e;
                        if (c >= pos) c -= 0x10000u;  // wraps to a huge value when there is no earlier half
                        if (c < pos && pos >= PRE && ld32(inb + c) == v) { cand = c; hitf = 1; }
                    }
                }
                const uint64_t hit = __ballot(hitf != 0);
                if (!hit) {
                    misses++;
                    base += 64 * (1 + (misses >> 4 > 7 ? 7 : misses >> 4));  // accelerate through incompressible data
                    continue;
                }
                misses = 0;
                const uint32_t win = __ffsll((unsigned long long)hit) - 1;
                const uint32_t mpos = __shfl(pos, win), mcand = __shfl(cand, win);
                // cooperative extension: 16 bytes per lane per piece, 4 pieces (4 KiB) in flight per step
                uint32_t ml = 4;
                for (;;) {
                    uint32_t good[4];
    #pragma unroll
                    for (int q = 0; q < 4; q++) {
                        const uint32_t o = ml + q * 1024 + lane * 16;
                        uint32_t g = 0;  // matching bytes in my 16-byte piece
                        if (mpos + o + 16 <= nq) {
                            uint4 x = ld128(inb + mpos + o), y = ld128(inb + mcand + o);
                            const uint32_t d0 = x.x ^ y.x, d1 = x.y ^ y.y, d2 = x.z ^ y.z, d3 = x.w ^ y.w;
                            g = 16;
                            if (d0 | d1 | d2 | d3) {  // rare: only the piece where the match ends
                                if (d0) g = (__ffs(d0) - 1) >> 3;
                                else if (d1) g = 4 + ((__ffs(d1) - 1) >> 3);
                                else if (d2) g = 8 + ((__ffs(d2) - 1) >> 3);
                                else g = 12 + ((__ffs(d3) - 1) >> 3);
                            }
                        } else {
                            while (g < 16 && mpos + o + g < nq && inb[mpos + o + g] == inb[mcand + o + g]) g++;
                        }
                        good[q] = g;
                    }
                    bool done = false;
    #pragma unroll
                    for (int q = 0; q < 4; q++) {
                        if (done) break;
                        const uint64_t partial = __ballot(good[q] != 16);
                        if (partial) {
                            const uint32_t fl = __ffsll((unsigned long long)partial) - 1;
                            ml += fl * 16 + __shfl(good[q], fl);
                            done = true;
                        } else {
                            ml += 1024;
                        }
                    }
                    if (done) break;
                }
                // emit: literals [anchor, mpos) then the match
                const uint32_t ll = mpos - anchor;
                wave_copy(lits + lit_total, inb + anchor, ll, lane);
                if (lane == 0) {
                    seqs[3 * nseq] = ll;
                    seqs[3 * nseq + 1] = ml - 3;
                    seqs[3 * nseq + 2] = mpos - mcand;
                }
                lit_total += ll;
                nseq++;
                anchor = mpos + ml;
                base = anchor;
            }
        }
        if (HASH_LOG == 11 && a.retry_list &&
            ((nseq >= max_seq && base < scan_end) || (a.high && (nseq > 8 || lit_total + (nq - anchor) >= HUF_MIN_LITS)))) {
            // Sequence budget spent before the end of the block: this is not periodic data.  Hand the block to the
            // wide variant (denser matcher, Huffman literals, parallel bitstream) instead of emitting the rest raw.
            // The higher effort tier hands over everything that is not the periodic shape this variant is for (a handful
            // of sequences, few literals): on real text two thirds of the rounds are below 16 KiB and never spend the
            // budget — one match per window, raw literals — and cost 7 % of the corpus against libzstd -1
            // (tools/ratio_by_size.py: rounds of 4-16 KiB at 0.57 here, 0.27 there).
            if (lane == 0) a.retry_list[atomicAdd(a.retry_count, 1u)] = item_id;
            continue;
        }
        // trailing literals (the wide variant has written most of them already)
        {
            const uint32_t e0 = (HASH_LOG == 13 && emitted > anchor) ? emitted : anchor;
            if (e0 < nq && !gave_up) wave_copy(lits + lit_total + (e0 - anchor), inb + e0, nq - e0, lane);  // (a block given up on goes out raw, straight from the input)
        }
        lit_total += nq - anchor;
        __syncthreads();  // sequences + literal bytes of all lanes are visible to lane 0

        ESTAMP(2);
        // ---- entropy stage: literals (Huffman in the wide variant when it pays), headers, sequences bitstream ----
        uint32_t lit_sec = 0;  // bytes of the literals section at blk + 3
        if (HIGH && lit_total >= HUF_MIN_LITS && !gave_up)
            lit_sec = huf_literals_any(S, lits, lit_total, lits + ((lit_total + 3) & ~3u), blk + 3, lane);
        else if (HASH_LOG == 13 && lit_total >= HUF_MIN_LITS && !gave_up)
            lit_sec = huf_literals(S, lits, lit_total, lits + ((lit_total + 3) & ~3u), blk + 3, lane);
        ESTAMP(3);
        bool raw = nseq == 0 && lit_sec == 0;
        uint32_t csize = 0;
        if (!raw) {
            if (!lit_sec) {
                if (lane == 0) {
                    // literals header: Raw_Literals_Block, size_format 3 (20-bit size, 3 bytes)
                    blk[3] = (uint8_t)(0 | (3 << 2) | ((lit_total & 15) << 4));
                    blk[4] = (uint8_t)(lit_total >> 4);
                    blk[5] = (uint8_t)(lit_total >> 12);
                }
                lit_sec = 3 + lit_total;
            }
            uint8_t *p = blk + 3 + lit_sec;
            const uint32_t hl = nseq < 128 ? 1u : (nseq < 0x7F00 ? 2u : 3u);
            if (lane == 0) {
                if (nseq < 128) p[0] = (uint8_t)nseq;
                else if (nseq < 0x7F00) { p[0] = (uint8_t)((nseq >> 8) + 128); p[1] = (uint8_t)nseq; }
                else { p[0] = 255; p[1] = (uint8_t)(nseq - 0x7F00); p[2] = (uint8_t)((nseq - 0x7F00) >> 8); }
                if (nseq) p[hl] = 0;  // LL, OF, ML all Predefined_Mode
            }
            if (nseq == 0) p += 1;  // a block of literals only: the sequences section is the single count byte
            else
            // wide variant (128 KiB blocks, thousands of sequences): wave-parallel bitstream; small variant: serial writer
            if (HIGH && nseq >= CUSTOM_FSE_MIN_SEQ) p = encode_sequences_custom(S, seqs, nseq, p + hl, lane);  // writes the modes byte itself
            else if (HASH_LOG == 13 && nseq > 8) p = encode_sequences(S, seqs, nseq, p + hl + 1, lane, HIGH ? 0u : 3u);
            else {
                // a handful of sequences (periodic data: one or two per block): the serial writer costs fewer
                // issue slots than a wave-wide pass, and those slots belong to the hash kernel running alongside
                uint32_t plen = 0;
                if (lane == 0) {
                    BitW b{p + hl + 1, 0, 0};
                    CState sl, so, sm;
                    for (int32_t i = (int32_t)nseq - 1; i >= 0; i--) {
                        const uint32_t ll = seqs[3 * i], mlb = seqs[3 * i + 1], ov = seqs[3 * i + 2] + (HIGH ? 0u : 3u);
                        const uint32_t lc = ll < 64 ? c_ll_code[ll] : (uint32_t)hib(ll) + 19;
                        const uint32_t mc = mlb < 128 ? c_ml_code[mlb] : (uint32_t)hib(mlb) + 36;
                        const uint32_t oc = (uint32_t)hib(ov);
                        if (i == (int32_t)nseq - 1) {
                            sm.init(S.tabs.ml_state, S.tabs.ml_tt, 6, mc);
                            so.init(S.tabs.of_state, S.tabs.of_tt, 5, oc);
                            sl.init(S.tabs.ll_state, S.tabs.ll_tt, 6, lc);
                        } else {
                            so.encode(b, oc);
                            sm.encode(b, mc);
                            sl.encode(b, lc);
                        }
                        b.add(ll - c_ll_base_e[lc], c_ll_bits_e[lc]);
                        b.add(mlb + 3 - c_ml_base_e[mc], c_ml_bits_e[mc]);
                        b.add(ov - (1u << oc), oc);
                    }
                    sm.flush(b);
                    so.flush(b);
                    sl.flush(b);
                    plen = (uint32_t)(b.close() - p);
                }
                p += suni(plen);
            }
            csize = (uint32_t)(p - (blk + 3));
            if (csize >= n) raw = true;
        }
        ESTAMP(4);
        // Higher effort tier, frames of several blocks: the frame ends with an EMPTY raw block (3 bytes, legal anywhere in
        // a frame) behind the last real one.  It tells this build's block scan that every block of the frame stands on its
        // own (k_block_scan: a first block with per-block entropy tables looks like any foreign frame otherwise, and a
        // foreign frame's blocks are not worth trying one by one).  To every other decoder it is a block of no bytes.
        const bool tail_mark = a.tail_mark && last && it.n_blocks > 1;  // (a short last block is encoded by the small variant)
        if (lane == 0) {
            uint32_t piece_len = 0, hdr_len = 0;
            const uint32_t last_bit = last && !tail_mark ? 1u : 0u;
            if (!raw) {
                const uint32_t bh = last_bit | (2u << 1) | (csize << 3);
                blk[0] = (uint8_t)bh; blk[1] = (uint8_t)(bh >> 8); blk[2] = (uint8_t)(bh >> 16);
                piece_len = 3 + csize;
            }
            if (raw) {
                const uint32_t bh = last_bit | (0u << 1) | (n << 3);
                blk[0] = (uint8_t)bh; blk[1] = (uint8_t)(bh >> 8); blk[2] = (uint8_t)(bh >> 16);
                piece_len = 3 + n;
            }
            if (tail_mark) piece_len += 3;
            if (it.block == 0) {  // frame header in front of block 0: magic, FHD (single segment), FCS
                uint8_t h[16];
                uint32_t k = 0;
                h[k++] = 0x28; h[k++] = 0xB5; h[k++] = 0x2F; h[k++] = 0xFD;
                if (rlen < 256) { h[k++] = 0x20; h[k++] = (uint8_t)rlen; }
                else if (rlen < 65792) { h[k++] = 0x60; uint32_t v = (uint32_t)rlen - 256; h[k++] = (uint8_t)v; h[k++] = (uint8_t)(v >> 8); }
                else if (rlen <= 0xFFFFFFFFull) { h[k++] = 0xA0; for (int i = 0; i < 4; i++) h[k++] = (uint8_t)(rlen >> (8 * i)); }
                else { h[k++] = 0xE0; for (int i = 0; i < 8; i++) h[k++] = (uint8_t)(rlen >> (8 * i)); }
                hdr_len = k;
                for (uint32_t i = 0; i < k; i++) blk[(int32_t)i - (int32_t)k] = h[i];
            }
            s_raw = raw ? 1u : 0u;
            a.piece_len[item_id] = piece_len + hdr_len;
            a.piece_start[item_id] = it.prov + HDR_ROOM - hdr_len;
        }
        __syncthreads();
        if (s_raw) wave_copy(blk + 3, in, n, lane);  // raw block: the input bytes after the header
        if (tail_mark && lane == 0) {
            uint8_t *const t = blk + 3 + (s_raw ? n : csize);
            t[0] = 1; t[1] = 0; t[2] = 0;  // Last_Block, Raw_Block, Block_Size 0
        }
    }
}

// ---- piece scan + gather ---------------------------------------------------------------------
// local exclusive scan of piece lengths inside blocks of 256 pieces + block totals (small workgroups: the
// scan and the gather run next to the hash kernel of the auxiliary stream and must fit into the wave slots
// it frees one workgroup at a time)
__global__ __launch_bounds__(256) void k_piece_scan(const uint32_t *piece_len, uint32_t n, uint64_t *local_excl,
                                                    uint64_t *block_tot) {
    __shared__ uint64_t wsum[4];
    const uint32_t i = blockIdx.x * 256 + threadIdx.x, lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    uint64_t v = i < n ? piece_len[i] : 0, inc = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        uint64_t y = __shfl_up(inc, d);
        if (lane >= (uint32_t)d) inc += y;
    }
    if (lane == 63) wsum[w] = inc;
    __syncthreads();
    uint64_t wbase = 0;
    for (uint32_t k = 0; k < w; k++) wbase += wsum[k];
    if (i < n) local_excl[i] = wbase + inc - v;
    if (threadIdx.x == 255) block_tot[blockIdx.x] = wbase + inc;
}

// one wave per piece (tables of big pieces: 128 KiB blocks, 64 KiB slices of stored rounds): copy it to its packed position;
// fill the per-round outputs
__global__ __launch_bounds__(256) void k_gather_wide(GatherArgs g) {
    const uint32_t lane = threadIdx.x & 63;
    const uint32_t piece = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (piece >= g.n_pieces) return;
    // base of my scan block = sum of the totals of the blocks before it
    const uint32_t sb = piece >> 8;
    uint64_t part = 0;
    for (uint32_t k = lane; k < sb; k += 64) part += g.block_tot[k];
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) part += __shfl_xor(part, d);
    const uint64_t off = part + g.local_excl[piece];
    const uint32_t len = g.piece_len[piece];
    const EncItem it = g.items[piece];
    const uint8_t *s = (it.flags & ITEM_SKIP) ? g.src + g.src_off[it.round] + it.prov : g.prov + g.piece_start[piece];
    if (g.stored && g.stored[it.round]) s = g.src + g.src_off[it.round] + (uint64_t)it.block * BLOCK_BYTES;  // raw bytes of this block
    uint8_t *d = g.blob_out + off;
    if (off + len <= g.blob_cap) {
        if (!(g.skip_stored_copy && (it.flags & ITEM_SKIP))) wave_copy(d, s, len, lane);
    } else if (lane == 0) atomicOr(g.overflow, 1u);
    if (lane == 0) {
        if (it.flags & ITEM_FIRST) g.blob_offset[it.round] = off;
        atomicAdd(reinterpret_cast<unsigned long long *>(&g.blob_size[it.round]), (unsigned long long)len);
        if (piece == g.n_pieces - 1) *g.total = off + len;
    }
}

// One wave per 64 consecutive pieces (a workgroup = the 256 pieces of one scan block): every lane does its piece's
// bookkeeping, pieces of <= GATHER_SMALL bytes are copied by their own lanes 16 bytes at a time, the others by the whole
// wave one after the other.  (First form: one wave per piece — 100,000 waves of five dependent round trips each for C2's
// 85-byte pieces: 0.055 ms of a 0.87 ms step.)
constexpr uint32_t GATHER_SMALL = 512;
__global__ __launch_bounds__(256) void k_gather(GatherArgs g) {
    const uint32_t lane = threadIdx.x & 63;
    const uint32_t piece = blockIdx.x * 256 + threadIdx.x;
    const bool on = piece < g.n_pieces;
    // base of this scan block = sum of the totals of the blocks before it
    uint64_t part = 0;
    for (uint32_t k = lane; k < blockIdx.x; k += 64) part += g.block_tot[k];
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) part += __shfl_xor(part, d);
    uint64_t off = 0;
    uint32_t len = 0;
    const uint8_t *s = g.src;
    bool copy = false;
    if (on) {
        off = part + g.local_excl[piece];
        len = g.piece_len[piece];
        const EncItem it = g.items[piece];
        s = (it.flags & ITEM_SKIP) ? g.src + g.src_off[it.round] + it.prov : g.prov + g.piece_start[piece];
        if (g.stored && g.stored[it.round]) s = g.src + g.src_off[it.round] + (uint64_t)it.block * BLOCK_BYTES;  // raw bytes of this block
        if (off + len <= g.blob_cap) copy = !(g.skip_stored_copy && (it.flags & ITEM_SKIP));
        else atomicOr(g.overflow, 1u);
        if (it.flags & ITEM_FIRST) g.blob_offset[it.round] = off;
        atomicAdd(reinterpret_cast<unsigned long long *>(&g.blob_size[it.round]), (unsigned long long)len);
        if (piece == g.n_pieces - 1) *g.total = off + len;
    }
    uint8_t *const d = g.blob_out + off;
    if (copy && len <= GATHER_SMALL) {  // lane = piece
        uint32_t o = 0;
        for (; o + 16 <= len; o += 16) {
            uint4 v;
            __builtin_memcpy(&v, s + o, 16);
            __builtin_memcpy(d + o, &v, 16);
        }
        for (; o < len; o++) d[o] = s[o];
    }
    // (wave-uniform) the longer pieces, the whole wave on each — four at a time: the first KiB of each of the four is
    // loaded (16 bytes per lane) before any of it is stored, so a step's round trips overlap (one piece per step, load
    // then store: 64 pieces of 1-2 KB, a table of small real-text rounds, were 64 dependent round trips per wave);
    // what a piece has beyond its first KiB follows piece by piece
    uint64_t bigm = __ballot(copy && len > GATHER_SMALL);
    while (bigm) {
        uint32_t jj[4], nn[4];
        uint64_t sj[4], dj[4];
        uint4 v[4];
#pragma unroll
        for (int q = 0; q < 4; q++) {
            nn[q] = 0; sj[q] = 0; dj[q] = 0; jj[q] = 0;
            if (bigm) {
                jj[q] = (uint32_t)__ffsll((long long)bigm) - 1;
                bigm &= bigm - 1;
                sj[q] = ((uint64_t)rdlane((uint32_t)((uint64_t)(uintptr_t)s >> 32), jj[q]) << 32) | rdlane((uint32_t)(uintptr_t)s, jj[q]);
                dj[q] = ((uint64_t)rdlane((uint32_t)((uint64_t)(uintptr_t)d >> 32), jj[q]) << 32) | rdlane((uint32_t)(uintptr_t)d, jj[q]);
                nn[q] = rdlane(len, jj[q]);
            }
            v[q] = make_uint4(0, 0, 0, 0);
            if (16 * lane + 16 <= nn[q]) __builtin_memcpy(&v[q], reinterpret_cast<const uint8_t *>((uintptr_t)sj[q]) + 16 * lane, 16);
        }
#pragma unroll
        for (int q = 0; q < 4; q++) {
            uint8_t *const dq = reinterpret_cast<uint8_t *>((uintptr_t)dj[q]);
            const uint8_t *const sq = reinterpret_cast<const uint8_t *>((uintptr_t)sj[q]);
            if (16 * lane + 16 <= nn[q]) __builtin_memcpy(dq + 16 * lane, &v[q], 16);
            const uint32_t first = nn[q] < 1024 ? nn[q] : 1024u, whole = first & ~15u;   // bytes of the first KiB moved as 16-byte pieces
            if (lane < first - whole) dq[whole + lane] = sq[whole + lane];
        }
#pragma unroll
        for (int q = 0; q < 4; q++)
            if (nn[q] > 1024)  // (wave-uniform)
                wave_copy(reinterpret_cast<uint8_t *>((uintptr_t)dj[q]) + 1024, reinterpret_cast<const uint8_t *>((uintptr_t)sj[q]) + 1024, nn[q] - 1024, lane);
    }
}

void launch_encode(const EncodeArgs &a, int grid, bool small_blocks, bool high, hipStream_t s) {
    if (!a.n_items) return;
    if (small_blocks) hipLaunchKernelGGL(k_zstd_encode<11>, dim3(grid), dim3(64), 0, s, a);
    else if (high) hipLaunchKernelGGL((k_zstd_encode<13, true>), dim3(grid), dim3(64), 0, s, a);
    else hipLaunchKernelGGL(k_zstd_encode<13>, dim3(grid), dim3(64), 0, s, a);
}

void launch_piece_scan(const uint32_t *piece_len, uint32_t n, uint64_t *local_excl, uint64_t *block_tot, hipStream_t s) {
    if (!n) return;
    hipLaunchKernelGGL(k_piece_scan, dim3((n + 255) / 256), dim3(256), 0, s, piece_len, n, local_excl, block_tot);
}

// Opt-in store-if-incompressible (reference wish list, TODO_NOW.md:L37-38): a round whose frame is not
// smaller than its input is emitted as-is (compressed=false) — one thread per round rewrites the lengths
// of the round's pieces to the raw block sizes before the scan.
__global__ __launch_bounds__(256) void k_store_decide(const uint32_t *first_item, const EncItem *items, const uint64_t *len,
                                                      const uint8_t *skip, uint32_t n_rounds, uint32_t *piece_len,
                                                      uint8_t *stored) {
    const uint32_t r = blockIdx.x * 256 + threadIdx.x;
    if (r >= n_rounds) return;
    uint8_t st = 0;
    const uint64_t L = len[r];
    if (!skip[r] && L > 0) {
        const uint32_t f = first_item[r], nb = items[f].n_blocks;
        uint64_t sum = 0;
        for (uint32_t k = 0; k < nb; k++) sum += piece_len[f + k];
        if (sum >= L) {
            st = 1;
            for (uint32_t k = 0; k < nb; k++) {
                const uint64_t o = (uint64_t)k * BLOCK_BYTES;
                piece_len[f + k] = (uint32_t)(L - o < BLOCK_BYTES ? L - o : BLOCK_BYTES);
            }
        }
    }
    stored[r] = st;
}

void launch_store_decide(const uint32_t *first_item, const EncItem *items, const uint64_t *len, const uint8_t *skip,
                         uint32_t n_rounds, uint32_t *piece_len, uint8_t *stored, hipStream_t s) {
    if (!n_rounds) return;
    hipLaunchKernelGGL(k_store_decide, dim3((n_rounds + 255) / 256), dim3(256), 0, s, first_item, items, len, skip, n_rounds,
                       piece_len, stored);
}

void launch_gather(const GatherArgs &g, hipStream_t s) {
    if (!g.n_pieces) return;
    if (g.small_pieces) hipLaunchKernelGGL(k_gather, dim3((g.n_pieces + 255) / 256), dim3(256), 0, s, g);
    else hipLaunchKernelGGL(k_gather_wide, dim3((g.n_pieces + 3) / 4), dim3(256), 0, s, g);
}

// ---- host: FSE encoding tables for the predefined distributions (RFC 8878 §3.1.1.3.2.2) ------
static void build_ctable(const int8_t *norm, int nsym, int log, uint16_t *state_table, FseSymTT *tt) {
    const int size = 1 << log;
    std::vector<int> cumul(nsym + 2, 0);
    std::vector<uint8_t> sym(size);
    int high = size - 1;
    for (int s = 0; s < nsym; s++) {
        int c = norm[s] == -1 ? 1 : norm[s];
        cumul[s + 1] = cumul[s] + c;
        if (norm[s] == -1) sym[high--] = (uint8_t)s;
    }
    const int step = (size >> 1) + (size >> 3) + 3, mask = size - 1;
    int pos = 0;
    for (int s = 0; s < nsym; s++)
        for (int i = 0; i < norm[s]; i++) {
            sym[pos] = (uint8_t)s;
            do { pos = (pos + step) & mask; } while (pos > high);
        }
    std::vector<int> cur(cumul.begin(), cumul.end());
    for (int u = 0; u < size; u++) state_table[cur[sym[u]]++] = (uint16_t)(size + u);
    int total = 0;
    for (int s = 0; s < nsym; s++) {
        int c = norm[s];
        if (c == 0) {
            tt[s].delta_nb_bits = ((uint32_t)(log + 1) << 16) - (1u << log);
            tt[s].delta_find_state = 0;
        } else if (c == -1 || c == 1) {
            tt[s].delta_nb_bits = ((uint32_t)log << 16) - (1u << log);
            tt[s].delta_find_state = total - 1;
            total++;
        } else {
            int hb = 31 - __builtin_clz((unsigned)(c - 1));
            uint32_t max_bits_out = (uint32_t)(log - hb);
            uint32_t min_state_plus = (uint32_t)c << max_bits_out;
            tt[s].delta_nb_bits = (max_bits_out << 16) - min_state_plus;
            tt[s].delta_find_state = total - c;
            total += c;
        }
    }
}

void build_encode_tables(EncTables *t) {
    static const int8_t ll[36] = {4, 3, 2, 2, 2, 2, 2, 2, 2, 2, 2, 2, 2, 1, 1, 1, 2, 2,
                                  2, 2, 2, 2, 2, 2, 2, 3, 2, 1, 1, 1, 1, 1, -1, -1, -1, -1};
    static const int8_t ml[53] = {1, 4, 3, 2, 2, 2, 2, 2, 2, 1, 1, 1, 1, 1, 1, 1, 1, 1,
                                  1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1,
                                  1, 1, 1, 1, 1, 1, 1, 1, 1, 1, -1, -1, -1, -1, -1, -1, -1};
    static const int8_t of[29] = {1, 1, 1, 1, 1, 1, 2, 2, 2, 1, 1, 1, 1, 1, 1,
                                  1, 1, 1, 1, 1, 1, 1, 1, 1, -1, -1, -1, -1, -1};
    memset(t, 0, sizeof *t);
    build_ctable(ll, 36, 6, t->ll_state, t->ll_tt);
    build_ctable(ml, 53, 6, t->ml_state, t->ml_tt);
    build_ctable(of, 29, 5, t->of_state, t->of_tt);
}

}  // namespace zn
