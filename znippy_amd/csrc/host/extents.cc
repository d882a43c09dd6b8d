// One vectorised pass over a row table's columns (host): the extents a run is validated against and the sums that size
// the batch path's pools — branch-free reductions, so the compiler's AVX-512 / AVX2 clones do 8 / 4 rows per step
// (the scalar loop was 0.18 ms per 100k rows: a third of znippy_rows_create).  All rows compressed (the caller checks).
#include <stddef.h>
#include <stdint.h>

extern "C" __attribute__((target_clones("avx512f", "avx2", "default")))
void zn_rows_extents(const uint64_t *bo, const uint64_t *bs, const uint64_t *oo, const uint64_t *us, size_t n, uint64_t out[8]) {
    uint64_t min_bo = ~0ull, max_bend = 0, max_oend = 0, wrap = 0, sum_us = 0, nblk = 0, n_big = 0, odd = 0;
    for (size_t i = 0; i < n; i++) {
        const uint64_t b = bo[i], be = b + bs[i], o = oo[i], u = us[i], oe = o + u;
        min_bo = b < min_bo ? b : min_bo;
        max_bend = be > max_bend ? be : max_bend;
        max_oend = oe > max_oend ? oe : max_oend;
        wrap |= (uint64_t)(be < b) | (uint64_t)(oe < o);
        sum_us += u;
        nblk += ((u + 131071) >> 17) + (uint64_t)(u == 0);   // 128 KiB blocks, one for an empty row
        n_big += (uint64_t)(u > 65536);
        odd |= o & 15;
    }
    out[0] = min_bo; out[1] = max_bend; out[2] = max_oend; out[3] = wrap; out[4] = sum_us; out[5] = nblk; out[6] = n_big; out[7] = odd;
}

// Round tables: the totals that size a table's device arrays, one branch-free pass (the arrays themselves are filled on the
// device: k_rounds_scan / k_rounds_fill in api.hip, same arithmetic).  slot(n) = enc_slot_bytes(n) (encode.h).
// out: [0] items, [1] provisional bytes, [2] small blocks, [3] wide blocks, [4] blob bound, [5] input bytes,
//      [6] bytes through the encoder, [7] some round but the last has a length that is not a multiple of 16
template <bool HAS_SKIP>
static inline __attribute__((always_inline)) void rounds_totals_body(const uint64_t *len, const uint8_t *skip, size_t n, uint64_t out[8]) {
    uint64_t items = 0, prov = 0, small = 0, wide = 0, bound = 0, in = 0, enc = 0, odd = 0;
    constexpr uint64_t slot_full = ((16 + 16 + 2 * 131072ull + (131072ull >> 2) + 64) + 15) & ~15ull;
    for (size_t i = 0; i < n; i++) {
        const uint64_t L = len[i];
        const uint64_t m_sk = HAS_SKIP ? (uint64_t)0 - (uint64_t)(skip[i] != 0) : 0, m_e = ~m_sk;  // all-ones masks (no 64-bit multiplies: AVX-512F has none)
        const uint64_t z = (uint64_t)(L == 0);
        const uint64_t nb = ((L + 131071) >> 17) + z, tail = L - ((nb - 1) << 17);
        const uint64_t np = ((L + 65535) >> 16) + z;
        const uint64_t slot_tail = ((16 + 16 + 2 * tail + (tail >> 2) + 64) + 15) & ~15ull;
        const uint64_t tail_wide = (uint64_t)(tail > 16 * 1024);
        const uint64_t full = nb - 1;  // full 128 KiB blocks: slot_full each (slot_full = 295,008 = 2^18 + 2^15 + 96: shifts and adds)
        const uint64_t full_slots = (full << 18) + (full << 15) + (full << 6) + (full << 5);
        items += (m_sk & np) | (m_e & nb);
        prov += m_e & (full_slots + slot_tail);
        wide += m_e & (full + tail_wide);
        small += m_e & (1 - tail_wide);
        const uint64_t hdrs = (L >> 17) + 1;
        bound += L + (m_e & (hdrs + hdrs + hdrs + 19));
        in += L;
        enc += m_e & L;
        odd |= (uint64_t)(i + 1 < n) & (uint64_t)((L & 15) != 0);
    }
    static_assert(slot_full == (1ull << 18) + (1ull << 15) + 96, "enc_slot_bytes(128 KiB) changed: update the shifts above");
    out[0] = items; out[1] = prov; out[2] = small; out[3] = wide; out[4] = bound; out[5] = in; out[6] = enc; out[7] = odd;
}
extern "C" __attribute__((target_clones("avx512f", "avx2", "default")))
void zn_rounds_totals(const uint64_t *len, const uint8_t *skip, size_t n, uint64_t out[8]) {
    if (skip) rounds_totals_body<true>(len, skip, n, out);
    else rounds_totals_body<false>(len, skip, n, out);
}

// A table written front to back (every blob behind the one before it, every row's bytes behind the one before it: what the
// reference's writer and reader produce) is its two size columns: the offset columns are their running sums.  One pass:
// checks that, narrows the sizes to 32 bits into dst (bs32[n] then us32[n]); returns 1 when the table is of that kind and
// every size fits 32 bits — the caller then sends 8 bytes per row to the device instead of 32.
extern "C" __attribute__((target_clones("avx512f", "avx2", "default")))
int zn_rows_pack32(const uint64_t *bo, const uint64_t *bs, const uint64_t *oo, const uint64_t *us, size_t n, uint32_t *dst) {
    if (n == 0) return 0;
    uint64_t bad = 0;
    uint32_t *const d_bs = dst, *const d_us = dst + n;
    for (size_t i = 0; i + 1 < n; i++) {
        bad |= (bo[i] + bs[i]) ^ bo[i + 1];
        bad |= (oo[i] + us[i]) ^ oo[i + 1];
        bad |= (bs[i] | us[i]) >> 32;
        d_bs[i] = (uint32_t)bs[i];
        d_us[i] = (uint32_t)us[i];
    }
    bad |= (bs[n - 1] | us[n - 1]) >> 32;
    d_bs[n - 1] = (uint32_t)bs[n - 1];
    d_us[n - 1] = (uint32_t)us[n - 1];
    return bad == 0;
}
