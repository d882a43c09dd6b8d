// BLAKE3 tile kernels for CDNA4 (gfx950): integer/byte work bounded by HBM bandwidth and VALU
// issue, no MFMA.  One wavefront = one Tile = up to 64 BLAKE3 leaves, lane = leaf; the parent
// tree is folded inside the wave with ds_bpermute shuffles (segmented, so several small
// chunks share a wave).  Store path: the leaf loop also copies the bytes it hashes
// (blob -> output), so hash + copy is a single pass over HBM.
#include "common.h"
#include "blake3_dev.h"

namespace zn {

__device__ __forceinline__ uint4 ld16(const uint8_t *p) {
    uint4 v;
    __builtin_memcpy(&v, p, 16);  // unaligned-access-mode: one global_load_dwordx4
    return v;
}
__device__ __forceinline__ void st16(uint8_t *p, uint4 v) { __builtin_memcpy(p, &v, 16); }

// Load one (possibly partial) 64-byte block into 16 little-endian words, zero padded.
__device__ __forceinline__ void load_block(const uint8_t *p, uint32_t n, uint32_t m[16]) {
    if (n == 64) {
        uint4 a = ld16(p), b = ld16(p + 16), c = ld16(p + 32), d = ld16(p + 48);
        m[0] = a.x; m[1] = a.y; m[2] = a.z; m[3] = a.w;
        m[4] = b.x; m[5] = b.y; m[6] = b.z; m[7] = b.w;
        m[8] = c.x; m[9] = c.y; m[10] = c.z; m[11] = c.w;
        m[12] = d.x; m[13] = d.y; m[14] = d.z; m[15] = d.w;
    } else {
#pragma unroll
        for (int i = 0; i < 16; i++) {
            uint32_t w = 0;
#pragma unroll
            for (int k = 0; k < 4; k++) {
                uint32_t idx = 4 * i + k;
                if (idx < n) w |= (uint32_t)p[idx] << (8 * k);
            }
            m[i] = w;
        }
    }
}

__device__ __forceinline__ void store_block(uint8_t *p, uint32_t n, const uint32_t m[16]) {
    if (n == 64) {
        st16(p, make_uint4(m[0], m[1], m[2], m[3]));
        st16(p + 16, make_uint4(m[4], m[5], m[6], m[7]));
        st16(p + 32, make_uint4(m[8], m[9], m[10], m[11]));
        st16(p + 48, make_uint4(m[12], m[13], m[14], m[15]));
    } else {
        for (uint32_t i = 0; i < n; i++) p[i] = (uint8_t)(m[i >> 2] >> (8 * (i & 3)));
    }
}

// Fold the CVs held by lanes [s, s+n) (one node per lane, node j in lane s+j) into lane s.
// All 64 lanes call this together; lanes outside any segment pass n = 0.  `final_root`:
// the fold ends at the unit's root (ROOT flag on the last parent).
__device__ __forceinline__ void fold_segments(uint32_t cv[8], uint32_t s, uint32_t n, bool final_root) {
    const uint32_t lane = threadIdx.x & 63;
    const uint32_t j = lane - s;
    while (__ballot(n > 1) != 0ull) {
        uint32_t li = (s + 2 * j) & 63, ri = (s + 2 * j + 1) & 63;
        uint32_t L[8], R[8];
#pragma unroll
        for (int i = 0; i < 8; i++) {
            L[i] = __shfl(cv[i], li);
            R[i] = __shfl(cv[i], ri);
        }
        if (n > 1 && j < (n + 1) / 2) {
            if (2 * j + 1 < n) b3::parent(cv, L, R, final_root && n == 2);
            else {
#pragma unroll
                for (int i = 0; i < 8; i++) cv[i] = L[i];
            }
        }
        if (n > 1) n = (n + 1) / 2;
    }
}

template <bool COPY>
__global__ __launch_bounds__(256) void k_hash_tiles(HashArgs a) {
    const uint32_t lane = threadIdx.x & 63;
    const uint32_t wave = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (wave >= a.n_tiles) return;
    const Tile t = a.tiles[wave];

    uint32_t unit, k, unit_leaves, seg_start;
    bool active = lane < t.n_leaves;
    if (t.n_units) {
        uint32_t cnt = 0;
        if (lane < t.n_units) {
            uint64_t L = a.len[t.first_unit + lane];
            cnt = L ? (uint32_t)((L + 1023) >> 10) : 1u;
        }
        uint32_t inc = cnt;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            uint32_t y = __shfl_up(inc, d);
            if (lane >= (uint32_t)d) inc += y;
        }
        // smallest i with inc[i] > lane
        uint32_t lo = 0, hi = t.n_units - 1;
#pragma unroll
        for (int it = 0; it < 6; it++) {
            uint32_t mid = (lo + hi) >> 1;
            uint32_t v = __shfl(inc, mid);
            if (lo < hi) {
                if (v > lane) hi = mid; else lo = mid + 1;
            }
        }
        uint32_t i = lo;
        unit_leaves = __shfl(cnt, i);
        seg_start = __shfl(inc, i) - unit_leaves;
        unit = t.first_unit + i;
        k = lane - seg_start;
    } else {
        unit = t.first_unit;
        k = t.first_leaf + lane;
        unit_leaves = t.n_leaves;  // nodes of this slice
        seg_start = 0;
    }
    if (!active) { unit = t.first_unit; k = 0; }
    if (a.status && a.status[unit] < 0) active = false;

    const uint64_t ulen = a.len[unit];
    const bool from_b = a.sel && a.sel[unit];
    const uint8_t *src = from_b ? a.srcB + a.offB[unit] : a.srcA + (a.offA[unit] - a.baseA);
    uint8_t *dst = (COPY && !from_b && a.srcB) ? a.srcB + a.offB[unit] : nullptr;
    const uint64_t leaf_off = (uint64_t)k << 10;
    uint32_t leaf_len = 0;
    if (active && ulen > leaf_off) leaf_len = (uint32_t)((ulen - leaf_off) < 1024 ? (ulen - leaf_off) : 1024);
    const uint32_t nblk = leaf_len ? (leaf_len + 63) >> 6 : 1u;
    const uint64_t total_leaves = ulen ? (ulen + 1023) >> 10 : 1;
    const bool single = total_leaves == 1;  // the leaf itself is the root

    uint32_t cv[8];
    b3::set_iv(cv);
    src += leaf_off;
    if (COPY && dst) dst += leaf_off;
    // wave-uniform trip count (16 for every full leaf)
    uint32_t maxblk = active ? nblk : 0;
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
        uint32_t o = __shfl_xor(maxblk, d);
        maxblk = o > maxblk ? o : maxblk;
    }
    maxblk = __builtin_amdgcn_readfirstlane(maxblk);
    for (uint32_t b = 0; b < maxblk; b++) {
        if (active && b < nblk) {
            uint32_t m[16];
            uint32_t rem = leaf_len - b * 64;
            uint32_t bl = leaf_len == 0 ? 0 : (rem < 64 ? rem : 64);
            load_block(src + b * 64, bl, m);
            if (COPY && dst) store_block(dst + b * 64, bl, m);
            uint32_t flags = (b == 0 ? b3::CHUNK_START : 0u) |
                             (b == nblk - 1 ? (b3::CHUNK_END | (single ? b3::ROOT : 0u)) : 0u);
            b3::compress(cv, m, k, 0, bl, flags);
        }
    }

    fold_segments(cv, seg_start, active ? unit_leaves : 0, t.n_units != 0);

    if (active && lane == seg_start) {
        uint32_t *o = t.n_units ? a.digests + (size_t)unit * 8 : a.tile_cv + (size_t)t.cv_index * 8;
#pragma unroll
        for (int i = 0; i < 8; i++) o[i] = cv[i];
    }
}

void launch_hash_tiles(const HashArgs &a, hipStream_t s) {
    if (!a.n_tiles) return;
    dim3 grid((a.n_tiles + 3) / 4), block(256);
    if (a.copy_to_B) hipLaunchKernelGGL(k_hash_tiles<true>, grid, block, 0, s, a);
    else hipLaunchKernelGGL(k_hash_tiles<false>, grid, block, 0, s, a);
}

// Finish units with more than 64 leaves: one wave folds the unit's tile CVs (each the root of
// a complete 64-leaf subtree, the last possibly partial) level by level, in place, then the
// last <= 64 nodes in registers.
__global__ __launch_bounds__(64) void k_merge_big(const BigUnit *big, uint32_t n_big, uint32_t *tile_cv,
                                                 uint32_t *digests) {
    const uint32_t lane = threadIdx.x;
    if (blockIdx.x >= n_big) return;
    const BigUnit u = big[blockIdx.x];
    uint32_t *base = tile_cv + (size_t)u.cv_base * 8;
    uint32_t m = u.n_cvs;
    while (m > 64) {
        uint32_t half = (m + 1) / 2;
        for (uint32_t p = 0; p * 64 < half; p++) {
            uint32_t j = p * 64 + lane;
            uint32_t out[8];
            bool on = j < half;
            if (on) {
                uint32_t L[8], R[8];
#pragma unroll
                for (int i = 0; i < 8; i++) L[i] = base[(size_t)(2 * j) * 8 + i];
                if (2 * j + 1 < m) {
#pragma unroll
                    for (int i = 0; i < 8; i++) R[i] = base[(size_t)(2 * j + 1) * 8 + i];
                    b3::parent(out, L, R, false);
                } else {
#pragma unroll
                    for (int i = 0; i < 8; i++) out[i] = L[i];
                }
            }
            __syncthreads();  // all reads of this pass done before its in-place writes
            if (on) {
#pragma unroll
                for (int i = 0; i < 8; i++) base[(size_t)j * 8 + i] = out[i];
            }
        }
        __syncthreads();  // level complete and visible before the next level reads it
        m = half;
    }
    uint32_t cv[8];
#pragma unroll
    for (int i = 0; i < 8; i++) cv[i] = lane < m ? base[(size_t)lane * 8 + i] : 0u;
    fold_segments(cv, 0, lane < m ? m : 0, true);
    if (lane == 0) {
#pragma unroll
        for (int i = 0; i < 8; i++) digests[(size_t)u.unit * 8 + i] = cv[i];
    }
}

void launch_merge_big(const BigUnit *big, uint32_t n_big, uint32_t *tile_cv, uint32_t *digests, hipStream_t s) {
    if (!n_big) return;
    hipLaunchKernelGGL(k_merge_big, dim3(n_big), dim3(64), 0, s, big, n_big, tile_cv, digests);
}

// Compare computed digests with the index's checksum column and accumulate the read loop's
// counters (decompress.rs:L140,L169-185): counters[0..5] = total_chunks, total_written_bytes,
// verified_bytes, corrupt_bytes, corrupt_rows, decode_errors.
__global__ __launch_bounds__(256) void k_verify(const uint32_t *digests, const uint8_t *checksum,
                                                const uint64_t *usize, const int32_t *status, uint32_t n_rows,
                                                uint64_t row_begin, unsigned long long *counters,
                                                uint64_t *corrupt_rows, uint32_t corrupt_cap) {
    const uint32_t r = blockIdx.x * 256 + threadIdx.x;
    unsigned long long v[6] = {0, 0, 0, 0, 0, 0};
    if (r < n_rows) {
        v[0] = 1;
        if (status && status[r] < 0) {
            v[5] = 1;
        } else {
            const unsigned long long len = usize[r];
            v[1] = len;
            bool ok = true;
            if (checksum) {
                const uint32_t *want = reinterpret_cast<const uint32_t *>(checksum) + (size_t)r * 8;
#pragma unroll
                for (int i = 0; i < 8; i++) ok = ok && (want[i] == digests[(size_t)r * 8 + i]);
            }
            if (ok) v[2] = len;
            else {
                v[3] = len;
                v[4] = 1;
                unsigned long long slot = atomicAdd(&counters[6], 1ull);
                if (corrupt_rows && slot < corrupt_cap) corrupt_rows[slot] = row_begin + r;
            }
        }
    }
    __shared__ unsigned long long red[6][4];
#pragma unroll
    for (int c = 0; c < 6; c++) {
        unsigned long long x = v[c];
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) x += __shfl_xor(x, d);
        if ((threadIdx.x & 63) == 0) red[c][threadIdx.x >> 6] = x;
    }
    __syncthreads();
    if (threadIdx.x < 6) {
        unsigned long long x = red[threadIdx.x][0] + red[threadIdx.x][1] + red[threadIdx.x][2] + red[threadIdx.x][3];
        if (x) atomicAdd(&counters[threadIdx.x], x);
    }
}

void launch_verify(const uint32_t *digests, const uint8_t *checksum, const uint64_t *usize,
                   const int32_t *status, uint32_t n_rows, uint64_t row_begin, uint64_t *counters,
                   uint64_t *corrupt_rows, uint32_t corrupt_cap, hipStream_t s) {
    if (!n_rows) return;
    hipLaunchKernelGGL(k_verify, dim3((n_rows + 255) / 256), dim3(256), 0, s, digests, checksum, usize, status,
                       n_rows, row_begin, reinterpret_cast<unsigned long long *>(counters), corrupt_rows, corrupt_cap);
}

}  // namespace zn
