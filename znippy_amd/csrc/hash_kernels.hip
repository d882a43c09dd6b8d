// BLAKE3 tile kernels for CDNA4 (gfx950): integer/byte work bounded by HBM bandwidth and VALU
// issue, no MFMA.  One wavefront = one Tile = up to 64 BLAKE3 leaves, lane = leaf; the parent
// tree is folded inside the wave with ds_bpermute shuffles (segmented, so several small
// chunks share a wave).  Store path: the leaf loop also copies the bytes it hashes
// (blob -> output), so hash + copy is a single pass over HBM.
#include "common.h"
#include "hash_dev.h"

namespace zn {

// Workgroup-level fold of the store path (k_hash_tiles<COPY, 2>), out of line: their registers are their own.  The node
// arrays are LDS and said to be (address space 3: ds_read / ds_write, not FLAT instructions through a generic pointer).
typedef __attribute__((address_space(3))) uint32_t lds_u32;
// First level of a wave's 64 * G leaf nodes (G whole slices), in place: lane l = parent of nodes 2l, 2l + 1.
template <int G>
__device__ __noinline__ void fold_first_level(lds_u32 *nodes) {
    const uint32_t lane = threadIdx.x & 63;
    const bool on = lane < 32 * G;
    const lds_u4a *c = reinterpret_cast<const lds_u4a *>(nodes + (size_t)(2 * lane) * 8);
    u4v l0 = u4v{0, 0, 0, 0}, l1 = l0, r0 = l0, r1 = l0;
    if (on) { l0 = c[0]; l1 = c[1]; r0 = c[2]; r1 = c[3]; }
    uint32_t L[8] = {l0.x, l0.y, l0.z, l0.w, l1.x, l1.y, l1.z, l1.w}, R[8] = {r0.x, r0.y, r0.z, r0.w, r1.x, r1.y, r1.z, r1.w}, cv[8];
    b3::parent(cv, L, R, false);
    __builtin_amdgcn_wave_barrier();  // every lane has read its children (one wave: LDS operations in program order)
    if (on) {
        lds_u4a *d = reinterpret_cast<lds_u4a *>(nodes + (size_t)lane * 8);
        d[0] = u4v{cv[0], cv[1], cv[2], cv[3]};
        d[1] = u4v{cv[4], cv[5], cv[6], cv[7]};
    }
}
// Levels two to six of the workgroup's NT = 4 G tiles (4 or 8): tile T = G * wave + g keeps its 32 first-level nodes at
// base + (wave * area_nodes + 32 g) nodes; the levels' outputs go to parts of the areas nobody uses any more (G = 1: all
// behind area 0's 2 KiB of leaf nodes; G = 2: 128 nodes behind area 0's 4 KiB, the rest behind area 1's); cvi on lane T =
// where T's CV goes.
template <int G>
__device__ __noinline__ void fold_wg_tiles(lds_u32 *base, uint32_t area_nodes, uint32_t *tile_cv, uint32_t cvi) {
    constexpr uint32_t NT = 4 * G;
    const uint32_t lane = threadIdx.x & 63;
    lds_u32 *const C2 = base + (size_t)(64 * G) * 8;                                         // NT x 16 nodes
    lds_u32 *const C3 = G == 1 ? C2 + 64 * 8 : base + (size_t)(area_nodes + 128) * 8;        // NT x 8
    lds_u32 *const C4 = C3 + NT * 8 * 8, *const C5 = C4 + NT * 4 * 8;                        // NT x 4, NT x 2
    auto parent_of = [&](const lds_u32 *src, bool on, uint32_t cv[8]) {  // parent(src[0], src[1]) on the lanes that have one
        uint32_t L[8], R[8];
        const lds_u4a *c = reinterpret_cast<const lds_u4a *>(src);
        u4v l0 = u4v{0, 0, 0, 0}, l1 = l0, r0 = l0, r1 = l0;
        if (on) { l0 = c[0]; l1 = c[1]; r0 = c[2]; r1 = c[3]; }
        L[0] = l0.x; L[1] = l0.y; L[2] = l0.z; L[3] = l0.w; L[4] = l1.x; L[5] = l1.y; L[6] = l1.z; L[7] = l1.w;
        R[0] = r0.x; R[1] = r0.y; R[2] = r0.z; R[3] = r0.w; R[4] = r1.x; R[5] = r1.y; R[6] = r1.z; R[7] = r1.w;
        b3::parent(cv, L, R, false);
    };
    auto level = [&](const lds_u32 *src, lds_u32 *dst, bool on) {
        uint32_t cv[8];
        parent_of(src, on, cv);
        if (on) {
            lds_u4a *d = reinterpret_cast<lds_u4a *>(dst);
            d[0] = u4v{cv[0], cv[1], cv[2], cv[3]};
            d[1] = u4v{cv[4], cv[5], cv[6], cv[7]};
        }
        __builtin_amdgcn_wave_barrier();
    };
#pragma unroll 1
    for (uint32_t p = 0; p < NT * 16 / 64; p++) {  // NT tiles x 16 parents
        const uint32_t T = 4 * p + (lane >> 4), j = lane & 15;
        level(base + ((size_t)(T / G) * area_nodes + 32 * (T % G) + 2 * j) * 8, C2 + (size_t)(T * 16 + j) * 8, true);
    }
    {   // NT x 8, NT x 4, NT x 2: lane = (tile, pair) while there are that many
        const uint32_t k3 = lane % (NT * 8), k4 = lane % (NT * 4), k5 = lane % (NT * 2);
        level(C2 + (size_t)((k3 >> 3) * 16 + 2 * (k3 & 7)) * 8, C3 + (size_t)k3 * 8, lane < NT * 8);
        level(C3 + (size_t)((k4 >> 2) * 8 + 2 * (k4 & 3)) * 8, C4 + (size_t)k4 * 8, lane < NT * 4);
        level(C4 + (size_t)((k5 >> 1) * 4 + 2 * (k5 & 1)) * 8, C5 + (size_t)k5 * 8, lane < NT * 2);
    }
    uint32_t cv[8];
    parent_of(C5 + (size_t)((lane % NT) * 2) * 8, lane < NT, cv);  // the tiles' CVs
    if (lane < NT) {
        typedef __attribute__((address_space(1))) u4v glb_u4a;
        glb_u4a *o = (glb_u4a *)(uintptr_t)(tile_cv + (size_t)cvi * 8);
        o[0] = u4v{cv[0], cv[1], cv[2], cv[3]};
        o[1] = u4v{cv[4], cv[5], cv[6], cv[7]};
    }
}

// FOLD_G = tiles per wave whose parent levels are folded together (hash_dev.h, FoldQueue).  More tiles per wave
// means fewer fold passes but coarser work items; the launcher picks it from the tile count so that a
// small batch still spreads over every SIMD a few times.
// SHIFT (store path only): the variant that can re-cut the bytes for destinations that are not 16-byte aligned; it
// needs more registers and LDS than the plain one, so it is launched only when the host knows of such a destination.
template <bool COPY, int FOLD_G, bool SHIFT = false>
__global__ __launch_bounds__(256, COPY ? (FOLD_G == 1 && !SHIFT ? 5 : 4) : 1) void k_hash_tiles(HashArgs a) {  // store path: registers follow the LDS (32 / 40 KB per block)
    // Store path: a wave's LDS is [FOLD_G - 1 tiles of nodes][stage].  The stage is in use while a tile's leaves are hashed,
    // so the leaf CVs of the wave's earlier tiles wait in front of it; the last tile's go to the head of the stage (free by
    // then) and the node array the fold works on is one piece.
    constexpr uint32_t KEEP_BYTES = COPY ? (FOLD_G - 1) * 64 * 32 : 0;
    constexpr uint32_t AREA = COPY ? KEEP_BYTES + (SHIFT ? STAGE_SHIFT_BYTES : STAGE_SWZ_BYTES) : 16;
    static_assert(!COPY || FOLD_G * 64 * 32 <= (int)AREA, "the area must hold the queued tiles' nodes");
    __shared__ __attribute__((aligned(16))) uint32_t s_nodes[FOLD_G > 1 && !COPY ? 4 : 1][FOLD_G > 1 && !COPY ? FOLD_G * 64 * 8 : 4];
    __shared__ __attribute__((aligned(16))) uint8_t s_area[COPY ? 4 : 1][AREA];
    // WGF (store path): when each of the workgroup's four waves holds nothing but whole slices of big units (FOLD_G of them),
    // a wave folds only the first level of its own leaf nodes and ONE wave takes levels two to six of the workgroup's 4 or 8
    // tiles together — per tile 16 leaf passes + 2.25 (one tile per wave; 6 when folded alone) or + 1.25 (two per wave; 3).
    constexpr bool WGF = COPY;
    constexpr int QG = FOLD_G > 1 ? FOLD_G : 2;  // (one tile per wave: the queue is only the fallback's vehicle)
    const uint32_t w = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const uint32_t first = (blockIdx.x * 4 + w) * FOLD_G;
    bool live = first < a.n_tiles;
    if (live && a.pass == PASS_SECOND && *a.pending_count == 0) {
        // second pass with nothing handed to the general decoder: only big-unit slices matter; skip the wave
        // outright when its tiles are all small ones (the common C2 case: one load per tile, no hashing)
        bool any_big = false;
        for (uint32_t g = 0; g < FOLD_G && first + g < a.n_tiles; g++) any_big |= a.tiles[first + g].n_units == 0;
        if (!any_big) live = false;
    }
    if (!WGF && !live) return;  // (WGF: every wave reaches the workgroup's barrier below)
    FoldQueue<QG> fq;
    uint32_t *nodes = COPY ? reinterpret_cast<uint32_t *>(s_area[w]) : s_nodes[FOLD_G > 1 ? w : 0];
    uint8_t *const stage = COPY ? s_area[w] + KEEP_BYTES : nullptr;
    for (uint32_t g = 0; live && g < FOLD_G && first + g < a.n_tiles; g++) {
        const Tile t = a.tiles[first + g];
        if (a.pass == PASS_SECOND && a.tile_done && a.tile_done[first + g]) continue;  // hashed by the fused block kernel
        if (a.pass == PASS_SECOND && t.n_units) {
            // second pass: a small tile matters only if the general decoder finished one of its rows
            if (*a.pending_count == 0) continue;
            const bool mine = lane < t.n_units && a.status[t.first_unit + lane] == 2;
            if (__ballot(mine) == 0ull) continue;
        }
        LeafOut lo;
        hash_tile_leaves<COPY, false, COPY, SHIFT>(a, t, nullptr, lo, stage);
        // (one tile per wave on the store path: only a whole slice is queued — the stage's head is free by now)
        if (FOLD_G > 1 ? FoldQueue<QG>::fits(t) : (WGF && t.n_units == 0 && t.n_leaves == 64 && __ballot(lo.active) == ~0ull)) fq.add(nodes, g, t, lo);
        else fold_tile_now(a, t, lo);
    }
    if constexpr (WGF) {
        uint32_t U = 0;
        const uint32_t n = fq.uniform(&U);
        const bool simple = n == 64 && U == (uint32_t)FOLD_G && __shfl(fq.tb_root, 0) == 0;  // FOLD_G whole slices (entry i on lane i)
        // the wave's word to the others: the last 32 bytes of its own area (the stage's end: free once its tiles are through,
        // and no level's output reaches there): [simple, where tile 0's CV goes, where tile 1's goes]
        uint32_t *const mine = reinterpret_cast<uint32_t *>(s_area[w] + AREA - 32);
        if (simple) {
            fold_first_level<FOLD_G>((lds_u32 *)nodes);
            if (lane < (uint32_t)FOLD_G) mine[1 + lane] = fq.tb_out;
            fq.tb_n = lane < (uint32_t)FOLD_G ? 32u : 0u;  // what is left of the wave's table: units of 32 nodes
            fq.tb_off = 32 * lane;
        }
        if (lane == 0) mine[0] = simple ? 1u : 0u;
        __syncthreads();
        uint32_t all = 1;
#pragma unroll
        for (int v = 0; v < 4; v++) all &= *reinterpret_cast<const uint32_t *>(s_area[v] + AREA - 32);
        if (!all) {  // a workgroup with anything else in it (a table's edges, rows of other shapes): every wave finishes its own
            if (simple) fq.fold_uniform_and_write(nodes, a, 32, FOLD_G);
            else if (n) fq.fold_uniform_and_write(nodes, a, n, U);
            else fq.fold_and_write(nodes, a);
            return;
        }
        if (w != (blockIdx.x * 2654435761u) >> 30) return;
        static_assert(AREA % 32 == 0 && AREA >= (FOLD_G == 1 ? 64 + 64 + 32 + 16 + 8 : 128 + 128) * 32 + 32, "room for the levels' outputs between the nodes and the waves' words");
        uint32_t cvi = 0;
        if (lane < 4 * FOLD_G) cvi = reinterpret_cast<const uint32_t *>(s_area[lane / FOLD_G] + AREA - 32)[1 + lane % FOLD_G];
        fold_wg_tiles<FOLD_G>((lds_u32 *)reinterpret_cast<uint32_t *>(&s_area[0][0]), AREA / 32, a.tile_cv, cvi);
    } else if (FOLD_G > 1) {
        fq.fold_and_write(nodes, a);
    }
}

template <int G>
static void launch_hash_tiles_g(const HashArgs &a, hipStream_t s) {
    const uint32_t waves = (a.n_tiles + G - 1) / G;
    dim3 grid((waves + 3) / 4), block(256);
    if constexpr (G <= 2) {
        if (a.copy_to_B) {
            // (a destination that is not 16-byte aligned: the re-cutting stage is 10 KiB — one tile per wave)
            if (a.misaligned_dst) hipLaunchKernelGGL((k_hash_tiles<true, 1, true>), dim3((a.n_tiles + 3) / 4), block, 0, s, a);
            else hipLaunchKernelGGL((k_hash_tiles<true, G, false>), grid, block, 0, s, a);
            return;
        }
    }
    hipLaunchKernelGGL((k_hash_tiles<false, G>), grid, block, 0, s, a);
}

void launch_hash_tiles(const HashArgs &a, hipStream_t s) {
    if (!a.n_tiles) return;
    static int cus = 0;
    if (!cus) {
        int dev = 0;
        (void)hipGetDevice(&dev);
        if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) cus = 256;
    }
    const uint64_t resident = (uint64_t)cus * 20;  // 5 waves per SIMD at this kernel's register count
    const int cap = a.fold_tiles_max > 0 ? a.fold_tiles_max : 4;
    if (a.copy_to_B) {
        // store path: 16-20 waves per CU (the stage), and a tile's six parent levels are 6 of its 22 compress passes: two
        // tiles per wave (19 passes each) as soon as that still gives every wave slot a wave.  (Four would need 14 KiB of LDS
        // per wave, or the waiting CVs in registers: measured, the spills cost more than the passes saved.)
        const uint64_t slots = (uint64_t)cus * 16;
        const int force = a.store_tiles;
        if (force == 2 || (force == 0 && cap >= 2 && a.n_tiles >= 2 * slots - slots / 8)) launch_hash_tiles_g<2>(a, s);
        else launch_hash_tiles_g<1>(a, s);
        return;
    }
    if (cap >= 4 && a.n_tiles >= 6 * resident) launch_hash_tiles_g<4>(a, s);
    else if (cap >= 2 && a.n_tiles >= 3 * resident) launch_hash_tiles_g<2>(a, s);
    else launch_hash_tiles_g<1>(a, s);
}

// Finish units with more than 64 leaves: one wave folds the unit's tile CVs (each the root of
// a complete 64-leaf subtree, the last possibly partial) level by level, in place, then the
// last <= 64 nodes in registers.
// Units with more than 64 tile CVs (rows above 4 MiB): the level-by-level fold of one wave is a chain of
// n_cvs / 32 dependent compress passes (a 200 MiB row: 100 of them, a 2 GiB row: 1,000).  Six levels of pairing turn
// every aligned group of 64 CVs — the last, shorter one included, with the odd node carried as in the tree — into
// one node whatever its neighbours are, so the groups are folded by independent waves first and the per-unit wave
// starts from ceil(n_cvs / 64) nodes.
__global__ __launch_bounds__(64) void k_merge_groups(const BigUnit *big, const uint32_t *grp_big, const uint32_t *grp_k,
                                                    uint32_t n_grp, uint32_t *tile_cv) {
    const uint32_t lane = threadIdx.x;
    if (blockIdx.x >= n_grp) return;
    const BigUnit u = big[grp_big[blockIdx.x]];
    const uint32_t g = grp_k[blockIdx.x];
    const uint32_t cnt = u.n_cvs - 64 * g < 64 ? u.n_cvs - 64 * g : 64;
    const uint32_t *base = tile_cv + ((size_t)u.cv_base + 64 * g) * 8;
    uint32_t cv[8];
#pragma unroll
    for (int i = 0; i < 8; i++) cv[i] = lane < cnt ? base[(size_t)lane * 8 + i] : 0u;
    fold_segments(cv, 0, lane < cnt ? cnt : 0, false);
    if (lane == 0) {
#pragma unroll
        for (int i = 0; i < 8; i++) tile_cv[((size_t)u.pad + g) * 8 + i] = cv[i];
    }
}

__global__ __launch_bounds__(64) void k_merge_big(const BigUnit *big, uint32_t n_big, uint32_t *tile_cv,
                                                 uint32_t *digests) {
    const uint32_t lane = threadIdx.x;
    if (blockIdx.x >= n_big) return;
    const BigUnit u = big[blockIdx.x];
    const bool grouped = u.n_cvs > 64;  // k_merge_groups has been over this unit
    uint32_t *base = tile_cv + (size_t)(grouped ? u.pad : u.cv_base) * 8;
    uint32_t m = grouped ? (u.n_cvs + 63) / 64 : u.n_cvs;
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

// Both steps in one launch for tables whose biggest unit has at most 4 groups (rows up to 16 MiB: the reference's 8 MiB
// file slices): a workgroup of 4 waves per unit — one per SIMD, a wave per group — the group nodes through LDS, wave 0
// finishes.  One dependent launch less behind the hash kernel.  (Not for bigger units: a workgroup lives on one CU, and
// 16 waves folding 50 groups of a 200 MiB row there took 0.19 ms where the two launches, a workgroup per group, take 0.03.)
constexpr uint32_t MERGE_UNIT_WAVES = 4, MERGE_UNIT_MAX_CVS = 64 * MERGE_UNIT_WAVES;
__global__ __launch_bounds__(64 * MERGE_UNIT_WAVES) void k_merge_units(const BigUnit *big, uint32_t n_big, const uint32_t *tile_cv, uint32_t *digests) {
    __shared__ __attribute__((aligned(16))) uint32_t s_grp[MERGE_UNIT_WAVES * 8];
    const uint32_t lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const BigUnit u = big[blockIdx.x < n_big ? blockIdx.x : 0];
    const bool grouped = u.n_cvs > 64;
    const uint32_t n_grp = (u.n_cvs + 63) / 64;
    if (grouped) {
        for (uint32_t g = w; g < n_grp; g += MERGE_UNIT_WAVES) {  // (wave-uniform; at most one round)
            const uint32_t cnt = u.n_cvs - 64 * g < 64 ? u.n_cvs - 64 * g : 64;
            const uint32_t *base = tile_cv + ((size_t)u.cv_base + 64 * g) * 8;
            uint32_t cv[8];
#pragma unroll
            for (int i = 0; i < 8; i++) cv[i] = lane < cnt ? base[(size_t)lane * 8 + i] : 0u;
            fold_segments(cv, 0, lane < cnt ? cnt : 0, false);
            if (lane == 0) {
#pragma unroll
                for (int i = 0; i < 8; i++) s_grp[g * 8 + i] = cv[i];
            }
        }
    }
    __syncthreads();
    if (w != 0 || blockIdx.x >= n_big) return;
    const uint32_t m = grouped ? n_grp : u.n_cvs;
    uint32_t cv[8];
    if (grouped) {
#pragma unroll
        for (int i = 0; i < 8; i++) cv[i] = lane < m ? s_grp[lane * 8 + i] : 0u;
    } else {
        const uint32_t *base = tile_cv + (size_t)u.cv_base * 8;
#pragma unroll
        for (int i = 0; i < 8; i++) cv[i] = lane < m ? base[(size_t)lane * 8 + i] : 0u;
    }
    fold_segments(cv, 0, lane < m ? m : 0, true);
    if (lane == 0) {
#pragma unroll
        for (int i = 0; i < 8; i++) digests[(size_t)u.unit * 8 + i] = cv[i];
    }
}

void launch_merge_big(const BigUnit *big, uint32_t n_big, uint32_t *tile_cv, uint32_t *digests, const uint32_t *grp_big,
                      const uint32_t *grp_k, uint32_t n_grp, uint32_t max_cvs, hipStream_t s) {
    if (!n_big) return;
    if (n_grp && max_cvs <= MERGE_UNIT_MAX_CVS) {
        hipLaunchKernelGGL(k_merge_units, dim3(n_big), dim3(64 * MERGE_UNIT_WAVES), 0, s, big, n_big, tile_cv, digests);
        return;
    }
    if (n_grp) hipLaunchKernelGGL(k_merge_groups, dim3(n_grp), dim3(64), 0, s, big, grp_big, grp_k, n_grp, tile_cv);
    hipLaunchKernelGGL(k_merge_big, dim3(n_big), dim3(64), 0, s, big, n_big, tile_cv, digests);
}

// Compare computed digests with the index's checksum column and accumulate the read loop's
// counters (decompress.rs:L140,L169-185): counters[0..5] = total_chunks, total_written_bytes,
// verified_bytes, corrupt_bytes, corrupt_rows, decode_errors.
__global__ __launch_bounds__(256) void k_verify(const uint32_t *digests, const uint8_t *checksum,
                                                const uint64_t *usize, const int32_t *status, uint32_t n_rows,
                                                uint64_t row_begin, unsigned long long *counters,
                                                uint64_t *corrupt_rows, uint32_t corrupt_cap, const uint32_t *lean_lists, uint32_t lean_mask) {
    // lean run (api.hip, rows_settle): the kernels behind the role-split one were not launched — if it did hand a row over or
    // leave a tile on its list ([0], [3]; [1], [5]: nobody ran who could have), the run's counters are flagged
    if (lean_lists && blockIdx.x == 0 && threadIdx.x == 0) {
        uint32_t any = 0;
        for (uint32_t i = 0; i < 8; i++) if ((lean_mask >> i) & 1u) any |= lean_lists[i];
        if (any) counters[7] = 1;
    }
    unsigned long long v[6] = {0, 0, 0, 0, 0, 0};
    // grid-stride: a few dozen workgroups, so that the counters' cache line takes a few hundred atomics per run
    // (one workgroup per 256 rows put 1,200 on it for 100k rows: ~13 us of the step)
    for (uint32_t r = blockIdx.x * 256 + threadIdx.x; r < n_rows; r += gridDim.x * 256) {
        v[0] += 1;
        if (status && status[r] < 0) {
            v[5] += 1;
        } else {
            const unsigned long long len = usize[r];
            v[1] += len;
            bool ok = true;
            if (checksum) {
                const uint4 *want = reinterpret_cast<const uint4 *>(checksum) + (size_t)r * 2;
                const uint4 *got = reinterpret_cast<const uint4 *>(digests) + (size_t)r * 2;
                const uint4 w0 = want[0], w1 = want[1], g0 = got[0], g1 = got[1];
                ok = w0.x == g0.x && w0.y == g0.y && w0.z == g0.z && w0.w == g0.w &&
                     w1.x == g1.x && w1.y == g1.y && w1.z == g1.z && w1.w == g1.w;
            }
            if (ok) v[2] += len;
            else {
                v[3] += len;
                v[4] += 1;
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
                   uint64_t *corrupt_rows, uint32_t corrupt_cap, hipStream_t s, const uint32_t *lean_lists, uint32_t lean_mask) {
    if (!n_rows) return;
    hipLaunchKernelGGL(k_verify, dim3(std::min<uint32_t>((n_rows + 255) / 256, 128)), dim3(256), 0, s, digests, checksum, usize, status,
                       n_rows, row_begin, reinterpret_cast<unsigned long long *>(counters), corrupt_rows, corrupt_cap, lean_lists, lean_mask);
}

// ---- measurement hook: the VALU floor of the hash ------------------------------------------------------------
// Nothing but BLAKE3 compressions (message in registers, no memory traffic), 4 waves per SIMD on every CU: what
// one 64-lane compress pass costs a SIMD when the integer VALU is the only thing in use.  bench.py multiplies it
// by the passes a step needs to report the floor the hash sets whatever the memory system does.
__global__ __launch_bounds__(256) void k_b3_pass_ubench(uint32_t *out, uint32_t seed, int passes, unsigned long long *clk) {
    const unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    uint32_t cv[8], m[16];
#pragma unroll
    for (int i = 0; i < 8; i++) cv[i] = threadIdx.x * 7 + i + seed;
#pragma unroll
    for (int i = 0; i < 16; i++) m[i] = threadIdx.x * 13 + i * seed;
#pragma unroll 1
    for (int p = 0; p < passes; p++) {
        b3::compress(cv, m, (uint32_t)p, 0, 64, 0);
        m[p & 15] ^= cv[0];
    }
    uint32_t x = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) x ^= cv[i];
    out[blockIdx.x * 256 + threadIdx.x] = x;
    if (clk && blockIdx.x == 7 && threadIdx.x == 0) {  // shader cycles and 100 MHz ticks of one wave's life: the clock the chip held
        clk[0] = __builtin_amdgcn_s_memtime() - c0;
        clk[1] = __builtin_amdgcn_s_memrealtime() - r0;
    }
}

int measure_b3_pass_ns(int cus, hipStream_t s, float *ns_per_pass_per_simd, float *ghz) {
    const int grid = cus * 4, passes = 200;  // 4 blocks of 4 waves per CU = 4 waves per SIMD
    uint32_t *d = nullptr;
    if (hipMalloc(&d, (size_t)grid * 256 * 4 + 64) != hipSuccess) return -3;
    unsigned long long *clk = reinterpret_cast<unsigned long long *>(d + (size_t)grid * 256);
    hipEvent_t t0, t1;
    (void)hipEventCreate(&t0);
    (void)hipEventCreate(&t1);
    float best = 0.f;
    for (int rep = 0; rep < 3; rep++) {
        (void)hipEventRecord(t0, s);
        hipLaunchKernelGGL(k_b3_pass_ubench, dim3(grid), dim3(256), 0, s, d, (uint32_t)(rep + 1), passes, clk);
        (void)hipEventRecord(t1, s);
        (void)hipEventSynchronize(t1);
        float ms = 0.f;
        (void)hipEventElapsedTime(&ms, t0, t1);
        if (rep && (best == 0.f || ms < best)) best = ms;  // first launch: code upload + clocks ramping
    }
    (void)hipEventDestroy(t0);
    (void)hipEventDestroy(t1);
    unsigned long long hc[2] = {0, 1};
    (void)hipMemcpy(hc, clk, 16, hipMemcpyDeviceToHost);
    if (ghz) *ghz = hc[1] ? (float)((double)hc[0] / (double)hc[1] * 0.1) : 0.f;
    (void)hipFree(d);
    *ns_per_pass_per_simd = best * 1e6f / (float)(passes * 4);
    return hipGetLastError() == hipSuccess ? 0 : -2;
}

}  // namespace zn
