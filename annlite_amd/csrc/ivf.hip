// ivf.hip -- the coarse-quantiser side of a pruned (IVF) PQ search (gfx950 only): which cells a query probes, the
// grouping of (query, cell) pairs into the scan's query tiles, and the merge of a query's per-cell lists.
//
// Reference seam: AnnLite._cell_selection (annlite/index.py:458-466: cdist(query, vq codebook) -> top_k(n_probe)),
// CellContainer.ivf_search (container.py:88-144: per-cell search, lists concatenated and re-sorted).  The reference
// sets n_probe = max(n_probe, n_cells) (index.py:94), i.e. it always visits every cell; the pruned search is the
// build's extension of the same structure.
//
// The scan itself is annlite_ivf_search_topk (scan.hip; M = 16, k <= 16: the byte-table kernel in cell tiles, exact sums in the
// tile, annlite_ivf_merge_lists below) or annlite_pq_search_tiles (the u16 tables, integer sums + annlite_ivf_rescore below): the rows
// of a cell are contiguous in the code table, a query tile = up to QT queries that probe the same cell, one work item per tile.
#include "common.h"
#include "scan_common.h"

namespace annlite {

// ---- cell selection ------------------------------------------------------------------------------
// One 256-thread workgroup per QB = 4 queries: the query vectors sit in LDS (read by broadcast), thread t owns the
// centroids t, t + 256, ... and keeps QB running sums while it streams a centroid row ONCE; the QB x C distances
// go to LDS, where one wave per query picks the P nearest under the fixed order (distance asc, cell asc).
constexpr int kSelQB = 4;
template <int KIND>  // 0: squared L2, 1: negative inner product
__global__ __launch_bounds__(256) void ivf_select_cells_kernel(const float *__restrict__ q, int B, int D,
                                                              const float *__restrict__ cent, int C, int P,
                                                              int32_t *__restrict__ cells) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float *qs = (float *)smem;               // [QB][D]
    float *ds = qs + (size_t)kSelQB * D;     // [QB][C]
    const int tid = threadIdx.x;
    const int b0 = blockIdx.x * kSelQB;
    for (int i = tid; i < kSelQB * D; i += 256) {
        const int qi = i / D, j = i - qi * D;
        qs[i] = b0 + qi < B ? q[(int64_t)(b0 + qi) * D + j] : 0.f;
    }
    __syncthreads();
    for (int c = tid; c < C; c += 256) {
        float acc[kSelQB];
#pragma unroll
        for (int u = 0; u < kSelQB; ++u) acc[u] = 0.f;
        const float *cr = cent + (int64_t)c * D;
        auto step = [&](float cv, int j) {  // sequential in j: one defined summation order per (query, cell)
#pragma unroll
            for (int u = 0; u < kSelQB; ++u) {
                const float qv = qs[u * D + j];
                if constexpr (KIND == 0) {
                    const float d = cv - qv;
                    acc[u] = __builtin_fmaf(d, d, acc[u]);
                } else {
                    acc[u] = __builtin_fmaf(cv, qv, acc[u]);
                }
            }
        };
        if ((D & 3) == 0) {
            for (int j = 0; j < D; j += 4) {  // 16-byte loads of the centroid row (threads are D floats apart)
                const f32x4 cv = *(const f32x4 *)(cr + j);
                step(cv.x, j);
                step(cv.y, j + 1);
                step(cv.z, j + 2);
                step(cv.w, j + 3);
            }
        } else {
            for (int j = 0; j < D; ++j) step(cr[j], j);
        }
#pragma unroll
        for (int u = 0; u < kSelQB; ++u) ds[u * C + c] = KIND == 0 ? acc[u] : -acc[u];
    }
    __syncthreads();
    // the P nearest, one wave per query: P rounds of "lane-local minimum over its cells -> wave argmin -> mark taken"
    // under the fixed order (distance asc, cell asc).  (Rank counting every cell against every other was C^2 per
    // query: 0.88 ms for 1024 queries x 1024 cells.)
    const int lane = tid & 63, wave = tid >> 6;
    for (int u = wave; u < kSelQB; u += 4) {
        if (b0 + u >= B) continue;
        float *row = ds + u * C;
        for (int p = 0; p < P; ++p) {
            float best = __builtin_inff();
            int bi = 0x7fffffff;
            for (int c = lane; c < C; c += 64) {
                const float v = row[c];
                if (v < best || (v == best && c < bi)) best = v, bi = c;
            }
            // wave argmin by DPP (round 6): row_shr 1 / 2 / 4 / 8 inside the 16-lane rows, row_bcast 15 / 31 across them -- lane 63 ends
            // up with the minimum of all 64 (the order (distance, cell) as a 64-bit key: ordered float bits, then the cell), read back
            // through an SGPR.  (Six __shfl_xor steps of two values each went through ds_bpermute: ~1 us per round, 16 of the
            // kernel's 26 us at 16 probes.)
            {
                uint32_t khi = f32_to_ordered(best + 0.f);  // (never NaN: the lane's minimum starts at +inf; -0 as +0: the float order's tie)
                uint32_t klo = (uint32_t)bi;
                auto step = [&](int ctrl_id) {
                    uint32_t ohi, olo;
                    // (old = all-ones: a lane without a source in this step keeps the key that loses every comparison)
                    switch (ctrl_id) {
                        case 0: ohi = (uint32_t)__builtin_amdgcn_update_dpp(-1, (int)khi, 0x111, 0xf, 0xf, false), olo = (uint32_t)__builtin_amdgcn_update_dpp(-1, (int)klo, 0x111, 0xf, 0xf, false); break;
                        case 1: ohi = (uint32_t)__builtin_amdgcn_update_dpp(-1, (int)khi, 0x112, 0xf, 0xf, false), olo = (uint32_t)__builtin_amdgcn_update_dpp(-1, (int)klo, 0x112, 0xf, 0xf, false); break;
                        case 2: ohi = (uint32_t)__builtin_amdgcn_update_dpp(-1, (int)khi, 0x114, 0xf, 0xf, false), olo = (uint32_t)__builtin_amdgcn_update_dpp(-1, (int)klo, 0x114, 0xf, 0xf, false); break;
                        case 3: ohi = (uint32_t)__builtin_amdgcn_update_dpp(-1, (int)khi, 0x118, 0xf, 0xf, false), olo = (uint32_t)__builtin_amdgcn_update_dpp(-1, (int)klo, 0x118, 0xf, 0xf, false); break;
                        case 4: ohi = (uint32_t)__builtin_amdgcn_update_dpp(-1, (int)khi, 0x142, 0xa, 0xf, false), olo = (uint32_t)__builtin_amdgcn_update_dpp(-1, (int)klo, 0x142, 0xa, 0xf, false); break;
                        default: ohi = (uint32_t)__builtin_amdgcn_update_dpp(-1, (int)khi, 0x143, 0xc, 0xf, false), olo = (uint32_t)__builtin_amdgcn_update_dpp(-1, (int)klo, 0x143, 0xc, 0xf, false); break;
                    }
                    if (ohi < khi || (ohi == khi && olo < klo)) khi = ohi, klo = olo;
                };
                step(0), step(1), step(2), step(3), step(4), step(5);
                const uint32_t whi = (uint32_t)__builtin_amdgcn_readlane((int)khi, 63);
                bi = __builtin_amdgcn_readlane((int)klo, 63);
                best = ordered_to_f32(whi);
            }
            // (a query with NaN / inf components finds no minimum: fall back to cell p so that the plan never sees an
            // out-of-range cell -- the results of such a query are meaningless either way)
            if (lane == 0) cells[(int64_t)(b0 + u) * P + p] = bi < C ? bi : p;
            if (bi < C && lane == (bi & 63)) row[bi] = __builtin_inff();  // taken (LDS ops of a wave execute in order)
        }
    }
}

// ---- grouping (query, cell) pairs into query tiles -----------------------------------------------
// ONE workgroup.  Pairs that probe the same cell are packed into tiles of `qt` slots; the tiles of a cell are
// consecutive and the cells come in `order` (descending size: the scan hands tiles out longest first).
//   vmap[v]      query of slot v, or -1 (padding)            [n_tiles_max * qt]
//   slot_of[i]   slot of pair i = (query i / P, probe i % P) [n_pairs]
//   tile_rows[t] (begin, end) of the tile's cell, begin = -1 for tiles past the last used one
// The slot order inside a cell depends on atomic arrival order; results do not (every slot has its own list).
// n_first > 0 (annlite_ivf_search_topk): the pairs of every query's n_first NEAREST cells (probe ranks < n_first) get tiles of their own,
// and those tiles come FIRST -- the scan shares a query's bound between its tiles, and the k best rows of the nearest cells bound the
// others' candidates from the start.  Internally 2 C "virtual cells": v = c (first class) or C + c, walked in `order` twice.
__device__ __forceinline__ void ivf_plan_body(const int32_t *__restrict__ cells, int n_pairs, int P, int C_real,
                                                       int qt, const int64_t *__restrict__ cell_rows,
                                                       const int32_t *__restrict__ order_real, int n_tiles_max,
                                                       int32_t *__restrict__ vmap, int32_t *__restrict__ slot_of,
                                                       int64_t *__restrict__ tile_rows, int32_t *__restrict__ n_tiles_used,
                                                       int n_first, unsigned char *smem) {
    const int C = n_first > 0 ? 2 * C_real : C_real;  // virtual cells
    int *cnt = (int *)smem;        // [C] pairs per cell
    int *tstart = cnt + C;         // [C] first tile of the cell
    int *wsum = tstart + C;        // [16]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    auto vcell = [&](int i) -> int {  // virtual cell of pair i
        const int c = (unsigned)cells[i] < (unsigned)C_real ? cells[i] : 0;
        return (n_first > 0 && i % P >= n_first) ? C_real + c : c;
    };
    auto order = [&](int j) -> int { return j < C_real ? order_real[j] : C_real + order_real[j - C_real]; };
    // (round 6) latency, not work, is what this single workgroup costs (it ran 24 us for 16384 pairs: every pair went cell load ->
    // LDS atomic -> store -> reload, one dependent global round trip per loop iteration).  Up to kReg pairs per thread now stay in
    // REGISTERS between the two passes (their loads all in flight together), and the cells' order / row ranges are requested before
    // the first barrier; larger batches keep the loops through memory.
    constexpr int kReg = 24, kCellReg = 4;  // (24: 1024 queries x 22 entries -- 16 probed cells, the nearest two in four parts -- stay in registers)
    const bool in_regs = n_pairs <= 1024 * kReg;
    const int per = (C + 1023) / 1024;  // cells per thread in the scan of the tile counts; thread t owns positions [t*per, (t+1)*per)
    const bool cells_in_regs = per <= kCellReg;
    int vc[kReg], rk[kReg];
    int c_loc[kCellReg];
    int64_t rb_loc[kCellReg], re_loc[kCellReg];
    if (in_regs) {
#pragma unroll
        for (int u = 0; u < kReg; ++u) {
            const int i = u * 1024 + tid;
            vc[u] = i < n_pairs ? vcell(i) : -1;
        }
    }
    if (cells_in_regs) {
#pragma unroll
        for (int u = 0; u < kCellReg; ++u) {
            const int j = tid * per + u;
            c_loc[u] = -1, rb_loc[u] = re_loc[u] = 0;
            if (u < per && j < C) {
                const int c = order(j), cr = c < C_real ? c : c - C_real;
                c_loc[u] = c, rb_loc[u] = cell_rows[2 * cr], re_loc[u] = cell_rows[2 * cr + 1];
            }
        }
    }
    for (int i = tid; i < C; i += 1024) cnt[i] = 0;
    for (int i = tid; i < n_tiles_max * qt; i += 1024) vmap[i] = -1;
    for (int i = tid; i < n_tiles_max; i += 1024) {
        tile_rows[2 * i] = -1;
        tile_rows[2 * i + 1] = -1;
    }
    __syncthreads();
    if (in_regs) {
#pragma unroll
        for (int u = 0; u < kReg; ++u) rk[u] = vc[u] >= 0 ? atomicAdd(&cnt[vc[u]], 1) : 0;  // rank inside the (virtual) cell
    } else {
        for (int i = tid; i < n_pairs; i += 1024) slot_of[i] = atomicAdd(&cnt[vcell(i)], 1);
    }
    __syncthreads();
    // exclusive scan of the tile counts in `order`
    int local = 0;
    if (cells_in_regs) {
#pragma unroll
        for (int u = 0; u < kCellReg; ++u)
            if (c_loc[u] >= 0) local += (cnt[c_loc[u]] + qt - 1) / qt;
    } else {
        for (int u = 0; u < per; ++u) {
            const int j = tid * per + u;
            if (j < C) local += (cnt[order(j)] + qt - 1) / qt;
        }
    }
    int incl = local;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const int v = __shfl_up(incl, o);
        if (lane >= o) incl += v;
    }
    if (lane == 63) wsum[wave] = incl;
    __syncthreads();
    int base = 0;
    for (int w = 0; w < wave; ++w) base += wsum[w];
    int run = base + incl - local;
    auto place = [&](int c, int64_t rb, int64_t re) {
        const int nt = (cnt[c] + qt - 1) / qt;
        tstart[c] = run;
        for (int t = 0; t < nt; ++t) {
            tile_rows[2 * (int64_t)(run + t)] = rb;
            tile_rows[2 * (int64_t)(run + t) + 1] = re;
        }
        run += nt;
    };
    if (cells_in_regs) {
#pragma unroll
        for (int u = 0; u < kCellReg; ++u)
            if (c_loc[u] >= 0) place(c_loc[u], rb_loc[u], re_loc[u]);
    } else {
        for (int u = 0; u < per; ++u) {
            const int j = tid * per + u;
            if (j < C) {
                const int c = order(j), cr = c < C_real ? c : c - C_real;
                place(c, cell_rows[2 * cr], cell_rows[2 * cr + 1]);
            }
        }
    }
    if (tid == 1023 && n_tiles_used) *n_tiles_used = run;
    __syncthreads();
    if (in_regs) {
#pragma unroll
        for (int u = 0; u < kReg; ++u) {
            const int i = u * 1024 + tid;
            if (vc[u] >= 0) {
                const int v = (tstart[vc[u]] + rk[u] / qt) * qt + rk[u] % qt;
                slot_of[i] = v;
                vmap[v] = i / P;
            }
        }
    } else {
        for (int i = tid; i < n_pairs; i += 1024) {
            const int r = slot_of[i];
            const int v = (tstart[vcell(i)] + r / qt) * qt + r % qt;
            slot_of[i] = v;
            vmap[v] = i / P;
        }
    }
}
__global__ __launch_bounds__(1024) void ivf_plan_kernel(const int32_t *__restrict__ cells, int n_pairs, int P, int C_real,
                                                       int qt, const int64_t *__restrict__ cell_rows,
                                                       const int32_t *__restrict__ order_real, int n_tiles_max,
                                                       int32_t *__restrict__ vmap, int32_t *__restrict__ slot_of,
                                                       int64_t *__restrict__ tile_rows, int32_t *__restrict__ n_tiles_used,
                                                       int n_first) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    ivf_plan_body(cells, n_pairs, P, C_real, qt, cell_rows, order_real, n_tiles_max, vmap, slot_of, tile_rows, n_tiles_used, n_first, smem);
}

// ---- exact re-score of a query's candidate lists ---------------------------------------------------
// One 256-thread workgroup per query.  The query's fp32 table [M][Ks] (what get_dist_mat returns for it) sits in
// LDS; the candidate rows its P probed (cell, slot) lists hold get their exact ascending-m fp32 ADC sum -- the
// reference's sum (space_pq.h:32-35 / pq_bindings.pyx:44-45), bit for bit -- and the k best under the fixed order
// (distance asc, id asc) survive: per wave in a sorted wave list, the four lists are merged at the end.
// A slot whose list overflowed (count 0xffffffff) is re-scored over ALL rows of its cell.
// codes: the cell-sorted table in the PLAIN layout.  ids: row_ids[row] (external offsets), < 2^32.
__global__ __launch_bounds__(256) void ivf_rescore_kernel(const float *__restrict__ lut, int B, int M, int Ks,
                                                         const uint8_t *__restrict__ codes,
                                                         const uint32_t *__restrict__ valid,
                                                         const uint32_t *__restrict__ cand, int cand_cap,
                                                         const uint32_t *__restrict__ cand_count,
                                                         const int32_t *__restrict__ slot_of, int P,
                                                         const int64_t *__restrict__ tile_rows, int qt,
                                                         const int64_t *__restrict__ row_ids, int64_t id_base, int k,
                                                         float *__restrict__ out_d, int64_t *__restrict__ out_i,
                                                         int sqrt_out) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float *tab = (float *)smem;                                               // [M][Ks]
    unsigned long long *wl = (unsigned long long *)(smem + (size_t)M * Ks * 4);  // [4][64]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int b = blockIdx.x;
    {
        const f32x4 *src = (const f32x4 *)(lut + (int64_t)b * M * Ks);
        for (int i = tid; i < M * Ks / 4; i += 256) ((f32x4 *)tab)[i] = src[i];
    }
    __syncthreads();
    const int km1 = k - 1;
    const int MW = M / 4;
    WaveList L;
    L.reset();
    uint32_t thr_hi = kKeyInfHi, thr_lo = kIdNone;
    for (int p = 0; p < P; ++p) {
        const int v = slot_of[(int64_t)b * P + p];
        const uint32_t n = cand_count[v];
        const bool range = n == 0xffffffffu;
        int64_t rb = 0, total = n;
        if (range) {
            rb = tile_rows[2 * (int64_t)(v / qt)];
            total = tile_rows[2 * (int64_t)(v / qt) + 1] - rb;
        }
        const uint32_t *cl = cand + (int64_t)v * cand_cap;
        for (int64_t i0 = (int64_t)wave * 64; i0 < total; i0 += 256) {
            const int64_t i = i0 + lane;
            bool act = i < total;
            int64_t row = 0;
            if (act) row = range ? rb + i : (int64_t)cl[i];
            if (act && range && valid) act = (valid[row >> 5] >> (row & 31)) & 1u;
            float d = 0.f;
            const uint32_t *p32 = (const uint32_t *)(codes + row * M);
            for (int w = 0; w < MW; ++w) {
                const uint32_t word = p32[w];
                const float *t = tab + (w * 4) * Ks;
                d += t[word & 0xffu];
                d += t[Ks + ((word >> 8) & 0xffu)];
                d += t[2 * Ks + ((word >> 16) & 0xffu)];
                d += t[3 * Ks + (word >> 24)];
            }
            const uint32_t hi = f32_to_ordered(d);
            const uint32_t lo = act ? (uint32_t)(row_ids ? row_ids[row] : row) : kIdNone;
            const unsigned long long pm = __ballot(act && key_less(hi, lo, thr_hi, thr_lo));
            if (pm) wavelist_offer(L, pm, hi, lo, km1, thr_hi, thr_lo, lane);
        }
    }
    wl[wave * 64 + lane] = ((unsigned long long)L.hi << 32) | L.lo;
    __syncthreads();
    if (wave == 0) {
        for (int w = 1; w < 4; ++w) {
            const unsigned long long o = wl[w * 64 + lane];
            wavelist_merge_sorted(L, (uint32_t)(o >> 32), (uint32_t)o, lane);
        }
        if (lane < k) {
            const bool none = (L.hi == kKeyInfHi && L.lo == kIdNone);
            const float d = none ? __builtin_inff() : ordered_to_f32(L.hi);
            out_d[(int64_t)b * k + lane] = sqrt_out && !none ? __builtin_sqrtf(d) : d;
            out_i[(int64_t)b * k + lane] = none ? (int64_t)-1 : id_base + (int64_t)L.lo;
        }
    }
}

// ---- candidate lists of a query as one dense id row (exact re-rank on the float vectors) -------------
// One wave per query: its P lists are copied back to back into ids[b][0..R), translated to external ids, the
// rest of the row is -1.  Lists that overflowed (count 0xffffffff) contribute nothing and entries beyond R are
// dropped: the float re-rank's candidate set is a heuristic either way.
__global__ __launch_bounds__(256) void ivf_candidate_ids_kernel(const uint32_t *__restrict__ cand, int cand_cap,
                                                               const uint32_t *__restrict__ cand_count,
                                                               const int32_t *__restrict__ slot_of, int B, int P,
                                                               const int64_t *__restrict__ row_ids, int64_t id_base,
                                                               int64_t *__restrict__ ids, int R) {
    const int lane = threadIdx.x & 63;
    const int b = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (b >= B) return;
    int64_t *row = ids + (int64_t)b * R;
    int off = 0;
    for (int p = 0; p < P && off < R; ++p) {
        const int v = slot_of[(int64_t)b * P + p];
        uint32_t n = cand_count[v];
        if (n == 0xffffffffu) n = 0;
        if ((int)n > R - off) n = (uint32_t)(R - off);
        for (uint32_t j = lane; j < n; j += 64) {
            const int64_t r = (int64_t)cand[(int64_t)v * cand_cap + j];
            row[off + j] = id_base + (row_ids ? row_ids[r] : r);
        }
        off += (int)n;
    }
    for (int j = off + lane; j < R; j += 64) row[j] = -1;
}

// ---- merge of a query's per-cell lists (annlite_ivf_search_topk) -----------------------------------
// The cell-tile scan (adc_scan_q8_kernel<..., TL>) leaves, per SLOT, the k smallest (exact ADC sum, table row) keys of the
// slot's cell, ascending.  One wave per query: the P lists of its slots (slot_of) -- 64 keys at a time, re-keyed by the rows'
// external ids (row_ids: ascending inside a cell, so a slot's k best under (distance, table row) ARE its k best under
// (distance, id); between cells the id decides) -- are folded into a sorted wave list; its first k entries are the result
// under the fixed order (distance asc, id asc): what CellContainer.ivf_search's concatenate-and-sort returns for the probed cells
// (container.py:88-144).
__global__ __launch_bounds__(256) void ivf_merge_lists_kernel(const unsigned long long *__restrict__ lists, int k,
                                                             const int32_t *__restrict__ slot_of, int B, int P,
                                                             const int64_t *__restrict__ row_ids, int64_t id_base,
                                                             float *__restrict__ out_d, int64_t *__restrict__ out_i, int sqrt_out) {
    const int lane = threadIdx.x & 63;
    const int b = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (b >= B) return;
    WaveList L;
    L.reset();
    const int total = P * k;
    for (int i0 = 0; i0 < total; i0 += 64) {
        const int i = i0 + lane;
        uint32_t hi = kKeyInfHi, lo = kIdNone;
        if (i < total) {
            const int p = i / k, j = i - p * k;
            const int v = slot_of[(int64_t)b * P + p];
            const unsigned long long key = lists[(int64_t)v * k + j];
            if (key != ~0ull) {
                hi = (uint32_t)(key >> 32);
                const uint32_t row = (uint32_t)key;
                lo = row_ids ? (uint32_t)row_ids[row] : row;
            }
        }
        wave_sort64(hi, lo, lane);
        wavelist_merge_sorted(L, hi, lo, lane);
    }
    if (lane < k) {
        const bool none = (L.hi == kKeyInfHi && L.lo == kIdNone);
        const float d = none ? __builtin_inff() : ordered_to_f32(L.hi);
        out_d[(int64_t)b * k + lane] = sqrt_out && !none ? __builtin_sqrtf(d) : d;
        out_i[(int64_t)b * k + lane] = none ? (int64_t)-1 : id_base + (int64_t)L.lo;
    }
}

// ---- the per-slot lists as candidate ids (annlite_ivf_search_candidates) ---------------------------
// out[b][p * k + j] = external id of key j of pair (b, p)'s list (the cell tiles' private lists: the cell's best <= k rows at or below the
// query's first bound, ascending), -1 where the list is shorter: the input of the exact re-rank (annlite_rerank_topk).
__global__ __launch_bounds__(256) void ivf_lists_to_ids_kernel(const unsigned long long *__restrict__ lists, int k,
                                                              const int32_t *__restrict__ slot_of, int64_t n_pairs,
                                                              const int64_t *__restrict__ row_ids, int64_t id_base,
                                                              int64_t *__restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n_pairs * k) return;
    const int64_t pair = i / k;
    const int j = (int)(i - pair * k);
    const unsigned long long key = lists[(int64_t)slot_of[pair] * k + j];
    const uint32_t row = (uint32_t)key;
    out[i] = key == ~0ull ? (int64_t)-1 : id_base + (row_ids ? row_ids[row] : (int64_t)row);
}

}  // namespace annlite

using namespace annlite;

extern "C" int annlite_ivf_select_cells(int kind, const float *queries_dev, int64_t B, int64_t D,
                                        const float *centroids_dev, int64_t C, int64_t P, int32_t *cells_dev,
                                        void *stream) {
    ANNLITE_REQUIRE(kind == 0 || kind == 1, "kind must be 0 (squared L2) or 1 (negative inner product), got %d", kind);
    ANNLITE_REQUIRE(B >= 0 && D >= 1 && C >= 1 && P >= 1 && P <= C, "bad shape B=%lld D=%lld C=%lld P=%lld", (long long)B,
                    (long long)D, (long long)C, (long long)P);
    const size_t lds = (size_t)kSelQB * (size_t)(D + C) * 4;
    ANNLITE_REQUIRE(lds <= 160 * 1024, "n_cells + dim too large for the selection kernel (%zu B of LDS)", lds);
    if (B == 0) return ANNLITE_OK;
    ANNLITE_REQUIRE(queries_dev && centroids_dev && cells_dev, "null device pointer");
    const unsigned grid = (unsigned)((B + kSelQB - 1) / kSelQB);
    if (kind == 0) {
        ANNLITE_HIP_TRY(hipFuncSetAttribute((const void *)ivf_select_cells_kernel<0>,
                                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL(ivf_select_cells_kernel<0>, dim3(grid), dim3(256), lds, (hipStream_t)stream, queries_dev, (int)B,
                           (int)D, centroids_dev, (int)C, (int)P, cells_dev);
    } else {
        ANNLITE_HIP_TRY(hipFuncSetAttribute((const void *)ivf_select_cells_kernel<1>,
                                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL(ivf_select_cells_kernel<1>, dim3(grid), dim3(256), lds, (hipStream_t)stream, queries_dev, (int)B,
                           (int)D, centroids_dev, (int)C, (int)P, cells_dev);
    }
    return launch_status("ivf_select_cells_kernel");
}

extern "C" int64_t annlite_ivf_max_tiles(int64_t B, int64_t P, int64_t C, int64_t qt) {
    if (B <= 0 || P <= 0 || C <= 0 || qt <= 0) return 0;
    const int64_t pairs = B * P;
    // every probed cell has at most one partly filled tile
    return pairs / qt + (C < pairs ? C : pairs);
}
// ... with the nearest cells' pairs in tiles of their own (annlite_ivf_plan_first): two classes of cells
extern "C" int64_t annlite_ivf_max_tiles_first(int64_t B, int64_t P, int64_t C, int64_t qt) {
    if (B <= 0 || P <= 0 || C <= 0 || qt <= 0) return 0;
    const int64_t pairs = B * P;
    return pairs / qt + (2 * C < pairs ? 2 * C : pairs);
}

extern "C" int annlite_ivf_plan(const int32_t *cells_dev, int64_t B, int64_t P, int64_t C, int64_t qt,
                                const int64_t *cell_rows_dev, const int32_t *cell_order_dev, int64_t n_tiles_max,
                                int32_t *vmap_dev, int32_t *slot_of_dev, int64_t *tile_rows_dev, int32_t *n_tiles_used_dev,
                                void *stream) {
    return annlite_ivf_plan_first(cells_dev, B, P, C, qt, cell_rows_dev, cell_order_dev, n_tiles_max, vmap_dev, slot_of_dev, tile_rows_dev,
                                  n_tiles_used_dev, 0, stream);
}

extern "C" int annlite_ivf_plan_first(const int32_t *cells_dev, int64_t B, int64_t P, int64_t C, int64_t qt,
                                      const int64_t *cell_rows_dev, const int32_t *cell_order_dev, int64_t n_tiles_max,
                                      int32_t *vmap_dev, int32_t *slot_of_dev, int64_t *tile_rows_dev, int32_t *n_tiles_used_dev,
                                      int64_t n_first, void *stream) {
    ANNLITE_REQUIRE(B >= 0 && P >= 1 && C >= 1 && C <= 16384 && qt >= 1 && n_first >= 0, "bad shape B=%lld P=%lld C=%lld qt=%lld n_first=%lld",
                    (long long)B, (long long)P, (long long)C, (long long)qt, (long long)n_first);
    if (n_first >= P) n_first = 0;  // (every pair in the first class: one class)
    ANNLITE_REQUIRE(n_tiles_max >= (n_first > 0 ? annlite_ivf_max_tiles_first(B, P, C, qt) : annlite_ivf_max_tiles(B, P, C, qt)),
                    "n_tiles_max=%lld too small (annlite_ivf_max_tiles / _first)", (long long)n_tiles_max);
    ANNLITE_REQUIRE(B * P < (1ll << 31) && n_tiles_max * qt < (1ll << 31), "too many (query, cell) pairs");
    if (B == 0) return ANNLITE_OK;
    ANNLITE_REQUIRE(cells_dev && cell_rows_dev && cell_order_dev && vmap_dev && slot_of_dev && tile_rows_dev,
                    "null device pointer");
    const size_t lds = (size_t)(2 * (n_first > 0 ? 2 * C : C) + 16) * 4;
    ANNLITE_HIP_TRY(hipFuncSetAttribute((const void *)ivf_plan_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(ivf_plan_kernel, dim3(1), dim3(1024), lds, (hipStream_t)stream, cells_dev, (int)(B * P), (int)P, (int)C,
                       (int)qt, cell_rows_dev, cell_order_dev, (int)n_tiles_max, vmap_dev, slot_of_dev, tile_rows_dev,
                       n_tiles_used_dev, (int)n_first);
    return launch_status("ivf_plan_kernel");
}

extern "C" int annlite_ivf_rescore(const float *lut_bmk_dev, int64_t B, int64_t M, int64_t Ks,
                                   const void *codes_plain_dev, int64_t N, const uint32_t *valid_bits_dev,
                                   const uint32_t *cand_dev, int64_t cand_cap, const uint32_t *cand_count_dev,
                                   const int32_t *slot_of_dev, int64_t P, const int64_t *tile_rows_dev, int64_t qt,
                                   const int64_t *row_ids_dev, int64_t id_base, int64_t k, float *out_dist_dev,
                                   int64_t *out_id_dev, int flags, void *stream) {
    ANNLITE_REQUIRE(B >= 0 && P >= 1 && k >= 1 && k <= 64, "bad B=%lld P=%lld k=%lld (k<=64)", (long long)B, (long long)P,
                    (long long)k);
    ANNLITE_REQUIRE(M >= 4 && M % 4 == 0 && Ks >= 1 && Ks <= 256 && (M * Ks) % 4 == 0, "bad M=%lld Ks=%lld (uint8 codes, M %% 4 == 0)",
                    (long long)M, (long long)Ks);
    ANNLITE_REQUIRE(N >= 0 && N < (1ll << 32) - 1 && qt >= 1 && cand_cap >= 1, "bad N / qt / cand_cap");
    if (B == 0) return ANNLITE_OK;
    ANNLITE_REQUIRE(lut_bmk_dev && codes_plain_dev && cand_dev && cand_count_dev && slot_of_dev && tile_rows_dev &&
                        out_dist_dev && out_id_dev,
                    "null device pointer");
    const size_t lds = (size_t)M * Ks * 4 + 4 * 64 * 8;
    ANNLITE_REQUIRE(lds <= 160 * 1024, "table of %zu B does not fit the LDS", lds);
    ANNLITE_HIP_TRY(hipFuncSetAttribute((const void *)ivf_rescore_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(ivf_rescore_kernel, dim3((unsigned)B), dim3(256), lds, (hipStream_t)stream, lut_bmk_dev, (int)B, (int)M,
                       (int)Ks, (const uint8_t *)codes_plain_dev, valid_bits_dev, cand_dev, (int)cand_cap, cand_count_dev,
                       slot_of_dev, (int)P, tile_rows_dev, (int)qt, row_ids_dev, id_base, (int)k, out_dist_dev, out_id_dev,
                       (flags & ANNLITE_FLAG_SQRT) ? 1 : 0);
    return launch_status("ivf_rescore_kernel");
}

extern "C" int annlite_ivf_candidate_ids(const uint32_t *cand_dev, int64_t cand_cap, const uint32_t *cand_count_dev,
                                         const int32_t *slot_of_dev, int64_t B, int64_t P, const int64_t *row_ids_dev,
                                         int64_t id_base, int64_t *out_ids_dev, int64_t R, void *stream) {
    ANNLITE_REQUIRE(B >= 0 && P >= 1 && R >= 1 && R < (1ll << 31) && cand_cap >= 1, "bad B=%lld P=%lld R=%lld", (long long)B,
                    (long long)P, (long long)R);
    if (B == 0) return ANNLITE_OK;
    ANNLITE_REQUIRE(cand_dev && cand_count_dev && slot_of_dev && out_ids_dev, "null device pointer");
    hipLaunchKernelGGL(ivf_candidate_ids_kernel, dim3((unsigned)((B + 3) / 4)), dim3(256), 0, (hipStream_t)stream, cand_dev,
                       (int)cand_cap, cand_count_dev, slot_of_dev, (int)B, (int)P, row_ids_dev, id_base, out_ids_dev, (int)R);
    return launch_status("ivf_candidate_ids_kernel");
}

extern "C" int annlite_ivf_merge_lists(const uint64_t *lists_dev, int64_t k, const int32_t *slot_of_dev, int64_t B, int64_t P,
                                       const int64_t *row_ids_dev, int64_t id_base, float *out_dist_dev, int64_t *out_id_dev,
                                       int flags, void *stream) {
    ANNLITE_REQUIRE(B >= 0 && P >= 1 && k >= 1 && k <= 64 && P * k < (1ll << 30), "bad B=%lld P=%lld k=%lld (k<=64)", (long long)B,
                    (long long)P, (long long)k);
    if (B == 0) return ANNLITE_OK;
    ANNLITE_REQUIRE(lists_dev && slot_of_dev && out_dist_dev && out_id_dev, "null device pointer");
    hipLaunchKernelGGL(ivf_merge_lists_kernel, dim3((unsigned)((B + 3) / 4)), dim3(256), 0, (hipStream_t)stream,
                       (const unsigned long long *)lists_dev, (int)k, slot_of_dev, (int)B, (int)P, row_ids_dev, id_base, out_dist_dev,
                       out_id_dev, (flags & ANNLITE_FLAG_SQRT) ? 1 : 0);
    return launch_status("ivf_merge_lists_kernel");
}

int annlite::launch_ivf_lists_to_ids(const unsigned long long *lists, int64_t k, const int32_t *slot_of, int64_t B, int64_t P,
                                     const int64_t *row_ids, int64_t id_base, int64_t *out_ids, hipStream_t st) {
    const int64_t total = B * P * k;
    if (total == 0) return ANNLITE_OK;
    hipLaunchKernelGGL(ivf_lists_to_ids_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, lists, (int)k, slot_of, B * P, row_ids,
                       id_base, out_ids);
    return launch_status("ivf_lists_to_ids_kernel");
}
