// scan_common.h -- what the scan translation units share: kernel arguments, the work-item mapping, the
// skew / byte helpers and the launcher prototypes (gfx950 only).
#pragma once
#include <stdlib.h>
#include <string.h>

#include <type_traits>
#include <utility>

#include "common.h"

namespace annlite {

struct ScanArgs {
    const void *codes;       // [N][M] code bytes
    const uint32_t *valid;   // optional bitmap
    const float *lut;        // tiled or BMK
    unsigned long long *partial;  // [B_pad][NS][k] keys
    int64_t N;
    int32_t Ks;
    int32_t B;
    int32_t k;
    int32_t n_tiles;
    int32_t n_slices;        // 1, 2, 4 or a multiple of 8
    int32_t n_items;         // work items (see item_map)
    int64_t slice_rows;      // multiple of 64
    const float *smax;       // [ceil16(B)] sum_m max_k |lut[b][m][k]|  (rounding slack of the fp32 sums)
    // quantised filter (qfilter kernel): 12-bit integer tables + the affine map back to distances
    const uint16_t *q16;     // [ceil16(B)/8][Ks][M][8] u16
    const float *qstep;      // [ceil16(B)]
    const double *qlo;       // [ceil16(B)] sum_m min_k lut[b][m][k]
    const float *qlom;       // [ceil16(B)][M] min_k lut[b][m][k] (byte-table kernel: it quantises the tables itself)
    unsigned long long *gkey; // [ceil16(B)] best k-th key any workgroup has proven for the query (device-scope
                             // atomic min; lets the 8+ row slices of a query tile share their progress)
    unsigned long long *gk2; // [ceil16(B)][n_slices] j-th key of every (query, slice) list, j = ceil(k/8): the 8
                             // concurrently scanned slices of a query hold >= k rows at or below the MAX of
                             // their j-th keys, a bound ~k/j times tighter than any single slice's own k-th
    int32_t jm1;             // j - 1
    const unsigned long long *gseed;  // [n_slices][n_tiles * 32] per-slice first bounds (byte-table kernel as the candidate
                             // generator of the re-rank stage: every slice keeps its OWN complete list, nothing is shared)
    // final merge inside the scan (shared mode): the LAST workgroup of a query tile to finish merges its slices
    unsigned int *tile_done; // [n_tiles] arrival counters, start at 0xffffffff (workspace fill); NULL = no in-kernel merge
    float *out_d;            // [B][k]   (or NULL with out_packed)
    int64_t *out_i;          // [B][k]
    int64_t *out_packed;     // [B][k][2] (global id, distance bits)
    int64_t row_base;
    int32_t sqrt_out;        // metric epilogue of EUCLIDEAN search (hnsw/index.py:164-165): out_d = sqrt(sum); never for packed
    int32_t flush_mask;      // a wave flushes its candidate queue every (flush_mask + 1) steps, staggered by wave
    int32_t dbg_skip;        // debug bitmask (ANNLITE_DEBUG_SKIP): 1 no gathers, 2 no insert/publish, 4 no event at all
    unsigned long long *dbg; // optional event counters (ANNLITE_DEBUG_COUNTERS=1): [0] slow-block entries,
                             // [1] (wave,query) events, [2] events with an insertion, [3] bound publications
    // tile mode (IVF cells, annlite_pq_search_tiles): every query tile scans its OWN row range, one work item per
    // tile (n_slices = 1); the tiles are handed out dynamically, longest first
    const int64_t *tile_rows;    // [n_tiles][2] (begin: multiple of 64, end); begin < 0: unused tile; NULL = slices
    int32_t tl_private;          // byte-table kernel in cell tiles (TL): 1 = the slots' lists are PRIVATE -- a query's bound is neither published to
                                 // nor imported from its other tiles (annlite_ivf_search_candidates: every list is a function of its own cell)
    const int32_t *vmap;         // [n_tiles * QT] >= 0: the query whose tables the slot scans with (q16 / smax / qstep
                                 // hold the REAL queries); < 0: padding slot (never passes the filter)
    unsigned int *item_counter;  // starts at 0xffffffff (workspace fill): next item = atomicAdd + 1
    uint32_t *cand;              // [n_tiles * QT][cand_cap] out: table rows of every slot that can be in its top-k
    uint32_t *cand_count;        // [n_tiles * QT] out: entries of the slot's list; 0xffffffff: it overflowed
    int32_t cand_cap;
    // byte-table kernel (scan_q8.hip): epochs end after steps e0, e0 * mul + (mul - 1), ...; ring_limit = candidates the
    // scanning waves may be ahead of the consumer wave
    int32_t q8_epoch0, q8_epoch_mul, q8_ring_limit, q8_import_mask;
    int32_t q8_target;                  // T of a slot right after its table is (re)built (<= 127)
    int32_t q8_thw_mask;                // the scanning waves pick up the workgroup's bounds every (mask + 1)-th step (1, 3 or 7)
    int32_t q8_rebuild_8ths;            // a slot wants a new table when its T has fallen below this many eighths of that
    // Kernel choice inside the library (scan.hip: byte-table launch with a guard, u16-table launch behind a gate):
    unsigned int *guard;                // byte-table kernel: [0] 0xffffffff = run, 0 = the launch gave up (its candidate rate
                                        // says the tables are not byte-table material: the gated u16 launch behind it redoes the
                                        // scan); [1] workgroups done; [2..3] u64 candidates seen (- 1).  All-ones from the fill.
    int32_t guard_abort;                // the launch may give up (a fallback launch is queued behind it)
    uint32_t guard_base;                // ... when a workgroup has seen more than guard_base + rows drawn / 32 candidates
    const unsigned int *gate;           // u16 kernels: run only if *gate == 0 (NULL: always)
    unsigned int *host_stats;           // host-mapped, optional: the last workgroup writes [1] gave up, [2..3] candidates, [4] B,
    uint32_t stats_seq;                 // [5] N (low 32 bits), then [0] = stats_seq
    // byte tables PREBUILT by the preparation launch (seed_bound_kernel<..., BUILD>; M = 16): the 8 workgroups of a query tile
    // copy their first table (128 KB, L2) instead of each converting the tile's 512 KB of fp32 tables
    const unsigned long long *gseed0;   // [ceil32(B)] the seed keys as the preparation launch left them (gkey moves on: atomic min)
    const uint8_t *btab;                // [n_tiles][kQ8Image16] LDS images (q8_entry16), quantised for gseed0; NULL: the workgroups build
    int32_t q8_early_merge;             // byte-table kernel: the FIRST workgroup of a tile to finish merges the others' lists as they
                                        // arrive (tile_done is then a bitmask of the slices still out); 0: the last one merges all
    uint32_t q8_merge_patience;         // ... and leaves (the last slice to arrive then merges all) after this many 100 MHz ticks
                                        // without an arrival (20000 = 200 us; ANNLITE_EARLY_MERGE_PATIENCE: tests force the path)
    int32_t q8_map_slices;              // byte-table kernel: work items mapped slice-per-XCD (item_map) instead of tile-per-XCD
                                        // (q8_item_map): an XCD streams ITS row slices once for all query tiles
    uint32_t q8_pos;                    // byte-table kernel, 16 < k <= 64 (64-key lists): the list positions p0 < p1 < p2 < p3 (one per byte) whose
                                        // keys a slice publishes -- cell i = "this slice holds p_i + 1 rows at or below this key"; G (p3 + 1) >= k
    int32_t q8_ilv_log;                 // byte-table kernel, shared bounds: > 0: the row slices are INTERLEAVED -- slice s owns the runs of 2^q8_ilv_log
                                        // blocks of 64 rows number s, s + n_slices, s + 2 n_slices, ... of the table instead of one contiguous
                                        // range (a table in cluster order has its queries' neighbourhoods in ONE range: that work item ran 1.5x the
                                        // others); 0: contiguous ranges of slice_rows rows
    unsigned long long *clk;            // optional (annlite_profile_enable): workgroup 0 leaves [0] shader cycles (s_memtime) and [1] 100 MHz
                                        // ticks at its start, [2] / [3] at its end -- the clock the kernel actually held
};

// work item -> (query tile, row slice).  item % 8 == blockIdx % 8 == the XCD the block lands on (speed
// only): with >= 8 slices an XCD owns the slices congruent to it and consecutive items of one XCD walk the
// tiles of the same slice (its L2 keeps the slice's rows); with fewer slices 8 / n_slices XCDs share one.
__device__ __forceinline__ bool item_map(const ScanArgs &a, int item, int &tile, int &slice) {
    const int xcd = item & 7, j = item >> 3;
    if (a.n_slices >= 8) {
        tile = j % a.n_tiles;
        slice = (j / a.n_tiles) * 8 + xcd;
        return true;
    }
    slice = xcd % a.n_slices;
    tile = j * (8 / a.n_slices) + xcd / a.n_slices;
    return tile < a.n_tiles;
}

// (code byte B of a dword) << SH in ONE VOP2-SDWA op (v_bfe_u32 + v_lshl_add_u32 are two 4.5-cycle VOP3 ops,
// scripts/valu_ubench.hip)
template <int BYTE>
__device__ __forceinline__ uint32_t byte_shl(uint32_t dword, uint32_t sh) {
    uint32_t r;
    if constexpr (BYTE == 0)
        asm("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_0" : "=v"(r) : "s"(sh), "v"(dword));
    else if constexpr (BYTE == 1)
        asm("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1" : "=v"(r) : "s"(sh), "v"(dword));
    else if constexpr (BYTE == 2)
        asm("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_2" : "=v"(r) : "s"(sh), "v"(dword));
    else
        asm("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_3" : "=v"(r) : "s"(sh), "v"(dword));
    return r;
}
// the four bytes of a dword, each << SH, in ONE asm statement: the compiler puts a hazard s_nop behind every
// asm statement it cannot look into -- 32 of them per step with one statement per byte
__device__ __forceinline__ void byte_shl4(uint32_t dword, uint32_t sh, uint32_t &r0, uint32_t &r1, uint32_t &r2,
                                          uint32_t &r3) {
    asm("v_lshlrev_b32_sdwa %0, %4, %5 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_0\n\t"
        "v_lshlrev_b32_sdwa %1, %4, %5 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1\n\t"
        "v_lshlrev_b32_sdwa %2, %4, %5 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_2\n\t"
        "v_lshlrev_b32_sdwa %3, %4, %5 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_3"
        : "=&v"(r0), "=&v"(r1), "=&v"(r2), "=&v"(r3)
        : "s"(sh), "v"(dword));
}
// both half-words of a dword, each << sh (SDWA word select): the address part of two uint16 codes
__device__ __forceinline__ void word_shl2(uint32_t x, uint32_t sh, uint32_t &o0, uint32_t &o1) {
    asm("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_0" : "=v"(o0) : "s"(sh), "v"(x));
    asm("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_1" : "=v"(o1) : "s"(sh), "v"(x));
}
constexpr int ilog2_c(int x) { return x <= 1 ? 0 : 1 + ilog2_c(x / 2); }

// ---- compile-time loops -------------------------------------------------------------------------
template <int I, int N, typename F>
__device__ __forceinline__ void static_for(F &&f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        static_for<I + 1, N>(f);
    }
}

// rotate the CW dwords of a code row left by `s` BYTES (s = 4*a + b, lane-varying but constant over
// the kernel): afterwards byte t of the row is the code of sub-space (s + t) mod M.
template <int CW>
__device__ __forceinline__ void rotate_row(uint32_t (&c)[CW], const bool (&abit)[8], uint32_t bsh) {
    // dword rotation by a, one conditional stage per bit of a
    int bit = 0;
    static_for<0, (CW > 1 ? (CW > 2 ? (CW > 4 ? (CW > 8 ? 4 : 3) : 2) : 1) : 0)>([&](auto ST) {
        constexpr int st = decltype(ST)::value;
        constexpr int sh = 1 << st;
        uint32_t n[CW];
#pragma unroll
        for (int i = 0; i < CW; ++i) n[i] = abit[st] ? c[(i + sh) % CW] : c[i];
#pragma unroll
        for (int i = 0; i < CW; ++i) c[i] = n[i];
    });
    (void)bit;
    // byte rotation by b across the dword ring
    uint32_t n[CW];
#pragma unroll
    for (int i = 0; i < CW; ++i) n[i] = __builtin_amdgcn_alignbyte(c[(i + 1) % CW], c[i], bsh);
#pragma unroll
    for (int i = 0; i < CW; ++i) c[i] = n[i];
}

// Extremes of a table column that may hold non-finite entries (a query with an inf / NaN coordinate; code words whose squared
// distance overflows).  The MINIMUM keeps -inf (a row with such an entry IS the nearest: L = -inf makes every bound "pass")
// and drops NaN (fminf).  The MAXIMUM -- range of the quantisation, rounding slack -- runs over the entries below +inf only:
// +inf and NaN entries quantise to the clip value, a valid lower bound of "beyond every number" (numpy's order puts NaN last),
// and a sum that contains one is non-finite whatever the rounding of the others.  Without this ONE overflowing code word
// made step and slack infinite, i.e. every row a candidate of every query.
__device__ __forceinline__ float col_max_arg(float v) { return v < __builtin_inff() ? v : -__builtin_inff(); }
__device__ __forceinline__ float finite_mag(float v) {
    const float a = __builtin_fabsf(v);
    return a < __builtin_inff() ? a : 0.f;
}

// integer filter bound (0x8000 | qthr) implied by a k-th key (see the kernel header for the derivation)
template <int M>
__device__ __forceinline__ unsigned short qbound_from_key(unsigned long long key, float smax_b, float qstep_b,
                                                          double qlo_b) {
    const uint32_t hi = (uint32_t)(key >> 32);
    if (hi == kKeyInfHi) return 0xffff;
    const double thr = (double)ordered_to_f32(hi);
    const double slack = (double)smax_b * (2.0 * M * 5.9604644775390625e-08 * (1.0 + 1.0 / 1024.0));
    double qd = (thr + slack - qlo_b) / (double)qstep_b;
    qd = __builtin_floor(qd) + 1.0;  // qthr
    if (!(qd < 32767.0)) qd = 32767.0;  // (a NaN -- non-finite threshold, slack or L -- lands HERE: everything passes)
    else if (!(qd > 0.0)) qd = 0.0;
    return (unsigned short)(0x8000u | (uint32_t)qd);
}


// ---- M = 64: the "wrap-coded" SKEWED layout ------------------------------------------------------
// With 64 sub-spaces a lane cannot afford one LDS base pointer per step (64 VGPRs), and the look-up address should cost
// ONE instruction.  The LDS table is two half tables of 32 sub-spaces ([257 rows][32 columns][8 B], 256-byte rows,
// the second one 0x10100 bytes behind the first), and a row is stored as two independently skewed halves: lane l (row n,
// n % 32 == l % 32) reads, at step t = 32 h + p, sub-space 32 h + (l % 32 + p) % 32.  Its address is
//     (stored byte << 8) | (l % 32) * 8 [| 0x10000 for h = 1]   -- one v_perm_b32 of the code dword and a lane constant --
// plus the instruction's immediate p * 8 [+ 0x100 for h = 1]: correct while l % 32 + p < 32.  Past that the true entry
// is 256 bytes lower, i.e. the SAME offset inside the PREVIOUS table row: the stored byte of a wrapped position is
// code - 1 (mod 256), and each half table carries one extra row 256 = copy of row 0 for code 0 - 1 = 255.
// stored byte j = 32 h + p of row n:  code[32 h + (p + n % 32) % 32] - [p + n % 32 >= 32]   (mod 256)
// 0x01 in every byte of dword w (positions 4w..4w+3) whose position p = j % 32 satisfies p + r >= 32  (r = n % 32)
__device__ __forceinline__ uint32_t wrap64_mask(int w, int r) {
    int nb = 4 * (w & 7) + 4 - (32 - r);  // number of (upper) bytes of the dword that wrap
    nb = nb < 0 ? 0 : (nb > 4 ? 4 : nb);
    return nb == 0 ? 0u : (0x01010101u << (8 * (4 - nb)));
}
// per-byte x - y / x + y (mod 256) for y in {0, 1} per byte, no borrow/carry across bytes
__device__ __forceinline__ uint32_t bytes_sub(uint32_t x, uint32_t y) {
    return ((x | 0x80808080u) - y) ^ ((x ^ ~y) & 0x80808080u);
}
__device__ __forceinline__ uint32_t bytes_add(uint32_t x, uint32_t y) {
    return ((x & 0x7f7f7f7fu) + y) ^ ((x ^ y) & 0x80808080u);
}


// M = 64 row in registers: PLAIN -> this row's stored form (r = n % 32), and back (ascending sub-space order)
__device__ __forceinline__ void skew64_rotate_halves(uint32_t (&c)[16], int s) {  // both halves left by s bytes (0..31)
    bool ab[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) ab[i] = (((s >> 2) >> i) & 1) != 0;
    uint32_t lo[8], hi[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) lo[i] = c[i], hi[i] = c[8 + i];
    rotate_row<8>(lo, ab, (uint32_t)(s & 3));
    rotate_row<8>(hi, ab, (uint32_t)(s & 3));
#pragma unroll
    for (int i = 0; i < 8; ++i) c[i] = lo[i], c[8 + i] = hi[i];
}
__device__ __forceinline__ void skew64_encode(uint32_t (&c)[16], int r) {
    skew64_rotate_halves(c, r);
#pragma unroll
    for (int i = 0; i < 16; ++i) c[i] = bytes_sub(c[i], wrap64_mask(i, r));
}
__device__ __forceinline__ void skew64_decode(uint32_t (&c)[16], int r) {
#pragma unroll
    for (int i = 0; i < 16; ++i) c[i] = bytes_add(c[i], wrap64_mask(i, r));
    skew64_rotate_halves(c, (32 - r) % 32);
}


// gk2: kGk2Keys u64 per (query, row slice).  The u16 kernels store one key there, the slice's j-th (j = ceil(k / G), G = the
// slices scanned concurrently); the byte-table kernel stores its j smallest keys (j <= kGk2Keys) -- see import_bounds.
#ifndef ANNLITE_GK2_KEYS
#define ANNLITE_GK2_KEYS 4
#endif
constexpr int kGk2Keys = ANNLITE_GK2_KEYS;
// (M = 64 keeps one key per cell: its kernel does not use the union rule, and the 4x larger array in front of its tables
// measured 5 % on the config-4 shape -- 0.95 against 0.90 ms per launch)
__host__ __device__ constexpr int gk2_cell_keys(int64_t M) { return M == 64 ? 1 : kGk2Keys; }

// ---- byte-table kernel (scan_q8.hip): shape constants and slot parameters, shared with the preparation launch (scan_prep.hip) ----
// Q8Cfg: entries are clipped at QMAX; a slot without a bound yet ("open": nothing seeded it) clips at QOPEN so that its T
// passes every row.  M <= 32: the byte sums ARE the filter sums (M * QMAX <= 240: a byte sum never carries), bounds are bytes
// (0x80 | T, T <= 127), 32 queries per workgroup.  M = 64 (WIDE): 8 queries per 8-byte entry, the byte sums of 16 look-ups
// (16 * 15 = 240) are widened into u16 sums four times per row, bounds are half-words (0x8000 | T), 8 queries per workgroup.
template <int M>
struct Q8Cfg {
    static constexpr bool WIDE = M == 64;
    static constexpr bool M8 = M == 8;  // M = 8: table [Ks][NQ entry groups][8 sub-spaces][16 B], permute addressing (see the kernel)
    // M = 32: ONE entry group (16 queries per workgroup: 32 sub-spaces x 16 B x 256 codes fill the LDS), two half tables of 16
    // sub-spaces, [256][16][16 B] each, the second 64 KB behind the first: look-up address = (half << 16) | (code << 8) | column,
    // ONE v_perm_b32 of the code dword with a lane constant (see the kernel)
    static constexpr bool M32 = M == 32;
    static constexpr int QMAX = WIDE ? 15 : 240 / M, QOPEN = WIDE ? 7 : 112 / M;
    static constexpr uint32_t TMAX = WIDE ? 32767u : 127u, TFLAG = TMAX + 1u;
};
// M = 16 (round 6): the LDS image of a tile's byte table -- TWO half tables by sub-space PARITY, 64 KB apart; a code's row of a half
// is 256 bytes = 8 sub-spaces x 2 entry groups x 16 B, both groups of a (code, sub-space) ADJACENT.  Entry (code, m, group g):
//     (p << 16) + (code << 8) + ((2 (m >> 1) + g + p) << 4),   p = m & 1
// so that a look-up address is ONE v_perm_b32 of the code dword with a lane constant (byte 0: the slot, byte 2: the half) and the
// second entry group is the first + 16 -- an immediate: one address instruction per TWO look-ups where the [Ks][2][16][16 B] layout
// needed a byte shift (code x 512 is not a byte move) and an add per look-up address.  The odd half is shifted by ONE slot (its last
// entry spills into the next code's row: 257 rows): the 16 lanes of a ds_read_b128 -- 16 different sub-spaces, the skew -- then hit
// 16 different 16-byte bank groups in both reads.  scripts/ubench/step_loop.hip ADDR=3: 0.883 -> 0.923 of the look-up roof.
constexpr int kQ8Image16 = 131072 + 256;
__host__ __device__ constexpr uint32_t q8_entry16(uint32_t code, uint32_t m, uint32_t g) {
    return ((m & 1u) << 16) + (code << 8) + ((2u * (m >> 1) + g + (m & 1u)) << 4);
}
// NQ = entry groups of 16 queries per workgroup (the second shape parameter): 2 everywhere but M = 8 with 512 < Ks <= 1024,
// where only one group's table fits the LDS (WIDE: 8 queries whatever NQ says)
template <int M, int NQ>
constexpr int q8_qt() { return Q8Cfg<M>::WIDE ? 8 : 16 * NQ; }
template <int M, int NQ>
__device__ __forceinline__ int q8_table_bytes(int Ks) {
    if (Q8Cfg<M>::M32) return 131072;  // two half tables [256][16][16 B] whatever Ks (<= 256) is: the halves' distance is an address bit
    if (M == 16 && NQ == 2) return kQ8Image16;  // two half tables by sub-space parity (q8_entry16), whatever Ks is
    return Q8Cfg<M>::WIDE ? (Ks + 1) * 512 : Ks * NQ * M * 16;  // WIDE: two half tables of 32 sub-spaces, [Ks + 1][32][8 B] each
}

// filter bound (TFLAG | T) implied by a k-th key for a table quantised with `step`:
// T = floor((thr + slack32 - L) / step * (1 + 2^-19)) + 1, clamped to TMAX (a NaN lands there too: everything passes)
template <int M>
__device__ __forceinline__ uint32_t q8_bound_from_key(unsigned long long key, float smax_b, float step, double qlo_b) {
    const uint32_t hi = (uint32_t)(key >> 32);
    if (hi == kKeyInfHi) return 2u * Q8Cfg<M>::TFLAG - 1u;
    const double thr = (double)ordered_to_f32(hi);
    const double slack = (double)smax_b * (2.0 * M * 5.9604644775390625e-08 * (1.0 + 1.0 / 1024.0));
    double qd = (thr + slack - qlo_b) / (double)step * (1.0 + 1.0 / 524288.0);
    qd = __builtin_floor(qd) + 1.0;
    if (!(qd < (double)Q8Cfg<M>::TMAX)) qd = (double)Q8Cfg<M>::TMAX;  // (a NaN lands HERE: everything passes)
    else if (!(qd > 0.0)) qd = 0.0;
    return Q8Cfg<M>::TFLAG | (uint32_t)qd;
}

template <int M>
__device__ __forceinline__ void q8_slot_params(bool real, unsigned long long key, float range, float smax_b, double L, int target,
                                               float &step, float &inv, float &clip, uint32_t &tbits) {
    clip = (float)Q8Cfg<M>::QOPEN;
    if (!real) {  // pad slot: all-zero table, never passes (TMAX - 0 has the flag bit clear)
        step = 1.f;
        inv = 0.f;
        tbits = Q8Cfg<M>::TMAX;
        return;
    }
    float open_step = range / (float)Q8Cfg<M>::QOPEN;  // no bound yet: the whole range, everything passes (S <= M * QOPEN <= TMAX)
    if (!(open_step > 1e-30f) || !(open_step < 1e30f)) open_step = 1.f;
    step = open_step;
    const uint32_t hi = (uint32_t)(key >> 32);
    if (hi != kKeyInfHi) {
        const double thr = (double)ordered_to_f32(hi);
        const double slack = (double)smax_b * (2.0 * M * 5.9604644775390625e-08 * (1.0 + 1.0 / 1024.0));
        const double R = thr + slack - L;
        if (R > 0.0 && R < 1e30) {
            float s = (float)(R / (double)(target - 1));  // T = target right after a (re)build
            const float smin = open_step * (1.f / 65536.f);
            if (!(s >= smin)) s = smin;  // (a larger step only lowers T)
            step = s;
            clip = (float)Q8Cfg<M>::QMAX;  // T <= target now and it only falls
        }
    }
    inv = 1.0f / step;
    tbits = q8_bound_from_key<M>(key, smax_b, step, L);
}


// ---- launchers of the scan kernels, one translation unit per kernel family -----------------------
// (scan_q8.hip: byte filter tables; scan_qfilter.hip: u16 filter tables, tile mode; scan_prep.hip: table build /
// quantisation parameters / seed bound)
struct LutBuild {  // annlite_pq_search_topk: the L2 tables are built by the quantisation launch itself
    const float *queries;
    const float *codebooks;
    int64_t D;
};
int launch_qfilter_scan(int id, bool skewed, const ScanArgs &a, int grid, hipStream_t st);
int launch_q8_scan(int id, bool skewed, const ScanArgs &a, int grid, hipStream_t st);
// q16 == NULL: only the per-query parameters (step, L, Smax, minima) are produced; qlom may be NULL
int launch_lut_quantise(int64_t M, int64_t Ks, int64_t B, int64_t bpad, const float *lut_dev, const LutBuild *build,
                        uint16_t *q16, float *qstep, double *qlo, float *smax, float *qlom, void *fill,
                        size_t fill_bytes, hipStream_t st, const unsigned int *gate = nullptr);
// n_seed_slices > 1: one bound per (query, row slice) -- rows [y * seed_stride, + S) of slice y, bound in gkey[y * gkey_stride + b]
int launch_seed_bound(int64_t M, bool skewed, const void *codes_dev, int code_bytes, int64_t S, const uint32_t *valid_bits_dev,
                      const float *lut_dev, int64_t B, int64_t Ks, int64_t k, const float *smax, unsigned long long *gkey,
                      hipStream_t st, int64_t N = 0, int n_seed_slices = 1, int64_t seed_stride = 0, int64_t gkey_stride = 0,
                      const unsigned int *gate = nullptr);

// byte-table plan: table build + quantisation parameters + workspace reset + seed bound in one launch (scan_prep.hip)
// gseed0 / btab (optional, both or none): the seed keys kept aside and the byte tables of every query tile quantised for them
// with `target` (ScanArgs::q8_target); dbg (optional): 8 phase stamps of the first and the last workgroup
int launch_seed_build(bool skewed, const void *codes_dev, int64_t S, int64_t N, const uint32_t *valid_bits_dev, const LutBuild &build,
                      float *lut_out, int64_t B, int64_t Ks, int64_t k, float *qstep, double *qlo, float *smax, float *qlom,
                      unsigned long long *gkey, void *fill, size_t fill_bytes, size_t gkey_bytes, hipStream_t st,
                      unsigned long long *gseed0 = nullptr, uint8_t *btab = nullptr, int target = 0,
                      unsigned long long *dbg = nullptr, unsigned long long *seedk = nullptr, const uint32_t *cand = nullptr,
                      int n_cand = 0);
// pruned search over cells (annlite_ivf_search_topk): the same launch with per-query seed rows (the query's nearest cell) and the
// byte tables per QUERY (bq [ceil16(B)][Ks][16], quantised for gseed0) instead of per tile
// the cell tiles' per-slot lists as ids, unmerged (ivf.hip): out[b][p * k + j] = id_base + row_ids[row] of key j of pair (b, p)'s list, -1 = none
int launch_ivf_lists_to_ids(const unsigned long long *lists, int64_t k, const int32_t *slot_of, int64_t B, int64_t P, const int64_t *row_ids,
                            int64_t id_base, int64_t *out_ids, hipStream_t st);
// item_counter (optional): reset to 0 for the scan behind the launch
int launch_seed_build_cells(bool skewed, const void *codes_dev, int64_t S, int64_t N, const uint32_t *valid_bits_dev,
                            const LutBuild &build, float *lut_out, int64_t B, int64_t Ks, int64_t k, float *qstep, double *qlo,
                            float *smax, float *qlom, unsigned long long *gkey, hipStream_t st, unsigned long long *gseed0,
                            uint8_t *bq, int target, const int32_t *cells, int64_t n_probe, const int64_t *cell_rows,
                            unsigned int *item_counter = nullptr, bool ip_tables = false, const int32_t *seed_cells = nullptr);
// seedk (optional): [B][kSeedKeys] the bounds implied by the seed's k smallest rows (annlite_pq_search_split)
constexpr int kSeedKeys = 16;
// seed_mfma.hip (round 6): per query the best row -- by a bf16 MFMA approximation of the ADC sum -- of each of kSeedCand disjoint
// groups of S / kSeedCand seed rows; the preparation launch takes its bound from the nominees' exact sums instead of scanning the
// S rows itself (launch_seed_build: cand)
constexpr int kSeedCand = 512;
int launch_seed_mfma(bool skewed, const float *queries_dev, int64_t B, const float *cb_dev, int64_t Ks, const void *codes_dev,
                     const uint32_t *valid_bits_dev, int64_t N, int64_t S, int chunk_log, uint32_t *cand_dev, hipStream_t st);

}  // namespace annlite
