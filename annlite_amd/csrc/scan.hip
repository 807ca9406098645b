// scan.hip -- batched asymmetric-distance (ADC) scan over a PQ code table + exact top-k.
//
// Reference semantics (jina-ai/annlite v0.5.11):
//   for each query b:  d[n] = sum_{m=0..M-1, ascending, fp32} lut[b][m][codes[n][m]]
//                      (bindings/pq_bindings.pyx:30-47,52-80 == include/hnswlib/space_pq.h:15-37)
//   then the k smallest (annlite/math.py:94-120) -- with the build's fixed tie-break (d asc, n asc).
//
// MI355X design (DESIGN.md "ADC scan"):
//   * The cost is B*N*M random 4-byte table look-ups, so the kernel is bound by LDS gather
//     bandwidth, not HBM.  Each workgroup keeps the tables of QT queries in LDS (M*Ks*4 B per query,
//     up to 128 KB of the CU's 160 KB) interleaved QI queries per entry ([k][h][m][QI]), so ONE
//     ds_read_b128 (QI=4) / ds_read_b64 (QI=2) returns the entries of QI queries for one code byte.
//   * Conflict-free gather by SKEWING the sub-space order across lanes: at step t lane l looks up
//     sub-space m = (l + t) mod M.  With the [k][h][m][QI] layout the LDS bank slot of an entry is
//     m mod 16 (b128) / m mod 32 (b64), so the lanes of every hardware conflict group hit distinct
//     slots whatever the code bytes are.  A non-skewed gather (all lanes same m, random k) costs
//     ~2.9x (max load of 16 balls in 16 bins).
//   * Bit-exact sums despite the skew: a lane holds its M looked-up values in VGPRs and adds them
//     in true ascending-m order with two exec-masked passes of v_pk_add_f32 (pass 1: steps t>=t0
//     = sub-spaces 0..s-1, pass 2: steps t<t0 = sub-spaces s..M-1; t0 = (M-s) mod M).  The exec
//     masks are compile-time constants because s = lane mod M.
//   * Codes stream from HBM once per XCD: the table is cut into >= 8 row slices, slice -> XCD by
//     blockIdx % 8, and the 32 workgroups of an XCD walk the SAME slice for different query tiles,
//     so all but the first reader hit the XCD's 4 MB L2.
//   * top-k without LDS: every wave keeps, per query, a sorted 64-entry list spread over its lanes
//     (common.h WaveList); a row is offered only if it beats the wave's current k-th distance.
//
// No fallback to CPU exists; unsupported shapes use the generic kernel below (LUT through L2).
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
    const float *smax;       // [ceil16(B)] sum_m max_k |lut[b][m][k]|  (filter kernel: rounding slack)
    // quantised filter (qfilter kernel): 12-bit integer tables + the affine map back to distances
    const uint16_t *q16;     // [ceil16(B)/8][Ks][M][8] u16
    const float *qstep;      // [ceil16(B)]
    const double *qlo;       // [ceil16(B)] sum_m min_k lut[b][m][k]
    unsigned long long *gkey; // [ceil16(B)] best k-th key any workgroup has proven for the query (device-scope
                             // atomic min; lets the 8+ row slices of a query tile share their progress)
    unsigned long long *gk2; // [ceil16(B)][n_slices] j-th key of every (query, slice) list, j = ceil(k/8): the 8
                             // concurrently scanned slices of a query hold >= k rows at or below the MAX of
                             // their j-th keys, a bound ~k/j times tighter than any single slice's own k-th
    int32_t jm1;             // j - 1
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

// ---- compile-time exec masks for the ordered accumulation ---------------------------------------
template <int M>
constexpr unsigned long long pass_mask(int t, int pass) {
    unsigned long long m = 0;
    for (int l = 0; l < 64; ++l) {
        const int s = l % M;
        const bool p1 = (s == 0) || (s >= M - t);
        if (pass == 0 ? p1 : !p1) m |= 1ull << l;
    }
    return m;
}

template <unsigned long long MASK>
__device__ __forceinline__ void masked_pk_add2(f32x2 &a0, f32x2 &a1, const f32x2 x0, const f32x2 x1) {
    if constexpr (MASK == 0ull) {
        return;
    } else if constexpr (MASK == ~0ull) {
        a0 += x0;
        a1 += x1;
    } else {
        unsigned long long sv;
        asm("s_mov_b64 %[sv], exec\n\t"
            "s_mov_b32 exec_lo, %[lo]\n\t"
            "s_mov_b32 exec_hi, %[hi]\n\t"
            "v_pk_add_f32 %[a0], %[a0], %[x0]\n\t"
            "v_pk_add_f32 %[a1], %[a1], %[x1]\n\t"
            "s_mov_b64 exec, %[sv]"
            : [a0] "+v"(a0), [a1] "+v"(a1), [sv] "=&s"(sv)
            : [x0] "v"(x0), [x1] "v"(x1), [lo] "i"((int)(uint32_t)(MASK & 0xffffffffull)),
              [hi] "i"((int)(uint32_t)(MASK >> 32)));
    }
}

template <unsigned long long MASK>
__device__ __forceinline__ void masked_pk_add1(f32x2 &a0, const f32x2 x0) {
    if constexpr (MASK == 0ull) {
        return;
    } else if constexpr (MASK == ~0ull) {
        a0 += x0;
    } else {
        unsigned long long sv;
        asm("s_mov_b64 %[sv], exec\n\t"
            "s_mov_b32 exec_lo, %[lo]\n\t"
            "s_mov_b32 exec_hi, %[hi]\n\t"
            "v_pk_add_f32 %[a0], %[a0], %[x0]\n\t"
            "s_mov_b64 exec, %[sv]"
            : [a0] "+v"(a0), [sv] "=&s"(sv)
            : [x0] "v"(x0), [lo] "i"((int)(uint32_t)(MASK & 0xffffffffull)),
              [hi] "i"((int)(uint32_t)(MASK >> 32)));
    }
}


// ---- ordered accumulation, 8 steps per asm statement --------------------------------------------
// One statement = 8 x { set exec to the compile-time lane mask of step t ; v_pk_add_f32 ... } and
// ONE restore of exec to all-ones (the main loop runs with full waves and uniform control flow).
// v1 of this kernel saved/restored exec around every step (4 SALU per 2 VALU): rocprof showed
// 343 SALU + 255 VALU per wave-step and the LDS pipe only 21 % busy (profiles/r01_*).
#define ANNLITE_MASK_LO(M, T, P) ((int)(uint32_t)(pass_mask<M>((T), (P)) & 0xffffffffull))
#define ANNLITE_MASK_HI(M, T, P) ((int)(uint32_t)(pass_mask<M>((T), (P)) >> 32))
#define ANNLITE_LOHALF(v) __builtin_shufflevector((v), (v), 0, 1)
#define ANNLITE_HIHALF(v) __builtin_shufflevector((v), (v), 2, 3)

#define ANNLITE_STEP_Q4(i)                                       \
    "s_mov_b32 exec_lo, %[m" #i "]\n\t"                          \
    "s_mov_b32 exec_hi, %[m" #i "]\n\t"                          \
    "v_pk_add_f32 %[a0], %[a0], %[x" #i "]\n\t"                  \
    "v_pk_add_f32 %[a1], %[a1], %[y" #i "]\n\t"

template <int M, int T0, int PASS>
__device__ __forceinline__ void pass8_q4(f32x2 &a0, f32x2 &a1, const f32x4 (&v)[M]) {
    static_assert(M <= 32 && T0 + 8 <= M, "lo == hi masks need a lane period <= 32");
    asm(ANNLITE_STEP_Q4(0) ANNLITE_STEP_Q4(1) ANNLITE_STEP_Q4(2) ANNLITE_STEP_Q4(3)
        ANNLITE_STEP_Q4(4) ANNLITE_STEP_Q4(5) ANNLITE_STEP_Q4(6) ANNLITE_STEP_Q4(7)
        "s_mov_b64 exec, -1"
        : [a0] "+v"(a0), [a1] "+v"(a1)
        : [x0] "v"(ANNLITE_LOHALF(v[T0 + 0])), [y0] "v"(ANNLITE_HIHALF(v[T0 + 0])),
          [x1] "v"(ANNLITE_LOHALF(v[T0 + 1])), [y1] "v"(ANNLITE_HIHALF(v[T0 + 1])),
          [x2] "v"(ANNLITE_LOHALF(v[T0 + 2])), [y2] "v"(ANNLITE_HIHALF(v[T0 + 2])),
          [x3] "v"(ANNLITE_LOHALF(v[T0 + 3])), [y3] "v"(ANNLITE_HIHALF(v[T0 + 3])),
          [x4] "v"(ANNLITE_LOHALF(v[T0 + 4])), [y4] "v"(ANNLITE_HIHALF(v[T0 + 4])),
          [x5] "v"(ANNLITE_LOHALF(v[T0 + 5])), [y5] "v"(ANNLITE_HIHALF(v[T0 + 5])),
          [x6] "v"(ANNLITE_LOHALF(v[T0 + 6])), [y6] "v"(ANNLITE_HIHALF(v[T0 + 6])),
          [x7] "v"(ANNLITE_LOHALF(v[T0 + 7])), [y7] "v"(ANNLITE_HIHALF(v[T0 + 7])),
          [m0] "i"(ANNLITE_MASK_LO(M, T0 + 0, PASS)), [m1] "i"(ANNLITE_MASK_LO(M, T0 + 1, PASS)),
          [m2] "i"(ANNLITE_MASK_LO(M, T0 + 2, PASS)), [m3] "i"(ANNLITE_MASK_LO(M, T0 + 3, PASS)),
          [m4] "i"(ANNLITE_MASK_LO(M, T0 + 4, PASS)), [m5] "i"(ANNLITE_MASK_LO(M, T0 + 5, PASS)),
          [m6] "i"(ANNLITE_MASK_LO(M, T0 + 6, PASS)), [m7] "i"(ANNLITE_MASK_LO(M, T0 + 7, PASS)));
}

#define ANNLITE_STEP_Q2(i)                                       \
    "s_mov_b32 exec_lo, %[l" #i "]\n\t"                          \
    "s_mov_b32 exec_hi, %[h" #i "]\n\t"                          \
    "v_pk_add_f32 %[a0], %[a0], %[x" #i "]\n\t"

template <int M, int T0, int PASS>
__device__ __forceinline__ void pass8_q2(f32x2 &a0, const f32x2 (&v)[M]) {
    static_assert(T0 + 8 <= M, "block out of range");
    asm(ANNLITE_STEP_Q2(0) ANNLITE_STEP_Q2(1) ANNLITE_STEP_Q2(2) ANNLITE_STEP_Q2(3)
        ANNLITE_STEP_Q2(4) ANNLITE_STEP_Q2(5) ANNLITE_STEP_Q2(6) ANNLITE_STEP_Q2(7)
        "s_mov_b64 exec, -1"
        : [a0] "+v"(a0)
        : [x0] "v"(v[T0 + 0]), [x1] "v"(v[T0 + 1]), [x2] "v"(v[T0 + 2]), [x3] "v"(v[T0 + 3]),
          [x4] "v"(v[T0 + 4]), [x5] "v"(v[T0 + 5]), [x6] "v"(v[T0 + 6]), [x7] "v"(v[T0 + 7]),
          [l0] "i"(ANNLITE_MASK_LO(M, T0 + 0, PASS)), [h0] "i"(ANNLITE_MASK_HI(M, T0 + 0, PASS)),
          [l1] "i"(ANNLITE_MASK_LO(M, T0 + 1, PASS)), [h1] "i"(ANNLITE_MASK_HI(M, T0 + 1, PASS)),
          [l2] "i"(ANNLITE_MASK_LO(M, T0 + 2, PASS)), [h2] "i"(ANNLITE_MASK_HI(M, T0 + 2, PASS)),
          [l3] "i"(ANNLITE_MASK_LO(M, T0 + 3, PASS)), [h3] "i"(ANNLITE_MASK_HI(M, T0 + 3, PASS)),
          [l4] "i"(ANNLITE_MASK_LO(M, T0 + 4, PASS)), [h4] "i"(ANNLITE_MASK_HI(M, T0 + 4, PASS)),
          [l5] "i"(ANNLITE_MASK_LO(M, T0 + 5, PASS)), [h5] "i"(ANNLITE_MASK_HI(M, T0 + 5, PASS)),
          [l6] "i"(ANNLITE_MASK_LO(M, T0 + 6, PASS)), [h6] "i"(ANNLITE_MASK_HI(M, T0 + 6, PASS)),
          [l7] "i"(ANNLITE_MASK_LO(M, T0 + 7, PASS)), [h7] "i"(ANNLITE_MASK_HI(M, T0 + 7, PASS)));
}


// ---- ordered accumulation without touching EXEC: per-lane 0/1 weights -----------------------------
// fma(v, 1.0f, acc) == acc + v (one rounding, identical to v_add_f32) and fma(v, 0.0f, acc) == acc for
// finite v, so "lane masked out" becomes "weight 0".  w[t] = (w1, w2) per lane: w1 = 1 if step t
// belongs to pass 1 for this lane (t >= t0) else 0, w2 = 1 - w1.  op_sel/op_sel_hi broadcast w1
// (pass 1) or w2 (pass 2) to both halves of the packed op.  No SALU at all: the exec-mask version
// was bound by the CU's single scalar unit (~180 SALU per 8192 look-ups, profiles/r01 notes).
__device__ __forceinline__ void wfma_p1(f32x2 &acc, const f32x2 v, const f32x2 w) {
    asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[1,0,1]" : "+v"(acc) : "v"(v), "v"(w));
}
__device__ __forceinline__ void wfma_p2(f32x2 &acc, const f32x2 v, const f32x2 w) {
    asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,1,0] op_sel_hi:[1,1,1]" : "+v"(acc) : "v"(v), "v"(w));
}
// MODE 2: the same with scalar (non-packed) v_fma_f32 -- A/B against the packed form
__device__ __forceinline__ void sfma(f32x2 &acc, const f32x2 v, const float w) {
    float ax = acc.x, ay = acc.y;
    asm("v_fma_f32 %0, %1, %2, %0" : "+v"(ax) : "v"(v.x), "v"(w));
    asm("v_fma_f32 %0, %1, %2, %0" : "+v"(ay) : "v"(v.y), "v"(w));
    acc.x = ax;
    acc.y = ay;
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

template <int QI>
struct LutVec;
template <>
struct LutVec<4> {
    typedef f32x4 type;
};
template <>
struct LutVec<2> {
    typedef f32x2 type;
};

// =================================================================================================
// Fast kernel: uint8 codes, Ks <= 256, M in {8,16,32,64}, k <= 64.
//   M  sub-spaces            QI queries interleaved per LDS entry (4 -> ds_read_b128, 2 -> b64)
//   NQ entry groups per WG   (QT = QI*NQ queries per workgroup)      NW waves per workgroup
// LDS byte address of (code k, group h, sub-space m): ((k*NQ + h)*M + m) * QI*4
// =================================================================================================
template <int M, int QI, int NQ, int NW, int WPS, bool SKEWED, int MODE>
__global__ __launch_bounds__(NW * 64, WPS) void adc_scan_fast_kernel(const ScanArgs a) {
    constexpr int QT = QI * NQ;
    constexpr int CW = M / 4;              // dwords per code row
    constexpr int EB = QI * 4;             // bytes per LDS entry
    constexpr int RB = M * EB;             // bytes per (k, h) row of entries
    constexpr int KSTRIDE = NQ * RB;       // bytes between consecutive codes k
    constexpr int NP = QI / 2;             // f32x2 pairs per entry
    typedef typename LutVec<QI>::type lutv_t;

    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int km1 = a.k - 1;

    // lane-constant skew
    const int s = lane % M;
    const uint32_t bsh = (uint32_t)(s & 3);
    bool abit[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) abit[i] = (((s >> 2) >> i) & 1) != 0;
    // byte offset inside a (k,h) row for step t: ((s+t) mod M) * EB
    const unsigned char *mbase[M];
#pragma unroll
    for (int t = 0; t < M; ++t) mbase[t] = smem + ((s + t) % M) * EB;
    // MODE 1: per-lane pass weights (w1, w2) for every step
    f32x2 wt[MODE >= 1 ? M : 1];
    if constexpr (MODE >= 1) {
#pragma unroll
        for (int t = 0; t < M; ++t) {
            const bool p1 = (s == 0) || (s >= M - t);
            wt[t] = (f32x2){p1 ? 1.f : 0.f, p1 ? 0.f : 1.f};
        }
    }

    const int n_items = a.n_items;
    const int64_t group_bytes = (int64_t)a.Ks * RB;  // one tiled-LUT group = [Ks][M][QI] floats

    for (int item = blockIdx.x; item < n_items; item += gridDim.x) {
        // item -> (slice, tile): slice % 8 == item % 8 == blockIdx % 8 (the XCD this block lands on,
        // speed only), consecutive items of one XCD walk the tiles of the same slice.
        int tile, slice;
        if (!item_map(a, item, tile, slice)) continue;

        __syncthreads();  // previous item's LDS readers are done
        {
            // fill the LUT tile: NQ groups of [Ks][M][QI] -> LDS [k][h][m][QI]; 16 B pieces
            const unsigned char *src0 = (const unsigned char *)a.lut + (int64_t)tile * NQ * group_bytes;
            constexpr int PIECES_PER_ROW = RB / 16;
            const int total = NQ * a.Ks * PIECES_PER_ROW;
            for (int idx = tid; idx < total; idx += NW * 64) {
                const int p = idx % PIECES_PER_ROW;
                const int kh = idx / PIECES_PER_ROW;  // = h*Ks + k  (source order)
                const int h = kh / a.Ks;
                const int kk = kh - h * a.Ks;
                const u32x4 v = *(const u32x4 *)(src0 + (int64_t)h * group_bytes + (int64_t)kk * RB + p * 16);
                *(u32x4 *)(smem + (kk * NQ + h) * RB + p * 16) = v;
            }
        }
        __syncthreads();

        WaveList list[QT];
        uint32_t thr_hi[QT], thr_lo[QT];
        float thr_f[QT];
#pragma unroll
        for (int q = 0; q < QT; ++q) {
            list[q].reset();
            thr_hi[q] = kKeyInfHi;
            thr_lo[q] = kIdNone;
            thr_f[q] = __builtin_inff();
        }

        const int64_t slice_begin = (int64_t)slice * a.slice_rows;
        int64_t slice_end = slice_begin + a.slice_rows;
        if (slice_end > a.N) slice_end = a.N;

        const uint32_t *codes32 = (const uint32_t *)a.codes;
        auto load_row = [&](int64_t row, uint32_t (&c)[CW]) {
            if (row >= a.N) row = a.N - 1;  // clamped, masked out below
            const uint32_t *p = codes32 + row * CW;
            if constexpr (CW == 2) {
                const u32x2 v = *(const u32x2 *)p;
                c[0] = v.x;
                c[1] = v.y;
            } else {
#pragma unroll
                for (int i = 0; i < CW / 4; ++i) {
                    const u32x4 v = *(const u32x4 *)(p + 4 * i);
                    c[4 * i + 0] = v.x;
                    c[4 * i + 1] = v.y;
                    c[4 * i + 2] = v.z;
                    c[4 * i + 3] = v.w;
                }
            }
        };

        // ---- software-pipelined row loop ---------------------------------------------------------
        // A wave's look-ups of quad h+1 (or of the NEXT row's quad 0) are issued chunk by chunk (8 steps)
        // as soon as pass 2 has consumed that chunk of the current quad, so the LDS latency of one chunk
        // hides behind the adds of the other(s) inside the same 16 (M) value registers.
        constexpr int NCH = M / 8;
        const int64_t stride = (int64_t)NW * 64;
        int64_t row0 = slice_begin + (int64_t)wave * 64;
        uint32_t cnext[CW];
        const unsigned char *addr[M];  // LDS pointers (32-bit): smem + code*KSTRIDE + moff[t]
        lutv_t val[M];
        auto make_addr = [&](uint32_t (&cc)[CW]) {
            if constexpr (!SKEWED) rotate_row<CW>(cc, abit, bsh);  // SKEWED tables are stored pre-rotated
            static_for<0, M>([&](auto T) {
                constexpr int t = decltype(T)::value;
                const uint32_t code = __builtin_amdgcn_ubfe(cc[t / 4], 8 * (t % 4), 8);
                addr[t] = mbase[t] + code * (uint32_t)KSTRIDE;
            });
        };
        auto issue_chunk = [&](auto C, auto H) {
            constexpr int c8 = decltype(C)::value * 8;
            constexpr int hoff = decltype(H)::value * RB;
            static_for<0, 8>([&](auto I) {
                constexpr int t = c8 + decltype(I)::value;
                val[t] = *(const lutv_t *)(addr[t] + hoff);
            });
        };
        if (row0 < slice_end) {
            uint32_t c0[CW];
            load_row(row0 + lane, c0);
            load_row(row0 + stride + lane, cnext);
            make_addr(c0);
            static_for<0, NCH>([&](auto C) { issue_chunk(C, std::integral_constant<int, 0>{}); });
        }

        for (; row0 < slice_end; row0 += stride) {
            // rows this wave-step may return
            unsigned long long vmask = ~0ull;
            if (slice_end - row0 < 64) vmask = (1ull << (int)(slice_end - row0)) - 1ull;
            if (a.valid) {
                const uint32_t *vw = a.valid + (row0 >> 5);
                unsigned long long vb = (unsigned long long)vw[0];
                if (row0 + 32 < a.N) vb |= (unsigned long long)vw[1] << 32;
                vmask &= vb;
            }

            f32x2 acc[NQ][NP];
#pragma unroll
            for (int h = 0; h < NQ; ++h)
#pragma unroll
                for (int p = 0; p < NP; ++p) acc[h][p] = (f32x2){0.f, 0.f};

            static_for<0, NQ>([&](auto H) {
                constexpr int h = decltype(H)::value;
                // ordered accumulation: pass 1 (steps t >= t0) over all chunks ...
                static_for<0, NCH>([&](auto C) {
                    constexpr int t0 = decltype(C)::value * 8;
                    if constexpr (MODE == 2) {
                        static_for<0, 8>([&](auto I) {
                            constexpr int t = t0 + decltype(I)::value;
                            if constexpr (QI == 4) {
                                sfma(acc[h][0], ANNLITE_LOHALF(val[t]), wt[t].x);
                                sfma(acc[h][1], ANNLITE_HIHALF(val[t]), wt[t].x);
                            } else {
                                sfma(acc[h][0], val[t], wt[t].x);
                            }
                        });
                    } else if constexpr (MODE == 1) {
                        static_for<0, 8>([&](auto I) {
                            constexpr int t = t0 + decltype(I)::value;
                            if constexpr (QI == 4) {
                                wfma_p1(acc[h][0], ANNLITE_LOHALF(val[t]), wt[t]);
                                wfma_p1(acc[h][1], ANNLITE_HIHALF(val[t]), wt[t]);
                            } else {
                                wfma_p1(acc[h][0], val[t], wt[t]);
                            }
                        });
                    } else if constexpr (QI == 4) pass8_q4<M, t0, 0>(acc[h][0], acc[h][1], val);
                    else pass8_q2<M, t0, 0>(acc[h][0], val);
                });
                // ... then pass 2 (t < t0) chunk by chunk, re-filling each chunk as soon as it is consumed
                static_for<0, NCH>([&](auto C) {
                    constexpr int cidx = decltype(C)::value;
                    constexpr int t0 = cidx * 8;
                    if constexpr (MODE == 2) {
                        static_for<0, 8>([&](auto I) {
                            constexpr int t = t0 + decltype(I)::value;
                            if constexpr (QI == 4) {
                                sfma(acc[h][0], ANNLITE_LOHALF(val[t]), wt[t].y);
                                sfma(acc[h][1], ANNLITE_HIHALF(val[t]), wt[t].y);
                            } else {
                                sfma(acc[h][0], val[t], wt[t].y);
                            }
                        });
                    } else if constexpr (MODE == 1) {
                        static_for<0, 8>([&](auto I) {
                            constexpr int t = t0 + decltype(I)::value;
                            if constexpr (QI == 4) {
                                wfma_p2(acc[h][0], ANNLITE_LOHALF(val[t]), wt[t]);
                                wfma_p2(acc[h][1], ANNLITE_HIHALF(val[t]), wt[t]);
                            } else {
                                wfma_p2(acc[h][0], val[t], wt[t]);
                            }
                        });
                    } else if constexpr (QI == 4) pass8_q4<M, t0, 1>(acc[h][0], acc[h][1], val);
                    else pass8_q2<M, t0, 1>(acc[h][0], val);
                    if constexpr (h + 1 < NQ) {
                        issue_chunk(C, std::integral_constant<int, h + 1>{});
                    } else {
                        if constexpr (cidx == 0) {
                            uint32_t cc[CW];
#pragma unroll
                            for (int i = 0; i < CW; ++i) cc[i] = cnext[i];
                            make_addr(cc);                                 // addresses of the next row
                            load_row(row0 + 2 * stride + lane, cnext);    // global prefetch, two rows ahead
                        }
                        issue_chunk(C, std::integral_constant<int, 0>{});
                    }
                });
            });

            // offer rows that can still enter a list (rare after warm-up): one branch for all queries
            const uint32_t rid = (uint32_t)(row0 + lane);
            float dq[QT];
            unsigned long long pmq[QT], any = 0;
#pragma unroll
            for (int q = 0; q < QT; ++q) {
                dq[q] = acc[q / QI][(q % QI) / 2][q % 2];
                pmq[q] = __ballot(dq[q] <= thr_f[q]) & vmask;
                any |= pmq[q];
            }
            if (any) {
#pragma unroll
                for (int q = 0; q < QT; ++q) {
                    if (pmq[q]) {
                        wavelist_offer(list[q], pmq[q], f32_to_ordered(dq[q]), rid, km1, thr_hi[q], thr_lo[q], lane);
                        thr_f[q] = (thr_hi[q] == kKeyInfHi) ? __builtin_inff() : ordered_to_f32(thr_hi[q]);
                    }
                }
            }
        }

        // ---- merge the NW per-wave lists of each query through LDS (re-using the LUT space) -----
        __syncthreads();
        unsigned long long *scratch = (unsigned long long *)smem;  // [QT][NW][64]
#pragma unroll
        for (int q = 0; q < QT; ++q)
            scratch[(q * NW + wave) * 64 + lane] = ((unsigned long long)list[q].hi << 32) | list[q].lo;
        __syncthreads();
        for (int q = wave; q < QT; q += NW) {
            WaveList L;
            unsigned long long key = scratch[(q * NW + 0) * 64 + lane];
            L.hi = (uint32_t)(key >> 32);
            L.lo = (uint32_t)key;
            uint32_t th = __builtin_amdgcn_readlane(L.hi, km1), tl = __builtin_amdgcn_readlane(L.lo, km1);
            for (int w = 1; w < NW; ++w) {
                key = scratch[(q * NW + w) * 64 + lane];
                const uint32_t chi = (uint32_t)(key >> 32), clo = (uint32_t)key;
                const unsigned long long pm = __ballot(lane <= km1 && key_less(chi, clo, th, tl));
                wavelist_offer(L, pm, chi, clo, km1, th, tl, lane);
            }
            const int b = tile * QT + q;
            if (b < a.B && lane <= km1)
                a.partial[((int64_t)b * a.n_slices + slice) * a.k + lane] = ((unsigned long long)L.hi << 32) | L.lo;
        }
    }
}


// =================================================================================================
// Filter kernel (default): the VALU cost of the ordered two-pass sum (2 lane-masked adds per
// look-up) bounds adc_scan_fast_kernel, so this version
//   1. adds the M values of a row in the lane's ROTATED order -- ONE plain v_pk_add_f32 per two
//      look-ups, values consumed as they arrive.  |d_fast - d_exact| <= 2*gamma_{M-1} * sum_m|v_m|
//      <= slack[q] := 2*M*2^-24 * Smax[q] * (1+2^-10), Smax[q] = sum_m max_k |lut[q][m][k]|
//      (lut_smax_kernel), because both are fp32 summations of the same M terms;
//   2. FILTERS: a row can only be in the top-k if d_exact <= thr, hence d_fast <= thr + slack;
//   3. for the few rows that pass, recomputes the EXACT ascending-m sum from the still-held
//      values (the two-pass masked add of the fast kernel) and offers (ordered(d_exact), id);
//   4. shares the k-th key between the waves of the workgroup through LDS (atomic min), so all
//      waves filter with the tightest bound any of them has proven;
//   5. inserts floods (first step of a work item) with a bitonic sort + merge instead of one
//      by one.
// Returned distances and ids are bit-identical to the fast kernel / the oracle.
// LDS: [LUT tile Ks*KSTRIDE][shthr f32 x QT (thr+slack) @ +0][shkey u64 x QT @ +64]
// =================================================================================================
template <int M, int NQ, int NW, int WPS, bool SKEWED, bool DBUF>
__global__ __launch_bounds__(NW * 64, WPS) void adc_scan_filter_kernel(const ScanArgs a) {
    constexpr int QI = 4;
    constexpr int QT = QI * NQ;
    constexpr int CW = M / 4;
    constexpr int EB = QI * 4;
    constexpr int RB = M * EB;
    constexpr int KSTRIDE = NQ * RB;
    static_assert(M % 8 == 0 && M <= 32, "QI=4 instantiations only");

    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int km1 = a.k - 1;
    const int s = lane % M;
    const uint32_t bsh = (uint32_t)(s & 3);
    bool abit[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) abit[i] = (((s >> 2) >> i) & 1) != 0;
    const unsigned char *mbase[M];
#pragma unroll
    for (int t = 0; t < M; ++t) mbase[t] = smem + ((s + t) % M) * EB;

    const int lut_bytes = a.Ks * KSTRIDE;
    volatile float *shthr = (volatile float *)(smem + lut_bytes);
    unsigned long long *shkey = (unsigned long long *)(smem + lut_bytes + 64);

    const int n_items = a.n_items;
    const int64_t group_bytes = (int64_t)a.Ks * RB;

    for (int item = blockIdx.x; item < n_items; item += gridDim.x) {
        int tile, slice;
        if (!item_map(a, item, tile, slice)) continue;

        __syncthreads();
        {
            const unsigned char *src0 = (const unsigned char *)a.lut + (int64_t)tile * NQ * group_bytes;
            constexpr int PIECES_PER_ROW = RB / 16;
            const int total = NQ * a.Ks * PIECES_PER_ROW;
            for (int idx = tid; idx < total; idx += NW * 64) {
                const int p = idx % PIECES_PER_ROW;
                const int kh = idx / PIECES_PER_ROW;
                const int h = kh / a.Ks;
                const int kk = kh - h * a.Ks;
                const u32x4 v = *(const u32x4 *)(src0 + (int64_t)h * group_bytes + (int64_t)kk * RB + p * 16);
                *(u32x4 *)(smem + (kk * NQ + h) * RB + p * 16) = v;
            }
            if (tid < QT) {
                shthr[tid] = __builtin_inff();
                shkey[tid] = ~0ull;
            }
        }
        float slack[QT];
#pragma unroll
        for (int q = 0; q < QT; ++q)
            slack[q] = a.smax[tile * QT + q] * (float)(2.0 * M * 5.9604644775390625e-08 * (1.0 + 1.0 / 1024.0));
        __syncthreads();

        WaveList list[QT];
#pragma unroll
        for (int q = 0; q < QT; ++q) list[q].reset();

        const int64_t slice_begin = (int64_t)slice * a.slice_rows;
        int64_t slice_end = slice_begin + a.slice_rows;
        if (slice_end > a.N) slice_end = a.N;

        const uint32_t *codes32 = (const uint32_t *)a.codes;
        auto load_row = [&](int64_t row, uint32_t (&c)[CW]) {
            if (row >= a.N) row = a.N - 1;
            const uint32_t *p = codes32 + row * CW;
            if constexpr (CW == 2) {
                const u32x2 v = *(const u32x2 *)p;
                c[0] = v.x;
                c[1] = v.y;
            } else {
#pragma unroll
                for (int i = 0; i < CW / 4; ++i) {
                    const u32x4 v = *(const u32x4 *)(p + 4 * i);
                    c[4 * i + 0] = v.x;
                    c[4 * i + 1] = v.y;
                    c[4 * i + 2] = v.z;
                    c[4 * i + 3] = v.w;
                }
            }
        };

        const int64_t stride = (int64_t)NW * 64;
        int64_t row0 = slice_begin + (int64_t)wave * 64;
        uint32_t cnext[CW];
        const unsigned char *addr[M];
        // DBUF: one landing buffer per entry group -- the next row's look-ups of group h are issued as
        // soon as group h of the current row has been filtered (more look-ups in flight per wave,
        // ~64 more VGPRs).  !DBUF: one buffer, refilled with the NEXT group right after the filter
        // (fewer registers -> more waves per SIMD).
        constexpr int NB = DBUF ? NQ : 1;
        f32x4 val[NB][M];
        auto make_addr = [&](uint32_t (&cc)[CW]) {
            if constexpr (!SKEWED) rotate_row<CW>(cc, abit, bsh);
            static_for<0, M>([&](auto T) {
                constexpr int t = decltype(T)::value;
                static_assert((KSTRIDE & (KSTRIDE - 1)) == 0, "KSTRIDE must be a power of two");
                addr[t] = mbase[t] + byte_shl<t % 4>(cc[t / 4], (uint32_t)ilog2_c(KSTRIDE));
            });
        };
        auto issue_group = [&](auto H) {
            constexpr int h = decltype(H)::value;
            static_for<0, M>([&](auto T) {
                constexpr int t = decltype(T)::value;
                val[DBUF ? h : 0][t] = *(const f32x4 *)(addr[t] + h * RB);
            });
        };
        // workgroup bound (thr + slack) of each group's 4 queries, re-read every step.  LDS returns
        // in order, so the read is issued BEFORE the refill look-ups of the group and consumed one
        // step later -- reading it at the point of use would drain the whole look-up queue.
        f32x4 th[NQ];
#pragma unroll
        for (int h = 0; h < NQ; ++h) th[h] = *(const f32x4 *)(smem + lut_bytes + h * 16);
        if (row0 < slice_end) {
            uint32_t c0[CW];
            load_row(row0 + lane, c0);
            load_row(row0 + stride + lane, cnext);
            make_addr(c0);
            static_for<0, NB>([&](auto H) { issue_group(H); });
        }

        int step_no = 0;
        for (; row0 < slice_end; row0 += stride, ++step_no) {
            unsigned long long vmask = ~0ull;
            if (slice_end - row0 < 64) vmask = (1ull << (int)(slice_end - row0)) - 1ull;
            if (a.valid) {
                const uint32_t *vw = a.valid + (row0 >> 5);
                unsigned long long vb = (unsigned long long)vw[0];
                if (row0 + 32 < a.N) vb |= (unsigned long long)vw[1] << 32;
                vmask &= vb;
            }
            const uint32_t rid = (uint32_t)(row0 + lane);
            const bool refresh = (step_no & 3) == 0;  // other waves' bounds are picked up every 4th step

            static_for<0, NQ>([&](auto H) {
                constexpr int h = decltype(H)::value;
                constexpr int hb = DBUF ? h : 0;
                // 1. fast sum, rotated order
                f32x4 fs = val[hb][0];
                static_for<1, M>([&](auto T) { fs += val[hb][decltype(T)::value]; });
                // 2. filter
                unsigned long long pm[4], any = 0;
#pragma unroll
                for (int jq = 0; jq < 4; ++jq) {
                    pm[jq] = __ballot(fs[jq] <= th[h][jq]) & vmask;
                    any |= pm[jq];
                }
                if (any) {
                    // 3. exact ascending-m sums of the 4 queries from the held values
                    f32x2 e0 = {0.f, 0.f}, e1 = {0.f, 0.f};
                    static_for<0, 2>([&](auto P) {
                        static_for<0, M / 8>([&](auto C) {
                            pass8_q4<M, decltype(C)::value * 8, decltype(P)::value>(e0, e1, val[hb]);
                        });
                    });
                    const float ex[4] = {e0.x, e0.y, e1.x, e1.y};
#pragma unroll
                    for (int jq = 0; jq < 4; ++jq) {
                        if (pm[jq]) {
                            const int q = h * 4 + jq;
                            const uint32_t khi = f32_to_ordered(ex[jq]);
                            const unsigned long long sk = *(volatile unsigned long long *)(shkey + q);
                            const uint32_t skhi = (uint32_t)(sk >> 32), sklo = (uint32_t)sk;
                            const unsigned long long px = __ballot(key_less(khi, rid, skhi, sklo)) & pm[jq];
                            if (px) {
                                wavelist_insert_many(list[q], px, khi, rid, lane);
                                // 4. publish this wave's k-th key if it tightens the workgroup bound
                                const uint32_t ohi = __builtin_amdgcn_readlane(list[q].hi, km1);
                                const uint32_t olo = __builtin_amdgcn_readlane(list[q].lo, km1);
                                if (lane == 0 && ohi != kKeyInfHi) {
                                    const unsigned long long mine = ((unsigned long long)ohi << 32) | olo;
                                    const unsigned long long old = atomicMin(shkey + q, mine);
                                    if (mine < old) shthr[q] = ordered_to_f32(ohi) + slack[q];
                                }
                            }
                        }
                    }
                }
                // plain LDS read (ds_read_b128) behind a compiler barrier so it is re-issued every step; a
                // volatile access would be lowered to a FLAT load + vmcnt(0)/lgkmcnt(0) drains
                if (refresh) {
                    asm volatile("" ::: "memory");
                    th[h] = *(const f32x4 *)(smem + lut_bytes + h * 16);
                }
                if constexpr (DBUF) {
                    // refill this group's buffer with the next row's look-ups
                    if constexpr (h == 0) {
                        uint32_t cc[CW];
#pragma unroll
                        for (int i = 0; i < CW; ++i) cc[i] = cnext[i];
                        make_addr(cc);
                        load_row(row0 + 2 * stride + lane, cnext);
                    }
                    issue_group(H);
                } else if constexpr (h + 1 < NQ) {
                    issue_group(std::integral_constant<int, h + 1>{});  // next group of the same row
                } else {
                    uint32_t cc[CW];
#pragma unroll
                    for (int i = 0; i < CW; ++i) cc[i] = cnext[i];
                    make_addr(cc);
                    load_row(row0 + 2 * stride + lane, cnext);
                    issue_group(std::integral_constant<int, 0>{});  // first group of the next row
                }
            });
        }

        // ---- merge the NW per-wave lists of each query through LDS (re-using the LUT space) -----
        __syncthreads();
        unsigned long long *scratch = (unsigned long long *)smem;  // [QT][NW][64]
#pragma unroll
        for (int q = 0; q < QT; ++q)
            scratch[(q * NW + wave) * 64 + lane] = ((unsigned long long)list[q].hi << 32) | list[q].lo;
        __syncthreads();
        for (int q = wave; q < QT; q += NW) {
            WaveList L;
            unsigned long long key = scratch[(q * NW + 0) * 64 + lane];
            L.hi = (uint32_t)(key >> 32);
            L.lo = (uint32_t)key;
            for (int w = 1; w < NW; ++w) {
                key = scratch[(q * NW + w) * 64 + lane];
                // every wave list is ascending over the lanes: sorted merge, keep the 64 smallest
                wavelist_merge_sorted(L, (uint32_t)(key >> 32), (uint32_t)key, lane);
            }
            const int b = tile * QT + q;
            if (b < a.B && lane <= km1)
                a.partial[((int64_t)b * a.n_slices + slice) * a.k + lane] = ((unsigned long long)L.hi << 32) | L.lo;
        }
    }
}



// One (wave, query) candidate event of the quantised-filter kernel.  Out of line on purpose (one shared
// copy stays hot in the instruction cache; inlined 16x the kernel had >10k instructions).
//
// The top-k of a (workgroup, query) lives ONCE in LDS: 64 sorted (key, id) entries + a 4-byte lock.
// A wave that has candidate rows (integer filter passed) gathers their exact fp32 sums, drops those
// that do not beat the list's current k-th key, takes the lock, merges, publishes the new integer
// bound.  The bound every wave filters with is therefore the k-th best of ALL rows the workgroup
// has seen (with per-wave lists it was only the best single wave's k-th: 2.7x more events).
extern __shared__ __attribute__((aligned(16))) unsigned char g_smem[];

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
    if (!(qd > 0.0)) qd = 0.0;
    if (!(qd < 32767.0)) qd = 32767.0;
    return (unsigned short)(0x8000u | (uint32_t)qd);
}


// ---- M = 64: the "wrap-coded" SKEWED layout ------------------------------------------------------
// With 64 sub-spaces a lane cannot afford one LDS base pointer per step (64 VGPRs).  Lane l (row n,
// n % 64 == l) reads sub-space u = l + t at step t; the address is (code << 9) + l*8 + t*8 with t*8 as the
// instruction's immediate -- correct while u < 64.  For u >= 64 the true entry is (code, u - 64): 512
// bytes lower, i.e. the SAME offset inside the PREVIOUS table row.  So the stored byte of a wrapped
// position is code - 1 (mod 256), and LDS carries one extra row 256 = copy of row 0 for code 0 - 1 = 255.
// stored byte j of row n:  code[(j + n) % 64] - [(j + n % 64) >= 64]   (mod 256)
// 0x01 in every byte of dword w (positions 4w..4w+3) whose position j satisfies j + r >= 64
__device__ __forceinline__ uint32_t wrap64_mask(int w, int r) {
    int nb = 4 * w + 4 - (64 - r);  // number of (upper) bytes of the dword that wrap
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

// what a queue flush needs and a work item keeps constant
struct FlushCtx {
    const uint8_t *codes;
    const float *lut;
    const float *smax;
    const float *qstep;
    const double *qlo;
    unsigned long long *gkey;
    unsigned long long *gk2;
    unsigned long long *dbg;
    int32_t Ks, b0, n_slices, slice, km1, jm1, skip;
    uint32_t list_off, lock_off, shq_off, gkl_off, gjl_off;
};

// Flush one wave's candidate queue: up to 64 (query, row) pairs whose integer sum passed the filter.
// Lane i takes entry i: re-reads the row's code bytes, gathers its exact ascending-m fp32 sum from the
// fp32 table in global memory, then the candidates are offered query by query to the shared lists.
// Batching matters: one candidate at a time paid the gather latency, the call and the lock ~3.6 us each
// (46 times per wave at 1.25M rows); a flush pays them once for everything queued since the last one.
template <int M, bool SKEWED>
__device__ __attribute__((noinline)) void qfilter_flush(const FlushCtx c, uint32_t queue_off, int qcnt) {
    constexpr int CW = M / 4;
    const int lane = threadIdx.x & 63;
    const bool act = lane < qcnt;
    const unsigned long long e = act ? ((const unsigned long long *)(g_smem + queue_off))[lane] : 0ull;
    const uint32_t rid = (uint32_t)e;
    const int q = (int)(e >> 32);
    float ex = 0.f;
    if (act && !(c.skip & 1)) {
        uint32_t cp[CW];
        const uint32_t *p = (const uint32_t *)(c.codes + (int64_t)rid * M);
#pragma unroll
        for (int i = 0; i < CW; ++i) cp[i] = p[i];
        if constexpr (SKEWED && M == 64) {
            const int r = (int)(rid % 64);
#pragma unroll
            for (int i = 0; i < CW; ++i) cp[i] = bytes_add(cp[i], wrap64_mask(i, r));  // undo the wrap coding
        }
        if constexpr (SKEWED) {
            // stored byte j of row n is the code of sub-space (j + n) mod M: rotate back by n mod M
            const int sinv = (M - (int)(rid % M)) % M;
            bool abit_inv[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) abit_inv[i] = (((sinv >> 2) >> i) & 1) != 0;
            rotate_row<CW>(cp, abit_inv, (uint32_t)(sinv & 3));
        }
        const int b = c.b0 + q;
        const float *lq = c.lut + ((int64_t)(b >> 2) * c.Ks) * (M * 4) + (b & 3);
        float vals[M];
#pragma unroll
        for (int m = 0; m < M; ++m) {
            const uint32_t code = (cp[m / 4] >> (8 * (m % 4))) & 0xffu;
            vals[m] = lq[((int64_t)code * M + m) * 4];
        }
#pragma unroll
        for (int m = 0; m < M; ++m) ex += vals[m];
    }
    const uint32_t khi = f32_to_ordered(ex);
    unsigned long long rem = __ballot(act);
    while (rem) {
        const int q0 = __builtin_amdgcn_readlane(q, __builtin_ctzll(rem));
        const unsigned long long pm = __ballot(act && q == q0);
        rem &= ~pm;
        if (c.dbg && lane == 0) {
            atomicAdd(c.dbg + 1, 1ull);
            atomicAdd(c.dbg + 4, (unsigned long long)__popcll(pm));
        }
        const int b = c.b0 + q0;
        unsigned long long *list = (unsigned long long *)(g_smem + c.list_off + q0 * 512);  // [64] ascending
        unsigned long long *gkl = (unsigned long long *)(g_smem + c.gkl_off + q0 * 8);
        // cheap pre-check against the current bound, without the lock (it only ever decreases): the list's
        // own k-th key or the best bound imported from the other workgroups, whichever is smaller
        unsigned long long kth = __hip_atomic_load(list + c.km1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        const unsigned long long gk = __hip_atomic_load(gkl, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        if (gk < kth) kth = gk;
        unsigned long long px = __ballot(key_less(khi, rid, (uint32_t)(kth >> 32), (uint32_t)kth)) & pm;
        if (!px || (c.skip & 2)) continue;
        if (c.dbg && lane == 0) atomicAdd(c.dbg + 2, 1ull);
        // ---- critical section ---------------------------------------------------------------------
        unsigned int *lock = (unsigned int *)(g_smem + c.lock_off + q0 * 4);
        for (;;) {
            unsigned int got = 0;
            if (lane == 0) got = (atomicCAS(lock, 0u, 1u) == 0u) ? 1u : 0u;
            if (__builtin_amdgcn_readfirstlane(got)) break;
            __builtin_amdgcn_s_sleep(2);
        }
        const unsigned long long le = __hip_atomic_load(list + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        WaveList L;
        L.hi = (uint32_t)(le >> 32);
        L.lo = (uint32_t)le;
        const uint32_t thi = __builtin_amdgcn_readlane(L.hi, c.km1), tlo = __builtin_amdgcn_readlane(L.lo, c.km1);
        px = __ballot(key_less(khi, rid, thi, tlo)) & px;  // the list may have tightened meanwhile
        if (px) {
            wavelist_insert_many(L, px, khi, rid, lane);
            __hip_atomic_store(list + lane, ((unsigned long long)L.hi << 32) | L.lo, __ATOMIC_RELAXED,
                               __HIP_MEMORY_SCOPE_WORKGROUP);
            const uint32_t ohi = __builtin_amdgcn_readlane(L.hi, c.km1);
            const uint32_t olo = __builtin_amdgcn_readlane(L.lo, c.km1);
            const unsigned long long okey = ((unsigned long long)ohi << 32) | olo;
            if (lane == 0 && ohi != kKeyInfHi && okey < gk) {
                if (c.dbg) atomicAdd(c.dbg + 3, 1ull);
                // tell the other workgroups of this query (other row slices) and remember it locally
                if (c.gkey) __hip_atomic_fetch_min(c.gkey + b, okey, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(gkl, okey, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                *(volatile unsigned short *)(g_smem + c.shq_off + q0 * 2) =
                    qbound_from_key<M>(okey, c.smax[b], c.qstep[b], c.qlo[b]);
            }
            if (c.gk2) {
                // this slice's j-th key, for the max-of-j-th bound the sibling slices compute
                const uint32_t jhi = __builtin_amdgcn_readlane(L.hi, c.jm1);
                const uint32_t jlo = __builtin_amdgcn_readlane(L.lo, c.jm1);
                const unsigned long long jkey = ((unsigned long long)jhi << 32) | jlo;
                volatile unsigned long long *gjl = (volatile unsigned long long *)(g_smem + c.gjl_off + q0 * 8);
                if (lane == 0 && jhi != kKeyInfHi && jkey < *gjl) {
                    *gjl = jkey;
                    __hip_atomic_store(c.gk2 + (int64_t)b * c.n_slices + c.slice, jkey, __ATOMIC_RELAXED,
                                       __HIP_MEMORY_SCOPE_AGENT);
                }
            }
        }
        // LDS executes one wave's instructions in order, so the list stores are visible before the release
        if (lane == 0) __hip_atomic_store(lock, 0u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
}

// =================================================================================================
// Quantised-filter kernel.  Same discipline as adc_scan_filter_kernel (cheap bound -> exact
// recompute for the few rows that pass -> bit-exact output) but the cheap bound is an INTEGER sum
// over a 12-bit quantised copy of the tables:
//     Q[q][m][k] = min(QMAX, floor((lut[q][m][k] - lo[q][m]) / step[q])),  QMAX = floor(32767 / M)
//   * 8 queries per 16-byte LDS entry (u16 each): one ds_read_b128 serves 8 look-ups per lane, half
//     the LDS bytes of the fp32 filter, and a workgroup holds 16 queries in the same 128 KB;
//   * two u16 partial sums share a dword and are added with ONE plain v_add_u32 (VOP2, 2.5 cycles per
//     wave-instruction vs 4.5 for v_pk_add_f32 -- scripts/valu_ubench.hip); M*QMAX < 32768, so the
//     low half never carries into the high half, and the filter test is ONE more VOP2 per dword:
//     (0x8000|qthr) - S keeps bit 15 of a half set iff S <= qthr (no borrow can cross the halves);
//   * bound: with L = sum_m lo[q][m], S = integer sum, every entry satisfies
//     lo + step*(Q - 0.002) <= v <= lo + step*(Q + 1.002), hence d_real >= L + step*(S - 0.04); a row
//     can be in the top-k only if d_exact <= thr, d_real <= thr + slack32, i.e.
//     S <= qthr := floor((thr + slack32 - L) / step) + 1   (computed in double when thr changes);
//   * rows with S <= qthr get their exact ascending-m fp32 sum from the fp32 table in global memory
//     (L2-resident: 16 queries x 16 KB per workgroup), then the usual (ordered(d), id) offer.
// LDS: [Q tile Ks*KSTRIDE][shq16 u16 x QT (0x8000|qthr) @ +0][locks u32 x QT @ +64][gkl u64 x QT @ +128]
//      [lists u64 x QT x 64 @ +256]
// =================================================================================================
template <int M, int NQ, int NW, int WPS, bool SKEWED>
__global__ __launch_bounds__(NW * 64, WPS) void adc_scan_qfilter_kernel(const ScanArgs a) {
    constexpr int QG = 8;                 // queries per LDS entry
    constexpr int QT = QG * NQ;           // queries per workgroup
    constexpr int CW = M / 4;
    constexpr int EB = 16;
    constexpr int RB = M * EB;
    constexpr int KSTRIDE = NQ * RB;
    static_assert(M % 8 == 0 && M <= 32 && (KSTRIDE & (KSTRIDE - 1)) == 0, "unsupported shape");

    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int km1 = a.k - 1;
    const int s = lane % M;
    // forward rotation (PLAIN tables) and its inverse (to read a row's bytes in true sub-space order)
    const uint32_t bsh = (uint32_t)(s & 3);
    bool abit[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) abit[i] = (((s >> 2) >> i) & 1) != 0;
    const unsigned char *mbase[M];
#pragma unroll
    for (int t = 0; t < M; ++t) mbase[t] = smem + ((s + t) % M) * EB;

    const int lut_bytes = a.Ks * KSTRIDE;
    const uint32_t shq_off = (uint32_t)lut_bytes, lock_off = shq_off + 64, gkl_off = shq_off + 128,
                   list_off = shq_off + 256, gjl_off = list_off + QT * 512, queue_off = gjl_off + 128;
    unsigned long long *gkl = (unsigned long long *)(smem + gkl_off);  // [QT] best published k-th key
    volatile uint16_t *shq = (volatile uint16_t *)(smem + shq_off);
    volatile uint32_t *locks = (volatile uint32_t *)(smem + lock_off);
    unsigned long long *lists = (unsigned long long *)(smem + list_off);  // [QT][64]

    const int n_items = a.n_items;
    const int64_t group_bytes = (int64_t)a.Ks * RB;

    for (int item = blockIdx.x; item < n_items; item += gridDim.x) {
        int tile, slice;
        if (!item_map(a, item, tile, slice)) continue;

        __syncthreads();
        {
            const unsigned char *src0 = (const unsigned char *)a.q16 + (int64_t)tile * NQ * group_bytes;
            constexpr int PIECES_PER_ROW = RB / 16;
            const int total = NQ * a.Ks * PIECES_PER_ROW;
            for (int idx = tid; idx < total; idx += NW * 64) {
                const int p = idx % PIECES_PER_ROW;
                const int kh = idx / PIECES_PER_ROW;
                const int h = kh / a.Ks;
                const int kk = kh - h * a.Ks;
                const u32x4 v = *(const u32x4 *)(src0 + (int64_t)h * group_bytes + (int64_t)kk * RB + p * 16);
                *(u32x4 *)(smem + (kk * NQ + h) * RB + p * 16) = v;
            }
            if (tid < QT) {
                locks[tid] = 0;
                // start from whatever other workgroups (other row slices, earlier items) already proved
                const int b = tile * QT + tid;
                const unsigned long long gk =
                    a.gkey ? __hip_atomic_load(a.gkey + b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : ~0ull;
                gkl[tid] = gk;
                ((unsigned long long *)(smem + gjl_off))[tid] = ~0ull;
                shq[tid] = qbound_from_key<M>(gk, a.smax[b], a.qstep[b], a.qlo[b]);
            }
            for (int idx = tid; idx < QT * 64; idx += NW * 64) lists[idx] = ~0ull;
        }
        __syncthreads();

        const int64_t slice_begin = (int64_t)slice * a.slice_rows;
        int64_t slice_end = slice_begin + a.slice_rows;
        if (slice_end > a.N) slice_end = a.N;

        const uint32_t *codes32 = (const uint32_t *)a.codes;
        auto load_row = [&](int64_t row, uint32_t (&c)[CW]) {
            if (row >= a.N) row = a.N - 1;
            const uint32_t *p = codes32 + row * CW;
            if constexpr (CW == 2) {
                const u32x2 v = *(const u32x2 *)p;
                c[0] = v.x;
                c[1] = v.y;
            } else {
#pragma unroll
                for (int i = 0; i < CW / 4; ++i) {
                    const u32x4 v = *(const u32x4 *)(p + 4 * i);
                    c[4 * i + 0] = v.x;
                    c[4 * i + 1] = v.y;
                    c[4 * i + 2] = v.z;
                    c[4 * i + 3] = v.w;
                }
            }
        };

        const int64_t stride = (int64_t)NW * 64;
        int64_t row0 = slice_begin + (int64_t)wave * 64;
        uint32_t ccur[CW], cnext[CW];   // rotated code bytes of the current / next row of this lane
        const unsigned char *addr[M];
        auto make_addr = [&](const uint32_t (&cc)[CW]) {
            static_for<0, CW>([&](auto W) {
                constexpr int w = decltype(W)::value;
                uint32_t o0, o1, o2, o3;
                byte_shl4(cc[w], (uint32_t)ilog2_c(KSTRIDE), o0, o1, o2, o3);
                addr[4 * w + 0] = mbase[4 * w + 0] + o0;
                addr[4 * w + 1] = mbase[4 * w + 1] + o1;
                addr[4 * w + 2] = mbase[4 * w + 2] + o2;
                addr[4 * w + 3] = mbase[4 * w + 3] + o3;
            });
        };
        // integer sums of one entry group: 4 dwords x (2 x u16)
        auto group_sum = [&](auto H, u32x4 &acc) {
            constexpr int h = decltype(H)::value;
            // issue all M look-ups of the group before the first add (the compiler otherwise re-used one
            // register pair and waited lgkmcnt(0) after every second load)
            u32x4 v[M];
            static_for<0, M>([&](auto T) {
                constexpr int t = decltype(T)::value;
                v[t] = *(const u32x4 *)(addr[t] + h * RB);
            });
            asm volatile("" ::: "memory");
            acc = v[0];
            static_for<1, M>([&](auto T) { acc += v[decltype(T)::value]; });
        };
        u32x4 thp[NQ];  // packed (0x8000 | qthr) of the group's 8 queries
#pragma unroll
        for (int h = 0; h < NQ; ++h) thp[h] = *(const u32x4 *)(smem + lut_bytes + h * 16);
        uint32_t vcur = ~0u, vnext = ~0u;
        auto load_valid = [&](int64_t row) -> uint32_t {
            if (!a.valid) return ~0u;
            if (row >= a.N) row = a.N - 1;
            return a.valid[row >> 5];
        };
        if (row0 < slice_end) {
            load_row(row0 + lane, ccur);
            if constexpr (!SKEWED) rotate_row<CW>(ccur, abit, bsh);
            load_row(row0 + stride + lane, cnext);
            vcur = load_valid(row0 + lane);
            vnext = load_valid(row0 + stride + lane);
        }

        const FlushCtx fc = {(const uint8_t *)a.codes, a.lut, a.smax, a.qstep, a.qlo, a.gkey, a.gk2, a.dbg,
                             a.Ks, tile * QT, a.n_slices, slice, km1, a.jm1, a.dbg_skip,
                             list_off, lock_off, shq_off, gkl_off, gjl_off};
        int qcnt = 0;  // entries in this wave's candidate queue
        int step_no = 0;
        for (; row0 < slice_end; row0 += stride, ++step_no) {
            unsigned long long vmask = ~0ull;
            if (slice_end - row0 < 64) vmask = (1ull << (int)(slice_end - row0)) - 1ull;
            // validity word of this lane's row, fetched one step ahead with the code bytes (a scalar load here
            // would make the wave drain lgkmcnt -- i.e. all its LDS look-ups -- before the first add)
            if (a.valid) vmask &= __ballot((vcur >> (lane & 31)) & 1u);
            const uint32_t rid = (uint32_t)(row0 + lane);
            make_addr(ccur);
            u32x4 acc[NQ];
            static_for<0, NQ>([&](auto H) { group_sum(H, acc[decltype(H)::value]); });

            // any (query, lane) with S <= qthr ?  (0x8000|qthr) - S has bit 15 of that half set
            uint32_t anyv = 0;
#pragma unroll
            for (int h = 0; h < NQ; ++h)
#pragma unroll
                for (int w = 0; w < 4; ++w) anyv |= thp[h][w] - acc[h][w];
            const unsigned long long anym = __ballot((anyv & 0x80008000u) != 0) & vmask;
            bool flushed = false;
            if (anym && !(a.dbg_skip & 4)) {
                if (a.dbg && lane == 0) atomicAdd(a.dbg + 0, 1ull);
                // queue (query, row) of every lane that passed; the exact work happens in batches
                unsigned long long *queue = (unsigned long long *)(smem + queue_off + wave * 512);
#pragma unroll
                for (int h = 0; h < NQ; ++h) {
#pragma unroll
                    for (int w = 0; w < 4; ++w) {
                        const uint32_t x = (thp[h][w] - acc[h][w]) & 0x80008000u;
                        if (__ballot(x != 0) & vmask) {
#pragma unroll
                            for (int half = 0; half < 2; ++half) {
                                const unsigned long long pm = __ballot((x & (half ? 0x80000000u : 0x8000u)) != 0) & vmask;
                                if (pm) {
                                    const int n = __popcll(pm);
                                    if (qcnt + n > 64) {
                                        qfilter_flush<M, SKEWED>(fc, queue_off + wave * 512, qcnt);
                                        qcnt = 0;
                                        flushed = true;
                                    }
                                    const int rank = __builtin_amdgcn_mbcnt_hi(
                                        (uint32_t)(pm >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)pm, 0u));
                                    if ((pm >> lane) & 1ull)
                                        queue[qcnt + rank] = ((unsigned long long)(h * QG + w * 2 + half) << 32) | rid;
                                    qcnt += n;
                                }
                            }
                        }
                    }
                }
            }
            // flush when half full, and every (flush_mask + 1) steps: a flush costs ~7 us whatever it holds (two
            // dependent global round trips + the list update), so waves flush rarely but staggered -- every few
            // steps SOME wave of the workgroup tightens the shared bound
            if (qcnt && (qcnt >= 32 || ((step_no + wave * 4) & a.flush_mask) == a.flush_mask)) {
                qfilter_flush<M, SKEWED>(fc, queue_off + wave * 512, qcnt);
                qcnt = 0;
                flushed = true;
            }
            // Import what the other workgroups of these queries (the other row slices) have proven: the best
            // k-th key any of them published and, per group of 8 concurrently scanned slices, the MAX of
            // their j-th keys (8 disjoint slices x j rows >= k rows at or below it; +1: that row itself must
            // still be accepted).  Bounds move on a log scale, so: steps 0, 1, 3, 7, ... then every 64th.
            if (a.gkey) {
                const bool pow2 = ((step_no + 1) & step_no) == 0;
                if (pow2 || (step_no & 63) == 63) {
                    const int rw = pow2 ? (__builtin_ctz((unsigned)step_no + 1u) % NW) : ((step_no >> 6) % NW);
                    if (wave == rw) {
#pragma unroll 1
                        for (int q0 = 0; q0 < QT; q0 += 8) {
                            const int q = q0 + (lane >> 3);
                            const int b = tile * QT + q;
                            unsigned long long bound =
                                __hip_atomic_load(a.gkey + b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            if (a.gk2) {
#pragma unroll 1
                                for (int g0 = 0; g0 < a.n_slices; g0 += 8) {
                                    unsigned long long v = 0ull;  // slots beyond n_slices never set the max
                                    if (g0 + (lane & 7) < a.n_slices)
                                        v = __hip_atomic_load(a.gk2 + (int64_t)b * a.n_slices + g0 + (lane & 7),
                                                              __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
                                    for (int o = 1; o < 8; o <<= 1) {
                                        const unsigned long long p = __shfl_xor(v, o);
                                        v = p > v ? p : v;
                                    }
                                    if (v != ~0ull && v + 1ull < bound) bound = v + 1ull;
                                }
                            }
                            if ((lane & 7) == 0 &&
                                bound < __hip_atomic_load(gkl + q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) {
                                __hip_atomic_store(gkl + q, bound, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                                const unsigned short nb = qbound_from_key<M>(bound, a.smax[b], a.qstep[b], a.qlo[b]);
                                if (nb < shq[q]) shq[q] = nb;
                            }
                        }
                    }
                }
            }
            // pick up the workgroup bound: every 4th step, and right after this wave's own events
            if (flushed || (step_no & 3) == 3) {
                asm volatile("" ::: "memory");
#pragma unroll
                for (int h = 0; h < NQ; ++h) thp[h] = *(const u32x4 *)(smem + lut_bytes + h * 16);
            }
            // next row
#pragma unroll
            for (int i = 0; i < CW; ++i) ccur[i] = cnext[i];
            if constexpr (!SKEWED) rotate_row<CW>(ccur, abit, bsh);
            load_row(row0 + 2 * stride + lane, cnext);
            vcur = vnext;
            vnext = load_valid(row0 + 2 * stride + lane);
        }

        if (qcnt) qfilter_flush<M, SKEWED>(fc, queue_off + wave * 512, qcnt);

        // ---- the shared lists ARE the workgroup's result for this (tile, slice) ----------------------
        __syncthreads();
        for (int q = wave; q < QT; q += NW) {
            const int b = tile * QT + q;
            // device-scope stores: the merging workgroup may sit on another XCD (own L2)
            if (b < a.B && lane <= km1)
                __hip_atomic_store(a.partial + ((int64_t)b * a.n_slices + slice) * a.k + lane, lists[q * 64 + lane],
                                   __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        if (a.tile_done) {
            // the last of the tile's n_slices workgroups to arrive merges them (saves the merge launch and the
            // ~10 us kernel boundary in front of it)
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's list stores have completed
            __syncthreads();
            volatile unsigned int *s_flag = (volatile unsigned int *)(smem + lock_off);  // locks are idle now
            if (tid == 0) {
                const unsigned int old =
                    __hip_atomic_fetch_add(a.tile_done + tile, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                *s_flag = (old + 1u == (unsigned int)(a.n_slices - 1)) ? 1u : 0u;
            }
            __syncthreads();
            if (*s_flag) {
                for (int q = wave; q < QT; q += NW) {
                    const int b = tile * QT + q;
                    if (b >= a.B) continue;
                    WaveList L;
                    L.reset();
                    for (int sl = 0; sl < a.n_slices; ++sl) {
                        unsigned long long key = ~0ull;
                        if (lane <= km1)
                            key = __hip_atomic_load(a.partial + ((int64_t)b * a.n_slices + sl) * a.k + lane,
                                                    __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        // every slice list is ascending over the lanes: sorted merge, keep the 64 smallest
                        wavelist_merge_sorted(L, (uint32_t)(key >> 32), (uint32_t)key, lane);
                    }
                    if (lane <= km1) {
                        const bool none = (L.hi == kKeyInfHi && L.lo == kIdNone);
                        const float d = none ? __builtin_inff() : ordered_to_f32(L.hi);
                        const int64_t id = none ? (int64_t)-1 : a.row_base + (int64_t)L.lo;
                        if (a.out_packed) {
                            a.out_packed[((int64_t)b * a.k + lane) * 2 + 0] = id;
                            a.out_packed[((int64_t)b * a.k + lane) * 2 + 1] = (int64_t)__float_as_uint(d);
                        } else {
                            a.out_d[(int64_t)b * a.k + lane] = a.sqrt_out ? __builtin_sqrtf(d) : d;
                            a.out_i[(int64_t)b * a.k + lane] = id;
                        }
                    }
                }
            }
            __syncthreads();  // s_flag (the lock words) is re-initialised by the next item
        }
    }
}


// =================================================================================================
// Quantised-filter kernel for M = 64 (BASELINE config 4: 768-d, 12-float sub-spaces).  Same discipline as
// adc_scan_qfilter_kernel; what differs is dictated by the table size (64 sub-spaces x 256 codes):
//   * 4 queries per 8-byte LDS entry (u16 each), table [Ks + 1][64][8 B] = 128.5 KB, ds_read_b64; a
//     half-wave reads sub-spaces (l + t) % 64 for 32 consecutive l: bank pair (l + t) % 32, conflict-free;
//   * no per-step LDS base registers: the wrap-coded SKEWED layout (wrap64_mask) makes the address
//     (stored byte << 9) + lane*8 with t*8 as the instruction's immediate -- one SDWA shift + one add;
//   * QMAX = floor(32767 / 64) = 511 (9-bit entries), look-ups issued in 4 chunks of 16;
//   * PLAIN tables are rotated and wrap-coded on the fly (slow path; the index plugin stores SKEWED).
// LDS: [table (Ks+1)*512][shq u16 x 4 @ +0][locks u32 x 4 @ +64][gkl u64 x 4 @ +128][lists u64 x 4 x 64 @ +256]
//      [gjl u64 x 4][queues u64 x NW x 64]
// =================================================================================================
template <int NW, bool SKEWED>
__global__ __launch_bounds__(NW * 64) void adc_scan_qfilter64_kernel(const ScanArgs a) {
    constexpr int M = 64, QT = 4, CW = 16, RB = 512;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int km1 = a.k - 1;
    // forward rotation of PLAIN rows by lane bytes
    const uint32_t bsh = (uint32_t)(lane & 3);
    bool abit[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) abit[i] = (((lane >> 2) >> i) & 1) != 0;

    const int lut_bytes = (a.Ks + 1) * RB;
    const uint32_t shq_off = (uint32_t)lut_bytes, lock_off = shq_off + 64, gkl_off = shq_off + 128,
                   list_off = shq_off + 256, gjl_off = list_off + QT * 512, queue_off = gjl_off + 128;
    unsigned long long *gkl = (unsigned long long *)(smem + gkl_off);
    volatile uint16_t *shq = (volatile uint16_t *)(smem + shq_off);
    volatile uint32_t *locks = (volatile uint32_t *)(smem + lock_off);
    unsigned long long *lists = (unsigned long long *)(smem + list_off);
    const unsigned char *lbase = smem + lane * 8;

    for (int item = blockIdx.x; item < a.n_items; item += gridDim.x) {
        int tile, slice;
        if (!item_map(a, item, tile, slice)) continue;

        __syncthreads();
        {
            const u32x4 *src = (const u32x4 *)((const unsigned char *)a.q16 + (int64_t)tile * a.Ks * RB);
            const int total = a.Ks * (RB / 16);
            for (int idx = tid; idx < total; idx += NW * 64) ((u32x4 *)smem)[idx] = src[idx];
            for (int idx = tid; idx < RB / 16; idx += NW * 64) ((u32x4 *)(smem + a.Ks * RB))[idx] = src[idx];  // row Ks = row 0
            if (tid < QT) {
                locks[tid] = 0;
                const int b = tile * QT + tid;
                const unsigned long long gk =
                    a.gkey ? __hip_atomic_load(a.gkey + b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : ~0ull;
                gkl[tid] = gk;
                ((unsigned long long *)(smem + gjl_off))[tid] = ~0ull;
                shq[tid] = qbound_from_key<M>(gk, a.smax[b], a.qstep[b], a.qlo[b]);
            }
            for (int idx = tid; idx < QT * 64; idx += NW * 64) lists[idx] = ~0ull;
        }
        __syncthreads();

        const int64_t slice_begin = (int64_t)slice * a.slice_rows;
        int64_t slice_end = slice_begin + a.slice_rows;
        if (slice_end > a.N) slice_end = a.N;

        const uint32_t *codes32 = (const uint32_t *)a.codes;
        auto load_row = [&](int64_t row, uint32_t (&c)[CW]) {
            if (row >= a.N) row = a.N - 1;
            const uint32_t *p = codes32 + row * CW;
#pragma unroll
            for (int i = 0; i < CW / 4; ++i) {
                const u32x4 v = *(const u32x4 *)(p + 4 * i);
                c[4 * i + 0] = v.x;
                c[4 * i + 1] = v.y;
                c[4 * i + 2] = v.z;
                c[4 * i + 3] = v.w;
            }
        };
        auto encode_plain = [&](uint32_t (&c)[CW]) {  // PLAIN row -> this lane's wrap-coded SKEWED row
            rotate_row<CW>(c, abit, bsh);
#pragma unroll
            for (int i = 0; i < CW; ++i) c[i] = bytes_sub(c[i], wrap64_mask(i, lane));
        };
        auto load_valid = [&](int64_t row) -> uint32_t {
            if (!a.valid) return ~0u;
            if (row >= a.N) row = a.N - 1;
            return a.valid[row >> 5];
        };

        const int64_t stride = (int64_t)NW * 64;
        int64_t row0 = slice_begin + (int64_t)wave * 64;
        uint32_t ccur[CW], cnext[CW];
        uint32_t vcur = ~0u, vnext = ~0u;
        u32x2 thp = *(const u32x2 *)(smem + shq_off);  // packed (0x8000 | qthr) of the 4 queries
        if (row0 < slice_end) {
            load_row(row0 + lane, ccur);
            if constexpr (!SKEWED) encode_plain(ccur);
            load_row(row0 + stride + lane, cnext);
            vcur = load_valid(row0 + lane);
            vnext = load_valid(row0 + stride + lane);
        }
        const FlushCtx fc = {(const uint8_t *)a.codes, a.lut, a.smax, a.qstep, a.qlo, a.gkey, a.gk2, a.dbg,
                             a.Ks, tile * QT, a.n_slices, slice, km1, a.jm1, a.dbg_skip,
                             list_off, lock_off, shq_off, gkl_off, gjl_off};
        int qcnt = 0;
        int step_no = 0;
        for (; row0 < slice_end; row0 += stride, ++step_no) {
            unsigned long long vmask = ~0ull;
            if (slice_end - row0 < 64) vmask = (1ull << (int)(slice_end - row0)) - 1ull;
            if (a.valid) vmask &= __ballot((vcur >> (lane & 31)) & 1u);
            const uint32_t rid = (uint32_t)(row0 + lane);

            u32x2 acc = {0u, 0u};
            static_for<0, 4>([&](auto C) {
                constexpr int c0 = decltype(C)::value * 16;
                u32x2 v[16];
                static_for<0, 4>([&](auto W) {
                    constexpr int t = c0 + decltype(W)::value * 4;
                    uint32_t o0, o1, o2, o3;
                    byte_shl4(ccur[t / 4], 9u, o0, o1, o2, o3);
                    v[t - c0 + 0] = *(const u32x2 *)(lbase + o0 + (t + 0) * 8);
                    v[t - c0 + 1] = *(const u32x2 *)(lbase + o1 + (t + 1) * 8);
                    v[t - c0 + 2] = *(const u32x2 *)(lbase + o2 + (t + 2) * 8);
                    v[t - c0 + 3] = *(const u32x2 *)(lbase + o3 + (t + 3) * 8);
                });
                asm volatile("" ::: "memory");
                static_for<0, 16>([&](auto I) { acc += v[decltype(I)::value]; });
            });

            const uint32_t x0 = (thp.x - acc.x) & 0x80008000u, x1 = (thp.y - acc.y) & 0x80008000u;
            const unsigned long long anym = __ballot((x0 | x1) != 0) & vmask;
            bool flushed = false;
            if (anym && !(a.dbg_skip & 4)) {
                if (a.dbg && lane == 0) atomicAdd(a.dbg + 0, 1ull);
#pragma unroll
                for (int w = 0; w < 2; ++w) {
                    const uint32_t x = w ? x1 : x0;
                    if (__ballot(x != 0) & vmask) {
#pragma unroll
                        for (int half = 0; half < 2; ++half) {
                            const unsigned long long pm = __ballot((x & (half ? 0x80000000u : 0x8000u)) != 0) & vmask;
                            if (pm) {
                                const int n = __popcll(pm);
                                if (qcnt + n > 64) {
                                    qfilter_flush<M, SKEWED>(fc, queue_off + wave * 512, qcnt);
                                    qcnt = 0;
                                    flushed = true;
                                }
                                const int rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(pm >> 32),
                                                                           __builtin_amdgcn_mbcnt_lo((uint32_t)pm, 0u));
                                unsigned long long *queue = (unsigned long long *)(smem + queue_off + wave * 512);
                                if ((pm >> lane) & 1ull) queue[qcnt + rank] = ((unsigned long long)(w * 2 + half) << 32) | rid;
                                qcnt += n;
                            }
                        }
                    }
                }
            }
            if (qcnt && (qcnt >= 32 || ((step_no + wave * 4) & a.flush_mask) == a.flush_mask)) {
                qfilter_flush<M, SKEWED>(fc, queue_off + wave * 512, qcnt);
                qcnt = 0;
                flushed = true;
            }
            // import the bounds of the other slices (see adc_scan_qfilter_kernel)
            if (a.gkey) {
                const bool pow2 = ((step_no + 1) & step_no) == 0;
                if (pow2 || (step_no & 63) == 63) {
                    const int rw = pow2 ? (__builtin_ctz((unsigned)step_no + 1u) % NW) : ((step_no >> 6) % NW);
                    if (wave == rw) {
                        const int q = lane >> 3;
                        if (q < QT) {
                            const int b = tile * QT + q;
                            unsigned long long bound =
                                __hip_atomic_load(a.gkey + b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            if (a.gk2) {
#pragma unroll 1
                                for (int g0 = 0; g0 < a.n_slices; g0 += 8) {
                                    unsigned long long v = 0ull;
                                    if (g0 + (lane & 7) < a.n_slices)
                                        v = __hip_atomic_load(a.gk2 + (int64_t)b * a.n_slices + g0 + (lane & 7),
                                                              __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
                                    for (int o = 1; o < 8; o <<= 1) {
                                        const unsigned long long p = __shfl_xor(v, o);
                                        v = p > v ? p : v;
                                    }
                                    if (v != ~0ull && v + 1ull < bound) bound = v + 1ull;
                                }
                            }
                            if ((lane & 7) == 0 &&
                                bound < __hip_atomic_load(gkl + q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) {
                                __hip_atomic_store(gkl + q, bound, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                                const unsigned short nb = qbound_from_key<M>(bound, a.smax[b], a.qstep[b], a.qlo[b]);
                                if (nb < shq[q]) shq[q] = nb;
                            }
                        }
                    }
                }
            }
            if (flushed || (step_no & 3) == 3) {
                asm volatile("" ::: "memory");
                thp = *(const u32x2 *)(smem + shq_off);
            }
#pragma unroll
            for (int i = 0; i < CW; ++i) ccur[i] = cnext[i];
            if constexpr (!SKEWED) encode_plain(ccur);
            load_row(row0 + 2 * stride + lane, cnext);
            vcur = vnext;
            vnext = load_valid(row0 + 2 * stride + lane);
        }
        if (qcnt) qfilter_flush<M, SKEWED>(fc, queue_off + wave * 512, qcnt);

        __syncthreads();
        for (int q = wave; q < QT; q += NW) {
            const int b = tile * QT + q;
            if (b < a.B && lane <= km1)
                __hip_atomic_store(a.partial + ((int64_t)b * a.n_slices + slice) * a.k + lane, lists[q * 64 + lane],
                                   __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        if (a.tile_done) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            volatile unsigned int *s_flag = (volatile unsigned int *)(smem + lock_off);
            if (tid == 0) {
                const unsigned int old =
                    __hip_atomic_fetch_add(a.tile_done + tile, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                *s_flag = (old + 1u == (unsigned int)(a.n_slices - 1)) ? 1u : 0u;
            }
            __syncthreads();
            if (*s_flag) {
                for (int q = wave; q < QT; q += NW) {
                    const int b = tile * QT + q;
                    if (b >= a.B) continue;
                    WaveList L;
                    L.reset();
                    for (int sl = 0; sl < a.n_slices; ++sl) {
                        unsigned long long key = ~0ull;
                        if (lane <= km1)
                            key = __hip_atomic_load(a.partial + ((int64_t)b * a.n_slices + sl) * a.k + lane,
                                                    __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        wavelist_merge_sorted(L, (uint32_t)(key >> 32), (uint32_t)key, lane);
                    }
                    if (lane <= km1) {
                        const bool none = (L.hi == kKeyInfHi && L.lo == kIdNone);
                        const float d = none ? __builtin_inff() : ordered_to_f32(L.hi);
                        const int64_t id = none ? (int64_t)-1 : a.row_base + (int64_t)L.lo;
                        if (a.out_packed) {
                            a.out_packed[((int64_t)b * a.k + lane) * 2 + 0] = id;
                            a.out_packed[((int64_t)b * a.k + lane) * 2 + 1] = (int64_t)__float_as_uint(d);
                        } else {
                            a.out_d[(int64_t)b * a.k + lane] = a.sqrt_out ? __builtin_sqrtf(d) : d;
                            a.out_i[(int64_t)b * a.k + lane] = id;
                        }
                    }
                }
            }
            __syncthreads();
        }
    }
}

// quantisation of the fp32 TILED table [Bpad/4][Ks][64][4] for the M = 64 kernel: one workgroup per group of 4
// queries; entries [g4][Ks][64][4 x u16] (8 bytes)
__global__ __launch_bounds__(1024) void lut_quantise64_kernel(const float *__restrict__ lut, int Ks, int qmax,
                                                            uint16_t *__restrict__ out, float *__restrict__ qstep,
                                                            double *__restrict__ qlo, float *__restrict__ smax) {
    constexpr int M = 64, KPT = 16;  // 1024 threads: 16 codes per sweep
    __shared__ float s_lo[KPT][M][4], s_hi[KPT][M][4];
    __shared__ float s_step[4];
    const int tid = threadIdx.x;
    const int m = tid % M, kr = tid / M;
    const int g4 = blockIdx.x;
    const f32x4 *base = (const f32x4 *)lut + (int64_t)g4 * Ks * M;
    f32x4 mn = {__builtin_inff(), __builtin_inff(), __builtin_inff(), __builtin_inff()};
    f32x4 mx = -mn;
    for (int k = kr; k < Ks; k += KPT) {
        const f32x4 v = base[(int64_t)k * M + m];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            mn[i] = fminf(mn[i], v[i]);
            mx[i] = fmaxf(mx[i], v[i]);
        }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        s_lo[kr][m][i] = mn[i];
        s_hi[kr][m][i] = mx[i];
    }
    __syncthreads();
    if (tid < M * 4) {
        const int mm = tid / 4, i = tid % 4;
        float l = s_lo[0][mm][i], h = s_hi[0][mm][i];
        for (int r = 1; r < KPT; ++r) {
            l = fminf(l, s_lo[r][mm][i]);
            h = fmaxf(h, s_hi[r][mm][i]);
        }
        s_lo[0][mm][i] = l;
        s_hi[0][mm][i] = h;
    }
    __syncthreads();
    if (tid < 4) {
        float range = 0.f, sm = 0.f;
        double Lsum = 0.0;
        for (int mm = 0; mm < M; ++mm) {
            const float l = s_lo[0][mm][tid], h = s_hi[0][mm][tid];
            range = fmaxf(range, h - l);
            sm += fmaxf(fabsf(l), fabsf(h));
            Lsum += (double)l;
        }
        float step = range / (float)qmax;
        if (!(step > 0.f)) step = 1.f;
        s_step[tid] = step;
        const int b = g4 * 4 + tid;
        qstep[b] = step;
        qlo[b] = Lsum;
        smax[b] = sm;
    }
    __syncthreads();
    float lo_r[4], st_r[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        lo_r[i] = s_lo[0][m][i];
        st_r[i] = s_step[i];
    }
    u32x2 *o = (u32x2 *)out + (int64_t)g4 * Ks * M;
    for (int k = kr; k < Ks; k += KPT) {
        const f32x4 v = base[(int64_t)k * M + m];
        uint32_t q[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            float t = floorf((v[i] - lo_r[i]) / st_r[i]);
            if (!(t > 0.f)) t = 0.f;
            if (t > (float)qmax) t = (float)qmax;
            q[i] = (uint32_t)t;
        }
        o[(int64_t)k * M + m] = (u32x2){q[0] | (q[1] << 16), q[2] | (q[3] << 16)};
    }
}

// Seed bound: a valid upper bound of the final k-th key from the first S rows, so the scan starts with a
// filter that passes ~k/S of the rows instead of all of them (the cold-start "flood" cost 16 waves x 16
// queries x 64 uncoalesced gathers per work item).
// One 16-wave workgroup per group of 4 queries: their fp32 TILED rows [Ks][M][4] (64 KB at M=16) are
// staged in LDS, so ONE ds_read_b128 per (row, m) feeds the four exact ascending-m sums (gathering the
// same entries from L2 cost a 64-byte request per 4 useful bytes: 150 us for 4096 rows x 1024 queries).
// Selection without sorting networks: every lane keeps the MIN distance of the rows it saw; the 1024
// (wave, lane) groups are disjoint, so the k-th smallest of their minima has >= k distinct rows at or
// below it -- a valid bound, and equal to the exact k-th distance of the S rows unless two of the k best
// rows fell into one lane (3 % at S=4096, k=10; then it is the (k+1)-th).  The k-th smallest is found by
// rank counting over LDS broadcasts (each lane counts the keys below its own): 64 keys per wave, then
// 16*k candidates per query.  (Sorted wave lists + a 4-level merge tree: 52 of 62 us in bitonic networks.)
// The seed rows are scanned again by the main kernel: only the bound leaves this kernel.
constexpr int kSeedWaves = 16;
// QPB queries per workgroup: 4 (one fp32 TILED group) where their rows fit the LDS, 2 for M = 64
template <int M, bool SKEWED, int QPB>
__global__ __launch_bounds__(kSeedWaves * 64) void seed_bound_kernel(const uint8_t *__restrict__ codes, int64_t S,
                                                                    const uint32_t *__restrict__ valid,
                                                                    const float *__restrict__ lut, int B, int Ks, int k,
                                                                    unsigned long long *__restrict__ gkey) {
    constexpr int CW = M / 4;
    constexpr int CH = M < 16 ? M : 16;  // look-ups in flight
    typedef float fq __attribute__((ext_vector_type(QPB)));
    static_assert(QPB == 4 || QPB == 2, "queries per block");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    // [Ks][M + 1] x QPB queries: the pad entry spreads a fixed m over the banks (slot (code*(M+1) + m) % 16 =
    // (code + m) % 16); unpadded, all 64 lanes of a look-up hit ONE slot-bank (16-way conflict)
    fq *tab = (fq *)smem;
    unsigned long long *cand = (unsigned long long *)(smem + (size_t)Ks * (M + 1) * sizeof(fq));  // [kSeedWaves][k]
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    constexpr int BPG = 4 / QPB;                 // blocks per fp32 TILED group of 4 queries
    const int g4 = blockIdx.x / BPG, h = blockIdx.x % BPG;
    {
        const f32x4 *src = (const f32x4 *)lut + (int64_t)g4 * Ks * M;
        for (int i = tid; i < Ks * M; i += kSeedWaves * 64) {
            const f32x4 e = src[i];
            if constexpr (QPB == 4) tab[i + i / M] = e;
            else tab[i + i / M] = h ? (fq){e.z, e.w} : (fq){e.x, e.y};
        }
    }
    __syncthreads();
    // inverse skew rotation of this lane's rows (row % M == lane % M: the rows of a wave start at a multiple of 64)
    const int sinv = (M - lane % M) % M;
    const uint32_t bsh_inv = (uint32_t)(sinv & 3);
    bool abit_inv[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) abit_inv[i] = (((sinv >> 2) >> i) & 1) != 0;

    uint32_t best[QPB];  // ordered distance keys
#pragma unroll
    for (int q = 0; q < QPB; ++q) best[q] = 0xffffffffu;
    for (int64_t r0 = (int64_t)wave * 64; r0 < S; r0 += kSeedWaves * 64) {
        const int64_t r = r0 + lane;
        bool ok = r < S;
        if (ok && valid) ok = (valid[r >> 5] >> (r & 31)) & 1u;
        uint32_t c[CW];
        {
            const uint32_t *p = (const uint32_t *)(codes + (r < S ? r : S - 1) * M);
#pragma unroll
            for (int i = 0; i < CW; ++i) c[i] = p[i];
        }
        if constexpr (SKEWED && M == 64) {
#pragma unroll
            for (int i = 0; i < CW; ++i) c[i] = bytes_add(c[i], wrap64_mask(i, lane));  // undo the wrap coding
        }
        if constexpr (SKEWED) rotate_row<CW>(c, abit_inv, bsh_inv);
        fq d;
#pragma unroll
        for (int q = 0; q < QPB; ++q) d[q] = 0.f;
        static_for<0, M / CH>([&](auto C) {
            constexpr int m0 = decltype(C)::value * CH;
            fq v[CH];
#pragma unroll
            for (int i = 0; i < CH; ++i) {
                const uint32_t code = (c[(m0 + i) / 4] >> (8 * ((m0 + i) % 4))) & 0xffu;
                v[i] = tab[code * (M + 1) + m0 + i];
            }
#pragma unroll
            for (int i = 0; i < CH; ++i) d += v[i];  // ascending m: the reference's order
        });
#pragma unroll
        for (int q = 0; q < QPB; ++q) {
            const uint32_t key = f32_to_ordered(d[q]);
            if (ok && key < best[q]) best[q] = key;
        }
    }
    // Selection keys: the low 10 bits of the ordered distance are replaced by (wave, lane), which makes the
    // 1024 keys of a query unique (rank = number of smaller keys, no tie handling) and costs at most 1023
    // ulps of tightness: the k smallest keys T_i bound k distinct rows by (T_i | 1023).
    const int nc = kSeedWaves * k;  // candidates per query
    uint32_t *cand32 = (uint32_t *)cand;                                  // [kSeedWaves][k]
    uint32_t *wkeys = (uint32_t *)cand + kSeedWaves * 64 + wave * 64;     // this wave's 64 lane minima
#pragma unroll 1
    for (int q = 0; q < QPB; ++q) {
        uint32_t mine = best[0];
#pragma unroll
        for (int qq = 1; qq < QPB; ++qq)
            if (q == qq) mine = best[qq];
        mine = (mine & ~1023u) | (uint32_t)(wave << 6) | (uint32_t)lane;
        // rank among the wave's 64: all lanes read the 64 keys back with wave-uniform addresses (broadcast)
        wkeys[lane] = mine;
        int rank = 0;
#pragma unroll
        for (int j = 0; j < 64; j += 4) {
            const u32x4 o = *(const u32x4 *)(wkeys + j);
            rank += (o.x < mine) + (o.y < mine) + (o.z < mine) + (o.w < mine);
        }
        if (rank < k) cand32[wave * k + rank] = mine;
        __syncthreads();
        // the k-th smallest of the 16*k candidates, same way: thread t ranks candidate t
        for (int t = tid; t < nc; t += kSeedWaves * 64) {
            const uint32_t me = cand32[t];
            int rk = 0;
#pragma unroll 8
            for (int j = 0; j < nc; ++j) rk += cand32[j] < me;
            const int b = g4 * 4 + h * QPB + q;
            // the bound admits every row at or below (me | 1023), whatever its id; +1 in the distance field
            // because the seed rows are in nobody's list: the scan must still ACCEPT the rows that set it
            if (rk == k - 1 && b < B && (me | 1023u) != 0xffffffffu)
                gkey[b] = ((unsigned long long)(me | 1023u) + 1ull) << 32;
        }
        __syncthreads();
    }
}

// ---- quantisation of the fp32 TILED table [Bpad/4][Ks][M][4] -------------------------------------
// pass 1: lo/hi per (query, sub-space): one wave per (group of 4 queries, m)
__global__ __launch_bounds__(256) void lut_minmax_kernel(const float *__restrict__ lut, int n_g4, int M, int Ks,
                                                        float *__restrict__ lo, float *__restrict__ hi) {
    const int lane = threadIdx.x & 63;
    const int w = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (w >= n_g4 * M) return;
    const int g = w / M, m = w - g * M;
    const f32x4 *base = (const f32x4 *)lut + (int64_t)g * Ks * M + m;
    f32x4 mn = {__builtin_inff(), __builtin_inff(), __builtin_inff(), __builtin_inff()};
    f32x4 mx = -mn;
    for (int k = lane; k < Ks; k += 64) {
        const f32x4 v = base[(int64_t)k * M];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            mn[i] = fminf(mn[i], v[i]);
            mx[i] = fmaxf(mx[i], v[i]);
        }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            mn[i] = fminf(mn[i], __shfl_xor(mn[i], o));
            mx[i] = fmaxf(mx[i], __shfl_xor(mx[i], o));
        }
    }
    if (lane < 4) {
        lo[(int64_t)(g * 4 + lane) * M + m] = mn[lane];
        hi[(int64_t)(g * 4 + lane) * M + m] = mx[lane];
    }
}
// pass 2: per query step / L / Smax
__global__ __launch_bounds__(256) void lut_qparams_kernel(const float *__restrict__ lo, const float *__restrict__ hi,
                                                         int Bpad, int M, int qmax, float *__restrict__ qstep,
                                                         double *__restrict__ qlo, float *__restrict__ smax) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= Bpad) return;
    float range = 0.f, sm = 0.f;
    double L = 0.0;
    for (int m = 0; m < M; ++m) {
        const float l = lo[(int64_t)b * M + m], h = hi[(int64_t)b * M + m];
        range = fmaxf(range, h - l);
        sm += fmaxf(fabsf(l), fabsf(h));
        L += (double)l;
    }
    float step = range / (float)qmax;
    if (!(step > 0.f)) step = 1.f;
    qstep[b] = step;
    qlo[b] = L;
    smax[b] = sm;
}
// pass 3: quantise; one thread per (group of 8 queries, k, m) -> one 16-byte store
__global__ __launch_bounds__(256) void lut_quant_kernel(const float *__restrict__ lut, int n_g8, int M, int Ks,
                                                       const float *__restrict__ lo, const float *__restrict__ qstep,
                                                       int qmax, uint16_t *__restrict__ out) {
    const int64_t id = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (id >= (int64_t)n_g8 * Ks * M) return;
    const int m = (int)(id % M);
    const int k = (int)((id / M) % Ks);
    const int g = (int)(id / ((int64_t)M * Ks));
    uint32_t pk[4];
#pragma unroll
    for (int half = 0; half < 2; ++half) {
        const f32x4 v = ((const f32x4 *)lut)[((int64_t)(g * 2 + half) * Ks + k) * M + m];
        uint32_t q[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int b = g * 8 + half * 4 + i;
            float t = floorf((v[i] - lo[(int64_t)b * M + m]) / qstep[b]);
            if (!(t > 0.f)) t = 0.f;
            if (t > (float)qmax) t = (float)qmax;
            q[i] = (uint32_t)t;
        }
        pk[half * 2 + 0] = q[0] | (q[1] << 16);
        pk[half * 2 + 1] = q[2] | (q[3] << 16);
    }
    ((u32x4 *)out)[id] = (u32x4){pk[0], pk[1], pk[2], pk[3]};
}

// The three passes above in one launch: one workgroup per group of 8 queries (two fp32 TILED groups of 4).
// Thread t owns sub-space m = t % M of codes k = t / M, t / M + 256 / M, ...: per-(query, m) min/max by an
// LDS tree, then (step, L, Smax) per query, then the 16-byte quantised entries.
template <int M>
__global__ __launch_bounds__(256) void lut_quantise_fused_kernel(const float *__restrict__ lut, int Ks, int qmax,
                                                                uint16_t *__restrict__ out,
                                                                float *__restrict__ qstep, double *__restrict__ qlo,
                                                                float *__restrict__ smax, u32x4 *__restrict__ fill,
                                                                int64_t fill_vec16) {
    constexpr int KPT = 256 / M;  // codes covered per sweep of the block
    // this launch also resets the scan's result lists and shared bounds to "none" (all-ones): one launch less
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < fill_vec16; i += (int64_t)gridDim.x * 256)
        fill[i] = (u32x4){~0u, ~0u, ~0u, ~0u};
    __shared__ float s_lo[KPT][M][8], s_hi[KPT][M][8];
    __shared__ float s_step[8];
    const int tid = threadIdx.x;
    const int m = tid % M, kr = tid / M;
    const int g8 = blockIdx.x;
    const f32x4 *base0 = (const f32x4 *)lut + (int64_t)(g8 * 2) * Ks * M;
    const f32x4 *base1 = base0 + (int64_t)Ks * M;
    float mn[8], mx[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        mn[i] = __builtin_inff();
        mx[i] = -__builtin_inff();
    }
    for (int k = kr; k < Ks; k += KPT) {
        const f32x4 v0 = base0[(int64_t)k * M + m], v1 = base1[(int64_t)k * M + m];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            mn[i] = fminf(mn[i], v0[i]);
            mx[i] = fmaxf(mx[i], v0[i]);
            mn[4 + i] = fminf(mn[4 + i], v1[i]);
            mx[4 + i] = fmaxf(mx[4 + i], v1[i]);
        }
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        s_lo[kr][m][i] = mn[i];
        s_hi[kr][m][i] = mx[i];
    }
    __syncthreads();
    // thread (m, i) for tid < M*8 folds the KPT partials
    if (tid < M * 8) {
        const int mm = tid / 8, i = tid % 8;
        float l = s_lo[0][mm][i], h = s_hi[0][mm][i];
        for (int r = 1; r < KPT; ++r) {
            l = fminf(l, s_lo[r][mm][i]);
            h = fmaxf(h, s_hi[r][mm][i]);
        }
        s_lo[0][mm][i] = l;
        s_hi[0][mm][i] = h;
    }
    __syncthreads();
    if (tid < 8) {
        float range = 0.f, sm = 0.f;
        double Lsum = 0.0;
        for (int mm = 0; mm < M; ++mm) {
            const float l = s_lo[0][mm][tid], h = s_hi[0][mm][tid];
            range = fmaxf(range, h - l);
            sm += fmaxf(fabsf(l), fabsf(h));
            Lsum += (double)l;
        }
        float step = range / (float)qmax;
        if (!(step > 0.f)) step = 1.f;
        s_step[tid] = step;
        const int b = g8 * 8 + tid;
        qstep[b] = step;
        qlo[b] = Lsum;
        smax[b] = sm;
    }
    __syncthreads();
    float lo_r[8], st_r[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        lo_r[i] = s_lo[0][m][i];
        st_r[i] = s_step[i];
    }
    u32x4 *o = (u32x4 *)out + (int64_t)g8 * Ks * M;
    for (int k = kr; k < Ks; k += KPT) {
        const f32x4 v0 = base0[(int64_t)k * M + m], v1 = base1[(int64_t)k * M + m];
        uint32_t q[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const float v = i < 4 ? v0[i] : v1[i - 4];
            float t = floorf((v - lo_r[i]) / st_r[i]);
            if (!(t > 0.f)) t = 0.f;
            if (t > (float)qmax) t = (float)qmax;
            q[i] = (uint32_t)t;
        }
        o[(int64_t)k * M + m] = (u32x4){q[0] | (q[1] << 16), q[2] | (q[3] << 16), q[4] | (q[5] << 16), q[6] | (q[7] << 16)};
    }
}

// annlite_pq_search_topk on the quantised-filter plan: the L2 tables are BUILT, reduced and quantised by one
// launch (query batch in, neighbours out).  One 1024-thread workgroup per group of 8 queries; thread
// (kr, m) = (tid / M, tid % M) owns sub-space m of codes kr, kr + 1024/M, ...: NSW = M/4 entries x 8
// queries stay in registers between the min/max pass and the quantisation, nothing is read back.
// entry = the reference's j-ascending fmaf chain over (codeword - query) (pq_bindings.pyx:204-206): the same
// bits as lut_l2_tiled_kernel.  Also resets the scan's result lists / shared bounds (fill).
template <int M>
__global__ __launch_bounds__(1024) void lut_l2_build_quantise_kernel(const float *__restrict__ queries, int B, int D,
                                                                    const float *__restrict__ cb, int Ks,
                                                                    float *__restrict__ lut, int qmax,
                                                                    uint16_t *__restrict__ out,
                                                                    float *__restrict__ qstep, double *__restrict__ qlo,
                                                                    float *__restrict__ smax, u32x4 *__restrict__ fill,
                                                                    int64_t fill_vec16) {
    constexpr int KPT = 1024 / M;   // codes per sweep
    constexpr int NSW = 256 / KPT;  // sweeps (Ks <= 256)
    for (int64_t i = (int64_t)blockIdx.x * 1024 + threadIdx.x; i < fill_vec16; i += (int64_t)gridDim.x * 1024)
        fill[i] = (u32x4){~0u, ~0u, ~0u, ~0u};
    __shared__ float s_lo[16][M][8], s_hi[16][M][8];
    __shared__ float s_step[8];
    extern __shared__ __attribute__((aligned(16))) unsigned char dyn_smem[];
    float *s_q = (float *)dyn_smem;  // the 8 queries, [8][D]
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int m = tid % M, kr = tid / M;
    const int g8 = blockIdx.x;
    const int dsub = D / M;
    for (int i = tid; i < 8 * D; i += 1024) {
        const int b = g8 * 8 + i / D;
        s_q[i] = b < B ? queries[(int64_t)b * D + i % D] : 0.f;
    }
    __syncthreads();
    f32x4 *base0 = (f32x4 *)lut + (int64_t)(g8 * 2) * Ks * M;
    f32x4 *base1 = base0 + (int64_t)Ks * M;
    f32x4 v0[NSW], v1[NSW];
    float mn[8], mx[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        mn[i] = __builtin_inff();
        mx[i] = -__builtin_inff();
    }
#pragma unroll
    for (int sw = 0; sw < NSW; ++sw) {
        const int k = kr + sw * KPT;
        float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        if (k < Ks) {
            const float *cw = cb + ((int64_t)m * Ks + k) * dsub;
            for (int j = 0; j < dsub; j += 4) {
                const f32x4 cj = *(const f32x4 *)(cw + j);
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const f32x4 qj = *(const f32x4 *)(s_q + i * D + m * dsub + j);
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float c = cj[e] - qj[e];
                        acc[i] = __builtin_fmaf(c, c, acc[i]);
                    }
                }
            }
#pragma unroll
            for (int i = 0; i < 8; ++i)
                if (g8 * 8 + i >= B) acc[i] = 0.f;  // pad queries -> 0, like lut_l2_tiled_kernel
            v0[sw] = (f32x4){acc[0], acc[1], acc[2], acc[3]};
            v1[sw] = (f32x4){acc[4], acc[5], acc[6], acc[7]};
            base0[(int64_t)k * M + m] = v0[sw];
            base1[(int64_t)k * M + m] = v1[sw];
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                mn[i] = fminf(mn[i], acc[i]);
                mx[i] = fmaxf(mx[i], acc[i]);
            }
        }
    }
    // lanes l, l + M, l + 2M, ... of a wave share m
#pragma unroll
    for (int o = M; o < 64; o <<= 1) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            mn[i] = fminf(mn[i], __shfl_xor(mn[i], o));
            mx[i] = fmaxf(mx[i], __shfl_xor(mx[i], o));
        }
    }
    if (lane < M) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            s_lo[wave][m][i] = mn[i];
            s_hi[wave][m][i] = mx[i];
        }
    }
    __syncthreads();
    if (tid < M * 8) {
        const int mm = tid / 8, i = tid % 8;
        float l = s_lo[0][mm][i], h = s_hi[0][mm][i];
        for (int r = 1; r < 16; ++r) {
            l = fminf(l, s_lo[r][mm][i]);
            h = fmaxf(h, s_hi[r][mm][i]);
        }
        s_lo[0][mm][i] = l;
        s_hi[0][mm][i] = h;
    }
    __syncthreads();
    if (tid < 8) {
        float range = 0.f, sm = 0.f;
        double Lsum = 0.0;
        for (int mm = 0; mm < M; ++mm) {
            const float l = s_lo[0][mm][tid], h = s_hi[0][mm][tid];
            range = fmaxf(range, h - l);
            sm += fmaxf(fabsf(l), fabsf(h));
            Lsum += (double)l;
        }
        float step = range / (float)qmax;
        if (!(step > 0.f)) step = 1.f;
        s_step[tid] = step;
        const int b = g8 * 8 + tid;
        qstep[b] = step;
        qlo[b] = Lsum;
        smax[b] = sm;
    }
    __syncthreads();
    float lo_r[8], st_r[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        lo_r[i] = s_lo[0][m][i];
        st_r[i] = s_step[i];
    }
    u32x4 *o = (u32x4 *)out + (int64_t)g8 * Ks * M;
#pragma unroll
    for (int sw = 0; sw < NSW; ++sw) {
        const int k = kr + sw * KPT;
        if (k >= Ks) continue;
        uint32_t q[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const float v = i < 4 ? v0[sw][i] : v1[sw][i - 4];
            float t = floorf((v - lo_r[i]) / st_r[i]);
            if (!(t > 0.f)) t = 0.f;
            if (t > (float)qmax) t = (float)qmax;
            q[i] = (uint32_t)t;
        }
        o[(int64_t)k * M + m] = (u32x4){q[0] | (q[1] << 16), q[2] | (q[3] << 16), q[4] | (q[5] << 16), q[6] | (q[7] << 16)};
    }
}

// Smax[b] = sum_m max_k |lut[b][m][k]| from the TILED table [Bpad/QI][Ks][M][QI]; one wave per group
__global__ __launch_bounds__(256) void lut_smax_kernel(const float *__restrict__ lut, int n_groups, int M, int Ks,
                                                      int QI, float *__restrict__ smax) {
    const int lane = threadIdx.x & 63;
    const int g = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (g >= n_groups) return;
    const float *base = lut + (int64_t)g * Ks * M * QI;
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    for (int m = 0; m < M; ++m) {
        float mx[4] = {0.f, 0.f, 0.f, 0.f};
        for (int k = lane; k < Ks; k += 64)
            for (int i = 0; i < QI; ++i) mx[i] = fmaxf(mx[i], fabsf(base[((int64_t)k * M + m) * QI + i]));
        for (int i = 0; i < QI; ++i) {
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) mx[i] = fmaxf(mx[i], __shfl_xor(mx[i], o));
            acc[i] += mx[i];
        }
    }
    if (lane == 0)
        for (int i = 0; i < QI; ++i) smax[g * QI + i] = acc[i];
}

// =================================================================================================
// Generic kernel: any M / Ks / code width (1,2,4 bytes), k <= 64.  One query per workgroup, the
// table is read through L2 in the reference's [B][M][Ks] layout.  Correct for every shape the
// reference accepts; used when no fast instantiation exists (e.g. Ks > 256 => uint16 codes).
// =================================================================================================
template <typename CODE_T, int NW>
__global__ __launch_bounds__(NW * 64) void adc_scan_generic_kernel(const ScanArgs a, int M) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int km1 = a.k - 1;
    const int n_items = a.n_tiles * a.n_slices;
    for (int item = blockIdx.x; item < n_items; item += gridDim.x) {
        const int b = item % a.n_tiles;  // QT == 1
        const int slice = item / a.n_tiles;
        const float *lut = a.lut + (int64_t)b * M * a.Ks;
        WaveList L;
        L.reset();
        uint32_t th = kKeyInfHi, tl = kIdNone;
        float tf = __builtin_inff();
        const int64_t slice_begin = (int64_t)slice * a.slice_rows;
        int64_t slice_end = slice_begin + a.slice_rows;
        if (slice_end > a.N) slice_end = a.N;
        const CODE_T *codes = (const CODE_T *)a.codes;
        for (int64_t row0 = slice_begin + (int64_t)wave * 64; row0 < slice_end; row0 += (int64_t)NW * 64) {
            int64_t row = row0 + lane;
            const bool inb = row < slice_end;
            if (!inb) row = a.N - 1;
            const CODE_T *cr = codes + row * M;
            float d = 0.f;
            for (int m = 0; m < M; ++m) d += lut[(int64_t)m * a.Ks + (int64_t)cr[m]];
            unsigned long long vmask = __ballot(inb);
            if (a.valid) {
                const uint32_t *vw = a.valid + (row0 >> 5);
                unsigned long long vb = (unsigned long long)vw[0];
                if (row0 + 32 < a.N) vb |= (unsigned long long)vw[1] << 32;
                vmask &= vb;
            }
            const unsigned long long pm = __ballot(d <= tf) & vmask;
            if (pm) {
                wavelist_offer(L, pm, f32_to_ordered(d), (uint32_t)(row0 + lane), km1, th, tl, lane);
                tf = (th == kKeyInfHi) ? __builtin_inff() : ordered_to_f32(th);
            }
        }
        __syncthreads();
        unsigned long long *scratch = (unsigned long long *)smem;  // [NW][64]
        scratch[wave * 64 + lane] = ((unsigned long long)L.hi << 32) | L.lo;
        __syncthreads();
        if (wave == 0) {
            for (int w = 1; w < NW; ++w) {
                const unsigned long long key = scratch[w * 64 + lane];
                const uint32_t chi = (uint32_t)(key >> 32), clo = (uint32_t)key;
                const unsigned long long pm = __ballot(lane <= km1 && key_less(chi, clo, th, tl));
                wavelist_offer(L, pm, chi, clo, km1, th, tl, lane);
            }
            if (lane <= km1)
                a.partial[((int64_t)b * a.n_slices + slice) * a.k + lane] = ((unsigned long long)L.hi << 32) | L.lo;
        }
        __syncthreads();
    }
}

// =================================================================================================
// Final merge: partial keys [B][NS][k] -> (dist f32, id i64) [B][k]; one wave per query.
// =================================================================================================
// With `out_packed` the result is written as [B][k][2] int64 (global id, distance bits) instead: ONE buffer,
// ONE all-gather per batch in the row-sharded search.
__global__ __launch_bounds__(256) void merge_partial_kernel(const unsigned long long *partial, int B, int NS,
                                                           int k, int64_t row_base, float *out_d,
                                                           int64_t *out_i, int64_t *out_packed, int sqrt_out) {
    const int lane = threadIdx.x & 63;
    const int b = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (b >= B) return;
    const int km1 = k - 1;
    const unsigned long long *src = partial + (int64_t)b * NS * k;
    const int total = NS * k;
    WaveList L;
    L.reset();
    uint32_t th = kKeyInfHi, tl = kIdNone;
    for (int base = 0; base < total; base += 64) {
        const int i = base + lane;
        unsigned long long key = ~0ull;
        if (i < total) key = src[i];
        const uint32_t chi = (uint32_t)(key >> 32), clo = (uint32_t)key;
        const unsigned long long pm = __ballot(key_less(chi, clo, th, tl));
        wavelist_offer(L, pm, chi, clo, km1, th, tl, lane);
    }
    if (lane <= km1) {
        const bool none = (L.hi == kKeyInfHi && L.lo == kIdNone);
        const float d = none ? __builtin_inff() : ordered_to_f32(L.hi);
        const int64_t id = none ? (int64_t)-1 : row_base + (int64_t)L.lo;
        if (out_packed) {
            out_packed[((int64_t)b * k + lane) * 2 + 0] = id;
            out_packed[((int64_t)b * k + lane) * 2 + 1] = (int64_t)__float_as_uint(d);
        } else {
            out_d[(int64_t)b * k + lane] = sqrt_out ? __builtin_sqrtf(d) : d;
            out_i[(int64_t)b * k + lane] = id;
        }
    }
}

// Unmerged candidates: partial keys [B][NS][k] -> (dist, id) [B][NS*k]  (rerank candidate generator)
__global__ __launch_bounds__(256) void export_partial_kernel(const unsigned long long *partial, int64_t total,
                                                            int64_t row_base, float *out_d, int64_t *out_i) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const unsigned long long key = partial[i];
    const uint32_t hi = (uint32_t)(key >> 32), lo = (uint32_t)key;
    const bool none = (hi == kKeyInfHi && lo == kIdNone);
    out_d[i] = none ? __builtin_inff() : ordered_to_f32(hi);
    out_i[i] = none ? (int64_t)-1 : row_base + (int64_t)lo;
}

// Merge G lists [G][B][k] of (dist, id) -> [B][k]  (after the RCCL all-gather).  One wave per query.
// `packed` != NULL: the lists come as [G][B][k][2] int64 (id, distance bits) -- merge_partial_kernel's packed form.
__global__ __launch_bounds__(256) void merge_lists_kernel(const float *dist, const int64_t *id, const int64_t *packed,
                                                         int G, int B, int k, float *out_d, int64_t *out_i,
                                                         int sqrt_out) {
    const int lane = threadIdx.x & 63;
    const int b = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (b >= B) return;
    const int km1 = k - 1;
    // ids are 64-bit here (global row ids of different shards), so the list is kept directly as
    // (float dist, int64 id) per lane; candidates are offered one by one (G*k is small).
    const int total = G * k;
    float ld = __builtin_inff();
    int64_t li = INT64_MAX;
    for (int c = 0; c < total; ++c) {
        const int g = c / k, j = c - g * k;
        const int64_t e = ((int64_t)g * B + b) * k + j;
        const float cd = packed ? __uint_as_float((uint32_t)packed[e * 2 + 1]) : dist[e];
        const int64_t ci = packed ? packed[e * 2] : id[e];
        if (ci < 0) continue;  // padding entry of a short shard
        const bool less = (ld < cd) || (ld == cd && li < ci);
        const int pos = __popcll(__ballot(less));
        if (pos > km1) continue;
        const float sd = __shfl_up(ld, 1);
        const int64_t si = __shfl_up(li, 1);
        if (lane == pos) {
            ld = cd;
            li = ci;
        } else if (lane > pos) {
            ld = sd;
            li = si;
        }
    }
    if (lane <= km1) {
        const bool none = (li == INT64_MAX);
        out_d[(int64_t)b * k + lane] = none ? __builtin_inff() : (sqrt_out ? __builtin_sqrtf(ld) : ld);
        out_i[(int64_t)b * k + lane] = none ? (int64_t)-1 : li;
    }
}

// Row-wise top-k of a dense matrix (annlite/math.py:94-120 with the fixed tie-break); wave per row.
__global__ __launch_bounds__(256) void topk_rows_kernel(const float *values, int B, int64_t N, int k,
                                                       int64_t id_base, float *out_d, int64_t *out_i) {
    const int lane = threadIdx.x & 63;
    const int b = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (b >= B) return;
    const int km1 = k - 1;
    const float *src = values + (int64_t)b * N;
    WaveList L;
    L.reset();
    uint32_t th = kKeyInfHi, tl = kIdNone;
    float tf = __builtin_inff();
    for (int64_t base = 0; base < N; base += 64) {
        const int64_t i = base + lane;
        const float d = (i < N) ? src[i] : __builtin_inff();
        const unsigned long long pm = __ballot(i < N && d <= tf);
        if (pm) {
            wavelist_offer(L, pm, f32_to_ordered(d), (uint32_t)i, km1, th, tl, lane);
            tf = (th == kKeyInfHi) ? __builtin_inff() : ordered_to_f32(th);
        }
    }
    if (lane <= km1) {
        const bool none = (L.hi == kKeyInfHi && L.lo == kIdNone);
        out_d[(int64_t)b * k + lane] = none ? __builtin_inff() : ordered_to_f32(L.hi);
        out_i[(int64_t)b * k + lane] = none ? (int64_t)-1 : id_base + (int64_t)L.lo;
    }
}

// =================================================================================================
// all-distances scan (operator seam) and gathered ADC
// =================================================================================================
template <typename CODE_T>
__global__ __launch_bounds__(256) void adc_dist_kernel(const float *adtable, int M, int Ks, const CODE_T *codes,
                                                      int64_t N, float *out) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float *t = (float *)smem;
    const bool in_lds = (int64_t)M * Ks * 4 <= 64 * 1024;
    if (in_lds) {
        for (int i = threadIdx.x; i < M * Ks; i += blockDim.x) t[i] = adtable[i];
        __syncthreads();
    }
    const float *tab = in_lds ? t : adtable;
    for (int64_t n = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; n < N; n += (int64_t)gridDim.x * blockDim.x) {
        const CODE_T *c = codes + n * M;
        float d = 0.f;
        for (int m = 0; m < M; ++m) d += tab[m * Ks + (int)c[m]];
        out[n] = d;
    }
}

template <typename CODE_T>
__global__ __launch_bounds__(256) void adc_gather_kernel(const float *lut, int B, int M, int Ks, const CODE_T *codes,
                                                        int64_t N, const int64_t *cand, int R, float *out) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (int64_t)B * R) return;
    const int b = (int)(i / R);
    const int64_t row = cand[i];
    if (row < 0 || row >= N) {
        out[i] = __builtin_inff();
        return;
    }
    const float *tab = lut + (int64_t)b * M * Ks;
    const CODE_T *c = codes + row * M;
    float d = 0.f;
    for (int m = 0; m < M; ++m) d += tab[m * Ks + (int)c[m]];
    out[i] = d;
}

// PLAIN <-> SKEWED code-table conversion (uint8): one thread per code byte
__global__ __launch_bounds__(256) void codes_skew_kernel(const uint8_t *__restrict__ in, int64_t N, int M,
                                                        const int64_t *__restrict__ ids, int64_t id_base,
                                                        uint8_t *__restrict__ out, int inverse) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= N * M) return;
    const int64_t i = t / M;
    const int j = (int)(t - i * M);
    const int64_t id = ids ? ids[i] : id_base + i;
    const int r = (int)(id % M);
    if (!inverse) {
        uint8_t v = in[i * M + (j + r) % M];
        if (M == 64 && j + r >= 64) v = (uint8_t)(v - 1);  // wrap-coded layout of the M = 64 kernel (see wrap64_mask)
        out[id * M + j] = v;
    } else {
        const int jj = ((j - r) % M + M) % M;  // stored position of sub-space j
        uint8_t v = in[id * M + jj];
        if (M == 64 && jj + r >= 64) v = (uint8_t)(v + 1);
        out[i * M + j] = v;
    }
}

// =================================================================================================
// host side
// =================================================================================================
struct FastCfg {
    int M, QI, NQ, NW, WPS, wg_per_cu, id;
    int mode;  // 0: exec-masked passes, 1/2: weight-fma passes, 3: FILTER kernel (fast fp32 sum + exact
               // recompute), 4: QFILTER kernel (12-bit integer tables, 8 queries per LDS entry)
    int qt() const { return (mode == 4 ? (M == 64 ? 4 : 8) : QI) * NQ; }
};

// Kernel variants per M.  ANNLITE_SCAN_VARIANT (env, read per call) selects among the M=16
// instantiations for A/B measurements; variant 0 is the default for every M.
static int scan_variant() {
    const char *e = getenv("ANNLITE_SCAN_VARIANT");
    return e ? atoi(e) : 0;
}

static bool fast_cfg(int64_t M, int64_t Ks, int code_bytes, int64_t k, FastCfg *c) {
    if (code_bytes != 1 || Ks > 256 || Ks < 1 || k > 64 || k < 1) return false;
    const int v = scan_variant();
    switch (M) {
        case 8:
            if (v >= 20 && v < 30) *c = {8, 4, 2, 8, 2, 1, 80, 0};
            else if (v == 9) *c = {8, 4, 2, 8, 2, 1, 81, 3};
            else *c = {8, 4, 2, 16, 4, 1, 830, 4};               // default: qfilter, 16 queries / WG
            return true;
        case 16:
            if (v == 0) { *c = {16, 4, 2, 16, 4, 1, 1631, 4}; return true; }  // default: qfilter, 16 queries / WG, 16 waves
            if (v == 8) { *c = {16, 4, 2, 12, 3, 1, 1601, 3}; return true; }  // fp32 filter kernel, 12 waves, single buffer
            if (v == 30) { *c = {16, 4, 2, 12, 3, 1, 1630, 4}; return true; }  // qfilter, 16 queries / WG, 12 waves
            if (v == 31) { *c = {16, 4, 2, 16, 4, 1, 1631, 4}; return true; }  // qfilter, 16 waves
            if (v == 32) { *c = {16, 4, 2, 8, 2, 1, 1632, 4}; return true; }   // qfilter, 8 waves
            if (v == 9) { *c = {16, 4, 2, 8, 2, 1, 1600, 3}; return true; }   // filter kernel, 8 waves, double buffer
            if (v == 10) { *c = {16, 4, 2, 12, 3, 1, 1601, 3}; return true; }  // filter kernel, 12 waves, single buffer
            if (v == 11) { *c = {16, 4, 1, 8, 4, 2, 1602, 3}; return true; }   // filter kernel, QT=4, 2 WG / CU
            if (v == 12) { *c = {16, 4, 2, 16, 4, 1, 1603, 3}; return true; }  // filter kernel, 16 waves, single buffer
            if (v == 1) *c = {16, 4, 1, 8, 4, 2, 161, 0};        // QT=4, 2 workgroups / CU
            else if (v == 2) *c = {16, 4, 2, 16, 4, 1, 162, 0};  // QT=8, 16 waves
            else if (v == 3) *c = {16, 4, 2, 12, 3, 1, 163, 0};  // QT=8, 12 waves (3 / SIMD)
            else if (v == 4) *c = {16, 4, 2, 8, 2, 1, 164, 1};   // QT=8, 8 waves, weight-fma
            else if (v == 5) *c = {16, 4, 2, 12, 3, 1, 165, 1};  // QT=8, 12 waves, weight-fma
            else if (v == 6) *c = {16, 4, 1, 8, 4, 2, 166, 1};   // QT=4, 2 WG / CU, weight-fma
            else if (v == 7) *c = {16, 4, 2, 8, 2, 1, 167, 2};   // QT=8, 8 waves, scalar weight-fma
            else if (v == 28) *c = {16, 4, 2, 8, 2, 1, 160, 0};  // QT=8, 8 waves, 1 workgroup / CU
            else *c = {16, 4, 2, 12, 3, 1, 163, 0};              // (v >= 20) two-pass kernel, QT=8, 12 waves
            return true;
        case 32:
            if (v >= 20 && v < 30) *c = {32, 4, 1, 8, 2, 1, 320, 0};
            else if (v == 8 || v == 9) *c = {32, 4, 1, 8, 2, 1, 321, 3};
            else *c = {32, 4, 1, 12, 3, 1, 3230, 4};             // default: qfilter, 8 queries / WG
            return true;
        case 64:
            if (v >= 20 && v < 30) *c = {64, 2, 1, 8, 2, 1, 640, 0};  // two-pass kernel (PLAIN tables only: SKEWED is wrap-coded)
            else if (v == 30) *c = {64, 4, 1, 16, 4, 1, 6430, 4};     // qfilter64, 16 waves (spills: 7.4 ms at C4 vs 5.9)
            else if (v == 32) *c = {64, 4, 1, 8, 2, 1, 6432, 4};      // qfilter64, 8 waves (6.7 ms)
            else *c = {64, 4, 1, 12, 3, 1, 6431, 4};                  // default: qfilter64, 4 queries / WG, 12 waves
            return true;
        default: return false;
    }
}

static int round_up(int64_t x, int64_t m) { return (int)(((x + m - 1) / m) * m); }

static void plan_slices(int64_t N, int n_tiles, int waves, int n_cu, bool xcd8, int *n_slices, int64_t *slice_rows) {
    // at least one slice per XCD; more when few query tiles exist, as long as every wave keeps
    // >= 8 steps of 64 rows
    const int64_t min_rows = (int64_t)waves * 64 * 8;
    // XCD-mapped plans: ONE work item per CU when the tiles allow it (every (query, slice) list pays its own
    // logarithmic number of candidate events: 4 slices instead of 8 was 10% faster at 1.25M rows x 1024
    // queries); 1, 2, 4 or a multiple of 8 slices (item_map)
    int64_t want = ((xcd8 ? 1 : 2) * (int64_t)n_cu + n_tiles - 1) / n_tiles;
    int64_t cap = N / min_rows;
    if (want > cap) want = cap;
    int64_t ns = want;
    if (xcd8) ns = want <= 1 ? 1 : want <= 2 ? 2 : want <= 4 ? 4 : ((want + 7) / 8) * 8;
    if (ns < 1) ns = 1;
    if (xcd8) {
        if (const char *e = getenv("ANNLITE_SCAN_SLICES")) {
            const int64_t v = atoll(e);
            if (v == 1 || v == 2 || v == 4 || (v >= 8 && v % 8 == 0 && v <= 4096)) ns = v;
        }
    }
    int64_t rows = (N + ns - 1) / ns;
    rows = ((rows + 63) / 64) * 64;
    if (rows < 64) rows = 64;
    *n_slices = (int)ns;
    *slice_rows = rows;
}

}  // namespace annlite

using namespace annlite;

template <int M, int NQ, int NW, int WPS, bool SKEWED>
static int launch_qfilter(const ScanArgs &a, int grid, hipStream_t st) {
    const size_t lds_lut = (size_t)a.Ks * NQ * M * 16;
    const size_t need = lds_lut + 256 + (size_t)8 * NQ * 64 * 8 + 128 + (size_t)NW * 512;
    auto fn = adc_scan_qfilter_kernel<M, NQ, NW, WPS, SKEWED>;
    ANNLITE_HIP_TRY(hipFuncSetAttribute((const void *)fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)need));
    hipLaunchKernelGGL(fn, dim3(grid), dim3(NW * 64), need, st, a);
    return launch_status("adc_scan_qfilter_kernel");
}

template <int NW, bool SKEWED>
static int launch_qfilter64(const ScanArgs &a, int grid, hipStream_t st) {
    const size_t need = (size_t)(a.Ks + 1) * 512 + 256 + (size_t)4 * 512 + 128 + (size_t)NW * 512;
    auto fn = adc_scan_qfilter64_kernel<NW, SKEWED>;
    ANNLITE_HIP_TRY(hipFuncSetAttribute((const void *)fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)need));
    hipLaunchKernelGGL(fn, dim3(grid), dim3(NW * 64), need, st, a);
    return launch_status("adc_scan_qfilter64_kernel");
}

template <int M, int NQ, int NW, int WPS, bool SKEWED, bool DBUF>
static int launch_filter(const ScanArgs &a, int grid, hipStream_t st) {
    const size_t lds_lut = (size_t)a.Ks * NQ * M * 16;
    size_t need = lds_lut + 128;
    const size_t scratch = (size_t)4 * NQ * NW * 64 * 8;
    if (need < scratch) need = scratch;
    auto fn = adc_scan_filter_kernel<M, NQ, NW, WPS, SKEWED, DBUF>;
    ANNLITE_HIP_TRY(hipFuncSetAttribute((const void *)fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)need));
    hipLaunchKernelGGL(fn, dim3(grid), dim3(NW * 64), need, st, a);
    return launch_status("adc_scan_filter_kernel");
}

extern "C" int annlite_scan_plan_query(int64_t N, int64_t M, int64_t Ks, int code_bytes, int64_t B, int64_t k,
                                       annlite_scan_plan *plan) {
    ANNLITE_REQUIRE(plan != nullptr, "plan is NULL");
    ANNLITE_REQUIRE(N >= 0 && M >= 1 && Ks >= 1 && B >= 0, "bad shape N=%lld M=%lld Ks=%lld B=%lld", (long long)N,
                    (long long)M, (long long)Ks, (long long)B);
    ANNLITE_REQUIRE(code_bytes == 1 || code_bytes == 2 || code_bytes == 4, "code_bytes must be 1, 2 or 4");
    ANNLITE_REQUIRE(k >= 1 && k <= 64, "k=%lld outside [1,64] (larger k: annlite_adc_dist + host-side selection)",
                    (long long)k);
    ANNLITE_REQUIRE(N < (1ll << 32) - 1, "N must be < 2^32-1 rows per call (shard the table)");
    FastCfg c;
    const int n_cu = device_cu_count();
    memset(plan, 0, sizeof(*plan));
    plan->max_k = 64;
    if (fast_cfg(M, Ks, code_bytes, k, &c)) {
        plan->fast = 1;
        plan->qi = c.QI;
        plan->qt = c.qt();
        plan->waves = c.NW;
        const int n_tiles = (int)((B + plan->qt - 1) / plan->qt);
        int ns;
        int64_t sr;
        plan_slices(N > 0 ? N : 1, n_tiles > 0 ? n_tiles : 1, c.NW, n_cu, true, &ns, &sr);
        plan->n_slices = ns;
        plan->lut_floats = ((B + 15) / 16) * 16 * M * Ks;  // padded to 16 queries
        // [partial keys][Smax f32 x Bpad][qstep f32 x Bpad][qlo f64 x Bpad][lo,hi f32 x Bpad*M][q16 u16 x Bpad*M*Ks]
        const int64_t bpad = ((B + 15) / 16) * 16;
        plan->workspace_bytes = (int64_t)n_tiles * plan->qt * ns * k * 8 + 256 + bpad * 4 + 256;
        if (c.mode == 4)
            plan->workspace_bytes += bpad * 4 + 256 + 2 * (bpad * 8 + 256) + (bpad * ns * 8 + 256) + (n_tiles * 4 + 256) +
                                     bpad * M * Ks * 2 + 256;
    } else {
        plan->fast = 0;
        plan->qi = 1;
        plan->qt = 1;
        plan->waves = 4;
        int ns;
        int64_t sr;
        plan_slices(N > 0 ? N : 1, B > 0 ? (int)B : 1, 4, n_cu, false, &ns, &sr);
        plan->n_slices = ns;
        plan->lut_floats = B * M * Ks;
        plan->workspace_bytes = B * ns * k * 8;
    }
    if (plan->workspace_bytes < 8) plan->workspace_bytes = 8;
    return ANNLITE_OK;
}

template <int M, int QI, int NQ, int NW, int WPS, bool SKEWED, int MODE>
static int launch_fast(const ScanArgs &a, int grid, hipStream_t st) {
    const size_t lds = (size_t)a.Ks * NQ * M * QI * 4;
    size_t need = lds;
    const size_t scratch = (size_t)QI * NQ * NW * 64 * 8;
    if (need < scratch) need = scratch;
    auto fn = adc_scan_fast_kernel<M, QI, NQ, NW, WPS, SKEWED, MODE>;
    ANNLITE_HIP_TRY(hipFuncSetAttribute((const void *)fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)need));
    hipLaunchKernelGGL(fn, dim3(grid), dim3(NW * 64), need, st, a);
    return launch_status("adc_scan_fast_kernel");
}

// ---- optional in-library timing of the dominant kernel (bench.py roofline leg) --------------------
static unsigned long long *g_dbg = nullptr;  // debug only (ANNLITE_DEBUG_COUNTERS): leaked 64-byte device buffer
static thread_local int g_prof_on = 0;
static thread_local hipEvent_t g_ev0 = nullptr, g_ev1 = nullptr;
static thread_local int g_ev_valid = 0;

static void prof_begin(hipStream_t st) {
    if (!g_prof_on) return;
    if (!g_ev0) {
        (void)hipEventCreate(&g_ev0);
        (void)hipEventCreate(&g_ev1);
    }
    (void)hipEventRecord(g_ev0, st);
}
static void prof_end(hipStream_t st) {
    if (!g_prof_on) return;
    (void)hipEventRecord(g_ev1, st);
    g_ev_valid = 1;
}

// run the scan kernels: fills workspace with the per-(query, slice) sorted key lists [B][NS][k]
// annlite_pq_search_topk: the L2 tables are built by the quantisation launch itself (lut_dev is then written)
struct LutBuild {
    const float *queries;
    const float *codebooks;
    int64_t D;
};
// where a scan that can merge its slices itself puts the final result (merged is set when it did)
struct ScanOut {
    float *d;
    int64_t *i;
    int64_t *packed;
    int64_t row_base;
    int sqrt_out;
    bool merged;
};

static int scan_partial(const void *codes_dev, int code_bytes, int codes_layout, int64_t N, int64_t M, int64_t Ks,
                        const uint32_t *valid_bits_dev, const float *lut_dev, int64_t B, int64_t k,
                        void *workspace_dev, size_t workspace_bytes, hipStream_t st, annlite_scan_plan *plan_out,
                        bool share_across_slices, const LutBuild *build = nullptr, ScanOut *outp = nullptr) {
    annlite_scan_plan plan;
    int rc = annlite_scan_plan_query(N, M, Ks, code_bytes, B, k, &plan);
    if (rc != ANNLITE_OK) return rc;
    *plan_out = plan;
    if (B == 0) return ANNLITE_OK;
    ANNLITE_REQUIRE(codes_layout == ANNLITE_CODES_PLAIN || (codes_layout == ANNLITE_CODES_SKEWED && plan.fast),
                    "codes_layout %d not supported by this plan (SKEWED needs the fast plan)", codes_layout);
    ANNLITE_REQUIRE(lut_dev && workspace_dev, "null device pointer");
    ANNLITE_REQUIRE(N == 0 || codes_dev, "codes_dev is NULL");
    if (workspace_bytes < (size_t)plan.workspace_bytes) {
        set_error("workspace %zu B < required %lld B", workspace_bytes, (long long)plan.workspace_bytes);
        return ANNLITE_ERR_WORKSPACE;
    }
    const int n_cu = device_cu_count();
    ScanArgs a = {};
    a.codes = codes_dev;
    a.valid = valid_bits_dev;
    a.lut = lut_dev;
    a.partial = (unsigned long long *)workspace_dev;
    a.N = N;
    a.Ks = (int)Ks;
    a.B = (int)B;
    a.k = (int)k;
    a.n_tiles = (int)((B + plan.qt - 1) / plan.qt);
    a.n_slices = plan.n_slices;
    a.smax = nullptr;
    a.q16 = nullptr;
    a.qstep = nullptr;
    a.qlo = nullptr;
    a.gkey = nullptr;
    a.dbg = nullptr;
    a.dbg_skip = getenv("ANNLITE_DEBUG_SKIP") ? atoi(getenv("ANNLITE_DEBUG_SKIP")) : 0;
    if (getenv("ANNLITE_DEBUG_COUNTERS")) {
        if (!g_dbg) ANNLITE_HIP_TRY(hipMalloc((void **)&g_dbg, 64));
        ANNLITE_HIP_TRY(hipMemsetAsync(g_dbg, 0, 64, st));
        a.dbg = g_dbg;
    }
    {
        int ns;
        int64_t sr;
        plan_slices(N > 0 ? N : 1, a.n_tiles, plan.waves, n_cu, plan.fast != 0, &ns, &sr);
        a.slice_rows = sr;
    }
    // slots of slices that hold no rows stay "none"
    size_t fill_bytes = 0;
    bool fused_fill = false;
    {
        // partial lists, and (quantised-filter plan) the shared bounds right behind them; the rest is scratch
        size_t fill = (size_t)plan.workspace_bytes;
        FastCfg c0;
        if (plan.fast && fast_cfg(M, Ks, code_bytes, k, &c0) && c0.mode == 4) {
            const size_t bpad = (size_t)((B + 15) / 16) * 16;
            auto r256 = [](size_t x) { return (x + 255) / 256 * 256; };
            fill = r256((size_t)a.n_tiles * plan.qt * plan.n_slices * k * 8) + r256(bpad * 8) +
                   r256(bpad * plan.n_slices * 8) + r256((size_t)a.n_tiles * 4);
        }
        fill_bytes = fill;
        FastCfg c1;
        fused_fill = N > 0 && plan.fast && fast_cfg(M, Ks, code_bytes, k, &c1) && c1.mode == 4 && M != 64;  // lut_quantise_fused_kernel
        if (!fused_fill) ANNLITE_HIP_TRY(hipMemsetAsync(workspace_dev, 0xff, fill, st));
    }
    if (N == 0) return ANNLITE_OK;
    const int n_items = (plan.fast && a.n_slices < 8) ? ((a.n_tiles + 8 / a.n_slices - 1) / (8 / a.n_slices)) * 8
                                                      : a.n_tiles * a.n_slices;
    a.n_items = n_items;
    if (plan.fast) {
        FastCfg c;
        fast_cfg(M, Ks, code_bytes, k, &c);
        int grid = n_items < n_cu * c.wg_per_cu ? n_items : n_cu * c.wg_per_cu;
        const bool sk = codes_layout == ANNLITE_CODES_SKEWED;
        if (c.mode == 4) {
            // quantise the fp32 tables: min/max -> (step, L, Smax) -> 12-bit codes, all inside the workspace
            const int64_t bpad = ((B + 15) / 16) * 16;
            char *wp = (char *)workspace_dev + (((int64_t)a.n_tiles * plan.qt * plan.n_slices * k * 8 + 255) / 256) * 256;
            auto carve = [&](int64_t bytes) { char *r = wp; wp += ((bytes + 255) / 256) * 256; return r; };
            // [gkey][gk2][tile_done] directly behind the partial lists: the one fill covers exactly these four
            unsigned long long *gk = (unsigned long long *)carve(bpad * 8);
            unsigned long long *gk2 = (unsigned long long *)carve(bpad * plan.n_slices * 8);
            unsigned int *tile_done = (unsigned int *)carve((int64_t)a.n_tiles * 4);
            if (share_across_slices && outp && (outp->packed || (outp->d && outp->i))) {
                a.tile_done = tile_done;
                a.out_d = outp->d;
                a.out_i = outp->i;
                a.out_packed = outp->packed;
                a.row_base = outp->row_base;
                a.sqrt_out = outp->sqrt_out;
                outp->merged = true;
            }
            float *smax = (float *)carve(bpad * 4);
            float *qstep = (float *)carve(bpad * 4);
            double *qlo = (double *)carve(bpad * 8);
            // per-slice lists stay complete (a superset generator for re-rank) unless sharing is requested
            a.gkey = share_across_slices ? gk : nullptr;
            a.gk2 = share_across_slices ? gk2 : nullptr;
            {
                const int64_t grp = plan.n_slices < 8 ? plan.n_slices : 8;  // slices scanned concurrently
                a.jm1 = (int)((k + grp - 1) / grp) - 1;
                a.flush_mask = 63;
                if (const char *e = getenv("ANNLITE_FLUSH_MASK")) a.flush_mask = atoi(e);
            }
            // (the q16 table sits behind the small arrays; carve order is irrelevant to the kernels)
            uint16_t *q16 = (uint16_t *)carve(bpad * M * Ks * 2);
            const int qmax = (int)(32767 / M);
            const unsigned n_g8 = (unsigned)(bpad / 8);
            {
                float *lut_rw = const_cast<float *>(lut_dev);
                u32x4 *fillp = (u32x4 *)workspace_dev;
                const int64_t fillv = (int64_t)(fill_bytes / 16);
#define ANNLITE_QUANT(MM)                                                                                            \
    if (build)                                                                                                       \
        hipLaunchKernelGGL((lut_l2_build_quantise_kernel<MM>), dim3(n_g8), dim3(1024), (size_t)(8 * build->D * 4), st,  \
                           build->queries, (int)B, (int)build->D, build->codebooks, (int)Ks, lut_rw, qmax, q16, qstep,   \
                           qlo, smax, fillp, fillv);                                                                 \
    else                                                                                                             \
        hipLaunchKernelGGL((lut_quantise_fused_kernel<MM>), dim3(n_g8), dim3(256), 0, st, lut_rw, (int)Ks, qmax, q16,  \
                           qstep, qlo, smax, fillp, fillv)
                if (M == 64)
                    hipLaunchKernelGGL(lut_quantise64_kernel, dim3((unsigned)(bpad / 4)), dim3(1024), 0, st, lut_dev, (int)Ks,
                                       qmax, q16, qstep, qlo, smax);
                else if (M == 8) { ANNLITE_QUANT(8); } else if (M == 16) { ANNLITE_QUANT(16); } else { ANNLITE_QUANT(32); }
#undef ANNLITE_QUANT
            }
            rc = launch_status("lut_quantise_fused_kernel");
            if (rc != ANNLITE_OK) return rc;
            if (share_across_slices && N >= 4096) {
                int64_t S = 8192;
                if (const char *e = getenv("ANNLITE_SEED_ROWS")) S = atoll(e);
                if (S > N) S = N;
                const bool skw = codes_layout == ANNLITE_CODES_SKEWED;
#define ANNLITE_SEED(MM, QPB_)                                                                                    \
    {                                                                                                             \
        auto fn = skw ? seed_bound_kernel<MM, true, QPB_> : seed_bound_kernel<MM, false, QPB_>;                   \
        const size_t lds = (size_t)Ks * (MM + 1) * 4 * QPB_ + (size_t)2 * kSeedWaves * 64 * 8; /* cand + minima */ \
        ANNLITE_HIP_TRY(hipFuncSetAttribute((const void *)fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); \
        hipLaunchKernelGGL(fn, dim3((unsigned)(((B + 3) / 4) * (4 / QPB_))), dim3(kSeedWaves * 64), lds, st,      \
                           (const uint8_t *)codes_dev, S, valid_bits_dev, lut_dev, (int)B, (int)Ks, (int)k, gk);   \
    }
                if (M == 8) ANNLITE_SEED(8, 4) else if (M == 16) ANNLITE_SEED(16, 4) else if (M == 32) ANNLITE_SEED(32, 4)
                else ANNLITE_SEED(64, 2)
#undef ANNLITE_SEED
                rc = launch_status("seed_bound_kernel");
                if (rc != ANNLITE_OK) return rc;
            }
            a.smax = smax;
            a.qstep = qstep;
            a.qlo = qlo;
            a.q16 = q16;
        }
        if (c.mode == 3) {
            // rounding slack of the fast filter sum needs Smax[b] = sum_m max_k |lut[b][m][k]|
            const int64_t part_bytes = (int64_t)a.n_tiles * plan.qt * plan.n_slices * k * 8;
            float *smax = (float *)((char *)workspace_dev + ((part_bytes + 255) / 256) * 256);
            const int n_groups = (int)(((B + 15) / 16) * 16 / c.QI);
            hipLaunchKernelGGL(lut_smax_kernel, dim3((n_groups + 3) / 4), dim3(256), 0, st, lut_dev, n_groups, (int)M,
                               (int)Ks, c.QI, smax);
            rc = launch_status("lut_smax_kernel");
            if (rc != ANNLITE_OK) return rc;
            a.smax = smax;
        }
        prof_begin(st);
#define ANNLITE_LAUNCH_F(MM, NQ_, NW_, WPS_, DB_) \
    (sk ? launch_filter<MM, NQ_, NW_, WPS_, true, DB_>(a, grid, st) : launch_filter<MM, NQ_, NW_, WPS_, false, DB_>(a, grid, st))
#define ANNLITE_LAUNCH_Q(MM, NQ_, NW_, WPS_) \
    (sk ? launch_qfilter<MM, NQ_, NW_, WPS_, true>(a, grid, st) : launch_qfilter<MM, NQ_, NW_, WPS_, false>(a, grid, st))
#define ANNLITE_LAUNCH_M(MM, QI_, NQ_, NW_, WPS_, MODE_) \
    (sk ? launch_fast<MM, QI_, NQ_, NW_, WPS_, true, MODE_>(a, grid, st) : launch_fast<MM, QI_, NQ_, NW_, WPS_, false, MODE_>(a, grid, st))
#define ANNLITE_LAUNCH(MM, QI_, NQ_, NW_, WPS_) ANNLITE_LAUNCH_M(MM, QI_, NQ_, NW_, WPS_, 0)
        if (c.id == 640 && sk) {
            set_error("the two-pass M=64 kernel reads PLAIN tables only (SKEWED M=64 tables are wrap-coded for the default kernel)");
            return ANNLITE_ERR_UNSUPPORTED;
        }
        switch (c.id) {
            case 830: rc = ANNLITE_LAUNCH_Q(8, 2, 16, 4); break;
            case 3230: rc = ANNLITE_LAUNCH_Q(32, 1, 12, 3); break;
            case 6430: rc = sk ? launch_qfilter64<16, true>(a, grid, st) : launch_qfilter64<16, false>(a, grid, st); break;
            case 6431: rc = sk ? launch_qfilter64<12, true>(a, grid, st) : launch_qfilter64<12, false>(a, grid, st); break;
            case 6432: rc = sk ? launch_qfilter64<8, true>(a, grid, st) : launch_qfilter64<8, false>(a, grid, st); break;
            case 1630: rc = ANNLITE_LAUNCH_Q(16, 2, 12, 3); break;
            case 1631: rc = ANNLITE_LAUNCH_Q(16, 2, 16, 4); break;
            case 1632: rc = ANNLITE_LAUNCH_Q(16, 2, 8, 2); break;
            case 81: rc = ANNLITE_LAUNCH_F(8, 2, 8, 2, true); break;
            case 1600: rc = ANNLITE_LAUNCH_F(16, 2, 8, 2, true); break;
            case 1601: rc = ANNLITE_LAUNCH_F(16, 2, 12, 3, false); break;
            case 1602: rc = ANNLITE_LAUNCH_F(16, 1, 8, 4, true); break;
            case 1603: rc = ANNLITE_LAUNCH_F(16, 2, 16, 4, false); break;
            case 321: rc = ANNLITE_LAUNCH_F(32, 1, 8, 2, true); break;
            case 80: rc = ANNLITE_LAUNCH(8, 4, 2, 8, 2); break;
            case 160: rc = ANNLITE_LAUNCH(16, 4, 2, 8, 2); break;
            case 161: rc = ANNLITE_LAUNCH(16, 4, 1, 8, 4); break;
            case 162: rc = ANNLITE_LAUNCH(16, 4, 2, 16, 4); break;
            case 163: rc = ANNLITE_LAUNCH(16, 4, 2, 12, 3); break;
            case 164: rc = ANNLITE_LAUNCH_M(16, 4, 2, 8, 2, 1); break;
            case 165: rc = ANNLITE_LAUNCH_M(16, 4, 2, 12, 3, 1); break;
            case 166: rc = ANNLITE_LAUNCH_M(16, 4, 1, 8, 4, 1); break;
            case 167: rc = ANNLITE_LAUNCH_M(16, 4, 2, 8, 2, 2); break;
            case 320: rc = ANNLITE_LAUNCH(32, 4, 1, 8, 2); break;
            default: rc = ANNLITE_LAUNCH(64, 2, 1, 8, 2); break;
        }
#undef ANNLITE_LAUNCH
#undef ANNLITE_LAUNCH_M
#undef ANNLITE_LAUNCH_F
#undef ANNLITE_LAUNCH_Q
        prof_end(st);
        return rc;
    }
    const int grid = n_items < n_cu * 8 ? n_items : n_cu * 8;
    const size_t lds = 4 * 64 * 8;
    prof_begin(st);
    if (code_bytes == 1)
        hipLaunchKernelGGL((adc_scan_generic_kernel<uint8_t, 4>), dim3(grid), dim3(256), lds, st, a, (int)M);
    else if (code_bytes == 2)
        hipLaunchKernelGGL((adc_scan_generic_kernel<uint16_t, 4>), dim3(grid), dim3(256), lds, st, a, (int)M);
    else
        hipLaunchKernelGGL((adc_scan_generic_kernel<uint32_t, 4>), dim3(grid), dim3(256), lds, st, a, (int)M);
    prof_end(st);
    return launch_status("adc_scan_generic_kernel");
}

extern "C" int annlite_debug_counters(uint64_t *out8) {
    ANNLITE_REQUIRE(out8 != nullptr, "out8 is NULL");
    if (!g_dbg) {
        set_error("no counters recorded (set ANNLITE_DEBUG_COUNTERS=1 before the scan)");
        return ANNLITE_ERR_INVALID;
    }
    ANNLITE_HIP_TRY(hipDeviceSynchronize());
    ANNLITE_HIP_TRY(hipMemcpy(out8, g_dbg, 64, hipMemcpyDeviceToHost));
    return ANNLITE_OK;
}

extern "C" int annlite_profile_enable(int on) {
    g_prof_on = on ? 1 : 0;
    g_ev_valid = 0;
    return ANNLITE_OK;
}

extern "C" int annlite_profile_last_scan_ms(float *ms) {
    ANNLITE_REQUIRE(ms != nullptr, "ms is NULL");
    if (!g_ev_valid) {
        set_error("no scan has been recorded (call annlite_profile_enable(1) first)");
        return ANNLITE_ERR_INVALID;
    }
    ANNLITE_HIP_TRY(hipEventSynchronize(g_ev1));
    ANNLITE_HIP_TRY(hipEventElapsedTime(ms, g_ev0, g_ev1));
    return ANNLITE_OK;
}

static int scan_topk_impl(const void *codes_dev, int code_bytes, int codes_layout, int64_t N, int64_t M, int64_t Ks,
                          const uint32_t *valid_bits_dev, const float *lut_dev, int64_t B, int64_t k, int64_t row_base,
                          float *out_dist_dev, int64_t *out_id_dev, int64_t *out_packed_dev, void *workspace_dev,
                          size_t workspace_bytes, void *stream, const LutBuild *build = nullptr, int flags = 0) {
    annlite_scan_plan plan;
    hipStream_t st = (hipStream_t)stream;
    ANNLITE_REQUIRE(B == 0 || out_packed_dev || (out_dist_dev && out_id_dev), "null output pointer");
    const int sqrt_out = (flags & ANNLITE_FLAG_SQRT) && !out_packed_dev ? 1 : 0;
    ScanOut so = {out_dist_dev, out_id_dev, out_packed_dev, row_base, sqrt_out, false};
    if (getenv("ANNLITE_NO_INKERNEL_MERGE")) so.d = nullptr, so.i = nullptr, so.packed = nullptr;
    int rc = scan_partial(codes_dev, code_bytes, codes_layout, N, M, Ks, valid_bits_dev, lut_dev, B, k, workspace_dev,
                          workspace_bytes, st, &plan, true, build, &so);
    if (rc != ANNLITE_OK || B == 0 || so.merged) return rc;
    hipLaunchKernelGGL(merge_partial_kernel, dim3((unsigned)((B + 3) / 4)), dim3(256), 0, st,
                       (const unsigned long long *)workspace_dev, (int)B, plan.n_slices, (int)k, row_base,
                       out_dist_dev, out_id_dev, out_packed_dev, sqrt_out);
    return launch_status("merge_partial_kernel");
}

extern "C" int annlite_adc_scan_topk(const void *codes_dev, int code_bytes, int codes_layout, int64_t N, int64_t M,
                                     int64_t Ks, const uint32_t *valid_bits_dev, const float *lut_dev, int64_t B,
                                     int64_t k, int64_t row_base, float *out_dist_dev, int64_t *out_id_dev,
                                     void *workspace_dev, size_t workspace_bytes, void *stream) {
    return scan_topk_impl(codes_dev, code_bytes, codes_layout, N, M, Ks, valid_bits_dev, lut_dev, B, k, row_base,
                          out_dist_dev, out_id_dev, nullptr, workspace_dev, workspace_bytes, stream);
}

extern "C" int annlite_adc_scan_topk_packed(const void *codes_dev, int code_bytes, int codes_layout, int64_t N, int64_t M,
                                            int64_t Ks, const uint32_t *valid_bits_dev, const float *lut_dev, int64_t B,
                                            int64_t k, int64_t row_base, int64_t *out_packed_dev, void *workspace_dev,
                                            size_t workspace_bytes, void *stream) {
    ANNLITE_REQUIRE(B == 0 || out_packed_dev, "null output pointer");
    return scan_topk_impl(codes_dev, code_bytes, codes_layout, N, M, Ks, valid_bits_dev, lut_dev, B, k, row_base, nullptr,
                          nullptr, out_packed_dev, workspace_dev, workspace_bytes, stream);
}

static size_t r256z(size_t x) { return (x + 255) / 256 * 256; }

extern "C" int annlite_pq_search_workspace_bytes(int64_t N, int64_t M, int64_t Ks, int code_bytes, int64_t B, int64_t k,
                                                 int64_t *bytes) {
    ANNLITE_REQUIRE(bytes != nullptr, "bytes is NULL");
    annlite_scan_plan plan;
    int rc = annlite_scan_plan_query(N, M, Ks, code_bytes, B, k, &plan);
    if (rc != ANNLITE_OK) return rc;
    *bytes = (int64_t)(r256z((size_t)plan.workspace_bytes) + r256z((size_t)plan.lut_floats * 4));
    return ANNLITE_OK;
}

extern "C" int annlite_pq_search_topk(int lut_kind, const float *queries_dev, int64_t B, int64_t D,
                                      const float *codebooks_dev, const void *codes_dev, int code_bytes, int codes_layout,
                                      int64_t N, int64_t M, int64_t Ks, const uint32_t *valid_bits_dev, int64_t k,
                                      int64_t row_base, float *out_dist_dev, int64_t *out_id_dev, int64_t *out_packed_dev,
                                      int flags, void *workspace_dev, size_t workspace_bytes, void *stream) {
    ANNLITE_REQUIRE(M >= 1 && D >= M && D % M == 0,
                    "input dimension must be dividable by number of sub-space (D=%lld, M=%lld)", (long long)D, (long long)M);
    annlite_scan_plan plan;
    int rc = annlite_scan_plan_query(N, M, Ks, code_bytes, B, k, &plan);
    if (rc != ANNLITE_OK) return rc;
    if (B == 0) return ANNLITE_OK;
    const size_t scan_ws = r256z((size_t)plan.workspace_bytes);
    const size_t need = scan_ws + r256z((size_t)plan.lut_floats * 4);
    if (workspace_bytes < need) {
        set_error("workspace %zu B < required %zu B (annlite_pq_search_workspace_bytes)", workspace_bytes, need);
        return ANNLITE_ERR_WORKSPACE;
    }
    ANNLITE_REQUIRE(queries_dev && codebooks_dev && workspace_dev, "null device pointer");
    float *lut = (float *)((char *)workspace_dev + scan_ws);
    FastCfg c;
    const bool fuse = N > 0 && plan.fast && fast_cfg(M, Ks, code_bytes, k, &c) && c.mode == 4 && M != 64 &&
                      lut_kind == ANNLITE_LUT_L2 && ((D / M) % 4) == 0 && !getenv("ANNLITE_NO_FUSED_LUT");
    if (!fuse) {
        rc = annlite_lut_build(lut_kind, queries_dev, B, D, codebooks_dev, M, Ks, lut,
                               plan.fast ? ANNLITE_LAYOUT_TILED : ANNLITE_LAYOUT_BMK, plan.qi, stream);
        if (rc != ANNLITE_OK) return rc;
        return scan_topk_impl(codes_dev, code_bytes, codes_layout, N, M, Ks, valid_bits_dev, lut, B, k, row_base,
                              out_dist_dev, out_id_dev, out_packed_dev, workspace_dev, scan_ws, stream, nullptr, flags);
    }
    const LutBuild lb = {queries_dev, codebooks_dev, D};
    return scan_topk_impl(codes_dev, code_bytes, codes_layout, N, M, Ks, valid_bits_dev, lut, B, k, row_base, out_dist_dev,
                          out_id_dev, out_packed_dev, workspace_dev, scan_ws, stream, &lb, flags);
}

extern "C" int annlite_adc_scan_candidates(const void *codes_dev, int code_bytes, int codes_layout, int64_t N,
                                           int64_t M, int64_t Ks, const uint32_t *valid_bits_dev,
                                           const float *lut_dev, int64_t B, int64_t k, int64_t row_base,
                                           float *out_dist_dev, int64_t *out_id_dev, void *workspace_dev,
                                           size_t workspace_bytes, void *stream) {
    annlite_scan_plan plan;
    hipStream_t st = (hipStream_t)stream;
    int rc = scan_partial(codes_dev, code_bytes, codes_layout, N, M, Ks, valid_bits_dev, lut_dev, B, k, workspace_dev,
                          workspace_bytes, st, &plan, false);
    if (rc != ANNLITE_OK || B == 0) return rc;
    ANNLITE_REQUIRE(out_dist_dev && out_id_dev, "null output pointer");
    const int64_t total = B * plan.n_slices * k;
    hipLaunchKernelGGL(export_partial_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st,
                       (const unsigned long long *)workspace_dev, total, row_base, out_dist_dev, out_id_dev);
    return launch_status("export_partial_kernel");
}

extern "C" int annlite_codes_skew(const void *in_dev, int64_t N, int64_t M, const int64_t *ids_dev, int64_t id_base,
                                  void *out_dev, int inverse, void *stream) {
    ANNLITE_REQUIRE(N >= 0 && M >= 1 && id_base >= 0, "bad shape");
    if (N == 0) return ANNLITE_OK;
    ANNLITE_REQUIRE(in_dev && out_dev, "null device pointer");
    const int64_t total = N * M;
    hipLaunchKernelGGL(codes_skew_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       (const uint8_t *)in_dev, N, (int)M, ids_dev, id_base, (uint8_t *)out_dev, inverse);
    return launch_status("codes_skew_kernel");
}

extern "C" int annlite_topk_merge(const float *dist_dev, const int64_t *id_dev, int64_t G, int64_t B, int64_t k,
                                  float *out_dist_dev, int64_t *out_id_dev, void *stream) {
    ANNLITE_REQUIRE(G >= 1 && B >= 0 && k >= 1 && k <= 64, "bad G=%lld B=%lld k=%lld (k<=64)", (long long)G,
                    (long long)B, (long long)k);
    if (B == 0) return ANNLITE_OK;
    ANNLITE_REQUIRE(dist_dev && id_dev && out_dist_dev && out_id_dev, "null device pointer");
    hipLaunchKernelGGL(merge_lists_kernel, dim3((unsigned)((B + 3) / 4)), dim3(256), 0, (hipStream_t)stream, dist_dev,
                       id_dev, (const int64_t *)nullptr, (int)G, (int)B, (int)k, out_dist_dev, out_id_dev, 0);
    return launch_status("merge_lists_kernel");
}

extern "C" int annlite_topk_merge_packed(const int64_t *packed_dev, int64_t G, int64_t B, int64_t k, float *out_dist_dev,
                                         int64_t *out_id_dev, int flags, void *stream) {
    ANNLITE_REQUIRE(G >= 1 && B >= 0 && k >= 1 && k <= 64, "bad G=%lld B=%lld k=%lld (k<=64)", (long long)G,
                    (long long)B, (long long)k);
    if (B == 0) return ANNLITE_OK;
    ANNLITE_REQUIRE(packed_dev && out_dist_dev && out_id_dev, "null device pointer");
    hipLaunchKernelGGL(merge_lists_kernel, dim3((unsigned)((B + 3) / 4)), dim3(256), 0, (hipStream_t)stream,
                       (const float *)nullptr, (const int64_t *)nullptr, packed_dev, (int)G, (int)B, (int)k, out_dist_dev,
                       out_id_dev, (flags & ANNLITE_FLAG_SQRT) ? 1 : 0);
    return launch_status("merge_lists_kernel");
}

extern "C" int annlite_topk_rows(const float *values_dev, int64_t B, int64_t N, int64_t k, int64_t id_base,
                                 float *out_dist_dev, int64_t *out_id_dev, void *stream) {
    ANNLITE_REQUIRE(B >= 0 && N >= 0 && k >= 1 && k <= 64, "bad B=%lld N=%lld k=%lld (k<=64)", (long long)B,
                    (long long)N, (long long)k);
    ANNLITE_REQUIRE(N < (1ll << 32) - 1, "N must be < 2^32-1");
    if (B == 0) return ANNLITE_OK;
    ANNLITE_REQUIRE(out_dist_dev && out_id_dev && (N == 0 || values_dev), "null device pointer");
    hipLaunchKernelGGL(topk_rows_kernel, dim3((unsigned)((B + 3) / 4)), dim3(256), 0, (hipStream_t)stream, values_dev,
                       (int)B, N, (int)k, id_base, out_dist_dev, out_id_dev);
    return launch_status("topk_rows_kernel");
}

extern "C" int annlite_adc_dist(const float *adtable_dev, int64_t M, int64_t Ks, const void *codes_dev, int code_bytes,
                                int64_t N, float *out_dev, void *stream) {
    ANNLITE_REQUIRE(M >= 1 && Ks >= 1 && N >= 0, "bad shape");
    ANNLITE_REQUIRE(code_bytes == 1 || code_bytes == 2 || code_bytes == 4, "code_bytes must be 1, 2 or 4");
    if (N == 0) return ANNLITE_OK;
    ANNLITE_REQUIRE(adtable_dev && codes_dev && out_dev, "null device pointer");
    const size_t tab = (size_t)M * Ks * 4;
    const size_t lds = tab <= 64 * 1024 ? tab : 0;
    int64_t blocks = (N + 255) / 256;
    const int64_t cap = (int64_t)device_cu_count() * 8;
    if (blocks > cap) blocks = cap;
    hipStream_t st = (hipStream_t)stream;
    if (code_bytes == 1)
        hipLaunchKernelGGL(adc_dist_kernel<uint8_t>, dim3((unsigned)blocks), dim3(256), lds, st, adtable_dev, (int)M,
                           (int)Ks, (const uint8_t *)codes_dev, N, out_dev);
    else if (code_bytes == 2)
        hipLaunchKernelGGL(adc_dist_kernel<uint16_t>, dim3((unsigned)blocks), dim3(256), lds, st, adtable_dev, (int)M,
                           (int)Ks, (const uint16_t *)codes_dev, N, out_dev);
    else
        hipLaunchKernelGGL(adc_dist_kernel<uint32_t>, dim3((unsigned)blocks), dim3(256), lds, st, adtable_dev, (int)M,
                           (int)Ks, (const uint32_t *)codes_dev, N, out_dev);
    return launch_status("adc_dist_kernel");
}

extern "C" int annlite_adc_gather(const float *lut_bmk_dev, int64_t B, int64_t M, int64_t Ks, const void *codes_dev,
                                  int code_bytes, int64_t N, const int64_t *cand_dev, int64_t R, float *out_dev,
                                  void *stream) {
    ANNLITE_REQUIRE(M >= 1 && Ks >= 1 && N >= 0 && B >= 0 && R >= 0, "bad shape");
    ANNLITE_REQUIRE(code_bytes == 1 || code_bytes == 2 || code_bytes == 4, "code_bytes must be 1, 2 or 4");
    if (B * R == 0) return ANNLITE_OK;
    ANNLITE_REQUIRE(lut_bmk_dev && cand_dev && out_dev && (N == 0 || codes_dev), "null device pointer");
    const int64_t total = B * R;
    const unsigned blocks = (unsigned)((total + 255) / 256);
    hipStream_t st = (hipStream_t)stream;
    if (code_bytes == 1)
        hipLaunchKernelGGL(adc_gather_kernel<uint8_t>, dim3(blocks), dim3(256), 0, st, lut_bmk_dev, (int)B, (int)M,
                           (int)Ks, (const uint8_t *)codes_dev, N, cand_dev, (int)R, out_dev);
    else if (code_bytes == 2)
        hipLaunchKernelGGL(adc_gather_kernel<uint16_t>, dim3(blocks), dim3(256), 0, st, lut_bmk_dev, (int)B, (int)M,
                           (int)Ks, (const uint16_t *)codes_dev, N, cand_dev, (int)R, out_dev);
    else
        hipLaunchKernelGGL(adc_gather_kernel<uint32_t>, dim3(blocks), dim3(256), 0, st, lut_bmk_dev, (int)B, (int)M,
                           (int)Ks, (const uint32_t *)codes_dev, N, cand_dev, (int)R, out_dev);
    return launch_status("adc_gather_kernel");
}
