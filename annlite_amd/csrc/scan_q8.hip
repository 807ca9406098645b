// scan_q8.hip -- the ADC scan with BYTE filter tables: 16 queries per 16-byte LDS entry, 32 queries per workgroup
// (DESIGN.md section 3.1 has the measurements behind every choice below).
//
// Same discipline as adc_scan_qfilter_kernel (scan_qfilter.hip): a cheap integer LOWER bound of every (query, row)
// distance from tables in LDS, the exact ascending-m fp32 sum (the reference's arithmetic, pq_bindings.pyx:30-47 ==
// space_pq.h:32-35) only for the rows whose bound beats the current k-th distance, shared top-k lists, bounds shared
// across the row slices -- bit-exact results.  What changes is the table:
//   * entries are BYTES, Q = min(QMAX, floor((lut - lo[q][m]) / step[q])), QMAX = 15 (M * QMAX <= 240: a byte sum never
//     carries): one ds_read_b128 serves 16 look-ups per lane and one v_add_u32 adds four of them, so the step loop has
//     the SAME instruction stream as the u16 kernel (32 ds_read_b128 + ~146 VALU per 64 rows) for twice the queries;
//   * 4-bit entries are enough because the quantisation follows the THRESHOLD, not the table's range: what decides the
//     number of rows passing the filter is the resolution relative to R = thr - L (L = sum_m lo): step = R / (T - 1) with
//     T = q8_target (88 for M = 16 with 16-key lists, else 96) when the table is built, entries above QMAX steps are clipped -- such a row is far outside;
//   * the threshold tightens over a scan, so the workgroup builds its table ITSELF from the fp32 TILED table (L2) with the
//     bound it starts from (seed kernel / other slices) and rebuilds it at an epoch end when the bounds of a quarter of its
//     queries have halved their T (epochs end after steps q8_epoch0 = 255, 4095, ... -- 15, 255, ... where the scan starts
//     without a seed bound from rows spread over the table: a barrier of all waves costs more than a finer table saves,
//     the sparse schedule is the guard against a first bound that is far off);
//   * filter: ((0x80 | T) - (S & 0x7f)) & ~S keeps bit 7 of a byte iff S <= T (T <= 127: no borrow crosses a byte).
// Bound: Q <= (v - lo) / step * (1 + 2^-22) (fp32 subtract, multiply by 1/step, round down), hence
//     d_real - L >= S * step * (1 - 2^-22);  a row can be in the top-k only if d_fp32 <= thr, i.e.
//     d_real <= thr + slack32  =>  S <= T := floor((thr + slack32 - L) / step * (1 + 2^-19)) + 1   (double).
// Clipping (min with QMAX) and a T clamped to 127 keep the bound valid for ANY step.
// The waves DRAW their 64-row blocks from a counter in LDS (the SIMD's arbiter favours its oldest wave and one wave alone
// issues at a third of the rate four reach together: a static deal left the last waves to run an epoch out alone).
// Candidates are handled by a CONSUMER wave: NW - 1 waves scan and only push (S, query, row) -- the M = 16 kernel under shared
// bounds: just the ROW, the consumer finds the queries (q8_row_pass_mask) -- into a ring in LDS; the last
// wave of the workgroup pops them in batches of up to 128, drops those whose S no longer passes, computes the exact sums,
// updates the lists -- it is their only writer: no locks -- and publishes the bounds (to the other workgroups one batch
// later, behind the next batch's gathers); it imports the sibling slices' bound: the k-th smallest of their j smallest
// keys.  The scanning waves never wait for global memory, the exact path has its own registers instead of being called
// with ~100 live VGPRs saved to scratch (the u16 kernel's out-of-line flush), and inlining it into the step loop made
// the compiler spill the loop-invariant LDS base registers into the hot path.
// LDS: [table Ks * 512][Q8Lds: bounds, ring control, block counter, slot parameters, lists u64 x32x16, ring u64 x1024,
//      insertion queues]
// Shapes (template <M, NW, SKEWED, NQ, CB = bytes per code>; launch_q8_scan ids):
//   1650  M = 16, uint8 codes, NQ = 2 entry groups = 32 queries per workgroup -- everything above; the headline kernel;
//   1651  the same kernel over CELL TILES (template flag TL; annlite_ivf_search_topk): a work item = up to 32 (query, cell) pairs
//         of ONE cell scanning that cell's rows; slot -> query through a map in LDS, the image gathered from per-query byte tables
//         (q8_gather_table), bounds shared by query, tiles drawn from a device counter; the step loop is 1650's;
//   6450  M = 64 ("WIDE"): 8 queries per 8-BYTE entry (ds_read_b64), byte sums of 16 look-ups widened into u16 sums (T up to
//         960), two half tables [Ks + 1][32][8 B] with wrap-coded SKEWED rows and one v_perm_b32 per address;
//   850   M = 8, uint16 codes, Ks <= 512, NQ = 2: table [Ks][2][8][16 B], one v_perm_b32 per address, the two entry groups
//         read in lane-dependent order (conflict-free), PLAIN rows rotated in registers;
//   851   M = 8, uint16 codes, Ks <= 1024, NQ = 1: 16 queries per workgroup (2-way bank conflicts, inherent).
//   852   M = 8, uint8 codes (Ks <= 256), NQ = 2: the 850 shape with a 64 KB table; SKEWED rows need no rotation.
//   3250  M = 32, uint8 codes, NQ = 1: 16 queries per workgroup (32 sub-spaces x 16 B x 256 codes = 128 KB), entries clipped at
//         7 (32 x 7 = 224: a byte sum never carries), two half tables [256][16][16 B] 64 KB apart, one v_perm_b32 per address
//         ((half << 16) | (code << 8) | column), candidates as (S, slot, row) entries like the M = 8 shapes.
#include "scan_lists.h"

#ifndef ANNLITE_Q8_EXP
#define ANNLITE_Q8_EXP 0  // (timing experiments, results wrong: 1 = look-ups without the adds, 2 = adds without the look-ups,
                          // 3 = code rows computed instead of loaded; 5 = one entry per hit row, no enumeration)
#endif

namespace annlite {

#ifndef ANNLITE_Q8_WDEPTH
#define ANNLITE_Q8_WDEPTH 16  // M = 64: landing registers (8-byte entries) of the look-up ring
#endif
#ifndef ANNLITE_Q8_STAGE_DEPTH
#define ANNLITE_Q8_STAGE_DEPTH 8  // look-ups in flight per lane and row in the consumer's row-queue stage (out of line: own registers; 16: no change)
#endif
#ifndef ANNLITE_Q8_THW_MASK
#define ANNLITE_Q8_THW_MASK 1  // the scanning waves pick up the workgroup's bounds every (mask + 1)-th step
#endif
#ifndef ANNLITE_Q8_ROWQ
#define ANNLITE_Q8_ROWQ 1  // M = 16: the scanning waves push ROWS (any query passes), the consumer finds the queries (see q8_rows_to_entries)
#endif
#ifndef ANNLITE_Q8_BUILD_UNROLL
#define ANNLITE_Q8_BUILD_UNROLL 2  // codes per thread whose table loads are in flight together in the (re)build (4 / 8: the build no
                                   // faster -- 15 -> 16 us -- and the allocator then spilled into the step loop: +12 %)
#endif
#ifndef ANNLITE_Q8_DEPTH
#define ANNLITE_Q8_DEPTH 8  // look-ups in flight per lane
#endif

// Q8Cfg: entries are clipped at QMAX (M * QMAX <= 240: a byte sum never carries); a slot without a bound yet
// ("open": nothing seeded it) clips at QOPEN, M * QOPEN <= 112, so that T = 127 passes every row.
// LDS accesses by BYTE ADDRESS in the LDS address space.  (Through generic pointers -- struct members, lambda captures --
// the compiler lost the address space and emitted FLAT instructions for the ring, the lists and the bounds: a flat access
// counts on the vector-memory counter as well, so every one of them made its wave wait for ALL its outstanding global
// loads -- the scanning waves' prefetched code rows, the consumer's table gathers.)
#define ANNLITE_LDS __attribute__((address_space(3)))
template <class T>
__device__ __forceinline__ T ldsv(uint32_t ad) {  // volatile load (another wave may have written it)
    return *(volatile ANNLITE_LDS T *)(uintptr_t)ad;
}
template <class T>
__device__ __forceinline__ void ldsv_st(uint32_t ad, T v) {
    *(volatile ANNLITE_LDS T *)(uintptr_t)ad = v;
}
__device__ __forceinline__ uint32_t lds_add_u32(uint32_t ad, uint32_t v) {
    return __hip_atomic_fetch_add((ANNLITE_LDS uint32_t *)(uintptr_t)ad, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
__device__ __forceinline__ uint32_t lds_base_addr() {
    return (uint32_t)(uintptr_t)(ANNLITE_LDS unsigned char *)g_smem;
}

// (Q8Cfg, q8_qt, q8_table_bytes, q8_bound_from_key, q8_slot_params: scan_common.h -- the preparation launch computes the same
// parameters when it prebuilds the byte tables)
// The workgroup's byte table from the fp32 TILED tables of its 32 queries: thread (m, h) = (tid % M, (tid / M) % 2)
// keeps the minima and 1/step of its 16 queries in registers and walks the codes.  RNE(t - 0.5) <= floor(t): the
// conversion's rounding mode does not matter for the bound.
// What the out-of-line parts of the kernel (table (re)build, end of a work item) read of the kernel's arguments comes
// straight from the KERNARG SEGMENT (constant memory, scalar loads at the point of use).  As fields of the by-value kernel
// parameter they are loaded at the kernel's entry and stay live in SGPRs across the step loop -- which sits at the register
// limit: a dozen SGPRs more spill into VGPR lanes and from there into scratch reloads inside the loop.  (The address of
// the parameter itself would make the compiler keep a copy of the whole block in scratch.)
typedef const ScanArgs __attribute__((address_space(4))) *q8_kernarg_ptr;
// (taken in the KERNEL and handed down: inside an out-of-line function the builtin returned a null pointer)
__device__ __forceinline__ q8_kernarg_ptr q8_kernarg() { return (q8_kernarg_ptr)__builtin_amdgcn_kernarg_segment_ptr(); }
struct Q8Build {  // what a (re)build needs, gathered from the kernarg segment inside q8_rebuild
    const float *lut, *qlom, *qstep, *smax;
    const double *qlo;
    const unsigned long long *gkey;  // first bounds: the shared array, or this slice's row of the per-slice seeds, or NULL
    int32_t Ks, B, k, target;
    const unsigned long long *gseed0;  // prebuilt first tables (ScanArgs::btab): the seed keys they were quantised for
    const uint8_t *btab;
};

// First table of a work item where the preparation launch has prebuilt it (ScanArgs::btab): the tile's LDS image, 16-byte
// loads from L2, instead of converting the tile's fp32 tables (4x the bytes, and the conversion).  Entry groups of 4 queries
// beyond the batch (a ragged last tile: nobody wrote their dword) are zeroed -- the all-zero table of a pad slot.
template <int M, int NW, int NQ>
__device__ __forceinline__ void q8_copy_table(const Q8Build &a, int tile, uint32_t tab_ad, int tid) {
    constexpr int NT = NW * 64;
    static_assert(M == 16 && NQ == 2, "the prebuilt image is the M = 16 kernel's (q8_entry16)");
    // the whole image, whatever Ks is (rows >= Ks of a half are never addressed): 16-byte entry i sits in slot i % 16 of its row, and
    // slot = 2 j + group + half (the odd half is shifted by one slot; its 257th row holds one entry, in slot 0)
    constexpr int n16 = kQ8Image16 / 16;
    const int n_g4 = ((a.B + 15) / 16) * 4;
    const u32x4 *src = (const u32x4 *)(a.btab + (int64_t)tile * (int64_t)kQ8Image16);
#pragma unroll 4
    for (int i = tid; i < n16; i += NT) {
        u32x4 v = src[i];
        const int half = i >= 4096 ? 1 : 0;
        const int g4 = tile * (4 * NQ) + (((i & 15) - half) & 1) * 4;
        if (g4 + 3 >= n_g4) {
            if (g4 + 0 >= n_g4) v.x = 0u;
            if (g4 + 1 >= n_g4) v.y = 0u;
            if (g4 + 2 >= n_g4) v.z = 0u;
            v.w = 0u;
        }
        *(ANNLITE_LDS u32x4 *)(uintptr_t)(tab_ad + 16u * (uint32_t)i) = v;
    }
}

template <int M, int NW, int NQ>
__device__ __forceinline__ void q8_build_table(const Q8Build &a, int tile, uint32_t tab_ad, uint32_t inv_ad, uint32_t clip_ad,
                                               int tid) {
    constexpr int RB = M * 16, NT = NW * 64, KHS = NT / M;
    static_assert(NT % M == 0 && KHS % NQ == 0, "build mapping");
    const int m = tid % M, kh0 = tid / M, h = kh0 % NQ;
    const int n_g4 = ((a.B + 15) / 16) * 4;  // fp32 TILED groups that exist (the table is padded to 16 queries)
    float lo_r[16], inv_r[16], clip_r[16];
    const float *src[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int g4 = tile * (4 * NQ) + h * 4 + i;
        const bool ok = g4 < n_g4;
        src[i] = a.lut + ((int64_t)(ok ? g4 : 0) * a.Ks * M + m) * 4;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            lo_r[4 * i + e] = ok ? a.qlom[(int64_t)(g4 * 4 + e) * M + m] : 0.f;
            inv_r[4 * i + e] = ok ? ldsv<float>(inv_ad + 4u * (uint32_t)(h * 16 + 4 * i + e)) : 0.f;
            clip_r[4 * i + e] = ldsv<float>(clip_ad + 4u * (uint32_t)(h * 16 + 4 * i + e));
        }
    }
#pragma unroll ANNLITE_Q8_BUILD_UNROLL
    for (int kh = kh0; kh < a.Ks * NQ; kh += KHS) {
        const int k = kh / NQ;
        uint32_t w[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const f32x4 v = *(const f32x4 *)(src[i] + (int64_t)k * M * 4);
            uint32_t pk = 0;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float t = __builtin_fmaf(v[e] - lo_r[4 * i + e], inv_r[4 * i + e], -0.5f);
                t = __builtin_fminf(t, clip_r[4 * i + e]);
                pk = __builtin_amdgcn_cvt_pk_u8_f32(t, e, pk);  // saturates below 0
            }
            w[i] = pk;
        }
        uint32_t ad;
        if constexpr (Q8Cfg<M>::M32)  // two half tables of 16 sub-spaces, 64 KB apart: (half << 16) | (code << 8) | column
            ad = tab_ad + ((uint32_t)(m >> 4) << 16) + ((uint32_t)k << 8) + (uint32_t)(m & 15) * 16u;
        else if constexpr (M == 16 && NQ == 2) ad = tab_ad + q8_entry16((uint32_t)k, (uint32_t)m, (uint32_t)h);  // two half tables by sub-space parity
        else ad = tab_ad + (uint32_t)((k * NQ + h) * RB + m * 16);
        *(ANNLITE_LDS u32x4 *)(uintptr_t)ad = (u32x4){w[0], w[1], w[2], w[3]};
    }
}

// Cell tiles (TL; annlite_ivf_search_topk): the 32 slots of a tile hold ARBITRARY queries (ScanArgs::vmap), so the tile's image cannot be
// prebuilt per tile -- the preparation launch leaves every query's byte table on its own, bq [B][Ks][16] (4 KB per query, quantised
// for the query's seed key like a prebuilt image), and the workgroup assembles its image from the 32 tables of its slots: entry
// (code, m, group g) = byte (code, m) of the 16 queries of group g.  Lane = (w, g, u, c2): it loads dword w (sub-spaces 4 w .. 4 w + 3)
// of code kk = 2 c2 + u + 8 (wave + 16 pass) from the 16 tables of group g (a wave reads 128 contiguous bytes of each of two tables
// per load), transposes 4 x 4 bytes at a time (v_perm_b32) into the four entries (m = 4 w + t) and stores them with ds_write_b128 in
// the order t ^ 2 u: the 16 lanes of a store then cover the 16 slots of one 256-byte row of a half twice over two codes -- 16 distinct
// bank groups (two lanes with the same code differ in (m >> 1, g), the two codes in m >> 1 parity).  Pad slots (query -1) get zeros.
template <int NW>
__device__ __forceinline__ void q8_gather_table(const uint8_t *bq, int Ks, uint32_t tab_ad, uint32_t qmap_ad, int tid) {
    const int lane = tid & 63, wave = tid >> 6;
    const uint32_t w = (uint32_t)lane & 3u, g = ((uint32_t)lane >> 2) & 1u, u = ((uint32_t)lane >> 3) & 1u, c2 = (uint32_t)lane >> 4;
    int32_t qb[16];  // byte offset of the slot's table, -1: pad
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        const int b = ldsv<int32_t>(qmap_ad + 4u * (g * 16u + (uint32_t)j));
        qb[j] = b < 0 ? -1 : b * (Ks * 16);
    }
    for (int kk = (int)(2u * c2 + u) + 8 * wave; kk < Ks; kk += 8 * NW) {
        uint32_t r[16];
#pragma unroll
        for (int j = 0; j < 16; ++j)
            r[j] = qb[j] < 0 ? 0u : *(const uint32_t *)(bq + (int64_t)qb[j] + (int64_t)(kk * 16) + (int64_t)(4u * w));
        u32x4 e[4];  // e[t]: the 16 queries' bytes of sub-space 4 w + t
#pragma unroll
        for (int v = 0; v < 4; ++v) {
            const uint32_t a = __builtin_amdgcn_perm(r[4 * v + 1], r[4 * v + 0], 0x05010400u);   // [r0.b0, r1.b0, r0.b1, r1.b1]
            const uint32_t a2 = __builtin_amdgcn_perm(r[4 * v + 1], r[4 * v + 0], 0x07030602u);  // [r0.b2, r1.b2, r0.b3, r1.b3]
            const uint32_t c = __builtin_amdgcn_perm(r[4 * v + 3], r[4 * v + 2], 0x05010400u);
            const uint32_t cc2 = __builtin_amdgcn_perm(r[4 * v + 3], r[4 * v + 2], 0x07030602u);
            e[0][v] = __builtin_amdgcn_perm(c, a, 0x05040100u);
            e[1][v] = __builtin_amdgcn_perm(c, a, 0x07060302u);
            e[2][v] = __builtin_amdgcn_perm(cc2, a2, 0x05040100u);
            e[3][v] = __builtin_amdgcn_perm(cc2, a2, 0x07060302u);
        }
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const uint32_t tt = (uint32_t)t ^ (2u * u);
            const u32x4 val = tt == 0u ? e[0] : tt == 1u ? e[1] : tt == 2u ? e[2] : e[3];
            *(ANNLITE_LDS u32x4 *)(uintptr_t)(tab_ad + q8_entry16((uint32_t)kk, 4u * w + tt, g)) = val;
        }
    }
}

// M = 64: the workgroup's byte table of its 8 queries (two fp32 TILED groups of 4) in the layout of the M = 64 u16 kernel
// -- two half tables of 32 sub-spaces, [Ks + 1][32 columns][8 B], the second HALF_B = (Ks + 1) * 256 bytes behind the first,
// row Ks a copy of row 0 (the wrap-coded SKEWED rows address one row down where a lane's column wraps: wrap64_mask) -- with
// byte entries.  Thread (kr, m) = (tid / 64, tid % 64) walks the codes kr, kr + NT / 64, ...
template <int NW>
__device__ __forceinline__ void q8_build_table_wide(const Q8Build &a, int tile, uint32_t tab_ad, uint32_t inv_ad, uint32_t clip_ad,
                                                    int tid) {
    constexpr int M = 64, NT = NW * 64, KPT = NT / M;
    const int m = tid % M, kr = tid / M;
    const int n_g4 = ((a.B + 15) / 16) * 4;
    const uint32_t half_b = (uint32_t)(a.Ks + 1) * 256u;
    float lo_r[8], inv_r[8], clip_r[8];
    const float *src[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int g4 = tile * 2 + i;
        const bool ok = g4 < n_g4;
        src[i] = a.lut + ((int64_t)(ok ? g4 : 0) * a.Ks * M + m) * 4;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            lo_r[4 * i + e] = ok ? a.qlom[(int64_t)(g4 * 4 + e) * M + m] : 0.f;
            inv_r[4 * i + e] = ok ? ldsv<float>(inv_ad + 4u * (uint32_t)(4 * i + e)) : 0.f;
            clip_r[4 * i + e] = ldsv<float>(clip_ad + 4u * (uint32_t)(4 * i + e));
        }
    }
    const uint32_t col = tab_ad + (uint32_t)(m >> 5) * half_b + (uint32_t)(m & 31) * 8u;
#pragma unroll 2
    for (int k = kr; k < a.Ks; k += KPT) {
        uint32_t w[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const f32x4 v = *(const f32x4 *)(src[i] + (int64_t)k * M * 4);
            uint32_t pk = 0;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float t = __builtin_fmaf(v[e] - lo_r[4 * i + e], inv_r[4 * i + e], -0.5f);
                t = __builtin_fminf(t, clip_r[4 * i + e]);
                pk = __builtin_amdgcn_cvt_pk_u8_f32(t, e, pk);  // saturates below 0
            }
            w[i] = pk;
        }
        *(ANNLITE_LDS u32x2 *)(uintptr_t)(col + (uint32_t)k * 256u) = (u32x2){w[0], w[1]};
        if (k == 0) *(ANNLITE_LDS u32x2 *)(uintptr_t)(col + (uint32_t)a.Ks * 256u) = (u32x2){w[0], w[1]};  // row Ks = row 0
    }
}

constexpr int kRingSize = 1024;   // candidate ring entries (u64 each): one private ring of kWaveRing entries per scanning wave
// bytes of a ring entry: the M = 16 kernel with 64-key lists always runs the row queue (bare 4-byte row ids -- what lets the 160 KB
// hold its 128 KB table, 16 KB of lists and the row queue's parking area); every other shape pushes (S << 40 | slot << 32 | row)
template <int M, int LK>
constexpr int q8_ring_entry_bytes() { return (LK == 64 && M == 16) ? 4 : 8; }
constexpr int kWaveRing = 64;     // (a wave-step pushes at most 64 entries at a time)
constexpr int kPopPerRing = 8;    // entries the consumer takes from one wave's ring per batch (8 x 15 rings <= 128 = two per lane)

// LDS map of a workgroup (absolute LDS byte addresses): [table Ks * 512][what follows]
//   shq      u8 [32]   current filter bound (0x80 | T) of every slot -- what the scanning waves load as thp
//   ring_ctl u32: +0 arrived (scanning waves that finished an epoch, cumulative), +4 the block counter the scanning waves draw
//            from, +16 tails [16] (entries wave w has pushed, mod 2^16: written by wave w only), +80 heads [16] (entries of wave
//            w's ring consumed: written by the consumer only)
//   gkl      u64 [32]  best bound known for the slot (own k-th key or imported)
//   step / inv / clip f32 [32], tb u8 [32] (T the table was built for), ctl u32 (+0 rebuild flag, +4 merge flag / arrival mask,
//            +8 the early merger's "patience is over" verdict)
//   tau      u64 [32]  the k-th key the consumer last published; c0, c1 f64 [32]: T = floor(thr * c1 + c0) + 1 (q8_bound)
//   list     u64 [32][16] the 16 smallest keys of every slot, ascending; gjl u64 [32] the j-th key last published
//   ring     u64 [16][kWaveRing]; qkey u64 [4][128], qslot u8 [4][128] insertion queues; chg u8 [32]; stamps u64 [4] (debug)
//   seen     u32 x 2 (guard statistics); rowq [64 lanes][48 B]: the row queue's parked masks / row ids / code bytes
//   qmap     i32 [32]  cell tiles (TL): the real query whose tables slot q scans with (ScanArgs::vmap), -1 = padding slot
struct Q8Lds {
    uint32_t tab, shq, ring_ctl, gkl, step, inv, tb, ctl, clip, tau, c0, c1, list, gjl, ring, qkey, qslot, chg, stamps, seen, rowq, qmap;
    // lk: keys per slot list -- 16 (k <= 16, four insertions at a time), or 64 (16 < k <= 64: one 64-lane list per slot, 16 KB;
    // its ring entries are the row queue's bare 4-byte row ids, which is what makes the 160 KB hold it)
    // re: bytes of a ring entry (q8_ring_entry_bytes: 4 = the row queue's bare row ids, 8 = (S, slot, row))
    __device__ __forceinline__ explicit Q8Lds(int lut_bytes, int lk = 16, int re = 8) {
        tab = lds_base_addr();
        shq = tab + (uint32_t)lut_bytes;
        ring_ctl = shq + 32;
        gkl = shq + 192;
        step = shq + 448;
        inv = shq + 576;
        tb = shq + 704;
        ctl = shq + 736;
        clip = shq + 768;
        tau = shq + 896;
        c0 = shq + 1152;
        c1 = shq + 1408;
        list = shq + 1664;
        gjl = list + 32u * (uint32_t)lk * 8u;
        ring = gjl + 32 * 8;
        qkey = ring + kRingSize * (uint32_t)re;
        qslot = qkey + 4 * 128 * 8;
        chg = qslot + 4 * 128;
        stamps = chg + 32;
        seen = stamps + 32;  // u32: candidates this workgroup has seen (guard statistics), u32: abort flag of later items
        rowq = seen + 16;    // [64 lanes][48 B]: the consumer's (pass masks, row ids) u32x4 + the two popped rows' 16 code bytes (row queue)
        qmap = rowq + 3072;  // i32 [32] (cell tiles only, TL: allocated behind the row queue's area): the query of every slot, -1 = padding
    }
    __device__ __forceinline__ uint32_t arrived() const { return ring_ctl; }
    __device__ __forceinline__ uint32_t blk_ctr() const { return ring_ctl + 4; }
    __device__ __forceinline__ uint32_t tails() const { return ring_ctl + 16; }
    __device__ __forceinline__ uint32_t heads() const { return ring_ctl + 80; }
};

// What the consumer wave keeps per slot (LDS): the 16 smallest keys (ordered distance << 32 | row) seen so far,
// ascending -- the kernel serves k <= 16 -- and tau, the k-th key it last published.  Insertions run FOUR AT A TIME, one
// per 16-lane row of the wave: slot q belongs to row q & 3; in a round every row takes the next pending candidate of
// one of its slots, its 16 lanes load the slot's list, count the entries in front of the candidate (ballot), shift the
// tail by one lane (DPP row_shr) and store.  The bounds of the slots that changed are published once per batch, one
// lane per slot.  (The u16 kernels' offer_to_list -- one slot at a time, 64-lane lists shifted through ds_bpermute,
// a double-precision division per insertion on one lane -- cost ~0.7 us per inserted candidate: 585 us per workgroup
// at 1.25M rows against a 290 us scan.  Unsorted bags compacted when full: ~1.5 us per compaction and a bound that
// lags 22 insertions behind -- 2x the candidates.)

// filter bound (TFLAG | T) of a slot from a k-th key: T = floor((thr + slack32 - L) / step * (1 + 2^-19)) + 1 with the
// slot's constants folded into c1 = (1 + 2^-19) / step, c0 = (slack32 - L) * c1 (set when the table is built)
template <int M>
__device__ __forceinline__ uint32_t q8_bound(unsigned long long key, double c0, double c1) {
    const uint32_t hi = (uint32_t)(key >> 32);
    if (hi == kKeyInfHi) return 2u * Q8Cfg<M>::TFLAG - 1u;
    double qd = __builtin_floor(__builtin_fma((double)ordered_to_f32(hi), c1, c0)) + 1.0;
    if (!(qd < (double)Q8Cfg<M>::TMAX)) qd = (double)Q8Cfg<M>::TMAX;  // (a NaN lands HERE: everything passes)
    else if (!(qd > 0.0)) qd = 0.0;
    return Q8Cfg<M>::TFLAG | (uint32_t)qd;
}
// where the scanning waves load a slot's bound from: a byte per slot; WIDE: half-words, slots 1 and 2 (5 and 6) swapped -- the
// order the widened sums come out in ((q0, q2), (q1, q3), (q4, q6), (q5, q7) per dword)
template <int M>
__device__ __forceinline__ uint32_t q8_bound_ad(const Q8Lds &o, int q) {
    if constexpr (Q8Cfg<M>::WIDE) return o.shq + 2u * (uint32_t)((q & 4) | ((q & 1) << 1) | ((q >> 1) & 1));
    else return o.shq + (uint32_t)q;
}
template <int M>
__device__ __forceinline__ uint32_t q8_ld_bound(const Q8Lds &o, int q) {
    if constexpr (Q8Cfg<M>::WIDE) return ldsv<unsigned short>(q8_bound_ad<M>(o, q));
    else return ldsv<unsigned char>(q8_bound_ad<M>(o, q));
}
template <int M>
__device__ __forceinline__ void q8_st_bound(const Q8Lds &o, int q, uint32_t v) {
    if constexpr (Q8Cfg<M>::WIDE) ldsv_st<unsigned short>(q8_bound_ad<M>(o, q), (unsigned short)v);
    else ldsv_st<unsigned char>(q8_bound_ad<M>(o, q), (unsigned char)v);
}
template <int M>
__device__ __forceinline__ uint32_t q8_ld_built(const Q8Lds &o, int q) {  // T the slot's table was built for
    if constexpr (Q8Cfg<M>::WIDE) return ldsv<unsigned short>(o.tb + 2u * (uint32_t)q);
    else return ldsv<unsigned char>(o.tb + (uint32_t)q);
}

__device__ __forceinline__ uint32_t dpp_row_shr1(uint32_t x) {
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x111, 0xf, 0xf, false);  // lane i <- lane i - 1 of its row
}

// consumer wave: pop up to 128 entries (two per lane).  An entry is (S << 40 | slot << 32 | row): entries whose integer
// sum no longer passes the slot's CURRENT bound are dropped before any global load (the bound tightens while a backlog
// waits); the others get their exact sum (the reference's arithmetic) and go into their slot's list.
// The global publication of a batch's bounds (the other row slices' workgroups import them) is DEFERRED to the next batch,
// behind the issue of its table gathers: the device-scope atomics take microseconds and the wave's memory counter is in
// order -- issued right away they sat in front of the next batch's gathers.
template <int QT, int LK = 16, bool TL = false>
__device__ __forceinline__ void q8_publish_global(const FlushCtx &c, const Q8Lds &o, int lane, unsigned long long &pend_o,
                                                  unsigned long long &pend_j) {
    if (lane < QT) {  // (slots beyond QT never change; their cells do not exist)
        int b = c.b0 + lane;
        if constexpr (TL) {  // cell tiles: the bound belongs to the slot's QUERY -- its other cells' tiles import it (gk2 is NULL)
            b = ldsv<int32_t>(o.qmap + 4u * (uint32_t)lane);
            if (b < 0) pend_o = ~0ull, b = 0;
        }
        if (pend_o != ~0ull && c.gkey) __hip_atomic_fetch_min(c.gkey + b, pend_o, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (pend_j != ~0ull && c.gk2) {
            unsigned long long *cell = c.gk2 + ((int64_t)b * c.n_slices + c.slice) * kGk2Keys;
            if constexpr (LK == 64) {
                // 16 < k <= 64: the keys at four list positions p_0 < p_1 < p_2 < p_3 (FlushCtx::pos) as they are NOW: cell i says
                // "this slice holds pos_i + 1 rows at or below this key" -- true of any later version too (keys only fall), so a
                // reader may mix versions (import_bounds: the weighted count)
                const uint32_t lst = o.list + (uint32_t)lane * (64u * 8u);
#pragma unroll
                for (int i = 0; i < kGk2Keys; ++i)
                    __hip_atomic_store(cell + i, ldsv<unsigned long long>(lst + 8u * (uint32_t)((c.pos >> (8 * i)) & 0xff)), __ATOMIC_RELAXED,
                                       __HIP_MEMORY_SCOPE_AGENT);
            } else
            if (c.jm1 < 2) {
                // this slice's j smallest keys as they are NOW (keys only fall; a reader may see a mix of two versions:
                // each key belongs to a row of this slice, and it drops a duplicate)
                const uint32_t lst = o.list + (uint32_t)lane * 128u;
                __hip_atomic_store(cell, ldsv<unsigned long long>(lst), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (c.jm1 == 1) __hip_atomic_store(cell + 1, ldsv<unsigned long long>(lst + 8), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            } else {
                __hip_atomic_store(cell, pend_j, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // the j-th key alone
            }
        }
    }
    pend_o = ~0ull;
    pend_j = ~0ull;
}

template <int M, bool SKEWED, int QT, int CB, bool ROWS_IN_LDS = false, int LK = 16, bool TL = false>
__device__ __forceinline__ void q8_consume(const FlushCtx &c, const Q8Lds &o, const unsigned long long (&e)[2], bool (&act)[2],
                                           int lane, uint32_t &n_kept, uint32_t &n_offered, unsigned long long &pend_o,
                                           unsigned long long &pend_j) {
    constexpr int CW = M * CB / 4;  // dwords of a code row (CB = bytes per code: 1, or 2 for uint16 codes)
    constexpr bool C16 = CB == 2;
    static_assert(CB == 1 || (CB == 2 && !SKEWED), "uint16 codes: PLAIN rows");
    int q[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        q[u] = (int)(e[u] >> 32) & 31;
        const uint32_t sv = (uint32_t)(e[u] >> 40) & 0xffffu;
        const uint32_t tq = q8_ld_bound<M>(o, q[u]);
        act[u] = act[u] && ((Q8Cfg<M>::TFLAG | sv) <= tq);  // still passes?  (pad slots, TMAX, never push)
    }
    n_kept += (uint32_t)(__popcll(__ballot(act[0])) + __popcll(__ballot(act[1])));
    // exact ascending-m fp32 sums of both rows (exact_row_sum's arithmetic; the loads of the two rows interleaved)
    float ex[2] = {0.f, 0.f};
    if constexpr (Q8Cfg<M>::WIDE || Q8Cfg<M>::M32) {
        // M = 64 / 32: one row at a time, its M table entries in rounds of 16 gathers (the sum stays the ascending-m chain)
        if (!(c.skip & 1)) {
#pragma unroll 1
            for (int u = 0; u < 2; ++u) {
                if (!__ballot(act[u])) continue;
                const uint32_t rid = act[u] ? (uint32_t)e[u] : 0u;
                uint32_t cp[CW];
                const uint32_t *p = (const uint32_t *)(c.codes + (int64_t)rid * M);
#pragma unroll
                for (int i = 0; i < CW; ++i) cp[i] = p[i];
                if constexpr (SKEWED && Q8Cfg<M>::WIDE) skew64_decode(cp, (int)(rid % 32));  // two skewed halves, wrap-coded
                else if constexpr (SKEWED) {  // stored byte j of row n is the code of sub-space (j + n) mod M: rotate back
                    const int sinv = (M - (int)(rid % M)) % M;
                    bool abit_inv[8];
#pragma unroll
                    for (int i = 0; i < 8; ++i) abit_inv[i] = (((sinv >> 2) >> i) & 1) != 0;
                    rotate_row<CW>(cp, abit_inv, (uint32_t)(sinv & 3));
                }
                const int b = c.b0 + (act[u] ? q[u] : 0);
                const float *lq = c.lut + ((int64_t)(b >> 2) * c.Ks) * (M * 4) + (b & 3);
                float sum = 0.f;
                static_for<0, M / 16>([&](auto C) {
                    constexpr int m0 = decltype(C)::value * 16;
                    float vals[16];
#pragma unroll
                    for (int j = 0; j < 16; ++j) {
                        const int m = m0 + j;
                        const uint32_t code = (cp[m / 4] >> (8 * (m % 4))) & 0xffu;
                        vals[j] = lq[((int64_t)code * M + m) * 4];
                    }
                    if constexpr (m0 == 0) {
                        if (u == 0) q8_publish_global<QT, LK>(c, o, lane, pend_o, pend_j);  // (the previous batch's, behind these gathers)
                    }
#pragma unroll
                    for (int j = 0; j < 16; ++j) sum += vals[j];
                });
                ex[u] = sum;
            }
        }
    } else
    if (!(c.skip & 1)) {
        uint32_t cp[2][CW];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            if constexpr (ROWS_IN_LDS) {  // (row queue: the stage that found the queries parked the rows' stored bytes -- lane, u)
                const u32x4 v = ldsv<u32x4>(o.rowq + 48u * (uint32_t)lane + 16u + 16u * (uint32_t)u);
                cp[u][0] = v.x, cp[u][1] = v.y, cp[u][2] = v.z, cp[u][3] = v.w;
            } else {
                const uint32_t *p = (const uint32_t *)(c.codes + (int64_t)(act[u] ? (uint32_t)e[u] : 0u) * (CW * 4));
#pragma unroll
                for (int i = 0; i < CW; ++i) cp[u][i] = p[i];
            }
        }
        float vals[2][M];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const uint32_t rid = act[u] ? (uint32_t)e[u] : 0u;
            if constexpr (SKEWED) {
                const int sinv = (M - (int)(rid % M)) % M;
                bool abit_inv[8];
#pragma unroll
                for (int i = 0; i < 8; ++i) abit_inv[i] = (((sinv >> 2) >> i) & 1) != 0;
                rotate_row<CW>(cp[u], abit_inv, (uint32_t)(sinv & 3));
            }
            int b = c.b0 + (act[u] ? q[u] : 0);
            if constexpr (TL) {  // cell tiles: the slot's query (a pad slot never passes: its b is not used)
                b = ldsv<int32_t>(o.qmap + 4u * (uint32_t)(act[u] ? q[u] : 0));
                b = b < 0 ? 0 : b;
            }
            // (TL: the preparation launch of the pruned search leaves the tables per QUERY, [b][Ks][M] -- see seed_bound_kernel<CELLS>)
            const float *lq = TL ? c.lut + (int64_t)b * c.Ks * M : c.lut + ((int64_t)(b >> 2) * c.Ks) * (M * 4) + (b & 3);
#pragma unroll
            for (int m = 0; m < M; ++m) {
                const uint32_t code = C16 ? (cp[u][m / 2] >> (16 * (m % 2))) & 0xffffu : (cp[u][m / 4] >> (8 * (m % 4))) & 0xffu;
                vals[u][m] = TL ? lq[(int64_t)code * M + m] : lq[((int64_t)code * M + m) * 4];
            }
        }
        q8_publish_global<QT, LK, TL>(c, o, lane, pend_o, pend_j);  // (the previous batch's, behind this batch's gathers)
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int m = 0; m < M; ++m) ex[u] += vals[u][m];
    }
    if (c.skip & 2) return;
    const int row = lane >> 4, li = lane & 15;
    // the candidates that beat their slot's current bound (own k-th key or the imported one) go to the queue of their
    // slot's row in LDS; then round t inserts entry t of every queue: no cross-lane traffic inside the loop
    int cnt[4] = {0, 0, 0, 0};
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        const unsigned long long key = ((unsigned long long)f32_to_key(ex[u]) << 32) | (uint32_t)e[u];  // (NaN behind +inf)
        bool pend = false;
        if (act[u]) {
            unsigned long long kth = ldsv<unsigned long long>(o.list + 8u * (uint32_t)(q[u] * LK + c.km1));
            const unsigned long long gk = ldsv<unsigned long long>(o.gkl + 8u * (uint32_t)q[u]);
            if (gk < kth) kth = gk;
            pend = key < kth;
        }
        if (pend) ldsv_st<unsigned char>(o.chg + (uint32_t)q[u], 1);
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) {
            const unsigned long long bm = __ballot(pend && (q[u] & 3) == rr);
            if (pend && (q[u] & 3) == rr) {
                const int idx = cnt[rr] + (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(bm >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)bm, 0u));
                ldsv_st<unsigned long long>(o.qkey + 8u * (uint32_t)(rr * 128 + idx), key);
                ldsv_st<unsigned char>(o.qslot + (uint32_t)(rr * 128 + idx), (unsigned char)q[u]);
            }
            cnt[rr] += __popcll(bm);
        }
    }
    n_offered += (uint32_t)(cnt[0] + cnt[1] + cnt[2] + cnt[3]);
    const int mycnt = row == 0 ? cnt[0] : row == 1 ? cnt[1] : row == 2 ? cnt[2] : cnt[3];
    int rounds = cnt[0] > cnt[1] ? cnt[0] : cnt[1];
    rounds = rounds > cnt[2] ? rounds : cnt[2];
    rounds = rounds > cnt[3] ? rounds : cnt[3];
    if constexpr (LK == 64) {
        // 64-key lists: lane = list position, ONE insertion per wave operation -- the keys in front of the candidate are a prefix
        // (the list is ascending), every lane behind it hands its key one position up THROUGH the LDS (no cross-lane traffic:
        // the wave's LDS operations execute in order, all reads of the old list precede the writes)
#pragma unroll 1
        for (int rr = 0; rr < 4; ++rr) {
            const int n_rr = rr == 0 ? cnt[0] : rr == 1 ? cnt[1] : rr == 2 ? cnt[2] : cnt[3];
#pragma unroll 1
            for (int t = 0; t < n_rr; ++t) {
                const unsigned long long ckey = ldsv<unsigned long long>(o.qkey + 8u * (uint32_t)(rr * 128 + t));
                const int cq = ldsv<unsigned char>(o.qslot + (uint32_t)(rr * 128 + t)) & 31;
                const uint32_t ent_ad = o.list + 8u * (uint32_t)(cq * 64 + lane);
                const unsigned long long ent = ldsv<unsigned long long>(ent_ad);
                const int pos = __popcll(__ballot(ent < ckey));
                if (pos <= c.km1) {  // (an earlier insertion of this batch may have pushed the k-th key below the candidate)
                    if (lane >= pos && lane < 63) ldsv_st<unsigned long long>(ent_ad + 8u, ent);
                    if (lane == pos) ldsv_st<unsigned long long>(ent_ad, ckey);
                }
            }
        }
    } else {
#pragma unroll 1
    for (int t = 0; t < rounds; ++t) {
        const bool valid = t < mycnt;
        const unsigned long long ckey = ldsv<unsigned long long>(o.qkey + 8u * (uint32_t)(row * 128 + t));
        const int cq = ldsv<unsigned char>(o.qslot + (uint32_t)(row * 128 + t)) & 31;
        const uint32_t ent_ad = o.list + 8u * (uint32_t)(cq * 16 + li);
        const unsigned long long ent = ldsv<unsigned long long>(ent_ad);
        const unsigned long long front = __ballot(valid && ent < ckey);  // a prefix of the row (the list is ascending)
        const int pos = __popc((uint32_t)(front >> (16 * row)) & 0xffffu);
        const unsigned long long sh = ((unsigned long long)dpp_row_shr1((uint32_t)(ent >> 32)) << 32) | dpp_row_shr1((uint32_t)ent);
        if (valid && li >= pos) ldsv_st<unsigned long long>(ent_ad, li == pos ? ckey : sh);
    }
    }
    const uint32_t changed = (uint32_t)__ballot(lane < 32 && ldsv<unsigned char>(o.chg + (uint32_t)(lane & 31)) != 0);
    if (lane < 32) ldsv_st<unsigned char>(o.chg + (uint32_t)lane, 0);
    // publish the bounds of the slots that changed: lane = slot
    if (lane < 32 && ((changed >> lane) & 1u)) {
        const unsigned long long okey = ldsv<unsigned long long>(o.list + 8u * (uint32_t)(lane * LK + c.km1));
        const unsigned long long jkey = ldsv<unsigned long long>(o.list + 8u * (uint32_t)(lane * LK + (LK == 64 ? 0 : c.jm1)));
        const uint32_t tau_ad = o.tau + 8u * (uint32_t)lane, gkl_ad = o.gkl + 8u * (uint32_t)lane;
        if (okey < ldsv<unsigned long long>(tau_ad)) {
            ldsv_st<unsigned long long>(tau_ad, okey);
            if (okey < ldsv<unsigned long long>(gkl_ad)) {  // tell the other workgroups of this query (the other row slices)
                pend_o = okey;  // (keys only fall)
                ldsv_st<unsigned long long>(gkl_ad, okey);
                const uint32_t nb = q8_bound<M>(okey, ldsv<double>(o.c0 + 8u * (uint32_t)lane), ldsv<double>(o.c1 + 8u * (uint32_t)lane));
                if (nb < q8_ld_bound<M>(o, lane)) q8_st_bound<M>(o, lane, nb);
            }
        }
        if constexpr (LK == 64) {
            if (c.gk2) pend_j = jkey;  // (the list changed: its keys at the four published positions go out with the next batch)
        } else
        if (c.gk2 && jkey != ~0ull) {  // the slice's j smallest keys changed: the sibling slices compute their bound from them
            const uint32_t gjl_ad = o.gjl + 8u * (uint32_t)lane;
            if (jkey < ldsv<unsigned long long>(gjl_ad)) {
                ldsv_st<unsigned long long>(gjl_ad, jkey);
                pend_j = jkey;
            }
        }
    }
}

// (Re)build of a workgroup's table: slot parameters (one thread per slot) from the best bound known for the query, then
// the byte table.  Out of line on purpose: it runs a dozen times per work item, and inlined into the step loop its 40
// live registers made the compiler spill the loop-invariant LDS base registers of the look-ups into the hot path.
template <int M, int NW, int NQ, int LK = 16, bool TL = false>
__device__ __attribute__((noinline)) void q8_rebuild(q8_kernarg_ptr ka, int tile, int first, int slice) {
    constexpr int QT = q8_qt<M, NQ>();
    static_assert(!TL || (M == 16 && NQ == 2 && LK == 16), "cell tiles: the M = 16 kernel with 16-key lists");
    // first bounds of the item's queries: what the seed launch left in the shared array, or -- candidate generator, nothing
    // shared between the slices -- in this slice's row of the per-slice seeds ([n_slices][n_tiles * 32])
    const Q8Build a = {ka->lut, ka->qlom, ka->qstep, ka->smax, ka->qlo,
                       ka->gkey ? ka->gkey : (ka->gseed ? ka->gseed + (int64_t)slice * (ka->n_tiles * QT) : nullptr),
                       ka->Ks, ka->B, ka->k, ka->q8_target, ka->gseed0, ka->btab};
    const bool prebuilt = first && a.btab != nullptr && M == 16 && NQ == 2;  // (TL: btab = the per-query byte tables, always there)
    const int tid = threadIdx.x;
    const Q8Lds o(q8_table_bytes<M, NQ>(a.Ks), LK, q8_ring_entry_bytes<M, LK>());
    if (tid < 32) {  // (the control block has 32 slots whatever QT is: the consumer's lanes 0 .. 31 look at all of them)
        const uint32_t t8 = 8u * (uint32_t)tid, t4 = 4u * (uint32_t)tid;
        int b = tile * QT + tid;
        bool real = tid < QT && b < a.B;
        if constexpr (TL) {  // cell tiles: slot -> query (the per-query arrays below are the REAL queries')
            b = tid < QT ? ka->vmap[tile * QT + tid] : -1;
            real = b >= 0;
            b = real ? b : 0;
            ldsv_st<int32_t>(o.qmap + t4, real ? b : -1);
        }
        unsigned long long key = ~0ull, key_now = ~0ull;
        if (first) {
            if (a.gkey && real) key = __hip_atomic_load(a.gkey + b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if constexpr (TL) key |= 0xffffffffull;  // (a key of ANOTHER cell's list: the distance alone bounds this cell's rows -- see import_bounds)
            key_now = key;
            // (prebuilt table: quantised for the SEED key -- the shared bound may have moved on since; it only tightens T below)
            if (prebuilt && real) key = a.gseed0[b];
            if (key < key_now) key_now = key;
            ldsv_st<unsigned long long>(o.gkl + t8, key_now);
            ldsv_st<unsigned long long>(o.gjl + t8, ~0ull);
            ldsv_st<unsigned long long>(o.tau + t8, ~0ull);
            ldsv_st<unsigned char>(o.chg + (uint32_t)tid, 0);
        } else {
            key = ldsv<unsigned long long>(o.tau + t8);
            const unsigned long long g = ldsv<unsigned long long>(o.gkl + t8);
            if (g < key) key = g;
        }
        float step, inv, clip;
        uint32_t tb;
        q8_slot_params<M>(real, key, real ? a.qstep[b] * (float)(32767 / M) : 0.f, real ? a.smax[b] : 0.f,
                          real ? a.qlo[b] : 0.0, a.target, step, inv, clip, tb);
        ldsv_st<float>(o.step + t4, step);
        ldsv_st<float>(o.inv + t4, inv);
        ldsv_st<float>(o.clip + t4, clip);
        {
            const double slack = real ? (double)a.smax[b] * (2.0 * M * 5.9604644775390625e-08 * (1.0 + 1.0 / 1024.0)) : 0.0;
            const double c1 = (1.0 + 1.0 / 524288.0) / (double)step;
            ldsv_st<double>(o.c1 + t8, c1);
            ldsv_st<double>(o.c0 + t8, (slack - (real ? a.qlo[b] : 0.0)) * c1);
        }
        if (tid < QT) {
            uint32_t tnow = tb;  // the bound to filter with: the table's own T, or what a tighter shared bound gives under its step
            if (prebuilt && real && key_now < key) {
                const uint32_t t2 = q8_bound_from_key<M>(key_now, a.smax[b], step, a.qlo[b]);
                tnow = t2 < tb ? t2 : tb;
            }
            q8_st_bound<M>(o, tid, tnow);
            if constexpr (Q8Cfg<M>::WIDE) ldsv_st<unsigned short>(o.tb + 2u * (uint32_t)tid, (unsigned short)tb);
            else ldsv_st<unsigned char>(o.tb + (uint32_t)tid, (unsigned char)tb);
        }
    }
    __syncthreads();
    if constexpr (Q8Cfg<M>::WIDE) q8_build_table_wide<NW>(a, tile, o.tab, o.inv, o.clip, tid);
    else if constexpr (TL) q8_gather_table<NW>(a.btab, a.Ks, o.tab, o.qmap, tid);  // (only ever the first table: no epoch ends in a cell tile)
    else if constexpr (M == 16 && NQ == 2) {
        if (prebuilt) q8_copy_table<M, NW, NQ>(a, tile, o.tab, tid);
        else q8_build_table<M, NW, NQ>(a, tile, o.tab, o.inv, o.clip, tid);
    } else q8_build_table<M, NW, NQ>(a, tile, o.tab, o.inv, o.clip, tid);
    __syncthreads();
}

// Final merge of a tile by the last of its workgroups to arrive (k <= 16): the k smallest of the n_slices * k keys the slices
// left in `partial`, by RANK COUNTING.  One wave per query; the keys of a query are contiguous, 64 per chunk (lane = key);
// a chunk is first cut down to the keys below the running k-th one, then every survivor and every entry of the running
// list learns its rank in the union from broadcasts (v_readlane of the few survivors) and writes itself to that slot of a
// 16-entry LDS scratch.  (merge_tile_slices folds one slice at a time through 64-lane bitonic merges, ds_bpermute chains:
// 19 us for 32 queries x 8 slices -- on the critical path of every launch, the merging workgroup is the last to leave.)
template <int NW>
__device__ __forceinline__ void q8_merge_tile(const ScanArgs &a, int b0, int QT, int km1, int wave, int lane, uint32_t scratch_ad) {
    int nq = a.B - b0;
    if (nq > QT) nq = QT;
    const int k = km1 + 1;
    const int total = a.n_slices * k;
    const uint32_t my_scratch = scratch_ad + (uint32_t)wave * 128u;  // u64 [16]
    auto rd64 = [&](unsigned long long v, int l) -> unsigned long long {
        return ((unsigned long long)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(v >> 32), l) << 32) |
               (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)v, l);
    };
    // (query, chunk) pairs of this wave, in order; the keys of pair p + 2 are requested before pair p is processed: the
    // loads are device-scope (another XCD's L2 does not see them sooner) and take ~2.5 us each -- chained, 4 of them were
    // most of the merging workgroup's 15-25 us
    const int nch = (total + 63) / 64;
    const int n_mine = wave < nq ? (nq - wave + NW - 1) / NW : 0;
    const int n_pairs = n_mine * nch;
    auto load_pair = [&](int p) -> unsigned long long {
        if (p >= n_pairs) return ~0ull;
        const int b = b0 + wave + NW * (p / nch), idx = 64 * (p % nch) + lane;
        if (idx >= total) return ~0ull;
        return __hip_atomic_load(a.partial + (int64_t)b * total + idx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    };
    unsigned long long n1 = load_pair(0), n2 = load_pair(1);
    unsigned long long best = ~0ull;  // lane j < k: the j-th smallest key so far
    // (first cut, see below: whole slices in the first chunk, and the smallest j with n_full * j >= k)
    const int n_full = (total < 64 ? total : 64) / k;
    const int cut_j = n_full > 0 ? (k + n_full - 1) / n_full : 0;  // (<= k whenever n_full >= 1)
#pragma unroll 1
    for (int p = 0; p < n_pairs; ++p) {
        unsigned long long key = n1;
        n1 = n2;
        n2 = load_pair(p + 2);
        const int c = p % nch;
        if (c == 0) {
            best = ~0ull;  // a new query
            // First cut of the first chunk (nothing to compare with yet: all 64 keys would go through the serial ranking
            // below, ~3 us per query on the critical path of the launch).  The slices' lists are ascending and n_full of them
            // lie wholly in this chunk: their j smallest keys, n_full * j >= k keys of distinct rows, are all <= the largest
            // j-th key -- so is the k-th smallest of the union.
            if (cut_j > 0) {
                unsigned long long cut = 0ull;
                for (int sl = 0; sl < n_full; ++sl) {
                    const unsigned long long v = rd64(key, sl * k + cut_j - 1);
                    cut = v > cut ? v : cut;
                }
                if (key > cut) key = ~0ull;
            }
        }
        const unsigned long long thr = rd64(best, km1);
        if (!(key < thr)) key = ~0ull;
        unsigned long long m = __ballot(key != ~0ull);
        if (m) {
            int r_c = 0, r_b = lane;  // ranks in the union of my chunk key / my list entry
            while (m) {
                const int L = __builtin_ctzll(m);
                m &= m - 1ull;
                const unsigned long long ck = rd64(key, L);
                r_c += ck < key ? 1 : 0;
                r_b += ck < best ? 1 : 0;
            }
#pragma unroll 1
            for (int j = 0; j < k; ++j) r_c += rd64(best, j) < key ? 1 : 0;
            if (lane < k && r_b < k) ldsv_st<unsigned long long>(my_scratch + 8u * (uint32_t)r_b, best);
            if (key != ~0ull && r_c < k) ldsv_st<unsigned long long>(my_scratch + 8u * (uint32_t)r_c, key);
            best = lane < k ? ldsv<unsigned long long>(my_scratch + 8u * (uint32_t)lane) : ~0ull;
        }
        if (c == nch - 1 && lane <= km1) {  // the query is complete
            const int b = b0 + wave + NW * (p / nch);
            const uint32_t hi = (uint32_t)(best >> 32), lo = (uint32_t)best;
            const bool none = (hi == kKeyInfHi && lo == kIdNone);
            const float d = none ? __builtin_inff() : ordered_to_f32(hi);
            const int64_t id = none ? (int64_t)-1 : a.row_base + (int64_t)lo;
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

// ROW QUEUE (M = 16, ANNLITE_Q8_ROWQ).  A scanning wave's step with a candidate used to enumerate its (lane, query) pairs in
// scalar code -- ~200 dynamic instructions per hit lane, 0.65 us per such step, during which the wave issues no look-ups:
// 31 of the 211 us of the step loop at a 1.25M-row shard, where a quarter of the wave-steps have a candidate.  Now it pushes
// only the ROW ids of its hit lanes (one ballot, lane-parallel stores) and the CONSUMER finds the queries: a popped row's
// byte sums against all 32 queries are recomputed from the table in LDS -- the scanning waves' own 32 look-ups, for up to
// 64 rows at once, one per lane -- and filtered with the bounds of NOW (tighter than at push time).  What comes out are the
// entries (slot << 32 | row) q8_consume has always taken.
// The row is rotated to the CONSUMER lane's skew (lane l reads sub-space (l + t) mod 16 at byte t: conflict-free look-ups).
template <bool SKEWED>
__device__ __forceinline__ void q8_row_pass_mask(const uint8_t *codes, uint32_t rid, bool act, int lane, uint32_t lds0, uint32_t shq_ad,
                                                 uint32_t row_park, uint32_t &mask) {
    // (q8_entry16: the second entry group of a look-up sits 16 bytes behind the first, a code's row is 256 bytes)
    constexpr int M = 16, CW = 4, NQ = 2, RB = 16, DEPTH = ANNLITE_Q8_STAGE_DEPTH, TOT = NQ * M;
    typedef const ANNLITE_LDS u32x4 *lds_entry_ptr;
    uint32_t cc[CW];
    {
        const u32x4 v = *(const u32x4 __attribute__((address_space(1))) *)(uintptr_t)(codes + (int64_t)(act ? rid : 0u) * M);
        cc[0] = v.x, cc[1] = v.y, cc[2] = v.z, cc[3] = v.w;
        ldsv_st<u32x4>(row_park, v);  // (the stored form: q8_consume's exact sums take it from here instead of a second global round trip)
    }
    const int s = lane % M;
    const int rot = SKEWED ? ((s - (int)(rid % M)) & (M - 1)) : s;  // stored byte j = code of sub-space (j + rid) mod M resp. j
    bool abit[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) abit[i] = (((rot >> 2) >> i) & 1) != 0;
    rotate_row<CW>(cc, abit, (uint32_t)(rot & 3));
    uint32_t addr[M];
    static_for<0, CW>([&](auto W) {
        constexpr int w = decltype(W)::value;
        uint32_t o0, o1, o2, o3;
        byte_shl4(cc[w], 8u, o0, o1, o2, o3);
        addr[4 * w + 0] = lds0 + q8_entry16(0u, (uint32_t)((s + 4 * w + 0) % M), 0u) + o0;
        addr[4 * w + 1] = lds0 + q8_entry16(0u, (uint32_t)((s + 4 * w + 1) % M), 0u) + o1;
        addr[4 * w + 2] = lds0 + q8_entry16(0u, (uint32_t)((s + 4 * w + 2) % M), 0u) + o2;
        addr[4 * w + 3] = lds0 + q8_entry16(0u, (uint32_t)((s + 4 * w + 3) % M), 0u) + o3;
    });
    u32x4 acc[NQ];
    u32x4 v[DEPTH];
    static_for<0, DEPTH>([&](auto I) {
        constexpr int i = decltype(I)::value;
        v[i] = *(lds_entry_ptr)(uintptr_t)(addr[i % M] + (uint32_t)((i / M) * RB));
    });
    static_for<0, TOT>([&](auto I) {
        constexpr int i = decltype(I)::value;
        asm volatile("" ::: "memory");
        if constexpr (i % M == 0) acc[i / M] = v[i % DEPTH];
        else acc[i / M] += v[i % DEPTH];
        if constexpr (i + DEPTH < TOT) {
            constexpr int j = i + DEPTH;
            v[i % DEPTH] = *(lds_entry_ptr)(uintptr_t)(addr[j % M] + (uint32_t)((j / M) * RB));
        }
    });
    // bit q of the mask: query slot q passes -- S <= T of NOW (the scanning waves' filter; slot = 4 * dword + byte)
    mask = 0u;
#pragma unroll
    for (int h = 0; h < NQ; ++h) {
        const u32x4 t = *(volatile ANNLITE_LDS u32x4 *)(uintptr_t)(shq_ad + 16u * (uint32_t)h);
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            const uint32_t sm = acc[h][w];
            const uint32_t hb = ((t[w] - (sm & 0x7f7f7f7fu)) & ~sm & 0x80808080u) >> 7;  // bit 8 j <- byte j passes
            // gather the four byte flags into bits 0..3: 0x01010101-spaced bits times 0x00204081 puts them at bits 21..24
            const uint32_t nib = ((hb * 0x00204081u) >> 21) & 0xfu;
            mask |= nib << (uint32_t)(16 * h + 4 * w);
        }
    }
    if (!act) mask = 0u;
}

// Both popped rows of a lane; OUT OF LINE: inlined into the consumer branch its ~110 live registers made the allocator spill
// three of the scanning branch's loop-invariant LDS bases into the step loop (scripts/check_q8_isa.sh).  Low word: row 0's mask.
template <bool SKEWED>
__device__ __attribute__((noinline)) unsigned long long q8_rows_pass_masks(const uint8_t *codes, uint32_t rid0, uint32_t rid1,
                                                                           uint32_t act_bits, uint32_t lds0, uint32_t shq_ad,
                                                                           uint32_t rowq_ad) {
    const int lane = (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
    const uint32_t park = rowq_ad + 48u * (uint32_t)lane;
    uint32_t m0, m1;
    q8_row_pass_mask<SKEWED>(codes, rid0, (act_bits & 1u) != 0u, lane, lds0, shq_ad, park + 16u, m0);
    q8_row_pass_mask<SKEWED>(codes, rid1, (act_bits & 2u) != 0u, lane, lds0, shq_ad, park + 32u, m1);
    return (unsigned long long)m0 | ((unsigned long long)m1 << 32);
}

// 16 < k <= 64 (64-key lists): the bound the sibling slices' published keys imply for query b.  A slice publishes the keys at four
// list positions p0 < p1 < p2 < p3 (ScanArgs::q8_pos): cell i says "this slice holds p_i + 1 rows at or below this key".  For a
// threshold t the slices of a group of 8 therefore hold at least
//     count(t) = sum over their cells <= t of w_i,   w_i = p_i - p_(i-1)   (p_(-1) = -1)
// rows at or below t: the cells of a slice are ascending, so the increments of its cells <= t add up to p_i + 1 of the largest
// of them (a reader that mixes two versions of a slice's cells only undercounts: every cell's own claim holds whatever the
// others say, keys only fall).  The smallest published key with count >= k has k rows of the table at or below it (+ 1: that row
// itself must still be accepted).  With 8 slices and positions (4, 6, 8, 12) the bound sits near global rank 56 for k = 50 -- the
// MAX of the slices' 7th keys (the rule for j > 2 below) near rank 89, and the candidates are in proportion (rank^1.6).
// Lane = (slot, half): a lane holds the 4 x 4 cells of four slices and ranks them against all 32 of the group.  Out of line: ~60
// live registers that must not reach the scanning branch's allocation.
__device__ __attribute__((noinline)) unsigned long long q8_weighted_bound(const unsigned long long *gk2, int b, int n_slices, int k,
                                                                          uint32_t pos, bool real) {
    const int lane = (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
    const int part = lane >> 5;
    int w[4];
    w[0] = (int)(pos & 0xffu) + 1;
#pragma unroll
    for (int i = 1; i < 4; ++i) w[i] = (int)((pos >> (8 * i)) & 0xffu) - (int)((pos >> (8 * (i - 1))) & 0xffu);
    unsigned long long best = ~0ull;
#pragma unroll 1
    for (int g0 = 0; g0 < n_slices; g0 += 8) {
        unsigned long long kk[16];
#pragma unroll
        for (int sl4 = 0; sl4 < 4; ++sl4) {
            const int sl = g0 + part * 4 + sl4;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                kk[4 * sl4 + i] = ~0ull;
                if (real && sl < n_slices)
                    kk[4 * sl4 + i] = __hip_atomic_load(gk2 + ((int64_t)b * n_slices + sl) * kGk2Keys + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
        int cnt[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) cnt[i] = 0;
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const unsigned long long oj = __shfl_xor(kk[j], 32);  // the other half's cell j
            const int wj = w[j & 3];
#pragma unroll
            for (int i = 0; i < 16; ++i) cnt[i] += (kk[j] <= kk[i] ? wj : 0) + (oj <= kk[i] ? wj : 0);
        }
        unsigned long long found = ~0ull;
#pragma unroll
        for (int i = 0; i < 16; ++i)
            if (cnt[i] >= k && kk[i] < found) found = kk[i];  // (an empty cell, ~0, is never below `found`)
        const unsigned long long o = __shfl_xor(found, 32);
        found = o < found ? o : found;
        if (found != ~0ull && found + 1ull < best) best = found + 1ull;
    }
    return best;
}

// work item -> (query tile, row slice).  item % 8 == the XCD the block lands on (speed only).  With >= 8 query tiles an
// XCD owns the tiles congruent to it, for ALL row slices: the fp32 tables the exact sums gather from (16 KB per query,
// 512 KB per tile) stay in that XCD's 4 MB L2 -- with the slice-per-XCD map of the u16 kernels (item_map) every XCD saw
// all the tables of the batch (16 MB at 1024 queries) and each of a candidate's 16 gathers was a cache-line fetch from
// the fabric: ~190 ns per candidate, the consumer wave's whole budget.  The code rows are then read by every XCD
// (8 x the table per launch, 1.3 GB at 10M rows: a fraction of a ms of HBM / Infinity Cache bandwidth).
__device__ __forceinline__ bool q8_item_map(const ScanArgs &a, int item, int &tile, int &slice) {
    if (a.n_tiles < 8 || a.q8_map_slices) return item_map(a, item, tile, slice);
    const int xcd = item & 7, j = item >> 3;
    const int tpx = (a.n_tiles + 7) >> 3;
    slice = j / tpx;
    tile = xcd + 8 * (j - slice * tpx);
    return tile < a.n_tiles && slice < a.n_slices;
}

// EARLY MERGER.  With one work item per CU the launch ends when the slowest tile has merged: its last workgroup to arrive used
// to merge all n_slices lists of the tile's queries then -- ~13 us (device-scope loads of n_slices * k keys per query, rank
// counting) on the critical path of EVERY launch, while the tile's first workgroup had left 20-40 us earlier.  Now the FIRST
// workgroup to finish stays: its own lists (LDS) are the running result, it polls the tile's arrival mask and folds the lists
// of the slices that have come in since (up to 64 / k slices per pass, one key per lane, the same rank counting) -- when the
// last slice arrives one short pass is left.  Only where every work item has its own workgroup (n_items <= grid) and
// n_slices <= 31 (one mask word + the flag below).  A waiting merger holds its CU, and the workgroups it waits for may not be
// resident yet (other launches on other streams): if EVERY resident workgroup were such a merger nothing would move.  So a
// merger that has seen no arrival for ScanArgs::q8_merge_patience ticks (200 us) LEAVES: it clears bit 31 of the mask ("no merger") and the
// workgroup that then clears the last slice bit merges everything from global memory, as before (q8_merge_tile).  The two
// atomics serialise: exactly one of them finishes the tile.
constexpr uint32_t kEarlyMergerBit = 0x80000000u;
template <int NW>
__device__ __forceinline__ void q8_early_merge(const ScanArgs &a, const Q8Lds &lds, int tile, int slice, int QT, int wave, int lane) {
    const int tid = threadIdx.x, k = a.k, km1 = k - 1, b0 = tile * QT;
    int nq = a.B - b0;
    if (nq > QT) nq = QT;
    const uint32_t low = (1u << a.n_slices) - 1u;
    uint32_t merged = 1u << slice;
    unsigned long long t_last = wall_clock64();
    const int per_pass = 64 / k;
    const uint32_t my_scratch = lds.ring + (uint32_t)wave * 128u;  // u64 [16]
    auto rd64 = [&](unsigned long long v, int l) -> unsigned long long {
        return ((unsigned long long)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(v >> 32), l) << 32) |
               (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)v, l);
    };
    // Every decision of the loop is WORKGROUP-uniform: thread 0 polls the mask, reads the clock and leaves both the mask and its
    // verdict ("patience is over") in LDS; all waves branch on those words after the barrier.  (Each wave comparing its own
    // clock reading let waves straddle the threshold: some left while others went on merging -- only for their own queries.)
    for (;;) {
        __syncthreads();
        if (tid == 0) {
            const uint32_t m = __hip_atomic_load(a.tile_done + tile, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const bool nothing_new = (~m & low & ~merged) == 0u && (~m & low) != low;
            ldsv_st<uint32_t>(lds.ctl + 4, m);
            ldsv_st<uint32_t>(lds.ctl + 8, (nothing_new && wall_clock64() - t_last >= (unsigned long long)a.q8_merge_patience) ? 1u : 0u);
        }
        __syncthreads();
        const uint32_t mask = ldsv<uint32_t>(lds.ctl + 4);  // bit s set: slice s is still out
        const bool timed_out = ldsv<uint32_t>(lds.ctl + 8) != 0u;
        const uint32_t arrived = ~mask & low & ~merged;
        if (!arrived) {
            if ((~mask & low) == low) break;  // every slice is in and merged
            if (timed_out) {
                // nothing has arrived for a long time: leave, unless everything turns out to be in (then finish here)
                __syncthreads();
                if (tid == 0)
                    ldsv_st<uint32_t>(lds.ctl + 4, __hip_atomic_fetch_and(a.tile_done + tile, ~kEarlyMergerBit, __ATOMIC_RELAXED,
                                                                          __HIP_MEMORY_SCOPE_AGENT));
                __syncthreads();
                if ((ldsv<uint32_t>(lds.ctl + 4) & low) != 0u) return;  // the last slice to arrive merges the tile (q8_finish_item)
                t_last = wall_clock64();
                continue;
            }
            __builtin_amdgcn_s_sleep(32);
            continue;
        }
        t_last = wall_clock64();
        // this pass: the per_pass lowest arrived slices, slice j of the pass in lanes [j k, (j + 1) k)
        int my_sl = -1;
        uint32_t take = 0u, rest = arrived;
        for (int j = 0; j < per_pass && rest; ++j) {
            const int s = __builtin_ctz(rest);
            rest &= rest - 1u;
            take |= 1u << s;
            if (lane >= j * k && lane < (j + 1) * k) my_sl = s;
        }
        unsigned long long key[2];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int q = wave + NW * u;
            key[u] = ~0ull;
            if (q < nq && my_sl >= 0)
                key[u] = __hip_atomic_load(a.partial + ((int64_t)(b0 + q) * a.n_slices + my_sl) * k + (lane % k), __ATOMIC_RELAXED,
                                           __HIP_MEMORY_SCOPE_AGENT);
        }
#pragma unroll 1
        for (int u = 0; u < 2; ++u) {
            const int q = wave + NW * u;
            if (q >= nq) continue;
            const uint32_t lst = lds.list + 8u * (uint32_t)(q * 16);
            unsigned long long best = lane < k ? ldsv<unsigned long long>(lst + 8u * (uint32_t)lane) : ~0ull;
            unsigned long long ky = key[u];
            if (!(ky < rd64(best, km1))) ky = ~0ull;
            unsigned long long m = __ballot(ky != ~0ull);
            if (!m) continue;
            int r_c = 0, r_b = lane;  // ranks in the union of my new key / my list entry
            while (m) {
                const int L = __builtin_ctzll(m);
                m &= m - 1ull;
                const unsigned long long ck = rd64(ky, L);
                r_c += ck < ky ? 1 : 0;
                r_b += ck < best ? 1 : 0;
            }
#pragma unroll 1
            for (int j = 0; j < k; ++j) r_c += rd64(best, j) < ky ? 1 : 0;
            if (lane < k && r_b < k) ldsv_st<unsigned long long>(my_scratch + 8u * (uint32_t)r_b, best);
            if (ky != ~0ull && r_c < k) ldsv_st<unsigned long long>(my_scratch + 8u * (uint32_t)r_c, ky);
            if (lane < k) ldsv_st<unsigned long long>(lst + 8u * (uint32_t)lane, ldsv<unsigned long long>(my_scratch + 8u * (uint32_t)lane));
        }
        merged |= take;
    }
    for (int q = wave; q < nq; q += NW) {
        if (lane > km1) continue;
        const unsigned long long best = ldsv<unsigned long long>(lds.list + 8u * (uint32_t)(q * 16 + lane));
        const int b = b0 + q;
        const uint32_t hi = (uint32_t)(best >> 32), lo = (uint32_t)best;
        const bool none = (hi == kKeyInfHi && lo == kIdNone);
        const float d = none ? __builtin_inff() : ordered_to_f32(hi);
        const int64_t id = none ? (int64_t)-1 : a.row_base + (int64_t)lo;
        if (a.out_packed) {
            a.out_packed[((int64_t)b * a.k + lane) * 2 + 0] = id;
            a.out_packed[((int64_t)b * a.k + lane) * 2 + 1] = (int64_t)__float_as_uint(d);
        } else {
            a.out_d[(int64_t)b * a.k + lane] = a.sqrt_out ? __builtin_sqrtf(d) : d;
            a.out_i[(int64_t)b * a.k + lane] = id;
        }
    }
}

// End of a work item: the lists ARE the workgroup's result for this (tile, slice) -- the final epoch_sync was the barrier:
// every candidate is in --; the last of the tile's workgroups to arrive merges the slices.  Out of line, arguments from the
// kernarg segment (see q8_kernarg).
template <int M, int NW, int NQ, int LK = 16, bool TL = false>
__device__ __attribute__((noinline)) void q8_finish_item(q8_kernarg_ptr ka, int tile, int slice) {
    constexpr int QT = q8_qt<M, NQ>();
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int km1 = ka->k - 1, B = ka->B, n_slices = ka->n_slices, k = ka->k;
    const Q8Lds lds(q8_table_bytes<M, NQ>(ka->Ks), LK, q8_ring_entry_bytes<M, LK>());
    unsigned long long *partial = ka->partial;
    for (int q = wave; q < QT; q += NW) {
        const int b = tile * QT + q;
        // device-scope stores: the merging workgroup may sit on another XCD (own L2)
        // (TL: b is the SLOT -- every slot of a used tile leaves its list, a pad slot's is empty; annlite_ivf_merge reads them by slot)
        if ((TL || b < B) && lane <= km1)
            __hip_atomic_store(partial + ((int64_t)b * n_slices + slice) * k + lane,
                               ldsv<unsigned long long>(lds.list + 8u * (uint32_t)(q * LK + lane)), __ATOMIC_RELAXED,
                               __HIP_MEMORY_SCOPE_AGENT);
    }
    unsigned int *tile_done = ka->tile_done;
    if constexpr (LK == 16 && !TL)  // (64-key lists: the slices are merged by merge_partial_kernel -- the in-kernel merges hold 16 keys per lane row)
    if (tile_done) {
        // the last of the tile's n_slices workgroups to arrive merges them
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's list stores have completed
        __syncthreads();
        const bool early = ka->q8_early_merge != 0;
        if (tid == 0) {
            if (early) {
                // tile_done: a mask of the slices still out + bit 31 "the early merger is there" (all-ones from the fill).  The FIRST
                // to arrive becomes the early merger (1); after it has left, the one that clears the last slice bit merges all (2)
                const unsigned int old = __hip_atomic_fetch_and(tile_done + tile, ~(1u << slice), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                const unsigned int low = (1u << n_slices) - 1u;
                ldsv_st<uint32_t>(lds.ctl + 4, old == 0xffffffffu ? 1u : (!(old & kEarlyMergerBit) && (old & low) == (1u << slice)) ? 2u : 0u);
            } else {      // ... a counter from -1: the LAST to arrive merges
                const unsigned int old = __hip_atomic_fetch_add(tile_done + tile, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                ldsv_st<uint32_t>(lds.ctl + 4, (old + 1u == (unsigned int)(n_slices - 1)) ? 1u : 0u);
            }
        }
        __syncthreads();
        if (ldsv<uint32_t>(lds.ctl + 4)) {
            ScanArgs a;  // (cold path: the merging workgroup reads the block once)
            __builtin_memcpy(&a, (const void *)ka, sizeof(ScanArgs));
            if (early && ldsv<uint32_t>(lds.ctl + 4) == 1u) q8_early_merge<NW>(a, lds, tile, slice, QT, wave, lane);
            else q8_merge_tile<NW>(a, tile * QT, QT, km1, wave, lane, lds.ring);
        }
        __syncthreads();
    }
}

// TL (cell tiles, annlite_ivf_search_topk -- DESIGN 8c): a work item = one tile of up to 32 (query, cell) pairs that probe the SAME cell,
// scanning that cell's rows (ScanArgs::tile_rows); slot q scans with the tables of query vmap[tile * 32 + q] (Q8Lds::qmap), the image is
// assembled from the queries' own byte tables (q8_gather_table), gkey is indexed by QUERY -- the tiles of a query's other cells import
// its bound --, no epoch ends (a cell is a few dozen steps), every slot's list goes to `partial` by slot (annlite_ivf_merge).
template <int M, int NW, bool SKEWED, int NQ, int CB, bool RQ, int LK = 16, bool TL = false>
__global__ __launch_bounds__(NW * 64, NW / 4) void adc_scan_q8_kernel(const ScanArgs a) {
    static_assert(!TL || (M == 16 && NQ == 2 && CB == 1 && RQ && LK == 16), "cell tiles: the M = 16 row-queue kernel with 16-key lists");
    static_assert(LK == 16 || (LK == 64 && (RQ || M != 16) && M != 64), "64-key lists: M = 16 with the row queue (4-byte ring entries), M = 8 / 32");
    constexpr uint32_t RE = (uint32_t)q8_ring_entry_bytes<M, LK>();  // bytes of a ring entry
    constexpr bool WIDE = Q8Cfg<M>::WIDE;
    constexpr bool M8 = Q8Cfg<M>::M8, M32 = Q8Cfg<M>::M32, C16 = CB == 2;
    constexpr bool ROWQ = RQ && ANNLITE_Q8_ROWQ != 0;  // (row queue: see q8_row_pass_mask)
    static_assert(!RQ || (M == 16 && NQ == 2 && CB == 1), "the row queue is the M = 16 kernel's");
    constexpr int QT = q8_qt<M, NQ>(), CW = M * CB / 4, EB = 16, RB = M * EB, KSTRIDE = NQ * RB;
    static_assert(CB == 1 || (CB == 2 && M8), "uint16 codes: the M = 8 shapes");
    static_assert(NQ == 2 || (NQ == 1 && M8 && C16) || (NQ == 1 && M32 && !C16),
                  "one entry group: the M = 8 / uint16 shape above Ks = 512, and M = 32");
    static_assert(!M32 || NQ == 1, "M = 32: 16 queries per workgroup");
    constexpr int NS = NW - 1;  // scanning waves; wave NS is the consumer
    static_assert(M % 8 == 0 && (M <= 32 || WIDE) && (KSTRIDE & (KSTRIDE - 1)) == 0 && !(C16 && SKEWED), "unsupported shape");

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int km1 = a.k - 1;

    const int lut_bytes = q8_table_bytes<M, NQ>(a.Ks);
    const Q8Lds lds(lut_bytes, LK, (int)RE);
    const q8_kernarg_ptr ka = q8_kernarg();

    if (a.guard && tid == 0) ldsv_st<uint32_t>(lds.seen, 0u);
    // (measurement hook, read from the kernarg segment: see q8_kernarg; not in the M = 8 shapes -- there the two scalar
    // reads moved the allocator's choices and a scratch reload appeared in the 851 shape's step loop)
    if constexpr (!M8) {
        if (blockIdx.x == 0 && tid == 0) {
            unsigned long long *clk = ka->clk;
            if (clk) clk[0] = __builtin_readcyclecounter(), clk[1] = wall_clock64();
        }
    }
    for (int it = 0;; ++it) {
        int item = blockIdx.x + it * gridDim.x;
        // TL: the tiles come longest first; a workgroup's first one is its own index, the later ones are DRAWN from a device counter (the
        // consumer wave requests the next number at the start of an item -- behind its first import -- and leaves it in LDS)
        if constexpr (TL) {
            if (it > 0 && a.item_counter) item = (int)ldsv<uint32_t>(lds.seen + 8);
            // (the first items of one XCD's workgroups as CONSECUTIVE tiles -- the 2-3 tiles of a cell streaming its rows through one L2 --
            // measured: no difference, profiles/r06/ivf_xcd_first_ab.txt)
        }
        if (item >= a.n_items) break;
        int tile, slice;
        if constexpr (TL) {
            tile = item, slice = 0;
            if (a.tile_rows[2 * (int64_t)tile] < 0) break;  // (tiles past the last used one: the used ones come first)
        } else if (!q8_item_map(a, item, tile, slice)) continue;
        if (a.guard && it > 0) {  // a launch that gave up (below) is redone by the launch behind it: no further items
            __syncthreads();
            if (tid == 0) ldsv_st<uint32_t>(lds.seen + 4, __hip_atomic_load(a.guard, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
            __syncthreads();
            if (ldsv<uint32_t>(lds.seen + 4) == 0u) break;
        }
        int64_t slice_begin = (int64_t)slice * a.slice_rows;
        int64_t slice_end = slice_begin + a.slice_rows;
        if constexpr (TL) slice_begin = a.tile_rows[2 * (int64_t)tile], slice_end = a.tile_rows[2 * (int64_t)tile + 1];  // the tile's cell
        if (slice_end > a.N) slice_end = a.N;
        // blocks of 64 rows of this work item: its contiguous range, or (q8_ilv_log > 0) the runs of G = 2^q8_ilv_log blocks number
        // slice, slice + n_slices, ... of the table's ceil(N / 64) blocks
        uint32_t item_blocks;
        if (a.q8_ilv_log > 0) {
            const uint32_t G = 1u << a.q8_ilv_log, per = (uint32_t)a.n_slices << a.q8_ilv_log;
            const uint32_t nb = (uint32_t)((a.N + 63) >> 6), rounds = nb / per, rest = nb - rounds * per;
            const uint32_t mine = rest > (uint32_t)slice * G ? rest - (uint32_t)slice * G : 0u;
            item_blocks = rounds * G + (mine < G ? mine : G);
        } else item_blocks = slice_end > slice_begin ? (uint32_t)((slice_end - slice_begin + 63) >> 6) : 0u;
        const int n_steps = (int)((item_blocks + (uint32_t)NS - 1u) / (uint32_t)NS);

        __syncthreads();  // every wave is done with the previous item
        // ANNLITE_DEBUG_COUNTERS: phase stamps of thread 0 (100 MHz wall clock; kept in LDS: four live 64-bit values pushed the
        // step loop over its register budget), summed over the work items in dbg[8..15]
        const uint32_t stamp_ad = lds.stamps;
        auto stamp = [&](int i) {
            if (a.dbg && tid == 0) ldsv_st<unsigned long long>(stamp_ad + 8u * (uint32_t)i, wall_clock64());
        };
        stamp(0);
        // end of an epoch (all waves): the consumer arrives last, with every pushed candidate in the lists.  Then: have
        // the bounds outrun the table?  A slot's T falls as its threshold tightens under a fixed step; rebuild when a
        // quarter of the real slots are below q8_rebuild_8ths / 8 of what their table was built for.
        auto epoch_sync = [&](bool final) {
            __syncthreads();
            if (final || TL) return;
            if (wave == 0) {
                const int q = lane & (QT - 1);
                const bool real = tile * QT + q < a.B && lane < QT;
                const uint32_t tn = q8_ld_bound<M>(lds, q) & Q8Cfg<M>::TMAX, tb = q8_ld_built<M>(lds, q) & Q8Cfg<M>::TMAX;
                const bool need = real && tn * 8u < tb * (uint32_t)a.q8_rebuild_8ths;
                const int n_need = __popcll(__ballot(need)), n_real = __popcll(__ballot(real));
                if (lane == 0) ldsv_st<uint32_t>(lds.ctl, (n_need > 0 && n_need * 4 >= n_real) ? 1u : 0u);
            }
            __syncthreads();
            if (ldsv<uint32_t>(lds.ctl)) {
                q8_rebuild<M, NW, NQ, LK, TL>(ka, tile, 0, slice);
                if (a.dbg && tid == 0) atomicAdd(a.dbg + 5, 1ull);
            }
        };
        for (int idx = tid; idx < 32 * LK; idx += NW * 64) ldsv_st<unsigned long long>(lds.list + 8u * (uint32_t)idx, ~0ull);  // (all 32 slots)
        if (tid < 16) {
            ldsv_st<uint32_t>(lds.tails() + 4u * (uint32_t)tid, 0);
            ldsv_st<uint32_t>(lds.heads() + 4u * (uint32_t)tid, 0);
        }
        if (tid == 0) {
            ldsv_st<uint32_t>(lds.arrived(), 0);
            ldsv_st<uint32_t>(lds.blk_ctr(), 0);  // the block counter the scanning waves draw from
        }
        q8_rebuild<M, NW, NQ, LK, TL>(ka, tile, 1, slice);  // (its barriers cover the initialisation above)
        stamp(1);

        // epochs end after steps q8_epoch0, q8_epoch0 * mul + (mul - 1), ... and after the last step
        if (wave == NS) {
            // ------------------------------------------------------------------------------- consumer wave
            const FlushCtx fc = {(const uint8_t *)a.codes, a.lut, a.smax, a.qstep, a.qlo, (TL && a.tl_private) ? nullptr : a.gkey, a.gk2, nullptr,
                                 a.Ks, tile * QT, a.n_slices, slice, km1, a.jm1, a.dbg_skip,
                                 0u, 0u, 0u, 0u, 0u, 0u, a.q8_pos};
            // what the other workgroups of these queries (the other row slices) have proven: the best k-th key any of
            // them published and, per group of 8 concurrently scanned slices, the k-th smallest of the keys they published
            // (below; +1: that row itself must still be accepted)
            auto import_bounds = [&]() {
                if (!a.gkey) return;
                if constexpr (TL) {
                    if (a.tl_private) return;  // (private lists: the slot's first bound, read when the image was built, and its own k-th key)
                }
                // lane = (slot, half): all loads in flight together -- one global round trip per group of 8 slices for the
                // whole tile
                const int q = lane & 31, part = lane >> 5;
                int b = tile * QT + q;
                bool real = q < QT && b < a.B;
                if constexpr (TL) {
                    b = ldsv<int32_t>(lds.qmap + 4u * (uint32_t)q);
                    real = b >= 0;
                }
                unsigned long long bound = ~0ull;
                if (real) bound = __hip_atomic_load(a.gkey + b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                // (TL: the key comes from ANOTHER cell's list, and the final order is by external id, which is ascending inside a cell only:
                // a row of this cell at the same distance must still be accepted -- the bound is the distance alone)
                if constexpr (TL) bound |= 0xffffffffull;
                if constexpr (LK == 64) {
                    if (a.gk2 && a.n_slices > 1) {
                        const unsigned long long wb = q8_weighted_bound(a.gk2, b, a.n_slices, a.k, a.q8_pos, real);
                        if (wb < bound) bound = wb;
                    }
                } else
                if (a.gk2 && a.jm1 < 2) {
                    // The G <= 8 concurrently scanned slices publish their j = ceil(k / G) <= 2 smallest keys: G j >= k keys
                    // of distinct rows, so the k-th smallest of them has k rows at or below it (+1: that row itself must
                    // still be accepted).  (The MAX of the slices' j-th keys, round 1's rule and the M = 64 kernel's, is the LARGEST of
                    // these keys: with 8 slices and k = 10 it sits near global rank 36, the 10th smallest of the 16 near
                    // rank 13 -- the candidates that pass the imported bound are in proportion.)
#pragma unroll 1
                    for (int g0 = 0; g0 < a.n_slices; g0 += 8) {
                        unsigned long long k0[8], k1[8];
#pragma unroll
                        for (int i = 0; i < 8; ++i) {
                            k0[i] = ~0ull;
                            k1[i] = ~0ull;
                            if (real && g0 + i < a.n_slices) {
                                const unsigned long long *cell = a.gk2 + ((int64_t)b * a.n_slices + g0 + i) * kGk2Keys;
                                k0[i] = __hip_atomic_load(cell, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                                if (a.jm1 == 1) k1[i] = __hip_atomic_load(cell + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            }
                        }
#pragma unroll
                        for (int i = 0; i < 8; ++i)
                            if (k1[i] == k0[i]) k1[i] = ~0ull;  // (two versions of the slice's list mixed: the same row twice)
                        // half 0 ranks the first keys, half 1 the second ones, each against all 16 (keys of distinct rows
                        // are distinct); the one of rank k - 1 is the bound
                        unsigned long long found = ~0ull;
#pragma unroll
                        for (int i = 0; i < 8; ++i) {
                            const unsigned long long me = part ? k1[i] : k0[i];
                            int rank = 0;
#pragma unroll
                            for (int jj = 0; jj < 8; ++jj) rank += (k0[jj] < me) + (k1[jj] < me);
                            if (rank == km1 && me != ~0ull) found = me;
                        }
                        const unsigned long long o = __shfl_xor(found, 32);
                        found = o < found ? o : found;
                        if (found != ~0ull && found + 1ull < bound) bound = found + 1ull;
                    }
                } else if (a.gk2) {
                    // j > 2 (fewer than 8 slices): the MAX of the slices' j-th keys (G disjoint slices x j rows >= k rows at
                    // or below it; +1 as above)
#pragma unroll 1
                    for (int g0 = 0; g0 < a.n_slices; g0 += 8) {
                        unsigned long long v[4];
#pragma unroll
                        for (int i = 0; i < 4; ++i) {
                            const int sl = g0 + part * 4 + i;
                            v[i] = 0ull;  // slices beyond n_slices never set the max
                            if (real && sl < a.n_slices)
                                v[i] = __hip_atomic_load(a.gk2 + ((int64_t)b * a.n_slices + sl) * kGk2Keys, __ATOMIC_RELAXED,
                                                         __HIP_MEMORY_SCOPE_AGENT);
                        }
                        unsigned long long m = v[0] > v[1] ? v[0] : v[1];
                        const unsigned long long m2 = v[2] > v[3] ? v[2] : v[3];
                        m = m > m2 ? m : m2;
                        const unsigned long long o = __shfl_xor(m, 32);
                        m = o > m ? o : m;
                        if (m != ~0ull && m + 1ull < bound) bound = m + 1ull;
                    }
                }
                const uint32_t gkl_ad = lds.gkl + 8u * (uint32_t)q;
                if (real && part == 0 && bound < ldsv<unsigned long long>(gkl_ad)) {
                    ldsv_st<unsigned long long>(gkl_ad, bound);
                    const uint32_t nb = q8_bound<M>(bound, ldsv<double>(lds.c0 + 8u * (uint32_t)q), ldsv<double>(lds.c1 + 8u * (uint32_t)q));
                    if (nb < q8_ld_bound<M>(lds, q)) q8_st_bound<M>(lds, q, nb);
                }
            };
            // Every scanning wave pushes into its OWN ring of kWaveRing entries (it alone writes the ring and its tail, the
            // consumer alone the head): no reservation, no atomics -- with one shared ring a push was an LDS atomic with return
            // plus a read of the head, two round trips through an LDS queue full of look-ups, during which the pushing wave
            // issued nothing (a quarter of the wave-steps at 1.25M rows x 1024 queries push).  Lane j < 60 serves ring j % 15,
            // entries j / 15 and j / 15 + 4 past the head: up to kPopPerRing entries per ring and batch, no cross-lane traffic.
            const int my_ring = lane % NS, my_i = lane / NS;  // (lanes >= 4 * NS: no ring)
            const bool has_ring = lane < 4 * NS;
            uint32_t head_v = 0;  // consumed entries of my_ring (the same value in the four lanes that serve it)
            uint32_t n_kept = 0, n_offered = 0;
            // Guard (a.guard): the byte filter is made for tables with structure -- a handful of rows per query pass.  On
            // tables without any (independent uniform codes) rows with a few clipped entries pass in their millions and the
            // u16-table kernel is an order of magnitude faster.  The consumer counts what it sees; past a budget that grows
            // with the rows scanned it declares the launch lost: the flag stops every workgroup (their block counters jump
            // past the slice: the scanning waves run out of blocks and meet at the barriers as usual; the rings are emptied
            // unprocessed), and the gated u16 launch queued behind this one redoes the scan.
            bool aborted = false;
            uint32_t n_seen = 0;
            auto give_up = [&]() {
                aborted = true;
                if (lane == 0) lds_add_u32(lds.blk_ctr(), 0x40000000u);
            };
            auto poll_guard = [&]() {
                if (a.guard && a.guard_abort && !aborted &&
                    __hip_atomic_load(a.guard, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0u)
                    give_up();
            };
            unsigned long long pend_o = ~0ull, pend_j = ~0ull;  // bounds not yet published to the other workgroups (lane = slot)
            bool rowq_more = false;  // (row queue) rows of the last pop still have queries to hand to q8_consume: parked in lds.rowq
            unsigned long long t_busy = 0;
            uint32_t n_batches = 0;
            int epoch = 0, epoch_step = a.q8_epoch0;  // the current epoch ends after step `epoch_step` (the last: after step n_steps - 1)
            for (;;) {
                const bool final = TL || epoch_step >= n_steps - 1;
                const uint32_t want = (uint32_t)NS * (uint32_t)(epoch + 1);  // `arrived` is cumulative over the item
                int idle = 0;
                uint32_t next_item = 0u;
                if constexpr (TL) {
                    if (a.item_counter && lane == 0)
                        next_item = gridDim.x + __hip_atomic_fetch_add(a.item_counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
                import_bounds();
                if constexpr (TL) {
                    if (a.item_counter && lane == 0) ldsv_st<uint32_t>(lds.seen + 8, next_item);
                }
                for (;;) {
                    const uint32_t arrived = ldsv<uint32_t>(lds.arrived());  // (read BEFORE the tails: a wave pushes, then arrives)
                    const uint32_t tail_v = has_ring ? ldsv<uint32_t>(lds.tails() + 4u * (uint32_t)my_ring) : 0u;
                    const uint32_t avail = has_ring ? ((tail_v - head_v) & 0xffffu) : 0u;
                    const bool any = __ballot(avail > 0) != 0;
                    // A non-final epoch does not wait for the backlog: the scanning waves stand at the barrier, what is in the
                    // rings is taken in the next epoch (only the last epoch's end needs every candidate in the lists)
                    if (arrived == want && !final) break;
                    if (any && aborted) {  // (the scan is lost: free the rings, nothing is processed)
                        uint32_t av = lane < NS ? avail : 0u;
#pragma unroll
                        for (int o = 32; o > 0; o >>= 1) av += __shfl_xor(av, o);
                        n_seen += av;
                        head_v = tail_v;
                        if (lane < NS) ldsv_st<uint32_t>(lds.heads() + 4u * (uint32_t)lane, head_v);
                        continue;
                    }
                    if (any || (ROWQ && rowq_more)) {
                        const unsigned long long t0 = a.dbg ? __builtin_readcyclecounter() : 0ull;
                        __builtin_amdgcn_s_setprio(3);  // (serial code on a SIMD shared with three or four scanning waves)
                        unsigned long long e[2];
                        bool act[2];
                        if (!(ROWQ && rowq_more)) {
                        // HEAVY load -- some wave has more than kPopPerRing entries waiting (a block full of near rows; the candidate
                        // generator of the re-rank stage, whose lists take 16 keys per slice) -- is taken 128 at a time whatever
                        // rings it sits in: ring r contributes take_r = min(avail_r, what is left of the 128) entries, slot s (two
                        // per lane) finds its ring by comparing with the prefix sums (15 broadcasts).  At 8 per ring and batch a
                        // burst took the consumer five batches, ~20 us -- with the workgroup at its final barrier when it came in
                        // the last steps -- and the candidate generator ran at 0.75 of its rate with one shared ring.
                        const bool heavy = __ballot(lane < NS && avail > (uint32_t)kPopPerRing) != 0;
                        if (heavy) {
                            uint32_t incl = lane < NS ? avail : 0u;  // lanes 0 .. NS-1: ring = lane
#pragma unroll
                            for (int o = 1; o < 16; o <<= 1) {
                                const uint32_t up = (uint32_t)__shfl_up((int)incl, o);
                                if (lane >= o) incl += up;
                            }
                            const uint32_t excl = incl - (lane < NS ? avail : 0u);
                            const uint32_t take = excl >= 128u ? 0u : (avail < 128u - excl ? avail : 128u - excl);
                            e[0] = e[1] = 0ull;
                            act[0] = act[1] = false;
                            uint32_t my_take = 0;
#pragma unroll 1
                            for (int r = 0; r < NS; ++r) {
                                const uint32_t ex_r = (uint32_t)__builtin_amdgcn_readlane((int)excl, r);
                                const uint32_t tk_r = (uint32_t)__builtin_amdgcn_readlane((int)take, r);
                                const uint32_t hd_r = (uint32_t)__builtin_amdgcn_readlane((int)head_v, r);
                                if (tk_r == 0u) continue;
#pragma unroll
                                for (int u = 0; u < 2; ++u) {
                                    const uint32_t sl = (uint32_t)lane + 64u * (uint32_t)u;
                                    if (sl >= ex_r && sl < ex_r + tk_r) {
                                        act[u] = true;
                                        const uint32_t ad = lds.ring + RE * ((uint32_t)r * kWaveRing + ((hd_r + sl - ex_r) & (kWaveRing - 1)));
                                        e[u] = RE == 4u ? (unsigned long long)ldsv<uint32_t>(ad) : ldsv<unsigned long long>(ad);
                                    }
                                }
                                if (has_ring && my_ring == r) my_take = tk_r;
                            }
                            head_v = (head_v + my_take) & 0xffffu;
                        } else {
#pragma unroll
                            for (int u = 0; u < 2; ++u) {
                                const uint32_t i = (uint32_t)(my_i + 4 * u);
                                act[u] = i < avail;
                                e[u] = 0ull;
                                if (act[u]) {
                                    const uint32_t ad = lds.ring + RE * ((uint32_t)my_ring * kWaveRing + ((head_v + i) & (kWaveRing - 1)));
                                    e[u] = RE == 4u ? (unsigned long long)ldsv<uint32_t>(ad) : ldsv<unsigned long long>(ad);
                                }
                            }
                            head_v = (head_v + avail) & 0xffffu;  // (avail <= kPopPerRing everywhere)
                        }
                        n_seen += (uint32_t)(__popcll(__ballot(act[0])) + __popcll(__ballot(act[1])));
                        if (lane < NS) ldsv_st<uint32_t>(lds.heads() + 4u * (uint32_t)lane, head_v);  // the wave may reuse the entries
                        }
                        if constexpr (ROWQ) {
                            // the popped entries are ROWS: which of the 32 queries pass them NOW?  (out of line.)  Then ROUNDS of
                            // (slot, row) entries, one per popped row and round -- a row passes for one query as a rule: one round.
                            // Masks and row ids are parked in LDS and every round is one pass of this loop (a loop of rounds around
                            // the inlined q8_consume cost the scanning branch a loop-invariant register: a scratch reload per step)
                            const uint32_t park = lds.rowq + 48u * (uint32_t)lane;
                            if (!rowq_more) {
                                const uint32_t rid0 = (uint32_t)e[0], rid1 = (uint32_t)e[1];
                                const unsigned long long pmm = q8_rows_pass_masks<SKEWED>((const uint8_t *)a.codes, rid0, rid1,
                                                                                          (act[0] ? 1u : 0u) | (act[1] ? 2u : 0u), lds.tab, lds.shq, lds.rowq);
                                ldsv_st<u32x4>(park, (u32x4){(uint32_t)pmm, (uint32_t)(pmm >> 32), rid0, rid1});
                            }
                            const u32x4 pk = ldsv<u32x4>(park);
                            uint32_t pmr[2] = {pk.x, pk.y};
#pragma unroll
                            for (int u = 0; u < 2; ++u) {
                                act[u] = pmr[u] != 0u;
                                const uint32_t q = act[u] ? (uint32_t)__builtin_ctz(pmr[u]) : 0u;
                                pmr[u] &= pmr[u] - 1u;
                                e[u] = ((unsigned long long)q << 32) | (u ? pk.w : pk.z);
                            }
                            ldsv_st<u32x2>(park, (u32x2){pmr[0], pmr[1]});
                            rowq_more = __ballot((pmr[0] | pmr[1]) != 0u) != 0;
                        }
                        q8_consume<M, SKEWED, QT, CB, ROWQ, LK, TL>(fc, lds, e, act, lane, n_kept, n_offered, pend_o, pend_j);
                        __builtin_amdgcn_s_setprio(0);
                        ++n_batches;
                        if (a.dbg) t_busy += __builtin_readcyclecounter() - t0;
                        idle = 0;
                        if ((n_batches & (uint32_t)a.q8_import_mask) == 0) {
                            import_bounds();  // (the other slices' progress)
                            poll_guard();
                        }
                        // budget: guard_base (1024) candidates + one per 16 rows the workgroup has drawn.  Measured per 32-query tile: data
                        // with structure ~1 per 1000 rows; uniform vectors through a trained codec ~1 per 31 (the byte tables are still
                        // 2x the u16 tables there); independent random codes ~1 per 4 (10x slower)
                        if (a.guard && a.guard_abort && !aborted && n_seen > a.guard_base + (ldsv<uint32_t>(lds.blk_ctr()) << 2)) {
                            if (lane == 0) __hip_atomic_store(a.guard, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            give_up();
                        }
                        continue;
                    }
                    if (arrived == want) break;
                    if ((++idle & 15) == 8) {
                        import_bounds();
                        poll_guard();
                    }
                    __builtin_amdgcn_s_sleep(4);
                }
                q8_publish_global<QT, LK, TL>(fc, lds, lane, pend_o, pend_j);
                epoch_sync(final);
                if (final) break;
                if (ldsv<uint32_t>(lds.ctl)) {
                    // the table was rebuilt: the integer sums of the waiting candidates are in the OLD table's steps --
                    // clear them, so that the stale-candidate check lets them through to the exact sum (every scanning wave has
                    // arrived: its pushes are complete)
                    if (has_ring && RE == 8u) {  // (row ids carry no sum)
                        const uint32_t tail_v = ldsv<uint32_t>(lds.tails() + 4u * (uint32_t)my_ring);
                        for (uint32_t i = (uint32_t)my_i; i < ((tail_v - head_v) & 0xffffu); i += 4u) {
                            const uint32_t ad = lds.ring + 8u * ((uint32_t)my_ring * kWaveRing + ((head_v + i) & (kWaveRing - 1)));
                            ldsv_st<unsigned long long>(ad, ldsv<unsigned long long>(ad) & ~(0xffffull << 40));
                        }
                    }
                }
                ++epoch;
                epoch_step = a.q8_epoch_mul * epoch_step + (a.q8_epoch_mul - 1);
            }
            if (a.guard && lane == 0) ldsv_st<uint32_t>(lds.seen, ldsv<uint32_t>(lds.seen) + n_seen);
            if (a.dbg && lane == 0 && !(a.dbg_skip & 8)) {  // ANNLITE_DEBUG_COUNTERS=1: [2] exact sums, [3] candidates that went into a list's queue,
                atomicAdd(a.dbg + 2, (unsigned long long)n_kept);     // [4] consumer cycles inside batches, [6] batches
                atomicAdd(a.dbg + 3, (unsigned long long)n_offered);
                atomicAdd(a.dbg + 4, t_busy);
                atomicAdd(a.dbg + 6, (unsigned long long)n_batches);
            }
        } else {
            // ------------------------------------------------------------------------------- scanning waves
            constexpr int NF = WIDE ? 4 : 4 * NQ;  // filter words per row: 4 dwords of byte sums per entry group (16 queries) / 4 dwords of u16 sums (WIDE: 8)
            // The lane constants of this branch (skew, look-up columns, rotation masks) are computed from OPAQUE copies of the lane
            // number: the compiler otherwise hoists them above the consumer / scanning branch, where the consumer's register needs
            // decide which of them are spilled -- and reloaded from scratch in every step (scripts/check_q8_isa_all.sh)
            int lane_pin = lane;
            asm volatile("" : "+v"(lane_pin));
            const int s = lane_pin % (WIDE ? 32 : M);
            const int rot_bytes = CB * s;  // PLAIN rows are rotated in registers: element (s + t) mod M to position t
            const uint32_t bsh = (uint32_t)(rot_bytes & 3);
            bool abit[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) abit[i] = (((rot_bytes >> 2) >> i) & 1) != 0;
            const unsigned long long rmask0 = __ballot(abit[0]), rmask1 = __ballot(abit[1]);  // (C16: see the step loop)
            // LDS byte addresses as integers
            typedef const ANNLITE_LDS u32x4 *lds_entry_ptr;
            typedef const ANNLITE_LDS u32x2 *lds_entry8_ptr;
            const uint32_t lds0 = lds.tab;
            // M = 16 (q8_entry16, round 6): the table is two half tables by sub-space PARITY at LDS address 0; look-up t of this lane reads
            // sub-space (s + t) mod 16: its address is ONE v_perm_b32 of the code dword with mbase[t] (byte 0: the slot of the first entry
            // group, byte 2: the half), the second entry group sits 16 bytes behind (immediate)
            constexpr bool P16 = M == 16 && NQ == 2 && CB == 1;
            uint32_t mbase[(WIDE || M8 || M32) ? 1 : M];
            if constexpr (!WIDE && !M8 && !M32) {
#pragma unroll
                for (int t = 0; t < M; ++t) mbase[t] = P16 ? q8_entry16(0u, (uint32_t)((s + t) % M), 0u) : lds0 + (uint32_t)(((s + t) % M) * EB);
                if constexpr (P16) {
                    if (lds0 != 0u) __builtin_trap();
                }
            }
            // M8 (M = 8; uint16 codes, or uint8 ones up to Ks = 256): a code row of the table is 256 bytes = [2 entry groups][8 sub-spaces][16 B], the address of
            // look-up t is (code << 8) | column byte -- ONE v_perm_b32 of the code dword with a lane constant (kx / ky: byte t of
            // the pair = the column of look-up t in the lane's FIRST / SECOND entry group).  Eight sub-spaces cover only half of
            // the 16 bank slots a ds_read_b128 lane group spans, and every hardware lane group holds each s = lane % 8 twice (lanes
            // l and l ^ 24 resp. l ^ 8 ... : they differ in lane bit 4): the lanes with bit 4 set read the entry groups in the
            // OTHER order -- 16 distinct slots per lane group, conflict-free.  Their sums[0..3] then belong to queries 16..31:
            // they load the filter words swapped (load_thw) and the candidate path un-swaps the slot.
            const uint32_t f16 = (M8 && NQ == 2) ? ((uint32_t)lane_pin >> 4) & 1u : 0u;
            uint32_t kx[2] = {0u, 0u}, ky[2] = {0u, 0u};
            if constexpr (M8) {
#pragma unroll
                for (int t = 0; t < 8; ++t) {
                    const uint32_t col = (uint32_t)(((s + t) % 8) * 16) + f16 * 128u;
                    kx[t / 4] |= col << (8 * (t % 4));
                    ky[t / 4] |= (col ^ 128u) << (8 * (t % 4));
                }
                if ((NQ == 2 && lds0 != 0u) || a.Ks > (C16 ? 1024 / NQ : 256)) __builtin_trap();
            }
            // C16 with ONE entry group (512 < Ks <= 1024): a code row of the table is 128 bytes = [8 sub-spaces][16 B] -- two code
            // rows per bank line, so two lanes of a lane group with the same sub-space collide whenever their codes have the
            // same parity (2-way conflicts, inherent: 160 KB do not hold a 256-byte row per code); address = (code << 7) + column
            uint32_t mcol[(M8 && NQ == 1) ? M : 1];
            if constexpr (M8 && NQ == 1) {
#pragma unroll
                for (int t = 0; t < M; ++t) mcol[t] = lds0 + (uint32_t)(((s + t) % M) * EB);
            }
            // M32: the table is two half tables of 16 sub-spaces, [256][16][16 B] each, 64 KB apart, at LDS address 0: the address of
            // look-up t is (half << 16) | (code << 8) | column -- ONE v_perm_b32 of the code dword with a lane constant: k32[t / 2]
            // holds (column, half) of look-ups t = 2 j and 2 j + 1 in its bytes (0, 1) and (2, 3).  A ds_read_b128 lane group (16
            // consecutive lanes) reads 16 consecutive sub-spaces mod 32: 16 distinct columns, conflict-free whatever the codes
            uint32_t k32[M32 ? 16 : 1];
            if constexpr (M32) {
#pragma unroll
                for (int j = 0; j < 16; ++j) {
                    const uint32_t s0 = (uint32_t)((s + 2 * j) % 32), s1 = (uint32_t)((s + 2 * j + 1) % 32);
                    k32[j] = ((s0 & 15u) << 4) | ((s0 >> 4) << 8) | ((s1 & 15u) << 20) | ((s1 >> 4) << 24);
                }
                if (lds0 != 0u || a.Ks > 256) __builtin_trap();
            }
            // WIDE: lane constant of the look-up addresses (the M = 64 u16 kernel's scheme): byte 0 = (lane % 32) * 8 (the column),
            // byte 2 = 0x01 (second half table); the table starts at LDS address 0 (all LDS is dynamic)
            const uint32_t lane_k = 0x00010000u | (uint32_t)((lane_pin & 31) * 8);
            if constexpr (WIDE) {
                if (lds0 != 0u || a.Ks != 256) __builtin_trap();
            }
            const uint32_t *codes32 = (const uint32_t *)a.codes;
            // WIDE: the table's base sits in a VGPR pair (made opaque to the compiler: as an SGPR pair it was spilled and
            // re-read from the kernarg segment in every step -- an s_load whose s_waitcnt lgkmcnt(0) also waited for the
            // block-counter atomic issued just before it)
            unsigned long long codes_v = (unsigned long long)(uintptr_t)a.codes, valid_v = (unsigned long long)(uintptr_t)a.valid;
            if constexpr (WIDE) asm volatile("" : "+v"(codes_v), "+v"(valid_v));
            // rows are < 2^32 per call (plan): 32-bit row arithmetic keeps the loop control in SGPRs
            const uint32_t n_rows = (uint32_t)a.N;
            const uint32_t n_blocks = item_blocks;
            // block b of the work item starts at row blk_off + ((b >> blk_log) * blk_per + (b & blk_msk)) * 64: contiguous range
            // (blk_log = 26: b >> 26 = 0 for every table the plan admits) or interleaved runs (ScanArgs::q8_ilv_log); scalar arithmetic
            const bool ilv = a.q8_ilv_log > 0;
            const uint32_t blk_log = ilv ? (uint32_t)a.q8_ilv_log : 26u, blk_msk = (1u << blk_log) - 1u;
            const uint32_t blk_per = ilv ? (uint32_t)a.n_slices << blk_log : 0u;
            const uint32_t blk_off = ilv ? ((uint32_t)slice << blk_log) << 6 : (uint32_t)slice_begin;
            const uint32_t s_end = ilv ? n_rows : (uint32_t)slice_end;  // (the rows behind the item's last block)
            auto blk_row = [&](uint32_t b) -> uint32_t { return blk_off + (((b >> blk_log) * blk_per + (b & blk_msk)) << 6); };
            // The waves DRAW their blocks of 64 rows from a counter in LDS.  (The SIMD's arbiter favours its oldest wave, and
            // one wave alone issues at about a third of the rate four reach together -- scripts/ubench/valu_cost.hip,
            // step_loop.hip.  With the rows dealt out statically the favoured waves finished their share of an epoch early
            // and the last ones ran it out alone: wave 0 sat at the epoch barriers for 45 % of the kernel.)
            auto draw = [&]() -> uint32_t {  // (lane 0's value; broadcast a step later, where it is first needed)
                uint32_t v = 0;
                if (lane == 0) v = lds_add_u32(lds.blk_ctr(), 1u);
                return v;
            };
            uint32_t ccur[CW], cnext[CW];
            uint32_t addr[(WIDE || M8 || M32) ? 1 : M];
            auto load_row = [&](uint32_t row, uint32_t (&c)[CW]) {
                if constexpr (ANNLITE_Q8_EXP == 3) {  // (timing experiment: no code rows from memory)
#pragma unroll
                    for (int i = 0; i < CW; ++i) c[i] = row * 0x9E3779B1u + (uint32_t)i * 0x85EBCA77u;
                    return;
                }
                if (row >= n_rows) row = n_rows - 1;
                typedef const uint32_t __attribute__((address_space(1))) *gptr_t;  // (global, not flat: a flat load also counts in lgkmcnt)
                const gptr_t p = (WIDE ? (gptr_t)codes_v : (gptr_t)(uintptr_t)codes32) + (int64_t)row * CW;
                if constexpr (CW == 2) {
                    const u32x2 v = *(const u32x2 __attribute__((address_space(1))) *)p;
                    c[0] = v.x;
                    c[1] = v.y;
                } else {
#pragma unroll
                    for (int i = 0; i < CW / 4; ++i) {
                        const u32x4 v = *(const u32x4 __attribute__((address_space(1))) *)(p + 4 * i);
                        c[4 * i + 0] = v.x;
                        c[4 * i + 1] = v.y;
                        c[4 * i + 2] = v.z;
                        c[4 * i + 3] = v.w;
                    }
                }
            };
            auto make_addr = [&](const uint32_t (&cc)[CW]) {
                if constexpr (P16)
                    static_for<0, M>([&](auto T) {
                        constexpr int t = decltype(T)::value;
                        // byte 0 <- mbase byte 0 (slot), byte 1 <- code byte t % 4, byte 2 <- mbase byte 2 (half), byte 3 <- 0
                        addr[t] = __builtin_amdgcn_perm(cc[t / 4], mbase[t], 0x0c020000u | ((4u + (uint32_t)(t % 4)) << 8));
                    });
                else if constexpr (!WIDE && !M8 && !M32)
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
            // the slots' bounds as the filter words see them: (0x80 | T) bytes of 32 queries / (0x8000 | T) half-words of 8
            auto load_thw = [&](uint32_t (&t)[NF]) {
                if constexpr (WIDE) {
                    const u32x4 v = *(volatile ANNLITE_LDS u32x4 *)(uintptr_t)lds.shq;
                    t[0] = v.x, t[1] = v.y, t[2] = v.z, t[3] = v.w;
                } else {
#pragma unroll
                    for (int h = 0; h < NQ; ++h) {
                        const u32x4 v = *(volatile ANNLITE_LDS u32x4 *)(uintptr_t)(lds.shq + 16u * ((uint32_t)h ^ f16));
                        t[4 * h + 0] = v.x, t[4 * h + 1] = v.y, t[4 * h + 2] = v.z, t[4 * h + 3] = v.w;
                    }
                }
            };
            uint32_t thw[NF];
            load_thw(thw);
            // byte sums of the row for both entry groups (4 dwords x 4 x u8 each): the 2 M look-ups run through a ring of
            // DEPTH landing registers -- look-up i + DEPTH is issued as soon as look-up i has been added (all M look-ups
            // of a group in flight, as the u16 kernel has them, takes 64 landing VGPRs: with them the allocator spilled
            // six of the 16 loop-invariant LDS base registers into the step loop)
            auto row_sums = [&](const uint32_t (&cc)[CW], uint32_t (&sums)[NF]) {
                if constexpr (WIDE) {
                    // M = 64: four chunks of 16 look-ups (ds_read_b64: 8 byte entries, ONE v_perm per address -- the wrap-coded
                    // SKEWED layout), their byte sums (<= 16 * 15 = 240) widened into the row's u16 sums: dword 0 = queries
                    // (0, 2), 1 = (1, 3), 2 = (4, 6), 3 = (5, 7)
                    sums[0] = sums[1] = sums[2] = sums[3] = 0u;
                    // The 64 look-ups run through ONE ring of WDEPTH landing registers, pairs at a time (a pair = one
                    // v_add3 per dword): look-ups i + WDEPTH, i + WDEPTH + 1 are issued as soon as pair i has been added, so
                    // the wave's LDS queue never drains at a chunk boundary (four separate chunks of 16, each waited out
                    // before the next was issued, ran at 0.47 of the look-up roof; the step is VALU/LDS co-bound at 0.89)
                    constexpr int WDEPTH = ANNLITE_Q8_WDEPTH;
                    static_assert(WDEPTH % 2 == 0 && WDEPTH >= 2 && WDEPTH <= 64, "pairs");
                    u32x2 v[WDEPTH];
                    auto fetch = [&](u32x2 &dst, auto T) {
                        constexpr int t = decltype(T)::value;
                        constexpr int half = t / 32;
                        // byte 0 <- lane_k byte 0, byte 1 <- code byte t % 4, byte 2 <- lane_k byte 2 (second half) or 0, byte 3 <- 0
                        constexpr uint32_t sel = 0x0c000000u | ((half ? 0x02u : 0x0cu) << 16) | ((4u + (uint32_t)(t % 4)) << 8);
                        const uint32_t ad = __builtin_amdgcn_perm(cc[t / 4], lane_k, sel);
                        dst = *(lds_entry8_ptr)(uintptr_t)(ad + (uint32_t)((t % 32) * 8 + half * 0x100));
                    };
                    static_for<0, WDEPTH>([&](auto I) { fetch(v[decltype(I)::value], I); });
                    u32x2 bs = {0u, 0u};
                    static_for<0, 32>([&](auto P) {
                        constexpr int i = 2 * decltype(P)::value;
                        asm volatile("" ::: "memory");
                        if constexpr (i % 16 == 0) bs = v[i % WDEPTH] + v[(i + 1) % WDEPTH];
                        else bs += v[i % WDEPTH] + v[(i + 1) % WDEPTH];
                        if constexpr (i + WDEPTH < 64) {
                            fetch(v[i % WDEPTH], std::integral_constant<int, i + WDEPTH>{});
                            fetch(v[(i + 1) % WDEPTH], std::integral_constant<int, i + WDEPTH + 1>{});
                        }
                        if constexpr (i % 16 == 14) {
                            sums[0] += bs.x & 0x00ff00ffu;
                            sums[1] += __builtin_amdgcn_perm(0u, bs.x, 0x0c030c01u);  // (bytes 1, 3 -> half-words)
                            sums[2] += bs.y & 0x00ff00ffu;
                            sums[3] += __builtin_amdgcn_perm(0u, bs.y, 0x0c030c01u);
                        }
                    });
                } else if constexpr (M8) {
                    // 8 look-ups in the lane's first entry group, then 8 in its second, through a ring of DEPTH landing registers
                    constexpr int DEPTH = ANNLITE_Q8_DEPTH, TOT = NQ * M;
                    u32x4 acc[NQ];
                    u32x4 v[DEPTH];
                    auto fetch = [&](u32x4 &dst, auto I) {
                        constexpr int i = decltype(I)::value, t = i % M, g = i / M;
                        // byte 0 <- column byte t % 4 of the lane constant, bytes 1..2 <- the 16-bit code, byte 3 <- 0
                        // (uint8 codes: byte 1 <- code byte t % 4, byte 2 <- 0)
                        constexpr uint32_t sel = C16 ? 0x0c000000u | ((uint32_t)(4 + 2 * (t % 2) + 1) << 16) | ((uint32_t)(4 + 2 * (t % 2)) << 8) | (uint32_t)(t % 4)
                                                     : 0x0c0c0000u | ((uint32_t)(4 + t % 4) << 8) | (uint32_t)(t % 4);
                        uint32_t ad;
                        if constexpr (NQ == 2) ad = __builtin_amdgcn_perm(cc[C16 ? t / 2 : t / 4], g ? ky[t / 4] : kx[t / 4], sel);
                        else ad = (((t % 2) ? cc[t / 2] >> 16 : cc[t / 2] & 0xffffu) << 7) + mcol[t];
                        dst = *(lds_entry_ptr)(uintptr_t)ad;
                    };
                    static_for<0, DEPTH>([&](auto I) { fetch(v[decltype(I)::value], I); });
                    static_for<0, TOT>([&](auto I) {
                        constexpr int i = decltype(I)::value;
                        asm volatile("" ::: "memory");
                        if constexpr (i % M == 0) acc[i / M] = v[i % DEPTH];
                        else acc[i / M] += v[i % DEPTH];
                        if constexpr (i + DEPTH < TOT) fetch(v[i % DEPTH], std::integral_constant<int, i + DEPTH>{});
                    });
#pragma unroll
                    for (int h = 0; h < NQ; ++h) sums[4 * h + 0] = acc[h].x, sums[4 * h + 1] = acc[h].y, sums[4 * h + 2] = acc[h].z, sums[4 * h + 3] = acc[h].w;
                } else if constexpr (M32) {
                    // 32 look-ups (one entry group: the byte sums of 32 entries clipped at 7 never carry) through a ring of DEPTH
                    // landing registers, one v_perm_b32 per address
                    constexpr int DEPTH = ANNLITE_Q8_DEPTH;
                    u32x4 acc;
                    u32x4 v[DEPTH];
                    auto fetch = [&](u32x4 &dst, auto I) {
                        constexpr int t = decltype(I)::value, b = t % 2;
                        // byte 0 <- the column (k32 byte 2 b), byte 1 <- code byte t % 4, byte 2 <- the half (k32 byte 2 b + 1), byte 3 <- 0
                        constexpr uint32_t sel = 0x0c000000u | ((uint32_t)(2 * b + 1) << 16) | ((uint32_t)(4 + t % 4) << 8) | (uint32_t)(2 * b);
                        const uint32_t ad = __builtin_amdgcn_perm(cc[t / 4], k32[t / 2], sel);
                        dst = *(lds_entry_ptr)(uintptr_t)ad;
                    };
                    static_for<0, DEPTH>([&](auto I) { fetch(v[decltype(I)::value], I); });
                    static_for<0, M>([&](auto I) {
                        constexpr int i = decltype(I)::value;
                        asm volatile("" ::: "memory");
                        if constexpr (i == 0) acc = v[0];
                        else acc += v[i % DEPTH];
                        if constexpr (i + DEPTH < M) fetch(v[i % DEPTH], std::integral_constant<int, i + DEPTH>{});
                    });
                    sums[0] = acc.x, sums[1] = acc.y, sums[2] = acc.z, sums[3] = acc.w;
                } else {
                    constexpr int DEPTH = ANNLITE_Q8_DEPTH, TOT = NQ * M;
                    constexpr int GOFF = P16 ? 16 : RB;  // distance of a look-up's second entry group
                    u32x4 acc[NQ];
                    u32x4 v[DEPTH];
                    auto fetch = [&](u32x4 &dst, uint32_t ad) {
                        if constexpr (ANNLITE_Q8_EXP == 2) asm volatile("" : "=v"(dst) : "v"(ad));
                        else dst = *(lds_entry_ptr)(uintptr_t)ad;
                    };
                    static_for<0, DEPTH>([&](auto I) {
                        constexpr int i = decltype(I)::value;
                        fetch(v[i], addr[i % M] + (uint32_t)((i / M) * GOFF));
                    });
                    static_for<0, TOT>([&](auto I) {
                        constexpr int i = decltype(I)::value;
                        asm volatile("" ::: "memory");
                        if constexpr (ANNLITE_Q8_EXP == 1) {
                            asm volatile("" ::"v"(v[i % DEPTH]));
                            if constexpr (i % M == 0) acc[i / M] = (u32x4){thw[4 * (i / M)], thw[4 * (i / M) + 1], thw[4 * (i / M) + 2], thw[4 * (i / M) + 3]};
                        } else {
                            if constexpr (i % M == 0) acc[i / M] = v[i % DEPTH];
                            else acc[i / M] += v[i % DEPTH];
                        }
                        if constexpr (i + DEPTH < TOT) {
                            constexpr int j = i + DEPTH;
                            fetch(v[i % DEPTH], addr[j % M] + (uint32_t)((j / M) * GOFF));
                        }
                    });
#pragma unroll
                    for (int h = 0; h < NQ; ++h) sums[4 * h + 0] = acc[h].x, sums[4 * h + 1] = acc[h].y, sums[4 * h + 2] = acc[h].z, sums[4 * h + 3] = acc[h].w;
                }
            };
            // bit per passing (query) field of filter word i: S <= T  <=>  (flag | T) - S keeps the flag bit (no borrow crosses a
            // field: S <= 127 resp. <= 960 < 0x8000)
            auto hits = [&](uint32_t th, uint32_t sm) -> uint32_t {
                if constexpr (WIDE) return (th - sm) & 0x80008000u;
                else return (th - (sm & 0x7f7f7f7fu)) & ~sm & 0x80808080u;
            };
            uint32_t vcur = ~0u, vnext = ~0u;
            const uint32_t *valid = a.valid;
            auto load_valid = [&](uint32_t row) -> uint32_t {
                if (!valid) return ~0u;
                if (row >= n_rows) row = n_rows - 1;
                if constexpr (WIDE) return ((const uint32_t __attribute__((address_space(1))) *)valid_v)[row >> 5];
                else return valid[row >> 5];
            };
            // The code bytes (and validity word) of a lane's row are fetched ONE STEP AHEAD: issued at the top of a step for
            // the next block, picked up at the top of that one; the number of the block after that is drawn in between.
            // (Fetching two steps ahead and rotating the registers at the end of the step made the compiler wait for the
            // load it had just issued: s_waitcnt vmcnt(0) every step.)
            uint32_t b_cur = (uint32_t)__builtin_amdgcn_readfirstlane((int)draw());
            uint32_t pend = draw();
            load_row(blk_row(b_cur) + lane, cnext);
            vnext = load_valid(blk_row(b_cur) + lane);
            uint32_t b_nxt = (uint32_t)__builtin_amdgcn_readfirstlane((int)pend);
            // The blocks are cut into epochs that end after block 15 * 16, 15 * 256, ... (q8_epoch0, q8_epoch_mul) and after the last one (the
            // epochs' barriers meet).  The step loop of an epoch contains no call and no barrier: the loop-invariant
            // registers stay put.
            uint32_t it_no = 0;
            const uint32_t thw_mask = (uint32_t)a.q8_thw_mask;
            uint32_t rs = 0;  // (pushed entries) | (consumed entries, as last read) << 16 of this wave's ring, both mod 2^16
            uint32_t n_slow = 0, n_push = 0;
            unsigned long long t_wait = 0;
            for (int epoch_step = a.q8_epoch0;; epoch_step = a.q8_epoch_mul * epoch_step + (a.q8_epoch_mul - 1)) {
                const bool final = TL || epoch_step >= n_steps - 1;
                const uint32_t end_blk = final ? n_blocks : (uint32_t)NS * (uint32_t)(epoch_step + 1);
                for (; b_cur < end_blk; ++it_no) {
                    const uint32_t row0 = blk_row(b_cur);  // (< s_end: b_cur < n_blocks)
                    pend = draw();
                    // PLAIN 16-byte rows (M = 16 / uint8, M = 8 / uint16): first rotation stage straight from the landing registers -- no copy
                    constexpr bool ROT4 = !SKEWED && !WIDE && CW == 4;
                    uint32_t rot1[ROT4 ? CW : 1];
                    if constexpr (ROT4) {
#pragma unroll
                        for (int i = 0; i < CW; ++i)
                            asm("v_cndmask_b32 %0, %1, %2, %3" : "=v"(rot1[i]) : "v"(cnext[i]), "v"(cnext[(i + 1) % CW]), "s"(rmask0));
                    } else {
#pragma unroll
                        for (int i = 0; i < CW; ++i) ccur[i] = cnext[i];
                    }
                    vcur = vnext;
                    {
                        const uint32_t row1 = blk_row(b_nxt) + lane;  // (past the slice at its end: clamped, unused)
                        load_row(row1, cnext);
                        vnext = load_valid(row1);
                    }
                    if constexpr (!SKEWED) {
                        if constexpr (WIDE) skew64_encode(ccur, lane & 31);  // PLAIN row -> this lane's wrap-coded SKEWED row
                        else if constexpr (ROT4) {
                            // rotate_row with the two stage conditions as LANE MASKS in SGPR pairs (as `bool`s the allocator kept
                            // them as 0/1 VGPRs and re-compared before every select: 14 v_cmp + 15 s_nop per step)
                            uint32_t n[CW];
#pragma unroll
                            for (int i = 0; i < CW; ++i)
                                asm("v_cndmask_b32 %0, %1, %2, %3" : "=v"(n[i]) : "v"(rot1[i]), "v"(rot1[(i + 2) % CW]), "s"(rmask1));
#pragma unroll
                            for (int i = 0; i < CW; ++i) ccur[i] = __builtin_amdgcn_alignbyte(n[(i + 1) % CW], n[i], bsh);
                        } else rotate_row<CW>(ccur, abit, bsh);
                    }
                    unsigned long long vmask = ~0ull;
                    if (s_end - row0 < 64u) vmask = (1ull << (s_end - row0)) - 1ull;
                    // validity word of this lane's row, fetched one step ahead with the code bytes
                    if (valid) vmask &= __ballot((vcur >> (lane & 31)) & 1u);
                    make_addr(ccur);
                    uint32_t sums[NF];
                    row_sums(ccur, sums);
                    // any (query, lane) with S <= T ?
                    uint32_t anyv = 0;
                    if constexpr (WIDE) {
#pragma unroll
                        for (int w = 0; w < NF; ++w) anyv |= thw[w] - sums[w];
                        anyv &= 0x80008000u;
                    } else {
                        // S < 128 and (0x80 | T) - (S & 0x7f) has bit 7 set (T <= 127: no borrow)
#pragma unroll
                        for (int w = 0; w < NF; ++w) anyv |= (thw[w] - (sums[w] & 0x7f7f7f7fu)) & ~sums[w];
                        anyv &= 0x80808080u;
                    }
                    unsigned long long rem = __ballot(anyv != 0) & vmask;
                    if constexpr (ROWQ) {
                        if (rem && !(a.dbg_skip & 4)) {
                            // push the ROW ids of the hit lanes (the consumer finds the queries: q8_row_pass_mask) -- one ring
                            // reservation, lane-parallel stores
                            ++n_slow;
                            const uint32_t n = (uint32_t)__popcll(rem);
                            uint32_t tl = rs & 0xffffu, hd = rs >> 16;
                            while (((tl + n - hd) & 0xffffu) > (uint32_t)kWaveRing) {  // the consumer is behind
                                hd = ldsv<uint32_t>(lds.heads() + 4u * (uint32_t)wave) & 0xffffu;
                                if (((tl + n - hd) & 0xffffu) <= (uint32_t)kWaveRing) break;
                                __builtin_amdgcn_s_sleep(8);
                            }
                            const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(rem >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)rem, 0u));
                            if (__builtin_amdgcn_inverse_ballot_w64(rem))  // (the hit lanes: exec <- rem)
                                ldsv_st<uint32_t>(lds.ring + RE * ((uint32_t)wave * kWaveRing + ((tl + rank) & (kWaveRing - 1))),
                                                  row0 + (uint32_t)lane);  // (the low word of the entry; the consumer reads nothing else of it)
                            tl = (tl + n) & 0xffffu;
                            if (lane == 0) ldsv_st<uint32_t>(lds.tails() + 4u * (uint32_t)wave, tl);
                            rs = tl | (hd << 16);
                            n_push += n;
                        }
                    } else
                    if (rem && !(a.dbg_skip & 4)) {
                        ++n_slow;
                        // The step's candidates -- (lane, query) pairs with S <= T -- are few (one or two lanes of a step that has
                        // any), so they are enumerated in SCALAR code: a hit lane's byte sums are read into SGPRs (v_readlane), the
                        // filter is redone there, the entries (S << 40 | slot << 32 | row) are staged lane by lane and
                        // pushed with ONE ring reservation.  (Per dword and byte with ballots and one reservation per hit byte --
                        // 32 unrolled copies, ~19 KB of code -- this path cost ~110 VALU instructions per step with a candidate, on
                        // top of the ~120 of the step itself: a quarter of the wave-steps at 1.25M rows x 1024 queries take it.)
                        uint32_t ts[NF];  // (C16: the filter words differ between the two lane halves -- read from the hit lane below)
#pragma unroll
                        for (int i = 0; i < NF; ++i) ts[i] = (uint32_t)__builtin_amdgcn_readfirstlane((int)thw[i]);
                        uint32_t e_lo = 0, e_hi = 0;  // lane j: staged entry j
                        int n = 0;
                        for (;;) {
                            if (n > 32 || (n > 0 && !rem)) {  // (a lane adds at most 32 entries)
                                // into the wave's own ring: entries [tail, tail + n), then the new tail (LDS executes a wave's
                                // instructions in order).  The head is re-read only when the cached one says the ring is full.
                                uint32_t tl = rs & 0xffffu, hd = rs >> 16;
                                while (((tl + (uint32_t)n - hd) & 0xffffu) > (uint32_t)kWaveRing) {  // the consumer is behind
                                    hd = ldsv<uint32_t>(lds.heads() + 4u * (uint32_t)wave) & 0xffffu;
                                    if (((tl + (uint32_t)n - hd) & 0xffffu) <= (uint32_t)kWaveRing) break;
                                    __builtin_amdgcn_s_sleep(8);
                                }
                                if (lane < n)
                                    ldsv_st<unsigned long long>(lds.ring + 8u * ((uint32_t)wave * kWaveRing + ((tl + (uint32_t)lane) & (kWaveRing - 1))),
                                                                ((unsigned long long)e_hi << 32) | e_lo);
                                tl = (tl + (uint32_t)n) & 0xffffu;
                                if (lane == 0) ldsv_st<uint32_t>(lds.tails() + 4u * (uint32_t)wave, tl);
                                rs = tl | (hd << 16);
                                n_push += (uint32_t)n;
                                n = 0;
                            }
                            if (!rem) break;
                            const int L = __builtin_ctzll(rem);
                            rem &= rem - 1ull;
                            const uint32_t rid = row0 + (uint32_t)L;
                            const uint32_t fL = M8 ? ((uint32_t)L >> 4) & 1u : 0u;  // that lane's sums[0..3] are of entry group fL
                            if constexpr (ANNLITE_Q8_EXP == 5) {  // (timing experiment: ONE entry per hit row, no enumeration -- results wrong)
                                if (lane == n) {
                                    e_hi = (uint32_t)L & 31u;
                                    e_lo = rid;
                                }
                                ++n;
                                continue;
                            }
                            static_for<0, NF>([&](auto I) {
                                constexpr int i = decltype(I)::value;
                                const uint32_t ss = (uint32_t)__builtin_amdgcn_readlane((int)sums[i], L);
                                if constexpr (M8 && NQ == 2) ts[i] = (uint32_t)__builtin_amdgcn_readlane((int)thw[i], L);
                                uint32_t bits = hits(ts[i], ss);
                                while (bits) {
                                    uint32_t sv, slot;  // (the consumer re-checks the sum)
                                    if constexpr (WIDE) {
                                        const uint32_t hf = (uint32_t)__builtin_ctz(bits) >> 4;  // half-word of dword i
                                        sv = (ss >> (16u * hf)) & 0xffffu;
                                        slot = (uint32_t)((i & 2) << 1) | (hf << 1) | (uint32_t)(i & 1);
                                    } else {
                                        const uint32_t by = (uint32_t)__builtin_ctz(bits) >> 3;
                                        sv = (ss >> (8u * by)) & 0xffu;
                                        slot = (uint32_t)(4 * i) + by;
                                        if constexpr (M8 && NQ == 2) slot ^= fL << 4;
                                    }
                                    bits &= bits - 1u;
                                    if (lane == n) {  // (scalar values into lane n: one compare, two conditional moves)
                                        e_hi = (sv << 8) | slot;
                                        e_lo = rid;
                                    }
                                    ++n;
                                }
                            });
                        }
                    }
                    // pick up the workgroup's bounds every 2nd step (every 8th where a work item scans >= 500k rows: ScanArgs::q8_thw_mask)
                    if ((it_no & thw_mask) == thw_mask) {
                        asm volatile("" ::: "memory");
                        load_thw(thw);
                    }
                    b_cur = b_nxt;
                    b_nxt = (uint32_t)__builtin_amdgcn_readfirstlane((int)pend);
                }
                if (lane == 0) lds_add_u32(lds.arrived(), 1u);
                const unsigned long long tw = a.dbg ? __builtin_readcyclecounter() : 0ull;
                if (final) stamp(2);
                epoch_sync(final);
                if (a.dbg) t_wait += __builtin_readcyclecounter() - tw;
                if (final) stamp(3);
                if (final) break;
                load_thw(thw);
            }
            if (a.dbg && lane == 0 && !(a.dbg_skip & 8)) {  // [0] wave-steps with a candidate, [1] entries pushed, [7] wave 0's cycles at epoch ends
                atomicAdd(a.dbg + 0, (unsigned long long)n_slow);
                atomicAdd(a.dbg + 1, (unsigned long long)n_push);
                if (wave == 0) atomicAdd(a.dbg + 7, t_wait);
            }
        }

        q8_finish_item<M, NW, NQ, LK, TL>(ka, tile, slice);
        if (a.dbg && tid == 0) {
            // [8] 2^62 - earliest start, [9] latest end, sums over the work items: [10] start, [11] init + first table build,
            // [12] thread 0's step loop, [13] its wait at the last barrier (the consumer's backlog, the slower waves),
            // [14] list store + merge, [15] work items
            const unsigned long long t_end = wall_clock64();
            const unsigned long long t_item = ldsv<unsigned long long>(stamp_ad), t_built = ldsv<unsigned long long>(stamp_ad + 8),
                                     t_scanned = ldsv<unsigned long long>(stamp_ad + 16), t_synced = ldsv<unsigned long long>(stamp_ad + 24);
            atomicMax(a.dbg + 8, (1ull << 62) - t_item);
            atomicMax(a.dbg + 9, t_end);
            atomicAdd(a.dbg + 10, t_item);
            atomicAdd(a.dbg + 11, t_built - t_item);
            atomicAdd(a.dbg + 12, t_scanned - t_built);
            atomicAdd(a.dbg + 13, t_synced - t_scanned);
            atomicAdd(a.dbg + 14, t_end - t_synced);
            const unsigned long long rec = atomicAdd(a.dbg + 15, 1ull);
            if (rec < 4096ull) {  // per-item record (annlite_debug_items): tile, slice, the five stamps
                unsigned long long *r = a.dbg + 16 + rec * 8;
                r[0] = (unsigned long long)tile;
                r[1] = (unsigned long long)slice;
                r[2] = t_item;
                r[3] = t_built;
                r[4] = t_scanned;
                r[5] = t_synced;
                r[6] = t_end;
                r[7] = (unsigned long long)blockIdx.x;
            }
        }
    }
    if constexpr (!M8) {
        if (blockIdx.x == 0 && tid == 0) {
            unsigned long long *clk = ka->clk;
            if (clk) clk[2] = __builtin_readcyclecounter(), clk[3] = wall_clock64();
        }
    }
    // ---- guard statistics: candidates seen by the whole launch, written to the caller's host-mapped block by the last
    // workgroup to leave (the library's kernel choice for the next calls: scan.hip) ----------------------------------------
    if (a.guard) {
        __syncthreads();
        if (tid == 0) {
            __hip_atomic_fetch_add((unsigned long long *)(a.guard + 2), (unsigned long long)ldsv<uint32_t>(lds.seen), __ATOMIC_RELAXED,
                                   __HIP_MEMORY_SCOPE_AGENT);
            const unsigned int old = __hip_atomic_fetch_add(a.guard + 1, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
            if (old + 1u == gridDim.x - 1u && a.host_stats) {
                const unsigned long long seen =
                    __hip_atomic_load((unsigned long long *)(a.guard + 2), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1ull;
                const unsigned int gave_up = __hip_atomic_load(a.guard, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0u ? 1u : 0u;
                __hip_atomic_store(a.host_stats + 1, gave_up, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                __hip_atomic_store(a.host_stats + 2, (unsigned int)seen, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                __hip_atomic_store(a.host_stats + 3, (unsigned int)(seen >> 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                __hip_atomic_store(a.host_stats + 4, (unsigned int)a.B, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                __hip_atomic_store(a.host_stats + 5, (unsigned int)a.N, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                __hip_atomic_store(a.host_stats + 0, a.stats_seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
            }
        }
    }
}

}  // namespace annlite

using namespace annlite;

template <int M, int NW, bool SKEWED, int NQ, int CB, bool RQ = false, int LK = 16, bool TL = false>
static int launch_q8(const ScanArgs &a, int grid, hipStream_t st) {
    constexpr int QT = 32;  // (the control block is laid out for 32 slots whatever the kernel uses)
    // (the row queue's parking area, 3072 B behind everything else, exists only for the kernels that run it: the 128 KB tables of
    // M = 32 / M = 8 with uint16 codes + 16 KB of 64-key lists + the 8 KB ring fit the 160 KB without it)
    const size_t need = (size_t)(M == 64 ? (a.Ks + 1) * 512 : M == 32 ? 131072 : (M == 16 && NQ == 2) ? kQ8Image16 : a.Ks * NQ * M * 16) + 1664 +
                        (size_t)QT * LK * 8 + QT * 8 +
                        (size_t)kRingSize * q8_ring_entry_bytes<M, LK>() + 4 * 128 * 9 + 32 + 32 + 16 + ((RQ || LK == 16) ? 3072 : 0) + (TL ? 128 : 0);
    ANNLITE_REQUIRE(need <= 160 * 1024, "byte-table kernel: %zu B of LDS", need);
    auto fn = adc_scan_q8_kernel<M, NW, SKEWED, NQ, CB, RQ, LK, TL>;
    ANNLITE_HIP_TRY(hipFuncSetAttribute((const void *)fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)need));
    hipLaunchKernelGGL(fn, dim3(grid), dim3(NW * 64), need, st, a);
    return launch_status("adc_scan_q8_kernel");
}

int annlite::launch_q8_scan(int id, bool sk, const ScanArgs &a, int grid, hipStream_t st) {
    switch (id) {
        case 1650:
            // shared bounds (the plain search): the row queue.  Unshared slices (the candidate generator of the re-rank stage: every
            // slice keeps 16 keys, the consumer is the bottleneck) keep the enumeration in the scanning waves -- with the row queue the
            // re-rank leg ran at 513 k q/s instead of 554 k
            if (a.gkey) return sk ? launch_q8<16, 16, true, 2, 1, true>(a, grid, st) : launch_q8<16, 16, false, 2, 1, true>(a, grid, st);
            return sk ? launch_q8<16, 16, true, 2, 1>(a, grid, st) : launch_q8<16, 16, false, 2, 1>(a, grid, st);
        case 1651:  // M = 16, k <= 16, cell tiles (annlite_ivf_search_topk): one work item per tile of (query, cell) pairs
            if (!a.gkey || !a.tile_rows || !a.vmap || !a.btab || !a.gseed0 || a.tile_done || a.gk2 || a.guard) {
                set_error("the cell-tile kernel needs vmap / tile_rows / per-query byte tables and shared bounds by query");
                return ANNLITE_ERR_UNSUPPORTED;
            }
            return sk ? launch_q8<16, 16, true, 2, 1, true, 16, true>(a, grid, st) : launch_q8<16, 16, false, 2, 1, true, 16, true>(a, grid, st);
        case 1664:  // M = 16, 16 < k <= 64: 64-key lists (one insertion per wave operation), the slices merged by merge_partial_kernel
            if (!a.gkey || a.tile_done) { set_error("the 64-key-list kernel serves the shared-bound search without an in-kernel merge"); return ANNLITE_ERR_UNSUPPORTED; }
            return sk ? launch_q8<16, 16, true, 2, 1, true, 64>(a, grid, st) : launch_q8<16, 16, false, 2, 1, true, 64>(a, grid, st);
        // 16 < k <= 64 for M = 8 / 32 (round 6: the reference's own PQ test searches topk = 50 at M = 8, Ks 256 / 512 / 768,
        // tests/test_pq_index.py:78-135): the same kernels with 64-key lists, merged by merge_partial_kernel like 1664
        case 864: case 3264: case 8650: case 8651:
            if (!a.gkey || a.tile_done) { set_error("the 64-key-list kernels serve the shared-bound search without an in-kernel merge"); return ANNLITE_ERR_UNSUPPORTED; }
            if (id == 864) return sk ? launch_q8<8, 16, true, 2, 1, false, 64>(a, grid, st) : launch_q8<8, 16, false, 2, 1, false, 64>(a, grid, st);
            if (id == 3264) return sk ? launch_q8<32, 16, true, 1, 1, false, 64>(a, grid, st) : launch_q8<32, 16, false, 1, 1, false, 64>(a, grid, st);
            if (sk) { set_error("uint16 codes: PLAIN rows only"); return ANNLITE_ERR_UNSUPPORTED; }
            return id == 8651 ? launch_q8<8, 16, false, 1, 2, false, 64>(a, grid, st) : launch_q8<8, 16, false, 2, 2, false, 64>(a, grid, st);
        case 6450: return sk ? launch_q8<64, 16, true, 2, 1>(a, grid, st) : launch_q8<64, 16, false, 2, 1>(a, grid, st);
        case 3250:  // M = 32: one entry group, 16 queries per workgroup, two half tables of 16 sub-spaces
            return sk ? launch_q8<32, 16, true, 1, 1>(a, grid, st) : launch_q8<32, 16, false, 1, 1>(a, grid, st);
        case 850:  // M = 8, uint16 codes (PLAIN rows): two entry groups (Ks <= 512)
        case 851:  // ... one (Ks <= 1024)
            if (sk) { set_error("uint16 codes: PLAIN rows only"); return ANNLITE_ERR_UNSUPPORTED; }
            return id == 851 ? launch_q8<8, 16, false, 1, 2>(a, grid, st) : launch_q8<8, 16, false, 2, 2>(a, grid, st);
        case 852:  // M = 8, uint8 codes (Ks <= 256): two entry groups, 64 KB table
            return sk ? launch_q8<8, 16, true, 2, 1>(a, grid, st) : launch_q8<8, 16, false, 2, 1>(a, grid, st);
        default: set_error("no byte-table kernel with id %d", id); return ANNLITE_ERR_UNSUPPORTED;
    }
}
