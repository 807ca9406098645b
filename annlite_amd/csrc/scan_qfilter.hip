// scan_qfilter.hip -- the default ADC scan kernels: integer filter over quantised tables in LDS, exact fp32
// recompute for the rows that pass (bit-exact results), shared top-k lists, bounds shared across row slices.
// See DESIGN.md section 3.1 for the design and its measured history.
#include "scan_lists.h"

namespace annlite {

// =================================================================================================
// Tile mode (IVF cells, annlite_pq_search_tiles): what the two kernels below share.
// A tile is short (a 39k-row cell = 38 steps of a 16-wave workgroup) and the exact fp32 recompute of its ~1300
// candidates (16 cold table gathers each) cost more than the scan itself, so tiles are scanned with INTEGERS only:
//   * every wave APPENDS the rows that pass the filter, with their integer sums S, to a per-slot buffer in LDS;
//   * in ROUNDS (after steps 0, 3, 15, 63, 255, ... and the last one) the wave that owns slot q folds the slot's
//     buffer into the slot's k smallest (S, row), which tightens the filter -- S <= Sk + margin, see below -- and
//     EMITS the rows that still pass to the slot's candidate list in global memory;
//   * the exact distances of the emitted rows are computed afterwards, per QUERY, by ivf_rescore_kernel (ivf.hip)
//     with the query's fp32 table in LDS.  A list that overflows is flagged: the re-score pass walks that cell.
// Filter bound from integers: every entry satisfies lo + step*(Q - 0.002) <= v <= lo + step*(Q + 1.002), so a row's
// exact-arithmetic sum is L + step*(S - 0.002 M) <= d_real <= L + step*(S + 1.002 M) and its fp32 sum is within
// slack32 of d_real.  For ANY k distinct rows of the cell with S <= Sk the cell's k-th fp32 distance is therefore
// <= U = L + step*(Sk + 1.002 M) + slack32, and a row can be in the top-k only if d_real <= U + slack32, i.e.
//     S <= Sk + margin,  margin = floor(1.004 M + 2 slack32 / step) + 1.
// The first Sk of a tile (the "seed") is the k-th smallest of the 4 NW minima over the 16-lane rows of every wave's
// first 64-row block (4 NW disjoint groups of 16 rows).
// LDS (on top of the streaming layout): [ccnt u32 x QT][cbuf u64 x QT x CAP] behind the wave queues; the gkl slots
// hold the margins, the gjl slots the emitted counts and overflow flags (one slice per tile: both are idle).
// =================================================================================================
template <int QT>
struct TileState {
    static constexpr int CAP = kTileCandBytes / (QT * 8);
    volatile uint32_t *ccnt;        // [QT] rows buffered since the last round
    uint32_t *gcnt;                 // [QT] rows emitted
    volatile uint32_t *gflag;       // [QT] the emitted list overflowed
    volatile uint32_t *marg;        // [QT]
    unsigned long long *cbuf;       // [QT][CAP] (S << 32 | row)
    unsigned long long *lists;      // [QT][64] k smallest (S, row) of every slot, ascending over the lanes
    volatile uint16_t *shq;         // [QT] 0x8000 | filter bound
    uint32_t *cand;                 // the tile's QT lists in global memory
    int cand_cap;

    __device__ __forceinline__ void bind(unsigned char *smem, uint32_t queue_end, uint32_t gjl_off, uint32_t gkl_off,
                                         unsigned long long *lists_, volatile uint16_t *shq_, const ScanArgs &a, int tile) {
        ccnt = (volatile uint32_t *)(smem + queue_end);
        cbuf = (unsigned long long *)(smem + queue_end + 64);
        gcnt = (uint32_t *)(smem + gjl_off);
        gflag = (volatile uint32_t *)(smem + gjl_off + 64);
        marg = (volatile uint32_t *)(smem + gkl_off);
        lists = lists_;
        shq = shq_;
        cand = a.cand + (int64_t)tile * QT * a.cand_cap;
        cand_cap = a.cand_cap;
    }
    template <int M>
    __device__ __forceinline__ void init_slot(int q, bool real, float smax_b, float qstep_b) {
        ccnt[q] = 0;
        gcnt[q] = 0;
        gflag[q] = 0;
        uint32_t mg = 0;
        if (real) {
            const double slack = (double)smax_b * (2.0 * M * 5.9604644775390625e-08 * (1.0 + 1.0 / 1024.0));
            double x = __builtin_floor(1.004 * M + 2.0 * slack / (double)qstep_b) + 1.0;
            if (!(x < 32767.0)) x = 32767.0;
            mg = (uint32_t)x;
        }
        marg[q] = mg;
        shq[q] = real ? (unsigned short)0xffff : (unsigned short)0x7fff;  // everything / nothing passes
    }
    __device__ __forceinline__ void tighten(int q, uint32_t sk) {
        uint32_t thr = sk + marg[q];
        if (thr > 32767u) thr = 32767u;
        const unsigned short nb = (unsigned short)(0x8000u | thr);
        if (nb < shq[q]) shq[q] = nb;
    }
    // rows of slot q go to its global list (or raise the flag: the re-score pass then walks the whole cell)
    __device__ __forceinline__ void emit(int q, unsigned long long pm, uint32_t rid, int lane) {
        const int n = __popcll(pm);
        uint32_t base = 0;
        if (lane == 0) base = atomicAdd(&gcnt[q], (uint32_t)n);
        base = (uint32_t)__builtin_amdgcn_readfirstlane((int)base);
        const int rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(pm >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)pm, 0u));
        if (base + (uint32_t)n > (uint32_t)cand_cap) {
            if (lane == 0) gflag[q] = 1u;
        } else if ((pm >> lane) & 1ull) {
            cand[(int64_t)q * cand_cap + base + rank] = rid;
        }
    }
    // the lanes in pm passed slot q's filter with integer sum sv
    __device__ __forceinline__ void append(int q, unsigned long long pm, uint32_t sv, uint32_t rid, int lane) {
        const int n = __popcll(pm);
        uint32_t base = 0;
        if (lane == 0) base = atomicAdd((uint32_t *)&ccnt[q], (uint32_t)n);
        base = (uint32_t)__builtin_amdgcn_readfirstlane((int)base);
        const int rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(pm >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)pm, 0u));
        const bool mine = (pm >> lane) & 1ull;
        const bool fits = base + (uint32_t)rank < (uint32_t)CAP;
        if (mine && fits) cbuf[q * CAP + base + rank] = ((unsigned long long)sv << 32) | rid;
        if (base + (uint32_t)n > (uint32_t)CAP) emit(q, __ballot(mine && !fits), rid, lane);  // buffer full
    }
    static __device__ __forceinline__ bool is_round(int step_no, int n_steps) {
        const int r = step_no + 1;
        return ((r & (r - 1)) == 0 && (__builtin_ctz((unsigned)r) & 1) == 0) || (r & 255) == 0 || r == n_steps;
    }
    // the calling wave owns slot q (between the two barriers of a round)
    __device__ __forceinline__ void round(int q, int km1, int lane) {
        uint32_t n = ccnt[q];
        if (n > (uint32_t)CAP) n = CAP;  // (what did not fit went straight to the global list)
        if (n == 0) return;
        const unsigned long long le = lists[q * 64 + lane];
        WaveList L;
        L.hi = (uint32_t)(le >> 32);
        L.lo = (uint32_t)le;
        for (uint32_t c0 = 0; c0 < n; c0 += 64) {
            const bool act = c0 + lane < n;
            const unsigned long long e = act ? cbuf[q * CAP + c0 + lane] : ~0ull;
            const uint32_t thi = __builtin_amdgcn_readlane(L.hi, km1), tlo = __builtin_amdgcn_readlane(L.lo, km1);
            const unsigned long long px = __ballot(act && key_less((uint32_t)(e >> 32), (uint32_t)e, thi, tlo));
            if (px) wavelist_insert_many(L, px, (uint32_t)(e >> 32), (uint32_t)e, lane);
        }
        lists[q * 64 + lane] = ((unsigned long long)L.hi << 32) | L.lo;
        const uint32_t sk = __builtin_amdgcn_readlane(L.hi, km1);
        uint32_t thr = 32767u;
        if (sk != kKeyInfHi) {
            thr = sk + marg[q];
            if (thr > 32767u) thr = 32767u;
            if (lane == 0 && (unsigned short)(0x8000u | thr) < shq[q]) shq[q] = (unsigned short)(0x8000u | thr);
        }
        for (uint32_t c0 = 0; c0 < n; c0 += 64) {  // emit what still passes the (tightened) filter
            const bool act = c0 + lane < n;
            const unsigned long long e = act ? cbuf[q * CAP + c0 + lane] : ~0ull;
            const unsigned long long pm = __ballot(act && (uint32_t)(e >> 32) <= thr);
            if (pm) emit(q, pm, (uint32_t)e, lane);
        }
        if (lane == 0) ccnt[q] = 0;
    }
    __device__ __forceinline__ void finish(int q, uint32_t *cand_count_slot) {
        const uint32_t g = gcnt[q];
        *cand_count_slot = (gflag[q] || g > (uint32_t)cand_cap) ? 0xffffffffu : g;
    }
};

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
// CODE16: uint16 codes (Ks up to 1024 at M = 8, 512 at M = 16: what fits the LDS with 8 queries per workgroup), PLAIN layout,
// row slices only -- the reference's own PQ tests run Ks = 512 and 768 (tests/test_pq_index.py:80-163)
template <int M, int NQ, int NW, int WPS, bool SKEWED, bool TILES, bool CODE16 = false>
__global__ __launch_bounds__(NW * 64, WPS) void adc_scan_qfilter_kernel(const ScanArgs a) {
    // (the u16-table pass behind a byte-table launch that may give up: ScanArgs::gate)
    if (a.gate && __hip_atomic_load(a.gate, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) return;
    static_assert(!CODE16 || (!SKEWED && !TILES), "uint16 code tables: PLAIN layout, row slices");
    constexpr int QG = 8;                 // queries per LDS entry
    constexpr int QT = QG * NQ;           // queries per workgroup
    constexpr int CW = CODE16 ? M / 2 : M / 4;  // dwords of a code row
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
    // forward rotation (PLAIN tables) by s codes = sb bytes
    const int sb = CODE16 ? 2 * s : s;
    const uint32_t bsh = (uint32_t)(sb & 3);
    bool abit[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) abit[i] = (((sb >> 2) >> i) & 1) != 0;
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

    for (int it = 0;; ++it) {
        int item = blockIdx.x + it * gridDim.x;
        int tile, slice;
        int64_t slice_begin, slice_end;
        unsigned int next_item = 0;  // (tile mode, thread 0)
        if constexpr (TILES) {
            // tile mode: one item per query tile with its own row range, taken from a device-wide counter.  The
            // index of the NEXT item is fetched while this one's table is loaded (two LDS slots, alternating), so
            // the atomic's round trip is off the critical path.
            volatile unsigned int *s_item = (volatile unsigned int *)(smem + shq_off + 48);
            if (it == 0 && tid == 0)
                s_item[0] = __hip_atomic_fetch_add(a.item_counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1u;
            __syncthreads();  // every wave is done with the previous item; the slot of this one is written
            item = (int)s_item[it & 1];
            if (item >= n_items) break;
            if (tid == 0) next_item = __hip_atomic_fetch_add(a.item_counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1u;
            tile = item;
            slice = 0;
            slice_begin = a.tile_rows[2 * tile];
            slice_end = a.tile_rows[2 * tile + 1];
            if (slice_begin < 0) {  // unused tile: nobody reads its outputs
                if (tid == 0) s_item[(it + 1) & 1] = next_item;
                continue;
            }
        } else {
            if (item >= n_items) break;
            if (!item_map(a, item, tile, slice)) continue;
            slice_begin = (int64_t)slice * a.slice_rows;
            slice_end = slice_begin + a.slice_rows;
        }
        if (slice_end > a.N) slice_end = a.N;
        TileState<QT> ts;
        if constexpr (TILES) ts.bind(smem, queue_off + NW * 512, gjl_off, gkl_off, lists, shq, a, tile);

        __syncthreads();
        volatile int32_t *s_vm = (volatile int32_t *)(smem + queue_off);  // tile mode: query of every slot (-1: padding)
        if constexpr (TILES) {
            if (tid < QT) s_vm[tid] = a.vmap[tile * QT + tid];
            __syncthreads();
        }
        {
            if constexpr (TILES) {
                // The slots of a tile hold ARBITRARY queries: their quantised tables are gathered from the tables of the
                // real queries ([B/8][Ks][M][8] u16, built once per batch) -- one u16 per (slot, code, sub-space), two
                // slots per LDS dword, a wave writes 64 consecutive dwords.  (Rebuilding the tables per slot from the
                // query vectors cost 0.14 of 0.9 ms at 18k slots for 1024 queries.)
                // A thread keeps its (sub-space, slot pair, entry group) for the whole fill -- only the code advances --
                // so its two source pointers are loop-invariant and the loads of several codes are in flight together.
                constexpr int NT = NW * 64, LPR = 4 * M, RPI = NT / LPR;  // lanes per (code, group) row, rows per sweep
                static_assert(NT % LPR == 0 && RPI % NQ == 0, "fill mapping");
                const unsigned char *q16b = (const unsigned char *)a.q16;
                const int sp = tid & 3, m = (tid >> 2) % M, kh0 = tid / LPR;
                const int h = kh0 % NQ, k0 = kh0 / NQ;
                const int r0 = s_vm[h * QG + sp * 2], r1 = s_vm[h * QG + sp * 2 + 1];
                const uint32_t m0 = r0 >= 0 ? 0xffffu : 0u, m1 = r1 >= 0 ? 0xffffu : 0u;  // padding slots: zeros
                const unsigned char *p0 = q16b + (int64_t)((r0 >= 0 ? r0 : 0) >> 3) * group_bytes + m * 16 + ((r0 >= 0 ? r0 : 0) & 7) * 2;
                const unsigned char *p1 = q16b + (int64_t)((r1 >= 0 ? r1 : 0) >> 3) * group_bytes + m * 16 + ((r1 >= 0 ? r1 : 0) & 7) * 2;
                unsigned char *dst = smem + h * RB + m * 16 + sp * 4;
#pragma unroll 8
                for (int kk = k0; kk < a.Ks; kk += RPI / NQ) {
                    const uint32_t lo = *(const uint16_t *)(p0 + kk * RB) & m0;
                    const uint32_t hi = *(const uint16_t *)(p1 + kk * RB) & m1;
                    *(uint32_t *)(dst + kk * (NQ * RB)) = lo | (hi << 16);
                }
            } else {
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
            }
            if (tid < QT) {
                locks[tid] = 0;
                // start from whatever other workgroups (other row slices, earlier items) already proved
                const int b = tile * QT + tid;
                const unsigned long long gk =
                    a.gkey ? __hip_atomic_load(a.gkey + b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : ~0ull;
                if constexpr (!TILES) {  // (tile mode keeps its margins / counters in these slots)
                    gkl[tid] = gk;
                    ((unsigned long long *)(smem + gjl_off))[tid] = ~0ull;
                }
                // pad queries of the last tile (b >= B) must never pass the filter (their all-zero tables give S = 0
                // for every row: 15 pad queries made a 1-query batch 15x slower than a 16-query one): 0x7fff - S
                // never has bit 15 set and never borrows from the neighbouring field
                bool real = b < a.B;
                if constexpr (TILES) {
                    const int rb = s_vm[tid];  // padding slots sit in every tile
                    real = real && rb >= 0;
                    ts.template init_slot<M>(tid, real, real ? a.smax[rb] : 0.f, real ? a.qstep[rb] : 1.f);
                } else
                shq[tid] = real ? qbound_from_key<M>(gk, a.smax[b], a.qstep[b], a.qlo[b]) : (unsigned short)0x7fff;
            }
            for (int idx = tid; idx < QT * 64; idx += NW * 64) lists[idx] = ~0ull;
            if (tid == 0) *(volatile uint32_t *)(smem + shq_off + 56) = 0;  // the block counter the waves draw from
            if constexpr (TILES)
                if (tid == 0) ((volatile unsigned int *)(smem + shq_off + 48))[(it + 1) & 1] = next_item;
        }
        __syncthreads();

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
                if constexpr (CODE16) {
                    uint32_t o0, o1;
                    word_shl2(cc[w], (uint32_t)ilog2_c(KSTRIDE), o0, o1);
                    addr[2 * w + 0] = mbase[2 * w + 0] + o0;
                    addr[2 * w + 1] = mbase[2 * w + 1] + o1;
                } else {
                    uint32_t o0, o1, o2, o3;
                    byte_shl4(cc[w], (uint32_t)ilog2_c(KSTRIDE), o0, o1, o2, o3);
                    addr[4 * w + 0] = mbase[4 * w + 0] + o0;
                    addr[4 * w + 1] = mbase[4 * w + 1] + o1;
                    addr[4 * w + 2] = mbase[4 * w + 2] + o2;
                    addr[4 * w + 3] = mbase[4 * w + 3] + o3;
                }
            });
        };
        // integer sums of one entry group: 4 dwords x (2 x u16)
        auto group_sum = [&](auto H, u32x4 &acc) {
            constexpr int h = decltype(H)::value;
            // issue all M look-ups of the group before the first add (the compiler otherwise re-used one
            // register pair and waited lgkmcnt(0) after every second load)
            // (tile mode keeps 8 look-ups in flight instead of M: its extra scalar state pushed 9 of the 16 loop-
            // invariant LDS base registers into scratch, reloaded every step -- 2.2 us per step instead of 1.3)
            constexpr int CH = (TILES && M > 8) ? 8 : M;
            static_for<0, M / CH>([&](auto C) {
                constexpr int c0 = decltype(C)::value * CH;
                u32x4 v[CH];
                static_for<0, CH>([&](auto T) {
                    constexpr int t = decltype(T)::value;
                    v[t] = *(const u32x4 *)(addr[c0 + t] + h * RB);
                });
                asm volatile("" ::: "memory");
                if constexpr (c0 == 0) acc = v[0];
                else acc += v[0];
                static_for<1, CH>([&](auto T) { acc += v[decltype(T)::value]; });
            });
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
        // the code bytes (and validity word) of a lane's row are fetched ONE STEP AHEAD: issued at the top of a step for the
        // next one and picked up at the top of that one (fetched two steps ahead and rotated at the END of the step, the
        // compiler waited for the load it had just issued -- s_waitcnt vmcnt(0) in every step)
        // Row slices: the waves DRAW their blocks of 64 rows from a counter in LDS (draw_block).  The SIMD's arbiter
        // favours its oldest wave and one wave alone issues at about a third of the rate four reach together
        // (scripts/ubench/valu_cost.hip, step_loop.hip): with the rows dealt out statically the favoured waves finished
        // their share early and the last ones ran the slice out alone.  (Tile mode keeps the static deal: its rounds'
        // barriers need every wave to run the same number of steps.)
        uint32_t *blk_ctr = (uint32_t *)(smem + shq_off + 56);
        auto draw_block = [&]() -> uint32_t {  // (lane 0's value; broadcast a step later, where it is first needed)
            uint32_t v = 0;
            if (lane == 0) v = atomicAdd(blk_ctr, 1u);
            return v;
        };
        uint32_t b_cur = 0, b_nxt = 0, b_pend = 0;
        if constexpr (!TILES) {
            b_cur = (uint32_t)__builtin_amdgcn_readfirstlane((int)draw_block());
            b_pend = draw_block();
            row0 = slice_begin + (int64_t)b_cur * 64;
        }
        if (TILES ? row0 < slice_end : true) {
            load_row(row0 + lane, cnext);
            vnext = load_valid(row0 + lane);
#pragma unroll
            for (int i = 0; i < CW; ++i) ccur[i] = cnext[i];  // (tile mode: the seed bound below looks at the first row)
            vcur = vnext;
            if constexpr (!SKEWED) rotate_row<CW>(ccur, abit, bsh);
        }
        if constexpr (!TILES) b_nxt = (uint32_t)__builtin_amdgcn_readfirstlane((int)b_pend);
        if constexpr (TILES) {
            // Integer seed bound (no separate seed launch): every 16-lane row of every wave takes the per-slot MINIMUM
            // integer sum S of its 16 rows; the 4 NW minima belong to distinct rows, so >= k rows have S <= Sk := the
            // k-th smallest of them, and the filter starts at S <= Sk + margin (TileState).  Without it every row of
            // the first steps passes: the slot buffers overflow into the global lists before the first round.
            constexpr int NG = NW * 4;  // one minimum per 16-lane row of every wave: NG disjoint groups of 16 rows
            if (a.k <= NG) {
                uint32_t *smin = (uint32_t *)(smem + queue_off);  // [NG][NQ * 4] packed minima (queues are idle)
                u32x4 sacc[NQ];
                const bool have = row0 < slice_end;
                bool ok = have && row0 + lane < slice_end;
                if (ok && a.valid) ok = (vcur >> (lane & 31)) & 1u;
                if (have) {
                    make_addr(ccur);
                    static_for<0, NQ>([&](auto H) { group_sum(H, sacc[decltype(H)::value]); });
                }
#pragma unroll
                for (int h = 0; h < NQ; ++h)
#pragma unroll
                    for (int w = 0; w < 4; ++w) {
                        uint32_t x = ok ? sacc[h][w] : 0x7fff7fffu;
#pragma unroll
                        for (int o = 1; o < 16; o <<= 1) {
                            const uint32_t y = (uint32_t)__shfl_xor((int)x, o);
                            asm("v_pk_min_u16 %0, %1, %2" : "=v"(x) : "v"(x), "v"(y));
                        }
                        if ((lane & 15) == 0) smin[(wave * 4 + (lane >> 4)) * (NQ * 4) + h * 4 + w] = x;
                    }
                __syncthreads();
                if (tid < QT * NG) {
                    // thread (q, i) ranks minimum i of slot q among the NG (ties by group); the one of rank k-1 sets the bound
                    const int q = tid / NG, i = tid - q * NG;
                    const int b = tile * QT + q;
                    if (b < a.B && a.vmap[b] >= 0) {
                        const volatile uint16_t *sm16 = (const volatile uint16_t *)smin;
                        const uint32_t vi = sm16[i * (NQ * 8) + q];
                        int rank = 0;
#pragma unroll 4
                        for (int j = 0; j < NG; ++j) {
                            const uint32_t vj = sm16[j * (NQ * 8) + q];
                            rank += (vj < vi) || (vj == vi && j < i);
                        }
                        if (rank == km1 && vi < 0x7fffu) ts.tighten(q, vi);
                    }
                }
                __syncthreads();
#pragma unroll
                for (int h = 0; h < NQ; ++h) thp[h] = *(const u32x4 *)(smem + lut_bytes + h * 16);
            }
        }

        const FlushCtx fc = {(const uint8_t *)a.codes, a.lut, a.smax, a.qstep, a.qlo, a.gkey, a.gk2, a.dbg,
                             a.Ks, tile * QT, a.n_slices, slice, km1, a.jm1, a.dbg_skip,
                             list_off, lock_off, shq_off, gkl_off, gjl_off, 0u};
        int qcnt = 0;  // entries in this wave's candidate queue
        int step_no = 0;
        if constexpr (TILES) {
            // integers only: append -> rounds -> emission (TileState); all waves run the same number of steps so
            // that the rounds' barriers meet
            const int n_steps = (int)((slice_end - slice_begin + stride - 1) / stride);
            for (; step_no < n_steps; ++step_no, row0 += stride) {
                if (row0 < slice_end) {
#pragma unroll
                    for (int i = 0; i < CW; ++i) ccur[i] = cnext[i];
                    vcur = vnext;
                    load_row(row0 + stride + lane, cnext);
                    vnext = load_valid(row0 + stride + lane);
                    if constexpr (!SKEWED) rotate_row<CW>(ccur, abit, bsh);
                    unsigned long long vmask = ~0ull;
                    if (slice_end - row0 < 64) vmask = (1ull << (int)(slice_end - row0)) - 1ull;
                    if (a.valid) vmask &= __ballot((vcur >> (lane & 31)) & 1u);
                    const uint32_t rid = (uint32_t)(row0 + lane);
                    make_addr(ccur);
                    u32x4 acc[NQ];
                    static_for<0, NQ>([&](auto H) { group_sum(H, acc[decltype(H)::value]); });
                    uint32_t anyv = 0;
#pragma unroll
                    for (int h = 0; h < NQ; ++h)
#pragma unroll
                        for (int w = 0; w < 4; ++w) anyv |= thp[h][w] - acc[h][w];
                    const unsigned long long anym = __ballot((anyv & 0x80008000u) != 0) & vmask;
                    if (anym && !(a.dbg_skip & 4)) {
                        if (a.dbg && lane == 0) atomicAdd(a.dbg + 0, 1ull);
#pragma unroll
                        for (int h = 0; h < NQ; ++h) {
#pragma unroll
                            for (int w = 0; w < 4; ++w) {
                                const uint32_t x = (thp[h][w] - acc[h][w]) & 0x80008000u;
                                if (__ballot(x != 0) & vmask) {
#pragma unroll
                                    for (int half = 0; half < 2; ++half) {
                                        const unsigned long long pm =
                                            __ballot((x & (half ? 0x80000000u : 0x8000u)) != 0) & vmask;
                                        if (pm)
                                            ts.append(h * QG + w * 2 + half, pm,
                                                      half ? (acc[h][w] >> 16) : (acc[h][w] & 0xffffu), rid, lane);
                                    }
                                }
                            }
                        }
                    }
                }
                if (TileState<QT>::is_round(step_no, n_steps)) {
                    __syncthreads();  // every wave's appends are in LDS
                    for (int q = wave; q < QT; q += NW) ts.round(q, km1, lane);
                    __syncthreads();  // bounds published, buffers empty
#pragma unroll
                    for (int h = 0; h < NQ; ++h) thp[h] = *(const u32x4 *)(smem + lut_bytes + h * 16);
                }
            }
        } else
        for (const uint32_t n_blocks = (uint32_t)((slice_end - slice_begin + 63) >> 6); b_cur < n_blocks; ++step_no) {
            row0 = slice_begin + (int64_t)b_cur * 64;  // (< slice_end)
            b_pend = draw_block();
#pragma unroll
            for (int i = 0; i < CW; ++i) ccur[i] = cnext[i];
            vcur = vnext;
            {
                const int64_t row1 = slice_begin + (int64_t)b_nxt * 64 + lane;  // (past the slice at its end: clamped, unused)
                load_row(row1, cnext);
                vnext = load_valid(row1);
            }
            if constexpr (!SKEWED) rotate_row<CW>(ccur, abit, bsh);
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
                                        qfilter_flush<M, SKEWED, CODE16>(fc, queue_off + wave * 512, qcnt);
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
                qfilter_flush<M, SKEWED, CODE16>(fc, queue_off + wave * 512, qcnt);
                qcnt = 0;
                flushed = true;
            }
            // Import what the other workgroups of these queries (the other row slices) have proven: the best
            // k-th key any of them published and, per group of 8 concurrently scanned slices, the k-th smallest
            // of their j smallest keys (sibling_bound; +1: that row itself must still be accepted).  Bounds
            // move on a log scale, so: steps 0, 1, 3, 7, ... then every 64th.
            if (!TILES && a.gkey) {  // (tile mode: one slice per tile, nothing to import)
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
                                    const unsigned long long v = sibling_bound<gk2_cell_keys(M)>(a.gk2, b, a.n_slices, g0, a.jm1, km1, lane);
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
            b_cur = b_nxt;
            b_nxt = (uint32_t)__builtin_amdgcn_readfirstlane((int)b_pend);
        }

        if (qcnt) qfilter_flush<M, SKEWED, CODE16>(fc, queue_off + wave * 512, qcnt);

        // ---- the shared lists ARE the workgroup's result for this (tile, slice) ----------------------
        __syncthreads();
        if constexpr (TILES) {
            // the tile's result = the candidate lists it emitted; their lengths (or "walk the whole cell")
            if (tid < QT) ts.finish(tid, a.cand_count + tile * QT + tid);
            continue;
        }
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
            if (*s_flag) merge_tile_slices<NW>(a, tile * QT, QT, km1, wave, lane, (unsigned long long *)(smem + queue_off));
            __syncthreads();  // s_flag (the lock words) is re-initialised by the next item
        }
    }
}


// =================================================================================================
// Quantised-filter kernel for M = 64 (BASELINE config 4: 768-d, 12-float sub-spaces).  Same discipline as
// adc_scan_qfilter_kernel; what differs is dictated by the table size (64 sub-spaces x 256 codes):
//   * 4 queries per 8-byte LDS entry (u16 each), ds_read_b64; two half tables of 32 sub-spaces, [Ks + 1][32][8 B] each
//     (128.5 KB together): a half-wave reads columns (l + p) % 32 for 32 consecutive l: bank pair (l + p) % 32,
//     conflict-free;
//   * no per-step LDS base registers and ONE VALU instruction per look-up address: the wrap-coded SKEWED layout
//     (wrap64_mask) makes the address v_perm_b32(code dword, lane constant) = (stored byte << 8) | (l % 32) * 8
//     [| 0x10000], with p * 8 [+ 0x100] as the instruction's immediate;
//   * PLAIN tables are rotated and wrap-coded on the fly (slow path; the index plugin stores SKEWED).
//   * QMAX = floor(32767 / 64) = 511 (9-bit entries), look-ups issued in 4 chunks of 16;
//   * PLAIN tables are rotated and wrap-coded on the fly (slow path; the index plugin stores SKEWED).
// LDS: [table (Ks+1)*512][shq u16 x 4 @ +0][locks u32 x 4 @ +64][gkl u64 x 4 @ +128][lists u64 x 4 x 64 @ +256]
//      [gjl u64 x 4][queues u64 x NW x 64]
// =================================================================================================
template <int NW, bool SKEWED, bool TILES>
__global__ __launch_bounds__(NW * 64) void adc_scan_qfilter64_kernel(const ScanArgs a) {
    constexpr int M = 64, QT = 4, CW = 16, RB = 512;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int km1 = a.k - 1;

    const int lut_bytes = (a.Ks + 1) * RB;
    const uint32_t shq_off = (uint32_t)lut_bytes, lock_off = shq_off + 64, gkl_off = shq_off + 128,
                   list_off = shq_off + 256, gjl_off = list_off + QT * 512, queue_off = gjl_off + 128;
    unsigned long long *gkl = (unsigned long long *)(smem + gkl_off);
    volatile uint16_t *shq = (volatile uint16_t *)(smem + shq_off);
    volatile uint32_t *locks = (volatile uint32_t *)(smem + lock_off);
    unsigned long long *lists = (unsigned long long *)(smem + list_off);
    // lane constant of the look-up addresses: byte 0 = (lane % 32) * 8 (the column), byte 2 = 0x01 (second half table).
    // The addresses are plain LDS byte offsets: the table starts at the workgroup's LDS address 0 (all LDS is dynamic)
    constexpr uint32_t HALF_B = (256u + 1u) * 256u;  // 0x10100: the second half table
    const uint32_t lane_k = 0x00010000u | (uint32_t)((lane & 31) * 8);
    if ((uint32_t)(uintptr_t)(__attribute__((address_space(3))) unsigned char *)smem != 0u) __builtin_trap();

    for (int it = 0;; ++it) {
        int item = blockIdx.x + it * gridDim.x;
        int tile, slice;
        int64_t slice_begin, slice_end;
        unsigned int next_item = 0;  // (tile mode, thread 0)
        if constexpr (TILES) {  // (see adc_scan_qfilter_kernel)
            volatile unsigned int *s_item = (volatile unsigned int *)(smem + shq_off + 48);
            if (it == 0 && tid == 0)
                s_item[0] = __hip_atomic_fetch_add(a.item_counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1u;
            __syncthreads();
            item = (int)s_item[it & 1];
            if (item >= a.n_items) break;
            if (tid == 0) next_item = __hip_atomic_fetch_add(a.item_counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1u;
            tile = item;
            slice = 0;
            slice_begin = a.tile_rows[2 * tile];
            slice_end = a.tile_rows[2 * tile + 1];
            if (slice_begin < 0) {
                if (tid == 0) s_item[(it + 1) & 1] = next_item;
                continue;
            }
        } else {
            if (item >= a.n_items) break;
            if (!item_map(a, item, tile, slice)) continue;
            slice_begin = (int64_t)slice * a.slice_rows;
            slice_end = slice_begin + a.slice_rows;
        }
        if (slice_end > a.N) slice_end = a.N;
        TileState<QT> ts;
        if constexpr (TILES) ts.bind(smem, queue_off + NW * 512, gjl_off, gkl_off, lists, shq, a, tile);

        __syncthreads();
        volatile int32_t *s_vm = (volatile int32_t *)(smem + queue_off);  // tile mode: query of every slot (-1: padding)
        if constexpr (TILES) {
            if (tid < QT) s_vm[tid] = a.vmap[tile * QT + tid];
            __syncthreads();
        }
        {
            if constexpr (TILES) {
                // gather the slots' tables from the real queries' ([B/4][Ks][64][4] u16); see adc_scan_qfilter_kernel
                constexpr int NT = NW * 64, LPR = 2 * M, RPI = NT / LPR;
                static_assert(NT % LPR == 0, "fill mapping");
                const unsigned char *q16b = (const unsigned char *)a.q16;
                const int64_t group_bytes = (int64_t)a.Ks * RB;
                const int sp = tid & 1, m = (tid >> 1) % M, k0 = tid / LPR;
                const int r0 = s_vm[sp * 2], r1 = s_vm[sp * 2 + 1];
                const uint32_t m0 = r0 >= 0 ? 0xffffu : 0u, m1 = r1 >= 0 ? 0xffffu : 0u;
                const unsigned char *p0 = q16b + (int64_t)((r0 >= 0 ? r0 : 0) >> 2) * group_bytes + m * 8 + ((r0 >= 0 ? r0 : 0) & 3) * 2;
                const unsigned char *p1 = q16b + (int64_t)((r1 >= 0 ? r1 : 0) >> 2) * group_bytes + m * 8 + ((r1 >= 0 ? r1 : 0) & 3) * 2;
                unsigned char *dst = smem + (m >> 5) * HALF_B + (m & 31) * 8 + sp * 4;
#pragma unroll 8
                for (int kk = k0; kk <= a.Ks; kk += RPI) {  // (row Ks = row 0)
                    const int ks = kk == a.Ks ? 0 : kk;
                    const uint32_t lo = *(const uint16_t *)(p0 + ks * RB) & m0;
                    const uint32_t hi = *(const uint16_t *)(p1 + ks * RB) & m1;
                    *(uint32_t *)(dst + kk * 256) = lo | (hi << 16);
                }
            } else {
                // global [Ks][64 sub-spaces][4 x u16] -> the two half tables (16-byte chunk c of code k: sub-spaces 2c, 2c + 1)
                const u32x4 *src = (const u32x4 *)((const unsigned char *)a.q16 + (int64_t)tile * a.Ks * RB);
                const int total = a.Ks * (RB / 16);
                for (int idx = tid; idx < total; idx += NW * 64) {
                    const int kk = idx >> 5, c = idx & 31;
                    const u32x4 e = src[idx];
                    *(u32x4 *)(smem + (c >> 4) * HALF_B + kk * 256 + (c & 15) * 16) = e;
                    if (kk == 0) *(u32x4 *)(smem + (c >> 4) * HALF_B + a.Ks * 256 + (c & 15) * 16) = e;  // row Ks = row 0
                }
            }
            if (tid < QT) {
                locks[tid] = 0;
                const int b = tile * QT + tid;
                const unsigned long long gk =
                    a.gkey ? __hip_atomic_load(a.gkey + b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : ~0ull;
                if constexpr (!TILES) {
                    gkl[tid] = gk;
                    ((unsigned long long *)(smem + gjl_off))[tid] = ~0ull;
                }
                // pad queries of the last tile (b >= B) must never pass the filter (their all-zero tables give S = 0
                // for every row: 15 pad queries made a 1-query batch 15x slower than a 16-query one): 0x7fff - S
                // never has bit 15 set and never borrows from the neighbouring field
                bool real = b < a.B;
                if constexpr (TILES) {
                    const int rb = s_vm[tid];
                    real = real && rb >= 0;
                    ts.template init_slot<M>(tid, real, real ? a.smax[rb] : 0.f, real ? a.qstep[rb] : 1.f);
                } else
                shq[tid] = real ? qbound_from_key<M>(gk, a.smax[b], a.qstep[b], a.qlo[b]) : (unsigned short)0x7fff;
            }
            for (int idx = tid; idx < QT * 64; idx += NW * 64) lists[idx] = ~0ull;
            if (tid == 0) *(volatile uint32_t *)(smem + shq_off + 56) = 0;  // the block counter the waves draw from
            if constexpr (TILES)
                if (tid == 0) ((volatile unsigned int *)(smem + shq_off + 48))[(it + 1) & 1] = next_item;
        }
        __syncthreads();

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
            skew64_encode(c, lane & 31);
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
        // the code bytes (and validity word) of a lane's row are fetched ONE STEP AHEAD: issued at the top of a step for the
        // next one and picked up at the top of that one (fetched two steps ahead and rotated at the END of the step, the
        // compiler waited for the load it had just issued -- s_waitcnt vmcnt(0) in every step)
        // row slices: the waves draw their blocks of 64 rows from a counter in LDS (see adc_scan_qfilter_kernel)
        uint32_t *blk_ctr = (uint32_t *)(smem + shq_off + 56);
        auto draw_block = [&]() -> uint32_t {
            uint32_t v = 0;
            if (lane == 0) v = atomicAdd(blk_ctr, 1u);
            return v;
        };
        uint32_t b_cur = 0, b_nxt = 0, b_pend = 0;
        if constexpr (!TILES) {
            b_cur = (uint32_t)__builtin_amdgcn_readfirstlane((int)draw_block());
            b_pend = draw_block();
            row0 = slice_begin + (int64_t)b_cur * 64;
        }
        if (TILES ? row0 < slice_end : true) {
            load_row(row0 + lane, cnext);
            vnext = load_valid(row0 + lane);
#pragma unroll
            for (int i = 0; i < CW; ++i) ccur[i] = cnext[i];  // (tile mode: the seed bound below looks at the first row)
            vcur = vnext;
            if constexpr (!SKEWED) encode_plain(ccur);
        }
        if constexpr (!TILES) b_nxt = (uint32_t)__builtin_amdgcn_readfirstlane((int)b_pend);
        // integer sums of this lane's row for the 4 queries (2 dwords x 2 u16), look-ups in 4 chunks of 16
        // (a ring of 16 landing registers refilled after every add, as the byte-table kernel has it, measured 2 % slower)
        typedef const u32x2 __attribute__((address_space(3))) *lds_entry_ptr;
        auto row_sums = [&](const uint32_t (&cc)[CW], u32x2 &acc) {
            static_for<0, 4>([&](auto C) {
                constexpr int c0 = decltype(C)::value * 16;
                constexpr int half = c0 / 32;
                u32x2 v[16];
                static_for<0, 16>([&](auto T) {
                    constexpr int t = c0 + decltype(T)::value;
                    // byte 0 <- lane_k byte 0, byte 1 <- code byte t % 4, byte 2 <- lane_k byte 2 (second half) or 0, byte 3 <- 0
                    constexpr uint32_t sel = 0x0c000000u | ((half ? 0x02u : 0x0cu) << 16) | ((4u + (uint32_t)(t % 4)) << 8);
                    const uint32_t ad = __builtin_amdgcn_perm(cc[t / 4], lane_k, sel);
                    v[t - c0] = *(lds_entry_ptr)(uintptr_t)(ad + (uint32_t)((t % 32) * 8 + half * 0x100));
                });
                asm volatile("" ::: "memory");
                static_for<0, 16>([&](auto I) { acc += v[decltype(I)::value]; });
            });
        };
        if constexpr (TILES) {
            // integer seed bound from the per-wave minima of the first rows (see adc_scan_qfilter_kernel)
            constexpr int NG = NW * 4;
            if (a.k <= NG) {
                uint32_t *smin = (uint32_t *)(smem + queue_off);  // [NG][2]
                u32x2 sacc = {0u, 0u};
                const bool have = row0 < slice_end;
                bool ok = have && row0 + lane < slice_end;
                if (ok && a.valid) ok = (vcur >> (lane & 31)) & 1u;
                if (have) row_sums(ccur, sacc);
#pragma unroll
                for (int w = 0; w < 2; ++w) {
                    uint32_t x = ok ? (w ? sacc.y : sacc.x) : 0x7fff7fffu;
#pragma unroll
                    for (int o = 1; o < 16; o <<= 1) {
                        const uint32_t y = (uint32_t)__shfl_xor((int)x, o);
                        asm("v_pk_min_u16 %0, %1, %2" : "=v"(x) : "v"(x), "v"(y));
                    }
                    if ((lane & 15) == 0) smin[(wave * 4 + (lane >> 4)) * 2 + w] = x;
                }
                __syncthreads();
                if (tid < QT * NG) {
                    const int q = tid / NG, i = tid - q * NG;
                    const int b = tile * QT + q;
                    if (b < a.B && a.vmap[b] >= 0) {
                        const volatile uint16_t *sm16 = (const volatile uint16_t *)smin;
                        const uint32_t vi = sm16[i * 4 + q];
                        int rank = 0;
#pragma unroll 4
                        for (int j = 0; j < NG; ++j) {
                            const uint32_t vj = sm16[j * 4 + q];
                            rank += (vj < vi) || (vj == vi && j < i);
                        }
                        if (rank == km1 && vi < 0x7fffu) ts.tighten(q, vi);
                    }
                }
                __syncthreads();
                thp = *(const u32x2 *)(smem + shq_off);
            }
        }
        const FlushCtx fc = {(const uint8_t *)a.codes, a.lut, a.smax, a.qstep, a.qlo, a.gkey, a.gk2, a.dbg,
                             a.Ks, tile * QT, a.n_slices, slice, km1, a.jm1, a.dbg_skip,
                             list_off, lock_off, shq_off, gkl_off, gjl_off, 0u};
        int qcnt = 0;
        int step_no = 0;
        if constexpr (TILES) {  // integers only: append -> rounds -> emission (TileState)
            const int n_steps = (int)((slice_end - slice_begin + stride - 1) / stride);
            for (; step_no < n_steps; ++step_no, row0 += stride) {
                if (row0 < slice_end) {
#pragma unroll
                    for (int i = 0; i < CW; ++i) ccur[i] = cnext[i];
                    vcur = vnext;
                    load_row(row0 + stride + lane, cnext);
                    vnext = load_valid(row0 + stride + lane);
                    if constexpr (!SKEWED) encode_plain(ccur);
                    unsigned long long vmask = ~0ull;
                    if (slice_end - row0 < 64) vmask = (1ull << (int)(slice_end - row0)) - 1ull;
                    if (a.valid) vmask &= __ballot((vcur >> (lane & 31)) & 1u);
                    const uint32_t rid = (uint32_t)(row0 + lane);
                    u32x2 acc = {0u, 0u};
                    row_sums(ccur, acc);
                    const uint32_t x0 = (thp.x - acc.x) & 0x80008000u, x1 = (thp.y - acc.y) & 0x80008000u;
                    const unsigned long long anym = __ballot((x0 | x1) != 0) & vmask;
                    if (anym && !(a.dbg_skip & 4)) {
                        if (a.dbg && lane == 0) atomicAdd(a.dbg + 0, 1ull);
#pragma unroll
                        for (int w = 0; w < 2; ++w) {
                            const uint32_t x = w ? x1 : x0;
                            const uint32_t sw = w ? acc.y : acc.x;
                            if (__ballot(x != 0) & vmask) {
#pragma unroll
                                for (int half = 0; half < 2; ++half) {
                                    const unsigned long long pm = __ballot((x & (half ? 0x80000000u : 0x8000u)) != 0) & vmask;
                                    if (pm) ts.append(w * 2 + half, pm, half ? (sw >> 16) : (sw & 0xffffu), rid, lane);
                                }
                            }
                        }
                    }
                }
                if (TileState<QT>::is_round(step_no, n_steps)) {
                    __syncthreads();
                    for (int q = wave; q < QT; q += NW) ts.round(q, km1, lane);
                    __syncthreads();
                    thp = *(const u32x2 *)(smem + shq_off);
                }
            }
        } else
        for (const uint32_t n_blocks = (uint32_t)((slice_end - slice_begin + 63) >> 6); b_cur < n_blocks; ++step_no) {
            row0 = slice_begin + (int64_t)b_cur * 64;  // (< slice_end)
            b_pend = draw_block();
#pragma unroll
            for (int i = 0; i < CW; ++i) ccur[i] = cnext[i];
            vcur = vnext;
            {
                const int64_t row1 = slice_begin + (int64_t)b_nxt * 64 + lane;  // (past the slice at its end: clamped, unused)
                load_row(row1, cnext);
                vnext = load_valid(row1);
            }
            if constexpr (!SKEWED) encode_plain(ccur);
            unsigned long long vmask = ~0ull;
            if (slice_end - row0 < 64) vmask = (1ull << (int)(slice_end - row0)) - 1ull;
            if (a.valid) vmask &= __ballot((vcur >> (lane & 31)) & 1u);
            const uint32_t rid = (uint32_t)(row0 + lane);

            u32x2 acc = {0u, 0u};
            row_sums(ccur, acc);

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
            if (!TILES && a.gkey) {  // (tile mode: one slice per tile, nothing to import)
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
                                    // (the j-th key alone, the MAX over the slices: candidates are not what limits this kernel)
                                    const unsigned long long v = sibling_bound<1>(a.gk2, b, a.n_slices, g0, a.jm1, km1, lane);
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
            b_cur = b_nxt;
            b_nxt = (uint32_t)__builtin_amdgcn_readfirstlane((int)b_pend);
        }
        if (qcnt) qfilter_flush<M, SKEWED>(fc, queue_off + wave * 512, qcnt);

        __syncthreads();
        if constexpr (TILES) {
            if (tid < QT) ts.finish(tid, a.cand_count + tile * QT + tid);
            continue;
        }
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
            if (*s_flag) merge_tile_slices<NW>(a, tile * QT, QT, km1, wave, lane, (unsigned long long *)(smem + queue_off));
            __syncthreads();
        }
    }
}

}  // namespace annlite

using namespace annlite;

template <int M, int NQ, int NW, int WPS, bool SKEWED>
static int launch_qfilter(const ScanArgs &a, int grid, hipStream_t st) {
    const size_t lds_lut = (size_t)a.Ks * NQ * M * 16;
    const size_t need = lds_lut + 256 + (size_t)8 * NQ * 64 * 8 + 128 + (size_t)NW * 512 +
                        (a.tile_rows ? (size_t)kTileCandBytes + 64 : 0);
    auto fn = a.tile_rows ? adc_scan_qfilter_kernel<M, NQ, NW, WPS, SKEWED, true>
                          : adc_scan_qfilter_kernel<M, NQ, NW, WPS, SKEWED, false>;
    ANNLITE_HIP_TRY(hipFuncSetAttribute((const void *)fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)need));
    hipLaunchKernelGGL(fn, dim3(grid), dim3(NW * 64), need, st, a);
    return launch_status("adc_scan_qfilter_kernel");
}

template <int M, int NQ, int NW, int WPS>
static int launch_qfilter_code16(const ScanArgs &a, int grid, hipStream_t st) {
    const size_t lds_lut = (size_t)a.Ks * NQ * M * 16;
    const size_t need = lds_lut + 256 + (size_t)8 * NQ * 64 * 8 + 128 + (size_t)NW * 512;
    auto fn = adc_scan_qfilter_kernel<M, NQ, NW, WPS, false, false, true>;
    ANNLITE_HIP_TRY(hipFuncSetAttribute((const void *)fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)need));
    hipLaunchKernelGGL(fn, dim3(grid), dim3(NW * 64), need, st, a);
    return launch_status("adc_scan_qfilter_kernel (uint16 codes)");
}

template <int NW, bool SKEWED>
static int launch_qfilter64(const ScanArgs &a, int grid, hipStream_t st) {
    const size_t need = (size_t)(a.Ks + 1) * 512 + 256 + (size_t)4 * 512 + 128 + (size_t)NW * 512 +
                        (a.tile_rows ? (size_t)kTileCandBytes + 64 : 0);
    auto fn = a.tile_rows ? adc_scan_qfilter64_kernel<NW, SKEWED, true> : adc_scan_qfilter64_kernel<NW, SKEWED, false>;
    ANNLITE_HIP_TRY(hipFuncSetAttribute((const void *)fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)need));
    hipLaunchKernelGGL(fn, dim3(grid), dim3(NW * 64), need, st, a);
    return launch_status("adc_scan_qfilter64_kernel");
}

int annlite::launch_qfilter_scan(int id, bool sk, const ScanArgs &a, int grid, hipStream_t st) {
#define ANNLITE_LAUNCH_Q(MM, NQ_, NW_, WPS_) \
    (sk ? launch_qfilter<MM, NQ_, NW_, WPS_, true>(a, grid, st) : launch_qfilter<MM, NQ_, NW_, WPS_, false>(a, grid, st))
    switch (id) {
        case 830: return ANNLITE_LAUNCH_Q(8, 2, 16, 4);
        case 3230: return ANNLITE_LAUNCH_Q(32, 1, 12, 3);
        case 1630: return ANNLITE_LAUNCH_Q(16, 2, 12, 3);
        case 1631: return ANNLITE_LAUNCH_Q(16, 2, 16, 4);
        case 1632: return ANNLITE_LAUNCH_Q(16, 2, 8, 2);
        case 8217: return launch_qfilter_code16<8, 2, 16, 4>(a, grid, st);    // uint16 codes, Ks <= 512, 16 queries / WG
        case 8216: return launch_qfilter_code16<8, 1, 16, 4>(a, grid, st);    // uint16 codes, Ks <= 1024
        case 16216: return launch_qfilter_code16<16, 1, 16, 4>(a, grid, st);  // uint16 codes, Ks <= 512
        case 6430: return sk ? launch_qfilter64<16, true>(a, grid, st) : launch_qfilter64<16, false>(a, grid, st);
        case 6431: return sk ? launch_qfilter64<12, true>(a, grid, st) : launch_qfilter64<12, false>(a, grid, st);
        case 6432: return sk ? launch_qfilter64<8, true>(a, grid, st) : launch_qfilter64<8, false>(a, grid, st);
        default: set_error("no quantised-filter kernel with id %d", id); return ANNLITE_ERR_UNSUPPORTED;
    }
#undef ANNLITE_LAUNCH_Q
}
