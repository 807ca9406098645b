// scan_prep.hip -- what runs before the quantised-filter scan of a batch: quantisation of the fp32 tables
// (optionally building them in the same launch) and the seed bound.
#include "scan_common.h"

namespace annlite {

// quantisation of the fp32 TILED table [Bpad/4][Ks][64][4] for the M = 64 kernel: one workgroup per group of 4
// queries; entries [g4][Ks][64][4 x u16] (8 bytes)
__global__ __launch_bounds__(1024) void lut_quantise64_kernel(const float *__restrict__ lut, int Ks, int qmax,
                                                            uint16_t *__restrict__ out, float *__restrict__ qstep,
                                                            double *__restrict__ qlo, float *__restrict__ smax,
                                                            float *__restrict__ qlom, const unsigned int *__restrict__ gate) {
    if (gate && __hip_atomic_load(gate, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) return;  // (see ScanArgs::gate)
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
            mx[i] = fmaxf(mx[i], col_max_arg(v[i]));  // (over the entries below +inf: col_max_arg)
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
        if (qlom) qlom[(int64_t)(g4 * 4 + i) * M + mm] = l;  // (the byte-table scan quantises the tables itself)
    }
    __syncthreads();
    if (tid < 4) {
        float range = 0.f, sm = 0.f;
        double Lsum = 0.0;
        for (int mm = 0; mm < M; ++mm) {
            const float l = s_lo[0][mm][tid], h = s_hi[0][mm][tid];
            range = fmaxf(range, h - l);
            sm += fmaxf(finite_mag(l), finite_mag(h));  // (rounding slack of FINITE sums)
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
    if (!out) return;  // (byte-table plan: parameters only)
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
// staged in LDS, so ONE ds_read_b128 per (row, m) feeds the four sums (gathering the same entries from L2
// cost a 64-byte request per 4 useful bytes: 150 us for 4096 rows x 1024 queries).
// The sums run in the lane's SKEWED order -- sub-space (lane + t) mod M at step t, as the scan kernels read
// -- so that the 16 lanes of an LDS access group hit 16 different 16-byte bank groups whatever the codes
// are.  (In ascending m all lanes read the same sub-space and the bank group was (code + m) mod 16: random
// codes, ~2.7-way conflicts, 59 us for 32768 rows against 13 us of look-ups.)  A rotated fp32 sum differs
// from the reference's ascending one by rounding only, at most slack32 = 2 M 2^-24 sum_m max|lut| -- the
// scan's own margin --, which is added to the bound.
// Selection without sorting networks: every lane keeps the MIN distance of the rows it saw; the 1024
// (wave, lane) groups are disjoint, so the k-th smallest of their minima has >= k distinct rows at or
// below it -- a valid bound, and equal to the exact k-th distance of the S rows unless two of the k best
// rows fell into one lane (3 % at S=4096, k=10; then it is the (k+1)-th).  The k-th smallest is found by
// rank counting over LDS broadcasts (each lane counts the keys below its own): 64 keys per wave, then
// 16*k candidates per query.  (Sorted wave lists + a 4-level merge tree: 52 of 62 us in bitonic networks.)
// The seed rows are scanned again by the main kernel: only the bound leaves this kernel.
constexpr int kSeedWaves = 16;
// BUILD: the workgroup BUILDS the L2 tables of its 4 queries itself -- straight into the LDS image the seed scan reads, and
// into the fp32 TILED table in global memory for the scan kernel -- and leaves their quantisation parameters; it also resets
// the scan's result lists / shared bounds (fill).  One launch instead of lut_l2_build_quantise_kernel + this kernel: no
// second launch boundary (kernel-end write-back, dispatch: ~10 us) and no read-back of the 16 MB the first one wrote.
struct SeedBuild {
    const float *queries;  // [B][D]
    const float *cb;       // [M][Ks][dsub]
    float *lut_out;        // fp32 TILED [ceil16(B) / 4][Ks][M][4]
    float *qstep, *smax, *qlom;
    double *qlo;
    u32x4 *fill;           // reset to all-ones: [0, fill_vec16) without [skip_lo, skip_hi) (the bounds this kernel writes itself)
    int64_t fill_vec16, skip_lo, skip_hi;
    int32_t D, qmax;
    const unsigned int *gate;  // (any launch) run only if *gate == 0: the u16-table pass behind a byte-table launch that may give up
    // BUILD, optional: the byte tables of the scan kernel's work items, built HERE once per query (the workgroup holds its 4
    // queries' fp32 tables in LDS and knows their seed bound) instead of by each of the tile's 8 scan workgroups from L2
    unsigned long long *gseed0;  // [..] the seed key of every query, kept aside (gkey itself is lowered by the scan)
    uint8_t *btab;               // [n_tiles][kQ8Image16] (q8_entry16): this workgroup writes its queries' dword of every entry
    int32_t target;              // T of a freshly built table (ScanArgs::q8_target)
    int32_t chunk_log;           // (any launch) the seed rows come in runs of 2^chunk_log blocks of 64 rows (seed_chunk_log())
    unsigned long long *seedk;   // optional [B][kSeedKeys]: the bounds implied by the seed's k smallest rows, ascending -- what
                                 // the OTHER ranks of a row-sharded search may prune with (annlite_pq_search_split)
    unsigned long long *dbg;     // optional: wall-clock stamps (100 MHz) of workgroup 0 [0..3] and the last one [4..7]:
                                 // start, tables built, rows scanned, end
    // BUILD, optional (round 6): the rows an MFMA launch has NOMINATED for every query (seed_mfma.hip: the best row, by a bf16
    // approximation of the ADC sum, of each of n_cand disjoint groups of seed rows; 0xffffffff = none) -- the bound then comes from
    // THEIR exact sums (n_cand x 4 rows per workgroup) instead of the exact sums of all S seed rows
    const uint32_t *cand;        // [B][n_cand] table rows, distinct per query
    int32_t n_cand;
    // CELLS (annlite_ivf_search_topk: a query scans only the rows of its n_probe nearest cells -- a bound from rows OUTSIDE them
    // would be wrong): query b's seed rows are 64-row blocks spread evenly over its NEAREST cell, rows [cell_rows[2 c], cell_rows[2 c + 1])
    // of the cell-sorted table, c = cells[b * n_probe]; and the byte tables go out per QUERY (a cell tile's slots hold arbitrary queries)
    const int32_t *cells;        // [B][n_probe], nearest first
    const int32_t *seed_cells;   // optional [B]: the entry whose rows seed query b's bound instead of cells[b * n_probe] (a cell probed in PARTS:
                                 // the whole cell's range -- rows of the probed ranges all the same)
    const int64_t *cell_rows;    // [C][2] (begin: multiple of 64, end)
    int32_t n_probe;
    uint8_t *bq;                 // [ceil4(B)][Ks][M] byte tables quantised for gseed0 (q8_gather_table assembles a tile's image from them)
    unsigned int *item_counter;  // reset to 0 for the scan behind this launch (its workgroups draw their cell tiles from it)
    float ip_inv_ks;             // IP tables (template parameter): float32(1 / Ks), the constant of pq.py:316-322
};
// BUILD (round 6): the seed bound from the rows an MFMA launch has nominated (seed_mfma.hip) instead of from S seed rows this
// kernel scans itself.  Query p of the workgroup's 4 takes ceil(n_cand / 64) wave-iterations: lane = nominee; a row's sums run in
// the lane's skewed order like the seed rows' (the row is rotated to the lane's skew: it sits at an arbitrary table row), only
// query p's sum is used -- the nominees of ONE query are distinct rows, two queries may nominate the same row.  Two iterations
// per wave at 512 nominees: both row ids are requested first, then both code rows (two dependent round trips for the pair, not
// four).  Returns the lane's minimum per query (+inf: none).  The fp32 tables [Ks][M] x 4 queries sit at LDS address 0 (PERM
// addressing).  (inlined: as an out-of-line call it cost the kernel a scratch frame -- 156 B -- for registers saved around the call)
template <int M, bool SKEWED>
__device__ __forceinline__ f32x4 seed_nominee_minima(const uint8_t *codes, const uint32_t *valid, int64_t ext,
                                                               const uint32_t *cand, int n_cand, int g4, int B) {
    static_assert(M == 16, "the fused preparation launch");
    constexpr int CW = M / 4, CH = 16;
    const int lane = (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    uint32_t moff[M];
#pragma unroll
    for (int t = 0; t < M; ++t) moff[t] = (uint32_t)((lane + t) % M) << 4;
    f32x4 best = {__builtin_inff(), __builtin_inff(), __builtin_inff(), __builtin_inff()};
    const int CI = (n_cand + 63) >> 6, total = 4 * CI;
    for (int it0 = wave; it0 < total; it0 += 2 * kSeedWaves) {
        uint32_t rowv[2];
        int pq[2];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int it = it0 + u * kSeedWaves;
            pq[u] = it < total ? it / CI : -1;
            const int p = pq[u] < 0 ? 0 : pq[u];
            const int c = (it - p * CI) * 64 + lane;
            const int b = g4 * 4 + p;
            rowv[u] = 0xffffffffu;
            if (pq[u] >= 0 && c < n_cand && b < B) rowv[u] = cand[(int64_t)b * n_cand + c];
        }
        uint32_t cr[2][CW], vw[2];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int64_t rr = (int64_t)rowv[u] < ext ? (int64_t)rowv[u] : 0;
            const u32x4 v = *(const u32x4 *)(codes + rr * M);
            cr[u][0] = v.x, cr[u][1] = v.y, cr[u][2] = v.z, cr[u][3] = v.w;
            vw[u] = valid ? valid[rr >> 5] : ~0u;
        }
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            if (pq[u] < 0) continue;  // (wave-uniform)
            const uint32_t row = rowv[u];
            const bool ok = (int64_t)row < ext && ((vw[u] >> (row & 31)) & 1u);
            // the row rotated to this lane's skew: byte t <- the code of sub-space (lane + t) mod M
            const int rot = SKEWED ? (int)((uint32_t)(lane - (int)(row % M)) & (uint32_t)(M - 1)) : lane % M;
            bool ab[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) ab[i] = (((rot >> 2) >> i) & 1) != 0;
            rotate_row<CW>(cr[u], ab, (uint32_t)(rot & 3));
            f32x4 d = {0.f, 0.f, 0.f, 0.f};
            f32x4 v[CH];
            static_for<0, CH>([&](auto I) {
                constexpr int t = decltype(I)::value;
                typedef const f32x4 __attribute__((address_space(3))) *lds_fq_ptr;
                const uint32_t ad = __builtin_amdgcn_perm(cr[u][t / 4], moff[t], 0x0c0c0000u | ((4u + (uint32_t)(t % 4)) << 8));
                v[t] = *(lds_fq_ptr)(uintptr_t)ad;
            });
#pragma unroll
            for (int i = 0; i < CH; ++i) d += v[i];  // (the lane's skewed order, like the seed rows')
            const float mine = pq[u] == 0 ? d[0] : pq[u] == 1 ? d[1] : pq[u] == 2 ? d[2] : d[3];
            const float dv = ok ? mine : __builtin_inff();
            if (pq[u] == 0) best[0] = fminf(best[0], dv);
            else if (pq[u] == 1) best[1] = fminf(best[1], dv);
            else if (pq[u] == 2) best[2] = fminf(best[2], dv);
            else best[3] = fminf(best[3], dv);
        }
    }
    return best;
}

// LDS behind the fp32 tables: [0, 16 KB) the selection's candidates u32 [QPB][16 waves * k] (BUILD: first the queries, the
// per-wave column minima / maxima), [16 KB, 20 KB) the waves' 64 lane minima, then the block counter and what the BUILD
// variant keeps until its end (column minima, parameters, seed keys)
constexpr int kSeedWkeyOff = 16384, kSeedCtrOff = 20480, kSeedKeepOff = 20544, kSeedLdsExtra = 24576;
// QPB queries per workgroup: 4 (one fp32 TILED group) where their rows fit the LDS, 2 for M = 64
// CODE16: uint16 codes (PLAIN tables)
// IP (with CELLS): the tables are float32(1 / Ks) - <codeword, query> (batch_precompute_adc_table_ip's j-ascending fmaf chain,
// pq_bindings.pyx:214-274, and the subtraction of pq.py:316-322) instead of the squared-L2 ones
template <int M, bool SKEWED, int QPB, bool CODE16 = false, bool BUILD = false, bool CELLS = false, bool IP = false>
__global__ __launch_bounds__(kSeedWaves * 64) void seed_bound_kernel(const uint8_t *__restrict__ codes, int64_t S,
                                                                    const uint32_t *__restrict__ valid,
                                                                    const float *__restrict__ lut, int B, int Ks, int k,
                                                                    const float *__restrict__ smax,
                                                                    unsigned long long *__restrict__ gkey,
                                                                    int64_t seed_stride, int64_t N, int gkey_stride,
                                                                    const SeedBuild sb) {
    static_assert(!(CODE16 && SKEWED), "uint16 code tables are PLAIN");
    static_assert(!BUILD || (QPB == 4 && !CODE16 && M <= 16), "the fused build serves the byte-table plan");
    static_assert(!CELLS || (BUILD && M == 16), "per-cell seeds: the fused build of the M = 16 byte-table plan");
    static_assert(!IP || CELLS, "inner-product tables are built here for the pruned search only");
    if (sb.gate && __hip_atomic_load(sb.gate, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) return;
    if constexpr (CELLS) {
        if (sb.item_counter && blockIdx.x == 0 && threadIdx.x == 0) *sb.item_counter = 0u;
    }
    auto stamp = [&](int i) {
        if constexpr (BUILD) {
            if (sb.dbg && threadIdx.x == 0 && blockIdx.y == 0 && (blockIdx.x == 0 || blockIdx.x == gridDim.x - 1))
                sb.dbg[(blockIdx.x == 0 ? 0 : 4) + i] = wall_clock64();
        }
    };
    stamp(0);
    // blockIdx.y = row slice (per-slice bounds of the candidate generator: S rows of [y * seed_stride, + seed_stride), bound ->
    // gkey[y * gkey_stride + b]; seed_stride is a multiple of 64, so the lanes keep their skew residues); one slice: S rows of
    // the table.  The S rows are ceil(S / 64) blocks of 64 rows in runs of 2^chunk_log blocks spread EVENLY over the extent (block j =
    // rows [(j >> chunk_log) * run_step + (j mod 2^chunk_log) * 64, + 64)): a table filled in cluster order has an
    // unrepresentative head, and a bound from it alone leaves the byte tables coarse for the whole scan (1.25M rows ordered along
    // one latent direction: 0.645 ms per 1024-query batch seeded from the head, 0.330 from spread rows; 10M rows 3.11 / 1.44 --
    // profiles/r05/seed_rows_spread.txt).  Any subset of the valid rows gives a correct bound.
    int64_t ext;
    {
        const int64_t base = (int64_t)blockIdx.y * seed_stride;
        ext = N - base;
        if (seed_stride > 0 && seed_stride < ext) ext = seed_stride;
        if (S > ext) S = ext;
        if (S <= 0) return;
        codes += base * M * (CODE16 ? 2 : 1);
        if (valid) valid += base >> 5;
        gkey += (int64_t)blockIdx.y * gkey_stride;
    }
    constexpr int CW = CODE16 ? M / 2 : M / 4;
    constexpr int CH = M < 16 ? M : 16;  // look-ups in flight
    constexpr bool PERM = M == 16 && QPB == 4 && !CODE16;  // one byte permute per look-up address (see below)
    typedef float fq __attribute__((ext_vector_type(QPB)));
    static_assert(QPB == 4 || QPB == 2, "queries per block");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    // [Ks][M] x QPB queries: entry (code, m) sits in bank group m mod 16 (QPB = 4; bank pair m mod 32 for QPB = 2)
    fq *tab = (fq *)smem;
    unsigned long long *cand = (unsigned long long *)(smem + (size_t)Ks * M * sizeof(fq));  // [kSeedWaves][k]
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    constexpr int BPG = 4 / QPB;                 // blocks per fp32 TILED group of 4 queries
    const int g4 = blockIdx.x / BPG, h = blockIdx.x % BPG;
    float smax_built[4] = {0.f, 0.f, 0.f, 0.f};  // (BUILD: the rounding slack of this workgroup's queries)
    if constexpr (BUILD) {
        // ---- reset of the scan's lists / bounds (all but the bounds this kernel writes) ----------------------------------
        for (int64_t i = (int64_t)blockIdx.x * (kSeedWaves * 64) + tid; i < sb.fill_vec16; i += (int64_t)gridDim.x * (kSeedWaves * 64))
            if (i < sb.skip_lo || i >= sb.skip_hi) sb.fill[i] = (u32x4){~0u, ~0u, ~0u, ~0u};
        // ---- the tables: thread (kr, m) = (tid / M, tid % M) owns sub-space m of codes kr, kr + 1024 / M, ...; entry = the
        // reference's j-ascending fmaf chain over (codeword - query) (pq_bindings.pyx:204-206): lut_l2_tiled_kernel's bits --------
        constexpr int KPT = kSeedWaves * 64 / M, NSW = 256 / KPT;
        float *s_q = (float *)cand;                                  // [4][D]
        float *s_lo = (float *)((unsigned char *)cand + 4096);       // [16 waves][M][4]
        float *s_hi = s_lo + kSeedWaves * M * 4;
        float *s_par = (float *)((unsigned char *)cand + kSeedKeepOff);  // [4] smax, [4] step; then L f64 [4], seed keys u64 [4], lo f32 [M][4]
        const int D = sb.D, dsub = D / M;
        for (int i = tid; i < 4 * D; i += kSeedWaves * 64) {
            const int b = g4 * 4 + i / D;
            s_q[i] = b < B ? sb.queries[(int64_t)b * D + i % D] : 0.f;
        }
        __syncthreads();
        const int m = tid % M, kr = tid / M;
        f32x4 *gout = (f32x4 *)sb.lut_out + (int64_t)g4 * Ks * M;
        float mn[4], mx[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) mn[i] = __builtin_inff(), mx[i] = -__builtin_inff();
#pragma unroll
        for (int sw = 0; sw < NSW; ++sw) {
            const int kk = kr + sw * KPT;
            if (kk < Ks) {
                float acc[4] = {0.f, 0.f, 0.f, 0.f};
                const float *cw = sb.cb + ((int64_t)m * Ks + kk) * dsub;
                for (int j = 0; j < dsub; j += 4) {
                    const f32x4 cj = *(const f32x4 *)(cw + j);
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const f32x4 qj = *(const f32x4 *)(s_q + i * D + m * dsub + j);
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            if constexpr (IP) {
                                acc[i] = __builtin_fmaf(cj[e], qj[e], acc[i]);
                            } else {
                                const float c = cj[e] - qj[e];
                                acc[i] = __builtin_fmaf(c, c, acc[i]);
                            }
                        }
                    }
                }
                if constexpr (IP) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) acc[i] = sb.ip_inv_ks - acc[i];
                }
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    if (g4 * 4 + i >= B) acc[i] = 0.f;  // pad queries -> 0, like lut_l2_tiled_kernel
                const f32x4 v = {acc[0], acc[1], acc[2], acc[3]};
                tab[kk * M + m] = v;
                if constexpr (CELLS) {
                    // per QUERY, [b][Ks][M]: the cell tiles' consumer gathers 16 entries of ONE query per exact sum from far memory (the
                    // batch's tables do not fit an XCD's L2 and a tile's 32 slots hold arbitrary queries) -- a query's 16 KB on its own take
                    // 128 cache lines where the TILED groups of four spread them over 512
                    float *gq = sb.lut_out + ((int64_t)(g4 * 4) * Ks + kk) * M + m;
#pragma unroll
                    for (int i = 0; i < 4; ++i) gq[(int64_t)i * Ks * M] = acc[i];
                } else
                gout[(int64_t)kk * M + m] = v;
#pragma unroll
                for (int i = 0; i < 4; ++i) mn[i] = fminf(mn[i], acc[i]), mx[i] = fmaxf(mx[i], col_max_arg(acc[i]));
            }
        }
        // lanes l, l + M, ... of a wave share m
#pragma unroll
        for (int o = M; o < 64; o <<= 1)
#pragma unroll
            for (int i = 0; i < 4; ++i) mn[i] = fminf(mn[i], __shfl_xor(mn[i], o)), mx[i] = fmaxf(mx[i], __shfl_xor(mx[i], o));
        if (lane < M)
#pragma unroll
            for (int i = 0; i < 4; ++i) s_lo[(wave * M + m) * 4 + i] = mn[i], s_hi[(wave * M + m) * 4 + i] = mx[i];
        __syncthreads();
        if (tid < M * 4) {
            const int mm = tid / 4, i = tid % 4;
            float l = s_lo[mm * 4 + i], hh = s_hi[mm * 4 + i];
            for (int r = 1; r < kSeedWaves; ++r) l = fminf(l, s_lo[(r * M + mm) * 4 + i]), hh = fmaxf(hh, s_hi[(r * M + mm) * 4 + i]);
            s_lo[mm * 4 + i] = l;
            s_hi[mm * 4 + i] = hh;
            sb.qlom[(int64_t)(g4 * 4 + i) * M + mm] = l;
            s_par[24 + mm * 4 + i] = l;  // (kept: the byte tables are quantised after the selection has reused s_lo)
        }
        __syncthreads();
        if (tid < 4) {
            float range = 0.f, sm = 0.f;
            double Lsum = 0.0;
            for (int mm = 0; mm < M; ++mm) {
                const float l = s_lo[mm * 4 + tid], hh = s_hi[mm * 4 + tid];
                range = fmaxf(range, hh - l);
                sm += fmaxf(finite_mag(l), finite_mag(hh));  // (rounding slack of FINITE sums)
                Lsum += (double)l;
            }
            float step = range / (float)sb.qmax;
            if (!(step > 0.f)) step = 1.f;
            const int b = g4 * 4 + tid;
            sb.qstep[b] = step;
            sb.qlo[b] = Lsum;
            sb.smax[b] = sm;
            s_par[tid] = sm;
            s_par[4 + tid] = step;
            ((double *)(s_par + 8))[tid] = Lsum;
            ((unsigned long long *)(s_par + 16))[tid] = ~0ull;
            gkey[b] = ~0ull;  // (the fill leaves the bounds to this kernel; the selection below overwrites it)
            if (sb.seedk)
                for (int j = 0; j < kSeedKeys; ++j) sb.seedk[(int64_t)b * kSeedKeys + j] = ~0ull;
            if constexpr (CELLS) {  // the query's nearest cell: [begin, end) and its blocks of 64 rows (behind the kept parameters)
                int32_t *s_cell = (int32_t *)(s_par + 96);  // [4] begin, [4] end, [4] blocks
                int64_t cb = 0, ce = 0;
                if (b < B) {
                    const int32_t c = sb.seed_cells ? sb.seed_cells[b] : sb.cells[(int64_t)b * sb.n_probe];
                    cb = sb.cell_rows[2 * (int64_t)c], ce = sb.cell_rows[2 * (int64_t)c + 1];
                }
                s_cell[tid] = (int32_t)cb;
                s_cell[4 + tid] = (int32_t)ce;
                s_cell[8 + tid] = (int32_t)((ce - cb + 63) >> 6);
            }
        }
        __syncthreads();
#pragma unroll
        for (int i = 0; i < 4; ++i) smax_built[i] = s_par[i];
        __syncthreads();  // (cand is reused by the selection)
        stamp(1);
    } else {
        const f32x4 *src = (const f32x4 *)lut + (int64_t)g4 * Ks * M;
        for (int i = tid; i < Ks * M; i += kSeedWaves * 64) {
            const f32x4 e = src[i];
            if constexpr (QPB == 4) tab[i] = e;
            else tab[i] = h ? (fq){e.z, e.w} : (fq){e.x, e.y};
        }
    }
    // the waves draw their blocks of 64 rows from this counter (see adc_scan_q8_kernel: a static deal leaves the waves the
    // SIMD's arbiter does not favour to finish alone, at a third of the issue rate)
    uint32_t *blk_ctr = (uint32_t *)((unsigned char *)cand + kSeedCtrOff);
    if (tid == 0) *blk_ctr = 0;
    __syncthreads();
    // forward skew rotation of PLAIN rows (row % M == lane % M: the rows of a wave start at a multiple of 64); SKEWED
    // rows are stored that way
    const int sfw = (lane % M) * (CODE16 ? 2 : 1);  // bytes
    const uint32_t bsh_fw = (uint32_t)(sfw & 3);
    bool abit_fw[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) abit_fw[i] = (((sfw >> 2) >> i) & 1) != 0;
    // sub-space of stored byte t of this lane's rows, as an element offset into a table row
    uint32_t moff[M];
#pragma unroll
    for (int t = 0; t < M; ++t) {
        if constexpr (M == 64) moff[t] = (uint32_t)(32 * (t / 32) + ((lane & 31) + t) % 32);  // two skewed halves
        else moff[t] = (uint32_t)((lane + t) % M);
        if constexpr (PERM) moff[t] <<= 4;  // (as a byte offset: the permute addressing below)
    }
    if constexpr (PERM) {
        if ((uint32_t)(uintptr_t)(__attribute__((address_space(3))) unsigned char *)smem != 0u) __builtin_trap();  // (all LDS is dynamic)
    }

    // per-lane minima of the rows' sums, as floats: v_min_f32 drops a NaN sum (a row whose sum is NaN -- a NaN / inf query
    // coordinate -- sorts behind every number and never sets the bound; as a raw key a NaN with the sign bit set would have
    // been the SMALLEST); the keys are formed after the loop
    float bestf[QPB];
#pragma unroll
    for (int q = 0; q < QPB; ++q) bestf[q] = __builtin_inff();
    // the code bytes (and validity word) of a lane's NEXT row are fetched while the current one is summed: the loop was
    // bound by one dependent global round trip per iteration (12.7 us per 8192 rows; the look-ups need ~4)
    auto fetch = [&](int64_t r, uint32_t (&cc)[CW], uint32_t &vw) {
        const int64_t rr = r < ext ? r : ext - 1;
        const uint32_t *p = (const uint32_t *)(codes + rr * M * (CODE16 ? 2 : 1));
#pragma unroll
        for (int i = 0; i < CW; ++i) cc[i] = p[i];
        vw = valid ? valid[rr >> 5] : ~0u;
    };
    uint32_t cn[CW], vn;
    auto draw_block = [&]() -> uint32_t {  // (lane 0's value; broadcast a step later)
        uint32_t v = 0;
        if (lane == 0) v = atomicAdd(blk_ctr, 1u);
        return v;
    };
    const uint32_t n_blocks = (uint32_t)((S + 63) >> 6);
    const int clog = sb.chunk_log;
    const uint32_t cmask = (1u << clog) - 1u;
    int64_t run_step = ((ext >> 6) / ((n_blocks + cmask) >> clog)) << 6;  // rows between the starts of two runs of seed blocks
    if (run_step < ((int64_t)64 << clog)) run_step = (int64_t)64 << clog;
    // CELLS: seed block b belongs to query b & 3 and is block (b >> 2) * stride of that query's nearest cell (stride: the cell's
    // blocks over the query's share n_blocks / 4 of the seed blocks, at least 1; past the cell's end: no rows)
    int32_t c_begin[4] = {0, 0, 0, 0}, c_end[4] = {0, 0, 0, 0}, c_stride[4] = {1, 1, 1, 1};
    if constexpr (CELLS) {
        const int32_t *s_cell = (const int32_t *)((const float *)((unsigned char *)cand + kSeedKeepOff) + 96);
        const int32_t per_q = (int32_t)(n_blocks >> 2) > 0 ? (int32_t)(n_blocks >> 2) : 1;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            c_begin[i] = __builtin_amdgcn_readfirstlane(s_cell[i]);
            c_end[i] = __builtin_amdgcn_readfirstlane(s_cell[4 + i]);
            const int32_t nb = __builtin_amdgcn_readfirstlane(s_cell[8 + i]);
            c_stride[i] = nb / per_q > 1 ? nb / per_q : 1;
        }
    }
    auto block_row = [&](uint32_t b) -> int64_t {
        if constexpr (CELLS) {
            const int i = (int)(b & 3u);
            const int32_t cb = i == 0 ? c_begin[0] : i == 1 ? c_begin[1] : i == 2 ? c_begin[2] : c_begin[3];
            const int32_t st = i == 0 ? c_stride[0] : i == 1 ? c_stride[1] : i == 2 ? c_stride[2] : c_stride[3];
            return (int64_t)cb + (((int64_t)(b >> 2) * st) << 6);
        } else
        return (int64_t)(b >> clog) * run_step + (int64_t)((b & cmask) << 6);
    };
    bool nominated = false;
    if constexpr (BUILD) nominated = sb.cand != nullptr;
    if constexpr (BUILD && PERM) {
        if (nominated) {  // the bound from the exact sums of the rows an MFMA launch nominated
            const f32x4 nb = seed_nominee_minima<M, SKEWED>(codes, valid, ext, sb.cand, sb.n_cand, g4, B);
#pragma unroll
            for (int q = 0; q < QPB; ++q) bestf[q] = nb[q];
        }
    }
    uint32_t b_cur = nominated ? n_blocks : (uint32_t)__builtin_amdgcn_readfirstlane((int)draw_block());
    uint32_t b_pend = nominated ? n_blocks : draw_block();
    if (!nominated) fetch(block_row(b_cur) + lane, cn, vn);
    uint32_t b_nxt = nominated ? n_blocks : (uint32_t)__builtin_amdgcn_readfirstlane((int)b_pend);
    while (b_cur < n_blocks) {
        const int64_t r = block_row(b_cur) + lane;
        b_pend = draw_block();
        bool ok = r < ext && ((vn >> (r & 31)) & 1u);
        if constexpr (CELLS) {  // (rows past the end of the block's cell belong to another cell or are padding)
            const int i = (int)(b_cur & 3u);
            const int32_t ce = i == 0 ? c_end[0] : i == 1 ? c_end[1] : i == 2 ? c_end[2] : c_end[3];
            ok = ok && r < (int64_t)ce;
        }
        uint32_t c[CW];
#pragma unroll
        for (int i = 0; i < CW; ++i) c[i] = cn[i];
        fetch(block_row(b_nxt) + lane, cn, vn);
        if constexpr (M == 64) {
            if constexpr (SKEWED) {
#pragma unroll
                for (int i = 0; i < CW; ++i) c[i] = bytes_add(c[i], wrap64_mask(i, lane & 31));  // undo the wrap coding only
            } else {
                skew64_rotate_halves(c, lane & 31);
            }
        } else if constexpr (!SKEWED) {
            rotate_row<CW>(c, abit_fw, bsh_fw);
        }
        fq d;
#pragma unroll
        for (int q = 0; q < QPB; ++q) d[q] = 0.f;
        static_for<0, M / CH>([&](auto C) {
            constexpr int t0 = decltype(C)::value * CH;
            fq v[CH];
            static_for<0, CH>([&](auto I) {
                constexpr int i = decltype(I)::value, t = t0 + i;
                if constexpr (PERM) {
                    // 256-byte table rows: the LDS address (code << 8) | (sub-space << 4) is one byte permute of the code
                    // dword with the step's constant (byte 0 <- moff16 byte 0, byte 1 <- code byte t % 4, rest 0)
                    typedef const fq __attribute__((address_space(3))) *lds_fq_ptr;
                    const uint32_t ad = __builtin_amdgcn_perm(c[t / 4], moff[t], 0x0c0c0000u | ((4u + (uint32_t)(t % 4)) << 8));
                    v[i] = *(lds_fq_ptr)(uintptr_t)ad;
                } else {
                    const uint32_t code = CODE16 ? (c[t / 2] >> (16 * (t % 2))) & 0xffffu : (c[t / 4] >> (8 * (t % 4))) & 0xffu;
                    v[i] = tab[code * M + moff[t]];
                }
            });
#pragma unroll
            for (int i = 0; i < CH; ++i) d += v[i];  // (the lane's skewed order: see the kernel's header)
        });
        if constexpr (CELLS) {
            // the block's rows are seed rows of ONE of the four queries.  (Evaluating only that query's sum -- 16 four-byte look-ups, 16
            // adds, the query's column as the instruction's offset, one copy of the look-ups per query behind a scalar branch -- was
            // SLOWER: 81.9 against ~69 us per launch at 65536 rows; four lanes of a wave share a sub-space: 4-way bank conflicts)
            const int i = (int)(b_cur & 3u);
#pragma unroll
            for (int q = 0; q < QPB; ++q) bestf[q] = fminf(bestf[q], (ok && q == i) ? d[q] : __builtin_inff());
        } else {
#pragma unroll
        for (int q = 0; q < QPB; ++q) bestf[q] = fminf(bestf[q], ok ? d[q] : __builtin_inff());
        }
        b_cur = b_nxt;
        b_nxt = (uint32_t)__builtin_amdgcn_readfirstlane((int)b_pend);
    }
    uint32_t best[QPB];  // ordered distance keys (a lane that saw no valid row: key(+inf), as before 0xffffffff | ... below)
#pragma unroll
    for (int q = 0; q < QPB; ++q) best[q] = bestf[q] < __builtin_inff() ? f32_to_ordered(bestf[q]) : 0xffffffffu;
    stamp(2);
    // Selection keys: the low 10 bits of the ordered distance are replaced by (wave, lane), which makes the
    // 1024 keys of a query unique (rank = number of smaller keys, no tie handling) and costs at most 1023
    // ulps of tightness: the k smallest keys T_i bound k distinct rows by (T_i | 1023).
    // All QPB queries go through the two ranking passes TOGETHER: a wave ranks its 64 lane minima query after query (its own
    // LDS traffic is in order: no barrier), ONE barrier, then thread t ranks candidate t % nc of query t / nc.  (One query
    // at a time -- two barriers and a 160-step serial ranking loop each -- was ~5 us of the launch.)
    const int nc = kSeedWaves * k;  // candidates per query
    uint32_t *cand32 = (uint32_t *)cand;                                                        // [QPB][kSeedWaves * k]
    uint32_t *wkeys = (uint32_t *)((unsigned char *)cand + kSeedWkeyOff) + wave * 64;          // this wave's 64 lane minima
#pragma unroll
    for (int q = 0; q < QPB; ++q) {
        const uint32_t mine = (best[q] & ~1023u) | (uint32_t)(wave << 6) | (uint32_t)lane;
        // rank among the wave's 64: all lanes read the 64 keys back with wave-uniform addresses (broadcast)
        wkeys[lane] = mine;
        int rank = 0;
#pragma unroll
        for (int j = 0; j < 64; j += 4) {
            const u32x4 o = *(const u32x4 *)(wkeys + j);
            rank += (o.x < mine) + (o.y < mine) + (o.z < mine) + (o.w < mine);
        }
        if (rank < k) cand32[q * nc + wave * k + rank] = mine;
        asm volatile("" ::: "memory");  // (the next query's keys go into the same 64 slots: after these reads)
    }
    __syncthreads();
    // k > 16 (the byte-table kernel's 64-key lists, round 5): ranking all 16 k candidates against each other is 16 k x 16 k
    // comparisons per query -- 48 us of the launch at k = 50, VALU-bound.  One level in between: the candidates of FOUR waves (4 k,
    // contiguous) are ranked among themselves and their k smallest go on (the k smallest of all are among them): 4 x (4 k)^2 +
    // (4 k)^2 comparisons instead of (16 k)^2.  (The waves' key slots behind the candidates are free after the barrier.)
    const uint32_t *sel = cand32;
    int nsel = nc;
    if (k > 16) {
        uint32_t *st2 = (uint32_t *)((unsigned char *)cand + kSeedWkeyOff);  // [QPB][4 groups][k]
        const int n4 = 4 * k;
        for (int t = tid; t < QPB * nc; t += kSeedWaves * 64) {
            const int q = t / nc, idx = t - q * nc, g = idx / n4;
            const uint32_t *cg = cand32 + q * nc + g * n4;
            const uint32_t me = cand32[t];
            int rk = 0;
#pragma unroll 4
            for (int j = 0; j < n4; j += 4) {
                const u32x4 o = *(const u32x4 *)(cg + j);
                rk += (o.x < me) + (o.y < me) + (o.z < me) + (o.w < me);
            }
            if (rk < k) st2[(q * 4 + g) * k + rk] = me;
        }
        __syncthreads();
        sel = st2;
        nsel = n4;
    }
    // the k-th smallest of every query's candidates (16 k of them, or the 4 k that came through the level above), same way
    for (int t = tid; t < QPB * nsel; t += kSeedWaves * 64) {
        const int q = t / nsel;
        const uint32_t *cq = sel + q * nsel;
        const uint32_t me = cq[t - q * nsel];
        int rk = 0;
        // (16-byte reads: the same address in every lane of a wave -- a broadcast)
#pragma unroll 4
        for (int j = 0; j < nsel; j += 4) {
            const u32x4 o = *(const u32x4 *)(cq + j);
            rk += (o.x < me) + (o.y < me) + (o.z < me) + (o.w < me);
        }
        const int b = g4 * 4 + h * QPB + q;
        bool publish = false;
        if constexpr (BUILD) publish = sb.seedk != nullptr && rk < k;  // (the k smallest, for the peers' union)
        // the bound admits every row at or below (me | 1023), whatever its id; +1 in the distance field
        // because the seed rows are in nobody's list: the scan must still ACCEPT the rows that set it
        if ((rk == k - 1 || publish) && b < B && (me | 1023u) != 0xffffffffu) {
            // + the rounding margin between this sum order and the reference's, rounded up
            float sm_b;
            if constexpr (BUILD) sm_b = q == 0 ? smax_built[0] : q == 1 ? smax_built[1] : q == 2 ? smax_built[2] : smax_built[3];
            else sm_b = smax[b];
            const float slack = sm_b * (float)(2.0 * M * 5.9604644775390625e-08 * (1.0 + 1.0 / 1024.0));
            float thr = ordered_to_f32(me | 1023u);
            thr = thr + slack;
            const uint32_t key = f32_to_ordered(thr) + 1u;
            const unsigned long long bound = ((unsigned long long)key + 1ull) << 32;
            if constexpr (BUILD) {
                if (publish) sb.seedk[(int64_t)b * kSeedKeys + rk] = bound;
            }
            if (rk == k - 1) {
                gkey[b] = bound;
                if constexpr (BUILD) ((unsigned long long *)((float *)((unsigned char *)cand + kSeedKeepOff) + 16))[q] = bound;
            }
        }
    }
    if constexpr (BUILD) {
        if (CELLS ? sb.bq != nullptr : sb.btab != nullptr) {
            // ---- the scan kernel's byte tables for these 4 queries, quantised for their seed bound: exactly what the scan
            // workgroups would build (q8_slot_params from the seed key, q8_build_table's conversion) -- built ONCE here, from
            // the fp32 tables still in LDS, instead of by the 8 workgroups of the tile from L2.  Query b = g4 * 4 + i sits in
            // tile b / 32, entry group (b % 32) / 16, byte b % 16 of an entry: this workgroup owns dword (b % 16) / 4. ----------
            __syncthreads();  // (the seed keys are in; the selection's LDS is dead)
            float *par = (float *)((unsigned char *)cand + kSeedKeepOff);
            float *s_inv = (float *)cand, *s_clip = s_inv + 4;
            if (tid < 4) {
                const int b = g4 * 4 + tid;
                const bool real = b < B;
                const unsigned long long key0 = ((const unsigned long long *)(par + 16))[tid];
                float step, inv, clip;
                uint32_t tb;
                q8_slot_params<M>(real, key0, real ? par[4 + tid] * (float)(32767 / M) : 0.f, real ? par[tid] : 0.f,
                                  real ? ((const double *)(par + 8))[tid] : 0.0, sb.target, step, inv, clip, tb);
                s_inv[tid] = real ? inv : 0.f;
                s_clip[tid] = clip;
                sb.gseed0[b] = key0;
            }
            __syncthreads();
            constexpr int KPT = kSeedWaves * 64 / M, NSW = 256 / KPT;
            const int m = tid % M, kr = tid / M;
            const int q0 = (g4 * 4) & 31;
            // (the tile's LDS image, q8_entry16: two half tables by sub-space parity, the two entry groups of a (code, sub-space) adjacent)
            uint8_t *img = sb.btab + (int64_t)((g4 * 4) >> 5) * (int64_t)kQ8Image16 + 4 * ((q0 & 15) >> 2);
            const uint32_t grp = (uint32_t)(q0 >> 4);
            float lo_r[4], inv_r[4], clip_r[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) lo_r[i] = par[24 + m * 4 + i], inv_r[i] = s_inv[i], clip_r[i] = s_clip[i];
#pragma unroll
            for (int sw = 0; sw < NSW; ++sw) {
                const int kk = kr + sw * KPT;
                if (kk < Ks) {
                    const fq v = tab[kk * M + m];
                    uint32_t pk = 0;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        float t = __builtin_fmaf(v[e] - lo_r[e], inv_r[e], -0.5f);
                        t = __builtin_fminf(t, clip_r[e]);
                        pk = __builtin_amdgcn_cvt_pk_u8_f32(t, e, pk);  // saturates below 0 (q8_build_table's conversion)
                    }
                    if constexpr (CELLS) {  // per query: bq[b][kk][m]
                        uint8_t *dst = sb.bq + ((int64_t)(g4 * 4) * Ks + kk) * M + m;
#pragma unroll
                        for (int e = 0; e < 4; ++e) dst[(int64_t)e * Ks * M] = (uint8_t)(pk >> (8 * e));
                    } else
                    *(uint32_t *)(img + q8_entry16((uint32_t)kk, (uint32_t)m, grp)) = pk;
                }
            }
        }
    }
    stamp(3);
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
            mx[i] = fmaxf(mx[i], col_max_arg(v[i]));  // (over the entries below +inf: col_max_arg)
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
        sm += fmaxf(finite_mag(l), finite_mag(h));  // (rounding slack of FINITE sums)
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
                                                                float *__restrict__ smax, float *__restrict__ qlom,
                                                                u32x4 *__restrict__ fill, int64_t fill_vec16,
                                                                const unsigned int *__restrict__ gate) {
    if (gate && __hip_atomic_load(gate, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) return;  // (see ScanArgs::gate)
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
            mx[i] = fmaxf(mx[i], col_max_arg(v0[i]));
            mn[4 + i] = fminf(mn[4 + i], v1[i]);
            mx[4 + i] = fmaxf(mx[4 + i], col_max_arg(v1[i]));
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
        if (qlom) qlom[(int64_t)(g8 * 8 + i) * M + mm] = l;
    }
    __syncthreads();
    if (tid < 8) {
        float range = 0.f, sm = 0.f;
        double Lsum = 0.0;
        for (int mm = 0; mm < M; ++mm) {
            const float l = s_lo[0][mm][tid], h = s_hi[0][mm][tid];
            range = fmaxf(range, h - l);
            sm += fmaxf(finite_mag(l), finite_mag(h));  // (rounding slack of FINITE sums)
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
    if (!out) return;  // (the byte-table scan quantises the tables itself)
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
                                                                    float *__restrict__ smax, float *__restrict__ qlom,
                                                                    u32x4 *__restrict__ fill, int64_t fill_vec16) {
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
            if (lut) {  // (tile mode never reads the fp32 tables of its slots: NULL)
                base0[(int64_t)k * M + m] = v0[sw];
                base1[(int64_t)k * M + m] = v1[sw];
            }
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                mn[i] = fminf(mn[i], acc[i]);
                mx[i] = fmaxf(mx[i], col_max_arg(acc[i]));
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
        if (qlom) qlom[(int64_t)(g8 * 8 + i) * M + mm] = l;
    }
    __syncthreads();
    if (tid < 8) {
        float range = 0.f, sm = 0.f;
        double Lsum = 0.0;
        for (int mm = 0; mm < M; ++mm) {
            const float l = s_lo[0][mm][tid], h = s_hi[0][mm][tid];
            range = fmaxf(range, h - l);
            sm += fmaxf(finite_mag(l), finite_mag(h));  // (rounding slack of FINITE sums)
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
    if (!out) return;  // (the byte-table scan quantises the tables itself)
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

}  // namespace annlite

using namespace annlite;

int annlite::launch_lut_quantise(int64_t M, int64_t Ks, int64_t B, int64_t bpad, const float *lut_dev, const LutBuild *build,
                                 uint16_t *q16, float *qstep, double *qlo, float *smax, float *qlom, void *fill,
                                 size_t fill_bytes, hipStream_t st, const unsigned int *gate) {
    const int qmax = (int)(32767 / M);
    const unsigned n_g8 = (unsigned)(bpad / 8);
    float *lut_rw = const_cast<float *>(lut_dev);
    u32x4 *fillp = (u32x4 *)fill;
    const int64_t fillv = (int64_t)(fill_bytes / 16);
#define ANNLITE_QUANT(MM)                                                                                            \
    if (build)                                                                                                       \
        hipLaunchKernelGGL((lut_l2_build_quantise_kernel<MM>), dim3(n_g8), dim3(1024), (size_t)(8 * build->D * 4), st,  \
                           build->queries, (int)B, (int)build->D, build->codebooks, (int)Ks, lut_rw, qmax, q16, qstep,   \
                           qlo, smax, qlom, fillp, fillv);                                                                 \
    else                                                                                                             \
        hipLaunchKernelGGL((lut_quantise_fused_kernel<MM>), dim3(n_g8), dim3(256), 0, st, lut_rw, (int)Ks, qmax, q16,  \
                           qstep, qlo, smax, qlom, fillp, fillv, gate)
    if (M == 64)  // (no fill, no build: the caller memsets and builds the tables itself)
        hipLaunchKernelGGL(lut_quantise64_kernel, dim3((unsigned)(bpad / 4)), dim3(1024), 0, st, lut_dev, (int)Ks, qmax, q16,
                           qstep, qlo, smax, qlom, gate);
    else if (M == 8) { ANNLITE_QUANT(8); } else if (M == 16) { ANNLITE_QUANT(16); } else { ANNLITE_QUANT(32); }
#undef ANNLITE_QUANT
    return launch_status("lut_quantise_fused_kernel");
}

// runs of 8 blocks (512 rows: 8 KB of 16-byte code rows) keep the seed launch's scattered reads to a page per run
// (ANNLITE_SEED_CHUNK_LOG=0..6: measurements)
static int seed_chunk_log() { return knobs().seed_chunk_log; }

int annlite::launch_seed_bound(int64_t M, bool skw, const void *codes_dev, int code_bytes, int64_t S, const uint32_t *valid_bits_dev,
                               const float *lut_dev, int64_t B, int64_t Ks, int64_t k, const float *smax,
                               unsigned long long *gk, hipStream_t st, int64_t N, int n_seed_slices, int64_t seed_stride,
                               int64_t gkey_stride, const unsigned int *gate) {
    SeedBuild nob = {};
    nob.gate = gate;
    nob.chunk_log = seed_chunk_log();
    const unsigned ny = (unsigned)(n_seed_slices > 0 ? n_seed_slices : 1);
    const int gstride = (int)gkey_stride;
    if (N <= 0) N = S;
#define ANNLITE_SEED(MM, QPB_)                                                                                    \
    {                                                                                                             \
        auto fn = skw ? seed_bound_kernel<MM, true, QPB_> : seed_bound_kernel<MM, false, QPB_>;                   \
        const size_t lds = (size_t)Ks * MM * 4 * QPB_ + (size_t)kSeedLdsExtra; /* cand + minima + counter */         \
        ANNLITE_HIP_TRY(hipFuncSetAttribute((const void *)fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); \
        hipLaunchKernelGGL(fn, dim3((unsigned)(((B + 3) / 4) * (4 / QPB_)), ny), dim3(kSeedWaves * 64), lds, st,  \
                           (const uint8_t *)codes_dev, S, valid_bits_dev, lut_dev, (int)B, (int)Ks, (int)k, smax, gk,    \
                           seed_stride, N, gstride, nob);                                                       \
    }
    if (code_bytes == 2) {  // uint16 codes: PLAIN tables, M = 8 / 16 (what the u16-table scan kernel takes)
#define ANNLITE_SEED16(MM)                                                                                         \
    {                                                                                                             \
        auto fn = seed_bound_kernel<MM, false, 4, true>;                                                          \
        const size_t lds = (size_t)Ks * MM * 16 + (size_t)kSeedLdsExtra;                                          \
        ANNLITE_HIP_TRY(hipFuncSetAttribute((const void *)fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); \
        hipLaunchKernelGGL(fn, dim3((unsigned)((B + 3) / 4), ny), dim3(kSeedWaves * 64), lds, st, (const uint8_t *)codes_dev, \
                           S, valid_bits_dev, lut_dev, (int)B, (int)Ks, (int)k, smax, gk, seed_stride, N, gstride, nob); \
    }
        if (M == 8) ANNLITE_SEED16(8) else ANNLITE_SEED16(16)
#undef ANNLITE_SEED16
        return launch_status("seed_bound_kernel (uint16 codes)");
    }
    if (M == 8) ANNLITE_SEED(8, 4) else if (M == 16) ANNLITE_SEED(16, 4) else if (M == 32) ANNLITE_SEED(32, 4)
    else ANNLITE_SEED(64, 2)
#undef ANNLITE_SEED
    return launch_status("seed_bound_kernel");
}

// The byte-table plan's whole preparation in ONE launch (M = 16, L2 tables, sub-vectors of a multiple of 4 floats, D <= 256):
// tables of every group of 4 queries built into LDS + global memory, quantisation parameters, reset of the lists / bounds,
// seed bound from S rows spread over the table's N (N <= 0: its first S).
int annlite::launch_seed_build(bool skw, const void *codes_dev, int64_t S, int64_t N, const uint32_t *valid_bits_dev, const LutBuild &build,
                               float *lut_out, int64_t B, int64_t Ks, int64_t k, float *qstep, double *qlo, float *smax, float *qlom,
                               unsigned long long *gk, void *fill, size_t fill_bytes, size_t gk_bytes, hipStream_t st,
                               unsigned long long *gseed0, uint8_t *btab, int target, unsigned long long *dbg,
                               unsigned long long *seedk, const uint32_t *cand, int n_cand) {
    constexpr int M = 16;
    SeedBuild sb;
    sb.queries = build.queries;
    sb.cb = build.codebooks;
    sb.lut_out = lut_out;
    sb.qstep = qstep;
    sb.smax = smax;
    sb.qlom = qlom;
    sb.qlo = qlo;
    sb.fill = (u32x4 *)fill;
    sb.fill_vec16 = (int64_t)(fill_bytes / 16);
    sb.skip_lo = (int64_t)(((char *)gk - (char *)fill) / 16);
    sb.skip_hi = sb.skip_lo + (int64_t)(gk_bytes / 16);
    sb.D = (int32_t)build.D;
    sb.qmax = 32767 / M;
    sb.gate = nullptr;
    sb.gseed0 = (gseed0 && btab) ? gseed0 : nullptr;
    sb.btab = (gseed0 && btab) ? btab : nullptr;
    sb.target = target;
    sb.chunk_log = seed_chunk_log();
    sb.dbg = dbg;
    sb.seedk = seedk;
    sb.cand = cand;
    sb.n_cand = cand ? n_cand : 0;
    sb.cells = nullptr;
    sb.cell_rows = nullptr;
    sb.n_probe = 0;
    sb.bq = nullptr;
    auto fn = skw ? seed_bound_kernel<M, true, 4, false, true> : seed_bound_kernel<M, false, 4, false, true>;
    const size_t lds = (size_t)Ks * M * 16 + (size_t)kSeedLdsExtra;
    ANNLITE_HIP_TRY(hipFuncSetAttribute((const void *)fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    const unsigned n_g4 = (unsigned)(((B + 15) / 16) * 4);
    hipLaunchKernelGGL(fn, dim3(n_g4, 1), dim3(kSeedWaves * 64), lds, st, (const uint8_t *)codes_dev, S, valid_bits_dev,
                       (const float *)nullptr, (int)B, (int)Ks, (int)k, (const float *)nullptr, gk, (int64_t)0, N > S ? N : S, 0, sb);
    return launch_status("seed_bound_kernel (fused table build)");
}

// The preparation launch of a pruned search over cells (annlite_ivf_search_topk): launch_seed_build's kernel with CELLS -- every
// query's first bound comes from S / 4 rows of its NEAREST cell, the byte tables go out per query (bq) with their seed keys (gseed0).
int annlite::launch_seed_build_cells(bool skw, const void *codes_dev, int64_t S, int64_t N, const uint32_t *valid_bits_dev,
                                     const LutBuild &build, float *lut_out, int64_t B, int64_t Ks, int64_t k, float *qstep, double *qlo,
                                     float *smax, float *qlom, unsigned long long *gk, hipStream_t st, unsigned long long *gseed0,
                                     uint8_t *bq, int target, const int32_t *cells, int64_t n_probe, const int64_t *cell_rows,
                                     unsigned int *item_counter, bool ip_tables, const int32_t *seed_cells) {
    constexpr int M = 16;
    SeedBuild sb = {};
    sb.queries = build.queries;
    sb.cb = build.codebooks;
    sb.lut_out = lut_out;
    sb.qstep = qstep;
    sb.smax = smax;
    sb.qlom = qlom;
    sb.qlo = qlo;
    sb.fill = nullptr;  // (nothing to reset: the scan writes every list it later reads, gkey is written here)
    sb.fill_vec16 = 0;
    sb.D = (int32_t)build.D;
    sb.qmax = 32767 / M;
    sb.gseed0 = gseed0;
    sb.target = target;
    sb.chunk_log = 0;
    sb.cells = cells;
    sb.seed_cells = seed_cells;
    sb.cell_rows = cell_rows;
    sb.n_probe = (int32_t)n_probe;
    sb.bq = bq;
    sb.item_counter = item_counter;
    sb.ip_inv_ks = (float)(1.0 / (double)Ks);
    const size_t lds = (size_t)Ks * M * 16 + (size_t)kSeedLdsExtra;
    auto fn = ip_tables ? (skw ? seed_bound_kernel<M, true, 4, false, true, true, true> : seed_bound_kernel<M, false, 4, false, true, true, true>)
                        : (skw ? seed_bound_kernel<M, true, 4, false, true, true> : seed_bound_kernel<M, false, 4, false, true, true>);
    ANNLITE_HIP_TRY(hipFuncSetAttribute((const void *)fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    const unsigned n_g4 = (unsigned)(((B + 15) / 16) * 4);
    hipLaunchKernelGGL(fn, dim3(n_g4, 1), dim3(kSeedWaves * 64), lds, st, (const uint8_t *)codes_dev, S, valid_bits_dev,
                       (const float *)nullptr, (int)B, (int)Ks, (int)k, (const float *)nullptr, gk, (int64_t)0, N > S ? N : S, 0, sb);
    return launch_status("seed_bound_kernel (cells)");
}
