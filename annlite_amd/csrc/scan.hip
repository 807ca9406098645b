// scan.hip -- the ADC scan's C entry points: plan, dispatch, merge / top-k / gather utility kernels.
//
// Reference semantics (jina-ai/annlite v0.5.11):
//   for each query b:  d[n] = sum_{m=0..M-1, ascending, fp32} lut[b][m][codes[n][m]]
//                      (bindings/pq_bindings.pyx:30-47,52-80 == include/hnswlib/space_pq.h:15-37)
//   then the k smallest (annlite/math.py:94-120) -- with the build's fixed tie-break (d asc, n asc).
//
// The scan kernels live in scan_q8.hip (byte filter tables: the default for M = 16, small k), scan_qfilter.hip (u16
// filter tables, tile mode) and scan_prep.hip (table build, quantisation parameters, seed bound); DESIGN.md section 3.  No fallback to the
// CPU exists; shapes without a fast kernel use the generic kernel below (tables through L2).
#include <mutex>

#include "scan_common.h"

namespace annlite {

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
            const unsigned long long pm = __ballot(!(d > tf)) & vmask;  // (a NaN sum is a candidate too: it sorts behind +inf)
            if (pm) {
                wavelist_offer(L, pm, f32_to_key(d), (uint32_t)(row0 + lane), km1, th, tl, lane);
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
// The same scan with the query's table IN LDS (round 6): shapes without a filter kernel whose fp32 table [M][Ks] fits
// (<= 144 KB: the reference example's M = 128 / 1-float sub-vectors, examples/pq_benchmark.py:44, odd M, uint16 codes at M = 16,
// ...).  One query per 16-wave workgroup; a lane's code row comes in 16-byte (or 4-byte) loads instead of one byte load per
// (row, sub-space) -- the generic kernel's M global byte loads per row, every one a separate cache-line request per lane, were
// 99 % of its time: 1M rows x 256 queries at M = 128 took 143 ms (2.3 10^11 look-ups/s).  The sum stays the reference's
// ascending-m fp32 chain (pq_bindings.pyx:30-47); all lanes read the same sub-space's row of the table at a time, so the LDS
// banks are hit by the codes' low bits (random: ~4-way conflicts) -- exact, simple, and ~20x the generic kernel.
// VEC: bytes per code-row load (16 where M * sizeof(CODE_T) is a multiple of 16, else 4, else sizeof(CODE_T)).
// =================================================================================================
template <typename CODE_T, int NW, int VEC>
__global__ __launch_bounds__(NW * 64) void adc_scan_lds_kernel(const ScanArgs a, int M) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    typedef __attribute__((address_space(3))) unsigned char *lds_bytes;
    const uint32_t tab_ad = (uint32_t)(uintptr_t)(lds_bytes)smem;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int km1 = a.k - 1;
    const int n_items = a.n_tiles * a.n_slices;
    const int n_tab = M * a.Ks;  // floats
    unsigned long long *scratch = (unsigned long long *)(smem + (((size_t)n_tab * 4 + 15) / 16) * 16);  // [NW][64]
    constexpr int CPV = VEC / (int)sizeof(CODE_T);  // codes per load
    const int n_vec = M / CPV;                      // (M * sizeof(CODE_T) is a multiple of VEC: the launcher picks VEC)
    int loaded = -1;
    for (int item = blockIdx.x; item < n_items; item += gridDim.x) {
        const int b = item % a.n_tiles;  // QT == 1
        const int slice = item / a.n_tiles;
        __syncthreads();  // (every wave is done with the previous item's table and scratch)
        if (b != loaded) {
            const float *lut = a.lut + (int64_t)b * n_tab;
            if ((n_tab & 3) == 0) {
                for (int i = tid; i < n_tab / 4; i += NW * 64) ((f32x4 *)smem)[i] = ((const f32x4 *)lut)[i];
            } else {
                for (int i = tid; i < n_tab; i += NW * 64) ((float *)smem)[i] = lut[i];
            }
            loaded = b;
        }
        __syncthreads();
        WaveList L;
        L.reset();
        uint32_t th = kKeyInfHi, tl = kIdNone;
        float tf = __builtin_inff();
        const int64_t slice_begin = (int64_t)slice * a.slice_rows;
        int64_t slice_end = slice_begin + a.slice_rows;
        if (slice_end > a.N) slice_end = a.N;
        const unsigned char *codes = (const unsigned char *)a.codes;
        const int64_t row_bytes = (int64_t)M * (int64_t)sizeof(CODE_T);
        for (int64_t row0 = slice_begin + (int64_t)wave * 64; row0 < slice_end; row0 += (int64_t)NW * 64) {
            int64_t row = row0 + lane;
            const bool inb = row < slice_end;
            if (!inb) row = a.N - 1;
            const unsigned char *cr = codes + row * row_bytes;
            float d = 0.f;
            uint32_t base = tab_ad;  // LDS byte address of sub-space m's row of the table
            const uint32_t ks4 = (uint32_t)a.Ks * 4u;
            for (int v = 0; v < n_vec; ++v) {
                uint32_t w[VEC >= 4 ? VEC / 4 : 1];
                if constexpr (VEC == 16) {
                    const u32x4 x = *(const u32x4 *)(cr + (int64_t)v * 16);
                    w[0] = x.x, w[1] = x.y, w[2] = x.z, w[3] = x.w;
                } else if constexpr (VEC == 4) {
                    w[0] = *(const uint32_t *)(cr + (int64_t)v * 4);
                } else {
                    w[0] = (uint32_t)(*(const CODE_T *)(cr + (int64_t)v * (int64_t)sizeof(CODE_T)));
                }
#pragma unroll
                for (int e = 0; e < CPV; ++e) {  // ascending m: the reference's sum order
                    uint32_t code;
                    if constexpr (sizeof(CODE_T) == 1) code = (w[e / 4] >> (8 * (e % 4))) & 0xffu;
                    else if constexpr (sizeof(CODE_T) == 2) code = (w[e / 2] >> (16 * (e % 2))) & 0xffffu;
                    else code = w[e];
                    d += *(const __attribute__((address_space(3))) float *)(uintptr_t)(base + (code << 2));
                    base += ks4;
                }
            }
            unsigned long long vmask = __ballot(inb);
            if (a.valid) {
                const uint32_t *vw = a.valid + (row0 >> 5);
                unsigned long long vb = (unsigned long long)vw[0];
                if (row0 + 32 < a.N) vb |= (unsigned long long)vw[1] << 32;
                vmask &= vb;
            }
            const unsigned long long pm = __ballot(!(d > tf)) & vmask;  // (a NaN sum is a candidate too: it sorts behind +inf)
            if (pm) {
                wavelist_offer(L, pm, f32_to_key(d), (uint32_t)(row0 + lane), km1, th, tl, lane);
                tf = (th == kKeyInfHi) ? __builtin_inff() : ordered_to_f32(th);
            }
        }
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

// reset of the scan's result lists / shared bounds / counters to all-ones where no launch of the plan does it on the way
__global__ __launch_bounds__(256) void fill_ones_kernel(u32x4 *p, int64_t n16) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n16) p[i] = (u32x4){~0u, ~0u, ~0u, ~0u};
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
    // (compared through the order-preserving keys, the scan kernels' own order: NaN sums behind +inf, numpy's rule)
    const int total = G * k;
    uint32_t lk = kKeyInfHi;
    int64_t li = INT64_MAX;
    for (int c = 0; c < total; ++c) {
        const int g = c / k, j = c - g * k;
        const int64_t e = ((int64_t)g * B + b) * k + j;
        const float cd = packed ? __uint_as_float((uint32_t)packed[e * 2 + 1]) : dist[e];
        const int64_t ci = packed ? packed[e * 2] : id[e];
        if (ci < 0) continue;  // padding entry of a short shard
        const uint32_t ck = f32_to_key(cd);
        const bool less = (lk < ck) || (lk == ck && li < ci);
        const int pos = __popcll(__ballot(less));
        if (pos > km1) continue;
        const uint32_t sk = (uint32_t)__shfl_up((int)lk, 1);
        const int64_t si = __shfl_up(li, 1);
        if (lane == pos) {
            lk = ck;
            li = ci;
        } else if (lane > pos) {
            lk = sk;
            li = si;
        }
    }
    if (lane <= km1) {
        const bool none = (li == INT64_MAX);
        const float ld = ordered_to_f32(lk);
        out_d[(int64_t)b * k + lane] = none ? __builtin_inff() : (sqrt_out ? __builtin_sqrtf(ld) : ld);
        out_i[(int64_t)b * k + lane] = none ? (int64_t)-1 : li;
    }
}

// Seed exchange of the row-sharded search (annlite_pq_search_split): all_keys [G][B][kSeedKeys] = every rank's bounds of its
// seed's k smallest rows.  One wave per query: the k-th smallest of the G * k keys (ties by position: equal keys of two ranks
// are two rows) has k distinct rows of the global table at or below it; the prepared batch's bound becomes min(own, that).
__global__ __launch_bounds__(256) void seed_union_kernel(const unsigned long long *__restrict__ all_keys, int G, int B, int k,
                                                        unsigned long long *__restrict__ gkey) {
    __shared__ unsigned long long s_keys[4][8 * kSeedKeys];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int b = blockIdx.x * 4 + w;
    if (b >= B) return;
    const int n = G * kSeedKeys;  // (G <= 8: at most 128 slots, two per lane)
    unsigned long long mine[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        const int idx = lane + 64 * u;
        mine[u] = ~0ull;
        if (idx < n && (idx % kSeedKeys) < k) mine[u] = all_keys[((int64_t)(idx / kSeedKeys) * B + b) * kSeedKeys + (idx % kSeedKeys)];
        if (idx < 8 * kSeedKeys) s_keys[w][idx] = mine[u];
    }
    // (a wave's own LDS traffic is in order: no barrier)
    unsigned long long found = ~0ull;
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        const int idx = lane + 64 * u;
        if (idx >= n || mine[u] == ~0ull) continue;
        int rank = 0;
        for (int j = 0; j < n; ++j) {
            const unsigned long long o = s_keys[w][j];
            rank += (o < mine[u] || (o == mine[u] && j < idx)) ? 1 : 0;
        }
        if (rank == k - 1) found = mine[u];
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const unsigned long long p = __shfl_xor(found, o);
        found = p < found ? p : found;
    }
    if (lane == 0 && found < gkey[b]) gkey[b] = found;
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
        const unsigned long long pm = __ballot(i < N && !(d > tf));  // (NaN values: candidates that sort behind +inf, numpy's order)
        if (pm) {
            wavelist_offer(L, pm, f32_to_key(d), (uint32_t)i, km1, th, tl, lane);
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
    if (M == 64) {  // two independently skewed halves, wrap-coded (see wrap64_mask)
        const int r = (int)(id % 32), h = j / 32, pp = j % 32;
        if (!inverse) {
            uint8_t v = in[i * M + 32 * h + (pp + r) % 32];
            if (pp + r >= 32) v = (uint8_t)(v - 1);
            out[id * M + j] = v;
        } else {
            const int ps = ((pp - r) % 32 + 32) % 32;  // stored position (within its half) of sub-space j
            uint8_t v = in[id * M + 32 * h + ps];
            if (ps + r >= 32) v = (uint8_t)(v + 1);
            out[i * M + j] = v;
        }
        return;
    }
    const int r = (int)(id % M);
    if (!inverse) {
        out[id * M + j] = in[i * M + (j + r) % M];
    } else {
        const int jj = ((j - r) % M + M) % M;  // stored position of sub-space j
        out[i * M + j] = in[id * M + jj];
    }
}

// =================================================================================================
// host side
// =================================================================================================
struct FastCfg {
    int M, QI, NQ, NW, WPS, wg_per_cu, id;
    int mode;  // 4: u16 filter tables (scan_qfilter.hip: 8 queries per LDS entry, M = 64: 4), 5: byte filter tables
               // (scan_q8.hip: 16 queries per LDS entry, tables quantised by the workgroup itself)
    int qt() const { return mode == 5 ? (M == 64 ? 8 : 16 * NQ) : (M == 64 ? 4 : 8) * NQ; }
    bool qf() const { return true; }  // integer filter + exact recompute, shared bounds (every fast kernel)
};

// Kernel variants per M.  ANNLITE_SCAN_VARIANT (environment, parsed at load -- common.h: Knobs; A/B measurements) selects among the
// instantiations; variant 0 is the default plan for every M.  Which of the two M = 16 kernels serves a table -- byte or u16
// filter tables -- is otherwise decided inside the library, per call (search_policy below): the entry points scope their
// choice to the call through this thread-local (never visible to the caller, never left set).
static thread_local int g_variant_scope = -1;
struct VariantScope {
    int prev;
    explicit VariantScope(int v) : prev(g_variant_scope) { g_variant_scope = v; }
    ~VariantScope() { g_variant_scope = prev; }
};
static int env_variant() { return knobs().scan_variant; }  // (ANNLITE_SCAN_VARIANT as parsed at load / the last reload)
static int scan_variant() {
    if (g_variant_scope >= 0) return g_variant_scope;
    const int e = env_variant();
    return e >= 0 ? e : 0;
}

// tiles: the plan of annlite_pq_search_tiles (IVF cells) -- the u16 kernels' tile mode
static bool fast_cfg(int64_t M, int64_t Ks, int code_bytes, int64_t k, FastCfg *c, bool tiles = false) {
    if (Ks < 1 || k > 64 || k < 1) return false;
    if (code_bytes == 2) {
        // uint16 codes (Ks > 256; the reference's PQ tests run Ks = 512 and 768 at M = 8): the u16-table kernel with 8 queries
        // per workgroup, as many codes as fit the LDS, PLAIN layout, row slices only
        if (tiles || knobs().no_fast_code16) return false;  // (the switch: A/B against the generic kernel)
        // M = 8, Ks <= 512, k <= 16: the byte-table kernel (scan_q8.hip, C16: table [Ks][2][8][16 B], 32 queries per workgroup,
        // conflict-free); ANNLITE_SCAN_VARIANT=31: the u16-table kernel (A/B)
        if (M == 8 && Ks <= 512 && k <= 16 && (scan_variant() == 0 || scan_variant() == 50)) { *c = {8, 4, 2, 16, 4, 1, 850, 5}; return true; }
        // (16 < k <= 64, round 6: the same kernel with 64-key lists -- only where the library asks for it, variant 50, as for M = 16)
        if (M == 8 && Ks <= 512 && k <= 64 && scan_variant() == 50) { *c = {8, 4, 2, 16, 4, 1, 8650, 5}; return true; }
        if (M == 8 && Ks <= 512) { *c = {8, 4, 2, 16, 4, 1, 8217, 4}; return true; }   // u16 tables, 16 queries per workgroup
        // 512 < Ks <= 1024: byte tables of ONE entry group (16 queries per workgroup; 2-way bank conflicts, inherent -- still
        // half the LDS time per query of the u16 tables' 8)
        if (M == 8 && Ks <= 1024 && k <= 16 && (scan_variant() == 0 || scan_variant() == 50)) { *c = {8, 4, 1, 16, 4, 1, 851, 5}; return true; }
        if (M == 8 && Ks <= 1024 && k <= 64 && scan_variant() == 50) { *c = {8, 4, 1, 16, 4, 1, 8651, 5}; return true; }
        if (M == 8 && Ks <= 1024) { *c = {8, 4, 1, 16, 4, 1, 8216, 4}; return true; }
        if (M == 16 && Ks <= 512) { *c = {16, 4, 1, 16, 4, 1, 16216, 4}; return true; }
        return false;
    }
    if (code_bytes != 1 || Ks > 256) return false;
    const int v = scan_variant();
    switch (M) {
        case 8:
            // default: byte tables (scan_q8.hip, M8 shape: table [Ks][2][8][16 B] = 64 KB, 32 queries / WG); k > 16, tile mode and
            // variant 31: u16 tables, 16 queries / WG
            if ((v == 0 || v == 50) && !tiles && k <= 16) { *c = {8, 4, 2, 16, 4, 1, 852, 5}; return true; }
            if (v == 50 && !tiles && k <= 64) { *c = {8, 4, 2, 16, 4, 1, 864, 5}; return true; }  // 64-key lists (round 6)
            *c = {8, 4, 2, 16, 4, 1, 830, 4};
            return true;
        case 16:
            // default: byte tables, 32 queries / WG, 15 scanning waves + 1 consumer (small k: the candidate generator of the
            // re-rank stage asks for 64 per slice and starts without a seed -- the u16 tables filter that much better)
            if ((v == 0 && !tiles && k <= 16) || (v == 50 && !tiles && k <= 16)) { *c = {16, 4, 2, 16, 4, 1, 1650, 5}; return true; }
            // 16 < k <= 64 (round 5): the byte-table kernel with 64-key lists -- only where the library asks for it (variant 50: the
            // shared-bound search, scan_topk_impl); the PUBLIC plan of these k stays the u16 plan (the candidate generator, tile mode)
            if (v == 50 && !tiles && k <= 64) { *c = {16, 4, 2, 16, 4, 1, 1664, 5}; return true; }
            if (v == 30) { *c = {16, 4, 2, 12, 3, 1, 1630, 4}; return true; }  // u16 tables, 12 waves
            if (v == 32) { *c = {16, 4, 2, 8, 2, 1, 1632, 4}; return true; }   // u16 tables, 8 waves
            *c = {16, 4, 2, 16, 4, 1, 1631, 4};                               // u16 tables, 16 queries / WG, 16 waves (variant 31)
            return true;
        case 32:
            // default: byte tables (scan_q8.hip, 3250: one entry group, 16 queries / WG, 15 scanning waves + 1 consumer -- 10M rows x 1024
            // queries 2.79 ms against the u16 tables' 7.41); k > 16, tile mode and variant 31: u16 tables, 8 queries / WG
            if ((v == 0 || v == 50) && !tiles && k <= 16) { *c = {32, 4, 1, 16, 4, 1, 3250, 5}; return true; }
            if (v == 50 && !tiles && k <= 64) { *c = {32, 4, 1, 16, 4, 1, 3264, 5}; return true; }  // 64-key lists (round 6)
            *c = {32, 4, 1, 12, 3, 1, 3230, 4};
            return true;
        case 64:
            // default: byte tables (scan_q8.hip, WIDE: 8 queries per 8-byte entry, u16 sums), 15 scanning waves + 1 consumer;
            // k > 16, tile mode and variant 31: the u16-table kernel (4 queries per entry, 12 waves)
            if ((v == 0 || v == 50) && !tiles && k <= 16 && Ks == 256) { *c = {64, 4, 1, 16, 4, 1, 6450, 5}; return true; }
            if (v == 30) *c = {64, 4, 1, 16, 4, 1, 6430, 4};       // qfilter64, 16 waves (spills: 7.4 ms at C4 vs 5.9)
            else if (v == 32) *c = {64, 4, 1, 8, 2, 1, 6432, 4};   // qfilter64, 8 waves (6.7 ms)
            else *c = {64, 4, 1, 12, 3, 1, 6431, 4};               // default: qfilter64, 4 queries / WG, 12 waves
            return true;
        default: return false;
    }
}

// shapes the byte-table kernel serves beyond k = 16 (64-key lists, scan_q8.hip ids 1664; round 6: 864 / 3264 / 8650 / 8651)
static bool lk64_shape(int64_t M, int64_t Ks, int code_bytes, int64_t k) {
    if (k <= 16 || k > 64) return false;
    if (code_bytes == 1) return (M == 16 || M == 8 || M == 32) && Ks <= 256;
    return code_bytes == 2 && M == 8 && Ks <= 1024;
}

// shapes without a filter kernel whose fp32 table [M][Ks] fits the LDS beside the waves' merge scratch: adc_scan_lds_kernel
static bool generic_table_in_lds(int64_t M, int64_t Ks) { return M * Ks * 4 <= 144 * 1024; }

static int round_up(int64_t x, int64_t m) { return (int)(((x + m - 1) / m) * m); }
// queries the per-query workspace arrays are sized for: whole tiles, at least the 16 the fp32 TILED table is padded to
static int64_t pad_queries(int64_t B, int qt) {
    const int64_t p = qt > 16 ? qt : 16;
    return ((B + p - 1) / p) * p;
}

// m32_bytes: the byte-table plan of M = 32 (64 tiles of 16 queries at 1024 queries): at >= 8M rows 8 slices instead of the 4 one
// work item per CU gives -- two work items per workgroup -- so that the slice-per-XCD map applies (an XCD streams an eighth of the
// 32-byte rows for all tiles instead of the whole table for its own: -2 % at 10M rows, profiles/r05/m32_8_slices_xcd_map_ab.txt)
static bool m32_wants_8_slices(int64_t N, int n_tiles) { return N >= 8000000 && n_tiles >= 32 && n_tiles <= 128; }

static void plan_slices(int64_t N, int n_tiles, int waves, int n_cu, bool xcd8, int *n_slices, int64_t *slice_rows,
                        int64_t B = 0, bool m32_bytes = false) {
    // at least one slice per XCD; more when few query tiles exist, as long as every wave keeps
    // >= 8 steps of 64 rows
    const int64_t min_rows = (int64_t)waves * 64 * (xcd8 ? 16 : 8);
    // XCD-mapped plans: ONE work item per CU when the tiles allow it (every (query, slice) list pays its own
    // logarithmic number of candidate events: 4 slices instead of 8 was 10% faster at 1.25M rows x 1024
    // queries); 1, 2, 4 or a multiple of 8 slices (item_map)
    int64_t want = ((xcd8 ? 1 : 2) * (int64_t)n_cu + n_tiles - 1) / n_tiles;
    int64_t cap = N / min_rows;
    if (want > cap) want = cap;
    // a single tile with several real queries: every (query, slice) list pays its own candidate events, 128
    // slices beat 256 from 3 queries on (10M rows: 8 queries 0.23 vs 0.27 ms, 16 queries 0.31 vs 0.40 ms)
    if (xcd8 && n_tiles == 1 && B > 2 && want > 128) want = 128;
    int64_t ns = want;
    if (xcd8) ns = want <= 1 ? 1 : want <= 2 ? 2 : want <= 4 ? 4 : ((want + 7) / 8) * 8;
    if (ns < 1) ns = 1;
    if (xcd8 && m32_bytes && ns == 4 && m32_wants_8_slices(N, n_tiles)) ns = 8;
    if (xcd8) {
        {
            const int64_t v = knobs().scan_slices;  // (ANNLITE_SCAN_SLICES)
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

static int plan_query_impl(int64_t N, int64_t M, int64_t Ks, int code_bytes, int64_t B, int64_t k, int force_ns,
                           annlite_scan_plan *plan);

// force_ns > 0: tile mode (every query tile scans its own row range as ONE work item)
static int plan_query_impl(int64_t N, int64_t M, int64_t Ks, int code_bytes, int64_t B, int64_t k, int force_ns,
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
    if (fast_cfg(M, Ks, code_bytes, k, &c, force_ns > 0)) {
        plan->fast = 1;
        plan->qi = c.QI;
        plan->qt = c.qt();
        plan->waves = c.NW;
        const int n_tiles = (int)((B + plan->qt - 1) / plan->qt);
        int ns;
        int64_t sr;
        plan_slices(N > 0 ? N : 1, n_tiles > 0 ? n_tiles : 1, c.NW, n_cu, true, &ns, &sr, B, c.mode == 5 && M == 32 && force_ns == 0);
        if (force_ns > 0) ns = force_ns;
        plan->n_slices = ns;
        plan->lut_floats = ((B + 15) / 16) * 16 * M * Ks;  // padded to 16 queries
        // [partial keys][Smax f32 x Bpad][qstep f32 x Bpad][qlo f64 x Bpad][lo,hi f32 x Bpad*M][q16 u16 x Bpad*M*Ks]
        const int64_t bpad = pad_queries(B, plan->qt);
        plan->workspace_bytes = (int64_t)n_tiles * plan->qt * ns * k * 8 + 256 + bpad * 4 + 256;
        if (c.qf())
            plan->workspace_bytes += bpad * 4 + 256 + 2 * (bpad * 8 + 256) + (bpad * ns * 8 * (c.mode == 5 ? kGk2Keys : gk2_cell_keys(M)) + 256) + (n_tiles * 4 + 256) +
                                     256 /* item counter */ + 256 /* guard block */ + bpad * M * Ks * 2 + 256;
        // byte-table plan chosen by default: the gated u16-table launch that redoes the scan if the byte-table launch gives
        // up (search_policy) works in its own region behind this one
        if (c.mode == 5 && g_variant_scope < 0 && env_variant() < 0) {
            annlite_scan_plan p2;
            VariantScope vs(31);
            if (plan_query_impl(N, M, Ks, code_bytes, B, k, force_ns, &p2) == ANNLITE_OK)
                plan->workspace_bytes = ((plan->workspace_bytes + 255) / 256) * 256 + p2.workspace_bytes;
        }
        // ... and the other way round for 16 < k <= 64 at M = 16: the public plan is the u16 plan, the library's search runs the
        // byte-table kernel (64-key lists) in a region of its own IN FRONT of it
        if (c.mode == 4 && k > 16 && lk64_shape(M, Ks, code_bytes, k) && force_ns == 0 && g_variant_scope < 0 && env_variant() < 0) {
            annlite_scan_plan p1;
            VariantScope vs(50);
            if (plan_query_impl(N, M, Ks, code_bytes, B, k, force_ns, &p1) == ANNLITE_OK)
                plan->workspace_bytes = ((p1.workspace_bytes + 255) / 256) * 256 + plan->workspace_bytes;
        }
    } else {
        plan->fast = 0;
        plan->qi = 1;
        plan->qt = 1;
        plan->waves = generic_table_in_lds(M, Ks) ? 16 : 4;  // (adc_scan_lds_kernel: one 16-wave workgroup per query and row slice)
        int ns;
        int64_t sr;
        plan_slices(N > 0 ? N : 1, B > 0 ? (int)B : 1, plan->waves, n_cu, false, &ns, &sr);
        plan->n_slices = ns;
        plan->lut_floats = B * M * Ks;
        plan->workspace_bytes = B * ns * k * 8;
    }
    if (plan->workspace_bytes < 8) plan->workspace_bytes = 8;
    return ANNLITE_OK;
}

extern "C" int annlite_scan_plan_query(int64_t N, int64_t M, int64_t Ks, int code_bytes, int64_t B, int64_t k,
                                       annlite_scan_plan *plan) {
    return plan_query_impl(N, M, Ks, code_bytes, B, k, 0, plan);
}

extern "C" int annlite_scan_plan_tiles(int64_t N, int64_t M, int64_t Ks, int code_bytes, int64_t V, int64_t k,
                                       annlite_scan_plan *plan) {
    return plan_query_impl(N, M, Ks, code_bytes, V, k, 1, plan);
}

static unsigned long long *g_dbg = nullptr;  // debug only (ANNLITE_DEBUG_COUNTERS): leaked device buffer: 16 counters + 4096 per-item records of 8
static unsigned long long *g_dbg_prep = nullptr;  // ... and the preparation launch's 8 phase stamps (annlite_debug_prep_timeline)
static thread_local int g_prof_on = 0;
static thread_local hipEvent_t g_ev0 = nullptr, g_ev1 = nullptr;
static thread_local int g_ev_valid = 0;
// (annlite_profile_enable) the byte-table kernel's cycle / wall-clock stamps, 4 x u64 PER DEVICE (a launch on device d writes
// device d's buffer -- one process may drive several GPUs: MultiGpuPQIndex), allocated by annlite_profile_enable(1) on the device
// that is current then (or by the first profiled launch on another one), leaked
static unsigned long long *g_clk_dev[32] = {};
static std::mutex g_clk_mutex;
static unsigned long long *clk_buffer(bool create) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 32) return nullptr;
    std::lock_guard<std::mutex> lock(g_clk_mutex);
    if (!g_clk_dev[dev] && create) {
        unsigned long long *p = nullptr;
        if (hipMalloc((void **)&p, 32) != hipSuccess) return nullptr;
        if (hipMemset(p, 0, 32) != hipSuccess) {
            (void)hipFree(p);
            return nullptr;
        }
        g_clk_dev[dev] = p;
    }
    return g_clk_dev[dev];
}

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
// where a scan that can merge its slices itself puts the final result (merged is set when it did)
struct ScanOut {
    float *d;
    int64_t *i;
    int64_t *packed;
    int64_t row_base;
    int sqrt_out;
    bool merged;
};
// tile mode (annlite_pq_search_tiles): query tile t = queries [t*qt, (t+1)*qt) scans rows tile_rows[t]
struct TileMode {
    const int64_t *tile_rows;  // [B / qt][2]
    const int32_t *vmap;       // [B]
    uint32_t *cand;            // [B][cand_cap] out
    uint32_t *cand_count;      // [B] out
    int64_t cand_cap;
    int64_t n_queries;         // real queries: the tables are built for THEM, slot s uses the tables of query vmap[s]
};

// kernel choice inside the library (search_policy): what a scan launch takes part in
struct GuardOpt {
    int abort_enabled;            // byte-table launch: it may give up (a gated u16 launch is queued behind it)
    unsigned int *host_stats;     // byte-table launch: the caller's host-mapped statistics block (or NULL)
    uint32_t seq;                 // ... and the sequence number its last workgroup leaves there
    const unsigned int *gate;     // u16 launch: run only if *gate == 0 (every kernel of the launch: quantisation, seed, scan)
    unsigned int *guard_out;      // out: the byte-table launch's guard block (the gate of the launch behind it)
};

// annlite_pq_search_split: the byte-table plan in two halves with the ranks' seed exchange in between
struct SplitOpt {
    int phase;                      // ANNLITE_PHASE_PREPARE: everything up to and including the preparation launch; _SCAN: the rest
    int64_t seed_rows;              // rows of this rank's seed (<= 0: the default)
    unsigned long long *seed_keys;  // PREPARE: [B][kSeedKeys] out
};

static int scan_partial(const void *codes_dev, int code_bytes, int codes_layout, int64_t N, int64_t M, int64_t Ks,
                        const uint32_t *valid_bits_dev, const float *lut_dev, int64_t B, int64_t k,
                        void *workspace_dev, size_t workspace_bytes, hipStream_t st, annlite_scan_plan *plan_out,
                        bool share_across_slices, const LutBuild *build = nullptr, ScanOut *outp = nullptr,
                        const TileMode *tm = nullptr, GuardOpt *gopt = nullptr, const SplitOpt *split = nullptr,
                        bool cand_seed = false) {
    annlite_scan_plan plan;
    int rc = plan_query_impl(N, M, Ks, code_bytes, B, k, tm ? 1 : 0, &plan);
    if (rc != ANNLITE_OK) return rc;
    *plan_out = plan;
    if (B == 0) return ANNLITE_OK;
    if (tm) {
        FastCfg ct;
        ANNLITE_REQUIRE(plan.fast && fast_cfg(M, Ks, code_bytes, k, &ct, true) && ct.mode == 4,
                        "tile mode needs the quantised-filter plan (M in {8,16,32,64}, Ks <= 256, uint8 codes)");
        ANNLITE_REQUIRE(B % plan.qt == 0 && tm->tile_rows && tm->vmap && tm->cand && tm->cand_count && tm->cand_cap >= 64,
                        "tile mode: B=%lld must be a multiple of the tile size %d", (long long)B, plan.qt);
    }
    ANNLITE_REQUIRE(codes_layout == ANNLITE_CODES_PLAIN || (codes_layout == ANNLITE_CODES_SKEWED && plan.fast && code_bytes == 1),
                    "codes_layout %d not supported by this plan (SKEWED needs the fast plan and uint8 codes)", codes_layout);
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
    const Knobs &kn = knobs();  // (one block for the whole call)
    a.dbg_skip = kn.debug_skip;
    if (kn.debug_counters && !(gopt && gopt->gate)) {  // (the gated pass leaves the first launch's counters alone)
        if (!g_dbg) ANNLITE_HIP_TRY(hipMalloc((void **)&g_dbg, 128 + 4096 * 64));
        if (!g_dbg_prep) {
            ANNLITE_HIP_TRY(hipMalloc((void **)&g_dbg_prep, 64));
            ANNLITE_HIP_TRY(hipMemset(g_dbg_prep, 0, 64));
        }
        ANNLITE_HIP_TRY(hipMemsetAsync(g_dbg, 0, 128, st));
        a.dbg = g_dbg;
        if (kn.debug_counters == 2) a.dbg_skip |= 8;  // phase stamps only (annlite_debug_timeline): the
                                                                           // per-wave event counters cost tens of microseconds
    }
    {
        int ns;
        int64_t sr;
        FastCfg cs;
        const bool m32_bytes = plan.fast && !tm && M == 32 && fast_cfg(M, Ks, code_bytes, k, &cs, false) && cs.mode == 5;
        plan_slices(N > 0 ? N : 1, a.n_tiles, plan.waves, n_cu, plan.fast != 0, &ns, &sr, B, m32_bytes);
        a.slice_rows = tm ? ((N + 63) / 64) * 64 : sr;
    }
    // slots of slices that hold no rows stay "none"
    size_t fill_bytes = 0;
    bool fused_fill = false;
    {
        // partial lists, and (quantised-filter plan) the shared bounds right behind them; the rest is scratch
        size_t fill = (size_t)plan.workspace_bytes;
        FastCfg c0;
        if (plan.fast && fast_cfg(M, Ks, code_bytes, k, &c0, tm != nullptr) && c0.qf()) {
            const size_t bpad = (size_t)pad_queries(B, plan.qt);
            auto r256 = [](size_t x) { return (x + 255) / 256 * 256; };
            fill = r256((size_t)a.n_tiles * plan.qt * plan.n_slices * k * 8) + r256(bpad * 8) +
                   r256(bpad * plan.n_slices * 8 * (c0.mode == 5 ? kGk2Keys : gk2_cell_keys(M))) + r256((size_t)a.n_tiles * 4) + 256 /* item counter */ +
                   256 /* guard block */;
        }
        fill_bytes = fill;
        FastCfg c1;
        fused_fill = N > 0 && plan.fast && fast_cfg(M, Ks, code_bytes, k, &c1, tm != nullptr) && c1.qf() && M != 64;  // lut_quantise_fused_kernel
        // (own kernel: one launch like any other of the plan, no second mechanism on the stream)
        if (split && !fused_fill) return ANNLITE_NOT_APPLICABLE;  // (the split search is the fused preparation launch's)
        if (!fused_fill) {
            const int64_t n16 = (int64_t)((fill + 15) / 16);
            hipLaunchKernelGGL(fill_ones_kernel, dim3((unsigned)((n16 + 255) / 256)), dim3(256), 0, st, (u32x4 *)workspace_dev, n16);
        }
    }
    if (N == 0) return ANNLITE_OK;
    const int n_items = tm ? a.n_tiles
                           : (plan.fast && a.n_slices < 8) ? ((a.n_tiles + 8 / a.n_slices - 1) / (8 / a.n_slices)) * 8
                                                           : a.n_tiles * a.n_slices;
    a.n_items = n_items;
    if (plan.fast) {
        FastCfg c;
        fast_cfg(M, Ks, code_bytes, k, &c, tm != nullptr);
        // M = 64 (64-byte rows: 640 MB at 10M rows): tile-per-XCD makes every XCD stream the whole table (FETCH_SIZE 6.2 GB per
        // launch, 36 % of the HBM peak, L2 hit rate 71 %) and its candidates are few -- slice-per-XCD instead: an XCD streams its
        // row slices once for all the query tiles that walk them together.  ANNLITE_Q8_MAP=0/1 overrides (A/B).
        a.q8_map_slices = (M == 64 || (M == 32 && c.mode == 5 && a.n_slices == 8 && m32_wants_8_slices(N, a.n_tiles))) ? 1 : 0;
        if (kn.q8_map >= 0) a.q8_map_slices = kn.q8_map ? 1 : 0;
        if (a.n_slices < 8) a.q8_map_slices = 0;
        if (c.mode == 5 && a.n_tiles >= 8 && !a.q8_map_slices) a.n_items = 8 * ((a.n_tiles + 7) / 8) * a.n_slices;  // q8_item_map
        // Interleaved row slices (ANNLITE_Q8_ILV=1..8, default off): slice s takes the runs of 2^ILV blocks of 64 rows number s, s + n_slices, ...
        // instead of one contiguous range -- built for tables in cluster order, where a query tile's neighbourhoods sit in ONE range and
        // that work item handles most of the tile's candidates (1.25M rows sorted along one latent direction: step loops of 175 ... 321 us
        // within one launch).  Measured (profiles/r05/interleaved_slices_ab.txt): such a table 1.456 -> 1.419 ms per batch at 10M rows but
        // 0.326 -> 0.337 at 1.25M (every slice now inserts the hot region's rows into its own list); the bench's tables -0.3 % (noise).  Exact
        // either way (332 GPU tests with it on); not adopted.  Never for the candidate generator of the re-rank stage: its per-slice seed
        // bounds are bounds of the slice's OWN contiguous rows.
        a.q8_ilv_log = 0;
        {
            const int t = kn.q8_ilv;
            if (c.mode == 5 && share_across_slices && !tm && a.n_slices >= 2 && t >= 0 && t <= 8) a.q8_ilv_log = t;
        }
        // epochs end after steps 15, 255, 4095 (x 15 blocks of 64 rows) and a slot asks for a new table when its T has halved:
        // 10M rows x 1024 queries 1.497 ms per launch with {3, 15, 63, ...} / T 64 / rebuild at 7/8, 1.445 with this, 1.441 with no
        // epoch at all (1.25M rows: 0.284 / 0.261 / 0.254) -- on the bench's data a barrier of all 16 waves costs more than a
        // finer table saves; the sparse schedule stays as the guard against a seed bound that is far off
        a.q8_epoch0 = 15;
        a.q8_epoch_mul = 16;
        a.q8_ring_limit = 384;
        a.q8_import_mask = 3;
        // (64-key lists: an import ranks 32 published keys per slot -- every 8th batch instead of every 4th: 1.79 against 1.87 ms at
        // k = 50, 10M rows; every 2nd 1.89, every 16th 1.84 -- profiles/r05/k50_knobs.txt)
        if (c.mode == 5 && k > 16) a.q8_import_mask = 7;
        // (M = 64: u16 sums of 64 entries clipped at 15.  10M x 768-d, 256 queries, ms per launch at T = 256 / 384 / 512 / 768: 2.09 / 2.04 /
        // 2.03 / 3.54 -- a finer table clips more entries of a row near the bound: at 768 the filter leaks)
        // (M = 16, k <= 16: 88 -- at 1.25M rows, where the consumer wave is 75-85 % busy, T = 80 / 88 take 0.2367 / 0.2368 ms per batch against
        // 0.2412 at 96 and 0.2548 / 0.2759 at 112 / 127: an entry is clipped at 15 steps, so a coarser step clips fewer entries of the rows near
        // the bound; 1M and 10M rows: no difference between 80, 88 and 96 -- profiles/r05/table_target_sweep.txt.  The 64-key lists stay at 96:
        // k = 50 at T = 80 / 96 / 112 1.92 / 1.87 / 1.83 ms, profiles/r05/k50_knobs.txt)
        a.q8_target = M == 64 ? 512 : (M == 16 && k <= 16) ? 88 : 96;
        a.q8_rebuild_8ths = 4;
        // how often the scanning waves re-read the workgroup's bounds (16 bytes of LDS + the filter words' registers): every 2nd step; every
        // 8th where a work item scans >= 500k rows -- the bounds of a long scan move rarely, a stale one only pushes a row the consumer
        // drops.  Same box, alternating (profiles/r06/thw_mask_ab.txt): 10M rows 759.0 / 760.7 k q/s at every 2nd step, 764.2 / 763.4 at every
        // 4th, 767.4 / 767.3 at every 8th; 1.25M rows 0.2346 / 0.2344 -> 0.2353 / 0.2347 -> 0.2357 / 0.2352 ms (the short scans keep every 2nd)
        a.q8_thw_mask = a.slice_rows >= 500000 ? 7 : 1;
        if (kn.q8_rebuild >= 0 && kn.q8_rebuild <= 8) a.q8_rebuild_8ths = kn.q8_rebuild;
        if (kn.q8_target >= 16 && kn.q8_target <= (M == 64 ? 960 : 127)) a.q8_target = kn.q8_target;
        if (kn.q8_tune_ok) {  // ANNLITE_Q8_TUNE = "epoch0,mul,ring_limit,import_mask" (measurements; validated when parsed)
            a.q8_epoch0 = kn.q8_tune[0];
            a.q8_epoch_mul = kn.q8_tune[1];
            a.q8_ring_limit = kn.q8_tune[2];
            a.q8_import_mask = kn.q8_tune[3];
        }
        int grid = a.n_items < n_cu * c.wg_per_cu ? a.n_items : n_cu * c.wg_per_cu;
        const bool sk = codes_layout == ANNLITE_CODES_SKEWED;
        if (c.qf()) {
            // quantise the fp32 tables: min/max -> (step, L, Smax) -> 12-bit codes, all inside the workspace
            const int64_t bpad = pad_queries(B, plan.qt);
            char *wp = (char *)workspace_dev + (((int64_t)a.n_tiles * plan.qt * plan.n_slices * k * 8 + 255) / 256) * 256;
            auto carve = [&](int64_t bytes) { char *r = wp; wp += ((bytes + 255) / 256) * 256; return r; };
            // [gkey][gk2][tile_done] directly behind the partial lists: the one fill covers exactly these four
            unsigned long long *gk = (unsigned long long *)carve(bpad * 8);
            // (the byte-table kernel publishes up to kGk2Keys keys per (query, slice) cell whatever M is)
            unsigned long long *gk2 = (unsigned long long *)carve(bpad * plan.n_slices * 8 * (c.mode == 5 ? kGk2Keys : gk2_cell_keys(M)));
            unsigned int *tile_done = (unsigned int *)carve((int64_t)a.n_tiles * 4);
            unsigned int *item_counter = (unsigned int *)carve(4);
            unsigned int *guard_blk = (unsigned int *)carve(64);  // (reset to all-ones by the fill, like the counters)
            if (gopt) {
                if (c.mode == 5) {
                    a.guard = guard_blk;
                    a.guard_abort = gopt->abort_enabled;
                    a.guard_base = 1024u * (uint32_t)((k + 15) / 16);  // (k > 16: the legitimate candidates grow with k -- the transient alone is ~k per query)
                    if (kn.guard_base >= 0) a.guard_base = (uint32_t)kn.guard_base;  // (ANNLITE_GUARD_BASE, tests: force the give-up path)
                    a.host_stats = gopt->host_stats;
                    a.stats_seq = gopt->seq;
                    gopt->guard_out = guard_blk;
                } else {
                    a.gate = gopt->gate;
                }
            }
            if (tm) {
                a.tile_rows = tm->tile_rows;
                a.vmap = tm->vmap;
                a.item_counter = item_counter;
                a.cand = tm->cand;
                a.cand_count = tm->cand_count;
                a.cand_cap = (int32_t)tm->cand_cap;
                if (outp) outp->merged = true;  // (the tile scan's result is the candidate lists)
            }
            // (64-key lists: the slices are merged by merge_partial_kernel, not in the scan)
            if (share_across_slices && outp && (outp->packed || (outp->d && outp->i)) && !(c.mode == 5 && k > 16)) {
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
                if (c.mode == 5 && k > 16) {
                    // 64-key lists: a slice publishes the keys at four list positions, ~0.64 / 0.96 / 1.28 / 1.92 of its share k / G of
                    // the k best rows (8 slices, k = 50: positions 4, 6, 8, 12 -- the weighted bound sits near rank 56; q8_weighted_bound)
                    const double share = (double)k / (double)grp;
                    double f[4] = {0.64, 0.96, 1.28, 1.92};
                    if (kn.q8_pos_ok)  // ANNLITE_Q8_POS = "f0,f1,f2,f3" (measurements)
                        for (int i = 0; i < 4; ++i) f[i] = kn.q8_pos[i];
                    int prev = 0;
                    a.q8_pos = 0;
                    for (int i = 0; i < 4; ++i) {
                        int p = (int)ceil(f[i] * share);
                        if (p <= prev) p = prev + 1;
                        if (p > 61 + i) p = 61 + i;  // (positions stay inside the 64-key list, strictly ascending)
                        prev = p;
                        a.q8_pos |= (uint32_t)(p - 1) << (8 * i);
                    }
                    a.jm1 = prev - 1;
                }
                a.flush_mask = 63;
                if (kn.flush_mask >= 0) a.flush_mask = kn.flush_mask;
            }
            // (the q16 table sits behind the small arrays; carve order is irrelevant to the kernels)
            // byte-table kernel: no u16 tables, the per-(query, sub-space) minima instead (same region, smaller)
            uint16_t *q16 = c.mode == 5 ? nullptr : (uint16_t *)carve(bpad * M * Ks * 2);
            float *qlom = c.mode == 5 ? (float *)carve(bpad * M * 4) : nullptr;
            // (tile mode with the fused L2 build: the slots' fp32 tables are never read -- do not store them)
            const int64_t Bq = tm ? tm->n_queries : B;  // tile mode: tables of the real queries only
            // seed rows of the shared first bound (tile mode seeds inside the scan kernel)
            int64_t S = 0;
            // (round 6) the candidate generator of the re-rank stage through the ONE preparation launch: its slices share nothing while
            // they scan, but they can START from one bound -- the k-th smallest exact sum among the seed rows spread over the whole
            // table is at or above the k-th key of the TABLE, so no slice can lose a row of the table's top-k to it; what a slice far
            // from the query loses are rows beyond that bound, of no use to a re-rank.  Saves the per-slice seed launch (8 x 8192 rows
            // against 32768), the separate table build / quantise launches and the work items' own table builds (prebuilt images).
            const bool cand_prep = cand_seed && !share_across_slices && c.mode == 5 && build && M == 16 && build->D <= 256 &&
                                   ((build->D / M) % 4) == 0 && !kn.no_fused_seed && !tm && N >= 4096 && k <= 16;
            if ((share_across_slices || cand_prep) && N >= 4096 && !tm) {
                S = 8192;
                // byte-table kernel: its candidate transient shrinks with a tighter first bound faster than the seed launch
                // grows (12 us per 8192 rows): 1.25M rows x 1024 queries 0.425 / 0.407 / 0.405 / 0.437 ms per batch at
                // 8k / 16k / 32k / 64k seed rows, 10M rows 1.852 / 1.838 / 1.831 / 1.857
                if (c.mode == 5 && M != 64)  // (M = 64: two queries per seed workgroup, 63 us per 8192 rows)
                    S = N / 32 < 8192 ? 8192 : N / 32 > 32768 ? 32768 : ((N / 32 + 1023) / 1024) * 1024;
                // 64-key lists (16 < k <= 64): the seed bound's rank in the table is k N / S -- the 32768 rows were tuned at k = 10; at
                // k = 50 they leave the first epochs consumer-bound (wave 0 waits 190 us at the epoch ends of a 10M-row launch, 40 us
                // with 131072 rows).  10M rows x 1024 queries, whole call at 32k / 64k / 128k / 256k seed rows: 1.836 / 1.764 / 1.71 /
                // 1.697 ms (profiles/r05/k50_seed_rows.txt): S x ceil(k / 16), at most a sixteenth of the table
                if (c.mode == 5 && k > 16) {
                    const int64_t s2 = S * ((k + 15) / 16), cap = N / 16 > S ? N / 16 : S;
                    S = ((s2 < cap ? s2 : cap) + 1023) / 1024 * 1024;
                }
                if (kn.seed_rows_set) S = kn.seed_rows;
                if (split && split->seed_rows > 0) S = split->seed_rows;  // (a rank of a row-sharded search: its share of the seed rows)
                if (S > N) S = N;
                if (S < 0) S = 0;  // (ANNLITE_SEED_ROWS=0: the scan starts without a bound)
            }
            // the S seed rows are 64-row blocks spread evenly over the table (ANNLITE_SEED_CONTIGUOUS=1: its first S rows -- A/B switch)
            const int64_t seed_extent = kn.seed_contiguous ? S : N;
            // with a first bound from rows spread over the whole table the early epoch end only costs its barrier (ms per batch with the
            // first end after step 15 / 255 / never: 1.25M rows 0.2387 / 0.2339 / 0.2338, 1M rows 0.2157 / 0.2099 / 0.2099, 10M rows 1.352 /
            // 1.347 / 1.342 -- profiles/r05/epoch_schedule_sweep.txt): the first end moves to step 255; the early one stays where the
            // scan starts without such a bound
            if (c.mode == 5 && S >= 8192 && seed_extent > S && !kn.q8_tune_set) a.q8_epoch0 = 255;
            // byte-table plan behind annlite_pq_search_topk: tables, parameters, reset and seed bound in ONE launch
            const bool one_prep = c.mode == 5 && build && S > 0 && M == 16 && build->D <= 256 && ((build->D / M) % 4) == 0 &&
                                  !kn.no_fused_seed;
            if (split && !one_prep) return ANNLITE_NOT_APPLICABLE;  // (nothing has been launched)
            if (one_prep && split && split->phase == ANNLITE_PHASE_SCAN) {
                // the batch was prepared by the PREPARE half on this workspace: the same carving, no launch
                if (!kn.no_prebuilt_tables) {
                    a.gseed0 = (unsigned long long *)carve(bpad * 8);
                    a.btab = (uint8_t *)carve((int64_t)a.n_tiles * kQ8Image16);
                }
            } else if (one_prep) {
                // ... and the scan work items' FIRST byte tables, once per query tile instead of once per (tile, slice) work item
                // (ANNLITE_NO_PREBUILT_TABLES: the workgroups convert the fp32 tables themselves, as before round 4 -- A/B switch)
                unsigned long long *gseed0 = nullptr;
                uint8_t *btab = nullptr;
                if (!kn.no_prebuilt_tables) {
                    gseed0 = (unsigned long long *)carve(bpad * 8);
                    btab = (uint8_t *)carve((int64_t)a.n_tiles * kQ8Image16);  // (inside the region the u16 plan uses for q16)
                }
                // Round 6, OPT-IN (ANNLITE_MFMA_SEED): the seed rows' exact scan (S x B x M look-up-adds on the VALU: 22 of the launch's
                // 40 us) replaced by an MFMA launch that NOMINATES kSeedCand rows per query (seed_mfma.hip) and the exact sums of those
                // nominees here.  Where it applies: 128-d vectors (8-float sub-vectors: one code word = one MFMA operand half), a batch
                // worth a 128-query workgroup tile, seed rows spread over the table (S rounded UP to a multiple of 8192 <= 131072: the
                // groups), room in the workspace.  Any k distinct valid rows bound the k-th key: results are bit-exact either way.
                // Measured (profiles/r06/mfma_seed_ab.txt): the preparation launch shrinks from 40.3 to 20.6 us, the nomination launch
                // takes more than that back (its A operand is gathered from a bf16 codebook in LDS -- bank conflicts of random codes --
                // behind a 6 us prologue that converts the codebooks): not the default.
                const uint32_t *cand = nullptr;
                int64_t S_prep = S;
                {
                    const int64_t S_m = ((S + 8191) / 8192) * 8192;
                    const size_t cand_bytes = (size_t)bpad * kSeedCand * 4;
                    char *ws_end = (char *)workspace_dev + plan.workspace_bytes;
                    char *cp = (char *)((((uintptr_t)wp + 255) / 256) * 256);
                    if (kn.mfma_seed && build->D == 128 && B >= 64 && S >= 8192 && S_m <= 131072 && S_m <= N && seed_extent == N &&
                        code_bytes == 1 && !(split && split->seed_rows > 0) && cp + cand_bytes <= ws_end) {
                        rc = launch_seed_mfma(codes_layout == ANNLITE_CODES_SKEWED, build->queries, B, build->codebooks, Ks, codes_dev,
                                              valid_bits_dev, N, S_m, knobs().seed_chunk_log, (uint32_t *)cp, st);
                        if (rc != ANNLITE_OK) return rc;
                        cand = (const uint32_t *)cp;
                        wp = cp + cand_bytes;
                        S_prep = S_m;
                    }
                }
                rc = launch_seed_build(codes_layout == ANNLITE_CODES_SKEWED, codes_dev, S_prep, seed_extent, valid_bits_dev, *build, const_cast<float *>(lut_dev),
                                       B, Ks, k, qstep, qlo, smax, qlom, gk, workspace_dev, fill_bytes, (size_t)(((bpad * 8 + 255) / 256) * 256), st,
                                       gseed0, btab, a.q8_target, a.dbg ? g_dbg_prep : nullptr, split ? split->seed_keys : nullptr, cand, kSeedCand);
                if (rc != ANNLITE_OK) return rc;
                a.gseed0 = gseed0;
                a.btab = btab;
                if (split && split->phase == ANNLITE_PHASE_PREPARE) return ANNLITE_OK;  // (the scan follows the seed exchange)
            } else {
                const unsigned int *gate = (gopt && c.mode != 5) ? gopt->gate : nullptr;
                rc = launch_lut_quantise(M, Ks, Bq, ((Bq + 15) / 16) * 16, (tm && build) ? nullptr : lut_dev, build, q16, qstep,
                                         qlo, smax, qlom, workspace_dev, fill_bytes, st, gate);
                if (rc != ANNLITE_OK) return rc;
                if (S > 0) {
                    rc = launch_seed_bound(M, codes_layout == ANNLITE_CODES_SKEWED, codes_dev, code_bytes, S, valid_bits_dev, lut_dev,
                                           B, Ks, k, smax, gk, st, seed_extent, 1, 0, 0, gate);
                    if (rc != ANNLITE_OK) return rc;
                }
            }
            if (!share_across_slices && c.mode == 5 && !tm && N >= 4096 && !(cand_prep && one_prep)) {
                // the byte-table kernel as the candidate generator of the re-rank stage: every slice keeps its own complete
                // list, so every (query, slice) gets its own first bound -- the k-th of the slice's first rows (the gk2 array,
                // unused without sharing and reset by the fill, holds them).  Without one the slice's table would start "open"
                // (everything passes until the first epoch end: 15k rows x 32 queries through the consumer wave).
                int64_t S = a.slice_rows / 64 < 2048 ? 2048 : a.slice_rows / 64 > 8192 ? 8192 : ((a.slice_rows / 64 + 1023) / 1024) * 1024;
                if (kn.seed_rows_set) S = kn.seed_rows;
                if (S > a.slice_rows) S = a.slice_rows;
                if (S > 0) {
                    rc = launch_seed_bound(M, codes_layout == ANNLITE_CODES_SKEWED, codes_dev, code_bytes, S, valid_bits_dev, lut_dev,
                                           B, Ks, k, smax, gk2, st, N, plan.n_slices, a.slice_rows, (int64_t)a.n_tiles * plan.qt);
                    if (rc != ANNLITE_OK) return rc;
                    a.gseed = gk2;
                }
            }
            a.smax = smax;
            a.qstep = qstep;
            a.qlo = qlo;
            a.q16 = q16;
            a.qlom = qlom;
        }
        // early merger (scan_q8.hip: q8_early_merge): every work item must have its own resident workgroup
        a.q8_early_merge = (c.mode == 5 && a.tile_done && a.n_slices >= 2 && a.n_slices <= 31 && a.n_items <= grid &&
                            !kn.no_early_merge) ? 1 : 0;
        a.q8_merge_patience = 20000u;
        if (g_prof_on && c.mode == 5 && !(gopt && gopt->gate)) {
            a.clk = clk_buffer(true);  // (this device's)
        } else if (g_prof_on && !(gopt && gopt->gate)) {
            if (unsigned long long *clk = clk_buffer(false))
                ANNLITE_HIP_TRY(hipMemsetAsync(clk, 0, 32, st));  // (another kernel serves this launch: no stale stamps)
        }
        if (kn.early_merge_patience >= 0) a.q8_merge_patience = (uint32_t)kn.early_merge_patience;
        const bool bracket = !(gopt && gopt->gate);  // (measurement hooks: the launch that does the work, not the gated pass)
        if (bracket) prof_begin(st);
        rc = c.mode == 5 ? launch_q8_scan(c.id, sk, a, grid, st) : launch_qfilter_scan(c.id, sk, a, grid, st);
        if (bracket) prof_end(st);
        return rc;
    }
    if (generic_table_in_lds(M, Ks)) {
        constexpr int NW = 16;
        const size_t tab = (((size_t)M * Ks * 4 + 15) / 16) * 16;
        const size_t lds_need = tab + (size_t)NW * 64 * 8;
        const int per_cu = tab > 64 * 1024 ? 1 : 2;
        const int grid_l = n_items < n_cu * per_cu ? n_items : n_cu * per_cu;
        const int64_t rb = M * code_bytes;
        const bool al16 = rb % 16 == 0 && ((uintptr_t)codes_dev % 16) == 0, al4 = rb % 4 == 0 && ((uintptr_t)codes_dev % 4) == 0;
        prof_begin(st);
#define ANNLITE_LDS_SCAN(T)                                                                                                       \
    {                                                                                                                             \
        auto fn = al16 ? adc_scan_lds_kernel<T, NW, 16> : al4 ? adc_scan_lds_kernel<T, NW, 4> : adc_scan_lds_kernel<T, NW, (int)sizeof(T)>; \
        ANNLITE_HIP_TRY(hipFuncSetAttribute((const void *)fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_need));       \
        hipLaunchKernelGGL(fn, dim3(grid_l), dim3(NW * 64), lds_need, st, a, (int)M);                                             \
    }
        if (code_bytes == 1) ANNLITE_LDS_SCAN(uint8_t)
        else if (code_bytes == 2) ANNLITE_LDS_SCAN(uint16_t)
        else ANNLITE_LDS_SCAN(uint32_t)
#undef ANNLITE_LDS_SCAN
        prof_end(st);
        return launch_status("adc_scan_lds_kernel");
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

extern "C" int annlite_debug_timeline(uint64_t *out8) {
    ANNLITE_REQUIRE(out8 != nullptr, "out8 is NULL");
    if (!g_dbg) {
        set_error("no counters recorded (set ANNLITE_DEBUG_COUNTERS=1 before the scan)");
        return ANNLITE_ERR_INVALID;
    }
    ANNLITE_HIP_TRY(hipDeviceSynchronize());
    ANNLITE_HIP_TRY(hipMemcpy(out8, g_dbg + 8, 64, hipMemcpyDeviceToHost));
    return ANNLITE_OK;
}

extern "C" int annlite_debug_prep_timeline(uint64_t *out8) {
    ANNLITE_REQUIRE(out8 != nullptr, "out8 is NULL");
    if (!g_dbg_prep) {
        set_error("no stamps recorded (set ANNLITE_DEBUG_COUNTERS before a search through annlite_pq_search_topk)");
        return ANNLITE_ERR_INVALID;
    }
    ANNLITE_HIP_TRY(hipDeviceSynchronize());
    ANNLITE_HIP_TRY(hipMemcpy(out8, g_dbg_prep, 64, hipMemcpyDeviceToHost));
    return ANNLITE_OK;
}

extern "C" int annlite_debug_seed_candidates(const float *queries_dev, int64_t B, int64_t D, const float *codebooks_dev,
                                             const void *codes_dev, int codes_layout, int64_t N, int64_t M, int64_t Ks,
                                             const uint32_t *valid_bits_dev, int64_t seed_rows, uint32_t *cand_dev, void *stream) {
    ANNLITE_REQUIRE(queries_dev && codebooks_dev && codes_dev && cand_dev, "null device pointer");
    ANNLITE_REQUIRE(codes_layout == ANNLITE_CODES_PLAIN || codes_layout == ANNLITE_CODES_SKEWED, "bad codes_layout %d", codes_layout);
    if (M != 16 || D != 128 || Ks < 1 || Ks > 256 || B < 1 || seed_rows < 8192 || seed_rows % 8192 != 0 || seed_rows > 131072 || seed_rows > N)
        return ANNLITE_NOT_APPLICABLE;
    return launch_seed_mfma(codes_layout == ANNLITE_CODES_SKEWED, queries_dev, B, codebooks_dev, Ks, codes_dev, valid_bits_dev, N, seed_rows,
                            knobs().seed_chunk_log, cand_dev, (hipStream_t)stream);
}

extern "C" int annlite_debug_items(uint64_t *out, int64_t max_items, int64_t *n_items) {
    ANNLITE_REQUIRE(out != nullptr && n_items != nullptr && max_items >= 0, "null output");
    if (!g_dbg) {
        set_error("no counters recorded (set ANNLITE_DEBUG_COUNTERS=1 before the scan)");
        return ANNLITE_ERR_INVALID;
    }
    ANNLITE_HIP_TRY(hipDeviceSynchronize());
    unsigned long long n = 0;
    ANNLITE_HIP_TRY(hipMemcpy(&n, g_dbg + 15, 8, hipMemcpyDeviceToHost));
    if (n > 4096) n = 4096;
    if ((int64_t)n > max_items) n = (unsigned long long)max_items;
    if (n) ANNLITE_HIP_TRY(hipMemcpy(out, g_dbg + 16, n * 64, hipMemcpyDeviceToHost));
    *n_items = (int64_t)n;
    return ANNLITE_OK;
}

extern "C" int annlite_profile_enable(int on) {
    g_prof_on = on ? 1 : 0;
    g_ev_valid = 0;
    if (on) (void)clk_buffer(true);  // (here, not on the launch path: the allocation synchronises)
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

// Revision of a kernel's memory behaviour (work-item map, table formats, what is read how often): bumped BY HAND with such a
// change.  profiles/traffic.json records it with every PMC pass; bench.py refuses a pass taken on another revision.
extern "C" int annlite_kernel_rev(const char *kernel) {
    ANNLITE_REQUIRE(kernel != nullptr, "kernel is NULL");
    static const struct { const char *name; int rev; } revs[] = {
        {"adc_scan_q8_kernel", 5},        // 4: round 4 (prebuilt byte tables, early merger, M = 64 slice-per-XCD); 5: round 6 (M = 16: the table's
                                          // LDS image is two half tables by sub-space parity -- 128.25 KB copied per work item whatever Ks is)
        {"adc_scan_qfilter_kernel", 1},
        {"adc_scan_qfilter64_kernel", 1},
        {"adc_scan_generic_kernel", 1},
        {"adc_scan_lds_kernel", 1},       // round 6: the generic scan with the query's fp32 table in LDS and 16-byte code-row loads
        {"graph_beam_search_kernel", 3},  // 2: round 5 (packed node records + prefetch, merge insertion, bucketed visited table); 3: round 6 (pair walk:
                                          // two records per step, one per half wave)
    };
    for (const auto &r : revs)
        if (strcmp(r.name, kernel) == 0) return r.rev;
    return 0;  // unknown kernel: no revision (never matches a recorded one)
}

// The shader clock the last profiled byte-table scan actually held: workgroup 0's s_memtime delta (shader cycles on gfx950) over
// its 100 MHz wall-clock delta.  The roofline prices a launch at the nominal 2.4 GHz; a power- or thermally-limited box holds
// less, and a reader of one bench line cannot otherwise tell a slow box from a slow kernel.
extern "C" int annlite_profile_last_scan_clock_mhz(float *mhz) {
    ANNLITE_REQUIRE(mhz != nullptr, "mhz is NULL");
    unsigned long long *clk = clk_buffer(false);  // (of the device that is current: where the profiled launch ran)
    if (!g_ev_valid || !clk) {
        set_error("no byte-table scan has been recorded (call annlite_profile_enable(1) first)");
        return ANNLITE_ERR_INVALID;
    }
    ANNLITE_HIP_TRY(hipEventSynchronize(g_ev1));
    unsigned long long h[4];
    ANNLITE_HIP_TRY(hipMemcpy(h, clk, 32, hipMemcpyDeviceToHost));
    if (h[3] <= h[1] || h[2] <= h[0]) {
        set_error("the last scan left no clock stamps (not a byte-table launch)");
        return ANNLITE_ERR_INVALID;
    }
    *mhz = (float)((double)(h[2] - h[0]) / (double)(h[3] - h[1]) * 100.0);
    return ANNLITE_OK;
}

// =================================================================================================
// Which M = 16 kernel serves a table: byte filter tables (scan_q8.hip: 32 queries per workgroup, the default) or u16 filter
// tables (scan_qfilter.hip).  The byte filter is built for code tables with structure -- what PQ is for: a handful of rows
// per query pass it.  On tables without any (independent uniform codes) it leaks and the u16 kernel is ~10x faster.  The
// choice is made HERE, per call, from what earlier launches measured -- no process-wide switch, nothing the caller sets:
//   * no state (a plain C-ABI consumer): the byte-table launch runs GUARDED -- its consumer waves count the candidates and
//     give the launch up when they exceed a budget (a few microseconds into a leaking scan); a u16-table pass whose three
//     launches are GATED on that flag is queued behind it and redoes the scan.  Without a leak the gated launches return at
//     once (~10 us per batch).
//   * with an annlite_scan_state (one per code table; the Python index owns one): every byte-table launch leaves its
//     candidate count in the state's host-mapped block; the next calls read it (plain host memory, no synchronisation)
//     and, once a launch has completed, run the byte-table kernel unguarded or the u16 kernel directly, until the table has
//     doubled.  The first call(s) run guarded.
// Results are identical whatever is chosen (both kernels are bit-exact).
// =================================================================================================
struct annlite_scan_state {
    unsigned int *host;   // hipHostMalloc'ed, mapped: [0] seq of the last completed byte-table launch, [1] gave up,
    unsigned int *dev;    //   [2..3] candidates seen, [4] B, [5] N  -- and its device address
    uint32_t seq;         // launches issued with this state
    uint32_t seen_seq;    // ... and the last one whose statistics were read
    int kernel;           // 0 undecided, 1 byte tables, 2 u16 tables
    int64_t rows;         // table size the decision was taken at (it is taken again when the table has doubled)
    uint64_t candidates;  // of the launch the decision rests on
    // grey zone (the byte-table launch completed, but with many candidates: e.g. uniform vectors -- 10^4 per query at 10M rows,
    // where it is still 2x the u16-table kernel, while independent random codes -- 8 * 10^4 -- are 10x slower): both kernels are
    // TIMED, one call each, with events in the caller's stream that are read without waiting
    hipEvent_t ev[2][2];  // [0 byte tables, 1 u16 tables][start, stop]
    int probe;            // 0 none, 1 the byte-table call is being timed, 2 waiting for it / the u16 call comes next, 3 waiting for the u16 call
    double work[2];       // tiles x rows of the timed calls (the two may see different batches)
    float ms[2];
};

enum SearchMode { kModePlain, kModeGuarded, kModeByteStats, kModeU16, kModeU16Probe };

static SearchMode search_policy(annlite_scan_state *s, int64_t N, int64_t M, int64_t Ks, int code_bytes, int64_t B, int64_t k,
                                bool tiles) {
    // the shapes with both a byte-table and a u16-table kernel: M = 8 / 16 / 32 with u8 codes, M = 8 / u16 codes up to Ks = 1024
    const bool both = ((M == 16 || M == 8 || M == 32) && code_bytes == 1 && Ks <= 256) || (M == 8 && code_bytes == 2 && Ks <= 1024);
    // (k > 16: the byte-table kernel with 64-key lists exists for M = 16 / uint8 codes; it needs a table worth seeding)
    if (tiles || !both || (k > 16 && !(lk64_shape(M, Ks, code_bytes, k) && N >= 65536)) || N <= 0 || B <= 0) return kModePlain;
    if (g_variant_scope >= 0 || env_variant() >= 0) return kModePlain;        // (an explicit variant: A/B measurements)
    if (knobs().no_inkernel_merge) return kModePlain;                // (debug switch: no guarded pass)
    if (!s) return kModeGuarded;
    const uint32_t seq = __atomic_load_n(s->host, __ATOMIC_ACQUIRE);
    if (seq != s->seen_seq && seq != 0) {  // a byte-table launch has completed since the last look
        s->seen_seq = seq;
        const bool gave_up = s->host[1] != 0;
        const uint64_t cand = (uint64_t)s->host[2] | ((uint64_t)s->host[3] << 32);
        const uint64_t b = s->host[4] ? s->host[4] : 1;
        // what a guarded launch's workgroups compare their own counts with (guard_base + rows drawn / 16 each), summed over
        // the launch: every query tile scans all N rows.  With structure: ~300 candidates per query at 10M rows -- 30x below it
        const uint64_t n_rows = s->host[5];
        const uint64_t kf = (uint64_t)((k + 15) / 16);  // (see guard_base: what counts as "few candidates" scales with k)
        const uint64_t budget = kf * 1024ull * 256ull + n_rows * ((b + 31) / 32) / 16;
        if (s->kernel == 0 && s->probe == 0) {
            s->rows = (int64_t)s->host[5];
            s->candidates = cand;
            if (gave_up) s->kernel = 2;                // the cliff: no need to time anything
            else if (cand <= budget / 16 + kf * 1024ull * b) s->kernel = 1;  // clean: the transient from the seed bound to the converged one (a few
                                                                        // hundred candidates per query whatever N) + structured data's ~1 per 1000 rows and tile
            else s->probe = 1;                          // grey zone: time this kernel now, the other one next
        } else if (s->kernel == 1 && gave_up) {
            s->kernel = 2;
            s->rows = (int64_t)s->host[5];
            s->candidates = cand;
        }
    }
    if (s->kernel != 0 && (N >= 2 * s->rows || N * 2 < s->rows)) s->kernel = 0, s->probe = 0;  // the table has changed size: measure again
    if (s->kernel == 0 && s->probe >= 2) {
        // timed calls: pick up what has completed (never wait)
        const int which = s->probe == 2 ? 0 : 1;
        const hipError_t ready = hipEventQuery(s->ev[which][1]);
        if (ready == hipErrorNotReady) (void)hipGetLastError();  // (not an error of this call: do not leave it behind)
        if (ready == hipSuccess) {
            float ms = 0.f;
            if (hipEventElapsedTime(&ms, s->ev[which][0], s->ev[which][1]) == hipSuccess && ms > 0.f) {
                s->ms[which] = ms;
                if (which == 0) return kModeU16Probe;  // (probe -> 3 when that call has been issued)
                const double r0 = (double)s->ms[0] / s->work[0], r1 = (double)s->ms[1] / s->work[1];
                s->kernel = r1 < r0 ? 2 : 1;
                s->probe = 0;
            } else {
                s->kernel = 1, s->probe = 0;  // (timing unavailable: the launch that completed within its budget stays)
            }
        } else if (which == 1) {
            return kModeU16;  // (until the timed u16 call has completed: results are the same either way)
        }
    }
    if (s->kernel == 2) return kModeU16;
    if (s->kernel == 1) return kModeByteStats;
    return kModeGuarded;
}

extern "C" int annlite_scan_state_create(annlite_scan_state **out) {
    ANNLITE_REQUIRE(out != nullptr, "out is NULL");
    annlite_scan_state *s = (annlite_scan_state *)calloc(1, sizeof(annlite_scan_state));
    ANNLITE_REQUIRE(s != nullptr, "out of memory");
    hipError_t e = hipHostMalloc((void **)&s->host, 64, hipHostMallocMapped);
    if (e == hipSuccess) {
        memset(s->host, 0, 64);
        e = hipHostGetDevicePointer((void **)&s->dev, s->host, 0);
    }
    for (int i = 0; i < 4 && e == hipSuccess; ++i) e = hipEventCreate(&s->ev[i / 2][i % 2]);
    if (e != hipSuccess) {
        if (s->host) (void)hipHostFree(s->host);
        free(s);
        return hip_fail(e, "annlite_scan_state_create");
    }
    *out = s;
    return ANNLITE_OK;
}

extern "C" int annlite_scan_state_destroy(annlite_scan_state *s) {
    if (!s) return ANNLITE_OK;
    if (s->host) (void)hipHostFree(s->host);  // (waits for the device work that may still write it)
    for (int i = 0; i < 4; ++i)
        if (s->ev[i / 2][i % 2]) (void)hipEventDestroy(s->ev[i / 2][i % 2]);
    free(s);
    return ANNLITE_OK;
}

extern "C" int annlite_scan_state_reset(annlite_scan_state *s) {
    ANNLITE_REQUIRE(s != nullptr, "state is NULL");
    s->kernel = 0;
    s->probe = 0;
    s->rows = 0;
    s->candidates = 0;
    s->seen_seq = __atomic_load_n(s->host, __ATOMIC_ACQUIRE);  // (what earlier launches left no longer counts)
    return ANNLITE_OK;
}

extern "C" int annlite_scan_state_info(annlite_scan_state *s, int32_t *kernel, int64_t *rows, uint64_t *candidates) {
    ANNLITE_REQUIRE(s != nullptr, "state is NULL");
    if (kernel) *kernel = s->kernel;
    if (rows) *rows = s->rows;
    if (candidates) *candidates = s->candidates;
    return ANNLITE_OK;
}

static int scan_topk_impl(const void *codes_dev, int code_bytes, int codes_layout, int64_t N, int64_t M, int64_t Ks,
                          const uint32_t *valid_bits_dev, const float *lut_dev, int64_t B, int64_t k, int64_t row_base,
                          float *out_dist_dev, int64_t *out_id_dev, int64_t *out_packed_dev, void *workspace_dev,
                          size_t workspace_bytes, void *stream, const LutBuild *build = nullptr, int flags = 0,
                          const TileMode *tm = nullptr, annlite_scan_state *state = nullptr) {
    annlite_scan_plan plan;
    hipStream_t st = (hipStream_t)stream;
    ANNLITE_REQUIRE(B == 0 || tm || out_packed_dev || (out_dist_dev && out_id_dev), "null output pointer");
    const int sqrt_out = (flags & ANNLITE_FLAG_SQRT) && !out_packed_dev ? 1 : 0;
    ScanOut so = {out_dist_dev, out_id_dev, out_packed_dev, row_base, sqrt_out, false};
    if (knobs().no_inkernel_merge && !tm) so.d = nullptr, so.i = nullptr, so.packed = nullptr;
    const SearchMode mode = search_policy(state, N, M, Ks, code_bytes, B, k, tm != nullptr);
    if (mode != kModePlain) {
        // (workspace: the public plan's size = byte-table region + u16 region; each pass checks its own)
        annlite_scan_plan pub;
        int rc0 = plan_query_impl(N, M, Ks, code_bytes, B, k, 0, &pub);
        if (rc0 != ANNLITE_OK) return rc0;
        if (workspace_bytes < (size_t)pub.workspace_bytes) {
            set_error("workspace %zu B < required %lld B", workspace_bytes, (long long)pub.workspace_bytes);
            return ANNLITE_ERR_WORKSPACE;
        }
        const double work = (double)((B + 31) / 32) * (double)N;
        if (mode == kModeU16 || mode == kModeU16Probe) {
            VariantScope vs(31);
            const bool timed = mode == kModeU16Probe;
            if (timed) ANNLITE_HIP_TRY(hipEventRecord(state->ev[1][0], st));
            const int rc = scan_topk_impl(codes_dev, code_bytes, codes_layout, N, M, Ks, valid_bits_dev, lut_dev, B, k, row_base, out_dist_dev,
                                          out_id_dev, out_packed_dev, workspace_dev, workspace_bytes, stream, build, flags, tm, nullptr);
            if (timed && rc == ANNLITE_OK) {
                ANNLITE_HIP_TRY(hipEventRecord(state->ev[1][1], st));
                state->work[1] = work;
                state->probe = 3;
            }
            return rc;
        }
        const bool timed = state && state->kernel == 0 && state->probe == 1;
        if (timed) ANNLITE_HIP_TRY(hipEventRecord(state->ev[0][0], st));
        GuardOpt g = {};
        g.abort_enabled = mode == kModeGuarded ? 1 : 0;
        if (state) {
            g.host_stats = state->dev;
            g.seq = ++state->seq ? state->seq : ++state->seq;  // (0 means "nothing completed yet")
        }
        int rc;
        size_t used;
        {
            VariantScope vs(50);
            annlite_scan_plan p1;
            rc = scan_partial(codes_dev, code_bytes, codes_layout, N, M, Ks, valid_bits_dev, lut_dev, B, k, workspace_dev,
                              workspace_bytes, st, &p1, true, build, &so, nullptr, &g);
            if (rc != ANNLITE_OK || B == 0) return rc;
            if (!so.merged) {  // 16 < k <= 64: the 64-key lists of the slices, merged here (a launch that gave up leaves garbage: the
                               // gated pass below overwrites it)
                ANNLITE_REQUIRE(k > 16, "the byte-table launch did not merge in-kernel");
                hipLaunchKernelGGL(merge_partial_kernel, dim3((unsigned)((B + 3) / 4)), dim3(256), 0, st, (const unsigned long long *)workspace_dev,
                                   (int)B, p1.n_slices, (int)k, row_base, out_dist_dev, out_id_dev, out_packed_dev, sqrt_out);
                rc = launch_status("merge_partial_kernel");
                if (rc != ANNLITE_OK) return rc;
            }
            used = ((size_t)p1.workspace_bytes + 255) / 256 * 256;
        }
        if (mode == kModeGuarded && g.guard_out) {
            // the same scan through the u16-table kernel, every launch of it gated on the byte-table launch having given up
            VariantScope vs(31);
            annlite_scan_plan p2;
            GuardOpt g2 = {};
            g2.gate = g.guard_out;
            ScanOut so2 = {out_dist_dev, out_id_dev, out_packed_dev, row_base, sqrt_out, false};
            rc = scan_partial(codes_dev, code_bytes, codes_layout, N, M, Ks, valid_bits_dev, lut_dev, B, k, (char *)workspace_dev + used,
                              workspace_bytes - used, st, &p2, true, nullptr, &so2, nullptr, &g2);
            if (rc != ANNLITE_OK) return rc;
            ANNLITE_REQUIRE(so2.merged, "the gated u16-table launch did not merge in-kernel");
        }
        if (timed) {
            ANNLITE_HIP_TRY(hipEventRecord(state->ev[0][1], st));
            state->work[0] = work;
            state->probe = 2;
        }
        return ANNLITE_OK;
    }
    int rc = scan_partial(codes_dev, code_bytes, codes_layout, N, M, Ks, valid_bits_dev, lut_dev, B, k, workspace_dev,
                          workspace_bytes, st, &plan, true, build, &so, tm);
    if (rc != ANNLITE_OK || B == 0 || so.merged) return rc;
    ANNLITE_REQUIRE(!tm, "tile mode: the scan did not merge in-kernel");
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

static int pq_search_impl(int lut_kind, const float *queries_dev, int64_t B, int64_t D, const float *codebooks_dev,
                          const void *codes_dev, int code_bytes, int codes_layout, int64_t N, int64_t M, int64_t Ks,
                          const uint32_t *valid_bits_dev, int64_t k, int64_t row_base, float *out_dist_dev,
                          int64_t *out_id_dev, int64_t *out_packed_dev, int flags, void *workspace_dev,
                          size_t workspace_bytes, void *stream, const TileMode *tm, int64_t n_slots = 0,
                          annlite_scan_state *state = nullptr) {
    ANNLITE_REQUIRE(M >= 1 && D >= M && D % M == 0,
                    "input dimension must be dividable by number of sub-space (D=%lld, M=%lld)", (long long)D, (long long)M);
    // tile mode: B real queries (tables), n_slots >= B scan slots (lists, bounds)
    const int64_t Bs = tm ? n_slots : B;
    annlite_scan_plan plan;
    int rc = plan_query_impl(N, M, Ks, code_bytes, Bs, k, tm ? 1 : 0, &plan);
    if (rc != ANNLITE_OK) return rc;
    if (Bs == 0 || B == 0) return ANNLITE_OK;
    const size_t scan_ws = r256z((size_t)plan.workspace_bytes);
    const size_t need = scan_ws + r256z((size_t)plan.lut_floats * 4);
    if (workspace_bytes < need) {
        set_error("workspace %zu B < required %zu B (annlite_pq_search_workspace_bytes)", workspace_bytes, need);
        return ANNLITE_ERR_WORKSPACE;
    }
    ANNLITE_REQUIRE(queries_dev && codebooks_dev && workspace_dev, "null device pointer");
    ANNLITE_REQUIRE(!tm || B <= Bs, "tile mode: more queries (%lld) than slots (%lld)", (long long)B, (long long)Bs);
    float *lut = (float *)((char *)workspace_dev + scan_ws);
    FastCfg c;
    const bool fuse = N > 0 && plan.fast && fast_cfg(M, Ks, code_bytes, k, &c, tm != nullptr) && c.qf() && M != 64 && Ks <= 256 &&
                      lut_kind == ANNLITE_LUT_L2 && ((D / M) % 4) == 0 && !knobs().no_fused_lut;
    if (!fuse) {
        rc = annlite_lut_build(lut_kind, queries_dev, B, D, codebooks_dev, M, Ks, lut,
                               plan.fast ? ANNLITE_LAYOUT_TILED : ANNLITE_LAYOUT_BMK, plan.qi, stream);
        if (rc != ANNLITE_OK) return rc;
        return scan_topk_impl(codes_dev, code_bytes, codes_layout, N, M, Ks, valid_bits_dev, lut, Bs, k, row_base,
                              out_dist_dev, out_id_dev, out_packed_dev, workspace_dev, scan_ws, stream, nullptr, flags, tm, state);
    }
    const LutBuild lb = {queries_dev, codebooks_dev, D};
    return scan_topk_impl(codes_dev, code_bytes, codes_layout, N, M, Ks, valid_bits_dev, lut, Bs, k, row_base, out_dist_dev,
                          out_id_dev, out_packed_dev, workspace_dev, scan_ws, stream, &lb, flags, tm, state);
}

extern "C" int annlite_pq_search_topk_ex(int lut_kind, const float *queries_dev, int64_t B, int64_t D,
                                         const float *codebooks_dev, const void *codes_dev, int code_bytes, int codes_layout,
                                         int64_t N, int64_t M, int64_t Ks, const uint32_t *valid_bits_dev, int64_t k,
                                         int64_t row_base, float *out_dist_dev, int64_t *out_id_dev, int64_t *out_packed_dev,
                                         int flags, void *workspace_dev, size_t workspace_bytes, void *stream,
                                         annlite_scan_state *state) {
    return pq_search_impl(lut_kind, queries_dev, B, D, codebooks_dev, codes_dev, code_bytes, codes_layout, N, M, Ks,
                          valid_bits_dev, k, row_base, out_dist_dev, out_id_dev, out_packed_dev, flags, workspace_dev,
                          workspace_bytes, stream, nullptr, 0, state);
}

extern "C" int annlite_pq_search_topk(int lut_kind, const float *queries_dev, int64_t B, int64_t D,
                                      const float *codebooks_dev, const void *codes_dev, int code_bytes, int codes_layout,
                                      int64_t N, int64_t M, int64_t Ks, const uint32_t *valid_bits_dev, int64_t k,
                                      int64_t row_base, float *out_dist_dev, int64_t *out_id_dev, int64_t *out_packed_dev,
                                      int flags, void *workspace_dev, size_t workspace_bytes, void *stream) {
    return pq_search_impl(lut_kind, queries_dev, B, D, codebooks_dev, codes_dev, code_bytes, codes_layout, N, M, Ks,
                          valid_bits_dev, k, row_base, out_dist_dev, out_id_dev, out_packed_dev, flags, workspace_dev,
                          workspace_bytes, stream, nullptr);
}

// ---- the search in two halves with the ranks' seed exchange in between (annlite_hip.h: annlite_pq_search_split) ----------------
static bool split_shape_ok(int lut_kind, int64_t N, int64_t M, int64_t Ks, int code_bytes, int64_t B, int64_t D, int64_t k) {
    return lut_kind == ANNLITE_LUT_L2 && M == 16 && code_bytes == 1 && Ks <= 256 && k >= 1 && k <= 16 && B >= 1 && N >= 4096 &&
           D <= 256 && D % M == 0 && ((D / M) % 4) == 0 && !knobs().no_fused_seed && !knobs().no_fused_lut &&
           !knobs().no_inkernel_merge && env_variant() < 0;
}

extern "C" int annlite_pq_search_split(int phase, int64_t seed_rows, int lut_kind, const float *queries_dev, int64_t B, int64_t D,
                                       const float *codebooks_dev, const void *codes_dev, int code_bytes, int codes_layout,
                                       int64_t N, int64_t M, int64_t Ks, const uint32_t *valid_bits_dev, int64_t k,
                                       int64_t row_base, float *out_dist_dev, int64_t *out_id_dev, int64_t *out_packed_dev,
                                       int flags, void *workspace_dev, size_t workspace_bytes, void *stream,
                                       annlite_scan_state *state, uint64_t *seed_keys_dev) {
    ANNLITE_REQUIRE(phase == ANNLITE_PHASE_PREPARE || phase == ANNLITE_PHASE_SCAN, "phase must be ANNLITE_PHASE_PREPARE or _SCAN");
    ANNLITE_REQUIRE(state != nullptr, "the split search needs the table's annlite_scan_state");
    if (!split_shape_ok(lut_kind, N, M, Ks, code_bytes, B, D, k)) return ANNLITE_NOT_APPLICABLE;
    if (phase == ANNLITE_PHASE_PREPARE) {
        ANNLITE_REQUIRE(seed_keys_dev != nullptr, "seed_keys_dev is NULL");
        // only once the library has settled on the byte-table kernel for this table (no guarded second pass to split)
        if (search_policy(state, N, M, Ks, code_bytes, B, k, false) != kModeByteStats) return ANNLITE_NOT_APPLICABLE;
    } else {
        ANNLITE_REQUIRE(out_packed_dev || (out_dist_dev && out_id_dev), "null output pointer");
    }
    VariantScope vs(50);
    annlite_scan_plan plan;
    int rc = plan_query_impl(N, M, Ks, code_bytes, B, k, 0, &plan);
    if (rc != ANNLITE_OK) return rc;
    const size_t scan_ws = r256z((size_t)plan.workspace_bytes);
    const size_t need = scan_ws + r256z((size_t)plan.lut_floats * 4);
    if (workspace_bytes < need) {
        set_error("workspace %zu B < required %zu B (annlite_pq_search_workspace_bytes)", workspace_bytes, need);
        return ANNLITE_ERR_WORKSPACE;
    }
    ANNLITE_REQUIRE(queries_dev && codebooks_dev && workspace_dev && codes_dev, "null device pointer");
    float *lut = (float *)((char *)workspace_dev + scan_ws);
    const LutBuild lb = {queries_dev, codebooks_dev, D};
    const int sqrt_out = (flags & ANNLITE_FLAG_SQRT) && !out_packed_dev ? 1 : 0;
    ScanOut so = {out_dist_dev, out_id_dev, out_packed_dev, row_base, sqrt_out, false};
    if (phase == ANNLITE_PHASE_PREPARE) so.packed = (int64_t *)seed_keys_dev;  // (any non-NULL output: the prepare half writes none)
    GuardOpt g = {};
    g.abort_enabled = 0;
    g.host_stats = state->dev;
    if (phase == ANNLITE_PHASE_SCAN) g.seq = ++state->seq ? state->seq : ++state->seq;
    const SplitOpt sp = {phase, seed_rows, (unsigned long long *)seed_keys_dev};
    annlite_scan_plan p1;
    rc = scan_partial(codes_dev, code_bytes, codes_layout, N, M, Ks, valid_bits_dev, lut, B, k, workspace_dev, scan_ws,
                      (hipStream_t)stream, &p1, true, &lb, &so, nullptr, &g, &sp);
    if (rc != ANNLITE_OK) return rc;
    if (phase == ANNLITE_PHASE_SCAN) ANNLITE_REQUIRE(so.merged, "the byte-table launch did not merge in-kernel");
    return ANNLITE_OK;
}

extern "C" int annlite_pq_search_seed_union(const uint64_t *all_keys_dev, int64_t G, int64_t N, int64_t M, int64_t Ks,
                                            int code_bytes, int64_t B, int64_t k, void *workspace_dev, size_t workspace_bytes,
                                            void *stream) {
    ANNLITE_REQUIRE(all_keys_dev && workspace_dev, "null device pointer");
    ANNLITE_REQUIRE(G >= 1 && G <= 8, "G=%lld outside [1, 8] (one node)", (long long)G);
    ANNLITE_REQUIRE(k >= 1 && k <= kSeedKeys && B >= 1, "bad B=%lld k=%lld", (long long)B, (long long)k);
    VariantScope vs(50);
    annlite_scan_plan plan;
    int rc = plan_query_impl(N, M, Ks, code_bytes, B, k, 0, &plan);
    if (rc != ANNLITE_OK) return rc;
    ANNLITE_REQUIRE(plan.fast && plan.qt == 32, "no byte-table plan for this shape");
    // the shared bounds sit right behind the per-(query, slice) lists (scan_partial's carving)
    const int64_t n_tiles = (B + plan.qt - 1) / plan.qt;
    const size_t off = r256z((size_t)n_tiles * plan.qt * plan.n_slices * k * 8);
    ANNLITE_REQUIRE(off + (size_t)pad_queries(B, plan.qt) * 8 <= workspace_bytes, "workspace too small");
    unsigned long long *gk = (unsigned long long *)((char *)workspace_dev + off);
    hipLaunchKernelGGL(seed_union_kernel, dim3((unsigned)((B + 3) / 4)), dim3(256), 0, (hipStream_t)stream,
                       (const unsigned long long *)all_keys_dev, (int)G, (int)B, (int)k, gk);
    return launch_status("seed_union_kernel");
}

extern "C" int annlite_pq_search_tiles_workspace_bytes(int64_t N, int64_t M, int64_t Ks, int code_bytes, int64_t V,
                                                       int64_t k, int64_t *bytes) {
    ANNLITE_REQUIRE(bytes != nullptr, "bytes is NULL");
    annlite_scan_plan plan;
    int rc = plan_query_impl(N, M, Ks, code_bytes, V, k, 1, &plan);
    if (rc != ANNLITE_OK) return rc;
    *bytes = (int64_t)(r256z((size_t)plan.workspace_bytes) + r256z((size_t)plan.lut_floats * 4));
    return ANNLITE_OK;
}

extern "C" int annlite_pq_search_tiles(int lut_kind, const float *queries_dev, int64_t B, int64_t D,
                                       const float *codebooks_dev, const void *codes_dev, int code_bytes, int codes_layout,
                                       int64_t N, int64_t M, int64_t Ks, const uint32_t *valid_bits_dev, int64_t k,
                                       int64_t V, const int64_t *tile_rows_dev, const int32_t *vmap_dev, uint32_t *cand_dev,
                                       int64_t cand_cap, uint32_t *cand_count_dev, void *workspace_dev,
                                       size_t workspace_bytes, void *stream) {
    ANNLITE_REQUIRE(V == 0 || (tile_rows_dev && vmap_dev && cand_dev && cand_count_dev), "null tile table / output");
    ANNLITE_REQUIRE(N > 0 || V == 0, "tile mode needs a non-empty code table");
    ANNLITE_REQUIRE(cand_cap >= 64 && cand_cap < (1ll << 30), "cand_cap=%lld outside [64, 2^30)", (long long)cand_cap);
    const TileMode tm = {tile_rows_dev, vmap_dev, cand_dev, cand_count_dev, cand_cap, B};
    return pq_search_impl(lut_kind, queries_dev, B, D, codebooks_dev, codes_dev, code_bytes, codes_layout, N, M, Ks,
                          valid_bits_dev, k, 0, nullptr, nullptr, nullptr, 0, workspace_dev, workspace_bytes, stream, &tm, V);
}

// ---- pruned search over cells on the byte-table kernel (round 6) --------------------------------------------------------
// annlite_ivf_plan (tiles of 32 slots) -> the preparation launch with per-cell seeds and per-query byte tables (launch_seed_build_cells)
// -> adc_scan_q8_kernel<16, ..., TL> (id 1651: exact sums in the tile, lists per slot, bounds shared by QUERY) -> annlite_ivf_merge_lists.
// Workspace carve (all 256-byte aligned): lists u64 [V][k] | gkey u64 [bpad] | gseed0 u64 [bpad] | qstep f32 | smax f32 | qlo f64 | qlom f32 [bpad][16] |
// fp32 TILED tables [bpad][Ks][16] | bq u8 [bpad][Ks][16] | vmap i32 [V] | slot_of i32 [B * P] | tile_rows i64 [T][2] | n_tiles_used
constexpr int kIvfQt = 32;
static size_t ivf_topk_carve(int64_t B, int64_t P, int64_t C, int64_t Ks, int64_t k, char *base, char **ptrs /* [13] */) {
    const int64_t bpad = pad_queries(B, kIvfQt);
    const int64_t T = annlite_ivf_max_tiles_first(B, P, C, kIvfQt), V = T * kIvfQt;
    const int64_t sizes[13] = {V * k * 8, bpad * 8, bpad * 8, bpad * 4, bpad * 4, bpad * 8, bpad * 16 * 4, bpad * Ks * 16 * 4, bpad * Ks * 16,
                               V * 4, B * P * 4, T * 16, 256};
    size_t off = 0;
    for (int i = 0; i < 13; ++i) {
        if (ptrs) ptrs[i] = base + off;
        off += r256z((size_t)sizes[i]);
    }
    return off;
}

extern "C" int annlite_ivf_search_topk_workspace_bytes(int64_t B, int64_t P, int64_t C, int64_t M, int64_t Ks, int64_t k, int64_t *bytes) {
    ANNLITE_REQUIRE(bytes != nullptr, "bytes is NULL");
    ANNLITE_REQUIRE(B >= 0 && P >= 1 && C >= 1 && P <= C && M == 16 && Ks >= 1 && Ks <= 256 && k >= 1 && k <= 16,
                    "annlite_ivf_search_topk serves M = 16, Ks <= 256, k <= 16 (B=%lld P=%lld C=%lld M=%lld Ks=%lld k=%lld)", (long long)B,
                    (long long)P, (long long)C, (long long)M, (long long)Ks, (long long)k);
    *bytes = (int64_t)ivf_topk_carve(B, P, C, Ks, k, nullptr, nullptr);
    return ANNLITE_OK;
}

// cand_ids_dev != NULL (annlite_ivf_search_candidates): the slots keep PRIVATE lists -- nothing is shared between a query's tiles, every
// (query, cell) list is the cell's best <= k rows at or below the query's first bound -- and the lists go out as ids, unmerged
static int ivf_search_impl(int lut_kind, const float *queries_dev, int64_t B, int64_t D, const float *codebooks_dev, int64_t M, int64_t Ks,
                           const void *codes_dev, int codes_layout, int64_t N, const uint32_t *valid_bits_dev,
                           const int32_t *cells_dev, int64_t P, int64_t C, const int64_t *cell_rows_dev,
                           const int32_t *cell_order_dev, const int64_t *row_ids_dev, int64_t id_base, int64_t k,
                           float *out_dist_dev, int64_t *out_id_dev, int flags, void *workspace_dev, size_t workspace_bytes,
                           void *stream, int64_t *cand_ids_dev, int64_t bound_rank, const int32_t *seed_cells_dev) {
    ANNLITE_REQUIRE(M == 16 && Ks >= 1 && Ks <= 256 && k >= 1 && k <= 16,
                    "annlite_ivf_search_topk serves M = 16, Ks <= 256, k <= 16 (got M=%lld Ks=%lld k=%lld): ANNLITE_NOT_APPLICABLE shapes take "
                    "annlite_pq_search_tiles + annlite_ivf_rescore", (long long)M, (long long)Ks, (long long)k);
    ANNLITE_REQUIRE(lut_kind == ANNLITE_LUT_L2 || lut_kind == ANNLITE_LUT_IPDIST,
                    "lut_kind must be ANNLITE_LUT_L2 or ANNLITE_LUT_IPDIST (the tables get_dist_mat builds), got %d", lut_kind);
    ANNLITE_REQUIRE(D >= M && D % M == 0 && D <= 256 && ((D / M) % 4) == 0,
                    "the fused table build needs D <= 256 and sub-vectors of a multiple of 4 floats (D=%lld)", (long long)D);
    ANNLITE_REQUIRE(B >= 0 && P >= 1 && C >= 1 && P <= C && C <= 16384 && N > 0 && N < (1ll << 31),
                    "bad shape B=%lld P=%lld C=%lld N=%lld", (long long)B, (long long)P, (long long)C, (long long)N);
    ANNLITE_REQUIRE(codes_layout == ANNLITE_CODES_PLAIN || codes_layout == ANNLITE_CODES_SKEWED, "codes_layout %d", codes_layout);
    if (B == 0) return ANNLITE_OK;
    ANNLITE_REQUIRE(queries_dev && codebooks_dev && codes_dev && cells_dev && cell_rows_dev && cell_order_dev &&
                        (cand_ids_dev || (out_dist_dev && out_id_dev)) && workspace_dev,
                    "null device pointer");
    char *ptr[13];
    const size_t need = ivf_topk_carve(B, P, C, Ks, k, (char *)workspace_dev, ptr);
    if (workspace_bytes < need) {
        set_error("workspace %zu B < required %zu B (annlite_ivf_search_topk_workspace_bytes)", workspace_bytes, need);
        return ANNLITE_ERR_WORKSPACE;
    }
    hipStream_t st = (hipStream_t)stream;
    const int64_t T = annlite_ivf_max_tiles_first(B, P, C, kIvfQt);
    unsigned long long *lists = (unsigned long long *)ptr[0], *gkey = (unsigned long long *)ptr[1], *gseed0 = (unsigned long long *)ptr[2];
    float *qstep = (float *)ptr[3], *smax = (float *)ptr[4], *qlom = (float *)ptr[6], *lut = (float *)ptr[7];
    double *qlo = (double *)ptr[5];
    uint8_t *bq = (uint8_t *)ptr[8];
    int32_t *vmap = (int32_t *)ptr[9], *slot_of = (int32_t *)ptr[10], *n_used = (int32_t *)ptr[12];
    int64_t *tile_rows = (int64_t *)ptr[11];
    const Knobs &kn = knobs();
    // the tiles of every query's nearest cells first (ANNLITE_IVF_FIRST = 0 .. P - 1 probes in the first class; measurements)
    // (measured, profiles/r06/ivf_first_tiles_ab.txt: slower at 1 / 2 / 4 -- the first class adds a scan of every cell for a few slots each
    // and the candidates do not shrink in proportion: off by default)
    int64_t n_first = kn.ivf_first >= 0 ? kn.ivf_first : 0;
    if (n_first >= P) n_first = 0;
    // (the plan as an extra workgroup of the preparation launch was measured too: that launch fills every CU with one workgroup each, the
    // extra one ran behind them -- 93.7 us against 69 + 24)
    int rc = annlite_ivf_plan_first(cells_dev, B, P, C, kIvfQt, cell_rows_dev, cell_order_dev, T, vmap, slot_of, tile_rows, n_used, n_first, stream);
    if (rc != ANNLITE_OK) return rc;
    // seed rows: S / 4 per query from its nearest cell (the workgroup's four queries share the launch's blocks)
    // (10M rows, 16 of 256 cells, scan kernel / whole call at 16k / 32k / 64k / 128k: 0.307 / 0.263 / 0.212 / 0.193 and 0.378 / 0.347 / 0.323 / 0.349 ms)
    // (call 50, same box, with the table target below: 65536 rows 0.3331 / 0.2783 ms per batch on one / two caller streams, 49152 rows + target 72
    // 0.3260 / 0.2666)
    int64_t S = 49152;
    if (kn.seed_rows_set && kn.seed_rows >= 256) S = kn.seed_rows;
    if (S > N) S = N;  // (the launch takes max(N, S) as the table's extent: S must not exceed it)
    // T of a slot's freshly built byte table: a cell tile never rebuilds and its bound falls by more than a long scan's, so a coarser
    // step than the exhaustive scan's 88 (fewer entries of the rows near the bound clip at 15 steps): scan kernel at 64 / 72 / 80 / 88:
    // 0.1927 / 0.1904 / 0.1964 / 0.2011 ms (profiles/r06/ivf_target_seed.txt)
    const int target = (kn.q8_target >= 16 && kn.q8_target <= 127) ? kn.q8_target : 72;
    const LutBuild lb = {queries_dev, codebooks_dev, D};
    const bool sk = codes_layout == ANNLITE_CODES_SKEWED;
    unsigned int *item_counter = (unsigned int *)(n_used + 16);
    // candidate generator: the first bound at the (bound_rank x k)-th smallest seed sum (at most the 64th: the seed lists' length).  Measured
    // at 10M rows, 16 of 256 cells, k = 16 (profiles/r06/ivf_cand_rank.txt): rank 1 / 2 / 4 -> 2.44 / 1.72 / 1.21 M q/s on two streams at
    // re-ranked recall@10 0.803 / 0.813 / 0.813 -- a looser bound only lengthens the far cells' lists (cut at k rows per cell either way)
    int64_t k_seed = k;
    if (cand_ids_dev) k_seed = k * bound_rank > 64 ? 64 : k * bound_rank;
    rc = launch_seed_build_cells(sk, codes_dev, S, N, valid_bits_dev, lb, lut, B, Ks, k_seed, qstep, qlo, smax, qlom, gkey, st, gseed0, bq, target,
                                 cells_dev, P, cell_rows_dev, item_counter, lut_kind == ANNLITE_LUT_IPDIST, seed_cells_dev);
    if (rc != ANNLITE_OK) return rc;
    ScanArgs a = {};
    a.codes = codes_dev;
    a.valid = valid_bits_dev;
    a.lut = lut;
    a.partial = lists;
    a.N = N;
    a.Ks = (int)Ks;
    a.B = (int)B;
    a.k = (int)k;
    a.n_tiles = (int)T;
    a.n_slices = 1;
    a.n_items = (int)T;
    a.slice_rows = ((N + 63) / 64) * 64;
    a.smax = smax;
    a.qstep = qstep;
    a.qlo = qlo;
    a.qlom = qlom;
    a.gkey = gkey;
    a.gk2 = nullptr;
    a.jm1 = 0;
    a.tile_rows = tile_rows;
    a.vmap = vmap;
    a.item_counter = kn.ivf_static_tiles ? nullptr : item_counter;  // (NULL: the tiles dealt round-robin -- A/B)
    a.tl_private = cand_ids_dev ? 1 : 0;
    a.gseed0 = gseed0;
    a.btab = bq;
    a.q8_epoch0 = 1 << 28;  // (no epoch ends in a cell tile)
    a.q8_epoch_mul = 1;
    a.q8_ring_limit = 384;
    a.q8_import_mask = 3;
    a.q8_target = target;
    a.q8_rebuild_8ths = 4;
    a.q8_thw_mask = 1;
    a.flush_mask = 63;
    a.dbg_skip = kn.debug_skip;
    if (kn.debug_counters) {  // (ANNLITE_DEBUG_COUNTERS: the scan's event counters / per-item stamps, as scan_partial wires them)
        if (!g_dbg) ANNLITE_HIP_TRY(hipMalloc((void **)&g_dbg, 128 + 4096 * 64));
        ANNLITE_HIP_TRY(hipMemsetAsync(g_dbg, 0, 128, st));
        a.dbg = g_dbg;
        if (kn.debug_counters == 2) a.dbg_skip |= 8;
    }
    if (kn.q8_tune_ok) a.q8_import_mask = kn.q8_tune[3];
    const int n_cu = device_cu_count();
    const int grid = (int)(T < n_cu ? T : n_cu);
    prof_begin(st);
    rc = launch_q8_scan(1651, sk, a, grid, st);
    prof_end(st);
    if (rc != ANNLITE_OK) return rc;
    if (cand_ids_dev) return launch_ivf_lists_to_ids((const unsigned long long *)lists, k, slot_of, B, P, row_ids_dev, id_base, cand_ids_dev, st);
    return annlite_ivf_merge_lists((const uint64_t *)lists, k, slot_of, B, P, row_ids_dev, id_base, out_dist_dev, out_id_dev, flags, stream);
}

extern "C" int annlite_ivf_search_topk(int lut_kind, const float *queries_dev, int64_t B, int64_t D, const float *codebooks_dev, int64_t M, int64_t Ks,
                                       const void *codes_dev, int codes_layout, int64_t N, const uint32_t *valid_bits_dev,
                                       const int32_t *cells_dev, int64_t P, int64_t C, const int64_t *cell_rows_dev,
                                       const int32_t *cell_order_dev, const int64_t *row_ids_dev, int64_t id_base, int64_t k,
                                       float *out_dist_dev, int64_t *out_id_dev, int flags, void *workspace_dev, size_t workspace_bytes,
                                       void *stream) {
    return ivf_search_impl(lut_kind, queries_dev, B, D, codebooks_dev, M, Ks, codes_dev, codes_layout, N, valid_bits_dev, cells_dev, P, C,
                           cell_rows_dev, cell_order_dev, row_ids_dev, id_base, k, out_dist_dev, out_id_dev, flags, workspace_dev, workspace_bytes,
                           stream, nullptr, 1, nullptr);
}

extern "C" int annlite_ivf_search_candidates(int lut_kind, const float *queries_dev, int64_t B, int64_t D, const float *codebooks_dev, int64_t M,
                                             int64_t Ks, const void *codes_dev, int codes_layout, int64_t N, const uint32_t *valid_bits_dev,
                                             const int32_t *cells_dev, int64_t P, int64_t C, const int64_t *cell_rows_dev,
                                             const int32_t *cell_order_dev, const int64_t *row_ids_dev, int64_t id_base, int64_t k,
                                             int64_t bound_rank, const int32_t *seed_cells_dev, int64_t *out_ids_dev, void *workspace_dev,
                                             size_t workspace_bytes, void *stream) {
    ANNLITE_REQUIRE(out_ids_dev != nullptr || B == 0, "out_ids_dev is NULL");
    ANNLITE_REQUIRE(bound_rank >= 1 && bound_rank <= 64, "bound_rank %lld (1 .. 64)", (long long)bound_rank);
    return ivf_search_impl(lut_kind, queries_dev, B, D, codebooks_dev, M, Ks, codes_dev, codes_layout, N, valid_bits_dev, cells_dev, P, C,
                           cell_rows_dev, cell_order_dev, row_ids_dev, id_base, k, nullptr, nullptr, 0, workspace_dev, workspace_bytes, stream,
                           out_ids_dev, bound_rank, seed_cells_dev);
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

// The candidate generator with the tables built inside the call (round 6): annlite_lut_build + annlite_adc_scan_candidates in one
// C call, and -- M = 16, L2 tables, 4-float-aligned sub-vectors, k <= 16 -- through the ONE preparation launch of the plain search
// (tables, quantisation parameters, list reset, ONE seed bound from rows spread over the table, prebuilt byte tables).
extern "C" int annlite_pq_search_candidates(int lut_kind, const float *queries_dev, int64_t B, int64_t D, const float *codebooks_dev,
                                            const void *codes_dev, int code_bytes, int codes_layout, int64_t N, int64_t M, int64_t Ks,
                                            const uint32_t *valid_bits_dev, int64_t k, int64_t row_base, float *out_dist_dev,
                                            int64_t *out_id_dev, void *workspace_dev, size_t workspace_bytes, void *stream) {
    ANNLITE_REQUIRE(M >= 1 && D >= M && D % M == 0,
                    "input dimension must be dividable by number of sub-space (D=%lld, M=%lld)", (long long)D, (long long)M);
    annlite_scan_plan plan;
    int rc = plan_query_impl(N, M, Ks, code_bytes, B, k, 0, &plan);
    if (rc != ANNLITE_OK) return rc;
    if (B == 0) return ANNLITE_OK;
    const size_t scan_ws = r256z((size_t)plan.workspace_bytes);
    const size_t need = scan_ws + r256z((size_t)plan.lut_floats * 4);
    if (workspace_bytes < need) {
        set_error("workspace %zu B < required %zu B (annlite_pq_search_workspace_bytes)", workspace_bytes, need);
        return ANNLITE_ERR_WORKSPACE;
    }
    ANNLITE_REQUIRE(queries_dev && codebooks_dev && workspace_dev && out_dist_dev && out_id_dev, "null device pointer");
    hipStream_t st = (hipStream_t)stream;
    float *lut = (float *)((char *)workspace_dev + scan_ws);
    FastCfg c;
    const bool fuse = N > 0 && plan.fast && fast_cfg(M, Ks, code_bytes, k, &c, false) && c.qf() && M != 64 && Ks <= 256 &&
                      lut_kind == ANNLITE_LUT_L2 && ((D / M) % 4) == 0 && !knobs().no_fused_lut;
    const LutBuild lb = {queries_dev, codebooks_dev, D};
    if (!fuse) {
        rc = annlite_lut_build(lut_kind, queries_dev, B, D, codebooks_dev, M, Ks, lut, plan.fast ? ANNLITE_LAYOUT_TILED : ANNLITE_LAYOUT_BMK,
                               plan.qi, stream);
        if (rc != ANNLITE_OK) return rc;
    }
    annlite_scan_plan used;
    rc = scan_partial(codes_dev, code_bytes, codes_layout, N, M, Ks, valid_bits_dev, lut, B, k, workspace_dev, scan_ws, st, &used, false,
                      fuse ? &lb : nullptr, nullptr, nullptr, nullptr, nullptr, /*cand_seed=*/fuse && !knobs().no_cand_seed);
    if (rc != ANNLITE_OK) return rc;
    const int64_t total = B * used.n_slices * k;
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
