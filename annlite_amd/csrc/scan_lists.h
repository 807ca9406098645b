// scan_lists.h -- what the quantised-filter scan kernels (scan_qfilter.hip, scan_q8.hip) share: the shared top-k
// lists in LDS, the exact fp32 recompute of queued candidates, and the in-kernel merge of a tile's row slices.
#pragma once
#include "scan_common.h"

namespace annlite {

// The top-k of a (workgroup, query) lives ONCE in LDS: 64 sorted (key, id) entries + a 4-byte lock.  The
// bound every wave filters with is the k-th best of ALL rows the workgroup has seen.
extern __shared__ __attribute__((aligned(16))) unsigned char g_smem[];
constexpr int kTileCandBytes = 12288;  // tile mode: LDS bytes of the per-slot candidate buffers (all slots together)

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
    // byte-entry kernel (scan_q8.hip): the filter bounds are bytes (0x80 | T) at shq_off + q and the quantisation step of a
    // slot is chosen by the workgroup itself (f32 [QT] in LDS at step_off); 0 = the u16 kernels (qstep[] in global memory)
    uint32_t step_off;
    // byte-table kernel with 64-key lists (16 < k <= 64): the four list positions (one per byte, ascending) whose keys a slice
    // publishes to its sibling slices (ScanArgs::q8_pos)
    uint32_t pos;
};

// byte filter bound (0x80 | T) implied by a k-th key for a table quantised with `step` (scan_q8.hip has the derivation)
template <int M>
__device__ __forceinline__ unsigned char qbound8_from_key(unsigned long long key, float smax_b, float step, double qlo_b) {
    const uint32_t hi = (uint32_t)(key >> 32);
    if (hi == kKeyInfHi) return 0xff;
    const double thr = (double)ordered_to_f32(hi);
    const double slack = (double)smax_b * (2.0 * M * 5.9604644775390625e-08 * (1.0 + 1.0 / 1024.0));
    double qd = (thr + slack - qlo_b) / (double)step * (1.0 + 1.0 / 524288.0);
    qd = __builtin_floor(qd) + 1.0;  // T
    if (!(qd < 127.0)) qd = 127.0;   // (a NaN lands HERE: everything passes)
    else if (!(qd > 0.0)) qd = 0.0;
    return (unsigned char)(0x80u | (uint32_t)qd);
}

// The bound the concurrently scanned sibling slices of a query have proven, from what they published in gk2 (cells of
// kGk2Keys keys per (query, slice)).  8 lanes per query, lane & 7 = slice within the group of 8; every lane returns the
// group's bound (~0: none).  With j = ceil(k / G) <= kGk2Keys a slice publishes its j smallest keys: G j >= k keys of distinct
// rows, so the k-th smallest of them has k rows at or below it.  (The MAX of the slices' j-th keys -- the fallback for
// larger j -- is the LARGEST of these keys: with 8 slices and k = 10 near global rank 36, the 10th smallest near rank 13;
// the candidates that pass the imported bound are in proportion.)
template <int CELL>  // keys per cell (gk2_cell_keys(M))
__device__ __forceinline__ unsigned long long sibling_bound(const unsigned long long *gk2, int64_t b, int n_slices, int g0,
                                                           int jm1, int km1, int lane) {
    const int sl = g0 + (lane & 7);
    const unsigned long long *cell = gk2 + (b * n_slices + sl) * CELL;
    if (CELL == kGk2Keys && jm1 < kGk2Keys) {
        unsigned long long kk[kGk2Keys];
#pragma unroll
        for (int i = 0; i < kGk2Keys; ++i) {
            kk[i] = ~0ull;
            if (sl < n_slices && i <= jm1) kk[i] = __hip_atomic_load(cell + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        // (a reader may see two versions of a slice's list mixed: the same row twice -- keys of distinct rows are distinct)
#pragma unroll
        for (int i = 1; i < kGk2Keys; ++i)
#pragma unroll
            for (int j = 0; j < i; ++j)
                if (kk[i] == kk[j]) kk[i] = ~0ull;
        int rk[kGk2Keys];
#pragma unroll
        for (int i = 0; i < kGk2Keys; ++i) rk[i] = 0;
#pragma unroll 1
        for (int o = 0; o < 8; ++o) {
#pragma unroll
            for (int t = 0; t < kGk2Keys; ++t) {
                const unsigned long long other = __shfl(kk[t], (lane & ~7) + o);
#pragma unroll
                for (int i = 0; i < kGk2Keys; ++i) rk[i] += other < kk[i];
            }
        }
        unsigned long long found = ~0ull;
#pragma unroll
        for (int i = 0; i < kGk2Keys; ++i)
            if (rk[i] == km1 && kk[i] != ~0ull) found = kk[i];
#pragma unroll
        for (int o = 1; o < 8; o <<= 1) {
            const unsigned long long p = __shfl_xor(found, o);
            found = p < found ? p : found;
        }
        return found;
    }
    unsigned long long v = 0ull;  // slots beyond n_slices never set the max
    if (sl < n_slices) v = __hip_atomic_load(cell, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
    for (int o = 1; o < 8; o <<= 1) {
        const unsigned long long p = __shfl_xor(v, o);
        v = p > v ? p : v;
    }
    return v;
}

// exact ascending-m fp32 sum of table row `rid` for query slot q (the reference's order, space_pq.h:32-35): re-reads
// the row's code bytes and gathers its M entries from the fp32 TILED table in global memory
// CODE16: uint16 codes (Ks > 256), PLAIN layout only
template <int M, bool SKEWED, bool CODE16 = false>
__device__ __forceinline__ float exact_row_sum(const FlushCtx &c, int q, uint32_t rid) {
    static_assert(!(CODE16 && SKEWED), "uint16 code tables are PLAIN");
    constexpr int CW = CODE16 ? M / 2 : M / 4;
    uint32_t cp[CW];
    const uint32_t *p = (const uint32_t *)(c.codes + (int64_t)rid * M * (CODE16 ? 2 : 1));
#pragma unroll
    for (int i = 0; i < CW; ++i) cp[i] = p[i];
    if constexpr (SKEWED && M == 64) {
        skew64_decode(cp, (int)(rid % 32));  // two skewed halves, wrap-coded
    } else if constexpr (SKEWED) {
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
        const uint32_t code = CODE16 ? (cp[m / 2] >> (16 * (m % 2))) & 0xffffu : (cp[m / 4] >> (8 * (m % 4))) & 0xffu;
        vals[m] = lq[((int64_t)code * M + m) * 4];
    }
    float ex = 0.f;
#pragma unroll
    for (int m = 0; m < M; ++m) ex += vals[m];
    return ex;
}

// Offer the candidates (khi, rid) of the lanes in pm to the shared list of query slot q0: pre-check against the
// current bound without the lock, then insert under the lock and publish the new k-th key / filter bound.
// LOCKED = false: the calling wave is the only one that touches the lists (the byte-table kernel's consumer wave)
template <int M, bool LOCKED = true>
__device__ __forceinline__ void offer_to_list(const FlushCtx &c, int q0, unsigned long long pm, uint32_t khi, uint32_t rid,
                                              int lane) {
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
    if (!px || (c.skip & 2)) return;
    if (c.dbg && lane == 0) atomicAdd(c.dbg + 2, 1ull);
    // ---- critical section ---------------------------------------------------------------------
    unsigned int *lock = (unsigned int *)(g_smem + c.lock_off + q0 * 4);
    if constexpr (LOCKED)
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
            // (min: in tile mode the integer seed bound can be tighter than the first k-th key)
            if (c.step_off) {
                volatile unsigned char *sp = (volatile unsigned char *)(g_smem + c.shq_off + q0);
                const float step = *(volatile float *)(g_smem + c.step_off + q0 * 4);
                const unsigned char nb = qbound8_from_key<M>(okey, c.smax[b], step, c.qlo[b]);
                if (nb < *sp) *sp = nb;
            } else {
                volatile unsigned short *sp = (volatile unsigned short *)(g_smem + c.shq_off + q0 * 2);
                const unsigned short nb = qbound_from_key<M>(okey, c.smax[b], c.qstep[b], c.qlo[b]);
                if (nb < *sp) *sp = nb;
            }
        }
        if (c.gk2) {
            // the slice's j smallest keys changed (or, with one key per cell, its j-th): what the sibling slices compute their
            // bound from (sibling_bound)
            const uint32_t jhi = __builtin_amdgcn_readlane(L.hi, c.jm1);
            const uint32_t jlo = __builtin_amdgcn_readlane(L.lo, c.jm1);
            const unsigned long long jkey = ((unsigned long long)jhi << 32) | jlo;
            volatile unsigned long long *gjl = (volatile unsigned long long *)(g_smem + c.gjl_off + q0 * 8);
            if (jhi != kKeyInfHi && jkey < *gjl) {  // (wave-uniform)
                unsigned long long *cell = c.gk2 + ((int64_t)b * c.n_slices + c.slice) * gk2_cell_keys(M);
                if (gk2_cell_keys(M) == kGk2Keys && c.jm1 < kGk2Keys) {  // the j smallest keys (sibling_bound; M = 64: see its import)
                    if (lane <= c.jm1)
                        __hip_atomic_store(cell + lane, ((unsigned long long)L.hi << 32) | L.lo, __ATOMIC_RELAXED,
                                           __HIP_MEMORY_SCOPE_AGENT);
                } else if (lane == 0) {
                    __hip_atomic_store(cell, jkey, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
                asm volatile("" ::: "memory");
                if (lane == 0) *gjl = jkey;
            }
        }
    }
    // LDS executes one wave's instructions in order, so the list stores are visible before the release
    if constexpr (LOCKED)
        if (lane == 0) __hip_atomic_store(lock, 0u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
}

// Flush one wave's candidate queue: up to 64 (query, row) pairs whose integer sum passed the filter.
// Lane i takes entry i: re-reads the row's code bytes, gathers its exact ascending-m fp32 sum from the
// fp32 table in global memory, then the candidates are offered query by query to the shared lists.
// Batching matters: one candidate at a time paid the gather latency, the call and the lock ~3.6 us each
// (46 times per wave at 1.25M rows); a flush pays them once for everything queued since the last one.
template <int M, bool SKEWED, bool CODE16 = false>
__device__ __forceinline__ void qfilter_flush_inline(const FlushCtx &c, uint32_t queue_off, int qcnt) {
    const int lane = threadIdx.x & 63;
    const bool act = lane < qcnt;
    const unsigned long long e = act ? ((const unsigned long long *)(g_smem + queue_off))[lane] : 0ull;
    const uint32_t rid = (uint32_t)e;
    const int q = (int)(e >> 32);
    float ex = 0.f;
    if (act && !(c.skip & 1)) ex = exact_row_sum<M, SKEWED, CODE16>(c, q, rid);
    const uint32_t khi = f32_to_key(ex);  // (NaN sums sort behind +inf: numpy's order)
    unsigned long long rem = __ballot(act);
    while (rem) {
        const int q0 = __builtin_amdgcn_readlane(q, __builtin_ctzll(rem));
        const unsigned long long pm = __ballot(act && q == q0);
        rem &= ~pm;
        offer_to_list<M>(c, q0, pm, khi, rid, lane);
    }
}

// out of line, one copy per kernel: the u16 kernels call it from three places of their step loop
template <int M, bool SKEWED, bool CODE16 = false>
__device__ __attribute__((noinline)) void qfilter_flush(const FlushCtx c, uint32_t queue_off, int qcnt) {
    qfilter_flush_inline<M, SKEWED, CODE16>(c, queue_off, qcnt);
}

// Final merge of a tile by the last workgroup to arrive.  The NW waves are dealt out over the tile's REAL
// queries: wave w folds the slices (w % wpq), (w % wpq) + wpq, ... of query w / wpq (a slice is merged only if
// it holds something better than the running k-th), the wpq lists of a query meet in LDS and its first wave
// writes the result.  Full tiles of big batches get one wave per query (few slices each); a single query with
// 256 slices gets all 16 waves (one wave folding 256 slices in sequence cost ~100 us of a 0.3 ms search).
template <int NW>
__device__ __forceinline__ void merge_tile_slices(const ScanArgs &a, int b0, int QT, int km1, int wave, int lane,
                                                  unsigned long long *scratch /* [NW][64], LDS */) {
    int nq = a.B - b0;
    if (nq > QT) nq = QT;
    for (int q0 = 0; q0 < nq; q0 += NW) {
        const int nqc = nq - q0 < NW ? nq - q0 : NW;
        const int wpq = NW / nqc;  // waves per query
        const int my_q = wave / wpq, my_part = wave - my_q * wpq;
        const bool active = my_q < nqc;
        const int b = b0 + q0 + my_q;
        WaveList L;
        L.reset();
        if (active) {
            uint32_t thi = kKeyInfHi, tlo = kIdNone;
            // four slice lists in flight per round trip (the loads are device-scope, ~1 us each when chained)
            for (int sl0 = my_part; sl0 < a.n_slices; sl0 += 4 * wpq) {
                unsigned long long key[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int sl = sl0 + u * wpq;
                    key[u] = ~0ull;
                    if (lane <= km1 && sl < a.n_slices)
                        key[u] = __hip_atomic_load(a.partial + ((int64_t)b * a.n_slices + sl) * a.k + lane,
                                                   __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const uint32_t chi = (uint32_t)(key[u] >> 32), clo = (uint32_t)key[u];
                    if (__ballot(key_less(chi, clo, thi, tlo))) {
                        wavelist_merge_sorted(L, chi, clo, lane);  // slice lists are ascending over the lanes
                        thi = __builtin_amdgcn_readlane(L.hi, km1);
                        tlo = __builtin_amdgcn_readlane(L.lo, km1);
                    }
                }
            }
        }
        if (wpq > 1) {
            scratch[wave * 64 + lane] = ((unsigned long long)L.hi << 32) | L.lo;
            __syncthreads();
            if (active && my_part == 0)
                for (int w = 1; w < wpq; ++w) {
                    const unsigned long long o = scratch[(wave + w) * 64 + lane];
                    wavelist_merge_sorted(L, (uint32_t)(o >> 32), (uint32_t)o, lane);
                }
        }
        if (active && my_part == 0 && lane <= km1) {
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
        if (wpq > 1) __syncthreads();  // scratch is reused by the next chunk
    }
}

}  // namespace annlite
