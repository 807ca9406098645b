// common.h -- shared device/host helpers for libannlite_hip (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "../../include/annlite_hip.h"

namespace annlite {

// ---- error plumbing (thread-local message, int status across the C ABI) -------------------------
void set_error(const char *fmt, ...);
int hip_fail(hipError_t e, const char *what);

#define ANNLITE_HIP_TRY(expr)                                     \
    do {                                                          \
        hipError_t _e = (expr);                                   \
        if (_e != hipSuccess) return ::annlite::hip_fail(_e, #expr); \
    } while (0)

#define ANNLITE_REQUIRE(cond, ...)              \
    do {                                        \
        if (!(cond)) {                          \
            ::annlite::set_error(__VA_ARGS__);  \
            return ANNLITE_ERR_INVALID;         \
        }                                       \
    } while (0)

inline int launch_status(const char *kernel) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return hip_fail(e, kernel);
    return ANNLITE_OK;
}

int device_cu_count();

// ---- measurement / test switches (ANNLITE_* environment variables) --------------------------------
// Read ONCE, when the library is loaded, into this block; the search path reads the block, never the environment (a getenv per
// launch races with a concurrent setenv, and a variable set later would silently change which kernel a production index runs).
// A process that changes the variables afterwards -- the tests' A/B switches -- says so: annlite_knobs_reload() parses the
// environment again and publishes a NEW block (readers hold the old or the new one, never a half-written one).
// "unset" is -1 for the integer knobs unless said otherwise.
struct Knobs {
    int scan_variant;           // ANNLITE_SCAN_VARIANT
    bool no_fast_code16;        // ANNLITE_NO_FAST_CODE16
    int64_t scan_slices;        // ANNLITE_SCAN_SLICES (0: unset)
    int debug_skip;             // ANNLITE_DEBUG_SKIP (0: unset)
    int debug_counters;         // ANNLITE_DEBUG_COUNTERS (0: unset; 2 = phase stamps only)
    int q8_map;                 // ANNLITE_Q8_MAP
    int q8_ilv;                 // ANNLITE_Q8_ILV
    int q8_rebuild;             // ANNLITE_Q8_REBUILD
    int q8_target;              // ANNLITE_Q8_TARGET
    bool q8_tune_set;           // ANNLITE_Q8_TUNE = "epoch0,mul,ring_limit,import_mask" (set: the string was given, valid or not)
    bool q8_tune_ok;
    int q8_tune[4];
    int64_t guard_base;         // ANNLITE_GUARD_BASE
    bool q8_pos_ok;             // ANNLITE_Q8_POS = "f0,f1,f2,f3"
    double q8_pos[4];
    int flush_mask;             // ANNLITE_FLUSH_MASK
    bool seed_rows_set;         // ANNLITE_SEED_ROWS (0 is a value: the scan starts without a bound)
    int64_t seed_rows;
    bool seed_contiguous;       // ANNLITE_SEED_CONTIGUOUS
    int seed_chunk_log;         // ANNLITE_SEED_CHUNK_LOG (clamped to [0, 6]; default 3)
    bool no_fused_seed;         // ANNLITE_NO_FUSED_SEED
    bool no_prebuilt_tables;    // ANNLITE_NO_PREBUILT_TABLES
    bool no_early_merge;        // ANNLITE_NO_EARLY_MERGE
    int64_t early_merge_patience;  // ANNLITE_EARLY_MERGE_PATIENCE
    bool no_inkernel_merge;     // ANNLITE_NO_INKERNEL_MERGE
    bool no_fused_lut;          // ANNLITE_NO_FUSED_LUT
    bool mfma_seed;             // ANNLITE_MFMA_SEED (opt-in: the seed bound from MFMA-nominated rows, seed_mfma.hip -- measured slower
                                // than the seed rows' exact scan as a whole)
    bool no_cand_seed;          // ANNLITE_NO_CAND_SEED (A/B: the candidate generator with per-slice seeds and its own table builds, as before round 6)
    int graph_hash_bits;        // ANNLITE_GRAPH_HASH_BITS
    bool graph_seq_insert;      // ANNLITE_GRAPH_SEQ_INSERT
    bool ivf_static_tiles;      // ANNLITE_IVF_STATIC_TILES (A/B: the cell tiles dealt round-robin instead of drawn from a counter)
    int ivf_first;              // ANNLITE_IVF_FIRST (annlite_ivf_search_topk: probes per query whose tiles come first; -1: the default)
};
const Knobs &knobs();

// ---- vector types -------------------------------------------------------------------------------
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

constexpr int kWave = 64;  // CDNA wavefront

// ---- order-preserving float <-> uint32 map ------------------------------------------------------
// key(a) < key(b)  <=>  a < b  for all non-NaN floats (-0.0 sorts just below +0.0; an ADC sum that
// starts from +0.0f can never be -0.0, see DESIGN.md "Numerics").
__device__ __forceinline__ uint32_t f32_to_ordered(float f) {
    uint32_t u = __float_as_uint(f);
    uint32_t mask = (uint32_t)((int32_t)u >> 31) | 0x80000000u;
    return u ^ mask;
}
__device__ __forceinline__ float ordered_to_f32(uint32_t k) {
    uint32_t mask = (k & 0x80000000u) ? 0x80000000u : 0xffffffffu;
    return __uint_as_float(k ^ mask);
}
// ... and for NaN: the reference selects with numpy's order (math.py:107-116: argpartition / argsort), which puts NaN behind
// every number, +inf included, whatever its sign bit.  Keys of EXACT sums go through this form (one canonical NaN key above
// key(+inf), below "none"); the raw map above would sort a NaN with the sign bit set in front of -inf.
__device__ __forceinline__ uint32_t f32_to_key(float f) { return f != f ? 0xffc00000u : f32_to_ordered(f); }
constexpr uint32_t kKeyInfHi = 0xffffffffu;  // sorts after every real distance (> key(+inf), > key(NaN))
constexpr uint32_t kIdNone = 0xffffffffu;

// ---- wave-resident sorted candidate list --------------------------------------------------------
// One list per (wave, query): lane i holds the i-th smallest (dist-key, row-id) pair seen so far,
// ascending in (key, id) -- the build's fixed tie-break (distance asc, row id asc).  Lists live in
// two VGPRs; nothing is staged in LDS, so the LUT can own the whole 160 KB.
struct WaveList {
    uint32_t hi;  // ordered distance key of this lane's entry
    uint32_t lo;  // row id of this lane's entry
    __device__ __forceinline__ void reset() {
        hi = kKeyInfHi;
        lo = kIdNone;
    }
};

// insert the wave-uniform candidate (chi, clo); caller has checked it beats the current k-th entry
// or does not care (inserting a worse candidate is harmless: it lands beyond the kept prefix).
__device__ __forceinline__ void wavelist_insert(WaveList &L, uint32_t chi, uint32_t clo, int lane) {
    const bool less = (L.hi < chi) || (L.hi == chi && L.lo < clo);  // a prefix of lanes, list sorted
    const int pos = __popcll(__ballot(less));
    const uint32_t shi = __shfl_up(L.hi, 1);
    const uint32_t slo = __shfl_up(L.lo, 1);
    if (lane == pos) {
        L.hi = chi;
        L.lo = clo;
    } else if (lane > pos) {
        L.hi = shi;
        L.lo = slo;
    }
}

__device__ __forceinline__ bool key_less(uint32_t ahi, uint32_t alo, uint32_t bhi, uint32_t blo) {
    return (ahi < bhi) || (ahi == bhi && alo < blo);
}

// ---- bulk path: bitonic networks over the 64 lanes (used when many rows arrive at once) ----------
// compare-exchange with the lane `stride` away; keep_min lanes keep the smaller key
__device__ __forceinline__ void wave_cmpx(uint32_t &hi, uint32_t &lo, int stride, bool keep_min) {
    const uint32_t phi = __shfl_xor(hi, stride), plo = __shfl_xor(lo, stride);
    const bool p_less = key_less(phi, plo, hi, lo);
    if (keep_min == p_less) {
        hi = phi;
        lo = plo;
    }
}
// ascending sort of one key per lane (21 compare-exchange stages)
__device__ __forceinline__ void wave_sort64(uint32_t &hi, uint32_t &lo, int lane) {
    // deliberately NOT unrolled: this is cold code that is inlined once per query of a workgroup tile;
    // 21 unrolled stages x 16 queries blew the kernel up to >10k instructions (I-cache thrash)
#pragma unroll 1
    for (int size = 2; size <= 64; size <<= 1) {
#pragma unroll 1
        for (int stride = size >> 1; stride > 0; stride >>= 1) {
            const bool asc = (lane & size) == 0;
            const bool lower = (lane & stride) == 0;
            wave_cmpx(hi, lo, stride, lower == asc);
        }
    }
}
// L <- the 64 smallest of (L  U  C) where C is ascending over the lanes: min(L[i], C[63-i]) is a
// bitonic sequence holding exactly those 64, then a 6-stage bitonic merge sorts it
__device__ __forceinline__ void wavelist_merge_sorted(WaveList &L, uint32_t chi, uint32_t clo, int lane) {
    const uint32_t rhi = __shfl(chi, 63 - lane), rlo = __shfl(clo, 63 - lane);
    if (key_less(rhi, rlo, L.hi, L.lo)) {
        L.hi = rhi;
        L.lo = rlo;
    }
#pragma unroll 1
    for (int stride = 32; stride > 0; stride >>= 1) wave_cmpx(L.hi, L.lo, stride, (lane & stride) == 0);
}
// insert the candidates of the lanes in `pm` (unconditionally; worse-than-64th entries fall off)
__device__ __forceinline__ void wavelist_insert_many(WaveList &L, unsigned long long pm, uint32_t hi, uint32_t lo,
                                                     int lane) {
    if (__popcll(pm) > 8) {
        const bool mine = (pm >> lane) & 1ull;
        uint32_t chi = mine ? hi : kKeyInfHi, clo = mine ? lo : kIdNone;
        wave_sort64(chi, clo, lane);
        wavelist_merge_sorted(L, chi, clo, lane);
    } else {
        while (pm) {
            const int src = __builtin_ctzll(pm);
            pm &= pm - 1;
            wavelist_insert(L, __builtin_amdgcn_readlane(hi, src), __builtin_amdgcn_readlane(lo, src), lane);
        }
    }
}

// Offer every lane's candidate (hi, lo) where `pm` has a bit set; keeps the list exact.
// thr_hi/thr_lo = key of the current k-th entry (wave-uniform), updated on every insertion.
__device__ __forceinline__ void wavelist_offer(WaveList &L, unsigned long long pm, uint32_t hi, uint32_t lo,
                                               int km1, uint32_t &thr_hi, uint32_t &thr_lo, int lane) {
    while (pm) {
        const int src = __builtin_ctzll(pm);
        pm &= pm - 1;
        const uint32_t chi = __builtin_amdgcn_readlane(hi, src);
        const uint32_t clo = __builtin_amdgcn_readlane(lo, src);
        if (key_less(chi, clo, thr_hi, thr_lo)) {
            wavelist_insert(L, chi, clo, lane);
            thr_hi = __builtin_amdgcn_readlane(L.hi, km1);
            thr_lo = __builtin_amdgcn_readlane(L.lo, km1);
        }
    }
}

}  // namespace annlite
