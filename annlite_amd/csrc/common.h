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
constexpr uint32_t kKeyInfHi = 0xffffffffu;  // sorts after every real distance (> key(+inf))
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
