// codec.hip -- PQ codec kernels: encode (nearest codeword), decode (gather), l2_normalize,
// the k-means building blocks of PQCodec.fit, and the exact re-rank distance.
//
// Reference (jina-ai/annlite v0.5.11):
//   encode        annlite/core/codec/pq.py:158-177  (scipy.cluster.vq.vq per sub-space, first min wins)
//   decode        annlite/core/codec/pq.py:179-198
//   l2_normalize  annlite/math.py:6-18
//   fit           annlite/core/codec/pq.py:89-115   (sklearn KMeans per sub-space: Lloyd iterations)
//   exact dist    annlite/math.py:21-61 (cdist) as used by FlatIndex.search, flat_index.py:15-39
#include "common.h"

namespace annlite {

// ---- encode -------------------------------------------------------------------------------------
// One workgroup = one sub-space m x 256 rows; the sub-codebook C[m] (Ks*dsub floats) sits in LDS and
// is read with wave-uniform addresses (LDS broadcast: conflict-free), the row's sub-vector stays in
// registers.  Distance = the same sequential fmaf chain as the L2 LUT, so encode and LUT agree on
// which codeword is nearest; strict '<' keeps the first minimum like scipy's vq.
template <int DSUB, typename CODE_T, bool ACCUM>
__global__ __launch_bounds__(256) void encode_kernel(const float *__restrict__ x, int64_t N, int D,
                                                    const float *__restrict__ cb, int M, int Ks,
                                                    CODE_T *__restrict__ codes, float *__restrict__ sums,
                                                    int32_t *__restrict__ counts, double *__restrict__ inertia) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float *cw = (float *)smem;  // [Ks][DSUB]
    const int m = blockIdx.y;
    for (int i = threadIdx.x; i < Ks * DSUB; i += blockDim.x) cw[i] = cb[(int64_t)m * Ks * DSUB + i];
    __syncthreads();
    for (int64_t n = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; n < N; n += (int64_t)gridDim.x * blockDim.x) {
        float xv[DSUB];
        const float *xr = x + n * D + m * DSUB;
#pragma unroll
        for (int j = 0; j < DSUB; ++j) xv[j] = xr[j];
        float best = __builtin_inff();
        int arg = 0;
        for (int k = 0; k < Ks; ++k) {
            const float *c = cw + k * DSUB;
            float acc = 0.f;
#pragma unroll
            for (int j = 0; j < DSUB; ++j) {
                const float d = c[j] - xv[j];
                acc = __builtin_fmaf(d, d, acc);
            }
            if (acc < best) {
                best = acc;
                arg = k;
            }
        }
        if constexpr (ACCUM) {
            // Lloyd accumulation straight into global sums (fp32 atomics; order-dependent rounding is
            // inherent to parallel k-means -- the reference's sklearn fit is not reproducible either)
            float *s = sums + ((int64_t)m * Ks + arg) * DSUB;
#pragma unroll
            for (int j = 0; j < DSUB; ++j) atomicAdd(s + j, xv[j]);
            atomicAdd(counts + (int64_t)m * Ks + arg, 1);
            if (inertia) atomicAdd(inertia + m, (double)best);
        } else {
            codes[n * M + m] = (CODE_T)arg;
        }
    }
}

// generic dsub (codebook through L2, sub-vector re-read): correct for any shape, slow
template <typename CODE_T, bool ACCUM>
__global__ __launch_bounds__(256) void encode_generic_kernel(const float *__restrict__ x, int64_t N, int D,
                                                            const float *__restrict__ cb, int M, int Ks, int dsub,
                                                            CODE_T *__restrict__ codes, float *__restrict__ sums,
                                                            int32_t *__restrict__ counts,
                                                            double *__restrict__ inertia) {
    const int m = blockIdx.y;
    for (int64_t n = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; n < N; n += (int64_t)gridDim.x * blockDim.x) {
        const float *xr = x + n * D + (int64_t)m * dsub;
        float best = __builtin_inff();
        int arg = 0;
        for (int k = 0; k < Ks; ++k) {
            const float *c = cb + ((int64_t)m * Ks + k) * dsub;
            float acc = 0.f;
            for (int j = 0; j < dsub; ++j) {
                const float d = c[j] - xr[j];
                acc = __builtin_fmaf(d, d, acc);
            }
            if (acc < best) {
                best = acc;
                arg = k;
            }
        }
        if constexpr (ACCUM) {
            float *s = sums + ((int64_t)m * Ks + arg) * dsub;
            for (int j = 0; j < dsub; ++j) atomicAdd(s + j, xr[j]);
            atomicAdd(counts + (int64_t)m * Ks + arg, 1);
            if (inertia) atomicAdd(inertia + m, (double)best);
        } else {
            codes[n * M + m] = (CODE_T)arg;
        }
    }
}

template <typename CODE_T, bool ACCUM>
static int launch_encode(const float *x, int64_t N, int64_t D, const float *cb, int64_t M, int64_t Ks, CODE_T *codes,
                         float *sums, int32_t *counts, double *inertia, hipStream_t st) {
    const int dsub = (int)(D / M);
    int64_t bx = (N + 255) / 256;
    const int64_t cap = ((int64_t)device_cu_count() * 16 + M - 1) / M;
    if (bx > cap) bx = cap;
    if (bx < 1) bx = 1;
    dim3 grid((unsigned)bx, (unsigned)M);
    const size_t lds = (size_t)Ks * dsub * 4;
#define ANNLITE_ENC(DS)                                                                                            \
    case DS:                                                                                                       \
        if (lds <= 64 * 1024) {                                                                                    \
            hipLaunchKernelGGL((encode_kernel<DS, CODE_T, ACCUM>), grid, dim3(256), lds, st, x, N, (int)D, cb,     \
                               (int)M, (int)Ks, codes, sums, counts, inertia);                                     \
            return launch_status("encode_kernel");                                                                 \
        }                                                                                                          \
        break;
    switch (dsub) {
        ANNLITE_ENC(1)
        ANNLITE_ENC(2)
        ANNLITE_ENC(3)
        ANNLITE_ENC(4)
        ANNLITE_ENC(5)
        ANNLITE_ENC(6)
        ANNLITE_ENC(8)
        ANNLITE_ENC(12)
        ANNLITE_ENC(16)
        ANNLITE_ENC(24)
        ANNLITE_ENC(32)
        default: break;
    }
#undef ANNLITE_ENC
    hipLaunchKernelGGL((encode_generic_kernel<CODE_T, ACCUM>), grid, dim3(256), 0, st, x, N, (int)D, cb, (int)M,
                       (int)Ks, dsub, codes, sums, counts, inertia);
    return launch_status("encode_generic_kernel");
}

// ---- decode -------------------------------------------------------------------------------------
template <typename CODE_T>
__global__ __launch_bounds__(256) void decode_kernel(const CODE_T *__restrict__ codes, int64_t N, int M, int Ks,
                                                    const float *__restrict__ cb, int dsub, float *__restrict__ out) {
    const int64_t D = (int64_t)M * dsub;
    const int64_t total = N * D;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t n = i / D;
        const int r = (int)(i - n * D);
        const int m = r / dsub, j = r - m * dsub;
        out[i] = cb[((int64_t)m * Ks + (int64_t)codes[n * M + m]) * dsub + j];
    }
}

// ---- l2_normalize: one wave per row ---------------------------------------------------------------
__global__ __launch_bounds__(256) void l2_normalize_kernel(const float *x, int64_t N, int D, float *out) {  // may alias
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= N) return;
    const float *xr = x + row * D;
    float s = 0.f;
    for (int j = lane; j < D; j += 64) s = __builtin_fmaf(xr[j], xr[j], s);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    float norm = __builtin_sqrtf(s);
    // annlite/math.py:14-16: constant_mask = norms < 10*eps ; norms[mask] = 1
    if (norm < 10.f * 1.1920928955078125e-07f) norm = 1.f;
    for (int j = lane; j < D; j += 64) out[row * D + j] = xr[j] / norm;
}

// ---- k-means update -------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void kmeans_update_kernel(const float *__restrict__ sums,
                                                           const int32_t *__restrict__ counts, int64_t total, int dsub,
                                                           float *__restrict__ cb) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int32_t c = counts[i / dsub];
    if (c > 0) cb[i] = sums[i] / (float)c;
}

// ---- exact distances over candidate lists: one wave per (query, candidate) -------------------------
__global__ __launch_bounds__(256) void exact_gather_kernel(int metric, const float *__restrict__ q, int B, int D,
                                                          const float *__restrict__ x, int64_t N,
                                                          const int64_t *__restrict__ cand, int R,
                                                          float *__restrict__ out) {
    const int lane = threadIdx.x & 63;
    const int64_t i = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (i >= (int64_t)B * R) return;
    const int b = (int)(i / R);
    const int64_t row = cand[i];
    if (row < 0 || row >= N) {
        if (lane == 0) out[i] = __builtin_inff();
        return;
    }
    const float *qr = q + (int64_t)b * D;
    const float *xr = x + row * D;
    float s = 0.f;
    if (metric == ANNLITE_METRIC_EUCLIDEAN) {
        for (int j = lane; j < D; j += 64) {
            const float d = xr[j] - qr[j];
            s = __builtin_fmaf(d, d, s);
        }
    } else {
        for (int j = lane; j < D; j += 64) s = __builtin_fmaf(xr[j], qr[j], s);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    if (lane == 0) out[i] = (metric == ANNLITE_METRIC_EUCLIDEAN) ? s : 1.f - s;
}

// ---- exact re-rank FUSED with the top-k (round 6): one wave per query ------------------------------------------------------------
// The candidate lists of a graph walk / of the scan's candidate mode are short (ef = 128 rows): a wave per (query, candidate)
// (exact_gather_kernel) followed by torch's masking, annlite_topk_rows, a gather and a sqrt was eight launches of a few
// microseconds each around 67 MB of reads -- a quarter of a config-5 batch.  Here the query's wave walks its own list: eight
// candidates' rows in flight at a time, the SAME per-lane partial sums and butterfly as exact_gather_kernel (bit-equal distances),
// rows screened by the validity bitmap, the k best kept in the wave-resident list under (distance, position in the list) --
// annlite_topk_rows' order -- and written out as (distance | sqrt, candidate id).
__global__ __launch_bounds__(256) void rerank_topk_kernel(int metric, const float *__restrict__ q, int B, int D,
                                                         const float *__restrict__ x, int64_t N, const int64_t *__restrict__ cand, int R,
                                                         const uint32_t *__restrict__ valid, int k, int do_sqrt,
                                                         float *__restrict__ out_d, int64_t *__restrict__ out_i) {
    const int lane = threadIdx.x & 63;
    const int b = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (b >= B) return;
    constexpr int U = 8;
    const float *qr = q + (int64_t)b * D;
    const int64_t *cr = cand + (int64_t)b * R;
    const int km1 = k - 1;
    WaveList L;
    L.reset();
    uint32_t th = kKeyInfHi, tl = kIdNone;
    for (int c0 = 0; c0 < R; c0 += 64) {
        // this block's 64 candidates: lane l screens candidate c0 + l
        const int ci = c0 + lane;
        int64_t row = ci < R ? cr[ci] : -1;
        if (row >= N) row = -1;
        if (row >= 0 && valid && !((valid[row >> 5] >> (row & 31)) & 1u)) row = -1;
        float mine = __builtin_inff();  // lane l ends up with candidate c0 + l's distance
        const int n_here = R - c0 < 64 ? R - c0 : 64;
        for (int u0 = 0; u0 < n_here; u0 += U) {
            float s[U];
            int64_t rw[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                rw[u] = __shfl(row, u0 + u);  // (wave-uniform)
                s[u] = 0.f;
            }
            for (int j = lane; j < D; j += 64) {
                const float qj = qr[j];
                float xv[U];
#pragma unroll
                for (int u = 0; u < U; ++u) xv[u] = rw[u] >= 0 ? x[rw[u] * D + j] : 0.f;
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    if (metric == ANNLITE_METRIC_EUCLIDEAN) {
                        const float d = xv[u] - qj;
                        s[u] = __builtin_fmaf(d, d, s[u]);
                    } else {
                        s[u] = __builtin_fmaf(xv[u], qj, s[u]);
                    }
                }
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
#pragma unroll
                for (int o = 32; o > 0; o >>= 1) s[u] += __shfl_xor(s[u], o);
                const float dist = (metric == ANNLITE_METRIC_EUCLIDEAN) ? s[u] : 1.f - s[u];
                if (lane == u0 + u) mine = dist;
            }
        }
        if (row < 0) mine = __builtin_inff();
        const float tf = (th == kKeyInfHi) ? __builtin_inff() : ordered_to_f32(th);
        const unsigned long long pm = __ballot(ci < R && !(mine > tf));  // (NaN: behind +inf, numpy's order -- as annlite_topk_rows)
        if (pm) wavelist_offer(L, pm, f32_to_key(mine), (uint32_t)ci, km1, th, tl, lane);
    }
    if (lane <= km1) {
        const bool none = (L.hi == kKeyInfHi && L.lo == kIdNone);
        float d = none ? __builtin_inff() : ordered_to_f32(L.hi);
        int64_t id = (none || d == __builtin_inff()) ? (int64_t)-1 : cr[L.lo];
        if (id >= N) id = -1;
        if (do_sqrt) d = __builtin_sqrtf(d);
        out_d[(int64_t)b * k + lane] = d;
        out_i[(int64_t)b * k + lane] = id;
    }
}

}  // namespace annlite

using namespace annlite;

extern "C" int annlite_rerank_topk(int metric, const float *queries_dev, int64_t B, int64_t D, const float *vectors_dev, int64_t N,
                                   const int64_t *cand_dev, int64_t R, const uint32_t *valid_bits_dev, int64_t k, int flags,
                                   float *out_dist_dev, int64_t *out_id_dev, void *stream) {
    ANNLITE_REQUIRE(metric >= 1 && metric <= 3, "bad metric %d", metric);
    ANNLITE_REQUIRE(B >= 0 && D >= 1 && N >= 0 && R >= 0 && k >= 1 && k <= 64, "bad shape (1 <= k <= 64)");
    if (B == 0) return ANNLITE_OK;
    ANNLITE_REQUIRE(queries_dev && out_dist_dev && out_id_dev && (R == 0 || cand_dev) && (N == 0 || vectors_dev), "null device pointer");
    hipLaunchKernelGGL(rerank_topk_kernel, dim3((unsigned)((B + 3) / 4)), dim3(256), 0, (hipStream_t)stream, metric, queries_dev, (int)B,
                       (int)D, vectors_dev, N, cand_dev, (int)R, valid_bits_dev, (int)k, (flags & ANNLITE_FLAG_SQRT) ? 1 : 0, out_dist_dev,
                       out_id_dev);
    return launch_status("rerank_topk_kernel");
}

extern "C" int annlite_pq_encode(const float *x_dev, int64_t N, int64_t D, const float *codebooks_dev, int64_t M,
                                 int64_t Ks, void *out_codes_dev, int code_bytes, void *stream) {
    ANNLITE_REQUIRE(N >= 0 && M >= 1 && Ks >= 1 && D >= M && D % M == 0, "input dimension must be Ds * M (D=%lld, M=%lld)",
                    (long long)D, (long long)M);
    ANNLITE_REQUIRE((code_bytes == 1 && Ks <= 256) || (code_bytes == 2 && Ks <= 65536) || code_bytes == 4,
                    "code_bytes=%d cannot hold Ks=%lld", code_bytes, (long long)Ks);
    if (N == 0) return ANNLITE_OK;
    ANNLITE_REQUIRE(x_dev && codebooks_dev && out_codes_dev, "null device pointer");
    hipStream_t st = (hipStream_t)stream;
    if (code_bytes == 1)
        return launch_encode<uint8_t, false>(x_dev, N, D, codebooks_dev, M, Ks, (uint8_t *)out_codes_dev, nullptr,
                                             nullptr, nullptr, st);
    if (code_bytes == 2)
        return launch_encode<uint16_t, false>(x_dev, N, D, codebooks_dev, M, Ks, (uint16_t *)out_codes_dev, nullptr,
                                              nullptr, nullptr, st);
    return launch_encode<uint32_t, false>(x_dev, N, D, codebooks_dev, M, Ks, (uint32_t *)out_codes_dev, nullptr,
                                          nullptr, nullptr, st);
}

extern "C" int annlite_kmeans_assign_accumulate(const float *x_dev, int64_t N, int64_t D, const float *codebooks_dev,
                                                int64_t M, int64_t Ks, float *sums_dev, int32_t *counts_dev,
                                                double *inertia_dev, void *stream) {
    ANNLITE_REQUIRE(N >= 0 && M >= 1 && Ks >= 1 && D >= M && D % M == 0, "input dimension must be Ds * M");
    if (N == 0) return ANNLITE_OK;
    ANNLITE_REQUIRE(x_dev && codebooks_dev && sums_dev && counts_dev, "null device pointer");
    return launch_encode<uint8_t, true>(x_dev, N, D, codebooks_dev, M, Ks, nullptr, sums_dev, counts_dev, inertia_dev,
                                        (hipStream_t)stream);
}

extern "C" int annlite_kmeans_update(const float *sums_dev, const int32_t *counts_dev, int64_t M, int64_t Ks,
                                     int64_t dsub, float *codebooks_dev, void *stream) {
    ANNLITE_REQUIRE(M >= 1 && Ks >= 1 && dsub >= 1, "bad shape");
    ANNLITE_REQUIRE(sums_dev && counts_dev && codebooks_dev, "null device pointer");
    const int64_t total = M * Ks * dsub;
    hipLaunchKernelGGL(kmeans_update_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       sums_dev, counts_dev, total, (int)dsub, codebooks_dev);
    return launch_status("kmeans_update_kernel");
}

extern "C" int annlite_pq_decode(const void *codes_dev, int code_bytes, int64_t N, int64_t M, int64_t Ks,
                                 const float *codebooks_dev, int64_t dsub, float *out_dev, void *stream) {
    ANNLITE_REQUIRE(N >= 0 && M >= 1 && Ks >= 1 && dsub >= 1, "bad shape");
    ANNLITE_REQUIRE(code_bytes == 1 || code_bytes == 2 || code_bytes == 4, "code_bytes must be 1, 2 or 4");
    if (N == 0) return ANNLITE_OK;
    ANNLITE_REQUIRE(codes_dev && codebooks_dev && out_dev, "null device pointer");
    const int64_t total = N * M * dsub;
    int64_t blocks = (total + 255) / 256;
    const int64_t cap = (int64_t)device_cu_count() * 16;
    if (blocks > cap) blocks = cap;
    hipStream_t st = (hipStream_t)stream;
    if (code_bytes == 1)
        hipLaunchKernelGGL(decode_kernel<uint8_t>, dim3((unsigned)blocks), dim3(256), 0, st,
                           (const uint8_t *)codes_dev, N, (int)M, (int)Ks, codebooks_dev, (int)dsub, out_dev);
    else if (code_bytes == 2)
        hipLaunchKernelGGL(decode_kernel<uint16_t>, dim3((unsigned)blocks), dim3(256), 0, st,
                           (const uint16_t *)codes_dev, N, (int)M, (int)Ks, codebooks_dev, (int)dsub, out_dev);
    else
        hipLaunchKernelGGL(decode_kernel<uint32_t>, dim3((unsigned)blocks), dim3(256), 0, st,
                           (const uint32_t *)codes_dev, N, (int)M, (int)Ks, codebooks_dev, (int)dsub, out_dev);
    return launch_status("decode_kernel");
}

extern "C" int annlite_l2_normalize(const float *x_dev, int64_t N, int64_t D, float *out_dev, void *stream) {
    ANNLITE_REQUIRE(N >= 0 && D >= 1, "bad shape");
    if (N == 0) return ANNLITE_OK;
    ANNLITE_REQUIRE(x_dev && out_dev, "null device pointer");
    hipLaunchKernelGGL(l2_normalize_kernel, dim3((unsigned)((N + 3) / 4)), dim3(256), 0, (hipStream_t)stream, x_dev, N,
                       (int)D, out_dev);
    return launch_status("l2_normalize_kernel");
}

extern "C" int annlite_exact_gather_dist(int metric, const float *queries_dev, int64_t B, int64_t D,
                                         const float *vectors_dev, int64_t N, const int64_t *cand_dev, int64_t R,
                                         float *out_dev, void *stream) {
    ANNLITE_REQUIRE(metric >= 1 && metric <= 3, "bad metric %d", metric);
    ANNLITE_REQUIRE(B >= 0 && D >= 1 && N >= 0 && R >= 0, "bad shape");
    if (B * R == 0) return ANNLITE_OK;
    ANNLITE_REQUIRE(queries_dev && cand_dev && out_dev && (N == 0 || vectors_dev), "null device pointer");
    const int64_t total = B * R;
    hipLaunchKernelGGL(exact_gather_kernel, dim3((unsigned)((total + 3) / 4)), dim3(256), 0, (hipStream_t)stream,
                       metric, queries_dev, (int)B, (int)D, vectors_dev, N, cand_dev, (int)R, out_dev);
    return launch_status("exact_gather_kernel");
}
