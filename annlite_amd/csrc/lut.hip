// lut.hip -- per-query sub-quantiser look-up tables (the "ADC tables").
//
// Reference (jina-ai/annlite v0.5.11):
//   L2 :  T[b,m,k] = sum_j (C[m,k,j] - q[b,m*dsub+j])^2      bindings/pq_bindings.pyx:149-210 (85-145: B=1)
//   IP :  S[b,m,k] = sum_j  C[m,k,j] * q[b,m*dsub+j]         bindings/pq_bindings.pyx:214-274
//   IPDIST = float32(1/Ks) - S                                annlite/core/codec/pq.py:316-322
// Both inner loops are compiled by the reference's flags into ONE fused multiply-add per j, in
// ascending j; the kernels below reproduce that chain bit-for-bit:
//   * L2 is not a contraction ((c-q)^2), so it is a VALU fmaf chain;
//   * IP is a real dense contraction [B,dsub]x[dsub,Ks] per sub-space and runs on the matrix cores
//     with v_mfma_f32_16x16x4_f32, whose result is bitwise a k-ordered fmaf chain
//     (cdna_hip_programming.md section 3 "FP32-input MFMA").
// Output layouts: BMK = the reference's [B][M][Ks]; TILED = [ceil(B/QI)][Ks][M][QI], the layout the
// scan kernel copies linearly into LDS (scan.hip).
#include "common.h"

namespace annlite {

// ---- L2, VALU fmaf chain ------------------------------------------------------------------------
// TILED mapping: one thread per (group bq, code k, sub-space m) computes QI chains and stores them
// contiguously -> fully coalesced 16 B (QI=4) stores.
template <int QI>
__global__ __launch_bounds__(256) void lut_l2_tiled_kernel(const float *__restrict__ q, int B, int D,
                                                          const float *__restrict__ cb, int M, int Ks, int dsub,
                                                          float *__restrict__ out, int64_t total) {
    const int64_t id = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (id >= total) return;
    const int m = (int)(id % M);
    const int k = (int)((id / M) % Ks);
    const int bq = (int)(id / ((int64_t)M * Ks));
    const float *cw = cb + ((int64_t)m * Ks + k) * dsub;
    float acc[QI];
#pragma unroll
    for (int i = 0; i < QI; ++i) acc[i] = 0.f;
    if ((dsub & 3) == 0) {
        // 16-byte loads (the scalar loop below was bound by the number of load instructions: the codeword
        // addresses of neighbouring lanes are Ks*dsub floats apart); same j-ascending fmaf chain
        for (int j = 0; j < dsub; j += 4) {
            const f32x4 cj = *(const f32x4 *)(cw + j);
#pragma unroll
            for (int i = 0; i < QI; ++i) {
                const int b = bq * QI + i;
                const f32x4 qj = (b < B) ? *(const f32x4 *)(q + (int64_t)b * D + m * dsub + j) : cj;  // pad -> 0
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float c = cj[e] - qj[e];
                    acc[i] = __builtin_fmaf(c, c, acc[i]);
                }
            }
        }
    } else {
        for (int j = 0; j < dsub; ++j) {
            const float cj = cw[j];
#pragma unroll
            for (int i = 0; i < QI; ++i) {
                const int b = bq * QI + i;
                const float qj = (b < B) ? q[(int64_t)b * D + m * dsub + j] : cj;  // pad queries -> 0
                const float c = cj - qj;
                acc[i] = __builtin_fmaf(c, c, acc[i]);
            }
        }
    }
#pragma unroll
    for (int i = 0; i < QI; ++i) out[id * QI + i] = acc[i];
}

// BMK mapping: one thread per (b, m, k), k fastest -> coalesced stores in the reference layout.
__global__ __launch_bounds__(256) void lut_bmk_kernel(int kind, const float *__restrict__ q, int B, int D,
                                                     const float *__restrict__ cb, int M, int Ks, int dsub,
                                                     float inv_ks, float *__restrict__ out, int64_t total) {
    const int64_t id = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (id >= total) return;
    const int k = (int)(id % Ks);
    const int m = (int)((id / Ks) % M);
    const int b = (int)(id / ((int64_t)M * Ks));
    const float *cw = cb + ((int64_t)m * Ks + k) * dsub;
    const float *qs = q + (int64_t)b * D + m * dsub;
    float acc = 0.f;
    if ((dsub & 3) == 0) {
        // 16-byte loads, same j-ascending chains (the scalar loops are bound by the number of load instructions)
        for (int j = 0; j < dsub; j += 4) {
            const f32x4 cj = *(const f32x4 *)(cw + j);
            const f32x4 qj = *(const f32x4 *)(qs + j);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                if (kind == ANNLITE_LUT_L2) {
                    const float c = cj[e] - qj[e];
                    acc = __builtin_fmaf(c, c, acc);
                } else {
                    acc = __builtin_fmaf(cj[e], qj[e], acc);
                }
            }
        }
    } else if (kind == ANNLITE_LUT_L2) {
        for (int j = 0; j < dsub; ++j) {
            const float c = cw[j] - qs[j];
            acc = __builtin_fmaf(c, c, acc);
        }
    } else {
        for (int j = 0; j < dsub; ++j) acc = __builtin_fmaf(cw[j], qs[j], acc);
    }
    if (kind == ANNLITE_LUT_IPDIST) acc = inv_ks - acc;
    out[id] = acc;
}

// ---- IP on the matrix cores ---------------------------------------------------------------------
// One wave = one 16(query) x 16(code) tile of one sub-space: D[b][k] = sum_j q[b][j] * C[k][j].
// v_mfma_f32_16x16x4_f32 operands: A: lane l -> A[i=l&15][kk=l>>4]; B: lane l -> B[kk=l>>4][j=l&15];
// C/D: col = l&15, row = (l>>4)*4 + r.  Chaining ceil(dsub/4) MFMAs on one accumulator gives the
// j-ascending fmaf chain of the reference (zero padding of a ragged tail adds +0.0 exactly).
// A lane ends up with 4 CONSECUTIVE queries of one code -> one 16 B store in the TILED layout.
template <int QI>
__global__ __launch_bounds__(256) void lut_ip_mfma_kernel(const float *__restrict__ q, int B, int D,
                                                         const float *__restrict__ cb, int M, int Ks, int dsub,
                                                         int ipdist, float inv_ks, float *__restrict__ out,
                                                         int layout, int n_btiles, int n_ktiles) {
    const int lane = threadIdx.x & 63;
    const int64_t wid = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int64_t n_w = (int64_t)n_btiles * n_ktiles * M;
    if (wid >= n_w) return;
    const int kt = (int)(wid % n_ktiles);
    const int m = (int)((wid / n_ktiles) % M);
    const int bt = (int)(wid / ((int64_t)n_ktiles * M));
    const int kk = lane >> 4;
    const int bi = bt * 16 + (lane & 15);  // A row (query)
    const int ki = kt * 16 + (lane & 15);  // B col (code)
    const float *qa = q + (int64_t)(bi < B ? bi : 0) * D + m * dsub;
    const float *cw = cb + ((int64_t)m * Ks + (ki < Ks ? ki : 0)) * dsub;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    for (int j0 = 0; j0 < dsub; j0 += 4) {
        const int j = j0 + kk;
        const float av = (j < dsub && bi < B) ? qa[j] : 0.f;
        const float bv = (j < dsub && ki < Ks) ? cw[j] : 0.f;
        // reference order is fma(codeword, query, acc); the product commutes exactly
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv, acc, 0, 0, 0);
    }
    if (ipdist) {
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[r] = inv_ks - acc[r];
    }
    if (ki >= Ks) return;
    const int b0 = bt * 16 + kk * 4;  // first of this lane's 4 queries
    if (layout == ANNLITE_LAYOUT_TILED) {
        // [bq][Ks][M][QI]
        if constexpr (QI == 4) {
            // pad queries (b >= B) were multiplied by zero rows: value 0 (or inv_ks) -- harmless
            float *dst = out + ((((int64_t)(b0 / 4)) * Ks + ki) * M + m) * 4;
            *(f32x4 *)dst = acc;
        } else {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int b = b0 + r;
                out[((((int64_t)(b / QI)) * Ks + ki) * M + m) * QI + (b % QI)] = acc[r];
            }
        }
    } else {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int b = b0 + r;
            if (b < B) out[((int64_t)b * M + m) * Ks + ki] = acc[r];
        }
    }
}

__global__ __launch_bounds__(256) void lut_retile_kernel(const float *__restrict__ in, int B, int M, int Ks, int QI,
                                                        float *__restrict__ out, int64_t total) {
    const int64_t id = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (id >= total) return;
    const int i = (int)(id % QI);
    const int m = (int)((id / QI) % M);
    const int k = (int)((id / ((int64_t)QI * M)) % Ks);
    const int bq = (int)(id / ((int64_t)QI * M * Ks));
    const int b = bq * QI + i;
    out[id] = (b < B) ? in[((int64_t)b * M + m) * Ks + k] : 0.f;
}

}  // namespace annlite

using namespace annlite;

extern "C" int annlite_lut_build(int kind, const float *queries_dev, int64_t B, int64_t D, const float *codebooks_dev,
                                 int64_t M, int64_t Ks, float *out_dev, int layout, int qi, void *stream) {
    ANNLITE_REQUIRE(kind == ANNLITE_LUT_L2 || kind == ANNLITE_LUT_IP || kind == ANNLITE_LUT_IPDIST, "bad LUT kind %d",
                    kind);
    ANNLITE_REQUIRE(layout == ANNLITE_LAYOUT_BMK || layout == ANNLITE_LAYOUT_TILED, "bad layout %d", layout);
    ANNLITE_REQUIRE(B >= 0 && M >= 1 && Ks >= 1 && D >= M && D % M == 0,
                    "input dimension must be dividable by number of sub-space (D=%lld, M=%lld)", (long long)D,
                    (long long)M);
    if (B == 0) return ANNLITE_OK;
    ANNLITE_REQUIRE(queries_dev && codebooks_dev && out_dev, "null device pointer");
    if (layout == ANNLITE_LAYOUT_TILED) ANNLITE_REQUIRE(qi == 4 || qi == 2 || qi == 1, "qi must be 4, 2 or 1");
    const int dsub = (int)(D / M);
    const float inv_ks = (float)(1.0 / (double)Ks);
    hipStream_t st = (hipStream_t)stream;

    if (layout == ANNLITE_LAYOUT_TILED && qi == 1) {
        // QI == 1 tiled layout is [B][Ks][M]; only used by tests of odd shapes -> build BMK then retile
        set_error("qi == 1 tiled layout: build BMK and call annlite_lut_retile");
        return ANNLITE_ERR_UNSUPPORTED;
    }
    if (kind == ANNLITE_LUT_L2) {
        if (layout == ANNLITE_LAYOUT_BMK) {
            const int64_t total = B * M * Ks;
            hipLaunchKernelGGL(lut_bmk_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, kind,
                               queries_dev, (int)B, (int)D, codebooks_dev, (int)M, (int)Ks, dsub, inv_ks, out_dev,
                               total);
            return launch_status("lut_bmk_kernel");
        }
        const int64_t nbq = ((B + 15) / 16) * 16 / qi;  // TILED buffers are padded to 16 queries
        const int64_t total = nbq * Ks * M;
        if (qi == 4)
            hipLaunchKernelGGL(lut_l2_tiled_kernel<4>, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st,
                               queries_dev, (int)B, (int)D, codebooks_dev, (int)M, (int)Ks, dsub, out_dev, total);
        else
            hipLaunchKernelGGL(lut_l2_tiled_kernel<2>, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st,
                               queries_dev, (int)B, (int)D, codebooks_dev, (int)M, (int)Ks, dsub, out_dev, total);
        return launch_status("lut_l2_tiled_kernel");
    }
    // inner-product kinds: matrix cores
    const int n_bt = (int)((B + 15) / 16), n_kt = (int)((Ks + 15) / 16);
    const int64_t n_w = (int64_t)n_bt * n_kt * M;
    const int ipdist = kind == ANNLITE_LUT_IPDIST;
    // TILED buffers are padded to a multiple of 16 queries (annlite_scan_plan.lut_floats); the
    // kernel writes whole 16-query tiles, pad queries come out as 0 (or 1/Ks) and are never returned.
    if (layout == ANNLITE_LAYOUT_TILED && qi == 2)
        hipLaunchKernelGGL(lut_ip_mfma_kernel<2>, dim3((unsigned)((n_w + 3) / 4)), dim3(256), 0, st, queries_dev,
                           (int)B, (int)D, codebooks_dev, (int)M, (int)Ks, dsub, ipdist, inv_ks, out_dev, layout, n_bt,
                           n_kt);
    else
        hipLaunchKernelGGL(lut_ip_mfma_kernel<4>, dim3((unsigned)((n_w + 3) / 4)), dim3(256), 0, st, queries_dev,
                           (int)B, (int)D, codebooks_dev, (int)M, (int)Ks, dsub, ipdist, inv_ks, out_dev, layout, n_bt,
                           n_kt);
    return launch_status("lut_ip_mfma_kernel");
}

extern "C" int annlite_lut_retile(const float *lut_bmk_dev, int64_t B, int64_t M, int64_t Ks, float *out_tiled_dev,
                                  int qi, void *stream) {
    ANNLITE_REQUIRE(B >= 0 && M >= 1 && Ks >= 1, "bad shape");
    ANNLITE_REQUIRE(qi == 4 || qi == 2 || qi == 1, "qi must be 4, 2 or 1");
    if (B == 0) return ANNLITE_OK;
    ANNLITE_REQUIRE(lut_bmk_dev && out_tiled_dev, "null device pointer");
    const int64_t nbq = ((B + 15) / 16) * 16 / qi;  // padded to 16 queries like annlite_lut_build
    const int64_t total = nbq * Ks * M * qi;
    hipLaunchKernelGGL(lut_retile_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       lut_bmk_dev, (int)B, (int)M, (int)Ks, qi, out_tiled_dev, total);
    return launch_status("lut_retile_kernel");
}
