/*
 * pq_oracle.c -- CPU restatement of the reference's PQ / ADC hot path, plain C.
 *
 * TEST INFRASTRUCTURE, NOT PRODUCT.  Only tests/, __graft_entry__.smoke() and bench.py's
 * `cpu_baseline` leg may load this library (via oracle/pq_oracle.py); annlite_amd/ never does.
 *
 * Every function restates one reference loop and cites it (paths relative to the reference repo,
 * jina-ai/annlite v0.5.11).  Arithmetic facts that make the restatement BIT-EXACT (established
 * against the compiled reference, see tests/test_oracle_vs_reference.py and SURVEY.md section 8c):
 *   - the LUT loops (`acc += c*c`, `acc += cw*q`) are compiled by the reference's own flags
 *     (-O3 -march=native, setup.py:125-144) into a fused multiply-add per j, sequential in j.
 *     They are written here as explicit fmaf() so the result does not depend on this file's
 *     -ffp-contract setting; the Makefile still compiles with -mfma so fmaf is one instruction.
 *   - the ADC sum is a plain fp32 add chain in ascending m starting from 0.0f.
 * Build: see oracle/Makefile  (gcc -O3 -march=x86-64-v3 -ffp-contract=off -fopenmp; v3 = AVX2+FMA, so
 * the .so built in the build container also runs on the GPU box host, unlike -march=native).
 */
#include <math.h>
#include <stddef.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#ifdef _OPENMP
#include <omp.h>
#endif

#define ORACLE_API __attribute__((visibility("default")))

/* ---------------------------------------------------------------------------------------------
 * L2^2 look-up table for ONE query.
 * reference: bindings/pq_bindings.pyx:85-145  precompute_adc_table
 *   for m: for k: acc = 0; for j: c = codebooks[m,k,j] - query[m*dsub+j]; acc += c*c
 * out: [M][Ks]
 * ------------------------------------------------------------------------------------------- */
ORACLE_API void oracle_precompute_adc_table(const float *query, int64_t D, int64_t dsub,
                                            int64_t Ks, const float *codebooks, float *out) {
    const int64_t M = D / dsub;
    for (int64_t m = 0; m < M; ++m) {
        const float *q = query + m * dsub;
        for (int64_t k = 0; k < Ks; ++k) {
            const float *cw = codebooks + (m * Ks + k) * dsub;
            float acc = 0.f;
            for (int64_t j = 0; j < dsub; ++j) {
                const float c = cw[j] - q[j];
                acc = fmaf(c, c, acc);
            }
            out[m * Ks + k] = acc;
        }
    }
}

/* reference: bindings/pq_bindings.pyx:149-210  batch_precompute_adc_table ; out: [B][M][Ks] */
ORACLE_API void oracle_batch_precompute_adc_table(const float *queries, int64_t B, int64_t D,
                                                  int64_t dsub, int64_t Ks, const float *codebooks,
                                                  float *out, int threads) {
    const int64_t M = D / dsub;
#pragma omp parallel for schedule(static) num_threads(threads > 0 ? threads : 1)
    for (int64_t b = 0; b < B; ++b)
        oracle_precompute_adc_table(queries + b * D, D, dsub, Ks, codebooks, out + b * M * Ks);
}

/* Inner-product look-up table.
 * reference: bindings/pq_bindings.pyx:214-274  batch_precompute_adc_table_ip
 *   acc = 0; for j: acc += codebooks[m,k,j] * queries[b, m*dsub+j]        ; out: [B][M][Ks] */
ORACLE_API void oracle_batch_precompute_adc_table_ip(const float *queries, int64_t B, int64_t D,
                                                     int64_t dsub, int64_t Ks,
                                                     const float *codebooks, float *out,
                                                     int threads) {
    const int64_t M = D / dsub;
#pragma omp parallel for schedule(static) num_threads(threads > 0 ? threads : 1)
    for (int64_t b = 0; b < B; ++b) {
        for (int64_t m = 0; m < M; ++m) {
            const float *q = queries + b * D + m * dsub;
            for (int64_t k = 0; k < Ks; ++k) {
                const float *cw = codebooks + (m * Ks + k) * dsub;
                float acc = 0.f;
                for (int64_t j = 0; j < dsub; ++j) acc = fmaf(cw[j], q[j], acc);
                out[(b * M + m) * Ks + k] = acc;
            }
        }
    }
}

/* Metric dispatch of the batched tables.
 * reference: annlite/core/codec/pq.py:293-325  PQCodec.get_dist_mat
 *   EUCLIDEAN(1)            -> batch_precompute_adc_table
 *   INNER_PRODUCT(2)/COSINE(3) -> float32(1/n_clusters) - batch_precompute_adc_table_ip   (pq.py:316-322;
 *       note 1/n_CLUSTERS: the summed distance is M/Ks - <q, x^>)
 * The cosine re-normalisation of x (pq.py:309-310) is done by the caller (oracle/pq_oracle.py uses
 * numpy exactly like annlite/math.py:6-18).  metric values: annlite/enums.py:25-28. */
ORACLE_API int oracle_get_dist_mat(int metric, const float *queries, int64_t B, int64_t D,
                                   int64_t dsub, int64_t Ks, const float *codebooks, float *out,
                                   int threads) {
    const int64_t M = D / dsub;
    if (metric == 1) {
        oracle_batch_precompute_adc_table(queries, B, D, dsub, Ks, codebooks, out, threads);
        return 0;
    }
    if (metric == 2 || metric == 3) {
        oracle_batch_precompute_adc_table_ip(queries, B, D, dsub, Ks, codebooks, out, threads);
        const float inv = (float)(1.0 / (double)Ks); /* python float 1/Ks, weak-scalar -> float32 */
        const int64_t n = B * M * Ks;
        for (int64_t i = 0; i < n; ++i) out[i] = inv - out[i];
        return 0;
    }
    return 1;
}

/* ---------------------------------------------------------------------------------------------
 * Flat ADC scan.
 * reference: bindings/pq_bindings.pyx:30-47 (dist_pqcode_to_codebook) + 52-80
 *            (dist_pqcodes_to_codebooks):  dist = 0; for m in range(M): dist += adtable[m, code[m]]
 * Same arithmetic as include/hnswlib/space_pq.h:15-37 (PQLookup) for one candidate row.
 * adtable: [M][Ks] ; codes: [N][M] (uint8 / uint16 / uint32, pq.py:56-60) ; out: [N]
 * ------------------------------------------------------------------------------------------- */
#define DEFINE_ADC_SCAN(NAME, T)                                                                 \
    ORACLE_API void NAME(const float *adtable, int64_t M, int64_t Ks, const T *codes, int64_t N, \
                         float *out, int threads) {                                              \
        _Pragma("omp parallel for schedule(static) num_threads(threads > 0 ? threads : 1)")      \
        for (int64_t n = 0; n < N; ++n) {                                                        \
            const T *c = codes + n * M;                                                          \
            float dist = 0.f;                                                                    \
            for (int64_t m = 0; m < M; ++m) dist += adtable[m * Ks + (int64_t)c[m]];             \
            out[n] = dist;                                                                       \
        }                                                                                        \
    }
DEFINE_ADC_SCAN(oracle_dist_pqcodes_to_codebooks_u8, uint8_t)
DEFINE_ADC_SCAN(oracle_dist_pqcodes_to_codebooks_u16, uint16_t)
DEFINE_ADC_SCAN(oracle_dist_pqcodes_to_codebooks_u32, uint32_t)

/* Gathered ADC (the HNSW-over-PQ rerank shape): dist for rows cand[i] only.
 * reference: include/hnswlib/space_pq.h:15-37 PQLookup, called per visited node. cand < 0 -> +inf. */
ORACLE_API void oracle_adc_gather_u8(const float *adtable, int64_t M, int64_t Ks,
                                     const uint8_t *codes, const int64_t *cand, int64_t R,
                                     float *out) {
    for (int64_t i = 0; i < R; ++i) {
        if (cand[i] < 0) { out[i] = INFINITY; continue; }
        const uint8_t *c = codes + cand[i] * M;
        float dist = 0.f;
        for (int64_t m = 0; m < M; ++m) dist += adtable[m * Ks + (int64_t)c[m]];
        out[i] = dist;
    }
}

/* ---------------------------------------------------------------------------------------------
 * top-k smallest.
 * reference: annlite/math.py:94-120 top_k (argpartition + argsort; tie order unspecified there).
 * The build fixes the tie-break to (distance ascending, row id ascending); that is what this
 * function returns and what the HIP kernels must reproduce bit-for-bit.  If k > N the tail is
 * padded with (+inf, -1).
 * ------------------------------------------------------------------------------------------- */
typedef struct { float d; int64_t i; } oracle_pair_t;

/* numpy's order, which the reference's argpartition / argsort use (math.py:107-116): NaN sorts behind every number,
 * +inf included, whatever its sign bit; all NaNs tie (and ties go by row id, the build's fixed rule). */
static int pair_less(const oracle_pair_t *a, const oracle_pair_t *b) {
    const int an = a->d != a->d, bn = b->d != b->d;
    if (an || bn) return an == bn ? a->i < b->i : bn;
    if (a->d < b->d) return 1;
    if (a->d > b->d) return 0;
    return a->i < b->i;
}

/* bounded max-heap of the k best so far; root = worst kept */
static void heap_sift_down(oracle_pair_t *h, int64_t n, int64_t i) {
    for (;;) {
        int64_t l = 2 * i + 1, r = l + 1, w = i;
        if (l < n && pair_less(&h[w], &h[l])) w = l;
        if (r < n && pair_less(&h[w], &h[r])) w = r;
        if (w == i) return;
        oracle_pair_t t = h[i]; h[i] = h[w]; h[w] = t;
        i = w;
    }
}

static int pair_cmp_qsort(const void *a, const void *b) {
    const oracle_pair_t *x = (const oracle_pair_t *)a, *y = (const oracle_pair_t *)b;
    if (pair_less(x, y)) return -1;
    if (pair_less(y, x)) return 1;
    return 0;
}

ORACLE_API void oracle_topk(const float *values, int64_t N, int64_t k, int64_t id_base,
                            float *out_d, int64_t *out_i) {
    oracle_pair_t *h = (oracle_pair_t *)malloc(sizeof(oracle_pair_t) * (size_t)(k > 0 ? k : 1));
    int64_t n = 0;
    for (int64_t i = 0; i < N; ++i) {
        oracle_pair_t p = {values[i], id_base + i};
        if (n < k) {
            h[n++] = p;
            if (n == k)
                for (int64_t j = k / 2 - 1; j >= 0; --j) heap_sift_down(h, n, j);
        } else if (pair_less(&p, &h[0])) {
            h[0] = p;
            heap_sift_down(h, n, 0);
        }
    }
    qsort(h, (size_t)n, sizeof(oracle_pair_t), pair_cmp_qsort);
    for (int64_t j = 0; j < k; ++j) {
        if (j < n) { out_d[j] = h[j].d; out_i[j] = h[j].i; }
        else { out_d[j] = INFINITY; out_i[j] = -1; }
    }
    free(h);
}

/* ---------------------------------------------------------------------------------------------
 * Batched flat PQ search = the PQIndex.search semantics, one query per LUT row.
 * reference: annlite/core/index/pq_index.py:29-56 (LUT -> adist over ALL capacity rows -> top_k),
 *            looped over queries like annlite/container.py:214.
 * lut: [B][M][Ks] (any metric, already built) ; codes: [N][M] u8 ; out: [B][k]
 * threads: OpenMP over queries (variant B "generous" baseline of BASELINE.md section 3);
 *          threads=1 is the reference's own execution model.
 * ------------------------------------------------------------------------------------------- */
ORACLE_API void oracle_adc_search_u8(const float *lut, int64_t B, int64_t M, int64_t Ks,
                                     const uint8_t *codes, int64_t N, int64_t k, int64_t id_base,
                                     float *out_d, int64_t *out_i, int threads) {
#pragma omp parallel num_threads(threads > 0 ? threads : 1)
    {
        float *dist = (float *)malloc(sizeof(float) * (size_t)(N > 0 ? N : 1));
#pragma omp for schedule(dynamic, 1)
        for (int64_t b = 0; b < B; ++b) {
            oracle_dist_pqcodes_to_codebooks_u8(lut + b * M * Ks, M, Ks, codes, N, dist, 1);
            oracle_topk(dist, N, k, id_base, out_d + b * k, out_i + b * k);
        }
        free(dist);
    }
}

/* ---------------------------------------------------------------------------------------------
 * Encode: nearest codeword per subspace, first minimum wins.
 * reference: annlite/core/codec/pq.py:158-177 -> scipy.cluster.vq.vq per subspace (scipy is an
 * un-vendored dependency, installed 1.15.3; its float32 path expands |x|^2+|c|^2-2x.c through a
 * BLAS GEMM, so near-ties are "parity unpinned" -- SURVEY.md section 8c).  This restatement uses
 * the exact direct form  d(k) = sum_j fma(c,c,.) with c = cw[j]-x[j]  (same arithmetic as the LUT),
 * and is checked against scipy's answer with "mismatch only where the top-2 gap is below 1e-5
 * relative" (tests/test_oracle_golden.py).  oracle/pq_oracle.py also exposes the literal scipy call.
 * out codes: [N][M] as uint32 (caller narrows to pq.py:56-60's dtype).
 * ------------------------------------------------------------------------------------------- */
ORACLE_API void oracle_encode(const float *x, int64_t N, int64_t D, int64_t dsub, int64_t Ks,
                              const float *codebooks, uint32_t *out, int threads) {
    const int64_t M = D / dsub;
#pragma omp parallel for schedule(static) num_threads(threads > 0 ? threads : 1)
    for (int64_t n = 0; n < N; ++n) {
        for (int64_t m = 0; m < M; ++m) {
            const float *q = x + n * D + m * dsub;
            float best = INFINITY;
            uint32_t arg = 0;
            for (int64_t k = 0; k < Ks; ++k) {
                const float *cw = codebooks + (m * Ks + k) * dsub;
                float acc = 0.f;
                for (int64_t j = 0; j < dsub; ++j) {
                    const float c = cw[j] - q[j];
                    acc = fmaf(c, c, acc);
                }
                if (acc < best) { best = acc; arg = (uint32_t)k; }
            }
            out[n * M + m] = arg;
        }
    }
}

/* Decode: gather codewords.  reference: annlite/core/codec/pq.py:179-198 */
ORACLE_API void oracle_decode_u8(const uint8_t *codes, int64_t N, int64_t M, int64_t dsub,
                                 int64_t Ks, const float *codebooks, float *out) {
    for (int64_t n = 0; n < N; ++n)
        for (int64_t m = 0; m < M; ++m)
            memcpy(out + (n * M + m) * dsub, codebooks + (m * Ks + codes[n * M + m]) * dsub,
                   sizeof(float) * (size_t)dsub);
}

ORACLE_API int oracle_max_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}
