/* ASan/UBSan driver for oracle/pq_oracle.c (test infrastructure): exercises every exported function on
 * small ragged shapes so that out-of-bounds reads/writes or UB in the C restatement are caught on CPU
 * (the reference itself runs with boundscheck off: bindings/pq_bindings.pyx:28-29, 50-51). */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

void oracle_precompute_adc_table(const float *, int64_t, int64_t, int64_t, const float *, float *);
void oracle_batch_precompute_adc_table(const float *, int64_t, int64_t, int64_t, int64_t, const float *, float *, int);
void oracle_batch_precompute_adc_table_ip(const float *, int64_t, int64_t, int64_t, int64_t, const float *, float *, int);
int oracle_get_dist_mat(int, const float *, int64_t, int64_t, int64_t, int64_t, const float *, float *, int);
void oracle_dist_pqcodes_to_codebooks_u8(const float *, int64_t, int64_t, const uint8_t *, int64_t, float *, int);
void oracle_dist_pqcodes_to_codebooks_u16(const float *, int64_t, int64_t, const uint16_t *, int64_t, float *, int);
void oracle_adc_gather_u8(const float *, int64_t, int64_t, const uint8_t *, const int64_t *, int64_t, float *);
void oracle_topk(const float *, int64_t, int64_t, int64_t, float *, int64_t *);
void oracle_adc_search_u8(const float *, int64_t, int64_t, int64_t, const uint8_t *, int64_t, int64_t, int64_t, float *,
                          int64_t *, int);
void oracle_encode(const float *, int64_t, int64_t, int64_t, int64_t, const float *, uint32_t *, int);
void oracle_decode_u8(const uint8_t *, int64_t, int64_t, int64_t, int64_t, const float *, float *);

static float frand(void) { return (float)rand() / (float)RAND_MAX; }

int main(void) {
    const int64_t shapes[][5] = {{3, 5, 17, 57, 4}, {16, 8, 256, 130, 3}, {8, 16, 256, 1, 1}, {4, 1, 2, 9, 2}}; /* M dsub Ks N B */
    for (unsigned s = 0; s < sizeof(shapes) / sizeof(shapes[0]); ++s) {
        const int64_t M = shapes[s][0], dsub = shapes[s][1], Ks = shapes[s][2], N = shapes[s][3], B = shapes[s][4], D = M * dsub;
        float *cb = malloc(sizeof(float) * M * Ks * dsub), *q = malloc(sizeof(float) * B * D), *x = malloc(sizeof(float) * N * D);
        float *lut = malloc(sizeof(float) * B * M * Ks), *dist = malloc(sizeof(float) * N), *dec = malloc(sizeof(float) * N * D);
        uint32_t *c32 = malloc(sizeof(uint32_t) * N * M);
        uint8_t *c8 = malloc((size_t)(N * M));
        uint16_t *c16 = malloc(sizeof(uint16_t) * N * M);
        for (int64_t i = 0; i < M * Ks * dsub; ++i) cb[i] = frand();
        for (int64_t i = 0; i < B * D; ++i) q[i] = frand();
        for (int64_t i = 0; i < N * D; ++i) x[i] = frand();
        oracle_precompute_adc_table(q, D, dsub, Ks, cb, lut);
        oracle_batch_precompute_adc_table(q, B, D, dsub, Ks, cb, lut, 2);
        oracle_batch_precompute_adc_table_ip(q, B, D, dsub, Ks, cb, lut, 2);
        for (int metric = 1; metric <= 3; ++metric)
            if (oracle_get_dist_mat(metric, q, B, D, dsub, Ks, cb, lut, 1) != 0) return 2;
        if (oracle_get_dist_mat(9, q, B, D, dsub, Ks, cb, lut, 1) == 0) return 3;
        oracle_encode(x, N, D, dsub, Ks, cb, c32, 2);
        for (int64_t i = 0; i < N * M; ++i) { c8[i] = (uint8_t)c32[i]; c16[i] = (uint16_t)c32[i]; }
        oracle_dist_pqcodes_to_codebooks_u8(lut, M, Ks, c8, N, dist, 2);
        oracle_dist_pqcodes_to_codebooks_u16(lut, M, Ks, c16, N, dist, 1);
        oracle_decode_u8(c8, N, M, dsub, Ks, cb, dec);
        const int64_t ks[] = {1, 10, N, N + 7};
        for (unsigned j = 0; j < 4; ++j) {
            const int64_t k = ks[j];
            float *od = malloc(sizeof(float) * B * k);
            int64_t *oi = malloc(sizeof(int64_t) * B * k);
            oracle_topk(dist, N, k, 100, od, oi);
            oracle_adc_search_u8(lut, B, M, Ks, c8, N, k, 0, od, oi, 2);
            for (int64_t b = 0; b < B; ++b)
                for (int64_t t = 1; t < k && t < N; ++t)
                    if (od[b * k + t] < od[b * k + t - 1]) return 4;
            free(od); free(oi);
        }
        int64_t cand[5] = {0, N - 1, -1, N / 2, 0};
        float g[5];
        oracle_adc_gather_u8(lut, M, Ks, c8, cand, 5, g);
        if (!isinf(g[2])) return 5;
        free(cb); free(q); free(x); free(lut); free(dist); free(dec); free(c32); free(c8); free(c16);
    }
    puts("sanitize OK");
    return 0;
}
