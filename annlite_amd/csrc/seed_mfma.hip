// seed_mfma.hip -- candidate rows for the seed bound from ONE bf16 MFMA contraction (round 6).
//
// The byte-table scan starts from a first bound per query: a key that provably has k rows of the table at or below it
// (scan_prep.hip: seed_bound_kernel).  Rounds 2-5 got it from the EXACT ADC sums of S seed rows spread over the table --
// S x B x M look-up-adds on the VALU / LDS (5.4 10^8 at 32768 rows x 1024 queries: 22 us of the 39 us preparation launch, at a
// quarter of the scan kernel's own look-up rate).  The bound does not need those sums: ANY k distinct valid rows give a valid
// bound -- the largest of their exact sums -- and it is tight when the rows are (nearly) the S rows' k best.  Finding rows that
// are NEARLY best is a dense contraction:
//     sum_m lut[b][m][code[n][m]]  =  |q_b - x^_n|^2  =  |q_b|^2 + |x^_n|^2 - 2 <q_b, x^_n>      (x^_n = the decoded row)
// i.e. [S rows x 128] x [128 x B queries] on the matrix cores, bf16 in / fp32 accumulate (v_mfma_f32_32x32x16_bf16, 8.6 GFLOP:
// ~4 us at the dense peak).  Nothing of it is ever returned: the launch only NOMINATES rows -- per query the best row of each of
// 512 disjoint groups of S / 512 seed rows, by approximate distance -- and the preparation launch computes the nominees' EXACT
// ascending-m fp32 sums (the reference's arithmetic, pq_bindings.pyx:30-47) and takes the k-th smallest of them, exactly as it
// did with the lane minima of the rows it scanned itself.  bf16 rounding can only make the nomination slightly worse (a
// slightly looser, still valid bound); results stay bit-exact.
//
// Workgroup = 128 queries x S / 32 seed rows, 8 waves = 2 query halves (64 queries: two 32-column MFMA blocks) x 4 row quarters.
//   B operand (queries, -2 q as bf16; + one extra K step carrying |q|^2 as a bf16 hi / lo pair): in registers for the whole kernel.
//   A operand (seed rows): never materialised -- the 8 bf16 a lane feeds to K step kk are the code word of sub-space
//     m = 2 kk + (lane >> 5) of row (lane & 31) of the tile: ONE ds_read_b128 from the bf16 copy of the codebooks in LDS
//     (64 KB, [m][code][8 bf16], converted by the workgroup from the fp32 codebooks at its start); the extra K step carries
//     |x^_n|^2 = sum_m |c_m|^2 (hi / lo pair, from a 16 KB table of code-word norms) -- +inf for rows that are deleted or
//     beyond the table, which then never win.
//   C / D: column = query, 16 rows per lane.  Per tile the lane's 16 sums become keys (float bits with the low byte replaced by
//     (tile in group, register)), their minimum joins the running minimum of the group -- 26 VALU per 32 x 32 tile beside 9 MFMAs.
// Output: cand u32 [B][512] table rows (0xffffffff: the group held no valid row).
#include "scan_common.h"

namespace annlite {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int kSeedMfmaLds = 65536 + 16384;  // bf16 codebooks [16][256][16 B] + code-word norms f32 [16][256]

struct SeedMfmaArgs {
    const float *queries;   // [B][128]
    const float *cb;        // [16][Ks][8]
    const uint8_t *codes;   // [N][16]
    const uint32_t *valid;  // optional bitmap
    uint32_t *cand;         // out [B][kSeedCand]
    int64_t N;              // rows of the table (the seed rows are spread over [0, N))
    int64_t run_step;       // rows between the starts of two runs of seed blocks (seed_row_of)
    int32_t B, Ks;
    int32_t tiles;          // 32-row tiles per wave: S / 32 slices / 4 quarters / 32 rows (even)
    int32_t chunk_log;      // runs of 2^chunk_log blocks of 64 rows
};

// seed row s of S (64-row blocks in runs of 2^clog blocks, the runs spread evenly over the table: scan_prep.hip)
__device__ __forceinline__ int64_t seed_row_of(int64_t s, int clog, int64_t run_step) {
    const int64_t b = s >> 6;
    return (b >> clog) * run_step + ((b & ((1 << clog) - 1)) << 6) + (s & 63);
}

__device__ __forceinline__ uint32_t bf16_bits(float f) {  // RNE, hardware conversion
    const __bf16 h = (__bf16)f;
    return (uint32_t)__builtin_bit_cast(unsigned short, h);
}
__device__ __forceinline__ float bf16_to_f32(uint32_t b) { return __uint_as_float(b << 16); }

template <bool SKEWED>
__global__ __launch_bounds__(512) void seed_mfma_candidates_kernel(const SeedMfmaArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    typedef __attribute__((address_space(3))) unsigned char *lds_bytes;
    const uint32_t lds0 = (uint32_t)(uintptr_t)(lds_bytes)smem;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int qh = wave & 1, h4 = wave >> 1;  // query half (64 queries), row quarter
    const int col = lane & 31, kh = lane >> 5;
    const int Ks = a.Ks;

    // ---- the fp32 code words this thread converts (requested first: their round trip runs under the query loads below) ----
    f32x4 cw[8][2];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
        const int idx = tid + 512 * u;
        const f32x4 *src = (const f32x4 *)(a.cb + (int64_t)(idx < 16 * Ks ? idx : 0) * 8);
        cw[u][0] = src[0];
        cw[u][1] = src[1];
    }

    // ---- B operand: this wave's 2 x 32 queries, -2 q in bf16, K step kk = dims [16 kk + 8 kh, + 8); step 8: (|q|^2 hi, lo, 1, 1) ----
    bf16x8 bq[2][9];
#pragma unroll
    for (int qb = 0; qb < 2; ++qb) {
        const int j = blockIdx.x * 128 + qh * 64 + qb * 32 + col;
        const bool real = j < a.B;
        const f32x4 *src = (const f32x4 *)(a.queries + (int64_t)(real ? j : 0) * 128 + 8 * kh);
        float nrm = 0.f;
#pragma unroll
        for (int kk = 0; kk < 8; ++kk) {
            f32x4 v0 = src[4 * kk], v1 = src[4 * kk + 1];  // (16 floats = 4 f32x4 per K step; this half's 8)
            if (!real) v0 = (f32x4){0.f, 0.f, 0.f, 0.f}, v1 = v0;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                nrm = __builtin_fmaf(v0[e], v0[e], nrm);
                nrm = __builtin_fmaf(v1[e], v1[e], nrm);
                bq[qb][kk][e] = (__bf16)(-2.f * v0[e]);
                bq[qb][kk][4 + e] = (__bf16)(-2.f * v1[e]);
            }
        }
        nrm += __shfl_xor(nrm, 32);  // (the other half's 64 dims)
        const uint32_t hi = bf16_bits(nrm);
        const uint32_t lo = bf16_bits(nrm - bf16_to_f32(hi));
        u32x4 ex = {0u, 0u, 0u, 0u};
        if (kh == 0) ex = (u32x4){hi | (lo << 16), 0x3f803f80u, 0u, 0u};  // (|q|^2 hi, lo, 1, 1, 0, 0, 0, 0)
        bq[qb][8] = __builtin_bit_cast(bf16x8, ex);
    }
    // ---- bf16 copy of the codebooks + the code words' squared norms (fp32, from the fp32 code words) ----------------------
#pragma unroll
    for (int u = 0; u < 8; ++u) {
        const int idx = tid + 512 * u;
        if (idx < 16 * Ks) {
            const int m = idx / Ks, code = idx - m * Ks;
            bf16x8 h;
            float nrm = 0.f;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                h[e] = (__bf16)cw[u][0][e];
                h[4 + e] = (__bf16)cw[u][1][e];
                nrm = __builtin_fmaf(cw[u][0][e], cw[u][0][e], nrm);
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) nrm = __builtin_fmaf(cw[u][1][e], cw[u][1][e], nrm);
            *(__attribute__((address_space(3))) bf16x8 *)(uintptr_t)(lds0 + (uint32_t)((m * 256 + code) * 16)) = h;
            *(__attribute__((address_space(3))) float *)(uintptr_t)(lds0 + 65536u + (uint32_t)((m * 256 + code) * 4)) = nrm;
        }
    }
    __syncthreads();  // (the codebooks are in LDS)

    // PLAIN order of a stored row: SKEWED rows hold the code of sub-space (j + n) mod 16 at byte j -- rotate left by (16 - n % 16);
    // n % 16 == lane % 16 (tiles start at multiples of 32 rows inside 64-row blocks that start at multiples of 64)
    const int sinv = (16 - (lane & 15)) & 15;
    bool abit[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) abit[i] = (((sinv >> 2) >> i) & 1) != 0;
    const uint32_t bsh = (uint32_t)(sinv & 3);
    const uint32_t off0 = 8u * (uint32_t)kh, off1 = 16u + 8u * (uint32_t)kh;  // bit offset of sub-space 2 kk + kh in dword kk >> 1
    const uint32_t cbase = lds0 + 4096u * (uint32_t)kh;   // + kk * 8192 (immediate) + (code << 4)
    const uint32_t nbase = lds0 + 65536u + 1024u * (uint32_t)kh;  // + kk * 2048 + (code << 2)

    const int nt = a.tiles, half_nt = nt >> 1;
    const int64_t s_wave = ((int64_t)blockIdx.y * 4 + h4) * nt * 32;  // first seed row of this wave
    auto tile_row0 = [&](int t) -> int64_t { return seed_row_of(s_wave + (int64_t)t * 32, a.chunk_log, a.run_step); };
    auto load_codes = [&](int64_t row, u32x4 &c, uint32_t &vw) {
        const int64_t rr = row < a.N ? row : a.N - 1;
        c = *(const u32x4 *)(a.codes + rr * 16);
        vw = a.valid ? a.valid[rr >> 5] : ~0u;
    };
    uint32_t run[2][2];
#pragma unroll
    for (int qb = 0; qb < 2; ++qb) run[qb][0] = run[qb][1] = 0xffffffffu;
    u32x4 cn;
    uint32_t vn;
    int64_t row_n = tile_row0(0) + col;
    load_codes(row_n, cn, vn);
    for (int t = 0; t < nt; ++t) {
        const int64_t row = row_n;
        uint32_t cc[4] = {cn.x, cn.y, cn.z, cn.w};
        const bool ok = row < a.N && ((vn >> (row & 31)) & 1u);
        if (t + 1 < nt) {
            row_n = tile_row0(t + 1) + col;
            load_codes(row_n, cn, vn);
        }
        if constexpr (SKEWED) rotate_row<4>(cc, abit, bsh);
        // look-up addresses of this lane's 8 code words, and |x^|^2 from the norms of all 16 (the partner lane has the other 8)
        uint32_t ad[8];
        float xn = 0.f;
#pragma unroll
        for (int kk = 0; kk < 8; ++kk) {
            const uint32_t code = __builtin_amdgcn_ubfe(cc[kk >> 1], (kk & 1) ? off1 : off0, 8u);
            ad[kk] = cbase + (code << 4);
            xn += *(const __attribute__((address_space(3))) float *)(uintptr_t)(nbase + (uint32_t)(kk * 2048) + (code << 2));
        }
        xn += __shfl_xor(xn, 32);
        bf16x8 ax;
        {
            uint32_t hi = bf16_bits(xn);
            uint32_t lo = bf16_bits(xn - bf16_to_f32(hi));
            if (!ok) hi = 0x7f80u, lo = 0u;  // +inf: a row that is deleted / beyond the table never wins
            u32x4 ex = {0u, 0u, 0u, 0u};
            if (kh == 0) ex = (u32x4){0x3f803f80u, hi | (lo << 16), 0u, 0u};  // (1, 1, |x^|^2 hi, lo, 0, 0, 0, 0)
            ax = __builtin_bit_cast(bf16x8, ex);
        }
        f32x16 acc0, acc1;
        {
            const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
            acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ax, bq[0][8], zero, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ax, bq[1][8], zero, 0, 0, 0);
        }
        static_for<0, 8>([&](auto KK) {
            constexpr int kk = decltype(KK)::value;
            const bf16x8 af = *(const __attribute__((address_space(3))) bf16x8 *)(uintptr_t)(ad[kk] + (uint32_t)(kk * 8192));
            acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, bq[0][kk], acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, bq[1][kk], acc1, 0, 0, 0);
        });
        // keys: float bits (>= 0 up to rounding; a slightly negative sum -- a near-duplicate of the query -- reads as a huge key and
        // is passed over: the bound only gets looser) with the low byte = (tile in its group) << 4 | register
        const int g = t >= half_nt ? 1 : 0;
        const uint32_t tbits = (uint32_t)(t - g * half_nt) << 4;
        uint32_t m0 = 0xffffffffu, m1 = 0xffffffffu;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const uint32_t k0 = (__float_as_uint(acc0[r]) & 0xffffff00u) | (uint32_t)r;
            const uint32_t k1 = (__float_as_uint(acc1[r]) & 0xffffff00u) | (uint32_t)r;
            m0 = k0 < m0 ? k0 : m0;
            m1 = k1 < m1 ? k1 : m1;
        }
        m0 |= tbits;
        m1 |= tbits;
        if (g == 0) {
            run[0][0] = m0 < run[0][0] ? m0 : run[0][0];
            run[1][0] = m1 < run[1][0] ? m1 : run[1][0];
        } else {
            run[0][1] = m0 < run[0][1] ? m0 : run[0][1];
            run[1][1] = m1 < run[1][1] ? m1 : run[1][1];
        }
    }
    // ---- the nominee of every (query, group): cand[b][slice * 16 + quarter * 4 + half * 2 + group] -------------------------
#pragma unroll
    for (int qb = 0; qb < 2; ++qb) {
        const int j = blockIdx.x * 128 + qh * 64 + qb * 32 + col;
        if (j >= a.B) continue;
#pragma unroll
        for (int g = 0; g < 2; ++g) {
            const uint32_t key = run[qb][g];
            uint32_t row = 0xffffffffu;
            if (key < 0x7f800000u) {  // (finite, non-negative approximate distance)
                const int r = (int)(key & 15u), t = g * half_nt + (int)((key >> 4) & 15u);
                const int row_in_tile = (r & 3) + 8 * (r >> 2) + 4 * kh;
                const int64_t tr = tile_row0(t) + row_in_tile;
                if (tr < a.N) row = (uint32_t)tr;
            }
            a.cand[(int64_t)j * kSeedCand + (int)blockIdx.y * 16 + h4 * 4 + kh * 2 + g] = row;
        }
    }
}

}  // namespace annlite

using namespace annlite;

// S seed rows (a multiple of 8192, <= 32 slices x 4 quarters x 32 tiles x 32 rows) spread over the table's N rows like the exact
// seed scan's (runs of 2^chunk_log blocks of 64 rows); cand [B][kSeedCand]
int annlite::launch_seed_mfma(bool skewed, const float *queries_dev, int64_t B, const float *cb_dev, int64_t Ks, const void *codes_dev,
                              const uint32_t *valid_bits_dev, int64_t N, int64_t S, int chunk_log, uint32_t *cand_dev, hipStream_t st) {
    ANNLITE_REQUIRE(S >= 8192 && S % 8192 == 0 && S <= 32 * 4 * 32 * 32 && S <= N && Ks >= 1 && Ks <= 256 && B >= 1,
                    "seed_mfma: bad shape S=%lld N=%lld Ks=%lld", (long long)S, (long long)N, (long long)Ks);
    SeedMfmaArgs a;
    a.queries = queries_dev;
    a.cb = cb_dev;
    a.codes = (const uint8_t *)codes_dev;
    a.valid = valid_bits_dev;
    a.cand = cand_dev;
    a.N = N;
    a.B = (int32_t)B;
    a.Ks = (int32_t)Ks;
    a.tiles = (int32_t)(S / (32 * 4 * 32));
    a.chunk_log = chunk_log;
    {
        // (the same spread as seed_bound_kernel's block_row: n_blocks blocks in runs of 2^chunk_log over the extent N)
        const int64_t n_blocks = S >> 6, cmask = (1 << chunk_log) - 1;
        int64_t run_step = ((N >> 6) / ((n_blocks + cmask) >> chunk_log)) << 6;
        if (run_step < ((int64_t)64 << chunk_log)) run_step = (int64_t)64 << chunk_log;
        a.run_step = run_step;
    }
    auto fn = skewed ? seed_mfma_candidates_kernel<true> : seed_mfma_candidates_kernel<false>;
    ANNLITE_HIP_TRY(hipFuncSetAttribute((const void *)fn, hipFuncAttributeMaxDynamicSharedMemorySize, kSeedMfmaLds));
    hipLaunchKernelGGL(fn, dim3((unsigned)((B + 127) / 128), 32), dim3(512), kSeedMfmaLds, st, a);
    return launch_status("seed_mfma_candidates_kernel");
}
