// scan_legacy.hip -- the two kernels the default replaced, kept as selectable variants (ANNLITE_SCAN_VARIANT,
// tests/test_gpu_parity.py::test_scan_kernel_variants_agree): the exact two-pass kernel (ordered fp32 sums for
// every row via compile-time EXEC masks) and the fp32 filter kernel (fast rotated-order sum as the bound).
#include "scan_common.h"

namespace annlite {

// =================================================================================================
// Fast kernel: uint8 codes, Ks <= 256, M in {8,16,32,64}, k <= 64.
//   M  sub-spaces            QI queries interleaved per LDS entry (4 -> ds_read_b128, 2 -> b64)
//   NQ entry groups per WG   (QT = QI*NQ queries per workgroup)      NW waves per workgroup
// LDS byte address of (code k, group h, sub-space m): ((k*NQ + h)*M + m) * QI*4
// =================================================================================================
template <int M, int QI, int NQ, int NW, int WPS, bool SKEWED, int MODE>
__global__ __launch_bounds__(NW * 64, WPS) void adc_scan_fast_kernel(const ScanArgs a) {
    constexpr int QT = QI * NQ;
    constexpr int CW = M / 4;              // dwords per code row
    constexpr int EB = QI * 4;             // bytes per LDS entry
    constexpr int RB = M * EB;             // bytes per (k, h) row of entries
    constexpr int KSTRIDE = NQ * RB;       // bytes between consecutive codes k
    constexpr int NP = QI / 2;             // f32x2 pairs per entry
    typedef typename LutVec<QI>::type lutv_t;

    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int km1 = a.k - 1;

    // lane-constant skew
    const int s = lane % M;
    const uint32_t bsh = (uint32_t)(s & 3);
    bool abit[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) abit[i] = (((s >> 2) >> i) & 1) != 0;
    // byte offset inside a (k,h) row for step t: ((s+t) mod M) * EB
    const unsigned char *mbase[M];
#pragma unroll
    for (int t = 0; t < M; ++t) mbase[t] = smem + ((s + t) % M) * EB;
    // MODE 1: per-lane pass weights (w1, w2) for every step
    f32x2 wt[MODE >= 1 ? M : 1];
    if constexpr (MODE >= 1) {
#pragma unroll
        for (int t = 0; t < M; ++t) {
            const bool p1 = (s == 0) || (s >= M - t);
            wt[t] = (f32x2){p1 ? 1.f : 0.f, p1 ? 0.f : 1.f};
        }
    }

    const int n_items = a.n_items;
    const int64_t group_bytes = (int64_t)a.Ks * RB;  // one tiled-LUT group = [Ks][M][QI] floats

    for (int item = blockIdx.x; item < n_items; item += gridDim.x) {
        // item -> (slice, tile): slice % 8 == item % 8 == blockIdx % 8 (the XCD this block lands on,
        // speed only), consecutive items of one XCD walk the tiles of the same slice.
        int tile, slice;
        if (!item_map(a, item, tile, slice)) continue;

        __syncthreads();  // previous item's LDS readers are done
        {
            // fill the LUT tile: NQ groups of [Ks][M][QI] -> LDS [k][h][m][QI]; 16 B pieces
            const unsigned char *src0 = (const unsigned char *)a.lut + (int64_t)tile * NQ * group_bytes;
            constexpr int PIECES_PER_ROW = RB / 16;
            const int total = NQ * a.Ks * PIECES_PER_ROW;
            for (int idx = tid; idx < total; idx += NW * 64) {
                const int p = idx % PIECES_PER_ROW;
                const int kh = idx / PIECES_PER_ROW;  // = h*Ks + k  (source order)
                const int h = kh / a.Ks;
                const int kk = kh - h * a.Ks;
                const u32x4 v = *(const u32x4 *)(src0 + (int64_t)h * group_bytes + (int64_t)kk * RB + p * 16);
                *(u32x4 *)(smem + (kk * NQ + h) * RB + p * 16) = v;
            }
        }
        __syncthreads();

        WaveList list[QT];
        uint32_t thr_hi[QT], thr_lo[QT];
        float thr_f[QT];
#pragma unroll
        for (int q = 0; q < QT; ++q) {
            list[q].reset();
            thr_hi[q] = kKeyInfHi;
            thr_lo[q] = kIdNone;
            thr_f[q] = __builtin_inff();
        }

        const int64_t slice_begin = (int64_t)slice * a.slice_rows;
        int64_t slice_end = slice_begin + a.slice_rows;
        if (slice_end > a.N) slice_end = a.N;

        const uint32_t *codes32 = (const uint32_t *)a.codes;
        auto load_row = [&](int64_t row, uint32_t (&c)[CW]) {
            if (row >= a.N) row = a.N - 1;  // clamped, masked out below
            const uint32_t *p = codes32 + row * CW;
            if constexpr (CW == 2) {
                const u32x2 v = *(const u32x2 *)p;
                c[0] = v.x;
                c[1] = v.y;
            } else {
#pragma unroll
                for (int i = 0; i < CW / 4; ++i) {
                    const u32x4 v = *(const u32x4 *)(p + 4 * i);
                    c[4 * i + 0] = v.x;
                    c[4 * i + 1] = v.y;
                    c[4 * i + 2] = v.z;
                    c[4 * i + 3] = v.w;
                }
            }
        };

        // ---- software-pipelined row loop ---------------------------------------------------------
        // A wave's look-ups of quad h+1 (or of the NEXT row's quad 0) are issued chunk by chunk (8 steps)
        // as soon as pass 2 has consumed that chunk of the current quad, so the LDS latency of one chunk
        // hides behind the adds of the other(s) inside the same 16 (M) value registers.
        constexpr int NCH = M / 8;
        const int64_t stride = (int64_t)NW * 64;
        int64_t row0 = slice_begin + (int64_t)wave * 64;
        uint32_t cnext[CW];
        const unsigned char *addr[M];  // LDS pointers (32-bit): smem + code*KSTRIDE + moff[t]
        lutv_t val[M];
        auto make_addr = [&](uint32_t (&cc)[CW]) {
            if constexpr (!SKEWED) rotate_row<CW>(cc, abit, bsh);  // SKEWED tables are stored pre-rotated
            static_for<0, M>([&](auto T) {
                constexpr int t = decltype(T)::value;
                const uint32_t code = __builtin_amdgcn_ubfe(cc[t / 4], 8 * (t % 4), 8);
                addr[t] = mbase[t] + code * (uint32_t)KSTRIDE;
            });
        };
        auto issue_chunk = [&](auto C, auto H) {
            constexpr int c8 = decltype(C)::value * 8;
            constexpr int hoff = decltype(H)::value * RB;
            static_for<0, 8>([&](auto I) {
                constexpr int t = c8 + decltype(I)::value;
                val[t] = *(const lutv_t *)(addr[t] + hoff);
            });
        };
        if (row0 < slice_end) {
            uint32_t c0[CW];
            load_row(row0 + lane, c0);
            load_row(row0 + stride + lane, cnext);
            make_addr(c0);
            static_for<0, NCH>([&](auto C) { issue_chunk(C, std::integral_constant<int, 0>{}); });
        }

        for (; row0 < slice_end; row0 += stride) {
            // rows this wave-step may return
            unsigned long long vmask = ~0ull;
            if (slice_end - row0 < 64) vmask = (1ull << (int)(slice_end - row0)) - 1ull;
            if (a.valid) {
                const uint32_t *vw = a.valid + (row0 >> 5);
                unsigned long long vb = (unsigned long long)vw[0];
                if (row0 + 32 < a.N) vb |= (unsigned long long)vw[1] << 32;
                vmask &= vb;
            }

            f32x2 acc[NQ][NP];
#pragma unroll
            for (int h = 0; h < NQ; ++h)
#pragma unroll
                for (int p = 0; p < NP; ++p) acc[h][p] = (f32x2){0.f, 0.f};

            static_for<0, NQ>([&](auto H) {
                constexpr int h = decltype(H)::value;
                // ordered accumulation: pass 1 (steps t >= t0) over all chunks ...
                static_for<0, NCH>([&](auto C) {
                    constexpr int t0 = decltype(C)::value * 8;
                    if constexpr (MODE == 2) {
                        static_for<0, 8>([&](auto I) {
                            constexpr int t = t0 + decltype(I)::value;
                            if constexpr (QI == 4) {
                                sfma(acc[h][0], ANNLITE_LOHALF(val[t]), wt[t].x);
                                sfma(acc[h][1], ANNLITE_HIHALF(val[t]), wt[t].x);
                            } else {
                                sfma(acc[h][0], val[t], wt[t].x);
                            }
                        });
                    } else if constexpr (MODE == 1) {
                        static_for<0, 8>([&](auto I) {
                            constexpr int t = t0 + decltype(I)::value;
                            if constexpr (QI == 4) {
                                wfma_p1(acc[h][0], ANNLITE_LOHALF(val[t]), wt[t]);
                                wfma_p1(acc[h][1], ANNLITE_HIHALF(val[t]), wt[t]);
                            } else {
                                wfma_p1(acc[h][0], val[t], wt[t]);
                            }
                        });
                    } else if constexpr (QI == 4) pass8_q4<M, t0, 0>(acc[h][0], acc[h][1], val);
                    else pass8_q2<M, t0, 0>(acc[h][0], val);
                });
                // ... then pass 2 (t < t0) chunk by chunk, re-filling each chunk as soon as it is consumed
                static_for<0, NCH>([&](auto C) {
                    constexpr int cidx = decltype(C)::value;
                    constexpr int t0 = cidx * 8;
                    if constexpr (MODE == 2) {
                        static_for<0, 8>([&](auto I) {
                            constexpr int t = t0 + decltype(I)::value;
                            if constexpr (QI == 4) {
                                sfma(acc[h][0], ANNLITE_LOHALF(val[t]), wt[t].y);
                                sfma(acc[h][1], ANNLITE_HIHALF(val[t]), wt[t].y);
                            } else {
                                sfma(acc[h][0], val[t], wt[t].y);
                            }
                        });
                    } else if constexpr (MODE == 1) {
                        static_for<0, 8>([&](auto I) {
                            constexpr int t = t0 + decltype(I)::value;
                            if constexpr (QI == 4) {
                                wfma_p2(acc[h][0], ANNLITE_LOHALF(val[t]), wt[t]);
                                wfma_p2(acc[h][1], ANNLITE_HIHALF(val[t]), wt[t]);
                            } else {
                                wfma_p2(acc[h][0], val[t], wt[t]);
                            }
                        });
                    } else if constexpr (QI == 4) pass8_q4<M, t0, 1>(acc[h][0], acc[h][1], val);
                    else pass8_q2<M, t0, 1>(acc[h][0], val);
                    if constexpr (h + 1 < NQ) {
                        issue_chunk(C, std::integral_constant<int, h + 1>{});
                    } else {
                        if constexpr (cidx == 0) {
                            uint32_t cc[CW];
#pragma unroll
                            for (int i = 0; i < CW; ++i) cc[i] = cnext[i];
                            make_addr(cc);                                 // addresses of the next row
                            load_row(row0 + 2 * stride + lane, cnext);    // global prefetch, two rows ahead
                        }
                        issue_chunk(C, std::integral_constant<int, 0>{});
                    }
                });
            });

            // offer rows that can still enter a list (rare after warm-up): one branch for all queries
            const uint32_t rid = (uint32_t)(row0 + lane);
            float dq[QT];
            unsigned long long pmq[QT], any = 0;
#pragma unroll
            for (int q = 0; q < QT; ++q) {
                dq[q] = acc[q / QI][(q % QI) / 2][q % 2];
                pmq[q] = __ballot(dq[q] <= thr_f[q]) & vmask;
                any |= pmq[q];
            }
            if (any) {
#pragma unroll
                for (int q = 0; q < QT; ++q) {
                    if (pmq[q]) {
                        wavelist_offer(list[q], pmq[q], f32_to_ordered(dq[q]), rid, km1, thr_hi[q], thr_lo[q], lane);
                        thr_f[q] = (thr_hi[q] == kKeyInfHi) ? __builtin_inff() : ordered_to_f32(thr_hi[q]);
                    }
                }
            }
        }

        // ---- merge the NW per-wave lists of each query through LDS (re-using the LUT space) -----
        __syncthreads();
        unsigned long long *scratch = (unsigned long long *)smem;  // [QT][NW][64]
#pragma unroll
        for (int q = 0; q < QT; ++q)
            scratch[(q * NW + wave) * 64 + lane] = ((unsigned long long)list[q].hi << 32) | list[q].lo;
        __syncthreads();
        for (int q = wave; q < QT; q += NW) {
            WaveList L;
            unsigned long long key = scratch[(q * NW + 0) * 64 + lane];
            L.hi = (uint32_t)(key >> 32);
            L.lo = (uint32_t)key;
            uint32_t th = __builtin_amdgcn_readlane(L.hi, km1), tl = __builtin_amdgcn_readlane(L.lo, km1);
            for (int w = 1; w < NW; ++w) {
                key = scratch[(q * NW + w) * 64 + lane];
                const uint32_t chi = (uint32_t)(key >> 32), clo = (uint32_t)key;
                const unsigned long long pm = __ballot(lane <= km1 && key_less(chi, clo, th, tl));
                wavelist_offer(L, pm, chi, clo, km1, th, tl, lane);
            }
            const int b = tile * QT + q;
            if (b < a.B && lane <= km1)
                a.partial[((int64_t)b * a.n_slices + slice) * a.k + lane] = ((unsigned long long)L.hi << 32) | L.lo;
        }
    }
}


// =================================================================================================
// Filter kernel (default): the VALU cost of the ordered two-pass sum (2 lane-masked adds per
// look-up) bounds adc_scan_fast_kernel, so this version
//   1. adds the M values of a row in the lane's ROTATED order -- ONE plain v_pk_add_f32 per two
//      look-ups, values consumed as they arrive.  |d_fast - d_exact| <= 2*gamma_{M-1} * sum_m|v_m|
//      <= slack[q] := 2*M*2^-24 * Smax[q] * (1+2^-10), Smax[q] = sum_m max_k |lut[q][m][k]|
//      (lut_smax_kernel), because both are fp32 summations of the same M terms;
//   2. FILTERS: a row can only be in the top-k if d_exact <= thr, hence d_fast <= thr + slack;
//   3. for the few rows that pass, recomputes the EXACT ascending-m sum from the still-held
//      values (the two-pass masked add of the fast kernel) and offers (ordered(d_exact), id);
//   4. shares the k-th key between the waves of the workgroup through LDS (atomic min), so all
//      waves filter with the tightest bound any of them has proven;
//   5. inserts floods (first step of a work item) with a bitonic sort + merge instead of one
//      by one.
// Returned distances and ids are bit-identical to the fast kernel / the oracle.
// LDS: [LUT tile Ks*KSTRIDE][shthr f32 x QT (thr+slack) @ +0][shkey u64 x QT @ +64]
// =================================================================================================
template <int M, int NQ, int NW, int WPS, bool SKEWED, bool DBUF>
__global__ __launch_bounds__(NW * 64, WPS) void adc_scan_filter_kernel(const ScanArgs a) {
    constexpr int QI = 4;
    constexpr int QT = QI * NQ;
    constexpr int CW = M / 4;
    constexpr int EB = QI * 4;
    constexpr int RB = M * EB;
    constexpr int KSTRIDE = NQ * RB;
    static_assert(M % 8 == 0 && M <= 32, "QI=4 instantiations only");

    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int km1 = a.k - 1;
    const int s = lane % M;
    const uint32_t bsh = (uint32_t)(s & 3);
    bool abit[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) abit[i] = (((s >> 2) >> i) & 1) != 0;
    const unsigned char *mbase[M];
#pragma unroll
    for (int t = 0; t < M; ++t) mbase[t] = smem + ((s + t) % M) * EB;

    const int lut_bytes = a.Ks * KSTRIDE;
    volatile float *shthr = (volatile float *)(smem + lut_bytes);
    unsigned long long *shkey = (unsigned long long *)(smem + lut_bytes + 64);

    const int n_items = a.n_items;
    const int64_t group_bytes = (int64_t)a.Ks * RB;

    for (int item = blockIdx.x; item < n_items; item += gridDim.x) {
        int tile, slice;
        if (!item_map(a, item, tile, slice)) continue;

        __syncthreads();
        {
            const unsigned char *src0 = (const unsigned char *)a.lut + (int64_t)tile * NQ * group_bytes;
            constexpr int PIECES_PER_ROW = RB / 16;
            const int total = NQ * a.Ks * PIECES_PER_ROW;
            for (int idx = tid; idx < total; idx += NW * 64) {
                const int p = idx % PIECES_PER_ROW;
                const int kh = idx / PIECES_PER_ROW;
                const int h = kh / a.Ks;
                const int kk = kh - h * a.Ks;
                const u32x4 v = *(const u32x4 *)(src0 + (int64_t)h * group_bytes + (int64_t)kk * RB + p * 16);
                *(u32x4 *)(smem + (kk * NQ + h) * RB + p * 16) = v;
            }
            if (tid < QT) {
                shthr[tid] = __builtin_inff();
                shkey[tid] = ~0ull;
            }
        }
        float slack[QT];
#pragma unroll
        for (int q = 0; q < QT; ++q)
            slack[q] = a.smax[tile * QT + q] * (float)(2.0 * M * 5.9604644775390625e-08 * (1.0 + 1.0 / 1024.0));
        __syncthreads();

        WaveList list[QT];
#pragma unroll
        for (int q = 0; q < QT; ++q) list[q].reset();

        const int64_t slice_begin = (int64_t)slice * a.slice_rows;
        int64_t slice_end = slice_begin + a.slice_rows;
        if (slice_end > a.N) slice_end = a.N;

        const uint32_t *codes32 = (const uint32_t *)a.codes;
        auto load_row = [&](int64_t row, uint32_t (&c)[CW]) {
            if (row >= a.N) row = a.N - 1;
            const uint32_t *p = codes32 + row * CW;
            if constexpr (CW == 2) {
                const u32x2 v = *(const u32x2 *)p;
                c[0] = v.x;
                c[1] = v.y;
            } else {
#pragma unroll
                for (int i = 0; i < CW / 4; ++i) {
                    const u32x4 v = *(const u32x4 *)(p + 4 * i);
                    c[4 * i + 0] = v.x;
                    c[4 * i + 1] = v.y;
                    c[4 * i + 2] = v.z;
                    c[4 * i + 3] = v.w;
                }
            }
        };

        const int64_t stride = (int64_t)NW * 64;
        int64_t row0 = slice_begin + (int64_t)wave * 64;
        uint32_t cnext[CW];
        const unsigned char *addr[M];
        // DBUF: one landing buffer per entry group -- the next row's look-ups of group h are issued as
        // soon as group h of the current row has been filtered (more look-ups in flight per wave,
        // ~64 more VGPRs).  !DBUF: one buffer, refilled with the NEXT group right after the filter
        // (fewer registers -> more waves per SIMD).
        constexpr int NB = DBUF ? NQ : 1;
        f32x4 val[NB][M];
        auto make_addr = [&](uint32_t (&cc)[CW]) {
            if constexpr (!SKEWED) rotate_row<CW>(cc, abit, bsh);
            static_for<0, M>([&](auto T) {
                constexpr int t = decltype(T)::value;
                static_assert((KSTRIDE & (KSTRIDE - 1)) == 0, "KSTRIDE must be a power of two");
                addr[t] = mbase[t] + byte_shl<t % 4>(cc[t / 4], (uint32_t)ilog2_c(KSTRIDE));
            });
        };
        auto issue_group = [&](auto H) {
            constexpr int h = decltype(H)::value;
            static_for<0, M>([&](auto T) {
                constexpr int t = decltype(T)::value;
                val[DBUF ? h : 0][t] = *(const f32x4 *)(addr[t] + h * RB);
            });
        };
        // workgroup bound (thr + slack) of each group's 4 queries, re-read every step.  LDS returns
        // in order, so the read is issued BEFORE the refill look-ups of the group and consumed one
        // step later -- reading it at the point of use would drain the whole look-up queue.
        f32x4 th[NQ];
#pragma unroll
        for (int h = 0; h < NQ; ++h) th[h] = *(const f32x4 *)(smem + lut_bytes + h * 16);
        if (row0 < slice_end) {
            uint32_t c0[CW];
            load_row(row0 + lane, c0);
            load_row(row0 + stride + lane, cnext);
            make_addr(c0);
            static_for<0, NB>([&](auto H) { issue_group(H); });
        }

        int step_no = 0;
        for (; row0 < slice_end; row0 += stride, ++step_no) {
            unsigned long long vmask = ~0ull;
            if (slice_end - row0 < 64) vmask = (1ull << (int)(slice_end - row0)) - 1ull;
            if (a.valid) {
                const uint32_t *vw = a.valid + (row0 >> 5);
                unsigned long long vb = (unsigned long long)vw[0];
                if (row0 + 32 < a.N) vb |= (unsigned long long)vw[1] << 32;
                vmask &= vb;
            }
            const uint32_t rid = (uint32_t)(row0 + lane);
            const bool refresh = (step_no & 3) == 0;  // other waves' bounds are picked up every 4th step

            static_for<0, NQ>([&](auto H) {
                constexpr int h = decltype(H)::value;
                constexpr int hb = DBUF ? h : 0;
                // 1. fast sum, rotated order
                f32x4 fs = val[hb][0];
                static_for<1, M>([&](auto T) { fs += val[hb][decltype(T)::value]; });
                // 2. filter
                unsigned long long pm[4], any = 0;
#pragma unroll
                for (int jq = 0; jq < 4; ++jq) {
                    pm[jq] = __ballot(fs[jq] <= th[h][jq]) & vmask;
                    any |= pm[jq];
                }
                if (any) {
                    // 3. exact ascending-m sums of the 4 queries from the held values
                    f32x2 e0 = {0.f, 0.f}, e1 = {0.f, 0.f};
                    static_for<0, 2>([&](auto P) {
                        static_for<0, M / 8>([&](auto C) {
                            pass8_q4<M, decltype(C)::value * 8, decltype(P)::value>(e0, e1, val[hb]);
                        });
                    });
                    const float ex[4] = {e0.x, e0.y, e1.x, e1.y};
#pragma unroll
                    for (int jq = 0; jq < 4; ++jq) {
                        if (pm[jq]) {
                            const int q = h * 4 + jq;
                            const uint32_t khi = f32_to_ordered(ex[jq]);
                            const unsigned long long sk = *(volatile unsigned long long *)(shkey + q);
                            const uint32_t skhi = (uint32_t)(sk >> 32), sklo = (uint32_t)sk;
                            const unsigned long long px = __ballot(key_less(khi, rid, skhi, sklo)) & pm[jq];
                            if (px) {
                                wavelist_insert_many(list[q], px, khi, rid, lane);
                                // 4. publish this wave's k-th key if it tightens the workgroup bound
                                const uint32_t ohi = __builtin_amdgcn_readlane(list[q].hi, km1);
                                const uint32_t olo = __builtin_amdgcn_readlane(list[q].lo, km1);
                                if (lane == 0 && ohi != kKeyInfHi) {
                                    const unsigned long long mine = ((unsigned long long)ohi << 32) | olo;
                                    const unsigned long long old = atomicMin(shkey + q, mine);
                                    if (mine < old) shthr[q] = ordered_to_f32(ohi) + slack[q];
                                }
                            }
                        }
                    }
                }
                // plain LDS read (ds_read_b128) behind a compiler barrier so it is re-issued every step; a
                // volatile access would be lowered to a FLAT load + vmcnt(0)/lgkmcnt(0) drains
                if (refresh) {
                    asm volatile("" ::: "memory");
                    th[h] = *(const f32x4 *)(smem + lut_bytes + h * 16);
                }
                if constexpr (DBUF) {
                    // refill this group's buffer with the next row's look-ups
                    if constexpr (h == 0) {
                        uint32_t cc[CW];
#pragma unroll
                        for (int i = 0; i < CW; ++i) cc[i] = cnext[i];
                        make_addr(cc);
                        load_row(row0 + 2 * stride + lane, cnext);
                    }
                    issue_group(H);
                } else if constexpr (h + 1 < NQ) {
                    issue_group(std::integral_constant<int, h + 1>{});  // next group of the same row
                } else {
                    uint32_t cc[CW];
#pragma unroll
                    for (int i = 0; i < CW; ++i) cc[i] = cnext[i];
                    make_addr(cc);
                    load_row(row0 + 2 * stride + lane, cnext);
                    issue_group(std::integral_constant<int, 0>{});  // first group of the next row
                }
            });
        }

        // ---- merge the NW per-wave lists of each query through LDS (re-using the LUT space) -----
        __syncthreads();
        unsigned long long *scratch = (unsigned long long *)smem;  // [QT][NW][64]
#pragma unroll
        for (int q = 0; q < QT; ++q)
            scratch[(q * NW + wave) * 64 + lane] = ((unsigned long long)list[q].hi << 32) | list[q].lo;
        __syncthreads();
        for (int q = wave; q < QT; q += NW) {
            WaveList L;
            unsigned long long key = scratch[(q * NW + 0) * 64 + lane];
            L.hi = (uint32_t)(key >> 32);
            L.lo = (uint32_t)key;
            for (int w = 1; w < NW; ++w) {
                key = scratch[(q * NW + w) * 64 + lane];
                // every wave list is ascending over the lanes: sorted merge, keep the 64 smallest
                wavelist_merge_sorted(L, (uint32_t)(key >> 32), (uint32_t)key, lane);
            }
            const int b = tile * QT + q;
            if (b < a.B && lane <= km1)
                a.partial[((int64_t)b * a.n_slices + slice) * a.k + lane] = ((unsigned long long)L.hi << 32) | L.lo;
        }
    }
}



}  // namespace annlite

using namespace annlite;

template <int M, int NQ, int NW, int WPS, bool SKEWED, bool DBUF>
static int launch_filter(const ScanArgs &a, int grid, hipStream_t st) {
    const size_t lds_lut = (size_t)a.Ks * NQ * M * 16;
    size_t need = lds_lut + 128;
    const size_t scratch = (size_t)4 * NQ * NW * 64 * 8;
    if (need < scratch) need = scratch;
    auto fn = adc_scan_filter_kernel<M, NQ, NW, WPS, SKEWED, DBUF>;
    ANNLITE_HIP_TRY(hipFuncSetAttribute((const void *)fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)need));
    hipLaunchKernelGGL(fn, dim3(grid), dim3(NW * 64), need, st, a);
    return launch_status("adc_scan_filter_kernel");
}

template <int M, int QI, int NQ, int NW, int WPS, bool SKEWED, int MODE>
static int launch_fast(const ScanArgs &a, int grid, hipStream_t st) {
    const size_t lds = (size_t)a.Ks * NQ * M * QI * 4;
    size_t need = lds;
    const size_t scratch = (size_t)QI * NQ * NW * 64 * 8;
    if (need < scratch) need = scratch;
    auto fn = adc_scan_fast_kernel<M, QI, NQ, NW, WPS, SKEWED, MODE>;
    ANNLITE_HIP_TRY(hipFuncSetAttribute((const void *)fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)need));
    hipLaunchKernelGGL(fn, dim3(grid), dim3(NW * 64), need, st, a);
    return launch_status("adc_scan_fast_kernel");
}

int annlite::launch_legacy_scan(int id, bool sk, const ScanArgs &a, int grid, hipStream_t st) {
#define ANNLITE_LAUNCH_F(MM, NQ_, NW_, WPS_, DB_) \
    (sk ? launch_filter<MM, NQ_, NW_, WPS_, true, DB_>(a, grid, st) : launch_filter<MM, NQ_, NW_, WPS_, false, DB_>(a, grid, st))
#define ANNLITE_LAUNCH_M(MM, QI_, NQ_, NW_, WPS_, MODE_) \
    (sk ? launch_fast<MM, QI_, NQ_, NW_, WPS_, true, MODE_>(a, grid, st) : launch_fast<MM, QI_, NQ_, NW_, WPS_, false, MODE_>(a, grid, st))
#define ANNLITE_LAUNCH(MM, QI_, NQ_, NW_, WPS_) ANNLITE_LAUNCH_M(MM, QI_, NQ_, NW_, WPS_, 0)
    if (id == 640 && sk) {
        set_error("the two-pass M=64 kernel reads PLAIN tables only (SKEWED M=64 tables are wrap-coded for the default kernel)");
        return ANNLITE_ERR_UNSUPPORTED;
    }
    switch (id) {
        case 81: return ANNLITE_LAUNCH_F(8, 2, 8, 2, true);
        case 1600: return ANNLITE_LAUNCH_F(16, 2, 8, 2, true);
        case 1601: return ANNLITE_LAUNCH_F(16, 2, 12, 3, false);
        case 1602: return ANNLITE_LAUNCH_F(16, 1, 8, 4, true);
        case 1603: return ANNLITE_LAUNCH_F(16, 2, 16, 4, false);
        case 321: return ANNLITE_LAUNCH_F(32, 1, 8, 2, true);
        case 80: return ANNLITE_LAUNCH(8, 4, 2, 8, 2);
        case 160: return ANNLITE_LAUNCH(16, 4, 2, 8, 2);
        case 161: return ANNLITE_LAUNCH(16, 4, 1, 8, 4);
        case 162: return ANNLITE_LAUNCH(16, 4, 2, 16, 4);
        case 163: return ANNLITE_LAUNCH(16, 4, 2, 12, 3);
        case 164: return ANNLITE_LAUNCH_M(16, 4, 2, 8, 2, 1);
        case 165: return ANNLITE_LAUNCH_M(16, 4, 2, 12, 3, 1);
        case 166: return ANNLITE_LAUNCH_M(16, 4, 1, 8, 4, 1);
        case 167: return ANNLITE_LAUNCH_M(16, 4, 2, 8, 2, 2);
        case 320: return ANNLITE_LAUNCH(32, 4, 1, 8, 2);
        case 640: return ANNLITE_LAUNCH(64, 2, 1, 8, 2);
        default: set_error("no legacy scan kernel with id %d", id); return ANNLITE_ERR_UNSUPPORTED;
    }
#undef ANNLITE_LAUNCH
#undef ANNLITE_LAUNCH_M
#undef ANNLITE_LAUNCH_F
}
