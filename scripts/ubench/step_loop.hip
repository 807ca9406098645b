// The byte-table scan kernel's step loop in isolation (one workgroup per CU, 16 waves, 128 KB table in LDS, conflict-free
// 16-byte look-ups, 4 dword adds per look-up): how close does the instruction mix get to the LDS rate of 4 cycles per
// ds_read_b128, and which part of the step costs what.  Knobs are macros; results go into DESIGN.md section 3.1.
//   hipcc --offload-arch=gfx950 -O3 -DADDR=1 -DFILTER=1 -DDEPTH=8 scripts/ubench/step_loop.hip -o step_loop && ./step_loop
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>

#ifndef ADDR
#define ADDR 1  // 0: loop-invariant addresses, 1: SDWA byte shift + add per look-up address (what the kernel does), 2: SDWA word shift only (prescaled)
                // 3 (round 6): ONE v_perm_b32 per PAIR of look-ups -- two half tables by sub-space parity, 64 KB apart, both entry groups of a
                //    (code, sub-space) adjacent: address = (parity << 16) | (code << 8) | (slot << 4), slot = 2 (m >> 1) + group + parity (the odd
                //    half table is shifted by one slot -- its last entry spills into the next code's row, 257 rows -- so that the 16 lanes of a
                //    read hit 16 different slots); the second group is the first + 16 (immediate)
#endif
#ifndef FILTER
#define FILTER 1  // 0: none, 1: the kernel's (and, sub, bitop3 per dword), 2: folded into the sums (and-reduce)
#endif
#ifndef DEPTH
#define DEPTH 8
#endif
#ifndef READS
#define READS 1  // 0: no LDS reads (VALU only)
#endif
#ifndef ADDS
#define ADDS 1  // 0: no adds (LDS only)
#endif
#ifndef NWAVES
#define NWAVES 16
#endif
#ifndef DYN
#define DYN 0  // 1: the waves draw their steps from a counter in LDS (the hardware favours the oldest wave of a SIMD: with a
               // static split the favoured waves finish early and the last ones run alone, at a third of the issue rate)
#endif

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef const u32x4 __attribute__((address_space(3))) *lds_entry_ptr;

template <int B, int E, typename F>
__device__ __forceinline__ void static_for(F &&f) {
    if constexpr (B < E) {
        f(std::integral_constant<int, B>{});
        static_for<B + 1, E>(f);
    }
}

__device__ __forceinline__ void byte_shl4(uint32_t x, uint32_t sh, uint32_t &o0, uint32_t &o1, uint32_t &o2, uint32_t &o3) {
    asm("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_0" : "=v"(o0) : "s"(sh), "v"(x));
    asm("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1" : "=v"(o1) : "s"(sh), "v"(x));
    asm("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_2" : "=v"(o2) : "s"(sh), "v"(x));
    asm("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_3" : "=v"(o3) : "s"(sh), "v"(x));
}
__device__ __forceinline__ void word_shl2(uint32_t x, uint32_t sh, uint32_t &o0, uint32_t &o1) {
    asm("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_0" : "=v"(o0) : "s"(sh), "v"(x));
    asm("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_1" : "=v"(o1) : "s"(sh), "v"(x));
}

__global__ __launch_bounds__(NWAVES * 64, NWAVES / 4) void step_loop(unsigned long long *out, const uint32_t *codes, int steps, uint32_t sh) {
    constexpr int M = 16, NQ = 2, RB = ADDR == 3 ? 16 : 256, TOT = NQ * M;  // (RB: distance of a look-up's second entry group)
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < 128 * 1024 / 4; i += NWAVES * 64) ((uint32_t *)smem)[i] = (uint32_t)i * 0x01010101u & 0x03030303u;
    __syncthreads();
    const int s = lane % M;
    uint32_t mbase[M];
#pragma unroll
    for (int t = 0; t < M; ++t) mbase[t] = (uint32_t)(((s + t) % M) * 16);
    uint32_t addr[M];
#if ADDR == 3
    uint32_t kperm[M];  // byte 0: slot << 4 of the step's FIRST read, byte 2: the half table (sub-space parity)
#pragma unroll
    for (int t = 0; t < M; ++t) {
        const uint32_t m = (uint32_t)((s + t) % M), par = m & 1u;
        kperm[t] = ((2u * (m >> 1) + par) << 4) | (par << 16);
    }
#endif
    u32x4 thp[NQ];
#pragma unroll
    for (int h = 0; h < NQ; ++h) thp[h] = (u32x4){0x80808080u | (uint32_t)lane, 0x81818181u, 0x82828282u, 0x83838383u};
    // the wave's code rows: 16 bytes per lane per step, from a small L2-resident array
    const u32x4 *crow = (const u32x4 *)codes + (size_t)(blockIdx.x * NWAVES + wave) * 64 + lane;
    u32x4 cnext = crow[0];
    uint32_t found = 0;
#if ADDR == 0
#pragma unroll
    for (int t = 0; t < M; ++t) addr[t] = mbase[t] + (((cnext.x >> t) & 0xffu) << 9);
#endif
    uint32_t *ctr = (uint32_t *)(smem + 128 * 1024);
    if (tid == 0) *ctr = 0;
    __syncthreads();
    const unsigned long long t0 = __builtin_readcyclecounter();
#if DYN
    const int total = steps * NWAVES;
    int nxt = 0;
    if (lane == 0) nxt = (int)atomicAdd(ctr, 1u);
    nxt = __builtin_amdgcn_readfirstlane(nxt);
    for (int st = nxt; st < total; st = nxt) {
        if (lane == 0) nxt = (int)atomicAdd(ctr, 1u);  // the next step's number: issued here, picked up at the end of this step
        const u32x4 ccur = cnext;
        cnext = crow[(size_t)((st + 1) & 63) * (gridDim.x * NWAVES * 64)];
#else
    for (int st = 0; st < steps; ++st) {
        const u32x4 ccur = cnext;
        cnext = crow[(size_t)((st + 1) & 63) * (gridDim.x * NWAVES * 64)];
#endif
#if ADDR == 1
        {
            const uint32_t cc[4] = {ccur.x, ccur.y, ccur.z, ccur.w};
            static_for<0, 4>([&](auto W) {
                constexpr int w = decltype(W)::value;
                uint32_t o0, o1, o2, o3;
                byte_shl4(cc[w], sh, o0, o1, o2, o3);
                addr[4 * w + 0] = mbase[4 * w + 0] + o0;
                addr[4 * w + 1] = mbase[4 * w + 1] + o1;
                addr[4 * w + 2] = mbase[4 * w + 2] + o2;
                addr[4 * w + 3] = mbase[4 * w + 3] + o3;
            });
        }
#elif ADDR == 3
        {
            const uint32_t cc[4] = {ccur.x, ccur.y, ccur.z, ccur.w};
            static_for<0, M>([&](auto T) {
                constexpr int t = decltype(T)::value;
                // byte 0 <- kperm byte 0, byte 1 <- code byte t % 4, byte 2 <- kperm byte 2, byte 3 <- 0
                addr[t] = __builtin_amdgcn_perm(cc[t / 4], kperm[t], 0x0c020000u | ((4u + (uint32_t)(t % 4)) << 8));
            });
        }
#elif ADDR == 2
        {
            const uint32_t cc[4] = {ccur.x, ccur.y, ccur.z, ccur.w};
            static_for<0, 4>([&](auto W) {  // (8 of the 16 addresses per dword pair: the prescaled table has 32 bytes per row)
                constexpr int w = decltype(W)::value;
                uint32_t o0, o1;
                word_shl2(cc[w] & 0x1fff1fffu, 4u, o0, o1);
                addr[4 * w + 0] = o0, addr[4 * w + 1] = o1;
                word_shl2((cc[w] >> 3) & 0x1fff1fffu, 4u, o0, o1);
                addr[4 * w + 2] = o0, addr[4 * w + 3] = o1;
            });
        }
#endif
        u32x4 acc[NQ];
        u32x4 v[DEPTH];
        auto fetch = [&](u32x4 &dst, uint32_t ad) {
            if constexpr (READS == 0) asm volatile("" : "=v"(dst) : "v"(ad));
            else dst = *(lds_entry_ptr)(uintptr_t)ad;
        };
        static_for<0, DEPTH>([&](auto I) {
            constexpr int i = decltype(I)::value;
            fetch(v[i], addr[i % M] + (uint32_t)((i / M) * RB));
        });
        static_for<0, TOT>([&](auto I) {
            constexpr int i = decltype(I)::value;
            asm volatile("" ::: "memory");
            if constexpr (ADDS == 0) {
                asm volatile("" ::"v"(v[i % DEPTH]));
                if constexpr (i % M == 0) acc[i / M] = thp[i / M];
            } else {
#if FILTER == 2
                if constexpr (i % M == 0) acc[i / M] = thp[i / M] + v[i % DEPTH];
#else
                if constexpr (i % M == 0) acc[i / M] = v[i % DEPTH];
#endif
                else acc[i / M] += v[i % DEPTH];
            }
            if constexpr (i + DEPTH < TOT) {
                constexpr int j = i + DEPTH;
                fetch(v[i % DEPTH], addr[j % M] + (uint32_t)((j / M) * RB));
            }
        });
#if FILTER == 1
        uint32_t anyv = 0;
#pragma unroll
        for (int h = 0; h < NQ; ++h)
#pragma unroll
            for (int w = 0; w < 4; ++w) anyv |= (thp[h][w] - (acc[h][w] & 0x7f7f7f7fu)) & ~acc[h][w];
        if (__ballot((anyv & 0x80808080u) != 0)) found += anyv;
#elif FILTER == 2
        uint32_t allv = ~0u;
#pragma unroll
        for (int h = 0; h < NQ; ++h)
#pragma unroll
            for (int w = 0; w < 4; ++w) allv &= acc[h][w];
        if (__ballot((~allv & 0x80808080u) != 0)) found += allv;
#else
        found += acc[0][0] + acc[1][3];
#endif
#if DYN
        nxt = __builtin_amdgcn_readfirstlane(nxt);
#endif
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    if (lane == 0) out[blockIdx.x * NWAVES + wave] = t1 - t0;
    if (found == 0x12345u) out[0] = found;
}

int main(int argc, char **argv) {
    const int steps = argc > 1 ? atoi(argv[1]) : 2000;
    const int grid = argc > 2 ? atoi(argv[2]) : 256;
    unsigned long long *d_out;
    uint32_t *d_codes;
    const size_t n_code_words = (size_t)64 * grid * NWAVES * 64 * 4;
    if (hipMalloc(&d_out, grid * NWAVES * 8) != hipSuccess || hipMalloc(&d_codes, n_code_words * 4) != hipSuccess) return 1;
    uint32_t *h = (uint32_t *)malloc(n_code_words * 4);
    uint32_t x = 12345;
    for (size_t i = 0; i < n_code_words; ++i) {
        x = x * 1664525u + 1013904223u;
        h[i] = ADDR == 2 ? (x >> 3) : x;
    }
    if (hipMemcpy(d_codes, h, n_code_words * 4, hipMemcpyHostToDevice) != hipSuccess) return 1;
    const size_t lds = 128 * 1024 + 4096;
    if (hipFuncSetAttribute((const void *)step_loop, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) return 1;
    for (int rep = 0; rep < 3; ++rep) {
        hipLaunchKernelGGL(step_loop, dim3(grid), dim3(NWAVES * 64), lds, 0, d_out, d_codes, steps, 9u);
        if (hipDeviceSynchronize() != hipSuccess) { printf("launch failed: %s\n", hipGetErrorString(hipGetLastError())); return 1; }
    }
    unsigned long long *ho = (unsigned long long *)malloc(grid * NWAVES * 8);
    if (hipMemcpy(ho, d_out, grid * NWAVES * 8, hipMemcpyDeviceToHost) != hipSuccess) return 1;
    unsigned long long mx = 0;
    double sum = 0;
    for (int i = 0; i < grid * NWAVES; ++i) { mx = ho[i] > mx ? ho[i] : mx; sum += (double)ho[i]; }
    const double per_step = (double)mx / steps;                    // cycles per step of one wave (16 waves run concurrently)
    const double reads = 32.0 * NWAVES;                            // ds_read_b128 per CU per step round
    printf("DYN=%d ADDR=%d FILTER=%d DEPTH=%d READS=%d ADDS=%d NWAVES=%d: %.0f cycles per step round (max wave; mean %.0f) -> %.2f cycles per look-up per CU (LDS roof 4.00), frac %.3f\n",
           DYN, ADDR, FILTER, DEPTH, READS, ADDS, NWAVES, per_step, sum / (grid * NWAVES) / steps, per_step / reads, 4.0 * reads / per_step);
    return 0;
}
