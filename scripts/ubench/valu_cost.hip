// Cycles per wave64 VALU instruction on gfx950, as the scan kernels use them: one workgroup on one CU, W waves per SIMD,
// 4 independent dependency chains per wave (what the byte-sum loop has: the 4 dwords of an entry group) or 1 chain.
//   hipcc --offload-arch=gfx950 -O2 scripts/ubench/valu_cost.hip -o /tmp/valu_cost && /tmp/valu_cost
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

#define REP4(x) x x x x
#define REP16(x) REP4(x) REP4(x) REP4(x) REP4(x)
#define REP64(x) REP16(x) REP16(x) REP16(x) REP16(x)

// 4 chains: a0..a3 each updated in turn; b, c are loop-invariant operands
#define KERNEL(NAME, BODY)                                                                                   \
    __global__ void NAME(unsigned long long *out, uint32_t seed, int iters) {                                    \
        uint32_t a0 = threadIdx.x + seed, a1 = a0 * 3u, a2 = a0 * 5u, a3 = a0 * 7u, b = seed | 1u, c = seed * 9u;  \
        uint32_t d0 = a0 ^ 11u, d1 = a1 ^ 13u, d2 = a2 ^ 17u, d3 = a3 ^ 19u;                                       \
        unsigned long long t0 = __builtin_readcyclecounter();                                                   \
        for (int i = 0; i < iters; ++i) {                                                                      \
            asm volatile(REP16(BODY) : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3) : "v"(b), "v"(c), "s"(seed)); \
        }                                                                                                      \
        unsigned long long t1 = __builtin_readcyclecounter();                                                   \
        if ((threadIdx.x & 63) == 0) out[threadIdx.x >> 6] = t1 - t0;                                           \
        if (a0 + a1 + a2 + a3 + d0 + d1 + d2 + d3 == 0x12345u) out[63] = a0;                                    \
    }

// each BODY = 4 instructions (one per chain); REP16 -> 64 instructions per loop trip
KERNEL(k_add_vop2, "v_add_u32 %0, %0, %8\n v_add_u32 %1, %1, %8\n v_add_u32 %2, %2, %8\n v_add_u32 %3, %3, %8\n")
KERNEL(k_add_vop2_1chain, "v_add_u32 %0, %0, %8\n v_add_u32 %0, %0, %8\n v_add_u32 %0, %0, %8\n v_add_u32 %0, %0, %8\n")
KERNEL(k_add3, "v_add3_u32 %0, %0, %8, %9\n v_add3_u32 %1, %1, %8, %9\n v_add3_u32 %2, %2, %8, %9\n v_add3_u32 %3, %3, %8, %9\n")
KERNEL(k_add3_1chain, "v_add3_u32 %0, %0, %8, %9\n v_add3_u32 %0, %0, %8, %9\n v_add3_u32 %0, %0, %8, %9\n v_add3_u32 %0, %0, %8, %9\n")
KERNEL(k_add3_3v, "v_add3_u32 %0, %0, %4, %5\n v_add3_u32 %1, %1, %5, %6\n v_add3_u32 %2, %2, %6, %7\n v_add3_u32 %3, %3, %7, %4\n")
KERNEL(k_sdwa_shl, "v_lshlrev_b32_sdwa %0, %10, %4 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_0\n v_lshlrev_b32_sdwa %1, %10, %4 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1\n v_lshlrev_b32_sdwa %2, %10, %4 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_2\n v_lshlrev_b32_sdwa %3, %10, %4 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_3\n")
KERNEL(k_sdwa_add, "v_add_u32_sdwa %0, %8, %4 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_0\n v_add_u32_sdwa %1, %8, %4 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1\n v_add_u32_sdwa %2, %8, %4 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_2\n v_add_u32_sdwa %3, %8, %4 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_3\n")
KERNEL(k_and_lit, "v_and_b32 %0, 0x7f7f7f7f, %0\n v_and_b32 %1, 0x7f7f7f7f, %1\n v_and_b32 %2, 0x7f7f7f7f, %2\n v_and_b32 %3, 0x7f7f7f7f, %3\n")
KERNEL(k_and_reg, "v_and_b32 %0, %8, %0\n v_and_b32 %1, %8, %1\n v_and_b32 %2, %8, %2\n v_and_b32 %3, %8, %3\n")
KERNEL(k_bitop3, "v_bitop3_b32 %0, %0, %8, %9 bitop3:0xdc\n v_bitop3_b32 %1, %1, %8, %9 bitop3:0xdc\n v_bitop3_b32 %2, %2, %8, %9 bitop3:0xdc\n v_bitop3_b32 %3, %3, %8, %9 bitop3:0xdc\n")
KERNEL(k_perm, "v_perm_b32 %0, %0, %8, %9\n v_perm_b32 %1, %1, %8, %9\n v_perm_b32 %2, %2, %8, %9\n v_perm_b32 %3, %3, %8, %9\n")
KERNEL(k_and_or, "v_and_or_b32 %0, %0, %8, %9\n v_and_or_b32 %1, %1, %8, %9\n v_and_or_b32 %2, %2, %8, %9\n v_and_or_b32 %3, %3, %8, %9\n")
KERNEL(k_lshl_add, "v_lshl_add_u32 %0, %0, 1, %9\n v_lshl_add_u32 %1, %1, 1, %9\n v_lshl_add_u32 %2, %2, 1, %9\n v_lshl_add_u32 %3, %3, 1, %9\n")
KERNEL(k_lshl_or, "v_lshl_or_b32 %0, %0, 1, %9\n v_lshl_or_b32 %1, %1, 1, %9\n v_lshl_or_b32 %2, %2, 1, %9\n v_lshl_or_b32 %3, %3, 1, %9\n")
KERNEL(k_pk_add_u16, "v_pk_add_u16 %0, %0, %8\n v_pk_add_u16 %1, %1, %8\n v_pk_add_u16 %2, %2, %8\n v_pk_add_u16 %3, %3, %8\n")
KERNEL(k_mad_u24, "v_mad_u32_u24 %0, %0, %8, %9\n v_mad_u32_u24 %1, %1, %8, %9\n v_mad_u32_u24 %2, %2, %8, %9\n v_mad_u32_u24 %3, %3, %8, %9\n")
KERNEL(k_mov, "v_mov_b32 %0, %4\n v_mov_b32 %1, %5\n v_mov_b32 %2, %6\n v_mov_b32 %3, %7\n")
KERNEL(k_sub, "v_sub_u32 %0, %8, %0\n v_sub_u32 %1, %8, %1\n v_sub_u32 %2, %8, %2\n v_sub_u32 %3, %8, %3\n")
KERNEL(k_bfe, "v_bfe_u32 %0, %0, 8, 8\n v_bfe_u32 %1, %1, 8, 8\n v_bfe_u32 %2, %2, 8, 8\n v_bfe_u32 %3, %3, 8, 8\n")
KERNEL(k_cvt_ubyte, "v_cvt_f32_ubyte1 %0, %4\n v_cvt_f32_ubyte2 %1, %5\n v_cvt_f32_ubyte3 %2, %6\n v_cvt_f32_ubyte0 %3, %7\n")
KERNEL(k_add_f32, "v_add_f32 %0, %0, %8\n v_add_f32 %1, %1, %8\n v_add_f32 %2, %2, %8\n v_add_f32 %3, %3, %8\n")
KERNEL(k_alignbit, "v_alignbit_b32 %0, %0, %8, 8\n v_alignbit_b32 %1, %1, %8, 8\n v_alignbit_b32 %2, %2, %8, 8\n v_alignbit_b32 %3, %3, %8, 8\n")
KERNEL(k_xad, "v_xad_u32 %0, %0, %8, %9\n v_xad_u32 %1, %1, %8, %9\n v_xad_u32 %2, %2, %8, %9\n v_xad_u32 %3, %3, %8, %9\n")
KERNEL(k_add_dpp, "v_add_u32_dpp %0, %0, %4 row_shr:1 row_mask:0xf bank_mask:0xf\n v_add_u32_dpp %1, %1, %5 row_shr:1 row_mask:0xf bank_mask:0xf\n v_add_u32_dpp %2, %2, %6 row_shr:1 row_mask:0xf bank_mask:0xf\n v_add_u32_dpp %3, %3, %7 row_shr:1 row_mask:0xf bank_mask:0xf\n")

// 64-bit / packed-fp32 forms: register pairs
#define KERNEL64(NAME, BODY)                                                                                 \
    __global__ void NAME(unsigned long long *out, uint32_t seed, int iters) {                                    \
        unsigned long long a0 = threadIdx.x + seed, a1 = a0 * 3u, a2 = a0 * 5u, a3 = a0 * 7u, b = seed | 1u;        \
        unsigned long long t0 = __builtin_readcyclecounter();                                                   \
        for (int i = 0; i < iters; ++i) {                                                                      \
            asm volatile(REP16(BODY) : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b));                         \
        }                                                                                                      \
        unsigned long long t1 = __builtin_readcyclecounter();                                                   \
        if ((threadIdx.x & 63) == 0) out[threadIdx.x >> 6] = t1 - t0;                                           \
        if (a0 + a1 + a2 + a3 == 0x12345u) out[63] = a0;                                                        \
    }
KERNEL64(k_lshl_add_u64, "v_lshl_add_u64 %0, %0, 0, %4\n v_lshl_add_u64 %1, %1, 0, %4\n v_lshl_add_u64 %2, %2, 0, %4\n v_lshl_add_u64 %3, %3, 0, %4\n")
KERNEL64(k_pk_add_f32, "v_pk_add_f32 %0, %0, %4\n v_pk_add_f32 %1, %1, %4\n v_pk_add_f32 %2, %2, %4\n v_pk_add_f32 %3, %3, %4\n")
KERNEL64(k_pk_mov, "v_pk_mov_b32 %0, %4, %4\n v_pk_mov_b32 %1, %4, %4\n v_pk_mov_b32 %2, %4, %4\n v_pk_mov_b32 %3, %4, %4\n")
KERNEL64(k_mov_b64, "v_mov_b64 %0, %4\n v_mov_b64 %1, %4\n v_mov_b64 %2, %4\n v_mov_b64 %3, %4\n")

typedef void (*kern_t)(unsigned long long *, uint32_t, int);
struct Entry { const char *name; kern_t fn; };

int main() {
    Entry tab[] = {
        {"v_add_u32 (VOP2), 4 chains", k_add_vop2}, {"v_add_u32 (VOP2), 1 chain", k_add_vop2_1chain},
        {"v_add3_u32, 4 chains, 1 vgpr + 2 invariant", k_add3}, {"v_add3_u32, 1 chain", k_add3_1chain},
        {"v_add3_u32, 3 distinct vgprs", k_add3_3v},
        {"v_lshlrev_b32_sdwa (byte sel)", k_sdwa_shl}, {"v_add_u32_sdwa (byte sel)", k_sdwa_add},
        {"v_and_b32 literal", k_and_lit}, {"v_and_b32 reg", k_and_reg}, {"v_bitop3_b32", k_bitop3}, {"v_perm_b32", k_perm},
        {"v_and_or_b32", k_and_or}, {"v_lshl_add_u32", k_lshl_add}, {"v_lshl_or_b32", k_lshl_or}, {"v_pk_add_u16", k_pk_add_u16},
        {"v_mad_u32_u24", k_mad_u24}, {"v_mov_b32", k_mov}, {"v_sub_u32", k_sub}, {"v_bfe_u32", k_bfe},
        {"v_cvt_f32_ubyteN", k_cvt_ubyte}, {"v_add_f32", k_add_f32}, {"v_alignbit_b32", k_alignbit}, {"v_xad_u32", k_xad},
        {"v_add_u32_dpp row_shr", k_add_dpp},
        {"v_lshl_add_u64", k_lshl_add_u64}, {"v_pk_add_f32", k_pk_add_f32}, {"v_pk_mov_b32", k_pk_mov}, {"v_mov_b64", k_mov_b64},
    };
    unsigned long long *d_out, h_out[64];
    if (hipMalloc(&d_out, sizeof(h_out)) != hipSuccess) return 1;
    const int iters = 2000;
    printf("%-46s %10s %10s %10s\n", "instruction", "1 wave/SIMD", "2 waves", "4 waves");
    for (auto &e : tab) {
        printf("%-46s", e.name);
        for (int waves : {4, 8, 16}) {  // waves per workgroup = per CU (one workgroup): 1, 2, 4 per SIMD
            hipLaunchKernelGGL(e.fn, dim3(1), dim3(waves * 64), 0, 0, d_out, 12345u, 10);  // warm
            hipLaunchKernelGGL(e.fn, dim3(1), dim3(waves * 64), 0, 0, d_out, 12345u, iters);
            if (hipMemcpy(h_out, d_out, sizeof(h_out), hipMemcpyDeviceToHost) != hipSuccess) return 1;
            unsigned long long mx = 0;
            for (int w = 0; w < waves; ++w) mx = h_out[w] > mx ? h_out[w] : mx;
            // (s_memtime counts shader clocks on gfx950) -> cycles per instruction PER SIMD
            const double cyc = (double)mx;
            const double per_simd_insts = (double)iters * 64.0 * (waves / 4);
            printf(" %10.2f", cyc / per_simd_insts);
        }
        printf("\n");
    }
    return 0;
}
