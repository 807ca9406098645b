// VALU issue-cost microbenchmark (gfx950): cycles per wave64 instruction per SIMD for v_add_f32,
// v_pk_add_f32, v_fma_f32, v_pk_fma_f32 with 8 independent accumulators, at 1/2/4 waves per SIMD.
// hipcc --offload-arch=gfx950 -O3 scripts/valu_ubench.hip -o gpurun_out/valu_ubench && gpurun_out/valu_ubench
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x2 __attribute__((ext_vector_type(2)));
#define REP8(X) X X X X X X X X
template <int KIND>
__global__ void k(float *out, int iters, float seed) {
    f32x2 a0 = {seed, 1}, a1 = {seed, 2}, a2 = {seed, 3}, a3 = {seed, 4}, a4 = {seed, 5}, a5 = {seed, 6}, a6 = {seed, 7}, a7 = {seed, 8};
    f32x2 x = {seed * 0.5f, seed * 0.25f}, w = {1.f, 1.f};
    for (int i = 0; i < iters; ++i) {
        if (KIND == 0) {
            REP8(asm volatile("v_add_f32 %0, %0, %8\n v_add_f32 %1, %1, %8\n v_add_f32 %2, %2, %8\n v_add_f32 %3, %3, %8\n"
                              "v_add_f32 %4, %4, %8\n v_add_f32 %5, %5, %8\n v_add_f32 %6, %6, %8\n v_add_f32 %7, %7, %8"
                              : "+v"(a0.x), "+v"(a1.x), "+v"(a2.x), "+v"(a3.x), "+v"(a4.x), "+v"(a5.x), "+v"(a6.x), "+v"(a7.x) : "v"(x.x));)
        } else if (KIND == 1) {
            REP8(asm volatile("v_pk_add_f32 %0, %0, %8\n v_pk_add_f32 %1, %1, %8\n v_pk_add_f32 %2, %2, %8\n v_pk_add_f32 %3, %3, %8\n"
                              "v_pk_add_f32 %4, %4, %8\n v_pk_add_f32 %5, %5, %8\n v_pk_add_f32 %6, %6, %8\n v_pk_add_f32 %7, %7, %8"
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(x));)
        } else if (KIND == 2) {
            REP8(asm volatile("v_fma_f32 %0, %8, %9, %0\n v_fma_f32 %1, %8, %9, %1\n v_fma_f32 %2, %8, %9, %2\n v_fma_f32 %3, %8, %9, %3\n"
                              "v_fma_f32 %4, %8, %9, %4\n v_fma_f32 %5, %8, %9, %5\n v_fma_f32 %6, %8, %9, %6\n v_fma_f32 %7, %8, %9, %7"
                              : "+v"(a0.x), "+v"(a1.x), "+v"(a2.x), "+v"(a3.x), "+v"(a4.x), "+v"(a5.x), "+v"(a6.x), "+v"(a7.x) : "v"(x.x), "v"(w.x));)
        } else if (KIND == 3) {
            REP8(asm volatile("v_pk_fma_f32 %0, %8, %9, %0\n v_pk_fma_f32 %1, %8, %9, %1\n v_pk_fma_f32 %2, %8, %9, %2\n v_pk_fma_f32 %3, %8, %9, %3\n"
                              "v_pk_fma_f32 %4, %8, %9, %4\n v_pk_fma_f32 %5, %8, %9, %5\n v_pk_fma_f32 %6, %8, %9, %6\n v_pk_fma_f32 %7, %8, %9, %7"
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(x), "v"(w));)
        } else if (KIND == 5) {
            REP8(asm volatile("v_pk_add_u16 %0, %0, %8\n v_pk_add_u16 %1, %1, %8\n v_pk_add_u16 %2, %2, %8\n v_pk_add_u16 %3, %3, %8\n"
                              "v_pk_add_u16 %4, %4, %8\n v_pk_add_u16 %5, %5, %8\n v_pk_add_u16 %6, %6, %8\n v_pk_add_u16 %7, %7, %8"
                              : "+v"(a0.x), "+v"(a1.x), "+v"(a2.x), "+v"(a3.x), "+v"(a4.x), "+v"(a5.x), "+v"(a6.x), "+v"(a7.x) : "v"(x.x));)
        } else if (KIND == 6) {
            REP8(asm volatile("v_add_u32 %0, %0, %8\n v_add_u32 %1, %1, %8\n v_add_u32 %2, %2, %8\n v_add_u32 %3, %3, %8\n"
                              "v_add_u32 %4, %4, %8\n v_add_u32 %5, %5, %8\n v_add_u32 %6, %6, %8\n v_add_u32 %7, %7, %8"
                              : "+v"(a0.x), "+v"(a1.x), "+v"(a2.x), "+v"(a3.x), "+v"(a4.x), "+v"(a5.x), "+v"(a6.x), "+v"(a7.x) : "v"(x.x));)
        } else if (KIND == 7) {
            REP8(asm volatile("v_lshlrev_b32_sdwa %0, %8, %0 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1\n v_lshlrev_b32_sdwa %1, %8, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1\n"
                              "v_lshlrev_b32_sdwa %2, %8, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1\n v_lshlrev_b32_sdwa %3, %8, %3 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1\n"
                              "v_lshlrev_b32_sdwa %4, %8, %4 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1\n v_lshlrev_b32_sdwa %5, %8, %5 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1\n"
                              "v_lshlrev_b32_sdwa %6, %8, %6 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1\n v_lshlrev_b32_sdwa %7, %8, %7 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1"
                              : "+v"(a0.x), "+v"(a1.x), "+v"(a2.x), "+v"(a3.x), "+v"(a4.x), "+v"(a5.x), "+v"(a6.x), "+v"(a7.x) : "v"(x.x));)
        } else {  // v_lshl_add_u32 (address-style integer op)
            REP8(asm volatile("v_lshl_add_u32 %0, %0, 1, %8\n v_lshl_add_u32 %1, %1, 1, %8\n v_lshl_add_u32 %2, %2, 1, %8\n v_lshl_add_u32 %3, %3, 1, %8\n"
                              "v_lshl_add_u32 %4, %4, 1, %8\n v_lshl_add_u32 %5, %5, 1, %8\n v_lshl_add_u32 %6, %6, 1, %8\n v_lshl_add_u32 %7, %7, 1, %8"
                              : "+v"(a0.x), "+v"(a1.x), "+v"(a2.x), "+v"(a3.x), "+v"(a4.x), "+v"(a5.x), "+v"(a6.x), "+v"(a7.x) : "v"(x.x));)
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0.x + a1.x + a2.x + a3.x + a4.x + a5.x + a6.x + a7.x + a0.y + a1.y + a2.y + a3.y + a4.y + a5.y + a6.y + a7.y;
}
template <int KIND>
void run(const char *name, int waves_per_simd, float *d) {
    const int iters = 20000;  // x 64 instructions
    const int threads = 64 * 4 * waves_per_simd;  // one WG per CU, waves spread over the 4 SIMDs
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    k<KIND><<<256, threads>>>(d, 100, 1.f);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k<KIND><<<256, threads>>>(d, iters, 1.f);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double instr_per_simd = (double)iters * 64 * waves_per_simd;
    printf("%-16s waves/SIMD=%d  %.3f ms  -> %.2f ns per wave-instr per SIMD (= %.2f cycles @2.4GHz)\n", name, waves_per_simd, ms,
           ms * 1e6 / instr_per_simd, ms * 1e6 / instr_per_simd * 2.4);
}
int main() {
    float *d; hipMalloc(&d, 256 * 1024 * 4);
    for (int w : {2, 4}) {
        run<0>("v_add_f32", w, d); run<1>("v_pk_add_f32", w, d); run<2>("v_fma_f32", w, d); run<3>("v_pk_fma_f32", w, d); run<4>("v_lshl_add_u32", w, d); run<5>("v_pk_add_u16", w, d); run<6>("v_add_u32", w, d); run<7>("v_lshlrev_sdwa", w, d);
    }
    return 0;
}
