// valu_rate.hip -- issue-rate microbenchmark for the VALU instructions the NW kernel uses.
// Build: hipcc -O3 --offload-arch=gfx950 valu_rate.hip -o valu_rate ; run on the GPU box.
// Each kernel runs ITER x 32 independent copies of one instruction per wave; with 8 waves
// per SIMD the reported cycles/instruction/SIMD is the pipe's issue interval.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define REP8(X) X X X X X X X X
#define REP32(X) REP8(X) REP8(X) REP8(X) REP8(X)

template <int OP>
__global__ __launch_bounds__(512) void k(int* out, int iters, int seed) {
    int a0 = threadIdx.x + seed, a1 = a0 * 3, a2 = a0 * 5, a3 = a0 * 7;
    int b0 = a0 ^ 0x55, b1 = a1 ^ 0x33, b2 = a2 ^ 0x77, b3 = a3 ^ 0x11;
    int c = seed;
    for (int it = 0; it < iters; ++it) {
        if (OP == 0) { REP8(asm volatile("v_add_u32 %0, %0, %4\n v_add_u32 %1, %1, %4\n v_add_u32 %2, %2, %4\n v_add_u32 %3, %3, %4" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(c));) }
        if (OP == 1) { REP8(asm volatile("v_max3_i32 %0, %0, %4, %5\n v_max3_i32 %1, %1, %4, %5\n v_max3_i32 %2, %2, %4, %5\n v_max3_i32 %3, %3, %4, %5" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b0), "v"(b1));) }
        if (OP == 2) { REP8(asm volatile("v_and_b32 %0, %0, %4\n v_and_b32 %1, %1, %4\n v_and_b32 %2, %2, %4\n v_and_b32 %3, %3, %4" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(c));) }
        if (OP == 3) { REP8(asm volatile("v_bfi_b32 %0, 12, %0, %4\n v_bfi_b32 %1, 12, %1, %4\n v_bfi_b32 %2, 12, %2, %4\n v_bfi_b32 %3, 12, %3, %4" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b0));) }
        if (OP == 4) { REP8(asm volatile("v_cmp_eq_u32 vcc, %0, %4\n v_cndmask_b32 %0, %0, %5, vcc\n v_cmp_eq_u32 vcc, %1, %4\n v_cndmask_b32 %1, %1, %5, vcc" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b0), "v"(b1) : "vcc");) }
        if (OP == 5) { REP8(asm volatile("v_add_f32 %0, %0, %4\n v_add_f32 %1, %1, %4\n v_add_f32 %2, %2, %4\n v_add_f32 %3, %3, %4" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(c));) }
        if (OP == 6) { REP8(asm volatile("v_max3_f32 %0, %0, %4, %5\n v_max3_f32 %1, %1, %4, %5\n v_max3_f32 %2, %2, %4, %5\n v_max3_f32 %3, %3, %4, %5" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b0), "v"(b1));) }
        if (OP == 7) { REP8(asm volatile("v_pk_add_u16 %0, %0, %4\n v_pk_add_u16 %1, %1, %4\n v_pk_add_u16 %2, %2, %4\n v_pk_add_u16 %3, %3, %4" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(c));) }
        if (OP == 8) { REP8(asm volatile("v_pk_max_i16 %0, %0, %4\n v_pk_max_i16 %1, %1, %4\n v_pk_max_i16 %2, %2, %4\n v_pk_max_i16 %3, %3, %4" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b0));) }
        if (OP == 9) { REP8(asm volatile("v_mov_b32_dpp %0, %1 wave_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %1, %2 wave_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %2, %3 wave_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %3, %0 wave_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));) }
        if (OP == 10) { REP8(asm volatile("v_perm_b32 %0, %0, %4, %5\n v_perm_b32 %1, %1, %4, %5\n v_perm_b32 %2, %2, %4, %5\n v_perm_b32 %3, %3, %4, %5" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b0), "v"(b1));) }
        if (OP == 11) { REP8(asm volatile("v_and_or_b32 %0, %0, %4, 21\n v_and_or_b32 %1, %1, %4, 21\n v_and_or_b32 %2, %2, %4, 21\n v_and_or_b32 %3, %3, %4, 21" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b0));) }
        if (OP == 12) { REP8(asm volatile("v_pk_add_f32 %0, %0, %2\n v_pk_add_f32 %1, %1, %2" : "+v"(*(double*)&a0), "+v"(*(double*)&a2) : "v"(*(double*)&b0));) }
        if (OP == 13) { REP8(asm volatile("v_max_i32 %0, %0, %4\n v_max_i32 %1, %1, %4\n v_max_i32 %2, %2, %4\n v_max_i32 %3, %3, %4" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b0));) }
        if (OP == 14) { REP8(asm volatile("v_fma_f32 %0, %0, %4, %5\n v_fma_f32 %1, %1, %4, %5\n v_fma_f32 %2, %2, %4, %5\n v_fma_f32 %3, %3, %4, %5" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b0), "v"(b1));) }
        if (OP == 15) { REP8(asm volatile("v_pk_mad_i16 %0, %0, %4, %5\n v_pk_mad_i16 %1, %1, %4, %5\n v_pk_mad_i16 %2, %2, %4, %5\n v_pk_mad_i16 %3, %3, %4, %5" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b0), "v"(b1));) }
        if (OP == 16) { REP8(asm volatile("v_max_f32 %0, %0, %4\n v_max_f32 %1, %1, %4\n v_max_f32 %2, %2, %4\n v_max_f32 %3, %3, %4" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b0));) }
        if (OP == 17) { REP8(asm volatile("v_pk_max_f16 %0, %0, %4\n v_pk_max_f16 %1, %1, %4\n v_pk_max_f16 %2, %2, %4\n v_pk_max_f16 %3, %3, %4" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b0));) }
        if (OP == 18) { REP8(asm volatile("v_pk_sub_i16 %0, %0, %4 clamp\n v_pk_sub_i16 %1, %1, %4 clamp\n v_pk_sub_i16 %2, %2, %4 clamp\n v_pk_sub_i16 %3, %3, %4 clamp" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b0));) }
        // mixes: does the 2-cycle rate of add survive between 4-cycle instructions?
        if (OP == 20) { REP8(asm volatile("v_add_u32 %0, %0, %4\n v_max3_i32 %1, %1, %5, %6\n v_add_u32 %2, %2, %4\n v_max3_i32 %3, %3, %5, %6" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(c), "v"(b0), "v"(b1));) }
        if (OP == 21) { REP8(asm volatile("v_add_u32 %0, %0, %4\n v_add_u32 %2, %2, %4\n v_max3_i32 %1, %1, %5, %6\n v_max3_i32 %3, %3, %5, %6" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(c), "v"(b0), "v"(b1));) }
        // the score-only cell of nw_score_kernel, 4 rows chained as in the kernel (8 instructions per cell)
        if (OP == 22) { REP8(asm volatile(
            "v_cmp_eq_u32 vcc, %4, %5\n v_cndmask_b32 %0, %6, %7, vcc\n v_add_u32 %0, %0, %1\n v_add_u32 %2, %2, %6\n v_add_u32 %3, %3, %7\n"
            "v_max3_i32 %1, %0, %2, %3\n v_max3_i32 %2, %0, %2, %3\n v_max3_i32 %3, %0, %1, %3\n"
            "v_cmp_eq_u32 vcc, %4, %6\n v_cndmask_b32 %0, %6, %7, vcc\n v_add_u32 %0, %0, %1\n v_add_u32 %2, %2, %6\n v_add_u32 %3, %3, %7\n"
            "v_max3_i32 %1, %0, %2, %3\n v_max3_i32 %2, %0, %2, %3\n v_max3_i32 %3, %0, %1, %3\n"
            "v_cmp_eq_u32 vcc, %4, %7\n v_cndmask_b32 %0, %6, %7, vcc\n v_add_u32 %0, %0, %1\n v_add_u32 %2, %2, %6\n v_add_u32 %3, %3, %7\n"
            "v_max3_i32 %1, %0, %2, %3\n v_max3_i32 %2, %0, %2, %3\n v_max3_i32 %3, %0, %1, %3\n"
            "v_cmp_eq_u32 vcc, %5, %7\n v_cndmask_b32 %0, %6, %7, vcc\n v_add_u32 %0, %0, %1\n v_add_u32 %2, %2, %6\n v_add_u32 %3, %3, %7\n"
            "v_max3_i32 %1, %0, %2, %3\n v_max3_i32 %2, %0, %2, %3\n v_max3_i32 %3, %0, %1, %3"
            : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(c), "v"(b0), "v"(b1), "v"(b2) : "vcc");) }
        // the same cell with the compare result in an SGPR pair and the select folded into VOP3
        if (OP == 23) { REP8(asm volatile(
            "v_cmp_eq_u32 s[20:21], %4, %5\n v_cndmask_b32 %0, %6, %7, s[20:21]\n v_add_u32 %0, %0, %1\n v_add_u32 %2, %2, %6\n v_add_u32 %3, %3, %7\n"
            "v_max3_i32 %1, %0, %2, %3\n v_max3_i32 %2, %0, %2, %3\n v_max3_i32 %3, %0, %1, %3\n"
            "v_cmp_eq_u32 s[22:23], %4, %6\n v_cndmask_b32 %0, %6, %7, s[22:23]\n v_add_u32 %0, %0, %1\n v_add_u32 %2, %2, %6\n v_add_u32 %3, %3, %7\n"
            "v_max3_i32 %1, %0, %2, %3\n v_max3_i32 %2, %0, %2, %3\n v_max3_i32 %3, %0, %1, %3\n"
            "v_cmp_eq_u32 s[24:25], %4, %7\n v_cndmask_b32 %0, %6, %7, s[24:25]\n v_add_u32 %0, %0, %1\n v_add_u32 %2, %2, %6\n v_add_u32 %3, %3, %7\n"
            "v_max3_i32 %1, %0, %2, %3\n v_max3_i32 %2, %0, %2, %3\n v_max3_i32 %3, %0, %1, %3\n"
            "v_cmp_eq_u32 s[26:27], %5, %7\n v_cndmask_b32 %0, %6, %7, s[26:27]\n v_add_u32 %0, %0, %1\n v_add_u32 %2, %2, %6\n v_add_u32 %3, %3, %7\n"
            "v_max3_i32 %1, %0, %2, %3\n v_max3_i32 %2, %0, %2, %3\n v_max3_i32 %3, %0, %1, %3"
            : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(c), "v"(b0), "v"(b1), "v"(b2) : "s20", "s21", "s22", "s23", "s24", "s25", "s26", "s27");) }
        if (OP == 24) { REP8(asm volatile("v_cmp_eq_u32 vcc, %0, %4\n v_cmp_eq_u32 vcc, %1, %4\n v_cmp_eq_u32 vcc, %2, %4\n v_cmp_eq_u32 vcc, %3, %4" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b0) : "vcc");) }
        if (OP == 25) { REP8(asm volatile("v_cndmask_b32 %0, %0, %4, vcc\n v_cndmask_b32 %1, %1, %4, vcc\n v_cndmask_b32 %2, %2, %4, vcc\n v_cndmask_b32 %3, %3, %4, vcc" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b0) : "vcc");) }

        // ---- round 2: candidates for a cheaper score-only cell ----
        if (OP == 30) { REP8(asm volatile("v_add_u32_sdwa %0, %0, sext(%4) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_0\n v_add_u32_sdwa %1, %1, sext(%4) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1\n v_add_u32_sdwa %2, %2, sext(%4) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_2\n v_add_u32_sdwa %3, %3, sext(%4) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_3" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(c));) }
        if (OP == 31) { REP8(asm volatile("v_pk_maximum3_f16 %0, %0, %4, %5\n v_pk_maximum3_f16 %1, %1, %4, %5\n v_pk_maximum3_f16 %2, %2, %4, %5\n v_pk_maximum3_f16 %3, %3, %4, %5" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b0), "v"(b1));) }
        if (OP == 32) { REP8(asm volatile("v_maximum3_f32 %0, %0, %4, %5\n v_maximum3_f32 %1, %1, %4, %5\n v_maximum3_f32 %2, %2, %4, %5\n v_maximum3_f32 %3, %3, %4, %5" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b0), "v"(b1));) }
        if (OP == 33) { REP8(asm volatile("v_add3_u32 %0, %0, %4, %5\n v_add3_u32 %1, %1, %4, %5\n v_add3_u32 %2, %2, %4, %5\n v_add3_u32 %3, %3, %4, %5" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b0), "v"(b1));) }
        if (OP == 34) { REP8(asm volatile("v_sub_u32_e64 %0, %0, %4 clamp\n v_sub_u32_e64 %1, %1, %4 clamp\n v_sub_u32_e64 %2, %2, %4 clamp\n v_sub_u32_e64 %3, %3, %4 clamp" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b0));) }
        if (OP == 35) { REP8(asm volatile("v_max_i32_dpp %0, %1, %0 wave_shr:1 row_mask:0xf bank_mask:0xf\n v_max_i32_dpp %1, %2, %1 wave_shr:1 row_mask:0xf bank_mask:0xf\n v_max_i32_dpp %2, %3, %2 wave_shr:1 row_mask:0xf bank_mask:0xf\n v_max_i32_dpp %3, %0, %3 wave_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));) }
        if (OP == 36) { REP8(asm volatile("v_lshl_add_u32 %0, %0, 1, %4\n v_lshl_add_u32 %1, %1, 1, %4\n v_lshl_add_u32 %2, %2, 1, %4\n v_lshl_add_u32 %3, %3, 1, %4" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b0));) }
        if (OP == 37) { REP8(asm volatile("v_max_i32_sdwa %0, %0, sext(%4) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_0\n v_max_i32_sdwa %1, %1, sext(%4) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_1\n v_max_i32_sdwa %2, %2, sext(%4) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_0\n v_max_i32_sdwa %3, %3, sext(%4) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_1" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b0));) }
        if (OP == 38) { REP8(asm volatile("v_pk_add_i16 %0, %0, %4\n v_pk_add_i16 %1, %1, %4\n v_pk_add_i16 %2, %2, %4\n v_pk_add_i16 %3, %3, %4" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(c));) }
        if (OP == 39) { REP8(asm volatile("v_mov_b32_dpp %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %1, %2 row_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %2, %3 row_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %3, %0 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));) }
        if (OP == 40) { REP8(asm volatile("v_xor_b32 %0, %0, %4\n v_or_b32 %1, %1, %4\n v_sub_u32 %2, %2, %4\n v_xor_b32 %3, %3, %4" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(c));) }
        if (OP == 41) { REP8(asm volatile("v_lshlrev_b32 %0, 1, %0\n v_ashrrev_i32 %1, 1, %1\n v_lshlrev_b32 %2, 1, %2\n v_ashrrev_i32 %3, 1, %3" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));) }
        if (OP == 42) { REP8(asm volatile("v_mad_u32_u24 %0, %0, %4, %5\n v_mad_u32_u24 %1, %1, %4, %5\n v_mad_u32_u24 %2, %2, %4, %5\n v_mad_u32_u24 %3, %3, %4, %5" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b0), "v"(b1));) }
        if (OP == 43) { REP8(asm volatile("v_mov_b32 %0, %1\n v_mov_b32 %1, %2\n v_mov_b32 %2, %3\n v_mov_b32 %3, %0" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));) }
        if (OP == 44) { REP8(asm volatile("v_add_u32_dpp %0, %1, %0 wave_shr:1 row_mask:0xf bank_mask:0xf\n v_add_u32_dpp %1, %2, %1 wave_shr:1 row_mask:0xf bank_mask:0xf\n v_add_u32_dpp %2, %3, %2 wave_shr:1 row_mask:0xf bank_mask:0xf\n v_add_u32_dpp %3, %0, %3 wave_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));) }
        if (OP == 45) { REP8(asm volatile("v_max_i16 %0, %0, %4\n v_max_i16 %1, %1, %4\n v_max_i16 %2, %2, %4\n v_max_i16 %3, %3, %4" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b0));) }
        if (OP == 46) { REP8(asm volatile("v_med3_i32 %0, %0, %4, %5\n v_med3_i32 %1, %1, %4, %5\n v_min3_i32 %2, %2, %4, %5\n v_min3_i32 %3, %3, %4, %5" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b0), "v"(b1));) }
        if (OP == 47) { REP8(asm volatile("v_cmp_gt_i32 vcc, %0, %4\n v_addc_co_u32 %0, vcc, %0, %4, vcc\n v_cmp_gt_i32 vcc, %1, %4\n v_addc_co_u32 %1, vcc, %1, %4, vcc" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b0) : "vcc");) }
        // new score-only cell, 4 rows chained: M = d_ul + sbyte (sdwa); D = max3(M, xg, yg); Dg = D + go;
        // xg' = max(Dg, xg); yg' = max(Dg, yg).  Registers: a0 = s bytes, a1 = D chain, a2 = xg, a3 = yg
        if (OP == 50) { REP8(asm volatile(
            "v_add_u32_sdwa %0, %4, sext(%5) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_0\n v_max3_i32 %1, %0, %2, %3\n v_add_u32 %0, %1, %6\n v_max_i32 %2, %0, %2\n v_max_i32 %3, %0, %3\n"
            "v_add_u32_sdwa %0, %1, sext(%5) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1\n v_max3_i32 %1, %0, %2, %7\n v_add_u32 %0, %1, %6\n v_max_i32 %2, %0, %2\n v_max_i32 %7, %0, %7\n"
            "v_add_u32_sdwa %0, %4, sext(%5) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_2\n v_max3_i32 %1, %0, %2, %3\n v_add_u32 %0, %1, %6\n v_max_i32 %2, %0, %2\n v_max_i32 %3, %0, %3\n"
            "v_add_u32_sdwa %0, %1, sext(%5) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_3\n v_max3_i32 %1, %0, %2, %7\n v_add_u32 %0, %1, %6\n v_max_i32 %2, %0, %2\n v_max_i32 %7, %0, %7"
            : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(c), "v"(b0), "v"(b1), "v"(b2));) }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + b2 + b3;
}

template <int OP>
static void run(const char* name, int instr_per_iter, int blocks_per_cu = 4) {
    int* out; hipMalloc(&out, 256 * 8 * 512 * sizeof(int));
    const int iters = 2000, blocks = 256 * blocks_per_cu;   // 4 blocks of 512 = 32 waves per CU = 8 per SIMD
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(512), 0, 0, out, 10, 1);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(512), 0, 0, out, iters, 1);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    // wave-instructions per SIMD = iters * instr_per_iter * 8 waves
    double wi = (double)iters * instr_per_iter * 2.0 * blocks_per_cu;
    double ns_per = ms * 1e6 / wi;
    printf("%-28s %8.3f ms  %6.3f ns per wave-instr per SIMD  (= %.2f cycles @2.4GHz, %.2f @2.0GHz)\n",
           name, ms, ns_per, ns_per * 2.4, ns_per * 2.0);
    hipFree(out);
}

int main() {
    run<0>("v_add_u32", 32); run<13>("v_max_i32", 32); run<1>("v_max3_i32", 32); run<2>("v_and_b32", 32);
    run<3>("v_bfi_b32", 32); run<11>("v_and_or_b32", 32); run<4>("v_cmp_eq+v_cndmask (pair=2)", 32);
    run<10>("v_perm_b32", 32); run<9>("v_mov_b32_dpp wave_shr", 32);
    run<5>("v_add_f32", 32); run<16>("v_max_f32", 32); run<6>("v_max3_f32", 32); run<14>("v_fma_f32", 32);
    run<12>("v_pk_add_f32", 16);
    run<7>("v_pk_add_u16", 32); run<8>("v_pk_max_i16", 32); run<15>("v_pk_mad_i16", 32);
    run<18>("v_pk_sub_i16 clamp", 32); run<17>("v_pk_max_f16", 32);
    run<20>("add,max3 alternating", 32); run<21>("add,add,max3,max3", 32);
    run<24>("v_cmp_eq_u32 alone", 32); run<25>("v_cndmask_b32 alone", 32);
    run<22>("score-only cell x4 (vcc), per instr", 8 * 32); run<23>("score-only cell x4 (sgpr mask), per instr", 8 * 32);

    printf("---- round 2 ----\n");
    run<25>("v_cndmask_b32 alone", 32);
    run<30>("v_add_u32_sdwa byte sext", 32); run<31>("v_pk_maximum3_f16", 32); run<32>("v_maximum3_f32", 32);
    run<33>("v_add3_u32", 32); run<34>("v_sub_u32 clamp (e64)", 32); run<35>("v_max_i32_dpp wave_shr", 32);
    run<36>("v_lshl_add_u32", 32); run<37>("v_max_i32_sdwa word sext", 32); run<38>("v_pk_add_i16", 32);
    run<39>("v_mov_b32_dpp row_shr", 32); run<40>("xor/or/sub/xor", 32); run<41>("lshlrev/ashrrev", 32);
    run<42>("v_mad_u32_u24", 32); run<43>("v_mov_b32", 32); run<44>("v_add_u32_dpp wave_shr", 32);
    run<45>("v_max_i16", 32); run<46>("med3/min3 i32", 32); run<47>("cmp_gt + addc (pair=2)", 32);
    run<50>("new cell x4 (sdwa,max3,add,max,max) per instr, 8 w/SIMD", 8 * 20);
    run<50>("new cell x4 per instr, 4 w/SIMD", 8 * 20, 2);
    run<50>("new cell x4 per instr, 2 w/SIMD", 8 * 20, 1);
    run<23>("old cell x4 (sgpr mask) per instr, 2 w/SIMD", 8 * 32, 1);
    run<23>("old cell x4 (sgpr mask) per instr, 4 w/SIMD", 8 * 32, 2);
    return 0;
}
