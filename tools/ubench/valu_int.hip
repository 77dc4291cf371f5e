// Micro-benchmark (development tool): what the integer VALU instructions of the ALAC / FLAC kernels cost on gfx950, in the SIMD's time per
// wave-instruction at 1..4 wavefronts per SIMD (eight independent chains per wavefront, so no dependency stalls).
//   hipcc --offload-arch=gfx950 -O3 -o tools/ubench/build/valu_int tools/ubench/valu_int.hip && tools/ubench/build/valu_int
#include <hip/hip_runtime.h>

#include <cstdio>
#include <vector>

#define REP8(x) x x x x x x x x
#define OPS8(op, tail) \
    REP8(asm volatile(op " %0, %0" tail "\n" op " %1, %1" tail "\n" op " %2, %2" tail "\n" op " %3, %3" tail "\n" op " %4, %4" tail "\n" op " %5, %5" tail "\n" op " %6, %6" tail "\n" op " %7, %7" tail \
                      : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c), "v"(d) : "vcc");)

template <int KIND>
__global__ void k(int *out, int iters, int seed) {
    const int s = seed + threadIdx.x;
    int a0 = s, a1 = s + 1, a2 = s + 2, a3 = s + 3, a4 = s + 4, a5 = s + 5, a6 = s + 6, a7 = s + 7, c = 3, d = s ^ 5;
    for (int it = 0; it < iters; ++it) {
        if constexpr (KIND == 0) { OPS8("v_sub_u32", ", %8") }
        else if constexpr (KIND == 1) { OPS8("v_xor_b32", ", %8") }
        else if constexpr (KIND == 2) { OPS8("v_mul_i32_i24", ", %8") }
        else if constexpr (KIND == 3) { OPS8("v_mad_i32_i24", ", %8, %9") }
        else if constexpr (KIND == 4) { OPS8("v_mul_lo_u32", ", %8") }
        else if constexpr (KIND == 5) { OPS8("v_med3_i32", ", %8, %9") }
        else if constexpr (KIND == 6) { OPS8("v_add3_u32", ", %8, %9") }
        else if constexpr (KIND == 7) { OPS8("v_lshl_add_u32", ", %8, %9") }
        else if constexpr (KIND == 8) { OPS8("v_ashrrev_i32", ", %8") }
        else if constexpr (KIND == 9) { OPS8("v_cndmask_b32", ", %8, vcc") }
        else if constexpr (KIND == 10) { OPS8("v_bfi_b32", ", %8, %9") }
        else if constexpr (KIND == 11) { OPS8("v_mad_u32_u24", ", %8, %9") }
        else if constexpr (KIND == 12) { OPS8("v_xad_u32", ", %8, %9") }
        else if constexpr (KIND == 13) { OPS8("v_sub_u32", ", 7") }
        else if constexpr (KIND == 14) { OPS8("v_add_f32", ", %8") }
        else if constexpr (KIND == 15) { OPS8("v_and_or_b32", ", %8, %9") }
        else if constexpr (KIND == 16) { OPS8("v_mad_i32_i24", ", 5, %9") }
        else if constexpr (KIND == 17) {  // the mask in an SGPR pair that nothing in the loop writes
            REP8(asm volatile("v_cndmask_b32 %0, %0, %8, s[20:21]\n v_cndmask_b32 %1, %1, %8, s[20:21]\n v_cndmask_b32 %2, %2, %8, s[20:21]\n v_cndmask_b32 %3, %3, %8, s[20:21]\n"
                              "v_cndmask_b32 %4, %4, %8, s[20:21]\n v_cndmask_b32 %5, %5, %8, s[20:21]\n v_cndmask_b32 %6, %6, %8, s[20:21]\n v_cndmask_b32 %7, %7, %8, s[20:21]"
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c), "v"(d) : "s20", "s21");)
        } else if constexpr (KIND == 18) {  // compare into vcc, select on vcc (the compiler's usual pair): 16 instructions per block
            REP8(asm volatile("v_cmp_gt_i32 vcc, %0, %8\n v_cndmask_b32 %0, %0, %9, vcc\n v_cmp_gt_i32 vcc, %1, %8\n v_cndmask_b32 %1, %1, %9, vcc\n v_cmp_gt_i32 vcc, %2, %8\n v_cndmask_b32 %2, %2, %9, vcc\n v_cmp_gt_i32 vcc, %3, %8\n v_cndmask_b32 %3, %3, %9, vcc\n"
                              "v_cmp_gt_i32 vcc, %4, %8\n v_cndmask_b32 %4, %4, %9, vcc\n v_cmp_gt_i32 vcc, %5, %8\n v_cndmask_b32 %5, %5, %9, vcc\n v_cmp_gt_i32 vcc, %6, %8\n v_cndmask_b32 %6, %6, %9, vcc\n v_cmp_gt_i32 vcc, %7, %8\n v_cndmask_b32 %7, %7, %9, vcc"
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c), "v"(d) : "vcc");)
        } else if constexpr (KIND == 19) {  // compare into an SGPR pair, select on it, the pairs interleaved two apart as a scheduler would
            REP8(asm volatile("v_cmp_gt_i32 s[20:21], %0, %8\n v_cmp_gt_i32 s[22:23], %1, %8\n v_cndmask_b32 %0, %0, %9, s[20:21]\n v_cndmask_b32 %1, %1, %9, s[22:23]\n v_cmp_gt_i32 s[24:25], %2, %8\n v_cmp_gt_i32 s[26:27], %3, %8\n v_cndmask_b32 %2, %2, %9, s[24:25]\n v_cndmask_b32 %3, %3, %9, s[26:27]\n"
                              "v_cmp_gt_i32 s[20:21], %4, %8\n v_cmp_gt_i32 s[22:23], %5, %8\n v_cndmask_b32 %4, %4, %9, s[20:21]\n v_cndmask_b32 %5, %5, %9, s[22:23]\n v_cmp_gt_i32 s[24:25], %6, %8\n v_cmp_gt_i32 s[26:27], %7, %8\n v_cndmask_b32 %6, %6, %9, s[24:25]\n v_cndmask_b32 %7, %7, %9, s[26:27]"
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c), "v"(d) : "s20", "s21", "s22", "s23", "s24", "s25", "s26", "s27");)
        } else if constexpr (KIND == 20) {  // compares only
            REP8(asm volatile("v_cmp_gt_i32 s[20:21], %0, %8\n v_cmp_gt_i32 s[22:23], %1, %8\n v_cmp_gt_i32 s[24:25], %2, %8\n v_cmp_gt_i32 s[26:27], %3, %8\n v_cmp_gt_i32 s[20:21], %4, %8\n v_cmp_gt_i32 s[22:23], %5, %8\n v_cmp_gt_i32 s[24:25], %6, %8\n v_cmp_gt_i32 s[26:27], %7, %8"
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c), "v"(d) : "s20", "s21", "s22", "s23", "s24", "s25", "s26", "s27");)
        } else if constexpr (KIND == 21) {  // compare (vcc, VOPC encoding) only
            REP8(asm volatile("v_cmp_gt_i32 vcc, %0, %8\n v_cmp_gt_i32 vcc, %1, %8\n v_cmp_gt_i32 vcc, %2, %8\n v_cmp_gt_i32 vcc, %3, %8\n v_cmp_gt_i32 vcc, %4, %8\n v_cmp_gt_i32 vcc, %5, %8\n v_cmp_gt_i32 vcc, %6, %8\n v_cmp_gt_i32 vcc, %7, %8"
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c), "v"(d) : "vcc");)
        } else if constexpr (KIND == 22) { OPS8("v_and_b32", ", %8") }
        else if constexpr (KIND == 23) { OPS8("v_cvt_f32_i32", "") }
        else if constexpr (KIND == 24) { OPS8("v_cvt_i32_f32", "") }
        else if constexpr (KIND == 25) { OPS8("v_mul_f32", ", %8") }
        else if constexpr (KIND == 26) { OPS8("v_max_i32", ", %8") }
        else if constexpr (KIND == 27) { OPS8("v_lshlrev_b32", ", 3") }
        else if constexpr (KIND == 28) { OPS8("v_mul_hi_u32", ", %8") }
        else if constexpr (KIND == 29) { OPS8("v_ffbh_u32", "") }
        else if constexpr (KIND == 30) { OPS8("v_cvt_f32_ubyte0", "") }
        else if constexpr (KIND == 31) { OPS8("v_rcp_f32", "") }
        else if constexpr (KIND == 32) { OPS8("v_floor_f32", "") }
        else if constexpr (KIND == 33) { OPS8("v_trunc_f32", "") }
        else if constexpr (KIND == 34) { OPS8("v_fma_f64", "") }
        else { OPS8("v_perm_b32", ", %8, %9") }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
}

// 64-bit accumulators: the FLAC sum's candidates
template <int KIND>
__global__ void k64(long long *out, int iters, int seed, int big) {
    const int s = seed + threadIdx.x;
    long long a0 = s, a1 = s + 1, a2 = s + 2, a3 = s + 3;
    double f0 = s, f1 = s + 1, f2 = s + 2, f3 = s + 3, fc = 1.0000001, fd = 0.5;
    double g[16];
    int gi[16];
    for (int i = 0; i < 16; ++i) { g[i] = 1.0 + 1e-9 * (s + i); gi[i] = 0x1234567 * (i + 1) + s; }
    int c = big ? 0x12345678 + s : 3, d = big ? (int)0x7edcba98 - s : s ^ 5;
    for (int it = 0; it < iters; ++it) {
        if constexpr (KIND == 0) {
            REP8(asm volatile("v_mad_i64_i32 %0, vcc, %4, %5, %0\n v_mad_i64_i32 %1, vcc, %4, %5, %1\n v_mad_i64_i32 %2, vcc, %4, %5, %2\n v_mad_i64_i32 %3, vcc, %4, %5, %3\n"
                              "v_mad_i64_i32 %0, vcc, %4, %5, %0\n v_mad_i64_i32 %1, vcc, %4, %5, %1\n v_mad_i64_i32 %2, vcc, %4, %5, %2\n v_mad_i64_i32 %3, vcc, %4, %5, %3"
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(c), "v"(d) : "vcc");)
        } else if constexpr (KIND == 1) {
            REP8(asm volatile("v_fma_f64 %0, %4, %5, %0\n v_fma_f64 %1, %4, %5, %1\n v_fma_f64 %2, %4, %5, %2\n v_fma_f64 %3, %4, %5, %3\n"
                              "v_fma_f64 %0, %4, %5, %0\n v_fma_f64 %1, %4, %5, %1\n v_fma_f64 %2, %4, %5, %2\n v_fma_f64 %3, %4, %5, %3"
                              : "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3) : "v"(fc), "v"(fd));)
        } else if constexpr (KIND == 2) {
            REP8(asm volatile("v_add_f64 %0, %4, %0\n v_add_f64 %1, %4, %1\n v_add_f64 %2, %4, %2\n v_add_f64 %3, %4, %3\n"
                              "v_add_f64 %0, %5, %0\n v_add_f64 %1, %5, %1\n v_add_f64 %2, %5, %2\n v_add_f64 %3, %5, %3"
                              : "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3) : "v"(fc), "v"(fd));)
        } else if constexpr (KIND == 3) {
            REP8(asm volatile("v_cvt_f64_i32 %0, %4\n v_cvt_f64_i32 %1, %5\n v_cvt_f64_i32 %2, %4\n v_cvt_f64_i32 %3, %5\n"
                              "v_cvt_f64_i32 %0, %5\n v_cvt_f64_i32 %1, %4\n v_cvt_f64_i32 %2, %5\n v_cvt_f64_i32 %3, %4"
                              : "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3) : "v"(c), "v"(d));)
        } else if constexpr (KIND == 5) {  // ONE dependent chain per wavefront
            REP8(asm volatile("v_mad_i64_i32 %0, vcc, %4, %5, %0\n v_mad_i64_i32 %0, vcc, %4, %5, %0\n v_mad_i64_i32 %0, vcc, %4, %5, %0\n v_mad_i64_i32 %0, vcc, %4, %5, %0\n"
                              "v_mad_i64_i32 %0, vcc, %4, %5, %0\n v_mad_i64_i32 %0, vcc, %4, %5, %0\n v_mad_i64_i32 %0, vcc, %4, %5, %0\n v_mad_i64_i32 %0, vcc, %4, %5, %0"
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(c), "v"(d) : "vcc");)
        } else if constexpr (KIND == 6) {  // two chains
            REP8(asm volatile("v_mad_i64_i32 %0, vcc, %4, %5, %0\n v_mad_i64_i32 %1, vcc, %4, %5, %1\n v_mad_i64_i32 %0, vcc, %4, %5, %0\n v_mad_i64_i32 %1, vcc, %4, %5, %1\n"
                              "v_mad_i64_i32 %0, vcc, %4, %5, %0\n v_mad_i64_i32 %1, vcc, %4, %5, %1\n v_mad_i64_i32 %0, vcc, %4, %5, %0\n v_mad_i64_i32 %1, vcc, %4, %5, %1"
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(c), "v"(d) : "vcc");)
        } else if constexpr (KIND == 7) {  // one dependent FP64 FMA chain
            REP8(asm volatile("v_fma_f64 %0, %4, %5, %0\n v_fma_f64 %0, %4, %5, %0\n v_fma_f64 %0, %4, %5, %0\n v_fma_f64 %0, %4, %5, %0\n"
                              "v_fma_f64 %0, %4, %5, %0\n v_fma_f64 %0, %4, %5, %0\n v_fma_f64 %0, %4, %5, %0\n v_fma_f64 %0, %4, %5, %0"
                              : "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3) : "v"(fc), "v"(fd));)
        } else if constexpr (KIND == 8) {  // two FP64 FMA chains
            REP8(asm volatile("v_fma_f64 %0, %4, %5, %0\n v_fma_f64 %1, %4, %5, %1\n v_fma_f64 %0, %4, %5, %0\n v_fma_f64 %1, %4, %5, %1\n"
                              "v_fma_f64 %0, %4, %5, %0\n v_fma_f64 %1, %4, %5, %1\n v_fma_f64 %0, %4, %5, %0\n v_fma_f64 %1, %4, %5, %1"
                              : "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3) : "v"(fc), "v"(fd));)
        } else if constexpr (KIND == 9) {  // one FP64 FMA chain, eight DIFFERENT operand pairs (16 more VGPR pairs read per block)
            REP8(asm volatile("v_fma_f64 %0, %1, %9, %0\n v_fma_f64 %0, %2, %10, %0\n v_fma_f64 %0, %3, %11, %0\n v_fma_f64 %0, %4, %12, %0\n"
                              "v_fma_f64 %0, %5, %13, %0\n v_fma_f64 %0, %6, %14, %0\n v_fma_f64 %0, %7, %15, %0\n v_fma_f64 %0, %8, %16, %0"
                              : "+v"(f0) : "v"(g[0]), "v"(g[1]), "v"(g[2]), "v"(g[3]), "v"(g[4]), "v"(g[5]), "v"(g[6]), "v"(g[7]),
                                "v"(g[8]), "v"(g[9]), "v"(g[10]), "v"(g[11]), "v"(g[12]), "v"(g[13]), "v"(g[14]), "v"(g[15]));)
        } else if constexpr (KIND == 10) {  // the same with v_mad_i64_i32 (sixteen 32-bit operands)
            REP8(asm volatile("v_mad_i64_i32 %0, vcc, %1, %9, %0\n v_mad_i64_i32 %0, vcc, %2, %10, %0\n v_mad_i64_i32 %0, vcc, %3, %11, %0\n v_mad_i64_i32 %0, vcc, %4, %12, %0\n"
                              "v_mad_i64_i32 %0, vcc, %5, %13, %0\n v_mad_i64_i32 %0, vcc, %6, %14, %0\n v_mad_i64_i32 %0, vcc, %7, %15, %0\n v_mad_i64_i32 %0, vcc, %8, %16, %0"
                              : "+v"(a0) : "v"(gi[0]), "v"(gi[1]), "v"(gi[2]), "v"(gi[3]), "v"(gi[4]), "v"(gi[5]), "v"(gi[6]), "v"(gi[7]),
                                "v"(gi[8]), "v"(gi[9]), "v"(gi[10]), "v"(gi[11]), "v"(gi[12]), "v"(gi[13]), "v"(gi[14]), "v"(gi[15]) : "vcc");)
        } else if constexpr (KIND == 11) {  // the carry-out in an SGPR pair (what hipcc emits) instead of vcc
            REP8(asm volatile("v_mad_i64_i32 %0, s[20:21], %4, %5, %0\n v_mad_i64_i32 %1, s[20:21], %4, %5, %1\n v_mad_i64_i32 %2, s[20:21], %4, %5, %2\n v_mad_i64_i32 %3, s[20:21], %4, %5, %3\n"
                              "v_mad_i64_i32 %0, s[20:21], %4, %5, %0\n v_mad_i64_i32 %1, s[20:21], %4, %5, %1\n v_mad_i64_i32 %2, s[20:21], %4, %5, %2\n v_mad_i64_i32 %3, s[20:21], %4, %5, %3"
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(c), "v"(d) : "s20", "s21");)
        } else if constexpr (KIND == 12) {  // ... and ONE chain of them
            REP8(asm volatile("v_mad_i64_i32 %0, s[20:21], %4, %5, %0\n v_mad_i64_i32 %0, s[20:21], %4, %5, %0\n v_mad_i64_i32 %0, s[20:21], %4, %5, %0\n v_mad_i64_i32 %0, s[20:21], %4, %5, %0\n"
                              "v_mad_i64_i32 %0, s[20:21], %4, %5, %0\n v_mad_i64_i32 %0, s[20:21], %4, %5, %0\n v_mad_i64_i32 %0, s[20:21], %4, %5, %0\n v_mad_i64_i32 %0, s[20:21], %4, %5, %0"
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(c), "v"(d) : "s20", "s21");)
        } else if constexpr (KIND == 13) {  // 64-bit add as the compiler writes it: add_co + addc_co through an SGPR pair (x4 = 8 instructions)
            REP8(asm volatile("v_add_co_u32 %0, s[20:21], %0, %2\n v_addc_co_u32 %1, s[20:21], %1, %3, s[20:21]\n v_add_co_u32 %0, s[20:21], %0, %2\n v_addc_co_u32 %1, s[20:21], %1, %3, s[20:21]\n"
                              "v_add_co_u32 %0, s[20:21], %0, %2\n v_addc_co_u32 %1, s[20:21], %1, %3, s[20:21]\n v_add_co_u32 %0, s[20:21], %0, %2\n v_addc_co_u32 %1, s[20:21], %1, %3, s[20:21]"
                              : "+v"(c), "+v"(d) : "v"(gi[0]), "v"(gi[1]) : "s20", "s21");)
        } else if constexpr (KIND == 14) {  // ... through vcc
            REP8(asm volatile("v_add_co_u32 %0, vcc, %0, %2\n v_addc_co_u32 %1, vcc, %1, %3, vcc\n v_add_co_u32 %0, vcc, %0, %2\n v_addc_co_u32 %1, vcc, %1, %3, vcc\n"
                              "v_add_co_u32 %0, vcc, %0, %2\n v_addc_co_u32 %1, vcc, %1, %3, vcc\n v_add_co_u32 %0, vcc, %0, %2\n v_addc_co_u32 %1, vcc, %1, %3, vcc"
                              : "+v"(c), "+v"(d) : "v"(gi[0]), "v"(gi[1]) : "vcc");)
        } else if constexpr (KIND == 15) {  // v_lshl_add_u64 (no carry-out)
            REP8(asm volatile("v_lshl_add_u64 %0, %0, 0, %4\n v_lshl_add_u64 %1, %1, 0, %4\n v_lshl_add_u64 %2, %2, 0, %4\n v_lshl_add_u64 %3, %3, 0, %4\n"
                              "v_lshl_add_u64 %0, %0, 0, %4\n v_lshl_add_u64 %1, %1, 0, %4\n v_lshl_add_u64 %2, %2, 0, %4\n v_lshl_add_u64 %3, %3, 0, %4"
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(a0));)
        } else if constexpr (KIND == 16) {  // 64 dependent multiply-adds in ONE asm statement: nothing between them
            asm volatile("v_mad_i64_i32 %0, vcc, %4, %5, %0\n v_mad_i64_i32 %0, vcc, %4, %5, %0\n v_mad_i64_i32 %0, vcc, %4, %5, %0\n v_mad_i64_i32 %0, vcc, %4, %5, %0\n v_mad_i64_i32 %0, vcc, %4, %5, %0\n v_mad_i64_i32 %0, vcc, %4, %5, %0\n v_mad_i64_i32 %0, vcc, %4, %5, %0\n v_mad_i64_i32 %0, vcc, %4, %5, %0\n v_mad_i64_i32 %0, vcc, %4, %5, %0\n v_mad_i64_i32 %0, vcc, %4, %5, %0\n v_mad_i64_i32 %0, vcc, %4, %5, %0\n v_mad_i64_i32 %0, vcc, %4, %5, %0\n v_mad_i64_i32 %0, vcc, %4, %5, %0\n v_mad_i64_i32 %0, vcc, %4, %5, %0\n v_mad_i64_i32 %0, vcc, %4, %5, %0\n v_mad_i64_i32 %0, vcc, %4, %5, %0\n v_mad_i64_i32 %0, vcc, %4, %5, %0\n v_mad_i64_i32 %0, vcc, %4, %5, %0\n v_mad_i64_i32 %0, vcc, %4, %5, %0\n v_mad_i64_i32 %0, vcc, %4, %5, %0\n v_mad_i64_i32 %0, vcc, %4, %5, %0\n v_mad_i64_i32 %0, vcc, %4, %5, %0\n v_mad_i64_i32 %0, vcc, %4, %5, %0\n v_mad_i64_i32 %0, vcc, %4, %5, %0\n v_mad_i64_i32 %0, vcc, %4, %5, %0\n v_mad_i64_i32 %0, vcc, %4, %5, %0\n v_mad_i64_i32 %0, vcc, %4, %5, %0\n v_mad_i64_i32 %0, vcc, %4, %5, %0\n v_mad_i64_i32 %0, vcc, %4, %5, %0\n v_mad_i64_i32 %0, vcc, %4, %5, %0\n v_mad_i64_i32 %0, vcc, %4, %5, %0\n v_mad_i64_i32 %0, vcc, %4, %5, %0\n v_mad_i64_i32 %0, vcc, %4, %5, %0\n v_mad_i64_i32 %0, vcc, %4, %5, %0\n v_mad_i64_i32 %0, vcc, %4, %5, %0\n v_mad_i64_i32 %0, vcc, %4, %5, %0\n v_mad_i64_i32 %0, vcc, %4, %5, %0\n v_mad_i64_i32 %0, vcc, %4, %5, %0\n v_mad_i64_i32 %0, vcc, %4, %5, %0\n v_mad_i64_i32 %0, vcc, %4, %5, %0\n v_mad_i64_i32 %0, vcc, %4, %5, %0\n v_mad_i64_i32 %0, vcc, %4, %5, %0\n v_mad_i64_i32 %0, vcc, %4, %5, %0\n v_mad_i64_i32 %0, vcc, %4, %5, %0\n v_mad_i64_i32 %0, vcc, %4, %5, %0\n v_mad_i64_i32 %0, vcc, %4, %5, %0\n v_mad_i64_i32 %0, vcc, %4, %5, %0\n v_mad_i64_i32 %0, vcc, %4, %5, %0\n v_mad_i64_i32 %0, vcc, %4, %5, %0\n v_mad_i64_i32 %0, vcc, %4, %5, %0\n v_mad_i64_i32 %0, vcc, %4, %5, %0\n v_mad_i64_i32 %0, vcc, %4, %5, %0\n v_mad_i64_i32 %0, vcc, %4, %5, %0\n v_mad_i64_i32 %0, vcc, %4, %5, %0\n v_mad_i64_i32 %0, vcc, %4, %5, %0\n v_mad_i64_i32 %0, vcc, %4, %5, %0\n v_mad_i64_i32 %0, vcc, %4, %5, %0\n v_mad_i64_i32 %0, vcc, %4, %5, %0\n v_mad_i64_i32 %0, vcc, %4, %5, %0\n v_mad_i64_i32 %0, vcc, %4, %5, %0\n v_mad_i64_i32 %0, vcc, %4, %5, %0\n v_mad_i64_i32 %0, vcc, %4, %5, %0\n v_mad_i64_i32 %0, vcc, %4, %5, %0\n v_mad_i64_i32 %0, vcc, %4, %5, %0" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(c), "v"(d) : "vcc");
        } else if constexpr (KIND == 17) {  // ... with an s_nop 0 after every fourth
            asm volatile("v_mad_i64_i32 %0, vcc, %4, %5, %0\n v_mad_i64_i32 %0, vcc, %4, %5, %0\n v_mad_i64_i32 %0, vcc, %4, %5, %0\n v_mad_i64_i32 %0, vcc, %4, %5, %0\n s_nop 0\n v_mad_i64_i32 %0, vcc, %4, %5, %0\n v_mad_i64_i32 %0, vcc, %4, %5, %0\n v_mad_i64_i32 %0, vcc, %4, %5, %0\n v_mad_i64_i32 %0, vcc, %4, %5, %0\n s_nop 0\n v_mad_i64_i32 %0, vcc, %4, %5, %0\n v_mad_i64_i32 %0, vcc, %4, %5, %0\n v_mad_i64_i32 %0, vcc, %4, %5, %0\n v_mad_i64_i32 %0, vcc, %4, %5, %0\n s_nop 0\n v_mad_i64_i32 %0, vcc, %4, %5, %0\n v_mad_i64_i32 %0, vcc, %4, %5, %0\n v_mad_i64_i32 %0, vcc, %4, %5, %0\n v_mad_i64_i32 %0, vcc, %4, %5, %0\n s_nop 0\n v_mad_i64_i32 %0, vcc, %4, %5, %0\n v_mad_i64_i32 %0, vcc, %4, %5, %0\n v_mad_i64_i32 %0, vcc, %4, %5, %0\n v_mad_i64_i32 %0, vcc, %4, %5, %0\n s_nop 0\n v_mad_i64_i32 %0, vcc, %4, %5, %0\n v_mad_i64_i32 %0, vcc, %4, %5, %0\n v_mad_i64_i32 %0, vcc, %4, %5, %0\n v_mad_i64_i32 %0, vcc, %4, %5, %0\n s_nop 0\n v_mad_i64_i32 %0, vcc, %4, %5, %0\n v_mad_i64_i32 %0, vcc, %4, %5, %0\n v_mad_i64_i32 %0, vcc, %4, %5, %0\n v_mad_i64_i32 %0, vcc, %4, %5, %0\n s_nop 0\n v_mad_i64_i32 %0, vcc, %4, %5, %0\n v_mad_i64_i32 %0, vcc, %4, %5, %0\n v_mad_i64_i32 %0, vcc, %4, %5, %0\n v_mad_i64_i32 %0, vcc, %4, %5, %0\n s_nop 0\n v_mad_i64_i32 %0, vcc, %4, %5, %0\n v_mad_i64_i32 %0, vcc, %4, %5, %0\n v_mad_i64_i32 %0, vcc, %4, %5, %0\n v_mad_i64_i32 %0, vcc, %4, %5, %0\n s_nop 0\n v_mad_i64_i32 %0, vcc, %4, %5, %0\n v_mad_i64_i32 %0, vcc, %4, %5, %0\n v_mad_i64_i32 %0, vcc, %4, %5, %0\n v_mad_i64_i32 %0, vcc, %4, %5, %0\n s_nop 0\n v_mad_i64_i32 %0, vcc, %4, %5, %0\n v_mad_i64_i32 %0, vcc, %4, %5, %0\n v_mad_i64_i32 %0, vcc, %4, %5, %0\n v_mad_i64_i32 %0, vcc, %4, %5, %0\n s_nop 0\n v_mad_i64_i32 %0, vcc, %4, %5, %0\n v_mad_i64_i32 %0, vcc, %4, %5, %0\n v_mad_i64_i32 %0, vcc, %4, %5, %0\n v_mad_i64_i32 %0, vcc, %4, %5, %0\n s_nop 0\n v_mad_i64_i32 %0, vcc, %4, %5, %0\n v_mad_i64_i32 %0, vcc, %4, %5, %0\n v_mad_i64_i32 %0, vcc, %4, %5, %0\n v_mad_i64_i32 %0, vcc, %4, %5, %0\n s_nop 0\n v_mad_i64_i32 %0, vcc, %4, %5, %0\n v_mad_i64_i32 %0, vcc, %4, %5, %0\n v_mad_i64_i32 %0, vcc, %4, %5, %0\n v_mad_i64_i32 %0, vcc, %4, %5, %0\n s_nop 0\n v_mad_i64_i32 %0, vcc, %4, %5, %0\n v_mad_i64_i32 %0, vcc, %4, %5, %0\n v_mad_i64_i32 %0, vcc, %4, %5, %0\n v_mad_i64_i32 %0, vcc, %4, %5, %0\n s_nop 0\n v_mad_i64_i32 %0, vcc, %4, %5, %0\n v_mad_i64_i32 %0, vcc, %4, %5, %0\n v_mad_i64_i32 %0, vcc, %4, %5, %0\n v_mad_i64_i32 %0, vcc, %4, %5, %0" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(c), "v"(d) : "vcc");
        } else if constexpr (KIND == 18) {  // 64 dependent FP64 FMAs in one statement
            asm volatile("v_fma_f64 %0, %4, %5, %0\n v_fma_f64 %0, %4, %5, %0\n v_fma_f64 %0, %4, %5, %0\n v_fma_f64 %0, %4, %5, %0\n v_fma_f64 %0, %4, %5, %0\n v_fma_f64 %0, %4, %5, %0\n v_fma_f64 %0, %4, %5, %0\n v_fma_f64 %0, %4, %5, %0\n v_fma_f64 %0, %4, %5, %0\n v_fma_f64 %0, %4, %5, %0\n v_fma_f64 %0, %4, %5, %0\n v_fma_f64 %0, %4, %5, %0\n v_fma_f64 %0, %4, %5, %0\n v_fma_f64 %0, %4, %5, %0\n v_fma_f64 %0, %4, %5, %0\n v_fma_f64 %0, %4, %5, %0\n v_fma_f64 %0, %4, %5, %0\n v_fma_f64 %0, %4, %5, %0\n v_fma_f64 %0, %4, %5, %0\n v_fma_f64 %0, %4, %5, %0\n v_fma_f64 %0, %4, %5, %0\n v_fma_f64 %0, %4, %5, %0\n v_fma_f64 %0, %4, %5, %0\n v_fma_f64 %0, %4, %5, %0\n v_fma_f64 %0, %4, %5, %0\n v_fma_f64 %0, %4, %5, %0\n v_fma_f64 %0, %4, %5, %0\n v_fma_f64 %0, %4, %5, %0\n v_fma_f64 %0, %4, %5, %0\n v_fma_f64 %0, %4, %5, %0\n v_fma_f64 %0, %4, %5, %0\n v_fma_f64 %0, %4, %5, %0\n v_fma_f64 %0, %4, %5, %0\n v_fma_f64 %0, %4, %5, %0\n v_fma_f64 %0, %4, %5, %0\n v_fma_f64 %0, %4, %5, %0\n v_fma_f64 %0, %4, %5, %0\n v_fma_f64 %0, %4, %5, %0\n v_fma_f64 %0, %4, %5, %0\n v_fma_f64 %0, %4, %5, %0\n v_fma_f64 %0, %4, %5, %0\n v_fma_f64 %0, %4, %5, %0\n v_fma_f64 %0, %4, %5, %0\n v_fma_f64 %0, %4, %5, %0\n v_fma_f64 %0, %4, %5, %0\n v_fma_f64 %0, %4, %5, %0\n v_fma_f64 %0, %4, %5, %0\n v_fma_f64 %0, %4, %5, %0\n v_fma_f64 %0, %4, %5, %0\n v_fma_f64 %0, %4, %5, %0\n v_fma_f64 %0, %4, %5, %0\n v_fma_f64 %0, %4, %5, %0\n v_fma_f64 %0, %4, %5, %0\n v_fma_f64 %0, %4, %5, %0\n v_fma_f64 %0, %4, %5, %0\n v_fma_f64 %0, %4, %5, %0\n v_fma_f64 %0, %4, %5, %0\n v_fma_f64 %0, %4, %5, %0\n v_fma_f64 %0, %4, %5, %0\n v_fma_f64 %0, %4, %5, %0\n v_fma_f64 %0, %4, %5, %0\n v_fma_f64 %0, %4, %5, %0\n v_fma_f64 %0, %4, %5, %0\n v_fma_f64 %0, %4, %5, %0" : "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3) : "v"(fc), "v"(fd));
        } else if constexpr (KIND == 19) {  // ... with an s_nop 0 after every fourth
            asm volatile("v_fma_f64 %0, %4, %5, %0\n v_fma_f64 %0, %4, %5, %0\n v_fma_f64 %0, %4, %5, %0\n v_fma_f64 %0, %4, %5, %0\n s_nop 0\n v_fma_f64 %0, %4, %5, %0\n v_fma_f64 %0, %4, %5, %0\n v_fma_f64 %0, %4, %5, %0\n v_fma_f64 %0, %4, %5, %0\n s_nop 0\n v_fma_f64 %0, %4, %5, %0\n v_fma_f64 %0, %4, %5, %0\n v_fma_f64 %0, %4, %5, %0\n v_fma_f64 %0, %4, %5, %0\n s_nop 0\n v_fma_f64 %0, %4, %5, %0\n v_fma_f64 %0, %4, %5, %0\n v_fma_f64 %0, %4, %5, %0\n v_fma_f64 %0, %4, %5, %0\n s_nop 0\n v_fma_f64 %0, %4, %5, %0\n v_fma_f64 %0, %4, %5, %0\n v_fma_f64 %0, %4, %5, %0\n v_fma_f64 %0, %4, %5, %0\n s_nop 0\n v_fma_f64 %0, %4, %5, %0\n v_fma_f64 %0, %4, %5, %0\n v_fma_f64 %0, %4, %5, %0\n v_fma_f64 %0, %4, %5, %0\n s_nop 0\n v_fma_f64 %0, %4, %5, %0\n v_fma_f64 %0, %4, %5, %0\n v_fma_f64 %0, %4, %5, %0\n v_fma_f64 %0, %4, %5, %0\n s_nop 0\n v_fma_f64 %0, %4, %5, %0\n v_fma_f64 %0, %4, %5, %0\n v_fma_f64 %0, %4, %5, %0\n v_fma_f64 %0, %4, %5, %0\n s_nop 0\n v_fma_f64 %0, %4, %5, %0\n v_fma_f64 %0, %4, %5, %0\n v_fma_f64 %0, %4, %5, %0\n v_fma_f64 %0, %4, %5, %0\n s_nop 0\n v_fma_f64 %0, %4, %5, %0\n v_fma_f64 %0, %4, %5, %0\n v_fma_f64 %0, %4, %5, %0\n v_fma_f64 %0, %4, %5, %0\n s_nop 0\n v_fma_f64 %0, %4, %5, %0\n v_fma_f64 %0, %4, %5, %0\n v_fma_f64 %0, %4, %5, %0\n v_fma_f64 %0, %4, %5, %0\n s_nop 0\n v_fma_f64 %0, %4, %5, %0\n v_fma_f64 %0, %4, %5, %0\n v_fma_f64 %0, %4, %5, %0\n v_fma_f64 %0, %4, %5, %0\n s_nop 0\n v_fma_f64 %0, %4, %5, %0\n v_fma_f64 %0, %4, %5, %0\n v_fma_f64 %0, %4, %5, %0\n v_fma_f64 %0, %4, %5, %0\n s_nop 0\n v_fma_f64 %0, %4, %5, %0\n v_fma_f64 %0, %4, %5, %0\n v_fma_f64 %0, %4, %5, %0\n v_fma_f64 %0, %4, %5, %0\n s_nop 0\n v_fma_f64 %0, %4, %5, %0\n v_fma_f64 %0, %4, %5, %0\n v_fma_f64 %0, %4, %5, %0\n v_fma_f64 %0, %4, %5, %0\n s_nop 0\n v_fma_f64 %0, %4, %5, %0\n v_fma_f64 %0, %4, %5, %0\n v_fma_f64 %0, %4, %5, %0\n v_fma_f64 %0, %4, %5, %0" : "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3) : "v"(fc), "v"(fd));
        } else if constexpr (KIND == 20) {  // v_fmac_f64 (the VOP2 form the compiler picks), 64 dependent
            asm volatile("v_fmac_f64 %0, %4, %5\n v_fmac_f64 %0, %4, %5\n v_fmac_f64 %0, %4, %5\n v_fmac_f64 %0, %4, %5\n v_fmac_f64 %0, %4, %5\n v_fmac_f64 %0, %4, %5\n v_fmac_f64 %0, %4, %5\n v_fmac_f64 %0, %4, %5\n v_fmac_f64 %0, %4, %5\n v_fmac_f64 %0, %4, %5\n v_fmac_f64 %0, %4, %5\n v_fmac_f64 %0, %4, %5\n v_fmac_f64 %0, %4, %5\n v_fmac_f64 %0, %4, %5\n v_fmac_f64 %0, %4, %5\n v_fmac_f64 %0, %4, %5\n v_fmac_f64 %0, %4, %5\n v_fmac_f64 %0, %4, %5\n v_fmac_f64 %0, %4, %5\n v_fmac_f64 %0, %4, %5\n v_fmac_f64 %0, %4, %5\n v_fmac_f64 %0, %4, %5\n v_fmac_f64 %0, %4, %5\n v_fmac_f64 %0, %4, %5\n v_fmac_f64 %0, %4, %5\n v_fmac_f64 %0, %4, %5\n v_fmac_f64 %0, %4, %5\n v_fmac_f64 %0, %4, %5\n v_fmac_f64 %0, %4, %5\n v_fmac_f64 %0, %4, %5\n v_fmac_f64 %0, %4, %5\n v_fmac_f64 %0, %4, %5\n v_fmac_f64 %0, %4, %5\n v_fmac_f64 %0, %4, %5\n v_fmac_f64 %0, %4, %5\n v_fmac_f64 %0, %4, %5\n v_fmac_f64 %0, %4, %5\n v_fmac_f64 %0, %4, %5\n v_fmac_f64 %0, %4, %5\n v_fmac_f64 %0, %4, %5\n v_fmac_f64 %0, %4, %5\n v_fmac_f64 %0, %4, %5\n v_fmac_f64 %0, %4, %5\n v_fmac_f64 %0, %4, %5\n v_fmac_f64 %0, %4, %5\n v_fmac_f64 %0, %4, %5\n v_fmac_f64 %0, %4, %5\n v_fmac_f64 %0, %4, %5\n v_fmac_f64 %0, %4, %5\n v_fmac_f64 %0, %4, %5\n v_fmac_f64 %0, %4, %5\n v_fmac_f64 %0, %4, %5\n v_fmac_f64 %0, %4, %5\n v_fmac_f64 %0, %4, %5\n v_fmac_f64 %0, %4, %5\n v_fmac_f64 %0, %4, %5\n v_fmac_f64 %0, %4, %5\n v_fmac_f64 %0, %4, %5\n v_fmac_f64 %0, %4, %5\n v_fmac_f64 %0, %4, %5\n v_fmac_f64 %0, %4, %5\n v_fmac_f64 %0, %4, %5\n v_fmac_f64 %0, %4, %5\n v_fmac_f64 %0, %4, %5" : "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3) : "v"(fc), "v"(fd));
        } else {
            REP8(asm volatile("v_mad_u64_u32 %0, vcc, %4, %5, %0\n v_mad_u64_u32 %1, vcc, %4, %5, %1\n v_mad_u64_u32 %2, vcc, %4, %5, %2\n v_mad_u64_u32 %3, vcc, %4, %5, %3\n"
                              "v_mad_u64_u32 %0, vcc, %4, %5, %0\n v_mad_u64_u32 %1, vcc, %4, %5, %1\n v_mad_u64_u32 %2, vcc, %4, %5, %2\n v_mad_u64_u32 %3, vcc, %4, %5, %3"
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(c), "v"(d) : "vcc");)
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + (long long)(f0 + f1 + f2 + f3);
}

template <int KIND>
void run64(const char *name, long long *d_out, int cus, int big = 0, int iters = 20000) {
    printf("%-28s", name);
    for (int wps : {1, 2, 3, 4}) {
        const int blocks = cus * wps;
        hipEvent_t e0, e1;
        hipEventCreate(&e0);
        hipEventCreate(&e1);
        hipLaunchKernelGGL(k64<KIND>, dim3(blocks), dim3(256), 0, 0, d_out, iters, 1, big);
        hipEventRecord(e0);
        hipLaunchKernelGGL(k64<KIND>, dim3(blocks), dim3(256), 0, 0, d_out, iters, 1, big);
        hipEventRecord(e1);
        hipDeviceSynchronize();
        float ms = 0;
        hipEventElapsedTime(&ms, e0, e1);
        printf("  %d w/SIMD: %.2f ns", wps, ms * 1e6 / ((double)iters * 64 * wps));
    }
    printf("   (SIMD time per wave-instruction)\n");
}

template <int KIND>
void run(const char *name, int *d_out, int cus) {
    const int iters = 20000;
    printf("%-28s", name);
    for (int wps : {1, 2, 3, 4}) {
        const int blocks = cus * wps;
        hipEvent_t e0, e1;
        hipEventCreate(&e0);
        hipEventCreate(&e1);
        hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(256), 0, 0, d_out, iters, 1);
        hipEventRecord(e0);
        hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(256), 0, 0, d_out, iters, 1);
        hipEventRecord(e1);
        hipDeviceSynchronize();
        float ms = 0;
        hipEventElapsedTime(&ms, e0, e1);
        printf("  %d w/SIMD: %.2f ns", wps, ms * 1e6 / ((double)iters * 64 * wps));
    }
    printf("   (SIMD time per wave-instruction)\n");
}

int main() {
    hipDeviceProp_t p;
    hipGetDeviceProperties(&p, 0);
    printf("%s: %d CUs, nominal %.2f GHz\n", p.name, p.multiProcessorCount, p.clockRate / 1e6);
    int *d;
    hipMalloc(&d, 256 * 4 * 256 * 16 * sizeof(int));
    const int cus = p.multiProcessorCount;
    run<14>("v_add_f32 (for scale)", d, cus);
    run<0>("v_sub_u32", d, cus);
    run<13>("v_sub_u32 inline const", d, cus);
    run<1>("v_xor_b32", d, cus);
    run<8>("v_ashrrev_i32", d, cus);
    run<9>("v_cndmask_b32 vcc", d, cus);
    run<2>("v_mul_i32_i24", d, cus);
    run<3>("v_mad_i32_i24", d, cus);
    run<16>("v_mad_i32_i24 inline const", d, cus);
    run<11>("v_mad_u32_u24", d, cus);
    run<4>("v_mul_lo_u32", d, cus);
    run<5>("v_med3_i32", d, cus);
    run<6>("v_add3_u32", d, cus);
    run<7>("v_lshl_add_u32", d, cus);
    run<10>("v_bfi_b32", d, cus);
    run<12>("v_xad_u32", d, cus);
    run<15>("v_and_or_b32", d, cus);
    run<17>("v_cndmask_b32 s[20:21]", d, cus);
    run<18>("v_cmp vcc + v_cndmask (x2)", d, cus);
    run<19>("v_cmp sgpr + v_cndmask (x2)", d, cus);
    run<20>("v_cmp_gt_i32 sgpr pair", d, cus);
    run<21>("v_cmp_gt_i32 vcc", d, cus);
    run<22>("v_and_b32", d, cus);
    run<23>("v_cvt_f32_i32", d, cus);
    run<24>("v_cvt_i32_f32", d, cus);
    run<25>("v_mul_f32", d, cus);
    run<26>("v_max_i32", d, cus);
    run<27>("v_lshlrev_b32 const", d, cus);
    run<28>("v_mul_hi_u32", d, cus);
    run<29>("v_ffbh_u32", d, cus);
    run<30>("v_cvt_f32_ubyte0", d, cus);
    run<31>("v_rcp_f32", d, cus);
    run<32>("v_floor_f32", d, cus);
    run<33>("v_trunc_f32", d, cus);
    run<35>("v_perm_b32", d, cus);
    long long *d64;
    hipMalloc(&d64, 256 * 4 * 256 * 16 * sizeof(long long));
    run64<0>("v_mad_i64_i32 small operands", d64, cus);
    run64<0>("v_mad_i64_i32 31-bit operands", d64, cus, 1);
    run64<4>("v_mad_u64_u32 31-bit operands", d64, cus, 1);
    run64<4>("v_mad_u64_u32", d64, cus);
    run64<11>("v_mad_i64_i32 -> s[20:21]", d64, cus, 1);
    run64<12>("... ONE chain -> s[20:21]", d64, cus, 1);
    run64<13>("add_co+addc_co via s[20:21]", d64, cus, 1);
    run64<14>("add_co+addc_co via vcc", d64, cus, 1);
    run64<15>("v_lshl_add_u64", d64, cus, 1);
    run64<16>("mad_i64 x64 dep., 30 ms kernels", d64, cus, 1, 250000);
    run64<18>("fma_f64 x64 dep., 30 ms kernels", d64, cus, 0, 250000);
    run64<16>("mad_i64 x64 dependent, 1 stmt", d64, cus, 1);
    run64<17>("... s_nop after every 4th", d64, cus, 1);
    run64<18>("fma_f64 x64 dependent, 1 stmt", d64, cus);
    run64<19>("... s_nop after every 4th", d64, cus);
    run64<20>("fmac_f64 x64 dependent", d64, cus);
    run64<5>("v_mad_i64_i32 ONE chain", d64, cus, 1);
    run64<6>("v_mad_i64_i32 two chains", d64, cus, 1);
    run64<7>("v_fma_f64 ONE chain", d64, cus);
    run64<9>("v_fma_f64 chain, 16 operands", d64, cus);
    run64<10>("v_mad_i64_i32 chain, 16 oper.", d64, cus);
    run64<8>("v_fma_f64 two chains", d64, cus);
    run64<1>("v_fma_f64", d64, cus);
    run64<2>("v_add_f64", d64, cus);
    run64<3>("v_cvt_f64_i32", d64, cus);
    return 0;
}
