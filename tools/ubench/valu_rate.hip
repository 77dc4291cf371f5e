// Micro-benchmark (development tool): issue rate of the VALU instructions the synthesis kernels lean on,
// wave64 on gfx950.  Build + run on the GPU box:
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/valu_rate tools/ubench/valu_rate.hip && /tmp/valu_rate
// Prints cycles per wavefront-instruction per SIMD at 1, 2 and 4 resident wavefronts per SIMD.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <vector>

typedef float v2f __attribute__((ext_vector_type(2)));

#define REP8(x) x x x x x x x x
#define BODY(asm_text)                                                                              \
    for (int it = 0; it < iters; ++it) {                                                            \
        REP8(asm volatile(asm_text : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c));) \
    }

template <int KIND>
__global__ void k(float *out, int iters, float seed) {
    const float s = seed + threadIdx.x;
    if constexpr (KIND == 0 || KIND == 2 || KIND == 5 || KIND == 6) {
        float a0 = s, a1 = s + 1, a2 = s + 2, a3 = s + 3, a4 = s + 4, a5 = s + 5, a6 = s + 6, a7 = s + 7, c = 1.0000001f;
        if constexpr (KIND == 0) BODY("v_mul_f32 %0, %0, %8\n v_mul_f32 %1, %1, %8\n v_mul_f32 %2, %2, %8\n v_mul_f32 %3, %3, %8\n v_mul_f32 %4, %4, %8\n v_mul_f32 %5, %5, %8\n v_mul_f32 %6, %6, %8\n v_mul_f32 %7, %7, %8")
        if constexpr (KIND == 2) BODY("v_add_f32 %0, %0, %8\n v_add_f32 %1, %1, %8\n v_add_f32 %2, %2, %8\n v_add_f32 %3, %3, %8\n v_add_f32 %4, %4, %8\n v_add_f32 %5, %5, %8\n v_add_f32 %6, %6, %8\n v_add_f32 %7, %7, %8")
        if constexpr (KIND == 5) BODY("v_mov_b32 %0, %1\n v_mov_b32 %1, %2\n v_mov_b32 %2, %3\n v_mov_b32 %3, %4\n v_mov_b32 %4, %5\n v_mov_b32 %5, %6\n v_mov_b32 %6, %7\n v_mov_b32 %7, %8")
        if constexpr (KIND == 6) BODY("v_fma_f32 %0, %0, %8, %8\n v_fma_f32 %1, %1, %8, %8\n v_fma_f32 %2, %2, %8, %8\n v_fma_f32 %3, %3, %8, %8\n v_fma_f32 %4, %4, %8, %8\n v_fma_f32 %5, %5, %8, %8\n v_fma_f32 %6, %6, %8, %8\n v_fma_f32 %7, %7, %8, %8")
        out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
    } else if constexpr (KIND == 1 || KIND == 3 || KIND == 7 || KIND == 8) {
        v2f a0 = {s, s}, a1 = a0 + 1.f, a2 = a0 + 2.f, a3 = a0 + 3.f, a4 = a0 + 4.f, a5 = a0 + 5.f, a6 = a0 + 6.f, a7 = a0 + 7.f;
        v2f c = {1.0000001f, 0.9999999f};
        if constexpr (KIND == 1) BODY("v_pk_mul_f32 %0, %0, %8\n v_pk_mul_f32 %1, %1, %8\n v_pk_mul_f32 %2, %2, %8\n v_pk_mul_f32 %3, %3, %8\n v_pk_mul_f32 %4, %4, %8\n v_pk_mul_f32 %5, %5, %8\n v_pk_mul_f32 %6, %6, %8\n v_pk_mul_f32 %7, %7, %8")
        if constexpr (KIND == 3) BODY("v_pk_add_f32 %0, %0, %8\n v_pk_add_f32 %1, %1, %8\n v_pk_add_f32 %2, %2, %8\n v_pk_add_f32 %3, %3, %8\n v_pk_add_f32 %4, %4, %8\n v_pk_add_f32 %5, %5, %8\n v_pk_add_f32 %6, %6, %8\n v_pk_add_f32 %7, %7, %8")
        if constexpr (KIND == 7) BODY("v_pk_add_f32 %0, %0, %8 op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[0,1]\n v_pk_add_f32 %1, %1, %8 op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[0,1]\n v_pk_add_f32 %2, %2, %8 op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[0,1]\n v_pk_add_f32 %3, %3, %8 op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[0,1]\n v_pk_add_f32 %4, %4, %8 op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[0,1]\n v_pk_add_f32 %5, %5, %8 op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[0,1]\n v_pk_add_f32 %6, %6, %8 op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[0,1]\n v_pk_add_f32 %7, %7, %8 op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[0,1]")
        if constexpr (KIND == 8) BODY("v_pk_fma_f32 %0, %0, %8, %8\n v_pk_fma_f32 %1, %1, %8, %8\n v_pk_fma_f32 %2, %2, %8, %8\n v_pk_fma_f32 %3, %3, %8, %8\n v_pk_fma_f32 %4, %4, %8, %8\n v_pk_fma_f32 %5, %5, %8, %8\n v_pk_fma_f32 %6, %6, %8, %8\n v_pk_fma_f32 %7, %7, %8, %8")
        const v2f r = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
        out[blockIdx.x * blockDim.x + threadIdx.x] = r.x + r.y;
    } else if constexpr (KIND >= 9) {
        int a0 = (int)s, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7, c = 0x00030005;
        if constexpr (KIND == 9) BODY("v_dot2_i32_i16 %0, %8, %8, %0\n v_dot2_i32_i16 %1, %8, %8, %1\n v_dot2_i32_i16 %2, %8, %8, %2\n v_dot2_i32_i16 %3, %8, %8, %3\n v_dot2_i32_i16 %4, %8, %8, %4\n v_dot2_i32_i16 %5, %8, %8, %5\n v_dot2_i32_i16 %6, %8, %8, %6\n v_dot2_i32_i16 %7, %8, %8, %7")
        if constexpr (KIND == 10) BODY("v_mad_i32_i24 %0, %8, %8, %0\n v_mad_i32_i24 %1, %8, %8, %1\n v_mad_i32_i24 %2, %8, %8, %2\n v_mad_i32_i24 %3, %8, %8, %3\n v_mad_i32_i24 %4, %8, %8, %4\n v_mad_i32_i24 %5, %8, %8, %5\n v_mad_i32_i24 %6, %8, %8, %6\n v_mad_i32_i24 %7, %8, %8, %7")
        if constexpr (KIND == 11) BODY("v_dot4_i32_i8 %0, %8, %8, %0\n v_dot4_i32_i8 %1, %8, %8, %1\n v_dot4_i32_i8 %2, %8, %8, %2\n v_dot4_i32_i8 %3, %8, %8, %3\n v_dot4_i32_i8 %4, %8, %8, %4\n v_dot4_i32_i8 %5, %8, %8, %5\n v_dot4_i32_i8 %6, %8, %8, %6\n v_dot4_i32_i8 %7, %8, %8, %7")
        if constexpr (KIND == 12) BODY("v_mul_lo_u32 %0, %0, %8\n v_mul_lo_u32 %1, %1, %8\n v_mul_lo_u32 %2, %2, %8\n v_mul_lo_u32 %3, %3, %8\n v_mul_lo_u32 %4, %4, %8\n v_mul_lo_u32 %5, %5, %8\n v_mul_lo_u32 %6, %6, %8\n v_mul_lo_u32 %7, %7, %8")
        out[blockIdx.x * blockDim.x + threadIdx.x] = (float)(a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7);
    } else {
        double a0 = s, a1 = s + 1, a2 = s + 2, a3 = s + 3, a4 = s + 4, a5 = s + 5, a6 = s + 6, a7 = s + 7, c = 1.0000001;
        BODY("v_fma_f64 %0, %0, %8, %8\n v_fma_f64 %1, %1, %8, %8\n v_fma_f64 %2, %2, %8, %8\n v_fma_f64 %3, %3, %8, %8\n v_fma_f64 %4, %4, %8, %8\n v_fma_f64 %5, %5, %8, %8\n v_fma_f64 %6, %6, %8, %8\n v_fma_f64 %7, %7, %8, %8")
        out[blockIdx.x * blockDim.x + threadIdx.x] = (float)(a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7);
    }
}

template <int KIND>
void run(const char *name, float *d_out, int cus, double ghz) {
    const int iters = 2000;
    for (int wps : {1, 2, 4}) {
        const int blocks = cus * wps;  // 256-thread blocks: 4 wavefronts = one per SIMD
        hipEvent_t e0, e1;
        hipEventCreate(&e0);
        hipEventCreate(&e1);
        hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(256), 0, 0, d_out, iters, 1.0f);
        hipEventRecord(e0);
        hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(256), 0, 0, d_out, iters, 1.0f);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms = 0;
        hipEventElapsedTime(&ms, e0, e1);
        const double instr_per_simd = (double)iters * 64 * wps;  // 8 asm blocks x 8 instructions
        printf("%-28s waves/SIMD=%d  %.3f ms  %.2f cycles per wave-instruction per SIMD (at %.2f GHz)\n", name, wps, ms,
               ms * 1e-3 * ghz * 1e9 / instr_per_simd, ghz);
    }
}

int main() {
    hipDeviceProp_t p;
    hipGetDeviceProperties(&p, 0);
    const double ghz = p.clockRate / 1e6;
    printf("%s: %d CUs, %.2f GHz\n", p.name, p.multiProcessorCount, ghz);
    float *d;
    hipMalloc(&d, 256 * 4 * 256 * 16 * sizeof(float));
    run<0>("v_mul_f32", d, p.multiProcessorCount, ghz);
    run<1>("v_pk_mul_f32", d, p.multiProcessorCount, ghz);
    run<2>("v_add_f32", d, p.multiProcessorCount, ghz);
    run<3>("v_pk_add_f32", d, p.multiProcessorCount, ghz);
    run<7>("v_pk_add_f32 op_sel+neg", d, p.multiProcessorCount, ghz);
    run<6>("v_fma_f32", d, p.multiProcessorCount, ghz);
    run<8>("v_pk_fma_f32", d, p.multiProcessorCount, ghz);
    run<5>("v_mov_b32", d, p.multiProcessorCount, ghz);
    run<4>("v_fma_f64", d, p.multiProcessorCount, ghz);
    run<9>("v_dot2_i32_i16", d, p.multiProcessorCount, ghz);
    run<11>("v_dot4_i32_i8", d, p.multiProcessorCount, ghz);
    run<10>("v_mad_i32_i24", d, p.multiProcessorCount, ghz);
    run<12>("v_mul_lo_u32", d, p.multiProcessorCount, ghz);
    return 0;
}
