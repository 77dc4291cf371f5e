// Micro-benchmark (development tool): what a wave64 f32 VALU instruction costs in REAL shader cycles, and what clock the part
// runs at under a pure VALU load.  s_memtime counts shader-clock cycles, s_memrealtime a constant 100 MHz clock: their ratio over a
// long VALU loop is the sustained clock; the loop's s_memtime delta / instructions issued per SIMD is the issue cost in cycles.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/valu_clock tools/ubench/valu_clock.hip && /tmp/valu_clock
#include <hip/hip_runtime.h>

#include <cstdio>
#include <vector>

#define REP8(x) x x x x x x x x

template <int KIND>
__global__ void k(float *out, unsigned long long *stamps, int iters, float seed) {
    const float s = seed + threadIdx.x;
    float a0 = s, a1 = s + 1, a2 = s + 2, a3 = s + 3, a4 = s + 4, a5 = s + 5, a6 = s + 6, a7 = s + 7, c = 1.0000001f;
    typedef float v2f __attribute__((ext_vector_type(2)));
    v2f p0 = {s, s + 1}, p1 = {s + 2, s + 3}, p2 = {s + 4, s + 5}, p3 = {s + 6, s + 7}, pc = {c, c};
    const unsigned long long t0 = __builtin_readcyclecounter(), w0 = wall_clock64();
    for (int it = 0; it < iters; ++it) {
        if constexpr (KIND == 0) {
            REP8(asm volatile("v_mul_f32 %0, %0, %8\n v_add_f32 %1, %1, %8\n v_mul_f32 %2, %2, %8\n v_add_f32 %3, %3, %8\n v_mul_f32 %4, %4, %8\n v_add_f32 %5, %5, %8\n v_mul_f32 %6, %6, %8\n v_add_f32 %7, %7, %8"
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c));)
        } else if constexpr (KIND == 1) {  // 32-bit literal operands (8-byte encodings), as in the transforms' constant multiplies
            REP8(asm volatile("v_mul_f32 %0, 0x3f7fff00, %0\n v_add_f32 %1, 0x33000000, %1\n v_mul_f32 %2, 0x3f7fff01, %2\n v_add_f32 %3, 0x33000001, %3\n v_mul_f32 %4, 0x3f7fff02, %4\n v_add_f32 %5, 0x33000002, %5\n v_mul_f32 %6, 0x3f7fff03, %6\n v_add_f32 %7, 0x33000003, %7"
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c));)
        } else if constexpr (KIND == 3) {  // every VALU instruction followed by an independent scalar one: 128 instructions per block
            REP8(asm volatile("v_mul_f32 %0, %0, %8\n s_add_u32 s20, s20, 1\n v_add_f32 %1, %1, %8\n s_add_u32 s21, s21, 1\n v_mul_f32 %2, %2, %8\n s_add_u32 s22, s22, 1\n v_add_f32 %3, %3, %8\n s_add_u32 s23, s23, 1\n"
                              "v_mul_f32 %4, %4, %8\n s_add_u32 s20, s20, 1\n v_add_f32 %5, %5, %8\n s_add_u32 s21, s21, 1\n v_mul_f32 %6, %6, %8\n s_add_u32 s22, s22, 1\n v_add_f32 %7, %7, %8\n s_add_u32 s23, s23, 1"
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c) : "s20", "s21", "s22", "s23", "scc");)
        } else if constexpr (KIND == 4) {  // every VALU instruction followed by s_nop 0
            REP8(asm volatile("v_mul_f32 %0, %0, %8\n s_nop 0\n v_add_f32 %1, %1, %8\n s_nop 0\n v_mul_f32 %2, %2, %8\n s_nop 0\n v_add_f32 %3, %3, %8\n s_nop 0\n"
                              "v_mul_f32 %4, %4, %8\n s_nop 0\n v_add_f32 %5, %5, %8\n s_nop 0\n v_mul_f32 %6, %6, %8\n s_nop 0\n v_add_f32 %7, %7, %8\n s_nop 0"
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c));)
        } else if constexpr (KIND == 5) {  // v_mov between the arithmetic instructions (register shuffling)
            REP8(asm volatile("v_mul_f32 %0, %0, %8\n v_mov_b32 %1, %0\n v_add_f32 %2, %2, %8\n v_mov_b32 %3, %2\n v_mul_f32 %4, %4, %8\n v_mov_b32 %5, %4\n v_add_f32 %6, %6, %8\n v_mov_b32 %7, %6"
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c));)
        } else if constexpr (KIND == 6) {  // packed f32: two lanes' worth per instruction (register pairs)
            REP8(asm volatile("v_pk_mul_f32 %0, %0, %4\n v_pk_add_f32 %1, %1, %4\n v_pk_mul_f32 %2, %2, %4\n v_pk_add_f32 %3, %3, %4\n v_pk_mul_f32 %0, %0, %4\n v_pk_add_f32 %1, %1, %4\n v_pk_mul_f32 %2, %2, %4\n v_pk_add_f32 %3, %3, %4"
                              : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(pc));)
        } else if constexpr (KIND == 7) {  // v_fma_f32 (the contract forbids it; for scale)
            REP8(asm volatile("v_fma_f32 %0, %0, %8, %8\n v_fma_f32 %1, %1, %8, %8\n v_fma_f32 %2, %2, %8, %8\n v_fma_f32 %3, %3, %8, %8\n v_fma_f32 %4, %4, %8, %8\n v_fma_f32 %5, %5, %8, %8\n v_fma_f32 %6, %6, %8, %8\n v_fma_f32 %7, %7, %8, %8"
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c));)
        } else {  // one dependent chain
            REP8(asm volatile("v_mul_f32 %0, %0, %8\n v_add_f32 %0, %0, %8\n v_mul_f32 %0, %0, %8\n v_add_f32 %0, %0, %8\n v_mul_f32 %0, %0, %8\n v_add_f32 %0, %0, %8\n v_mul_f32 %0, %0, %8\n v_add_f32 %0, %0, %8"
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c));)
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter(), w1 = wall_clock64();
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + p0.x + p0.y + p1.x + p1.y + p2.x + p2.y + p3.x + p3.y;
    if ((threadIdx.x & 63) == 0) {
        const unsigned wv = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
        unsigned hw;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
        stamps[4 * wv + 0] = t1 - t0;
        stamps[4 * wv + 1] = w1 - w0;
        stamps[4 * wv + 2] = w0;
        stamps[4 * wv + 3] = hw;
    }
}

template <int KIND>
void run(const char *name, float *d_out, unsigned long long *d_st, int cus, int per_block = 64) {
    for (int iters : {2000, 20000}) {
        for (int wps : {1, 2, 3, 4}) {
            const int blocks = cus * wps;  // 256-thread blocks: one wavefront per SIMD each
            hipEvent_t e0, e1;
            hipEventCreate(&e0);
            hipEventCreate(&e1);
            hipEventRecord(e0);
            hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(256), 0, 0, d_out, d_st, iters, 1.0f);
            hipEventRecord(e1);
            hipDeviceSynchronize();
            float ms = 0;
            hipEventElapsedTime(&ms, e0, e1);
            const int nw = blocks * 4;
            std::vector<unsigned long long> st(4 * (size_t)nw);
            hipMemcpy(st.data(), d_st, st.size() * 8, hipMemcpyDeviceToHost);
            double cyc = 0, ticks = 0;
            unsigned long long first = ~0ull, last_start = 0, cmax = 0;
            for (int w = 0; w < nw; ++w) {
                cyc += st[4 * w];
                ticks += st[4 * w + 1];
                if (st[4 * w] > cmax) cmax = st[4 * w];
                if (st[4 * w + 2] < first) first = st[4 * w + 2];
                if (st[4 * w + 2] > last_start) last_start = st[4 * w + 2];
            }
            cyc /= nw;
            ticks /= nw;
            const double instr = (double)iters * per_block;
            printf("%-20s iters=%-6d waves/SIMD=%d  kernel %.3f ms; per wavefront: %.0f shader cycles (max %llu) in %.1f us -> %.3f GHz, %.2f cycles per instruction; "
                   "last wavefront started %.1f us after the first; SIMD rate %.2f ns per wave-instruction\n",
                   name, iters, wps, ms, cyc, cmax, ticks / 100.0, cyc / (ticks * 10.0), cyc / instr, (last_start - first) / 100.0,
                   ms * 1e6 / (instr * wps));
        }
    }
}

int main() {
    hipDeviceProp_t p;
    hipGetDeviceProperties(&p, 0);
    printf("%s: %d CUs, nominal %.2f GHz\n", p.name, p.multiProcessorCount, p.clockRate / 1e6);
    float *d;
    unsigned long long *st;
    hipMalloc(&d, 256 * 4 * 256 * 16 * sizeof(float));
    hipMalloc(&st, 256 * 4 * 4 * 4 * 32);
    run<0>("mul/add registers", d, st, p.multiProcessorCount);
    run<1>("mul/add literals", d, st, p.multiProcessorCount);
    run<2>("one dependent chain", d, st, p.multiProcessorCount);
    run<3>("VALU + SALU pairs", d, st, p.multiProcessorCount, 128);
    run<4>("VALU + s_nop pairs", d, st, p.multiProcessorCount, 128);
    run<5>("mul/add + v_mov", d, st, p.multiProcessorCount);
    run<6>("v_pk_mul/add_f32", d, st, p.multiProcessorCount);
    run<7>("v_fma_f32", d, st, p.multiProcessorCount);
    return 0;
}
