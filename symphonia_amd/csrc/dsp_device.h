// Device-side arithmetic primitives shared by the kernels.
//
// ARITHMETIC CONTRACT: the reference never fuses a multiply with an add, so every
// a*b +/- c*d here must compile to two rounded multiplies and one rounded add/sub.  The build
// passes -ffp-contract=off (build.py) and keeps f32 denormals on (no
// -fgpu-flush-denormals-to-zero); tests/test_build.py greps the ISA for v_fma/v_mac/v_mad.
#pragma once

#include <hip/hip_runtime.h>

#include "symaccel_internal.h"

namespace symaccel {

// Pointer to a host-generated constant table, in the CONSTANT address space: the tables are written
// once before any launch, so uniform-index reads may use scalar loads (s_load_dword through the
// scalar cache into SGPRs) instead of per-lane vector loads that each occupy a VGPR.
using cf32p = const __attribute__((address_space(4))) float *;
__device__ __forceinline__ cf32p as_const(const float *p) {
    return (cf32p)p;  // deliberate address-space cast (global -> constant), same 64-bit representation
}

// Explicit counter waits around LDS-DMA loads (global_load_lds): hipcc neither orders a ds_read behind a pending LDS-DMA
// write nor an LDS-DMA write behind pending ds_reads of the same bytes -- both waits are the kernel's.
#if defined(__HIP_DEVICE_COMPILE__)
#define SYM_WAIT_VMCNT0() asm volatile("s_waitcnt vmcnt(0)" ::: "memory")
#define SYM_WAIT_LGKMCNT0() asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory")
#else
#define SYM_WAIT_VMCNT0() ((void)0)
#define SYM_WAIT_LGKMCNT0() ((void)0)
#endif

// Streaming accesses.  The batch is read once and written once, so the big loads and stores carry the non-temporal
// hint (`nt` on the global_load / global_store): measured on config 2, +4 % (0.239 -> 0.229 ms; mostly the stores,
// which otherwise allocate in the memory-side cache on their way to HBM).  SYM_NT bit 0: loads, bit 1: stores.
#ifndef SYM_NT
#define SYM_NT 3
#endif
typedef float nt_f2 __attribute__((ext_vector_type(2)));
typedef float nt_f4 __attribute__((ext_vector_type(4)));
typedef int nt_i4 __attribute__((ext_vector_type(4)));
#if defined(__HIP_DEVICE_COMPILE__) && (SYM_NT & 1)
__device__ __forceinline__ float2 ld_stream(const float2 *p) {
    const nt_f2 v = __builtin_nontemporal_load(reinterpret_cast<const nt_f2 *>(p));
    return make_float2(v.x, v.y);
}
__device__ __forceinline__ float4 ld_stream(const float4 *p) {
    const nt_f4 v = __builtin_nontemporal_load(reinterpret_cast<const nt_f4 *>(p));
    return make_float4(v.x, v.y, v.z, v.w);
}
__device__ __forceinline__ int4 ld_stream(const int4 *p) {
    const nt_i4 v = __builtin_nontemporal_load(reinterpret_cast<const nt_i4 *>(p));
    return make_int4(v.x, v.y, v.z, v.w);
}
#else
__device__ __forceinline__ float2 ld_stream(const float2 *p) { return *p; }
__device__ __forceinline__ float4 ld_stream(const float4 *p) { return *p; }
__device__ __forceinline__ int4 ld_stream(const int4 *p) { return *p; }
#endif
// SYM_ST_POLICY (build knob, measurement): the cache-policy bits of the 16-byte PCM stores -- 0: nt (the default, what
// __builtin_nontemporal_store emits), 1: sc1 nt, 2: sc0 sc1 nt, 3: sc0 sc1 (system scope, temporal), 4: sc1.
#ifndef SYM_ST_POLICY
#define SYM_ST_POLICY 0
#endif
#if defined(__HIP_DEVICE_COMPILE__) && (SYM_NT & 2)
__device__ __forceinline__ void st_stream(float4 *p, float4 v) {
#if SYM_ST_POLICY == 0
    __builtin_nontemporal_store(nt_f4{v.x, v.y, v.z, v.w}, reinterpret_cast<nt_f4 *>(p));
#else
    const nt_f4 x{v.x, v.y, v.z, v.w};
#if SYM_ST_POLICY == 1
    asm volatile("global_store_dwordx4 %0, %1, off sc1 nt" ::"v"(p), "v"(x) : "memory");
#elif SYM_ST_POLICY == 2
    asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1 nt" ::"v"(p), "v"(x) : "memory");
#elif SYM_ST_POLICY == 3
    asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" ::"v"(p), "v"(x) : "memory");
#elif SYM_ST_POLICY == 4
    asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(p), "v"(x) : "memory");
#else  // 5: nt again, through the same inline asm (the control for 1..4: the compiler's waitcnt bookkeeping does not see asm stores)
    asm volatile("global_store_dwordx4 %0, %1, off nt" ::"v"(p), "v"(x) : "memory");
#endif
#endif
}
__device__ __forceinline__ void st_stream(int4 *p, int4 v) {
    __builtin_nontemporal_store(nt_i4{v.x, v.y, v.z, v.w}, reinterpret_cast<nt_i4 *>(p));
}
__device__ __forceinline__ void st_stream(float *p, float v) { __builtin_nontemporal_store(v, p); }
#else
__device__ __forceinline__ void st_stream(float4 *p, float4 v) { *p = v; }
__device__ __forceinline__ void st_stream(int4 *p, int4 v) { *p = v; }
__device__ __forceinline__ void st_stream(float *p, float v) { *p = v; }
#endif

// Complex<f32>.  Two builds of the same arithmetic (every operation rounds exactly as num-complex's scalar code does):
//   SYM_PACKED_C32 0: scalar v_mul_f32 / v_add_f32 (rounds 1-2, and the emulation build);
//   SYM_PACKED_C32 1: a complex value is an aligned register pair and the butterflies are v_pk_mul_f32 / v_pk_add_f32 with
//     op_sel / neg modifiers: a complex product is 3 instructions instead of 6, a butterfly's sum and difference 2 instead of 4.
// Round 1 measured packed f32 at half the issue rate of scalar f32 on the SIMD's VALU port and dropped it.  Round 3's
// tools/ubench/valu_clock.hip shows the other half of the picture: a WAVEFRONT issues at most one instruction per ~4.9 cycles
// whatever it is, so at one or two wavefronts per SIMD -- where the transform kernels run, for their registers -- the instruction
// count, not the port, is what the arithmetic costs.
// Chosen per source file (a file defines SYM_PACKED_C32_DEFAULT 1 in front of its includes; the build knob SYM_PACKED_C32 overrides
// every file): measured in one call (profiles/r03s_packed_c32.txt) the packed form wins where the kernel has the registers for it
// (Vorbis 256 / 2048 +7 %, the 2048-point Fft +16 %) and loses where the aligned pairs push a kernel over its register budget
// (Vorbis 4096 / 8192-sample blocks -16 / -23 %, AAC -2 %).
#ifndef SYM_PACKED_C32_DEFAULT
#define SYM_PACKED_C32_DEFAULT 0
#endif
#ifndef SYM_PACKED_C32
#define SYM_PACKED_C32 SYM_PACKED_C32_DEFAULT
#endif
#if SYM_PACKED_C32 && !defined(SYMACCEL_EMULATED_HIP)
#define SYM_C32_IS_PACKED 1
typedef float v2f_t __attribute__((ext_vector_type(2)));
struct c32 {
    union {
        struct {
            float x, y;  // re, im
        };
        v2f_t v;
    };
};
__device__ __forceinline__ c32 c_pack(v2f_t v) {
    c32 r;
    r.v = v;
    return r;
}
// v_pk_add_f32 with source modifiers the compiler's instruction selection does not produce from generic vector code
#define SYM_PK_ADD(name, mods)                                                \
    __device__ __forceinline__ v2f_t name(v2f_t a, v2f_t b) {                 \
        v2f_t r;                                                              \
        asm("v_pk_add_f32 %0, %1, %2 " mods : "=v"(r) : "v"(a), "v"(b));      \
        return r;                                                             \
    }
SYM_PK_ADD(pk_sub_add, "neg_lo:[0,1]")                                    // (a.x - b.x, a.y + b.y)
SYM_PK_ADD(pk_add_rsub, "neg_hi:[1,0]")                                   // (a.x + b.x, b.y - a.y)
SYM_PK_ADD(pk_add_swap_sub, "op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[0,1]")  // (a.x + b.y, a.y - b.x)
SYM_PK_ADD(pk_sub_swap_add, "op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1]")  // (a.x - b.y, a.y + b.x)
SYM_PK_ADD(pk_lolo_sub_add, "op_sel:[0,1] op_sel_hi:[0,1] neg_lo:[0,1]")  // (a.x - b.y, a.x + b.y)
#undef SYM_PK_ADD

__device__ __forceinline__ c32 c_add(c32 a, c32 b) { return c_pack(a.v + b.v); }
__device__ __forceinline__ c32 c_sub(c32 a, c32 b) { return c_pack(a.v - b.v); }
// num-complex `Mul`: (a.re*b.re - a.im*b.im, a.re*b.im + a.im*b.re)
__device__ __forceinline__ c32 c_mul(c32 a, c32 b) {
    const v2f_t t1 = v2f_t{a.x, a.x} * b.v;              // (a.x b.x, a.x b.y)
    const v2f_t t2 = v2f_t{a.y, a.y} * v2f_t{b.y, b.x};  // (a.y b.y, a.y b.x)
    return c_pack(pk_sub_add(t1, t2));
}
// Imdct pre-twiddle (mdct.rs:81-88): even = spec[2i], odd = -spec[N-1-2i],
// z = (odd*w.im - even*w.re, odd*w.re + even*w.im).  `mirrored_line` = spec[N-1-2i] (not yet negated).
__device__ __forceinline__ c32 pre_twiddle(float even_line, float mirrored_line, c32 w) {
    const float odd = -mirrored_line;
    const v2f_t t1 = v2f_t{odd, odd} * v2f_t{w.y, w.x};
    const v2f_t t2 = v2f_t{even_line, even_line} * w.v;
    return c_pack(pk_sub_add(t1, t2));
}
// Imdct post-twiddle (mdct.rs:104 / 123): val = w * x.conj() = (w.x x.x - w.y (-x.y), w.x (-x.y) + w.y x.x); negating a factor
// negates the product exactly, so this is (w.x x.x + w.y x.y, w.y x.x - w.x x.y) with the same four roundings
__device__ __forceinline__ c32 post_twiddle(c32 x, c32 w) {
    const v2f_t t1 = v2f_t{w.x, w.x} * x.v;              // (w.x x.x, w.x x.y)
    const v2f_t t2 = v2f_t{w.y, w.y} * v2f_t{x.y, x.x};  // (w.y x.y, w.y x.x)
    return c_pack(pk_add_rsub(t1, t2));
}

// One radix-2 DIT butterfly: q already twiddled.  e' = e + q, o' = e - q.
__device__ __forceinline__ void bfly(c32 &e, c32 &o, c32 q) {
    const c32 p = e;
    e = c_add(p, q);
    o = c_sub(p, q);
}

#define SYM_FRAC_1_SQRT_2 0.70710678118654752440f

// Twiddles of the unrolled fft4/fft8 combine steps (no_simd.rs:405-447), compile-time forms, fused with their butterfly:
// q = -i v = (v.y, -v.x):  e' = (p.x + v.y, p.y - v.x), o' = (p.x - v.y, p.y + v.x)
__device__ __forceinline__ void bfly_minus_i(c32 &e, c32 &o) {
    const c32 p = e, v = o;
    e = c_pack(pk_add_swap_sub(p.v, v.v));
    o = c_pack(pk_sub_swap_add(p.v, v.v));
}
__device__ __forceinline__ c32 tw_minus_i(c32 v) { return c32{v.y, -v.x}; }            // k = n/4
__device__ __forceinline__ c32 tw_n8(c32 v) {                                          // k = n/8: (a + b, b - a)
    const v2f_t t = v.v * v2f_t{SYM_FRAC_1_SQRT_2, SYM_FRAC_1_SQRT_2};
    return c_pack(pk_add_swap_sub(t, t));
}
__device__ __forceinline__ c32 tw_3n8(c32 v) {                                         // k = 3n/8: (a - b, a + b)
    const v2f_t t = v.v * v2f_t{-SYM_FRAC_1_SQRT_2, -SYM_FRAC_1_SQRT_2};
    return c_pack(pk_lolo_sub_add(t, t));
}

// fft8 (no_simd.rs:405-454) on 8 values already in bit-reversed order, in registers.
__device__ __forceinline__ void fft8_regs(c32 (&x)[8]) {
    bfly(x[0], x[1], x[1]);  // fft2 x4
    bfly(x[2], x[3], x[3]);
    bfly(x[4], x[5], x[5]);
    bfly(x[6], x[7], x[7]);
    bfly(x[0], x[2], x[2]);  // fft4 x2: k=0 plain, k=1 multiply by -i
    bfly_minus_i(x[1], x[3]);
    bfly(x[4], x[6], x[6]);
    bfly_minus_i(x[5], x[7]);
    bfly(x[0], x[4], x[4]);  // fft8 combine
    bfly(x[1], x[5], tw_n8(x[5]));
    bfly_minus_i(x[2], x[6]);
    bfly(x[3], x[7], tw_3n8(x[7]));
}
#else
#define SYM_C32_IS_PACKED 0
struct c32 {
    float x, y;  // re, im
};

__device__ __forceinline__ c32 c_add(c32 a, c32 b) { return c32{a.x + b.x, a.y + b.y}; }
__device__ __forceinline__ c32 c_sub(c32 a, c32 b) { return c32{a.x - b.x, a.y - b.y}; }
// num-complex `Mul`: (a.re*b.re - a.im*b.im, a.re*b.im + a.im*b.re)
__device__ __forceinline__ c32 c_mul(c32 a, c32 b) { return c32{a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x}; }

// Imdct pre-twiddle (mdct.rs:81-88): even = spec[2i], odd = -spec[N-1-2i],
// z = (odd*w.im - even*w.re, odd*w.re + even*w.im).  `mirrored_line` = spec[N-1-2i] (not yet negated).
__device__ __forceinline__ c32 pre_twiddle(float even_line, float mirrored_line, c32 w) {
    const float odd = -mirrored_line;
    return c32{odd * w.y - even_line * w.x, odd * w.x + even_line * w.y};
}
// Imdct post-twiddle (mdct.rs:104 / 123): val = w * x.conj()
__device__ __forceinline__ c32 post_twiddle(c32 x, c32 w) { return c_mul(w, c32{x.x, -x.y}); }

// One radix-2 DIT butterfly: q already twiddled.  e' = e + q, o' = e - q.
__device__ __forceinline__ void bfly(c32 &e, c32 &o, c32 q) {
    const c32 p = e;
    e = c_add(p, q);
    o = c_sub(p, q);
}

#define SYM_FRAC_1_SQRT_2 0.70710678118654752440f

// Twiddles of the unrolled fft4/fft8 combine steps (no_simd.rs:405-447), compile-time forms.
__device__ __forceinline__ c32 tw_minus_i(c32 v) { return c32{v.y, -v.x}; }            // k = n/4
__device__ __forceinline__ c32 tw_n8(c32 v) {                                          // k = n/8
    const float a = SYM_FRAC_1_SQRT_2 * v.x, b = SYM_FRAC_1_SQRT_2 * v.y;
    return c32{a + b, b - a};
}
__device__ __forceinline__ c32 tw_3n8(c32 v) {                                         // k = 3n/8
    const float a = -SYM_FRAC_1_SQRT_2 * v.x, b = -SYM_FRAC_1_SQRT_2 * v.y;
    return c32{a - b, a + b};
}

// fft8 (no_simd.rs:405-454) on 8 values already in bit-reversed order, in registers.
__device__ __forceinline__ void fft8_regs(c32 (&x)[8]) {
    bfly(x[0], x[1], x[1]);  // fft2 x4
    bfly(x[2], x[3], x[3]);
    bfly(x[4], x[5], x[5]);
    bfly(x[6], x[7], x[7]);
    bfly(x[0], x[2], x[2]);  // fft4 x2: k=0 plain, k=1 multiply by -i
    bfly(x[1], x[3], tw_minus_i(x[3]));
    bfly(x[4], x[6], x[6]);
    bfly(x[5], x[7], tw_minus_i(x[7]));
    bfly(x[0], x[4], x[4]);  // fft8 combine
    bfly(x[1], x[5], tw_n8(x[5]));
    bfly(x[2], x[6], tw_minus_i(x[6]));
    bfly(x[3], x[7], tw_3n8(x[7]));
}
#endif

// Lane-dependent twiddle of the fft16 / fft32 combine step: `form` 0 = complex product with w
// (w holds the literal, or (+-c,-c) for the k = n/8, 3n/8 strength-reduced forms, which are the
// same roundings), 1 = k == 0 (no twiddle), 2 = multiply by -i.  Selection, not branching, so a
// wavefront with mixed k stays converged and every lane gets the reference's exact operations.
__device__ __forceinline__ c32 tw_small(c32 v, c32 w, int form) {
    const c32 g = c_mul(w, v);
    c32 q;
    q.x = form == 1 ? v.x : (form == 2 ? v.y : g.x);
    q.y = form == 1 ? v.y : (form == 2 ? -v.x : g.y);
    return q;
}

__device__ __forceinline__ unsigned rev_bits(unsigned x, int bits) { return __brev(x) >> (32 - bits); }

}  // namespace symaccel
