// Device-side arithmetic primitives shared by the kernels.
//
// ARITHMETIC CONTRACT: the reference never fuses a multiply with an add, so every
// a*b +/- c*d here must compile to two rounded multiplies and one rounded add/sub.  The build
// passes -ffp-contract=off (build.py) and keeps f32 denormals on (no
// -fgpu-flush-denormals-to-zero); tests/test_build.py greps the ISA for v_fma/v_mac/v_mad.
#pragma once

#include <hip/hip_runtime.h>

#include <pk_f32.h>

#include "symaccel_internal.h"

namespace symaccel {

// Pointer to a host-generated constant table, in the CONSTANT address space: the tables are written
// once before any launch, so uniform-index reads may use scalar loads (s_load_dword through the
// scalar cache into SGPRs) instead of per-lane vector loads that each occupy a VGPR.
using cf32p = const __attribute__((address_space(4))) float *;
__device__ __forceinline__ cf32p as_const(const float *p) {
    return (cf32p)p;  // deliberate address-space cast (global -> constant), same 64-bit representation
}

// Complex<f32> as a packed pair (x = re, y = im): one v_pk_* instruction per complex add, three per
// complex multiply (pk_f32.h).
using c32 = v2f;

__device__ __forceinline__ c32 c_add(c32 a, c32 b) { return a + b; }
__device__ __forceinline__ c32 c_sub(c32 a, c32 b) { return a - b; }
// num-complex `Mul`: (a.re*b.re - a.im*b.im, a.re*b.im + a.im*b.re)
__device__ __forceinline__ c32 c_mul(c32 a, c32 b) {
    return pk_add_neg_lo(pk_mul_xx(a, b), pk_mul_yy_swap(a, b));
}
// w * x.conj()  (mdct.rs:104 / 123)
__device__ __forceinline__ c32 c_mul_conj(c32 w, c32 x) {
    return pk_add_neg_lo(pk_mul_xx_conj(w, x), pk_mul_yy_swap_conj(w, x));
}

// Imdct pre-twiddle (mdct.rs:81-88): even = spec[2i], odd = -spec[N-1-2i],
// z = (odd*w.im - even*w.re, odd*w.re + even*w.im).  `mirrored_line` = spec[N-1-2i] (not yet negated).
__device__ __forceinline__ c32 pre_twiddle(float even_line, float mirrored_line, c32 w) {
    const v2f em{even_line, mirrored_line};
    return pk_add_neg_lo(pk_mul_nyy_swap(em, w), pk_mul_xx(em, w));
}
// Imdct post-twiddle (mdct.rs:104 / 123): val = w * x.conj()
__device__ __forceinline__ c32 post_twiddle(c32 x, c32 w) { return c_mul_conj(w, x); }

// One radix-2 DIT butterfly: q already twiddled.  e' = e + q, o' = e - q.
__device__ __forceinline__ void bfly(c32 &e, c32 &o, c32 q) {
    const c32 p = e;
    e = p + q;
    o = p - q;
}
// Butterfly whose twiddle is -i (k = n/4): q = (o.im, -o.re), folded into the adds.
__device__ __forceinline__ void bfly_minus_i(c32 &e, c32 &o) {
    const c32 p = e, v = o;
    e = pk_add_mi(p, v);
    o = pk_sub_mi(p, v);
}

#define SYM_FRAC_1_SQRT_2 0.70710678118654752440f

// Twiddles of the unrolled fft4/fft8 combine steps (no_simd.rs:405-447), compile-time forms.
__device__ __forceinline__ c32 tw_minus_i(c32 v) { return c32{v.y, -v.x}; }            // k = n/4
__device__ __forceinline__ c32 tw_n8(c32 v) {                                          // k = n/8: (a + b, b - a)
    return pk_sum_diff(v * c32{SYM_FRAC_1_SQRT_2, SYM_FRAC_1_SQRT_2});
}
__device__ __forceinline__ c32 tw_3n8(c32 v) {                                         // k = 3n/8: (a - b, a + b)
    return pk_diff_sum(v * c32{-SYM_FRAC_1_SQRT_2, -SYM_FRAC_1_SQRT_2});
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
