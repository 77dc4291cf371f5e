// Packed-f32 arithmetic for gfx950 (v_pk_mul_f32 / v_pk_add_f32: two IEEE f32 operations per lane per
// instruction, full rate -- the only way to reach the FP32 vector peak on CDNA without FMA, and FMA is
// ruled out by the arithmetic contract).  Each packed half rounds exactly like the
// scalar instruction; op_sel / neg_lo / neg_hi only route and negate inputs, which is exact.
//
// The compiler selects packed instructions for plain element-wise v2f expressions by itself; the
// helpers below pin the forms it does not find on its own (a negation on one half only, swizzled adds).
// Included as <pk_f32.h>: the test-only CPU emulation build shadows this file with plain C++
// (tests/emu/include/pk_f32.h), because inline gfx950 assembly cannot run there.
#pragma once

#include <hip/hip_runtime.h>

namespace symaccel {

typedef float v2f __attribute__((ext_vector_type(2)));

#define SYM_PK_OP2(name, text)                                   \
    __device__ __forceinline__ v2f name(v2f a, v2f b) {          \
        v2f r;                                                   \
        asm(text : "=v"(r) : "v"(a), "v"(b));                    \
        return r;                                                \
    }
#define SYM_PK_OP1(name, text)                                   \
    __device__ __forceinline__ v2f name(v2f a) {                 \
        v2f r;                                                   \
        asm(text : "=v"(r) : "v"(a));                            \
        return r;                                                \
    }

// (a.x * b.x, a.x * b.y)
SYM_PK_OP2(pk_mul_xx, "v_pk_mul_f32 %0, %1, %2 op_sel_hi:[0,1]")
// (a.y * b.y, a.y * b.x)
SYM_PK_OP2(pk_mul_yy_swap, "v_pk_mul_f32 %0, %1, %2 op_sel:[1,1] op_sel_hi:[1,0]")
// (a.x * b.x, a.x * -b.y)
SYM_PK_OP2(pk_mul_xx_conj, "v_pk_mul_f32 %0, %1, %2 op_sel_hi:[0,1] neg_hi:[0,1]")
// (a.y * -b.y, a.y * b.x)
SYM_PK_OP2(pk_mul_yy_swap_conj, "v_pk_mul_f32 %0, %1, %2 op_sel:[1,1] op_sel_hi:[1,0] neg_lo:[0,1]")
// (-a.y * b.y, -a.y * b.x)
SYM_PK_OP2(pk_mul_nyy_swap, "v_pk_mul_f32 %0, %1, %2 op_sel:[1,1] op_sel_hi:[1,0] neg_lo:[1,0] neg_hi:[1,0]")
// (a.x - b.x, a.y + b.y)
SYM_PK_OP2(pk_add_neg_lo, "v_pk_add_f32 %0, %1, %2 neg_lo:[0,1]")
// (a.x + b.x, a.y - b.y)
SYM_PK_OP2(pk_add_neg_hi, "v_pk_add_f32 %0, %1, %2 neg_hi:[0,1]")
// a + (b.y, -b.x) = (a.x + b.y, a.y - b.x)
SYM_PK_OP2(pk_add_mi, "v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[0,1]")
// a - (b.y, -b.x) = (a.x - b.y, a.y + b.x)
SYM_PK_OP2(pk_sub_mi, "v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1]")
// (a.x + a.y, a.y - a.x)
SYM_PK_OP1(pk_sum_diff, "v_pk_add_f32 %0, %1, %1 op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[0,1]")
// (a.x - a.y, a.x + a.y)
SYM_PK_OP1(pk_diff_sum, "v_pk_add_f32 %0, %1, %1 op_sel:[0,1] op_sel_hi:[0,1] neg_lo:[0,1]")

#undef SYM_PK_OP1
#undef SYM_PK_OP2

}  // namespace symaccel
