// TEST-ONLY stand-in for symphonia_amd/csrc/pk_f32.h (see there): the same helpers in plain C++ so the
// CPU emulation build can run the kernels' logic.  Each half is one rounded f32 operation, as on the GPU.
#pragma once

#include <hip/hip_runtime.h>

namespace symaccel {

struct alignas(8) v2f {
    float x, y;
};
static inline v2f operator+(v2f a, v2f b) { return v2f{a.x + b.x, a.y + b.y}; }
static inline v2f operator-(v2f a, v2f b) { return v2f{a.x - b.x, a.y - b.y}; }
static inline v2f operator*(v2f a, v2f b) { return v2f{a.x * b.x, a.y * b.y}; }
static inline v2f operator-(v2f a) { return v2f{-a.x, -a.y}; }
static inline v2f &operator+=(v2f &a, v2f b) { a = a + b; return a; }

static inline v2f pk_mul_xx(v2f a, v2f b) { return v2f{a.x * b.x, a.x * b.y}; }
static inline v2f pk_mul_yy_swap(v2f a, v2f b) { return v2f{a.y * b.y, a.y * b.x}; }
static inline v2f pk_mul_xx_conj(v2f a, v2f b) { return v2f{a.x * b.x, a.x * -b.y}; }
static inline v2f pk_mul_yy_swap_conj(v2f a, v2f b) { return v2f{a.y * -b.y, a.y * b.x}; }
static inline v2f pk_mul_nyy_swap(v2f a, v2f b) { return v2f{-a.y * b.y, -a.y * b.x}; }
static inline v2f pk_add_neg_lo(v2f a, v2f b) { return v2f{a.x - b.x, a.y + b.y}; }
static inline v2f pk_add_neg_hi(v2f a, v2f b) { return v2f{a.x + b.x, a.y - b.y}; }
static inline v2f pk_add_mi(v2f a, v2f b) { return v2f{a.x + b.y, a.y - b.x}; }
static inline v2f pk_sub_mi(v2f a, v2f b) { return v2f{a.x - b.y, a.y + b.x}; }
static inline v2f pk_sum_diff(v2f a) { return v2f{a.x + a.y, a.y - a.x}; }
static inline v2f pk_diff_sum(v2f a) { return v2f{a.x - a.y, a.x + a.y}; }

}  // namespace symaccel
