// ALAC integer path: the adaptive (sign-LMS) linear predictor of an element channel and the mid/side
// decorrelation -- bit-exact, wrapping i32 arithmetic as the reference's release build computes it.
//
// Reference: symphonia-codec-alac/src/lib.rs:165-264 (ElementChannel::predict), :659-661 (clip_msbs),
// :664-671 (decorrelate_mid_side).
//
// MI355X mapping: the same one-lane-per-block tiling as flac.hip (lane_tiles.h): the recurrence -- a FIR over the
// previous `order` outputs relative to out[i - order - 1], then a data-dependent coefficient update that stops as
// soon as the residual changes sign -- is serial inside an element channel and independent across them.  The
// latest outputs are a short per-lane shift register (as long as the largest order in the wavefront: 4, 8, 16 or
// 32 taps); out[i - order - 1], a per-lane distance, is read back from the lane's LDS row; the early `break` of the
// update loop is a per-lane predicate; both prediction passes of the mode-15 / order-31 case are fused into the one
// streaming pass.  Bound: integer ALU (about 13 x order operations per sample), not HBM.
#include "lane_tiles.h"

// (the attribute's argument is an expression of a template parameter: hidden from the host-only emulation build, whose
// compiler does not know the attribute and cannot parse that)
#ifndef SYM_ALAC_UPDATE
#define SYM_ALAC_UPDATE 1  // 0: the round-5 sign-LMS update in the narrow form too (A/B); 1: the residual carried as -|res| (alac_step)
#endif
#ifndef SYM_ALAC_UNROLL
#define SYM_ALAC_UNROLL 2  // groups of four samples per iteration of the hot instantiation's tile loop (measured: 1 -> 2.378 ms, 2 -> 2.33, 4 -> 2.32; 8 crashes hipcc's register allocator)
#endif
#ifndef SYM_ALAC_SMALL_WAVES
#define SYM_ALAC_SMALL_WAVES 3  // wavefronts per SIMD of the orders-<=-8 instantiations (build-time tuning knob)
#endif
#if defined(__HIPCC__)
#define SYM_ALAC_OCCUPANCY(n) __attribute__((amdgpu_waves_per_eu(n, n)))
#else
#define SYM_ALAC_OCCUPANCY(n)
#endif

namespace symaccel {

namespace {

__device__ __forceinline__ int32_t wrap_sub(int32_t a, int32_t b) { return (int32_t)((uint32_t)a - (uint32_t)b); }
__device__ __forceinline__ int32_t wrap_mul(int32_t a, int32_t b) { return (int32_t)((uint32_t)a * (uint32_t)b); }
// The two multiplies of a predictor tap.  v_mul_lo_u32 issues at a quarter of the rate of the other integer operations;
// v_mul_i32_i24 (the low 32 bits of a signed 24 x 24-bit product) at the full rate, and it IS wrap_mul whenever both
// operands are representable in 24 signed bits.  The kernel exists in both forms; each wavefront proves or fails a bound
// on its 64 blocks before the first sample (alac_predict_kernel, "narrow") and only the matching form processes it.
template <bool M24>
__device__ __forceinline__ int32_t tap_mul(int32_t a, int32_t b) {
    if constexpr (M24) return __mul24(a, b);
    else return wrap_mul(a, b);
}
// a * b + c (wrapping): v_mad_i32_i24 in the narrow form -- one instruction of the multiplies' cost class instead of a multiply
// and an add (tools/ubench/valu_int.hip: 1.85 ns per wave-instruction on a SIMD for v_mul_i32_i24, v_mad_i32_i24, v_mul_lo_u32 and
// every other three-operand / compare / select instruction, 1.05 ns for add, sub, xor, and, shifts)
template <bool M24>
__device__ __forceinline__ int32_t tap_mad(int32_t a, int32_t b, int32_t c) {
    if constexpr (M24) {
#if defined(__HIP_DEVICE_COMPILE__)
        int32_t r;
        asm("v_mad_i32_i24 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
        return r;
#else
        return (int32_t)((uint32_t)__mul24(a, b) + (uint32_t)c);
#endif
    }
    else {
        // v_mul_lo_u32 + v_add_u32, kept apart: fused, hipcc emits chains of v_mad_u64_u32, the form that ran the FLAC integer kernel
        // 2 - 4 x slower than the same sum issued from asm statements (profiles/HISTORY.md round 5)
        int32_t m = wrap_mul(a, b);
#if defined(__HIP_DEVICE_COMPILE__)
        asm("" : "+v"(m));
#endif
        return (int32_t)((uint32_t)m + (uint32_t)c);
    }
}
// signum(v) = median(v, -1, 1): ONE instruction.  Written as `v > 1 ? 1 : (v < -1 ? -1 : v)` or as max(min(v, 1), -1), hipcc
// emits two compares and two selects for it -- a sixth of the adaptive update's instructions.
__device__ __forceinline__ int32_t signum_i32(int32_t v) {
#if defined(__HIP_DEVICE_COMPILE__)
    int32_t r;
    asm("v_med3_i32 %0, %1, -1, 1" : "=v"(r) : "v"(v));
    return r;
#else
    return v > 1 ? 1 : (v < -1 ? -1 : v);
#endif
}
// |a - b| + c on unsigned operands: one instruction
__device__ __forceinline__ uint32_t sad_u32(uint32_t a, uint32_t b, uint32_t c) {
#if defined(__HIP_DEVICE_COMPILE__)
    uint32_t r;
    asm("v_sad_u32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
#else
    return (a > b ? a - b : b - a) + c;
#endif
}
// clip_msbs (lib.rs:659-661)
__device__ __forceinline__ int32_t clip_msbs(int32_t v, uint32_t num) { return (int32_t)((uint32_t)v << num) >> num; }

// NC = 32: any order; out[i - order - 1] is read back from the LDS tiles.  NC = 8: the instantiation for wavefronts whose
// orders are all <= 8 -- a ninth history register holds out[i - order - 1], there is no LDS read-back, no second tile,
// and the kernel fits four wavefronts per SIMD instead of two.
template <int NC>
struct AlacLane {
    int32_t c[NC];      // lpc_coeffs (lib.rs:79), adapted in place; entries >= order stay 0
    int32_t h[NC == 32 ? 32 : NC + 1];  // shift register of the latest outputs: h[k] = out[i - 1 - k]
    int32_t p1_prev;    // previous output of the first (order-1) pass of the double predictor (lib.rs:185-189)
    unsigned order, shift, clip;
    uint32_t ceil_mask;  // 2^shift - 1 (the sign-LMS update of the narrow form)
    uint32_t half;       // (1 << shift) >> 1: the prediction's rounding term (lib.rs:219)
    bool enabled, twice;
    bool any_twice;  // wave-uniform: some block of the wavefront runs the double predictor (lib.rs:185)
};

// clip_msbs to the lane's bit depth.  Narrow form (bps <= 23): sign-extending the low bps bits is ONE v_bfe_i32 (the field's width is a
// 5-bit operand, so a 32-bit channel cannot take this form) instead of a shift pair.
template <bool M24, int NC>
__device__ __forceinline__ int32_t clip_bits(int32_t v, const AlacLane<NC> &L) {
#if defined(__HIP_DEVICE_COMPILE__)
    if constexpr (M24) return __builtin_amdgcn_sbfe(v, 0u, 32u - L.clip);
#endif
    return clip_msbs(v, L.clip);
}

// One sample.  `past_far` = out[i - order - 1] as read back from LDS (valid for order >= 3).
// FULL: every lane's order equals TAPS (wave-uniform), so no tap needs neutralising.
// STEADY: every lane of the wavefront is enabled and past its warm-up samples (i > order): no per-sample conditions.
template <int TAPS, bool M24, bool FULL, int NC, bool STEADY = false, bool TW = true, bool QF = false>
__device__ __forceinline__ int32_t alac_step(AlacLane<NC> &L, int32_t x, unsigned i, int32_t past_far) {
    // SYM_ALAC_UPDATE == 2 (narrow form): the history registers hold the outputs with the sign bit flipped, i.e. in unsigned order, so that
    // |h[k] - past0| + rounding is ONE v_sad_u32 on them (the bias cancels in every difference)
    constexpr uint32_t HB = (M24 && SYM_ALAC_UPDATE == 2) ? 0x80000000u : 0u;
    if (STEADY || L.enabled) {
        // first pass of the double predictor: out[i] = clip(out[i] + out[i-1]) over the whole block
        if constexpr (STEADY) {
            if constexpr (TW) {  // (TW = false: no block of the wavefront runs the double predictor -- order 31 / mode 15 is rare)
                const int32_t x2 = clip_bits<M24>(wrap_add(x, L.p1_prev), L);
                x = L.twice ? x2 : x;
            }
        } else {
            if (L.twice && i >= 1) x = clip_msbs(wrap_add(x, L.p1_prev), L.clip);
        }
        L.p1_prev = x;
        if (!STEADY && i >= 1 && i <= L.order) {
            x = clip_msbs(wrap_add(x, (int32_t)((uint32_t)L.h[0] ^ HB)), L.clip);  // warm-up samples (lib.rs:196-198)
        } else if (STEADY || i > L.order) {
            int32_t res = x;
            int32_t past0;
            if constexpr (NC == 32) {
                past0 = L.order == 1 ? L.h[1] : (L.order == 2 ? L.h[2] : (int32_t)((uint32_t)past_far ^ HB));
            } else if constexpr (FULL) {
                past0 = L.h[TAPS];
            } else {
                past0 = L.h[1];
#pragma unroll
                for (int k = 2; k <= TAPS; ++k) past0 = L.order == (unsigned)k ? L.h[k] : past0;
            }
            // the taps' differences h[k] - past0: the prediction multiplies them, the update (narrow form) reads them again
            int32_t dk[TAPS];
#pragma unroll
            for (int k = 0; k < TAPS; ++k) dk[k] = wrap_sub(L.h[k], past0);
            int32_t sum = (int32_t)L.half;  // (the rounding term leads the sum)
            if constexpr (M24 && TAPS <= 8) {
                // one chain, from the oldest sample to the newest (wrapping sums re-associate): the join of two chains is an instruction, and the newest
                // difference -- the one that waits for the previous sample -- comes last
#pragma unroll
                for (int k = TAPS - 1; k >= 0; --k) sum = tap_mad<M24>(L.c[k], dk[k], sum);
            } else {
                int32_t sum1 = 0;  // (two chains of multiply-adds)
#pragma unroll
                for (int k = 0; k < TAPS; k += 2) {
                    sum = tap_mad<M24>(L.c[k], dk[k], sum);
                    if (k + 1 < TAPS) sum1 = tap_mad<M24>(L.c[k + 1], dk[k + 1], sum1);
                }
                sum = wrap_add(sum, sum1);
            }
            const int32_t val = sum >> L.shift;
            x = clip_bits<M24>(wrap_add(wrap_add(x, (int32_t)((uint32_t)past0 ^ HB)), val), L);
            // sign-LMS update (lib.rs:224-260): from the oldest sample (coefficient order-1) to the newest, until the
            // residual reaches or crosses zero (the reference's `break`, here a per-lane predicate).
            // Shape of the code (it decides the kernel's speed; one lane per block means every predicate is per lane):
            //  * predicates are 0 / -1 words in VGPRs and applied with AND / XOR / BFI, never `?:` on a bool that lives
            //    across taps -- hipcc keeps such bools as SGPR-pair masks, ran out of SGPRs at 32 taps, spilled them
            //    into VGPR lanes and branched around every coefficient update;
            //  * tap k subtracts t_k = (1 + j) * step_k from the residual, and neither t_k nor the direction the
            //    coefficient moves in depends on the residual: only the residual itself and "still active" are serial (the
            //    residual keeps being updated after the reference's `break`, where nothing reads it any more).  Round 2 first
            //    carried a separate running sum of the t_k to shorten that chain; one instruction per tap more and, with three
            //    or four wavefronts per SIMD, nothing gained (profiles/r02zg_alac_med3_ab.txt);
            //  * a tap beyond the lane's own order (TAPS is the wavefront's maximum) is neutralised by zeroing its
            //    difference v: step, t_k and the coefficient move are then 0, and since those taps come first, while the
            //    running sum is still 0, the activity test sees the untouched residual and changes nothing.
            //  * the instruction mix (round 5, tools/ubench/valu_int.hip): a multiply, a multiply-add, a compare or a select costs 1.75 x
            //    an add / xor / shift on the SIMD, so +-|v| >> shift is what the reference writes -- (sign * val) >> shift, ONE multiply by
            //    the signed direction, which the coefficient needs anyway -- instead of the five cheap operations of a branch-free abs,
            //    and the residual takes its tap as a multiply-add.  In the narrow form val = past0 - sample is -dk[k] with no wrap (the
            //    differences fit 24 bits), so the update works on dk[k] with the signs folded; the full-width form keeps val as the
            //    reference computes it (past0 - sample wraps to the same sign as sample - past0 at -2^31).
            //    (Taking the sign AFTER a conditional negate -- xor, sub, median instead of median, negate, select -- measured equal: 3.00 against 2.98 ms.)
            if constexpr ((M24 && SYM_ALAC_UPDATE != 0) || QF) {
                // Round 6, narrow form (and the "mid" wavefronts of the wide form: QF, below).  The serial part of a tap was xor + compare + select ("still on the residual's side") and the
                // direction cost xor + sub per tap; both disappear when the residual is carried as Q = -|res|:
                //  * res > 0: every tap subtracts (1 + j) * (|val| >> shift) >= 0 and the lane stays active while res > 0;
                //    res < 0: every tap subtracts (1 + j) * ((-|val|) >> shift) <= 0, active while res < 0.  With
                //    a_k = |val| >> shift resp. -((-|val|) >> shift) = (|val| + 2^shift - 1) >> shift (an arithmetic shift floors, so the
                //    negative case is a ceiling) both are Q += (1 + j) * a_k, active while Q < 0: the test is the sign bit (ashr + and).
                //    No wrap: |val| < 2^23 (narrow) and 1 + j <= 32, so a still-active Q (< 0) moves by less than 2^28 per tap -- and
                //    once inactive the mask is 0 for good, whatever Q does afterwards.
                //  * the coefficient moves by signum(-val) * (res > 0 ? 1 : -1) while active: ONE multiply-add with the mask that already
                //    carries the direction (+-1, or 0 once inactive).
                // |val| is signum(v) * v (one multiply of the 1.8 ns class; the signum is needed for the coefficient anyway).
                // QF (wide form, orders <= 8): the same carried residual for wavefronts whose blocks are "mid" (alac_narrow_kernel: outputs of at
                // most 26 bits, so |val| < 2^27 and (1 + j) * a_k < 2^30 -- an active Q cannot wrap); |val| is max(v, -v) there (v does not fit a
                // 24-bit multiply), the prediction keeps its full 32-bit multiplies.  Both forms compute the reference's update bit for bit, so
                // a block may change between them from tile to tile (only full steady tiles take this one).
                const int32_t s = res >> 31;                                               // -1: negative residual
                int32_t Q = wrap_sub(s, res ^ s);                                          // -|res|
                int32_t actdir = signum_i32(res);                                          // +1 / -1 while active, 0 for res == 0 (one v_med3_i32)
                const uint32_t rnd = (uint32_t)s & L.ceil_mask;                            // 2^shift - 1 for a negative residual: the ceiling
#pragma unroll
                for (int k = TAPS - 1; k >= 0; --k) {
                    const int32_t nk = FULL ? TAPS - k : (int32_t)L.order - k;             // (1 + j); <= 0 beyond the order
                    int32_t v = dk[k];                                                     // -val
                    if constexpr (!FULL) v &= wrap_sub(0, nk) >> 31;                       // 0 beyond the lane's order: a_k = 0, no move
                    const int32_t sg = signum_i32(v);
                    uint32_t a;
                    if constexpr (HB != 0 && FULL) a = sad_u32((uint32_t)L.h[k], (uint32_t)past0, rnd) >> L.shift;
                    else if constexpr (!M24) a = ((uint32_t)max(v, wrap_sub(0, v)) + rnd) >> L.shift;
                    else a = ((uint32_t)__mul24(sg, v) + rnd) >> L.shift;                  // |val| >> shift, resp. its ceiling
                    L.c[k] = tap_mad<true>(sg, actdir, L.c[k]);
                    if constexpr (!M24) Q = wrap_add(Q, wrap_mul(nk, (int32_t)a));         // (FULL: 1 + j is a literal -- shifts and adds)
                    else Q = tap_mad<true>(nk, (int32_t)a, Q);
                    actdir &= Q >> 31;
                }
            } else {
            const int32_t pm = res > 0 ? 0 : -1;      // 0: the residual is positive, -1: negative (zero: never active)
            int32_t act = res != 0 ? -1 : 0;
#pragma unroll
            for (int k = TAPS - 1; k >= 0; --k) {
                const int32_t nk = (int32_t)L.order - k;                                   // (1 + j); <= 0 beyond the order
                int32_t v = M24 ? dk[k] : wrap_sub(past0, L.h[k]);                         // narrow: -val; full width: val
                if constexpr (!FULL) v &= wrap_sub(0, nk) >> 31;                           // 0 beyond the lane's order
                // narrow: pm ? -sign(-val) : sign(-val) = minus the direction below; full width: pm ? -sign(val) : sign(val)
                const int32_t dir = wrap_sub(signum_i32(v) ^ pm, pm);
                const int32_t step = tap_mul<M24>(dir, v) >> L.shift;                      // (+-sign * val) >> shift = +-|val| >> shift (the two minus signs of the narrow form cancel)
                if constexpr (M24) L.c[k] = wrap_add(L.c[k], dir & act);                   // c -= +-sign, while still active
                else L.c[k] = wrap_sub(L.c[k], dir & act);
                res = tap_mad<M24>(wrap_sub(0, nk), step, res);                            // the residual after this tap
                act = (res ^ pm) > pm ? act : 0;                                           // > 0 resp. < 0: still on the residual's side
            }
            }
        }
    }
#pragma unroll
    for (int k = (NC == 32 ? TAPS - 1 : TAPS); k >= 1; --k) L.h[k] = L.h[k - 1];
    L.h[0] = (int32_t)((uint32_t)x ^ HB);
    return x;
}

// 32 samples of one tile.  `row` = this tile's LDS row of the lane, `prev_row` = the previous tile's (still intact in
// the other LDS buffer); t0 = absolute index of column 0.  Rows are written back four samples at a time, so anything
// older than three samples can be read back from LDS: that is where out[i - order - 1] comes from for order >= 3.
template <int TAPS, bool M24, bool FULL, int NC, bool STEADY = false, bool TW = true, bool QF = false>
__device__ __forceinline__ void alac_steps32(AlacLane<NC> &L, int32_t *row, const int32_t *prev_row, unsigned t0, int n_valid) {
    // (the hot instantiation -- steady, every lane at the wavefront's order, no double predictor -- unrolled SYM_ALAC_UNROLL groups deep: the nine history
    // registers rotate by four per group, which hipcc resolves with register copies at the loop's back edge; the other instantiations stay rolled for the
    // instruction cache)
    constexpr int kUnroll = (STEADY && FULL && !TW && NC == 8) ? SYM_ALAC_UNROLL : 1;
#pragma unroll kUnroll
    for (int u0 = 0; u0 < 32; u0 += 4) {
        const int4 v = *reinterpret_cast<const int4 *>(row + u0);
        int32_t xs[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int u = u0 + q;
            if (STEADY || u < n_valid) {
                int32_t far = 0;
                if constexpr (NC == 32) {
                    const int idx = u - (int)L.order - 1;
                    far = idx >= 0 ? row[idx >= 0 ? idx : 0] : prev_row[32 + (idx < -32 ? -32 : idx)];
                }
                xs[q] = alac_step<TAPS, M24, FULL, NC, STEADY, TW, QF>(L, xs[q], t0 + (unsigned)u, far);
            }
        }
        *reinterpret_cast<int4 *>(row + u0) = make_int4(xs[0], xs[1], xs[2], xs[3]);
    }
}

// decorrelate_mid_side (lib.rs:664-671) for one sample of a pair: out0 = s0 + s1 - ((s1 * weight) >> shift),
// out1 = out0 - s1; weight == 0 leaves the pair alone (lib.rs:552).  `own` / `other`: this row's and the pair's other
// row's predicted sample; rows 2p / 2p+1 are channels 0 / 1 of pair p.
__device__ __forceinline__ int32_t alac_mixed(int32_t weight, uint32_t shift, bool is_ch1, int32_t own, int32_t other) {
    if (weight == 0) return own;
    const int32_t s0 = is_ch1 ? other : own, s1 = is_ch1 ? own : other;
    const int32_t o0 = wrap_sub(wrap_add(s0, s1), wrap_mul(s1, weight) >> shift);
    return is_ch1 ? wrap_sub(o0, s1) : o0;
}
__device__ __forceinline__ void alac_store_mixed(int32_t *__restrict__ buf, const int32_t *tile, const int32_t *row_weight,
                                                 const uint8_t *row_shift, size_t blk0, size_t n_blocks, unsigned stride,
                                                 unsigned t0, unsigned cols, int lane, bool fast) {
    if (fast) {
        // a lane takes four columns of BOTH rows of a pair (32 pairs per tile: four rounds of eight pairs x eight column groups): every
        // predicted sample is read from LDS once, the product is shared, and "weight == 0 leaves the pair alone" is a mask, not the
        // per-sample divergent branch `alac_mixed` compiles to
        const int q = lane & 7, psub = lane >> 3;
#pragma unroll 1
        for (int k = 0; k < 4; ++k) {
            const int r0 = 2 * (8 * k + psub);
            const int4 a = *reinterpret_cast<const int4 *>(tile + r0 * kStride + 4 * q);
            const int4 b = *reinterpret_cast<const int4 *>(tile + (r0 + 1) * kStride + 4 * q);
            const int32_t w = row_weight[r0];
            const uint32_t sh = row_shift[r0];
            const uint32_t wz = w == 0 ? 0xffffffffu : 0u;
            auto mix = [&](int32_t s0, int32_t s1, int32_t &o0, int32_t &o1) {
                const uint32_t t = (uint32_t)(wrap_mul(s1, w) >> sh);          // lib.rs:664-671
                o0 = (int32_t)((uint32_t)s0 + (((uint32_t)s1 - t) & ~wz));      // s0 + s1 - t, or s0
                const uint32_t x = (uint32_t)s0 - t;                            // (s0 + s1 - t) - s1
                o1 = (int32_t)(x ^ ((x ^ (uint32_t)s1) & wz));                 // ... or s1
            };
            int4 o0, o1;
            mix(a.x, b.x, o0.x, o1.x);
            mix(a.y, b.y, o0.y, o1.y);
            mix(a.z, b.z, o0.z, o1.z);
            mix(a.w, b.w, o0.w, o1.w);
            int32_t *dst = buf + (blk0 + (size_t)r0) * stride + t0 + 4u * (unsigned)q;
            *reinterpret_cast<int4 *>(dst) = o0;
            *reinterpret_cast<int4 *>(dst + stride) = o1;
        }
    } else {
        const int c = lane & 31, rsub = lane >> 5;
#pragma unroll 4
        for (int r = rsub; r < kRows; r += 2) {
            if (blk0 + (size_t)r < n_blocks && (unsigned)c < cols)
                buf[(blk0 + (size_t)r) * stride + t0 + (unsigned)c] =
                    alac_mixed(row_weight[r], row_shift[r], (r & 1) != 0, tile[r * kStride + c], tile[(r ^ 1) * kStride + c]);
        }
    }
}

// MIX: blocks 2p / 2p+1 are the two channels of element pair p; decorrelate_mid_side runs as the predicted tile is
// written back (decode_element, lib.rs:541-560, in one pass).
// "narrow": every multiply of a block has operands of at most 24 signed bits, so that v_mul_i32_i24 computes what the
// reference's wrapping i32 multiply computes.  The operands are (coefficient, difference of two earlier outputs) and
// (1..32, a shifted-down such difference):
//  * every output but a block's first is clip_msbs'd to the channel's bit depth (lib.rs:196-198, 221); with a depth of
//    at most 23 bits and a first sample inside [-2^22, 2^22) any difference of two outputs is inside +-(2^23 - 1);
//  * a coefficient moves by at most one per sample (lib.rs:236-257): starting inside +-2^22, a block of at most 2^21
//    samples keeps it inside +-2^23.
// Anything else (24-bit channels -- 25 bits on a side channel --, a caller's arbitrary i32 data) takes the full 32-bit
// multiply.  One flag per wavefront of 64 blocks, computed BEFORE the in-place prediction touches the buffer.
__global__ __launch_bounds__(64) void alac_narrow_kernel(const int32_t *__restrict__ buf, const symaccel_alac_desc *__restrict__ desc,
                                                         const int32_t *__restrict__ coeffs, size_t n_blocks, unsigned blocksize, unsigned stride,
                                                         uint8_t *__restrict__ narrow_flag) {
    const size_t my = (size_t)blockIdx.x * kRows + threadIdx.x;
    bool narrow = true, small = true, mid = true;  // small: order <= 8 (the register-only instantiation, AlacLane<8>); mid: see alac_step, QF
    if (my < n_blocks) {
        const symaccel_alac_desc d = desc[my];
        const unsigned order = d.lpc_order > 31u ? 31u : d.lpc_order;
        const bool enabled = (d.mode == 0 || d.mode >= 15) && order != 0;
        if (enabled) {
            small = order <= 8u;
            const unsigned bps = d.bps < 1u ? 1u : (d.bps > 32u ? 32u : d.bps);
            const int32_t first = buf[my * (size_t)stride];
            narrow = bps <= 23u && blocksize <= (1u << 21) && first >= -(1 << 22) && first < (1 << 22);
            mid = bps <= 26u && first >= -(1 << 25) && first < (1 << 25);
            const int4 *cp = reinterpret_cast<const int4 *>(coeffs + my * 32);
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int4 v = cp[j];
                const int32_t c[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    if ((unsigned)(4 * j + q) < order) narrow = narrow && c[q] > -(1 << 22) && c[q] < (1 << 22);
            }
        }
    }
    const bool all = __all(narrow) != 0, all_small = __all(small) != 0, all_mid = __all(mid) != 0;
    if (threadIdx.x == 0) narrow_flag[blockIdx.x] = (uint8_t)((all ? 1 : 0) | (all_small ? 2 : 0) | (all_mid ? 4 : 0));
}

// M24: the instantiation for wavefronts whose blocks are all "narrow" (see there); SMALL: for wavefronts whose orders are
// all <= 8 (AlacLane<8>: half the LDS, under 128 VGPRs -- four wavefronts per SIMD instead of two).  All four
// instantiations are launched over the whole grid and a wavefront returns at once from those that are not its own (the
// loops in one kernel cost the register allocator its occupancy; and the choice cannot be re-derived by a later launch,
// the buffer being predicted in place by an earlier one).
template <bool MIX, bool M24, bool SMALL>
__global__ __launch_bounds__(64) SYM_ALAC_OCCUPANCY(SMALL ? SYM_ALAC_SMALL_WAVES : 2) void alac_predict_kernel(
    int32_t *__restrict__ buf, const symaccel_alac_desc *__restrict__ desc, const int32_t *__restrict__ coeffs,
    size_t n_blocks, unsigned blocksize, unsigned stride, const int32_t *__restrict__ pair_weight,
    const uint8_t *__restrict__ pair_shift, const uint8_t *__restrict__ narrow_flag) {
    const unsigned wave_class = narrow_flag[blockIdx.x];  // alac_narrow_kernel
    if (((wave_class & 1u) != 0) != M24 || ((wave_class & 2u) != 0) != SMALL) return;  // another instantiation's wavefront
    const bool mid = (wave_class & 4u) != 0;
    constexpr int NC = SMALL ? 8 : 32;
    __shared__ __attribute__((aligned(16))) int32_t tiles[(SMALL ? 1 : 2) * kTileWords];
    __shared__ int32_t row_weight[kRows];
    __shared__ uint8_t row_shift[kRows];
    const int lane = (int)threadIdx.x;
    const size_t blk0 = (size_t)blockIdx.x * kRows;
    const size_t my = blk0 + (size_t)lane;
    const bool have = my < n_blocks;
    if constexpr (MIX) {  // ordered by the first tile's wave_sync
        row_weight[lane] = have ? pair_weight[my >> 1] : 0;
        row_shift[lane] = have ? (uint8_t)(pair_shift[my >> 1] & 31u) : 0;
    }

    AlacLane<NC> L;
#pragma unroll
    for (int j = 0; j < NC; ++j) L.c[j] = 0;
#pragma unroll
    for (int j = 0; j < (NC == 32 ? 32 : NC + 1); ++j) L.h[j] = (M24 && SYM_ALAC_UPDATE == 2) ? (int32_t)0x80000000u : 0;  // (zero, in the history's representation)
    L.p1_prev = 0;
    L.order = L.shift = L.clip = 0;
    L.ceil_mask = 0;
    L.half = 0;
    L.enabled = L.twice = false;
    if (have) {
        const symaccel_alac_desc d = desc[my];
        const bool valid_mode = d.mode == 0 || d.mode >= 15;  // lib.rs:167-169 (mode is a 4-bit field)
        L.order = d.lpc_order > 31u ? 31u : d.lpc_order;
        L.shift = d.shift & 31u;
        L.ceil_mask = (1u << L.shift) - 1u;
        L.half = (1u << L.shift) >> 1;
        L.clip = 32u - (d.bps < 1u ? 1u : (d.bps > 32u ? 32u : d.bps));
        L.enabled = valid_mode && L.order != 0;                // lib.rs:173-175
        L.twice = L.order == 31 || d.mode == 15;               // lib.rs:185
        const int4 *cp = reinterpret_cast<const int4 *>(coeffs + my * 32);
#pragma unroll
        for (int j = 0; j < NC / 4; ++j) {
            const int4 v = cp[j];
            L.c[4 * j] = (unsigned)(4 * j) < L.order ? v.x : 0;
            L.c[4 * j + 1] = (unsigned)(4 * j + 1) < L.order ? v.y : 0;
            L.c[4 * j + 2] = (unsigned)(4 * j + 2) < L.order ? v.z : 0;
            L.c[4 * j + 3] = (unsigned)(4 * j + 3) < L.order ? v.w : 0;
        }
    }
    L.any_twice = __any(L.enabled && L.twice) != 0;
    const unsigned max_order = wave_max(L.enabled ? L.order : 0u);
    const bool full8 = __all(!have || !L.enabled || L.order == 8u) != 0 && max_order == 8u;
    const bool steady_ok = __all(!have || L.enabled) != 0;  // (with full8: past sample 8 no lane has a per-sample condition left)

    const bool aligned = (stride & 3u) == 0 && blk0 + kRows <= n_blocks;  // rows `stride` words apart (>= blocksize)
    const unsigned n_tiles = (blocksize + kCols - 1) / kCols;
    TilePrefetch pre;
#pragma unroll
    for (int k = 0; k < 8; ++k) pre.v[k] = make_int4(0, 0, 0, 0);
    if (aligned && blocksize >= (unsigned)kCols) tile_issue_loads(buf, pre, blk0, stride, 0, lane);
    for (unsigned t = 0; t < n_tiles; ++t) {
        const unsigned t0 = t * kCols;
        const unsigned cols = min((unsigned)kCols, blocksize - t0);
        const bool fast = aligned && cols == (unsigned)kCols;
        int32_t *tile = tiles + (SMALL ? 0u : (t & 1u) * kTileWords);
        const int32_t *prev_tile = tiles + (SMALL ? 0u : ((t & 1u) ^ 1u) * kTileWords);
        if constexpr (SMALL) wave_sync();  // one tile: the previous round's write-back has read it
        if (fast)
            tile_commit(pre, tile, lane);
        else
            tile_fetch_slow(buf, tile, blk0, n_blocks, stride, t0, cols, lane);
        wave_sync();
        if (aligned && t0 + 2u * kCols <= blocksize) tile_issue_loads(buf, pre, blk0, stride, t0 + kCols, lane);
        if (have) {
            int32_t *row = tile + lane * kStride;
            const int32_t *prow = prev_tile + lane * kStride;
            // steady: a full tile past every lane's warm-up samples, every block of the wavefront enabled
            const bool steady = steady_ok && cols == (unsigned)kCols && t0 > max_order;
            if constexpr (SMALL) {
                if (full8) {  // the common stream: every block of the wavefront has order 8
                    // (wide form: the instantiation without the double-predictor pass is the "mid" one, QF; other wavefronts take the general steady form)
                    if (steady && !L.any_twice && (M24 || mid)) alac_steps32<8, M24, true, 8, true, false, !M24>(L, row, prow, t0, (int)cols);
                    else if (steady) alac_steps32<8, M24, true, 8, true>(L, row, prow, t0, (int)cols);
                    else alac_steps32<8, M24, true>(L, row, prow, t0, (int)cols);
                } else if (max_order <= 4) {
                    if (steady) alac_steps32<4, M24, false, 8, true>(L, row, prow, t0, (int)cols);
                    else alac_steps32<4, M24, false>(L, row, prow, t0, (int)cols);
                } else if (max_order <= 6 && steady) {
                    // orders up to 6 (what ffmpeg's encoder writes: 4 .. 6 per frame, so a wavefront of blocks from many streams is mixed): six taps in the steady
                    // tiles; the first and the ragged last tile take the eight-tap form below (taps beyond a lane's order are neutral there: same results)
                    if (!L.any_twice) alac_steps32<6, M24, false, 8, true, false>(L, row, prow, t0, (int)cols);
                    else alac_steps32<6, M24, false, 8, true>(L, row, prow, t0, (int)cols);
                } else {
                    if (steady) alac_steps32<8, M24, false, 8, true>(L, row, prow, t0, (int)cols);
                    else alac_steps32<8, M24, false>(L, row, prow, t0, (int)cols);
                }
            } else {
                // (no steady variant here: a second copy of the 16- / 32-tap step beside the first does not fit the register
                // file -- 284 to 336 bytes of scratch)
                if (max_order <= 16)
                    alac_steps32<16, M24, false>(L, row, prow, t0, (int)cols);
                else
                    alac_steps32<32, M24, false>(L, row, prow, t0, (int)cols);
            }
        }
        wave_sync();
        if constexpr (MIX) {
            alac_store_mixed(buf, tile, row_weight, row_shift, blk0, n_blocks, stride, t0, cols, lane, fast);
        } else {
            if (fast)
                tile_store_fast(buf, tile, blk0, stride, t0, lane);
            else
                tile_store_slow(buf, tile, blk0, n_blocks, stride, t0, cols, lane);
        }
    }
}

// lib.rs:664-671, pair p: out0 = ch0[p][..], out1 = ch1[p][..]; weight == 0 leaves the pair alone (lib.rs:552)
__global__ void alac_mid_side_kernel(const int32_t *__restrict__ weight, const uint8_t *__restrict__ shift,
                                     int32_t *__restrict__ ch0, int32_t *__restrict__ ch1, size_t n_pairs, size_t blocksize) {
    const size_t total = n_pairs * blocksize;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const size_t p = i / blocksize;
        const int32_t w = weight[p];
        if (w == 0) continue;
        const int32_t a = ch0[i], b = ch1[i];
        const int32_t s0 = wrap_sub(wrap_add(a, b), wrap_mul(b, w) >> (shift[p] & 31u));
        ch0[i] = s0;
        ch1[i] = wrap_sub(s0, b);
    }
}

}  // namespace

int launch_alac_predict(symaccel_ctx *ctx, int32_t *d_buf, const symaccel_alac_desc *d_desc, const int32_t *d_coeffs,
                        size_t n_blocks, size_t blocksize, const int32_t *d_pair_weight, const uint8_t *d_pair_shift, size_t stride) {
    if (stride == 0) stride = blocksize;  // rows back to back
    if (stride < blocksize || stride > 0xffffffffu) return SYMACCEL_ERR_INVALID_ARG;
    const size_t grid = (n_blocks + kRows - 1) / kRows;
    if (grid > 0x7fffffffu) return SYMACCEL_ERR_INVALID_ARG;
    // one byte per wavefront: 24-bit multiplies are exact for its 64 blocks.  A buffer of its own: the context's shared scratch
    // synchronises the stream when it grows, and another entry point interleaved on the same context may hold it.
    if (ctx->alac_flags_bytes < grid) {
        const size_t want = grid < 4096 ? 4096 : grid + grid / 2;
        void *nf = nullptr;
        SYM_TRY(ctx_alloc(ctx, &nf, want, false));
        if (ctx->alac_flags) {
            SYM_GPU(ctx, hipStreamSynchronize(ctx->stream));  // (launches queued earlier still read the old buffer)
            SYM_GPU(ctx, hipFree(ctx->alac_flags));
        }
        ctx->alac_flags = nf;
        ctx->alac_flags_bytes = want;
    }
    void *flags = ctx->alac_flags;
    hipLaunchKernelGGL(alac_narrow_kernel, dim3((unsigned)grid), dim3(64), 0, ctx->stream, d_buf, d_desc, d_coeffs, n_blocks,
                       (unsigned)blocksize, (unsigned)stride, (uint8_t *)flags);
    // four launches over the same grid, one per (24-bit multiplies, orders <= 8) class: each wavefront runs in exactly one
#define SYM_ALAC_LAUNCH(MIX, M24, SMALL)                                                                                     \
    hipLaunchKernelGGL((alac_predict_kernel<MIX, M24, SMALL>), dim3((unsigned)grid), dim3(64), 0, ctx->stream, d_buf, d_desc, \
                       d_coeffs, n_blocks, (unsigned)blocksize, (unsigned)stride, d_pair_weight, d_pair_shift, (const uint8_t *)flags)
    if (d_pair_weight) {
        SYM_ALAC_LAUNCH(true, true, true);
        SYM_ALAC_LAUNCH(true, true, false);
        SYM_ALAC_LAUNCH(true, false, true);
        SYM_ALAC_LAUNCH(true, false, false);
    } else {
        SYM_ALAC_LAUNCH(false, true, true);
        SYM_ALAC_LAUNCH(false, true, false);
        SYM_ALAC_LAUNCH(false, false, true);
        SYM_ALAC_LAUNCH(false, false, false);
    }
#undef SYM_ALAC_LAUNCH
    SYM_GPU(ctx, hipGetLastError());
    return SYMACCEL_OK;
}

int launch_alac_mid_side(symaccel_ctx *ctx, const int32_t *d_weight, const uint8_t *d_shift, int32_t *d_ch0, int32_t *d_ch1,
                         size_t n_pairs, size_t blocksize) {
    const size_t total = n_pairs * blocksize;
    const unsigned grid = (unsigned)((total + 255) / 256 < 16384 ? (total + 255) / 256 : 16384);
    hipLaunchKernelGGL(alac_mid_side_kernel, dim3(grid), dim3(256), 0, ctx->stream, d_weight, d_shift, d_ch0, d_ch1, n_pairs,
                       blocksize);
    SYM_GPU(ctx, hipGetLastError());
    return SYMACCEL_OK;
}

}  // namespace symaccel
