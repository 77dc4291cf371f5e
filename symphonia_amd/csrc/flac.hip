// FLAC integer path: fixed / LPC predictor restore, wasted-bits shift, stereo decorrelation and
// the final left-justification shift -- bit-exact (i64 accumulation, wrapping i32 adds).
//
// Reference: symphonia-bundle-flac/src/decoder.rs:663-710 (fixed_predict), :716-752 (lpc_predict,
// dispatch :487-504), :403-409 (samples_shl), :32-82 (decorrelate_*), :239-242 (<< (32 - bps)).
//
// MI355X mapping (DESIGN.md "flac_restore"): the recurrence is serial inside a subframe and
// independent across subframes, so one LANE owns one subframe.  A wavefront walks its 64 subframes
// in tiles of 32 samples staged through LDS (16-byte global accesses, eight 128-byte row segments per
// instruction; rows of 36 words so every lane reads its own row with conflict-free b128); the next tile's
// loads are in flight while the recurrence runs over the current one; the last 32 samples stay in registers
// (circular, statically indexed by unrolling 32 steps).  All orders use one formula
//   pred_i = sum_{j < order} c_j * s[i-1-j]   (exact 64-bit),   s[i] += (i32)(pred_i >> shift)
// which equals the reference's prefill + main loop (its padded taps are zero) and, with the
// binomial coefficients and shift 0, its fixed predictors.  The sum runs on the FP64 FMA pipe when every
// coefficient of the wavefront is below 2^16 in magnitude (exact, see lpc_steps32_f64), else on v_mad_i64_i32.
// Bound: FP64 FMA issue (32 FMAs per 8 B), NOT HBM.
#include <type_traits>

#include "lane_tiles.h"

namespace symaccel {

namespace {

// 32 recurrence steps with statically indexed circular history: before step u the most recent
// sample sits in h[(u + 31) & 31].  Integer path: any coefficient magnitude.
// ALL: as in lpc_steps32_f64 -- every sample of the tile is predicted in every lane, no per-step predicate.
// The ORDER of the taps in the chain (round 6 experiment; the sum is exact whatever the association).  Newest sample first -- the order the sum is written in --
// makes the first FMA of sample u + 1 wait for the last instruction of sample u; oldest first, only the LAST FMA of a sample needs its predecessor.  If the
// kernel were bound by the latency of the dependent chain this (and SYM_FLAC_GROUP below) would show; it does not: 32 v_fmac_f64 cost a SIMD ~3.6 ns each.
#ifndef SYM_FLAC_OLDEST_FIRST
#define SYM_FLAC_OLDEST_FIRST 0  // (measured equal, 7.67-7.71 ms both ways: hipcc keeps a sample's chain together in either order and two wavefronts per SIMD cover it -- profiles/r06zz12_flac_order_ab.txt)
#endif
template <int TAPS, bool ALL = false>
__device__ __forceinline__ void lpc_steps32(int32_t (&h)[32], const int32_t (&c)[32], int32_t *row, int col0,
                                            int first_pred, int n_valid, uint32_t shift, uint32_t wasted) {
    int32_t xs[4];
#pragma unroll
    for (int u = 0; u < 32; ++u) {
        if ((u & 3) == 0) {  // four samples per LDS access (columns past n_valid are padding inside the row)
            const int4 v = *reinterpret_cast<const int4 *>(row + col0 + u);
            xs[0] = v.x; xs[1] = v.y; xs[2] = v.z; xs[3] = v.w;
        }
        if (u < n_valid) {
            int32_t x = xs[u & 3];
            if (ALL || col0 + u >= first_pred) {
                int64_t acc = 0;
#if defined(__HIP_DEVICE_COMPILE__)
                // v_mad_i64_i32 issued from asm statements, four taps each: the sequence hipcc emits for the plain C++ sum -- the same 32
                // dependent instructions back to back, carry-out in an SGPR pair -- ran config 5 through this kernel in 19.8 ms, this form
                // in 9.4 (with vcc or an SGPR pair as carry-out alike).  Not explained: in isolation (tools/ubench/valu_int.hip) every
                // variant of the instruction issues in 1.85 ns (profiles/HISTORY.md round 5).
                static_assert(TAPS % 4 == 0, "groups of four taps");
#pragma unroll
                for (int jj = 0; jj < TAPS; jj += 4) {
                    // (SYM_FLAC_OLDEST_FIRST, see there: the groups from the oldest samples to the newest, and inside a group too)
                    const int j = SYM_FLAC_OLDEST_FIRST ? TAPS - 4 - jj : jj;
                    constexpr int R = SYM_FLAC_OLDEST_FIRST ? 3 : 0;  // tap j + (R ^ k) is the k-th of the group
                    asm("v_mad_i64_i32 %0, vcc, %1, %5, %0\n\tv_mad_i64_i32 %0, vcc, %2, %6, %0\n\tv_mad_i64_i32 %0, vcc, %3, %7, %0\n\tv_mad_i64_i32 %0, vcc, %4, %8, %0"
                        : "+v"(acc)
                        : "v"(c[j + (R ^ 0)]), "v"(c[j + (R ^ 1)]), "v"(c[j + (R ^ 2)]), "v"(c[j + (R ^ 3)]), "v"(h[(u + 31 - j - (R ^ 0)) & 31]),
                          "v"(h[(u + 31 - j - (R ^ 1)) & 31]), "v"(h[(u + 31 - j - (R ^ 2)) & 31]), "v"(h[(u + 31 - j - (R ^ 3)) & 31])
                        : "vcc");
                }
#else
#pragma unroll
                for (int j = 0; j < TAPS; ++j) acc += (int64_t)c[j] * (int64_t)h[(u + 31 - j) & 31];
#endif
                x = wrap_add(x, (int32_t)(acc >> shift));
            }
            h[u & 31] = x;
            xs[u & 3] = (int32_t)((uint32_t)x << wasted);  // samples_shl (decoder.rs:403-409)
        }
        if ((u & 3) == 3) *reinterpret_cast<int4 *>(row + col0 + u - 3) = make_int4(xs[0], xs[1], xs[2], xs[3]);
    }
}

// The same recurrence with the dot product carried by the FP64 FMA pipe, which is exact here:
// sum |c_j| < 2^20 (checked per wavefront; a valid stream has <= 32 coefficients of qlp precision <= 15 bits,
// decoder.rs:467-471) and |s| <= 2^31, so every partial sum is below 2^51 in magnitude (2^53 with the offset below):
// no FMA ever rounds and the sum is the exact integer whatever the association.  (v_mad_i64_i32 issues at the
// same rate -- tools/ubench/valu_int.hip -- but the kernel built on it is not faster: 8.7 - 9.4 ms against 8.3 for config 5.)
// (acc >> shift) as i32 without leaving the integer domain: the sum is started from 2^52 + 2^51 instead of 0, so the f64
// that comes out is 2^52 + (2^51 + acc), whose 52 mantissa bits ARE the integer 2^51 + acc (|acc| < 2^51 because a lane only
// takes this path when the magnitudes of its coefficients sum to less than 2^20, load_params).  Its low word is acc mod 2^32,
// its high word minus 0x43380000 (exponent field and the 2^51) is floor(acc / 2^32), and the wanted 32 bits of the arithmetic
// shift are one v_alignbit_b32 of the two.  (Round 1 took floor(ldexp(acc, -shift)) mod 2^32 in f64: six more FP64-rate
// instructions per sample, a tenth of the loop.)
// Partial accumulators of the 32-tap sum (build-time knob).  Four, to break the dependent FMA chain, was round 1's guess;
// measured in round 2 (profiles/r02zh_flac_ab.txt): 4 -> 9.85 ms, 2 -> 9.79 ms, 1 -> 8.60 ms for config 5 -- with two wavefronts
// per SIMD the chain's latency is covered anyway, and the extra accumulators cost FP64-rate adds and registers.
#ifndef SYM_FLAC_PARTS
#define SYM_FLAC_PARTS 1
#endif
constexpr double kFlacMagic = 6755399441055744.0;  // 2^52 + 2^51
__device__ __forceinline__ int32_t flac_shifted(double acc_plus_magic, int shift) {
    const uint64_t bits = __builtin_bit_cast(uint64_t, acc_plus_magic);
    const uint32_t lo = (uint32_t)bits, hi = (uint32_t)(bits >> 32) - 0x43380000u;
#if defined(__HIP_DEVICE_COMPILE__)
    return (int32_t)__builtin_amdgcn_alignbit(hi, lo, (uint32_t)shift);
#else
    return (int32_t)(uint32_t)((((uint64_t)hi << 32) | lo) >> (shift & 31));
#endif
}
// ALL: every sample of the tile is predicted in every lane (the tile lies behind the warm-up samples of the wavefront's highest
// order): no per-lane predicate and no divergent branch per step.  The one or two tiles in front of that take the general form.
// SYM_FLAC_GROUP (round 6): G consecutive samples' sums in flight at once.  Sample n + q needs the outputs n + q - 1 - j for tap j: every tap j >= q reads samples
// from before the group, so the G chains run their taps TAPS - 1 .. q interleaved (G independent accumulators: a dependent v_fmac_f64 issues only every ~17 cycles,
// two wavefronts per SIMD cover two of those, hipcc's schedulers keep a sample's chain together whatever the tap order), then the group finishes in order: sample n,
// its output into the q taps the later chains still miss, sample n + 1, ...  The sums are exact whatever their association (see above): same bits.
#ifndef SYM_FLAC_GROUP
#define SYM_FLAC_GROUP 1  // (measured: 4 -> 7.92 ms, 2 -> 8.0, 1 -> 7.70 for config 5: the kernel is bound by the FP64 pipe itself, not by the chain's latency -- profiles/r06zz13_flac_group_ab.txt)
#endif
template <int TAPS>
__device__ __forceinline__ void lpc_steps32_f64_grouped(double (&h)[32], const double (&c)[32], int32_t *row, int col0, int shift, uint32_t wasted) {
    constexpr int G = SYM_FLAC_GROUP;
    static_assert(TAPS >= G && 4 % G == 0, "group of 1, 2 or 4 samples");
#pragma unroll
    for (int u0 = 0; u0 < 32; u0 += 4) {
        const int4 v = *reinterpret_cast<const int4 *>(row + col0 + u0);
        int32_t xs[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int g0 = 0; g0 < 4; g0 += G) {
            const int ub = u0 + g0;  // first sample of the group
            double acc[G];
#pragma unroll
            for (int q = 0; q < G; ++q) acc[q] = kFlacMagic;
#pragma unroll
            for (int j = TAPS - 1; j >= 0; --j) {
#pragma unroll
                for (int q = 0; q < G; ++q)
                    if (j >= q) acc[q] = __builtin_fma(c[j], h[(ub + q + 31 - j) & 31], acc[q]);
            }
#pragma unroll
            for (int q = 0; q < G; ++q) {
                const int32_t x = wrap_add(xs[g0 + q], flac_shifted(acc[q], shift));
                const double xd = (double)x;
                h[(ub + q) & 31] = xd;
                xs[g0 + q] = (int32_t)((uint32_t)x << wasted);  // samples_shl (decoder.rs:403-409)
#pragma unroll
                for (int r = q + 1; r < G; ++r) acc[r] = __builtin_fma(c[r - q - 1], xd, acc[r]);  // tap r - q - 1 of sample ub + r reads sample ub + q
            }
        }
        *reinterpret_cast<int4 *>(row + col0 + u0) = make_int4(xs[0], xs[1], xs[2], xs[3]);
    }
}

template <int TAPS, bool ALL = false>
__device__ __forceinline__ void lpc_steps32_f64(double (&h)[32], const double (&c)[32], int32_t *row, int col0,
                                                int first_pred, int n_valid, int shift, uint32_t wasted) {
    if constexpr (ALL && SYM_FLAC_GROUP > 1 && TAPS >= SYM_FLAC_GROUP) {
        lpc_steps32_f64_grouped<TAPS>(h, c, row, col0, shift, wasted);
        return;
    }
    int32_t xs[4];
#pragma unroll
    for (int u = 0; u < 32; ++u) {
        if ((u & 3) == 0) {
            const int4 v = *reinterpret_cast<const int4 *>(row + col0 + u);
            xs[0] = v.x; xs[1] = v.y; xs[2] = v.z; xs[3] = v.w;
        }
        if (u < n_valid) {
            int32_t x = xs[u & 3];
            if (ALL || col0 + u >= first_pred) {
                constexpr int P = TAPS >= 8 ? SYM_FLAC_PARTS : 1;
                double part[P];
#pragma unroll
                for (int q = 0; q < P; ++q) part[q] = q == 0 ? kFlacMagic : 0.0;
#pragma unroll
                for (int jj = 0; jj < TAPS; ++jj) {
                    const int j = SYM_FLAC_OLDEST_FIRST ? TAPS - 1 - jj : jj;
                    part[j % P] = __builtin_fma(c[j], h[(u + 31 - j) & 31], part[j % P]);
                }
                double acc = part[0];
                if constexpr (P == 4) acc = (part[0] + part[1]) + (part[2] + part[3]);
                if constexpr (P == 2) acc = part[0] + part[1];
                x = wrap_add(x, flac_shifted(acc, shift));
            }
            h[u & 31] = (double)x;
            xs[u & 3] = (int32_t)((uint32_t)x << wasted);  // samples_shl (decoder.rs:403-409)
        }
        if ((u & 3) == 3) *reinterpret_cast<int4 *>(row + col0 + u - 3) = make_int4(xs[0], xs[1], xs[2], xs[3]);
    }
}

__device__ __forceinline__ int iabs_sat(int32_t v) { return v < 0 ? (v == INT32_MIN ? INT32_MAX : -v) : v; }

// Per-lane subframe parameters, shared by both kernels.
struct LaneParams {
    int32_t c[32];
    unsigned order, shift, wasted, max_order;
    bool use_f64;
};

__device__ __forceinline__ void load_params(LaneParams &p, const symaccel_flac_desc *__restrict__ desc,
                                            const int32_t *__restrict__ coeffs, size_t my, bool have,
                                            unsigned blocksize) {
#pragma unroll
    for (int j = 0; j < 32; ++j) p.c[j] = 0;
    p.order = p.shift = p.wasted = 0;
    if (have) {
        const symaccel_flac_desc d = desc[my];
        p.wasted = d.wasted_bits & 31u;
        if (d.kind == SYMACCEL_FLAC_LPC) {
            p.order = d.order > 32u ? 32u : d.order;
            // a shift above 31 encodes a negative qlp shift, which the reference refuses (decoder.rs:506-508; the status kernel
            // flags the block): clamped, so that such a block's (meaningless) output does not depend on whether its wavefront
            // took the f64 path (v_alignbit uses shift & 31) or the i64 path (a full 64-bit shift)
            p.shift = d.shift > 31u ? 31u : d.shift;
            const int4 *cp = reinterpret_cast<const int4 *>(coeffs + my * 32);
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int4 v = cp[j];
                p.c[4 * j] = (unsigned)(4 * j) < p.order ? v.x : 0;
                p.c[4 * j + 1] = (unsigned)(4 * j + 1) < p.order ? v.y : 0;
                p.c[4 * j + 2] = (unsigned)(4 * j + 2) < p.order ? v.z : 0;
                p.c[4 * j + 3] = (unsigned)(4 * j + 3) < p.order ? v.w : 0;
            }
        } else if (d.kind == SYMACCEL_FLAC_FIXED) {
            p.order = d.order > 4u ? 4u : d.order;
            // decoder.rs:679-707: s(i) = sum binom * s(i-k)
            p.c[0] = p.order == 1 ? 1 : p.order == 2 ? 2 : p.order == 3 ? 3 : p.order == 4 ? 4 : 0;
            p.c[1] = p.order == 2 ? -1 : p.order == 3 ? -3 : p.order == 4 ? -6 : 0;
            p.c[2] = p.order == 3 ? 1 : p.order == 4 ? 4 : 0;
            p.c[3] = p.order == 4 ? -1 : 0;
        }
        if (p.order > blocksize) p.order = blocksize;  // decoder.rs:456-458 would have rejected the frame
    }
    // Does any lane of this wavefront need more than 4 / 12 taps?  (wave-uniform specialisation)
    p.max_order = wave_max(p.order);
    // FP64 path iff in every lane of the wavefront the coefficient magnitudes sum to less than 2^20 (always, for valid
    // streams: <= 32 coefficients of <= 15 bits): with |sample| <= 2^31 every partial sum then stays below 2^51 in
    // magnitude -- exact in f64 whatever the association, and inside what flac_shifted can take apart
    unsigned long long csum = 0;
#pragma unroll
    for (int j = 0; j < 32; ++j) csum += (unsigned)iabs_sat(p.c[j]);
    p.use_f64 = wave_max(csum < (1ull << 20) ? 0u : 1u) == 0u;
}

// One kernel per arithmetic path, so each is register-allocated for what it keeps live (the FP64
// path: 32 coefficients + 32 history samples as doubles = 128 VGPRs).  Both are launched over the same
// grid; a wavefront whose subframes belong to the other path exits after reading its parameters.
// DECOR: blocks 2p / 2p+1 are the two channels of pair p; the stereo decorrelation (decoder.rs:32-82) and the final
// `<< (32 - bps)` (decoder.rs:239-242) are applied as the restored tile is written back -- no separate pass over HBM.
template <bool F64, bool DECOR>
__device__ __forceinline__ void flac_restore_body(int32_t *__restrict__ buf, const symaccel_flac_desc *__restrict__ desc,
                                                  const int32_t *__restrict__ coeffs, size_t n_blocks,
                                                  unsigned blocksize, unsigned stride, int32_t *tiles,
                                                  const uint8_t *__restrict__ pair_mode, uint32_t out_shift, uint8_t *row_mode) {
    const int lane = (int)threadIdx.x;
    const size_t blk0 = (size_t)blockIdx.x * kRows;
    const size_t my = blk0 + (size_t)lane;
    const bool have = my < n_blocks;
    LaneParams p;
    load_params(p, desc, coeffs, my, have, blocksize);
    if (p.use_f64 != F64) return;
    if constexpr (DECOR) row_mode[lane] = have ? pair_mode[my >> 1] : 0;  // ordered by the first tile's wave_sync

    using T = typename std::conditional<F64, double, int32_t>::type;
    T c[32], h[32];
#pragma unroll
    for (int j = 0; j < 32; ++j) {
        c[j] = (T)p.c[j];
        h[j] = (T)0;
    }
    const bool aligned = (stride & 3u) == 0 && blk0 + kRows <= n_blocks;  // rows `stride` words apart (>= blocksize): 16-byte aligned rows
    const unsigned n_tiles = (blocksize + kCols - 1) / kCols;
    TilePrefetch pre;
#pragma unroll
    for (int k = 0; k < 8; ++k) pre.v[k] = make_int4(0, 0, 0, 0);
    // Order of a round (gfx950 counts vector loads and stores with ONE in-order counter): recurrence over tile t -- tile t + 1 from its
    // prefetch registers into the other LDS tile -- loads of tile t + 2 issued -- tile t written back.  The wait for a tile's loads then
    // sits where nothing younger than those loads is in flight: they and the previous tile's stores were issued a whole recurrence
    // earlier.  (Until round 5 the write-back came first, and the wait in front of the next tile's LDS commit -- vmcnt(7) .. vmcnt(0)
    // at the loop's back edge -- also waited for the stores just issued: the write latency was exposed once per tile.)
    auto fetch_tile = [&](unsigned t) {  // tile t: prefetch registers (or, ragged, global memory) -> LDS; then the loads of tile t + 1
        const unsigned t0 = t * kCols;
        const unsigned cols = min((unsigned)kCols, blocksize - t0);
        int32_t *tile = tiles + (t & 1u) * kTileWords;
        if (aligned && cols == (unsigned)kCols)
            tile_commit(pre, tile, lane);
        else
            tile_fetch_slow(buf, tile, blk0, n_blocks, stride, t0, cols, lane);
        if (aligned && t0 + 2u * kCols <= blocksize)  // the tile after it is a full one
            tile_issue_loads(buf, pre, blk0, stride, t0 + kCols, lane);
    };
    if (aligned && blocksize >= (unsigned)kCols) tile_issue_loads(buf, pre, blk0, stride, 0, lane);
    if (n_tiles > 0) fetch_tile(0);
    wave_sync();
    for (unsigned t = 0; t < n_tiles; ++t) {
        const unsigned t0 = t * kCols;
        const unsigned cols = min((unsigned)kCols, blocksize - t0);
        const bool fast = aligned && cols == (unsigned)kCols;
        int32_t *tile = tiles + (t & 1u) * kTileWords;
        if (have) {
            int32_t *row = tile + lane * kStride;
            const int first_pred = (int)p.order - (int)t0;  // column index of the first predicted sample
            if constexpr (F64) {
                if (t0 < p.max_order)  // a tile that still holds warm-up samples of some lane: all 32 taps (zero beyond a lane's order), predicated
                    lpc_steps32_f64<32, false>(h, c, row, 0, first_pred, (int)cols, (int)p.shift, p.wasted);
                else if (p.max_order <= 4)
                    lpc_steps32_f64<4, true>(h, c, row, 0, first_pred, (int)cols, (int)p.shift, p.wasted);
                else if (p.max_order <= 12)
                    lpc_steps32_f64<12, true>(h, c, row, 0, first_pred, (int)cols, (int)p.shift, p.wasted);
                else
                    lpc_steps32_f64<32, true>(h, c, row, 0, first_pred, (int)cols, (int)p.shift, p.wasted);
            } else {
                if (t0 >= p.max_order && p.max_order > 12)  // behind every lane's warm-up samples (config 5 through this kernel: 8.27 -> 7.98 ms)
                    lpc_steps32<32, true>(h, c, row, 0, first_pred, (int)cols, p.shift, p.wasted);
                else if (p.max_order <= 4)
                    lpc_steps32<4>(h, c, row, 0, first_pred, (int)cols, p.shift, p.wasted);
                else if (p.max_order <= 12)
                    lpc_steps32<12>(h, c, row, 0, first_pred, (int)cols, p.shift, p.wasted);
                else
                    lpc_steps32<32>(h, c, row, 0, first_pred, (int)cols, p.shift, p.wasted);
            }
        }
        wave_sync();  // tile t is complete
        if (t + 1 < n_tiles) fetch_tile(t + 1);
        if constexpr (DECOR) {
            if (fast)
                tile_store_decorrelate_fast(buf, tile, row_mode, out_shift, blk0, stride, t0, lane);
            else
                tile_store_decorrelate_slow(buf, tile, row_mode, out_shift, blk0, n_blocks, stride, t0, cols, lane);
        } else {
            if (fast)
                tile_store_fast(buf, tile, blk0, stride, t0, lane);
            else
                tile_store_slow(buf, tile, blk0, n_blocks, stride, t0, cols, lane);
        }
        wave_sync();  // tile t + 1 is in place for every lane; tile t has been read: the round after next may overwrite it
    }
}

// (build knob for the A/B on padded rows -- with rows 16 KiB apart three wavefronts per SIMD never moved the time, the row pitch was the bound)
#ifndef SYM_FLAC_WAVES
#define SYM_FLAC_WAVES 2
#endif
template <bool DECOR>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(SYM_FLAC_WAVES, SYM_FLAC_WAVES))) void flac_restore_f64_kernel(
    int32_t *__restrict__ buf, const symaccel_flac_desc *__restrict__ desc, const int32_t *__restrict__ coeffs,
    size_t n_blocks, unsigned blocksize, unsigned stride, const uint8_t *__restrict__ pair_mode, uint32_t out_shift) {
    __shared__ __attribute__((aligned(16))) int32_t tiles[2 * kTileWords];
    __shared__ uint8_t row_mode[kRows];
    flac_restore_body<true, DECOR>(buf, desc, coeffs, n_blocks, blocksize, stride, tiles, pair_mode, out_shift, row_mode);
}

template <bool DECOR>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(3, 3))) void flac_restore_i64_kernel(
    int32_t *__restrict__ buf, const symaccel_flac_desc *__restrict__ desc, const int32_t *__restrict__ coeffs,
    size_t n_blocks, unsigned blocksize, unsigned stride, const uint8_t *__restrict__ pair_mode, uint32_t out_shift) {
    __shared__ __attribute__((aligned(16))) int32_t tiles[2 * kTileWords];
    __shared__ uint8_t row_mode[kRows];
    flac_restore_body<false, DECOR>(buf, desc, coeffs, n_blocks, blocksize, stride, tiles, pair_mode, out_shift, row_mode);
}

// decoder.rs:32-82 + :239-242
__global__ void flac_decorrelate_kernel(const uint8_t *__restrict__ mode, int32_t *__restrict__ ch0,
                                        int32_t *__restrict__ ch1, size_t n_pairs, size_t blocksize,
                                        uint32_t out_shift) {
    const size_t total = n_pairs * blocksize;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const unsigned m = mode[i / blocksize];
        int32_t a = ch0[i], b = ch1[i];
        if (m == 1) {  // left/side: side = left - side
            b = (int32_t)((uint32_t)a - (uint32_t)b);
        } else if (m == 2) {  // mid/side
            const int32_t mid = (int32_t)(((uint32_t)a << 1) | ((uint32_t)b & 1u));
            const int32_t sd = b;
            a = (int32_t)((uint32_t)mid + (uint32_t)sd) >> 1;
            b = (int32_t)((uint32_t)mid - (uint32_t)sd) >> 1;
        } else if (m == 3) {  // right/side: ch0 = side, ch1 = right: side += right
            a = wrap_add(a, b);
        }
        ch0[i] = (int32_t)((uint32_t)a << out_shift);
        ch1[i] = (int32_t)((uint32_t)b << out_shift);
    }
}

}  // namespace

int launch_flac_restore(symaccel_ctx *ctx, int32_t *d_buf, const symaccel_flac_desc *d_desc, const int32_t *d_coeffs,
                        size_t n_blocks, size_t blocksize, const uint8_t *d_pair_mode, uint32_t out_shift, size_t stride) {
    if (stride == 0) stride = blocksize;  // rows back to back
    if (stride < blocksize || stride > 0xffffffffu) return SYMACCEL_ERR_INVALID_ARG;
    const size_t grid = (n_blocks + kRows - 1) / kRows;
    if (grid > 0x7fffffffu) return SYMACCEL_ERR_INVALID_ARG;
    const dim3 g((unsigned)grid), b(64);
    if (d_pair_mode) {
        hipLaunchKernelGGL(flac_restore_f64_kernel<true>, g, b, 0, ctx->stream, d_buf, d_desc, d_coeffs, n_blocks,
                           (unsigned)blocksize, (unsigned)stride, d_pair_mode, out_shift);
        hipLaunchKernelGGL(flac_restore_i64_kernel<true>, g, b, 0, ctx->stream, d_buf, d_desc, d_coeffs, n_blocks,
                           (unsigned)blocksize, (unsigned)stride, d_pair_mode, out_shift);
    } else {
        hipLaunchKernelGGL(flac_restore_f64_kernel<false>, g, b, 0, ctx->stream, d_buf, d_desc, d_coeffs, n_blocks,
                           (unsigned)blocksize, (unsigned)stride, d_pair_mode, out_shift);
        hipLaunchKernelGGL(flac_restore_i64_kernel<false>, g, b, 0, ctx->stream, d_buf, d_desc, d_coeffs, n_blocks,
                           (unsigned)blocksize, (unsigned)stride, d_pair_mode, out_shift);
    }
    SYM_GPU(ctx, hipGetLastError());
    return SYMACCEL_OK;
}

int launch_flac_decorrelate(symaccel_ctx *ctx, const uint8_t *d_mode, int32_t *d_ch0, int32_t *d_ch1, size_t n_pairs,
                            size_t blocksize, uint32_t out_shift) {
    const size_t total = n_pairs * blocksize;
    const unsigned grid = (unsigned)((total + 255) / 256 < 16384 ? (total + 255) / 256 : 16384);
    hipLaunchKernelGGL(flac_decorrelate_kernel, dim3(grid), dim3(256), 0, ctx->stream, d_mode, d_ch0, d_ch1, n_pairs,
                       blocksize, out_shift);
    SYM_GPU(ctx, hipGetLastError());
    return SYMACCEL_OK;
}

}  // namespace symaccel
