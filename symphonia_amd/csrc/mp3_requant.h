// Device helpers shared by mp3_requant.hip and mp3_stereo.hip: the per-slot scale of requantize
// (symphonia-bundle-mp3/src/layer3/requantize.rs:239-353) and the sample mapping of read_huffman_samples (:117-147).
#pragma once

#include <hip/hip_runtime.h>

#include "symaccel_internal.h"

namespace symaccel {

constexpr int kMp3Slots = 40;     // 39 scale-factor slots + the "no band" slot (scale 1.0: x * 1.0f == x)
constexpr int kMp3PowLds = 1024;  // POW43 entries kept in LDS (4 KiB): almost every magnitude of a real spectrum is small

// `2^(0.25 (A - B)) as f32` of scale slot `slot` of one granule-channel; slot kMp3Unscaled (lines no band covers): 1.0
__device__ __forceinline__ float mp3_slot_scale(const DevTables &tb, const symaccel_mp3_requant &d, int slot, int switch_point) {
    // pre-emphasis, ISO/IEC 11172-3 Table B.6 (requantize.rs:256-257)
    constexpr unsigned char kPre[22] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 3, 3, 3, 2, 0};
    if (slot >= kMp3Unscaled) return 1.0f;
    const bool is_short = d.block_type == SYMACCEL_MP3_SHORT;
    const int sw = is_short ? (d.is_mixed ? switch_point : 0) : 64;  // slots below sw are long bands
    const int shift = (d.flags & SYMACCEL_MP3_RQ_SCALEFAC_SCALE) ? 2 : 1;
    const int gain = (int)d.global_gain - 210;
    int e;
    if (slot < sw) {  // requantize_long (requantize.rs:260-291)
        const int pre = ((d.flags & SYMACCEL_MP3_RQ_PREFLAG) && slot < 22) ? kPre[slot] : 0;
        e = gain - (((int)d.scalefacs[slot] + pre) << shift);
    } else {          // requantize_short (requantize.rs:315-352): window = slot index mod 3 within the short part
        const int win = (slot - sw) % 3;
        e = gain - 8 * (int)d.subblock_gain[win] - ((int)d.scalefacs[slot] << shift);
    }
    int idx = e - kMp3Pow2abMinE;
    idx = idx < 0 ? 0 : (idx >= kMp3Pow2abLen ? kMp3Pow2abLen - 1 : idx);
    return tb.mp3_pow2ab[idx];
}

// index of the line -> slot map requantize uses for this granule-channel (DevTables::mp3_band_map row)
__device__ __forceinline__ int mp3_requant_kind(const symaccel_mp3_requant &d) {
    return d.block_type == SYMACCEL_MP3_SHORT ? (d.is_mixed ? 2 : 1) : 0;
}

// (1.0 - 2.0 * sign_bit) * POW43[|s|] is +-POW43[|s|] exactly; zeros and the rzero partition are +0.0.
// pow43_lo: the first `lds_n` table entries in LDS.
__device__ __forceinline__ float mp3_sample_value(const DevTables &tb, const float *pow43_lo, int s, bool in_rzero,
                                                  int lds_n = kMp3PowLds) {
    int mag = s < 0 ? -s : s;
    mag = mag > 8206 ? 8206 : mag;
    const float p = mag < lds_n ? pow43_lo[mag] : tb.mp3_pow43[mag];
    return (in_rzero || s == 0) ? 0.0f : (s < 0 ? -p : p);
}

// ---- joint stereo (layer3/stereo.rs): shared by mp3_stereo.hip and the fused front of mp3.hip -------------------

struct SfbEdges {  // band edge tables of one sample rate (layer3/common.rs:9-172), passed to kernels by value
    int16_t longb[23], shortb[40], mixed[40];
    int16_t mixed_len, mixed_switch;
};
inline SfbEdges make_sfb_edges(const HostTables &t, int sr) {
    SfbEdges e;
    for (int i = 0; i < 23; ++i) e.longb[i] = (int16_t)t.mp3_sfb_long[sr][i];
    for (int i = 0; i < 40; ++i) {
        e.shortb[i] = (int16_t)t.mp3_sfb_short[sr][i];
        e.mixed[i] = (int16_t)t.mp3_sfb_mixed[sr][i];
    }
    e.mixed_len = (int16_t)t.mp3_sfb_mixed_len[sr];
    e.mixed_switch = (int16_t)t.mp3_sfb_switch[sr];
    return e;
}

constexpr float kMp3Frac1Sqrt2 = 0.70710678118654752440f;  // f32::consts::FRAC_1_SQRT_2 (stereo.rs:143-144)

// What stereo() does to a granule, decided per scale-factor band: bands in is_mask are intensity coded, bands in
// ms_mask (and every line below `bound`, when mid/side is on) are mid/side coded, the rest is left alone.
struct Mp3StereoPlan {
    int bound;
    unsigned long long ms_mask, is_mask;
};

// is_pos of band k (stereo.rs:226-228, 369-371): long blocks copy band 20 into band 21, short / mixed blocks copy
// scalefacs[33..36] into positions 36..39
__device__ __forceinline__ int mp3_is_pos(const symaccel_mp3_stereo &d, bool is_short, int k) {
    return d.scalefacs1[is_short ? (k < 36 ? k : k - 3) : (k < 21 ? k : 20)];
}

// The intensity-stereo band walk of process_intensity_long_block (stereo.rs:196-260) and
// process_intensity_short_block (:264-483) on a bit mask of the bands in which channel 1 is non-zero (bit k = band k
// of the block's edge table).  Pure scalar work on wave-uniform values: every lane computes the same plan.
__device__ __forceinline__ Mp3StereoPlan mp3_stereo_walk(const symaccel_mp3_stereo &d, const SfbEdges &e, unsigned long long nzmask,
                                                         int end, int rzero1) {
    const bool mid_side = d.flags & SYMACCEL_MP3_ST_MID_SIDE;
    const bool is_short = d.block_type == SYMACCEL_MP3_SHORT, is_mixed = is_short && d.is_mixed;
    const int inv_pos = (d.flags & SYMACCEL_MP3_ST_MPEG1) ? 7 : 31;  // INTENSITY_INV_POS_* (stereo.rs:19-29)
    Mp3StereoPlan plan{end, 0ull, 0ull};
    auto zero_band = [&](int k) {  // process_intensity (stereo.rs:165-186) as an action for band k
        if (mp3_is_pos(d, is_short, k) < inv_pos)
            plan.is_mask |= 1ull << k;
        else if (mid_side)
            plan.ms_mask |= 1ull << k;
    };
    auto nz = [&](int k) { return (nzmask >> k) & 1ull; };
    if (!is_short) {
        for (int i = 21; i >= 0; --i) {
            const int start = e.longb[i];
            if (!(start >= rzero1 || !nz(i))) break;
            zero_band(i);
            plan.bound = start;
        }
    } else {
        const int16_t *bands = is_mixed ? e.mixed : e.shortb;
        const int n_edges = is_mixed ? e.mixed_len : 40, sw = is_mixed ? e.mixed_switch : 0;
        const int n_groups = (n_edges - sw - 3 + 2) / 3;  // groups of three windows (stereo.rs:379-386)
        bool wz0 = true, wz1 = true, wz2 = true, found_bound = false;
        for (int gi = n_groups - 1; gi >= 0; --gi) {
            const int k0 = sw + 3 * gi;  // bands k0, k0 + 1, k0 + 2 = windows 0, 1, 2
#pragma unroll
            for (int w = 2; w >= 0; --w) {
                const int k = k0 + w;
                bool &wz = w == 2 ? wz2 : (w == 1 ? wz1 : wz0);
                wz = wz && !nz(k);
                if (wz)
                    zero_band(k);
                else if (mid_side)
                    plan.ms_mask |= 1ull << k;
            }
            plan.bound = bands[k0];
            found_bound = !wz0 && !wz1 && !wz2;
            if (found_bound) break;
        }
        if (!found_bound && is_mixed) {  // the long bands of a mixed block, stereo.rs:450-478
            for (int i = sw - 1; i >= 0; --i) {
                if (nz(i)) break;
                zero_band(i);
                plan.bound = bands[i];
            }
        }
    }
    return plan;
}

constexpr int kMp3StNone = 0, kMp3StMidSide = 1, kMp3StIntensity = 2;

// Lane k < 40 turns the plan's bit k into band k's action and, for an intensity band, its (left, right) ratios
// (process_intensity, stereo.rs:165-186), for mp3_stereo_apply to read back per line.  `ratios`: DevTables::mp3_is_ratios.
__device__ __forceinline__ void mp3_stereo_expand(const Mp3StereoPlan &plan, const symaccel_mp3_stereo &d, const float *ratios, int k,
                                                  int *act, float *kl, float *kr) {
    const unsigned long long bit = 1ull << k;
    int a = kMp3StNone;
    if (plan.is_mask & bit) {
        const int table = (d.flags & SYMACCEL_MP3_ST_MPEG1) ? 0 : 7 + 32 * ((d.flags & SYMACCEL_MP3_ST_IS_SCALE) ? 1 : 0);
        const int is_pos = mp3_is_pos(d, d.block_type == SYMACCEL_MP3_SHORT, k);
        kl[k] = ratios[2 * (table + is_pos)];
        kr[k] = ratios[2 * (table + is_pos) + 1];
        a = kMp3StIntensity;
    } else if (plan.ms_mask & bit) {
        a = kMp3StMidSide;
    }
    act[k] = a;
}

// One line of the pair: mid/side below the intensity bound (stereo.rs:541-543), the band's action from it on.
// Returns true when (a, b) changed.
__device__ __forceinline__ bool mp3_stereo_apply(float &a, float &b, int line, int bound, bool mid_side, bool intensity, int band,
                                                 const int *act, const float *kl, const float *kr) {
    int action = kMp3StNone;
    if (line < bound)
        action = mid_side ? kMp3StMidSide : kMp3StNone;
    else if (intensity)
        action = act[band];
    if (action == kMp3StMidSide) {  // process_mid_side (stereo.rs:139-148)
        const float left = (a + b) * kMp3Frac1Sqrt2, right = (a - b) * kMp3Frac1Sqrt2;
        a = left;
        b = right;
        return true;
    }
    if (action == kMp3StIntensity) {  // process_intensity (stereo.rs:174-180)
        const float is = a;
        a = kl[band] * is;
        b = kr[band] * is;
        return true;
    }
    return false;
}

}  // namespace symaccel
