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

// (1.0 - 2.0 * sign_bit) * POW43[|s|] is +-POW43[|s|] exactly; zeros and the rzero partition are +0.0
__device__ __forceinline__ float mp3_sample_value(const DevTables &tb, const float *pow43_lo, int s, bool in_rzero) {
    int mag = s < 0 ? -s : s;
    mag = mag > 8206 ? 8206 : mag;
    const float p = mag < kMp3PowLds ? pow43_lo[mag] : tb.mp3_pow43[mag];
    return (in_rzero || s == 0) ? 0.0f : (s < 0 ? -p : p);
}

}  // namespace symaccel
