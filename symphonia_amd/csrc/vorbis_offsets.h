// Shared by the three Vorbis synthesis kernels: the packed offsets and the fused floor x residue load.
//
// Packed Vorbis offsets from the block flags alone (lib.rs:296-331: block k reads n_k / 2 lines and yields (n_{k-1} + n_k) / 4
// samples): every wavefront derives the offsets of its segment's first block itself, so no scan kernel runs in front of the
// synthesis kernels (vorbis_wave.hip, vorbis_wave2.hip, vorbis_wg.hip; the LDS-staged generic kernel of vorbis.hip keeps its scan).
#pragma once

#include "dsp_device.h"

namespace symaccel {

namespace {

// Number of long blocks among the first `b` blocks of a chain (flags: one byte per block, non-zero = long).
// 1024 flags per step: each lane takes an aligned group of 16 (flags outside [0, b) are masked off).
__device__ __forceinline__ unsigned count_long_before(const uint8_t *f, long b, int lane) {
    unsigned cnt = 0;  // per-lane partial count, reduced once at the end
    const long mis = (long)(reinterpret_cast<uintptr_t>(f) & 15u);  // f - mis is 16-byte aligned
    const uint4 *base = reinterpret_cast<const uint4 *>(f - mis);
    for (long i0 = -mis; i0 < b; i0 += 1024) {
        const long i = i0 + 16 * lane;  // index of this lane's first flag
        if (i < b && i + 16 > 0) {
            // One aligned 16-byte load per lane.  In the ragged first / last group the load also covers bytes outside
            // [0, b): they share an aligned 16-byte unit with a byte that is inside, so the access cannot fault, and
            // they are masked off below (a byte-wise tail would be 16 dependent loads, each waited for in turn).
            const uint4 v = base[(i + mis) >> 4];
            const unsigned w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                const bool inside = i + q >= 0 && i + q < b;
                cnt += (inside && ((w[q >> 2] >> (8 * (q & 3))) & 255u)) ? 1u : 0u;
            }
        }
    }
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) cnt += (unsigned)__shfl_xor((int)cnt, m);
    return cnt;
}

// Index of the last long block among the first `b` blocks of a chain, or -1: the same 1024-flags-per-step scan, from
// the top (the answer can be thousands of blocks back; a flag-by-flag walk would be that many dependent loads).
__device__ __forceinline__ long last_long_before(const uint8_t *f, long b, int lane) {
    const long mis = (long)(reinterpret_cast<uintptr_t>(f) & 15u);
    const uint4 *base = reinterpret_cast<const uint4 *>(f - mis);
    for (long i0 = ((b - 1 + mis) / 1024) * 1024 - mis; i0 >= -mis; i0 -= 1024) {
        const long i = i0 + 16 * lane;
        long best = -1;
        if (i < b && i + 16 > 0) {
            const uint4 v = base[(i + mis) >> 4];
            const unsigned w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                const bool inside = i + q >= 0 && i + q < b;
                if (inside && ((w[q >> 2] >> (8 * (q & 3))) & 255u)) best = i + q;  // ascending q: the last hit stays
            }
        }
        int hi = (int)best;  // < 2^31 blocks per chain (launch_vorbis_wave)
#pragma unroll
        for (int m = 32; m >= 1; m >>= 1) {
            const int o = __shfl_xor(hi, m);
            hi = o > hi ? o : hi;
        }
        if (hi >= 0) return hi;
    }
    return -1;
}

// Packed offsets of block b of a chain whose blocks are bs0 (flag 0) or bs1 (flag 1) samples long: with S(b) = the sum of the first b
// block sizes = bs0 b + (bs1 - bs0) L(b), L(b) = long blocks before b,
//   spectrum offset = S(b) / 2,   PCM offset = (n_{-1} + S(b - 1) + S(b)) / 4,
// where n_{-1} is the size the first block is paired with (its own when there is no previous block, lib.rs:298).
struct VorbisPackedAt {
    uint32_t spec, pcm;
};
// (out of line for the kernels that have no registers to spare -- it runs once per segment, in front of everything else)
__device__ __attribute__((noinline)) unsigned count_long_before_call(const uint8_t *f, long b, int lane) { return count_long_before(f, b, lane); }
template <bool INLINE = true>
__device__ __forceinline__ uint32_t vorbis_sizes_before(const uint8_t *f, long b, int bs0, int bs1, int lane) {
    return (uint32_t)bs0 * (uint32_t)b + (uint32_t)(bs1 - bs0) * (INLINE ? count_long_before(f, b, lane) : count_long_before_call(f, b, lane));
}
template <bool INLINE = true>
__device__ __forceinline__ VorbisPackedAt vorbis_packed_at(const uint8_t *f, long b, int pf0, int bs0, int bs1, int lane) {
    VorbisPackedAt at = {0u, 0u};
    if (b > 0) {
        const uint32_t s_b = vorbis_sizes_before<INLINE>(f, b, bs0, bs1, lane);
        const uint32_t n_prev = (uint32_t)(f[b - 1] ? bs1 : bs0);
        const uint32_t n_m1 = (uint32_t)(pf0 < 0 ? (f[0] ? bs1 : bs0) : (pf0 ? bs1 : bs0));
        at.spec = s_b / 2u;
        at.pcm = (s_b + n_m1 + (s_b - n_prev)) / 4u;
    }
    return at;
}

// FUSED (all three synthesis kernels): 0 = `spectra` is the spectrum; 1 = `spectra` is the floor curve and `residue` the residue, both
// f32 (lib.rs:289-291: *f *= r as the lines are loaded); 2 = `spectra` is the RESIDUE and `residue` the floor curve as dB-table
// indices, one byte per line (symaccel_vorbis_floor1_y_device): the four bytes that belong to a lane's float4 are one 32-bit load.
template <int FUSED>
__device__ __forceinline__ const float *res_at(const float *rp, size_t lines) {
    if constexpr (FUSED == 2) return reinterpret_cast<const float *>(reinterpret_cast<const uint8_t *>(rp) + lines);
    else return FUSED ? rp + lines : nullptr;
}
// x *= table[y] for the four lines of a float4 (floor.rs:822 + lib.rs:289-291)
__device__ __forceinline__ void mul_floor_y(float4 &x, uint32_t yb, const float *dbt) {
    x.x = dbt[yb & 255u] * x.x;
    x.y = dbt[(yb >> 8) & 255u] * x.y;
    x.z = dbt[(yb >> 16) & 255u] * x.z;
    x.w = dbt[yb >> 24] * x.w;
}

}  // namespace

}  // namespace symaccel
