// One-lane-per-block tile machinery shared by the integer predictor kernels (flac.hip, alac.hip): a wavefront owns
// 64 blocks (subframes / element channels), one per lane, and walks them in tiles of 32 samples staged through LDS.
#pragma once

#include <hip/hip_runtime.h>

#include <climits>
#include <cstdint>

#include "dsp_device.h"

namespace symaccel {

constexpr int kRows = 64;           // subframes per wavefront (one per lane)
constexpr int kCols = 32;           // samples per tile
constexpr int kStride = kCols + 4;  // 36 words: rows stay 16-byte aligned; one-row-per-lane b128 access is conflict-free
constexpr int kTileWords = kRows * kStride;

// Tile I/O.  A tile is 64 subframes x 32 samples.  Fast path (full tile, 16-byte aligned rows): every lane moves
// int4, one wavefront instruction covers eight 128-byte row segments; the loads of tile k+1 are issued BEFORE the
// recurrence runs over tile k and land in registers while it computes (the recurrence alone keeps the FP64 pipe at
// ~90 %; un-overlapped tile traffic cost another 40 % of wall time).  Two LDS tiles alternate.
struct TilePrefetch {
    int4 v[8];
};
__device__ __forceinline__ void tile_issue_loads(const int32_t *__restrict__ buf, TilePrefetch &p, size_t blk0,
                                                 unsigned stride, unsigned t0, int lane) {
    const int q = lane & 7, rsub = lane >> 3;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const int r = 8 * k + rsub;
        p.v[k] = ld_stream(reinterpret_cast<const int4 *>(buf + (blk0 + (size_t)r) * stride + t0 + 4u * (unsigned)q));
    }
}
__device__ __forceinline__ void tile_commit(const TilePrefetch &p, int32_t *tile, int lane) {
    const int q = lane & 7, rsub = lane >> 3;
#pragma unroll
    for (int k = 0; k < 8; ++k) *reinterpret_cast<int4 *>(tile + (8 * k + rsub) * kStride + 4 * q) = p.v[k];
}
__device__ __forceinline__ void tile_store_fast(int32_t *__restrict__ buf, const int32_t *tile, size_t blk0,
                                                unsigned stride, unsigned t0, int lane) {
    const int q = lane & 7, rsub = lane >> 3;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const int r = 8 * k + rsub;
        st_stream(reinterpret_cast<int4 *>(buf + (blk0 + (size_t)r) * stride + t0 + 4u * (unsigned)q),
                  *reinterpret_cast<const int4 *>(tile + r * kStride + 4 * q));
    }
}
// Ragged tiles (last columns of a block size that is not a multiple of 32, unaligned rows, last subframes).
__device__ __forceinline__ void tile_fetch_slow(const int32_t *__restrict__ buf, int32_t *tile, size_t blk0, size_t n_blocks,
                                                unsigned stride, unsigned t0, unsigned cols, int lane) {
    const int c = lane & 31, rsub = lane >> 5;
#pragma unroll 1
    for (int r0 = 0; r0 < kRows; r0 += 16) {
        int32_t v[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const size_t b = blk0 + (size_t)(r0 + 2 * k + rsub);
            v[k] = (b < n_blocks && (unsigned)c < cols) ? buf[b * stride + t0 + (unsigned)c] : 0;
        }
#pragma unroll
        for (int k = 0; k < 8; ++k) tile[(r0 + 2 * k + rsub) * kStride + c] = v[k];
    }
}
__device__ __forceinline__ void tile_store_slow(int32_t *__restrict__ buf, const int32_t *tile, size_t blk0, size_t n_blocks,
                                                unsigned stride, unsigned t0, unsigned cols, int lane) {
    const int c = lane & 31, rsub = lane >> 5;
#pragma unroll 4
    for (int r = rsub; r < kRows; r += 2) {
        if (blk0 + (size_t)r < n_blocks && (unsigned)c < cols)
            buf[(blk0 + (size_t)r) * stride + t0 + (unsigned)c] = tile[r * kStride + c];
    }
}

// Stereo decorrelation of one restored sample (flac decoder.rs:32-82) followed by the left-justification shift
// (decoder.rs:239-242).  `own` is the sample of this row's channel, `other` the same sample of the pair's other channel;
// rows 2p / 2p+1 of a wavefront's tile are channel 0 / 1 of pair p.  mode: 0 independent, 1 left/side, 2 mid/side,
// 3 right/side (ch0 = side, ch1 = right).
__device__ __forceinline__ int32_t flac_decorrelated(unsigned mode, bool is_ch1, int32_t own, int32_t other, uint32_t out_shift) {
    const int32_t a = is_ch1 ? other : own, b = is_ch1 ? own : other;  // (ch0, ch1) as decoded
    int32_t v = own;
    if (mode == 1) {  // left/side: side = left - side
        v = is_ch1 ? (int32_t)((uint32_t)a - (uint32_t)b) : a;
    } else if (mode == 2) {  // mid/side
        const int32_t mid = (int32_t)(((uint32_t)a << 1) | ((uint32_t)b & 1u));
        v = is_ch1 ? ((int32_t)((uint32_t)mid - (uint32_t)b) >> 1) : ((int32_t)((uint32_t)mid + (uint32_t)b) >> 1);
    } else if (mode == 3) {  // right/side: side += right
        v = is_ch1 ? b : (int32_t)((uint32_t)a + (uint32_t)b);
    }
    return (int32_t)((uint32_t)v << out_shift);
}
// Both channels of one restored sample pair at once, without a branch: the four modes as three 0 / -1 masks of the pair's mode.
// (Written per row with `if (mode == ...)`, hipcc built a tree of divergent branches per SAMPLE -- a dozen scalar and branch
// instructions each, four samples per row segment; round 5.)  a / b = channel 0 / 1 as decoded; decoder.rs:32-82.
__device__ __forceinline__ void flac_decorrelate_pair(int32_t a, int32_t b, int32_t is1, int32_t is2, int32_t is3, uint32_t out_shift,
                                                      int32_t &o0, int32_t &o1) {
    const uint32_t ua = (uint32_t)a, ub = (uint32_t)b;
    const uint32_t d = ua - ub;                                         // left/side: right = left - side
    const uint32_t l0 = ua + (ub & (uint32_t)is3);                      // right/side: left = side + right; else channel 0 as it is
    const uint32_t l1 = ub ^ ((ub ^ d) & (uint32_t)is1);                // left/side: channel 1 = left - side; else as it is
    const uint32_t mid = (ua << 1) | (ub & 1u);                         // mid/side (decoder.rs:60-72)
    const uint32_t h0 = (uint32_t)((int32_t)(mid + ub) >> 1), h1 = (uint32_t)((int32_t)(mid - ub) >> 1);
    o0 = (int32_t)((l0 ^ ((l0 ^ h0) & (uint32_t)is2)) << out_shift);
    o1 = (int32_t)((l1 ^ ((l1 ^ h1) & (uint32_t)is2)) << out_shift);
}
// Tile write-back with the decorrelation fused in: a lane takes four columns of BOTH rows of a pair (32 pairs per tile: four
// rounds of eight pairs x eight column groups), so each restored sample is read from LDS once and the pair's arithmetic is shared.
// `row_mode[r]` = mode of the pair row r belongs to.
// SYM_FLAC_STORE_SWITCH (build knob, default 1): the pair's mode is constant over the lane's EIGHT samples of a round, so the round
// branches on it once -- an independent pair costs two shifts per sample pair, a mid/side pair nine instructions, instead of the 21 of
// the branch-free form for every pair (round 5's form, kept as 0: it wins only when all four modes meet in most wavefront rounds).
// The branch is per ROUND, not per sample: round 5 removed a compiler-built tree of divergent branches per sample.
// Measured: with rows 16 KiB apart both forms take the same 7.65 ms for config 5 (the row pitch was the bound: profiles/r06l..); on rows at
// symaccel_row_stride() the switch is 1.2 % ahead, 6.89 against 6.975 ms, twice in alternation (profiles/r06zz33_flac_variants_ab.txt) -- default since.
#ifndef SYM_FLAC_STORE_SWITCH
#define SYM_FLAC_STORE_SWITCH 1
#endif
__device__ __forceinline__ void tile_store_decorrelate_fast(int32_t *__restrict__ buf, const int32_t *tile,
                                                            const uint8_t *row_mode, uint32_t out_shift, size_t blk0,
                                                            unsigned stride, unsigned t0, int lane) {
    const int q = lane & 7, psub = lane >> 3;
#pragma unroll 1
    for (int k = 0; k < 4; ++k) {  // (not unrolled: the FP64 kernel has no register to spare for two rounds in flight)
        const int r0 = 2 * (8 * k + psub);
        const int4 a = *reinterpret_cast<const int4 *>(tile + r0 * kStride + 4 * q);
        const int4 b = *reinterpret_cast<const int4 *>(tile + (r0 + 1) * kStride + 4 * q);
        const int32_t m = (int32_t)row_mode[r0];
        int4 o0, o1;
#if SYM_FLAC_STORE_SWITCH
        uint32_t x0[4] = {(uint32_t)a.x, (uint32_t)a.y, (uint32_t)a.z, (uint32_t)a.w}, x1[4] = {(uint32_t)b.x, (uint32_t)b.y, (uint32_t)b.z, (uint32_t)b.w};
        if (m == 2) {  // mid/side (decoder.rs:60-72)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const uint32_t mid = (x0[i] << 1) | (x1[i] & 1u), sd = x1[i];
                x0[i] = (uint32_t)((int32_t)(mid + sd) >> 1);
                x1[i] = (uint32_t)((int32_t)(mid - sd) >> 1);
            }
        } else if (m == 1) {  // left/side: right = left - side
#pragma unroll
            for (int i = 0; i < 4; ++i) x1[i] = x0[i] - x1[i];
        } else if (m == 3) {  // right/side: left = side + right
#pragma unroll
            for (int i = 0; i < 4; ++i) x0[i] = x0[i] + x1[i];
        }
        o0 = make_int4((int32_t)(x0[0] << out_shift), (int32_t)(x0[1] << out_shift), (int32_t)(x0[2] << out_shift), (int32_t)(x0[3] << out_shift));
        o1 = make_int4((int32_t)(x1[0] << out_shift), (int32_t)(x1[1] << out_shift), (int32_t)(x1[2] << out_shift), (int32_t)(x1[3] << out_shift));
#else
        // mode == k as 0 / -1: (m ^ k) - 1 is negative only for m == k (m is 0 .. 3)
        const int32_t is1 = ((m ^ 1) - 1) >> 31, is2 = ((m ^ 2) - 1) >> 31, is3 = ((m ^ 3) - 1) >> 31;
        flac_decorrelate_pair(a.x, b.x, is1, is2, is3, out_shift, o0.x, o1.x);
        flac_decorrelate_pair(a.y, b.y, is1, is2, is3, out_shift, o0.y, o1.y);
        flac_decorrelate_pair(a.z, b.z, is1, is2, is3, out_shift, o0.z, o1.z);
        flac_decorrelate_pair(a.w, b.w, is1, is2, is3, out_shift, o0.w, o1.w);
#endif
        int32_t *dst = buf + (blk0 + (size_t)r0) * stride + t0 + 4u * (unsigned)q;
        st_stream(reinterpret_cast<int4 *>(dst), o0);
        st_stream(reinterpret_cast<int4 *>(dst + stride), o1);
    }
}
__device__ __forceinline__ void tile_store_decorrelate_slow(int32_t *__restrict__ buf, const int32_t *tile,
                                                            const uint8_t *row_mode, uint32_t out_shift, size_t blk0,
                                                            size_t n_blocks, unsigned stride, unsigned t0, unsigned cols,
                                                            int lane) {
    const int c = lane & 31, rsub = lane >> 5;
#pragma unroll 4
    for (int r = rsub; r < kRows; r += 2) {
        if (blk0 + (size_t)r < n_blocks && (unsigned)c < cols)
            buf[(blk0 + (size_t)r) * stride + t0 + (unsigned)c] =
                flac_decorrelated(row_mode[r], (r & 1) != 0, tile[r * kStride + c], tile[(r ^ 1) * kStride + c], out_shift);
    }
}

// The workgroup is one wavefront: order its LDS traffic with wavefront-scope fences only.  (__syncthreads() carries a
// workgroup-scope release, which makes the wavefront wait for its outstanding GLOBAL stores at every tile.)
__device__ __forceinline__ void wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

__device__ __forceinline__ int32_t wrap_add(int32_t a, int32_t b) { return (int32_t)((uint32_t)a + (uint32_t)b); }

__device__ __forceinline__ unsigned wave_max(unsigned v) {
    for (int m = 32; m >= 1; m >>= 1) {
        const unsigned o = (unsigned)__shfl_xor((int)v, m);
        v = v > o ? v : o;
    }
    return v;
}

}  // namespace symaccel
