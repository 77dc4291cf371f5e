// One-lane-per-block tile machinery shared by the integer predictor kernels (flac.hip, alac.hip): a wavefront owns
// 64 blocks (subframes / element channels), one per lane, and walks them in tiles of 32 samples staged through LDS.
#pragma once

#include <hip/hip_runtime.h>

#include <climits>
#include <cstdint>

#include "symaccel_internal.h"

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
                                                 unsigned blocksize, unsigned t0, int lane) {
    const int q = lane & 7, rsub = lane >> 3;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const int r = 8 * k + rsub;
        p.v[k] = *reinterpret_cast<const int4 *>(buf + (blk0 + (size_t)r) * blocksize + t0 + 4u * (unsigned)q);
    }
}
__device__ __forceinline__ void tile_commit(const TilePrefetch &p, int32_t *tile, int lane) {
    const int q = lane & 7, rsub = lane >> 3;
#pragma unroll
    for (int k = 0; k < 8; ++k) *reinterpret_cast<int4 *>(tile + (8 * k + rsub) * kStride + 4 * q) = p.v[k];
}
__device__ __forceinline__ void tile_store_fast(int32_t *__restrict__ buf, const int32_t *tile, size_t blk0,
                                                unsigned blocksize, unsigned t0, int lane) {
    const int q = lane & 7, rsub = lane >> 3;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const int r = 8 * k + rsub;
        *reinterpret_cast<int4 *>(buf + (blk0 + (size_t)r) * blocksize + t0 + 4u * (unsigned)q) =
            *reinterpret_cast<const int4 *>(tile + r * kStride + 4 * q);
    }
}
// Ragged tiles (last columns of a block size that is not a multiple of 32, unaligned rows, last subframes).
__device__ __forceinline__ void tile_fetch_slow(const int32_t *__restrict__ buf, int32_t *tile, size_t blk0, size_t n_blocks,
                                                unsigned blocksize, unsigned t0, unsigned cols, int lane) {
    const int c = lane & 31, rsub = lane >> 5;
#pragma unroll 1
    for (int r0 = 0; r0 < kRows; r0 += 16) {
        int32_t v[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const size_t b = blk0 + (size_t)(r0 + 2 * k + rsub);
            v[k] = (b < n_blocks && (unsigned)c < cols) ? buf[b * blocksize + t0 + (unsigned)c] : 0;
        }
#pragma unroll
        for (int k = 0; k < 8; ++k) tile[(r0 + 2 * k + rsub) * kStride + c] = v[k];
    }
}
__device__ __forceinline__ void tile_store_slow(int32_t *__restrict__ buf, const int32_t *tile, size_t blk0, size_t n_blocks,
                                                unsigned blocksize, unsigned t0, unsigned cols, int lane) {
    const int c = lane & 31, rsub = lane >> 5;
#pragma unroll 4
    for (int r = rsub; r < kRows; r += 2) {
        if (blk0 + (size_t)r < n_blocks && (unsigned)c < cols)
            buf[(blk0 + (size_t)r) * blocksize + t0 + (unsigned)c] = tile[r * kStride + c];
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
