// Polyphase synthesis building blocks shared by the Layer III kernel (mp3.hip) and the Layer I / II kernel
// (mpa_polyphase.hip): the 32-point DCT of synthesis.rs:348-844 and the V-vector mapping of synthesis.rs:247-263.
#pragma once

#include "dsp_device.h"
#include "mp3_literals.h"

namespace symaccel {

constexpr int kSStride = 36;   // row stride (floats) of the slot-major transpose S[slot][32]: b128-aligned and
                               // conflict-free for one-row-per-lane b128 access
constexpr int kHistOld = 16;   // previous time slots kept (15 are read by the window, the 16th completes v_vec)
constexpr int kDwStride = 20;  // window-coefficient rows [sample i][16], padded: conflict-free b128
constexpr int kDwFloats = 32 * kDwStride;

// Order this wavefront's LDS accesses: its lanes exchange data through LDS; the hardware executes one
// wavefront's DS instructions in order, the fences stop the compiler from reordering them.
__device__ __forceinline__ void wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// ---- dct32 (B.G. Lee), synthesis.rs:348-844, as the recursion the reference flattens ---------
template <int N>
__device__ __forceinline__ void dct_lee(float *x) {
    constexpr const float *mc = kMp3Lit;
    if constexpr (N == 2) {
        const float a = x[0] + x[1], b = (x[0] - x[1]) * mc[MP3C_COS1];
        x[0] = a;
        x[1] = b;
    } else {
        constexpr int H = N / 2;
        constexpr int cofs = N == 32 ? MP3C_COS16 : N == 16 ? MP3C_COS8 : N == 8 ? MP3C_COS4 : MP3C_COS2;
        float t[N];
#pragma unroll
        for (int i = 0; i < H; ++i) {
            t[i] = x[i] + x[N - 1 - i];
            t[H + i] = (x[i] - x[N - 1 - i]) * mc[cofs + i];
        }
        dct_lee<H>(t);
        dct_lee<H>(t + H);
#pragma unroll
        for (int i = 0; i < H - 1; ++i) {
            x[2 * i] = t[i];
            x[2 * i + 1] = t[H + i] + t[H + i + 1];
        }
        x[N - 2] = t[H - 1];
        x[N - 1] = t[N - 1];
    }
}

// V-row entries as +-copies of the dct32 output row d (synthesis.rs:247-263).
// first half  V[i]      : i=0 d[16] | 1..15 d[16+i] | 16 -> 0.0 | 17..31 -d[48-i]
// second half V[32 + i] : i=0 -d[16] | 1..15 -d[16-i] | 16 -d[0] | 17..31 -d[i-16]
struct VMap {
    int fidx, sidx;  // source index into d
    int fkind;       // 0: +d, 1: -d, 2: literal 0.0
};
__device__ __forceinline__ VMap vmap(int i) {
    VMap m;
    m.fidx = i == 0 ? 16 : (i < 16 ? 16 + i : (i == 16 ? 0 : 48 - i));
    m.fkind = i < 16 ? 0 : (i == 16 ? 2 : 1);
    m.sidx = i == 0 ? 16 : (i < 16 ? 16 - i : (i == 16 ? 0 : i - 16));
    return m;
}

// The same mapping with the sign as an XOR mask and the literal 0.0 of V[16] as a read of a zero column: the dct32
// pass stores 0.0 in columns 32..35 of every row it writes (kSStride = 36 leaves them free), so
//   V[i]      = bits(S[row][fcol]) ^ fsign      (negation is exact; +0.0 ^ 0 is the reference's literal +0.0)
//   V[32 + i] = bits(S[row][scol]) ^ 0x80000000
// -- one VALU instruction per entry instead of a compare/select chain, and two live registers less per lane.
struct VMapX {
    int fcol, scol;
    unsigned fsign;
};
__device__ __forceinline__ VMapX vmapx(int i) {
    const VMap m = vmap(i);
    VMapX x;
    x.fcol = m.fkind == 2 ? 32 : m.fidx;
    x.scol = m.sidx;
    x.fsign = m.fkind == 1 ? 0x80000000u : 0u;
    return x;
}

}  // namespace symaccel
