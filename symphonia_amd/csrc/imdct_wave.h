// Wavefront-level 1024-line / 128-line Imdct (512-point / 64-point complex FFT) shared by the AAC-LC kernel
// (Imdct::new_scaled(1024, 1/2048), (128, 1/256)) and the Vorbis 2048/256 kernel (Imdct::new(1024), (128)):
// symphonia-core/src/dsp/mdct.rs:67-146 over the radix-2 DIT graph of dsp/fft/no_simd.rs.
//
// One 64-lane wavefront owns one transform; its lanes exchange data through a private LDS work array of
// kWaveLds floats with wave-local ordering only (no workgroup barrier).  See aac.hip for the mapping.
#pragma once

#include "dsp_device.h"

namespace symaccel {

// SYM_LDS_ABLATE (measurement only, never in the product build: results are WRONG): bit mask of LDS phases to leave out or to turn into
// broadcasts, for attributing SQ_LDS_BANK_CONFLICT to a phase (tools/gpu_r6n.sh).  1 = mirror ds_bpermute of the long pre-twiddle,
// 2 = pre-twiddle twiddle reads as broadcasts, 4 = T1 exchange (pass 1 -> 2), 8 = T2 exchange (pass 2 -> 3), 16 = natural-order write of Z,
// 32 = post_slot's reads of Z as broadcasts, 64 = post_slot's twiddle reads as broadcasts, 128 = window reads as broadcasts,
// 256 = the dB-table look-ups of the byte-plane form as broadcasts.
#ifndef SYM_LDS_ABLATE
#define SYM_LDS_ABLATE 0
#endif
constexpr int kWaveLds = 2264;  // floats of private LDS per wavefront (FFT work array, see the T1/T2 layouts)

// Order this wavefront's LDS accesses (its lanes exchange data through LDS; the hardware executes
// one wavefront's DS instructions in order, the fences stop the compiler from reordering them).
__device__ __forceinline__ void wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// Workgroup barrier that orders LDS traffic ONLY.  __syncthreads() is a workgroup-scope fence over every address space: hipcc puts
// `s_waitcnt vmcnt(0)` in front of the s_barrier, i.e. every wavefront first waits for its outstanding global loads (the
// prefetched next frame) and for its PCM stores to reach the L2 -- a round trip to memory per barrier.  Wavefronts that exchange
// data through LDS alone need the LDS counter, nothing else.
__device__ __forceinline__ void wg_sync_lds() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}

// LDS complex index of element (B, j, k) = logical position 64B + 8j + k of the FFT work array.
// Both layouts are SEPARABLE -- lane base + per-instruction constant on the write AND the read side,
// so every ds instruction uses one address VGPR plus an immediate offset -- and conflict-free for
// ds_write_b64 / ds_read_b64 (integer carries do what an XOR swizzle would; tools/aac_wave_model.py).
// T1: pass-1 lanes (B, j) write k = 0..7, pass-2 lanes (B, k) read j = 0..7.
__device__ __forceinline__ int lds_t1_lane_w(int B, int j) { return B + 8 * (j >> 2) + 288 * (j & 3); }
__device__ __forceinline__ constexpr int lds_t1_inst_w(int k) { return 36 * k; }
__device__ __forceinline__ int lds_t1_lane_r(int B, int k) { return B + 36 * k; }
__device__ __forceinline__ constexpr int lds_t1_inst_r(int j) { return 8 * (j >> 2) + 288 * (j & 3); }
// T2: pass-2 lanes (B, k) write j = 0..7, pass-3 lanes k' = 8j + k read B = 0..7.
__device__ __forceinline__ int lds_t2_lane_w(int B, int k) { return k + 8 * (B & 1) + 64 * (B >> 1) + 256 * (B & 1); }
__device__ __forceinline__ constexpr int lds_t2_inst_w(int j) { return 8 * j; }
__device__ __forceinline__ int lds_t2_lane_r(int j, int k) { return k + 8 * j; }
__device__ __forceinline__ constexpr int lds_t2_inst_r(int B) { return 8 * (B & 1) + 64 * (B >> 1) + 256 * (B & 1); }
static_assert(2 * (7 + 8 + 288 * 3 + 36 * 7 + 1) <= kWaveLds, "FFT work array must fit the per-wave LDS");
static_assert(128 * 7 + 96 + 128 + 1024 <= kWaveLds, "the skewed short-window rows and a parked 1024-float delay line must fit the per-wave LDS");

__device__ __forceinline__ c32 ld_c(const cpx *p) {
    const cpx v = *p;
    return c32{v.re, v.im};
}

struct LaneTables {
    // pass 2 (stages 4-6): fft16 combine k, fft32 combine k and k+8, merge W64[8j + k]
    c32 w16;
    int f16;
    c32 w32[2];
    int f32[2];
    c32 w64[4];
    // pass 3 (stages 7-9): W128[k'], W256[k' + 64 b], W512[k' + 64 B]
    c32 w128;
    c32 w256[2];
    c32 w512[4];
    __device__ __forceinline__ c32 W16() const { return w16; }
    __device__ __forceinline__ int F16() const { return f16; }
    __device__ __forceinline__ c32 W32(int i) const { return w32[i]; }
    __device__ __forceinline__ int F32(int i) const { return f32[i]; }
    __device__ __forceinline__ c32 W64(int j) const { return w64[j]; }
    __device__ __forceinline__ c32 W128() const { return w128; }
    __device__ __forceinline__ c32 W256(int b) const { return w256[b]; }
    __device__ __forceinline__ c32 W512(int B) const { return w512[B]; }
};

// The same twiddles read from an LDS copy at the point of use (504 complex values per workgroup): 31 VGPRs per lane less
// than LaneTables, for 14 conflict-free ds_read_b64 per 512-point transform -- what lets a kernel built around
// fft512_wave fit three wavefronts per SIMD.  Layout: small16[8] | small32[16] | W64[32] | W128[64] | W256[128] | W512[256].
constexpr int kLaneTabComplex = 8 + 16 + 32 + 64 + 128 + 256;
struct LaneTablesLds {
    const c32 *t;
    int lane, k, forms;  // forms: f16 | f32[0] << 2 | f32[1] << 4
    __device__ __forceinline__ c32 W16() const { return t[k]; }
    __device__ __forceinline__ int F16() const { return forms & 3; }
    __device__ __forceinline__ c32 W32(int i) const { return t[8 + k + 8 * i]; }
    __device__ __forceinline__ int F32(int i) const { return (forms >> (2 + 2 * i)) & 3; }
    __device__ __forceinline__ c32 W64(int j) const { return t[24 + 8 * j + k]; }
    __device__ __forceinline__ c32 W128() const { return t[56 + lane]; }
    __device__ __forceinline__ c32 W256(int b) const { return t[120 + lane + 64 * b]; }
    __device__ __forceinline__ c32 W512(int B) const { return t[248 + lane + 64 * B]; }
};
// Stages 4-6 from registers (17 VGPRs), stages 7-9 from the LDS copy (seven ds_read_b64 per 512-point transform): 14 VGPRs
// per lane less than LaneTables.  What the packed build of the Vorbis 256 / 2048 kernel needs to stay inside 256 VGPRs.
struct LaneTablesMixed {
    c32 w16;
    int f16;
    c32 w32[2];
    int f32[2];
    c32 w64[4];
    const c32 *t;
    int lane;
    __device__ __forceinline__ c32 W16() const { return w16; }
    __device__ __forceinline__ int F16() const { return f16; }
    __device__ __forceinline__ c32 W32(int i) const { return w32[i]; }
    __device__ __forceinline__ int F32(int i) const { return f32[i]; }
    __device__ __forceinline__ c32 W64(int j) const { return w64[j]; }
    __device__ __forceinline__ c32 W128() const { return t[56 + lane]; }
    __device__ __forceinline__ c32 W256(int b) const { return t[120 + lane + 64 * b]; }
    __device__ __forceinline__ c32 W512(int B) const { return t[248 + lane + 64 * B]; }
};
// workgroup-cooperative fill of the LDS copy (call before a __syncthreads)
__device__ __forceinline__ void fill_lane_tables_lds(const DevTables &tb, c32 *t, int tid, int n_threads) {
    for (int i = tid; i < kLaneTabComplex; i += n_threads) {
        const cpx *src = i < 8 ? tb.small16 + i : (i < 24 ? tb.small32 + (i - 8) : tb.fft_merge + (i - 24));
        t[i] = ld_c(src);
    }
}
__device__ __forceinline__ LaneTablesLds lane_tables_lds(const DevTables &tb, const c32 *t, int lane) {
    LaneTablesLds lt;
    lt.t = t;
    lt.lane = lane;
    lt.k = lane & 7;
    lt.forms = (int)tb.small16_form[lt.k] | ((int)tb.small32_form[lt.k] << 2) | ((int)tb.small32_form[lt.k + 8] << 4);
    return lt;
}


__device__ __forceinline__ void load_lane_tables(const DevTables &tb, int lane, LaneTables &t) {
    const int k = lane & 7;
    t.w16 = ld_c(tb.small16 + k);
    t.f16 = tb.small16_form[k];
    t.w32[0] = ld_c(tb.small32 + k);
    t.f32[0] = tb.small32_form[k];
    t.w32[1] = ld_c(tb.small32 + k + 8);
    t.f32[1] = tb.small32_form[k + 8];
    const cpx *w64 = tb.fft_merge + 0, *w128 = tb.fft_merge + 32, *w256 = tb.fft_merge + 96,
              *w512 = tb.fft_merge + 224;
#pragma unroll
    for (int j = 0; j < 4; ++j) t.w64[j] = ld_c(w64 + 8 * j + k);
    t.w128 = ld_c(w128 + lane);
    t.w256[0] = ld_c(w256 + lane);
    t.w256[1] = ld_c(w256 + lane + 64);
#pragma unroll
    for (int B = 0; B < 4; ++B) t.w512[B] = ld_c(w512 + lane + 64 * B);
}

__device__ __forceinline__ void load_lane_tables_mixed(const DevTables &tb, const c32 *lds_copy, int lane, LaneTablesMixed &t) {
    const int k = lane & 7;
    t.w16 = ld_c(tb.small16 + k);
    t.f16 = tb.small16_form[k];
    t.w32[0] = ld_c(tb.small32 + k);
    t.f32[0] = tb.small32_form[k];
    t.w32[1] = ld_c(tb.small32 + k + 8);
    t.f32[1] = tb.small32_form[k + 8];
#pragma unroll
    for (int j = 0; j < 4; ++j) t.w64[j] = ld_c(tb.fft_merge + 8 * j + k);
    t.t = lds_copy;
    t.lane = lane;
}

// Stages 4-6 of the radix-2 graph on the eight values u[j] = a[64B + 8j + k] of one lane.
template <class LT>
__device__ __forceinline__ void pass2_regs(c32 (&u)[8], const LT &t) {
    {
        const c32 w16 = t.W16();
        const int f16 = t.F16();
#pragma unroll
        for (int j = 0; j < 8; j += 2) bfly(u[j], u[j + 1], tw_small(u[j + 1], w16, f16));  // fft16 combine
    }
    {
        const c32 wa = t.W32(0), wb = t.W32(1);
        const int fa = t.F32(0), fb = t.F32(1);
#pragma unroll
        for (int h = 0; h < 8; h += 4) {                                                      // fft32 combine
            bfly(u[h + 0], u[h + 2], tw_small(u[h + 2], wa, fa));
            bfly(u[h + 1], u[h + 3], tw_small(u[h + 3], wb, fb));
        }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) bfly(u[j], u[j + 4], c_mul(u[j + 4], t.W64(j)));             // merge, step 32
}

// Stages 7-9 on v[B] = a[64B + k'].
template <class LT>
__device__ __forceinline__ void pass3_regs(c32 (&v)[8], const LT &t) {
    {
        const c32 w128 = t.W128();
#pragma unroll
        for (int B = 0; B < 8; B += 2) bfly(v[B], v[B + 1], c_mul(v[B + 1], w128));          // step 64
    }
    {
        const c32 wa = t.W256(0), wb = t.W256(1);
#pragma unroll
        for (int h = 0; h < 8; h += 4) {                                                      // step 128
            bfly(v[h + 0], v[h + 2], c_mul(v[h + 2], wa));
            bfly(v[h + 1], v[h + 3], c_mul(v[h + 3], wb));
        }
    }
#pragma unroll
    for (int B = 0; B < 4; ++B) bfly(v[B], v[B + 4], c_mul(v[B + 4], t.W512(B)));            // step 256
}

// The u[r] of pass 1 must be presented to fft8 in bit-reversed order: u[r] = z[.. rev3(r)].
__device__ __forceinline__ void bitrev8(c32 (&z)[8]) {
    c32 t = z[1];
    z[1] = z[4];
    z[4] = t;
    t = z[3];
    z[3] = z[6];
    z[6] = t;
}

// 512-point FFT of the pre-twiddled z[m + 64 s] held by lane m; leaves Z[0..512) in natural order
// in the wavefront's LDS (complex index = position).
template <class LT>
__device__ __forceinline__ void fft512_wave(c32 (&z)[8], int lane, c32 *lds, const LT &lt) {
    // ---- pass 1: fft8 over z[m + 64*rev3(r)] -> a[8*rev6(m) + r], i.e. element (B, j, k=r)
    bitrev8(z);
    fft8_regs(z);
    {
        const int B = (int)rev_bits((unsigned)lane & 7u, 3), j = (int)rev_bits((unsigned)lane >> 3, 3);
#if !(SYM_LDS_ABLATE & 4)
        c32 *w = lds + lds_t1_lane_w(B, j);
#pragma unroll
        for (int r = 0; r < 8; ++r) w[lds_t1_inst_w(r)] = z[r];
#endif
    }
    wave_sync();
    // ---- pass 2: lane (B, k) gathers j = 0..7
    const int B2 = lane >> 3, k2 = lane & 7;
    {
#if !(SYM_LDS_ABLATE & 4)
        const c32 *r = lds + lds_t1_lane_r(B2, k2);
#pragma unroll
        for (int j = 0; j < 8; ++j) z[j] = r[lds_t1_inst_r(j)];
#endif
    }
    pass2_regs(z, lt);
    wave_sync();
#if !(SYM_LDS_ABLATE & 8)
    {
        c32 *w = lds + lds_t2_lane_w(B2, k2);
#pragma unroll
        for (int j = 0; j < 8; ++j) w[lds_t2_inst_w(j)] = z[j];
    }
#endif
    wave_sync();
    // ---- pass 3: lane k' = 8j + k gathers B = 0..7
#if !(SYM_LDS_ABLATE & 8)
    {
        const c32 *r = lds + lds_t2_lane_r(lane >> 3, lane & 7);
#pragma unroll
        for (int B = 0; B < 8; ++B) z[B] = r[lds_t2_inst_r(B)];
    }
#endif
    pass3_regs(z, lt);
    wave_sync();
#if !(SYM_LDS_ABLATE & 16)
#pragma unroll
    for (int B = 0; B < 8; ++B) lds[64 * B + lane] = z[B];  // natural order Z[64B + k']
#endif
    wave_sync();
}

// ---- any power-of-two size from 16 to 512 points: 512 / P transforms in ONE pass of the same three register passes --------------
// A 512-point pass over the work array IS 512 / P independent P-point transforms laid side by side (transform T at positions
// T P .. T P + P - 1) as long as the radix-2 stages stop after log2 P of them: stages 1-3 and 4-6 work inside 64-position blocks,
// stages 7, 8, 9 pair positions 64, 128, 256 apart.  What changes with P is only (a) which inputs a lane brings: the DIT input order
// is the bit reversal over log2 P bits INSIDE a transform, so lane l = (T, u), u < P / 8, owns x_T[u + (P / 8) s], s = 0..7, and its
// fft8 lands at positions 8 g .. 8 g + 7 with g = T P / 8 + rev(u); and (b) where the passes stop.  Every butterfly is the one
// the reference's fft16 / fft32 / transform() does for that size (no_simd.rs:221-454): same operands, same twiddle values.
// In:  z[s] = x_T[u + (P / 8) s] for lane (T, u).
// Out: logp <= 6: z[j] = X at position 64 (lane >> 3) + 8 j + (lane & 7);   logp >= 7: z[B] = X at position 64 B + lane.
//      Position p belongs to transform p >> logp, bin p & (P - 1).  The work array is free again on return (after a wave_sync).
// T1M, the pass-1 -> pass-2 exchange layout of fft_wave_multi: element (B, j, k) at B + fj(j) + 36 k with
// fj(j) = 18 j0 + 100 j1 + 344 j2 (j's bits).  T1 is conflict-free only for the lane order of the 512-point transform; with P < 512
// the sixteen lanes of a ds_write_b64 group hold (2 B) x (8 j), (4 B) x (4 j of one parity) or (8 B) x (j, j + 4), and all three
// need fj(j) = 2 j (mod 16).  The weights are the smallest that keep the layout injective (largest index 7 + 462 + 252 = 721);
// tools/vorbis_lds_model.py: 32 + 16 LDS cycles per exchange for every P (T1: 128 + 16 for P <= 128, 64 + 16 for P = 256).
__device__ __forceinline__ int lds_t1m_lane_w(int B, int j) { return B + 18 * (j & 1) + 100 * ((j >> 1) & 1) + 344 * (j >> 2); }
__device__ __forceinline__ constexpr int lds_t1m_inst_r(int j) { return 18 * (j & 1) + 100 * ((j >> 1) & 1) + 344 * (j >> 2); }
static_assert(2 * (7 + 462 + 36 * 7 + 1) <= kWaveLds, "T1M must fit the per-wave LDS");

template <class LT>
__device__ __forceinline__ void fft_wave_multi(c32 (&z)[8], int lane, c32 *lds, const LT &lt, int logp) {
    bitrev8(z);
    fft8_regs(z);
    {
        const int gbits = logp - 3;  // 1 .. 6
        const int T = lane >> gbits, u = lane & ((1 << gbits) - 1);
        const int g = (T << gbits) + (int)rev_bits((unsigned)u, gbits);
        c32 *w = lds + lds_t1m_lane_w(g >> 3, g & 7);
#pragma unroll
        for (int r = 0; r < 8; ++r) w[lds_t1_inst_w(r)] = z[r];
    }
    wave_sync();
    const int B2 = lane >> 3, k2 = lane & 7;
    {
        const c32 *r = lds + lds_t1_lane_r(B2, k2);
#pragma unroll
        for (int j = 0; j < 8; ++j) z[j] = r[lds_t1m_inst_r(j)];
    }
    {   // stages 4-6, as many as the size has (wave-uniform branches)
        const c32 w16 = lt.W16();
        const int f16 = lt.F16();
#pragma unroll
        for (int j = 0; j < 8; j += 2) bfly(z[j], z[j + 1], tw_small(z[j + 1], w16, f16));  // fft16 combine
    }
    if (logp >= 5) {
        const c32 wa = lt.W32(0), wb = lt.W32(1);
        const int fa = lt.F32(0), fb = lt.F32(1);
#pragma unroll
        for (int h = 0; h < 8; h += 4) {  // fft32 combine
            bfly(z[h + 0], z[h + 2], tw_small(z[h + 2], wa, fa));
            bfly(z[h + 1], z[h + 3], tw_small(z[h + 3], wb, fb));
        }
    }
    if (logp >= 6) {
#pragma unroll
        for (int j = 0; j < 4; ++j) bfly(z[j], z[j + 4], c_mul(z[j + 4], lt.W64(j)));  // merge, step 32
    }
    wave_sync();
    if (logp <= 6) return;
    {
        c32 *w = lds + lds_t2_lane_w(B2, k2);
#pragma unroll
        for (int j = 0; j < 8; ++j) w[lds_t2_inst_w(j)] = z[j];
    }
    wave_sync();
    {
        const c32 *r = lds + lds_t2_lane_r(lane >> 3, lane & 7);
#pragma unroll
        for (int B = 0; B < 8; ++B) z[B] = r[lds_t2_inst_r(B)];
    }
    {
        const c32 w128 = lt.W128();
#pragma unroll
        for (int B = 0; B < 8; B += 2) bfly(z[B], z[B + 1], c_mul(z[B + 1], w128));  // step 64
    }
    if (logp >= 8) {
        const c32 wa = lt.W256(0), wb = lt.W256(1);
#pragma unroll
        for (int h = 0; h < 8; h += 4) {  // step 128
            bfly(z[h + 0], z[h + 2], c_mul(z[h + 2], wa));
            bfly(z[h + 1], z[h + 3], c_mul(z[h + 3], wb));
        }
    }
    if (logp >= 9) {
#pragma unroll
        for (int B = 0; B < 4; ++B) bfly(z[B], z[B + 4], c_mul(z[B + 4], lt.W512(B)));  // step 256
    }
    wave_sync();
}

// a group's input: 1024 floats, 16 B per lane and load (floats past `valid_floats` read as zero)
__device__ __forceinline__ void multi_fetch(const float *src, size_t valid_floats, int lane, float4 (&v)[4]) {
    const float4 *s4 = reinterpret_cast<const float4 *>(src);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int i4 = lane + 64 * q;
        v[q] = (size_t)(4 * i4) < valid_floats ? ld_stream(s4 + i4) : make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    }
}

// Post-twiddle (mdct.rs:94-137) of what fft_wave_multi left in the registers, scattered into `pcm`: transform T's 4 P output
// samples at pcm[4 P T ..] in natural order (vec0 | vec1 | vec2 | vec3 of P samples each).
__device__ __forceinline__ void multi_post_twiddle(const c32 (&z)[8], int lane, int logp, const c32 *tw, float *pcm) {
    const int P = 1 << logp, n4 = P >> 1;
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        const int p = logp <= 6 ? 64 * (lane >> 3) + 8 * q + (lane & 7) : 64 * q + lane;
        const int T = p >> logp, k = p & (P - 1);
        const c32 val = post_twiddle(z[q], tw[k]);
        float *vec0 = pcm + ((size_t)T << (logp + 2)), *vec1 = vec0 + P, *vec2 = vec1 + P, *vec3 = vec2 + P;
        if (k < n4) {
            const int fi = 2 * k, ri = P - 1 - 2 * k;
            vec0[ri] = -val.y;
            vec1[fi] = val.y;
            vec2[ri] = val.x;
            vec3[fi] = val.x;
        } else {
            const int i = k - n4;
            const int fi = 2 * i, ri = P - 1 - 2 * i;
            vec0[fi] = -val.x;
            vec1[ri] = val.x;
            vec2[fi] = val.y;
            vec3[ri] = val.y;
        }
    }
}
// Padding of the work area for transforms of up to 64 points (blocks of up to 256 samples), in floats per transform (input: 2 P
// lines) and per block (output: 4 P samples): 4 for P = 16, 8 for P = 32 and 64 (multiples of four keep the 16-byte accesses
// aligned).  tools/vorbis_lds_model.py: pre-twiddle reads 384 -> 48 LDS cycles (P = 16), output scatter 256 -> 128.
__device__ __forceinline__ constexpr int multi_pad(int logp) { return logp == 4 ? 4 : (logp <= 6 ? 8 : 0); }
static_assert(32 * (64 + multi_pad(4)) <= kWaveLds && 16 * (128 + multi_pad(5)) <= kWaveLds && 8 * (256 + multi_pad(6)) <= kWaveLds,
              "the padded outputs of a group must fit the per-wave LDS");

// The same scatter with the size known at compile time.  Position p = (part that depends on q) + (part that depends on the lane),
// and the two never carry into each other: for P >= 128, p = 64 q + lane; for P <= 64, p = 8 q + (64 (lane >> 3) + (lane & 7)).
// So T = Tq + Tl and k = kq + kl with Tq, kq compile-time, and `k < n4` is decided by kq alone: every store below is one
// ds_write_b32 from one of two per-lane base addresses (index rising / falling with the lane) plus an immediate offset.
// OPAD: floats between the blocks' outputs (block T at pcm + T (4 P + OPAD)): for P <= 64 the lanes of a store belong to several
// blocks, and with a stride of 4 P floats they all meet in the same banks (multi_pad below).
template <int LOGP, int OPAD = 0>
__device__ __forceinline__ void multi_post_twiddle_ct(const c32 (&z)[8], int lane, const c32 *tw, float *pcm) {
    constexpr int P = 1 << LOGP, n4 = P >> 1, g = LOGP <= 6 ? 8 : 64, ST = 4 * P + OPAD;
    const int kl = LOGP <= 6 ? (lane & 7) : lane;
    const int Tl = LOGP <= 6 ? ((lane >> 3) << (6 - LOGP)) : 0;
    float *up = pcm + Tl * ST + 2 * kl;                  // index 2 kl + c
    float *down = pcm + Tl * ST + (2 * g - 1 - 2 * kl);  // index c - 2 kl, rebased so that c - (2 g - 1) >= 0 below
    const c32 *twl = tw + kl;
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        const int pq = LOGP <= 6 ? 8 * q : 64 * q;
        const int kq = pq & (P - 1), Tq = pq >> LOGP;
        const c32 val = post_twiddle(z[q], twl[kq]);
        const int base = Tq * ST;
        if (kq < n4) {  // vec0[ri] = -y, vec1[fi] = y, vec2[ri] = x, vec3[fi] = x with fi = 2 k, ri = P - 1 - 2 k
            const int fi = 2 * kq, ri = P - 1 - 2 * kq - (2 * g - 1);
            down[base + ri] = -val.y;
            up[base + P + fi] = val.y;
            down[base + 2 * P + ri] = val.x;
            up[base + 3 * P + fi] = val.x;
        } else {        // vec0[fi] = -x, vec1[ri] = x, vec2[fi] = y, vec3[ri] = y with fi = 2 (k - n4)
            const int fi = 2 * (kq - n4), ri = P - 1 - 2 * (kq - n4) - (2 * g - 1);
            up[base + fi] = -val.x;
            down[base + P + ri] = val.x;
            up[base + 2 * P + fi] = val.y;
            down[base + 3 * P + ri] = val.y;
        }
    }
}
static_assert(2048 <= kWaveLds, "the Imdct output of a 512-point pass (2048 samples) must fit the per-wave LDS");

// Post-twiddle (mdct.rs:94-137) of the four FFT bins that feed output slot m2 = lane + 64h:
//   x[q]  = pcm[j(q)]         (first half of the 2048-sample IMDCT output)
//   x2[q] = pcm[1024 + j(q)]  with j(q) = 4*m2 + q for q < 4, 1020 - 4*m2 + (q - 4) for q >= 4.
// Twiddles tw[254-2m2 .. 255-2m2] and tw[256+2m2 .. 257+2m2] come from the shared LDS table.
__device__ __forceinline__ void post_slot(const c32 *lds, const c32 *tw, int m2, float (&x)[8], float (&x2)[8]) {
    const int mz = (SYM_LDS_ABLATE & 32) ? (m2 & 64) : m2, mt = (SYM_LDS_ABLATE & 64) ? (m2 & 64) : m2;
    const c32 vB = post_twiddle(lds[254 - 2 * mz], tw[254 - 2 * mt]);
    const c32 vA = post_twiddle(lds[255 - 2 * mz], tw[255 - 2 * mt]);
    const c32 vC = post_twiddle(lds[256 + 2 * mz], tw[256 + 2 * mt]);
    const c32 vD = post_twiddle(lds[257 + 2 * mz], tw[257 + 2 * mt]);
    x[0] = -vC.x;  // vec0[4m2 .. 4m2+3]
    x[1] = -vA.y;
    x[2] = -vD.x;
    x[3] = -vB.y;
    x[4] = vB.y;   // vec1[508-4m2 .. 511-4m2]
    x[5] = vD.x;
    x[6] = vA.y;
    x[7] = vC.x;
    x2[0] = vC.y;  // vec2[4m2 ..]
    x2[1] = vA.x;
    x2[2] = vD.y;
    x2[3] = vB.x;
    x2[4] = vB.x;  // vec3[508-4m2 ..]
    x2[5] = vD.y;
    x2[6] = vA.x;
    x2[7] = vC.y;
}

// Start of short window w's 128 floats in the per-wave LDS area, for the staged lines and for the half-stored output H.
// Rows exactly 128 floats apart would put the eight windows on the same banks: the transform's strided per-window
// reads and writes (32 ds_*_b32 instructions, lanes = (window, column)) would then be 4-way conflicted in each 32-lane
// group -- measured: 65 % of imdct128_wave_kernel's LDS cycles (57 % with the skew).  These multiples of four (rows stay 16-byte aligned
// for ys4; non-decreasing, so rows do not overlap; bank offsets 0, 16, 24, 8, 8, 24, 16, 0 mod 32) halve that;
// tests/models/lds_sim.py finds nothing better among aligned skews (tests/test_models.py).
__device__ __forceinline__ constexpr int short_row(int w) {
    // skews {0, 16, 24, 40, 40, 56, 80, 96} = 8 x the nibbles of 0xCA755320 (arithmetic, not a table: w is often a
    // per-lane run-time value, and a table would be a memory lookup in the middle of the transform)
    return 128 * w + 8 * (int)((0xCA755320u >> (4 * (w & 7))) & 15u);
}
static_assert(short_row(0) == 0 && short_row(1) == 144 && short_row(2) == 280 && short_row(3) == 424 && short_row(4) == 552 &&
              short_row(5) == 696 && short_row(6) == 848 && short_row(7) == 992, "short_row skews");
constexpr int kShortRowsEnd = 128 * 7 + 96 + 128;  // floats of per-wave LDS the eight rows span

// Eight 128-line IMDCTs (dsp.rs:80-83).  The frame's 1024 lines are staged at ldsf[short_row(w) ..].  Of each
// window's 256 outputs v0|v1|v2|v3 only v1 and v2 are kept: H[w][0..64) = v1, H[w][64..128) = v2 in
// ldsf[short_row(w) ..]; v0[x] = -v1[63-x] and v3[x] = v2[63-x] exactly (mdct.rs:108-136 writes the same
// value, negated for v0, to both).
template <class LT>
__device__ __forceinline__ void imdct_short_wave(int lane, float *ldsf, const cpx *tw_short, const LT &lt) {
    c32 *lds = reinterpret_cast<c32 *>(ldsf);
    // pass 1: lane (w, c) owns z_w[c + 8 s] = pre_twiddle(x_w[2i], x_w[127 - 2i]), i = c + 8 s
    const int w = lane >> 3, c = lane & 7;
    c32 z[8];
#pragma unroll
    for (int s = 0; s < 8; ++s) {
        const int i = c + 8 * s;
        z[s] = pre_twiddle(ldsf[short_row(w) + 2 * i], ldsf[short_row(w) + 127 - 2 * i], ld_c(tw_short + i));
    }
    wave_sync();
    bitrev8(z);
    fft8_regs(z);  // -> a_w[8*rev3(c) + r] = element (B = w, j = rev3(c), k = r)
    // (the exchange in the T1M layout: T1's is conflict-free only for the 512-point transform's lane order -- here the sixteen lanes
    // of a store hold two windows x eight j, whose T1 addresses fall on the same banks four at a time: 128 + 16 LDS cycles per
    // exchange against 32 + 16, tools/vorbis_lds_model.py)
    {
        const int j = (int)rev_bits((unsigned)c, 3);
        c32 *wp = lds + lds_t1m_lane_w(w, j);
#pragma unroll
        for (int r = 0; r < 8; ++r) wp[lds_t1_inst_w(r)] = z[r];
    }
    wave_sync();
    {
        const c32 *rp = lds + lds_t1_lane_r(w, c);
#pragma unroll
        for (int j = 0; j < 8; ++j) z[j] = rp[lds_t1m_inst_r(j)];
    }
    pass2_regs(z, lt);  // 64-point FFT done: z[j] = Z_w[8j + k], k = c
    wave_sync();
    // post-twiddle (mdct.rs:94-137 with n2 = 64, n4 = 32)
    float *o = ldsf + short_row(w);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int i = 8 * j + c;
        const c32 val = post_twiddle(z[j], ld_c(tw_short + i));
        if (j < 4) {
            o[2 * i] = val.y;            // v1[fi]   (v0[ri] = -val.y is its mirror)
            o[64 + 63 - 2 * i] = val.x;  // v2[ri]   (v3[fi] = val.x is its mirror)
        } else {
            const int i2 = i - 32;
            o[63 - 2 * i2] = val.x;      // v1[ri]   (v0[fi] = -val.x)
            o[64 + 2 * i2] = val.y;      // v2[fi]   (v3[ri] = val.y)
        }
    }
    wave_sync();
}

__device__ __forceinline__ void store_slot(float *frame, int m2, const float (&v)[8]) {
    float4 *o4 = reinterpret_cast<float4 *>(frame);
    o4[m2] = make_float4(v[0], v[1], v[2], v[3]);
    o4[255 - m2] = make_float4(v[4], v[5], v[6], v[7]);
}
__device__ __forceinline__ void store_slot_stream(float *frame, int m2, const float (&v)[8]) {
    float4 *o4 = reinterpret_cast<float4 *>(frame);
    st_stream(o4 + m2, make_float4(v[0], v[1], v[2], v[3]));
    st_stream(o4 + 255 - m2, make_float4(v[4], v[5], v[6], v[7]));
}
__device__ __forceinline__ void load_slot(const float *frame, int m2, float (&v)[8]) {
    const float4 *i4 = reinterpret_cast<const float4 *>(frame);
    const float4 a = i4[m2], b = i4[255 - m2];
    v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w;
    v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
}


// y_s[i0 .. i0+3] (i0 a multiple of 4, in 0..256): Imdct output of short block `w` from the half-stored H of imdct_short_wave: v0 = -reverse(v1), v1 = H[0..64), v2 = H[64..128), v3 = reverse(v2).
__device__ __forceinline__ void ys4(const float *H, int w, int i0, float (&v)[4]) {
    const float *h = H + short_row(w);
    if (i0 < 64) {
        const float4 r = *reinterpret_cast<const float4 *>(h + 60 - i0);
        v[0] = -r.w; v[1] = -r.z; v[2] = -r.y; v[3] = -r.x;
    } else if (i0 < 192) {
        const float4 r = *reinterpret_cast<const float4 *>(h + i0 - 64);
        v[0] = r.x; v[1] = r.y; v[2] = r.z; v[3] = r.w;
    } else {
        const float4 r = *reinterpret_cast<const float4 *>(h + 316 - i0);
        v[0] = r.w; v[1] = r.z; v[2] = r.y; v[3] = r.x;
    }
}


}  // namespace symaccel
