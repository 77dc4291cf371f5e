// AAC-LC synthesis: Dsp::synth (symphonia-codec-aac/src/aac/dsp.rs:57-158) = Imdct (1x1024 or
// 8x128, symphonia-core/src/dsp/mdct.rs:67-146) + window + overlap-add, batched over chains.
//
// MI355X mapping (DESIGN.md "aac_synth"):
//  * one 64-lane wavefront walks a SEGMENT of consecutive frames of one chain; the 1024-sample
//    delay line stays in 16 VGPRs/lane between frames, so HBM traffic is the algorithmic
//    4 KiB in + 4 KiB out per channel-frame (+ one halo frame re-read per segment: segments
//    start by recomputing the previous frame's delay, which depends only on that frame's input);
//  * a workgroup is four such wavefronts that share only the LDS-resident tables (Imdct twiddles,
//    KBD and sine long windows, 12 KiB); they never synchronise with each other after start-up --
//    each wavefront orders its own LDS traffic with wave-local fences;
//  * the 512-point complex FFT is three radix-8 register passes (stages 1-3, 4-6, 7-9 of the
//    reference's radix-2 DIT graph -- identical operands and roundings, only the schedule differs)
//    with conflict-free LDS transposes between them whose addresses are lane base + immediate;
//  * spectrum loads are 8 B/lane coalesced; the bit-reversal is absorbed in the lane mapping
//    (lane m owns z[m + 64 s]) so the input never round-trips through LDS; the mirrored odd
//    lines come from lane 63-m via ds_bpermute;
//  * the post-twiddle emits each lane's outputs as four contiguous float4 per frame
//    (slots m2 and 255-m2), the same positions for dst and for the next delay line, so PCM
//    stores are 1 KiB-coalesced and the overlap-add is lane-local.
// Roofline: HBM-bound, 8192 B algorithmic per channel-frame, ~27 kflop (no FMA) -> 3.3 flop/B.
#include "dsp_device.h"

namespace symaccel {

namespace {

constexpr int kWaves = 4;                 // wavefronts per workgroup
constexpr int kWaveLds = 2264;            // floats of private LDS per wavefront (FFT work array, see below)
constexpr int kTabTw = 0;                 // shared LDS tables: Imdct twiddles, 512 complex
constexpr int kTabKbd = 1024;             //   KBD long window, 1024 f32
constexpr int kTabSine = 2048;            //   sine long window, 1024 f32
constexpr int kTabFloats = 3072;

enum : int { ONLY_LONG = 0, LONG_START = 1, EIGHT_SHORT = 2, LONG_STOP = 3 };
constexpr int kP0 = 512 - 64;  // SHORT_WIN_POINT0 (dsp.rs:19)
constexpr int kP1 = 512 + 64;  // SHORT_WIN_POINT1 (dsp.rs:20)

// Order this wavefront's LDS accesses (its lanes exchange data through LDS; the hardware executes
// one wavefront's DS instructions in order, the fences stop the compiler from reordering them).
__device__ __forceinline__ void wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
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
};

__device__ __forceinline__ c32 ld_c(const cpx *p) {
    const cpx v = *p;
    return c32{v.re, v.im};
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

// Stages 4-6 of the radix-2 graph on the eight values u[j] = a[64B + 8j + k] of one lane.
__device__ __forceinline__ void pass2_regs(c32 (&u)[8], const LaneTables &t) {
#pragma unroll
    for (int j = 0; j < 8; j += 2) bfly(u[j], u[j + 1], tw_small(u[j + 1], t.w16, t.f16));  // fft16 combine
#pragma unroll
    for (int h = 0; h < 8; h += 4) {                                                          // fft32 combine
        bfly(u[h + 0], u[h + 2], tw_small(u[h + 2], t.w32[0], t.f32[0]));
        bfly(u[h + 1], u[h + 3], tw_small(u[h + 3], t.w32[1], t.f32[1]));
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) bfly(u[j], u[j + 4], c_mul(u[j + 4], t.w64[j]));             // merge, step 32
}

// Stages 7-9 on v[B] = a[64B + k'].
__device__ __forceinline__ void pass3_regs(c32 (&v)[8], const LaneTables &t) {
#pragma unroll
    for (int B = 0; B < 8; B += 2) bfly(v[B], v[B + 1], c_mul(v[B + 1], t.w128));            // step 64
#pragma unroll
    for (int h = 0; h < 8; h += 4) {                                                          // step 128
        bfly(v[h + 0], v[h + 2], c_mul(v[h + 2], t.w256[0]));
        bfly(v[h + 1], v[h + 3], c_mul(v[h + 3], t.w256[1]));
    }
#pragma unroll
    for (int B = 0; B < 4; ++B) bfly(v[B], v[B + 4], c_mul(v[B + 4], t.w512[B]));            // step 256
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
__device__ __forceinline__ void fft512_wave(c32 (&z)[8], int lane, c32 *lds, const LaneTables &lt) {
    // ---- pass 1: fft8 over z[m + 64*rev3(r)] -> a[8*rev6(m) + r], i.e. element (B, j, k=r)
    bitrev8(z);
    fft8_regs(z);
    {
        const int B = (int)rev_bits((unsigned)lane & 7u, 3), j = (int)rev_bits((unsigned)lane >> 3, 3);
        c32 *w = lds + lds_t1_lane_w(B, j);
#pragma unroll
        for (int r = 0; r < 8; ++r) w[lds_t1_inst_w(r)] = z[r];
    }
    wave_sync();
    // ---- pass 2: lane (B, k) gathers j = 0..7
    const int B2 = lane >> 3, k2 = lane & 7;
    {
        const c32 *r = lds + lds_t1_lane_r(B2, k2);
#pragma unroll
        for (int j = 0; j < 8; ++j) z[j] = r[lds_t1_inst_r(j)];
    }
    pass2_regs(z, lt);
    wave_sync();
    {
        c32 *w = lds + lds_t2_lane_w(B2, k2);
#pragma unroll
        for (int j = 0; j < 8; ++j) w[lds_t2_inst_w(j)] = z[j];
    }
    wave_sync();
    // ---- pass 3: lane k' = 8j + k gathers B = 0..7
    {
        const c32 *r = lds + lds_t2_lane_r(lane >> 3, lane & 7);
#pragma unroll
        for (int B = 0; B < 8; ++B) z[B] = r[lds_t2_inst_r(B)];
    }
    pass3_regs(z, lt);
    wave_sync();
#pragma unroll
    for (int B = 0; B < 8; ++B) lds[64 * B + lane] = z[B];  // natural order Z[64B + k']
    wave_sync();
}

// Post-twiddle (mdct.rs:94-137) of the four FFT bins that feed output slot m2 = lane + 64h:
//   x[q]  = pcm[j(q)]         (first half of the 2048-sample IMDCT output)
//   x2[q] = pcm[1024 + j(q)]  with j(q) = 4*m2 + q for q < 4, 1020 - 4*m2 + (q - 4) for q >= 4.
// Twiddles tw[254-2m2 .. 255-2m2] and tw[256+2m2 .. 257+2m2] come from the shared LDS table.
__device__ __forceinline__ void post_slot(const c32 *lds, const c32 *tw, int m2, float (&x)[8], float (&x2)[8]) {
    const c32 vB = post_twiddle(lds[254 - 2 * m2], tw[254 - 2 * m2]);
    const c32 vA = post_twiddle(lds[255 - 2 * m2], tw[255 - 2 * m2]);
    const c32 vC = post_twiddle(lds[256 + 2 * m2], tw[256 + 2 * m2]);
    const c32 vD = post_twiddle(lds[257 + 2 * m2], tw[257 + 2 * m2]);
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

// Eight 128-line IMDCTs (dsp.rs:80-83).  The frame's 1024 lines are staged in ldsf[0..1024).  Of each
// window's 256 outputs v0|v1|v2|v3 only v1 and v2 are kept: H[w][0..64) = v1, H[w][64..128) = v2 in
// ldsf[128 w ..]; v0[x] = -v1[63-x] and v3[x] = v2[63-x] exactly (mdct.rs:108-136 writes the same
// value, negated for v0, to both).
__device__ __forceinline__ void imdct_short_wave(int lane, float *ldsf, const DevTables &tb, const LaneTables &lt) {
    c32 *lds = reinterpret_cast<c32 *>(ldsf);
    // pass 1: lane (w, c) owns z_w[c + 8 s] = pre_twiddle(x_w[2i], x_w[127 - 2i]), i = c + 8 s
    const int w = lane >> 3, c = lane & 7;
    c32 z[8];
#pragma unroll
    for (int s = 0; s < 8; ++s) {
        const int i = c + 8 * s;
        z[s] = pre_twiddle(ldsf[128 * w + 2 * i], ldsf[128 * w + 127 - 2 * i], ld_c(tb.aac_tw_short + i));
    }
    wave_sync();
    bitrev8(z);
    fft8_regs(z);  // -> a_w[8*rev3(c) + r] = element (B = w, j = rev3(c), k = r)
    {
        const int j = (int)rev_bits((unsigned)c, 3);
        c32 *wp = lds + lds_t1_lane_w(w, j);
#pragma unroll
        for (int r = 0; r < 8; ++r) wp[lds_t1_inst_w(r)] = z[r];
    }
    wave_sync();
    {
        const c32 *rp = lds + lds_t1_lane_r(w, c);
#pragma unroll
        for (int j = 0; j < 8; ++j) z[j] = rp[lds_t1_inst_r(j)];
    }
    pass2_regs(z, lt);  // 64-point FFT done: z[j] = Z_w[8j + k], k = c
    wave_sync();
    // post-twiddle (mdct.rs:94-137 with n2 = 64, n4 = 32)
    float *o = ldsf + 128 * w;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int i = 8 * j + c;
        const c32 val = post_twiddle(z[j], ld_c(tb.aac_tw_short + i));
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

// src_w[i] of dsp.rs:86-101 (i in 0..256) from the half-stored windows.
__device__ __forceinline__ float short_src(const float *H, int w, int i) {
    const float *h = H + 128 * w;
    if (i < 64) return -h[63 - i];
    if (i < 192) return h[i - 64];
    return h[319 - i];
}

// pcm_short[q] of dsp.rs:86-101 with the reference's operation order (including the `0.0 +` of the
// `+=` onto the zero-filled buffer for windows > 0).
__device__ __forceinline__ float pcm_short_at(const float *H, int q, const float *short_win,
                                              const float *prev_short_win) {
    const int w = q >> 7, i = q & 127;
    float acc = 0.0f;
    bool have = false;
    if (w >= 1) {  // right half of window w-1: src[i + 128] * short_win[127 - i]
        const float a = short_src(H, w - 1, 128 + i) * short_win[127 - i];
        acc = (w - 1 == 0) ? a : (0.0f + a);
        have = true;
    }
    if (w <= 7) {  // left half of window w
        const float b = short_src(H, w, i) * (w == 0 ? prev_short_win[i] : short_win[i]);
        acc = have ? (acc + b) : b;
    }
    return acc;
}

__device__ __forceinline__ void store_slot(float *frame, int m2, const float (&v)[8]) {
    float4 *o4 = reinterpret_cast<float4 *>(frame);
    o4[m2] = make_float4(v[0], v[1], v[2], v[3]);
    o4[255 - m2] = make_float4(v[4], v[5], v[6], v[7]);
}
__device__ __forceinline__ void load_slot(const float *frame, int m2, float (&v)[8]) {
    const float4 *i4 = reinterpret_cast<const float4 *>(frame);
    const float4 a = i4[m2], b = i4[255 - m2];
    v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w;
    v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
}

// Effective output window of a LONG_STOP frame at sample j (dsp.rs:118-127): j < 448 -> dst = delay
// (flagged by the caller), 448..575 -> prev_short_win[j-448], >= 576 -> delay + pcm (pcm * 1.0 is exact).
__device__ __forceinline__ float stop_window(const DevTables &tb, int prev_shape, int j) {
    const float *psw = prev_shape ? tb.aac_kbd_short : tb.aac_sine_short;
    return (j >= kP0 && j < kP1) ? psw[j - kP0] : 1.0f;
}
// Effective delay window of a LONG_START frame at sample j (dsp.rs:146-155), multiplying pcm[1024 + j]:
// j < 448 -> copy (x * 1.0 exact), 448..575 -> short_win[127 - (j-448)], >= 576 -> literal 0.0 (caller).
__device__ __forceinline__ float start_window(const DevTables &tb, int shape, int j) {
    const float *sw = shape ? tb.aac_kbd_short : tb.aac_sine_short;
    return (j >= kP0 && j < kP1) ? sw[127 - (j - kP0)] : 1.0f;
}

#ifndef SYM_AAC_MIN_WAVES
#define SYM_AAC_MIN_WAVES 2  // wavefronts per SIMD the register allocation must allow (build-time tuning knob)
#endif
__global__ __launch_bounds__(64 * kWaves, SYM_AAC_MIN_WAVES) void aac_synth_kernel(
    DevTables tb, const float *__restrict__ coeffs, const uint8_t *__restrict__ side,
    const float *__restrict__ delay_in, float *__restrict__ delay_out, float *__restrict__ pcm,
    unsigned frames_per_chain, unsigned seg_len, unsigned segs_per_chain, unsigned n_items) {
    __shared__ __attribute__((aligned(16))) float tabs[kTabFloats];
    __shared__ __attribute__((aligned(16))) float wave_lds[kWaves][kWaveLds];

    // ---- shared tables -> LDS (once per workgroup)
    for (int i = (int)threadIdx.x; i < 1024; i += 64 * kWaves) {
        tabs[kTabTw + i] = reinterpret_cast<const float *>(tb.aac_tw_long)[i];
        tabs[kTabKbd + i] = tb.aac_kbd_long[i];
        tabs[kTabSine + i] = tb.aac_sine_long[i];
    }
    __syncthreads();  // the only workgroup-wide barrier; wavefronts are independent from here on

    const int lane = (int)threadIdx.x & 63, wave = (int)threadIdx.x >> 6;
    const unsigned item = blockIdx.x * kWaves + (unsigned)wave;
    if (item >= n_items) return;
    float *ldsf = wave_lds[wave];
    c32 *lds = reinterpret_cast<c32 *>(ldsf);
    const c32 *tw = reinterpret_cast<const c32 *>(tabs + kTabTw);

    const unsigned chain = item / segs_per_chain, seg = item % segs_per_chain;
    const unsigned t_begin = seg * seg_len;
    const unsigned t_end = min(t_begin + seg_len, frames_per_chain);
    const size_t chain_base = (size_t)chain * frames_per_chain;

    LaneTables lt;
    load_lane_tables(tb, lane, lt);

    // delay line, in slot layout: dl[h][0..3] = delay[4m2 + q], dl[h][4..7] = delay[1020 - 4m2 + q]
    float dl[2][8];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        if (t_begin == 0) {
            load_slot(delay_in + (size_t)chain * 1024, lane + 64 * h, dl[h]);
        } else {
#pragma unroll
            for (int q = 0; q < 8; ++q) dl[h][q] = 0.0f;
        }
    }

    // frame t_begin-1 is the halo: it only rebuilds the delay line
    const long t_first = t_begin == 0 ? 0 : (long)t_begin - 1;
    float2 line[8];  // line[s] = (spec[2m + 128 s], spec[2m + 128 s + 1]), 512 B coalesced per load
    {
        const float2 *src = reinterpret_cast<const float2 *>(coeffs + (chain_base + (size_t)t_first) * 1024);
#pragma unroll
        for (int s = 0; s < 8; ++s) line[s] = src[lane + 64 * s];
    }

    for (long t = t_first; t < (long)t_end; ++t) {
        const bool emit = t >= (long)t_begin;
        const unsigned sb = side[chain_base + (size_t)t];
        const int seq = (int)(sb & 3u);
        const int shape = (int)((sb >> 2) & 1u), prev_shape = (int)((sb >> 3) & 1u);
        float *frame_out = pcm + (chain_base + (size_t)t) * 1024;

        // ---- consume the prefetched lines
        c32 z[8];
        if (seq != EIGHT_SHORT) {
            // pre-twiddle z[m + 64 s] with tw[m + 64 s]; the mirrored (odd) line sits in lane 63-m's load 7-s
            const int mirror = (63 - lane) * 4;
#pragma unroll
            for (int s = 0; s < 8; ++s) {
                const float mirrored =
                    __int_as_float(__builtin_amdgcn_ds_bpermute(mirror, __float_as_int(line[7 - s].y)));
                z[s] = pre_twiddle(line[s].x, mirrored, tw[lane + 64 * s]);
            }
        } else {
#pragma unroll
            for (int s = 0; s < 8; ++s) {  // the short transform indexes lines per 128-line window: stage in LDS
                ldsf[2 * lane + 128 * s] = line[s].x;
                ldsf[2 * lane + 128 * s + 1] = line[s].y;
            }
            wave_sync();
        }
        if (t + 1 < (long)t_end) {  // prefetch the next frame; it lands while this one is transformed
            const float2 *src = reinterpret_cast<const float2 *>(coeffs + (chain_base + (size_t)t + 1) * 1024);
#pragma unroll
            for (int s = 0; s < 8; ++s) line[s] = src[lane + 64 * s];
        }

        if (seq != EIGHT_SHORT) {
            fft512_wave(z, lane, lds, lt);
            const float *wprev = tabs + (prev_shape ? kTabKbd : kTabSine);  // prev_long_win (dsp.rs:71-74)
            const float *wcur = tabs + (shape ? kTabKbd : kTabSine);        // long_win (dsp.rs:66-69)
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int m2 = lane + 64 * h;
                float x[8], x2[8];
                post_slot(lds, tw, m2, x, x2);
                // ---- output samples (dsp.rs:105-129): dst = delay + pcm * w, or delay alone
                float wo[8], dst[8];
                if (seq == LONG_STOP) {
#pragma unroll
                    for (int q = 0; q < 8; ++q)
                        wo[q] = stop_window(tb, prev_shape, q < 4 ? 4 * m2 + q : 1020 - 4 * m2 + (q - 4));
                } else {
                    load_slot(wprev, m2, wo);
                }
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    const int j = q < 4 ? 4 * m2 + q : 1020 - 4 * m2 + (q - 4);
                    const float v = dl[h][q] + (x[q] * wo[q]);
                    dst[q] = (seq == LONG_STOP && j < kP0) ? dl[h][q] : v;
                }
                if (emit) store_slot(frame_out, m2, dst);
                // ---- delay for the next frame (dsp.rs:132-157): pcm[1024 + j] * long_win[1023 - j] (the
                // slot's two float4 read backwards), a short-window slope, or literal zero
                float wd[8];
                if (seq == LONG_START) {
#pragma unroll
                    for (int q = 0; q < 8; ++q)
                        wd[q] = start_window(tb, shape, q < 4 ? 4 * m2 + q : 1020 - 4 * m2 + (q - 4));
                } else {
                    float wr[8];
                    load_slot(wcur, m2, wr);
#pragma unroll
                    for (int q = 0; q < 8; ++q) wd[q] = wr[7 - q];
                }
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    const int j = q < 4 ? 4 * m2 + q : 1020 - 4 * m2 + (q - 4);
                    const float v = x2[q] * wd[q];
                    dl[h][q] = (seq == LONG_START && j >= kP1) ? 0.0f : v;
                }
            }
            wave_sync();  // Z in LDS is overwritten by the next frame
        } else {
            // ---- eight short windows (rare): everything through LDS in natural order
            float *dly = ldsf + 1024;
            imdct_short_wave(lane, ldsf, tb, lt);  // H[8][128] in ldsf[0..1024)
#pragma unroll
            for (int h = 0; h < 2; ++h) store_slot(dly, lane + 64 * h, dl[h]);  // (the FFT work array overlapped dly)
            wave_sync();
            const float *sw = shape ? tb.aac_kbd_short : tb.aac_sine_short;
            const float *psw = prev_shape ? tb.aac_kbd_short : tb.aac_sine_short;
#pragma unroll 1
            for (int e = 0; e < 16; ++e) {
                const int j = lane + 64 * e;
                const float d = dly[j];
                const float o = j < kP0 ? d : d + pcm_short_at(ldsf, j - kP0, sw, psw);      // dsp.rs:111-117
                if (emit) frame_out[j] = o;
                dly[j] = j < kP1 ? pcm_short_at(ldsf, j + kP1, sw, psw) : 0.0f;               // dsp.rs:138-145
            }
            wave_sync();
#pragma unroll
            for (int h = 0; h < 2; ++h) load_slot(dly, lane + 64 * h, dl[h]);
            wave_sync();  // LDS is overwritten by the next frame
        }
    }

    if (t_end == frames_per_chain) {
        float *d = delay_out + (size_t)chain * 1024;
#pragma unroll
        for (int h = 0; h < 2; ++h) store_slot(d, lane + 64 * h, dl[h]);
    }
}

}  // namespace

int launch_aac(symaccel_ctx *ctx, const float *d_coeffs, const uint8_t *d_side, const float *d_delay_in,
               float *d_delay_out, float *d_pcm, size_t n_chains, size_t frames_per_chain) {
    if (frames_per_chain > 0x7fffffffu) return SYMACCEL_ERR_INVALID_ARG;
    unsigned seg = ctx->segment > 0 ? (unsigned)ctx->segment : 32u;
    if (seg > frames_per_chain) seg = (unsigned)frames_per_chain;
    const size_t segs = (frames_per_chain + seg - 1) / seg;
    const size_t items = n_chains * segs;
    const size_t grid = (items + kWaves - 1) / kWaves;
    if (items > 0xffffffffu || grid > 0x7fffffffu) return SYMACCEL_ERR_INVALID_ARG;
    hipLaunchKernelGGL(aac_synth_kernel, dim3((unsigned)grid), dim3(64 * kWaves), 0, ctx->stream, ctx->dev, d_coeffs,
                       d_side, d_delay_in, d_delay_out, d_pcm, (unsigned)frames_per_chain, seg, (unsigned)segs,
                       (unsigned)items);
    SYM_GPU(ctx, hipGetLastError());
    return SYMACCEL_OK;
}

}  // namespace symaccel
