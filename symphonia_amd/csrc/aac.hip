// AAC-LC synthesis: Dsp::synth (symphonia-codec-aac/src/aac/dsp.rs:57-158) = Imdct (1x1024 or
// 8x128, symphonia-core/src/dsp/mdct.rs:67-146) + window + overlap-add, batched over chains.
//
// MI355X mapping (DESIGN.md "aac_synth"):
//  * one 64-lane wavefront walks a SEGMENT of consecutive frames of one chain; the 1024-sample
//    delay line stays in 16 VGPRs/lane between frames, so HBM traffic is the algorithmic
//    4 KiB in + 4 KiB out per channel-frame (+ one halo frame re-read per segment: segments
//    start by recomputing the previous frame's delay, which depends only on that frame's input);
//  * the 512-point complex FFT is three radix-8 register passes (stages 1-3, 4-6, 7-9 of the
//    reference's radix-2 DIT graph -- identical operands and roundings, only the schedule differs)
//    with conflict-free XOR-swizzled LDS transposes between them (tools/lds_sim.py);
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

constexpr int kAacLds = 2048;  // floats per wavefront: 512 complex for the long path, pcm_long[2048] for short

enum : int { ONLY_LONG = 0, LONG_START = 1, EIGHT_SHORT = 2, LONG_STOP = 3 };
constexpr int kP0 = 512 - 64;  // SHORT_WIN_POINT0 (dsp.rs:19)
constexpr int kP1 = 512 + 64;  // SHORT_WIN_POINT1 (dsp.rs:20)

// LDS complex index of element (B, j, k) = logical position 64B + 8j + k.
// T1: pass-1 lanes (B, j) write k = 0..7, pass-2 lanes (B, k) read j = 0..7.
__device__ __forceinline__ int lds_t1(int B, int j, int k) {
    return B * 64 + (j & 1) * 32 + ((((j >> 1) ^ (B >> 1)) & 1) * 16) + ((((j >> 2) ^ B) & 1) * 8) + (k ^ B);
}
// T2: pass-2 lanes (B, k) write j = 0..7, pass-3 lanes k' = 8j + k read B = 0..7.
__device__ __forceinline__ int lds_t2(int B, int j, int k) { return (B * 64 + j * 8 + k) ^ ((B & 1) << 3); }

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

// mdct.rs:81-88 for one line pair
__device__ __forceinline__ c32 pre_twiddle(float even_line, float mirrored_line, c32 w) {
    const float odd = -mirrored_line;
    return c32{odd * w.im - even_line * w.re, odd * w.re + even_line * w.im};
}

// mdct.rs:104 / 123: val = w * x.conj()
__device__ __forceinline__ c32 post_twiddle(c32 x, c32 w) { return c_mul(w, c32{x.re, -x.im}); }

struct LongConsts {
    c32 tw_pre[8];      // Imdct twiddle tw[lane + 64 s]
    c32 tw_post[2][4];  // per output slot: tw[255-2m2], tw[254-2m2], tw[256+2m2], tw[257+2m2]
    float kbd[2][8];    // long windows at the slot's two float4 positions: [h][0..3] = win[4m2+q],
    float sine[2][8];   //                                                   [h][4..7] = win[1020-4m2+q]
};

__device__ __forceinline__ void load_long_consts(const DevTables &tb, int lane, LongConsts &c) {
#pragma unroll
    for (int s = 0; s < 8; ++s) c.tw_pre[s] = ld_c(tb.aac_tw_long + lane + 64 * s);
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const int m2 = lane + 64 * h;
        c.tw_post[h][0] = ld_c(tb.aac_tw_long + 255 - 2 * m2);
        c.tw_post[h][1] = ld_c(tb.aac_tw_long + 254 - 2 * m2);
        c.tw_post[h][2] = ld_c(tb.aac_tw_long + 256 + 2 * m2);
        c.tw_post[h][3] = ld_c(tb.aac_tw_long + 257 + 2 * m2);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            c.kbd[h][q] = tb.aac_kbd_long[4 * m2 + q];
            c.kbd[h][4 + q] = tb.aac_kbd_long[1020 - 4 * m2 + q];
            c.sine[h][q] = tb.aac_sine_long[4 * m2 + q];
            c.sine[h][4 + q] = tb.aac_sine_long[1020 - 4 * m2 + q];
        }
    }
}

// Raw IMDCT output owned by a lane: for slot h (m2 = lane + 64h)
//   lo[h][q]   = pcm[4*m2 + q]              hi[h][q]   = pcm[1020 - 4*m2 + q]            (first 1024)
//   lo2[h][q]  = pcm[1024 + 4*m2 + q]       hi2[h][q]  = pcm[1024 + 1020 - 4*m2 + q]     (second 1024)
struct LanePcm {
    float lo[2][4], hi[2][4], lo2[2][4], hi2[2][4];
};

// 1024-line IMDCT of one frame by one wavefront.  `line` holds the lane's 8 coalesced float2
// loads: line[s] = (spec[2m + 128 s], spec[2m + 128 s + 1]).
__device__ __forceinline__ void imdct_long_wave(const float2 (&line)[8], int lane, c32 *lds, const LongConsts &lc,
                                                const LaneTables &lt, LanePcm &out) {
    // ---- pre-twiddle z[m + 64 s]; the mirrored (odd) line sits in lane 63-m's load 7-s
    c32 z[8];
    const int mirror = (63 - lane) * 4;
#pragma unroll
    for (int s = 0; s < 8; ++s) {
        const float mirrored = __int_as_float(__builtin_amdgcn_ds_bpermute(mirror, __float_as_int(line[7 - s].y)));
        z[s] = pre_twiddle(line[s].x, mirrored, lc.tw_pre[s]);
    }
    // ---- pass 1: fft8 over z[m + 64*rev3(r)] -> a[8*rev6(m) + r], i.e. element (B, j, k=r)
    bitrev8(z);
    fft8_regs(z);
    {
        const int B = (int)rev_bits((unsigned)lane & 7u, 3), j = (int)rev_bits((unsigned)lane >> 3, 3);
#pragma unroll
        for (int r = 0; r < 8; ++r) lds[lds_t1(B, j, r)] = z[r];
    }
    __syncthreads();
    // ---- pass 2: lane (B, k) gathers j = 0..7
    const int B2 = lane >> 3, k2 = lane & 7;
    c32 u[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) u[j] = lds[lds_t1(B2, j, k2)];
    pass2_regs(u, lt);
    __syncthreads();
#pragma unroll
    for (int j = 0; j < 8; ++j) lds[lds_t2(B2, j, k2)] = u[j];
    __syncthreads();
    // ---- pass 3: lane k' = 8j + k gathers B = 0..7
#pragma unroll
    for (int B = 0; B < 8; ++B) u[B] = lds[lds_t2(B, lane >> 3, lane & 7)];
    pass3_regs(u, lt);
    __syncthreads();
#pragma unroll
    for (int B = 0; B < 8; ++B) lds[64 * B + lane] = u[B];  // natural order Z[64B + k']
    __syncthreads();
    // ---- post-twiddle into the lane's output slots (mdct.rs:94-137)
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const int m2 = lane + 64 * h;
        const c32 vB = post_twiddle(lds[254 - 2 * m2], lc.tw_post[h][1]);
        const c32 vA = post_twiddle(lds[255 - 2 * m2], lc.tw_post[h][0]);
        const c32 vC = post_twiddle(lds[256 + 2 * m2], lc.tw_post[h][2]);
        const c32 vD = post_twiddle(lds[257 + 2 * m2], lc.tw_post[h][3]);
        out.lo[h][0] = -vC.re;  // vec0[4m2 .. 4m2+3]
        out.lo[h][1] = -vA.im;
        out.lo[h][2] = -vD.re;
        out.lo[h][3] = -vB.im;
        out.hi[h][0] = vB.im;   // vec1[508-4m2 .. 511-4m2]
        out.hi[h][1] = vD.re;
        out.hi[h][2] = vA.im;
        out.hi[h][3] = vC.re;
        out.lo2[h][0] = vC.im;  // vec2[4m2 ..]
        out.lo2[h][1] = vA.re;
        out.lo2[h][2] = vD.im;
        out.lo2[h][3] = vB.re;
        out.hi2[h][0] = vB.re;  // vec3[508-4m2 ..]
        out.hi2[h][1] = vD.im;
        out.hi2[h][2] = vA.re;
        out.hi2[h][3] = vC.im;
    }
    __syncthreads();  // LDS is reused by the next frame
}

// Eight 128-line IMDCTs (dsp.rs:80-83) into pcm_long[2048] in LDS (reference layout).
__device__ __forceinline__ void imdct_short_wave(const float2 (&line)[8], int lane, float *ldsf, const DevTables &tb,
                                                 const LaneTables &lt) {
    c32 *lds = reinterpret_cast<c32 *>(ldsf);
    // stage the frame's 1024 lines in LDS (the short transform indexes them per 128-line window)
#pragma unroll
    for (int s = 0; s < 8; ++s) {
        ldsf[2 * lane + 128 * s] = line[s].x;
        ldsf[2 * lane + 128 * s + 1] = line[s].y;
    }
    __syncthreads();
    // pass 1: lane (w, c) owns z_w[c + 8 s] = pre_twiddle(x_w[2i], x_w[127 - 2i]), i = c + 8 s
    const int w = lane >> 3, c = lane & 7;
    c32 z[8];
#pragma unroll
    for (int s = 0; s < 8; ++s) {
        const int i = c + 8 * s;
        z[s] = pre_twiddle(ldsf[128 * w + 2 * i], ldsf[128 * w + 127 - 2 * i], ld_c(tb.aac_tw_short + i));
    }
    __syncthreads();
    bitrev8(z);
    fft8_regs(z);  // -> a_w[8*rev3(c) + r] = element (B = w, j = rev3(c), k = r)
    {
        const int j = (int)rev_bits((unsigned)c, 3);
#pragma unroll
        for (int r = 0; r < 8; ++r) lds[lds_t1(w, j, r)] = z[r];
    }
    __syncthreads();
    c32 u[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) u[j] = lds[lds_t1(w, j, c)];
    pass2_regs(u, lt);  // 64-point FFT done: u[j] = Z_w[8j + k], k = c
    __syncthreads();
    // post-twiddle (mdct.rs:94-137 with n2 = 64, n4 = 32) into pcm_long[256 w ..]
    float *o = ldsf + 256 * w;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int i = 8 * j + c;
        const c32 val = post_twiddle(u[j], ld_c(tb.aac_tw_short + i));
        if (j < 4) {
            const int fi = 2 * i, ri = 63 - 2 * i;
            o[ri] = -val.im;
            o[64 + fi] = val.im;
            o[128 + ri] = val.re;
            o[192 + fi] = val.re;
        } else {
            const int i2 = i - 32;
            const int fi = 2 * i2, ri = 63 - 2 * i2;
            o[fi] = -val.re;
            o[64 + ri] = val.re;
            o[128 + fi] = val.im;
            o[192 + ri] = val.im;
        }
    }
    __syncthreads();
}

// pcm_short[q] of dsp.rs:86-101, rebuilt from pcm_long in LDS with the reference's operation order
// (including the `0.0 +` of the `+=` onto the zero-filled buffer for windows > 0).
__device__ __forceinline__ float pcm_short_at(const float *pcm_long, int q, const float *short_win,
                                              const float *prev_short_win) {
    const int w = q >> 7, i = q & 127;
    float acc = 0.0f;
    bool have = false;
    if (w >= 1) {  // right half of window w-1: src[i + 128] * short_win[127 - i]
        const float a = pcm_long[256 * (w - 1) + 128 + i] * short_win[127 - i];
        acc = (w - 1 == 0) ? a : (0.0f + a);
        have = true;
    }
    if (w <= 7) {  // left half of window w
        const float b = pcm_long[256 * w + i] * (w == 0 ? prev_short_win[i] : short_win[i]);
        acc = have ? (acc + b) : b;
    }
    return acc;
}

__global__ __launch_bounds__(64) void aac_synth_kernel(DevTables tb, const float *__restrict__ coeffs,
                                                       const uint8_t *__restrict__ side,
                                                       const float *__restrict__ delay_in,
                                                       float *__restrict__ delay_out, float *__restrict__ pcm,
                                                       unsigned frames_per_chain, unsigned seg_len,
                                                       unsigned segs_per_chain) {
    __shared__ __attribute__((aligned(16))) float ldsf[kAacLds];
    c32 *lds = reinterpret_cast<c32 *>(ldsf);
    const int lane = (int)threadIdx.x;
    const unsigned chain = blockIdx.x / segs_per_chain, seg = blockIdx.x % segs_per_chain;
    const unsigned t_begin = seg * seg_len;
    const unsigned t_end = min(t_begin + seg_len, frames_per_chain);
    const size_t chain_base = (size_t)chain * frames_per_chain;

    LaneTables lt;
    LongConsts lc;
    load_lane_tables(tb, lane, lt);
    load_long_consts(tb, lane, lc);

    // delay line, in slot layout: dl[h][0..3] = delay[4m2 + q], dl[h][4..7] = delay[1020 - 4m2 + q]
    float dl[2][8];
    if (t_begin == 0) {
        const float4 *d4 = reinterpret_cast<const float4 *>(delay_in + (size_t)chain * 1024);
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int m2 = lane + 64 * h;
            const float4 a = d4[m2], b = d4[255 - m2];
            dl[h][0] = a.x; dl[h][1] = a.y; dl[h][2] = a.z; dl[h][3] = a.w;
            dl[h][4] = b.x; dl[h][5] = b.y; dl[h][6] = b.z; dl[h][7] = b.w;
        }
    } else {
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int q = 0; q < 8; ++q) dl[h][q] = 0.0f;
    }

    // frame t_begin-1 is the halo: it only rebuilds the delay line
    const long t_first = t_begin == 0 ? 0 : (long)t_begin - 1;
    float2 line[8];
    {
        const float2 *src = reinterpret_cast<const float2 *>(coeffs + (chain_base + (size_t)t_first) * 1024);
#pragma unroll
        for (int s = 0; s < 8; ++s) line[s] = src[lane + 64 * s];
    }

    for (long t = t_first; t < (long)t_end; ++t) {
        const bool emit = t >= (long)t_begin;
        const unsigned sb = side[chain_base + (size_t)t];
        const int seq = (int)(sb & 3u);
        const bool shape = (sb >> 2) & 1u, prev_shape = (sb >> 3) & 1u;

        float2 cur[8];
#pragma unroll
        for (int s = 0; s < 8; ++s) cur[s] = line[s];
        if (t + 1 < (long)t_end) {  // prefetch the next frame while this one is transformed
            const float2 *src = reinterpret_cast<const float2 *>(coeffs + (chain_base + (size_t)t + 1) * 1024);
#pragma unroll
            for (int s = 0; s < 8; ++s) line[s] = src[lane + 64 * s];
        }

        float dst[2][8];
        if (seq != EIGHT_SHORT) {
            LanePcm p;
            imdct_long_wave(cur, lane, lds, lc, lt, p);
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int m2 = lane + 64 * h;
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    const int j = q < 4 ? 4 * m2 + q : 1020 - 4 * m2 + (q - 4);
                    const float x = q < 4 ? p.lo[h][q] : p.hi[h][q - 4];       // pcm_long[j]
                    const float x2 = q < 4 ? p.lo2[h][q] : p.hi2[h][q - 4];    // pcm_long[1024 + j]
                    // window values at j and at 1023 - j (the slot's other float4, reversed)
                    const int qr = q < 4 ? 7 - q : 3 - (q - 4);
                    const float wprev = prev_shape ? lc.kbd[h][q] : lc.sine[h][q];
                    const float wcur_rev = shape ? lc.kbd[h][qr] : lc.sine[h][qr];
                    // ---- output samples (dsp.rs:105-129)
                    float d;
                    if (seq == LONG_STOP) {
                        if (j < kP0) {
                            d = dl[h][q];
                        } else if (j < kP1) {
                            const float *psw = prev_shape ? tb.aac_kbd_short : tb.aac_sine_short;
                            d = dl[h][q] + x * psw[j - kP0];
                        } else {
                            d = dl[h][q] + x;
                        }
                    } else {
                        d = dl[h][q] + (x * wprev);
                    }
                    dst[h][q] = d;
                    // ---- delay for the next frame (dsp.rs:132-157)
                    float nd;
                    if (seq == LONG_START) {
                        if (j < kP0) {
                            nd = x2;
                        } else if (j < kP1) {
                            const float *sw = shape ? tb.aac_kbd_short : tb.aac_sine_short;
                            nd = x2 * sw[127 - (j - kP0)];
                        } else {
                            nd = 0.0f;
                        }
                    } else {
                        nd = x2 * wcur_rev;
                    }
                    dl[h][q] = nd;
                }
            }
        } else {
            imdct_short_wave(cur, lane, ldsf, tb, lt);
            const float *sw = shape ? tb.aac_kbd_short : tb.aac_sine_short;
            const float *psw = prev_shape ? tb.aac_kbd_short : tb.aac_sine_short;
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int m2 = lane + 64 * h;
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    const int j = q < 4 ? 4 * m2 + q : 1020 - 4 * m2 + (q - 4);
                    // dsp.rs:111-117
                    dst[h][q] = j < kP0 ? dl[h][q] : dl[h][q] + pcm_short_at(ldsf, j - kP0, sw, psw);
                    // dsp.rs:138-145
                    dl[h][q] = j < kP1 ? pcm_short_at(ldsf, j + kP1, sw, psw) : 0.0f;
                }
            }
            __syncthreads();  // pcm_long in LDS is overwritten by the next frame
        }

        if (emit) {
            float4 *o4 = reinterpret_cast<float4 *>(pcm + (chain_base + (size_t)t) * 1024);
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int m2 = lane + 64 * h;
                o4[m2] = make_float4(dst[h][0], dst[h][1], dst[h][2], dst[h][3]);
                o4[255 - m2] = make_float4(dst[h][4], dst[h][5], dst[h][6], dst[h][7]);
            }
        }
    }

    if (t_end == frames_per_chain) {
        float4 *d4 = reinterpret_cast<float4 *>(delay_out + (size_t)chain * 1024);
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int m2 = lane + 64 * h;
            d4[m2] = make_float4(dl[h][0], dl[h][1], dl[h][2], dl[h][3]);
            d4[255 - m2] = make_float4(dl[h][4], dl[h][5], dl[h][6], dl[h][7]);
        }
    }
}

}  // namespace

int launch_aac(symaccel_ctx *ctx, const float *d_coeffs, const uint8_t *d_side, const float *d_delay_in,
               float *d_delay_out, float *d_pcm, size_t n_chains, size_t frames_per_chain) {
    if (frames_per_chain > 0x7fffffffu) return SYMACCEL_ERR_INVALID_ARG;
    unsigned seg = ctx->segment > 0 ? (unsigned)ctx->segment : 32u;
    if (seg > frames_per_chain) seg = (unsigned)frames_per_chain;
    const size_t segs = (frames_per_chain + seg - 1) / seg;
    const size_t grid = n_chains * segs;
    if (grid > 0x7fffffffu) return SYMACCEL_ERR_INVALID_ARG;
    hipLaunchKernelGGL(aac_synth_kernel, dim3((unsigned)grid), dim3(64), 0, ctx->stream, ctx->dev, d_coeffs, d_side,
                       d_delay_in, d_delay_out, d_pcm, (unsigned)frames_per_chain, seg, (unsigned)segs);
    SYM_GPU(ctx, hipGetLastError());
    return SYMACCEL_OK;
}

}  // namespace symaccel
