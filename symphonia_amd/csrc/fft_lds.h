// Workgroup-cooperative radix-2 DIT FFT over LDS, any power-of-two size 2..4096, reproducing the
// operation DAG of symphonia-core/src/dsp/fft/no_simd.rs: bit-reversed input, unrolled
// fft2..fft32 base cases with their strength-reduced twiddles (no_simd.rs:289-454), then the
// breadth-first merge passes of `transform` (no_simd.rs:221-281).
//
// Layout: a workgroup holds `points` complex values = (points / nf) transforms of size nf back to
// back, already bit-reverse permuted within each transform.  Logical position p lives at
// lds[pad(p)], pad(p) = p + (p >> 5): one complex of padding per 32-chunk, so a thread that owns a
// whole 32-chunk (stride 66 dwords between lanes) reads it conflict-free with ds_read_b64.
#pragma once

#include "dsp_device.h"

namespace symaccel {

__device__ __forceinline__ int fft_pad(int p) { return p + (p >> 5); }
constexpr int fft_padded(int points) { return points + (points >> 5) + 1; }

// Twiddle of element k of an N-point combine step (N = 16 or 32 table driven; N <= 8 static).
template <int N>
__device__ __forceinline__ c32 small_twiddle(c32 v, int k, const cpx *tab) {
    if (k == 0) return v;
    if (4 * k == N) return tw_minus_i(v);
    if (8 * k == N) return tw_n8(v);
    if (8 * k == 3 * N) return tw_3n8(v);
    const cpx w = tab[k];  // literal (cos, -sin) of no_simd.rs:307-324 / 374-383
    return c_mul(c32{w.re, w.im}, v);
}

// fftN on registers, N in {2,4,8,16,32}; x in bit-reversed order (no_simd.rs:289-454).
template <int N>
__device__ __forceinline__ void fft_small_regs(c32 *x, const cpx *small16, const cpx *small32) {
    if constexpr (N == 2) {
        bfly(x[0], x[1], x[1]);
    } else {
        constexpr int H = N / 2;
        fft_small_regs<H>(x, small16, small32);
        fft_small_regs<H>(x + H, small16, small32);
#pragma unroll
        for (int k = 0; k < H; ++k) {
            const c32 q = small_twiddle<N>(x[H + k], k, N == 16 ? small16 : small32);
            bfly(x[k], x[H + k], q);
        }
    }
}

template <int C>
__device__ __forceinline__ void fft_chunk_phase(c32 *lds, int points, const DevTables &tb) {
    for (int q = (int)threadIdx.x; q < points / C; q += (int)blockDim.x) {
        c32 x[C];
        const int base = fft_pad(q * C);  // a chunk never straddles a pad slot (C <= 32)
#pragma unroll
        for (int i = 0; i < C; ++i) x[i] = lds[base + i];
        fft_small_regs<C>(x, tb.small16, tb.small32);
#pragma unroll
        for (int i = 0; i < C; ++i) lds[base + i] = x[i];
    }
}

// In-place FFTs of every transform in the workgroup's LDS tile.  Ends with a barrier.
// `small_cases`: Fft::fft dispatches fft2 .. fft16 itself before transform(); Ifft calls transform() directly, which has no
// case below 32 points (no_simd.rs:221-281: its 64-point chunk loop is empty) -- such an Ifft runs no butterflies at all.
__device__ __forceinline__ void wg_fft_lds(c32 *lds, int nf, int points, const DevTables &tb, bool small_cases = true) {
    __syncthreads();
    if (nf < 32 && !small_cases) return;
    switch (nf) {
        case 2: fft_chunk_phase<2>(lds, points, tb); break;
        case 4: fft_chunk_phase<4>(lds, points, tb); break;
        case 8: fft_chunk_phase<8>(lds, points, tb); break;
        case 16: fft_chunk_phase<16>(lds, points, tb); break;
        default:
            // 32 points and more: fft8 on every 8-group with one thread per group, then the fft16 and fft32 combine steps
            // of every 32-chunk as one pass with eight threads per chunk (k, k + 8, k + 16, k + 24 each) -- the same
            // butterflies as fft_small_regs<32>, which kept 32 values per thread and only points / 32 threads busy.
            fft_chunk_phase<8>(lds, points, tb);
            __syncthreads();
#pragma unroll 2
            for (int g = (int)threadIdx.x; g < points / 4; g += (int)blockDim.x) {
                const int k = g & 7, base = ((g >> 3) << 5) + k;
                const int i0 = fft_pad(base), i1 = fft_pad(base + 8), i2 = fft_pad(base + 16), i3 = fft_pad(base + 24);
                const cpx wa = tb.small16[k], wb = tb.small32[k], wc = tb.small32[k + 8];
                const int fa = tb.small16_form[k], fb = tb.small32_form[k], fc = tb.small32_form[k + 8];
                c32 a0 = lds[i0], a1 = lds[i1], a2 = lds[i2], a3 = lds[i3];
                bfly(a0, a1, tw_small(a1, c32{wa.re, wa.im}, fa));  // fft16 combine (no_simd.rs:307-345), both halves of the chunk
                bfly(a2, a3, tw_small(a3, c32{wa.re, wa.im}, fa));
                bfly(a0, a2, tw_small(a2, c32{wb.re, wb.im}, fb));  // fft32 combine (no_simd.rs:374-405)
                bfly(a1, a3, tw_small(a3, c32{wc.re, wc.im}, fc));
                lds[i0] = a0;
                lds[i1] = a1;
                lds[i2] = a2;
                lds[i3] = a3;
            }
            break;
    }
    // merge passes, step = 32, 64, ..., nf/2 (no_simd.rs:247-279).  Two consecutive stages are done per pass through LDS
    // where two remain: a thread takes the four elements k, k + step, k + 2 step, k + 3 step of a 4 * step block, runs the
    // two `step` butterflies and then the two `2 * step` butterflies on them in registers -- the reference's operations on
    // the reference's operands, half the LDS round trips and barriers.
    int step = 32;
    for (; 2 * step < nf; step <<= 2) {
        __syncthreads();
        const cpx *w1 = tb.fft_merge + (step - 32);      // W_{2 * step}
        const cpx *w2 = tb.fft_merge + (2 * step - 32);  // W_{4 * step}
#pragma unroll 2
        for (int g = (int)threadIdx.x; g < points / 4; g += (int)blockDim.x) {
            const int k = g & (step - 1);
            const int base = ((g - k) << 2) + k;
            const int i0 = fft_pad(base), i1 = fft_pad(base + step), i2 = fft_pad(base + 2 * step), i3 = fft_pad(base + 3 * step);
            const cpx wa = w1[k], wb = w2[k], wc = w2[k + step];
            c32 a0 = lds[i0], a1 = lds[i1], a2 = lds[i2], a3 = lds[i3];
            bfly(a0, a1, c_mul(a1, c32{wa.re, wa.im}));  // stage `step`, blocks of 2 * step
            bfly(a2, a3, c_mul(a3, c32{wa.re, wa.im}));
            bfly(a0, a2, c_mul(a2, c32{wb.re, wb.im}));  // stage `2 * step`, blocks of 4 * step
            bfly(a1, a3, c_mul(a3, c32{wc.re, wc.im}));
            lds[i0] = a0;
            lds[i1] = a1;
            lds[i2] = a2;
            lds[i3] = a3;
        }
    }
    for (; step < nf; step <<= 1) {  // an odd stage left over
        __syncthreads();
        const cpx *w = tb.fft_merge + (step - 32);  // W_{2*step}: offset (2*step)/2 - 32
#pragma unroll 4
        for (int b = (int)threadIdx.x; b < points / 2; b += (int)blockDim.x) {
            const int k = b & (step - 1);
            const int e = ((b - k) << 1) + k;  // block start * 2 + k
            const int ie = fft_pad(e), io = fft_pad(e + step);
            const cpx wk = w[k];
            c32 ev = lds[ie], ov = lds[io];
            const c32 q = c_mul(ov, c32{wk.re, wk.im});  // o * w
            bfly(ev, ov, q);
            lds[ie] = ev;
            lds[io] = ov;
        }
    }
    __syncthreads();
}

}  // namespace symaccel
