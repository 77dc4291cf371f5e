// Generic batched Fft / Imdct kernels (any power-of-two size the reference's decoders can ask
// for).  These serve the `symaccel_fft_c32*` / `symaccel_imdct_f32*` entry points; the AAC, MP3 and
// Vorbis paths have their own fused kernels.
//
// Reference: symphonia-core/src/dsp/fft/no_simd.rs:70-141, 221-454; dsp/mdct.rs:67-146.
// Bound: HBM (n*4 B in, n*8 B out per transform; ~5 flop/B without FMA).
#include "fft_lds.h"

namespace symaccel {

namespace {

constexpr int kThreads = 256;
constexpr int kMaxPoints = 4096;

__device__ __forceinline__ int tile_points(int nf) { return nf >= 2048 ? nf : 2048; }

__global__ __launch_bounds__(kThreads) void fft_kernel(DevTables tb, int nf, int log2nf, const float2 *in,
                                                       float2 *out, size_t count) {
    __shared__ c32 lds[fft_padded(kMaxPoints)];
    const int points = tile_points(nf);
    const int per_wg = points / nf;
    const size_t first = (size_t)blockIdx.x * (size_t)per_wg;
    // Fft::fft (no_simd.rs:121-140): y[i] = x[perm[i]]; in-place variant swaps to the same order.
    for (int idx = (int)threadIdx.x; idx < points; idx += kThreads) {
        const int t = idx >> log2nf, i = idx & (nf - 1);
        c32 v{0.0f, 0.0f};
        if (first + (size_t)t < count) {
            const float2 x = in[(first + (size_t)t) * (size_t)nf + (size_t)i];
            v = c32{x.x, x.y};
        }
        lds[fft_pad((t << log2nf) + (int)rev_bits((unsigned)i, log2nf))] = v;
    }
    wg_fft_lds(lds, nf, points, tb);
    for (int idx = (int)threadIdx.x; idx < points; idx += kThreads) {
        const int t = idx >> log2nf;
        if (first + (size_t)t < count) {
            const c32 v = lds[fft_pad(idx)];
            out[(first + (size_t)t) * (size_t)nf + (size_t)(idx & (nf - 1))] = make_float2(v.x, v.y);
        }
    }
}

__global__ __launch_bounds__(kThreads) void imdct_kernel(DevTables tb, const cpx *tw, int n, int log2nf,
                                                         const float *spec, float *out, size_t count) {
    __shared__ c32 lds[fft_padded(kMaxPoints)];
    const int nf = n >> 1, n4 = n >> 2;
    const int points = tile_points(nf);
    const int per_wg = points / nf;
    const size_t first = (size_t)blockIdx.x * (size_t)per_wg;
    // pre-FFT twiddle (mdct.rs:81-88), stored straight into bit-reversed order
    for (int idx = (int)threadIdx.x; idx < points; idx += kThreads) {
        const int t = idx >> log2nf, i = idx & (nf - 1);
        c32 z{0.0f, 0.0f};
        if (first + (size_t)t < count) {
            const float *s = spec + (first + (size_t)t) * (size_t)n;
            const cpx w = tw[i];
            z = pre_twiddle(s[2 * i], s[n - 1 - 2 * i], c32{w.re, w.im});
        }
        lds[fft_pad((t << log2nf) + (int)rev_bits((unsigned)i, log2nf))] = z;
    }
    wg_fft_lds(lds, nf, points, tb);
    // post-FFT twiddle and expansion into the four quarter vectors (mdct.rs:94-137)
    for (int idx = (int)threadIdx.x; idx < points; idx += kThreads) {
        const int t = idx >> log2nf, k = idx & (nf - 1);
        if (first + (size_t)t >= count) continue;
        float *o = out + (first + (size_t)t) * 2 * (size_t)n;
        float *vec0 = o, *vec1 = o + nf, *vec2 = o + 2 * nf, *vec3 = o + 3 * nf;
        const c32 x = lds[fft_pad(idx)];
        const cpx w = tw[k];
        const c32 val = post_twiddle(x, c32{w.re, w.im});  // w * x.conj()
        if (k < n4) {
            const int fi = 2 * k, ri = nf - 1 - 2 * k;
            vec0[ri] = -val.y;
            vec1[fi] = val.y;
            vec2[ri] = val.x;
            vec3[fi] = val.x;
        } else {
            const int i = k - n4;
            const int fi = 2 * i, ri = nf - 1 - 2 * i;
            vec0[fi] = -val.x;
            vec1[ri] = val.x;
            vec2[fi] = val.y;
            vec3[ri] = val.y;
        }
    }
}

int ilog2(int v) {
    int l = 0;
    while ((1 << l) < v) ++l;
    return l;
}

}  // namespace

int launch_fft(symaccel_ctx *ctx, int n, const float *d_in, float *d_out, size_t count) {
    const int points = n >= 2048 ? n : 2048;
    const size_t per_wg = (size_t)(points / n);
    const size_t grid = (count + per_wg - 1) / per_wg;
    if (grid > 0x7fffffffu) return SYMACCEL_ERR_INVALID_ARG;
    hipLaunchKernelGGL(fft_kernel, dim3((unsigned)grid), dim3(kThreads), 0, ctx->stream, ctx->dev, n, ilog2(n),
                       (const float2 *)d_in, (float2 *)d_out, count);
    SYM_GPU(ctx, hipGetLastError());
    return SYMACCEL_OK;
}

int launch_imdct(symaccel_ctx *ctx, const ImdctPlan &plan, const float *d_spec, float *d_out, size_t count) {
    const int nf = plan.n / 2;
    const int points = nf >= 2048 ? nf : 2048;
    const size_t per_wg = (size_t)(points / nf);
    const size_t grid = (count + per_wg - 1) / per_wg;
    if (grid > 0x7fffffffu) return SYMACCEL_ERR_INVALID_ARG;
    hipLaunchKernelGGL(imdct_kernel, dim3((unsigned)grid), dim3(kThreads), 0, ctx->stream, ctx->dev,
                       (const cpx *)plan.d_twiddle, plan.n, ilog2(nf), d_spec, d_out, count);
    SYM_GPU(ctx, hipGetLastError());
    return SYMACCEL_OK;
}

}  // namespace symaccel
