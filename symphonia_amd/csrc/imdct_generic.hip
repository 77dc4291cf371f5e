// Generic batched Fft / Imdct kernels (any power-of-two size the reference's decoders can ask
// for).  These serve the `symaccel_fft_c32*` / `symaccel_imdct_f32*` entry points; the AAC, MP3 and
// Vorbis paths have their own fused kernels.
//
// Reference: symphonia-core/src/dsp/fft/no_simd.rs:70-141, 221-454; dsp/mdct.rs:67-146.
// Bound: HBM (n*4 B in, n*8 B out per transform; ~5 flop/B without FMA).
#include "fft_lds.h"
#include "imdct_wave.h"

namespace symaccel {

namespace {

constexpr int kThreads = 256;
constexpr int kMaxPoints = 4096;

__device__ __forceinline__ int tile_points(int nf) { return nf >= 2048 ? nf : 2048; }

// inverse: Ifft::ifft (no_simd.rs:143-219) = the same forward transform between a re <-> im swap on the way in and a swap
// with the 1/n scale on the way out.
__global__ __launch_bounds__(kThreads) void fft_kernel(DevTables tb, int nf, int log2nf, const float2 *in,
                                                       float2 *out, size_t count, int inverse, float c) {
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
            v = inverse ? c32{x.y, x.x} : c32{x.x, x.y};
        }
        lds[fft_pad((t << log2nf) + (int)rev_bits((unsigned)i, log2nf))] = v;
    }
    wg_fft_lds(lds, nf, points, tb, !inverse);
    // c = 1.0 / n as f32 (no_simd.rs:181, 212), formed on the host: a device division expands into fused operations
    for (int idx = (int)threadIdx.x; idx < points; idx += kThreads) {
        const int t = idx >> log2nf;
        if (first + (size_t)t < count) {
            const c32 v = lds[fft_pad(idx)];
            out[(first + (size_t)t) * (size_t)nf + (size_t)(idx & (nf - 1))] = inverse ? make_float2(c * v.y, c * v.x) : make_float2(v.x, v.y);
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

// ---- Transforms of more than 4096 complex points (Fft 8192 .. 65536, Imdct 16384 .. 131072: legal in the reference,
// no_simd.rs:77-80 / mdct.rs:37-40, used by none of its codecs).  The reference's breadth-first schedule is kept: the
// permuted input is transformed in 4096-point tiles in LDS (stages up to step 2048, exactly wg_fft_lds), the remaining
// merge stages (step 4096 .. nf / 2, no_simd.rs:247-279) run as one global-memory pass each over a scratch copy.  Every
// butterfly is the reference's; correctness over speed (these sizes are nobody's hot path).
enum : int { kBigFft = 0, kBigIfft = 1, kBigImdct = 2 };

template <int MODE>
__global__ __launch_bounds__(kThreads) void fft_big_tile_kernel(DevTables tb, int nf, int log2nf, const void *src_v, const cpx *tw,
                                                                c32 *work, size_t count) {
    __shared__ c32 lds[fft_padded(kMaxPoints)];
    const unsigned tiles = (unsigned)(nf >> 12);
    const size_t t = blockIdx.x / tiles;
    const unsigned j = blockIdx.x % tiles;
    for (int idx = (int)threadIdx.x; idx < 4096; idx += kThreads) {
        // y[pos] = x[perm[pos]] (no_simd.rs:101-107, 126-128; Ifft :167-169 with the re <-> im swap; Imdct: the
        // pre-twiddled z of mdct.rs:81-88)
        const unsigned pos = (j << 12) + (unsigned)idx;
        const unsigned i = rev_bits(pos, log2nf);
        c32 v;
        if constexpr (MODE == kBigImdct) {
            const int n = nf << 1;
            const float *s = static_cast<const float *>(src_v) + t * (size_t)n;
            const cpx w = tw[i];
            v = pre_twiddle(s[2 * i], s[n - 1 - 2 * (int)i], c32{w.re, w.im});
        } else {
            const float2 x = static_cast<const float2 *>(src_v)[t * (size_t)nf + i];
            v = MODE == kBigIfft ? c32{x.y, x.x} : c32{x.x, x.y};
        }
        lds[fft_pad(idx)] = v;
    }
    wg_fft_lds(lds, 4096, 4096, tb);
    for (int idx = (int)threadIdx.x; idx < 4096; idx += kThreads) work[t * (size_t)nf + ((size_t)j << 12) + (size_t)idx] = lds[fft_pad(idx)];
}

// one merge stage (no_simd.rs:222-238): q = o * w; e' = e + q; o' = e - q.  `scale_swap`: the last stage of an Ifft
// writes (c * im, c * re) (no_simd.rs:181-186).
__global__ __launch_bounds__(kThreads) void fft_big_merge_kernel(DevTables tb, int nf, int step, const c32 *src, c32 *dst, size_t count,
                                                                 int scale_swap, float c) {
    const size_t half = (size_t)nf >> 1, total = count * half;
    for (size_t g = (size_t)blockIdx.x * kThreads + threadIdx.x; g < total; g += (size_t)gridDim.x * kThreads) {
        const size_t t = g / half;
        const unsigned b = (unsigned)(g % half);
        const unsigned k = b & (unsigned)(step - 1);
        const size_t e = t * (size_t)nf + (((size_t)(b - k)) << 1) + k, o = e + (size_t)step;
        const cpx wk = tb.fft_merge[(size_t)(step - 32) + k];  // W_{2 * step}
        c32 ev = src[e], ov = src[o];
        const c32 q = c_mul(ov, c32{wk.re, wk.im});
        bfly(ev, ov, q);
        if (scale_swap) {
            ev = c32{c * ev.y, c * ev.x};
            ov = c32{c * ov.y, c * ov.x};
        }
        dst[e] = ev;
        dst[o] = ov;
    }
}

// post-FFT twiddle and expansion into the four quarter vectors (mdct.rs:94-137), from the scratch copy
__global__ __launch_bounds__(kThreads) void imdct_big_post_kernel(const cpx *tw, int n, const c32 *work, float *out, size_t count) {
    const int nf = n >> 1, n4 = n >> 2;
    const size_t total = count * (size_t)nf;
    for (size_t g = (size_t)blockIdx.x * kThreads + threadIdx.x; g < total; g += (size_t)gridDim.x * kThreads) {
        const size_t t = g / (size_t)nf;
        const int k = (int)(g % (size_t)nf);
        float *o = out + t * 2 * (size_t)n;
        float *vec0 = o, *vec1 = o + nf, *vec2 = o + 2 * nf, *vec3 = o + 3 * nf;
        const c32 x = work[g];
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

// ---- wavefront-per-transform fast paths for the two sizes the AAC and Vorbis 256/2048 decoders use (imdct_wave.h):
// n = 1024 (one 512-point FFT in three radix-8 register passes) and n = 128 (eight 64-point FFTs at once).
constexpr int kWaveWaves = 4;

__global__ __launch_bounds__(64 * kWaveWaves, 2) void imdct1024_wave_kernel(DevTables tb, const cpx *__restrict__ tw_g,
                                                                             const float *__restrict__ spec,
                                                                             float *__restrict__ out, size_t count,
                                                                             unsigned per_wave) {
    __shared__ __attribute__((aligned(16))) float tw_lds[1024];
    __shared__ __attribute__((aligned(16))) float wave_lds[kWaveWaves][kWaveLds];
    for (int i = (int)threadIdx.x; i < 1024; i += 64 * kWaveWaves) tw_lds[i] = reinterpret_cast<const float *>(tw_g)[i];
    __syncthreads();
    const int lane = (int)threadIdx.x & 63, wave = (int)threadIdx.x >> 6;
    const size_t first = ((size_t)blockIdx.x * kWaveWaves + (size_t)wave) * per_wave;
    if (first >= count) return;
    const size_t last = first + per_wave < count ? first + per_wave : count;
    c32 *lds = reinterpret_cast<c32 *>(wave_lds[wave]);
    const c32 *tw = reinterpret_cast<const c32 *>(tw_lds);
    LaneTables lt;
    load_lane_tables(tb, lane, lt);
    float2 line[8];
    {
        const float2 *src = reinterpret_cast<const float2 *>(spec + first * 1024);
#pragma unroll
        for (int s = 0; s < 8; ++s) line[s] = src[lane + 64 * s];
    }
    for (size_t t = first; t < last; ++t) {
        c32 z[8];
        const int mirror = (63 - lane) * 4;
#pragma unroll
        for (int s = 0; s < 8; ++s) {
            const float mirrored = __int_as_float(__builtin_amdgcn_ds_bpermute(mirror, __float_as_int(line[7 - s].y)));
            z[s] = pre_twiddle(line[s].x, mirrored, tw[lane + 64 * s]);
        }
        if (t + 1 < last) {
            const float2 *src = reinterpret_cast<const float2 *>(spec + (t + 1) * 1024);
#pragma unroll
            for (int s = 0; s < 8; ++s) line[s] = src[lane + 64 * s];
        }
        fft512_wave(z, lane, lds, lt);
        float *o = out + t * 2048;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            float x[8], x2[8];
            post_slot(lds, tw, lane + 64 * h, x, x2);
            store_slot(o, lane + 64 * h, x);
            store_slot(o + 1024, lane + 64 * h, x2);
        }
        wave_sync();  // Z in LDS is overwritten by the next transform
    }
}

__global__ __launch_bounds__(64 * kWaveWaves, 2) void imdct128_wave_kernel(DevTables tb, const cpx *__restrict__ tw_g,
                                                                            const float *__restrict__ spec,
                                                                            float *__restrict__ out, size_t count,
                                                                            unsigned groups_per_wave) {
    __shared__ __attribute__((aligned(16))) float wave_lds[kWaveWaves][kWaveLds];
    const int lane = (int)threadIdx.x & 63, wave = (int)threadIdx.x >> 6;
    float *ldsf = wave_lds[wave];
    LaneTables lt;
    load_lane_tables(tb, lane, lt);
    const size_t g0 = ((size_t)blockIdx.x * kWaveWaves + (size_t)wave) * groups_per_wave;  // groups of 8 transforms
    for (size_t g = g0; g < g0 + groups_per_wave && g * 8 < count; ++g) {
        const size_t t0 = g * 8;
        const int nt = (int)(count - t0 < 8 ? count - t0 : 8);
        const float2 *src = reinterpret_cast<const float2 *>(spec + t0 * 128);
#pragma unroll
        for (int s = 0; s < 8; ++s) {
            if (s < nt) {
                const float2 v = src[lane + 64 * s];
                ldsf[short_row(s) + 2 * lane] = v.x;
                ldsf[short_row(s) + 2 * lane + 1] = v.y;
            }
        }
        wave_sync();
        imdct_short_wave(lane, ldsf, tw_g, lt);  // H[w] = ldsf[short_row(w) ..]; ends with a wave_sync
        const int w = lane >> 3, c = lane & 7;
        if (w < nt) {
            float4 *o4 = reinterpret_cast<float4 *>(out + (t0 + (size_t)w) * 256);
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                float v[4];
                ys4(ldsf, w, 4 * (c + 8 * e), v);
                o4[c + 8 * e] = make_float4(v[0], v[1], v[2], v[3]);
            }
        }
        wave_sync();
    }
}

int ilog2(int v) {
    int l = 0;
    while ((1 << l) < v) ++l;
    return l;
}

}  // namespace

// The big-transform driver: src -> scratch (tiles), merge stages in scratch, the last one into `final_dst` (Fft / Ifft) or,
// for Imdct (final_dst == nullptr), staying in scratch for the post-twiddle.  Returns the scratch pointer in *work_out.
template <int MODE>
static int run_big_fft(symaccel_ctx *ctx, int nf, const void *src, const cpx *tw, c32 *final_dst, size_t count, c32 **work_out) {
    const size_t tiles = (size_t)(nf >> 12);
    if (count * tiles > 0x7fffffffu) return SYMACCEL_ERR_INVALID_ARG;
    void *scratch = nullptr;
    SYM_TRY(ctx_scratch(ctx, count * (size_t)nf * sizeof(c32), &scratch));
    c32 *work = static_cast<c32 *>(scratch);
    hipLaunchKernelGGL(fft_big_tile_kernel<MODE>, dim3((unsigned)(count * tiles)), dim3(kThreads), 0, ctx->stream, ctx->dev, nf, ilog2(nf),
                       src, tw, work, count);
    const size_t pairs = count * (size_t)(nf >> 1);
    const unsigned grid = (unsigned)((pairs + kThreads - 1) / kThreads < 65536 ? (pairs + kThreads - 1) / kThreads : 65536);
    for (int step = 4096; step < nf; step <<= 1) {
        const bool last = (step << 1) == nf;
        c32 *dst = last && final_dst ? final_dst : work;
        hipLaunchKernelGGL(fft_big_merge_kernel, dim3(grid), dim3(kThreads), 0, ctx->stream, ctx->dev, nf, step, (const c32 *)work, dst, count,
                           (last && MODE == kBigIfft) ? 1 : 0, 1.0f / (float)nf);
    }
    SYM_GPU(ctx, hipGetLastError());
    if (work_out) *work_out = work;
    return SYMACCEL_OK;
}

int launch_fft(symaccel_ctx *ctx, int n, const float *d_in, float *d_out, size_t count, bool inverse) {
    if (n > kMaxPoints)  // (reads everything of d_in before the last stage writes d_out: d_in == d_out is fine)
        return inverse ? run_big_fft<kBigIfft>(ctx, n, d_in, nullptr, reinterpret_cast<c32 *>(d_out), count, nullptr)
                       : run_big_fft<kBigFft>(ctx, n, d_in, nullptr, reinterpret_cast<c32 *>(d_out), count, nullptr);
    const int points = n >= 2048 ? n : 2048;
    const size_t per_wg = (size_t)(points / n);
    const size_t grid = (count + per_wg - 1) / per_wg;
    if (grid > 0x7fffffffu) return SYMACCEL_ERR_INVALID_ARG;
    hipLaunchKernelGGL(fft_kernel, dim3((unsigned)grid), dim3(kThreads), 0, ctx->stream, ctx->dev, n, ilog2(n),
                       (const float2 *)d_in, (float2 *)d_out, count, inverse ? 1 : 0, 1.0f / (float)n);
    SYM_GPU(ctx, hipGetLastError());
    return SYMACCEL_OK;
}

int launch_imdct(symaccel_ctx *ctx, const ImdctPlan &plan, const float *d_spec, float *d_out, size_t count) {
    if (plan.n == 1024 || plan.n == 128) {
        // enough wavefronts to fill the chip several times over, several transforms per wavefront
        const size_t units = plan.n == 1024 ? count : (count + 7) / 8;
        size_t per_wave = units / ((size_t)ctx->n_cus * 8 * 4);
        per_wave = per_wave < 1 ? 1 : (per_wave > 32 ? 32 : per_wave);
        const size_t waves = (units + per_wave - 1) / per_wave;
        const size_t grid = (waves + kWaveWaves - 1) / kWaveWaves;
        if (grid > 0x7fffffffu) return SYMACCEL_ERR_INVALID_ARG;
        if (plan.n == 1024)
            hipLaunchKernelGGL(imdct1024_wave_kernel, dim3((unsigned)grid), dim3(64 * kWaveWaves), 0, ctx->stream, ctx->dev,
                               (const cpx *)plan.d_twiddle, d_spec, d_out, count, (unsigned)per_wave);
        else
            hipLaunchKernelGGL(imdct128_wave_kernel, dim3((unsigned)grid), dim3(64 * kWaveWaves), 0, ctx->stream, ctx->dev,
                               (const cpx *)plan.d_twiddle, d_spec, d_out, count, (unsigned)per_wave);
        SYM_GPU(ctx, hipGetLastError());
        return SYMACCEL_OK;
    }
    const int nf = plan.n / 2;
    if (nf > kMaxPoints) {
        c32 *work = nullptr;
        SYM_TRY((run_big_fft<kBigImdct>(ctx, nf, d_spec, (const cpx *)plan.d_twiddle, nullptr, count, &work)));
        const size_t total = count * (size_t)nf;
        const unsigned g = (unsigned)((total + kThreads - 1) / kThreads < 65536 ? (total + kThreads - 1) / kThreads : 65536);
        hipLaunchKernelGGL(imdct_big_post_kernel, dim3(g), dim3(kThreads), 0, ctx->stream, (const cpx *)plan.d_twiddle, plan.n,
                           (const c32 *)work, d_out, count);
        SYM_GPU(ctx, hipGetLastError());
        return SYMACCEL_OK;
    }
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
