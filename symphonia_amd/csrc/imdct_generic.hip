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
// SYM_MULTI_WAVE (build knob): 1 = transforms of 16 .. 512 points run on the register-pass kernels (fft_wave_multi), 0 = on the LDS-staged
// generic kernels as before (kept for the A/B and for every larger size).
#ifndef SYM_MULTI_WAVE
#define SYM_MULTI_WAVE 1
#endif
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

// ---- wavefront-per-transform fast path for n = 1024 (one 512-point FFT in three radix-8 register passes, imdct_wave.h; every other
// size up to 1024 lines runs the multi-transform kernel below -- n = 128 used to have a kernel of its own here, 3.7 TB/s against the
// 4.9 TB/s of the general one, profiles/r03g_core_transforms_ab.txt).
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

// ---- every size from 16 to 512 points on the register-pass FFT: 512 / P transforms per wavefront pass (fft_wave_multi, imdct_wave.h).
// A group = 512 / P consecutive transforms = 1024 input floats (Imdct: lines; Fft: 512 complex points) and, for Imdct, 2048 output
// floats, both contiguous in the batch: loads and stores are 16 B per lane, 1 KiB per instruction.
// Sizes whose per-lane input pieces would be shorter than 128 B (P <= 64) go through an LDS staging copy of the group's input.
__global__ __launch_bounds__(64 * kWaveWaves, 2) void imdct_multi_wave_kernel(DevTables tb, const cpx *__restrict__ tw_g, int logp,
                                                                               const float *__restrict__ spec, float *__restrict__ out,
                                                                               size_t count, unsigned groups_per_wave) {
    __shared__ __attribute__((aligned(16))) float tw_lds[1024];
    __shared__ __attribute__((aligned(16))) float wave_lds[kWaveWaves][kWaveLds];
    const int P = 1 << logp;
    for (int i = (int)threadIdx.x; i < 2 * P; i += 64 * kWaveWaves) tw_lds[i] = reinterpret_cast<const float *>(tw_g)[i];
    __syncthreads();
    const int lane = (int)threadIdx.x & 63, wave = (int)threadIdx.x >> 6;
    float *ldsf = wave_lds[wave];
    c32 *lds = reinterpret_cast<c32 *>(ldsf);
    const c32 *tw = reinterpret_cast<const c32 *>(tw_lds);
    LaneTables lt;
    load_lane_tables(tb, lane, lt);
    const size_t per_group = (size_t)512 >> logp;  // transforms per group
    const size_t n_groups = (count + per_group - 1) / per_group;
    const size_t g0 = ((size_t)blockIdx.x * kWaveWaves + (size_t)wave) * groups_per_wave;
    if (g0 >= n_groups) return;
    const size_t g1 = g0 + groups_per_wave < n_groups ? g0 + groups_per_wave : n_groups;
    float4 v[4];
    multi_fetch(spec + g0 * 1024, (count - g0 * per_group) * (size_t)(2 * P), lane, v);
    for (size_t g = g0; g < g1; ++g) {
        // the group's 1024 lines -> LDS (natural order), from there each lane's pairs and their mirrored odd lines
#pragma unroll
        for (int q = 0; q < 4; ++q) reinterpret_cast<float4 *>(ldsf)[lane + 64 * q] = v[q];
        wave_sync();
        if (g + 1 < g1) multi_fetch(spec + (g + 1) * 1024, (count - (g + 1) * per_group) * (size_t)(2 * P), lane, v);
        c32 z[8];
        {
            const int gbits = logp - 3, G = 1 << gbits;
            const int T = lane >> gbits, u = lane & (G - 1);
            const float *sT = ldsf + ((size_t)T << (logp + 1));
#pragma unroll
            for (int s = 0; s < 8; ++s) {
                const int i = u + (s << gbits);
                const float2 pr = *reinterpret_cast<const float2 *>(sT + 2 * i);
                z[s] = pre_twiddle(pr.x, sT[2 * P - 1 - 2 * i], tw[i]);
            }
        }
        wave_sync();
        fft_wave_multi(z, lane, lds, lt, logp);
        multi_post_twiddle(z, lane, logp, tw, ldsf);
        wave_sync();
        {
            float4 *o4 = reinterpret_cast<float4 *>(out + g * 2048);
            const size_t valid = (count - g * per_group) * (size_t)(4 * P);  // output floats of this group that exist
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const int i4 = lane + 64 * q;
                if ((size_t)(4 * i4) < valid) st_stream(o4 + i4, reinterpret_cast<const float4 *>(ldsf)[i4]);
            }
        }
        wave_sync();
    }
}

// Fft::fft / Ifft::ifft (no_simd.rs:96-219) for 16 .. 512 points: a group = 512 points = 512 / n transforms.
__global__ __launch_bounds__(64 * kWaveWaves, 2) void fft_multi_wave_kernel(DevTables tb, int logp, const float *__restrict__ in,
                                                                             float *__restrict__ out, size_t count, unsigned groups_per_wave,
                                                                             int inverse, float c) {
    __shared__ __attribute__((aligned(16))) float wave_lds[kWaveWaves][kWaveLds];
    const int P = 1 << logp;
    const int lane = (int)threadIdx.x & 63, wave = (int)threadIdx.x >> 6;
    float *ldsf = wave_lds[wave];
    c32 *lds = reinterpret_cast<c32 *>(ldsf);
    LaneTables lt;
    load_lane_tables(tb, lane, lt);
    const size_t per_group = (size_t)512 >> logp;
    const size_t n_groups = (count + per_group - 1) / per_group;
    const size_t g0 = ((size_t)blockIdx.x * kWaveWaves + (size_t)wave) * groups_per_wave;
    if (g0 >= n_groups) return;
    const size_t g1 = g0 + groups_per_wave < n_groups ? g0 + groups_per_wave : n_groups;
    float4 v[4];
    multi_fetch(in + g0 * 1024, (count - g0 * per_group) * (size_t)(2 * P), lane, v);
    for (size_t g = g0; g < g1; ++g) {
#pragma unroll
        for (int q = 0; q < 4; ++q) reinterpret_cast<float4 *>(ldsf)[lane + 64 * q] = v[q];
        wave_sync();
        if (g + 1 < g1) multi_fetch(in + (g + 1) * 1024, (count - (g + 1) * per_group) * (size_t)(2 * P), lane, v);
        c32 z[8];
        {
            const int gbits = logp - 3, G = 1 << gbits;
            const int T = lane >> gbits, u = lane & (G - 1);
            const c32 *xT = lds + ((size_t)T << logp);
#pragma unroll
            for (int s = 0; s < 8; ++s) {
                const c32 x = xT[u + (s << gbits)];
                z[s] = inverse ? c32{x.y, x.x} : x;  // Ifft: re <-> im on the way in (no_simd.rs:160-166)
            }
        }
        wave_sync();
        fft_wave_multi(z, lane, lds, lt, logp);
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const int p = logp <= 6 ? 64 * (lane >> 3) + 8 * q + (lane & 7) : 64 * q + lane;
            // c = 1.0 / n as f32 (no_simd.rs:181, 212), formed on the host
            lds[p] = inverse ? c32{c * z[q].y, c * z[q].x} : z[q];
        }
        wave_sync();
        {
            float4 *o4 = reinterpret_cast<float4 *>(out + g * 1024);
            const size_t valid = (count - g * per_group) * (size_t)(2 * P);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int i4 = lane + 64 * q;
                if ((size_t)(4 * i4) < valid) st_stream(o4 + i4, reinterpret_cast<const float4 *>(ldsf)[i4]);
            }
        }
        wave_sync();
    }
}

// (the 1024- and 2048-point register-pass kernels live in imdct_big.hip: complex arithmetic as packed f32 there)

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
    // (an Ifft of fewer than 32 points runs no butterflies in the reference, no_simd.rs:221-281: left to the generic kernel)
    if (n >= 16 && n <= 512 && !(inverse && n < 32) && (SYM_MULTI_WAVE & 3)) {
        const int logp = ilog2(n);
        const size_t groups = (count + (size_t)(512 >> logp) - 1) / (size_t)(512 >> logp);
        size_t per_wave = groups / ((size_t)ctx->n_cus * 8 * 4);
        per_wave = per_wave < 1 ? 1 : (per_wave > 32 ? 32 : per_wave);
        const size_t waves = (groups + per_wave - 1) / per_wave;
        const size_t grid = (waves + kWaveWaves - 1) / kWaveWaves;
        if (grid > 0x7fffffffu) return SYMACCEL_ERR_INVALID_ARG;
        hipLaunchKernelGGL(fft_multi_wave_kernel, dim3((unsigned)grid), dim3(64 * kWaveWaves), 0, ctx->stream, ctx->dev, logp, d_in, d_out,
                           count, (unsigned)per_wave, inverse ? 1 : 0, 1.0f / (float)n);
        SYM_GPU(ctx, hipGetLastError());
        return SYMACCEL_OK;
    }
    if ((n == 1024 || n == 2048) && (SYM_MULTI_WAVE & 1)) return launch_fft_big_wave(ctx, n, d_in, d_out, count, inverse);
    if (n == 4096 && (SYM_MULTI_WAVE & 1)) return launch_fft4096_wg(ctx, d_in, d_out, count, inverse);
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
    if (plan.n == 1024 && !(SYM_MULTI_WAVE & 2)) {
        // enough wavefronts to fill the chip several times over, several transforms per wavefront
        size_t per_wave = count / ((size_t)ctx->n_cus * 8 * 4);
        per_wave = per_wave < 1 ? 1 : (per_wave > 32 ? 32 : per_wave);
        const size_t waves = (count + per_wave - 1) / per_wave;
        const size_t grid = (waves + kWaveWaves - 1) / kWaveWaves;
        if (grid > 0x7fffffffu) return SYMACCEL_ERR_INVALID_ARG;
        hipLaunchKernelGGL(imdct1024_wave_kernel, dim3((unsigned)grid), dim3(64 * kWaveWaves), 0, ctx->stream, ctx->dev,
                           (const cpx *)plan.d_twiddle, d_spec, d_out, count, (unsigned)per_wave);
        SYM_GPU(ctx, hipGetLastError());
        return SYMACCEL_OK;
    }
    const int nf = plan.n / 2;
    if (nf >= 16 && nf <= 512 && (SYM_MULTI_WAVE & 3)) {
        const int logp = ilog2(nf);
        const size_t groups = (count + (size_t)(512 >> logp) - 1) / (size_t)(512 >> logp);
        size_t per_wave = groups / ((size_t)ctx->n_cus * 8 * 4);
        per_wave = per_wave < 1 ? 1 : (per_wave > 32 ? 32 : per_wave);
        const size_t waves = (groups + per_wave - 1) / per_wave;
        const size_t grid = (waves + kWaveWaves - 1) / kWaveWaves;
        if (grid > 0x7fffffffu) return SYMACCEL_ERR_INVALID_ARG;
        hipLaunchKernelGGL(imdct_multi_wave_kernel, dim3((unsigned)grid), dim3(64 * kWaveWaves), 0, ctx->stream, ctx->dev,
                           (const cpx *)plan.d_twiddle, logp, d_spec, d_out, count, (unsigned)per_wave);
        SYM_GPU(ctx, hipGetLastError());
        return SYMACCEL_OK;
    }
    if ((nf == 1024 || nf == 2048) && (SYM_MULTI_WAVE & 1))
        return launch_imdct_big_wave(ctx, (const cpx *)plan.d_twiddle, nf, d_spec, d_out, count);
    if (nf == 4096 && (SYM_MULTI_WAVE & 1)) return launch_imdct8192_wg(ctx, (const cpx *)plan.d_twiddle, d_spec, d_out, count);
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
