// The 1024- and 2048-point register-pass kernels (Fft of 1024 / 2048 points, Imdct of 2048 / 4096 lines): symaccel_fft_c32* /
// symaccel_imdct_f32* for those sizes (imdct_generic.hip dispatches here).  A translation unit of its own because it is built with
// complex arithmetic as packed f32 on register pairs (dsp_device.h): the 2048-point Fft gains 17 % (4.4 -> 5.1 TB/s,
// profiles/r03z_packed_sustained.txt), the other three instantiations are unchanged; the LDS-staged generic kernels lose 3-12 % packed.
//
// Reference: symphonia-core/src/dsp/fft/no_simd.rs:70-141, 221-454; dsp/mdct.rs:67-146.
#define SYM_PACKED_C32_DEFAULT 1
#include "imdct_wave.h"

namespace symaccel {

namespace {

constexpr int kWaveWaves = 4;

// ---- 1024 and 2048 points (Imdct of 2048 / 4096 lines): R = P / 512 sub-transforms of 512 points, one after the other through the same
// register passes, then the last log2 R radix-2 stages IN REGISTERS.  With bit-reversed DIT input, position block r (512 positions)
// is the transform of the inputs with index = rev(r) mod R -- x[R m + c], m = 0..511 -- so a lane that owns m = lane + 64 s for every
// c holds, after the R passes, positions 512 r + 64 B + lane of every block: exactly the operands the last stages pair (p, p + 512
// with W_1024[p]; p, p + 1024 with W_2048[p], no_simd.rs:221-281), lane-local.  Its inputs are R consecutive complex values
// (Imdct: 2 R consecutive lines) per (lane, s): 16 or 32 contiguous bytes per lane, contiguous across lanes.
template <int R>
__device__ __forceinline__ void fft_big_regs(c32 (&x)[R][8], int lane, c32 *lds, const LaneTables &lt, const c32 *w_merge) {
    static_assert(R == 2 || R == 4, "two or four 512-point sub-transforms");
    // pass r transforms the inputs with index = rev(r) (mod R); its result replaces them in x[rev(r)]
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const int c = R == 2 ? r : ((r & 1) << 1 | (r >> 1));
        fft_wave_multi(x[c], lane, lds, lt, 9);  // x[c][B] = position 512 r + 64 B + lane
    }
    auto blk = [](int r) { return R == 2 ? r : ((r & 1) << 1 | (r >> 1)); };  // where block r lives
    // step 512: blocks (0, 1) and (2, 3), twiddle W_1024[64 B + lane] (fft_merge offset 480)
#pragma unroll
    for (int r = 0; r < R; r += 2)
#pragma unroll
        for (int B = 0; B < 8; ++B) bfly(x[blk(r)][B], x[blk(r + 1)][B], c_mul(x[blk(r + 1)][B], w_merge[64 * B + lane]));
    if constexpr (R == 4) {
        // step 1024: blocks (0, 2) and (1, 3), twiddle W_2048[512 r + 64 B + lane] (fft_merge offset 992 = w_merge + 512)
#pragma unroll
        for (int r = 0; r < 2; ++r)
#pragma unroll
            for (int B = 0; B < 8; ++B) bfly(x[blk(r)][B], x[blk(r + 2)][B], c_mul(x[blk(r + 2)][B], w_merge[512 + 512 * r + 64 * B + lane]));
    }
}
template <int R>
__device__ __forceinline__ constexpr int big_blk(int r) { return R == 2 ? r : ((r & 1) << 1 | (r >> 1)); }

// MODE 0: Fft, 1: Ifft (re <-> im on the way in, swap + 1 / n on the way out, no_simd.rs:160-186)
template <int R, int MODE>
__global__ __launch_bounds__(64 * kWaveWaves, 2) void fft_big_wave_kernel(DevTables tb, const float *__restrict__ in, float *__restrict__ out, size_t count,
                                                                           unsigned per_wave, float c) {
    constexpr int P = 512 * R;
    __shared__ __attribute__((aligned(16))) c32 w_merge[512 + (R == 4 ? 1024 : 0)];  // W_1024 | W_2048
    __shared__ __attribute__((aligned(16))) float wave_lds[kWaveWaves][kWaveLds];
    for (int i = (int)threadIdx.x; i < 512 + (R == 4 ? 1024 : 0); i += 64 * kWaveWaves) w_merge[i] = ld_c(tb.fft_merge + 480 + i);
    __syncthreads();
    const int lane = (int)threadIdx.x & 63, wave = (int)threadIdx.x >> 6;
    c32 *lds = reinterpret_cast<c32 *>(wave_lds[wave]);
    LaneTables lt;
    load_lane_tables(tb, lane, lt);
    const size_t t0 = ((size_t)blockIdx.x * kWaveWaves + (size_t)wave) * per_wave;
    for (size_t t = t0; t < t0 + per_wave && t < count; ++t) {
        c32 x[R][8];
        const float4 *src = reinterpret_cast<const float4 *>(in + t * (size_t)(2 * P));
#pragma unroll
        for (int s = 0; s < 8; ++s)
#pragma unroll
            for (int h = 0; h < R / 2; ++h) {  // 2 complex values per 16-byte load: x[R m + 2 h], x[R m + 2 h + 1], m = lane + 64 s
                const float4 v = ld_stream(src + (size_t)(lane + 64 * s) * (R / 2) + h);
                x[2 * h][s] = MODE ? c32{v.y, v.x} : c32{v.x, v.y};
                x[2 * h + 1][s] = MODE ? c32{v.w, v.z} : c32{v.z, v.w};
            }
        fft_big_regs<R>(x, lane, lds, lt, w_merge);
        float2 *dst = reinterpret_cast<float2 *>(out + t * (size_t)(2 * P));
#pragma unroll
        for (int r = 0; r < R; ++r)
#pragma unroll
            for (int B = 0; B < 8; ++B) {
                const c32 v = x[big_blk<R>(r)][B];
                dst[512 * r + 64 * B + lane] = MODE ? make_float2(c * v.y, c * v.x) : make_float2(v.x, v.y);
            }
    }
}

template <int R>
__global__ __launch_bounds__(64 * kWaveWaves, 2) void imdct_big_wave_kernel(DevTables tb, const cpx *__restrict__ tw_g, const float *__restrict__ spec,
                                                                             float *__restrict__ out, size_t count, unsigned per_wave) {
    constexpr int P = 512 * R, N = 2 * P;
    __shared__ __attribute__((aligned(16))) c32 w_merge[512 + (R == 4 ? 1024 : 0)];
    __shared__ __attribute__((aligned(16))) c32 tw[P];  // the Imdct's own twiddles (pre- and post-twiddle, mdct.rs:45-54)
    __shared__ __attribute__((aligned(16))) float wave_lds[kWaveWaves][kWaveLds];
    for (int i = (int)threadIdx.x; i < 512 + (R == 4 ? 1024 : 0); i += 64 * kWaveWaves) w_merge[i] = ld_c(tb.fft_merge + 480 + i);
    for (int i = (int)threadIdx.x; i < P; i += 64 * kWaveWaves) tw[i] = ld_c(tw_g + i);
    __syncthreads();
    const int lane = (int)threadIdx.x & 63, wave = (int)threadIdx.x >> 6;
    float *ldsf = wave_lds[wave];
    c32 *lds = reinterpret_cast<c32 *>(ldsf);
    LaneTables lt;
    load_lane_tables(tb, lane, lt);
    const size_t t0 = ((size_t)blockIdx.x * kWaveWaves + (size_t)wave) * per_wave;
    for (size_t t = t0; t < t0 + per_wave && t < count; ++t) {
        // lines 2 R m .. 2 R m + 2 R - 1, m = lane + 64 s: the pairs (even line, odd line) of z-indices R m + c, c < R
        float4 v[8][R / 2];
        const float4 *src = reinterpret_cast<const float4 *>(spec + t * (size_t)N);
#pragma unroll
        for (int s = 0; s < 8; ++s)
#pragma unroll
            for (int h = 0; h < R / 2; ++h) v[s][h] = ld_stream(src + (size_t)(lane + 64 * s) * (R / 2) + h);
        c32 x[R][8];
        const int mirror = (63 - lane) * 4;
#pragma unroll
        for (int cc = 0; cc < R; ++cc)
#pragma unroll
            for (int s = 0; s < 8; ++s) {
                // z[i] = pre_twiddle(spec[2 i], spec[N - 1 - 2 i], tw[i]), i = R m + cc; the mirrored line is the odd line of pair
                // P - 1 - i = R (511 - m) + (R - 1 - cc): lane 63 - lane, load 7 - s, pair R - 1 - cc
                const int cm = R - 1 - cc;
                const float4 vm = v[7 - s][cm >> 1];
                const float odd_there = (cm & 1) ? vm.w : vm.y;
                const float mirrored = __int_as_float(__builtin_amdgcn_ds_bpermute(mirror, __float_as_int(odd_there)));
                const float4 vh = v[s][cc >> 1];
                const float even = (cc & 1) ? vh.z : vh.x;
                x[cc][s] = pre_twiddle(even, mirrored, tw[R * (lane + 64 * s) + cc]);
            }
        fft_big_regs<R>(x, lane, lds, lt, w_merge);
        // post-twiddle (mdct.rs:94-137): val = tw[p] * conj(X[p]); every value goes to one place in each of the four output
        // vectors (P samples each): one vector per round through the LDS work area, stored 16 B per lane
        float *o = out + t * (size_t)(2 * N);
#pragma unroll
        for (int round = 0; round < 4; ++round) {
#pragma unroll
            for (int r = 0; r < R; ++r)
#pragma unroll
                for (int B = 0; B < 8; ++B) {
                    const int p = 512 * r + 64 * B + lane;
                    const c32 val = post_twiddle(x[big_blk<R>(r)][B], tw[p]);
                    constexpr int n4 = P / 2;
                    float f;
                    int at;
                    if (p < n4) {
                        const int fi = 2 * p, ri = P - 1 - 2 * p;
                        f = round == 0 ? -val.y : (round == 1 ? val.y : val.x);
                        at = (round == 0 || round == 2) ? ri : fi;
                    } else {
                        const int i = p - n4;
                        const int fi = 2 * i, ri = P - 1 - 2 * i;
                        f = round == 0 ? -val.x : (round == 1 ? val.x : val.y);
                        at = (round == 0 || round == 2) ? fi : ri;
                    }
                    ldsf[at] = f;
                }
            wave_sync();
            float4 *o4 = reinterpret_cast<float4 *>(o + (size_t)round * P);
#pragma unroll
            for (int q = 0; q < P / 256; ++q) st_stream(o4 + lane + 64 * q, reinterpret_cast<const float4 *>(ldsf)[lane + 64 * q]);
            wave_sync();
        }
    }
}

}  // namespace

int launch_fft_big_wave(symaccel_ctx *ctx, int n, const float *d_in, float *d_out, size_t count, bool inverse) {
    if (n != 1024 && n != 2048) return SYMACCEL_ERR_INVALID_ARG;
    size_t per_wave = count / ((size_t)ctx->n_cus * 8 * 4);
    per_wave = per_wave < 1 ? 1 : (per_wave > 16 ? 16 : per_wave);
    const size_t waves = (count + per_wave - 1) / per_wave;
    const size_t grid = (waves + kWaveWaves - 1) / kWaveWaves;
    if (grid > 0x7fffffffu) return SYMACCEL_ERR_INVALID_ARG;
#define SYM_BIG_FFT(R, MODE)                                                                                                                   \
    hipLaunchKernelGGL((fft_big_wave_kernel<R, MODE>), dim3((unsigned)grid), dim3(64 * kWaveWaves), 0, ctx->stream, ctx->dev, d_in, d_out, count, \
                       (unsigned)per_wave, 1.0f / (float)n)
    if (n == 1024) { if (inverse) SYM_BIG_FFT(2, 1); else SYM_BIG_FFT(2, 0); }
    else { if (inverse) SYM_BIG_FFT(4, 1); else SYM_BIG_FFT(4, 0); }
#undef SYM_BIG_FFT
    SYM_GPU(ctx, hipGetLastError());
    return SYMACCEL_OK;
}

int launch_imdct_big_wave(symaccel_ctx *ctx, const cpx *d_twiddle, int nf, const float *d_spec, float *d_out, size_t count) {
    if (nf != 1024 && nf != 2048) return SYMACCEL_ERR_INVALID_ARG;
    size_t per_wave = count / ((size_t)ctx->n_cus * 8 * 4);
    per_wave = per_wave < 1 ? 1 : (per_wave > 16 ? 16 : per_wave);
    const size_t waves = (count + per_wave - 1) / per_wave;
    const size_t grid = (waves + kWaveWaves - 1) / kWaveWaves;
    if (grid > 0x7fffffffu) return SYMACCEL_ERR_INVALID_ARG;
    if (nf == 1024)
        hipLaunchKernelGGL(imdct_big_wave_kernel<2>, dim3((unsigned)grid), dim3(64 * kWaveWaves), 0, ctx->stream, ctx->dev, d_twiddle, d_spec,
                           d_out, count, (unsigned)per_wave);
    else
        hipLaunchKernelGGL(imdct_big_wave_kernel<4>, dim3((unsigned)grid), dim3(64 * kWaveWaves), 0, ctx->stream, ctx->dev, d_twiddle, d_spec,
                           d_out, count, (unsigned)per_wave);
    SYM_GPU(ctx, hipGetLastError());
    return SYMACCEL_OK;
}

}  // namespace symaccel
