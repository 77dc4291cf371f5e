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
template <int R, class LT>
__device__ __forceinline__ void fft_big_regs(c32 (&x)[R][8], int lane, c32 *lds, const LT &lt, const c32 *w_merge) {
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

// ---- 4096 points (Fft of 4096 points, Imdct of 8192 lines): the four wavefronts of a WORKGROUP take one transform together.
// Position block q (1024 positions) of the bit-reversed DIT order is the transform of the inputs with index = rev2(q) (mod 4): wavefront
// q computes it as a 1024-point transform of its own (fft_big_regs<2>: two 512-point sub-transforms in registers, the step-512 stage
// lane-local), publishes its 1024 results in an LDS exchange area, and after ONE workgroup barrier every lane takes four positions
// p = 256 q + 64 i + lane and runs the last two radix-2 stages on the operands X_0[p], X_1[p], X_2[p], X_3[p] (step 1024: W_2048[p] on
// blocks (0, 1) and (2, 3); step 2048: W_4096[p] and W_4096[p + 1024]; no_simd.rs:247-279) -- which yields X[p], X[p + 1024], X[p + 2048],
// X[p + 3072].  Eight sub-transforms in ONE wavefront do not fit the register file; the LDS-staged generic kernel these sizes ran on
// before reaches 2.6 TB/s.  LDS per workgroup: 36 KiB of FFT work areas + 36 KiB staging / exchange + 4 KiB W_1024: two workgroups per CU.
constexpr int kWgPoints = 4096;

// The lane's inputs of its wavefront's 1024-point transform are the z-indices e(s, c) = 8 (lane + 64 s) + 4 c + rev2(wave): 8 bytes
// out of every 64.  Loaded like that the kernel ran at 2.8 TB/s (the same kernel with contiguous -- wrong -- indices: 5.25 TB/s,
// SYMACCEL_TUNE_WG4096_ABLATE=1), so the workgroup loads the transform's 32 KiB coalesced (16 B per lane), stages it in the exchange
// area with one complex of padding per eight -- slot(e) = e + e / 8: the gather's lane stride becomes 18 dwords, conflict-free for
// ds_read_b64 -- and every wavefront gathers its inputs from there.
#ifndef SYM_WG4096_ABLATE
#define SYM_WG4096_ABLATE 0  // measurement only: 1 = contiguous (wrong) gather indices
#endif
constexpr int kWgStage = kWgPoints + kWgPoints / 8;  // complex slots of the staging / exchange area
__device__ __forceinline__ int wg4096_input_index(int wave, int lane, int s, int c) {
#if SYM_WG4096_ABLATE & 1
    return lane + 64 * s + 512 * c + 1024 * wave;
#else
    return 8 * (lane + 64 * s) + 4 * c + (((wave & 1) << 1) | (wave >> 1));
#endif
}
__device__ __forceinline__ int wg4096_slot(int e) { return e + (e >> 3); }
// thread tid's eight 16-byte pieces of a transform's input (2 complex values each): float4 index tid + 256 j
__device__ __forceinline__ void wg4096_fetch(const float *src, int tid, float4 (&pre)[8]) {
    const float4 *s4 = reinterpret_cast<const float4 *>(src);
#pragma unroll
    for (int j = 0; j < 8; ++j) pre[j] = ld_stream(s4 + tid + 256 * j);
}
__device__ __forceinline__ void wg4096_stage(const float4 (&pre)[8], int tid, c32 *xch) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int e = 2 * (tid + 256 * j);  // (e and e + 1 share a group of eight: adjacent slots)
        xch[wg4096_slot(e)] = c32{pre[j].x, pre[j].y};
        xch[wg4096_slot(e) + 1] = c32{pre[j].z, pre[j].w};
    }
}

// publish + the last two stages.  In: x[r][B] = position 512 r + 64 B + lane of this wavefront's block.  Out: v[i][q] = X[p_i + 1024 q],
// p_i = 256 wave + 64 i + lane.  `xch` is free again after the caller's next workgroup barrier.
__device__ __forceinline__ void wg4096_finish(const c32 (&x)[2][8], int wave, int lane, c32 *xch, const c32 (&w2k)[4], const c32 (&w4a)[4],
                                              const c32 (&w4b)[4], c32 (&v)[4][4]) {
#pragma unroll
    for (int r = 0; r < 2; ++r)
#pragma unroll
        for (int B = 0; B < 8; ++B) xch[1024 * wave + 512 * r + 64 * B + lane] = x[r][B];
    wg_sync_lds();
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int p = 256 * wave + 64 * i + lane;
#pragma unroll
        for (int q = 0; q < 4; ++q) v[i][q] = xch[1024 * q + p];
        bfly(v[i][0], v[i][1], c_mul(v[i][1], w2k[i]));  // step 1024: the even-index half ...
        bfly(v[i][2], v[i][3], c_mul(v[i][3], w2k[i]));  //            ... and the odd-index half
        bfly(v[i][0], v[i][2], c_mul(v[i][2], w4a[i]));  // step 2048: X[p], X[p + 2048]
        bfly(v[i][1], v[i][3], c_mul(v[i][3], w4b[i]));  //            X[p + 1024], X[p + 3072]
        // v[i]: X[p], X[p + 1024], X[p + 2048], X[p + 3072] sit in v[i][0], v[i][1], v[i][2], v[i][3]
    }
}

__device__ __forceinline__ void wg4096_lane_twiddles(const DevTables &tb, int wave, int lane, c32 (&w2k)[4], c32 (&w4a)[4], c32 (&w4b)[4]) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int p = 256 * wave + 64 * i + lane;
        w2k[i] = ld_c(tb.fft_merge + 992 + p);          // W_2048[p]
        w4a[i] = ld_c(tb.fft_merge + 2016 + p);         // W_4096[p]
        w4b[i] = ld_c(tb.fft_merge + 2016 + 1024 + p);  // W_4096[p + 1024]
    }
}

// MODE 0: Fft, 1: Ifft (re <-> im on the way in, swap + 1 / n on the way out, no_simd.rs:160-186)
template <int MODE>
__global__ __launch_bounds__(256, 2) void fft4096_wg_kernel(DevTables tb, const float *__restrict__ in, float *__restrict__ out, size_t count,
                                                            unsigned per_wg, float c) {
    __shared__ __attribute__((aligned(16))) c32 w_merge[512];  // W_1024
    __shared__ __attribute__((aligned(16))) float wave_lds[4][kWaveLds];
    __shared__ __attribute__((aligned(16))) c32 xch[kWgStage];
    for (int i = (int)threadIdx.x; i < 512; i += 256) w_merge[i] = ld_c(tb.fft_merge + 480 + i);
    __syncthreads();
    const int lane = (int)threadIdx.x & 63, wave = (int)threadIdx.x >> 6;
    c32 *lds = reinterpret_cast<c32 *>(wave_lds[wave]);
    LaneTables lt;
    load_lane_tables(tb, lane, lt);
    c32 w2k[4], w4a[4], w4b[4];
    wg4096_lane_twiddles(tb, wave, lane, w2k, w4a, w4b);
    const size_t t0 = (size_t)blockIdx.x * per_wg;
    const size_t t1 = t0 + per_wg < count ? t0 + per_wg : count;
    float4 pre[8];
    if (t0 < t1) wg4096_fetch(in + t0 * (size_t)(2 * kWgPoints), (int)threadIdx.x, pre);
    for (size_t t = t0; t < t1; ++t) {  // (the same trip count for the four wavefronts: barriers inside)
        wg4096_stage(pre, (int)threadIdx.x, xch);
        wg_sync_lds();
        c32 x[2][8];
#pragma unroll
        for (int s = 0; s < 8; ++s)
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const c32 v = xch[wg4096_slot(wg4096_input_index(wave, lane, s, h))];
                x[h][s] = MODE ? c32{v.y, v.x} : v;
            }
        if (t + 1 < t1) wg4096_fetch(in + (t + 1) * (size_t)(2 * kWgPoints), (int)threadIdx.x, pre);  // lands during the transform
        fft_big_regs<2>(x, lane, lds, lt, w_merge);
        wg_sync_lds();  // every wavefront has gathered its inputs: the area becomes the exchange area
        c32 v[4][4];
        wg4096_finish(x, wave, lane, xch, w2k, w4a, w4b, v);
        float2 *dst = reinterpret_cast<float2 *>(out + t * (size_t)(2 * kWgPoints));
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const c32 r = v[i][q];
                dst[256 * wave + 64 * i + lane + 1024 * q] = MODE ? make_float2(c * r.y, c * r.x) : make_float2(r.x, r.y);
            }
        wg_sync_lds();  // the exchange area is free for the next transform
    }
}

// Imdct of 8192 lines (mdct.rs:67-146 with n2 = 4096): the staged "complex" e is the line pair (spec[2 e], spec[2 e + 1]); z[e] needs
// spec[2 e] and spec[8191 - 2 e] = the odd line of pair 4095 - e, whose slot moves against the lane index with the same 18-dword stride.
// The four output vectors (4096 samples each) leave through the exchange area two at a time, in natural order, 16 B per lane.
__global__ __launch_bounds__(256, 2) void imdct8192_wg_kernel(DevTables tb, const cpx *__restrict__ tw_g, const float *__restrict__ spec,
                                                              float *__restrict__ out, size_t count, unsigned per_wg) {
    constexpr int P = kWgPoints, N = 2 * P;
    __shared__ __attribute__((aligned(16))) c32 w_merge[512];  // W_1024
    __shared__ __attribute__((aligned(16))) float wave_lds[4][kWaveLds];
    __shared__ __attribute__((aligned(16))) c32 xch[kWgStage];
    __shared__ __attribute__((aligned(16))) c32 lane_tab[kLaneTabComplex];  // the FFT's lane twiddles: read at the point of use here (registers)
    fill_lane_tables_lds(tb, lane_tab, (int)threadIdx.x, 256);
    for (int i = (int)threadIdx.x; i < 512; i += 256) w_merge[i] = ld_c(tb.fft_merge + 480 + i);
    __syncthreads();
    const int lane = (int)threadIdx.x & 63, wave = (int)threadIdx.x >> 6, tid = (int)threadIdx.x;
    c32 *lds = reinterpret_cast<c32 *>(wave_lds[wave]);
    float *xf = reinterpret_cast<float *>(xch);
    const LaneTablesLds lt = lane_tables_lds(tb, lane_tab, lane);
    // pre-twiddles of this lane's sixteen inputs: the same sixteen for every transform of the walk (8 bytes out of every 64 of the
    // table: loaded once)
    c32 twp[2][8];
#pragma unroll
    for (int s = 0; s < 8; ++s)
#pragma unroll
        for (int h = 0; h < 2; ++h) twp[h][s] = ld_c(tw_g + wg4096_input_index(wave, lane, s, h));
    const size_t t0 = (size_t)blockIdx.x * per_wg;
    const size_t t1 = t0 + per_wg < count ? t0 + per_wg : count;
    float4 pre[8];
    if (t0 < t1) wg4096_fetch(spec + t0 * (size_t)N, tid, pre);
    for (size_t t = t0; t < t1; ++t) {
        wg4096_stage(pre, tid, xch);
        wg_sync_lds();
        c32 x[2][8];
#pragma unroll
        for (int s = 0; s < 8; ++s)
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int e = wg4096_input_index(wave, lane, s, h);
                x[h][s] = pre_twiddle(xf[2 * wg4096_slot(e)], xf[2 * wg4096_slot(P - 1 - e) + 1], twp[h][s]);
            }
        if (t + 1 < t1) wg4096_fetch(spec + (t + 1) * (size_t)N, tid, pre);  // lands during the transform
        fft_big_regs<2>(x, lane, lds, lt, w_merge);
        wg_sync_lds();  // every wavefront has gathered its inputs: the area becomes the exchange area
        c32 v[4][4];
        {
            c32 w2k[4], w4a[4], w4b[4];
            wg4096_lane_twiddles(tb, wave, lane, w2k, w4a, w4b);  // (coalesced, L2-resident: not kept across the walk -- registers)
            wg4096_finish(x, wave, lane, xch, w2k, w4a, w4b, v);
        }
        // post-twiddle (mdct.rs:94-137, n2 = 4096, n4 = 2048) in place
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int q = 0; q < 4; ++q) v[i][q] = post_twiddle(v[i][q], ld_c(tw_g + 256 * wave + 64 * i + lane + 1024 * q));
        float *o = out + t * (size_t)(2 * N);
#pragma unroll
        for (int half = 0; half < 2; ++half) {  // vec0 | vec1, then vec2 | vec3: 8192 floats through the exchange area
            wg_sync_lds();  // the exchange operands (first half) / the previous pair's reads (second half) are done
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int k = 256 * wave + 64 * i + lane + 1024 * q;
                    const c32 val = v[i][q];
                    // (q < 2 <=> k < n4: a compile-time branch)
                    if (q < 2) {
                        const int fi = 2 * k, ri = P - 1 - 2 * k;
                        xf[ri] = half == 0 ? -val.y : val.x;      // vec0[ri] / vec2[ri]
                        xf[P + fi] = half == 0 ? val.y : val.x;   // vec1[fi] / vec3[fi]
                    } else {
                        const int i2 = k - P / 2;
                        const int fi = 2 * i2, ri = P - 1 - 2 * i2;
                        xf[fi] = half == 0 ? -val.x : val.y;      // vec0[fi] / vec2[fi]
                        xf[P + ri] = half == 0 ? val.x : val.y;   // vec1[ri] / vec3[ri]
                    }
                }
            wg_sync_lds();
            float4 *o4 = reinterpret_cast<float4 *>(o + (size_t)half * (2 * P));
#pragma unroll
            for (int j = 0; j < 8; ++j) st_stream(o4 + tid + 256 * j, reinterpret_cast<const float4 *>(xf)[tid + 256 * j]);
        }
        wg_sync_lds();  // the area is free for the next transform's staging
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

int launch_fft4096_wg(symaccel_ctx *ctx, const float *d_in, float *d_out, size_t count, bool inverse) {
    // two workgroups per CU resident; a workgroup walks `per_wg` transforms
    size_t per_wg = count / ((size_t)ctx->n_cus * 2 * 4);
    per_wg = per_wg < 1 ? 1 : (per_wg > 16 ? 16 : per_wg);
    const size_t grid = (count + per_wg - 1) / per_wg;
    if (grid > 0x7fffffffu) return SYMACCEL_ERR_INVALID_ARG;
    if (inverse)
        hipLaunchKernelGGL(fft4096_wg_kernel<1>, dim3((unsigned)grid), dim3(256), 0, ctx->stream, ctx->dev, d_in, d_out, count, (unsigned)per_wg,
                           1.0f / (float)kWgPoints);
    else
        hipLaunchKernelGGL(fft4096_wg_kernel<0>, dim3((unsigned)grid), dim3(256), 0, ctx->stream, ctx->dev, d_in, d_out, count, (unsigned)per_wg,
                           1.0f / (float)kWgPoints);
    SYM_GPU(ctx, hipGetLastError());
    return SYMACCEL_OK;
}

int launch_imdct8192_wg(symaccel_ctx *ctx, const cpx *d_twiddle, const float *d_spec, float *d_out, size_t count) {
    size_t per_wg = count / ((size_t)ctx->n_cus * 2 * 4);
    per_wg = per_wg < 1 ? 1 : (per_wg > 16 ? 16 : per_wg);
    const size_t grid = (count + per_wg - 1) / per_wg;
    if (grid > 0x7fffffffu) return SYMACCEL_ERR_INVALID_ARG;
    hipLaunchKernelGGL(imdct8192_wg_kernel, dim3((unsigned)grid), dim3(256), 0, ctx->stream, ctx->dev, d_twiddle, d_spec, d_out, count,
                       (unsigned)per_wg);
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
