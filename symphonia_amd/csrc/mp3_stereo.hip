// MP3 Layer III joint stereo, the stage between requantisation and the synthesis tail:
//   symphonia-bundle-mp3/src/layer3/stereo.rs:485-556 (stereo), 196-260 (process_intensity_long_block),
//   264-483 (process_intensity_short_block), 139-186 (process_mid_side / process_intensity), 31-118 (ratio tables).
//
// One wavefront per granule (both channels, 2 x 576 lines in registers as groups of four lines, 16-byte accesses).
// Which bands are intensity coded depends on the data: the reference walks the scale-factor bands of channel 1 from
// the top while they are all zero (per window for short blocks).  Here every lane flags the band of each non-zero
// line it holds in LDS, then the wavefront runs the reference's band walk on those <= 39 flags (wave-uniform scalar
// work) and leaves one action per band -- none, mid/side, or intensity with its (left, right) ratios -- plus the
// intensity bound in LDS; finally every lane applies the action of its lines' bands, or mid/side below the bound.
//
// FUSED: the wavefront first requantises both channels from the quantised Huffman samples (mp3_requant.h: the
// arithmetic of mp3_requant.hip) into those registers, so the f32 spectra make no round trip through HBM between
// the two stages: 2 B in + 4 B out per line instead of 2 + 4 + 4 + 4.
#include <hip/hip_runtime.h>

#include "mp3_requant.h"

namespace symaccel {

namespace {

struct SfbEdges {  // band edge tables of one sample rate (layer3/common.rs:9-172), by value
    int16_t longb[23], shortb[40], mixed[40];
    int16_t mixed_len, mixed_switch;
};

__device__ __forceinline__ void wave_sync() {  // order this wavefront's own LDS traffic (no workgroup barrier needed)
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

constexpr int kNone = 0, kMidSide = 1, kIntensity = 2;
constexpr int kWaves = 4;

template <bool FUSED>
__global__ __launch_bounds__(64 * kWaves) void mp3_stereo_kernel(DevTables tb, float *__restrict__ xr, unsigned granules_per_chain,
                                                                 const int32_t *__restrict__ pair_chains,
                                                                 const symaccel_mp3_stereo *__restrict__ desc, int sr, SfbEdges e,
                                                                 unsigned n_items, const int16_t *__restrict__ quant,
                                                                 const symaccel_mp3_requant *__restrict__ rq_desc) {
    __shared__ int nz_all[kWaves][40], act_all[kWaves][40];
    __shared__ float kl_all[kWaves][40], kr_all[kWaves][40];
    __shared__ float scale_all[FUSED ? kWaves : 1][2][kMp3Slots];
    __shared__ float pow43_lo[FUSED ? kMp3PowLds : 1];
    if (FUSED) {
        for (int k = (int)threadIdx.x; k < kMp3PowLds; k += 64 * kWaves) pow43_lo[k] = tb.mp3_pow43[k];
        __syncthreads();  // the only workgroup-wide barrier, before any wavefront leaves
    }
    const int lane = (int)threadIdx.x & 63, wave = (int)threadIdx.x >> 6;
    const unsigned item = blockIdx.x * kWaves + (unsigned)wave;
    if (item >= n_items) return;
    int *nz = nz_all[wave], *act = act_all[wave];
    float *kl = kl_all[wave], *kr = kr_all[wave];
    const unsigned pair = item / granules_per_chain, g = item % granules_per_chain;
    const symaccel_mp3_stereo &d = desc[item];
    const bool mid_side = d.flags & SYMACCEL_MP3_ST_MID_SIDE, intensity = d.flags & SYMACCEL_MP3_ST_INTENSITY;
    if (!FUSED && !mid_side && !intensity) return;  // stereo.rs:491-500: not joint stereo
    const size_t gc0 = (size_t)pair_chains[2 * pair] * granules_per_chain + g, gc1 = (size_t)pair_chains[2 * pair + 1] * granules_per_chain + g;
    float *ch0 = xr + gc0 * 576, *ch1 = xr + gc1 * 576;
    const int rzero1 = d.rzero1 > 576 ? 576 : (int)d.rzero1;
    int end = d.rzero0 > d.rzero1 ? d.rzero0 : d.rzero1;  // stereo.rs:522
    end = end > 576 ? 576 : end;
    const bool is_short = d.block_type == SYMACCEL_MP3_SHORT, is_mixed = is_short && d.is_mixed;
    const uint8_t *map = tb.mp3_band_map + (size_t)(sr * 4 + (is_short ? (is_mixed ? 3 : 1) : 0)) * 576;

    // The granule as 144 groups of four lines: lane l holds groups l, l + 64 and (l < 16) l + 128 -- 16-byte loads and
    // stores of the spectra, 8-byte loads of the quantised samples, 4-byte loads of the band maps.
    constexpr int kQ = 3;
    float a[4 * kQ], b[4 * kQ];
    int band[4 * kQ];
    bool have[kQ];
#pragma unroll
    for (int qq = 0; qq < kQ; ++qq) have[qq] = lane + 64 * qq < 144;
#pragma unroll
    for (int i = 0; i < 4 * kQ; ++i) {
        a[i] = b[i] = 0.0f;
        band[i] = 0;
    }
    auto unpack_map = [&](const uint8_t *m, int grp, int *dst) {
        const uchar4 v = reinterpret_cast<const uchar4 *>(m)[grp];
        dst[0] = v.x; dst[1] = v.y; dst[2] = v.z; dst[3] = v.w;
    };
    if (FUSED) {
        // read_huffman_samples' values + requantize for both channels (requantize.rs:117-147, 239-380)
        const symaccel_mp3_requant &r0 = rq_desc[gc0], &r1 = rq_desc[gc1];
        float(*scale)[kMp3Slots] = scale_all[FUSED ? wave : 0];
        if (lane < kMp3Slots) {
            scale[0][lane] = mp3_slot_scale(tb, r0, lane, e.mixed_switch);
            scale[1][lane] = mp3_slot_scale(tb, r1, lane, e.mixed_switch);
        }
        wave_sync();
        const uint8_t *m0 = tb.mp3_band_map + (size_t)(sr * 4 + mp3_requant_kind(r0)) * 576;
        const uint8_t *m1 = tb.mp3_band_map + (size_t)(sr * 4 + mp3_requant_kind(r1)) * 576;
        const int rq0 = r0.rzero > 576 ? 576 : (int)r0.rzero, rq1 = r1.rzero > 576 ? 576 : (int)r1.rzero;
#pragma unroll
        for (int qq = 0; qq < kQ; ++qq) {
            if (!have[qq]) continue;
            const int grp = lane + 64 * qq;
            const short4 s0 = reinterpret_cast<const short4 *>(quant + gc0 * 576)[grp];
            const short4 s1 = reinterpret_cast<const short4 *>(quant + gc1 * 576)[grp];
            const int v0[4] = {s0.x, s0.y, s0.z, s0.w}, v1[4] = {s1.x, s1.y, s1.z, s1.w};
            int k0[4], k1[4];
            unpack_map(m0, grp, k0);
            unpack_map(m1, grp, k1);
            unpack_map(map, grp, band + 4 * qq);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int line = 4 * grp + j;
                a[4 * qq + j] = mp3_sample_value(tb, pow43_lo, v0[j], line >= rq0) * scale[0][k0[j]];
                b[4 * qq + j] = mp3_sample_value(tb, pow43_lo, v1[j], line >= rq1) * scale[1][k1[j]];
            }
        }
    } else {
#pragma unroll
        for (int qq = 0; qq < kQ; ++qq) {
            if (!have[qq]) continue;
            const int grp = lane + 64 * qq;
            const float4 x0 = reinterpret_cast<const float4 *>(ch0)[grp], x1 = reinterpret_cast<const float4 *>(ch1)[grp];
            a[4 * qq] = x0.x; a[4 * qq + 1] = x0.y; a[4 * qq + 2] = x0.z; a[4 * qq + 3] = x0.w;
            b[4 * qq] = x1.x; b[4 * qq + 1] = x1.y; b[4 * qq + 2] = x1.z; b[4 * qq + 3] = x1.w;
            unpack_map(map, grp, band + 4 * qq);
        }
    }

    int bound = end;
    if (intensity) {
        if (lane < 40) {
            nz[lane] = 0;
            act[lane] = kNone;
        }
        wave_sync();
#pragma unroll
        for (int i = 0; i < 4 * kQ; ++i)
            if (have[i / 4] && b[i] != 0.0f) nz[band[i]] = 1;  // is_zero_band (stereo.rs:189-192), one flag per band
        wave_sync();
        // ---- the band walk (identical in every lane; lane 0 records the decisions)
        const int table = (d.flags & SYMACCEL_MP3_ST_MPEG1) ? 0 : 7 + 32 * ((d.flags & SYMACCEL_MP3_ST_IS_SCALE) ? 1 : 0);
        const int inv_pos = (d.flags & SYMACCEL_MP3_ST_MPEG1) ? 7 : 31;  // INTENSITY_INV_POS_* (stereo.rs:19-29)
        auto zero_band = [&](int k, int is_pos) {  // process_intensity (stereo.rs:165-186) as an action for band k
            if (lane != 0) return;
            if (is_pos < inv_pos) {
                act[k] = kIntensity;
                kl[k] = tb.mp3_is_ratios[2 * (table + is_pos)];
                kr[k] = tb.mp3_is_ratios[2 * (table + is_pos) + 1];
            } else {
                act[k] = mid_side ? kMidSide : kNone;
            }
        };
        if (!is_short) {
            // process_intensity_long_block (stereo.rs:196-260); is_pos[21] = is_pos[20] (:226-228)
            for (int i = 21; i >= 0; --i) {
                const int start = e.longb[i];
                if (!(start >= rzero1 || nz[i] == 0)) break;
                zero_band(i, d.scalefacs1[i < 21 ? i : 20]);
                bound = start;
            }
        } else {
            // process_intensity_short_block (stereo.rs:264-483).  Band k of the edge table uses is_pos[k], where
            // is_pos[..36] = scalefacs[..36] and is_pos[36..39] = scalefacs[33..36] (:369-371).
            const int16_t *bands = is_mixed ? e.mixed : e.shortb;
            const int n_edges = is_mixed ? e.mixed_len : 40, sw = is_mixed ? e.mixed_switch : 0;
            const int n_groups = (n_edges - sw - 3 + 2) / 3;  // groups of three windows (stereo.rs:379-386)
            bool wz0 = true, wz1 = true, wz2 = true, found_bound = false;
            for (int gi = n_groups - 1; gi >= 0; --gi) {
                const int k0 = sw + 3 * gi;  // bands k0, k0 + 1, k0 + 2 = windows 0, 1, 2
#pragma unroll
                for (int w = 2; w >= 0; --w) {
                    const int k = k0 + w;
                    bool &wz = w == 2 ? wz2 : (w == 1 ? wz1 : wz0);
                    wz = wz && nz[k] == 0;
                    if (wz)
                        zero_band(k, d.scalefacs1[k < 36 ? k : k - 3]);
                    else if (mid_side && lane == 0)
                        act[k] = kMidSide;
                }
                bound = bands[k0];
                found_bound = !wz0 && !wz1 && !wz2;
                if (found_bound) break;
            }
            if (!found_bound && is_mixed) {  // the long bands of a mixed block, stereo.rs:450-478
                for (int i = sw - 1; i >= 0; --i) {
                    if (nz[i] != 0) break;
                    zero_band(i, d.scalefacs1[i]);
                    bound = bands[i];
                }
            }
        }
        wave_sync();
    }

    // ---- apply: mid/side below the intensity bound (stereo.rs:541-543), the band's action from it on
    constexpr float kFrac1Sqrt2 = 0.70710678118654752440f;  // f32::consts::FRAC_1_SQRT_2
#pragma unroll
    for (int qq = 0; qq < kQ; ++qq) {
        if (!have[qq]) continue;
        const int grp = lane + 64 * qq;
        bool touched = FUSED;  // fused: xr is this kernel's output, untouched lines are stored too
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int i = 4 * qq + j, line = 4 * grp + j;
            int action = kNone;
            if (line < bound)
                action = mid_side ? kMidSide : kNone;
            else if (intensity)
                action = act[band[i]];
            if (action == kMidSide) {  // process_mid_side (stereo.rs:139-148)
                const float left = (a[i] + b[i]) * kFrac1Sqrt2, right = (a[i] - b[i]) * kFrac1Sqrt2;
                a[i] = left;
                b[i] = right;
                touched = true;
            } else if (action == kIntensity) {  // process_intensity (stereo.rs:174-180)
                const float is = a[i];
                a[i] = kl[band[i]] * is;
                b[i] = kr[band[i]] * is;
                touched = true;
            }
        }
        if (touched) {
            reinterpret_cast<float4 *>(ch0)[grp] = make_float4(a[4 * qq], a[4 * qq + 1], a[4 * qq + 2], a[4 * qq + 3]);
            reinterpret_cast<float4 *>(ch1)[grp] = make_float4(b[4 * qq], b[4 * qq + 1], b[4 * qq + 2], b[4 * qq + 3]);
        }
    }
}

}  // namespace

int launch_mp3_stereo(symaccel_ctx *ctx, float *d_xr, size_t granules_per_chain, const int32_t *d_pair_chains,
                      const symaccel_mp3_stereo *d_desc, int sr, size_t n_pairs, const int16_t *d_quant,
                      const symaccel_mp3_requant *d_rq_desc) {
    const size_t items = n_pairs * granules_per_chain, grid = (items + kWaves - 1) / kWaves;
    if (items > 0xffffffffu || granules_per_chain > 0xffffffffu || grid > 0x7fffffffu) return SYMACCEL_ERR_INVALID_ARG;
    const HostTables &t = host_tables();
    SfbEdges e;
    for (int i = 0; i < 23; ++i) e.longb[i] = (int16_t)t.mp3_sfb_long[sr][i];
    for (int i = 0; i < 40; ++i) {
        e.shortb[i] = (int16_t)t.mp3_sfb_short[sr][i];
        e.mixed[i] = (int16_t)t.mp3_sfb_mixed[sr][i];
    }
    e.mixed_len = (int16_t)t.mp3_sfb_mixed_len[sr];
    e.mixed_switch = (int16_t)t.mp3_sfb_switch[sr];
    if (d_quant)
        hipLaunchKernelGGL(mp3_stereo_kernel<true>, dim3((unsigned)grid), dim3(64 * kWaves), 0, ctx->stream, ctx->dev, d_xr,
                           (unsigned)granules_per_chain, d_pair_chains, d_desc, sr, e, (unsigned)items, d_quant, d_rq_desc);
    else
        hipLaunchKernelGGL(mp3_stereo_kernel<false>, dim3((unsigned)grid), dim3(64 * kWaves), 0, ctx->stream, ctx->dev, d_xr,
                           (unsigned)granules_per_chain, d_pair_chains, d_desc, sr, e, (unsigned)items, d_quant, d_rq_desc);
    SYM_GPU(ctx, hipGetLastError());
    return SYMACCEL_OK;
}

}  // namespace symaccel
