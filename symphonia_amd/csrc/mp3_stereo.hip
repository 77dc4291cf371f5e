// MP3 Layer III joint stereo, the stage between requantisation and the synthesis tail:
//   symphonia-bundle-mp3/src/layer3/stereo.rs:485-556 (stereo), 196-260 (process_intensity_long_block),
//   264-483 (process_intensity_short_block), 139-186 (process_mid_side / process_intensity), 31-118 (ratio tables).
//
// One wavefront per granule (both channels, 2 x 576 lines in registers, nine lines per lane and channel, coalesced).
// Which bands are intensity coded depends on the data: the reference walks the scale-factor bands of channel 1 from
// the top while they are all zero (per window for short blocks).  Here every lane flags the band of each non-zero
// line it holds in LDS, then the wavefront runs the reference's band walk on those <= 39 flags (wave-uniform scalar
// work) and leaves one action per band -- none, mid/side, or intensity with its (left, right) ratios -- plus the
// intensity bound in LDS; finally every lane applies the action of its lines' bands, or mid/side below the bound.
#include <hip/hip_runtime.h>

#include "symaccel_internal.h"

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

__global__ __launch_bounds__(64 * kWaves) void mp3_stereo_kernel(DevTables tb, float *__restrict__ xr, unsigned granules_per_chain,
                                                                 const int32_t *__restrict__ pair_chains,
                                                                 const symaccel_mp3_stereo *__restrict__ desc, int sr, SfbEdges e,
                                                                 unsigned n_items) {
    __shared__ int nz_all[kWaves][40], act_all[kWaves][40];
    __shared__ float kl_all[kWaves][40], kr_all[kWaves][40];
    const int lane = (int)threadIdx.x & 63, wave = (int)threadIdx.x >> 6;
    const unsigned item = blockIdx.x * kWaves + (unsigned)wave;
    if (item >= n_items) return;
    int *nz = nz_all[wave], *act = act_all[wave];
    float *kl = kl_all[wave], *kr = kr_all[wave];
    const unsigned pair = item / granules_per_chain, g = item % granules_per_chain;
    const symaccel_mp3_stereo &d = desc[item];
    const bool mid_side = d.flags & SYMACCEL_MP3_ST_MID_SIDE, intensity = d.flags & SYMACCEL_MP3_ST_INTENSITY;
    if (!mid_side && !intensity) return;  // stereo.rs:491-500: not joint stereo
    float *ch0 = xr + ((size_t)pair_chains[2 * pair] * granules_per_chain + g) * 576;
    float *ch1 = xr + ((size_t)pair_chains[2 * pair + 1] * granules_per_chain + g) * 576;
    const int rzero1 = d.rzero1 > 576 ? 576 : (int)d.rzero1;
    int end = d.rzero0 > d.rzero1 ? d.rzero0 : d.rzero1;  // stereo.rs:522
    end = end > 576 ? 576 : end;
    const bool is_short = d.block_type == SYMACCEL_MP3_SHORT, is_mixed = is_short && d.is_mixed;
    const uint8_t *map = tb.mp3_band_map + (size_t)(sr * 4 + (is_short ? (is_mixed ? 3 : 1) : 0)) * 576;

    float a[9], b[9];
    int band[9];
#pragma unroll
    for (int q = 0; q < 9; ++q) {
        a[q] = ch0[lane + 64 * q];
        b[q] = ch1[lane + 64 * q];
        band[q] = map[lane + 64 * q];
    }

    int bound = end;
    if (intensity) {
        if (lane < 40) {
            nz[lane] = 0;
            act[lane] = kNone;
        }
        wave_sync();
#pragma unroll
        for (int q = 0; q < 9; ++q)
            if (b[q] != 0.0f) nz[band[q]] = 1;  // is_zero_band (stereo.rs:189-192), one flag per band
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
    for (int q = 0; q < 9; ++q) {
        const int line = lane + 64 * q;
        int action = kNone;
        if (line < bound)
            action = mid_side ? kMidSide : kNone;
        else if (intensity)
            action = act[band[q]];
        if (action == kMidSide) {  // process_mid_side (stereo.rs:139-148)
            const float left = (a[q] + b[q]) * kFrac1Sqrt2, right = (a[q] - b[q]) * kFrac1Sqrt2;
            ch0[line] = left;
            ch1[line] = right;
        } else if (action == kIntensity) {
            ch0[line] = kl[band[q]] * a[q];
            ch1[line] = kr[band[q]] * a[q];
        }
    }
}

}  // namespace

int launch_mp3_stereo(symaccel_ctx *ctx, float *d_xr, size_t granules_per_chain, const int32_t *d_pair_chains,
                      const symaccel_mp3_stereo *d_desc, int sr, size_t n_pairs) {
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
    hipLaunchKernelGGL(mp3_stereo_kernel, dim3((unsigned)grid), dim3(64 * kWaves), 0, ctx->stream, ctx->dev, d_xr,
                       (unsigned)granules_per_chain, d_pair_chains, d_desc, sr, e, (unsigned)items);
    SYM_GPU(ctx, hipGetLastError());
    return SYMACCEL_OK;
}

}  // namespace symaccel
