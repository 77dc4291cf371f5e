// MP3 Layer III joint stereo, the stage between requantisation and the synthesis tail:
//   symphonia-bundle-mp3/src/layer3/stereo.rs:485-556 (stereo), 196-260 (process_intensity_long_block),
//   264-483 (process_intensity_short_block), 139-186 (process_mid_side / process_intensity), 31-118 (ratio tables).
//
// One wavefront per granule (both channels, 2 x 576 lines in registers as groups of four lines, 16-byte accesses).
// Which bands are intensity coded depends on the data: the reference walks the scale-factor bands of channel 1 from
// the top while they are all zero (per window for short blocks).  Here every lane flags the band of each non-zero
// line it holds in a 40-bit mask (two LDS atomic ORs per lane), then every lane runs the reference's band walk on that
// mask -- scalar work on wave-uniform values, no memory -- which yields the intensity bound and two band masks
// (intensity coded / mid-side coded); finally every lane applies its lines' bands' action, or mid/side below the bound.
//
// FUSED: the wavefront first requantises both channels from the quantised Huffman samples (mp3_requant.h: the
// arithmetic of mp3_requant.hip) into those registers, so the f32 spectra make no round trip through HBM between
// the two stages: 2 B in + 4 B out per line instead of 2 + 4 + 4 + 4.
#include <hip/hip_runtime.h>

#include "mp3_requant.h"

namespace symaccel {

namespace {

__device__ __forceinline__ void wave_sync() {  // order this wavefront's own LDS traffic (no workgroup barrier needed)
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

constexpr int kWaves = 4;

template <bool FUSED>
__global__ __launch_bounds__(64 * kWaves) void mp3_stereo_kernel(DevTables tb, float *__restrict__ xr, unsigned granules_per_chain,
                                                                 const int32_t *__restrict__ pair_chains,
                                                                 const symaccel_mp3_stereo *__restrict__ desc, int sr, SfbEdges e,
                                                                 unsigned n_items, const int16_t *__restrict__ quant,
                                                                 const symaccel_mp3_requant *__restrict__ rq_desc) {
    __shared__ unsigned nz_all[kWaves][2];  // bit mask of the bands in which channel 1 is non-zero
    __shared__ int act_all[kWaves][40];     // per band: what the plan says (mp3_stereo_expand)
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
    unsigned *nzw = nz_all[wave];
    int *act = act_all[wave];
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
    if (FUSED) {
        // read_huffman_samples' values + requantize for both channels (requantize.rs:117-147, 239-380)
        const symaccel_mp3_requant &r0 = rq_desc[gc0], &r1 = rq_desc[gc1];
        // the quantised samples only need the granule's address: requested before the scale factors are turned into
        // scales (descriptor fields -> table look-up -> LDS), so that the two trips to HBM overlap
        short4 s0[kQ], s1[kQ];
#pragma unroll
        for (int qq = 0; qq < kQ; ++qq) {
            const int grp = have[qq] ? lane + 64 * qq : 0;
            s0[qq] = reinterpret_cast<const short4 *>(quant + gc0 * 576)[grp];
            s1[qq] = reinterpret_cast<const short4 *>(quant + gc1 * 576)[grp];
        }
        float(*scale)[kMp3Slots] = scale_all[FUSED ? wave : 0];
        if (lane < kMp3Slots) {
            scale[0][lane] = mp3_slot_scale(tb, r0, lane, e.mixed_switch);
            scale[1][lane] = mp3_slot_scale(tb, r1, lane, e.mixed_switch);
        }
        wave_sync();
        const uint8_t *m0 = tb.mp3_band_map + (size_t)(sr * 4 + mp3_requant_kind(r0)) * 576;
        const uint8_t *m1 = tb.mp3_band_map + (size_t)(sr * 4 + mp3_requant_kind(r1)) * 576;
        const int rq0 = r0.rzero > 576 ? 576 : (int)r0.rzero, rq1 = r1.rzero > 576 ? 576 : (int)r1.rzero;
        // every load of the granule first, the arithmetic after: the sample mapping below has a (rare) global table read
        // behind a branch per sample, and a load placed after such a branch is not issued before it -- as one loop this
        // was three dependent trips to HBM per granule instead of one
        uchar4 k0[kQ], k1[kQ], kb[kQ];
#pragma unroll
        for (int qq = 0; qq < kQ; ++qq) {
            const int grp = have[qq] ? lane + 64 * qq : 0;
            k0[qq] = reinterpret_cast<const uchar4 *>(m0)[grp];
            k1[qq] = reinterpret_cast<const uchar4 *>(m1)[grp];
            kb[qq] = reinterpret_cast<const uchar4 *>(map)[grp];
        }
#pragma unroll
        for (int qq = 0; qq < kQ; ++qq) {
            if (!have[qq]) continue;
            const int grp = lane + 64 * qq;
            const int v0[4] = {s0[qq].x, s0[qq].y, s0[qq].z, s0[qq].w}, v1[4] = {s1[qq].x, s1[qq].y, s1[qq].z, s1[qq].w};
            const int i0[4] = {k0[qq].x, k0[qq].y, k0[qq].z, k0[qq].w}, i1[4] = {k1[qq].x, k1[qq].y, k1[qq].z, k1[qq].w};
            band[4 * qq] = kb[qq].x; band[4 * qq + 1] = kb[qq].y; band[4 * qq + 2] = kb[qq].z; band[4 * qq + 3] = kb[qq].w;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int line = 4 * grp + j;
                a[4 * qq + j] = mp3_sample_value(tb, pow43_lo, v0[j], line >= rq0) * scale[0][i0[j]];
                b[4 * qq + j] = mp3_sample_value(tb, pow43_lo, v1[j], line >= rq1) * scale[1][i1[j]];
            }
        }
    } else {
        float4 x0[kQ], x1[kQ];
        uchar4 kb[kQ];
#pragma unroll
        for (int qq = 0; qq < kQ; ++qq) {  // (all the loads before the first use, as above)
            const int grp = have[qq] ? lane + 64 * qq : 0;
            x0[qq] = reinterpret_cast<const float4 *>(ch0)[grp];
            x1[qq] = reinterpret_cast<const float4 *>(ch1)[grp];
            kb[qq] = reinterpret_cast<const uchar4 *>(map)[grp];
        }
#pragma unroll
        for (int qq = 0; qq < kQ; ++qq) {
            if (!have[qq]) continue;
            a[4 * qq] = x0[qq].x; a[4 * qq + 1] = x0[qq].y; a[4 * qq + 2] = x0[qq].z; a[4 * qq + 3] = x0[qq].w;
            b[4 * qq] = x1[qq].x; b[4 * qq + 1] = x1[qq].y; b[4 * qq + 2] = x1[qq].z; b[4 * qq + 3] = x1[qq].w;
            band[4 * qq] = kb[qq].x; band[4 * qq + 1] = kb[qq].y; band[4 * qq + 2] = kb[qq].z; band[4 * qq + 3] = kb[qq].w;
        }
    }

    Mp3StereoPlan plan{end, 0ull, 0ull};
    if (intensity) {
        if (lane < 2) nzw[lane] = 0u;
        wave_sync();
        unsigned long long mine = 0ull;  // is_zero_band (stereo.rs:189-192): one bit per band
#pragma unroll
        for (int i = 0; i < 4 * kQ; ++i)
            if (have[i / 4] && b[i] != 0.0f) mine |= 1ull << band[i];
        if ((unsigned)mine) atomicOr(&nzw[0], (unsigned)mine);
        if ((unsigned)(mine >> 32)) atomicOr(&nzw[1], (unsigned)(mine >> 32));
        wave_sync();
        // ---- the band walk: scalar work on the 40-bit mask, identical in every lane
        const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)nzw[0]), hi = (unsigned)__builtin_amdgcn_readfirstlane((int)nzw[1]);
        plan = mp3_stereo_walk(d, e, (unsigned long long)lo | ((unsigned long long)hi << 32), end, rzero1);
        if (lane < 40) mp3_stereo_expand(plan, d, tb.mp3_is_ratios, lane, act, kl, kr);
        wave_sync();
    }

    // ---- apply
#pragma unroll
    for (int qq = 0; qq < kQ; ++qq) {
        if (!have[qq]) continue;
        const int grp = lane + 64 * qq;
        bool touched = FUSED;  // fused: xr is this kernel's output, untouched lines are stored too
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int i = 4 * qq + j;
            touched |= mp3_stereo_apply(a[i], b[i], 4 * grp + j, plan.bound, mid_side, intensity, band[i], act, kl, kr);
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
    const SfbEdges e = make_sfb_edges(host_tables(), sr);
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
