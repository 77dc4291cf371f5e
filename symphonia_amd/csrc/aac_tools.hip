// AAC spectral tools between the spectrum decoder and Dsp::synth:
//   joint-stereo decoding of a channel pair   symphonia-codec-aac/src/aac/cpe.rs:110-157
//   the filtering loops of Tns::synth         symphonia-codec-aac/src/aac/ics/tns.rs:180-195
// Both work in place on the chain-major coefficient array aac_synth_kernel consumes.
//
// Joint stereo is a streaming pass: one workgroup per channel-pair frame, one thread per four lines (the swb offsets
// of every AAC sample rate are multiples of four, so a float4 never straddles a band); only bands that are coded
// mid/side or intensity are touched, so the traffic is proportional to their share.
// TNS is an all-pole filter running along the spectrum -- a serial recurrence of up to 20 taps per line.  The
// parallel axis is the filter: one lane per filter of a flat list, the tap history in registers, sixteen lines
// (one 64-byte sector) per lane and round, the next sixteen fetched while the current ones are filtered.  The taps are
// applied in the reference's order, each as a rounded multiply and a rounded subtract, and only `min(order, lines
// filtered so far)` of them (tns.rs:184, 191).  (Staging the lines through an LDS tile of 64 filters x 32 lines, as the
// integer predictors do -- 128-byte row segments on the HBM side -- was built and measured twice, before and after the
// tap loop became compile-time: 0.50 against 0.38 ms, then 0.33 against 0.30 ms for 131 072 order-12 filters.  Not kept;
// profiles/r02n_tns_alac_ab.txt.)
#include <hip/hip_runtime.h>

#include "symaccel_internal.h"

namespace symaccel {

namespace {

__global__ __launch_bounds__(256) void aac_joint_stereo_kernel(AacBandMaps maps, float *__restrict__ coeffs,
                                                               unsigned frames_per_chain, const int32_t *__restrict__ pair_chains,
                                                               const symaccel_aac_js_frame *__restrict__ desc) {
    const unsigned pf = blockIdx.x, pair = pf / frames_per_chain, f = pf % frames_per_chain;
    const symaccel_aac_js_frame &d = desc[pf];
    const int t = (int)threadIdx.x;
    int sfb, slot;
    if (d.num_windows == 1) {
        sfb = maps.long4[t];
        slot = sfb;
    } else {
        sfb = maps.short4[t & 31];
        slot = (t >> 5) * 16 + sfb;
    }
    if (sfb >= (int)d.max_sfb || slot >= 128) return;  // (slot < 128 always holds for valid descriptors)
    const int mode = d.mode[slot];
    if (mode != SYMACCEL_AAC_JS_MS && mode != SYMACCEL_AAC_JS_INTENSITY) return;
    float4 *lp = reinterpret_cast<float4 *>(coeffs + ((size_t)pair_chains[2 * pair] * frames_per_chain + f) * 1024) + t;
    float4 *rp = reinterpret_cast<float4 *>(coeffs + ((size_t)pair_chains[2 * pair + 1] * frames_per_chain + f) * 1024) + t;
    const float4 l = *lp;
    if (mode == SYMACCEL_AAC_JS_INTENSITY) {  // cpe.rs:124-139: right = scale * left
        const float s = d.scale[slot];
        *rp = make_float4(s * l.x, s * l.y, s * l.z, s * l.w);
    } else {  // cpe.rs:144-154: (m, s) -> (m + s, m - s)
        const float4 r = *rp;
        *lp = make_float4(l.x + r.x, l.y + r.y, l.z + r.z, l.w + r.w);
        *rp = make_float4(l.x - r.x, l.y - r.y, l.z - r.z, l.w - r.w);
    }
}

constexpr int kTnsMaxOrder = 20;  // TNS_MAX_ORDER, tns.rs:22
constexpr int kTnsGroup = 16;     // lines per lane and round

// One lane's filter: where it walks and what it multiplies by.
struct TnsLane {
    float *x;        // first line filtered (start, or end - 1 for a downward filter)
    int len, order;  // lines in the range; taps
    bool down, aligned;
};

// Sixteen lines (64 bytes) per lane and round: the filter walks its range in groups of sixteen -- four 16-byte loads and
// stores when the range is 16-byte aligned, as every range built from swb offsets is; scalar accesses otherwise and for a
// ragged last group -- so every 64-byte sector a lane touches crosses the L2 -> L1 path once.
__device__ __forceinline__ void tns_fetch(const TnsLane &L, int m0, float (&v)[kTnsGroup]) {
    if (m0 >= L.len) return;
    if (L.aligned && m0 + kTnsGroup <= L.len) {
#pragma unroll
        for (int q = 0; q < kTnsGroup / 4; ++q) {
            const float4 f4 = *reinterpret_cast<const float4 *>(L.down ? L.x - m0 - 4 * q - 3 : L.x + m0 + 4 * q);
            v[4 * q + 0] = L.down ? f4.w : f4.x;
            v[4 * q + 1] = L.down ? f4.z : f4.y;
            v[4 * q + 2] = L.down ? f4.y : f4.z;
            v[4 * q + 3] = L.down ? f4.x : f4.w;
        }
    } else {
#pragma unroll
        for (int k = 0; k < kTnsGroup; ++k)
            if (m0 + k < L.len) v[k] = L.x[L.down ? -(long)(m0 + k) : (long)(m0 + k)];
    }
}
__device__ __forceinline__ void tns_store(const TnsLane &L, int m0, const float (&v)[kTnsGroup]) {
    if (m0 >= L.len) return;
    if (L.aligned && m0 + kTnsGroup <= L.len) {
#pragma unroll
        for (int q = 0; q < kTnsGroup / 4; ++q)
            *reinterpret_cast<float4 *>(L.down ? L.x - m0 - 4 * q - 3 : L.x + m0 + 4 * q) =
                L.down ? make_float4(v[4 * q + 3], v[4 * q + 2], v[4 * q + 1], v[4 * q]) : make_float4(v[4 * q], v[4 * q + 1], v[4 * q + 2], v[4 * q + 3]);
    } else {
#pragma unroll
        for (int k = 0; k < kTnsGroup; ++k)
            if (m0 + k < L.len) L.x[L.down ? -(long)(m0 + k) : (long)(m0 + k)] = v[k];
    }
}

// The walk with the tap count as a compile-time constant: TAPS = the wavefront's highest order rounded up to 4, 8, 12, 16
// or 20.  (With a run-time bound the compiler evaluated all 20 taps of every line behind selects: 125 instructions per
// line for order-12 filters.)  coeffs[i] -= coeffs[i -+ (j + 1)] * lpc[j] in tap order (tns.rs:180-195); a tap that does
// not apply subtracts +0.0 instead -- x - +0.0 == x for every x, -0.0 and NaN included -- so the predicate is applied to
// the PRODUCT, off the serial chain.  Two forms of that predicate:
//  * START (the first TAPS lines of a range): line m only sees min(order, m) earlier lines (tns.rs:184, 191) -- a compare
//    per tap;
//  * steady state: the bound is the lane's order, the same for every line: a 0 / ~0 word per tap, ANDed onto the product
//    (one AND instead of a compare and a select).
// Lines past a lane's range are computed too (their results are never stored and the lane's history no longer matters), so
// the history shift is an unconditional register renaming.
template <int TAPS, bool START>
__device__ __forceinline__ void tns_lines16(float (&cur)[kTnsGroup], float (&h)[TAPS], const float (&lpc)[TAPS], const unsigned (&mask)[TAPS],
                                            int order, int m0) {
#pragma unroll
    for (int k = 0; k < kTnsGroup; ++k) {
        float acc = cur[k];
        const int m = m0 + k;
        const int lim = order < m ? order : m;
#pragma unroll
        for (int j = 0; j < TAPS; ++j) {
            float term;
            if constexpr (START) term = j < lim ? h[j] * lpc[j] : 0.0f;
            else term = __uint_as_float(__float_as_uint(h[j] * lpc[j]) & mask[j]);
            acc = acc - term;
        }
        cur[k] = acc;
#pragma unroll
        for (int j = TAPS - 1; j >= 1; --j) h[j] = h[j - 1];
        h[0] = acc;
    }
}

template <int TAPS>
__device__ __forceinline__ void tns_walk(const TnsLane &L, const float *lpc_all, int max_len) {
    float lpc[TAPS], h[TAPS];
    unsigned mask[TAPS];
#pragma unroll
    for (int j = 0; j < TAPS; ++j) {
        lpc[j] = lpc_all[j];
        h[j] = 0.0f;
        mask[j] = j < L.order ? 0xffffffffu : 0u;
    }
    // The next group is fetched while the current one is filtered: the inputs are the unfiltered lines, independent of the outputs.
    float cur[kTnsGroup], nxt[kTnsGroup];
#pragma unroll
    for (int k = 0; k < kTnsGroup; ++k) cur[k] = nxt[k] = 0.0f;
    tns_fetch(L, 0, cur);
    for (int m0 = 0; m0 < max_len; m0 += kTnsGroup) {
        tns_fetch(L, m0 + kTnsGroup, nxt);
        if (m0 < TAPS)  // (wave-uniform) the groups that contain a range's first TAPS lines
            tns_lines16<TAPS, true>(cur, h, lpc, mask, L.order, m0);
        else
            tns_lines16<TAPS, false>(cur, h, lpc, mask, L.order, m0);
        tns_store(L, m0, cur);
#pragma unroll
        for (int k = 0; k < kTnsGroup; ++k) cur[k] = nxt[k];
    }
}

__global__ __launch_bounds__(64) void aac_tns_kernel(float *__restrict__ coeffs, unsigned n_frames,
                                                     const symaccel_aac_tns_filter *__restrict__ filters, unsigned n_filters) {
    const unsigned idx = blockIdx.x * 64u + threadIdx.x;
    TnsLane L{coeffs, 0, 0, false, false};
    float lpc[kTnsMaxOrder];
#pragma unroll
    for (int j = 0; j < kTnsMaxOrder; ++j) lpc[j] = 0.0f;
    if (idx < n_filters) {
        const symaccel_aac_tns_filter &f = filters[idx];
        const int start = f.start, end = f.end;
        if (f.frame < n_frames && start < end && end <= 1024 && f.order >= 1 && f.order <= kTnsMaxOrder) {
            L.order = f.order;
            L.len = end - start;
            L.down = f.direction != 0;
            L.x = coeffs + (size_t)f.frame * 1024 + (L.down ? end - 1 : start);
            L.aligned = ((reinterpret_cast<uintptr_t>(L.x) + (L.down ? 4 : 0)) & 15u) == 0;  // x = first line (up) / last line (down)
#pragma unroll
            for (int j = 0; j < kTnsMaxOrder; ++j) lpc[j] = f.lpc[j];
        }
    }
    // wave-uniform bounds: the longest range and the highest order among the wavefront's filters
    int max_len = L.len, max_order = L.order;
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) {
        const int a = __shfl_xor(max_len, m), b = __shfl_xor(max_order, m);
        max_len = a > max_len ? a : max_len;
        max_order = b > max_order ? b : max_order;
    }
    if (max_order <= 4)
        tns_walk<4>(L, lpc, max_len);
    else if (max_order <= 8)
        tns_walk<8>(L, lpc, max_len);
    else if (max_order <= 12)
        tns_walk<12>(L, lpc, max_len);
    else if (max_order <= 16)
        tns_walk<16>(L, lpc, max_len);
    else
        tns_walk<20>(L, lpc, max_len);
}

}  // namespace

int launch_aac_joint_stereo(symaccel_ctx *ctx, const AacBandMaps &maps, float *d_coeffs, size_t frames_per_chain,
                            const int32_t *d_pair_chains, const symaccel_aac_js_frame *d_desc, size_t n_pairs) {
    const size_t grid = n_pairs * frames_per_chain;
    if (grid > 0x7fffffffu || frames_per_chain > 0xffffffffu) return SYMACCEL_ERR_INVALID_ARG;
    hipLaunchKernelGGL(aac_joint_stereo_kernel, dim3((unsigned)grid), dim3(256), 0, ctx->stream, maps, d_coeffs,
                       (unsigned)frames_per_chain, d_pair_chains, d_desc);
    SYM_GPU(ctx, hipGetLastError());
    return SYMACCEL_OK;
}

int launch_aac_tns(symaccel_ctx *ctx, float *d_coeffs, size_t n_frames, const symaccel_aac_tns_filter *d_filters,
                   size_t n_filters) {
    const size_t grid = (n_filters + 63) / 64;
    if (n_frames > 0xffffffffu || n_filters > 0xffffffffu || grid > 0x7fffffffu) return SYMACCEL_ERR_INVALID_ARG;
    hipLaunchKernelGGL(aac_tns_kernel, dim3((unsigned)grid), dim3(64), 0, ctx->stream, d_coeffs, (unsigned)n_frames, d_filters,
                       (unsigned)n_filters);
    SYM_GPU(ctx, hipGetLastError());
    return SYMACCEL_OK;
}

}  // namespace symaccel
