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
// (one 64-byte sector) per lane and round, the next sixteen requested while the current ones are filtered; the four lanes
// of a quad move their four groups together (64-byte requests) and transpose them among themselves.  The taps are
// applied in the reference's order, each as a rounded multiply and a rounded subtract, and only `min(order, lines
// filtered so far)` of them (tns.rs:184, 191).  (Staging the lines through an LDS tile of 64 filters x 32 lines, as the
// integer predictors do -- 128-byte row segments on the HBM side -- was built and measured twice, before and after the
// tap loop became compile-time: 0.50 against 0.38 ms, then 0.33 against 0.30 ms for 131 072 order-12 filters.  Not kept;
// profiles/r02n_tns_alac_ab.txt, profiles/r02zc_tns_ab.txt.)
#include <hip/hip_runtime.h>

#include <cstdlib>

#include "symaccel_internal.h"

namespace symaccel {

namespace {

// `list` (optional): the channel-pair frames to decode, pair * frames_per_chain + frame each -- the workgroups of a launch over
// a list touch nothing else (the frames that carry TNS filters in front of the fused pair walk: symaccel_aac_decode_pipelined).
__global__ __launch_bounds__(256) void aac_joint_stereo_kernel(AacBandMaps maps, float *__restrict__ coeffs,
                                                               unsigned frames_per_chain, const int32_t *__restrict__ pair_chains,
                                                               const symaccel_aac_js_frame *__restrict__ desc,
                                                               const uint32_t *__restrict__ list, unsigned n_pair_frames) {
    const unsigned pf = list ? list[blockIdx.x] : blockIdx.x;
    if (pf >= n_pair_frames) return;  // (a list entry outside the batch)
    const unsigned pair = pf / frames_per_chain, f = pf % frames_per_chain;
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

// The descriptors of the listed pair frames become "nothing coded": what the fused pair walk must see for frames whose joint
// stereo was decoded in place by the list pass above.
__global__ void aac_js_consume_kernel(symaccel_aac_js_frame *__restrict__ desc, const uint32_t *__restrict__ list, unsigned n_list,
                                      unsigned n_pair_frames) {
    const unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n_list && list[i] < n_pair_frames) desc[list[i]].max_sfb = 0;
}

constexpr int kTnsMaxOrder = 20;  // TNS_MAX_ORDER, tns.rs:22
constexpr int kTnsGroup = 16;     // lines per lane and round
constexpr int kTnsSinkSlots = 256;  // wavefront slots of the sink buffer the small pass aims its idle pieces at
static_assert((size_t)kTnsSinkSlots * 64 * 2 * kTnsGroup * sizeof(float) <= kSinkBytes, "sink slots");

// One lane's filter: where it walks and what it multiplies by.
struct TnsLane {
    float *x;        // first line filtered (start, or end - 1 for a downward filter)
    int len, order;  // lines in the range; taps
    bool down, aligned;
};

// Value of lane (lane ^ S) of the same quad, S = 1 or 2: one DPP move (quad_perm), no LDS.
template <int S>
__device__ __forceinline__ uint32_t quad_xor(uint32_t v) {
#if defined(__HIP_DEVICE_COMPILE__)
    return (uint32_t)__builtin_amdgcn_mov_dpp((int)v, S == 1 ? 0xB1 : 0x4E, 0xf, 0xf, true);  // quad_perm [1,0,3,2] / [2,3,0,1]
#else
    return (uint32_t)__shfl_xor((int)v, S);
#endif
}
// Value of lane J of the caller's quad.
template <int J>
__device__ __forceinline__ uint32_t quad_bcast(uint32_t v) {
#if defined(__HIP_DEVICE_COMPILE__)
    return (uint32_t)__builtin_amdgcn_mov_dpp((int)v, J * 0x55, 0xf, 0xf, true);  // quad_perm [J,J,J,J]
#else
    return (uint32_t)__shfl((int)v, (int)((threadIdx.x & 60u) | (unsigned)J));
#endif
}
// Transpose of a 4 x 4 matrix of 16-byte elements held one row per lane of a quad: afterwards element j of lane i is what
// was element i of lane j.  Two butterfly steps (lane bit 0 with index bit 0, lane bit 1 with index bit 1).
__device__ __forceinline__ void quad_transpose(uint4 (&e)[4], int lane) {
    const bool b0 = (lane & 1) != 0, b1 = (lane & 2) != 0;
    auto swap_step = [](uint4 &lo, uint4 &hi, bool b, auto xchg) {
        uint32_t *l = &lo.x, *h = &hi.x;
#pragma unroll
        for (int d = 0; d < 4; ++d) {
            const uint32_t r = xchg(b ? l[d] : h[d]);  // the lower lane sends its upper element, the upper lane its lower one
            l[d] = b ? r : l[d];
            h[d] = b ? h[d] : r;
        }
    };
    swap_step(e[0], e[1], b0, [](uint32_t v) { return quad_xor<1>(v); });
    swap_step(e[2], e[3], b0, [](uint32_t v) { return quad_xor<1>(v); });
    swap_step(e[0], e[2], b1, [](uint32_t v) { return quad_xor<2>(v); });
    swap_step(e[1], e[3], b1, [](uint32_t v) { return quad_xor<2>(v); });
}

// The four filters of a quad, as every lane of the quad sees them: filter j's first line (element index into coeffs),
// length and whether a full group of it can be moved as four aligned 16-byte pieces.
struct TnsQuad {
    size_t idx[4];
    int len[4];
    bool down[4], aligned[4];
};
__device__ __forceinline__ TnsQuad tns_quad(const TnsLane &L, const float *coeffs) {
    TnsQuad Q;
    const uint64_t p = (uint64_t)(L.x - coeffs);
    const uint32_t lo = (uint32_t)p, hi = (uint32_t)(p >> 32), meta = (uint32_t)L.len | (L.down ? 1u << 16 : 0u) | (L.aligned ? 1u << 17 : 0u);
    auto take = [&](auto bc, int j) {
        const uint32_t m = bc(meta);
        Q.idx[j] = (size_t)((uint64_t)bc(lo) | (uint64_t)bc(hi) << 32);
        Q.len[j] = (int)(m & 0xffffu);
        Q.down[j] = (m >> 16 & 1u) != 0;
        Q.aligned[j] = (m >> 17 & 1u) != 0;
    };
    take([](uint32_t v) { return quad_bcast<0>(v); }, 0);
    take([](uint32_t v) { return quad_bcast<1>(v); }, 1);
    take([](uint32_t v) { return quad_bcast<2>(v); }, 2);
    take([](uint32_t v) { return quad_bcast<3>(v); }, 3);
    return Q;
}

// Sixteen lines (64 bytes) per lane and round.  A lane on its own would move them as four 16-byte accesses, each to a line of
// its own: 64 separate requests per wavefront instruction -- the address unit was busy 84 % of the kernel and the L2 saw the
// stores as 16-byte writes (profiles/r02z_tns_sq_counters.txt).  So the four lanes of a quad move their four filters'
// groups TOGETHER: instruction j covers the 64-byte group of filter j, a quarter per lane (one contiguous 64-byte request),
// and a 4 x 4 transpose inside the quad (DPP moves, no LDS) hands every lane its own filter's sixteen lines.
//  * tns_request only ISSUES the loads (unconditionally: a filter with no full aligned group at m0 reads coeffs[0..3]
//    instead, which is discarded) -- no branch, no use of the data, so they stay in flight while the previous group is
//    filtered; tns_take transposes them when the group's turn comes.
//  * A filter whose group is ragged or unaligned is moved by its own lane, scalar, when its turn comes.
struct TnsRaw {
    uint4 e[4];
};
__device__ __forceinline__ void tns_request(const float *coeffs, const TnsQuad &Q, int lane, int m0, TnsRaw &raw) {
    const int i = lane & 3;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const bool full = Q.aligned[j] && m0 + kTnsGroup <= Q.len[j];
        const size_t at = Q.down[j] ? Q.idx[j] - (size_t)(m0 + 4 * i + 3) : Q.idx[j] + (size_t)(m0 + 4 * i);
        raw.e[j] = *reinterpret_cast<const uint4 *>(coeffs + (full ? at : (size_t)0));
    }
}
__device__ __forceinline__ void tns_take(const TnsLane &L, TnsRaw &raw, int lane, int m0, float (&v)[kTnsGroup]) {
    quad_transpose(raw.e, lane);
    if (m0 >= L.len) return;
    if (L.aligned && m0 + kTnsGroup <= L.len) {
#pragma unroll
        for (int q = 0; q < kTnsGroup / 4; ++q) {
            v[4 * q + 0] = __uint_as_float(L.down ? raw.e[q].w : raw.e[q].x);
            v[4 * q + 1] = __uint_as_float(L.down ? raw.e[q].z : raw.e[q].y);
            v[4 * q + 2] = __uint_as_float(L.down ? raw.e[q].y : raw.e[q].z);
            v[4 * q + 3] = __uint_as_float(L.down ? raw.e[q].x : raw.e[q].w);
        }
    } else {
#pragma unroll
        for (int k = 0; k < kTnsGroup; ++k)
            if (m0 + k < L.len) v[k] = L.x[L.down ? -(long)(m0 + k) : (long)(m0 + k)];
    }
}
__device__ __forceinline__ void tns_store(float *coeffs, const TnsLane &L, const TnsQuad &Q, int lane, int m0, const float (&v)[kTnsGroup]) {
    const int i = lane & 3;
    uint4 e[4];
#pragma unroll
    for (int q = 0; q < kTnsGroup / 4; ++q)
        e[q] = L.down ? make_uint4(__float_as_uint(v[4 * q + 3]), __float_as_uint(v[4 * q + 2]), __float_as_uint(v[4 * q + 1]), __float_as_uint(v[4 * q]))
                      : make_uint4(__float_as_uint(v[4 * q]), __float_as_uint(v[4 * q + 1]), __float_as_uint(v[4 * q + 2]), __float_as_uint(v[4 * q + 3]));
    quad_transpose(e, lane);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        if (Q.aligned[j] && m0 + kTnsGroup <= Q.len[j])
            *reinterpret_cast<uint4 *>(coeffs + (Q.down[j] ? Q.idx[j] - (size_t)(m0 + 4 * i + 3) : Q.idx[j] + (size_t)(m0 + 4 * i))) = e[j];
    }
    if (m0 < L.len && !(L.aligned && m0 + kTnsGroup <= L.len)) {
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
__device__ __forceinline__ void tns_walk(float *coeffs, const TnsLane &L, const float *lpc_all, int max_len, int lane) {
    const TnsQuad Q = tns_quad(L, coeffs);
    float lpc[TAPS], h[TAPS];
    unsigned mask[TAPS];
#pragma unroll
    for (int j = 0; j < TAPS; ++j) {
        lpc[j] = lpc_all[j];
        h[j] = 0.0f;
        mask[j] = j < L.order ? 0xffffffffu : 0u;
    }
    // The inputs are the unfiltered lines, independent of the outputs: the next group is requested before the current one
    // is filtered (two alternating register groups, the round loop unrolled by two so that every index is static).
    TnsRaw raw[2];
    float cur[kTnsGroup];
#pragma unroll
    for (int k = 0; k < kTnsGroup; ++k) cur[k] = 0.0f;
    tns_request(coeffs, Q, lane, 0, raw[0]);
    for (int m0 = 0; m0 < max_len; m0 += 2 * kTnsGroup) {
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            const int m = m0 + r * kTnsGroup;
            if (m < max_len) {  // (wave-uniform)
                tns_request(coeffs, Q, lane, m + kTnsGroup, raw[r ^ 1]);
                tns_take(L, raw[r], lane, m, cur);
                if (m < TAPS)  // (wave-uniform) the groups that contain a range's first TAPS lines
                    tns_lines16<TAPS, true>(cur, h, lpc, mask, L.order, m);
                else
                    tns_lines16<TAPS, false>(cur, h, lpc, mask, L.order, m);
                tns_store(coeffs, L, Q, lane, m, cur);
            }
        }
    }
}

// ---- two filters per lane (round 6).  The filter pass runs below one wavefront per SIMD (39 k filters of a 131 072-frame batch
// are 616 wavefronts for 1024 SIMDs) and a wavefront issues one instruction per ~4.9 cycles whatever it is: what bounds the
// pass is the NUMBER of instructions a wavefront issues per line, 3 per tap (multiply, mask, subtract).  With two filters per
// lane every tap is one v_pk_mul_f32 and one v_pk_add_f32 (neg) for BOTH -- the same IEEE multiply and subtract per
// component, so the same bits -- and when every filter of the wavefront has the class's full order (the common case: AAC-LC
// long windows, order 12) the mask disappears as well: 2 instructions per tap and line PAIR instead of 6.
typedef float v2f __attribute__((vector_size(8)));  // (GCC / clang vector extension: the emulation build is g++)
#ifndef SYM_TNS_ABLATE
#define SYM_TNS_ABLATE 0
#endif
#ifndef SYM_TNS_AHEAD
#define SYM_TNS_AHEAD 2  // groups in flight behind the current one in a small pass (build-time tuning knob)
#endif

// MODE 0: the groups that hold a range's first TAPS lines; 1: steady state, orders below TAPS present (mask on the product);
// 2: steady state, every filter of the wavefront has order == TAPS.
// (Forming the products of line m + 1 between the subtractions of line m -- they only need lines up to m - 1 -- would also remove the wait
// state a packed instruction costs when it reads the result of the instruction right before it: mul, nop, sub per tap today.  The compiler
// sinks every product back to its use, and an empty asm that pins it costs the same wait state at its boundary; not kept.)
template <int TAPS, int MODE>
__device__ __forceinline__ void tns_lines16_2(v2f (&cur)[kTnsGroup], v2f (&h)[TAPS], const v2f (&lpc)[TAPS], const unsigned (&mask)[2][TAPS],
                                              int order0, int order1, int m0) {
#pragma unroll
    for (int k = 0; k < kTnsGroup; ++k) {
        v2f acc = cur[k];
        const int m = m0 + k;
        const int lim0 = order0 < m ? order0 : m, lim1 = order1 < m ? order1 : m;
#pragma unroll
        for (int j = 0; j < TAPS; ++j) {
            v2f term = h[j] * lpc[j];
            if constexpr (MODE == 0) term = v2f{j < lim0 ? term[0] : 0.0f, j < lim1 ? term[1] : 0.0f};
            else if constexpr (MODE == 1)
                term = v2f{__uint_as_float(__float_as_uint(term[0]) & mask[0][j]), __uint_as_float(__float_as_uint(term[1]) & mask[1][j])};
            acc = acc - term;
        }
        cur[k] = acc;
#pragma unroll
        for (int j = TAPS - 1; j >= 1; --j) h[j] = h[j - 1];
        h[0] = acc;
    }
}

// The walk of tns_walk with two filters per lane: the lines of both move as there (quads, 64-byte requests), the arithmetic is packed.
// FULL: every filter of the wavefront has order == TAPS (an instantiation of its own: the masks of the other cost 2 x TAPS registers)
template <int TAPS, bool FULL>
__device__ __forceinline__ void tns_walk2(float *coeffs, const TnsLane &L0, const TnsLane &L1, const float (&lpc0)[TAPS], const float (&lpc1)[TAPS],
                                          int max_len, int lane) {
    const TnsQuad Q0 = tns_quad(L0, coeffs), Q1 = tns_quad(L1, coeffs);
    v2f lpc[TAPS], h[TAPS];
    unsigned mask[2][TAPS];
#pragma unroll
    for (int j = 0; j < TAPS; ++j) {
        lpc[j] = v2f{lpc0[j], lpc1[j]};
        h[j] = v2f{0.0f, 0.0f};
        mask[0][j] = j < L0.order ? 0xffffffffu : 0u;
        mask[1][j] = j < L1.order ? 0xffffffffu : 0u;
    }
    TnsRaw raw0[2], raw1[2];
    float c0[kTnsGroup], c1[kTnsGroup];
#pragma unroll
    for (int k = 0; k < kTnsGroup; ++k) c0[k] = c1[k] = 0.0f;
    tns_request(coeffs, Q0, lane, 0, raw0[0]);
    tns_request(coeffs, Q1, lane, 0, raw1[0]);
    for (int m0 = 0; m0 < max_len; m0 += 2 * kTnsGroup) {
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            const int m = m0 + r * kTnsGroup;
            if (m < max_len) {  // (wave-uniform)
                tns_request(coeffs, Q0, lane, m + kTnsGroup, raw0[r ^ 1]);
                tns_request(coeffs, Q1, lane, m + kTnsGroup, raw1[r ^ 1]);
                tns_take(L0, raw0[r], lane, m, c0);
                tns_take(L1, raw1[r], lane, m, c1);
                v2f cur[kTnsGroup];
#pragma unroll
                for (int k = 0; k < kTnsGroup; ++k) cur[k] = v2f{c0[k], c1[k]};
                if (m < TAPS)  // (wave-uniform) the groups that contain a range's first TAPS lines
                    tns_lines16_2<TAPS, 0>(cur, h, lpc, mask, L0.order, L1.order, m);
                else
                    tns_lines16_2<TAPS, FULL ? 2 : 1>(cur, h, lpc, mask, L0.order, L1.order, m);
#pragma unroll
                for (int k = 0; k < kTnsGroup; ++k) {
                    c0[k] = cur[k][0];
                    c1[k] = cur[k][1];
                }
                tns_store(coeffs, L0, Q0, lane, m, c0);
                tns_store(coeffs, L1, Q1, lane, m, c1);
            }
        }
    }
}

// ---- the SMALL pass (below ~two wavefronts per SIMD: 39 k filters of a 131 072-frame batch are 308 wavefronts of 128 filters for 1024
// SIMDs).  Nothing is bound by the L2 or by HBM then: the pass takes as long as ONE wavefront needs for its own filters.  Two things
// differ from the walks above:
//  * every lane moves its own filters' lines, four 16-byte pieces per group (64 separate requests per instruction: fine at this
//    occupancy, too slow at 2048 wavefronts -- profiles/r02z_tns_sq_counters.txt); no quad exchange (two 4 x 4 transposes per group
//    and filter in DPP moves and selects);
//  * the round is STRAIGHT-LINE code: every piece is loaded and stored unconditionally -- a piece that lies outside its range (or belongs
//    to a lane whose group is ragged / unaligned) reads coeffs[0..3] and writes a per-lane slot of the context's sink buffer.  gfx950
//    counts vector loads and stores with ONE in-order counter: with the ragged path's loads and the stores inside per-lane branches the
//    compiler cannot know how many follow a prefetch and waited with vmcnt(0) in every round -- for the loads it had just issued and the
//    stores of the round before, a memory round trip per sixteen lines whatever the number of groups requested ahead.  The ragged path
//    (a length that is not a multiple of four lines, an unaligned range: no swb table produces either) sits behind ONE wave-uniform branch.
// Measured on 39 424 order-12 filters of 512 lines (profiles/r06z6 .. r06z11): one filter per lane with the quad exchange (round 5) 94 us;
// two per lane, packed, quad exchange 97 (90 with the ILP scheduling strategy, build.py); lane by lane 83; straight-line 70; ILP strategy
// 68.  By ablation the arithmetic alone takes 40 us and the movement alone 69; the SQ counters say 19.6 k VALU instructions per wavefront
// (58 % of its cycles; 384 of a round's 614 are the packed taps) and a quarter of its cycles waiting.  An exchange through LDS (a quad
// moves 64 contiguous bytes, four ds_write_b128 + four ds_read_b128 per group instead of the DPP transposes) measured slower, 76 us.
// End of a ragged section: everything in flight is waited for HERE, inside the wave-uniform branch.  Without it the compiler merges "an
// unknown number of loads / stores may be pending" into the straight-line path at the join and waits with vmcnt(0) there.
__device__ __forceinline__ void tns_slow_path_drain() {
#if defined(__HIP_DEVICE_COMPILE__)
    __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0), expcnt / lgkmcnt untouched (gfx9 encoding: vmcnt [3:0] and [15:14], expcnt [6:4], lgkmcnt [11:8])
#endif
}
struct TnsPieces {
    const float *base;  // lowest address of the group's 64 bytes
    bool whole[4];      // piece q (lines 4 q .. 4 q + 3 of the group, ascending ADDRESSES) moves as one 16-byte access
    bool slow;          // the lane moves this group line by line
};
__device__ __forceinline__ TnsPieces tns_pieces(const TnsLane &L, int m) {
    TnsPieces P;
    const int left = L.len - m;  // lines of the range from m on
    P.slow = left > 0 && (!L.aligned || (left < kTnsGroup && (left & 3) != 0));
    P.base = L.down ? L.x - (m + kTnsGroup - 1) : L.x + m;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        // ascending addresses: an upward filter's piece q holds lines 4 q .. 4 q + 3, a downward one's lines 12 - 4 q .. 15 - 4 q
        const int last = L.down ? kTnsGroup - 4 * q : 4 * q + 4;  // the piece is whole when the group has that many lines
        P.whole[q] = L.aligned && !P.slow && left >= last;
    }
    return P;
}

template <int TAPS, bool FULL>
__device__ __forceinline__ void tns_walk2_direct(float *coeffs, float *sink_lane, const TnsLane &L0, const TnsLane &L1, const float (&lpc0)[TAPS],
                                                 const float (&lpc1)[TAPS], int max_len) {
    v2f lpc[TAPS], h[TAPS];
    unsigned mask[2][TAPS];
#pragma unroll
    for (int j = 0; j < TAPS; ++j) {
        lpc[j] = v2f{lpc0[j], lpc1[j]};
        h[j] = v2f{0.0f, 0.0f};
        mask[0][j] = j < L0.order ? 0xffffffffu : 0u;
        mask[1][j] = j < L1.order ? 0xffffffffu : 0u;
    }
    auto request = [&](const TnsLane &L, int m, TnsRaw &raw) {
        const TnsPieces P = tns_pieces(L, m);
#pragma unroll
        for (int q = 0; q < 4; ++q) raw.e[q] = *reinterpret_cast<const uint4 *>(P.whole[q] ? P.base + 4 * q : coeffs);
    };
    auto take = [&](const TnsLane &L, const TnsRaw &raw, float (&v)[kTnsGroup]) {  // (pieces that are not whole: never stored)
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            const uint4 up = raw.e[p], dn = raw.e[3 - p];
            v[4 * p + 0] = __uint_as_float(L.down ? dn.w : up.x);
            v[4 * p + 1] = __uint_as_float(L.down ? dn.z : up.y);
            v[4 * p + 2] = __uint_as_float(L.down ? dn.y : up.z);
            v[4 * p + 3] = __uint_as_float(L.down ? dn.x : up.w);
        }
    };
    auto store = [&](const TnsLane &L, const TnsPieces &P, float *sink16, const float (&v)[kTnsGroup]) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int p = 3 - q;
            const uint4 e = L.down ? make_uint4(__float_as_uint(v[4 * p + 3]), __float_as_uint(v[4 * p + 2]), __float_as_uint(v[4 * p + 1]), __float_as_uint(v[4 * p]))
                                   : make_uint4(__float_as_uint(v[4 * q]), __float_as_uint(v[4 * q + 1]), __float_as_uint(v[4 * q + 2]), __float_as_uint(v[4 * q + 3]));
            *reinterpret_cast<uint4 *>(P.whole[q] ? const_cast<float *>(P.base) + 4 * q : sink16 + 4 * q) = e;
        }
    };
    // kAhead groups of sixteen lines are in flight behind the one being filtered (a ring of kAhead + 1 register groups, the round loop
    // unrolled by as many so that every index is static)
    constexpr int kAhead = SYM_TNS_AHEAD, kRing = kAhead + 1;
    TnsRaw raw0[kRing], raw1[kRing];
    float c0[kTnsGroup], c1[kTnsGroup];
#pragma unroll
    for (int a = 0; a < kAhead; ++a) {
        request(L0, a * kTnsGroup, raw0[a]);
        request(L1, a * kTnsGroup, raw1[a]);
    }
    for (int m0 = 0; m0 < max_len; m0 += kRing * kTnsGroup) {
#pragma unroll
        for (int r = 0; r < kRing; ++r) {
            const int m = m0 + r * kTnsGroup;
            if (m < max_len) {  // (wave-uniform)
#if SYM_TNS_ABLATE == 2  // (measurement build, results wrong on purpose: only the first groups are loaded, only the last group is stored)
                if (m == 0) {
#endif
                request(L0, m + kAhead * kTnsGroup, raw0[(r + kAhead) % kRing]);
                request(L1, m + kAhead * kTnsGroup, raw1[(r + kAhead) % kRing]);
#if SYM_TNS_ABLATE == 2
                }
#endif
                const TnsPieces P0 = tns_pieces(L0, m), P1 = tns_pieces(L1, m);
                take(L0, raw0[r], c0);
                take(L1, raw1[r], c1);
                const bool any_slow = __any(P0.slow || P1.slow) != 0;
                if (any_slow) {  // (wave-uniform; rare)
                    if (P0.slow) {
#pragma unroll
                        for (int k = 0; k < kTnsGroup; ++k)
                            if (m + k < L0.len) c0[k] = L0.x[L0.down ? -(long)(m + k) : (long)(m + k)];
                    }
                    if (P1.slow) {
#pragma unroll
                        for (int k = 0; k < kTnsGroup; ++k)
                            if (m + k < L1.len) c1[k] = L1.x[L1.down ? -(long)(m + k) : (long)(m + k)];
                    }
                    tns_slow_path_drain();
                }
                v2f cur[kTnsGroup];
#pragma unroll
                for (int k = 0; k < kTnsGroup; ++k) cur[k] = v2f{c0[k], c1[k]};
#if SYM_TNS_ABLATE != 1  // (measurement build, results wrong on purpose: the lines pass through unfiltered)
                if (m < TAPS)  // (wave-uniform) the groups that contain a range's first TAPS lines
                    tns_lines16_2<TAPS, 0>(cur, h, lpc, mask, L0.order, L1.order, m);
                else
                    tns_lines16_2<TAPS, FULL ? 2 : 1>(cur, h, lpc, mask, L0.order, L1.order, m);
#endif
#pragma unroll
                for (int k = 0; k < kTnsGroup; ++k) {
                    c0[k] = cur[k][0];
                    c1[k] = cur[k][1];
                }
#if SYM_TNS_ABLATE == 2
                if (m + kTnsGroup >= max_len) {
#endif
                store(L0, P0, sink_lane, c0);
                store(L1, P1, sink_lane + kTnsGroup, c1);
#if SYM_TNS_ABLATE == 2
                }
#endif
                if (any_slow) {
                    if (P0.slow) {
#pragma unroll
                        for (int k = 0; k < kTnsGroup; ++k)
                            if (m + k < L0.len) L0.x[L0.down ? -(long)(m + k) : (long)(m + k)] = c0[k];
                    }
                    if (P1.slow) {
#pragma unroll
                        for (int k = 0; k < kTnsGroup; ++k)
                            if (m + k < L1.len) L1.x[L1.down ? -(long)(m + k) : (long)(m + k)] = c1[k];
                    }
                    tns_slow_path_drain();
                }
            }
        }
    }
}

// Whether filter idx is one the pass runs, and its order (0 otherwise)
__device__ __forceinline__ int tns_filter_order(const symaccel_aac_tns_filter *filters, unsigned idx, unsigned n_filters, unsigned n_frames) {
    if (idx >= n_filters) return 0;
    const symaccel_aac_tns_filter &f = filters[idx];
    const int start = f.start, end = f.end;
    const bool valid = f.frame < n_frames && start < end && end <= 1024 && f.order >= 1 && f.order <= kTnsMaxOrder;
    return valid ? (int)f.order : 0;
}
__device__ __forceinline__ int wave_max(int v) {
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) {
        const int b = __shfl_xor(v, m);
        v = b > v ? b : v;
    }
    return v;
}
template <int TAPS>
__device__ __forceinline__ TnsLane tns_lane(float *coeffs, const symaccel_aac_tns_filter *filters, unsigned idx, int order, float (&lpc)[TAPS]) {
    TnsLane L{coeffs, 0, 0, false, false};
#pragma unroll
    for (int j = 0; j < TAPS; ++j) lpc[j] = 0.0f;
    if (order > 0) {
        const symaccel_aac_tns_filter &f = filters[idx];
        const int start = f.start, end = f.end;
        L.order = order;
        L.len = end - start;
        L.down = f.direction != 0;
        L.x = coeffs + (size_t)f.frame * 1024 + (L.down ? end - 1 : start);
        L.aligned = ((reinterpret_cast<uintptr_t>(L.x) + (L.down ? 4 : 0)) & 15u) == 0;  // x = first line (up) / last line (down)
#pragma unroll
        for (int j = 0; j < TAPS; ++j) lpc[j] = f.lpc[j];
    }
    return L;
}

// A block of 128 consecutive filters of the flat list on ONE wavefront, filters b + lane and b + 64 + lane on lane `lane`, with the tap count
// of the block's class (its highest order rounded up to 4, 8 or 12) as a compile-time constant.
template <int TAPS, bool DIRECT>
__device__ __forceinline__ void tns_pair_block(float *coeffs, const symaccel_aac_tns_filter *filters, float *sink, unsigned idx0, unsigned idx1, int order0,
                                               int order1, int lane) {
    const bool all_full = wave_max(((order0 != 0 && order0 != TAPS) || (order1 != 0 && order1 != TAPS)) ? 1 : 0) == 0;
    float lpc0[TAPS], lpc1[TAPS];
    const TnsLane L0 = tns_lane<TAPS>(coeffs, filters, idx0, order0, lpc0), L1 = tns_lane<TAPS>(coeffs, filters, idx1, order1, lpc1);
    const int max_len = wave_max(L0.len > L1.len ? L0.len : L1.len);  // wave-uniform bound: the longest range among the block's filters
    if constexpr (DIRECT) {
        // the lane's 128 bytes of the sink (symaccel_internal.h: kSinkBytes, never read), a slot of 8 KiB per wavefront
        float *sink_lane = sink + (size_t)(blockIdx.x % (unsigned)kTnsSinkSlots) * (64 * 2 * kTnsGroup) + lane * (2 * kTnsGroup);
        if (all_full)  // (wave-uniform)
            tns_walk2_direct<TAPS, true>(coeffs, sink_lane, L0, L1, lpc0, lpc1, max_len);
        else
            tns_walk2_direct<TAPS, false>(coeffs, sink_lane, L0, L1, lpc0, lpc1, max_len);
    } else {
        if (all_full)  // (wave-uniform)
            tns_walk2<TAPS, true>(coeffs, L0, L1, lpc0, lpc1, max_len, lane);
        else
            tns_walk2<TAPS, false>(coeffs, L0, L1, lpc0, lpc1, max_len, lane);
    }
}

// ONE launch for every tap class: a workgroup of two wavefronts takes a block of 128 filters.
//  * orders up to 12: wavefront 0 alone, two filters per lane, the tap count a compile-time constant of the block's class (4, 8 or 12: the
//    walks need 195 .. 227 registers each, so the shared allocation costs nothing); wavefront 1 leaves at once;
//  * orders 13 .. 20: both wavefronts, one filter per lane (forty packed coefficient and history registers more than the pair form has room for).
// A kernel per class (rounds 2 .. 5) was a launch each -- 3-5 us of stream time that most batches leave at once.
template <bool DIRECT>
__global__ __launch_bounds__(128) void aac_tns_kernel(float *__restrict__ coeffs, unsigned n_frames, const symaccel_aac_tns_filter *__restrict__ filters,
                                                      unsigned n_filters, float *__restrict__ sink) {
    const unsigned wave = threadIdx.x >> 6, lane = threadIdx.x & 63u;
    const unsigned idx0 = blockIdx.x * 128u + lane, idx1 = idx0 + 64u;
    // the block's class first: only the order bytes (one 4-byte word per filter)
    const int order0 = tns_filter_order(filters, idx0, n_filters, n_frames), order1 = tns_filter_order(filters, idx1, n_filters, n_frames);
    const int max_order = wave_max(order0 > order1 ? order0 : order1);  // (the same in both wavefronts)
    if (max_order == 0) return;
    if (max_order > 12) {
        float lpc[kTnsMaxOrder];
        const TnsLane L = tns_lane<kTnsMaxOrder>(coeffs, filters, wave ? idx1 : idx0, wave ? order1 : order0, lpc);
        tns_walk<kTnsMaxOrder>(coeffs, L, lpc, wave_max(L.len), (int)lane);
        return;
    }
    if (wave != 0) return;
    if (max_order > 8) tns_pair_block<12, DIRECT>(coeffs, filters, sink, idx0, idx1, order0, order1, (int)lane);
    else if (max_order > 4) tns_pair_block<8, DIRECT>(coeffs, filters, sink, idx0, idx1, order0, order1, (int)lane);
    else tns_pair_block<4, DIRECT>(coeffs, filters, sink, idx0, idx1, order0, order1, (int)lane);
}

}  // namespace

int launch_aac_joint_stereo(symaccel_ctx *ctx, const AacBandMaps &maps, float *d_coeffs, size_t frames_per_chain,
                            const int32_t *d_pair_chains, const symaccel_aac_js_frame *d_desc, size_t n_pairs, const uint32_t *d_list,
                            size_t n_list) {
    const size_t all = n_pairs * frames_per_chain;
    const size_t grid = d_list ? n_list : all;
    if (all > 0x7fffffffu || grid > 0x7fffffffu || frames_per_chain > 0xffffffffu) return SYMACCEL_ERR_INVALID_ARG;
    if (grid == 0) return SYMACCEL_OK;
    hipLaunchKernelGGL(aac_joint_stereo_kernel, dim3((unsigned)grid), dim3(256), 0, ctx->stream, maps, d_coeffs,
                       (unsigned)frames_per_chain, d_pair_chains, d_desc, d_list, (unsigned)all);
    SYM_GPU(ctx, hipGetLastError());
    return SYMACCEL_OK;
}

int launch_aac_js_consume(symaccel_ctx *ctx, symaccel_aac_js_frame *d_desc, const uint32_t *d_list, size_t n_list, size_t n_pair_frames) {
    if (n_list == 0) return SYMACCEL_OK;
    if (n_list > 0x7fffffffu || n_pair_frames > 0x7fffffffu) return SYMACCEL_ERR_INVALID_ARG;
    hipLaunchKernelGGL(aac_js_consume_kernel, dim3((unsigned)((n_list + 255) / 256)), dim3(256), 0, ctx->stream, d_desc, d_list,
                       (unsigned)n_list, (unsigned)n_pair_frames);
    SYM_GPU(ctx, hipGetLastError());
    return SYMACCEL_OK;
}

int launch_aac_tns(symaccel_ctx *ctx, float *d_coeffs, size_t n_frames, const symaccel_aac_tns_filter *d_filters,
                   size_t n_filters) {
    const size_t blocks = (n_filters + 127) / 128;  // blocks of 128 filters: a workgroup each
    if (n_frames > 0xffffffffu || n_filters > 0xffffffffu || blocks > 0x7fffffffu) return SYMACCEL_ERR_INVALID_ARG;
    if (n_filters == 0) return SYMACCEL_OK;
#define SYM_TNS_LAUNCH(KERNEL) \
    hipLaunchKernelGGL((KERNEL), dim3((unsigned)blocks), dim3(128), 0, ctx->stream, d_coeffs, (unsigned)n_frames, d_filters, (unsigned)n_filters, fsink)
    // below two wavefronts per SIMD the lanes move their lines themselves (tns_walk2_direct); development knob: SYMACCEL_TNS_DIRECT = 0 / 1
    const char *direct_env = std::getenv("SYMACCEL_TNS_DIRECT");  // (read per launch: the tests run both forms in one process)
    const int direct_knob = direct_env ? std::atoi(direct_env) : -1;
    const bool direct = direct_knob >= 0 ? direct_knob != 0 : blocks <= (size_t)8 * (size_t)ctx->n_cus;
    void *sink = nullptr;
    if (direct) SYM_TRY(ctx_sink(ctx, &sink));
    float *fsink = static_cast<float *>(sink);
    if (direct) SYM_TNS_LAUNCH(aac_tns_kernel<true>);
    else SYM_TNS_LAUNCH(aac_tns_kernel<false>);
#undef SYM_TNS_LAUNCH
    SYM_GPU(ctx, hipGetLastError());
    return SYMACCEL_OK;
}

}  // namespace symaccel
