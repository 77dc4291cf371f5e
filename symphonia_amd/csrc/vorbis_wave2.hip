// Vorbis synthesis on the register-pass FFT for EVERY block-size pair with long blocks of up to 2048 samples
// (bs0_exp 6 .. 11, bs1_exp <= 11; the 256 / 2048 pair keeps its own kernel, vorbis_wave.hip): DspChannel::synth
// (symphonia-codec-vorbis/src/dsp.rs:68-145) with Imdct::new(bs / 2) per block size (vorbis/lib.rs:123-124,
// symphonia-core/src/dsp/mdct.rs:67-146).
//
// MI355X mapping: the wavefront-per-chain-segment scheme of vorbis_wave.hip, with fft_wave_multi (imdct_wave.h) as the
// transform: a pass over the 512-point work array is 512 / P independent P-point FFTs (P = bs / 4), so a GROUP is a run of up
// to 2048 / bs consecutive blocks of one size -- one long block of 2048, two of 1024, ... thirty-two short ones of 64 -- that
// is fetched (1 KiB per load instruction), transformed and post-twiddled together.  The group's Imdct output (bs samples per
// block, natural order) lands in the wavefront's LDS work area; `overlap` (dsp.rs:125) is a per-wavefront LDS array with
// exactly the reference's contents, so the three window cases of dsp.rs:85-122 are loops over LDS with 16-byte accesses and
// the stale upper part a short block leaves behind needs no special case.  Blocks after the first of a group lap with their
// predecessor inside the work area.  Packed offsets: counted from the flags at the segment start (vorbis_offsets.h), running sums after that.
// HBM traffic per channel-block: 4 * (n / 2) B in + 4 * (prev_n + n) / 4 B out (+ one halo block per segment).
#include "imdct_wave.h"
#include "vorbis_offsets.h"

namespace symaccel {

namespace {

// wavefronts per workgroup: four; two for long blocks of 4096 / 8192 samples (their overlap arrays are 8 / 16 KiB each)
template <int MAXE1>
constexpr int vw2_waves() { return MAXE1 > 11 ? 2 : 4; }

// out[k .. k+3] = ov[k ..] * win[len - 1 - k ..] + y[k ..] * win[k ..], k = 4 c, c = lane, lane + 64, ...  (dsp.rs:140-144)
__device__ __forceinline__ void ola_span(float *__restrict__ o, const float *ov, const float *y, const float *win, int len, int lane, bool emit) {
    if (!emit) return;
    for (int k = 4 * lane; k < len; k += 256) {
        const float4 a = *reinterpret_cast<const float4 *>(ov + k), b = *reinterpret_cast<const float4 *>(y + k);
        const float4 wf = *reinterpret_cast<const float4 *>(win + k);
        const float4 wr = *reinterpret_cast<const float4 *>(win + len - 4 - k);  // win[len-4-k .. len-1-k], used back to front
        st_stream(reinterpret_cast<float4 *>(o + k),
                  make_float4(a.x * wr.w + b.x * wf.x, a.y * wr.z + b.y * wf.y, a.z * wr.y + b.z * wf.z, a.w * wr.x + b.w * wf.w));
    }
}
__device__ __forceinline__ void copy_span(float *__restrict__ o, const float *src, int len, int lane, bool emit) {
    if (!emit) return;
    for (int k = 4 * lane; k < len; k += 256) st_stream(reinterpret_cast<float4 *>(o + k), *reinterpret_cast<const float4 *>(src + k));
}

// ---- blocks of 4096 and 8192 samples (P = bs / 4 = 1024 or 2048 FFT points): R = P / 512 sub-transforms of 512 points through the
// register passes, the last log2 R stages in registers (the scheme of imdct_big_wave_kernel, imdct_generic.hip).  The block's output
// is four vectors of P samples (left half = vec0 | vec1, right half = vec2 | vec3); they pass through the LDS work area ONE AT A
// TIME: the two left vectors are overlap-added / copied out as they appear (against `overlap`, whose old contents they need), the
// two right vectors then replace overlap[0 .. bs / 2).  Tables (twiddles, windows, W_1024 | W_2048) are read from global memory:
// two wavefronts of this size class keep 50 KiB of LDS busy as it is.
// `keep_below`: overlap[k] for k < keep_below is left as it is (the stale-state rebuild after a short tail; 0 otherwise).
template <int R, int FUSED, class LT>
__device__ __forceinline__ void vorbis_big_block(const float *__restrict__ spec, const float *__restrict__ res, const cpx *__restrict__ tw_g,
                                                 const cpx *__restrict__ w_merge_g, const float *__restrict__ win_long,
                                                 const float *__restrict__ win_short, int flag, int pflag, int bs0, int bs1, float *ldsf,
                                                 float *ovl, const LT &lt, int lane_i, float *__restrict__ o, bool emit, int keep_below,
                                                 const float *dbt) {
    constexpr int P = 512 * R, N = 2 * P;
    // (every global index below is UNSIGNED and 32 bits wide: a uniform base pointer plus a zero-extended lane offset is one SGPR pair +
    // one VGPR + an immediate in the instruction; signed 64-bit indices made the compiler keep a 64-bit address per access live across
    // the whole block loop -- 150 bytes of scratch and no second wavefront per SIMD)
#if defined(__HIP_DEVICE_COMPILE__)
    // (and everything derived from the lane index is a per-block value too, not a loop invariant held across the block loop)
    asm volatile("" : "+v"(lane_i));
#endif
    const unsigned lane = (unsigned)lane_i;
    // The table pointers are the same for every block, so the compiler forms the 64-bit address of every table access ONCE, in front of
    // the block loop, and keeps them all: ~100 address pairs, parked in AGPRs, one wavefront per SIMD.  Passing the (uniform) bases
    // through an empty asm per block makes the addresses per-block values: a few VALU adds per access instead of 200 registers.
#if defined(__HIP_DEVICE_COMPILE__)
    asm volatile("" : "+s"(tw_g), "+s"(w_merge_g), "+s"(win_long), "+s"(win_short));
#endif
    c32 *lds = reinterpret_cast<c32 *>(ldsf);
    const c32 *tw = reinterpret_cast<const c32 *>(tw_g);
    c32 x[R][8];
    {
        // Lines 2 R m .. 2 R m + 2 R - 1 of m = lane + 64 s are the (even, odd) pairs of the FFT inputs R m + c.  Loads s and 7 - s
        // are consumed together (the mirrored odd line of load s sits in load 7 - s of lane 63 - lane), one such pair of loads ahead
        // of the arithmetic: 2 x 2 x R / 2 float4 live instead of all sixteen, and at most 2 R twiddles from global memory at a time.
        const float4 *src = reinterpret_cast<const float4 *>(spec);
        const float4 *rs = reinterpret_cast<const float4 *>(res);
        // (FUSED 2: the table indices travel with the lines -- ya / yb, four bytes per float4 -- and the multiply happens where the
        // pair is consumed, so the loads of the next pair stay in flight during this pair's arithmetic)
        auto load_pair = [&](int sp, float4 (&a)[R / 2], float4 (&b)[R / 2], uint32_t (&ya)[R / 2], uint32_t (&yb)[R / 2]) {
#pragma unroll
            for (int h = 0; h < R / 2; ++h) {
                a[h] = ld_stream((src + ((lane + 64u * (unsigned)sp) * (unsigned)(R / 2) + (unsigned)h)));
                b[h] = ld_stream((src + ((lane + 64u * (unsigned)(7 - sp)) * (unsigned)(R / 2) + (unsigned)h)));
            }
            if constexpr (FUSED == 2) {
                const uint32_t *ry = reinterpret_cast<const uint32_t *>(res);
#pragma unroll
                for (int h = 0; h < R / 2; ++h) {
                    ya[h] = ry[(lane + 64u * (unsigned)sp) * (unsigned)(R / 2) + (unsigned)h];
                    yb[h] = ry[(lane + 64u * (unsigned)(7 - sp)) * (unsigned)(R / 2) + (unsigned)h];
                }
            }
            if constexpr (FUSED == 1) {  // lib.rs:289-291: *f *= r
#pragma unroll
                for (int h = 0; h < R / 2; ++h) {
                    const float4 qa = ld_stream((rs + ((lane + 64u * (unsigned)sp) * (unsigned)(R / 2) + (unsigned)h))), qb = ld_stream((rs + ((lane + 64u * (unsigned)(7 - sp)) * (unsigned)(R / 2) + (unsigned)h)));
                    a[h].x *= qa.x; a[h].y *= qa.y; a[h].z *= qa.z; a[h].w *= qa.w;
                    b[h].x *= qb.x; b[h].y *= qb.y; b[h].z *= qb.z; b[h].w *= qb.w;
                }
            }
        };
        const int mirror = (int)((63u - lane) * 4u);
        float4 a[R / 2], b[R / 2], na[R / 2], nb[R / 2];
        uint32_t ya[R / 2], yb[R / 2], nya[R / 2], nyb[R / 2];
        // (the pre-twiddles of a pair come from global memory as well: requested one pair ahead, with the lines.  Requesting the first
        // pair of the NEXT block from inside this one as well was measured: no gain, 18 registers -- profiles/r04m_vorbis_big_blocks.txt)
        c32 pta[R], ptb[R], npta[R], nptb[R];
        auto load_tw = [&](int sp, c32 (&ta)[R], c32 (&tb_)[R]) {
#pragma unroll
            for (int cc = 0; cc < R; ++cc) {
                ta[cc] = tw[(unsigned)R * (lane + 64u * (unsigned)sp) + (unsigned)cc];
                tb_[cc] = tw[(unsigned)R * (lane + 64u * (unsigned)(7 - sp)) + (unsigned)cc];
            }
        };
        load_pair(0, a, b, ya, yb);
        load_tw(0, pta, ptb);
#pragma unroll
        for (int sp = 0; sp < 4; ++sp) {
            if (sp + 1 < 4) {
                load_pair(sp + 1, na, nb, nya, nyb);
                load_tw(sp + 1, npta, nptb);
            }
            if constexpr (FUSED == 2) {
#pragma unroll
                for (int h = 0; h < R / 2; ++h) {
                    mul_floor_y(a[h], ya[h], dbt);
                    mul_floor_y(b[h], yb[h], dbt);
                }
            }
#pragma unroll
            for (int cc = 0; cc < R; ++cc) {
                const int cm = R - 1 - cc;
                // load s = sp: its mirror is in load 7 - sp (b) of lane 63 - lane; load s = 7 - sp: its mirror is in load sp (a)
                const float odd_b = (cm & 1) ? b[cm >> 1].w : b[cm >> 1].y, odd_a = (cm & 1) ? a[cm >> 1].w : a[cm >> 1].y;
                const float mir_lo = __int_as_float(__builtin_amdgcn_ds_bpermute(mirror, __float_as_int(odd_b)));
                const float mir_hi = __int_as_float(__builtin_amdgcn_ds_bpermute(mirror, __float_as_int(odd_a)));
                const float ev_lo = (cc & 1) ? a[cc >> 1].z : a[cc >> 1].x, ev_hi = (cc & 1) ? b[cc >> 1].z : b[cc >> 1].x;
                x[cc][sp] = pre_twiddle(ev_lo, mir_lo, pta[cc]);
                x[cc][7 - sp] = pre_twiddle(ev_hi, mir_hi, ptb[cc]);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int h = 0; h < R / 2; ++h) {
                a[h] = na[h];
                b[h] = nb[h];
                ya[h] = nya[h];
                yb[h] = nyb[h];
            }
#pragma unroll
            for (int cc = 0; cc < R; ++cc) {
                pta[cc] = npta[cc];
                ptb[cc] = nptb[cc];
            }
        }
    }
    // the R sub-transforms and the last stages (imdct_generic.hip: fft_big_regs, restated here on the global twiddle table)
    auto blk = [](int r) { return R == 2 ? r : ((r & 1) << 1 | (r >> 1)); };
    const c32 *wm = reinterpret_cast<const c32 *>(w_merge_g);
    if constexpr (R == 2) {
        // 4096-sample blocks have the registers to software-pipeline the table loads: the merge twiddles are requested in front of the
        // sub-transforms, the post-twiddles of sub-block 0 in front of the merge arithmetic, those of sub-block 1 in front of the first
        // post-twiddle -- every round trip to the L2 runs under arithmetic instead of being waited for (this kernel runs two wavefronts
        // per SIMD: what one wavefront waits for is not hidden by others).
        c32 wmv[8], pt0[8], pt1[8];
#pragma unroll
        for (int B = 0; B < 8; ++B) wmv[B] = wm[64u * (unsigned)B + lane];
        __builtin_amdgcn_sched_barrier(0);
        fft_wave_multi(x[0], lane_i, lds, lt, 9);
        fft_wave_multi(x[1], lane_i, lds, lt, 9);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int B = 0; B < 8; ++B) pt0[B] = tw[64u * (unsigned)B + lane];
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int B = 0; B < 8; ++B) bfly(x[0][B], x[1][B], c_mul(x[1][B], wmv[B]));
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int B = 0; B < 8; ++B) pt1[B] = tw[512u + 64u * (unsigned)B + lane];
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int B = 0; B < 8; ++B) x[0][B] = post_twiddle(x[0][B], pt0[B]);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int B = 0; B < 8; ++B) x[1][B] = post_twiddle(x[1][B], pt1[B]);
        __builtin_amdgcn_sched_barrier(0);
    } else {
#pragma unroll
        for (int r = 0; r < R; ++r) fft_wave_multi(x[blk(r)], lane_i, lds, lt, 9);
        // (the merge twiddles come from global memory; left alone the scheduler requests all 8 + 16 of them at once and keeps them in
        // 48 more registers next to the transform: four at a time, with a scheduling fence between the groups)
#pragma unroll
        for (int r = 0; r < R; r += 2)
#pragma unroll
            for (int B0 = 0; B0 < 8; B0 += 4) {
#pragma unroll
                for (int B = B0; B < B0 + 4; ++B) bfly(x[blk(r)][B], x[blk(r + 1)][B], c_mul(x[blk(r + 1)][B], wm[64u * (unsigned)B + lane]));
                __builtin_amdgcn_sched_barrier(0);
            }
#pragma unroll
        for (int r = 0; r < 2; ++r)
#pragma unroll
            for (int B0 = 0; B0 < 8; B0 += 4) {
#pragma unroll
                for (int B = B0; B < B0 + 4; ++B) bfly(x[blk(r)][B], x[blk(r + 2)][B], c_mul(x[blk(r + 2)][B], wm[512u + 512u * (unsigned)r + 64u * (unsigned)B + lane]));
                __builtin_amdgcn_sched_barrier(0);
            }
        // post-twiddle once, in place (mdct.rs:104 / 123): the rounds below only pick components
#pragma unroll
        for (int r = 0; r < R; ++r)
#pragma unroll
            for (int B0 = 0; B0 < 8; B0 += 4) {
#pragma unroll
                for (int B = B0; B < B0 + 4; ++B) x[blk(r)][B] = post_twiddle(x[blk(r)][B], tw[512u * (unsigned)r + 64u * (unsigned)B + lane]);
                __builtin_amdgcn_sched_barrier(0);
            }
    }
    const int bs = 4 * P;
    const float *win = (flag && pflag) ? win_long : win_short;  // dsp.rs:83
    const int start = (bs1 - bs0) / 4;
    // long -> short with a short block of this size: the unity part of the old overlap goes out first (dsp.rs:97)
    if (emit && pflag && !flag) {
        for (unsigned k = 4u * lane; k < (unsigned)start; k += 256u) st_stream(reinterpret_cast<float4 *>((o + (k))), *reinterpret_cast<const float4 *>(ovl + k));
    }
    // The common window case (the block and its predecessor have this size), R = 2: a round's window values -- 4 + 4 float4 per
    // lane, from global memory -- are requested in front of the round's scatter and arrive while it runs; the generic loop below
    // waits for every chunk's two loads in turn (eight exposed round trips to the L2 per block).
    const bool same_fast = R == 2 && emit && pflag == flag;
    float4 wfv[R == 2 ? 4 : 1], wrv[R == 2 ? 4 : 1];
#pragma unroll
    for (int round = 0; round < 4; ++round) {
        if constexpr (R == 2) {
            if (same_fast && round < 2) {
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const unsigned k = (unsigned)(round * P) + 4u * lane + 256u * (unsigned)c;
                    wfv[c] = *reinterpret_cast<const float4 *>(win + k);
                    wrv[c] = *reinterpret_cast<const float4 *>(win + ((unsigned)(N - 4) - k));
                }
            }
        }
        // vector `round` -> the work area (mdct.rs:94-137: every FFT bin gives one sample to each of the four vectors)
#pragma unroll
        for (int r = 0; r < R; ++r)
#pragma unroll
            for (int B = 0; B < 8; ++B) {
                const int p = 512 * r + 64 * B + lane_i;
                const c32 val = x[blk(r)][B];
                constexpr int n4 = P / 2;
                float f;
                int at;
                if (p < n4) {
                    f = round == 0 ? -val.y : (round == 1 ? val.y : val.x);
                    at = (round == 0 || round == 2) ? P - 1 - 2 * p : 2 * p;
                } else {
                    const int i = p - n4;
                    f = round == 0 ? -val.x : (round == 1 ? val.x : val.y);
                    at = (round == 0 || round == 2) ? 2 * i : P - 1 - 2 * i;
                }
                ldsf[at] = f;
            }
        wave_sync();
        if (round < 2) {
            if (same_fast) {
                if constexpr (R == 2) {
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        const unsigned kk = 4u * lane + 256u * (unsigned)c, k = (unsigned)(round * P) + kk;
                        const float4 y = *reinterpret_cast<const float4 *>(ldsf + kk), a = *reinterpret_cast<const float4 *>(ovl + k);
                        const float4 wf = wfv[c], wr = wrv[c];
                        st_stream(reinterpret_cast<float4 *>(o + k),
                                  make_float4(a.x * wr.w + y.x * wf.x, a.y * wr.z + y.y * wf.y, a.z * wr.y + y.z * wf.z, a.w * wr.x + y.w * wf.w));
                    }
                }
            } else if (emit) {
                const int k0 = round * P;  // the work area holds left[k0 .. k0 + P)
                for (unsigned kk = 4u * lane; kk < (unsigned)P; kk += 256u) {
                    const unsigned k = (unsigned)k0 + kk;
                    const float4 y = *reinterpret_cast<const float4 *>(ldsf + kk);
                    if (pflag == flag) {  // dsp.rs:85-90 over bs / 2 samples
                        const float4 a = *reinterpret_cast<const float4 *>(ovl + k);
                        const float4 wf = *reinterpret_cast<const float4 *>((win + (k))), wr = *reinterpret_cast<const float4 *>((win + ((unsigned)(N - 4) - k)));
                        st_stream(reinterpret_cast<float4 *>((o + (k))),
                                  make_float4(a.x * wr.w + y.x * wf.x, a.y * wr.z + y.y * wf.y, a.z * wr.y + y.z * wf.z, a.w * wr.x + y.w * wf.w));
                    } else if (pflag) {  // long -> short (dsp.rs:91-106): out[start + k] = overlap[start + k] * ws[len-1-k] + imdct[k] * ws[k]
                        const unsigned len = (unsigned)bs0 / 2u;
                        const float4 a = *reinterpret_cast<const float4 *>(ovl + (unsigned)start + k);
                        const float4 wf = *reinterpret_cast<const float4 *>((win + (k))), wr = *reinterpret_cast<const float4 *>((win + (len - 4u - k)));
                        st_stream(reinterpret_cast<float4 *>((o + ((unsigned)start + k))),
                                  make_float4(a.x * wr.w + y.x * wf.x, a.y * wr.z + y.y * wf.y, a.z * wr.y + y.z * wf.z, a.w * wr.x + y.w * wf.w));
                    } else {  // short -> long (dsp.rs:107-122): imdct[start .. end) laps with overlap[0 .. len), imdct[end ..) is copied
                        const unsigned len = (unsigned)bs0 / 2u, end = (unsigned)start + len;
                        if (k >= (unsigned)start && k < end) {
                            const unsigned j = k - (unsigned)start;
                            const float4 a = *reinterpret_cast<const float4 *>(ovl + j);
                            const float4 wf = *reinterpret_cast<const float4 *>((win + (j))), wr = *reinterpret_cast<const float4 *>((win + (len - 4u - j)));
                            st_stream(reinterpret_cast<float4 *>((o + (j))),
                                      make_float4(a.x * wr.w + y.x * wf.x, a.y * wr.z + y.y * wf.y, a.z * wr.y + y.z * wf.z, a.w * wr.x + y.w * wf.w));
                        } else if (k >= end) {
                            st_stream(reinterpret_cast<float4 *>((o + (len + (k - end)))), y);
                        }
                    }
                }
            }
        } else {
            const int k0 = (round - 2) * P;  // overlap[k0 .. k0 + P) = this vector (dsp.rs:125)
            for (int kk = 4 * lane_i; kk < P; kk += 256)
                if (k0 + kk >= keep_below) *reinterpret_cast<float4 *>(ovl + k0 + kk) = *reinterpret_cast<const float4 *>(ldsf + kk);
        }
        wave_sync();
    }
    (void)bs;
}

// One group's transform with the block size known at compile time: the lines (natural order, in the work area) -> pre-twiddle ->
// 512 / P transforms of P = 2^LOGP points -> post-twiddle -> the blocks' 4 P output samples each, natural order, in the work area.
// (With LOGP a constant every LDS address is a per-lane base plus an immediate and the stage count of the FFT is fixed: ~250 VALU
// instructions per group less than the run-time form -- the kernel is bound by VALU issue, profiles/r04i_vorbis_pairs.txt.)
template <int LOGP, class LT>
__device__ __forceinline__ void vw2_transform(int lane, float *ldsf, const c32 *tw, const LT &lt) {
    constexpr int P = 1 << LOGP, gbits = LOGP - 3, G = 1 << gbits, PAD = multi_pad(LOGP);
    c32 z[8];
    {
        const int T = lane >> gbits, u = lane & (G - 1);
        const float *sT = ldsf + T * (2 * P + PAD);
        const float *fwd = sT + 2 * u, *bwd = sT + (2 * P - 1) - 2 * u;
        const c32 *twu = tw + u;
#pragma unroll
        for (int s = 0; s < 8; ++s) {
            const float2 pr = *reinterpret_cast<const float2 *>(fwd + 2 * (s << gbits));
            z[s] = pre_twiddle(pr.x, bwd[-2 * (s << gbits)], twu[s << gbits]);
        }
    }
    wave_sync();
    fft_wave_multi(z, lane, reinterpret_cast<c32 *>(ldsf), lt, LOGP);
    multi_post_twiddle_ct<LOGP, PAD>(z, lane, tw, ldsf);  // block i of the group: ldsf[i * (bs + PAD) ..)
    wave_sync();
}

// (the run-time form: the big-block instantiations, which hold a 1024- or 2048-point transform in registers as well, have no room
// for six copies of the group routine)
template <class LT>
__device__ __forceinline__ void vw2_transform_rt(int lane, float *ldsf, const c32 *tw, const LT &lt, int logp) {
    const int P = 1 << logp;
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
    fft_wave_multi(z, lane, reinterpret_cast<c32 *>(ldsf), lt, logp);
    multi_post_twiddle(z, lane, logp, tw, ldsf);
    wave_sync();
}

// MAXE1: the largest long-block exponent the instantiation serves.  Up to 1024-sample long blocks the tables and the overlap
// arrays are half the size and three workgroups fit a CU (52 KiB of LDS each, <= 168 VGPRs): the kernel is bound by the latency
// of a group's dependent steps, and a third wavefront per SIMD is worth more than anything else here.  (The fused variant carries
// sixteen more registers -- the prefetched residue lines -- and would spill at 168: it stays at two wavefronts per SIMD.)
// BIG0 (big-block instantiations): 0 = the short blocks are at most 2048 samples (group path), 2 / 4 = they are 4096 / 8192 samples
// themselves.  One instantiation per case so that a kernel holds one copy of each block routine it needs and no other: everything a
// routine keeps loop-invariant (addresses, table pointers) is live across the block loop, and the copies add up.
template <int FUSED, int MAXE1, int BIG0 = 0>
__global__ __launch_bounds__(64 * vw2_waves<MAXE1>(), MAXE1 <= 10 ? 3 : 2) void vorbis_synth_wave2_kernel(
    DevTables tb, int e0, int e1, const cpx *__restrict__ tw_short, const cpx *__restrict__ tw_long,
    const float *__restrict__ win_short, const float *__restrict__ win_long, const float *__restrict__ spectra,
    const float *__restrict__ residue, size_t spec_stride, const uint8_t *__restrict__ flags, const int32_t *__restrict__ prev_flag_in,
    int32_t *__restrict__ prev_flag_out, const float *__restrict__ overlap_in, float *__restrict__ overlap_out,
    float *__restrict__ pcm, size_t pcm_stride, unsigned nb, unsigned seg_len, unsigned segs_per_chain, unsigned n_items) {
    // shared tables: Imdct twiddles and left window halves of both block sizes (bs / 2 floats each)
    constexpr bool kBig = MAXE1 > 11;  // long blocks of 4096 / 8192 samples: tables stay in global memory, see vorbis_big_block
    constexpr int kWaves = vw2_waves<MAXE1>();
    __shared__ __attribute__((aligned(16))) float tabs[kBig ? 4 : (2 << MAXE1)];
    __shared__ __attribute__((aligned(16))) float wave_lds[kWaves][kWaveLds];
    __shared__ __attribute__((aligned(16))) float wave_ovl[kWaves][(1 << MAXE1) / 2];
    const int bs0 = 1 << e0, bs1 = 1 << e1;
    __shared__ float db_lds[FUSED == 2 ? 256 : 1];  // FLOOR1_INVERSE_DB_TABLE (floor.rs:785-825), read where the residue is multiplied
    if constexpr (FUSED == 2) {
        for (int i = (int)threadIdx.x; i < 256; i += 64 * kWaves) db_lds[i] = tb.vorbis_floor1_db[i];  // (in front of the barriers below)
    }
    const float *dbt = db_lds;
    const float *t_twl, *t_wl, *t_tws, *t_ws;
    __shared__ __attribute__((aligned(16))) c32 lane_tab[kBig ? kLaneTabComplex : 1];
    if constexpr (kBig) {
        t_twl = reinterpret_cast<const float *>(tw_long);
        t_wl = win_long;
        t_tws = reinterpret_cast<const float *>(tw_short);
        t_ws = win_short;
        fill_lane_tables_lds(tb, lane_tab, (int)threadIdx.x, 64 * kWaves);
        __syncthreads();
    } else {
        float *l_twl = tabs, *l_wl = tabs + bs1 / 2, *l_tws = tabs + bs1, *l_ws = tabs + bs1 + bs0 / 2;
        for (int i = (int)threadIdx.x; i < bs1 / 2; i += 64 * kWaves) {
            l_twl[i] = reinterpret_cast<const float *>(tw_long)[i];
            l_wl[i] = win_long[i];
            if (i < bs0 / 2) {
                l_tws[i] = reinterpret_cast<const float *>(tw_short)[i];
                l_ws[i] = win_short[i];
            }
        }
        t_twl = l_twl;
        t_wl = l_wl;
        t_tws = l_tws;
        t_ws = l_ws;
        __syncthreads();  // the only workgroup-wide barrier
    }

    // (the wavefront index through readfirstlane: the compiler then knows that the chain, and every pointer derived from it, is
    // wave-uniform and keeps them in SGPRs instead of a 64-bit VGPR pair each)
    const int lane = (int)threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
    const unsigned item = blockIdx.x * kWaves + (unsigned)wave;
    if (item >= n_items) return;
    float *ldsf = wave_lds[wave];
    c32 *lds = reinterpret_cast<c32 *>(ldsf);
    float *ovl = wave_ovl[wave];
    const unsigned chain = item / segs_per_chain, seg = item % segs_per_chain;
    const unsigned b_begin = seg * seg_len, b_end = min(b_begin + seg_len, nb);
    const uint8_t *f = flags + (size_t)chain * nb;
    const float *sp = spectra + (size_t)chain * spec_stride;
    const float *rp = res_at<FUSED>(residue, (size_t)chain * spec_stride);
    float *out = pcm + (size_t)chain * pcm_stride;
    const int pf0 = prev_flag_in[chain];
    // (the big-block instantiations read the FFT's lane twiddles from an LDS copy: 31 VGPRs less in a kernel that holds a whole
    // 2048-point transform in registers)
    auto lt = [&]() {
        if constexpr (kBig) {
            return lane_tables_lds(tb, lane_tab, lane);
        } else {
            LaneTables l;
            load_lane_tables(tb, lane, l);
            return l;
        }
    }();

    // overlap (dsp.rs:125): the caller's state at a chain's start; zero in front of a later segment, whose halo block rebuilds
    // the part the next block reads
    for (int k = 4 * lane; k < bs1 / 2; k += 256)
        *reinterpret_cast<float4 *>(ovl + k) = b_begin == 0 ? *reinterpret_cast<const float4 *>(overlap_in + (size_t)chain * (size_t)(bs1 / 2) + k)
                                                             : make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    bool hi_fresh = b_begin == 0 || e0 == e1;  // overlap[bs0/2 .. bs1/2) is what the reference would hold at this point

    // Block flags as wave-uniform bit masks (bit = 1: long; blocks past b_end read as the opposite of the run they would extend):
    // one coalesced byte load + ballot per 64 blocks instead of a dependent global load per block.
    const long b_first = b_begin == 0 ? 0 : (long)b_begin - 1;  // the halo block rebuilds the overlap only
    auto load_mask = [&](long base) -> unsigned long long {
        const long idx = base + lane;
        return __ballot(idx < (long)b_end && f[idx] != 0);
    };
    long wbase = b_first;
    unsigned long long m0 = load_mask(wbase), m1 = load_mask(wbase + 64);
    const int cap0 = e0 > 11 ? 1 : 2048 >> e0, cap1 = e1 > 11 ? 1 : 2048 >> e1;  // blocks per group (one of 4096 / 8192 samples)
    // A group = a run of consecutive blocks with one flag, at most 2048 / bs of them, never crossing b_end.
    auto group_at = [&](long bb, int &flag_out) -> int {
        if (bb >= (long)b_end) {
            flag_out = 1;
            return 0;
        }
        while (bb - wbase >= 64) {
            m0 = m1;
            wbase += 64;
            m1 = load_mask(wbase + 64);
        }
        const int off = (int)(bb - wbase);
        unsigned long long w = off == 0 ? m0 : ((m0 >> off) | (m1 << (64 - off)));
        flag_out = (int)(w & 1ull);
        if (flag_out) w = ~w;
        int run = w ? __builtin_ctzll(w) : 64;  // blocks with the same flag from bb on (as far as the two masks reach)
        const long left = (long)b_end - bb;
        if ((long)run > left) run = (int)left;
        const int cap = flag_out ? cap1 : cap0;
        return run < cap ? run : cap;
    };

    long b = b_first;
    int flag = 1;
    int glen = group_at(b, flag);
    // flag of the block before b (lib.rs:298: the first block of a stream pairs with itself)
    int pflag = b == 0 ? (pf0 < 0 ? flag : (pf0 ? 1 : 0)) : (f[b - 1] ? 1 : 0);
    const VorbisPackedAt at0 = vorbis_packed_at(f, b, pf0, bs0, bs1, lane);
    uint32_t os_cur = at0.spec, op_cur = at0.pcm;
    float4 v[4], r[4];
    const int lane_of_wave = lane;
    auto fetch = [&](uint32_t off, int fl, int n_blocks) {
        const size_t valid = (size_t)n_blocks << ((fl ? e1 : e0) - 1);
        int lane = lane_of_wave;
        if constexpr (kBig) {
            // (the big-block instantiations: what the group path derives from the lane index -- load addresses, 64-bit -- would be
            // kept as loop invariants across the block routine, which has no registers for them: recomputed per group instead)
#if defined(__HIP_DEVICE_COMPILE__)
            asm volatile("" : "+v"(lane));
#endif
        }
        multi_fetch(sp + off, valid, lane, v);
        if constexpr (FUSED == 1) multi_fetch(rp + off, valid, lane, r);
        if constexpr (FUSED == 2) {
            const uint32_t *ry = reinterpret_cast<const uint32_t *>(res_at<2>(rp, off));
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int i4 = lane + 64 * q;
                const uint32_t yw = (size_t)(4 * i4) < valid ? ry[i4] : 0u;
                if constexpr (kBig) {  // (no prefetch in the big-block instantiations: multiplied here, one float4 at a time)
                    mul_floor_y(v[q], yw, dbt);
                    __builtin_amdgcn_sched_barrier(0);
                } else {
                    r[q].x = __uint_as_float(yw);
                }
            }
        }
    };
    // (the big-block instantiations do not prefetch: 16 / 32 registers that would be live across a 2048-point transform)
    if (!kBig && glen > 0) fetch(os_cur, flag, glen);

    // The walk, then -- at a chain's end, if no long block refreshed overlap[bs0/2 .. bs1/2) in this segment -- ONE more trip through
    // the same loop body for the most recent long block in front of the segment (`rebuild`: nothing is emitted, only the upper part
    // of `overlap` is taken from it): dsp.rs:125 rewrites only the first bs / 2 entries, so that part is still what that block left,
    // never used for PCM but part of the state the reference carries.  With no such block the incoming state is kept.
    bool rebuild = false;
    int keep_below = 0;
    while (true) {
        if (b >= (long)b_end) {
            if (rebuild || b_end != nb || hi_fresh) break;
            long bl = -1;
            for (long base = ((long)b_begin - 1) & ~63l; base >= 0; base -= 64) {
                const long idx = base + lane;
                const unsigned long long m = __ballot(idx < (long)b_begin && f[idx] != 0);
                if (m) {
                    bl = base + 63 - __builtin_clzll(m);
                    break;
                }
            }
            if (bl < 0) {
                for (int k = bs0 / 2 + 4 * lane; k < bs1 / 2; k += 256)
                    *reinterpret_cast<float4 *>(ovl + k) = *reinterpret_cast<const float4 *>(overlap_in + (size_t)chain * (size_t)(bs1 / 2) + k);
                wave_sync();
                break;
            }
            rebuild = true;
            keep_below = bs0 / 2;
            b = bl;
            glen = 1;
            flag = pflag = 1;
            os_cur = vorbis_sizes_before(f, bl, bs0, bs1, lane) / 2u;
            if constexpr (!kBig) fetch(os_cur, 1, 1);
        }
        const int e = flag ? e1 : e0, bs = 1 << e, logp = e - 2, P = 1 << logp;
        const long nb_next = b + glen;
        int flag_next = 1;
        const int glen_next = rebuild ? 0 : group_at(nb_next, flag_next);
        const uint32_t os_next = os_cur + ((uint32_t)glen << (e - 1));
        const c32 *tw = reinterpret_cast<const c32 *>(flag ? t_twl : t_tws);
        hi_fresh = hi_fresh || flag;
        const bool emit = !rebuild && b >= (long)b_begin;
        if constexpr (kBig) {
            if (e > 11) {
                // ---- one block of 4096 / 8192 samples (a group of its own)
                const cpx *twg = flag ? tw_long : tw_short;
                constexpr int R1 = MAXE1 == 12 ? 2 : 4;
                if (BIG0 == 0 || BIG0 == R1 || flag)
                    vorbis_big_block<R1, FUSED>(sp + os_cur, res_at<FUSED>(rp, os_cur), twg, tb.fft_merge + 480, win_long, win_short, flag, pflag, bs0,
                                                bs1, ldsf, ovl, lt, lane, out + op_cur, emit, keep_below, dbt);
                else if constexpr (BIG0 != 0 && BIG0 != R1)
                    vorbis_big_block<BIG0, FUSED>(sp + os_cur, res_at<FUSED>(rp, os_cur), twg, tb.fft_merge + 480, win_long, win_short, flag, pflag, bs0,
                                                  bs1, ldsf, ovl, lt, lane, out + op_cur, emit, keep_below, dbt);
                if (rebuild) break;
                op_cur += (uint32_t)((pflag ? bs1 : bs0) + bs) >> 2;
                os_cur = os_next;
                pflag = flag;
                b = nb_next;
                glen = glen_next;
                flag = flag_next;
                continue;
            }
        }
        if constexpr (BIG0 == 0) {

        // (big-block instantiations: the group path's lane-derived addresses are per-group values, not loop invariants that sit in
        // registers through the block routine -- see `fetch`)
        int lane = lane_of_wave;
#if defined(__HIP_DEVICE_COMPILE__)
        if constexpr (kBig) asm volatile("" : "+v"(lane));
#endif
        // ---- the group's lines -> LDS (natural order), multiplied by the residue on the way (lib.rs:289-291: *f *= r)
        const int pad = kBig ? 0 : multi_pad(logp), ostride = bs + pad;  // (the run-time form of the big-block instantiations is unpadded)
        if constexpr (kBig) fetch(os_cur, flag, glen);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            float4 x = v[q];
            if constexpr (FUSED == 2 && !kBig) mul_floor_y(x, __float_as_uint(r[q].x), dbt);
            if constexpr (FUSED == 1) {
                x.x *= r[q].x;
                x.y *= r[q].y;
                x.z *= r[q].z;
                x.w *= r[q].w;
            }
            // (transforms of up to 64 points: multi_pad(logp) floats between the transforms' lines, see imdct_wave.h)
            const int e4 = 4 * (lane + 64 * q);
            *reinterpret_cast<float4 *>(ldsf + e4 + pad * (e4 >> (logp + 1))) = x;
        }
        wave_sync();
        if (!kBig && glen_next > 0) fetch(os_next, flag_next, glen_next);  // the next group travels while this one is transformed
        if constexpr (kBig) vw2_transform_rt(lane, ldsf, tw, lt, logp);
        else switch (logp) {  // (wave-uniform; an instantiation holds the sizes it can meet)
            case 4: vw2_transform<4>(lane, ldsf, tw, lt); break;
            case 5: vw2_transform<5>(lane, ldsf, tw, lt); break;
            case 6: vw2_transform<6>(lane, ldsf, tw, lt); break;
            case 7: vw2_transform<7>(lane, ldsf, tw, lt); break;
            case 8: vw2_transform<8>(lane, ldsf, tw, lt); break;
            default:
                if constexpr (MAXE1 >= 11) vw2_transform<9>(lane, ldsf, tw, lt);
                break;
        }

        // ---- the group's first block against `overlap` (dsp.rs:85-122)
        {
            float *o = out + op_cur;
            const float *left = ldsf;
            if (pflag == flag) {
                ola_span(o, ovl, left, flag ? t_wl : t_ws, bs / 2, lane, emit);
            } else if (pflag) {  // long -> short: overlap[..start) at unity gain, then bs0 / 2 overlap-added samples
                const int start = (bs1 - bs0) / 4;
                copy_span(o, ovl, start, lane, emit);
                ola_span(o + start, ovl + start, left, t_ws, bs0 / 2, lane, emit);
            } else {             // short -> long: bs0 / 2 overlap-added samples, then imdct[end .. bs1 / 2) at unity gain
                const int start = (bs1 - bs0) / 4, len = bs0 / 2, end = start + len;
                ola_span(o, ovl, left + start, t_ws, len, lane, emit);
                copy_span(o + len, left + end, bs1 / 2 - end, lane, emit);
            }
        }
        // ---- the rest of the run laps with its predecessor inside the work area (equal sizes: dsp.rs:85-90)
        const uint32_t first_len = (uint32_t)((pflag ? bs1 : bs0) + bs) >> 2;
        if (glen > 1) {
            const int half = bs >> 1, per_block = half >> 2;  // float4 chunks per block
            const float *win = flag ? t_wl : t_ws;
            // (all of them are emitted: only a halo block precedes b_begin, and a halo block is its group's first)
            for (int c = lane; c < (glen - 1) * per_block; c += 64) {
                const int i = 1 + (c >> (e - 3)), k = 4 * (c & (per_block - 1));  // (per_block = 1 << (e - 3): no integer division)
                const float *ov = ldsf + (i - 1) * ostride + half, *y = ldsf + i * ostride;
                const float4 a = *reinterpret_cast<const float4 *>(ov + k), bq = *reinterpret_cast<const float4 *>(y + k);
                const float4 wf = *reinterpret_cast<const float4 *>(win + k);
                const float4 wr = *reinterpret_cast<const float4 *>(win + half - 4 - k);
                st_stream(reinterpret_cast<float4 *>(out + op_cur + first_len + (size_t)(i - 1) * half + k),
                          make_float4(a.x * wr.w + bq.x * wf.x, a.y * wr.z + bq.y * wf.y, a.z * wr.y + bq.z * wf.z, a.w * wr.x + bq.w * wf.w));
            }
        }
        wave_sync();  // `overlap` has been read
        // overlap[..bs / 2) = right half of the run's last block (dsp.rs:125); what lies above stays
        {
            const float *right = ldsf + (glen - 1) * ostride + (bs >> 1);
            for (int k = keep_below + 4 * lane; k < bs / 2; k += 256) *reinterpret_cast<float4 *>(ovl + k) = *reinterpret_cast<const float4 *>(right + k);
        }
        wave_sync();  // the work area is overwritten by the next group
        if (rebuild) break;
        op_cur += first_len + (uint32_t)(glen - 1) * (uint32_t)(bs >> 1);
        os_cur = os_next;
        pflag = flag;
        b = nb_next;
        glen = glen_next;
        flag = flag_next;
        }  // BIG0 == 0
    }

    if (b_end == nb) {
        for (int k = 4 * lane; k < bs1 / 2; k += 256)
            *reinterpret_cast<float4 *>(overlap_out + (size_t)chain * (size_t)(bs1 / 2) + k) = *reinterpret_cast<const float4 *>(ovl + k);
        if (lane == 0) prev_flag_out[chain] = f[nb - 1] ? 1 : 0;  // lib.rs:328
    }
}

}  // namespace

int launch_vorbis_wave2(symaccel_ctx *ctx, int bs0_exp, int bs1_exp, const cpx *tw_short, const cpx *tw_long, const float *win_short,
                        const float *win_long, const float *d_spectra, const float *d_residue, size_t spec_stride,
                        const uint8_t *d_block_flag, const int32_t *d_prev_in, int32_t *d_prev_out, const float *d_overlap_in,
                        float *d_overlap_out, float *d_pcm, size_t pcm_stride, size_t n_chains, unsigned nb, unsigned seg, int floor_mode) {
    const size_t segs = (nb + seg - 1) / seg;
    const size_t items = n_chains * segs;
    const size_t kWaves = bs1_exp > 11 ? 2 : 4;
    const size_t grid = (items + kWaves - 1) / kWaves;
    if (items > 0xffffffffu || grid > 0x7fffffffu) return SYMACCEL_ERR_INVALID_ARG;
#define SYM_VW2_LAUNCH3(FUSED, MAXE1, BIG0)                                                                                                  \
    hipLaunchKernelGGL((vorbis_synth_wave2_kernel<FUSED, MAXE1, BIG0>), dim3((unsigned)grid), dim3(64 * kWaves), 0, ctx->stream, ctx->dev, bs0_exp, \
                       bs1_exp, tw_short, tw_long, win_short, win_long, d_spectra, d_residue, spec_stride, d_block_flag, d_prev_in, d_prev_out, \
                       d_overlap_in, d_overlap_out, d_pcm, pcm_stride, nb, seg, (unsigned)segs, (unsigned)items)
#define SYM_VW2_LAUNCH(FUSED, MAXE1)                                                                                                        \
    hipLaunchKernelGGL((vorbis_synth_wave2_kernel<FUSED, MAXE1>), dim3((unsigned)grid), dim3(64 * kWaves), 0, ctx->stream, ctx->dev, bs0_exp, \
                       bs1_exp, tw_short, tw_long, win_short, win_long, d_spectra, d_residue, spec_stride, d_block_flag, d_prev_in, d_prev_out, \
                       d_overlap_in, d_overlap_out, d_pcm, pcm_stride, nb, seg, (unsigned)segs, (unsigned)items)
    const int mode = floor_mode == 2 ? 2 : (d_residue ? 1 : 0);
#define SYM_VW2_MODES(LAUNCH) do { if (mode == 2) { LAUNCH(2); } else if (mode == 1) { LAUNCH(1); } else { LAUNCH(0); } } while (0)
    if (bs1_exp <= 10) {
#define SYM_VW2_10(FUSED) SYM_VW2_LAUNCH(FUSED, 10)
        SYM_VW2_MODES(SYM_VW2_10);
#undef SYM_VW2_10
    } else if (bs1_exp == 11) {
#define SYM_VW2_11(FUSED) SYM_VW2_LAUNCH(FUSED, 11)
        SYM_VW2_MODES(SYM_VW2_11);
#undef SYM_VW2_11
    } else {
        // the big-block instantiations: by long size and by whether the short size is big as well.  Long blocks of 8192 samples belong
        // to vorbis_synth_wg_kernel (vorbis_wg.hip) in the product build: the one-wavefront-per-block instantiations for them -- 256
        // VGPRs + AGPR spills, one wavefront per SIMD, 0.13 of the roofline for the 4096 / 8192 pair -- exist in the SYM_VORBIS_WG = 0
        // development build only (the A/B partner).
        const int big0 = bs0_exp <= 11 ? 0 : (bs0_exp == 12 ? 2 : 4);
#if SYM_VORBIS_WG
#define SYM_VW2_13(FUSED) return SYMACCEL_ERR_UNSUPPORTED
#else
#define SYM_VW2_13(FUSED)                                        \
    do {                                                         \
        if (big0 == 0) SYM_VW2_LAUNCH3(FUSED, 13, 0);            \
        else if (big0 == 2) SYM_VW2_LAUNCH3(FUSED, 13, 2);       \
        else SYM_VW2_LAUNCH3(FUSED, 13, 4);                      \
    } while (0)
#endif
#define SYM_VW2_BIG(FUSED)                                                                      \
    do {                                                                                        \
        if (bs1_exp == 12) {                                                                    \
            if (big0 == 0) SYM_VW2_LAUNCH3(FUSED, 12, 0); else SYM_VW2_LAUNCH3(FUSED, 12, 2);   \
        } else SYM_VW2_13(FUSED);                                                               \
    } while (0)
        SYM_VW2_MODES(SYM_VW2_BIG);
#undef SYM_VW2_BIG
#undef SYM_VW2_13
    }
#undef SYM_VW2_MODES
#undef SYM_VW2_LAUNCH
#undef SYM_VW2_LAUNCH3
    SYM_GPU(ctx, hipGetLastError());
    return SYMACCEL_OK;
}

}  // namespace symaccel
