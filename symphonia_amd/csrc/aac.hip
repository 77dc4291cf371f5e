// AAC-LC synthesis: Dsp::synth (symphonia-codec-aac/src/aac/dsp.rs:57-158) = Imdct (1x1024 or
// 8x128, symphonia-core/src/dsp/mdct.rs:67-146) + window + overlap-add, batched over chains.
//
// MI355X mapping (DESIGN.md "aac_synth"):
//  * a WORKGROUP of four wavefronts walks a SEGMENT of consecutive frames of one chain, four consecutive
//    frames per step (wave j: frame t0 + 4 i + j: 16 KiB contiguous in, 16 KiB out); a frame's delay
//    line depends on that frame's input alone, so the four transforms of a step are independent: each
//    wavefront parks the delay line it produces in an LDS slot, the workgroup meets at an LDS-only
//    barrier, each wavefront adds its predecessor's slot and stores the PCM, a second barrier frees
//    the slots -- two barriers per step and channel.  HBM traffic is the algorithmic 4 KiB in +
//    4 KiB out per channel-frame (+ one halo frame re-read per segment: segments start by
//    recomputing the previous frame's delay).  (The wavefront walk this replaced -- one wavefront per
//    segment, delay line in 16 VGPRs, no barriers after start-up -- is csrc/experiments/aac_wave_walk.h.)
//  * the wavefronts share the LDS-resident tables (Imdct twiddles, KBD and sine windows, 13 KiB) and
//    order their own LDS traffic inside a transform with wave-local fences;
//  * the 512-point complex FFT is three radix-8 register passes (stages 1-3, 4-6, 7-9 of the
//    reference's radix-2 DIT graph -- identical operands and roundings, only the schedule differs)
//    with conflict-free LDS transposes between them whose addresses are lane base + immediate;
//  * spectrum loads are 8 B/lane coalesced; the bit-reversal is absorbed in the lane mapping
//    (lane m owns z[m + 64 s]) so the input never round-trips through LDS; the mirrored odd
//    lines come from lane 63-m via ds_bpermute;
//  * the post-twiddle emits each lane's outputs as four contiguous float4 per frame
//    (slots m2 and 255-m2), the same positions for dst and for the next delay line, so PCM
//    stores are 1 KiB-coalesced and the overlap-add is lane-local.
// Roofline: HBM-bound, 8192 B algorithmic per channel-frame, ~27 kflop (no FMA) -> 3.3 flop/B.
#include <type_traits>

#include "imdct_wave.h"

namespace symaccel {

namespace {

// Build variants (tuning knob SYM_AAC_VARIANT, see build.py):
//   0  four wavefronts per workgroup, lane twiddles in 31 VGPRs, two wavefronts per SIMD;
//   1  six wavefronts per workgroup, lane twiddles read from a 4 KiB LDS copy, the halo frame peeled out of the
//      frame loop (its PCM stores and their address arithmetic disappear at compile time) -- sized for THREE
//      wavefronts per SIMD (<= 168 VGPRs, 2 workgroups x 71.6 KiB of LDS per CU).
#ifndef SYM_AAC_VARIANT
#define SYM_AAC_VARIANT 0
#endif
#if SYM_AAC_VARIANT == 1
constexpr int kWaves = 6;
#define SYM_AAC_WAVES_PER_SIMD 3
#else
constexpr int kWaves = 4;                 // wavefronts per workgroup
#endif
constexpr int kTabTw = 0;                 // shared LDS tables: Imdct twiddles, 512 complex
constexpr int kTabKbd = 1024;             //   KBD long window, 1024 f32
constexpr int kTabSine = 2048;            //   sine long window, 1024 f32
constexpr int kTabKbdShort = 3072;        //   KBD short window, 128 f32
constexpr int kTabSineShort = 3200;       //   sine short window, 128 f32
constexpr int kTabFloats = 3328;

enum : int { ONLY_LONG = 0, LONG_START = 1, EIGHT_SHORT = 2, LONG_STOP = 3 };
constexpr int kP0 = 512 - 64;  // SHORT_WIN_POINT0 (dsp.rs:19)
constexpr int kP1 = 512 + 64;  // SHORT_WIN_POINT1 (dsp.rs:20)

// pcm_short[q0 .. q0+3] of dsp.rs:86-101 (q0 a multiple of 4) with the reference's operation order, including the
// `0.0 +` of the `+=` onto the zero-filled buffer for windows > 0.  H = the eight half-stored short transforms.
__device__ __forceinline__ void pcm_short4(const float *H, int q0, const float *short_win, const float *prev_short_win,
                                           float (&acc)[4]) {
    const int w = q0 >> 7, i0 = q0 & 127;
    bool have = false;
    if (w >= 1) {  // right half of window w-1: src[128 + i] * short_win[127 - i]
        float a[4];
        ys4(H, w - 1, 128 + i0, a);
        const float4 wr = *reinterpret_cast<const float4 *>(short_win + 124 - i0);  // short_win[124-i0 .. 127-i0], reversed
        a[0] *= wr.w; a[1] *= wr.z; a[2] *= wr.y; a[3] *= wr.x;
#pragma unroll
        for (int q = 0; q < 4; ++q) acc[q] = (w - 1 == 0) ? a[q] : (0.0f + a[q]);
        have = true;
    }
    if (w <= 7) {  // left half of window w
        float bq[4];
        ys4(H, w, i0, bq);
        const float4 wf = *reinterpret_cast<const float4 *>((w == 0 ? prev_short_win : short_win) + i0);
        bq[0] *= wf.x; bq[1] *= wf.y; bq[2] *= wf.z; bq[3] *= wf.w;
#pragma unroll
        for (int q = 0; q < 4; ++q) acc[q] = have ? (acc[q] + bq[q]) : bq[q];
    }
}

// Effective output window of a LONG_STOP frame for the float4 at sample j0 (dsp.rs:118-127): j < 448 -> dst = delay
// (flagged by the caller), 448..575 -> prev_short_win[j-448], >= 576 -> delay + pcm (pcm * 1.0 is exact).
__device__ __forceinline__ void stop_window4(const float *psw, int j0, float *w) {
    float4 v = make_float4(1.0f, 1.0f, 1.0f, 1.0f);
    if (j0 >= kP0 && j0 < kP1) v = *reinterpret_cast<const float4 *>(psw + j0 - kP0);
    w[0] = v.x; w[1] = v.y; w[2] = v.z; w[3] = v.w;
}
// Effective delay window of a LONG_START frame for the float4 at sample j0 (dsp.rs:146-155), multiplying
// pcm[1024 + j]: j < 448 -> copy (x * 1.0 exact), 448..575 -> short_win[127 - (j-448)], >= 576 -> literal 0.0 (caller).
__device__ __forceinline__ void start_window4(const float *sw, int j0, float *w) {
    float4 v = make_float4(1.0f, 1.0f, 1.0f, 1.0f);
    if (j0 >= kP0 && j0 < kP1) {
        const float4 r = *reinterpret_cast<const float4 *>(sw + 124 - (j0 - kP0));  // sw[127-(j-448)], j = j0..j0+3
        v = make_float4(r.w, r.z, r.y, r.x);
    }
    w[0] = v.x; w[1] = v.y; w[2] = v.z; w[3] = v.w;
}

#if defined(SYM_AAC_WAVES_PER_SIMD) && !defined(SYM_AAC_MIN_WAVES)
#define SYM_AAC_MIN_WAVES SYM_AAC_WAVES_PER_SIMD
#endif
#ifndef SYM_AAC_MIN_WAVES
#define SYM_AAC_MIN_WAVES 2  // wavefronts per SIMD the register allocation must allow (build-time tuning knob)
#endif
#ifndef SYM_AAC_PREFETCH
#define SYM_AAC_PREFETCH 1  // frames of spectral lines in flight ahead of the one being transformed (1 or 2)
#endif
// Which walk (build knob): 0 = the wavefront walk, 1 = the workgroup walk (two LDS-only barriers per step; the product).  A third
// form -- dedicated delay slots and point-to-point LDS flags instead of barriers -- measured the same as 1 and was removed
// (profiles/r03f_aac_quad2_ab.txt).  Which of the two is faster depends on the clock the board runs at: during the ~25 ms after
// an idle period (1.4 GHz under this kernel) the arithmetic binds and the wavefront walk is 4-7 % ahead; at the sustained clock
// the access pattern binds and the workgroup walk is 5-9 % ahead (0.210 against 0.222-0.234 ms, profiles/r03y_aac_sustained.txt).
#ifndef SYM_AAC_QUAD
#define SYM_AAC_QUAD 1
#endif
__device__ __forceinline__ float2 ld_line(const float2 *p) { return ld_stream(p); }
// PCM of one output slot (the two float4 of store_slot), streamed
__device__ __forceinline__ void st_slot(float *frame, int m2, const float (&v)[8]) {
    float4 *o4 = reinterpret_cast<float4 *>(frame);
    st_stream(o4 + m2, make_float4(v[0], v[1], v[2], v[3]));
    st_stream(o4 + 255 - m2, make_float4(v[4], v[5], v[6], v[7]));
}


#if !SYM_AAC_QUAD
#include "experiments/aac_wave_walk.h"  // development only: the wavefront walk, the A/B partner of the kernel below
#endif



// ---- the workgroup walk -------------------------------------------------------------------------------------------------
// The four wavefronts of a workgroup take FOUR CONSECUTIVE frames of one chain per step (wave j: frame t0 + 4 i + j), so a
// workgroup reads and writes 16 KiB of contiguous memory per step instead of four 4 KiB pieces 256 KiB apart.  Measured with
// the copy probe of the same shapes (profiles/r03c_probe_patterns.txt): 5.3 TB/s against 4.6 TB/s, and the ablation of
// profiles/r03b_aac_ablate_ab.txt says the wavefront walk above sits exactly on the slower of the two.
// What makes it possible: the delay line a frame leaves behind depends on THAT frame's input alone (the second half of its
// IMDCT output, windowed: dsp.rs:132-157), so the four transforms of a step are independent; only the overlap-add needs
// the neighbour's result.  Every wavefront writes its frame's delay line into an LDS slot (natural sample order), the
// workgroup meets at a barrier, every wavefront adds its predecessor's slot to its own first half and stores the PCM.
// Wave 0's predecessor is wave 3 of the step before: that slot is a double-buffered 4 KiB carry buffer of its own (the
// other three live in the owners' FFT work areas, which the next transform overwrites -- hence the second barrier).
// A segment starts with a step in which only wave 3 runs: the halo frame (or the chain's incoming delay) fills the carry.
constexpr int kSlotOff = kShortRowsEnd;  // floats: the slot lies behind the eight short-window rows, inside the FFT work area
static_assert(kSlotOff + 1024 <= kWaveLds, "delay slot must fit the per-wave LDS behind the short-window rows");

// JS (symaccel_aac_synth_js_*): joint-stereo decoding (cpe.rs:110-157) happens as the lines are loaded, with a channel PAIR per
// workgroup.  Wave j still owns frame t0 + 4 i + j of a step, but of BOTH channels: it loads the left and the right channel's
// lines (each line of the batch is read exactly once, and every channel keeps its 16 KiB-contiguous steps), the frame's stereo
// map in line-aligned form (`mode4` / `scale4`: one entry per group of four lines, what aac_js_expand_kernel makes of the
// 644-byte descriptor), decodes in registers what aac_joint_stereo_kernel would have written back --
//     mid/side (cpe.rs:144-154):  left = m + s, right = m - s;     intensity (cpe.rs:124-139):  right = scale * left --
// and then runs the two phases of the walk for the left channel and again for the right one (own carry buffers, the same two
// barriers per channel).  The decoded spectra never go to HBM: 4 B per line read (+ 5 B per four lines of map) and 4 B per
// sample written, instead of a read-modify-write pass over both channels in front of the synthesis.  Frames with TNS still
// take the separate kernels (joint stereo, then TNS) and an all-zero map here.  Chains that belong to no pair are taken by
// the plain instantiation, launched over all chains: a workgroup whose chain is in `chain` (the index the expansion fills)
// returns at once.  The lane-index trick of the Vorbis big-block kernels (an empty asm per step) keeps the pair instantiation
// inside 256 registers.
struct AacJsArgs {
    const int2 *chain;                  // per chain: .x = the partner chain (-1: not part of a pair) -- the plain kernel's skip list
    const int32_t *pair_chains;         // [pair][2]: the left and the right chain
    const symaccel_aac_js_frame *desc;  // [pair][frame]
    AacBandMaps maps;                   // line / 4 -> scale-factor band (long / short windows)
    unsigned n_chains;                  // chains of the batch: the bound of pair_chains[]
};

template <bool JS>
__global__ __launch_bounds__(256, SYM_AAC_MIN_WAVES) void aac_synth_quad_kernel(
    DevTables tb, const float *__restrict__ coeffs, const uint8_t *__restrict__ side,
    const float *__restrict__ delay_in, float *__restrict__ delay_out, float *__restrict__ pcm,
    unsigned frames_per_chain, unsigned seg_steps, unsigned segs_per_chain, AacJsArgs js) {
    __shared__ __attribute__((aligned(16))) float tabs[kTabFloats];
    __shared__ __attribute__((aligned(16))) float wave_lds[4][kWaveLds];
    constexpr int CH = JS ? 2 : 1;  // channels a workgroup walks
    __shared__ __attribute__((aligned(16))) float carry[CH][2][1024];
    if constexpr (!JS) {
        if (js.chain && js.chain[blockIdx.x / segs_per_chain].x >= 0) return;  // a paired chain: the pair instantiation's
    }
    for (int i = (int)threadIdx.x; i < 1024; i += 256) {
        tabs[kTabTw + i] = reinterpret_cast<const float *>(tb.aac_tw_long)[i];
        tabs[kTabKbd + i] = tb.aac_kbd_long[i];
        tabs[kTabSine + i] = tb.aac_sine_long[i];
        if (i < 128) {
            tabs[kTabKbdShort + i] = tb.aac_kbd_short[i];
            tabs[kTabSineShort + i] = tb.aac_sine_short[i];
        }
    }
    const int lane = (int)threadIdx.x & 63, wave = (int)threadIdx.x >> 6;
    float *ldsf = wave_lds[wave];
    c32 *lds = reinterpret_cast<c32 *>(ldsf);
    const c32 *tw = reinterpret_cast<const c32 *>(tabs + kTabTw);
    const unsigned unit = blockIdx.x / segs_per_chain, seg = blockIdx.x % segs_per_chain;
    // the chain of channel 0 (the only one of the plain instantiation) and of channel 1
    const unsigned chain = JS ? (unsigned)js.pair_chains[2 * unit] : unit, chain1 = JS ? (unsigned)js.pair_chains[2 * unit + 1] : unit;
    if constexpr (JS) {
        if (chain >= js.n_chains || chain1 >= js.n_chains || chain == chain1) return;  // (device-resident indices are bounded, not trusted)
    }
    const long t_begin = (long)seg * seg_steps * 4;
    const long t_end = t_begin + (long)seg_steps * 4 < (long)frames_per_chain ? t_begin + (long)seg_steps * 4 : (long)frames_per_chain;
    const long n_steps = (t_end - t_begin + 3) / 4;
    const size_t chain_base0 = (size_t)chain * frames_per_chain, chain_base1 = (size_t)chain1 * frames_per_chain;
    LaneTables lt;
    load_lane_tables(tb, lane, lt);

    // the carry in front of the segment: the caller's delay line at a chain's start, else what the halo frame leaves (step -1)
    if (t_begin == 0 && wave == 3) {
#pragma unroll
        for (int ch = 0; ch < CH; ++ch) {
            const float4 *src = reinterpret_cast<const float4 *>(delay_in + (size_t)(ch ? chain1 : chain) * 1024);
#pragma unroll
            for (int q = 0; q < 4; ++q) reinterpret_cast<float4 *>(carry[ch][1])[lane + 64 * q] = src[lane + 64 * q];
        }
    }
    // `line` always holds the lines of frame t_loaded: the halo frame for wave 3 of a later segment, else the frame of step 0
    long t_loaded = (wave == 3 && t_begin > 0) ? t_begin - 1 : t_begin + wave;
    float2 line[8];
    unsigned sb_next = 0;
    // JS: channel 1's lines and side byte, and the stereo map of frame t_loaded
    float2 line1[8];
    // The frame's 644-byte stereo descriptor travels with the lines as three dwords per lane (161 in all) and is laid down in the
    // wavefront's LDS area at the top of the step (the area is idle there), where the lanes look up mode[] and scale[] of their
    // eight line pairs: which (window group, band) a pair belongs to depends on the frame's window sequence, and reading the map
    // from LDS at decode time keeps that dependency out of the loads.
    uint32_t dq[3] = {0u, 0u, 0u};
    unsigned sb1_next = 0;
    uint32_t sfb_long[2] = {0u, 0u};  // the band of the lane's eight line pairs in a long window, a byte each
    unsigned sfb_short = 0;           //   ... in a short window (the same for all eight: pair s lies in window s)
    auto fetch1 = [&](long tf, int ln) {  // the right channel's lines + the frame's stereo descriptor
        if constexpr (JS) {
            const float2 *ps = reinterpret_cast<const float2 *>(coeffs + (chain_base1 + (size_t)tf) * 1024);
            const uint32_t *d = reinterpret_cast<const uint32_t *>(js.desc + (size_t)unit * frames_per_chain + (size_t)tf);
#pragma unroll
            for (int s = 0; s < 8; ++s) line1[s] = ld_line(ps + ln + 64 * s);
            dq[0] = d[ln];
            dq[1] = d[ln + 64];
            dq[2] = ln < 33 ? d[ln + 128] : 0u;
            sb1_next = side[chain_base1 + (size_t)tf];
        }
    };
    if constexpr (JS) {
#pragma unroll
        for (int s = 0; s < 8; ++s) sfb_long[s >> 2] |= (uint32_t)js.maps.long4[(lane >> 1) + 32 * s] << (8 * (s & 3));
        sfb_short = js.maps.short4[(lane >> 1) & 31];
    }
    if (t_loaded < t_end) {
        const float2 *src = reinterpret_cast<const float2 *>(coeffs + (chain_base0 + (size_t)t_loaded) * 1024);
#pragma unroll
        for (int s = 0; s < 8; ++s) line[s] = ld_line(src + lane + 64 * s);
        sb_next = side[chain_base0 + (size_t)t_loaded];
        fetch1(t_loaded, lane);
    } else {
        t_loaded = -100;
#pragma unroll
        for (int s = 0; s < 8; ++s) line[s] = make_float2(0.0f, 0.0f);
        if constexpr (JS) {
#pragma unroll
            for (int s = 0; s < 8; ++s) line1[s] = make_float2(0.0f, 0.0f);
        }
    }
    __syncthreads();  // tables and the carry are in place

    long t = t_begin - 4 + wave;  // this wavefront's frame of step i
    const int lane_of_wave = lane;
    for (long i = -1; i < n_steps; ++i, t += 4) {
        int lane = lane_of_wave;
        if constexpr (JS) {
#if defined(__HIP_DEVICE_COMPILE__)
            asm volatile("" : "+v"(lane));  // (what the step derives from the lane index is recomputed per step, not held in ~70 registers)
#endif
        }
        const bool active = t == t_loaded;  // (step -1: only a wave 3 with a halo frame)
        const bool emit = i >= 0;
        const unsigned sb0 = sb_next, sb1 = sb1_next;
        if constexpr (JS) {
            if (active) {  // both channels of the frame, decoded in registers (cpe.rs:110-157)
                uint32_t *dl = reinterpret_cast<uint32_t *>(ldsf);  // the descriptor: num_windows | max_sfb << 8 | .. ; mode[128]; scale[128]
                dl[lane] = dq[0];
                dl[lane + 64] = dq[1];
                if (lane < 33) dl[lane + 128] = dq[2];
                wave_sync();
                const unsigned hdr = dl[0];
                const bool one = (hdr & 255u) == 1u;
                const int max_sfb = (int)((hdr >> 8) & 255u);
                const uint8_t *mode = reinterpret_cast<const uint8_t *>(dl + 1);
                const float *scale = reinterpret_cast<const float *>(dl + 33);
#pragma unroll
                for (int s = 0; s < 8; ++s) {
                    const float2 l = line[s], r = line1[s];
                    // (window group, band) of the pair; bands at or above max_sfb are not coded (cpe.rs:113-122, aac_joint_stereo_kernel)
                    const int sfb = one ? (int)((sfb_long[s >> 2] >> (8 * (s & 3))) & 255u) : (int)sfb_short;
                    const int slot = one ? sfb : 16 * s + sfb;
                    const bool coded = sfb < max_sfb && slot < 128;
                    const int sl = slot < 127 ? slot : 127;
                    const unsigned m = mode[sl];
                    const float sc = scale[sl];
                    const bool ms = coded && m == SYMACCEL_AAC_JS_MS, is = coded && m == SYMACCEL_AAC_JS_INTENSITY;
                    line[s] = ms ? make_float2(l.x + r.x, l.y + r.y) : l;
                    line1[s] = ms ? make_float2(l.x - r.x, l.y - r.y) : (is ? make_float2(sc * l.x, sc * l.y) : r);
                }
                wave_sync();  // (the area is the transform's again)
            }
        }
        const long tn = t < t_begin ? t_begin + 3 : t + 4;  // (after the halo frame t_begin - 1 wave 3 continues with t_begin + 3)
#pragma unroll 1
        for (int ch = 0; ch < CH; ++ch) {
        if constexpr (JS) {
#if defined(__HIP_DEVICE_COMPILE__)
            asm volatile("" : "+v"(lane));  // (per channel too)
#endif
        }
        const unsigned sb = ch ? sb1 : sb0;
        const size_t chain_base = ch ? chain_base1 : chain_base0;
        const int seq = (int)(sb & 3u);
        const int shape = (int)((sb >> 2) & 1u), prev_shape = (int)((sb >> 3) & 1u);
        float *my_slot = wave == 3 ? carry[ch][i & 1] : ldsf + kSlotOff;
        const float *prev_slot = wave == 0 ? carry[ch][(i + 1) & 1] : wave_lds[wave - 1] + kSlotOff;
        float xw[2][8];  // long frames: first half of the IMDCT output times its window, waiting for the predecessor's delay
        auto ln_at = [&](int s) -> float2 {
            if constexpr (JS) return ch ? line1[s] : line[s];
            else return line[s];
        };

        // ---------------- phase 1: transform this wavefront's frame, publish the delay line it leaves behind
        if (active) {
            c32 z[8];
            if (seq != EIGHT_SHORT) {
                const int mirror = (63 - lane) * 4;
#pragma unroll
                for (int s = 0; s < 8; ++s) {
                    const float mirrored = __int_as_float(__builtin_amdgcn_ds_bpermute(mirror, __float_as_int(ln_at(7 - s).y)));
                    z[s] = pre_twiddle(ln_at(s).x, mirrored, tw[lane + 64 * s]);
                }
            } else {
#pragma unroll
                for (int s = 0; s < 8; ++s) {  // the short transform indexes lines per 128-line window: stage in LDS
                    ldsf[short_row(s) + 2 * lane] = ln_at(s).x;
                    ldsf[short_row(s) + 2 * lane + 1] = ln_at(s).y;
                }
                wave_sync();
            }
            // this wavefront's frame of the next step lands while this one is transformed (pairs: the left channel's lines during the
            // left channel's transform, the right channel's and the descriptor during the right channel's -- `line1` holds the frame
            // being decoded until then; requesting all of it during the left channel's transform, through a second register set,
            // measured slower: 0.283 against 0.257 ms)
            if (tn < t_end) {
                if (ch == 0) {
                    sb_next = side[chain_base0 + (size_t)tn];
                    const float2 *src = reinterpret_cast<const float2 *>(coeffs + (chain_base0 + (size_t)tn) * 1024);
#pragma unroll
                    for (int s = 0; s < 8; ++s) line[s] = ld_line(src + lane + 64 * s);
                }
                if (ch == CH - 1) fetch1(tn, lane);
            }
            if (seq != EIGHT_SHORT) {
                fft512_wave(z, lane, lds, lt);
                const float *wprev = tabs + (prev_shape ? kTabKbd : kTabSine);  // prev_long_win (dsp.rs:71-74)
                const float *wcur = tabs + (shape ? kTabKbd : kTabSine);        // long_win (dsp.rs:66-69)
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const int m2 = lane + 64 * h;
                    float x[8], x2[8];
                    post_slot(lds, tw, m2, x, x2);
                    float wo[8];
                    if (seq == LONG_STOP) {
                        const float *psw = tabs + (prev_shape ? kTabKbdShort : kTabSineShort);
                        stop_window4(psw, 4 * m2, wo);
                        stop_window4(psw, 1020 - 4 * m2, wo + 4);
                    } else {
                        load_slot(wprev, m2, wo);
                    }
#pragma unroll
                    for (int q = 0; q < 8; ++q) xw[h][q] = x[q] * wo[q];
                    // the delay this frame leaves (dsp.rs:132-157): pcm[1024 + j] * long_win[1023 - j], a short-window slope,
                    // or literal zero
                    float wd[8], nd[8];
                    if (seq == LONG_START) {
                        const float *sw = tabs + (shape ? kTabKbdShort : kTabSineShort);
                        start_window4(sw, 4 * m2, wd);
                        start_window4(sw, 1020 - 4 * m2, wd + 4);
                    } else {
                        float wr[8];
                        load_slot(wcur, m2, wr);
#pragma unroll
                        for (int q = 0; q < 8; ++q) wd[q] = wr[7 - q];
                    }
#pragma unroll
                    for (int q = 0; q < 8; ++q) {
                        const int j = q < 4 ? 4 * m2 + q : 1020 - 4 * m2 + (q - 4);
                        const float v = x2[q] * wd[q];
                        nd[q] = (seq == LONG_START && j >= kP1) ? 0.0f : v;
                    }
                    store_slot(my_slot, m2, nd);
                }
            } else {
                imdct_short_wave(lane, ldsf, tb.aac_tw_short, lt);  // H[w] = ldsf[short_row(w) ..]
                const float *sw = tabs + (shape ? kTabKbdShort : kTabSineShort);
                const float *psw = tabs + (prev_shape ? kTabKbdShort : kTabSineShort);
#pragma unroll 1
                for (int e = 0; e < 4; ++e) {  // dsp.rs:138-145
                    const int j0 = 4 * lane + 256 * e;
                    float nd[4] = {0.0f, 0.0f, 0.0f, 0.0f};
                    if (j0 < kP1) pcm_short4(ldsf, j0 + kP1, sw, psw, nd);
                    *reinterpret_cast<float4 *>(my_slot + j0) = make_float4(nd[0], nd[1], nd[2], nd[3]);
                }
            }
        }
        wg_sync_lds();  // every delay line of this step is in its slot (LDS-only: the prefetch and the PCM stores stay in flight)

        // ---------------- phase 2: overlap-add with the predecessor's delay line, PCM out
        if (active && emit) {
            float *frame_out = pcm + (chain_base + (size_t)t) * 1024;
            if (seq != EIGHT_SHORT) {
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const int m2 = lane + 64 * h;
                    float dl[8], dst[8];
                    load_slot(prev_slot, m2, dl);
#pragma unroll
                    for (int q = 0; q < 8; ++q) {  // dsp.rs:105-129: dst = delay + pcm * w, or delay alone
                        const int j = q < 4 ? 4 * m2 + q : 1020 - 4 * m2 + (q - 4);
                        const float v = dl[q] + xw[h][q];
                        dst[q] = (seq == LONG_STOP && j < kP0) ? dl[q] : v;
                    }
                    st_slot(frame_out, m2, dst);
                }
            } else {
                const float *sw = tabs + (shape ? kTabKbdShort : kTabSineShort);
                const float *psw = tabs + (prev_shape ? kTabKbdShort : kTabSineShort);
#pragma unroll 1
                for (int e = 0; e < 4; ++e) {  // dsp.rs:111-117
                    const int j0 = 4 * lane + 256 * e;
                    const float4 d = *reinterpret_cast<const float4 *>(prev_slot + j0);
                    float o[4] = {d.x, d.y, d.z, d.w};
                    if (j0 >= kP0) {
                        float ps[4];
                        pcm_short4(ldsf, j0 - kP0, sw, psw, ps);
#pragma unroll
                        for (int q = 0; q < 4; ++q) o[q] = o[q] + ps[q];
                    }
                    st_stream(reinterpret_cast<float4 *>(frame_out + j0), make_float4(o[0], o[1], o[2], o[3]));
                }
            }
            if (t + 1 == (long)frames_per_chain) {  // the chain's last frame: its delay line is the outgoing state
                float4 *d = reinterpret_cast<float4 *>(delay_out + (size_t)(ch ? chain1 : chain) * 1024);
#pragma unroll
                for (int q = 0; q < 4; ++q) d[lane + 64 * q] = reinterpret_cast<const float4 *>(my_slot)[lane + 64 * q];
            }
        }
        wg_sync_lds();  // the slots are free: the next transforms overwrite them
        }  // ch
        if (active) t_loaded = tn < t_end ? tn : -100;
    }
}

// The per-chain index of a batch with channel pairs: .x = the partner chain, or -1 (after the memset) for a chain outside every
// pair -- what tells the plain instantiation which chains are not its own.
__global__ void aac_js_index_kernel(const int32_t *__restrict__ pair_chains, unsigned n_pairs, unsigned n_chains, int2 *__restrict__ chain_index) {
    const unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
    // (pair_chains lives in device memory: an index outside the batch must not become a write outside the scratch)
    if (i < 2 * n_pairs && (unsigned)pair_chains[i] < n_chains && (unsigned)pair_chains[i ^ 1u] < n_chains)
        chain_index[pair_chains[i]] = make_int2(pair_chains[i ^ 1u], (int)i);
}

}  // namespace

int launch_aac(symaccel_ctx *ctx, const float *d_coeffs, const uint8_t *d_side, const float *d_delay_in,
               float *d_delay_out, float *d_pcm, size_t n_chains, size_t frames_per_chain, const AacBandMaps *maps,
               const int32_t *d_pair_chains, const symaccel_aac_js_frame *d_js_desc, size_t n_pairs, void *d_js_scratch) {
    if (frames_per_chain > 0x7fffffffu) return SYMACCEL_ERR_INVALID_ARG;
#if SYM_AAC_QUAD
    {
        // a workgroup walks seg_steps steps of four frames; two workgroups per CU (one wavefront per SIMD each)
        const size_t steps_per_chain = (frames_per_chain + 3) / 4;
        unsigned seg_steps;
        if (ctx->segment > 0) seg_steps = (unsigned)((ctx->segment + 3) / 4);
        else seg_steps = choose_segment(ctx, n_chains, steps_per_chain, SYM_AAC_MIN_WAVES, 1, 1, 1);
        if (seg_steps > steps_per_chain) seg_steps = (unsigned)steps_per_chain;
        const size_t segs = (steps_per_chain + seg_steps - 1) / seg_steps;
        const size_t grid = n_chains * segs;
        if (grid > 0x7fffffffu) return SYMACCEL_ERR_INVALID_ARG;
        AacJsArgs js{nullptr, nullptr, nullptr, AacBandMaps{}, (unsigned)n_chains};
        if (n_pairs == 0 || !maps) {
            hipLaunchKernelGGL(aac_synth_quad_kernel<false>, dim3((unsigned)grid), dim3(256), 0, ctx->stream, ctx->dev, d_coeffs, d_side,
                               d_delay_in, d_delay_out, d_pcm, (unsigned)frames_per_chain, seg_steps, (unsigned)segs, js);
        } else {
            // scratch: chain_index[n_chains] int2
            if (n_pairs * frames_per_chain > 0x7fffffffu) return SYMACCEL_ERR_INVALID_ARG;
            // the per-chain index is what the PLAIN instantiation skips paired chains by: built only when there are chains outside the pairs
            // (a batch of stereo streams has none -- the memset and the index kernel were two launches, ~9 us of a 0.26 ms step)
            const bool unpaired = 2 * n_pairs < n_chains;
            int2 *chain_index = unpaired ? reinterpret_cast<int2 *>(d_js_scratch) : nullptr;
            if (unpaired) {
                SYM_GPU(ctx, hipMemsetAsync(chain_index, 0xff, n_chains * sizeof(int2), ctx->stream));  // partner -1: not part of a pair
                hipLaunchKernelGGL(aac_js_index_kernel, dim3((unsigned)((2 * n_pairs + 255) / 256)), dim3(256), 0, ctx->stream, d_pair_chains,
                                   (unsigned)n_pairs, (unsigned)n_chains, chain_index);
                SYM_GPU(ctx, hipGetLastError());
            }
            js = AacJsArgs{chain_index, d_pair_chains, d_js_desc, *maps, (unsigned)n_chains};
            // the pairs: a workgroup per (pair, segment) -- half as many walks as chains, so the segments are chosen for n_pairs walks
            unsigned pseg_steps;
            if (ctx->segment > 0) pseg_steps = (unsigned)((ctx->segment + 3) / 4);
            else pseg_steps = choose_segment(ctx, n_pairs, steps_per_chain, SYM_AAC_MIN_WAVES, 1, 1, 1);
            if (pseg_steps > steps_per_chain) pseg_steps = (unsigned)steps_per_chain;
            const size_t psegs = (steps_per_chain + pseg_steps - 1) / pseg_steps;
            if (n_pairs * psegs > 0x7fffffffu) return SYMACCEL_ERR_INVALID_ARG;
            hipLaunchKernelGGL(aac_synth_quad_kernel<true>, dim3((unsigned)(n_pairs * psegs)), dim3(256), 0, ctx->stream, ctx->dev, d_coeffs, d_side,
                               d_delay_in, d_delay_out, d_pcm, (unsigned)frames_per_chain, pseg_steps, (unsigned)psegs, js);
            SYM_GPU(ctx, hipGetLastError());
            if (unpaired)  // the chains outside every pair: the plain walk over all chains, paired ones return at once
                hipLaunchKernelGGL(aac_synth_quad_kernel<false>, dim3((unsigned)grid), dim3(256), 0, ctx->stream, ctx->dev, d_coeffs, d_side,
                                   d_delay_in, d_delay_out, d_pcm, (unsigned)frames_per_chain, seg_steps, (unsigned)segs, js);
        }
        SYM_GPU(ctx, hipGetLastError());
        return SYMACCEL_OK;
    }
#else
    return launch_aac_wave_walk(ctx, d_coeffs, d_side, d_delay_in, d_delay_out, d_pcm, n_chains, frames_per_chain);
#endif
}

}  // namespace symaccel
