// MP3 Layer III synthesis tail: reorder -> antialias -> hybrid synthesis (36/12-point IMDCT +
// window + overlap) -> frequency inversion -> 32-band polyphase synthesis, batched over chains.
//
// Reference: symphonia-bundle-mp3/src/layer3/hybrid_synthesis.rs:153-485, 559-779;
//            symphonia-bundle-mp3/src/synthesis.rs:158-844; caller layer3/mod.rs:440-476.
//
// MI355X mapping (DESIGN.md "mp3_synth"): a 64-lane wavefront carries TWO chains (one per 32-lane
// half) and is its own workgroup -- no workgroup barriers, only wave-local LDS ordering.  Each half
// walks a segment of consecutive granules of its chain, with everything that crosses granules held
// in REGISTERS:
//   * lane = sub-band for antialias + hybrid synthesis: the sub-band's 18 lines and its 18-sample
//     overlap are registers; the 8 antialias butterflies per boundary exchange with lanes sb-1 / sb+1;
//   * lane = time slot for the 18 dct32s (one LDS transpose in, one out, both conflict-free b128);
//   * lane = output sample i for the 512-tap window: the lane keeps the two V-vector entries it needs
//     from each of the last 16 + 18 time slots (V[i] and V[32+i], which are +-copies of dct32 outputs,
//     synthesis.rs:247-263) in registers, so the 16-tap dot product per sample reads no memory.
// The next granule's 576 lines and its side word are prefetched (16 B/lane, 512 B per half-wave) while the current
// one is transformed; the side word is decoded only when its round starts (touching it earlier makes the compiler
// wait for the whole prefetch right after issuing it).  Segments other than a chain's first start with a two-granule
// halo (granule g-2 rebuilds the overlap, granule g-1 rebuilds the 16-slot history); both depend only on inputs.
// PCM stores are 4 B per lane (128 B per half-wave): transposing the granule through LDS for 16-byte stores was
// measured and is not faster.  LDS per wavefront: 5 KiB (granule tiles, then the dct32 transpose) + synthesis-window
// rows 2.5 KiB + the four IMDCT windows 0.6 KiB (per-lane block types: LDS broadcast instead of vector global loads).
// Roofline: HBM-bound on paper, 2304 B in + 2304 B out per granule-channel, ~34 kflop (no FMA) -> 7.4 flop/B; in
// practice latency / VALU-issue bound at three wavefronts per SIMD.
#include "mp3_common.h"
#include "mp3_requant.h"

namespace symaccel {

namespace {

#ifndef SYM_MP3_VARIANT
#define SYM_MP3_VARIANT 4
#endif
// Build variants (tuning knob SYM_MP3_VARIANT, see build.py; DESIGN.md 4.2 has the measurements -- all within 3 % of
// each other, which is the finding):
//   0  one wavefront per workgroup; one granule of spectral lines in flight; the window pass stores each PCM sample as
//      it is produced (4 B per lane, 128 B per half-wave and instruction);
//   2  the granule's PCM is collected in an LDS tile and stored as float4 (16 B per lane) when the window pass is done;
//      four wavefronts per workgroup share the window tables, which pays for the extra 4.5 KiB of LDS per wavefront;
//   3  two granules of spectral lines in flight instead of one;
//   4  prefetch and PCM stores issued unconditionally, the prefetched lines consumed at the end of the round (SYM_MP3_SINK
//      below): the waits are for the lines only -- measured equal during the clock ramp (profiles/r03k_mp3_sink_ab.txt: the
//      write acknowledgements were not what the kernel waits for), 1 % ahead at the sustained clock together with the packed
//      window pass (profiles/r03x_sustained_ab.txt): the product.
// Round 3's measurements (tools/ubench/valu_clock.hip, tools/kernel_clock_probe.py): the loads and stores alone
// run at 5.3-5.6 TB/s (a loads-and-stores-only build, profiles/r03*_mp3_ablate*; removed from this file in round 4); a wavefront issues one instruction per ~4.9 cycles, a SIMD's VALU port accepts one
// plain f32 instruction per ~2.5 cycles, so two wavefronts saturate it and the third of a SIMD gets the leftovers (walks of 45
// rounds finish after 212 / 231 / 315 us on every SIMD); under this kernel's VALU + LDS + HBM load the part clocks at 1.7 GHz
// (2.3 GHz for the same loads and stores without the arithmetic).  The kernel is bound by VALU issue at a throttled clock.
#define SYM_MP3_OTILE (SYM_MP3_VARIANT == 2)
#define SYM_MP3_PREFETCH2 (SYM_MP3_VARIANT == 3)
#ifndef SYM_MP3_WG_WAVES
#define SYM_MP3_WG_WAVES (SYM_MP3_OTILE ? 4 : 1)
#endif
constexpr int kWgWaves = SYM_MP3_WG_WAVES;       // wavefronts per workgroup
constexpr int kTileFloats = 2 * 576;             // two granule tiles (one per half-wave)
// per-wavefront LDS: the granule tiles, overlaid by the dct32 transpose S (the tiles are dead before S is written);
// variant 2: + the PCM tiles
constexpr int kSBase = 0;
constexpr int kOBase = kSBase + 2 * 18 * kSStride;
constexpr int kMetaBase = kOBase + (SYM_MP3_OTILE ? kTileFloats : 0);  // 4 words: (chain, ends-its-chain flag) of each half-wave, for the epilogue
constexpr int kWaveFloats = kMetaBase + 4;
// per-workgroup LDS tables: synthesis window rows, then the four 36-entry IMDCT windows
constexpr int kTabFloats = kDwFloats + 4 * 36;
static_assert(2 * 18 * kSStride >= kTileFloats, "the transpose area must hold the two granule tiles");

// ---- 36-point IMDCT (Szu-Wei Lee), hybrid_synthesis.rs:559-779 -------------------------------

// sdct_ii_9 (hybrid_synthesis.rs:720-779); writes y[0], y[2], ..., y[16]
__device__ __forceinline__ void sdct_ii_9(const float (&x)[9], float *y) {
    constexpr const float *D = kMp3Lit + MP3C_SDCT9_D;  // compile-time literals (mp3_literals.h)
    const float a01 = x[3] + x[5], a02 = x[3] - x[5], a03 = x[6] + x[2], a04 = x[6] - x[2];
    const float a05 = x[1] + x[7], a06 = x[1] - x[7], a07 = x[8] + x[0], a08 = x[8] - x[0];
    const float a09 = x[4] + a05, a10 = a01 + a03, a11 = a10 + a07, a12 = a03 - a07;
    const float a13 = a01 - a07, a14 = a01 - a03, a15 = a02 - a04, a16 = a15 + a08;
    const float a17 = a04 + a08, a18 = a02 - a08, a19 = a02 + a04, a20 = 2.0f * x[4] - a05;
    const float m1 = D[0] * a06, m2 = D[1] * a12, m3 = D[2] * a13, m4 = D[3] * a14;
    const float m5 = D[0] * a16, m6 = D[4] * a17, m7 = D[5] * a18, m8 = D[6] * a19;
    const float a21 = a20 + m2, a22 = a20 - m2, a23 = a20 + m3;
    const float a24 = m1 + m6, a25 = m1 - m6, a26 = m1 + m7;
    y[0] = a09 + a11;
    y[2] = m8 - a26;
    y[4] = m4 - a21;
    y[6] = m5;
    y[8] = a22 - m3;
    y[10] = a25 - m7;
    y[12] = a11 - 2.0f * a09;
    y[14] = a24 + m8;
    y[16] = a23 + m4;
}

// dct_iv (hybrid_synthesis.rs:608-660) incl. sdct_ii_18 (:665-716)
__device__ __forceinline__ void dct_iv_18(const float (&x)[18], float (&y)[19]) {
    constexpr const float *mc = kMp3Lit;
    float s[18];
#pragma unroll
    for (int i = 0; i < 18; ++i) s[i] = mc[MP3C_DCT_IV + i] * x[i];
    float even[9], odd[9];
#pragma unroll
    for (int i = 0; i < 9; ++i) even[i] = s[i] + s[17 - i];
    sdct_ii_9(even, &y[0]);
#pragma unroll
    for (int i = 0; i < 9; ++i) odd[i] = mc[MP3C_SDCT18 + i] * (s[i] - s[17 - i]);
    sdct_ii_9(odd, &y[1]);
#pragma unroll
    for (int i = 3; i <= 17; i += 2) y[i] -= y[i - 2];
    y[0] /= 2.0f;
#pragma unroll
    for (int i = 1; i < 18; ++i) y[i] = (y[i] / 2.0f) - y[i - 1];
}

// imdct36 (hybrid_synthesis.rs:571-603): x[18] in place, overlap[18] in/out
// `window`: the block type's 36-entry window in LDS (the block type differs between the two chains of a
// wavefront, so the window is per lane: LDS broadcast reads instead of per-lane global loads)
__device__ __forceinline__ void imdct36(float (&x)[18], float (&overlap)[18], const float *window) {
    float dct[19];
    dct_iv_18(x, dct);
#pragma unroll
    for (int i = 0; i < 9; ++i) x[i] = overlap[i] + dct[9 + i] * window[i];
#pragma unroll
    for (int i = 9; i < 18; ++i) x[i] = overlap[i] - dct[27 - i - 1] * window[i];
#pragma unroll
    for (int i = 18; i < 27; ++i) overlap[i - 18] = -dct[27 - i - 1] * window[i];
#pragma unroll
    for (int i = 27; i < 36; ++i) overlap[i - 18] = -dct[i - 27] * window[i];
}

// imdct12_win (hybrid_synthesis.rs:363-455)
__device__ __forceinline__ void imdct12_win(float (&x)[18], float (&overlap)[18], cf32p window) {
    constexpr const float *cos12 = kMp3Lit + MP3C_COS12;
    float tmp[36];
#pragma unroll
    for (int i = 0; i < 36; ++i) tmp[i] = 0.0f;
#pragma unroll
    for (int w = 0; w < 3; ++w) {
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const float *cl = cos12 + 6 * i, *cr = cos12 + 6 * (i + 3);
            const float yl = (x[w] * cl[0]) + (x[3 + w] * cl[1]) + (x[6 + w] * cl[2]) + (x[9 + w] * cl[3]) +
                             (x[12 + w] * cl[4]) + (x[15 + w] * cl[5]);
            const float yr = (x[w] * cr[0]) + (x[3 + w] * cr[1]) + (x[6 + w] * cr[2]) + (x[9 + w] * cr[3]) +
                             (x[12 + w] * cr[4]) + (x[15 + w] * cr[5]);
            tmp[6 + 6 * w + 3 - i - 1] += -yl * window[3 - i - 1];
            tmp[6 + 6 * w + i + 3] += yl * window[i + 3];
            tmp[6 + 6 * w + i + 6] += yr * window[i + 6];
            tmp[6 + 6 * w + 12 - i - 1] += yr * window[12 - i - 1];
        }
    }
#pragma unroll
    for (int i = 0; i < 18; ++i) {
        x[i] = tmp[i] + overlap[i];
        overlap[i] = tmp[i + 18];
    }
}

__device__ __forceinline__ void fetch_granule(const float *granule, int hl, float4 (&line)[5]) {
    const float4 *src = reinterpret_cast<const float4 *>(granule);
#pragma unroll
    for (int q = 0; q < 4; ++q) line[q] = ld_stream(src + hl + 32 * q);
    line[4] = ld_stream(src + 128 + (hl & 15));  // lanes 16..31 re-read float4 128..143 (same cache lines) and ignore it
}

#ifndef SYM_MP3_FRONT
#define SYM_MP3_FRONT 0  // bit 0: requantize on packed pairs + SDWA addresses, bit 1: mid/side through v_permlane32_swap (round 6: see pk_abs_clamp_i16; measured, not faster); 0: the round-5 form
#endif
#ifndef SYM_MP3_PACKED
#define SYM_MP3_PACKED 1
#endif
// SYM_MP3_PACKED 1: the window pass computes TWO time slots per instruction stream with v_pk_mul_f32 / v_pk_add_f32 (the two
// 16-tap sums of slots 2p and 2p + 1 are independent and use the same coefficients).  A wavefront issues at most one
// instruction per ~4.9 cycles whatever the instruction is (tools/ubench/valu_clock.hip), a packed instruction costs one
// such slot and ~1.7 plain ones on the SIMD's VALU port: 288 issue slots instead of 576 per granule pair.  The V history
// is kept as register PAIRS of raw dct32 outputs: HA[t] / HB[t], t = 16 + slot (t < 16: the previous granule), with
//   PA[k] = (HA[2k], HA[2k + 1]),  PB[k] = (HB[2k - 1], HB[2k]),
// so that both operand pairs of slot pair p and tap j are the aligned pairs PA[8 + p - j], PB[8 + p - j].  The signs of
// synthesis.rs:247-263 (V[i] = +-d[..], V[32 + i] = -d[..]) are folded into the lane's window coefficients -- (-a) * d and
// a * (-d) are the same bits -- and applied to the history only where it enters or leaves the kernel.
typedef float v2f __attribute__((vector_size(8)));  // (GCC / clang vector extension: the emulation build is g++)
#ifndef SYM_MP3_PAIR_GROUP
#define SYM_MP3_PAIR_GROUP 3
#endif
#ifndef SYM_MP3_SLOT_GROUP
#define SYM_MP3_SLOT_GROUP 3
#endif
#ifndef SYM_MP3_WAVES
#define SYM_MP3_WAVES 3  // wavefronts per SIMD the register allocation must allow (build-time tuning knob)
#endif
#ifndef SYM_MP3_FUSED_WG_WAVES
#define SYM_MP3_FUSED_WG_WAVES 4  // wavefronts per workgroup of the fused (int16 -> PCM) kernel
#endif
#ifndef SYM_MP3_FUSED_WAVES
#define SYM_MP3_FUSED_WAVES 2     // wavefronts per SIMD its register allocation must allow (build-time tuning knob)
#endif
// SYM_MP3_SINK (variant 4): gfx950 has ONE in-order counter for vector loads and stores (vmcnt).  With the PCM stores inside
// `if (emit)` and the prefetch inside `if (r + 1 < my_rounds)` the compiler cannot know how many stores follow the prefetch,
// and at the loop header it also has to honour the state of the loop's entry (the first fetch, no store behind it): it waits
// with vmcnt(5) ... vmcnt(0), i.e. until the stores issued a few instructions earlier are acknowledged by memory as well --
// the write latency was exposed once per round.  Here every round issues its prefetch and its 18 stores unconditionally
// (lanes that must not emit -- halo rounds, an idle half-wave -- aim at a per-wavefront slot of a sink buffer; a half-wave
// without a next granule re-reads granule 0), and the prefetched lines are consumed at the END of the round body, in the
// same straight-line region: the waits become vmcnt(22) ... vmcnt(18), for the lines only.
#define SYM_MP3_SINK (SYM_MP3_VARIANT == 4)
constexpr int kSinkSlotFloats = 2048;  // 8 KiB per wavefront slot: two half-waves x 18 rows of 128 B
constexpr int kSinkSlots = 256;
// ---- the fused front (FUSED = true; symaccel_mp3_decode_device): the wavefront's two half-waves are the two CHANNELS of one
// stream (half 0 = channel 0), walking the same segment of the same granules, and what they load is what the entropy
// decoder produced -- 576 quantised Huffman samples (int16) and the 52-byte requantize record per granule-channel, one
// 48-byte joint-stereo record per granule -- instead of the f32 spectra.  At the end of round r the front turns granule
// r + 1 into the LDS tile the round starts from: requantize (requantize.rs:117-147, 239-380: POW43 look-up, the band's
// 2^(0.25 (A - B)), one rounded multiply; mp3_requant.h, the arithmetic of mp3_requantize_kernel) with lane = sub-band
// (lines 18 hl .. 18 hl + 17, the layout the hybrid synthesis wants), then joint stereo (stereo.rs:485-556) against the
// other half-wave's lines (ds_bpermute with lane ^ 32) -- mid/side below the intensity bound, the band walk of
// process_intensity_* on the channel-1 zero-band mask when the frame uses intensity stereo (mp3_requant.h, shared with
// mp3_stereo_kernel).  The f32 spectra never exist in HBM: int16 in, PCM out, one pass.
// Both look-up tables of requantize live in LDS IN FULL (POW43: 8207 entries, 32 KiB; 2^(0.25 e): 1346 entries, 5.3 KiB), shared by
// the workgroup's four wavefronts.  A global load anywhere in the front -- even in a branch that is never taken: the compiler
// places the wait where the paths join -- makes the wavefront wait for its own 18 PCM stores of the round (gfx950 counts
// loads and stores with one in-order counter); 69 KiB of LDS per workgroup still leaves two workgroups (two wavefronts per
// SIMD) per CU.
constexpr int kFrontPow = 8208;                         // POW43 (8207 entries, padded)
constexpr int kFrontMapFloats = 4 * 576 / 4;            // the four line -> band maps of this sample rate (bytes)
constexpr int kMapShift = (SYM_MP3_FRONT & 1) ? 2 : 0;            // the maps in LDS hold 4 x band (a byte offset into the scale table): see SYM_MP3_FRONT
constexpr int kP2MinE = kMp3Pow2abMinE, kP2Len = kMp3Pow2abLen;
constexpr int kFrontEdgeFloats = (sizeof(SfbEdges) + 3) / 4;  // the band edge tables (the intensity walk reads them through a pointer)
constexpr int kFrontTabFloats = kFrontPow + kFrontMapFloats + kP2Len + kFrontEdgeFloats;
constexpr int kFwScale = 0;                             // per wavefront: scale[2][40]
constexpr int kFwRq = kFwScale + 2 * kMp3Slots;         //   the two requantize records (13 dwords each, padded to 14)
constexpr int kFwSt = kFwRq + 2 * 14;                   //   the joint-stereo record (12 dwords)
constexpr int kFwNz = kFwSt + 12;                       //   zero-band mask of channel 1 (2 words)
constexpr int kFwAct = kFwNz + 2;                       //   per band: action, left ratio, right ratio
constexpr int kFwKl = kFwAct + 40;
constexpr int kFwKr = kFwKl + 40;
constexpr int kFrontWaveFloats = kFwKr + 40;
static_assert(sizeof(symaccel_mp3_requant) == 52 && sizeof(symaccel_mp3_stereo) == 48, "records are fetched as 13 / 12 dwords");

// Round 6.  With two wavefronts per SIMD the fused kernel is bound by how many instructions a wavefront can issue (one per ~4.9 cycles, whatever
// the instruction: tools/ubench/valu_clock.hip), so the front is written for the NUMBER of instructions per line:
//  * |sample| and its clamp to POW43's last entry on the PACKED 16-bit pairs as the entropy decoder stores them (v_pk_sub_i16, v_pk_max_i16,
//    v_pk_min_u16: three instructions per two lines instead of six; -32768 stays -32768, i.e. 32768 unsigned, and clamps like every |sample| > 8206);
//  * the POW43 address (4 x magnitude) straight from a half of the packed word (v_lshlrev_b32_sdwa, src1_sel:WORD_n): no unpacking at all -- the
//    sign of an odd line is bit 31 of the raw word, of an even line bit 31 of the word shifted up;
//  * the band maps sit in LDS pre-multiplied by four, so a line's scale address is ONE v_add_u32_sdwa with a byte select;
//  * the rzero partition (literal +0.0, requantize.rs:117-147) is applied to the finished line;
//  * mid/side (stereo.rs:139-148): v_permlane32_swap_b32 puts channel 0's lines 2k / 2k + 1 into one register (low / high half-wave) and channel
//    1's into another: one add, one subtract, two multiplies and a second swap per TWO lines and both channels, instead of a ds_bpermute, a sign
//    flip, an add and a multiply per line -- the same IEEE operations on the same operands (c0 + c1 commutes, c0 - c1 is c0 + (-c1));
//  * no select behind mid/side when the stereo record's bound is not below either channel's rzero (what a front end that fills both records from
//    one granule always produces): the lines behind the bound are +0.0 in both channels and (0 + 0) k = (0 - 0) k = +0.0.
__device__ __forceinline__ uint32_t pk_abs_clamp_i16(uint32_t w) {
#if defined(__HIP_DEVICE_COMPILE__)
    uint32_t n, m;
    asm("v_pk_sub_i16 %0, 0, %1" : "=v"(n) : "v"(w));
    asm("v_pk_max_i16 %0, %1, %2" : "=v"(m) : "v"(n), "v"(w));
    asm("v_pk_min_u16 %0, %1, %2" : "=v"(m) : "v"(m), "v"(0x200e200eu));
    return m;
#else
    auto one = [](uint32_t h) {
        const int v = (int)(int16_t)(uint16_t)h;
        const uint32_t a = (uint32_t)(v < 0 ? -v : v) & 0xffffu;  // (-32768: 32768)
        return a > 8206u ? 8206u : a;
    };
    return one(w & 0xffffu) | (one(w >> 16) << 16);
#endif
}
// 4 x (half `hi` of a packed pair of u16)
template <int HI>
__device__ __forceinline__ uint32_t pk_half_times4(uint32_t m) {
#if defined(__HIP_DEVICE_COMPILE__)
    uint32_t r;
    if constexpr (HI) asm("v_lshlrev_b32_sdwa %0, 2, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_1" : "=v"(r) : "v"(m));
    else asm("v_lshlrev_b32_sdwa %0, 2, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_0" : "=v"(r) : "v"(m));
    return r;
#else
    return ((HI ? m >> 16 : m) & 0xffffu) << 2;
#endif
}
// base + byte B (0 / 1) of w
template <int B, typename A>
__device__ __forceinline__ A add_byte(A base, uint32_t w) {
#if defined(__HIP_DEVICE_COMPILE__)
    A r;
    if constexpr (B) asm("v_add_u32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1" : "=v"(r) : "v"(base), "v"(w));
    else asm("v_add_u32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_0" : "=v"(r) : "v"(base), "v"(w));
    return r;
#else
    return base + ((w >> (8 * B)) & 0xffu);
#endif
}
// x.hi <-> y.lo (lanes 32..63 of x with lanes 0..31 of y)
__device__ __forceinline__ void half_wave_swap(float &x, float &y) {
#if defined(__HIP_DEVICE_COMPILE__)
    const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(y), false, false);
    x = __uint_as_float(r[0]);
    y = __uint_as_float(r[1]);
#else
    const float xo = __shfl_xor(x, 32), yo = __shfl_xor(y, 32);
    const bool low = (threadIdx.x & 32u) == 0;
    const float nx = low ? x : yo, ny = low ? xo : y;
    x = nx;
    y = ny;
#endif
}

// The exponent A - B of mp3_slot_scale (mp3_requant.h; requantize.rs:260-352) of scale slot `slot` (< kMp3Unscaled), without a
// memory table: the pre-emphasis values of ISO/IEC 11172-3 Table B.6 are two bits per band in a constant.
__device__ __forceinline__ int front_slot_exponent(const symaccel_mp3_requant &d, int slot, int switch_point) {
    // bands 0..21: 0 0 0 0 0 0 0 0 0 0 0 1 1 1 1 2 2 3 3 3 2 0
    constexpr unsigned long long kPre2 = (1ull << 22) | (1ull << 24) | (1ull << 26) | (1ull << 28) | (2ull << 30) | (2ull << 32) | (3ull << 34) |
                                         (3ull << 36) | (3ull << 38) | (2ull << 40);
    const bool is_short = d.block_type == SYMACCEL_MP3_SHORT;
    const int sw = is_short ? (d.is_mixed ? switch_point : 0) : 64;
    const int shift = (d.flags & SYMACCEL_MP3_RQ_SCALEFAC_SCALE) ? 2 : 1;
    const int gain = (int)d.global_gain - 210;
    if (slot < sw) {
        const int pre = ((d.flags & SYMACCEL_MP3_RQ_PREFLAG) && slot < 22) ? (int)((kPre2 >> (2 * slot)) & 3ull) : 0;
        return gain - (((int)d.scalefacs[slot] + pre) << shift);
    }
    const int win = (slot - sw) % 3;
    return gain - 8 * (int)d.subblock_gain[win] - ((int)d.scalefacs[slot] << shift);
}

// the 18 quantised samples of lane hl's sub-band (36 bytes, 4-byte aligned: 9 dwords)
__device__ __forceinline__ void fetch_quant(const int16_t *granule, int hl, uint32_t (&qw)[9]) {
    const uint32_t *src = reinterpret_cast<const uint32_t *>(granule) + 9 * hl;
#pragma unroll
    for (int k = 0; k < 9; ++k) qw[k] = src[k];
}

// The intensity-stereo plan of one granule (stereo.rs:196-483): which bands are intensity coded is decided from the zero bands of
// channel 1 (the half-wave 1 lanes flag the bands of their non-zero lines), then every lane runs the reference's band walk
// on that mask and lanes 0..39 turn it into per-band actions and ratios.  Rare (low bit rate streams), long, branchy:
// kept OUT of the synthesis loop's instruction stream.
// Works on the two half-waves' LDS tiles (the requantised lines of channel 0 and channel 1, natural order), in place; every argument
// is a scalar or a pointer into LDS -- an array passed by reference would live in scratch memory in the CALLER too.
__device__ __attribute__((noinline)) void mp3_front_intensity(const float *is_ratios, const SfbEdges *e, const symaccel_mp3_stereo *sd, float *tiles,
                                                               const uint8_t *bmap_row, int hl, int half, float *fw) {
    unsigned *nzw = reinterpret_cast<unsigned *>(fw + kFwNz);
    int *act = reinterpret_cast<int *>(fw + kFwAct);
    float *kl = fw + kFwKl, *kr = fw + kFwKr;
    const bool mid_side = sd->flags & SYMACCEL_MP3_ST_MID_SIDE;
    const int rzero1 = sd->rzero1 > 576 ? 576 : (int)sd->rzero1;
    int end = sd->rzero0 > sd->rzero1 ? sd->rzero0 : sd->rzero1;  // stereo.rs:522
    end = end > 576 ? 576 : end;
    const uint8_t *bmap = bmap_row + 18 * hl;
    float c0[18], c1[18];
#pragma unroll
    for (int i = 0; i < 18; ++i) {
        c0[i] = tiles[18 * hl + i];
        c1[i] = tiles[576 + 18 * hl + i];
    }
    if (half == 1) {  // is_zero_band (stereo.rs:189-192) of channel 1: one bit per band
        unsigned long long mine = 0ull;
#pragma unroll
        for (int i = 0; i < 18; ++i)
            if (c1[i] != 0.0f) mine |= 1ull << (bmap[i] >> kMapShift);
        if ((unsigned)mine) atomicOr(&nzw[0], (unsigned)mine);
        if ((unsigned)(mine >> 32)) atomicOr(&nzw[1], (unsigned)(mine >> 32));
    }
    wave_sync();  // (also: every lane has read its lines of both tiles)
    const Mp3StereoPlan plan = mp3_stereo_walk(*sd, *e, (unsigned long long)nzw[0] | ((unsigned long long)nzw[1] << 32), end, rzero1);
    const int k = half == 0 ? hl : 32 + hl;
    if (k < 40) mp3_stereo_expand(plan, *sd, is_ratios, k, act, kl, kr);
    wave_sync();
#pragma unroll
    for (int i = 0; i < 18; ++i) {
        mp3_stereo_apply(c0[i], c1[i], 18 * hl + i, plan.bound, mid_side, true, bmap[i] >> kMapShift, act, kl, kr);
        tiles[576 * half + 18 * hl + i] = half == 0 ? c0[i] : c1[i];
    }
}

// Granule in registers -> the half-wave's LDS tile (natural line order).  Every lane of the wavefront calls this together.
__device__ __forceinline__ void mp3_front(const DevTables &tb, const SfbEdges *e_lds, int mixed_switch, const uint32_t (&qw)[9], uint32_t dq, int hl,
                                          int half, bool pair_live, const float *pow43_lo, const uint8_t *maps, const float *p2_lo, float *fw,
                                          float *tile) {
    float *scale = fw + kFwScale + half * kMp3Slots;
    uint32_t *rqw = reinterpret_cast<uint32_t *>(fw + kFwRq) + 14 * half;
    uint32_t *stw = reinterpret_cast<uint32_t *>(fw + kFwSt);
    unsigned *nzw = reinterpret_cast<unsigned *>(fw + kFwNz);
    wave_sync();  // (the records and tables of the previous granule are no longer read)
    if (hl < 13) rqw[hl] = dq;
    else if (hl < 25 && half == 0) stw[hl - 13] = dq;
    if (hl < 2 && half == 0) nzw[hl] = 0u;
    wave_sync();
    const symaccel_mp3_requant &rd = *reinterpret_cast<const symaccel_mp3_requant *>(rqw);
    const symaccel_mp3_stereo &sd = *reinterpret_cast<const symaccel_mp3_stereo *>(stw);
    {   // the slots' 2^(0.25 (A - B)) (mp3_slot_scale, mp3_requant.h): lane hl has slot hl and, lanes 0..7, slot 32 + hl
        auto slot_scale = [&](int slot) {
            if (slot >= kMp3Unscaled) return 1.0f;  // lines no band covers: x * 1.0f == x
            int idx = front_slot_exponent(rd, slot, mixed_switch) - kP2MinE;
            idx = idx < 0 ? 0 : (idx >= kP2Len ? kP2Len - 1 : idx);
            return p2_lo[idx];
        };
        scale[hl] = slot_scale(hl);
        if (hl < kMp3Slots - 32) scale[32 + hl] = slot_scale(32 + hl);
    }
    wave_sync();
    const int rz = rd.rzero > 576 ? 576 : (int)rd.rzero;
    const uint8_t *smap = maps + 576 * mp3_requant_kind(rd) + 18 * hl;
    // mp3_sample_value (mp3_requant.h) x the band's scale, with every table in LDS and NO conditional around a load: the
    // compiler turns `cond ? table[i] : 0` into a branch per line with the LDS round trip inside it -- 54 dependent round trips
    // per granule.  Here the index is made harmless instead (a zeroed sample reads POW43[0] = +0.0), the 18 + 9 + 18 reads
    // of a phase are independent of each other, and the sign is OR-ed in (POW43 >= 0).
#if SYM_MP3_FRONT & 1
    float a[18];
    {
        unsigned sm[9];  // the lane's 18 bytes of 4 x slot, two per 16-bit read (18 hl is even)
#pragma unroll
        for (int k = 0; k < 9; ++k) sm[k] = reinterpret_cast<const uint16_t *>(smap)[k];
        const char *pow_b = reinterpret_cast<const char *>(pow43_lo);
        // the scale table's address as the 32-bit LDS address it is: base + byte select is then the whole address computation of a line's scale
#if defined(__HIP_DEVICE_COMPILE__)
        typedef __attribute__((address_space(3))) const float *lds_cfp;
        const uint32_t scale_addr = (uint32_t)(uintptr_t)(lds_cfp)scale;
#define SYM_FRONT_SCALE(off) (*(lds_cfp)(uintptr_t)(off))
#else
        const uintptr_t scale_addr = reinterpret_cast<uintptr_t>(scale);
#define SYM_FRONT_SCALE(off) (*reinterpret_cast<const float *>(off))
#endif
        float pw[18], sc[18];
#pragma unroll
        for (int k = 0; k < 9; ++k) {
            const uint32_t m = pk_abs_clamp_i16(qw[k]);
            pw[2 * k] = *reinterpret_cast<const float *>(pow_b + pk_half_times4<0>(m));
            pw[2 * k + 1] = *reinterpret_cast<const float *>(pow_b + pk_half_times4<1>(m));
            sc[2 * k] = SYM_FRONT_SCALE(add_byte<0>(scale_addr, sm[k]));
            sc[2 * k + 1] = SYM_FRONT_SCALE(add_byte<1>(scale_addr, sm[k]));
        }
#pragma unroll
        for (int i = 0; i < 18; ++i) {
            const uint32_t w = qw[i >> 1];
            const uint32_t sign_src = (i & 1) ? w : w << 16;  // little endian: the even sample is the low half
            const float v = __uint_as_float((sign_src & 0x80000000u) | __float_as_uint(pw[i])) * sc[i];
            a[i] = 18 * hl + i < rz ? v : 0.0f;                // the rzero partition: literal +0.0 (requantize.rs:117-147)
        }
    }
#else
    int mag[18];
    unsigned sgn[18];
#pragma unroll
    for (int i = 0; i < 18; ++i) {
        const uint32_t w = qw[i >> 1];
        int sv = (i & 1) ? (int)w >> 16 : (int)(w << 16) >> 16;  // little endian: the even sample is the low half
        sv = 18 * hl + i < rz ? sv : 0;                           // the rzero partition: literal +0.0 (requantize.rs:117-147)
        const int m = sv < 0 ? -sv : sv;
        mag[i] = m > 8206 ? 8206 : m;
        sgn[i] = (unsigned)sv & 0x80000000u;
    }
    float pw[18];
#pragma unroll
    for (int i = 0; i < 18; ++i) pw[i] = pow43_lo[mag[i]];
    unsigned sm[9];  // the 18 slot indices of the lane's lines, two per 16-bit read (18 hl is even)
#pragma unroll
    for (int k = 0; k < 9; ++k) sm[k] = reinterpret_cast<const uint16_t *>(smap)[k];
    float a[18];
#pragma unroll
    for (int i = 0; i < 18; ++i) {
        const float sc = scale[(sm[i >> 1] >> (8 * (i & 1))) & 255u];
        a[i] = __uint_as_float(__float_as_uint(pw[i]) | sgn[i]) * sc;
    }
#endif
    // ---- joint stereo (both half-waves hold the same lines of their channel)
    const bool mid_side = pair_live && (sd.flags & SYMACCEL_MP3_ST_MID_SIDE), intensity = pair_live && (sd.flags & SYMACCEL_MP3_ST_INTENSITY);
    if (mid_side && !intensity) {  // (wave-uniform: one record per pair and granule) mid/side alone, stereo.rs:139-148, 541-543
        int end = sd.rzero0 > sd.rzero1 ? sd.rzero0 : sd.rzero1;  // stereo.rs:522
        end = end > 576 ? 576 : end;
        // (c0 + c1) in channel 0, (c0 - c1) in channel 1: the other channel's line plus this one's with its sign flipped in the second
        // half-wave -- IEEE subtraction IS addition of the negated operand, and addition commutes, so both are the reference's rounded
        // results; two selects and a subtract per line less than choosing c0 / c1 and the operation (compares and selects cost 1.75 x
        // an add or a xor on the SIMD: profiles/r05q_valu_int.txt)
#if SYM_MP3_FRONT & 2
        // (see SYM_MP3_FRONT) `plain`: nothing non-zero at or behind the bound in either channel -- wave-uniform, both records are in LDS
        const uint32_t rz_a = reinterpret_cast<const symaccel_mp3_requant *>(fw + kFwRq)->rzero;
        const uint32_t rz_b = reinterpret_cast<const symaccel_mp3_requant *>(reinterpret_cast<const uint32_t *>(fw + kFwRq) + 14)->rzero;
        const bool plain = (int)(rz_a > 576u ? 576u : rz_a) <= end && (int)(rz_b > 576u ? 576u : rz_b) <= end;
        if (plain) {
#pragma unroll
            for (int k = 0; k < 9; ++k) {
                float x = a[2 * k], y = a[2 * k + 1];
                half_wave_swap(x, y);  // x: channel 0's lines 2k | 2k + 1, y: channel 1's
                float sum = (x + y) * kMp3Frac1Sqrt2, dif = (x - y) * kMp3Frac1Sqrt2;
                half_wave_swap(sum, dif);  // sum: line 2k as channel 0 | channel 1 want it, dif: line 2k + 1
                a[2 * k] = sum;
                a[2 * k + 1] = dif;
            }
        } else {
#pragma unroll
            for (int k = 0; k < 9; ++k) {
                float x = a[2 * k], y = a[2 * k + 1];
                half_wave_swap(x, y);
                float sum = (x + y) * kMp3Frac1Sqrt2, dif = (x - y) * kMp3Frac1Sqrt2;
                half_wave_swap(sum, dif);
                a[2 * k] = 18 * hl + 2 * k < end ? sum : a[2 * k];
                a[2 * k + 1] = 18 * hl + 2 * k + 1 < end ? dif : a[2 * k + 1];
            }
        }
#else
        const unsigned flip = half == 0 ? 0u : 0x80000000u;
#pragma unroll
        for (int i = 0; i < 18; ++i) {
            const float b = __shfl_xor(a[i], 32);
            const float v = (b + __uint_as_float(__float_as_uint(a[i]) ^ flip)) * kMp3Frac1Sqrt2;
            a[i] = 18 * hl + i < end ? v : a[i];
        }
#endif
    }
    float2 *t2 = reinterpret_cast<float2 *>(tile + 18 * hl);  // 72 B lane stride: conflict-free b64
#pragma unroll
    for (int k = 0; k < 9; ++k) t2[k] = make_float2(a[2 * k], a[2 * k + 1]);
    if (intensity) {  // rare (low bit rates): the band walk and its application, on the two tiles, out of line
        const bool is_short = sd.block_type == SYMACCEL_MP3_SHORT, is_mixed = is_short && sd.is_mixed;
        wave_sync();
        mp3_front_intensity(tb.mp3_is_ratios, e_lds, &sd, tile - 576 * half, maps + 576 * (is_short ? (is_mixed ? 3 : 1) : 0), hl, half, fw);
    }
}

#ifdef SYMACCEL_EMULATED_HIP
#define SYM_MP3_WAVES_ATTR(fused)
#else
#define SYM_MP3_WAVES_ATTR(fused) \
    __attribute__((amdgpu_waves_per_eu((fused) ? SYM_MP3_FUSED_WAVES : SYM_MP3_WAVES, (fused) ? SYM_MP3_FUSED_WAVES : SYM_MP3_WAVES)))
#endif
template <int WGW, bool FUSED>
__global__ __launch_bounds__(64 * WGW) SYM_MP3_WAVES_ATTR(FUSED) void mp3_synth_kernel(
    DevTables tb, const float *__restrict__ xr, const symaccel_mp3_side *__restrict__ side, int sr,
    const float *__restrict__ overlap_in, const float *__restrict__ vvec_in, const int32_t *__restrict__ vfront_in,
    float *__restrict__ overlap_out, float *__restrict__ vvec_out, int32_t *__restrict__ vfront_out,
    float *__restrict__ pcm, float *__restrict__ sink, unsigned n_chains, unsigned granules_per_chain, unsigned seg_len,
    unsigned segs_per_chain, const int16_t *__restrict__ quant, const symaccel_mp3_requant *__restrict__ rq_desc,
    const symaccel_mp3_stereo *__restrict__ st_desc, const int32_t *__restrict__ pair_chains, SfbEdges edges, unsigned chain_bound) {
    constexpr int kWgWaves = WGW;  // (shadows the file-level constant: wavefronts per workgroup of THIS instantiation)
    constexpr int kWaveFloatsT = kWaveFloats + (FUSED ? kFrontWaveFloats : 0);
    __shared__ __attribute__((aligned(16))) float lds_tab[kTabFloats + (FUSED ? kFrontTabFloats : 0)];
    __shared__ __attribute__((aligned(16))) float lds_wave[WGW][kWaveFloatsT];
    const int wave = (int)threadIdx.x >> 6;
    const int half = ((int)threadIdx.x >> 5) & 1, hl = (int)threadIdx.x & 31;
    float *lds = lds_wave[wave];
    float *tile = lds + half * 576;               // the granule's 576 lines, natural order
    float *S = lds + kSBase + half * (18 * kSStride);  // S[slot][32]: dct32 transpose
#if SYM_MP3_OTILE
    float *O = lds + kOBase + half * 576;              // the granule's PCM, natural order
#endif
    cf32p mc = as_const(tb.mp3_consts);

    // Window coefficients of sample index i = hl: D[64j + i] (j = 0..7), D[64j + 32 + i], as one 64-byte LDS row
    // per i.  They are only live during the window pass, where they are re-read from LDS (4 x b128) every granule.
    float *dwt = lds_tab;
    if ((int)threadIdx.x < 32) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
#if SYM_MP3_PACKED
            dwt[hl * kDwStride + j] = __uint_as_float(__float_as_uint(tb.mp3_consts[MP3C_SYNTH_D + 64 * j + hl]) ^ vmapx(hl).fsign);
            dwt[hl * kDwStride + 8 + j] = -tb.mp3_consts[MP3C_SYNTH_D + 64 * j + 32 + hl];
#else
            dwt[hl * kDwStride + j] = tb.mp3_consts[MP3C_SYNTH_D + 64 * j + hl];
            dwt[hl * kDwStride + 8 + j] = tb.mp3_consts[MP3C_SYNTH_D + 64 * j + 32 + hl];
#endif
        }
    }
    float *imdct_win = lds_tab + kDwFloats;  // the four 36-entry IMDCT windows (hybrid_synthesis.rs:31-101)
    for (int i = (int)threadIdx.x; i < 4 * 36; i += 64 * kWgWaves) imdct_win[i] = tb.mp3_consts[MP3C_IMDCT_WIN + i];
    float *pow43_lo = lds_tab + kTabFloats;                                                 // (FUSED only)
    uint8_t *front_maps = reinterpret_cast<uint8_t *>(lds_tab + kTabFloats + kFrontPow);    // (FUSED only)
    float *p2_lo = lds_tab + kTabFloats + kFrontPow + kFrontMapFloats;                      // (FUSED only)
    SfbEdges *e_lds = reinterpret_cast<SfbEdges *>(lds_tab + kTabFloats + kFrontPow + kFrontMapFloats + kP2Len);  // (FUSED only)
    const int mixed_switch = edges.mixed_switch;
    float *fw = lds + kWaveFloats;                                                          // (FUSED only)
    if (FUSED) {
        for (int i = (int)threadIdx.x; i < 8207; i += 64 * kWgWaves) pow43_lo[i] = tb.mp3_pow43[i];
        const uint32_t *msrc = reinterpret_cast<const uint32_t *>(tb.mp3_band_map + (size_t)sr * 4 * 576);
        for (int i = (int)threadIdx.x; i < kFrontMapFloats; i += 64 * kWgWaves) reinterpret_cast<uint32_t *>(front_maps)[i] = msrc[i] << kMapShift;  // (bands < 64: no carry between bytes)
        for (int i = (int)threadIdx.x; i < kP2Len; i += 64 * kWgWaves) p2_lo[i] = tb.mp3_pow2ab[i];
        if (threadIdx.x == 0) *e_lds = edges;
    }
    if (kWgWaves > 1 || FUSED) __syncthreads();  // the only workgroup-wide barrier; wavefronts are independent from here on

    // unfused: every half-wave takes its own (chain, segment); fused: the wavefront takes a (pair, segment), half = channel
    const unsigned witem = blockIdx.x * (unsigned)kWgWaves + (unsigned)wave;
    const unsigned item = FUSED ? witem : witem * 2u + (unsigned)half;
    const bool in_range = item < n_chains * segs_per_chain;  // (fused: n_chains is the number of pairs)
    const unsigned unit = in_range ? item / segs_per_chain : 0, seg = in_range ? item % segs_per_chain : 0;
    int fchain = 0;
    if (FUSED && in_range) fchain = pair_chains[2 * unit + (unsigned)half];  // -1: a mono stream's missing second channel
    if (FUSED && fchain >= (int)chain_bound) fchain = -1;  // (unit_chains lives in device memory: an index outside the batch is skipped, not followed)
    const bool live = in_range && fchain >= 0;
    const unsigned chain = FUSED ? (live ? (unsigned)fchain : 0u) : unit;
    bool pair_live = false;  // both channels present: joint stereo can apply
    if (FUSED) pair_live = in_range && pair_chains[2 * unit] >= 0 && pair_chains[2 * unit + 1] >= 0 && pair_chains[2 * unit] < (int)chain_bound &&
                           pair_chains[2 * unit + 1] < (int)chain_bound;
    const unsigned g_begin = seg * seg_len;
    const unsigned g_end = live ? min(g_begin + seg_len, granules_per_chain) : g_begin;
    const VMapX vm = vmapx(hl);

    // ---- incoming state.  oA[16 + r] = V_r[i], oB[16 + r] = V_r[32 + i] for the previous granules' time slots r < 0
    float overlap[18];
#if SYM_MP3_PACKED
    v2f PA[17], PB[18];
#define SYM_HA(t) PA[(t) / 2][(t) & 1]              /* HA[t], t = 0..33 */
#define SYM_HB(t) PB[((t) + 1) / 2][((t) + 1) & 1]  /* HB[t], t = -1..34 */
#pragma unroll
    for (int k = 0; k < 17; ++k) PA[k] = v2f{0.0f, 0.0f};
#pragma unroll
    for (int k = 0; k < 18; ++k) PB[k] = v2f{0.0f, 0.0f};
#else
    float oA[kHistOld], oB[kHistOld];
#pragma unroll
    for (int r = 0; r < kHistOld; ++r) oA[r] = oB[r] = 0.0f;
#endif
    const bool first_seg = g_begin == 0;
#pragma unroll
    for (int i = 0; i < 18; ++i) overlap[i] = 0.0f;
    if (live && first_seg) {
#pragma unroll
        for (int i = 0; i < 18; ++i) overlap[i] = overlap_in[(size_t)chain * 576 + 18 * hl + i];
        const int v_front = vfront_in[chain] & 15;
        const float *vv = vvec_in + (size_t)chain * 1024;
#pragma unroll
        for (int m = 1; m <= kHistOld; ++m) {  // slot -m sits in FIFO row (v_front + m) & 15 (synthesis.rs:335)
            const float *row = vv + 64 * ((v_front + m) & 15);
#if SYM_MP3_PACKED
            SYM_HA(kHistOld - m) = __uint_as_float(__float_as_uint(row[hl]) ^ vm.fsign);
            SYM_HB(kHistOld - m) = -row[32 + hl];
#else
            oA[kHistOld - m] = row[hl];
            oB[kHistOld - m] = row[32 + hl];
#endif
        }
    }

    // halo: g_begin-2 (overlap only), g_begin-1 (history only).  Per half-wave the walk is described by three small
    // integers (rounds of its own, first round that rebuilds the history, first round that emits) and ONE running
    // 32-bit granule index `gi` = chain * granules_per_chain + g (launch_mp3 checks that it fits); every global
    // address is formed from it where it is used -- 64-bit pointers carried through the loop cost two VGPRs each.
    const unsigned g_first = first_seg ? 0u : g_begin - 2u;
    const int my_rounds = live ? (int)(g_end - g_first) : 0;
    const int hist_from = first_seg ? 0 : 1, emit_from = first_seg ? 0 : 2;
    int rounds = my_rounds;  // both halves run the same number of rounds (wave-uniform loop)
    {
        const int other = __shfl(rounds, (int)((threadIdx.x & 63u) ^ 32u));
        rounds = rounds > other ? rounds : other;
    }
    rounds = __builtin_amdgcn_readfirstlane(rounds);
    unsigned gi = chain * granules_per_chain + g_first;

    if (hl == 0) {  // what the epilogue needs of the above, parked in LDS (see there)
        unsigned *meta_w = reinterpret_cast<unsigned *>(lds + kMetaBase) + 2 * half;
        meta_w[0] = chain;
        meta_w[1] = (live && g_end == granules_per_chain) ? 1u : 0u;
    }

    // The granule's 576 lines as 144 float4: lane hl holds float4 hl + 32 q, q = 0..3, and (hl < 16) float4 128 + hl.
    // 16 B per lane: 512 B coalesced per half-wave and instruction (narrow accesses are issue-bound, not HBM-bound).
    // side info of the next granule, kept as the raw 32-bit word: any arithmetic on it here would make the
    // compiler wait for the load (and every line load issued before it) right after issuing them
    static_assert(sizeof(symaccel_mp3_side) == 4, "side info is fetched as one dword");
    const uint32_t *side_raw = reinterpret_cast<const uint32_t *>(side);
    uint32_t sd_next = 0;
    float4 line[5];
#pragma unroll
    for (int q = 0; q < 5; ++q) line[q] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    // fused: the granule's quantised samples (9 dwords per lane) and one dword of its records (lanes 0..12: the channel's
    // requantize record, lanes 13..24: the pair's joint-stereo record), fetched like the lines -- one granule ahead
    uint32_t qw[9] = {0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u};
    uint32_t dq = 0u;
    unsigned sti = FUSED ? unit * granules_per_chain + g_first : 0u;  // index of the pair's joint-stereo record of granule gi
    auto fetch_front = [&](unsigned g_idx, unsigned st_idx, int lane) {
        fetch_quant(quant + (size_t)g_idx * 576, lane, qw);
        if (lane < 13)
            dq = reinterpret_cast<const uint32_t *>(rq_desc + g_idx)[lane];
        else if (lane < 25)
            dq = reinterpret_cast<const uint32_t *>(st_desc + st_idx)[lane - 13];
    };
    if (my_rounds > 0) {
        if (FUSED)
            fetch_front(gi, sti, hl);
        else
            fetch_granule(xr + (size_t)gi * 576, hl, line);
        sd_next = side_raw[gi];
    }
#if SYM_MP3_PREFETCH2
    // two granules in flight: round r consumes `line` (granule r), `line2` holds granule r + 1 and granule r + 2 is
    // requested into it once it has moved up -- a granule then has two rounds, not ~60 % of one, to arrive
    uint32_t sd_next2 = 0;
    float4 line2[5];
#pragma unroll
    for (int q = 0; q < 5; ++q) line2[q] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    if (my_rounds > 1) {
        fetch_granule(xr + (size_t)(gi + 1u) * 576, hl, line2);
        sd_next2 = side_raw[gi + 1u];
    }
#endif

#if SYM_MP3_SINK
    if (!FUSED) {   // round 0's granule -> LDS tile (later rounds: at the end of the round before; fused: by the pre-round below)
        float4 *t4 = reinterpret_cast<float4 *>(tile);
#pragma unroll
        for (int q = 0; q < 4; ++q) t4[hl + 32 * q] = line[q];
        if (hl < 16) t4[128 + hl] = line[4];
    }
    uint32_t sd_cur = sd_next;
    asm volatile("" : "+v"(sd_cur));
#endif
    // fused: the loop starts with a PRE-ROUND (r = -1) that runs nothing but the front on granule 0 -- the front is a few hundred
    // instructions and has exactly one copy in the instruction stream this way
    if (FUSED) --gi;
    for (int r = FUSED ? -1 : 0; r < rounds; ++r, ++gi) {
        unsigned hlg = (unsigned)hl;  // the lane's offset in global addresses, opaque for the same reason as gi:
        asm volatile("" : "+v"(gi), "+v"(hlg));  // keeps the address arithmetic in the loop (see above)
        const bool active = r < my_rounds;
        const bool need_hist = active && r >= hist_from;  // halo granule g_begin-2 only rebuilds overlap
        const bool emit = active && r >= emit_from;

        do {  // (the round proper; skipped by the fused kernel's pre-round)
        if (FUSED && r < 0) break;
        int bt = 0, mixed = 0, rzero = 0;
        if (active) {
#if SYM_MP3_SINK
            const uint32_t sd = sd_cur;
#else
            const uint32_t sd = sd_next;  // fetched one granule ahead, with the lines (little-endian struct layout)
#endif
            bt = (int)(sd & 0xffu);
            mixed = (sd & 0xff00u) ? 1 : 0;
            rzero = (int)(sd >> 16) > 576 ? 576 : (int)(sd >> 16);
        }
        // ---- granule -> LDS tile (natural order), then lane sb gathers its 18 lines
#if !SYM_MP3_SINK
        if (active) {
            float4 *t4 = reinterpret_cast<float4 *>(tile);
#pragma unroll
            for (int q = 0; q < 4; ++q) t4[hl + 32 * q] = line[q];
            if (hl < 16) t4[128 + hl] = line[4];
        }
#endif
        wave_sync();
        float y[18];
#pragma unroll
        for (int i = 0; i < 18; ++i) y[i] = 0.0f;
        if (active) {
            if (bt == SYMACCEL_MP3_SHORT) {
                // reorder (hybrid_synthesis.rs:153-215) applied as a gather through the precomputed source map
                const int32_t *map = tb.mp3_reorder_map + (size_t)(sr * 2 + mixed) * 576;
                const int32_t *ends = tb.mp3_reorder_end + (size_t)(sr * 2 + mixed) * 577;
                const int r_start = ends[0], r_end = ends[rzero];
                rzero = rzero > r_end ? rzero : r_end;
                int hls = hl;  // (opaque: the 18 line indices are formed here, in the rare path, not kept in VGPRs across the loop)
                asm volatile("" : "+v"(hls));
#pragma unroll
                for (int i = 0; i < 18; ++i) {
                    const int idx = 18 * hls + i;
                    y[i] = tile[(idx >= r_start && idx < r_end) ? map[idx] : idx];
                }
            } else {
                const float2 *t2 = reinterpret_cast<const float2 *>(tile + 18 * hl);  // 72 B lane stride: conflict-free b64
#pragma unroll
                for (int k = 0; k < 9; ++k) {
                    const float2 v = t2[k];
                    y[2 * k] = v.x;
                    y[2 * k + 1] = v.y;
                }
            }
        }
        // ---- antialias (hybrid_synthesis.rs:218-277): boundary sb pairs lane sb-1's y[17-i] with lane sb's y[i]
        {
            int lim = 0;  // boundaries 1 .. lim-1 are processed
            if (active && !(bt == SYMACCEL_MP3_SHORT && !mixed)) {
                const int sb_limit = bt == SYMACCEL_MP3_SHORT ? 2 : 32;
                lim = rzero / 18 + 2;
                lim = lim < sb_limit ? lim : sb_limit;
                lim = lim < 32 ? lim : 32;
                rzero = 18 * lim;
            }
            const bool below = hl >= 1 && hl < lim;      // boundary hl (with lane hl-1) is active
            const bool above = hl + 1 < lim;             // boundary hl+1 (with lane hl+1) is active
            // all 16 neighbour exchanges are issued back to back (one wait), then the butterflies
            float lower_nb[8], upper_nb[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                lower_nb[i] = __shfl_up(y[17 - i], 1);   // lane sb-1's lower element
                upper_nb[i] = __shfl_down(y[i], 1);      // lane sb+1's upper element
            }
            __builtin_amdgcn_sched_barrier(0);
            float lo_new[8], up_new[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                constexpr const float *cs_t = kMp3Lit + MP3C_CS, *ca_t = kMp3Lit + MP3C_CA;
                const float cs = cs_t[i], ca = ca_t[i];
                up_new[i] = y[i] * cs + lower_nb[i] * ca;       // upper' = upper * cs + lower * ca
                lo_new[i] = y[17 - i] * cs - upper_nb[i] * ca;  // lower' = lower * cs - upper * ca
            }
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                y[i] = below ? up_new[i] : y[i];
                y[17 - i] = above ? lo_new[i] : y[17 - i];
            }
        }
        // ---- hybrid synthesis (hybrid_synthesis.rs:280-359): lane = sub-band
        if (active) {
            const int sb = hl;
            const int sb_limit = (rzero + 17) / 18;
            const int sb_split = bt == SYMACCEL_MP3_SHORT ? (mixed ? 2 : 0) : 32;
            if (sb >= sb_limit) {
#pragma unroll
                for (int i = 0; i < 18; ++i) {
                    y[i] = overlap[i];
                    overlap[i] = 0.0f;
                }
            } else if (sb < sb_split) {
                const int wi = bt == SYMACCEL_MP3_START ? 1 : (bt == SYMACCEL_MP3_END ? 3 : 0);
                imdct36(y, overlap, imdct_win + 36 * wi);
            } else {
                imdct12_win(y, overlap, mc + MP3C_IMDCT_WIN + 36 * 2);
            }
            // frequency_inversion (hybrid_synthesis.rs:458-485)
            if (sb & 1) {
#pragma unroll
                for (int i = 1; i < 18; i += 2) y[i] = -y[i];
            }
        }
#if SYM_MP3_PREFETCH2
#pragma unroll
        for (int q = 0; q < 5; ++q) line[q] = line2[q];
        sd_next = sd_next2;
        if (r + 2 < my_rounds) {
            fetch_granule(xr + (size_t)(gi + 2u) * 576, (int)hlg, line2);
            sd_next2 = side_raw[gi + 2u];
        }
#elif SYM_MP3_SINK
        {   // prefetch the next granule (granule 0 where there is none: always a valid address, never used)
            const unsigned gn = r + 1 < my_rounds ? gi + 1u : 0u;
            if (FUSED) {
                ++sti;
                fetch_front(gn, r + 1 < my_rounds ? sti : 0u, (int)hlg);
            } else {
                fetch_granule(xr + (size_t)gn * 576, (int)hlg, line);
            }
            sd_next = side_raw[gn];
        }
#else
        if (r + 1 < my_rounds) {  // prefetch the next granule; it lands during the dct32 and window passes
            fetch_granule(xr + (size_t)(gi + 1u) * 576, (int)hlg, line);
            sd_next = side_raw[gi + 1u];
        }
#endif
        wave_sync();  // the previous granule's window pass has read S
        if (need_hist) {
#pragma unroll
            for (int b = 0; b < 18; ++b) S[b * kSStride + hl] = y[b];
        }
        wave_sync();
        // ---- 18 x dct32 (synthesis.rs:165-245): lane = time slot, row in / row out
        if (need_hist && hl < 18) {
            float d[32];
            float4 *row = reinterpret_cast<float4 *>(S + hl * kSStride);
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const float4 v = row[k];
                d[4 * k] = v.x;
                d[4 * k + 1] = v.y;
                d[4 * k + 2] = v.z;
                d[4 * k + 3] = v.w;
            }
            dct_lee<32>(d);
#pragma unroll
            for (int k = 0; k < 8; ++k) row[k] = make_float4(d[4 * k], d[4 * k + 1], d[4 * k + 2], d[4 * k + 3]);
            row[8] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);  // the zero column V[16] reads (the granule tile overwrote it)
        }
        wave_sync();
        // ---- windowing (synthesis.rs:309-324), one time slot at a time: fetch the slot's two V entries for
        // this lane's sample index (synthesis.rs:247-263), then 16 taps with every operand in registers.
        // (Lanes / granules that do not need the history read stale LDS into nA/nB and never use it.)
        float dw0[8], dw1[8];
        {
            const float4 *r4 = reinterpret_cast<const float4 *>(dwt + hl * kDwStride);
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                const float4 a = r4[k], c = r4[2 + k];
                dw0[4 * k] = a.x; dw0[4 * k + 1] = a.y; dw0[4 * k + 2] = a.z; dw0[4 * k + 3] = a.w;
                dw1[4 * k] = c.x; dw1[4 * k + 1] = c.y; dw1[4 * k + 2] = c.z; dw1[4 * k + 3] = c.w;
            }
        }
#if SYM_MP3_SINK && !SYM_MP3_OTILE
        float *outp;
        {
            unsigned tid3 = threadIdx.x;
            asm volatile("" : "+v"(tid3));  // (formed here, not carried through the loop)
            float *sink_lane = sink + (size_t)((blockIdx.x * (unsigned)kWgWaves + (tid3 >> 6)) % (unsigned)kSinkSlots) * kSinkSlotFloats +
                               ((tid3 >> 5) & 1u) * 576u + (tid3 & 31u);
            outp = emit ? pcm + (size_t)gi * 576 + hlg : sink_lane;
        }
#endif
#if SYM_MP3_PACKED
        // this granule's 18 slots: HA[16 + b] = S[b][fcol], HB[16 + b] = S[b][scol], read as the pairs they are kept in (one
        // ds_read2_b32 each, all issued up front; the sums below wait for them pair by pair)
#pragma unroll
        for (int q = 0; q < 9; ++q) PA[8 + q] = v2f{S[(2 * q) * kSStride + vm.fcol], S[(2 * q + 1) * kSStride + vm.fcol]};
        PB[8][1] = S[vm.scol];
#pragma unroll
        for (int q = 1; q < 9; ++q) PB[8 + q] = v2f{S[(2 * q - 1) * kSStride + vm.scol], S[(2 * q) * kSStride + vm.scol]};
        PB[17][0] = S[17 * kSStride + vm.scol];
        // kPairGroup slot pairs advance together: a packed instruction that reads the result of the instruction right before it
        // costs a wait state (the compiler pads with s_nop, an issue slot each); interleaved chains never are back to back
        constexpr int kPairGroup = SYM_MP3_PAIR_GROUP;
        static_assert(9 % kPairGroup == 0, "whole groups");
#pragma unroll
        for (int q0 = 0; q0 < 9; q0 += kPairGroup) {  // time slots 2q and 2q + 1
            v2f acc[kPairGroup];
#pragma unroll
            for (int g = 0; g < kPairGroup; ++g) acc[g] = v2f{0.0f, 0.0f};
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const v2f d0 = {dw0[j], dw0[j]}, d1 = {dw1[j], dw1[j]};
                v2f pa[kPairGroup], pb[kPairGroup];
#pragma unroll
                for (int g = 0; g < kPairGroup; ++g) pa[g] = PA[8 + q0 + g - j] * d0;
#pragma unroll
                for (int g = 0; g < kPairGroup; ++g) acc[g] += pa[g];
#pragma unroll
                for (int g = 0; g < kPairGroup; ++g) pb[g] = PB[8 + q0 + g - j] * d1;
#pragma unroll
                for (int g = 0; g < kPairGroup; ++g) acc[g] += pb[g];
            }
#pragma unroll
            for (int g = 0; g < kPairGroup; ++g) {
                const int q = q0 + g;
#if SYM_MP3_OTILE
                O[32 * (2 * q) + hl] = acc[g][0];
                O[32 * (2 * q + 1) + hl] = acc[g][1];
#elif SYM_MP3_SINK
                st_stream(outp + 32 * (2 * q), acc[g][0]);
                st_stream(outp + 32 * (2 * q + 1), acc[g][1]);
#else
                if (emit) {
                    st_stream(pcm + (size_t)gi * 576 + 32 * (2 * q) + hlg, acc[g][0]);
                    st_stream(pcm + (size_t)gi * 576 + 32 * (2 * q + 1) + hlg, acc[g][1]);
                }
#endif
            }
        }
#else
        float nA[18], nB[18];
        // the two LDS reads of slot b + 1 are issued before the taps of slot b (the store's branch per slot otherwise
        // pins each read directly in front of its first use: 18 exposed LDS round trips per granule)
        float ra = S[vm.fcol], rb = S[vm.scol];
        // kSlotGroup time slots per store branch: the 16-tap sums of a group are independent chains in ONE basic block (the
        // per-slot `if (emit)` used to fence every chain off from the next)
        constexpr int kSlotGroup = SYM_MP3_SLOT_GROUP;
        static_assert(18 % kSlotGroup == 0, "whole groups");
#pragma unroll
        for (int b0 = 0; b0 < 18; b0 += kSlotGroup) {
            float accs[kSlotGroup];
#pragma unroll
            for (int g = 0; g < kSlotGroup; ++g) {
                const int b = b0 + g;
                nA[b] = __uint_as_float(__float_as_uint(ra) ^ vm.fsign);  // V[i]
                nB[b] = -rb;                                               // V[32 + i]
                if (b + 1 < 18) {
                    ra = S[(b + 1) * kSStride + vm.fcol];
                    rb = S[(b + 1) * kSStride + vm.scol];
                }
                float acc = 0.0f;
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const int ra_ = b - 2 * j, rb_ = b - 2 * j - 1;
                    acc += (ra_ >= 0 ? nA[ra_ >= 0 ? ra_ : 0] : oA[ra_ < 0 ? kHistOld + ra_ : 0]) * dw0[j];
                    acc += (rb_ >= 0 ? nB[rb_ >= 0 ? rb_ : 0] : oB[rb_ < 0 ? kHistOld + rb_ : 0]) * dw1[j];
                }
                accs[g] = acc;
            }
#if SYM_MP3_OTILE
#pragma unroll
            for (int g = 0; g < kSlotGroup; ++g) O[32 * (b0 + g) + hl] = accs[g];
#elif SYM_MP3_SINK
#pragma unroll
            for (int g = 0; g < kSlotGroup; ++g) st_stream(outp + 32 * (b0 + g), accs[g]);
#else
            if (emit) {
#pragma unroll
                for (int g = 0; g < kSlotGroup; ++g) st_stream(pcm + (size_t)gi * 576 + 32 * (b0 + g) + hlg, accs[g]);
            }
#endif
        }
#endif
#if SYM_MP3_OTILE
        wave_sync();
        if (emit) {  // the granule's 144 float4, lane hl stores float4 hl + 32 q (and 128 + hl for hl < 16)
            const float4 *o4 = reinterpret_cast<const float4 *>(O);
            float4 *dst = reinterpret_cast<float4 *>(pcm + (size_t)gi * 576) + hlg;
            float4 v[5];
#pragma unroll
            for (int q = 0; q < 4; ++q) v[q] = o4[hl + 32 * q];
            v[4] = o4[128 + (hl & 15)];
#pragma unroll
            for (int q = 0; q < 4; ++q) st_stream(dst + 32 * q, v[q]);
            if (hl < 16) st_stream(dst + 128, v[4]);
        }
#endif
        wave_sync();  // the window pass has read S; the next round's tile goes to the same LDS
        // ---- slide the history: slots 2..17 of this granule become slots -16..-1
        if (need_hist) {
#if SYM_MP3_PACKED
#pragma unroll
            for (int k = 0; k < 8; ++k) PA[k] = PA[k + 9];   // HA[t] <- HA[t + 18]
#pragma unroll
            for (int k = 0; k < 9; ++k) PB[k] = PB[k + 9];   // HB[t] <- HB[t + 18], up to HB[15] = PB[8].x
#else
#pragma unroll
            for (int m = 0; m < kHistOld; ++m) {
                oA[m] = nA[2 + m];
                oB[m] = nB[2 + m];
            }
#endif
        }
        } while (0);
#if SYM_MP3_SINK
        if (FUSED) {
            mp3_front(tb, e_lds, mixed_switch, qw, dq, hl, half, pair_live, pow43_lo, front_maps, p2_lo, fw, tile);
            sd_cur = sd_next;
            asm volatile("" : "+v"(sd_cur));
        } else {
            float4 *t4 = reinterpret_cast<float4 *>(tile);
#pragma unroll
            for (int q = 0; q < 4; ++q) t4[hl + 32 * q] = line[q];
            if (hl < 16) t4[128 + hl] = line[4];
            sd_cur = sd_next;
            asm volatile("" : "+v"(sd_cur));  // (the side word is waited for HERE, behind the lines, not at the loop header)
        }
#endif
    }

    // ---- outgoing state (only the segment that ends the chain).  The chain index and the addresses derived from it are
    // re-read here from LDS through an opaque copy of the thread index: kept live across the main loop they (and the
    // reciprocal of the division that produced them) cost seven VGPRs, which at the 168-register budget of three
    // wavefronts per SIMD were spilled to scratch around the loop.
    wave_sync();
    unsigned tid2 = threadIdx.x;
    asm volatile("" : "+v"(tid2));
    const int hl2 = (int)(tid2 & 31u);
    const unsigned *meta = reinterpret_cast<const unsigned *>(lds_wave[tid2 >> 6] + kMetaBase) + 2 * ((tid2 >> 5) & 1u);
    const unsigned chain2 = meta[0];
    if (meta[1] != 0u) {
#pragma unroll
        for (int i = 0; i < 18; ++i) overlap_out[(size_t)chain2 * 576 + 18 * hl2 + i] = overlap[i];
        // v_vec[16][64] + v_front exactly as the reference leaves them: v_front moves back one row per time
        // slot (synthesis.rs:335) and row (v_front + m) & 15 holds slot -m, m = 1..16.
        const int vf0 = vfront_in[chain2] & 15;
        const int vf_final = (int)(((unsigned)vf0 + 15u * 18u * granules_per_chain) & 15u);
        float *vv = vvec_out + (size_t)chain2 * 1024;
#pragma unroll
        for (int m = 1; m <= kHistOld; ++m) {
            float *row = vv + 64 * ((vf_final + m) & 15);
#if SYM_MP3_PACKED
            row[hl2] = __uint_as_float(__float_as_uint(SYM_HA(kHistOld - m)) ^ vmapx(hl2).fsign);
            row[32 + hl2] = -SYM_HB(kHistOld - m);
#else
            row[hl2] = oA[kHistOld - m];
            row[32 + hl2] = oB[kHistOld - m];
#endif
        }
        if (hl2 == 0) vfront_out[chain2] = vf_final;
    }
}

}  // namespace

int launch_mp3(symaccel_ctx *ctx, const float *d_xr, const symaccel_mp3_side *d_side, int sr,
               const float *d_overlap_in, const float *d_vvec_in, const int32_t *d_vfront_in, float *d_overlap_out,
               float *d_vvec_out, int32_t *d_vfront_out, float *d_pcm, size_t n_chains, size_t granules_per_chain) {
    // (the kernel indexes granules of the whole batch with 32 bits: 2^32 granules are 9.9 TB of spectra)
    if (granules_per_chain > 0x3fffffffu || n_chains > 0x3fffffffu || n_chains * granules_per_chain > 0xffffffffu)
        return SYMACCEL_ERR_INVALID_ARG;
    // (the two-granule halo needs segment starts >= 2)
    const unsigned seg = choose_segment(ctx, n_chains, granules_per_chain, 4 * SYM_MP3_WAVES, 2, 2, 2);
    const size_t segs = (granules_per_chain + seg - 1) / seg;
    const size_t items = n_chains * segs;
    const size_t grid = (items + 2 * kWgWaves - 1) / (2 * kWgWaves);
    if (grid > 0x7fffffffu) return SYMACCEL_ERR_INVALID_ARG;
    static_assert((size_t)kSinkSlots * kSinkSlotFloats * sizeof(float) <= kSinkBytes, "sink slots");
    void *sink = nullptr;
    SYM_TRY(ctx_sink(ctx, &sink));
    hipLaunchKernelGGL((mp3_synth_kernel<kWgWaves, false>), dim3((unsigned)grid), dim3(64 * kWgWaves), 0, ctx->stream, ctx->dev, d_xr, d_side, sr,
                       d_overlap_in, d_vvec_in, d_vfront_in, d_overlap_out, d_vvec_out, d_vfront_out, d_pcm,
                       static_cast<float *>(sink), (unsigned)n_chains, (unsigned)granules_per_chain, seg, (unsigned)segs,
                       (const int16_t *)nullptr, (const symaccel_mp3_requant *)nullptr, (const symaccel_mp3_stereo *)nullptr,
                       (const int32_t *)nullptr, SfbEdges{}, (unsigned)n_chains);
    SYM_GPU(ctx, hipGetLastError());
    return SYMACCEL_OK;
}

// int16 Huffman samples + records -> PCM in one kernel (the fused front, see mp3_front).  d_pair_chains[n_pairs][2]: the two
// chains of each stream (the second may be -1: a mono stream); every chain of the batch appears exactly once.
int launch_mp3_decode(symaccel_ctx *ctx, const int16_t *d_quant, const symaccel_mp3_requant *d_rq_desc, const int32_t *d_pair_chains,
                      const symaccel_mp3_stereo *d_st_desc, size_t n_pairs, const symaccel_mp3_side *d_side, int sr,
                      const float *d_overlap_in, const float *d_vvec_in, const int32_t *d_vfront_in, float *d_overlap_out,
                      float *d_vvec_out, int32_t *d_vfront_out, float *d_pcm, size_t n_chains, size_t granules_per_chain) {
#if SYM_MP3_SINK && !SYM_MP3_OTILE && !SYM_MP3_PREFETCH2
    if (granules_per_chain > 0x3fffffffu || n_chains > 0x3fffffffu || n_chains * granules_per_chain > 0xffffffffu || n_pairs > n_chains)
        return SYMACCEL_ERR_INVALID_ARG;
    constexpr int kFw = SYM_MP3_FUSED_WG_WAVES;  // wavefronts per workgroup: they share the 6.4 KiB of front tables
    // (a wavefront carries one pair: twice the units of work per chain-segment of the unfused kernel)
    const unsigned seg = choose_segment(ctx, n_pairs, granules_per_chain, 4 * SYM_MP3_FUSED_WAVES, 1, 2, 2);
    const size_t segs = (granules_per_chain + seg - 1) / seg;
    const size_t items = n_pairs * segs;
    const size_t grid = (items + kFw - 1) / kFw;
    if (grid > 0x7fffffffu) return SYMACCEL_ERR_INVALID_ARG;
    void *sink = nullptr;
    SYM_TRY(ctx_sink(ctx, &sink));
    const SfbEdges e = make_sfb_edges(host_tables(), sr);
    hipLaunchKernelGGL((mp3_synth_kernel<kFw, true>), dim3((unsigned)grid), dim3(64 * kFw), 0, ctx->stream, ctx->dev, (const float *)nullptr, d_side,
                       sr, d_overlap_in, d_vvec_in, d_vfront_in, d_overlap_out, d_vvec_out, d_vfront_out, d_pcm, static_cast<float *>(sink),
                       (unsigned)n_pairs, (unsigned)granules_per_chain, seg, (unsigned)segs, d_quant, d_rq_desc, d_st_desc, d_pair_chains, e,
                       (unsigned)n_chains);
    SYM_GPU(ctx, hipGetLastError());
    return SYMACCEL_OK;
#else
    (void)ctx; (void)d_quant; (void)d_rq_desc; (void)d_pair_chains; (void)d_st_desc; (void)n_pairs; (void)d_side; (void)sr; (void)d_overlap_in;
    (void)d_vvec_in; (void)d_vfront_in; (void)d_overlap_out; (void)d_vvec_out; (void)d_vfront_out; (void)d_pcm; (void)n_chains;
    (void)granules_per_chain;
    return SYMACCEL_ERR_UNSUPPORTED;  // (the fused front is written for the product's schedule, SYM_MP3_VARIANT 4)
#endif
}

}  // namespace symaccel
