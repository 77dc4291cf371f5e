// Internal declarations shared by the host side of libsymaccel (not part of the ABI).
#pragma once

#include <cstddef>
#include <cstdint>
#include <map>
#include <string>
#include <utility>
#include <vector>

#include "../../include/symaccel.h"
#include <hip/hip_runtime.h>

namespace symaccel {

struct cpx {
    float re, im;
};

constexpr int kMp3Pow2abMinE = -1300, kMp3Pow2abLen = 1346;  // exponents a - b of requantize.rs:280, 343
constexpr int kMp3Unscaled = 39;                             // band-map value of lines no band covers

// Constant tables, generated on the host with the libm calls the reference uses (tables.cpp).
struct HostTables {
    // AAC (symphonia-codec-aac/src/aac/window.rs:28-63, dsp.rs:34-54)
    std::vector<float> aac_kbd_long, aac_kbd_short, aac_sine_long, aac_sine_short;
    std::vector<cpx> aac_tw_long, aac_tw_short;  // Imdct twiddles N=1024 (scale 1/2048), N=128 (1/256)
    // FFT (symphonia-core/src/dsp/fft/no_simd.rs:16-36, 307-324, 374-383)
    std::vector<cpx> fft_merge;   // W64 | W128 | ... | W4096 back to back (offset of W_n = n/2 - 32)
    cpx small16[8], small32[16];  // combine constants of fft16 / fft32 in "general form"
    uint8_t small16_form[8], small32_form[16];  // 0 general, 1 identity, 2 multiply by -i
    // MP3 (hybrid_synthesis.rs:53-149, 611-630, 668-678, 722-730; synthesis.rs:13-142, 354-396)
    float mp3_imdct_win[4][36];
    float mp3_cos12[6][6];
    float mp3_cs[8], mp3_ca[8];
    float mp3_dct_iv_scale[18], mp3_sdct18_scale[9], mp3_sdct9_d[7];
    float mp3_cos16[16], mp3_cos8[8], mp3_cos4[4], mp3_cos2[2], mp3_cos1;
    float mp3_synth_d[512];
    int32_t mp3_sfb_short[9][40];
    int32_t mp3_sfb_mixed[9][40];
    int32_t mp3_sfb_mixed_len[9];
    int32_t mp3_sfb_switch[9];
    // MP3 requantisation (requantize.rs:28-31, 256-257, 280, 343; layer3/common.rs:9-56)
    int32_t mp3_sfb_long[9][23];
    float mp3_pow43[8207];
    float mp3_pow2ab[kMp3Pow2abLen];
    uint8_t mp3_band_map[9][4][576];  // [sample rate][long, short, mixed (requantize), mixed (stereo)][line] -> band / slot
                                      // index; kMp3Unscaled = no band (requantize's mixed map only)
    float mp3_is_ratios[7 + 64][2];   // INTENSITY_STEREO_RATIOS_MPEG1[7] | _MPEG2[2][32] as (left, right) (stereo.rs:31-118)
    // Vorbis (floor.rs:21-112)
    float vorbis_floor1_db[256];
};

const HostTables &host_tables();
void make_imdct_twiddles(int n, double scale, cpx *dst);  // mdct.rs:45-54
void make_fft_twiddles(int n, cpx *dst);                  // no_simd.rs:16-36
void make_vorbis_window(int bs, float *dst);              // vorbis window.rs:11-24

inline size_t fft_merge_offset(int n) { return (size_t)(n / 2 - 32); }  // n >= 64

// Device-resident copy of the tables the kernels read (plain pointers, passed by value).
struct DevTables {
    const float *aac_kbd_long, *aac_kbd_short, *aac_sine_long, *aac_sine_short;
    const cpx *aac_tw_long, *aac_tw_short;
    const cpx *fft_merge;
    const cpx *small16, *small32;        // 8 + 16 complex
    const uint8_t *small16_form, *small32_form;
    const float *mp3_consts;             // packed, see Mp3ConstLayout
    const int32_t *mp3_reorder_map;      // [9 sample rates][2 (plain, mixed)][576] source index
    const int32_t *mp3_reorder_end;      // [9][2][577]: reorder end index `i` for each input rzero
    const float *vorbis_floor1_db;
    const float *mp3_pow43, *mp3_pow2ab;
    const uint8_t *mp3_band_map;         // [9][4][576]
    const float *mp3_is_ratios;          // [71][2]
};

// Offsets (in floats) inside DevTables::mp3_consts.
enum Mp3ConstLayout {
    MP3C_IMDCT_WIN = 0,    // 4*36
    MP3C_COS12 = 144,      // 36
    MP3C_CS = 180,         // 8
    MP3C_CA = 188,         // 8
    MP3C_DCT_IV = 196,     // 18
    MP3C_SDCT18 = 214,     // 9
    MP3C_SDCT9_D = 223,    // 7
    MP3C_COS16 = 230,      // 16
    MP3C_COS8 = 246,       // 8
    MP3C_COS4 = 254,       // 4
    MP3C_COS2 = 258,       // 2
    MP3C_COS1 = 260,       // 1
    MP3C_SYNTH_D = 264,    // 512
    MP3C_TOTAL = 776
};

struct ImdctPlan {
    int n;
    cpx *d_twiddle;  // n/2 complex on the device
};

}  // namespace symaccel

struct symaccel_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    hipStream_t own_stream = nullptr;
    int segment = 0;  // 0 = choose per launch (choose_segment)
    int n_cus = 256;  // compute units of the device (MI355X: 256)
    std::string last_error;
    symaccel::DevTables dev{};
    std::vector<void *> allocations;  // freed on destroy
    std::map<std::pair<int, uint64_t>, symaccel::ImdctPlan> imdct_plans;
    std::map<int, float *> vorbis_windows;  // bs -> device window (bs/2 floats)
    // growable device scratch (state double-buffering, per-block offsets)
    void *scratch = nullptr;
    size_t scratch_bytes = 0;
    // host <-> device staging of the *_pipelined entry points (stage.cpp), created on first use and kept: two copy streams,
    // six events, one growable device arena the chunk buffers are carved from
    hipStream_t stage_in = nullptr, stage_out = nullptr;
    hipEvent_t stage_events[6] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    void *stage_arena = nullptr;
    size_t stage_arena_bytes = 0;
    // small dedicated device buffers (not shared with `scratch`, whose growth synchronises the stream)
    void *sink = nullptr;  // 2 MiB nobody reads: target of the stores of wavefronts / lanes that must not emit (ctx_sink)
    void *alac_flags = nullptr;
    size_t alac_flags_bytes = 0;
};

namespace symaccel {

int ctx_fail(symaccel_ctx *ctx, hipError_t err, const char *where);
int ctx_alloc(symaccel_ctx *ctx, void **out, size_t bytes, bool tracked = true);
int ctx_scratch(symaccel_ctx *ctx, size_t bytes, void **out);
// The context's sink buffer (kSinkBytes, allocated on first use, never read).  Kernels whose loops must issue a FIXED number of
// stores per round (the vmcnt bookkeeping of aac.hip / mp3.hip) aim the stores of halo rounds at a slot of it.
constexpr size_t kSinkBytes = 2u << 20;
int ctx_sink(symaccel_ctx *ctx, void **out);
int ctx_upload(symaccel_ctx *ctx, const void *src, size_t bytes, const void **out);
int get_imdct_plan(symaccel_ctx *ctx, int n, double scale, const ImdctPlan **out);
unsigned choose_segment(const symaccel_ctx *ctx, size_t n_chains, size_t frames_per_chain, unsigned waves_per_cu,
                        unsigned items_per_wave, unsigned halo, unsigned min_seg);
int get_vorbis_window(symaccel_ctx *ctx, int bs, const float **out);

// Entry points run on the context's device and leave the caller's current device as they found it (a process that
// drives several GPUs through one thread must not have its "current device" moved by a library call).
class DeviceGuard {
public:
    explicit DeviceGuard(symaccel_ctx *ctx) {
        if (hipGetDevice(&prev_) != hipSuccess) prev_ = -1;
        if (prev_ != ctx->device) {
            const hipError_t e = hipSetDevice(ctx->device);
            if (e != hipSuccess) {
                status_ = ctx_fail(ctx, e, "hipSetDevice");
                prev_ = -1;
            } else {
                switched_ = true;
            }
        }
    }
    ~DeviceGuard() {
        if (switched_ && prev_ >= 0) (void)hipSetDevice(prev_);
    }
    DeviceGuard(const DeviceGuard &) = delete;
    DeviceGuard &operator=(const DeviceGuard &) = delete;
    bool ok() const { return status_ == SYMACCEL_OK; }
    int status() const { return status_; }

private:
    int prev_ = -1;
    int status_ = SYMACCEL_OK;
    bool switched_ = false;
};

#define SYM_TRY(expr)                                                   \
    do {                                                                \
        int _st = (expr);                                               \
        if (_st != SYMACCEL_OK) return _st;                             \
    } while (0)
#define SYM_GPU(ctx, expr)                                              \
    do {                                                                \
        hipError_t _e = (expr);                                         \
        if (_e != hipSuccess) return ::symaccel::ctx_fail((ctx), _e, #expr); \
    } while (0)

// line / 4 -> scale-factor band of the caller's swb offset tables (255 = past the last band), passed by value
struct AacBandMaps {
    uint8_t long4[256];
    uint8_t short4[32];
};

// kernel launchers (one per .hip file)
int launch_fft_big_wave(symaccel_ctx *ctx, int n, const float *d_in, float *d_out, size_t count, bool inverse);            // imdct_big.hip
int launch_fft4096_wg(symaccel_ctx *ctx, const float *d_in, float *d_out, size_t count, bool inverse);                         // imdct_big.hip
int launch_imdct8192_wg(symaccel_ctx *ctx, const cpx *d_twiddle, const float *d_spec, float *d_out, size_t count);              // imdct_big.hip
int launch_imdct_big_wave(symaccel_ctx *ctx, const cpx *d_twiddle, int nf, const float *d_spec, float *d_out, size_t count);  // imdct_big.hip
int launch_fft(symaccel_ctx *ctx, int n, const float *d_in, float *d_out, size_t count, bool inverse = false);
int launch_imdct(symaccel_ctx *ctx, const ImdctPlan &plan, const float *d_spec, float *d_out, size_t count);
// (maps / d_pair_chains / d_js_desc / n_pairs / d_js_scratch: joint stereo on load, symaccel_aac_synth_js_*; scratch of
// aac_js_scratch_bytes(n_chains, n_pairs, frames_per_chain) bytes)
int launch_aac(symaccel_ctx *ctx, const float *d_coeffs, const uint8_t *d_side, const float *d_delay_in,
               float *d_delay_out, float *d_pcm, size_t n_chains, size_t frames_per_chain, const AacBandMaps *maps = nullptr,
               const int32_t *d_pair_chains = nullptr, const symaccel_aac_js_frame *d_js_desc = nullptr, size_t n_pairs = 0,
               void *d_js_scratch = nullptr);
inline size_t aac_js_scratch_bytes(size_t n_chains, size_t n_pairs, size_t frames_per_chain) {
    (void)n_pairs;
    (void)frames_per_chain;
    return ((n_chains * 8 + 255) & ~(size_t)255) + 256;
}
int launch_mp3(symaccel_ctx *ctx, const float *d_xr, const symaccel_mp3_side *d_side, int sr,
               const float *d_overlap_in, const float *d_vvec_in, const int32_t *d_vfront_in,
               float *d_overlap_out, float *d_vvec_out, int32_t *d_vfront_out, float *d_pcm,
               size_t n_chains, size_t granules_per_chain);
int launch_vorbis(symaccel_ctx *ctx, int bs0_exp, int bs1_exp, const float *d_spectra, const float *d_residue,
                  size_t spec_stride,
                  const uint8_t *d_block_flag, const int32_t *d_prev_in, int32_t *d_prev_out,
                  const float *d_overlap_in, float *d_overlap_out, float *d_pcm, size_t pcm_stride,
                  size_t n_chains, size_t blocks_per_chain, void *d_offsets, const uint8_t *d_floor_y = nullptr);
int launch_vorbis_wave(symaccel_ctx *ctx, const cpx *tw_short, const cpx *tw_long, const float *win_short,
                       const float *win_long, const float *d_spectra, const float *d_residue, size_t spec_stride,
                       const uint8_t *d_block_flag,
                       const int32_t *d_prev_in, int32_t *d_prev_out, const float *d_overlap_in, float *d_overlap_out,
                       float *d_pcm, size_t pcm_stride, size_t n_chains, unsigned nb, unsigned seg, int floor_mode);
// SYM_VORBIS_WG (build knob): 1 = pairs with long blocks of 8192 samples run the workgroup-cooperative vorbis_synth_wg_kernel, 2 = those
// with 4096-sample long blocks too, 0 = vorbis_synth_wave2_kernel's one-wavefront-per-block form for all of them (kept for the A/B).
#ifndef SYM_VORBIS_WG
#define SYM_VORBIS_WG 1
#endif
#ifndef SYM_VORBIS_WG_SHARED
#define SYM_VORBIS_WG_SHARED 1  // vorbis_wg.hip: exchange / group work areas inside the staging area, three workgroups per CU
#endif
int launch_vorbis_wg(symaccel_ctx *ctx, int bs0_exp, int bs1_exp, const cpx *tw_short, const cpx *tw_long, const float *win_short,
                     const float *win_long, const float *d_spectra, const float *d_residue, size_t spec_stride, const uint8_t *d_block_flag,
                     const int32_t *d_prev_in, int32_t *d_prev_out, const float *d_overlap_in, float *d_overlap_out, float *d_pcm,
                     size_t pcm_stride, size_t n_chains, unsigned nb, unsigned seg, int floor_mode);  // vorbis_wg.hip
int launch_vorbis_wave2(symaccel_ctx *ctx, int bs0_exp, int bs1_exp, const cpx *tw_short, const cpx *tw_long, const float *win_short,
                        const float *win_long, const float *d_spectra, const float *d_residue, size_t spec_stride,
                        const uint8_t *d_block_flag, const int32_t *d_prev_in, int32_t *d_prev_out, const float *d_overlap_in,
                        float *d_overlap_out, float *d_pcm, size_t pcm_stride, size_t n_chains, unsigned nb, unsigned seg, int floor_mode);
int launch_vorbis_coupling(symaccel_ctx *ctx, float *d_mag, float *d_ang, size_t n);
int launch_vorbis_prepare(symaccel_ctx *ctx, float *d_residue, size_t spec_stride, unsigned channels_per_stream, size_t n_streams,
                          size_t blocks, const uint32_t *d_block_off, const uint8_t *d_steps, const uint32_t *d_step_first,
                          const uint8_t *d_kill);
int launch_vorbis_dot(symaccel_ctx *ctx, float *d_floor, const float *d_residue, size_t total);
int launch_vorbis_deinterleave(symaccel_ctx *ctx, const float *d_type2, float *d_planar, int n_ch, size_t n2,
                               size_t count);
int launch_vorbis_floor1(symaccel_ctx *ctx, const uint32_t *h_setup, int n_posts, int multiplier,
                         const uint32_t *d_y, uint32_t n, float *d_floor, size_t count, const float *d_residue = nullptr,
                         uint8_t *d_floor_y = nullptr, const uint32_t *d_line_offs = nullptr);
int launch_vorbis_floor1_pair(symaccel_ctx *ctx, const uint32_t *const h_setup[2], const int n_posts[2], const int multiplier[2],
                              const uint32_t *const d_y[2], const uint32_t n[2], const size_t count[2], const uint32_t *const d_line_offs[2],
                              uint8_t *d_floor_y);
int launch_aac_joint_stereo(symaccel_ctx *ctx, const AacBandMaps &maps, float *d_coeffs, size_t frames_per_chain,
                            const int32_t *d_pair_chains, const symaccel_aac_js_frame *d_desc, size_t n_pairs,
                            const uint32_t *d_list = nullptr, size_t n_list = 0);
int launch_aac_js_consume(symaccel_ctx *ctx, symaccel_aac_js_frame *d_desc, const uint32_t *d_list, size_t n_list, size_t n_pair_frames);
bool aac_band_maps(const uint16_t *swb_long, int n_swb_long, const uint16_t *swb_short, int n_swb_short, AacBandMaps *maps);
int launch_aac_tns(symaccel_ctx *ctx, float *d_coeffs, size_t n_frames, const symaccel_aac_tns_filter *d_filters,
                   size_t n_filters);
int launch_mp3_decode(symaccel_ctx *ctx, const int16_t *d_quant, const symaccel_mp3_requant *d_rq_desc, const int32_t *d_pair_chains,
                      const symaccel_mp3_stereo *d_st_desc, size_t n_pairs, const symaccel_mp3_side *d_side, int sr,
                      const float *d_overlap_in, const float *d_vvec_in, const int32_t *d_vfront_in, float *d_overlap_out,
                      float *d_vvec_out, int32_t *d_vfront_out, float *d_pcm, size_t n_chains, size_t granules_per_chain);
int launch_mp3_stereo(symaccel_ctx *ctx, float *d_xr, size_t granules_per_chain, const int32_t *d_pair_chains,
                      const symaccel_mp3_stereo *d_desc, int sr, size_t n_pairs, const int16_t *d_quant = nullptr,
                      const symaccel_mp3_requant *d_rq_desc = nullptr);
int launch_mp3_requantize(symaccel_ctx *ctx, const int16_t *d_quant, const symaccel_mp3_requant *d_desc, int sr,
                          float *d_xr, size_t n);
int launch_mpa_polyphase(symaccel_ctx *ctx, int n_frames, const float *d_in, const float *d_vvec_in,
                         const int32_t *d_vfront_in, float *d_vvec_out, int32_t *d_vfront_out, float *d_pcm, size_t n_chains,
                         size_t packets_per_chain);
int launch_alac_predict(symaccel_ctx *ctx, int32_t *d_buf, const symaccel_alac_desc *d_desc, const int32_t *d_coeffs,
                        size_t n_blocks, size_t blocksize, const int32_t *d_pair_weight = nullptr,
                        const uint8_t *d_pair_shift = nullptr, size_t stride = 0 /* words between rows; 0 = blocksize */);
int launch_alac_mid_side(symaccel_ctx *ctx, const int32_t *d_weight, const uint8_t *d_shift, int32_t *d_ch0, int32_t *d_ch1,
                         size_t n_pairs, size_t blocksize);
int launch_state_copy(symaccel_ctx *ctx, void *dst0, const void *src0, size_t bytes0, void *dst1, const void *src1,
                      size_t bytes1, void *dst2, const void *src2, size_t bytes2);
int launch_flac_status(symaccel_ctx *ctx, const symaccel_flac_desc *d_desc, size_t n, size_t blocksize, int8_t *d_status);
int launch_alac_status(symaccel_ctx *ctx, const symaccel_alac_desc *d_desc, size_t n, int8_t *d_status);
int launch_floor1_status(symaccel_ctx *ctx, const uint32_t *d_y, size_t count, int n_posts, int8_t *d_status);
int launch_tns_status(symaccel_ctx *ctx, const symaccel_aac_tns_filter *d_filters, size_t n, size_t n_frames, int8_t *d_status);
// symaccel_row_stride() pads rows that are a multiple of 2 KiB long (1024 samples and more) by this fraction of the row; the sweeps behind it:
// profiles/r06zz30_stride_sweep.txt, r06zz31_stride_sweep.txt
constexpr size_t kRowPadDiv = 8;
int launch_flac_restore(symaccel_ctx *ctx, int32_t *d_buf, const symaccel_flac_desc *d_desc,
                        const int32_t *d_coeffs, size_t n_blocks, size_t blocksize, const uint8_t *d_pair_mode = nullptr,
                        uint32_t out_shift = 0, size_t stride = 0 /* words between rows; 0 = blocksize */);
// batch_copy.hip: one piece (<= kBatchCopyPiece bytes) per workgroup, host (page-locked) <-> device in either direction
struct BatchCopyDesc {
    const void *src;
    void *dst;
    uint32_t bytes, pad;
};
constexpr size_t kBatchCopyPiece = 16384;
int launch_batch_copy(symaccel_ctx *ctx, hipStream_t stream, const BatchCopyDesc *descs, size_t n, bool scatter);
int launch_batch_flag(symaccel_ctx *ctx, hipStream_t stream, uint64_t *h_flag, uint64_t seq);  // (h_flag: page-locked host memory)
int launch_probe_copy(symaccel_ctx *ctx, const void *d_src, void *d_dst, size_t bytes, unsigned frames_per_wavefront, unsigned flags);
int launch_flac_decorrelate(symaccel_ctx *ctx, const uint8_t *d_mode, int32_t *d_ch0, int32_t *d_ch1,
                            size_t n_pairs, size_t blocksize, uint32_t out_shift);

}  // namespace symaccel
