/*
 * symaccel.h -- C ABI of the MI355X-native batched synthesis backend for Symphonia's DSP hot path.
 *
 * This is the drop-in boundary (SURVEY.md 8b / DESIGN.md "Boundary"): every entry point is what a
 * thin `unsafe extern "C"` block in a Rust shim crate binds in place of one reference call site.
 * Plain pointers and sizes only; no C++ types, no exceptions, no torch types.  The reference
 * items replaced are cited per function as <file>:<lines> relative to the Symphonia tree (0.6.1).
 *
 * Conventions
 *  - All `_device` entry points take DEVICE pointers (HBM resident) and only enqueue work on the
 *    context's stream; call symaccel_sync() (or synchronise the stream you handed in) before
 *    reading results.  Entry points without the suffix take HOST pointers, stage through HBM and
 *    return after the results are in the host buffers.
 *  - "chain" = one channel of one stream: the unit that carries overlap state from one codec
 *    frame to the next.  Batches are chain-major: x[chain][frame][...].  Chains are independent.
 *  - State buffers (`*_io`) are read at the first frame of the batch and hold the state after the
 *    last frame on return, so consecutive calls continue a stream exactly like consecutive
 *    decode() calls on the reference decoder.  reset() in the reference == zero the state.
 *    The `_pp_device` ("ping-pong") variants take the incoming and the outgoing state as two
 *    DISTINCT buffers (aliasing is rejected): a streaming caller keeps two state buffers and swaps
 *    them after every call.  They are a single kernel launch; the `_io` variants, whose segments of
 *    one chain run concurrently while one reads and another writes the state, stage the new state in
 *    context scratch and add one small copy kernel.
 *  - A context is bound to ONE stream at a time: symaccel_ctx_set_stream() drains the previous
 *    stream before switching.  Entry points leave the calling thread's current HIP device unchanged.
 *  - Return value: SYMACCEL_OK (0) or a negative symaccel_status.  INVALID_ARG corresponds to the
 *    reference's assert!/panic class, UNSUPPORTED to Error::Unsupported, DEVICE/OOM to
 *    Error::IoError (symphonia-core/src/errors.rs:38-54).  symaccel_strerror() returns static
 *    strings (usable as &'static str).  On error no output buffer is partially trusted: the
 *    caller clears its AudioBuffer as symphonia-core/src/codecs/audio.rs:278 requires.
 *  - What is validated: sizes, counts, null pointers, aliasing of ping-pong state, the ranges of
 *    scalar parameters, and -- where the describing data is in HOST memory (every host-pointer entry
 *    point; index lists such as mag_index / ang_index) -- the layout it implies against the strides
 *    given (symaccel_vorbis_synth checks the packed spectrum / PCM sizes that follow from the block
 *    flags).  Data that lives in DEVICE memory is trusted like the device pointers themselves: the
 *    block flags and pair_chains[] of the *_device entry points index the caller's own buffers, and
 *    a caller that builds them from an untrusted stream bounds them first (the reference's readers
 *    do: vorbis/lib.rs:461-470, aac/cpe.rs:53-74).  *_device Vorbis calls still reject a spec_stride
 *    below the all-short-blocks minimum.
 *  - Thread safety: distinct contexts may be used from distinct threads concurrently; one context
 *    is externally synchronised (matches `&mut self` on AudioDecoder, audio.rs:251-298).
 *  - There is NO CPU fallback: without a HIP device symaccel_ctx_create() fails with
 *    SYMACCEL_ERR_DEVICE.
 */
#ifndef SYMACCEL_H
#define SYMACCEL_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SYMACCEL_ABI_VERSION 8 /* 2: *_pp_device entry points, per-block status arrays, lookahead staging; 3: multi-GPU, probe; 4: mp3_decode_*device, vorbis floor_y; 5: aac_decode_pipelined, aac_joint_stereo_list, vorbis_decode; 6: batcher; 7: batch kinds for Vorbis from posts, FLAC and ALAC (symaccel_batch_slot has six input planes), lanes, per-ticket status; 8: *_strided_device (padded row pitch for the FLAC / ALAC planes), symaccel_row_stride */

typedef enum symaccel_status {
    SYMACCEL_OK = 0,
    SYMACCEL_ERR_INVALID_ARG = -1, /* reference: assert!/panic (e.g. mdct.rs:37-40, 76-78) */
    SYMACCEL_ERR_UNSUPPORTED = -2, /* reference: Error::Unsupported */
    SYMACCEL_ERR_DEVICE = -3,      /* HIP error / no device; reference class: Error::IoError */
    SYMACCEL_ERR_OOM = -4,         /* device or host allocation failed */
    SYMACCEL_ERR_DECODE = -5,      /* reference: Error::DecodeError -- malformed stream data: discard the packet, keep going */
} symaccel_status;

typedef struct symaccel_ctx symaccel_ctx;

int symaccel_abi_version(void);
const char *symaccel_strerror(int status);
/* Last HIP error text seen by this context ("" if none); static storage inside the context. */
const char *symaccel_last_error(const symaccel_ctx *ctx);

/* Create a context on HIP device `device` (ordinal).  Builds every constant table on the host
 * with the same libm calls the reference uses and uploads them. */
int symaccel_ctx_create(int device, symaccel_ctx **out);
void symaccel_ctx_destroy(symaccel_ctx *ctx);
/* Use an existing hipStream_t (e.g. PyTorch's current stream).  NULL = the context's own stream. */
int symaccel_ctx_set_stream(symaccel_ctx *ctx, void *hip_stream);
/* Block until everything enqueued on the context's stream has finished. */
int symaccel_sync(symaccel_ctx *ctx);
/* Tuning knob: frames (granules / blocks) one wavefront walks sequentially before the next
 * segment starts with a one-frame halo recompute.  0 = library default. */
int symaccel_ctx_set_segment(symaccel_ctx *ctx, int frames_per_segment);

/* ------------------------------------------------------------------ host memory and staged batches
 * Page-locked host memory for the batch buffers a shim hands to the host-pointer entry points: with it the
 * `_pipelined` entry points below overlap H2D, kernels and D2H (pageable memory works too, without the overlap). */
int symaccel_host_alloc(size_t bytes, void **out);
int symaccel_host_free(void *p);
int symaccel_host_register(void *p, size_t bytes);   /* pin memory the caller already owns */
int symaccel_host_unregister(void *p);

/* ------------------------------------------------------------------ core dsp (symphonia-core) */

/* Fft::fft / Fft::fft_inplace (symphonia-core/src/dsp/fft/no_simd.rs:96-140): `count` forward
 * complex FFTs of size n (power of two, 2 <= n <= 65536: the reference's own limit, no_simd.rs:77-80),
 * interleaved (re, im) f32.  d_in == d_out is allowed (fft_inplace).  More than 4096 points (used by none of the
 * reference's codecs) take a global-memory path through count * n * 8 bytes of context scratch. */
int symaccel_fft_c32_device(symaccel_ctx *ctx, int n, const float *d_in, float *d_out, size_t count);
int symaccel_fft_c32(symaccel_ctx *ctx, int n, const float *h_in, float *h_out, size_t count);
/* Ifft::ifft / Ifft::ifft_inplace (no_simd.rs:143-219): the forward transform() between a re <-> im swap of the permuted
 * input and a swap of the output scaled by 1.0 / n.  Same sizes and layout as the forward entry points.  As in the
 * reference, an inverse transform of fewer than 32 points only permutes, swaps and scales: transform() has no case for
 * them (Fft::fft dispatches fft2 .. fft16 before calling it, Ifft does not). */
int symaccel_ifft_c32_device(symaccel_ctx *ctx, int n, const float *d_in, float *d_out, size_t count);
int symaccel_ifft_c32(symaccel_ctx *ctx, int n, const float *h_in, float *h_out, size_t count);

/* Imdct::new_scaled(n, scale).imdct(spec, out) (symphonia-core/src/dsp/mdct.rs:35-146), `count`
 * times: spec[count][n] -> out[count][2n].  n = power of two, 4 <= n <= 131072 (mdct.rs:37-40); above 8192 the
 * global-memory path of the large Fft (count * n * 4 bytes of context scratch). */
int symaccel_imdct_f32_device(symaccel_ctx *ctx, int n, double scale, const float *d_spec,
                              float *d_out, size_t count);
int symaccel_imdct_f32(symaccel_ctx *ctx, int n, double scale, const float *h_spec, float *h_out,
                       size_t count);

/* --------------------------------------------------------------------------------- AAC-LC */

#define SYMACCEL_AAC_ONLY_LONG 0u   /* symphonia-codec-aac/src/aac/common.rs:17-20 */
#define SYMACCEL_AAC_LONG_START 1u
#define SYMACCEL_AAC_EIGHT_SHORT 2u
#define SYMACCEL_AAC_LONG_STOP 3u
/* one side byte per channel-frame: window_sequence | window_shape<<2 | prev_window_shape<<3 */
#define SYMACCEL_AAC_SIDE(seq, shape, prev_shape) \
    ((uint8_t)(((seq) & 3u) | (((shape) & 1u) << 2) | (((prev_shape) & 1u) << 3)))

/* Dsp::synth (symphonia-codec-aac/src/aac/dsp.rs:57-158) for n_chains x frames_per_chain
 * channel-frames: coeffs[chain][frame][1024] f32 (post pulse/TNS, i.e. what Ics::synth_channel
 * hands to Dsp::synth, ics/mod.rs:449-468), side[chain][frame], delay_io[chain][1024],
 * pcm[chain][frame][1024]. */
int symaccel_aac_synth_device(symaccel_ctx *ctx, const float *d_coeffs, const uint8_t *d_side,
                              float *d_delay_io, float *d_pcm, size_t n_chains,
                              size_t frames_per_chain);
int symaccel_aac_synth(symaccel_ctx *ctx, const float *h_coeffs, const uint8_t *h_side,
                       float *h_delay_io, float *h_pcm, size_t n_chains, size_t frames_per_chain);
/* symaccel_aac_synth with pinned / chunked / double-buffered staging: the batch is cut into chunks of `chunk_frames`
 * frames per chain (0 = library default, ~32 MiB of spectra per chunk); chunk k+1 is copied in and chunk k-1 copied out
 * while chunk k is transformed.  Same arguments and results as symaccel_aac_synth (which routes here for big batches). */
int symaccel_aac_synth_pipelined(symaccel_ctx *ctx, const float *h_coeffs, const uint8_t *h_side, float *h_delay_io,
                                 float *h_pcm, size_t n_chains, size_t frames_per_chain, size_t chunk_frames);
/* The same with the delay lines as separate in / out buffers (d_delay_in != d_delay_out): one launch. */
int symaccel_aac_synth_pp_device(symaccel_ctx *ctx, const float *d_coeffs, const uint8_t *d_side,
                                 const float *d_delay_in, float *d_delay_out, float *d_pcm,
                                 size_t n_chains, size_t frames_per_chain);

/* Spectral tools between the spectrum decoder and Dsp::synth, in place on coeffs[chain][frame][1024].
 *
 * Joint-stereo decoding of channel pairs (aac/cpe.rs:110-157).  pair_chains[p] = {left chain, right chain};
 * desc[p][frame]: per scale-factor band of every window the tool to apply -- index [sfb] for one long window,
 * [w * 16 + sfb] for eight short windows (the host expands window groups to windows, cpe.rs:116-119):
 * SYMACCEL_AAC_JS_MS = (m, s) -> (m + s, m - s); SYMACCEL_AAC_JS_INTENSITY = right = scale * left with
 * scale = dir * factor * ics1.scales[g][sfb] exactly as cpe.rs:127-131 forms it; 0 = neither (which is also how the
 * noise-substitution bands of cpe.rs:140-143 are expressed).  swb_long / swb_short: HOST arrays, the swb offsets of
 * the stream's sample rate (ICS get_bands(), ics/mod.rs:361-363), n_swb + 1 entries each, multiples of four.
 * Frames carrying pulse data are decoded on the host (Pulse::synth sits between the two stages, ics/mod.rs:452-454):
 * give them mode 0 everywhere. */
#define SYMACCEL_AAC_JS_MS 1u
#define SYMACCEL_AAC_JS_INTENSITY 2u
typedef struct symaccel_aac_js_frame {
    uint8_t num_windows; /* 1, or 8 for EIGHT_SHORT_SEQUENCE */
    uint8_t max_sfb;
    uint8_t pad[2];
    uint8_t mode[128];
    float scale[128];
} symaccel_aac_js_frame; /* 644 bytes */
int symaccel_aac_joint_stereo_device(symaccel_ctx *ctx, float *d_coeffs, size_t frames_per_chain,
                                     const int32_t *d_pair_chains, const symaccel_aac_js_frame *d_desc, size_t n_pairs,
                                     const uint16_t *swb_long, int n_swb_long, const uint16_t *swb_short,
                                     int n_swb_short);

/* symaccel_aac_joint_stereo_device for a LIST of channel-pair frames only: d_pair_frames[i] = pair * frames_per_chain + frame
 * (entries outside the batch are skipped; an entry listed twice is decoded twice).  What the frames that carry TNS filters need in
 * front of symaccel_aac_tns_device when everything else takes the fused walk below; traffic proportional to the list. */
int symaccel_aac_joint_stereo_list_device(symaccel_ctx *ctx, float *d_coeffs, size_t frames_per_chain,
                                          const int32_t *d_pair_chains, const symaccel_aac_js_frame *d_desc, size_t n_pairs,
                                          const uint16_t *swb_long, int n_swb_long, const uint16_t *swb_short, int n_swb_short,
                                          const uint32_t *d_pair_frames, size_t n_pair_frames);

/* symaccel_aac_synth_*_device with the joint-stereo decoding of the channel pairs done AS THE LINES ARE LOADED (cpe.rs:110-157 +
 * dsp.rs:57-158 in one kernel): d_coeffs holds what the spectrum decoder produced (mid / side or intensity-coded where the
 * descriptors say so), pair_chains / js_desc / swb tables as symaccel_aac_joint_stereo_device takes them; chains that belong to
 * no pair are synthesised as they are.  A chain of a pair reads its partner's lines beside its own, so the decoded spectra never
 * go to HBM and the separate read-modify-write pass disappears.  Frames that also carry TNS filters (ics/mod.rs:452-468: TNS
 * runs between joint stereo and the transform) must be decoded by symaccel_aac_joint_stereo_device + symaccel_aac_tns_device
 * first and get an all-zero mode row here.  `_pp_`: state in / state out, one synthesis launch. */
int symaccel_aac_synth_js_pp_device(symaccel_ctx *ctx, const float *d_coeffs, const uint8_t *d_side,
                                    const int32_t *d_pair_chains, const symaccel_aac_js_frame *d_js_desc, size_t n_pairs,
                                    const uint16_t *swb_long, int n_swb_long, const uint16_t *swb_short, int n_swb_short,
                                    const float *d_delay_in, float *d_delay_out, float *d_pcm, size_t n_chains,
                                    size_t frames_per_chain);
int symaccel_aac_synth_js_device(symaccel_ctx *ctx, const float *d_coeffs, const uint8_t *d_side,
                                 const int32_t *d_pair_chains, const symaccel_aac_js_frame *d_js_desc, size_t n_pairs,
                                 const uint16_t *swb_long, int n_swb_long, const uint16_t *swb_short, int n_swb_short,
                                 float *d_delay_io, float *d_pcm, size_t n_chains, size_t frames_per_chain);

/* The filtering loops of Tns::synth (aac/ics/tns.rs:180-195) for a flat list of filters.  The host keeps the
 * bitstream-side work (coefficient decoding with `sin`, tns.rs:39-106, and the band arithmetic of tns.rs:149-175) and
 * passes, per filter, the line range [start, end) it computed, the order, the direction and coef[0..order).
 * frame = index of the channel-frame in coeffs[n_frames][1024].  The filters of one frame cover disjoint ranges
 * (tns.rs:163-166), so the list order is free.  Entries with frame >= n_frames, start >= end, end > 1024 or an order
 * outside 1..20 are skipped. */
typedef struct symaccel_aac_tns_filter {
    uint32_t frame;
    uint16_t start, end;
    uint8_t order;     /* 1 ..= TNS_MAX_ORDER (20) */
    uint8_t direction; /* TnsCoeffs::direction: 0 = towards higher lines */
    uint8_t pad[2];
    float lpc[20];
} symaccel_aac_tns_filter; /* 92 bytes */
int symaccel_aac_tns_device(symaccel_ctx *ctx, float *d_coeffs, size_t n_frames,
                            const symaccel_aac_tns_filter *d_filters, size_t n_filters);

/* The AAC-LC tail from what the spectrum decoder produces, HOST memory in and out (cpe.rs:110-157 joint stereo, then per channel
 * ics/mod.rs:449-468: TNS and Dsp::synth): coeffs[chain][frame][1024] as decoded (mid / side or intensity coded where js_desc says
 * so), side[chain][frame], pair_chains[n_pairs][2] + js_desc[pair][frame] (n_pairs may be 0; a chain in at most one pair), the swb
 * tables as symaccel_aac_joint_stereo_device takes them, and the TNS filters of the whole batch as a flat list (frame = chain *
 * frames_per_chain + frame; n_tns may be 0).  Chunked like the other *_pipelined entry points.  Per chunk: the channel-pair frames
 * that carry a TNS filter in either channel get their joint stereo decoded in place first (a list pass over those frames only),
 * then the filters run (one lane per filter -- the recurrence is serial along the spectrum, so it stays a pass of its own), then
 * ONE walk decodes the joint stereo of every other frame as it loads the lines and synthesizes all of them.  Frames carrying pulse
 * data (Pulse::synth sits between joint stereo and TNS, ics/mod.rs:452-454) are the caller's: joint stereo + pulse on the host,
 * mode 0 everywhere here.  delay_io / pcm as symaccel_aac_synth_pipelined. */
int symaccel_aac_decode_pipelined(symaccel_ctx *ctx, const float *h_coeffs, const uint8_t *h_side, const int32_t *h_pair_chains,
                                  const symaccel_aac_js_frame *h_js_desc, size_t n_pairs, const uint16_t *swb_long, int n_swb_long,
                                  const uint16_t *swb_short, int n_swb_short, const symaccel_aac_tns_filter *h_tns, size_t n_tns,
                                  float *h_delay_io, float *h_pcm, size_t n_chains, size_t frames_per_chain, size_t chunk_frames);

/* Pulse::synth (aac/ics/pulse.rs:64-105, with iquant / requant :19-33), the tool between joint stereo and TNS
 * (Ics::synth_channel, ics/mod.rs:452-454).  HOST function on HOST memory: the tool raises values to the 4/3 and 3/4
 * power with f32 `powf`, which the reference takes from libm; this library calls the same libm so the values are the
 * reference's.  It is part of the CPU front end (it runs on the dequantised spectra before they are uploaded), not a
 * fallback for a kernel.  h_coeffs[n_frames][1024]; one record per channel-frame that carries pulse data (long windows
 * only, as in the reference); scales0 = Ics::scales[0] (one scale per scale-factor band); swb_long: n_swb_long + 1 offsets. */
typedef struct symaccel_aac_pulse {
    uint32_t frame;           /* index into h_coeffs */
    uint8_t number_pulse;     /* 1..4 */
    uint8_t pulse_start_sfb;
    uint8_t pulse_offset[4];
    uint8_t pulse_amp[4];
    uint8_t pad[2];
    float scales0[64];
} symaccel_aac_pulse; /* 272 bytes */
int symaccel_host_aac_pulse(float *h_coeffs, size_t n_frames, const symaccel_aac_pulse *h_pulse, size_t n_pulse,
                            const uint16_t *swb_long, int n_swb_long);

/* Per-filter status of symaccel_aac_tns_device's list (what the kernel skips): 0, or SYMACCEL_ERR_INVALID_ARG for an
 * entry with frame >= n_frames, start >= end, end > 1024 or an order outside 1..20.  d_status[n_filters] int8. */
int symaccel_aac_tns_status_device(symaccel_ctx *ctx, size_t n_frames, const symaccel_aac_tns_filter *d_filters,
                                   size_t n_filters, int8_t *d_status);

/* --------------------------------------------------------------------------------- MP3 */

#define SYMACCEL_MP3_LONG 0u /* BlockType, symphonia-bundle-mp3/src/layer3/common.rs:174-185 */
#define SYMACCEL_MP3_START 1u
#define SYMACCEL_MP3_SHORT 2u
#define SYMACCEL_MP3_END 3u

/* per granule-channel side record: the GranuleChannel fields the synthesis tail reads */
typedef struct symaccel_mp3_side {
    uint8_t block_type; /* SYMACCEL_MP3_* */
    uint8_t is_mixed;   /* BlockType::Short { is_mixed } */
    uint16_t rzero;     /* GranuleChannel::rzero as left by the parser (<= 576) */
} symaccel_mp3_side;

/* The per-channel tail of Layer3::decode's granule loop (layer3/mod.rs:440-476): reorder,
 * antialias, hybrid_synthesis, frequency_inversion (layer3/hybrid_synthesis.rs:153-485) and
 * synthesis::synthesis with n_frames = 18 (synthesis.rs:158-336).
 * xr[chain][granule][576] f32 = samples after requantize + stereo; side[chain][granule];
 * sample_rate_idx 0..8 selects the scale-factor-band table used by reorder.
 * State per chain: overlap_io[32][18], vvec_io[16][64], vfront_io (int32, 0..15).
 * pcm[chain][granule][576]. */
int symaccel_mp3_synth_device(symaccel_ctx *ctx, const float *d_xr, const symaccel_mp3_side *d_side,
                              int sample_rate_idx, float *d_overlap_io, float *d_vvec_io,
                              int32_t *d_vfront_io, float *d_pcm, size_t n_chains,
                              size_t granules_per_chain);
int symaccel_mp3_synth(symaccel_ctx *ctx, const float *h_xr, const symaccel_mp3_side *h_side,
                       int sample_rate_idx, float *h_overlap_io, float *h_vvec_io,
                       int32_t *h_vfront_io, float *h_pcm, size_t n_chains,
                       size_t granules_per_chain);
int symaccel_mp3_synth_pipelined(symaccel_ctx *ctx, const float *h_xr, const symaccel_mp3_side *h_side,
                                 int sample_rate_idx, float *h_overlap_io, float *h_vvec_io, int32_t *h_vfront_io,
                                 float *h_pcm, size_t n_chains, size_t granules_per_chain, size_t chunk_granules);
/* The same with SynthesisState / overlap as separate in / out buffers (pairwise distinct): one launch. */
int symaccel_mp3_synth_pp_device(symaccel_ctx *ctx, const float *d_xr, const symaccel_mp3_side *d_side,
                                 int sample_rate_idx, const float *d_overlap_in, const float *d_vvec_in,
                                 const int32_t *d_vfront_in, float *d_overlap_out, float *d_vvec_out,
                                 int32_t *d_vfront_out, float *d_pcm, size_t n_chains,
                                 size_t granules_per_chain);

/* Requantisation, the stage in front of stereo + the synthesis tail: the value mapping of
 * read_huffman_samples (layer3/requantize.rs:117-147, 172-205, 234: a decoded Huffman sample s becomes
 * (1 - 2 sign) * POW43[|s|], 0.0 for s == 0 and from rzero on; POW43: requantize.rs:28-31) followed by requantize
 * (requantize.rs:239-380: every scale-factor band times `f64::powf(2.0, 0.25 * (A - B)) as f32`, long / short /
 * mixed band tables of layer3/common.rs:9-172 -- including the reference's handling of a mixed block's last long
 * band, whose edge list stops one band short of the first short band, requantize.rs:368-372).
 * quant[n][576]: the signed quantised samples (|s| <= 8206; larger magnitudes are clamped to the table end);
 * desc[n]; xr[n][576] in the layout symaccel_mp3_synth_device consumes.  Both tables come from the host's libm.
 * Domain: subblock_gain < 8, scalefacs[i] + 3 <= 255 (every value the bitstream can code); exponents outside the
 * table (unreachable inside the domain) are clamped. */
#define SYMACCEL_MP3_RQ_SCALEFAC_SCALE 1u /* GranuleChannel::scalefac_scale */
#define SYMACCEL_MP3_RQ_PREFLAG 2u        /* GranuleChannel::preflag */
typedef struct symaccel_mp3_requant { /* the GranuleChannel fields requantize reads (layer3/mod.rs) */
    uint8_t global_gain;
    uint8_t flags;            /* SYMACCEL_MP3_RQ_* */
    uint8_t block_type;       /* SYMACCEL_MP3_* */
    uint8_t is_mixed;
    uint8_t subblock_gain[3];
    uint8_t reserved;
    uint16_t rzero;           /* first sample of the rzero partition (<= 576) */
    uint8_t scalefacs[39];
    uint8_t pad[3];
} symaccel_mp3_requant;       /* 52 bytes */
int symaccel_mp3_requantize_device(symaccel_ctx *ctx, const int16_t *d_quant, const symaccel_mp3_requant *d_desc,
                                   int sample_rate_idx, float *d_xr, size_t n_granule_channels);
int symaccel_mp3_requantize(symaccel_ctx *ctx, const int16_t *h_quant, const symaccel_mp3_requant *h_desc,
                            int sample_rate_idx, float *h_xr, size_t n_granule_channels);

/* Joint stereo, the stage between requantisation and the synthesis tail: stereo() (layer3/stereo.rs:485-556) with the
 * intensity-stereo band scans process_intensity_long_block (:196-260) and process_intensity_short_block (:264-483) --
 * which bands are intensity coded is decided from the requantised values of channel 1 (zero bands from the top,
 * per window for short blocks), exactly as the reference does -- mid/side everywhere below the intensity bound, and
 * the ratio tables of stereo.rs:31-118 (host libm: tan / powf).  In place on xr[chain][granule][576];
 * pair_chains[p] = {channel-0 chain, channel-1 chain}; desc[p][granule].  The caller keeps the reference's check
 * that both channels carry the same block type (stereo.rs:502-504) and afterwards gives both channels
 * rzero = max(rzero0, rzero1) (stereo.rs:549-553) in the side records of symaccel_mp3_synth. */
#define SYMACCEL_MP3_ST_MID_SIDE 1u  /* Mode::Layer3 { mid_side, .. } */
#define SYMACCEL_MP3_ST_INTENSITY 2u /* Mode::Layer3 { .., intensity } */
#define SYMACCEL_MP3_ST_MPEG1 4u     /* FrameHeader::is_mpeg1() */
#define SYMACCEL_MP3_ST_IS_SCALE 8u  /* channels[1].scalefac_compress & 1 (selects the MPEG-2 / 2.5 ratio table) */
typedef struct symaccel_mp3_stereo {
    uint8_t flags;          /* SYMACCEL_MP3_ST_* */
    uint8_t block_type;     /* SYMACCEL_MP3_*, of both channels */
    uint8_t is_mixed;
    uint8_t reserved;
    uint16_t rzero0, rzero1;
    uint8_t scalefacs1[39]; /* channels[1].scalefacs: the intensity positions */
    uint8_t pad;
} symaccel_mp3_stereo;      /* 48 bytes */
int symaccel_mp3_stereo_device(symaccel_ctx *ctx, float *d_xr, size_t granules_per_chain, const int32_t *d_pair_chains,
                               const symaccel_mp3_stereo *d_desc, int sample_rate_idx, size_t n_pairs);

/* symaccel_mp3_requantize_device followed by symaccel_mp3_stereo_device for the channel pairs of a batch in one pass:
 * the requantised spectra stay in registers between the two stages.  quant / rq_desc / xr are indexed
 * [chain][granule] like d_xr above; only the chains named in pair_chains are read and written (mono chains go through
 * symaccel_mp3_requantize_device).  Granules whose desc has neither joint-stereo flag are requantised only. */
int symaccel_mp3_requantize_stereo_device(symaccel_ctx *ctx, const int16_t *d_quant, const symaccel_mp3_requant *d_rq_desc,
                                          size_t granules_per_chain, const int32_t *d_pair_chains,
                                          const symaccel_mp3_stereo *d_desc, int sample_rate_idx, float *d_xr,
                                          size_t n_pairs);

/* The whole Layer III tail from what the entropy decoder produces, HOST memory in and out (layer3/mod.rs:421-477: requantize,
 * stereo, then the per-channel synthesis): quant[chain][granule][576] int16 Huffman samples and rq_desc[chain][granule];
 * pair_chains[n_pairs][2] + st_desc[pair][granule] for the joint-stereo channel pairs (n_pairs may be 0; a chain appears in at
 * most one pair); side[chain][granule] as symaccel_mp3_synth takes it (with the rzero the reference has AFTER stereo,
 * stereo.rs:549-553); state and pcm as symaccel_mp3_synth_pipelined.  Chunked like the other *_pipelined entry points: what
 * crosses PCIe on the way in is 2 bytes per line plus the records -- half the bytes of the f32 spectra -- and the requantised
 * spectra never leave the device. */
int symaccel_mp3_decode_pipelined(symaccel_ctx *ctx, const int16_t *h_quant, const symaccel_mp3_requant *h_rq_desc,
                                  const int32_t *h_pair_chains, const symaccel_mp3_stereo *h_st_desc, size_t n_pairs,
                                  const symaccel_mp3_side *h_side, int sample_rate_idx, float *h_overlap_io, float *h_vvec_io,
                                  int32_t *h_vfront_io, float *h_pcm, size_t n_chains, size_t granules_per_chain,
                                  size_t chunk_granules);

/* The same tail in ONE kernel on device buffers (layer3/mod.rs:421-477 fused: requantize + stereo happen in the synthesis
 * kernel's load path -- csrc/mp3.hip `mp3_front` -- so the requantised spectra never exist in HBM: 2 bytes per line + the
 * records in, 4 bytes per sample out).  unit_chains[n_units][2]: the chains of every stream of the batch, {channel 0,
 * channel 1} or {chain, -1} for a mono stream; every chain appears exactly once.  st_desc[n_units][granule] (read, but
 * ignored, for mono units); quant / rq_desc / side / pcm / state indexed by chain as above.  `_pp_`: separate state in /
 * out buffers (pairwise distinct), one launch; the plain form updates the state in place. */
int symaccel_mp3_decode_pp_device(symaccel_ctx *ctx, const int16_t *d_quant, const symaccel_mp3_requant *d_rq_desc,
                                  const int32_t *d_unit_chains, const symaccel_mp3_stereo *d_st_desc, size_t n_units,
                                  const symaccel_mp3_side *d_side, int sample_rate_idx, const float *d_overlap_in,
                                  const float *d_vvec_in, const int32_t *d_vfront_in, float *d_overlap_out, float *d_vvec_out,
                                  int32_t *d_vfront_out, float *d_pcm, size_t n_chains, size_t granules_per_chain);
int symaccel_mp3_decode_device(symaccel_ctx *ctx, const int16_t *d_quant, const symaccel_mp3_requant *d_rq_desc,
                               const int32_t *d_unit_chains, const symaccel_mp3_stereo *d_st_desc, size_t n_units,
                               const symaccel_mp3_side *d_side, int sample_rate_idx, float *d_overlap_io, float *d_vvec_io,
                               int32_t *d_vfront_io, float *d_pcm, size_t n_chains, size_t granules_per_chain);

/* synthesis::synthesis alone (synthesis.rs:158-336) as Layer I and Layer II use it: n_frames = 12
 * (layer1/mod.rs:193) or 36 (layer2/mod.rs:383) time slots per packet and channel.  in[chain][packet][32 * n_frames]
 * sub-band-major (in[n_frames * i + b], synthesis.rs:168-170); pcm[chain][packet][32 * n_frames]; state per chain:
 * vvec_io[16][64], vfront_io (the reference's SynthesisState).  Other n_frames: SYMACCEL_ERR_UNSUPPORTED. */
int symaccel_mpa_polyphase_device(symaccel_ctx *ctx, int n_frames, const float *d_in, float *d_vvec_io,
                                  int32_t *d_vfront_io, float *d_pcm, size_t n_chains, size_t packets_per_chain);
int symaccel_mpa_polyphase(symaccel_ctx *ctx, int n_frames, const float *h_in, float *h_vvec_io,
                           int32_t *h_vfront_io, float *h_pcm, size_t n_chains, size_t packets_per_chain);
int symaccel_mpa_polyphase_pp_device(symaccel_ctx *ctx, int n_frames, const float *d_in, const float *d_vvec_in,
                                     const int32_t *d_vfront_in, float *d_vvec_out, int32_t *d_vfront_out,
                                     float *d_pcm, size_t n_chains, size_t packets_per_chain);

/* --------------------------------------------------------------------------------- Vorbis */

/* DspChannel::synth for every channel of every block (symphonia-codec-vorbis/src/dsp.rs:68-126,
 * called from lib.rs:296-331).  A chain's spectra (floor x residue, lib.rs:282-292) are packed
 * back to back: block b with flag f contributes bs_f/2 floats; a chain starts at
 * d_spectra + chain * spec_stride.  block_flag[chain][block] in {0,1}.  prev_flag_io[chain]:
 * -1 = no previous block (Dsp::prev_block_flag == None), else 0/1.  overlap_io[chain][bs1/2].
 * PCM is packed the same way: block b contributes (prev_n + n)/4 floats (lib.rs:303); a chain
 * starts at d_pcm + chain * pcm_stride.  6 <= bs0_exp <= bs1_exp <= 13. */
int symaccel_vorbis_synth_device(symaccel_ctx *ctx, int bs0_exp, int bs1_exp, const float *d_spectra,
                                 size_t spec_stride, const uint8_t *d_block_flag,
                                 int32_t *d_prev_flag_io, float *d_overlap_io, float *d_pcm,
                                 size_t pcm_stride, size_t n_chains, size_t blocks_per_chain);
/* The same with the dot product fused (lib.rs:282-292 + dsp.rs:68-126): the spectrum of every channel-block is
 * d_floor[i] * d_residue[i], multiplied as the lines are loaded -- both arrays packed like d_spectra.  Saves the
 * separate pass of symaccel_vorbis_dot_product_device over HBM. */
int symaccel_vorbis_synth_fr_device(symaccel_ctx *ctx, int bs0_exp, int bs1_exp, const float *d_floor,
                                    const float *d_residue, size_t spec_stride, const uint8_t *d_block_flag,
                                    int32_t *d_prev_flag_io, float *d_overlap_io, float *d_pcm, size_t pcm_stride,
                                    size_t n_chains, size_t blocks_per_chain);
int symaccel_vorbis_synth(symaccel_ctx *ctx, int bs0_exp, int bs1_exp, const float *h_spectra,
                          size_t spec_stride, const uint8_t *h_block_flag, int32_t *h_prev_flag_io,
                          float *h_overlap_io, float *h_pcm, size_t pcm_stride, size_t n_chains,
                          size_t blocks_per_chain);
/* Either of the two above with the state as separate in / out buffers (pairwise distinct): d_residue == NULL means
 * d_spectra holds floor x residue already, otherwise d_spectra is the floor and the dot product is fused. */
int symaccel_vorbis_synth_pp_device(symaccel_ctx *ctx, int bs0_exp, int bs1_exp, const float *d_spectra,
                                    const float *d_residue, size_t spec_stride, const uint8_t *d_block_flag,
                                    const int32_t *d_prev_flag_in, int32_t *d_prev_flag_out,
                                    const float *d_overlap_in, float *d_overlap_out, float *d_pcm,
                                    size_t pcm_stride, size_t n_chains, size_t blocks_per_chain);

/* The same with the floor curve as dB-table INDICES, one byte per line (symaccel_vorbis_floor1_y_device; floor.rs:785-825 writes
 * FLOOR1_INVERSE_DB_TABLE[y] and nothing else): floor_y[chain][spec_stride] bytes in the spectrum's packed layout, residue as
 * d_spectra would be.  The kernels look the table up as they load the residue -- spectrum[i] = table[y[i]] * residue[i], the
 * value floor.rs:822 stores times lib.rs:289-291's `*f *= r`, one rounded multiply -- so a floor-1 stream costs 1 byte per line
 * written + 5 bytes per line read in front of the PCM instead of a curve (or a multiplied spectrum) in f32.  A channel marked
 * do-not-decode (lib.rs:284-287: floor all zero) is a zero residue with any y.  spec_stride and pcm_stride multiples of 4,
 * d_floor_y and d_residue 16-byte aligned (other layouts: SYMACCEL_ERR_INVALID_ARG / SYMACCEL_ERR_UNSUPPORTED). */
int symaccel_vorbis_synth_fy_pp_device(symaccel_ctx *ctx, int bs0_exp, int bs1_exp, const uint8_t *d_floor_y,
                                       const float *d_residue, size_t spec_stride, const uint8_t *d_block_flag,
                                       const int32_t *d_prev_flag_in, int32_t *d_prev_flag_out, const float *d_overlap_in,
                                       float *d_overlap_out, float *d_pcm, size_t pcm_stride, size_t n_chains,
                                       size_t blocks_per_chain);
int symaccel_vorbis_synth_fy_device(symaccel_ctx *ctx, int bs0_exp, int bs1_exp, const uint8_t *d_floor_y,
                                    const float *d_residue, size_t spec_stride, const uint8_t *d_block_flag,
                                    int32_t *d_prev_flag_io, float *d_overlap_io, float *d_pcm, size_t pcm_stride,
                                    size_t n_chains, size_t blocks_per_chain);

/* The Vorbis tail from what the packet decoder produces, HOST memory in and out (lib.rs:250-331: inverse coupling, floor-1 curve
 * synthesis, dot product, inverse MDCT + windowing + overlap-add).  The channels of a stream are `channels_per_stream` consecutive
 * chains with equal block flags (one mode per packet, lib.rs:170-178).
 *   residue[chain][spec_stride]   the decoded residue vectors, packed like the spectra of symaccel_vorbis_synth; all zeros for a
 *                                 channel the reference marks do-not-decode (lib.rs:196-228)
 *   floor[chain][block]           index into `floors`, or SYMACCEL_VORBIS_FLOOR_UNUSED: the channel's floor is unused in this packet
 *                                 (floor.rs:660-668), its curve is all zeros (lib.rs:206-209) -- the spectrum is 0.0 * residue
 *   posts[chain][block][posts_stride]   the floor1_Y values read from the packet (floor.rs:680-740), each <= 511
 *                                 (SYMACCEL_ERR_UNSUPPORTED otherwise: symaccel_vorbis_floor1_status_device's domain)
 *   floors[n_floors]              the floor-1 configurations of the setup header the batch uses (x list, multiplier)
 *   coupling[][2], coupling_first[n_streams * blocks_per_chain + 1]   the (magnitude, angle) channel pairs of block b of stream s are
 *                                 coupling[coupling_first[s * blocks + b] .. coupling_first[s * blocks + b + 1]), applied in order
 *                                 (the mapping of the packet's mode, lib.rs:252-278; channel indices inside the stream)
 * prev_flag_io / overlap_io / pcm as symaccel_vorbis_synth.  On the device: one pass applies the coupling steps in place (and the
 * zero floors), the floor curves are rendered as one byte per line (symaccel_vorbis_floor1_y_device, one launch per floor
 * configuration and block size), and the synthesis kernel multiplies table[y] * residue as it loads the lines
 * (symaccel_vorbis_synth_fy_device).  Floor-0 streams take symaccel_host_vorbis_floor0 + symaccel_vorbis_synth instead. */
#define SYMACCEL_VORBIS_FLOOR_UNUSED 255u
typedef struct symaccel_vorbis_floor1_cfg {
    uint8_t multiplier; /* floor1_multiplier, 1..4 */
    uint8_t n_posts;    /* 2..65 */
    uint8_t pad[2];
    uint32_t x_list[65]; /* floor1_X_list, in bitstream order */
} symaccel_vorbis_floor1_cfg; /* 264 bytes */
int symaccel_vorbis_decode(symaccel_ctx *ctx, int bs0_exp, int bs1_exp, const float *h_residue, size_t spec_stride,
                           const uint8_t *h_block_flag, const uint8_t *h_floor, const uint32_t *h_posts, size_t posts_stride,
                           const symaccel_vorbis_floor1_cfg *h_floors, size_t n_floors, size_t channels_per_stream,
                           const uint8_t *h_coupling, const uint32_t *h_coupling_first, int32_t *h_prev_flag_io, float *h_overlap_io,
                           float *h_pcm, size_t pcm_stride, size_t n_chains, size_t blocks_per_chain);

/* Inverse coupling (lib.rs:252-278) of `n_pairs` (magnitude, angle) vector pairs of n floats,
 * in place: pair p uses d_residue + mag_index[p]*n and d_residue + ang_index[p]*n.  Pairs are
 * applied in order (coupling steps may chain).  Index arrays are HOST arrays (<= 256 entries). */
int symaccel_vorbis_inverse_coupling_device(symaccel_ctx *ctx, float *d_residue, size_t n,
                                            const uint32_t *mag_index, const uint32_t *ang_index,
                                            size_t n_pairs);
/* Dot product floor[i] *= residue[i] over `total` floats (lib.rs:282-292). */
int symaccel_vorbis_dot_product_device(symaccel_ctx *ctx, float *d_floor, const float *d_residue,
                                       size_t total);
/* Residue type-2 de-interleave (residue.rs:177-218): type2[count][n2*n_ch] -> planar[count][n_ch][n2]. */
int symaccel_vorbis_deinterleave2_device(symaccel_ctx *ctx, const float *d_type2, float *d_planar,
                                         int n_ch, size_t n2, size_t count);
/* Floor-1 curve synthesis, steps 1 and 2 (floor.rs:568-653, 776-825) for `count` channel-blocks
 * sharing one floor configuration: x_list[n_posts] (HOST, from the setup header), multiplier
 * 1..4, y[count][n_posts] (DEVICE, decoded floor1_Y values, each <= 511: see symaccel_vorbis_floor1_status_device; a block
 * with a larger value gets an unspecified curve), n = blocksize / 2 (a multiple of 16, <= 4096), floor[count][n] out. */
int symaccel_vorbis_floor1_device(symaccel_ctx *ctx, const uint32_t *x_list, int n_posts,
                                  int multiplier, const uint32_t *d_y, uint32_t n, float *d_floor,
                                  size_t count);
/* The same with the dot product of lib.rs:282-292 fused into the curve's store: spectrum[i] = floor[i] * residue[i], one
 * rounded multiply per line (the reference's `*f *= r`), for channels whose residue is decoded.  The curve never goes to
 * HBM and the separate dot-product pass disappears: feed the result to symaccel_vorbis_synth_*.  d_spectrum may be
 * d_residue (in place). */
int symaccel_vorbis_floor1_dot_device(symaccel_ctx *ctx, const uint32_t *x_list, int n_posts, int multiplier,
                                      const uint32_t *d_y, uint32_t n, const float *d_residue, float *d_spectrum,
                                      size_t count);
/* symaccel_vorbis_floor1_dot_device for the blocks of one size class of a MIXED stream: block b's lines are at line
 * d_line_offsets[b] (multiples of 4) of d_residue / d_spectrum -- the packed layout of symaccel_vorbis_synth_*. */
int symaccel_vorbis_floor1_dot_at_device(symaccel_ctx *ctx, const uint32_t *x_list, int n_posts, int multiplier,
                                         const uint32_t *d_y, uint32_t n, const uint32_t *d_line_offsets,
                                         const float *d_residue, float *d_spectrum, size_t count);
/* The curve as dB-table INDICES, one byte per line (floor.rs:785-825: every value render_line writes is
 * FLOOR1_INVERSE_DB_TABLE[y], y = 0..255): floor_y[.. + i] = the y of line i, block b at byte offset d_line_offsets[b] (multiples
 * of 4; NULL: b * n).  With the offsets of a stream's blocks in the packed spectrum layout of symaccel_vorbis_synth_* this
 * writes the plane symaccel_vorbis_synth_fy_* reads -- one call per block-size class of a mixed stream.  1 byte per line to
 * HBM instead of a 4-byte curve (or a read-multiply-write pass over the residue); d_floor_y 4-byte aligned. */
int symaccel_vorbis_floor1_y_device(symaccel_ctx *ctx, const uint32_t *x_list, int n_posts, int multiplier,
                                    const uint32_t *d_y, uint32_t n, const uint32_t *d_line_offsets, uint8_t *d_floor_y,
                                    size_t count);
/* Several such renders into ONE plane -- the block-size classes of a floor, the floors of a stream (vorbis/lib.rs:206-232 picks a floor
 * per channel and block) -- as jobs of one call: two jobs share a launch (their workgroups fill each other's last round).  The jobs' lines
 * must not overlap; each job's fields are symaccel_vorbis_floor1_y_device's arguments; jobs with count 0 are skipped; every job is
 * checked before anything is launched. */
typedef struct symaccel_vorbis_floor1_job {
    const uint32_t *x_list;
    int n_posts, multiplier;
    const uint32_t *d_y;
    uint32_t n;
    const uint32_t *d_line_offsets;
    size_t count;
} symaccel_vorbis_floor1_job;
int symaccel_vorbis_floor1_y_jobs_device(symaccel_ctx *ctx, const symaccel_vorbis_floor1_job *jobs, size_t n_jobs, uint8_t *d_floor_y);
/* Per-block status of the y rows (d_status[count] int8): 0, or SYMACCEL_ERR_UNSUPPORTED for a block with a value above
 * 511.  floor1_Y values are codebook entry numbers (floor.rs:698-712) that a conforming stream keeps below the floor's
 * range (<= 256); the reference computes whatever a larger value implies in i32.  Up to 511 the kernels above reproduce
 * it exactly (render_point in integers wherever a difference of final_y values leaves the closed form's proven range;
 * the final_y stay inside 16 bits); beyond, a block is the caller's to render on the CPU. */
int symaccel_vorbis_floor1_status_device(symaccel_ctx *ctx, int n_posts, const uint32_t *d_y, size_t count,
                                         int8_t *d_status);

/* Floor 0 (floor.rs:124-397) -- HOST functions on HOST memory, for the same reason as symaccel_host_aac_pulse: f64
 * atan / floor and f32 cos / sqrt / exp from libm, in the reference's operation order.  Floor-0 streams are rare (the
 * reference encoder never produces them); the curve is computed on the host and uploaded like a floor-1 curve.
 *  - bark_map (floor.rs:358-376): the map of one block size, n = blocksize / 2 entries.
 *  - floor0_coeffs: the `coeff = 2 cos(coeff)` step that ends Floor0::read_channel (floor.rs:246-248), in place.
 *  - floor0: Floor0::synthesis + linear_floor0_value (floor.rs:262-340, 379-390) for one channel-block; returns
 *    SYMACCEL_ERR_DECODE where the reference returns decode_error("vorbis: invalid floor0 coefficients"). */
int symaccel_host_vorbis_bark_map(uint32_t n, uint16_t floor0_rate, uint16_t floor0_bark_map_size, int32_t *h_map);
int symaccel_host_vorbis_floor0_coeffs(float *h_coeffs, int order);
int symaccel_host_vorbis_floor0(const float *h_coeffs, int order, const int32_t *h_map, uint32_t n,
                                uint16_t floor0_bark_map_size, uint8_t amplitude_bits, uint8_t amplitude_offset,
                                uint64_t amplitude, float *h_floor);

/* --------------------------------------------------------------------------------- FLAC */

#define SYMACCEL_FLAC_VERBATIM 0u /* constant / verbatim subframe: predictor is a no-op */
#define SYMACCEL_FLAC_FIXED 1u    /* fixed_predict, decoder.rs:663-710 */
#define SYMACCEL_FLAC_LPC 2u      /* lpc_predict, decoder.rs:716-752 */

typedef struct symaccel_flac_desc {
    uint8_t kind;        /* SYMACCEL_FLAC_* */
    uint8_t order;       /* fixed: 0..4, lpc: 1..32 (<= blocksize) */
    uint8_t shift;       /* qlp_coeff_shift, 0..15 (negative shifts: reference returns Unsupported) */
    uint8_t wasted_bits; /* dropped_bps for samples_shl, decoder.rs:396-409 */
} symaccel_flac_desc;

/* Predictor restore for n_blocks subframes of `blocksize` samples, in place:
 * buf[block][blocksize] i32 holds `order` warm-up samples followed by residuals
 * (decoder.rs:446-511); coeffs[block][32] in bitstream order (first coefficient multiplies the
 * most recent sample; unused entries ignored).  Bit-exact i64 accumulation, wrapping i32 add. */
int symaccel_flac_restore_device(symaccel_ctx *ctx, int32_t *d_buf, const symaccel_flac_desc *d_desc,
                                 const int32_t *d_coeffs, size_t n_blocks, size_t blocksize);
int symaccel_flac_restore(symaccel_ctx *ctx, int32_t *h_buf, const symaccel_flac_desc *h_desc,
                          const int32_t *h_coeffs, size_t n_blocks, size_t blocksize);
int symaccel_flac_restore_pipelined(symaccel_ctx *ctx, int32_t *h_buf, const symaccel_flac_desc *h_desc,
                                    const int32_t *h_coeffs, size_t n_blocks, size_t blocksize, size_t chunk_blocks);
/* Per-block status of a descriptor array, as the reference would have judged each subframe (d_status[n_blocks] int8):
 * 0 = decodable; SYMACCEL_ERR_DECODE = predictor order greater than the block size (decoder.rs:431-433, 456-458), an
 * unknown kind, a fixed order above 4 or an LPC order outside 1..32; SYMACCEL_ERR_UNSUPPORTED = shift > 31, the
 * encoding of a negative qlp shift (decoder.rs:506-508 returns Unsupported).  The restore kernels clamp such blocks
 * instead of faulting; a shim uses this array to fail exactly the packets the reference fails. */
int symaccel_flac_block_status_device(symaccel_ctx *ctx, const symaccel_flac_desc *d_desc, size_t n_blocks,
                                      size_t blocksize, int8_t *d_status);
/* Predictor restore with the stereo decorrelation and the final shift fused into the write-back: blocks 2p and 2p+1
 * are channel 0 and channel 1 of pair p (n_blocks even), pair_mode[p] as for symaccel_flac_decorrelate_device below,
 * out_shift = 32 - bits_per_sample.  One pass over HBM instead of two (decoder.rs:199-242 in one kernel). */
int symaccel_flac_restore_stereo_device(symaccel_ctx *ctx, int32_t *d_buf, const symaccel_flac_desc *d_desc,
                                        const int32_t *d_coeffs, const uint8_t *d_pair_mode, uint32_t out_shift,
                                        size_t n_blocks, size_t blocksize);
/* The same restore over rows `stride` words apart (ABI 8): buf[block][stride] i32, the first `blocksize` words of a row are the
 * subframe, the rest is padding the kernels neither read nor write.  stride >= blocksize (0 = blocksize: rows back to back);
 * a multiple of 4 keeps the 16-byte tile path.  d_pair_mode NULL = symaccel_flac_restore_device, else the fused stereo form
 * (n_blocks even).  Why it exists: one LANE owns a subframe, so a wavefront moves 64 row segments of 128 B per tile, one per
 * row; with rows 4, 8, 16 or 32 KiB apart (4096 samples = 16 KiB, the block size of nearly every FLAC stream) those 64
 * segments fall on a fraction of the HBM channels and the kernel is bound by that instead of its arithmetic (FLAC row).
 * A caller that owns the layout -- the batcher's device planes, a shim that decodes residuals straight into a batch buffer --
 * asks symaccel_row_stride() for the pitch.  The reference has no such notion: its buffers are one Vec<i32> per channel
 * (decoder.rs:199-242), i.e. every row already lives at an unrelated address. */
int symaccel_flac_restore_strided_device(symaccel_ctx *ctx, int32_t *d_buf, const symaccel_flac_desc *d_desc,
                                         const int32_t *d_coeffs, const uint8_t *d_pair_mode, uint32_t out_shift,
                                         size_t n_blocks, size_t blocksize, size_t stride);
/* The row pitch (in i32 words, a multiple of 4, >= blocksize) the lane-per-block kernels (FLAC restore, ALAC predict) run
 * fastest at for blocks of `blocksize` samples: blocksize rounded up to 4, plus an eighth of the row when it is 1024 samples or more and
 * a multiple of 512 (4096 -> 4608: config 5's batch goes from 0.57 to 0.66 of the HBM peak).  Pure arithmetic, no context. */
size_t symaccel_row_stride(size_t blocksize);
/* decorrelate_{left_side,mid_side,right_side} (decoder.rs:32-82) then `<< out_shift`
 * (decoder.rs:239-242, out_shift = 32 - bits_per_sample, 0 = none) over n_pairs channel pairs:
 * mode[pair] in {0 independent, 1 left/side, 2 mid/side, 3 right/side}; ch0/ch1[pair][blocksize]. */
int symaccel_flac_decorrelate_device(symaccel_ctx *ctx, const uint8_t *d_mode, int32_t *d_ch0,
                                     int32_t *d_ch1, size_t n_pairs, size_t blocksize,
                                     uint32_t out_shift);
int symaccel_flac_decorrelate(symaccel_ctx *ctx, const uint8_t *h_mode, int32_t *h_ch0, int32_t *h_ch1,
                              size_t n_pairs, size_t blocksize, uint32_t out_shift);

/* --------------------------------------------------------------------------------- ALAC */

/* per element-channel descriptor: the ElementChannel fields predict() reads (symphonia-codec-alac/src/lib.rs:71-80) */
typedef struct symaccel_alac_desc {
    uint8_t mode;      /* 0, or 15 (double predictor); 1..14 are invalid (lib.rs:167-169): the block is left as is */
    uint8_t lpc_order; /* 0..31; 0 = no prediction (lib.rs:173-175) */
    uint8_t shift;     /* coefficient quantisation shift, 0..15 */
    uint8_t bps;       /* prediction bit width: outputs are sign-extended from `bps` bits (clip_msbs, lib.rs:659) */
} symaccel_alac_desc;

/* ElementChannel::predict (lib.rs:165-264) for n_blocks element channels of `blocksize` samples, in place:
 * buf[block][blocksize] i32 holds the Rice-decoded residuals; coeffs[block][32] as read from the bitstream
 * (lib.rs:94-98; adapted on a private copy, like the reference's per-packet ElementChannel).
 * Bit-exact wrapping i32 arithmetic. */
int symaccel_alac_predict_device(symaccel_ctx *ctx, int32_t *d_buf, const symaccel_alac_desc *d_desc,
                                 const int32_t *d_coeffs, size_t n_blocks, size_t blocksize);
int symaccel_alac_predict(symaccel_ctx *ctx, int32_t *h_buf, const symaccel_alac_desc *h_desc,
                          const int32_t *h_coeffs, size_t n_blocks, size_t blocksize);
/* Per-block status (d_status[n_blocks] int8): 0, or SYMACCEL_ERR_DECODE for modes 1..14, which the reference rejects
 * with decode_error("alac: invalid mode") (lib.rs:167-169) and the predict kernels leave untouched. */
int symaccel_alac_block_status_device(symaccel_ctx *ctx, const symaccel_alac_desc *d_desc, size_t n_blocks,
                                      int8_t *d_status);
/* predict with decorrelate_mid_side fused into the write-back (decode_element, lib.rs:541-560, in one pass over HBM):
 * blocks 2p and 2p+1 are the two channels of pair p (n_blocks even); pair_weight[p] (0 = no mixing), pair_shift[p]. */
int symaccel_alac_predict_stereo_device(symaccel_ctx *ctx, int32_t *d_buf, const symaccel_alac_desc *d_desc,
                                        const int32_t *d_coeffs, const int32_t *d_pair_weight,
                                        const uint8_t *d_pair_shift, size_t n_blocks, size_t blocksize);
/* predict over rows `stride` words apart (ABI 8; see symaccel_flac_restore_strided_device): d_pair_weight / d_pair_shift NULL =
 * symaccel_alac_predict_device, else the fused mid/side form (n_blocks even). */
int symaccel_alac_predict_strided_device(symaccel_ctx *ctx, int32_t *d_buf, const symaccel_alac_desc *d_desc,
                                         const int32_t *d_coeffs, const int32_t *d_pair_weight, const uint8_t *d_pair_shift,
                                         size_t n_blocks, size_t blocksize, size_t stride);
/* decorrelate_mid_side (lib.rs:664-671) over n_pairs channel pairs: weight[pair] (0 = pair left alone, lib.rs:552),
 * shift[pair] (<= 31, lib.rs:555); ch0/ch1[pair][blocksize]. */
int symaccel_alac_mid_side_device(symaccel_ctx *ctx, const int32_t *d_weight, const uint8_t *d_shift,
                                  int32_t *d_ch0, int32_t *d_ch1, size_t n_pairs, size_t blocksize);
int symaccel_alac_mid_side(symaccel_ctx *ctx, const int32_t *h_weight, const uint8_t *h_shift, int32_t *h_ch0,
                           int32_t *h_ch1, size_t n_pairs, size_t blocksize);

/* ------------------------------------------------------------------ cross-stream batcher (csrc/batcher.cpp)
 * AudioDecoder::decode_ref (symphonia-core/src/codecs/audio.rs:279-297) sees one packet of one track and the registry builds
 * every decoder from (params, opts) alone (codecs/registry.rs:330-341): a decoder cannot see its siblings, so N decoders
 * batching their own look-ahead are N small launches and N PCIe round trips.  The batcher is the coalescing point below the
 * trait: decoders of one process share it, SUBMIT their batches and come back for the results; whatever is pending when
 * somebody needs a result -- or once `flush_bytes` of input have piled up -- goes to the device as ONE batch per
 * (kind, param, units_per_chain) group, the chains of all submissions side by side in the chain-major layout of the entry
 * points above.  Results are bit-identical to the per-stream calls (chains are independent).
 *
 * A submission has the planes of the entry point its kind names, all chain-major over the submission's own chains:
 *   SYMACCEL_BATCH_AAC_SYNTH   symaccel_aac_synth:  in = { coeffs[chain][unit][1024] f32, side[chain][unit] u8 };
 *                              state = { delay[chain][1024] }; out = pcm[chain][unit][1024]; param ignored
 *   SYMACCEL_BATCH_MP3_SYNTH   symaccel_mp3_synth:  in = { xr[chain][unit][576] f32, side[chain][unit] (symaccel_mp3_side) };
 *                              state = { overlap[chain][576], vvec[chain][1024], vfront[chain] i32 }; out = pcm[chain][unit][576];
 *                              param = sample_rate_idx
 *   SYMACCEL_BATCH_MP3_DECODE  symaccel_mp3_decode_pipelined for ONE stream (n_chains = 1, or 2 = one channel pair):
 *                              in = { quant[chain][unit][576] i16, rq_desc[chain][unit], side[chain][unit], st_desc[unit] (one row
 *                              per SUBMISSION; ignored for n_chains = 1) }; state / out / param as MP3_SYNTH
 *   SYMACCEL_BATCH_VORBIS_SYNTH symaccel_vorbis_synth with every chain's planes at their largest, so that submissions of one
 *                              block-size pair and block count share a launch whatever their block flags:
 *                              in = { spectra[chain][unit * bs1 / 2] f32 (the chain's packed spectrum at the front), block_flag[chain][unit] };
 *                              state = { prev_flag[chain] i32, overlap[chain][bs1 / 2] }; out = pcm[chain][unit * bs1 / 2] (the chain's
 *                              packed PCM at the front: spec_stride = pcm_stride = unit * bs1 / 2); param = bs0_exp | bs1_exp << 8.
 *                              Only the lines / samples the flags account for cross the link.
 *   SYMACCEL_BATCH_AAC_DECODE  symaccel_aac_decode_pipelined for ONE stream (its n_chains channels, any number of them paired):
 *                              in = { coeffs, side as AAC_SYNTH, blob[chain] }; state / out as AAC_SYNTH; param = the index
 *                              symaccel_batcher_aac_bands() gave the stream's scale-factor-band tables (-1: the submission has no
 *                              jointly coded pair and reads no table -- it shares launches with streams of any table).  The blob (every chain
 *                              contributes 64 + units * (322 + 8 * 92) bytes to ONE contiguous region of the submission) holds
 *                              { u32 n_pairs, n_tns, 0, 0 }, pair_chains[n_pairs][2] i32 (chains of the submission), padded to 16
 *                              bytes, js_desc[n_pairs][unit] (644 B each), padded to 16 bytes, tns[n_tns] (92 B each, frame = chain *
 *                              units + frame inside the submission): symaccel_batcher_submit_aac_decode() writes it.  Per launch the
 *                              pair frames that carry TNS get their joint stereo decoded by a list pass, the filters of the whole
 *                              group run, and ONE walk synthesises every stream of the group (cpe.rs:110-157, ics/mod.rs:449-468).
 *   SYMACCEL_BATCH_VORBIS_DECODE symaccel_vorbis_decode for ONE stream (n_chains = its channels): residue + floor-1 posts + coupling steps in,
 *                              PCM out (lib.rs:250-331) -- inverse coupling, the floor curves as one byte per line, the dot product in the
 *                              synthesis kernel's load path.  in = { residue[chain][unit * bs1 / 2] f32 (packed at the front, zeros for
 *                              a do-not-decode channel), block_flag[chain][unit], floor[chain][unit] u8 (the index
 *                              symaccel_batcher_vorbis_floor() gave the configuration, or SYMACCEL_VORBIS_FLOOR_UNUSED),
 *                              posts[chain][unit][65] u32, coupling blob (ONE per submission: first[unit + 1] u32 padded to 16 bytes, then
 *                              the (magnitude, angle) channel byte pairs of block b at [first[b], first[b + 1]); room for max(8, n_chains *
 *                              (n_chains - 1)) steps per block on average, SYMACCEL_ERR_INVALID_ARG from submit beyond) }; state / out as
 *                              VORBIS_SYNTH; param = bs0_exp | bs1_exp << 8 | n_chains << 16, so that streams of one block-size pair,
 *                              channel count and block count share a launch whatever their flags, floors and coupling
 *   SYMACCEL_BATCH_FLAC_RESTORE symaccel_flac_restore across streams: a chain is ONE SUBFRAME, units_per_chain = the block size (words per
 *                              subframe slot; shorter blocks zero-padded), so submissions of any number of frames share a launch:
 *                              in = { buf[chain][unit] i32 (warm-up samples + residuals; the RESULT overwrites it: slot.out == slot.input[0]),
 *                              desc[chain] (symaccel_flac_desc), coeffs[chain][32] i32 }; no state; param = 0.  With param = 0x100 | out_shift
 *                              the chains are channel pairs (2p, 2p + 1), in[3] = pair_mode[chain / 2] u8, and the decorrelation and the
 *                              left-justification happen in the same kernel (symaccel_flac_restore_stereo_device, decoder.rs:199-242)
 *   SYMACCEL_BATCH_ALAC_PREDICT symaccel_alac_predict across streams, the same shape: in = { buf[chain][unit] i32 (in place), desc[chain]
 *                              (symaccel_alac_desc), coeffs[chain][32] }; param = 0.  With param = 0x100: in[3] = pair_weight[chain / 2] i32,
 *                              in[4] = pair_shift[chain / 2] u8 (symaccel_alac_predict_stereo_device, lib.rs:541-560)
 * `units_per_chain` = frames (AAC) / granules (MP3) / blocks (Vorbis) / words (FLAC, ALAC) per chain.  Two forms:
 *   zero-copy:  reserve() hands out a slot of page-locked staging memory (the front end writes its output straight into the DMA
 *               source), commit() says it is filled, wait() blocks until slot.out / slot.state hold the PCM and the state AFTER
 *               the batch, release() gives the slot back.  Commit a reservation before waiting for anything on the same thread.
 *   copy:       submit() = reserve + memcpy + commit from caller memory; collect() = wait + memcpy into the `state_io` / `out`
 *               pointers given to submit() + release.  The `in` planes are free again when submit() returns; `state_io` and `out`
 *               must stay valid until collect().
 * A stream submits batch n + 1 only after batch n was collected (the state it starts from).  Thread-safe; the context is driven
 * through the batcher only while one exists.  Status is kept PER TICKET: a submission whose own descriptors do not add up (a
 * malformed AAC blob, a FLAC order above the block size, floor1_Y values above 511 ...) runs as an empty description and fails
 * alone with the status the per-stream entry point would have returned; a reservation released before commit() runs as zeros;
 * only a device error fails every ticket of the launch.  A group that closes is enqueued on one of the batcher's LANES (a context
 * of its own -- kernel stream, scratch, tables -- plus two copy streams; lane 0 is the caller's context) without the batcher's
 * mutex: other threads keep submitting and collecting meanwhile, and the gather of one group overlaps the scatter of another. */
#define SYMACCEL_BATCH_AAC_SYNTH 1
#define SYMACCEL_BATCH_MP3_SYNTH 2
#define SYMACCEL_BATCH_MP3_DECODE 3
#define SYMACCEL_BATCH_VORBIS_SYNTH 4
#define SYMACCEL_BATCH_AAC_DECODE 5
#define SYMACCEL_BATCH_VORBIS_DECODE 6
#define SYMACCEL_BATCH_FLAC_RESTORE 7
#define SYMACCEL_BATCH_ALAC_PREDICT 8
#define SYMACCEL_BATCH_MAX_INPUTS 6
typedef struct symaccel_batcher symaccel_batcher;
typedef struct symaccel_batch_slot {
    void *input[6]; /* SYMACCEL_BATCH_MAX_INPUTS */
    void *state[3];
    void *out; /* FLAC / ALAC: == input[0] (in place) */
    size_t input_bytes[6]; /* sizes of this submission's planes */
    size_t state_bytes[3];
    size_t out_bytes;
} symaccel_batch_slot;
typedef struct symaccel_batcher_stats {
    uint64_t submissions;           /* reserve() / submit() calls */
    uint64_t launches;              /* groups sent to the device */
    uint64_t chunks;                /* kernel launches (a group is cut into chunks of submissions for the copy / compute overlap) */
    uint64_t chains_launched;       /* sum of chains over all launches */
    uint64_t max_chains_per_launch;
    uint64_t staging_bytes;         /* page-locked memory held */
    uint64_t pending;               /* submissions not yet launched */
    uint64_t failed_tickets;        /* submissions that came back with a status of their own (or of a failed launch) */
    uint64_t lanes;                 /* pipelines in use (contexts + copy-stream pairs) */
    uint64_t mutex_wait_ns;         /* time callers spent blocked on the batcher's mutex, summed over threads ... */
    uint64_t mutex_contended;       /* ... and how many acquisitions found it taken */
    uint64_t launch_host_ns;        /* host time spent building copy descriptors and enqueueing launches (outside the mutex) */
    uint64_t lane_wait_ns;          /* time launchers waited for their lane (another group being enqueued on it) */
    uint64_t launch_api_ns;         /* the part of launch_host_ns spent inside HIP launch / event calls */
    uint64_t group_allocs;          /* device / page-locked allocations made for launches (0 in the steady state) */
    uint64_t flag_wait_ns;          /* time callers spent waiting for a launch's completion flag (device + link time they could not hide) */
    uint64_t slots_peak;            /* most submissions alive at once (reserved and not yet released) */
    uint64_t blocks;                /* device-side blocks in the pool (memory + descriptors + events; reused as soon as a launch completes) */
    uint64_t commit_to_launch_ns;   /* summed over submissions: commit -> the closing of their group (how long they sat pending) */
    uint64_t waits, waits_blocked;  /* waits on a completion word; those that found it not yet written */
    uint64_t launch_to_done_ns, launches_timed; /* for launches somebody had to wait for: enqueue finished -> completion seen, summed; how many */
} symaccel_batcher_stats;
/* flush_bytes: input bytes of one group after which it is launched without anybody waiting (0 = 64 MiB); also sizes the
 * staging memory of a group (input + output + state, page-locked, pooled and reused). */
int symaccel_batcher_create(symaccel_ctx *ctx, size_t flush_bytes, symaccel_batcher **out);
int symaccel_batcher_destroy(symaccel_batcher *b);
/* lanes: how many pipelines closing groups are dealt to (1..8, 0 = leave as is; default 2 -- lanes beyond the first are contexts the
 * batcher creates on first use); hint_bytes: what a pending group must hold for symaccel_batcher_hint() to launch it (0 = leave as is) */
int symaccel_batcher_configure(symaccel_batcher *b, int lanes, size_t hint_bytes);
/* the text of the last device error a launch or a wait of this batcher met ("" if none), copied under the batcher's mutex */
int symaccel_batcher_last_error(symaccel_batcher *b, char *buf, size_t capacity);
int symaccel_batcher_reserve(symaccel_batcher *b, int kind, int param, size_t n_chains, size_t units_per_chain,
                             symaccel_batch_slot *slot, uint64_t *ticket);
int symaccel_batcher_commit(symaccel_batcher *b, uint64_t ticket);
/* slot may be NULL; otherwise it is filled in again (same pointers as reserve() gave) */
int symaccel_batcher_wait(symaccel_batcher *b, uint64_t ticket, symaccel_batch_slot *slot);
int symaccel_batcher_release(symaccel_batcher *b, uint64_t ticket);
int symaccel_batcher_submit(symaccel_batcher *b, int kind, int param, size_t n_chains, size_t units_per_chain,
                            const void **input, void **state_io, void *out, uint64_t *ticket); /* input[6], state_io[3] */
/* submit() with the argument lists of the entry points the kinds stand for (symaccel_aac_synth, symaccel_mp3_synth,
 * symaccel_mp3_decode_pipelined for one stream: n_chains 1, or 2 = one channel pair with st_desc[granule]; st_desc may be NULL for 1) */
int symaccel_batcher_submit_aac_synth(symaccel_batcher *b, const float *coeffs, const uint8_t *side, float *delay_io, float *pcm,
                                      size_t n_chains, size_t frames_per_chain, uint64_t *ticket);
int symaccel_batcher_submit_mp3_synth(symaccel_batcher *b, const float *xr, const symaccel_mp3_side *side, int sample_rate_idx,
                                      float *overlap_io, float *vvec_io, int32_t *vfront_io, float *pcm, size_t n_chains,
                                      size_t granules_per_chain, uint64_t *ticket);
int symaccel_batcher_submit_mp3_decode(symaccel_batcher *b, const int16_t *quant, const symaccel_mp3_requant *rq_desc,
                                       const symaccel_mp3_stereo *st_desc, const symaccel_mp3_side *side, int sample_rate_idx,
                                       float *overlap_io, float *vvec_io, int32_t *vfront_io, float *pcm, size_t n_chains,
                                       size_t granules_per_chain, uint64_t *ticket);
/* Register the scale-factor-band offset tables AAC_DECODE submissions refer to (as symaccel_aac_joint_stereo_device takes them: n + 1
 * offsets each): the same tables give the same index; at most 64 distinct ones per batcher.  The index is the submissions' `param`. */
int symaccel_batcher_aac_bands(symaccel_batcher *b, const uint16_t *swb_long, int n_swb_long, const uint16_t *swb_short, int n_swb_short,
                               int *bands);
int symaccel_batcher_submit_aac_decode(symaccel_batcher *b, int bands, const float *coeffs, const uint8_t *side,
                                       const int32_t *pair_chains, const symaccel_aac_js_frame *js_desc, size_t n_pairs,
                                       const symaccel_aac_tns_filter *tns, size_t n_tns, float *delay_io, float *pcm, size_t n_chains,
                                       size_t frames_per_chain, uint64_t *ticket);
int symaccel_batcher_submit_vorbis_synth(symaccel_batcher *b, int bs0_exp, int bs1_exp, const float *spectra, const uint8_t *block_flag,
                                         int32_t *prev_flag_io, float *overlap_io, float *pcm, size_t n_chains, size_t blocks_per_chain,
                                         uint64_t *ticket); /* spectra / pcm: [chain][blocks_per_chain * bs1 / 2], packed at the front */
/* Register a floor-1 configuration VORBIS_DECODE submissions refer to (floor.rs:510-555: multiplier, x list in bitstream order): the
 * same configuration gives the same index; at most 255 distinct ones per batcher (SYMACCEL_ERR_UNSUPPORTED beyond: such a stream
 * keeps batching per stream through symaccel_vorbis_decode). */
int symaccel_batcher_vorbis_floor(symaccel_batcher *b, const symaccel_vorbis_floor1_cfg *cfg, int *index);
/* symaccel_vorbis_decode's argument list for ONE stream (posts_stride 65; `floor` holds symaccel_batcher_vorbis_floor() indices) */
int symaccel_batcher_submit_vorbis_decode(symaccel_batcher *b, int bs0_exp, int bs1_exp, const float *residue, const uint8_t *block_flag,
                                          const uint8_t *floor, const uint32_t *posts, const uint8_t *coupling,
                                          const uint32_t *coupling_first, int32_t *prev_flag_io, float *overlap_io, float *pcm,
                                          size_t n_chains, size_t blocks_per_chain, uint64_t *ticket);
/* symaccel_flac_restore (pair_mode NULL, out_shift 0) / symaccel_flac_restore_stereo_device (pair_mode[n_blocks / 2]) for the subframes
 * of one stream's batch; collect() writes the result over buf_io */
int symaccel_batcher_submit_flac_restore(symaccel_batcher *b, int32_t *buf_io, const symaccel_flac_desc *desc, const int32_t *coeffs,
                                         const uint8_t *pair_mode, uint32_t out_shift, size_t n_blocks, size_t blocksize, uint64_t *ticket);
/* symaccel_alac_predict (pair_weight / pair_shift NULL) / symaccel_alac_predict_stereo_device for one stream's batch, in place */
int symaccel_batcher_submit_alac_predict(symaccel_batcher *b, int32_t *buf_io, const symaccel_alac_desc *desc, const int32_t *coeffs,
                                         const int32_t *pair_weight, const uint8_t *pair_shift, size_t n_blocks, size_t blocksize,
                                         uint64_t *ticket);
/* wait + copy the PCM and the state after the batch into the `*_io` / `pcm` pointers given to submit + release */
int symaccel_batcher_collect(symaccel_batcher *b, uint64_t ticket);
/* give up a submission (seek, reset): wait until nothing of it is in flight, write nothing, release */
int symaccel_batcher_abandon(symaccel_batcher *b, uint64_t ticket);
/* launch everything pending now (nobody has to wait for it) */
int symaccel_batcher_flush(symaccel_batcher *b);
/* "results will be wanted soon": launch the pending groups that are worth a launch of their own (4 MiB of input, or flush_bytes / 8),
 * leave smaller ones to grow -- what a decoder calls when a quarter of its current batch is left */
int symaccel_batcher_hint(symaccel_batcher *b);
/* bytes per chain of every plane of a kind (per submission for MP3_DECODE's input[3]); unused planes 0 */
int symaccel_batcher_plane_bytes(int kind, int param, size_t units_per_chain, size_t *input_bytes, size_t *state_bytes, size_t *out_bytes); /* [6], [3] */
int symaccel_batcher_get_stats(symaccel_batcher *b, symaccel_batcher_stats *out);

/* ------------------------------------------------------------------------- multi-GPU */

/* Chains are independent, so a batch shards across GPUs by whole streams with no data-path collective (the unit is
 * the packet loop of symphonia-codec-vorbis/src/lib.rs:296-331, one stream's channels).  One process (or thread) per GPU, one
 * context each.  symaccel_shard_range gives rank's contiguous, balanced slice [first, first + count) of n_streams streams (the
 * first n_streams % world ranks take one more). */
int symaccel_shard_range(size_t n_streams, int world, int rank, size_t *first, size_t *count);

/* For a batch that starts and ends on ONE rank (BASELINE config 4's "batch split / gather"): the root sends every rank its slice
 * of a stream-major device buffer d_all[n_streams][bytes_per_stream] (scatter), or receives the slices back (gather); every
 * rank's own slice is d_mine[count][bytes_per_stream].  ncclSend / ncclRecv inside one group on `comm` (an ncclComm_t of RCCL,
 * one per rank) and the context's stream: the root's world - 1 transfers run over its xGMI links side by side, its own slice is
 * a device-to-device copy.  Asynchronous like every *_device call.  world == 1 needs no communicator.  One-to-all traffic is
 * bounded by the root's links (7 x ~153 GB/s): a deployment that can keeps producers per GPU instead.
 * librccl.so is loaded at first use (SYMACCEL_RCCL_LIB overrides the name); SYMACCEL_ERR_UNSUPPORTED if there is none. */
int symaccel_scatter_streams(symaccel_ctx *ctx, void *comm, int world, int rank, int root, const void *d_all, void *d_mine,
                             size_t n_streams, size_t bytes_per_stream);
int symaccel_gather_streams(symaccel_ctx *ctx, void *comm, int world, int rank, int root, const void *d_mine, void *d_all,
                            size_t n_streams, size_t bytes_per_stream);

/* scatter -> step -> gather, chunked and overlapped.  Every rank's slice (symaccel_shard_range) is cut into n_chunks pieces of whole
 * streams; on a second stream of the context chunk c + 1 travels from the root to the ranks while `step(user, first, count)` --
 * the caller's synthesis calls on streams [first, first + count) of ITS slice, queued on the context's stream -- processes chunk
 * c and chunk c - 1's result travels back.  With the root's link busy in both directions at once the exchange costs about
 * max(scatter, gather) + one chunk instead of scatter + step + gather.  d_all_in / d_all_out: the whole batch on the root
 * (ignored elsewhere); d_mine_in / d_mine_out: this rank's slice, in_bytes_per_stream / out_bytes_per_stream each stream.  The
 * result is complete once the context's stream is synchronised.  A step that returns non-zero aborts THIS rank's call with
 * SYMACCEL_ERR_DEVICE; the other ranks are not told (their transfers with it never complete: tear the communicator down).  Uses the
 * context's staging stream and events: not concurrently with a *_pipelined host-memory call on the same context. */
typedef int (*symaccel_step_fn)(void *user, size_t first_local_stream, size_t n_local_streams);
int symaccel_exchange_pipelined(symaccel_ctx *ctx, void *comm, int world, int rank, int root, const void *d_all_in,
                                void *d_mine_in, size_t in_bytes_per_stream, void *d_all_out, void *d_mine_out,
                                size_t out_bytes_per_stream, size_t n_streams, int n_chunks, symaccel_step_fn step,
                                void *user);

/* Communicator set-up without RCCL's headers: rank 0 makes an id (ncclGetUniqueId) and hands its 128 bytes to the other ranks by
 * whatever means the host has (a pipe, MPI, torch.distributed); every rank then creates its communicator on its context's device
 * (ncclCommInitRank: collective, call it on all ranks) and destroys it at the end. */
typedef struct symaccel_unique_id {
    char internal[128];
} symaccel_unique_id;
int symaccel_comm_unique_id(symaccel_unique_id *id);
int symaccel_comm_init(symaccel_ctx *ctx, const symaccel_unique_id *id, int world, int rank, void **comm);
int symaccel_comm_destroy(void *comm);

/* A caller-supplied point-to-point transport instead of RCCL (MPI, shared memory, a test double): send / recv move `bytes`
 * bytes of DEVICE memory to / from rank `peer` on `stream` and return 0 on success; `comm` is passed through untouched;
 * group_start / group_end may be null.  NULL restores RCCL.  Process-wide; set it before the first exchange. */
typedef struct symaccel_transport {
    int (*group_start)(void);
    int (*group_end)(void);
    int (*send)(const void *d_buf, size_t bytes, int peer, void *comm, void *stream);
    int (*recv)(void *d_buf, size_t bytes, int peer, void *comm, void *stream);
} symaccel_transport;
int symaccel_multi_set_transport(const symaccel_transport *transport);

/* ------------------------------------------------------------------ measurement probe */

/* Device-to-device copy of `bytes` (a multiple of 4096; distinct, 16-byte-aligned buffers) with the traffic shape of the
 * synthesis kernels -- every byte read once and written once -- so that a benchmark can quote a workload against the copy
 * rate the SAME run reaches.  frames_per_wavefront 0: a plain grid-stride 16 B/lane copy; k > 0: every
 * wavefront streams k consecutive 4 KiB frames, the access pattern of a wavefront that walks a k-frame segment of one
 * chain.  flags bit 0: non-temporal loads and stores (what the synthesis kernels use); bits 1-2 (k > 0 only): 0 copy, 1 read
 * only, 2 write only; bit 3 (k > 0 only): the four wavefronts of a workgroup share 4 k consecutive frames round-robin
 * (16 KiB contiguous per workgroup and step); bit 4: eight wavefronts (two neighbouring workgroups) share 8 k frames, bits 3 + 4:
 * sixteen; bit 5 (with bit 3 alone, copy): the workgroup walk over a window-major layout, [step][workgroup][4 frames] -- the grid-wide
 * footprint of a step is one contiguous window.  Not part of any decode path. */
int symaccel_probe_copy_device(symaccel_ctx *ctx, const void *d_src, void *d_dst, size_t bytes,
                               uint32_t frames_per_wavefront, uint32_t flags);

/* ------------------------------------------------------------- table read-back (for tests) */

enum symaccel_table {
    SYMACCEL_TABLE_AAC_KBD_LONG = 0,   /* 1024 f32, window.rs:37-52 alpha 4 */
    SYMACCEL_TABLE_AAC_KBD_SHORT = 1,  /* 128 */
    SYMACCEL_TABLE_AAC_SINE_LONG = 2,  /* 1024, window.rs:30-35 */
    SYMACCEL_TABLE_AAC_SINE_SHORT = 3, /* 128 */
    SYMACCEL_TABLE_MP3_SYNTH_D = 4,    /* 512, synthesis.rs:13-142 */
    SYMACCEL_TABLE_MP3_IMDCT_WIN = 5,  /* 4*36, hybrid_synthesis.rs:53-92 */
    SYMACCEL_TABLE_VORBIS_FLOOR1_DB = 6, /* 256, vorbis floor.rs:21-112 */
    SYMACCEL_TABLE_MP3_CONSTS = 7,      /* 264: the hybrid-synthesis / dct32 constants in the kernels' packed order */
    SYMACCEL_TABLE_MP3_POW43 = 8,       /* 8207, requantize.rs:28-31 */
    SYMACCEL_TABLE_MP3_POW2AB = 9       /* 1346: 2^(0.25 e), e = -1300 .. 45 (requantize.rs:280, 343) */
};
/* Copies the HOST copy of a constant table; returns the number of floats, or a negative status. */
int symaccel_table_f32(const symaccel_ctx *ctx, int table, float *dst, size_t capacity);
/* Imdct twiddles (n/2 complex) and FFT merge twiddles W_n (n/2 complex), as generated on the host. */
int symaccel_imdct_twiddles(int n, double scale, float *dst);
int symaccel_fft_twiddles(int n, float *dst);

#ifdef __cplusplus
}
#endif
#endif /* SYMACCEL_H */
