/*
 * symoracle -- CPU restatement of Symphonia's DSP hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing in the product (symphonia_amd/, the
 * libsymaccel C-ABI) may include, link or call this.  Only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg use it, and only as
 * the checker / the timed CPU baseline -- never as the thing shipped.
 *
 * What it restates: the reference's in-tree (non-SIMD) scalar path, operation
 * for operation (same multiplies/adds on the same operands, no FMA
 * contraction: build with -O2 -ffp-contract=off -fno-fast-math).  Each function
 * cites the reference file:line it follows (paths relative to the Symphonia
 * tree, workspace version 0.6.1).
 *
 * Pinning status: the reference (Rust) cannot be compiled in this image, so no
 * oracle/_ref exists.  Instead the reference's own source TEXT is executed:
 * tools/rsinterp (an interpreter for the subset of Rust the DSP code is written
 * in) runs the functions of every hot-path row from /root/reference on seeded
 * inputs, and tools/rs2fixtures.py commits inputs and outputs as bit patterns
 * under tests/golden/rs_fixtures/ (manifest.json lists the reference file:line
 * of every entry).  tests/test_rs_fixtures.py holds every function below to
 * those fixtures BIT FOR BIT: Fft / Ifft / Imdct, AAC Dsp::synth and windows,
 * MP3 imdct36 / imdct12_win / antialias / reorder / dct32 / synthesis /
 * requantize / stereo, Vorbis synth / overlap_add / coupling / floor-0 /
 * floor-1, FLAC and ALAC predictors and decorrelation, AAC joint stereo / TNS /
 * pulse.  The reference's in-source known-answer tests (tests/golden/
 * ref_kats.json, at the reference's own 1e-5 tolerance), f64 closed forms and
 * encoder identities remain as an independent second pin.
 * Not pinned: the reference's `opt-simd` build, which delegates to rustfft
 * (un-vendored): against that build only the 1e-5 criterion is meaningful.
 */
#ifndef SYMORACLE_H
#define SYMORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- core dsp (symphonia-core/src/dsp) ---------------------------------- */

/* Fft::fft_inplace (fft/no_simd.rs:96-118).  x = n interleaved (re,im). */
void so_fft_inplace(float *x, int n);
/* Fft::fft (fft/no_simd.rs:121-140). */
void so_fft(const float *x, float *y, int n);
void so_ifft(const float *x, float *y, int n);

typedef struct so_imdct so_imdct;
/* Imdct::new_scaled (mdct.rs:35-60). */
so_imdct *so_imdct_new(int n, double scale);
void so_imdct_free(so_imdct *m);
/* Imdct::imdct (mdct.rs:67-146): spec[n] -> out[2n]. */
void so_imdct_run(so_imdct *m, const float *spec, float *out);
/* count back-to-back transforms, spec[count][n] -> out[count][2n]. */
void so_imdct_batch(int n, double scale, const float *spec, float *out, size_t count);
/* read back the twiddle table (n/2 complex) for table-parity tests. */
void so_imdct_twiddles(int n, double scale, float *tw_out);
/* FFT merge twiddles W_n[k], k < n/2 (fft/no_simd.rs:16-36). */
void so_fft_twiddles(int n, float *tw_out);

/* constant-table read-back, for the literal-parity tests */
void so_fft_small_twiddles(int n, float *tw_out);
void so_mp3_constants(float *dst117);
void so_mp3_sfb_tables(int sample_rate_idx, int32_t *dst81);
void so_mp3_sfb_long(int sample_rate_idx, int32_t *dst23);

/* Requantisation (layer3/requantize.rs).  NO test in the reference: parity unpinned by the reference; pinned in
 * tests/ by the ISO 11172-3 closed form xr = sign(s) |s|^(4/3) 2^(0.25 (A - B)) in f64 and by the band tables
 * recorded in tests/golden/ref_kats.json. */
#define SO_MP3_RQ_SCALEFAC_SCALE 1
#define SO_MP3_RQ_PREFLAG 2
#define SO_MP3_POW2AB_MIN_E (-1300) /* >= -210 - 8*7 - ((255 + 3) << 2) */
#define SO_MP3_POW2AB_LEN 1346      /* up to e = 45 = 255 - 210 */
typedef struct so_mp3_requant {     /* the GranuleChannel fields requantize reads (layer3/mod.rs) */
    uint8_t global_gain;
    uint8_t flags;                  /* SO_MP3_RQ_* */
    uint8_t block_type;             /* SO_MP3_* */
    uint8_t is_mixed;
    uint8_t subblock_gain[3];
    uint8_t reserved;
    uint16_t rzero;
    uint8_t scalefacs[39];
    uint8_t pad[3];
} so_mp3_requant;                   /* 52 bytes */
/* Joint stereo (layer3/stereo.rs:485-556).  NO test in the reference: parity unpinned by the reference; pinned in
 * tests/ by an independent numpy restatement of the band scan and the ISO ratio closed forms. */
#define SO_MP3_ST_MID_SIDE 1   /* Mode::Layer3 { mid_side, .. } */
#define SO_MP3_ST_INTENSITY 2  /* Mode::Layer3 { .., intensity } */
#define SO_MP3_ST_MPEG1 4      /* FrameHeader::is_mpeg1() */
#define SO_MP3_ST_IS_SCALE 8   /* channels[1].scalefac_compress & 1 (MPEG-2 / 2.5 ratio table select) */
typedef struct so_mp3_stereo_desc {
    uint8_t flags;       /* SO_MP3_ST_* */
    uint8_t block_type;  /* of both channels (stereo.rs:502-504) */
    uint8_t is_mixed;
    uint8_t reserved;
    uint16_t rzero0, rzero1;
    uint8_t scalefacs1[39]; /* channels[1].scalefacs: the intensity positions */
    uint8_t pad;
} so_mp3_stereo_desc;    /* 48 bytes */
void so_mp3_intensity_ratios(float *mpeg1_7x2, float *mpeg2_2x32x2);
void so_mp3_stereo(float *ch0_576, float *ch1_576, const so_mp3_stereo_desc *d, int sample_rate_idx);
void so_mp3_pow43(float *dst8207);
void so_mp3_pow2ab(float *dst /* SO_MP3_POW2AB_LEN */);
/* is576: the signed quantised samples the Huffman stage decodes (|s| <= 8206); xr576 out. */
void so_mp3_requantize(const int16_t *is576, const so_mp3_requant *ch, int sample_rate_idx, float *xr576);
void so_mp3_requantize_batch(const int16_t *is, const so_mp3_requant *ch, int sample_rate_idx, float *xr, size_t n);
void so_vorbis_floor1_table(float *dst256);

/* ---- AAC-LC (symphonia-codec-aac/src/aac/{dsp,window}.rs) ---------------- */

/* generate_window (window.rs:28-52). kind 0 = sine, 1 = KBD(alpha). half=true. */
void so_aac_window(int kbd, float alpha, int size, float *dst);
/* Dsp::synth (dsp.rs:57-158) for one channel-frame. */
void so_aac_synth(const float *coeffs, float *delay, int seq, int window_shape,
                  int prev_window_shape, float *dst);
/* AAC spectral tools in front of Dsp::synth: joint stereo (aac/cpe.rs:110-157) and one TNS filter
 * (aac/ics/tns.rs:180-195).  NO test in the reference: parity unpinned by the reference; pinned in tests/ by the
 * defining arithmetic (sum/difference, scalar multiply) and by scipy's all-pole filter in f64. */
void so_aac_joint_stereo(float *left1024, float *right1024, int num_windows, int max_sfb, const uint16_t *bands,
                         const uint8_t *mode128, const float *scale128);
void so_aac_tns_filter(float *coeffs1024, int start, int end, int order, int direction, const float *lpc);

/* n_chains independent channels, frames_per_chain consecutive frames each.
 * coeffs[chain][frame][1024], side[chain][frame] = seq | shape<<2 | prev<<3,
 * delay[chain][1024] in/out, pcm[chain][frame][1024]. */
void so_aac_synth_batch(const float *coeffs, const uint8_t *side, float *delay, float *pcm,
                        size_t n_chains, size_t frames_per_chain);

/* ---- MP3 (symphonia-bundle-mp3/src/{layer3/hybrid_synthesis,synthesis}.rs) */

enum { SO_MP3_LONG = 0, SO_MP3_START = 1, SO_MP3_SHORT = 2, SO_MP3_END = 3 };

/* reorder (hybrid_synthesis.rs:153-215); returns the updated rzero. */
int so_mp3_reorder(float *buf576, int block_type, int is_mixed, int sample_rate_idx, int rzero);
/* antialias (hybrid_synthesis.rs:218-277); returns the updated rzero. */
int so_mp3_antialias(float *buf576, int block_type, int is_mixed, int rzero);
/* hybrid_synthesis (hybrid_synthesis.rs:280-359). overlap[32][18]. */
void so_mp3_hybrid(float *buf576, float *overlap, int block_type, int is_mixed, int rzero);
void so_mp3_imdct36(float *x18, const float *window36, float *overlap18);
void so_mp3_imdct12_win(float *x18, const float *window36, float *overlap18);
/* IMDCT_WINDOWS[4][36] (hybrid_synthesis.rs:53-92). */
void so_mp3_imdct_windows(float *dst144);
/* frequency_inversion (hybrid_synthesis.rs:458-485). */
void so_mp3_frequency_inversion(float *buf576);
/* dct32 (synthesis.rs:348-844). */
void so_mp3_dct32(const float *x32, float *y32);
/* SYNTHESIS_D (synthesis.rs:13-142). */
void so_mp3_synthesis_window(float *dst512);
/* synthesis (synthesis.rs:158-336). v_vec[16][64], *v_front in/out. */
void so_mp3_polyphase(float *v_vec, int *v_front, int n_frames, const float *in, float *out);
/* Layer3 granule loop tail (layer3/mod.rs:421-477) for n_chains channels x
 * granules_per_chain: xr[chain][gr][576]; side[chain][gr] = {block_type u8,
 * is_mixed u8, rzero u16 little endian}; state in/out per chain:
 * overlap[32][18], v_vec[16][64], v_front (int32). pcm[chain][gr][576]. */
void so_mp3_synth_batch(const float *xr, const uint8_t *side, int sample_rate_idx,
                        float *overlap, float *v_vec, int32_t *v_front, float *pcm,
                        size_t n_chains, size_t granules_per_chain);

/* ---- Vorbis (symphonia-codec-vorbis/src/{dsp,window,lib,floor,residue}.rs) */

/* generate_win_curve (window.rs:11-24): left half, bs/2 values. */
void so_vorbis_window(int bs, float *dst);
/* inverse coupling of one (magnitude, angle) pair (lib.rs:265-277). */
void so_vorbis_inverse_coupling(float *magnitude, float *angle, size_t n);
/* dot product floor *= residue (lib.rs:289-291). */
void so_vorbis_dot_product(float *floor, const float *residue, size_t n);
/* residue type-2 de-interleave (residue.rs:177-218): type2[n_ch * n2] -> ch[c][n2]. */
void so_vorbis_deinterleave2(const float *type2, float *planar, int n_ch, size_t n2);
/* floor1 synthesis step 1 + step 2 (floor.rs:568-653, 776-825).
 * x_list[n_posts], y[n_posts] (decoded floor1_Y), multiplier 1..4;
 * neighbours / sort order are derived as the setup parser does (floor.rs:500-555).
 * floor_out[n] receives the rendered curve. */
void so_vorbis_floor1(const uint32_t *x_list, const uint32_t *y, int n_posts, int multiplier,
                      uint32_t n, float *floor_out);
/* DspChannel::synth chain (dsp.rs:68-126 + lib.rs:296-331) for n_chains
 * channels x blocks_per_chain.  spectra packed back to back per chain: a block
 * with flag f contributes bs_f/2 floats.  block_flag[chain][block] (0/1).
 * prev_flag[chain]: -1 = None (lib.rs:298), else 0/1; in/out.
 * overlap[chain][bs1/2] in/out.  pcm packed back to back per chain, block b
 * contributes (prev_n + n)/4 floats.  spec_stride / pcm_stride = floats per
 * chain in the packed arrays.  Returns 0. */
int so_vorbis_synth_batch(int bs0_exp, int bs1_exp, const float *spectra, size_t spec_stride,
                          const uint8_t *block_flag, int32_t *prev_flag, float *overlap,
                          float *pcm, size_t pcm_stride, size_t n_chains,
                          size_t blocks_per_chain);

/* ---- FLAC (symphonia-bundle-flac/src/decoder.rs) ------------------------- */

/* fixed_predict (decoder.rs:663-710). */
void so_flac_fixed_predict(int order, int32_t *buf, size_t len);
/* lpc_predict via the order dispatch of decode_linear (decoder.rs:487-504,
 * 716-752).  coeffs[order] in bitstream order (first coefficient multiplies
 * the most recent sample), as read at decoder.rs:479-481. */
void so_flac_lpc_predict(int order, const int32_t *coeffs, uint32_t shift, int32_t *buf,
                         size_t len);
/* decorrelate_{left_side,mid_side,right_side} (decoder.rs:32-82). mode:
 * 0 independent (no-op), 1 left/side, 2 mid/side, 3 right/side. ch0/ch1 are the
 * two planes in frame order (for right/side ch0 = side, ch1 = right). */
void so_flac_decorrelate(int mode, int32_t *ch0, int32_t *ch1, size_t len);
/* samples_shl (decoder.rs:403-409) and the final << (32-bps) (decoder.rs:239-242). */
void so_flac_shl(int32_t *buf, size_t len, uint32_t shift);
/* rice_signed_to_i32 (decoder.rs:634-644). */
int32_t so_flac_rice_signed_to_i32(uint32_t word);
/* n_blocks subframes of `blocksize` samples.  desc[block] = {kind u8 (0 none /
 * verbatim, 1 fixed, 2 lpc), order u8, shift u8, wasted_bits u8}; coeffs[block][32]. */
void so_flac_restore_batch(int32_t *buf, const uint8_t *desc, const int32_t *coeffs,
                           size_t n_blocks, size_t blocksize);

/* ---- ALAC (symphonia-codec-alac/src/lib.rs) ------------------------------- */

/* ElementChannel::predict (lib.rs:165-264).  Returns -1 for an invalid mode (lib.rs:167-169). */
int so_alac_predict(int32_t *out, size_t len, uint32_t mode, uint32_t lpc_order, uint32_t shift, uint32_t bps,
                    const int32_t *coeffs_in);
/* decorrelate_mid_side (lib.rs:664-671) */
void so_alac_decorrelate_mid_side(int32_t *out0, int32_t *out1, size_t len, int32_t weight, uint32_t shift);
/* desc[block] = {mode u8, lpc_order u8, shift u8, bps u8}; coeffs[block][32]. */
void so_alac_predict_batch(int32_t *buf, const uint8_t *desc, const int32_t *coeffs, size_t n_blocks, size_t blocksize);

#ifdef __cplusplus
}
#endif
/* ---- added in round 2: AAC pulse tool, Vorbis floor 0 (both libm-dependent) ---- */
void so_aac_iquant_requant(const float *val, float scale, float *iq, float *rq, size_t n);
void so_aac_pulse(float *coeffs, const int32_t *bands, int n_bands_plus_1, const float *scales0, int number_pulse,
                  int pulse_start_sfb, const int32_t *pulse_offset, const int32_t *pulse_amp);
void so_vorbis_bark_map(uint32_t n, uint32_t rate16, uint32_t map_size16, int32_t *map);
void so_vorbis_floor0_coeffs(float *coeffs, int order);
int so_vorbis_floor0(const float *coeffs, int order, const int32_t *map, uint32_t n, uint32_t bark_map_size,
                     uint32_t amplitude_bits, uint32_t amplitude_offset, uint64_t amplitude, float *floor_out);

/* cpu_simd.c: the headline workload (AAC-LC ONLY_LONG / KBD) with 16 chains per vector; n_chains % 16 == 0.  Same bits as
 * so_aac_synth_batch. */
void so_aac_long_kbd_batch_simd(const float *coeffs, float *delay, float *pcm, size_t n_chains, size_t frames_per_chain);

#endif
