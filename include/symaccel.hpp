// symaccel.hpp -- C++17 host-side mirror of the reference's DSP interfaces on top of the symaccel C ABI.
//
// The reference (pdeljanov/Symphonia 0.6.1) is Rust; this image has no Rust toolchain, so the host side
// above the C ABI is written in C++ (and, for the tests / bench harness, in Python: symphonia_amd/).  Every
// type below mirrors one reference item -- same name, same argument meaning, same error behaviour:
//
//   dsp::mdct::Imdct           symphonia-core/src/dsp/mdct.rs:16-146
//   dsp::fft::Fft              symphonia-core/src/dsp/fft/no_simd.rs:70-141
//   aac::Dsp                   symphonia-codec-aac/src/aac/dsp.rs:22-158
//   mp3::SynthesisState, mp3::GranuleChannel, mp3::synthesize_granule
//                              symphonia-bundle-mp3/src/synthesis.rs:145-336, layer3/hybrid_synthesis.rs:153-485,
//                              the per-channel tail of Layer3::decode (layer3/mod.rs:440-476)
//   vorbis::Windows / Dsp / DspChannel   symphonia-codec-vorbis/src/dsp.rs:12-145, window.rs:11-39
//   flac::lpc_predict / fixed_predict / decorrelate_*   symphonia-bundle-flac/src/decoder.rs:32-82, 663-752
//   alac::ElementChannel::predict / decorrelate_mid_side   symphonia-codec-alac/src/lib.rs:165-264, 664-671
//
// Error behaviour: what the reference asserts / panics on (slice lengths, non power-of-two sizes) throws
// std::invalid_argument; what it would return as Error::Unsupported throws Error{Kind::Unsupported}; HIP /
// device / allocation failures throw Error{Kind::IoError} (symphonia-core/src/errors.rs:38-54).  There is no
// CPU fallback: constructing a Context without an MI355X throws.
//
// The per-packet calls here are batches of one and are latency-bound by design (one launch + PCIe round trip
// per call); the *_batch members are the path the GPU is for.
#ifndef SYMACCEL_HPP
#define SYMACCEL_HPP

#include <array>
#include <algorithm>
#include <complex>
#include <functional>
#include <cstdint>
#include <memory>
#include <mutex>
#include <optional>
#include <set>
#include <stdexcept>
#include <string>
#include <utility>
#include <cstring>
#include <typeindex>
#include <vector>

#include "symaccel.h"

namespace symphonia_accel {

class Error : public std::runtime_error {
public:
    enum class Kind { IoError, Unsupported };  // errors.rs:38-54
    Error(Kind k, int status, const std::string &what) : std::runtime_error(what), kind(k), status(status) {}
    Kind kind;
    int status;
};

inline void check(int status, const symaccel_ctx *ctx = nullptr) {
    if (status >= 0) return;
    std::string msg = symaccel_strerror(status);
    if (ctx) {
        const char *extra = symaccel_last_error(ctx);
        if (extra && *extra) msg += std::string(": ") + extra;
    }
    if (status == SYMACCEL_ERR_INVALID_ARG) throw std::invalid_argument(msg);  // the reference's assert!/panic class
    throw Error(status == SYMACCEL_ERR_UNSUPPORTED ? Error::Kind::Unsupported : Error::Kind::IoError, status, msg);
}

// One context = one HIP device + stream + device-resident constant tables.  Externally synchronised, like
// `&mut self` on the reference's decoders; distinct contexts may be used from distinct threads.
class Context {
public:
    explicit Context(int device = 0) { check(symaccel_ctx_create(device, &ctx_)); }
    ~Context() {
        if (ctx_) symaccel_ctx_destroy(ctx_);
    }
    Context(const Context &) = delete;
    Context &operator=(const Context &) = delete;
    Context(Context &&o) noexcept : ctx_(std::exchange(o.ctx_, nullptr)) {}
    symaccel_ctx *raw() const { return ctx_; }
    void sync() { check(symaccel_sync(ctx_), ctx_); }
    void set_segment(int frames) { check(symaccel_ctx_set_segment(ctx_, frames), ctx_); }

private:
    symaccel_ctx *ctx_ = nullptr;
};

// The cross-stream batcher (symaccel_batcher_*, csrc/batcher.cpp): one per process and context, shared by every decoder of the
// process.  Decoders submit their look-ahead batches; one launch per (kind, units) group serves all of them.  Thread-safe.
class Batcher {
public:
    explicit Batcher(Context &ctx, std::size_t flush_bytes = 0) : ctx_(ctx) { check(symaccel_batcher_create(ctx.raw(), flush_bytes, &b_), ctx.raw()); }
    ~Batcher() {
        if (b_) symaccel_batcher_destroy(b_);
    }
    Batcher(const Batcher &) = delete;
    Batcher &operator=(const Batcher &) = delete;
    symaccel_batcher *raw() const { return b_; }
    Context &context() const { return ctx_; }
    void flush() { check(symaccel_batcher_flush(b_), ctx_.raw()); }
    void hint() { check(symaccel_batcher_hint(b_), ctx_.raw()); }
    symaccel_batcher_stats stats() const {
        symaccel_batcher_stats s{};
        check(symaccel_batcher_get_stats(b_, &s), ctx_.raw());
        return s;
    }
    // The process-wide batcher (the Rust shim's Pool::shared(), bindings/rust/symphonia-accel-hip/src/ctx.rs): device 0, the library's
    // default flush size, created on first use and never destroyed (decoders of any thread may hold it until the process ends).  This is
    // what a factory that sees (params, options) alone -- CodecRegistry::make_audio_decoder below, registry.rs:330-341 -- builds through.
    // Throws what Context / Batcher construction throws (no device: Error{IoError}); a failed first attempt is not retried.
    static Batcher &shared() {
        static std::once_flag once;
        static Batcher *inst = nullptr;
        static std::string failure;
        std::call_once(once, [] {
            try {
                auto *ctx = new Context(0);
                try {
                    inst = new Batcher(*ctx, 0);
                } catch (...) {
                    delete ctx;
                    throw;
                }
            } catch (const std::exception &e) {
                failure = e.what();
            }
        });
        if (!inst) throw Error(Error::Kind::IoError, SYMACCEL_ERR_DEVICE, "shared batcher: " + failure);
        return *inst;
    }

private:
    Context &ctx_;
    symaccel_batcher *b_ = nullptr;
};

namespace dsp {
namespace mdct {

// Imdct (mdct.rs:16-20): N spectral lines in, 2N samples out.
class Imdct {
public:
    // Imdct::new(n) (mdct.rs:27-29)
    Imdct(Context &ctx, std::size_t n) : Imdct(ctx, n, 1.0) {}
    // Imdct::new_scaled(n, scale) (mdct.rs:35-60); panics unless n is a power of two (mdct.rs:37-40)
    static Imdct new_scaled(Context &ctx, std::size_t n, double scale) { return Imdct(ctx, n, scale); }

    // imdct(&mut self, spec: &[f32], out: &mut [f32]) (mdct.rs:67-146); asserts spec.len() == n, out.len() == 2n
    void imdct(const float *spec, std::size_t spec_len, float *out, std::size_t out_len) {
        if (spec_len != n_ || out_len != 2 * n_) throw std::invalid_argument("Imdct::imdct: slice lengths (mdct.rs:76-78)");
        check(symaccel_imdct_f32(ctx_.raw(), (int)n_, scale_, spec, out, 1), ctx_.raw());
    }
    // `count` transforms back to back: spec[count][n] -> out[count][2n]
    void imdct_batch(const float *spec, float *out, std::size_t count) {
        check(symaccel_imdct_f32(ctx_.raw(), (int)n_, scale_, spec, out, count), ctx_.raw());
    }

private:
    Imdct(Context &ctx, std::size_t n, double scale) : ctx_(ctx), n_(n), scale_(scale) {
        if (n < 4 || (n & (n - 1)) != 0) throw std::invalid_argument("Imdct: n must be a power of two (mdct.rs:37)");
        if (n > 8192) throw Error(Error::Kind::Unsupported, SYMACCEL_ERR_UNSUPPORTED, "Imdct: n > 8192");
    }
    Context &ctx_;
    std::size_t n_;
    double scale_;
};

}  // namespace mdct

namespace fft {

using Complex = std::complex<float>;  // num_complex::Complex<f32>, same (re, im) layout

// Fft (no_simd.rs:70-141).  MAX_SIZE is the reference's (1 << 16: its permutation table is u16).
class Fft {
public:
    static constexpr std::size_t MAX_SIZE = 65536;
    Fft(Context &ctx, std::size_t n) : ctx_(ctx), n_(n) {
        if (n < 2 || (n & (n - 1)) != 0) throw std::invalid_argument("Fft: n must be a power of two (no_simd.rs:77)");
        if (n > MAX_SIZE) throw std::invalid_argument("Fft: n > MAX_SIZE (no_simd.rs:80)");
    }
    std::size_t size() const { return n_; }
    // fft_inplace(&mut self, x: &mut [Complex<f32>]) (no_simd.rs:96-118)
    void fft_inplace(Complex *x, std::size_t len) {
        if (len != n_) throw std::invalid_argument("Fft::fft_inplace: slice length (no_simd.rs:97)");
        check(symaccel_fft_c32(ctx_.raw(), (int)n_, reinterpret_cast<const float *>(x), reinterpret_cast<float *>(x), 1),
              ctx_.raw());
    }
    // fft(&mut self, x: &[Complex<f32>], y: &mut [Complex<f32>]) (no_simd.rs:121-140)
    void fft(const Complex *x, std::size_t x_len, Complex *y, std::size_t y_len) {
        if (x_len != n_ || y_len != n_) throw std::invalid_argument("Fft::fft: slice lengths (no_simd.rs:122-123)");
        check(symaccel_fft_c32(ctx_.raw(), (int)n_, reinterpret_cast<const float *>(x), reinterpret_cast<float *>(y), 1),
              ctx_.raw());
    }

private:
    Context &ctx_;
    std::size_t n_;
};

// Ifft (no_simd.rs:143-219): the forward transform between a re <-> im swap on the way in and a swap with the 1 / n scale
// on the way out.  (As in the reference, fewer than 32 points only permute, swap and scale: its transform() has no case
// for them.)
class Ifft {
public:
    static constexpr std::size_t MAX_SIZE = 65536;
    Ifft(Context &ctx, std::size_t n) : ctx_(ctx), n_(n) {
        if (n < 2 || (n & (n - 1)) != 0) throw std::invalid_argument("Ifft: n must be a power of two (no_simd.rs:152)");
        if (n > MAX_SIZE) throw std::invalid_argument("Ifft: n > MAX_SIZE (no_simd.rs:155)");
    }
    std::size_t size() const { return n_; }
    // ifft_inplace(&mut self, x: &mut [Complex<f32>]) (no_simd.rs:189-218)
    void ifft_inplace(Complex *x, std::size_t len) {
        if (len != n_) throw std::invalid_argument("Ifft::ifft_inplace: slice length (no_simd.rs:191)");
        check(symaccel_ifft_c32(ctx_.raw(), (int)n_, reinterpret_cast<const float *>(x), reinterpret_cast<float *>(x), 1), ctx_.raw());
    }
    // ifft(&mut self, x: &[Complex<f32>], y: &mut [Complex<f32>]) (no_simd.rs:166-187)
    void ifft(const Complex *x, std::size_t x_len, Complex *y, std::size_t y_len) {
        if (x_len != n_ || y_len != n_) throw std::invalid_argument("Ifft::ifft: slice lengths (no_simd.rs:168-169)");
        check(symaccel_ifft_c32(ctx_.raw(), (int)n_, reinterpret_cast<const float *>(x), reinterpret_cast<float *>(y), 1), ctx_.raw());
    }

private:
    Context &ctx_;
    std::size_t n_;
};

}  // namespace fft
}  // namespace dsp

namespace aac {

// window sequences (aac/common.rs:17-20)
constexpr std::uint8_t ONLY_LONG_SEQUENCE = 0, LONG_START_SEQUENCE = 1, EIGHT_SHORT_SEQUENCE = 2, LONG_STOP_SEQUENCE = 3;

// aac::dsp::Dsp (aac/dsp.rs:22-158).  The windows, the two Imdcts and the scratch buffers of the reference's
// struct live in the context (device tables / LDS).
class Dsp {
public:
    explicit Dsp(Context &ctx) : ctx_(ctx) {}
    // synth(&mut self, coeffs: &[f32; 1024], delay: &mut [f32; 1024], seq: u8, window_shape: bool,
    //       prev_window_shape: bool, dst: &mut [f32])  (aac/dsp.rs:57-65)
    void synth(const std::array<float, 1024> &coeffs, std::array<float, 1024> &delay, std::uint8_t seq, bool window_shape,
               bool prev_window_shape, float *dst, std::size_t dst_len) {
        if (dst_len < 1024) throw std::invalid_argument("Dsp::synth: dst shorter than 1024 samples");
        const std::uint8_t side = SYMACCEL_AAC_SIDE(seq, window_shape ? 1u : 0u, prev_window_shape ? 1u : 0u);
        check(symaccel_aac_synth(ctx_.raw(), coeffs.data(), &side, delay.data(), dst, 1, 1), ctx_.raw());
    }
    // chain-major batch: coeffs[chain][frame][1024], side[chain][frame], delay[chain][1024], pcm[chain][frame][1024]
    void synth_batch(const float *coeffs, const std::uint8_t *side, float *delay_io, float *pcm, std::size_t n_chains,
                     std::size_t frames_per_chain) {
        check(symaccel_aac_synth(ctx_.raw(), coeffs, side, delay_io, pcm, n_chains, frames_per_chain), ctx_.raw());
    }

private:
    Context &ctx_;
};

// The stages between the spectrum decoder and Dsp::synth, batched and in place on DEVICE memory (the array
// Dsp::synth_batch's device twin consumes): joint-stereo decoding of ChannelPair::decode (aac/cpe.rs:110-157) and the
// filtering loops of Tns::synth (aac/ics/tns.rs:180-195).  Descriptors: see symaccel.h.
inline void joint_stereo_device(Context &ctx, float *d_coeffs, std::size_t frames_per_chain, const std::int32_t *d_pair_chains,
                                const symaccel_aac_js_frame *d_desc, std::size_t n_pairs, const std::vector<std::uint16_t> &swb_long,
                                const std::vector<std::uint16_t> &swb_short) {
    if (swb_long.size() < 2 || swb_short.size() < 2) throw std::invalid_argument("joint_stereo: swb offset tables");
    check(symaccel_aac_joint_stereo_device(ctx.raw(), d_coeffs, frames_per_chain, d_pair_chains, d_desc, n_pairs, swb_long.data(),
                                           (int)swb_long.size() - 1, swb_short.data(), (int)swb_short.size() - 1),
          ctx.raw());
}
inline void tns_device(Context &ctx, float *d_coeffs, std::size_t n_frames, const symaccel_aac_tns_filter *d_filters,
                       std::size_t n_filters) {
    check(symaccel_aac_tns_device(ctx.raw(), d_coeffs, n_frames, d_filters, n_filters), ctx.raw());
}

}  // namespace aac

namespace mp3 {

// BlockType (layer3/common.rs:174-185)
enum class BlockType : std::uint8_t { Long = 0, Start = 1, Short = 2, End = 3 };

// The GranuleChannel fields the synthesis tail reads (layer3/common.rs:187-230)
struct GranuleChannel {
    BlockType block_type = BlockType::Long;
    bool is_mixed = false;   // BlockType::Short { is_mixed }
    std::uint16_t rzero = 576;
};

// SynthesisState (synthesis.rs:145-154) plus the hybrid-synthesis overlap of one channel (layer3/mod.rs:258)
struct SynthesisState {
    std::array<float, 32 * 18> overlap{};
    std::array<float, 16 * 64> v_vec{};
    std::int32_t v_front = 0;
};

// One channel of one granule: reorder, antialias, hybrid_synthesis, frequency_inversion (hybrid_synthesis.rs:153-485)
// and synthesis::synthesis with n_frames = 18 (synthesis.rs:158-336).  samples = post requantize + stereo.
inline void synthesize_granule(Context &ctx, int sample_rate_idx, const GranuleChannel &gc, const std::array<float, 576> &samples,
                               SynthesisState &state, float *out, std::size_t out_len) {
    if (out_len < 576) throw std::invalid_argument("synthesize_granule: out shorter than 576 samples");
    const symaccel_mp3_side side{(std::uint8_t)gc.block_type, (std::uint8_t)(gc.is_mixed ? 1 : 0), gc.rzero};
    check(symaccel_mp3_synth(ctx.raw(), samples.data(), &side, sample_rate_idx, state.overlap.data(), state.v_vec.data(),
                             &state.v_front, out, 1, 1),
          ctx.raw());
}
inline void synthesize_batch(Context &ctx, int sample_rate_idx, const float *xr, const symaccel_mp3_side *side, float *overlap_io,
                             float *v_vec_io, std::int32_t *v_front_io, float *pcm, std::size_t n_chains,
                             std::size_t granules_per_chain) {
    check(symaccel_mp3_synth(ctx.raw(), xr, side, sample_rate_idx, overlap_io, v_vec_io, v_front_io, pcm, n_chains,
                             granules_per_chain),
          ctx.raw());
}

// requantize(header, channel, buf) (requantize.rs:356-380) together with the sample mapping of read_huffman_samples
// (requantize.rs:117-147): `quant` holds the signed quantised Huffman samples, `buf` receives xr.
struct RequantizeChannel {  // the GranuleChannel fields requantize reads (layer3/common.rs:187-230)
    std::uint8_t global_gain = 210;
    bool scalefac_scale = false, preflag = false;
    BlockType block_type = BlockType::Long;
    bool is_mixed = false;
    std::array<std::uint8_t, 3> subblock_gain{};
    std::uint16_t rzero = 576;
    std::array<std::uint8_t, 39> scalefacs{};
};
inline symaccel_mp3_requant to_abi(const RequantizeChannel &c) {
    symaccel_mp3_requant d{};
    d.global_gain = c.global_gain;
    d.flags = (std::uint8_t)((c.scalefac_scale ? SYMACCEL_MP3_RQ_SCALEFAC_SCALE : 0u) | (c.preflag ? SYMACCEL_MP3_RQ_PREFLAG : 0u));
    d.block_type = (std::uint8_t)c.block_type;
    d.is_mixed = c.is_mixed ? 1 : 0;
    for (int w = 0; w < 3; ++w) d.subblock_gain[w] = c.subblock_gain[(std::size_t)w];
    d.rzero = c.rzero;
    for (int i = 0; i < 39; ++i) d.scalefacs[i] = c.scalefacs[(std::size_t)i];
    return d;
}
inline void requantize(Context &ctx, int sample_rate_idx, const RequantizeChannel &channel, const std::array<std::int16_t, 576> &quant,
                       std::array<float, 576> &buf) {
    const symaccel_mp3_requant d = to_abi(channel);
    check(symaccel_mp3_requantize(ctx.raw(), quant.data(), &d, sample_rate_idx, buf.data(), 1), ctx.raw());
}
inline void requantize_batch(Context &ctx, int sample_rate_idx, const std::int16_t *quant, const symaccel_mp3_requant *desc, float *xr,
                             std::size_t n_granule_channels) {
    check(symaccel_mp3_requantize(ctx.raw(), quant, desc, sample_rate_idx, xr, n_granule_channels), ctx.raw());
}

// stereo(header, granule, ch) (stereo.rs:485-556), batched and in place on DEVICE memory: xr[chain][granule][576]
inline void stereo_device(Context &ctx, int sample_rate_idx, float *d_xr, std::size_t granules_per_chain,
                          const std::int32_t *d_pair_chains, const symaccel_mp3_stereo *d_desc, std::size_t n_pairs) {
    check(symaccel_mp3_stereo_device(ctx.raw(), d_xr, granules_per_chain, d_pair_chains, d_desc, sample_rate_idx, n_pairs), ctx.raw());
}

// synthesis(state, n_frames, in_samples, out) (synthesis.rs:158-336) as Layer I (n_frames 12) and Layer II (36) call it
inline void synthesis(Context &ctx, SynthesisState &state, std::size_t n_frames, const float *in_samples, std::size_t in_len,
                      float *out, std::size_t out_len) {
    if (in_len != 32 * n_frames || out_len < 32 * n_frames) throw std::invalid_argument("synthesis: slice lengths (synthesis.rs:162)");
    check(symaccel_mpa_polyphase(ctx.raw(), (int)n_frames, in_samples, state.v_vec.data(), &state.v_front, out, 1, 1), ctx.raw());
}

}  // namespace mp3

namespace vorbis {

// Dsp + DspChannel (vorbis/dsp.rs:12-66): block sizes, and per channel the overlap of the previous block.
class Dsp {
public:
    // bs0_exp / bs1_exp from the identification header (vorbis/lib.rs:404-406, 461-470)
    Dsp(Context &ctx, int bs0_exp, int bs1_exp) : ctx_(ctx), bs0_exp_(bs0_exp), bs1_exp_(bs1_exp) {
        if (bs0_exp < 6 || bs1_exp > 13 || bs0_exp > bs1_exp) throw std::invalid_argument("vorbis::Dsp: block sizes");
    }
    int bs0_exp() const { return bs0_exp_; }
    int bs1_exp() const { return bs1_exp_; }
    Context &context() { return ctx_; }

private:
    Context &ctx_;
    int bs0_exp_, bs1_exp_;
};

class DspChannel {
public:
    explicit DspChannel(Dsp &dsp) : dsp_(dsp), overlap_((std::size_t)1 << (dsp.bs1_exp() - 1), 0.0f) {}
    // reset (dsp.rs:128-132)
    void reset() { overlap_.assign(overlap_.size(), 0.0f); }
    // synth(&mut self, block_flag, prev_block_flag (lib.rs:298: None pairs the first block with itself), ..., buf)
    // spectrum = floor x residue, bs/2 values; returns the (prev_n + n) / 4 samples of lib.rs:303 in `out`.
    std::size_t synth(bool block_flag, std::optional<bool> prev_block_flag, const float *spectrum, float *out, std::size_t out_cap) {
        const int n = 1 << (block_flag ? dsp_.bs1_exp() : dsp_.bs0_exp());
        const bool pf = prev_block_flag.value_or(block_flag);
        const int prev_n = 1 << (pf ? dsp_.bs1_exp() : dsp_.bs0_exp());
        const std::size_t n_out = (std::size_t)(prev_n + n) / 4;
        if (out_cap < n_out) throw std::invalid_argument("DspChannel::synth: out too short");
        std::uint8_t flag = block_flag ? 1 : 0;
        std::int32_t pflag = prev_block_flag ? (*prev_block_flag ? 1 : 0) : -1;
        check(symaccel_vorbis_synth(dsp_.context().raw(), dsp_.bs0_exp(), dsp_.bs1_exp(), spectrum, (std::size_t)n / 2, &flag,
                                    &pflag, overlap_.data(), out, n_out, 1, 1),
              dsp_.context().raw());
        return n_out;
    }
    const std::vector<float> &overlap() const { return overlap_; }

private:
    Dsp &dsp_;
    std::vector<float> overlap_;
};

}  // namespace vorbis

namespace flac {

// lpc_predict::<N>(order, coeffs, coeff_shift, buf) via the dispatch of decode_linear (decoder.rs:487-504, 716-752):
// coeffs in bitstream order (the first multiplies the most recent sample); buf = warm-up samples then residuals.
inline void lpc_predict(Context &ctx, std::size_t order, const std::int32_t *coeffs, std::uint32_t coeff_shift, std::int32_t *buf,
                        std::size_t len) {
    if (order < 1 || order > 32 || order > len) throw std::invalid_argument("lpc_predict: order (decoder.rs:456-461)");
    if (coeff_shift > 31) throw Error(Error::Kind::Unsupported, SYMACCEL_ERR_UNSUPPORTED, "lpc_predict: shift");
    std::int32_t c[32] = {0};
    for (std::size_t j = 0; j < order; ++j) c[j] = coeffs[j];
    const symaccel_flac_desc d{SYMACCEL_FLAC_LPC, (std::uint8_t)order, (std::uint8_t)coeff_shift, 0};
    check(symaccel_flac_restore(ctx.raw(), buf, &d, c, 1, len), ctx.raw());
}
// fixed_predict(order, buf) (decoder.rs:663-710)
inline void fixed_predict(Context &ctx, std::size_t order, std::int32_t *buf, std::size_t len) {
    if (order > 4) throw std::invalid_argument("fixed_predict: order > 4 (decoder.rs:664)");
    std::int32_t c[32] = {0};
    const symaccel_flac_desc d{SYMACCEL_FLAC_FIXED, (std::uint8_t)order, 0, 0};
    check(symaccel_flac_restore(ctx.raw(), buf, &d, c, 1, len), ctx.raw());
}
inline void restore_batch(Context &ctx, std::int32_t *buf, const symaccel_flac_desc *desc, const std::int32_t *coeffs,
                          std::size_t n_blocks, std::size_t blocksize) {
    check(symaccel_flac_restore(ctx.raw(), buf, desc, coeffs, n_blocks, blocksize), ctx.raw());
}
// A caller that owns a device batch plane (as the Batcher does for its FLAC / ALAC groups) gives its rows this pitch, in words: rows a power-of-two number of KiB
// apart -- 4096-sample blocks back to back -- alias on the HBM channels (symaccel_row_stride, include/symaccel.h; the reference has no counterpart: one Vec<i32>
// per channel, decoder.rs:199-242).  The device form over such a plane: the first `blocksize` words of a row are the subframe, the padding is left alone;
// d_pair_mode null = restore only, else restore + decorrelate + `<< out_shift` (decoder.rs:199-242 in one pass).
inline std::size_t row_stride(std::size_t blocksize) { return symaccel_row_stride(blocksize); }
inline void restore_strided_device(Context &ctx, std::int32_t *d_buf, const symaccel_flac_desc *d_desc, const std::int32_t *d_coeffs,
                                   const std::uint8_t *d_pair_mode, std::uint32_t out_shift, std::size_t n_blocks, std::size_t blocksize,
                                   std::size_t stride) {
    check(symaccel_flac_restore_strided_device(ctx.raw(), d_buf, d_desc, d_coeffs, d_pair_mode, out_shift, n_blocks, blocksize, stride), ctx.raw());
}
namespace detail {
inline void decorrelate(Context &ctx, std::uint8_t mode, std::int32_t *ch0, std::int32_t *ch1, std::size_t len) {
    check(symaccel_flac_decorrelate(ctx.raw(), &mode, ch0, ch1, 1, len, 0), ctx.raw());
}
}  // namespace detail
// decorrelate_left_side(left, side): side = left - side (decoder.rs:32-41)
inline void decorrelate_left_side(Context &ctx, const std::int32_t *left, std::int32_t *side, std::size_t len) {
    std::vector<std::int32_t> l(left, left + len);
    detail::decorrelate(ctx, 1, l.data(), side, len);
}
// decorrelate_mid_side(mid, side) in place (decoder.rs:43-70)
inline void decorrelate_mid_side(Context &ctx, std::int32_t *mid, std::int32_t *side, std::size_t len) {
    detail::decorrelate(ctx, 2, mid, side, len);
}
// decorrelate_right_side(right, side): side += right (decoder.rs:72-82)
inline void decorrelate_right_side(Context &ctx, const std::int32_t *right, std::int32_t *side, std::size_t len) {
    std::vector<std::int32_t> r(right, right + len);
    detail::decorrelate(ctx, 3, side, r.data(), len);
}

}  // namespace flac

namespace alac {

// ElementChannel (symphonia-codec-alac/src/lib.rs:71-80): the fields predict() reads.
struct ElementChannel {
    std::uint32_t bps = 16, mode = 0, shift = 0, lpc_order = 0;
    std::array<std::int32_t, 32> lpc_coeffs{};

    // predict(&mut self, out: &mut [i32]) -> Result<()> (lib.rs:165-264); decode_error on an invalid mode
    void predict(Context &ctx, std::int32_t *out, std::size_t len) const {
        if (mode > 0 && mode < 15) throw Error(Error::Kind::IoError, SYMACCEL_ERR_INVALID_ARG, "alac: invalid mode");
        if (lpc_order > 31 || bps == 0 || bps > 32) throw std::invalid_argument("alac::ElementChannel: field range");
        const symaccel_alac_desc d{(std::uint8_t)mode, (std::uint8_t)lpc_order, (std::uint8_t)shift, (std::uint8_t)bps};
        check(symaccel_alac_predict(ctx.raw(), out, &d, lpc_coeffs.data(), 1, len), ctx.raw());
    }
};
inline void predict_batch(Context &ctx, std::int32_t *buf, const symaccel_alac_desc *desc, const std::int32_t *coeffs,
                          std::size_t n_blocks, std::size_t blocksize) {
    check(symaccel_alac_predict(ctx.raw(), buf, desc, coeffs, n_blocks, blocksize), ctx.raw());
}
// predict over a device plane whose rows are `stride` words apart (flac::row_stride); d_pair_weight / d_pair_shift null = predict only, else with
// decorrelate_mid_side fused (lib.rs:541-560)
inline void predict_strided_device(Context &ctx, std::int32_t *d_buf, const symaccel_alac_desc *d_desc, const std::int32_t *d_coeffs,
                                   const std::int32_t *d_pair_weight, const std::uint8_t *d_pair_shift, std::size_t n_blocks,
                                   std::size_t blocksize, std::size_t stride) {
    check(symaccel_alac_predict_strided_device(ctx.raw(), d_buf, d_desc, d_coeffs, d_pair_weight, d_pair_shift, n_blocks, blocksize, stride), ctx.raw());
}
// decorrelate_mid_side(out0, out1, weight, shift) (lib.rs:664-671)
inline void decorrelate_mid_side(Context &ctx, std::int32_t *out0, std::int32_t *out1, std::size_t len, std::int32_t weight,
                                 std::uint8_t shift) {
    if (shift > 31) throw Error(Error::Kind::IoError, SYMACCEL_ERR_INVALID_ARG, "alac: mid_side_shift is greater than 31 bit");
    check(symaccel_alac_mid_side(ctx.raw(), &weight, &shift, out0, out1, 1, len), ctx.raw());
}

}  // namespace alac


// --------------------------------------------------------------------------------------------- codecs: the trait side
//
// symphonia_core::codecs::audio::AudioDecoder (symphonia-core/src/codecs/audio.rs:251-298):
//     fn reset(&mut self);  fn decode(&mut self, packet: &Packet) -> Result<GenericAudioBufferRef<'_>>;
//     fn finalize(&mut self) -> FinalizeResult;  fn last_decoded(&self) -> GenericAudioBufferRef<'_>;
// A GPU pays a launch and a PCIe round trip per call, so a drop-in decoder cannot transform one packet per decode():
// LookaheadDecoder keeps the trait's method set and its ownership rules (the decoder owns the buffer it returns a
// borrow of; the borrow is valid until the next &mut call; the buffer is cleared on error, audio.rs:278) and batches
// underneath: when decode(packet) finds nothing pre-computed for `packet`, it takes that packet plus up to K-1 packets
// the demuxer side can already see (the `peek` source: the shim wraps the FormatReader so that it reads ahead), runs ONE
// batch call over all of them -- every channel is a chain, the K packets are K consecutive frames of each chain, the
// state (delay lines, overlap, V FIFO) enters and leaves through the *_io buffers -- and hands the frames back one per
// call.  A packet that is not the one expected next (the caller seeked or dropped packets) invalidates the look-ahead.
// `Codec` supplies the packet type and the batch call (AacLc, Mp3 below); the entropy decode that produces the packets'
// spectra stays in the reference's CPU code.
namespace codecs {

template <class S>
struct AudioBufferRefT {  // GenericAudioBufferRef::F32 / ::S32 (audio/generic.rs:381-401): planar, one slice per channel
    std::vector<const S *> planes;
    std::size_t frames = 0;
    bool is_empty() const { return frames == 0; }
};
using AudioBufferRef = AudioBufferRefT<float>;          // the transform codecs
using AudioBufferRefS32 = AudioBufferRefT<std::int32_t>;  // FLAC (AudioBuffer<i32>, flac/decoder.rs:103)

struct FinalizeResult {  // codecs/audio.rs:230-236
    std::optional<bool> verify_ok;
};

// With a `Batcher` (second constructor) the same decoder coalesces ACROSS streams: its batches are submitted to the process-wide
// batcher instead of being transformed by a call of their own, and the next batch is submitted early -- as soon as half of the
// current one has been handed out -- so that by the time this stream needs it, the decoders of the other streams have submitted
// theirs and one launch (one PCIe round trip) serves all of them.  The returned planes then point into the batcher's page-locked
// result slot (valid, like every buffer of the trait, until the next &mut call).
template <class Codec>
class LookaheadDecoder {
public:
    using Packet = typename Codec::Packet;
    using Sample = typename Codec::Sample;                // f32 for the transform codecs, i32 for FLAC
    using Buffer = AudioBufferRefT<Sample>;
    using Peek = std::function<std::optional<Packet>()>;  // the next packet of the same track the demuxer can see, if any

    LookaheadDecoder(Context &ctx, const typename Codec::Params &params, std::size_t lookahead, Peek peek)
        : ctx_(ctx), codec_(params), lookahead_(lookahead < 1 ? 1 : lookahead), peek_(std::move(peek)) {
        reset();
    }
    LookaheadDecoder(Batcher &batcher, const typename Codec::Params &params, std::size_t lookahead, Peek peek)
        : ctx_(batcher.context()), codec_(params), lookahead_(lookahead < 1 ? 1 : lookahead), peek_(std::move(peek)),
          batcher_(Codec::kBatchKind ? &batcher : nullptr) {
        if (batcher_) codec_.attach(batcher);  // (what a codec registers with the batcher once: AacLcCoded's band tables)
        reset();
    }
    // Zero-copy parse: a front end that can parse the demuxer's next packet STRAIGHT INTO the batcher's page-locked slot.  The batch
    // length is fixed first -- `avail()` = the packets of this track the demuxer can see now --, the slot is reserved for it, and
    // `parse_into(view, i)` parses the next packet into position i of the batch (`view`: the codec's BatchView, the destination of
    // every plane in the chain-major layout) and returns its id; nullopt = no packet after all (the batch is re-packed shorter).
    // Nothing is copied between the parser's output and the DMA source.  For codecs with `kDirect` (fixed-size units: AacLc, Mp3,
    // Mp3Huffman); the Rust twin: `BatchCodec::reserve` / `commit` (bindings/rust/symphonia-accel-hip/src/lookahead.rs).
    struct Direct {
        std::function<std::size_t()> avail;
        std::function<std::optional<std::uint64_t>(const typename Codec::BatchView &, std::size_t)> parse_into;
    };
    LookaheadDecoder(Batcher &batcher, const typename Codec::Params &params, std::size_t lookahead, Direct direct)
        : ctx_(batcher.context()), codec_(params), lookahead_(lookahead < 1 ? 1 : lookahead), batcher_(&batcher), direct_(std::move(direct)) {
        static_assert(Codec::kDirect, "this codec's batch layout depends on the packets' content: use the Peek form");
        if (!direct_.avail || !direct_.parse_into) throw std::invalid_argument("LookaheadDecoder: direct source");
        codec_.attach(batcher);
        reset();
    }
    ~LookaheadDecoder() { drop_tickets(); }
    LookaheadDecoder(const LookaheadDecoder &) = delete;
    LookaheadDecoder &operator=(const LookaheadDecoder &) = delete;

    // AudioDecoder::reset (audio.rs:252-257): "must be called after a seek"; state as after construction
    void reset() {
        drop_tickets();
        codec_.reset_state();
        ready_.clear();
        head_ = 0;
        have_last_packet_ = false;
        clear_last();
    }

    // AudioDecoder::decode (audio.rs:259-281)
    const Buffer &decode(const Packet &packet) {
        if (head_ < ready_.size() && ready_[head_] != Codec::id(packet)) {
            // Not the packet the look-ahead was computed for: the caller skipped packets without reset().  A
            // frame-by-frame decoder would now continue from the state the LAST RETURNED packet left, but the carried
            // state here is already that of the end of the batch.  Every codec on this path has a one-packet memory
            // (the delay line / overlap / V FIFO after a packet depend on that packet's input alone -- the same property
            // the kernels' segment halo uses), so replaying the last returned packet rebuilds exactly that state.
            drop_lookahead();
            if (have_last_packet_) replay_last();
        }
        if (head_ >= ready_.size()) {
            try {
                if (next_live_ && !next_ids_.empty() && next_ids_[0] == Codec::id(packet)) {
                    take_next();  // the batch submitted ahead starts with this packet
                } else {
                    // (a batch submitted ahead for packets the caller then skipped is dropped unseen: the carried state is still
                    // the one the last returned packet left, because the current batch was handed out to its end)
                    if (next_live_) drop_lookahead();
                    fill(packet);
                }
            } catch (...) {
                clear_last();  // audio.rs:278: "implementors of this function must clear the internal buffer if an error occurs"
                drop_tickets();
                ready_.clear();
                head_ = 0;
                throw;
            }
        }
        publish(head_++);
        last_packet_ = packet;
        have_last_packet_ = true;
        if (batcher_) {
            // half of the batch handed out: the next one is submitted (the other streams' decoders do the same around now); a
            // quarter left: whatever is pending goes to the device, so that the copies and the kernels run while the rest is consumed
            const std::size_t left = ready_.size() - head_;
            if (!next_live_ && 2 * left <= ready_.size()) submit_ahead();
            if (next_live_ && !hinted_ && 4 * left <= ready_.size()) {
                hinted_ = true;
                batcher_->hint();
            }
        }
        return last_;
    }

    FinalizeResult finalize() { return FinalizeResult{}; }  // (the f32 codecs verify nothing, like the reference's)
    const Buffer &last_decoded() const { return last_; }
    std::size_t lookahead() const { return lookahead_; }
    std::size_t batches_run() const { return batches_; }

private:
    std::vector<Packet> gather(const Packet *first) {
        std::vector<Packet> batch;
        if (first) batch.push_back(*first);
        // packets pulled earlier whose frames were dropped by a discontinuity are not replayed: the demuxer moved on
        while (batch.size() < lookahead_) {
            std::optional<Packet> nxt = peek_ ? peek_() : std::nullopt;
            if (!nxt) break;
            batch.push_back(std::move(*nxt));
        }
        return batch;
    }
    void fill(const Packet &first) {
        if (direct_.avail) {
            if constexpr (Codec::kDirect) {
                submit_direct(&first);
                take_next();
                return;
            }
        }
        std::vector<Packet> batch = gather(&first);
        if (batcher_) {
            if constexpr (Codec::kBatchKind != 0) {
                submit(batch);
                take_next();
                return;
            }
        }
        codec_.decode_batch(ctx_, batch, pcm_);
        pcm_base_ = pcm_.data();
        ++batches_;
        ready_.clear();
        for (const Packet &p : batch) ready_.push_back(Codec::id(p));
        batch_len_ = batch.size();
        head_ = 0;
    }
    // ---- batcher mode
    void submit(const std::vector<Packet> &batch) {
        if constexpr (Codec::kBatchKind != 0) {
            symaccel_batch_slot slot;
            check(symaccel_batcher_reserve(batcher_->raw(), Codec::kBatchKind, codec_.batch_param(), codec_.batch_chains(batch.size()),
                                           codec_.batch_units(batch.size()), &slot, &next_ticket_),
                  ctx_.raw());
            next_live_ = true;
            try {
                codec_.fill_slot(batch, slot);  // the parsed packets and the carried state go into the page-locked slot
            } catch (...) {
                symaccel_batcher_release(batcher_->raw(), next_ticket_);
                next_live_ = false;
                throw;
            }
            check(symaccel_batcher_commit(batcher_->raw(), next_ticket_), ctx_.raw());
            next_ids_.clear();
            for (const Packet &p : batch) next_ids_.push_back(Codec::id(p));
            hinted_ = false;
        }
    }
    void replay_last() {
        std::vector<Packet> one(1, last_packet_);
        if (batcher_) {
            if constexpr (Codec::kBatchKind != 0) {  // (through the batcher: the context is not ours to call while a batcher drives it)
                submit(one);
                symaccel_batch_slot slot;
                const int st = symaccel_batcher_wait(batcher_->raw(), next_ticket_, &slot);
                if (st == SYMACCEL_OK) codec_.take_state(slot);
                symaccel_batcher_release(batcher_->raw(), next_ticket_);
                next_live_ = false;
                next_ids_.clear();
                check(st, ctx_.raw());
                return;
            }
        }
        std::vector<Sample> scratch;
        codec_.decode_batch(ctx_, one, scratch);
    }
    void submit_ahead() {
        // the state the next batch starts from is the one the current batch left: known since take_next() copied it out
        if (direct_.avail) {
            if constexpr (Codec::kDirect) {
                submit_direct(nullptr);
                return;
            }
        }
        std::vector<Packet> batch = gather(nullptr);
        if (!batch.empty()) submit(batch);
    }
    // the direct form of submit(): reserve for the packets the demuxer can see, parse them into the slot, commit
    void submit_direct(const Packet *first) {
        if constexpr (Codec::kDirect) {
            const std::size_t head = first ? 1 : 0;
            std::size_t k = head + std::min(lookahead_ - head, direct_.avail());
            if (k == 0) return;
            symaccel_batch_slot slot;
            auto reserve = [&](std::size_t n, symaccel_batch_slot *sl, std::uint64_t *t) {
                check(symaccel_batcher_reserve(batcher_->raw(), Codec::kBatchKind, codec_.batch_param(), codec_.batch_chains(n), codec_.batch_units(n), sl, t),
                      ctx_.raw());
            };
            reserve(k, &slot, &next_ticket_);
            next_live_ = true;
            next_ids_.clear();
            try {
                typename Codec::BatchView view = codec_.view(slot, k);
                std::size_t j = 0;
                if (first) {
                    codec_.put(view, 0, *first);
                    next_ids_.push_back(Codec::id(*first));
                    j = 1;
                }
                for (; j < k; ++j) {
                    const std::optional<std::uint64_t> id = direct_.parse_into(view, j);
                    if (!id) break;
                    next_ids_.push_back(*id);
                }
                if (j == 0) {  // nothing there after all
                    symaccel_batcher_release(batcher_->raw(), next_ticket_);
                    next_live_ = false;
                    return;
                }
                if (j < k) {
                    // fewer packets than the demuxer announced (a corrupt one ends the batch): the rows move to a slot of the right
                    // shape -- units are part of the layout AND of the state the batch leaves -- and the long one goes back unused
                    symaccel_batch_slot shorter;
                    std::uint64_t t2 = 0;
                    reserve(j, &shorter, &t2);
                    typename Codec::BatchView to = codec_.view(shorter, j);
                    for (std::size_t i = 0; i < j; ++i) codec_.put(to, i, codec_.extract(view, i));
                    symaccel_batcher_release(batcher_->raw(), next_ticket_);
                    next_ticket_ = t2;
                    slot = shorter;
                }
                codec_.put_state(slot);
            } catch (...) {
                symaccel_batcher_release(batcher_->raw(), next_ticket_);
                next_live_ = false;
                throw;
            }
            check(symaccel_batcher_commit(batcher_->raw(), next_ticket_), ctx_.raw());
            hinted_ = false;
        }
    }
    void take_next() {
        if constexpr (Codec::kBatchKind != 0) {
            symaccel_batch_slot slot;
            const int st = symaccel_batcher_wait(batcher_->raw(), next_ticket_, &slot);
            if (st != SYMACCEL_OK) {
                symaccel_batcher_release(batcher_->raw(), next_ticket_);
                next_live_ = false;
                check(st, ctx_.raw());
            }
            if (cur_live_) symaccel_batcher_release(batcher_->raw(), cur_ticket_);
            cur_ticket_ = next_ticket_;
            cur_live_ = true;
            next_live_ = false;
            codec_.take_state(slot);
            pcm_base_ = static_cast<const Sample *>(slot.out);
            ready_.swap(next_ids_);
            next_ids_.clear();
            batch_len_ = ready_.size();
            head_ = 0;
            ++batches_;
        }
    }
    void drop_tickets() {
        if (!batcher_) return;
        if (next_live_) {
            symaccel_batcher_wait(batcher_->raw(), next_ticket_, nullptr);  // (its result is not wanted, its slot must be idle)
            symaccel_batcher_release(batcher_->raw(), next_ticket_);
            next_live_ = false;
        }
        if (cur_live_) {
            symaccel_batcher_release(batcher_->raw(), cur_ticket_);
            cur_live_ = false;
        }
        next_ids_.clear();
        pcm_base_ = nullptr;
    }
    void drop_lookahead() {
        drop_tickets();
        clear_last();
        ready_.clear();
        head_ = 0;
    }
    void publish(std::size_t i) {
        // where packet i of the last batch lies in a channel's plane is the codec's business: fixed-size frames are
        // [channel][packet][frames]; Vorbis packs (prev_n + n) / 4 samples per packet and none for the first one
        const std::size_t nch = codec_.channels();
        last_.planes.resize(nch);
        for (std::size_t c = 0; c < nch; ++c) last_.planes[c] = pcm_base_ + codec_.plane_offset(c, i, batch_len_);
        last_.frames = codec_.packet_frames(i);
    }
    void clear_last() {
        last_.planes.assign(codec_.channels(), nullptr);
        last_.frames = 0;
    }

    Context &ctx_;
    Codec codec_;
    std::size_t lookahead_;
    Peek peek_;
    Batcher *batcher_ = nullptr;
    Direct direct_{};                        // (empty: the Peek form)
    std::vector<Sample> pcm_;                // [channel][packet of the batch][frames_per_packet]: planar per packet
    const Sample *pcm_base_ = nullptr;       // pcm_.data(), or the batcher's result slot of the current batch
    std::vector<std::uint64_t> ready_;       // ids of the batch's packets, in order
    std::vector<std::uint64_t> next_ids_;    // ... of the batch submitted ahead
    std::uint64_t cur_ticket_ = 0, next_ticket_ = 0;
    bool cur_live_ = false, next_live_ = false, hinted_ = false;
    Packet last_packet_{};                   // the packet decode() returned last (replayed after a discontinuity)
    bool have_last_packet_ = false;
    std::size_t head_ = 0, batch_len_ = 0, batches_ = 0;
    Buffer last_;
};

// AAC-LC: one packet = one raw_data_block = 1024 frames per channel.  What the CPU side (reference parser + spectral
// tools) hands over per channel: the dequantised coefficients Ics::synth_channel would give Dsp::synth, the window
// sequence and the window shapes (ics/mod.rs:449-468).
struct AacLc {
    using Sample = float;
    struct Params {
        std::size_t channels = 2;
    };
    struct Packet {
        std::uint64_t ts = 0;                // Packet::ts: identifies the packet
        std::vector<float> coeffs;           // [channel][1024]
        std::vector<std::uint8_t> side;      // [channel]: SYMACCEL_AAC_SIDE(seq, shape, prev_shape)
    };
    explicit AacLc(const Params &p) : nch_(p.channels), delay_(p.channels * 1024, 0.0f) {}
    static std::uint64_t id(const Packet &p) { return p.ts; }
    std::size_t channels() const { return nch_; }
    std::size_t frames_per_packet() const { return 1024; }
    std::size_t packet_frames(std::size_t) const { return 1024; }
    std::size_t plane_offset(std::size_t c, std::size_t i, std::size_t k) const { return (c * k + i) * 1024; }
    void reset_state() { std::fill(delay_.begin(), delay_.end(), 0.0f); }  // AacDecoder::reset: delay lines zeroed
    void decode_batch(Context &ctx, const std::vector<Packet> &batch, std::vector<float> &pcm) {
        const std::size_t k = batch.size();
        in_.resize(nch_ * k * 1024);
        side_.resize(nch_ * k);
        pcm.resize(nch_ * k * 1024);
        gather(batch, in_.data(), side_.data());
        check(symaccel_aac_synth(ctx.raw(), in_.data(), side_.data(), delay_.data(), pcm.data(), nch_, k), ctx.raw());
    }
    // the cross-stream batcher's view of the same batch (LookaheadDecoder's second constructor)
    static constexpr int kBatchKind = SYMACCEL_BATCH_AAC_SYNTH;
    static constexpr bool kDirect = true;
    void attach(Batcher &) {}
    int batch_param() const { return 0; }
    std::size_t units_per_packet() const { return 1; }
    std::size_t batch_chains(std::size_t) const { return nch_; }  // a submission of k packets: nch chains of k units
    std::size_t batch_units(std::size_t k) const { return k; }
    // where packet i of a batch of k lies in a slot's planes (what a parser that writes straight into the slot needs: LookaheadDecoder::Direct)
    struct BatchView {
        float *coeffs = nullptr;
        std::uint8_t *side = nullptr;
        std::size_t k = 0, nch = 0;
        float *coeffs_at(std::size_t c, std::size_t i) const { return coeffs + (c * k + i) * 1024; }
        std::uint8_t &side_at(std::size_t c, std::size_t i) const { return side[c * k + i]; }
    };
    BatchView view(const symaccel_batch_slot &slot, std::size_t k) const {
        return BatchView{static_cast<float *>(slot.input[0]), static_cast<std::uint8_t *>(slot.input[1]), k, nch_};
    }
    void put(const BatchView &v, std::size_t i, const Packet &p) const {
        if (p.coeffs.size() != nch_ * 1024 || p.side.size() != nch_) throw std::invalid_argument("AacLc: packet shape");
        for (std::size_t c = 0; c < nch_; ++c) {
            std::copy_n(p.coeffs.data() + c * 1024, 1024, v.coeffs_at(c, i));
            v.side_at(c, i) = p.side[c];
        }
    }
    Packet extract(const BatchView &v, std::size_t i) const {
        Packet p;
        p.coeffs.resize(nch_ * 1024);
        p.side.resize(nch_);
        for (std::size_t c = 0; c < nch_; ++c) {
            std::copy_n(v.coeffs_at(c, i), 1024, p.coeffs.data() + c * 1024);
            p.side[c] = v.side_at(c, i);
        }
        return p;
    }
    void put_state(const symaccel_batch_slot &slot) const { std::memcpy(slot.state[0], delay_.data(), delay_.size() * sizeof(float)); }
    void fill_slot(const std::vector<Packet> &batch, const symaccel_batch_slot &slot) {
        const BatchView v = view(slot, batch.size());
        for (std::size_t i = 0; i < batch.size(); ++i) put(v, i, batch[i]);
        put_state(slot);
    }
    void take_state(const symaccel_batch_slot &slot) { std::memcpy(delay_.data(), slot.state[0], delay_.size() * sizeof(float)); }

private:
    void gather(const std::vector<Packet> &batch, float *in, std::uint8_t *side) const {
        const std::size_t k = batch.size();
        for (std::size_t i = 0; i < k; ++i) {
            if (batch[i].coeffs.size() != nch_ * 1024 || batch[i].side.size() != nch_) throw std::invalid_argument("AacLc: packet shape");
            for (std::size_t c = 0; c < nch_; ++c) {
                std::copy_n(batch[i].coeffs.data() + c * 1024, 1024, in + (c * k + i) * 1024);
                side[c * k + i] = batch[i].side[c];
            }
        }
    }
    std::size_t nch_;
    std::vector<float> delay_, in_;
    std::vector<std::uint8_t> side_;
};

// AAC-LC one stage earlier: what the spectrum decoder produces (cpe.rs:110-157 and ics/mod.rs:449-468 still to come) -- the
// coefficients as decoded (mid / side or intensity coded where the descriptors say so), per jointly coded channel pair its
// descriptor, and the TNS filters of the packet; joint stereo, TNS and Dsp::synth run on the device (symaccel_aac_decode_pipelined,
// or SYMACCEL_BATCH_AAC_DECODE through the batcher: the pair frames with TNS take a list pass, the filters run, one walk synthesises).
struct AacLcCoded {
    using Sample = float;
    struct Params {
        std::size_t channels = 2;
        std::vector<std::uint16_t> swb_long, swb_short;  // the stream's scale-factor-band offsets (n + 1 values each)
    };
    struct Packet {
        std::uint64_t ts = 0;
        std::vector<float> coeffs;                                       // [channel][1024]
        std::vector<std::uint8_t> side;                                  // [channel]
        std::vector<std::pair<std::size_t, symaccel_aac_js_frame>> joint;  // (left channel of the pair, its descriptor)
        std::vector<symaccel_aac_tns_filter> tns;                        // frame = channel
    };
    explicit AacLcCoded(const Params &p) : nch_(p.channels), swb_long_(p.swb_long), swb_short_(p.swb_short), delay_(p.channels * 1024, 0.0f) {
        if (swb_long_.size() < 2 || swb_short_.size() < 2) throw std::invalid_argument("AacLcCoded: band tables");
    }
    static std::uint64_t id(const Packet &p) { return p.ts; }
    std::size_t channels() const { return nch_; }
    std::size_t packet_frames(std::size_t) const { return 1024; }
    std::size_t plane_offset(std::size_t c, std::size_t i, std::size_t k) const { return (c * k + i) * 1024; }
    void reset_state() { std::fill(delay_.begin(), delay_.end(), 0.0f); }
    void decode_batch(Context &ctx, const std::vector<Packet> &batch, std::vector<float> &pcm) {
        const std::size_t k = batch.size();
        in_.resize(nch_ * k * 1024);
        side_.resize(nch_ * k);
        pcm.resize(nch_ * k * 1024);
        describe(batch, in_.data(), side_.data());
        const std::size_t n_pairs = pairs_.size() / 2;
        check(symaccel_aac_decode_pipelined(ctx.raw(), in_.data(), side_.data(), n_pairs ? pairs_.data() : nullptr, n_pairs ? js_.data() : nullptr,
                                            n_pairs, swb_long_.data(), (int)swb_long_.size() - 1, swb_short_.data(), (int)swb_short_.size() - 1,
                                            tns_.empty() ? nullptr : tns_.data(), tns_.size(), delay_.data(), pcm.data(), nch_, k, 0),
              ctx.raw());
    }
    static constexpr int kBatchKind = SYMACCEL_BATCH_AAC_DECODE;
    static constexpr bool kDirect = false;  // (the descriptor blob is assembled from the whole batch)
    struct BatchView {};
    void attach(Batcher &b) {
        check(symaccel_batcher_aac_bands(b.raw(), swb_long_.data(), (int)swb_long_.size() - 1, swb_short_.data(), (int)swb_short_.size() - 1, &bands_),
              b.context().raw());
    }
    int batch_param() const { return bands_; }
    std::size_t units_per_packet() const { return 1; }
    std::size_t batch_chains(std::size_t) const { return nch_; }  // a submission of k packets: nch chains of k units
    std::size_t batch_units(std::size_t k) const { return k; }
    void fill_slot(const std::vector<Packet> &batch, const symaccel_batch_slot &slot) {
        const std::size_t k = batch.size();
        describe(batch, static_cast<float *>(slot.input[0]), static_cast<std::uint8_t *>(slot.input[1]));
        // the descriptor blob (include/symaccel.h, SYMACCEL_BATCH_AAC_DECODE): header, pair list, js rows, filters, 16-byte aligned parts
        const std::size_t n_pairs = pairs_.size() / 2;
        const std::size_t off_js = 16 + ((n_pairs * 8 + 15) & ~(std::size_t)15);
        const std::size_t off_tns = off_js + ((n_pairs * k * sizeof(symaccel_aac_js_frame) + 15) & ~(std::size_t)15);
        if (off_tns + tns_.size() * sizeof(symaccel_aac_tns_filter) > slot.input_bytes[2]) throw std::invalid_argument("AacLcCoded: more than 8 TNS filters per channel-frame");
        char *blob = static_cast<char *>(slot.input[2]);
        const std::uint32_t header[4] = {(std::uint32_t)n_pairs, (std::uint32_t)tns_.size(), 0u, 0u};
        std::memcpy(blob, header, 16);
        if (n_pairs) {
            std::memcpy(blob + 16, pairs_.data(), n_pairs * 8);
            std::memcpy(blob + off_js, js_.data(), n_pairs * k * sizeof(symaccel_aac_js_frame));
        }
        if (!tns_.empty()) std::memcpy(blob + off_tns, tns_.data(), tns_.size() * sizeof(symaccel_aac_tns_filter));
        std::memcpy(slot.state[0], delay_.data(), delay_.size() * sizeof(float));
    }
    void take_state(const symaccel_batch_slot &slot) { std::memcpy(delay_.data(), slot.state[0], delay_.size() * sizeof(float)); }

private:
    // the batch in the entry points' terms: chain-major spectra + side bytes, the pairs coded jointly anywhere in the batch, their
    // descriptors [pair][packet] (a packet in which a pair is not jointly coded gets the empty descriptor), the filters re-based
    void describe(const std::vector<Packet> &batch, float *in, std::uint8_t *side) {
        const std::size_t k = batch.size();
        std::vector<std::size_t> lefts;
        for (std::size_t i = 0; i < k; ++i) {
            const Packet &p = batch[i];
            if (p.coeffs.size() != nch_ * 1024 || p.side.size() != nch_) throw std::invalid_argument("AacLcCoded: packet shape");
            for (std::size_t c = 0; c < nch_; ++c) {
                std::copy_n(p.coeffs.data() + c * 1024, 1024, in + (c * k + i) * 1024);
                side[c * k + i] = p.side[c];
            }
            for (const auto &j : p.joint)
                if (j.first + 1 < nch_ && std::find(lefts.begin(), lefts.end(), j.first) == lefts.end()) lefts.push_back(j.first);
        }
        std::sort(lefts.begin(), lefts.end());
        pairs_.clear();
        for (std::size_t l : lefts) {
            pairs_.push_back((std::int32_t)l);
            pairs_.push_back((std::int32_t)l + 1);
        }
        symaccel_aac_js_frame none{};
        none.num_windows = 1;
        js_.assign(lefts.size() * k, none);
        tns_.clear();
        for (std::size_t i = 0; i < k; ++i) {
            for (const auto &j : batch[i].joint) {
                const auto it = std::find(lefts.begin(), lefts.end(), j.first);
                if (it != lefts.end()) js_[(std::size_t)(it - lefts.begin()) * k + i] = j.second;
            }
            for (symaccel_aac_tns_filter f : batch[i].tns) {
                f.frame = (std::uint32_t)((std::size_t)f.frame * k + i);  // [channel][packet of the batch]
                tns_.push_back(f);
            }
        }
    }
    std::size_t nch_;
    std::vector<std::uint16_t> swb_long_, swb_short_;
    std::vector<float> delay_, in_;
    std::vector<std::uint8_t> side_;
    std::vector<std::int32_t> pairs_;
    std::vector<symaccel_aac_js_frame> js_;
    std::vector<symaccel_aac_tns_filter> tns_;
    int bands_ = -1;
};

// MPEG-1/2 Layer III: one packet = one frame = `granules` granules of 576 frames per channel (2 for MPEG-1, 1 for
// MPEG-2 / 2.5, common.rs:173-178).  Per granule-channel: the requantised, stereo-processed samples and the side fields
// the synthesis tail reads (layer3/mod.rs:440-476).
struct Mp3 {
    using Sample = float;
    struct Params {
        std::size_t channels = 2;
        std::size_t granules = 2;
        int sample_rate_idx = 0;
    };
    struct Packet {
        std::uint64_t ts = 0;
        std::vector<float> xr;                 // [granule][channel][576]
        std::vector<symaccel_mp3_side> side;   // [granule][channel]
    };
    explicit Mp3(const Params &p)
        : nch_(p.channels), ngr_(p.granules), sr_(p.sample_rate_idx), overlap_(p.channels * 576, 0.0f), vvec_(p.channels * 1024, 0.0f),
          vfront_(p.channels, 0) {}
    static std::uint64_t id(const Packet &p) { return p.ts; }
    std::size_t channels() const { return nch_; }
    std::size_t frames_per_packet() const { return 576 * ngr_; }
    std::size_t packet_frames(std::size_t) const { return 576 * ngr_; }
    std::size_t plane_offset(std::size_t c, std::size_t i, std::size_t k) const { return (c * k + i) * 576 * ngr_; }
    void reset_state() {  // MpaDecoder::reset: fresh State (overlap and SynthesisState::default())
        std::fill(overlap_.begin(), overlap_.end(), 0.0f);
        std::fill(vvec_.begin(), vvec_.end(), 0.0f);
        std::fill(vfront_.begin(), vfront_.end(), 0);
    }
    void decode_batch(Context &ctx, const std::vector<Packet> &batch, std::vector<float> &pcm) {
        const std::size_t k = batch.size(), g = k * ngr_;
        in_.resize(nch_ * g * 576);
        side_.resize(nch_ * g);
        pcm.resize(nch_ * g * 576);
        gather(batch, in_.data(), side_.data());
        check(symaccel_mp3_synth(ctx.raw(), in_.data(), side_.data(), sr_, overlap_.data(), vvec_.data(), vfront_.data(), pcm.data(), nch_, g),
              ctx.raw());
    }
    static constexpr int kBatchKind = SYMACCEL_BATCH_MP3_SYNTH;
    static constexpr bool kDirect = true;
    void attach(Batcher &) {}
    int batch_param() const { return sr_; }
    std::size_t units_per_packet() const { return ngr_; }
    std::size_t batch_chains(std::size_t) const { return nch_; }
    std::size_t batch_units(std::size_t k) const { return k * ngr_; }
    struct BatchView {  // granule gr of packet i of channel c is unit i * ngr + gr of chain c
        float *xr = nullptr;
        symaccel_mp3_side *side = nullptr;
        std::size_t k = 0, nch = 0, ngr = 0;
        std::size_t unit(std::size_t c, std::size_t i, std::size_t gr) const { return c * k * ngr + i * ngr + gr; }
    };
    BatchView view(const symaccel_batch_slot &slot, std::size_t k) const {
        return BatchView{static_cast<float *>(slot.input[0]), static_cast<symaccel_mp3_side *>(slot.input[1]), k, nch_, ngr_};
    }
    void put(const BatchView &v, std::size_t i, const Packet &p) const {
        if (p.xr.size() != ngr_ * nch_ * 576 || p.side.size() != ngr_ * nch_) throw std::invalid_argument("Mp3: packet shape");
        for (std::size_t gr = 0; gr < ngr_; ++gr)
            for (std::size_t c = 0; c < nch_; ++c) {
                std::copy_n(p.xr.data() + (gr * nch_ + c) * 576, 576, v.xr + v.unit(c, i, gr) * 576);
                v.side[v.unit(c, i, gr)] = p.side[gr * nch_ + c];
            }
    }
    Packet extract(const BatchView &v, std::size_t i) const {
        Packet p;
        p.xr.resize(ngr_ * nch_ * 576);
        p.side.resize(ngr_ * nch_);
        for (std::size_t gr = 0; gr < ngr_; ++gr)
            for (std::size_t c = 0; c < nch_; ++c) {
                std::copy_n(v.xr + v.unit(c, i, gr) * 576, 576, p.xr.data() + (gr * nch_ + c) * 576);
                p.side[gr * nch_ + c] = v.side[v.unit(c, i, gr)];
            }
        return p;
    }
    void put_state(const symaccel_batch_slot &slot) const {
        std::memcpy(slot.state[0], overlap_.data(), overlap_.size() * sizeof(float));
        std::memcpy(slot.state[1], vvec_.data(), vvec_.size() * sizeof(float));
        std::memcpy(slot.state[2], vfront_.data(), vfront_.size() * sizeof(std::int32_t));
    }
    void fill_slot(const std::vector<Packet> &batch, const symaccel_batch_slot &slot) {
        const BatchView v = view(slot, batch.size());
        for (std::size_t i = 0; i < batch.size(); ++i) put(v, i, batch[i]);
        put_state(slot);
    }
    void take_state(const symaccel_batch_slot &slot) {
        std::memcpy(overlap_.data(), slot.state[0], overlap_.size() * sizeof(float));
        std::memcpy(vvec_.data(), slot.state[1], vvec_.size() * sizeof(float));
        std::memcpy(vfront_.data(), slot.state[2], vfront_.size() * sizeof(std::int32_t));
    }

private:
    void gather(const std::vector<Packet> &batch, float *in, symaccel_mp3_side *side) const {
        const std::size_t k = batch.size(), g = k * ngr_;
        for (std::size_t i = 0; i < k; ++i) {
            if (batch[i].xr.size() != ngr_ * nch_ * 576 || batch[i].side.size() != ngr_ * nch_) throw std::invalid_argument("Mp3: packet shape");
            for (std::size_t gr = 0; gr < ngr_; ++gr)
                for (std::size_t c = 0; c < nch_; ++c) {
                    std::copy_n(batch[i].xr.data() + (gr * nch_ + c) * 576, 576, in + (c * g + i * ngr_ + gr) * 576);
                    side[c * g + i * ngr_ + gr] = batch[i].side[gr * nch_ + c];
                }
        }
    }
    std::size_t nch_, ngr_;
    int sr_;
    std::vector<float> overlap_, vvec_, in_;
    std::vector<std::int32_t> vfront_;
    std::vector<symaccel_mp3_side> side_;
};

// The same Layer III packets one stage earlier: what the entropy decoder produces (layer3/mod.rs:393-420) -- the Huffman samples as
// int16, the GranuleChannel fields requantize reads, one stereo record per granule for a two-channel stream -- handed to
// symaccel_mp3_decode_pipelined, which runs requantize + joint stereo + the synthesis tail on the device (layer3/mod.rs:421-477).
// Half the bytes of Mp3's f32 spectra cross PCIe, and the host keeps neither POW43 nor the band loops.
struct Mp3Huffman {
    using Sample = float;
    struct Params {
        std::size_t channels = 2;  // 1, or 2 (one joint-stereo pair)
        std::size_t granules = 2;
        int sample_rate_idx = 0;
    };
    struct Packet {
        std::uint64_t ts = 0;
        std::vector<std::int16_t> quant;            // [granule][channel][576]
        std::vector<symaccel_mp3_requant> rq;       // [granule][channel]
        std::vector<symaccel_mp3_stereo> stereo;    // [granule] (two-channel streams)
        std::vector<symaccel_mp3_side> side;        // [granule][channel], rzero as it is AFTER stereo (stereo.rs:549-553)
    };
    explicit Mp3Huffman(const Params &p)
        : nch_(p.channels), ngr_(p.granules), sr_(p.sample_rate_idx), overlap_(p.channels * 576, 0.0f), vvec_(p.channels * 1024, 0.0f),
          vfront_(p.channels, 0) {
        if (p.channels < 1 || p.channels > 2) throw std::invalid_argument("Mp3Huffman: one or two channels");
    }
    static std::uint64_t id(const Packet &p) { return p.ts; }
    std::size_t channels() const { return nch_; }
    std::size_t packet_frames(std::size_t) const { return 576 * ngr_; }
    std::size_t plane_offset(std::size_t c, std::size_t i, std::size_t k) const { return (c * k + i) * 576 * ngr_; }
    void reset_state() {
        std::fill(overlap_.begin(), overlap_.end(), 0.0f);
        std::fill(vvec_.begin(), vvec_.end(), 0.0f);
        std::fill(vfront_.begin(), vfront_.end(), 0);
    }
    void decode_batch(Context &ctx, const std::vector<Packet> &batch, std::vector<float> &pcm) {
        const std::size_t k = batch.size(), g = k * ngr_;
        q_.resize(nch_ * g * 576);
        rq_.resize(nch_ * g);
        side_.resize(nch_ * g);
        st_.resize(nch_ == 2 ? g : 0);
        pcm.resize(nch_ * g * 576);
        gather(batch, q_.data(), rq_.data(), side_.data(), st_.data());
        const std::int32_t pair[2] = {0, 1};
        check(symaccel_mp3_decode_pipelined(ctx.raw(), q_.data(), rq_.data(), nch_ == 2 ? pair : nullptr, nch_ == 2 ? st_.data() : nullptr,
                                            nch_ == 2 ? 1 : 0, side_.data(), sr_, overlap_.data(), vvec_.data(), vfront_.data(), pcm.data(), nch_, g, 0),
              ctx.raw());
    }

    static constexpr int kBatchKind = SYMACCEL_BATCH_MP3_DECODE;  // one stream per submission: exactly what this codec is
    static constexpr bool kDirect = true;
    void attach(Batcher &) {}
    int batch_param() const { return sr_; }
    std::size_t units_per_packet() const { return ngr_; }
    std::size_t batch_chains(std::size_t) const { return nch_; }
    std::size_t batch_units(std::size_t k) const { return k * ngr_; }
    struct BatchView {  // granule gr of packet i of channel c is unit i * ngr + gr of chain c; the stereo records are per granule of the STREAM
        std::int16_t *quant = nullptr;
        symaccel_mp3_requant *rq = nullptr;
        symaccel_mp3_side *side = nullptr;
        symaccel_mp3_stereo *stereo = nullptr;
        std::size_t k = 0, nch = 0, ngr = 0;
        std::size_t unit(std::size_t c, std::size_t i, std::size_t gr) const { return c * k * ngr + i * ngr + gr; }
    };
    BatchView view(const symaccel_batch_slot &slot, std::size_t k) const {
        return BatchView{static_cast<std::int16_t *>(slot.input[0]), static_cast<symaccel_mp3_requant *>(slot.input[1]),
                         static_cast<symaccel_mp3_side *>(slot.input[2]), static_cast<symaccel_mp3_stereo *>(slot.input[3]), k, nch_, ngr_};
    }
    void put(const BatchView &v, std::size_t i, const Packet &p) const {
        if (p.quant.size() != ngr_ * nch_ * 576 || p.rq.size() != ngr_ * nch_ || p.side.size() != ngr_ * nch_ || p.stereo.size() != (nch_ == 2 ? ngr_ : 0))
            throw std::invalid_argument("Mp3Huffman: packet shape");
        for (std::size_t gr = 0; gr < ngr_; ++gr) {
            for (std::size_t c = 0; c < nch_; ++c) {
                const std::size_t dst = v.unit(c, i, gr);
                std::copy_n(p.quant.data() + (gr * nch_ + c) * 576, 576, v.quant + dst * 576);
                v.rq[dst] = p.rq[gr * nch_ + c];
                v.side[dst] = p.side[gr * nch_ + c];
            }
            if (nch_ == 2) v.stereo[i * ngr_ + gr] = p.stereo[gr];
            else std::memset(&v.stereo[i * ngr_ + gr], 0, sizeof(symaccel_mp3_stereo));
        }
    }
    Packet extract(const BatchView &v, std::size_t i) const {
        Packet p;
        p.quant.resize(ngr_ * nch_ * 576);
        p.rq.resize(ngr_ * nch_);
        p.side.resize(ngr_ * nch_);
        p.stereo.resize(nch_ == 2 ? ngr_ : 0);
        for (std::size_t gr = 0; gr < ngr_; ++gr) {
            for (std::size_t c = 0; c < nch_; ++c) {
                const std::size_t src = v.unit(c, i, gr);
                std::copy_n(v.quant + src * 576, 576, p.quant.data() + (gr * nch_ + c) * 576);
                p.rq[gr * nch_ + c] = v.rq[src];
                p.side[gr * nch_ + c] = v.side[src];
            }
            if (nch_ == 2) p.stereo[gr] = v.stereo[i * ngr_ + gr];
        }
        return p;
    }
    void put_state(const symaccel_batch_slot &slot) const {
        std::memcpy(slot.state[0], overlap_.data(), overlap_.size() * sizeof(float));
        std::memcpy(slot.state[1], vvec_.data(), vvec_.size() * sizeof(float));
        std::memcpy(slot.state[2], vfront_.data(), vfront_.size() * sizeof(std::int32_t));
    }
    void fill_slot(const std::vector<Packet> &batch, const symaccel_batch_slot &slot) {
        const BatchView v = view(slot, batch.size());
        for (std::size_t i = 0; i < batch.size(); ++i) put(v, i, batch[i]);
        put_state(slot);
    }
    void take_state(const symaccel_batch_slot &slot) {
        std::memcpy(overlap_.data(), slot.state[0], overlap_.size() * sizeof(float));
        std::memcpy(vvec_.data(), slot.state[1], vvec_.size() * sizeof(float));
        std::memcpy(vfront_.data(), slot.state[2], vfront_.size() * sizeof(std::int32_t));
    }

private:
    void gather(const std::vector<Packet> &batch, std::int16_t *q, symaccel_mp3_requant *rq, symaccel_mp3_side *side, symaccel_mp3_stereo *st) const {
        const std::size_t k = batch.size(), g = k * ngr_;
        for (std::size_t i = 0; i < k; ++i) {
            const Packet &p = batch[i];
            if (p.quant.size() != ngr_ * nch_ * 576 || p.rq.size() != ngr_ * nch_ || p.side.size() != ngr_ * nch_ ||
                p.stereo.size() != (nch_ == 2 ? ngr_ : 0))
                throw std::invalid_argument("Mp3Huffman: packet shape");
            for (std::size_t gr = 0; gr < ngr_; ++gr) {
                for (std::size_t c = 0; c < nch_; ++c) {
                    const std::size_t dst = c * g + i * ngr_ + gr;
                    std::copy_n(p.quant.data() + (gr * nch_ + c) * 576, 576, q + dst * 576);
                    rq[dst] = p.rq[gr * nch_ + c];
                    side[dst] = p.side[gr * nch_ + c];
                }
                if (nch_ == 2) st[i * ngr_ + gr] = p.stereo[gr];
            }
        }
    }
    std::size_t nch_, ngr_;
    int sr_;
    std::vector<float> overlap_, vvec_;
    std::vector<std::int32_t> vfront_;
    std::vector<std::int16_t> q_;
    std::vector<symaccel_mp3_requant> rq_;
    std::vector<symaccel_mp3_stereo> st_;
    std::vector<symaccel_mp3_side> side_;
};

// Vorbis: one packet = one audio block of the size its mode's block flag selects.  What the CPU side hands over per
// channel: floor x residue, the n / 2 lines DspChannel::synth receives (lib.rs:282-331).  A packet yields
// (prev_n + n) / 4 frames -- none for the first block after a reset (dsp.rs:77-80) -- so the batch's PCM is packed
// (include/symaccel.h, "Vorbis") and the per-packet spans are what publish() asks for.
struct Vorbis {
    using Sample = float;
    struct Params {
        std::size_t channels = 2;
        int bs0_exp = 8, bs1_exp = 11;
    };
    struct Packet {
        std::uint64_t ts = 0;
        bool long_block = false;
        std::vector<float> spectra;  // [channel][n / 2]
    };
    explicit Vorbis(const Params &p)
        : nch_(p.channels), e0_(p.bs0_exp), e1_(p.bs1_exp), prev_(p.channels, -1), overlap_(p.channels * ((std::size_t)1 << (p.bs1_exp - 1)), 0.0f) {}
    static std::uint64_t id(const Packet &p) { return p.ts; }
    std::size_t channels() const { return nch_; }
    std::size_t packet_frames(std::size_t i) const { return emits_[i] ? off_[i + 1] - off_[i] : 0; }
    std::size_t plane_offset(std::size_t c, std::size_t i, std::size_t) const { return c * stride_ + off_[i]; }
    void reset_state() {  // Dsp::reset (dsp.rs:45-56): no block to lap with, overlap zeroed
        std::fill(prev_.begin(), prev_.end(), -1);
        std::fill(overlap_.begin(), overlap_.end(), 0.0f);
    }
    void decode_batch(Context &ctx, const std::vector<Packet> &batch, std::vector<float> &pcm) {
        const std::size_t k = batch.size();
        std::size_t lines = 0, samples = 0;
        layout(batch, &lines, &samples);
        stride_ = samples;
        in_.resize(nch_ * lines);
        flags_.resize(nch_ * k);
        pcm.assign(nch_ * samples, 0.0f);
        gather(batch, in_.data(), lines, flags_.data());
        check(symaccel_vorbis_synth(ctx.raw(), e0_, e1_, in_.data(), lines, flags_.data(), prev_.data(), overlap_.data(), pcm.data(), samples,
                                    nch_, k),
              ctx.raw());
    }
    // the cross-stream batcher's view: every chain's planes at their largest (k * bs1 / 2), the packed data at the front -- so that
    // streams with different block flags share a launch (SYMACCEL_BATCH_VORBIS_SYNTH)
    static constexpr int kBatchKind = SYMACCEL_BATCH_VORBIS_SYNTH;
    static constexpr bool kDirect = false;  // (the packed layout depends on every packet's block flag)
    struct BatchView {};
    void attach(Batcher &) {}
    int batch_param() const { return e0_ | (e1_ << 8); }
    std::size_t units_per_packet() const { return 1; }
    std::size_t batch_chains(std::size_t) const { return nch_; }  // a submission of k packets: nch chains of k units
    std::size_t batch_units(std::size_t k) const { return k; }
    void fill_slot(const std::vector<Packet> &batch, const symaccel_batch_slot &slot) {
        const std::size_t k = batch.size(), cap = k << (e1_ - 1);
        // (the batch being handed out keeps its offsets until this one is collected: the submitted batch's layout waits in next_*)
        std::vector<std::size_t> cur_off, cur_spec;
        std::vector<bool> cur_emits;
        cur_off.swap(off_);
        cur_spec.swap(spec_off_);
        cur_emits.swap(emits_);
        std::size_t lines = 0, samples = 0;
        layout(batch, &lines, &samples);
        gather(batch, static_cast<float *>(slot.input[0]), cap, static_cast<std::uint8_t *>(slot.input[1]));
        next_off_.swap(off_);
        next_emits_.swap(emits_);
        next_stride_ = cap;
        off_.swap(cur_off);
        spec_off_.swap(cur_spec);
        emits_.swap(cur_emits);
        std::memcpy(slot.state[0], prev_.data(), prev_.size() * sizeof(std::int32_t));
        std::memcpy(slot.state[1], overlap_.data(), overlap_.size() * sizeof(float));
    }
    void take_state(const symaccel_batch_slot &slot) {
        std::memcpy(prev_.data(), slot.state[0], prev_.size() * sizeof(std::int32_t));
        std::memcpy(overlap_.data(), slot.state[1], overlap_.size() * sizeof(float));
        off_.swap(next_off_);
        emits_.swap(next_emits_);
        stride_ = next_stride_;
    }

private:
    // packed offsets of the batch's blocks inside a chain: spec_off_[i] lines, off_[i] samples (a first block owns n / 2 untouched slots)
    void layout(const std::vector<Packet> &batch, std::size_t *lines_out, std::size_t *samples_out) {
        const std::size_t k = batch.size();
        const std::size_t bs[2] = {(std::size_t)1 << e0_, (std::size_t)1 << e1_};
        spec_off_.assign(k, 0);
        off_.assign(k + 1, 0);
        emits_.assign(k, false);
        std::size_t lines = 0, samples = 0;
        int prev = prev_[0];
        for (std::size_t i = 0; i < k; ++i) {
            const std::size_t n = bs[batch[i].long_block ? 1 : 0];
            if (batch[i].spectra.size() != nch_ * n / 2) throw std::invalid_argument("Vorbis: packet shape");
            spec_off_[i] = lines;
            off_[i] = samples;
            emits_[i] = prev >= 0;
            lines += n / 2;
            samples += prev >= 0 ? (bs[prev] + n) / 4 : n / 2;
            prev = batch[i].long_block ? 1 : 0;
        }
        off_[k] = samples;
        *lines_out = lines;
        *samples_out = samples;
    }
    void gather(const std::vector<Packet> &batch, float *in, std::size_t spec_stride, std::uint8_t *flags) const {
        const std::size_t k = batch.size();
        const std::size_t bs[2] = {(std::size_t)1 << e0_, (std::size_t)1 << e1_};
        for (std::size_t i = 0; i < k; ++i) {
            const std::size_t half = bs[batch[i].long_block ? 1 : 0] / 2;
            for (std::size_t c = 0; c < nch_; ++c) {
                std::copy_n(batch[i].spectra.data() + c * half, half, in + c * spec_stride + spec_off_[i]);
                flags[c * k + i] = batch[i].long_block ? 1 : 0;
            }
        }
    }
    std::size_t nch_;
    int e0_, e1_;
    std::vector<std::int32_t> prev_;
    std::vector<float> overlap_, in_;
    std::vector<std::uint8_t> flags_;
    std::vector<std::size_t> off_, spec_off_, next_off_;
    std::vector<bool> emits_, next_emits_;
    std::size_t stride_ = 0, next_stride_ = 0;
};

// FLAC: one packet = one frame = one subframe per channel.  What the CPU side (frame / subframe headers, Rice decode:
// flac/decoder.rs:381-660) hands over per channel: `blocksize` words -- the warm-up samples followed by the residuals,
// exactly what fixed_predict / lpc_predict receive --, the subframe descriptor and its quantised coefficients, and per
// frame the channel assignment.  The predictor restore of every subframe of the batch runs in one device call
// (decoder.rs:663-752), then the stereo decorrelation and the `<< (32 - bps)` of decoder.rs:199-242.  FLAC carries no
// state from frame to frame, so reset() has nothing to zero; block sizes may differ from packet to packet (the last
// frame; variable-block-size streams): every subframe gets a slot of the batch's largest block size.
struct Flac {
    using Sample = std::int32_t;
    struct Params {
        std::size_t channels = 2;
        std::uint32_t bits_per_sample = 16;
        std::size_t max_blocksize = 4096;  // STREAMINFO's maximum block size: the slot every subframe gets with the batcher
    };
    struct Packet {
        std::uint64_t ts = 0;
        std::size_t blocksize = 0;
        std::vector<std::int32_t> words;        // [channel][blocksize]
        std::vector<symaccel_flac_desc> desc;   // [channel]
        std::vector<std::int32_t> coeffs;       // [channel][32], reference order (decoder.rs:716-752)
        std::uint8_t pair_mode = 0;             // 0 independent, 1 left/side, 2 mid/side, 3 right/side (two-channel frames)
    };
    explicit Flac(const Params &p) : nch_(p.channels), shift_(32u - p.bits_per_sample), max_bs_(p.max_blocksize) {
        if (p.channels == 0 || p.bits_per_sample == 0 || p.bits_per_sample > 32 || p.max_blocksize == 0 || p.max_blocksize > 65535)
            throw std::invalid_argument("Flac: parameters");
    }
    // With the batcher (SYMACCEL_BATCH_FLAC_RESTORE) a chain is ONE SUBFRAME and the unit count is the stream's maximum block size, so
    // the batches of every FLAC stream with that block size share launches whatever their lengths; a two-channel stream takes the
    // fused form (param 0x100 | shift: decorrelation and left-justification in the restore kernel's write-back, decoder.rs:199-242),
    // every other layout the plain one with the shift on the host.
    static constexpr int kBatchKind = SYMACCEL_BATCH_FLAC_RESTORE;
    static constexpr bool kDirect = false;
    struct BatchView {};
    void attach(Batcher &) {}
    int batch_param() const { return nch_ == 2 ? (int)(0x100u | shift_) : 0; }
    std::size_t batch_chains(std::size_t k) const { return k * nch_; }
    std::size_t batch_units(std::size_t) const { return max_bs_; }
    void fill_slot(const std::vector<Packet> &batch, const symaccel_batch_slot &slot) {
        const std::size_t k = batch.size();
        std::int32_t *words = static_cast<std::int32_t *>(slot.input[0]);
        symaccel_flac_desc *desc = static_cast<symaccel_flac_desc *>(slot.input[1]);
        std::int32_t *coeffs = static_cast<std::int32_t *>(slot.input[2]);
        next_lens_.assign(k, 0);
        for (std::size_t i = 0; i < k; ++i) {
            const Packet &p = batch[i];
            if (p.blocksize == 0 || p.blocksize > max_bs_ || p.words.size() != nch_ * p.blocksize || p.desc.size() != nch_ || p.coeffs.size() != nch_ * 32 ||
                p.pair_mode > 3 || (p.pair_mode != 0 && nch_ != 2))
                throw std::invalid_argument("Flac: packet shape");
            next_lens_[i] = p.blocksize;
            for (std::size_t c = 0; c < nch_; ++c) {
                std::int32_t *row = words + (i * nch_ + c) * max_bs_;
                std::copy_n(p.words.data() + c * p.blocksize, p.blocksize, row);
                std::fill(row + p.blocksize, row + max_bs_, 0);  // (restoring the padding is harmless: it is never published)
                desc[i * nch_ + c] = p.desc[c];
                std::copy_n(p.coeffs.data() + c * 32, 32, coeffs + (i * nch_ + c) * 32);
            }
            if (nch_ == 2) static_cast<std::uint8_t *>(slot.input[3])[i] = p.pair_mode;
        }
    }
    void take_state(const symaccel_batch_slot &slot) {
        lens_.swap(next_lens_);
        stride_ = max_bs_;
        if (nch_ != 2) {  // decoder.rs:239-242 (the two-channel form did it on the device)
            std::int32_t *words = static_cast<std::int32_t *>(slot.out);
            for (std::size_t i = 0; i < slot.out_bytes / 4; ++i) words[i] = (std::int32_t)((std::uint32_t)words[i] << shift_);
        }
    }
    static std::uint64_t id(const Packet &p) { return p.ts; }
    std::size_t channels() const { return nch_; }
    std::size_t packet_frames(std::size_t i) const { return lens_[i]; }
    std::size_t plane_offset(std::size_t c, std::size_t i, std::size_t) const { return (i * nch_ + c) * stride_; }
    void reset_state() {}
    void decode_batch(Context &ctx, const std::vector<Packet> &batch, std::vector<std::int32_t> &pcm) {
        const std::size_t k = batch.size();
        stride_ = 0;
        lens_.assign(k, 0);
        for (std::size_t i = 0; i < k; ++i) {
            const Packet &p = batch[i];
            if (p.blocksize == 0 || p.blocksize > 65535 || p.words.size() != nch_ * p.blocksize || p.desc.size() != nch_ ||
                p.coeffs.size() != nch_ * 32 || p.pair_mode > 3 || (p.pair_mode != 0 && nch_ != 2))
                throw std::invalid_argument("Flac: packet shape");
            lens_[i] = p.blocksize;
            stride_ = std::max(stride_, p.blocksize);
        }
        pcm.assign(k * nch_ * stride_, 0);
        desc_.resize(k * nch_);
        coeffs_.resize(k * nch_ * 32);
        modes_.resize(k);
        for (std::size_t i = 0; i < k; ++i) {
            const Packet &p = batch[i];
            for (std::size_t c = 0; c < nch_; ++c) {
                std::copy_n(p.words.data() + c * p.blocksize, p.blocksize, pcm.data() + (i * nch_ + c) * stride_);
                desc_[i * nch_ + c] = p.desc[c];
                // a slot longer than its block: the restore runs over `stride_` words, the tail is zero residual and is never published
                std::copy_n(p.coeffs.data() + c * 32, 32, coeffs_.data() + (i * nch_ + c) * 32);
            }
            modes_[i] = p.pair_mode;
        }
        check(symaccel_flac_restore(ctx.raw(), pcm.data(), desc_.data(), coeffs_.data(), k * nch_, stride_), ctx.raw());
        if (nch_ == 2) {
            // pair i = (channel 0, channel 1) of packet i: rows 2 i and 2 i + 1 of the batch
            ch0_.resize(k * stride_);
            ch1_.resize(k * stride_);
            for (std::size_t i = 0; i < k; ++i) {
                std::copy_n(pcm.data() + (2 * i) * stride_, stride_, ch0_.data() + i * stride_);
                std::copy_n(pcm.data() + (2 * i + 1) * stride_, stride_, ch1_.data() + i * stride_);
            }
            check(symaccel_flac_decorrelate(ctx.raw(), modes_.data(), ch0_.data(), ch1_.data(), k, stride_, shift_), ctx.raw());
            for (std::size_t i = 0; i < k; ++i) {
                std::copy_n(ch0_.data() + i * stride_, stride_, pcm.data() + (2 * i) * stride_);
                std::copy_n(ch1_.data() + i * stride_, stride_, pcm.data() + (2 * i + 1) * stride_);
            }
        } else {
            for (std::int32_t &v : pcm) v = (std::int32_t)((std::uint32_t)v << shift_);  // decoder.rs:239-242
        }
    }

private:
    std::size_t nch_;
    std::uint32_t shift_;
    std::size_t max_bs_;
    std::size_t stride_ = 0;
    std::vector<std::size_t> lens_, next_lens_;
    std::vector<symaccel_flac_desc> desc_;
    std::vector<std::int32_t> coeffs_, ch0_, ch1_;
    std::vector<std::uint8_t> modes_;
};

// CodecRegistry (symphonia-core/src/codecs/registry.rs:220-341).  The reference's registry maps a codec id to a factory that is handed
// (params, options) and NOTHING ELSE -- `RegisterableAudioDecoder::try_registry_new` (registry.rs:34-44) -- so a decoder it builds cannot
// be told about its siblings: the factories here find the process-wide batcher themselves (`Batcher::shared()`), exactly as the Rust shim's
// `try_registry_new` goes through `Pool::shared()` (decoder.rs, aac.rs).  `register_audio_decoder` = registry.rs:252-269,
// `make_audio_decoder` = registry.rs:330-341 (`Error::Unsupported("core (codec): unsupported codec")` for an id nobody registered).
// The packet source stands where the reference has the FormatReader the decoder is read behind (LookaheadReader in the shim).
struct AudioDecoderOptions {  // codecs/audio.rs:208-222 (`verify`) + the look-ahead of this backend (the shim's DEFAULT_LOOKAHEAD = 256)
    bool verify = false;
    std::size_t lookahead = 256;
};

class CodecRegistry {
public:
    template <class Codec>
    void register_audio_decoder() {
        ids_.insert(std::type_index(typeid(Codec)));
    }
    template <class Codec>
    bool is_registered() const {
        return ids_.count(std::type_index(typeid(Codec))) != 0;
    }
    template <class Codec>
    std::unique_ptr<LookaheadDecoder<Codec>> make_audio_decoder(const typename Codec::Params &params, const AudioDecoderOptions &opts,
                                                                typename LookaheadDecoder<Codec>::Peek peek) const {
        require<Codec>();
        return std::make_unique<LookaheadDecoder<Codec>>(Batcher::shared(), params, opts.lookahead, std::move(peek));
    }
    // (zero-copy parse, for the codecs with kDirect)
    template <class Codec>
    std::unique_ptr<LookaheadDecoder<Codec>> make_audio_decoder(const typename Codec::Params &params, const AudioDecoderOptions &opts,
                                                                typename LookaheadDecoder<Codec>::Direct direct) const {
        require<Codec>();
        return std::make_unique<LookaheadDecoder<Codec>>(Batcher::shared(), params, opts.lookahead, std::move(direct));
    }

private:
    template <class Codec>
    void require() const {
        if (!is_registered<Codec>()) throw Error(Error::Kind::Unsupported, SYMACCEL_ERR_UNSUPPORTED, "core (codec): unsupported codec");
    }
    std::set<std::type_index> ids_;
};

// symphonia::default::register_enabled_codecs' counterpart for this backend (the shim's `register()`): every codec of the twin
inline void register_enabled_codecs(CodecRegistry &registry) {
    registry.register_audio_decoder<AacLc>();
    registry.register_audio_decoder<AacLcCoded>();
    registry.register_audio_decoder<Mp3>();
    registry.register_audio_decoder<Mp3Huffman>();
    registry.register_audio_decoder<Vorbis>();
    registry.register_audio_decoder<Flac>();
}

}  // namespace codecs

}  // namespace symphonia_accel

#endif  // SYMACCEL_HPP
