// GPU test of the C++ host mirror (include/symaccel.hpp), written the way the reference writes its unit tests:
// known-answer vectors against f64 closed forms at the reference's tolerance (mdct.rs:177-201, dsp/fft/mod.rs:155-186),
// plus frame-by-frame decoder-style sequences through the per-packet calls, compared bit-for-bit with the CPU oracle
// (oracle/symoracle.c is linked here as the checker only).  Built and run by tests/test_cpp_host.py.
#include <cmath>
#include <cstdio>
#include <cstring>
#include <random>
#include <vector>

#include "symaccel.hpp"
#include "symoracle.h"
#include "kats.inc"  // generated from tests/golden/ref_kats.json: kImdct32Input[32], kFft64Input[64][2]

using namespace symphonia_accel;

static int g_failures = 0;
#define EXPECT(cond, ...)                                  \
    do {                                                   \
        if (!(cond)) {                                     \
            ++g_failures;                                  \
            std::printf("FAIL %s:%d: ", __FILE__, __LINE__); \
            std::printf(__VA_ARGS__);                      \
            std::printf("\n");                             \
        }                                                  \
    } while (0)

static bool same_bits(const float *a, const float *b, size_t n) {
    for (size_t i = 0; i < n; ++i) {
        if (a[i] == b[i]) continue;  // also +0 == -0
        if (std::isnan(a[i]) && std::isnan(b[i])) continue;
        return false;
    }
    return true;
}

// mdct.rs:154-175
static void imdct_analytical(const float *x, double *y, size_t n, double scale) {
    for (size_t i = 0; i < 2 * n; ++i) {
        double acc = 0.0;
        for (size_t j = 0; j < n; ++j) acc += x[j] * std::cos(M_PI / (4.0 * n) * (2.0 * i + 1.0 + n) * (2.0 * j + 1.0));
        y[i] = scale * acc;
    }
}

static void verify_imdct(Context &ctx) {  // mdct.rs:177-201
    const double scale = std::sqrt(2.0 / 64.0);
    auto imdct = dsp::mdct::Imdct::new_scaled(ctx, 32, scale);
    float actual[64];
    double expected[64];
    imdct.imdct(kImdct32Input, 32, actual, 64);
    imdct_analytical(kImdct32Input, expected, 32, scale);
    for (int i = 0; i < 64; ++i) EXPECT(std::fabs(actual[i] - expected[i]) < 1e-5, "imdct[%d] %g vs %g", i, actual[i], expected[i]);
    bool threw = false;
    try {
        imdct.imdct(kImdct32Input, 31, actual, 64);  // mdct.rs:76 assert
    } catch (const std::invalid_argument &) {
        threw = true;
    }
    EXPECT(threw, "Imdct::imdct accepted a wrong spec length");
    threw = false;
    try {
        dsp::mdct::Imdct bad(ctx, 24);  // mdct.rs:37 assert
    } catch (const std::invalid_argument &) {
        threw = true;
    }
    EXPECT(threw, "Imdct::new accepted a non power of two");
}

static void verify_fft(Context &ctx) {  // dsp/fft/mod.rs:155-186
    using dsp::fft::Complex;
    std::vector<Complex> x(64), y(64), z(64);
    for (int i = 0; i < 64; ++i) x[i] = Complex(kFft64Input[i][0], kFft64Input[i][1]);
    dsp::fft::Fft fft(ctx, 64);
    EXPECT(fft.size() == 64, "Fft::size");
    fft.fft(x.data(), 64, y.data(), 64);
    for (int k = 0; k < 64; ++k) {  // naive DFT in f64 (dsp/fft/mod.rs:38-64)
        std::complex<double> acc = 0;
        for (int j = 0; j < 64; ++j) acc += std::complex<double>(x[j]) * std::polar(1.0, -2.0 * M_PI * k * j / 64.0);
        EXPECT(std::abs(std::complex<double>(y[k]) - acc) < 1e-5, "fft[%d]", k);
    }
    z = x;
    fft.fft_inplace(z.data(), 64);
    EXPECT(std::memcmp(z.data(), y.data(), 64 * sizeof(Complex)) == 0, "fft_inplace != fft");
    float ref[128];
    so_fft(reinterpret_cast<const float *>(x.data()), ref, 64);
    EXPECT(same_bits(ref, reinterpret_cast<const float *>(y.data()), 128), "fft differs from the oracle");
    // Ifft (no_simd.rs:143-219): against the oracle bit for bit, and ifft(fft(x)) == x to the reference's own 1e-5
    dsp::fft::Ifft ifft(ctx, 64);
    ifft.ifft(x.data(), 64, z.data(), 64);
    so_ifft(reinterpret_cast<const float *>(x.data()), ref, 64);
    EXPECT(same_bits(ref, reinterpret_cast<const float *>(z.data()), 128), "ifft differs from the oracle");
    z = y;
    ifft.ifft_inplace(z.data(), 64);
    for (int k = 0; k < 64; ++k) EXPECT(std::abs(z[k] - x[k]) < 1e-5f * 32.0f, "ifft(fft(x))[%d]", k);
    // the reference's limit (1 << 16 points) is the mirror's limit
    bool threw = false;
    try {
        dsp::fft::Fft too_big(ctx, 1u << 17);
    } catch (const std::invalid_argument &) {
        threw = true;
    }
    EXPECT(threw, "Fft::new accepted more than MAX_SIZE points");
}

static void verify_aac(Context &ctx) {  // decoder-style: one Dsp::synth per frame (aac/mod.rs:217-220)
    std::mt19937 rng(7);
    std::normal_distribution<float> nd(0.f, 100.f);
    const std::uint8_t seqs[] = {0, 1, 2, 2, 3, 0, 0, 1, 3, 0};
    aac::Dsp dsp(ctx);
    std::array<float, 1024> delay{}, delay_ref{}, coeffs;
    for (auto &d : delay) d = nd(rng) * 0.01f;
    delay_ref = delay;
    bool prev_shape = false;
    for (int t = 0; t < 10; ++t) {
        for (auto &c : coeffs) c = nd(rng);
        const bool shape = (rng() & 1) != 0;
        float dst[1024], want[1024];
        dsp.synth(coeffs, delay, seqs[t], shape, prev_shape, dst, 1024);
        so_aac_synth(coeffs.data(), delay_ref.data(), seqs[t], shape, prev_shape, want);
        EXPECT(same_bits(dst, want, 1024), "aac frame %d (seq %d) differs from the oracle", t, seqs[t]);
        EXPECT(same_bits(delay.data(), delay_ref.data(), 1024), "aac delay after frame %d", t);
        prev_shape = shape;
    }
}

static void verify_mp3(Context &ctx) {  // Layer3::decode granule loop tail (layer3/mod.rs:440-476), one channel
    std::mt19937 rng(9);
    std::normal_distribution<float> nd(0.f, 0.1f);
    mp3::SynthesisState st;
    std::vector<float> ov(576, 0.f), vv(1024, 0.f);
    std::int32_t vf = 0;
    const mp3::BlockType bts[] = {mp3::BlockType::Long, mp3::BlockType::Start, mp3::BlockType::Short, mp3::BlockType::Short,
                                  mp3::BlockType::End, mp3::BlockType::Long};
    for (int g = 0; g < 6; ++g) {
        std::array<float, 576> xr{};
        mp3::GranuleChannel gc;
        gc.block_type = bts[g];
        gc.is_mixed = (g == 3);
        gc.rzero = (std::uint16_t)(2 * (rng() % 289));
        for (int i = 0; i < gc.rzero; ++i) xr[i] = nd(rng);
        float out[576], want[576];
        mp3::synthesize_granule(ctx, 0, gc, xr, st, out, 576);
        const std::uint8_t side[4] = {(std::uint8_t)gc.block_type, (std::uint8_t)gc.is_mixed, (std::uint8_t)(gc.rzero & 0xff),
                                      (std::uint8_t)(gc.rzero >> 8)};
        so_mp3_synth_batch(xr.data(), side, 0, ov.data(), vv.data(), &vf, want, 1, 1);
        EXPECT(same_bits(out, want, 576), "mp3 granule %d differs from the oracle", g);
        EXPECT(same_bits(st.overlap.data(), ov.data(), 576) && same_bits(st.v_vec.data(), vv.data(), 1024) && st.v_front == vf,
               "mp3 state after granule %d", g);
    }
}

static void verify_mp3_requantize(Context &ctx) {  // Layer3::decode: read_huffman_samples' values + requantize (layer3/mod.rs:393-400)
    std::mt19937 rng(21);
    for (int g = 0; g < 6; ++g) {
        mp3::RequantizeChannel ch;
        ch.global_gain = (std::uint8_t)(120 + rng() % 100);
        ch.scalefac_scale = g & 1;
        ch.preflag = g & 2;
        ch.block_type = g < 3 ? mp3::BlockType::Long : mp3::BlockType::Short;
        ch.is_mixed = g == 5;
        for (auto &v : ch.subblock_gain) v = (std::uint8_t)(rng() % 8);
        for (auto &v : ch.scalefacs) v = (std::uint8_t)(rng() % 16);
        ch.rzero = (std::uint16_t)(g == 0 ? 576 : rng() % 577);
        std::array<std::int16_t, 576> q{};
        for (int i = 0; i < 576; ++i) q[(size_t)i] = (std::int16_t)((int)(rng() % 61) - 30);
        q[7] = 8206;
        q[8] = -8206;
        std::array<float, 576> got{};
        float want[576];
        mp3::requantize(ctx, g % 3, ch, q, got);
        so_mp3_requant d{};
        const symaccel_mp3_requant abi = mp3::to_abi(ch);
        static_assert(sizeof(d) == sizeof(abi), "the oracle's record mirrors the ABI record");
        std::memcpy(&d, &abi, sizeof d);
        so_mp3_requantize(q.data(), &d, g % 3, want);
        EXPECT(same_bits(got.data(), want, 576), "mp3 requantize case %d differs from the oracle", g);
    }
}

static void verify_vorbis(Context &ctx) {  // Decoder::decode_inner channel loop (vorbis/lib.rs:296-331)
    std::mt19937 rng(11);
    std::normal_distribution<float> nd(0.f, 0.25f);
    vorbis::Dsp dsp(ctx, 8, 11);
    vorbis::DspChannel ch(dsp);
    const bool flags[] = {true, true, false, false, false, true, false, true};
    std::vector<float> ov_ref(1024, 0.f);
    std::int32_t pf_ref = -1;
    std::optional<bool> prev;
    for (int b = 0; b < 8; ++b) {
        const int n = flags[b] ? 2048 : 256;
        std::vector<float> spec(n / 2), out(1024), want(1024);
        for (auto &v : spec) v = nd(rng);
        const std::size_t got = ch.synth(flags[b], prev, spec.data(), out.data(), out.size());
        const std::uint8_t f = flags[b];
        so_vorbis_synth_batch(8, 11, spec.data(), spec.size(), &f, &pf_ref, ov_ref.data(), want.data(), want.size(), 1, 1);
        const int prev_n = (prev.value_or(flags[b])) ? 2048 : 256;
        EXPECT(got == (std::size_t)(prev_n + n) / 4, "vorbis block %d output length", b);
        EXPECT(same_bits(out.data(), want.data(), got), "vorbis block %d differs from the oracle", b);
        EXPECT(same_bits(ch.overlap().data(), ov_ref.data(), 1024), "vorbis overlap after block %d", b);
        prev = flags[b];
    }
}

static void verify_flac(Context &ctx) {  // decode_linear / decode_fixed_linear (decoder.rs:446-511, 425-444)
    std::mt19937 rng(13);
    for (std::size_t order : {1u, 8u, 12u, 32u}) {
        std::vector<std::int32_t> buf(4096), ref, co(order);
        for (auto &v : buf) v = (std::int32_t)(rng() % (1u << 16)) - (1 << 15);
        for (auto &c : co) c = (std::int32_t)(rng() % 2001) - 1000;
        co[0] = 6000;
        ref = buf;
        flac::lpc_predict(ctx, order, co.data(), 12, buf.data(), buf.size());
        so_flac_lpc_predict((int)order, co.data(), 12, ref.data(), ref.size());
        EXPECT(buf == ref, "lpc_predict order %zu", order);
    }
    for (std::size_t order = 0; order <= 4; ++order) {
        std::vector<std::int32_t> buf(1000), ref;
        for (auto &v : buf) v = (std::int32_t)(rng() % 4001) - 2000;
        ref = buf;
        flac::fixed_predict(ctx, order, buf.data(), buf.size());
        so_flac_fixed_predict((int)order, ref.data(), ref.size());
        EXPECT(buf == ref, "fixed_predict order %zu", order);
    }
    EXPECT(flac::row_stride(4096) == 4608 && flac::row_stride(4000) == 4000 && flac::row_stride(1023) == 1152, "row_stride");
    std::vector<std::int32_t> a(777), b(777), ra, rb;
    for (auto &v : a) v = (std::int32_t)rng();
    for (auto &v : b) v = (std::int32_t)rng();
    ra = a, rb = b;
    flac::decorrelate_mid_side(ctx, a.data(), b.data(), a.size());
    so_flac_decorrelate(2, ra.data(), rb.data(), ra.size());
    EXPECT(a == ra && b == rb, "decorrelate_mid_side");
    ra = a, rb = b;
    flac::decorrelate_left_side(ctx, a.data(), b.data(), a.size());
    so_flac_decorrelate(1, ra.data(), rb.data(), ra.size());
    EXPECT(b == rb, "decorrelate_left_side");
    ra = a, rb = b;
    flac::decorrelate_right_side(ctx, a.data(), b.data(), a.size());  // (right = a, side = b): side += right
    so_flac_decorrelate(3, rb.data(), ra.data(), ra.size());
    EXPECT(b == rb, "decorrelate_right_side");
    bool threw = false;
    try {
        flac::lpc_predict(ctx, 33, a.data(), 0, b.data(), b.size());
    } catch (const std::invalid_argument &) {
        threw = true;
    }
    EXPECT(threw, "lpc_predict accepted order 33");
}

static void verify_alac(Context &ctx) {  // decode_element (alac/lib.rs:540-560)
    std::mt19937 rng(17);
    for (std::uint32_t order : {4u, 8u, 31u}) {
        alac::ElementChannel e;
        e.bps = 16, e.mode = order == 8 ? 15u : 0u, e.shift = 9, e.lpc_order = order;
        for (std::uint32_t j = 0; j < order; ++j) e.lpc_coeffs[j] = (std::int32_t)(rng() % 1001) - 500;
        std::vector<std::int32_t> buf(4096), ref;
        for (auto &v : buf) v = (std::int32_t)(rng() % 2001) - 1000;
        ref = buf;
        e.predict(ctx, buf.data(), buf.size());
        so_alac_predict(ref.data(), ref.size(), e.mode, e.lpc_order, e.shift, e.bps, e.lpc_coeffs.data());
        EXPECT(buf == ref, "alac predict order %u", order);
    }
    std::vector<std::int32_t> a(500), b(500), ra, rb;
    for (auto &v : a) v = (std::int32_t)rng();
    for (auto &v : b) v = (std::int32_t)rng();
    ra = a, rb = b;
    alac::decorrelate_mid_side(ctx, a.data(), b.data(), a.size(), 3, 2);
    so_alac_decorrelate_mid_side(ra.data(), rb.data(), ra.size(), 3, 2);
    EXPECT(a == ra && b == rb, "alac decorrelate_mid_side");
    bool threw = false;
    try {
        alac::ElementChannel bad;
        bad.mode = 7, bad.lpc_order = 4;
        bad.predict(ctx, a.data(), a.size());
    } catch (const Error &) {
        threw = true;
    }
    EXPECT(threw, "alac accepted mode 7");
}

int main(int argc, char **argv) {
    if (argc > 1 && std::strcmp(argv[1], "--expect-no-device") == 0) {
        // there is no CPU path: without an MI355X the context constructor must fail with the IoError class
        try {
            Context ctx(0);
        } catch (const Error &e) {
            return e.kind == Error::Kind::IoError ? 0 : 2;
        }
        std::printf("a Context was created without a GPU\n");
        return 1;
    }
    Context ctx(0);
    verify_imdct(ctx);
    verify_fft(ctx);
    verify_aac(ctx);
    verify_mp3(ctx);
    verify_mp3_requantize(ctx);
    verify_vorbis(ctx);
    verify_flac(ctx);
    verify_alac(ctx);
    std::printf(g_failures ? "%d failure(s)\n" : "host mirror: all checks passed\n", g_failures);
    return g_failures ? 1 : 0;
}
