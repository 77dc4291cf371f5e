// Host-side constant tables.  Every table is generated with the same libm calls, in the same
// precision and order as the reference, so the uploaded bits are the
// reference's bits; the kernels never call device transcendental functions.
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>

#include "spec_tables.h"
#include "symaccel_internal.h"

namespace symaccel {

namespace {

constexpr double kPi = 3.14159265358979323846264338327950288;

// symphonia-codec-aac/src/aac/window.rs:56-63 (63-term Horner series, f64)
double bessel_i0(double x) {
    double v = 1.0;
    for (int n = 63; n >= 1; --n) {
        v *= x / (double)(n * n);
        v += 1.0;
    }
    return v;
}

// window.rs:30-35: all in f32, libm sinf
void aac_sine(int size, std::vector<float> &w) {
    w.resize(size);
    const float pi_f = 3.14159265358979323846264338327950288f;
    const float param = pi_f / (float)(2 * size);
    for (int n = 0; n < size; ++n) w[n] = sinf(((float)n + 0.5f) * param) * 1.0f;
}

// window.rs:37-52: alpha^2 formed in f32 then widened; cumulative Bessel sums in f64
void aac_kbd(float alpha, int size, std::vector<float> &w) {
    w.resize(size);
    const float pi_f = 3.14159265358979323846264338327950288f;
    const float t = alpha * pi_f / (float)size;
    const double alpha2 = (double)(t * t);
    std::vector<double> kb(size);
    double sum = 0.0;
    for (int n = 0; n < size; ++n) {
        sum += bessel_i0((double)((long)n * (long)(size - n)) * alpha2);
        kb[n] = sum;
    }
    sum += 1.0;
    for (int n = 0; n < size; ++n) w[n] = (float)std::sqrt(kb[n] / sum);
}

void build_small(int n, cpx *w, uint8_t *form) {
    // fft16 / fft32 combine constants (no_simd.rs:307-324, 374-383).  k = n/8 and 3n/8 are
    // the reference's (a+b, b-a) / (a-b, a+b) forms; multiplying by (c,-c) / (-c,-c) with the
    // plain complex product performs exactly those operations (DESIGN.md "Arithmetic contract").
    const float c = 0.70710678118654752440f;  // f32::consts::FRAC_1_SQRT_2
    for (int k = 0; k < n / 2; ++k) {
        form[k] = 0;
        if (k == 0) {
            w[k] = cpx{1.0f, 0.0f};
            form[k] = 1;
        } else if (4 * k == n) {
            w[k] = cpx{0.0f, -1.0f};
            form[k] = 2;
        } else if (8 * k == n) {
            w[k] = cpx{c, -c};
        } else if (8 * k == 3 * n) {
            w[k] = cpx{-c, -c};
        } else {
            w[k].re = (float)std::cos(2.0 * kPi * k / n);
            w[k].im = (float)(-std::sin(2.0 * kPi * k / n));
        }
    }
}

// Short scale-factor band widths (ISO/IEC 11172-3 Table B.8, 13818-3 Table B.2), one row per
// sample-rate index; the band edge table (layer3/common.rs:60-106) repeats each width 3 times.
const unsigned char kShortWidths[9][13] = {
    {4, 4, 4, 4, 6, 8, 10, 12, 14, 18, 22, 30, 56},  {4, 4, 4, 4, 6, 6, 10, 12, 14, 16, 20, 26, 66},
    {4, 4, 4, 4, 6, 8, 12, 16, 20, 26, 34, 42, 12},  {4, 4, 4, 6, 6, 8, 10, 14, 18, 26, 32, 42, 18},
    {4, 4, 4, 6, 8, 10, 12, 14, 18, 24, 32, 44, 12}, {4, 4, 4, 6, 8, 10, 12, 14, 18, 24, 30, 40, 18},
    {4, 4, 4, 6, 8, 10, 12, 14, 18, 24, 30, 40, 18}, {4, 4, 4, 6, 8, 10, 12, 14, 18, 24, 30, 40, 18},
    {8, 8, 8, 12, 16, 20, 24, 28, 36, 2, 2, 2, 26},
};
// Long scale-factor band widths (ISO/IEC 11172-3 Table B.8, 13818-3 Table B.2; 8 kHz as the reference has it):
// the band edge table (layer3/common.rs:9-56) is their running sum.
const unsigned char kLongWidths[9][22] = {
    {4, 4, 4, 4, 4, 4, 6, 6, 8, 8, 10, 12, 16, 20, 24, 28, 34, 42, 50, 54, 76, 158},
    {4, 4, 4, 4, 4, 4, 6, 6, 6, 8, 10, 12, 16, 18, 22, 28, 34, 40, 46, 54, 54, 192},
    {4, 4, 4, 4, 4, 4, 6, 6, 8, 10, 12, 16, 20, 24, 30, 38, 46, 56, 68, 84, 102, 26},
    {6, 6, 6, 6, 6, 6, 8, 10, 12, 14, 16, 20, 24, 28, 32, 38, 46, 52, 60, 68, 58, 54},
    {6, 6, 6, 6, 6, 6, 8, 10, 12, 14, 16, 18, 22, 26, 32, 38, 46, 54, 62, 70, 76, 36},
    {6, 6, 6, 6, 6, 6, 8, 10, 12, 14, 16, 20, 24, 28, 32, 38, 46, 52, 60, 68, 58, 54},
    {6, 6, 6, 6, 6, 6, 8, 10, 12, 14, 16, 20, 24, 28, 32, 38, 46, 52, 60, 68, 58, 54},
    {6, 6, 6, 6, 6, 6, 8, 10, 12, 14, 16, 20, 24, 28, 32, 38, 46, 52, 60, 68, 58, 54},
    {12, 12, 12, 12, 12, 12, 16, 20, 24, 28, 32, 40, 48, 56, 64, 76, 90, 2, 2, 2, 2, 2},
};
// Long-band edges in front of the short bands of a mixed block (layer3/common.rs:108-168).
const unsigned char kMixedLong[9][9] = {
    {0, 4, 8, 12, 16, 20, 24, 30, 36}, {0, 4, 8, 12, 16, 20, 24, 30, 36}, {0, 4, 8, 12, 16, 20, 24, 30, 36},
    {0, 6, 12, 18, 24, 30, 36, 0, 0},  {0, 6, 12, 18, 24, 30, 36, 0, 0},  {0, 6, 12, 18, 24, 30, 36, 0, 0},
    {0, 6, 12, 18, 24, 30, 36, 0, 0},  {0, 6, 12, 18, 24, 30, 36, 0, 0},  {0, 12, 24, 36, 0, 0, 0, 0, 0},
};
const unsigned char kMixedLongLen[9] = {9, 9, 9, 7, 7, 7, 7, 7, 4};
const unsigned short kMixed8kTail[36] = {40,  44,  48,  56,  64,  72,  84,  96,  108, 124, 140, 156,
                                         176, 196, 216, 240, 264, 288, 316, 344, 372, 408, 444, 480,
                                         482, 484, 486, 488, 490, 492, 494, 496, 498, 524, 550, 576};

void build(HostTables &t) {
    aac_kbd(4.0f, 1024, t.aac_kbd_long);
    aac_kbd(6.0f, 128, t.aac_kbd_short);
    aac_sine(1024, t.aac_sine_long);
    aac_sine(128, t.aac_sine_short);
    t.aac_tw_long.resize(512);
    t.aac_tw_short.resize(64);
    make_imdct_twiddles(1024, 1.0 / 2048.0, t.aac_tw_long.data());  // aac/dsp.rs:49
    make_imdct_twiddles(128, 1.0 / 256.0, t.aac_tw_short.data());   // aac/dsp.rs:50

    // W64 .. W65536 (no_simd.rs:16-67 generates them lazily per size; the largest legal Fft has 65536 points)
    t.fft_merge.resize(fft_merge_offset(131072));
    for (int n = 64; n <= 65536; n <<= 1) make_fft_twiddles(n, t.fft_merge.data() + fft_merge_offset(n));
    build_small(16, t.small16, t.small16_form);
    build_small(32, t.small32, t.small32_form);

    // MP3 windows (hybrid_synthesis.rs:53-92)
    const double pi36 = kPi / 36.0, pi12 = kPi / 12.0, pi24 = kPi / 24.0;
    std::memset(t.mp3_imdct_win, 0, sizeof t.mp3_imdct_win);
    for (int i = 0; i < 36; ++i) t.mp3_imdct_win[0][i] = (float)std::sin(pi36 * ((double)i + 0.5));
    for (int i = 0; i < 18; ++i) t.mp3_imdct_win[1][i] = t.mp3_imdct_win[0][i];
    for (int i = 18; i < 24; ++i) t.mp3_imdct_win[1][i] = 1.0f;
    for (int i = 24; i < 30; ++i) t.mp3_imdct_win[1][i] = (float)std::sin(pi12 * ((double)(i - 18) + 0.5));
    for (int i = 0; i < 12; ++i) t.mp3_imdct_win[2][i] = (float)std::sin(pi12 * ((double)i + 0.5));
    for (int i = 6; i < 12; ++i) t.mp3_imdct_win[3][i] = (float)std::sin(pi12 * ((double)(i - 6) + 0.5));
    for (int i = 12; i < 18; ++i) t.mp3_imdct_win[3][i] = 1.0f;
    for (int i = 18; i < 36; ++i) t.mp3_imdct_win[3][i] = t.mp3_imdct_win[0][i];
    // hybrid_synthesis.rs:105-119
    for (int i = 0; i < 6; ++i)
        for (int k = 0; k < 6; ++k)
            t.mp3_cos12[i][k] = (float)std::cos(pi24 * (double)((2 * (i + 3) + 7) * (2 * k + 1)));
    // hybrid_synthesis.rs:136-149
    const double ci[8] = {-0.6, -0.535, -0.33, -0.185, -0.095, -0.041, -0.0142, -0.0037};
    for (int i = 0; i < 8; ++i) {
        const double s = std::sqrt(1.0 + (ci[i] * ci[i]));
        t.mp3_cs[i] = (float)(1.0 / s);
        t.mp3_ca[i] = (float)(ci[i] / s);
    }
    // decimal literals of hybrid_synthesis.rs:611-630, 668-678, 722-730 and synthesis.rs:354-396:
    // each equals its f64 closed form rounded to f32 (tests/test_tables.py compares the bits with
    // the reference's literals recorded in tests/golden/ref_kats.json).
    for (int m = 0; m < 18; ++m) t.mp3_dct_iv_scale[m] = (float)(2.0 * std::cos(kPi * (2 * m + 1) / 72.0));
    for (int m = 0; m < 9; ++m) t.mp3_sdct18_scale[m] = (float)(2.0 * std::cos(kPi * (2 * m + 1) / 36.0));
    t.mp3_sdct18_scale[4] = 1.41421356237309504880168872420969808f;  // f32::consts::SQRT_2
    t.mp3_sdct9_d[0] = (float)(-std::sqrt(3.0));
    t.mp3_sdct9_d[1] = (float)(-2.0 * std::cos(8.0 * kPi / 9.0));
    t.mp3_sdct9_d[2] = (float)(-2.0 * std::cos(4.0 * kPi / 9.0));
    t.mp3_sdct9_d[3] = (float)(-2.0 * std::cos(2.0 * kPi / 9.0));
    t.mp3_sdct9_d[4] = (float)(-2.0 * std::sin(8.0 * kPi / 9.0));
    t.mp3_sdct9_d[5] = (float)(-2.0 * std::sin(4.0 * kPi / 9.0));
    t.mp3_sdct9_d[6] = (float)(-2.0 * std::sin(2.0 * kPi / 9.0));
    for (int i = 0; i < 16; ++i) t.mp3_cos16[i] = (float)(1.0 / (2.0 * std::cos(kPi * (2 * i + 1) / 64.0)));
    for (int i = 0; i < 8; ++i) t.mp3_cos8[i] = (float)(1.0 / (2.0 * std::cos(kPi * (2 * i + 1) / 32.0)));
    for (int i = 0; i < 4; ++i) t.mp3_cos4[i] = (float)(1.0 / (2.0 * std::cos(kPi * (2 * i + 1) / 16.0)));
    for (int i = 0; i < 2; ++i) t.mp3_cos2[i] = (float)(1.0 / (2.0 * std::cos(kPi * (2 * i + 1) / 8.0)));
    t.mp3_cos1 = 0.7071067811865475f;
    // synthesis.rs:13-142: D[i] = n[i] / 65536 as printed by the standard (9 decimals) -> f32
    for (int i = 0; i < 512; ++i) {
        char lit[32];
        std::snprintf(lit, sizeof lit, "%.9f", (double)SYM_MP3_SYNTH_WINDOW_Q16[i] / 65536.0);
        t.mp3_synth_d[i] = std::strtof(lit, nullptr);
    }
    // band edge tables
    for (int sr = 0; sr < 9; ++sr) {
        int acc = 0, n = 0;
        t.mp3_sfb_short[sr][n++] = 0;
        for (int b = 0; b < 13; ++b)
            for (int w = 0; w < 3; ++w) t.mp3_sfb_short[sr][n++] = (acc += kShortWidths[sr][b]);
        int m = 0;
        for (int i = 0; i < kMixedLongLen[sr]; ++i) t.mp3_sfb_mixed[sr][m++] = kMixedLong[sr][i];
        if (sr == 8) {
            for (int i = 0; i < 36; ++i) t.mp3_sfb_mixed[sr][m++] = kMixed8kTail[i];
        } else {
            int k = 0;
            while (t.mp3_sfb_short[sr][k] != 36) ++k;
            for (++k; k < 40; ++k) t.mp3_sfb_mixed[sr][m++] = t.mp3_sfb_short[sr][k];
        }
        t.mp3_sfb_mixed_len[sr] = m;
        for (; m < 40; ++m) t.mp3_sfb_mixed[sr][m] = -1;
        t.mp3_sfb_switch[sr] = kMixedLongLen[sr] - 1;  // SFB_MIXED_SWITCH_POINT
        t.mp3_sfb_long[sr][0] = 0;
        for (int b = 0; b < 22; ++b) t.mp3_sfb_long[sr][b + 1] = t.mp3_sfb_long[sr][b] + kLongWidths[sr][b];
        // line -> scale index.  Long: the band; short: the (band, window) slot = scalefacs index; mixed: the long
        // bands of requantize_long(bands[..switch]) (one fewer than the edges), then scalefacs[switch + i] for the
        // short slots; lines in between are covered by no band (requantize.rs:368-372).
        uint8_t(*map)[576] = t.mp3_band_map[sr];
        std::memset(map, kMp3Unscaled, 4 * 576);
        for (int b = 0; b < 22; ++b)
            for (int i = t.mp3_sfb_long[sr][b]; i < t.mp3_sfb_long[sr][b + 1]; ++i) map[0][i] = (uint8_t)b;
        for (int b = 0; b < 39; ++b)
            for (int i = t.mp3_sfb_short[sr][b]; i < t.mp3_sfb_short[sr][b + 1]; ++i) map[1][i] = (uint8_t)b;
        const int sw = t.mp3_sfb_switch[sr], len = t.mp3_sfb_mixed_len[sr];
        for (int b = 0; b + 1 < sw; ++b)
            for (int i = t.mp3_sfb_mixed[sr][b]; i < t.mp3_sfb_mixed[sr][b + 1]; ++i) map[2][i] = (uint8_t)b;
        for (int b = sw; b + 1 < len; ++b)
            for (int i = t.mp3_sfb_mixed[sr][b]; i < t.mp3_sfb_mixed[sr][b + 1]; ++i) map[2][i] = (uint8_t)b;
        // stereo's view of a mixed block: every interval of the edge table (long bands bands[..switch + 1], stereo.rs:357)
        for (int b = 0; b + 1 < len; ++b)
            for (int i = t.mp3_sfb_mixed[sr][b]; i < t.mp3_sfb_mixed[sr][b + 1]; ++i) map[3][i] = (uint8_t)b;
    }
    // requantize.rs:28-31: f32::powf(i as f32, 4.0 / 3.0); requantize.rs:280, 343: f64::powf(2.0, 0.25 * e) as f32
    // stereo.rs:105-116 (MPEG-1: tan) and :60-79 (MPEG-2 / 2.5: powf of 1/sqrt(sqrt 2) or 1/sqrt 2)
    for (int is_pos = 0; is_pos < 7; ++is_pos) {
        const double is_ratio = std::tan((kPi / 12.0) * (double)is_pos);
        t.mp3_is_ratios[is_pos][0] = (float)(is_ratio / (1.0 + is_ratio));
        t.mp3_is_ratios[is_pos][1] = (float)(1.0 / (1.0 + is_ratio));
    }
    t.mp3_is_ratios[6][0] = 1.0f;
    t.mp3_is_ratios[6][1] = 0.0f;
    const double is_scale[2] = {1.0 / std::sqrt(1.41421356237309504880168872420969808), 0.707106781186547524400844362104849039};
    for (int k = 0; k < 2; ++k)
        for (int is_pos = 0; is_pos < 32; ++is_pos) {
            float *r = t.mp3_is_ratios[7 + 32 * k + is_pos];
            if (is_pos & 1) {
                r[0] = (float)::pow(is_scale[k], (double)(is_pos + 1) / 2.0);
                r[1] = 1.0f;
            } else {
                r[0] = 1.0f;
                r[1] = (float)::pow(is_scale[k], (double)is_pos / 2.0);
            }
        }
    for (int i = 0; i < 8207; ++i) t.mp3_pow43[i] = ::powf((float)i, 4.0f / 3.0f);
    for (int i = 0; i < kMp3Pow2abLen; ++i) t.mp3_pow2ab[i] = (float)::pow(2.0, 0.25 * (double)(kMp3Pow2abMinE + i));
    for (int i = 0; i < 256; ++i) std::memcpy(&t.vorbis_floor1_db[i], &SYM_VORBIS_FLOOR1_DB_BITS[i], 4);
}

}  // namespace

// mdct.rs:45-54
void make_imdct_twiddles(int n, double scale, cpx *dst) {
    const int n2 = n / 2;
    const double alpha = 1.0 / 8.0 + (std::signbit(scale) ? (double)n2 : 0.0);
    const double pi_n = kPi / (double)n;
    const double sqrt_scale = std::sqrt(std::fabs(scale));
    for (int k = 0; k < n2; ++k) {
        const double theta = pi_n * (alpha + (double)k);
        dst[k].re = (float)(sqrt_scale * std::cos(theta));
        dst[k].im = (float)(sqrt_scale * std::sin(theta));
    }
}

// no_simd.rs:16-36
void make_fft_twiddles(int n, cpx *dst) {
    const int half = n >> 1;
    const double theta = kPi / (double)half;
    for (int k = 0; k < half; ++k) {
        const double angle = theta * (double)k;
        dst[k].re = (float)std::cos(angle);
        dst[k].im = (float)(-std::sin(angle));
    }
}

// symphonia-codec-vorbis/src/window.rs:11-24
void make_vorbis_window(int bs, float *dst) {
    const int len = bs / 2;
    const double denom = (double)len;
    const double half_pi = 1.57079632679489661923132169163975144;
    for (int i = 0; i < len; ++i) {
        const double frac = half_pi * (((double)i + 0.5) / denom);
        const double s = std::sin(frac);
        dst[i] = (float)std::sin(half_pi * (s * s));
    }
}

const HostTables &host_tables() {
    static HostTables tables;
    static std::once_flag once;
    std::call_once(once, [] { build(tables); });
    return tables;
}

}  // namespace symaccel
