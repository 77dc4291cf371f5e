// Host-side stages of the decode path that depend on the C library's transcendental functions and are far too rare /
// too small for a kernel: the AAC pulse tool and the Vorbis floor-0 curve.  The reference evaluates them with Rust's
// f32::powf / cos / exp and f64::atan, which lower to libm calls -- the same glibc this library links -- so calling
// them here, with the reference's operation order, reproduces its values; a device `powf` would not.  They run on the
// HOST, on host buffers, before the spectra / floor curves are handed to the device entry points; they are part of the
// decode front end that stays on the CPU (like the entropy decoder), not a fallback for any kernel.
//
//   Pulse::synth, iquant, requant       symphonia-codec-aac/src/aac/ics/pulse.rs:19-33, 64-105
//   bark, bark_map                      symphonia-codec-vorbis/src/floor.rs:353-376
//   Floor0::read_channel (2 cos step)   floor.rs:246-248
//   Floor0::synthesis                   floor.rs:262-340
//   linear_floor0_value                 floor.rs:379-390
#include <cmath>
#include <cstdint>
#include <limits>

#include "../../include/symaccel.h"

namespace {

inline float iquant(float val) {  // pulse.rs:19-21: val^(4/3) with the sign carried; the exponent is an f32 quotient
    const float e = 4.0f / 3.0f;
    return val < 0.0f ? -std::pow(-val, e) : std::pow(val, e);
}

inline float requant(float val, float scale) {  // pulse.rs:24-33 -- note: raises `val`, not val / scale
    if (scale == 0.0f) return 0.0f;
    const float bval = val / scale;
    const float e = 3.0f / 4.0f;
    return bval >= 0.0f ? std::pow(val, e) : -std::pow(-val, e);
}

inline double bark(double x) {  // floor.rs:353-356
    return (13.1 * std::atan(0.00074 * x)) + (2.24 * std::atan(0.0000000185 * x * x)) + (0.0001 * x);
}

}  // namespace

extern "C" {

int symaccel_host_aac_pulse(float *h_coeffs, size_t n_frames, const symaccel_aac_pulse *h_pulse, size_t n_pulse,
                            const uint16_t *swb_long, int n_swb_long) {
    if (n_pulse == 0) return SYMACCEL_OK;
    if (!h_coeffs || !h_pulse || !swb_long || n_swb_long < 1 || n_swb_long > 64) return SYMACCEL_ERR_INVALID_ARG;
    for (size_t p = 0; p < n_pulse; ++p) {
        const symaccel_aac_pulse &pl = h_pulse[p];
        if (pl.frame >= n_frames || pl.number_pulse < 1 || pl.number_pulse > 4) return SYMACCEL_ERR_INVALID_ARG;
    }
    for (size_t p = 0; p < n_pulse; ++p) {
        const symaccel_aac_pulse &pl = h_pulse[p];
        float *coeffs = h_coeffs + (size_t)pl.frame * 1024;
        if ((int)pl.pulse_start_sfb >= n_swb_long) continue;  // pulse.rs:70-72 (bands.len() - 1 == n_swb)
        unsigned k = swb_long[pl.pulse_start_sfb];
        int band = pl.pulse_start_sfb;
        for (unsigned pno = 0; pno < pl.number_pulse; ++pno) {
            k += pl.pulse_offset[pno];
            if (k >= 1024) break;
            while (swb_long[band + 1] <= k) ++band;
            const float scale = pl.scales0[band];
            float base = coeffs[k];
            if (base != 0.0f) base = requant(coeffs[k], scale);
            if (base > 0.0f)
                base += (float)pl.pulse_amp[pno];
            else
                base -= (float)pl.pulse_amp[pno];
            coeffs[k] = iquant(base) * scale;
        }
    }
    return SYMACCEL_OK;
}

int symaccel_host_vorbis_bark_map(uint32_t n, uint16_t floor0_rate, uint16_t floor0_bark_map_size, int32_t *h_map) {
    if (n == 0) return SYMACCEL_OK;
    if (!h_map || n > 4096u) return SYMACCEL_ERR_INVALID_ARG;
    const int32_t foobar_min = (int32_t)floor0_bark_map_size - 1;
    const double rate = (double)floor0_rate;
    const double rate_by_2n = rate / (2.0 * (double)n);
    const double c = (double)floor0_bark_map_size / bark(0.5 * rate);
    for (uint32_t i = 0; i < n; ++i) {
        const double v = std::floor(bark(rate_by_2n * (double)i) * c);
        int32_t foobar;  // `as i32`: saturating, NaN -> 0
        if (!(v == v))
            foobar = 0;
        else if (v >= 2147483647.0)
            foobar = std::numeric_limits<int32_t>::max();
        else if (v <= -2147483648.0)
            foobar = std::numeric_limits<int32_t>::min();
        else
            foobar = (int32_t)v;
        h_map[i] = foobar < foobar_min ? foobar : foobar_min;
    }
    return SYMACCEL_OK;
}

int symaccel_host_vorbis_floor0_coeffs(float *h_coeffs, int order) {
    if (order < 0 || order > 255 || (order > 0 && !h_coeffs)) return SYMACCEL_ERR_INVALID_ARG;
    for (int i = 0; i < order; ++i) h_coeffs[i] = 2.0f * std::cos(h_coeffs[i]);
    return SYMACCEL_OK;
}

int symaccel_host_vorbis_floor0(const float *h_coeffs, int order, const int32_t *h_map, uint32_t n,
                                uint16_t floor0_bark_map_size, uint8_t amplitude_bits, uint8_t amplitude_offset,
                                uint64_t amplitude, float *h_floor) {
    if (n == 0) return SYMACCEL_OK;
    if (!h_coeffs || !h_map || !h_floor || order < 0 || order > 255 || amplitude_bits > 63) return SYMACCEL_ERR_INVALID_ARG;
    const float pi_f = 3.14159265358979323846264338327950288f;
    const float omega_step = pi_f / (float)floor0_bark_map_size;
    uint32_t i = 0;
    for (;;) {
        const int32_t iter_cond = h_map[i];
        const float omega = omega_step * (float)iter_cond;
        const float cos_omega = std::cos(omega);
        const float two_cos_omega = 2.0f * cos_omega;
        float p = 1.0f, q = 1.0f;
        int j = 0;
        for (; j + 1 < order; j += 2) {  // chunks_exact(2)
            p *= h_coeffs[j + 1] - two_cos_omega;
            q *= h_coeffs[j] - two_cos_omega;
        }
        if (j < order) {  // the remainder of an odd order
            q *= h_coeffs[j] - two_cos_omega;
            p = p * p * (1.0f - (cos_omega * cos_omega));
            q = q * q * 0.25f;
        } else {
            p = p * p * ((1.0f - cos_omega) / 2.0f);
            q = q * q * ((1.0f + cos_omega) / 2.0f);
        }
        if (p + q == 0.0f) return SYMACCEL_ERR_DECODE;  // decode_error("vorbis: invalid floor0 coefficients")
        const float a = (float)(amplitude * (uint64_t)amplitude_offset);  // wrapping_mul, `as f32`
        const float b = std::sqrt(p + q) * (float)((UINT64_C(1) << amplitude_bits) - 1);
        const float value = std::exp(0.11512925f * ((a / b) - (float)amplitude_offset));
        while (i < n && h_map[i] == iter_cond) h_floor[i++] = value;
        if (i >= n) break;
    }
    return SYMACCEL_OK;
}

}  // extern "C"
