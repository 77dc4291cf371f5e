/*
 * symoracle.c -- CPU restatement of Symphonia's DSP hot path.
 * TEST INFRASTRUCTURE ONLY: see symoracle.h.  Build: see oracle/Makefile
 * (-O2 -ffp-contract=off -fno-fast-math: every a*b+c below is two roundings).
 */
#include "symoracle.h"

#include <math.h>
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "spec_tables.h"

#ifndef M_PI
#define M_PI 3.14159265358979323846
#endif

typedef struct {
    float re, im;
} cpx;

static inline cpx c_add(cpx a, cpx b) { cpx r = {a.re + b.re, a.im + b.im}; return r; }
static inline cpx c_sub(cpx a, cpx b) { cpx r = {a.re - b.re, a.im - b.im}; return r; }
/* num-complex 0.4 `Mul`: (a.re*b.re - a.im*b.im, a.re*b.im + a.im*b.re). */
static inline cpx c_mul(cpx a, cpx b)
{
    cpx r = {a.re * b.re - a.im * b.im, a.re * b.im + a.im * b.re};
    return r;
}

/* ======================================================================== */
/* FFT: symphonia-core/src/dsp/fft/no_simd.rs                                */
/* ======================================================================== */

/* f32 FRAC_1_SQRT_2 (no_simd.rs:302, 368, 412). */
#define SO_FRAC_1_SQRT_2 0.70710678118654752440f

/*
 * fft2..fft32 (no_simd.rs:289-454) written once as the recursion those five
 * unrolled bodies spell out: transform both halves, twiddle the upper half,
 * then x[k] = x0[k] + x1p[k], x[k+h] = x0[k] - x1p[k].  The twiddle of element
 * k of an n-point stage takes the same strength-reduced form the reference
 * uses: k == 0 none; k == n/4 -> (im, -re); k == n/8 -> (a+b, b-a) with
 * a = c*re, b = c*im; k == 3n/8 -> (a-b, a+b) with a = -c*re, b = -c*im;
 * otherwise a full complex multiply by the literal (cos, -sin).  The literals
 * in the reference equal (float)cos(2*pi*k/n), (float)(-sin(2*pi*k/n))
 * bit-for-bit (checked by tests/test_oracle_vs_reference_literals.py).
 */
static cpx g_small_tw[33][16];  /* [n][k], n in {16, 32} */
static cpx *g_merge_tw[17];     /* index log2(n), n = 64 .. 65536 */
static pthread_once_t g_fft_once = PTHREAD_ONCE_INIT;
static void fft_merge_twiddles(int n, cpx *w);

static void fft_tables_init(void)
{
    for (int n = 16; n <= 32; n <<= 1)
        for (int k = 0; k < n / 2; k++) {
            g_small_tw[n][k].re = (float)cos(2.0 * M_PI * k / n);
            g_small_tw[n][k].im = (float)(-sin(2.0 * M_PI * k / n));
        }
    for (int lg = 6; lg <= 16; lg++) {
        int n = 1 << lg;
        g_merge_tw[lg] = (cpx *)malloc(sizeof(cpx) * (size_t)(n / 2));
        fft_merge_twiddles(n, g_merge_tw[lg]);
    }
}

static void fft_small(cpx *x, int n)
{
    if (n == 1)
        return;
    if (n == 2) {
        cpx x0 = x[0];
        x[0] = c_add(x0, x[1]);
        x[1] = c_sub(x0, x[1]);
        return;
    }
    int h = n / 2;
    fft_small(x, h);
    fft_small(x + h, h);
    for (int k = 0; k < h; k++) {
        cpx v = x[h + k], q;
        if (k == 0) {
            q = v;
        } else if (4 * k == n) {
            q.re = v.im;
            q.im = -v.re;
        } else if (8 * k == n) {
            float a = SO_FRAC_1_SQRT_2 * v.re, b = SO_FRAC_1_SQRT_2 * v.im;
            q.re = a + b;
            q.im = b - a;
        } else if (8 * k == 3 * n) {
            float a = -SO_FRAC_1_SQRT_2 * v.re, b = -SO_FRAC_1_SQRT_2 * v.im;
            q.re = a - b;
            q.im = a + b;
        } else {
            q = c_mul(g_small_tw[n][k], v);
        }
        cpx e = x[k];
        x[k] = c_add(e, q);
        x[h + k] = c_sub(e, q);
    }
}

/* fft_twiddle_table! (no_simd.rs:16-36): W[k] = (cos(theta k), -sin(theta k)),
 * theta = pi / (n/2), f64 math, cast to f32. */
static void fft_merge_twiddles(int n, cpx *w)
{
    int half = n >> 1;
    double theta = M_PI / (double)half;
    for (int k = 0; k < half; k++) {
        double angle = theta * (double)k;
        w[k].re = (float)cos(angle);
        w[k].im = (float)(-sin(angle));
    }
}

void so_fft_twiddles(int n, float *tw_out) { fft_merge_twiddles(n, (cpx *)tw_out); }

/* literal twiddles of fft16 / fft32 (no_simd.rs:307-324, 374-383), k < n/2 */
void so_fft_small_twiddles(int n, float *tw_out)
{
    pthread_once(&g_fft_once, fft_tables_init);
    memcpy(tw_out, g_small_tw[n], sizeof(cpx) * (size_t)(n / 2));
}

/* transform (no_simd.rs:221-281): fft32 on every 32-chunk, then merge passes
 * step = 32, 64, ... n/2 with q = o*w; e' = e+q; o' = e-q. */
static void fft_transform(cpx *x, int n)
{
    pthread_once(&g_fft_once, fft_tables_init);
    if (n <= 32) {
        fft_small(x, n);
        return;
    }
    for (int c = 0; c < n; c += 32)
        fft_small(x + c, 32);
    int lg = 6;
    for (int step = 32; step < n; step <<= 1, lg++) {
        const cpx *w = g_merge_tw[lg];
        for (int base = 0; base < n; base += step << 1) {
            cpx *even = x + base, *odd = x + base + step;
            for (int k = 0; k < step; k++) {
                cpx p = even[k];
                cpx q = c_mul(odd[k], w[k]);
                even[k] = c_add(p, q);
                odd[k] = c_sub(p, q);
            }
        }
    }
}

/* perm[i] = reverse_bits(i as u16) >> (leading_zeros(n as u16) + 1) (no_simd.rs:83-85) */
static unsigned bitrev(unsigned i, int n)
{
    unsigned r = 0;
    for (int m = n >> 1; m > 0; m >>= 1) {
        r = (r << 1) | (i & 1u);
        i >>= 1;
    }
    return r;
}

void so_fft_inplace(float *xf, int n)
{
    cpx *x = (cpx *)xf;
    /* swap-based permutation (no_simd.rs:101-107) */
    for (int i = 0; i < n; i++) {
        int j = (int)bitrev((unsigned)i, n);
        if (i < j) {
            cpx t = x[i];
            x[i] = x[j];
            x[j] = t;
        }
    }
    fft_transform(x, n);
}

void so_fft(const float *xf, float *yf, int n)
{
    const cpx *x = (const cpx *)xf;
    cpx *y = (cpx *)yf;
    for (int i = 0; i < n; i++)
        y[i] = x[bitrev((unsigned)i, n)];
    fft_transform(y, n);
}

/* Ifft::ifft / ifft_inplace (no_simd.rs:143-219): y[i] = swap(x[perm[i]]); transform(y, n); y[i] = (c * y[i].im,
 * c * y[i].re) with c = 1.0 / n as f32.  NOTE transform() itself (no_simd.rs:221-281) handles exactly 32 points and
 * >= 64 points; called with fewer than 32 -- which only Ifft does, Fft::fft dispatching fft2..fft16 before it -- its
 * chunks_exact_mut(64) loop has nothing to iterate over: an Ifft of 2..16 points permutes, swaps and scales, nothing more.
 * The in-place form swaps while permuting (i <= j) and gives the same values. */
void so_ifft(const float *xf, float *yf, int n)
{
    const cpx *x = (const cpx *)xf;
    cpx *y = (cpx *)yf;
    for (int i = 0; i < n; i++) {
        cpx v = x[bitrev((unsigned)i, n)];
        y[i].re = v.im;
        y[i].im = v.re;
    }
    if (n >= 32)
        fft_transform(y, n);
    const float c = 1.0f / (float)n;
    for (int i = 0; i < n; i++) {
        cpx v = y[i];
        y[i].re = c * v.im;
        y[i].im = c * v.re;
    }
}


/* ======================================================================== */
/* IMDCT: symphonia-core/src/dsp/mdct.rs                                     */
/* ======================================================================== */

struct so_imdct {
    int n;
    cpx *twiddle;
    cpx *scratch;
};

so_imdct *so_imdct_new(int n, double scale)
{
    so_imdct *m = (so_imdct *)malloc(sizeof(*m));
    int n2 = n / 2;
    m->n = n;
    m->twiddle = (cpx *)malloc(sizeof(cpx) * (size_t)(n2 > 0 ? n2 : 1));
    m->scratch = (cpx *)malloc(sizeof(cpx) * (size_t)(n2 > 0 ? n2 : 1));
    /* mdct.rs:45-54 */
    double alpha = 1.0 / 8.0 + (signbit(scale) ? (double)n2 : 0.0);
    double pi_n = M_PI / (double)n;
    double sqrt_scale = sqrt(fabs(scale));
    for (int k = 0; k < n2; k++) {
        double theta = pi_n * (alpha + (double)k);
        double re = sqrt_scale * cos(theta);
        double im = sqrt_scale * sin(theta);
        m->twiddle[k].re = (float)re;
        m->twiddle[k].im = (float)im;
    }
    return m;
}

void so_imdct_free(so_imdct *m)
{
    if (!m)
        return;
    free(m->twiddle);
    free(m->scratch);
    free(m);
}

void so_imdct_twiddles(int n, double scale, float *tw_out)
{
    so_imdct *m = so_imdct_new(n, scale);
    memcpy(tw_out, m->twiddle, sizeof(cpx) * (size_t)(n / 2));
    so_imdct_free(m);
}

void so_imdct_run(so_imdct *m, const float *spec, float *out)
{
    int n = m->n, n2 = n >> 1, n4 = n >> 2;
    cpx *z = m->scratch;
    const cpx *tw = m->twiddle;

    /* mdct.rs:81-88 */
    for (int i = 0; i < n2; i++) {
        cpx w = tw[i];
        float even = spec[i * 2];
        float odd = -spec[n - 1 - i * 2];
        z[i].re = odd * w.im - even * w.re;
        z[i].im = odd * w.re + even * w.im;
    }

    so_fft_inplace((float *)z, n2); /* mdct.rs:91 */

    float *vec0 = out, *vec1 = out + n2, *vec2 = out + 2 * n2, *vec3 = out + 3 * n2;

    /* mdct.rs:101-118: val = w * x.conj() */
    for (int i = 0; i < n4; i++) {
        cpx x = z[i], w = tw[i];
        cpx xc = {x.re, -x.im};
        cpx val = c_mul(w, xc);
        int fi = 2 * i, ri = n2 - 1 - 2 * i;
        vec0[ri] = -val.im;
        vec1[fi] = val.im;
        vec2[ri] = val.re;
        vec3[fi] = val.re;
    }
    /* mdct.rs:120-137 */
    for (int i = 0; i < n4; i++) {
        cpx x = z[n4 + i], w = tw[n4 + i];
        cpx xc = {x.re, -x.im};
        cpx val = c_mul(w, xc);
        int fi = 2 * i, ri = n2 - 1 - 2 * i;
        vec0[fi] = -val.re;
        vec1[ri] = val.re;
        vec2[fi] = val.im;
        vec3[ri] = val.im;
    }
}

void so_imdct_batch(int n, double scale, const float *spec, float *out, size_t count)
{
    so_imdct *m = so_imdct_new(n, scale);
    for (size_t i = 0; i < count; i++)
        so_imdct_run(m, spec + i * (size_t)n, out + i * 2 * (size_t)n);
    so_imdct_free(m);
}

/* ======================================================================== */
/* AAC: symphonia-codec-aac/src/aac/{window,dsp}.rs                          */
/* ======================================================================== */

/* bessel_i0 (window.rs:56-63) */
static double aac_bessel_i0(double inval)
{
    double val = 1.0;
    for (int n = 63; n >= 1; n--) {
        val *= inval / (double)(n * n);
        val += 1.0;
    }
    return val;
}

/* generate_window with half = true, scale = 1.0 (window.rs:28-52; call sites dsp.rs:37-43) */
void so_aac_window(int kbd, float alpha, int size, float *dst)
{
    const float pi_f = 3.14159265358979323846264338327950288f; /* f32::consts::PI */
    if (!kbd) {
        float param = pi_f / (float)(2 * size);
        for (int n = 0; n < size; n++)
            dst[n] = sinf(((float)n + 0.5f) * param) * 1.0f;
    } else {
        float dlen = (float)size;
        float t = alpha * pi_f / dlen;
        double alpha2 = (double)(t * t);
        double *kb = (double *)malloc(sizeof(double) * (size_t)size);
        double sum = 0.0;
        for (int n = 0; n < size; n++) {
            double b = aac_bessel_i0((double)((long)n * (long)(size - n)) * alpha2);
            sum += b;
            kb[n] = sum;
        }
        sum += 1.0;
        for (int n = 0; n < size; n++)
            dst[n] = (float)sqrt(kb[n] / sum);
        free(kb);
    }
}

#define AAC_ONLY_LONG 0
#define AAC_LONG_START 1
#define AAC_EIGHT_SHORT 2
#define AAC_LONG_STOP 3
#define AAC_P0 (512 - 64) /* SHORT_WIN_POINT0, dsp.rs:19 */
#define AAC_P1 (512 + 64) /* SHORT_WIN_POINT1, dsp.rs:20 */

typedef struct {
    int ready;
    float kbd_long[1024], kbd_short[128], sine_long[1024], sine_short[128];
    so_imdct *imdct_long, *imdct_short;
} aac_dsp;

static aac_dsp g_aac;

/* Dsp::new (dsp.rs:34-54) */
static pthread_once_t g_aac_once = PTHREAD_ONCE_INIT;
static void aac_init(void)
{
    {
        so_aac_window(1, 4.0f, 1024, g_aac.kbd_long);
        so_aac_window(1, 6.0f, 128, g_aac.kbd_short);
        so_aac_window(0, 0.0f, 1024, g_aac.sine_long);
        so_aac_window(0, 0.0f, 128, g_aac.sine_short);
        g_aac.imdct_long = so_imdct_new(1024, 1.0 / 2048.0);
        g_aac.imdct_short = so_imdct_new(128, 1.0 / 256.0);
        g_aac.ready = 1;
    }
}
static aac_dsp *aac_get(void)
{
    pthread_once(&g_aac_once, aac_init);
    return &g_aac;
}

/* Dsp::synth (dsp.rs:57-158) */
static void aac_synth_with(aac_dsp *d, so_imdct *il, so_imdct *is, const float *coeffs,
                           float *delay, int seq, int window_shape, int prev_window_shape,
                           float *dst)
{
    float pcm_long[2048], pcm_short[1152];
    const float *long_win = window_shape ? d->kbd_long : d->sine_long;
    const float *short_win = window_shape ? d->kbd_short : d->sine_short;
    const float *prev_long_win = prev_window_shape ? d->kbd_long : d->sine_long;
    const float *prev_short_win = prev_window_shape ? d->kbd_short : d->sine_short;

    if (seq != AAC_EIGHT_SHORT) {
        so_imdct_run(il, coeffs, pcm_long);
    } else {
        for (int w = 0; w < 8; w++)
            so_imdct_run(is, coeffs + 128 * w, pcm_long + 256 * w);
        for (int i = 0; i < 1152; i++)
            pcm_short[i] = 0.0f;
        for (int w = 0; w < 8; w++) {
            const float *src = pcm_long + 256 * w;
            if (w > 0) {
                for (int i = 0; i < 128; i++) {
                    pcm_short[w * 128 + i] += src[i] * short_win[i];
                    pcm_short[w * 128 + i + 128] += src[i + 128] * short_win[127 - i];
                }
            } else {
                for (int i = 0; i < 128; i++) {
                    pcm_short[i] = src[i] * prev_short_win[i];
                    pcm_short[i + 128] = src[i + 128] * short_win[127 - i];
                }
            }
        }
    }

    switch (seq) {
    case AAC_ONLY_LONG:
    case AAC_LONG_START:
        for (int i = 0; i < 1024; i++)
            dst[i] = delay[i] + (pcm_long[i] * prev_long_win[i]);
        break;
    case AAC_EIGHT_SHORT:
        for (int i = 0; i < AAC_P0; i++)
            dst[i] = delay[i];
        for (int i = AAC_P0; i < 1024; i++)
            dst[i] = delay[i] + pcm_short[i - AAC_P0];
        break;
    default: /* LONG_STOP */
        for (int i = 0; i < AAC_P0; i++)
            dst[i] = delay[i];
        for (int i = AAC_P0; i < AAC_P1; i++)
            dst[i] = delay[i] + pcm_long[i] * prev_short_win[i - AAC_P0];
        for (int i = AAC_P1; i < 1024; i++)
            dst[i] = delay[i] + pcm_long[i];
        break;
    }

    switch (seq) {
    case AAC_ONLY_LONG:
    case AAC_LONG_STOP:
        for (int i = 0; i < 1024; i++)
            delay[i] = pcm_long[i + 1024] * long_win[1023 - i];
        break;
    case AAC_EIGHT_SHORT:
        for (int i = 0; i < AAC_P1; i++)
            delay[i] = pcm_short[i + 512 + 64];
        for (int i = AAC_P1; i < 1024; i++)
            delay[i] = 0.0f;
        break;
    default: /* LONG_START */
        for (int i = 0; i < AAC_P0; i++)
            delay[i] = pcm_long[1024 + i];
        for (int i = AAC_P0; i < AAC_P1; i++)
            delay[i] = pcm_long[i + 1024] * short_win[127 - (i - AAC_P0)];
        for (int i = AAC_P1; i < 1024; i++)
            delay[i] = 0.0f;
        break;
    }
}

void so_aac_synth(const float *coeffs, float *delay, int seq, int window_shape,
                  int prev_window_shape, float *dst)
{
    aac_dsp *d = aac_get();
    aac_synth_with(d, d->imdct_long, d->imdct_short, coeffs, delay, seq, window_shape,
                   prev_window_shape, dst);
}

void so_aac_synth_batch(const float *coeffs, const uint8_t *side, float *delay, float *pcm,
                        size_t n_chains, size_t frames_per_chain)
{
    aac_dsp *d = aac_get();
    /* private Imdct scratch so concurrent callers (cpu_baseline threads) do not share it */
    so_imdct *il = so_imdct_new(1024, 1.0 / 2048.0), *is = so_imdct_new(128, 1.0 / 256.0);
    for (size_t c = 0; c < n_chains; c++) {
        for (size_t f = 0; f < frames_per_chain; f++) {
            size_t idx = c * frames_per_chain + f;
            uint8_t s = side[idx];
            aac_synth_with(d, il, is, coeffs + idx * 1024, delay + c * 1024, s & 3, (s >> 2) & 1,
                           (s >> 3) & 1, pcm + idx * 1024);
        }
    }
    so_imdct_free(il);
    so_imdct_free(is);
}

/* ---- AAC spectral tools in front of Dsp::synth ---------- */

/* Joint-stereo decoding of one channel pair (aac/cpe.rs:110-157).  bands: the swb offsets of the window length in
 * use (ICS get_bands), mode/scale per [w * 16 + sfb] for eight short windows or [sfb] for one long window:
 * mode 2 = intensity (right = scale * left, scale = dir * factor * scales[g][sfb] as cpe.rs:131 forms it),
 * mode 1 = mid/side, 0 = neither (incl. the noise-substitution bands cpe.rs:140-143 skips). */
void so_aac_joint_stereo(float *left, float *right, int num_windows, int max_sfb, const uint16_t *bands,
                         const uint8_t *mode, const float *scale)
{
    for (int w = 0; w < num_windows; w++) {
        for (int sfb = 0; sfb < max_sfb; sfb++) {
            const int start = w * 128 + bands[sfb], end = w * 128 + bands[sfb + 1];
            const int slot = num_windows == 1 ? sfb : w * 16 + sfb;
            if (mode[slot] == 2) {
                for (int i = start; i < end; i++)
                    right[i] = scale[slot] * left[i];
            } else if (mode[slot] == 1) {
                for (int i = start; i < end; i++) {
                    const float tmp = left[i] - right[i];
                    left[i] += right[i];
                    right[i] = tmp;
                }
            }
        }
    }
}

/* One TNS filter of Tns::synth (aac/ics/tns.rs:180-195) over coeffs[start..end): the all-pole filter runs up
 * (direction 0) or down the spectrum and only reaches back to samples inside its own range (j < min(order, m)). */
void so_aac_tns_filter(float *coeffs, int start, int end, int order, int direction, const float *lpc)
{
    if (!direction) {
        for (int m = 0, i = start; i < end; i++, m++) {
            const int lim = order < m ? order : m;
            for (int j = 0; j < lim; j++)
                coeffs[i] -= coeffs[i - j - 1] * lpc[j];
        }
    } else {
        for (int m = 0, i = end - 1; i >= start; i--, m++) {
            const int lim = order < m ? order : m;
            for (int j = 0; j < lim; j++)
                coeffs[i] -= coeffs[i + j + 1] * lpc[j];
        }
    }
}

/* Pulse::synth with iquant / requant (symphonia-codec-aac/src/aac/ics/pulse.rs:19-33, 64-105).  f32 `powf` is the C
 * library's powf, as it is for the reference (Rust's f32::powf lowers to the same libm call).  `requant` really does
 * raise `val`, not `val / scale`, to the 3/4 power (pulse.rs:27-33: bval only decides the sign branch); the exponents
 * 4.0 / 3.0 and 3.0 / 4.0 are f32 divisions. */
static float aac_iquant(float val)
{
    const float e = 4.0f / 3.0f;
    return val < 0.0f ? -powf(-val, e) : powf(val, e);
}
static float aac_requant(float val, float scale)
{
    if (scale == 0.0f)
        return 0.0f;
    const float bval = val / scale;
    const float e = 3.0f / 4.0f;
    return bval >= 0.0f ? powf(val, e) : -powf(-val, e);
}
void so_aac_iquant_requant(const float *val, float scale, float *iq, float *rq, size_t n)
{
    for (size_t i = 0; i < n; i++) {
        iq[i] = aac_iquant(val[i]);
        rq[i] = aac_requant(val[i], scale);
    }
}
void so_aac_pulse(float *coeffs, const int32_t *bands, int n_bands_plus_1, const float *scales0, int number_pulse,
                  int pulse_start_sfb, const int32_t *pulse_offset, const int32_t *pulse_amp)
{
    if (pulse_start_sfb >= n_bands_plus_1 - 1) /* pulse.rs:70-72 */
        return;
    int k = bands[pulse_start_sfb];
    int band = pulse_start_sfb;
    for (int pno = 0; pno < number_pulse; pno++) {
        k += pulse_offset[pno];
        if (k >= 1024)
            return;
        while (bands[band + 1] <= k)
            band++;
        const float scale = scales0[band];
        float base = coeffs[k];
        if (base != 0.0f)
            base = aac_requant(coeffs[k], scale);
        if (base > 0.0f)
            base += (float)pulse_amp[pno];
        else
            base -= (float)pulse_amp[pno];
        coeffs[k] = aac_iquant(base) * scale;
    }
}

/* ======================================================================== */
/* MP3: symphonia-bundle-mp3/src/layer3/hybrid_synthesis.rs, synthesis.rs    */
/* ======================================================================== */

typedef struct {
    int ready;
    float imdct_windows[4][36];
    float half_cos_12[6][6];
    float cs[8], ca[8];
    float dct_iv_scale[18], sdct18_scale[9], sdct9_d[7];
    float cos16[16], cos8[8], cos4[4], cos2[2], cos1;
    float synth_d[512];
    int sfb_short[9][40];
    int sfb_mixed[9][40];
    int sfb_mixed_len[9];
    int sfb_mixed_switch[9];
    int sfb_long[9][23];
    float pow43[8207];                 /* requantize.rs:28-31 */
    float pow2ab[SO_MP3_POW2AB_LEN];   /* 2^(0.25 e), e = SO_MP3_POW2AB_MIN_E .. (requantize.rs:280, 343) */
    float is_mpeg1[7][2];              /* INTENSITY_STEREO_RATIOS_MPEG1 (stereo.rs:83-118) */
    float is_mpeg2[2][32][2];          /* INTENSITY_STEREO_RATIOS_MPEG2 (stereo.rs:31-81) */
} mp3_tables;

static mp3_tables g_mp3;

/* Short scale-factor band widths per sample-rate index (ISO/IEC 11172-3 Table
 * B.8, 13818-3 Table B.2); SFB_SHORT_BANDS (layer3/common.rs:60-106) is the
 * running sum of each width taken three times. */
static const unsigned char MP3_SHORT_WIDTHS[9][13] = {
    {4, 4, 4, 4, 6, 8, 10, 12, 14, 18, 22, 30, 56},   /* 44.1k */
    {4, 4, 4, 4, 6, 6, 10, 12, 14, 16, 20, 26, 66},   /* 48k   */
    {4, 4, 4, 4, 6, 8, 12, 16, 20, 26, 34, 42, 12},   /* 32k   */
    {4, 4, 4, 6, 6, 8, 10, 14, 18, 26, 32, 42, 18},   /* 22.05k */
    {4, 4, 4, 6, 8, 10, 12, 14, 18, 24, 32, 44, 12},  /* 24k   */
    {4, 4, 4, 6, 8, 10, 12, 14, 18, 24, 30, 40, 18},  /* 16k   */
    {4, 4, 4, 6, 8, 10, 12, 14, 18, 24, 30, 40, 18},  /* 11.025k */
    {4, 4, 4, 6, 8, 10, 12, 14, 18, 24, 30, 40, 18},  /* 12k   */
    {8, 8, 8, 12, 16, 20, 24, 28, 36, 2, 2, 2, 26},   /* 8k    */
};
/* Long scale-factor band widths (ISO/IEC 11172-3 Table B.8, 13818-3 Table B.2; the 8 kHz row as the
 * reference has it); SFB_LONG_BANDS (layer3/common.rs:9-56) is their running sum. */
static const unsigned char MP3_LONG_WIDTHS[9][22] = {
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
/* Pre-emphasis (ISO/IEC 11172-3 Table B.6; requantize.rs:256-257) */
static const unsigned char MP3_PRE_EMPHASIS[22] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 3, 3, 3, 2, 0};
/* Long-band prefix of SFB_MIXED_BANDS (layer3/common.rs:108-168), up to and
 * including the boundary at 36; the short bands that follow start at 36. */
static const unsigned char MP3_MIXED_PREFIX[9][9] = {
    {0, 4, 8, 12, 16, 20, 24, 30, 36}, {0, 4, 8, 12, 16, 20, 24, 30, 36},
    {0, 4, 8, 12, 16, 20, 24, 30, 36}, {0, 6, 12, 18, 24, 30, 36, 0, 0},
    {0, 6, 12, 18, 24, 30, 36, 0, 0},  {0, 6, 12, 18, 24, 30, 36, 0, 0},
    {0, 6, 12, 18, 24, 30, 36, 0, 0},  {0, 6, 12, 18, 24, 30, 36, 0, 0},
    {0, 12, 24, 36, 0, 0, 0, 0, 0},
};
static const unsigned char MP3_MIXED_PREFIX_LEN[9] = {9, 9, 9, 7, 7, 7, 7, 7, 4};
/* 8 kHz tail after 36 (the reference's "educated guess", common.rs:160-167) */
static const unsigned short MP3_MIXED_8K_TAIL[] = {
    40,  44,  48,  56,  64,  72,  84,  96,  108, 124, 140, 156, 176, 196, 216, 240, 264, 288,
    316, 344, 372, 408, 444, 480, 482, 484, 486, 488, 490, 492, 494, 496, 498, 524, 550, 576};

static pthread_once_t g_mp3_once = PTHREAD_ONCE_INIT;
static void mp3_init(void)
{
    mp3_tables *t = &g_mp3;
    /* IMDCT_WINDOWS (hybrid_synthesis.rs:53-92) */
    const double PI_36 = M_PI / 36.0, PI_12 = M_PI / 12.0;
    memset(t->imdct_windows, 0, sizeof(t->imdct_windows));
    for (int i = 0; i < 36; i++)
        t->imdct_windows[0][i] = (float)sin(PI_36 * ((double)i + 0.5));
    for (int i = 0; i < 18; i++)
        t->imdct_windows[1][i] = (float)sin(PI_36 * ((double)i + 0.5));
    for (int i = 18; i < 24; i++)
        t->imdct_windows[1][i] = 1.0f;
    for (int i = 24; i < 30; i++)
        t->imdct_windows[1][i] = (float)sin(PI_12 * ((double)(i - 18) + 0.5));
    for (int i = 0; i < 12; i++)
        t->imdct_windows[2][i] = (float)sin(PI_12 * ((double)i + 0.5));
    for (int i = 6; i < 12; i++)
        t->imdct_windows[3][i] = (float)sin(PI_12 * ((double)(i - 6) + 0.5));
    for (int i = 12; i < 18; i++)
        t->imdct_windows[3][i] = 1.0f;
    for (int i = 18; i < 36; i++)
        t->imdct_windows[3][i] = (float)sin(PI_36 * ((double)i + 0.5));
    /* IMDCT_HALF_COS_12 (hybrid_synthesis.rs:105-119) */
    const double PI_24 = M_PI / 24.0;
    for (int i = 0; i < 6; i++)
        for (int k = 0; k < 6; k++) {
            int n = (2 * (i + 3) + (12 / 2) + 1) * (2 * k + 1);
            t->half_cos_12[i][k] = (float)cos(PI_24 * (double)n);
        }
    /* ANTIALIAS_CS_CA (hybrid_synthesis.rs:136-149) */
    static const double C[8] = {-0.6, -0.535, -0.33, -0.185, -0.095, -0.041, -0.0142, -0.0037};
    for (int i = 0; i < 8; i++) {
        double s = sqrt(1.0 + (C[i] * C[i]));
        t->cs[i] = (float)(1.0 / s);
        t->ca[i] = (float)(C[i] / s);
    }
    /* decimal literals of hybrid_synthesis.rs:611-630, 668-678, 722-730 and
     * synthesis.rs:354-396: all equal the f64 closed form rounded to f32
     * (verified against the reference literals by the local-only test). */
    for (int m = 0; m < 18; m++)
        t->dct_iv_scale[m] = (float)(2.0 * cos(M_PI * (2 * m + 1) / 72.0));
    for (int m = 0; m < 9; m++)
        t->sdct18_scale[m] = (float)(2.0 * cos(M_PI * (2 * m + 1) / 36.0));
    t->sdct18_scale[4] = 1.41421356237309504880168872420969808f; /* f32::consts::SQRT_2 */
    t->sdct9_d[0] = (float)(-sqrt(3.0));
    t->sdct9_d[1] = (float)(-2.0 * cos(8.0 * M_PI / 9.0));
    t->sdct9_d[2] = (float)(-2.0 * cos(4.0 * M_PI / 9.0));
    t->sdct9_d[3] = (float)(-2.0 * cos(2.0 * M_PI / 9.0));
    t->sdct9_d[4] = (float)(-2.0 * sin(8.0 * M_PI / 9.0));
    t->sdct9_d[5] = (float)(-2.0 * sin(4.0 * M_PI / 9.0));
    t->sdct9_d[6] = (float)(-2.0 * sin(2.0 * M_PI / 9.0));
    for (int i = 0; i < 16; i++)
        t->cos16[i] = (float)(1.0 / (2.0 * cos(M_PI * (2 * i + 1) / 64.0)));
    for (int i = 0; i < 8; i++)
        t->cos8[i] = (float)(1.0 / (2.0 * cos(M_PI * (2 * i + 1) / 32.0)));
    for (int i = 0; i < 4; i++)
        t->cos4[i] = (float)(1.0 / (2.0 * cos(M_PI * (2 * i + 1) / 16.0)));
    for (int i = 0; i < 2; i++)
        t->cos2[i] = (float)(1.0 / (2.0 * cos(M_PI * (2 * i + 1) / 8.0)));
    t->cos1 = 0.7071067811865475f;
    /* SYNTHESIS_D (synthesis.rs:13-142) from the Q16 numerators */
    for (int i = 0; i < 512; i++) {
        char lit[32];
        snprintf(lit, sizeof lit, "%.9f", (double)SYM_MP3_SYNTH_WINDOW_Q16[i] / 65536.0);
        t->synth_d[i] = strtof(lit, NULL);
    }
    /* scale-factor band tables (layer3/common.rs:60-172) */
    for (int sr = 0; sr < 9; sr++) {
        int acc = 0, n = 0;
        t->sfb_short[sr][n++] = 0;
        for (int b = 0; b < 13; b++)
            for (int w = 0; w < 3; w++) {
                acc += MP3_SHORT_WIDTHS[sr][b];
                t->sfb_short[sr][n++] = acc;
            }
        int m = 0;
        for (int i = 0; i < MP3_MIXED_PREFIX_LEN[sr]; i++)
            t->sfb_mixed[sr][m++] = MP3_MIXED_PREFIX[sr][i];
        if (sr == 8) {
            for (size_t i = 0; i < sizeof(MP3_MIXED_8K_TAIL) / sizeof(MP3_MIXED_8K_TAIL[0]); i++)
                t->sfb_mixed[sr][m++] = MP3_MIXED_8K_TAIL[i];
        } else {
            int k = 0;
            while (t->sfb_short[sr][k] != 36)
                k++;
            for (k = k + 1; k < 40; k++)
                t->sfb_mixed[sr][m++] = t->sfb_short[sr][k];
        }
        t->sfb_mixed_len[sr] = m;
        t->sfb_mixed_switch[sr] = MP3_MIXED_PREFIX_LEN[sr] - 1; /* SFB_MIXED_SWITCH_POINT */
        t->sfb_long[sr][0] = 0;
        for (int b = 0; b < 22; b++)
            t->sfb_long[sr][b + 1] = t->sfb_long[sr][b] + MP3_LONG_WIDTHS[sr][b];
    }
    /* POW43 (requantize.rs:28-31): f32::powf(i as f32, 4.0 / 3.0) */
    for (int i = 0; i < 8207; i++)
        t->pow43[i] = powf((float)i, 4.0f / 3.0f);
    /* f64::powf(2.0, 0.25 * f64::from(a - b)) as f32 (requantize.rs:280, 343) for every exponent reachable */
    for (int i = 0; i < SO_MP3_POW2AB_LEN; i++)
        t->pow2ab[i] = (float)pow(2.0, 0.25 * (double)(SO_MP3_POW2AB_MIN_E + i));
    /* stereo.rs:105-116: is_ratio = tan(is_pos PI/12); (is_ratio / (1 + is_ratio), 1 / (1 + is_ratio)); [6] = (1, 0) */
    for (int is_pos = 0; is_pos < 7; is_pos++) {
        const double is_ratio = tan((M_PI / 12.0) * (double)is_pos);
        t->is_mpeg1[is_pos][0] = (float)(is_ratio / (1.0 + is_ratio));
        t->is_mpeg1[is_pos][1] = (float)(1.0 / (1.0 + is_ratio));
    }
    t->is_mpeg1[6][0] = 1.0f;
    t->is_mpeg1[6][1] = 0.0f;
    /* stereo.rs:60-79: i0 = 1 / sqrt(SQRT_2) or FRAC_1_SQRT_2; odd: (i0^((is_pos + 1) / 2), 1), even: (1, i0^(is_pos / 2)) */
    const double is_scale[2] = {1.0 / sqrt(M_SQRT2), M_SQRT1_2};
    for (int k = 0; k < 2; k++)
        for (int is_pos = 0; is_pos < 32; is_pos++) {
            if (is_pos & 1) {
                t->is_mpeg2[k][is_pos][0] = (float)pow(is_scale[k], (double)(is_pos + 1) / 2.0);
                t->is_mpeg2[k][is_pos][1] = 1.0f;
            } else {
                t->is_mpeg2[k][is_pos][0] = 1.0f;
                t->is_mpeg2[k][is_pos][1] = (float)pow(is_scale[k], (double)is_pos / 2.0);
            }
        }
    t->ready = 1;
}
static mp3_tables *mp3_get(void)
{
    pthread_once(&g_mp3_once, mp3_init);
    return &g_mp3;
}

/* dct_iv SCALE[18] | sdct_ii_18 SCALE[9] | sdct_ii_9 D[7] | COS_16 | COS_8 | COS_4 | COS_2 | COS_1
 * | IMDCT_HALF_COS_12[36] | cs[8] | ca[8]  (117 floats) */
void so_mp3_constants(float *dst)
{
    mp3_tables *t = mp3_get();
    memcpy(dst, t->dct_iv_scale, 18 * 4);
    memcpy(dst + 18, t->sdct18_scale, 9 * 4);
    memcpy(dst + 27, t->sdct9_d, 7 * 4);
    memcpy(dst + 34, t->cos16, 16 * 4);
    memcpy(dst + 50, t->cos8, 8 * 4);
    memcpy(dst + 58, t->cos4, 4 * 4);
    memcpy(dst + 62, t->cos2, 2 * 4);
    dst[64] = t->cos1;
    memcpy(dst + 65, t->half_cos_12, 36 * 4);
    memcpy(dst + 101, t->cs, 8 * 4);
    memcpy(dst + 109, t->ca, 8 * 4);
}
/* SFB_SHORT_BANDS[sr][40] | SFB_MIXED_BANDS[sr] (padded to 40 with -1) | switch point */
void so_mp3_sfb_tables(int sr, int32_t *dst81)
{
    mp3_tables *t = mp3_get();
    for (int i = 0; i < 40; i++) {
        dst81[i] = t->sfb_short[sr][i];
        dst81[40 + i] = i < t->sfb_mixed_len[sr] ? t->sfb_mixed[sr][i] : -1;
    }
    dst81[80] = t->sfb_mixed_switch[sr];
}

/* SFB_LONG_BANDS[sr][23] (layer3/common.rs:9-56) */
void so_mp3_sfb_long(int sr, int32_t *dst23)
{
    for (int i = 0; i < 23; i++)
        dst23[i] = mp3_get()->sfb_long[sr][i];
}
void so_mp3_pow43(float *dst8207) { memcpy(dst8207, mp3_get()->pow43, 8207 * 4); }
void so_mp3_pow2ab(float *dst) { memcpy(dst, mp3_get()->pow2ab, SO_MP3_POW2AB_LEN * 4); }

/* requantize_long (requantize.rs:239-293) over the band edges bands[0..n_edges) */
static void mp3_requantize_long(const so_mp3_requant *ch, const int *bands, int n_edges, float *buf)
{
    mp3_tables *t = mp3_get();
    const int a = (int)ch->global_gain - 210;
    const int scalefac_shift = (ch->flags & SO_MP3_RQ_SCALEFAC_SCALE) ? 2 : 1;
    for (int i = 0; i + 1 < n_edges; i++) {
        const int start = bands[i], end = bands[i + 1];
        if (start >= (int)ch->rzero)
            break;
        const int pre = (ch->flags & SO_MP3_RQ_PREFLAG) ? MP3_PRE_EMPHASIS[i] : 0;
        const int b = ((int)ch->scalefacs[i] + pre) << scalefac_shift;
        const float pow2ab = t->pow2ab[(a - b) - SO_MP3_POW2AB_MIN_E];
        const int band_end = end < (int)ch->rzero ? end : (int)ch->rzero;
        for (int k = start; k < band_end; k++)
            buf[k] *= pow2ab;
    }
}
/* requantize_short (requantize.rs:296-353) */
static void mp3_requantize_short(const so_mp3_requant *ch, const int *bands, int n_edges, int sw, float *buf)
{
    mp3_tables *t = mp3_get();
    const int gain = (int)ch->global_gain - 210;
    const int a[3] = {gain - 8 * (int)ch->subblock_gain[0], gain - 8 * (int)ch->subblock_gain[1],
                      gain - 8 * (int)ch->subblock_gain[2]};
    const int scalefac_shift = (ch->flags & SO_MP3_RQ_SCALEFAC_SCALE) ? 2 : 1;
    for (int i = 0; i + 1 < n_edges; i++) {
        const int start = bands[i], end = bands[i + 1];
        if (start >= (int)ch->rzero)
            break;
        const int b = (int)ch->scalefacs[sw + i] << scalefac_shift;
        const float pow2ab = t->pow2ab[(a[i % 3] - b) - SO_MP3_POW2AB_MIN_E];
        const int win_end = end < (int)ch->rzero ? end : (int)ch->rzero;
        for (int k = start; k < win_end; k++)
            buf[k] *= pow2ab;
    }
}
/* The sample values read_huffman_samples leaves in buf (requantize.rs:117-147, 172-205, 234):
 * (1 - 2 sign) * POW43[|s|] for a non-zero quantised sample, 0.0 otherwise and from rzero on;
 * then requantize (requantize.rs:356-380). */
void so_mp3_requantize(const int16_t *is576, const so_mp3_requant *ch, int sr, float *xr576)
{
    mp3_tables *t = mp3_get();
    for (int i = 0; i < 576; i++) {
        const int s = is576[i];
        if (i >= (int)ch->rzero || s == 0)
            xr576[i] = 0.0f;
        else
            xr576[i] = (1.0f - 2.0f * (s < 0 ? 1.0f : 0.0f)) * t->pow43[s < 0 ? -s : s];
    }
    if (ch->block_type == SO_MP3_SHORT && !ch->is_mixed) {
        mp3_requantize_short(ch, t->sfb_short[sr], 40, 0, xr576);
    } else if (ch->block_type == SO_MP3_SHORT) {
        /* the mixed table is split at the switch point; the long part takes the edges bands[..switch], i.e. it
         * stops one band short of the first short band: the lines in between stay unscaled (requantize.rs:368-372) */
        const int sw = t->sfb_mixed_switch[sr];
        mp3_requantize_long(ch, t->sfb_mixed[sr], sw, xr576);
        mp3_requantize_short(ch, t->sfb_mixed[sr] + sw, t->sfb_mixed_len[sr] - sw, sw, xr576);
    } else {
        mp3_requantize_long(ch, t->sfb_long[sr], 23, xr576);
    }
}
void so_mp3_requantize_batch(const int16_t *is, const so_mp3_requant *ch, int sr, float *xr, size_t n)
{
    for (size_t g = 0; g < n; g++)
        so_mp3_requantize(is + g * 576, ch + g, sr, xr + g * 576);
}

/* ---- joint stereo (layer3/stereo.rs) ---------------------------------------------------------------------------- */

void so_mp3_intensity_ratios(float *mpeg1_14, float *mpeg2_128)
{
    memcpy(mpeg1_14, mp3_get()->is_mpeg1, sizeof(mp3_get()->is_mpeg1));
    memcpy(mpeg2_128, mp3_get()->is_mpeg2, sizeof(mp3_get()->is_mpeg2));
}
/* process_mid_side (stereo.rs:139-148) */
static void mp3_mid_side(float *mid, float *side, int n)
{
    for (int i = 0; i < n; i++) {
        const float left = (mid[i] + side[i]) * 0.70710678118654752440f;
        const float right = (mid[i] - side[i]) * 0.70710678118654752440f;
        mid[i] = left;
        side[i] = right;
    }
}
/* process_intensity (stereo.rs:165-186) */
static void mp3_intensity(int is_pos, const float (*table)[2], int is_max, int mid_side, float *ch0, float *ch1, int n)
{
    if (is_pos < is_max) {
        const float ratio_l = table[is_pos][0], ratio_r = table[is_pos][1];
        for (int i = 0; i < n; i++) {
            const float is = ch0[i];
            ch0[i] = ratio_l * is;
            ch1[i] = ratio_r * is;
        }
    } else if (mid_side) {
        mp3_mid_side(ch0, ch1, n);
    }
}
static int mp3_zero_band(const float *band, int n) /* is_zero_band (stereo.rs:189-192) */
{
    for (int i = 0; i < n; i++)
        if (band[i] != 0.0f)
            return 0;
    return 1;
}
/* stereo (stereo.rs:485-556) with process_intensity_long_block (:196-260) and process_intensity_short_block (:264-483).
 * The caller has checked that both channels carry the same block type (stereo.rs:502-504) and sets both channels'
 * rzero to max(rzero0, rzero1) afterwards (:549-553). */
void so_mp3_stereo(float *ch0, float *ch1, const so_mp3_stereo_desc *d, int sr)
{
    mp3_tables *t = mp3_get();
    const int mid_side = (d->flags & SO_MP3_ST_MID_SIDE) != 0, intensity = (d->flags & SO_MP3_ST_INTENSITY) != 0;
    if (!mid_side && !intensity)
        return;
    const int end = d->rzero0 > d->rzero1 ? d->rzero0 : d->rzero1;
    int bound = end;
    if (intensity) {
        const float (*table)[2];
        int inv_pos;
        if (d->flags & SO_MP3_ST_MPEG1) {
            table = t->is_mpeg1;
            inv_pos = 7;
        } else {
            table = t->is_mpeg2[(d->flags & SO_MP3_ST_IS_SCALE) ? 1 : 0];
            inv_pos = 31;
        }
        if (d->block_type == SO_MP3_SHORT) {
            const int *bands;
            int n_edges, sw = 0;
            if (d->is_mixed) {
                bands = t->sfb_mixed[sr];
                n_edges = t->sfb_mixed_len[sr];
                sw = t->sfb_mixed_switch[sr];
            } else {
                bands = t->sfb_short[sr];
                n_edges = 40;
            }
            int is_pos[39]; /* stereo.rs:369-371 */
            for (int i = 0; i < 36; i++)
                is_pos[i] = d->scalefacs1[i];
            for (int i = 0; i < 3; i++)
                is_pos[36 + i] = d->scalefacs1[33 + i];
            const int *sb = bands + sw;
            const int n_short = n_edges - sw;
            int sfi = d->is_mixed ? n_edges - 1 : 39;
            int window_is_zero[3] = {1, 1, 1}, found_bound = 0;
            /* groups of four consecutive edges at every third position, from the top (stereo.rs:379-386) */
            int n_groups = 0;
            for (int g = 0; g + 3 < n_short; g += 3)
                n_groups++;
            for (int gi = n_groups - 1; gi >= 0; gi--) {
                const int s[4] = {sb[3 * gi], sb[3 * gi + 1], sb[3 * gi + 2], sb[3 * gi + 3]};
                for (int w = 2; w >= 0; w--) { /* stereo.rs:392-448, windows 2, 1, 0 */
                    const int a = s[w], b = s[w + 1];
                    window_is_zero[w] = window_is_zero[w] && mp3_zero_band(ch1 + a, b - a);
                    if (window_is_zero[w])
                        mp3_intensity(is_pos[sfi - 1], table, inv_pos, mid_side, ch0 + a, ch1 + a, b - a);
                    else if (mid_side)
                        mp3_mid_side(ch0 + a, ch1 + a, b - a);
                    sfi--;
                }
                bound = s[0];
                found_bound = !window_is_zero[0] && !window_is_zero[1] && !window_is_zero[2];
                if (found_bound)
                    break;
            }
            if (!found_bound && d->is_mixed) { /* the long bands bands[..switch + 1], stereo.rs:450-478 */
                for (int i = sw - 1; i >= 0; i--) {
                    const int a = bands[i], b = bands[i + 1];
                    if (!mp3_zero_band(ch1 + a, b - a))
                        break;
                    mp3_intensity(is_pos[sfi - 1], table, inv_pos, mid_side, ch0 + a, ch1 + a, b - a);
                    sfi--;
                    bound = a;
                }
            }
        } else {
            const int *bands = t->sfb_long[sr];
            int is_pos[22]; /* stereo.rs:226-228 */
            for (int i = 0; i < 22; i++)
                is_pos[i] = d->scalefacs1[i];
            is_pos[21] = is_pos[20];
            for (int i = 21; i >= 0; i--) {
                const int a = bands[i], b = bands[i + 1];
                if (!(a >= (int)d->rzero1 || mp3_zero_band(ch1 + a, b - a)))
                    break;
                mp3_intensity(is_pos[i], table, inv_pos, mid_side, ch0 + a, ch1 + a, b - a);
                bound = a;
            }
        }
    }
    if (mid_side && bound > 0)
        mp3_mid_side(ch0, ch1, bound);
}

void so_mp3_imdct_windows(float *dst144) { memcpy(dst144, mp3_get()->imdct_windows, 144 * 4); }
void so_mp3_synthesis_window(float *dst512) { memcpy(dst512, mp3_get()->synth_d, 512 * 4); }

/* reorder (hybrid_synthesis.rs:153-215) */
int so_mp3_reorder(float *buf, int block_type, int is_mixed, int sr, int rzero)
{
    if (block_type != SO_MP3_SHORT)
        return rzero;
    mp3_tables *t = mp3_get();
    const int *bands;
    int n_bands;
    if (is_mixed) {
        int sw = t->sfb_mixed_switch[sr];
        bands = t->sfb_mixed[sr] + sw;
        n_bands = t->sfb_mixed_len[sr] - sw;
    } else {
        bands = t->sfb_short[sr];
        n_bands = 40;
    }
    float reorder_buf[576];
    memset(reorder_buf, 0, sizeof reorder_buf);
    int start = bands[0];
    int i = start;
    /* zip(bands, bands[1..], bands[2..], bands[3..]).step_by(3) */
    for (int b = 0; b + 3 < n_bands; b += 3) {
        int s0 = bands[b], s1 = bands[b + 1], s2 = bands[b + 2], s3 = bands[b + 3];
        if (s0 >= rzero)
            break;
        /* zip stops at the shortest of the three windows */
        int len = s1 - s0;
        if (s2 - s1 < len)
            len = s2 - s1;
        if (s3 - s2 < len)
            len = s3 - s2;
        for (int k = 0; k < len; k++) {
            reorder_buf[i + 0] = buf[s0 + k];
            reorder_buf[i + 1] = buf[s1 + k];
            reorder_buf[i + 2] = buf[s2 + k];
            i += 3;
        }
    }
    for (int k = start; k < i; k++)
        buf[k] = reorder_buf[k];
    return rzero > i ? rzero : i;
}

/* antialias (hybrid_synthesis.rs:218-277) */
int so_mp3_antialias(float *samples, int block_type, int is_mixed, int rzero)
{
    int sb_limit;
    if (block_type == SO_MP3_SHORT) {
        if (!is_mixed)
            return rzero;
        sb_limit = 2;
    } else {
        sb_limit = 32;
    }
    mp3_tables *t = mp3_get();
    int sb_rzero = rzero / 18;
    int lim = sb_limit < sb_rzero + 2 ? sb_limit : sb_rzero + 2;
    if (lim > 32)
        lim = 32;
    rzero = 18 * lim;
    for (int sb = 18; sb < rzero; sb += 18) {
        for (int i = 0; i < 8; i++) {
            int li = sb - 1 - i, ui = sb + i;
            float lower = samples[li], upper = samples[ui];
            samples[li] = lower * t->cs[i] - upper * t->ca[i];
            samples[ui] = upper * t->cs[i] + lower * t->ca[i];
        }
    }
    return rzero;
}

/* sdct_ii_9 (hybrid_synthesis.rs:720-779); y has stride 2 */
static void mp3_sdct_ii_9(const mp3_tables *t, const float *x, float *y)
{
    const float *D = t->sdct9_d;
    float a01 = x[3] + x[5], a02 = x[3] - x[5], a03 = x[6] + x[2], a04 = x[6] - x[2];
    float a05 = x[1] + x[7], a06 = x[1] - x[7], a07 = x[8] + x[0], a08 = x[8] - x[0];
    float a09 = x[4] + a05, a10 = a01 + a03, a11 = a10 + a07, a12 = a03 - a07;
    float a13 = a01 - a07, a14 = a01 - a03, a15 = a02 - a04, a16 = a15 + a08;
    float a17 = a04 + a08, a18 = a02 - a08, a19 = a02 + a04, a20 = 2.0f * x[4] - a05;
    float m1 = D[0] * a06, m2 = D[1] * a12, m3 = D[2] * a13, m4 = D[3] * a14;
    float m5 = D[0] * a16, m6 = D[4] * a17, m7 = D[5] * a18, m8 = D[6] * a19;
    float a21 = a20 + m2, a22 = a20 - m2, a23 = a20 + m3;
    float a24 = m1 + m6, a25 = m1 - m6, a26 = m1 + m7;
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

/* sdct_ii_18 (hybrid_synthesis.rs:665-716) */
static void mp3_sdct_ii_18(const mp3_tables *t, const float *x, float *y)
{
    float even[9], odd[9];
    for (int i = 0; i < 9; i++)
        even[i] = x[i] + x[17 - i];
    mp3_sdct_ii_9(t, even, y);
    for (int i = 0; i < 9; i++)
        odd[i] = t->sdct18_scale[i] * (x[i] - x[17 - i]);
    mp3_sdct_ii_9(t, odd, y + 1);
    for (int i = 3; i <= 17; i += 2)
        y[i] -= y[i - 2];
}

/* dct_iv (hybrid_synthesis.rs:608-660) */
static void mp3_dct_iv(const mp3_tables *t, const float *x, float *y)
{
    float samples[18];
    for (int i = 0; i < 18; i++)
        samples[i] = t->dct_iv_scale[i] * x[i];
    mp3_sdct_ii_18(t, samples, y);
    y[0] /= 2.0f;
    for (int i = 1; i < 17; i++)
        y[i] = (y[i] / 2.0f) - y[i - 1];
    y[17] = (y[17] / 2.0f) - y[16];
}

/* imdct36 (hybrid_synthesis.rs:571-603) */
void so_mp3_imdct36(float *x, const float *window, float *overlap)
{
    mp3_tables *t = mp3_get();
    float dct[18];
    mp3_dct_iv(t, x, dct);
    for (int i = 0; i < 9; i++)
        x[i] = overlap[i] + dct[9 + i] * window[i];
    for (int i = 9; i < 18; i++)
        x[i] = overlap[i] - dct[27 - i - 1] * window[i];
    for (int i = 18; i < 27; i++)
        overlap[i - 18] = -dct[27 - i - 1] * window[i];
    for (int i = 27; i < 36; i++)
        overlap[i - 18] = -dct[i - 27] * window[i];
}

/* imdct12_win (hybrid_synthesis.rs:363-455) */
void so_mp3_imdct12_win(float *x, const float *window, float *overlap)
{
    mp3_tables *t = mp3_get();
    float tmp[36];
    for (int i = 0; i < 36; i++)
        tmp[i] = 0.0f;
    for (int w = 0; w < 3; w++) {
        for (int i = 0; i < 3; i++) {
            const float *cl = t->half_cos_12[i], *cr = t->half_cos_12[i + 3];
            float yl = (x[w] * cl[0]) + (x[3 * 1 + w] * cl[1]) + (x[3 * 2 + w] * cl[2]) +
                       (x[3 * 3 + w] * cl[3]) + (x[3 * 4 + w] * cl[4]) + (x[3 * 5 + w] * cl[5]);
            float yr = (x[w] * cr[0]) + (x[3 * 1 + w] * cr[1]) + (x[3 * 2 + w] * cr[2]) +
                       (x[3 * 3 + w] * cr[3]) + (x[3 * 4 + w] * cr[4]) + (x[3 * 5 + w] * cr[5]);
            tmp[6 + 6 * w + 3 - i - 1] += -yl * window[3 - i - 1];
            tmp[6 + 6 * w + i + 3] += yl * window[i + 3];
            tmp[6 + 6 * w + i + 6] += yr * window[i + 6];
            tmp[6 + 6 * w + 12 - i - 1] += yr * window[12 - i - 1];
        }
    }
    for (int i = 0; i < 18; i++) {
        x[i] = tmp[i] + overlap[i];
        overlap[i] = tmp[i + 18];
    }
}

/* hybrid_synthesis (hybrid_synthesis.rs:280-359) */
void so_mp3_hybrid(float *samples, float *overlap, int block_type, int is_mixed, int rzero)
{
    mp3_tables *t = mp3_get();
    int sb_limit = (rzero + 17) / 18;
    int sb_split = (block_type == SO_MP3_SHORT) ? (is_mixed ? 2 : 0) : 32;
    if (sb_split > 0) {
        const float *window = block_type == SO_MP3_START ? t->imdct_windows[1]
                              : block_type == SO_MP3_END ? t->imdct_windows[3]
                                                         : t->imdct_windows[0];
        int sb_long_end = sb_split < sb_limit ? sb_split : sb_limit;
        for (int sb = 0; sb < sb_long_end; sb++)
            so_mp3_imdct36(samples + 18 * sb, window, overlap + 18 * sb);
    }
    if (sb_split < 32) {
        const float *window = t->imdct_windows[2];
        int sb_short_begin = sb_split < sb_limit ? sb_split : sb_limit;
        for (int sb = sb_short_begin; sb < sb_limit; sb++)
            so_mp3_imdct12_win(samples + 18 * sb, window, overlap + 18 * sb);
    }
    for (int sb = sb_limit; sb < 32; sb++) {
        for (int i = 0; i < 18; i++) {
            samples[18 * sb + i] = overlap[18 * sb + i];
            overlap[18 * sb + i] = 0.0f;
        }
    }
}

/* frequency_inversion (hybrid_synthesis.rs:458-485) */
void so_mp3_frequency_inversion(float *samples)
{
    for (int i = 18; i < 576; i += 36)
        for (int j = 1; j < 18; j += 2)
            samples[i + j] = -samples[i + j];
}

/*
 * dct32 (synthesis.rs:348-844).  The reference flattens B.G. Lee's recursion by
 * hand; the recursion it flattens is
 *   t[i]     = x[i] + x[n-1-i]
 *   t[n/2+i] = (x[i] - x[n-1-i]) * COS_{n/2}[i]
 *   a = dct(t[0..n/2]), b = dct(t[n/2..n])
 *   y[2i] = a[i]; y[2i+1] = b[i] + b[i+1] (i < n/2-1); y[n-2] = a[n/2-1]; y[n-1] = b[n/2-1]
 * with the 2-point base [x0 + x1, (x0 - x1) * COS_1]; restated as that recursion.
 */
static void mp3_dct_lee(const mp3_tables *t, float *x, int n)
{
    if (n == 2) {
        float a = x[0] + x[1], b = (x[0] - x[1]) * t->cos1;
        x[0] = a;
        x[1] = b;
        return;
    }
    int h = n / 2;
    const float *c = n == 32 ? t->cos16 : n == 16 ? t->cos8 : n == 8 ? t->cos4 : t->cos2;
    float tmp[32];
    for (int i = 0; i < h; i++) {
        tmp[i] = x[i] + x[n - 1 - i];
        tmp[h + i] = (x[i] - x[n - 1 - i]) * c[i];
    }
    mp3_dct_lee(t, tmp, h);
    mp3_dct_lee(t, tmp + h, h);
    for (int i = 0; i < h - 1; i++) {
        x[2 * i] = tmp[i];
        x[2 * i + 1] = tmp[h + i] + tmp[h + i + 1];
    }
    x[n - 2] = tmp[h - 1];
    x[n - 1] = tmp[n - 1];
}

void so_mp3_dct32(const float *x, float *y)
{
    float tmp[32];
    memcpy(tmp, x, sizeof tmp);
    mp3_dct_lee(mp3_get(), tmp, 32);
    memcpy(y, tmp, sizeof tmp);
}

/* synthesis (synthesis.rs:158-336) */
void so_mp3_polyphase(float *v_all, int *v_front_io, int n_frames, const float *in, float *out)
{
    mp3_tables *t = mp3_get();
    int v_front = *v_front_io;
    float s_vec[32], d_vec[32];
    for (int b = 0; b < n_frames; b++) {
        for (int i = 0; i < 32; i++)
            s_vec[i] = in[n_frames * i + b];
        float *v = v_all + 64 * v_front;
        so_mp3_dct32(s_vec, d_vec);
        /* synthesis.rs:247-263 */
        for (int k = 0; k < 15; k++) {
            v[47 - k] = -d_vec[1 + k];
            v[49 + k] = -d_vec[1 + k];
            v[31 - k] = -d_vec[17 + k];
            v[1 + k] = d_vec[17 + k];
        }
        v[0] = d_vec[16];
        v[32] = -d_vec[16];
        v[48] = -d_vec[0];
        v[16] = 0.0f;
        /* synthesis.rs:309-324 */
        float o_vec[32];
        for (int i = 0; i < 32; i++)
            o_vec[i] = 0.0f;
        for (int j = 0; j < 8; j++) {
            int v_start = v_front + (j << 1);
            const float *v0 = v_all + 64 * ((v_start + 0) & 0xf);
            const float *v1 = v_all + 64 * ((v_start + 1) & 0xf) + 32;
            int k = j << 6;
            for (int i = 0; i < 32; i++) {
                o_vec[i] += v0[i] * t->synth_d[k + i + 0];
                o_vec[i] += v1[i] * t->synth_d[k + i + 32];
            }
        }
        memcpy(out + (b << 5), o_vec, sizeof o_vec);
        v_front = (v_front + 15) & 0xf;
    }
    *v_front_io = v_front;
}

/* layer3/mod.rs:421-477, per-channel tail of the granule loop */
void so_mp3_synth_batch(const float *xr, const uint8_t *side, int sr, float *overlap, float *v_vec,
                        int32_t *v_front, float *pcm, size_t n_chains, size_t granules_per_chain)
{
    for (size_t c = 0; c < n_chains; c++) {
        int vf = v_front[c];
        for (size_t g = 0; g < granules_per_chain; g++) {
            size_t idx = c * granules_per_chain + g;
            const uint8_t *s = side + 4 * idx;
            int block_type = s[0], is_mixed = s[1];
            int rzero = (int)s[2] | ((int)s[3] << 8);
            float buf[576];
            memcpy(buf, xr + idx * 576, sizeof buf);
            rzero = so_mp3_reorder(buf, block_type, is_mixed, sr, rzero);
            rzero = so_mp3_antialias(buf, block_type, is_mixed, rzero);
            so_mp3_hybrid(buf, overlap + c * 576, block_type, is_mixed, rzero);
            so_mp3_frequency_inversion(buf);
            so_mp3_polyphase(v_vec + c * 1024, &vf, 18, buf, pcm + idx * 576);
        }
        v_front[c] = vf;
    }
}

/* ======================================================================== */
/* Vorbis: symphonia-codec-vorbis/src/{window,dsp,lib,floor,residue}.rs      */
/* ======================================================================== */

/* generate_win_curve (window.rs:11-24) */
void so_vorbis_window(int bs, float *dst)
{
    int len = bs / 2;
    double denom = (double)len;
    const double FRAC_PI_2 = 1.57079632679489661923132169163975144;
    for (int i = 0; i < len; i++) {
        double num = (double)i + 0.5;
        double frac = FRAC_PI_2 * (num / denom);
        double s = sin(frac);
        dst[i] = (float)sin(FRAC_PI_2 * (s * s)); /* powi(2) == s*s */
    }
}

/* lib.rs:265-277 */
void so_vorbis_inverse_coupling(float *mag, float *ang, size_t n)
{
    for (size_t i = 0; i < n; i++) {
        float m = mag[i], a = ang[i], nm, na;
        if (m > 0.0f) {
            if (a > 0.0f) {
                nm = m;
                na = m - a;
            } else {
                nm = m + a;
                na = m;
            }
        } else {
            if (a > 0.0f) {
                nm = m;
                na = m + a;
            } else {
                nm = m - a;
                na = m;
            }
        }
        mag[i] = nm;
        ang[i] = na;
    }
}

/* lib.rs:289-291 */
void so_vorbis_dot_product(float *floor, const float *residue, size_t n)
{
    for (size_t i = 0; i < n; i++)
        floor[i] *= residue[i];
}

/* residue.rs:177-218 (both the 2-channel and the generic arm give
 * planar[c][i] = type2[i * n_ch + c]) */
void so_vorbis_deinterleave2(const float *type2, float *planar, int n_ch, size_t n2)
{
    for (int c = 0; c < n_ch; c++)
        for (size_t i = 0; i < n2; i++)
            planar[(size_t)c * n2 + i] = type2[i * (size_t)n_ch + (size_t)c];
}

static float floor1_db(int y)
{
    union {
        uint32_t u;
        float f;
    } v;
    v.u = SYM_VORBIS_FLOOR1_DB_BITS[y];
    return v.f;
}

void so_vorbis_floor1_table(float *dst256)
{
    for (int i = 0; i < 256; i++)
        dst256[i] = floor1_db(i);
}

/* render_point (floor.rs:776-782) */
static int32_t floor1_render_point(uint32_t x0, int32_t y0, uint32_t x1, int32_t y1, uint32_t x)
{
    int32_t dy = y1 - y0;
    uint32_t adx = x1 - x0;
    uint32_t ady = (uint32_t)(dy < 0 ? -dy : dy);
    uint32_t err = ady * (x - x0);
    uint32_t off = err / adx;
    return dy < 0 ? y0 - (int32_t)off : y0 + (int32_t)off;
}

/* render_line (floor.rs:785-825) */
static void floor1_render_line(uint32_t x0, int32_t y0, uint32_t x1, int32_t y1, size_t n, float *v)
{
    if ((size_t)x0 >= n)
        return;
    int32_t dy = y1 - y0;
    int32_t adx = (int32_t)(x1 - x0);
    int32_t base = dy / adx;
    int32_t y = y0;
    int32_t sy = dy < 0 ? base - 1 : base + 1;
    int32_t ady = (dy < 0 ? -dy : dy) - (base < 0 ? -base : base) * adx;
    v[x0] = floor1_db(y);
    int32_t err = 0;
    size_t x_begin = (size_t)x0 + 1;
    size_t x_end = n < (size_t)x1 ? n : (size_t)x1;
    if (x_begin > x_end)
        return;
    for (size_t x = x_begin; x < x_end; x++) {
        err += ady;
        if (err >= adx) {
            err -= adx;
            y += sy;
        } else {
            y += base;
        }
        v[x] = floor1_db(y);
    }
}

/* Floor 0 (symphonia-codec-vorbis/src/floor.rs).  bark / bark_map: :353-376 (f64 atan / floor from libm); the
 * `2 cos(coeff)` step that ends read_channel: :246-248 (f32 cos = cosf); synthesis: :262-340; linear_floor0_value: :379-390
 * (f32 sqrt, expf; the literal 0.11512925 is an f32). */
static double vorbis_bark(double x)
{
    return (13.1 * atan(0.00074 * x)) + (2.24 * atan(0.0000000185 * x * x)) + (0.0001 * x);
}
void so_vorbis_bark_map(uint32_t n, uint32_t rate16, uint32_t map_size16, int32_t *map)
{
    const int32_t foobar_min = (int32_t)map_size16 - 1;
    const double rate = (double)rate16;
    const double rate_by_2n = rate / (2.0 * (double)n);
    const double c = (double)map_size16 / vorbis_bark(0.5 * rate);
    for (uint32_t i = 0; i < n; i++) {
        const double v = floor(vorbis_bark(rate_by_2n * (double)i) * c);
        int32_t foobar = v >= 2147483647.0 ? INT32_MAX : v <= -2147483648.0 ? INT32_MIN : (int32_t)v; /* `as i32` saturates */
        map[i] = foobar < foobar_min ? foobar : foobar_min;
    }
}
void so_vorbis_floor0_coeffs(float *coeffs, int order)
{
    for (int i = 0; i < order; i++)
        coeffs[i] = 2.0f * cosf(coeffs[i]);
}
/* coeffs = the 2 cos(..) values; map = the bark map of this block size (n entries); returns 0, or -1 for the
 * reference's decode_error("vorbis: invalid floor0 coefficients") (floor.rs:315-317). */
int so_vorbis_floor0(const float *coeffs, int order, const int32_t *map, uint32_t n, uint32_t bark_map_size,
                     uint32_t amplitude_bits, uint32_t amplitude_offset, uint64_t amplitude, float *floor_out)
{
    const float pi_f = 3.14159265358979323846264338327950288f;
    const float omega_step = pi_f / (float)bark_map_size;
    uint32_t i = 0;
    for (;;) {
        const int32_t iter_cond = map[i];
        const float omega = omega_step * (float)iter_cond;
        const float cos_omega = cosf(omega);
        const float two_cos_omega = 2.0f * cos_omega;
        float p = 1.0f, q = 1.0f;
        int j = 0;
        for (; j + 1 < order; j += 2) {
            p *= coeffs[j + 1] - two_cos_omega;
            q *= coeffs[j] - two_cos_omega;
        }
        if (j < order) { /* odd order: the chunks_exact(2) remainder */
            q *= coeffs[j] - two_cos_omega;
            p = p * p * (1.0f - (cos_omega * cos_omega));
            q = q * q * 0.25f;
        } else {
            p = p * p * ((1.0f - cos_omega) / 2.0f);
            q = q * q * ((1.0f + cos_omega) / 2.0f);
        }
        if (p + q == 0.0f)
            return -1;
        /* linear_floor0_value */
        const float a = (float)(amplitude * (uint64_t)amplitude_offset); /* wrapping_mul, then `as f32` */
        const float b = sqrtf(p + q) * (float)((UINT64_C(1) << amplitude_bits) - 1);
        const float value = expf(0.11512925f * ((a / b) - (float)amplitude_offset));
        while (i < n && map[i] == iter_cond) {
            floor_out[i] = value;
            i++;
        }
        if (i >= n)
            break;
    }
    return 0;
}

void so_vorbis_floor1(const uint32_t *x_list, const uint32_t *yv, int n_posts, int multiplier,
                      uint32_t n, float *floor_out)
{
    int lo_n[65], hi_n[65], order[65] = {0}, flag[65];  /* (order[0] is read below even for an empty post list) */
    int32_t final_y[65] = {0};
    /* find_neighbors (floor.rs:748-773) for every post; sort order (floor.rs:547-554) */
    for (int x = 0; x < n_posts; x++) {
        uint32_t bound = x_list[x], low = 0, high = 0xffffffffu;
        int rl = 0, rh = 0;
        for (int i = 0; i < x; i++) {
            uint32_t xv = x_list[i];
            if (xv > low && xv < bound) {
                low = xv;
                rl = i;
            }
            if (xv < high && xv > bound) {
                high = xv;
                rh = i;
            }
        }
        lo_n[x] = rl;
        hi_n[x] = rh;
        order[x] = x;
    }
    for (int i = 1; i < n_posts; i++) { /* stable insertion sort == sort_by_key */
        int k = order[i], j = i - 1;
        while (j >= 0 && x_list[order[j]] > x_list[k]) {
            order[j + 1] = order[j];
            j--;
        }
        order[j + 1] = k;
    }
    /* synthesis_step1 (floor.rs:568-625) */
    static const int32_t RANGE[4] = {256, 128, 86, 64};
    int32_t range = RANGE[multiplier - 1];
    flag[0] = flag[1] = 1;
    final_y[0] = (int32_t)yv[0];
    final_y[1] = (int32_t)yv[1];
    for (int i = 2; i < n_posts; i++) {
        int lo = lo_n[i], hi = hi_n[i];
        int32_t predicted =
            floor1_render_point(x_list[lo], final_y[lo], x_list[hi], final_y[hi], x_list[i]);
        int32_t val = (int32_t)yv[i];
        int32_t highroom = range - predicted, lowroom = predicted;
        if (val != 0) {
            int32_t room = 2 * (highroom < lowroom ? highroom : lowroom);
            flag[lo] = flag[hi] = flag[i] = 1;
            if (val >= room)
                final_y[i] = highroom > lowroom ? val - lowroom + predicted
                                                : predicted - val + highroom - 1;
            else
                final_y[i] = (val & 1) ? predicted - ((val + 1) / 2) : predicted + (val / 2);
        } else {
            flag[i] = 0;
            final_y[i] = predicted;
        }
    }
    /* synthesis_step2 (floor.rs:627-653) */
    int32_t mult = multiplier;
    uint32_t hx = 0, lx = 0;
    int32_t hy = 0;
    int32_t ly = final_y[order[0]] * mult;
    ly = ly < 0 ? 0 : ly > 255 ? 255 : ly;
    for (int k = 1; k < n_posts; k++) {
        int i = order[k];
        if (flag[i]) {
            hy = final_y[i] * mult;
            hy = hy < 0 ? 0 : hy > 255 ? 255 : hy;
            hx = x_list[i];
            floor1_render_line(lx, ly, hx, hy, n, floor_out);
            lx = hx;
            ly = hy;
        }
    }
    if (hx < n)
        floor1_render_line(hx, hy, n, hy, n, floor_out);
}

/* overlap_add (dsp.rs:135-145): out = s0 * win[len-1-k] + s1 * win[k] */
static void vorbis_overlap_add(float *out, const float *left, const float *right, const float *win,
                               int len)
{
    for (int k = 0; k < len; k++)
        out[k] = left[k] * win[len - 1 - k] + right[k] * win[k];
}

int so_vorbis_synth_batch(int bs0_exp, int bs1_exp, const float *spectra, size_t spec_stride,
                          const uint8_t *block_flag, int32_t *prev_flag, float *overlap_all,
                          float *pcm, size_t pcm_stride, size_t n_chains, size_t blocks_per_chain)
{
    int bs0 = 1 << bs0_exp, bs1 = 1 << bs1_exp;
    float *win_short = (float *)malloc(sizeof(float) * (size_t)(bs0 / 2));
    float *win_long = (float *)malloc(sizeof(float) * (size_t)(bs1 / 2));
    float *imdct = (float *)malloc(sizeof(float) * (size_t)bs1);
    so_vorbis_window(bs0, win_short);
    so_vorbis_window(bs1, win_long);
    so_imdct *im_short = so_imdct_new(bs0 >> 1, 1.0), *im_long = so_imdct_new(bs1 >> 1, 1.0);
    for (size_t c = 0; c < n_chains; c++) {
        const float *sp = spectra + c * spec_stride;
        float *out = pcm + c * pcm_stride;
        float *overlap = overlap_all + c * (size_t)(bs1 / 2);
        int32_t pf = prev_flag[c];
        for (size_t b = 0; b < blocks_per_chain; b++) {
            int flag = block_flag[c * blocks_per_chain + b] ? 1 : 0;
            int pflag = pf < 0 ? flag : pf;        /* lib.rs:298 */
            int bs = flag ? bs1 : bs0;             /* dsp.rs:77 */
            int prev_n = pflag ? bs1 : bs0;        /* lib.rs:300-301 */
            int out_len = (prev_n + bs) / 4;       /* lib.rs:303 */
            so_imdct_run(flag ? im_long : im_short, sp, imdct); /* dsp.rs:80 */
            const float *win = (flag && pflag) ? win_long : win_short; /* dsp.rs:83 */
            if (pflag == flag) {
                vorbis_overlap_add(out, overlap, imdct, win, bs / 2);
            } else if (pflag && !flag) {
                int start = (bs1 - bs0) / 4;
                memcpy(out, overlap, sizeof(float) * (size_t)start);
                vorbis_overlap_add(out + start, overlap + start, imdct, win, bs0 / 2);
            } else {
                int start = (bs1 - bs0) / 4, end = start + bs0 / 2;
                vorbis_overlap_add(out, overlap, imdct + start, win, bs0 / 2);
                memcpy(out + bs0 / 2, imdct + end, sizeof(float) * (size_t)(bs1 / 2 - end));
            }
            memcpy(overlap, imdct + bs / 2, sizeof(float) * (size_t)(bs / 2)); /* dsp.rs:125 */
            sp += bs / 2;
            out += out_len;
            pf = flag; /* lib.rs:328 */
        }
        prev_flag[c] = pf;
    }
    so_imdct_free(im_short);
    so_imdct_free(im_long);
    free(win_short);
    free(win_long);
    free(imdct);
    return 0;
}

/* ======================================================================== */
/* FLAC: symphonia-bundle-flac/src/decoder.rs                                */
/* ======================================================================== */

static inline int32_t wrap_add32(int32_t a, int32_t b)
{
    return (int32_t)((uint32_t)a + (uint32_t)b);
}

/* fixed_predict (decoder.rs:663-710); i64 Wrapping arithmetic, truncating cast */
void so_flac_fixed_predict(int order, int32_t *buf, size_t len)
{
    switch (order) {
    case 0:
        break;
    case 1:
        for (size_t i = 1; i < len; i++)
            buf[i] = wrap_add32(buf[i], buf[i - 1]);
        break;
    case 2:
        for (size_t i = 2; i < len; i++) {
            uint64_t a = (uint64_t)(-1 * (int64_t)buf[i - 2]);
            uint64_t b = (uint64_t)(2 * (int64_t)buf[i - 1]);
            buf[i] = wrap_add32(buf[i], (int32_t)(uint32_t)(a + b));
        }
        break;
    case 3:
        for (size_t i = 3; i < len; i++) {
            uint64_t a = (uint64_t)(1 * (int64_t)buf[i - 3]);
            uint64_t b = (uint64_t)(-3 * (int64_t)buf[i - 2]);
            uint64_t c = (uint64_t)(3 * (int64_t)buf[i - 1]);
            buf[i] = wrap_add32(buf[i], (int32_t)(uint32_t)(a + b + c));
        }
        break;
    case 4:
        for (size_t i = 4; i < len; i++) {
            uint64_t a = (uint64_t)(-1 * (int64_t)buf[i - 4]);
            uint64_t b = (uint64_t)(4 * (int64_t)buf[i - 3]);
            uint64_t c = (uint64_t)(-6 * (int64_t)buf[i - 2]);
            uint64_t d = (uint64_t)(4 * (int64_t)buf[i - 1]);
            buf[i] = wrap_add32(buf[i], (int32_t)(uint32_t)(a + b + c + d));
        }
        break;
    default:
        break;
    }
}

/* lpc_predict::<N> (decoder.rs:716-752): coeffs[N] right-aligned, oldest first */
static void flac_lpc_n(int n_max, size_t order, const int32_t *coeffs_n, uint32_t shift,
                       int32_t *buf, size_t len)
{
    size_t N = (size_t)n_max;
    size_t n_prefill = (N < len ? N : len) - order;
    /* The i64 sum of the reference (`.sum::<i64>()`, release build) wraps; signed overflow is undefined in C, so the
     * accumulation is done in uint64_t (same bits) and reinterpreted for the arithmetic shift. */
    for (size_t i = order; i < order + n_prefill; i++) {
        uint64_t predicted = 0;
        for (size_t j = 0; j < order; j++)
            predicted += (uint64_t)((int64_t)coeffs_n[N - order + j] * (int64_t)buf[i - order + j]);
        buf[i] = wrap_add32(buf[i], (int32_t)((int64_t)predicted >> shift));
    }
    if (len <= N)
        return;
    for (size_t i = N; i < len; i++) {
        uint64_t predicted = 0;
        for (size_t j = 0; j < N; j++)
            predicted += (uint64_t)((int64_t)coeffs_n[j] * (int64_t)buf[i - N + j]);
        buf[i] = wrap_add32(buf[i], (int32_t)((int64_t)predicted >> shift));
    }
}

void so_flac_lpc_predict(int order, const int32_t *coeffs, uint32_t shift, int32_t *buf,
                         size_t len)
{
    /* decoder.rs:477-481: first coefficient read lands at index 31, and so on down */
    int32_t qlp[32];
    memset(qlp, 0, sizeof qlp);
    for (int j = 0; j < order; j++)
        qlp[31 - j] = coeffs[j];
    /* decoder.rs:497-504 */
    int n = order <= 4 ? 4 : order <= 6 ? 6 : order <= 8 ? 8 : order <= 10 ? 10 : order <= 12 ? 12 : 32;
    flac_lpc_n(n, (size_t)order, qlp + 32 - n, shift, buf, len);
}

void so_flac_decorrelate(int mode, int32_t *ch0, int32_t *ch1, size_t len)
{
    switch (mode) {
    case 1: /* left/side: decoder.rs:32-36; ch0 = left, ch1 = side */
        for (size_t i = 0; i < len; i++)
            ch1[i] = (int32_t)((uint32_t)ch0[i] - (uint32_t)ch1[i]);
        break;
    case 2: /* mid/side: decoder.rs:38-73 */
        for (size_t i = 0; i < len; i++) {
            int32_t mid = (int32_t)(((uint32_t)ch0[i] << 1) | ((uint32_t)ch1[i] & 1u));
            int32_t side = ch1[i];
            ch0[i] = (int32_t)((uint32_t)mid + (uint32_t)side) >> 1;
            ch1[i] = (int32_t)((uint32_t)mid - (uint32_t)side) >> 1;
        }
        break;
    case 3: /* right/side: decoder.rs:75-79; ch0 = side, ch1 = right */
        for (size_t i = 0; i < len; i++)
            ch0[i] = wrap_add32(ch0[i], ch1[i]);
        break;
    default:
        break;
    }
}

void so_flac_shl(int32_t *buf, size_t len, uint32_t shift)
{
    if (shift == 0)
        return;
    for (size_t i = 0; i < len; i++)
        buf[i] = (int32_t)((uint32_t)buf[i] << (shift & 31));
}

int32_t so_flac_rice_signed_to_i32(uint32_t word)
{
    int32_t div2 = (int32_t)(word >> 1);
    int32_t sign = -(int32_t)(word & 1u);
    return div2 ^ sign;
}

void so_flac_restore_batch(int32_t *buf, const uint8_t *desc, const int32_t *coeffs,
                           size_t n_blocks, size_t blocksize)
{
    for (size_t b = 0; b < n_blocks; b++) {
        const uint8_t *d = desc + 4 * b;
        int32_t *x = buf + b * blocksize;
        if (d[0] == 1)
            so_flac_fixed_predict(d[1], x, blocksize);
        else if (d[0] == 2)
            so_flac_lpc_predict(d[1], coeffs + 32 * b, d[2], x, blocksize);
        so_flac_shl(x, blocksize, d[3]); /* samples_shl(dropped_bps) decoder.rs:396-409 */
    }
}

/* ========================================================================== */
/* ALAC (symphonia-codec-alac/src/lib.rs)                                      */
/* ========================================================================== */

/* clip_msbs (lib.rs:659-661): sign-extend from (32 - num) bits. */
static inline int32_t alac_clip_msbs(int32_t val, uint32_t num) {
    return (int32_t)((uint32_t)val << num) >> num;
}
static inline int32_t wadd(int32_t a, int32_t b) { return (int32_t)((uint32_t)a + (uint32_t)b); }
static inline int32_t wsub(int32_t a, int32_t b) { return (int32_t)((uint32_t)a - (uint32_t)b); }
static inline int32_t wmul(int32_t a, int32_t b) { return (int32_t)((uint32_t)a * (uint32_t)b); }

/* ElementChannel::predict (lib.rs:165-264).  coeffs_in[32] as read by try_read (lib.rs:94-98); the sign-LMS
 * adaptation works on a private copy, as the reference's per-packet ElementChannel does.  Returns 0, or -1 for the
 * reference's decode_error("alac: invalid mode") (lib.rs:167-169), in which case `out` is left untouched.
 * Release-mode Rust arithmetic: the plain `+ - *` of the reference wrap. */
int so_alac_predict(int32_t *out, size_t len, uint32_t mode, uint32_t lpc_order, uint32_t shift, uint32_t bps,
                    const int32_t *coeffs_in) {
    if (mode > 0 && mode < 15) return -1;
    if (lpc_order == 0 || len == 0) return 0;
    int32_t coeffs[32];
    memcpy(coeffs, coeffs_in, sizeof coeffs);
    const uint32_t num_clip_bits = 32 - bps;
    if (lpc_order == 31 || mode == 15) {
        for (size_t i = 1; i < len; i++) out[i] = alac_clip_msbs(wadd(out[i], out[i - 1]), num_clip_bits);
    }
    const size_t order = lpc_order;
    const size_t warm_end = 1 + order < len ? 1 + order : len;
    for (size_t i = 1; i < warm_end; i++) out[i] = alac_clip_msbs(wadd(out[i], out[i - 1]), num_clip_bits);
    for (size_t i = 1 + order; i < len; i++) {
        int32_t res = out[i];
        const int32_t past0 = out[i - order - 1];
        int32_t sum = 0;
        /* coeffs[..order].iter().rev().zip(&out[i-order..i]): coeff[order-1-j] with out[i-order+j] */
        for (size_t j = 0; j < order; j++) sum = wadd(sum, wmul(coeffs[order - 1 - j], wsub(out[i - order + j], past0)));
        const int32_t val = wadd(sum, (int32_t)((1u << shift) >> 1)) >> shift;
        out[i] = alac_clip_msbs(wadd(wadd(out[i], past0), val), num_clip_bits);
        if (res > 0) {
            for (size_t j = 0; j < order; j++) {
                const int32_t v = wsub(past0, out[i - order + j]);
                const int32_t sign = (v > 0) - (v < 0);
                coeffs[order - 1 - j] = wsub(coeffs[order - 1 - j], sign);
                res = wsub(res, wmul((int32_t)(1 + j), wmul(sign, v) >> shift));
                if (res <= 0) break;
            }
        } else if (res < 0) {
            for (size_t j = 0; j < order; j++) {
                const int32_t v = wsub(past0, out[i - order + j]);
                const int32_t sign = (v > 0) - (v < 0);
                coeffs[order - 1 - j] = wadd(coeffs[order - 1 - j], sign);
                res = wsub(res, wmul((int32_t)(1 + j), wmul(-sign, v) >> shift));
                if (res >= 0) break;
            }
        }
    }
    return 0;
}

/* decorrelate_mid_side (lib.rs:664-671) */
void so_alac_decorrelate_mid_side(int32_t *out0, int32_t *out1, size_t len, int32_t weight, uint32_t shift) {
    for (size_t i = 0; i < len; i++) {
        const int32_t s0 = wsub(wadd(out0[i], out1[i]), wmul(out1[i], weight) >> shift);
        out0[i] = s0;
        out1[i] = wsub(s0, out1[i]);
    }
}

/* n_blocks element-channels of `blocksize` samples, in place.  desc[block] = {mode u8, lpc_order u8, shift u8,
 * bps u8}; coeffs[block][32].  Blocks with an invalid mode are left untouched. */
void so_alac_predict_batch(int32_t *buf, const uint8_t *desc, const int32_t *coeffs, size_t n_blocks, size_t blocksize) {
    for (size_t b = 0; b < n_blocks; b++)
        so_alac_predict(buf + b * blocksize, blocksize, desc[4 * b], desc[4 * b + 1], desc[4 * b + 2], desc[4 * b + 3],
                        coeffs + 32 * b);
}
