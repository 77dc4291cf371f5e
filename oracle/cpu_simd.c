/*
 * cpu_simd.c -- the headline workload (AAC-LC, ONLY_LONG frames) vectorised ACROSS chains for the CPU baseline.
 * TEST / BENCH INFRASTRUCTURE ONLY (see symoracle.h): bench.py's `cpu_baseline` leg times it next to the GPU number.
 *
 * The reference's default build runs its FFT through rustfft's SIMD kernels (BENCHMARKS.md:15); there is no Rust
 * toolchain here, so the strongest honest CPU figure we can produce is this: the same operation DAG as the scalar
 * restatement (symoracle.c: no_simd.rs radix-2 graph, mdct.rs pre/post twiddle, aac/dsp.rs window + overlap-add), with
 * every scalar replaced by a vector of W = 16 lanes, one lane per chain, built with -O3 -march=native
 * -ffp-contract=off.  Lanes never interact, every lane performs exactly the scalar sequence of rounded operations, so
 * the output is bit-identical to the scalar oracle (tests/test_cpu_baseline.py) -- it is a faster schedule, not a
 * different algorithm.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "symoracle.h"

#define W 16
typedef float vf __attribute__((vector_size(4 * W)));
typedef struct {
    vf re, im;
} vcpx;

static inline vcpx v_add(vcpx a, vcpx b) { vcpx r = {a.re + b.re, a.im + b.im}; return r; }
static inline vcpx v_sub(vcpx a, vcpx b) { vcpx r = {a.re - b.re, a.im - b.im}; return r; }
static inline vf splat(float x) { return (vf){x, x, x, x, x, x, x, x, x, x, x, x, x, x, x, x}; }
/* num-complex Mul with a scalar twiddle on the RIGHT (odd * w) */
static inline vcpx v_mul_sw(vcpx a, float wre, float wim)
{
    vcpx r = {a.re * wre - a.im * wim, a.re * wim + a.im * wre};
    return r;
}
/* ... and with the scalar on the LEFT (w * x): operand order as in the scalar code, the products commute bit-exactly */
static inline vcpx s_mul_v(float wre, float wim, vcpx b)
{
    vcpx r = {wre * b.re - wim * b.im, wre * b.im + wim * b.re};
    return r;
}

#define FRAC_1_SQRT_2 0.70710678118654752440f

typedef struct {
    float small_tw[33][16][2];
    float *merge_tw[10]; /* log2 n = 6..9 */
    float tw[512][2];    /* Imdct::new_scaled(1024, 1/2048) twiddles */
    float kbd_long[1024];
    uint16_t perm[512];
} simd_tables;

static simd_tables g_t;
static int g_ready;

static void tables_init(void)
{
    for (int n = 16; n <= 32; n <<= 1) so_fft_small_twiddles(n, &g_t.small_tw[n][0][0]);
    for (int lg = 6; lg <= 9; lg++) {
        g_t.merge_tw[lg] = (float *)malloc(sizeof(float) * (size_t)(1 << lg));
        so_fft_twiddles(1 << lg, g_t.merge_tw[lg]);
    }
    so_imdct_twiddles(1024, 1.0 / 2048.0, &g_t.tw[0][0]);
    so_aac_window(1, 4.0f, 1024, g_t.kbd_long);
    for (unsigned i = 0; i < 512; i++) {
        unsigned r = 0, x = i;
        for (int m = 256; m > 0; m >>= 1) {
            r = (r << 1) | (x & 1u);
            x >>= 1;
        }
        g_t.perm[i] = (uint16_t)r;
    }
    g_ready = 1;
}

/* fft_small of symoracle.c (fft2 .. fft32, no_simd.rs:289-454) on vectors */
static void vfft_small(vcpx *x, int n)
{
    if (n == 1) return;
    if (n == 2) {
        vcpx x0 = x[0];
        x[0] = v_add(x0, x[1]);
        x[1] = v_sub(x0, x[1]);
        return;
    }
    const int h = n / 2;
    vfft_small(x, h);
    vfft_small(x + h, h);
    for (int k = 0; k < h; k++) {
        vcpx v = x[h + k], q;
        if (k == 0) {
            q = v;
        } else if (4 * k == n) {
            q.re = v.im;
            q.im = -v.re;
        } else if (8 * k == n) {
            vf a = FRAC_1_SQRT_2 * v.re, b = FRAC_1_SQRT_2 * v.im;
            q.re = a + b;
            q.im = b - a;
        } else if (8 * k == 3 * n) {
            vf a = -FRAC_1_SQRT_2 * v.re, b = -FRAC_1_SQRT_2 * v.im;
            q.re = a - b;
            q.im = a + b;
        } else {
            q = s_mul_v(g_t.small_tw[n][k][0], g_t.small_tw[n][k][1], v);
        }
        vcpx e = x[k];
        x[k] = v_add(e, q);
        x[h + k] = v_sub(e, q);
    }
}

/* W chains x frames_per_chain ONLY_LONG / KBD frames: coeffs[chain][frame][1024], delay[chain][1024] in/out,
 * pcm[chain][frame][1024].  n_chains must be a multiple of W. */
void so_aac_long_kbd_batch_simd(const float *coeffs, float *delay, float *pcm, size_t n_chains, size_t frames_per_chain)
{
    if (!g_ready) tables_init();
    vcpx *z = (vcpx *)aligned_alloc(64, sizeof(vcpx) * 512);
    vf *spec = (vf *)aligned_alloc(64, sizeof(vf) * 1024);
    vf *out = (vf *)aligned_alloc(64, sizeof(vf) * 2048);
    vf *dl = (vf *)aligned_alloc(64, sizeof(vf) * 1024);
    for (size_t c0 = 0; c0 + W <= n_chains; c0 += W) {
        for (int i = 0; i < 1024; i++)
            for (int l = 0; l < W; l++) dl[i][l] = delay[(c0 + (size_t)l) * 1024 + (size_t)i];
        for (size_t t = 0; t < frames_per_chain; t++) {
            /* lanes <- chains (a transpose of W x 1024 floats) */
            for (int l = 0; l < W; l++) {
                const float *src = coeffs + ((c0 + (size_t)l) * frames_per_chain + t) * 1024;
                for (int i = 0; i < 1024; i++) spec[i][l] = src[i];
            }
            /* mdct.rs:81-88 */
            for (int i = 0; i < 512; i++) {
                const float wre = g_t.tw[i][0], wim = g_t.tw[i][1];
                const vf even = spec[2 * i], odd = -spec[1023 - 2 * i];
                z[i].re = odd * wim - even * wre;
                z[i].im = odd * wre + even * wim;
            }
            /* no_simd.rs:101-107 */
            for (int i = 0; i < 512; i++) {
                const int j = g_t.perm[i];
                if (i < j) {
                    vcpx tmp = z[i];
                    z[i] = z[j];
                    z[j] = tmp;
                }
            }
            /* no_simd.rs:221-281 */
            for (int c = 0; c < 512; c += 32) vfft_small(z + c, 32);
            int lg = 6;
            for (int step = 32; step < 512; step <<= 1, lg++) {
                const float *w = g_t.merge_tw[lg];
                for (int base = 0; base < 512; base += step << 1) {
                    vcpx *even = z + base, *odd = z + base + step;
                    for (int k = 0; k < step; k++) {
                        vcpx p = even[k];
                        vcpx q = v_mul_sw(odd[k], w[2 * k], w[2 * k + 1]);
                        even[k] = v_add(p, q);
                        odd[k] = v_sub(p, q);
                    }
                }
            }
            /* mdct.rs:94-137 */
            vf *vec0 = out, *vec1 = out + 512, *vec2 = out + 1024, *vec3 = out + 1536;
            for (int i = 0; i < 256; i++) {
                vcpx xc = {z[i].re, -z[i].im};
                vcpx val = s_mul_v(g_t.tw[i][0], g_t.tw[i][1], xc);
                const int fi = 2 * i, ri = 511 - 2 * i;
                vec0[ri] = -val.im;
                vec1[fi] = val.im;
                vec2[ri] = val.re;
                vec3[fi] = val.re;
            }
            for (int i = 0; i < 256; i++) {
                vcpx xc = {z[256 + i].re, -z[256 + i].im};
                vcpx val = s_mul_v(g_t.tw[256 + i][0], g_t.tw[256 + i][1], xc);
                const int fi = 2 * i, ri = 511 - 2 * i;
                vec0[fi] = -val.re;
                vec1[ri] = val.re;
                vec2[fi] = val.im;
                vec3[ri] = val.im;
            }
            /* aac/dsp.rs:105-110, 132-137 (ONLY_LONG, KBD both sides); spec is reused as the PCM staging area */
            for (int i = 0; i < 1024; i++) {
                spec[i] = dl[i] + (out[i] * g_t.kbd_long[i]);
                dl[i] = out[i + 1024] * g_t.kbd_long[1023 - i];
            }
            for (int l = 0; l < W; l++) {
                float *dst = pcm + ((c0 + (size_t)l) * frames_per_chain + t) * 1024;
                for (int i = 0; i < 1024; i++) dst[i] = spec[i][l];
            }
        }
        for (int i = 0; i < 1024; i++)
            for (int l = 0; l < W; l++) delay[(c0 + (size_t)l) * 1024 + (size_t)i] = dl[i][l];
    }
    free(z);
    free(spec);
    free(out);
    free(dl);
}
