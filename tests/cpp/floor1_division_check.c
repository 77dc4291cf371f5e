/* Exhaustive check of the closed form the floor-1 render kernel uses for render_line's integer DDA (vorbis.hip,
 * vorbis_floor1_kernel; reference: symphonia-codec-vorbis/src/floor.rs:785-825):
 *     floor(|dy| * t / adx) == trunc(f32(t) * slope + half),  slope = f32(|dy|) / f32(adx),  half = 0.5f / f32(adx)
 * for every adx <= maxadx (argv[1], default 4096 = the largest n the ABI accepts), |dy| <= 255, 0 <= t < adx, and for the
 * negated slope / half (dy < 0: the conversion truncates towards zero).  Plain IEEE f32 operations, no contraction: the
 * same arithmetic the device executes.  tests/test_vorbis_floor1_closed_form.py builds and runs it.
 * render_point's closed form is NOT exact beyond adx = 4096 (first failure: adx 17019, dx 13456, |dy| 449): the kernel
 * takes the integer form for such posts (vorbis.hip, `wide`).  */
#include <stdio.h>
#include <stdlib.h>
/* render_point (floor.rs:776-782) in step 1 of the same kernel: the divisor and dx = x - x0 belong to the setup, |dy| varies:
 *     floor(|dy| * dx / adx) == trunc(f32(|dy|) * ratio + half),  ratio = f32(dx) / f32(adx),  for |dy| <= 511, dx < adx  */
static long check_render_point(int maxadx) {
    long bad = 0;
    for (int adx = 1; adx <= maxadx; ++adx) {
        float fadx = (float)adx, hinv = 0.5f / fadx;
        for (int dx = 0; dx < adx; ++dx) {
            float ratio = (float)dx / fadx;
            for (int ady = 0; ady <= 511; ++ady) {
                volatile float p = (float)ady * ratio;
                volatile float q = p + hinv;
                volatile float pn = (float)(-ady) * ratio;
                volatile float qn = pn + (-hinv);
                int want = (int)(((long)ady * dx) / adx);
                bad += (int)q != want;
                bad += (int)qn != -want;
            }
        }
    }
    return bad;
}

int main(int argc, char **argv) {
    int maxadx = argc > 1 ? atoi(argv[1]) : 4096;
    long bad = check_render_point(maxadx), total = 0;
    printf("render_point: bad %ld\n", bad);
    for (int adx = 1; adx <= maxadx; ++adx) {
        volatile float fadx = (float)adx;
        float hinv = 0.5f / fadx;
        for (int ady = 0; ady <= 255; ++ady) {
            float slope = (float)ady / fadx;
            for (int t = 0; t < adx; ++t) {
                volatile float p = (float)t * slope;
                volatile float q = p + hinv;
                int s = (int)q;
                int want = (int)(((long)ady * t) / adx);
                bad += s != want;
                ++total;
                // negative direction: trunc toward zero of the negated expression
                volatile float pn = (float)t * (-slope);
                volatile float qn = pn + (-hinv);
                bad += (int)qn != -want;
            }
        }
    }
    printf("maxadx %d total %ld bad %ld\n", maxadx, total, bad);
    /* render_line segments longer than the block: 4096 < adx <= 65535 (x values have 16 bits), of which only the first
     * n <= 4096 lines are rendered; argv[2] = stride over adx (1 = every adx) */
    int stride = argc > 2 ? atoi(argv[2]) : 16;
    long wbad = 0, wtotal = 0;
    if (stride > 0) {
#pragma omp parallel for reduction(+ : wbad, wtotal) schedule(dynamic, 16)
        for (int adx = 4097; adx <= 65535; adx += stride) {
            volatile float fadx = (float)adx;
            float hinv = 0.5f / fadx;
            for (int ady = 0; ady <= 255; ++ady) {
                float slope = (float)ady / fadx;
                for (int t = 0; t < 4096; ++t) {
                    volatile float p = (float)t * slope;
                    volatile float q = p + hinv;
                    int want = (int)(((long)ady * t) / adx);
                    wbad += (int)q != want;
                    volatile float pn = (float)t * (-slope);
                    volatile float qn = pn + (-hinv);
                    wbad += (int)qn != -want;
                    ++wtotal;
                }
            }
        }
        printf("wide segments: stride %d total %ld bad %ld\n", stride, wtotal, wbad);
    }
    return bad != 0 || wbad != 0;
}
