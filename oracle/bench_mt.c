/*
 * bench_mt.c -- multi-threaded timing driver for the CPU oracle (TEST / BENCH INFRASTRUCTURE ONLY).
 *
 * bench.py's `cpu_baseline` leg times the oracle (the C restatement of the reference's scalar,
 * single-threaded path) on every host core.  Doing the fan-out from Python threads serialises on the
 * allocator / GIL at 256 threads, so the fan-out lives here: `threads` pthreads, each with private
 * state and output buffers, each running the same read-only input until `seconds` have elapsed
 * (time-bounded, so an over-subscribed or CPU-quota'd box cannot stretch the run).  Returns the wall
 * time of the timed region (start barrier -> last finish); *total_reps = batches completed by all threads.
 */
#define _GNU_SOURCE
#include <pthread.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include "symoracle.h"

typedef struct job {
    int kind; /* 0 aac, 1 mp3, 2 vorbis, 3 flac, 4 alac, 5 aac (ONLY_LONG / KBD) vectorised across chains (cpu_simd.c) */
    double seconds;
    long *total_reps;
    pthread_mutex_t *lock;
    const void *in0, *in1, *in2;
    size_t n_chains, per_chain, stride_in, stride_out;
    int p0, p1;
    pthread_barrier_t *start;
} job;

static void run_once(const job *j, void *state, void *out) {
    switch (j->kind) {
    case 0:
        so_aac_synth_batch((const float *)j->in0, (const uint8_t *)j->in1, (float *)state, (float *)out, j->n_chains,
                           j->per_chain);
        break;
    case 1: {
        float *ov = (float *)state, *vv = ov + j->n_chains * 576;
        int32_t *vf = (int32_t *)(vv + j->n_chains * 1024);
        so_mp3_synth_batch((const float *)j->in0, (const uint8_t *)j->in1, j->p0, ov, vv, vf, (float *)out, j->n_chains,
                           j->per_chain);
        break;
    }
    case 2: {
        float *ov = (float *)state;
        int32_t *pf = (int32_t *)(ov + j->n_chains * ((size_t)1 << (j->p1 - 1)));
        for (size_t c = 0; c < j->n_chains; ++c) pf[c] = -1;
        so_vorbis_synth_batch(j->p0, j->p1, (const float *)j->in0, j->stride_in, (const uint8_t *)j->in1, pf, ov,
                              (float *)out, j->stride_out, j->n_chains, j->per_chain);
        break;
    }
    case 5:
        so_aac_long_kbd_batch_simd((const float *)j->in0, (float *)state, (float *)out, j->n_chains, j->per_chain);
        break;
    case 4:
        memcpy(out, j->in0, j->n_chains * j->per_chain * sizeof(int32_t));
        so_alac_predict_batch((int32_t *)out, (const uint8_t *)j->in1, (const int32_t *)j->in2, j->n_chains, j->per_chain);
        break;
    default:
        memcpy(out, j->in0, j->n_chains * j->per_chain * sizeof(int32_t));
        so_flac_restore_batch((int32_t *)out, (const uint8_t *)j->in1, (const int32_t *)j->in2, j->n_chains, j->per_chain);
        break;
    }
}

static double now_s(void) {
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

static void *worker(void *arg) {
    const job *j = (const job *)arg;
    size_t state_bytes, out_bytes;
    switch (j->kind) {
    case 0:
    case 5: state_bytes = j->n_chains * 1024 * 4; out_bytes = j->n_chains * j->per_chain * 1024 * 4; break;
    case 1: state_bytes = j->n_chains * (576 + 1024 + 1) * 4; out_bytes = j->n_chains * j->per_chain * 576 * 4; break;
    case 2: state_bytes = j->n_chains * (((size_t)1 << (j->p1 - 1)) + 1) * 4; out_bytes = j->n_chains * j->stride_out * 4; break;
    default: state_bytes = 4; out_bytes = j->n_chains * j->per_chain * 4; break;
    }
    void *state = calloc(1, state_bytes), *out = calloc(1, out_bytes);
    run_once(j, state, out); /* warm: tables, page faults */
    pthread_barrier_wait(j->start);
    const double deadline = now_s() + j->seconds;
    long reps = 0;
    do {
        run_once(j, state, out);
        ++reps;
    } while (now_s() < deadline);
    pthread_mutex_lock(j->lock);
    *j->total_reps += reps;
    pthread_mutex_unlock(j->lock);
    free(state);
    free(out);
    return NULL;
}

/* kind: 0 aac (in0 coeffs, in1 side), 1 mp3 (in0 xr, in1 side, p0 sample_rate_idx), 2 vorbis (in0 spectra,
 * in1 flags, p0/p1 bs exps, strides), 3 flac (in0 buf, in1 desc, in2 coeffs; per_chain = blocksize). */
double so_bench_mt(int kind, int threads, double seconds, long *total_reps, const void *in0, const void *in1, const void *in2,
                   size_t n_chains, size_t per_chain, size_t stride_in, size_t stride_out, int p0, int p1) {
    pthread_barrier_t start;
    pthread_barrier_init(&start, NULL, (unsigned)threads + 1);
    pthread_mutex_t lock = PTHREAD_MUTEX_INITIALIZER;
    *total_reps = 0;
    job j = {kind, seconds, total_reps, &lock, in0, in1, in2, n_chains, per_chain, stride_in, stride_out, p0, p1, &start};
    pthread_t *th = (pthread_t *)malloc(sizeof(pthread_t) * (size_t)threads);
    for (int i = 0; i < threads; ++i) pthread_create(&th[i], NULL, worker, &j);
    pthread_barrier_wait(&start);
    const double t0 = now_s();
    for (int i = 0; i < threads; ++i) pthread_join(th[i], NULL);
    const double dt = now_s() - t0;
    free(th);
    pthread_barrier_destroy(&start);
    return dt;
}
