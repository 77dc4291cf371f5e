// MP3 Layer III synthesis tail: reorder -> antialias -> hybrid synthesis (36/12-point IMDCT +
// window + overlap) -> frequency inversion -> 32-band polyphase synthesis, batched over chains.
//
// Reference: symphonia-bundle-mp3/src/layer3/hybrid_synthesis.rs:153-485, 559-779;
//            symphonia-bundle-mp3/src/synthesis.rs:158-844; caller layer3/mod.rs:440-476.
//
// MI355X mapping (DESIGN.md "mp3_synth"): a 64-lane wavefront carries TWO chains (one per
// 32-lane half).  Each half walks a segment of consecutive granules of its chain:
//   lane = sub-band for the hybrid stage (the 18-sample overlap of sub-band sb lives in lane sb's
//   registers across granules), lane = time slot for the 18 dct32s, lane = output sample index
//   for the 512-tap window (its 16 window coefficients live in registers).
// LDS per chain: X/S tile (576 f32, the granule in sub-band-major then slot-major order) and the
// polyphase history H[16 + 18][32] holding dct32 outputs of the previous 16 and the current 18
// time slots (the reference's V rows are +-copies of those 32 values, synthesis.rs:247-263).
// Segments other than a chain's first start with a two-granule halo (granule g-2 rebuilds the
// overlap, granule g-1 rebuilds the 15-slot history); both depend only on those granules' inputs.
// Roofline: HBM-bound, 2304 B in + 2304 B out per granule-channel, ~32 kflop -> 7 flop/B.
#include "dsp_device.h"

namespace symaccel {

namespace {

constexpr int kHalf = 32;
constexpr int kXStride = 19;   // sub-band-major tile X[sb][18], padded: 19 is odd -> conflict-free column reads
constexpr int kSStride = 33;   // slot-major tile S[slot][32], padded
constexpr int kHistOld = 16;  // previous slots kept (15 are read by the window, the 16th completes v_vec)
constexpr int kHistRows = kHistOld + 18;

struct Mp3Shared {
    float tile[32 * kXStride > 18 * kSStride ? 32 * kXStride : 18 * kSStride];
    float hist[kHistRows][32];
};

// ---- 36-point IMDCT (Szu-Wei Lee), hybrid_synthesis.rs:559-779 -------------------------------

// sdct_ii_9 (hybrid_synthesis.rs:720-779); writes y[0], y[2], ..., y[16]
__device__ __forceinline__ void sdct_ii_9(const float (&x)[9], float *y, const float *D) {
    const float a01 = x[3] + x[5], a02 = x[3] - x[5], a03 = x[6] + x[2], a04 = x[6] - x[2];
    const float a05 = x[1] + x[7], a06 = x[1] - x[7], a07 = x[8] + x[0], a08 = x[8] - x[0];
    const float a09 = x[4] + a05, a10 = a01 + a03, a11 = a10 + a07, a12 = a03 - a07;
    const float a13 = a01 - a07, a14 = a01 - a03, a15 = a02 - a04, a16 = a15 + a08;
    const float a17 = a04 + a08, a18 = a02 - a08, a19 = a02 + a04, a20 = 2.0f * x[4] - a05;
    const float m1 = D[0] * a06, m2 = D[1] * a12, m3 = D[2] * a13, m4 = D[3] * a14;
    const float m5 = D[0] * a16, m6 = D[4] * a17, m7 = D[5] * a18, m8 = D[6] * a19;
    const float a21 = a20 + m2, a22 = a20 - m2, a23 = a20 + m3;
    const float a24 = m1 + m6, a25 = m1 - m6, a26 = m1 + m7;
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

// dct_iv (hybrid_synthesis.rs:608-660) incl. sdct_ii_18 (:665-716)
__device__ __forceinline__ void dct_iv_18(const float (&x)[18], float (&y)[19], const float *mc) {
    float s[18];
#pragma unroll
    for (int i = 0; i < 18; ++i) s[i] = mc[MP3C_DCT_IV + i] * x[i];
    float even[9], odd[9];
#pragma unroll
    for (int i = 0; i < 9; ++i) even[i] = s[i] + s[17 - i];
    sdct_ii_9(even, &y[0], mc + MP3C_SDCT9_D);
#pragma unroll
    for (int i = 0; i < 9; ++i) odd[i] = mc[MP3C_SDCT18 + i] * (s[i] - s[17 - i]);
    sdct_ii_9(odd, &y[1], mc + MP3C_SDCT9_D);
#pragma unroll
    for (int i = 3; i <= 17; i += 2) y[i] -= y[i - 2];
    y[0] /= 2.0f;
#pragma unroll
    for (int i = 1; i < 18; ++i) y[i] = (y[i] / 2.0f) - y[i - 1];
}

// imdct36 (hybrid_synthesis.rs:571-603): x[18] in place, overlap[18] in/out
__device__ __forceinline__ void imdct36(float (&x)[18], float (&overlap)[18], const float *window, const float *mc) {
    float dct[19];
    dct_iv_18(x, dct, mc);
#pragma unroll
    for (int i = 0; i < 9; ++i) x[i] = overlap[i] + dct[9 + i] * window[i];
#pragma unroll
    for (int i = 9; i < 18; ++i) x[i] = overlap[i] - dct[27 - i - 1] * window[i];
#pragma unroll
    for (int i = 18; i < 27; ++i) overlap[i - 18] = -dct[27 - i - 1] * window[i];
#pragma unroll
    for (int i = 27; i < 36; ++i) overlap[i - 18] = -dct[i - 27] * window[i];
}

// imdct12_win (hybrid_synthesis.rs:363-455)
__device__ __forceinline__ void imdct12_win(float (&x)[18], float (&overlap)[18], const float *window, const float *mc) {
    const float *cos12 = mc + MP3C_COS12;
    float tmp[36];
#pragma unroll
    for (int i = 0; i < 36; ++i) tmp[i] = 0.0f;
#pragma unroll
    for (int w = 0; w < 3; ++w) {
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const float *cl = cos12 + 6 * i, *cr = cos12 + 6 * (i + 3);
            const float yl = (x[w] * cl[0]) + (x[3 + w] * cl[1]) + (x[6 + w] * cl[2]) + (x[9 + w] * cl[3]) +
                             (x[12 + w] * cl[4]) + (x[15 + w] * cl[5]);
            const float yr = (x[w] * cr[0]) + (x[3 + w] * cr[1]) + (x[6 + w] * cr[2]) + (x[9 + w] * cr[3]) +
                             (x[12 + w] * cr[4]) + (x[15 + w] * cr[5]);
            tmp[6 + 6 * w + 3 - i - 1] += -yl * window[3 - i - 1];
            tmp[6 + 6 * w + i + 3] += yl * window[i + 3];
            tmp[6 + 6 * w + i + 6] += yr * window[i + 6];
            tmp[6 + 6 * w + 12 - i - 1] += yr * window[12 - i - 1];
        }
    }
#pragma unroll
    for (int i = 0; i < 18; ++i) {
        x[i] = tmp[i] + overlap[i];
        overlap[i] = tmp[i + 18];
    }
}

// ---- dct32 (B.G. Lee), synthesis.rs:348-844, as the recursion the reference flattens ---------
template <int N>
__device__ __forceinline__ void dct_lee(float *x, const float *mc) {
    if constexpr (N == 2) {
        const float a = x[0] + x[1], b = (x[0] - x[1]) * mc[MP3C_COS1];
        x[0] = a;
        x[1] = b;
    } else {
        constexpr int H = N / 2;
        constexpr int cofs = N == 32 ? MP3C_COS16 : N == 16 ? MP3C_COS8 : N == 8 ? MP3C_COS4 : MP3C_COS2;
        float t[N];
#pragma unroll
        for (int i = 0; i < H; ++i) {
            t[i] = x[i] + x[N - 1 - i];
            t[H + i] = (x[i] - x[N - 1 - i]) * mc[cofs + i];
        }
        dct_lee<H>(t, mc);
        dct_lee<H>(t + H, mc);
#pragma unroll
        for (int i = 0; i < H - 1; ++i) {
            x[2 * i] = t[i];
            x[2 * i + 1] = t[H + i] + t[H + i + 1];
        }
        x[N - 2] = t[H - 1];
        x[N - 1] = t[N - 1];
    }
}

// V-row entries as +-copies of the dct32 output row d (synthesis.rs:247-263).
// first half  V[i]      : i=0 d[16] | 1..15 d[16+i] | 16 -> 0.0 | 17..31 -d[48-i]
// second half V[32 + i] : i=0 -d[16] | 1..15 -d[16-i] | 16 -d[0] | 17..31 -d[i-16]
struct VMap {
    int fidx, sidx;  // source index into d
    int fkind;       // 0: +d, 1: -d, 2: literal 0.0
};
__device__ __forceinline__ VMap vmap(int i) {
    VMap m;
    m.fidx = i == 0 ? 16 : (i < 16 ? 16 + i : (i == 16 ? 0 : 48 - i));
    m.fkind = i < 16 ? 0 : (i == 16 ? 2 : 1);
    m.sidx = i == 0 ? 16 : (i < 16 ? 16 - i : (i == 16 ? 0 : i - 16));
    return m;
}

__global__ __launch_bounds__(64) void mp3_synth_kernel(
    DevTables tb, const float *__restrict__ xr, const symaccel_mp3_side *__restrict__ side, int sr,
    const float *__restrict__ overlap_in, const float *__restrict__ vvec_in, const int32_t *__restrict__ vfront_in,
    float *__restrict__ overlap_out, float *__restrict__ vvec_out, int32_t *__restrict__ vfront_out,
    float *__restrict__ pcm, unsigned n_chains, unsigned granules_per_chain, unsigned seg_len,
    unsigned segs_per_chain) {
    __shared__ Mp3Shared sh[2];
    const int half = (int)threadIdx.x >> 5, hl = (int)threadIdx.x & 31;
    Mp3Shared &S = sh[half];
    const float *mc = tb.mp3_consts;

    const unsigned item = blockIdx.x * 2u + (unsigned)half;
    const bool live = item < n_chains * segs_per_chain;
    const unsigned chain = live ? item / segs_per_chain : 0, seg = live ? item % segs_per_chain : 0;
    const unsigned g_begin = seg * seg_len;
    const unsigned g_end = live ? min(g_begin + seg_len, granules_per_chain) : g_begin;
    const size_t chain_base = (size_t)chain * granules_per_chain;

    // per-lane window coefficients: D[64j + i], D[64j + 32 + i]
    float dw0[8], dw1[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        dw0[j] = mc[MP3C_SYNTH_D + 64 * j + hl];
        dw1[j] = mc[MP3C_SYNTH_D + 64 * j + 32 + hl];
    }
    const VMap vm = vmap(hl);

    // ---- incoming state
    float overlap[18];
    const bool first_seg = g_begin == 0;
    if (live && first_seg) {
#pragma unroll
        for (int i = 0; i < 18; ++i) overlap[i] = overlap_in[(size_t)chain * 576 + 18 * hl + i];
        const int v_front = vfront_in[chain] & 15;
        // history rows 0..15 = time slots -16..-1; slot -m sits in FIFO row (v_front + m) & 15
        const float *vv = vvec_in + (size_t)chain * 1024;
        for (int m = 1; m <= kHistOld; ++m) {
            const float *row = vv + 64 * ((v_front + m) & 15);
            // invert synthesis.rs:247-263: d[16+k] = V[k] (k=0..15), d[0] = -V[48], d[k] = -V[48+k]
            S.hist[kHistOld - m][hl] = hl >= 16 ? row[hl - 16] : -row[48 + hl];
        }
    } else {
#pragma unroll
        for (int i = 0; i < 18; ++i) overlap[i] = 0.0f;
    }
    __syncthreads();

    // halo: g_begin-2 (overlap only), g_begin-1 (history only)
    const long g_first = first_seg ? 0 : (long)g_begin - 2;
    long g_stop = (long)g_end;
    // both halves must run the same number of barrier rounds
    long rounds = g_stop - g_first;
    {
        const long other = __shfl((int)rounds, (int)(threadIdx.x ^ 32u));
        rounds = rounds > other ? rounds : other;
    }

    for (long r = 0; r < rounds; ++r) {
        const long g = g_first + r;
        const bool active = live && g < g_stop;
        const bool need_hist = active && g >= (long)g_begin - 1;  // halo granule g_begin-2 only rebuilds overlap
        const bool emit = active && g >= (long)g_begin;

        int bt = 0, mixed = 0, rzero = 0;
        if (active) {
            const symaccel_mp3_side sd = side[chain_base + (size_t)g];
            bt = sd.block_type;
            mixed = sd.is_mixed ? 1 : 0;
            rzero = sd.rzero > 576 ? 576 : sd.rzero;
        }
        // ---- load + reorder (hybrid_synthesis.rs:153-215) into X[sb][18]
        if (active) {
            const float *src = xr + (chain_base + (size_t)g) * 576;
            int r_start = 576, r_end = 576;
            const int32_t *map = tb.mp3_reorder_map + (size_t)(sr * 2 + mixed) * 576;
            if (bt == SYMACCEL_MP3_SHORT) {
                const int32_t *ends = tb.mp3_reorder_end + (size_t)(sr * 2 + mixed) * 577;
                r_start = ends[0];
                r_end = ends[rzero];
                rzero = rzero > r_end ? rzero : r_end;
            }
#pragma unroll
            for (int q = 0; q < 18; ++q) {
                const int i = hl + 32 * q;
                const int s = (i >= r_start && i < r_end) ? map[i] : i;
                S.tile[(i / 18) * kXStride + (i % 18)] = src[s];
            }
        }
        __syncthreads();
        // ---- antialias (hybrid_synthesis.rs:218-277)
        if (active && !(bt == SYMACCEL_MP3_SHORT && !mixed)) {
            const int sb_limit = bt == SYMACCEL_MP3_SHORT ? 2 : 32;
            int lim = rzero / 18 + 2;
            lim = lim < sb_limit ? lim : sb_limit;
            lim = lim < 32 ? lim : 32;
            rzero = 18 * lim;
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const int task = hl + 32 * q;  // 31 boundaries x 8 butterflies
                const int sb = 1 + (task >> 3), i = task & 7;
                if (sb < 32 && 18 * sb < rzero) {
                    float *lo = &S.tile[(sb - 1) * kXStride + 17 - i], *up = &S.tile[sb * kXStride + i];
                    const float lower = *lo, upper = *up;
                    const float cs = mc[MP3C_CS + i], ca = mc[MP3C_CA + i];
                    *lo = lower * cs - upper * ca;
                    *up = upper * cs + lower * ca;
                }
            }
        }
        __syncthreads();
        // ---- hybrid synthesis (hybrid_synthesis.rs:280-359): lane = sub-band
        float y[18];
        if (active) {
            const int sb = hl;
#pragma unroll
            for (int i = 0; i < 18; ++i) y[i] = S.tile[sb * kXStride + i];
            const int sb_limit = (rzero + 17) / 18;
            const int sb_split = bt == SYMACCEL_MP3_SHORT ? (mixed ? 2 : 0) : 32;
            if (sb >= sb_limit) {
#pragma unroll
                for (int i = 0; i < 18; ++i) {
                    y[i] = overlap[i];
                    overlap[i] = 0.0f;
                }
            } else if (sb < sb_split) {
                const int wi = bt == SYMACCEL_MP3_START ? 1 : (bt == SYMACCEL_MP3_END ? 3 : 0);
                imdct36(y, overlap, mc + MP3C_IMDCT_WIN + 36 * wi, mc);
            } else {
                imdct12_win(y, overlap, mc + MP3C_IMDCT_WIN + 36 * 2, mc);
            }
            // frequency_inversion (hybrid_synthesis.rs:458-485)
            if (sb & 1) {
#pragma unroll
                for (int i = 1; i < 18; i += 2) y[i] = -y[i];
            }
        }
        __syncthreads();  // everyone has read X; reuse the tile as S[slot][sb]
        if (need_hist) {
#pragma unroll
            for (int b = 0; b < 18; ++b) S.tile[b * kSStride + hl] = y[b];
        }
        __syncthreads();
        // ---- 18 x dct32 (synthesis.rs:165-245): lane = time slot
        if (need_hist && hl < 18) {
            float d[32];
#pragma unroll
            for (int i = 0; i < 32; ++i) d[i] = S.tile[hl * kSStride + i];
            dct_lee<32>(d, mc);
#pragma unroll
            for (int i = 0; i < 32; ++i) S.hist[kHistOld + hl][i] = d[i];
        }
        __syncthreads();
        // ---- windowing (synthesis.rs:309-324): lane = sample index i within the 32-sample block
        if (emit) {
            float *dst = pcm + (chain_base + (size_t)g) * 576;
            for (int b = 0; b < 18; ++b) {
                float acc = 0.0f;
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const float df = S.hist[kHistOld + b - 2 * j][vm.fidx];
                    const float v0 = vm.fkind == 0 ? df : (vm.fkind == 1 ? -df : 0.0f);  // V[i]
                    const float v1 = -S.hist[kHistOld + b - 2 * j - 1][vm.sidx];          // V[32 + i]
                    acc += v0 * dw0[j];
                    acc += v1 * dw1[j];
                }
                dst[32 * b + hl] = acc;
            }
        }
        __syncthreads();
        // ---- slide the history: slots 2..17 of this granule become slots -16..-1
        float keep[kHistOld];
        if (need_hist) {
#pragma unroll
            for (int m = 0; m < kHistOld; ++m) keep[m] = S.hist[18 + m][hl];
        }
        __syncthreads();
        if (need_hist) {
#pragma unroll
            for (int m = 0; m < kHistOld; ++m) S.hist[m][hl] = keep[m];
        }
        __syncthreads();
    }

    // ---- outgoing state (only the segment that ends the chain)
    if (live && g_end == granules_per_chain) {
#pragma unroll
        for (int i = 0; i < 18; ++i) overlap_out[(size_t)chain * 576 + 18 * hl + i] = overlap[i];
        // Rebuild v_vec[16][64] + v_front exactly as the reference leaves them: v_front moves back one
        // row per time slot (synthesis.rs:335) and row (v_front + m) & 15 holds slot -m, m = 1..16.
        const int vf0 = vfront_in[chain] & 15;
        const int vf_final = (int)(((unsigned)vf0 + 15u * 18u * granules_per_chain) & 15u);
        float *vv = vvec_out + (size_t)chain * 1024;
        for (int m = 1; m <= kHistOld; ++m) {
            float *row = vv + 64 * ((vf_final + m) & 15);
            const float *d = S.hist[kHistOld - m];
            const float df = d[vm.fidx];
            row[hl] = vm.fkind == 0 ? df : (vm.fkind == 1 ? -df : 0.0f);  // synthesis.rs:247-263
            row[32 + hl] = -d[vm.sidx];
        }
        if (hl == 0) vfront_out[chain] = vf_final;
    }
}

}  // namespace

int launch_mp3(symaccel_ctx *ctx, const float *d_xr, const symaccel_mp3_side *d_side, int sr,
               const float *d_overlap_in, const float *d_vvec_in, const int32_t *d_vfront_in, float *d_overlap_out,
               float *d_vvec_out, int32_t *d_vfront_out, float *d_pcm, size_t n_chains, size_t granules_per_chain) {
    if (granules_per_chain > 0x3fffffffu || n_chains > 0x3fffffffu) return SYMACCEL_ERR_INVALID_ARG;
    unsigned seg = ctx->segment > 0 ? (unsigned)ctx->segment : 32u;
    if (seg < 2) seg = 2;  // the two-granule halo needs segment starts >= 2
    if (seg > granules_per_chain) seg = (unsigned)granules_per_chain;
    const size_t segs = (granules_per_chain + seg - 1) / seg;
    const size_t items = n_chains * segs;
    const size_t grid = (items + 1) / 2;
    if (grid > 0x7fffffffu) return SYMACCEL_ERR_INVALID_ARG;
    hipLaunchKernelGGL(mp3_synth_kernel, dim3((unsigned)grid), dim3(64), 0, ctx->stream, ctx->dev, d_xr, d_side, sr,
                       d_overlap_in, d_vvec_in, d_vfront_in, d_overlap_out, d_vvec_out, d_vfront_out, d_pcm,
                       (unsigned)n_chains, (unsigned)granules_per_chain, seg, (unsigned)segs);
    SYM_GPU(ctx, hipGetLastError());
    return SYMACCEL_OK;
}

}  // namespace symaccel
