// Vorbis synthesis for block-size pairs whose LONG blocks have 4096 or 8192 samples (bs1_exp 12 / 13): DspChannel::synth
// (symphonia-codec-vorbis/src/dsp.rs:68-145) with Imdct::new(bs / 2) per block size (vorbis/lib.rs:123-124,
// symphonia-core/src/dsp/mdct.rs:67-146).
//
// MI355X mapping: a WORKGROUP of four wavefronts walks a chain segment; a block of 4096 / 8192 samples (an FFT of P = 1024 / 2048
// points = S = 2 / 4 sub-transforms of 512 points) is taken by the four wavefronts TOGETHER, the scheme of fft4096_wg_kernel
// (imdct_big.hip): the block's lines are loaded coalesced (16 B per lane) and staged in an LDS area with one pair of padding per S
// pairs -- the gathers' lane stride becomes 2 (S + 1) dwords: conflict-free --; wavefront q < S gathers the z-indices S m + rev(q),
// pre-twiddles them and runs ONE 512-point register-pass transform (sixteen data registers); the results meet in the same area,
// every lane runs the last log2 S radix-2 stages and the post-twiddle on two positions p (operands one from each sub-transform),
// the left half of the Imdct output goes through the area in natural order for the overlap-add (all 256 lanes, 16-byte accesses),
// the right half is written straight into `overlap` (a workgroup-wide LDS array with exactly the reference's contents).
// Runs of blocks of up to 2048 samples in such a stream (the pair's short blocks) are groups of vorbis_wave2.hip, up to four at a
// time, one per wavefront, against the same `overlap`.  The kernel this replaces ran the S sub-transforms one after the other in ONE
// wavefront (vorbis_big_block, vorbis_wave2.hip): 64 / 128 data registers + hoisted table addresses, one wavefront per SIMD for
// 8192-sample blocks -- 2.2 / 1.3 TB/s.
// HBM traffic per channel-block: 4 * (n / 2) B in + 4 * (prev_n + n) / 4 B out (+ one halo block per segment).
#include "imdct_wave.h"
#include "vorbis_offsets.h"

namespace symaccel {

namespace {

constexpr int kWgWaves = 4;
// floats of FFT / group work area per wavefront: a group's 2048 output samples, or the exchange layouts of fft_wave_multi (T1M: 722
// complex values, T2: 520) -- less than the kWaveLds of the kernels that run the 512-point T1 layout, which makes room for the
// dB table of the fused floor x residue load inside this kernel's 2 x 80 KiB
constexpr int kWgWaveLds = 2176;
static_assert(2 * (7 + 462 + 36 * 7 + 1) <= kWgWaveLds && 2048 <= kWgWaveLds, "work area");
// SYM_VORBIS_WG_SHARED (build knob) 1: the wavefronts' FFT exchange areas and the short-block groups' work areas live INSIDE the
// block's staging area (two more barriers per block: between the gathers and the first exchange, between the merge's reads and
// the left half's writes; three groups of short blocks in flight instead of four): 46 KiB of LDS instead of 80, three workgroups
// per CU instead of two.  0: separate arrays, as measured in rounds 3-4 (profiles/r04m_vorbis_big_blocks.txt).
// (default in symaccel_internal.h)
constexpr int kWgGroupWaves = SYM_VORBIS_WG_SHARED ? 3 : 4;          // groups of short blocks in flight
constexpr int kWgFftWork = SYM_VORBIS_WG_SHARED ? 1456 : kWgWaveLds;  // floats of FFT exchange area per wavefront (T1M: 1444)
static_assert(2 * (7 + 462 + 36 * 7 + 1) <= kWgFftWork && 1024 <= kWgFftWork, "exchange area");

template <int S>
__device__ __forceinline__ int wgv_slot(int e) { return e + e / S; }

// One block of bs = 2048 S samples, by the whole workgroup.  `area`: >= 512 S + 512 complex slots (staging with padding, later the
// left half of the output: 1024 S floats); `work`: the four FFT work areas (kWgWaveLds floats each; a sub-transform's results are
// exchanged through its own work area, idle by then); `ldsf`: this wavefront's work area; `ovl`: dsp.rs:125.  Four barriers.
// Every table value the block needs is requested FIRST, in front of the staging: the loads then travel while the workgroup stages
// and meets, instead of being exposed one barrier interval at a time (the kernel is bound by the latency of a block's steps).
// Barriers inside: every wavefront of the workgroup must call this with the same arguments.
// `keep_below`: overlap[k] for k < keep_below is left as it is (the stale-state rebuild after a short tail; 0 otherwise).
template <int S, int FUSED, class LT>
__device__ __forceinline__ void vorbis_wg_block(const DevTables &tb, const float *__restrict__ spec, const float *__restrict__ res,
                                                const cpx *__restrict__ tw_g, const float *__restrict__ win_long,
                                                const float *__restrict__ win_short, int flag, int pflag, int bs0, int bs1, c32 *area,
                                                float *work, float *ldsf, float *ovl, const LT &lt, int tid, float *__restrict__ o, bool emit,
                                                int keep_below, float4 (&pre)[S], bool pre_valid, const float *__restrict__ next_spec,
                                                const float *__restrict__ next_res, const float *dbt) {
    static_assert(S == 2 || S == 4, "two or four 512-point sub-transforms");
    // (what the routine derives from the thread index -- 64-bit load / store addresses, LDS offsets -- would be kept across the
    // block loop as loop invariants, and the routine has no registers for them: recomputed per block, a dozen VALU instructions)
#if defined(__HIP_DEVICE_COMPILE__)
    asm volatile("" : "+v"(tid));
#endif
    constexpr int P = 512 * S, N = 2 * P;  // FFT points; lines = samples of one half of the block
    const int lane = tid & 63, wave = tid >> 6;
    float *af = reinterpret_cast<float *>(area);
    float *left = af;  // (the staging area again, once the gathers are done: see the barriers below)
    c32 *lds = reinterpret_cast<c32 *>(ldsf);
    const int rq = S == 2 ? (wave & 1) : (((wave & 1) << 1) | ((wave >> 1) & 1));
    // the pre-twiddles first: they are needed right behind the first barrier
    c32 twp[8];
    if (wave < S) {
#pragma unroll
        for (int s = 0; s < 8; ++s) twp[s] = ld_c(tw_g + S * (lane + 64 * s) + rq);
    }
    const bool same = pflag == flag;
    const float *win = (flag && pflag) ? win_long : win_short;  // dsp.rs:83
    // ---- the block's N lines, coalesced, times the residue (lib.rs:289-291: *f *= r), staged as pairs e = (line 2 e, line 2 e + 1)
    // (`pre`: the lines, already multiplied, if the block before prefetched them; `next_spec`: the block to prefetch now, or null.
    // With the table loads out of the way the block's own lines are the longest wait left in its chain of steps.)
    auto fetch = [&](const float *sp_, const float *rs_, float4 (&v)[S]) {
        const float4 *s4 = reinterpret_cast<const float4 *>(sp_);
        const float4 *r4 = reinterpret_cast<const float4 *>(rs_);
#pragma unroll
        for (int j = 0; j < S; ++j) v[j] = ld_stream(s4 + tid + 256 * j);
        if constexpr (FUSED == 1) {
#pragma unroll
            for (int j = 0; j < S; ++j) {
                const float4 q = ld_stream(r4 + tid + 256 * j);
                v[j].x *= q.x;
                v[j].y *= q.y;
                v[j].z *= q.z;
                v[j].w *= q.w;
            }
        }
        if constexpr (FUSED == 2) {
            const uint32_t *ry = reinterpret_cast<const uint32_t *>(rs_);
#pragma unroll
            for (int j = 0; j < S; ++j) {
                mul_floor_y(v[j], ry[tid + 256 * j], dbt);
                __builtin_amdgcn_sched_barrier(0);  // (one float4's table values at a time: 256 VGPRs are in use)
            }
        }
    };
    if (!pre_valid) fetch(spec, res, pre);
#pragma unroll
    for (int j = 0; j < S; ++j) {
        const int e = 2 * (tid + 256 * j);  // (e and e + 1 share a group of S: adjacent slots)
        area[wgv_slot<S>(e)] = c32{pre[j].x, pre[j].y};
        area[wgv_slot<S>(e) + 1] = c32{pre[j].z, pre[j].w};
    }
    if (next_spec) fetch(next_spec, next_res, pre);  // lands during this block's transform
    wg_sync_lds();
    // ---- wavefront q < S: the 512-point transform of the z-indices S m + rev(q) (mdct.rs:81-88 on the way in)
    c32 x[8];
    if (wave < S) {
#pragma unroll
        for (int s = 0; s < 8; ++s) {
            const int e = S * (lane + 64 * s) + rq;
            x[s] = pre_twiddle(af[2 * wgv_slot<S>(e)], af[2 * wgv_slot<S>(P - 1 - e) + 1], twp[s]);
        }
    }
    if constexpr (SYM_VORBIS_WG_SHARED) wg_sync_lds();  // every gather is done: the exchange areas (inside the staging area) may be written
    // the twiddles of the last stages and of the post-twiddle: requested here, they travel during the sub-transform
    c32 tpost[2][S], w1[2], w2a[2], w2b[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int p = 128 * wave + 64 * i + lane;
        w1[i] = ld_c(tb.fft_merge + 480 + p);  // W_1024[p]
        if constexpr (S == 4) {
            w2a[i] = ld_c(tb.fft_merge + 992 + p);        // W_2048[p]
            w2b[i] = ld_c(tb.fft_merge + 992 + 512 + p);  // W_2048[p + 512]
        }
#pragma unroll
        for (int j = 0; j < S; ++j) tpost[i][j] = ld_c(tw_g + p + 512 * j);
    }
    if (wave < S) fft_wave_multi(x, lane, lds, lt, 9);  // x[B] = position 64 B + lane of block `wave`
    // the common window case (the block and its predecessor have this size): every lane's window values, N / 1024 x (4 + 4); they
    // travel during the exchange and the last stages
    float4 wfv[N / 1024], wrv[N / 1024];
    if (emit && same) {
#pragma unroll
        for (int it = 0; it < N / 1024; ++it) {
            const unsigned k = 4u * (unsigned)tid + 1024u * (unsigned)it;
            wfv[it] = *reinterpret_cast<const float4 *>(win + k);
            wrv[it] = *reinterpret_cast<const float4 *>(win + ((unsigned)(N - 4) - k));
        }
    }
    // the sub-transform's results go into its own (now idle) work area: no barrier between the gathers and the exchange
    if (wave < S) {
#pragma unroll
        for (int B = 0; B < 8; ++B) lds[64 * B + lane] = x[B];
    }
    wg_sync_lds();  // every sub-transform is published (and every gather done: the staging area is free)
    // ---- the last log2 S stages (no_simd.rs:247-279) and the post-twiddle (mdct.rs:104 / 123) on positions p = 128 wave + 64 i + lane
    c32 val[2][S];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int p = 128 * wave + 64 * i + lane;
#pragma unroll
        for (int j = 0; j < S; ++j) val[i][j] = reinterpret_cast<const c32 *>(work + (size_t)j * kWgFftWork)[p];
    }
    if constexpr (SYM_VORBIS_WG_SHARED) wg_sync_lds();  // every operand is in registers: the left half may overwrite the exchange areas
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        if constexpr (S == 2) {
            bfly(val[i][0], val[i][1], c_mul(val[i][1], w1[i]));  // X[p], X[p + 512]
        } else {
            bfly(val[i][0], val[i][1], c_mul(val[i][1], w1[i]));
            bfly(val[i][2], val[i][3], c_mul(val[i][3], w1[i]));
            bfly(val[i][0], val[i][2], c_mul(val[i][2], w2a[i]));  // W_2048[p]:       X[p], X[p + 1024]
            bfly(val[i][1], val[i][3], c_mul(val[i][3], w2b[i]));  // W_2048[p + 512]: X[p + 512], X[p + 1536]
        }
#pragma unroll
        for (int j = 0; j < S; ++j) val[i][j] = post_twiddle(val[i][j], tpost[i][j]);  // bin k = p + 512 j
    }
    // ---- left half = vec0 | vec1 (mdct.rs:94-137: every bin gives one sample to each of the four vectors), natural order
    constexpr int n4 = P / 2;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < S; ++j) {
            const int k = 128 * wave + 64 * i + lane + 512 * j;
            const c32 v = val[i][j];
            if (j < S / 2) {  // k < n4 = 256 S <=> j < S / 2: a compile-time branch
                left[P - 1 - 2 * k] = -v.y;  // vec0[ri]
                left[P + 2 * k] = v.y;       // vec1[fi]
            } else {
                const int i2 = k - n4;
                left[2 * i2] = -v.x;               // vec0[fi]
                left[P + P - 1 - 2 * i2] = v.x;    // vec1[ri]
            }
        }
    wg_sync_lds();
    // ---- overlap-add of the left half (dsp.rs:85-122), 16 bytes per lane
    if (emit && same) {  // dsp.rs:85-90 over bs / 2 samples, window values already here
#pragma unroll
        for (int it = 0; it < N / 1024; ++it) {
            const unsigned k = 4u * (unsigned)tid + 1024u * (unsigned)it;
            const float4 y = *reinterpret_cast<const float4 *>(left + k), a = *reinterpret_cast<const float4 *>(ovl + k);
            const float4 wf = wfv[it], wr = wrv[it];
            st_stream(reinterpret_cast<float4 *>(o + k),
                      make_float4(a.x * wr.w + y.x * wf.x, a.y * wr.z + y.y * wf.y, a.z * wr.y + y.z * wf.z, a.w * wr.x + y.w * wf.w));
        }
    } else if (emit) {
        const unsigned start = (unsigned)(bs1 - bs0) / 4u;
        if (pflag && !flag) {  // long -> short with a short block of this size: the unity part of the old overlap goes out first (dsp.rs:97)
            for (unsigned k = 4u * (unsigned)tid; k < start; k += 1024u) st_stream(reinterpret_cast<float4 *>(o + k), *reinterpret_cast<const float4 *>(ovl + k));
        }
        for (unsigned k = 4u * (unsigned)tid; k < (unsigned)N; k += 1024u) {
            const float4 y = *reinterpret_cast<const float4 *>(left + k);
            if (pflag) {  // long -> short (dsp.rs:91-106): out[start + k] = overlap[start + k] * ws[len-1-k] + imdct[k] * ws[k]
                const unsigned len = (unsigned)bs0 / 2u;
                const float4 a = *reinterpret_cast<const float4 *>(ovl + start + k);
                const float4 wf = *reinterpret_cast<const float4 *>(win + k), wr = *reinterpret_cast<const float4 *>(win + (len - 4u - k));
                st_stream(reinterpret_cast<float4 *>(o + start + k),
                          make_float4(a.x * wr.w + y.x * wf.x, a.y * wr.z + y.y * wf.y, a.z * wr.y + y.z * wf.z, a.w * wr.x + y.w * wf.w));
            } else {  // short -> long (dsp.rs:107-122): imdct[start .. end) laps with overlap[0 .. len), imdct[end ..) is copied
                const unsigned len = (unsigned)bs0 / 2u, end = start + len;
                if (k >= start && k < end) {
                    const unsigned j = k - start;
                    const float4 a = *reinterpret_cast<const float4 *>(ovl + j);
                    const float4 wf = *reinterpret_cast<const float4 *>(win + j), wr = *reinterpret_cast<const float4 *>(win + (len - 4u - j));
                    st_stream(reinterpret_cast<float4 *>(o + j),
                              make_float4(a.x * wr.w + y.x * wf.x, a.y * wr.z + y.y * wf.y, a.z * wr.y + y.z * wf.z, a.w * wr.x + y.w * wf.w));
                } else if (k >= end) {
                    st_stream(reinterpret_cast<float4 *>(o + (len + (k - end))), y);
                }
            }
        }
    }
    wg_sync_lds();  // `overlap` and the left half have been read
    // ---- right half = vec2 | vec3 -> overlap[0 .. bs / 2) (dsp.rs:125)
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < S; ++j) {
            const int k = 128 * wave + 64 * i + lane + 512 * j;
            const c32 v = val[i][j];
            int a0, a1;
            float f0, f1;
            if (j < S / 2) {
                a0 = P - 1 - 2 * k;  // vec2[ri]
                a1 = P + 2 * k;      // vec3[fi]
                f0 = f1 = v.x;
            } else {
                const int i2 = k - n4;
                a0 = 2 * i2;              // vec2[fi]
                a1 = P + P - 1 - 2 * i2;  // vec3[ri]
                f0 = f1 = v.y;
            }
            if (a0 >= keep_below) ovl[a0] = f0;
            if (a1 >= keep_below) ovl[a1] = f1;
        }
    // (no barrier here: the next reader of `overlap` is behind one -- a following block's overlap-add comes after its own barriers,
    // a following group starts with one)
}

// MAXE1: the long-block exponent (12 / 13).  BIG0: 0 = the short blocks have at most 2048 samples (groups, wavefront 0), 2 / 4 = they
// have 4096 / 8192 samples themselves (S of their cooperative transform).
template <int FUSED, int MAXE1, int BIG0>
__global__ __launch_bounds__(64 * kWgWaves, SYM_VORBIS_WG_SHARED ? 3 : 2) void vorbis_synth_wg_kernel(
    DevTables tb, int e0, int e1, const cpx *__restrict__ tw_short, const cpx *__restrict__ tw_long,
    const float *__restrict__ win_short, const float *__restrict__ win_long, const float *__restrict__ spectra,
    const float *__restrict__ residue, size_t spec_stride, const uint8_t *__restrict__ flags, const int32_t *__restrict__ prev_flag_in,
    int32_t *__restrict__ prev_flag_out, const float *__restrict__ overlap_in, float *__restrict__ overlap_out,
    float *__restrict__ pcm, size_t pcm_stride, unsigned nb, unsigned seg_len, unsigned segs_per_chain) {
    constexpr int S1 = (1 << MAXE1) / 2048;  // 2 / 4
    constexpr int PL = 512 * S1;
#if SYM_VORBIS_WG_SHARED
    // staging (P + P / S slots, S >= 2) -> the wavefronts' exchange areas -> left half; or the work areas of three groups of short blocks
    constexpr int kAreaFloats = 2 * (PL + PL / 2) > kWgGroupWaves * kWgWaveLds ? 2 * (PL + PL / 2) : kWgGroupWaves * kWgWaveLds;
    static_assert(kWgWaves * kWgFftWork <= kAreaFloats, "exchange areas inside the staging area");
    __shared__ __attribute__((aligned(16))) float area_f[kAreaFloats];
    c32 *area = reinterpret_cast<c32 *>(area_f);
    float *wave_base = area_f;
#else
    __shared__ __attribute__((aligned(16))) c32 area[PL + PL / 2];  // staging (P + P / S slots, S >= 2) -> exchange -> left half
    __shared__ __attribute__((aligned(16))) float wave_lds_arr[kWgWaves][kWgWaveLds];
    float *wave_base = &wave_lds_arr[0][0];
#endif
    __shared__ float db_lds[FUSED == 2 ? 256 : 1];  // FLOOR1_INVERSE_DB_TABLE for the fused floor x residue load
    if constexpr (FUSED == 2) db_lds[threadIdx.x] = tb.vorbis_floor1_db[threadIdx.x];  // (256 threads; the barriers of the first block come first)
    __shared__ __attribute__((aligned(16))) float ovl[(1 << MAXE1) / 2];
    __shared__ __attribute__((aligned(16))) c32 lane_tab[kLaneTabComplex];  // the FFT's lane twiddles, read at the point of use (31 VGPRs)
    fill_lane_tables_lds(tb, lane_tab, (int)threadIdx.x, 64 * kWgWaves);
    const int bs0 = 1 << e0, bs1 = 1 << e1;
    const int tid = (int)threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    float *ldsf = wave_base + wave * kWgWaveLds;    // a group's work area (wave < kWgGroupWaves)
    float *fft_ldsf = wave_base + wave * kWgFftWork;  // the wavefront's FFT exchange area inside a cooperative block
    c32 *lds = reinterpret_cast<c32 *>(ldsf);
    const unsigned item = blockIdx.x;
    const unsigned chain = item / segs_per_chain, seg = item % segs_per_chain;
    const unsigned b_begin = seg * seg_len, b_end = min(b_begin + seg_len, nb);
    const uint8_t *f = flags + (size_t)chain * nb;
    const float *sp = spectra + (size_t)chain * spec_stride;
    const float *rp = res_at<FUSED>(residue, (size_t)chain * spec_stride);
    float *out = pcm + (size_t)chain * pcm_stride;
    const int pf0 = prev_flag_in[chain];
    const LaneTablesLds lt = lane_tables_lds(tb, lane_tab, lane);

    // overlap (dsp.rs:125): the caller's state at a chain's start; zero in front of a later segment, whose halo block rebuilds
    // the part the next block reads
    for (int k = 4 * tid; k < bs1 / 2; k += 1024)
        *reinterpret_cast<float4 *>(ovl + k) = b_begin == 0 ? *reinterpret_cast<const float4 *>(overlap_in + (size_t)chain * (size_t)(bs1 / 2) + k)
                                                             : make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    wg_sync_lds();
    bool hi_fresh = b_begin == 0 || e0 == e1;  // overlap[bs0/2 .. bs1/2) is what the reference would hold at this point

    // Block flags as wave-uniform bit masks (every wavefront builds the same ones: the walk is identical in all four)
    const long b_first = b_begin == 0 ? 0 : (long)b_begin - 1;  // the halo block rebuilds the overlap only
    auto load_mask = [&](long base) -> unsigned long long {
        const long idx = base + lane;
        return __ballot(idx < (long)b_end && f[idx] != 0);
    };
    long wbase = b_first;
    unsigned long long m0 = load_mask(wbase), m1 = load_mask(wbase + 64);
    const int cap0 = e0 > 11 ? 1 : 2048 >> e0;  // blocks per group (one of 4096 / 8192 samples)
    auto group_at = [&](long bb, int &flag_out) -> int {
        if (bb >= (long)b_end) {
            flag_out = 1;
            return 0;
        }
        while (bb - wbase >= 64) {
            m0 = m1;
            wbase += 64;
            m1 = load_mask(wbase + 64);
        }
        const int off = (int)(bb - wbase);
        unsigned long long w = off == 0 ? m0 : ((m0 >> off) | (m1 << (64 - off)));
        flag_out = (int)(w & 1ull);
        if (flag_out) w = ~w;
        int run = w ? __builtin_ctzll(w) : 64;
        const long left = (long)b_end - bb;
        if ((long)run > left) run = (int)left;
        const int cap = flag_out ? 1 : cap0;
        return run < cap ? run : cap;
    };

    long b = b_first;
    int flag = 1;
    int glen = group_at(b, flag);
    // flag of the block before b (lib.rs:298: the first block of a stream pairs with itself)
    int pflag = b == 0 ? (pf0 < 0 ? flag : (pf0 ? 1 : 0)) : (f[b - 1] ? 1 : 0);
    // packed offsets of the segment's first block: counted from the flags by every wavefront (vorbis_offsets.h), running sums after that
    const VorbisPackedAt at0 = vorbis_packed_at<false>(f, b, pf0, bs0, bs1, lane);
    uint32_t os_cur = at0.spec, op_cur = at0.pcm;

    // The walk, then -- at a chain's end, if no long block refreshed overlap[bs0/2 .. bs1/2) in this segment -- ONE more trip through
    // the same loop body for the most recent long block in front of the segment (`rebuild`: nothing is emitted, only the upper part
    // of `overlap` is taken from it; vorbis_wave2.hip has the reasoning).
    bool rebuild = false;
    int keep_below = 0;
    float4 pre[S1];  // the next long block's lines, requested while the block before it is transformed
    bool pre_valid = false;
#pragma unroll
    for (int j = 0; j < S1; ++j) pre[j] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    while (true) {
        if (b >= (long)b_end) {
            if (rebuild || b_end != nb || hi_fresh) break;
            long bl = -1;
            for (long base = ((long)b_begin - 1) & ~63l; base >= 0; base -= 64) {
                const long idx = base + lane;
                const unsigned long long m = __ballot(idx < (long)b_begin && f[idx] != 0);
                if (m) {
                    bl = base + 63 - __builtin_clzll(m);
                    break;
                }
            }
            if (bl < 0) {
                for (int k = bs0 / 2 + 4 * tid; k < bs1 / 2; k += 1024)
                    *reinterpret_cast<float4 *>(ovl + k) = *reinterpret_cast<const float4 *>(overlap_in + (size_t)chain * (size_t)(bs1 / 2) + k);
                wg_sync_lds();
                break;
            }
            rebuild = true;
            keep_below = bs0 / 2;
            b = bl;
            glen = 1;
            flag = pflag = 1;
            os_cur = vorbis_sizes_before<false>(f, bl, bs0, bs1, lane) / 2u;
        }
        const int e = flag ? e1 : e0, bs = 1 << e, logp = e - 2, P = 1 << logp;
        long nb_next = b + glen;
        int flag_next = 1;
        int glen_next = rebuild ? 0 : group_at(nb_next, flag_next);
        uint32_t os_next = os_cur + ((uint32_t)glen << (e - 1));
        hi_fresh = hi_fresh || flag;
        const bool emit = !rebuild && b >= (long)b_begin;
        if (e > 11) {
            // ---- one block of 4096 / 8192 samples, by the whole workgroup
            const cpx *twg = flag ? tw_long : tw_short;
            if (BIG0 == 0 || BIG0 == S1 || flag) {
                // (a block of the long size; the next one is prefetched if it has the same size -- the common case)
                const bool next_same = glen_next > 0 && (flag_next ? e1 : e0) == e;
                vorbis_wg_block<S1, FUSED>(tb, sp + os_cur, res_at<FUSED>(rp, os_cur), twg, win_long, win_short, flag, pflag, bs0, bs1, area,
                                           wave_base, fft_ldsf, ovl, lt, tid, out + op_cur, emit, keep_below, pre, pre_valid,
                                           next_same ? sp + os_next : nullptr, next_same ? res_at<FUSED>(rp, os_next) : nullptr, db_lds);
                pre_valid = next_same;
            } else if constexpr (BIG0 != 0 && BIG0 != S1) {
                float4 tmp[BIG0];
                vorbis_wg_block<BIG0, FUSED>(tb, sp + os_cur, res_at<FUSED>(rp, os_cur), twg, win_long, win_short, flag, pflag, bs0, bs1, area,
                                             wave_base, fft_ldsf, ovl, lt, tid, out + op_cur, emit, keep_below, tmp, false, nullptr, nullptr, db_lds);
                pre_valid = false;
            }
            if (rebuild) break;
            op_cur += (uint32_t)((pflag ? bs1 : bs0) + bs) >> 2;
        } else if constexpr (BIG0 == 0) {
            // ---- a run of short blocks: up to four GROUPS of 2048 / bs blocks (vorbis_wave2.hip's group: one pass of the multi-
            // transform FFT), one per wavefront.  The groups' transforms and the overlap-adds inside a group are independent; a
            // group's FIRST block laps with the last block of the group before it, which sits in the neighbour's work area.
            int gl0 = glen, gl1 = 0, gl2 = 0, gl3 = 0, ng = 1, run_blocks = glen;
            while (ng < kWgGroupWaves && glen_next > 0 && flag_next == 0) {
                if (ng == 1) gl1 = glen_next; else if (ng == 2) gl2 = glen_next; else gl3 = glen_next;
                ++ng;
                run_blocks += glen_next;
                glen_next = group_at(b + run_blocks, flag_next);
            }
            nb_next = b + run_blocks;
            os_next = os_cur + ((uint32_t)run_blocks << (e - 1));
            // this wavefront's group: `before` blocks of the run in front of it, `mine` blocks, its predecessor's `prev_len`
            const int before = wave == 0 ? 0 : (wave == 1 ? gl0 : (wave == 2 ? gl0 + gl1 : gl0 + gl1 + gl2));
            const int mine = wave == 0 ? gl0 : (wave == 1 ? gl1 : (wave == 2 ? gl2 : gl3));
            const int prev_len = wave == 1 ? gl0 : (wave == 2 ? gl1 : gl2);
            const uint32_t first_len = (uint32_t)((pflag ? bs1 : bs0) + bs) >> 2;
            const int half = bs >> 1;
            auto ola = [&](float *oo, const float *ov, const float *y, const float *win, int len, bool em) {
                if (!em) return;
                for (int k = 4 * lane; k < len; k += 256) {
                    const float4 a = *reinterpret_cast<const float4 *>(ov + k), bq = *reinterpret_cast<const float4 *>(y + k);
                    const float4 wf = *reinterpret_cast<const float4 *>(win + k);
                    const float4 wr = *reinterpret_cast<const float4 *>(win + len - 4 - k);
                    st_stream(reinterpret_cast<float4 *>(oo + k),
                              make_float4(a.x * wr.w + bq.x * wf.x, a.y * wr.z + bq.y * wf.y, a.z * wr.y + bq.z * wf.z, a.w * wr.x + bq.w * wf.w));
                }
            };
            wg_sync_lds();  // a block before this run has finished writing `overlap` and reading the work areas
            if (wave < ng) {
                const c32 *tw = reinterpret_cast<const c32 *>(tw_short);
                const uint32_t os_mine = os_cur + ((uint32_t)before << (e - 1));
                float4 v[4];
                const size_t valid = (size_t)mine << (e - 1);
                multi_fetch(sp + os_mine, valid, lane, v);
                if constexpr (FUSED == 2) {
                    const uint32_t *ry = reinterpret_cast<const uint32_t *>(res_at<2>(rp, os_mine));
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const int i4 = lane + 64 * q;
                        mul_floor_y(v[q], (size_t)(4 * i4) < valid ? ry[i4] : 0u, db_lds);
                    }
                }
                if constexpr (FUSED == 1) {
                    float4 r[4];
                    multi_fetch(rp + os_mine, valid, lane, r);
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        v[q].x *= r[q].x;
                        v[q].y *= r[q].y;
                        v[q].z *= r[q].z;
                        v[q].w *= r[q].w;
                    }
                }
#pragma unroll
                for (int q = 0; q < 4; ++q) reinterpret_cast<float4 *>(ldsf)[lane + 64 * q] = v[q];
                wave_sync();
                c32 z[8];
                {
                    const int gbits = logp - 3, G = 1 << gbits;
                    const int T = lane >> gbits, u = lane & (G - 1);
                    const float *sT = ldsf + ((size_t)T << (logp + 1));
#pragma unroll
                    for (int s = 0; s < 8; ++s) {
                        const int i = u + (s << gbits);
                        const float2 pr = *reinterpret_cast<const float2 *>(sT + 2 * i);
                        z[s] = pre_twiddle(pr.x, sT[2 * P - 1 - 2 * i], tw[i]);
                    }
                }
                wave_sync();
                fft_wave_multi(z, lane, lds, lt, logp);
                multi_post_twiddle(z, lane, logp, tw, ldsf);  // block i of the group: ldsf[i * bs .. (i + 1) * bs)
                wave_sync();
                // blocks 1 .. mine - 1 lap with their predecessor inside the work area (equal sizes: dsp.rs:85-90); all of them are
                // emitted: only a halo block precedes b_begin, and a halo block is the run's first
                if (mine > 1) {
                    const int per_block = half >> 2;  // float4 chunks per block
                    for (int c = lane; c < (mine - 1) * per_block; c += 64) {
                        const int i = 1 + c / per_block, k = 4 * (c % per_block);
                        const float *ov = ldsf + (size_t)(i - 1) * bs + half, *y = ldsf + (size_t)i * bs;
                        const float4 a = *reinterpret_cast<const float4 *>(ov + k), bq = *reinterpret_cast<const float4 *>(y + k);
                        const float4 wf = *reinterpret_cast<const float4 *>(win_short + k);
                        const float4 wr = *reinterpret_cast<const float4 *>(win_short + half - 4 - k);
                        st_stream(reinterpret_cast<float4 *>(out + op_cur + first_len + (size_t)(before + i - 1) * half + k),
                                  make_float4(a.x * wr.w + bq.x * wf.x, a.y * wr.z + bq.y * wf.y, a.z * wr.y + bq.z * wf.z, a.w * wr.x + bq.w * wf.w));
                    }
                }
            }
            wg_sync_lds();  // every group's output is in its work area
            if (wave == 0) {
                // the run's first block against `overlap` (dsp.rs:85-106)
                float *o = out + op_cur;
                if (pflag == flag) {  // short -> short
                    ola(o, ovl, ldsf, win_short, half, emit);
                } else {              // long -> short: overlap[..start) at unity gain, then bs0 / 2 overlap-added samples
                    const int start = (bs1 - bs0) / 4;
                    if (emit)
                        for (int k = 4 * lane; k < start; k += 256) st_stream(reinterpret_cast<float4 *>(o + k), *reinterpret_cast<const float4 *>(ovl + k));
                    ola(o + start, ovl + start, ldsf, win_short, bs0 / 2, emit);
                }
            } else if (wave < ng) {
                // a later group's first block against the last block of the group before it (short -> short)
                const float *prev_right = wave_base + (wave - 1) * kWgWaveLds + (size_t)(prev_len - 1) * bs + half;
                ola(out + op_cur + first_len + (size_t)(before - 1) * half, prev_right, ldsf, win_short, half, true);
            }
            wg_sync_lds();  // `overlap` and the neighbours' right halves have been read
            // overlap[..bs / 2) = right half of the run's last block (dsp.rs:125); what lies above stays
            if (wave == ng - 1) {
                const float *right = ldsf + (size_t)(mine - 1) * bs + half;
                for (int k = keep_below + 4 * lane; k < half; k += 256) *reinterpret_cast<float4 *>(ovl + k) = *reinterpret_cast<const float4 *>(right + k);
            }
            wg_sync_lds();  // `overlap` is complete for whoever takes the next block; the work areas are free
            op_cur += first_len + (uint32_t)(run_blocks - 1) * (uint32_t)half;
        }
        os_cur = os_next;
        pflag = flag;
        b = nb_next;
        glen = glen_next;
        flag = flag_next;
    }

    if (b_end == nb) {
        wg_sync_lds();  // `overlap` is complete
        for (int k = 4 * tid; k < bs1 / 2; k += 1024)
            *reinterpret_cast<float4 *>(overlap_out + (size_t)chain * (size_t)(bs1 / 2) + k) = *reinterpret_cast<const float4 *>(ovl + k);
        if (tid == 0) prev_flag_out[chain] = f[nb - 1] ? 1 : 0;  // lib.rs:328
    }
}

}  // namespace

int launch_vorbis_wg(symaccel_ctx *ctx, int bs0_exp, int bs1_exp, const cpx *tw_short, const cpx *tw_long, const float *win_short,
                     const float *win_long, const float *d_spectra, const float *d_residue, size_t spec_stride, const uint8_t *d_block_flag,
                     const int32_t *d_prev_in, int32_t *d_prev_out, const float *d_overlap_in, float *d_overlap_out, float *d_pcm,
                     size_t pcm_stride, size_t n_chains, unsigned nb, unsigned seg, int floor_mode) {
    // (the 4096 / 8192 pair: both cooperative block routines in one instantiation, BIG0 = 2 under MAXE1 = 13)
    if (bs1_exp != 12 && bs1_exp != 13) return SYMACCEL_ERR_INVALID_ARG;
#if SYM_VORBIS_WG != 2
    if (bs1_exp == 12) return SYMACCEL_ERR_INVALID_ARG;  // (those instantiations exist in the SYM_VORBIS_WG = 2 build only)
#endif
    const size_t segs = (nb + seg - 1) / seg;
    const size_t grid = n_chains * segs;
    if (grid > 0x7fffffffu) return SYMACCEL_ERR_INVALID_ARG;
#define SYM_VWG_LAUNCH(FUSED, MAXE1, BIG0)                                                                                                      \
    hipLaunchKernelGGL((vorbis_synth_wg_kernel<FUSED, MAXE1, BIG0>), dim3((unsigned)grid), dim3(64 * kWgWaves), 0, ctx->stream, ctx->dev, bs0_exp, \
                       bs1_exp, tw_short, tw_long, win_short, win_long, d_spectra, d_residue, spec_stride, d_block_flag, d_prev_in, d_prev_out, \
                       d_overlap_in, d_overlap_out, d_pcm, pcm_stride, nb, seg, (unsigned)segs)
    const int big0 = bs0_exp <= 11 ? 0 : (bs0_exp == 12 ? 2 : 4);
#if SYM_VORBIS_WG == 2
#define SYM_VWG_12(FUSED) do { if (big0 == 0) SYM_VWG_LAUNCH(FUSED, 12, 0); else SYM_VWG_LAUNCH(FUSED, 12, 2); } while (0)
#else
#define SYM_VWG_12(FUSED) do { } while (0)
#endif
#define SYM_VWG_BIG(FUSED)                                                                    \
    do {                                                                                      \
        if (bs1_exp == 12) {                                                                  \
            SYM_VWG_12(FUSED);                                                                \
        } else if (big0 == 0) SYM_VWG_LAUNCH(FUSED, 13, 0);                                   \
        else if (big0 == 2) SYM_VWG_LAUNCH(FUSED, 13, 2);                                     \
        else SYM_VWG_LAUNCH(FUSED, 13, 4);                                                    \
    } while (0)
    if (floor_mode == 2) SYM_VWG_BIG(2); else if (d_residue) SYM_VWG_BIG(1); else SYM_VWG_BIG(0);
#undef SYM_VWG_BIG
#undef SYM_VWG_12
#undef SYM_VWG_LAUNCH
    SYM_GPU(ctx, hipGetLastError());
    return SYMACCEL_OK;
}

}  // namespace symaccel
