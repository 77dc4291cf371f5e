// Vorbis synthesis, fast path for the 256 / 2048 block-size pair (bs0_exp = 8, bs1_exp = 11: what
// practically every 44.1 / 48 kHz stream uses, and BASELINE config 4): DspChannel::synth
// (symphonia-codec-vorbis/src/dsp.rs:68-145) with Imdct::new(1024) / Imdct::new(128)
// (vorbis/lib.rs:123-124, symphonia-core/src/dsp/mdct.rs:67-146).
//
// MI355X mapping: the same wavefront-per-chain-segment scheme as aac_synth_kernel (imdct_wave.h): a long
// block is one 512-point FFT in three radix-8 register passes, a RUN of up to eight consecutive short
// blocks is eight 64-point FFTs done at once by the same lanes (8 lanes per block).  The right half of the
// previous block's Imdct output (the `overlap` of dsp.rs:125) stays in 16 VGPRs per lane in the same slot
// layout the post-twiddle produces, so the common long -> long overlap-add is lane-local:
//     out[k] = overlap[k] * win[1023 - k] + pcm[k] * win[k]                       (dsp.rs:85-90, 140-144)
// Block-size transitions stay in registers too: the 128 overlap-added samples of a long <-> short transition live in
// the slots of lanes 48..63, the short overlap in lanes 0..31 (one ds_bpermute hop), the copied samples are stored
// straight from the slot registers; short-block output is read from the half-stored LDS result of the short pass.
// The packed spectrum / PCM offsets of a segment's first block follow from the number of long blocks before it, which
// every wavefront counts itself (16 flags per lane and step); no separate scan kernel runs for this block-size pair.
// HBM traffic per channel-block: 4 * (n/2) B in + 4 * (prev_n + n)/4 B out (+ one halo block per segment).
#define SYM_PACKED_C32_DEFAULT 1  // complex arithmetic as v_pk_*_f32 on register pairs: +7 % here (dsp_device.h)
#include "imdct_wave.h"
#include "vorbis_offsets.h"

namespace symaccel {

namespace {

// the group's spectral lines (and, fused, the residue lines they are multiplied with): 512 B coalesced per load.
// FUSED 0: `sp` is the spectrum.  1: `sp` is the floor curve, `rp` the residue (f32 both).  2: `sp` is the RESIDUE and `rp` the
// floor curve as dB-table indices, one byte per line (symaccel_vorbis_floor1_y_device): a lane's pair of lines is one 16-bit load,
// carried in res[s].x as a bit pattern; apply_residue looks the two table values up in LDS (`dbt`, 256 floats).
template <int FUSED>
__device__ __forceinline__ void fetch_lines(const float *sp, const float *rp, uint32_t off, int n_loads, int lane,
                                            float2 (&line)[8], float2 (&res)[8]) {
    const float2 *src = reinterpret_cast<const float2 *>(sp + off);
#pragma unroll
    for (int s = 0; s < 8; ++s)
        if (s < n_loads) line[s] = ld_stream(src + lane + 64 * s);
    if constexpr (FUSED == 1) {
        const float2 *rs = reinterpret_cast<const float2 *>(rp + off);
#pragma unroll
        for (int s = 0; s < 8; ++s)
            if (s < n_loads) res[s] = ld_stream(rs + lane + 64 * s);
    }
    if constexpr (FUSED == 2) {
        const uint16_t *ys = reinterpret_cast<const uint16_t *>(reinterpret_cast<const uint8_t *>(rp) + off);
#pragma unroll
        for (int s = 0; s < 8; ++s)
            if (s < n_loads) res[s].x = __uint_as_float((unsigned)ys[lane + 64 * s]);
    }
}
template <int FUSED>
__device__ __forceinline__ void apply_residue(float2 (&line)[8], const float2 (&res)[8], const float *dbt) {
    if constexpr (FUSED == 1) {
#pragma unroll
        for (int s = 0; s < 8; ++s) {
            line[s].x *= res[s].x;  // lib.rs:289-291: *f *= r
            line[s].y *= res[s].y;
        }
    }
    if constexpr (FUSED == 2) {
#pragma unroll
        for (int s = 0; s < 8; ++s) {
            const unsigned yy = __float_as_uint(res[s].x);
#if SYM_LDS_ABLATE & 256
            line[s].x = dbt[s] * line[s].x + __uint_as_float(yy & 1u);
            line[s].y = dbt[s + 8] * line[s].y;
#else
            line[s].x = dbt[yy & 255u] * line[s].x;  // floor.rs:822 (the curve's value) and lib.rs:289-291 (*f *= r) in one step
            line[s].y = dbt[yy >> 8] * line[s].y;
#endif
        }
    }
}

// Build variant (tuning knob SYM_VORBIS_WAVES, see build.py): 2 = four wavefronts per workgroup, lane twiddles in 31 VGPRs,
// two wavefronts per SIMD (the default); 3 = six wavefronts per workgroup, lane twiddles read from a 4 KiB LDS copy,
// register budget of three wavefronts per SIMD.
#ifndef SYM_VORBIS_WAVES
#define SYM_VORBIS_WAVES 2
#endif
constexpr int kWaves = SYM_VORBIS_WAVES == 3 ? 6 : 4;
constexpr int kTabTw = 0;        // shared LDS tables: Imdct(1024) twiddles, 512 complex
constexpr int kTabWin = 1024;    //   long window (left half of the 2048-sample window), 1024 f32
constexpr int kTabWs = 2048;     //   short window (left half of the 256-sample window), 128 f32
constexpr int kTabTws = 2048 + 128;  //   Imdct(128) twiddles, 64 complex
constexpr int kTabFloats = 2048 + 128 + 128;
constexpr int kBs0 = 256, kBs1 = 2048;

// out[q] = ov[q] * ws[127 - (k + q)] + y[q] * ws[k + q], q = 0..3 (vorbis dsp.rs:140-144 with the short window)
__device__ __forceinline__ void ola_short4(const float *ws, int k, const float (&ov)[4], const float (&y)[4], float4 &o) {
    const float4 wf = *reinterpret_cast<const float4 *>(ws + k);
    const float4 wr = *reinterpret_cast<const float4 *>(ws + 124 - k);  // ws[124-k .. 127-k], used back to front
    o.x = ov[0] * wr.w + y[0] * wf.x;
    o.y = ov[1] * wr.z + y[1] * wf.y;
    o.z = ov[2] * wr.y + y[2] * wf.z;
    o.w = ov[3] * wr.x + y[3] * wf.w;
}

// FUSED: the spectrum is floor[i] * residue[i] (the dot product of lib.rs:282-292), multiplied as the lines are consumed --
// one rounded multiply per line, exactly the reference's `*f *= r`, without a separate pass over HBM.
template <int FUSED>
__global__ __launch_bounds__(64 * kWaves, SYM_VORBIS_WAVES) void vorbis_synth_wave_kernel(
    DevTables tb, const cpx *__restrict__ tw_short, const cpx *__restrict__ tw_long,
    const float *__restrict__ win_short, const float *__restrict__ win_long, const float *__restrict__ spectra,
    const float *__restrict__ residue, size_t spec_stride, const uint8_t *__restrict__ flags, const int32_t *__restrict__ prev_flag_in,
    int32_t *__restrict__ prev_flag_out, const float *__restrict__ overlap_in, float *__restrict__ overlap_out,
    float *__restrict__ pcm, size_t pcm_stride, unsigned nb, unsigned seg_len,
    unsigned segs_per_chain, unsigned n_items) {
    __shared__ __attribute__((aligned(16))) float tabs[kTabFloats + (FUSED == 2 ? 256 : 0)];  // (+ FLOOR1_INVERSE_DB_TABLE)
    __shared__ __attribute__((aligned(16))) float wave_lds[kWaves][kWaveLds];
    __shared__ unsigned wave_chain[kWaves];  // the wavefront's chain index, parked for the epilogue (see there)
#if SYM_VORBIS_WAVES == 3 || SYM_C32_IS_PACKED
    __shared__ __attribute__((aligned(16))) c32 lane_tab[kLaneTabComplex];
    fill_lane_tables_lds(tb, lane_tab, (int)threadIdx.x, 64 * kWaves);
#endif

    for (int i = (int)threadIdx.x; i < 1024; i += 64 * kWaves) {
        tabs[kTabTw + i] = reinterpret_cast<const float *>(tw_long)[i];
        tabs[kTabWin + i] = win_long[i];
        if (i < 128) {
            tabs[kTabWs + i] = win_short[i];
            tabs[kTabTws + i] = reinterpret_cast<const float *>(tw_short)[i];
        }
    }
    if constexpr (FUSED == 2) tabs[kTabFloats + ((int)threadIdx.x & 255)] = tb.vorbis_floor1_db[(int)threadIdx.x & 255];
    const float *dbt = tabs + kTabFloats;
    __syncthreads();  // the only workgroup-wide barrier

    const int lane = (int)threadIdx.x & 63, wave = (int)threadIdx.x >> 6;
    const unsigned item = blockIdx.x * kWaves + (unsigned)wave;
    if (item >= n_items) return;
    float *ldsf = wave_lds[wave];
    c32 *lds = reinterpret_cast<c32 *>(ldsf);
    const c32 *tw = reinterpret_cast<const c32 *>(tabs + kTabTw);
    const float *wl = tabs + kTabWin, *ws = tabs + kTabWs;
    const cpx *tws = reinterpret_cast<const cpx *>(tabs + kTabTws);

    const unsigned chain = item / segs_per_chain, seg = item % segs_per_chain;
    if (lane == 0) wave_chain[wave] = chain;
    const unsigned b_begin = seg * seg_len, b_end = min(b_begin + seg_len, nb);
    const uint8_t *f = flags + (size_t)chain * nb;
    const float *sp = spectra + (size_t)chain * spec_stride;
    // (FUSED 2: `residue` is the byte plane of table indices, `spectra` the residue)
    const float *rp = FUSED == 2 ? reinterpret_cast<const float *>(reinterpret_cast<const uint8_t *>(residue) + (size_t)chain * spec_stride)
                                 : (FUSED ? residue + (size_t)chain * spec_stride : nullptr);
    float *out = pcm + (size_t)chain * pcm_stride;
    const int pf0 = prev_flag_in[chain];

#if SYM_VORBIS_WAVES == 3
    const LaneTablesLds lt = lane_tables_lds(tb, lane_tab, lane);
#elif SYM_C32_IS_PACKED
    LaneTablesMixed lt;  // (complex values as aligned register pairs need the 14 VGPRs the last three stages' twiddles took)
    load_lane_tables_mixed(tb, lane_tab, lane, lt);
#else
    LaneTables lt;
    load_lane_tables(tb, lane, lt);
#endif

    // overlap (dsp.rs:125), slot layout: dl[h][0..3] = overlap[4 m2 + q], dl[h][4..7] = overlap[1020 - 4 m2 + q]
    float dl[2][8];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        if (b_begin == 0) {
            load_slot(overlap_in + (size_t)chain * 1024, lane + 64 * h, dl[h]);
        } else {
#pragma unroll
            for (int q = 0; q < 8; ++q) dl[h][q] = 0.0f;
        }
    }
    bool hi_fresh = b_begin == 0;  // overlap[128..1024) is what the reference would hold at this point

    // Block flags as wave-uniform bit masks (bit = 1: long; blocks past b_end read as long): m0 covers blocks
    // [wbase, wbase + 64), m1 the next 64.  One coalesced byte load + ballot per 64 blocks replaces a dependent
    // global load per block in the group bookkeeping; the packed offsets are carried as running sums.
    const long b_first = b_begin == 0 ? 0 : (long)b_begin - 1;  // the halo block rebuilds the overlap only
    auto load_mask = [&](long base) -> unsigned long long {
        const long idx = base + lane;
        const bool is_long = idx >= (long)b_end || f[idx] != 0;
        return __ballot(is_long);
    };
    long wbase = b_first;
    unsigned long long m0 = load_mask(wbase), m1 = load_mask(wbase + 64);
    // A group = one long block, or a run of up to 8 consecutive short blocks (never crossing b_end).
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
        const unsigned long long w = off == 0 ? m0 : ((m0 >> off) | (m1 << (64 - off)));
        flag_out = (int)(w & 1ull);
        if (flag_out) return 1;
        const int run = w ? __builtin_ctzll(w) : 64;
        return run < 8 ? run : 8;
    };

    long b = b_first;
    int flag = 1;
    int glen = group_at(b, flag);
    // flag of the block before b (lib.rs:298: the first block of a stream pairs with itself)
    int pflag = b == 0 ? (pf0 < 0 ? flag : (pf0 ? 1 : 0)) : (f[b - 1] ? 1 : 0);
    // Packed offsets of block b: with S(b) = sum of the first b block sizes = 256 b + 1792 L(b), L(b) = long blocks before b,
    //   spectrum offset = S(b) / 2,   PCM offset = (S(b) + n_{-1} + S(b-1)) / 4  (lib.rs:303: block k yields (n_{k-1} + n_k) / 4),
    // where n_{-1} is the size the first block is paired with (its own when there is no previous block, lib.rs:298).
    auto sizes_before = [&](long bb) -> uint32_t { return 256u * (uint32_t)bb + 1792u * count_long_before(f, bb, lane); };
    uint32_t os_cur = 0, op_cur = 0;
    if (b_first > 0) {
        const uint32_t s_b = sizes_before(b_first);
        const uint32_t n_prev = f[b_first - 1] ? 2048u : 256u;                  // size of block b_first - 1
        const uint32_t n_m1 = pf0 < 0 ? (f[0] ? 2048u : 256u) : (pf0 ? 2048u : 256u);
        os_cur = s_b / 2;
        op_cur = (s_b + n_m1 + (s_b - n_prev)) / 4;
    }
    float2 line[8], res[8];  // the group's lines: 1024 (long) or 128 per short block
#pragma unroll
    for (int s = 0; s < 8; ++s) line[s] = res[s] = make_float2(0.0f, 0.0f);
    if (glen > 0) fetch_lines<FUSED>(sp, rp, os_cur, flag ? 8 : glen, lane, line, res);

    while (b < (long)b_end) {
        const long nb_next = b + glen;
        int flag_next = 1;
        const int glen_next = group_at(nb_next, flag_next);
        const uint32_t os_next = os_cur + (flag ? 1024u : 128u * (uint32_t)glen);

        if (flag) {
            // ------------------------------------------------------------------ one long block
            const bool emit = b >= (long)b_begin;
            c32 z[8];
            const int mirror = (63 - lane) * 4;
            apply_residue<FUSED>(line, res, dbt);
#pragma unroll
            for (int s = 0; s < 8; ++s) {
#if SYM_LDS_ABLATE & 1
                const float mirrored = line[7 - s].y + (float)mirror;
#else
                const float mirrored = __int_as_float(__builtin_amdgcn_ds_bpermute(mirror, __float_as_int(line[7 - s].y)));
#endif
                z[s] = pre_twiddle(line[s].x, mirrored, tw[((SYM_LDS_ABLATE & 2) ? 0 : lane) + 64 * s]);
            }
            if (glen_next > 0) fetch_lines<FUSED>(sp, rp, os_next, flag_next ? 8 : glen_next, lane, line, res);  // prefetch
            fft512_wave(z, lane, lds, lt);
            // (the two slot halves h = 0, 1 are post-twiddled, overlap-added, stored and turned into the new overlap one
            // after the other: 16 instead of 32 live outputs)
            float *o = out + op_cur;
            if (pflag) {
                // long -> long (dsp.rs:85-90): out[k] = overlap[k] * win[1023 - k] + pcm[k] * win[k]
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    float x[8], x2[8], w[8], dst[8];
                    post_slot(lds, tw, lane + 64 * h, x, x2);
                    load_slot(wl, ((SYM_LDS_ABLATE & 128) ? 0 : lane) + 64 * h, w);
#pragma unroll
                    for (int q = 0; q < 8; ++q) dst[q] = dl[h][q] * w[7 - q] + x[q] * w[q];
                    if (emit) store_slot_stream(o, lane + 64 * h, dst);
#pragma unroll
                    for (int q = 0; q < 8; ++q) dl[h][q] = x2[q];  // overlap = imdct[1024..2048) (dsp.rs:125)
                }
                wave_sync();  // Z in LDS is overwritten by the next group
            } else {
                // short -> long (dsp.rs:107-122): out[k] = overlap[k] * ws[127-k] + imdct[448+k] * ws[k] for k < 128,
                // then imdct[576..1024) copied.  imdct[448..576) sits in the slots of lanes 48..63 (m2 = 112 + t);
                // the short overlap[0..128) sits in dl[0][0..3] of lanes 0..31: lane 48+t takes lane t's (for its
                // first float4) and lane 31-t's (for its second) through ds_bpermute.
                const int t = lane >= 48 ? lane - 48 : 0;
                float ovA[4], ovB[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    ovA[q] = __int_as_float(__builtin_amdgcn_ds_bpermute(4 * t, __float_as_int(dl[0][q])));
                    ovB[q] = __int_as_float(__builtin_amdgcn_ds_bpermute(4 * (31 - t), __float_as_int(dl[0][q])));
                }
                float4 *o4 = reinterpret_cast<float4 *>(o);
                {
                    float x[8], x2[8];
                    post_slot(lds, tw, lane, x, x2);
                    // copied part: second float4 of every slot with 1020 - 4 m2 >= 576, i.e. m2 <= 111
                    if (emit) st_stream(o4 + (143 - lane), make_float4(x[4], x[5], x[6], x[7]));           // (572 - 4 lane) / 4
#pragma unroll
                    for (int q = 0; q < 8; ++q) dl[0][q] = x2[q];
                }
                {
                    float x[8], x2[8];
                    post_slot(lds, tw, lane + 64, x, x2);
                    if (emit) {
                        if (lane < 48) st_stream(o4 + (79 - lane), make_float4(x[4], x[5], x[6], x[7]));  // (316 - 4 lane) / 4
                        if (lane >= 48) {
                            const float ya[4] = {x[0], x[1], x[2], x[3]}, yb[4] = {x[4], x[5], x[6], x[7]};
                            float4 ra, rb;
                            ola_short4(ws, 4 * t, ovA, ya, ra);
                            ola_short4(ws, 124 - 4 * t, ovB, yb, rb);
                            st_stream(o4 + (t), ra);
                            st_stream(o4 + (31 - t), rb);
                        }
                    }
#pragma unroll
                    for (int q = 0; q < 8; ++q) dl[1][q] = x2[q];
                }
                wave_sync();  // Z in LDS is overwritten by the next group
            }
            hi_fresh = true;
        } else {
            // ------------------------------------------------------------------ a run of `glen` short blocks
            apply_residue<FUSED>(line, res, dbt);
#pragma unroll
            for (int s = 0; s < 8; ++s) {  // block s of the run -> window s of the eight-way short transform
                if (s < glen) {
                    ldsf[short_row(s) + 2 * lane] = line[s].x;
                    ldsf[short_row(s) + 2 * lane + 1] = line[s].y;
                }
            }
            wave_sync();
            if (glen_next > 0) fetch_lines<FUSED>(sp, rp, os_next, flag_next ? 8 : glen_next, lane, line, res);
            imdct_short_wave(lane, ldsf, tws, lt);  // H[w] = ldsf[short_row(w) ..]; ends with a wave_sync
            // PCM of the run's blocks is packed back to back: the first contributes 576 (after a long) or 128
            const uint32_t first_len = pflag ? 576u : 128u;
            {   // ---- the run's first block
                const bool emit = b >= (long)b_begin;
                float4 *o4 = reinterpret_cast<float4 *>(out + op_cur);
                if (pflag) {
                    // long -> short (dsp.rs:91-106): overlap[0..448) at unity gain, then 128 overlap-added samples.
                    // overlap[448..576) sits in the slots of lanes 48..63 (m2 = 112 + t).
                    if (emit) {
                        st_stream(o4 + (lane), make_float4(dl[0][0], dl[0][1], dl[0][2], dl[0][3]));                     // 4 m2, m2 < 64
                        if (lane < 48) st_stream(o4 + (64 + lane), make_float4(dl[1][0], dl[1][1], dl[1][2], dl[1][3]));  // m2 < 112
                        if (lane >= 48) {
                            const int t = lane - 48;
                            const float oa[4] = {dl[1][0], dl[1][1], dl[1][2], dl[1][3]}, ob[4] = {dl[1][4], dl[1][5], dl[1][6], dl[1][7]};
                            float ya[4], yb[4];
                            ys4(ldsf, 0, 4 * t, ya);
                            ys4(ldsf, 0, 124 - 4 * t, yb);
                            float4 ra, rb;
                            ola_short4(ws, 4 * t, oa, ya, ra);
                            ola_short4(ws, 124 - 4 * t, ob, yb, rb);
                            st_stream(o4 + (112 + t), ra);        // (448 + 4 t) / 4
                            st_stream(o4 + (143 - t), rb);        // (448 + 124 - 4 t) / 4
                        }
                    }
                } else if (emit && lane < 32) {
                    // short -> short (dsp.rs:85-90) against the carried overlap[0..128) = first float4 of lanes 0..31
                    const float ov[4] = {dl[0][0], dl[0][1], dl[0][2], dl[0][3]};
                    float y[4];
                    ys4(ldsf, 0, 4 * lane, y);
                    float4 r;
                    ola_short4(ws, 4 * lane, ov, y, r);
                    st_stream(o4 + (lane), r);
                }
            }
            // ---- the rest, two blocks per round (one per half-wavefront): short -> short (dsp.rs:85-90), lane l of a
            // half produces out[4 l .. 4 l + 3] of its block from the two neighbouring transforms in LDS
            {
                const int l32 = lane & 31;
                for (int i0 = 1; i0 < glen; i0 += 2) {
                    const int i = i0 + (lane >> 5);
                    if (i < glen && b + i >= (long)b_begin) {
                        float ov[4], y[4];
                        ys4(ldsf, i - 1, kBs0 / 2 + 4 * l32, ov);
                        ys4(ldsf, i, 4 * l32, y);
                        float4 r;
                        ola_short4(ws, 4 * l32, ov, y, r);
                        st_stream(reinterpret_cast<float4 *>(out + op_cur + first_len + 128u * (uint32_t)(i - 1)) + l32, r);
                    }
                }
            }
            // overlap[0..128) = imdct[128..256) of the run's last block (dsp.rs:125); the rest is left as it was
            if (lane < 32) {
                float v[4];
                ys4(ldsf, glen - 1, kBs0 / 2 + 4 * lane, v);
#pragma unroll
                for (int q = 0; q < 4; ++q) dl[0][q] = v[q];
            }
            wave_sync();  // H is overwritten by the next group
        }
        op_cur += flag ? (pflag ? 1024u : 576u) : ((pflag ? 576u : 128u) + 128u * (uint32_t)(glen - 1));
        os_cur = os_next;
        pflag = flag;  // every block of a group has the group's flag
        b = nb_next;
        glen = glen_next;
        flag = flag_next;
    }

    if (b_end == nb) {
        // The chain index and the addresses derived from it are re-derived here from LDS through an opaque copy of the
        // thread index: kept live across the main loop they were spilled to scratch (the loop runs at the full 256-VGPR
        // budget of two wavefronts per SIMD).
        wave_sync();
        unsigned tid2 = threadIdx.x;
        asm volatile("" : "+v"(tid2));
        const unsigned chain2 = wave_chain[tid2 >> 6];
        const uint8_t *f2 = flags + (size_t)chain2 * nb;
        const float *sp2 = spectra + (size_t)chain2 * spec_stride;
        const float *rp2 = FUSED == 2 ? reinterpret_cast<const float *>(reinterpret_cast<const uint8_t *>(residue) + (size_t)chain2 * spec_stride)
                                      : (FUSED ? residue + (size_t)chain2 * spec_stride : nullptr);
        if (!hi_fresh) {
            // The chain ends in short blocks and this segment never saw a long one: overlap[128..1024) still
            // holds what the most recent long block left there (never used for PCM, but part of the state the
            // reference carries).  Rebuild it from that block's spectrum, or keep the incoming state.
            const long bl = b_begin > 0 ? last_long_before(f2, (long)b_begin, lane) : -1;
            float keep[2][8];
#pragma unroll
            for (int h = 0; h < 2; ++h)
#pragma unroll
                for (int q = 0; q < 8; ++q) keep[h][q] = dl[h][q];
            if (bl >= 0) {
                c32 z[8];
                fetch_lines<FUSED>(sp2, rp2, (256u * (uint32_t)bl + 1792u * count_long_before(f2, bl, lane)) / 2, 8, lane, line, res);
                apply_residue<FUSED>(line, res, dbt);
                const int mirror = (63 - lane) * 4;
#pragma unroll
                for (int s = 0; s < 8; ++s) {
                    const float mirrored = __int_as_float(__builtin_amdgcn_ds_bpermute(mirror, __float_as_int(line[7 - s].y)));
                    z[s] = pre_twiddle(line[s].x, mirrored, tw[lane + 64 * s]);
                }
                fft512_wave(z, lane, lds, lt);
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    float x[8];
                    post_slot(lds, tw, lane + 64 * h, x, dl[h]);
                }
                wave_sync();
            } else {
#pragma unroll
                for (int h = 0; h < 2; ++h) load_slot(overlap_in + (size_t)chain2 * 1024, lane + 64 * h, dl[h]);
            }
            // overlap[0..128) comes from the short blocks of this segment
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int m2 = lane + 64 * h;
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    const int j = q < 4 ? 4 * m2 + q : 1020 - 4 * m2 + (q - 4);
                    if (j < kBs0 / 2) dl[h][q] = keep[h][q];
                }
            }
        }
        float *d = overlap_out + (size_t)chain2 * 1024;
#pragma unroll
        for (int h = 0; h < 2; ++h) store_slot(d, lane + 64 * h, dl[h]);
        if (lane == 0) prev_flag_out[chain2] = f2[nb - 1] ? 1 : 0;  // lib.rs:328
    }
}

}  // namespace

int launch_vorbis_wave(symaccel_ctx *ctx, const cpx *tw_short, const cpx *tw_long, const float *win_short,
                       const float *win_long, const float *d_spectra, const float *d_residue, size_t spec_stride,
                       const uint8_t *d_block_flag,
                       const int32_t *d_prev_in, int32_t *d_prev_out, const float *d_overlap_in, float *d_overlap_out,
                       float *d_pcm, size_t pcm_stride, size_t n_chains, unsigned nb,
                       unsigned seg, int floor_mode) {
    const size_t segs = (nb + seg - 1) / seg;
    const size_t items = n_chains * segs;
    const size_t grid = (items + kWaves - 1) / kWaves;
    if (items > 0xffffffffu || grid > 0x7fffffffu) return SYMACCEL_ERR_INVALID_ARG;
#define SYM_VW_LAUNCH(MODE)                                                                                                           \
    hipLaunchKernelGGL(vorbis_synth_wave_kernel<MODE>, dim3((unsigned)grid), dim3(64 * kWaves), 0, ctx->stream, ctx->dev, tw_short, tw_long, \
                       win_short, win_long, d_spectra, d_residue, spec_stride, d_block_flag, d_prev_in, d_prev_out, d_overlap_in,       \
                       d_overlap_out, d_pcm, pcm_stride, nb, seg, (unsigned)segs, (unsigned)items)
    if (floor_mode == 2) SYM_VW_LAUNCH(2);
    else if (d_residue) SYM_VW_LAUNCH(1);
    else SYM_VW_LAUNCH(0);
#undef SYM_VW_LAUNCH
    SYM_GPU(ctx, hipGetLastError());
    return SYMACCEL_OK;
}

}  // namespace symaccel
