// Vorbis synthesis on the register-pass FFT for EVERY block-size pair with long blocks of up to 2048 samples
// (bs0_exp 6 .. 11, bs1_exp <= 11; the 256 / 2048 pair keeps its own kernel, vorbis_wave.hip): DspChannel::synth
// (symphonia-codec-vorbis/src/dsp.rs:68-145) with Imdct::new(bs / 2) per block size (vorbis/lib.rs:123-124,
// symphonia-core/src/dsp/mdct.rs:67-146).
//
// MI355X mapping: the wavefront-per-chain-segment scheme of vorbis_wave.hip, with fft_wave_multi (imdct_wave.h) as the
// transform: a pass over the 512-point work array is 512 / P independent P-point FFTs (P = bs / 4), so a GROUP is a run of up
// to 2048 / bs consecutive blocks of one size -- one long block of 2048, two of 1024, ... thirty-two short ones of 64 -- that
// is fetched (1 KiB per load instruction), transformed and post-twiddled together.  The group's Imdct output (bs samples per
// block, natural order) lands in the wavefront's LDS work area; `overlap` (dsp.rs:125) is a per-wavefront LDS array with
// exactly the reference's contents, so the three window cases of dsp.rs:85-122 are loops over LDS with 16-byte accesses and
// the stale upper part a short block leaves behind needs no special case.  Blocks after the first of a group lap with their
// predecessor inside the work area.  Packed offsets: from the scan kernel at the segment start, running sums after that.
// HBM traffic per channel-block: 4 * (n / 2) B in + 4 * (prev_n + n) / 4 B out (+ one halo block per segment).
#include "imdct_wave.h"

namespace symaccel {

namespace {

constexpr int kWaves = 4;

// out[k .. k+3] = ov[k ..] * win[len - 1 - k ..] + y[k ..] * win[k ..], k = 4 c, c = lane, lane + 64, ...  (dsp.rs:140-144)
__device__ __forceinline__ void ola_span(float *__restrict__ o, const float *ov, const float *y, const float *win, int len, int lane, bool emit) {
    if (!emit) return;
    for (int k = 4 * lane; k < len; k += 256) {
        const float4 a = *reinterpret_cast<const float4 *>(ov + k), b = *reinterpret_cast<const float4 *>(y + k);
        const float4 wf = *reinterpret_cast<const float4 *>(win + k);
        const float4 wr = *reinterpret_cast<const float4 *>(win + len - 4 - k);  // win[len-4-k .. len-1-k], used back to front
        st_stream(reinterpret_cast<float4 *>(o + k),
                  make_float4(a.x * wr.w + b.x * wf.x, a.y * wr.z + b.y * wf.y, a.z * wr.y + b.z * wf.z, a.w * wr.x + b.w * wf.w));
    }
}
__device__ __forceinline__ void copy_span(float *__restrict__ o, const float *src, int len, int lane, bool emit) {
    if (!emit) return;
    for (int k = 4 * lane; k < len; k += 256) st_stream(reinterpret_cast<float4 *>(o + k), *reinterpret_cast<const float4 *>(src + k));
}

// MAXE1: the largest long-block exponent the instantiation serves.  Up to 1024-sample long blocks the tables and the overlap
// arrays are half the size and three workgroups fit a CU (52 KiB of LDS each, <= 168 VGPRs): the kernel is bound by the latency
// of a group's dependent steps, and a third wavefront per SIMD is worth more than anything else here.  (The fused variant carries
// sixteen more registers -- the prefetched residue lines -- and would spill at 168: it stays at two wavefronts per SIMD.)
template <bool FUSED, int MAXE1>
__global__ __launch_bounds__(64 * kWaves, (MAXE1 <= 10 && !FUSED) ? 3 : 2) void vorbis_synth_wave2_kernel(
    DevTables tb, int e0, int e1, const cpx *__restrict__ tw_short, const cpx *__restrict__ tw_long,
    const float *__restrict__ win_short, const float *__restrict__ win_long, const float *__restrict__ spectra,
    const float *__restrict__ residue, size_t spec_stride, const uint8_t *__restrict__ flags, const int32_t *__restrict__ prev_flag_in,
    int32_t *__restrict__ prev_flag_out, const float *__restrict__ overlap_in, float *__restrict__ overlap_out,
    float *__restrict__ pcm, size_t pcm_stride, const uint32_t *__restrict__ offs, unsigned nb, unsigned seg_len,
    unsigned segs_per_chain, unsigned n_items) {
    // shared tables: Imdct twiddles and left window halves of both block sizes (bs / 2 floats each)
    __shared__ __attribute__((aligned(16))) float tabs[2 << MAXE1];
    __shared__ __attribute__((aligned(16))) float wave_lds[kWaves][kWaveLds];
    __shared__ __attribute__((aligned(16))) float wave_ovl[kWaves][(1 << MAXE1) / 2];
    const int bs0 = 1 << e0, bs1 = 1 << e1;
    float *t_twl = tabs, *t_wl = tabs + bs1 / 2, *t_tws = tabs + bs1, *t_ws = tabs + bs1 + bs0 / 2;
    for (int i = (int)threadIdx.x; i < bs1 / 2; i += 64 * kWaves) {
        t_twl[i] = reinterpret_cast<const float *>(tw_long)[i];
        t_wl[i] = win_long[i];
        if (i < bs0 / 2) {
            t_tws[i] = reinterpret_cast<const float *>(tw_short)[i];
            t_ws[i] = win_short[i];
        }
    }
    __syncthreads();  // the only workgroup-wide barrier

    const int lane = (int)threadIdx.x & 63, wave = (int)threadIdx.x >> 6;
    const unsigned item = blockIdx.x * kWaves + (unsigned)wave;
    if (item >= n_items) return;
    float *ldsf = wave_lds[wave];
    c32 *lds = reinterpret_cast<c32 *>(ldsf);
    float *ovl = wave_ovl[wave];
    const unsigned chain = item / segs_per_chain, seg = item % segs_per_chain;
    const unsigned b_begin = seg * seg_len, b_end = min(b_begin + seg_len, nb);
    const uint8_t *f = flags + (size_t)chain * nb;
    const uint32_t *os = offs + (size_t)chain * 2 * (nb + 1), *op = os + (nb + 1);
    const float *sp = spectra + (size_t)chain * spec_stride;
    const float *rp = FUSED ? residue + (size_t)chain * spec_stride : nullptr;
    float *out = pcm + (size_t)chain * pcm_stride;
    const int pf0 = prev_flag_in[chain];
    LaneTables lt;
    load_lane_tables(tb, lane, lt);

    // overlap (dsp.rs:125): the caller's state at a chain's start; zero in front of a later segment, whose halo block rebuilds
    // the part the next block reads
    for (int k = 4 * lane; k < bs1 / 2; k += 256)
        *reinterpret_cast<float4 *>(ovl + k) = b_begin == 0 ? *reinterpret_cast<const float4 *>(overlap_in + (size_t)chain * (size_t)(bs1 / 2) + k)
                                                             : make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    bool hi_fresh = b_begin == 0 || e0 == e1;  // overlap[bs0/2 .. bs1/2) is what the reference would hold at this point

    // Block flags as wave-uniform bit masks (bit = 1: long; blocks past b_end read as the opposite of the run they would extend):
    // one coalesced byte load + ballot per 64 blocks instead of a dependent global load per block.
    const long b_first = b_begin == 0 ? 0 : (long)b_begin - 1;  // the halo block rebuilds the overlap only
    auto load_mask = [&](long base) -> unsigned long long {
        const long idx = base + lane;
        return __ballot(idx < (long)b_end && f[idx] != 0);
    };
    long wbase = b_first;
    unsigned long long m0 = load_mask(wbase), m1 = load_mask(wbase + 64);
    const int cap0 = 2048 >> e0, cap1 = 2048 >> e1;  // blocks per group
    // A group = a run of consecutive blocks with one flag, at most 2048 / bs of them, never crossing b_end.
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
        int run = w ? __builtin_ctzll(w) : 64;  // blocks with the same flag from bb on (as far as the two masks reach)
        const long left = (long)b_end - bb;
        if ((long)run > left) run = (int)left;
        const int cap = flag_out ? cap1 : cap0;
        return run < cap ? run : cap;
    };

    long b = b_first;
    int flag = 1;
    int glen = group_at(b, flag);
    // flag of the block before b (lib.rs:298: the first block of a stream pairs with itself)
    int pflag = b == 0 ? (pf0 < 0 ? flag : (pf0 ? 1 : 0)) : (f[b - 1] ? 1 : 0);
    uint32_t os_cur = glen > 0 ? os[b] : 0u, op_cur = glen > 0 ? op[b] : 0u;
    float4 v[4], r[4];
    auto fetch = [&](uint32_t off, int fl, int n_blocks) {
        const size_t valid = (size_t)n_blocks << ((fl ? e1 : e0) - 1);
        multi_fetch(sp + off, valid, lane, v);
        if constexpr (FUSED) multi_fetch(rp + off, valid, lane, r);
    };
    if (glen > 0) fetch(os_cur, flag, glen);

    while (b < (long)b_end) {
        const int e = flag ? e1 : e0, bs = 1 << e, logp = e - 2, P = 1 << logp;
        const long nb_next = b + glen;
        int flag_next = 1;
        const int glen_next = group_at(nb_next, flag_next);
        const uint32_t os_next = os_cur + ((uint32_t)glen << (e - 1));
        const c32 *tw = reinterpret_cast<const c32 *>(flag ? t_twl : t_tws);
        hi_fresh = hi_fresh || flag;

        // ---- the group's lines -> LDS (natural order), multiplied by the residue on the way (lib.rs:289-291: *f *= r)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            float4 x = v[q];
            if constexpr (FUSED) {
                x.x *= r[q].x;
                x.y *= r[q].y;
                x.z *= r[q].z;
                x.w *= r[q].w;
            }
            reinterpret_cast<float4 *>(ldsf)[lane + 64 * q] = x;
        }
        wave_sync();
        if (glen_next > 0) fetch(os_next, flag_next, glen_next);  // the next group travels while this one is transformed
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

        // ---- the group's first block against `overlap` (dsp.rs:85-122)
        {
            const bool emit = b >= (long)b_begin;
            float *o = out + op_cur;
            const float *left = ldsf;
            if (pflag == flag) {
                ola_span(o, ovl, left, flag ? t_wl : t_ws, bs / 2, lane, emit);
            } else if (pflag) {  // long -> short: overlap[..start) at unity gain, then bs0 / 2 overlap-added samples
                const int start = (bs1 - bs0) / 4;
                copy_span(o, ovl, start, lane, emit);
                ola_span(o + start, ovl + start, left, t_ws, bs0 / 2, lane, emit);
            } else {             // short -> long: bs0 / 2 overlap-added samples, then imdct[end .. bs1 / 2) at unity gain
                const int start = (bs1 - bs0) / 4, len = bs0 / 2, end = start + len;
                ola_span(o, ovl, left + start, t_ws, len, lane, emit);
                copy_span(o + len, left + end, bs1 / 2 - end, lane, emit);
            }
        }
        // ---- the rest of the run laps with its predecessor inside the work area (equal sizes: dsp.rs:85-90)
        const uint32_t first_len = (uint32_t)((pflag ? bs1 : bs0) + bs) >> 2;
        if (glen > 1) {
            const int half = bs >> 1, per_block = half >> 2;  // float4 chunks per block
            const float *win = flag ? t_wl : t_ws;
            // (all of them are emitted: only a halo block precedes b_begin, and a halo block is its group's first)
            for (int c = lane; c < (glen - 1) * per_block; c += 64) {
                const int i = 1 + c / per_block, k = 4 * (c % per_block);
                const float *ov = ldsf + (size_t)(i - 1) * bs + half, *y = ldsf + (size_t)i * bs;
                const float4 a = *reinterpret_cast<const float4 *>(ov + k), bq = *reinterpret_cast<const float4 *>(y + k);
                const float4 wf = *reinterpret_cast<const float4 *>(win + k);
                const float4 wr = *reinterpret_cast<const float4 *>(win + half - 4 - k);
                st_stream(reinterpret_cast<float4 *>(out + op_cur + first_len + (size_t)(i - 1) * half + k),
                          make_float4(a.x * wr.w + bq.x * wf.x, a.y * wr.z + bq.y * wf.y, a.z * wr.y + bq.z * wf.z, a.w * wr.x + bq.w * wf.w));
            }
        }
        wave_sync();  // `overlap` has been read
        // overlap[..bs / 2) = right half of the run's last block (dsp.rs:125); what lies above stays
        {
            const float *right = ldsf + (size_t)(glen - 1) * bs + (bs >> 1);
            for (int k = 4 * lane; k < bs / 2; k += 256) *reinterpret_cast<float4 *>(ovl + k) = *reinterpret_cast<const float4 *>(right + k);
        }
        wave_sync();  // the work area is overwritten by the next group
        op_cur += first_len + (uint32_t)(glen - 1) * (uint32_t)(bs >> 1);
        os_cur = os_next;
        pflag = flag;
        b = nb_next;
        glen = glen_next;
        flag = flag_next;
    }

    if (b_end == nb) {
        if (!hi_fresh) {
            // The chain ends in short blocks and this segment never saw a long one: overlap[bs0/2 .. bs1/2) still holds what the
            // most recent long block left there (dsp.rs:125 only rewrites the first bs / 2 entries; never used for PCM, but part
            // of the state the reference carries).  Rebuild it from that block, or keep the incoming state if the batch has no
            // long block in front of this segment.
            long bl = -1;
            for (long base = ((long)b_begin - 1) & ~63l; base >= 0; base -= 64) {
                const long idx = base + lane;
                const unsigned long long m = __ballot(idx < (long)b_begin && f[idx] != 0);
                if (m) {
                    bl = base + 63 - __builtin_clzll(m);
                    break;
                }
            }
            if (bl >= 0) {
                const int logp = e1 - 2, P = 1 << logp;
                const c32 *tw = reinterpret_cast<const c32 *>(t_twl);
                multi_fetch(sp + os[bl], (size_t)bs1 / 2, lane, v);
                if constexpr (FUSED) multi_fetch(rp + os[bl], (size_t)bs1 / 2, lane, r);
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    float4 x = v[q];
                    if constexpr (FUSED) {
                        x.x *= r[q].x;
                        x.y *= r[q].y;
                        x.z *= r[q].z;
                        x.w *= r[q].w;
                    }
                    reinterpret_cast<float4 *>(ldsf)[lane + 64 * q] = x;
                }
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
                multi_post_twiddle(z, lane, logp, tw, ldsf);
                wave_sync();
                for (int k = bs0 / 2 + 4 * lane; k < bs1 / 2; k += 256)
                    *reinterpret_cast<float4 *>(ovl + k) = *reinterpret_cast<const float4 *>(ldsf + bs1 / 2 + k);
            } else {
                for (int k = bs0 / 2 + 4 * lane; k < bs1 / 2; k += 256)
                    *reinterpret_cast<float4 *>(ovl + k) = *reinterpret_cast<const float4 *>(overlap_in + (size_t)chain * (size_t)(bs1 / 2) + k);
            }
            wave_sync();
        }
        for (int k = 4 * lane; k < bs1 / 2; k += 256)
            *reinterpret_cast<float4 *>(overlap_out + (size_t)chain * (size_t)(bs1 / 2) + k) = *reinterpret_cast<const float4 *>(ovl + k);
        if (lane == 0) prev_flag_out[chain] = f[nb - 1] ? 1 : 0;  // lib.rs:328
    }
}

}  // namespace

int launch_vorbis_wave2(symaccel_ctx *ctx, int bs0_exp, int bs1_exp, const cpx *tw_short, const cpx *tw_long, const float *win_short,
                        const float *win_long, const float *d_spectra, const float *d_residue, size_t spec_stride,
                        const uint8_t *d_block_flag, const int32_t *d_prev_in, int32_t *d_prev_out, const float *d_overlap_in,
                        float *d_overlap_out, float *d_pcm, size_t pcm_stride, const uint32_t *d_offs, size_t n_chains, unsigned nb,
                        unsigned seg) {
    const size_t segs = (nb + seg - 1) / seg;
    const size_t items = n_chains * segs;
    const size_t grid = (items + kWaves - 1) / kWaves;
    if (items > 0xffffffffu || grid > 0x7fffffffu) return SYMACCEL_ERR_INVALID_ARG;
#define SYM_VW2_LAUNCH(FUSED, MAXE1)                                                                                                        \
    hipLaunchKernelGGL((vorbis_synth_wave2_kernel<FUSED, MAXE1>), dim3((unsigned)grid), dim3(64 * kWaves), 0, ctx->stream, ctx->dev, bs0_exp, \
                       bs1_exp, tw_short, tw_long, win_short, win_long, d_spectra, d_residue, spec_stride, d_block_flag, d_prev_in, d_prev_out, \
                       d_overlap_in, d_overlap_out, d_pcm, pcm_stride, d_offs, nb, seg, (unsigned)segs, (unsigned)items)
    if (bs1_exp <= 10) {
        if (d_residue) SYM_VW2_LAUNCH(true, 10); else SYM_VW2_LAUNCH(false, 10);
    } else {
        if (d_residue) SYM_VW2_LAUNCH(true, 11); else SYM_VW2_LAUNCH(false, 11);
    }
#undef SYM_VW2_LAUNCH
    SYM_GPU(ctx, hipGetLastError());
    return SYMACCEL_OK;
}

}  // namespace symaccel
