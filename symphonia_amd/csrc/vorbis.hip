// Vorbis synthesis: DspChannel::synth (symphonia-codec-vorbis/src/dsp.rs:68-145) = Imdct of the
// floor x residue spectrum + the three window/overlap cases, plus the streaming helpers around it:
// inverse coupling and dot product (lib.rs:252-292), residue type-2 de-interleave
// (residue.rs:177-218) and floor-1 curve rendering (floor.rs:568-653, 776-825).
//
// MI355X mapping (DESIGN.md "vorbis_synth"): one wavefront per (chain, segment of blocks); blocks
// have two sizes, so a tiny scan kernel first turns the block-flag sequence into packed spectrum /
// PCM offsets.  Each block: pre-twiddle into LDS (bit-reversed), LDS FFT (fft_lds.h), post-twiddle
// into an LDS PCM tile, overlap-add against the previous block's right half kept in LDS, coalesced
// PCM store.  Segments start with a one-block halo that only rebuilds the overlap.
// Roofline: HBM-bound: 4*(n/2) B in + 4*(prev_n + n)/4 B out per channel-block.
#include <type_traits>
#include "fft_lds.h"

namespace symaccel {

namespace {

// Threads of the generic kernel's workgroup (one workgroup per (chain, segment); its wavefronts share every block's FFT):
// one wavefront for block sizes up to 2048 -- their FFTs are at most 512 points, four wavefronts would mostly wait at the
// barriers (128 / 1024: 2.6 ms with one, 3.8 ms with four) --, four for the 4096 and 8192 blocks (512 / 4096: 5.0 ms
// with one, 2.0 ms with four; profiles/r02w_vorbis_pairs.txt).
template <int MAXBS>
constexpr int vorbis_threads() { return MAXBS > 2048 ? 256 : 64; }

// ---- packed offsets -----------------------------------------------------------------------
// offs[chain][0 .. nb] (spectrum) and offs[chain][nb+1 .. 2nb+1] (pcm), exclusive prefix sums.
__global__ __launch_bounds__(256) void vorbis_offsets_kernel(const uint8_t *__restrict__ flags,
                                                             const int32_t *__restrict__ prev_flag_in,
                                                             uint32_t *__restrict__ offs, unsigned nb, int bs0,
                                                             int bs1) {
    __shared__ uint32_t part_s[256], part_p[256];
    const unsigned chain = blockIdx.x, tid = threadIdx.x;
    const uint8_t *f = flags + (size_t)chain * nb;
    uint32_t *os = offs + (size_t)chain * 2 * (nb + 1), *op = os + (nb + 1);
    const unsigned per = (nb + 255) / 256;
    const unsigned b0 = tid * per, b1 = min(b0 + per, nb);
    const int pf0 = prev_flag_in[chain];
    uint32_t ss = 0, sp = 0;
    for (unsigned b = b0; b < b1; ++b) {
        const int fl = f[b] ? 1 : 0;
        const int pf = b == 0 ? (pf0 < 0 ? fl : (pf0 ? 1 : 0)) : (f[b - 1] ? 1 : 0);
        const int n = fl ? bs1 : bs0, pn = pf ? bs1 : bs0;
        ss += (uint32_t)(n / 2);
        sp += (uint32_t)((pn + n) / 4);  // lib.rs:303
    }
    part_s[tid] = ss;
    part_p[tid] = sp;
    __syncthreads();
    if (tid == 0) {
        uint32_t a = 0, c = 0;
        for (int i = 0; i < 256; ++i) {
            const uint32_t ts = part_s[i], tp = part_p[i];
            part_s[i] = a;
            part_p[i] = c;
            a += ts;
            c += tp;
        }
    }
    __syncthreads();
    ss = part_s[tid];
    sp = part_p[tid];
    for (unsigned b = b0; b < b1; ++b) {
        const int fl = f[b] ? 1 : 0;
        const int pf = b == 0 ? (pf0 < 0 ? fl : (pf0 ? 1 : 0)) : (f[b - 1] ? 1 : 0);
        const int n = fl ? bs1 : bs0, pn = pf ? bs1 : bs0;
        os[b] = ss;
        op[b] = sp;
        ss += (uint32_t)(n / 2);
        sp += (uint32_t)((pn + n) / 4);
        if (b == nb - 1) {
            os[nb] = ss;
            op[nb] = sp;
        }
    }
}

// ---- synthesis ----------------------------------------------------------------------------

template <int MAXBS>
struct VorbisShared {
    c32 fft[fft_padded(MAXBS / 4)];
    float pcm[MAXBS];        // Imdct output of the current block (2N = bs floats)
    float overlap[MAXBS / 2];  // right half of the previous block's Imdct output (dsp.rs:125)
};

// A block's inputs as the lanes consume them: lane `tid` owns the FFT points i = tid + kVT * j.  Fetching a block is
// separate from transforming it so that block b + 1 is in flight while block b is transformed: with a run-time trip count
// and dependent 4-byte loads the pre-twiddle loop used to wait for HBM once per iteration (27 us per 2048-sample block).
template <int MAXBS>
struct VorbisLines {
    static constexpr int J = MAXBS / 4 / vorbis_threads<MAXBS>();  // points per lane of the largest block
    float even[J], mirrored[J];  // spec[2 i], spec[n - 1 - 2 i] (already multiplied by the residue when fused)
    c32 w[J];                    // the block size's Imdct twiddles tw[i] (pre- and post-twiddle use the same ones)
};

template <int MAXBS>
__device__ __forceinline__ void vorbis_fetch_block(VorbisLines<MAXBS> &L, const float *__restrict__ spec,
                                                   const float *__restrict__ res, int bs, const cpx *__restrict__ tw) {
    constexpr int kVT = vorbis_threads<MAXBS>();
    const int n = bs >> 1, nf = bs >> 2;
    float re[VorbisLines<MAXBS>::J], rm[VorbisLines<MAXBS>::J];
#pragma unroll
    for (int j = 0; j < VorbisLines<MAXBS>::J; ++j) {
        const int i = (int)threadIdx.x + kVT * j;
        const bool in = i < nf;
        const int ii = in ? i : 0;
        const cpx w = tw[ii];
        L.w[j] = c32{w.re, w.im};
        L.even[j] = in ? spec[2 * ii] : 0.0f;
        L.mirrored[j] = in ? spec[n - 1 - 2 * ii] : 0.0f;
        re[j] = (res && in) ? res[2 * ii] : 1.0f;
        rm[j] = (res && in) ? res[n - 1 - 2 * ii] : 1.0f;
    }
    if (res) {  // fused dot product (lib.rs:289-291): *f *= r
#pragma unroll
        for (int j = 0; j < VorbisLines<MAXBS>::J; ++j) {
            L.even[j] *= re[j];
            L.mirrored[j] *= rm[j];
        }
    }
}

// Imdct of one fetched block into sh.pcm[0 .. bs) (mdct.rs:67-146).
template <int MAXBS>
__device__ __forceinline__ void vorbis_imdct_block(VorbisShared<MAXBS> &sh, const VorbisLines<MAXBS> &L, int bs, int log2nf,
                                                   const DevTables &tb) {
    constexpr int kVT = vorbis_threads<MAXBS>();
    const int nf = bs >> 2, n4 = bs >> 3;
#pragma unroll
    for (int j = 0; j < VorbisLines<MAXBS>::J; ++j) {
        const int i = (int)threadIdx.x + kVT * j;
        if (i < nf) sh.fft[fft_pad((int)rev_bits((unsigned)i, log2nf))] = pre_twiddle(L.even[j], L.mirrored[j], L.w[j]);
    }
    wg_fft_lds(sh.fft, nf, nf, tb);
    float *vec0 = sh.pcm, *vec1 = sh.pcm + nf, *vec2 = sh.pcm + 2 * nf, *vec3 = sh.pcm + 3 * nf;
#pragma unroll
    for (int j = 0; j < VorbisLines<MAXBS>::J; ++j) {
        const int k = (int)threadIdx.x + kVT * j;
        if (k >= nf) continue;
        const c32 x = sh.fft[fft_pad(k)];
        const c32 val = post_twiddle(x, L.w[j]);
        if (k < n4) {
            const int fi = 2 * k, ri = nf - 1 - 2 * k;
            vec0[ri] = -val.y;
            vec1[fi] = val.y;
            vec2[ri] = val.x;
            vec3[fi] = val.x;
        } else {
            const int i = k - n4;
            const int fi = 2 * i, ri = nf - 1 - 2 * i;
            vec0[fi] = -val.x;
            vec1[ri] = val.x;
            vec2[fi] = val.y;
            vec3[ri] = val.y;
        }
    }
    __syncthreads();
}

template <int MAXBS>
__global__ __launch_bounds__(vorbis_threads<MAXBS>()) void vorbis_synth_kernel(
    DevTables tb, int bs0_exp, int bs1_exp, const cpx *__restrict__ tw_short, const cpx *__restrict__ tw_long,
    const float *__restrict__ win_short, const float *__restrict__ win_long, const float *__restrict__ spectra,
    const float *__restrict__ residue, size_t spec_stride, const uint8_t *__restrict__ flags,
    const int32_t *__restrict__ prev_flag_in, int32_t *__restrict__ prev_flag_out, const float *__restrict__ overlap_in,
    float *__restrict__ overlap_out, float *__restrict__ pcm, size_t pcm_stride, const uint32_t *__restrict__ offs,
    unsigned nb, unsigned seg_len, unsigned segs_per_chain) {
    constexpr int kVT = vorbis_threads<MAXBS>();
    __shared__ VorbisShared<MAXBS> sh;
    const int tid = (int)threadIdx.x;
    const unsigned chain = blockIdx.x / segs_per_chain, seg = blockIdx.x % segs_per_chain;
    const unsigned b_begin = seg * seg_len, b_end = min(b_begin + seg_len, nb);
    const int bs0 = 1 << bs0_exp, bs1 = 1 << bs1_exp;
    const uint8_t *f = flags + (size_t)chain * nb;
    const uint32_t *os = offs + (size_t)chain * 2 * (nb + 1), *op = os + (nb + 1);
    const float *sp = spectra + (size_t)chain * spec_stride;
    const float *rp = residue ? residue + (size_t)chain * spec_stride : nullptr;
    float *out = pcm + (size_t)chain * pcm_stride;
    const int pf0 = prev_flag_in[chain];

    if (b_begin == 0) {
        for (int i = tid; i < bs1 / 2; i += kVT) sh.overlap[i] = overlap_in[(size_t)chain * (size_t)(bs1 / 2) + i];
    }
    __syncthreads();

    const long b_first = b_begin == 0 ? 0 : (long)b_begin - 1;  // halo block rebuilds the overlap only
    bool hi_fresh = b_begin == 0 || bs0 == bs1;  // overlap[bs0/2 .. bs1/2) is what the reference would hold here
    VorbisLines<MAXBS> cur, nxt;
    int flag_next = b_first < (long)b_end ? (f[b_first] ? 1 : 0) : 0;
    if (b_first < (long)b_end) {
        const uint32_t o0 = os[b_first];
        vorbis_fetch_block<MAXBS>(nxt, sp + o0, rp ? rp + o0 : nullptr, flag_next ? bs1 : bs0, flag_next ? tw_long : tw_short);
    }
    for (long b = b_first; b < (long)b_end; ++b) {
        const int flag = flag_next;
        const int pflag = b == 0 ? (pf0 < 0 ? flag : (pf0 ? 1 : 0)) : (f[b - 1] ? 1 : 0);  // lib.rs:298
        const int bs = flag ? bs1 : bs0;
        hi_fresh = hi_fresh || flag;
        cur = nxt;
        if (b + 1 < (long)b_end) {  // the next block's lines and twiddles travel while this one is transformed
            flag_next = f[b + 1] ? 1 : 0;
            const uint32_t o1 = os[b + 1];
            vorbis_fetch_block<MAXBS>(nxt, sp + o1, rp ? rp + o1 : nullptr, flag_next ? bs1 : bs0, flag_next ? tw_long : tw_short);
        }
        vorbis_imdct_block<MAXBS>(sh, cur, bs, (flag ? bs1_exp : bs0_exp) - 2, tb);
        if (b >= (long)b_begin) {
            float *__restrict__ o = out + op[b];
            const float *__restrict__ win = (flag && pflag) ? win_long : win_short;  // dsp.rs:83
            // (the loops below are unrolled by four so that the window loads of four iterations are in flight together:
            // with a run-time trip count the compiler otherwise waits for each iteration's pair of loads in turn)
            if (pflag == flag) {  // dsp.rs:85-90
                const int len = bs / 2;
#pragma unroll 4
                for (int k = tid; k < len; k += kVT)
                    o[k] = sh.overlap[k] * win[len - 1 - k] + sh.pcm[k] * win[k];
            } else if (pflag && !flag) {  // long -> short, dsp.rs:91-106
                const int start = (bs1 - bs0) / 4, len = bs0 / 2;
#pragma unroll 4
                for (int k = tid; k < start; k += kVT) o[k] = sh.overlap[k];
#pragma unroll 4
                for (int k = tid; k < len; k += kVT)
                    o[start + k] = sh.overlap[start + k] * win[len - 1 - k] + sh.pcm[k] * win[k];
            } else {  // short -> long, dsp.rs:107-122
                const int start = (bs1 - bs0) / 4, len = bs0 / 2, end = start + len;
#pragma unroll 4
                for (int k = tid; k < len; k += kVT)
                    o[k] = sh.overlap[k] * win[len - 1 - k] + sh.pcm[start + k] * win[k];
#pragma unroll 4
                for (int k = tid; k < bs1 / 2 - end; k += kVT) o[len + k] = sh.pcm[end + k];
            }
        }
        __syncthreads();
#pragma unroll 4
        for (int k = tid; k < bs / 2; k += kVT) sh.overlap[k] = sh.pcm[bs / 2 + k];  // dsp.rs:125
        __syncthreads();
    }

    if (b_end == nb) {
        if (!hi_fresh) {
            // The chain ends in short blocks and this segment never saw a long one: overlap[bs0/2 .. bs1/2) still holds
            // what the most recent long block left there (dsp.rs:125 only rewrites the first bs/2 entries; never used
            // for PCM, but part of the state the reference carries).  Rebuild it from that block, or keep the incoming
            // state if the batch has no long block before this segment.
            __shared__ long bl_shared;
            if (tid < 64) {  // the first wavefront searches 64 flags at a time: one coalesced byte load + ballot per step
                long found = -1;
                for (long base = ((long)b_begin - 1) & ~63l; base >= 0; base -= 64) {
                    const long idx = base + tid;
                    const unsigned long long m = __ballot(idx < (long)b_begin && f[idx] != 0);
                    if (m) {
                        found = base + 63 - __builtin_clzll(m);
                        break;
                    }
                }
                if (tid == 0) bl_shared = found;
            }
            __syncthreads();
            const long bl = bl_shared;
            if (bl >= 0) {
                vorbis_fetch_block<MAXBS>(cur, sp + os[bl], rp ? rp + os[bl] : nullptr, bs1, tw_long);
                vorbis_imdct_block<MAXBS>(sh, cur, bs1, bs1_exp - 2, tb);
                for (int k = bs0 / 2 + tid; k < bs1 / 2; k += kVT) sh.overlap[k] = sh.pcm[bs1 / 2 + k];
            } else {
                for (int k = bs0 / 2 + tid; k < bs1 / 2; k += kVT)
                    sh.overlap[k] = overlap_in[(size_t)chain * (size_t)(bs1 / 2) + k];
            }
            __syncthreads();
        }
        for (int i = tid; i < bs1 / 2; i += kVT) overlap_out[(size_t)chain * (size_t)(bs1 / 2) + i] = sh.overlap[i];
        if (tid == 0) prev_flag_out[chain] = f[nb - 1] ? 1 : 0;  // lib.rs:328
    }
}

// ---- streaming helpers ----------------------------------------------------------------------

// What stands between the residue decoder and the dot product for a whole batch in the packed layout, in place (lib.rs:250-292):
// the inverse coupling steps of every block in their order (steps may chain, lib.rs:252), then -- for the channels whose floor is
// unused but whose residue was decoded because a coupling partner's floor is in use (lib.rs:215-228) -- the product with the
// all-zero floor, `0.0 * r` (lib.rs:289-291: the sign of the zero, and a NaN from an infinite residue, are the reference's).
// A workgroup per (stream, block): the channels of a stream share the block flags, so block b's lines sit at the same offset of
// every channel's row.  block_off[stream][blocks + 1] in lines; steps[][2] = (magnitude, angle) channel of the stream;
// step_first[stream * blocks + block] .. [+ 1) = the block's steps; kill[chain][block] != 0: times +0.0.
__global__ __launch_bounds__(256) void vorbis_prepare_kernel(float *__restrict__ residue, size_t spec_stride, unsigned cps, unsigned blocks,
                                                             const uint32_t *__restrict__ block_off, const uint8_t *__restrict__ steps,
                                                             const uint32_t *__restrict__ step_first, const uint8_t *__restrict__ kill) {
    const unsigned sb = blockIdx.x, stream = sb / blocks, b = sb % blocks;
    const uint32_t off = block_off[(size_t)stream * (blocks + 1) + b], n2 = block_off[(size_t)stream * (blocks + 1) + b + 1] - off;
    const uint32_t s0 = step_first[sb], s1 = step_first[sb + 1];
    float *base = residue + (size_t)stream * cps * spec_stride + off;
    for (uint32_t i = threadIdx.x; i < n2; i += 256) {
        for (uint32_t s = s0; s < s1; ++s) {
            float *mp = base + (size_t)steps[2 * s] * spec_stride + i, *ap = base + (size_t)steps[2 * s + 1] * spec_stride + i;
            const float m = *mp, a = *ap;
            float nm, na;
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
            *mp = nm;
            *ap = na;
        }
        for (unsigned c = 0; c < cps; ++c)
            if (kill[((size_t)stream * cps + c) * blocks + b]) {
                float *p = base + (size_t)c * spec_stride + i;
                *p = 0.0f * *p;
            }
    }
}

// lib.rs:265-277
__global__ void vorbis_coupling_kernel(float *__restrict__ mag, float *__restrict__ ang, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const float m = mag[i], a = ang[i];
        float nm, na;
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

// lib.rs:289-291
__global__ void vorbis_dot_kernel(float *__restrict__ floor, const float *__restrict__ residue, size_t total) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x)
        floor[i] *= residue[i];
}

// residue.rs:177-218: planar[c][i] = type2[i * n_ch + c]
__global__ void vorbis_deinterleave_kernel(const float *__restrict__ type2, float *__restrict__ planar, int n_ch,
                                           size_t n2, size_t count) {
    const size_t per = n2 * (size_t)n_ch, total = per * count;
    for (size_t o = (size_t)blockIdx.x * blockDim.x + threadIdx.x; o < total; o += (size_t)gridDim.x * blockDim.x) {
        const size_t blk = o / per, r = o % per;
        const size_t c = r / n2, i = r % n2;
        planar[o] = type2[blk * per + i * (size_t)n_ch + c];
    }
}

// Floor-1 curve synthesis (floor.rs:568-653, 776-825).  A wavefront takes 64 channel-blocks:
//   step 1 (the post-value recurrence over <= 65 posts, with the setup's neighbour tables) runs one LANE per block
//   -- the post index, neighbours and x values are wave-uniform (kernel arguments, scalar), only the y values differ;
//   step 2 builds each block's list of line end points (x-sorted, flagged posts only) in LDS;
//   rendering takes the blocks one after the other: one lane per SEGMENT derives the segment's constants once (LDS table),
//   then every lane renders 16 consecutive x through the closed form of render_line's integer DDA (see the render loop).
__device__ __forceinline__ void wave_sync_lds() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// byte `sel` of `old` replaced by the f32 value converted to 0 .. 255 (v_cvt_pk_u8_f32; the values it is given are exact integers)
__device__ __forceinline__ uint32_t cvt_pk_u8(float v, int sel, uint32_t old) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_cvt_pk_u8_f32(v, (uint32_t)sel, old);
#else
    const float cl = v < 0.0f ? 0.0f : (v > 255.0f ? 255.0f : v);
    return (old & ~(255u << (8 * sel))) | ((uint32_t)cl << (8 * sel));
#endif
}

// Across the wavefront: excl = the maximum of v over the LOWER lanes (0 in lane 0), total = the maximum over all lanes.  Six DPP
// steps (within rows of 16 lanes, then the rows' last lanes broadcast to the rows above), one lane shift and one v_readlane -- the
// values are unsigned, 0 is the identity.
__device__ __forceinline__ void wave_prefix_max(uint32_t v, uint32_t &excl, uint32_t &total) {
#if defined(__HIP_DEVICE_COMPILE__)
    int s = (int)v;
#define SYM_DPP_MAX(ctrl, rows) s = max(s, __builtin_amdgcn_update_dpp(s, s, ctrl, rows, 0xf, true))  // (lanes without a source: 0 or s itself)
    SYM_DPP_MAX(0x111, 0xf);  // row_shr:1
    SYM_DPP_MAX(0x112, 0xf);  // row_shr:2
    SYM_DPP_MAX(0x114, 0xf);  // row_shr:4
    SYM_DPP_MAX(0x118, 0xf);  // row_shr:8
    SYM_DPP_MAX(0x142, 0xa);  // row_bcast:15 into rows 1 and 3
    SYM_DPP_MAX(0x143, 0xc);  // row_bcast:31 into rows 2 and 3
#undef SYM_DPP_MAX
    excl = (uint32_t)__builtin_amdgcn_update_dpp(0, s, 0x138, 0xf, 0xf, false);  // wave_shr:1
    total = (uint32_t)__builtin_amdgcn_readlane(s, 63);
#else
    const unsigned lane = threadIdx.x & 63u;
    int s = (int)v;
    for (unsigned d = 1; d < 64; d <<= 1) {
        const int t = __shfl_up(s, d);
        if (lane >= d) s = s > t ? s : t;
    }
    const int e = __shfl_up(s, 1);
    excl = lane ? (uint32_t)e : 0u;
    total = (uint32_t)__shfl(s, 63);
#endif
}

struct Floor1Setup {  // per floor configuration, derived on the host like the setup parser does (floor.rs:540-555)
    // Everything a loop iteration needs sits at an address that depends on the loop counter only: the scalar loads of
    // several iterations go out together instead of lo -> x[lo] chains of dependent round trips.
    uint32_t nb[65];    // post i: lo | hi << 8 (floor1_x_list_neighbors) | wide << 16 (neighbour span above 4096, see `wide`)
    float ratio[65];    //         (x[i] - x[lo]) / (x[hi] - x[lo]) and
    float half[65];     //         0.5 / (x[hi] - x[lo]), both rounded to f32: render_point in closed form (see step 1)
    uint32_t wide[65];  //         (x[i] - x[lo]) | (x[hi] - x[lo]) << 16: render_point in integers, as the reference writes it, for the
                        //         posts and lanes outside the closed form's proven range (spans above 4096, |dy| above 511)
    uint32_t ord[65];   // x-sorted position k: order[k] | x[order[k]] << 16
};

// A workgroup of four wavefronts takes 64 channel-blocks.  Steps 1 and 2a are a chain of <= 65 dependent posts per block
// whose instruction count does not depend on how many lanes are busy, so ONE wavefront runs them, a lane per block, for all
// 64 blocks (the other three wait at the barrier and cost no issue slots); then every wavefront renders 16 of the blocks.
// Measured (profiles/r02zb_floor1_ab.txt): a wavefront per 16 or 32 blocks doing everything itself spends as many
// instructions in the post chain as in the render, and the kernel is bound by instruction issue and dependent latency, not
// by HBM.
// SYM_F1_ABLATE (measurement only, never in the product build: results are WRONG): 1 = no post chain (synthesis_step1), 2 = no point
// lists (two points per block), 4 = no render pass, 8 = no segment tables / maps.  profiles/r06o_floor1_phases.txt
#ifndef SYM_F1_ABLATE
#define SYM_F1_ABLATE 0
#endif
#ifndef SYM_F1_LANE16
#define SYM_F1_LANE16 1  // the byte render with consecutive lines per lane (floor1_workgroup, MODE 2): sixteen, four for blocks of <= 256 lines; 2: sixteen for every size; 0: four groups of four lines 256 apart, as the f32 forms
#endif
constexpr int kF1B = 64;                 // channel-blocks per workgroup
#ifndef SYM_F1_WAVES
#define SYM_F1_WAVES 4
#endif
constexpr int kF1Waves = SYM_F1_WAVES;   // wavefronts per workgroup
constexpr int kF1Stride = kF1B + 1;      // LDS row stride of the per-block lists [entry][block]: conflict-free both ways

// MODE 0: the curve (f32) -> floor_out[block][n];  1: curve * residue -> floor_out (the dot product of lib.rs:282-292);
//      2: the curve's dB-table INDICES, one byte per line -> (uint8_t *)floor_out + line_offs[block] (or block * n): every value
//         render_line writes is FLOOR1_INVERSE_DB_TABLE[y], y in 0..255 (floor.rs:785-825), so one byte per line carries the whole
//         curve; the synthesis kernels look the table up as they load the residue (symaccel_vorbis_synth_fy_*): 1 B / line
//         written here and read there instead of 4 + 4 + 4 B / line for a multiplied spectrum.
template <int MODE, int NMAX>
__device__ __forceinline__ void floor1_workgroup(const Floor1Setup &st, int n_posts, int multiplier, const uint32_t *__restrict__ yv, uint32_t n,
                                                 float *floor_out, const float *__restrict__ db, size_t count, const float *residue,
                                                 const uint32_t *__restrict__ line_offs, unsigned wg) {
    constexpr bool DOT = MODE == 1;
    // LDS per workgroup: 13 KiB of point lists + 8.2 KiB (n <= 1024; 20 KiB otherwise) that first hold final_y and then, per
    // wavefront, a segment table and a segment-start map (n bytes): 22.6 KiB, seven workgroups per CU
    constexpr int kPerWave = 67 * 16 + NMAX;  // bytes
    constexpr int kOverlay = 65 * kF1B * 2 > kF1Waves * kPerWave ? 65 * kF1B * 2 : kF1Waves * kPerWave;
    __shared__ __attribute__((aligned(16))) uint8_t overlay[kOverlay];
    __shared__ uint16_t segx[67 * kF1Stride];     // first the y values [post][block], then the points' x
    __shared__ uint8_t segy[67 * kF1Stride];      //                                               ... and y (0..255)
    __shared__ uint8_t ns_of[kF1B];               // points 0 .. ns_of[b] of block b
    __shared__ unsigned long long used_lo[kF1B];  // floor_step2_flag of posts 0..63 ...
    __shared__ uint8_t used_hi[kF1B];             // ... and of post 64
    __shared__ float dbl[256];
    const int tid = (int)threadIdx.x, lane = tid & 63, wave = tid >> 6;
    int16_t *fy = reinterpret_cast<int16_t *>(overlay);  // final_y[post][block] (|final_y| < 2^9), steps 1 and 2a
    // the render's: constants of the segments of the block being rendered (x0 | 4 y0 << 16, +-|dy| / adx, +-0.5 / adx) ...
    uint4 *segc = reinterpret_cast<uint4 *>(overlay + wave * kPerWave);
    uint8_t *mark = overlay + wave * kPerWave + 67 * 16;  // ... and its segment-start map
    const size_t blk0 = (size_t)wg * kF1B;
    const int nb = (int)(count - blk0 < (size_t)kF1B ? count - blk0 : (size_t)kF1B);
    if (tid < 256) dbl[tid] = db[tid];
    // the y rows of the 64 blocks are contiguous: coalesced load by the whole workgroup, transposed into LDS
    {
        const uint32_t *src = yv + blk0 * (size_t)n_posts;
        const int total = nb * n_posts;
        // element e = block * n_posts + post; (block, post) advance by 256 elements without a division per element
        constexpr int kT = 64 * kF1Waves;
        const int dq = kT / n_posts, dr = kT % n_posts;
        int blk = tid / n_posts, post = tid % n_posts;
        for (int e = tid; e < total; e += kT) {
            segx[post * kF1Stride + blk] = (uint16_t)src[e];
            post += dr;
            blk += dq + (post >= n_posts ? 1 : 0);
            post -= post >= n_posts ? n_posts : 0;
        }
    }
    __syncthreads();
    if (wave == 1) {
        // floor_step2_flag (floor.rs:599-601): a property of the y values alone, so a second wavefront derives it while the
        // first walks the post chain.  Bits (posts 0 and 1 are always used): 64 + one for post 64, set and tested without
        // indexing an array by a run-time value (that would put the array in scratch memory)
        unsigned long long flag_lo = 3ull;
        unsigned flag_hi = 0u;
#pragma unroll 4
        for (int i = 2; i < n_posts; ++i) {
            const uint32_t pn = st.nb[i];
            const int lo = (int)(pn & 255u), hi = (int)((pn >> 8) & 255u);
            const bool coded = segx[i * kF1Stride + lane] != 0;
            const unsigned long long used = (lo < 64 ? 1ull << lo : 0ull) | (hi < 64 ? 1ull << hi : 0ull) | (i < 64 ? 1ull << i : 0ull);
            flag_lo |= coded ? used : 0ull;
            flag_hi |= (coded && (lo == 64 || hi == 64 || i == 64)) ? 1u : 0u;
        }
        used_lo[lane] = flag_lo;
        used_hi[lane] = (uint8_t)flag_hi;
    }
    if (wave == 0) {
        // ---- synthesis_step1 (floor.rs:568-625): lane = block.  The chain from one post to the next is what this phase costs
        // (each post needs final_y of two earlier ones), so what does not depend on it -- the post's constants (scalar loads)
        // and its y value -- is requested one post ahead, and the chain itself is as short as it gets:
        // render_point (floor.rs:776-782) is y0 +- floor(|dy| * (x - x0) / adx), and
        //     floor(|dy| * dx / adx) == trunc(f32(|dy|) * ratio + half),   ratio = f32(dx) / f32(adx), half = 0.5f / f32(adx)
        // exactly for |dy| <= 511 (final_y stays within -42 .. 255) and adx <= 4096 (tests/cpp/floor1_division_check.c);
        // the sign of dy goes into both terms (the conversion truncates towards zero).
        const int32_t range = multiplier == 1 ? 256 : multiplier == 2 ? 128 : multiplier == 3 ? 86 : 64;
        fy[0 * kF1B + lane] = (int16_t)segx[0 * kF1Stride + lane];
        fy[1 * kF1B + lane] = (int16_t)segx[1 * kF1Stride + lane];
        uint32_t pn_next = st.nb[2];
        float ratio_next = st.ratio[2], half_next = st.half[2];
        int32_t val_next = (int32_t)segx[2 * kF1Stride + lane];
        for (int i = 2; i < ((SYM_F1_ABLATE & 1) ? 3 : n_posts); ++i) {
            const uint32_t pn = pn_next;
            const int32_t val = val_next;
            const float ratio = ratio_next, half = half_next;
            {
                const int j = i + 1 < n_posts ? i + 1 : i;
                pn_next = st.nb[j];
                ratio_next = st.ratio[j];
                half_next = st.half[j];
                val_next = (int32_t)segx[j * kF1Stride + lane];
            }
            const int lo = (int)(pn & 255u), hi = (int)((pn >> 8) & 255u);
            const int32_t py0 = fy[lo * kF1B + lane];
            const int32_t dy = fy[hi * kF1B + lane] - py0;
            const float fdy = (float)dy;
            int32_t predicted = py0 + (int32_t)(fdy * ratio + __builtin_copysignf(half, fdy));
            const uint32_t ady = (uint32_t)(dy < 0 ? -dy : dy);
            if ((pn & 0x10000u) || ady > 511u) {
                // Outside the closed form's proven range, the integer form of floor.rs:776-782 itself:
                //  * neighbours more than 4096 apart (a floor whose posts reach past every block size: rangebits up to 15 are
                //    legal; the closed form first fails at adx = 17019) -- a property of the setup, wave-uniform, and no
                //    conforming encoder's setup has it;
                //  * |dy| above 511: only y values beyond the floor's range produce it (they are codebook entry numbers,
                //    floor.rs:698-712, so a hostile stream can carry them; up to 511 the final_y stay inside int16) -- per lane,
                //    and no lane of a conforming stream takes it.
                const uint32_t w = st.wide[i];
                const uint32_t off = (ady * (w & 0xffffu)) / (w >> 16);
                predicted = dy < 0 ? py0 - (int32_t)off : py0 + (int32_t)off;
            }
            // floor.rs:596-621 as selects (the lanes of a wavefront take all the branches anyway); lowroom = predicted, so
            // `val - lowroom + predicted` is val and `predicted - val + highroom - 1` is range - val - 1
            const int32_t highroom = range - predicted, lowroom = predicted;
            const int32_t room = 2 * (highroom < lowroom ? highroom : lowroom);
            const int32_t far = highroom > lowroom ? val : range - val - 1;
            const int32_t near = (val & 1) ? predicted - ((val + 1) >> 1) : predicted + (val >> 1);
            const int32_t fin = val == 0 ? predicted : (val >= room ? far : near);
            fy[i * kF1B + lane] = (int16_t)fin;
        }
    }
    __syncthreads();  // final_y and the flags are complete; every lane has consumed its y values: segx / segy become the point lists
    if (wave == 0) {
        const unsigned long long flag_lo = used_lo[lane];
        const unsigned flag_hi = used_hi[lane];

        // ---- synthesis_step2 (floor.rs:627-653), first half: the x-sorted list of line end points of this lane's block
        int ns = 0;
        int32_t ly = fy[(st.ord[0] & 255u) * kF1B + lane] * multiplier;
        ly = ly < 0 ? 0 : (ly > 255 ? 255 : ly);
        segx[0 * kF1Stride + lane] = 0;  // (x = 0, y = ly)
        segy[0 * kF1Stride + lane] = (uint8_t)ly;
        uint32_t hx = 0;
        int32_t hy = 0;
        // (the reads do not depend on the list being built: unrolled, their latencies overlap)
#pragma unroll 4
        for (int k = 1; k < ((SYM_F1_ABLATE & 2) ? 2 : n_posts); ++k) {
            const uint32_t po = st.ord[k];
            const int i = (int)(po & 255u);
            int32_t py = fy[i * kF1B + lane] * multiplier;
            py = py < 0 ? 0 : (py > 255 ? 255 : py);
            if (i < 64 ? (unsigned)((flag_lo >> (i & 63)) & 1ull) : flag_hi) {
                hy = py;
                hx = po >> 16;
                ++ns;
                segx[ns * kF1Stride + lane] = (uint16_t)hx;
                segy[ns * kF1Stride + lane] = (uint8_t)hy;
            }
        }
        if (hx < n) {  // flat tail (floor.rs:650-652)
            ++ns;
            segx[ns * kF1Stride + lane] = (uint16_t)n;
            segy[ns * kF1Stride + lane] = (uint8_t)hy;
        }
        ns_of[lane] = (uint8_t)ns;
    }
    __syncthreads();

    // ---- render_line for every segment (floor.rs:785-825): wavefront w takes blocks w, w + 4, ... one after the other.
    // Which segment an x belongs to comes from a byte map of the segment starts (scattered by the lanes that hold the
    // points); the segments' constants from a table built one lane per segment.
    for (int b = wave; b < nb; b += kF1Waves) {
        const int nsb = (int)ns_of[b];  // points 0 .. nsb of block b
        // (line_offs: block b's lines start at line line_offs[b] of the plane instead of b * n -- a block-size class of a mixed stream
        // rendered straight into the packed layout the synthesis kernels read)
        const size_t line0 = line_offs ? (size_t)line_offs[blk0 + (size_t)b] : (blk0 + (size_t)b) * (size_t)n;
        float *out = floor_out + line0;
        uint8_t *yout = nullptr;
        if constexpr (MODE == 2) yout = reinterpret_cast<uint8_t *>(floor_out) + line0;
        // fused dot product (lib.rs:282-292): the curve is multiplied by the block's residue as it is stored -- one rounded
        // multiply per line, the reference's `*f *= r` -- so the curve itself never goes to HBM (residue may be `out`)
        const float *rin = DOT ? residue + line0 : nullptr;
        // segment-start map: mark[x_k] = k + 1 for the points with x_k < n (x values are distinct)
#if !(SYM_F1_ABLATE & 8)
        for (uint32_t i = (uint32_t)lane; i < (n + 3u) / 4u; i += 64) reinterpret_cast<uint32_t *>(mark)[i] = 0u;
        wave_sync_lds();
#endif
        // ... and the constants of segment k (point k to point k + 1), one lane per segment.  The integer DDA of render_line
        // (floor.rs:785-825: y += base every x, one more step of sign(dy) whenever err overflows adx) has the closed form
        //     y(x) = y0 + sign(dy) * floor(|dy| * t / adx),   t = x - x0 < adx
        // (base * t + sign * floor(ady * t / adx) with base = sign * floor(|dy| / adx), ady = |dy| mod adx), and the floor is
        //     trunc(f32(t) * slope + half),   slope = f32(|dy|) / f32(adx),   half = 0.5f / f32(adx)      (both rounded)
        // EXACTLY: |dy| * t / adx + 0.5 / adx is at least 0.5 / adx away from an integer on either side, the three roundings
        // move the value by less than 255 * 3 * 2^-24 < 0.5 / 8192.  tests/cpp/floor1_division_check.c walks every
        // (adx <= 4096, |dy| <= 255, t < adx).  The sign goes into slope and half (the conversion truncates towards zero).
        // Segments LONGER than 4096 exist (posts past the block, or few flagged posts under a large rangebits: adx up to
        // 65535), but only their first n <= 4096 lines are rendered: the value is below 255 * 4096 / adx there, the three
        // roundings move it by less than 3 * 255 * 4096 * 2^-24 / adx = 0.19 / adx < 0.5 / adx -- exact again; the same
        // program walks (4096 < adx <= 65535, |dy| <= 255, t < 4096) (every adx with SYM_SLOW_TESTS=1, every 16th otherwise).
        for (int k = lane; k <= ((SYM_F1_ABLATE & 8) ? -1 : nsb); k += 64) {
            const int k1 = k + 1 <= nsb ? k + 1 : k;
            const uint32_t xk = segx[k * kF1Stride + b];
            if (xk < n) mark[xk] = (uint8_t)(k + 1);
            const int32_t y0 = (int32_t)segy[k * kF1Stride + b];
            const int32_t dy = (int32_t)segy[k1 * kF1Stride + b] - y0;
            int32_t adx = (int32_t)segx[k1 * kF1Stride + b] - (int32_t)xk;
            adx = adx > 0 ? adx : 1;
            const float fadx = (float)adx;
            // MODE 2 (bytes out): x0 and y0 as floats -- the line's arithmetic stays in f32 (exact: small integers) and ends in
            // v_cvt_pk_u8_f32, which converts and places the byte in one instruction
            if constexpr (MODE == 2)
                segc[k] = make_uint4(__float_as_uint((float)xk), __float_as_uint((float)dy / fadx),
                                     __float_as_uint((dy < 0 ? -0.5f : 0.5f) / fadx), __float_as_uint((float)y0));
            else
                segc[k] = make_uint4(xk | ((uint32_t)y0 << 18), __float_as_uint((float)dy / fadx),
                                     __float_as_uint((dy < 0 ? -0.5f : 0.5f) / fadx), 0u);
        }
        wave_sync_lds();
        int carry = 1;  // segment (index + 1) in force before the current pass; x = 0 always starts segment 0
        // 1024 lines per pass in four groups of 256: in group j lane l renders x = p0 + 256 j + 4 l .. + 3, so that a store
        // instruction writes 1 KiB without a gap (16 consecutive x per lane left every 64-byte unit of a store three quarters
        // empty: four times the write requests).  The pass is straight-line code -- every LDS round trip (map, lane reads,
        // table, dB values) is issued for all four groups before the first result is needed.
#if SYM_F1_LANE16
        if constexpr (MODE == 2) {
            // Bytes out: lane l renders the SIXTEEN consecutive lines p0 + 16 l .. + 15 of a pass -- 16 bytes per lane, so a store instruction still writes 1 KiB without a
            // gap (the four-lines-per-group layout below exists for the f32 forms, where 16 consecutive lines per lane are 64 bytes) -- with ONE prefix maximum across the
            // wavefront, one 16-byte read of the segment-start map and one 16-byte store per 16 lines instead of four of each; the segment in force runs on through the
            // lane's lines.  (`aligned16`: wave-uniform; a packed layout whose block offsets are not multiples of 16 takes four 4-byte stores.)
            const bool aligned16 = ((reinterpret_cast<uintptr_t>(yout)) & 15u) == 0;
            // W = words of four lines per lane and pass: 4 (sixteen lines, passes of 1024) -- or 1 for blocks of at most 256 lines (the short blocks of a stream:
            // a 128-line block keeps 32 lanes busy for four lines each instead of 8 lanes for sixteen; a quarter of the pass's instructions)
            auto render_bytes = [&](auto w_tag) {
                constexpr int W = decltype(w_tag)::value;
                for (uint32_t p0 = 0; p0 < ((SYM_F1_ABLATE & 4) ? 0u : n); p0 += 256u * (uint32_t)W) {
                    const uint32_t x0 = p0 + 4u * (uint32_t)W * (uint32_t)lane;  // (n is a multiple of 16)
                    const bool live = x0 < n;
                    uint32_t m[W];
#pragma unroll
                    for (int j = 0; j < W; ++j) m[j] = 0u;
                    if (live) {
                        if constexpr (W == 4) {
                            const uint4 mm = *reinterpret_cast<const uint4 *>(mark + x0);
                            m[0] = mm.x; m[1] = mm.y; m[2] = mm.z; m[3] = mm.w;
                        } else {
                            m[0] = *reinterpret_cast<const uint32_t *>(mark + x0);
                        }
                    }
                    uint32_t mine = 0;
#pragma unroll
                    for (int j = 0; j < W; ++j) mine = max(mine, max(max(m[j] & 255u, (m[j] >> 8) & 255u), max((m[j] >> 16) & 255u, m[j] >> 24)));
                    uint32_t before, last;
                    wave_prefix_max(mine, before, last);
                    int seg_id = (int)before > carry ? (int)before : carry;  // (index + 1) of the segment in force in front of the lane's first line
                    carry = (int)last > carry ? (int)last : carry;
                    const float xf0 = (float)x0;
                    uint32_t yb[W];
#pragma unroll
                    for (int j = 0; j < W; ++j) {
                        yb[j] = 0u;
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            const int v = (int)((m[j] >> (8 * q)) & 255u);
                            seg_id = v > seg_id ? v : seg_id;
                            const uint4 c = segc[seg_id - 1];
                            const float tf = (xf0 + (float)(4 * j + q)) - __uint_as_float(c.x);
                            const float yf = __uint_as_float(c.w) + __builtin_truncf(tf * __uint_as_float(c.y) + __uint_as_float(c.z));
                            yb[j] = cvt_pk_u8(yf, q, yb[j]);
                        }
                    }
                    if (live) {
                        if constexpr (W == 4) {
                            if (aligned16) {
                                *reinterpret_cast<uint4 *>(yout + x0) = make_uint4(yb[0], yb[1], yb[2], yb[3]);
                            } else {
#pragma unroll
                                for (int j = 0; j < 4; ++j) *reinterpret_cast<uint32_t *>(yout + x0 + 4u * (uint32_t)j) = yb[j];
                            }
                        } else {
                            *reinterpret_cast<uint32_t *>(yout + x0) = yb[0];
                        }
                    }
                }
            };
            if (SYM_F1_LANE16 == 1 && n <= 256u) render_bytes(std::integral_constant<int, 1>{});  // (wave-uniform; SYM_F1_LANE16 = 2: sixteen lines per lane for every block size)
            else render_bytes(std::integral_constant<int, 4>{});
            wave_sync_lds();  // the next block's segment-start map overwrites this one's
            continue;
        }
#endif
        for (uint32_t p0 = 0; p0 < ((SYM_F1_ABLATE & 4) ? 0u : n); p0 += 1024) {
            uint32_t xb[4], m[4];
            float4 rr[4];  // the lines' residue, requested now: the render hides the latency
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                xb[j] = p0 + 256u * (uint32_t)j + 4u * (uint32_t)lane;  // (n is a multiple of 16)
                rr[j] = make_float4(1.0f, 1.0f, 1.0f, 1.0f);
                if constexpr (DOT) {
                    if (xb[j] < n) rr[j] = *reinterpret_cast<const float4 *>(rin + xb[j]);
                }
                m[j] = xb[j] < n ? *reinterpret_cast<const uint32_t *>(mark + xb[j]) : 0u;
            }
            // the segment in force just before a lane's first x: the highest mark below it -- marks grow with x, so that is the
            // maximum over the lower lanes (a prefix maximum across the wavefront), or else the highest mark of the groups before
            uint32_t before[4], last[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const uint32_t mine = max(max(m[j] & 255u, (m[j] >> 8) & 255u), max((m[j] >> 16) & 255u, m[j] >> 24));
                wave_prefix_max(mine, before[j], last[j]);
            }
            float res[4][4];
            uint32_t ybytes[4] = {0u, 0u, 0u, 0u};
            float xf[4][4];  // MODE 2: the lines' x as floats (one conversion per group, exact increments)
            if constexpr (MODE == 2) {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    xf[j][0] = (float)xb[j];
#pragma unroll
                    for (int q = 1; q < 4; ++q) xf[j][q] = xf[j][0] + (float)q;
                }
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                int seg_id = (int)before[j] > carry ? (int)before[j] : carry;  // (index + 1) of the segment
                carry = (int)last[j] > carry ? (int)last[j] : carry;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int v = (int)((m[j] >> (8 * q)) & 255u);
                    seg_id = v > seg_id ? v : seg_id;
                    const uint4 c = segc[seg_id - 1];
                    if constexpr (MODE == 2) {
                        // the same closed form on the f32 side: t = x - x0 as a difference of two exactly represented integers, the
                        // truncation as v_trunc_f32, y0 + steps as an f32 add of small integers; y is in 0 .. 255 for every rendered x
                        // (lanes past the list convert garbage, saturated by the instruction, and do not store)
                        const float tf = xf[j][q] - __uint_as_float(c.x);
                        const float yf = __uint_as_float(c.w) + __builtin_truncf(tf * __uint_as_float(c.y) + __uint_as_float(c.z));
                        ybytes[j] = cvt_pk_u8(yf, q, ybytes[j]);
                    } else {
                        const int32_t t = (int32_t)(xb[j] + (uint32_t)q) - (int32_t)(c.x & 0xffffu);
                        const int32_t steps = (int32_t)((float)t * __uint_as_float(c.y) + __uint_as_float(c.z));
                        int32_t y4 = (int32_t)(c.x >> 16) + (steps << 2);  // byte offset of the table entry (y0 sits at bit 18)
                        y4 = y4 < 0 ? 0 : (y4 > 1020 ? 1020 : y4);        // (in range for every rendered x; guards the lanes past the list)
                        res[j][q] = *reinterpret_cast<const float *>(reinterpret_cast<const char *>(dbl) + y4);
                    }
                }
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                if constexpr (MODE == 2) {
                    if (xb[j] < n) *reinterpret_cast<uint32_t *>(yout + xb[j]) = ybytes[j];  // 256 B per store instruction
                } else if (xb[j] < n) {
                    float4 v = make_float4(res[j][0], res[j][1], res[j][2], res[j][3]);
                    if constexpr (DOT) v = make_float4(v.x * rr[j].x, v.y * rr[j].y, v.z * rr[j].z, v.w * rr[j].w);
                    *reinterpret_cast<float4 *>(out + xb[j]) = v;
                }
            }
        }
        wave_sync_lds();  // the next block's segment-start map overwrites this one's
    }
}

template <int MODE, int NMAX>
__global__ __launch_bounds__(64 * kF1Waves) void vorbis_floor1_kernel(Floor1Setup st, int n_posts, int multiplier,
                                                                      const uint32_t *__restrict__ yv, uint32_t n,
                                                                      float *floor_out, const float *__restrict__ db,
                                                                      size_t count, const float *residue,
                                                                      const uint32_t *__restrict__ line_offs) {
    floor1_workgroup<MODE, NMAX>(st, n_posts, multiplier, yv, n, floor_out, db, count, residue, line_offs, blockIdx.x);
}

// Two floor-1 jobs in ONE grid (byte form): the block-size classes of a stream's floor -- or two floors -- are independent renders into
// the same plane, and as two launches the second waits for the first's last, partly filled round of workgroups (a launch of 3078
// workgroups is 1.7 rounds of the 1792 resident ones).  Workgroups below job[0].grid belong to the first job; everything a workgroup
// reads from its job is wave-uniform (scalar loads from the kernel-argument segment at a selected offset).
struct Floor1Job {
    Floor1Setup st;
    const uint32_t *yv;
    const uint32_t *line_offs;
    size_t count;
    uint32_t n;
    int n_posts, multiplier;
    unsigned grid;
};
struct Floor1Jobs {
    Floor1Job job[2];
};
template <int NMAX>
__global__ __launch_bounds__(64 * kF1Waves) void vorbis_floor1_pair_kernel(Floor1Jobs jobs, uint8_t *plane, const float *__restrict__ db) {
    const unsigned g0 = jobs.job[0].grid;
    const bool second = blockIdx.x >= g0;
    const Floor1Job &j = jobs.job[second ? 1 : 0];
    floor1_workgroup<2, NMAX>(j.st, j.n_posts, j.multiplier, j.yv, j.n, reinterpret_cast<float *>(plane), db, j.count, nullptr, j.line_offs,
                              second ? blockIdx.x - g0 : blockIdx.x);
}

int ilog2(int v) {
    int l = 0;
    while ((1 << l) < v) ++l;
    return l;
}

}  // namespace

// SYM_VORBIS_WAVE2 (build knob): 1 = block-size pairs other than 256 / 2048 with bs1 <= 2048 run vorbis_synth_wave2_kernel, 0 = the
// LDS-staged generic kernel as before (kept for the A/B and for 4096 / 8192-sample blocks).
#ifndef SYM_VORBIS_WAVE2
#define SYM_VORBIS_WAVE2 1
#endif

int launch_vorbis(symaccel_ctx *ctx, int bs0_exp, int bs1_exp, const float *d_spectra, const float *d_residue,
                  size_t spec_stride,
                  const uint8_t *d_block_flag, const int32_t *d_prev_in, int32_t *d_prev_out,
                  const float *d_overlap_in, float *d_overlap_out, float *d_pcm, size_t pcm_stride, size_t n_chains,
                  size_t blocks_per_chain, void *d_offsets, const uint8_t *d_floor_y) {
    if (blocks_per_chain > 0x3fffffffu) return SYMACCEL_ERR_INVALID_ARG;
    // d_floor_y: the floor curve as dB-table indices, one byte per line in the spectrum's packed layout (symaccel_vorbis_floor1_y_device);
    // d_spectra is then the RESIDUE.  The kernels take the plane through their `residue` argument (FUSED = 2).
    const int floor_mode = d_floor_y ? 2 : (d_residue ? 1 : 0);
    if (d_floor_y) d_residue = reinterpret_cast<const float *>(d_floor_y);
    const ImdctPlan *ps = nullptr, *pl = nullptr;
    SYM_TRY(get_imdct_plan(ctx, (1 << bs0_exp) >> 1, 1.0, &ps));  // vorbis/lib.rs:123
    SYM_TRY(get_imdct_plan(ctx, (1 << bs1_exp) >> 1, 1.0, &pl));  // vorbis/lib.rs:124
    const float *ws = nullptr, *wl = nullptr;
    SYM_TRY(get_vorbis_window(ctx, 1 << bs0_exp, &ws));
    SYM_TRY(get_vorbis_window(ctx, 1 << bs1_exp, &wl));
    const unsigned nb = (unsigned)blocks_per_chain;
    const bool wave_path = bs0_exp == 8 && bs1_exp == 11;
    // resident items per CU: eight wavefronts of the wavefront kernel; of the generic one, eight one-wavefront workgroups
    // (226 VGPRs: two per SIMD) or two 256-thread workgroups (66 KiB of LDS each)
    // every other pair with long blocks of up to 2048 samples: the multi-transform wavefront kernel (vorbis_wave2.hip; 16-byte
    // accesses: the strides must keep every chain 16-byte aligned), unless the build knob keeps the LDS-staged generic kernel
    const bool wave2_path = !wave_path && spec_stride % 4 == 0 && pcm_stride % 4 == 0 &&
                            ((uintptr_t)d_spectra | (uintptr_t)d_residue | (uintptr_t)d_pcm | (uintptr_t)d_overlap_in | (uintptr_t)d_overlap_out) % 16 == 0 &&
                            SYM_VORBIS_WAVE2;
    // long blocks of 8192 samples: the workgroup-cooperative kernel (vorbis_wg.hip), two workgroups resident per CU.  (SYM_VORBIS_WG 2
    // sends the 4096-sample pairs there as well: measured slower than the one-wavefront-per-block form, 2.0 against 2.2 TB/s -- only
    // two of its four wavefronts have a sub-transform to do.  The 4096 / 8192 pair runs BOTH cooperative block routines in one
    // instantiation, <FUSED, 13, 2>: 142 VGPRs, no scratch, three workgroups per CU.)
    const bool wg_path = wave2_path && ((bs1_exp == 13 && SYM_VORBIS_WG) || (bs1_exp == 12 && SYM_VORBIS_WG == 2));
    const unsigned seg = choose_segment(ctx, n_chains, nb, wg_path ? (SYM_VORBIS_WG_SHARED ? 3 : 2) : (wave2_path && bs1_exp <= 10 ? 12 : (wave2_path && bs1_exp == 13 ? 6 : ((wave_path || wave2_path) ? 8 : (bs1_exp > 11 ? 2 : 8)))), 1, 1, 1);
    const size_t segs = (nb + seg - 1) / seg;
    const size_t grid = n_chains * segs;
    if (grid > 0x7fffffffu) return SYMACCEL_ERR_INVALID_ARG;
    if (wave_path) {
        // the 256 / 2048 pair: wavefront-per-chain-segment kernel with register-resident overlap (vorbis_wave.hip);
        // it derives the packed offsets itself
        return launch_vorbis_wave(ctx, (const cpx *)ps->d_twiddle, (const cpx *)pl->d_twiddle, ws, wl, d_spectra,
                                  d_residue, spec_stride, d_block_flag, d_prev_in, d_prev_out, d_overlap_in, d_overlap_out, d_pcm,
                                  pcm_stride, n_chains, nb, seg, floor_mode);
    }
    // (both derive the packed offsets from the flags themselves, vorbis_offsets.h)
    if (wg_path)
        return launch_vorbis_wg(ctx, bs0_exp, bs1_exp, (const cpx *)ps->d_twiddle, (const cpx *)pl->d_twiddle, ws, wl, d_spectra, d_residue,
                                spec_stride, d_block_flag, d_prev_in, d_prev_out, d_overlap_in, d_overlap_out, d_pcm, pcm_stride, n_chains, nb, seg, floor_mode);
    if (wave2_path)
        return launch_vorbis_wave2(ctx, bs0_exp, bs1_exp, (const cpx *)ps->d_twiddle, (const cpx *)pl->d_twiddle, ws, wl, d_spectra, d_residue,
                                   spec_stride, d_block_flag, d_prev_in, d_prev_out, d_overlap_in, d_overlap_out, d_pcm, pcm_stride, n_chains, nb, seg, floor_mode);
    // the byte plane needs the 16-byte-aligned kernels above
    if (floor_mode == 2) return SYMACCEL_ERR_UNSUPPORTED;
    // the LDS-staged generic kernel (unaligned strides / pointers): a scan kernel turns the flag sequence into packed offsets (room for
    // them comes from the ABI wrapper: n_chains * (blocks_per_chain + 1) * 2 words)
    uint32_t *offs = (uint32_t *)(((uintptr_t)d_offsets + 255) & ~(uintptr_t)255);
    hipLaunchKernelGGL(vorbis_offsets_kernel, dim3((unsigned)n_chains), dim3(256), 0, ctx->stream, d_block_flag,
                       d_prev_in, offs, nb, 1 << bs0_exp, 1 << bs1_exp);
    SYM_GPU(ctx, hipGetLastError());
    if (bs1_exp <= 11) {
        hipLaunchKernelGGL(vorbis_synth_kernel<2048>, dim3((unsigned)grid), dim3(vorbis_threads<2048>()), 0, ctx->stream, ctx->dev,
                           bs0_exp, bs1_exp, (const cpx *)ps->d_twiddle, (const cpx *)pl->d_twiddle, ws, wl, d_spectra,
                           d_residue, spec_stride, d_block_flag, d_prev_in, d_prev_out, d_overlap_in, d_overlap_out, d_pcm,
                           pcm_stride, (const uint32_t *)offs, nb, seg, (unsigned)segs);
    } else {
        hipLaunchKernelGGL(vorbis_synth_kernel<8192>, dim3((unsigned)grid), dim3(vorbis_threads<8192>()), 0, ctx->stream, ctx->dev,
                           bs0_exp, bs1_exp, (const cpx *)ps->d_twiddle, (const cpx *)pl->d_twiddle, ws, wl, d_spectra,
                           d_residue, spec_stride, d_block_flag, d_prev_in, d_prev_out, d_overlap_in, d_overlap_out, d_pcm,
                           pcm_stride, (const uint32_t *)offs, nb, seg, (unsigned)segs);
    }
    SYM_GPU(ctx, hipGetLastError());
    return SYMACCEL_OK;
}

int launch_vorbis_prepare(symaccel_ctx *ctx, float *d_residue, size_t spec_stride, unsigned channels_per_stream, size_t n_streams,
                          size_t blocks, const uint32_t *d_block_off, const uint8_t *d_steps, const uint32_t *d_step_first,
                          const uint8_t *d_kill) {
    const size_t grid = n_streams * blocks;
    if (grid == 0) return SYMACCEL_OK;
    if (grid > 0x7fffffffu || blocks > 0xffffffffu) return SYMACCEL_ERR_INVALID_ARG;
    hipLaunchKernelGGL(vorbis_prepare_kernel, dim3((unsigned)grid), dim3(256), 0, ctx->stream, d_residue, spec_stride, channels_per_stream,
                       (unsigned)blocks, d_block_off, d_steps, d_step_first, d_kill);
    SYM_GPU(ctx, hipGetLastError());
    return SYMACCEL_OK;
}

int launch_vorbis_coupling(symaccel_ctx *ctx, float *d_mag, float *d_ang, size_t n) {
    const unsigned grid = (unsigned)((n + 255) / 256 < 4096 ? (n + 255) / 256 : 4096);
    hipLaunchKernelGGL(vorbis_coupling_kernel, dim3(grid), dim3(256), 0, ctx->stream, d_mag, d_ang, n);
    SYM_GPU(ctx, hipGetLastError());
    return SYMACCEL_OK;
}

int launch_vorbis_dot(symaccel_ctx *ctx, float *d_floor, const float *d_residue, size_t total) {
    const unsigned grid = (unsigned)((total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192);
    hipLaunchKernelGGL(vorbis_dot_kernel, dim3(grid), dim3(256), 0, ctx->stream, d_floor, d_residue, total);
    SYM_GPU(ctx, hipGetLastError());
    return SYMACCEL_OK;
}

int launch_vorbis_deinterleave(symaccel_ctx *ctx, const float *d_type2, float *d_planar, int n_ch, size_t n2,
                               size_t count) {
    const size_t total = n2 * (size_t)n_ch * count;
    const unsigned grid = (unsigned)((total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192);
    hipLaunchKernelGGL(vorbis_deinterleave_kernel, dim3(grid), dim3(256), 0, ctx->stream, d_type2, d_planar, n_ch, n2,
                       count);
    SYM_GPU(ctx, hipGetLastError());
    return SYMACCEL_OK;
}

static void floor1_derive(Floor1Setup &st, const uint32_t *h_setup, int n_posts) {
    for (int k = 0; k < n_posts; ++k) {
        const uint32_t i = h_setup[195 + k] & 255u;
        st.ord[k] = i | (h_setup[i] & 0xffffu) << 16;
    }
    for (int i = 2; i < n_posts; ++i) {
        const uint32_t lo = h_setup[65 + i] & 255u, hi = h_setup[130 + i] & 255u;
        const int adx = (int)h_setup[hi] - (int)h_setup[lo];  // > 0: the wrapper checked that the x values are distinct
        const float fadx = (float)(adx > 0 ? adx : 1);
        const bool wide = adx > 4096;  // tests/cpp/floor1_division_check.c proves the closed form up to here
        st.nb[i] = lo | hi << 8 | (wide ? 0x10000u : 0u);
        st.ratio[i] = (float)((int)h_setup[i] - (int)h_setup[lo]) / fadx;
        st.half[i] = 0.5f / fadx;
        st.wide[i] = ((h_setup[i] - h_setup[lo]) & 0xffffu) | (uint32_t)(adx > 0 ? adx : 1) << 16;  // (x[i] - x[lo]) | adx << 16
    }
}

int launch_vorbis_floor1(symaccel_ctx *ctx, const uint32_t *h_setup, int n_posts, int multiplier, const uint32_t *d_y,
                         uint32_t n, float *d_floor, size_t count, const float *d_residue, uint8_t *d_floor_y,
                         const uint32_t *d_line_offs) {
    const size_t grid = (count + kF1B - 1) / kF1B;
    if (grid > 0x7fffffffu) return SYMACCEL_ERR_INVALID_ARG;
    Floor1Setup st{};  // passed by value: the kernel reads it with scalar loads (wave-uniform indices)
    floor1_derive(st, h_setup, n_posts);
    // instantiated per block class: the segment-start map is n bytes of LDS, and LDS is what bounds the resident wavefronts
#define SYM_F1_LAUNCH(MODE, NMAX)                                                                                                    \
    hipLaunchKernelGGL((vorbis_floor1_kernel<MODE, NMAX>), dim3((unsigned)grid), dim3(64 * kF1Waves), 0, ctx->stream, st, n_posts, multiplier, d_y, \
                       n, d_floor_y ? reinterpret_cast<float *>(d_floor_y) : d_floor, ctx->dev.vorbis_floor1_db, count, d_residue, d_line_offs)
    if (d_floor_y) {  // the curve as table indices, one byte per line
        if (n <= 1024) SYM_F1_LAUNCH(2, 1024); else SYM_F1_LAUNCH(2, 4096);
    } else if (d_residue) {
        if (n <= 1024) SYM_F1_LAUNCH(1, 1024); else SYM_F1_LAUNCH(1, 4096);
    } else {
        if (n <= 1024) SYM_F1_LAUNCH(0, 1024); else SYM_F1_LAUNCH(0, 4096);
    }
#undef SYM_F1_LAUNCH
    SYM_GPU(ctx, hipGetLastError());
    return SYMACCEL_OK;
}

int launch_vorbis_floor1_pair(symaccel_ctx *ctx, const uint32_t *const h_setup[2], const int n_posts[2], const int multiplier[2],
                              const uint32_t *const d_y[2], const uint32_t n[2], const size_t count[2], const uint32_t *const d_line_offs[2],
                              uint8_t *d_floor_y) {
    Floor1Jobs jobs{};
    size_t grid = 0;
    for (int k = 0; k < 2; ++k) {
        Floor1Job &j = jobs.job[k];
        floor1_derive(j.st, h_setup[k], n_posts[k]);
        j.yv = d_y[k];
        j.line_offs = d_line_offs[k];
        j.count = count[k];
        j.n = n[k];
        j.n_posts = n_posts[k];
        j.multiplier = multiplier[k];
        const size_t g = (count[k] + kF1B - 1) / kF1B;
        if (g > 0x3fffffffu) return SYMACCEL_ERR_INVALID_ARG;
        j.grid = (unsigned)g;
        grid += g;
    }
    if (n[0] <= 1024 && n[1] <= 1024)
        hipLaunchKernelGGL((vorbis_floor1_pair_kernel<1024>), dim3((unsigned)grid), dim3(64 * kF1Waves), 0, ctx->stream, jobs, d_floor_y, ctx->dev.vorbis_floor1_db);
    else
        hipLaunchKernelGGL((vorbis_floor1_pair_kernel<4096>), dim3((unsigned)grid), dim3(64 * kF1Waves), 0, ctx->stream, jobs, d_floor_y, ctx->dev.vorbis_floor1_db);
    SYM_GPU(ctx, hipGetLastError());
    return SYMACCEL_OK;
}

}  // namespace symaccel
