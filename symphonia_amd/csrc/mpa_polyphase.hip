// MPEG-1/2 Layer I and Layer II synthesis: the 32-band polyphase filterbank alone, synthesis::synthesis with
// n_frames = 12 (Layer I, layer1/mod.rs:193) or 36 (Layer II, layer2/mod.rs:383):
// symphonia-bundle-mp3/src/synthesis.rs:158-336.  (Layer III fuses it behind the hybrid synthesis: mp3.hip.)
//
// Same mapping as the Layer III kernel's tail: a wavefront carries two chains (32 lanes each); lane = sub-band for
// the load (the sub-band-major input `in[n_frames * i + b]` gives every lane its own contiguous run), lane = time
// slot for the dct32s (LDS transpose, row stride 36, conflict-free b128), lane = output sample i for the 512-tap
// window with the V entries of the previous 16 slots in registers.  One packet (12 or 36 slots) per round; segments
// start with ceil(16 / n_frames) halo packets that only rebuild the V history.
#include "mp3_common.h"

namespace symaccel {

namespace {

template <int NF>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(2, 2))) void mpa_polyphase_kernel(
    DevTables tb, const float *__restrict__ in, const float *__restrict__ vvec_in, const int32_t *__restrict__ vfront_in,
    float *__restrict__ vvec_out, int32_t *__restrict__ vfront_out, float *__restrict__ pcm, unsigned n_chains,
    unsigned packets_per_chain, unsigned seg_len, unsigned segs_per_chain) {
    constexpr int kPacket = 32 * NF;                       // floats per channel-packet
    constexpr int kHalo = (kHistOld + NF - 1) / NF;        // packets that rebuild the 16-slot history
    constexpr int kTile = 2 * kPacket, kS = 2 * NF * kSStride;
    __shared__ __attribute__((aligned(16))) float lds[(kS > kTile ? kS : kTile) + kDwFloats];
    const int lane = (int)threadIdx.x, half = lane >> 5, hl = lane & 31;
    float *tile = lds + half * kPacket;          // the packet as stored: [sub-band][slot]
    float *S = lds + half * (NF * kSStride);     // S[slot][sub-band] (aliases the tiles; ordered by wave_sync)
    float *dwt = lds + (kS > kTile ? kS : kTile);

    const unsigned item = blockIdx.x * 2u + (unsigned)half;
    const bool live = item < n_chains * segs_per_chain;
    const unsigned chain = live ? item / segs_per_chain : 0, seg = live ? item % segs_per_chain : 0;
    const unsigned p_begin = seg * seg_len;
    const unsigned p_end = live ? min(p_begin + seg_len, packets_per_chain) : p_begin;
    const size_t chain_base = (size_t)chain * packets_per_chain;

    if (half == 0) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            dwt[hl * kDwStride + j] = tb.mp3_consts[MP3C_SYNTH_D + 64 * j + hl];
            dwt[hl * kDwStride + 8 + j] = tb.mp3_consts[MP3C_SYNTH_D + 64 * j + 32 + hl];
        }
    }
    const VMap vm = vmap(hl);

    // oA[16 + r] = V_r[i], oB[16 + r] = V_r[32 + i] for the previous packets' time slots r < 0
    float oA[kHistOld], oB[kHistOld];
    const bool first_seg = p_begin == 0;
#pragma unroll
    for (int r = 0; r < kHistOld; ++r) oA[r] = oB[r] = 0.0f;
    if (live && first_seg) {
        const int v_front = vfront_in[chain] & 15;
        const float *vv = vvec_in + (size_t)chain * 1024;
#pragma unroll
        for (int m = 1; m <= kHistOld; ++m) {  // slot -m sits in FIFO row (v_front + m) & 15 (synthesis.rs:335)
            const float *row = vv + 64 * ((v_front + m) & 15);
            oA[kHistOld - m] = row[hl];
            oB[kHistOld - m] = row[32 + hl];
        }
    }

    const long p_first = first_seg ? 0 : ((long)p_begin > kHalo ? (long)p_begin - kHalo : 0);
    const long p_stop = (long)p_end;
    long rounds = live ? p_stop - p_first : 0;  // both halves run the same number of rounds
    {
        const long other = __shfl((int)rounds, lane ^ 32);
        rounds = rounds > other ? rounds : other;
    }

    for (long r = 0; r < rounds; ++r) {
        const long p = p_first + r;
        const bool active = live && p < p_stop;
        const bool emit = active && p >= (long)p_begin;
        // ---- packet -> LDS (16 B per lane, coalesced), then lane sb takes its NF consecutive slots
        if (active) {
            const float4 *src = reinterpret_cast<const float4 *>(in + (chain_base + (size_t)p) * kPacket);
            float4 *t4 = reinterpret_cast<float4 *>(tile);
#pragma unroll
            for (int q = 0; q < kPacket / 128; ++q) t4[hl + 32 * q] = src[hl + 32 * q];
        }
        wave_sync();
        float y[NF];
        {
            const float4 *t4 = reinterpret_cast<const float4 *>(tile + NF * hl);  // NF * 4 B lane stride: conflict-free b128
#pragma unroll
            for (int k = 0; k < NF / 4; ++k) {
                const float4 v = t4[k];
                y[4 * k] = v.x; y[4 * k + 1] = v.y; y[4 * k + 2] = v.z; y[4 * k + 3] = v.w;
            }
        }
        wave_sync();  // every lane has read its run; the buffer becomes S[slot][sb]
#pragma unroll
        for (int b = 0; b < NF; ++b) S[b * kSStride + hl] = y[b];
        wave_sync();
        // ---- dct32 of every (chain, slot) row (synthesis.rs:165-245): lane = row, 2 * NF rows over 64 lanes
        for (int row_id = lane; row_id < 2 * NF; row_id += 64) {
            float d[32];
            float4 *row = reinterpret_cast<float4 *>(lds + (row_id / NF) * (NF * kSStride) + (row_id % NF) * kSStride);
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const float4 v = row[k];
                d[4 * k] = v.x; d[4 * k + 1] = v.y; d[4 * k + 2] = v.z; d[4 * k + 3] = v.w;
            }
            dct_lee<32>(d);
#pragma unroll
            for (int k = 0; k < 8; ++k) row[k] = make_float4(d[4 * k], d[4 * k + 1], d[4 * k + 2], d[4 * k + 3]);
        }
        wave_sync();
        // ---- windowing (synthesis.rs:309-324): 16 taps per sample, operands in registers
        float dw0[8], dw1[8];
        {
            const float4 *r4 = reinterpret_cast<const float4 *>(dwt + hl * kDwStride);
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                const float4 a = r4[k], c = r4[2 + k];
                dw0[4 * k] = a.x; dw0[4 * k + 1] = a.y; dw0[4 * k + 2] = a.z; dw0[4 * k + 3] = a.w;
                dw1[4 * k] = c.x; dw1[4 * k + 1] = c.y; dw1[4 * k + 2] = c.z; dw1[4 * k + 3] = c.w;
            }
        }
        float nA[NF], nB[NF];
#pragma unroll
        for (int b = 0; b < NF; ++b) {
            const float df = S[b * kSStride + vm.fidx];
            nA[b] = vm.fkind == 0 ? df : (vm.fkind == 1 ? -df : 0.0f);  // V[i]      (synthesis.rs:247-263)
            nB[b] = -S[b * kSStride + vm.sidx];                          // V[32 + i]
            float acc = 0.0f;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int ra = b - 2 * j, rb = b - 2 * j - 1;
                acc += (ra >= 0 ? nA[ra >= 0 ? ra : 0] : oA[ra < 0 ? kHistOld + ra : 0]) * dw0[j];
                acc += (rb >= 0 ? nB[rb >= 0 ? rb : 0] : oB[rb < 0 ? kHistOld + rb : 0]) * dw1[j];
            }
            if (emit) (pcm + (chain_base + (size_t)p) * kPacket)[32 * b + hl] = acc;
        }
        wave_sync();  // the window pass has read S; the next round's packet goes to the same LDS
        // ---- slide the history: the packet's last 16 slots become slots -16..-1 (NF = 12: four old ones stay)
        if (active) {
#pragma unroll
            for (int m = 0; m < kHistOld; ++m) {
                const int src = m + NF - kHistOld;  // slot index within this packet, negative = an older slot
                oA[m] = src >= 0 ? nA[src >= 0 ? src : 0] : oA[src < 0 ? kHistOld + src : 0];
                oB[m] = src >= 0 ? nB[src >= 0 ? src : 0] : oB[src < 0 ? kHistOld + src : 0];
            }
        }
    }

    if (live && p_end == packets_per_chain) {
        const int vf0 = vfront_in[chain] & 15;
        const int vf_final = (int)(((unsigned)vf0 + 15u * (unsigned)NF * packets_per_chain) & 15u);  // synthesis.rs:335
        float *vv = vvec_out + (size_t)chain * 1024;
#pragma unroll
        for (int m = 1; m <= kHistOld; ++m) {
            float *row = vv + 64 * ((vf_final + m) & 15);
            row[hl] = oA[kHistOld - m];
            row[32 + hl] = oB[kHistOld - m];
        }
        if (hl == 0) vfront_out[chain] = vf_final;
    }
}

}  // namespace

int launch_mpa_polyphase(symaccel_ctx *ctx, int n_frames, const float *d_in, const float *d_vvec_in,
                         const int32_t *d_vfront_in, float *d_vvec_out, int32_t *d_vfront_out, float *d_pcm, size_t n_chains,
                         size_t packets_per_chain) {
    if (packets_per_chain > 0x3fffffffu || n_chains > 0x3fffffffu) return SYMACCEL_ERR_INVALID_ARG;
    const unsigned halo = n_frames == 12 ? 2u : 1u;
    const unsigned seg = choose_segment(ctx, n_chains, packets_per_chain, 8, 2, halo, halo);
    const size_t segs = (packets_per_chain + seg - 1) / seg;
    const size_t items = n_chains * segs;
    const size_t grid = (items + 1) / 2;
    if (grid > 0x7fffffffu) return SYMACCEL_ERR_INVALID_ARG;
    if (n_frames == 12) {
        hipLaunchKernelGGL(mpa_polyphase_kernel<12>, dim3((unsigned)grid), dim3(64), 0, ctx->stream, ctx->dev, d_in, d_vvec_in,
                           d_vfront_in, d_vvec_out, d_vfront_out, d_pcm, (unsigned)n_chains, (unsigned)packets_per_chain, seg,
                           (unsigned)segs);
    } else if (n_frames == 36) {
        hipLaunchKernelGGL(mpa_polyphase_kernel<36>, dim3((unsigned)grid), dim3(64), 0, ctx->stream, ctx->dev, d_in, d_vvec_in,
                           d_vfront_in, d_vvec_out, d_vfront_out, d_pcm, (unsigned)n_chains, (unsigned)packets_per_chain, seg,
                           (unsigned)segs);
    } else {
        return SYMACCEL_ERR_UNSUPPORTED;
    }
    SYM_GPU(ctx, hipGetLastError());
    return SYMACCEL_OK;
}

}  // namespace symaccel
