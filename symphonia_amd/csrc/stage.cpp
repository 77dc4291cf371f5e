// Host-pointer entry points: pinned, chunked, double-buffered staging.
//
// The device entry points assume the batch is resident in HBM.  A decoder adapter (AudioDecoder::decode gets a packet
// in host memory and must hand back a buffer in host memory) pays PCIe both ways: for a config-2 AAC batch that is 1 GiB
// each way against a quarter of a millisecond of kernel time.  The `*_pipelined` entry points -- which the plain
// host-pointer entry points of ctx.cpp route to for large batches -- cut the batch into chunks along the frame axis and
// keep three queues busy at once:
//
//      copy-in stream :  H2D(k+1) ..........
//      compute stream :  kernel(k) ......        (state carried chunk to chunk in two ping-pong device buffers)
//      copy-out stream:  D2H(k-1) ..........
//
// with two device buffer sets, so the wall time tends to max(H2D, D2H, kernel) instead of their sum.  The overlap needs
// page-locked host memory: allocate the batch buffers with symaccel_host_alloc() or pin existing ones with
// symaccel_host_register(); pageable memory still works (the runtime stages it) but serialises.
#include <algorithm>
#include <cstring>
#include <vector>

#include "symaccel_internal.h"

using namespace symaccel;

namespace {

struct Pipe {
    symaccel_ctx *ctx;
    hipStream_t s_in = nullptr, s_out = nullptr;
    hipEvent_t ev_in[2] = {nullptr, nullptr}, ev_k[2] = {nullptr, nullptr}, ev_out[2] = {nullptr, nullptr};
    std::vector<std::pair<void **, size_t>> wanted;
    bool queued = false;
    explicit Pipe(symaccel_ctx *c) : ctx(c) {}
    // An early (error) return must not leave copies to or from the caller's host buffers in flight: whatever was queued is
    // waited for before the call returns.  The streams, events and the arena belong to the context and stay.
    ~Pipe() {
        if (!queued) return;
        if (s_in) (void)hipStreamSynchronize(s_in);
        if (ctx->stream) (void)hipStreamSynchronize(ctx->stream);
        if (s_out) (void)hipStreamSynchronize(s_out);
    }
    int init() {
        if (!ctx->stage_in) SYM_GPU(ctx, hipStreamCreate(&ctx->stage_in));
        if (!ctx->stage_out) SYM_GPU(ctx, hipStreamCreate(&ctx->stage_out));
        for (hipEvent_t &e : ctx->stage_events)
            if (!e) SYM_GPU(ctx, hipEventCreateWithFlags(&e, hipEventDisableTiming));
        s_in = ctx->stage_in;
        s_out = ctx->stage_out;
        for (int b = 0; b < 2; ++b) {
            ev_in[b] = ctx->stage_events[b];
            ev_k[b] = ctx->stage_events[2 + b];
            ev_out[b] = ctx->stage_events[4 + b];
        }
        return SYMACCEL_OK;
    }
    // chunk buffers: requested one by one, carved from the context's arena by commit() (which grows it if needed; the
    // previous call drained everything, so a smaller arena can be freed at once)
    int alloc(void **p, size_t bytes) {
        *p = nullptr;
        wanted.emplace_back(p, bytes);
        return SYMACCEL_OK;
    }
    int commit() {
        size_t total = 0;
        for (auto &w : wanted) total += (w.second + 255) & ~(size_t)255;
        if (total > ctx->stage_arena_bytes) {
            if (ctx->stage_arena) {
                SYM_GPU(ctx, hipFree(ctx->stage_arena));
                ctx->stage_arena = nullptr;
                ctx->stage_arena_bytes = 0;
            }
            void *a = nullptr;
            SYM_TRY(ctx_alloc(ctx, &a, total, false));
            ctx->stage_arena = a;
            ctx->stage_arena_bytes = total;
        }
        size_t off = 0;
        for (auto &w : wanted) {
            *w.first = static_cast<char *>(ctx->stage_arena) + off;
            off += (w.second + 255) & ~(size_t)255;
        }
        queued = true;  // from here on work may be in flight
        return SYMACCEL_OK;
    }
    // everything queued so far, on all three streams
    int drain() {
        SYM_GPU(ctx, hipStreamSynchronize(s_in));
        SYM_GPU(ctx, hipStreamSynchronize(ctx->stream));
        SYM_GPU(ctx, hipStreamSynchronize(s_out));
        queued = false;
        return SYMACCEL_OK;
    }
};

// rows x width bytes between a [rows][src_pitch] and a [rows][dst_pitch] layout
int copy_rows(symaccel_ctx *ctx, void *dst, size_t dpitch, const void *src, size_t spitch, size_t width, size_t rows, hipMemcpyKind kind,
              hipStream_t s) {
    if (width == 0 || rows == 0) return SYMACCEL_OK;
    if (dpitch == width && spitch == width) {
        SYM_GPU(ctx, hipMemcpyAsync(dst, src, width * rows, kind, s));
    } else {
        SYM_GPU(ctx, hipMemcpy2DAsync(dst, dpitch, src, spitch, width, rows, kind, s));
    }
    return SYMACCEL_OK;
}

size_t pick_chunk(size_t units_per_chain, size_t bytes_per_unit_all_chains, size_t requested) {
    if (requested > 0) return std::min(requested, units_per_chain);
    // ~32 MiB of input per chunk: large enough to fill the GPU (hundreds of wavefront segments) and to amortise the
    // per-copy latency, small enough that the first kernel starts early and the last copy-out finishes soon after the last kernel
    size_t c = ((size_t)32 << 20) / std::max<size_t>(1, bytes_per_unit_all_chains);
    c = std::max<size_t>(c, 8);
    return std::min(c, units_per_chain);
}

}  // namespace

extern "C" {

int symaccel_host_alloc(size_t bytes, void **out) {
    if (!out) return SYMACCEL_ERR_INVALID_ARG;
    *out = nullptr;
    if (bytes == 0) bytes = 16;
    const hipError_t e = hipHostMalloc(out, bytes, hipHostMallocDefault);
    if (e != hipSuccess) return e == hipErrorOutOfMemory ? SYMACCEL_ERR_OOM : SYMACCEL_ERR_DEVICE;
    return SYMACCEL_OK;
}

int symaccel_host_free(void *p) {
    if (!p) return SYMACCEL_OK;
    return hipHostFree(p) == hipSuccess ? SYMACCEL_OK : SYMACCEL_ERR_DEVICE;
}

int symaccel_host_register(void *p, size_t bytes) {
    if (!p || bytes == 0) return SYMACCEL_ERR_INVALID_ARG;
    return hipHostRegister(p, bytes, hipHostRegisterDefault) == hipSuccess ? SYMACCEL_OK : SYMACCEL_ERR_DEVICE;
}

int symaccel_host_unregister(void *p) {
    if (!p) return SYMACCEL_ERR_INVALID_ARG;
    return hipHostUnregister(p) == hipSuccess ? SYMACCEL_OK : SYMACCEL_ERR_DEVICE;
}

int symaccel_aac_synth_pipelined(symaccel_ctx *ctx, const float *h_coeffs, const uint8_t *h_side, float *h_delay_io, float *h_pcm,
                                 size_t n_chains, size_t frames_per_chain, size_t chunk_frames) {
    if (!ctx) return SYMACCEL_ERR_INVALID_ARG;
    if (n_chains == 0 || frames_per_chain == 0) return SYMACCEL_OK;
    if (!h_coeffs || !h_side || !h_delay_io || !h_pcm) return SYMACCEL_ERR_INVALID_ARG;
    DeviceGuard dev(ctx);
    if (!dev.ok()) return dev.status();
    const size_t cf = pick_chunk(frames_per_chain, n_chains * 4096, chunk_frames);
    Pipe pp(ctx);
    SYM_TRY(pp.init());
    float *d_in[2], *d_out[2], *d_state[2];
    uint8_t *d_side[2];
    for (int b = 0; b < 2; ++b) {
        SYM_TRY(pp.alloc((void **)&d_in[b], n_chains * cf * 4096));
        SYM_TRY(pp.alloc((void **)&d_out[b], n_chains * cf * 4096));
        SYM_TRY(pp.alloc((void **)&d_side[b], n_chains * cf));
        SYM_TRY(pp.alloc((void **)&d_state[b], n_chains * 4096));
    }
    SYM_TRY(pp.commit());
    SYM_GPU(ctx, hipMemcpyAsync(d_state[0], h_delay_io, n_chains * 4096, hipMemcpyHostToDevice, ctx->stream));
    size_t k = 0;
    for (size_t t0 = 0; t0 < frames_per_chain; t0 += cf, ++k) {
        const size_t nf = std::min(cf, frames_per_chain - t0);
        const int b = (int)(k & 1);
        if (k >= 2) {  // buffer set b is free once chunk k-2's kernel has read its input and its PCM has left
            SYM_GPU(ctx, hipStreamWaitEvent(pp.s_in, pp.ev_k[b], 0));
            SYM_GPU(ctx, hipStreamWaitEvent(ctx->stream, pp.ev_out[b], 0));
        }
        SYM_TRY(copy_rows(ctx, d_in[b], nf * 4096, h_coeffs + t0 * 1024, frames_per_chain * 4096, nf * 4096, n_chains, hipMemcpyHostToDevice,
                          pp.s_in));
        SYM_TRY(copy_rows(ctx, d_side[b], nf, h_side + t0, frames_per_chain, nf, n_chains, hipMemcpyHostToDevice, pp.s_in));
        SYM_GPU(ctx, hipEventRecord(pp.ev_in[b], pp.s_in));
        SYM_GPU(ctx, hipStreamWaitEvent(ctx->stream, pp.ev_in[b], 0));
        SYM_TRY(launch_aac(ctx, d_in[b], d_side[b], d_state[k & 1], d_state[(k + 1) & 1], d_out[b], n_chains, nf));
        SYM_GPU(ctx, hipEventRecord(pp.ev_k[b], ctx->stream));
        SYM_GPU(ctx, hipStreamWaitEvent(pp.s_out, pp.ev_k[b], 0));
        SYM_TRY(copy_rows(ctx, h_pcm + t0 * 1024, frames_per_chain * 4096, d_out[b], nf * 4096, nf * 4096, n_chains, hipMemcpyDeviceToHost,
                          pp.s_out));
        SYM_GPU(ctx, hipEventRecord(pp.ev_out[b], pp.s_out));
    }
    SYM_GPU(ctx, hipMemcpyAsync(h_delay_io, d_state[k & 1], n_chains * 4096, hipMemcpyDeviceToHost, ctx->stream));
    return pp.drain();
}

int symaccel_aac_decode_pipelined(symaccel_ctx *ctx, const float *h_coeffs, const uint8_t *h_side, const int32_t *h_pair_chains,
                                  const symaccel_aac_js_frame *h_js_desc, size_t n_pairs, const uint16_t *swb_long, int n_swb_long,
                                  const uint16_t *swb_short, int n_swb_short, const symaccel_aac_tns_filter *h_tns, size_t n_tns,
                                  float *h_delay_io, float *h_pcm, size_t n_chains, size_t frames_per_chain, size_t chunk_frames) {
    if (!ctx) return SYMACCEL_ERR_INVALID_ARG;
    AacBandMaps maps;
    if (!aac_band_maps(swb_long, n_swb_long, swb_short, n_swb_short, &maps)) return SYMACCEL_ERR_INVALID_ARG;
    if (n_chains == 0 || frames_per_chain == 0) return SYMACCEL_OK;
    if (!h_coeffs || !h_side || !h_delay_io || !h_pcm) return SYMACCEL_ERR_INVALID_ARG;
    if ((n_pairs && (!h_pair_chains || !h_js_desc)) || (n_tns && !h_tns) || 2 * n_pairs > n_chains) return SYMACCEL_ERR_INVALID_ARG;
    if (n_chains * frames_per_chain > 0xffffffffu) return SYMACCEL_ERR_INVALID_ARG;
    // pair_chains is in host memory: every chain at most once, inside the batch; pair_of[chain] = its pair
    std::vector<int32_t> pair_of(n_chains, -1);
    for (size_t p = 0; p < 2 * n_pairs; ++p) {
        const int32_t c = h_pair_chains[p];
        if (c < 0 || (size_t)c >= n_chains || pair_of[(size_t)c] >= 0) return SYMACCEL_ERR_INVALID_ARG;
        pair_of[(size_t)c] = (int32_t)(p / 2);
    }
    DeviceGuard dev(ctx);
    if (!dev.ok()) return dev.status();
    const size_t cf = pick_chunk(frames_per_chain, n_chains * 4096, chunk_frames);
    const size_t n_chunks = (frames_per_chain + cf - 1) / cf;
    // TNS runs between joint stereo and the transform (ics/mod.rs:452-468), and it is a serial recurrence along the spectrum: it
    // stays a pass of its own (one lane per filter, csrc/aac_tools.hip).  The filters are sorted into the chunks of the pipeline
    // (frame indices re-based to the chunk's [chain][frame] layout); a channel-pair frame with a filter in either channel has its
    // joint stereo decoded in place IN FRONT of the filters (a list pass over those frames only) and is marked "nothing coded"
    // for the fused pair walk.  Frames without TNS -- the bulk of a stream -- are read once, by the walk.
    std::vector<std::vector<symaccel_aac_tns_filter>> tns(n_chunks);
    std::vector<std::vector<uint32_t>> tns_pf(n_chunks);
    for (size_t i = 0; i < n_tns; ++i) {
        symaccel_aac_tns_filter f = h_tns[i];
        if ((size_t)f.frame >= n_chains * frames_per_chain) continue;  // (what symaccel_aac_tns_device skips)
        const size_t chain = f.frame / frames_per_chain, t = f.frame % frames_per_chain, k = t / cf;
        const size_t nf = std::min(cf, frames_per_chain - k * cf);
        f.frame = (uint32_t)(chain * nf + (t - k * cf));
        tns[k].push_back(f);
        if (pair_of[chain] >= 0) tns_pf[k].push_back((uint32_t)((size_t)pair_of[chain] * nf + (t - k * cf)));
    }
    size_t max_tns = 0, max_pf = 0;
    for (size_t k = 0; k < n_chunks; ++k) {
        std::sort(tns_pf[k].begin(), tns_pf[k].end());
        tns_pf[k].erase(std::unique(tns_pf[k].begin(), tns_pf[k].end()), tns_pf[k].end());
        max_tns = std::max(max_tns, tns[k].size());
        max_pf = std::max(max_pf, tns_pf[k].size());
    }
    Pipe pp(ctx);
    SYM_TRY(pp.init());
    float *d_in[2], *d_out[2], *d_state[2];
    uint8_t *d_side[2];
    symaccel_aac_js_frame *d_js[2];
    symaccel_aac_tns_filter *d_tns[2];
    uint32_t *d_pf[2];
    int32_t *d_pairs;
    void *d_index;
    for (int b = 0; b < 2; ++b) {
        SYM_TRY(pp.alloc((void **)&d_in[b], n_chains * cf * 4096));
        SYM_TRY(pp.alloc((void **)&d_out[b], n_chains * cf * 4096));
        SYM_TRY(pp.alloc((void **)&d_side[b], n_chains * cf));
        SYM_TRY(pp.alloc((void **)&d_state[b], n_chains * 4096));
        SYM_TRY(pp.alloc((void **)&d_js[b], std::max<size_t>(1, n_pairs * cf) * sizeof(symaccel_aac_js_frame)));
        SYM_TRY(pp.alloc((void **)&d_tns[b], std::max<size_t>(1, max_tns) * sizeof(symaccel_aac_tns_filter)));
        SYM_TRY(pp.alloc((void **)&d_pf[b], std::max<size_t>(1, max_pf) * 4));
    }
    SYM_TRY(pp.alloc((void **)&d_pairs, std::max<size_t>(1, n_pairs) * 8));
    SYM_TRY(pp.alloc(&d_index, aac_js_scratch_bytes(n_chains, n_pairs, cf)));
    SYM_TRY(pp.commit());
    if (n_pairs) SYM_GPU(ctx, hipMemcpyAsync(d_pairs, h_pair_chains, n_pairs * 8, hipMemcpyHostToDevice, ctx->stream));
    SYM_GPU(ctx, hipMemcpyAsync(d_state[0], h_delay_io, n_chains * 4096, hipMemcpyHostToDevice, ctx->stream));
    size_t k = 0;
    for (size_t t0 = 0; t0 < frames_per_chain; t0 += cf, ++k) {
        const size_t nf = std::min(cf, frames_per_chain - t0);
        const int b = (int)(k & 1);
        if (k >= 2) {  // buffer set b is free once chunk k-2's kernels have read its input and its PCM has left
            SYM_GPU(ctx, hipStreamWaitEvent(pp.s_in, pp.ev_k[b], 0));
            SYM_GPU(ctx, hipStreamWaitEvent(ctx->stream, pp.ev_out[b], 0));
        }
        SYM_TRY(copy_rows(ctx, d_in[b], nf * 4096, h_coeffs + t0 * 1024, frames_per_chain * 4096, nf * 4096, n_chains, hipMemcpyHostToDevice,
                          pp.s_in));
        SYM_TRY(copy_rows(ctx, d_side[b], nf, h_side + t0, frames_per_chain, nf, n_chains, hipMemcpyHostToDevice, pp.s_in));
        if (n_pairs)
            SYM_TRY(copy_rows(ctx, d_js[b], nf * sizeof(symaccel_aac_js_frame), h_js_desc + t0, frames_per_chain * sizeof(symaccel_aac_js_frame),
                              nf * sizeof(symaccel_aac_js_frame), n_pairs, hipMemcpyHostToDevice, pp.s_in));
        if (!tns[k].empty())
            SYM_GPU(ctx, hipMemcpyAsync(d_tns[b], tns[k].data(), tns[k].size() * sizeof(symaccel_aac_tns_filter), hipMemcpyHostToDevice, pp.s_in));
        if (!tns_pf[k].empty())
            SYM_GPU(ctx, hipMemcpyAsync(d_pf[b], tns_pf[k].data(), tns_pf[k].size() * 4, hipMemcpyHostToDevice, pp.s_in));
        SYM_GPU(ctx, hipEventRecord(pp.ev_in[b], pp.s_in));
        SYM_GPU(ctx, hipStreamWaitEvent(ctx->stream, pp.ev_in[b], 0));
        if (!tns_pf[k].empty()) {  // cpe.rs:110-157 for the pair frames that carry TNS, in place; then they are plain frames for the walk
            SYM_TRY(launch_aac_joint_stereo(ctx, maps, d_in[b], nf, d_pairs, d_js[b], n_pairs, d_pf[b], tns_pf[k].size()));
            SYM_TRY(launch_aac_js_consume(ctx, d_js[b], d_pf[b], tns_pf[k].size(), n_pairs * nf));
        }
        if (!tns[k].empty()) SYM_TRY(launch_aac_tns(ctx, d_in[b], n_chains * nf, d_tns[b], tns[k].size()));  // tns.rs:180-195
        SYM_TRY(launch_aac(ctx, d_in[b], d_side[b], d_state[k & 1], d_state[(k + 1) & 1], d_out[b], n_chains, nf, n_pairs ? &maps : nullptr,
                           d_pairs, d_js[b], n_pairs, d_index));
        SYM_GPU(ctx, hipEventRecord(pp.ev_k[b], ctx->stream));
        SYM_GPU(ctx, hipStreamWaitEvent(pp.s_out, pp.ev_k[b], 0));
        SYM_TRY(copy_rows(ctx, h_pcm + t0 * 1024, frames_per_chain * 4096, d_out[b], nf * 4096, nf * 4096, n_chains, hipMemcpyDeviceToHost,
                          pp.s_out));
        SYM_GPU(ctx, hipEventRecord(pp.ev_out[b], pp.s_out));
    }
    SYM_GPU(ctx, hipMemcpyAsync(h_delay_io, d_state[k & 1], n_chains * 4096, hipMemcpyDeviceToHost, ctx->stream));
    return pp.drain();
}

int symaccel_mp3_synth_pipelined(symaccel_ctx *ctx, const float *h_xr, const symaccel_mp3_side *h_side, int sample_rate_idx,
                                 float *h_overlap_io, float *h_vvec_io, int32_t *h_vfront_io, float *h_pcm, size_t n_chains,
                                 size_t granules_per_chain, size_t chunk_granules) {
    if (!ctx || sample_rate_idx < 0 || sample_rate_idx > 8) return SYMACCEL_ERR_INVALID_ARG;
    if (n_chains == 0 || granules_per_chain == 0) return SYMACCEL_OK;
    if (!h_xr || !h_side || !h_overlap_io || !h_vvec_io || !h_vfront_io || !h_pcm) return SYMACCEL_ERR_INVALID_ARG;
    DeviceGuard dev(ctx);
    if (!dev.ok()) return dev.status();
    size_t cg = pick_chunk(granules_per_chain, n_chains * 2304, chunk_granules);
    if (cg < 2 && granules_per_chain >= 2) cg = 2;  // the kernel's two-granule halo wants segments of at least two
    Pipe pp(ctx);
    SYM_TRY(pp.init());
    float *d_in[2], *d_out[2], *d_ov[2], *d_vv[2];
    symaccel_mp3_side *d_side[2];
    int32_t *d_vf[2];
    for (int b = 0; b < 2; ++b) {
        SYM_TRY(pp.alloc((void **)&d_in[b], n_chains * cg * 2304));
        SYM_TRY(pp.alloc((void **)&d_out[b], n_chains * cg * 2304));
        SYM_TRY(pp.alloc((void **)&d_side[b], n_chains * cg * sizeof(symaccel_mp3_side)));
        SYM_TRY(pp.alloc((void **)&d_ov[b], n_chains * 2304));
        SYM_TRY(pp.alloc((void **)&d_vv[b], n_chains * 4096));
        SYM_TRY(pp.alloc((void **)&d_vf[b], n_chains * 4));
    }
    SYM_TRY(pp.commit());
    SYM_GPU(ctx, hipMemcpyAsync(d_ov[0], h_overlap_io, n_chains * 2304, hipMemcpyHostToDevice, ctx->stream));
    SYM_GPU(ctx, hipMemcpyAsync(d_vv[0], h_vvec_io, n_chains * 4096, hipMemcpyHostToDevice, ctx->stream));
    SYM_GPU(ctx, hipMemcpyAsync(d_vf[0], h_vfront_io, n_chains * 4, hipMemcpyHostToDevice, ctx->stream));
    size_t k = 0;
    for (size_t g0 = 0; g0 < granules_per_chain; g0 += cg, ++k) {
        const size_t ng = std::min(cg, granules_per_chain - g0);
        const int b = (int)(k & 1);
        if (k >= 2) {
            SYM_GPU(ctx, hipStreamWaitEvent(pp.s_in, pp.ev_k[b], 0));
            SYM_GPU(ctx, hipStreamWaitEvent(ctx->stream, pp.ev_out[b], 0));
        }
        SYM_TRY(copy_rows(ctx, d_in[b], ng * 2304, h_xr + g0 * 576, granules_per_chain * 2304, ng * 2304, n_chains, hipMemcpyHostToDevice,
                          pp.s_in));
        SYM_TRY(copy_rows(ctx, d_side[b], ng * 4, h_side + g0, granules_per_chain * 4, ng * 4, n_chains, hipMemcpyHostToDevice, pp.s_in));
        SYM_GPU(ctx, hipEventRecord(pp.ev_in[b], pp.s_in));
        SYM_GPU(ctx, hipStreamWaitEvent(ctx->stream, pp.ev_in[b], 0));
        const int si = (int)(k & 1), so = (int)((k + 1) & 1);
        SYM_TRY(launch_mp3(ctx, d_in[b], d_side[b], sample_rate_idx, d_ov[si], d_vv[si], d_vf[si], d_ov[so], d_vv[so], d_vf[so], d_out[b],
                           n_chains, ng));
        SYM_GPU(ctx, hipEventRecord(pp.ev_k[b], ctx->stream));
        SYM_GPU(ctx, hipStreamWaitEvent(pp.s_out, pp.ev_k[b], 0));
        SYM_TRY(copy_rows(ctx, h_pcm + g0 * 576, granules_per_chain * 2304, d_out[b], ng * 2304, ng * 2304, n_chains, hipMemcpyDeviceToHost,
                          pp.s_out));
        SYM_GPU(ctx, hipEventRecord(pp.ev_out[b], pp.s_out));
    }
    const int sf = (int)(k & 1);
    SYM_GPU(ctx, hipMemcpyAsync(h_overlap_io, d_ov[sf], n_chains * 2304, hipMemcpyDeviceToHost, ctx->stream));
    SYM_GPU(ctx, hipMemcpyAsync(h_vvec_io, d_vv[sf], n_chains * 4096, hipMemcpyDeviceToHost, ctx->stream));
    SYM_GPU(ctx, hipMemcpyAsync(h_vfront_io, d_vf[sf], n_chains * 4, hipMemcpyDeviceToHost, ctx->stream));
    return pp.drain();
}

int symaccel_mp3_decode_pipelined(symaccel_ctx *ctx, const int16_t *h_quant, const symaccel_mp3_requant *h_rq_desc,
                                  const int32_t *h_pair_chains, const symaccel_mp3_stereo *h_st_desc, size_t n_pairs,
                                  const symaccel_mp3_side *h_side, int sample_rate_idx, float *h_overlap_io, float *h_vvec_io,
                                  int32_t *h_vfront_io, float *h_pcm, size_t n_chains, size_t granules_per_chain, size_t chunk_granules) {
    if (!ctx || sample_rate_idx < 0 || sample_rate_idx > 8) return SYMACCEL_ERR_INVALID_ARG;
    if (n_chains == 0 || granules_per_chain == 0) return SYMACCEL_OK;
    if (!h_quant || !h_rq_desc || !h_side || !h_overlap_io || !h_vvec_io || !h_vfront_io || !h_pcm) return SYMACCEL_ERR_INVALID_ARG;
    if (n_pairs && (!h_pair_chains || !h_st_desc)) return SYMACCEL_ERR_INVALID_ARG;
    // pair_chains is in host memory: check it (every chain at most once, inside the batch) and learn whether every chain is paired
    std::vector<uint8_t> paired(n_chains, 0);
    for (size_t p = 0; p < 2 * n_pairs; ++p) {
        const int32_t c = h_pair_chains[p];
        if (c < 0 || (size_t)c >= n_chains || paired[(size_t)c]) return SYMACCEL_ERR_INVALID_ARG;
        paired[(size_t)c] = 1;
    }
    // the fused kernel takes every stream as a unit: the pairs, then {chain, -1} for every chain no pair names (mono)
    std::vector<int32_t> units(h_pair_chains ? h_pair_chains : nullptr, h_pair_chains ? h_pair_chains + 2 * n_pairs : nullptr);
    for (size_t c = 0; c < n_chains; ++c)
        if (!paired[c]) {
            units.push_back((int32_t)c);
            units.push_back(-1);
        }
    const size_t n_units = units.size() / 2;
    DeviceGuard dev(ctx);
    if (!dev.ok()) return dev.status();
    size_t cg = pick_chunk(granules_per_chain, n_chains * 1152, chunk_granules);
    if (cg < 2 && granules_per_chain >= 2) cg = 2;  // the synthesis kernel's two-granule halo wants segments of at least two
    Pipe pp(ctx);
    SYM_TRY(pp.init());
    int16_t *d_q[2];
    symaccel_mp3_requant *d_rq[2];
    symaccel_mp3_stereo *d_st[2];
    symaccel_mp3_side *d_side[2];
    float *d_out[2], *d_ov[2], *d_vv[2];
    int32_t *d_vf[2], *d_pairs;
    for (int b = 0; b < 2; ++b) {
        SYM_TRY(pp.alloc((void **)&d_q[b], n_chains * cg * 1152));
        SYM_TRY(pp.alloc((void **)&d_rq[b], n_chains * cg * sizeof(symaccel_mp3_requant)));
        SYM_TRY(pp.alloc((void **)&d_st[b], n_units * cg * sizeof(symaccel_mp3_stereo)));
        SYM_TRY(pp.alloc((void **)&d_side[b], n_chains * cg * sizeof(symaccel_mp3_side)));
        SYM_TRY(pp.alloc((void **)&d_out[b], n_chains * cg * 2304));
        SYM_TRY(pp.alloc((void **)&d_ov[b], n_chains * 2304));
        SYM_TRY(pp.alloc((void **)&d_vv[b], n_chains * 4096));
        SYM_TRY(pp.alloc((void **)&d_vf[b], n_chains * 4));
    }
    SYM_TRY(pp.alloc((void **)&d_pairs, n_units * 8));
    SYM_TRY(pp.commit());
    // (the requantised spectra exist in registers and LDS only: csrc/mp3.hip mp3_front)
    SYM_GPU(ctx, hipMemcpy(d_pairs, units.data(), n_units * 8, hipMemcpyHostToDevice));  // (`units` is a local: a blocking copy)
    // The stereo records of mono units are never INTERPRETED: the kernel masks the joint-stereo flags of a unit whose second chain is
    // -1 (`pair_live`, csrc/mp3.hip), whatever rows [n_pairs, n_units) hold.  They are zeroed all the same, so that what the kernel
    // loads there is defined memory.
    for (int b = 0; b < 2; ++b) SYM_GPU(ctx, hipMemsetAsync(d_st[b], 0, n_units * cg * sizeof(symaccel_mp3_stereo), pp.s_in));
    SYM_GPU(ctx, hipMemcpyAsync(d_ov[0], h_overlap_io, n_chains * 2304, hipMemcpyHostToDevice, ctx->stream));
    SYM_GPU(ctx, hipMemcpyAsync(d_vv[0], h_vvec_io, n_chains * 4096, hipMemcpyHostToDevice, ctx->stream));
    SYM_GPU(ctx, hipMemcpyAsync(d_vf[0], h_vfront_io, n_chains * 4, hipMemcpyHostToDevice, ctx->stream));
    size_t k = 0;
    for (size_t g0 = 0; g0 < granules_per_chain; g0 += cg, ++k) {
        const size_t ng = std::min(cg, granules_per_chain - g0);
        const int b = (int)(k & 1);
        if (k >= 2) {
            SYM_GPU(ctx, hipStreamWaitEvent(pp.s_in, pp.ev_k[b], 0));
            SYM_GPU(ctx, hipStreamWaitEvent(ctx->stream, pp.ev_out[b], 0));
        }
        // what the entropy decoder produced: 2 bytes per line + 52-byte records (+ one 48-byte record per pair), a quarter of
        // the f32 spectra's bytes plus the side words
        SYM_TRY(copy_rows(ctx, d_q[b], ng * 1152, h_quant + g0 * 576, granules_per_chain * 1152, ng * 1152, n_chains, hipMemcpyHostToDevice, pp.s_in));
        SYM_TRY(copy_rows(ctx, d_rq[b], ng * sizeof(symaccel_mp3_requant), h_rq_desc + g0, granules_per_chain * sizeof(symaccel_mp3_requant),
                          ng * sizeof(symaccel_mp3_requant), n_chains, hipMemcpyHostToDevice, pp.s_in));
        if (n_pairs)
            SYM_TRY(copy_rows(ctx, d_st[b], ng * sizeof(symaccel_mp3_stereo), h_st_desc + g0, granules_per_chain * sizeof(symaccel_mp3_stereo),
                              ng * sizeof(symaccel_mp3_stereo), n_pairs, hipMemcpyHostToDevice, pp.s_in));
        SYM_TRY(copy_rows(ctx, d_side[b], ng * 4, h_side + g0, granules_per_chain * 4, ng * 4, n_chains, hipMemcpyHostToDevice, pp.s_in));
        if (n_units > n_pairs && ng != cg)  // a short last chunk: the kernel indexes st_desc with stride ng, so the mono units' rows move
            SYM_GPU(ctx, hipMemsetAsync(d_st[b] + n_pairs * ng, 0, (n_units - n_pairs) * ng * sizeof(symaccel_mp3_stereo), pp.s_in));  // (defined, not needed: see above)
        SYM_GPU(ctx, hipEventRecord(pp.ev_in[b], pp.s_in));
        SYM_GPU(ctx, hipStreamWaitEvent(ctx->stream, pp.ev_in[b], 0));
        // requantize, joint stereo and the synthesis tail (layer3/mod.rs:421-477) in one kernel
        const int si = (int)(k & 1), so = (int)((k + 1) & 1);
        SYM_TRY(launch_mp3_decode(ctx, d_q[b], d_rq[b], d_pairs, d_st[b], n_units, d_side[b], sample_rate_idx, d_ov[si], d_vv[si], d_vf[si],
                                  d_ov[so], d_vv[so], d_vf[so], d_out[b], n_chains, ng));
        SYM_GPU(ctx, hipEventRecord(pp.ev_k[b], ctx->stream));
        SYM_GPU(ctx, hipStreamWaitEvent(pp.s_out, pp.ev_k[b], 0));
        SYM_TRY(copy_rows(ctx, h_pcm + g0 * 576, granules_per_chain * 2304, d_out[b], ng * 2304, ng * 2304, n_chains, hipMemcpyDeviceToHost, pp.s_out));
        SYM_GPU(ctx, hipEventRecord(pp.ev_out[b], pp.s_out));
    }
    const int sf = (int)(k & 1);
    SYM_GPU(ctx, hipMemcpyAsync(h_overlap_io, d_ov[sf], n_chains * 2304, hipMemcpyDeviceToHost, ctx->stream));
    SYM_GPU(ctx, hipMemcpyAsync(h_vvec_io, d_vv[sf], n_chains * 4096, hipMemcpyDeviceToHost, ctx->stream));
    SYM_GPU(ctx, hipMemcpyAsync(h_vfront_io, d_vf[sf], n_chains * 4, hipMemcpyDeviceToHost, ctx->stream));
    return pp.drain();
}

int symaccel_flac_restore_pipelined(symaccel_ctx *ctx, int32_t *h_buf, const symaccel_flac_desc *h_desc, const int32_t *h_coeffs,
                                    size_t n_blocks, size_t blocksize, size_t chunk_blocks) {
    if (!ctx || blocksize > 65535) return SYMACCEL_ERR_INVALID_ARG;
    if (n_blocks == 0 || blocksize == 0) return SYMACCEL_OK;
    if (!h_buf || !h_desc || !h_coeffs) return SYMACCEL_ERR_INVALID_ARG;
    DeviceGuard dev(ctx);
    if (!dev.ok()) return dev.status();
    size_t cb = chunk_blocks ? chunk_blocks : std::max<size_t>(64, ((size_t)32 << 20) / (blocksize * 4));
    cb = std::min(cb, n_blocks);
    Pipe pp(ctx);
    SYM_TRY(pp.init());
    int32_t *d_buf[2], *d_co[2];
    symaccel_flac_desc *d_desc[2];
    for (int b = 0; b < 2; ++b) {
        SYM_TRY(pp.alloc((void **)&d_buf[b], cb * blocksize * 4));
        SYM_TRY(pp.alloc((void **)&d_co[b], cb * 32 * 4));
        SYM_TRY(pp.alloc((void **)&d_desc[b], cb * sizeof(symaccel_flac_desc)));
    }
    SYM_TRY(pp.commit());
    size_t k = 0;
    for (size_t b0 = 0; b0 < n_blocks; b0 += cb, ++k) {
        const size_t nb = std::min(cb, n_blocks - b0);
        const int b = (int)(k & 1);
        if (k >= 2) SYM_GPU(ctx, hipStreamWaitEvent(pp.s_in, pp.ev_out[b], 0));  // in place: the buffer is free once its result has left
        SYM_GPU(ctx, hipMemcpyAsync(d_buf[b], h_buf + b0 * blocksize, nb * blocksize * 4, hipMemcpyHostToDevice, pp.s_in));
        SYM_GPU(ctx, hipMemcpyAsync(d_co[b], h_coeffs + b0 * 32, nb * 32 * 4, hipMemcpyHostToDevice, pp.s_in));
        SYM_GPU(ctx, hipMemcpyAsync(d_desc[b], h_desc + b0, nb * sizeof(symaccel_flac_desc), hipMemcpyHostToDevice, pp.s_in));
        SYM_GPU(ctx, hipEventRecord(pp.ev_in[b], pp.s_in));
        SYM_GPU(ctx, hipStreamWaitEvent(ctx->stream, pp.ev_in[b], 0));
        SYM_TRY(launch_flac_restore(ctx, d_buf[b], d_desc[b], d_co[b], nb, blocksize));
        SYM_GPU(ctx, hipEventRecord(pp.ev_k[b], ctx->stream));
        SYM_GPU(ctx, hipStreamWaitEvent(pp.s_out, pp.ev_k[b], 0));
        SYM_GPU(ctx, hipMemcpyAsync(h_buf + b0 * blocksize, d_buf[b], nb * blocksize * 4, hipMemcpyDeviceToHost, pp.s_out));
        SYM_GPU(ctx, hipEventRecord(pp.ev_out[b], pp.s_out));
    }
    return pp.drain();
}

}  // extern "C"
