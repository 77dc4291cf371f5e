// Context management and the C-ABI entry points (argument checking, host<->HBM staging, state
// double-buffering).  The kernels live in the .hip files; nothing here computes audio.
#include <cstring>
#include <new>

#include "symaccel_internal.h"

using namespace symaccel;

namespace symaccel {

int ctx_fail(symaccel_ctx *ctx, hipError_t err, const char *where) {
    if (ctx) {
        ctx->last_error = std::string(where) + ": " + hipGetErrorString(err);
    }
    return err == hipErrorOutOfMemory ? SYMACCEL_ERR_OOM : SYMACCEL_ERR_DEVICE;
}

int ctx_alloc(symaccel_ctx *ctx, void **out, size_t bytes, bool tracked) {
    *out = nullptr;
    if (bytes == 0) bytes = 16;
    SYM_GPU(ctx, hipMalloc(out, bytes));
    if (tracked) ctx->allocations.push_back(*out);
    return SYMACCEL_OK;
}

int ctx_sink(symaccel_ctx *ctx, void **out) {
    if (!ctx->sink) SYM_TRY(ctx_alloc(ctx, &ctx->sink, kSinkBytes, false));
    *out = ctx->sink;
    return SYMACCEL_OK;
}

int ctx_scratch(symaccel_ctx *ctx, size_t bytes, void **out) {
    if (bytes > ctx->scratch_bytes) {
        if (ctx->scratch) {
            // earlier work on the stream may still read the old scratch
            SYM_GPU(ctx, hipStreamSynchronize(ctx->stream));
            SYM_GPU(ctx, hipFree(ctx->scratch));
            ctx->scratch = nullptr;
            ctx->scratch_bytes = 0;
        }
        size_t want = bytes + bytes / 4 + 4096;
        SYM_TRY(ctx_alloc(ctx, &ctx->scratch, want, false));
        ctx->scratch_bytes = want;
    }
    *out = ctx->scratch;
    return SYMACCEL_OK;
}

int ctx_upload(symaccel_ctx *ctx, const void *src, size_t bytes, const void **out) {
    void *d = nullptr;
    SYM_TRY(ctx_alloc(ctx, &d, bytes));
    SYM_GPU(ctx, hipMemcpy(d, src, bytes, hipMemcpyHostToDevice));
    *out = d;
    return SYMACCEL_OK;
}

// Segment length for the chain-walking kernels.  Every item (chain, segment) costs seg + halo frames; the
// machine runs `slots` items at a time (resident wavefronts x items per wavefront), so a launch takes about
// ceil(items / slots) rounds of (seg + halo) frames.  Pick the seg that minimises that: it fills the chip
// with a whole number of rounds (no half-empty tail round) at the smallest halo overhead.
unsigned choose_segment(const symaccel_ctx *ctx, size_t n_chains, size_t frames_per_chain, unsigned waves_per_cu,
                        unsigned items_per_wave, unsigned halo, unsigned min_seg) {
    if (frames_per_chain == 0) return 1;
    if (ctx->segment > 0) {
        unsigned s = (unsigned)ctx->segment < min_seg ? min_seg : (unsigned)ctx->segment;
        return s > frames_per_chain ? (unsigned)frames_per_chain : s;
    }
    const double slots = (double)ctx->n_cus * waves_per_cu * items_per_wave;
    const unsigned lo = min_seg, hi = frames_per_chain < 256 ? (unsigned)frames_per_chain : 256u;
    if (hi <= lo) return hi < 1 ? 1 : hi;
    unsigned best = hi;
    double best_cost = 1e300;
    for (unsigned seg = lo; seg <= hi; ++seg) {
        const double items = (double)n_chains * (double)((frames_per_chain + seg - 1) / seg);
        const double rounds = items <= slots ? 1.0 : (double)(size_t)((items + slots - 1) / slots);
        // an under-filled single round still runs faster per wavefront; credit half of the idle fraction
        const double fill = items < slots ? 0.5 + 0.5 * items / slots : 1.0;
        const double cost = rounds * (double)(seg + halo) * fill;
        if (cost < best_cost * 0.999 || (cost <= best_cost * 1.001 && seg > best)) {
            if (cost < best_cost) best_cost = cost;
            best = seg;
        }
    }
    return best;
}

int get_imdct_plan(symaccel_ctx *ctx, int n, double scale, const ImdctPlan **out) {
    uint64_t bits;
    std::memcpy(&bits, &scale, 8);
    auto key = std::make_pair(n, bits);
    auto it = ctx->imdct_plans.find(key);
    if (it == ctx->imdct_plans.end()) {
        std::vector<cpx> tw((size_t)n / 2);
        make_imdct_twiddles(n, scale, tw.data());
        const void *d = nullptr;
        SYM_TRY(ctx_upload(ctx, tw.data(), tw.size() * sizeof(cpx), &d));
        it = ctx->imdct_plans.emplace(key, ImdctPlan{n, (cpx *)d}).first;
    }
    *out = &it->second;
    return SYMACCEL_OK;
}

int get_vorbis_window(symaccel_ctx *ctx, int bs, const float **out) {
    auto it = ctx->vorbis_windows.find(bs);
    if (it == ctx->vorbis_windows.end()) {
        std::vector<float> w((size_t)bs / 2);
        make_vorbis_window(bs, w.data());
        const void *d = nullptr;
        SYM_TRY(ctx_upload(ctx, w.data(), w.size() * sizeof(float), &d));
        it = ctx->vorbis_windows.emplace(bs, (float *)d).first;
    }
    *out = it->second;
    return SYMACCEL_OK;
}

namespace {

// reorder (hybrid_synthesis.rs:153-215) as a gather: for each (sample-rate, mixed) the source
// index of every destination line, and for every incoming rzero the index `i` the reference's
// interleave loop stops at (bands that start at or beyond rzero are skipped).
void build_reorder_tables(const HostTables &t, std::vector<int32_t> &map, std::vector<int32_t> &end) {
    map.assign(9 * 2 * 576, 0);
    end.assign(9 * 2 * 577, 0);
    for (int sr = 0; sr < 9; ++sr) {
        for (int mixed = 0; mixed < 2; ++mixed) {
            const int32_t *bands;
            int n_bands;
            if (mixed) {
                bands = t.mp3_sfb_mixed[sr] + t.mp3_sfb_switch[sr];
                n_bands = t.mp3_sfb_mixed_len[sr] - t.mp3_sfb_switch[sr];
            } else {
                bands = t.mp3_sfb_short[sr];
                n_bands = 40;
            }
            int32_t *m = &map[(size_t)(sr * 2 + mixed) * 576];
            int32_t *e = &end[(size_t)(sr * 2 + mixed) * 577];
            for (int i = 0; i < 576; ++i) m[i] = i;
            const int start = bands[0];
            for (int rz = 0; rz <= 576; ++rz) e[rz] = start;
            int i = start;
            for (int b = 0; b + 3 < n_bands; b += 3) {
                const int s0 = bands[b], s1 = bands[b + 1], s2 = bands[b + 2], s3 = bands[b + 3];
                int len = s1 - s0;
                if (s2 - s1 < len) len = s2 - s1;
                if (s3 - s2 < len) len = s3 - s2;
                for (int k = 0; k < len; ++k) {
                    if (i + 2 < 576) {
                        m[i + 0] = s0 + k;
                        m[i + 1] = s1 + k;
                        m[i + 2] = s2 + k;
                    }
                    i += 3;
                }
                // this band is processed iff s0 < rzero
                for (int rz = s0 + 1; rz <= 576; ++rz) e[rz] = i;
            }
        }
    }
}

// DevTables::mp3_consts layout (Mp3ConstLayout) from the host tables.
static std::vector<float> pack_mp3_consts(const HostTables &t) {
    std::vector<float> mc(MP3C_TOTAL, 0.0f);
    std::memcpy(&mc[MP3C_IMDCT_WIN], t.mp3_imdct_win, 144 * 4);
    std::memcpy(&mc[MP3C_COS12], t.mp3_cos12, 36 * 4);
    std::memcpy(&mc[MP3C_CS], t.mp3_cs, 8 * 4);
    std::memcpy(&mc[MP3C_CA], t.mp3_ca, 8 * 4);
    std::memcpy(&mc[MP3C_DCT_IV], t.mp3_dct_iv_scale, 18 * 4);
    std::memcpy(&mc[MP3C_SDCT18], t.mp3_sdct18_scale, 9 * 4);
    std::memcpy(&mc[MP3C_SDCT9_D], t.mp3_sdct9_d, 7 * 4);
    std::memcpy(&mc[MP3C_COS16], t.mp3_cos16, 16 * 4);
    std::memcpy(&mc[MP3C_COS8], t.mp3_cos8, 8 * 4);
    std::memcpy(&mc[MP3C_COS4], t.mp3_cos4, 4 * 4);
    std::memcpy(&mc[MP3C_COS2], t.mp3_cos2, 2 * 4);
    mc[MP3C_COS1] = t.mp3_cos1;
    std::memcpy(&mc[MP3C_SYNTH_D], t.mp3_synth_d, 512 * 4);
    return mc;
}

int upload_tables(symaccel_ctx *ctx) {
    const HostTables &t = host_tables();
    DevTables &d = ctx->dev;
    const void *p = nullptr;
#define UP(field, src, bytes)                         \
    SYM_TRY(ctx_upload(ctx, (src), (bytes), &p));     \
    d.field = (decltype(d.field))p
    UP(aac_kbd_long, t.aac_kbd_long.data(), 1024 * 4);
    UP(aac_kbd_short, t.aac_kbd_short.data(), 128 * 4);
    UP(aac_sine_long, t.aac_sine_long.data(), 1024 * 4);
    UP(aac_sine_short, t.aac_sine_short.data(), 128 * 4);
    UP(aac_tw_long, t.aac_tw_long.data(), 512 * sizeof(cpx));
    UP(aac_tw_short, t.aac_tw_short.data(), 64 * sizeof(cpx));
    UP(fft_merge, t.fft_merge.data(), t.fft_merge.size() * sizeof(cpx));
    UP(small16, t.small16, sizeof t.small16);
    UP(small32, t.small32, sizeof t.small32);
    UP(small16_form, t.small16_form, sizeof t.small16_form);
    UP(small32_form, t.small32_form, sizeof t.small32_form);
    std::vector<float> mc = pack_mp3_consts(t);
    UP(mp3_consts, mc.data(), mc.size() * 4);
    std::vector<int32_t> rmap, rend;
    build_reorder_tables(t, rmap, rend);
    UP(mp3_reorder_map, rmap.data(), rmap.size() * 4);
    UP(mp3_reorder_end, rend.data(), rend.size() * 4);
    UP(vorbis_floor1_db, t.vorbis_floor1_db, 256 * 4);
    UP(mp3_pow43, t.mp3_pow43, sizeof t.mp3_pow43);
    UP(mp3_pow2ab, t.mp3_pow2ab, sizeof t.mp3_pow2ab);
    UP(mp3_band_map, t.mp3_band_map, sizeof t.mp3_band_map);
    UP(mp3_is_ratios, t.mp3_is_ratios, sizeof t.mp3_is_ratios);
#undef UP
    return SYMACCEL_OK;
}

// Host-pointer convenience wrappers stage through tracked-free temporaries.
struct DevBuf {
    symaccel_ctx *ctx;
    void *p = nullptr;
    explicit DevBuf(symaccel_ctx *c) : ctx(c) {}
    ~DevBuf() {
        if (p) (void)hipFree(p);
    }
    int alloc(size_t bytes) { return ctx_alloc(ctx, &p, bytes, false); }
    int from_host(const void *h, size_t bytes) {
        SYM_TRY(alloc(bytes));
        if (bytes) SYM_GPU(ctx, hipMemcpyAsync(p, h, bytes, hipMemcpyHostToDevice, ctx->stream));
        return SYMACCEL_OK;
    }
    int to_host(void *h, size_t bytes) {
        if (bytes) SYM_GPU(ctx, hipMemcpyAsync(h, p, bytes, hipMemcpyDeviceToHost, ctx->stream));
        return SYMACCEL_OK;
    }
};

bool pow2(long v) { return v > 0 && (v & (v - 1)) == 0; }
constexpr size_t kPipelineBytes = (size_t)64 << 20;  // host batches from this size on are staged in overlapped chunks (stage.cpp)
constexpr int kMaxFft = 65536;  // the reference's limit (no_simd.rs:77-80); above 4096 points: imdct_generic.hip, "big" path

}  // namespace
}  // namespace symaccel

// ------------------------------------------------------------------------------------ ABI

extern "C" {

int symaccel_abi_version(void) { return SYMACCEL_ABI_VERSION; }

const char *symaccel_strerror(int status) {
    switch (status) {
        case SYMACCEL_OK: return "ok";
        case SYMACCEL_ERR_INVALID_ARG: return "symaccel: invalid argument";
        case SYMACCEL_ERR_UNSUPPORTED: return "symaccel: unsupported configuration";
        case SYMACCEL_ERR_DEVICE: return "symaccel: HIP device error";
        case SYMACCEL_ERR_OOM: return "symaccel: out of memory";
        case SYMACCEL_ERR_DECODE: return "symaccel: malformed stream data";
        default: return "symaccel: unknown status";
    }
}

const char *symaccel_last_error(const symaccel_ctx *ctx) { return ctx ? ctx->last_error.c_str() : ""; }

int symaccel_ctx_create(int device, symaccel_ctx **out) {
    if (!out) return SYMACCEL_ERR_INVALID_ARG;
    *out = nullptr;
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count <= 0) return SYMACCEL_ERR_DEVICE;  // no CPU path
    if (device < 0 || device >= count) return SYMACCEL_ERR_INVALID_ARG;
    symaccel_ctx *ctx = new (std::nothrow) symaccel_ctx();
    if (!ctx) return SYMACCEL_ERR_OOM;
    ctx->device = device;
    int st = SYMACCEL_OK;
    int prev_device = -1;
    if (hipGetDevice(&prev_device) != hipSuccess) prev_device = -1;
    if (hipSetDevice(device) != hipSuccess || hipStreamCreate(&ctx->own_stream) != hipSuccess) {
        st = SYMACCEL_ERR_DEVICE;
    } else {
        ctx->stream = ctx->own_stream;
        int cus = 0;
        if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device) == hipSuccess && cus > 0)
            ctx->n_cus = cus;
        st = upload_tables(ctx);
    }
    if (st != SYMACCEL_OK) {
        symaccel_ctx_destroy(ctx);
    } else {
        *out = ctx;
    }
    if (prev_device >= 0 && prev_device != device) (void)hipSetDevice(prev_device);  // the caller's current device stays put
    return st;
}

void symaccel_ctx_destroy(symaccel_ctx *ctx) {
    if (!ctx) return;
    int prev_device = -1;
    if (hipGetDevice(&prev_device) != hipSuccess) prev_device = -1;
    (void)hipSetDevice(ctx->device);
    if (ctx->stream) (void)hipStreamSynchronize(ctx->stream);
    for (void *p : ctx->allocations) (void)hipFree(p);
    if (ctx->scratch) (void)hipFree(ctx->scratch);
    if (ctx->stage_in) (void)hipStreamSynchronize(ctx->stage_in);
    if (ctx->stage_out) (void)hipStreamSynchronize(ctx->stage_out);
    if (ctx->stage_arena) (void)hipFree(ctx->stage_arena);
    if (ctx->alac_flags) (void)hipFree(ctx->alac_flags);
    if (ctx->sink) (void)hipFree(ctx->sink);
    for (hipEvent_t e : ctx->stage_events)
        if (e) (void)hipEventDestroy(e);
    if (ctx->stage_in) (void)hipStreamDestroy(ctx->stage_in);
    if (ctx->stage_out) (void)hipStreamDestroy(ctx->stage_out);
    if (ctx->own_stream) (void)hipStreamDestroy(ctx->own_stream);
    if (prev_device >= 0 && prev_device != ctx->device) (void)hipSetDevice(prev_device);
    delete ctx;
}

int symaccel_ctx_set_stream(symaccel_ctx *ctx, void *hip_stream) {
    if (!ctx) return SYMACCEL_ERR_INVALID_ARG;
    hipStream_t next = hip_stream ? (hipStream_t)hip_stream : ctx->own_stream;
    if (next != ctx->stream) {
        // The context's scratch (state double buffers of the *_io entry points) is ordered by the stream alone:
        // work still queued on the old stream must not overlap work on the new one.
        DeviceGuard dev(ctx);
        if (!dev.ok()) return dev.status();
        if (ctx->stream) SYM_GPU(ctx, hipStreamSynchronize(ctx->stream));
        ctx->stream = next;
    }
    return SYMACCEL_OK;
}

int symaccel_sync(symaccel_ctx *ctx) {
    if (!ctx) return SYMACCEL_ERR_INVALID_ARG;
    SYM_GPU(ctx, hipStreamSynchronize(ctx->stream));
    return SYMACCEL_OK;
}

int symaccel_ctx_set_segment(symaccel_ctx *ctx, int frames_per_segment) {
    if (!ctx || frames_per_segment < 0) return SYMACCEL_ERR_INVALID_ARG;
    ctx->segment = frames_per_segment;
    return SYMACCEL_OK;
}

// ---- core ---------------------------------------------------------------------------------

static int fft_c32_device(symaccel_ctx *ctx, int n, const float *d_in, float *d_out, size_t count, bool inverse) {
    if (!ctx || !pow2(n) || n < 2 || n > 65536) return SYMACCEL_ERR_INVALID_ARG;  // no_simd.rs:77-80, 152-156
    if (n > kMaxFft) return SYMACCEL_ERR_UNSUPPORTED;
    if (count == 0) return SYMACCEL_OK;
    if (!d_in || !d_out) return SYMACCEL_ERR_INVALID_ARG;
    DeviceGuard dev(ctx);
    if (!dev.ok()) return dev.status();
    return launch_fft(ctx, n, d_in, d_out, count, inverse);
}
int symaccel_fft_c32_device(symaccel_ctx *ctx, int n, const float *d_in, float *d_out, size_t count) {
    return fft_c32_device(ctx, n, d_in, d_out, count, false);
}
int symaccel_ifft_c32_device(symaccel_ctx *ctx, int n, const float *d_in, float *d_out, size_t count) {
    return fft_c32_device(ctx, n, d_in, d_out, count, true);
}

int symaccel_imdct_f32_device(symaccel_ctx *ctx, int n, double scale, const float *d_spec, float *d_out,
                              size_t count) {
    if (!ctx || !pow2(n) || n < 4 || n > 131072) return SYMACCEL_ERR_INVALID_ARG;  // mdct.rs:37-40
    if (n > 2 * kMaxFft) return SYMACCEL_ERR_UNSUPPORTED;
    if (count == 0) return SYMACCEL_OK;
    if (!d_spec || !d_out) return SYMACCEL_ERR_INVALID_ARG;
    DeviceGuard dev(ctx);
    if (!dev.ok()) return dev.status();
    const ImdctPlan *plan = nullptr;
    SYM_TRY(get_imdct_plan(ctx, n, scale, &plan));
    return launch_imdct(ctx, *plan, d_spec, d_out, count);
}

static int fft_c32_host(symaccel_ctx *ctx, int n, const float *h_in, float *h_out, size_t count, bool inverse) {
    if (!ctx || !pow2(n) || n < 2 || n > 65536) return SYMACCEL_ERR_INVALID_ARG;
    if (n > kMaxFft) return SYMACCEL_ERR_UNSUPPORTED;
    if (count == 0) return SYMACCEL_OK;
    if (!h_in || !h_out) return SYMACCEL_ERR_INVALID_ARG;
    DeviceGuard dev(ctx);
    if (!dev.ok()) return dev.status();
    DevBuf buf(ctx);
    const size_t bytes = count * (size_t)n * 8;
    SYM_TRY(buf.from_host(h_in, bytes));
    SYM_TRY(launch_fft(ctx, n, (const float *)buf.p, (float *)buf.p, count, inverse));
    SYM_TRY(buf.to_host(h_out, bytes));
    return symaccel_sync(ctx);
}
int symaccel_fft_c32(symaccel_ctx *ctx, int n, const float *h_in, float *h_out, size_t count) {
    return fft_c32_host(ctx, n, h_in, h_out, count, false);
}
int symaccel_ifft_c32(symaccel_ctx *ctx, int n, const float *h_in, float *h_out, size_t count) {
    return fft_c32_host(ctx, n, h_in, h_out, count, true);
}

int symaccel_imdct_f32(symaccel_ctx *ctx, int n, double scale, const float *h_spec, float *h_out, size_t count) {
    if (!ctx || !pow2(n) || n < 4 || n > 131072) return SYMACCEL_ERR_INVALID_ARG;
    if (n > 2 * kMaxFft) return SYMACCEL_ERR_UNSUPPORTED;
    if (count == 0) return SYMACCEL_OK;
    if (!h_spec || !h_out) return SYMACCEL_ERR_INVALID_ARG;
    DeviceGuard dev(ctx);
    if (!dev.ok()) return dev.status();
    DevBuf in(ctx), out(ctx);
    SYM_TRY(in.from_host(h_spec, count * (size_t)n * 4));
    SYM_TRY(out.alloc(count * (size_t)n * 8));
    SYM_TRY(symaccel_imdct_f32_device(ctx, n, scale, (const float *)in.p, (float *)out.p, count));
    SYM_TRY(out.to_host(h_out, count * (size_t)n * 8));
    return symaccel_sync(ctx);
}

// ---- AAC ----------------------------------------------------------------------------------

static bool build_band_map(const uint16_t *swb, int n_swb, int lines, int max_bands, uint8_t *map4);

// Dsp::synth with the channel pairs' joint-stereo decoding (cpe.rs:110-157) done as the lines are loaded
static int aac_synth_js(symaccel_ctx *ctx, const float *d_coeffs, const uint8_t *d_side, const int32_t *d_pair_chains,
                        const symaccel_aac_js_frame *d_js_desc, size_t n_pairs, const uint16_t *swb_long, int n_swb_long,
                        const uint16_t *swb_short, int n_swb_short, const float *d_delay_in, float *d_delay_out, float *d_pcm,
                        size_t n_chains, size_t frames_per_chain, void **scratch_tail, size_t tail_bytes) {
    AacBandMaps maps;
    if (!build_band_map(swb_long, n_swb_long, 1024, 64, maps.long4) || !build_band_map(swb_short, n_swb_short, 128, 16, maps.short4))
        return SYMACCEL_ERR_INVALID_ARG;
    if (n_pairs && (!d_pair_chains || !d_js_desc)) return SYMACCEL_ERR_INVALID_ARG;
    if (2 * n_pairs > n_chains) return SYMACCEL_ERR_INVALID_ARG;
    void *scratch = nullptr;
    const size_t js_bytes = aac_js_scratch_bytes(n_chains, n_pairs, frames_per_chain);
    SYM_TRY(ctx_scratch(ctx, js_bytes + tail_bytes, &scratch));
    if (scratch_tail) *scratch_tail = static_cast<char *>(scratch) + js_bytes;
    float *out_state = d_delay_out ? d_delay_out : reinterpret_cast<float *>(static_cast<char *>(scratch) + js_bytes);
    return launch_aac(ctx, d_coeffs, d_side, d_delay_in, out_state, d_pcm, n_chains, frames_per_chain, &maps, d_pair_chains, d_js_desc, n_pairs,
                      scratch);
}

int symaccel_aac_synth_js_pp_device(symaccel_ctx *ctx, const float *d_coeffs, const uint8_t *d_side, const int32_t *d_pair_chains,
                                    const symaccel_aac_js_frame *d_js_desc, size_t n_pairs, const uint16_t *swb_long, int n_swb_long,
                                    const uint16_t *swb_short, int n_swb_short, const float *d_delay_in, float *d_delay_out,
                                    float *d_pcm, size_t n_chains, size_t frames_per_chain) {
    if (!ctx) return SYMACCEL_ERR_INVALID_ARG;
    if (n_chains == 0 || frames_per_chain == 0) return SYMACCEL_OK;
    if (!d_coeffs || !d_side || !d_delay_in || !d_delay_out || !d_pcm || d_delay_in == d_delay_out) return SYMACCEL_ERR_INVALID_ARG;
    DeviceGuard dev(ctx);
    if (!dev.ok()) return dev.status();
    return aac_synth_js(ctx, d_coeffs, d_side, d_pair_chains, d_js_desc, n_pairs, swb_long, n_swb_long, swb_short, n_swb_short, d_delay_in,
                        d_delay_out, d_pcm, n_chains, frames_per_chain, nullptr, 0);
}

int symaccel_aac_synth_js_device(symaccel_ctx *ctx, const float *d_coeffs, const uint8_t *d_side, const int32_t *d_pair_chains,
                                 const symaccel_aac_js_frame *d_js_desc, size_t n_pairs, const uint16_t *swb_long, int n_swb_long,
                                 const uint16_t *swb_short, int n_swb_short, float *d_delay_io, float *d_pcm, size_t n_chains,
                                 size_t frames_per_chain) {
    if (!ctx) return SYMACCEL_ERR_INVALID_ARG;
    if (n_chains == 0 || frames_per_chain == 0) return SYMACCEL_OK;
    if (!d_coeffs || !d_side || !d_delay_io || !d_pcm) return SYMACCEL_ERR_INVALID_ARG;
    DeviceGuard dev(ctx);
    if (!dev.ok()) return dev.status();
    const size_t state_bytes = n_chains * 1024 * sizeof(float);
    void *tail = nullptr;  // the new state goes to scratch and is copied back (segments of a chain run concurrently)
    SYM_TRY(aac_synth_js(ctx, d_coeffs, d_side, d_pair_chains, d_js_desc, n_pairs, swb_long, n_swb_long, swb_short, n_swb_short, d_delay_io,
                         nullptr, d_pcm, n_chains, frames_per_chain, &tail, state_bytes));
    return launch_state_copy(ctx, d_delay_io, tail, state_bytes, nullptr, nullptr, 0, nullptr, nullptr, 0);
}

int symaccel_aac_synth_pp_device(symaccel_ctx *ctx, const float *d_coeffs, const uint8_t *d_side, const float *d_delay_in,
                                 float *d_delay_out, float *d_pcm, size_t n_chains, size_t frames_per_chain) {
    if (!ctx) return SYMACCEL_ERR_INVALID_ARG;
    if (n_chains == 0 || frames_per_chain == 0) return SYMACCEL_OK;
    if (!d_coeffs || !d_side || !d_delay_in || !d_delay_out || !d_pcm || d_delay_in == d_delay_out) return SYMACCEL_ERR_INVALID_ARG;
    DeviceGuard dev(ctx);
    if (!dev.ok()) return dev.status();
    // one launch, nothing else: the first segment of a chain reads delay_in, the last one writes delay_out
    return launch_aac(ctx, d_coeffs, d_side, d_delay_in, d_delay_out, d_pcm, n_chains, frames_per_chain);
}

int symaccel_aac_synth_device(symaccel_ctx *ctx, const float *d_coeffs, const uint8_t *d_side, float *d_delay_io,
                              float *d_pcm, size_t n_chains, size_t frames_per_chain) {
    if (!ctx) return SYMACCEL_ERR_INVALID_ARG;
    if (n_chains == 0 || frames_per_chain == 0) return SYMACCEL_OK;
    if (!d_coeffs || !d_side || !d_delay_io || !d_pcm) return SYMACCEL_ERR_INVALID_ARG;
    DeviceGuard dev(ctx);
    if (!dev.ok()) return dev.status();
    // Segments of one chain run concurrently: the first reads the incoming delay line while the
    // last writes the outgoing one, so the new state goes to scratch and is copied back after
    // (callers that keep two state buffers use the _pp_ entry point and skip the copy).
    void *scratch = nullptr;
    const size_t state_bytes = n_chains * 1024 * sizeof(float);
    SYM_TRY(ctx_scratch(ctx, state_bytes, &scratch));
    SYM_TRY(launch_aac(ctx, d_coeffs, d_side, d_delay_io, (float *)scratch, d_pcm, n_chains, frames_per_chain));
    SYM_TRY(launch_state_copy(ctx, d_delay_io, scratch, state_bytes, nullptr, nullptr, 0, nullptr, nullptr, 0));
    return SYMACCEL_OK;
}

int symaccel_aac_synth(symaccel_ctx *ctx, const float *h_coeffs, const uint8_t *h_side, float *h_delay_io,
                       float *h_pcm, size_t n_chains, size_t frames_per_chain) {
    if (!ctx) return SYMACCEL_ERR_INVALID_ARG;
    if (n_chains == 0 || frames_per_chain == 0) return SYMACCEL_OK;
    if (!h_coeffs || !h_side || !h_delay_io || !h_pcm) return SYMACCEL_ERR_INVALID_ARG;
    if (n_chains * frames_per_chain * 4096 >= kPipelineBytes && frames_per_chain >= 16)
        return symaccel_aac_synth_pipelined(ctx, h_coeffs, h_side, h_delay_io, h_pcm, n_chains, frames_per_chain, 0);
    DeviceGuard dev(ctx);
    if (!dev.ok()) return dev.status();
    const size_t nf = n_chains * frames_per_chain;
    DevBuf coeffs(ctx), side(ctx), delay(ctx), pcm(ctx);
    SYM_TRY(coeffs.from_host(h_coeffs, nf * 4096));
    SYM_TRY(side.from_host(h_side, nf));
    SYM_TRY(delay.from_host(h_delay_io, n_chains * 4096));
    SYM_TRY(pcm.alloc(nf * 4096));
    SYM_TRY(symaccel_aac_synth_device(ctx, (const float *)coeffs.p, (const uint8_t *)side.p, (float *)delay.p,
                                      (float *)pcm.p, n_chains, frames_per_chain));
    SYM_TRY(pcm.to_host(h_pcm, nf * 4096));
    SYM_TRY(delay.to_host(h_delay_io, n_chains * 4096));
    return symaccel_sync(ctx);
}

// ---- MP3 ----------------------------------------------------------------------------------

int symaccel_mp3_synth_pp_device(symaccel_ctx *ctx, const float *d_xr, const symaccel_mp3_side *d_side, int sample_rate_idx,
                                 const float *d_overlap_in, const float *d_vvec_in, const int32_t *d_vfront_in,
                                 float *d_overlap_out, float *d_vvec_out, int32_t *d_vfront_out, float *d_pcm, size_t n_chains,
                                 size_t granules_per_chain) {
    if (!ctx || sample_rate_idx < 0 || sample_rate_idx > 8) return SYMACCEL_ERR_INVALID_ARG;
    if (n_chains == 0 || granules_per_chain == 0) return SYMACCEL_OK;
    if (!d_xr || !d_side || !d_overlap_in || !d_vvec_in || !d_vfront_in || !d_overlap_out || !d_vvec_out || !d_vfront_out || !d_pcm)
        return SYMACCEL_ERR_INVALID_ARG;
    if (d_overlap_in == d_overlap_out || d_vvec_in == d_vvec_out || d_vfront_in == d_vfront_out) return SYMACCEL_ERR_INVALID_ARG;
    DeviceGuard dev(ctx);
    if (!dev.ok()) return dev.status();
    return launch_mp3(ctx, d_xr, d_side, sample_rate_idx, d_overlap_in, d_vvec_in, d_vfront_in, d_overlap_out, d_vvec_out,
                      d_vfront_out, d_pcm, n_chains, granules_per_chain);
}

int symaccel_mp3_synth_device(symaccel_ctx *ctx, const float *d_xr, const symaccel_mp3_side *d_side,
                              int sample_rate_idx, float *d_overlap_io, float *d_vvec_io, int32_t *d_vfront_io,
                              float *d_pcm, size_t n_chains, size_t granules_per_chain) {
    if (!ctx || sample_rate_idx < 0 || sample_rate_idx > 8) return SYMACCEL_ERR_INVALID_ARG;
    if (n_chains == 0 || granules_per_chain == 0) return SYMACCEL_OK;
    if (!d_xr || !d_side || !d_overlap_io || !d_vvec_io || !d_vfront_io || !d_pcm) return SYMACCEL_ERR_INVALID_ARG;
    DeviceGuard dev(ctx);
    if (!dev.ok()) return dev.status();
    const size_t ov_bytes = n_chains * 576 * 4, vv_bytes = n_chains * 1024 * 4, vf_bytes = n_chains * 4;
    void *scratch = nullptr;
    SYM_TRY(ctx_scratch(ctx, ov_bytes + vv_bytes + vf_bytes, &scratch));
    float *ov_out = (float *)scratch;
    float *vv_out = ov_out + n_chains * 576;
    int32_t *vf_out = (int32_t *)(vv_out + n_chains * 1024);
    SYM_TRY(launch_mp3(ctx, d_xr, d_side, sample_rate_idx, d_overlap_io, d_vvec_io, d_vfront_io, ov_out, vv_out,
                       vf_out, d_pcm, n_chains, granules_per_chain));
    SYM_TRY(launch_state_copy(ctx, d_overlap_io, ov_out, ov_bytes, d_vvec_io, vv_out, vv_bytes, d_vfront_io, vf_out,
                              vf_bytes));
    return SYMACCEL_OK;
}

int symaccel_mp3_decode_pp_device(symaccel_ctx *ctx, const int16_t *d_quant, const symaccel_mp3_requant *d_rq_desc,
                                  const int32_t *d_unit_chains, const symaccel_mp3_stereo *d_st_desc, size_t n_units,
                                  const symaccel_mp3_side *d_side, int sample_rate_idx, const float *d_overlap_in,
                                  const float *d_vvec_in, const int32_t *d_vfront_in, float *d_overlap_out, float *d_vvec_out,
                                  int32_t *d_vfront_out, float *d_pcm, size_t n_chains, size_t granules_per_chain) {
    if (!ctx || sample_rate_idx < 0 || sample_rate_idx > 8) return SYMACCEL_ERR_INVALID_ARG;
    if (n_chains == 0 || granules_per_chain == 0) return SYMACCEL_OK;
    if (!d_quant || !d_rq_desc || !d_unit_chains || !d_st_desc || !d_side || !d_overlap_in || !d_vvec_in || !d_vfront_in ||
        !d_overlap_out || !d_vvec_out || !d_vfront_out || !d_pcm)
        return SYMACCEL_ERR_INVALID_ARG;
    if (n_units == 0 || n_units > n_chains || 2 * n_units < n_chains) return SYMACCEL_ERR_INVALID_ARG;  // every chain in exactly one unit
    if (d_overlap_in == d_overlap_out || d_vvec_in == d_vvec_out || d_vfront_in == d_vfront_out) return SYMACCEL_ERR_INVALID_ARG;
    DeviceGuard dev(ctx);
    if (!dev.ok()) return dev.status();
    return launch_mp3_decode(ctx, d_quant, d_rq_desc, d_unit_chains, d_st_desc, n_units, d_side, sample_rate_idx, d_overlap_in,
                             d_vvec_in, d_vfront_in, d_overlap_out, d_vvec_out, d_vfront_out, d_pcm, n_chains, granules_per_chain);
}

int symaccel_mp3_decode_device(symaccel_ctx *ctx, const int16_t *d_quant, const symaccel_mp3_requant *d_rq_desc,
                               const int32_t *d_unit_chains, const symaccel_mp3_stereo *d_st_desc, size_t n_units,
                               const symaccel_mp3_side *d_side, int sample_rate_idx, float *d_overlap_io, float *d_vvec_io,
                               int32_t *d_vfront_io, float *d_pcm, size_t n_chains, size_t granules_per_chain) {
    if (!ctx || sample_rate_idx < 0 || sample_rate_idx > 8) return SYMACCEL_ERR_INVALID_ARG;
    if (n_chains == 0 || granules_per_chain == 0) return SYMACCEL_OK;
    if (!d_overlap_io || !d_vvec_io || !d_vfront_io) return SYMACCEL_ERR_INVALID_ARG;
    DeviceGuard dev(ctx);
    if (!dev.ok()) return dev.status();
    const size_t ov_bytes = n_chains * 576 * 4, vv_bytes = n_chains * 1024 * 4, vf_bytes = n_chains * 4;
    void *scratch = nullptr;
    SYM_TRY(ctx_scratch(ctx, ov_bytes + vv_bytes + vf_bytes, &scratch));
    float *ov_out = (float *)scratch;
    float *vv_out = ov_out + n_chains * 576;
    int32_t *vf_out = (int32_t *)(vv_out + n_chains * 1024);
    SYM_TRY(symaccel_mp3_decode_pp_device(ctx, d_quant, d_rq_desc, d_unit_chains, d_st_desc, n_units, d_side, sample_rate_idx, d_overlap_io,
                                          d_vvec_io, d_vfront_io, ov_out, vv_out, vf_out, d_pcm, n_chains, granules_per_chain));
    SYM_TRY(launch_state_copy(ctx, d_overlap_io, ov_out, ov_bytes, d_vvec_io, vv_out, vv_bytes, d_vfront_io, vf_out, vf_bytes));
    return SYMACCEL_OK;
}

int symaccel_mp3_synth(symaccel_ctx *ctx, const float *h_xr, const symaccel_mp3_side *h_side, int sample_rate_idx,
                       float *h_overlap_io, float *h_vvec_io, int32_t *h_vfront_io, float *h_pcm, size_t n_chains,
                       size_t granules_per_chain) {
    if (!ctx || sample_rate_idx < 0 || sample_rate_idx > 8) return SYMACCEL_ERR_INVALID_ARG;
    if (n_chains == 0 || granules_per_chain == 0) return SYMACCEL_OK;
    if (!h_xr || !h_side || !h_overlap_io || !h_vvec_io || !h_vfront_io || !h_pcm) return SYMACCEL_ERR_INVALID_ARG;
    if (n_chains * granules_per_chain * 2304 >= kPipelineBytes && granules_per_chain >= 16)
        return symaccel_mp3_synth_pipelined(ctx, h_xr, h_side, sample_rate_idx, h_overlap_io, h_vvec_io, h_vfront_io, h_pcm, n_chains,
                                            granules_per_chain, 0);
    DeviceGuard dev(ctx);
    if (!dev.ok()) return dev.status();
    const size_t ng = n_chains * granules_per_chain;
    DevBuf xr(ctx), side(ctx), ov(ctx), vv(ctx), vf(ctx), pcm(ctx);
    SYM_TRY(xr.from_host(h_xr, ng * 576 * 4));
    SYM_TRY(side.from_host(h_side, ng * sizeof(symaccel_mp3_side)));
    SYM_TRY(ov.from_host(h_overlap_io, n_chains * 576 * 4));
    SYM_TRY(vv.from_host(h_vvec_io, n_chains * 1024 * 4));
    SYM_TRY(vf.from_host(h_vfront_io, n_chains * 4));
    SYM_TRY(pcm.alloc(ng * 576 * 4));
    SYM_TRY(symaccel_mp3_synth_device(ctx, (const float *)xr.p, (const symaccel_mp3_side *)side.p, sample_rate_idx,
                                      (float *)ov.p, (float *)vv.p, (int32_t *)vf.p, (float *)pcm.p, n_chains,
                                      granules_per_chain));
    SYM_TRY(pcm.to_host(h_pcm, ng * 576 * 4));
    SYM_TRY(ov.to_host(h_overlap_io, n_chains * 576 * 4));
    SYM_TRY(vv.to_host(h_vvec_io, n_chains * 1024 * 4));
    SYM_TRY(vf.to_host(h_vfront_io, n_chains * 4));
    return symaccel_sync(ctx);
}

// ---- Vorbis -------------------------------------------------------------------------------

static int vorbis_check(symaccel_ctx *ctx, int bs0_exp, int bs1_exp) {
    // vorbis/lib.rs:404-406, 461-470: block sizes 2^6..2^13, bs0 <= bs1
    if (!ctx || bs0_exp < 6 || bs1_exp > 13 || bs0_exp > bs1_exp) return SYMACCEL_ERR_INVALID_ARG;
    return SYMACCEL_OK;
}
// What can be said about the strides of a device-pointer call without reading the (device-resident) flags: even a chain
// of short blocks only needs blocks * bs0/2 lines.  The exact check is the host-pointer entry point's (and the caller's).
static int vorbis_stride_floor(int bs0_exp, size_t spec_stride, size_t blocks_per_chain) {
    return spec_stride < blocks_per_chain * ((size_t)1 << (bs0_exp - 1)) ? SYMACCEL_ERR_INVALID_ARG : SYMACCEL_OK;
}

static int vorbis_synth_device(symaccel_ctx *ctx, int bs0_exp, int bs1_exp, const float *d_spectra, const float *d_residue,
                                 size_t spec_stride, const uint8_t *d_block_flag, int32_t *d_prev_flag_io,
                                 float *d_overlap_io, float *d_pcm, size_t pcm_stride, size_t n_chains,
                                 size_t blocks_per_chain, const uint8_t *d_floor_y = nullptr) {
    SYM_TRY(vorbis_check(ctx, bs0_exp, bs1_exp));
    if (n_chains == 0 || blocks_per_chain == 0) return SYMACCEL_OK;
    if (!d_spectra || !d_block_flag || !d_prev_flag_io || !d_overlap_io || !d_pcm) return SYMACCEL_ERR_INVALID_ARG;
    SYM_TRY(vorbis_stride_floor(bs0_exp, spec_stride, blocks_per_chain));
    DeviceGuard dev(ctx);
    if (!dev.ok()) return dev.status();
    const size_t half1 = (size_t)1 << (bs1_exp - 1);
    const size_t ov_bytes = n_chains * half1 * 4, pf_bytes = n_chains * 4;
    void *scratch = nullptr;
    const size_t off_bytes = n_chains * (blocks_per_chain + 1) * 2 * sizeof(uint32_t);
    SYM_TRY(ctx_scratch(ctx, ov_bytes + pf_bytes + 512 + off_bytes, &scratch));
    float *ov_out = (float *)scratch;
    int32_t *pf_out = (int32_t *)(ov_out + n_chains * half1);
    SYM_TRY(launch_vorbis(ctx, bs0_exp, bs1_exp, d_spectra, d_residue, spec_stride, d_block_flag, d_prev_flag_io, pf_out,
                          d_overlap_io, ov_out, d_pcm, pcm_stride, n_chains, blocks_per_chain, (char *)scratch + ov_bytes + pf_bytes, d_floor_y));
    SYM_TRY(launch_state_copy(ctx, d_overlap_io, ov_out, ov_bytes, d_prev_flag_io, pf_out, pf_bytes, nullptr, nullptr, 0));
    return SYMACCEL_OK;
}

static int vorbis_synth_pp(symaccel_ctx *ctx, int bs0_exp, int bs1_exp, const float *d_spectra, const float *d_residue,
                           size_t spec_stride, const uint8_t *d_block_flag, const int32_t *d_prev_flag_in,
                           int32_t *d_prev_flag_out, const float *d_overlap_in, float *d_overlap_out, float *d_pcm,
                           size_t pcm_stride, size_t n_chains, size_t blocks_per_chain, const uint8_t *d_floor_y) {
    SYM_TRY(vorbis_check(ctx, bs0_exp, bs1_exp));
    if (n_chains == 0 || blocks_per_chain == 0) return SYMACCEL_OK;
    if (!d_spectra || !d_block_flag || !d_prev_flag_in || !d_prev_flag_out || !d_overlap_in || !d_overlap_out || !d_pcm)
        return SYMACCEL_ERR_INVALID_ARG;
    if (d_prev_flag_in == d_prev_flag_out || d_overlap_in == d_overlap_out) return SYMACCEL_ERR_INVALID_ARG;
    SYM_TRY(vorbis_stride_floor(bs0_exp, spec_stride, blocks_per_chain));
    DeviceGuard dev(ctx);
    if (!dev.ok()) return dev.status();
    void *scratch = nullptr;  // per-block offsets of the generic block-size pairs
    SYM_TRY(ctx_scratch(ctx, n_chains * (blocks_per_chain + 1) * 2 * sizeof(uint32_t) + 256, &scratch));
    return launch_vorbis(ctx, bs0_exp, bs1_exp, d_spectra, d_residue, spec_stride, d_block_flag, d_prev_flag_in, d_prev_flag_out,
                         d_overlap_in, d_overlap_out, d_pcm, pcm_stride, n_chains, blocks_per_chain, scratch, d_floor_y);
}

int symaccel_vorbis_synth_pp_device(symaccel_ctx *ctx, int bs0_exp, int bs1_exp, const float *d_spectra, const float *d_residue,
                                    size_t spec_stride, const uint8_t *d_block_flag, const int32_t *d_prev_flag_in,
                                    int32_t *d_prev_flag_out, const float *d_overlap_in, float *d_overlap_out, float *d_pcm,
                                    size_t pcm_stride, size_t n_chains, size_t blocks_per_chain) {
    return vorbis_synth_pp(ctx, bs0_exp, bs1_exp, d_spectra, d_residue, spec_stride, d_block_flag, d_prev_flag_in, d_prev_flag_out,
                           d_overlap_in, d_overlap_out, d_pcm, pcm_stride, n_chains, blocks_per_chain, nullptr);
}

// floor curve as table indices (one byte per line) x residue, multiplied in the synthesis kernels' load path
static int vorbis_fy_check(const uint8_t *d_floor_y, const float *d_residue, size_t spec_stride, size_t pcm_stride, size_t n_chains,
                           size_t blocks_per_chain) {
    if (n_chains == 0 || blocks_per_chain == 0) return SYMACCEL_OK;
    if (!d_floor_y || !d_residue) return SYMACCEL_ERR_INVALID_ARG;
    // 16-byte accesses on the residue and the PCM, 4-byte accesses on the byte plane, every chain aligned alike
    if (spec_stride % 4 || pcm_stride % 4 || ((uintptr_t)d_floor_y | (uintptr_t)d_residue) % 16) return SYMACCEL_ERR_INVALID_ARG;
    return SYMACCEL_OK;
}
int symaccel_vorbis_synth_fy_pp_device(symaccel_ctx *ctx, int bs0_exp, int bs1_exp, const uint8_t *d_floor_y, const float *d_residue,
                                       size_t spec_stride, const uint8_t *d_block_flag, const int32_t *d_prev_flag_in,
                                       int32_t *d_prev_flag_out, const float *d_overlap_in, float *d_overlap_out, float *d_pcm,
                                       size_t pcm_stride, size_t n_chains, size_t blocks_per_chain) {
    SYM_TRY(vorbis_fy_check(d_floor_y, d_residue, spec_stride, pcm_stride, n_chains, blocks_per_chain));
    return vorbis_synth_pp(ctx, bs0_exp, bs1_exp, d_residue, nullptr, spec_stride, d_block_flag, d_prev_flag_in, d_prev_flag_out,
                           d_overlap_in, d_overlap_out, d_pcm, pcm_stride, n_chains, blocks_per_chain, d_floor_y);
}
int symaccel_vorbis_synth_fy_device(symaccel_ctx *ctx, int bs0_exp, int bs1_exp, const uint8_t *d_floor_y, const float *d_residue,
                                    size_t spec_stride, const uint8_t *d_block_flag, int32_t *d_prev_flag_io, float *d_overlap_io,
                                    float *d_pcm, size_t pcm_stride, size_t n_chains, size_t blocks_per_chain) {
    SYM_TRY(vorbis_fy_check(d_floor_y, d_residue, spec_stride, pcm_stride, n_chains, blocks_per_chain));
    return vorbis_synth_device(ctx, bs0_exp, bs1_exp, d_residue, nullptr, spec_stride, d_block_flag, d_prev_flag_io, d_overlap_io,
                               d_pcm, pcm_stride, n_chains, blocks_per_chain, d_floor_y);
}

int symaccel_vorbis_synth_device(symaccel_ctx *ctx, int bs0_exp, int bs1_exp, const float *d_spectra,
                                 size_t spec_stride, const uint8_t *d_block_flag, int32_t *d_prev_flag_io,
                                 float *d_overlap_io, float *d_pcm, size_t pcm_stride, size_t n_chains,
                                 size_t blocks_per_chain) {
    return vorbis_synth_device(ctx, bs0_exp, bs1_exp, d_spectra, nullptr, spec_stride, d_block_flag, d_prev_flag_io,
                               d_overlap_io, d_pcm, pcm_stride, n_chains, blocks_per_chain);
}

int symaccel_vorbis_synth_fr_device(symaccel_ctx *ctx, int bs0_exp, int bs1_exp, const float *d_floor,
                                    const float *d_residue, size_t spec_stride, const uint8_t *d_block_flag,
                                    int32_t *d_prev_flag_io, float *d_overlap_io, float *d_pcm, size_t pcm_stride,
                                    size_t n_chains, size_t blocks_per_chain) {
    if (!d_residue) return SYMACCEL_ERR_INVALID_ARG;
    return vorbis_synth_device(ctx, bs0_exp, bs1_exp, d_floor, d_residue, spec_stride, d_block_flag, d_prev_flag_io,
                               d_overlap_io, d_pcm, pcm_stride, n_chains, blocks_per_chain);
}

int symaccel_vorbis_synth(symaccel_ctx *ctx, int bs0_exp, int bs1_exp, const float *h_spectra, size_t spec_stride,
                          const uint8_t *h_block_flag, int32_t *h_prev_flag_io, float *h_overlap_io, float *h_pcm,
                          size_t pcm_stride, size_t n_chains, size_t blocks_per_chain) {
    if (!ctx || bs0_exp < 6 || bs1_exp > 13 || bs0_exp > bs1_exp) return SYMACCEL_ERR_INVALID_ARG;
    if (n_chains == 0 || blocks_per_chain == 0) return SYMACCEL_OK;
    if (!h_spectra || !h_block_flag || !h_prev_flag_io || !h_overlap_io || !h_pcm) return SYMACCEL_ERR_INVALID_ARG;
    // The flags are on the host here: check the packed layout they imply against the caller's strides before anything
    // is launched (a block of n samples reads n/2 lines and owns (prev_n + n)/4 PCM slots -- see the layout note in
    // symaccel.h; a first block without a previous one owns n/2 slots and leaves them untouched).  Strides that are too
    // small would make one chain's blocks overwrite the next chain's.
    for (size_t c = 0; c < n_chains; ++c) {
        size_t lines = 0, samples = 0;
        int prev = h_prev_flag_io[c];  // -1: no previous block (lapping state empty)
        for (size_t b = 0; b < blocks_per_chain; ++b) {
            const int flag = h_block_flag[c * blocks_per_chain + b] ? 1 : 0;
            const size_t n = (size_t)1 << (flag ? bs1_exp : bs0_exp);
            lines += n / 2;
            samples += (prev >= 0 ? (((size_t)1 << (prev ? bs1_exp : bs0_exp)) + n) / 4 : n / 2);
            prev = flag;
        }
        if (lines > spec_stride || samples > pcm_stride) return SYMACCEL_ERR_INVALID_ARG;
    }
    DeviceGuard dev(ctx);
    if (!dev.ok()) return dev.status();
    const size_t half1 = (size_t)1 << (bs1_exp - 1);
    DevBuf sp(ctx), bf(ctx), pf(ctx), ov(ctx), pcm(ctx);
    SYM_TRY(sp.from_host(h_spectra, n_chains * spec_stride * 4));
    SYM_TRY(bf.from_host(h_block_flag, n_chains * blocks_per_chain));
    SYM_TRY(pf.from_host(h_prev_flag_io, n_chains * 4));
    SYM_TRY(ov.from_host(h_overlap_io, n_chains * half1 * 4));
    SYM_TRY(pcm.alloc(n_chains * pcm_stride * 4));
    SYM_GPU(ctx, hipMemsetAsync(pcm.p, 0, n_chains * pcm_stride * 4, ctx->stream));
    SYM_TRY(symaccel_vorbis_synth_device(ctx, bs0_exp, bs1_exp, (const float *)sp.p, spec_stride,
                                         (const uint8_t *)bf.p, (int32_t *)pf.p, (float *)ov.p, (float *)pcm.p,
                                         pcm_stride, n_chains, blocks_per_chain));
    SYM_TRY(pcm.to_host(h_pcm, n_chains * pcm_stride * 4));
    SYM_TRY(ov.to_host(h_overlap_io, n_chains * half1 * 4));
    SYM_TRY(pf.to_host(h_prev_flag_io, n_chains * 4));
    return symaccel_sync(ctx);
}

int symaccel_vorbis_decode(symaccel_ctx *ctx, int bs0_exp, int bs1_exp, const float *h_residue, size_t spec_stride,
                           const uint8_t *h_block_flag, const uint8_t *h_floor, const uint32_t *h_posts, size_t posts_stride,
                           const symaccel_vorbis_floor1_cfg *h_floors, size_t n_floors, size_t channels_per_stream,
                           const uint8_t *h_coupling, const uint32_t *h_coupling_first, int32_t *h_prev_flag_io, float *h_overlap_io,
                           float *h_pcm, size_t pcm_stride, size_t n_chains, size_t blocks_per_chain) {
    SYM_TRY(vorbis_check(ctx, bs0_exp, bs1_exp));
    if (n_chains == 0 || blocks_per_chain == 0) return SYMACCEL_OK;
    if (!h_residue || !h_block_flag || !h_floor || !h_prev_flag_io || !h_overlap_io || !h_pcm || !h_coupling_first) return SYMACCEL_ERR_INVALID_ARG;
    const size_t cps = channels_per_stream, nb = blocks_per_chain;
    if (cps == 0 || cps > 255 || n_chains % cps || n_floors > 255 || (n_floors && (!h_floors || !h_posts))) return SYMACCEL_ERR_INVALID_ARG;
    if (spec_stride % 4 || pcm_stride % 4) return SYMACCEL_ERR_INVALID_ARG;
    const size_t n_streams = n_chains / cps;
    for (size_t f = 0; f < n_floors; ++f)
        if (h_floors[f].n_posts < 2 || h_floors[f].n_posts > 65 || h_floors[f].n_posts > posts_stride || h_floors[f].multiplier < 1 ||
            h_floors[f].multiplier > 4)
            return SYMACCEL_ERR_INVALID_ARG;
    // Everything that describes the batch is in host memory: check it.  The channels of a stream share their block flags and
    // their previous flag (one mode per packet, lib.rs:170-178); the packed layout must fit the strides (as symaccel_vorbis_synth);
    // a floor index names a configuration; coupling steps name two different channels of the stream, in range.
    std::vector<uint32_t> block_off(n_streams * (nb + 1));
    for (size_t st = 0; st < n_streams; ++st) {
        const size_t c0 = st * cps;
        for (size_t c = c0; c < c0 + cps; ++c) {
            if (h_prev_flag_io[c] != h_prev_flag_io[c0]) return SYMACCEL_ERR_INVALID_ARG;
            for (size_t b = 0; b < nb; ++b)
                if ((h_block_flag[c * nb + b] != 0) != (h_block_flag[c0 * nb + b] != 0)) return SYMACCEL_ERR_INVALID_ARG;
        }
        size_t lines = 0, samples = 0;
        int prev = h_prev_flag_io[c0];
        for (size_t b = 0; b < nb; ++b) {
            const int flag = h_block_flag[c0 * nb + b] ? 1 : 0;
            const size_t n = (size_t)1 << (flag ? bs1_exp : bs0_exp);
            block_off[st * (nb + 1) + b] = (uint32_t)lines;
            lines += n / 2;
            samples += (prev >= 0 ? (((size_t)1 << (prev ? bs1_exp : bs0_exp)) + n) / 4 : n / 2);
            prev = flag;
        }
        block_off[st * (nb + 1) + nb] = (uint32_t)lines;
        if (lines > spec_stride || samples > pcm_stride || lines > 0xffffffffu) return SYMACCEL_ERR_INVALID_ARG;
    }
    if (n_chains * spec_stride > 0xffffffffu) return SYMACCEL_ERR_INVALID_ARG;  // (line offsets into the plane are 32-bit)
    const size_t n_sb = n_streams * nb;
    if (h_coupling_first[0] != 0) return SYMACCEL_ERR_INVALID_ARG;
    for (size_t i = 0; i < n_sb; ++i)
        if (h_coupling_first[i + 1] < h_coupling_first[i]) return SYMACCEL_ERR_INVALID_ARG;
    const size_t n_steps = h_coupling_first[n_sb];
    if (n_steps && !h_coupling) return SYMACCEL_ERR_INVALID_ARG;
    for (size_t s = 0; s < n_steps; ++s)
        if (h_coupling[2 * s] >= cps || h_coupling[2 * s + 1] >= cps || h_coupling[2 * s] == h_coupling[2 * s + 1]) return SYMACCEL_ERR_INVALID_ARG;
    // the channel-blocks of every (floor configuration, block size) class: their floor1_Y rows and where their lines are
    std::vector<std::vector<uint32_t>> ys(n_floors * 2), offs(n_floors * 2);
    std::vector<uint8_t> kill(n_chains * nb, 0);
    bool any_kill = false;
    for (size_t c = 0; c < n_chains; ++c) {
        const size_t st = c / cps;
        for (size_t b = 0; b < nb; ++b) {
            const unsigned f = h_floor[c * nb + b];
            if (f == SYMACCEL_VORBIS_FLOOR_UNUSED) {
                kill[c * nb + b] = 1;
                any_kill = true;
                continue;
            }
            if (f >= n_floors) return SYMACCEL_ERR_INVALID_ARG;
            const size_t k = 2 * f + (h_block_flag[c * nb + b] ? 1 : 0);
            const uint32_t *y = h_posts + (c * nb + b) * posts_stride;
            for (unsigned i = 0; i < h_floors[f].n_posts; ++i) {
                if (y[i] > 511u) return SYMACCEL_ERR_UNSUPPORTED;  // (symaccel_vorbis_floor1_status_device's domain)
                ys[k].push_back(y[i]);
            }
            offs[k].push_back((uint32_t)(c * spec_stride + block_off[st * (nb + 1) + b]));
        }
    }
    DeviceGuard dev(ctx);
    if (!dev.ok()) return dev.status();
    const size_t half1 = (size_t)1 << (bs1_exp - 1);
    DevBuf res(ctx), bf(ctx), pf(ctx), ov(ctx), pcm(ctx), plane(ctx), d_off(ctx), d_steps(ctx), d_first(ctx), d_kill(ctx);
    SYM_TRY(res.from_host(h_residue, n_chains * spec_stride * 4));
    SYM_TRY(bf.from_host(h_block_flag, n_chains * nb));
    SYM_TRY(pf.from_host(h_prev_flag_io, n_chains * 4));
    SYM_TRY(ov.from_host(h_overlap_io, n_chains * half1 * 4));
    SYM_TRY(pcm.alloc(n_chains * pcm_stride * 4));
    SYM_TRY(plane.alloc(n_chains * spec_stride));
    SYM_GPU(ctx, hipMemsetAsync(pcm.p, 0, n_chains * pcm_stride * 4, ctx->stream));
    SYM_GPU(ctx, hipMemsetAsync(plane.p, 0, n_chains * spec_stride, ctx->stream));
    if (n_steps || any_kill) {  // lib.rs:250-278 + the zero floors of lib.rs:206-209 under lib.rs:289-291
        SYM_TRY(d_off.from_host(block_off.data(), block_off.size() * 4));
        SYM_TRY(d_first.from_host(h_coupling_first, (n_sb + 1) * 4));
        SYM_TRY(d_kill.from_host(kill.data(), kill.size()));
        if (n_steps) SYM_TRY(d_steps.from_host(h_coupling, n_steps * 2));
        SYM_TRY(launch_vorbis_prepare(ctx, (float *)res.p, spec_stride, (unsigned)cps, n_streams, nb, (const uint32_t *)d_off.p,
                                      (const uint8_t *)d_steps.p, (const uint32_t *)d_first.p, (const uint8_t *)d_kill.p));
    }
    // floor.rs:568-653 + 776-825 as one byte per line: the (floor, class) pairs are jobs, two per launch
    std::vector<DevBuf> keep;
    keep.reserve(4 * n_floors);
    std::vector<symaccel_vorbis_floor1_job> jobs;
    for (size_t k = 0; k < 2 * n_floors; ++k) {
        if (offs[k].empty()) continue;
        const symaccel_vorbis_floor1_cfg &cfg = h_floors[k / 2];
        keep.emplace_back(ctx);
        DevBuf &dy = keep.back();
        SYM_TRY(dy.from_host(ys[k].data(), ys[k].size() * 4));
        keep.emplace_back(ctx);
        DevBuf &dof = keep.back();
        SYM_TRY(dof.from_host(offs[k].data(), offs[k].size() * 4));
        const uint32_t n2 = (uint32_t)1 << ((k & 1 ? bs1_exp : bs0_exp) - 1);
        jobs.push_back(symaccel_vorbis_floor1_job{cfg.x_list, cfg.n_posts, cfg.multiplier, (const uint32_t *)dy.p, n2, (const uint32_t *)dof.p, offs[k].size()});
    }
    SYM_TRY(symaccel_vorbis_floor1_y_jobs_device(ctx, jobs.data(), jobs.size(), (uint8_t *)plane.p));
    // lib.rs:282-292 (one rounded multiply per line, in the load path) + dsp.rs:68-126
    SYM_TRY(symaccel_vorbis_synth_fy_device(ctx, bs0_exp, bs1_exp, (const uint8_t *)plane.p, (const float *)res.p, spec_stride,
                                            (const uint8_t *)bf.p, (int32_t *)pf.p, (float *)ov.p, (float *)pcm.p, pcm_stride, n_chains, nb));
    SYM_TRY(pcm.to_host(h_pcm, n_chains * pcm_stride * 4));
    SYM_TRY(ov.to_host(h_overlap_io, n_chains * half1 * 4));
    SYM_TRY(pf.to_host(h_prev_flag_io, n_chains * 4));
    return symaccel_sync(ctx);
}

int symaccel_vorbis_inverse_coupling_device(symaccel_ctx *ctx, float *d_residue, size_t n, const uint32_t *mag_index,
                                            const uint32_t *ang_index, size_t n_pairs) {
    if (!ctx || n_pairs > 256) return SYMACCEL_ERR_INVALID_ARG;
    if (n == 0 || n_pairs == 0) return SYMACCEL_OK;
    if (!d_residue || !mag_index || !ang_index) return SYMACCEL_ERR_INVALID_ARG;
    DeviceGuard dev(ctx);
    if (!dev.ok()) return dev.status();
    for (size_t p = 0; p < n_pairs; ++p) {  // ordered: coupling steps may chain (lib.rs:252)
        if (mag_index[p] == ang_index[p]) return SYMACCEL_ERR_INVALID_ARG;  // lib.rs:253 debug_assert
        SYM_TRY(launch_vorbis_coupling(ctx, d_residue + (size_t)mag_index[p] * n, d_residue + (size_t)ang_index[p] * n, n));
    }
    return SYMACCEL_OK;
}

int symaccel_vorbis_dot_product_device(symaccel_ctx *ctx, float *d_floor, const float *d_residue, size_t total) {
    if (!ctx) return SYMACCEL_ERR_INVALID_ARG;
    if (total == 0) return SYMACCEL_OK;
    if (!d_floor || !d_residue) return SYMACCEL_ERR_INVALID_ARG;
    DeviceGuard dev(ctx);
    if (!dev.ok()) return dev.status();
    return launch_vorbis_dot(ctx, d_floor, d_residue, total);
}

int symaccel_vorbis_deinterleave2_device(symaccel_ctx *ctx, const float *d_type2, float *d_planar, int n_ch,
                                         size_t n2, size_t count) {
    if (!ctx || n_ch < 1 || n_ch > 32) return SYMACCEL_ERR_INVALID_ARG;  // lib.rs:439-441
    if (n2 == 0 || count == 0) return SYMACCEL_OK;
    if (!d_type2 || !d_planar || d_type2 == d_planar) return SYMACCEL_ERR_INVALID_ARG;
    DeviceGuard dev(ctx);
    if (!dev.ok()) return dev.status();
    return launch_vorbis_deinterleave(ctx, d_type2, d_planar, n_ch, n2, count);
}

// Setup-time derivations the reference does once per floor (floor.rs:540-555, 748-773): neighbours of every post and the x-sorted
// visiting order; with the argument checks every floor-1 entry point shares.  setup[65 * 4]: x | low neighbour | high neighbour | sort order
static int vorbis_floor1_setup(const uint32_t *x_list, int n_posts, int multiplier, uint32_t n, uint32_t *setup) {
    if (n_posts < 2 || n_posts > 65 || multiplier < 1 || multiplier > 4 || !x_list) return SYMACCEL_ERR_INVALID_ARG;
    std::memset(setup, 0, 65 * 4 * sizeof(uint32_t));
    for (int x = 0; x < n_posts; ++x) {
        uint32_t bound = x_list[x], low = 0, high = 0xffffffffu, rl = 0, rh = 0;
        for (int i = 0; i < x; ++i) {
            const uint32_t xv = x_list[i];
            if (xv > low && xv < bound) { low = xv; rl = (uint32_t)i; }
            if (xv < high && xv > bound) { high = xv; rh = (uint32_t)i; }
        }
        setup[x] = x_list[x];
        setup[65 + x] = rl;
        setup[130 + x] = rh;
        setup[195 + x] = (uint32_t)x;
    }
    for (int i = 1; i < n_posts; ++i) {  // stable, like sort_by_key
        const uint32_t k = setup[195 + i];
        int j = i - 1;
        while (j >= 0 && x_list[setup[195 + j]] > x_list[k]) { setup[195 + j + 1] = setup[195 + j]; --j; }
        setup[195 + j + 1] = k;
    }
    for (int i = 1; i < n_posts; ++i)  // render_line divides by (x1 - x0): equal x would panic in the reference
        if (x_list[setup[195 + i]] == x_list[setup[195 + i - 1]]) return SYMACCEL_ERR_INVALID_ARG;
    for (int i = 0; i < n_posts; ++i)
        if (x_list[i] > 0xffffu) return SYMACCEL_ERR_INVALID_ARG;  // floor1_X values have at most 15 bits (rangebits)
    if (n > 4096u || (n & 15u)) return SYMACCEL_ERR_INVALID_ARG;  // n = blocksize / 2, blocksize = 2^6 .. 2^13 (lib.rs:404-406)
    return SYMACCEL_OK;
}

static int vorbis_floor1(symaccel_ctx *ctx, const uint32_t *x_list, int n_posts, int multiplier, const uint32_t *d_y, uint32_t n,
                         float *d_floor, size_t count, const float *d_residue, uint8_t *d_floor_y = nullptr,
                         const uint32_t *d_line_offs = nullptr) {
    if (!ctx || n_posts < 2 || n_posts > 65 || multiplier < 1 || multiplier > 4) return SYMACCEL_ERR_INVALID_ARG;
    if (count == 0 || n == 0) return SYMACCEL_OK;
    if (!x_list || !d_y || (!d_floor && !d_floor_y)) return SYMACCEL_ERR_INVALID_ARG;
    DeviceGuard dev(ctx);
    if (!dev.ok()) return dev.status();
    uint32_t setup[65 * 4];
    SYM_TRY(vorbis_floor1_setup(x_list, n_posts, multiplier, n, setup));
    // the derived tables travel as a kernel argument: no staging copy, no stream synchronisation
    return launch_vorbis_floor1(ctx, setup, n_posts, multiplier, d_y, n, d_floor, count, d_residue, d_floor_y, d_line_offs);
}

int symaccel_vorbis_floor1_device(symaccel_ctx *ctx, const uint32_t *x_list, int n_posts, int multiplier,
                                  const uint32_t *d_y, uint32_t n, float *d_floor, size_t count) {
    return vorbis_floor1(ctx, x_list, n_posts, multiplier, d_y, n, d_floor, count, nullptr);
}
int symaccel_vorbis_floor1_dot_device(symaccel_ctx *ctx, const uint32_t *x_list, int n_posts, int multiplier,
                                      const uint32_t *d_y, uint32_t n, const float *d_residue, float *d_spectrum, size_t count) {
    if (count != 0 && n != 0 && !d_residue) return SYMACCEL_ERR_INVALID_ARG;
    return vorbis_floor1(ctx, x_list, n_posts, multiplier, d_y, n, d_spectrum, count, d_residue);
}
int symaccel_vorbis_floor1_dot_at_device(symaccel_ctx *ctx, const uint32_t *x_list, int n_posts, int multiplier, const uint32_t *d_y,
                                         uint32_t n, const uint32_t *d_line_offsets, const float *d_residue, float *d_spectrum,
                                         size_t count) {
    if (count != 0 && n != 0 && (!d_residue || !d_line_offsets)) return SYMACCEL_ERR_INVALID_ARG;
    return vorbis_floor1(ctx, x_list, n_posts, multiplier, d_y, n, d_spectrum, count, d_residue, nullptr, d_line_offsets);
}
int symaccel_vorbis_floor1_y_device(symaccel_ctx *ctx, const uint32_t *x_list, int n_posts, int multiplier, const uint32_t *d_y,
                                    uint32_t n, const uint32_t *d_line_offsets, uint8_t *d_floor_y, size_t count) {
    if (count != 0 && n != 0 && (!d_floor_y || ((uintptr_t)d_floor_y & 3u))) return SYMACCEL_ERR_INVALID_ARG;
    return vorbis_floor1(ctx, x_list, n_posts, multiplier, d_y, n, nullptr, count, nullptr, d_floor_y, d_line_offsets);
}

int symaccel_vorbis_floor1_y_jobs_device(symaccel_ctx *ctx, const symaccel_vorbis_floor1_job *jobs, size_t n_jobs, uint8_t *d_floor_y) {
    if (!ctx || (n_jobs && !jobs)) return SYMACCEL_ERR_INVALID_ARG;
    // every job is checked before anything is launched
    std::vector<const symaccel_vorbis_floor1_job *> live;
    for (size_t k = 0; k < n_jobs; ++k) {
        const symaccel_vorbis_floor1_job &j = jobs[k];
        if (j.n_posts < 2 || j.n_posts > 65 || j.multiplier < 1 || j.multiplier > 4) return SYMACCEL_ERR_INVALID_ARG;
        if (j.count == 0 || j.n == 0) continue;
        if (!j.x_list || !j.d_y || !d_floor_y || ((uintptr_t)d_floor_y & 3u)) return SYMACCEL_ERR_INVALID_ARG;
        live.push_back(&j);
    }
    if (live.empty()) return SYMACCEL_OK;
    DeviceGuard dev(ctx);
    if (!dev.ok()) return dev.status();
    std::vector<uint32_t> setups(live.size() * 65 * 4);
    for (size_t k = 0; k < live.size(); ++k)
        SYM_TRY(vorbis_floor1_setup(live[k]->x_list, live[k]->n_posts, live[k]->multiplier, live[k]->n, setups.data() + k * 65 * 4));
    size_t k = 0;
    for (; k + 1 < live.size(); k += 2) {  // two jobs per grid
        const symaccel_vorbis_floor1_job &a = *live[k], &b = *live[k + 1];
        const uint32_t *const su[2] = {setups.data() + k * 65 * 4, setups.data() + (k + 1) * 65 * 4};
        const int np[2] = {a.n_posts, b.n_posts}, mu[2] = {a.multiplier, b.multiplier};
        const uint32_t *const dy[2] = {a.d_y, b.d_y};
        const uint32_t nn[2] = {a.n, b.n};
        const size_t cn[2] = {a.count, b.count};
        const uint32_t *const lo[2] = {a.d_line_offsets, b.d_line_offsets};
        SYM_TRY(launch_vorbis_floor1_pair(ctx, su, np, mu, dy, nn, cn, lo, d_floor_y));
    }
    if (k < live.size()) {
        const symaccel_vorbis_floor1_job &a = *live[k];
        SYM_TRY(launch_vorbis_floor1(ctx, setups.data() + k * 65 * 4, a.n_posts, a.multiplier, a.d_y, a.n, nullptr, a.count, nullptr, d_floor_y, a.d_line_offsets));
    }
    return SYMACCEL_OK;
}

// ---- FLAC ---------------------------------------------------------------------------------

int symaccel_flac_restore_device(symaccel_ctx *ctx, int32_t *d_buf, const symaccel_flac_desc *d_desc,
                                 const int32_t *d_coeffs, size_t n_blocks, size_t blocksize) {
    if (!ctx || blocksize > 65535) return SYMACCEL_ERR_INVALID_ARG;  // frame.rs:58 (u16 block size)
    if (n_blocks == 0 || blocksize == 0) return SYMACCEL_OK;
    if (!d_buf || !d_desc || !d_coeffs) return SYMACCEL_ERR_INVALID_ARG;
    DeviceGuard dev(ctx);
    if (!dev.ok()) return dev.status();
    return launch_flac_restore(ctx, d_buf, d_desc, d_coeffs, n_blocks, blocksize);
}

int symaccel_flac_restore(symaccel_ctx *ctx, int32_t *h_buf, const symaccel_flac_desc *h_desc, const int32_t *h_coeffs,
                          size_t n_blocks, size_t blocksize) {
    if (!ctx || blocksize > 65535) return SYMACCEL_ERR_INVALID_ARG;
    if (n_blocks == 0 || blocksize == 0) return SYMACCEL_OK;
    if (!h_buf || !h_desc || !h_coeffs) return SYMACCEL_ERR_INVALID_ARG;
    for (size_t b = 0; b < n_blocks; ++b) {  // decoder.rs:361, 456-458, 506-508
        const symaccel_flac_desc &d = h_desc[b];
        if (d.kind > SYMACCEL_FLAC_LPC || d.order > blocksize || d.shift > 31 || d.wasted_bits > 31) return SYMACCEL_ERR_INVALID_ARG;
        if (d.kind == SYMACCEL_FLAC_FIXED && d.order > 4) return SYMACCEL_ERR_INVALID_ARG;
        if (d.kind == SYMACCEL_FLAC_LPC && (d.order < 1 || d.order > 32)) return SYMACCEL_ERR_INVALID_ARG;
    }
    if (n_blocks * blocksize * 4 >= kPipelineBytes && n_blocks >= 256)
        return symaccel_flac_restore_pipelined(ctx, h_buf, h_desc, h_coeffs, n_blocks, blocksize, 0);
    DeviceGuard dev(ctx);
    if (!dev.ok()) return dev.status();
    DevBuf buf(ctx), desc(ctx), co(ctx);
    SYM_TRY(buf.from_host(h_buf, n_blocks * blocksize * 4));
    SYM_TRY(desc.from_host(h_desc, n_blocks * sizeof(symaccel_flac_desc)));
    SYM_TRY(co.from_host(h_coeffs, n_blocks * 32 * 4));
    SYM_TRY(symaccel_flac_restore_device(ctx, (int32_t *)buf.p, (const symaccel_flac_desc *)desc.p,
                                         (const int32_t *)co.p, n_blocks, blocksize));
    SYM_TRY(buf.to_host(h_buf, n_blocks * blocksize * 4));
    return symaccel_sync(ctx);
}

int symaccel_flac_restore_stereo_device(symaccel_ctx *ctx, int32_t *d_buf, const symaccel_flac_desc *d_desc,
                                        const int32_t *d_coeffs, const uint8_t *d_pair_mode, uint32_t out_shift,
                                        size_t n_blocks, size_t blocksize) {
    if (!ctx || blocksize > 0xffffffffu || out_shift > 31 || (n_blocks & 1)) return SYMACCEL_ERR_INVALID_ARG;
    if (n_blocks == 0 || blocksize == 0) return SYMACCEL_OK;
    if (!d_buf || !d_desc || !d_coeffs || !d_pair_mode) return SYMACCEL_ERR_INVALID_ARG;
    DeviceGuard dev(ctx);
    if (!dev.ok()) return dev.status();
    return launch_flac_restore(ctx, d_buf, d_desc, d_coeffs, n_blocks, blocksize, d_pair_mode, out_shift);
}

int symaccel_flac_restore_strided_device(symaccel_ctx *ctx, int32_t *d_buf, const symaccel_flac_desc *d_desc,
                                         const int32_t *d_coeffs, const uint8_t *d_pair_mode, uint32_t out_shift, size_t n_blocks,
                                         size_t blocksize, size_t stride) {
    if (stride == 0) stride = blocksize;
    if (!ctx || blocksize > 65535 || stride < blocksize || stride > 0xffffffffu || out_shift > 31) return SYMACCEL_ERR_INVALID_ARG;
    if (d_pair_mode && (n_blocks & 1)) return SYMACCEL_ERR_INVALID_ARG;
    if (n_blocks == 0 || blocksize == 0) return SYMACCEL_OK;
    if (!d_buf || !d_desc || !d_coeffs) return SYMACCEL_ERR_INVALID_ARG;
    DeviceGuard dev(ctx);
    if (!dev.ok()) return dev.status();
    return launch_flac_restore(ctx, d_buf, d_desc, d_coeffs, n_blocks, blocksize, d_pair_mode, d_pair_mode ? out_shift : 0, stride);
}

// Rows 4, 8, 16 or 32 KiB apart put the 64 row segments (128 B each) a wavefront moves per tile on a fraction of the HBM channels: the FLAC kernel runs 4096-sample
// blocks at 0.57 of the HBM peak with rows back to back, 0.63 with 256 B of padding, 0.66 with 2 KiB, 0.67-0.68 with 2.5-3 KiB, 0.67 with 4 or 8 KiB (128 B: 0.53,
// 64 B, which breaks the 128-byte alignment of the segments: 0.42); 1024-sample blocks 0.49 -> 0.62, 2048 0.51 -> 0.64, 8192 0.54 -> 0.67 with an eighth of a row
// (profiles/r06zz30_stride_sweep.txt, r06zz31_stride_sweep.txt; ALAC: 0.386 -> 0.42).  What spreads the segments is where a row STARTS modulo 16 KiB:
// dealing a wavefront rows that are 80 KiB apart in the compact plane changes nothing (profiles/r06zz34_interleave_ab.txt); an eighth of the row steps the starts by 2 KiB (4096-sample rows).
size_t symaccel_row_stride(size_t blocksize) {
    size_t s = (blocksize + 3) & ~(size_t)3;
    if (s >= 1024 && (s % 512) == 0) s += s / kRowPadDiv;
    return s;
}

int symaccel_flac_decorrelate(symaccel_ctx *ctx, const uint8_t *h_mode, int32_t *h_ch0, int32_t *h_ch1, size_t n_pairs,
                              size_t blocksize, uint32_t out_shift) {
    if (!ctx || out_shift > 31) return SYMACCEL_ERR_INVALID_ARG;
    if (n_pairs == 0 || blocksize == 0) return SYMACCEL_OK;
    if (!h_mode || !h_ch0 || !h_ch1) return SYMACCEL_ERR_INVALID_ARG;
    DeviceGuard dev(ctx);
    if (!dev.ok()) return dev.status();
    DevBuf mode(ctx), c0(ctx), c1(ctx);
    const size_t bytes = n_pairs * blocksize * 4;
    SYM_TRY(mode.from_host(h_mode, n_pairs));
    SYM_TRY(c0.from_host(h_ch0, bytes));
    SYM_TRY(c1.from_host(h_ch1, bytes));
    SYM_TRY(launch_flac_decorrelate(ctx, (const uint8_t *)mode.p, (int32_t *)c0.p, (int32_t *)c1.p, n_pairs, blocksize,
                                    out_shift));
    SYM_TRY(c0.to_host(h_ch0, bytes));
    SYM_TRY(c1.to_host(h_ch1, bytes));
    return symaccel_sync(ctx);
}

int symaccel_flac_decorrelate_device(symaccel_ctx *ctx, const uint8_t *d_mode, int32_t *d_ch0, int32_t *d_ch1,
                                     size_t n_pairs, size_t blocksize, uint32_t out_shift) {
    if (!ctx || out_shift > 31) return SYMACCEL_ERR_INVALID_ARG;
    if (n_pairs == 0 || blocksize == 0) return SYMACCEL_OK;
    if (!d_mode || !d_ch0 || !d_ch1) return SYMACCEL_ERR_INVALID_ARG;
    DeviceGuard dev(ctx);
    if (!dev.ok()) return dev.status();
    return launch_flac_decorrelate(ctx, d_mode, d_ch0, d_ch1, n_pairs, blocksize, out_shift);
}

// ---- per-record status arrays --------------------------------------------------------------

int symaccel_flac_block_status_device(symaccel_ctx *ctx, const symaccel_flac_desc *d_desc, size_t n_blocks, size_t blocksize,
                                      int8_t *d_status) {
    if (!ctx || blocksize > 65535) return SYMACCEL_ERR_INVALID_ARG;
    if (n_blocks == 0) return SYMACCEL_OK;
    if (!d_desc || !d_status) return SYMACCEL_ERR_INVALID_ARG;
    DeviceGuard dev(ctx);
    if (!dev.ok()) return dev.status();
    return launch_flac_status(ctx, d_desc, n_blocks, blocksize, d_status);
}

int symaccel_vorbis_floor1_status_device(symaccel_ctx *ctx, int n_posts, const uint32_t *d_y, size_t count, int8_t *d_status) {
    if (!ctx || n_posts < 2 || n_posts > 65) return SYMACCEL_ERR_INVALID_ARG;
    if (count == 0) return SYMACCEL_OK;
    if (!d_y || !d_status) return SYMACCEL_ERR_INVALID_ARG;
    DeviceGuard dev(ctx);
    if (!dev.ok()) return dev.status();
    return launch_floor1_status(ctx, d_y, count, n_posts, d_status);
}

int symaccel_alac_block_status_device(symaccel_ctx *ctx, const symaccel_alac_desc *d_desc, size_t n_blocks, int8_t *d_status) {
    if (!ctx) return SYMACCEL_ERR_INVALID_ARG;
    if (n_blocks == 0) return SYMACCEL_OK;
    if (!d_desc || !d_status) return SYMACCEL_ERR_INVALID_ARG;
    DeviceGuard dev(ctx);
    if (!dev.ok()) return dev.status();
    return launch_alac_status(ctx, d_desc, n_blocks, d_status);
}

int symaccel_aac_tns_status_device(symaccel_ctx *ctx, size_t n_frames, const symaccel_aac_tns_filter *d_filters, size_t n_filters,
                                   int8_t *d_status) {
    if (!ctx) return SYMACCEL_ERR_INVALID_ARG;
    if (n_filters == 0) return SYMACCEL_OK;
    if (!d_filters || !d_status) return SYMACCEL_ERR_INVALID_ARG;
    DeviceGuard dev(ctx);
    if (!dev.ok()) return dev.status();
    return launch_tns_status(ctx, d_filters, n_filters, n_frames, d_status);
}

// ---- tables -------------------------------------------------------------------------------

// line/4 -> band map of one swb offset table; false if the table is not what ICS get_bands() can return
static bool build_band_map(const uint16_t *swb, int n_swb, int lines, int max_bands, uint8_t *map4) {
    if (!swb || n_swb < 1 || n_swb > max_bands || swb[0] != 0) return false;
    std::memset(map4, 255, (size_t)lines / 4);
    for (int b = 0; b < n_swb; ++b) {
        if (swb[b + 1] <= swb[b] || swb[b + 1] > lines || (swb[b] & 3) || (swb[b + 1] & 3)) return false;
        for (int i = swb[b] / 4; i < swb[b + 1] / 4; ++i) map4[i] = (uint8_t)b;
    }
    return true;
}

int symaccel_aac_joint_stereo_device(symaccel_ctx *ctx, float *d_coeffs, size_t frames_per_chain,
                                     const int32_t *d_pair_chains, const symaccel_aac_js_frame *d_desc, size_t n_pairs,
                                     const uint16_t *swb_long, int n_swb_long, const uint16_t *swb_short, int n_swb_short) {
    if (!ctx) return SYMACCEL_ERR_INVALID_ARG;
    AacBandMaps maps;
    // long windows have at most 51 bands (ISO/IEC 14496-3 Table 4.139: 8 kHz... 49/51), short ones at most 15
    if (!build_band_map(swb_long, n_swb_long, 1024, 64, maps.long4) || !build_band_map(swb_short, n_swb_short, 128, 16, maps.short4))
        return SYMACCEL_ERR_INVALID_ARG;
    if (n_pairs == 0 || frames_per_chain == 0) return SYMACCEL_OK;
    if (!d_coeffs || !d_pair_chains || !d_desc) return SYMACCEL_ERR_INVALID_ARG;
    DeviceGuard dev(ctx);
    if (!dev.ok()) return dev.status();
    return launch_aac_joint_stereo(ctx, maps, d_coeffs, frames_per_chain, d_pair_chains, d_desc, n_pairs);
}

int symaccel_aac_joint_stereo_list_device(symaccel_ctx *ctx, float *d_coeffs, size_t frames_per_chain, const int32_t *d_pair_chains,
                                          const symaccel_aac_js_frame *d_desc, size_t n_pairs, const uint16_t *swb_long, int n_swb_long,
                                          const uint16_t *swb_short, int n_swb_short, const uint32_t *d_pair_frames,
                                          size_t n_pair_frames) {
    if (!ctx) return SYMACCEL_ERR_INVALID_ARG;
    AacBandMaps maps;
    if (!aac_band_maps(swb_long, n_swb_long, swb_short, n_swb_short, &maps)) return SYMACCEL_ERR_INVALID_ARG;
    if (n_pairs == 0 || frames_per_chain == 0 || n_pair_frames == 0) return SYMACCEL_OK;
    if (!d_coeffs || !d_pair_chains || !d_desc || !d_pair_frames) return SYMACCEL_ERR_INVALID_ARG;
    DeviceGuard dev(ctx);
    if (!dev.ok()) return dev.status();
    return launch_aac_joint_stereo(ctx, maps, d_coeffs, frames_per_chain, d_pair_chains, d_desc, n_pairs, d_pair_frames, n_pair_frames);
}

int symaccel_aac_tns_device(symaccel_ctx *ctx, float *d_coeffs, size_t n_frames, const symaccel_aac_tns_filter *d_filters,
                            size_t n_filters) {
    if (!ctx) return SYMACCEL_ERR_INVALID_ARG;
    if (n_filters == 0 || n_frames == 0) return SYMACCEL_OK;
    if (!d_coeffs || !d_filters) return SYMACCEL_ERR_INVALID_ARG;
    DeviceGuard dev(ctx);
    if (!dev.ok()) return dev.status();
    return launch_aac_tns(ctx, d_coeffs, n_frames, d_filters, n_filters);
}

int symaccel_mp3_stereo_device(symaccel_ctx *ctx, float *d_xr, size_t granules_per_chain, const int32_t *d_pair_chains,
                               const symaccel_mp3_stereo *d_desc, int sample_rate_idx, size_t n_pairs) {
    if (!ctx || sample_rate_idx < 0 || sample_rate_idx > 8) return SYMACCEL_ERR_INVALID_ARG;
    if (n_pairs == 0 || granules_per_chain == 0) return SYMACCEL_OK;
    if (!d_xr || !d_pair_chains || !d_desc) return SYMACCEL_ERR_INVALID_ARG;
    DeviceGuard dev(ctx);
    if (!dev.ok()) return dev.status();
    return launch_mp3_stereo(ctx, d_xr, granules_per_chain, d_pair_chains, d_desc, sample_rate_idx, n_pairs);
}

int symaccel_mp3_requantize_stereo_device(symaccel_ctx *ctx, const int16_t *d_quant, const symaccel_mp3_requant *d_rq_desc,
                                          size_t granules_per_chain, const int32_t *d_pair_chains,
                                          const symaccel_mp3_stereo *d_desc, int sample_rate_idx, float *d_xr, size_t n_pairs) {
    if (!ctx || sample_rate_idx < 0 || sample_rate_idx > 8) return SYMACCEL_ERR_INVALID_ARG;
    if (n_pairs == 0 || granules_per_chain == 0) return SYMACCEL_OK;
    if (!d_quant || !d_rq_desc || !d_xr || !d_pair_chains || !d_desc) return SYMACCEL_ERR_INVALID_ARG;
    DeviceGuard dev(ctx);
    if (!dev.ok()) return dev.status();
    return launch_mp3_stereo(ctx, d_xr, granules_per_chain, d_pair_chains, d_desc, sample_rate_idx, n_pairs, d_quant, d_rq_desc);
}

int symaccel_mp3_requantize_device(symaccel_ctx *ctx, const int16_t *d_quant, const symaccel_mp3_requant *d_desc,
                                   int sample_rate_idx, float *d_xr, size_t n) {
    if (!ctx || sample_rate_idx < 0 || sample_rate_idx > 8) return SYMACCEL_ERR_INVALID_ARG;
    if (n == 0) return SYMACCEL_OK;
    if (!d_quant || !d_desc || !d_xr) return SYMACCEL_ERR_INVALID_ARG;
    DeviceGuard dev(ctx);
    if (!dev.ok()) return dev.status();
    return launch_mp3_requantize(ctx, d_quant, d_desc, sample_rate_idx, d_xr, n);
}

int symaccel_mp3_requantize(symaccel_ctx *ctx, const int16_t *h_quant, const symaccel_mp3_requant *h_desc,
                            int sample_rate_idx, float *h_xr, size_t n) {
    if (!ctx || sample_rate_idx < 0 || sample_rate_idx > 8) return SYMACCEL_ERR_INVALID_ARG;
    if (n == 0) return SYMACCEL_OK;
    if (!h_quant || !h_desc || !h_xr) return SYMACCEL_ERR_INVALID_ARG;
    DeviceGuard dev(ctx);
    if (!dev.ok()) return dev.status();
    DevBuf q(ctx), d(ctx), x(ctx);
    SYM_TRY(q.from_host(h_quant, n * 576 * sizeof(int16_t)));
    SYM_TRY(d.from_host(h_desc, n * sizeof(symaccel_mp3_requant)));
    SYM_TRY(x.alloc(n * 576 * 4));
    SYM_TRY(launch_mp3_requantize(ctx, (const int16_t *)q.p, (const symaccel_mp3_requant *)d.p, sample_rate_idx, (float *)x.p, n));
    SYM_TRY(x.to_host(h_xr, n * 576 * 4));
    return symaccel_sync(ctx);
}

int symaccel_mpa_polyphase_pp_device(symaccel_ctx *ctx, int n_frames, const float *d_in, const float *d_vvec_in,
                                     const int32_t *d_vfront_in, float *d_vvec_out, int32_t *d_vfront_out, float *d_pcm,
                                     size_t n_chains, size_t packets_per_chain) {
    if (!ctx) return SYMACCEL_ERR_INVALID_ARG;
    if (n_frames != 12 && n_frames != 36) return SYMACCEL_ERR_UNSUPPORTED;
    if (n_chains == 0 || packets_per_chain == 0) return SYMACCEL_OK;
    if (!d_in || !d_vvec_in || !d_vfront_in || !d_vvec_out || !d_vfront_out || !d_pcm) return SYMACCEL_ERR_INVALID_ARG;
    if (d_vvec_in == d_vvec_out || d_vfront_in == d_vfront_out) return SYMACCEL_ERR_INVALID_ARG;
    DeviceGuard dev(ctx);
    if (!dev.ok()) return dev.status();
    return launch_mpa_polyphase(ctx, n_frames, d_in, d_vvec_in, d_vfront_in, d_vvec_out, d_vfront_out, d_pcm, n_chains,
                                packets_per_chain);
}

int symaccel_mpa_polyphase_device(symaccel_ctx *ctx, int n_frames, const float *d_in, float *d_vvec_io,
                                  int32_t *d_vfront_io, float *d_pcm, size_t n_chains, size_t packets_per_chain) {
    if (!ctx) return SYMACCEL_ERR_INVALID_ARG;
    if (n_frames != 12 && n_frames != 36) return SYMACCEL_ERR_UNSUPPORTED;  // Layer I / Layer II (Layer III: mp3_synth)
    if (n_chains == 0 || packets_per_chain == 0) return SYMACCEL_OK;
    if (!d_in || !d_vvec_io || !d_vfront_io || !d_pcm) return SYMACCEL_ERR_INVALID_ARG;
    DeviceGuard dev(ctx);
    if (!dev.ok()) return dev.status();
    const size_t vv_bytes = n_chains * 1024 * 4, vf_bytes = n_chains * 4;
    void *scratch = nullptr;
    SYM_TRY(ctx_scratch(ctx, vv_bytes + vf_bytes, &scratch));
    float *vv_out = (float *)scratch;
    int32_t *vf_out = (int32_t *)(vv_out + n_chains * 1024);
    SYM_TRY(launch_mpa_polyphase(ctx, n_frames, d_in, d_vvec_io, d_vfront_io, vv_out, vf_out, d_pcm, n_chains,
                                 packets_per_chain));
    SYM_TRY(launch_state_copy(ctx, d_vvec_io, vv_out, vv_bytes, d_vfront_io, vf_out, vf_bytes, nullptr, nullptr, 0));
    return SYMACCEL_OK;
}

int symaccel_mpa_polyphase(symaccel_ctx *ctx, int n_frames, const float *h_in, float *h_vvec_io, int32_t *h_vfront_io,
                           float *h_pcm, size_t n_chains, size_t packets_per_chain) {
    if (!ctx) return SYMACCEL_ERR_INVALID_ARG;
    if (n_frames != 12 && n_frames != 36) return SYMACCEL_ERR_UNSUPPORTED;
    if (n_chains == 0 || packets_per_chain == 0) return SYMACCEL_OK;
    if (!h_in || !h_vvec_io || !h_vfront_io || !h_pcm) return SYMACCEL_ERR_INVALID_ARG;
    DeviceGuard dev(ctx);
    if (!dev.ok()) return dev.status();
    const size_t bytes = n_chains * packets_per_chain * 32 * (size_t)n_frames * 4;
    DevBuf in(ctx), vv(ctx), vf(ctx), pcm(ctx);
    SYM_TRY(in.from_host(h_in, bytes));
    SYM_TRY(vv.from_host(h_vvec_io, n_chains * 1024 * 4));
    SYM_TRY(vf.from_host(h_vfront_io, n_chains * 4));
    SYM_TRY(pcm.alloc(bytes));
    SYM_TRY(symaccel_mpa_polyphase_device(ctx, n_frames, (const float *)in.p, (float *)vv.p, (int32_t *)vf.p, (float *)pcm.p,
                                          n_chains, packets_per_chain));
    SYM_TRY(pcm.to_host(h_pcm, bytes));
    SYM_TRY(vv.to_host(h_vvec_io, n_chains * 1024 * 4));
    SYM_TRY(vf.to_host(h_vfront_io, n_chains * 4));
    return symaccel_sync(ctx);
}

int symaccel_alac_predict_device(symaccel_ctx *ctx, int32_t *d_buf, const symaccel_alac_desc *d_desc,
                                 const int32_t *d_coeffs, size_t n_blocks, size_t blocksize) {
    if (!ctx || blocksize > 0xffffffffu) return SYMACCEL_ERR_INVALID_ARG;
    if (n_blocks == 0 || blocksize == 0) return SYMACCEL_OK;
    if (!d_buf || !d_desc || !d_coeffs) return SYMACCEL_ERR_INVALID_ARG;
    DeviceGuard dev(ctx);
    if (!dev.ok()) return dev.status();
    return launch_alac_predict(ctx, d_buf, d_desc, d_coeffs, n_blocks, blocksize);
}

int symaccel_alac_predict_stereo_device(symaccel_ctx *ctx, int32_t *d_buf, const symaccel_alac_desc *d_desc,
                                        const int32_t *d_coeffs, const int32_t *d_pair_weight, const uint8_t *d_pair_shift,
                                        size_t n_blocks, size_t blocksize) {
    if (!ctx || blocksize > 0xffffffffu || (n_blocks & 1)) return SYMACCEL_ERR_INVALID_ARG;
    if (n_blocks == 0 || blocksize == 0) return SYMACCEL_OK;
    if (!d_buf || !d_desc || !d_coeffs || !d_pair_weight || !d_pair_shift) return SYMACCEL_ERR_INVALID_ARG;
    DeviceGuard dev(ctx);
    if (!dev.ok()) return dev.status();
    return launch_alac_predict(ctx, d_buf, d_desc, d_coeffs, n_blocks, blocksize, d_pair_weight, d_pair_shift);
}

int symaccel_alac_predict_strided_device(symaccel_ctx *ctx, int32_t *d_buf, const symaccel_alac_desc *d_desc,
                                         const int32_t *d_coeffs, const int32_t *d_pair_weight, const uint8_t *d_pair_shift,
                                         size_t n_blocks, size_t blocksize, size_t stride) {
    if (stride == 0) stride = blocksize;
    if (!ctx || blocksize > 0xffffffffu || stride < blocksize || stride > 0xffffffffu) return SYMACCEL_ERR_INVALID_ARG;
    if ((d_pair_weight != nullptr) != (d_pair_shift != nullptr) || (d_pair_weight && (n_blocks & 1))) return SYMACCEL_ERR_INVALID_ARG;
    if (n_blocks == 0 || blocksize == 0) return SYMACCEL_OK;
    if (!d_buf || !d_desc || !d_coeffs) return SYMACCEL_ERR_INVALID_ARG;
    DeviceGuard dev(ctx);
    if (!dev.ok()) return dev.status();
    return launch_alac_predict(ctx, d_buf, d_desc, d_coeffs, n_blocks, blocksize, d_pair_weight, d_pair_shift, stride);
}

int symaccel_alac_predict(symaccel_ctx *ctx, int32_t *h_buf, const symaccel_alac_desc *h_desc, const int32_t *h_coeffs,
                          size_t n_blocks, size_t blocksize) {
    if (!ctx || blocksize > 0xffffffffu) return SYMACCEL_ERR_INVALID_ARG;
    if (n_blocks == 0 || blocksize == 0) return SYMACCEL_OK;
    if (!h_buf || !h_desc || !h_coeffs) return SYMACCEL_ERR_INVALID_ARG;
    DeviceGuard dev(ctx);
    if (!dev.ok()) return dev.status();
    DevBuf buf(ctx), desc(ctx), co(ctx);
    const size_t bytes = n_blocks * blocksize * 4;
    SYM_TRY(buf.from_host(h_buf, bytes));
    SYM_TRY(desc.from_host(h_desc, n_blocks * sizeof(symaccel_alac_desc)));
    SYM_TRY(co.from_host(h_coeffs, n_blocks * 32 * 4));
    SYM_TRY(launch_alac_predict(ctx, (int32_t *)buf.p, (const symaccel_alac_desc *)desc.p, (const int32_t *)co.p, n_blocks,
                                blocksize));
    SYM_TRY(buf.to_host(h_buf, bytes));
    return symaccel_sync(ctx);
}

int symaccel_alac_mid_side_device(symaccel_ctx *ctx, const int32_t *d_weight, const uint8_t *d_shift, int32_t *d_ch0,
                                  int32_t *d_ch1, size_t n_pairs, size_t blocksize) {
    if (!ctx) return SYMACCEL_ERR_INVALID_ARG;
    if (n_pairs == 0 || blocksize == 0) return SYMACCEL_OK;
    if (!d_weight || !d_shift || !d_ch0 || !d_ch1) return SYMACCEL_ERR_INVALID_ARG;
    DeviceGuard dev(ctx);
    if (!dev.ok()) return dev.status();
    return launch_alac_mid_side(ctx, d_weight, d_shift, d_ch0, d_ch1, n_pairs, blocksize);
}

int symaccel_alac_mid_side(symaccel_ctx *ctx, const int32_t *h_weight, const uint8_t *h_shift, int32_t *h_ch0,
                           int32_t *h_ch1, size_t n_pairs, size_t blocksize) {
    if (!ctx) return SYMACCEL_ERR_INVALID_ARG;
    if (n_pairs == 0 || blocksize == 0) return SYMACCEL_OK;
    if (!h_weight || !h_shift || !h_ch0 || !h_ch1) return SYMACCEL_ERR_INVALID_ARG;
    DeviceGuard dev(ctx);
    if (!dev.ok()) return dev.status();
    DevBuf w(ctx), sh(ctx), c0(ctx), c1(ctx);
    const size_t bytes = n_pairs * blocksize * 4;
    SYM_TRY(w.from_host(h_weight, n_pairs * 4));
    SYM_TRY(sh.from_host(h_shift, n_pairs));
    SYM_TRY(c0.from_host(h_ch0, bytes));
    SYM_TRY(c1.from_host(h_ch1, bytes));
    SYM_TRY(launch_alac_mid_side(ctx, (const int32_t *)w.p, (const uint8_t *)sh.p, (int32_t *)c0.p, (int32_t *)c1.p, n_pairs,
                                 blocksize));
    SYM_TRY(c0.to_host(h_ch0, bytes));
    SYM_TRY(c1.to_host(h_ch1, bytes));
    return symaccel_sync(ctx);
}

int symaccel_table_f32(const symaccel_ctx *, int table, float *dst, size_t capacity) {
    const HostTables &t = host_tables();
    const float *src = nullptr;
    size_t n = 0;
    switch (table) {
        case SYMACCEL_TABLE_AAC_KBD_LONG: src = t.aac_kbd_long.data(); n = 1024; break;
        case SYMACCEL_TABLE_AAC_KBD_SHORT: src = t.aac_kbd_short.data(); n = 128; break;
        case SYMACCEL_TABLE_AAC_SINE_LONG: src = t.aac_sine_long.data(); n = 1024; break;
        case SYMACCEL_TABLE_AAC_SINE_SHORT: src = t.aac_sine_short.data(); n = 128; break;
        case SYMACCEL_TABLE_MP3_SYNTH_D: src = t.mp3_synth_d; n = 512; break;
        case SYMACCEL_TABLE_MP3_IMDCT_WIN: src = &t.mp3_imdct_win[0][0]; n = 144; break;
        case SYMACCEL_TABLE_VORBIS_FLOOR1_DB: src = t.vorbis_floor1_db; n = 256; break;
        case SYMACCEL_TABLE_MP3_POW43: src = t.mp3_pow43; n = 8207; break;
        case SYMACCEL_TABLE_MP3_POW2AB: src = t.mp3_pow2ab; n = kMp3Pow2abLen; break;
        case SYMACCEL_TABLE_MP3_CONSTS: {
            static const std::vector<float> mc = pack_mp3_consts(t);
            src = mc.data();
            n = MP3C_SYNTH_D;
            break;
        }
        default: return SYMACCEL_ERR_INVALID_ARG;
    }
    if (!dst || capacity < n) return SYMACCEL_ERR_INVALID_ARG;
    std::memcpy(dst, src, n * 4);
    return (int)n;
}

int symaccel_probe_copy_device(symaccel_ctx *ctx, const void *d_src, void *d_dst, size_t bytes, uint32_t frames_per_wavefront,
                               uint32_t flags) {
    if (!ctx) return SYMACCEL_ERR_INVALID_ARG;
    if (bytes == 0) return SYMACCEL_OK;
    if (!d_src || !d_dst || bytes % 4096 != 0 || ((uintptr_t)d_src | (uintptr_t)d_dst) % 16 != 0 || flags > 63u || ((flags >> 1) & 3u) == 3u)
        return SYMACCEL_ERR_INVALID_ARG;
    const uintptr_t a = (uintptr_t)d_src, b = (uintptr_t)d_dst;
    if (a < b + bytes && b < a + bytes) return SYMACCEL_ERR_INVALID_ARG;  // overlapping buffers
    DeviceGuard dev(ctx);
    if (!dev.ok()) return dev.status();
    return launch_probe_copy(ctx, d_src, d_dst, bytes, frames_per_wavefront, flags);
}

int symaccel_imdct_twiddles(int n, double scale, float *dst) {
    if (!pow2(n) || n < 2 || !dst) return SYMACCEL_ERR_INVALID_ARG;
    make_imdct_twiddles(n, scale, (cpx *)dst);
    return n / 2;
}

int symaccel_fft_twiddles(int n, float *dst) {
    if (!pow2(n) || n < 2 || !dst) return SYMACCEL_ERR_INVALID_ARG;
    make_fft_twiddles(n, (cpx *)dst);
    return n / 2;
}

}  // extern "C"

namespace symaccel {
bool aac_band_maps(const uint16_t *swb_long, int n_swb_long, const uint16_t *swb_short, int n_swb_short, AacBandMaps *maps) {
    // long windows have at most 51 bands (ISO/IEC 14496-3 Table 4.139: 8 kHz... 49/51), short ones at most 15
    return build_band_map(swb_long, n_swb_long, 1024, 64, maps->long4) && build_band_map(swb_short, n_swb_short, 128, 16, maps->short4);
}
}  // namespace symaccel
