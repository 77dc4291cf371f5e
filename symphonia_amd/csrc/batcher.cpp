// Cross-stream batcher: many decoders, one launch.
//
// A `Hip*Decoder` (bindings/rust) or a `codecs::LookaheadDecoder` (include/symaccel.hpp) batches the look-ahead of ITS stream:
// 256 packets x 2 channels is 2 MiB per call, and N decoders are N small calls, each paying its own launch and its own PCIe
// round trip.  The reference's trait gives a decoder no view of its siblings (AudioDecoder::decode_ref sees one packet of one
// track, symphonia-core/src/codecs/audio.rs:279-297; the registry builds decoders from (params, opts) alone, registry.rs:330-341),
// so the coalescing point is below the trait, here: decoders SUBMIT their batches to a batcher shared by the process and come
// back for the result later; whatever is pending when somebody needs a result (or when `flush_bytes` of input have piled up)
// goes to the device as ONE batch per (kind, units per chain) group -- the chains of every submission side by side in the
// chain-major layout the kernels already take, so nothing in the kernels knows about streams.
//
//   reserve()  -> a slot of page-locked staging memory the front end writes its spectra / samples / records into (no copy
//                 between the parser's output and the DMA source) + a ticket
//   commit()   -> the slot is filled
//   wait()     -> launches the ticket's group if nobody has yet (and everything else that is pending), blocks until the
//                 group's results are in page-locked memory; slot.out / slot.state now hold PCM and the carried state
//   release()  -> the slot may be reused
//
// A group is transferred and transformed in chunks of submissions: H2D(c + 1) || kernel(c) || D2H(c - 1) on the context's three
// streams (the chunks are chains, which are independent: no carried state between chunks, unlike stage.cpp's frame-axis
// chunks).  Groups are pooled: in the steady state nothing is allocated.  Thread-safe: submissions may come from any thread; a
// flush waits for reservations of the group that are still being filled.  A context that has a batcher is driven through the
// batcher only (the context itself is externally synchronised, include/symaccel.h "Thread safety").
#include <algorithm>
#include <condition_variable>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <mutex>
#include <vector>

#include "symaccel_internal.h"

using namespace symaccel;

namespace {

constexpr int kMaxIn = 4, kMaxState = 3;

// what a kind's planes weigh: bytes per chain (per ticket for `in_per_ticket`) for `units` frames / granules per chain
struct PlaneSizes {
    size_t in[kMaxIn] = {0, 0, 0, 0};
    bool in_per_ticket[kMaxIn] = {false, false, false, false};
    size_t state[kMaxState] = {0, 0, 0};
    size_t out = 0;
    int n_in = 0, n_state = 0;
};

bool plane_sizes(int kind, int param, size_t units, PlaneSizes *ps) {
    *ps = PlaneSizes();
    switch (kind) {
    case SYMACCEL_BATCH_AAC_DECODE:  // symaccel_aac_decode_pipelined for one stream: coeffs, side, the stream's descriptor blob | delay | pcm
        ps->n_in = 3;
        ps->in[0] = units * 4096;
        ps->in[1] = units;
        // the blob (aac_blob_*): header + pair list + joint-stereo rows (at most one pair per two chains) + TNS filters (at most
        // eight per channel-frame: one per window of an EIGHT_SHORT frame), sized per chain so that it scales with the stream
        ps->in[2] = 64 + units * (sizeof(symaccel_aac_js_frame) / 2 + 8 * sizeof(symaccel_aac_tns_filter));
        ps->n_state = 1;
        ps->state[0] = 4096;
        ps->out = units * 4096;
        return true;
    case SYMACCEL_BATCH_VORBIS_SYNTH: {  // symaccel_vorbis_synth with every chain's planes at their largest: spectra, flags | prev, overlap | pcm
        const int e0 = param & 255, e1 = (param >> 8) & 255;
        if (e0 < 6 || e1 > 13 || e0 > e1 || (param >> 16)) return false;  // (the block sizes a Vorbis stream can have, lib.rs:404-406)
        const size_t half = (size_t)1 << (e1 - 1);
        ps->n_in = 2;
        ps->in[0] = units * half * 4;  // a block has at most bs1 / 2 lines ...
        ps->in[1] = units;
        ps->n_state = 2;
        ps->state[0] = 4;
        ps->state[1] = half * 4;
        ps->out = units * half * 4;    // ... and yields at most bs1 / 2 samples
        return true;
    }
    case SYMACCEL_BATCH_AAC_SYNTH:  // symaccel_aac_synth: coeffs, side | delay | pcm
        ps->n_in = 2;
        ps->in[0] = units * 4096;
        ps->in[1] = units;
        ps->n_state = 1;
        ps->state[0] = 4096;
        ps->out = units * 4096;
        return true;
    case SYMACCEL_BATCH_MP3_SYNTH:  // symaccel_mp3_synth: xr, side | overlap, vvec, vfront | pcm
        ps->n_in = 2;
        ps->in[0] = units * 2304;
        ps->in[1] = units * sizeof(symaccel_mp3_side);
        ps->n_state = 3;
        ps->state[0] = 2304;
        ps->state[1] = 4096;
        ps->state[2] = 4;
        ps->out = units * 2304;
        return true;
    case SYMACCEL_BATCH_MP3_DECODE:  // symaccel_mp3_decode_pipelined, one stream per submission: quant, rq_desc, side, st_desc (per stream)
        ps->n_in = 4;
        ps->in[0] = units * 1152;
        ps->in[1] = units * sizeof(symaccel_mp3_requant);
        ps->in[2] = units * sizeof(symaccel_mp3_side);
        ps->in[3] = units * sizeof(symaccel_mp3_stereo);
        ps->in_per_ticket[3] = true;
        ps->n_state = 3;
        ps->state[0] = 2304;
        ps->state[1] = 4096;
        ps->state[2] = 4;
        ps->out = units * 2304;
        return true;
    default:
        return false;
    }
}

size_t round256(size_t v) { return (v + 255) & ~(size_t)255; }

size_t in_bytes_per_chain(const PlaneSizes &ps) {
    size_t s = 0;
    for (int i = 0; i < ps.n_in; ++i)
        if (!ps.in_per_ticket[i]) s += ps.in[i];
    return s;
}

// a submission's page-locked slot: [in 0 | .. | state 0 | .. | out], every plane on a 256-byte boundary
struct SlotLayout {
    size_t in[kMaxIn] = {}, state[kMaxState] = {}, out = 0, bytes = 0;
    size_t in_bytes[kMaxIn] = {}, state_bytes[kMaxState] = {}, out_bytes = 0;
};
SlotLayout slot_layout(const PlaneSizes &ps, size_t n_chains) {
    SlotLayout l;
    size_t off = 0;
    for (int i = 0; i < ps.n_in; ++i) {
        l.in[i] = off;
        l.in_bytes[i] = ps.in[i] * (ps.in_per_ticket[i] ? 1 : n_chains);
        off += round256(l.in_bytes[i]);
    }
    for (int i = 0; i < ps.n_state; ++i) {
        l.state[i] = off;
        l.state_bytes[i] = ps.state[i] * n_chains;
        off += round256(l.state_bytes[i]);
    }
    l.out = off;
    l.out_bytes = ps.out * n_chains;
    off += round256(l.out_bytes);
    l.bytes = off;
    return l;
}

// The descriptor blob of an AAC_DECODE submission (plane in[2], n_chains * ps.in[2] bytes): what symaccel_aac_decode_pipelined takes
// beside the spectra -- [AacBlobHeader][pair_chains: n_pairs x 2 i32, chains of THIS submission][js rows: n_pairs x units x 644 B]
// [TNS filters: n_tns x 92 B, frame = chain * units + frame inside this submission]
struct AacBlobHeader {
    uint32_t n_pairs, n_tns, pad[2];
};
inline size_t aac_blob_pairs(size_t) { return sizeof(AacBlobHeader); }
inline size_t aac_blob_js(size_t n_pairs) { return sizeof(AacBlobHeader) + ((n_pairs * 8 + 15) & ~(size_t)15); }
inline size_t aac_blob_tns(size_t n_pairs, size_t units) { return aac_blob_js(n_pairs) + ((n_pairs * units * sizeof(symaccel_aac_js_frame) + 15) & ~(size_t)15); }
inline size_t aac_blob_bytes(size_t n_pairs, size_t units, size_t n_tns) { return aac_blob_tns(n_pairs, units) + n_tns * sizeof(symaccel_aac_tns_filter); }

struct Group;

struct Ticket {
    Group *group = nullptr;
    uint32_t gen = 0;
    uint32_t first_chain = 0, n_chains = 0, ordinal = 0;
    bool live = false, committed = false;
    char *slot = nullptr;  // page-locked, slot_layout(group's planes, n_chains)
    size_t slot_bytes = 0;
    // copy form (symaccel_batcher_submit): where collect() puts the results
    void *user_state[kMaxState] = {nullptr, nullptr, nullptr};
    void *user_out = nullptr;
};

enum class GroupState { Free, Open, Closed, Launched };

// One launch: the submissions of one shape that were pending together.  Host side: the submissions' own slots.  Device side: one
// allocation, cut at launch time (when the number of chains is known) and kept with the group object for the next launch it serves.
struct Group {
    int kind = 0, param = 0;
    size_t units = 0;
    PlaneSizes ps;
    size_t cap_chains = 0;  // reservations accepted before the group is launched and a fresh one opened
    size_t chains = 0, tickets = 0, uncommitted = 0, live = 0;
    GroupState state = GroupState::Free;
    int status = SYMACCEL_OK;
    std::vector<uint32_t> ticket_ids;  // the submissions, in order (index into symaccel_batcher::tickets)
    char *d_base = nullptr;
    size_t d_bytes = 0;
    char *d_in[kMaxIn] = {}, *d_state_in[kMaxState] = {}, *d_state_out[kMaxState] = {}, *d_out = nullptr;
    int32_t *d_units = nullptr;  // MP3_DECODE: unit_chains of every chunk, relative to the chunk's first chain
    // AAC_DECODE: the group's pair list, joint-stereo rows, TNS filters, the pair frames that carry TNS, the walk's chain index
    int32_t *d_aac_pairs = nullptr;
    symaccel_aac_js_frame *d_aac_js = nullptr;
    symaccel_aac_tns_filter *d_aac_tns = nullptr;
    uint32_t *d_aac_pf = nullptr;
    void *d_aac_index = nullptr;
    size_t aac_pairs = 0, aac_tns = 0;  // totals of the group (counted when it is launched)
    struct {                             // ... and of the chunk being launched: first pair / filter / TNS pair frame and their counts
        size_t p0, np, f0, nf, q0, nq;
    } aac_chunk{};
    bool bad_blob = false;               // a submission's descriptor blob did not add up: the group's tickets fail with INVALID_ARG
    // page-locked: the copy descriptors of the launch (read by batch_copy_kernel straight from here) and the unit list
    char *h_desc = nullptr;
    size_t h_desc_bytes = 0;
    hipEvent_t ev_in[2] = {nullptr, nullptr}, ev_k[2] = {nullptr, nullptr}, done = nullptr;
};

}  // namespace

struct symaccel_batcher {
    symaccel_ctx *ctx = nullptr;
    size_t flush_bytes = 0;
    std::mutex mu;
    std::condition_variable cv;
    std::vector<std::unique_ptr<Group>> groups;
    std::vector<Ticket> tickets;
    std::vector<uint32_t> free_tickets;
    symaccel_batcher_stats stats{};
    std::string last_error;
    size_t hint_bytes = 0;  // what a group must hold for a hint to launch it
    // page-locked slot memory: slabs, carved into slots by size; a released slot goes to the free list of its size (the shapes of a
    // running service repeat: in the steady state nothing is allocated)
    struct Slab {
        char *base;
        size_t bytes, used;
    };
    std::vector<Slab> slabs;
    std::vector<std::pair<size_t, std::vector<char *>>> free_slots;
    // AAC_DECODE: the scale-factor-band tables a stream's joint-stereo descriptors refer to, registered once per stream shape
    // (symaccel_batcher_aac_bands); a submission names its table by index (`param`), which is part of the group key
    struct Bands {
        std::vector<uint16_t> swb_long, swb_short;
        AacBandMaps maps;
    };
    std::vector<Bands> bands;
};

namespace {

constexpr size_t kSlabBytes = (size_t)32 << 20;

int slot_alloc(symaccel_batcher *b, size_t bytes, char **out) {
    for (auto &cls : b->free_slots)
        if (cls.first == bytes && !cls.second.empty()) {
            *out = cls.second.back();
            cls.second.pop_back();
            return SYMACCEL_OK;
        }
    for (auto &sl : b->slabs)
        if (sl.bytes - sl.used >= bytes) {
            *out = sl.base + sl.used;
            sl.used += bytes;
            return SYMACCEL_OK;
        }
    const size_t want = std::max(kSlabBytes, bytes);
    void *h = nullptr;
    DeviceGuard dev(b->ctx);
    if (!dev.ok()) return dev.status();
    if (hipHostMalloc(&h, want, hipHostMallocDefault) != hipSuccess) {
        (void)hipGetLastError();
        return SYMACCEL_ERR_OOM;
    }
    b->slabs.push_back({static_cast<char *>(h), want, bytes});
    b->stats.staging_bytes += want;
    *out = static_cast<char *>(h);
    return SYMACCEL_OK;
}

void slot_free(symaccel_batcher *b, char *p, size_t bytes) {
    if (!p) return;
    for (auto &cls : b->free_slots)
        if (cls.first == bytes) {
            cls.second.push_back(p);
            return;
        }
    b->free_slots.push_back({bytes, {p}});
}

void group_free(Group *g) {
    if (g->d_base) (void)hipFree(g->d_base);
    if (g->h_desc) (void)hipHostFree(g->h_desc);
    for (hipEvent_t e : {g->ev_in[0], g->ev_in[1], g->ev_k[0], g->ev_k[1], g->done})
        if (e) (void)hipEventDestroy(e);
    g->d_base = nullptr;
    g->h_desc = nullptr;
}

// the device side of a closed group: sized for the chains it holds, planes carved out; grown (never shrunk) across reuses
int group_device(symaccel_batcher *b, Group *g, size_t n_pieces_bound) {
    symaccel_ctx *ctx = b->ctx;
    const PlaneSizes &ps = g->ps;
    size_t total = 0, off_in[kMaxIn], off_si[kMaxState], off_so[kMaxState], off_out, off_units;
    for (int i = 0; i < ps.n_in; ++i) {
        off_in[i] = total;
        total += round256(ps.in[i] * (ps.in_per_ticket[i] ? g->tickets : g->chains));
    }
    for (int i = 0; i < ps.n_state; ++i) {
        off_si[i] = total;
        total += round256(ps.state[i] * g->chains);
        off_so[i] = total;
        total += round256(ps.state[i] * g->chains);
    }
    off_out = total;
    total += round256(ps.out * g->chains);
    off_units = total;
    total += round256(g->tickets * 8);
    size_t off_ap = 0, off_aj = 0, off_at = 0, off_af = 0, off_ai = 0;
    if (g->kind == SYMACCEL_BATCH_AAC_DECODE) {
        off_ap = total;
        total += round256(std::max<size_t>(1, g->aac_pairs) * 8);
        off_aj = total;
        total += round256(std::max<size_t>(1, g->aac_pairs) * g->units * sizeof(symaccel_aac_js_frame));
        off_at = total;
        total += round256(std::max<size_t>(1, g->aac_tns) * sizeof(symaccel_aac_tns_filter));
        off_af = total;
        total += round256(std::max<size_t>(1, g->aac_tns) * 4);
        off_ai = total;
        total += round256(aac_js_scratch_bytes(g->chains, g->aac_pairs, g->units));
    }
    if (total > g->d_bytes) {
        if (g->d_base) SYM_GPU(ctx, hipFree(g->d_base));
        g->d_base = nullptr;
        g->d_bytes = 0;
        void *d = nullptr;
        SYM_TRY(ctx_alloc(ctx, &d, total + total / 4, false));
        g->d_base = static_cast<char *>(d);
        g->d_bytes = total + total / 4;
    }
    for (int i = 0; i < ps.n_in; ++i) g->d_in[i] = g->d_base + off_in[i];
    for (int i = 0; i < ps.n_state; ++i) {
        g->d_state_in[i] = g->d_base + off_si[i];
        g->d_state_out[i] = g->d_base + off_so[i];
    }
    g->d_out = g->d_base + off_out;
    g->d_units = reinterpret_cast<int32_t *>(g->d_base + off_units);
    if (g->kind == SYMACCEL_BATCH_AAC_DECODE) {
        g->d_aac_pairs = reinterpret_cast<int32_t *>(g->d_base + off_ap);
        g->d_aac_js = reinterpret_cast<symaccel_aac_js_frame *>(g->d_base + off_aj);
        g->d_aac_tns = reinterpret_cast<symaccel_aac_tns_filter *>(g->d_base + off_at);
        g->d_aac_pf = reinterpret_cast<uint32_t *>(g->d_base + off_af);
        g->d_aac_index = g->d_base + off_ai;
    }
    // (behind the descriptors: the MP3 unit list, or AAC_DECODE's rebased pair list, TNS filters and TNS pair frames)
    const size_t desc_bytes = round256(n_pieces_bound * sizeof(BatchCopyDesc)) + round256(g->tickets * 8) +
                              (g->kind == SYMACCEL_BATCH_AAC_DECODE ? round256(g->aac_pairs * 8) + round256(g->aac_tns * sizeof(symaccel_aac_tns_filter)) +
                                                                          round256(g->aac_tns * 4) : 0);
    if (desc_bytes > g->h_desc_bytes) {
        if (g->h_desc) (void)hipHostFree(g->h_desc);
        g->h_desc = nullptr;
        g->h_desc_bytes = 0;
        void *h = nullptr;
        if (hipHostMalloc(&h, desc_bytes + desc_bytes / 4, hipHostMallocDefault) != hipSuccess) {
            (void)hipGetLastError();
            return SYMACCEL_ERR_OOM;
        }
        g->h_desc = static_cast<char *>(h);
        g->h_desc_bytes = desc_bytes + desc_bytes / 4;
    }
    for (hipEvent_t *e : {&g->ev_in[0], &g->ev_in[1], &g->ev_k[0], &g->ev_k[1], &g->done})
        if (!*e) SYM_GPU(ctx, hipEventCreateWithFlags(e, hipEventDisableTiming));
    return SYMACCEL_OK;
}

// the kernels of one chunk: chains [c0, c0 + nc), submissions [t0, t0 + nt)
int launch_chunk(symaccel_batcher *b, Group *g, size_t c0, size_t nc, size_t t0, size_t nt) {
    symaccel_ctx *ctx = b->ctx;
    const PlaneSizes &ps = g->ps;
    auto in = [&](int i) { return g->d_in[i] + (ps.in_per_ticket[i] ? t0 : c0) * ps.in[i]; };
    auto si = [&](int i) { return g->d_state_in[i] + c0 * ps.state[i]; };
    auto so = [&](int i) { return g->d_state_out[i] + c0 * ps.state[i]; };
    char *out = g->d_out + c0 * ps.out;
    switch (g->kind) {
    case SYMACCEL_BATCH_AAC_SYNTH:
        return launch_aac(ctx, (const float *)in(0), (const uint8_t *)in(1), (const float *)si(0), (float *)so(0), (float *)out, nc, g->units);
    case SYMACCEL_BATCH_MP3_SYNTH:
        return launch_mp3(ctx, (const float *)in(0), (const symaccel_mp3_side *)in(1), g->param, (const float *)si(0), (const float *)si(1),
                          (const int32_t *)si(2), (float *)so(0), (float *)so(1), (int32_t *)so(2), (float *)out, nc, g->units);
    case SYMACCEL_BATCH_MP3_DECODE:
        return launch_mp3_decode(ctx, (const int16_t *)in(0), (const symaccel_mp3_requant *)in(1), g->d_units + 2 * t0,
                                 (const symaccel_mp3_stereo *)in(3), nt, (const symaccel_mp3_side *)in(2), g->param, (const float *)si(0),
                                 (const float *)si(1), (const int32_t *)si(2), (float *)so(0), (float *)so(1), (int32_t *)so(2), (float *)out, nc,
                                 g->units);
    case SYMACCEL_BATCH_AAC_DECODE: {
        // symaccel_aac_decode_pipelined's kernel sequence (csrc/stage.cpp) on the chunk: the pair frames that carry TNS get their joint
        // stereo decoded in place (a list pass), the filters run, ONE walk decodes the joint stereo of every other frame on load
        if (g->param < 0 || (size_t)g->param >= b->bands.size()) return SYMACCEL_ERR_INVALID_ARG;
        const AacBandMaps &maps = b->bands[(size_t)g->param].maps;
        const auto &ch = g->aac_chunk;
        const int32_t *pairs = g->d_aac_pairs + 2 * ch.p0;
        symaccel_aac_js_frame *js = g->d_aac_js + ch.p0 * g->units;
        if (ch.nq) {
            SYM_TRY(launch_aac_joint_stereo(ctx, maps, (float *)in(0), g->units, pairs, js, ch.np, g->d_aac_pf + ch.q0, ch.nq));
            SYM_TRY(launch_aac_js_consume(ctx, js, g->d_aac_pf + ch.q0, ch.nq, ch.np * g->units));
        }
        if (ch.nf) SYM_TRY(launch_aac_tns(ctx, (float *)in(0), nc * g->units, g->d_aac_tns + ch.f0, ch.nf));
        return launch_aac(ctx, (const float *)in(0), (const uint8_t *)in(1), (const float *)si(0), (float *)so(0), (float *)out, nc, g->units,
                          ch.np ? &maps : nullptr, pairs, js, ch.np, g->d_aac_index);
    }
    case SYMACCEL_BATCH_VORBIS_SYNTH: {
        const int e0 = g->param & 255, e1 = (g->param >> 8) & 255;
        const size_t cap = g->units << (e1 - 1);  // floats per chain of the spectrum and the PCM planes
        return symaccel_vorbis_synth_pp_device(ctx, e0, e1, (const float *)in(0), nullptr, cap, (const uint8_t *)in(1), (const int32_t *)si(0),
                                               (int32_t *)so(0), (const float *)si(1), (float *)so(1), (float *)out, cap, nc, g->units);
    }
    default:
        return SYMACCEL_ERR_INVALID_ARG;
    }
}

// Vorbis: how much of a chain's spectrum / PCM plane its blocks fill (lines of the packed spectrum; samples of the packed PCM:
// lib.rs:303 -- a block yields (prev_n + n) / 4, the first block after a reset keeps n / 2 slots): only that much crosses the link
void vorbis_used(const uint8_t *flags, size_t nb, int32_t prev, int e0, int e1, size_t *lines, size_t *samples) {
    const size_t bs[2] = {(size_t)1 << e0, (size_t)1 << e1};
    size_t l = 0, s = 0;
    int p = prev < 0 ? -1 : (prev ? 1 : 0);
    for (size_t i = 0; i < nb; ++i) {
        const int f = flags[i] ? 1 : 0;
        l += bs[f] / 2;
        s += p >= 0 ? (bs[p] + bs[f]) / 4 : bs[f] / 2;
        p = f;
    }
    *lines = l;
    *samples = s;
}

size_t pieces_of(size_t bytes) { return (bytes + kBatchCopyPiece - 1) / kBatchCopyPiece; }

void add_pieces(BatchCopyDesc *&w, const char *src, char *dst, size_t bytes) {
    for (size_t o = 0; o < bytes; o += kBatchCopyPiece) {
        w->src = src + o;
        w->dst = dst + o;
        w->bytes = (uint32_t)std::min(kBatchCopyPiece, bytes - o);
        w->pad = 0;
        ++w;
    }
}

// Everything of a closed group: per chunk of submissions ONE gather launch (slots -> HBM, the kernels' chain-major layout), the
// synthesis kernel, ONE scatter launch (HBM -> slots); `done` is recorded behind the last scatter.
int launch_group_inner(symaccel_batcher *b, Group *g) {
    symaccel_ctx *ctx = b->ctx;
    const PlaneSizes &ps = g->ps;
    if (!ctx->stage_in) SYM_GPU(ctx, hipStreamCreate(&ctx->stage_in));
    if (!ctx->stage_out) SYM_GPU(ctx, hipStreamCreate(&ctx->stage_out));
    hipStream_t s_in = ctx->stage_in, s_out = ctx->stage_out;
    // an upper bound of the copy pieces: every plane of every submission, rounded up
    size_t bound = 0;
    for (uint32_t id : g->ticket_ids) {
        const Ticket &t = b->tickets[id];
        for (int i = 0; i < ps.n_in; ++i) bound += pieces_of(ps.in[i] * (ps.in_per_ticket[i] ? 1 : t.n_chains));
        for (int i = 0; i < ps.n_state; ++i) bound += 2 * pieces_of(ps.state[i] * t.n_chains);
        bound += pieces_of(ps.out * t.n_chains);
    }
    bound += g->tickets + 8;  // (the unit list's pieces, one per chunk at most)
    if (g->kind == SYMACCEL_BATCH_VORBIS_SYNTH) bound += 2 * g->chains;  // (spectra and PCM go chain by chain, each rounded up)
    g->aac_pairs = g->aac_tns = 0;
    g->bad_blob = false;
    if (g->kind == SYMACCEL_BATCH_AAC_DECODE) {
        // what the submissions' blobs announce (a blob that does not fit its plane is read as "nothing": the group fails afterwards)
        for (uint32_t id : g->ticket_ids) {
            Ticket &t = b->tickets[id];
            AacBlobHeader *h = reinterpret_cast<AacBlobHeader *>(t.slot + slot_layout(ps, t.n_chains).in[2]);
            if (2 * (size_t)h->n_pairs > t.n_chains || aac_blob_bytes(h->n_pairs, g->units, h->n_tns) > ps.in[2] * t.n_chains) {
                g->bad_blob = true;
                h->n_pairs = h->n_tns = 0;
            }
            g->aac_pairs += h->n_pairs;
            g->aac_tns += h->n_tns;
        }
        bound += 3 * g->tickets + 8;  // (pair list, filters, TNS pair frames: one piece list each per chunk)
    }
    SYM_TRY(group_device(b, g, bound));
    BatchCopyDesc *descs = reinterpret_cast<BatchCopyDesc *>(g->h_desc);
    int32_t *h_units = reinterpret_cast<int32_t *>(g->h_desc + round256(bound * sizeof(BatchCopyDesc)));
    // AAC_DECODE: the group's pair list, filters and TNS pair frames with the indices the chunk's kernels want, built here
    int32_t *h_pairs = reinterpret_cast<int32_t *>(reinterpret_cast<char *>(h_units) + round256(g->tickets * 8));
    symaccel_aac_tns_filter *h_tns = reinterpret_cast<symaccel_aac_tns_filter *>(reinterpret_cast<char *>(h_pairs) + round256(g->aac_pairs * 8));
    uint32_t *h_pf = reinterpret_cast<uint32_t *>(reinterpret_cast<char *>(h_tns) + round256(g->aac_tns * sizeof(symaccel_aac_tns_filter)));
    size_t aac_p = 0, aac_f = 0, aac_q = 0;  // pairs / filters / TNS pair frames placed so far
    BatchCopyDesc *w = descs;
    const size_t per_chain = std::max<size_t>(1, in_bytes_per_chain(ps));
    // a third of the group per chunk, 8 .. 32 MiB of input: a chunk costs three launches and two event hops (~40 us), which 2 MiB
    // chunks (44 us on the link) did not amortise -- 22.7 GB/s each way at look-ahead 64 against 37.9 at 256 (profiles/r05c_*);
    // consecutive GROUPS overlap on the three streams anyway, so a small group is one chunk
    const size_t chunk_bytes = std::min<size_t>((size_t)32 << 20, std::max<size_t>((size_t)8 << 20, g->chains * per_chain / 3));
    const size_t chunk_chains = std::max<size_t>(1, chunk_bytes / per_chain);
    size_t t0 = 0, k = 0;
    while (t0 < g->tickets) {
        const size_t c0 = b->tickets[g->ticket_ids[t0]].first_chain;
        size_t t1 = t0, nc = 0;
        while (t1 < g->tickets && (nc == 0 || nc + b->tickets[g->ticket_ids[t1]].n_chains <= chunk_chains)) nc += b->tickets[g->ticket_ids[t1++]].n_chains;
        const size_t nt = t1 - t0;
        const int e = (int)(k & 1);
        // ---- gather: the submissions' planes into the chain-major device arrays
        BatchCopyDesc *g0 = w;
        const size_t chunk_p0 = aac_p, chunk_f0 = aac_f, chunk_q0 = aac_q;
        for (size_t ti = t0; ti < t1; ++ti) {
            const Ticket &t = b->tickets[g->ticket_ids[ti]];
            const SlotLayout l = slot_layout(ps, t.n_chains);
            for (int i = 0; i < ps.n_in; ++i) {
                if (g->kind == SYMACCEL_BATCH_AAC_DECODE && i == 2) {
                    // the blob is taken apart: joint-stereo rows go as they are, the pair list and the filters are re-based to the
                    // chunk (chain index relative to the chunk's first chain, pair index relative to its first pair)
                    const char *blob = t.slot + l.in[2];
                    const AacBlobHeader *h = reinterpret_cast<const AacBlobHeader *>(blob);
                    const int32_t rel = (int32_t)(t.first_chain - c0);
                    const int32_t *pc = reinterpret_cast<const int32_t *>(blob + aac_blob_pairs(h->n_pairs));
                    std::vector<int32_t> pair_of(t.n_chains, -1);
                    for (uint32_t q = 0; q < h->n_pairs; ++q) {
                        const int32_t a = pc[2 * q], bb = pc[2 * q + 1];
                        if (a < 0 || bb < 0 || (size_t)a >= t.n_chains || (size_t)bb >= t.n_chains || a == bb || pair_of[(size_t)a] >= 0 || pair_of[(size_t)bb] >= 0) {
                            g->bad_blob = true;  // (the pair keeps its place -- the js rows are laid out by pair -- as an inert self-less entry)
                            h_pairs[2 * (aac_p + q)] = rel;
                            h_pairs[2 * (aac_p + q) + 1] = rel;
                            continue;
                        }
                        pair_of[(size_t)a] = pair_of[(size_t)bb] = (int32_t)q;
                        h_pairs[2 * (aac_p + q)] = rel + a;
                        h_pairs[2 * (aac_p + q) + 1] = rel + bb;
                    }
                    add_pieces(w, blob + aac_blob_js(h->n_pairs), reinterpret_cast<char *>(g->d_aac_js + aac_p * g->units),
                               (size_t)h->n_pairs * g->units * sizeof(symaccel_aac_js_frame));
                    const symaccel_aac_tns_filter *tf = reinterpret_cast<const symaccel_aac_tns_filter *>(blob + aac_blob_tns(h->n_pairs, g->units));
                    for (uint32_t q = 0; q < h->n_tns; ++q) {
                        symaccel_aac_tns_filter f = tf[q];
                        const size_t chain = f.frame / g->units, frame = f.frame % g->units;
                        if (chain >= t.n_chains) f.frame = 0xffffffffu;  // (what symaccel_aac_tns_device skips)
                        else f.frame = (uint32_t)(((size_t)rel + chain) * g->units + frame);
                        h_tns[aac_f++] = f;
                        if (chain < t.n_chains && pair_of[chain] >= 0)  // a pair frame with TNS: joint stereo first, in place (list pass)
                            h_pf[aac_q++] = (uint32_t)((aac_p - chunk_p0 + (size_t)pair_of[chain]) * g->units + frame);
                    }
                    aac_p += h->n_pairs;
                    continue;
                }
                if (g->kind == SYMACCEL_BATCH_VORBIS_SYNTH && i == 0) {  // the packed spectrum: what the chain's blocks fill, not the plane
                    for (size_t c = 0; c < t.n_chains; ++c) {
                        size_t lines, samples;
                        vorbis_used(reinterpret_cast<const uint8_t *>(t.slot + l.in[1]) + c * g->units, g->units,
                                    reinterpret_cast<const int32_t *>(t.slot + l.state[0])[c], g->param & 255, (g->param >> 8) & 255, &lines, &samples);
                        add_pieces(w, t.slot + l.in[0] + c * ps.in[0], g->d_in[0] + ((size_t)t.first_chain + c) * ps.in[0], lines * 4);
                    }
                    continue;
                }
                add_pieces(w, t.slot + l.in[i], g->d_in[i] + (ps.in_per_ticket[i] ? ti : (size_t)t.first_chain) * ps.in[i], l.in_bytes[i]);
            }
            for (int i = 0; i < ps.n_state; ++i)
                add_pieces(w, t.slot + l.state[i], g->d_state_in[i] + (size_t)t.first_chain * ps.state[i], l.state_bytes[i]);
            if (g->kind == SYMACCEL_BATCH_MP3_DECODE) {  // (unit_chains are relative to the first chain of the chunk the submission falls into)
                const int32_t rel = (int32_t)(t.first_chain - c0);
                h_units[2 * ti] = rel;
                h_units[2 * ti + 1] = t.n_chains == 2 ? rel + 1 : -1;
            }
        }
        if (g->kind == SYMACCEL_BATCH_MP3_DECODE)
            add_pieces(w, reinterpret_cast<const char *>(h_units + 2 * t0), reinterpret_cast<char *>(g->d_units + 2 * t0), nt * 8);
        if (g->kind == SYMACCEL_BATCH_AAC_DECODE) {
            // (a pair frame listed twice -- both channels carry filters -- would be decoded twice: the list is made unique)
            std::sort(h_pf + chunk_q0, h_pf + aac_q);
            aac_q = (size_t)(std::unique(h_pf + chunk_q0, h_pf + aac_q) - h_pf);
            g->aac_chunk = {chunk_p0, aac_p - chunk_p0, chunk_f0, aac_f - chunk_f0, chunk_q0, aac_q - chunk_q0};
            add_pieces(w, reinterpret_cast<const char *>(h_pairs + 2 * chunk_p0), reinterpret_cast<char *>(g->d_aac_pairs + 2 * chunk_p0), (aac_p - chunk_p0) * 8);
            add_pieces(w, reinterpret_cast<const char *>(h_tns + chunk_f0), reinterpret_cast<char *>(g->d_aac_tns + chunk_f0),
                       (aac_f - chunk_f0) * sizeof(symaccel_aac_tns_filter));
            add_pieces(w, reinterpret_cast<const char *>(h_pf + chunk_q0), reinterpret_cast<char *>(g->d_aac_pf + chunk_q0), (aac_q - chunk_q0) * 4);
        }
        SYM_TRY(launch_batch_copy(ctx, s_in, g0, (size_t)(w - g0)));
        SYM_GPU(ctx, hipEventRecord(g->ev_in[e], s_in));
        SYM_GPU(ctx, hipStreamWaitEvent(ctx->stream, g->ev_in[e], 0));
        SYM_TRY(launch_chunk(b, g, c0, nc, t0, nt));
        SYM_GPU(ctx, hipEventRecord(g->ev_k[e], ctx->stream));
        SYM_GPU(ctx, hipStreamWaitEvent(s_out, g->ev_k[e], 0));
        // ---- scatter: PCM and the state after the batch back into the submissions' slots
        BatchCopyDesc *s0 = w;
        for (size_t ti = t0; ti < t1; ++ti) {
            const Ticket &t = b->tickets[g->ticket_ids[ti]];
            const SlotLayout l = slot_layout(ps, t.n_chains);
            if (g->kind == SYMACCEL_BATCH_VORBIS_SYNTH) {
                // (the state planes of the slot still hold the state BEFORE the batch here: the scatter that overwrites them is the
                // one being built)
                for (size_t c = 0; c < t.n_chains; ++c) {
                    size_t lines, samples;
                    vorbis_used(reinterpret_cast<const uint8_t *>(t.slot + l.in[1]) + c * g->units, g->units,
                                reinterpret_cast<const int32_t *>(t.slot + l.state[0])[c], g->param & 255, (g->param >> 8) & 255, &lines, &samples);
                    add_pieces(w, g->d_out + ((size_t)t.first_chain + c) * ps.out, t.slot + l.out + c * ps.out, samples * 4);
                }
            } else {
                add_pieces(w, g->d_out + (size_t)t.first_chain * ps.out, t.slot + l.out, l.out_bytes);
            }
            for (int i = 0; i < ps.n_state; ++i)
                add_pieces(w, g->d_state_out[i] + (size_t)t.first_chain * ps.state[i], t.slot + l.state[i], l.state_bytes[i]);
        }
        SYM_TRY(launch_batch_copy(ctx, s_out, s0, (size_t)(w - s0)));
        b->stats.chunks += 1;
        t0 = t1;
        ++k;
    }
    return SYMACCEL_OK;
}

// mu held.  Close the group, wait until every reservation of it is filled, launch.  On return the group is Launched (its status
// says whether the launch worked) -- or somebody else has launched it meanwhile.
void flush_group(symaccel_batcher *b, Group *g, std::unique_lock<std::mutex> &lock) {
    if (g->state != GroupState::Open) return;
    g->state = GroupState::Closed;
    b->cv.wait(lock, [&] { return g->uncommitted == 0; });
    if (g->state != GroupState::Closed) return;
    int st = SYMACCEL_OK;
    if (g->tickets) {
        DeviceGuard dev(b->ctx);
        st = dev.ok() ? launch_group_inner(b, g) : dev.status();
        if (st != SYMACCEL_OK) {
            // a launch that failed half way: nothing of this group may still be in flight when its slots are reused, and the
            // copy-out stream does not follow what the other two were left with -- drain all three (error path only)
            b->last_error = b->ctx->last_error;
            if (b->ctx->stage_in) (void)hipStreamSynchronize(b->ctx->stage_in);
            if (b->ctx->stream) (void)hipStreamSynchronize(b->ctx->stream);
            if (b->ctx->stage_out) (void)hipStreamSynchronize(b->ctx->stage_out);
        }
        // `done` sits behind the last scatter, which follows the last kernel, which follows the last gather
        if (dev.ok() && b->ctx->stage_out && g->done) (void)hipEventRecord(g->done, b->ctx->stage_out);
    }
    g->status = st != SYMACCEL_OK ? st : (g->bad_blob ? SYMACCEL_ERR_INVALID_ARG : SYMACCEL_OK);
    g->state = GroupState::Launched;
    b->stats.launches += 1;
    b->stats.chains_launched += g->chains;
    b->stats.max_chains_per_launch = std::max<uint64_t>(b->stats.max_chains_per_launch, g->chains);
    b->cv.notify_all();
}

Group *open_group(symaccel_batcher *b, int kind, int param, size_t units, const PlaneSizes &ps, size_t n_chains, int *st) {
    *st = SYMACCEL_OK;
    Group *spare = nullptr;
    for (auto &up : b->groups) {
        Group *g = up.get();
        if (g->state == GroupState::Open && g->kind == kind && g->param == param && g->units == units) return g;
        if (g->state == GroupState::Free && (!spare || g->d_bytes > spare->d_bytes)) spare = g;  // (the one with the most device memory)
    }
    if (!spare) {
        b->groups.emplace_back(new Group());
        spare = b->groups.back().get();
    }
    Group *g = spare;
    g->kind = kind;
    g->units = units;
    g->ps = ps;
    g->param = param;
    // what a group takes before it is launched unasked: flush_bytes of input
    g->cap_chains = std::max<size_t>(n_chains, std::max<size_t>(2, b->flush_bytes / std::max<size_t>(1, in_bytes_per_chain(ps))));
    g->chains = g->tickets = g->uncommitted = g->live = 0;
    g->ticket_ids.clear();
    g->status = SYMACCEL_OK;
    g->state = GroupState::Open;
    return g;
}

Ticket *find_ticket(symaccel_batcher *b, uint64_t id) {
    const uint32_t idx = (uint32_t)(id & 0xffffffffu), gen = (uint32_t)(id >> 32);
    if (idx >= b->tickets.size()) return nullptr;
    Ticket *t = &b->tickets[idx];
    return t->live && t->gen == gen ? t : nullptr;
}

void fill_slot(const Group *g, const Ticket *t, symaccel_batch_slot *slot) {
    const SlotLayout l = slot_layout(g->ps, t->n_chains);
    std::memset(slot, 0, sizeof(*slot));
    for (int i = 0; i < g->ps.n_in; ++i) {
        slot->input[i] = t->slot + l.in[i];
        slot->input_bytes[i] = l.in_bytes[i];
    }
    for (int i = 0; i < g->ps.n_state; ++i) {
        slot->state[i] = t->slot + l.state[i];
        slot->state_bytes[i] = l.state_bytes[i];
    }
    slot->out = t->slot + l.out;
    slot->out_bytes = l.out_bytes;
}

}  // namespace

extern "C" {

int symaccel_batcher_create(symaccel_ctx *ctx, size_t flush_bytes, symaccel_batcher **out) {
    if (!ctx || !out) return SYMACCEL_ERR_INVALID_ARG;
    *out = nullptr;
    symaccel_batcher *b = new (std::nothrow) symaccel_batcher();
    if (!b) return SYMACCEL_ERR_OOM;
    b->ctx = ctx;
    b->flush_bytes = flush_bytes ? flush_bytes : (size_t)64 << 20;
    b->hint_bytes = std::min<size_t>((size_t)4 << 20, b->flush_bytes / 8);
    if (const char *e = std::getenv("SYMACCEL_BATCHER_HINT_MB"))  // development knob (tools/gpu_r5g.sh): the hint threshold in MiB
        if (std::atoi(e) > 0) b->hint_bytes = (size_t)std::atoi(e) << 20;
    *out = b;
    return SYMACCEL_OK;
}

int symaccel_batcher_destroy(symaccel_batcher *b) {
    if (!b) return SYMACCEL_OK;
    {
        DeviceGuard dev(b->ctx);
        // nothing of ours may still be in flight when the staging memory goes
        if (b->ctx->stage_in) (void)hipStreamSynchronize(b->ctx->stage_in);
        if (b->ctx->stream) (void)hipStreamSynchronize(b->ctx->stream);
        if (b->ctx->stage_out) (void)hipStreamSynchronize(b->ctx->stage_out);
        for (auto &g : b->groups) group_free(g.get());
        for (auto &sl : b->slabs) (void)hipHostFree(sl.base);
    }
    delete b;
    return SYMACCEL_OK;
}

int symaccel_batcher_reserve(symaccel_batcher *b, int kind, int param, size_t n_chains, size_t units_per_chain, symaccel_batch_slot *slot,
                             uint64_t *ticket) {
    if (!b || !slot || !ticket || n_chains == 0 || units_per_chain == 0 || n_chains > 0x7fffffffu) return SYMACCEL_ERR_INVALID_ARG;
    PlaneSizes ps;
    if (kind == SYMACCEL_BATCH_AAC_SYNTH) param = 0;
    if (!plane_sizes(kind, param, units_per_chain, &ps)) return SYMACCEL_ERR_INVALID_ARG;
    if ((kind == SYMACCEL_BATCH_MP3_SYNTH || kind == SYMACCEL_BATCH_MP3_DECODE) && (param < 0 || param > 8)) return SYMACCEL_ERR_INVALID_ARG;  // sample_rate_idx
    if (kind == SYMACCEL_BATCH_MP3_DECODE && n_chains > 2) return SYMACCEL_ERR_INVALID_ARG;            // one stream per submission
    if (kind == SYMACCEL_BATCH_AAC_DECODE) {
        std::unique_lock<std::mutex> peek(b->mu);
        if (param < 0 || (size_t)param >= b->bands.size()) return SYMACCEL_ERR_INVALID_ARG;  // (symaccel_batcher_aac_bands first)
    }
    std::unique_lock<std::mutex> lock(b->mu);
    int st = SYMACCEL_OK;
    Group *g = open_group(b, kind, param, units_per_chain, ps, n_chains, &st);
    if (g && g->chains + n_chains > g->cap_chains) {  // full: it goes, a fresh one opens
        flush_group(b, g, lock);
        g = open_group(b, kind, param, units_per_chain, ps, n_chains, &st);
    }
    if (!g) return st;
    uint32_t idx;
    if (!b->free_tickets.empty()) {
        idx = b->free_tickets.back();
        b->free_tickets.pop_back();
    } else {
        idx = (uint32_t)b->tickets.size();
        b->tickets.emplace_back();
    }
    const SlotLayout lay = slot_layout(ps, n_chains);
    char *mem = nullptr;
    st = slot_alloc(b, lay.bytes, &mem);
    if (st != SYMACCEL_OK) {
        b->free_tickets.push_back(idx);
        return st;
    }
    Ticket *t = &b->tickets[idx];
    const uint32_t gen = t->gen + 1;
    *t = Ticket();
    t->gen = gen;
    t->group = g;
    t->first_chain = (uint32_t)g->chains;
    t->n_chains = (uint32_t)n_chains;
    t->ordinal = (uint32_t)g->tickets;
    t->live = true;
    t->slot = mem;
    t->slot_bytes = lay.bytes;
    g->ticket_ids.push_back(idx);
    g->chains += n_chains;
    g->tickets += 1;
    g->uncommitted += 1;
    g->live += 1;
    b->stats.submissions += 1;
    fill_slot(g, t, slot);
    *ticket = ((uint64_t)gen << 32) | idx;
    return SYMACCEL_OK;
}

int symaccel_batcher_commit(symaccel_batcher *b, uint64_t ticket) {
    if (!b) return SYMACCEL_ERR_INVALID_ARG;
    std::unique_lock<std::mutex> lock(b->mu);
    Ticket *t = find_ticket(b, ticket);
    if (!t || t->committed) return SYMACCEL_ERR_INVALID_ARG;
    t->committed = true;
    Group *g = t->group;
    g->uncommitted -= 1;
    if (g->uncommitted == 0) b->cv.notify_all();
    // enough input has piled up: to the device, nobody has to ask
    if (g->state == GroupState::Open && g->chains * in_bytes_per_chain(g->ps) >= b->flush_bytes) flush_group(b, g, lock);
    return SYMACCEL_OK;
}

int symaccel_batcher_flush(symaccel_batcher *b) {
    if (!b) return SYMACCEL_ERR_INVALID_ARG;
    std::unique_lock<std::mutex> lock(b->mu);
    for (size_t i = 0; i < b->groups.size(); ++i)  // (index loop: flush_group drops the lock while it waits for commits)
        if (b->groups[i]->state == GroupState::Open && b->groups[i]->tickets) flush_group(b, b->groups[i].get(), lock);
    return SYMACCEL_OK;
}

int symaccel_batcher_hint(symaccel_batcher *b) {
    if (!b) return SYMACCEL_ERR_INVALID_ARG;
    std::unique_lock<std::mutex> lock(b->mu);
    // "results will be wanted soon": whatever is worth a launch of its own goes now, so that the copies and the kernels run while
    // the callers are still busy with their current batches; a group below that size waits for more submissions (or for a waiter)
    const size_t worth = b->hint_bytes;
    for (size_t i = 0; i < b->groups.size(); ++i) {
        Group *g = b->groups[i].get();
        if (g->state == GroupState::Open && g->tickets && g->chains * in_bytes_per_chain(g->ps) >= worth) flush_group(b, g, lock);
    }
    return SYMACCEL_OK;
}

int symaccel_batcher_wait(symaccel_batcher *b, uint64_t ticket, symaccel_batch_slot *slot) {
    if (!b) return SYMACCEL_ERR_INVALID_ARG;
    Group *g;
    {
        std::unique_lock<std::mutex> lock(b->mu);
        Ticket *t = find_ticket(b, ticket);
        if (!t || !t->committed) return SYMACCEL_ERR_INVALID_ARG;
        g = t->group;
        if (g->state == GroupState::Open) {
            // somebody needs a result: everything pending goes now -- the waiter's group first, then its siblings of other shapes,
            // which would otherwise each cost their first waiter a round trip of their own
            flush_group(b, g, lock);
            for (size_t i = 0; i < b->groups.size(); ++i)
                if (b->groups[i]->state == GroupState::Open && b->groups[i]->tickets) flush_group(b, b->groups[i].get(), lock);
        }
        b->cv.wait(lock, [&] { return g->state == GroupState::Launched; });
        if (slot) fill_slot(g, find_ticket(b, ticket), slot);
    }
    if (g->tickets && g->done) {
        DeviceGuard dev(b->ctx);
        if (!dev.ok()) return dev.status();
        const hipError_t e = hipEventSynchronize(g->done);
        if (e != hipSuccess) return ctx_fail(b->ctx, e, "hipEventSynchronize(batch done)");
    }
    return g->status;
}

int symaccel_batcher_release(symaccel_batcher *b, uint64_t ticket) {
    if (!b) return SYMACCEL_ERR_INVALID_ARG;
    std::unique_lock<std::mutex> lock(b->mu);
    Ticket *t = find_ticket(b, ticket);
    if (!t) return SYMACCEL_ERR_INVALID_ARG;
    Group *g = t->group;
    if (!t->committed) {  // abandoned before it was filled: the slot's content is whatever it is, nobody reads the result
        t->committed = true;
        g->uncommitted -= 1;
        if (g->uncommitted == 0) b->cv.notify_all();
    }
    // The slot goes back to the pool -- but the group's copies may still be reading or writing it (a release without a wait, or
    // before the launch): the group is launched if it has not been, and drained, first.  (The common order -- wait, read, release --
    // finds the event signalled.)
    if (g->state == GroupState::Open) flush_group(b, g, lock);
    b->cv.wait(lock, [&] { return g->state == GroupState::Launched; });
    t = find_ticket(b, ticket);  // (the table may have grown while the lock was dropped)
    if (!t) return SYMACCEL_ERR_INVALID_ARG;
    if (g->tickets && g->done) {
        hipEvent_t done = g->done;
        lock.unlock();
        {
            DeviceGuard dev(b->ctx);
            if (dev.ok()) (void)hipEventSynchronize(done);
        }
        lock.lock();
        t = find_ticket(b, ticket);
        if (!t) return SYMACCEL_ERR_INVALID_ARG;
    }
    slot_free(b, t->slot, t->slot_bytes);
    t->slot = nullptr;
    t->live = false;
    b->free_tickets.push_back((uint32_t)(ticket & 0xffffffffu));
    g->live -= 1;
    if (g->live == 0 && g->state == GroupState::Launched) g->state = GroupState::Free;
    return SYMACCEL_OK;
}

int symaccel_batcher_plane_bytes(int kind, int param, size_t units_per_chain, size_t *in_bytes, size_t *state_bytes, size_t *out_bytes) {
    PlaneSizes ps;
    if (!plane_sizes(kind, param, units_per_chain, &ps)) return SYMACCEL_ERR_INVALID_ARG;
    for (int i = 0; i < kMaxIn; ++i)
        if (in_bytes) in_bytes[i] = ps.in[i];
    for (int i = 0; i < kMaxState; ++i)
        if (state_bytes) state_bytes[i] = ps.state[i];
    if (out_bytes) *out_bytes = ps.out;
    return SYMACCEL_OK;
}

int symaccel_batcher_submit(symaccel_batcher *b, int kind, int param, size_t n_chains, size_t units_per_chain, const void **in,
                            void **state_io, void *out, uint64_t *ticket) {
    if (!b || !in || !state_io || !out || !ticket) return SYMACCEL_ERR_INVALID_ARG;
    if (kind == SYMACCEL_BATCH_AAC_DECODE) return SYMACCEL_ERR_INVALID_ARG;  // (symaccel_batcher_submit_aac_decode writes the blob)
    PlaneSizes ps;
    if (!plane_sizes(kind, kind == SYMACCEL_BATCH_AAC_SYNTH ? 0 : param, units_per_chain, &ps)) return SYMACCEL_ERR_INVALID_ARG;
    for (int i = 0; i < ps.n_in; ++i)
        if (!in[i] && !(kind == SYMACCEL_BATCH_MP3_DECODE && i == 3 && n_chains == 1)) return SYMACCEL_ERR_INVALID_ARG;
    for (int i = 0; i < ps.n_state; ++i)
        if (!state_io[i]) return SYMACCEL_ERR_INVALID_ARG;
    symaccel_batch_slot slot;
    uint64_t id = 0;
    SYM_TRY(symaccel_batcher_reserve(b, kind, param, n_chains, units_per_chain, &slot, &id));
    for (int i = 0; i < ps.n_in; ++i) {
        if (in[i])
            std::memcpy(slot.input[i], in[i], slot.input_bytes[i]);
        else
            std::memset(slot.input[i], 0, slot.input_bytes[i]);
    }
    for (int i = 0; i < ps.n_state; ++i) std::memcpy(slot.state[i], state_io[i], slot.state_bytes[i]);
    {
        std::unique_lock<std::mutex> lock(b->mu);
        Ticket *t = find_ticket(b, id);
        for (int i = 0; i < ps.n_state; ++i) t->user_state[i] = state_io[i];
        t->user_out = out;
    }
    *ticket = id;
    return symaccel_batcher_commit(b, id);
}

int symaccel_batcher_submit_aac_synth(symaccel_batcher *b, const float *coeffs, const uint8_t *side, float *delay_io, float *pcm, size_t n_chains,
                                      size_t frames_per_chain, uint64_t *ticket) {
    const void *in[4] = {coeffs, side, nullptr, nullptr};
    void *st[3] = {delay_io, nullptr, nullptr};
    return symaccel_batcher_submit(b, SYMACCEL_BATCH_AAC_SYNTH, 0, n_chains, frames_per_chain, in, st, pcm, ticket);
}

int symaccel_batcher_submit_mp3_synth(symaccel_batcher *b, const float *xr, const symaccel_mp3_side *side, int sample_rate_idx, float *overlap_io,
                                      float *vvec_io, int32_t *vfront_io, float *pcm, size_t n_chains, size_t granules_per_chain, uint64_t *ticket) {
    const void *in[4] = {xr, side, nullptr, nullptr};
    void *st[3] = {overlap_io, vvec_io, vfront_io};
    return symaccel_batcher_submit(b, SYMACCEL_BATCH_MP3_SYNTH, sample_rate_idx, n_chains, granules_per_chain, in, st, pcm, ticket);
}

int symaccel_batcher_submit_mp3_decode(symaccel_batcher *b, const int16_t *quant, const symaccel_mp3_requant *rq_desc, const symaccel_mp3_stereo *st_desc,
                                       const symaccel_mp3_side *side, int sample_rate_idx, float *overlap_io, float *vvec_io, int32_t *vfront_io,
                                       float *pcm, size_t n_chains, size_t granules_per_chain, uint64_t *ticket) {
    const void *in[4] = {quant, rq_desc, side, st_desc};
    void *st[3] = {overlap_io, vvec_io, vfront_io};
    return symaccel_batcher_submit(b, SYMACCEL_BATCH_MP3_DECODE, sample_rate_idx, n_chains, granules_per_chain, in, st, pcm, ticket);
}

int symaccel_batcher_aac_bands(symaccel_batcher *b, const uint16_t *swb_long, int n_swb_long, const uint16_t *swb_short, int n_swb_short, int *bands) {
    if (!b || !bands || !swb_long || !swb_short || n_swb_long < 1 || n_swb_short < 1 || n_swb_long > 63 || n_swb_short > 15) return SYMACCEL_ERR_INVALID_ARG;
    symaccel_batcher::Bands nb;
    if (!aac_band_maps(swb_long, n_swb_long, swb_short, n_swb_short, &nb.maps)) return SYMACCEL_ERR_INVALID_ARG;
    nb.swb_long.assign(swb_long, swb_long + n_swb_long + 1);
    nb.swb_short.assign(swb_short, swb_short + n_swb_short + 1);
    std::unique_lock<std::mutex> lock(b->mu);
    for (size_t i = 0; i < b->bands.size(); ++i)
        if (b->bands[i].swb_long == nb.swb_long && b->bands[i].swb_short == nb.swb_short) {
            *bands = (int)i;
            return SYMACCEL_OK;
        }
    if (b->bands.size() >= 64) return SYMACCEL_ERR_UNSUPPORTED;
    b->bands.push_back(std::move(nb));
    *bands = (int)b->bands.size() - 1;
    return SYMACCEL_OK;
}

int symaccel_batcher_submit_aac_decode(symaccel_batcher *b, int bands, const float *coeffs, const uint8_t *side, const int32_t *pair_chains,
                                       const symaccel_aac_js_frame *js_desc, size_t n_pairs, const symaccel_aac_tns_filter *tns, size_t n_tns,
                                       float *delay_io, float *pcm, size_t n_chains, size_t frames_per_chain, uint64_t *ticket) {
    if (!b || !coeffs || !side || !delay_io || !pcm || !ticket || n_chains == 0 || frames_per_chain == 0) return SYMACCEL_ERR_INVALID_ARG;
    if ((n_pairs && (!pair_chains || !js_desc)) || (n_tns && !tns) || 2 * n_pairs > n_chains) return SYMACCEL_ERR_INVALID_ARG;
    {  // every chain in at most one pair, inside the submission (what symaccel_aac_decode_pipelined checks)
        std::vector<uint8_t> seen(n_chains, 0);
        for (size_t i = 0; i < 2 * n_pairs; ++i) {
            const int32_t c = pair_chains[i];
            if (c < 0 || (size_t)c >= n_chains || seen[(size_t)c]) return SYMACCEL_ERR_INVALID_ARG;
            seen[(size_t)c] = 1;
        }
    }
    PlaneSizes ps;
    if (!plane_sizes(SYMACCEL_BATCH_AAC_DECODE, bands, frames_per_chain, &ps)) return SYMACCEL_ERR_INVALID_ARG;
    if (aac_blob_bytes(n_pairs, frames_per_chain, n_tns) > ps.in[2] * n_chains) return SYMACCEL_ERR_INVALID_ARG;  // (more than 8 filters per channel-frame)
    symaccel_batch_slot slot;
    uint64_t id = 0;
    SYM_TRY(symaccel_batcher_reserve(b, SYMACCEL_BATCH_AAC_DECODE, bands, n_chains, frames_per_chain, &slot, &id));
    std::memcpy(slot.input[0], coeffs, slot.input_bytes[0]);
    std::memcpy(slot.input[1], side, slot.input_bytes[1]);
    char *blob = static_cast<char *>(slot.input[2]);
    AacBlobHeader h{(uint32_t)n_pairs, (uint32_t)n_tns, {0, 0}};
    std::memcpy(blob, &h, sizeof h);
    if (n_pairs) {
        std::memcpy(blob + aac_blob_pairs(n_pairs), pair_chains, n_pairs * 8);
        std::memcpy(blob + aac_blob_js(n_pairs), js_desc, n_pairs * frames_per_chain * sizeof(symaccel_aac_js_frame));
    }
    if (n_tns) std::memcpy(blob + aac_blob_tns(n_pairs, frames_per_chain), tns, n_tns * sizeof(symaccel_aac_tns_filter));
    std::memcpy(slot.state[0], delay_io, slot.state_bytes[0]);
    {
        std::unique_lock<std::mutex> lock(b->mu);
        Ticket *t = find_ticket(b, id);
        t->user_state[0] = delay_io;
        t->user_out = pcm;
    }
    *ticket = id;
    return symaccel_batcher_commit(b, id);
}

int symaccel_batcher_submit_vorbis_synth(symaccel_batcher *b, int bs0_exp, int bs1_exp, const float *spectra, const uint8_t *block_flag,
                                         int32_t *prev_flag_io, float *overlap_io, float *pcm, size_t n_chains, size_t blocks_per_chain,
                                         uint64_t *ticket) {
    if (bs0_exp < 0 || bs0_exp > 255 || bs1_exp < 0 || bs1_exp > 255) return SYMACCEL_ERR_INVALID_ARG;
    const void *in[4] = {spectra, block_flag, nullptr, nullptr};
    void *st[3] = {prev_flag_io, overlap_io, nullptr};
    return symaccel_batcher_submit(b, SYMACCEL_BATCH_VORBIS_SYNTH, bs0_exp | (bs1_exp << 8), n_chains, blocks_per_chain, in, st, pcm, ticket);
}

int symaccel_batcher_collect(symaccel_batcher *b, uint64_t ticket) {
    if (!b) return SYMACCEL_ERR_INVALID_ARG;
    symaccel_batch_slot slot;
    void *user_state[kMaxState], *user_out;
    {
        std::unique_lock<std::mutex> lock(b->mu);
        Ticket *t = find_ticket(b, ticket);
        if (!t || !t->user_out) return SYMACCEL_ERR_INVALID_ARG;
        for (int i = 0; i < kMaxState; ++i) user_state[i] = t->user_state[i];
        user_out = t->user_out;
    }
    const int st = symaccel_batcher_wait(b, ticket, &slot);
    if (st == SYMACCEL_OK) {
        std::memcpy(user_out, slot.out, slot.out_bytes);
        for (int i = 0; i < kMaxState; ++i)
            if (user_state[i] && slot.state[i]) std::memcpy(user_state[i], slot.state[i], slot.state_bytes[i]);
    }
    const int rel = symaccel_batcher_release(b, ticket);
    return st != SYMACCEL_OK ? st : rel;
}

int symaccel_batcher_abandon(symaccel_batcher *b, uint64_t ticket) {
    if (!b) return SYMACCEL_ERR_INVALID_ARG;
    return symaccel_batcher_release(b, ticket);  // (release drains the ticket's group before the slot goes back)
}

int symaccel_batcher_get_stats(symaccel_batcher *b, symaccel_batcher_stats *out) {
    if (!b || !out) return SYMACCEL_ERR_INVALID_ARG;
    std::unique_lock<std::mutex> lock(b->mu);
    *out = b->stats;
    uint64_t pending = 0;
    for (auto &g : b->groups)
        if (g->state == GroupState::Open || g->state == GroupState::Closed) pending += g->tickets;
    out->pending = pending;
    return SYMACCEL_OK;
}

}  // extern "C"
