// Cross-stream batcher: many decoders, one launch.
//
// A `Hip*Decoder` (bindings/rust) or a `codecs::LookaheadDecoder` (include/symaccel.hpp) batches the look-ahead of ITS stream:
// 256 packets x 2 channels is 2 MiB per call, and N decoders are N small calls, each paying its own launch and its own PCIe
// round trip.  The reference's trait gives a decoder no view of its siblings (AudioDecoder::decode_ref sees one packet of one
// track, symphonia-core/src/codecs/audio.rs:279-297; the registry builds decoders from (params, opts) alone, registry.rs:330-341),
// so the coalescing point is below the trait, here: decoders SUBMIT their batches to a batcher shared by the process and come
// back for the result later; whatever is pending when somebody needs a result (or when `flush_bytes` of input have piled up)
// goes to the device as ONE batch per (kind, param, units per chain) group -- the chains of every submission side by side in the
// chain-major layout the kernels already take, so nothing in the kernels knows about streams.
//
//   reserve()  -> a slot of page-locked staging memory the front end writes its spectra / samples / records into (no copy
//                 between the parser's output and the DMA source) + a ticket
//   commit()   -> the slot is filled
//   wait()     -> launches the ticket's group if nobody has yet (and everything else that is pending), blocks until the
//                 group's results are in page-locked memory; slot.out / slot.state now hold PCM and the carried state
//   release()  -> the slot may be reused
//
// A group is transferred and transformed in chunks of submissions: gather(c + 1) || kernel(c) || scatter(c - 1) on the three
// streams of a LANE (the chunks are chains, which are independent: no carried state between chunks, unlike stage.cpp's
// frame-axis chunks).  A lane is a context of its own (stream, scratch, tables) plus two copy streams and a mutex; a group that
// closes is handed to the next lane and ENQUEUED OUTSIDE the batcher's mutex -- the threads of other streams keep reserving,
// committing and collecting while one thread builds the copy descriptors and launches, and the gather of one group overlaps the
// scatter of another on a different lane.  Groups are pooled: in the steady state nothing is allocated.  Thread-safe: submissions
// may come from any thread; a flush waits for reservations of the group that are still being filled.  The status of a launch is
// kept PER TICKET: a submission whose descriptors do not add up fails alone, its neighbours in the launch succeed.  A context that
// has a batcher is driven through the batcher only (the context itself is externally synchronised, include/symaccel.h "Thread safety").
#include <algorithm>
#include <chrono>
#include <condition_variable>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <mutex>
#include <thread>
#include <vector>

#include "symaccel_internal.h"

using namespace symaccel;

namespace {

constexpr int kMaxIn = 6, kMaxState = 3;  // (symaccel_batch_slot's input[] / state[])
constexpr size_t kVorbisPosts = 65;      // floor1_Y values per channel-block in a VORBIS_DECODE submission (floor.rs:510-520)

// what a kind's planes weigh: bytes per chain (per ticket for `in_per_ticket`, per `in_div` chains otherwise) for `units` frames /
// granules / blocks / words per chain
struct PlaneSizes {
    size_t in[kMaxIn] = {0, 0, 0, 0, 0, 0};
    bool in_per_ticket[kMaxIn] = {false, false, false, false, false, false};
    bool in_host_only[kMaxIn] = {false, false, false, false, false, false};  // read by the host when the group is launched; never copied as it is
    uint8_t in_div[kMaxIn] = {1, 1, 1, 1, 1, 1};                             // 2: one element per channel PAIR (chains 2p, 2p + 1)
    size_t state[kMaxState] = {0, 0, 0};
    size_t out = 0;
    bool in_place = false;  // the result overwrites input[0] (FLAC / ALAC: the entry points they stand for work in place)
    int n_in = 0, n_state = 0;
};

// coupling steps a block may carry in a VORBIS_DECODE submission: every ordered channel pair once, at least 8 (a mapping may list up to
// 256 steps, lib.rs:604-640 -- a stream with more than this per block keeps batching per stream through symaccel_vorbis_decode)
inline size_t vorbis_max_steps(size_t nch) { return std::min<size_t>(256, std::max<size_t>(8, nch * (nch - 1))); }

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
        ps->in_host_only[2] = true;
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
    case SYMACCEL_BATCH_VORBIS_DECODE: {  // symaccel_vorbis_decode for ONE stream: residue, flags, floor index, posts, coupling blob | prev, overlap | pcm
        const int e0 = param & 255, e1 = (param >> 8) & 255, nch = (param >> 16) & 255;
        if (e0 < 6 || e1 > 13 || e0 > e1 || nch < 1 || (param >> 24)) return false;
        const size_t half = (size_t)1 << (e1 - 1);
        ps->n_in = 5;
        ps->in[0] = units * half * 4;
        ps->in[1] = units;
        ps->in[2] = units;  // floor configuration of every channel-block (symaccel_batcher_vorbis_floor's index), or ..._FLOOR_UNUSED
        ps->in_host_only[2] = true;
        ps->in[3] = units * kVorbisPosts * 4;
        ps->in_host_only[3] = true;
        // the coupling steps of the stream's blocks: first[units + 1] u32, padded to 16 bytes, then (magnitude, angle) byte pairs
        ps->in[4] = (((units + 1) * 4 + 15) & ~(size_t)15) + units * 2 * vorbis_max_steps((size_t)nch);
        ps->in_per_ticket[4] = true;
        ps->in_host_only[4] = true;
        ps->n_state = 2;
        ps->state[0] = 4;
        ps->state[1] = half * 4;
        ps->out = units * half * 4;
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
    case SYMACCEL_BATCH_FLAC_RESTORE:  // symaccel_flac_restore(_stereo_device): buf (in place), desc, coeffs [, pair_mode]; units = block size
        if (units > 65535 || (param & ~0x11f)) return false;  // frame.rs:58 (u16 block size); param = 0, or 0x100 | out_shift
        ps->n_in = (param & 0x100) ? 4 : 3;
        ps->in[0] = units * 4;
        ps->in[1] = sizeof(symaccel_flac_desc);
        ps->in[2] = 32 * 4;
        if (param & 0x100) {
            ps->in[3] = 1;
            ps->in_div[3] = 2;
        }
        ps->in_place = true;
        return true;
    case SYMACCEL_BATCH_ALAC_PREDICT:  // symaccel_alac_predict(_stereo_device): buf (in place), desc, coeffs [, pair_weight, pair_shift]
        if (units > 0x3fffffffu || (param & ~0x100)) return false;
        ps->n_in = (param & 0x100) ? 5 : 3;
        ps->in[0] = units * 4;
        ps->in[1] = sizeof(symaccel_alac_desc);
        ps->in[2] = 32 * 4;
        if (param & 0x100) {
            ps->in[3] = 4;
            ps->in_div[3] = 2;
            ps->in[4] = 1;
            ps->in_div[4] = 2;
        }
        ps->in_place = true;
        return true;
    default:
        return false;
    }
}

size_t round256(size_t v) { return (v + 255) & ~(size_t)255; }

// What a launch costs is not the same for every kind: the FLAC / ALAC kernels walk a block's recurrence with ONE lane (4096 samples at
// ~120 ns each: half a millisecond whatever the launch holds), so their groups are worth launching only when they are large -- four
// times the input of the transform kinds before a group goes unasked, and a hint launches nothing below `flush_bytes`
// (measured: 0.39 -> 0.74 M packets/s of 4096-sample stereo frames at 256 streams, profiles/r06k_decoders.jsonl).
inline bool serial_kind(int kind) { return kind == SYMACCEL_BATCH_FLAC_RESTORE || kind == SYMACCEL_BATCH_ALAC_PREDICT; }
inline size_t flush_threshold(size_t flush_bytes, int kind) { return serial_kind(kind) ? 4 * flush_bytes : flush_bytes; }
inline size_t hint_threshold(size_t hint_bytes, size_t flush_bytes, int kind) { return serial_kind(kind) ? std::max(hint_bytes, 2 * flush_bytes) : hint_bytes; }

inline size_t plane_bytes(const PlaneSizes &ps, int i, size_t n_chains) {
    return ps.in_per_ticket[i] ? ps.in[i] : ps.in[i] * (n_chains / ps.in_div[i]);
}

size_t in_bytes_per_chain(const PlaneSizes &ps) {
    size_t s = 0;
    for (int i = 0; i < ps.n_in; ++i)
        if (!ps.in_per_ticket[i]) s += ps.in[i] / ps.in_div[i];
    return s;
}

// a submission's page-locked slot: [in 0 | .. | state 0 | .. | out], every plane on a 256-byte boundary (in place: out IS in 0)
struct SlotLayout {
    size_t in[kMaxIn] = {}, state[kMaxState] = {}, out = 0, bytes = 0;
    size_t in_bytes[kMaxIn] = {}, state_bytes[kMaxState] = {}, out_bytes = 0;
};
SlotLayout slot_layout(const PlaneSizes &ps, size_t n_chains) {
    SlotLayout l;
    size_t off = 0;
    for (int i = 0; i < ps.n_in; ++i) {
        l.in[i] = off;
        l.in_bytes[i] = plane_bytes(ps, i, n_chains);
        off += round256(l.in_bytes[i]);
    }
    for (int i = 0; i < ps.n_state; ++i) {
        l.state[i] = off;
        l.state_bytes[i] = ps.state[i] * n_chains;
        off += round256(l.state_bytes[i]);
    }
    if (ps.in_place) {
        l.out = l.in[0];
        l.out_bytes = l.in_bytes[0];
    } else {
        l.out = off;
        l.out_bytes = ps.out * n_chains;
        off += round256(l.out_bytes);
    }
    l.bytes = off;
    return l;
}

// Slots are pooled by size CLASS, not by exact size: eight classes per octave (at most 12.5 % of padding), so the odd shapes of a
// running service -- tail batches at the end of a stream, short look-ahead batches, MP3 granule totals -- reuse each other's memory
// instead of each carving a slab of its own.
size_t slot_class(size_t bytes) {
    if (bytes <= 4096) return 4096;
    size_t top = (size_t)1 << (63 - __builtin_clzll((unsigned long long)bytes));
    const size_t step = top >> 3;
    return (bytes + step - 1) & ~(step - 1);
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

// The coupling blob of a VORBIS_DECODE submission (plane in[4]): first[units + 1] u32 -- the steps of block b are
// [first[b], first[b + 1]) --, padded to 16 bytes, then the steps as (magnitude channel, angle channel) byte pairs
inline size_t vorbis_blob_steps(size_t units) { return ((units + 1) * 4 + 15) & ~(size_t)15; }

struct Group;

using Clock = std::chrono::steady_clock;

struct Ticket {
    Group *group = nullptr;
    uint32_t gen = 0;
    uint32_t first_chain = 0, n_chains = 0, ordinal = 0;
    bool live = false, committed = false;
    Clock::time_point committed_at{};
    int status = SYMACCEL_OK;  // of THIS submission, once its group is launched
    char *slot = nullptr;      // page-locked, slot_layout(group's planes, n_chains)
    size_t slot_bytes = 0;     // (the size class it was carved for)
    // copy form (symaccel_batcher_submit): where collect() puts the results
    void *user_state[kMaxState] = {nullptr, nullptr, nullptr};
    void *user_out = nullptr;
};

// what a launch needs of a submission: copied out of the ticket table when the group closes, because the launch runs outside the
// batcher's mutex and the table may grow meanwhile
struct TicketView {
    char *slot = nullptr;
    uint32_t first_chain = 0, n_chains = 0;
    int status = SYMACCEL_OK;
};

enum class GroupState { Free, Open, Closed, Launching, Launched };

struct Lane;
struct Group;

// The device side of a launch: one allocation (cut into the planes of whatever group is launched with it), the page-locked copy
// descriptors and lists, the events of the chunk pipeline.  Blocks are POOLED apart from the groups: a group stays around until the
// last of its submissions is released -- the decoders hold their current batch for as long as they hand it out -- but the device
// memory is free again as soon as the scatter has finished, so a handful of blocks serve any number of groups in flight.
struct Block {
    char *d_base = nullptr;
    size_t d_bytes = 0;
    char *h_desc = nullptr;
    size_t h_desc_bytes = 0;
    hipEvent_t ev_in[2] = {nullptr, nullptr}, ev_k[2] = {nullptr, nullptr};
    // completion: a word of page-locked memory the launch's last kernel writes its sequence number into (batch_flag_kernel); the
    // numbers of a block only grow, so "flag >= seq" stays true for a launch that has finished however often the block is reused
    uint64_t *h_flag = nullptr;
    uint64_t seq = 0;
    Group *owner = nullptr;  // the group whose launch is (or was last) using it; nullptr: never used or given back
};

inline uint64_t read_flag(const uint64_t *flag) { return __atomic_load_n(flag, __ATOMIC_ACQUIRE); }

// One launch: the submissions of one shape that were pending together.  Host side: the submissions' own slots.  Device side: a
// Block, cut at launch time (when the number of chains is known).
struct Group {
    int kind = 0, param = 0;
    size_t units = 0;
    PlaneSizes ps;
    size_t cap_chains = 0;  // reservations accepted before the group is launched and a fresh one opened
    size_t chains = 0, tickets = 0, uncommitted = 0, live = 0;
    Clock::time_point launched_at{};  // (enqueued: statistics)
    GroupState state = GroupState::Free;
    int status = SYMACCEL_OK;          // of the launch as a whole (a device error fails every ticket)
    std::vector<uint32_t> ticket_ids;  // the submissions, in order (index into symaccel_batcher::tickets)
    std::vector<TicketView> views;     // ... as the launch sees them
    Lane *lane = nullptr;              // where it was (is being) enqueued
    Block *block = nullptr;            // its device side, until somebody has seen the launch complete
    bool completed = false;            // its launch has been seen complete (or failed and was drained): nothing of it is in flight
    const uint64_t *done_flag = nullptr;  // where its launch reports completion, and the number that means "this launch"
    uint64_t done_seq = 0;
    char *d_in[kMaxIn] = {}, *d_state_in[kMaxState] = {}, *d_state_out[kMaxState] = {}, *d_out = nullptr;
    size_t row_pitch = 0;  // FLAC_RESTORE / ALAC_PREDICT: bytes between the chains (rows) of d_in[0] when the device plane is padded (symaccel_row_stride), 0 = rows back to back
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
    AacBandMaps aac_maps{};              // the band tables `param` names (copied when the group closes)
    // VORBIS_DECODE: the byte plane of the floor curves, the floor1_Y rows and line offsets by (configuration, block size) class, the
    // blocks' line offsets, the channel-blocks without a floor, the coupling steps
    uint8_t *d_vb_plane = nullptr, *d_vb_kill = nullptr, *d_vb_steps = nullptr;
    uint32_t *d_vb_ys = nullptr, *d_vb_offs = nullptr, *d_vb_boff = nullptr, *d_vb_first = nullptr;
    size_t vb_steps = 0;                 // coupling steps of the group (counted when it is launched)
    std::vector<symaccel_vorbis_floor1_cfg> vb_floors;  // the registered configurations (copied when the group closes)
    struct VbClass {
        uint32_t cfg, n2;
        size_t ys0, offs0, count;
    };
    struct {
        std::vector<VbClass> classes;
        size_t boff0, kill0, first0, steps0, n_steps;
        bool prepare;
    } vb_chunk{};
};

// One pipeline: a context (kernel stream, scratch, tables) and two copy streams.  Lane 0 is the caller's context; the others are the
// batcher's own.  `mu` serialises the enqueues of a lane (a context is externally synchronised).
struct Lane {
    symaccel_ctx *ctx = nullptr;
    bool owned = false;
    std::mutex mu;
};

inline uint64_t ns_since(Clock::time_point t0) { return (uint64_t)std::chrono::duration_cast<std::chrono::nanoseconds>(Clock::now() - t0).count(); }

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
    size_t hint_bytes = 0;  // what a group must hold for a hint to launch it ...
    size_t busy_groups = 4, busy_hint_bytes = (size_t)48 << 20;  // ... and while at least `busy_groups` launches are in flight
    std::vector<std::unique_ptr<Lane>> lanes;
    size_t want_lanes = 2, next_lane = 0;
    std::vector<std::unique_ptr<Block>> blocks;
    // page-locked slot memory: slabs, carved into slots by size class; a released slot goes to the free list of its class (the
    // shapes of a running service repeat: in the steady state nothing is allocated)
    struct Slab {
        char *base;
        size_t bytes, used;
    };
    std::vector<Slab> slabs;
    uint64_t slots_live = 0;
    bool growing = false;  // a caller is page-locking a new slab (outside the mutex): the others wait for it instead of adding their own
    std::vector<std::pair<size_t, std::vector<char *>>> free_slots;
    // AAC_DECODE: the scale-factor-band tables a stream's joint-stereo descriptors refer to, registered once per stream shape
    // (symaccel_batcher_aac_bands); a submission names its table by index (`param`), which is part of the group key
    struct Bands {
        std::vector<uint16_t> swb_long, swb_short;
        AacBandMaps maps;
    };
    std::vector<Bands> bands;
    // VORBIS_DECODE: the floor-1 configurations the submissions' floor planes index (symaccel_batcher_vorbis_floor)
    std::vector<symaccel_vorbis_floor1_cfg> floors;
};

namespace {

constexpr size_t kSlabBytes = (size_t)32 << 20;

// (the batcher's mutex, with the time spent waiting for it on the books: symaccel_batcher_stats::mutex_wait_ns)
struct Locked {
    std::unique_lock<std::mutex> lock;
    explicit Locked(symaccel_batcher *b) : lock(b->mu, std::defer_lock) {
        if (!lock.try_lock()) {
            const Clock::time_point t0 = Clock::now();
            lock.lock();
            b->stats.mutex_wait_ns += ns_since(t0);
            b->stats.mutex_contended += 1;
        }
    }
};

// mu held on entry and on return; DROPPED while a new slab is page-locked (hipHostMalloc of tens of MiB takes milliseconds: held, it
// stalled every other caller -- 1.1 to 1.6 s of summed mutex wait in a 0.4 s run, profiles/r06c_decoders.jsonl).  Slabs double in
// size up to 128 MiB, so a service that grows does so in a few steps.
int slot_alloc(symaccel_batcher *b, std::unique_lock<std::mutex> &lock, size_t bytes, char **out) {
    for (;;) {
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
        if (b->growing) {  // somebody is page-locking a slab right now: its memory will do for this caller too
            b->cv.wait(lock, [&] { return !b->growing; });
            continue;
        }
        static const size_t slab0 = [] {  // (test knob: the first slab's size in KiB)
            const char *e = std::getenv("SYMACCEL_BATCHER_SLAB_KB");
            return e && std::atol(e) > 0 ? (size_t)std::atol(e) << 10 : kSlabBytes;
        }();
        size_t want = slab0;
        if (!b->slabs.empty()) want = std::min<size_t>(4 * slab0, 2 * b->slabs.back().bytes);
        want = std::max(want, bytes);
        void *h = nullptr;
        b->growing = true;
        lock.unlock();
        int st = SYMACCEL_OK;
        {
            DeviceGuard dev(b->ctx);
            if (!dev.ok()) st = dev.status();
            else if (hipHostMalloc(&h, want, hipHostMallocDefault) != hipSuccess) {
                (void)hipGetLastError();
                st = SYMACCEL_ERR_OOM;
            }
        }
        lock.lock();
        b->growing = false;
        b->cv.notify_all();
        if (st != SYMACCEL_OK) return st;
        b->slabs.push_back({static_cast<char *>(h), want, 0});
        b->stats.staging_bytes += want;
    }
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

void block_free(Block *k) {
    if (k->d_base) (void)hipFree(k->d_base);
    if (k->h_desc) (void)hipHostFree(k->h_desc);
    for (hipEvent_t e : {k->ev_in[0], k->ev_in[1], k->ev_k[0], k->ev_k[1]})
        if (e) (void)hipEventDestroy(e);
    if (k->h_flag) (void)hipHostFree(k->h_flag);
    k->h_flag = nullptr;
    k->d_base = nullptr;
    k->h_desc = nullptr;
}

// Sizes of the lists a launch builds on the host (page-locked, behind the copy descriptors) and mirrors on the device
struct ListSizes {
    size_t units = 0;                                      // MP3_DECODE
    size_t aac_pairs = 0, aac_tns = 0, aac_pf = 0;         // AAC_DECODE
    size_t vb_boff = 0, vb_kill = 0, vb_first = 0, vb_steps = 0, vb_ys = 0, vb_offs = 0;  // VORBIS_DECODE
};
ListSizes list_sizes(const Group *g) {
    ListSizes s;
    s.units = round256(g->tickets * 8);
    if (g->kind == SYMACCEL_BATCH_AAC_DECODE) {
        s.aac_pairs = round256(std::max<size_t>(1, g->aac_pairs) * 8);
        s.aac_tns = round256(std::max<size_t>(1, g->aac_tns) * sizeof(symaccel_aac_tns_filter));
        s.aac_pf = round256(std::max<size_t>(1, g->aac_tns) * 4);
    }
    if (g->kind == SYMACCEL_BATCH_VORBIS_DECODE) {
        s.vb_boff = round256(g->tickets * (g->units + 1) * 4);
        s.vb_kill = round256(g->chains * g->units);
        s.vb_first = round256((g->tickets * g->units + g->tickets + 1) * 4);  // (every chunk's list starts with a 0 of its own)
        s.vb_steps = round256(std::max<size_t>(1, g->vb_steps) * 2);
        s.vb_ys = round256(g->chains * g->units * kVorbisPosts * 4);
        s.vb_offs = round256(g->chains * g->units * 4);
    }
    return s;
}

// the device side of a closed group: sized for the chains it holds, planes carved out; grown (never shrunk) across reuses
int group_device(symaccel_ctx *ctx, Group *g, size_t n_pieces_bound, uint64_t *n_allocs) {
    const PlaneSizes &ps = g->ps;
    Block *blk = g->block;
    const ListSizes ls = list_sizes(g);
    size_t total = 0, off_in[kMaxIn] = {}, off_si[kMaxState], off_so[kMaxState], off_out = 0, off_units;
    for (int i = 0; i < ps.n_in; ++i) {
        off_in[i] = total;
        if (ps.in_host_only[i]) continue;
        total += round256(ps.in_per_ticket[i] ? ps.in[i] * g->tickets : (i == 0 && g->row_pitch ? g->row_pitch : ps.in[i]) * ((g->chains + ps.in_div[i] - 1) / ps.in_div[i]));
    }
    for (int i = 0; i < ps.n_state; ++i) {
        off_si[i] = total;
        total += round256(ps.state[i] * g->chains);
        off_so[i] = total;
        total += round256(ps.state[i] * g->chains);
    }
    if (!ps.in_place) {
        off_out = total;
        total += round256(ps.out * g->chains);
    }
    off_units = total;
    total += ls.units;
    size_t off_ap = 0, off_aj = 0, off_at = 0, off_af = 0, off_ai = 0;
    if (g->kind == SYMACCEL_BATCH_AAC_DECODE) {
        off_ap = total;
        total += ls.aac_pairs;
        off_aj = total;
        total += round256(std::max<size_t>(1, g->aac_pairs) * g->units * sizeof(symaccel_aac_js_frame));
        off_at = total;
        total += ls.aac_tns;
        off_af = total;
        total += ls.aac_pf;
        off_ai = total;
        total += round256(aac_js_scratch_bytes(g->chains, g->aac_pairs, g->units));
    }
    size_t off_vp = 0, off_vy = 0, off_vo = 0, off_vb = 0, off_vk = 0, off_vf = 0, off_vs = 0;
    if (g->kind == SYMACCEL_BATCH_VORBIS_DECODE) {
        off_vp = total;
        total += round256(g->chains * (g->units << (((g->param >> 8) & 255) - 1)));  // one byte per line
        off_vy = total;
        total += ls.vb_ys;
        off_vo = total;
        total += ls.vb_offs;
        off_vb = total;
        total += ls.vb_boff;
        off_vk = total;
        total += ls.vb_kill;
        off_vf = total;
        total += ls.vb_first;
        off_vs = total;
        total += ls.vb_steps;
    }
    // (hipFree waits for the whole device, hipMalloc is not cheap either: a group's memory is sized for what the group can hold at
    // most -- cap_chains, i.e. flush_bytes of input -- the first time, so that groups of fewer submissions never have to grow later;
    // measured before: 0.87 ms of host time per launch with four caller threads, profiles/r06b_decoders.jsonl)
    const size_t scale_num = std::max(g->cap_chains, g->chains), scale_den = std::max<size_t>(1, g->chains);
    if (total > blk->d_bytes) {
        if (blk->d_base) SYM_GPU(ctx, hipFree(blk->d_base));
        blk->d_base = nullptr;
        blk->d_bytes = 0;
        void *d = nullptr;
        const size_t want = std::max(total + total / 4, (total / scale_den + 1) * scale_num + ((size_t)1 << 20));
        SYM_TRY(ctx_alloc(ctx, &d, want, false));
        blk->d_base = static_cast<char *>(d);
        blk->d_bytes = want;
        *n_allocs += 1;
    }
    char *const d_base = blk->d_base;
    for (int i = 0; i < ps.n_in; ++i) g->d_in[i] = d_base + off_in[i];
    for (int i = 0; i < ps.n_state; ++i) {
        g->d_state_in[i] = d_base + off_si[i];
        g->d_state_out[i] = d_base + off_so[i];
    }
    g->d_out = ps.in_place ? g->d_in[0] : d_base + off_out;
    g->d_units = reinterpret_cast<int32_t *>(d_base + off_units);
    if (g->kind == SYMACCEL_BATCH_AAC_DECODE) {
        g->d_aac_pairs = reinterpret_cast<int32_t *>(d_base + off_ap);
        g->d_aac_js = reinterpret_cast<symaccel_aac_js_frame *>(d_base + off_aj);
        g->d_aac_tns = reinterpret_cast<symaccel_aac_tns_filter *>(d_base + off_at);
        g->d_aac_pf = reinterpret_cast<uint32_t *>(d_base + off_af);
        g->d_aac_index = d_base + off_ai;
    }
    if (g->kind == SYMACCEL_BATCH_VORBIS_DECODE) {
        g->d_vb_plane = reinterpret_cast<uint8_t *>(d_base + off_vp);
        g->d_vb_ys = reinterpret_cast<uint32_t *>(d_base + off_vy);
        g->d_vb_offs = reinterpret_cast<uint32_t *>(d_base + off_vo);
        g->d_vb_boff = reinterpret_cast<uint32_t *>(d_base + off_vb);
        g->d_vb_kill = reinterpret_cast<uint8_t *>(d_base + off_vk);
        g->d_vb_first = reinterpret_cast<uint32_t *>(d_base + off_vf);
        g->d_vb_steps = reinterpret_cast<uint8_t *>(d_base + off_vs);
    }
    // (behind the descriptors: the lists of list_sizes())
    const size_t desc_bytes = round256(n_pieces_bound * sizeof(BatchCopyDesc)) + ls.units + ls.aac_pairs + ls.aac_tns + ls.aac_pf + ls.vb_boff +
                              ls.vb_kill + ls.vb_first + ls.vb_steps + ls.vb_ys + ls.vb_offs;
    if (desc_bytes > blk->h_desc_bytes) {
        if (blk->h_desc) (void)hipHostFree(blk->h_desc);
        blk->h_desc = nullptr;
        blk->h_desc_bytes = 0;
        void *h = nullptr;
        const size_t want = std::max(desc_bytes + desc_bytes / 4, (desc_bytes / scale_den + 1) * scale_num + 65536);
        if (hipHostMalloc(&h, want, hipHostMallocDefault) != hipSuccess) {
            (void)hipGetLastError();
            return SYMACCEL_ERR_OOM;
        }
        blk->h_desc = static_cast<char *>(h);
        blk->h_desc_bytes = want;
        *n_allocs += 1;
    }
    for (hipEvent_t *e : {&blk->ev_in[0], &blk->ev_in[1], &blk->ev_k[0], &blk->ev_k[1]})
        if (!*e) SYM_GPU(ctx, hipEventCreateWithFlags(e, hipEventDisableTiming));
    return SYMACCEL_OK;
}

// the kernels of one chunk: chains [c0, c0 + nc), submissions [t0, t0 + nt)
int launch_chunk(symaccel_ctx *ctx, Group *g, size_t c0, size_t nc, size_t t0, size_t nt) {
    const PlaneSizes &ps = g->ps;
    auto in = [&](int i) { return g->d_in[i] + (ps.in_per_ticket[i] ? t0 * ps.in[i] : (c0 / ps.in_div[i]) * (i == 0 && g->row_pitch ? g->row_pitch : ps.in[i])); };
    auto si = [&](int i) { return g->d_state_in[i] + c0 * ps.state[i]; };
    auto so = [&](int i) { return g->d_state_out[i] + c0 * ps.state[i]; };
    char *out = g->d_out + c0 * (ps.in_place ? ps.in[0] : ps.out);
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
        const AacBandMaps &maps = g->aac_maps;
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
    case SYMACCEL_BATCH_VORBIS_DECODE: {
        // symaccel_vorbis_decode's kernel sequence (csrc/ctx.cpp) on the chunk: the coupling steps and the zero floors in place
        // (lib.rs:250-278, 206-209), the floor curves as one byte per line -- the (configuration, block size) classes two per launch
        // (floor.rs:568-653, 776-825) --, then the synthesis with table[y] * residue in its load path (lib.rs:282-292, dsp.rs:68-126)
        const int e0 = g->param & 255, e1 = (g->param >> 8) & 255, nch = (g->param >> 16) & 255;
        const size_t cap = g->units << (e1 - 1);
        const auto &ch = g->vb_chunk;
        uint8_t *plane = g->d_vb_plane + c0 * cap;
        SYM_GPU(ctx, hipMemsetAsync(plane, 0, nc * cap, ctx->stream));
        if (ch.prepare)
            SYM_TRY(launch_vorbis_prepare(ctx, (float *)in(0), cap, (unsigned)nch, nt, g->units, g->d_vb_boff + ch.boff0, g->d_vb_steps + 2 * ch.steps0,
                                          g->d_vb_first + ch.first0, g->d_vb_kill + ch.kill0));
        {
            std::vector<symaccel_vorbis_floor1_job> jobs;  // two classes per launch
            jobs.reserve(ch.classes.size());
            for (const Group::VbClass &k : ch.classes) {
                const symaccel_vorbis_floor1_cfg &cfg = g->vb_floors[k.cfg];
                jobs.push_back(symaccel_vorbis_floor1_job{cfg.x_list, cfg.n_posts, cfg.multiplier, g->d_vb_ys + k.ys0, k.n2, g->d_vb_offs + k.offs0, k.count});
            }
            SYM_TRY(symaccel_vorbis_floor1_y_jobs_device(ctx, jobs.data(), jobs.size(), plane));
        }
        return symaccel_vorbis_synth_fy_pp_device(ctx, e0, e1, plane, (const float *)in(0), cap, (const uint8_t *)in(1), (const int32_t *)si(0),
                                                  (int32_t *)so(0), (const float *)si(1), (float *)so(1), (float *)out, cap, nc, g->units);
    }
    case SYMACCEL_BATCH_FLAC_RESTORE:  // decoder.rs:663-752 (+ :32-82, :239-242 with the pair modes)
        return launch_flac_restore(ctx, (int32_t *)in(0), (const symaccel_flac_desc *)in(1), (const int32_t *)in(2), nc, g->units,
                                   (g->param & 0x100) ? (const uint8_t *)in(3) : nullptr, (uint32_t)(g->param & 31), g->row_pitch / 4);
    case SYMACCEL_BATCH_ALAC_PREDICT:  // alac/lib.rs:165-264 (+ :664-671 with the pair parameters)
        return launch_alac_predict(ctx, (int32_t *)in(0), (const symaccel_alac_desc *)in(1), (const int32_t *)in(2), nc, g->units,
                                   (g->param & 0x100) ? (const int32_t *)in(3) : nullptr, (g->param & 0x100) ? (const uint8_t *)in(4) : nullptr, g->row_pitch / 4);
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

// ---- what a submission's descriptors say, judged alone (its neighbours in the launch are not failed for it)

int check_aac_blob(const PlaneSizes &ps, size_t units, TicketView &v) {
    AacBlobHeader *h = reinterpret_cast<AacBlobHeader *>(v.slot + slot_layout(ps, v.n_chains).in[2]);
    if (2 * (size_t)h->n_pairs > v.n_chains || aac_blob_bytes(h->n_pairs, units, h->n_tns) > ps.in[2] * v.n_chains) return SYMACCEL_ERR_INVALID_ARG;
    const int32_t *pc = reinterpret_cast<const int32_t *>(reinterpret_cast<const char *>(h) + aac_blob_pairs(h->n_pairs));
    std::vector<uint8_t> seen(v.n_chains, 0);
    for (uint32_t q = 0; q < 2 * h->n_pairs; ++q) {
        const int32_t c = pc[q];
        if (c < 0 || (size_t)c >= v.n_chains || seen[(size_t)c]) return SYMACCEL_ERR_INVALID_ARG;
        seen[(size_t)c] = 1;
    }
    return SYMACCEL_OK;
}

int check_flac(const PlaneSizes &ps, size_t units, int param, const TicketView &v) {
    const SlotLayout l = slot_layout(ps, v.n_chains);
    const symaccel_flac_desc *d = reinterpret_cast<const symaccel_flac_desc *>(v.slot + l.in[1]);
    for (size_t c = 0; c < v.n_chains; ++c) {  // what symaccel_flac_restore checks (decoder.rs:361, 456-458, 506-508)
        if (d[c].kind > SYMACCEL_FLAC_LPC || d[c].order > units || d[c].shift > 31 || d[c].wasted_bits > 31) return SYMACCEL_ERR_INVALID_ARG;
        if (d[c].kind == SYMACCEL_FLAC_FIXED && d[c].order > 4) return SYMACCEL_ERR_INVALID_ARG;
        if (d[c].kind == SYMACCEL_FLAC_LPC && (d[c].order < 1 || d[c].order > 32)) return SYMACCEL_ERR_INVALID_ARG;
    }
    if (param & 0x100) {
        const uint8_t *pm = reinterpret_cast<const uint8_t *>(v.slot + l.in[3]);
        for (size_t p = 0; p < v.n_chains / 2; ++p)
            if (pm[p] > 3) return SYMACCEL_ERR_INVALID_ARG;
    }
    return SYMACCEL_OK;
}

int check_alac(const PlaneSizes &ps, int param, const TicketView &v) {
    const SlotLayout l = slot_layout(ps, v.n_chains);
    const symaccel_alac_desc *d = reinterpret_cast<const symaccel_alac_desc *>(v.slot + l.in[1]);
    for (size_t c = 0; c < v.n_chains; ++c) {
        if (d[c].mode > 0 && d[c].mode < 15) return SYMACCEL_ERR_DECODE;  // lib.rs:167-169, "alac: invalid mode" (symaccel_alac_block_status_device)
        if (d[c].lpc_order > 31 || d[c].shift > 31 || d[c].bps < 1 || d[c].bps > 32) return SYMACCEL_ERR_INVALID_ARG;
    }
    if (param & 0x100) {
        const uint8_t *sh = reinterpret_cast<const uint8_t *>(v.slot + l.in[4]);
        for (size_t p = 0; p < v.n_chains / 2; ++p)
            if (sh[p] > 31) return SYMACCEL_ERR_INVALID_ARG;  // lib.rs:555
    }
    return SYMACCEL_OK;
}

// a VORBIS_DECODE submission (what symaccel_vorbis_decode checks of a stream); *steps = its coupling steps
int check_vorbis(const Group *g, const TicketView &v, size_t *steps) {
    const PlaneSizes &ps = g->ps;
    const size_t nb = g->units, nch = v.n_chains;
    const SlotLayout l = slot_layout(ps, nch);
    const uint8_t *flags = reinterpret_cast<const uint8_t *>(v.slot + l.in[1]);
    const uint8_t *floor = reinterpret_cast<const uint8_t *>(v.slot + l.in[2]);
    const uint32_t *posts = reinterpret_cast<const uint32_t *>(v.slot + l.in[3]);
    const int32_t *prev = reinterpret_cast<const int32_t *>(v.slot + l.state[0]);
    *steps = 0;
    // the channels of a stream share their block flags and their previous flag (one mode per packet, lib.rs:170-178)
    for (size_t c = 1; c < nch; ++c) {
        if (prev[c] != prev[0]) return SYMACCEL_ERR_INVALID_ARG;
        for (size_t b = 0; b < nb; ++b)
            if ((flags[c * nb + b] != 0) != (flags[b] != 0)) return SYMACCEL_ERR_INVALID_ARG;
    }
    for (size_t cb = 0; cb < nch * nb; ++cb) {
        const unsigned f = floor[cb];
        if (f == SYMACCEL_VORBIS_FLOOR_UNUSED) continue;
        if (f >= g->vb_floors.size()) return SYMACCEL_ERR_INVALID_ARG;
        const uint32_t *y = posts + cb * kVorbisPosts;
        for (unsigned i = 0; i < g->vb_floors[f].n_posts; ++i)
            if (y[i] > 511u) return SYMACCEL_ERR_UNSUPPORTED;  // (symaccel_vorbis_floor1_status_device's domain)
    }
    const uint32_t *first = reinterpret_cast<const uint32_t *>(v.slot + l.in[4]);
    const uint8_t *st = reinterpret_cast<const uint8_t *>(v.slot + l.in[4] + vorbis_blob_steps(nb));
    if (first[0] != 0) return SYMACCEL_ERR_INVALID_ARG;
    for (size_t b = 0; b < nb; ++b)
        if (first[b + 1] < first[b]) return SYMACCEL_ERR_INVALID_ARG;
    const size_t n = first[nb];
    if (vorbis_blob_steps(nb) + 2 * n > ps.in[4]) return SYMACCEL_ERR_INVALID_ARG;
    for (size_t s = 0; s < n; ++s)
        if (st[2 * s] >= nch || st[2 * s + 1] >= nch || st[2 * s] == st[2 * s + 1]) return SYMACCEL_ERR_INVALID_ARG;  // lib.rs:253
    *steps = n;
    return SYMACCEL_OK;
}

// Everything of a closed group: per chunk of submissions ONE gather launch (slots -> HBM, the kernels' chain-major layout), the
// synthesis kernel(s), ONE scatter launch (HBM -> slots); `done` is recorded behind the last scatter.  Runs on `lane`, outside the
// batcher's mutex: nothing of the batcher but the group itself (and the slots its views point at) is touched.
int launch_group_inner(Lane *lane, Group *g, uint64_t *n_chunks, uint64_t *api_ns, uint64_t *n_allocs) {
    symaccel_ctx *ctx = lane->ctx;
    const PlaneSizes &ps = g->ps;
    if (!ctx->stage_in) SYM_GPU(ctx, hipStreamCreate(&ctx->stage_in));
    if (!ctx->stage_out) SYM_GPU(ctx, hipStreamCreate(&ctx->stage_out));
    hipStream_t s_in = ctx->stage_in, s_out = ctx->stage_out;
    std::vector<TicketView> &views = g->views;
    // an upper bound of the copy pieces: every plane of every submission, rounded up
    size_t bound = 0;
    for (const TicketView &v : views) {
        for (int i = 0; i < ps.n_in; ++i)
            if (!ps.in_host_only[i]) bound += pieces_of(plane_bytes(ps, i, v.n_chains));
        for (int i = 0; i < ps.n_state; ++i) bound += 2 * pieces_of(ps.state[i] * v.n_chains);
        bound += pieces_of((ps.in_place ? ps.in[0] : ps.out) * v.n_chains);
    }
    bound += g->tickets + 8;  // (the unit list's pieces, one per chunk at most)
    // FLAC / ALAC: the device plane's rows at the pitch the lane-per-block kernels run fastest at (symaccel_row_stride: rows 4 / 8 / 16 / 32 KiB apart -- the
    // 4096-sample blocks of nearly every stream -- put a wavefront's 64 row segments on a fraction of the HBM channels); the slots stay compact, the
    // gather / scatter go row by row (development knob: SYMACCEL_BATCH_ROW_PAD=0 keeps the rows back to back)
    static const bool row_pad = [] {
        const char *e = std::getenv("SYMACCEL_BATCH_ROW_PAD");
        return !e || std::atol(e) != 0;
    }();
    g->row_pitch = 0;
    if (serial_kind(g->kind) && row_pad && symaccel_row_stride(g->units) * 4 != ps.in[0]) {
        g->row_pitch = symaccel_row_stride(g->units) * 4;
        bound += 2 * g->chains;  // (each row rounded up)
    }
    if (g->kind == SYMACCEL_BATCH_VORBIS_SYNTH || g->kind == SYMACCEL_BATCH_VORBIS_DECODE) bound += 2 * g->chains;  // (spectra and PCM go chain by chain, each rounded up)
    g->aac_pairs = g->aac_tns = 0;
    g->vb_steps = 0;
    // ---- the submissions' own descriptors, each judged alone: one that does not add up is neutralised (it runs as an empty
    // description, its ticket fails) and the rest of the launch goes ahead
    if (g->kind == SYMACCEL_BATCH_AAC_DECODE) {
        for (TicketView &v : views) {
            AacBlobHeader *h = reinterpret_cast<AacBlobHeader *>(v.slot + slot_layout(ps, v.n_chains).in[2]);
            v.status = check_aac_blob(ps, g->units, v);
            if (v.status == SYMACCEL_OK && h->n_pairs && g->param < 0) v.status = SYMACCEL_ERR_INVALID_ARG;  // (pairs need a band table)
            if (v.status != SYMACCEL_OK) h->n_pairs = h->n_tns = 0;
            g->aac_pairs += h->n_pairs;
            g->aac_tns += h->n_tns;
        }
        bound += 3 * g->tickets + 8;  // (pair list, filters, TNS pair frames: one piece list each per chunk)
        for (const TicketView &v : views) bound += pieces_of(ps.in[2] * v.n_chains);  // (the joint-stereo rows of the blob)
    }
    if (g->kind == SYMACCEL_BATCH_FLAC_RESTORE)
        for (TicketView &v : views) v.status = check_flac(ps, g->units, g->param, v);
    if (g->kind == SYMACCEL_BATCH_ALAC_PREDICT)
        for (TicketView &v : views) v.status = check_alac(ps, g->param, v);
    if (g->kind == SYMACCEL_BATCH_VORBIS_DECODE) {
        for (TicketView &v : views) {
            size_t steps = 0;
            v.status = check_vorbis(g, v, &steps);
            g->vb_steps += steps;
        }
        const ListSizes ls = list_sizes(g);
        bound += 6 * (g->tickets + 8) + pieces_of(ls.vb_boff) + pieces_of(ls.vb_kill) + pieces_of(ls.vb_first) + pieces_of(ls.vb_steps) + pieces_of(ls.vb_ys) +
                 pieces_of(ls.vb_offs);
    }
    SYM_TRY(group_device(ctx, g, bound, n_allocs));
    const ListSizes ls = list_sizes(g);
    Block *blk = g->block;
    BatchCopyDesc *descs = reinterpret_cast<BatchCopyDesc *>(blk->h_desc);
    char *lists = blk->h_desc + round256(bound * sizeof(BatchCopyDesc));
    auto carve = [&](size_t bytes) {
        char *p = lists;
        lists += bytes;
        return p;
    };
    int32_t *h_units = reinterpret_cast<int32_t *>(carve(ls.units));
    // AAC_DECODE: the group's pair list, filters and TNS pair frames with the indices the chunk's kernels want, built here
    int32_t *h_pairs = reinterpret_cast<int32_t *>(carve(ls.aac_pairs));
    symaccel_aac_tns_filter *h_tns = reinterpret_cast<symaccel_aac_tns_filter *>(carve(ls.aac_tns));
    uint32_t *h_pf = reinterpret_cast<uint32_t *>(carve(ls.aac_pf));
    size_t aac_p = 0, aac_f = 0, aac_q = 0;  // pairs / filters / TNS pair frames placed so far
    // VORBIS_DECODE: line offsets of the blocks, channel-blocks without a floor, coupling steps, floor1_Y rows and line offsets by class
    uint32_t *h_boff = reinterpret_cast<uint32_t *>(carve(ls.vb_boff));
    uint8_t *h_kill = reinterpret_cast<uint8_t *>(carve(ls.vb_kill));
    uint32_t *h_first = reinterpret_cast<uint32_t *>(carve(ls.vb_first));
    uint8_t *h_steps = reinterpret_cast<uint8_t *>(carve(ls.vb_steps));
    uint32_t *h_ys = reinterpret_cast<uint32_t *>(carve(ls.vb_ys));
    uint32_t *h_offs = reinterpret_cast<uint32_t *>(carve(ls.vb_offs));
    size_t vb_first_at = 0, vb_steps_at = 0, vb_ys_at = 0, vb_offs_at = 0;
    BatchCopyDesc *w = descs;
    // A bulk plane (a submission's spectra, its PCM) of `dma_bytes` or more goes through a copy ENGINE (hipMemcpyAsync on the lane's copy
    // stream) instead of the piece list: the engines move large PCIe payloads, a kernel's 64-byte accesses pay a header per 64 bytes
    // in both directions -- two kernels copying against each other reached 30 + 30 GB/s, the engines 44 + 44 (profiles/r06d_*).  The
    // small planes of a chunk (records, state, lists) still share ONE gather / scatter launch.
    struct Dma {
        const char *src;
        char *dst;
        size_t bytes;
    };
    std::vector<Dma> dma;
    static const size_t dma_bytes = [] {  // development knob: SYMACCEL_BATCH_DMA_KB (0 = everything through the copy kernels)
        const char *e = std::getenv("SYMACCEL_BATCH_DMA_KB");
        return e ? (size_t)std::atol(e) << 10 : (size_t)0;
    }();
    auto bulk = [&](const char *src, char *dst, size_t bytes) {
        if (dma_bytes && bytes >= dma_bytes) dma.push_back({src, dst, bytes});
        else add_pieces(w, src, dst, bytes);
    };
    const size_t per_chain = std::max<size_t>(1, in_bytes_per_chain(ps));
    // half of the group per chunk, 8 .. 32 MiB of input: a chunk costs three launches and two event hops (~40 us), which 2 MiB
    // chunks (44 us on the link) did not amortise -- 22.7 GB/s each way at look-ahead 64 against 37.9 at 256 (profiles/r05c_*);
    // consecutive GROUPS overlap on the lanes anyway, so a small group is one chunk (one or two chunks measure the same, three or
    // six are slower: profiles/r06z4_big_groups.jsonl, r06z5_copy_grid.jsonl)
    // (development knobs: SYMACCEL_BATCH_CHUNKS = chunks a full group is cut into, SYMACCEL_BATCH_CHUNK_MIN_KB = the smallest chunk)
    static const size_t chunk_div = [] {
        const char *e = std::getenv("SYMACCEL_BATCH_CHUNKS");
        const long v = e ? std::atol(e) : 2;
        return (size_t)(v < 1 ? 1 : (v > 64 ? 64 : v));
    }();
    static const size_t chunk_min = [] {
        const char *e = std::getenv("SYMACCEL_BATCH_CHUNK_MIN_KB");
        const long v = e ? std::atol(e) : 8192;
        return (size_t)(v < 64 ? 64 : v) << 10;
    }();
    const size_t chunk_bytes = std::min<size_t>((size_t)32 << 20, std::max<size_t>(chunk_min, g->chains * per_chain / chunk_div));
    const size_t chunk_chains = std::max<size_t>(1, chunk_bytes / per_chain);
    size_t t0 = 0, k = 0;
    while (t0 < g->tickets) {
        const size_t c0 = views[t0].first_chain;
        size_t t1 = t0, nc = 0;
        while (t1 < g->tickets && (nc == 0 || nc + views[t1].n_chains <= chunk_chains)) nc += views[t1++].n_chains;
        const size_t nt = t1 - t0;
        const int e = (int)(k & 1);
        // ---- gather: the submissions' planes into the chain-major device arrays
        BatchCopyDesc *g0 = w;
        const size_t chunk_p0 = aac_p, chunk_f0 = aac_f, chunk_q0 = aac_q;
        for (size_t ti = t0; ti < t1; ++ti) {
            const TicketView &t = views[ti];
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
                        const int32_t a = pc[2 * q], bb = pc[2 * q + 1];  // (in range and distinct: check_aac_blob)
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
                if (ps.in_host_only[i]) continue;
                if ((g->kind == SYMACCEL_BATCH_VORBIS_SYNTH || g->kind == SYMACCEL_BATCH_VORBIS_DECODE) && i == 0) {
                    // the packed spectrum: what the chain's blocks fill, not the plane
                    for (size_t c = 0; c < t.n_chains; ++c) {
                        size_t lines, samples;
                        vorbis_used(reinterpret_cast<const uint8_t *>(t.slot + l.in[1]) + c * g->units, g->units,
                                    reinterpret_cast<const int32_t *>(t.slot + l.state[0])[c], g->param & 255, (g->param >> 8) & 255, &lines, &samples);
                        bulk(t.slot + l.in[0] + c * ps.in[0], g->d_in[0] + ((size_t)t.first_chain + c) * ps.in[0], lines * 4);
                    }
                    continue;
                }
                if (i == 0 && g->row_pitch) {  // compact rows of the slot -> padded rows of the device plane
                    for (size_t c = 0; c < t.n_chains; ++c) bulk(t.slot + l.in[0] + c * ps.in[0], g->d_in[0] + ((size_t)t.first_chain + c) * g->row_pitch, ps.in[0]);
                    continue;
                }
                bulk(t.slot + l.in[i], g->d_in[i] + (ps.in_per_ticket[i] ? ti : (size_t)t.first_chain / ps.in_div[i]) * ps.in[i], l.in_bytes[i]);
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
        if (g->kind == SYMACCEL_BATCH_VORBIS_DECODE) {
            // the chunk's streams in symaccel_vorbis_decode's terms: where every block's lines start, which channel-blocks have no
            // floor, the coupling steps block by block, and per (floor configuration, block size) class the floor1_Y rows and the
            // byte offsets of their lines in the chunk's plane
            const int e0 = g->param & 255, e1 = (g->param >> 8) & 255;
            const size_t nb = g->units, cap = nb << (e1 - 1), nch = (size_t)((g->param >> 16) & 255);
            auto &ch = g->vb_chunk;
            ch.classes.clear();
            ch.boff0 = t0 * (nb + 1);
            ch.kill0 = c0 * nb;
            ch.first0 = vb_first_at;
            ch.steps0 = vb_steps_at;
            bool any_kill = false;
            std::vector<uint32_t> count(512, 0);
            h_first[vb_first_at] = 0;
            for (size_t ti = t0; ti < t1; ++ti) {
                const TicketView &t = views[ti];
                const SlotLayout l = slot_layout(ps, t.n_chains);
                const uint8_t *flags = reinterpret_cast<const uint8_t *>(t.slot + l.in[1]);
                const uint8_t *floor = reinterpret_cast<const uint8_t *>(t.slot + l.in[2]);
                uint32_t *boff = h_boff + ti * (nb + 1);
                size_t lines = 0;
                for (size_t b = 0; b < nb; ++b) {
                    boff[b] = (uint32_t)lines;
                    lines += (size_t)1 << ((flags[b] ? e1 : e0) - 1);
                }
                boff[nb] = (uint32_t)lines;
                const bool ok = t.status == SYMACCEL_OK;
                for (size_t c = 0; c < nch; ++c)
                    for (size_t b = 0; b < nb; ++b) {
                        const unsigned f = ok ? floor[c * nb + b] : SYMACCEL_VORBIS_FLOOR_UNUSED;
                        const bool kill = f == SYMACCEL_VORBIS_FLOOR_UNUSED;
                        h_kill[((size_t)t.first_chain + c) * nb + b] = kill ? 1 : 0;
                        any_kill |= kill;
                        if (!kill) count[2 * f + (flags[b] ? 1 : 0)] += 1;
                    }
                // the steps: block by block behind the chunk's list (a failed submission has none)
                const uint32_t *first = reinterpret_cast<const uint32_t *>(t.slot + l.in[4]);
                const uint8_t *st = reinterpret_cast<const uint8_t *>(t.slot + l.in[4] + vorbis_blob_steps(nb));
                uint32_t *out_first = h_first + vb_first_at + (ti - t0) * nb;
                const uint32_t base = out_first[0];
                for (size_t b = 0; b < nb; ++b) out_first[b + 1] = base + (ok ? first[b + 1] : 0);
                if (ok && first[nb]) std::memcpy(h_steps + 2 * (vb_steps_at + base), st, 2 * (size_t)first[nb]);
            }
            ch.n_steps = h_first[vb_first_at + nt * nb];
            ch.prepare = ch.n_steps != 0 || any_kill;
            // classes in (configuration, block size) order, their rows behind each other
            std::vector<size_t> ys_at(512, 0), offs_at(512, 0);
            for (size_t kc = 0; kc < 512; ++kc) {
                if (!count[kc]) continue;
                const symaccel_vorbis_floor1_cfg &cfg = g->vb_floors[kc / 2];
                ys_at[kc] = vb_ys_at;
                offs_at[kc] = vb_offs_at;
                ch.classes.push_back({(uint32_t)(kc / 2), (uint32_t)1 << ((kc & 1 ? e1 : e0) - 1), vb_ys_at, vb_offs_at, count[kc]});
                vb_ys_at += (size_t)count[kc] * cfg.n_posts;
                vb_offs_at += count[kc];
            }
            const size_t ys_begin = ch.classes.empty() ? vb_ys_at : ch.classes.front().ys0, offs_begin = ch.classes.empty() ? vb_offs_at : ch.classes.front().offs0;
            for (size_t ti = t0; ti < t1; ++ti) {
                const TicketView &t = views[ti];
                if (t.status != SYMACCEL_OK) continue;
                const SlotLayout l = slot_layout(ps, t.n_chains);
                const uint8_t *flags = reinterpret_cast<const uint8_t *>(t.slot + l.in[1]);
                const uint8_t *floor = reinterpret_cast<const uint8_t *>(t.slot + l.in[2]);
                const uint32_t *posts = reinterpret_cast<const uint32_t *>(t.slot + l.in[3]);
                const uint32_t *boff = h_boff + ti * (nb + 1);
                for (size_t c = 0; c < nch; ++c)
                    for (size_t b = 0; b < nb; ++b) {
                        const unsigned f = floor[c * nb + b];
                        if (f == SYMACCEL_VORBIS_FLOOR_UNUSED) continue;
                        const size_t kc = 2 * f + (flags[b] ? 1 : 0);
                        const unsigned np = g->vb_floors[f].n_posts;
                        std::memcpy(h_ys + ys_at[kc], posts + (c * nb + b) * kVorbisPosts, np * 4);
                        ys_at[kc] += np;
                        h_offs[offs_at[kc]++] = (uint32_t)(((size_t)t.first_chain - c0 + c) * cap + boff[b]);
                    }
            }
            if (ch.prepare) {
                add_pieces(w, reinterpret_cast<const char *>(h_boff + ch.boff0), reinterpret_cast<char *>(g->d_vb_boff + ch.boff0), nt * (nb + 1) * 4);
                add_pieces(w, reinterpret_cast<const char *>(h_kill + ch.kill0), reinterpret_cast<char *>(g->d_vb_kill + ch.kill0), nc * nb);
                add_pieces(w, reinterpret_cast<const char *>(h_first + ch.first0), reinterpret_cast<char *>(g->d_vb_first + ch.first0), (nt * nb + 1) * 4);
                add_pieces(w, reinterpret_cast<const char *>(h_steps + 2 * ch.steps0), reinterpret_cast<char *>(g->d_vb_steps + 2 * ch.steps0), 2 * ch.n_steps);
            }
            add_pieces(w, reinterpret_cast<const char *>(h_ys + ys_begin), reinterpret_cast<char *>(g->d_vb_ys + ys_begin), (vb_ys_at - ys_begin) * 4);
            add_pieces(w, reinterpret_cast<const char *>(h_offs + offs_begin), reinterpret_cast<char *>(g->d_vb_offs + offs_begin), (vb_offs_at - offs_begin) * 4);
            vb_first_at += nt * nb + 1;
            vb_steps_at += ch.n_steps;
        }
        const Clock::time_point api0 = Clock::now();
        // SYMACCEL_BATCH_FAKE_MIRROR (measurement only: the results are WRONG): the chunk's engine copies as ONE copy from a page-locked
        // dummy -- what a group's bulk plane would cost the link if the submissions' slots were carved from a contiguous mirror of the
        // device layout (profiles/r06y_fake_mirror.jsonl)
        static const bool fake_mirror = [] { const char *e = std::getenv("SYMACCEL_BATCH_FAKE_MIRROR"); return e && std::atoi(e) != 0; }();
        static char *fake_host = nullptr;
        constexpr size_t kFakeBytes = (size_t)160 << 20;
        if (fake_mirror && !dma.empty()) {
            static std::mutex fake_mu;
            std::lock_guard<std::mutex> fl(fake_mu);
            if (!fake_host) SYM_GPU(ctx, hipHostMalloc(reinterpret_cast<void **>(&fake_host), 2 * kFakeBytes, hipHostMallocDefault));
        }
        if (fake_mirror && !dma.empty()) {
            size_t total = 0;
            for (const Dma &m : dma) total += m.bytes;
            SYM_GPU(ctx, hipMemcpyAsync(dma[0].dst, fake_host, std::min(total, kFakeBytes), hipMemcpyHostToDevice, s_in));
            dma.clear();
        }
        for (const Dma &m : dma) SYM_GPU(ctx, hipMemcpyAsync(m.dst, m.src, m.bytes, hipMemcpyHostToDevice, s_in));
        dma.clear();
        SYM_TRY(launch_batch_copy(ctx, s_in, g0, (size_t)(w - g0), false));
        SYM_GPU(ctx, hipEventRecord(blk->ev_in[e], s_in));
        SYM_GPU(ctx, hipStreamWaitEvent(ctx->stream, blk->ev_in[e], 0));
        SYM_TRY(launch_chunk(ctx, g, c0, nc, t0, nt));
        SYM_GPU(ctx, hipEventRecord(blk->ev_k[e], ctx->stream));
        SYM_GPU(ctx, hipStreamWaitEvent(s_out, blk->ev_k[e], 0));
        *api_ns += ns_since(api0);
        // ---- scatter: PCM and the state after the batch back into the submissions' slots
        BatchCopyDesc *s0 = w;
        for (size_t ti = t0; ti < t1; ++ti) {
            const TicketView &t = views[ti];
            const SlotLayout l = slot_layout(ps, t.n_chains);
            if (g->kind == SYMACCEL_BATCH_VORBIS_SYNTH || g->kind == SYMACCEL_BATCH_VORBIS_DECODE) {
                // (the state planes of the slot still hold the state BEFORE the batch here: the scatter that overwrites them is the
                // one being built)
                for (size_t c = 0; c < t.n_chains; ++c) {
                    size_t lines, samples;
                    vorbis_used(reinterpret_cast<const uint8_t *>(t.slot + l.in[1]) + c * g->units, g->units,
                                reinterpret_cast<const int32_t *>(t.slot + l.state[0])[c], g->param & 255, (g->param >> 8) & 255, &lines, &samples);
                    bulk(g->d_out + ((size_t)t.first_chain + c) * ps.out, t.slot + l.out + c * ps.out, samples * 4);
                }
            } else if (g->row_pitch) {  // (in place: d_out is d_in[0], padded rows)
                for (size_t c = 0; c < t.n_chains; ++c) bulk(g->d_out + ((size_t)t.first_chain + c) * g->row_pitch, t.slot + l.out + c * ps.in[0], ps.in[0]);
            } else {
                bulk(g->d_out + (size_t)t.first_chain * (ps.in_place ? ps.in[0] : ps.out), t.slot + l.out, l.out_bytes);
            }
            for (int i = 0; i < ps.n_state; ++i)
                add_pieces(w, g->d_state_out[i] + (size_t)t.first_chain * ps.state[i], t.slot + l.state[i], l.state_bytes[i]);
        }
        const Clock::time_point api1 = Clock::now();
        if (fake_mirror && !dma.empty()) {
            size_t total = 0;
            for (const Dma &m : dma) total += m.bytes;
            SYM_GPU(ctx, hipMemcpyAsync(fake_host + kFakeBytes, dma[0].src, std::min(total, kFakeBytes), hipMemcpyDeviceToHost, s_out));
            dma.clear();
        }
        for (const Dma &m : dma) SYM_GPU(ctx, hipMemcpyAsync(m.dst, m.src, m.bytes, hipMemcpyDeviceToHost, s_out));
        dma.clear();
        SYM_TRY(launch_batch_copy(ctx, s_out, s0, (size_t)(w - s0), true));
        *api_ns += ns_since(api1);
        *n_chunks += 1;
        t0 = t1;
        ++k;
    }
    return SYMACCEL_OK;
}

// the device side of a closing group (mu held): a block nobody is using -- never used, given back, or whose last launch is seen
// complete now (its owner then needs it no more) --, else a new one (its memory is allocated by the launch, outside the mutex)
Block *pick_block(symaccel_batcher *b, Group *g) {
    Block *found = nullptr;
    for (auto &up : b->blocks) {
        Block *k = up.get();
        if (k->owner == nullptr) {
            found = k;
            break;
        }
        Group *o = k->owner;
        if (o->state == GroupState::Launched && (o->completed || read_flag(k->h_flag) >= k->seq)) {  // (a read of host memory: no runtime call)
            o->completed = true;
            o->block = nullptr;
            found = k;
            break;
        }
    }
    if (!found) {
        void *h = nullptr;
        if (hipHostMalloc(&h, 64, hipHostMallocDefault) != hipSuccess) {
            (void)hipGetLastError();
            return nullptr;
        }
        std::memset(h, 0, 64);
        b->blocks.emplace_back(new Block());
        found = b->blocks.back().get();
        found->h_flag = static_cast<uint64_t *>(h);
        b->stats.blocks = b->blocks.size();
    }
    found->owner = g;
    found->seq += 1;
    g->block = found;
    g->completed = false;
    g->done_flag = found->h_flag;
    g->done_seq = found->seq;
    return found;
}

// the lane a closing group goes to (mu held): round robin over the lanes that exist; the second and later ones are made on demand
Lane *pick_lane(symaccel_batcher *b) {
    if (b->lanes.empty()) {
        b->lanes.emplace_back(new Lane());
        b->lanes.back()->ctx = b->ctx;
    }
    const size_t want = std::max<size_t>(1, b->want_lanes);
    const size_t idx = b->next_lane++ % want;
    while (b->lanes.size() <= idx) {
        symaccel_ctx *c = nullptr;
        if (symaccel_ctx_create(b->ctx->device, &c) != SYMACCEL_OK) {  // (no second context: everything stays on the lanes there are)
            b->want_lanes = b->lanes.size();
            return b->lanes[idx % b->lanes.size()].get();
        }
        c->segment = b->ctx->segment;
        b->lanes.emplace_back(new Lane());
        b->lanes.back()->ctx = c;
        b->lanes.back()->owned = true;
    }
    b->stats.lanes = b->lanes.size();
    return b->lanes[idx].get();
}

// mu held.  Close the group, wait until every reservation of it is filled, enqueue it on a lane WITHOUT the mutex.  On return the
// group is Launched (its status says whether the launch worked) -- or somebody else is launching / has launched it.
void flush_group(symaccel_batcher *b, Group *g, std::unique_lock<std::mutex> &lock) {
    if (g->state != GroupState::Open) return;
    if (g->tickets == 0) {  // (opened and never filled: nothing to launch, nobody to release it)
        g->state = GroupState::Free;
        return;
    }
    g->state = GroupState::Closed;
    b->cv.wait(lock, [&] { return g->uncommitted == 0; });
    if (g->state != GroupState::Closed) return;
    int st = SYMACCEL_OK;
    uint64_t chunks = 0, host_ns = 0, lane_ns = 0, api_ns = 0, allocs = 0;
    std::string err;
    if (g->tickets) {
        // what the launch needs of the batcher, copied while the mutex is still ours
        g->views.resize(g->tickets);
        const Clock::time_point closing = Clock::now();
        for (size_t i = 0; i < g->tickets; ++i) {
            const Ticket &t = b->tickets[g->ticket_ids[i]];
            g->views[i] = TicketView{t.slot, t.first_chain, t.n_chains, SYMACCEL_OK};
            if (t.committed_at != Clock::time_point{}) b->stats.commit_to_launch_ns += (uint64_t)std::chrono::duration_cast<std::chrono::nanoseconds>(closing - t.committed_at).count();
        }
        if (g->kind == SYMACCEL_BATCH_AAC_DECODE && g->param >= 0) g->aac_maps = b->bands[(size_t)g->param].maps;  // (the index was checked by reserve())
        if (g->kind == SYMACCEL_BATCH_VORBIS_DECODE) g->vb_floors = b->floors;
        Lane *lane = pick_lane(b);
        g->lane = lane;
        if (!pick_block(b, g)) {  // (no page-locked word for a new block's completion flag)
            for (size_t i = 0; i < g->tickets; ++i) b->tickets[g->ticket_ids[i]].status = SYMACCEL_ERR_OOM;
            g->status = SYMACCEL_ERR_OOM;
            g->completed = true;
            g->state = GroupState::Launched;
            b->cv.notify_all();
            return;
        }
        g->state = GroupState::Launching;
        lock.unlock();
        {
            const Clock::time_point t0 = Clock::now();
            std::unique_lock<std::mutex> lane_lock(lane->mu);
            lane_ns = ns_since(t0);
            const Clock::time_point t1 = Clock::now();
            DeviceGuard dev(lane->ctx);
            st = dev.ok() ? launch_group_inner(lane, g, &chunks, &api_ns, &allocs) : dev.status();
            if (st != SYMACCEL_OK) {
                // a launch that failed half way: nothing of this group may still be in flight when its slots are reused, and the
                // copy-out stream does not follow what the other two were left with -- drain all three (error path only)
                err = lane->ctx->last_error;
                if (lane->ctx->stage_in) (void)hipStreamSynchronize(lane->ctx->stage_in);
                if (lane->ctx->stream) (void)hipStreamSynchronize(lane->ctx->stream);
                if (lane->ctx->stage_out) (void)hipStreamSynchronize(lane->ctx->stage_out);
            }
            // the completion flag is written behind the last scatter, which follows the last kernel, which follows the last gather
            if (st == SYMACCEL_OK) st = launch_batch_flag(lane->ctx, lane->ctx->stage_out, g->block->h_flag, g->done_seq);
            if (st != SYMACCEL_OK && err.empty()) {
                err = lane->ctx->last_error;
                if (lane->ctx->stage_out) (void)hipStreamSynchronize(lane->ctx->stage_out);
            }
            host_ns = ns_since(t1);
        }
        lock.lock();
        if (st != SYMACCEL_OK) {
            b->last_error = err;
            g->completed = true;  // (the lane's streams were drained)
        }
        for (size_t i = 0; i < g->tickets; ++i) {
            Ticket &t = b->tickets[g->ticket_ids[i]];
            t.status = st != SYMACCEL_OK ? st : g->views[i].status;
            if (t.status != SYMACCEL_OK) b->stats.failed_tickets += 1;
        }
    }
    g->status = st;
    g->state = GroupState::Launched;
    g->launched_at = Clock::now();
    b->stats.launches += 1;
    b->stats.chunks += chunks;
    b->stats.launch_host_ns += host_ns;
    b->stats.launch_api_ns += api_ns;
    b->stats.group_allocs += allocs;
    b->stats.lane_wait_ns += lane_ns;
    b->stats.chains_launched += g->chains;
    b->stats.max_chains_per_launch = std::max<uint64_t>(b->stats.max_chains_per_launch, g->chains);
    b->cv.notify_all();
}

Group *open_group(symaccel_batcher *b, int kind, int param, size_t units, const PlaneSizes &ps, size_t n_chains) {
    Group *spare = nullptr;
    for (auto &up : b->groups) {
        Group *g = up.get();
        if (g->state == GroupState::Open && g->kind == kind && g->units == units) {
            if (g->param == param) return g;
            // AAC_DECODE: a submission without jointly coded pairs (param -1) reads no band table: it rides with whatever table the
            // group has, and a group that so far holds only such submissions takes the table of the first one that does need it
            if (kind == SYMACCEL_BATCH_AAC_DECODE && (param == -1 || g->param == -1)) {
                if (g->param == -1) g->param = param;
                return g;
            }
        }
        if (g->state == GroupState::Free && !spare) spare = g;
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
    g->cap_chains = std::max<size_t>(n_chains, std::max<size_t>(2, flush_threshold(b->flush_bytes, kind) / std::max<size_t>(1, in_bytes_per_chain(ps))));
    g->chains = g->tickets = g->uncommitted = g->live = 0;
    g->ticket_ids.clear();
    g->status = SYMACCEL_OK;
    g->block = nullptr;
    g->completed = false;
    g->lane = nullptr;
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

// Wait until nothing of a launched group is in flight (no mutex held).  The launch's last kernel writes the group's sequence number
// into a word of page-locked memory: the waiter reads that word -- spinning briefly, then yielding, then sleeping in steps of 20 us
// -- and calls nothing in the runtime.  A word that never arrives (a hung or lost device) is a device error after `kFlagTimeout`.
constexpr double kFlagTimeout = 60.0;

int sync_done(symaccel_batcher *b, Group *g) {
    const uint64_t *flag;
    uint64_t seq;
    {
        Locked l(b);
        if (!g->tickets || g->completed || !g->done_flag) return SYMACCEL_OK;
        flag = g->done_flag;
        seq = g->done_seq;
    }
    const Clock::time_point t0 = Clock::now();
    for (unsigned spins = 0; read_flag(flag) < seq; ++spins) {
        if (spins < 2000) {
#if defined(__x86_64__)
            __builtin_ia32_pause();
#endif
        } else if (spins < 4000) {
            std::this_thread::yield();
        } else {
            std::this_thread::sleep_for(std::chrono::microseconds(20));
            if ((spins & 1023) == 0 && std::chrono::duration<double>(Clock::now() - t0).count() > kFlagTimeout) {
                Locked l(b);
                b->last_error = "the completion flag of a batch never arrived";
                return SYMACCEL_ERR_DEVICE;
            }
        }
    }
    Locked l(b);
    const uint64_t waited = ns_since(t0);
    b->stats.waits += 1;
    if (waited > 2000) {  // (the word was not there yet)
        b->stats.waits_blocked += 1;
        if (!g->completed) b->stats.launch_to_done_ns += ns_since(g->launched_at), b->stats.launches_timed += 1;  // (first to see it: enqueue -> completion seen)
    }
    g->completed = true;
    b->stats.flag_wait_ns += waited;
    return SYMACCEL_OK;
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
    if (const char *e = std::getenv("SYMACCEL_BATCHER_BUSY_GROUPS"))  // development knobs: the busy rule of symaccel_batcher_hint (0 = off) ...
        b->busy_groups = (size_t)std::max(0, std::atoi(e));
    if (const char *e = std::getenv("SYMACCEL_BATCHER_BUSY_HINT_MB"))  // ... and its threshold
        if (std::atoi(e) > 0) b->busy_hint_bytes = (size_t)std::atoi(e) << 20;
    if (const char *e = std::getenv("SYMACCEL_BATCHER_LANES"))  // development knob: the number of lanes (symaccel_batcher_configure)
        if (std::atoi(e) > 0) b->want_lanes = std::min(8, std::atoi(e));
    *out = b;
    return SYMACCEL_OK;
}

int symaccel_batcher_configure(symaccel_batcher *b, int lanes, size_t hint_bytes) {
    if (!b || lanes < 0 || lanes > 8) return SYMACCEL_ERR_INVALID_ARG;
    Locked l(b);
    if (lanes) b->want_lanes = (size_t)lanes;  // (lanes already made stay; fewer are used from now on)
    if (hint_bytes) b->hint_bytes = hint_bytes;
    return SYMACCEL_OK;
}

int symaccel_batcher_destroy(symaccel_batcher *b) {
    if (!b) return SYMACCEL_OK;
    {
        DeviceGuard dev(b->ctx);
        // nothing of ours may still be in flight when the staging memory goes
        for (auto &ln : b->lanes) {
            std::unique_lock<std::mutex> lane_lock(ln->mu);
            if (ln->ctx->stage_in) (void)hipStreamSynchronize(ln->ctx->stage_in);
            if (ln->ctx->stream) (void)hipStreamSynchronize(ln->ctx->stream);
            if (ln->ctx->stage_out) (void)hipStreamSynchronize(ln->ctx->stage_out);
        }
        for (auto &k : b->blocks) block_free(k.get());
        for (auto &sl : b->slabs) (void)hipHostFree(sl.base);
    }
    for (auto &ln : b->lanes)
        if (ln->owned) symaccel_ctx_destroy(ln->ctx);
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
    if (kind == SYMACCEL_BATCH_VORBIS_DECODE && n_chains != (size_t)((param >> 16) & 255)) return SYMACCEL_ERR_INVALID_ARG;  // one stream per submission
    for (int i = 0; i < ps.n_in; ++i)
        if (ps.in_div[i] == 2 && (n_chains & 1)) return SYMACCEL_ERR_INVALID_ARG;  // channel pairs
    Locked locked(b);
    std::unique_lock<std::mutex> &lock = locked.lock;
    if (kind == SYMACCEL_BATCH_AAC_DECODE && param != -1 && (param < 0 || (size_t)param >= b->bands.size())) return SYMACCEL_ERR_INVALID_ARG;  // (symaccel_batcher_aac_bands first; -1: no pairs, no table)
    // the slot first: page-locking a new slab drops the mutex, and the group must be chosen in one piece with the reservation
    const SlotLayout lay = slot_layout(ps, n_chains);
    const size_t cls = slot_class(lay.bytes);
    char *mem = nullptr;
    SYM_TRY(slot_alloc(b, lock, cls, &mem));
    Group *g = open_group(b, kind, param, units_per_chain, ps, n_chains);
    while (g->chains + n_chains > g->cap_chains && g->tickets) {  // full: it goes, a fresh one opens (or one somebody else opened meanwhile)
        flush_group(b, g, lock);
        g = open_group(b, kind, param, units_per_chain, ps, n_chains);
    }
    uint32_t idx;
    if (!b->free_tickets.empty()) {
        idx = b->free_tickets.back();
        b->free_tickets.pop_back();
    } else {
        idx = (uint32_t)b->tickets.size();
        b->tickets.emplace_back();
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
    t->slot_bytes = cls;
    g->ticket_ids.push_back(idx);
    g->chains += n_chains;
    g->tickets += 1;
    g->uncommitted += 1;
    g->live += 1;
    b->stats.submissions += 1;
    b->slots_live += 1;
    b->stats.slots_peak = std::max<uint64_t>(b->stats.slots_peak, b->slots_live);
    fill_slot(g, t, slot);
    *ticket = ((uint64_t)gen << 32) | idx;
    return SYMACCEL_OK;
}

int symaccel_batcher_commit(symaccel_batcher *b, uint64_t ticket) {
    if (!b) return SYMACCEL_ERR_INVALID_ARG;
    Locked locked(b);
    Ticket *t = find_ticket(b, ticket);
    if (!t || t->committed) return SYMACCEL_ERR_INVALID_ARG;
    t->committed = true;
    t->committed_at = Clock::now();
    Group *g = t->group;
    g->uncommitted -= 1;
    if (g->uncommitted == 0) b->cv.notify_all();
    // enough input has piled up: to the device, nobody has to ask
    if (g->state == GroupState::Open && g->chains * in_bytes_per_chain(g->ps) >= flush_threshold(b->flush_bytes, g->kind)) flush_group(b, g, locked.lock);
    return SYMACCEL_OK;
}

int symaccel_batcher_flush(symaccel_batcher *b) {
    if (!b) return SYMACCEL_ERR_INVALID_ARG;
    Locked locked(b);
    for (size_t i = 0; i < b->groups.size(); ++i)  // (index loop: flush_group drops the lock while it waits for commits and while it enqueues)
        if (b->groups[i]->state == GroupState::Open && b->groups[i]->tickets) flush_group(b, b->groups[i].get(), locked.lock);
    return SYMACCEL_OK;
}

int symaccel_batcher_hint(symaccel_batcher *b) {
    if (!b) return SYMACCEL_ERR_INVALID_ARG;
    Locked locked(b);
    // "results will be wanted soon": whatever is worth a launch of its own goes now, so that the copies and the kernels run while
    // the callers are still busy with their current batches; a group below that size waits for more submissions (or for a waiter).
    // What is worth a launch depends on the device: while it is idle a small group is (its latency is hidden behind the callers' work);
    // while it has a queue -- `busy_groups` launches whose completion word is outstanding (a read of page-locked memory) -- an early launch
    // buys nothing, and LARGER groups move faster: the copy kernels of a 6 MiB group carry 28 GB/s each way, those of a 40 MiB group 31
    // (AAC at S = 256: 3.45 -> 3.92 M packets/s, S = 64: 3.38 -> 3.62, S = 4: 1.78 -> 1.98; profiles/r06z1_hint.jsonl, r06z3_busy.jsonl)
    size_t in_flight = 0;
    for (auto &up : b->groups) {
        const Group *o = up.get();
        if (o->state == GroupState::Closed || o->state == GroupState::Launching) in_flight += 1;
        else if (o->state == GroupState::Launched && o->tickets && !o->completed && o->done_flag && read_flag(o->done_flag) < o->done_seq) in_flight += 1;
    }
    const bool busy = b->busy_groups && in_flight >= b->busy_groups;
    for (size_t i = 0; i < b->groups.size(); ++i) {
        Group *g = b->groups[i].get();
        size_t worth = hint_threshold(b->hint_bytes, b->flush_bytes, g->kind);
        if (busy) worth = std::max(worth, std::min(b->busy_hint_bytes, b->flush_bytes));
        if (g->state == GroupState::Open && g->tickets && g->chains * in_bytes_per_chain(g->ps) >= worth) flush_group(b, g, locked.lock);
    }
    return SYMACCEL_OK;
}

int symaccel_batcher_wait(symaccel_batcher *b, uint64_t ticket, symaccel_batch_slot *slot) {
    if (!b) return SYMACCEL_ERR_INVALID_ARG;
    Group *g;
    int status;
    {
        Locked locked(b);
        std::unique_lock<std::mutex> &lock = locked.lock;
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
        t = find_ticket(b, ticket);  // (the table may have grown while the lock was dropped)
        if (!t) return SYMACCEL_ERR_INVALID_ARG;
        if (slot) fill_slot(g, t, slot);
        status = t->status;
    }
    const int st = sync_done(b, g);
    return st != SYMACCEL_OK ? st : status;
}

int symaccel_batcher_release(symaccel_batcher *b, uint64_t ticket) {
    if (!b) return SYMACCEL_ERR_INVALID_ARG;
    Locked locked(b);
    std::unique_lock<std::mutex> &lock = locked.lock;
    Ticket *t = find_ticket(b, ticket);
    if (!t) return SYMACCEL_ERR_INVALID_ARG;
    Group *g = t->group;
    if (!t->committed) {
        // abandoned before it was filled (a front end that threw half way): what the slot holds is whatever it is, and it will be
        // launched with its group -- as an EMPTY description: every plane zeroed (no pairs, no filters, verbatim blocks, silence), so
        // that the neighbours' launch never reads garbage descriptors.  Error path: the memset runs without the mutex (the group
        // cannot be launched while this reservation is open).
        char *mem = t->slot;
        const size_t bytes = slot_layout(g->ps, t->n_chains).bytes;
        lock.unlock();
        std::memset(mem, 0, bytes);
        lock.lock();
        t = find_ticket(b, ticket);
        if (!t) return SYMACCEL_ERR_INVALID_ARG;
        if (!t->committed) {
            t->committed = true;
            g->uncommitted -= 1;
            if (g->uncommitted == 0) b->cv.notify_all();
        }
    }
    // The slot goes back to the pool -- but the group's copies may still be reading or writing it (a release without a wait, or
    // before the launch): the group is launched if it has not been, and drained, first.  (The common order -- wait, read, release --
    // finds the event signalled.)
    if (g->state == GroupState::Open) flush_group(b, g, lock);
    b->cv.wait(lock, [&] { return g->state == GroupState::Launched; });
    if (g->tickets && !g->completed) {
        lock.unlock();
        (void)sync_done(b, g);
        lock.lock();
    }
    t = find_ticket(b, ticket);  // (the table may have grown while the lock was dropped)
    if (!t) return SYMACCEL_ERR_INVALID_ARG;
    slot_free(b, t->slot, t->slot_bytes);
    b->slots_live -= 1;
    t->slot = nullptr;
    t->live = false;
    b->free_tickets.push_back((uint32_t)(ticket & 0xffffffffu));
    g->live -= 1;
    if (g->live == 0 && g->state == GroupState::Launched) {
        // (every submission was drained before it was released: the launch is complete, its block serves the next one)
        if (g->block && g->block->owner == g) g->block->owner = nullptr;
        g->block = nullptr;
        g->state = GroupState::Free;
    }
    return SYMACCEL_OK;
}

int symaccel_batcher_plane_bytes(int kind, int param, size_t units_per_chain, size_t *in_bytes, size_t *state_bytes, size_t *out_bytes) {
    PlaneSizes ps;
    if (!plane_sizes(kind, param, units_per_chain, &ps)) return SYMACCEL_ERR_INVALID_ARG;
    for (int i = 0; i < kMaxIn; ++i)
        if (in_bytes) in_bytes[i] = ps.in[i];
    for (int i = 0; i < kMaxState; ++i)
        if (state_bytes) state_bytes[i] = ps.state[i];
    if (out_bytes) *out_bytes = ps.in_place ? ps.in[0] : ps.out;
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
        Locked locked(b);
        Ticket *t = find_ticket(b, id);
        for (int i = 0; i < ps.n_state; ++i) t->user_state[i] = state_io[i];
        t->user_out = out;
    }
    *ticket = id;
    return symaccel_batcher_commit(b, id);
}

int symaccel_batcher_submit_aac_synth(symaccel_batcher *b, const float *coeffs, const uint8_t *side, float *delay_io, float *pcm, size_t n_chains,
                                      size_t frames_per_chain, uint64_t *ticket) {
    const void *in[kMaxIn] = {coeffs, side, nullptr, nullptr, nullptr, nullptr};
    void *st[3] = {delay_io, nullptr, nullptr};
    return symaccel_batcher_submit(b, SYMACCEL_BATCH_AAC_SYNTH, 0, n_chains, frames_per_chain, in, st, pcm, ticket);
}

int symaccel_batcher_submit_mp3_synth(symaccel_batcher *b, const float *xr, const symaccel_mp3_side *side, int sample_rate_idx, float *overlap_io,
                                      float *vvec_io, int32_t *vfront_io, float *pcm, size_t n_chains, size_t granules_per_chain, uint64_t *ticket) {
    const void *in[kMaxIn] = {xr, side, nullptr, nullptr, nullptr, nullptr};
    void *st[3] = {overlap_io, vvec_io, vfront_io};
    return symaccel_batcher_submit(b, SYMACCEL_BATCH_MP3_SYNTH, sample_rate_idx, n_chains, granules_per_chain, in, st, pcm, ticket);
}

int symaccel_batcher_submit_mp3_decode(symaccel_batcher *b, const int16_t *quant, const symaccel_mp3_requant *rq_desc, const symaccel_mp3_stereo *st_desc,
                                       const symaccel_mp3_side *side, int sample_rate_idx, float *overlap_io, float *vvec_io, int32_t *vfront_io,
                                       float *pcm, size_t n_chains, size_t granules_per_chain, uint64_t *ticket) {
    const void *in[kMaxIn] = {quant, rq_desc, side, st_desc, nullptr, nullptr};
    void *st[3] = {overlap_io, vvec_io, vfront_io};
    return symaccel_batcher_submit(b, SYMACCEL_BATCH_MP3_DECODE, sample_rate_idx, n_chains, granules_per_chain, in, st, pcm, ticket);
}

int symaccel_batcher_aac_bands(symaccel_batcher *b, const uint16_t *swb_long, int n_swb_long, const uint16_t *swb_short, int n_swb_short, int *bands) {
    if (!b || !bands || !swb_long || !swb_short || n_swb_long < 1 || n_swb_short < 1 || n_swb_long > 63 || n_swb_short > 15) return SYMACCEL_ERR_INVALID_ARG;
    symaccel_batcher::Bands nb;
    if (!aac_band_maps(swb_long, n_swb_long, swb_short, n_swb_short, &nb.maps)) return SYMACCEL_ERR_INVALID_ARG;
    nb.swb_long.assign(swb_long, swb_long + n_swb_long + 1);
    nb.swb_short.assign(swb_short, swb_short + n_swb_short + 1);
    Locked locked(b);
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
    {
        Locked locked(b);
        if (bands != -1 && (bands < 0 || (size_t)bands >= b->bands.size())) return SYMACCEL_ERR_INVALID_ARG;  // (tables that were never registered)
    }
    // (a batch without a jointly coded pair reads no band table: it shares a launch with streams of any table)
    SYM_TRY(symaccel_batcher_reserve(b, SYMACCEL_BATCH_AAC_DECODE, n_pairs ? bands : -1, n_chains, frames_per_chain, &slot, &id));
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
        Locked locked(b);
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
    const void *in[kMaxIn] = {spectra, block_flag, nullptr, nullptr, nullptr, nullptr};
    void *st[3] = {prev_flag_io, overlap_io, nullptr};
    return symaccel_batcher_submit(b, SYMACCEL_BATCH_VORBIS_SYNTH, bs0_exp | (bs1_exp << 8), n_chains, blocks_per_chain, in, st, pcm, ticket);
}

int symaccel_batcher_vorbis_floor(symaccel_batcher *b, const symaccel_vorbis_floor1_cfg *cfg, int *index) {
    if (!b || !cfg || !index) return SYMACCEL_ERR_INVALID_ARG;
    if (cfg->n_posts < 2 || cfg->n_posts > kVorbisPosts || cfg->multiplier < 1 || cfg->multiplier > 4) return SYMACCEL_ERR_INVALID_ARG;
    symaccel_vorbis_floor1_cfg c{};
    c.multiplier = cfg->multiplier;
    c.n_posts = cfg->n_posts;
    for (unsigned i = 0; i < cfg->n_posts; ++i) {
        if (cfg->x_list[i] > 0xffffu) return SYMACCEL_ERR_INVALID_ARG;  // floor1_X values have at most 15 bits (rangebits)
        for (unsigned j = 0; j < i; ++j)
            if (cfg->x_list[j] == cfg->x_list[i]) return SYMACCEL_ERR_INVALID_ARG;  // render_line divides by (x1 - x0)
        c.x_list[i] = cfg->x_list[i];
    }
    Locked locked(b);
    for (size_t i = 0; i < b->floors.size(); ++i)
        if (std::memcmp(&b->floors[i], &c, sizeof c) == 0) {
            *index = (int)i;
            return SYMACCEL_OK;
        }
    if (b->floors.size() >= SYMACCEL_VORBIS_FLOOR_UNUSED) return SYMACCEL_ERR_UNSUPPORTED;
    b->floors.push_back(c);
    *index = (int)b->floors.size() - 1;
    return SYMACCEL_OK;
}

int symaccel_batcher_submit_vorbis_decode(symaccel_batcher *b, int bs0_exp, int bs1_exp, const float *residue, const uint8_t *block_flag,
                                          const uint8_t *floor, const uint32_t *posts, const uint8_t *coupling, const uint32_t *coupling_first,
                                          int32_t *prev_flag_io, float *overlap_io, float *pcm, size_t n_chains, size_t blocks_per_chain,
                                          uint64_t *ticket) {
    if (!b || !residue || !block_flag || !floor || !posts || !coupling_first || !prev_flag_io || !overlap_io || !pcm || !ticket) return SYMACCEL_ERR_INVALID_ARG;
    if (bs0_exp < 0 || bs0_exp > 255 || bs1_exp < 0 || bs1_exp > 255 || n_chains == 0 || n_chains > 255 || blocks_per_chain == 0) return SYMACCEL_ERR_INVALID_ARG;
    const int param = bs0_exp | (bs1_exp << 8) | ((int)n_chains << 16);
    PlaneSizes ps;
    if (!plane_sizes(SYMACCEL_BATCH_VORBIS_DECODE, param, blocks_per_chain, &ps)) return SYMACCEL_ERR_INVALID_ARG;
    const size_t n_steps = coupling_first[blocks_per_chain];
    if ((n_steps && !coupling) || vorbis_blob_steps(blocks_per_chain) + 2 * n_steps > ps.in[4]) return SYMACCEL_ERR_INVALID_ARG;
    symaccel_batch_slot slot;
    uint64_t id = 0;
    SYM_TRY(symaccel_batcher_reserve(b, SYMACCEL_BATCH_VORBIS_DECODE, param, n_chains, blocks_per_chain, &slot, &id));
    std::memcpy(slot.input[0], residue, slot.input_bytes[0]);
    std::memcpy(slot.input[1], block_flag, slot.input_bytes[1]);
    std::memcpy(slot.input[2], floor, slot.input_bytes[2]);
    std::memcpy(slot.input[3], posts, slot.input_bytes[3]);
    char *blob = static_cast<char *>(slot.input[4]);
    std::memcpy(blob, coupling_first, (blocks_per_chain + 1) * 4);
    if (n_steps) std::memcpy(blob + vorbis_blob_steps(blocks_per_chain), coupling, 2 * n_steps);
    std::memcpy(slot.state[0], prev_flag_io, slot.state_bytes[0]);
    std::memcpy(slot.state[1], overlap_io, slot.state_bytes[1]);
    {
        Locked locked(b);
        Ticket *t = find_ticket(b, id);
        t->user_state[0] = prev_flag_io;
        t->user_state[1] = overlap_io;
        t->user_out = pcm;
    }
    *ticket = id;
    return symaccel_batcher_commit(b, id);
}

int symaccel_batcher_submit_flac_restore(symaccel_batcher *b, int32_t *buf_io, const symaccel_flac_desc *desc, const int32_t *coeffs,
                                         const uint8_t *pair_mode, uint32_t out_shift, size_t n_blocks, size_t blocksize, uint64_t *ticket) {
    if (out_shift > 31 || (!pair_mode && out_shift)) return SYMACCEL_ERR_INVALID_ARG;
    const void *in[kMaxIn] = {buf_io, desc, coeffs, pair_mode, nullptr, nullptr};
    void *st[3] = {nullptr, nullptr, nullptr};
    return symaccel_batcher_submit(b, SYMACCEL_BATCH_FLAC_RESTORE, pair_mode ? (int)(0x100 | out_shift) : 0, n_blocks, blocksize, in, st, buf_io, ticket);
}

int symaccel_batcher_submit_alac_predict(symaccel_batcher *b, int32_t *buf_io, const symaccel_alac_desc *desc, const int32_t *coeffs,
                                         const int32_t *pair_weight, const uint8_t *pair_shift, size_t n_blocks, size_t blocksize, uint64_t *ticket) {
    if ((pair_weight == nullptr) != (pair_shift == nullptr)) return SYMACCEL_ERR_INVALID_ARG;
    const void *in[kMaxIn] = {buf_io, desc, coeffs, pair_weight, pair_shift, nullptr};
    void *st[3] = {nullptr, nullptr, nullptr};
    return symaccel_batcher_submit(b, SYMACCEL_BATCH_ALAC_PREDICT, pair_weight ? 0x100 : 0, n_blocks, blocksize, in, st, buf_io, ticket);
}

int symaccel_batcher_collect(symaccel_batcher *b, uint64_t ticket) {
    if (!b) return SYMACCEL_ERR_INVALID_ARG;
    symaccel_batch_slot slot;
    void *user_state[kMaxState], *user_out;
    {
        Locked locked(b);
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
    Locked locked(b);
    *out = b->stats;
    uint64_t pending = 0;
    for (auto &g : b->groups)
        if (g->state == GroupState::Open || g->state == GroupState::Closed) pending += g->tickets;
    out->pending = pending;
    return SYMACCEL_OK;
}

int symaccel_batcher_last_error(symaccel_batcher *b, char *buf, size_t capacity) {
    if (!b || !buf || capacity == 0) return SYMACCEL_ERR_INVALID_ARG;
    Locked locked(b);
    const size_t n = std::min(capacity - 1, b->last_error.size());
    std::memcpy(buf, b->last_error.data(), n);
    buf[n] = 0;
    return SYMACCEL_OK;
}

}  // extern "C"
