// Trait-level throughput: S streams decoded packet by packet through codecs::LookaheadDecoder (include/symaccel.hpp, the compiled
// twin of the Rust shim's decoders), host memory in -> host memory out, the way a transcoding server drives
// AudioDecoder::decode_ref (symphonia-core/src/codecs/audio.rs:279-297): T worker threads, every thread owns S / T streams and
// calls decode() on them round-robin.  Two modes:
//   batcher     every decoder submits to ONE process-wide symaccel_batcher (csrc/batcher.cpp): one launch per group of streams
//   per-stream  every decoder batches its own look-ahead (one small call per stream and batch; the context is shared, so the
//               calls are serialised by a mutex -- the context is externally synchronised)
// The packets are pre-parsed synthetic spectra (what the CPU front end hands over); the "demuxer + parser" of a stream is a
// peek that copies the next packet out of a small pool (a parser writes its output somewhere too).  Nothing here links the
// oracle: bench.py times the CPU port beside this with its own harness.  One JSON line on stdout.
//
//   decoders_bench --codec aac|aacd|mp3|mp3h|vorbis|flac --streams S --lookahead L --packets P --threads T [--per-stream] [--direct] [--in-phase] [--flush-mb M] [--lanes N] [--via-registry]
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <mutex>
#include <random>
#include <string>
#include <thread>
#include <vector>

#include "symaccel.hpp"

using namespace symphonia_accel;
using namespace symphonia_accel::codecs;
using Clock = std::chrono::steady_clock;

namespace {

struct Args {
    std::string codec = "aac";
    size_t streams = 64, lookahead = 64, packets = 512, threads = 1, flush_mb = 0, warm = 0, lanes = 0;
    bool per_stream = false, direct = false, in_phase = false, via_registry = false;
};

constexpr size_t kPool = 32;

template <class Codec>
std::vector<typename Codec::Packet> make_pool(unsigned seed);

template <>
std::vector<AacLc::Packet> make_pool<AacLc>(unsigned seed) {
    std::mt19937 rng(seed);
    std::normal_distribution<float> nd(0.0f, 50.0f);
    std::vector<AacLc::Packet> pool(kPool);
    for (auto &p : pool) {
        p.coeffs.resize(2 * 1024);
        for (size_t i = 0; i < 2048; ++i) p.coeffs[i] = (i % 1024) < 672 ? nd(rng) : 0.0f;
        p.side.assign(2, SYMACCEL_AAC_SIDE(0u, 1u, 1u));  // ONLY_LONG, KBD
    }
    return pool;
}

template <>
std::vector<Mp3::Packet> make_pool<Mp3>(unsigned seed) {
    std::mt19937 rng(seed);
    std::normal_distribution<float> nd(0.0f, 0.1f);
    std::vector<Mp3::Packet> pool(kPool);
    for (auto &p : pool) {
        p.xr.resize(2 * 2 * 576);
        for (auto &v : p.xr) v = nd(rng);
        p.side.assign(4, symaccel_mp3_side{0, 0, 576});
    }
    return pool;
}

template <>
std::vector<Mp3Huffman::Packet> make_pool<Mp3Huffman>(unsigned seed) {
    std::mt19937 rng(seed);
    std::vector<Mp3Huffman::Packet> pool(kPool);
    for (auto &p : pool) {
        p.quant.resize(2 * 2 * 576);
        for (auto &v : p.quant) v = (int16_t)((int)(rng() % 81) - 40);
        p.rq.resize(4);
        for (auto &r : p.rq) {
            std::memset(&r, 0, sizeof r);
            r.global_gain = 150;
            r.rzero = 576;
            for (int s = 0; s < 39; ++s) r.scalefacs[s] = (uint8_t)(rng() % 4);
        }
        p.stereo.resize(2);
        for (auto &s : p.stereo) {
            std::memset(&s, 0, sizeof s);
            s.flags = SYMACCEL_MP3_ST_MID_SIDE | SYMACCEL_MP3_ST_MPEG1;
            s.rzero0 = s.rzero1 = 576;
        }
        p.side.assign(4, symaccel_mp3_side{0, 0, 576});
    }
    return pool;
}

template <>
std::vector<Vorbis::Packet> make_pool<Vorbis>(unsigned seed) {  // BASELINE config 4's shape: 8 channels, 2048 / 256 mixed, mostly long blocks
    std::mt19937 rng(seed);
    std::normal_distribution<float> nd(0.0f, 0.25f);
    std::vector<Vorbis::Packet> pool(kPool);
    for (auto &p : pool) {
        p.long_block = rng() % 10 != 0;
        p.spectra.resize(8 * (p.long_block ? 1024 : 128));
        for (auto &v : p.spectra) v = nd(rng);
    }
    return pool;
}

template <>
std::vector<Flac::Packet> make_pool<Flac>(unsigned seed) {  // BASELINE config 5's shape per packet: two subframes of 4096 samples, LPC order 32, 24 bits
    std::mt19937 rng(seed);
    std::vector<Flac::Packet> pool(kPool);
    for (auto &p : pool) {
        p.blocksize = 4096;
        p.words.resize(2 * 4096);
        for (auto &v : p.words) v = (int32_t)(rng() % 2001) - 1000;
        p.desc.assign(2, symaccel_flac_desc{SYMACCEL_FLAC_LPC, 32, 12, 0});
        p.coeffs.resize(2 * 32);
        for (auto &c : p.coeffs) c = (int32_t)(rng() % 401) - 200;
        p.pair_mode = (uint8_t)(rng() % 4);
    }
    return pool;
}

static const uint16_t kSwbLong[] = {0, 4, 8, 12, 16, 20, 24, 28, 32, 36, 40, 48, 56, 64, 72, 80, 88, 96, 108, 120, 132, 144, 160, 176, 196, 216,
                                    240, 264, 292, 320, 352, 384, 416, 448, 480, 512, 544, 576, 608, 640, 672, 704, 736, 768, 800, 832, 864, 896,
                                    928, 1024};
static const uint16_t kSwbShort[] = {0, 4, 8, 12, 16, 20, 28, 36, 44, 56, 68, 80, 96, 112, 128};

template <>
std::vector<AacLcCoded::Packet> make_pool<AacLcCoded>(unsigned seed) {  // long blocks, the pair mid/side- or intensity-coded, TNS on 30 % of the frames
    std::mt19937 rng(seed);
    std::normal_distribution<float> nd(0.0f, 50.0f);
    std::vector<AacLcCoded::Packet> pool(kPool);
    for (auto &p : pool) {
        p.coeffs.resize(2 * 1024);
        for (size_t i = 0; i < 2048; ++i) p.coeffs[i] = (i % 1024) < 672 ? nd(rng) : 0.0f;
        p.side.assign(2, SYMACCEL_AAC_SIDE(0u, 1u, 1u));
        symaccel_aac_js_frame d;
        std::memset(&d, 0, sizeof d);
        d.num_windows = 1;
        d.max_sfb = 40;
        for (int s = 0; s < 128; ++s) {
            const unsigned r = rng() % 10;
            d.mode[s] = (uint8_t)(r < 6 ? SYMACCEL_AAC_JS_MS : (r < 8 ? SYMACCEL_AAC_JS_INTENSITY : 0));
            d.scale[s] = 0.5f;
        }
        p.joint.emplace_back(0, d);
        for (uint32_t c = 0; c < 2; ++c)
            if (rng() % 10 < 3) {
                symaccel_aac_tns_filter f;
                std::memset(&f, 0, sizeof f);
                f.frame = c;
                f.start = 160;
                f.end = 672;
                f.order = 10;
                for (int q = 0; q < 10; ++q) f.lpc[q] = 0.05f * (float)((int)(rng() % 9) - 4);
                p.tns.push_back(f);
            }
    }
    return pool;
}

// --direct: the "parser" writes the next packet straight into the batcher's slot (LookaheadDecoder::Direct) -- one copy of the
// parsed packet (pool -> slot) instead of two (pool -> Packet -> slot)
template <class Codec>
void parse_into(const typename Codec::BatchView &, std::size_t, const typename Codec::Packet &) {}
template <>
void parse_into<AacLc>(const AacLc::BatchView &v, std::size_t i, const AacLc::Packet &p) {
    for (size_t c = 0; c < v.nch; ++c) {
        std::memcpy(v.coeffs_at(c, i), p.coeffs.data() + c * 1024, 4096);
        v.side_at(c, i) = p.side[c];
    }
}
template <>
void parse_into<Mp3>(const Mp3::BatchView &v, std::size_t i, const Mp3::Packet &p) {
    for (size_t gr = 0; gr < v.ngr; ++gr)
        for (size_t c = 0; c < v.nch; ++c) {
            std::memcpy(v.xr + v.unit(c, i, gr) * 576, p.xr.data() + (gr * v.nch + c) * 576, 2304);
            v.side[v.unit(c, i, gr)] = p.side[gr * v.nch + c];
        }
}
template <>
void parse_into<Mp3Huffman>(const Mp3Huffman::BatchView &v, std::size_t i, const Mp3Huffman::Packet &p) {
    for (size_t gr = 0; gr < v.ngr; ++gr) {
        for (size_t c = 0; c < v.nch; ++c) {
            const size_t dst = v.unit(c, i, gr);
            std::memcpy(v.quant + dst * 576, p.quant.data() + (gr * v.nch + c) * 576, 1152);
            v.rq[dst] = p.rq[gr * v.nch + c];
            v.side[dst] = p.side[gr * v.nch + c];
        }
        v.stereo[i * v.ngr + gr] = p.stereo[gr];
    }
}

template <class Codec>
typename Codec::Params params();
template <>
AacLcCoded::Params params<AacLcCoded>() {
    AacLcCoded::Params p;
    p.channels = 2;
    p.swb_long.assign(kSwbLong, kSwbLong + sizeof kSwbLong / sizeof kSwbLong[0]);
    p.swb_short.assign(kSwbShort, kSwbShort + sizeof kSwbShort / sizeof kSwbShort[0]);
    return p;
}
template <>
AacLc::Params params<AacLc>() { return AacLc::Params{2}; }
template <>
Mp3::Params params<Mp3>() { return Mp3::Params{2, 2, 0}; }
template <>
Mp3Huffman::Params params<Mp3Huffman>() { return Mp3Huffman::Params{2, 2, 0}; }
template <>
Vorbis::Params params<Vorbis>() { return Vorbis::Params{8, 8, 11}; }
template <>
Flac::Params params<Flac>() { return Flac::Params{2, 24, 4096}; }

template <class Codec>
int run(const Args &a, const char *codec_name, size_t frames_per_packet, size_t bytes_in_per_packet) {
    using Packet = typename Codec::Packet;
    using Decoder = LookaheadDecoder<Codec>;
    // --via-registry: the decoders come from CodecRegistry::make_audio_decoder, which is handed (params, options) and the packet source
    // and nothing else (registry.rs:330-341) -- no Batcher is passed anywhere: the factory finds the process-wide one, as the Rust shim's
    // try_registry_new does.  The harness looks at Batcher::shared() itself only to read the statistics (and for --lanes / --flush-mb).
    std::unique_ptr<Context> own_ctx;
    std::unique_ptr<Batcher> own_batcher;
    Batcher *batcher = nullptr;
    CodecRegistry registry;
    if (a.via_registry) {
        if (a.per_stream) throw std::invalid_argument("--via-registry builds pooled decoders: not with --per-stream");
        register_enabled_codecs(registry);
        if (a.flush_mb) throw std::invalid_argument("--via-registry: the shared batcher has the library's flush size");
        batcher = &Batcher::shared();
    } else {
        own_ctx.reset(new Context(0));
        if (!a.per_stream) {
            own_batcher.reset(new Batcher(*own_ctx, a.flush_mb << 20));
            batcher = own_batcher.get();
        }
    }
    Context &ctx = a.via_registry ? batcher->context() : *own_ctx;
    if (batcher && a.lanes) check(symaccel_batcher_configure(batcher->raw(), (int)a.lanes, 0), ctx.raw());
    const std::vector<Packet> pool = make_pool<Codec>(17);
    struct Stream {
        size_t cursor = 0, limit = 0, salt = 0, pos = 0, lead = 0;
        std::unique_ptr<Decoder> dec;
    };
    std::vector<std::unique_ptr<Stream>> streams(a.streams);
    const size_t total = a.warm + a.packets;
    for (size_t s = 0; s < a.streams; ++s) {
        streams[s].reset(new Stream());
        Stream *st = streams[s].get();
        st->salt = s * 7;
        // streams do not start together in a service: stream s of a thread's S / T streams is (s / T) * L / (S / T) packets ahead of the
        // first, so that the batches of a thread's streams end at different times (--in-phase: every stream at the same packet)
        const size_t per_thread = std::max<size_t>(1, (a.streams + std::min(a.threads, a.streams) - 1) / std::min(a.threads, a.streams));
        st->lead = a.in_phase ? 0 : (s / std::min(a.threads, a.streams)) * a.lookahead / per_thread;
        st->limit = total + st->lead + 2 * a.lookahead;  // (no stream ends inside the timed region: a tail batch is a shape of its own)
        auto peek = [st, &pool]() -> std::optional<Packet> {
            if (st->cursor >= st->limit) return std::nullopt;
            Packet p = pool[(st->cursor + st->salt) % kPool];  // (the copy a parser's output costs)
            p.ts = st->cursor++;
            return p;
        };
        if (batcher && a.direct) {
            if constexpr (Codec::kDirect) {
                typename Decoder::Direct d;
                d.avail = [st]() { return st->limit - std::min(st->limit, st->cursor); };
                d.parse_into = [st, &pool](const typename Codec::BatchView &v, size_t i) -> std::optional<uint64_t> {
                    if (st->cursor >= st->limit) return std::nullopt;
                    parse_into<Codec>(v, i, pool[(st->cursor + st->salt) % kPool]);
                    return st->cursor++;
                };
                if (a.via_registry)
                    st->dec = registry.make_audio_decoder<Codec>(params<Codec>(), AudioDecoderOptions{false, a.lookahead}, d);
                else
                    st->dec.reset(new Decoder(*batcher, params<Codec>(), a.lookahead, d));
            } else {
                throw std::invalid_argument("--direct: this codec's layout depends on the packets (use the peek form)");
            }
        } else if (a.via_registry)
            st->dec = registry.make_audio_decoder<Codec>(params<Codec>(), AudioDecoderOptions{false, a.lookahead}, peek);
        else if (batcher)
            st->dec.reset(new Decoder(*batcher, params<Codec>(), a.lookahead, peek));
        else
            st->dec.reset(new Decoder(ctx, params<Codec>(), a.lookahead, peek));
    }
    std::mutex ctx_mu;  // per-stream mode: one context, externally synchronised
    std::atomic<size_t> failures{0};
    std::atomic<uint64_t> checksum{0};
    // What decode() is handed: in the warm-up the parsed packet in full (a cold decoder transforms it at once); afterwards only its
    // identity -- the trait's packet carries the COMPRESSED bytes, and a packet the look-ahead has already parsed is not parsed
    // again (a decoder that did need the content here fails with "packet shape" and is counted in `failures`).
    auto phase = [&](size_t first, size_t count, bool full, bool lead) {
        std::vector<std::thread> ths;
        const size_t T = std::min(a.threads, a.streams);
        for (size_t t = 0; t < T; ++t)
            ths.emplace_back([&, t]() {
                uint64_t acc = 0;
                try {
                    for (size_t step = first; step < first + count + (lead ? a.lookahead : 0); ++step)
                        for (size_t s = t; s < a.streams; s += T) {
                            Stream &st = *streams[s];
                            if (lead && step >= first + count + st.lead) continue;  // (the warm-up runs stream s `lead` packets further)
                            const size_t i = st.pos++;
                            Packet p;
                            if (full) p = pool[(i + st.salt) % kPool];
                            p.ts = i;
                            if (st.cursor <= i) st.cursor = i + 1;
                            if (batcher) {
                                const auto &buf = st.dec->decode(p);
                                uint32_t w;
                                if (buf.frames) std::memcpy(&w, buf.planes[0] + (i % 1024 % buf.frames), 4);  // (touch the result)
                                else w = 0;
                                acc += w;
                            } else {
                                std::lock_guard<std::mutex> lock(ctx_mu);
                                const auto &buf = st.dec->decode(p);
                                uint32_t w = 0;
                                if (buf.frames) std::memcpy(&w, buf.planes[0] + (i % 1024 % buf.frames), 4);
                                acc += w;
                            }
                        }
                } catch (const std::exception &e) {
                    std::fprintf(stderr, "decode failed: %s\n", e.what());
                    failures += 1;
                }
                checksum += acc;
            });
        for (auto &th : ths) th.join();
    };
    if (a.warm) phase(0, a.warm, true, true);
    symaccel_batcher_stats s0{};
    if (batcher) s0 = batcher->stats();
    const auto t0 = Clock::now();
    phase(a.warm, a.packets, !batcher, false);  // (a per-stream decoder transforms the packet it is handed when its batch is used up)
    const double secs = std::chrono::duration<double>(Clock::now() - t0).count();
    symaccel_batcher_stats s1{};
    if (batcher) s1 = batcher->stats();
    size_t batches = 0;
    for (auto &st : streams) batches += st->dec->batches_run();
    streams.clear();
    const double n = (double)a.streams * (double)a.packets;
    std::printf("{\"codec\": \"%s\", \"mode\": \"%s\", \"direct\": %s, \"in_phase\": %s, \"streams\": %zu, \"lookahead\": %zu, \"threads\": %zu, \"packets\": %.0f, \"seconds\": %.6f, "
                "\"packets_per_s\": %.1f, \"frames_per_packet\": %zu, \"host_bytes_in_per_packet\": %zu, \"host_bytes_out_per_packet\": %zu, "
                "\"GBps_each_way\": [%.3f, %.3f], \"decoder_batches\": %zu, \"launches\": %llu, \"kernel_launches\": %llu, \"max_chains_per_launch\": %llu, "
                "\"staging_bytes\": %llu, \"staging_grew_bytes\": %llu, \"slots_peak\": [%llu, %llu], \"lanes\": %llu, \"mutex_wait_ms\": %.3f, \"mutex_contended\": %llu, \"launch_host_ms\": %.3f, \"launch_api_ms\": %.3f, \"lane_wait_ms\": %.3f, \"group_allocs\": %llu, \"blocks\": %llu, \"flag_wait_ms\": %.3f, \"commit_to_launch_ms_per_submission\": %.3f, \"waits\": %llu, \"waits_blocked\": %llu, \"launch_to_done_ms\": %.3f, "
                "\"failed_tickets\": %llu, \"failures\": %zu, \"checksum\": %llu}\n",
                codec_name, a.via_registry ? "registry" : (batcher ? "batcher" : "per-stream"), a.direct ? "true" : "false", a.in_phase ? "true" : "false", a.streams, a.lookahead, std::min(a.threads, a.streams), n, secs, n / secs, frames_per_packet,
                bytes_in_per_packet, frames_per_packet * params<Codec>().channels * 4, n * bytes_in_per_packet / secs / 1e9,
                n * frames_per_packet * params<Codec>().channels * 4 / secs / 1e9, batches,
                (unsigned long long)(s1.launches - s0.launches), (unsigned long long)(s1.chunks - s0.chunks),
                (unsigned long long)s1.max_chains_per_launch, (unsigned long long)s1.staging_bytes, (unsigned long long)(s1.staging_bytes - s0.staging_bytes), (unsigned long long)s0.slots_peak, (unsigned long long)s1.slots_peak, (unsigned long long)s1.lanes,
                (double)(s1.mutex_wait_ns - s0.mutex_wait_ns) / 1e6, (unsigned long long)(s1.mutex_contended - s0.mutex_contended),
                (double)(s1.launch_host_ns - s0.launch_host_ns) / 1e6, (double)(s1.launch_api_ns - s0.launch_api_ns) / 1e6,
                (double)(s1.lane_wait_ns - s0.lane_wait_ns) / 1e6, (unsigned long long)(s1.group_allocs - s0.group_allocs), (unsigned long long)s1.blocks, (double)(s1.flag_wait_ns - s0.flag_wait_ns) / 1e6, (double)(s1.commit_to_launch_ns - s0.commit_to_launch_ns) / 1e6 / (double)std::max<uint64_t>(1, s1.submissions - s0.submissions), (unsigned long long)(s1.waits - s0.waits), (unsigned long long)(s1.waits_blocked - s0.waits_blocked), (double)(s1.launch_to_done_ns - s0.launch_to_done_ns) / 1e6 / (double)std::max<uint64_t>(1, s1.launches_timed - s0.launches_timed),
                (unsigned long long)(s1.failed_tickets - s0.failed_tickets), failures.load(),
                (unsigned long long)checksum.load());
    return failures.load() ? 1 : 0;
}

}  // namespace

int main(int argc, char **argv) {
    Args a;
    for (int i = 1; i < argc; ++i) {
        const std::string k = argv[i];
        auto val = [&]() -> size_t { return i + 1 < argc ? (size_t)std::strtoull(argv[++i], nullptr, 10) : 0; };
        if (k == "--codec" && i + 1 < argc) a.codec = argv[++i];
        else if (k == "--streams") a.streams = val();
        else if (k == "--lookahead") a.lookahead = val();
        else if (k == "--packets") a.packets = val();
        else if (k == "--threads") a.threads = val();
        else if (k == "--flush-mb") a.flush_mb = val();
        else if (k == "--warm") a.warm = val();
        else if (k == "--lanes") a.lanes = val();
        else if (k == "--per-stream") a.per_stream = true;
        else if (k == "--direct") a.direct = true;
        else if (k == "--in-phase") a.in_phase = true;
        else if (k == "--via-registry") a.via_registry = true;
        else {
            std::fprintf(stderr, "unknown argument %s\n", k.c_str());
            return 2;
        }
    }
    if (!a.streams || !a.lookahead || !a.packets || !a.threads) return 2;
    if (!a.warm) a.warm = 4 * a.lookahead;  // past the cold start (every stream's first batch is a launch of its own) and the pool's growth
    try {
        if (a.codec == "aac") return run<AacLc>(a, "aac", 1024, 2 * 1024 * 4 + 2);
        if (a.codec == "aacd") return run<AacLcCoded>(a, "aacd", 1024, 2 * 1024 * 4 + 2 + 644 + 55);  // (+ 0.6 TNS filters of 92 B)
        if (a.codec == "mp3") return run<Mp3>(a, "mp3", 1152, 4 * 576 * 4 + 16);
        if (a.codec == "mp3h") return run<Mp3Huffman>(a, "mp3h", 1152, 4 * 576 * 2 + 4 * 52 + 2 * 48 + 16);
        // (Vorbis: 8 channels, nine blocks in ten long -- the per-packet figures are those of a long block after a long block)
        if (a.codec == "vorbis") return run<Vorbis>(a, "vorbis", 1024, 8 * 1024 * 4);
        if (a.codec == "flac") return run<Flac>(a, "flac", 4096, 2 * 4096 * 4 + 2 * 132 + 1);
    } catch (const std::exception &e) {
        std::fprintf(stderr, "decoders_bench: %s\n", e.what());
        return 1;
    }
    std::fprintf(stderr, "unknown codec %s\n", a.codec.c_str());
    return 2;
}
