// LookaheadDecoder (include/symaccel.hpp): the AudioDecoder method set over batched device calls (AAC-LC, MP3, Vorbis, FLAC).  A synthetic track of
// parsed packets is decoded through decode() one packet at a time; every returned buffer must be bit-identical to what a
// frame-by-frame decoder (the CPU oracle, linked as the checker only) produces -- across batch boundaries, across a
// reset() (seek) and across a discontinuity without reset().  Built against the real library on the GPU box and against
// the CPU emulation build elsewhere (tests/test_lookahead.py).
#include <cstdio>
#include <cstring>
#include <memory>
#include <random>
#include <vector>

#include "symaccel.hpp"
#include "symoracle.h"

using namespace symphonia_accel;
using namespace symphonia_accel::codecs;

static int g_failures = 0;
#define EXPECT(cond, ...)                                    \
    do {                                                     \
        if (!(cond)) {                                       \
            ++g_failures;                                    \
            std::printf("FAIL %s:%d: ", __FILE__, __LINE__); \
            std::printf(__VA_ARGS__);                        \
            std::printf("\n");                               \
        }                                                    \
    } while (0)

static bool same_bits(const float *a, const float *b, size_t n) {
    for (size_t i = 0; i < n; ++i)
        if (!(a[i] == b[i])) return false;  // (+0 == -0; the tests use no NaN)
    return true;
}

// ---- AAC: a legal window-sequence walk per packet, both channels share it (like a common_window CPE)
static std::vector<AacLc::Packet> aac_track(size_t n, size_t nch, unsigned seed) {
    std::mt19937 rng(seed);
    std::normal_distribution<float> nd(0.0f, 50.0f);
    std::vector<AacLc::Packet> t(n);
    int cur = 0, prev_shape = 1;
    for (size_t i = 0; i < n; ++i) {
        cur = (cur == 0 || cur == 3) ? ((rng() % 4 == 0) ? 1 : 0) : ((rng() % 2) ? 2 : 3);
        const int shape = (int)(rng() % 2);
        t[i].ts = 1000 + 1024 * i;
        t[i].coeffs.resize(nch * 1024);
        for (auto &v : t[i].coeffs) v = nd(rng);
        t[i].side.assign(nch, SYMACCEL_AAC_SIDE((unsigned)cur, (unsigned)shape, (unsigned)prev_shape));
        prev_shape = shape;
    }
    return t;
}

struct AacOracle {  // a frame-by-frame decoder: one so_aac_synth_batch call per packet and channel
    size_t nch;
    std::vector<float> delay;
    explicit AacOracle(size_t c) : nch(c), delay(c * 1024, 0.0f) {}
    void reset() { std::fill(delay.begin(), delay.end(), 0.0f); }
    std::vector<float> decode(const AacLc::Packet &p) {
        std::vector<float> pcm(nch * 1024);
        for (size_t c = 0; c < nch; ++c) so_aac_synth_batch(p.coeffs.data() + c * 1024, &p.side[c], delay.data() + c * 1024, pcm.data() + c * 1024, 1, 1);
        return pcm;
    }
};

static void test_aac(Context &ctx, size_t lookahead, Batcher *batcher = nullptr) {
    const size_t nch = 2, n = 41;
    auto track = aac_track(n, nch, 7 + (unsigned)lookahead);
    size_t cursor = 0;  // the demuxer's read position: decode(track[i]) is called with cursor == i + 1
    auto peek = [&]() -> std::optional<AacLc::Packet> {
        if (cursor >= track.size()) return std::nullopt;
        return track[cursor++];
    };
    std::optional<LookaheadDecoder<AacLc>> holder;
    if (batcher) holder.emplace(*batcher, AacLc::Params{nch}, lookahead, peek);
    else holder.emplace(ctx, AacLc::Params{nch}, lookahead, peek);
    LookaheadDecoder<AacLc> &dec = *holder;
    AacOracle ref(nch);
    EXPECT(dec.last_decoded().is_empty(), "last_decoded() must be empty before the first decode");
    auto step = [&](size_t i) {
        if (cursor <= i) cursor = i + 1;
        const AudioBufferRef &buf = dec.decode(track[i]);
        const std::vector<float> want = ref.decode(track[i]);
        EXPECT(buf.frames == 1024 && buf.planes.size() == nch, "buffer shape at packet %zu", i);
        for (size_t c = 0; c < nch; ++c)
            EXPECT(same_bits(buf.planes[c], want.data() + c * 1024, 1024), "AAC K=%zu packet %zu channel %zu differs from the frame-by-frame decoder", lookahead, i, c);
        EXPECT(dec.last_decoded().planes == buf.planes && dec.last_decoded().frames == 1024, "last_decoded() != what decode() returned");
    };
    for (size_t i = 0; i < 17; ++i) step(i);
    // seek: the caller resets the decoder and continues somewhere else (audio.rs:252-257)
    dec.reset();
    ref.reset();
    EXPECT(dec.last_decoded().is_empty(), "reset() must clear the buffer");
    cursor = 23;
    for (size_t i = 23; i < 30; ++i) step(i);
    // a discontinuity WITHOUT reset (packets dropped): both decoders just carry their state on
    cursor = 33;
    for (size_t i = 33; i < n; ++i) step(i);
    if (lookahead > 1) EXPECT(dec.batches_run() < n, "no batching happened (%zu batches)", dec.batches_run());
}

// ---- MP3
static std::vector<Mp3::Packet> mp3_track(size_t n, size_t nch, size_t ngr, unsigned seed) {
    std::mt19937 rng(seed);
    std::normal_distribution<float> nd(0.0f, 0.1f);
    std::vector<Mp3::Packet> t(n);
    int state = 0;  // 0 long, 1 after start (short run), 2 end pending
    for (size_t i = 0; i < n; ++i) {
        t[i].ts = 5000 + 1152 * i;
        t[i].xr.resize(ngr * nch * 576);
        t[i].side.resize(ngr * nch);
        for (size_t gr = 0; gr < ngr; ++gr) {
            int bt;
            if (state == 0) bt = (rng() % 5 == 0) ? 1 : 0;
            else if (state == 1) bt = 2;
            else bt = 3;
            state = bt == 1 ? 1 : (bt == 2 ? ((rng() % 2) ? 1 : 2) : 0);
            const unsigned rz = 2 * (unsigned)(rng() % 289);
            for (size_t c = 0; c < nch; ++c) {
                float *x = t[i].xr.data() + (gr * nch + c) * 576;
                for (unsigned k = 0; k < 576; ++k) x[k] = k < rz ? nd(rng) : 0.0f;
                t[i].side[gr * nch + c] = symaccel_mp3_side{(uint8_t)bt, (uint8_t)(bt == 2 && rng() % 3 == 0), (uint16_t)rz};
            }
        }
    }
    return t;
}

static void test_mp3(Context &ctx, size_t lookahead, Batcher *batcher = nullptr) {
    const size_t nch = 2, ngr = 2, n = 25;
    const int sr = 1;
    auto track = mp3_track(n, nch, ngr, 11 + (unsigned)lookahead);
    size_t cursor = 0;
    auto peek = [&]() -> std::optional<Mp3::Packet> {
        if (cursor >= track.size()) return std::nullopt;
        return track[cursor++];
    };
    std::optional<LookaheadDecoder<Mp3>> holder;
    if (batcher) holder.emplace(*batcher, Mp3::Params{nch, ngr, sr}, lookahead, peek);
    else holder.emplace(ctx, Mp3::Params{nch, ngr, sr}, lookahead, peek);
    LookaheadDecoder<Mp3> &dec = *holder;
    std::vector<float> ov(nch * 576, 0.0f), vv(nch * 1024, 0.0f);
    std::vector<int32_t> vf(nch, 0);
    for (size_t i = 0; i < n; ++i) {
        if (i == 12) {  // seek
            dec.reset();
            std::fill(ov.begin(), ov.end(), 0.0f);
            std::fill(vv.begin(), vv.end(), 0.0f);
            std::fill(vf.begin(), vf.end(), 0);
            cursor = i;
        }
        if (cursor <= i) cursor = i + 1;
        const AudioBufferRef &buf = dec.decode(track[i]);
        EXPECT(buf.frames == 1152, "mp3 frames per packet");
        for (size_t c = 0; c < nch; ++c) {
            float want[1152];
            for (size_t gr = 0; gr < ngr; ++gr) {
                const symaccel_mp3_side &sd = track[i].side[gr * nch + c];
                const uint8_t side[4] = {sd.block_type, sd.is_mixed, (uint8_t)(sd.rzero & 255), (uint8_t)(sd.rzero >> 8)};
                so_mp3_synth_batch(track[i].xr.data() + (gr * nch + c) * 576, side, sr, ov.data() + c * 576, vv.data() + c * 1024, &vf[c], want + gr * 576, 1, 1);
            }
            EXPECT(same_bits(buf.planes[c], want, 1152), "MP3 K=%zu packet %zu channel %zu differs from the frame-by-frame decoder", lookahead, i, c);
        }
    }
}

// ---- MP3 one stage earlier: int16 Huffman samples + records in, the device requantises and runs the joint stereo
static void test_mp3_huffman(Context &ctx, size_t lookahead, Batcher *batcher = nullptr) {
    const size_t nch = 2, ngr = 2, n = 19;
    const int sr = 0;
    std::mt19937 rng(501 + (unsigned)lookahead);
    std::vector<Mp3Huffman::Packet> track(n);
    for (size_t i = 0; i < n; ++i) {
        Mp3Huffman::Packet &p = track[i];
        p.ts = 7000 + 1152 * i;
        p.quant.resize(ngr * nch * 576);
        p.rq.resize(ngr * nch);
        p.stereo.resize(ngr);
        p.side.resize(ngr * nch);
        for (size_t gr = 0; gr < ngr; ++gr) {
            const unsigned rz0 = 2 * (unsigned)(rng() % 289), rz1 = 2 * (unsigned)(rng() % 289);
            const unsigned flags = (unsigned)(rng() % 4) | SYMACCEL_MP3_ST_MPEG1;
            symaccel_mp3_stereo &sd = p.stereo[gr];
            std::memset(&sd, 0, sizeof sd);
            sd.flags = (uint8_t)flags;
            sd.rzero0 = (uint16_t)rz0;
            sd.rzero1 = (uint16_t)rz1;
            for (int s = 0; s < 39; ++s) sd.scalefacs1[s] = (uint8_t)(rng() % 8);
            const unsigned end = rz0 > rz1 ? rz0 : rz1;
            for (size_t c = 0; c < nch; ++c) {
                const unsigned rz = c == 0 ? rz0 : rz1;
                int16_t *q = p.quant.data() + (gr * nch + c) * 576;
                for (unsigned k = 0; k < 576; ++k) q[k] = k < rz ? (int16_t)((int)(rng() % 61) - 30) : (int16_t)0;
                symaccel_mp3_requant &r = p.rq[gr * nch + c];
                std::memset(&r, 0, sizeof r);
                r.global_gain = (uint8_t)(140 + rng() % 40);
                r.flags = (uint8_t)(rng() % 4);
                r.rzero = (uint16_t)rz;
                for (int s = 0; s < 39; ++s) r.scalefacs[s] = (uint8_t)(rng() % 4);
                p.side[gr * nch + c] = symaccel_mp3_side{0, 0, (uint16_t)((flags & 3u) ? end : rz)};
            }
        }
    }
    size_t cursor = 0;
    auto peek = [&]() -> std::optional<Mp3Huffman::Packet> {
        if (cursor >= track.size()) return std::nullopt;
        return track[cursor++];
    };
    std::optional<LookaheadDecoder<Mp3Huffman>> holder;
    if (batcher) holder.emplace(*batcher, Mp3Huffman::Params{nch, ngr, sr}, lookahead, peek);
    else holder.emplace(ctx, Mp3Huffman::Params{nch, ngr, sr}, lookahead, peek);
    LookaheadDecoder<Mp3Huffman> &dec = *holder;
    std::vector<float> ov(nch * 576, 0.0f), vv(nch * 1024, 0.0f);
    std::vector<int32_t> vf(nch, 0);
    for (size_t i = 0; i < n; ++i) {
        if (i == 8) {  // seek
            dec.reset();
            std::fill(ov.begin(), ov.end(), 0.0f);
            std::fill(vv.begin(), vv.end(), 0.0f);
            std::fill(vf.begin(), vf.end(), 0);
            cursor = i;
        }
        if (cursor <= i) cursor = i + 1;
        const AudioBufferRef &buf = dec.decode(track[i]);
        const Mp3Huffman::Packet &p = track[i];
        float want[2][1152];
        for (size_t gr = 0; gr < ngr; ++gr) {
            float xr[2][576];
            for (size_t c = 0; c < nch; ++c)
                so_mp3_requantize(p.quant.data() + (gr * nch + c) * 576, reinterpret_cast<const so_mp3_requant *>(&p.rq[gr * nch + c]), sr, xr[c]);
            so_mp3_stereo(xr[0], xr[1], reinterpret_cast<const so_mp3_stereo_desc *>(&p.stereo[gr]), sr);
            for (size_t c = 0; c < nch; ++c) {
                const symaccel_mp3_side &sd = p.side[gr * nch + c];
                const uint8_t side[4] = {sd.block_type, sd.is_mixed, (uint8_t)(sd.rzero & 255), (uint8_t)(sd.rzero >> 8)};
                so_mp3_synth_batch(xr[c], side, sr, ov.data() + c * 576, vv.data() + c * 1024, &vf[c], want[c] + gr * 576, 1, 1);
            }
        }
        for (size_t c = 0; c < nch; ++c)
            EXPECT(same_bits(buf.planes[c], want[c], 1152), "MP3 (Huffman in) K=%zu packet %zu channel %zu differs from the frame-by-frame decoder", lookahead, i, c);
    }
}

// ---- Vorbis: mixed block sizes, a variable number of frames per packet (none for the first block after a reset)
static void test_vorbis(Context &ctx, size_t lookahead, int e0, int e1, Batcher *batcher = nullptr) {
    const size_t nch = 2, n = 37;
    const size_t bs[2] = {(size_t)1 << e0, (size_t)1 << e1};
    std::mt19937 rng(100 + (unsigned)lookahead + (unsigned)e1);
    std::normal_distribution<float> nd(0.0f, 0.25f);
    std::vector<Vorbis::Packet> track(n);
    for (size_t i = 0; i < n; ++i) {
        track[i].ts = 9000 + 7 * i;
        track[i].long_block = rng() % 3 != 0;
        track[i].spectra.resize(nch * bs[track[i].long_block] / 2);
        for (auto &v : track[i].spectra) v = nd(rng);
    }
    size_t cursor = 0;
    auto peek = [&]() -> std::optional<Vorbis::Packet> {
        if (cursor >= track.size()) return std::nullopt;
        return track[cursor++];
    };
    std::optional<LookaheadDecoder<Vorbis>> holder;
    if (batcher) holder.emplace(*batcher, Vorbis::Params{nch, e0, e1}, lookahead, peek);
    else holder.emplace(ctx, Vorbis::Params{nch, e0, e1}, lookahead, peek);
    LookaheadDecoder<Vorbis> &dec = *holder;
    // the frame-by-frame decoder: one oracle call per packet and channel, state carried like DspChannel does
    std::vector<int32_t> prev(nch, -1);
    std::vector<float> ov(nch * bs[1] / 2, 0.0f);
    auto step = [&](size_t i) {
        if (cursor <= i) cursor = i + 1;
        const AudioBufferRef &buf = dec.decode(track[i]);
        const size_t nb = bs[track[i].long_block];
        for (size_t c = 0; c < nch; ++c) {
            const bool emits = prev[c] >= 0;
            const size_t frames = emits ? (bs[prev[c]] + nb) / 4 : 0, slots = emits ? frames : nb / 2;
            std::vector<float> want(slots, 0.0f);
            const uint8_t flag = track[i].long_block ? 1 : 0;
            so_vorbis_synth_batch(e0, e1, track[i].spectra.data() + c * nb / 2, nb / 2, &flag, &prev[c], ov.data() + c * bs[1] / 2, want.data(), slots, 1, 1);
            EXPECT(buf.frames == frames, "Vorbis K=%zu packet %zu: %zu frames, expected %zu", lookahead, i, buf.frames, frames);
            EXPECT(buf.frames != frames || same_bits(buf.planes[c], want.data(), frames), "Vorbis %d/%d K=%zu packet %zu channel %zu differs from the frame-by-frame decoder", e0, e1, lookahead, i, c);
        }
    };
    for (size_t i = 0; i < 15; ++i) step(i);
    dec.reset();  // seek
    std::fill(prev.begin(), prev.end(), -1);
    std::fill(ov.begin(), ov.end(), 0.0f);
    cursor = 19;
    for (size_t i = 19; i < 27; ++i) step(i);
    cursor = 30;  // packets dropped without reset()
    for (size_t i = 30; i < n; ++i) step(i);
}

// ---- FLAC: integer samples, no carried state, block sizes that differ from packet to packet
static void test_flac(Context &ctx, size_t lookahead, size_t nch, uint32_t bps, Batcher *batcher = nullptr) {
    const size_t n = 23;
    std::mt19937 rng(77 + (unsigned)lookahead + (unsigned)nch);
    std::vector<Flac::Packet> track(n);
    for (size_t i = 0; i < n; ++i) {
        Flac::Packet &p = track[i];
        p.ts = 900 + 4096 * i;
        p.blocksize = i + 1 == n ? 777 : (rng() % 4 == 0 ? 1152 : 4096);
        p.words.resize(nch * p.blocksize);
        p.desc.resize(nch);
        p.coeffs.assign(nch * 32, 0);
        p.pair_mode = nch == 2 ? (uint8_t)(rng() % 4) : 0;
        for (size_t c = 0; c < nch; ++c) {
            const unsigned kind = rng() % 3;
            const unsigned order = kind == 0 ? 0 : (kind == 1 ? rng() % 5 : 1 + rng() % 32);
            const unsigned shift = kind == 2 ? 8 + rng() % 6 : 0;
            const unsigned wasted = rng() % 5 == 0 ? 1 + rng() % 3 : 0;
            p.desc[c] = symaccel_flac_desc{(uint8_t)kind, (uint8_t)order, (uint8_t)shift, (uint8_t)wasted};
            if (kind == 2)
                for (unsigned j = 0; j < order; ++j) p.coeffs[c * 32 + j] = (int32_t)(rng() % 2001) - 1000;
            for (size_t k = 0; k < p.blocksize; ++k) p.words[c * p.blocksize + k] = (int32_t)(rng() % 4001) - 2000;
        }
    }
    size_t cursor = 0;
    auto peek = [&]() -> std::optional<Flac::Packet> {
        if (cursor >= track.size()) return std::nullopt;
        return track[cursor++];
    };
    std::unique_ptr<LookaheadDecoder<Flac>> holder(batcher ? new LookaheadDecoder<Flac>(*batcher, Flac::Params{nch, bps, 4096}, lookahead, peek)
                                                           : new LookaheadDecoder<Flac>(ctx, Flac::Params{nch, bps, 4096}, lookahead, peek));
    LookaheadDecoder<Flac> &dec = *holder;
    for (size_t i = 0; i < n; ++i) {
        if (i == 9) {  // seek
            dec.reset();
            EXPECT(dec.last_decoded().is_empty(), "reset() must clear the buffer");
            cursor = i;
        }
        if (i == 15) cursor = i;  // (i == 14 is skipped below: packets dropped without reset())
        if (i == 14) continue;
        if (cursor <= i) cursor = i + 1;
        const AudioBufferRefS32 &buf = dec.decode(track[i]);
        const Flac::Packet &p = track[i];
        EXPECT(buf.frames == p.blocksize, "flac frames per packet: %zu vs %zu", buf.frames, p.blocksize);
        std::vector<int32_t> want(p.words);
        for (size_t c = 0; c < nch; ++c) {
            const uint8_t d[4] = {p.desc[c].kind, p.desc[c].order, p.desc[c].shift, p.desc[c].wasted_bits};
            so_flac_restore_batch(want.data() + c * p.blocksize, d, p.coeffs.data() + c * 32, 1, p.blocksize);
        }
        if (nch == 2) so_flac_decorrelate(p.pair_mode, want.data(), want.data() + p.blocksize, p.blocksize);
        for (size_t c = 0; c < nch; ++c) {
            so_flac_shl(want.data() + c * p.blocksize, p.blocksize, 32 - bps);
            EXPECT(std::memcmp(buf.planes[c], want.data() + c * p.blocksize, p.blocksize * 4) == 0,
                   "FLAC K=%zu packet %zu channel %zu differs from the frame-by-frame decoder", lookahead, i, c);
        }
    }
}

// ---- AAC-LC one stage earlier: coded spectra + joint-stereo descriptors + TNS filters in, the device decodes, filters and synthesises
static const uint16_t kSwbLong[] = {0, 4, 8, 12, 16, 20, 24, 28, 32, 36, 40, 48, 56, 64, 72, 80, 88, 96, 108, 120, 132, 144, 160, 176, 196, 216,
                                    240, 264, 292, 320, 352, 384, 416, 448, 480, 512, 544, 576, 608, 640, 672, 704, 736, 768, 800, 832, 864, 896,
                                    928, 1024};
static const uint16_t kSwbShort[] = {0, 4, 8, 12, 16, 20, 28, 36, 44, 56, 68, 80, 96, 112, 128};

static void test_aac_coded(Context &ctx, size_t lookahead, Batcher *batcher = nullptr) {
    const size_t nch = 3, n = 29;  // channels 0 / 1 a pair, channel 2 on its own
    std::mt19937 rng(77 + (unsigned)lookahead);
    std::normal_distribution<float> nd(0.0f, 40.0f);
    std::uniform_real_distribution<float> ud(-0.4f, 0.4f);
    std::vector<AacLcCoded::Packet> track(n);
    int cur = 0, prev_shape = 1;
    for (size_t i = 0; i < n; ++i) {
        AacLcCoded::Packet &p = track[i];
        cur = (cur == 0 || cur == 3) ? ((rng() % 4 == 0) ? 1 : 0) : ((rng() % 2) ? 2 : 3);
        const int shape = (int)(rng() % 2);
        p.ts = 3000 + 1024 * i;
        p.coeffs.resize(nch * 1024);
        for (auto &v : p.coeffs) v = nd(rng);
        p.side.assign(nch, SYMACCEL_AAC_SIDE((unsigned)cur, (unsigned)shape, (unsigned)prev_shape));
        prev_shape = shape;
        const bool is_short = cur == 2;
        if (rng() % 4 != 0) {  // the pair is jointly coded in three packets of four
            symaccel_aac_js_frame d;
            std::memset(&d, 0, sizeof d);
            d.num_windows = is_short ? 8 : 1;
            d.max_sfb = (uint8_t)(is_short ? 1 + rng() % 14 : 1 + rng() % 49);
            for (int s = 0; s < 128; ++s) {
                d.mode[s] = (uint8_t)(rng() % 3);
                d.scale[s] = ud(rng) * 4.0f;
            }
            p.joint.emplace_back(0, d);
        }
        for (size_t c = 0; c < nch; ++c) {
            if (rng() % 3) continue;  // TNS on a third of the channel-frames
            const int windows = is_short ? 1 + (int)(rng() % 3) : 1;
            for (int k = 0; k < windows; ++k) {
                symaccel_aac_tns_filter f;
                std::memset(&f, 0, sizeof f);
                const int w = is_short ? (int)(rng() % 8) : 0, span = is_short ? 128 : 1024;
                int a = (int)(rng() % span), b2 = (int)(rng() % span);
                if (a > b2) std::swap(a, b2);
                if (a == b2) b2 = a + 1;
                bool clash = false;  // (the filters of one frame cover disjoint ranges, tns.rs:163-166: one per window here)
                for (const auto &g : p.tns) clash = clash || (g.frame == c && g.start / 128 == (unsigned)w && is_short);
                if (clash) continue;
                f.frame = (uint32_t)c;
                f.start = (uint16_t)(128 * w + a);
                f.end = (uint16_t)(128 * w + b2);
                f.order = (uint8_t)(1 + rng() % (is_short ? 7 : 12));
                f.direction = (uint8_t)(rng() % 2);
                for (int q = 0; q < f.order; ++q) f.lpc[q] = ud(rng);
                p.tns.push_back(f);
            }
        }
    }
    AacLcCoded::Params params;
    params.channels = nch;
    params.swb_long.assign(kSwbLong, kSwbLong + sizeof kSwbLong / sizeof kSwbLong[0]);
    params.swb_short.assign(kSwbShort, kSwbShort + sizeof kSwbShort / sizeof kSwbShort[0]);
    size_t cursor = 0;
    auto peek = [&]() -> std::optional<AacLcCoded::Packet> {
        if (cursor >= track.size()) return std::nullopt;
        return track[cursor++];
    };
    std::optional<LookaheadDecoder<AacLcCoded>> holder;
    if (batcher) holder.emplace(*batcher, params, lookahead, peek);
    else holder.emplace(ctx, params, lookahead, peek);
    LookaheadDecoder<AacLcCoded> &dec = *holder;
    // the frame-by-frame decoder: joint stereo on the pair (cpe.rs:110-157), then per channel TNS and Dsp::synth (ics/mod.rs:449-468)
    std::vector<float> delay(nch * 1024, 0.0f);
    auto step = [&](size_t i) {
        if (cursor <= i) cursor = i + 1;
        const AudioBufferRef &buf = dec.decode(track[i]);
        const AacLcCoded::Packet &p = track[i];
        std::vector<float> x = p.coeffs;
        for (const auto &j : p.joint)
            so_aac_joint_stereo(x.data() + j.first * 1024, x.data() + (j.first + 1) * 1024, j.second.num_windows, j.second.max_sfb,
                                j.second.num_windows == 1 ? kSwbLong : kSwbShort, j.second.mode, j.second.scale);
        for (const auto &f : p.tns) so_aac_tns_filter(x.data() + f.frame * 1024, f.start, f.end, f.order, f.direction, f.lpc);
        for (size_t c = 0; c < nch; ++c) {
            float want[1024];
            so_aac_synth_batch(x.data() + c * 1024, &p.side[c], delay.data() + c * 1024, want, 1, 1);
            EXPECT(same_bits(buf.planes[c], want, 1024), "AAC (coded spectra in) K=%zu packet %zu channel %zu differs from the frame-by-frame decoder", lookahead, i, c);
        }
    };
    for (size_t i = 0; i < 13; ++i) step(i);
    dec.reset();  // seek
    std::fill(delay.begin(), delay.end(), 0.0f);
    cursor = 17;
    for (size_t i = 17; i < 22; ++i) step(i);
    cursor = 24;  // packets dropped without reset()
    for (size_t i = 24; i < n; ++i) step(i);
}

// ---- many streams, one batcher: S AAC decoders called round-robin like a server's worker would; every buffer equals the
// frame-by-frame decoder's, and the batcher ran far fewer launches than the decoders ran batches
// via_registry: the decoders come from CodecRegistry::make_audio_decoder (registry.rs:330-341), which is handed (params, options) and the
// packet source alone; they must still meet in ONE batcher -- the process-wide one -- and share launches
static void test_cross_stream(Context &ctx, size_t n_streams, size_t lookahead, bool direct = false, bool via_registry = false) {
    const size_t n = 37;
    std::unique_ptr<Batcher> own;
    if (!via_registry) own.reset(new Batcher(ctx));
    Batcher &batcher = via_registry ? Batcher::shared() : *own;
    const symaccel_batcher_stats before = batcher.stats();
    CodecRegistry registry;
    if (via_registry) {
        try {
            registry.make_audio_decoder<AacLc>(AacLc::Params{2}, AudioDecoderOptions{}, LookaheadDecoder<AacLc>::Peek([]() -> std::optional<AacLc::Packet> { return std::nullopt; }));
            EXPECT(false, "an empty registry made a decoder");
        } catch (const Error &e) {
            EXPECT(e.kind == Error::Kind::Unsupported, "empty registry: wrong error kind");
        }
        register_enabled_codecs(registry);
        EXPECT(registry.is_registered<AacLc>() && registry.is_registered<Flac>(), "register_enabled_codecs");
    }
    struct Stream {
        std::vector<AacLc::Packet> track;
        size_t cursor = 0, nch = 2;
        std::unique_ptr<LookaheadDecoder<AacLc>> dec;
        std::unique_ptr<AacOracle> ref;
    };
    std::vector<std::unique_ptr<Stream>> streams;
    for (size_t s = 0; s < n_streams; ++s) {
        streams.emplace_back(new Stream());
        Stream *st = streams.back().get();
        st->nch = 1 + s % 2;
        st->track = aac_track(n, st->nch, 900 + (unsigned)s);
        if (direct) {
            // zero-copy parse (LookaheadDecoder::Direct): the "parser" writes packet by packet into the batcher's slot.  Every third
            // stream's demuxer announces three packets more than it has: those batches end short and are re-packed.
            LookaheadDecoder<AacLc>::Direct d;
            const size_t lie = s % 3 == 2 ? 3 : 0;
            d.avail = [st, lie]() { return st->track.size() - std::min(st->track.size(), st->cursor) + lie; };
            d.parse_into = [st](const AacLc::BatchView &v, size_t i) -> std::optional<std::uint64_t> {
                if (st->cursor >= st->track.size()) return std::nullopt;
                const AacLc::Packet &p = st->track[st->cursor];
                for (size_t c = 0; c < st->nch; ++c) {
                    std::memcpy(v.coeffs_at(c, i), p.coeffs.data() + c * 1024, 4096);
                    v.side_at(c, i) = p.side[c];
                }
                ++st->cursor;
                return p.ts;
            };
            if (via_registry)
                st->dec = registry.make_audio_decoder<AacLc>(AacLc::Params{st->nch}, AudioDecoderOptions{false, lookahead}, d);
            else
                st->dec.reset(new LookaheadDecoder<AacLc>(batcher, AacLc::Params{st->nch}, lookahead, d));
        } else {
            LookaheadDecoder<AacLc>::Peek peek = [st]() -> std::optional<AacLc::Packet> {
                if (st->cursor >= st->track.size()) return std::nullopt;
                return st->track[st->cursor++];
            };
            if (via_registry)
                st->dec = registry.make_audio_decoder<AacLc>(AacLc::Params{st->nch}, AudioDecoderOptions{false, lookahead}, peek);
            else
                st->dec.reset(new LookaheadDecoder<AacLc>(batcher, AacLc::Params{st->nch}, lookahead, peek));
        }
        st->ref.reset(new AacOracle(st->nch));
    }
    size_t batches = 0;
    for (size_t i = 0; i < n; ++i)
        for (auto &up : streams) {
            Stream &st = *up;
            if (st.cursor <= i) st.cursor = i + 1;
            const AudioBufferRef &buf = st.dec->decode(st.track[i]);
            const std::vector<float> want = st.ref->decode(st.track[i]);
            for (size_t c = 0; c < st.nch; ++c)
                EXPECT(same_bits(buf.planes[c], want.data() + c * 1024, 1024), "cross-stream S=%zu K=%zu packet %zu differs", n_streams, lookahead, i);
        }
    for (auto &up : streams) batches += up->dec->batches_run();
    symaccel_batcher_stats stats = batcher.stats();
    stats.submissions -= before.submissions;  // (the shared batcher lives across tests)
    stats.launches -= before.launches;
    EXPECT(stats.submissions >= batches, "submissions %llu < batches %zu", (unsigned long long)stats.submissions, batches);
    if (lookahead >= 4 && n_streams >= 4)
        EXPECT(stats.launches * 2 <= stats.submissions, "the batcher did not coalesce: %llu launches for %llu submissions", (unsigned long long)stats.launches,
               (unsigned long long)stats.submissions);
    std::printf("cross-stream%s%s S=%zu K=%zu: %llu submissions in %llu launches (largest: %llu chains)\n", direct ? " (direct)" : "", via_registry ? " (registry)" : "", n_streams, lookahead,
                (unsigned long long)stats.submissions, (unsigned long long)stats.launches, (unsigned long long)stats.max_chains_per_launch);
    streams.clear();  // (decoders release their tickets before the batcher goes)
}

int main(int argc, char **argv) {
    if (argc > 1 && std::strcmp(argv[1], "--expect-no-device") == 0) {
        try {
            Context ctx(0);
        } catch (const std::exception &) {
            std::printf("no device: refused as expected\n");
            return 0;
        }
        std::printf("a context was created without a device\n");
        return 1;
    }
    Context ctx(0);
    for (size_t k : {size_t(1), size_t(4), size_t(9), size_t(64)}) test_aac(ctx, k);
    for (size_t k : {size_t(1), size_t(3), size_t(8)}) test_mp3(ctx, k);
    for (size_t k : {size_t(1), size_t(6)}) test_mp3_huffman(ctx, k);
    for (size_t k : {size_t(1), size_t(4), size_t(11)}) test_aac_coded(ctx, k);
    for (size_t k : {size_t(1), size_t(5), size_t(16)}) test_vorbis(ctx, k, 8, 11);
    test_vorbis(ctx, 6, 6, 9);
    for (size_t k : {size_t(1), size_t(4), size_t(32)}) test_flac(ctx, k, 2, 16);
    test_flac(ctx, 5, 1, 24);
    test_flac(ctx, 7, 3, 20);
    {  // the same single-stream scripts (seek, discontinuity) through the cross-stream batcher
        Batcher batcher(ctx);
        for (size_t k : {size_t(1), size_t(4), size_t(9), size_t(64)}) test_aac(ctx, k, &batcher);
        for (size_t k : {size_t(1), size_t(3), size_t(8)}) test_mp3(ctx, k, &batcher);
        for (size_t k : {size_t(1), size_t(6)}) test_mp3_huffman(ctx, k, &batcher);
        for (size_t k : {size_t(1), size_t(4), size_t(11)}) test_aac_coded(ctx, k, &batcher);
        for (size_t k : {size_t(1), size_t(5), size_t(16)}) test_vorbis(ctx, k, 8, 11, &batcher);
        test_vorbis(ctx, 6, 6, 9, &batcher);
        test_vorbis(ctx, 4, 12, 13, &batcher);
        for (size_t k : {size_t(1), size_t(4), size_t(32)}) test_flac(ctx, k, 2, 16, &batcher);  // (the fused stereo form)
        test_flac(ctx, 5, 1, 24, &batcher);
        test_flac(ctx, 7, 3, 20, &batcher);
    }
    test_cross_stream(ctx, 1, 8);
    test_cross_stream(ctx, 7, 8);
    test_cross_stream(ctx, 16, 3);
    test_cross_stream(ctx, 7, 8, true);
    test_cross_stream(ctx, 16, 5, true);
    test_cross_stream(ctx, 7, 8, false, true);
    test_cross_stream(ctx, 16, 5, true, true);
    if (g_failures == 0) std::printf("all checks passed\n");
    return g_failures ? 1 : 0;
}
