// EXPERIMENT -- not part of the product build.  Included by aac.hip only when a development knob asks for it
// (SYMACCEL_TUNE_AAC_QUAD=0, symphonia_amd/build.py: tuned builds go to symphonia_amd/build/tuned/, never to the product library).
//
// The WAVEFRONT walk of AAC-LC synthesis: the round-1 / round-2 kernel, kept as the A/B partner of the workgroup walk
// (aac_synth_quad_kernel in aac.hip, the product since round 3; profiles/r03y_aac_sustained.txt has the comparison) together with
// its measurement-only build knobs: SYM_AAC_SINK (stores and prefetch unconditional), SYM_AAC_CLOCK (cycle counters written into
// the PCM), SYM_AAC_ABLATE (bit 0: no PCM stores, bit 1: every frame read from the chain's first frame, bit 2: no transform;
// results wrong by construction -- profiles/r03*_aac_ablate*.txt).
#pragma once

// MEASUREMENT ONLY (results are wrong by construction; never set in the product build, profiles/r03*_aac_ablate*.txt):
// bit 0: no PCM stores; bit 1: every frame's lines are read from the chain's first frame (an L2-resident 512 KiB instead of
// 512 MiB from HBM); bit 2: no transform -- the pre-twiddled lines are stored as they are (the kernel's load / prefetch /
// store skeleton at its own occupancy).  Which direction, or whether the arithmetic between them, binds the kernel.
#ifndef SYM_AAC_ABLATE
#define SYM_AAC_ABLATE 0
#endif
// SYM_AAC_SINK 1 (the wavefront walk, one frame in flight): gfx950 counts vector loads and stores with ONE in-order counter
// (vmcnt).  With the PCM stores under `if (emit)`, the prefetch under `if (t + 1 < t_end)` and the prefetched lines consumed
// at the top of the loop, the compiler has to assume "no store followed the prefetch" and waits with vmcnt(0) at the loop
// header: every frame the wavefront stood still until the four stores it had issued a moment earlier were acknowledged by
// memory, with nothing of its own in flight meanwhile.  Here the prefetch and the stores are unconditional (the halo frame's
// stores go to a per-wavefront slot of the context's sink buffer, the last frame re-reads itself) and the prefetched registers
// are touched at the END of the frame, behind the stores, in straight-line code: the wait becomes vmcnt(4) -- for the lines,
// not for the stores -- and the loop header needs none.
#ifndef SYM_AAC_SINK
#define SYM_AAC_SINK 0
#endif
// MEASUREMENT ONLY (needs SYM_AAC_SINK; corrupts the first PCM frame of every segment): shader cycles, 100 MHz ticks and the
// cycles spent in the wait for the prefetched frame, per wavefront (tools/kernel_clock_probe.py)
#ifndef SYM_AAC_CLOCK
#define SYM_AAC_CLOCK 0
#endif
// the wait for the prefetched frame, placed where it is called (the registers must be valid from here on)
__device__ __forceinline__ void touch_prefetch(float2 (&line)[8], unsigned &sb) {
#pragma unroll
    for (int s = 0; s < 8; ++s) asm volatile("" : "+v"(line[s].x), "+v"(line[s].y));
    asm volatile("" : "+v"(sb));
}
__global__ __launch_bounds__(64 * kWaves, SYM_AAC_MIN_WAVES) void aac_synth_kernel(
    DevTables tb, const float *__restrict__ coeffs, const uint8_t *__restrict__ side,
    const float *__restrict__ delay_in, float *__restrict__ delay_out, float *__restrict__ pcm, float *__restrict__ sink,
    unsigned frames_per_chain, unsigned seg_len, unsigned segs_per_chain, unsigned n_items) {
    __shared__ __attribute__((aligned(16))) float tabs[kTabFloats];
    __shared__ __attribute__((aligned(16))) float wave_lds[kWaves][kWaveLds];
#if SYM_AAC_VARIANT == 1
    __shared__ __attribute__((aligned(16))) c32 lane_tab[kLaneTabComplex];
    fill_lane_tables_lds(tb, lane_tab, (int)threadIdx.x, 64 * kWaves);
#endif

    // ---- shared tables -> LDS (once per workgroup)
    for (int i = (int)threadIdx.x; i < 1024; i += 64 * kWaves) {
        tabs[kTabTw + i] = reinterpret_cast<const float *>(tb.aac_tw_long)[i];
        tabs[kTabKbd + i] = tb.aac_kbd_long[i];
        tabs[kTabSine + i] = tb.aac_sine_long[i];
        if (i < 128) {
            tabs[kTabKbdShort + i] = tb.aac_kbd_short[i];
            tabs[kTabSineShort + i] = tb.aac_sine_short[i];
        }
    }
    __syncthreads();  // the only workgroup-wide barrier; wavefronts are independent from here on

    const int lane = (int)threadIdx.x & 63, wave = (int)threadIdx.x >> 6;
    const unsigned item = blockIdx.x * kWaves + (unsigned)wave;
    if (item >= n_items) return;
    float *ldsf = wave_lds[wave];
    c32 *lds = reinterpret_cast<c32 *>(ldsf);
    const c32 *tw = reinterpret_cast<const c32 *>(tabs + kTabTw);

    const unsigned chain = item / segs_per_chain, seg = item % segs_per_chain;
    const unsigned t_begin = seg * seg_len;
    const unsigned t_end = min(t_begin + seg_len, frames_per_chain);
    const size_t chain_base = (size_t)chain * frames_per_chain;

#if SYM_AAC_VARIANT == 1
    const LaneTablesLds lt = lane_tables_lds(tb, lane_tab, lane);
#else
    LaneTables lt;
    load_lane_tables(tb, lane, lt);
#endif

    // delay line, in slot layout: dl[h][0..3] = delay[4m2 + q], dl[h][4..7] = delay[1020 - 4m2 + q]
    float dl[2][8];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        if (t_begin == 0) {
            load_slot(delay_in + (size_t)chain * 1024, lane + 64 * h, dl[h]);
        } else {
#pragma unroll
            for (int q = 0; q < 8; ++q) dl[h][q] = 0.0f;
        }
    }

    // frame t_begin-1 is the halo: it only rebuilds the delay line
    const long t_first = t_begin == 0 ? 0 : (long)t_begin - 1;
    float2 line[8];  // line[s] = (spec[2m + 128 s], spec[2m + 128 s + 1]), 512 B coalesced per load
    {
        const float2 *src = reinterpret_cast<const float2 *>(coeffs + (chain_base + (size_t)t_first) * 1024);
#pragma unroll
        for (int s = 0; s < 8; ++s) line[s] = ld_line(src + lane + 64 * s);
    }
#if SYM_AAC_PREFETCH == 2
    float2 line2[8];  // the frame after that
    {
        const long t2 = t_first + 1 < (long)t_end ? t_first + 1 : t_first;
        const float2 *src = reinterpret_cast<const float2 *>(coeffs + (chain_base + (size_t)t2) * 1024);
#pragma unroll
        for (int s = 0; s < 8; ++s) line2[s] = ld_line(src + lane + 64 * s);
    }
#endif

    unsigned sb_next = side[chain_base + (size_t)t_first];  // side bytes are fetched one frame ahead, like the lines
#if SYM_AAC_CLOCK
    const unsigned long long clk_t0 = __builtin_readcyclecounter(), clk_w0 = wall_clock64();
    unsigned long long clk_wait = 0;
#endif
#if SYM_AAC_SINK
    touch_prefetch(line, sb_next);  // (nothing pending at the loop header, from either side)
    float *const sink_frame = sink + (size_t)(item % (unsigned)(kSinkBytes / 4096)) * 1024;
#endif
    // one frame of the walk.  Variant 1 wraps the body in a lambda whose `emit_c` is a compile-time true/false so that
    // the halo frame can be peeled; variant 0 keeps the plain loop (its register allocation is the measured one).
#if SYM_AAC_VARIANT == 1
    auto do_frame = [&](long t, auto emit_c) {
        const bool emit = emit_c;
#else
    for (long t = t_first; t < (long)t_end; ++t) {
        const bool emit = t >= (long)t_begin;
#endif
#if SYM_AAC_SINK
        // the side byte is the same for the whole wavefront: as a scalar the window-sequence branches are uniform branches, of
        // which exactly one runs (an exec-masked if / else has a static path through neither arm, and on that path no store
        // follows the prefetch: the bookkeeping described above would be back to vmcnt(0))
        const unsigned sb = (unsigned)__builtin_amdgcn_readfirstlane((int)sb_next);
#else
        const unsigned sb = sb_next;
#endif
        const int seq = (int)(sb & 3u);
        const int shape = (int)((sb >> 2) & 1u), prev_shape = (int)((sb >> 3) & 1u);
#if SYM_AAC_SINK
        float *frame_out = emit ? pcm + (chain_base + (size_t)t) * 1024 : sink_frame;  // (wave-uniform)
#define SYM_AAC_EMIT true
#else
        float *frame_out = pcm + (chain_base + (size_t)t) * 1024;
#define SYM_AAC_EMIT emit
#endif

        // ---- consume the prefetched lines
        c32 z[8];
        if (seq != EIGHT_SHORT) {
            // pre-twiddle z[m + 64 s] with tw[m + 64 s]; the mirrored (odd) line sits in lane 63-m's load 7-s
            const int mirror = (63 - lane) * 4;
#pragma unroll
            for (int s = 0; s < 8; ++s) {
                const float mirrored =
                    __int_as_float(__builtin_amdgcn_ds_bpermute(mirror, __float_as_int(line[7 - s].y)));
                z[s] = pre_twiddle(line[s].x, mirrored, tw[lane + 64 * s]);
            }
        } else {
#pragma unroll
            for (int s = 0; s < 8; ++s) {  // the short transform indexes lines per 128-line window: stage in LDS
                ldsf[short_row(s) + 2 * lane] = line[s].x;
                ldsf[short_row(s) + 2 * lane + 1] = line[s].y;
            }
            wave_sync();
        }
#if SYM_AAC_PREFETCH == 2
        // two frames in flight: frame t + 1 (loaded during frame t - 1) moves up, frame t + 2 is requested now
        if (t + 1 < (long)t_end) sb_next = side[chain_base + (size_t)t + 1];
#pragma unroll
        for (int s = 0; s < 8; ++s) line[s] = line2[s];
        if (t + 2 < (long)t_end) {
            const float2 *src = reinterpret_cast<const float2 *>(coeffs + (chain_base + (size_t)t + 2) * 1024);
#pragma unroll
            for (int s = 0; s < 8; ++s) line2[s] = ld_line(src + lane + 64 * s);
        }
#elif SYM_AAC_SINK
        {   // prefetch the next frame (the last frame of the walk re-reads itself: a valid address, never used)
            const size_t tn = t + 1 < (long)t_end ? (size_t)t + 1 : (size_t)t;
            sb_next = side[chain_base + tn];
            const float2 *src = reinterpret_cast<const float2 *>(coeffs + (chain_base + ((SYM_AAC_ABLATE & 2) ? (size_t)0 : tn)) * 1024);
#pragma unroll
            for (int s = 0; s < 8; ++s) line[s] = ld_line(src + lane + 64 * s);
        }
#else
        if (t + 1 < (long)t_end) {  // prefetch the next frame; it lands while this one is transformed
            sb_next = side[chain_base + (size_t)t + 1];
            const float2 *src = reinterpret_cast<const float2 *>(coeffs + (chain_base + ((SYM_AAC_ABLATE & 2) ? (size_t)0 : (size_t)t + 1)) * 1024);
#pragma unroll
            for (int s = 0; s < 8; ++s) line[s] = ld_line(src + lane + 64 * s);
        }
#endif

#if SYM_AAC_ABLATE & 4
        if (true) {
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                float dst[8];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    dst[2 * q] = z[4 * h + q].x + dl[h][2 * q];
                    dst[2 * q + 1] = z[4 * h + q].y + dl[h][2 * q + 1];
                }
                if (SYM_AAC_EMIT && !(SYM_AAC_ABLATE & 1)) st_slot(frame_out, lane + 64 * h, dst);
#pragma unroll
                for (int q = 0; q < 8; ++q) dl[h][q] = dst[q] * 0.5f;
            }
        } else
#endif
        if (seq != EIGHT_SHORT) {
            fft512_wave(z, lane, lds, lt);
            const float *wprev = tabs + (prev_shape ? kTabKbd : kTabSine);  // prev_long_win (dsp.rs:71-74)
            const float *wcur = tabs + (shape ? kTabKbd : kTabSine);        // long_win (dsp.rs:66-69)
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int m2 = lane + 64 * h;
                float x[8], x2[8];
                post_slot(lds, tw, m2, x, x2);
                // ---- output samples (dsp.rs:105-129): dst = delay + pcm * w, or delay alone
                float wo[8], dst[8];
                if (seq == LONG_STOP) {
                    const float *psw = tabs + (prev_shape ? kTabKbdShort : kTabSineShort);
                    stop_window4(psw, 4 * m2, wo);
                    stop_window4(psw, 1020 - 4 * m2, wo + 4);
                } else {
                    load_slot(wprev, m2, wo);
                }
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    const int j = q < 4 ? 4 * m2 + q : 1020 - 4 * m2 + (q - 4);
                    const float v = dl[h][q] + (x[q] * wo[q]);
                    dst[q] = (seq == LONG_STOP && j < kP0) ? dl[h][q] : v;
                }
                if (SYM_AAC_EMIT && !(SYM_AAC_ABLATE & 1)) st_slot(frame_out, m2, dst);
                // ---- delay for the next frame (dsp.rs:132-157): pcm[1024 + j] * long_win[1023 - j] (the
                // slot's two float4 read backwards), a short-window slope, or literal zero
                float wd[8];
                if (seq == LONG_START) {
                    const float *sw = tabs + (shape ? kTabKbdShort : kTabSineShort);
                    start_window4(sw, 4 * m2, wd);
                    start_window4(sw, 1020 - 4 * m2, wd + 4);
                } else {
                    float wr[8];
                    load_slot(wcur, m2, wr);
#pragma unroll
                    for (int q = 0; q < 8; ++q) wd[q] = wr[7 - q];
                }
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    const int j = q < 4 ? 4 * m2 + q : 1020 - 4 * m2 + (q - 4);
                    const float v = x2[q] * wd[q];
                    dl[h][q] = (seq == LONG_START && j >= kP1) ? 0.0f : v;
                }
            }
#if SYM_AAC_SINK
#if SYM_AAC_CLOCK
            const unsigned long long cw0 = __builtin_readcyclecounter();
#endif
            touch_prefetch(line, sb_next);  // vmcnt(4): the next frame's lines, not this frame's stores
#if SYM_AAC_CLOCK
            clk_wait += __builtin_readcyclecounter() - cw0;
#endif
#endif
            wave_sync();  // Z in LDS is overwritten by the next frame
        } else {
            // ---- eight short windows (rare): everything through LDS in natural order
            float *dly = ldsf + kShortRowsEnd;  // after the eight (skewed) short-window rows
            imdct_short_wave(lane, ldsf, tb.aac_tw_short, lt);  // H[w] = ldsf[short_row(w) ..]
#pragma unroll
            for (int h = 0; h < 2; ++h) store_slot(dly, lane + 64 * h, dl[h]);  // (the FFT work array overlapped dly)
            wave_sync();
            const float *sw = tabs + (shape ? kTabKbdShort : kTabSineShort);
            const float *psw = tabs + (prev_shape ? kTabKbdShort : kTabSineShort);
#pragma unroll 1
            for (int e = 0; e < 4; ++e) {  // four consecutive samples per lane and round
                const int j0 = 4 * lane + 256 * e;
                float4 *d4 = reinterpret_cast<float4 *>(dly + j0);
                const float4 d = *d4;
                float o[4] = {d.x, d.y, d.z, d.w};
                if (j0 >= kP0) {  // dsp.rs:111-117
                    float ps[4];
                    pcm_short4(ldsf, j0 - kP0, sw, psw, ps);
#pragma unroll
                    for (int q = 0; q < 4; ++q) o[q] = o[q] + ps[q];
                }
                if (SYM_AAC_EMIT) st_stream(reinterpret_cast<float4 *>(frame_out + j0), make_float4(o[0], o[1], o[2], o[3]));
                float nd[4] = {0.0f, 0.0f, 0.0f, 0.0f};  // dsp.rs:138-145
                if (j0 < kP1) pcm_short4(ldsf, j0 + kP1, sw, psw, nd);
                *d4 = make_float4(nd[0], nd[1], nd[2], nd[3]);
            }
            wave_sync();
#pragma unroll
            for (int h = 0; h < 2; ++h) load_slot(dly, lane + 64 * h, dl[h]);
#if SYM_AAC_SINK
            touch_prefetch(line, sb_next);
#endif
            wave_sync();  // LDS is overwritten by the next frame
        }
#if SYM_AAC_VARIANT == 1
    };
    long t = t_first;
    if (t < (long)t_begin) do_frame(t++, std::false_type{});
#pragma unroll 1
    for (; t < (long)t_end; ++t) do_frame(t, std::true_type{});
#else
    }
#endif

#if SYM_AAC_CLOCK
    if (lane == 0) {
        const unsigned long long dc = __builtin_readcyclecounter() - clk_t0, dw = wall_clock64() - clk_w0;
        unsigned *o = reinterpret_cast<unsigned *>(pcm + (chain_base + (size_t)t_begin) * 1024);
        unsigned hw, xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        o[0] = 0x51a7c10cu;
        o[1] = (unsigned)dc;
        o[2] = (unsigned)dw;
        o[3] = (unsigned)(t_end - (unsigned)t_first);
        o[4] = (unsigned)(clk_w0 & 0xffffffffu);
        o[5] = hw;
        o[6] = xcc;
        o[7] = (unsigned)clk_wait;
    }
#endif
    if (t_end == frames_per_chain) {
        float *d = delay_out + (size_t)chain * 1024;
#pragma unroll
        for (int h = 0; h < 2; ++h) store_slot(d, lane + 64 * h, dl[h]);
    }
}

// the launch of the wavefront walk (aac.hip: launch_aac, when SYM_AAC_QUAD == 0)
inline int launch_aac_wave_walk(symaccel_ctx *ctx, const float *d_coeffs, const uint8_t *d_side, const float *d_delay_in, float *d_delay_out,
                                float *d_pcm, size_t n_chains, size_t frames_per_chain) {
    const unsigned seg = choose_segment(ctx, n_chains, frames_per_chain, 4 * SYM_AAC_MIN_WAVES, 1, 1, 1);
    const size_t segs = (frames_per_chain + seg - 1) / seg;
    const size_t items = n_chains * segs;
    const size_t grid = (items + kWaves - 1) / kWaves;
    if (items > 0xffffffffu || grid > 0x7fffffffu) return SYMACCEL_ERR_INVALID_ARG;
    void *sink = nullptr;
    SYM_TRY(ctx_sink(ctx, &sink));
    hipLaunchKernelGGL(aac_synth_kernel, dim3((unsigned)grid), dim3(64 * kWaves), 0, ctx->stream, ctx->dev, d_coeffs,
                       d_side, d_delay_in, d_delay_out, d_pcm, static_cast<float *>(sink), (unsigned)frames_per_chain, seg, (unsigned)segs,
                       (unsigned)items);
    SYM_GPU(ctx, hipGetLastError());
    return SYMACCEL_OK;
}
