//! `HipAacDecoder`: AAC-LC with everything behind the spectrum decoder on the MI355X -- the joint-stereo decoding of channel pairs
//! (symphonia-codec-aac/src/aac/cpe.rs:110-157), the TNS filters (aac/ics/tns.rs:149-199) and the synthesis tail (Dsp::synth,
//! aac/dsp.rs:57-158; caller ics/mod.rs:449-468).  The front end hands over the coefficients as decoded + a joint-stereo
//! descriptor per channel-pair frame + the TNS filters as a list (`symaccel_aac_decode_pipelined`); a front end that delivers
//! finished coefficients (`ParsedAac::fused == None`) takes `symaccel_aac_synth`.
use std::sync::{Arc, Mutex};

use symphonia_codec_aac::{AacDecoder, CodedChannel, JointStereo, SynthBackend, JOINT_STEREO_INTENSITY, JOINT_STEREO_MID_SIDE};
use symphonia_core::audio::{Audio, AudioBuffer, AudioMut, AudioSpec, GenericAudioBufferRef};
use symphonia_core::codecs::audio::well_known::CODEC_ID_AAC;
use symphonia_core::codecs::audio::{AudioCodecParameters, AudioDecoder, AudioDecoderOptions, FinalizeResult};
use symphonia_core::codecs::registry::{RegisterableAudioDecoder, SupportedAudioCodec};
use symphonia_core::codecs::CodecInfo;
use symphonia_core::errors::{decode_error, unsupported_error, Result};
use symphonia_core::packet::PacketRef;
use symphonia_core::support_audio_codec;

use crate::ctx::{check, Context, Pinned, Pool};
use crate::ffi;
use crate::lookahead::{BatchCodec, Lookahead};

/// One raw_data_block after the CPU front end: per channel the 1024 coefficients `Ics::synth_channel` would hand to
/// `Dsp::synth` (after pulse and TNS, ics/mod.rs:449-468) and the side byte SYMACCEL_AAC_SIDE(seq, shape, prev_shape).
pub struct ParsedAac {
    pub coeffs: Vec<f32>,
    pub side: Vec<u8>,
    pub fused: Option<FusedAac>,
}

/// What is still to be done to `ParsedAac::coeffs` when they are the SPECTRUM decoder's output (after pulse data, in front of joint
/// stereo and TNS: `ChannelPair::decode_ga_cpe` stopping at cpe.rs:109, `Ics::synth_channel` at ics/mod.rs:456).
pub struct FusedAac {
    /// the jointly coded channel pairs of the packet: (plane of the left channel, the pair's descriptor)
    pub joint: Vec<(usize, ffi::SymaccelAacJsFrame)>,
    /// the TNS filters of the packet's channels; `frame` = the channel's plane
    pub tns: Vec<ffi::SymaccelAacTnsFilter>,
    /// scale factor band offsets of the stream's sample rate, where a descriptor of this packet uses them (else empty)
    pub swb_long: Vec<u16>,
    pub swb_short: Vec<u16>,
}

/// The CPU front end: the reference's `AacDecoder::decode_inner` up to (not including) `synth_audio`
/// (symphonia-codec-aac/src/aac/mod.rs:170-225), vendored because `mod aac` is private to its crate.
pub trait AacFrontEnd: Send + Sync {
    /// The stream's parameters as the reference's decoder amends them (sample rate and channels from the
    /// AudioSpecificConfig, aac/mod.rs:99-102).
    fn params(&self) -> &AudioCodecParameters;
    fn channels(&self) -> usize;
    fn parse(&mut self, packet: &PacketRef<'_>) -> Result<ParsedAac>;
    /// `AudioDecoder::reset` (aac/mod.rs:252-256): the parse stage forgets the window shape of the previous frame.
    fn reset(&mut self);
}

/// What the reference's decoder hands its `SynthBackend` for one packet (bindings/rust/patches/symphonia-codec-aac.diff).
#[derive(Default)]
pub struct AacRecord {
    pub coeffs: Vec<f32>, // [channel][1024]
    pub side: Vec<u8>,    // [channel]
    pub seen: Vec<bool>,  // [channel]: the packet carried this channel
    pub joint: Vec<(usize, ffi::SymaccelAacJsFrame)>, // the fused form (synth_element): see FusedAac
    pub tns: Vec<ffi::SymaccelAacTnsFilter>,
    pub swb_long: Vec<u16>,
    pub swb_short: Vec<u16>,
}

impl AacRecord {
    fn begin_packet(&mut self, nch: usize) {
        self.coeffs.clear();
        self.coeffs.resize(nch * 1024, 0.0);
        self.side.clear();
        self.side.resize(nch, 0);
        self.seen.clear();
        self.seen.resize(nch, false);
        self.joint.clear();
        self.tns.clear();
        self.swb_long.clear();
        self.swb_short.clear();
    }

    fn channel(&mut self, channel: usize, coeffs: &[f32; 1024], window_sequence: u8, window_shape: bool, prev_window_shape: bool) {
        if channel < self.seen.len() {
            self.coeffs[channel * 1024..(channel + 1) * 1024].copy_from_slice(coeffs);
            // SYMACCEL_AAC_SIDE(seq, shape, prev_shape), include/symaccel.h
            self.side[channel] = (window_sequence & 3) | ((window_shape as u8) << 2) | ((prev_window_shape as u8) << 3);
            self.seen[channel] = true;
        }
    }
}

/// The `SynthBackend` handed to the reference's `AacDecoder`: the coefficients `Dsp::synth` would transform (after
/// joint stereo, pulse data and TNS -- all of that is the reference's code) are recorded, nothing is synthesized.
pub struct Recorder(pub Arc<Mutex<AacRecord>>, pub bool);

impl SynthBackend for Recorder {
    fn synth(
        &mut self,
        channel: usize,
        coeffs: &[f32; 1024],
        _delay: &mut [f32; 1024],
        window_sequence: u8,
        window_shape: bool,
        prev_window_shape: bool,
        _dst: &mut [f32],
    ) {
        let mut rec = self.0.lock().expect("aac record poisoned");
        rec.channel(channel, coeffs, window_sequence, window_shape, prev_window_shape);
    }

    /// `Recorder(_, true)`: take the elements in front of joint stereo and TNS (the second seam of the patch).
    fn takes_elements(&self) -> bool {
        self.1
    }

    fn synth_element(&mut self, joint: Option<&JointStereo>, channels: &mut [CodedChannel<'_>], _abuf: &mut AudioBuffer<f32>) {
        let mut rec = self.0.lock().expect("aac record poisoned");
        for ch in channels.iter() {
            rec.channel(ch.channel, ch.coeffs, ch.window_sequence, ch.window_shape, ch.prev_window_shape);
            for f in ch.tns.iter() {
                // symaccel_aac_tns_filter: the line range and the taps Tns::synth would use (ics/tns.rs:149-199)
                rec.tns.push(ffi::SymaccelAacTnsFilter {
                    frame: ch.channel as u32,
                    start: f.start as u16,
                    end: f.end as u16,
                    order: f.order as u8,
                    direction: f.direction as u8,
                    pad: [0; 2],
                    lpc: f.coef,
                });
            }
        }
        if let (Some(js), Some(left)) = (joint, channels.first()) {
            let mut d = ffi::SymaccelAacJsFrame { num_windows: js.num_windows as u8, max_sfb: js.max_sfb as u8, pad: [0; 2], mode: [0; 128], scale: [0.0; 128] };
            for i in 0..128 {
                d.mode[i] = match js.mode[i] {
                    JOINT_STEREO_MID_SIDE => ffi::SYMACCEL_AAC_JS_MS as u8,
                    JOINT_STEREO_INTENSITY => ffi::SYMACCEL_AAC_JS_INTENSITY as u8,
                    _ => 0,
                };
                d.scale[i] = js.scale[i];
            }
            rec.joint.push((left.channel, d));
            let swb: Vec<u16> = js.bands.iter().map(|b| *b as u16).collect();
            if js.num_windows == 1 {
                rec.swb_long = swb;
            }
            else {
                rec.swb_short = swb;
            }
        }
    }
}

/// `AacFrontEnd` over the reference's own decoder with the recording backend installed: element parsing, Huffman decoding,
/// dequantisation, PNS and pulse data are symphonia-codec-aac's code, unmodified.  `fused` (the default) stops there -- the
/// recorder takes the elements in front of joint stereo and TNS and `parse` returns the coefficients as decoded + descriptors
/// (`ParsedAac::fused`); without it the reference's M/S, intensity stereo and TNS run too.
pub struct SeamFrontEnd {
    dec: AacDecoder,
    rec: Arc<Mutex<AacRecord>>,
    nch: usize,
    fused: bool,
}

impl SeamFrontEnd {
    pub fn try_new(params: &AudioCodecParameters, opts: &AudioDecoderOptions) -> Result<Self> {
        Self::try_new_at(params, opts, true)
    }

    /// `fused == false`: the first-generation seam (behind joint stereo, pulse data and TNS).
    pub fn try_new_at(params: &AudioCodecParameters, opts: &AudioDecoderOptions, fused: bool) -> Result<Self> {
        let rec: Arc<Mutex<AacRecord>> = Arc::new(Mutex::new(AacRecord::default()));
        let dec = AacDecoder::try_new_with_backend(params, opts, Box::new(Recorder(rec.clone(), fused)))?;
        // the decoder amends its parameters from the AudioSpecificConfig (aac/mod.rs:99-102)
        let Some(channels) = dec.codec_params().channels.clone() else {
            return unsupported_error("aac: channels are required");
        };
        Ok(SeamFrontEnd { dec, rec, nch: channels.count(), fused })
    }
}

impl AacFrontEnd for SeamFrontEnd {
    fn params(&self) -> &AudioCodecParameters {
        self.dec.codec_params()
    }

    fn channels(&self) -> usize {
        self.nch
    }

    fn parse(&mut self, packet: &PacketRef<'_>) -> Result<ParsedAac> {
        self.rec.lock().expect("aac record poisoned").begin_packet(self.nch);
        self.dec.decode_ref(packet)?;
        let rec = self.rec.lock().expect("aac record poisoned");
        if rec.seen.iter().any(|s| !*s) {
            // the reference would have left the missing channel's plane as it was rendered: not a stream this path takes
            return decode_error("aac: the packet does not carry every channel of the stream");
        }
        let fused = if self.fused {
            Some(FusedAac { joint: rec.joint.clone(), tns: rec.tns.clone(), swb_long: rec.swb_long.clone(), swb_short: rec.swb_short.clone() })
        }
        else {
            None
        };
        Ok(ParsedAac { coeffs: rec.coeffs.clone(), side: rec.side.clone(), fused })
    }

    fn reset(&mut self) {
        // AacDecoder::reset (aac/mod.rs:252-256): every pair's IcsInfo (the previous window shape) and delay line
        self.dec.reset();
    }
}

struct AacBatch {
    ctx: Context,
    front: Box<dyn AacFrontEnd>,
    nch: usize,
    max_batch: usize,
    coeffs: Pinned<f32>, // [channel][packet][1024]
    side: Vec<u8>,       // [channel][packet]
    delay: Vec<f32>,     // [channel][1024]: the delay lines between batches
    pcm: Pinned<f32>,    // [channel][packet][1024]
    batch_len: usize,
    // the fused form: scale factor band offsets of the stream (from the first descriptor that carries them; until then a table
    // that passes the library's checks and that no descriptor refers to)
    swb_long: Vec<u16>,
    swb_short: Vec<u16>,
    // the cross-stream batcher: the batch submitted ahead (its PCM and delay lines land in `pcm` / `delay` at collect)
    pool: Option<Arc<Pool>>,
    ticket: Option<u64>,
    next_batch_len: usize,
    buf: AudioBuffer<f32>,
}

impl BatchCodec for AacBatch {
    type Parsed = ParsedAac;

    fn parse(&mut self, packet: &PacketRef<'_>) -> Result<ParsedAac> {
        self.front.parse(packet)
    }

    fn transform(&mut self, batch: &[ParsedAac]) -> Result<()> {
        let k = batch.len();
        self.gather(batch);
        self.batch_len = k;
        if k > 0 && batch.iter().all(|p| p.fused.is_some()) {
            return self.transform_fused(batch);
        }
        // SAFETY: all pointers cover nch * k (* 1024) elements; the call returns after the PCM is in `pcm`.
        check(
            unsafe {
                ffi::symaccel_aac_synth(
                    self.ctx.raw(),
                    self.coeffs.as_slice().as_ptr(),
                    self.side.as_ptr(),
                    self.delay.as_mut_ptr(),
                    self.pcm.as_mut_slice().as_mut_ptr(),
                    self.nch,
                    k,
                )
            },
            self.ctx.raw(),
        )
    }

    fn publish(&mut self, i: usize) {
        self.buf.clear();
        self.buf.render_uninit(Some(1024));
        for c in 0..self.nch {
            let src = (c * self.batch_len + i) * 1024;
            if let Some(plane) = self.buf.plane_mut(c) {
                plane[..1024].copy_from_slice(&self.pcm.as_slice()[src..src + 1024]);
            }
        }
    }

    fn reset_state(&mut self) {
        self.front.reset();
        self.delay.fill(0.0);
    }

    fn clear(&mut self) {
        self.buf.clear();
    }

    fn pooled(&self) -> bool {
        self.pool.is_some()
    }

    /// `symaccel_batcher_submit_aac_decode`: the stream's next batch -- coded spectra, joint-stereo descriptors, TNS filters -- goes
    /// to the process-wide batcher; the inputs are copied before the call returns, PCM and delay lines are written by `collect`.
    fn submit(&mut self, batch: &[ParsedAac]) -> Result<()> {
        let Some(pool) = self.pool.clone() else {
            return unsupported_error("aac: no batcher");
        };
        if batch.is_empty() || !batch.iter().all(|p| p.fused.is_some()) || self.ticket.is_some() {
            return unsupported_error("aac: the batcher takes the spectrum decoder's output, one batch at a time");
        }
        let k = batch.len();
        self.gather(batch);
        let (pair_chains, desc, tns) = self.describe(batch);
        let n_pairs = pair_chains.len() / 2;
        let mut bands: i32 = -1;
        // SAFETY: the tables hold one entry more than the count passed; `bands` is a valid out-pointer.
        check(
            unsafe {
                ffi::symaccel_batcher_aac_bands(
                    pool.raw(),
                    self.swb_long.as_ptr(),
                    (self.swb_long.len() - 1) as i32,
                    self.swb_short.as_ptr(),
                    (self.swb_short.len() - 1) as i32,
                    &mut bands,
                )
            },
            pool.ctx_raw(),
        )?;
        let mut ticket = 0u64;
        // SAFETY: coeffs / side cover nch * k (* 1024) elements, desc n_pairs * k records; they are copied before the call returns.
        // `delay` and `pcm` are fields of self and stay where they are until `collect` / `abandon`.
        check(
            unsafe {
                ffi::symaccel_batcher_submit_aac_decode(
                    pool.raw(),
                    bands,
                    self.coeffs.as_slice().as_ptr(),
                    self.side.as_ptr(),
                    if n_pairs > 0 { pair_chains.as_ptr() } else { std::ptr::null() },
                    if n_pairs > 0 { desc.as_ptr() } else { std::ptr::null() },
                    n_pairs,
                    if tns.is_empty() { std::ptr::null() } else { tns.as_ptr() },
                    tns.len(),
                    self.delay.as_mut_ptr(),
                    self.pcm.as_mut_slice().as_mut_ptr(),
                    self.nch,
                    k,
                    &mut ticket,
                )
            },
            pool.ctx_raw(),
        )?;
        self.ticket = Some(ticket);
        self.next_batch_len = k;
        Ok(())
    }

    fn collect(&mut self) -> Result<()> {
        let (Some(pool), Some(ticket)) = (self.pool.clone(), self.ticket.take()) else {
            return unsupported_error("aac: nothing was submitted");
        };
        // SAFETY: a live ticket of this pool's batcher; the pointers given to submit are fields of self.
        check(unsafe { ffi::symaccel_batcher_collect(pool.raw(), ticket) }, pool.ctx_raw())?;
        self.batch_len = self.next_batch_len;
        Ok(())
    }

    fn hint(&mut self) {
        if let Some(pool) = &self.pool {
            // SAFETY: a live batcher.
            unsafe { ffi::symaccel_batcher_hint(pool.raw()) };
        }
    }

    fn abandon(&mut self) {
        if let (Some(pool), Some(ticket)) = (self.pool.clone(), self.ticket.take()) {
            // SAFETY: a live ticket; nothing is written to the delay lines or the PCM.
            unsafe { ffi::symaccel_batcher_abandon(pool.raw(), ticket) };
        }
    }
}

impl Drop for AacBatch {
    fn drop(&mut self) {
        BatchCodec::abandon(self); // (a batch still with the batcher points at this struct's buffers)
    }
}

impl AacBatch {
    /// The batch's coefficients and side bytes in the chain-major staging layout ([channel][packet of the batch]).
    fn gather(&mut self, batch: &[ParsedAac]) {
        let k = batch.len();
        for (i, p) in batch.iter().enumerate() {
            for c in 0..self.nch {
                let dst = (c * k + i) * 1024;
                self.coeffs.as_mut_slice()[dst..dst + 1024].copy_from_slice(&p.coeffs[c * 1024..(c + 1) * 1024]);
                self.side[c * k + i] = p.side[c];
            }
        }
    }

    /// cpe.rs:110-157 + ics/mod.rs:456-468 for the whole batch in one call: joint stereo, TNS and Dsp::synth on the device from
    /// the spectrum decoder's coefficients (already staged in `coeffs` / `side` by `transform`).
    /// What the fused entry points take beside the spectra: the channel pairs that are jointly coded anywhere in the batch, their
    /// descriptors ([pair][packet]) and the TNS filters of the batch (frame = channel * k + packet).
    fn describe(&mut self, batch: &[ParsedAac]) -> (Vec<i32>, Vec<ffi::SymaccelAacJsFrame>, Vec<ffi::SymaccelAacTnsFilter>) {
        let k = batch.len();
        // the channel pairs that are jointly coded anywhere in the batch (a stream's element layout is fixed: aac/mod.rs:121-129)
        let mut lefts: Vec<usize> = Vec::new();
        for p in batch.iter() {
            let Some(f) = &p.fused else { continue };
            for (left, _) in f.joint.iter() {
                if !lefts.contains(left) && *left + 1 < self.nch {
                    lefts.push(*left);
                }
            }
            if !f.swb_long.is_empty() {
                self.swb_long = f.swb_long.clone();
            }
            if !f.swb_short.is_empty() {
                self.swb_short = f.swb_short.clone();
            }
        }
        lefts.sort();
        let n_pairs = lefts.len();
        let mut pair_chains: Vec<i32> = Vec::with_capacity(2 * n_pairs);
        for left in lefts.iter() {
            pair_chains.push(*left as i32);
            pair_chains.push(*left as i32 + 1);
        }
        let none = ffi::SymaccelAacJsFrame { num_windows: 1, max_sfb: 0, pad: [0; 2], mode: [0; 128], scale: [0.0; 128] };
        let mut desc = vec![none; n_pairs * k];
        let mut tns: Vec<ffi::SymaccelAacTnsFilter> = Vec::new();
        for (i, p) in batch.iter().enumerate() {
            let Some(f) = &p.fused else { continue };
            for (left, d) in f.joint.iter() {
                if let Some(pair) = lefts.iter().position(|l| l == left) {
                    desc[pair * k + i] = *d;
                }
            }
            for t in f.tns.iter() {
                let mut t = *t;
                t.frame = (t.frame as usize * k + i) as u32; // [channel][packet of the batch]
                tns.push(t);
            }
        }
        (pair_chains, desc, tns)
    }

    /// cpe.rs:110-157 + ics/mod.rs:456-468 for the whole batch in one call (the coefficients are already staged by `transform`).
    fn transform_fused(&mut self, batch: &[ParsedAac]) -> Result<()> {
        let k = batch.len();
        let (pair_chains, desc, tns) = self.describe(batch);
        let n_pairs = pair_chains.len() / 2;
        // SAFETY: coeffs / side / pcm cover nch * k (* 1024) elements, delay nch * 1024, desc n_pairs * k records, the swb tables
        // one entry more than the count passed; null is passed for an empty list.  The call returns after the PCM is in `pcm`.
        check(
            unsafe {
                ffi::symaccel_aac_decode_pipelined(
                    self.ctx.raw(),
                    self.coeffs.as_slice().as_ptr(),
                    self.side.as_ptr(),
                    if n_pairs > 0 { pair_chains.as_ptr() } else { std::ptr::null() },
                    if n_pairs > 0 { desc.as_ptr() } else { std::ptr::null() },
                    n_pairs,
                    self.swb_long.as_ptr(),
                    (self.swb_long.len() - 1) as i32,
                    self.swb_short.as_ptr(),
                    (self.swb_short.len() - 1) as i32,
                    if tns.is_empty() { std::ptr::null() } else { tns.as_ptr() },
                    tns.len(),
                    self.delay.as_mut_ptr(),
                    self.pcm.as_mut_slice().as_mut_ptr(),
                    self.nch,
                    k,
                    0,
                )
            },
            self.ctx.raw(),
        )
    }
}

/// AAC-LC decoder with the same observable behaviour as `symphonia_codec_aac::AacDecoder`.
pub struct HipAacDecoder {
    params: AudioCodecParameters,
    batch: AacBatch,
    la: Lookahead<ParsedAac>,
}

impl HipAacDecoder {
    pub fn try_new(params: &AudioCodecParameters, opts: &AudioDecoderOptions, front: Box<dyn AacFrontEnd>, max_batch: usize) -> Result<Self> {
        Self::try_new_with_pool(params, opts, front, max_batch, None)
    }

    /// The same decoder submitting to the process-wide cross-stream batcher (`Pool::shared()`, `SYMACCEL_BATCH_AAC_DECODE`): with
    /// many streams open, joint stereo, TNS and synthesis of all of them run in one launch (csrc/batcher.cpp).
    pub fn try_new_pooled(params: &AudioCodecParameters, opts: &AudioDecoderOptions, front: Box<dyn AacFrontEnd>, max_batch: usize) -> Result<Self> {
        Self::try_new_with_pool(params, opts, front, max_batch, Some(Pool::shared()?))
    }

    pub fn try_new_with_pool(
        _params: &AudioCodecParameters,
        _opts: &AudioDecoderOptions,
        front: Box<dyn AacFrontEnd>,
        max_batch: usize,
        pool: Option<Arc<Pool>>,
    ) -> Result<Self> {
        let params = front.params().clone();
        let (Some(rate), Some(channels)) = (params.sample_rate, params.channels.clone()) else {
            return unsupported_error("aac: sample rate and channels are required");
        };
        let nch = front.channels();
        let max_batch = max_batch.max(1);
        Ok(HipAacDecoder {
            params,
            batch: AacBatch {
                ctx: Context::new(0)?,
                front,
                nch,
                max_batch,
                coeffs: Pinned::new(nch * max_batch * 1024)?,
                side: vec![0; nch * max_batch],
                delay: vec![0.0; nch * 1024],
                pcm: Pinned::new(nch * max_batch * 1024)?,
                batch_len: 0,
                swb_long: vec![0, 1024],
                swb_short: vec![0, 128],
                pool,
                ticket: None,
                next_batch_len: 0,
                buf: AudioBuffer::new(AudioSpec::new(rate, channels), 1024),
            },
            la: Lookahead::new(max_batch),
        })
    }
}

impl AudioDecoder for HipAacDecoder {
    fn reset(&mut self) {
        self.la.reset_with(&mut self.batch); // (a batch still with the cross-stream batcher is given up first)
        self.batch.reset_state();
    }

    fn codec_info(&self) -> &CodecInfo {
        &Self::supported_codecs()[0].info
    }

    fn codec_params(&self) -> &AudioCodecParameters {
        &self.params
    }

    fn decode_ref(&mut self, packet: &PacketRef<'_>) -> Result<GenericAudioBufferRef<'_>> {
        // (Lookahead::decode clears the buffer on every error path: codecs/audio.rs:278)
        self.la.decode(&mut self.batch, packet)?;
        Ok(self.batch.buf.as_generic_audio_buffer_ref())
    }

    fn finalize(&mut self) -> FinalizeResult {
        Default::default()
    }

    fn last_decoded(&self) -> GenericAudioBufferRef<'_> {
        self.batch.buf.as_generic_audio_buffer_ref()
    }
}

impl RegisterableAudioDecoder for HipAacDecoder {
    fn try_registry_new(params: &AudioCodecParameters, opts: &AudioDecoderOptions) -> Result<Box<dyn AudioDecoder>> {
        // The registry builds every decoder from (params, opts) alone (codecs/registry.rs:330-341): the decoders it builds find each
        // other in the process-wide `Pool` -- their look-ahead batches go to the device in common launches (csrc/batcher.cpp).  Only
        // when the pool cannot be created does a decoder batch on its own.
        // No front end, no device, no memory: the decoder that was registered below this one takes the track.
        let built = crate::frontends::aac_front_end(params, opts).and_then(|front| match Pool::shared() {
            Ok(pool) => HipAacDecoder::try_new_with_pool(params, opts, front, crate::DEFAULT_LOOKAHEAD, Some(pool)),
            Err(_) => HipAacDecoder::try_new(params, opts, front, crate::DEFAULT_LOOKAHEAD),
        });
        match built {
            Ok(decoder) => Ok(Box::new(decoder)),
            Err(e) => crate::fallback::make(params, opts, e),
        }
    }

    fn supported_codecs() -> &'static [SupportedAudioCodec] {
        &[support_audio_codec!(CODEC_ID_AAC, "aac", "Advanced Audio Coding (MI355X synthesis)")]
    }
}
