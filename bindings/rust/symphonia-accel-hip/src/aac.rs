//! `HipAacDecoder`: AAC-LC with the synthesis tail (Dsp::synth, symphonia-codec-aac/src/aac/dsp.rs:57-158) on the MI355X.
use std::sync::{Arc, Mutex};

use symphonia_codec_aac::{AacDecoder, SynthBackend};
use symphonia_core::audio::{Audio, AudioBuffer, AudioMut, AudioSpec, GenericAudioBufferRef};
use symphonia_core::codecs::audio::well_known::CODEC_ID_AAC;
use symphonia_core::codecs::audio::{AudioCodecParameters, AudioDecoder, AudioDecoderOptions, FinalizeResult};
use symphonia_core::codecs::registry::{RegisterableAudioDecoder, SupportedAudioCodec};
use symphonia_core::codecs::CodecInfo;
use symphonia_core::errors::{decode_error, unsupported_error, Result};
use symphonia_core::packet::PacketRef;
use symphonia_core::support_audio_codec;

use crate::ctx::{check, Context, Pinned};
use crate::ffi;
use crate::lookahead::{BatchCodec, Lookahead};

/// One raw_data_block after the CPU front end: per channel the 1024 coefficients `Ics::synth_channel` would hand to
/// `Dsp::synth` (after pulse and TNS, ics/mod.rs:449-468) and the side byte SYMACCEL_AAC_SIDE(seq, shape, prev_shape).
pub struct ParsedAac {
    pub coeffs: Vec<f32>,
    pub side: Vec<u8>,
}

/// The CPU front end: the reference's `AacDecoder::decode_inner` up to (not including) `synth_audio`
/// (symphonia-codec-aac/src/aac/mod.rs:170-225), vendored because `mod aac` is private to its crate.
pub trait AacFrontEnd: Send + Sync {
    /// The stream's parameters as the reference's decoder amends them (sample rate and channels from the
    /// AudioSpecificConfig, aac/mod.rs:99-102).
    fn params(&self) -> &AudioCodecParameters;
    fn channels(&self) -> usize;
    fn parse(&mut self, packet: &PacketRef<'_>) -> Result<ParsedAac>;
}

/// What the reference's decoder hands its `SynthBackend` for one packet (bindings/rust/patches/symphonia-codec-aac.diff).
#[derive(Default)]
pub struct AacRecord {
    pub coeffs: Vec<f32>, // [channel][1024]
    pub side: Vec<u8>,    // [channel]
    pub seen: Vec<bool>,  // [channel]: the packet carried this channel
}

impl AacRecord {
    fn begin_packet(&mut self, nch: usize) {
        self.coeffs.clear();
        self.coeffs.resize(nch * 1024, 0.0);
        self.side.clear();
        self.side.resize(nch, 0);
        self.seen.clear();
        self.seen.resize(nch, false);
    }
}

/// The `SynthBackend` handed to the reference's `AacDecoder`: the coefficients `Dsp::synth` would transform (after
/// joint stereo, pulse data and TNS -- all of that is the reference's code) are recorded, nothing is synthesized.
pub struct Recorder(pub Arc<Mutex<AacRecord>>);

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
        if channel < rec.seen.len() {
            rec.coeffs[channel * 1024..(channel + 1) * 1024].copy_from_slice(coeffs);
            // SYMACCEL_AAC_SIDE(seq, shape, prev_shape), include/symaccel.h
            rec.side[channel] = (window_sequence & 3) | ((window_shape as u8) << 2) | ((prev_window_shape as u8) << 3);
            rec.seen[channel] = true;
        }
    }
}

/// `AacFrontEnd` over the reference's own decoder with the recording backend installed: element parsing, Huffman decoding,
/// dequantisation, M/S and intensity stereo, PNS, pulse data and TNS are symphonia-codec-aac's code, unmodified.
pub struct SeamFrontEnd {
    dec: AacDecoder,
    rec: Arc<Mutex<AacRecord>>,
    nch: usize,
}

impl SeamFrontEnd {
    pub fn try_new(params: &AudioCodecParameters, opts: &AudioDecoderOptions) -> Result<Self> {
        let rec: Arc<Mutex<AacRecord>> = Arc::new(Mutex::new(AacRecord::default()));
        let dec = AacDecoder::try_new_with_backend(params, opts, Box::new(Recorder(rec.clone())))?;
        // the decoder amends its parameters from the AudioSpecificConfig (aac/mod.rs:99-102)
        let Some(channels) = dec.codec_params().channels.clone() else {
            return unsupported_error("aac: channels are required");
        };
        Ok(SeamFrontEnd { dec, rec, nch: channels.count() })
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
        Ok(ParsedAac { coeffs: rec.coeffs.clone(), side: rec.side.clone() })
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
    buf: AudioBuffer<f32>,
}

impl BatchCodec for AacBatch {
    type Parsed = ParsedAac;

    fn parse(&mut self, packet: &PacketRef<'_>) -> Result<ParsedAac> {
        self.front.parse(packet)
    }

    fn transform(&mut self, batch: &[ParsedAac]) -> Result<()> {
        let k = batch.len();
        for (i, p) in batch.iter().enumerate() {
            for c in 0..self.nch {
                let dst = (c * k + i) * 1024;
                self.coeffs.as_mut_slice()[dst..dst + 1024].copy_from_slice(&p.coeffs[c * 1024..(c + 1) * 1024]);
                self.side[c * k + i] = p.side[c];
            }
        }
        self.batch_len = k;
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
        self.delay.fill(0.0);
    }

    fn clear(&mut self) {
        self.buf.clear();
    }
}

/// AAC-LC decoder with the same observable behaviour as `symphonia_codec_aac::AacDecoder`.
pub struct HipAacDecoder {
    params: AudioCodecParameters,
    batch: AacBatch,
    la: Lookahead<ParsedAac>,
}

impl HipAacDecoder {
    pub fn try_new(_params: &AudioCodecParameters, _opts: &AudioDecoderOptions, front: Box<dyn AacFrontEnd>, max_batch: usize) -> Result<Self> {
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
                buf: AudioBuffer::new(AudioSpec::new(rate, channels), 1024),
            },
            la: Lookahead::new(max_batch),
        })
    }
}

impl AudioDecoder for HipAacDecoder {
    fn reset(&mut self) {
        self.batch.reset_state();
        self.la.reset();
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
        // no front end, no device, no memory: the decoder that was registered below this one takes the track
        match crate::frontends::aac_front_end(params, opts).and_then(|front| HipAacDecoder::try_new(params, opts, front, crate::DEFAULT_LOOKAHEAD)) {
            Ok(decoder) => Ok(Box::new(decoder)),
            Err(e) => crate::fallback::make(params, opts, e),
        }
    }

    fn supported_codecs() -> &'static [SupportedAudioCodec] {
        &[support_audio_codec!(CODEC_ID_AAC, "aac", "Advanced Audio Coding (MI355X synthesis)")]
    }
}
