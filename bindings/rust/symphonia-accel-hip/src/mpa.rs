//! `HipMpaDecoder`: MPEG-1/2/2.5 Layer III with the synthesis tail -- reorder, antialias, hybrid synthesis, frequency
//! inversion and the polyphase filterbank (symphonia-bundle-mp3/src/layer3/hybrid_synthesis.rs:153-485,
//! synthesis.rs:158-336; caller layer3/mod.rs:440-476) -- on the MI355X.
use std::sync::{Arc, Mutex};

use symphonia_bundle_mp3::backend::{GranuleSide, SynthBackend};
use symphonia_bundle_mp3::MpaDecoder;
use symphonia_core::audio::{Audio, AudioBuffer, AudioMut, AudioSpec, GenericAudioBufferRef};
use symphonia_core::codecs::audio::well_known::CODEC_ID_MP3;
use symphonia_core::codecs::audio::{AudioCodecParameters, AudioDecoder, AudioDecoderOptions};
use symphonia_core::errors::{decode_error, unsupported_error, Result};
use symphonia_core::packet::PacketRef;
use symphonia_core::support_audio_codec;

use crate::ctx::{check, Context, Pinned};
use crate::decoder::DecoderBatch;
use crate::ffi;
use crate::lookahead::{BatchCodec, Lookahead};

/// One frame after the CPU front end (`Layer3::decode` up to, not including, the per-channel tail of
/// layer3/mod.rs:440-476): per granule and channel the 576 samples after requantize + stereo processing and the three
/// `GranuleChannel` fields the tail reads.  `xr[granule][channel][576]`, `side[granule][channel]`.
pub struct ParsedMpa {
    pub trim: (usize, usize), // frames to trim from the start / end of the decoded packet when gapless (decoder.rs:128-131)
    pub n_granules: usize, // 2 for MPEG-1, 1 for MPEG-2 / 2.5
    pub xr: Vec<f32>,
    pub side: Vec<ffi::SymaccelMp3Side>,
}

/// The reference's bitstream reader, Huffman decoder, requantizer and stereo processor, vendored (they are private to
/// symphonia-bundle-mp3).  `sample_rate_idx` selects the scale-factor-band table `reorder` uses (0..8).
pub trait MpaFrontEnd: Send + Sync {
    fn channels(&self) -> usize;
    fn sample_rate_idx(&self) -> i32;
    fn parse(&mut self, packet: &PacketRef<'_>) -> Result<ParsedMpa>;
}

/// What the reference's decoder hands its `SynthBackend` for one packet (bindings/rust/patches/symphonia-bundle-mp3.diff).
#[derive(Default)]
pub struct MpaRecord {
    pub xr: Vec<f32>,                    // [granule][channel][576], in call order
    pub side: Vec<ffi::SymaccelMp3Side>, // [granule][channel]
    pub sample_rate_idx: i32,
    pub resets: usize,
}

/// The `SynthBackend` handed to the reference's `MpaDecoder`: the samples of every granule-channel after requantisation
/// and joint-stereo processing are recorded, nothing is synthesized.
pub struct Recorder(pub Arc<Mutex<MpaRecord>>);

impl SynthBackend for Recorder {
    fn synth_granule(&mut self, _channel: usize, side: &GranuleSide, samples: &mut [f32; 576], _out: &mut [f32]) {
        // (the decoder calls channel 0, channel 1 of granule 0, then of granule 1: layer3/mod.rs:421-477)
        let mut rec = self.0.lock().expect("mp3 record poisoned");
        rec.xr.extend_from_slice(&samples[..]);
        rec.side.push(ffi::SymaccelMp3Side { block_type: side.block_type, is_mixed: side.is_mixed as u8, rzero: side.rzero.min(576) as u16 });
        rec.sample_rate_idx = side.sample_rate_idx as i32;
    }

    fn reset(&mut self) {
        self.0.lock().expect("mp3 record poisoned").resets += 1;
    }
}

/// Index of `rate` in the order the reference's frame header uses (common.rs: 44100, 48000, 32000, 22050, 24000, 16000,
/// 11025, 12000, 8000).
fn sample_rate_index(rate: u32) -> Option<i32> {
    const RATES: [u32; 9] = [44_100, 48_000, 32_000, 22_050, 24_000, 16_000, 11_025, 12_000, 8_000];
    RATES.iter().position(|r| *r == rate).map(|i| i as i32)
}

/// `MpaFrontEnd` over the reference's own decoder with the recording backend installed: header, side info, bit reservoir,
/// scale factors, Huffman decoding, requantisation and joint stereo are symphonia-bundle-mp3's code, unmodified.
pub struct SeamFrontEnd {
    dec: MpaDecoder,
    rec: Arc<Mutex<MpaRecord>>,
    nch: usize,
    sr_idx: i32,
}

impl SeamFrontEnd {
    pub fn try_new(params: &AudioCodecParameters, _opts: &AudioDecoderOptions) -> Result<Self> {
        let (Some(rate), Some(channels)) = (params.sample_rate, params.channels.clone()) else {
            return unsupported_error("mp3: sample rate and channels are required");
        };
        let Some(sr_idx) = sample_rate_index(rate) else {
            return unsupported_error("mp3: not an MPEG audio sample rate");
        };
        let rec: Arc<Mutex<MpaRecord>> = Arc::new(Mutex::new(MpaRecord::default()));
        // the front end never trims: the trim of a packet is applied to what the device produced (MpaBatch::publish)
        let opts = AudioDecoderOptions { gapless: false, ..Default::default() };
        let dec = MpaDecoder::try_new_with_backend(params, &opts, Box::new(Recorder(rec.clone())))?;
        Ok(SeamFrontEnd { dec, rec, nch: channels.count(), sr_idx })
    }
}

impl MpaFrontEnd for SeamFrontEnd {
    fn channels(&self) -> usize {
        self.nch
    }

    fn sample_rate_idx(&self) -> i32 {
        self.sr_idx
    }

    fn parse(&mut self, packet: &PacketRef<'_>) -> Result<ParsedMpa> {
        {
            let mut rec = self.rec.lock().expect("mp3 record poisoned");
            rec.xr.clear();
            rec.side.clear();
        }
        self.dec.decode_ref(packet)?;
        let rec = self.rec.lock().expect("mp3 record poisoned");
        if rec.side.is_empty() || rec.side.len() % self.nch != 0 || rec.sample_rate_idx != self.sr_idx {
            // a mid-stream change of the channel mode or the sample rate: the reference fails such packets too
            // (decoder.rs:105-110, "invalid audio buffer signal spec for packet")
            return decode_error("mp3: the frame does not match the stream's channels or sample rate");
        }
        Ok(ParsedMpa {
            trim: (packet.trim_start.get() as usize, packet.trim_end.get() as usize),
            n_granules: rec.side.len() / self.nch,
            xr: rec.xr.clone(),
            side: rec.side.clone(),
        })
    }
}

pub struct MpaBatch {
    ctx: Context,
    front: Box<dyn MpaFrontEnd>,
    nch: usize,
    xr: Pinned<f32>,                  // [channel][granule of the batch][576]
    side: Vec<ffi::SymaccelMp3Side>,  // [channel][granule of the batch]
    overlap: Vec<f32>,                // [channel][32][18]
    vvec: Vec<f32>,                   // [channel][16][64]
    vfront: Vec<i32>,                 // [channel]
    pcm: Pinned<f32>,                 // [channel][granule of the batch][576]
    first_granule: Vec<usize>,        // per packet of the batch: index of its first granule; one extra entry = total
    trims: Vec<(usize, usize)>,       // per packet of the batch
    gapless: bool,
    buf: AudioBuffer<f32>,
}

impl BatchCodec for MpaBatch {
    type Parsed = ParsedMpa;

    fn parse(&mut self, packet: &PacketRef<'_>) -> Result<ParsedMpa> {
        self.front.parse(packet)
    }

    fn transform(&mut self, batch: &[ParsedMpa]) -> Result<()> {
        self.first_granule.clear();
        self.trims.clear();
        let mut total = 0usize;
        for p in batch {
            self.first_granule.push(total);
            self.trims.push(p.trim);
            total += p.n_granules;
        }
        self.first_granule.push(total);
        for (i, p) in batch.iter().enumerate() {
            for g in 0..p.n_granules {
                for c in 0..self.nch {
                    let dst = c * total + self.first_granule[i] + g;
                    let src = (g * self.nch + c) * 576;
                    self.xr.as_mut_slice()[dst * 576..(dst + 1) * 576].copy_from_slice(&p.xr[src..src + 576]);
                    self.side[dst] = p.side[g * self.nch + c];
                }
            }
        }
        // SAFETY: every buffer covers nch * total (* 576) elements (sized for max_batch frames of two granules); the
        // call returns after the PCM and the updated state are back in host memory.
        check(
            unsafe {
                ffi::symaccel_mp3_synth(
                    self.ctx.raw(),
                    self.xr.as_slice().as_ptr(),
                    self.side.as_ptr(),
                    self.front.sample_rate_idx(),
                    self.overlap.as_mut_ptr(),
                    self.vvec.as_mut_ptr(),
                    self.vfront.as_mut_ptr(),
                    self.pcm.as_mut_slice().as_mut_ptr(),
                    self.nch,
                    total,
                )
            },
            self.ctx.raw(),
        )
    }

    fn publish(&mut self, i: usize) {
        let total = *self.first_granule.last().unwrap_or(&0);
        let (g0, g1) = (self.first_granule[i], self.first_granule[i + 1]);
        let frames = (g1 - g0) * 576;
        self.buf.clear();
        self.buf.render_uninit(Some(frames));
        for c in 0..self.nch {
            let src = (c * total + g0) * 576;
            if let Some(plane) = self.buf.plane_mut(c) {
                plane[..frames].copy_from_slice(&self.pcm.as_slice()[src..src + frames]);
            }
        }
        if self.gapless {
            // decoder.rs:128-131
            self.buf.trim(self.trims[i].0, self.trims[i].1);
        }
    }

    fn reset_state(&mut self) {
        // Layer3::reset + synthesis state (layer3/mod.rs, synthesis.rs:140-156): overlap, V vector and its front index
        self.overlap.fill(0.0);
        self.vvec.fill(0.0);
        self.vfront.fill(0);
    }

    fn clear(&mut self) {
        self.buf.clear();
    }
}

impl DecoderBatch for MpaBatch {
    fn buffer(&self) -> GenericAudioBufferRef<'_> {
        self.buf.as_generic_audio_buffer_ref()
    }
}

crate::hip_decoder!(
    HipMpaDecoder,
    MpaBatch,
    ParsedMpa,
    crate::frontends::mpa_front_end,
    &[support_audio_codec!(CODEC_ID_MP3, "mp3", "MPEG Audio Layer 3 (MI355X synthesis)")],
    "MP3 decoder with the same observable behaviour as `symphonia_bundle_mp3::MpaDecoder` for Layer III streams."
);

impl HipMpaDecoder {
    pub fn try_new(params: &AudioCodecParameters, opts: &AudioDecoderOptions, front: Box<dyn MpaFrontEnd>, max_batch: usize) -> Result<Self> {
        let (Some(rate), Some(channels)) = (params.sample_rate, params.channels.clone()) else {
            return unsupported_error("mp3: sample rate and channels are required");
        };
        let nch = front.channels();
        let max_batch = max_batch.max(1);
        let granules = 2 * max_batch;
        Ok(HipMpaDecoder {
            params: params.clone(),
            batch: MpaBatch {
                ctx: Context::new(0)?,
                front,
                nch,
                xr: Pinned::new(nch * granules * 576)?,
                side: vec![ffi::SymaccelMp3Side { block_type: 0, is_mixed: 0, rzero: 0 }; nch * granules],
                overlap: vec![0.0; nch * 576],
                vvec: vec![0.0; nch * 1024],
                vfront: vec![0; nch],
                pcm: Pinned::new(nch * granules * 576)?,
                first_granule: Vec::with_capacity(max_batch + 1),
                trims: Vec::with_capacity(max_batch),
                gapless: opts.gapless,
                buf: AudioBuffer::new(AudioSpec::new(rate, channels), 1152),
            },
            la: Lookahead::new(max_batch),
        })
    }
}
