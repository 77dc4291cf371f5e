//! `HipVorbisDecoder`: the per-channel synthesis of every audio packet -- Imdct, windowing and the lapped overlap-add of
//! `DspChannel::synth` (symphonia-codec-vorbis/src/dsp.rs:68-145, called from lib.rs:296-331) -- on the MI355X.
use symphonia_core::audio::{Audio, AudioBuffer, AudioMut, AudioSpec, GenericAudioBufferRef};
use symphonia_core::codecs::audio::well_known::CODEC_ID_VORBIS;
use symphonia_core::codecs::audio::{AudioCodecParameters, AudioDecoderOptions};
use symphonia_core::errors::{unsupported_error, Result};
use symphonia_core::packet::PacketRef;
use symphonia_core::support_audio_codec;

use crate::ctx::{check, Context, Pinned};
use crate::decoder::DecoderBatch;
use crate::ffi;
use crate::lookahead::{BatchCodec, Lookahead};

/// One audio packet after the CPU front end (mode / window flags, floor decode, residue decode, inverse coupling and the
/// floor x residue product: lib.rs:186-292): per channel the n/2 spectral lines `DspChannel::synth` receives.
/// Channels whose floor is unused ("do not decode") carry zeros, as `synth` sees them (dsp.rs:72-75).
pub struct ParsedVorbis {
    pub long_block: bool,  // block_flag of the packet's mode (lib.rs:203-214)
    pub spectra: Vec<f32>, // [channel][n / 2], n = the block size the flag selects
}

pub trait VorbisFrontEnd: Send + Sync {
    fn channels(&self) -> usize;
    /// (bs0_exp, bs1_exp) of the identification header (lib.rs:404-406)
    fn block_exps(&self) -> (i32, i32);
    fn parse(&mut self, packet: &PacketRef<'_>) -> Result<ParsedVorbis>;
}

pub struct VorbisBatch {
    ctx: Context,
    front: Box<dyn VorbisFrontEnd>,
    nch: usize,
    bs: [usize; 2],           // block sizes of flag 0 / 1
    spectra: Pinned<f32>,     // [channel][packed lines of the batch]
    flags: Vec<u8>,           // [channel][packet] (the same flag for every channel of a packet)
    prev_flag: Vec<i32>,      // [channel]: -1 = no previous block (after reset), else the last block's flag
    overlap: Vec<f32>,        // [channel][bs1 / 2]
    pcm: Pinned<f32>,         // [channel][packed samples of the batch]
    spec_stride: usize,
    pcm_stride: usize,
    pcm_off: Vec<usize>,      // per packet of the batch: offset of its samples in a channel's PCM; one extra = total
    emits: Vec<bool>,         // per packet: false for the first block after a reset (dsp.rs:77-80: nothing to lap with)
    buf: AudioBuffer<f32>,
}

impl BatchCodec for VorbisBatch {
    type Parsed = ParsedVorbis;

    fn parse(&mut self, packet: &PacketRef<'_>) -> Result<ParsedVorbis> {
        self.front.parse(packet)
    }

    fn transform(&mut self, batch: &[ParsedVorbis]) -> Result<()> {
        let k = batch.len();
        // the packed layout of include/symaccel.h ("Vorbis"): block b owns n_b / 2 lines and (prev_n + n_b) / 4 samples;
        // a first block without a previous one owns n_b / 2 sample slots and leaves them untouched
        let mut prev: i32 = self.prev_flag[0];
        let (mut lines, mut samples) = (0usize, 0usize);
        let mut spec_off = Vec::with_capacity(k);
        self.pcm_off.clear();
        self.emits.clear();
        for p in batch {
            let n = self.bs[p.long_block as usize];
            spec_off.push(lines);
            self.pcm_off.push(samples);
            self.emits.push(prev >= 0);
            lines += n / 2;
            samples += if prev >= 0 { (self.bs[prev as usize] + n) / 4 } else { n / 2 };
            prev = p.long_block as i32;
        }
        self.pcm_off.push(samples);
        self.spec_stride = lines;
        self.pcm_stride = samples;
        for (i, p) in batch.iter().enumerate() {
            let half = self.bs[p.long_block as usize] / 2;
            for c in 0..self.nch {
                let dst = c * lines + spec_off[i];
                self.spectra.as_mut_slice()[dst..dst + half].copy_from_slice(&p.spectra[c * half..(c + 1) * half]);
                self.flags[c * k + i] = p.long_block as u8;
            }
        }
        let (bs0_exp, bs1_exp) = self.front.block_exps();
        // SAFETY: the buffers cover nch chains of `lines` / `samples` / `k` elements (sized for max_batch long blocks).
        check(
            unsafe {
                ffi::symaccel_vorbis_synth(
                    self.ctx.raw(),
                    bs0_exp,
                    bs1_exp,
                    self.spectra.as_slice().as_ptr(),
                    lines,
                    self.flags.as_ptr(),
                    self.prev_flag.as_mut_ptr(),
                    self.overlap.as_mut_ptr(),
                    self.pcm.as_mut_slice().as_mut_ptr(),
                    samples,
                    self.nch,
                    k,
                )
            },
            self.ctx.raw(),
        )
    }

    fn publish(&mut self, i: usize) {
        let frames = if self.emits[i] { self.pcm_off[i + 1] - self.pcm_off[i] } else { 0 };
        self.buf.clear();
        self.buf.render_uninit(Some(frames));
        for c in 0..self.nch {
            let src = c * self.pcm_stride + self.pcm_off[i];
            if let Some(plane) = self.buf.plane_mut(c) {
                plane[..frames].copy_from_slice(&self.pcm.as_slice()[src..src + frames]);
            }
        }
    }

    fn reset_state(&mut self) {
        // Dsp::reset (dsp.rs:45-56): lapping state empty, overlap zeroed
        self.prev_flag.fill(-1);
        self.overlap.fill(0.0);
    }

    fn clear(&mut self) {
        self.buf.clear();
    }
}

impl DecoderBatch for VorbisBatch {
    fn buffer(&self) -> GenericAudioBufferRef<'_> {
        self.buf.as_generic_audio_buffer_ref()
    }
}

crate::hip_decoder!(
    HipVorbisDecoder,
    VorbisBatch,
    ParsedVorbis,
    crate::frontends::vorbis_front_end,
    &[support_audio_codec!(CODEC_ID_VORBIS, "vorbis", "Vorbis (MI355X synthesis)")],
    "Vorbis decoder with the same observable behaviour as `symphonia_codec_vorbis::VorbisDecoder`."
);

impl HipVorbisDecoder {
    pub fn try_new(params: &AudioCodecParameters, _opts: &AudioDecoderOptions, front: Box<dyn VorbisFrontEnd>, max_batch: usize) -> Result<Self> {
        let (Some(rate), Some(channels)) = (params.sample_rate, params.channels.clone()) else {
            return unsupported_error("vorbis: sample rate and channels are required");
        };
        let nch = front.channels();
        let max_batch = max_batch.max(1);
        let (bs0_exp, bs1_exp) = front.block_exps();
        let bs = [1usize << bs0_exp, 1usize << bs1_exp];
        Ok(HipVorbisDecoder {
            params: params.clone(),
            batch: VorbisBatch {
                ctx: Context::new(0)?,
                front,
                nch,
                bs,
                spectra: Pinned::new(nch * max_batch * bs[1] / 2)?,
                flags: vec![0; nch * max_batch],
                prev_flag: vec![-1; nch],
                overlap: vec![0.0; nch * bs[1] / 2],
                pcm: Pinned::new(nch * max_batch * bs[1] / 2)?,
                spec_stride: 0,
                pcm_stride: 0,
                pcm_off: Vec::with_capacity(max_batch + 1),
                emits: Vec::with_capacity(max_batch),
                buf: AudioBuffer::new(AudioSpec::new(rate, channels), bs[1] / 2),
            },
            la: Lookahead::new(max_batch),
        })
    }
}
