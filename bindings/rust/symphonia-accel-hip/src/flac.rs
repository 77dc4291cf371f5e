//! `HipFlacDecoder`: the predictor stage of every subframe (fixed and LPC, symphonia-bundle-flac/src/decoder.rs:663-752),
//! the stereo decorrelation (:32-82) and the left-justification shift (:239-242) on the MI355X.  FLAC carries no state
//! from frame to frame, so a batch is simply many frames' subframes side by side.
use symphonia_core::audio::{Audio, AudioBuffer, AudioMut, AudioSpec, GenericAudioBufferRef};
use symphonia_core::codecs::audio::well_known::CODEC_ID_FLAC;
use symphonia_core::codecs::audio::{AudioCodecParameters, AudioDecoderOptions};
use symphonia_core::errors::{decode_error, unsupported_error, Result};
use symphonia_core::packet::PacketRef;
use symphonia_core::support_audio_codec;

use crate::ctx::{check, Context, Pinned};
use crate::decoder::DecoderBatch;
use crate::ffi;
use crate::lookahead::{BatchCodec, Lookahead};

/// One frame after the CPU front end (frame header, subframe headers, Rice-decoded residuals: decoder.rs:381-660):
/// per channel `blocksize` words -- the warm-up samples followed by the residuals, exactly what `fixed_predict` /
/// `lpc_predict` receive --, the subframe's descriptor and quantised coefficients, and the frame's channel assignment.
pub struct ParsedFlac {
    pub blocksize: usize,
    pub words: Vec<i32>,                 // [channel][blocksize]
    pub desc: Vec<ffi::SymaccelFlacDesc>, // [channel]
    pub coeffs: Vec<i32>,                // [channel][32], reference order (decoder.rs:716-752)
    pub pair_mode: u8,                   // 0 independent, 1 left/side, 2 mid/side, 3 right/side (stereo frames only)
    pub out_shift: u32,                  // 32 - bits per sample (decoder.rs:239-242)
}

pub trait FlacFrontEnd: Send + Sync {
    fn channels(&self) -> usize;
    fn max_blocksize(&self) -> usize;
    fn parse(&mut self, packet: &PacketRef<'_>) -> Result<ParsedFlac>;
}

pub struct FlacBatch {
    ctx: Context,
    front: Box<dyn FlacFrontEnd>,
    nch: usize,
    stride: usize,                      // words per subframe slot of the current batch (its largest block size)
    words: Pinned<i32>,                 // [packet][channel][stride]
    desc: Vec<ffi::SymaccelFlacDesc>,   // [packet][channel]
    coeffs: Vec<i32>,                   // [packet][channel][32]
    lens: Vec<usize>,                   // block size of each packet of the batch
    modes: Vec<u8>,
    shifts: Vec<u32>,
    buf: AudioBuffer<i32>,
}

impl BatchCodec for FlacBatch {
    type Parsed = ParsedFlac;

    fn parse(&mut self, packet: &PacketRef<'_>) -> Result<ParsedFlac> {
        let p = self.front.parse(packet)?;
        if p.blocksize == 0 || p.blocksize > self.front.max_blocksize() {
            return decode_error("flac: block size outside the stream's bounds");
        }
        Ok(p)
    }

    fn transform(&mut self, batch: &[ParsedFlac]) -> Result<()> {
        // Block sizes may differ (the last frame of a stream, variable-block-size streams): every subframe gets a slot
        // of the batch's largest block size, zero-padded -- predicting the padding is harmless, it is never read back.
        let k = batch.len();
        self.stride = batch.iter().map(|p| p.blocksize).max().unwrap_or(0);
        self.lens.clear();
        self.modes.clear();
        self.shifts.clear();
        let words = self.words.as_mut_slice();
        for (i, p) in batch.iter().enumerate() {
            for c in 0..self.nch {
                let slot = (i * self.nch + c) * self.stride;
                words[slot..slot + p.blocksize].copy_from_slice(&p.words[c * p.blocksize..(c + 1) * p.blocksize]);
                words[slot + p.blocksize..slot + self.stride].fill(0);
                self.desc[i * self.nch + c] = p.desc[c];
                self.coeffs[(i * self.nch + c) * 32..(i * self.nch + c + 1) * 32].copy_from_slice(&p.coeffs[c * 32..(c + 1) * 32]);
            }
            self.lens.push(p.blocksize);
            self.modes.push(if self.nch == 2 { p.pair_mode } else { 0 });
            self.shifts.push(p.out_shift);
        }
        // SAFETY: `words`, `desc` and `coeffs` cover k * nch subframes of `stride` words (sized for max_batch frames of
        // the stream's maximum block size).
        check(
            unsafe {
                ffi::symaccel_flac_restore(self.ctx.raw(), words.as_mut_ptr(), self.desc.as_ptr(), self.coeffs.as_ptr(), k * self.nch, self.stride)
            },
            self.ctx.raw(),
        )
    }

    fn publish(&mut self, i: usize) {
        let n = self.lens[i];
        self.buf.clear();
        self.buf.render_uninit(Some(n));
        let words = self.words.as_slice();
        let base = i * self.nch * self.stride;
        // decorrelation + left-justification of ONE frame is 2 * blocksize operations: done here on the copy out (the
        // batched device form, symaccel_flac_restore_stereo_device, is for callers that keep the PCM on the GPU)
        let shift = self.shifts[i];
        for c in 0..self.nch {
            if let Some(plane) = self.buf.plane_mut(c) {
                for t in 0..n {
                    let own = words[base + c * self.stride + t];
                    let v = if self.nch == 2 && self.modes[i] != 0 {
                        let a = words[base + t];
                        let b = words[base + self.stride + t];
                        match (self.modes[i], c) {
                            (1, 1) => a.wrapping_sub(b),                                  // left/side: right = left - side
                            (2, _) => {
                                let mid = (a << 1) | (b & 1);                             // decoder.rs:38-73
                                if c == 0 { mid.wrapping_add(b) >> 1 } else { mid.wrapping_sub(b) >> 1 }
                            }
                            (3, 0) => a.wrapping_add(b),                                  // right/side: left = side + right
                            _ => own,
                        }
                    }
                    else {
                        own
                    };
                    plane[t] = v.wrapping_shl(shift);
                }
            }
        }
    }

    fn reset_state(&mut self) {}

    fn clear(&mut self) {
        self.buf.clear();
    }
}

impl DecoderBatch for FlacBatch {
    fn buffer(&self) -> GenericAudioBufferRef<'_> {
        self.buf.as_generic_audio_buffer_ref()
    }
}

crate::hip_decoder!(
    HipFlacDecoder,
    FlacBatch,
    ParsedFlac,
    crate::frontends::flac_front_end,
    &[support_audio_codec!(CODEC_ID_FLAC, "flac", "Free Lossless Audio Codec (MI355X predictors)")],
    "FLAC decoder with the same observable behaviour as `symphonia_bundle_flac::FlacDecoder` (verification off)."
);

impl HipFlacDecoder {
    pub fn try_new(params: &AudioCodecParameters, _opts: &AudioDecoderOptions, front: Box<dyn FlacFrontEnd>, max_batch: usize) -> Result<Self> {
        let (Some(rate), Some(channels)) = (params.sample_rate, params.channels.clone()) else {
            return unsupported_error("flac: sample rate and channels are required");
        };
        let nch = front.channels();
        let max_batch = max_batch.max(1);
        let max_bs = front.max_blocksize();
        Ok(HipFlacDecoder {
            params: params.clone(),
            batch: FlacBatch {
                ctx: Context::new(0)?,
                front,
                nch,
                stride: max_bs,
                words: Pinned::new(max_batch * nch * max_bs)?,
                desc: vec![ffi::SymaccelFlacDesc { kind: 0, order: 0, shift: 0, wasted_bits: 0 }; max_batch * nch],
                coeffs: vec![0; max_batch * nch * 32],
                lens: Vec::with_capacity(max_batch),
                modes: Vec::with_capacity(max_batch),
                shifts: Vec::with_capacity(max_batch),
                buf: AudioBuffer::new(AudioSpec::new(rate, channels), max_bs),
            },
            la: Lookahead::new(max_batch),
        })
    }
}
