//! `HipFlacDecoder`: the predictor stage of every subframe (fixed and LPC, symphonia-bundle-flac/src/decoder.rs:663-752),
//! the stereo decorrelation (:32-82) and the left-justification shift (:239-242) on the MI355X.  FLAC carries no state
//! from frame to frame, so a batch is simply many frames' subframes side by side.
use std::sync::{Arc, Mutex};

use symphonia_bundle_flac::{Decorrelation, FlacDecoder, SynthBackend};
use symphonia_core::audio::{Audio, AudioBuffer, AudioMut, AudioSpec, GenericAudioBufferRef};
use symphonia_core::codecs::audio::well_known::CODEC_ID_FLAC;
use symphonia_core::codecs::audio::{AudioCodecParameters, AudioDecoder, AudioDecoderOptions};
use symphonia_core::errors::{decode_error, unsupported_error, Result};
use symphonia_core::packet::PacketRef;
use symphonia_core::support_audio_codec;

use crate::ctx::{check, BatchSlot, Context, Pinned, Pool};
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
    pub coeffs: Vec<i32>,                // [channel][32], bitstream order (the first coefficient multiplies the most recent sample)
    pub pair_mode: u8,                   // 0 independent, 1 left/side, 2 mid/side, 3 right/side (stereo frames only)
    pub out_shift: u32,                  // 32 - bits per sample (decoder.rs:239-242)
}

pub trait FlacFrontEnd: Send + Sync {
    /// The stream's parameters as the reference's decoder amends them from STREAMINFO (decoder.rs:110-118): sample rate,
    /// channels, bits per sample, maximum block size.
    fn params(&self) -> &AudioCodecParameters;
    fn channels(&self) -> usize;
    fn max_blocksize(&self) -> usize;
    fn parse(&mut self, packet: &PacketRef<'_>) -> Result<ParsedFlac>;
}

/// What the reference's decoder tells its `SynthBackend` about one frame (bindings/rust/patches/symphonia-bundle-flac.diff):
/// the recording backend below stores it instead of computing anything, so the decoder's `AudioBuffer` is left holding
/// every subframe's warm-up samples and residuals -- the input of the batched device call.
#[derive(Default)]
pub struct FlacRecord {
    pub desc: Vec<ffi::SymaccelFlacDesc>, // [channel]
    pub coeffs: Vec<i32>,                 // [channel][32], bitstream order
    pub pair_mode: u8,
    pub out_shift: u32,
}

impl FlacRecord {
    fn begin_frame(&mut self, nch: usize) {
        self.desc.clear();
        self.desc.resize(nch, ffi::SymaccelFlacDesc { kind: 0, order: 0, shift: 0, wasted_bits: 0 });
        self.coeffs.clear();
        self.coeffs.resize(nch * 32, 0);
        self.pair_mode = 0;
        self.out_shift = 0;
    }
}

/// The `SynthBackend` handed to the reference's `FlacDecoder`: every operation after entropy decoding is recorded, none
/// is performed.
pub struct Recorder(pub Arc<Mutex<FlacRecord>>);

impl SynthBackend for Recorder {
    fn fixed_predict(&mut self, channel: usize, order: u32, _buf: &mut [i32]) {
        let mut rec = self.0.lock().expect("flac record poisoned");
        rec.desc[channel].kind = ffi::SYMACCEL_FLAC_FIXED as u8;
        rec.desc[channel].order = order as u8;
    }

    fn lpc_predict(&mut self, channel: usize, order: u32, qlp_coeffs: &[i32; 32], qlp_coeff_shift: u32, _buf: &mut [i32]) {
        let mut rec = self.0.lock().expect("flac record poisoned");
        rec.desc[channel].kind = ffi::SYMACCEL_FLAC_LPC as u8;
        rec.desc[channel].order = order as u8;
        rec.desc[channel].shift = qlp_coeff_shift as u8;
        // the decoder stores the first coefficient it reads at index 31 (decoder.rs:477-481); the C ABI wants bitstream order
        for j in 0..order as usize {
            rec.coeffs[channel * 32 + j] = qlp_coeffs[31 - j];
        }
    }

    fn samples_shl(&mut self, channel: usize, shift: u32, _buf: &mut [i32]) {
        let mut rec = self.0.lock().expect("flac record poisoned");
        rec.desc[channel].wasted_bits = shift as u8;
    }

    fn decorrelate(&mut self, mode: Decorrelation, _plane0: &mut [i32], _plane1: &mut [i32]) {
        let mut rec = self.0.lock().expect("flac record poisoned");
        rec.pair_mode = match mode {
            Decorrelation::LeftSide => 1,
            Decorrelation::MidSide => 2,
            Decorrelation::RightSide => 3,
        };
    }

    fn left_justify(&mut self, shift: u32, _buf: &mut AudioBuffer<i32>) {
        let mut rec = self.0.lock().expect("flac record poisoned");
        rec.out_shift = shift;
    }
}

/// `FlacFrontEnd` over the reference's own decoder with the recording backend installed: frame sync, frame header, subframe
/// headers and Rice decoding are symphonia-bundle-flac's code, unmodified.
pub struct SeamFrontEnd {
    dec: FlacDecoder,
    rec: Arc<Mutex<FlacRecord>>,
    nch: usize,
    max_bs: usize,
}

impl SeamFrontEnd {
    pub fn try_new(params: &AudioCodecParameters) -> Result<Self> {
        let rec: Arc<Mutex<FlacRecord>> = Arc::new(Mutex::new(FlacRecord::default()));
        // (verification is the caller's: the MD5 would be fed the residuals)
        let opts = AudioDecoderOptions { verify: false, ..Default::default() };
        let dec = FlacDecoder::try_new_with_backend(params, &opts, Box::new(Recorder(rec.clone())))?;
        // the decoder amends its parameters from STREAMINFO (decoder.rs:110-118)
        let (Some(channels), Some(max_bs)) = (dec.codec_params().channels.clone(), dec.codec_params().max_frames_per_packet) else {
            return unsupported_error("flac: stream info is required");
        };
        Ok(SeamFrontEnd { dec, rec, nch: channels.count(), max_bs: max_bs as usize })
    }
}

impl FlacFrontEnd for SeamFrontEnd {
    fn params(&self) -> &AudioCodecParameters {
        self.dec.codec_params()
    }

    fn channels(&self) -> usize {
        self.nch
    }

    fn max_blocksize(&self) -> usize {
        self.max_bs
    }

    fn parse(&mut self, packet: &PacketRef<'_>) -> Result<ParsedFlac> {
        self.rec.lock().expect("flac record poisoned").begin_frame(self.nch);
        let planes = match self.dec.decode_ref(packet)? {
            GenericAudioBufferRef::S32(planes) => planes,
            _ => return decode_error("flac: the front end's buffer is not 32-bit"),
        };
        let blocksize = planes.frames();
        let mut words = Vec::with_capacity(self.nch * blocksize);
        for c in 0..self.nch {
            match planes.plane(c) {
                Some(plane) => words.extend_from_slice(plane),
                None => return decode_error("flac: missing audio plane"),
            }
        }
        let rec = self.rec.lock().expect("flac record poisoned");
        Ok(ParsedFlac { blocksize, words, desc: rec.desc.clone(), coeffs: rec.coeffs.clone(), pair_mode: rec.pair_mode, out_shift: rec.out_shift })
    }
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
    // the cross-stream batcher (SYMACCEL_BATCH_FLAC_RESTORE: a chain is a subframe, units = the stream's maximum block size, so the
    // batches of every FLAC stream with that block size share launches whatever their lengths): `cur` holds the batch being handed
    // out -- its restored words are read where the device left them, in the page-locked slot --, `next` the one submitted ahead
    pool: Option<Arc<Pool>>,
    cur: Option<BatchSlot>,
    next: Option<BatchSlot>,
    next_lens: Vec<usize>,
    next_modes: Vec<u8>,
    next_shifts: Vec<u32>,
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
        // (a batch that came through the batcher is done with: this one is published from `words`)
        if let (Some(pool), Some(old)) = (self.pool.clone(), self.cur.take()) {
            pool.release(old);
        }
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
        // the restored subframes: in the batcher's slot (zero-copy: the device wrote them there), or in this decoder's own buffer
        let words: &[i32] = match &self.cur {
            Some(slot) => slot.out::<i32>(),
            None => self.words.as_slice(),
        };
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

    fn pooled(&self) -> bool {
        self.pool.is_some()
    }

    /// The stream's next batch goes to the process-wide batcher: the subframes are written straight into a page-locked slot
    /// (`Pool::reserve` -> fill -> `commit`); the device restores them in place, in one launch with the other streams' batches.
    fn submit(&mut self, batch: &[ParsedFlac]) -> Result<()> {
        let Some(pool) = self.pool.clone() else {
            return unsupported_error("flac: no batcher");
        };
        if batch.is_empty() || self.next.is_some() {
            return unsupported_error("flac: one batch at a time");
        }
        let (k, nch) = (batch.len(), self.nch);
        let stride = self.front.max_blocksize(); // (the same for every batch of the stream: the group key)
        let mut slot = pool.reserve(ffi::SYMACCEL_BATCH_FLAC_RESTORE as i32, 0, k * nch, stride)?;
        self.next_lens.clear();
        self.next_modes.clear();
        self.next_shifts.clear();
        {
            let words = slot.input::<i32>(0);
            for (i, p) in batch.iter().enumerate() {
                for c in 0..nch {
                    let at = (i * nch + c) * stride;
                    words[at..at + p.blocksize].copy_from_slice(&p.words[c * p.blocksize..(c + 1) * p.blocksize]);
                    words[at + p.blocksize..at + stride].fill(0);
                }
            }
        }
        {
            let desc = slot.input::<ffi::SymaccelFlacDesc>(1);
            for (i, p) in batch.iter().enumerate() {
                for c in 0..nch {
                    desc[i * nch + c] = p.desc[c];
                }
            }
        }
        {
            let coeffs = slot.input::<i32>(2);
            for (i, p) in batch.iter().enumerate() {
                for c in 0..nch {
                    coeffs[(i * nch + c) * 32..(i * nch + c + 1) * 32].copy_from_slice(&p.coeffs[c * 32..(c + 1) * 32]);
                }
            }
        }
        for p in batch {
            self.next_lens.push(p.blocksize);
            self.next_modes.push(if nch == 2 { p.pair_mode } else { 0 });
            self.next_shifts.push(p.out_shift);
        }
        if let Err(e) = pool.commit(&mut slot) {
            pool.release(slot);
            return Err(e);
        }
        self.next = Some(slot);
        Ok(())
    }

    fn collect(&mut self) -> Result<()> {
        let (Some(pool), Some(mut slot)) = (self.pool.clone(), self.next.take()) else {
            return unsupported_error("flac: nothing was submitted");
        };
        if let Err(e) = pool.wait(&mut slot) {
            pool.release(slot);
            return Err(e);
        }
        if let Some(old) = self.cur.take() {
            pool.release(old);
        }
        self.cur = Some(slot);
        self.stride = self.front.max_blocksize();
        std::mem::swap(&mut self.lens, &mut self.next_lens);
        std::mem::swap(&mut self.modes, &mut self.next_modes);
        std::mem::swap(&mut self.shifts, &mut self.next_shifts);
        Ok(())
    }

    fn hint(&mut self) {
        if let Some(pool) = &self.pool {
            pool.hint();
        }
    }

    fn abandon(&mut self) {
        if let (Some(pool), Some(slot)) = (self.pool.clone(), self.next.take()) {
            pool.release(slot);
        }
    }
}

impl Drop for FlacBatch {
    fn drop(&mut self) {
        BatchCodec::abandon(self);
        if let (Some(pool), Some(slot)) = (self.pool.clone(), self.cur.take()) {
            pool.release(slot);
        }
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
    pub fn try_new(params: &AudioCodecParameters, opts: &AudioDecoderOptions, front: Box<dyn FlacFrontEnd>, max_batch: usize) -> Result<Self> {
        Self::try_new_with_pool(params, opts, front, max_batch, None)
    }

    /// The same decoder submitting to the process-wide cross-stream batcher (`Pool::shared()`): with many streams open, the
    /// subframes of all of them are restored in one launch (csrc/batcher.cpp, SYMACCEL_BATCH_FLAC_RESTORE).
    pub fn try_new_pooled(params: &AudioCodecParameters, opts: &AudioDecoderOptions, front: Box<dyn FlacFrontEnd>, max_batch: usize) -> Result<Self> {
        Self::try_new_with_pool(params, opts, front, max_batch, Some(Pool::shared()?))
    }

    pub fn try_new_with_pool(
        _params: &AudioCodecParameters,
        opts: &AudioDecoderOptions,
        front: Box<dyn FlacFrontEnd>,
        max_batch: usize,
        pool: Option<Arc<Pool>>,
    ) -> Result<Self> {
        if opts.verify {
            // the MD5 of STREAMINFO is computed over the decoded audio inside the reference's decoder; here the caller verifies
            return unsupported_error("flac: verification is not available with the batched predictors");
        }
        let params = front.params().clone();
        let (Some(rate), Some(channels)) = (params.sample_rate, params.channels.clone()) else {
            return unsupported_error("flac: sample rate and channels are required");
        };
        let nch = front.channels();
        let max_batch = max_batch.max(1);
        let max_bs = front.max_blocksize();
        Ok(HipFlacDecoder {
            params,
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
                pool,
                cur: None,
                next: None,
                next_lens: Vec::with_capacity(max_batch),
                next_modes: Vec::with_capacity(max_batch),
                next_shifts: Vec::with_capacity(max_batch),
                buf: AudioBuffer::new(AudioSpec::new(rate, channels), max_bs),
            },
            la: Lookahead::new(max_batch),
        })
    }
}
