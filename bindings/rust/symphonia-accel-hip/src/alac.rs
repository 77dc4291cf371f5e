//! `HipAlacDecoder`: the adaptive predictor of every element channel (symphonia-codec-alac/src/lib.rs:165-264) on the MI355X;
//! the mid-side decorrelation (:664-671), the separately coded low bits (:574-598) and the left-justification (:409-414) of ONE
//! packet are a few operations per sample and happen on the copy out.  ALAC carries no state from packet to packet, so a batch
//! is simply many packets' element channels side by side.
use std::sync::{Arc, Mutex};

use symphonia_codec_alac::{AlacDecoder, Prediction, SynthBackend};
use symphonia_common::apple::audio::alac::MagicCookie;
use symphonia_core::audio::{Audio, AudioBuffer, AudioMut, AudioSpec, GenericAudioBufferRef};
use symphonia_core::codecs::audio::well_known::CODEC_ID_ALAC;
use symphonia_core::codecs::audio::{AudioCodecParameters, AudioDecoder, AudioDecoderOptions};
use symphonia_core::errors::{decode_error, unsupported_error, Result};
use symphonia_core::packet::PacketRef;
use symphonia_core::support_audio_codec;

use crate::ctx::{check, BatchSlot, Context, Pinned, Pool};
use crate::decoder::DecoderBatch;
use crate::ffi;
use crate::lookahead::{BatchCodec, Lookahead};

/// The mid-side parameters of one channel pair element (lib.rs:541-560).
#[derive(Clone, Copy)]
pub struct AlacPair {
    pub plane0: usize,
    pub plane1: usize,
    pub weight: i32,
    pub shift: u8,
}

/// The separately coded low bits of one element (lib.rs:531-539, 574-598); interleaved for a channel pair.
#[derive(Clone)]
pub struct AlacTail {
    pub plane0: usize,
    pub plane1: Option<usize>,
    pub shift: u8,
    pub bits: Vec<u16>,
}

/// One packet after the CPU front end (element headers, adaptive Golomb-Rice decoding: lib.rs:83-163, 470-539): per plane
/// `frames` words -- the residuals of a compressed element, or the samples of an uncompressed one (predictor order 0) --, the
/// element channel's predictor, and what follows the predictor.
pub struct ParsedAlac {
    pub frames: usize,
    pub words: Vec<i32>,                 // [plane][frames]
    pub desc: Vec<ffi::SymaccelAlacDesc>, // [plane]
    pub coeffs: Vec<i32>,                // [plane][32]
    pub pairs: Vec<AlacPair>,
    pub tails: Vec<AlacTail>,
    pub out_shift: u32,                  // 32 - bit depth
}

pub trait AlacFrontEnd: Send + Sync {
    /// The stream's parameters as the reference's decoder amends them from the magic cookie (lib.rs:305-308).
    fn params(&self) -> &AudioCodecParameters;
    fn channels(&self) -> usize;
    fn max_frames(&self) -> usize;
    fn parse(&mut self, packet: &PacketRef<'_>) -> Result<ParsedAlac>;
}

/// What the reference's decoder tells its `SynthBackend` about one packet (bindings/rust/patches/symphonia-codec-alac.diff): the
/// recording backend below stores it instead of computing anything, so the decoder's `AudioBuffer` is left holding every element
/// channel's residuals -- the input of the batched device call.
#[derive(Default)]
pub struct AlacRecord {
    pub desc: Vec<ffi::SymaccelAlacDesc>,
    pub coeffs: Vec<i32>,
    pub pairs: Vec<AlacPair>,
    pub tails: Vec<AlacTail>,
    pub out_shift: u32,
}

impl AlacRecord {
    fn begin_packet(&mut self, nch: usize) {
        self.desc.clear();
        // (order 0: no prediction -- planes of uncompressed elements, and planes the packet does not code, stay as they are)
        self.desc.resize(nch, ffi::SymaccelAlacDesc { mode: 0, lpc_order: 0, shift: 0, bps: 32 });
        self.coeffs.clear();
        self.coeffs.resize(nch * 32, 0);
        self.pairs.clear();
        self.tails.clear();
        self.out_shift = 0;
    }
}

/// The `SynthBackend` handed to the reference's `AlacDecoder`: every operation after entropy decoding is recorded, none is
/// performed.
pub struct Recorder(pub Arc<Mutex<AlacRecord>>);

impl SynthBackend for Recorder {
    fn predict(&mut self, channel: usize, pred: &Prediction, _out: &mut [i32]) -> Result<()> {
        // lib.rs:167-169: the check the predictor itself starts with
        if pred.mode > 0 && pred.mode < 15 {
            return decode_error("alac: invalid mode");
        }
        let mut rec = self.0.lock().expect("alac record poisoned");
        rec.desc[channel] = ffi::SymaccelAlacDesc { mode: pred.mode as u8, lpc_order: pred.lpc_order as u8, shift: pred.shift as u8, bps: pred.bps as u8 };
        for j in 0..32 {
            rec.coeffs[channel * 32 + j] = pred.lpc_coeffs[j];
        }
        Ok(())
    }

    fn decorrelate_mid_side(&mut self, channel0: usize, channel1: usize, _out0: &mut [i32], _out1: &mut [i32], weight: i32, shift: u8) {
        let mut rec = self.0.lock().expect("alac record poisoned");
        rec.pairs.push(AlacPair { plane0: channel0, plane1: channel1, weight, shift });
    }

    fn append_tail_bits(&mut self, channel0: usize, channel1: Option<usize>, shift: u8, tail_bits: &[u16], _out0: &mut [i32], _out1: Option<&mut [i32]>) {
        let mut rec = self.0.lock().expect("alac record poisoned");
        rec.tails.push(AlacTail { plane0: channel0, plane1: channel1, shift, bits: tail_bits.to_vec() });
    }

    fn left_justify(&mut self, shift: u32, _buf: &mut AudioBuffer<i32>) {
        let mut rec = self.0.lock().expect("alac record poisoned");
        rec.out_shift = shift;
    }
}

/// `AlacFrontEnd` over the reference's own decoder with the recording backend installed: element parsing and the adaptive
/// Golomb-Rice decoder are symphonia-codec-alac's code, unmodified.
pub struct SeamFrontEnd {
    dec: AlacDecoder,
    rec: Arc<Mutex<AlacRecord>>,
    nch: usize,
    max_frames: usize,
}

impl SeamFrontEnd {
    pub fn try_new(params: &AudioCodecParameters, opts: &AudioDecoderOptions) -> Result<Self> {
        let rec: Arc<Mutex<AlacRecord>> = Arc::new(Mutex::new(AlacRecord::default()));
        let dec = AlacDecoder::try_new_with_backend(params, opts, Box::new(Recorder(rec.clone())))?;
        // the decoder has accepted the magic cookie; its buffer is sized for the cookie's frame length (lib.rs:288-298)
        let cookie = match &params.extra_data {
            Some(extra_data) => MagicCookie::read(extra_data)?,
            None => return unsupported_error("alac: missing extra data"),
        };
        Ok(SeamFrontEnd { dec, rec, nch: usize::from(cookie.num_channels), max_frames: cookie.frame_length as usize })
    }
}

impl AlacFrontEnd for SeamFrontEnd {
    fn params(&self) -> &AudioCodecParameters {
        self.dec.codec_params()
    }

    fn channels(&self) -> usize {
        self.nch
    }

    fn max_frames(&self) -> usize {
        self.max_frames
    }

    fn parse(&mut self, packet: &PacketRef<'_>) -> Result<ParsedAlac> {
        self.rec.lock().expect("alac record poisoned").begin_packet(self.nch);
        let planes = match self.dec.decode_ref(packet)? {
            GenericAudioBufferRef::S32(planes) => planes,
            _ => return decode_error("alac: the front end's buffer is not 32-bit"),
        };
        let frames = planes.frames();
        let mut words = Vec::with_capacity(self.nch * frames);
        for c in 0..self.nch {
            match planes.plane(c) {
                Some(plane) => words.extend_from_slice(plane),
                None => return decode_error("alac: missing audio plane"),
            }
        }
        let rec = self.rec.lock().expect("alac record poisoned");
        Ok(ParsedAlac {
            frames,
            words,
            desc: rec.desc.clone(),
            coeffs: rec.coeffs.clone(),
            pairs: rec.pairs.clone(),
            tails: rec.tails.clone(),
            out_shift: rec.out_shift,
        })
    }
}

pub struct AlacBatch {
    ctx: Context,
    front: Box<dyn AlacFrontEnd>,
    nch: usize,
    stride: usize,                      // words per element-channel slot of the current batch (its longest packet)
    words: Pinned<i32>,                 // [packet][plane][stride]
    desc: Vec<ffi::SymaccelAlacDesc>,   // [packet][plane]
    coeffs: Vec<i32>,                   // [packet][plane][32]
    lens: Vec<usize>,
    pairs: Vec<Vec<AlacPair>>,
    tails: Vec<Vec<AlacTail>>,
    shifts: Vec<u32>,
    // the cross-stream batcher (SYMACCEL_BATCH_ALAC_PREDICT: a chain is an element channel, units = the stream's frame length): `cur`
    // holds the batch being handed out -- what follows the predictor is applied in place in the page-locked slot --, `next` the one
    // submitted ahead
    pool: Option<Arc<Pool>>,
    cur: Option<BatchSlot>,
    next: Option<BatchSlot>,
    next_lens: Vec<usize>,
    next_pairs: Vec<Vec<AlacPair>>,
    next_tails: Vec<Vec<AlacTail>>,
    next_shifts: Vec<u32>,
    buf: AudioBuffer<i32>,
}

impl BatchCodec for AlacBatch {
    type Parsed = ParsedAlac;

    fn parse(&mut self, packet: &PacketRef<'_>) -> Result<ParsedAlac> {
        let p = self.front.parse(packet)?;
        if p.frames > self.front.max_frames() {
            return decode_error("alac: frame length outside the stream's bounds");
        }
        Ok(p)
    }

    fn transform(&mut self, batch: &[ParsedAlac]) -> Result<()> {
        // (a batch that came through the batcher is done with: this one is published from `words`)
        if let (Some(pool), Some(old)) = (self.pool.clone(), self.cur.take()) {
            pool.release(old);
        }
        // Packets may differ in length (the last one of a stream): every element channel gets a slot of the batch's longest
        // packet, zero-padded -- predicting the padding is harmless, it is never read back.
        let k = batch.len();
        self.stride = batch.iter().map(|p| p.frames).max().unwrap_or(0);
        self.lens.clear();
        self.pairs.clear();
        self.tails.clear();
        self.shifts.clear();
        let words = self.words.as_mut_slice();
        for (i, p) in batch.iter().enumerate() {
            for c in 0..self.nch {
                let slot = (i * self.nch + c) * self.stride;
                words[slot..slot + p.frames].copy_from_slice(&p.words[c * p.frames..(c + 1) * p.frames]);
                words[slot + p.frames..slot + self.stride].fill(0);
                self.desc[i * self.nch + c] = p.desc[c];
                self.coeffs[(i * self.nch + c) * 32..(i * self.nch + c + 1) * 32].copy_from_slice(&p.coeffs[c * 32..(c + 1) * 32]);
            }
            self.lens.push(p.frames);
            self.pairs.push(p.pairs.clone());
            self.tails.push(p.tails.clone());
            self.shifts.push(p.out_shift);
        }
        if self.stride == 0 {
            return Ok(());
        }
        // SAFETY: `words`, `desc` and `coeffs` cover k * nch element channels of `stride` words (sized for max_batch packets of
        // the stream's frame length).
        check(
            unsafe {
                ffi::symaccel_alac_predict(self.ctx.raw(), words.as_mut_ptr(), self.desc.as_ptr(), self.coeffs.as_ptr(), k * self.nch, self.stride)
            },
            self.ctx.raw(),
        )
    }

    fn publish(&mut self, i: usize) {
        let n = self.lens[i];
        self.buf.clear();
        self.buf.render_uninit(Some(n));
        // the predicted element channels: in the batcher's slot (in place: the result plane IS input plane 0), or in this decoder's buffer
        let words: &mut [i32] = match &mut self.cur {
            Some(slot) => slot.input::<i32>(0),
            None => self.words.as_mut_slice(),
        };
        let base = i * self.nch * self.stride;
        // what follows the predictor, in the decoder's order (lib.rs:541-598, 409-414), on this packet's slots
        for pair in &self.pairs[i] {
            for t in 0..n {
                let s0 = words[base + pair.plane0 * self.stride + t];
                let s1 = words[base + pair.plane1 * self.stride + t];
                let left = s0 + s1 - ((s1 * pair.weight) >> pair.shift); // lib.rs:668-669
                words[base + pair.plane0 * self.stride + t] = left;
                words[base + pair.plane1 * self.stride + t] = left - s1;
            }
        }
        for tail in &self.tails[i] {
            for t in 0..n {
                match tail.plane1 {
                    Some(plane1) => {
                        let a = base + tail.plane0 * self.stride + t;
                        let b = base + plane1 * self.stride + t;
                        words[a] = (words[a] << tail.shift) | tail.bits[2 * t] as i32;
                        words[b] = (words[b] << tail.shift) | tail.bits[2 * t + 1] as i32;
                    }
                    None => {
                        let a = base + tail.plane0 * self.stride + t;
                        words[a] = (words[a] << tail.shift) | tail.bits[t] as i32;
                    }
                }
            }
        }
        let shift = self.shifts[i];
        for c in 0..self.nch {
            if let Some(plane) = self.buf.plane_mut(c) {
                for t in 0..n {
                    plane[t] = words[base + c * self.stride + t].wrapping_shl(shift);
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

    /// The stream's next batch goes to the process-wide batcher: residuals, predictors and coefficients are written straight into a
    /// page-locked slot; the device predicts in place, in one launch with the other streams' batches.
    fn submit(&mut self, batch: &[ParsedAlac]) -> Result<()> {
        let Some(pool) = self.pool.clone() else {
            return unsupported_error("alac: no batcher");
        };
        if batch.is_empty() || self.next.is_some() {
            return unsupported_error("alac: one batch at a time");
        }
        let (k, nch) = (batch.len(), self.nch);
        let stride = self.front.max_frames(); // (the same for every batch of the stream: the group key)
        if stride == 0 {
            return unsupported_error("alac: empty packets");
        }
        let mut slot = pool.reserve(ffi::SYMACCEL_BATCH_ALAC_PREDICT as i32, 0, k * nch, stride)?;
        {
            let words = slot.input::<i32>(0);
            for (i, p) in batch.iter().enumerate() {
                for c in 0..nch {
                    let at = (i * nch + c) * stride;
                    words[at..at + p.frames].copy_from_slice(&p.words[c * p.frames..(c + 1) * p.frames]);
                    words[at + p.frames..at + stride].fill(0);
                }
            }
        }
        {
            let desc = slot.input::<ffi::SymaccelAlacDesc>(1);
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
        self.next_lens.clear();
        self.next_pairs.clear();
        self.next_tails.clear();
        self.next_shifts.clear();
        for p in batch {
            self.next_lens.push(p.frames);
            self.next_pairs.push(p.pairs.clone());
            self.next_tails.push(p.tails.clone());
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
            return unsupported_error("alac: nothing was submitted");
        };
        if let Err(e) = pool.wait(&mut slot) {
            pool.release(slot);
            return Err(e);
        }
        if let Some(old) = self.cur.take() {
            pool.release(old);
        }
        self.cur = Some(slot);
        self.stride = self.front.max_frames();
        std::mem::swap(&mut self.lens, &mut self.next_lens);
        std::mem::swap(&mut self.pairs, &mut self.next_pairs);
        std::mem::swap(&mut self.tails, &mut self.next_tails);
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

impl Drop for AlacBatch {
    fn drop(&mut self) {
        BatchCodec::abandon(self);
        if let (Some(pool), Some(slot)) = (self.pool.clone(), self.cur.take()) {
            pool.release(slot);
        }
    }
}

impl DecoderBatch for AlacBatch {
    fn buffer(&self) -> GenericAudioBufferRef<'_> {
        self.buf.as_generic_audio_buffer_ref()
    }
}

crate::hip_decoder!(
    HipAlacDecoder,
    AlacBatch,
    ParsedAlac,
    crate::frontends::alac_front_end,
    &[support_audio_codec!(CODEC_ID_ALAC, "alac", "Apple Lossless Audio Codec (MI355X predictors)")],
    "ALAC decoder with the same observable behaviour as `symphonia_codec_alac::AlacDecoder`."
);

impl HipAlacDecoder {
    pub fn try_new(params: &AudioCodecParameters, opts: &AudioDecoderOptions, front: Box<dyn AlacFrontEnd>, max_batch: usize) -> Result<Self> {
        Self::try_new_with_pool(params, opts, front, max_batch, None)
    }

    /// The same decoder submitting to the process-wide cross-stream batcher (`Pool::shared()`): the element channels of every open
    /// ALAC stream with this frame length are predicted in one launch (csrc/batcher.cpp, SYMACCEL_BATCH_ALAC_PREDICT).
    pub fn try_new_pooled(params: &AudioCodecParameters, opts: &AudioDecoderOptions, front: Box<dyn AlacFrontEnd>, max_batch: usize) -> Result<Self> {
        Self::try_new_with_pool(params, opts, front, max_batch, Some(Pool::shared()?))
    }

    pub fn try_new_with_pool(
        _params: &AudioCodecParameters,
        _opts: &AudioDecoderOptions,
        front: Box<dyn AlacFrontEnd>,
        max_batch: usize,
        pool: Option<Arc<Pool>>,
    ) -> Result<Self> {
        let params = front.params().clone();
        let (Some(rate), Some(channels)) = (params.sample_rate, params.channels.clone()) else {
            return unsupported_error("alac: sample rate and channels are required");
        };
        let nch = front.channels();
        let max_batch = max_batch.max(1);
        let max_frames = front.max_frames();
        Ok(HipAlacDecoder {
            params,
            batch: AlacBatch {
                ctx: Context::new(0)?,
                front,
                nch,
                stride: max_frames,
                words: Pinned::new(max_batch * nch * max_frames)?,
                desc: vec![ffi::SymaccelAlacDesc { mode: 0, lpc_order: 0, shift: 0, bps: 32 }; max_batch * nch],
                coeffs: vec![0; max_batch * nch * 32],
                lens: Vec::with_capacity(max_batch),
                pairs: Vec::with_capacity(max_batch),
                tails: Vec::with_capacity(max_batch),
                shifts: Vec::with_capacity(max_batch),
                pool,
                cur: None,
                next: None,
                next_lens: Vec::with_capacity(max_batch),
                next_pairs: Vec::with_capacity(max_batch),
                next_tails: Vec::with_capacity(max_batch),
                next_shifts: Vec::with_capacity(max_batch),
                buf: AudioBuffer::new(AudioSpec::new(rate, channels), max_frames),
            },
            la: Lookahead::new(max_batch),
        })
    }
}
