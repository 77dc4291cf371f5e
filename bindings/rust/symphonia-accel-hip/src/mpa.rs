//! `HipMpaDecoder`: MPEG-1/2/2.5 Layer III with everything behind the entropy decoder on the MI355X -- requantization and
//! joint stereo (symphonia-bundle-mp3/src/layer3/requantize.rs:240-380, stereo.rs:485-556), then the synthesis tail: reorder,
//! antialias, hybrid synthesis, frequency inversion and the polyphase filterbank (layer3/hybrid_synthesis.rs:153-485,
//! synthesis.rs:158-336; caller layer3/mod.rs:421-476).  The front end hands over what `read_huffman_samples` produced as
//! 16-bit integers + the side records (`symaccel_mp3_decode_pipelined`: 2 bytes per spectral line across PCIe instead of 4,
//! one kernel); a front end that delivers requantized f32 spectra (`ParsedMpa::fused == None`) takes `symaccel_mp3_synth`.
use std::sync::{Arc, Mutex, OnceLock};

use symphonia_bundle_mp3::backend::{GranuleQuant, GranuleSide, GranuleStereo, SynthBackend};
use symphonia_bundle_mp3::MpaDecoder;
use symphonia_core::audio::{Audio, AudioBuffer, AudioMut, AudioSpec, GenericAudioBufferRef};
use symphonia_core::codecs::audio::well_known::CODEC_ID_MP3;
use symphonia_core::codecs::audio::{AudioCodecParameters, AudioDecoder, AudioDecoderOptions};
use symphonia_core::errors::{decode_error, unsupported_error, Result};
use symphonia_core::packet::PacketRef;
use symphonia_core::support_audio_codec;

use crate::ctx::{check, BatchSlot, Context, Pinned, Pool};
use crate::decoder::DecoderBatch;
use crate::ffi;
use crate::lookahead::{BatchCodec, Lookahead};

/// One frame after the CPU front end (`Layer3::decode` up to, not including, the per-channel tail of
/// layer3/mod.rs:440-476): per granule and channel the 576 samples after requantize + stereo processing and the three
/// `GranuleChannel` fields the tail reads.  `xr[granule][channel][576]`, `side[granule][channel]`.
pub struct ParsedMpa {
    pub trim: (usize, usize), // frames to trim from the start / end of the decoded packet when gapless (decoder.rs:128-131)
    pub n_granules: usize, // 2 for MPEG-1, 1 for MPEG-2 / 2.5
    pub xr: Vec<f32>,      // empty when `fused` is there
    pub side: Vec<ffi::SymaccelMp3Side>,
    pub fused: Option<FusedMpa>,
}

/// One frame as the ENTROPY decoder leaves it (`Layer3::decode` up to, not including, `requantize` at layer3/mod.rs:421-428):
/// the Huffman samples as integers and what requantize (requantize.rs:240-380) and stereo (stereo.rs:485-556) read of the
/// side information.  `ParsedMpa::side` then carries the rzero the reference has AFTER stereo (stereo.rs:549-553).
pub struct FusedMpa {
    pub quant: Vec<i16>,                  // [granule][channel][576]
    pub rq: Vec<ffi::SymaccelMp3Requant>, // [granule][channel]
    pub st: Vec<ffi::SymaccelMp3Stereo>,  // [granule]; flags 0 where the frame is not joint-stereo coded
}

/// The reference's bitstream reader, Huffman decoder, requantizer and stereo processor, vendored (they are private to
/// symphonia-bundle-mp3).  `sample_rate_idx` selects the scale-factor-band table `reorder` uses (0..8).
pub trait MpaFrontEnd: Send + Sync {
    fn channels(&self) -> usize;
    fn sample_rate_idx(&self) -> i32;
    fn parse(&mut self, packet: &PacketRef<'_>) -> Result<ParsedMpa>;
    /// `AudioDecoder::reset` (decoder.rs:149-152): the parse stage forgets the bit reservoir.
    fn reset(&mut self);
}

/// What the reference's decoder hands its `SynthBackend` for one packet (bindings/rust/patches/symphonia-bundle-mp3.diff).
#[derive(Default)]
pub struct MpaRecord {
    pub xr: Vec<f32>,                    // [granule][channel][576], in call order
    pub side: Vec<ffi::SymaccelMp3Side>, // [granule][channel]
    pub quant: Vec<i16>,                  // the fused form (decode_granule): [granule][channel][576]
    pub rq: Vec<ffi::SymaccelMp3Requant>, // [granule][channel]
    pub st: Vec<ffi::SymaccelMp3Stereo>,  // [granule]
    pub bad_sample: bool,                 // a sample that is not sign * s^(4/3) of an integer s (cannot happen: see quantised)
    pub sample_rate_idx: i32,
    pub resets: usize,
}

/// `POW43` of the reference (requantize.rs:22-31): `f32::powf(i as f32, 4.0 / 3.0)` for i in 0..8207, the values
/// `read_huffman_samples` stores for a Huffman sample i (requantize.rs:126, 139; +-1.0 in the count1 partition).
static SAMPLE_VALUES: OnceLock<Vec<f32>> = OnceLock::new();

fn pow43() -> &'static Vec<f32> {
    SAMPLE_VALUES.get_or_init(|| (0..8207).map(|i| f32::powf(i as f32, 4.0 / 3.0)).collect())
}

/// The Huffman sample behind a value of `read_huffman_samples`: the s with `POW43[s] == |x|`, signed.  The table is strictly
/// increasing, so the s is unique; it is found from the rounded 3/4 power and CHECKED against the table (`None`: x is not a
/// table value, which the reference's Huffman decoder cannot produce).  A seam that handed the integers over directly would
/// save this step; it keeps the patch to the reference at one `if` in the granule loop instead.
fn quantised(x: f32) -> Option<i16> {
    if x == 0.0 {
        return Some(0);
    }
    let table = pow43();
    let a = x.abs();
    let guess = (f32::powf(a, 0.75) + 0.5) as usize;
    for s in [guess, guess + 1, guess.saturating_sub(1)] {
        if s < table.len() && table[s] == a {
            return Some(if x < 0.0 { -(s as i16) } else { s as i16 });
        }
    }
    None
}

fn block_type_of(side: &GranuleSide) -> (u8, u8) {
    (side.block_type, side.is_mixed as u8)
}

/// The `SynthBackend` handed to the reference's `MpaDecoder`: the samples of every granule-channel after requantisation
/// and joint-stereo processing are recorded, nothing is synthesized.
pub struct Recorder(pub Arc<Mutex<MpaRecord>>, pub bool);

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

    /// `Recorder(_, true)`: take the granules in front of requantize (the second seam of the patch).
    fn takes_granules(&self) -> bool {
        self.1
    }

    fn decode_granule(
        &mut self,
        _gr: usize,
        quant: &[GranuleQuant],
        stereo: Option<GranuleStereo>,
        samples: &[[f32; 576]; 2],
        _out: &mut AudioBuffer<f32>,
    ) {
        let mut rec = self.0.lock().expect("mp3 record poisoned");
        // stereo.rs:549-553: with either joint-stereo tool on, both channels end at the larger rzero
        let joint = match stereo {
            Some(st) => st.mid_side || st.intensity,
            None => false,
        };
        let end = quant.iter().map(|q| q.side.rzero).max().unwrap_or(0).min(576);
        for (ch, q) in quant.iter().enumerate() {
            for x in samples[ch].iter() {
                match quantised(*x) {
                    Some(v) => rec.quant.push(v),
                    None => {
                        rec.quant.push(0);
                        rec.bad_sample = true;
                    }
                }
            }
            let (block_type, is_mixed) = block_type_of(&q.side);
            let rzero = q.side.rzero.min(576) as u16;
            let mut flags = 0u8;
            if q.scalefac_scale {
                flags |= ffi::SYMACCEL_MP3_RQ_SCALEFAC_SCALE as u8;
            }
            if q.preflag {
                flags |= ffi::SYMACCEL_MP3_RQ_PREFLAG as u8;
            }
            rec.rq.push(ffi::SymaccelMp3Requant {
                global_gain: q.global_gain,
                flags,
                block_type,
                is_mixed,
                subblock_gain: q.subblock_gain,
                reserved: 0,
                rzero,
                scalefacs: q.scalefacs,
                pad: [0; 3],
            });
            rec.side.push(ffi::SymaccelMp3Side { block_type, is_mixed, rzero: if joint { end as u16 } else { rzero } });
            rec.sample_rate_idx = q.side.sample_rate_idx as i32;
        }
        let mut st = ffi::SymaccelMp3Stereo { flags: 0, block_type: 0, is_mixed: 0, reserved: 0, rzero0: 0, rzero1: 0, scalefacs1: [0; 39], pad: 0 };
        if let (Some(mode), 2) = (stereo, quant.len()) {
            if mode.mid_side {
                st.flags |= ffi::SYMACCEL_MP3_ST_MID_SIDE as u8;
            }
            if mode.intensity {
                st.flags |= ffi::SYMACCEL_MP3_ST_INTENSITY as u8;
            }
            if mode.is_mpeg1 {
                st.flags |= ffi::SYMACCEL_MP3_ST_MPEG1 as u8;
            }
            if quant[1].scalefac_compress & 1 != 0 {
                st.flags |= ffi::SYMACCEL_MP3_ST_IS_SCALE as u8;
            }
            let (block_type, is_mixed) = block_type_of(&quant[1].side);
            st.block_type = block_type;
            st.is_mixed = is_mixed;
            st.rzero0 = quant[0].side.rzero.min(576) as u16;
            st.rzero1 = quant[1].side.rzero.min(576) as u16;
            st.scalefacs1 = quant[1].scalefacs;
        }
        rec.st.push(st);
    }
}

/// Index of `rate` in the order the reference's frame header uses (common.rs: 44100, 48000, 32000, 22050, 24000, 16000,
/// 11025, 12000, 8000).
fn sample_rate_index(rate: u32) -> Option<i32> {
    const RATES: [u32; 9] = [44_100, 48_000, 32_000, 22_050, 24_000, 16_000, 11_025, 12_000, 8_000];
    RATES.iter().position(|r| *r == rate).map(|i| i as i32)
}

/// `MpaFrontEnd` over the reference's own decoder with the recording backend installed: header, side information, bit reservoir,
/// scale factors and Huffman decoding are symphonia-bundle-mp3's code, unmodified.  `fused` (the default) stops there -- the
/// recorder takes the granules in front of requantize and `parse` returns integers + records (`ParsedMpa::fused`); without
/// it the reference's requantize and stereo run too and `parse` returns f32 spectra.
pub struct SeamFrontEnd {
    dec: MpaDecoder,
    rec: Arc<Mutex<MpaRecord>>,
    nch: usize,
    sr_idx: i32,
    fused: bool,
}

impl SeamFrontEnd {
    pub fn try_new(params: &AudioCodecParameters, opts: &AudioDecoderOptions) -> Result<Self> {
        Self::try_new_at(params, opts, true)
    }

    /// `fused == false`: the first-generation seam (behind requantize + stereo), f32 spectra across PCIe.
    pub fn try_new_at(params: &AudioCodecParameters, _opts: &AudioDecoderOptions, fused: bool) -> Result<Self> {
        let (Some(rate), Some(channels)) = (params.sample_rate, params.channels.clone()) else {
            return unsupported_error("mp3: sample rate and channels are required");
        };
        let Some(sr_idx) = sample_rate_index(rate) else {
            return unsupported_error("mp3: not an MPEG audio sample rate");
        };
        let rec: Arc<Mutex<MpaRecord>> = Arc::new(Mutex::new(MpaRecord::default()));
        // the front end never trims: the trim of a packet is applied to what the device produced (MpaBatch::publish)
        let opts = AudioDecoderOptions { gapless: false, ..Default::default() };
        let dec = MpaDecoder::try_new_with_backend(params, &opts, Box::new(Recorder(rec.clone(), fused)))?;
        Ok(SeamFrontEnd { dec, rec, nch: channels.count(), sr_idx, fused })
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
            rec.quant.clear();
            rec.rq.clear();
            rec.st.clear();
            rec.bad_sample = false;
        }
        self.dec.decode_ref(packet)?;
        let rec = self.rec.lock().expect("mp3 record poisoned");
        if rec.side.is_empty() || rec.side.len() % self.nch != 0 || rec.sample_rate_idx != self.sr_idx {
            // a mid-stream change of the channel mode or the sample rate: the reference fails such packets too
            // (decoder.rs:105-110, "invalid audio buffer signal spec for packet")
            return decode_error("mp3: the frame does not match the stream's channels or sample rate");
        }
        if rec.bad_sample {
            return decode_error("mp3: a spectral sample is not a Huffman table value");
        }
        let fused = if self.fused { Some(FusedMpa { quant: rec.quant.clone(), rq: rec.rq.clone(), st: rec.st.clone() }) } else { None };
        Ok(ParsedMpa {
            trim: (packet.trim_start.get() as usize, packet.trim_end.get() as usize),
            n_granules: rec.side.len() / self.nch,
            xr: rec.xr.clone(),
            side: rec.side.clone(),
            fused,
        })
    }

    fn reset(&mut self) {
        // MpaDecoder::reset (decoder.rs:149-152, through the seam patch): a new Layer3 state -- empty bit reservoir -- around
        // the same backend
        self.dec.reset();
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
    quant: Pinned<i16>,               // the fused form: [channel][granule of the batch][576] Huffman samples,
    rq: Vec<ffi::SymaccelMp3Requant>, //   [channel][granule of the batch],
    st: Vec<ffi::SymaccelMp3Stereo>,  //   [granule of the batch] (the one channel pair of a stereo stream)
    first_granule: Vec<usize>,        // per packet of the batch: index of its first granule; one extra entry = total
    trims: Vec<(usize, usize)>,       // per packet of the batch
    // the cross-stream batcher (SYMACCEL_BATCH_MP3_DECODE, one stream per submission): `cur` holds the batch being handed out -- its PCM
    // is read where the device left it, in the page-locked slot --, `next` the one submitted ahead (its state lands in `overlap` /
    // `vvec` / `vfront` at collect)
    pool: Option<Arc<Pool>>,
    cur: Option<BatchSlot>,
    next: Option<BatchSlot>,
    next_first_granule: Vec<usize>,
    next_trims: Vec<(usize, usize)>,
    gapless: bool,
    buf: AudioBuffer<f32>,
}

impl BatchCodec for MpaBatch {
    type Parsed = ParsedMpa;

    fn parse(&mut self, packet: &PacketRef<'_>) -> Result<ParsedMpa> {
        self.front.parse(packet)
    }

    fn transform(&mut self, batch: &[ParsedMpa]) -> Result<()> {
        // (a batch that came through the batcher is done with: this one is published from `pcm`)
        if let (Some(pool), Some(old)) = (self.pool.clone(), self.cur.take()) {
            pool.release(old);
        }
        self.first_granule.clear();
        self.trims.clear();
        let mut total = 0usize;
        for p in batch {
            self.first_granule.push(total);
            self.trims.push(p.trim);
            total += p.n_granules;
        }
        self.first_granule.push(total);
        // (every packet of a batch comes from the same front end: all fused, or none)
        if batch.iter().all(|p| p.fused.is_some()) && !batch.is_empty() {
            return self.transform_fused(batch, total);
        }
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
        // the batch's PCM: in the batcher's slot (zero-copy: the device wrote it there), or in this decoder's own buffer
        let pcm: &[f32] = match &self.cur {
            Some(slot) => slot.out::<f32>(),
            None => self.pcm.as_slice(),
        };
        for c in 0..self.nch {
            let src = (c * total + g0) * 576;
            if let Some(plane) = self.buf.plane_mut(c) {
                plane[..frames].copy_from_slice(&pcm[src..src + frames]);
            }
        }
        if self.gapless {
            // decoder.rs:128-131
            self.buf.trim(self.trims[i].0, self.trims[i].1);
        }
    }

    fn reset_state(&mut self) {
        // the parse stage: bit reservoir (a seek lands in the middle of a stream whose main data reaches back)
        self.front.reset();
        // Layer3::reset + synthesis state (layer3/mod.rs, synthesis.rs:140-156): overlap, V vector and its front index
        self.overlap.fill(0.0);
        self.vvec.fill(0.0);
        self.vfront.fill(0);
    }

    fn clear(&mut self) {
        self.buf.clear();
    }

    fn pooled(&self) -> bool {
        self.pool.is_some()
    }

    /// The stream's next batch goes to the process-wide batcher (one stream = one submission of 1 or 2 chains): the Huffman samples and
    /// the side records are written straight into a page-locked slot (`Pool::reserve` -> fill -> `commit`); requantize, stereo and the
    /// synthesis tail run in one launch with the other streams' batches (layer3/mod.rs:421-477).
    fn submit(&mut self, batch: &[ParsedMpa]) -> Result<()> {
        let Some(pool) = self.pool.clone() else {
            return unsupported_error("mp3: no batcher");
        };
        if batch.is_empty() || !batch.iter().all(|p| p.fused.is_some()) || self.next.is_some() {
            return unsupported_error("mp3: the batcher takes the entropy decoder's integers, one batch at a time");
        }
        self.next_first_granule.clear();
        self.next_trims.clear();
        let mut total = 0usize;
        for p in batch {
            self.next_first_granule.push(total);
            self.next_trims.push(p.trim);
            total += p.n_granules;
        }
        self.next_first_granule.push(total);
        let nch = self.nch;
        let mut slot = pool.reserve(ffi::SYMACCEL_BATCH_MP3_DECODE as i32, self.front.sample_rate_idx(), nch, total)?;
        {
            // [channel][granule of the batch][576] Huffman samples
            let quant = slot.input::<i16>(0);
            for (i, p) in batch.iter().enumerate() {
                let Some(f) = &p.fused else { continue };
                for g in 0..p.n_granules {
                    for c in 0..nch {
                        let dst = (c * total + self.next_first_granule[i] + g) * 576;
                        let src = (g * nch + c) * 576;
                        quant[dst..dst + 576].copy_from_slice(&f.quant[src..src + 576]);
                    }
                }
            }
        }
        {
            let rq = slot.input::<ffi::SymaccelMp3Requant>(1);
            for (i, p) in batch.iter().enumerate() {
                let Some(f) = &p.fused else { continue };
                for g in 0..p.n_granules {
                    for c in 0..nch {
                        rq[c * total + self.next_first_granule[i] + g] = f.rq[g * nch + c];
                    }
                }
            }
        }
        {
            let side = slot.input::<ffi::SymaccelMp3Side>(2);
            for (i, p) in batch.iter().enumerate() {
                for g in 0..p.n_granules {
                    for c in 0..nch {
                        side[c * total + self.next_first_granule[i] + g] = p.side[g * nch + c];
                    }
                }
            }
        }
        {
            // one joint-stereo record per granule of the STREAM (all zeros for a mono stream)
            let zero_st = ffi::SymaccelMp3Stereo { flags: 0, block_type: 0, is_mixed: 0, reserved: 0, rzero0: 0, rzero1: 0, scalefacs1: [0; 39], pad: 0 };
            let st = slot.input::<ffi::SymaccelMp3Stereo>(3);
            for (i, p) in batch.iter().enumerate() {
                let Some(f) = &p.fused else { continue };
                for g in 0..p.n_granules {
                    st[self.next_first_granule[i] + g] = if nch == 2 { f.st[g] } else { zero_st };
                }
            }
        }
        slot.state::<f32>(0).copy_from_slice(&self.overlap);
        slot.state::<f32>(1).copy_from_slice(&self.vvec);
        slot.state::<i32>(2).copy_from_slice(&self.vfront);
        if let Err(e) = pool.commit(&mut slot) {
            pool.release(slot);
            return Err(e);
        }
        self.next = Some(slot);
        Ok(())
    }

    fn collect(&mut self) -> Result<()> {
        let (Some(pool), Some(mut slot)) = (self.pool.clone(), self.next.take()) else {
            return unsupported_error("mp3: nothing was submitted");
        };
        if let Err(e) = pool.wait(&mut slot) {
            pool.release(slot);
            return Err(e);
        }
        // the state after the batch; the PCM stays where it is
        self.overlap.copy_from_slice(slot.state::<f32>(0));
        self.vvec.copy_from_slice(slot.state::<f32>(1));
        self.vfront.copy_from_slice(slot.state::<i32>(2));
        if let Some(old) = self.cur.take() {
            pool.release(old);
        }
        self.cur = Some(slot);
        std::mem::swap(&mut self.first_granule, &mut self.next_first_granule);
        std::mem::swap(&mut self.trims, &mut self.next_trims);
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

impl Drop for MpaBatch {
    fn drop(&mut self) {
        BatchCodec::abandon(self);
        if let (Some(pool), Some(slot)) = (self.pool.clone(), self.cur.take()) {
            pool.release(slot);
        }
    }
}

impl MpaBatch {
    /// layer3/mod.rs:421-477 for the whole batch in one call: requantize, stereo and the synthesis tail on the device from the
    /// entropy decoder's integers (2 bytes per line in, 4 bytes per sample out).
    /// The fused batch in the chain-major staging layout (`first`: index of every packet's first granule).
    fn gather_fused(&mut self, batch: &[ParsedMpa], total: usize, first: &[usize]) {
        let zero_st = ffi::SymaccelMp3Stereo { flags: 0, block_type: 0, is_mixed: 0, reserved: 0, rzero0: 0, rzero1: 0, scalefacs1: [0; 39], pad: 0 };
        for (i, p) in batch.iter().enumerate() {
            let Some(f) = &p.fused else { continue };
            for g in 0..p.n_granules {
                let at = first[i] + g;
                for c in 0..self.nch {
                    let dst = c * total + at;
                    let src = (g * self.nch + c) * 576;
                    self.quant.as_mut_slice()[dst * 576..(dst + 1) * 576].copy_from_slice(&f.quant[src..src + 576]);
                    self.rq[dst] = f.rq[g * self.nch + c];
                    self.side[dst] = p.side[g * self.nch + c];
                }
                self.st[at] = if self.nch == 2 { f.st[g] } else { zero_st };
            }
        }
    }

    fn transform_fused(&mut self, batch: &[ParsedMpa], total: usize) -> Result<()> {
        let first = self.first_granule.clone();
        self.gather_fused(batch, total, &first);
        let pair: [i32; 2] = [0, 1];
        let n_pairs = if self.nch == 2 { 1 } else { 0 };
        // SAFETY: every buffer covers nch * total (* 576) elements and st covers `total` records (sized for max_batch frames of
        // two granules); pair names the two chains of the batch; the call returns after the PCM and the updated state are back in
        // host memory.
        check(
            unsafe {
                ffi::symaccel_mp3_decode_pipelined(
                    self.ctx.raw(),
                    self.quant.as_slice().as_ptr(),
                    self.rq.as_ptr(),
                    pair.as_ptr(),
                    self.st.as_ptr(),
                    n_pairs,
                    self.side.as_ptr(),
                    self.front.sample_rate_idx(),
                    self.overlap.as_mut_ptr(),
                    self.vvec.as_mut_ptr(),
                    self.vfront.as_mut_ptr(),
                    self.pcm.as_mut_slice().as_mut_ptr(),
                    self.nch,
                    total,
                    0,
                )
            },
            self.ctx.raw(),
        )
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
        Self::try_new_with_pool(params, opts, front, max_batch, None)
    }

    /// The same decoder submitting to the process-wide cross-stream batcher (`Pool::shared()`): with many streams open, the
    /// batches of all of them go to the device in one launch (csrc/batcher.cpp).
    pub fn try_new_pooled(params: &AudioCodecParameters, opts: &AudioDecoderOptions, front: Box<dyn MpaFrontEnd>, max_batch: usize) -> Result<Self> {
        Self::try_new_with_pool(params, opts, front, max_batch, Some(Pool::shared()?))
    }

    pub fn try_new_with_pool(
        params: &AudioCodecParameters,
        opts: &AudioDecoderOptions,
        front: Box<dyn MpaFrontEnd>,
        max_batch: usize,
        pool: Option<Arc<Pool>>,
    ) -> Result<Self> {
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
                quant: Pinned::new(nch * granules * 576)?,
                rq: vec![
                    ffi::SymaccelMp3Requant { global_gain: 0, flags: 0, block_type: 0, is_mixed: 0, subblock_gain: [0; 3], reserved: 0, rzero: 0, scalefacs: [0; 39], pad: [0; 3] };
                    nch * granules
                ],
                st: vec![ffi::SymaccelMp3Stereo { flags: 0, block_type: 0, is_mixed: 0, reserved: 0, rzero0: 0, rzero1: 0, scalefacs1: [0; 39], pad: 0 }; granules],
                first_granule: Vec::with_capacity(max_batch + 1),
                trims: Vec::with_capacity(max_batch),
                pool,
                cur: None,
                next: None,
                next_first_granule: Vec::with_capacity(max_batch + 1),
                next_trims: Vec::with_capacity(max_batch),
                gapless: opts.gapless,
                buf: AudioBuffer::new(AudioSpec::new(rate, channels), 1152),
            },
            la: Lookahead::new(max_batch),
        })
    }
}
