//! `HipVorbisDecoder`: everything behind the packet decoder on the MI355X -- inverse coupling, floor-1 curve synthesis and the dot
//! product (symphonia-codec-vorbis/src/lib.rs:250-292, floor.rs:568-653), then the per-channel synthesis of every audio packet:
//! Imdct, windowing and the lapped overlap-add of `DspChannel::synth` (dsp.rs:68-145, called from lib.rs:296-331).  The front end
//! hands over residue vectors + floor posts + the coupling steps of the packet (`symaccel_vorbis_decode`); a front end that delivers
//! finished spectra (`ParsedVorbis::fused == None`; any stream with a floor of type 0) takes `symaccel_vorbis_synth`.
use std::sync::{Arc, Mutex};

use symphonia_codec_vorbis::backend::{CodedChannel, Floor1Config, SynthBackend};
use symphonia_codec_vorbis::VorbisDecoder;
use symphonia_core::audio::{Audio, AudioBuffer, AudioMut, AudioSpec, GenericAudioBufferRef};
use symphonia_core::codecs::audio::well_known::CODEC_ID_VORBIS;
use symphonia_core::codecs::audio::{AudioCodecParameters, AudioDecoder, AudioDecoderOptions};
use symphonia_core::errors::{decode_error, unsupported_error, Result};
use symphonia_core::packet::PacketRef;
use symphonia_core::support_audio_codec;

use crate::ctx::{check, BatchSlot, Context, Pinned, Pool};
use crate::decoder::DecoderBatch;
use crate::ffi;
use crate::lookahead::{BatchCodec, Lookahead};

/// One audio packet after the CPU front end (mode / window flags, floor decode, residue decode, inverse coupling and the
/// floor x residue product: lib.rs:186-292): per channel the n/2 spectral lines `DspChannel::synth` receives.
/// Channels whose floor is unused ("do not decode") carry zeros, as `synth` sees them (dsp.rs:72-75).
pub struct ParsedVorbis {
    pub trim: (usize, usize), // frames to trim from the start / end of the decoded packet (lib.rs:333-342; the decoder is gapless)
    pub long_block: bool,  // block_flag of the packet's mode (lib.rs:203-214)
    pub spectra: Vec<f32>, // [channel][n / 2], n = the block size the flag selects; with `fused`: the RESIDUE vectors
    pub fused: Option<FusedVorbis>,
}

/// What is still to be done to `ParsedVorbis::spectra` when they are the packet decoder's residue vectors (lib.rs:230-248; zeros
/// for a do-not-decode channel): the inverse coupling steps, the floor curves and the dot product (lib.rs:250-292).
pub struct FusedVorbis {
    pub floor: Vec<u8>,    // [channel]: index into `VorbisFrontEnd::floors`, or SYMACCEL_VORBIS_FLOOR_UNUSED
    pub posts: Vec<u32>,   // [channel][POSTS]: floor1_Y as read from the packet
    pub coupling: Vec<u8>, // [(magnitude channel, angle channel)], in the mapping's order
}

/// Posts per channel-block in the batch arrays (floor1_values <= 65: floor.rs:510-520).
pub const POSTS: usize = 65;

pub trait VorbisFrontEnd: Send + Sync {
    fn channels(&self) -> usize;
    /// (bs0_exp, bs1_exp) of the identification header (lib.rs:404-406)
    fn block_exps(&self) -> (i32, i32);
    fn parse(&mut self, packet: &PacketRef<'_>) -> Result<ParsedVorbis>;
    /// The floor-1 configurations `FusedVorbis::floor` indexes (empty: this front end never returns `fused`).
    fn floors(&self) -> Vec<ffi::SymaccelVorbisFloor1Cfg>;
    /// `AudioDecoder::reset`: nothing in the Vorbis parse stage outlives a packet (the lapping state is the batch's).
    fn reset(&mut self);
}

/// What the reference's decoder hands its `SynthBackend` (bindings/rust/patches/symphonia-codec-vorbis.diff).
#[derive(Default)]
pub struct VorbisRecord {
    pub n_channels: usize,
    pub bs0_exp: i32,
    pub bs1_exp: i32,
    pub long_block: bool,
    pub spectra: Vec<f32>, // [channel][n / 2]
    pub seen: usize,
    pub floors: Vec<ffi::SymaccelVorbisFloor1Cfg>, // the fused form (configure_floors, synth_block): see FusedVorbis
    pub floor: Vec<u8>,
    pub posts: Vec<u32>,
    pub coupling: Vec<u8>,
}

/// The `SynthBackend` handed to the reference's `VorbisDecoder`: every channel's floor x residue spectrum is recorded,
/// nothing is synthesized.
pub struct Recorder(pub Arc<Mutex<VorbisRecord>>, pub bool);

impl SynthBackend for Recorder {
    fn configure(&mut self, n_channels: usize, bs0_exp: u8, bs1_exp: u8) {
        let mut rec = self.0.lock().expect("vorbis record poisoned");
        rec.n_channels = n_channels;
        rec.bs0_exp = bs0_exp as i32;
        rec.bs1_exp = bs1_exp as i32;
    }

    fn synth(&mut self, channel: usize, block_flag: bool, _prev_block_flag: bool, spectrum: &[f32], _out: &mut [f32]) {
        let mut rec = self.0.lock().expect("vorbis record poisoned");
        let half = spectrum.len();
        if rec.spectra.len() != rec.n_channels * half {
            rec.spectra.clear();
            rec.spectra.resize(rec.n_channels * half, 0.0);
        }
        rec.long_block = block_flag;
        rec.spectra[channel * half..(channel + 1) * half].copy_from_slice(spectrum);
        rec.seen += 1;
    }

    fn reset(&mut self) {}

    /// `Recorder(_, true)`: take the blocks in front of the inverse coupling (the second seam of the patch).
    fn takes_blocks(&self) -> bool {
        self.1
    }

    fn configure_floors(&mut self, floors: &[Option<Floor1Config>]) {
        let mut rec = self.0.lock().expect("vorbis record poisoned");
        rec.floors.clear();
        for f in floors.iter() {
            let mut cfg = ffi::SymaccelVorbisFloor1Cfg { multiplier: 1, n_posts: 0, pad: [0; 2], x_list: [0; 65] };
            if let Some(f) = f {
                cfg.multiplier = f.multiplier;
                cfg.n_posts = f.x_list.len().min(POSTS) as u8;
                for (dst, x) in cfg.x_list.iter_mut().zip(f.x_list.iter()) {
                    *dst = *x;
                }
            }
            rec.floors.push(cfg);
        }
    }

    fn synth_block(&mut self, block_flag: bool, _prev_block_flag: bool, couplings: &[(u8, u8)], channels: &[CodedChannel<'_>], _out: &mut AudioBuffer<f32>) {
        let mut rec = self.0.lock().expect("vorbis record poisoned");
        let nch = rec.n_channels;
        let half = channels.first().map(|c| c.residue.len()).unwrap_or(0);
        rec.long_block = block_flag;
        rec.spectra.clear();
        rec.spectra.resize(nch * half, 0.0);
        rec.floor.clear();
        rec.floor.resize(nch, ffi::SYMACCEL_VORBIS_FLOOR_UNUSED as u8);
        rec.posts.clear();
        rec.posts.resize(nch * POSTS, 0);
        rec.coupling.clear();
        for ch in channels.iter() {
            let c = ch.plane;
            if c >= nch {
                continue;
            }
            // a do-not-decode channel's residue buffer is stale (residue.rs:253-259 returns before touching it): its spectrum is 0
            if !ch.do_not_decode {
                rec.spectra[c * half..(c + 1) * half].copy_from_slice(ch.residue);
            }
            if let Some((index, posts)) = ch.floor {
                rec.floor[c] = index as u8;
                for (dst, y) in rec.posts[c * POSTS..(c + 1) * POSTS].iter_mut().zip(posts.iter()) {
                    *dst = *y;
                }
            }
            rec.seen += 1;
        }
        // the steps name channels by their position in `channels`: the batch keeps channels by audio plane
        for (m, a) in couplings.iter() {
            if let (Some(mc), Some(ac)) = (channels.get(*m as usize), channels.get(*a as usize)) {
                rec.coupling.push(mc.plane as u8);
                rec.coupling.push(ac.plane as u8);
            }
        }
    }
}

/// `VorbisFrontEnd` over the reference's own decoder with the recording backend installed: setup headers, codebooks, floor
/// and residue decoding are symphonia-codec-vorbis's code, unmodified.  `fused` (the default, for streams whose floors are all of
/// type 1) stops there -- the recorder takes the blocks in front of the inverse coupling and `parse` returns residue vectors + posts
/// + coupling steps (`ParsedVorbis::fused`); without it the reference's coupling, floor synthesis and dot product run too.
pub struct SeamFrontEnd {
    dec: VorbisDecoder,
    rec: Arc<Mutex<VorbisRecord>>,
}

impl SeamFrontEnd {
    pub fn try_new(params: &AudioCodecParameters, opts: &AudioDecoderOptions) -> Result<Self> {
        Self::try_new_at(params, opts, true)
    }

    /// `fused == false`: the first-generation seam (behind the dot product).
    pub fn try_new_at(params: &AudioCodecParameters, _opts: &AudioDecoderOptions, fused: bool) -> Result<Self> {
        let rec: Arc<Mutex<VorbisRecord>> = Arc::new(Mutex::new(VorbisRecord::default()));
        // the front end never silences or trims: both are applied to what the device produced (VorbisBatch::publish)
        let opts = AudioDecoderOptions { gapless: false, ..Default::default() };
        let dec = VorbisDecoder::try_new_with_backend(params, &opts, Box::new(Recorder(rec.clone(), fused)))?;
        Ok(SeamFrontEnd { dec, rec })
    }
}

impl VorbisFrontEnd for SeamFrontEnd {
    fn channels(&self) -> usize {
        self.rec.lock().expect("vorbis record poisoned").n_channels
    }

    fn block_exps(&self) -> (i32, i32) {
        let rec = self.rec.lock().expect("vorbis record poisoned");
        (rec.bs0_exp, rec.bs1_exp)
    }

    fn parse(&mut self, packet: &PacketRef<'_>) -> Result<ParsedVorbis> {
        self.rec.lock().expect("vorbis record poisoned").seen = 0;
        self.dec.decode_ref(packet)?;
        let rec = self.rec.lock().expect("vorbis record poisoned");
        if rec.seen != rec.n_channels {
            return decode_error("vorbis: the packet did not reach the synthesis stage for every channel");
        }
        // (the decoder hands blocks over only if the recorder takes them AND every floor is of type 1: then `floors` is filled)
        let fused = if rec.floors.is_empty() {
            None
        }
        else {
            Some(FusedVorbis { floor: rec.floor.clone(), posts: rec.posts.clone(), coupling: rec.coupling.clone() })
        };
        Ok(ParsedVorbis {
            trim: (packet.trim_start.get() as usize, packet.trim_end.get() as usize),
            long_block: rec.long_block,
            spectra: rec.spectra.clone(),
            fused,
        })
    }

    fn floors(&self) -> Vec<ffi::SymaccelVorbisFloor1Cfg> {
        self.rec.lock().expect("vorbis record poisoned").floors.clone()
    }

    fn reset(&mut self) {
        // VorbisDecoder::reset (lib.rs:367-374): Dsp::reset -- the reference's own lapping state, which the recording backend
        // never reads -- kept in step anyway
        self.dec.reset();
    }
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
    emits: Vec<bool>,         // per packet: false for the first block after a reset (lib.rs:335-338: silenced when gapless)
    trims: Vec<(usize, usize)>, // per packet of the batch
    floors: Vec<ffi::SymaccelVorbisFloor1Cfg>, // the fused form: the stream's floor-1 configurations,
    floor: Vec<u8>,           //   [channel][packet] floor index or SYMACCEL_VORBIS_FLOOR_UNUSED,
    posts: Vec<u32>,          //   [channel][packet][POSTS]
    // the cross-stream batcher (SYMACCEL_BATCH_VORBIS_DECODE for a front end that hands on residue + posts, SYMACCEL_BATCH_VORBIS_SYNTH for
    // one that hands on spectra): every chain's planes at their largest -- max_batch x bs1 / 2 --, the packed data at the front, so that
    // streams with different block flags share a launch.  `cur` holds the batch being handed out (its PCM is read where the device left
    // it), `next` the one submitted ahead; `floor_index` = what the batcher calls this stream's floor configurations.
    pool: Option<Arc<Pool>>,
    cur: Option<BatchSlot>,
    next: Option<BatchSlot>,
    floor_index: Vec<u8>,
    next_pcm_off: Vec<usize>,
    next_emits: Vec<bool>,
    next_trims: Vec<(usize, usize)>,
    next_stride: usize,
    buf: AudioBuffer<f32>,
}

impl BatchCodec for VorbisBatch {
    type Parsed = ParsedVorbis;

    fn parse(&mut self, packet: &PacketRef<'_>) -> Result<ParsedVorbis> {
        self.front.parse(packet)
    }

    fn transform(&mut self, batch: &[ParsedVorbis]) -> Result<()> {
        // (a batch that came through the batcher is done with: this one is published from `pcm`)
        if let (Some(pool), Some(old)) = (self.pool.clone(), self.cur.take()) {
            pool.release(old);
        }
        let k = batch.len();
        // the packed layout of include/symaccel.h ("Vorbis"): block b owns n_b / 2 lines and (prev_n + n_b) / 4 samples;
        // a first block without a previous one owns n_b / 2 sample slots and leaves them untouched
        let mut prev: i32 = self.prev_flag[0];
        let (mut lines, mut samples) = (0usize, 0usize);
        let mut spec_off = Vec::with_capacity(k);
        self.pcm_off.clear();
        self.emits.clear();
        self.trims.clear();
        for p in batch {
            self.trims.push(p.trim);
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
        if k > 0 && batch.iter().all(|p| p.fused.is_some()) {
            return self.transform_fused(batch, lines, samples);
        }
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
        // the batch's PCM: in the batcher's slot (zero-copy: the device wrote it there), or in this decoder's own buffer
        let pcm: &[f32] = match &self.cur {
            Some(slot) => slot.out::<f32>(),
            None => self.pcm.as_slice(),
        };
        for c in 0..self.nch {
            let src = c * self.pcm_stride + self.pcm_off[i];
            if let Some(plane) = self.buf.plane_mut(c) {
                plane[..frames].copy_from_slice(&pcm[src..src + frames]);
            }
        }
        // lib.rs:333-342 (gapless): the first packet after a reset is silenced (`emits`), every other one is trimmed
        if self.emits[i] {
            self.buf.trim(self.trims[i].0, self.trims[i].1);
        }
    }

    fn reset_state(&mut self) {
        self.front.reset();
        // Dsp::reset (dsp.rs:45-56): lapping state empty, overlap zeroed
        self.prev_flag.fill(-1);
        self.overlap.fill(0.0);
    }

    fn clear(&mut self) {
        self.buf.clear();
    }

    fn pooled(&self) -> bool {
        self.pool.is_some()
    }

    /// The stream's next batch goes to the process-wide batcher: residue (or spectra), flags, floor indices, posts and the coupling steps
    /// are written straight into a page-locked slot; coupling, floor curves, dot product and synthesis run in one launch with the other
    /// streams' batches (lib.rs:250-331).
    fn submit(&mut self, batch: &[ParsedVorbis]) -> Result<()> {
        let Some(pool) = self.pool.clone() else {
            return unsupported_error("vorbis: no batcher");
        };
        if batch.is_empty() || self.next.is_some() {
            return unsupported_error("vorbis: one batch at a time");
        }
        let (k, nch) = (batch.len(), self.nch);
        let fused = batch.iter().all(|p| p.fused.is_some());
        if !fused && batch.iter().any(|p| p.fused.is_some()) {
            return unsupported_error("vorbis: a batch is all residue + posts, or all spectra");
        }
        let (bs0_exp, bs1_exp) = self.front.block_exps();
        let cap = k * self.bs[1] / 2; // a chain's spectrum / PCM plane at its largest
        // the packed layout of the batch inside a chain (as `transform`)
        let mut prev: i32 = self.prev_flag[0];
        let (mut lines, mut samples) = (0usize, 0usize);
        let mut spec_off = Vec::with_capacity(k);
        self.next_pcm_off.clear();
        self.next_emits.clear();
        self.next_trims.clear();
        for p in batch {
            let n = self.bs[p.long_block as usize];
            spec_off.push(lines);
            self.next_pcm_off.push(samples);
            self.next_emits.push(prev >= 0);
            self.next_trims.push(p.trim);
            lines += n / 2;
            samples += if prev >= 0 { (self.bs[prev as usize] + n) / 4 } else { n / 2 };
            prev = p.long_block as i32;
        }
        self.next_pcm_off.push(samples);
        self.next_stride = cap;
        let (kind, param) = if fused {
            (ffi::SYMACCEL_BATCH_VORBIS_DECODE as i32, bs0_exp | (bs1_exp << 8) | ((nch as i32) << 16))
        }
        else {
            (ffi::SYMACCEL_BATCH_VORBIS_SYNTH as i32, bs0_exp | (bs1_exp << 8))
        };
        let mut slot = pool.reserve(kind, param, nch, k)?;
        {
            let spectra = slot.input::<f32>(0);
            for (i, p) in batch.iter().enumerate() {
                let half = self.bs[p.long_block as usize] / 2;
                for c in 0..nch {
                    let dst = c * cap + spec_off[i];
                    spectra[dst..dst + half].copy_from_slice(&p.spectra[c * half..(c + 1) * half]);
                }
            }
        }
        {
            let flags = slot.input::<u8>(1);
            for (i, p) in batch.iter().enumerate() {
                for c in 0..nch {
                    flags[c * k + i] = p.long_block as u8;
                }
            }
        }
        if fused {
            let mut steps: Vec<u8> = Vec::new();
            let mut first: Vec<u32> = Vec::with_capacity(k + 1);
            first.push(0);
            {
                let floor = slot.input::<u8>(2);
                for (i, p) in batch.iter().enumerate() {
                    let Some(f) = &p.fused else { continue };
                    for c in 0..nch {
                        // (the stream's floor numbers become the batcher's: registered when the decoder was built)
                        let local = f.floor[c] as usize;
                        floor[c * k + i] = if local < self.floor_index.len() { self.floor_index[local] } else { ffi::SYMACCEL_VORBIS_FLOOR_UNUSED as u8 };
                    }
                    steps.extend_from_slice(&f.coupling);
                    first.push((steps.len() / 2) as u32);
                }
            }
            {
                let posts = slot.input::<u32>(3);
                for (i, p) in batch.iter().enumerate() {
                    let Some(f) = &p.fused else { continue };
                    for c in 0..nch {
                        let dst = (c * k + i) * POSTS;
                        posts[dst..dst + POSTS].copy_from_slice(&f.posts[c * POSTS..(c + 1) * POSTS]);
                    }
                }
            }
            {
                // the coupling blob: first[k + 1] u32 (little endian), padded to 16 bytes, then the (magnitude, angle) pairs
                let blob = slot.input::<u8>(4);
                let steps_at = ((k + 1) * 4 + 15) & !15;
                if steps_at + steps.len() > blob.len() {
                    pool.release(slot);
                    return unsupported_error("vorbis: more coupling steps than a batcher submission holds");
                }
                for (b, v) in first.iter().enumerate() {
                    blob[4 * b..4 * b + 4].copy_from_slice(&v.to_le_bytes());
                }
                blob[steps_at..steps_at + steps.len()].copy_from_slice(&steps);
            }
        }
        slot.state::<i32>(0).copy_from_slice(&self.prev_flag);
        slot.state::<f32>(1).copy_from_slice(&self.overlap);
        if let Err(e) = pool.commit(&mut slot) {
            pool.release(slot);
            return Err(e);
        }
        self.next = Some(slot);
        Ok(())
    }

    fn collect(&mut self) -> Result<()> {
        let (Some(pool), Some(mut slot)) = (self.pool.clone(), self.next.take()) else {
            return unsupported_error("vorbis: nothing was submitted");
        };
        if let Err(e) = pool.wait(&mut slot) {
            pool.release(slot);
            return Err(e);
        }
        // the lapping state after the batch; the PCM stays where it is
        self.prev_flag.copy_from_slice(slot.state::<i32>(0));
        self.overlap.copy_from_slice(slot.state::<f32>(1));
        if let Some(old) = self.cur.take() {
            pool.release(old);
        }
        self.cur = Some(slot);
        self.pcm_stride = self.next_stride;
        std::mem::swap(&mut self.pcm_off, &mut self.next_pcm_off);
        std::mem::swap(&mut self.emits, &mut self.next_emits);
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

impl Drop for VorbisBatch {
    fn drop(&mut self) {
        BatchCodec::abandon(self);
        if let (Some(pool), Some(slot)) = (self.pool.clone(), self.cur.take()) {
            pool.release(slot);
        }
    }
}

impl VorbisBatch {
    /// lib.rs:250-331 for the whole batch in one call: inverse coupling, floor curves, dot product and synthesis on the device from
    /// the packet decoder's residue vectors (already packed into `spectra` by `transform`) and posts.
    fn transform_fused(&mut self, batch: &[ParsedVorbis], lines: usize, samples: usize) -> Result<()> {
        let k = batch.len();
        let mut coupling: Vec<u8> = Vec::new();
        let mut first: Vec<u32> = Vec::with_capacity(k + 1);
        first.push(0);
        for (i, p) in batch.iter().enumerate() {
            let Some(f) = &p.fused else { continue };
            for c in 0..self.nch {
                self.floor[c * k + i] = f.floor[c];
                let dst = (c * k + i) * POSTS;
                self.posts[dst..dst + POSTS].copy_from_slice(&f.posts[c * POSTS..(c + 1) * POSTS]);
            }
            coupling.extend_from_slice(&f.coupling);
            first.push((coupling.len() / 2) as u32);
        }
        let (bs0_exp, bs1_exp) = self.front.block_exps();
        // SAFETY: residue / flags / floor / posts / pcm cover nch chains of `lines` / k / k / k * POSTS / `samples` elements (sized for
        // max_batch long blocks), `first` k + 1 entries, `coupling` two bytes per step; null stands for an empty step list.  The
        // call returns after the PCM and the updated state are back in host memory.
        check(
            unsafe {
                ffi::symaccel_vorbis_decode(
                    self.ctx.raw(),
                    bs0_exp,
                    bs1_exp,
                    self.spectra.as_slice().as_ptr(),
                    lines,
                    self.flags.as_ptr(),
                    self.floor.as_ptr(),
                    self.posts.as_ptr(),
                    POSTS,
                    self.floors.as_ptr(),
                    self.floors.len(),
                    self.nch,
                    if coupling.is_empty() { std::ptr::null() } else { coupling.as_ptr() },
                    first.as_ptr(),
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
    pub fn try_new(params: &AudioCodecParameters, opts: &AudioDecoderOptions, front: Box<dyn VorbisFrontEnd>, max_batch: usize) -> Result<Self> {
        Self::try_new_with_pool(params, opts, front, max_batch, None)
    }

    /// The same decoder submitting to the process-wide cross-stream batcher (`Pool::shared()`): the batches of every open Vorbis stream
    /// with these block sizes and this channel count go to the device in one launch (csrc/batcher.cpp, SYMACCEL_BATCH_VORBIS_DECODE).
    pub fn try_new_pooled(params: &AudioCodecParameters, opts: &AudioDecoderOptions, front: Box<dyn VorbisFrontEnd>, max_batch: usize) -> Result<Self> {
        Self::try_new_with_pool(params, opts, front, max_batch, Some(Pool::shared()?))
    }

    pub fn try_new_with_pool(
        params: &AudioCodecParameters,
        opts: &AudioDecoderOptions,
        front: Box<dyn VorbisFrontEnd>,
        max_batch: usize,
        pool: Option<Arc<Pool>>,
    ) -> Result<Self> {
        if !opts.gapless {
            // Without gapless support the reference returns the first block after a reset windowed against silence
            // (lib.rs:316-331 with an all-zero overlap); the batched call does not compute that half block.  The decoder
            // registered below this one takes such tracks (fallback.rs).
            return unsupported_error("vorbis: the batched decoder implements the gapless (default) behaviour only");
        }
        let (Some(rate), Some(channels)) = (params.sample_rate, params.channels.clone()) else {
            return unsupported_error("vorbis: sample rate and channels are required");
        };
        let nch = front.channels();
        let max_batch = max_batch.max(1);
        let (bs0_exp, bs1_exp) = front.block_exps();
        let bs = [1usize << bs0_exp, 1usize << bs1_exp];
        let floors = front.floors();
        // the stream's floor-1 configurations under the batcher's numbers (a setup entry that is no floor 1 has no posts to render: unused);
        // a batcher that has no room for them (255 distinct configurations) leaves this stream batching on its own
        let mut pool = pool;
        let mut floor_index: Vec<u8> = Vec::with_capacity(floors.len());
        if let Some(p) = pool.clone() {
            for cfg in floors.iter() {
                if cfg.n_posts < 2 {
                    floor_index.push(ffi::SYMACCEL_VORBIS_FLOOR_UNUSED as u8);
                    continue;
                }
                match p.vorbis_floor(cfg) {
                    Ok(index) => floor_index.push(index),
                    Err(_) => {
                        pool = None;
                        break;
                    }
                }
            }
        }
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
                trims: Vec::with_capacity(max_batch),
                floors,
                floor: vec![0; nch * max_batch],
                posts: vec![0; nch * max_batch * POSTS],
                pool,
                cur: None,
                next: None,
                floor_index,
                next_pcm_off: Vec::with_capacity(max_batch + 1),
                next_emits: Vec::with_capacity(max_batch),
                next_trims: Vec::with_capacity(max_batch),
                next_stride: 0,
                buf: AudioBuffer::new(AudioSpec::new(rate, channels), bs[1] / 2),
            },
            la: Lookahead::new(max_batch),
        })
    }
}
