//! Look-ahead batching behind the `AudioDecoder` method set (the Rust twin of `codecs::LookaheadDecoder` in
//! include/symaccel.hpp; tests/test_rust_shim.py executes THIS file under tools/rsinterp with a mock codec and a mock
//! demuxer, on the packet script of tests/cpp/lookahead_test.cpp).
//!
//! `AudioDecoder::decode_ref(&mut self, &PacketRef)` (codecs/audio.rs:281-285) sees one packet; a GPU wants thousands of
//! frames per call.  The bridge is a reader-side look-ahead: `LookaheadReader` wraps the application's `FormatReader`
//! (it IS a `FormatReader`, so the application keeps calling `next_packet` / `seek` on a `Box<dyn FormatReader>`), reads
//! `depth` packets ahead and keeps the packets of each track in a queue the track's decoder can see.  When `decode_ref(p)`
//! finds nothing pre-computed for `p`, the decoder parses `p` and the queued packets that follow it (CPU, the reference's
//! own parser), transforms all of them in ONE batch call, and serves the following `decode_ref` calls from the result.
//! Without a `LookaheadReader` there is no queue and every call is a batch of one: correct, slow.
//!
//! Decoders are created by the registry from `(params, opts)` alone (codecs/registry.rs:34-44), so a decoder cannot be
//! handed its reader.  It finds it: the reader records the identity of the packet it returned last for each track
//! (track id, pts, address and length of the payload), and a decoder that is asked to decode a packet looks for the
//! live reader that handed out exactly that packet.  Two open containers with equal track ids therefore never share
//! a queue; an application that clones packets before decoding simply gets batches of one.
use std::collections::{HashMap, VecDeque};
use std::sync::{Arc, Mutex, Weak};

use symphonia_core::errors::{unsupported_error, Result};
use symphonia_core::formats::{Attachment, FormatInfo, FormatReader, MediaInfo, SeekMode, SeekTo, SeekedTo, Track};
use symphonia_core::io::MediaSourceStream;
use symphonia_core::meta::{ChapterGroup, Metadata};
use symphonia_core::packet::{Packet, PacketRef};

/// Identity of a packet in the hands of the application.
#[derive(Clone, Copy, PartialEq, Eq, Debug)]
pub struct PacketKey {
    pub track_id: u32,
    pub pts: i64,
    pub data: usize,
    pub len: usize,
}

impl PacketKey {
    pub fn of(packet: &PacketRef<'_>) -> PacketKey {
        PacketKey { track_id: packet.track_id, pts: packet.pts.get(), data: packet.data.as_ptr() as usize, len: packet.data.len() }
    }
}

/// Packets of one track the demuxer has already read but the application has not yet been given.
#[derive(Default)]
pub struct TrackQueue {
    /// oldest first; the front is what `next_packet` returns next for this track
    pub packets: VecDeque<Packet>,
    /// the packet `next_packet` returned last for this track
    pub last_out: Option<PacketKey>,
}

/// What one `LookaheadReader` shares with the decoders of its tracks.
#[derive(Default)]
pub struct Shared {
    pub tracks: HashMap<u32, TrackQueue>,
}

pub type SharedHandle = Arc<Mutex<Shared>>;

/// Every live reader (weak: a dropped reader disappears from the list at the next lookup).
static READERS: Mutex<Vec<Weak<Mutex<Shared>>>> = Mutex::new(Vec::new());

/// The live reader that handed out the packet `key`, if any.
pub fn find_reader(key: &PacketKey) -> Option<SharedHandle> {
    let mut readers = READERS.lock().expect("reader list poisoned");
    readers.retain(|w| w.strong_count() > 0);
    for w in readers.iter() {
        if let Some(handle) = w.upgrade() {
            let hit = match handle.lock().expect("look-ahead state poisoned").tracks.get(&key.track_id) {
                Some(q) => q.last_out == Some(*key),
                None => false,
            };
            if hit {
                return Some(handle);
            }
        }
    }
    None
}

/// A `FormatReader` that stays `depth` packets ahead of what it returns.
pub struct LookaheadReader {
    inner: Box<dyn FormatReader>,
    depth: usize,
    /// everything read ahead, in container order (clones of what sits in the track queues)
    pending: VecDeque<Packet>,
    shared: SharedHandle,
    eof: bool,
    /// an error of the inner reader that occurred while reading AHEAD: returned once every packet read before it has been
    /// handed out, exactly where the inner reader would have returned it
    deferred: Option<symphonia_core::errors::Error>,
}

impl LookaheadReader {
    pub fn new(inner: Box<dyn FormatReader>, depth: usize) -> Self {
        let shared: SharedHandle = Arc::new(Mutex::new(Shared::default()));
        READERS.lock().expect("reader list poisoned").push(Arc::downgrade(&shared));
        LookaheadReader { inner, depth: depth.max(1), pending: VecDeque::new(), shared, eof: false, deferred: None }
    }

    /// The state the decoders of this reader's tracks look at (tests; an application never needs it).
    pub fn shared(&self) -> SharedHandle {
        self.shared.clone()
    }

    fn refill(&mut self) {
        while !self.eof && self.deferred.is_none() && self.pending.len() < self.depth {
            match self.inner.next_packet() {
                Ok(Some(p)) => {
                    let mut shared = self.shared.lock().expect("look-ahead state poisoned");
                    shared.tracks.entry(p.track_id).or_default().packets.push_back(p.clone());
                    drop(shared);
                    self.pending.push_back(p);
                }
                Ok(None) => self.eof = true,
                // reading stops here; the packets already read are still delivered, then the error (next_packet)
                Err(e) => self.deferred = Some(e),
            }
        }
    }

    fn drop_lookahead(&mut self) {
        self.pending.clear();
        let mut shared = self.shared.lock().expect("look-ahead state poisoned");
        for q in shared.tracks.values_mut() {
            q.packets.clear();
            q.last_out = None;
        }
        drop(shared);
        self.eof = false;
        self.deferred = None;
    }
}

impl FormatReader for LookaheadReader {
    fn format_info(&self) -> &FormatInfo {
        self.inner.format_info()
    }

    fn media_info(&self) -> &MediaInfo {
        self.inner.media_info()
    }

    fn attachments(&self) -> &[Attachment] {
        self.inner.attachments()
    }

    fn chapters(&self) -> Option<&ChapterGroup> {
        self.inner.chapters()
    }

    fn metadata(&mut self) -> Metadata<'_> {
        self.inner.metadata()
    }

    /// formats/mod.rs:591: everything read ahead is stale.  The application resets the decoders afterwards, as the
    /// reference requires (codecs/audio.rs:252-257).
    fn seek(&mut self, mode: SeekMode, to: SeekTo) -> Result<SeekedTo> {
        self.drop_lookahead();
        self.inner.seek(mode, to)
    }

    fn tracks(&self) -> &[Track] {
        self.inner.tracks()
    }

    /// formats/mod.rs:646: the oldest packet read ahead.  It leaves its track's queue (the queue holds what FOLLOWS the
    /// packet the application is about to decode) and becomes the track's `last_out`.  An error of the inner reader is
    /// returned where the inner reader would have returned it: after every packet that was read before it (a damaged file
    /// still plays up to the damage); the call after that reads on, like a second call on the inner reader would.
    fn next_packet(&mut self) -> Result<Option<Packet>> {
        self.refill();
        if self.pending.is_empty() {
            if let Some(e) = self.deferred.take() {
                return Err(e);
            }
        }
        let next = self.pending.pop_front();
        if let Some(p) = &next {
            let mut shared = self.shared.lock().expect("look-ahead state poisoned");
            let q = shared.tracks.entry(p.track_id).or_default();
            q.packets.pop_front();
            q.last_out = Some(PacketKey::of(&p.as_packet_ref()));
        }
        Ok(next)
    }

    fn into_inner<'s>(self: Box<Self>) -> MediaSourceStream<'s>
    where
        Self: 's,
    {
        self.inner.into_inner()
    }
}

/// What a codec module supplies to the generic decoder: the CPU parse of one packet and the batched device transform.
pub trait BatchCodec {
    /// What the CPU front end extracts from one packet: dequantised spectra / residuals + the side fields of the synthesis stage.
    type Parsed;
    /// Entropy decode + dequantise one packet with the reference's CPU code (vendored: the parse modules of the
    /// reference's codec crates are private).  Errors here are the reference's `DecodeError`s.
    fn parse(&mut self, packet: &PacketRef<'_>) -> Result<Self::Parsed>;
    /// Transform `batch` (consecutive packets of this track) in one call; the state (delay lines / overlap / V FIFO)
    /// enters and leaves through the codec's `*_io` buffers.  Writes planar PCM: `pcm[ch][i]` = packet i of channel ch.
    fn transform(&mut self, batch: &[Self::Parsed]) -> Result<()>;
    /// Copy packet `i` of the last batch into the decoder-owned `AudioBuffer` (render + plane copies).
    fn publish(&mut self, i: usize);
    /// `AudioDecoder::reset`: zero the carried state.
    fn reset_state(&mut self);
    /// Clear the `AudioBuffer` (the trait demands it on error, codecs/audio.rs:278).
    fn clear(&mut self);

    // ---- the cross-stream batcher (csrc/batcher.cpp, `symaccel_batcher_*`; the C++ twin: LookaheadDecoder's second constructor).
    // A decoder cannot see its siblings (decode_ref gets one packet of one track, codecs/audio.rs:279-297), so N decoders batching
    // their own look-ahead are N small launches.  A codec that is `pooled()` hands its batches to the process-wide batcher instead:
    // `submit` returns at once, `collect` blocks until the batch is done -- and whatever the decoders of the OTHER streams have
    // submitted meanwhile went to the device in the same launch.  The defaults describe a codec without a batcher.
    /// Does this codec submit to the shared batcher?
    fn pooled(&self) -> bool {
        false
    }
    /// Hand `batch` (the packets that follow the ones already transformed) to the batcher; the state it starts from is the state
    /// the previous batch left.  Its PCM and the state after it arrive with `collect`.
    fn submit(&mut self, _batch: &[Self::Parsed]) -> Result<()> {
        unsupported_error("codec has no batcher")
    }
    /// Wait for the submitted batch: `publish` now serves ITS packets.
    fn collect(&mut self) -> Result<()> {
        unsupported_error("codec has no batcher")
    }
    /// The submitted batch will be wanted soon (a quarter of the current one is left).
    fn hint(&mut self) {}
    /// Drop the submitted batch unseen (reset, discontinuity): its results are never looked at, the carried state stays.
    fn abandon(&mut self) {}
}

/// Batching state shared by the five decoders.
pub struct Lookahead<P> {
    /// the packets of the last batch, parsed, and their pts, in order; `head` = the next one to hand out
    parsed: Vec<P>,
    ready: Vec<i64>,
    head: usize,
    max_batch: usize,
    /// the reader this decoder's packets come from, once found
    reader: Option<SharedHandle>,
    /// a pooled codec's NEXT batch, parsed (and, if `next_live`, submitted) while the current one is still being handed out
    next_parsed: Vec<P>,
    next_ready: Vec<i64>,
    next_live: bool,
    hinted: bool,
}

impl<P> Lookahead<P> {
    pub fn new(max_batch: usize) -> Self {
        Lookahead {
            parsed: Vec::new(),
            ready: Vec::new(),
            head: 0,
            max_batch: max_batch.max(1),
            reader: None,
            next_parsed: Vec::new(),
            next_ready: Vec::new(),
            next_live: false,
            hinted: false,
        }
    }

    /// `AudioDecoder::reset` (audio.rs:252-257): nothing pre-computed survives.
    pub fn reset(&mut self) {
        self.parsed.clear();
        self.ready.clear();
        self.head = 0;
        self.next_parsed.clear();
        self.next_ready.clear();
        // (a batch that is still with the batcher is dropped by `drop_next`, which needs the codec: decode / reset_with do that first)
        self.next_live = false;
        self.hinted = false;
    }

    /// `AudioDecoder::reset` for a pooled codec: the batch submitted ahead is given up before the state is zeroed.
    pub fn reset_with<C: BatchCodec<Parsed = P>>(&mut self, codec: &mut C) {
        self.drop_next(codec);
        self.reset();
    }

    fn drop_next<C: BatchCodec<Parsed = P>>(&mut self, codec: &mut C) {
        if self.next_live {
            codec.abandon();
        }
        self.next_live = false;
        self.next_parsed.clear();
        self.next_ready.clear();
    }

    /// Batches transformed so far would be a counter in the C++ twin; here: packets waiting in the current batch.
    pub fn precomputed(&self) -> usize {
        self.ready.len() - self.head
    }

    /// The body of `decode_ref`: returns after `codec.publish` has filled the decoder's buffer with `packet`'s audio.
    /// On error the codec's buffer is cleared (audio.rs:278) and nothing pre-computed is kept.
    pub fn decode<C: BatchCodec<Parsed = P>>(&mut self, codec: &mut C, packet: &PacketRef<'_>) -> Result<()> {
        if self.head < self.ready.len() && self.ready[self.head] != packet.pts.get() {
            // Not the packet the look-ahead was computed for: the caller skipped packets without reset().  A frame-by-frame
            // decoder would continue from the state the LAST RETURNED packet left, but the carried state is already that of
            // the end of the batch.  Every codec on this path has a one-packet memory (the delay line / overlap / V FIFO after
            // a packet depend on that packet's input alone), so replaying the last returned packet rebuilds exactly that
            // state (include/symaccel.hpp, LookaheadDecoder::decode, does the same).
            self.drop_next(codec);
            let replay = if self.head >= 1 { Some(self.head - 1) } else { None };
            if let Some(i) = replay {
                if let Err(e) = codec.transform(&self.parsed[i..i + 1]) {
                    codec.clear();
                    self.reset_with(codec);
                    return Err(e);
                }
            }
            self.reset();
        }
        if self.head >= self.ready.len() {
            let taken = if !self.next_ready.is_empty() && self.next_ready[0] == packet.pts.get() {
                // the batch prepared ahead starts with this packet: collect it (or, if it could not be submitted, transform it now)
                let r = if self.next_live { codec.collect() } else { codec.transform(&self.next_parsed) };
                self.next_live = false;
                match r {
                    Ok(()) => {
                        std::mem::swap(&mut self.parsed, &mut self.next_parsed);
                        std::mem::swap(&mut self.ready, &mut self.next_ready);
                        self.next_parsed.clear();
                        self.next_ready.clear();
                        self.head = 0;
                        true
                    }
                    Err(e) => {
                        codec.clear();
                        self.reset_with(codec);
                        return Err(e);
                    }
                }
            }
            else {
                // (prepared for packets the caller then skipped: dropped unseen -- the carried state is still the one the last
                // returned packet left, because the current batch was handed out to its end)
                self.drop_next(codec);
                false
            };
            if !taken {
                if let Err(e) = self.fill(codec, packet) {
                    codec.clear();
                    self.reset_with(codec);
                    return Err(e);
                }
            }
        }
        codec.publish(self.head);
        self.head += 1;
        if codec.pooled() {
            // half of the batch handed out: the next one is parsed and submitted (the decoders of the other streams do the same
            // around now); a quarter left: whatever is pending goes to the device while the rest is consumed
            let left = self.ready.len() - self.head;
            if self.next_ready.is_empty() && 2 * left <= self.ready.len() {
                self.submit_ahead(codec, packet, left);
            }
            if self.next_live && !self.hinted && 4 * left <= self.ready.len() {
                self.hinted = true;
                codec.hint();
            }
        }
        Ok(())
    }

    /// Parse the packets that FOLLOW the current batch (the reader's queue holds what follows `packet`: first the `left` packets
    /// of the current batch that have not been handed out yet, then the new ones) and submit them.  Nothing here fails the call:
    /// a corrupt packet ends the batch (it fails at its own decode_ref), a failed submit leaves the parsed batch to be transformed
    /// by the call that needs it.
    fn submit_ahead<C: BatchCodec<Parsed = P>>(&mut self, codec: &mut C, packet: &PacketRef<'_>, left: usize) {
        let Some(reader) = self.reader.clone() else { return };
        let mut parsed = Vec::with_capacity(self.max_batch);
        let mut ids = Vec::with_capacity(self.max_batch);
        {
            let shared = reader.lock().expect("look-ahead state poisoned");
            let Some(q) = shared.tracks.get(&packet.track_id) else { return };
            if q.last_out != Some(PacketKey::of(packet)) || q.packets.len() <= left {
                return;
            }
            // the queue must continue the current batch exactly (the parser is stateful: no gaps, no repeats)
            let mut i = 0;
            for p in q.packets.iter().take(left) {
                if p.pts.get() != self.ready[self.head + i] {
                    return;
                }
                i += 1;
            }
            for p in q.packets.iter().skip(left).take(self.max_batch) {
                match codec.parse(&p.as_packet_ref()) {
                    Ok(x) => {
                        parsed.push(x);
                        ids.push(p.pts.get());
                    }
                    Err(_) => break,
                }
            }
        }
        if parsed.is_empty() {
            return;
        }
        self.next_live = codec.submit(&parsed).is_ok();
        self.next_parsed = parsed;
        self.next_ready = ids;
        self.hinted = false;
    }

    fn fill<C: BatchCodec<Parsed = P>>(&mut self, codec: &mut C, packet: &PacketRef<'_>) -> Result<()> {
        let mut parsed = Vec::with_capacity(self.max_batch);
        let mut ids = Vec::with_capacity(self.max_batch);
        parsed.push(codec.parse(packet)?);
        ids.push(packet.pts.get());
        if self.reader.is_none() {
            self.reader = find_reader(&PacketKey::of(packet));
        }
        if let Some(reader) = &self.reader {
            let shared = reader.lock().expect("look-ahead state poisoned");
            // only if `packet` is the one this reader handed out last for the track do the queued packets follow it
            if let Some(q) = shared.tracks.get(&packet.track_id) {
                if q.last_out == Some(PacketKey::of(packet)) {
                    for p in q.packets.iter().take(self.max_batch - 1) {
                        match codec.parse(&p.as_packet_ref()) {
                            Ok(x) => {
                                parsed.push(x);
                                ids.push(p.pts.get());
                            }
                            // a corrupt packet further ahead ends the batch: it fails when its own decode_ref comes
                            Err(_) => break,
                        }
                    }
                }
            }
        }
        codec.transform(&parsed)?;
        self.parsed = parsed;
        self.ready = ids;
        self.head = 0;
        Ok(())
    }
}
