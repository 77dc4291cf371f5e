//! Look-ahead batching behind the `AudioDecoder` method set (the Rust twin of `codecs::LookaheadDecoder` in
//! include/symaccel.hpp, which is the version that is compiled and tested in the repository).
//!
//! `AudioDecoder::decode_ref(&mut self, &PacketRef)` (codecs/audio.rs:281-285) sees one packet; a GPU wants thousands of
//! frames per call.  The bridge is a reader-side look-ahead: `LookaheadReader` wraps the application's `FormatReader`,
//! reads K packets ahead and publishes the packets of each track in a queue the track's decoder can see.  When
//! `decode_ref(p)` finds nothing pre-computed for `p`, the decoder parses `p` and the queued packets that follow it (CPU,
//! the reference's own parser), transforms all of them in ONE batch call, and serves the following `decode_ref` calls
//! from the result.  Without a `LookaheadReader` the queue is empty and every call is a batch of one: correct, slow.
use std::collections::{HashMap, VecDeque};
use std::sync::{Arc, Mutex, OnceLock};

use symphonia_core::errors::Result;
use symphonia_core::formats::{FormatReader, SeekMode, SeekTo, SeekedTo, Track};
use symphonia_core::packet::Packet;

/// Packets of one track the demuxer has already read but the application has not yet handed to the decoder.
#[derive(Default)]
pub struct TrackQueue {
    pub packets: VecDeque<Packet>,
}

static QUEUES: OnceLock<Mutex<HashMap<u32, Arc<Mutex<TrackQueue>>>>> = OnceLock::new();

/// The queue of a track (created on first use).  Keyed by track id: one look-ahead reader per process and container is
/// the expected shape; applications with several open containers give each its own id space through `LookaheadReader::with_id_base`.
pub fn track_queue(track_id: u32) -> Arc<Mutex<TrackQueue>> {
    let map = QUEUES.get_or_init(|| Mutex::new(HashMap::new()));
    map.lock().expect("queue map poisoned").entry(track_id).or_default().clone()
}

/// A `FormatReader` that stays `depth` packets ahead of what it returns.
pub struct LookaheadReader {
    inner: Box<dyn FormatReader>,
    depth: usize,
    pending: VecDeque<Packet>,
    eof: bool,
}

impl LookaheadReader {
    pub fn new(inner: Box<dyn FormatReader>, depth: usize) -> Self {
        LookaheadReader { inner, depth: depth.max(1), pending: VecDeque::new(), eof: false }
    }

    fn refill(&mut self) -> Result<()> {
        while !self.eof && self.pending.len() < self.depth {
            match self.inner.next_packet()? {
                Some(p) => {
                    track_queue(p.track_id).lock().expect("track queue poisoned").packets.push_back(p.clone());
                    self.pending.push_back(p);
                }
                None => self.eof = true,
            }
        }
        Ok(())
    }

    fn drop_lookahead(&mut self) {
        for p in self.pending.drain(..) {
            track_queue(p.track_id).lock().expect("track queue poisoned").packets.clear();
        }
        self.eof = false;
    }

    pub fn tracks(&self) -> &[Track] {
        self.inner.tracks()
    }

    /// `FormatReader::next_packet` (formats/mod.rs:646): the oldest pre-read packet.
    pub fn next_packet(&mut self) -> Result<Option<Packet>> {
        self.refill()?;
        let next = self.pending.pop_front();
        if let Some(p) = &next {
            // the application now owns this packet: it leaves the decoder-visible queue when the decoder consumes it
            let q = track_queue(p.track_id);
            let mut q = q.lock().expect("track queue poisoned");
            if q.packets.front().map(|f| f.pts == p.pts) == Some(true) {
                q.packets.pop_front();
            }
        }
        Ok(next)
    }

    /// `FormatReader::seek` (formats/mod.rs:591): everything read ahead is stale.  The application resets the decoder
    /// afterwards, as the reference requires (codecs/audio.rs:252-257).
    pub fn seek(&mut self, mode: SeekMode, to: SeekTo) -> Result<SeekedTo> {
        self.drop_lookahead();
        self.inner.seek(mode, to)
    }
}

/// What a codec module supplies to the generic decoder: the CPU parse of one packet and the batched device transform.
pub trait BatchCodec {
    /// What the CPU front end extracts from one packet: dequantised spectra / residuals + the side fields of the synthesis stage.
    type Parsed;
    /// Entropy decode + dequantise one packet with the reference's CPU code (vendored: the parse modules of the
    /// reference's codec crates are private).  Errors here are the reference's `DecodeError`s.
    fn parse(&mut self, packet: &Packet) -> Result<Self::Parsed>;
    /// Transform `batch` (consecutive packets of this track) in one call; the state (delay lines / overlap / V FIFO)
    /// enters and leaves through the codec's `*_io` buffers.  Writes planar PCM: `pcm[ch][i]` = packet i of channel ch.
    fn transform(&mut self, batch: &[Self::Parsed]) -> Result<()>;
    /// Copy packet `i` of the last batch into the decoder-owned `AudioBuffer` (render + plane copies).
    fn publish(&mut self, i: usize);
    /// `AudioDecoder::reset`: zero the carried state.
    fn reset_state(&mut self);
    /// Clear the `AudioBuffer` (the trait demands it on error, codecs/audio.rs:278).
    fn clear(&mut self);
}

/// Batching state shared by the four decoders.
pub struct Lookahead {
    /// pts of the packets of the last batch, in order; `head` = the next one to hand out.
    ready: Vec<u64>,
    head: usize,
    max_batch: usize,
}

impl Lookahead {
    pub fn new(max_batch: usize) -> Self {
        Lookahead { ready: Vec::new(), head: 0, max_batch: max_batch.max(1) }
    }

    pub fn reset(&mut self) {
        self.ready.clear();
        self.head = 0;
    }

    /// The body of `decode_ref`: returns after `codec.publish` has filled the decoder's buffer with `packet`'s audio.
    pub fn decode<C: BatchCodec>(&mut self, codec: &mut C, packet: &Packet) -> Result<()> {
        if self.head < self.ready.len() && self.ready[self.head] != packet.pts.get() {
            // discontinuity without reset(): drop the pre-computed frames (the C++ twin also replays the last returned
            // packet to rebuild the exact state; the same one-packet-memory argument applies here)
            self.reset();
        }
        if self.head >= self.ready.len() {
            let mut parsed = Vec::with_capacity(self.max_batch);
            let mut ids = Vec::with_capacity(self.max_batch);
            parsed.push(codec.parse(packet)?);
            ids.push(packet.pts.get());
            {
                let q = track_queue(packet.track_id);
                let q = q.lock().expect("track queue poisoned");
                for p in q.packets.iter().take(self.max_batch - 1) {
                    match codec.parse(p) {
                        Ok(x) => {
                            parsed.push(x);
                            ids.push(p.pts.get());
                        }
                        // a corrupt packet further ahead ends the batch: it fails when its own decode_ref comes
                        Err(_) => break,
                    }
                }
            }
            if let Err(e) = codec.transform(&parsed) {
                codec.clear();
                self.reset();
                return Err(e);
            }
            self.ready = ids;
            self.head = 0;
        }
        codec.publish(self.head);
        self.head += 1;
        Ok(())
    }
}
