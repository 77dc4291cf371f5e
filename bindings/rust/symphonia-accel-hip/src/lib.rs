//! symphonia-accel-hip -- drop-in `AudioDecoder`s whose synthesis stage runs on an AMD MI355X through libsymaccel.
//!
//! ```ignore
//! let mut registry = symphonia::default::get_codecs().clone();   // or CodecRegistry::new() + your own set
//! symphonia_accel_hip::register(&mut registry);                  // Tier::Preferred: wins over the CPU decoders
//! let reader = symphonia_accel_hip::LookaheadReader::new(format_reader, 256);
//! ```
//! Everything above the decoders -- demuxers, probe, symphonia-check, symphonia-play -- is untouched: they only see
//! `Box<dyn AudioDecoder>` (symphonia-check/src/main.rs:144-147).
//!
//! Status: this crate is NOT compiled in the repository's build image (no Rust toolchain).  Its compiled and tested twin
//! is `codecs::LookaheadDecoder` in include/symaccel.hpp (tests/cpp/lookahead_test.cpp).  The parse stages
//! (`frontends`) have to be vendored from the reference's codec crates because their `mod`s are private
//! (symphonia-codec-aac/src/aac/mod.rs:29-34, symphonia-codec-vorbis/src/lib.rs:37-42, symphonia-bundle-mp3/src/lib.rs:18-40).
#![allow(clippy::needless_range_loop)]

mod aac;
mod ctx;
pub mod decoder;
mod ffi;
mod flac;
pub mod frontends;
mod lookahead;
mod mpa;
mod vorbis;

pub use aac::{AacFrontEnd, HipAacDecoder, ParsedAac};
pub use flac::{FlacFrontEnd, HipFlacDecoder, ParsedFlac};
pub use mpa::{HipMpaDecoder, MpaFrontEnd, ParsedMpa};
pub use vorbis::{HipVorbisDecoder, ParsedVorbis, VorbisFrontEnd};
pub use ctx::{Context, Pinned};
pub use lookahead::{track_queue, BatchCodec, Lookahead, LookaheadReader, TrackQueue};

use symphonia_core::codecs::registry::{CodecRegistry, Tier};

/// Packets per batch when a `LookaheadReader` feeds the decoder.
pub const DEFAULT_LOOKAHEAD: usize = 256;

/// Register the accelerated decoders at `Tier::Preferred` (symphonia-core/src/codecs/registry.rs:252-269): the registry
/// looks preferred -> standard -> fallback (`:152-154`), so the CPU decoders stay available underneath.
pub fn register(registry: &mut CodecRegistry) {
    registry.register_audio_decoder_at_tier::<HipAacDecoder>(Tier::Preferred);
    registry.register_audio_decoder_at_tier::<HipMpaDecoder>(Tier::Preferred);
    registry.register_audio_decoder_at_tier::<HipVorbisDecoder>(Tier::Preferred);
    registry.register_audio_decoder_at_tier::<HipFlacDecoder>(Tier::Preferred);
}
