//! symphonia-accel-hip -- drop-in `AudioDecoder`s whose synthesis stage runs on an AMD MI355X through libsymaccel.
//!
//! ```ignore
//! let mut registry = symphonia::default::get_codecs().clone();   // or CodecRegistry::new() + your own set
//! symphonia_accel_hip::register(&mut registry);                  // Tier::Preferred: wins over the CPU decoders
//! let reader: Box<dyn FormatReader> = Box::new(symphonia_accel_hip::LookaheadReader::new(format_reader, 256));
//! ```
//! Everything above the decoders -- demuxers, probe, symphonia-check, symphonia-play -- is untouched: they only see
//! `Box<dyn AudioDecoder>` and `Box<dyn FormatReader>` (symphonia-check/src/main.rs:144-147).
//!
//! The parse stages are the reference's own decoders: this crate builds against codec crates with the seam patches of
//! `bindings/rust/patches/` applied (a `pub trait SynthBackend` where each decoder calls its DSP; `frontends.rs`).
//!
//! Status: this crate is NOT compiled in the repository's build image (no Rust toolchain).  What IS checked there: every
//! file parses, every `use symphonia_core::...` names an item the reference tree defines, every `impl AudioDecoder /
//! RegisterableAudioDecoder / FormatReader` matches the trait text of the reference (tests/test_rust_shim.py); the patches
//! apply to the reference tree, the patched files parse and this crate's `SynthBackend` impls match the patched traits
//! (tests/test_seam_patches.py); and the crate is EXECUTED under the repository's Rust interpreter (tools/rsinterp) with its
//! `extern "C"` block bound to libsymaccel through ctypes: `lookahead.rs` + `fallback.rs` with mocks (test_rust_shim.py), the
//! five codec adapters + `ctx.rs` with front ends that replay the reference-text fixtures (tests/test_rust_adapters.py), and
//! all five paths whole -- the patched reference decoder as front end, `aac.rs` / `mpa.rs` / `vorbis.rs` / `flac.rs` /
//! `alac.rs`, `decoder.rs`, `lookahead.rs`, `ctx.rs` -- on packet bytes against the unpatched reference decoder
//! (tests/test_{aac,mp3,vorbis,flac,alac}_packets.py).
#![allow(clippy::needless_range_loop)]

mod aac;
mod ctx;
pub mod decoder;
pub mod fallback;
mod ffi;
mod alac;
mod flac;
pub mod frontends;
mod lookahead;
mod mpa;
mod vorbis;

pub use aac::{AacFrontEnd, HipAacDecoder, ParsedAac};
pub use alac::{AlacFrontEnd, HipAlacDecoder, ParsedAlac};
pub use decoder::DecoderBatch;
pub use ctx::{Context, Pinned, Pool};
pub use flac::{FlacFrontEnd, HipFlacDecoder, ParsedFlac};
pub use lookahead::{find_reader, BatchCodec, Lookahead, LookaheadReader, PacketKey, Shared, SharedHandle, TrackQueue};
pub use mpa::{HipMpaDecoder, MpaFrontEnd, ParsedMpa};
pub use vorbis::{HipVorbisDecoder, ParsedVorbis, VorbisFrontEnd};

use symphonia_core::codecs::audio::AudioCodecId;
use symphonia_core::codecs::registry::{CodecRegistry, RegisterableAudioDecoder};
use symphonia_core::common::Tier;

/// Packets per batch when a `LookaheadReader` feeds the decoder.
pub const DEFAULT_LOOKAHEAD: usize = 256;

/// Register the accelerated decoders at `Tier::Preferred` (symphonia-core/src/codecs/registry.rs:252-269).
///
/// The registry does NOT fall through to the next tier when a preferred factory fails (`:152-154`, `:330-341`), so
///  * a codec whose parse stage is not in this build (`frontends::AVAILABLE`) is not registered at all -- the CPU decoder
///    keeps the codec --, and
///  * for the codecs that are registered, the factory that was in force before is remembered (`fallback::remember`) and
///    `try_registry_new` delegates to it when the device decoder cannot be built (no GPU, out of memory).
pub fn register(registry: &mut CodecRegistry) {
    register_one::<HipAacDecoder>(registry, frontends::AVAILABLE.aac);
    register_one::<HipMpaDecoder>(registry, frontends::AVAILABLE.mpa);
    register_one::<HipVorbisDecoder>(registry, frontends::AVAILABLE.vorbis);
    register_one::<HipFlacDecoder>(registry, frontends::AVAILABLE.flac);
    register_one::<HipAlacDecoder>(registry, frontends::AVAILABLE.alac);
}

/// Register one decoder type above whatever the registry holds for its codecs, if its front end is present.
pub fn register_one<D: RegisterableAudioDecoder>(registry: &mut CodecRegistry, front_end_present: bool) {
    if !front_end_present {
        return;
    }
    let ids: Vec<AudioCodecId> = D::supported_codecs().iter().map(|c| c.id).collect();
    fallback::remember(registry, &ids);
    registry.register_audio_decoder_at_tier::<D>(Tier::Preferred);
}
