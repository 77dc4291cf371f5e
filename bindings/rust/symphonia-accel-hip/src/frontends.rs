//! CPU front ends: bitstream parse, Huffman / VQ / Rice decode, dequantisation -- the reference's own code, up to the
//! point where the synthesis stage starts.  The reference keeps these in private modules, so a shim has to carry a copy
//! of the parse stage (or the reference has to grow a `pub trait SynthBackend`).  The copy is mechanical:
//!  * AAC: `AacDecoder::decode_inner` (symphonia-codec-aac/src/aac/mod.rs:170-225), stopped before `synth_audio`,
//!    returning `ics.coeffs` + `info.window_sequence / window_shape / prev_window_shape` per channel;
//!  * MP3: `Layer3::decode` (symphonia-bundle-mp3/src/layer3/mod.rs:300-440) up to the per-channel tail, returning the
//!    requantized + stereo-processed samples and `block_type / is_mixed / rzero` per granule-channel;
//!  * Vorbis: `VorbisDecoder::decode_inner` (symphonia-codec-vorbis/src/lib.rs:186-292) up to `dsp.channels[..].synth`,
//!    returning floor x residue per channel and the mode's block flag;
//!  * FLAC: `FlacDecoder::decode_inner` (symphonia-bundle-flac/src/decoder.rs:200-300) with `read_subframe` stopped
//!    before `fixed_predict` / `lpc_predict` (:456-520), returning warm-up + residual words, the subframe descriptor
//!    and the quantised coefficients.
//!
//! None of the four copies is part of this source drop, and the crate is honest about it: `AVAILABLE` says so per codec,
//! `register` (lib.rs) does not put a decoder above the CPU one for a codec whose front end is absent, and a decoder
//! built directly through `try_registry_new` hands the track to the CPU decoder it was registered above (fallback.rs).
use symphonia_core::codecs::audio::AudioCodecParameters;
use symphonia_core::errors::{unsupported_error, Result};

use crate::aac::AacFrontEnd;
use crate::flac::FlacFrontEnd;
use crate::mpa::MpaFrontEnd;
use crate::vorbis::VorbisFrontEnd;

/// Which vendored parse stages this build contains.
pub struct Available {
    pub aac: bool,
    pub mpa: bool,
    pub vorbis: bool,
    pub flac: bool,
}

pub const AVAILABLE: Available = Available { aac: false, mpa: false, vorbis: false, flac: false };

const NOT_IN_DROP: &str = "symphonia-accel-hip: the vendored parse stage of this codec is not part of this source drop";

/// Build the AAC-LC front end for a track.  (Vendored parser goes here; see the module comment.)
pub fn aac_front_end(_params: &AudioCodecParameters) -> Result<Box<dyn AacFrontEnd>> {
    unsupported_error(NOT_IN_DROP)
}

pub fn mpa_front_end(_params: &AudioCodecParameters) -> Result<Box<dyn MpaFrontEnd>> {
    unsupported_error(NOT_IN_DROP)
}

pub fn vorbis_front_end(_params: &AudioCodecParameters) -> Result<Box<dyn VorbisFrontEnd>> {
    unsupported_error(NOT_IN_DROP)
}

pub fn flac_front_end(_params: &AudioCodecParameters) -> Result<Box<dyn FlacFrontEnd>> {
    unsupported_error(NOT_IN_DROP)
}
