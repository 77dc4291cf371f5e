//! CPU front ends: bitstream parse, Huffman / VQ decode, dequantisation -- the reference's own code, up to the point
//! where the synthesis stage starts.  The reference keeps these in private modules, so a shim has to carry a copy of
//! the parse stage (or the reference has to grow a `pub trait SynthBackend`).  The copy is mechanical:
//! take `AacDecoder::decode_inner` (symphonia-codec-aac/src/aac/mod.rs:170-225), stop before `synth_audio`, and return
//! `ics.coeffs` + `info.window_sequence / window_shape / prev_window_shape` per channel.
use symphonia_core::codecs::audio::AudioCodecParameters;
use symphonia_core::errors::{unsupported_error, Result};

use crate::aac::AacFrontEnd;

/// Build the AAC-LC front end for a track.  (Vendored parser goes here; see the module comment.)
pub fn aac_front_end(_params: &AudioCodecParameters) -> Result<Box<dyn AacFrontEnd>> {
    unsupported_error("symphonia-accel-hip: the vendored AAC parse stage is not part of this source drop")
}
