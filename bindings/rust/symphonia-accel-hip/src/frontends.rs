//! CPU front ends: bitstream parse, Huffman / VQ / Rice decode, dequantisation, joint stereo -- the reference's own code, up to
//! the point where the synthesis stage starts.
//!
//! The reference keeps those stages in private modules (symphonia-codec-aac/src/aac/mod.rs:29-34,
//! symphonia-codec-vorbis/src/lib.rs:37-42, symphonia-bundle-mp3/src/lib.rs:18-40), so this crate does not carry a copy of
//! them: it builds against codec crates with the SEAM PATCHES of `bindings/rust/patches/` applied
//! (`symphonia-bundle-flac.diff`, `symphonia-codec-aac.diff`, `symphonia-bundle-mp3.diff`, `symphonia-codec-vorbis.diff`,
//! `symphonia-codec-alac.diff`).
//! Each patch adds a `pub trait SynthBackend` at the place where the decoder calls its DSP
//! (flac decoder.rs:199-242 + 446-511, aac ics/mod.rs:449-468, mp3 layer3/mod.rs:421-477, vorbis lib.rs:316-331, alac lib.rs:541-598)
//! with the
//! crate's own CPU code as the default, and a `try_new_with_backend` constructor.  The front ends below are the reference's
//! decoders with a RECORDING backend installed: `parse(packet)` runs the reference's `decode_ref` -- which now stops short
//! of the DSP -- and returns what the backend was handed.  The batched device call then does the DSP for many packets.
//!
//! tests/test_seam_patches.py applies the patches to a copy of the reference tree and checks that every patched file still
//! parses and that the `SynthBackend` impls in this crate match the patched traits; tests/test_flac_packets.py EXECUTES the
//! patched FLAC decoder + this crate's FLAC path on packet bytes (under the repository's Rust interpreter) against the
//! unpatched reference decoder.
use symphonia_core::codecs::audio::{AudioCodecParameters, AudioDecoderOptions};
use symphonia_core::errors::Result;

use crate::aac::AacFrontEnd;
use crate::alac::AlacFrontEnd;
use crate::flac::FlacFrontEnd;
use crate::mpa::MpaFrontEnd;
use crate::vorbis::VorbisFrontEnd;

/// Which front ends this build contains: all five, through the seam patches (kept as a table so that a build against an
/// unpatched codec crate can switch a codec off instead of failing to link; see `register`).
pub struct Available {
    pub aac: bool,
    pub mpa: bool,
    pub vorbis: bool,
    pub flac: bool,
    pub alac: bool,
}

pub const AVAILABLE: Available = Available { aac: true, mpa: true, vorbis: true, flac: true, alac: true };

pub fn aac_front_end(params: &AudioCodecParameters, opts: &AudioDecoderOptions) -> Result<Box<dyn AacFrontEnd>> {
    Ok(Box::new(crate::aac::SeamFrontEnd::try_new(params, opts)?))
}

pub fn mpa_front_end(params: &AudioCodecParameters, opts: &AudioDecoderOptions) -> Result<Box<dyn MpaFrontEnd>> {
    Ok(Box::new(crate::mpa::SeamFrontEnd::try_new(params, opts)?))
}

pub fn vorbis_front_end(params: &AudioCodecParameters, opts: &AudioDecoderOptions) -> Result<Box<dyn VorbisFrontEnd>> {
    Ok(Box::new(crate::vorbis::SeamFrontEnd::try_new(params, opts)?))
}

pub fn flac_front_end(params: &AudioCodecParameters, _opts: &AudioDecoderOptions) -> Result<Box<dyn FlacFrontEnd>> {
    Ok(Box::new(crate::flac::SeamFrontEnd::try_new(params)?))
}

pub fn alac_front_end(params: &AudioCodecParameters, opts: &AudioDecoderOptions) -> Result<Box<dyn AlacFrontEnd>> {
    Ok(Box::new(crate::alac::SeamFrontEnd::try_new(params, opts)?))
}
