//! The `AudioDecoder` / `RegisterableAudioDecoder` boilerplate shared by the MPEG audio, Vorbis and FLAC decoders
//! (`aac.rs` spells the same thing out by hand and is the one to read first).
//!
//! A codec module supplies a `BatchCodec` (parse one packet on the CPU, transform a batch on the GPU, publish one packet
//! of the batch into the decoder-owned `AudioBuffer`) that also exposes that buffer; the macro wraps it in a struct
//! with the trait's observable behaviour: `reset()` zeroes the carried state and drops pre-computed frames
//! (codecs/audio.rs:252-257), a failed `decode_ref` leaves the buffer cleared (:273-278), `last_decoded()` returns the
//! buffer of the last successful call (:291-297), `finalize()` has nothing to verify (no decoder in this crate
//! carries a checksum: FLAC's MD5 is computed by the caller over what we return, as with the reference's decoder when
//! `verify` is off).
use symphonia_core::audio::GenericAudioBufferRef;

use crate::lookahead::BatchCodec;

/// A `BatchCodec` that owns the `AudioBuffer` its `publish` fills.
pub trait DecoderBatch: BatchCodec {
    fn buffer(&self) -> GenericAudioBufferRef<'_>;
}

#[macro_export]
macro_rules! hip_decoder {
    ($name:ident, $batch:ty, $parsed:ty, $front_end:path, $codecs:expr, $doc:literal) => {
        #[doc = $doc]
        pub struct $name {
            params: symphonia_core::codecs::audio::AudioCodecParameters,
            batch: $batch,
            la: $crate::lookahead::Lookahead<$parsed>,
        }

        impl symphonia_core::codecs::audio::AudioDecoder for $name {
            fn reset(&mut self) {
                self.la.reset_with(&mut self.batch);  // (a batch still with the cross-stream batcher is given up first)
                $crate::lookahead::BatchCodec::reset_state(&mut self.batch);
            }

            fn codec_info(&self) -> &symphonia_core::codecs::CodecInfo {
                use symphonia_core::codecs::registry::RegisterableAudioDecoder;
                &Self::supported_codecs()[0].info
            }

            fn codec_params(&self) -> &symphonia_core::codecs::audio::AudioCodecParameters {
                &self.params
            }

            fn decode_ref(
                &mut self,
                packet: &symphonia_core::packet::PacketRef<'_>,
            ) -> symphonia_core::errors::Result<symphonia_core::audio::GenericAudioBufferRef<'_>> {
                // (Lookahead::decode clears the buffer on every error path: codecs/audio.rs:278)
                self.la.decode(&mut self.batch, packet)?;
                Ok($crate::decoder::DecoderBatch::buffer(&self.batch))
            }

            fn finalize(&mut self) -> symphonia_core::codecs::audio::FinalizeResult {
                Default::default()
            }

            fn last_decoded(&self) -> symphonia_core::audio::GenericAudioBufferRef<'_> {
                $crate::decoder::DecoderBatch::buffer(&self.batch)
            }
        }

        impl symphonia_core::codecs::registry::RegisterableAudioDecoder for $name {
            fn try_registry_new(
                params: &symphonia_core::codecs::audio::AudioCodecParameters,
                opts: &symphonia_core::codecs::audio::AudioDecoderOptions,
            ) -> symphonia_core::errors::Result<Box<dyn symphonia_core::codecs::audio::AudioDecoder>> {
                // The registry builds every decoder from (params, opts) alone (codecs/registry.rs:330-341): the decoders it builds
                // find each other in the process-wide `Pool` -- their look-ahead batches go to the device in common launches
                // (csrc/batcher.cpp).  Only when the pool cannot be created does a decoder batch on its own.
                // No front end, no device, no memory: the decoder that was registered below this one takes the track.
                let built = $front_end(params, opts).and_then(|front| match $crate::ctx::Pool::shared() {
                    Ok(pool) => Self::try_new_with_pool(params, opts, front, $crate::DEFAULT_LOOKAHEAD, Some(pool)),
                    Err(_) => Self::try_new(params, opts, front, $crate::DEFAULT_LOOKAHEAD),
                });
                match built {
                    Ok(decoder) => Ok(Box::new(decoder)),
                    Err(e) => $crate::fallback::make(params, opts, e),
                }
            }

            fn supported_codecs() -> &'static [symphonia_core::codecs::registry::SupportedAudioCodec] {
                $codecs
            }
        }
    };
}
