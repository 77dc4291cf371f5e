//! TEST ONLY.  The two registry methods a `register()` -> `make_audio_decoder()` round trip needs, for the stand-in `CodecRegistry` of
//! tests/rust/core_stubs.rs / registry_stubs.rs, restated from the reference (symphonia-core/src/codecs/registry.rs:252-269: every codec a
//! decoder type supports is entered at the tier with a factory that calls the type's `try_registry_new`; :330-341: the most preferred
//! entry's factory builds the decoder, and its error is the caller's -- no fall-through).

impl CodecRegistry {
    pub fn register_audio_decoder_at_tier<C: RegisterableAudioDecoder>(&mut self, tier: Tier) {
        for codec in C::supported_codecs() {
            let reg = RegisteredAudioDecoder { codec: codec.id, factory: |params, opts| C::try_registry_new(params, opts) };
            match tier {
                Tier::Preferred => self.preferred.insert(codec.id, reg),
                Tier::Standard => self.standard.insert(codec.id, reg),
                Tier::Fallback => self.fallback.insert(codec.id, reg),
            };
        }
    }

    pub fn make_registered_audio_decoder(&self, params: &AudioCodecParameters, opts: &AudioDecoderOptions) -> Result<Box<dyn AudioDecoder>> {
        if let Some(codec) = self.get_audio_decoder(params.codec) {
            Ok((codec.factory)(params, opts)?)
        }
        else {
            unsupported_error("core (codec): unsupported audio codec")
        }
    }
}
