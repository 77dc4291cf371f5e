//! TEST ONLY.  The registry items bindings/rust/symphonia-accel-hip/src/fallback.rs uses, for the harness configurations that load
//! the reference's own errors.rs / packet.rs instead of tests/rust/core_stubs.rs (which carries the same stand-ins; see there for
//! the citations: codecs/registry.rs:93-165, 195-269, 330-341).

pub enum Tier {
    Preferred,
    Standard,
    Fallback,
}

pub struct RegisteredAudioDecoder {
    pub codec: AudioCodecId,
    pub factory: fn(&AudioCodecParameters, &AudioDecoderOptions) -> Result<i64>,
}

pub struct CodecRegistry {
    preferred: HashMap<AudioCodecId, RegisteredAudioDecoder>,
    standard: HashMap<AudioCodecId, RegisteredAudioDecoder>,
    fallback: HashMap<AudioCodecId, RegisteredAudioDecoder>,
}

impl CodecRegistry {
    pub fn new() -> Self {
        CodecRegistry { preferred: HashMap::new(), standard: HashMap::new(), fallback: HashMap::new() }
    }

    pub fn get_audio_decoder(&self, id: AudioCodecId) -> Option<&RegisteredAudioDecoder> {
        self.preferred.get(&id).or_else(|| self.standard.get(&id)).or_else(|| self.fallback.get(&id))
    }
}
