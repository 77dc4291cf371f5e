//! TEST ONLY.  Stand-ins for the few symphonia-core items bindings/rust/symphonia-accel-hip/src/{lookahead,fallback}.rs
//! use, so that those two files can be EXECUTED under tools/rsinterp (tests/test_rust_shim.py).  Same names, fields and
//! method signatures as the reference (symphonia-core/src/packet.rs:48-119, 144-201; units.rs:30-52; errors.rs:38-54;
//! codecs/registry.rs:99-165, 195-269); test_rust_shim.py checks that against the reference's own text where
//! /root/reference is present.  Bodies are the obvious ones.

pub struct Timestamp(i64);

impl Timestamp {
    pub const fn new(ts: i64) -> Self {
        Timestamp(ts)
    }
    pub const fn get(self) -> i64 {
        self.0
    }
}

pub struct Packet {
    pub track_id: u32,
    pub pts: Timestamp,
    pub data: Vec<u8>,
}

impl Packet {
    pub fn new(track_id: u32, pts: Timestamp, data: Vec<u8>) -> Self {
        Packet { track_id, pts, data }
    }
    pub fn as_packet_ref(&self) -> PacketRef<'_> {
        PacketRef { track_id: self.track_id, pts: self.pts, data: &self.data }
    }
}

pub struct PacketRef<'a> {
    pub track_id: u32,
    pub pts: Timestamp,
    pub data: &'a [u8],
}

pub enum Error {
    IoError(&'static str),
    DecodeError(&'static str),
    SeekError(&'static str),
    Unsupported(&'static str),
    ResetRequired,
}

pub fn decode_error<T>(desc: &'static str) -> Result<T> {
    Err(Error::DecodeError(desc))
}

pub fn unsupported_error<T>(feature: &'static str) -> Result<T> {
    Err(Error::Unsupported(feature))
}

pub enum Tier {
    Preferred,
    Standard,
    Fallback,
}

pub struct AudioCodecId(u32);

pub struct AudioCodecParameters {
    pub codec: AudioCodecId,
}

pub struct AudioDecoderOptions {
    pub gapless: bool,
}

pub struct RegisteredAudioDecoder {
    pub codec: AudioCodecId,
    pub factory: fn(&AudioCodecParameters, &AudioDecoderOptions) -> Result<i64>,
}

/// codecs/registry.rs:131-165: three maps, looked up preferred -> standard -> fallback, NO fall-through on a factory error
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

    pub fn get_audio_decoder_at_tier(&self, tier: Tier, id: AudioCodecId) -> Option<&RegisteredAudioDecoder> {
        match tier {
            Tier::Preferred => self.preferred.get(&id),
            Tier::Standard => self.standard.get(&id),
            Tier::Fallback => self.fallback.get(&id),
        }
    }

    pub fn register_at_tier(&mut self, tier: Tier, id: AudioCodecId, factory: fn(&AudioCodecParameters, &AudioDecoderOptions) -> Result<i64>) {
        let reg = RegisteredAudioDecoder { codec: id, factory };
        match tier {
            Tier::Preferred => self.preferred.insert(id, reg),
            Tier::Standard => self.standard.insert(id, reg),
            Tier::Fallback => self.fallback.insert(id, reg),
        };
    }

    /// codecs/registry.rs:330-341
    pub fn make_audio_decoder(&self, params: &AudioCodecParameters, opts: &AudioDecoderOptions) -> Result<i64> {
        if let Some(codec) = self.get_audio_decoder(params.codec) {
            Ok((codec.factory)(params, opts)?)
        }
        else {
            unsupported_error("core (codec): unsupported audio codec")
        }
    }
}
