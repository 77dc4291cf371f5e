//! TEST ONLY.  Stand-ins for the symphonia-core CONTAINER types the codec adapters of bindings/rust/symphonia-accel-hip and the
//! reference's decoders hold their output in, so that those files can be EXECUTED under tools/rsinterp (tests/test_rust_adapters.py,
//! tests/test_flac_packets.py).  Same names, fields and method signatures as the reference (symphonia-core/src/audio/mod.rs:31-72,
//! audio/buf.rs:60-501, audio/channels.rs:276-318, audio/generic.rs, codecs/audio.rs:300-400, codecs/mod.rs:24-40,
//! codecs/registry.rs:24-32, 398-419); bodies are the obvious ones (the reference's AudioBuffer is written with SmallVec,
//! mem::take on `&mut [Vec<S>]` and iterator adapters the interpreter's subset does not cover).  Everything that PARSES or
//! COMPUTES -- bit readers, CRCs, frame and subframe decoding -- is the reference's own text, loaded from /root/reference.

pub enum Channels {
    Discrete(u16),
    None,
}

impl Channels {
    pub fn count(&self) -> usize {
        match self {
            Channels::Discrete(count) => usize::from(*count),
            Channels::None => 0,
        }
    }
}

/// audio/channels.rs: the layouts Apple Lossless names (symphonia-common/src/apple/audio/alac.rs:152-161); the stand-in keeps the count
pub mod layouts {
    pub const CHANNEL_LAYOUT_MONO: Channels = Channels::Discrete(1);
    pub const CHANNEL_LAYOUT_STEREO: Channels = Channels::Discrete(2);
    pub const CHANNEL_LAYOUT_MPEG_3P0_B: Channels = Channels::Discrete(3);
    pub const CHANNEL_LAYOUT_MPEG_4P0_B: Channels = Channels::Discrete(4);
    pub const CHANNEL_LAYOUT_MPEG_5P0_D: Channels = Channels::Discrete(5);
    pub const CHANNEL_LAYOUT_MPEG_5P1_D: Channels = Channels::Discrete(6);
    pub const CHANNEL_LAYOUT_AAC_6P1: Channels = Channels::Discrete(7);
    pub const CHANNEL_LAYOUT_MPEG_7P1_B: Channels = Channels::Discrete(8);
}

pub struct AudioSpec {
    rate: u32,
    channels: Channels,
}

impl AudioSpec {
    pub fn new(rate: u32, channels: Channels) -> Self {
        AudioSpec { rate, channels }
    }
    pub fn rate(&self) -> u32 {
        self.rate
    }
    pub fn channels(&self) -> &Channels {
        &self.channels
    }
}

/// audio/buf.rs:60-501 (the methods the decoders on the path call)
pub struct AudioBuffer<S> {
    spec: AudioSpec,
    planes: Vec<Vec<S>>,
    num_frames: usize,
    capacity: usize,
}

impl<S> AudioBuffer<S> {
    pub fn new(spec: AudioSpec, capacity: usize) -> Self {
        let num_channels = spec.channels().count();
        let mut planes = Vec::new();
        for _ in 0..num_channels {
            // `S::MID`: the harness supplies it (tests/rs_harness.py `sample=`), since `S` is inferred from the declared type of the
            // field the buffer is stored in, which the interpreter does not track
            planes.push(vec![audio_stub_sample_mid(); capacity]);
        }
        AudioBuffer { spec, planes, num_frames: 0, capacity }
    }
    pub fn is_unused(&self) -> bool {
        self.capacity == 0 || self.spec.channels.count() == 0
    }
    pub fn capacity(&self) -> usize {
        self.capacity
    }
    pub fn clear(&mut self) {
        self.num_frames = 0;
    }
    pub fn render_uninit(&mut self, num_frames: Option<usize>) -> usize {
        let num_new_frames = num_frames.unwrap_or(self.capacity - self.num_frames);
        assert!(self.num_frames + num_new_frames <= self.capacity(), "capacity will be exceeded");
        self.num_frames += num_new_frames;
        num_new_frames
    }
    /// audio/buf.rs render_silence: the new frames are the sample format's mid-point (0 for the i32 buffer of the one caller, ALAC)
    pub fn render_silence(&mut self, num_frames: Option<usize>) {
        let start = self.num_frames;
        let num_new_frames = self.render_uninit(num_frames);
        for plane in &mut self.planes {
            for i in start..start + num_new_frames {
                plane[i] = 0;
            }
        }
    }
    pub fn shift(&mut self, shift: usize) {
        if shift >= self.num_frames {
            self.clear();
        }
        else if shift > 0 {
            for plane in &mut self.planes {
                for i in shift..self.num_frames {
                    plane[i - shift] = plane[i];
                }
            }
            self.num_frames -= shift;
        }
    }
    pub fn truncate(&mut self, num_frames: usize) {
        if num_frames < self.num_frames {
            self.num_frames = num_frames;
        }
    }
    pub fn trim(&mut self, start: usize, end: usize) {
        self.truncate(self.frames().saturating_sub(end));
        self.shift(start);
    }
    pub fn spec(&self) -> &AudioSpec {
        &self.spec
    }
    pub fn num_planes(&self) -> usize {
        self.planes.len()
    }
    pub fn is_empty(&self) -> bool {
        self.num_frames == 0
    }
    pub fn frames(&self) -> usize {
        self.num_frames
    }
    pub fn plane(&self, idx: usize) -> Option<&[S]> {
        if idx < self.planes.len() { Some(&self.planes[idx][0..self.num_frames]) } else { None }
    }
    pub fn plane_mut(&mut self, idx: usize) -> Option<&mut [S]> {
        if idx < self.planes.len() { Some(&mut self.planes[idx][0..self.num_frames]) } else { None }
    }
    pub fn plane_pair_mut(&mut self, idx0: usize, idx1: usize) -> Option<(&mut [S], &mut [S])> {
        assert!(idx0 != idx1, "plane indicies cannot be the same");
        if idx0 < self.planes.len() && idx1 < self.planes.len() {
            Some((&mut self.planes[idx0][0..self.num_frames], &mut self.planes[idx1][0..self.num_frames]))
        }
        else {
            None
        }
    }
    pub fn apply<F>(&mut self, f: F) {
        for plane in &mut self.planes {
            for i in 0..self.num_frames {
                plane[i] = f(plane[i]);
            }
        }
    }
    // `as_generic_audio_buffer_ref` (audio/generic.rs:381-400) needs to know `S`: the harness supplies it as a native method
    // that wraps the buffer in GenericAudioBufferRef::S32 or ::F32 by looking at the samples (tests/rs_harness.py)
}

/// audio/generic.rs:381-400 (the two sample formats the decoders on the path produce)
pub enum GenericAudioBufferRef<'a> {
    S32(&'a AudioBuffer<i32>),
    F32(&'a AudioBuffer<f32>),
}

/// codecs/mod.rs:24-40
pub struct CodecInfo {
    pub short_name: &'static str,
    pub long_name: &'static str,
}

/// codecs/registry.rs:24-32
pub struct SupportedAudioCodec {
    pub id: AudioCodecId,
    pub info: CodecInfo,
}

/// codecs/registry.rs:398-419
macro_rules! support_audio_codec {
    ($id:expr, $short_name:expr, $long_name:expr) => {
        SupportedAudioCodec { id: $id, info: CodecInfo { short_name: $short_name, long_name: $long_name } }
    };
}

pub struct AudioCodecId(u32);

pub const CODEC_ID_VORBIS: AudioCodecId = AudioCodecId(0x1000);
pub const CODEC_ID_MP3: AudioCodecId = AudioCodecId(0x1003);
pub const CODEC_ID_AAC: AudioCodecId = AudioCodecId(0x1004);
pub const CODEC_ID_FLAC: AudioCodecId = AudioCodecId(0x2000);
pub const CODEC_ID_ALAC: AudioCodecId = AudioCodecId(0x2003);

/// codecs/audio.rs:300-400 (the fields the decoders on the path read or amend)
pub struct AudioCodecParameters {
    pub codec: AudioCodecId,
    pub sample_rate: Option<u32>,
    pub bits_per_sample: Option<u32>,
    pub channels: Option<Channels>,
    pub max_frames_per_packet: Option<u64>,
    pub extra_data: Option<Box<[u8]>>,
}

impl AudioCodecParameters {
    pub fn with_sample_rate(&mut self, sample_rate: u32) -> &mut Self {
        self.sample_rate = Some(sample_rate);
        self
    }
    pub fn with_bits_per_sample(&mut self, bits_per_sample: u32) -> &mut Self {
        self.bits_per_sample = Some(bits_per_sample);
        self
    }
    pub fn with_max_frames_per_packet(&mut self, len: u64) -> &mut Self {
        self.max_frames_per_packet = Some(len);
        self
    }
    pub fn with_channels(&mut self, channels: Channels) -> &mut Self {
        self.channels = Some(channels);
        self
    }
}

pub struct AudioDecoderOptions {
    pub verify: bool,
    pub gapless: bool,
}

pub struct FinalizeResult {
    pub verify_ok: Option<bool>,
}

/// symphonia-bundle-flac/src/validate.rs: the MD5 of the decoded audio, only fed when `AudioDecoderOptions::verify` is set
pub struct Validator {
    fed: u64,
}

impl Validator {
    pub fn update(&mut self, buf: &AudioBuffer<i32>, _bps: u32) {
        self.fed += buf.frames() as u64;
    }
}
