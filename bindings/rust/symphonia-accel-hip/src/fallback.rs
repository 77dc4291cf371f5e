//! The decoders this crate was registered above.
//!
//! The registry answers with the most preferred entry and calls its factory; when that factory fails the error goes to
//! the application -- there is no fall-through to the next tier (codecs/registry.rs:152-154, 330-341).  A `Tier::Preferred`
//! decoder that cannot be built on this machine (no GPU, no memory, no parse stage for the codec) would therefore turn a
//! decodable track into an error.  So `register` remembers, per codec id, the factory the registry would have used
//! without this crate, and `try_registry_new` hands the track to it when the accelerated decoder cannot be constructed.
use std::collections::HashMap;
use std::sync::Mutex;

use symphonia_core::codecs::audio::{AudioCodecId, AudioCodecParameters, AudioDecoder, AudioDecoderOptions};
use symphonia_core::codecs::registry::{AudioDecoderFactoryFn, CodecRegistry};
use symphonia_core::errors::{Error, Result};

/// Per codec id: `Some(factory)` = the decoder that was in force when this crate registered above it, `None` = there was none.
///
/// Process-wide BY CONSTRUCTION: `try_registry_new(params, opts)` (codecs/registry.rs:34-44) is not told which registry is
/// asking, so the table cannot be keyed by registry.  An application that holds several registries with DIFFERENT decoders
/// below this crate for the same codec gets the first one's decoder as the fall-back in all of them; registering this crate
/// in one registry only (or building those registries from the same base set, the common case:
/// `symphonia::default::get_codecs()`) avoids it.
static BELOW: Mutex<Option<HashMap<AudioCodecId, Option<AudioDecoderFactoryFn>>>> = Mutex::new(None);

/// Record what `registry` answers for `ids` right now (call before registering above it).  The FIRST lookup wins, also when
/// it finds nothing: a second `register` call on the same registry would otherwise find this crate's own factory there, and
/// `make` would call it, which calls `make` again.
pub fn remember(registry: &CodecRegistry, ids: &[AudioCodecId]) {
    let mut below = BELOW.lock().expect("fallback table poisoned");
    let map = below.get_or_insert_with(HashMap::new);
    for id in ids {
        if !map.contains_key(id) {
            let found = registry.get_audio_decoder(*id).map(|r| r.factory);
            map.insert(*id, found);
        }
    }
}

/// The factory recorded for `id`, if any.
pub fn factory_below(id: AudioCodecId) -> Option<AudioDecoderFactoryFn> {
    let below = BELOW.lock().expect("fallback table poisoned");
    match below.as_ref() {
        Some(map) => match map.get(&id) {
            Some(entry) => *entry,
            None => None,
        },
        None => None,
    }
}

/// Build the decoder that was registered below this crate for `params.codec`; `why` (the reason the accelerated decoder
/// could not be built) is returned when there is none.
pub fn make(params: &AudioCodecParameters, opts: &AudioDecoderOptions, why: Error) -> Result<Box<dyn AudioDecoder>> {
    match factory_below(params.codec) {
        Some(factory) => {
            log::warn!("symphonia-accel-hip: {why}; the CPU decoder takes this track");
            factory(params, opts)
        }
        None => Err(why),
    }
}
