//! TEST ONLY.  Scripted CPU front ends for executing the codec adapters of bindings/rust/symphonia-accel-hip under tools/rsinterp
//! (tests/test_rust_adapters.py): each implements the adapter's `*FrontEnd` trait and answers `parse(packet)` with the entry of
//! a script the test filled from the reference-text fixtures (tests/golden/rs_fixtures) -- packet payload byte 0 (+ 256 * byte 1)
//! is the index into the script, a payload starting with 255 does not parse (the reference's DecodeError).

fn script_index(packet: &PacketRef<'_>) -> Result<usize> {
    if packet.data[0] == 255 {
        return decode_error("mock: corrupt packet");
    }
    Ok(packet.data[0] as usize + 256 * (packet.data[1] as usize))
}

pub struct ScriptedAacFront {
    pub params: AudioCodecParameters,
    pub nch: usize,
    pub script: Vec<ParsedAac>,
    pub parses: usize,
    pub resets: usize,
}

impl AacFrontEnd for ScriptedAacFront {
    fn params(&self) -> &AudioCodecParameters {
        &self.params
    }
    fn channels(&self) -> usize {
        self.nch
    }
    fn parse(&mut self, packet: &PacketRef<'_>) -> Result<ParsedAac> {
        let i = script_index(packet)?;
        self.parses += 1;
        Ok(ParsedAac { coeffs: self.script[i].coeffs.clone(), side: self.script[i].side.clone(), fused: None })
    }
    fn reset(&mut self) {
        self.resets += 1;
    }
}

pub struct ScriptedMpaFront {
    pub nch: usize,
    pub sr_idx: i32,
    pub script: Vec<ParsedMpa>,
    pub parses: usize,
    pub resets: usize,
}

impl MpaFrontEnd for ScriptedMpaFront {
    fn channels(&self) -> usize {
        self.nch
    }
    fn sample_rate_idx(&self) -> i32 {
        self.sr_idx
    }
    fn parse(&mut self, packet: &PacketRef<'_>) -> Result<ParsedMpa> {
        let i = script_index(packet)?;
        self.parses += 1;
        let p = &self.script[i];
        Ok(ParsedMpa { trim: p.trim, n_granules: p.n_granules, xr: p.xr.clone(), side: p.side.clone(), fused: None })
    }
    fn reset(&mut self) {
        self.resets += 1;
    }
}

pub struct ScriptedVorbisFront {
    pub nch: usize,
    pub bs0_exp: i32,
    pub bs1_exp: i32,
    pub script: Vec<ParsedVorbis>,
    pub parses: usize,
    pub resets: usize,
}

impl VorbisFrontEnd for ScriptedVorbisFront {
    fn channels(&self) -> usize {
        self.nch
    }
    fn block_exps(&self) -> (i32, i32) {
        (self.bs0_exp, self.bs1_exp)
    }
    fn parse(&mut self, packet: &PacketRef<'_>) -> Result<ParsedVorbis> {
        let i = script_index(packet)?;
        self.parses += 1;
        let p = &self.script[i];
        Ok(ParsedVorbis { trim: p.trim, long_block: p.long_block, spectra: p.spectra.clone(), fused: None })
    }
    fn floors(&self) -> Vec<SymaccelVorbisFloor1Cfg> {
        Vec::new()
    }
    fn reset(&mut self) {
        self.resets += 1;
    }
}

pub struct ScriptedFlacFront {
    pub params: AudioCodecParameters,
    pub nch: usize,
    pub max_bs: usize,
    pub script: Vec<ParsedFlac>,
    pub parses: usize,
}

impl FlacFrontEnd for ScriptedFlacFront {
    fn params(&self) -> &AudioCodecParameters {
        &self.params
    }
    fn channels(&self) -> usize {
        self.nch
    }
    fn max_blocksize(&self) -> usize {
        self.max_bs
    }
    fn parse(&mut self, packet: &PacketRef<'_>) -> Result<ParsedFlac> {
        let i = script_index(packet)?;
        self.parses += 1;
        let p = &self.script[i];
        Ok(ParsedFlac {
            blocksize: p.blocksize,
            words: p.words.clone(),
            desc: p.desc.clone(),
            coeffs: p.coeffs.clone(),
            pair_mode: p.pair_mode,
            out_shift: p.out_shift,
        })
    }
}

pub struct ScriptedAlacFront {
    pub params: AudioCodecParameters,
    pub nch: usize,
    pub max_frames: usize,
    pub script: Vec<ParsedAlac>,
    pub parses: usize,
}

impl AlacFrontEnd for ScriptedAlacFront {
    fn params(&self) -> &AudioCodecParameters {
        &self.params
    }
    fn channels(&self) -> usize {
        self.nch
    }
    fn max_frames(&self) -> usize {
        self.max_frames
    }
    fn parse(&mut self, packet: &PacketRef<'_>) -> Result<ParsedAlac> {
        let i = script_index(packet)?;
        self.parses += 1;
        let p = &self.script[i];
        Ok(ParsedAlac {
            frames: p.frames,
            words: p.words.clone(),
            desc: p.desc.clone(),
            coeffs: p.coeffs.clone(),
            pairs: p.pairs.clone(),
            tails: p.tails.clone(),
            out_shift: p.out_shift,
        })
    }
}
