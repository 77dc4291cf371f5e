//! TEST ONLY.  A mock codec and a mock demuxer for executing bindings/rust/symphonia-accel-hip/src/lookahead.rs under
//! tools/rsinterp (tests/test_rust_shim.py): the Rust counterpart of tests/cpp/lookahead_test.cpp's harness.

/// A codec with a one-packet memory, like every codec on the path: the audio of a packet is a function of the packet's
/// value and of the value of the packet decoded before it; reset() forgets.  A payload whose first byte is 255 does not
/// parse (the reference's DecodeError); `fail_transform_in` batches from now, transform fails (a device error).
pub struct MockCodec {
    pub state: i64,
    pub batch_out: Vec<i64>,
    pub buffer: Option<i64>,
    pub batch_sizes: Vec<i64>,
    pub parses: i64,
    pub fail_transform_in: i64,
    // the cross-stream batcher form (BatchCodec::pooled / submit / collect / hint / abandon)
    pub pool: bool,
    pub submitted: Vec<i64>,
    pub in_flight: bool,
    pub submits: i64,
    pub collects: i64,
    pub hints: i64,
    pub abandons: i64,
    pub fail_submit_in: i64,
}

impl MockCodec {
    pub fn new() -> Self {
        MockCodec {
            state: 0,
            batch_out: Vec::new(),
            buffer: None,
            batch_sizes: Vec::new(),
            parses: 0,
            fail_transform_in: -1,
            pool: false,
            submitted: Vec::new(),
            in_flight: false,
            submits: 0,
            collects: 0,
            hints: 0,
            abandons: 0,
            fail_submit_in: -1,
        }
    }

    pub fn new_pooled() -> Self {
        let mut c = MockCodec::new();
        c.pool = true;
        c
    }
}

impl BatchCodec for MockCodec {
    type Parsed = i64;

    fn parse(&mut self, packet: &PacketRef<'_>) -> Result<i64> {
        self.parses += 1;
        if packet.data[0] == 255 {
            return decode_error("mock: corrupt packet");
        }
        Ok(packet.data[0] as i64 + 256 * (packet.data[1] as i64))
    }

    fn transform(&mut self, batch: &[i64]) -> Result<()> {
        if self.fail_transform_in == 0 {
            self.fail_transform_in = -1;
            return Err(Error::IoError("mock: device error"));
        }
        if self.fail_transform_in > 0 {
            self.fail_transform_in -= 1;
        }
        self.batch_out.clear();
        for v in batch.iter() {
            self.batch_out.push(*v * 100000 + self.state);
            self.state = *v;
        }
        self.batch_sizes.push(batch.len() as i64);
        Ok(())
    }

    fn publish(&mut self, i: usize) {
        self.buffer = Some(self.batch_out[i]);
    }

    fn reset_state(&mut self) {
        self.state = 0;
    }

    fn clear(&mut self) {
        self.buffer = None;
    }

    fn pooled(&self) -> bool {
        self.pool
    }

    fn submit(&mut self, batch: &[i64]) -> Result<()> {
        if self.in_flight {
            return Err(Error::IoError("mock: one batch at a time"));
        }
        if self.fail_submit_in == 0 {
            self.fail_submit_in = -1;
            return Err(Error::IoError("mock: submit failed"));
        }
        if self.fail_submit_in > 0 {
            self.fail_submit_in -= 1;
        }
        self.submitted.clear();
        for v in batch.iter() {
            self.submitted.push(*v);
        }
        self.in_flight = true;
        self.submits += 1;
        Ok(())
    }

    fn collect(&mut self) -> Result<()> {
        if !self.in_flight {
            return Err(Error::IoError("mock: nothing submitted"));
        }
        self.in_flight = false;
        self.collects += 1;
        let batch = self.submitted.clone();
        self.transform(&batch)
    }

    fn hint(&mut self) {
        self.hints += 1;
    }

    fn abandon(&mut self) {
        if self.in_flight {
            self.abandons += 1;
        }
        self.in_flight = false;
        self.submitted.clear();
    }
}

/// The inner `FormatReader`: a list of packets (possibly of several tracks, interleaved) and a read position.
pub struct MockReader {
    pub packets: Vec<Packet>,
    pub pos: usize,
    pub reads: i64,
    pub fail_at: i64,
}

impl MockReader {
    pub fn new(packets: Vec<Packet>) -> Self {
        MockReader { packets, pos: 0, reads: 0, fail_at: -1 }
    }

    pub fn next_packet(&mut self) -> Result<Option<Packet>> {
        self.reads += 1;
        if self.fail_at == self.pos as i64 {
            self.fail_at = -1;
            return Err(Error::IoError("mock: read error"));
        }
        if self.pos >= self.packets.len() {
            return Ok(None);
        }
        let p = self.packets[self.pos].clone();
        self.pos += 1;
        Ok(Some(p))
    }

    /// (the reference's `seek(mode, to)` returns a `SeekedTo`; here: the packet index)
    pub fn seek(&mut self, _mode: i64, to: usize) -> Result<usize> {
        self.pos = to;
        Ok(to)
    }
}

pub fn make_packet(track_id: u32, pts: i64, value: i64) -> Packet {
    let mut data = Vec::new();
    data.push((value % 256) as u8);
    data.push((value / 256) as u8);
    Packet::new(track_id, Timestamp::new(pts), data)
}

pub fn corrupt_packet(track_id: u32, pts: i64) -> Packet {
    let mut data = Vec::new();
    data.push(255u8);
    data.push(0u8);
    Packet::new(track_id, Timestamp::new(pts), data)
}

pub fn cpu_factory(params: &AudioCodecParameters, _opts: &AudioDecoderOptions) -> Result<i64> {
    Ok(1000 + params.codec.0 as i64)
}

pub fn other_cpu_factory(params: &AudioCodecParameters, _opts: &AudioDecoderOptions) -> Result<i64> {
    Ok(2000 + params.codec.0 as i64)
}

/// what `try_registry_new` of a Hip*Decoder does when its front end / device is missing (decoder.rs, hip_decoder!)
pub fn hip_factory(params: &AudioCodecParameters, opts: &AudioDecoderOptions) -> Result<i64> {
    match unsupported_error::<i64>("mock: no front end") {
        Ok(decoder) => Ok(decoder),
        Err(e) => make(params, opts, e),
    }
}
