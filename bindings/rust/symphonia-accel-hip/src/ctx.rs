//! Safe ownership of a `symaccel_ctx` and of page-locked batch buffers.
use std::ffi::CStr;
use std::ptr;
use std::sync::{Arc, Mutex};

use symphonia_core::errors::{Error, Result};

use crate::ffi;

/// Map a symaccel status to the reference's error convention (include/symaccel.h, "Conventions";
/// symphonia-core/src/errors.rs:38-54): INVALID_ARG is the assert!/panic class, UNSUPPORTED -> Error::Unsupported,
/// DECODE -> Error::DecodeError (discard the packet, keep going), DEVICE / OOM -> Error::IoError.
pub(crate) fn check(status: i32, ctx: *const ffi::SymaccelCtx) -> Result<()> {
    if status >= 0 {
        return Ok(());
    }
    // SAFETY: symaccel_strerror returns a pointer to a static NUL-terminated string.
    let msg: &'static str = unsafe { CStr::from_ptr(ffi::symaccel_strerror(status)) }.to_str().unwrap_or("symaccel: error");
    match status {
        ffi::SYMACCEL_ERR_INVALID_ARG => panic!("{msg}"),
        ffi::SYMACCEL_ERR_UNSUPPORTED => Err(Error::Unsupported(msg)),
        ffi::SYMACCEL_ERR_DECODE => Err(Error::DecodeError(msg)),
        _ => {
            let detail = if ctx.is_null() {
                String::new()
            }
            else {
                // SAFETY: the context outlives this call; the string lives inside it.
                let text = unsafe { CStr::from_ptr(ffi::symaccel_last_error(ctx)) };
                text.to_string_lossy().into_owned()
            };
            Err(Error::IoError(std::io::Error::other(format!("{msg}: {detail}"))))
        }
    }
}

/// One HIP device + stream + constant tables.  `&mut self` on the decoders gives the external synchronisation the C
/// ABI asks for; the raw pointer is `Send + Sync` because the library is thread-safe across contexts and a context is
/// only ever touched through `&mut`.
pub struct Context {
    raw: *mut ffi::SymaccelCtx,
}

unsafe impl Send for Context {}
unsafe impl Sync for Context {}

impl Context {
    pub fn new(device: i32) -> Result<Self> {
        let mut raw = ptr::null_mut();
        // SAFETY: `raw` is a valid out-pointer.
        check(unsafe { ffi::symaccel_ctx_create(device, &mut raw) }, ptr::null())?;
        Ok(Context { raw })
    }

    pub(crate) fn raw(&self) -> *mut ffi::SymaccelCtx {
        self.raw
    }
}

impl Drop for Context {
    fn drop(&mut self) {
        // SAFETY: created by symaccel_ctx_create, destroyed once.
        unsafe { ffi::symaccel_ctx_destroy(self.raw) }
    }
}

/// The process-wide cross-stream batcher (csrc/batcher.cpp): ONE context and one `symaccel_batcher` shared by every pooled
/// decoder of the process.  A decoder cannot see its siblings (codecs/audio.rs:279-297, registry.rs:330-341); the pool is where
/// their look-ahead batches meet and go to the device in one launch.  The batcher is thread-safe; the pool's context is driven
/// through the batcher only.
pub struct Pool {
    batcher: *mut ffi::SymaccelBatcher,
    ctx: Context, // (declared after `batcher`: Drop destroys the batcher first, then the field drops the context)
}

unsafe impl Send for Pool {}
unsafe impl Sync for Pool {}

static POOL: Mutex<Option<Arc<Pool>>> = Mutex::new(None);
static POOL_DEVICE: Mutex<i32> = Mutex::new(0);

impl Pool {
    /// The device the shared pool is created on (default 0).  One process per GPU: a launcher that does not narrow the visible
    /// devices per rank calls this with the local rank BEFORE `register()` / the first decoder; once the pool exists it stays
    /// where it is and the call returns `false`.
    pub fn set_device(device: i32) -> bool {
        let slot = POOL.lock().expect("pool poisoned");
        if slot.is_some() {
            return false;
        }
        *POOL_DEVICE.lock().expect("pool poisoned") = device;
        true
    }

    /// The shared pool, created on first use (the device of `set_device`, the library's default flush size).
    pub fn shared() -> Result<Arc<Pool>> {
        let mut slot = POOL.lock().expect("pool poisoned");
        if let Some(pool) = slot.as_ref() {
            return Ok(pool.clone());
        }
        let device = *POOL_DEVICE.lock().expect("pool poisoned");
        let ctx = Context::new(device)?;
        let mut batcher = ptr::null_mut();
        // SAFETY: valid context, valid out-pointer.
        check(unsafe { ffi::symaccel_batcher_create(ctx.raw(), 0, &mut batcher) }, ctx.raw())?;
        let pool = Arc::new(Pool { batcher, ctx });
        *slot = Some(pool.clone());
        Ok(pool)
    }

    pub(crate) fn raw(&self) -> *mut ffi::SymaccelBatcher {
        self.batcher
    }

    pub(crate) fn ctx_raw(&self) -> *mut ffi::SymaccelCtx {
        self.ctx.raw()
    }
}

/// A reservation with the cross-stream batcher (`symaccel_batcher_reserve`): a slot of page-locked staging memory the front end's
/// output is written STRAIGHT into -- the planes of the entry point the batch kind stands for, chain-major over this submission's
/// chains (include/symaccel.h, "cross-stream batcher") -- and the ticket that names it.  `Pool::commit` says it is filled,
/// `Pool::wait` blocks until `out` / `state` hold the PCM and the state after the batch (whatever the decoders of the OTHER streams
/// had pending went to the device in the same launch), `Pool::release` gives the slot back.  Nothing is copied between the parser's
/// output and the DMA source, and nothing between the DMA target and `AudioBuffer`'s planes but `publish`'s own plane copies.
pub struct BatchSlot {
    raw: ffi::SymaccelBatchSlot,
    ticket: u64,
    committed: bool,
}

// SAFETY: the slot's planes are owned by this reservation until `Pool::release`; the batcher itself is thread-safe.
unsafe impl Send for BatchSlot {}
unsafe impl Sync for BatchSlot {}

impl BatchSlot {
    /// Plane `i` of the submission's input as elements of `T` (the kind's layout: f32 spectra, u8 flags, #[repr(C)] records ...).
    pub fn input<T: Copy>(&mut self, i: usize) -> &mut [T] {
        // SAFETY: the batcher handed out `input_bytes[i]` bytes at `input[i]`, 256-byte aligned, exclusively ours until release.
        unsafe { std::slice::from_raw_parts_mut(self.raw.input[i] as *mut T, self.raw.input_bytes[i] / std::mem::size_of::<T>()) }
    }

    /// State plane `i`: the state the batch starts from (written before `commit`), the state it left (after `wait`).
    pub fn state<T: Copy>(&mut self, i: usize) -> &mut [T] {
        // SAFETY: as `input`.
        unsafe { std::slice::from_raw_parts_mut(self.raw.state[i] as *mut T, self.raw.state_bytes[i] / std::mem::size_of::<T>()) }
    }

    /// The result plane (after `wait`).  FLAC / ALAC work in place: it is input plane 0.
    pub fn out<T: Copy>(&self) -> &[T] {
        // SAFETY: as `input`; written by the device before `wait` returned.
        unsafe { std::slice::from_raw_parts(self.raw.out as *const T, self.raw.out_bytes / std::mem::size_of::<T>()) }
    }

    pub fn is_committed(&self) -> bool {
        self.committed
    }
}

impl Pool {
    /// `check` for calls into the batcher: a device error's text is the BATCHER's (`symaccel_batcher_last_error`, copied under its
    /// mutex) -- the shared context's own error string may be written by another thread's launch at any time.
    fn check(&self, status: i32) -> Result<()> {
        if status >= 0 || status == ffi::SYMACCEL_ERR_INVALID_ARG || status == ffi::SYMACCEL_ERR_UNSUPPORTED || status == ffi::SYMACCEL_ERR_DECODE {
            return check(status, ptr::null());
        }
        let mut text = [0 as core::ffi::c_char; 256];
        // SAFETY: a live batcher; `text` has room for 256 bytes including the terminator the library writes.
        unsafe { ffi::symaccel_batcher_last_error(self.batcher, text.as_mut_ptr(), text.len()) };
        // SAFETY: NUL-terminated by the library (or all zeros).
        let detail = unsafe { CStr::from_ptr(text.as_ptr()) }.to_string_lossy().into_owned();
        // SAFETY: symaccel_strerror returns a pointer to a static NUL-terminated string.
        let msg = unsafe { CStr::from_ptr(ffi::symaccel_strerror(status)) }.to_str().unwrap_or("symaccel: error");
        Err(Error::IoError(std::io::Error::other(format!("{msg}: {detail}"))))
    }

    /// `symaccel_batcher_reserve`: a slot for `n_chains` chains of `units` frames / granules / blocks / words each.
    pub fn reserve(&self, kind: i32, param: i32, n_chains: usize, units: usize) -> Result<BatchSlot> {
        let mut raw = ffi::SymaccelBatchSlot {
            input: [ptr::null_mut(); 6],
            state: [ptr::null_mut(); 3],
            out: ptr::null_mut(),
            input_bytes: [0; 6],
            state_bytes: [0; 3],
            out_bytes: 0,
        };
        let mut ticket = 0u64;
        // SAFETY: a live batcher, valid out-pointers.
        self.check(unsafe { ffi::symaccel_batcher_reserve(self.batcher, kind, param, n_chains, units, &mut raw, &mut ticket) })?;
        Ok(BatchSlot { raw, ticket, committed: false })
    }

    /// The slot is filled: it goes to the device with the next launch of its group.
    pub fn commit(&self, slot: &mut BatchSlot) -> Result<()> {
        // SAFETY: a live ticket of this batcher.
        self.check(unsafe { ffi::symaccel_batcher_commit(self.batcher, slot.ticket) })?;
        slot.committed = true;
        Ok(())
    }

    /// Block until the submission's results are in its slot.  The status is this submission's own: a batch whose descriptors do
    /// not add up fails alone, the neighbours of its launch do not (include/symaccel.h, "Status is kept PER TICKET").
    pub fn wait(&self, slot: &mut BatchSlot) -> Result<()> {
        // SAFETY: a live, committed ticket; the slot record is filled in again with the same pointers.
        self.check(unsafe { ffi::symaccel_batcher_wait(self.batcher, slot.ticket, &mut slot.raw) })
    }

    /// Give the slot back (a reservation that was never committed runs as zeros with its group; nobody looks at the result).
    pub fn release(&self, slot: BatchSlot) {
        // SAFETY: a live ticket; the batcher drains whatever of it is in flight before the memory is reused.
        unsafe { ffi::symaccel_batcher_release(self.batcher, slot.ticket) };
    }

    /// "Results will be wanted soon": pending groups worth a launch of their own go to the device now.
    pub fn hint(&self) {
        // SAFETY: a live batcher.
        unsafe { ffi::symaccel_batcher_hint(self.batcher) };
    }

    /// `symaccel_batcher_get_stats`: submissions, launches, what the callers waited for (mutex, lanes, completion flags).
    pub fn stats(&self) -> Result<ffi::SymaccelBatcherStats> {
        let mut s = ffi::SymaccelBatcherStats {
            submissions: 0,
            launches: 0,
            chunks: 0,
            chains_launched: 0,
            max_chains_per_launch: 0,
            staging_bytes: 0,
            pending: 0,
            failed_tickets: 0,
            lanes: 0,
            mutex_wait_ns: 0,
            mutex_contended: 0,
            launch_host_ns: 0,
            lane_wait_ns: 0,
            launch_api_ns: 0,
            group_allocs: 0,
            flag_wait_ns: 0,
            slots_peak: 0,
            blocks: 0,
            commit_to_launch_ns: 0,
            waits: 0,
            waits_blocked: 0,
            launch_to_done_ns: 0,
            launches_timed: 0,
        };
        // SAFETY: a live batcher, a valid out-pointer.
        self.check(unsafe { ffi::symaccel_batcher_get_stats(self.batcher, &mut s) })?;
        Ok(s)
    }

    /// Register a floor-1 configuration of a Vorbis stream's setup header; the index goes into the `floor` plane of the stream's
    /// `SYMACCEL_BATCH_VORBIS_DECODE` submissions.  The same configuration gives the same index in every stream.
    pub fn vorbis_floor(&self, cfg: &ffi::SymaccelVorbisFloor1Cfg) -> Result<u8> {
        let mut index: i32 = -1;
        // SAFETY: a live batcher, a valid record, a valid out-pointer.
        self.check(unsafe { ffi::symaccel_batcher_vorbis_floor(self.batcher, cfg, &mut index) })?;
        Ok(index as u8)
    }
}

impl Drop for Pool {
    fn drop(&mut self) {
        // SAFETY: created by symaccel_batcher_create, destroyed once, before its context.
        unsafe { ffi::symaccel_batcher_destroy(self.batcher) };
    }
}

/// Page-locked host memory (symaccel_host_alloc): what the staged host entry points need to overlap H2D, kernels and D2H.
pub struct Pinned<T: Copy> {
    ptr: *mut T,
    len: usize,
}

unsafe impl<T: Copy + Send> Send for Pinned<T> {}
unsafe impl<T: Copy + Sync> Sync for Pinned<T> {}

impl<T: Copy + Default> Pinned<T> {
    pub fn new(len: usize) -> Result<Self> {
        let mut p: *mut core::ffi::c_void = ptr::null_mut();
        // SAFETY: valid out-pointer; the allocation is released in Drop.
        check(unsafe { ffi::symaccel_host_alloc(len.max(1) * std::mem::size_of::<T>(), &mut p) }, ptr::null())?;
        let ptr = p as *mut T;
        for i in 0..len {
            // SAFETY: inside the allocation.
            unsafe { ptr.add(i).write(T::default()) };
        }
        Ok(Pinned { ptr, len })
    }
}

impl<T: Copy> Pinned<T> {
    pub fn as_slice(&self) -> &[T] {
        // SAFETY: `len` initialised elements.
        unsafe { std::slice::from_raw_parts(self.ptr, self.len) }
    }

    pub fn as_mut_slice(&mut self) -> &mut [T] {
        // SAFETY: unique access through &mut self.
        unsafe { std::slice::from_raw_parts_mut(self.ptr, self.len) }
    }
}

impl<T: Copy> Drop for Pinned<T> {
    fn drop(&mut self) {
        // SAFETY: allocated by symaccel_host_alloc.
        unsafe { ffi::symaccel_host_free(self.ptr as *mut core::ffi::c_void) };
    }
}
