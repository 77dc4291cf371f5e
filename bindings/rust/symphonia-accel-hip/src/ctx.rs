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

impl Pool {
    /// The shared pool, created on first use (device 0, the library's default flush size).
    pub fn shared() -> Result<Arc<Pool>> {
        let mut slot = POOL.lock().expect("pool poisoned");
        if let Some(pool) = slot.as_ref() {
            return Ok(pool.clone());
        }
        let ctx = Context::new(0)?;
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
