//! Raw bindings, generated from include/symaccel.h by tools/gen_rust_ffi.py.
include!("../../symaccel_sys.rs");
