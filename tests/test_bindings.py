"""bindings/rust/symaccel_sys.rs -- the raw FFI a Rust shim crate would `include!` -- is generated
from include/symaccel.h by tools/gen_rust_ffi.py.  There is no Rust toolchain in the image, so the file cannot be
compiled here; these checks keep it structurally in step with the header and with the ctypes binding the tests use."""
import re
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "tools"))

import gen_rust_ffi  # noqa: E402
from symphonia_amd import _ffi, backend  # noqa: E402


def test_generated_file_is_current():
    text, _, _ = gen_rust_ffi.generate()
    assert (ROOT / "bindings" / "rust" / "symaccel_sys.rs").read_text() == text, "run python tools/gen_rust_ffi.py"


def test_every_abi_symbol_is_bound_with_the_c_parameter_count():
    text, _, funcs = gen_rust_ffi.generate()
    rust = dict(re.findall(r"pub fn (symaccel_\w+)\((.*?)\)", text))
    assert sorted(rust) == sorted(_ffi.ABI_SYMBOLS)
    for name, _, params in funcs:
        n_rust = len([p for p in rust[name].split(",") if p.strip()])
        assert n_rust == len(params), name
    # pointers keep their constness: a `const float *` must not become `*mut f32`
    assert "d_in: *const f32, d_out: *mut f32" in text
    assert "out: *mut *mut SymaccelCtx" in text


def test_record_sizes_match_the_ctypes_side():
    _, structs, _ = gen_rust_ffi.generate()
    sizes = {name: gen_rust_ffi.struct_size(fields) for name, fields in structs}
    assert sizes["symaccel_mp3_requant"] == backend.MP3_REQUANT_DTYPE.itemsize == 52
    assert sizes["symaccel_mp3_stereo"] == backend.MP3_STEREO_DTYPE.itemsize == 48
    assert sizes["symaccel_aac_js_frame"] == backend.AAC_JS_DTYPE.itemsize == 644
    assert sizes["symaccel_aac_tns_filter"] == backend.AAC_TNS_DTYPE.itemsize == 92
    assert sizes["symaccel_flac_desc"] == backend.FLAC_DESC_DTYPE.itemsize == 4
    assert sizes["symaccel_mp3_side"] == sizes["symaccel_alac_desc"] == 4
