#!/usr/bin/env python3
"""Generate the Rust FFI declarations a `symphonia-accel-hip` shim crate needs from include/symaccel.h:

    python tools/gen_rust_ffi.py            # writes bindings/rust/symaccel_sys.rs

Every `typedef struct` becomes a #[repr(C)] struct, every prototype an entry of one `unsafe extern "C"` block, every
`#define` of an integer constant a `pub const`.  The image has no Rust toolchain, so the output is NOT compiled here;
tests/test_bindings.py checks it structurally against the header (same functions, same parameter counts, same record
sizes)."""
import re
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
HEADER = ROOT / "include" / "symaccel.h"
OUT = ROOT / "bindings" / "rust" / "symaccel_sys.rs"

SCALARS = {"int": "i32", "double": "f64", "float": "f32", "size_t": "usize", "int32_t": "i32", "uint32_t": "u32",
           "int16_t": "i16", "uint16_t": "u16", "uint8_t": "u8", "int8_t": "i8", "uint64_t": "u64", "int64_t": "i64", "char": "core::ffi::c_char", "void": "core::ffi::c_void"}
SIZES = {"usize": 8, "core::ffi::c_char": 1, "u8": 1, "i8": 1, "u64": 8, "i64": 8, "i16": 2, "u16": 2, "i32": 4, "u32": 4, "f32": 4, "f64": 8}


# records that hold function pointers (the generator only understands scalar fields)
CALLBACK_STRUCTS = [
    "#[repr(C)]", "#[derive(Clone, Copy)]", "pub struct SymaccelTransport {",
    "    pub group_start: Option<unsafe extern \"C\" fn() -> i32>,",
    "    pub group_end: Option<unsafe extern \"C\" fn() -> i32>,",
    "    pub send: Option<unsafe extern \"C\" fn(d_buf: *const core::ffi::c_void, bytes: usize, peer: i32, comm: *mut core::ffi::c_void, stream: *mut core::ffi::c_void) -> i32>,",
    "    pub recv: Option<unsafe extern \"C\" fn(d_buf: *mut core::ffi::c_void, bytes: usize, peer: i32, comm: *mut core::ffi::c_void, stream: *mut core::ffi::c_void) -> i32>,",
    "}", "",
    "/// `symaccel_step_fn`: the caller's synthesis calls on streams [first, first + count) of its slice (symaccel_exchange_pipelined)",
    "pub type SymaccelStepFn = Option<unsafe extern \"C\" fn(user: *mut core::ffi::c_void, first_local_stream: usize, n_local_streams: usize) -> i32>;",
    ""]


def camel(name):
    return "".join(p.capitalize() for p in name.split("_"))


def strip_comments(text):
    return re.sub(r"/\*.*?\*/", "", text, flags=re.S)


def rust_type(ctype):
    ctype = " ".join(ctype.split())
    m = re.fullmatch(r"(const )?(\w+)( \*+|\*+)?", ctype)
    if not m:
        raise ValueError(ctype)
    const, base, ptr = m.group(1), m.group(2), (m.group(3) or "").strip()
    base = SCALARS.get(base, camel(base) if base.startswith("symaccel_") else None)
    if base is None:
        raise ValueError(ctype)
    for _ in ptr:
        base = ("*const " if const else "*mut ") + base
        const = None
    return base


def parse(text):
    text = strip_comments(text)
    structs = []
    for m in re.finditer(r"typedef struct (\w+) \{(.*?)\} (\w+);", text, flags=re.S):
        if "(*" in m.group(2):  # a table of function pointers: written out by hand (CALLBACK_STRUCTS)
            continue
        fields = []
        for decl in m.group(2).split(";"):
            decl = " ".join(decl.split())
            if not decl:
                continue
            ctype, names = decl.rsplit(" ", 1)[0], decl[len(decl.rsplit(" ", 1)[0]) + 1:]
            # `uint16_t rzero0, rzero1` / `uint8_t scalefacs[39]`
            toks = decl.split(" ")
            nt = 2 if toks[0] == "const" else 1  # `const uint32_t *x_list`
            first, *rest = [n.strip() for n in " ".join(toks[nt:]).split(",")]
            ctype = " ".join(toks[:nt])
            for n in [first] + rest:
                ft = ctype
                while n.startswith("*"):  # `void *in[4]`: a pointer field
                    ft, n = ft + " *", n[1:].strip()
                am = re.fullmatch(r"(\w+)\[(\d+)\]", n)
                fields.append((am.group(1), ft, int(am.group(2))) if am else (n, ft, None))
        structs.append((m.group(3), fields))
    funcs = []
    for m in re.finditer(r"^(int|void|size_t|const char \*)\s*(symaccel_\w+)\(([^;{]*?)\);", text, flags=re.S | re.M):
        params = []
        body = " ".join(m.group(3).split())
        if body not in ("void", ""):
            for p in body.split(","):
                p = p.strip()
                pm = re.fullmatch(r"(.*?)(\w+)", p)
                params.append((pm.group(2), pm.group(1).strip()))
        funcs.append((m.group(2), m.group(1).strip(), params))
    consts = [(n, int(v.rstrip("u"), 0)) for n, v in re.findall(r"^#define (SYMACCEL_\w+) (-?\d+u?|0x[0-9a-fA-F]+u?)\s*$", text, flags=re.M)]
    enums = []
    for m in re.finditer(r"(\bSYMACCEL_\w+) = (-?\d+)", text):
        enums.append((m.group(1), int(m.group(2))))
    return structs, funcs, consts, enums


def struct_size(fields):
    size, align = 0, 1
    for _, ctype, count in fields:
        s = 8 if ctype.endswith("*") else SIZES[SCALARS[ctype]]
        align = max(align, s)
        size = (size + s - 1) // s * s + s * (count or 1)
    return (size + align - 1) // align * align


def generate():
    structs, funcs, consts, enums = parse(HEADER.read_text())
    out = ["// GENERATED by tools/gen_rust_ffi.py from include/symaccel.h -- do not edit.",
           "// Raw FFI of libsymaccel for a `symphonia-accel-hip` shim crate (INTEGRATION.md section 2).",
           "// Not compiled in the repository's image (no Rust toolchain); checked structurally by tests/test_bindings.py.",
           "#![allow(non_camel_case_types, dead_code)]", "",
           "#[repr(C)]", "pub struct SymaccelCtx {", "    _private: [u8; 0],", "}", "",
           "#[repr(C)]", "pub struct SymaccelBatcher {", "    _private: [u8; 0],", "}", ""]
    for name, value in enums + consts:
        out.append("pub const %s: %s = %d;" % (name, "i32" if value < 0 or name.startswith("SYMACCEL_ERR") or name == "SYMACCEL_OK" else "u32", value))
    out.append("")
    for name, fields in structs:
        out += ["#[repr(C)]", "#[derive(Clone, Copy)]", "pub struct %s {" % camel(name)]
        for fname, ctype, count in fields:
            rt = rust_type(ctype) if ctype.endswith("*") else SCALARS[ctype]
            out.append("    pub %s: %s," % (fname, "[%s; %d]" % (rt, count) if count else rt))
        out += ["}", "const _: () = assert!(core::mem::size_of::<%s>() == %d);" % (camel(name), struct_size(fields)), ""]
    out += CALLBACK_STRUCTS
    out += ['#[link(name = "symaccel")]', 'unsafe extern "C" {']
    for name, ret, params in funcs:
        args = ", ".join("%s: %s" % (pn, rust_type(pt)) for pn, pt in params)
        rret = {"int": " -> i32", "void": "", "size_t": " -> usize", "const char *": " -> *const core::ffi::c_char"}[ret]
        out.append("    pub fn %s(%s)%s;" % (name, args, rret))
    out += ["}", ""]
    return "\n".join(out), structs, funcs


def main():
    text, _, _ = generate()
    OUT.parent.mkdir(parents=True, exist_ok=True)
    OUT.write_text(text)
    print("wrote", OUT)


if __name__ == "__main__":
    main()
