"""CPU-side checks of the drop-in boundary: the hipcc-built library loads, exports every symbol
include/symaccel.h declares, serves the host-side table read-backs, and refuses to create a context
without a GPU (there is no CPU path).  No compute calls."""
import re
import subprocess
import sys
from pathlib import Path

import numpy as np
import pytest

import oracle
from symphonia_amd import SymaccelError, _ffi

ROOT = Path(__file__).resolve().parent.parent
HEADER = ROOT / "include" / "symaccel.h"


def declared_symbols():
    text = re.sub(r"/\*.*?\*/", "", HEADER.read_text(), flags=re.S)
    return sorted(set(re.findall(r"\b(symaccel_[a-z0-9_]+)\s*\(", text)))


@pytest.fixture(scope="module")
def lib():
    from symphonia_amd import build
    build.build()
    return _ffi.Library()


def test_header_and_binding_agree():
    assert declared_symbols() == sorted(_ffi.ABI_SYMBOLS)


def test_library_exports_every_declared_symbol(lib):
    for name in declared_symbols():
        assert hasattr(lib.dll, name), name
    assert lib.dll.symaccel_abi_version() == 8


def test_strerror_is_static_text(lib):
    assert lib.dll.symaccel_strerror(0).decode() != ""
    for st in (-1, -2, -3, -4):
        assert lib.dll.symaccel_strerror(st).decode() not in ("", lib.dll.symaccel_strerror(0).decode())


def test_host_tables_match_oracle(lib):
    """Tables are generated on the host with the reference's libm calls."""
    assert np.array_equal(lib.table(_ffi.TABLE_AAC_KBD_LONG), oracle.aac_window(True, 4.0, 1024))
    assert np.array_equal(lib.table(_ffi.TABLE_AAC_KBD_SHORT), oracle.aac_window(True, 6.0, 128))
    assert np.array_equal(lib.table(_ffi.TABLE_AAC_SINE_LONG), oracle.aac_window(False, 0.0, 1024))
    assert np.array_equal(lib.table(_ffi.TABLE_MP3_SYNTH_D), oracle.mp3_synthesis_window())
    assert np.array_equal(lib.table(_ffi.TABLE_MP3_IMDCT_WIN).reshape(4, 36), oracle.mp3_imdct_windows())
    for n, scale in ((1024, 1.0 / 2048), (128, 1.0 / 256), (1024, 1.0), (32, -2.0)):
        assert np.array_equal(lib.imdct_twiddles(n, scale).view(np.float32), oracle.imdct_twiddles(n, scale).view(np.float32))
    for n in (64, 128, 512, 4096):
        assert np.array_equal(lib.fft_twiddles(n).view(np.float32), oracle.fft_twiddles(n).view(np.float32))


def test_mp3_literal_header_matches_host_table(lib):
    """csrc/mp3_literals.h (compile-time immediates in mp3_synth_kernel) == the host-generated table, bit for bit."""
    text = (ROOT / "symphonia_amd" / "csrc" / "mp3_literals.h").read_text()
    vals = [float.fromhex(m) for m in re.findall(r"(-?0x[0-9a-f.]+p[-+]?\d+)f,", text)]
    table = lib.table(_ffi.TABLE_MP3_CONSTS)
    assert len(vals) == 264 == len(table)
    assert np.array_equal(np.array(vals, dtype=np.float32).view(np.uint32), table.view(np.uint32))
    assert np.array_equal(table[:144].reshape(4, 36), oracle.mp3_imdct_windows())


def test_no_cpu_fallback(lib):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from symphonia_amd import Context
    with pytest.raises(SymaccelError) as e:
        Context(0, library=lib)
    assert e.value.status == _ffi.ERR_DEVICE


def test_product_never_imports_the_oracle():
    """oracle/ is test infrastructure: nothing under symphonia_amd/ may import, link or execute it."""
    for path in (ROOT / "symphonia_amd").rglob("*"):
        if path.suffix in (".py", ".cpp", ".hip", ".h"):
            text = path.read_text()
            assert not re.search(r"^\s*(import|from)\s+oracle\b", text, flags=re.M), path
            assert "symoracle" not in text, path
    code = "import sys; import symphonia_amd; sys.exit(1 if 'oracle' in sys.modules else 0)"
    assert subprocess.run([sys.executable, "-c", code], cwd=str(ROOT)).returncode == 0


def test_header_is_plain_c(tmp_path):
    """include/symaccel.h is the boundary a C (or Rust bindgen / cgo-style) binding consumes: it must compile as strict C99
    on its own, and the record sizes the bindings rely on must be what the documentation says."""
    import subprocess
    src = tmp_path / "abi.c"
    src.write_text('#include "symaccel.h"\n'
                   '#define SIZE_IS(t, n) typedef char size_of_##t[(sizeof(t) == (n)) ? 1 : -1]\n'
                   'SIZE_IS(symaccel_mp3_side, 4);\nSIZE_IS(symaccel_mp3_requant, 52);\nSIZE_IS(symaccel_mp3_stereo, 48);\n'
                   'SIZE_IS(symaccel_flac_desc, 4);\nSIZE_IS(symaccel_alac_desc, 4);\nSIZE_IS(symaccel_aac_js_frame, 644);\n'
                   'SIZE_IS(symaccel_aac_tns_filter, 92);\n'
                   'int use(void) { return SYMACCEL_ABI_VERSION + (int)SYMACCEL_ERR_OOM; }\n')
    out = subprocess.run(["gcc", "-std=c99", "-pedantic", "-Wall", "-Wextra", "-Werror", "-I", str(ROOT / "include"), "-c", str(src),
                          "-o", str(tmp_path / "abi.o")], capture_output=True, text=True)
    assert out.returncode == 0, out.stderr
