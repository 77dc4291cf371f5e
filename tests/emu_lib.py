"""Loads the TEST-ONLY CPU emulation build of the kernels (tests/emu) behind the normal binding."""
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "tests" / "emu"))

_lib = None


def emu_library():
    global _lib
    if _lib is None:
        import os
        import build_emu
        from symphonia_amd import Library
        # SYMACCEL_EMU_SANITIZED=asan|tsan: the build whose host C++ is compiled under a sanitizer (tests/test_sanitizers.py runs the
        # batcher suites that way, with the sanitizer's runtime pre-loaded into the interpreter)
        kind = os.environ.get("SYMACCEL_EMU_SANITIZED", "")
        _lib = Library(build_emu.build_sanitized(kind) if kind else build_emu.build())
    return _lib


@pytest.fixture(scope="module")
def emu_ctx():
    from symphonia_amd import Context
    ctx = Context(0, library=emu_library())
    yield ctx
    ctx.close()
