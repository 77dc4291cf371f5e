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
        import build_emu
        from symphonia_amd import Library
        _lib = Library(build_emu.build())
    return _lib


@pytest.fixture(scope="module")
def emu_ctx():
    from symphonia_amd import Context
    ctx = Context(0, library=emu_library())
    yield ctx
    ctx.close()
