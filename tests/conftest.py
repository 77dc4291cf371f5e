import os
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with gpurun / at round end)")
    config.addinivalue_line("markers", "localref: needs /root/reference (build container only)")
    config.addinivalue_line("markers", "sanitize: the host C++ under ASan + UBSan / TSan on the emulation build (CPU only; part of the default CPU suite)")


def pytest_collection_modifyitems(config, items):
    have_ref = Path("/root/reference").exists()
    for item in items:
        if "localref" in item.keywords and not have_ref:
            item.add_marker(pytest.mark.skip(reason="/root/reference not present"))
