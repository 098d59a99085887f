import os
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def gpu():
    """Initialise libsynthhip on cuda:0.  GPU tests FAIL (not skip) when the HIP library or the
    device is missing: a silent fallback would void every parity claim."""
    from synthesizer_amd import _native as N
    N.ensure_init(int(os.environ.get("SYNTHHIP_DEVICE", "0")))
    return N
