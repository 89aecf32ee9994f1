import json
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


@pytest.fixture(scope="session")
def arrangements():
    return json.load(open(ROOT / "tests" / "golden" / "arrangements.json"))


@pytest.fixture(scope="session", autouse=True)
def _build_native():
    """Build the oracle, the engine library and the test-only host emulation once per session."""
    import os

    import __graft_entry__ as g

    if os.environ.get("UPR_SKIP_BUILD") != "1":     # (development loops only: the libraries are already built)
        g.build()
