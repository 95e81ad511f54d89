"""pytest configuration: registers the ``gpu`` marker and makes the repo root
importable (``oracle`` package, ``__graft_entry__`` loader for the
``flow-timesnet_amd`` directory, which is not a legal Python identifier)."""
import json
import sys
from pathlib import Path

import numpy as np
import pytest

ROOT = Path(__file__).resolve().parents[1]
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))

GOLDEN = ROOT / "tests" / "golden"


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


@pytest.fixture(scope="session")
def manifest():
    return json.loads((GOLDEN / "manifest.json").read_text())["cases"]


@pytest.fixture(scope="session")
def golden():
    def load(name):
        with np.load(GOLDEN / f"{name}.npz") as z:
            return {k: z[k] for k in z.files}
    return load


@pytest.fixture(scope="session")
def ftn():
    """The product package (imported through the loader in __graft_entry__)."""
    import __graft_entry__ as ge
    return ge.load_package()
