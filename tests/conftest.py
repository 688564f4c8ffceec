import json
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden():
    with open(ROOT / "tests" / "golden" / "reference_vectors.json") as f:
        return json.load(f)


def golden_vectors():
    with open(ROOT / "tests" / "golden" / "reference_vectors.json") as f:
        return json.load(f)["vectors"]


@pytest.fixture
def knob(monkeypatch):
    """knob(CST_LANE_GEO="small"): sets debug switches of the library for ONE test.  The library reads its CST_* switches once, when
    it is loaded (include/constriction_amd.h, "Debug switches"); cst_debug_reload_knobs makes it read them again."""
    from constriction_amd import _native

    def set_knobs(**kw):
        for k, v in kw.items():
            monkeypatch.setenv(k, str(v))
        _native.reload_knobs()
    yield set_knobs
    monkeypatch.undo()
    _native.reload_knobs()
