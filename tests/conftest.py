import os
import sys

import pytest

os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")   # as the library asks when it is loaded first (csrc/api.hip): before HIP initialises

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden():
    import json
    with open(os.path.join(ROOT, "tests", "golden", "reference_fixture.json")) as f:
        return json.load(f)
