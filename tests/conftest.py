import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

TESTS = os.path.join(ROOT, "tests")
if TESTS not in sys.path:
    sys.path.insert(0, TESTS)          # `from helpers.compare import grad_close`

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run on the GPU box with -m gpu)")


@pytest.fixture(scope="session")
def golden():
    import numpy as np

    def load(name):
        return np.load(os.path.join(GOLDEN, name), allow_pickle=False)

    return load


def pytest_sessionfinish(session, exitstatus):
    """Keep the gradient-comparison statistics of a GPU run (how many entries used the near-tie allowance)."""
    try:
        from helpers import compare
        out = os.path.join(ROOT, "gpurun_out")
        if compare.RECORD and os.path.isdir(out):
            compare.dump_record(os.path.join(out, "grad_close_counts.json"))
    except Exception:
        pass
