import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_golden(name):
    return dict(np.load(os.path.join(GOLDEN, name), allow_pickle=False))


@pytest.fixture(scope="session")
def golden():
    return load_golden


def assert_close(actual, expected, rel=1e-5, name=""):
    """Float parity bar of BASELINE.json north_star: within `rel` relative, measured against the
    tensor's own scale (max |expected|) so that entries that are ~0 by cancellation do not
    demand absolute accuracy below fp32 rounding of the tensor's magnitude."""
    actual = np.asarray(actual, dtype=np.float64)
    expected = np.asarray(expected, dtype=np.float64)
    assert actual.shape == expected.shape, (name, actual.shape, expected.shape)
    scale = max(float(np.abs(expected).max()), 1e-30) if expected.size else 1.0
    err = float(np.abs(actual - expected).max()) / scale if expected.size else 0.0
    assert err <= rel, "%s: max err / scale = %.3e > %.1e" % (name, err, rel)
    return err
