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


@pytest.fixture(scope="session")
def golden_mdct():
    return np.load(os.path.join(GOLDEN, "mdct4.npz"))


@pytest.fixture(scope="session")
def golden_networks():
    return np.load(os.path.join(GOLDEN, "networks.npz"))


@pytest.fixture(scope="session")
def golden_model():
    return np.load(os.path.join(GOLDEN, "model_step.npz"))


def mdct_cases(g):
    out = []
    for row in g["cases"]:
        name, n_fft, hop, win, center, shape = str(row).split(",")
        out.append((name, int(n_fft), int(hop), int(win), bool(int(center)), tuple(int(s) for s in shape.split("x"))))
    return out


def rel_err(a, b):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-300))


def cosine(a, b):
    a = np.asarray(a, dtype=np.float64).ravel()
    b = np.asarray(b, dtype=np.float64).ravel()
    return float(a @ b / max(np.linalg.norm(a) * np.linalg.norm(b), 1e-300))


def assert_grad_close(key, got, ref, rtol=1e-4, bias_floor=2e-2):
    """Relative L2 check for parameter gradients.  Conv biases that feed an InstanceNorm have an
    exactly-zero true gradient, so reference and candidate both hold pure rounding noise there:
    those are compared against an absolute floor instead."""
    got = np.asarray(got, dtype=np.float64)
    ref = np.asarray(ref, dtype=np.float64)
    assert got.shape == ref.shape, key
    floor = bias_floor if key.endswith(".bias") else 1e-6
    err = np.linalg.norm(got - ref)
    assert err <= rtol * np.linalg.norm(ref) + floor, (key, err, np.linalg.norm(ref))
