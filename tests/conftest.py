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


def noise_bias_keys(keys):
    """Names (state_dict keys, reference order) of the conv biases that sit directly in front of an InstanceNorm2d: the
    normalisation removes any per-channel constant, so their true gradient is exactly zero and reference and candidate
    both hold rounding noise there.  Every OTHER bias has a real gradient and gets no absolute floor:
      * generators (`model*.N.bias`): all but the LAST conv (the 7x7 -> Tanh head, reference networks.py:207 / :163);
      * discriminators (`scale{i}_layer{j}.0.bias` or `layer{i}.N.bias`): all but the first (Conv + LeakyReLU,
        networks.py:343) and the last (1-channel head, :359) conv of every scale."""
    biases = [k for k in keys if k.endswith(".bias")]
    if not biases:
        return set()
    if biases[0].startswith(("scale", "layer")):
        scales = {}
        for k in biases:
            scales.setdefault(k.split("_")[0].split(".")[0], []).append(k)
        real = set()
        for ks in scales.values():
            real.add(ks[0]); real.add(ks[-1])
        return set(biases) - real
    return set(biases[:-1])


def assert_grad_close(key, got, ref, rtol=1e-4, bias_floor=2e-2, noise_biases=None):
    """Relative L2 check for parameter gradients.  `noise_biases` (from noise_bias_keys) names the conv biases in front
    of an InstanceNorm, whose gradient is pure rounding noise: only those are compared against the absolute
    `bias_floor`.  Without `noise_biases` no bias gets a floor unless the caller passes one explicitly for that key."""
    got = np.asarray(got, dtype=np.float64)
    ref = np.asarray(ref, dtype=np.float64)
    assert got.shape == ref.shape, key
    name = key.split(":", 1)[-1]
    if noise_biases is None:
        floor = bias_floor if (key.endswith(".bias") and bias_floor != 2e-2) else 1e-6
    else:
        floor = bias_floor if name in noise_biases else 1e-6
    err = np.linalg.norm(got - ref)
    assert err <= rtol * np.linalg.norm(ref) + floor, (key, err, np.linalg.norm(ref))
