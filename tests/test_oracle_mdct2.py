"""Pin oracle/mdct2.py (DCT_2N_native / IDCT_2N_native / MDCT2 / IMDCT2 restatement) against the reference's own KAT
(test/DCT_test.ipynb cell 34) and golden vectors produced by the reference (tools/gen_golden.py)."""
import os

import numpy as np
import pytest

from conftest import GOLDEN, mdct_cases, rel_err
from oracle import mdct2 as M2


@pytest.fixture(scope="module")
def g2():
    return np.load(os.path.join(GOLDEN, "mdct2.npz"))


def test_dct_kat(g2):
    a = np.arange(1, 17, dtype=np.float64)
    kat = np.array([17.0, -6.4741, 0, -0.70977, 0, -0.24805, 0, -0.12005, 0, -0.066354, 0, -0.037880, 0, -0.019813, 0, -0.0061858])
    assert np.max(np.abs(M2.dct_2n(a) - kat)) < 5e-5                      # printed values of the notebook
    assert np.max(np.abs(M2.dct_2n(a) - g2["kat_dct"])) < 2e-6
    assert np.allclose(M2.idct_2n(M2.dct_2n(a)), 2 * a, atol=1e-10)      # idct(dct(a)) = 2a
    assert np.allclose(g2["kat_idct_dct"], 2 * a, atol=1e-4)
    assert rel_err(M2.dct_2n(g2["dct_x"]), g2["dct_y"]) < 1e-6
    assert rel_err(M2.idct_2n(g2["idct_x"]), g2["idct_y"]) < 1e-6


def test_mdct2_imdct2(g2):
    for name, n_fft, hop, win, center, shape in mdct_cases(g2):
        w = g2[f"{name}_w"]
        S = M2.mdct2_forward(g2[f"{name}_x"], n_fft, hop, win, w, center)
        assert S.shape == g2[f"{name}_S"].shape
        assert rel_err(S, g2[f"{name}_S"]) < 2e-6, name                  # reference runs its FFT in fp32
        y = M2.imdct2_forward(g2[f"{name}_S"], n_fft, hop, win, w, center)
        assert y.shape == g2[f"{name}_y"].shape and rel_err(y, g2[f"{name}_y"]) < 2e-6, name
        yo = M2.imdct2_forward(g2[f"{name}_S"], n_fft, hop, win, w, center, out_length=shape[-1])
        assert yo.shape == g2[f"{name}_y_outlen"].shape
