"""Oracle (oracle/evaltail.py) against the reference's own util.imdct / compute_matrics outputs (tests/golden/evaltail.npz,
made by tools/gen_golden.py), plus the restated STFT against torch.stft."""
import os

import numpy as np
import pytest
import torch

from oracle import evaltail as E
from oracle import mdct4 as M

G = np.load(os.path.join(os.path.dirname(__file__), "golden", "evaltail.npz"))
G4 = np.load(os.path.join(os.path.dirname(__file__), "golden", "mdct4.npz"))

IMDCT_TAGS = ["ex_b2_u6", "ex_b1_u3", "ex_b2_u1", "pl_b2_u1", "pl_b2_u6"]
MET_TAGS = ["n64_b3", "n64_1d", "n1024_b2", "n64_nc"]


def test_keep_rows_python_float_quirk():
    assert E.keep_rows(512, 6.0) == 85 and E.keep_rows(32, 6.0) == 5 and E.keep_rows(32, 3.0) == 10
    assert E.keep_rows(512, 1.0) == 512


@pytest.mark.parametrize("tag", IMDCT_TAGS)
def test_imdct_tail_matches_reference(tag):
    n_fft, hop, H, W, explicit, up, nmin, nmax = G[f"imdct_{tag}_meta"]
    n_fft, hop, W = int(n_fft), int(hop), int(W)
    pseudo = G[f"imdct_{tag}_pseudo"] if f"imdct_{tag}_pseudo" in G else None
    frames = E.decode_signed(G[f"imdct_{tag}_spectro"], G[f"imdct_{tag}_pha"], nmin, nmax, 1e-7, up, bool(explicit), pseudo)
    audio = M.imdct4_forward(frames, n_fft, hop, n_fft, G4[f"kbdwin_{n_fft}"], True, (W - 1) * hop) / 2
    ref = G[f"imdct_{tag}_audio"]
    assert audio.shape == ref.shape
    assert np.abs(audio - ref).max() <= 2e-5 * np.abs(ref).max()      # reference runs the dB maths in fp32


@pytest.mark.parametrize("tag", MET_TAGS)
def test_metrics_match_reference(tag):
    N, hop, win, center = (int(v) for v in G[f"met_{tag}_meta"])
    mse, snr_sr, snr_lr, lsd, _ = E.compute_metrics(G[f"met_{tag}_hr"], G[f"met_{tag}_lr"], G[f"met_{tag}_sr"], N, hop, win,
                                                    G[f"met_{tag}_win2"], bool(center))
    ref = G[f"met_{tag}_out"]
    np.testing.assert_allclose([mse, snr_sr, snr_lr, lsd], ref, rtol=2e-5)


def test_power_spectrogram_is_torch_stft():
    g = torch.Generator().manual_seed(3)
    x = torch.randn(2, 700, generator=g, dtype=torch.float64)
    for n_fft, hop, win, center in ((128, 64, 128, True), (128, 32, 96, True), (64, 64, 64, False)):
        w = torch.hann_window(win, dtype=torch.float64)
        S = torch.stft(x, n_fft, hop, win, w, center=center, pad_mode="reflect", onesided=True, return_complex=True).abs() ** 2
        P = E.power_spectrogram(x.numpy(), n_fft, hop, win, w.numpy(), center)
        np.testing.assert_allclose(P, S.numpy(), rtol=1e-9, atol=1e-12)
