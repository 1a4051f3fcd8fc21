"""HIP generation/eval tail (util.imdct, compute_matrics) through the C ABI against the reference's own outputs
(tests/golden/evaltail.npz) and the oracle on larger seeded inputs."""
import os
import types

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

G = np.load(os.path.join(os.path.dirname(__file__), "golden", "evaltail.npz"))
IMDCT_TAGS = ["ex_b2_u6", "ex_b1_u3", "ex_b2_u1", "pl_b2_u1", "pl_b2_u6"]
MET_TAGS = ["n64_b3", "n64_1d", "n1024_b2", "n64_nc"]


def _opt(N, hop, win, center):
    return types.SimpleNamespace(n_fft=N, hop_length=hop, win_length=win, center=bool(center), hr_sampling_rate=48000)


@pytest.mark.parametrize("tag", IMDCT_TAGS)
def test_imdct_tail_matches_reference(tag, monkeypatch):
    from pix2pixhdaudiosr_amd.models.mdct import IMDCT4
    from pix2pixhdaudiosr_amd.util import util as U
    n_fft, hop, H, W, explicit, up, nmin, nmax = G[f"imdct_{tag}_meta"]
    n_fft, hop, W, explicit = int(n_fft), int(hop), int(W), bool(explicit)
    dev = torch.device("cuda:0")
    spectro = torch.from_numpy(G[f"imdct_{tag}_spectro"]).to(dev)
    pha = torch.from_numpy(G[f"imdct_{tag}_pha"]).to(dev)
    if f"imdct_{tag}_pseudo" in G:                     # replay the reference's random-sign draw
        pseudo = torch.from_numpy(G[f"imdct_{tag}_pseudo"]).to(dev)
        monkeypatch.setattr(torch, "randint", lambda low, high, size, device=None: ((pseudo + 1) / 2).to(torch.int64).reshape(size))
    _imdct = IMDCT4(n_fft=n_fft, hop_length=hop, win_length=n_fft, window=U.kbdwin, out_length=(W - 1) * hop, device=dev)
    norm = {"min": torch.tensor(nmin, dtype=torch.float32), "max": torch.tensor(nmax, dtype=torch.float32)}
    audio = U.imdct(spectro, pha if explicit else pha.unsqueeze(1), norm, _imdct, up_ratio=up, explicit_encoding=explicit)
    ref = G[f"imdct_{tag}_audio"]
    assert tuple(audio.shape) == ref.shape
    err = np.abs(audio.cpu().numpy() - ref).max()
    assert err <= 1e-4 * np.abs(ref).max(), err        # fp32 dB chain (exp10 of values up to -35 dB), tolerance 1e-4 rel


@pytest.mark.parametrize("tag", MET_TAGS)
def test_metrics_match_reference(tag):
    from pix2pixhdaudiosr_amd.util import util as U
    N, hop, win, center = (int(v) for v in G[f"met_{tag}_meta"])
    dev = torch.device("cuda:0")
    hr, lr, sr = (torch.from_numpy(G[f"met_{tag}_{k}"]).to(dev) for k in ("hr", "lr", "sr"))
    out = U.compute_matrics(hr, lr, sr, _opt(N, hop, win, center))
    assert len(out) == 7 and out[3:6] == (0, 0, 0)
    got = np.array([out[0], out[1], out[2], out[6]])
    np.testing.assert_allclose(got, G[f"met_{tag}_out"], rtol=1e-4)      # floating point: 1e-4 relative


def test_metrics_full_size_against_oracle():
    """One 48 kHz segment batch at the shipped geometry (n_fft 1024 -> 2048-point STFT) and the n_fft 2048 variant
    (4096-point STFT, the largest supported)."""
    from oracle import evaltail as E
    from pix2pixhdaudiosr_amd.util import util as U
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(77)
    for N, B, T in ((1024, 4, 130560), (2048, 2, 65536)):
        hr = 0.1 * torch.randn(B, T, generator=g)
        lr = hr + 0.05 * torch.randn(B, T, generator=g)
        sr = 1.3 * hr + 0.02 * torch.randn(B, T, generator=g) - 0.02
        res, matched = U.audio_metrics(hr.to(dev), lr.to(dev), sr.to(dev), N, N // 2, N, True)
        mse, snr_sr, snr_lr, lsd, sr_m = E.compute_metrics(hr.numpy(), lr.numpy(), sr.numpy(), N, N // 2, N,
                                                           U.kbdwin(2 * N).numpy(), True)
        np.testing.assert_allclose(res.cpu().numpy(), [mse, snr_sr, snr_lr, lsd], rtol=1e-4)
        np.testing.assert_allclose(matched.cpu().numpy(), sr_m, rtol=0, atol=2e-6)


def test_metrics_identical_signals_and_errors():
    from pix2pixhdaudiosr_amd import _lib
    from pix2pixhdaudiosr_amd.util import util as U
    dev = torch.device("cuda:0")
    x = 0.1 * torch.randn(2, 4096, device=dev)
    res, matched = U.audio_metrics(x, x + 0.01, x, 64, 32, 64, True)
    r = res.cpu().numpy()
    assert r[0] < 1e-12 and r[3] < 1e-3 and r[1] > 60          # sr == hr: zero error, zero LSD, SNR limited by fp32 rounding
    with pytest.raises(_lib.P2PHDError):
        U.audio_metrics(x[:, :40], x[:, :40], x[:, :40], 64, 32, 64, True)      # shorter than the reflect padding
    with pytest.raises(_lib.P2PHDError):
        U.audio_metrics(x, x, x, 4096, 2048, 4096, True)                        # 8192-point STFT unsupported
