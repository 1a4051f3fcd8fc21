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


def test_generation_flow_end_to_end(tmp_path):
    """generate_audio.py:13-49 composed from the built rows: wav -> AudioTestDataset (GPU LR round trip, segments) ->
    model.inference -> util.imdct -> concatenate -> compute_matrics; plus the identity property of the tail: the HR
    spectrogram itself, decoded with its true signs and up_ratio 1, reproduces the HR audio."""
    from math import sqrt
    from test_gpu_model import make_opt
    from pix2pixhdaudiosr_amd.data import wavio
    from pix2pixhdaudiosr_amd.data.data_loader import CreateDataLoader
    from pix2pixhdaudiosr_amd.models.mdct import IMDCT4
    from pix2pixhdaudiosr_amd.models.models import create_model
    from pix2pixhdaudiosr_amd.util import util as U
    F = np.load(os.path.join(os.path.dirname(__file__), "golden", "feeder.npz"))
    pcm = torch.from_numpy(F["test_wav_excerpt_i16"].astype(np.float32) / 32768.0)
    clip = str(tmp_path / "clip.wav")
    wavio.save(clip, pcm, 48000)
    seg = 127 * 32                                                     # (frames - 1) * hop
    opt = make_opt(isTrain=False, phase='test', dataroot=clip, segment_length=seg, batchSize=2, nThreads=0, is_lr_input=False,
                   max_dataset_size=float("inf"), serial_batches=True, checkpoints_dir=str(tmp_path))
    torch.manual_seed(1234)
    create_model(make_opt(checkpoints_dir=str(tmp_path))).save('latest')      # the checkpoint generation loads (which_epoch latest)
    loader = CreateDataLoader(opt)
    model = create_model(opt)
    model.eval()
    _imdct = IMDCT4(window=U.kbdwin, win_length=opt.win_length, hop_length=opt.hop_length, n_fft=opt.n_fft, center=opt.center,
                    out_length=opt.segment_length, device='cuda')
    up_ratio = opt.hr_sampling_rate / opt.lr_sampling_rate
    audio = []
    with torch.no_grad():
        for data in loader.load_data():
            sr_spectro, lr_pha, norm_param, lr_spectro = model.module.inference(data['label'], None)
            assert tuple(sr_spectro.shape) == (data['label'].shape[0], 2, 32, 128)
            audio.append(U.imdct(spectro=sr_spectro.abs(), pha=lr_pha.squeeze(1), norm_param=norm_param, _imdct=_imdct,
                                 up_ratio=up_ratio, explicit_encoding=True))
    audio = sqrt(up_ratio - 1) * torch.cat(audio, dim=0).view(1, -1)
    ds = loader.dataset
    n = ds.raw_audio.size(-1)
    assert audio.shape[-1] >= n and torch.isfinite(audio).all()
    out = U.compute_matrics(ds.raw_audio, ds.lr_audio[..., :n], audio[..., :n], opt)
    assert len(out) == 7 and all(np.isfinite(v) for v in out) and out[2] > 5.0      # LR round trip keeps the low band: SNR_LR > 5 dB
    # identity of the decode tail on a real clip
    hr = ds.raw_audio[:, :seg]
    spectro, pha, norm = model.to_spectro(hr, mask=False)
    rec = U.imdct(spectro, pha.squeeze(1), norm, _imdct, up_ratio=1, explicit_encoding=True)
    # explicit encoding stores alpha*pos+(1-alpha)*neg and (1-alpha)*pos+alpha*neg: their sum is |s| -> _imdct(|s|*sign)/2 = x/2 ... x
    # (util.imdct halves because the reference's IMDCT2 returns 2x; with IMDCT4 the factor shows up here)
    err = (2 * rec.reshape(1, -1) - hr).abs().max().item()
    assert err <= 2e-4 * hr.abs().max().item() + 1e-5, err
