"""Oracle: generation/eval tail -- util.imdct and compute_matrics in numpy fp64.  TEST INFRASTRUCTURE ONLY.

Restates (reference paths):
  imdct            util/util.py:104-131   |x|*(max-min)+min -> 10*10^(x/20) - min_value -> explicit: amplitude = ch0 + ch1,
                                          sign = pha on the first int(H/up_ratio) rows, sign(ch0 - ch1) above ->
                                          permute to [B, frames, bins] -> _imdct(.) / 2
  compute_matrics  util/util.py:133-184   per-row moment matching of sr to hr, MSE, SNR(sr), SNR(lr), LSD over a
                                          power spectrogram of n_fft = 2*opt.n_fft with a KBD window of 2*win_length
Caller: generate_audio.py:40-59.

Parity note: torchaudio is absent from this image, so `torchaudio.functional.spectrogram` is restated from
its documented definition = |torch.stft(center, pad_mode='reflect', onesided)|^power; tests/golden/evaltail.npz holds
the reference's own compute_matrics / imdct outputs obtained with torchaudio.functional stubbed by exactly that
(tools/gen_golden.py), so the dB one-liners and the STFT are pinned to torch, not to a torchaudio release.
"""
import numpy as np


def db_to_amplitude(x, ref=10.0, power=0.5):
    return ref * np.power(np.power(10.0, 0.1 * x), power)


def keep_rows(H, up_ratio):
    """util/util.py:117,124: int(size*(1/up_ratio)) evaluated in Python floats."""
    return int(H * (1 / up_ratio)) if up_ratio > 1 else H


def decode_signed(spectro, pha, nmin, nmax, min_value=1e-7, up_ratio=1, explicit_encoding=False, pseudo_sign=None):
    """util/util.py:104-126 up to (not including) the inverse transform.  spectro [B,C,H,W]; pha [B,1,H,W] or [B,H,W];
    returns signed amplitudes [B, W, H] (the `.permute(0,2,1)` of :128-130).  `pseudo_sign` replaces the torch.randint
    draw of the non-explicit branch (:122)."""
    s = np.abs(np.asarray(spectro, np.float64)) * (nmax - nmin) + nmin
    s = db_to_amplitude(s) - min_value
    pha = np.asarray(pha, np.float64)
    H = s.shape[-2]
    if explicit_encoding:
        pha = pha.reshape((-1,) + pha.shape[-2:])
        pseudo = np.sign(s[..., 0, :, :] - s[..., 1, :, :])
        amp = s[..., 0, :, :] + s[..., 1, :, :]
        if up_ratio > 1:
            k = keep_rows(H, up_ratio)
            pha = np.concatenate([pha[..., :k, :], pseudo[..., k:, :]], axis=-2)
    else:
        amp = s[:, 0]
        pha = pha.reshape((-1,) + pha.shape[-2:])
        if up_ratio > 1:
            k = keep_rows(H, up_ratio)
            pha = np.concatenate([pha[..., :k, :], np.asarray(pseudo_sign, np.float64).reshape(pha.shape)[..., k:, :]], axis=-2)
    return np.ascontiguousarray(np.transpose(amp * pha, (0, 2, 1)))


def reflect_index(i, n):
    """torch 'reflect' padding index (no edge repeat)."""
    i = np.abs(i)
    return np.where(i >= n, 2 * (n - 1) - i, i)


def power_spectrogram(x, n_fft, hop, win, window, center=True):
    """|STFT|^2, onesided, shape [B, n_fft/2+1, frames] (torch.stft conventions: window centred in the n_fft frame)."""
    x = np.atleast_2d(np.asarray(x, np.float64))
    T = x.shape[-1]
    w = np.zeros(n_fft)
    left = (n_fft - win) // 2
    w[left:left + win] = np.asarray(window, np.float64)
    pad = n_fft // 2 if center else 0
    frames = 1 + (T + 2 * pad - n_fft) // hop
    idx = np.arange(n_fft)[None, :] + hop * np.arange(frames)[:, None] - pad
    idx = reflect_index(idx, T)
    fr = x[:, idx] * w
    S = np.fft.rfft(fr, axis=-1)
    return np.transpose(S.real ** 2 + S.imag ** 2, (0, 2, 1))


def match_moments(sr, hr):
    """util/util.py:139-140 (torch.std is the unbiased estimator)."""
    sr = (sr - sr.mean(-1, keepdims=True)) / sr.std(-1, ddof=1, keepdims=True)
    return sr * hr.std(-1, ddof=1, keepdims=True) + hr.mean(-1, keepdims=True)


def compute_metrics(hr, lr, sr, n_fft, hop, win, window2, center=True):
    """util/util.py:133-184.  `window2` = kbdwin(2*win).  Returns (mse, snr_sr, snr_lr, lsd, sr_matched)."""
    hr = np.atleast_2d(np.asarray(hr, np.float64))
    lr = np.atleast_2d(np.asarray(lr, np.float64))
    sr = match_moments(np.atleast_2d(np.asarray(sr, np.float64)), hr)
    mse = ((sr - hr) ** 2).mean()
    snr_sr = (10 * np.log10((hr ** 2).sum(-1) / ((sr - hr) ** 2).sum(-1))).mean()
    snr_lr = (10 * np.log10((hr ** 2).sum(-1) / ((lr - hr) ** 2).sum(-1))).mean()
    ph = power_spectrogram(hr, 2 * n_fft, 2 * hop, 2 * win, window2, center)
    ps = power_spectrogram(sr, 2 * n_fft, 2 * hop, 2 * win, window2, center)
    d = np.log10(ph + 1e-6) - np.log10(ps + 1e-6)
    lsd = np.sqrt((d ** 2).mean(-2)).mean()
    return mse, snr_sr, snr_lr, lsd, sr
