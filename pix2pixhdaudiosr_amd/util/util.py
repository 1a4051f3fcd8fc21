"""Host-side helpers of the hot path with the reference's names (util/util.py)."""
import os

import torch


def mkdir(path):
    """util/util.py:50-52 of the reference (options/base_options.py:100 and the visualizer call these)."""
    os.makedirs(path, exist_ok=True)


def mkdirs(paths):
    """util/util.py:43-48: one path or a list of them."""
    for p in ([paths] if isinstance(paths, str) else list(paths)):
        mkdir(p)


def kbdwin(N: int, beta: float = 12.0, device='cpu') -> torch.Tensor:
    """MATLAB-style Kaiser-Bessel-derived window, same arithmetic as util/util.py:186-193
    of the reference (a one-off host constant: the kernels only read it)."""
    assert N % 2 == 0, "N must be even"
    w = torch.kaiser_window(window_length=N // 2 + 1, beta=beta * torch.pi, periodic=False, device=device)
    w_sum = w.sum()
    wdw_half = torch.sqrt(torch.cumsum(w, dim=0) / w_sum)[:-1]
    return torch.cat((wdw_half, wdw_half.flip(dims=(0,))), dim=0)


def imdct(spectro, pha, norm_param, _imdct, min_value=1e-7, up_ratio=1, explicit_encoding=False):
    """Generation tail of the reference (util/util.py:104-131, caller generate_audio.py:40-42): de-normalise, dB ->
    amplitude, restore the sign (LR sign on the low band, sign(ch0 - ch1) -- or a random sign without explicit
    encoding -- above it), and run the inverse transform `_imdct` on [B, frames, bins]; returns `_imdct(.) / 2`.
    One HIP launch (p2phd_spectro_decode_signed) replaces the elementwise chain and the permute."""
    from .. import _lib
    dev = spectro.device
    _lib.require_gpu_tensor(spectro, "spectro")
    x = spectro.float()
    if x.dim() == 3:
        x = x.unsqueeze(1)
    x = x.contiguous()
    B, Cc, M, Fr = x.shape
    if explicit_encoding and Cc != 2:
        raise ValueError(f"imdct: explicit encoding needs 2 channels, got {Cc}")
    if not explicit_encoding and Cc != 1:
        raise ValueError(f"imdct: plain encoding needs 1 channel, got {Cc}")
    p = pha.to(dev).float().reshape(-1, M, Fr)
    if p.shape[0] != B:
        raise ValueError(f"imdct: pha batch {p.shape[0]} != spectro batch {B}")
    keep = int(M * (1 / up_ratio)) if up_ratio > 1 else M
    if not explicit_encoding and keep < M:
        pseudo = (2 * torch.randint(low=0, high=2, size=(B, M, Fr), device=dev) - 1).float()
        p = torch.cat((p[:, :keep], pseudo[:, keep:]), dim=1)
    p = p.contiguous()
    mm = torch.stack([torch.as_tensor(norm_param['min']).float().reshape(()),
                      torch.as_tensor(norm_param['max']).float().reshape(())]).to(dev).contiguous()
    spec = torch.empty((B, Fr, M), dtype=torch.float32, device=dev)
    _lib.check(_lib.lib().p2phd_spectro_decode_signed(_lib.ptr(x), _lib.ptr(p), _lib.ptr(mm), B, Fr, M, Cc, keep,
                                                      float(min_value), 1.0, _lib.ptr(spec), _lib.stream_ptr()),
               "spectro_decode_signed")
    return _imdct(spec) / 2


_STFT_TABLES = {}


def _stft_tables(n2, device):
    from .. import _lib
    key = (n2, str(device))
    if key not in _STFT_TABLES:
        host = torch.empty(_lib.lib().p2phd_stft_tables_floats(n2), dtype=torch.float32)
        _lib.check(_lib.lib().p2phd_stft_tables_fill(n2, _lib.ptr(host)), "stft_tables_fill")
        _STFT_TABLES[key] = host.to(device)
    return _STFT_TABLES[key]


def audio_metrics(hr_audio, lr_audio, sr_audio, n_fft, hop_length, win_length, center=True):
    """Device-side compute_matrics: returns (result4 tensor [mse, snr_sr, snr_lr, lsd] on the GPU, sr moment-matched to
    hr).  No host synchronisation; `compute_matrics` below is the reference-shaped wrapper."""
    from .. import _lib
    _lib.require_gpu_tensor(sr_audio, "sr_audio")
    dev = sr_audio.device
    T = sr_audio.shape[-1]
    sr = sr_audio.float().reshape(-1, T).contiguous()
    hr = hr_audio.to(dev).float().reshape(-1, T).contiguous()
    lr = lr_audio.to(dev).float().reshape(-1, T).contiguous()
    if hr.shape != sr.shape or lr.shape != sr.shape:
        raise ValueError(f"compute_matrics: shapes differ: hr {tuple(hr.shape)} lr {tuple(lr.shape)} sr {tuple(sr.shape)}")
    B = sr.shape[0]
    n2, hop2, win2 = 2 * int(n_fft), 2 * int(hop_length), 2 * int(win_length)
    window2 = kbdwin(win2).to(dev).contiguous()
    L = _lib.lib()
    nbytes = L.p2phd_metrics_workspace_bytes(B, T, n2, hop2, win2, int(bool(center)))
    if nbytes == 0:
        raise _lib.P2PHDError("compute_matrics: " + L.p2phd_last_error().decode("utf-8", "replace"))
    ws = torch.empty(nbytes // 8 + 1, dtype=torch.float64, device=dev)
    matched = torch.empty_like(sr)
    result = torch.empty(4, dtype=torch.float32, device=dev)
    _lib.check(L.p2phd_audio_metrics(_lib.ptr(hr), _lib.ptr(lr), _lib.ptr(sr), B, T, n2, hop2, win2, _lib.ptr(window2),
                                     _lib.ptr(_stft_tables(n2, dev)), int(bool(center)), _lib.ptr(matched), _lib.ptr(result),
                                     _lib.ptr(ws), _lib.stream_ptr()), "audio_metrics")
    return result, matched.reshape(sr_audio.shape)


def compute_matrics(hr_audio, lr_audio, sr_audio, opt):
    """MSE / SNR / LSD of the reference (util/util.py:133-184); same 7-tuple `(mse, snr_sr, snr_lr, 0, 0, 0, lsd)` of
    Python floats (the segmental-SNR and PESQ slots are constant zeros there too)."""
    result, _ = audio_metrics(hr_audio, lr_audio, sr_audio, opt.n_fft, opt.hop_length, opt.win_length, opt.center)
    mse, snr_sr, snr_lr, lsd = result.tolist()
    return mse, snr_sr, snr_lr, 0, 0, 0, lsd
