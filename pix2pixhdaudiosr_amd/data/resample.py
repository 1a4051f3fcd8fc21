"""GPU sample-rate conversion for the feeder: `resample(waveform, orig_freq, new_freq)` with the call shape of
`torchaudio.functional.resample` (data/audio_dataset.py:55-57,109-113) on p2phd_resample_fwd.  No CPU path."""
import torch

from .. import _lib

_KERNELS = {}


def _kernel(orig_freq, new_freq, lpw, rolloff, device):
    key = (orig_freq, new_freq, lpw, rolloff, str(device))
    if key not in _KERNELS:
        L = _lib.lib()
        n = L.p2phd_resample_kernel_floats(orig_freq, new_freq, lpw, rolloff)
        if n == 0:
            raise _lib.P2PHDError("resample: " + L.p2phd_last_error().decode("utf-8", "replace"))
        host = torch.empty(n, dtype=torch.float32)
        _lib.check(L.p2phd_resample_kernel_fill(orig_freq, new_freq, lpw, rolloff, _lib.ptr(host)), "resample_kernel_fill")
        _KERNELS[key] = host.to(device)
    return _KERNELS[key]


def resample(waveform, orig_freq, new_freq, lowpass_filter_width=6, rolloff=0.99):
    orig_freq, new_freq = int(orig_freq), int(new_freq)
    if orig_freq == new_freq:
        return waveform
    _lib.require_gpu_tensor(waveform, "waveform")
    x = waveform.float()
    T = x.shape[-1]
    B = 1
    for d in x.shape[:-1]:
        B *= d
    x2 = x.reshape(B, T).contiguous()
    L = _lib.lib()
    kern = _kernel(orig_freq, new_freq, int(lowpass_filter_width), float(rolloff), x.device)     # validates the rates
    T_out = L.p2phd_resample_out_len(T, orig_freq, new_freq)
    out = torch.empty((B, T_out), dtype=torch.float32, device=x.device)
    _lib.check(L.p2phd_resample_fwd(_lib.ptr(x2), B, T, orig_freq, new_freq, int(lowpass_filter_width), float(rolloff),
                                    _lib.ptr(kern), _lib.ptr(out), T_out, _lib.stream_ptr()), "resample_fwd")
    return out.reshape(x.shape[:-1] + (T_out,))
