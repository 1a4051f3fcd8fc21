"""Oracle: DCT_2N_native / IDCT_2N_native and MDCT2 / IMDCT2 in numpy fp64.  TEST INFRASTRUCTURE ONLY.

Restates (reference paths):
  DCT_2N_native.forward    dct/dct_native.py:16-34   zero-pad to 2N, rfft, first N bins, * 1/N, (re,im)*(2cos,2sin)(pi k/2N), re+im
  IDCT_2N_native.forward   dct/dct_native.py:45-68   x*(2cos,2sin), zero-extend to 2N bins, irfft(n=2N)[:N] * N
  MDCT2.forward            models/mdct.py:377-403    pad (len(signal) quirk) -> unfold -> window -> zero-pad -> dct
  IMDCT2.forward           models/mdct.py:432-454    idct/2 -> window -> fold -> centre crop -> out_length
KAT: dct([1..16]) = [17, -6.4741, 0, -0.70977, ...], idct(dct(a)) = 2a (test/DCT_test.ipynb cell 34).
"""
import numpy as np

from .mdct4 import frame_layout, _unfold


def dct_2n(x):
    x = np.asarray(x, dtype=np.float64)
    N = x.shape[-1]
    y = np.fft.rfft(np.concatenate([x, np.zeros_like(x)], axis=-1), axis=-1)[..., :N] / N
    k = np.arange(N)
    return y.real * (2 * np.cos(np.pi * k / (2 * N))) + y.imag * (2 * np.sin(np.pi * k / (2 * N)))


def idct_2n(X):
    X = np.asarray(X, dtype=np.float64)
    N = X.shape[-1]
    k = np.arange(N)
    z = X * (2 * np.cos(np.pi * k / (2 * N))) + 1j * X * (2 * np.sin(np.pi * k / (2 * N)))
    z = np.concatenate([z, np.zeros_like(z)], axis=-1)            # second half of the 2N-bin buffer is ignored by irfft
    return np.fft.irfft(z[..., : N + 1], n=2 * N, axis=-1)[..., :N] * N


def mdct2_forward(signal, n_fft, hop, win, window, center=True):
    signal = np.asarray(signal)
    start_pad, end_pad, _ = frame_layout(signal.shape, hop, win, center)
    x = np.pad(signal, [(0, 0)] * (signal.ndim - 1) + [(start_pad, end_pad)])
    x = _unfold(x, win, hop) * np.asarray(window)
    if n_fft > win:
        x = np.pad(x, [(0, 0)] * (x.ndim - 1) + [(0, n_fft - win)])
    return dct_2n(x)


def imdct2_forward(spec, n_fft, hop, win, window, center=True, out_length=None):
    spec = np.asarray(spec)
    assert spec.ndim == 3 and spec.shape[-1] == n_fft
    s = idct_2n(spec) / 2.0
    if n_fft > win:
        s = s[..., :win]
    s = s * np.asarray(window)
    B, F, _ = s.shape
    out_len = (F - 1) * hop + win
    out = np.zeros((B, out_len))
    for t in range(F):
        out[:, t * hop: t * hop + win] += s[:, t]
    if center:
        out = out[..., win // 2: out_len - win // 2]
    if out_length is not None:
        out = out[..., :out_length]
    return out[:, None, None, :]
