"""MDCT4 / IMDCT4 with the reference's constructor and call signatures (models/mdct.py:461-566),
running on the HIP kernels of libp2phd_hip.so (csrc/mdct.hip) through the C ABI.

Differences from the reference, all deliberate and documented in DESIGN.md:
  * the result is float32 (the reference's complex128 twiddles make it float64; pass
    ``out_dtype=torch.float64`` to get the reference dtype, the values are the fp32 ones);
  * ``n_fft`` must be a power of two in [16, 4096] and ``pad_mode`` must be 'constant';
  * there is no CPU path: tensors are moved to ``device`` (as the reference does) and the call
    raises if the HIP library is missing.
The ``len(signal)`` quirk of MDCT4.forward (mdct.py:488: the padding is derived from the size of
dim 0, i.e. the batch size for a [B, T] input) is reproduced bit-exactly by
``p2phd_mdct4_frame_layout``.
"""
import ctypes as C

import torch
from torch import nn

from .. import _lib


def _make_window(window, win_length, device):
    if window is None:
        window = torch.ones
    if callable(window):
        win_length = int(win_length)
        w = window(win_length)
    else:
        w = window
        win_length = len(window)
    return w.detach().to(device=device, dtype=torch.float32).contiguous(), win_length


class _Tables:
    _cache = {}

    @classmethod
    def get(cls, n_fft, device):
        key = (n_fft, str(device))
        t = cls._cache.get(key)
        if t is None:
            L = _lib.lib()
            n = L.p2phd_mdct4_tables_floats(n_fft)
            host = torch.empty(n, dtype=torch.float32)
            _lib.check(L.p2phd_mdct4_tables_fill(n_fft, C.c_void_p(host.data_ptr())), "mdct4_tables_fill")
            t = host.to(device)
            cls._cache[key] = t
        return t


def frame_layout(dim0, T, hop, win, center):
    sp, ep, nf = C.c_int64(), C.c_int64(), C.c_int64()
    _lib.check(_lib.lib().p2phd_mdct4_frame_layout(int(dim0), int(T), int(hop), int(win), int(bool(center)),
                                                   C.byref(sp), C.byref(ep), C.byref(nf)), "frame_layout")
    return sp.value, ep.value, nf.value


def _run_mdct(x2d, n_fft, hop, win, window, tables, start_pad, n_frames, scale):
    B, T = x2d.shape
    out = torch.empty((B, n_frames, n_fft // 2), dtype=torch.float32, device=x2d.device)
    _lib.check(_lib.lib().p2phd_mdct4_fwd(_lib.ptr(x2d), B, T, n_fft, hop, win, _lib.ptr(window), _lib.ptr(tables),
                                          start_pad, n_frames, scale, _lib.ptr(out), _lib.stream_ptr()), "mdct4_fwd")
    return out


def _run_imdct(spec, n_fft, hop, win, window, tables, crop, out_len, scale):
    B, F, _ = spec.shape
    out = torch.empty((B, out_len), dtype=torch.float32, device=spec.device)
    _lib.check(_lib.lib().p2phd_imdct4_fwd(_lib.ptr(spec), B, F, n_fft, hop, win, _lib.ptr(window), _lib.ptr(tables),
                                           crop, out_len, scale, _lib.ptr(out), _lib.stream_ptr()), "imdct4_fwd")
    return out


class _MDCT4Fn(torch.autograd.Function):
    """x[B,T] -> S[B,F,N/2]; backward is the adjoint = un-normalised IMDCT core + overlap-add."""

    @staticmethod
    def forward(ctx, x2d, n_fft, hop, win, window, tables, start_pad, n_frames, scale):
        ctx.cfg = (n_fft, hop, win, start_pad, x2d.shape[1], scale)
        ctx.save_for_backward(window, tables)
        return _run_mdct(x2d, n_fft, hop, win, window, tables, start_pad, n_frames, scale)

    @staticmethod
    def backward(ctx, g):
        n_fft, hop, win, start_pad, T, scale = ctx.cfg
        window, tables = ctx.saved_tensors
        gx = _run_imdct(g.contiguous().float(), n_fft, hop, win, window, tables, start_pad, T, scale)
        return gx, None, None, None, None, None, None, None, None


class _IMDCT4Fn(torch.autograd.Function):
    """S[B,F,N/2] -> y[B,out_len]; backward is the adjoint = MDCT core of the re-padded cotangent."""

    @staticmethod
    def forward(ctx, spec, n_fft, hop, win, window, tables, crop, out_len, scale):
        ctx.cfg = (n_fft, hop, win, crop, spec.shape[1], scale)
        ctx.save_for_backward(window, tables)
        return _run_imdct(spec, n_fft, hop, win, window, tables, crop, out_len, scale)

    @staticmethod
    def backward(ctx, g):
        n_fft, hop, win, crop, F, scale = ctx.cfg
        window, tables = ctx.saved_tensors
        gs = _run_mdct(g.contiguous().float(), n_fft, hop, win, window, tables, crop, F, scale)
        return gs, None, None, None, None, None, None, None, None


class _Base(nn.Module):
    def _setup(self, n_fft, hop_length, win_length, window, center, pad_mode, device, out_dtype):
        self.n_fft = n_fft
        self.pad_mode = pad_mode
        self.device = device
        self.hop_length = hop_length
        self.center = center
        self.out_dtype = out_dtype
        self.window, self.win_length = _make_window(window, win_length, device)
        assert self.win_length <= self.n_fft, 'Window lenth %d should be no more than fft length %d' % (self.win_length, self.n_fft)
        assert self.hop_length <= self.win_length, 'You hopped more than one frame'
        if n_fft < 16 or n_fft > 4096 or (n_fft & (n_fft - 1)):
            raise NotImplementedError("n_fft must be a power of two in [16, 4096] for the HIP MDCT, got %d" % n_fft)
        if pad_mode != 'constant':
            raise NotImplementedError("only pad_mode='constant' is implemented (the reference default)")

    def _tables(self):
        return _Tables.get(self.n_fft, self.window.device)


class MDCT4(_Base):
    def __init__(self, n_fft=2048, hop_length=None, win_length=None, window=None, center=True, pad_mode='constant',
                 device='cuda', out_dtype=None) -> None:
        super().__init__()
        self._setup(n_fft, hop_length, win_length, window, center, pad_mode, device, out_dtype)

    def forward(self, signal, _dim0=None):
        """`_dim0` (internal): the value the reference's len(signal) quirk would see -- lets a caller stack several [B, T]
        batches along dim 0 into ONE launch and still get the frame layout of a [B, T] call (Pix2PixHDModel.encode_input)."""
        signal = signal.to(self.device)
        start_pad, _, n_frames = frame_layout(len(signal) if _dim0 is None else int(_dim0), signal.shape[-1], self.hop_length,
                                              self.win_length, self.center)
        lead = signal.shape[:-1]
        x2d = signal.reshape(-1, signal.shape[-1]).to(torch.float32).contiguous()
        _lib.require_gpu_tensor(x2d, "MDCT4 input")
        S = _MDCT4Fn.apply(x2d, self.n_fft, self.hop_length, self.win_length, self.window, self._tables(),
                           start_pad, n_frames, 1.0)
        S = S.reshape(*lead, n_frames, self.n_fft // 2)
        return S if self.out_dtype is None else S.to(self.out_dtype)


class IMDCT4(_Base):
    def __init__(self, n_fft=2048, hop_length=None, win_length=None, window=None, center=True, pad_mode='constant',
                 out_length=None, device='cuda', out_dtype=None) -> None:
        super().__init__()
        self._setup(n_fft, hop_length, win_length, window, center, pad_mode, device, out_dtype)
        self.out_length = out_length

    def forward(self, signal):
        assert signal.dim() == 3, 'Only tensors shaped in BHW are supported, got tensor of shape %s' % (str(signal.size()))
        assert signal.size()[-1] == self.n_fft // 2, 'The last dim of input tensor should match the n_fft. Expected %d ,got %d' % (self.n_fft, signal.size()[-1])
        spec = signal.to(self.device).to(torch.float32).contiguous()
        _lib.require_gpu_tensor(spec, "IMDCT4 input")
        F = spec.shape[1]
        full = (F - 1) * self.hop_length + self.win_length                   # mdct.py:559
        if self.center:
            half = self.win_length // 2
            crop, out_len = half, max(full - half - (self.win_length - half), 0)   # [win//2 : -win//2], mdct.py:564
        else:
            crop, out_len = 0, full
        if self.out_length is not None:
            out_len = min(out_len, int(self.out_length))                    # mdct.py:566
        y = _IMDCT4Fn.apply(spec, self.n_fft, self.hop_length, self.win_length, self.window, self._tables(),
                            crop, out_len, 4.0 / self.n_fft)
        y = y.reshape(spec.shape[0], 1, 1, out_len)                          # fold's [B, C=1, 1, L] output shape
        return y if self.out_dtype is None else y.to(self.out_dtype)


# ------------------------------------------------------------------------------------------
# MDCT2 / IMDCT2 (models/mdct.py:352-454 of the reference) on csrc/dct.hip
# ------------------------------------------------------------------------------------------
class _DctTables:
    _cache = {}

    @classmethod
    def get(cls, n_fft, device):
        key = (n_fft, str(device))
        t = cls._cache.get(key)
        if t is None:
            L = _lib.lib()
            host = torch.empty(L.p2phd_dct_tables_floats(n_fft), dtype=torch.float32)
            _lib.check(L.p2phd_dct_tables_fill(n_fft, C.c_void_p(host.data_ptr())), "dct_tables_fill")
            t = host.to(device)
            cls._cache[key] = t
        return t


def _run_mdct2(x2d, n_fft, hop, win, window, tables, start_pad, n_frames, scale, k0):
    B, T = x2d.shape
    out = torch.empty((B, n_frames, n_fft), dtype=torch.float32, device=x2d.device)
    _lib.check(_lib.lib().p2phd_mdct2_fwd(_lib.ptr(x2d), B, T, n_fft, hop, win, _lib.ptr(window), _lib.ptr(tables), start_pad,
                                          n_frames, scale, k0, _lib.ptr(out), _lib.stream_ptr()), "mdct2_fwd")
    return out


def _run_imdct2(spec, n_fft, hop, win, window, tables, crop, out_len, scale, k0):
    B, F, _ = spec.shape
    out = torch.empty((B, out_len), dtype=torch.float32, device=spec.device)
    _lib.check(_lib.lib().p2phd_imdct2_fwd(_lib.ptr(spec), B, F, n_fft, hop, win, _lib.ptr(window), _lib.ptr(tables), crop,
                                           out_len, scale, k0, _lib.ptr(out), _lib.stream_ptr()), "imdct2_fwd")
    return out


class _MDCT2Fn(torch.autograd.Function):
    """Framed, windowed DCT-II (DCT_2N_native scaling).  Adjoint = inverse kernel with scale s/N, k0 -> 2 k0."""

    @staticmethod
    def forward(ctx, x2d, n_fft, hop, win, window, tables, start_pad, n_frames, scale, k0):
        ctx.cfg = (n_fft, hop, win, start_pad, x2d.shape[1], scale, k0)
        ctx.save_for_backward(window, tables)
        return _run_mdct2(x2d, n_fft, hop, win, window, tables, start_pad, n_frames, scale, k0)

    @staticmethod
    def backward(ctx, g):
        n_fft, hop, win, start_pad, T, scale, k0 = ctx.cfg
        window, tables = ctx.saved_tensors
        gx = _run_imdct2(g.contiguous().float(), n_fft, hop, win, window, tables, start_pad, T, scale / n_fft, 2.0 * k0)
        return (gx,) + (None,) * 9


class _IMDCT2Fn(torch.autograd.Function):
    """Framed DCT-III + window + overlap-add.  Adjoint = forward kernel with scale s*N, k0 -> k0/2."""

    @staticmethod
    def forward(ctx, spec, n_fft, hop, win, window, tables, crop, out_len, scale, k0):
        ctx.cfg = (n_fft, hop, win, crop, spec.shape[1], scale, k0)
        ctx.save_for_backward(window, tables)
        return _run_imdct2(spec, n_fft, hop, win, window, tables, crop, out_len, scale, k0)

    @staticmethod
    def backward(ctx, g):
        n_fft, hop, win, crop, F, scale, k0 = ctx.cfg
        window, tables = ctx.saved_tensors
        gs = _run_mdct2(g.contiguous().float(), n_fft, hop, win, window, tables, crop, F, scale * n_fft, 0.5 * k0)
        return (gs,) + (None,) * 9


def _check_dct_op(op, kind):
    from ..dct.dct_native import DCT_2N_native, IDCT_2N_native
    want = DCT_2N_native if kind == 'dct' else IDCT_2N_native
    if op is not None and not isinstance(op, want):
        raise NotImplementedError("the HIP MDCT2/IMDCT2 fuse the %s operator: pass %s() (or None), got %r"
                                  % (kind, want.__name__, type(op).__name__))


class _Base2(_Base):
    def _setup2(self, n_fft, hop_length, win_length, window, center, pad_mode, device, out_dtype):
        if n_fft > 2048:
            raise NotImplementedError("MDCT2/IMDCT2 on the HIP path support n_fft <= 2048, got %d" % n_fft)
        self._setup(n_fft, hop_length, win_length, window, center, pad_mode, device, out_dtype)

    def _tables(self):
        return _DctTables.get(self.n_fft, self.window.device)


class MDCT2(_Base2):
    def __init__(self, n_fft=2048, hop_length=None, win_length=None, window=None, center=True, pad_mode='constant',
                 device='cuda', dct_op=None, out_dtype=None) -> None:
        super().__init__()
        _check_dct_op(dct_op, 'dct')
        self._setup2(n_fft, hop_length, win_length, window, center, pad_mode, device, out_dtype)

    def forward(self, signal, return_ola=False, _dim0=None):
        if return_ola:
            raise NotImplementedError("return_ola (time-domain discriminator frames) is outside the hot path")
        signal = signal.to(self.device)
        start_pad, _, n_frames = frame_layout(len(signal) if _dim0 is None else int(_dim0), signal.shape[-1], self.hop_length,
                                              self.win_length, self.center)
        lead = signal.shape[:-1]
        x2d = signal.reshape(-1, signal.shape[-1]).to(torch.float32).contiguous()
        _lib.require_gpu_tensor(x2d, "MDCT2 input")
        S = _MDCT2Fn.apply(x2d, self.n_fft, self.hop_length, self.win_length, self.window, self._tables(),
                           start_pad, n_frames, 1.0, 1.0)
        S = S.reshape(*lead, n_frames, self.n_fft)
        return S if self.out_dtype is None else S.to(self.out_dtype)


class IMDCT2(_Base2):
    def __init__(self, n_fft=2048, hop_length=None, win_length=None, window=None, center=True, pad_mode='constant',
                 out_length=None, device='cuda', idct_op=None, out_dtype=None) -> None:
        super().__init__()
        _check_dct_op(idct_op, 'idct')
        self._setup2(n_fft, hop_length, win_length, window, center, pad_mode, device, out_dtype)
        self.out_length = out_length

    def forward(self, signal):
        assert signal.dim() == 3, 'Only tensors shaped in BHW are supported, got tensor of shape %s' % (str(signal.size()))
        assert signal.size()[-1] == self.n_fft, 'The last dim of input tensor should match the n_fft. Expected %d ,got %d' % (self.n_fft, signal.size()[-1])
        spec = signal.to(self.device).to(torch.float32).contiguous()
        _lib.require_gpu_tensor(spec, "IMDCT2 input")
        F = spec.shape[1]
        full = (F - 1) * self.hop_length + self.win_length                   # mdct.py:447
        if self.center:
            half = self.win_length // 2
            crop, out_len = half, max(full - half - (self.win_length - half), 0)   # mdct.py:452
        else:
            crop, out_len = 0, full
        if self.out_length is not None:
            out_len = min(out_len, int(self.out_length))
        y = _IMDCT2Fn.apply(spec, self.n_fft, self.hop_length, self.win_length, self.window, self._tables(),
                            crop, out_len, 0.5, 1.0)                          # idct(.)/2, mdct.py:437
        y = y.reshape(spec.shape[0], 1, 1, out_len)
        return y if self.out_dtype is None else y.to(self.out_dtype)
