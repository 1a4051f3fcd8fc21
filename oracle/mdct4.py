"""Oracle: KBD window, MDCT4 / IMDCT4 and their adjoints in numpy (fp64 / complex128).

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

Each function restates, stage by stage, the reference code it cites
(paths relative to the reference checkout):

  kbdwin            util/util.py:186-193
  frame_layout      models/mdct.py:486-500  (incl. the ``len(signal)`` quirk, :488)
  mdct4_forward     models/mdct.py:483-513
  imdct4_forward    models/mdct.py:539-566

The reference keeps its twiddles in complex128 (mdct.py:483-484,539-540), so its
results are float64 whatever the input dtype; the oracle does the same.
"""
import numpy as np


def kbdwin(N, beta=12.0, dtype=np.float32):
    """Kaiser-Bessel-derived window, MATLAB style (util/util.py:186-193).

    The reference evaluates ``torch.kaiser_window`` in the default dtype (fp32);
    ``dtype`` selects where the cumulative sum / sqrt are rounded (the notebooks
    use an fp64 variant, test/DCT_test.ipynb cell 30).
    """
    assert N % 2 == 0, "N must be even"
    w = np.kaiser(N // 2 + 1, beta * np.pi).astype(dtype)      # periodic=False
    w_sum = w.sum(dtype=dtype)
    half = np.sqrt(np.cumsum(w, dtype=dtype) / w_sum)[:-1]
    return np.concatenate([half, half[::-1]]).astype(dtype)


def frame_layout(shape, hop, win, center=True):
    """(start_pad, end_pad, n_frames) exactly as models/mdct.py:488-500 computes them.

    ``signal_len = len(signal)`` is the size of dim 0: the sample count for a
    1-D signal but the *batch size* for a [B, T] signal (mdct.py:488).
    """
    signal_len = int(shape[0])
    T = int(shape[-1])
    start_pad = hop if center else 0
    additional = signal_len % hop
    end_pad = start_pad
    if additional:
        end_pad = start_pad + hop - additional
    padded = T + start_pad + end_pad
    n_frames = (padded - win) // hop + 1 if padded >= win else 0
    return start_pad, end_pad, n_frames


def _unfold(x, win, hop):
    n = (x.shape[-1] - win) // hop + 1
    idx = np.arange(win)[None, :] + hop * np.arange(n)[:, None]
    return x[..., idx]


def mdct4_forward(signal, n_fft, hop, win, window, center=True, stages=None):
    """models/mdct.py:486-513.  Returns float64 [..., frames, n_fft//2]."""
    signal = np.asarray(signal)
    start_pad, end_pad, _ = frame_layout(signal.shape, hop, win, center)
    pad = [(0, 0)] * (signal.ndim - 1) + [(start_pad, end_pad)]
    x = np.pad(signal, pad)                                    # :497
    x = _unfold(x, win, hop)                                   # :500
    x = x * np.asarray(window)                                 # :503 (input dtype)
    if n_fft > win:
        x = np.pad(x, [(0, 0)] * (x.ndim - 1) + [(0, n_fft - win)])   # :507
    n = np.arange(n_fft, dtype=np.float64)
    exp1 = np.exp(-1j * np.pi / n_fft * n)                     # :483
    k2 = np.arange(1, n_fft, 2, dtype=np.float64)
    exp2 = np.exp(-1j * (np.pi / (2 * n_fft) + np.pi / 4) * k2)   # :484
    s1 = x * exp1                                              # :509
    s2 = np.fft.fft(s1, axis=-1)[..., : n_fft // 2]            # :510
    s3 = exp2 * s2                                             # :511
    if stages is not None:
        stages.update(windowed=x, S_exp1=s1, S_fft=s2, S_exp2=s3)
    return np.real(s3)                                         # :513


def imdct4_forward(spec, n_fft, hop, win, window, center=True, out_length=None):
    """models/mdct.py:542-566.  spec [B, frames, n_fft//2] -> float64 [B, 1, 1, T]."""
    spec = np.asarray(spec)
    assert spec.ndim == 3 and spec.shape[-1] == n_fft // 2
    k2 = np.arange(1, n_fft, 2, dtype=np.float64)
    exp1 = np.exp(-1j * (np.pi / (2 * n_fft) + np.pi / 4) * k2)   # :539
    n2 = np.arange(0, 2 * n_fft, 2, dtype=np.float64)
    exp2 = np.exp(-1j * np.pi / (2 * n_fft) * n2)              # :540
    s = exp1 * spec                                            # :547
    s = np.fft.fft(s, n=n_fft, axis=-1)                        # :548 (zero-extended)
    s = np.real(s * exp2)                                      # :549
    if n_fft > win:
        s = s[..., :win]                                       # :553
    s = s * np.asarray(window)                                 # :556
    B, F, _ = s.shape
    out_len = (F - 1) * hop + win                              # :559
    out = np.zeros((B, out_len), dtype=s.dtype)
    for t in range(F):                                         # fold == overlap-add, :560
        out[:, t * hop: t * hop + win] += s[:, t]
    out = 4.0 / n_fft * out
    if center:
        out = out[..., win // 2: out_len - win // 2]           # :564  (-win//2 == -(win//2) for even win)
    if out_length is not None:
        out = out[..., :out_length]                            # :566
    return out[:, None, None, :]


# ---- adjoints (what autograd gives the reference; README.md:107-110 shows MDCT4.backward) ----

def _cos_kernel(n_fft):
    n = np.arange(n_fft, dtype=np.float64)[:, None]
    k = np.arange(n_fft // 2, dtype=np.float64)[None, :]
    return np.cos(2 * np.pi / n_fft * (n + 0.5 + n_fft / 4) * (k + 0.5))    # [n, k]


def mdct4_direct(signal, n_fft, hop, win, window, center=True):
    """Direct O(N^2) cosine-sum definition of what mdct4_forward computes (small sizes only)."""
    signal = np.asarray(signal, dtype=np.float64)
    start_pad, end_pad, _ = frame_layout(signal.shape, hop, win, center)
    x = np.pad(signal, [(0, 0)] * (signal.ndim - 1) + [(start_pad, end_pad)])
    x = _unfold(x, win, hop) * np.asarray(window, dtype=np.float64)
    if n_fft > win:
        x = np.pad(x, [(0, 0)] * (x.ndim - 1) + [(0, n_fft - win)])
    return x @ _cos_kernel(n_fft)


def mdct4_backward(grad_spec, in_shape, n_fft, hop, win, window, center=True):
    """d<grad_spec, MDCT4(x)>/dx : crop(fold(window * (C g))) ; derived adjoint of mdct4_forward."""
    g = np.asarray(grad_spec, dtype=np.float64)
    start_pad, end_pad, F = frame_layout(in_shape, hop, win, center)
    T = int(in_shape[-1])
    fr = g @ _cos_kernel(n_fft).T                              # [..., F, n_fft]
    fr = fr[..., :win] * np.asarray(window, dtype=np.float64)
    lead = g.shape[:-2]
    out = np.zeros(lead + (T + start_pad + end_pad,), dtype=np.float64)
    for t in range(F):
        out[..., t * hop: t * hop + win] += fr[..., t, :]
    return out[..., start_pad: start_pad + T]


def imdct4_backward(grad_audio, n_frames, n_fft, hop, win, window, center=True):
    """d<grad_audio, IMDCT4(S)>/dS : (4/N) * C^T (window * unfold(pad(g))) ; derived adjoint."""
    g = np.asarray(grad_audio, dtype=np.float64)
    g = g.reshape(g.shape[0], -1)
    out_len = (n_frames - 1) * hop + win
    full = np.zeros((g.shape[0], out_len), dtype=np.float64)
    off = win // 2 if center else 0
    full[:, off: off + g.shape[-1]] = g
    fr = _unfold(full, win, hop) * np.asarray(window, dtype=np.float64)
    if n_fft > win:
        fr = np.pad(fr, [(0, 0), (0, 0), (0, n_fft - win)])
    return 4.0 / n_fft * (fr @ _cos_kernel(n_fft))
