"""Host-side helpers of the hot path with the reference's names (util/util.py)."""
import torch


def kbdwin(N: int, beta: float = 12.0, device='cpu') -> torch.Tensor:
    """MATLAB-style Kaiser-Bessel-derived window, same arithmetic as util/util.py:186-193
    of the reference (a one-off host constant: the kernels only read it)."""
    assert N % 2 == 0, "N must be even"
    w = torch.kaiser_window(window_length=N // 2 + 1, beta=beta * torch.pi, periodic=False, device=device)
    w_sum = w.sum()
    wdw_half = torch.sqrt(torch.cumsum(w, dim=0) / w_sum)[:-1]
    return torch.cat((wdw_half, wdw_half.flip(dims=(0,))), dim=0)
