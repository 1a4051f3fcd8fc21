import sys, os
sys.path.insert(0, os.getcwd())
import torch
from bench import time_graphed
from pix2pixhdaudiosr_amd.models import mdct as MM
from pix2pixhdaudiosr_amd.util.util import kbdwin
from pix2pixhdaudiosr_amd import _lib
L = _lib.lib()
for n_fft, frames in ((1024, 256), (2048, 512)):
    hop = n_fft // 2; T = (frames - 1) * hop
    w = kbdwin(n_fft).cuda(); tables = MM._Tables.get(n_fft, w.device)
    for rows in (32, 64):
        x = 0.1 * torch.randn(rows, T, device="cuda")
        sp, _, nf = MM.frame_layout(32, T, hop, n_fft, True)
        ref = None
        for it in (1, 2, 3, 4, 0):
            _lib.check(L.p2phd_set_option(b"mdct_iters", it))
            S = MM._run_mdct(x, n_fft, hop, n_fft, w, tables, sp, nf, 1.0)
            if ref is None: ref = S.clone()
            assert torch.equal(S, ref), (n_fft, rows, it)
            t = time_graphed(lambda: MM._run_mdct(x, n_fft, hop, n_fft, w, tables, sp, nf, 1.0))
            by = rows * nf * 4 * n_fft
            print(f"n_fft {n_fft} rows {rows} iters {it}: {t*1e6:.1f} us {by/t/1e12:.2f} TB/s", flush=True)
