#!/usr/bin/env python3
"""Why two runs of the same code on the golden clip can differ by 1.6 % in the generator gradient: the L1 feature-matching
loss has the gradient sign(fake - real) / n per element, and on tests/golden/model_step.npz ONE element of the third
discriminator feature (scale 0) has |fake - real| of the order of the fp32 rounding noise that the atomically summed
InstanceNorm statistics leave in `fake`.  Prints, over several runs, the smallest |fake - real| of every matched feature
and whether the sign pattern changed.  (Found with the backward trace hook _ops._BWD_TRACE; the guard-band run of
tests/test_gpu_guard.py rules out stray writes.)"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    from test_gpu_model import _model
    from pix2pixhdaudiosr_amd import _ops
    g = np.load(os.path.join(ROOT, "tests", "golden", "model_step.npz"))
    lr, hr = torch.from_numpy(g["lr"]).cuda(), torch.from_numpy(g["hr"]).cuda()
    signs = None
    for run in range(8):
        m = _model(g, mask=False)
        m._phase_a_forward(lr, hr)
        _ops._BWD_TRACE[0] = []
        for r, _ in m._g_stages():
            r()
        tr, _ops._BWD_TRACE[0] = _ops._BWD_TRACE[0], None
        das = [t[2] for t in tr if isinstance(t[0], tuple) and t[0][1] == 1]
        s = [torch.sign(d) for d in das]
        if signs is None:
            signs = s
        flips = [int((a != b).sum()) for a, b in zip(s, signs)]
        print(f"run {run}: sign flips of d(L1)/d(fake) per feature vs run 0: {flips}")


if __name__ == "__main__":
    main()
