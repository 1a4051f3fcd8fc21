#!/usr/bin/env python3
"""bf16 against fp32 over many optimisation steps (verdict r2 item 4): the SAME reduced-width configs[1] model (ngf / ndf
given on the command line, 512x256, 4 down-samplings, 9 blocks, 2-scale D), identical initial weights, a fixed set of
batches cycled in the same order, N graph-replayed steps each; prints the four losses of both runs at checkpoints and the
relative distance of the generated spectrogram on a held-out clip.
usage: soak_pair.py [steps=200] [ngf=16] [batch=2] [n_batches=4]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bench import make_opt
from pix2pixhdaudiosr_amd.models.models import create_model
from pix2pixhdaudiosr_amd import _ops

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 200
ngf = int(sys.argv[2]) if len(sys.argv) > 2 else 16
B = int(sys.argv[3]) if len(sys.argv) > 3 else 2
NB = int(sys.argv[4]) if len(sys.argv) > 4 else 4


def run(fp16):
    opt = make_opt(B, dtype_bf16=fp16)
    opt.ngf, opt.ndf, opt.mask = ngf, ngf, False
    torch.manual_seed(1234)
    m = create_model(opt)
    T = 255 * opt.hop_length
    g = torch.Generator(device="cuda").manual_seed(7)
    data = [(0.1 * torch.randn(B, T, device="cuda", generator=g), 0.1 * torch.randn(B, T, device="cuda", generator=g)) for _ in range(NB)]
    held = 0.1 * torch.randn(B, T, device="cuda", generator=g)
    traj = []
    for i in range(steps):
        lr, hr = data[i % NB]
        ld = m.train_step_graphed(lr, hr)
        if (i + 1) % max(1, steps // 8) == 0:
            traj.append((i + 1, {k: float(v) for k, v in ld.items()}))
    torch.cuda.synchronize()
    with torch.no_grad():
        sr = m.inference(held, None)[0].float().cpu()
    return traj, sr


t32, sr32 = run(False)
t16, sr16 = run(True)
print(f"steps {steps} ngf {ngf} batch {B} batches {NB}")
for (i, a), (_, b) in zip(t32, t16):
    print(f"  step {i:4d} " + "  ".join(f"{k} {a[k]:.4f}|{b[k]:.4f}" for k in a))
print(f"held-out generated spectrogram: bf16 vs fp32 rel err {float((sr16 - sr32).norm() / sr32.norm()):.3e}")
