#!/usr/bin/env python3
"""Soak: N graph-replayed steps of configs[1] at B=32; reports step time, device step counter, finiteness of every parameter
and the loss trajectory (a long run catches leaks, arena overflow, drift into NaN)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bench import make_opt
from pix2pixhdaudiosr_amd.models.models import create_model
n = int(sys.argv[1]) if len(sys.argv) > 1 else 200
opt = make_opt(32)
torch.manual_seed(1234)
model = create_model(opt)
T = 255 * opt.hop_length
hr = 0.1 * torch.randn(32, T, device="cuda"); lr = 0.1 * torch.randn(32, T, device="cuda")
for _ in range(3):
    model.train_step_graphed(lr, hr)
torch.cuda.synchronize()
m0 = torch.cuda.memory_allocated()
traj = []
t0 = time.perf_counter()
for i in range(n):
    ld = model.train_step_graphed(lr, hr)
    if i % max(1, n // 8) == 0:
        traj.append({k: round(float(v), 3) for k, v in ld.items()})
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / n
ok = all(torch.isfinite(p).all().item() for p in model.parameters())
print(f"{n} replays: {dt*1e3:.2f} ms/step, steps_taken {model.optimizer_G.steps_taken()}, params finite {ok}, "
      f"allocated {m0/2**30:.1f} -> {torch.cuda.memory_allocated()/2**30:.1f} GiB, reserved {torch.cuda.memory_reserved()/2**30:.1f} GiB")
for t in traj:
    print("  ", t)
