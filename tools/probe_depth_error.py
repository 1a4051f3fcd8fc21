#!/usr/bin/env python3
"""How fp32 rounding grows through the 28 conv layers of the configs[1] generator at 512x256: the HIP fp32 path and the
torch-CPU fp32 oracle, layer by layer, each against the SAME network evaluated in fp64 (the exact result up to 1e-15).
Answers whether a whole-network difference of a few 1e-4 between the two fp32 implementations is a defect of one of them
or the distance any two fp32 summation orders end up at."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def rel(a, b):
    a, b = a.double(), b.double()
    return float((a - b).norm() / b.norm())


def layerwise(p, x, n_down, n_blocks):
    from oracle import networks as N
    outs = []
    x = N._c7(p, "model.1", x); outs.append(x)
    idx = 4
    for _ in range(n_down):
        x = N._down(p, f"model.{idx}", x); outs.append(x); idx += 3
    for _ in range(n_blocks):
        x = N._resblock(p, f"model.{idx}", x); outs.append(x); idx += 1
    for _ in range(n_down):
        x = N._up(p, f"model.{idx}", x); outs.append(x); idx += 3
    x = torch.tanh(N._c7(p, f"model.{idx + 1}", x, norm_relu=False)); outs.append(x)
    return outs


def main():
    from oracle import networks as N
    from pix2pixhdaudiosr_amd.models import networks as PN
    from pix2pixhdaudiosr_amd import _ops
    torch.set_num_threads(max(1, min(16, len(os.sched_getaffinity(0)))))
    spec = N.global_generator_spec(2, 2, 48, 4, 9)
    p = N.init_params(spec, seed=1)
    x = torch.rand(1, 2, 512, 256, generator=torch.Generator().manual_seed(9))
    with torch.no_grad():
        o32 = layerwise(p, x, 4, 9)
        o64 = layerwise({k: v.double() for k, v in p.items()}, x.double(), 4, 9)
    net = PN.define_G(2, 2, 48, "global", 4, 9, 0, 0, "instance", [], dtype=torch.float32, verbose=False)
    net.load_state_dict({k: p[k] for k in net.state_dict().keys()})
    net = net.cuda()
    _ops.bump_weight_epoch()
    steps = net._steps('model')
    cuts = []
    with torch.no_grad():
        net.forward_physical(net._to_phys(x.cuda()), cuts=cuts, cut_after=set(range(len(steps))))
    print(f"{'layer':>5s} {'channels':>8s} {'HIP f32 vs f64':>15s} {'CPU f32 vs f64':>15s} {'HIP vs CPU f32':>15s}")
    for j, (c, a32, a64) in enumerate(zip(cuts, o32, o64)):
        ch = a64.shape[1]
        h = c.permute(0, 3, 1, 2)[:, :ch].float().cpu()
        print(f"{j:5d} {ch:8d} {rel(h, a64):15.3e} {rel(a32, a64):15.3e} {rel(h, a32):15.3e}", flush=True)


if __name__ == "__main__":
    main()
