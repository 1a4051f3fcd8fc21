#!/usr/bin/env python3
"""Where does the bf16 generator drift from the fp32 one?  configs[1] GlobalGenerator (real widths, 512x256, one sample,
oracle-initialised weights, a dB spectrogram of noise as input): relative L2 difference of every step's output."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from oracle import model as OM, mdct4 as M4
from pix2pixhdaudiosr_amd.models import networks as PN
from pix2pixhdaudiosr_amd import _ops
oo = OM.default_opt(ngf=48, netG="global", n_downsample_global=4, n_blocks_global=9, mask=False)
pG = OM.N.init_params(OM.netG_spec(oo), seed=1)
hr, lr, _ = OM.synthetic_batch(1, oo, seed=6)
x, _, _ = OM.to_spectro(lr, oo, M4.kbdwin(oo.win_length), mask=False)
outs = {}
for name, dt in (("f32", torch.float32), ("bf16", torch.bfloat16)):
    net = PN.define_G(2, 2, 48, "global", 4, 9, 0, 3, "instance", [], dtype=dt, verbose=False)
    net.load_state_dict({k: pG[k] for k in net.state_dict().keys()})
    net = net.cuda(); _ops.bump_weight_epoch()
    n_steps = len(net._steps("model"))
    cuts = []
    center = os.environ.get("CENTER", "1") == "1"
    xin = net.input_physical(x.cuda()) if center else net._to_phys(x.cuda())
    _ops._STATS_TRACE[0] = [] if name == "f32" else None
    with torch.no_grad():
        net.forward_physical(xin, cuts=cuts, cut_after=set(range(n_steps)))
    if name == "f32":
        for j, (spec, st, hw) in enumerate(_ops._STATS_TRACE[0]):
            mean = st[0, :spec.cout, 0]; sd = (st[0, :spec.cout, 1] / hw).sqrt()
            r = (mean.abs() / sd.clamp_min(1e-12))
            print(f"conv {j:2d} ({spec.cin:3d}->{spec.cout:3d}): |mean|/sigma of the raw output per channel: median {float(r.median()):.2f}  p90 {float(r.quantile(0.9)):.2f}  max {float(r.max()):.2f}")
        _ops._STATS_TRACE[0] = None
    outs[name] = [c.float().cpu() for c in cuts]
for i, (a, b) in enumerate(zip(outs["f32"], outs["bf16"])):
    C = a.shape[-1]
    e = float((a - b).norm() / a.norm())
    print(f"step {i:2d} out {tuple(a.shape)}: bf16 vs fp32 rel {e:.4f}")
