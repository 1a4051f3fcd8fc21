#!/usr/bin/env python3
"""Debug: configs[0] step, finiteness of every stage (run with P2PHD_OPTIONS / P2PHD_DPAIR variants)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
from oracle import model as OM
from test_gpu_fullsize import _opt, _load_from
from pix2pixhdaudiosr_amd.models.models import create_model
from pix2pixhdaudiosr_amd import _ops
oo = OM.default_opt(ngf=32, netG="global", n_downsample_global=4, n_blocks_global=9, mask=True)
pG = OM.N.init_params(OM.netG_spec(oo), seed=11)
pD = OM.N.init_params(OM.netD_spec(oo), seed=12)
hr, lr, noise = OM.synthetic_batch(2, oo, seed=7)
m = create_model(_opt(ngf=32, mask=True))
_load_from(m.netG, pG); _load_from(m.netD, pD)
_ops.bump_weight_epoch()
for rep in range(3):
    enc = m.encode_input(lr, None, hr, None, noise=noise)
    ls, hs = enc[0], enc[2]
    print(rep, "lr_spectro finite", bool(torch.isfinite(ls).all()), float(ls.min()), float(ls.max()), "hr", bool(torch.isfinite(hs).all()), float(hs.min()), float(hs.max()))
    with torch.no_grad():
        sr = m.netG(ls)
        print(rep, "  sr finite", bool(torch.isfinite(sr).all()), float(sr.abs().max()))
        fe = m.netD(torch.cat((ls, sr), 1))
        for i, s in enumerate(fe):
            print(rep, "  D scale", i, [f"{float(f.abs().max()):.3g}" for f in s])
    ld = m.train_step(lr, hr, noise=noise)
    print(rep, "  losses", {k: float(v) for k, v in ld.items()})
