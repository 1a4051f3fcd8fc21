#!/usr/bin/env python3
"""Run-to-run bit identity of the whole configs[1] training step at the benchmarked size (B = 32, bf16): N times forward + both
backward passes from the SAME weights and input (no optimiser update in between), losses and every gradient element that
carries signal compared bit for bit with the first run.  Every reduction on the path has a fixed order and every LDS ring
its waits counted by hand -- a miscounted wait shows up here as one differing run in a few hundred launches
(tests/test_gpu_conv.py::test_halo_loop_equals_the_generic_loop found one that way).

    [FP16=1] python tools/stress_determinism.py [N = 40]            (GPU)  ->  profiles/r04_determinism_stress.log
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch  # noqa: E402

from bench import make_opt  # noqa: E402
from conftest import noise_bias_keys  # noqa: E402
from pix2pixhdaudiosr_amd.models.models import create_model  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 40
B = int(os.environ.get("B", "32"))
torch.manual_seed(1234)
opt = make_opt(B, fp16_storage=os.environ.get("FP16") == "1")       # FP16=1: the fp16 build of the library (unscaled losses here)
opt.mask = False                                                 # no random mask rows: identical inputs by construction
m = create_model(opt)
gen = torch.Generator().manual_seed(5)
T = 255 * 512
hr = (0.1 * torch.randn(B, T, generator=gen)).cuda()
lr = (0.1 * torch.randn(B, T, generator=gen)).cuda()


def signal_mask(net, o):
    names = [k for k, _ in net.named_parameters()]
    nb = noise_bias_keys(names)
    mask = torch.ones(o._total, dtype=torch.bool, device=o.flat_g.device)
    for k, p, off in zip(names, o._params, o._offs):
        if k in nb:
            mask[off:off + p.numel()] = False
    return mask


mG, mD = signal_mask(m.netG, m.optimizer_G), signal_mask(m.netD, m.optimizer_D)
first, bad = None, 0
for i in range(N):
    ld = m._phase_a(lr, hr)
    m._phase_b()
    m.optimizer_G._flush_fresh(); m.optimizer_D._flush_fresh()
    torch.cuda.synchronize()
    cur = ({k: float(v) for k, v in ld.items()}, m.optimizer_G.flat_g[mG].clone(), m.optimizer_D.flat_g[mD].clone())
    if first is None:
        first = cur
        print(f"B={B} run 0: losses {cur[0]}; |gG| {float(cur[1].norm()):.6e} |gD| {float(cur[2].norm()):.6e}", flush=True)
        continue
    same = cur[0] == first[0] and torch.equal(cur[1], first[1]) and torch.equal(cur[2], first[2])
    if not same:
        bad += 1
        dG = int((cur[1] != first[1]).sum()); dD = int((cur[2] != first[2]).sum())
        print(f"run {i}: DIFFERS: losses equal {cur[0] == first[0]}, {dG} generator / {dD} discriminator gradient elements differ", flush=True)
print(f"{N} runs of forward + both backward passes from the same state: {N - 1 - bad} bit-identical to the first, {bad} differing")
sys.exit(1 if bad else 0)
