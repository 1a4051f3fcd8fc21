#!/usr/bin/env python3
"""Target for one rocprofv3 --pmc pass over the WHOLE GlobalGenerator forward + backward (north_star: ">= 40 % MFMA utilisation on
GlobalGenerator fwd+bwd at 512x256 bf16"): the benchmarked generator (bench.make_opt(32), configs[1]) on a [32, 2, 512, 256] input,
every weight gradient into the flat buffer, run EAGERLY three times (same call as bench.time_generator, which replays it as a graph).
An MDCT launch in front of every pass is the marker tools/summarize_generator_pmc.py cuts the dispatch list at: the last segment is
the measured pass (workspaces, packed weights and the autograd graph shape exist by then).

    rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d DIR -- python3 tools/pmc_generator.py
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench  # noqa: E402
from pix2pixhdaudiosr_amd import _ops  # noqa: E402
from pix2pixhdaudiosr_amd.models import mdct as MM  # noqa: E402
from pix2pixhdaudiosr_amd.models.models import create_model  # noqa: E402

B = 32
torch.manual_seed(0)
model = create_model(bench.make_opt(B))
netG, optG = model.netG, model.optimizer_G
x = torch.rand(B, 2, 512, 256, device="cuda")
marker = MM.MDCT4(n_fft=1024, hop_length=512, win_length=1024, device="cuda")
sig = torch.randn(2, 8192, device="cuda")
gy = None
for it in range(3):
    marker(sig)                                                     # marker launch (mdct4_*_kernel): not part of the generator
    _ops.begin_step(model.device)
    y = netG.forward_physical(netG.input_physical(x))
    _ops.end_arena(model.device)
    if gy is None:
        gy = (torch.randn(y.shape, device="cuda") * 1e-3).to(y.dtype)
        gy[..., 2:] = 0
    y.backward(gy, inputs=list(optG._params))
    torch.cuda.synchronize()
print("done")
