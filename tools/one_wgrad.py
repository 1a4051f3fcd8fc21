#!/usr/bin/env python3
"""Run one layer's weight gradient a few times: target for rocprofv3 (--pmc / --kernel-trace).
usage: one_wgrad.py cin cout H W k stride pad pad_mode [batch]"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pix2pixhdaudiosr_amd import _ops
a = [int(v) for v in sys.argv[1:]]
cin, cout, H, W, k, stride, pad, pad_mode = a[:8]
batch = a[8] if len(a) > 8 else 32
spec = _ops.ConvSpec(cin, cout, k, stride, pad, pad_mode, False, 0, True, _ops.ACT_RELU)
x = torch.randn(batch, H, W, _ops.cpitch(cin), device="cuda").to(torch.bfloat16)
d = spec.desc(batch, H, W, torch.bfloat16)
Ho, Wo = spec.out_size(d)
dy = torch.randn(batch, Ho, Wo, _ops.cpitch(cout), device="cuda").to(torch.bfloat16)
dw = torch.empty(cout, cin, k, k, device="cuda")
L = _ops.lib()
ws = torch.empty(max(L.p2phd_conv_wgrad_workspace_bytes(C.byref(d)), 16), dtype=torch.uint8, device="cuda")
for _ in range(4):
    _ops.check(L.p2phd_conv_wgrad(C.byref(d), _ops.ptr(x), _ops.ptr(dy), _ops.ptr(dw), None, _ops.ptr(ws), _ops.stream_ptr()))
torch.cuda.synchronize()
print("done")
