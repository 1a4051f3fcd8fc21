#!/usr/bin/env python3
"""Run one conv layer's forward (incl. InstanceNorm statistics) a few times: target for rocprofv3 --kernel-trace --stats.
usage: one_layer.py cin cout H W k stride pad pad_mode transposed opad [batch]"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pix2pixhdaudiosr_amd import _ops
a = [int(v) for v in sys.argv[1:]]
cin, cout, H, W, k, stride, pad, pad_mode, transposed, opad = a[:10]
batch = a[10] if len(a) > 10 else 32
spec = _ops.ConvSpec(cin, cout, k, stride, pad, pad_mode, bool(transposed), opad, True, _ops.ACT_RELU)
x = torch.randn(batch, H, W, _ops.cpitch(cin), device="cuda").to(torch.bfloat16)
w = (torch.randn(cin, cout, k, k, device="cuda") if transposed else torch.randn(cout, cin, k, k, device="cuda")) * 0.02
d = spec.desc(batch, H, W, torch.bfloat16)
Ho, Wo = spec.out_size(d)
wp = spec.packed(w, 0, d)
y = torch.empty(batch, Ho, Wo, _ops.cpitch(cout), device="cuda", dtype=torch.bfloat16)
stats = torch.zeros(batch, _ops.cpitch(cout), 2, device="cuda")
L = _ops.lib()
nbytes = L.p2phd_conv_fwd_workspace_bytes(C.byref(d))
ws = torch.empty(max(nbytes, 16), dtype=torch.uint8, device="cuda")
for _ in range(6):
    _ops.check(L.p2phd_conv_fwd(C.byref(d), _ops.ptr(x), _ops.ptr(wp), None, 0, _ops.ptr(y), _ops.ptr(stats), _ops.ptr(ws), _ops.stream_ptr()))
torch.cuda.synchronize()
print("done", Ho, Wo)
