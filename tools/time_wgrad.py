#!/usr/bin/env python3
"""Time one layer's weight gradient (kernel + slab sum) with whatever library P2PHD_LIB selects.
usage: time_wgrad.py cin cout H W k stride pad pad_mode [batch]"""
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
call = lambda: _ops.check(L.p2phd_conv_wgrad(C.byref(d), _ops.ptr(x), _ops.ptr(dy), _ops.ptr(dw), None, _ops.ptr(ws), _ops.stream_ptr()))
for _ in range(3):
    call()
res = []
for r in range(5):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(10):
        call()
    e1.record(); torch.cuda.synchronize()
    res.append(e0.elapsed_time(e1) * 100)
res.sort()
print(os.environ.get("P2PHD_LIB", "default").split("/")[-1], " ".join(sys.argv[1:]), f"wgrad+unpack median {res[2]:.1f} us min {res[0]:.1f} us")
