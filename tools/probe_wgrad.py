#!/usr/bin/env python3
"""Cycle breakdown of the weight-gradient kernel on one layer (needs a -DP2PHD_PROBE build selected with P2PHD_LIB).
usage: probe_wgrad.py cin cout H W k stride pad pad_mode [batch]"""
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
torch.cuda.synchronize()
buf = (C.c_ulonglong * 8)()
L.p2phd_debug_probe(buf, 1)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); call(); e1.record(); torch.cuda.synchronize()
L.p2phd_debug_probe(buf, 1)
wait, bar, tot, nst, _, _, nw = (float(v) for v in buf[:7])
print(f"wgrad {' '.join(sys.argv[1:])}: {e0.elapsed_time(e1)*1e3:.0f} us incl. unpack; per workgroup: loop+prologue {tot/nw:.0f} cyc, {nst/nw:.0f} steps "
      f"({tot/nst:.0f} cyc/step: wait {wait/nst:.0f}, barrier {bar/nst:.0f}); workgroups {nw:.0f}")
