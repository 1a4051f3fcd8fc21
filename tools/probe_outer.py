#!/usr/bin/env python3
"""Where a tile of a short-K outer layer spends its cycles: needs a library built with -DP2PHD_PROBE (tools/ablate_gconv.sh),
selected with P2PHD_LIB.  usage: probe_outer.py cin cout H W stride transposed [dgrad]"""
import ctypes as C
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pix2pixhdaudiosr_amd import _ops
a = sys.argv[1:]
cin, cout, H, W, stride, transposed = (int(v) for v in a[:6])
dgrad = len(a) > 6
B = 32
L = _ops.lib()
spec = _ops.ConvSpec(cin, cout, 3, stride, 1, 0, bool(transposed), 1 if transposed else 0, True, 0)
d = spec.desc(B, H, W, torch.bfloat16)
Ho, Wo = spec.out_size(d)
x = torch.randn(B, H, W, _ops.cpitch(cin), device="cuda").to(torch.bfloat16)
w = torch.randn((cin, cout, 3, 3) if transposed else (cout, cin, 3, 3), device="cuda") * 0.02
y = torch.empty(B, Ho, Wo, _ops.cpitch(cout), device="cuda", dtype=torch.bfloat16)
dy = torch.randn_like(y); gx = torch.empty_like(x)
stats = torch.zeros(B, _ops.cpitch(cout), 2, device="cuda")
ws = _ops.workspace(max(L.p2phd_conv_fwd_workspace_bytes(C.byref(d)), L.p2phd_conv_dgrad_workspace_bytes(C.byref(d)), 1), "cuda")
wp = spec.packed(w, 1 if dgrad else 0, d)
if dgrad:
    f = lambda: _ops.check(L.p2phd_conv_dgrad(C.byref(d), _ops.ptr(dy), _ops.ptr(wp), None, _ops.ptr(gx), _ops.ptr(ws), _ops.stream_ptr()))
else:
    f = lambda: _ops.check(L.p2phd_conv_fwd(C.byref(d), _ops.ptr(x), _ops.ptr(wp), None, 0, _ops.ptr(y), _ops.ptr(stats), _ops.ptr(ws), _ops.stream_ptr()))
f(); f(); torch.cuda.synchronize()
buf = (C.c_ulonglong * 8)()
if not hasattr(L, "p2phd_debug_probe"):                 # any other library: timing only
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        f()
    e1.record(); torch.cuda.synchronize()
    print(os.environ.get("P2PHD_LIB", "default").split("/")[-1], os.environ.get("P2PHD_OPTIONS", ""), " ".join(a), f"{e0.elapsed_time(e1)*100:.0f} us")
    sys.exit(0)
L.p2phd_debug_probe(buf, 1)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); f(); e1.record(); torch.cuda.synchronize()
L.p2phd_debug_probe(buf, 1)
v = [float(q) for q in buf]
n = v[6]
print(os.environ.get("P2PHD_LIB", "default").split("/")[-1], os.environ.get("P2PHD_OPTIONS", ""), " ".join(a), f"{e0.elapsed_time(e1)*1e3:.0f} us, {n:.0f} workgroups (x launches inside the call)")
if "fine" in os.environ.get("P2PHD_LIB", ""):
    names = ["table build", "addresses + DMA issue", "first wait + barrier + frags", "K loop", "stats + LDS staging", "barrier", "(count)", "store loop"]
else:
    names = ["loop: vmcnt wait", "loop: barrier", "loop: total", "(K slabs)", "prologue", "epilogue", "(count)", "whole tile"]
for k, nm in enumerate(names):
    print(f"   {nm:32s} {v[k]/n:10.0f} cycles/workgroup")
