#!/usr/bin/env python3
"""Timing of the generator's 7x7 end layers at configs[1] size (B=32, 512x256): forward with / without InstanceNorm
statistics, dedicated kernel vs the generic W-fold path."""
import ctypes as C
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pix2pixhdaudiosr_amd import _ops, _lib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench

L = _ops.lib()
B = int(os.environ.get("B", "32"))
H, W = 512, 256
def run(cin, cout, tag):
    dt = torch.bfloat16
    spec = _ops.ConvSpec(cin, cout, 7, 1, 3, 1, False, 0, True, 0)
    d = spec.desc(B, H, W, dt)
    x = torch.randn(B, H, W, _ops.cpitch(cin), device="cuda").to(dt)
    w = torch.randn(cout, cin, 7, 7, device="cuda") * 0.02
    bias = torch.randn(cout, device="cuda")
    y = torch.empty(B, H, W, _ops.cpitch(cout), device="cuda", dtype=dt)
    stats = torch.zeros(B, _ops.cpitch(cout), 2, device="cuda")
    ws = torch.empty(max(L.p2phd_conv_fwd_workspace_bytes(C.byref(d)), 256), dtype=torch.uint8, device="cuda")
    nbytes = (x.numel() + y.numel()) * 2
    for generic in (0, 1):
        _lib.check(L.p2phd_set_option(b"c7_generic", generic))
        wp = _ops.ConvSpec(cin, cout, 7, 1, 3, 1, False, 0, True, 0).packed(w, 0, d)
        for st in (stats, None):
            f = lambda: _ops.check(L.p2phd_conv_fwd(C.byref(d), _ops.ptr(x), _ops.ptr(wp), _ops.ptr(bias), 0, _ops.ptr(y), _ops.ptr(st), _ops.ptr(ws), _ops.stream_ptr()))
            t = bench.time_graphed(f, 10)
            print(f"{tag} {'generic' if generic else 'dedicated'} stats={'yes' if st is not None else 'no '}: {t*1e6:7.1f} us  {nbytes/t/1e12:.2f} TB/s", flush=True)
    _lib.check(L.p2phd_set_option(b"c7_generic", 0))
def run_out(cin=48):
    dt = torch.bfloat16
    spec = _ops.ConvSpec(cin, 2, 7, 1, 3, 1, False, 0, False, 2)
    d = spec.desc(B, H, W, dt)
    x = torch.randn(B, H, W, _ops.cpitch(cin), device="cuda").to(dt)
    w = torch.randn(2, cin, 7, 7, device="cuda") * 0.02
    bias = torch.randn(2, device="cuda")
    y = torch.empty(B, H, W, 8, device="cuda", dtype=dt)
    ws = torch.empty(max(L.p2phd_conv_fwd_workspace_bytes(C.byref(d)), 256), dtype=torch.uint8, device="cuda")
    wp = spec.packed(w, 0, d)
    nbytes = (x.numel() + y.numel()) * 2
    for abl in (0, 1, 2, 3, 4, 7):
        _lib.check(L.p2phd_set_option(b"c7_abl", abl))
        f = lambda: _ops.check(L.p2phd_conv_fwd(C.byref(d), _ops.ptr(x), _ops.ptr(wp), _ops.ptr(bias), 2, _ops.ptr(y), None, _ops.ptr(ws), _ops.stream_ptr()))
        t = bench.time_graphed(f, 10)
        print(f"c7 {cin}->2 fwd abl={abl} (1 no MFMA, 2 no acc adds, 4 no row fetch): {t*1e6:7.1f} us  {nbytes/t/1e12:.2f} TB/s", flush=True)
    _lib.check(L.p2phd_set_option(b"c7_abl", 0))
if os.environ.get("OUT"):
    run_out(); sys.exit(0)
run(2, 48, "c7 2->48")
y = torch.empty(B, H, W, 48, device="cuda", dtype=torch.bfloat16); y2 = torch.randn(B, H, W, 48, device="cuda").to(torch.bfloat16)
t = bench.time_graphed(lambda: y.zero_(), 10); print(f"memset 403 MB: {t*1e6:.1f} us  {y.numel()*2/t/1e12:.2f} TB/s written")
t = bench.time_graphed(lambda: y.copy_(y2), 10); print(f"copy 403 MB: {t*1e6:.1f} us  {y.numel()*4/t/1e12:.2f} TB/s read+written")
