#!/usr/bin/env python3
"""A/B of the tile choice (p2phd_set_option gconv_bm) on the generator's short-K outer layers (48<->96<->192<->384, conv
and transposed conv) at configs[1] geometry: forward and input gradient, HIP events.  usage: ab_outer.py [B]"""
import ctypes as C
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pix2pixhdaudiosr_amd import _ops

B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
L = _ops.lib()


def timeit(fn, iters=8):
    fn(); fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


def layer(name, cin, cout, k, stride, pad, transposed, opad, H, W):
    dt = torch.bfloat16
    spec = _ops.ConvSpec(cin, cout, k, stride, pad, 0, transposed, opad, True, 0)
    d = spec.desc(B, H, W, dt)
    Ho, Wo = spec.out_size(d)
    x = torch.randn(B, H, W, _ops.cpitch(cin), device="cuda").to(dt)
    w = torch.randn((cin, cout, k, k) if transposed else (cout, cin, k, k), device="cuda") * 0.02
    y = torch.empty(B, Ho, Wo, _ops.cpitch(cout), device="cuda", dtype=dt)
    dy = torch.randn_like(y)
    stats = torch.zeros(B, _ops.cpitch(cout), 2, device="cuda")
    gx = torch.empty_like(x)
    ws = _ops.workspace(max(L.p2phd_conv_fwd_workspace_bytes(C.byref(d)), L.p2phd_conv_dgrad_workspace_bytes(C.byref(d)), 1), "cuda")
    row = [f"{name:24s}"]
    for bm in (0, 128, 256, 192):
        L.p2phd_set_option(b"gconv_bm", bm)
        wp0 = spec.packed(w, 0, d); wp1 = spec.packed(w, 1, d)
        f = lambda: _ops.check(L.p2phd_conv_fwd(C.byref(d), _ops.ptr(x), _ops.ptr(wp0), None, 0, _ops.ptr(y), _ops.ptr(stats), _ops.ptr(ws), _ops.stream_ptr()))
        g = lambda: _ops.check(L.p2phd_conv_dgrad(C.byref(d), _ops.ptr(dy), _ops.ptr(wp1), None, _ops.ptr(gx), _ops.ptr(ws), _ops.stream_ptr()))
        try:
            row.append(f"bm{bm}: fwd {timeit(f):6.0f} dgrad {timeit(g):6.0f}")
        except Exception as e:
            row.append(f"bm{bm}: {type(e).__name__}")
    L.p2phd_set_option(b"gconv_bm", 0)
    print(" | ".join(row), flush=True)


ch, H, W = 48, 512, 256
for i in range(4):
    layer(f"G down {ch}->{ch*2} s2", ch, ch * 2, 3, 2, 1, False, 0, H, W)
    ch, H, W = ch * 2, H // 2, W // 2
for i in range(4):
    layer(f"G up {ch}->{ch//2} convT", ch, ch // 2, 3, 2, 1, True, 1, H, W)
    ch, H, W = ch // 2, H * 2, W * 2
