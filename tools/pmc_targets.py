#!/usr/bin/env python3
"""Targets for the rocprofv3 --pmc passes (program directly after `--`): the three MFMA kernels the roofline talks about,
ten launches each -- trunk forward (gconv 256x192), discriminator 256->512 forward (gconv 256x256), trunk weight gradient
(wgrad_kernel<256>) -- and (round 4) the two marching kernels of the generator's outermost stride-2 layers, at BASELINE
configs[1] shapes, B = 32, bf16 (DT=f16 in the environment: the fp16 build, for the clock comparison in DESIGN.md)."""
import ctypes as C
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pix2pixhdaudiosr_amd import _ops

HALF = torch.float16 if os.environ.get("DT") == "f16" else torch.bfloat16
L = _ops.lib_for(HALF)
B = 32
def layer(cin, cout, k, pad, pad_mode, H, W, stride=1, transposed=False):
    spec = _ops.ConvSpec(cin, cout, k, stride, pad, pad_mode, transposed, 1 if transposed else 0, True, 0)
    d = spec.desc(B, H, W, HALF)
    Ho, Wo = spec.out_size(d)
    x = torch.randn(B, H, W, cin, device="cuda").to(HALF)
    w = torch.randn((cin, cout, k, k) if transposed else (cout, cin, k, k), device="cuda") * 0.02
    y = torch.empty(B, Ho, Wo, cout, device="cuda", dtype=HALF)
    dy = torch.randn_like(y)
    stats = torch.zeros(B, cout, 2, device="cuda")
    gw = torch.empty_like(w)
    wsb = max(L.p2phd_conv_fwd_workspace_bytes(C.byref(d)), L.p2phd_conv_wgrad_workspace_bytes(C.byref(d)), 256)
    ws = torch.empty(wsb, dtype=torch.uint8, device="cuda")
    wp = spec.packed(w, 0, d)
    fwd = lambda: _ops.check(L.p2phd_conv_fwd(C.byref(d), _ops.ptr(x), _ops.ptr(wp), None, 0, _ops.ptr(y), _ops.ptr(stats), _ops.ptr(ws), _ops.stream_ptr()))
    wg = lambda: _ops.check(L.p2phd_conv_wgrad(C.byref(d), _ops.ptr(x), _ops.ptr(dy), _ops.ptr(gw), None, _ops.ptr(ws), _ops.stream_ptr()))
    return fwd, wg

t_fwd, t_wg = layer(768, 768, 3, 1, 1, 32, 16)
d_fwd, _ = layer(256, 512, 4, 2, 0, 65, 33)
m_s, _ = layer(48, 96, 3, 1, 0, 512, 256, stride=2)
m_u, _ = layer(96, 48, 3, 1, 0, 256, 128, stride=2, transposed=True)
for f in (t_fwd, d_fwd, t_wg, m_s, m_u):
    for _ in range(10):
        f()
    torch.cuda.synchronize()
print("done")
