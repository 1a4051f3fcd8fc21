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


def d_first():
    """round 5: the discriminator's first layer on csrc/dfirst.hip, 2B = 64 samples"""
    N = 2 * B
    spec = _ops.ConvSpec(4, 64, 4, 2, 2, 0, False, 0, False, _ops.ACT_LRELU)
    d = spec.desc(N, 512, 256, HALF)
    x = torch.zeros(N, 512, 256, 8, device="cuda", dtype=HALF); x[..., :4] = torch.randn(N, 512, 256, 4, device="cuda").to(HALF)
    w = torch.randn(64, 4, 4, 4, device="cuda") * 0.1
    b = torch.zeros(64, device="cuda")
    Ho, Wo = spec.out_size(d)
    y = torch.empty(N, Ho, Wo, 64, device="cuda", dtype=HALF)
    wp = spec.packed(w, 0, d)
    ws = torch.empty(max(L.p2phd_conv_fwd_workspace_bytes(C.byref(d)), 256), dtype=torch.uint8, device="cuda")
    return lambda: _ops.check(L.p2phd_conv_fwd(C.byref(d), _ops.ptr(x), _ops.ptr(wp), _ops.ptr(b), _ops.ACT_LRELU, _ops.ptr(y), None, _ops.ptr(ws), _ops.stream_ptr()))


def d_last():
    """round 5: the discriminator's head on csrc/dlast.hip, 2B = 64 samples: forward; input gradient with addend + fused sums"""
    N, H, W, CH = 2 * B, 66, 34, 512
    spec = _ops.ConvSpec(CH, 1, 4, 1, 2, 0, False, 0, False, _ops.ACT_NONE)
    d = spec.desc(N, H, W, HALF)
    Ho, Wo = spec.out_size(d)
    x = torch.randn(N, H, W, CH, device="cuda").to(HALF)
    w = torch.randn(1, CH, 4, 4, device="cuda") * 0.05
    b = torch.zeros(1, device="cuda")
    dy = torch.zeros(N, Ho, Wo, 8, device="cuda", dtype=HALF); dy[..., 0] = torch.randn(N, Ho, Wo, device="cuda").to(HALF)
    addend = torch.randn_like(x)
    st = torch.zeros(N, CH, 2, device="cuda"); st[..., 1] = H * W
    y = torch.empty(N, Ho, Wo, 8, device="cuda", dtype=HALF)
    gx = torch.empty_like(x)
    bst = torch.zeros(N, CH, 2, device="cuda")
    wp0, wp1 = spec.packed(w, 0, d), spec.packed(w, 1, d)
    ws = torch.empty(max(L.p2phd_conv_fwd_workspace_bytes(C.byref(d)), L.p2phd_conv_dgrad_bsum_workspace_bytes(C.byref(d)), 256), dtype=torch.uint8, device="cuda")
    fwd = lambda: _ops.check(L.p2phd_conv_fwd(C.byref(d), _ops.ptr(x), _ops.ptr(wp0), _ops.ptr(b), 0, _ops.ptr(y), None, _ops.ptr(ws), _ops.stream_ptr()))
    dgr = lambda: _ops.check(L.p2phd_conv_dgrad_bsum(C.byref(d), _ops.ptr(dy), _ops.ptr(wp1), _ops.ptr(addend), _ops.ptr(gx), _ops.ptr(x), _ops.ptr(st),
                                                     _ops.ACT_LRELU, 1e-5, _ops.ptr(bst), _ops.ptr(ws), _ops.stream_ptr()))
    return fwd, dgr


df = d_first()
dl_fwd, dl_dgr = d_last()
for f in (t_fwd, d_fwd, t_wg, m_s, m_u, df, dl_fwd, dl_dgr):
    for _ in range(10):
        f()
    torch.cuda.synchronize()
print("done")
