#!/usr/bin/env python3
"""Consumer input gradient + producer InstanceNorm backward on the generator's big planes (configs[1], B = 32, bf16):
two-pass form (p2phd_conv_dgrad, then reduce + apply) against the fused form (p2phd_conv_dgrad_bsum, then apply only)."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pix2pixhdaudiosr_amd import _ops
L = _ops.lib()
B, dt = int(os.environ.get("B", "32")), torch.bfloat16

def t_us(fn, it=5):
    fn(); torch.cuda.synchronize()
    ts = []
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(it): fn()
        e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / it * 1e3)
    return sorted(ts)[2]

def case(name, cin, cout, k, stride, pad, transposed, opad, H, W, act=None):
    """consumer conv reads the producer plane [B, H, W, cin]"""
    spec = _ops.ConvSpec(cin, cout, k, stride, pad, 0, transposed, opad, True, _ops.ACT_RELU)
    d = spec.desc(B, H, W, dt)
    Ho, Wo = spec.out_size(d)
    Cpi, Cpo = _ops.cpitch(cin), _ops.cpitch(cout)
    yprev = torch.randn(B, H, W, Cpi, device="cuda").to(dt)
    stats = torch.zeros(B, Cpi, 2, device="cuda"); stats[..., 1] = H * W
    dy = torch.randn(B, Ho, Wo, Cpo, device="cuda").to(dt)
    w = torch.randn((cin, cout, k, k) if transposed else (cout, cin, k, k), device="cuda") * 0.02
    wp = spec.packed(w, 1, d)
    gx = torch.empty_like(yprev); dprev = torch.empty_like(yprev)
    bst = torch.empty(B, Cpi, 2, device="cuda")
    ws = torch.empty(max(L.p2phd_conv_dgrad_workspace_bytes(C.byref(d)), L.p2phd_conv_dgrad_bsum_workspace_bytes(C.byref(d)), 256), dtype=torch.uint8, device="cuda")
    P, S = _ops.ptr, _ops.stream_ptr
    dg = lambda: _ops.check(L.p2phd_conv_dgrad(C.byref(d), P(dy), P(wp), None, P(gx), P(ws), S()))
    two = lambda: _ops.check(L.p2phd_instnorm_act_bwd(d.dtype, P(gx), P(yprev), P(stats), P(bst), P(dprev), None, B, H * W, cin, 1e-5, (act or _ops.ACT_RELU), S()))
    dgb = lambda: _ops.check(L.p2phd_conv_dgrad_bsum(C.byref(d), P(dy), P(wp), None, P(gx), P(yprev), P(stats), _ops.ACT_RELU, 1e-5, P(bst), P(ws), S()))
    app = lambda: _ops.check(L.p2phd_instnorm_act_bwd_apply(d.dtype, P(gx), P(yprev), P(stats), P(bst), P(dprev), None, 0, B, H * W, cin, 1e-5, _ops.ACT_RELU, S()))
    assert L.p2phd_conv_dgrad_bsum_ok(C.byref(d))
    a, b, c_, e = t_us(dg), t_us(two), t_us(dgb), t_us(app)
    print(f"{name:34s} dgrad {a:6.0f} + two-pass {b:6.0f} = {a+b:6.0f} us | dgrad+sums {c_:6.0f} + apply {e:6.0f} = {c_+e:6.0f} us", flush=True)

case("c7in-out 512x256x48 <- down1", 48, 96, 3, 2, 1, False, 0, 512, 256)
case("down1-out 256x128x96 <- down2", 96, 192, 3, 2, 1, False, 0, 256, 128)
case("down2-out 128x64x192 <- down3", 192, 384, 3, 2, 1, False, 0, 128, 64)
case("down3-out 64x32x384 <- down4", 384, 768, 3, 2, 1, False, 0, 64, 32)
case("up1-out 64x32x384 <- up2 (convT)", 384, 192, 3, 2, 1, True, 1, 64, 32)
case("up2-out 128x64x192 <- up3 (convT)", 192, 96, 3, 2, 1, True, 1, 128, 64)
case("up3-out 256x128x96 <- up4 (convT)", 96, 48, 3, 2, 1, True, 1, 256, 128)

print("discriminator (LeakyReLU planes):")
case("D 64->128 out 129x65x128 <- 128->256", 128, 256, 4, 2, 2, False, 0, 129, 65)
case("D 128->256 out 65x33x256 <- 256->512", 256, 512, 4, 1, 2, False, 0, 65, 33)
case("D 256->512 out 66x34x512 <- 512->1", 512, 1, 4, 1, 2, False, 0, 66, 34)
